#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of chain_kernel from in-kernel s_memtime stamps (run on the GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrterm_amd import _native as nat
from sdrterm_amd.engine import Engine
from tests.golden import prng

def main():
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    center = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
    eng = Engine(256_000_000, 64, 'fm', 'h', row_freqs=[center], omega_out=12500)
    one = np.frombuffer(prng.raw_bytes('h', 32768 * 64, 3, 8000), dtype=np.uint8)
    d_raw = nat.DeviceBuffer(n_chunks * eng.chunk_bytes)
    for i in range(0, n_chunks, 64):
        nat.check(nat.lib().sdrt_memcpy_h2d(d_raw.ptr + i * eng.chunk_bytes, one.ctypes.data, min(64, n_chunks - i) * eng.chunk_bytes))
    d_out = nat.DeviceBuffer(n_chunks * eng.m * 8)
    for _ in range(3):
        eng.process_device(d_raw.ptr, n_chunks, d_out.ptr)
    nat.check(nat.lib().sdrt_device_synchronize())
    nw = min(n_chunks, 4096)
    st = np.zeros((nw, 4, 16), dtype=np.uint64)
    lib = nat.lib()
    lib.sdrt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    nat.check(lib.sdrt_debug_stamps(eng._ctx, d_raw.ptr, n_chunks, d_out.ptr, st.ctypes.data, nw))
    s = st.astype(np.int64)
    names = ['prologue', 'main loop', 'tables+edge+barrier', 'phase2 carries', 'phase3 combine+store']
    valid = s[:, :, 0] > 0
    print('workgroups stamped', int(valid[:, 0].sum()))
    for i, nm in enumerate(names):
        d = (s[:, :, i + 1] - s[:, :, i])[valid]
        print(f'{nm:28s} median {np.median(d):9.0f}  mean {d.mean():9.0f} cycles')
    if (s[:, :, 13] > 0).any():
        inner = ['pass0 k-loop (issue)', 'pass0 combine (MFMA drain)', 'pass1 k-loop (issue)', 'pass1 combine', 'phasor+scans', 'aggregates+outputs']
        pts = [1, 8, 9, 10, 11, 12, 13]
        for i, nm in enumerate(inner):
            d = (s[:, :, pts[i + 1]] - s[:, :, pts[i]])[valid]
            print(f'  first super-tile: {nm:28s} median {np.median(d):9.0f}  mean {d.mean():9.0f} cycles')
    tot = (s[:, :, 5] - s[:, :, 0])[valid]
    print(f'{"start -> end":28s} median {np.median(tot):9.0f}  mean {tot.mean():9.0f} cycles')

if __name__ == '__main__':
    main()
