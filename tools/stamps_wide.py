#!/usr/bin/env python3
"""Diagnostic: phase shares of one steady-state iteration of chain_i8w_kernel from in-kernel s_memtime stamps (GPU box)."""
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
    code = sys.argv[4] if len(sys.argv) > 4 else 'h'          # 'f': float32 records (16384-sample chunks, two to an item)
    iq = len(sys.argv) > 5 and sys.argv[5] == 'iq'               # --correct-iq (fused form)
    eng = Engine(256_000_000, 64, 'fm', code, row_freqs=[center], omega_out=12500, correct_iq=iq)
    assert eng.stage_a == 'i8w', eng.stage_a
    one = np.frombuffer(prng.raw_bytes(code, eng.n * 64, 3, 8000), dtype=np.uint8)
    d_raw = nat.DeviceBuffer(n_chunks * eng.chunk_bytes)
    for i in range(0, n_chunks, 64):
        nat.check(nat.lib().sdrt_memcpy_h2d(d_raw.ptr + i * eng.chunk_bytes, one.ctypes.data, min(64, n_chunks - i) * eng.chunk_bytes))
    d_out = nat.DeviceBuffer(n_chunks * eng.m * 8)
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # e.g. 1800 launches of 2^30 samples = 2 s of load before the stamped launch
    for _ in range(warm):
        eng.process_device(d_raw.ptr, n_chunks, d_out.ptr)
    nat.check(nat.lib().sdrt_device_synchronize())
    nw = 512                                   # stamp slots: 4 waves x 16 each -> 256 workgroups of 8 waves
    st = np.zeros((nw, 4, 16), dtype=np.uint64)
    lib = nat.lib()
    lib.sdrt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    nat.check(lib.sdrt_debug_stamps(eng._ctx, d_raw.ptr, n_chunks, d_out.ptr, st.ctypes.data, nw))
    s = st.reshape(nw // 2, 8, 16).astype(np.int64)
    valid = s[:, :, 0] > 0
    print('workgroups stamped', int(valid[:, 0].sum()))
    def show(nm, a_, b_, sel=None):
        d = (s[:, :, b_] - s[:, :, a_])
        ok = valid & (s[:, :, a_] > 0) & (s[:, :, b_] > 0)
        if sel is not None:
            ok = ok & sel
        d = d[ok]
        print(f'{nm:34s} median {np.median(d):9.0f}  mean {d.mean():9.0f}  p90 {np.percentile(d, 90):9.0f}')
    show('P2 carries (2 of 8 waves)', 0, 1)
    show('K + pieces (tile slot 0)', 1, 2)
    show('recombination + queue', 2, 3)
    show('barrier wait', 3, 4)
    show('iteration', 0, 4)
    if 'wide_dual=1' in os.environ.get('SDRTERM_DIAG', ''):
        for a_, b_, nm in ((1, 5, 'top -> k-step 0'), (5, 6, 'k-steps 0-1'), (6, 7, 'k-steps 2-3'), (7, 8, 'k-steps 4-5'), (8, 2, 'k-steps 6-7')):
            show('DUAL ' + nm, a_, b_)
    k = (s[:, :, 2] - s[:, :, 1])
    print('K by wave (median):', np.median(k, axis=0))
    core = (s[:, :, 12] - s[:, :, 10])[valid]
    real = (s[:, :, 13] - s[:, :, 11])[valid]
    cnt = s[:, :, 14][valid]
    print(f'in-kernel clock: {np.median(core / real) * 100:.0f} MHz (core cycles per 100 MHz real-time tick x 100), '
          f'{np.median(core / cnt):.0f} core cycles and {np.median(real / cnt) * 10:.0f} ns per item and workgroup')


if __name__ == '__main__':
    main()
