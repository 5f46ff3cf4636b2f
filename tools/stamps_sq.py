#!/usr/bin/env python3
"""Diagnostic: phase lengths of one steady-state chunk of chain_sq_kernel from in-kernel s_memtime stamps (GPU box).
   tools/stamps_sq.py [decimation] [chunks] [waves per workgroup of the build] [encoding h / B]; needs a library built with EXTRA=-DSDRT_STAMPS
   (make -C sdrterm_amd/csrc OUT=../libsdrterm_hip_exp.so OBJ=_obj_exp EXTRA=-DSDRT_STAMPS; SDRTERM_HIP_LIB= that file)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrterm_amd import _native as nat
from sdrterm_amd.engine import Engine
from tests.golden import prng


def main():
    q = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    wv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    code = sys.argv[4] if len(sys.argv) > 4 else 'h'          # 'B': uint8 records (65536-sample chunks, two pairs of tiles per wave: the stamps are the last pair's)
    eng = Engine(256_000_000, q, 'fm', code, row_freqs=[15000], omega_out=12500)
    assert eng.stage_a == 'sq', eng.stage_a
    one = np.frombuffer(prng.raw_bytes(code, eng.n * 64, 3, 100 if code in 'bB' else 8000), dtype=np.uint8)
    d_raw = nat.DeviceBuffer(n_chunks * eng.chunk_bytes)
    for i in range(0, n_chunks, 64):
        nat.check(nat.lib().sdrt_memcpy_h2d(d_raw.ptr + i * eng.chunk_bytes, one.ctypes.data, min(64, n_chunks - i) * eng.chunk_bytes))
    d_out = nat.DeviceBuffer(n_chunks * eng.m * 8)
    for _ in range(20):
        eng.process_device(d_raw.ptr, n_chunks, d_out.ptr)
    nat.check(nat.lib().sdrt_device_synchronize())
    nwg = 256
    nw = nwg * wv // 4                                    # stamp slots come in fours
    st = np.zeros((nw, 4, 16), dtype=np.uint64)
    lib = nat.lib()
    lib.sdrt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    nat.check(lib.sdrt_debug_stamps(eng._ctx, d_raw.ptr, n_chunks, d_out.ptr, st.ctypes.data, nw))
    s = st.reshape(nwg, wv, 16).astype(np.int64)
    valid = s[:, :, 0] > 0
    print(f'decimation {q}, {n_chunks} chunks, {wv} waves per workgroup; waves stamped', int(valid.sum()), '(100 MHz ticks x 10 = ns)' if False else '')

    def show(nm, a_, b_):
        ok = valid & (s[:, :, a_] > 0) & (s[:, :, b_] > 0)
        d = (s[:, :, b_] - s[:, :, a_])[ok]
        if d.size:
            print(f'{nm:44s} median {np.median(d):9.0f}  mean {d.mean():9.0f}  p90 {np.percentile(d, 90):9.0f}   (s_memtime ticks)')
    show('E edge operators', 0, 1)
    show('A1 loads + aggregates + in-tile scans', 1, 2)
    show('barrier', 2, 3)
    show('P2 tile carries (wave 0 works)', 3, 4)
    show('barrier', 4, 5)
    show('A2 set-up (states of the columns), tile 0', 5, 6)
    show('A2 row tiles, tile 0', 6, 7)
    show('after A2 -> barrier', 7, 10)
    for nm, a_, b_ in (('row tile 3, quarter 0 (+ recombination)', 8, 9), ('quarter 1 (+ states\' shares, first half)', 9, 12), ('quarter 2 (+ second half, rows)', 12, 13), ('quarter 3 (+ ybias, store)', 13, 14)):
        show(nm, a_, b_)
    show('barrier', 10, 11)
    show('chunk', 0, 11)


if __name__ == '__main__':
    main()
