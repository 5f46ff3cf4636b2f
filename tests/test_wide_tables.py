"""CPU check of the host-built operator tables of the wide-column kernel (sdrterm_amd/csrc/operators.cpp): a numpy
emulation of the kernel's arithmetic from those tables (tools/emulate_wide.py: digit planes de-quantised, column
aggregates, in-tile scans, tile carries, edge operators, IQ correction folded into the planes + rank-one term) must
reproduce scipy.signal.decimate, i.e. what the reference computes at dsp_processor.py:147 (and read_file.py:64-77)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.parametrize('q,n,theta,lam,off0', [
    (64, 32768, 0.0, 0.0, 0j),
    (64, 32768, 0.37, 0.0, 0j),
    (32, 32768, 0.1, 0.0, 0j),
    (64, 65536, -1.1, 0.0, 0j),
    (64, 32768, 0.37, 50 / 1024000, 300 - 200j),
    (32, 32768, -0.2, 0.01, -1000 + 50j),
    # decimation 16 / 8 / 4: 16 / 32 / 64 outputs per column in 2 / 4 / 8 passes of eight (sdrt_build_*_pass)
    (16, 32768, 0.3, 0.0, 0j),
    (8, 32768, -0.4, 0.0, 0j),
    (4, 32768, 0.0, 0.0, 0j),
    (16, 65536, 0.2, 0.0, 0j),
])
def test_tables_reproduce_scipy_decimate(q, n, theta, lam, off0):
    import emulate_wide
    assert emulate_wide.run(q, n, theta, lam=lam, off0=off0) < 2e-11


@pytest.mark.parametrize('q,theta,nout,boff', [(64, 0.37, 4, 0), (32, -0.2, 8, 0), (64, 0.0, 4, 0)])
def test_iq_planes_are_quantised_consistently(q, theta, nout, boff):
    """--correct-iq forms of the wide kernel (operators.cpp: quantise_planes, "consistent"): what makes a stream that starts silent
    cancel to exact zeros.  From the digit planes the builder emits: every row's and component's integer weight sum is a multiple of 16
    (32896 x sum stays an exact double), the rank-one vectors are exactly combine x those sums, and the 48 ints behind them are minus
    half the sum of the row's lowest digits (what the 16-bit forms open their lowest accumulators at); with that an all-zero column
    (low slices -128, high slices 0) cancels exactly in the arithmetic the kernel performs."""
    import ctypes as C
    import numpy as np
    from scipy.signal import cheby1
    import emulate_wide
    lib = emulate_wide.lib
    LC, S, KS = 256, 5, 8
    NR = 2 * nout
    sos = np.ascontiguousarray(cheby1(8, 0.05, 0.8 / q, output='sos'))
    samp = np.exp(1j * theta * np.arange(LC))
    samp_ri = np.ascontiguousarray(np.stack([samp.real, samp.imag], -1))
    main = np.zeros(2 * S * KS * 64 * 16, np.int8)
    loc = np.zeros(S * KS * 4 * NR * 16, np.int8)
    cm, cl = C.c_double(), C.c_double()
    gvec = np.zeros((60, 2))
    rc = lib.sdrt_build_i8_weights_wide_pass(sos.ctypes.data_as(C.c_void_p), 4, q, C.c_double(1.0),
                                             samp_ri.ctypes.data_as(C.c_void_p) if theta else None, S, C.c_double(50 / 1024000), nout, boff,
                                             main.ctypes.data_as(C.c_void_p), C.byref(cm), loc.ctypes.data_as(C.c_void_p), C.byref(cl),
                                             gvec.ctypes.data_as(C.c_void_p))
    assert rc == 0
    opening = gvec.view(np.int32).reshape(-1)[4 * 48:4 * 48 + 48]           # 48 ints behind the 48 pairs of doubles
    mainp = main.reshape(2, S, KS, 4, 16, 16).astype(np.int64)             # tile, digit, kstep, kg, row, j
    locp = loc.reshape(S, KS, 4, NR, 16).astype(np.int64)
    pw = 256 ** np.arange(S, dtype=np.int64)
    for name, rows, digits, combine, g0 in (('aggregate', 32, lambda r: mainp[r // 16, :, :, :, r % 16, :], cm.value, 0),
                                            ('local', NR, lambda r: locp[:, :, :, r, :], cl.value, 32)):
        for r in range(rows):
            d = digits(r)                                                   # [digit][kstep][kg][j], j = 2 (sample % 8) + component
            for comp in (0, 1):
                isum = int((d[..., comp::2].sum(axis=(1, 2, 3)) * pw).sum())
                assert isum % 16 == 0, (name, r, comp, isum)
                assert gvec[g0 + r, comp] == combine * isum, (name, r, comp)
            low = int(d[0].sum())
            assert low % 2 == 0 and opening[g0 + r] == -low // 2, (name, r, low, opening[g0 + r])
            # an all-zero int16 column: high slices 0, low slices -128; the level of (lowest digit x low slice) is not computed, the
            # accumulators of the level above open at `opening` (units of 256): everything in integers, then the rank-one term
            levels = -128 * d[1:].sum(axis=(1, 2, 3))                       # digit d >= 1 x low slice -> level d
            total = int((levels * pw[1:]).sum()) + 256 * int(opening[g0 + r])
            k = combine * float(total)
            assert k + 128.0 * gvec[g0 + r, 1] + 128.0 * gvec[g0 + r, 0] == 0.0, (name, r)
    assert (opening[32 + NR:] == 0).all()
