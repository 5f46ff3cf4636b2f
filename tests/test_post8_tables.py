"""CPU checks of the eight-lanes-per-item FM demodulation stage (sdrterm_amd/csrc/kernels_post8.hip):
  * the host-built tables (operators.cpp: sdrt_build_post8_tables) against the numpy model of the kernel's resampler
    (tools/model_post8.py), which itself reproduces scipy.signal.resample -- what demodulation.py:35-38 calls;
  * the lowpass constants against scipy.signal.sosfilt on one block structure (dsp_processor.py:32-36);
  * the exchange image against the LDS bank rules (tools/lds_banks.py): every wave-instruction conflict free."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


class Post8Tables(C.Structure):
    _fields_ = [('N', C.c_int32), ('LPI', C.c_int32), ('reserved', C.c_int32 * 2),
                ('h', C.c_double * 4), ('pcp', C.c_double * 6), ('prp', C.c_double * 3),
                ('g', C.c_double * 8), ('gr', C.c_double * 4),
                ('pc4', C.c_double * 2), ('pr4', C.c_double), ('pc8', C.c_double * 2), ('pr8', C.c_double),
                ('pc32', C.c_double * 2), ('pr32', C.c_double),
                ('pc_q', C.c_double * 8), ('pr_q', C.c_double * 4),
                ('pc_lane', C.c_double * 32), ('pr_lane', C.c_double * 16),
                ('pc_blk', C.c_double * 2), ('pr_blk', C.c_double)]


def _build(N, LPI, sos):
    from sdrterm_amd import _native
    lib = C.CDLL(_native.LIB_PATH)
    pt = Post8Tables()
    tab = np.zeros((4 * 16 * LPI + 8 * LPI, 2))
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    rc = lib.sdrt_build_post8_tables(sos.ctypes.data_as(C.c_void_p), sos.shape[0], N, LPI, C.byref(pt), tab.ctypes.data_as(C.c_void_p))
    tab = tab[..., 0] + 1j * tab[..., 1]
    return rc, pt, tab[:4 * 16 * LPI].reshape(4, 16 * LPI), tab[4 * 16 * LPI:].reshape(8, LPI)


@pytest.mark.parametrize('N,LPI', [(128, 4), (256, 8), (512, 16)])
def test_transform_tables_match_the_model(N, LPI):
    import model_post8
    from sdrterm_amd.engine import output_sos
    rc, _pt, tab, _lanes = _build(N, LPI, output_sos(4000000, 12500))
    assert rc == 0
    tw1, tw2, alpha, beta = model_post8.tables(N, LPI)
    for got, exp in zip(tab, (tw1, tw2, alpha, beta)):
        assert np.abs(got - exp.reshape(-1)).max() < 2e-15 * np.abs(exp).max()      # (numpy evaluates its phasors in double)
    # ... and the model is scipy's resampler
    from scipy.signal import resample
    a = np.random.default_rng(5).uniform(-np.pi, np.pi, N)
    ref = resample(a, 2 * N)
    assert np.abs(model_post8.resample_odd(a, LPI) - ref[1::2]).max() < 1e-13
    assert np.abs(ref[0::2] - a).max() < 1e-13


@pytest.mark.parametrize('fs_dec,omega', [(4000000, 12500), (16000, 5000), (8000000, 12500)])
def test_lowpass_constants_reproduce_sosfilt(fs_dec, omega):
    """The kernel's block form of the output lowpass from the table constants, in numpy, against scipy.signal.sosfilt."""
    from scipy.signal import sosfilt
    from sdrterm_amd.engine import output_sos
    sos = output_sos(fs_dec, omega)
    rc, pt, _, lanes = _build(256, 8, sos)
    assert rc == 0
    cx = lambda a, i=0: complex(a[2 * i], a[2 * i + 1])
    x = np.random.default_rng(3).normal(size=512)
    ref = sosfilt(sos, x)
    LPI = 8
    out = np.zeros(512)
    Cc, Cr = 0j, 0.0
    for n1 in range(16):
        se, te, y0 = [], [], []
        for l in range(LPI):
            xs = x[32 * n1 + 4 * l: 32 * n1 + 4 * l + 4]
            y0.append([sum(pt.h[j - i] * xs[i] for i in range(j + 1)) for j in range(4)])
            se.append(xs[3] + cx(pt.pcp, 0) * xs[2] + cx(pt.pcp, 1) * xs[1] + cx(pt.pcp, 2) * xs[0])
            te.append(xs[3] + pt.prp[0] * xs[2] + pt.prp[1] * xs[1] + pt.prp[2] * xs[0])
        # inclusive scan as the kernel does it: inside quads, then quad 0's total into quad 1
        sc, sr = list(se), list(te)
        for step in (1, 2):         # the lane rows of the table: multipliers, zero where the quad permutation wraps
            sc = [sc[l] + lanes[step - 1, l] * sc[l - step if (l & 3) >= step else l] for l in range(LPI)]
            sr = [sr[l] + (lanes[2, l].real, lanes[2, l].imag)[step - 1] * sr[l - step if (l & 3) >= step else l] for l in range(LPI)]
        # state before the lane's first sample: the lane before it in the quad + the quad before + the block's starting state
        ex = [lanes[7, l].imag * sc[l - 1] + (lanes[4, l] * sc[3] if l >= 4 else 0) for l in range(LPI)]
        exr = [lanes[7, l].imag * sr[l - 1] + (lanes[5, l].imag * sr[3] if l >= 4 else 0) for l in range(LPI)]
        sc = [sc[l] + (lanes[3, l] * sc[3] if l >= 4 else 0) for l in range(LPI)]
        sr = [sr[l] + (lanes[5, l].real * sr[3] if l >= 4 else 0) for l in range(LPI)]
        for l in range(LPI):
            s = lanes[6, l] * Cc + ex[l]
            t = lanes[7, l].real * Cr + exr[l]
            for j in range(4):
                out[32 * n1 + 4 * l + j] = y0[l][j] + (cx(pt.g, j) * s).real + pt.gr[j] * t
        Cc = cx(pt.pc_blk) * Cc + sc[7]
        Cr = pt.pr_blk * Cr + sr[7]
    assert np.abs(out - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def test_builder_refuses_other_lowpass_shapes():
    from scipy.signal import butter
    assert _build(256, 8, butter(4, 0.1, output='sos'))[0] != 0          # two complex pole pairs
    from sdrterm_amd.engine import output_sos
    assert _build(300, 8, output_sos(16000, 5000))[0] != 0


def test_exchange_image_is_conflict_free():
    from lds_banks import cycles
    SO, LPI = 136, 8
    lanes = [(lane >> 3, lane & 7) for lane in range(64)]
    S = lambda l, ci: l if ci == 0 else (16 - l if l else 8)
    slot = lambda o, n2, k1: 16 * (o * SO + 17 * n2 + k1)          # (SO is looked up when called: the second half redefines it)
    for k1 in range(16):                                     # forward: lane writes (n2 = l, k1); back: lane reads (n2 = l, c = k1)
        assert cycles('write_b128', [slot(o, l, k1) for o, l in lanes]) == 8
        assert cycles('read_b128', [slot(o, l, k1) for o, l in lanes]) == 4
    for ci in range(2):
        for j in range(8):                                   # forward: lane reads (n2 = j, c); back: lane writes (n2 = j, c)
            assert cycles('read_b128', [slot(o, j, S(l, ci)) for o, l in lanes]) == 4
            assert cycles('write_b128', [slot(o, j, S(l, ci)) for o, l in lanes]) == 8
    # the tables: every lane group reads the same eight consecutive entries
    for row in range(16):
        assert cycles('read_b128', [16 * (row * LPI + l) for _o, l in lanes]) == 4
    # sixteen lanes per item (1024 outputs per chunk): an item's image is 16 x 17 = 272 slots, one bin series per lane
    SO, LPI = 272, 16
    lanes = [(lane >> 4, lane & 15) for lane in range(64)]
    for k1 in range(16):
        assert cycles('write_b128', [slot(o, l, k1) for o, l in lanes]) == 8
        assert cycles('read_b128', [slot(o, l, k1) for o, l in lanes]) == 4
    for j in range(16):
        assert cycles('read_b128', [slot(o, j, l) for o, l in lanes]) == 4
        assert cycles('write_b128', [slot(o, j, l) for o, l in lanes]) == 8
    # four lanes per item (256 outputs per chunk): 4 x 17 = 68 slots, four bin series per lane
    SO, LPI = 68, 4
    lanes = [(lane >> 2, lane & 3) for lane in range(64)]
    S4 = lambda l, ci: [l, (16 - l) if l else 8, 4 + l, 12 - l][ci]
    for k1 in range(16):
        assert cycles('write_b128', [slot(o, l, k1) for o, l in lanes]) == 8
        assert cycles('read_b128', [slot(o, l, k1) for o, l in lanes]) == 4
    for ci in range(4):
        for j in range(4):
            assert cycles('read_b128', [slot(o, j, S4(l, ci)) for o, l in lanes]) == 4
            assert cycles('write_b128', [slot(o, j, S4(l, ci)) for o, l in lanes]) == 8
