"""numpy model of the one-pass small-decimation form (csrc/kernels_sq.hip): columns of 256 samples, the local Toeplitz part of the
zero-phase decimator inside a column + the causal / anticausal modal states at the column boundaries (tables: sdrt_build_decim_tables),
against scipy.signal.decimate as oracle/chain.c restates it (the reference: src/dsp/dsp_processor.py:147).  Runs on the CPU.
   python tools/model_sq.py"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_chain, ref_chain

class DecimTables(C.Structure):
    _fields_ = [('L', C.c_int32), ('ksteps', C.c_int32), ('out_stride', C.c_int32), ('edge', C.c_int32), ('G', C.c_int32), ('reserved', C.c_int32),
                ('W', C.c_double * (16 * 64)), ('mu_pow', C.c_double * (5 * 4 * 2)), ('cF', C.c_double * (32 * 4 * 2)), ('cG', C.c_double * (32 * 4 * 2)),
                ('tpow', C.c_double * (65 * 4 * 2)), ('d0p', C.c_double), ('head', C.c_double * (4 * 28 * 2)), ('tailF', C.c_double * (4 * 4 * 2)),
                ('tailFc', C.c_double * (4 * 4 * 2)), ('tailU', C.c_double * (4 * 28 * 2)), ('pole', C.c_double * 8), ('ac', C.c_double * 8), ('aa', C.c_double * 8)]

def cplx(a, shape):
    v = np.array(a[:]).reshape(shape + (2,))
    return v[..., 0] + 1j * v[..., 1]

lib = C.CDLL(os.path.join(ROOT, 'sdrterm_amd', 'libsdrterm_hip.so'))
def tables(q):
    sos = np.ascontiguousarray(ref_chain.cheby_sos(q))
    t = DecimTables(); err = C.c_char_p()
    rc = lib.sdrt_build_decim_tables(sos.ctypes.data_as(C.c_void_p), 4, q, 1, C.byref(t), C.byref(err))
    assert rc == 0, err.value
    return t, sos

def model(x, q, theta, LC=256):
    """x complex (integer-valued) chunk of n samples; returns decimated samples via the column form."""
    n = x.size
    t, sos = tables(q)
    p, ac, aa = cplx(t.pole, (4,)), cplx(t.ac, (4,)), cplx(t.aa, (4,))
    head, tailU = cplx(t.head, (4, 28)), cplx(t.tailU, (4, 28))
    tailF, tailFc = cplx(t.tailF, (4, 4)), cplx(t.tailFc, (4, 4))
    d0p = t.d0p
    u = x * np.exp(1j * theta * np.arange(n))
    streams = [u.real, u.imag]
    NC = n // LC
    out = np.zeros(n // q, dtype=complex)
    for st, v in enumerate(streams):
        # column aggregates
        j = np.arange(LC)
        A = np.zeros((NC, 4), complex); B = np.zeros((NC, 4), complex)
        for c in range(NC):
            vc = v[c * LC:(c + 1) * LC]
            for i in range(4):
                A[c, i] = np.sum(ac[i] * p[i] ** (LC - 1 - j) * vc)
                B[c, i] = np.sum(aa[i] * p[i] ** j * vc)
        F = np.zeros((NC + 1, 4), complex); G = np.zeros((NC + 1, 4), complex)
        F[0] = head @ v[:28]
        for c in range(NC):
            F[c + 1] = p ** LC * F[c] + A[c]
        G[NC] = tailF @ F[NC] + tailFc @ np.conj(F[NC]) + tailU @ v[n - 28:]
        for c in range(NC - 1, -1, -1):
            G[c] = B[c] + p ** LC * G[c + 1]
        # local Toeplitz kernel h[d], d = j - s
        d = np.arange(-(LC - 1), LC)
        h = np.zeros(d.size)
        for i in range(4):
            h += np.where(d >= 0, 2 * (aa[i] * p[i] ** np.maximum(d, 0)).real, 2 * (ac[i] * p[i] ** np.maximum(-d - 1, 0)).real)
        h[LC - 1] += d0p
        y = np.zeros(n // q)
        for c in range(NC):
            vc = v[c * LC:(c + 1) * LC]
            for b in range(LC // q):
                s = b * q
                loc = np.sum(h[(j - s) + LC - 1] * vc)
                sh = 2 * np.sum((p ** s * F[c]).real) + 2 * np.sum((p ** (LC - s) * G[c + 1]).real)
                y[c * (LC // q) + b] = loc + sh
        out += y if st == 0 else 1j * y
    return out

def rel_err(q, n=4096, seed=3):
    """Largest deviation of the column form from the oracle's decimate, relative to the output's peak."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-3000, 3000, n) + 1j * rng.integers(-3000, 3000, n)
    theta = -2 * np.pi * 15700 / 1024000
    ref = c_chain.decimate(x * np.exp(1j * theta * np.arange(n)), q, ref_chain.cheby_sos(q))
    got = model(x, q, theta)
    return float(np.abs(got - ref).max() / np.abs(ref).max())


if __name__ == '__main__':
    for q in (2, 4, 8, 16):
        print(q, 'rel err', rel_err(q))
