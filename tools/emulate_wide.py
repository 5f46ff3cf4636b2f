"""CPU emulation of the wide-column int8 kernel's arithmetic (tables from operators.cpp) against scipy's decimate.
Development aid: validates sdrt_build_wide_tables / sdrt_build_i8_weights_wide and the algorithm of kernels_i8w.hip."""
import ctypes as C
import sys
import numpy as np
from scipy.signal import cheby1, decimate

import os
_DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'sdrterm_amd', 'libsdrterm_hip.so')
lib = C.CDLL(sys.argv[1] if __name__ == '__main__' and len(sys.argv) > 1 else _DEFAULT)     # host-side table builders only
lib.sdrt_build_wide_edge.restype = None
LC, TB, NM, EDGE = 256, 16, 4, 27


class WideTables(C.Structure):
    _fields_ = [('q', C.c_int32), ('nout', C.c_int32), ('boff', C.c_int32), ('nout_total', C.c_int32),
                ('mu_pow', C.c_double * (4 * NM * 2)), ('tile_pow', C.c_double * (17 * NM * 2)),
                ('bp2', C.c_double * (18 * NM * 2)), ('cF2', C.c_double * (TB * 8 * NM * 2)), ('cG2', C.c_double * (TB * 8 * NM * 2)),
                ('head', C.c_double * (NM * 28 * 2)), ('tailF', C.c_double * (NM * NM * 2)), ('tailFc', C.c_double * (NM * NM * 2)),
                ('tailU', C.c_double * (NM * 28 * 2))]


def cplx(a, shape):
    v = np.ctypeslib.as_array(a).reshape(*shape, 2)
    return v[..., 0] + 1j * v[..., 1]


def run(q, n, theta, seed=1, S=5, lam=0.0, off0=0j):
    """Decimations below 32: a column has more than eight outputs and the kernel takes them eight per pass (tables and local planes per
    pass: sdrt_build_wide_tables_pass, sdrt_build_i8_weights_wide_pass); the passes' outputs interleave to the chunk's."""
    nout_total = LC // q
    nout = min(8, nout_total)
    parts = [run_pass(q, n, theta, seed, S, lam, off0, nout, p * nout) for p in range(nout_total // nout)]
    ref = parts[0][1]
    y = np.stack([pt[0] for pt in parts], axis=1).reshape(-1, nout_total // nout, nout)       # [column][pass][b]
    y = y.reshape(-1)
    e = np.abs(y - ref).max() / np.abs(ref).max()
    print(f'q={q} n={n} theta={theta}: max rel err {e:.3e} ({len(parts)} pass(es))')
    return e


def run_pass(q, n, theta, seed, S, lam, off0, nout, boff):
    rng = np.random.default_rng(seed)
    sos = np.ascontiguousarray(cheby1(8, 0.05, 0.8 / q, output='sos'))
    wt = WideTables()
    err = C.c_char_p()
    rc = lib.sdrt_build_wide_tables_pass(sos.ctypes.data_as(C.c_void_p), 4, q, C.c_double(1.0), nout, boff, C.byref(wt), C.byref(err))
    assert rc == 0, err.value
    assert (wt.nout, wt.boff, wt.nout_total) == (nout, boff, LC // q)
    NR = 2 * nout
    KS = 8
    t = np.arange(LC)
    samp = np.exp(1j * theta * t)
    samp_ri = np.ascontiguousarray(np.stack([samp.real, samp.imag], -1))
    main = np.zeros(2 * S * KS * 64 * 16, np.int8)
    loc = np.zeros(S * KS * 4 * NR * 16, np.int8)
    cm, cl = C.c_double(), C.c_double()
    gvec = np.zeros((60, 2))                 # SDRT_WIDE_GV: 48 vector pairs + 48 ints
    rc = lib.sdrt_build_i8_weights_wide_pass(sos.ctypes.data_as(C.c_void_p), 4, q, C.c_double(1.0),
                                             samp_ri.ctypes.data_as(C.c_void_p) if theta else None, S, C.c_double(lam), nout, boff,
                                             main.ctypes.data_as(C.c_void_p), C.byref(cm), loc.ctypes.data_as(C.c_void_p), C.byref(cl),
                                             gvec.ctypes.data_as(C.c_void_p))
    assert rc == 0
    # de-quantise into float weight matrices W[row][2*LC] by K position
    main = main.reshape(2, S, KS, 4, 16, 16)      # tile, digit, kstep, kg, row, j
    loc = loc.reshape(S, KS, 4, NR, 16)
    Wm = np.zeros((32, LC, 2))
    Wl = np.zeros((NR, LC, 2))
    for ks in range(KS):
        for kg in range(4):
            for j in range(16):
                s_, c_ = 32 * ks + 8 * kg + (j >> 1), j & 1
                for d in range(S):
                    Wm[:16, s_, c_] += main[0, d, ks, kg, :, j].astype(float) * 256.0 ** d
                    Wm[16:, s_, c_] += main[1, d, ks, kg, :, j].astype(float) * 256.0 ** d
                    Wl[:, s_, c_] += loc[d, ks, kg, :, j].astype(float) * 256.0 ** d
    Wm *= cm.value
    Wl *= cl.value
    x = (rng.integers(-20000, 20000, n) + 1j * rng.integers(-20000, 20000, n)).astype(np.complex128)
    ncol = n // LC
    nt = ncol // TB
    X = x.reshape(ncol, LC)
    blk = np.exp(1j * theta * LC * np.arange(ncol))
    # IQ correction reference: z[i] = x[i] - off; off += z[i]*lam  (read_file.py:64-77), offsets at column starts
    offs = np.zeros(n + 1, complex)
    zc = x.copy()
    if lam:
        off = off0
        for i in range(n):
            offs[i] = off
            zc[i] = x[i] - off
            off += zc[i] * lam
        offs[n] = off
    # stage A
    A = np.einsum('rt,ct->cr', Wm[:, :, 0], X.real) + np.einsum('rt,ct->cr', Wm[:, :, 1], X.imag)     # [col][32]
    Lv = np.einsum('rt,ct->cr', Wl[:, :, 0], X.real) + np.einsum('rt,ct->cr', Wl[:, :, 1], X.imag)    # [col][NR]
    if lam:
        d = -offs[np.arange(ncol) * LC]                # (c - off) with c = 0 here
        A += d.real[:, None] * gvec[None, :32, 0] + d.imag[:, None] * gvec[None, :32, 1]
        Lv += d.real[:, None] * gvec[None, 32:32 + NR, 0] + d.imag[:, None] * gvec[None, 32:32 + NR, 1]
    # rows: 16*kind + 4*mode + 2*stream + part
    A = A.reshape(ncol, 2, 4, 2, 2)
    st = A[..., 0] + 1j * A[..., 1]                # [col][kind][mode][stream] complex states
    # column phasor rotates the data: stream' = Re/Im of psi * (s_re + i s_im) componentwise over the complex states
    sr, si = st[..., 0], st[..., 1]
    pr, pi = blk.real[:, None, None], blk.imag[:, None, None]
    st = np.stack([pr * sr - pi * si, pi * sr + pr * si], -1)
    Lv = Lv.reshape(ncol, nout, 2)
    lr, li = Lv[..., 0], Lv[..., 1]
    Lv = np.stack([blk.real[:, None] * lr - blk.imag[:, None] * li, blk.imag[:, None] * lr + blk.real[:, None] * li], -1)
    mu = cplx(wt.mu_pow, (4, NM))[0]              # p^LC
    tp = cplx(wt.tile_pow, (17, NM))
    bp2 = cplx(wt.bp2, (18, NM))          # [b]: from the column start; [nout + 1 + b]: from the column end
    cF2 = cplx(wt.cF2, (TB * 8, NM))
    cG2 = cplx(wt.cG2, (TB * 8, NM))
    head = cplx(wt.head, (NM, 28))
    tailF = cplx(wt.tailF, (NM, NM))
    tailFc = cplx(wt.tailFc, (NM, NM))
    tailU = cplx(wt.tailU, (NM, 28))
    st = st.reshape(nt, TB, 2, 4, 2)
    # in-tile inclusive scans
    Cin = np.zeros((nt, TB, 4, 2), complex)
    Ain = np.zeros((nt, TB, 4, 2), complex)
    for c in range(TB):
        Cin[:, c] = st[:, c, 0] + (mu[None, :, None] * Cin[:, c - 1] if c else 0)
    for c in range(TB - 1, -1, -1):
        Ain[:, c] = st[:, c, 1] + (mu[None, :, None] * Ain[:, c + 1] if c < TB - 1 else 0)
    Cx = np.zeros_like(Cin)
    Cx[:, 1:] = Cin[:, :-1]
    Ax = np.zeros_like(Ain)
    Ax[:, :-1] = Ain[:, 1:]
    aggF, aggG = Cin[:, TB - 1], Ain[:, 0]         # [tile][mode][stream]
    yl = np.zeros((nt, TB, nout, 2))
    for b in range(nout):
        yl[:, :, b, :] = Lv.reshape(nt, TB, nout, 2)[:, :, b, :] + \
            (bp2[b][None, None, :, None] * Cx + bp2[nout + 1 + b][None, None, :, None] * Ax).real.sum(axis=2)
    # edge operators: the decimator sees u = z e^{i theta t}
    u = zc * np.exp(1j * theta * np.arange(n))
    blk_ri = np.ascontiguousarray(np.stack([blk.real, blk.imag], -1))
    ab = np.zeros((16, 32, 4))
    pq = np.zeros((16, 4))
    lib.sdrt_build_wide_edge(C.byref(wt), samp_ri.ctypes.data_as(C.c_void_p) if theta else None,
                             blk_ri.ctypes.data_as(C.c_void_p), n, C.c_double(lam), ab.ctypes.data_as(C.c_void_p),
                             pq.ctypes.data_as(C.c_void_p))
    Aj = ab[..., 0] + 1j * ab[..., 1]
    Bj = ab[..., 2] + 1j * ab[..., 3]
    Pq = pq[:, 0] + 1j * pq[:, 1]
    Qq = pq[:, 2] + 1j * pq[:, 3]
    F0 = np.zeros((4, 2), complex)
    Gu = np.zeros((4, 2), complex)
    for o16 in range(16):
        kind, o = o16 >> 3, o16 & 7
        win = x[:32] if kind == 0 else x[n - 32:]
        ow = offs[0] if kind == 0 else offs[n - 32]
        val = (Aj[o16] * win.real).sum() + (Bj[o16] * win.imag).sum() - (Pq[o16] * ow.real + Qq[o16] * ow.imag)
        (F0 if kind == 0 else Gu)[o >> 1, o & 1] = val
    F = np.zeros((nt + 1, 4, 2), complex)
    F[0] = F0
    for t_ in range(nt):
        F[t_ + 1] = tp[1][:, None] * F[t_] + aggF[t_]
    Gend = Gu + np.einsum('ij,js->is', tailF, F[nt]) + np.einsum('ij,js->is', tailFc, np.conj(F[nt]))
    G = np.zeros((nt + 1, 4, 2), complex)
    G[nt] = Gend
    for t_ in range(nt - 1, -1, -1):
        G[t_] = tp[1][:, None] * G[t_ + 1] + aggG[t_]
    y = np.zeros((nt, TB, nout, 2))
    for c in range(TB):
        for b in range(nout):
            pos = c * nout + b
            y[:, c, b, :] = yl[:, c, b, :] + (cF2[pos][None, :, None] * F[:-1] + cG2[pos][None, :, None] * G[1:]).real.sum(axis=1)
    y = (y[..., 0] + 1j * y[..., 1]).reshape(-1, nout)           # [column][b]
    return y, decimate(u, q)


if __name__ == '__main__':
    ok = True
    for q, n, th in [(64, 32768, 0.0), (64, 32768, 0.37), (32, 32768, 0.1), (64, 65536, -1.1), (128, 32768, 0.2), (256, 32768, 0.0), (16, 32768, 0.3),
                     (8, 32768, -0.4), (4, 32768, 0.0)]:
        ok &= run(q, n, th) < 2e-11
    for q, n, th in [(64, 32768, 0.37), (32, 32768, -0.2)]:
        ok &= run(q, n, th, lam=50 / 1024000, off0=300 - 200j) < 2e-11
        ok &= run(q, n, th, lam=0.01, off0=-1000 + 50j) < 2e-11
    print('OK' if ok else 'FAIL')
