// Host-side construction of the operator tables (tables.h) in 113-bit arithmetic (__float128).
//
// What is restated here is the operator scipy applies when the reference calls
//     y[:] = decimate(x, self._decimationFactor)            src/dsp/dsp_processor.py:147
// i.e. sosfiltfilt(cheby1(8, 0.05, 0.8/q, 'sos'), x)[::q] with its defaults: odd extension by 27 samples,
// sosfilt_zi initial conditions scaled by the first sample, forward pass, backward pass, strip, stride.
// Instead of running the recursion sample by sample, the forward-backward filter is expanded in partial
// fractions over its 8 poles p_i (4 conjugate pairs):
//     H(z) H(1/z)  =  d0 + sum_i alpha_i [ z^-1/(1 - p_i z^-1) + z/(1 - p_i z) ]
// so every output is   d0 u[t] + sum_i alpha_i ( phi_i(t) + Z_i(t) )   with a causal first-order state
// phi_i and an anticausal first-order state Z_i per pole.  The edge handling (odd extension and zi) turns
// into the initial values of phi_i at the left edge and of Z_i at the right edge, both linear in a few
// edge samples; those linear maps are obtained below by evaluating the literal definition on unit inputs.
// Everything is built in __float128 and rounded once to double, so table error is ~1e-16 relative.
#include "tables.h"

#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace {

typedef __float128 R;

struct C {
    R re, im;
    C() : re(0), im(0) {}
    C(R r) : re(r), im(0) {}
    C(R r, R i) : re(r), im(i) {}
};
inline C operator+(C a, C b) { return C(a.re + b.re, a.im + b.im); }
inline C operator-(C a, C b) { return C(a.re - b.re, a.im - b.im); }
inline C operator*(C a, C b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
inline C operator/(C a, C b) {
    R d = b.re * b.re + b.im * b.im;
    return C((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
}
inline C conj(C a) { return C(a.re, -a.im); }
inline C cpow(C a, int n) {
    C r(1);
    C b = a;
    while (n > 0) {
        if (n & 1) r = r * b;
        b = b * b;
        n >>= 1;
    }
    return r;
}
inline R rsqrt128(R x) {  // sqrt by Newton from a double seed
    if (x <= 0) return 0;
    R y = (R) std::sqrt((double) x);
    for (int i = 0; i < 4; ++i) y = (y + x / y) / 2;
    return y;
}
inline void put(double *dst, C v) {
    dst[0] = (double) v.re;
    dst[1] = (double) v.im;
}

struct Modal {
    C p[8], r[8], alpha[8], X[8][8];
    R D, d0;
};

// Poles, residues of the cascade H(z) = prod (b0 + b1 z^-1 + b2 z^-2)/(1 + a1 z^-1 + a2 z^-2) = D + sum r_i/(z - p_i),
// and the partial-fraction data of H(z)H(1/z).  Index 2i / 2i+1 = pole with Im > 0 / its conjugate.
bool build_modal(const double *sos, int nsec, Modal &m, const char **err) {
    if (nsec != 4) { if (err) *err = "decimator must have 4 second-order sections (order 8)"; return false; }
    R b[4][3], a[4][3];
    for (int s = 0; s < 4; ++s) {
        R a0 = sos[6 * s + 3];
        if (a0 == 0) { if (err) *err = "a0 == 0 in decimator SOS"; return false; }
        for (int k = 0; k < 3; ++k) { b[s][k] = (R) sos[6 * s + k] / a0; a[s][k] = (R) sos[6 * s + 3 + k] / a0; }
        R disc = a[s][1] * a[s][1] - 4 * a[s][2];
        if (!(disc < 0)) { if (err) *err = "decimator section has real poles (unsupported)"; return false; }
        R sq = rsqrt128(-disc);
        m.p[2 * s] = C(-a[s][1] / 2, sq / 2);
        m.p[2 * s + 1] = C(-a[s][1] / 2, -sq / 2);
    }
    m.D = 1;
    for (int s = 0; s < 4; ++s) m.D = m.D * b[s][0];
    for (int i = 0; i < 8; ++i) {
        C num(1), den(1), p = m.p[i];
        for (int s = 0; s < 4; ++s) num = num * (C(b[s][0]) * p * p + C(b[s][1]) * p + C(b[s][2]));
        for (int j = 0; j < 8; ++j) if (j != i) den = den * (p - m.p[j]);
        m.r[i] = num / den;
    }
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) m.X[i][j] = m.r[j] / (C(1) - m.p[i] * m.p[j]);
    C sumrho(0);
    for (int j = 0; j < 8; ++j) {
        C s(0);
        for (int i = 0; i < 8; ++i) s = s + m.r[i] / (C(1) - m.p[i] * m.p[j]);
        C rho = m.r[j] * s;
        m.alpha[j] = C(m.D) * m.r[j] + m.p[j] * rho;
        sumrho = sumrho + rho;
    }
    m.d0 = m.D * m.D + sumrho.re;
    return true;
}

}  // namespace

extern "C" int sdrt_build_decim_tables(const double *sos, int sections, int q, int G, DecimTables *out, const char **err) {
    if (G < 1 || G > SDRT_GMAX) { if (err) *err = "unsupported blocks-per-column"; return -3; }
    static const char *kBadQ = "decimation must be a power of two >= 2";
    if (q < 2 || (q & (q - 1))) { if (err) *err = kBadQ; return -1; }
    Modal m;
    if (!build_modal(sos, sections, m, err)) return -2;
    std::memset(out, 0, sizeof(*out));
    const int L = q < SDRT_MAXL ? q : SDRT_MAXL;
    out->L = L;
    out->ksteps = L / 4;
    out->out_stride = q / L;
    out->edge = SDRT_EDGE;
    out->G = G;

    C P[4], ac[4], aa[4], mu[4];
    C sum_a_over_p(0);
    for (int i = 0; i < 8; ++i) sum_a_over_p = sum_a_over_p + m.alpha[i] / m.p[i];
    out->d0p = (double) (m.d0 - sum_a_over_p.re);
    for (int i = 0; i < 4; ++i) {
        P[i] = m.p[2 * i];
        ac[i] = m.alpha[2 * i];
        aa[i] = m.alpha[2 * i] / P[i];
        mu[i] = cpow(P[i], L);
        put(out->pole[i], P[i]);
        put(out->ac[i], ac[i]);
        put(out->aa[i], aa[i]);
    }
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < L; ++j) {
            C wc = ac[i] * cpow(P[i], L - 1 - j);
            C wa = aa[i] * cpow(P[i], j);
            out->W[i][j] = (double) wc.re;
            out->W[4 + i][j] = (double) wc.im;
            out->W[8 + i][j] = (double) wa.re;
            out->W[12 + i][j] = (double) wa.im;
        }
        const int pw[5] = {1, 2, 4, 8, 16};
        for (int k = 0; k < 5; ++k) put(out->mu_pow[k][i], cpow(mu[i], G * pw[k]));      // column-to-column powers
        for (int k = 0; k <= SDRT_TPOW; ++k) put(out->tpow[k][i], cpow(mu[i], SDRT_TB * G * k));
        for (int nb = 0; nb < SDRT_TB * G; ++nb) {
            put(out->cF[nb][i], cpow(mu[i], nb));
            put(out->cG[nb][i], cpow(mu[i], SDRT_TB * G - nb));
        }
    }

    // Left edge: phi_i just before the first real sample, for a real stream u[0..27] (literal evaluation:
    // e[t] = 2u[0] - u[27-t], t < 27; phi(0) = e[0]/(1-p); phi(t+1) = p phi(t) + e[t]).  Scaled by alpha_i.
    const int E = SDRT_EDGE;
    for (int l = 0; l <= E; ++l) {
        R u[E + 1];
        for (int k = 0; k <= E; ++k) u[k] = (k == l) ? 1 : 0;
        for (int i = 0; i < 4; ++i) {
            C phi = C(2 * u[0] - u[E]) / (C(1) - P[i]);
            for (int t = 0; t < E; ++t) phi = P[i] * phi + C(2 * u[0] - u[E - t]);
            put(out->head[i][l], ac[i] * phi);
        }
    }

    // Right edge: anticausal state (after consuming the first extension sample) as a real-linear function of the
    // causal states F_j at the end of the data (scaled coordinates) and of the last 28 samples of the stream.
    // Literal evaluation over all 8 poles of: forward through the 27 extension samples, y_f[N-1], the backward
    // initial state zi*y_f[N-1] in modal form, the decoupling term, then back through the extension.
    auto tail_eval = [&](const C Fs[4], const R ut[E + 1], C G[4]) {
        C phi[8], prev[8];
        for (int i = 0; i < 4; ++i) {
            C ph = Fs[i] / ac[i];
            phi[2 * i] = ph;
            phi[2 * i + 1] = conj(ph);
        }
        R et[E];
        for (int t = 0; t < E; ++t) et[t] = 2 * ut[E] - ut[E - 1 - t];   // 2u[n-1] - u[n-2-t]
        for (int t = 0; t < E; ++t) {
            for (int i = 0; i < 8; ++i) { prev[i] = phi[i]; phi[i] = m.p[i] * phi[i] + C(et[t]); }
        }
        C yf(m.D * et[E - 1]);
        for (int i = 0; i < 8; ++i) yf = yf + m.r[i] * prev[i];
        C Zp[8];
        for (int i = 0; i < 8; ++i) {
            C gam = yf / (C(1) - m.p[i]);
            for (int j = 0; j < 8; ++j) gam = gam - m.X[i][j] * phi[j];
            Zp[i] = m.r[i] * gam / m.alpha[i];
        }
        for (int t = E - 1; t >= 1; --t)
            for (int i = 0; i < 8; ++i) Zp[i] = m.p[i] * Zp[i] + C(et[t]);
        for (int i = 0; i < 4; ++i) G[i] = aa[i] * (m.p[2 * i] * Zp[2 * i] + C(et[0]));
    };
    {
        R zero_u[E + 1];
        for (int k = 0; k <= E; ++k) zero_u[k] = 0;
        for (int j = 0; j < 4; ++j) {
            C F1[4], Fi[4], G1[4], Gi[4];
            F1[j] = C(1);
            Fi[j] = C(0, 1);
            tail_eval(F1, zero_u, G1);
            tail_eval(Fi, zero_u, Gi);
            for (int i = 0; i < 4; ++i) {
                // G1 = a + b, Gi = i(a - b)  ->  a = (G1 - i Gi)/2, b = (G1 + i Gi)/2
                C iGi = C(0, 1) * Gi[i];
                put(out->tailF[i][j], (G1[i] - iGi) / C(2));
                put(out->tailFc[i][j], (G1[i] + iGi) / C(2));
            }
        }
        C F0[4];
        for (int l = 0; l <= E; ++l) {
            R u[E + 1];
            for (int k = 0; k <= E; ++k) u[k] = (k == l) ? 1 : 0;
            C G[4];
            tail_eval(F0, u, G);
            for (int i = 0; i < 4; ++i) put(out->tailU[i][l], G[i]);
        }
    }
    return 0;
}

namespace {

// Poles and residues of an SOS cascade in parallel form, y[t] = D x[t] + sum_k Re(r_k s_k(t)), s_k(t+1) = p_k s_k(t) + x[t]:
// one entry per conjugate pair (its residue doubled) or real pole, complex ones first.  False when the cascade has repeated
// poles, FIR sections or more modes than SDRT_OUT_MAXMODES.
struct OutModes {
    int nm = 0, ncomplex = 0;
    R D = 1;
    C p[16], r[16];
};

bool out_modes(const double *sos, int sections, OutModes &om) {
    if (sections < 1 || sections > 8) return false;
    C poles[16];
    int npoles = 0;
    struct Num { R c[3]; int deg; } nums[8];
    R D = 1;
    for (int s = 0; s < sections; ++s) {
        R a0 = sos[6 * s + 3];
        if (a0 == 0) return false;
        R b0 = (R) sos[6 * s] / a0, b1 = (R) sos[6 * s + 1] / a0, b2 = (R) sos[6 * s + 2] / a0;
        R a1 = (R) sos[6 * s + 4] / a0, a2 = (R) sos[6 * s + 5] / a0;
        D = D * b0;
        if (a2 != 0) {
            R disc = a1 * a1 - 4 * a2;
            if (disc == 0) return false;                   // repeated pole
            if (disc < 0) {
                R sq = rsqrt128(-disc);
                poles[npoles++] = C(-a1 / 2, sq / 2);
                poles[npoles++] = C(-a1 / 2, -sq / 2);
            } else {
                R sq = rsqrt128(disc);
                poles[npoles++] = C((-a1 + sq) / 2);
                poles[npoles++] = C((-a1 - sq) / 2);
            }
            nums[s].c[0] = b0; nums[s].c[1] = b1; nums[s].c[2] = b2; nums[s].deg = 2;   // b0 z^2 + b1 z + b2
        } else if (a1 != 0) {
            if (b2 != 0) return false;                     // improper: extra pole at z = 0
            poles[npoles++] = C(-a1);
            nums[s].c[0] = b0; nums[s].c[1] = b1; nums[s].c[2] = 0; nums[s].deg = 1;    // b0 z + b1
        } else {
            if (b1 != 0 || b2 != 0) return false;          // FIR section
            nums[s].c[0] = b0; nums[s].deg = 0;
        }
    }
    // distinctness
    for (int i = 0; i < npoles; ++i)
        for (int j = i + 1; j < npoles; ++j) {
            C d = poles[i] - poles[j];
            if ((double) (d.re * d.re + d.im * d.im) < 1e-24) return false;
        }
    int nm = 0;
    C rp[16], rr[16];
    for (int i = 0; i < npoles; ++i) {
        if (poles[i].im < 0) continue;                     // represented by its conjugate
        C num(1), den(1), p = poles[i];
        for (int s = 0; s < sections; ++s) {
            C v;
            if (nums[s].deg == 2) v = C(nums[s].c[0]) * p * p + C(nums[s].c[1]) * p + C(nums[s].c[2]);
            else if (nums[s].deg == 1) v = C(nums[s].c[0]) * p + C(nums[s].c[1]);
            else v = C(nums[s].c[0]);
            num = num * v;
        }
        for (int j = 0; j < npoles; ++j) if (j != i) den = den * (p - poles[j]);
        C r = num / den;
        if (poles[i].im > 0) r = r * C(2);
        rp[nm] = p;
        rr[nm] = r;
        ++nm;
    }
    if (nm > SDRT_OUT_MAXMODES) return false;
    // complex modes first, real ones after (the kernels handle the two kinds separately)
    int w = 0;
    om.ncomplex = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < nm; ++k)
            if ((rp[k].im != 0) == (pass == 0)) { om.p[w] = rp[k]; om.r[w] = rr[k]; ++w; if (pass == 0) om.ncomplex++; }
    om.nm = nm;
    om.D = D;
    return true;
}

}  // namespace

extern "C" int sdrt_build_out_tables(const double *sos, int sections, int m, OutTables *out) {
    std::memset(out, 0, sizeof(*out));
    out->nmodes = -1;
    out->E = m >= 64 ? m / 64 : 1;
    OutModes om;
    if (!out_modes(sos, sections, om)) return 0;
    out->ncomplex = om.ncomplex;
    out->D = (double) om.D;
    for (int k = 0; k < om.nm; ++k) {
        put(out->r[k], om.r[k]);
        put(out->p[k], om.p[k]);
        for (int i = 0; i < 6; ++i) put(out->pstep[k][i], cpow(om.p[k], out->E * (1 << i)));
        for (int j = 0; j < 16; ++j) put(out->prow[k][j], cpow(om.p[k], out->E * (j + 1)));
    }
    out->nmodes = om.nm;
    return 0;
}

extern "C" int sdrt_build_post8_tables(const double *out_sos, int sections, int N, int LPI, Post8Tables *out, double *fft_tab) {
    if (!(LPI == 4 || LPI == 8 || LPI == 16) || N != 2 * SDRT_P8_PPL * LPI) return -1;
    OutModes om;
    if (!out_modes(out_sos, sections, om) || om.nm != 2 || om.ncomplex != 1) return -2;
    std::memset(out, 0, sizeof(*out));
    out->N = N;
    out->LPI = LPI;
    const C pc = om.p[0], rc = om.r[0], prc = om.p[1], rrc = om.r[1];       // the real pole as a complex number with zero imaginary part
    for (int j = 0; j < 4; ++j) {
        const C gj = rc * cpow(pc, j), grj = rrc * cpow(prc, j);
        put(out->g[j], gj);
        out->gr[j] = (double) grj.re;
        // impulse response of the run from a zero state: h[0] = D, h[j] = Re(rc pc^(j-1)) + rr pr^(j-1)
        out->h[j] = j == 0 ? (double) om.D : (double) ((rc * cpow(pc, j - 1)).re + (rrc * cpow(prc, j - 1)).re);
    }
    for (int j = 0; j < 3; ++j) {
        put(out->pcp[j], cpow(pc, j + 1));
        out->prp[j] = (double) cpow(prc, j + 1).re;
    }
    put(out->pc4, cpow(pc, 4));
    out->pr4 = (double) cpow(prc, 4).re;
    put(out->pc8, cpow(pc, 8));
    out->pr8 = (double) cpow(prc, 8).re;
    put(out->pc32, cpow(pc, 32));
    out->pr32 = (double) cpow(prc, 32).re;
    for (int q = 0; q < 4; ++q) {
        put(out->pc_q[q], cpow(pc, 4 * (q + 1)));
        out->pr_q[q] = (double) cpow(prc, 4 * (q + 1)).re;
    }
    for (int l = 0; l < 16; ++l) {
        put(out->pc_lane[l], cpow(pc, 4 * l));
        out->pr_lane[l] = (double) cpow(prc, 4 * l).re;
    }
    put(out->pc_blk, cpow(pc, 4 * LPI));
    out->pr_blk = (double) cpow(prc, 4 * LPI).re;

    // transform tables (layouts: tables.h).  Angles are reduced exactly in integers before the trigonometric calls.
    const int P = SDRT_P8_PPL, H = N / 2, T = P * LPI, NC = P / LPI;
    const long double pi = 3.141592653589793238462643383279502884L;
    auto W = [&](long num, long den, double *dst, bool conj_) {       // exp(-2 pi i num/den) (or its conjugate)
        const long k = ((num % den) + den) % den;
        const long double ang = 2.0L * pi * (long double) k / (long double) den;
        dst[0] = (double) cosl(ang);
        dst[1] = (double) (conj_ ? sinl(ang) : -sinl(ang));
    };
    // the bin series lane l finishes (closed under k1 -> 16 - k1): 16 lanes: l; 8: l, 16 - l (lane 0: 0, 8); 4: those and 4 + l, 12 - l
    auto S = [&](int l, int ci) { return LPI == 16 ? l : (ci == 0 ? l : (ci == 1 ? (l ? 16 - l : 8) : (ci == 2 ? 4 + l : 12 - l))); };
    double *tw1 = fft_tab, *tw2 = fft_tab + 2 * T, *al = fft_tab + 4 * T, *be = fft_tab + 6 * T, *lp = fft_tab + 8 * T;
    // the lowpass constants that depend on the lane, [8][LPI] complex slots: scan multipliers inside the quad (zero where the quad
    // permutation hands a lane its own or an earlier lane's value), from the quad before, from the block's start
    for (int l = 0; l < LPI; ++l) {
        const int q = l & 3;
        auto slot = [&](int j) { return lp + 2 * (j * LPI + l); };
        slot(0)[0] = q >= 1 ? out->pc4[0] : 0.0; slot(0)[1] = q >= 1 ? out->pc4[1] : 0.0;
        slot(1)[0] = q >= 2 ? out->pc8[0] : 0.0; slot(1)[1] = q >= 2 ? out->pc8[1] : 0.0;
        slot(2)[0] = q >= 1 ? out->pr4 : 0.0;    slot(2)[1] = q >= 2 ? out->pr8 : 0.0;
        slot(3)[0] = out->pc_q[q][0];            slot(3)[1] = out->pc_q[q][1];
        slot(4)[0] = out->pc_lane[q][0];         slot(4)[1] = out->pc_lane[q][1];
        slot(5)[0] = out->pr_q[q];               slot(5)[1] = out->pr_lane[q];
        slot(6)[0] = out->pc_lane[l][0];         slot(6)[1] = out->pc_lane[l][1];
        slot(7)[0] = out->pr_lane[l];            slot(7)[1] = q >= 1 ? 1.0 : 0.0;
    }
    for (int k1 = 0; k1 < P; ++k1)
        for (int l = 0; l < LPI; ++l) W((long) l * k1, H, tw1 + 2 * (k1 * LPI + l), false);
    // sh[k] = exp(i pi k / N) / N for k < H, 0 at k = H (scipy drops what is left of the Nyquist bin); theta = 2 pi k / N:
    //   alpha_k = sh[k] (1 - sin theta) + conj(sh[H - k]) (1 + sin theta),  beta_k = i cos theta (sh[k] - conj(sh[H - k]))
    auto sh = [&](int k, long double &re, long double &im) {
        if (k >= H) { re = im = 0.0L; return; }
        const long double ang = pi * (long double) k / (long double) N;
        re = cosl(ang) / (long double) N;
        im = sinl(ang) / (long double) N;
    };
    for (int ci = 0; ci < NC; ++ci)
        for (int j = 0; j < LPI; ++j)
            for (int l = 0; l < LPI; ++l) {
                const int c = S(l, ci), idx = (ci * LPI + j) * LPI + l;
                W((long) j * c, H, tw2 + 2 * idx, true);
                const int k = c + P * j;
                long double ar, ai, br, bi;
                sh(k, ar, ai);
                sh(H - k, br, bi);
                bi = -bi;                                   // conj(sh[H - k])
                const long double th = 2.0L * pi * (long double) k / (long double) N, sn = sinl(th), cs = cosl(th);
                al[2 * idx] = (double) (ar * (1 - sn) + br * (1 + sn));
                al[2 * idx + 1] = (double) (ai * (1 - sn) + bi * (1 + sn));
                // i cos theta (dr + i di) = cos theta (-di + i dr)
                be[2 * idx] = (double) (-cs * (ai - bi));
                be[2 * idx + 1] = (double) (cs * (ar - br));
            }
    return 0;
}

extern "C" void sdrt_build_nco(double f, double fs, int tile_len, int n_tiles, double *tile, double *tilephase) {
    // tile[t], t < tile_len: exp(-2 pi i (f/fs) t);  tilephase[T], T < n_tiles: exp(-2 pi i (f/fs) tile_len T)
    const long double two_pi = 6.283185307179586476925286766559005768L;
    auto phasor = [&](long double t, double *dst) {
        // exp(-2 pi i (f/fs) t): reduce f*t modulo fs before scaling so the angle stays in (-2pi, 2pi)
        long double cyc = fmodl((long double) f * t, (long double) fs) / (long double) fs;
        long double ang = -two_pi * cyc;
        dst[0] = (double) cosl(ang);
        dst[1] = (double) sinl(ang);
    };
    for (int t = 0; t < tile_len; ++t) phasor((long double) t, tile + 2 * t);
    for (int T = 0; T < n_tiles; ++T) phasor((long double) T * (long double) tile_len, tilephase + 2 * T);
}

extern "C" void sdrt_build_resample_odd(int N, double *h) {
    // scipy.signal.resample(a, 2N), real a (demodulation.py:35-38): X = rfft(a); Y[:N/2+1] = X; if N even the bin
    // N/2 is halved; y = irfft(Y, 2N) * 2.  For a unit impulse at 0 the odd outputs are
    //   h[d] = y[2d+1] = (1/N) [ 1 + 2 sum_{f=1}^{F} cos(pi f (2d+1)/N) ],  F = N/2-1 (N even; the halved bin
    //   contributes cos(pi(2d+1)/2) = 0) or (N-1)/2 (N odd).
    const long double pi = 3.141592653589793238462643383279502884L;
    const int F = (N % 2 == 0) ? N / 2 - 1 : (N - 1) / 2;
    for (int d = 0; d < N; ++d) {
        long double s = 1.0L;
        for (int f = 1; f <= F; ++f) {
            // reduce f*(2d+1) modulo 2N exactly in integers before calling cosl
            long long k = ((long long) f * (2LL * d + 1)) % (2LL * N);
            s += 2.0L * cosl(pi * (long double) k / (long double) N);
        }
        h[d] = (double) (s / (long double) N);
    }
}

extern "C" void sdrt_build_fft_resample(int N, double *tw, double *shift) {
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int k = 0; k < N; ++k) {
        // exp(-2 pi i k/N) with the angle reduced to the first octant by symmetry would be tighter still; cosl/sinl of an
        // argument < 2 pi in long double is good to ~1e-19
        const long double ang = -2.0L * pi * (long double) k / (long double) N;
        tw[2 * k] = (double) cosl(ang);
        tw[2 * k + 1] = (double) sinl(ang);
        const int fs = k < N / 2 ? k : k - N;
        if (2 * k == N) { shift[2 * k] = 0.0; shift[2 * k + 1] = 0.0; continue; }
        const long double a2 = pi * (long double) fs / (long double) N;
        shift[2 * k] = (double) (cosl(a2) / (long double) N);
        shift[2 * k + 1] = (double) (sinl(a2) / (long double) N);
    }
}

// Packed form of the same resampler (post_fftp_kernel): the N real pair angles a[n] go two to a complex number, z[n] = a[2n] +
// i a[2n+1], through ONE transform of H = N/2 points each way.  With Z = DFT_H(z) the packed spectrum of the half-sample shifted
// sequence o (the odd output samples, demodulation.py:35-38) is  Wp[k] = alpha_k Z[k] + beta_k conj(Z[(H - k) mod H])  and
// IDFT_H(Wp) (unnormalised) = o[2n] + i o[2n+1].  tw: [H] exp(-2 pi i k / H); ab: [H][4] = alpha (re, im), beta (re, im).
extern "C" void sdrt_build_fft_resample_packed(int N, double *tw, double *ab) {
    const long double pi = 3.141592653589793238462643383279502884L;
    const int H = N / 2;
    typedef std::complex<long double> CL;
    auto shift = [&](int k) -> CL {            // as sdrt_build_fft_resample: exp(i pi f / N) / N, zero at the Nyquist bin
        if (2 * k == N) return CL(0, 0);
        const int fs = k < N / 2 ? k : k - N;
        const long double a2 = pi * (long double) fs / (long double) N;
        return CL(cosl(a2), sinl(a2)) / (long double) N;
    };
    for (int k = 0; k < H; ++k) {
        const long double ang = -2.0L * pi * (long double) k / (long double) H;
        tw[2 * k] = (double) cosl(ang);
        tw[2 * k + 1] = (double) sinl(ang);
        const long double an = -2.0L * pi * (long double) k / (long double) N;
        const CL w(cosl(an), sinl(an)), iw = CL(0, 1) / w, half(0.5L, 0), wi = w / CL(0, 2);
        const CL s0 = shift(k), s1 = shift(k + H);
        const CL cz0 = s0 * (half + wi), cz1 = s1 * (half - wi), cc0 = s0 * (half - wi), cc1 = s1 * (half + wi);
        const CL al = ((cz0 + cz1) + iw * (cz0 - cz1)), be = ((cc0 + cc1) + iw * (cc0 - cc1));      // (the factor 2 of the packing folded in)
        ab[4 * k] = (double) al.real();
        ab[4 * k + 1] = (double) al.imag();
        ab[4 * k + 2] = (double) be.real();
        ab[4 * k + 3] = (double) be.imag();
    }
}

namespace {

// quantise rows x cols reals (row-major, leading dimension ld) to S balanced base-256 digits against one power-of-two
// scale; emit(d, r, c, digit) receives every digit.  Returns the combine factor scale * 2^-(8S-1), or 0 on overflow.
// consistent (round 4; the --correct-iq forms of the wide kernel, columns alternating re / im coefficients): rowsum[2 r + comp]
// receives the integer sum of the QUANTISED weights of row r and component comp -- made a multiple of 16 by moving up to 8 weights
// one unit each (so that 32896 x sum, the unsigned 16-bit records' constant, is still an exact double; the sums of the unmoved
// weights are already off by ~5 units) -- and dhalf[r] HALF the sum of the row's lowest digits.  Why: those forms carry the
// constant c the byte slices leave out in a rank-one term, value += (c - offset) g[row], and the 16-bit forms never compute the
// product lowest digit x low slice (kernels_i8w.hip: LMIN).  For an all-zero chunk (low slices = -128 everywhere) the matrix cores
// then deliver exactly -128 x (sum of the weights less their lowest digits); with g = the sum of the QUANTISED weights and the
// accumulators of the level above opened at -dhalf (units of 256: -128 x the lowest digits' sum) the two cancel to the last bit --
// the reference's exact zeros for a stream that starts silent (offset 0), instead of the FM angle of a 1e-10 residue.  For other
// input the opening value is a constant per row of 2^-41 of full scale, below the products that are left out.
template <class Emit>
double quantise_planes(const double *W, int rows, int cols, int ld, int S, Emit emit, bool consistent = false,
                       long long *rowsum = nullptr, int *dhalf = nullptr) {
    double maxabs = 0.0;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) maxabs = std::fmax(maxabs, std::fabs(W[(size_t) r * ld + c]));
    if (!(maxabs > 0.0)) return 0.0;
    int e = 0;
    std::frexp(maxabs * 1.002, &e);
    const double scale = std::ldexp(1.0, e);
    const int B = 8 * S - 1;
    std::vector<long long> I((size_t) cols);
    auto low = [](long long v) { return ((v + 128) % 256 + 256) % 256 - 128; };      // balanced lowest digit in [-128, 127]
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c) I[c] = std::llround(std::ldexp(W[(size_t) r * ld + c] / scale, B));
        if (consistent) {
            // every component's sum becomes a multiple of 16: up to 8 weights move one unit each, those that sit farthest from
            // their unrounded values, towards them
            for (int comp = 0; comp < 2; ++comp) {
                long long sum = 0;
                for (int c = comp; c < cols; c += 2) sum += I[c];
                int adj = (int) (((sum % 16) + 16) % 16);        // sum = 16 k + adj
                adj = adj <= 8 ? -adj : 16 - adj;
                const int dir = adj > 0 ? 1 : -1;
                for (; adj != 0; adj -= dir) {
                    int best = -1;
                    double gain = -2.0;
                    for (int c = comp; c < cols; c += 2) {
                        const double t = (std::ldexp(W[(size_t) r * ld + c] / scale, B) - (double) I[c]) * dir;   // > 0: the move reduces the error
                        const long long nd = low(I[c]) + dir;
                        if (nd < -128 || nd > 127 || std::fabs(t) > 0.5000001) continue;       // in range, and not moved already
                        if (t > gain) { gain = t; best = c; }
                    }
                    if (best < 0) return 0.0;
                    I[best] += dir;
                }
            }
            long long d0 = 0;
            for (int c = 0; c < cols; ++c) d0 += low(I[c]);
            if (d0 & 1) return 0.0;                               // (the sum of the weights is even, and so is that of their lowest digits)
            if (dhalf) dhalf[r] = (int) (-(d0 / 2));          // what the lowest accumulators open at
            if (rowsum) {
                rowsum[2 * r] = rowsum[2 * r + 1] = 0;
                for (int c = 0; c < cols; ++c) rowsum[2 * r + (c & 1)] += I[c];
            }
        }
        for (int c = 0; c < cols; ++c) {
            long long v = I[c];
            for (int d = 0; d < S; ++d) {
                const long long dig = low(v);
                v = (v - dig) / 256;
                emit(d, r, c, (signed char) dig);
            }
            if (v != 0) return 0.0;
        }
    }
    return std::ldexp(scale, -B);
}

}  // namespace

extern "C" int sdrt_build_i8_weights16(const double *sos, int sections, const DecimTables *tab, double data_scale,
                                       const double *nco_samp, int S, signed char *out, double *combine,
                                       signed char *mid_out, double *mid_combine) {
    Modal m;
    const char *err = nullptr;
    if (!build_modal(sos, sections, m, &err)) return -2;
    const int L = tab->L, G = tab->G, LC = L * G, cols = 2 * LC, KS = LC / 32;
    if (LC % 32) return -3;
    auto tr = [&](int t) { return nco_samp ? nco_samp[2 * t] : 1.0; };
    auto ti = [&](int t) { return nco_samp ? nco_samp[2 * t + 1] : 0.0; };
    auto kpos = [&](int c, int &ks, int &kg, int &j) {
        const int sample = c >> 1, comp = c & 1;
        kg = sample / (LC / 4);
        const int within = sample % (LC / 4);
        ks = within / 8;
        j = 2 * (within % 8) + comp;
    };
    std::vector<double> Wf((size_t) 32 * cols, 0.0);          // rows 0..15 causal tile, 16..31 anticausal tile
    C P[4], ac[4], aa[4];
    for (int i = 0; i < 4; ++i) { P[i] = m.p[2 * i]; ac[i] = m.alpha[2 * i]; aa[i] = m.alpha[2 * i] / P[i]; }
    for (int mode = 0; mode < 4; ++mode)
        for (int kind = 0; kind < 2; ++kind)
            for (int t = 0; t < LC; ++t) {
                const C w = kind == 0 ? ac[mode] * cpow(P[mode], LC - 1 - t) : aa[mode] * cpow(P[mode], t);
                const double wr = (double) w.re * data_scale, wi = (double) w.im * data_scale;
                const double cxr[2][2] = {{wr * tr(t), wi * tr(t)}, {wr * ti(t), wi * ti(t)}};      // [stream][part], coefficient of xr
                const double cxi[2][2] = {{-wr * ti(t), -wi * ti(t)}, {wr * tr(t), wi * tr(t)}};    // coefficient of xi
                for (int stream = 0; stream < 2; ++stream)
                    for (int part = 0; part < 2; ++part) {
                        const int r = 16 * kind + 4 * mode + 2 * stream + part;
                        Wf[(size_t) r * cols + 2 * t] = cxr[stream][part];
                        Wf[(size_t) r * cols + 2 * t + 1] = cxi[stream][part];
                    }
            }
    *combine = quantise_planes(Wf.data(), 32, cols, cols, S, [&](int d, int r, int c, signed char dig) {
        int ks, kg, j;
        kpos(c, ks, kg, j);
        out[(((((size_t) (r >> 4) * S + d) * KS + ks) * 64) + ((r & 15) + 16 * kg)) * 16 + j] = dig;
    });
    if (*combine == 0.0) return -1;
    if (G > 1) {
        const int NR = 2 * (G - 1);
        C sum_a_over_p(0);
        for (int i = 0; i < 8; ++i) sum_a_over_p = sum_a_over_p + m.alpha[i] / m.p[i];
        const double d0p = (double) (m.d0 - sum_a_over_p.re) * data_scale;
        std::vector<double> Wm((size_t) NR * cols, 0.0);
        for (int g = 1; g < G; ++g)
            for (int t = 0; t < LC; ++t) {
                R acc = 0;
                for (int i = 0; i < 4; ++i) {
                    const C w = t < L * g ? ac[i] * cpow(P[i], L * g - 1 - t) : aa[i] * cpow(P[i], t - L * g);
                    acc = acc + 2 * w.re;
                }
                double mt = (double) acc * data_scale;
                if (t == L * g) mt += d0p;
                Wm[(size_t) (0 + 2 * (g - 1)) * cols + 2 * t] = mt * tr(t);        // stream re: mt (xr tr - xi ti)
                Wm[(size_t) (0 + 2 * (g - 1)) * cols + 2 * t + 1] = -mt * ti(t);
                Wm[(size_t) (1 + 2 * (g - 1)) * cols + 2 * t] = mt * ti(t);        // stream im: mt (xr ti + xi tr)
                Wm[(size_t) (1 + 2 * (g - 1)) * cols + 2 * t + 1] = mt * tr(t);
            }
        *mid_combine = quantise_planes(Wm.data(), NR, cols, cols, S, [&](int d, int r, int c, signed char dig) {
            int ks, kg, j;
            kpos(c, ks, kg, j);
            mid_out[(((((size_t) d * KS + ks) * 4 + kg) * NR) + r) * 16 + j] = dig;
        });
        if (*mid_combine == 0.0) return -1;
    }
    return 0;
}

extern "C" void sdrt_build_iq_pow(double lambda, int count, double *pw) {
    R a = (R) 1 - (R) lambda;
    R v = 1;
    for (int j = 0; j <= count; ++j) {
        pw[j] = (double) v;
        v = v * a;
    }
}

// ---- wide-column kernel (kernels_i8w.hip) --------------------------------------------------------------------------
extern "C" int sdrt_build_wide_tables(const double *sos, int sections, int q, double data_scale, WideTables *out,
                                      const char **err) {
    if (q < SDRT_WIDE_LC / SDRT_WIDE_MAXOUT) { if (err) *err = "wide columns need 32 <= q <= 256"; return -1; }
    return sdrt_build_wide_tables_pass(sos, sections, q, data_scale, SDRT_WIDE_LC / (q > 0 ? q : 1), 0, out, err);
}

extern "C" int sdrt_build_wide_tables_pass(const double *sos, int sections, int q, double data_scale, int nout, int boff,
                                           WideTables *out, const char **err) {
    const int LC = SDRT_WIDE_LC;
    if (q < 2 || q > LC || (q & (q - 1))) { if (err) *err = "wide columns need a power-of-two decimation <= 256"; return -1; }
    if (nout < 1 || nout > SDRT_WIDE_MAXOUT || boff < 0 || (boff + nout) * q > LC) { if (err) *err = "bad output range of the pass"; return -1; }
    Modal m;
    if (!build_modal(sos, sections, m, err)) return -2;
    DecimTables base;
    if (int rc = sdrt_build_decim_tables(sos, sections, q, 1, &base, err)) return rc;
    std::memset(out, 0, sizeof(*out));
    out->q = q;
    out->nout = nout;
    out->boff = boff;
    out->nout_total = LC / q;
    for (int i = 0; i < 4; ++i) {
        const C P = m.p[2 * i];
        const C mu = cpow(P, LC);
        for (int k = 0; k < 4; ++k) put(out->mu_pow[k][i], cpow(mu, 1 << k));
        for (int k = 0; k <= SDRT_WIDE_MAXTILES; ++k) put(out->tile_pow[k][i], cpow(mu, SDRT_TB * k));
        for (int b = 0; b <= nout; ++b) {
            put(out->bp2[b][i], C(2) * cpow(P, (boff + b) * q));
            put(out->bp2[nout + 1 + b][i], C(2) * cpow(P, LC - (boff + b) * q));
        }
        for (int pos = 0; pos < SDRT_TB * nout; ++pos) {
            const int s = (pos / nout) * LC + (boff + pos % nout) * q;      // the output's sample position in the tile
            put(out->cF2[pos][i], C(2) * cpow(P, s));
            put(out->cG2[pos][i], C(2) * cpow(P, SDRT_TB * LC - s));
        }
    }
    for (int i = 0; i < SDRT_NMODE; ++i) {
        for (int l = 0; l <= SDRT_EDGE; ++l)
            for (int z = 0; z < 2; ++z) {
                out->head[i][l][z] = base.head[i][l][z] * data_scale;
                out->tailU[i][l][z] = base.tailU[i][l][z] * data_scale;
            }
        for (int j = 0; j < SDRT_NMODE; ++j)
            for (int z = 0; z < 2; ++z) { out->tailF[i][j][z] = base.tailF[i][j][z]; out->tailFc[i][j][z] = base.tailFc[i][j][z]; }
    }
    return 0;
}

namespace {
// row (coefficients of x_re at even, of x_im at odd positions, LC samples) -> row of W (I - T); g-sums of the ORIGINAL row
void fold_iq(double *row, int LC, double lambda, double *g2) {
    const R lam = (R) lambda, om = (R) 1 - lam;
    for (int comp = 0; comp < 2; ++comp) {
        R gsum = 0, gp = 1;
        for (int t = 0; t < LC; ++t) { gsum = gsum + (R) row[2 * t + comp] * gp; gp = gp * om; }
        g2[comp] = (double) gsum;
        R Sacc = 0;                                   // S_j = sum_{i>j} c[i] lambda (1-lambda)^(i-1-j)
        for (int j = LC - 1; j >= 0; --j) {
            const R c = (R) row[2 * j + comp];
            row[2 * j + comp] = (double) (c - Sacc);
            Sacc = lam * c + om * Sacc;
        }
    }
}
}  // namespace

extern "C" int sdrt_build_i8_weights_wide(const double *sos, int sections, int q, double data_scale, const double *nco_samp,
                                          int S, double iq_lambda, signed char *main_out, double *main_combine,
                                          signed char *loc_out, double *loc_combine, double *gvec) {
    if (q < SDRT_WIDE_LC / SDRT_WIDE_MAXOUT) return -3;
    return sdrt_build_i8_weights_wide_pass(sos, sections, q, data_scale, nco_samp, S, iq_lambda, SDRT_WIDE_LC / (q > 0 ? q : 1), 0, main_out,
                                           main_combine, loc_out, loc_combine, gvec);
}

extern "C" int sdrt_build_i8_weights_wide_pass(const double *sos, int sections, int q, double data_scale, const double *nco_samp,
                                               int S, double iq_lambda, int nout, int boff, signed char *main_out,
                                               double *main_combine, signed char *loc_out, double *loc_combine, double *gvec) {
    Modal m;
    const char *err = nullptr;
    if (!build_modal(sos, sections, m, &err)) return -2;
    const int LC = SDRT_WIDE_LC, cols = 2 * LC, KS = LC / 32;
    if (q < 2 || q > LC || (q & (q - 1)) || nout < 1 || nout > SDRT_WIDE_MAXOUT || boff < 0 || (boff + nout) * q > LC) return -3;
    const int NR = 2 * nout;
    auto tr = [&](int t) { return nco_samp ? nco_samp[2 * t] : 1.0; };
    auto ti = [&](int t) { return nco_samp ? nco_samp[2 * t + 1] : 0.0; };
    auto kpos = [&](int c, int &ks, int &kg, int &j) {
        const int sample = c >> 1, comp = c & 1;
        ks = sample / 32;
        kg = (sample % 32) / 8;
        j = 2 * (sample % 8) + comp;
    };
    C P[4], ac[4], aa[4];
    for (int i = 0; i < 4; ++i) { P[i] = m.p[2 * i]; ac[i] = m.alpha[2 * i]; aa[i] = m.alpha[2 * i] / P[i]; }
    // column aggregates: causal alpha p^(LC-1-t) (state handed to the next column), anticausal (alpha/p) p^t (state at
    // the column start); u = x e^{i theta t} split into its real streams
    std::vector<double> Wf((size_t) 32 * cols, 0.0);
    for (int mode = 0; mode < 4; ++mode)
        for (int kind = 0; kind < 2; ++kind)
            for (int t = 0; t < LC; ++t) {
                const C w = kind == 0 ? ac[mode] * cpow(P[mode], LC - 1 - t) : aa[mode] * cpow(P[mode], t);
                const double wr = (double) w.re * data_scale, wi = (double) w.im * data_scale;
                const double cxr[2][2] = {{wr * tr(t), wi * tr(t)}, {wr * ti(t), wi * ti(t)}};      // [stream][part], coefficient of xr
                const double cxi[2][2] = {{-wr * ti(t), -wi * ti(t)}, {wr * tr(t), wi * tr(t)}};    // coefficient of xi
                for (int stream = 0; stream < 2; ++stream)
                    for (int part = 0; part < 2; ++part) {
                        const int r = 16 * kind + 4 * mode + 2 * stream + part;
                        Wf[(size_t) r * cols + 2 * t] = cxr[stream][part];
                        Wf[(size_t) r * cols + 2 * t + 1] = cxi[stream][part];
                    }
            }
    if (gvec)                                  // lambda == 0: the rows stay, gvec = the rows' sums (float32: the slices' constant)
        for (int r = 0; r < 32; ++r) fold_iq(&Wf[(size_t) r * cols], LC, iq_lambda, gvec + 2 * r);
    // --correct-iq: g = the sum of the QUANTISED weights, and half the sum of every row's lowest digits behind the vectors
    // (quantise_planes: a stream that starts silent then cancels to exact zeros)
    const bool consistent = gvec != nullptr && iq_lambda > 0.0;
    std::vector<long long> rs((size_t) 32 * 2);
    int *open = gvec ? (int *) (gvec + 2 * 48) : nullptr;          // [48] after the 48 vector pairs (SDRT_WIDE_GV)
    if (open) memset(open, 0, 48 * sizeof(int));
    *main_combine = quantise_planes(Wf.data(), 32, cols, cols, S, [&](int d, int r, int c, signed char dig) {
        int ks, kg, j;
        kpos(c, ks, kg, j);
        main_out[(((((size_t) (r >> 4) * S + d) * KS + ks) * 64) + ((r & 15) + 16 * kg)) * 16 + j] = dig;
    }, consistent, consistent ? rs.data() : nullptr, consistent ? open : nullptr);
    if (*main_combine == 0.0) return -1;
    if (consistent)
        for (int r = 0; r < 32; ++r) { gvec[2 * r] = *main_combine * (double) rs[2 * r]; gvec[2 * r + 1] = *main_combine * (double) rs[2 * r + 1]; }
    // local rows: output at sample b q of the column, the part that depends on this column's samples:
    //   d0' u[bq] + sum_i 2 Re( alpha_i sum_{t < bq} p_i^(bq-1-t) u[t] + (alpha_i/p_i) sum_{t >= bq} p_i^(t-bq) u[t] )
    C sum_a_over_p(0);
    for (int i = 0; i < 8; ++i) sum_a_over_p = sum_a_over_p + m.alpha[i] / m.p[i];
    const double d0p = (double) (m.d0 - sum_a_over_p.re) * data_scale;
    std::vector<double> Wl((size_t) NR * cols, 0.0);
    for (int b = 0; b < nout; ++b)
        for (int t = 0; t < LC; ++t) {
            const int so = (boff + b) * q;                   // the output's sample inside the column
            R acc = 0;
            for (int i = 0; i < 4; ++i) {
                const C w = t < so ? ac[i] * cpow(P[i], so - 1 - t) : aa[i] * cpow(P[i], t - so);
                acc = acc + 2 * w.re;
            }
            double mt = (double) acc * data_scale;
            if (t == so) mt += d0p;
            Wl[(size_t) (2 * b) * cols + 2 * t] = mt * tr(t);            // stream re: mt (xr tr - xi ti)
            Wl[(size_t) (2 * b) * cols + 2 * t + 1] = -mt * ti(t);
            Wl[(size_t) (2 * b + 1) * cols + 2 * t] = mt * ti(t);        // stream im: mt (xr ti + xi tr)
            Wl[(size_t) (2 * b + 1) * cols + 2 * t + 1] = mt * tr(t);
        }
    if (gvec) {
        for (int r = 0; r < 16; ++r) gvec[2 * (32 + r)] = gvec[2 * (32 + r) + 1] = 0.0;
        for (int r = 0; r < NR; ++r) fold_iq(&Wl[(size_t) r * cols], LC, iq_lambda, gvec + 2 * (32 + r));
    }
    std::vector<long long> rl((size_t) NR * 2);
    *loc_combine = quantise_planes(Wl.data(), NR, cols, cols, S, [&](int d, int r, int c, signed char dig) {
        int ks, kg, j;
        kpos(c, ks, kg, j);
        loc_out[(((((size_t) d * KS + ks) * 4 + kg) * NR) + r) * 16 + j] = dig;
    }, consistent, consistent ? rl.data() : nullptr, consistent ? open + 32 : nullptr);
    if (*loc_combine == 0.0) return -1;
    if (consistent)
        for (int r = 0; r < NR; ++r) { gvec[2 * (32 + r)] = *loc_combine * (double) rl[2 * r]; gvec[2 * (32 + r) + 1] = *loc_combine * (double) rl[2 * r + 1]; }
    return 0;
}

extern "C" int sdrt_build_i8_offset_rows(double lambda, int S, signed char *out, double *combine, double *consts) {
    // the column's contribution to the IQ offset (read_file.py:64-77: off' = (1-lambda) off + lambda x):
    //   off(column end) = (1-lambda)^LC off(column start) + sum_j w_j x_j,   w_j = lambda (1-lambda)^(LC-1-j)
    const int LC = SDRT_WIDE_LC, cols = 2 * LC, KS = LC / 32;
    if (!(lambda > 0.0) || !(lambda < 1.0) || S > 8) return -3;
    const R lam = (R) lambda, om = (R) 1 - lam;
    std::vector<R> pw((size_t) LC + 1);
    pw[0] = 1;
    for (int t = 1; t <= LC; ++t) pw[t] = pw[t - 1] * om;
    std::vector<double> W((size_t) 2 * cols, 0.0);
    R wsum = 0;
    for (int j = 0; j < LC; ++j) {
        const R w = lam * pw[LC - 1 - j];
        wsum = wsum + w;
        W[(size_t) 0 * cols + 2 * j] = (double) w;          // row "re": coefficient of x_re
        W[(size_t) 1 * cols + 2 * j + 1] = (double) w;      // row "im": coefficient of x_im
    }
    memset(out, 0, (size_t) KS * 64 * 16);
    long long rs[4] = {0, 0, 0, 0};
    *combine = quantise_planes(W.data(), 2, cols, cols, S, [&](int d, int r, int c, signed char dig) {
        const int sample = c >> 1, comp = c & 1, ks = sample / 32, kg = (sample % 32) / 8, j = 2 * (sample % 8) + comp;
        out[(((size_t) ks * 64) + ((2 * d + r) + 16 * kg)) * 16 + j] = dig;
    }, true, rs);
    if (*combine == 0.0) return -1;
    // the sum of the QUANTISED weights: what the constant c of the slices adds; for an all-zero column the matrix cores deliver
    // exactly -c x this (every product is computed here), and the offset stays 0.0
    (void) wsum;
    consts[0] = *combine * (double) rs[0];
    R p = pw[LC];
    for (int k = 0; k < 5; ++k) { consts[1 + k] = (double) p; p = p * p; }      // (1-lambda)^(LC 2^k), k = 0..4
    R p32 = 1;
    for (int t = 0; t < 32; ++t) p32 = p32 * om;
    consts[6] = (double) ((R) 1 / p32);                                          // (1-lambda)^-32
    for (int l = 0; l < 32; ++l) consts[7 + l] = (double) (lam * pw[31 - l]);     // tail window weights lambda (1-lambda)^(31-l)
    for (int nb = 0; nb < SDRT_TB; ++nb) consts[39 + nb] = (double) ([&] { R q = 1; for (int t = 0; t < nb; ++t) q = q * pw[LC]; return q; })();   // (1-lambda)^(LC nb)
    return 0;
}

extern "C" int sdrt_build_i8_weights_wide_rs(const double *sos, int sections, int q, double data_scale, int S,
                                             signed char *main_out, double *main_combine, signed char *loc_out,
                                             double *loc_combine) {
    Modal m;
    const char *err = nullptr;
    if (!build_modal(sos, sections, m, &err)) return -2;
    const int LC = SDRT_WIDE_LC, KS = LC / 64;
    if (q < LC / SDRT_WIDE_MAXOUT || q > LC || (q & (q - 1))) return -3;
    const int nout = LC / q;
    C P[4], ac[4], aa[4];
    for (int i = 0; i < 4; ++i) { P[i] = m.p[2 * i]; ac[i] = m.alpha[2 * i]; aa[i] = m.alpha[2 * i] / P[i]; }
    std::vector<double> Wf((size_t) 16 * LC, 0.0);
    for (int mode = 0; mode < 4; ++mode)
        for (int kind = 0; kind < 2; ++kind)
            for (int t = 0; t < LC; ++t) {
                const C w = kind == 0 ? ac[mode] * cpow(P[mode], LC - 1 - t) : aa[mode] * cpow(P[mode], t);
                Wf[(size_t) (4 * mode + 2 * kind + 0) * LC + t] = (double) w.re * data_scale;
                Wf[(size_t) (4 * mode + 2 * kind + 1) * LC + t] = (double) w.im * data_scale;
            }
    *main_combine = quantise_planes(Wf.data(), 16, LC, LC, S, [&](int d, int r, int c, signed char dig) {
        const int ks = c / 64, kg = (c % 64) / 16, j = c % 16;
        main_out[((((size_t) d * KS + ks) * 64) + (r + 16 * kg)) * 16 + j] = dig;
    });
    if (*main_combine == 0.0) return -1;
    C sum_a_over_p(0);
    for (int i = 0; i < 8; ++i) sum_a_over_p = sum_a_over_p + m.alpha[i] / m.p[i];
    const double d0p = (double) (m.d0 - sum_a_over_p.re) * data_scale;
    std::vector<double> Wl((size_t) nout * LC, 0.0);
    for (int b = 0; b < nout; ++b)
        for (int t = 0; t < LC; ++t) {
            R acc = 0;
            for (int i = 0; i < 4; ++i) {
                const C w = t < b * q ? ac[i] * cpow(P[i], b * q - 1 - t) : aa[i] * cpow(P[i], t - b * q);
                acc = acc + 2 * w.re;
            }
            Wl[(size_t) b * LC + t] = (double) acc * data_scale + (t == b * q ? d0p : 0.0);
        }
    *loc_combine = quantise_planes(Wl.data(), nout, LC, LC, S, [&](int d, int r, int c, signed char dig) {
        const int ks = c / 64, kg = (c % 64) / 16, j = c % 16;
        loc_out[(((((size_t) d * KS + ks) * 4 + kg) * nout) + r) * 16 + j] = dig;
    });
    if (*loc_combine == 0.0) return -1;
    return 0;
}

extern "C" void sdrt_build_wide_edge(const WideTables *wt, const double *nco_samp, const double *nco_blk, int n, double iq_lambda,
                                     double *ab, double *pq) {
    const int LC = SDRT_WIDE_LC, WN = 32;
    const R lam = (R) iq_lambda, om = (R) 1 - lam;
    R g[WN];                                           // (1-lambda)^l
    g[0] = 1;
    for (int l = 1; l < WN; ++l) g[l] = g[l - 1] * om;
    for (int o16 = 0; o16 < 16; ++o16) {
        const int kind = o16 >> 3, o = o16 & 7, mode = o >> 1, st = o & 1;
        C hc1[WN], hc2[WN];                            // h_l c1_l, h_l c2_l: coefficient of z_re[l], z_im[l]
        for (int l = 0; l < WN; ++l) {
            C h(0);
            long idx;
            if (kind == 0) {
                idx = l;
                if (l <= SDRT_EDGE) h = C((R) wt->head[mode][l][0], (R) wt->head[mode][l][1]);
            } else {
                idx = (long) n - WN + l;
                if (l >= WN - 1 - SDRT_EDGE) h = C((R) wt->tailU[mode][l - (WN - 1 - SDRT_EDGE)][0], (R) wt->tailU[mode][l - (WN - 1 - SDRT_EDGE)][1]);
            }
            C rot(1);
            if (nco_samp) {
                const C a((R) nco_samp[2 * (idx % LC)], (R) nco_samp[2 * (idx % LC) + 1]);
                const C b((R) nco_blk[2 * (idx / LC)], (R) nco_blk[2 * (idx / LC) + 1]);
                rot = a * b;
            }
            // stream st of rot * (zr + i zi): st 0: rot.re zr - rot.im zi; st 1: rot.im zr + rot.re zi
            const R c1 = st ? rot.im : rot.re, c2 = st ? rot.re : -rot.im;
            hc1[l] = h * C(c1);
            hc2[l] = h * C(c2);
        }
        C P(0), Q(0);
        for (int j = 0; j < WN; ++j) {
            // column j of (I - T): 1 at l = j, -lambda (1-lambda)^(l-1-j) for l > j
            C A = hc1[j], B = hc2[j];
            if (iq_lambda > 0.0) {
                R t = lam;
                for (int l = j + 1; l < WN; ++l) {
                    A = A - hc1[l] * C(t);
                    B = B - hc2[l] * C(t);
                    t = t * om;
                }
                P = P + hc1[j] * C(g[j]);
                Q = Q + hc2[j] * C(g[j]);
            }
            double *dst = ab + ((size_t) o16 * WN + j) * 4;
            dst[0] = (double) A.re; dst[1] = (double) A.im; dst[2] = (double) B.re; dst[3] = (double) B.im;
        }
        pq[o16 * 4 + 0] = (double) P.re; pq[o16 * 4 + 1] = (double) P.im;
        pq[o16 * 4 + 2] = (double) Q.re; pq[o16 * 4 + 3] = (double) Q.im;
    }
}

// ---- chain_sq_kernel (decimation 2 .. 16 in one pass, kernels_sq.hip) --------------------------------------------------------------
namespace {
typedef long double LD;
struct CL { LD re, im; };
inline CL cmulL(CL a, CL b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// p^k, k = 0 .. kmax
std::vector<CL> powersL(const double *p2, int kmax) {
    std::vector<CL> r((size_t) kmax + 1);
    const CL p = {(LD) p2[0], (LD) p2[1]};
    r[0] = {1.0L, 0.0L};
    for (int k = 1; k <= kmax; ++k) r[k] = cmulL(r[k - 1], p);
    return r;
}
}  // namespace

extern "C" int sdrt_build_sq_tables(const DecimTables *tab, int q, SqTables *out, double *share) {
    const int LC = SDRT_SQ_LC;
    if (!tab || !out || !share || q < 2 || q > 16 || (q & (q - 1))) return -1;
    memset(out, 0, sizeof(*out));
    out->q = q;
    out->nout = LC / q;
    out->ncp = 2 * q >= 16 ? 1 : 16 / (2 * q);
    out->nrt = 2 * out->nout / 16;
    for (int i = 0; i < SDRT_NMODE; ++i) {
        const std::vector<CL> pw = powersL(tab->pole[i], LC);
        out->mu[i][0] = (double) pw[LC].re;
        out->mu[i][1] = (double) pw[LC].im;
        for (int b = 0; b < out->nout; ++b) {
            const int s = b * q;
            share[((size_t) b * 8 + i) * 2 + 0] = (double) (2.0L * pw[s].re);
            share[((size_t) b * 8 + i) * 2 + 1] = (double) (2.0L * pw[s].im);
            share[((size_t) b * 8 + 4 + i) * 2 + 0] = (double) (2.0L * pw[LC - s].re);
            share[((size_t) b * 8 + 4 + i) * 2 + 1] = (double) (2.0L * pw[LC - s].im);
        }
    }
    memcpy(out->head, tab->head, sizeof(out->head));
    memcpy(out->tailF, tab->tailF, sizeof(out->tailF));
    memcpy(out->tailFc, tab->tailFc, sizeof(out->tailFc));
    memcpy(out->tailU, tab->tailU, sizeof(out->tailU));
    return 0;
}

extern "C" int sdrt_build_sq_row(const DecimTables *tab, int q, const double *nco_samp, int S, signed char *agg, double *agg_combine,
                                 signed char *master, double *loc_combine) {
    const int LC = SDRT_SQ_LC, KS = LC / 32;
    if (!tab || !agg || !master || q < 2 || q > 16 || (q & (q - 1))) return -1;
    const int ncp = 2 * q >= 16 ? 1 : 16 / (2 * q);
    std::vector<CL> pw[SDRT_NMODE];
    for (int i = 0; i < SDRT_NMODE; ++i) pw[i] = powersL(tab->pole[i], LC);
    auto phasor = [&](int j) -> CL {                       // exp(i theta j), j in (-256, 256)
        if (!nco_samp) return {1.0L, 0.0L};
        const int a = j < 0 ? -j : j;
        return {(LD) nco_samp[2 * a], j < 0 ? -(LD) nco_samp[2 * a + 1] : (LD) nco_samp[2 * a + 1]};
    };
    // column aggregates: row r = 4 mode + 2 stream + part of tile 0 (causal: ac p^(255 - j)) and tile 1 (anticausal: aa p^j); column
    // 2 j + comp of the 512: what the row takes of component comp of sample j, the sample's phasor exp(i theta j) folded in
    //   stream re = c xr - s xi,  stream im = s xr + c xi;  part re / im of (w x stream) = w.re / w.im x stream
    {
        const int cols = 2 * LC;
        std::vector<double> W((size_t) 32 * cols);
        for (int tile = 0; tile < 2; ++tile)
            for (int i = 0; i < SDRT_NMODE; ++i) {
                const CL a = tile == 0 ? CL{(LD) tab->ac[i][0], (LD) tab->ac[i][1]} : CL{(LD) tab->aa[i][0], (LD) tab->aa[i][1]};
                for (int j = 0; j < LC; ++j) {
                    const CL w = cmulL(a, pw[i][tile == 0 ? LC - 1 - j : j]);
                    const CL ph = phasor(j);
                    for (int st = 0; st < 2; ++st)
                        for (int pt = 0; pt < 2; ++pt) {
                            const LD wv = pt ? w.im : w.re;
                            const LD k0 = st == 0 ? ph.re : ph.im, k1 = st == 0 ? -ph.im : ph.re;
                            const int r = 16 * tile + 4 * i + 2 * st + pt;
                            W[(size_t) r * cols + 2 * j] = (double) (wv * k0);
                            W[(size_t) r * cols + 2 * j + 1] = (double) (wv * k1);
                        }
                }
            }
        *agg_combine = quantise_planes(W.data(), 32, cols, cols, S, [&](int d, int r, int c, signed char dig) {
            // K position c = 2 j + comp, j = 32 ks + 8 kg + jj: byte 2 jj + comp of lane (kg, row)
            const int j = c >> 1, comp = c & 1, ks = j >> 5, kg = (j >> 3) & 3, jj = j & 7;
            agg[((((size_t) (r >> 4) * S + d) * KS + ks) * 64 + (16 * kg + (r & 15))) * 16 + 2 * jj + comp] = dig;
        });
        if (*agg_combine == 0.0) return -1;
    }
    // local master: m[dd] = h[dd] exp(i theta dd), dd = j - s in (-256, 256);  h[dd] = sum_i 2 Re(aa_i p_i^dd) (dd >= 0, + d0p at 0),
    // sum_i 2 Re(ac_i p_i^(-dd - 1)) (dd < 0).  Part 0 (real part of the output): (m.re, -m.im) on (re, im); part 1: (m.im, m.re).
    {
        const int cols = SDRT_SQ_MASTER;
        std::vector<double> W((size_t) 2 * cols, 0.0);
        for (int dd = -(LC - 1); dd <= LC - 1; ++dd) {
            LD h = dd == 0 ? (LD) tab->d0p : 0.0L;
            for (int i = 0; i < SDRT_NMODE; ++i) {
                const CL a = dd >= 0 ? CL{(LD) tab->aa[i][0], (LD) tab->aa[i][1]} : CL{(LD) tab->ac[i][0], (LD) tab->ac[i][1]};
                h += 2.0L * cmulL(a, pw[i][dd >= 0 ? dd : -dd - 1]).re;
            }
            const CL ph = phasor(dd);
            const int pos = 2 * (dd + 256);
            W[(size_t) 0 * cols + pos] = (double) (h * ph.re);
            W[(size_t) 0 * cols + pos + 1] = (double) (-h * ph.im);
            W[(size_t) 1 * cols + pos] = (double) (h * ph.im);
            W[(size_t) 1 * cols + pos + 1] = (double) (h * ph.re);
        }
        memset(master, 0, (size_t) SDRT_SQ_MASTER_BYTES(q, S, ncp));
        const size_t CS = (size_t) SDRT_SQ_CS(q) * 16, PS = (size_t) SDRT_SQ_PS(q, S, ncp) * 16;
        *loc_combine = quantise_planes(W.data(), 2, cols, cols, S, [&](int d, int r, int c, signed char dig) {
            for (int cp = 0; cp < ncp; ++cp) {
                const int y = c + 2 * cp * q;                      // copy cp: the sequence delayed by 2 cp q bytes
                if (y < SDRT_SQ_MASTER) master[(size_t) r * PS + ((size_t) d * ncp + cp) * CS + y] = dig;
            }
        });
        if (*loc_combine == 0.0) return -1;
    }
    return 0;
}
