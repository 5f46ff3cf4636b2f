#pragma once
// Shared device helpers of the MI355X-native sdrterm DSP hot path (gfx950 only).  Included by kernels*.hip.
//
// Kernels (DESIGN.md sections 3, 4):
//   kernels_i8w.hip   chain_i8w_kernel   unpack .. decimate on the int8 matrix cores, 256-sample columns, persistent and
//                                        software-pipelined (the default for 8/16-bit integer input, q = 32 / 64)
//   kernels_i8t.hip   chain_i8t_kernel   the same arithmetic with one block of <= 64 samples per column (other q / chunk sizes)
//   kernels.hip       chain_kernel       stage A on the f64 matrix cores (float / 32-bit input, IQ correction off the wide path),
//                                        IQ prefix kernels, stand-alone discriminators
//   kernels_post.hip  post_fft_kernel / post_kernel / smooth_kernel   FM/AM discriminator, resampler, output lowpass, --smooth-output
//
// Reference steps replaced (peads/sdrterm, paths relative to its root):
//   read_file.py:100-103 (unpack/_normalize/_correctIq), demodulation.py:71-79 (shiftFreq),
//   dsp_processor.py:147 (scipy decimate), demodulation.py:25-48 (fmDemod/amDemod), dsp_processor.py:32-36,149
//   (applyFilters/sosfilt), dsp_processor.py:159-160 (savgol_filter).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"


typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

struct cd {
    double re, im;
};
__device__ __forceinline__ cd mk(double r, double i) { cd c; c.re = r; c.im = i; return c; }
__device__ __forceinline__ cd cmul(cd a, cd b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ cd cadd(cd a, cd b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cd cfma(cd a, cd b, cd c) {  // a*b + c
    return mk(fma(a.re, b.re, fma(-a.im, b.im, c.re)), fma(a.re, b.im, fma(a.im, b.re, c.im)));
}

// decimated sample k of an item, less the share of the IQ offset at its chunk's start (same operations as iq_fix_kernel)
__device__ __forceinline__ cd load_y(const cd *y, const cd *R, const cd h, const int k) {
    cd v = y[k];
    if (R) {
        const cd r = R[k];
        v.re = fma(-h.re, r.re, fma(h.im, r.im, v.re));
        v.im = fma(-h.re, r.im, fma(-h.im, r.re, v.im));
    }
    return v;
}

// which (row, chunk) item of the batch a work item of the demodulation stage is (PostArgs.row_map)
__device__ __forceinline__ size_t post_item(const PostArgs &a, size_t w) {
    if (!a.row_map) return w;
    const size_t k = w / a.n_chunks;
    return (size_t) a.row_map[k] * a.n_chunks + (w - k * a.n_chunks);
}

// atan2 for the pair angles (demodulation.py:30-32: angle()).  t = min(|x|,|y|) / max(|x|,|y|) in [0, 1] by one reciprocal, two
// Newton steps and a residual correction; atan(t) = t q(t^2), q of degree 16 (tools/fit_atan.py: 4e-15 over [0, 1]); octant
// and sign put back with selects.  Signed zeros as C99 / numpy: (+-0, +0) -> +-0, (+-0, -0) -> +-pi.  39 instructions against
// the library's 70 (half of post_fft8_kernel's arithmetic was atan2; round 5: every demodulation kernel takes this one).
__device__ __forceinline__ double atan2_pi(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(fmax(ax, ay), 2.2250738585072014e-308), mn = fmin(ax, ay);    // (0, 0): t = 0 / tiny = 0
    double r = __builtin_amdgcn_rcp(mx);
    r = fma(fma(-mx, r, 1.0), r, r);
    r = fma(fma(-mx, r, 1.0), r, r);
    double t = mn * r;
    t = fma(fma(-mx, t, mn), r, t);
    const double s = t * t;
    double p = 7.03825190772905e-05;
    p = fma(p, s, -0.0006752461871228126);
    p = fma(p, s, 0.0030574650642366205);
    p = fma(p, s, -0.008772278251586507);
    p = fma(p, s, 0.01816674618350375);
    p = fma(p, s, -0.02955359506524126);
    p = fma(p, s, 0.0404892227192373);
    p = fma(p, s, -0.049763055792462756);
    p = fma(p, s, 0.057934563768719034);
    p = fma(p, s, -0.06645912139302747);
    p = fma(p, s, 0.07688758456599037);
    p = fma(p, s, -0.09090481197187922);
    p = fma(p, s, 0.11111076687817362);
    p = fma(p, s, -0.1428571258641315);
    p = fma(p, s, 0.19999999955253506);
    p = fma(p, s, -0.3333333333285598);
    p = fma(p, s, 0.9999999999999907);
    double u = t * p;
    u = ay > ax ? 1.5707963267948966 - u : u;
    u = __double2hiint(x) < 0 ? 3.141592653589793 - u : u;
    return copysign(u, y);
}

// DPP row shifts inside each 16-lane row, zero fill at the row boundary.  row_shr:n = 0x110+n, row_shl:n = 0x100+n.
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);     // bound_ctrl: lanes shifted in from outside the row read 0
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ cd dpp_row(cd v) { return mk(dpp_row<CTRL>(v.re), dpp_row<CTRL>(v.im)); }

// compile-time loop: f(std::integral_constant<int, S>{}) for S = 0 .. N-1 (DPP controls must be literal constants)
template <int S, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (S < N) {
        f(std::integral_constant<int, S>{});
        static_for<S + 1, N>(f);
    }
}

// Hillis-Steele inclusive scan along the lanes of one wave: S_l = sum_{j<=l} mu^(l-j) x_j  (x = 0 beyond the data).
__device__ __forceinline__ cd wave_scan_geometric(cd x, cd mu, int lane) {
    cd mpow = mu;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = 1 << i;
        cd t = mk(__shfl_up(x.re, d), __shfl_up(x.im, d));
        if (lane < d) t = mk(0.0, 0.0);
        x = cfma(mpow, t, x);
        mpow = cmul(mpow, mpow);
    }
    return x;
}

// Circulant product of the FM resampler for one wave: lane l owns the R outputs j = R*l + r of
// z_odd[j] = sum_k a[k] hr[(k - j) mod N], N = 64 R.  A sliding register window feeds the R outputs from one new LDS
// value of hr per k; a[k] is a broadcast LDS read.
template <int R>
__device__ __forceinline__ void resample_wave(const double *a, const double *hr, double *zodd, int N, int lane) {
    const int kq = N, k0 = 0, j0 = R * lane, msk = N - 1;
    double acc[R], B[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r] = 0.0; B[r] = r ? hr[(k0 - j0 - r) & msk] : 0.0; }
    for (int k = k0; k < k0 + kq; k += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            B[(R - u) % R] = hr[(k + u - j0) & msk];
            const double ak = a[k + u];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(ak, B[(r - u + R) % R], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i = 2 * (j0 + r) + 1; zodd[i + (i >> 3)] = acc[r]; }   // padded layout of the post kernels (zpos)
}

// ---- input decoding (read_file.py:51,100-101; file_util.py:44-56) -------------------------------------------------
template <int DT> struct Enc;
template <> struct Enc<SDRT_S8> { typedef int8_t raw; static constexpr int size = 1; };
template <> struct Enc<SDRT_U8> { typedef uint8_t raw; static constexpr int size = 1; };
template <> struct Enc<SDRT_S16> { typedef uint16_t raw; static constexpr int size = 2; };
template <> struct Enc<SDRT_U16> { typedef uint16_t raw; static constexpr int size = 2; };
template <> struct Enc<SDRT_S32> { typedef uint32_t raw; static constexpr int size = 4; };
template <> struct Enc<SDRT_U32> { typedef uint32_t raw; static constexpr int size = 4; };
template <> struct Enc<SDRT_F32> { typedef uint32_t raw; static constexpr int size = 4; };
template <> struct Enc<SDRT_F64> { typedef uint64_t raw; static constexpr int size = 8; };

template <int DT>
__device__ __forceinline__ double decode(typename Enc<DT>::raw v, bool swap) {
    if constexpr (DT == SDRT_S8) return (double) (int) v;
    else if constexpr (DT == SDRT_U8) return (double) (unsigned) v;
    else if constexpr (DT == SDRT_S16) { uint16_t u = swap ? __builtin_bswap16(v) : v; return (double) (int) (int16_t) u; }
    else if constexpr (DT == SDRT_U16) { uint16_t u = swap ? __builtin_bswap16(v) : v; return (double) (unsigned) u; }
    else if constexpr (DT == SDRT_S32) { uint32_t u = swap ? __builtin_bswap32(v) : v; return (double) (int32_t) u; }
    else if constexpr (DT == SDRT_U32) { uint32_t u = swap ? __builtin_bswap32(v) : v; return (double) u; }
    else if constexpr (DT == SDRT_F32) { uint32_t u = swap ? __builtin_bswap32(v) : v; return (double) __uint_as_float(u); }
    else { uint64_t u = swap ? __builtin_bswap64(v) : v; return __longlong_as_double((long long) u); }
}

// BYTES contiguous bytes per lane (BYTES-aligned) into 32-bit words held in registers, widest loads available.
// STREAM: once-read data (the raw sample stream of the persistent kernel): non-temporal loads keep it from displacing the
// tables and digit planes that every workgroup re-reads from L2.
template <int BYTES, bool STREAM = false>
__device__ __forceinline__ void load_words(const unsigned char *p, uint32_t *dst) {
    if constexpr (BYTES % 16 == 0) {
#pragma unroll
        for (int i = 0; i < BYTES / 16; ++i) {
            typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
            uint4 v;
            if constexpr (STREAM) {
                const v4u_ t = __builtin_nontemporal_load((const v4u_ *) p + i);
                v = make_uint4(t[0], t[1], t[2], t[3]);
            } else {
                v = ((const uint4 *) p)[i];
            }
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
    } else if constexpr (BYTES == 8) {
        const uint2 v = *(const uint2 *) p;
        dst[0] = v.x; dst[1] = v.y;
    } else if constexpr (BYTES == 4) {
        dst[0] = *(const uint32_t *) p;
    } else {
        static_assert(BYTES == 2, "unsupported per-lane byte count");
        dst[0] = *(const uint16_t *) p;
    }
}

// item IDX (compile-time) of the encoding out of the word array: pure register arithmetic, no memory punning
template <int DT, int IDX>
__device__ __forceinline__ typename Enc<DT>::raw get_item(const uint32_t *wds) {
    constexpr int SZ = Enc<DT>::size;
    if constexpr (SZ == 1) return (typename Enc<DT>::raw) ((wds[IDX >> 2] >> (8 * (IDX & 3))) & 0xffu);
    else if constexpr (SZ == 2) return (typename Enc<DT>::raw) ((wds[IDX >> 1] >> (16 * (IDX & 1))) & 0xffffu);
    else if constexpr (SZ == 4) return (typename Enc<DT>::raw) wds[IDX];
    else return (typename Enc<DT>::raw) ((uint64_t) wds[2 * IDX] | ((uint64_t) wds[2 * IDX + 1] << 32));
}


struct LdsLayout {
    int o, carF, carG, cF, cG, tailF, tailFc, f0, gu, tpow, total;   // offsets in 16-byte units
};
__host__ __device__ inline LdsLayout lds_layout(int n, int L, int m, int G) {
    const int nblk = n / L, nt = nblk / (SDRT_TB * G);
    LdsLayout l;
    int p = 0;
    l.o = p; p += nblk;                      // local output part per block
    l.carF = p; p += (nt + 1) * 8;           // [t][mode][stream]
    l.carG = p; p += (nt + 1) * 8;
    l.cF = p; p += SDRT_TB * G * 4;
    l.cG = p; p += SDRT_TB * G * 4;
    l.tailF = p; p += 16;
    l.tailFc = p; p += 16;
    l.f0 = p; p += 8;
    l.gu = p; p += 8;
    l.tpow = p; p += nt <= SDRT_TPOW ? (nt + 1) * 4 : 0;   // tile multiplier powers [k][mode]
    l.total = p;
    return l;
}


// ---------------------------------------------------------------------------------------------------------------
// Everything after the main loop, shared by both chain kernels: tables + edge operators, tile carries, combine, store of
// the decimated samples (the post kernel demodulates, resamples and low-passes them).  `edge(idx)` returns sample idx of this chunk as the decimator sees
// it (after bias / IQ correction / NCO).  `ybias` (optional, [m]) is added to the decimated samples: the int8 kernel
// filters x - offset and adds the filter's response to the constant offset here.
// ---------------------------------------------------------------------------------------------------------------
template <class EdgeFn>
__device__ __forceinline__ void chain_tail(const DecimArgs &a, const int L, const int row, const long chunk, unsigned char *smem,
                                           unsigned long long *stp, EdgeFn edge, const cd *ybias, const bool copy_tables = true) {
#define SDRT_STAMP(i) do { if (stp) stp[i] = __builtin_amdgcn_s_memtime(); } while (0)
    const DecimTables *__restrict__ tab = a.tab;
    const int n = a.n, m = a.m, nt = a.n_tiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TBG = SDRT_TB * tab->G;                // blocks per tile
    const LdsLayout lo = lds_layout(n, L, m, tab->G);
    cd *base = (cd *) smem;
    cd *o_lds = base + lo.o, *carF = base + lo.carF, *carG = base + lo.carG;
    cd *cF = base + lo.cF, *cG = base + lo.cG, *tailF = base + lo.tailF;
    cd *tailFc = base + lo.tailFc, *F0 = base + lo.f0, *Gu = base + lo.gu, *tpw = base + lo.tpow;
    // ---- tables for the later phases and the edge operators (all loads issued together, one round trip) -------------
    {
        const int o = tid >> 5, l = tid & 31, mode = o >> 1, st = o & 1;
        cd f = mk(0.0, 0.0), gq = mk(0.0, 0.0);
        if (l <= SDRT_EDGE) {
            // head sample l and tail sample n-28+l of this chunk, after normalise / IQ correction / NCO; this thread
            // needs stream st
            const long ih = l, it = (long) n - (SDRT_EDGE + 1) + l;
            const cd eh = edge(ih), et = edge(it);
            const double hr_ = eh.re, hi_ = eh.im, tr_ = et.re, ti_ = et.im;
            const double uh = st ? hi_ : hr_, ut = st ? ti_ : tr_;
            f = mk(tab->head[mode][l][0] * uh, tab->head[mode][l][1] * uh);
            gq = mk(tab->tailU[mode][l][0] * ut, tab->tailU[mode][l][1] * ut);
        }
        if (copy_tables) {
            if (tid < TBG * 4) {
                cF[tid] = mk(tab->cF[tid >> 2][tid & 3][0], tab->cF[tid >> 2][tid & 3][1]);
                cG[tid] = mk(tab->cG[tid >> 2][tid & 3][0], tab->cG[tid >> 2][tid & 3][1]);
            } else if (tid >= 128 && tid < 144) {
                const int i = tid - 128;
                tailF[i] = mk(tab->tailF[i >> 2][i & 3][0], tab->tailF[i >> 2][i & 3][1]);
                tailFc[i] = mk(tab->tailFc[i >> 2][i & 3][0], tab->tailFc[i >> 2][i & 3][1]);
            }
            if (nt <= SDRT_TPOW)
                for (int i = tid; i < (nt + 1) * 4; i += 256) tpw[i] = mk(tab->tpow[i >> 2][i & 3][0], tab->tpow[i >> 2][i & 3][1]);
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
            f.re += __shfl_xor(f.re, d);
            f.im += __shfl_xor(f.im, d);
            gq.re += __shfl_xor(gq.re, d);
            gq.im += __shfl_xor(gq.im, d);
        }
        if (l == 0) { F0[o] = f; Gu[o] = gq; }
    }
    __syncthreads();
    SDRT_STAMP(3);

    // ---- phase 2: carries across tiles as wave-level scans.  Chain o = mode*2 + stream; wave w owns chains 2w, 2w+1.
    // F[t+1] = mu16 F[t] + aggF[t]  (slots of carF),  G[t] = mu16 G[t+1] + aggG[t]  (slots of carG).
    {
        for (int cc = 0; cc < 2; ++cc) {
            const int o = wave * 2 + cc, mode = o >> 1;
            const cd mu16 = mk(tab->mu_pow[4][mode][0], tab->mu_pow[4][mode][1]);
            cd carry = F0[o];
            if (lane == 0) carF[o] = carry;
            for (int t0 = 0; t0 < nt; t0 += 64) {
                const int t = t0 + lane;
                cd x = t < nt ? carF[(t + 1) * 8 + o] : mk(0.0, 0.0);
                if (lane == 0) x = cfma(mu16, carry, x);
                x = wave_scan_geometric(x, mu16, lane);
                if (t < nt) carF[(t + 1) * 8 + o] = x;
                carry = mk(__shfl(x.re, 63), __shfl(x.im, 63));
            }
        }
        __syncthreads();
        for (int cc = 0; cc < 2; ++cc) {
            const int o = wave * 2 + cc, mode = o >> 1, st = o & 1;
            const cd mu16 = mk(tab->mu_pow[4][mode][0], tab->mu_pow[4][mode][1]);
            cd G = Gu[o];
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cd Fj = carF[nt * 8 + j * 2 + st];
                G = cfma(tailF[mode * 4 + j], Fj, G);
                G = cfma(tailFc[mode * 4 + j], mk(Fj.re, -Fj.im), G);
            }
            if (lane == 0) carG[nt * 8 + o] = G;
            cd carry = G;
            for (int t0 = 0; t0 < nt; t0 += 64) {
                const int t = nt - 1 - (t0 + lane);              // lanes walk the tiles from the right edge
                cd x = t >= 0 ? carG[t * 8 + o] : mk(0.0, 0.0);
                if (lane == 0) x = cfma(mu16, carry, x);
                x = wave_scan_geometric(x, mu16, lane);
                if (t >= 0) carG[t * 8 + o] = x;
                carry = mk(__shfl(x.re, 63), __shfl(x.im, 63));
            }
        }
        __syncthreads();

    }
    SDRT_STAMP(4);
    // ---- phase 3: add the carried parts at the output blocks -> decimated samples y, stored for the post kernel ---------
    const int stride = tab->out_stride;
    double *yd = a.ydec + ((size_t) row * a.n_chunks + chunk) * (size_t) (a.fix_pairs ? m / 2 : m) * 2;
    for (int k = tid; k < m; k += 256) {
        const int kb = k * stride, tile = kb / TBG, nbk = kb % TBG;
        cd v = o_lds[kb];
#pragma unroll
        for (int mode = 0; mode < 4; ++mode) {
            const cd f = cF[nbk * 4 + mode], gq = cG[nbk * 4 + mode];
            const cd Fa = carF[tile * 8 + mode * 2 + 0], Fb = carF[tile * 8 + mode * 2 + 1];
            const cd Ga = carG[(tile + 1) * 8 + mode * 2 + 0], Gb = carG[(tile + 1) * 8 + mode * 2 + 1];
            v.re += 2.0 * (f.re * Fa.re - f.im * Fa.im + gq.re * Ga.re - gq.im * Ga.im);
            v.im += 2.0 * (f.re * Fb.re - f.im * Fb.im + gq.re * Gb.re - gq.im * Gb.im);
        }
        if (ybias) { v.re += ybias[k].re; v.im += ybias[k].im; }
        else { v.re += 0.0; v.im += 0.0; }           // -0.0 -> +0.0: angle(-0) would be pi where the reference's exact zeros give 0
        if (a.fix_pairs) {
            // the second look of stage_a = auto: the item's place holds pair products (as the wide kernel leaves them); output
            // k + 1 sits in the next lane (m is even and a multiple of the wave here)
            const cd w = mk(__shfl_down(v.re, 1), __shfl_down(v.im, 1));
            if (!(k & 1)) *(double2 *) (yd + (size_t) k) = make_double2(v.re * w.re + v.im * w.im, v.im * w.re - v.re * w.im);
        } else
        *(double2 *) (yd + 2 * (size_t) k) = make_double2(v.re, v.im);
    }
    SDRT_STAMP(5);
}
#undef SDRT_STAMP

}  // namespace
