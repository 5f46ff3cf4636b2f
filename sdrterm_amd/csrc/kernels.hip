// Device kernels of the MI355X-native sdrterm DSP hot path (gfx950 only).
//
// K1  decim_demod_kernel   one workgroup per (chunk, row): unpack -> normalise -> [IQ correction] -> NCO ->
//                          zero-phase order-8 decimator in modal block form (stage A on the f64 matrix cores,
//                          in-tile scans with DPP row shifts, tile carries through LDS) -> FM pair-angle / AM.
// K2  post_kernel          one workgroup per (chunk, row): FM resample m/2 -> m, output lowpass sosfilt.
//
// Reference steps replaced (peads/sdrterm, paths relative to its root):
//   read_file.py:100-103 (unpack/_normalize/_correctIq), demodulation.py:71-79 (shiftFreq),
//   dsp_processor.py:147 (scipy decimate), demodulation.py:25-48 (fmDemod/amDemod), dsp_processor.py:32-36,149
//   (applyFilters/sosfilt).
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// DPP row shifts inside each 16-lane row, zero fill at the row boundary.  row_shr:n = 0x110+n, row_shl:n = 0x100+n.
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ cd dpp_row(cd v) { return mk(dpp_row<CTRL>(v.re), dpp_row<CTRL>(v.im)); }

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

// BYTES contiguous bytes per lane, BYTES-aligned, widest loads available.
template <int BYTES>
__device__ __forceinline__ void load_bytes(const unsigned char *p, void *dst) {
    if constexpr (BYTES % 16 == 0) {
#pragma unroll
        for (int i = 0; i < BYTES / 16; ++i) ((uint4 *) dst)[i] = ((const uint4 *) p)[i];
    } else if constexpr (BYTES == 8) {
        *(uint2 *) dst = *(const uint2 *) p;
    } else if constexpr (BYTES == 4) {
        *(uint32_t *) dst = *(const uint32_t *) p;
    } else {
        static_assert(BYTES == 2, "unsupported per-lane byte count");
        *(uint16_t *) dst = *(const uint16_t *) p;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// K1: decimate + demodulate.
// Workgroup = 256 threads (4 waves) per (chunk, row).  A tile is 16 consecutive blocks of L samples; wave w takes
// tiles w, w+4, ...  Lane l = (g = l>>4, nb = l&15) loads the L/4 consecutive samples [nb*L + g*L/4, +L/4) of its
// tile: they are the k-slices {4s+g} of the MFMA B operand for block nb (column nb).
// ---------------------------------------------------------------------------------------------------------------
template <int L, int DT, bool NCO>
__global__ __launch_bounds__(256) void decim_demod_kernel(DecimArgs a) {
    constexpr int KS = L / 4;                       // k-steps == samples per lane per tile
    constexpr int ITEM = Enc<DT>::size;
    constexpr int BYTES = KS * 2 * ITEM;            // per lane per tile
    constexpr int TILE = SDRT_TB * L;               // samples per tile
    typedef typename Enc<DT>::raw raw_t;

    extern __shared__ __align__(16) unsigned char smem[];
    const DecimTables *__restrict__ tab = a.tab;
    const int n = a.n, m = a.m, nt = a.n_tiles, nblk = n / L;
    const int row = blockIdx.x % a.rows;
    const long chunk = blockIdx.x / a.rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nb = lane & 15;
    const bool swap = a.swap != 0;

    // LDS carve-up (all 16-byte aligned)
    cd *o_lds = (cd *) smem;                         // [nblk]      local output part per block
    cd *y_lds = o_lds + nblk;                        // [m]         decimated samples
    cd *aggF = y_lds + m;                            // [nt][4][2]  causal tile aggregates (mode, stream)
    cd *aggG = aggF + nt * 8;                        // [nt][4][2]
    cd *carF = aggG + nt * 8;                        // [nt+1][4][2]
    cd *carG = carF + (nt + 1) * 8;                  // [nt+1][4][2]
    cd *edge_u = carG + (nt + 1) * 8;                // [2][28] head / tail samples after normalise/NCO
    cd *nco_lds = edge_u + 2 * (SDRT_EDGE + 1);      // [TILE]      only if NCO

    const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk * (size_t) n * 2 * ITEM;
    const double nsc = a.norm_scale, nbias = a.norm_bias;
    const cd *nco_tile_g = (const cd *) a.nco_tile + (size_t) row * TILE;
    const cd *nco_phase_g = (const cd *) a.nco_phase + (size_t) row * nt;

    if constexpr (NCO) {
        for (int i = tid; i < TILE; i += 256) nco_lds[i] = nco_tile_g[i];
    }
    // Edge samples (first and last 28 of the chunk), after normalise and NCO.
    if (tid < 2 * (SDRT_EDGE + 1)) {
        const int which = tid / (SDRT_EDGE + 1), l = tid % (SDRT_EDGE + 1);
        const long idx = which ? (long) n - (SDRT_EDGE + 1) + l : l;
        const raw_t *rp = (const raw_t *) chunk_raw + 2 * idx;
        double xr = fma(decode<DT>(rp[0], swap), nsc, nbias);
        double xi = decode<DT>(rp[1], swap) * nsc;
        if constexpr (NCO) {
            const int tl = (int) (idx % TILE);
            cd c = cmul(nco_tile_g[tl], nco_phase_g[idx / TILE]);
            cd u = cmul(mk(xr, xi), c);
            xr = u.re;
            xi = u.im;
        }
        edge_u[tid] = mk(xr, xi);
    }

    // Stage-A weights for this lane: A operand of v_mfma_f64_16x16x4: lane holds A[row = l&15][k = l>>4].
    double w[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) w[s] = tab->W[lane & 15][g * KS + s];
    // scan multipliers of this lane's mode g
    cd mup[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mup[k] = mk(tab->mu_pow[k][g][0], tab->mu_pow[k][g][1]);
    const double d0p = tab->d0p;

    if constexpr (NCO) __syncthreads();

    for (int tile = wave; tile < nt; tile += 4) {
        alignas(16) unsigned char buf[BYTES < 16 ? 16 : BYTES];
        const size_t soff = (size_t) tile * TILE + (size_t) nb * L + (size_t) g * KS;  // first sample of this lane
        load_bytes<BYTES>(chunk_raw + soff * 2 * ITEM, buf);
        const raw_t *rb = (const raw_t *) buf;

        v4d acc_r = {0.0, 0.0, 0.0, 0.0}, acc_i = {0.0, 0.0, 0.0, 0.0};
        double u0r = 0.0, u0i = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            double xr = fma(decode<DT>(rb[2 * s], swap), nsc, nbias);
            double xi = decode<DT>(rb[2 * s + 1], swap) * nsc;
            if constexpr (NCO) {
                const cd c = nco_lds[nb * L + g * KS + s];
                const double tr = xr * c.re - xi * c.im;
                xi = xr * c.im + xi * c.re;
                xr = tr;
            }
            if (s == 0) { u0r = xr; u0i = xi; }
            acc_r = __builtin_amdgcn_mfma_f64_16x16x4f64(w[s], xr, acc_r, 0, 0, 0);
            acc_i = __builtin_amdgcn_mfma_f64_16x16x4f64(w[s], xi, acc_i, 0, 0, 0);
        }
        // D layout: lane (g, nb), register v holds row g + 4v of column nb:
        //   v=0 Re causal(mode g), v=1 Im causal, v=2 Re anticausal, v=3 Im anticausal; acc_r / acc_i = re / im stream.
        cd Cr = mk(acc_r[0], acc_r[1]), Ar = mk(acc_r[2], acc_r[3]);
        cd Ci = mk(acc_i[0], acc_i[1]), Ai = mk(acc_i[2], acc_i[3]);
        if constexpr (NCO) {
            // the tile's common phasor psi multiplies the complex DATA: re' = pr*re - pi*im, im' = pi*re + pr*im
            const cd psi = nco_phase_g[tile];
            cd t;
            t = mk(psi.re * Cr.re - psi.im * Ci.re, psi.re * Cr.im - psi.im * Ci.im);
            Ci = mk(psi.im * Cr.re + psi.re * Ci.re, psi.im * Cr.im + psi.re * Ci.im);
            Cr = t;
            t = mk(psi.re * Ar.re - psi.im * Ai.re, psi.re * Ar.im - psi.im * Ai.im);
            Ai = mk(psi.im * Ar.re + psi.re * Ai.re, psi.im * Ar.im + psi.re * Ai.im);
            Ar = t;
            const double tr = psi.re * u0r - psi.im * u0i;
            u0i = psi.im * u0r + psi.re * u0i;
            u0r = tr;
        }
        // in-tile inclusive scans over the 16 blocks (lanes of one DPP row): causal with row_shr, anticausal with row_shl
        Cr = cfma(mup[0], dpp_row<0x111>(Cr), Cr);
        Ci = cfma(mup[0], dpp_row<0x111>(Ci), Ci);
        Ar = cfma(mup[0], dpp_row<0x101>(Ar), Ar);
        Ai = cfma(mup[0], dpp_row<0x101>(Ai), Ai);
        Cr = cfma(mup[1], dpp_row<0x112>(Cr), Cr);
        Ci = cfma(mup[1], dpp_row<0x112>(Ci), Ci);
        Ar = cfma(mup[1], dpp_row<0x102>(Ar), Ar);
        Ai = cfma(mup[1], dpp_row<0x102>(Ai), Ai);
        Cr = cfma(mup[2], dpp_row<0x114>(Cr), Cr);
        Ci = cfma(mup[2], dpp_row<0x114>(Ci), Ci);
        Ar = cfma(mup[2], dpp_row<0x104>(Ar), Ar);
        Ai = cfma(mup[2], dpp_row<0x104>(Ai), Ai);
        Cr = cfma(mup[3], dpp_row<0x118>(Cr), Cr);
        Ci = cfma(mup[3], dpp_row<0x118>(Ci), Ci);
        Ar = cfma(mup[3], dpp_row<0x108>(Ar), Ar);
        Ai = cfma(mup[3], dpp_row<0x108>(Ai), Ai);
        // tile aggregates: causal inclusive at the last block, anticausal inclusive at the first block
        if (nb == 15) { aggF[(tile * 4 + g) * 2 + 0] = Cr; aggF[(tile * 4 + g) * 2 + 1] = Ci; }
        if (nb == 0) { aggG[(tile * 4 + g) * 2 + 0] = Ar; aggG[(tile * 4 + g) * 2 + 1] = Ai; }
        // local output part: direct term + 2 Re(exclusive causal + inclusive anticausal), summed over the 4 modes
        const double cexr = dpp_row<0x111>(Cr.re), cexi = dpp_row<0x111>(Ci.re);
        double pr = 2.0 * (cexr + Ar.re), pi = 2.0 * (cexi + Ai.re);
        if (g == 0) { pr = fma(d0p, u0r, pr); pi = fma(d0p, u0i, pi); }
        pr += __shfl_xor(pr, 16);
        pi += __shfl_xor(pi, 16);
        pr += __shfl_xor(pr, 32);
        pi += __shfl_xor(pi, 32);
        if (g == 0) o_lds[tile * SDRT_TB + nb] = mk(pr, pi);
    }
    __syncthreads();

    // Phase 2: carries across tiles.  Threads 0..7 = (mode, stream).
    if (tid < 8) {
        const int mode = tid & 3, st = tid >> 2;
        const cd mu16 = mk(tab->mu_pow[4][mode][0], tab->mu_pow[4][mode][1]);
        cd F = mk(0.0, 0.0);
        for (int l = 0; l <= SDRT_EDGE; ++l) {
            const double u = st ? edge_u[l].im : edge_u[l].re;
            F.re = fma(tab->head[mode][l][0], u, F.re);
            F.im = fma(tab->head[mode][l][1], u, F.im);
        }
        carF[mode * 2 + st] = F;
        for (int t = 0; t < nt; ++t) {
            F = cfma(mu16, F, aggF[(t * 4 + mode) * 2 + st]);
            carF[((t + 1) * 4 + mode) * 2 + st] = F;
        }
    }
    __syncthreads();
    if (tid < 8) {
        const int mode = tid & 3, st = tid >> 2;
        const cd mu16 = mk(tab->mu_pow[4][mode][0], tab->mu_pow[4][mode][1]);
        cd G = mk(0.0, 0.0);
        for (int j = 0; j < 4; ++j) {
            const cd Fj = carF[(nt * 4 + j) * 2 + st];
            G = cfma(mk(tab->tailF[mode][j][0], tab->tailF[mode][j][1]), Fj, G);
            G = cfma(mk(tab->tailFc[mode][j][0], tab->tailFc[mode][j][1]), mk(Fj.re, -Fj.im), G);
        }
        for (int l = 0; l <= SDRT_EDGE; ++l) {
            const cd e = edge_u[(SDRT_EDGE + 1) + l];
            const double u = st ? e.im : e.re;
            G.re = fma(tab->tailU[mode][l][0], u, G.re);
            G.im = fma(tab->tailU[mode][l][1], u, G.im);
        }
        carG[(nt * 4 + mode) * 2 + st] = G;
        for (int t = nt - 1; t >= 0; --t) {
            G = cfma(mu16, G, aggG[(t * 4 + mode) * 2 + st]);
            carG[(t * 4 + mode) * 2 + st] = G;
        }
    }
    __syncthreads();

    // Phase 3: add the carried parts at the output blocks.
    const int stride = tab->out_stride;
    for (int k = tid; k < m; k += 256) {
        const int kb = k * stride, tile = kb >> 4, nbk = kb & 15;
        cd v = o_lds[kb];
#pragma unroll
        for (int mode = 0; mode < 4; ++mode) {
            const double fr = tab->cF[nbk][mode][0], fi = tab->cF[nbk][mode][1];
            const double gr = tab->cG[nbk][mode][0], gi = tab->cG[nbk][mode][1];
            const cd F0 = carF[(tile * 4 + mode) * 2 + 0], F1 = carF[(tile * 4 + mode) * 2 + 1];
            const cd G0 = carG[((tile + 1) * 4 + mode) * 2 + 0], G1 = carG[((tile + 1) * 4 + mode) * 2 + 1];
            v.re += 2.0 * (fr * F0.re - fi * F0.im + gr * G0.re - gi * G0.im);
            v.im += 2.0 * (fr * F1.re - fi * F1.im + gr * G1.re - gi * G1.im);
        }
        y_lds[k] = v;
        if (a.ydec) {
            double *yd = a.ydec + (((size_t) row * a.n_chunks + chunk) * m + k) * 2;
            yd[0] = v.re;
            yd[1] = v.im;
        }
    }
    __syncthreads();

    // Phase 4: demodulate (demodulation.py:25-32 pair angle; :41-48 |d^2|).  FM resampling happens in K2.
    if (a.mode == SDRT_FM) {
        const int h = m >> 1;
        double *mid = a.mid + ((size_t) row * a.n_chunks + chunk) * h;
        for (int j = tid; j < h; j += 256) {
            const cd p = y_lds[2 * j], c = y_lds[2 * j + 1];
            const double re = p.re * c.re + p.im * c.im;   // d[i] * conj(d[i+1])
            const double im = p.im * c.re - p.re * c.im;
            mid[j] = atan2(im, re);
        }
    } else {
        double *mid = a.mid + ((size_t) row * a.n_chunks + chunk) * m;
        for (int k = tid; k < m; k += 256) {
            const cd v = y_lds[k];
            mid[k] = v.re * v.re + v.im * v.im;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K2: FM resample (scipy.signal.resample m/2 -> m as a circulant product on the odd samples) and output lowpass.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void post_kernel(PostArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m, tid = threadIdx.x;
    const size_t item = blockIdx.x;   // row * n_chunks + chunk
    double *z = (double *) smem;      // [m]
    if (a.mode == SDRT_FM) {
        const int N = m >> 1;
        double *av = z + m;           // [N]
        double *h = av + N;           // [N]
        const double *mid = a.mid + item * N;
        for (int i = tid; i < N; i += 256) { av[i] = mid[i]; h[i] = a.resample_odd[i]; }
        __syncthreads();
        for (int j = tid; j < N; j += 256) {
            double s = 0.0;
            for (int k = 0; k <= j; ++k) s = fma(av[k], h[j - k], s);
            for (int k = j + 1; k < N; ++k) s = fma(av[k], h[j - k + N], s);
            z[2 * j] = av[j];
            z[2 * j + 1] = s;
        }
    } else {
        const double *mid = a.mid + item * m;
        for (int i = tid; i < m; i += 256) z[i] = mid[i];
    }
    __syncthreads();
    // scipy.signal.sosfilt, zero initial state per chunk (dsp_processor.py:32-36,149): transposed direct form II.
    if (tid == 0) {
        double st[2 * SDRT_MAX_OUT_SECTIONS];
        for (int s = 0; s < 2 * SDRT_MAX_OUT_SECTIONS; ++s) st[s] = 0.0;
        double *out = a.out + item * m;
        for (int i = 0; i < m; ++i) {
            double xc = z[i];
            for (int s = 0; s < a.out_sections; ++s) {
                const double *c = a.out_sos + 6 * s;
                const double xn = fma(c[0], xc, st[2 * s]);
                st[2 * s] = fma(c[1], xc, fma(-c[4], xn, st[2 * s + 1]));
                st[2 * s + 1] = fma(c[2], xc, -c[5] * xn);
                xc = xn;
            }
            out[i] = xc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA layout self-test: D = A(16x4) * B(4x16) with one v_mfma_f64_16x16x4 and the operand maps K1 assumes.
// ---------------------------------------------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(const double *A, const double *B, double *D) {
    const int lane = threadIdx.x & 63;
    const double av = A[(lane & 15) * 4 + (lane >> 4)];      // A[i = l&15][k = l>>4]
    const double bv = B[(lane >> 4) * 16 + (lane & 15)];     // B[k = l>>4][j = l&15]
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) D[((lane >> 4) + 4 * v) * 16 + (lane & 15)] = acc[v];   // row (l>>4)+4v, col l&15
}

// ---- launch helpers ------------------------------------------------------------------------------------------
size_t decim_lds_bytes(int n, int L, int m, bool nco) {
    const int nblk = n / L, nt = nblk / SDRT_TB;
    size_t cds = (size_t) nblk + m + 2 * (size_t) nt * 8 + 2 * (size_t) (nt + 1) * 8 + 2 * (SDRT_EDGE + 1);
    if (nco) cds += (size_t) SDRT_TB * L;
    return cds * 16;
}

template <int L, int DT>
static hipError_t launch_decim_ldt(const DecimArgs &a, bool nco, dim3 grid, size_t lds, hipStream_t s) {
    // more than 64 KiB of dynamic LDS has to be opted into per kernel (gfx950 has 160 KiB per workgroup)
    if (nco) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *) decim_demod_kernel<L, DT, true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((decim_demod_kernel<L, DT, true>), grid, dim3(256), lds, s, a);
    } else {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *) decim_demod_kernel<L, DT, false>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((decim_demod_kernel<L, DT, false>), grid, dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

template <int L>
static hipError_t launch_decim_l(const DecimArgs &a, int dt, bool nco, dim3 grid, size_t lds, hipStream_t s) {
    switch (dt) {
        case SDRT_S8: return launch_decim_ldt<L, SDRT_S8>(a, nco, grid, lds, s);
        case SDRT_U8: return launch_decim_ldt<L, SDRT_U8>(a, nco, grid, lds, s);
        case SDRT_S16: return launch_decim_ldt<L, SDRT_S16>(a, nco, grid, lds, s);
        case SDRT_U16: return launch_decim_ldt<L, SDRT_U16>(a, nco, grid, lds, s);
        case SDRT_S32: return launch_decim_ldt<L, SDRT_S32>(a, nco, grid, lds, s);
        case SDRT_U32: return launch_decim_ldt<L, SDRT_U32>(a, nco, grid, lds, s);
        case SDRT_F32: return launch_decim_ldt<L, SDRT_F32>(a, nco, grid, lds, s);
        case SDRT_F64: return launch_decim_ldt<L, SDRT_F64>(a, nco, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_decim(const DecimArgs &a, int L, int dt, bool nco, size_t n_chunks, hipStream_t s) {
    const dim3 grid((unsigned) (n_chunks * (size_t) a.rows));
    const size_t lds = decim_lds_bytes(a.n, L, a.m, nco);
    switch (L) {
        case 8: return launch_decim_l<8>(a, dt, nco, grid, lds, s);
        case 16: return launch_decim_l<16>(a, dt, nco, grid, lds, s);
        case 32: return launch_decim_l<32>(a, dt, nco, grid, lds, s);
        case 64: return launch_decim_l<64>(a, dt, nco, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

size_t post_lds_bytes(int m, int mode) { return (size_t) (mode == SDRT_FM ? 2 * m : m) * sizeof(double); }

hipError_t launch_post(const PostArgs &a, size_t items, hipStream_t s) {
    hipLaunchKernelGGL(post_kernel, dim3((unsigned) items), dim3(256), post_lds_bytes(a.m, a.mode), s, a);
    return hipGetLastError();
}

hipError_t launch_mfma_selftest(const double *A, const double *B, double *D, hipStream_t s) {
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, s, A, B, D);
    return hipGetLastError();
}
