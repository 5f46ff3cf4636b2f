// chain_sq_kernel: unpack .. decimate for decimation 2, 4, 8, 16 of 16-bit records in ONE pass over the input, on the int8 matrix
// cores (src/sdrterm.py:80 makes 2 the CLI's default; scipy.signal.decimate at src/dsp/dsp_processor.py:147).
//
// What the other forms do there: the wide kernel walks the chunk 2 .. 8 times (eight outputs of every 256-sample column per pass,
// the column aggregates recomputed in each), decimation 2 runs the per-sample recurrences in fp64 (176 multiply-adds per sample,
// the input read three times).  Here a column is 256 samples with ALL of its nout = 256 / q outputs:
//   y[256 c + s] = sum_j h[j - s] u[256 c + j]  +  2 Re sum_i ( p_i^s F_i(c) + p_i^(256 - s) G_i(c + 1) )          s = b q
//   (h: the zero-phase decimator's impulse response cut to the column; F / G: causal / anticausal modal states at the column's
//   start / end, tables.h: DecimTables pole / ac / aa; numpy model: tools/model_sq.py, 1e-15 .. 7e-14 from scipy)
// * the local part is a CONVOLUTION: row b of the local operator is row 0 shifted by b q samples, so the digit planes of all rows
//   are windows of one master sequence per digit (1 KB; with the NCO: h[d] e^(i theta d), the output rotated by e^(i theta t) in
//   fp64 afterwards) -- A fragments are read from LDS at 16-byte aligned offsets of it (q < 8: 16 / 2q copies shifted by 2 q bytes),
//   whatever the number of outputs: 40 KB at decimation 2 where row planes would take 640 KB;
// * the column aggregates (two 16-row tiles, as in the wide kernel) are computed ONCE per column, scanned over the chunk's columns
//   (wave scans, 128 columns), and every output adds its share of its column's two states.
// Arithmetic: exact integer slicing (x = 256 hi + lo' + offset), S = 5 balanced base-256 digits of the weights, int32 sums, one fp64
// recombination per value -- as kernels_i8w.hip; the constant the slices leave out and the normalisation
// bias come back through ybias, calibrated with this kernel's own response to a zero / a unit chunk (api.hip: setup_sq).
//
// One persistent 512-thread workgroup per CU keeps a row's planes in LDS (80 KB aggregates + 10 .. 40 KB master) and walks over
// chunks; wave w <-> tile w (16 columns); a tile's byte slices stay in registers from the aggregate phase to the local phase.
//   per chunk:  E  edge operators (head / tail, 28 samples each)      A1  aggregate tiles -> LDS      | barrier
//               P2 column scans, 16 chains over the waves (causal, barrier, anticausal)                | barrier
//               A2 local tiles (nout / 8 of them), states' shares, ybias, store                        | barrier
#include "kernels_common.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));

namespace {

template <int DT> struct SqEnc;        // slices per component, what each slice XORs with, what they leave out of the value
template <> struct SqEnc<SDRT_S16> { static constexpr unsigned xlo = 0x80808080u, xhi = 0u; static constexpr double off = 128.0; };
template <> struct SqEnc<SDRT_U16> { static constexpr unsigned xlo = 0x80808080u, xhi = 0x80808080u; static constexpr double off = 32896.0; };

constexpr int SQ_S = SDRT_I8_DIGITS, SQ_LC = SDRT_SQ_LC, SQ_KS = SQ_LC / 32, SQ_MAXCOL = 128, SQ_WV = 8;

struct SqLds {          // byte offsets
    int agg, master, fc, gs, f0, gu, psi, samp, total;
};
__host__ __device__ inline SqLds sq_lds(int ncp) {
    SqLds l;
    int p = 0;
    l.agg = p; p += 2 * SQ_S * SQ_KS * 64 * 16;
    l.master = p; p += 2 * SQ_S * ncp * SDRT_SQ_MASTER;
    l.fc = p; p += (SQ_MAXCOL + 1) * 8 * 16;          // causal state at the start of column c (c = NC: the chunk's end), [c][mode][stream]
    l.gs = p; p += (SQ_MAXCOL + 1) * 8 * 16;          // anticausal state at the start of column c (inclusive), c = NC: beyond the end
    l.f0 = p; p += 8 * 16;
    l.gu = p; p += 8 * 16;
    l.psi = p; p += SQ_MAXCOL * 16;                   // exp(i theta 256 c)
    l.samp = p; p += SQ_LC * 16;                      // exp(i theta j), j < 256
    l.total = p;
    return l;
}

template <int DT, int QS>
__global__ __launch_bounds__(64 * SQ_WV, 2) void chain_sq_kernel(const SqArgs a) {
    constexpr int q = 1 << QS, NOUT = SQ_LC / q, NRT = 2 * NOUT / 16, NCP = 2 * q >= 16 ? 1 : 16 / (2 * q), S = SQ_S, KS = SQ_KS;
    typedef typename Enc<DT>::raw raw_t;
    extern __shared__ __align__(16) unsigned char smem[];
    const SqLds lo = sq_lds(NCP);
    const v4i *aggL = (const v4i *) (smem + lo.agg);
    const v4i *masterL = (const v4i *) (smem + lo.master);
    cd *FcL = (cd *) (smem + lo.fc), *GsL = (cd *) (smem + lo.gs), *F0L = (cd *) (smem + lo.f0), *GuL = (cd *) (smem + lo.gu);
    cd *psiL = (cd *) (smem + lo.psi), *sampL = (cd *) (smem + lo.samp);
    const SqTables *__restrict__ tab = a.tab;
    const int n = a.n, m = a.m, NC = n / SQ_LC, ntiles = NC / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, nb = lane & 15;
    const bool swap = a.swap != 0, nco = a.nco_samp != nullptr;
    // which row, which chunks: workgroup b serves row b % rows, chunks (b / rows), + wpr, ...
    const int row = (int) (blockIdx.x % (unsigned) a.rows);
    const size_t wpr = gridDim.x / (unsigned) a.rows, first = blockIdx.x / (unsigned) a.rows;
    if (first >= wpr) return;

    // ---- the row's planes and phasors into LDS, once ------------------------------------------------------------------------------
    {
        const v4i *src = (const v4i *) (a.agg + (size_t) row * (2 * S * KS * 64 * 16));
        v4i *dst = (v4i *) (smem + lo.agg);
        for (int i = tid; i < 2 * S * KS * 64; i += 64 * SQ_WV) dst[i] = src[i];
        const v4i *srm = (const v4i *) (a.master + (size_t) row * (2 * S * NCP * SDRT_SQ_MASTER));
        v4i *dsm = (v4i *) (smem + lo.master);
        for (int i = tid; i < 2 * S * NCP * SDRT_SQ_MASTER / 16; i += 64 * SQ_WV) dsm[i] = srm[i];
        for (int i = tid; i < NC; i += 64 * SQ_WV) psiL[i] = nco ? ((const cd *) a.nco_col)[(size_t) row * NC + i] : mk(1.0, 0.0);
        for (int i = tid; i < SQ_LC; i += 64 * SQ_WV) sampL[i] = nco ? ((const cd *) a.nco_samp)[(size_t) row * SQ_LC + i] : mk(1.0, 0.0);
    }
    __syncthreads();
    const double cagg = a.comb_agg * 256.0, cloc = a.comb_loc * 256.0;       // accumulator index a holds level a + 1
    const double off = SqEnc<DT>::off;

    for (size_t chunk = first; chunk < a.n_chunks; chunk += wpr) {
        const unsigned char *chunk_raw = (const unsigned char *) a.raw + chunk * (size_t) n * 4;
        // ---- E: edge operators.  F_0,i = sum_l head[i][l] u[l], the sample part of G_end,i = sum_l tailU[i][l] u[n - 28 + l], per
        // real stream; u = (x - off) exp(i theta t): the same x - off the slices carry (ybias returns what is left out)
        {
            const bool tail = tid >= 256;
            const int o = (tid & 255) >> 5, l = tid & 31, mode = o >> 1, st = o & 1;
            cd f = mk(0.0, 0.0);
            if (l <= SDRT_EDGE) {
                const long idx = tail ? (long) n - (SDRT_EDGE + 1) + l : l;
                const raw_t *rp = (const raw_t *) chunk_raw + 2 * idx;
                const double xr = decode<DT>(rp[0], swap) - off, xi = decode<DT>(rp[1], swap) - off;
                const cd ph = cmul(sampL[idx & (SQ_LC - 1)], psiL[idx / SQ_LC]);
                const double us = st ? xr * ph.im + xi * ph.re : xr * ph.re - xi * ph.im;
                const double *tb = tail ? tab->tailU[mode][l] : tab->head[mode][l];
                f = mk(tb[0] * us, tb[1] * us);
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) {
                f.re += __shfl_xor(f.re, d);
                f.im += __shfl_xor(f.im, d);
            }
            if (l == 0) (tail ? GuL : F0L)[o] = f;
        }
        // ---- A1: the wave's tile: its 16 columns' byte slices (kept for A2) and their aggregates -------------------------------------
        v4i bhi[KS], blo[KS];
        const int tile = wave, col = tile * 16 + nb;
        const bool have = tile < ntiles;
        if (have) {
            const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
            uint32_t cur[KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                load_words<32>(chunk_raw + ((size_t) col * SQ_LC + 32 * ks + 8 * g) * 4, cur[ks]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    blo[ks][w] = (int) (__builtin_amdgcn_perm(cur[ks][2 * w + 1], cur[ks][2 * w], sel_lo) ^ SqEnc<DT>::xlo);
                    bhi[ks][w] = (int) (__builtin_amdgcn_perm(cur[ks][2 * w + 1], cur[ks][2 * w], sel_hi) ^ SqEnc<DT>::xhi);
                }
            const cd ps = psiL[col];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                v4i acc[S];
#pragma unroll
                for (int d = 0; d < S; ++d) acc[d] = (v4i) {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int d = 0; d < S; ++d) {
                        const v4i A = aggL[((rt * S + d) * KS + ks) * 64 + lane];
                        acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[ks], acc[d], 0, 0, 0);                  // digit d x high slice: level d + 1
                        if (d >= 1) acc[d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo[ks], acc[d - 1], 0, 0, 0);  // x low slice: level d
                    }
                double val[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double sum = (double) acc[S - 1][v];
#pragma unroll
                    for (int d = S - 2; d >= 0; --d) sum = fma(sum, 256.0, (double) acc[d][v]);
                    val[v] = sum * cagg;
                }
                // lane (g, nb): mode g of column nb; (val0, val1) the state of the re stream, (val2, val3) of the im stream.  The column's
                // common phasor rotates the complex DATA: re' = c re - s im, im' = s re + c im (stream by stream, both state components)
                const cd sre = mk(fma(ps.re, val[0], -(ps.im * val[2])), fma(ps.re, val[1], -(ps.im * val[3])));
                const cd sim = mk(fma(ps.im, val[0], ps.re * val[2]), fma(ps.im, val[1], ps.re * val[3]));
                if (rt == 0) { FcL[(col + 1) * 8 + g * 2 + 0] = sre; FcL[(col + 1) * 8 + g * 2 + 1] = sim; }
                else { GsL[col * 8 + g * 2 + 0] = sre; GsL[col * 8 + g * 2 + 1] = sim; }
            }
        }
        __syncthreads();
        // ---- P2: the states at the column boundaries.  Chain o = mode * 2 + stream; wave w scans chains 2 w, 2 w + 1 ----------------------
        {
            const cd mu = mk(tab->mu[wave & 3][0], tab->mu[wave & 3][1]);          // chains 2 w, 2 w + 1 share mode w (w < 4 only)
            if (wave < 4) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int o = wave * 2 + cc;
                    cd carry = F0L[o];
                    if (lane == 0) FcL[o] = carry;
                    for (int t0 = 0; t0 < NC; t0 += 64) {
                        const int c = t0 + lane, valid = NC - t0 < 64 ? NC - t0 : 64;
                        cd x = lane < valid ? FcL[(c + 1) * 8 + o] : mk(0.0, 0.0);
                        if (lane == 0) x = cfma(mu, carry, x);
                        x = wave_scan_geometric(x, mu, lane);
                        if (lane < valid) FcL[(c + 1) * 8 + o] = x;
                        carry = mk(__shfl(x.re, valid - 1), __shfl(x.im, valid - 1));
                    }
                }
            }
        }
        __syncthreads();
        if (wave < 4) {
            const cd mu = mk(tab->mu[wave][0], tab->mu[wave][1]);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int o = wave * 2 + cc, mode = wave, st = cc;
                cd G = GuL[o];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const cd Fj = FcL[NC * 8 + j * 2 + st];
                    G = cfma(mk(tab->tailF[mode][j][0], tab->tailF[mode][j][1]), Fj, G);
                    G = cfma(mk(tab->tailFc[mode][j][0], tab->tailFc[mode][j][1]), mk(Fj.re, -Fj.im), G);
                }
                if (lane == 0) GsL[NC * 8 + o] = G;
                cd carry = G;
                for (int t0 = 0; t0 < NC; t0 += 64) {
                    const int c = NC - 1 - (t0 + lane), valid = NC - t0 < 64 ? NC - t0 : 64;       // lanes walk the columns from the right
                    cd x = lane < valid ? GsL[c * 8 + o] : mk(0.0, 0.0);
                    if (lane == 0) x = cfma(mu, carry, x);
                    x = wave_scan_geometric(x, mu, lane);
                    if (lane < valid) GsL[c * 8 + o] = x;
                    carry = mk(__shfl(x.re, valid - 1), __shfl(x.im, valid - 1));
                }
            }
        }
        __syncthreads();
        // ---- A2: the local part of every output of the tile's columns, the states' shares, ybias, store --------------------------------------
        if (have) {
            // this lane's outputs belong to column nb: its states (mode i, stream): start of the column (causal), start of the next (anticausal)
            cd Fs[8], Gn[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { Fs[i] = FcL[col * 8 + i]; Gn[i] = GsL[(col + 1) * 8 + i]; }
            const cd ps = psiL[col];
            const int r = lane & 15, part = r & 1, bl = r >> 1;                 // A operand: row r of the row tile = output bl, part
            const size_t item = (size_t) row * a.n_chunks + chunk;
            double *yd = a.ydec + item * (size_t) (a.pair_out ? m / 2 : m) * 2;
            const cd *yb = a.ybias ? (const cd *) a.ybias + (size_t) row * m : nullptr;
            for (int rt = 0; rt < NRT; ++rt) {
                const int s = (rt * 8 + bl) << QS;                                // sample of the column the A row's output sits on
                // 16-byte units: ((part S + d) NCP + copy) 64 + (64 ks + 16 g - 16 (s >> 3) + 512) / 16
                const int abase = (part * S * NCP + ((s & 7) >> QS)) * (SDRT_SQ_MASTER / 16) + g - (s >> 3) + 32;
                v4i acc[S];
#pragma unroll
                for (int d = 0; d < S; ++d) acc[d] = (v4i) {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int d = 0; d < S; ++d) {
                        const v4i A = masterL[abase + d * NCP * (SDRT_SQ_MASTER / 16) + 4 * ks];
                        acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[ks], acc[d], 0, 0, 0);
                        if (d >= 1) acc[d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo[ks], acc[d - 1], 0, 0, 0);
                    }
                double val[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double sum = (double) acc[S - 1][v];
#pragma unroll
                    for (int d = S - 2; d >= 0; --d) sum = fma(sum, 256.0, (double) acc[d][v]);
                    val[v] = sum * cloc;
                }
                // result rows 4 g + v of column nb: outputs b0 = 8 rt + 2 g (val0 + i val1) and b0 + 1 (val2 + i val3)
                cd y2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int b = rt * 8 + 2 * g + h, so = b << QS;
                    const cd rot = cmul(ps, sampL[so]);                             // exp(i theta (256 c + s)): the output's own phasor
                    cd y = mk(fma(rot.re, val[2 * h], -(rot.im * val[2 * h + 1])), fma(rot.re, val[2 * h + 1], rot.im * val[2 * h]));
                    const cd *sh = (const cd *) a.share + (size_t) b * 8;          // 2 p_i^s (i < 4), 2 p_i^(256 - s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const cd pf = sh[i], pg = sh[4 + i];
                        y.re = fma(pf.re, Fs[2 * i].re, fma(-pf.im, Fs[2 * i].im, fma(pg.re, Gn[2 * i].re, fma(-pg.im, Gn[2 * i].im, y.re))));
                        y.im = fma(pf.re, Fs[2 * i + 1].re, fma(-pf.im, Fs[2 * i + 1].im, fma(pg.re, Gn[2 * i + 1].re, fma(-pg.im, Gn[2 * i + 1].im, y.im))));
                    }
                    const int k = col * NOUT + b;
                    if (yb) { y.re += yb[k].re; y.im += yb[k].im; }
                    else { y.re += 0.0; y.im += 0.0; }           // -0.0 -> +0.0 (angle(-0) would be pi)
                    y2[h] = y;
                }
                const int k0 = col * NOUT + rt * 8 + 2 * g;
                if (a.pair_out) {
                    // FM without a debug buffer: the product the discriminator takes the angle of (demodulation.py:25-32)
                    *(double2 *) (yd + (size_t) k0) = make_double2(y2[0].re * y2[1].re + y2[0].im * y2[1].im, y2[0].im * y2[1].re - y2[0].re * y2[1].im);
                } else {
                    *(double2 *) (yd + 2 * (size_t) k0) = make_double2(y2[0].re, y2[0].im);
                    *(double2 *) (yd + 2 * (size_t) k0 + 2) = make_double2(y2[1].re, y2[1].im);
                }
            }
        }
        __syncthreads();            // the state arrays are free for the next chunk
    }
}

template <int DT>
hipError_t launch_sq_q(const SqArgs &a, int q, dim3 grid, hipStream_t s) {
    auto go = [&](auto qc) -> hipError_t {
        constexpr int QS = decltype(qc)::value, qq = 1 << QS, NCP = 2 * qq >= 16 ? 1 : 16 / (2 * qq);
        const size_t lds = (size_t) sq_lds(NCP).total;
        static thread_local int done_dev = -1;
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (done_dev != dev) {
            hipError_t e = hipFuncSetAttribute((const void *) chain_sq_kernel<DT, QS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            done_dev = dev;
        }
        hipLaunchKernelGGL((chain_sq_kernel<DT, QS>), grid, dim3(64 * SQ_WV), lds, s, a);
        return hipGetLastError();
    };
    switch (q) {
        case 2: return go(std::integral_constant<int, 1>{});
        case 4: return go(std::integral_constant<int, 2>{});
        case 8: return go(std::integral_constant<int, 3>{});
        case 16: return go(std::integral_constant<int, 4>{});
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool chain_sq_supported(int q, int dt, int n) {
    const int nc = n / SQ_LC;
    return (q == 2 || q == 4 || q == 8 || q == 16) && (dt == SDRT_S16 || dt == SDRT_U16) && n % (16 * SQ_LC) == 0 && nc >= 16 && nc <= SQ_MAXCOL;
}

hipError_t launch_chain_sq(const SqArgs &a, int q, int dt, int n_cus, hipStream_t s) {
    if (!a.n_chunks) return hipSuccess;
    // one workgroup per CU: a multiple of the rows, every workgroup keeps one row's planes
    size_t wpr = (size_t) n_cus / (size_t) a.rows;
    if (wpr < 1) wpr = 1;
    if (wpr > a.n_chunks) wpr = a.n_chunks;
    const dim3 grid((unsigned) (wpr * (size_t) a.rows));
    if (dt == SDRT_S16) return launch_sq_q<SDRT_S16>(a, q, grid, s);
    if (dt == SDRT_U16) return launch_sq_q<SDRT_U16>(a, q, grid, s);
    return hipErrorInvalidValue;
}
