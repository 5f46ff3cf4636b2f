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
// One persistent 256-thread workgroup per CU keeps a row's planes in LDS (80 KB aggregates + 10 .. 40 KB master) and walks over
// chunks; ONE wave per SIMD with the whole register file (512 registers: the slices of its two tiles, 128 of them, stay in
// registers from the aggregate phase to the local phase; with two waves per SIMD at 256 registers the allocator spilled addresses
// and reloaded them between dependent LDS reads: profiles/r05_experiments.txt section 7); wave w <-> tiles w, w + 4 (16 columns each).
//   per chunk:  E  edge operators (head / tail, 28 samples each)
//               A1 aggregate tiles, scanned over the tile's 16 columns in the lanes of a DPP row -> LDS         | barrier
//               P2 tile carries: eight chains in eight lanes of one wave, the tiles one after the other       | barrier
//               A2 local tiles (nout / 8 of them), states' shares, ybias, store                               | barrier
#include "kernels_common.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));

#ifndef SDRT_SQ_WAVES
#define SDRT_SQ_WAVES 8          // waves per workgroup: 8 (two per SIMD, a tile each) or 4 (one per SIMD with 512 registers, two tiles each)
#endif

namespace {

template <int DT> struct SqEnc;        // slices per component, what each slice XORs with, what they leave out of the value
template <> struct SqEnc<SDRT_S16> { static constexpr unsigned xlo = 0x80808080u, xhi = 0u; static constexpr double off = 128.0; };
template <> struct SqEnc<SDRT_U16> { static constexpr unsigned xlo = 0x80808080u, xhi = 0x80808080u; static constexpr double off = 32896.0; };

constexpr int SQ_S = SDRT_I8_DIGITS, SQ_LC = SDRT_SQ_LC, SQ_KS = SQ_LC / 32, SQ_MAXCOL = 128, SQ_WV = SDRT_SQ_WAVES, SQ_TPW = 8 / SQ_WV, SQ_MAXT = SQ_WV * SQ_TPW;

struct SqLds {          // byte offsets
    int agg, master, fx, gx, tf, tg, cf, cg, f0, gu, psi, samp, cst, share, total;
};
__host__ __device__ inline SqLds sq_lds(int ncp, int nout) {
    const int q = SQ_LC / nout;
    SqLds l;
    int p = 0;
    l.agg = p; p += 2 * SQ_S * SQ_KS * 64 * 16;
    l.master = p; p += SDRT_SQ_MASTER_BYTES(q, SQ_S, ncp);
    l.fx = p; p += SQ_MAXCOL * 8 * 16;                // causal state at the start of column c when its TILE starts from zero, [c][mode][stream]
    l.gx = p; p += SQ_MAXCOL * 8 * 16;                // anticausal state at the start of column c + 1 when the tile's right neighbour is zero
    l.tf = p; p += SQ_MAXT * 8 * 16;                  // the tiles' totals ...
    l.tg = p; p += SQ_MAXT * 8 * 16;
    l.cf = p; p += (SQ_MAXT + 1) * 8 * 16;              // ... and what enters each tile from the left / from the right
    l.cg = p; p += (SQ_MAXT + 1) * 8 * 16;
    l.f0 = p; p += 8 * 16;
    l.gu = p; p += 8 * 16;
    l.psi = p; p += nout <= 64 ? SQ_MAXCOL * 16 : 0;  // exp(i theta 256 c)       (decimation 2: both phasor tables are read from memory,
    l.samp = p; p += nout <= 64 ? SQ_LC * 16 : 0;     // exp(i theta j), j < 256    the LDS is full)
    l.cst = p; p += 4 * 13 * 16;                      // per mode: mu^(1, 2, 4, 8, 16), tailF[mode][0..3], tailFc[mode][0..3]
    l.share = p; p += nout <= 64 ? nout * 8 * 16 : 0; // the states' shares per output (decimation 2: 16 KB, read from memory instead)
    l.total = p;
    return l;
}

// mu^k, k < 16, from the squarings mp[0..3] = mu^(1, 2, 4, 8)
__device__ __forceinline__ cd mu_pow(const cd *mp, int k) {
    cd r = k & 1 ? mp[0] : mk(1.0, 0.0);
    if (k & 2) r = cmul(r, mp[1]);
    if (k & 4) r = cmul(r, mp[2]);
    if (k & 8) r = cmul(r, mp[3]);
    return r;
}

template <int DT, int QS>
__global__ __launch_bounds__(64 * SQ_WV, SQ_WV / 4) void chain_sq_kernel(const SqArgs a) {
    constexpr int q = 1 << QS, NOUT = SQ_LC / q, NRT = 2 * NOUT / 16, NCP = 2 * q >= 16 ? 1 : 16 / (2 * q), S = SQ_S, KS = SQ_KS;
    constexpr bool SHL = NOUT <= 64;                  // the share table fits the LDS
    typedef typename Enc<DT>::raw raw_t;
    extern __shared__ __align__(16) unsigned char smem[];
    const SqLds lo = sq_lds(NCP, NOUT);
    const v4i *aggL = (const v4i *) (smem + lo.agg);
    const v4i *masterL = (const v4i *) (smem + lo.master);
    cd *FxL = (cd *) (smem + lo.fx), *GxL = (cd *) (smem + lo.gx), *TFL = (cd *) (smem + lo.tf), *TGL = (cd *) (smem + lo.tg);
    cd *CFL = (cd *) (smem + lo.cf), *CGL = (cd *) (smem + lo.cg), *F0L = (cd *) (smem + lo.f0), *GuL = (cd *) (smem + lo.gu);
    cd *psiL = (cd *) (smem + lo.psi), *sampL = (cd *) (smem + lo.samp), *cstL = (cd *) (smem + lo.cst), *shareL = (cd *) (smem + lo.share);
    // exp(i theta 256 c) / exp(i theta j): from the LDS, or (decimation 2) from the row's tables in memory
    const cd *psiG = (const cd *) a.nco_col + (size_t) blockIdx.x % (unsigned) a.rows * (size_t) (a.n / SQ_LC);
    const cd *sampG = (const cd *) a.nco_samp + (size_t) blockIdx.x % (unsigned) a.rows * SQ_LC;
    auto psi_at = [&](int c) -> cd { if constexpr (SHL) return psiL[c]; else return a.nco_samp ? psiG[c] : mk(1.0, 0.0); };
    auto samp_at = [&](int j) -> cd { if constexpr (SHL) return sampL[j]; else return a.nco_samp ? sampG[j] : mk(1.0, 0.0); };
    const SqTables *__restrict__ tab = a.tab;
    const int n = a.n, m = a.m, NC = n / SQ_LC, ntiles = NC / 16;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const bool swap = a.swap != 0, nco = a.nco_samp != nullptr;
    int lane = lane0, g = lane0 >> 4, nb = lane0 & 15;
    // which row, which chunks: workgroup b serves row b % rows, chunks (b / rows), + wpr, ...
    const int row = (int) (blockIdx.x % (unsigned) a.rows);
    const size_t wpr = gridDim.x / (unsigned) a.rows, first = blockIdx.x / (unsigned) a.rows;
    if (first >= wpr) return;

    // ---- the row's planes, phasors and constants into LDS, once ---------------------------------------------------------------------
    {
        const v4i *src = (const v4i *) (a.agg + (size_t) row * (2 * S * KS * 64 * 16));
        v4i *dst = (v4i *) (smem + lo.agg);
        for (int i = tid; i < 2 * S * KS * 64; i += 64 * SQ_WV) dst[i] = src[i];
        const v4i *srm = (const v4i *) (a.master + (size_t) row * SDRT_SQ_MASTER_BYTES(q, S, NCP));
        v4i *dsm = (v4i *) (smem + lo.master);
        for (int i = tid; i < SDRT_SQ_MASTER_BYTES(q, S, NCP) / 16; i += 64 * SQ_WV) dsm[i] = srm[i];
        if constexpr (SHL) {
            for (int i = tid; i < NC; i += 64 * SQ_WV) psiL[i] = nco ? ((const cd *) a.nco_col)[(size_t) row * NC + i] : mk(1.0, 0.0);
            for (int i = tid; i < SQ_LC; i += 64 * SQ_WV) sampL[i] = nco ? ((const cd *) a.nco_samp)[(size_t) row * SQ_LC + i] : mk(1.0, 0.0);
        }
        if constexpr (SHL) for (int i = tid; i < NOUT * 8; i += 64 * SQ_WV) shareL[i] = ((const cd *) a.share)[i];
        if (tid < 4) {
            cd *cs = cstL + tid * 13;
            cd v = mk(tab->mu[tid][0], tab->mu[tid][1]);
            for (int k = 0; k < 5; ++k) { cs[k] = v; v = cmul(v, v); }
            for (int j = 0; j < 4; ++j) { cs[5 + j] = mk(tab->tailF[tid][j][0], tab->tailF[tid][j][1]); cs[9 + j] = mk(tab->tailFc[tid][j][0], tab->tailFc[tid][j][1]); }
        }
    }
    __syncthreads();
    const double cagg = a.comb_agg * 256.0, cloc = a.comb_loc * 256.0;       // accumulator index a holds level a + 1
    const double off = SqEnc<DT>::off;
    // this thread's edge coefficient (the same for every chunk)
    const int e_o = (tid >> 5) & 7, e_l = tid & 31;
    cd e_head = mk(0.0, 0.0), e_tailc = mk(0.0, 0.0);
    if (e_l <= SDRT_EDGE) {
        e_head = mk(tab->head[e_o >> 1][e_l][0], tab->head[e_o >> 1][e_l][1]);
        e_tailc = mk(tab->tailU[e_o >> 1][e_l][0], tab->tailU[e_o >> 1][e_l][1]);
    }
    unsigned touched = 0;                                                     // (the next chunk's lines pulled towards the L2: see A2)

    for (size_t chunk = first; chunk < a.n_chunks; chunk += wpr) {
        const unsigned char *chunk_raw = (const unsigned char *) a.raw + chunk * (size_t) n * 4;
        // (the lane's coordinates made opaque per chunk: everything addressed through them is then computed where it is used.  Hoisted
        // out of this loop -- sixteen load addresses, the LDS addresses of every phase -- those values outlive the whole iteration, the
        // allocator spills them, and every reload waits for all memory traffic in flight, one load address at a time)
        asm volatile("" : "+v"(lane), "+v"(g), "+v"(nb));
        // ---- E: edge operators.  F_0,i = sum_l head[i][l] u[l], the sample part of G_end,i = sum_l tailU[i][l] u[n - 28 + l], per
        // real stream; u = (x - off) exp(i theta t): the same x - off the slices carry (ybias returns what is left out)
        {
            const int st = e_o & 1;
            cd f = mk(0.0, 0.0), gq = mk(0.0, 0.0);
            if (e_l <= SDRT_EDGE && tid < 256) {
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const long idx = tl ? (long) n - (SDRT_EDGE + 1) + e_l : e_l;
                    const raw_t *rp = (const raw_t *) chunk_raw + 2 * idx;
                    const double xr = decode<DT>(rp[0], swap) - off, xi = decode<DT>(rp[1], swap) - off;
                    const cd ph = cmul(samp_at((int) (idx & (SQ_LC - 1))), psi_at((int) (idx / SQ_LC)));
                    const double us = st ? xr * ph.im + xi * ph.re : xr * ph.re - xi * ph.im;
                    if (tl) gq = mk(e_tailc.re * us, e_tailc.im * us); else f = mk(e_head.re * us, e_head.im * us);
                }
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) {
                f.re += __shfl_xor(f.re, d);
                f.im += __shfl_xor(f.im, d);
                gq.re += __shfl_xor(gq.re, d);
                gq.im += __shfl_xor(gq.im, d);
            }
            if (e_l == 0 && tid < 256) { F0L[e_o] = f; GuL[e_o] = gq; }
        }
        // ---- A1: the wave's tile: its 16 columns' byte slices (kept for A2), their aggregates, the scans over the tile's columns ------------
        v4i bhi[SQ_TPW][KS], blo[SQ_TPW][KS];
        uint32_t cur[SQ_TPW][KS][8];
#pragma unroll
        for (int j = 0; j < SQ_TPW; ++j)
            if (wave + SQ_WV * j < ntiles)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    load_words<32>(chunk_raw + ((size_t) ((wave + SQ_WV * j) * 16 + nb) * SQ_LC + 32 * ks + 8 * g) * 4, cur[j][ks]);
        // (every load of both tiles in flight before anything waits for one: left to itself the compiler sinks each k-step's loads to
        // its matrix instructions -- eight memory round trips one after the other)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < SQ_TPW; ++j) {
        const int tile = wave + SQ_WV * j, col = tile * 16 + nb;
        const bool have = tile < ntiles;
        if (have) {
            const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    blo[j][ks][w] = (int) (__builtin_amdgcn_perm(cur[j][ks][2 * w + 1], cur[j][ks][2 * w], sel_lo) ^ SqEnc<DT>::xlo);
                    bhi[j][ks][w] = (int) (__builtin_amdgcn_perm(cur[j][ks][2 * w + 1], cur[j][ks][2 * w], sel_hi) ^ SqEnc<DT>::xhi);
                }
            __builtin_amdgcn_sched_barrier(0);
            const cd ps = psi_at(col);
            const cd *cstG = cstL + g * 13;                                   // mode g = this lane's aggregate rows
            const cd m1 = cstG[0], m2 = cstG[1], m4 = cstG[2], m8 = cstG[3];
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
                        acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[j][ks], acc[d], 0, 0, 0);                  // digit d x high slice: level d + 1
                        if (d >= 1) acc[d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo[j][ks], acc[d - 1], 0, 0, 0);  // x low slice: level d
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
                cd x0 = mk(fma(ps.re, val[0], -(ps.im * val[2])), fma(ps.re, val[1], -(ps.im * val[3])));
                cd x1 = mk(fma(ps.im, val[0], ps.re * val[2]), fma(ps.im, val[1], ps.re * val[3]));
                if (rt == 0) {
                    // causal: inclusive scan over the tile's 16 columns (the lanes of a DPP row), F(c + 1) = mu F(c) + A_c from zero
                    x0 = cfma(m1, dpp_row<0x111>(x0), x0); x1 = cfma(m1, dpp_row<0x111>(x1), x1);
                    x0 = cfma(m2, dpp_row<0x112>(x0), x0); x1 = cfma(m2, dpp_row<0x112>(x1), x1);
                    x0 = cfma(m4, dpp_row<0x114>(x0), x0); x1 = cfma(m4, dpp_row<0x114>(x1), x1);
                    x0 = cfma(m8, dpp_row<0x118>(x0), x0); x1 = cfma(m8, dpp_row<0x118>(x1), x1);
                    if (nb == 15) { TFL[tile * 8 + g * 2] = x0; TFL[tile * 8 + g * 2 + 1] = x1; }
                    FxL[col * 8 + g * 2] = dpp_row<0x111>(x0);           // exclusive: the state at the column's own start
                    FxL[col * 8 + g * 2 + 1] = dpp_row<0x111>(x1);
                } else {
                    // anticausal: G(c) = B_c + mu G(c + 1), from zero beyond the tile
                    x0 = cfma(m1, dpp_row<0x101>(x0), x0); x1 = cfma(m1, dpp_row<0x101>(x1), x1);
                    x0 = cfma(m2, dpp_row<0x102>(x0), x0); x1 = cfma(m2, dpp_row<0x102>(x1), x1);
                    x0 = cfma(m4, dpp_row<0x104>(x0), x0); x1 = cfma(m4, dpp_row<0x104>(x1), x1);
                    x0 = cfma(m8, dpp_row<0x108>(x0), x0); x1 = cfma(m8, dpp_row<0x108>(x1), x1);
                    if (nb == 0) { TGL[tile * 8 + g * 2] = x0; TGL[tile * 8 + g * 2 + 1] = x1; }
                    GxL[col * 8 + g * 2] = dpp_row<0x101>(x0);           // the state at the start of the NEXT column
                    GxL[col * 8 + g * 2 + 1] = dpp_row<0x101>(x1);
                }
            }
        }
        }
        __syncthreads();
        // ---- P2: what enters every tile from the left / right: eight chains (mode, stream) in eight lanes, the tiles one after the other
        if (wave == 0) {
            const int o = lane & 7, mode = o >> 1, st = o & 1;
            const cd *cs = cstL + mode * 13;
            const cd mu16 = cs[4];
            cd c = F0L[o];
            for (int t = 0; t < ntiles; ++t) {
                if (lane < 8) CFL[t * 8 + o] = c;
                c = cfma(mu16, c, TFL[t * 8 + o]);
            }
            // G beyond the chunk's end: the tail samples' share + what the final causal states of all four modes (same stream) add
            cd G = GuL[o];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cd Fj = mk(__shfl(c.re, j * 2 + st), __shfl(c.im, j * 2 + st));
                G = cfma(cs[5 + j], Fj, G);
                G = cfma(cs[9 + j], mk(Fj.re, -Fj.im), G);
            }
            for (int t = ntiles - 1; t >= 0; --t) {
                if (lane < 8) CGL[t * 8 + o] = G;                       // the state at the start of column 16 (t + 1)
                G = cfma(mu16, G, TGL[t * 8 + o]);
            }
        }
        __syncthreads();
        // ---- A2: the local part of every output of the tile's columns, the states' shares, ybias, store --------------------------------------
#pragma unroll
        for (int j = 0; j < SQ_TPW; ++j) {
        const int tile = wave + SQ_WV * j, col = tile * 16 + nb;
        if (tile < ntiles) {
            // (while the matrix cores work: one word of every 128-byte line of this wave's tile of the NEXT chunk, so that the next
            // iteration's loads find it in the L2 -- the workgroup is alone on its CU and nothing else covers that round trip)
            if (chunk + wpr < a.n_chunks) {
                const unsigned char *nx = chunk_raw + wpr * (size_t) n * 4 + (size_t) tile * 16 * SQ_LC * 4;
                touched ^= *(const unsigned *) (nx + (size_t) lane * 128) ^ *(const unsigned *) (nx + (size_t) (lane + 64) * 128);
            }
            // this lane's outputs belong to column nb: the states at its start (causal) and at the start of the next one (anticausal),
            // every (mode, stream): the tile-local value + the tile's carry through nb / 15 - nb columns
            cd Fs[8], Gn[8];
#pragma unroll
            for (int md = 0; md < 4; ++md) {
                const cd pf = mu_pow(cstL + md * 13, nb), pg = mu_pow(cstL + md * 13, 15 - nb);
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int i = md * 2 + st;
                    Fs[i] = cfma(pf, CFL[tile * 8 + i], FxL[col * 8 + i]);
                    Gn[i] = cfma(pg, CGL[tile * 8 + i], GxL[col * 8 + i]);
                }
            }
            const cd ps = psi_at(col);
            const int r = lane & 15, part = r & 1, bl = r >> 1;                 // A operand: row r of the row tile = output bl, part
            const size_t item = (size_t) row * a.n_chunks + chunk;
            double *yd = a.ydec + item * (size_t) (a.pair_out ? m / 2 : m) * 2;
            const cd *yb = a.ybias ? (const cd *) a.ybias + (size_t) row * m : nullptr;
#pragma unroll 1
            for (int rt = 0; rt < NRT; ++rt) {
                const int s = (rt * 8 + bl) << QS;                                // sample of the column the A row's output sits on
                // 16-byte units (tables.h: SDRT_SQ_CS / _PS): part PS + (d NCP + copy) CS + (64 ks + 16 g - 16 (s >> 3) + 512) / 16
                const int abase = part * SDRT_SQ_PS(q, S, NCP) + ((s & 7) >> QS) * SDRT_SQ_CS(q) + g - (s >> 3) + 32;
                v4i acc[S];
#pragma unroll
                for (int d = 0; d < S; ++d) acc[d] = (v4i) {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int d = 0; d < S; ++d) {
                        const v4i A = masterL[abase + d * NCP * SDRT_SQ_CS(q) + 4 * ks];
                        acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[j][ks], acc[d], 0, 0, 0);
                        if (d >= 1) acc[d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo[j][ks], acc[d - 1], 0, 0, 0);
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
                    const cd rot = cmul(ps, samp_at(so));                             // exp(i theta (256 c + s)): the output's own phasor
                    cd y = mk(fma(rot.re, val[2 * h], -(rot.im * val[2 * h + 1])), fma(rot.re, val[2 * h + 1], rot.im * val[2 * h]));
                    const cd *sh = (SHL ? (const cd *) shareL : (const cd *) a.share) + (size_t) b * 8;          // 2 p_i^s (i < 4), 2 p_i^(256 - s)
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
        }
        __syncthreads();            // the state arrays are free for the next chunk
    }
    if (touched == 0x9e3779b9u && a.n < 0) a.ydec[0] = 0.0;            // (never: keeps the touches alive)
}

template <int DT>
hipError_t launch_sq_q(const SqArgs &a, int q, dim3 grid, hipStream_t s) {
    auto go = [&](auto qc) -> hipError_t {
        constexpr int QS = decltype(qc)::value, qq = 1 << QS, NCP = 2 * qq >= 16 ? 1 : 16 / (2 * qq);
        const size_t lds = (size_t) sq_lds(NCP, SQ_LC / qq).total;
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
    // decimation 16 runs here too (the template covers it, SDRTERM_DIAG=sq=16 selects it) but two passes of the wide kernel are faster
    // there: 389 against 238 GS/s (profiles/r05_experiments.txt section 7) -- a chunk costs this kernel ~30 us before its first local tile
    const char *e = sdrt_diag("sq");
    const int maxq = e && atoi(e) == 16 ? 16 : 8;
    return (q == 2 || q == 4 || q == 8 || q == 16) && q <= maxq && (dt == SDRT_S16 || dt == SDRT_U16) && n % (16 * SQ_LC) == 0 && nc >= 16 && nc <= SQ_MAXCOL;
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
