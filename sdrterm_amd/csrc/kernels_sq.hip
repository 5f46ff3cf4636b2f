// chain_sq_kernel: unpack .. decimate for decimation 2, 4, 8, 16 of 8- and 16-bit records in ONE pass over the input, on the int8 matrix
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
//   (a DPP row scan per 16-column tile, the tiles' totals chained by eight lanes), and every output adds its share of its column's
//   two states.
// Arithmetic: exact integer slicing (x = 256 hi + lo' + offset), S = 5 balanced base-256 digits of the weights, int32 sums, one fp64
// recombination per value -- as kernels_i8w.hip; the constant the slices leave out and the normalisation
// bias come back through ybias, calibrated with this kernel's own response to a zero / a unit chunk (api.hip: setup_sq).
//
// One persistent 256-thread workgroup per CU keeps a row's planes and tables in LDS (80 KB aggregates + 10 .. 40 KB master + the
// states' shares and the phasors: 152 of 160 KB at decimation 2) and walks over chunks; ONE wave per SIMD with the whole register
// file, wave w <-> tiles w, w + 4 (16 columns each): the byte slices of both tiles (128 registers) stay in registers from the
// aggregate phase to the local phase, and every A fragment read from LDS feeds the matrix instructions of BOTH tiles.
//   per chunk:  E  edge operators (head / tail, 28 samples each); their loads go out before A1's, their sums come while A1's fly
//               A1 aggregate tiles, scanned over the tile's 16 columns in the lanes of a DPP row; lane (g, nb) keeps mode g of
//                  column nb in registers, the tile totals go to LDS                                            | barrier
//               P2 tile carries: eight chains in eight lanes of one wave, the tiles one after the other       | barrier
//               A2 per row tile (8 outputs x 2 parts of every column of both tiles) 72 + 72 matrix instructions; between their
//                  quarters the vector work of the PREVIOUS row tile: recombination and rotation, what the lane's mode adds to
//                  all eight outputs of its column (reduce-scattered over the four lane groups by permlane swaps: every lane ends
//                  with the four modes' sum for its own two outputs), ybias, pair product, store; the next chunk's records
//                  asked for behind the last matrix instruction                                       (no third barrier)
//   (8-bit records: one byte slice, 16 tiles per chunk: A1 and A2 twice per chunk, for the wave's tiles w, w + 4 and w + 8, w + 12)
// LDS reads of A fragments are issued a quarter (two k-steps, ten fragments) ahead of the matrix instructions that take them --
// written one fragment before its use the compiler waited for every single read (lgkmcnt(0) forty times per row tile: 4700 cycles
// per row tile and tile against 1152 of matrix work, profiles/r05_experiments.txt section 7).  What is left per row tile of both
// tiles (5000 cycles) is the SUM of 2304 cycles of matrix instructions and ~300 fp64 vector instructions at the 7.5 cycles they
// cost with one wave on the SIMD: on this chip the two do not overlap (r02_ubench_coexec.txt: 8 matrix + 22 fp64 FMAs take the sum
// of their times in every arrangement), so neither alternating them instruction by instruction nor two waves per SIMD moved it
// (section 8 of the experiments).
#include "kernels_common.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

namespace {

template <int DT> struct SqEnc;        // slices per component, what each slice XORs with, what they leave out of the value
template <> struct SqEnc<SDRT_S16> { static constexpr unsigned xlo = 0x80808080u, xhi = 0u; static constexpr double off = 128.0; };
template <> struct SqEnc<SDRT_U16> { static constexpr unsigned xlo = 0x80808080u, xhi = 0x80808080u; static constexpr double off = 32896.0; };
// 8-bit records (round 5): ONE slice -- the record's own bytes (the unsigned ones made signed by the XOR)
template <> struct SqEnc<SDRT_S8> { static constexpr unsigned xlo = 0u, xhi = 0u; static constexpr double off = 0.0; };
template <> struct SqEnc<SDRT_U8> { static constexpr unsigned xlo = 0u, xhi = 0x80808080u; static constexpr double off = 128.0; };

// (a wave works on TWO tiles at a time; the reference's chunk of 131072 bytes is 8 tiles of 16-bit records -- one such pair per
// wave -- or 16 tiles of 8-bit records: two pairs per wave, one after the other through the aggregate and the local phase)
constexpr int SQ_S = SDRT_I8_DIGITS, SQ_LC = SDRT_SQ_LC, SQ_KS = SQ_LC / 32, SQ_WV = 4, SQ_TPW = 2;
__host__ __device__ constexpr int sq_max_tiles(int bytes_per_component) { return bytes_per_component == 1 ? 16 : 8; }

struct SqLds {          // byte offsets
    int agg, master, tf, tg, cf, cg, f0, gu, psi, samp, cst, share, total;
};
__host__ __device__ inline SqLds sq_lds(int ncp, int nout, int maxt) {
    const int q = SQ_LC / nout;
    SqLds l;
    int p = 0;
    l.agg = p; p += 2 * SQ_S * SQ_KS * 64 * 16;
    l.master = p; p += SDRT_SQ_MASTER_BYTES(q, SQ_S, ncp);
    l.tf = p; p += maxt * 8 * 16;                     // the tiles' totals ...
    l.tg = p; p += maxt * 8 * 16;
    l.cf = p; p += (maxt + 1) * 8 * 16;                 // ... and what enters each tile from the left / from the right
    l.cg = p; p += (maxt + 1) * 8 * 16;
    l.f0 = p; p += 8 * 16;
    l.gu = p; p += 8 * 16;
    l.psi = p; p += maxt * 16 * 16;                   // exp(i theta 256 c)
    l.samp = p; p += SQ_LC * 16;                      // exp(i theta j), j < 256
    l.cst = p; p += 4 * 13 * 16;                      // per mode: mu^(1, 2, 4, 8, 16), tailF[mode][0..3], tailFc[mode][0..3]
    l.share = p; p += nout * 8 * 16;                  // the states' shares per output
    l.total = p;
    return l;
}

// x' + y' after exchanging halves: lanes 0-31 get x(l) + x(l+32), lanes 32-63 get y(l-32) + y(l)
__device__ __forceinline__ double sq_swap32_add(double x, double y) {
    const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned) __double2loint(x), (unsigned) __double2loint(y), false, false);
    const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned) __double2hiint(x), (unsigned) __double2hiint(y), false, false);
    return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}
// rows (16 lanes) 0 and 2 get x(row) + x(row+1), rows 1 and 3 get y(row-1) + y(row)
__device__ __forceinline__ double sq_swap16_add(double x, double y) {
    const v2u lo = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(x), (unsigned) __double2loint(y), false, false);
    const v2u hi = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(x), (unsigned) __double2hiint(y), false, false);
    return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}

// mu^k, k < 16, from the squarings mp[0..3] = mu^(1, 2, 4, 8)
__device__ __forceinline__ cd mu_pow(const cd *mp, int k) {
    cd r = k & 1 ? mp[0] : mk(1.0, 0.0);
    if (k & 2) r = cmul(r, mp[1]);
    if (k & 4) r = cmul(r, mp[2]);
    if (k & 8) r = cmul(r, mp[3]);
    return r;
}

#ifndef SDRT_SQ_TOUCH
#define SDRT_SQ_TOUCH 128          // bytes per touched word of the next chunk (64 and 32 read the same: section 8 of the experiments)
#endif
#define SQ_MFMA(A, B, C) __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C, 0, 0, 0)

template <int DT, int QS, bool PAIR>
__global__ __launch_bounds__(64 * SQ_WV, 1) void chain_sq_kernel(const SqArgs a) {
    constexpr int q = 1 << QS, NOUT = SQ_LC / q, NRT = 2 * NOUT / 16, NCP = 2 * q >= 16 ? 1 : 16 / (2 * q), S = SQ_S, KS = SQ_KS;
    constexpr int DSTR = NCP * SDRT_SQ_CS(q);          // master: 16-byte units from one digit's copies to the next digit's
    // touches of the next chunk: one word per SDRT_SQ_TOUCH bytes of the wave's two tiles, spread over the row tiles
    constexpr int TUNITS = 16 * SQ_LC * 4 / SDRT_SQ_TOUCH, NTOUCH = SQ_TPW * TUNITS / 64, TPS = (NTOUCH + NRT - 1) / NRT;
    static_assert(NRT % 2 == 0 && KS == 8 && SQ_TPW == 2, "the local phase walks row tiles in pairs, k-steps in quarters, two tiles per wave");
    typedef typename Enc<DT>::raw raw_t;
    // 8-bit records: one byte slice, 16 bytes of a column per lane and k-step, two pairs of tiles per wave
    constexpr bool B8 = Enc<DT>::size == 1;
    constexpr int NP = B8 ? 2 : 1, TT = SQ_TPW * NP, RW = B8 ? 4 : 8, SB = 2 * Enc<DT>::size, MAXT = sq_max_tiles(Enc<DT>::size);
    extern __shared__ __align__(16) unsigned char smem[];
    const SqLds lo = sq_lds(NCP, NOUT, MAXT);
    const v4i *aggL = (const v4i *) (smem + lo.agg);
    const v4i *masterL = (const v4i *) (smem + lo.master);
    cd *TFL = (cd *) (smem + lo.tf), *TGL = (cd *) (smem + lo.tg);
    cd *CFL = (cd *) (smem + lo.cf), *CGL = (cd *) (smem + lo.cg), *F0L = (cd *) (smem + lo.f0), *GuL = (cd *) (smem + lo.gu);
    cd *psiL = (cd *) (smem + lo.psi), *sampL = (cd *) (smem + lo.samp), *cstL = (cd *) (smem + lo.cst), *shareL = (cd *) (smem + lo.share);
    const SqTables *__restrict__ tab = a.tab;
    const int n = a.n, m = a.m, NC = n / SQ_LC, ntiles = NC / 16;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const bool swap = a.swap != 0;
    int lane = lane0, g = lane0 >> 4, nb = lane0 & 15;
    // which row, which chunks: workgroup b serves row b % rows, chunks (b / rows), + wpr, ...
    const int row = (int) (blockIdx.x % (unsigned) a.rows);
    const size_t wpr = gridDim.x / (unsigned) a.rows, first = blockIdx.x / (unsigned) a.rows;
    if (first >= wpr) return;

    // ---- the row's planes, phasors (ones without a shift) and constants into LDS, once ----------------------------------------------
    {
        const v4i *src = (const v4i *) (a.agg + (size_t) row * (2 * S * KS * 64 * 16));
        v4i *dst = (v4i *) (smem + lo.agg);
        for (int i = tid; i < 2 * S * KS * 64; i += 64 * SQ_WV) dst[i] = src[i];
        const v4i *srm = (const v4i *) (a.master + (size_t) row * SDRT_SQ_MASTER_BYTES(q, S, NCP));
        v4i *dsm = (v4i *) (smem + lo.master);
        for (int i = tid; i < SDRT_SQ_MASTER_BYTES(q, S, NCP) / 16; i += 64 * SQ_WV) dsm[i] = srm[i];
        for (int i = tid; i < NC; i += 64 * SQ_WV) psiL[i] = ((const cd *) a.nco_col)[(size_t) row * NC + i];
        for (int i = tid; i < SQ_LC; i += 64 * SQ_WV) sampL[i] = ((const cd *) a.nco_samp)[(size_t) row * SQ_LC + i];
        for (int i = tid; i < NOUT * 8; i += 64 * SQ_WV) shareL[i] = ((const cd *) a.share)[i];
        if (tid < 4) {
            cd *cs = cstL + tid * 13;
            cd v = mk(tab->mu[tid][0], tab->mu[tid][1]);
            for (int k = 0; k < 5; ++k) { cs[k] = v; v = cmul(v, v); }
            for (int j = 0; j < 4; ++j) { cs[5 + j] = mk(tab->tailF[tid][j][0], tab->tailF[tid][j][1]); cs[9 + j] = mk(tab->tailFc[tid][j][0], tab->tailFc[tid][j][1]); }
        }
    }
    __syncthreads();
    // Every workgroup does the same work in the same time: started together they all read their 128 KB of a chunk in the same few
    // microseconds (32 MB at the memory's full rate, 10 - 17 thousand cycles of waiting per chunk, the next chunk's lines touched
    // ahead or not: an XCD's L2 is exactly one round of its workgroups) and leave the memory idle for the rest of the chunk.  With
    // a.stagger the workgroups of an XCD start a.stagger x (position in the XCD) sleeps of 64 cycles apart and stay apart
    if (a.stagger > 0)
        for (int i = (int) (blockIdx.x >> 3) * a.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(1);
    // (two slices: accumulator index a holds level a + 1 -- digit a x high slice, digit a + 1 x low slice; one slice: level a)
    const double cagg = a.comb_agg * (B8 ? 1.0 : 256.0), cloc = a.comb_loc * (B8 ? 1.0 : 256.0);
    const double off = SqEnc<DT>::off;
    // (this thread's edge coefficients, the phasors of its two edge samples and the powers that carry a tile's entry states to its
    // columns are the same for every chunk -- and are fetched / formed again in every chunk: 32 registers held through the local phase
    // are what that phase is short of)
    const int e_o = (tid >> 5) & 7, e_l = tid & 31;
    unsigned touched = 0;                                                     // (the next chunk's lines pulled towards the L2: see A2)
    // A chunk's records are asked for as soon as the registers for them are free: behind the last matrix instruction of the chunk
    // before (the byte slices are dead there), so that they travel during its last row tile's vector work and this chunk's E.
    // E: the chunk's first and last 28 records; A1: the wave's two tiles, 16 columns each (lane (g, nb): 32 bytes of column nb per k-step)
    raw_t e_raw[2][2] = {{0, 0}, {0, 0}};
    uint32_t cur[TT][KS][RW];
    auto ask_for = [&](const unsigned char *chunk_raw) {
        if (e_l <= SDRT_EDGE) {
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const long idx = tl ? (long) n - (SDRT_EDGE + 1) + e_l : e_l;
                const raw_t *rp = (const raw_t *) chunk_raw + 2 * idx;
                e_raw[tl][0] = rp[0];
                e_raw[tl][1] = rp[1];
            }
        }
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            // (a tile the chunk does not have reads the wave's first tile again: its sums are never used, and a branch around each load
            // made every load wait for the one before -- 80 k cycles for the 32 of a chunk of 8-bit records)
            const int tl = wave + SQ_WV * j < ntiles ? wave + SQ_WV * j : 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                load_words<RW * 4>(chunk_raw + ((size_t) (tl * 16 + (lane0 & 15)) * SQ_LC + 32 * ks + 8 * (lane0 >> 4)) * SB, cur[j][ks]);
        }
    };
    ask_for((const unsigned char *) a.raw + first * (size_t) n * SB);

    for (size_t chunk = first; chunk < a.n_chunks; chunk += wpr) {
        const unsigned char *chunk_raw = (const unsigned char *) a.raw + chunk * (size_t) n * SB;
        // (the lane's coordinates made opaque per chunk: everything addressed through them is then computed where it is used.  Hoisted
        // out of this loop -- sixteen load addresses, the LDS addresses of every phase -- those values outlive the whole iteration, the
        // allocator spills them, and every reload waits for all memory traffic in flight, one load address at a time)
        asm volatile("" : "+v"(lane), "+v"(g), "+v"(nb));
#ifdef SDRT_STAMPS
        // diagnostic build (tools/stamps_sq.py): phase stamps of the workgroup's third chunk
        unsigned long long *stp = (a.stamps && chunk == first + 2 * wpr && (int) (blockIdx.x * SQ_WV + wave) < a.stamp_wgs * 4 && lane0 == 0)
                                      ? a.stamps + ((size_t) blockIdx.x * SQ_WV + wave) * 16 : nullptr;
#define SQ_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                         if (stp) stp[i] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SQ_STAMP(i) do { } while (0)
#endif
        SQ_STAMP(0);
        // ---- E: this thread's two edge coefficients (its records: ask_for)
        cd e_head = mk(0.0, 0.0), e_tailc = mk(0.0, 0.0);
        if (e_l <= SDRT_EDGE) {
            e_head = mk(tab->head[e_o >> 1][e_l][0], tab->head[e_o >> 1][e_l][1]);
            e_tailc = mk(tab->tailU[e_o >> 1][e_l][0], tab->tailU[e_o >> 1][e_l][1]);
        }
        // ---- E (sums).  F_0,i = sum_l head[i][l] u[l], the sample part of G_end,i = sum_l tailU[i][l] u[n - 28 + l], per real
        // stream; u = (x - off) exp(i theta t): the same x - off the slices carry (ybias returns what is left out)
        {
            const int st = e_o & 1;
            cd f = mk(0.0, 0.0), gq = mk(0.0, 0.0);
            if (e_l <= SDRT_EDGE) {
                const int i1 = n - (SDRT_EDGE + 1) + e_l;
                const cd e_ph0 = cmul(sampL[e_l], psiL[0]), e_ph1 = cmul(sampL[i1 & (SQ_LC - 1)], psiL[i1 / SQ_LC]);
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const double xr = decode<DT>(e_raw[tl][0], swap) - off, xi = decode<DT>(e_raw[tl][1], swap) - off;
                    const cd ph = tl ? e_ph1 : e_ph0;
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
            if (e_l == 0) { F0L[e_o] = f; GuL[e_o] = gq; }
        }
        SQ_STAMP(1);
        // ---- A1: byte slices (kept for A2), the aggregates of both tiles from ONE read of every A fragment, the scans over a tile's columns
        v4i bhi[TT][KS], blo[B8 ? 1 : TT][KS];
        // lane (g, nb): mode g of column nb of tile j: the causal state at the column's start, the anticausal one at the start of the
        // next column, per real stream -- first as if the tile's neighbours were zero (A1), then with the tile's carries (A2)
        cd Fl[TT][2], Gl[TT][2];
        {
            if constexpr (B8) {
#pragma unroll
                for (int j = 0; j < TT; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int w = 0; w < 4; ++w) bhi[j][ks][w] = (int) (cur[j][ks][w] ^ SqEnc<DT>::xhi);
            } else {
                const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
#pragma unroll
                for (int j = 0; j < TT; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            blo[j][ks][w] = (int) (__builtin_amdgcn_perm(cur[j][ks][2 * w + 1], cur[j][ks][2 * w], sel_lo) ^ SqEnc<DT>::xlo);
                            bhi[j][ks][w] = (int) (__builtin_amdgcn_perm(cur[j][ks][2 * w + 1], cur[j][ks][2 * w], sel_hi) ^ SqEnc<DT>::xhi);
                        }
            }
            // (each slice ONE value in four consecutive accumulation registers from here on: only matrix instructions read them.  Left
            // to the allocator the slices sat there anyway, word by word, and were copied out in front of every matrix instruction
            // that takes one: 150 of the 590 vector instructions of a row tile)
#pragma unroll
            for (int j = 0; j < TT; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if constexpr (B8) asm volatile("" : "+a"(bhi[j][ks]));
                    else asm volatile("" : "+a"(bhi[j][ks]), "+a"(blo[j][ks]));
                }
        }
        static_for<0, NP>([&](auto pc) {
        constexpr int JB = SQ_TPW * decltype(pc)::value;          // this pair's tiles: wave + 4 (JB + j)
        if (wave + SQ_WV * JB < ntiles) {
            v4i Aq[2][2 * S], acc[SQ_TPW][S];
            auto load_agg = [&](auto stc) {
                constexpr int st = decltype(stc)::value, rt = st >> 2, kq = st & 3;
#pragma unroll
                for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
                    for (int d = 0; d < S; ++d) Aq[st & 1][ksl * S + d] = aggL[((rt * S + d) * KS + 2 * kq + ksl) * 64 + lane];
            };
            load_agg(std::integral_constant<int, 0>{});
            static_for<0, 8>([&](auto stc) {
                constexpr int st = decltype(stc)::value, rt = st >> 2, kq = st & 3;
                if constexpr (kq == 0) {
#pragma unroll
                    for (int j = 0; j < SQ_TPW; ++j)
#pragma unroll
                        for (int d = 0; d < S; ++d) acc[j][d] = (v4i) {0, 0, 0, 0};
                }
                if constexpr (st + 1 < 8) load_agg(std::integral_constant<int, st + 1>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
                    for (int d = 0; d < S; ++d)
#pragma unroll
                        for (int j = 0; j < SQ_TPW; ++j) {
                            acc[j][d] = SQ_MFMA(Aq[st & 1][ksl * S + d], bhi[JB + j][2 * kq + ksl], acc[j][d]);                     // digit d x high slice: level d + 1
                            if constexpr (!B8) if (d >= 1) acc[j][d - 1] = SQ_MFMA(Aq[st & 1][ksl * S + d], blo[B8 ? 0 : JB + j][2 * kq + ksl], acc[j][d - 1]);    // x low slice: level d
                        }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (kq == 3) {
                    const cd *cstG = cstL + g * 13;                                   // mode g = this lane's aggregate rows
                    const cd m1 = cstG[0], m2 = cstG[1], m4 = cstG[2], m8 = cstG[3];
#pragma unroll
                    for (int j = 0; j < SQ_TPW; ++j) {
                        const int tile = wave + SQ_WV * (JB + j), col = tile * 16 + nb;
                        if (tile >= ntiles) { if constexpr (rt == 0) Fl[JB + j][0] = Fl[JB + j][1] = mk(0.0, 0.0); else Gl[JB + j][0] = Gl[JB + j][1] = mk(0.0, 0.0); }
                        else {
                            const cd ps = psiL[col];
                            double val[4];
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                double sum = (double) acc[j][S - 1][v];
#pragma unroll
                                for (int d = S - 2; d >= 0; --d) sum = fma(sum, 256.0, (double) acc[j][d][v]);
                                val[v] = sum * cagg;
                            }
                            // lane (g, nb): mode g of column nb; (val0, val1) the state of the re stream, (val2, val3) of the im stream.  The column's
                            // common phasor rotates the complex DATA: re' = c re - s im, im' = s re + c im (stream by stream, both state components)
                            cd x0 = mk(fma(ps.re, val[0], -(ps.im * val[2])), fma(ps.re, val[1], -(ps.im * val[3])));
                            cd x1 = mk(fma(ps.im, val[0], ps.re * val[2]), fma(ps.im, val[1], ps.re * val[3]));
                            if constexpr (rt == 0) {
                                // causal: inclusive scan over the tile's 16 columns (the lanes of a DPP row), F(c + 1) = mu F(c) + A_c from zero
                                x0 = cfma(m1, dpp_row<0x111>(x0), x0); x1 = cfma(m1, dpp_row<0x111>(x1), x1);
                                x0 = cfma(m2, dpp_row<0x112>(x0), x0); x1 = cfma(m2, dpp_row<0x112>(x1), x1);
                                x0 = cfma(m4, dpp_row<0x114>(x0), x0); x1 = cfma(m4, dpp_row<0x114>(x1), x1);
                                x0 = cfma(m8, dpp_row<0x118>(x0), x0); x1 = cfma(m8, dpp_row<0x118>(x1), x1);
                                if (nb == 15) { TFL[tile * 8 + g * 2] = x0; TFL[tile * 8 + g * 2 + 1] = x1; }
                                Fl[JB + j][0] = dpp_row<0x111>(x0);           // exclusive: the state at the column's own start
                                Fl[JB + j][1] = dpp_row<0x111>(x1);
                            } else {
                                // anticausal: G(c) = B_c + mu G(c + 1), from zero beyond the tile
                                x0 = cfma(m1, dpp_row<0x101>(x0), x0); x1 = cfma(m1, dpp_row<0x101>(x1), x1);
                                x0 = cfma(m2, dpp_row<0x102>(x0), x0); x1 = cfma(m2, dpp_row<0x102>(x1), x1);
                                x0 = cfma(m4, dpp_row<0x104>(x0), x0); x1 = cfma(m4, dpp_row<0x104>(x1), x1);
                                x0 = cfma(m8, dpp_row<0x108>(x0), x0); x1 = cfma(m8, dpp_row<0x108>(x1), x1);
                                if (nb == 0) { TGL[tile * 8 + g * 2] = x0; TGL[tile * 8 + g * 2 + 1] = x1; }
                                Gl[JB + j][0] = dpp_row<0x101>(x0);           // the state at the start of the NEXT column
                                Gl[JB + j][1] = dpp_row<0x101>(x1);
                            }
                        }
                    }
                }
            });
        }
        else {
#pragma unroll
            for (int j = 0; j < SQ_TPW; ++j) Fl[JB + j][0] = Fl[JB + j][1] = Gl[JB + j][0] = Gl[JB + j][1] = mk(0.0, 0.0);
        }
        });
        SQ_STAMP(2);
        __syncthreads();
        SQ_STAMP(3);
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
        SQ_STAMP(4);
        __syncthreads();
        SQ_STAMP(5);
        // ---- A2: the local part of every output of both tiles' columns, the states' shares, ybias, store ------------------------------------
        static_for<0, NP>([&](auto pc) {
        constexpr int JB = SQ_TPW * decltype(pc)::value;
        const bool have1 = wave + SQ_WV * (JB + 1) < ntiles;
        if (wave + SQ_WV * JB < ntiles) {
            // the columns' true states: the tile-local value + the tile's carry through nb / 15 - nb columns (mode g)
            {
                const cd pfc = mu_pow(cstL + g * 13, nb), pgc = mu_pow(cstL + g * 13, 15 - nb);
#pragma unroll
                for (int j = 0; j < SQ_TPW; ++j) {
                    const int tcl = wave + SQ_WV * (JB + j) < ntiles ? wave + SQ_WV * (JB + j) : 0;
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        Fl[JB + j][st] = cfma(pfc, CFL[tcl * 8 + g * 2 + st], Fl[JB + j][st]);
                        Gl[JB + j][st] = cfma(pgc, CGL[tcl * 8 + g * 2 + st], Gl[JB + j][st]);
                    }
                }
            }
            const int r = lane & 15, part = r & 1, bl = r >> 1;                 // A operand: row r of the row tile = output bl, part
            // 16-byte units (tables.h: SDRT_SQ_CS / _PS): part PS + (d NCP + copy) CS + (64 ks + 16 g - 16 (s >> 3) + 512) / 16 for the
            // output at sample s = (8 rt + bl) q of the column: row tile rt sits rt q units before row tile 0
            const int s0 = bl << QS;
            const int abase0 = part * SDRT_SQ_PS(q, S, NCP) + ((s0 & 7) >> QS) * SDRT_SQ_CS(q) + g - (s0 >> 3) + 32;
            const size_t item = (size_t) row * a.n_chunks + chunk;
            double *yd = a.ydec + item * (size_t) (PAIR ? m / 2 : m) * 2;
            const cd *yb = (const cd *) a.ybias + (size_t) row * m;
            // (one word of every SDRT_SQ_TOUCH bytes of this wave's tiles of the NEXT chunk, spread over the row tiles: the next
            // iteration's loads find at least part of it on the way)
            const bool more = chunk + wpr < a.n_chunks;
            const unsigned char *nx = chunk_raw + (more ? wpr * (size_t) n * SB : 0);
            unsigned tv = 0;

            v4i Aq[2][2 * S], accA[SQ_TPW][S], accB[SQ_TPW][S];
            auto load_master = [&](auto kqc, const int rt) {
                constexpr int kq = decltype(kqc)::value;
                const int ab = abase0 - rt * q;
#pragma unroll
                for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
                    for (int d = 0; d < S; ++d) Aq[kq & 1][ksl * S + d] = masterL[ab + d * DSTR + 4 * (2 * kq + ksl)];
            };
            // the vector unit's share of row tile rt, both tiles, in four pieces (one behind each quarter of the next row tile's matrix
            // work; each piece's LDS reads go out ahead of that quarter, beside the A fragments of the next):
            //   0  recombination of the int32 sums, the outputs' own phasors            -> y2[tile][h], outputs 8 rt + 2 g + h of column nb
            //   1  what the lane's mode g adds to outputs 8 rt + {0, 1, 4, 5} of its column (both streams), halves of the wave exchanged
            //   2  the same for {2, 3, 6, 7}; rows exchanged: every lane now holds the four modes' sum for its own two outputs
            //   3  ybias, pair product, store
            cd y2[SQ_TPW][2], smp[2], psj[SQ_TPW], pf[4], pg[4], ybv[SQ_TPW][2];
            double r1[SQ_TPW][2][2];                        // [tile][k = 0, 1][re, im] after the exchange of halves
            auto epi_recombine_loads = [&](const int rt) {
#pragma unroll
                for (int h = 0; h < 2; ++h) smp[h] = sampL[(rt * 8 + 2 * g + h) << QS];
#pragma unroll
                for (int j = 0; j < SQ_TPW; ++j) psj[j] = psiL[(wave + SQ_WV * (JB + j) < ntiles ? wave + SQ_WV * (JB + j) : 0) * 16 + nb];
            };
            auto epi_recombine = [&](const v4i (&pv)[SQ_TPW][S]) {
#pragma unroll
                for (int j = 0; j < SQ_TPW; ++j) {
                    double val[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        double sum = (double) pv[j][S - 1][v];
#pragma unroll
                        for (int d = S - 2; d >= 0; --d) sum = fma(sum, 256.0, (double) pv[j][d][v]);
                        val[v] = sum * cloc;
                    }
                    // result rows 4 g + v of column nb: outputs b0 (val0 + i val1) and b0 + 1 (val2 + i val3)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const cd rot = cmul(psj[j], smp[h]);                             // exp(i theta (256 c + s)): the output's own phasor
                        y2[j][h] = mk(fma(rot.re, val[2 * h], -(rot.im * val[2 * h + 1])), fma(rot.re, val[2 * h + 1], rot.im * val[2 * h]));
                    }
                }
            };
            // share[b][i] = 2 p_i^(b q), share[b][4 + i] = 2 p_i^(256 - b q): what the states at the column's start / at the next column's
            // start add to output b (inside 2 Re), per real stream
            auto epi_modes_loads = [&](const int rt, auto kc) {
                constexpr int k0 = decltype(kc)::value;           // 0: outputs {0, 1 | 4, 5} of the row tile, 2: {2, 3 | 6, 7}
                const cd *sh = shareL + (size_t) (rt * 8) * 8 + g;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int b = k0 + (u & 1) + 4 * (u >> 1);
                    pf[u] = sh[b * 8];
                    pg[u] = sh[b * 8 + 4];
                }
            };
            auto epi_modes = [&](auto kc) {
                constexpr int k0 = decltype(kc)::value;
#pragma unroll
                for (int j = 0; j < SQ_TPW; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            // outputs k0 + k (wanted by the lower half of the wave) and k0 + k + 4 (by the upper half)
                            const double lo_ = fma(pf[k].re, Fl[JB + j][c].re, fma(-pf[k].im, Fl[JB + j][c].im, fma(pg[k].re, Gl[JB + j][c].re, -(pg[k].im * Gl[JB + j][c].im))));
                            const double hi_ = fma(pf[2 + k].re, Fl[JB + j][c].re, fma(-pf[2 + k].im, Fl[JB + j][c].im, fma(pg[2 + k].re, Gl[JB + j][c].re, -(pg[2 + k].im * Gl[JB + j][c].im))));
                            const double sw = sq_swap32_add(lo_, hi_);
                            if constexpr (k0 == 0) r1[j][k][c] = sw;
                            else {
                                // rows: even g takes outputs 0, 1 of its half, odd g 2, 3
                                if (c == 0) y2[j][k].re += sq_swap16_add(r1[j][k][0], sw);
                                else y2[j][k].im += sq_swap16_add(r1[j][k][1], sw);
                            }
                        }
            };
            // ybias of a row tile's outputs: asked for three quarters of a row tile before the store takes it, and BEFORE that row tile's
            // touch of the next chunk -- a wait for a load also waits for every older load and store, and the touch goes to HBM
            auto load_yb = [&](const int rt) {
#pragma unroll
                for (int j = 0; j < SQ_TPW; ++j) {
                    const int tcl = wave + SQ_WV * (JB + j) < ntiles ? wave + SQ_WV * (JB + j) : 0;
                    const cd *p = yb + (size_t) (tcl * 16 + nb) * NOUT + rt * 8 + 2 * g;
                    ybv[j][0] = p[0];
                    ybv[j][1] = p[1];
                }
            };
            auto epi_store = [&](const int rt) {
#pragma unroll
                for (int j = 0; j < SQ_TPW; ++j) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        y2[j][h].re = (y2[j][h].re + ybv[j][h].re) + 0.0;           // -0.0 -> +0.0 (angle(-0) would be pi)
                        y2[j][h].im = (y2[j][h].im + ybv[j][h].im) + 0.0;
                    }
                    if (wave + SQ_WV * (JB + j) < ntiles) {
                        const int col = (wave + SQ_WV * (JB + j)) * 16 + nb, k0 = col * NOUT + rt * 8 + 2 * g;
                        if constexpr (PAIR) {
                            // FM without a debug buffer: the product the discriminator takes the angle of (demodulation.py:25-32)
                            *(double2 *) (yd + (size_t) k0) = make_double2(y2[j][0].re * y2[j][1].re + y2[j][0].im * y2[j][1].im, y2[j][0].im * y2[j][1].re - y2[j][0].re * y2[j][1].im);
                        } else {
                            *(double2 *) (yd + 2 * (size_t) k0) = make_double2(y2[j][0].re, y2[j][0].im);
                            *(double2 *) (yd + 2 * (size_t) k0 + 2) = make_double2(y2[j][1].re, y2[j][1].im);
                        }
                    }
                }
            };
            // one row tile: 4 quarters of 36 matrix instructions into `ac`; behind each the piece of the previous row tile's vector work (`pv`)
            auto step = [&](auto epc, auto lastc, v4i (&ac)[SQ_TPW][S], const v4i (&pv)[SQ_TPW][S], const int rt) {
                constexpr bool EPI = decltype(epc)::value, LAST = decltype(lastc)::value;
#pragma unroll
                for (int j = 0; j < SQ_TPW; ++j)
#pragma unroll
                    for (int d = 0; d < S; ++d) ac[j][d] = (v4i) {0, 0, 0, 0};
                static_for<0, 4>([&](auto kqc) {
                    constexpr int kq = decltype(kqc)::value;
                    if constexpr (kq < 3) load_master(std::integral_constant<int, kq + 1>{}, rt);
                    else if constexpr (!LAST) load_master(std::integral_constant<int, 0>{}, rt + 1);
                    if constexpr (EPI) {
                        if constexpr (kq == 0) epi_recombine_loads(rt - 1);
                        if constexpr (kq == 1) epi_modes_loads(rt - 1, std::integral_constant<int, 0>{});
                        if constexpr (kq == 2) epi_modes_loads(rt - 1, std::integral_constant<int, 2>{});
                    }
                    if constexpr (kq == 0) {
                        if constexpr (EPI) load_yb(rt - 1);
                        touched ^= tv;
                        tv = 0;
#pragma unroll
                        for (int t = 0; t < TPS; ++t) {
                            const int ti = rt * TPS + t;                                  // touch ti covers units 64 ti .. 64 ti + 63 of the wave's TUNITS per tile
                            const bool ok = !B8 && a.touch && ti < NTOUCH && (ti * 64 < TUNITS || have1);
                            const int u = ti * 64 + lane;
                            // (not touching: the chunk's first word -- in bounds for every encoding and chunk size)
                            const size_t at = ok ? (size_t) (wave * 16 * SQ_LC * 4) + (size_t) (u / TUNITS) * (SQ_WV * 16 * SQ_LC * 4) + (size_t) (u % TUNITS) * SDRT_SQ_TOUCH : 0;
                            tv ^= *(const unsigned *) (nx + at);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
                        for (int d = 0; d < S; ++d)
#pragma unroll
                            for (int j = 0; j < SQ_TPW; ++j) {
                                ac[j][d] = SQ_MFMA(Aq[kq & 1][ksl * S + d], bhi[JB + j][2 * kq + ksl], ac[j][d]);
                                if constexpr (!B8) if (d >= 1) ac[j][d - 1] = SQ_MFMA(Aq[kq & 1][ksl * S + d], blo[B8 ? 0 : JB + j][2 * kq + ksl], ac[j][d - 1]);
                            }
                    if constexpr (EPI) {
                        if constexpr (kq == 0) epi_recombine(pv);
                        if constexpr (kq == 1) epi_modes(std::integral_constant<int, 0>{});
                        if constexpr (kq == 2) epi_modes(std::integral_constant<int, 2>{});
                        if constexpr (kq == 3) epi_store(rt - 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            constexpr std::true_type yes{};
            constexpr std::false_type no{};
            SQ_STAMP(6);
            load_master(std::integral_constant<int, 0>{}, 0);
            step(no, no, accA, accB, 0);
#pragma unroll 1
            for (int rt = 1; rt < NRT - 1; rt += 2) {
                step(yes, no, accB, accA, rt);
                step(yes, no, accA, accB, rt + 1);
            }
            step(yes, yes, accB, accA, NRT - 1);
            SQ_STAMP(7);
            load_yb(NRT - 1);
            __builtin_amdgcn_sched_barrier(0);
            // (behind ybias: the last row tile's store waits for that, not for these; the last chunk asks for itself again)
            if constexpr (JB + SQ_TPW == TT) ask_for(nx);
            __builtin_amdgcn_sched_barrier(0);
            epi_recombine_loads(NRT - 1);
            epi_recombine(accB);
            epi_modes_loads(NRT - 1, std::integral_constant<int, 0>{});
            epi_modes(std::integral_constant<int, 0>{});
            epi_modes_loads(NRT - 1, std::integral_constant<int, 2>{});
            epi_modes(std::integral_constant<int, 2>{});
            epi_store(NRT - 1);
            touched ^= tv;
        } else if constexpr (JB + SQ_TPW == TT) {
            // (a wave without this pair, in a chunk shorter than the reference's, asks here: on BOTH paths, so that the registers of the
            // records are not held through the local phase for the sake of the path that does not write them)
            ask_for(chunk_raw + (chunk + wpr < a.n_chunks ? wpr * (size_t) n * SB : 0));
        }
        });
        // (no barrier here: what the next chunk writes before its first barrier -- F0 / Gu in E, the tile totals in A1 -- was read by
        // wave 0 before this chunk's second barrier; the carries it writes in P2 come after a barrier every wave reaches only behind
        // its reads of them here)
        SQ_STAMP(10);
        SQ_STAMP(11);
    }
    if (touched == 0x9e3779b9u && a.n < 0) a.ydec[0] = 0.0;            // (never: keeps the touches alive)
}

template <int DT>
hipError_t launch_sq_q(const SqArgs &a, int q, dim3 grid, hipStream_t s) {
    auto go = [&](auto qc, auto pc) -> hipError_t {
        constexpr int QS = decltype(qc)::value, qq = 1 << QS, NCP = 2 * qq >= 16 ? 1 : 16 / (2 * qq);
        constexpr bool PAIR = decltype(pc)::value;
        const size_t lds = (size_t) sq_lds(NCP, SQ_LC / qq, sq_max_tiles(Enc<DT>::size)).total;
        static thread_local int done_dev = -1;
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (done_dev != dev) {
            hipError_t e = hipFuncSetAttribute((const void *) chain_sq_kernel<DT, QS, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            done_dev = dev;
        }
        hipLaunchKernelGGL((chain_sq_kernel<DT, QS, PAIR>), grid, dim3(64 * SQ_WV), lds, s, a);
        return hipGetLastError();
    };
    auto pick = [&](auto qc) -> hipError_t { return a.pair_out ? go(qc, std::true_type{}) : go(qc, std::false_type{}); };
    switch (q) {
        case 2: return pick(std::integral_constant<int, 1>{});
        case 4: return pick(std::integral_constant<int, 2>{});
        case 8: return pick(std::integral_constant<int, 3>{});
        case 16: return pick(std::integral_constant<int, 4>{});
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool chain_sq_supported(int q, int dt, int n) {
    const int nc = n / SQ_LC;
    // (SDRTERM_DIAG=sq=8 leaves decimation 16 to the two passes of the wide kernel, as before the A fragments were read ahead:
    // 389 - 413 GS/s there against 457 here, profiles/r05_experiments.txt section 9)
    const char *e = sdrt_diag("sq");
    const int maxq = e && atoi(e) == 8 ? 8 : 16;
    const bool two = dt == SDRT_S16 || dt == SDRT_U16, one = dt == SDRT_S8 || dt == SDRT_U8;
    if (one && sdrt_diag_int("sq8", 1) == 0) return false;                  // (SDRTERM_DIAG=sq8=0: 8-bit records on the forms they had before round 5)
    return (q == 2 || q == 4 || q == 8 || q == 16) && q <= maxq && (two || one) && n % (16 * SQ_LC) == 0 && nc >= 16 && nc <= 16 * sq_max_tiles(one ? 1 : 2);
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
    if (dt == SDRT_S8) return launch_sq_q<SDRT_S8>(a, q, grid, s);
    if (dt == SDRT_U8) return launch_sq_q<SDRT_U8>(a, q, grid, s);
    return hipErrorInvalidValue;
}
