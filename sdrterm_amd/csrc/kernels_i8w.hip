// chain_i8w_kernel: the chain kernel with WIDE matrix-core columns (256 samples) on v_mfma_i32_16x16x64_i8, persistent
// workgroups and a three-stage software pipeline.
//
// Why: with one block of L <= 64 samples per matrix-core column (kernels_i8t.hip) every 64 samples cost a recombination of
// 8 int32 sums to fp64, four DPP scan steps and an output reduction, and that fp64 vector work bounds the kernel.  Here a
// column is LC = 256 samples: the matrix cores also produce, per column, the column-local part of each of its nout = LC/q
// outputs (a third row tile), so the per-column fp64 work is spent once per 256 samples.  The digit planes for that (up to
// 120 KB) only fit LDS once per CU, hence one persistent 512-thread workgroup per CU that walks over its (chunk, row) items
// with the planes resident, and because the tile-carry and combine phases of an item are short and latency bound they are
// software-pipelined against the matrix-core phase of the following items:
//
//   iteration it:   E(it)  edge operators of item it                      (all threads, 16 outputs x 32 lanes)
//                   M(it)  matrix-core phase: wave w <-> tile w (16 columns = 4096 samples); column aggregates -> LDS,
//                          column-local outputs stay in registers of the lane that will finish them
//                   P2(it-1) tile carries of item it-1: two waves (one per real stream), DPP scans over <= 16 tiles
//                   P3(it-2) add the carried states to the outputs of item it-2 and store them
//   one workgroup barrier per iteration; the LDS carry buffers are triple buffered.
//
// Arithmetic: identical to kernels_i8t.hip (error-free int8 slicing, S digit planes, exact int32 sums, one fp64
// recombination), see DESIGN.md section 3.  The lowest (digit 0 x low slice) product of 16-bit encodings is dropped: it
// is below the quantisation step of the planes.
#include "kernels_common.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

namespace {
template <int DT> struct W8Enc;   // slices per component, what each slice XORs with
template <> struct W8Enc<SDRT_S8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0u, xhi = 0u; };
template <> struct W8Enc<SDRT_U8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };
template <> struct W8Enc<SDRT_S16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };
template <> struct W8Enc<SDRT_U16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0x80808080u; };

// x' + y' after exchanging halves: lanes 0-31 get x(l) + x(l+32), lanes 32-63 get y(l-32) + y(l)
__device__ __forceinline__ double swap32_add(double x, double y) {
    const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned) __double2loint(x), (unsigned) __double2loint(y), false, false);
    const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned) __double2hiint(x), (unsigned) __double2hiint(y), false, false);
    return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}
// rows (16 lanes) 0 and 2 get x(row) + x(row+1), rows 1 and 3 get y(row-1) + y(row)
__device__ __forceinline__ double swap16_add(double x, double y) {
    const v2u lo = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(x), (unsigned) __double2loint(y), false, false);
    const v2u hi = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(x), (unsigned) __double2hiint(y), false, false);
    return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}

struct WideLds {        // byte offsets
    int wmain, wloc, cF, cG, bp, tpw, tailF, tailFc, buf, buf_stride, total;
    // per buffer (offsets in 16-byte units from its start): carF [(NT+1)*8], carG [(NT+1)*8], F0 [8], Gu [8]
};
__host__ __device__ inline WideLds wide_lds(int S, int nout, int nt) {
    WideLds l;
    int p = 0;
    l.wmain = p; p += 2 * S * 8 * 1024;
    l.wloc = p; p += S * 8 * 1024;                  // local tile padded to 16 rows (rows >= 2 nout are zero)
    l.cF = p; p += SDRT_TB * nout * 4 * 16;
    l.cG = p; p += SDRT_TB * nout * 4 * 16;
    l.bp = p; p += (nout + 1) * 4 * 16;
    l.tpw = p; p += (SDRT_WIDE_MAXTILES + 1) * 4 * 16;
    l.tailF = p; p += 16 * 16;
    l.tailFc = p; p += 16 * 16;
    l.buf = p;
    l.buf_stride = (2 * (nt + 1) * 8 + 16) * 16;
    p += 3 * l.buf_stride;
    l.total = p;
    return l;
}
}  // namespace

template <int DT, int NOUT, int NT, int S>
__global__ __launch_bounds__(512, 2) void chain_i8w_kernel(DecimArgs a) {
    constexpr int LC = SDRT_WIDE_LC, KS = LC / 32;
    constexpr int NSL = W8Enc<DT>::slices;
    constexpr int ITEM = Enc<DT>::size;
    constexpr int NR = 2 * NOUT;                     // rows of the local tile
    constexpr int OPL = NOUT / 4;                    // outputs a lane finishes per tile
    constexpr int TPW = NT / 8;                      // tiles per wave and item
    constexpr int WPK = 4 * ITEM;                    // 32-bit words per lane and k-step (8 complex samples)
    constexpr int TILE = SDRT_TB * LC;               // samples per tile
    constexpr int NSTEP = KS * TPW;                  // k-steps per wave and item
    constexpr int PF = 3;                            // k-steps of raw samples in flight
    static_assert(NOUT == 4 || NOUT == 8, "4 or 8 outputs per column");
    static_assert(NT == 8 || NT == 16, "8 or 16 tiles per chunk");
    typedef typename Enc<DT>::raw raw_t;

    extern __shared__ __align__(16) unsigned char smem[];
    const WideTables *__restrict__ wt = a.wtab;
    const int n = a.n, m = a.m;
    const bool nco = a.nco_tile != nullptr;
    const bool swap = a.swap != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nb = lane & 15;
    const WideLds lo = wide_lds(S, NOUT, NT);
    const v4i *wmainL = (const v4i *) (smem + lo.wmain);
    const v4i *wlocL = (const v4i *) (smem + lo.wloc);
    cd *cFL = (cd *) (smem + lo.cF), *cGL = (cd *) (smem + lo.cG), *bpL = (cd *) (smem + lo.bp), *tpwL = (cd *) (smem + lo.tpw);
    cd *tailFL = (cd *) (smem + lo.tailF), *tailFcL = (cd *) (smem + lo.tailFc);
    auto carF = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride); };
    auto carG = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride) + (NT + 1) * 8; };
    auto F0L = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride) + 2 * (NT + 1) * 8; };
    auto GuL = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride) + 2 * (NT + 1) * 8 + 8; };

    // ---- which items: workgroup (xcd, slot) keeps one row (digit planes resident) and walks over chunks of its XCD ------
    const unsigned bid = blockIdx.x, xcd = bid & 7u, slot = bid >> 3, nslot = gridDim.x >> 3;
    unsigned row0, rstep, stream, nstreams;
    if ((unsigned) a.rows <= nslot) {
        nstreams = nslot / (unsigned) a.rows;
        row0 = slot % (unsigned) a.rows;
        stream = slot / (unsigned) a.rows;
        rstep = (unsigned) a.rows;
        if (stream >= nstreams) return;
    } else {
        nstreams = 1; stream = 0; row0 = slot; rstep = nslot;
    }
    // chunk of step k: ((k * nstreams + stream) * 8 + xcd); the groups of 8 chunks this XCD sees: gi < ngroups
    const long ngroups = (long) a.n_chunks > (long) xcd ? ((long) a.n_chunks - (long) xcd + 7) / 8 : 0;
    const int cnt = ngroups > (long) stream ? (int) ((ngroups - (long) stream + (long) nstreams - 1) / (long) nstreams) : 0;
    if (cnt == 0) return;
    auto chunk_of = [&](int k) -> long { return ((long) k * (long) nstreams + (long) stream) * 8 + (long) xcd; };

    // ---- tables that do not depend on the row -> LDS ------------------------------------------------------------------
    for (int i = tid; i < SDRT_TB * NOUT * 4; i += 512) {
        cFL[i] = mk(wt->cF2[i >> 2][i & 3][0], wt->cF2[i >> 2][i & 3][1]);
        cGL[i] = mk(wt->cG2[i >> 2][i & 3][0], wt->cG2[i >> 2][i & 3][1]);
    }
    if (tid < (NOUT + 1) * 4) bpL[tid] = mk(wt->bp2[tid >> 2][tid & 3][0], wt->bp2[tid >> 2][tid & 3][1]);
    if (tid >= 64 && tid < 64 + (SDRT_WIDE_MAXTILES + 1) * 4) {
        const int i = tid - 64;
        tpwL[i] = mk(wt->tile_pow[i >> 2][i & 3][0], wt->tile_pow[i >> 2][i & 3][1]);
    }
    if (tid >= 192 && tid < 208) {
        const int i = tid - 192;
        tailFL[i] = mk(wt->tailF[i >> 2][i & 3][0], wt->tailF[i >> 2][i & 3][1]);
        tailFcL[i] = mk(wt->tailFc[i >> 2][i & 3][0], wt->tailFc[i >> 2][i & 3][1]);
    }
    cd mup[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mup[k] = mk(wt->mu_pow[k][g][0], wt->mu_pow[k][g][1]);
    // edge operator of this thread: output o16 = tid >> 5 (kind: 0 head -> F0, 1 tail -> Gu; chain o = mode*2 + stream), sample l
    const int e_o16 = tid >> 5, e_l = tid & 31, e_kind = e_o16 >> 3, e_o = e_o16 & 7, e_mode = e_o >> 1, e_st = e_o & 1;
    const long e_idx = e_kind ? (long) n - (SDRT_EDGE + 1) + e_l : (long) e_l;
    cd e_tab = mk(0.0, 0.0);
    if (e_l <= SDRT_EDGE)
        e_tab = e_kind ? mk(wt->tailU[e_mode][e_l][0], wt->tailU[e_mode][e_l][1]) : mk(wt->head[e_mode][e_l][0], wt->head[e_mode][e_l][1]);
    const double xoff = a.x_offset;
    const double cs_main = a.w8_combine * (NSL == 2 ? 256.0 : 1.0), cs_loc = a.w8mid_combine * (NSL == 2 ? 256.0 : 1.0);
    const size_t chunk_bytes = (size_t) n * 2 * ITEM;
    // this lane's bytes of k-step s of a tile: column nb, samples 32 s + 8 g .. + 7
    const size_t lane_off = ((size_t) nb * LC + (size_t) g * 8) * 2 * ITEM;

    for (unsigned row = row0; row < (unsigned) a.rows; row += rstep) {
        __syncthreads();                             // nobody still reads the previous row's planes
        {
            const v4i *wg = (const v4i *) (a.w8 + (size_t) row * (size_t) (2 * S * KS * 1024));
            v4i *dst = (v4i *) (smem + lo.wmain);
            for (int i = tid; i < 2 * S * KS * 64; i += 512) dst[i] = wg[i];
            const v4i *wl = (const v4i *) (a.w8mid + (size_t) row * (size_t) (S * KS * 4 * NR * 16));
            v4i *dl = (v4i *) (smem + lo.wloc);
            // global layout is compact ([digit][kstep][kg][NR rows]); in LDS every (digit, kstep, kg) group has 16 rows
            for (int i = tid; i < S * KS * 64; i += 512)
                dl[i] = (i & 15) < NR ? wl[(i >> 4) * NR + (i & 15)] : (v4i) {0, 0, 0, 0};
        }
        const cd *nco_samp_g = nco ? (const cd *) a.nco_tile + (size_t) row * LC : nullptr;                // exp(i theta j), j < LC
        const cd *nco_blk_g = nco ? (const cd *) a.nco_phase + (size_t) row * (size_t) (n / LC) : nullptr;  // exp(i theta LC c)
        cd e_rot = mk(1.0, 0.0);
        if (nco && e_l <= SDRT_EDGE) e_rot = cmul(nco_samp_g[e_idx % LC], nco_blk_g[e_idx / LC]);
        const cd *ybias = a.ybias ? (const cd *) a.ybias + (size_t) row * m : nullptr;
        __syncthreads();

        uint32_t raw[4][WPK];
        auto issue = [&](const unsigned char *chunk_raw, auto sc) {
            // k-step st (0 .. NSTEP-1) of the item: tile wave + 8 (st / KS), k-step st % KS
            constexpr int st = decltype(sc)::value;
            const unsigned char *p = chunk_raw + ((size_t) (wave + 8 * (st / KS)) * TILE + (size_t) (st % KS) * 32) * 2 * ITEM + lane_off;
            load_words<WPK * 4>(p, raw[st & 3]);
        };
        {
            const unsigned char *c0 = (const unsigned char *) a.raw + (size_t) chunk_of(0) * chunk_bytes;
            static_for<0, PF>([&](auto sc) { issue(c0, sc); });
        }
        cd yq[2][TPW][OPL];                          // column-local outputs of the two previous items
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int e = 0; e < OPL; ++e) yq[i][j][e] = mk(0.0, 0.0);

        int bcur = 0;                                // LDS buffer of iteration it; it-1 -> (bcur+2)%3, it-2 -> (bcur+1)%3
        for (int it = 0; it < cnt + 2; ++it) {
            const int b1 = bcur == 0 ? 2 : bcur - 1, b2 = bcur == 2 ? 0 : bcur + 1;
            // ---- E(it): raw edge sample first (its latency hides behind P3/P2) --------------------------------------------
            const bool have_m = it < cnt;
            const unsigned char *chunk_raw = nullptr;
            raw_t e_re = 0, e_im = 0;
            if (have_m) {
                chunk_raw = (const unsigned char *) a.raw + (size_t) chunk_of(it) * chunk_bytes;
                if (e_l <= SDRT_EDGE) {
                    const raw_t *rp = (const raw_t *) chunk_raw + 2 * e_idx;
                    e_re = rp[0];
                    e_im = rp[1];
                }
            }
            // ---- P3(it-2): carried states -> outputs, store ---------------------------------------------------------------
            if (it >= 2) {
                const long chunk2 = chunk_of(it - 2);
                double *yd = a.ydec + ((size_t) row * a.n_chunks + (size_t) chunk2) * (size_t) m * 2;
                const cd *cf = carF(b2), *cg = carG(b2);
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const int tile = wave + 8 * j;
#pragma unroll
                    for (int e = 0; e < OPL; ++e) {
                        const int b = g + 4 * e, pos = nb * NOUT + b, k = (tile * SDRT_TB + nb) * NOUT + b;
                        cd v = yq[0][j][e];
#pragma unroll
                        for (int mode = 0; mode < 4; ++mode) {
                            const cd f = cFL[pos * 4 + mode], gq = cGL[pos * 4 + mode];
                            const cd Fa = cf[tile * 8 + mode * 2 + 0], Fb = cf[tile * 8 + mode * 2 + 1];
                            const cd Ga = cg[(tile + 1) * 8 + mode * 2 + 0], Gb = cg[(tile + 1) * 8 + mode * 2 + 1];
                            v.re += f.re * Fa.re - f.im * Fa.im + gq.re * Ga.re - gq.im * Ga.im;
                            v.im += f.re * Fb.re - f.im * Fb.im + gq.re * Gb.re - gq.im * Gb.im;
                        }
                        if (ybias) { const cd yb = ybias[k]; v.re += yb.re; v.im += yb.im; }
                        *(double2 *) (yd + 2 * (size_t) k) = make_double2(v.re, v.im);
                    }
                }
            }
            // ---- P2(it-1): tile carries, one wave per real stream; DPP rows = modes, lanes of a row = tiles -----------------
            if (it >= 1 && it - 1 < cnt && (wave >> 1) == (it & 3)) {
                const int st = wave & 1, o = g * 2 + st, t = nb;
                cd *cf = carF(b1), *cg = carG(b1);
                const cd m1 = tpwL[1 * 4 + g], m2 = tpwL[2 * 4 + g], m4 = tpwL[4 * 4 + g], m8 = tpwL[8 * 4 + g];
                cd x = t < NT ? cf[(t + 1) * 8 + o] : mk(0.0, 0.0);
                x = cfma(m1, dpp_row<0x111>(x), x);
                x = cfma(m2, dpp_row<0x112>(x), x);
                x = cfma(m4, dpp_row<0x114>(x), x);
                if constexpr (NT > 8) x = cfma(m8, dpp_row<0x118>(x), x);
                const cd f0 = F0L(b1)[o];
                x = cfma(tpwL[(t < NT ? t + 1 : 0) * 4 + g], f0, x);
                if (t < NT) cf[(t + 1) * 8 + o] = x;
                if (t == 0) cf[o] = f0;
                // anticausal state at the right edge from the causal states there (all four modes of this stream) + edge samples
                cd G = GuL(b1)[o];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const cd Fj = cf[NT * 8 + j * 2 + st];
                    G = cfma(tailFL[g * 4 + j], Fj, G);
                    G = cfma(tailFcL[g * 4 + j], mk(Fj.re, -Fj.im), G);
                }
                cd y = t < NT ? cg[t * 8 + o] : mk(0.0, 0.0);
                y = cfma(m1, dpp_row<0x101>(y), y);
                y = cfma(m2, dpp_row<0x102>(y), y);
                y = cfma(m4, dpp_row<0x104>(y), y);
                if constexpr (NT > 8) y = cfma(m8, dpp_row<0x108>(y), y);
                y = cfma(tpwL[(t < NT ? NT - t : 0) * 4 + g], G, y);
                if (t < NT) cg[t * 8 + o] = y;
                if (t == 0) cg[NT * 8 + o] = G;
            }
            // the queue of column-local outputs moves on
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int e = 0; e < OPL; ++e) yq[0][j][e] = yq[1][j][e];

            if (have_m) {
                // ---- E(it) ------------------------------------------------------------------------------------------------
                {
                    cd f = mk(0.0, 0.0);
                    if (e_l <= SDRT_EDGE) {
                        const double xr = decode<DT>(e_re, swap) - xoff, xi = decode<DT>(e_im, swap) - xoff;
                        const cd u = cmul(mk(xr, xi), e_rot);
                        const double us = e_st ? u.im : u.re;
                        f = mk(e_tab.re * us, e_tab.im * us);
                    }
#pragma unroll
                    for (int d = 16; d >= 1; d >>= 1) {
                        f.re += __shfl_xor(f.re, d);
                        f.im += __shfl_xor(f.im, d);
                    }
                    if (e_l == 0) { if (e_kind) GuL(bcur)[e_o] = f; else F0L(bcur)[e_o] = f; }
                }
                // ---- M(it) ------------------------------------------------------------------------------------------------
                cd *cf = carF(bcur), *cg = carG(bcur);
                const unsigned char *next_raw = it + 1 < cnt ? (const unsigned char *) a.raw + (size_t) chunk_of(it + 1) * chunk_bytes : nullptr;
                static_for<0, TPW>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int tile = wave + 8 * j;
                    v4i acc[3][S];
                    v4i Afrag[2][S];
                    auto load_frags = [&](auto ic, v4i *dst) {
                        constexpr int i = decltype(ic)::value, s = i / 3, rt = i % 3;
                        static_for<0, S>([&](auto dc) {
                            constexpr int d = decltype(dc)::value;
                            if constexpr (rt < 2) {
                                dst[d] = wmainL[((rt * S + d) * KS + s) * 64 + lane];
                            } else {
                                dst[d] = wlocL[(d * KS + s) * 64 + lane];
                            }
                        });
                    };
                    load_frags(std::integral_constant<int, 0>{}, Afrag[0]);
                    v4i blo, bhi;
                    static_for<0, KS * 3>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, s = i / 3, rt = i % 3, stg = j * KS + s;
                        if constexpr (rt == 0) {
                            // slices of this k-step's 8 complex samples; then its raw registers take the k-step PF ahead
                            const uint32_t *cur = raw[stg & 3];
                            if constexpr (ITEM == 2) {
                                const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
#pragma unroll
                                for (int w = 0; w < 4; ++w) {
                                    blo[w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], sel_lo) ^ W8Enc<DT>::xlo);
                                    bhi[w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], sel_hi) ^ W8Enc<DT>::xhi);
                                }
                            } else {
#pragma unroll
                                for (int w = 0; w < 4; ++w) { blo[w] = (int) (cur[w] ^ W8Enc<DT>::xlo); bhi[w] = 0; }
                            }
                            if constexpr (stg + PF < NSTEP) {
                                issue(chunk_raw, std::integral_constant<int, stg + PF>{});
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (i + 1 < KS * 3) load_frags(std::integral_constant<int, i + 1>{}, Afrag[(i + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<0, S>([&](auto dc) {
                            constexpr int d = decltype(dc)::value;
                            const v4i A = Afrag[i & 1][d];
                            if constexpr (NSL == 2) {
                                // digit d x high slice -> level d+1 (accumulator d); digit d x low slice -> level d (accumulator d-1)
                                if constexpr (s == 0) acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                else acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi, acc[rt][d], 0, 0, 0);
                                if constexpr (d >= 1) acc[rt][d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, acc[rt][d - 1], 0, 0, 0);
                            } else {
                                if constexpr (s == 0) acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                else acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, acc[rt][d], 0, 0, 0);
                            }
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    // the next item's first k-steps go out now (the ring is free), if this was the item's last tile
                    if constexpr (j == TPW - 1) {
                        if (next_raw) static_for<0, PF>([&](auto sc) { issue(next_raw, sc); });
                    }
                    // int32 levels -> fp64: value = cs * sum_q 256^q acc[q]
                    double val[8], locv[4];
                    static_for<0, 3>([&](auto rc) {
                        constexpr int rt = decltype(rc)::value;
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            double dsum = (double) acc[rt][S - 1][v];
#pragma unroll
                            for (int q = S - 2; q >= 0; --q) dsum = fma(dsum, 256.0, (double) acc[rt][q][v]);
                            if constexpr (rt < 2) val[4 * rt + v] = dsum * cs_main; else locv[v] = dsum * cs_loc;
                        }
                    });
                    // lane (g, nb): mode g of column nb: rows 4g+v = 2*stream + part
                    cd Cr = mk(val[0], val[1]), Ci = mk(val[2], val[3]), Ar = mk(val[4], val[5]), Ai = mk(val[6], val[7]);
                    if (nco) {
                        const cd psi = nco_blk_g[tile * SDRT_TB + nb];   // the column's common phasor multiplies the complex DATA
                        cd t;
                        t = mk(psi.re * Cr.re - psi.im * Ci.re, psi.re * Cr.im - psi.im * Ci.im);
                        Ci = mk(psi.im * Cr.re + psi.re * Ci.re, psi.im * Cr.im + psi.re * Ci.im);
                        Cr = t;
                        t = mk(psi.re * Ar.re - psi.im * Ai.re, psi.re * Ar.im - psi.im * Ai.im);
                        Ai = mk(psi.im * Ar.re + psi.re * Ai.re, psi.im * Ar.im + psi.re * Ai.im);
                        Ar = t;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {      // local rows of this lane: outputs 2g, 2g+1, (stream re, stream im) each
                            const double t2 = psi.re * locv[2 * h] - psi.im * locv[2 * h + 1];
                            locv[2 * h + 1] = psi.im * locv[2 * h] + psi.re * locv[2 * h + 1];
                            locv[2 * h] = t2;
                        }
                    }
                    // inclusive scans over the 16 columns of the tile (lanes of one DPP row)
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
                    // tile aggregates: causal inclusive at the last column -> slot tile+1, anticausal inclusive at the first -> slot tile
                    if (nb == 15) { cf[(tile + 1) * 8 + g * 2 + 0] = Cr; cf[(tile + 1) * 8 + g * 2 + 1] = Ci; }
                    if (nb == 0) { cg[tile * 8 + g * 2 + 0] = Ar; cg[tile * 8 + g * 2 + 1] = Ai; }
                    // states at the column's edges from inside the tile: causal at its start, anticausal at its end
                    const cd Cxr = dpp_row<0x111>(Cr), Cxi = dpp_row<0x111>(Ci), Axr = dpp_row<0x101>(Ar), Axi = dpp_row<0x101>(Ai);
                    // this mode's share of every output of the column, plus the local rows where this lane holds them
                    double part[NOUT][2];
#pragma unroll
                    for (int b = 0; b < NOUT; ++b) {
                        const cd bc = bpL[b * 4 + g], ba = bpL[(NOUT - b) * 4 + g];
                        part[b][0] = bc.re * Cxr.re - bc.im * Cxr.im + ba.re * Axr.re - ba.im * Axr.im;
                        part[b][1] = bc.re * Cxi.re - bc.im * Cxi.im + ba.re * Axi.re - ba.im * Axi.im;
                        const double mine = (g == (b >> 1)) ? 1.0 : 0.0;
                        part[b][0] = fma(mine, locv[2 * (b & 1)], part[b][0]);
                        part[b][1] = fma(mine, locv[2 * (b & 1) + 1], part[b][1]);
                    }
                    // sum over the four modes (lanes 16 apart) as a reduce-scatter: lane g ends up with outputs g (+4)
#pragma unroll
                    for (int e = 0; e < OPL; ++e) {
                        const int b0 = 4 * e;
                        const double r0 = swap32_add(part[b0][0], part[b0 + 2][0]), i0 = swap32_add(part[b0][1], part[b0 + 2][1]);
                        const double r1 = swap32_add(part[b0 + 1][0], part[b0 + 3][0]), i1 = swap32_add(part[b0 + 1][1], part[b0 + 3][1]);
                        yq[1][j][e] = mk(swap16_add(r0, r1), swap16_add(i0, i1));
                    }
                });
            }
            __syncthreads();
            bcur = bcur == 2 ? 0 : bcur + 1;
        }
    }
}

// permlane swap self-test: out[0][l] = swap32_add(x[l], y[l]), out[1][l] = swap16_add(x[l], y[l])
__global__ void swap_selftest_kernel(const double *x, const double *y, double *out) {
    const int l = threadIdx.x & 63;
    out[l] = swap32_add(x[l], y[l]);
    out[64 + l] = swap16_add(x[l], y[l]);
}

// ---- launch helpers -----------------------------------------------------------------------------------------------
bool chain_i8w_supported(int q, int dt, int n) {
    if (!(q == 32 || q == 64)) return false;
    if (!(dt == SDRT_S8 || dt == SDRT_U8 || dt == SDRT_S16 || dt == SDRT_U16)) return false;
    const int nt = n / (SDRT_TB * SDRT_WIDE_LC);
    return n % (SDRT_TB * SDRT_WIDE_LC) == 0 && (nt == 8 || nt == 16);
}

size_t chain_i8w_lds_bytes(int q, int n) {
    return (size_t) wide_lds(SDRT_I8_DIGITS, SDRT_WIDE_LC / q, n / (SDRT_TB * SDRT_WIDE_LC)).total;
}

template <int DT, int NOUT, int NT>
static hipError_t launch_i8w_k(const DecimArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void *) chain_i8w_kernel<DT, NOUT, NT, SDRT_I8_DIGITS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((chain_i8w_kernel<DT, NOUT, NT, SDRT_I8_DIGITS>), grid, dim3(512), lds, s, a);
    return hipGetLastError();
}

template <int DT>
static hipError_t launch_i8w_d(const DecimArgs &a, int nout, int nt, dim3 grid, size_t lds, hipStream_t s) {
    if (nout == 4 && nt == 8) return launch_i8w_k<DT, 4, 8>(a, grid, lds, s);
    if (nout == 4 && nt == 16) return launch_i8w_k<DT, 4, 16>(a, grid, lds, s);
    if (nout == 8 && nt == 8) return launch_i8w_k<DT, 8, 8>(a, grid, lds, s);
    if (nout == 8 && nt == 16) return launch_i8w_k<DT, 8, 16>(a, grid, lds, s);
    return hipErrorInvalidValue;
}

hipError_t launch_chain_i8w(const DecimArgs &a, int q, int dt, size_t n_chunks, int n_cus, hipStream_t s) {
    const int nout = SDRT_WIDE_LC / q, nt = a.n / (SDRT_TB * SDRT_WIDE_LC);
    // one workgroup per CU, in multiples of 8 (one per XCD); fewer when there is less work than CUs
    size_t items = n_chunks * (size_t) a.rows;
    size_t wgs = (size_t) (n_cus > 8 ? n_cus : 8);
    if (items < wgs) wgs = items;
    wgs = (wgs + 7) / 8 * 8;
    const dim3 grid((unsigned) wgs);
    const size_t lds = chain_i8w_lds_bytes(q, a.n);
    switch (dt) {
        case SDRT_S8: return launch_i8w_d<SDRT_S8>(a, nout, nt, grid, lds, s);
        case SDRT_U8: return launch_i8w_d<SDRT_U8>(a, nout, nt, grid, lds, s);
        case SDRT_S16: return launch_i8w_d<SDRT_S16>(a, nout, nt, grid, lds, s);
        case SDRT_U16: return launch_i8w_d<SDRT_U16>(a, nout, nt, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_swap_selftest(const double *x, const double *y, double *out, hipStream_t s) {
    hipLaunchKernelGGL(swap_selftest_kernel, dim3(1), dim3(64), 0, s, x, y, out);
    return hipGetLastError();
}
