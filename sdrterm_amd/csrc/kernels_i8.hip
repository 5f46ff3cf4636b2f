// chain_i8_kernel: the fused chain with stage A on the int8 matrix cores (v_mfma_i32_32x32x32_i8), for integer encodings.
//
// Idea (error-free splitting, "Ozaki scheme"): the samples are small integers, so the block sums of stage A
//     v_b[r] = sum_k W[r][k] x_b[k]          (32 real rows x 2L real columns, NCO phasor folded into W)
// can be computed exactly in integer arithmetic once W is quantised: W ~ c * sum_d 256^d D_d with S signed base-256
// digit planes D_d, and x = 256*hi + lo' + offset with both slices in [-128, 127].  Every digit-plane x slice product
// is an int8 GEMM with exact int32 accumulation (|sum| <= 2L*128*128 = 2^21 per plane); planes of equal weight share an
// accumulator, and one fp64 recombination per output restores v_b to 2^-(8S) relative accuracy (S = 5: 2e-12 measured).
// The constant `offset` never enters the matrix product: the decimator is linear, so its response to the constant is
// a fixed vector per row, precomputed once with the f64 kernel and added to the decimated samples (ybias).
// What follows stage A (block phasor, in-tile scans, carries, demodulation, output filter) is the f64 code of
// kernels_common.hpp; only 1/16 of the input rate reaches it.
#include "kernels_common.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

template <int DT> struct I8Enc;   // slices per component, what each slice XORs with, the constant left out
template <> struct I8Enc<SDRT_S8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0u, xhi = 0u; static constexpr double off = 0.0; };
template <> struct I8Enc<SDRT_U8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; static constexpr double off = 128.0; };
template <> struct I8Enc<SDRT_S16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; static constexpr double off = 128.0; };
template <> struct I8Enc<SDRT_U16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0x80808080u; static constexpr double off = 32896.0; };

}  // namespace

// Lane l = (n = l & 31: column within the 32-column super-tile, h = l >> 5: which half of the column's samples).
// A column is G consecutive blocks of L samples (LC = G L samples): the matrix cores deliver, per column,
//   pass 0: the 32 aggregate rows.  Row r of the 32x32 result (r = (v & 3) + 8 (v >> 2) + 4 h for accumulator register v):
//           part (v & 1) | stream ((v >> 1) & 1) | mode>>1 (= h) | kind ((v >> 2) & 1) | mode&1 (v >> 3),
//           so every lane holds complete complex numbers for both streams of two modes (2h, 2h+1), both kinds;
//   pass 1 (G > 1): per internal block boundary g the column-local part of that boundary's output (rows stream + 2(g-1),
//           landing in registers v = 0..2(G-1)-1 of the h = 0 lanes).
// The in-tile scans then run over columns instead of blocks, which divides the fp64 VALU work per sample by G.
template <int L, int DT, int S, int G>
__global__ __launch_bounds__(256, 2) void chain_i8_kernel(DecimArgs a) {
    constexpr int ITEM = Enc<DT>::size;
    constexpr int NSL = I8Enc<DT>::slices;
    constexpr int LC = L * G;                      // samples per column
    constexpr int KS8 = LC / 16;                   // MFMA k-steps per pass (32 int8 of K each)
    constexpr int RPK = 8 * 2 * ITEM / 4;          // raw dwords per k-step (8 samples): 8 for 16-bit, 4 for 8-bit items
    constexpr int RING = KS8 >= 4 ? 4 : KS8;       // k-steps in flight (loads issued RING k-steps ahead of their use)
    constexpr int NLV = S + NSL - 1;               // accumulator levels (weight 256^level)
    constexpr int ST = 32 * LC;                    // samples per super-tile
    constexpr int NP = G > 1 ? 2 : 1;              // passes over the super-tile's data
    constexpr int NR = 2 * (G - 1);                // useful rows of pass 1
    static_assert(KS8 % RING == 0, "ring must divide the k-steps");
    static_assert(G == 1 || G == 2, "columns of 1 or 2 blocks");
    typedef typename Enc<DT>::raw raw_t;

    extern __shared__ __align__(16) unsigned char smem[];
    const DecimTables *__restrict__ tab = a.tab;
    const int n = a.n, m = a.m, nt = a.n_tiles, nst = nt / 2;
    const unsigned bid = blockIdx.x;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const long chunk = (long) (slot / (unsigned) a.rows) * 8 + xcd;
    const int row = (int) (slot % (unsigned) a.rows);
    if (chunk >= (long) a.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, nb = lane & 15, tsel = (lane >> 4) & 1;
    const bool swap = a.swap != 0;
    const bool nco = a.nco_tile != nullptr;

    unsigned long long *stp = (a.stamps && (int) bid < a.stamp_wgs && lane == 0) ? a.stamps + ((size_t) bid * 4 + wave) * 16 : nullptr;
#define SDRT_STAMP(i) do { if (stp) stp[i] = __builtin_amdgcn_s_memtime(); } while (0)
    SDRT_STAMP(0);

    const LdsLayout lo = lds_layout(n, L, m, G);
    cd *base = (cd *) smem;
    cd *o_lds = base + lo.o, *carF = base + lo.carF, *carG = base + lo.carG;
    cd *mupL = base + lo.total;                              // [4 modes][4 powers] column multipliers
    cd *mhL = mupL + 16;                                     // [4 modes] p^L (carry of a column-start state to its mid boundary)
    v4i *w8L = (v4i *) (mhL + 4);                            // [S][KS8][64] main A operands
    v4i *wmL = w8L + S * KS8 * 64;                           // [S][KS8][2][NR] mid A operands (only NR rows are non-zero)

    const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk * (size_t) n * 2 * ITEM;
    const cd *nco_samp_g = nco ? (const cd *) a.nco_tile + (size_t) row * LC : nullptr;          // exp(i theta t), t < LC
    const cd *nco_blk_g = nco ? (const cd *) a.nco_phase + (size_t) row * (size_t) (n / LC) : nullptr;   // per column
    const double xoff = a.x_offset;

    // ---- prologue: this row's digit planes and the scan multipliers into LDS -------------------------------------------
    {
        const v4i *wg = (const v4i *) (a.w8 + (size_t) row * (size_t) (S * KS8 * 64 * 16));
        for (int i = tid; i < S * KS8 * 64; i += 256) w8L[i] = wg[i];
        if constexpr (G > 1) {
            const v4i *wm = (const v4i *) (a.w8mid + (size_t) row * (size_t) (S * KS8 * 2 * NR * 16));
            for (int i = tid; i < S * KS8 * 2 * NR; i += 256) wmL[i] = wm[i];
        }
        if (tid < 16) mupL[tid] = mk(tab->mu_pow[tid & 3][tid >> 2][0], tab->mu_pow[tid & 3][tid >> 2][1]);   // [mode][power]
        if (tid >= 64 && tid < 68) mhL[tid - 64] = mk(tab->cF[1][tid - 64][0], tab->cF[1][tid - 64][1]);
    }
    const double d0p = tab->d0p;
    const double cs = a.w8_combine, csm = a.w8mid_combine;

    // raw samples of this lane: [col*LC + h*LC/2, +LC/2) of each super-tile, streamed k-step by k-step through a small ring
    uint32_t ring[RING][RPK];
    const size_t lane_off = ((size_t) (lane & 31) * LC + (size_t) hh * (LC / 2)) * 2 * ITEM;
    auto load_step = [&](int st, int s, uint32_t *dst) {     // the 8 samples of k-step s
        load_words<RPK * 4>(chunk_raw + (size_t) st * ST * 2 * ITEM + lane_off + (size_t) s * RPK * 4, dst);
    };
    if (wave < nst) {
#pragma unroll
        for (int s = 0; s < RING; ++s) load_step(wave, s, ring[s]);
    }
    __syncthreads();
    SDRT_STAMP(1);

    for (int st = wave; st < nst; st += 4) {
        const bool more = st + 4 < nst;
        double val[16];                 // pass 0: the lane's 16 aggregate values
        double midv[NR > 0 ? NR : 1];   // pass 1: column-local parts of the internal boundaries (valid in h == 0 lanes)
        double u0r = 0.0, u0i = 0.0;
        static_for<0, NP>([&](auto pc) {
            constexpr int pass = decltype(pc)::value;
            v16i acc[NLV];
#pragma unroll
            for (int q = 0; q < NLV; ++q)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[q][v] = 0;
            static_for<0, KS8>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                uint32_t *r = ring[s % RING];
                v4i blo, bhi;
                if constexpr (ITEM == 2) {
                    // dword = [re.lo re.hi im.lo im.hi] (little-endian records; swapped records exchange the selectors)
                    const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
                    if constexpr (s == 0 && pass == 0) {
                        const uint16_t re = (uint16_t) (r[0] & 0xffffu), im = (uint16_t) (r[0] >> 16);
                        u0r = decode<DT>(re, swap) - xoff;
                        u0i = decode<DT>(im, swap) - xoff;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        blo[i] = (int) (__builtin_amdgcn_perm(r[2 * i + 1], r[2 * i], sel_lo) ^ I8Enc<DT>::xlo);
                        bhi[i] = (int) (__builtin_amdgcn_perm(r[2 * i + 1], r[2 * i], sel_hi) ^ I8Enc<DT>::xhi);
                    }
                } else {
                    if constexpr (s == 0 && pass == 0) {
                        u0r = decode<DT>((raw_t) (r[0] & 0xffu), swap) - xoff;
                        u0i = decode<DT>((raw_t) ((r[0] >> 8) & 0xffu), swap) - xoff;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) blo[i] = (int) (r[i] ^ I8Enc<DT>::xlo);
                }
                // refill this ring slot with the k-step that is RING steps ahead in (pass, k-step, super-tile) order
                if constexpr (s + RING < KS8) {
                    load_step(st, s + RING, r);
                } else if constexpr (pass + 1 < NP) {
                    load_step(st, s + RING - KS8, r);                 // pass 1 re-reads the super-tile (L1/L2 hits)
                } else {
                    if (more) load_step(st + 4, s + RING - KS8, r);
                }
                // digit planes of this k-step; all `lo` products first, then all `hi` products, so that two matrix
                // instructions on the same accumulator are S instructions apart (no back-to-back dependency stalls)
                v4i A[S];
                static_for<0, S>([&](auto dc) {
                    constexpr int d = decltype(dc)::value;
                    if constexpr (pass == 0) {
                        A[d] = w8L[(d * KS8 + s) * 64 + lane];
                    } else {
                        A[d] = (v4i) {0, 0, 0, 0};
                        if ((lane & 31) < NR) A[d] = wmL[((d * KS8 + s) * 2 + hh) * NR + (lane & 31)];
                    }
                });
                static_for<0, S>([&](auto dc) {
                    constexpr int d = decltype(dc)::value;
                    acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[d], blo, acc[d], 0, 0, 0);
                });
                if constexpr (NSL == 2) {
                    static_for<0, S>([&](auto dc) {
                        constexpr int d = decltype(dc)::value;
                        acc[d + 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[d], bhi, acc[d + 1], 0, 0, 0);
                    });
                }
            });
            // recombine the levels: value = c * sum_q 256^q acc[q], pairs of levels first in int32 (|acc| <= 2^22)
            if (st == wave) { if constexpr (pass == 0) SDRT_STAMP(8); else SDRT_STAMP(10); }
            static_for<0, (pass == 0 ? 16 : NR)>([&](auto vc) {
                constexpr int v = decltype(vc)::value;
                double dsum = 0.0;
                static_for<0, (NLV + 1) / 2>([&](auto pp) {
                    constexpr int p = (NLV + 1) / 2 - 1 - decltype(pp)::value;     // highest pair first
                    int t = acc[2 * p][v];
                    if constexpr (2 * p + 1 < NLV) t += acc[2 * p + 1][v] * 256;
                    dsum = fma(dsum, 65536.0, (double) t);
                });
                if constexpr (pass == 0) val[v] = dsum * cs; else midv[v] = dsum * csm;
            });
            if (st == wave) { if constexpr (pass == 0) SDRT_STAMP(9); else SDRT_STAMP(11); }
        });

        // complex numbers of this lane: index q = kind + 2*(mode & 1); stream re = (val[4q], val[4q+1]), im = (+2, +3)
        cd Sr[4], Si[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { Sr[q] = mk(val[4 * q], val[4 * q + 1]); Si[q] = mk(val[4 * q + 2], val[4 * q + 3]); }
        const int col = st * 32 + (lane & 31);
        if (nco) {
            const cd psi = nco_blk_g[col];          // the column's common phasor multiplies the complex DATA (mixes the streams)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const cd t = mk(psi.re * Sr[q].re - psi.im * Si[q].re, psi.re * Sr[q].im - psi.im * Si[q].im);
                Si[q] = mk(psi.im * Sr[q].re + psi.re * Si[q].re, psi.im * Sr[q].im + psi.re * Si[q].im);
                Sr[q] = t;
            }
            const double tr = psi.re * u0r - psi.im * u0i;
            u0i = psi.im * u0r + psi.re * u0i;
            u0r = tr;
            if constexpr (G > 1) {
                const double t2 = psi.re * midv[0] - psi.im * midv[1];
                midv[1] = psi.im * midv[0] + psi.re * midv[1];
                midv[0] = t2;
            }
        }
        // ---- in-tile inclusive scans over the 16 columns of a DPP row: causal for even q, anticausal for odd q ---------
        static_for<0, 4>([&](auto kc) {
            constexpr int k = decltype(kc)::value;            // step distance 1 << k
            const cd m0 = mupL[(2 * hh + 0) * 4 + k], m1 = mupL[(2 * hh + 1) * 4 + k];
            Sr[0] = cfma(m0, dpp_row<0x110 + (1 << k)>(Sr[0]), Sr[0]);
            Si[0] = cfma(m0, dpp_row<0x110 + (1 << k)>(Si[0]), Si[0]);
            Sr[1] = cfma(m0, dpp_row<0x100 + (1 << k)>(Sr[1]), Sr[1]);
            Si[1] = cfma(m0, dpp_row<0x100 + (1 << k)>(Si[1]), Si[1]);
            Sr[2] = cfma(m1, dpp_row<0x110 + (1 << k)>(Sr[2]), Sr[2]);
            Si[2] = cfma(m1, dpp_row<0x110 + (1 << k)>(Si[2]), Si[2]);
            Sr[3] = cfma(m1, dpp_row<0x100 + (1 << k)>(Sr[3]), Sr[3]);
            Si[3] = cfma(m1, dpp_row<0x100 + (1 << k)>(Si[3]), Si[3]);
        });
        if (st == wave) SDRT_STAMP(12);
        const int tile = st * 2 + tsel;
        if (nb == 15) {
            carF[(tile + 1) * 8 + (2 * hh + 0) * 2 + 0] = Sr[0]; carF[(tile + 1) * 8 + (2 * hh + 0) * 2 + 1] = Si[0];
            carF[(tile + 1) * 8 + (2 * hh + 1) * 2 + 0] = Sr[2]; carF[(tile + 1) * 8 + (2 * hh + 1) * 2 + 1] = Si[2];
        }
        if (nb == 0) {
            carG[tile * 8 + (2 * hh + 0) * 2 + 0] = Sr[1]; carG[tile * 8 + (2 * hh + 0) * 2 + 1] = Si[1];
            carG[tile * 8 + (2 * hh + 1) * 2 + 0] = Sr[3]; carG[tile * 8 + (2 * hh + 1) * 2 + 1] = Si[3];
        }
        // output at the column start: direct term + 2 Re(exclusive causal + inclusive anticausal) over the 4 modes
        // (2 of them in this lane, 2 in lane ^ 32)
        const cd Cx0r = dpp_row<0x111>(Sr[0]), Cx0i = dpp_row<0x111>(Si[0]);     // causal exclusive, modes 2h / 2h+1
        const cd Cx1r = dpp_row<0x111>(Sr[2]), Cx1i = dpp_row<0x111>(Si[2]);
        double pr = 2.0 * (Cx0r.re + Cx1r.re + Sr[1].re + Sr[3].re);
        double pi = 2.0 * (Cx0i.re + Cx1i.re + Si[1].re + Si[3].re);
        if (hh == 0) { pr = fma(d0p, u0r, pr); pi = fma(d0p, u0i, pi); }
        pr += __shfl_xor(pr, 32);
        pi += __shfl_xor(pi, 32);
        if (hh == 0) o_lds[(tile * SDRT_TB + nb) * G] = mk(pr, pi);
        if constexpr (G > 1) {
            // output at the mid boundary: column-local part (pass 1) + the states at the column's edges carried by p^L:
            //   2 Re sum_i p_i^L ( causal state at the column start + anticausal state at the column end )
            const cd Ax0r = dpp_row<0x101>(Sr[1]), Ax0i = dpp_row<0x101>(Si[1]);     // anticausal exclusive
            const cd Ax1r = dpp_row<0x101>(Sr[3]), Ax1i = dpp_row<0x101>(Si[3]);
            const cd h0 = mhL[2 * hh + 0], h1 = mhL[2 * hh + 1];
            double qr = 2.0 * (h0.re * (Cx0r.re + Ax0r.re) - h0.im * (Cx0r.im + Ax0r.im) +
                               h1.re * (Cx1r.re + Ax1r.re) - h1.im * (Cx1r.im + Ax1r.im));
            double qi = 2.0 * (h0.re * (Cx0i.re + Ax0i.re) - h0.im * (Cx0i.im + Ax0i.im) +
                               h1.re * (Cx1i.re + Ax1i.re) - h1.im * (Cx1i.im + Ax1i.im));
            if (hh == 0) { qr += midv[0]; qi += midv[1]; }
            qr += __shfl_xor(qr, 32);
            qi += __shfl_xor(qi, 32);
            if (hh == 0) o_lds[(tile * SDRT_TB + nb) * G + 1] = mk(qr, qi);
        }
        if (st == wave) SDRT_STAMP(13);
    }
    SDRT_STAMP(2);
#undef SDRT_STAMP
    auto edge = [&](long idx) -> cd {
        const raw_t *rp = (const raw_t *) chunk_raw + 2 * idx;
        double xr = decode<DT>(rp[0], swap) - xoff, xi = decode<DT>(rp[1], swap) - xoff;
        if (nco) {
            const cd u = cmul(mk(xr, xi), cmul(nco_samp_g[idx % LC], nco_blk_g[idx / LC]));
            xr = u.re;
            xi = u.im;
        }
        return mk(xr, xi);
    };
    chain_tail(a, L, row, chunk, smem, stp, edge, a.ybias ? (const cd *) a.ybias + (size_t) row * m : (const cd *) nullptr);
}

// MFMA i8 32x32x32 self-test: D(32x32) = A(32x32) * B(32x32), operands packed as this file assumes
// (lane = row/col + 32 h, byte j of the lane's 16 <-> inner index 16 h + j; result row (v&3) + 8 (v>>2) + 4 h, col lane & 31).
__global__ void i8_selftest_kernel(const int *A, const int *B, int *D) {
    const int lane = threadIdx.x & 63, rc = lane & 31, h = lane >> 5;
    v4i av, bv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned pa = 0, pb = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = 16 * h + 4 * i + b;
            pa |= ((unsigned) (A[rc * 32 + k] & 0xff)) << (8 * b);
            pb |= ((unsigned) (B[k * 32 + rc] & 0xff)) << (8 * b);
        }
        av[i] = (int) pa;
        bv[i] = (int) pb;
    }
    v16i acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; ++v) D[((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + rc] = acc[v];
}

// ---- launch helpers -----------------------------------------------------------------------------------------------
bool chain_i8_supported(int L, int dt, int n) {
    // super-tiles of 32 columns of G blocks; the data ring needs at least two k-steps per pass
    return (L == 16 || L == 32 || L == 64) && (dt == SDRT_S8 || dt == SDRT_U8 || dt == SDRT_S16 || dt == SDRT_U16) &&
           n % (32 * SDRT_I8_G * L) == 0;
}

size_t chain_i8_lds_bytes(int n, int L, int m) {
    const int G = SDRT_I8_G, KS = L * G / 16, NR = 2 * (G - 1);
    return ((size_t) lds_layout(n, L, m, G).total + 20) * 16 + (size_t) SDRT_I8_DIGITS * KS * (64 + 2 * NR) * 16;
}

template <int L, int DT>
static hipError_t launch_i8_k(const DecimArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) chain_i8_kernel<L, DT, SDRT_I8_DIGITS, SDRT_I8_G>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((chain_i8_kernel<L, DT, SDRT_I8_DIGITS, SDRT_I8_G>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int L>
static hipError_t launch_i8_l(const DecimArgs &a, int dt, dim3 grid, size_t lds, hipStream_t s) {
    switch (dt) {
        case SDRT_S8: return launch_i8_k<L, SDRT_S8>(a, grid, lds, s);
        case SDRT_U8: return launch_i8_k<L, SDRT_U8>(a, grid, lds, s);
        case SDRT_S16: return launch_i8_k<L, SDRT_S16>(a, grid, lds, s);
        case SDRT_U16: return launch_i8_k<L, SDRT_U16>(a, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_chain_i8(const DecimArgs &a, int L, int dt, bool nco, size_t n_chunks, hipStream_t s) {
    (void) nco;
    const size_t groups = (n_chunks + 7) / 8;
    const dim3 grid((unsigned) (groups * 8 * (size_t) a.rows));
    const size_t lds = chain_i8_lds_bytes(a.n, L, a.m);
    switch (L) {
        case 16: return launch_i8_l<16>(a, dt, grid, lds, s);
        case 32: return launch_i8_l<32>(a, dt, grid, lds, s);
        case 64: return launch_i8_l<64>(a, dt, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_i8_selftest(const int *A, const int *B, int *D, hipStream_t s) {
    hipLaunchKernelGGL(i8_selftest_kernel, dim3(1), dim3(64), 0, s, A, B, D);
    return hipGetLastError();
}
