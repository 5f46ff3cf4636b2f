// chain_i8t_kernel: the chain kernel with stage A on the int8 matrix cores, v_mfma_i32_16x16x64_i8 form.
// Same lane mapping, scans and tail as the f64 kernel (kernels.hip): lane (g, nb) = (sample quarter / mode, block of
// the tile).  Only the stage-A product differs: the lane's L/4 samples are split into int8 slices (x = 256 hi + lo' +
// offset) and multiplied with S signed base-256 digit planes of the weights (NCO phasor folded in) -- exact int32
// accumulation, one fp64 recombination per output.
#include "kernels_common.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));

namespace {
template <int DT> struct I8Enc;   // slices per component, what each slice XORs with
template <> struct I8Enc<SDRT_S8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0u, xhi = 0u; };
template <> struct I8Enc<SDRT_U8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };
template <> struct I8Enc<SDRT_S16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };
template <> struct I8Enc<SDRT_U16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0x80808080u; };
// float32 records: every matrix-core column (L samples) is taken as block fixed point against the column's own exponent:
// i = trunc(x 2^(30 - E)), |x| < 2^E for all of the column's samples; the four bytes of i are the four slices (the lower
// three made signed by the XOR; the constant that leaves out comes back as a rank-one term, DecimArgs.wsum).  What is lost
// is below 2^-30 of the column's largest sample.
template <> struct I8Enc<SDRT_F32> { static constexpr int slices = 4; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };
}  // namespace

template <int LB, int DT, int S, int G>
__global__ __launch_bounds__(256, (G > 1 || DT == SDRT_F32 ? 2 : 3)) void chain_i8t_kernel(DecimArgs a) {
    // LB = block length (output spacing), a matrix-core column is G consecutive blocks: L = G * LB samples
    constexpr int L = LB * G;
    constexpr int NSL = I8Enc<DT>::slices;
    constexpr int KS64 = L / 32;                    // 64-deep MFMA k-steps per column (2 components x 8 samples x 4 lane groups)
    constexpr bool F32 = DT == SDRT_F32;
    // digit d x slice s adds to level d + s (weight 256^level); float32: levels below LMIN are under the slices' own
    // truncation step and are not computed (14 of the 20 products remain)
    constexpr int LMIN = F32 ? 3 : 0;
    constexpr int NLV = S + NSL - 1 - LMIN;         // accumulator levels kept
    constexpr int NR = 2 * (G - 1);                 // rows of the compact mid tile
    static_assert(G == 1 || G == 2, "columns of 1 or 2 blocks");
    const bool nco = a.nco_tile != nullptr;
    constexpr int KS = L / 4;                       // samples per lane per tile
    constexpr int ITEM = Enc<DT>::size;
    constexpr int BYTES = KS * 2 * ITEM;            // per lane per tile
    constexpr int TILE = SDRT_TB * L;               // samples per tile
    constexpr int NW = BYTES < 4 ? 1 : BYTES / 4;    // 32-bit words per lane per tile
    typedef typename Enc<DT>::raw raw_t;

    extern __shared__ __align__(16) unsigned char smem[];
    const DecimTables *__restrict__ tab = a.tab;
    const int n = a.n, m = a.m, nt = a.n_tiles;
    // (chunk, row) from the block id: all rows of a chunk get ids that share an XCD (ids are dealt round-robin
    // over 8 XCDs), so the chunk is pulled through one L2 only.  Placement changes speed only, never results.
    const unsigned bid = blockIdx.x;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const long chunk = (long) (slot / (unsigned) a.rows) * 8 + xcd;
    const int row = (int) (slot % (unsigned) a.rows);
    if (chunk >= (long) a.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nb = lane & 15;
    const bool swap = a.swap != 0;

    // diagnostic phase stamps (a.stamps == nullptr in every production call)
    unsigned long long *stp = (a.stamps && (int) bid < a.stamp_wgs && lane == 0) ? a.stamps + ((size_t) bid * 4 + wave) * 16 : nullptr;
#define SDRT_STAMP(i) do { if (stp) stp[i] = __builtin_amdgcn_s_memtime(); } while (0)
    SDRT_STAMP(0);

    const LdsLayout lo = lds_layout(n, LB, m, G);
    cd *base = (cd *) smem;
    cd *o_lds = base + lo.o, *carF = base + lo.carF, *carG = base + lo.carG;

    const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk * (size_t) n * 2 * ITEM;
    const double xoff = a.x_offset;
    const cd *nco_samp_g = nco ? (const cd *) a.nco_tile + (size_t) row * L : nullptr;              // exp(i theta j), j < L
    const cd *nco_blk_g = nco ? (const cd *) a.nco_phase + (size_t) row * (size_t) (n / L) : nullptr;   // exp(i theta L B), B < n/L

    // digit planes of this row -> LDS, in A-operand order [tile][digit][kstep][lane]; compact mid tile [digit][kstep][kg][NR]
    v4i *w8L = (v4i *) (base + lo.total);
    v4i *wmL = w8L + 2 * S * KS64 * 64;
    {
        const v4i *wg = (const v4i *) (a.w8 + (size_t) row * (size_t) (2 * S * KS64 * 64 * 16));
        for (int i = tid; i < 2 * S * KS64 * 64; i += 256) w8L[i] = wg[i];
        if constexpr (G > 1) {
            const v4i *wm = (const v4i *) (a.w8mid + (size_t) row * (size_t) (S * KS64 * 4 * NR * 16));
            for (int i = tid; i < S * KS64 * 4 * NR; i += 256) wmL[i] = wm[i];
        }
    }
    cd mh = mk(1.0, 0.0);                            // p_g^LB: carries a column-edge state to the mid boundary (G = 2)
    if constexpr (G > 1) mh = mk(tab->cF[1][g][0], tab->cF[1][g][1]);
    const double csm = a.w8mid_combine;
    cd mup[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mup[k] = mk(tab->mu_pow[k][g][0], tab->mu_pow[k][g][1]);
    const double d0p = tab->d0p;
    const double cs = a.w8_combine;
    double wsum[8];                                  // float32: sum of this lane's eight weight rows (both row tiles), see I8Enc<SDRT_F32>
#pragma unroll
    for (int v = 0; v < 8; ++v) wsum[v] = F32 ? a.wsum[(size_t) row * 32 + (v >> 2) * 16 + 4 * g + (v & 3)] : 0.0;

    uint32_t cur[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) cur[i] = 0;
    const size_t lane_off = ((size_t) nb * L + (size_t) g * KS) * 2 * ITEM;
    if (wave < nt) load_words<BYTES>(chunk_raw + (size_t) wave * TILE * 2 * ITEM + lane_off, cur);

    __syncthreads();
    SDRT_STAMP(1);
    // ---- main loop over this wave's tiles, next tile's bytes prefetched ----------------------------------------------
    for (int tile = wave; tile < nt; tile += 4) {
        cd psi = mk(1.0, 0.0);
        if (nco) psi = nco_blk_g[tile * SDRT_TB + nb];   // phasor common to this lane's block

        // slices of this lane's L/4 samples: dword = [re.lo re.hi im.lo im.hi] -> 8 samples (16 int8) per k-step and slice
        double u0r, u0i;
        v4i blo[KS64], bhi[KS64];
        v4i bsl[F32 ? KS64 : 1][4];                  // float32: [k-step][slice]
        double colscale = 1.0;                       // float32: 2^(E - 30) of this lane's column
        if constexpr (F32) {
            // the column's exponent: largest magnitude over this lane's samples, then over the four lanes of the column
            unsigned mx = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const unsigned w = swap ? __builtin_bswap32(cur[i]) : cur[i];
                cur[i] = w;
                mx = max(mx, w & 0x7fffffffu);
            }
            mx = max(mx, (unsigned) __shfl_xor((int) mx, 16));
            mx = max(mx, (unsigned) __shfl_xor((int) mx, 32));
            const int e = (int) (mx >> 23) - 126;    // |x| < 2^e for every sample of the column (e = -126 for an all-zero column)
            colscale = mx == 0 ? 0.0 : __hiloint2double((1023 + e - 30) << 20, 0);     // a silent column contributes exact zeros
            u0r = (double) __uint_as_float(cur[0]);
            u0i = (double) __uint_as_float(cur[1]);
#pragma unroll
            for (int s = 0; s < KS64; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // four values (two samples) -> fixed point -> 4 x 4 byte transpose: one dword per slice
                    int q0 = (int) ldexpf(__uint_as_float(cur[16 * s + 4 * i + 0]), 30 - e);
                    int q1 = (int) ldexpf(__uint_as_float(cur[16 * s + 4 * i + 1]), 30 - e);
                    int q2 = (int) ldexpf(__uint_as_float(cur[16 * s + 4 * i + 2]), 30 - e);
                    int q3 = (int) ldexpf(__uint_as_float(cur[16 * s + 4 * i + 3]), 30 - e);
                    const unsigned a01 = __builtin_amdgcn_perm((unsigned) q1, (unsigned) q0, 0x05010400u);   // [q0.b0 q1.b0 q0.b1 q1.b1]
                    const unsigned a23 = __builtin_amdgcn_perm((unsigned) q3, (unsigned) q2, 0x05010400u);
                    const unsigned c01 = __builtin_amdgcn_perm((unsigned) q1, (unsigned) q0, 0x07030602u);   // [q0.b2 q1.b2 q0.b3 q1.b3]
                    const unsigned c23 = __builtin_amdgcn_perm((unsigned) q3, (unsigned) q2, 0x07030602u);
                    bsl[s][0][i] = (int) (__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
                    bsl[s][1][i] = (int) (__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
                    bsl[s][2][i] = (int) (__builtin_amdgcn_perm(c23, c01, 0x05040100u) ^ 0x80808080u);
                    bsl[s][3][i] = (int) __builtin_amdgcn_perm(c23, c01, 0x07060302u);
                }
        } else if constexpr (ITEM == 2) {
            const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
            u0r = decode<DT>((uint16_t) (cur[0] & 0xffffu), swap) - xoff;
            u0i = decode<DT>((uint16_t) (cur[0] >> 16), swap) - xoff;
#pragma unroll
            for (int s = 0; s < KS64; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    blo[s][i] = (int) (__builtin_amdgcn_perm(cur[8 * s + 2 * i + 1], cur[8 * s + 2 * i], sel_lo) ^ I8Enc<DT>::xlo);
                    bhi[s][i] = (int) (__builtin_amdgcn_perm(cur[8 * s + 2 * i + 1], cur[8 * s + 2 * i], sel_hi) ^ I8Enc<DT>::xhi);
                }
        } else {
            u0r = decode<DT>((raw_t) (cur[0] & 0xffu), swap) - xoff;
            u0i = decode<DT>((raw_t) ((cur[0] >> 8) & 0xffu), swap) - xoff;
#pragma unroll
            for (int s = 0; s < KS64; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) { blo[s][i] = (int) (cur[4 * s + i] ^ I8Enc<DT>::xlo); bhi[s][i] = 0; }
        }
        // stage A on the int8 matrix cores: per row tile (0 causal, 1 anticausal) S digit planes x slices, exact int32 sums
        // the slices are extracted: the raw words are free, start the next tile's loads right away (a full tile of lead)
        if (tile + 4 < nt) load_words<BYTES>(chunk_raw + (size_t) (tile + 4) * TILE * 2 * ITEM + lane_off, cur);
        double val[8], midv[NR > 0 ? NR : 1];
        // Stage A on the int8 matrix cores.  Steps = (row tile, k-step) pairs; the S digit-plane fragments of step i+1 are
        // read from LDS while the 2S matrix instructions of step i issue (two register sets, statically alternated).
        constexpr int NT = G > 1 ? 3 : 2, NSTEP = NT * KS64;
        v4i Afrag[2][S];
        auto load_frags = [&](auto ic, v4i *dst) {
            constexpr int i = decltype(ic)::value, rt = i / KS64, s = i % KS64;
            static_for<0, S>([&](auto dc) {
                constexpr int d = decltype(dc)::value;
                if constexpr (rt < 2) {
                    dst[d] = w8L[((rt * S + d) * KS64 + s) * 64 + lane];
                } else {
                    dst[d] = (v4i) {0, 0, 0, 0};
                    if (nb < NR) dst[d] = wmL[((d * KS64 + s) * 4 + g) * NR + nb];    // A operand: lane = row + 16 kg
                }
            });
        };
        load_frags(std::integral_constant<int, 0>{}, Afrag[0]);
        v4i acc[NLV];
        static_for<0, NSTEP>([&](auto ic) {
            constexpr int i = decltype(ic)::value, rt = i / KS64, s = i % KS64;
            if constexpr (s == 0) {
#pragma unroll
                for (int q = 0; q < NLV; ++q) acc[q] = (v4i) {0, 0, 0, 0};
            }
            // keep the order: next step's LDS reads first, then this step's matrix instructions (hipcc would otherwise sink
            // every read to just before its consumer and wait for it there)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + 1 < NSTEP) load_frags(std::integral_constant<int, i + 1>{}, Afrag[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, S>([&](auto dc) {
                constexpr int d = decltype(dc)::value;
                if constexpr (F32) {
                    static_for<0, 4>([&](auto slc) {
                        constexpr int sl = decltype(slc)::value;
                        if constexpr (d + sl >= LMIN)
                            acc[d + sl - LMIN] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Afrag[i & 1][d], bsl[s][sl], acc[d + sl - LMIN], 0, 0, 0);
                    });
                } else {
                    acc[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Afrag[i & 1][d], blo[s], acc[d], 0, 0, 0);
                    if constexpr (NSL == 2) acc[d + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Afrag[i & 1][d], bhi[s], acc[d + 1], 0, 0, 0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (F32 && s == KS64 - 1) {
                // up to four products of 2^14 x 128 terms per level: no pairing in int32; value = cs 2^(E-30) 256^LMIN sum_q 256^q acc[q]
                // + the constant the XOR left out of the three lower slices, times the row's weight sum
                static_for<0, 4>([&](auto vc) {
                    constexpr int v = decltype(vc)::value;
                    double dsum = (double) acc[NLV - 1][v];
#pragma unroll
                    for (int q = NLV - 2; q >= 0; --q) dsum = fma(dsum, 256.0, (double) acc[q][v]);
                    val[4 * rt + v] = fma(dsum, cs * 16777216.0, 8421504.0 * wsum[4 * rt + v]) * colscale;
                });
            } else if constexpr (s == KS64 - 1) {
                // value = cs * sum_q 256^q acc[q]: pairs of levels first in int32 (|acc| <= 2^22), then fp64
                static_for<0, (rt < 2 ? 4 : NR)>([&](auto vc) {
                    constexpr int v = decltype(vc)::value;
                    double dsum = 0.0;
                    static_for<0, (NLV + 1) / 2>([&](auto pp) {
                        constexpr int p = (NLV + 1) / 2 - 1 - decltype(pp)::value;
                        int t = acc[2 * p][v];
                        if constexpr (2 * p + 1 < NLV) t += acc[2 * p + 1][v] * 256;
                        dsum = fma(dsum, 65536.0, (double) t);
                    });
                    if constexpr (rt < 2) val[4 * rt + v] = dsum * cs; else midv[v] = dsum * csm;
                });
            }
        });
        // result layout of the 16x16 tile: lane (g, nb) register v = row 4 g + v of column nb: mode g, v = 2*stream + part
        cd Cr = mk(val[0], val[1]), Ci = mk(val[2], val[3]), Ar = mk(val[4], val[5]), Ai = mk(val[6], val[7]);
        if (nco) {
            // the block's common phasor multiplies the complex DATA: re' = pr*re - pi*im, im' = pi*re + pr*im
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
            if constexpr (G > 1) {
                const double t2 = psi.re * midv[0] - psi.im * midv[1];
                midv[1] = psi.im * midv[0] + psi.re * midv[1];
                midv[0] = t2;
            }
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
        // tile aggregates go straight into the carry arrays: causal inclusive at the last block -> slot t+1,
        // anticausal inclusive at the first block -> slot t.  Index = slot*8 + mode*2 + stream.
        if (nb == 15) { carF[(tile + 1) * 8 + g * 2 + 0] = Cr; carF[(tile + 1) * 8 + g * 2 + 1] = Ci; }
        if (nb == 0) { carG[tile * 8 + g * 2 + 0] = Ar; carG[tile * 8 + g * 2 + 1] = Ai; }
        // local output part: direct term + 2 Re(exclusive causal + inclusive anticausal), summed over the 4 modes
        const double cexr = dpp_row<0x111>(Cr.re), cexi = dpp_row<0x111>(Ci.re);
        double pr = 2.0 * (cexr + Ar.re), pi = 2.0 * (cexi + Ai.re);
        if (g == 0) { pr = fma(d0p, u0r, pr); pi = fma(d0p, u0i, pi); }
        pr += __shfl_xor(pr, 16);
        pi += __shfl_xor(pi, 16);
        pr += __shfl_xor(pr, 32);
        pi += __shfl_xor(pi, 32);
        if (g == 0) o_lds[(tile * SDRT_TB + nb) * G] = mk(pr, pi);
        if constexpr (G > 1) {
            // output at the mid boundary: column-local part (mid tile, valid in the g == 0 lanes) + the states at the column's
            // edges carried by p^LB: 2 Re sum_modes p^LB ( causal state at the column start + anticausal state at its end )
            const cd Cxr = mk(cexr, dpp_row<0x111>(Cr.im)), Cxi = mk(cexi, dpp_row<0x111>(Ci.im));
            const cd Axr = dpp_row<0x101>(Ar), Axi = dpp_row<0x101>(Ai);
            double qr = 2.0 * (mh.re * (Cxr.re + Axr.re) - mh.im * (Cxr.im + Axr.im));
            double qi = 2.0 * (mh.re * (Cxi.re + Axi.re) - mh.im * (Cxi.im + Axi.im));
            if (g == 0) { qr += midv[0]; qi += midv[1]; }
            qr += __shfl_xor(qr, 16);
            qi += __shfl_xor(qi, 16);
            qr += __shfl_xor(qr, 32);
            qi += __shfl_xor(qi, 32);
            if (g == 0) o_lds[(tile * SDRT_TB + nb) * G + 1] = mk(qr, qi);
        }
    }
    SDRT_STAMP(2);
#undef SDRT_STAMP
    auto edge = [&](long idx) -> cd {
        const raw_t *rp = (const raw_t *) chunk_raw + 2 * idx;
        double xr = decode<DT>(rp[0], swap) - xoff, xi = decode<DT>(rp[1], swap) - xoff;
        if (nco) {
            const cd u = cmul(mk(xr, xi), cmul(nco_samp_g[idx % L], nco_blk_g[idx / L]));
            xr = u.re;
            xi = u.im;
        }
        return mk(xr, xi);
    };
    chain_tail(a, LB, row, chunk, smem, stp, edge, a.ybias ? (const cd *) a.ybias + (size_t) row * m : (const cd *) nullptr);
}


// MFMA i8 16x16x64 self-test: D(16x16) = A(16x64) * B(64x16), operands packed as the kernel assumes (lane = row/col + 16 kg,
// byte j of the lane's 16 <-> inner index 16 kg + j; result row 4 (lane >> 4) + v, col lane & 15).
__global__ void i8t_selftest_kernel(const int *A, const int *B, int *D) {
    const int lane = threadIdx.x & 63, rc = lane & 15, kg = lane >> 4;
    v4i av, bv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned pa = 0, pb = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = 16 * kg + 4 * i + b;
            pa |= ((unsigned) (A[rc * 64 + k] & 0xff)) << (8 * b);
            pb |= ((unsigned) (B[k * 16 + rc] & 0xff)) << (8 * b);
        }
        av[i] = (int) pa;
        bv[i] = (int) pb;
    }
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) D[(4 * kg + v) * 16 + rc] = acc[v];
}

// ---- launch helpers -----------------------------------------------------------------------------------------------
bool chain_i8t_supported(int L, int dt, int n) {
    return (L == 32 || L == 64) && (dt == SDRT_S8 || dt == SDRT_U8 || dt == SDRT_S16 || dt == SDRT_U16 || dt == SDRT_F32) &&
           n % (SDRT_TB * L * SDRT_I8T_G) == 0;
}

size_t chain_i8t_lds_bytes(int n, int L, int m) {
    const int G = SDRT_I8T_G, KS = L * G / 32, NR = 2 * (G - 1);
    return (size_t) lds_layout(n, L, m, G).total * 16 + (size_t) SDRT_I8_DIGITS * KS * (2 * 64 + 4 * NR) * 16;
}

template <int L, int DT>
static hipError_t launch_i8t_k(const DecimArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) chain_i8t_kernel<L, DT, SDRT_I8_DIGITS, SDRT_I8T_G>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((chain_i8t_kernel<L, DT, SDRT_I8_DIGITS, SDRT_I8T_G>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int L>
static hipError_t launch_i8t_l(const DecimArgs &a, int dt, dim3 grid, size_t lds, hipStream_t s) {
    switch (dt) {
        case SDRT_S8: return launch_i8t_k<L, SDRT_S8>(a, grid, lds, s);
        case SDRT_U8: return launch_i8t_k<L, SDRT_U8>(a, grid, lds, s);
        case SDRT_S16: return launch_i8t_k<L, SDRT_S16>(a, grid, lds, s);
        case SDRT_U16: return launch_i8t_k<L, SDRT_U16>(a, grid, lds, s);
        case SDRT_F32: return launch_i8t_k<L, SDRT_F32>(a, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_chain_i8t(const DecimArgs &a, int L, int dt, size_t n_chunks, hipStream_t s) {
    const size_t groups = (n_chunks + 7) / 8;
    const dim3 grid((unsigned) (groups * 8 * (size_t) a.rows));
    const size_t lds = chain_i8t_lds_bytes(a.n, L, a.m);
    switch (L) {
        case 32: return launch_i8t_l<32>(a, dt, grid, lds, s);
        case 64: return launch_i8t_l<64>(a, dt, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_i8t_selftest(const int *A, const int *B, int *D, hipStream_t s) {
    hipLaunchKernelGGL(i8t_selftest_kernel, dim3(1), dim3(64), 0, s, A, B, D);
    return hipGetLastError();
}
