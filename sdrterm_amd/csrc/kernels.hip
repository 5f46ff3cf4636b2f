// chain_kernel (f64 stage A on v_mfma_f64_16x16x4) + IQ prefix kernels + stand-alone discriminators + launchers.
#include "kernels_common.hpp"

// FIX: the second look of stage_a = auto -- the workgroups walk over the listed items (DecimArgs.fix_list).  A form of its own:
// wrapping the body in that loop costs the normal launch its register allocation (36 -> 264 bytes of scratch at L = 64).
template <int L, int DT, bool NCO, bool IQ, bool FIX = false>
__global__ __launch_bounds__(256, 4) void chain_kernel(DecimArgs a) {
    constexpr int KS = L / 4;                       // k-steps == samples per lane per tile
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
    // (second look of stage_a = auto: the workgroups walk over the listed items instead; one body, so the loop wraps it)
    const unsigned fix_n = FIX ? *a.fix_count : 0u;
    for (unsigned fw = bid;; fw += gridDim.x) {
    long chunk;
    int row;
    if constexpr (FIX) {
        if (fw >= fix_n) return;
        const int item = a.fix_list[fw];
        row = item / (int) a.n_chunks;
        chunk = item - (long) row * (long) a.n_chunks;
        if (fw != bid) __syncthreads();              // the previous item's LDS is done with
    } else {
        const unsigned xcd = bid & 7u, slot = bid >> 3;
        chunk = (long) (slot / (unsigned) a.rows) * 8 + xcd;
        row = (int) (slot % (unsigned) a.rows);
        if (chunk >= (long) a.n_chunks) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nb = lane & 15;
    const bool swap = a.swap != 0;

    // diagnostic phase stamps (a.stamps == nullptr in every production call)
    unsigned long long *stp = (a.stamps && (int) bid < a.stamp_wgs && lane == 0) ? a.stamps + ((size_t) bid * 4 + wave) * 16 : nullptr;
#define SDRT_STAMP(i) do { if (stp) stp[i] = __builtin_amdgcn_s_memtime(); } while (0)
    SDRT_STAMP(0);

    const LdsLayout lo = lds_layout(n, L, m, 1);
    cd *base = (cd *) smem;
    cd *o_lds = base + lo.o, *carF = base + lo.carF, *carG = base + lo.carG;

    const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk * (size_t) n * 2 * ITEM;
    const double nbias = a.norm_bias;
    const int ngroups = n / KS;                      // IQ correction: offsets are tabulated per KS-sample group
    const cd *iq_goff = IQ ? (const cd *) a.iq_group_off + (size_t) chunk * ngroups : nullptr;
    cd iq_c0 = mk(0.0, 0.0);
    if constexpr (IQ) iq_c0 = ((const cd *) a.iq_chunk_off)[chunk];
    const double lam = a.iq_lambda;
    const cd *nco_samp_g = (const cd *) a.nco_tile + (size_t) row * L;              // exp(i theta j), j < L
    const cd *nco_blk_g = (const cd *) a.nco_phase + (size_t) row * (size_t) (n / L);   // exp(i theta L B), B < n/L

    // Stage-A weights for this lane: A operand of v_mfma_f64_16x16x4: lane holds A[row = l&15][k = l>>4].
    double w[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) w[s] = tab->W[lane & 15][g * KS + s];
    cd mup[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mup[k] = mk(tab->mu_pow[k][g][0], tab->mu_pow[k][g][1]);
    const double d0p = tab->d0p;

    uint32_t cur[NW], nxt[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) { cur[i] = 0; nxt[i] = 0; }
    const size_t lane_off = ((size_t) nb * L + (size_t) g * KS) * 2 * ITEM;
    if (wave < nt) load_words<BYTES>(chunk_raw + (size_t) wave * TILE * 2 * ITEM + lane_off, cur);

    // NCO: lane (g, nb) keeps exp(i theta (g*KS + nb % KS)); k-step s broadcasts entry s of each 16-lane row (DPP
    // row_newbcast), which is the phasor of the sample that row's lanes feed to the MFMA in that step.
    cd ncoL = mk(1.0, 0.0);
    if constexpr (NCO) ncoL = nco_samp_g[g * KS + (nb % KS)];

    SDRT_STAMP(1);
    // ---- main loop over this wave's tiles, next tile's bytes prefetched ----------------------------------------------
    for (int tile = wave; tile < nt; tile += 4) {
        if (tile + 4 < nt) load_words<BYTES>(chunk_raw + (size_t) (tile + 4) * TILE * 2 * ITEM + lane_off, nxt);
        cd psi = mk(1.0, 0.0);
        if constexpr (NCO) psi = nco_blk_g[tile * SDRT_TB + nb];   // phasor common to this lane's block
        cd off = mk(0.0, 0.0);
        if constexpr (IQ) {
            const int grp = (tile * SDRT_TB + nb) * 4 + g;          // this lane's KS-sample group within the chunk
            const cd go = iq_goff[grp];
            const double pw = a.iq_pow[grp];
            off = mk(fma(pw, iq_c0.re, go.re), fma(pw, iq_c0.im, go.im));
        }

        v4d acc_r = {0.0, 0.0, 0.0, 0.0}, acc_i = {0.0, 0.0, 0.0, 0.0};
        double u0r = 0.0, u0i = 0.0;
        static_for<0, KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            double xr = decode<DT>(get_item<DT, 2 * s>(cur), swap) + nbias;
            double xi = decode<DT>(get_item<DT, 2 * s + 1>(cur), swap);
            if constexpr (IQ) {          // z = x - off; off += z * lambda  (read_file.py:74-76)
                xr -= off.re;
                xi -= off.im;
                off.re = fma(xr, lam, off.re);
                off.im = fma(xi, lam, off.im);
            }
            if constexpr (NCO) {
                const cd c = dpp_row<0x150 + s>(ncoL);
                const double tr = xr * c.re - xi * c.im;
                xi = xr * c.im + xi * c.re;
                xr = tr;
            }
            if (s == 0) { u0r = xr; u0i = xi; }
            acc_r = __builtin_amdgcn_mfma_f64_16x16x4f64(w[s], xr, acc_r, 0, 0, 0);
            acc_i = __builtin_amdgcn_mfma_f64_16x16x4f64(w[s], xi, acc_i, 0, 0, 0);
        });
        // D layout: lane (g, nb), register v holds row g + 4v of column nb:
        //   v=0 Re causal(mode g), v=1 Im causal, v=2 Re anticausal, v=3 Im anticausal; acc_r / acc_i = re / im stream.
        cd Cr = mk(acc_r[0], acc_r[1]), Ar = mk(acc_r[2], acc_r[3]);
        cd Ci = mk(acc_i[0], acc_i[1]), Ai = mk(acc_i[2], acc_i[3]);
        if constexpr (NCO) {
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
        if (g == 0) o_lds[tile * SDRT_TB + nb] = mk(pr, pi);
#pragma unroll
        for (int i = 0; i < NW; ++i) cur[i] = nxt[i];
    }
    SDRT_STAMP(2);
#undef SDRT_STAMP
    // sample idx of this chunk as the decimator sees it: bias, IQ correction (walk from the group start), NCO
    auto edge = [&](long idx) -> cd {
        const raw_t *rp = (const raw_t *) chunk_raw + 2 * idx;
        double xr = decode<DT>(rp[0], swap) + nbias, xi = decode<DT>(rp[1], swap);
        if constexpr (IQ) {
            const long grp = idx / KS;
            const cd go = iq_goff[grp];
            const double pw = a.iq_pow[grp];
            cd o2 = mk(fma(pw, iq_c0.re, go.re), fma(pw, iq_c0.im, go.im));
            for (long j = grp * KS; j < idx; ++j) {
                const raw_t *rj = (const raw_t *) chunk_raw + 2 * j;
                const double zr = decode<DT>(rj[0], swap) + nbias - o2.re, zi = decode<DT>(rj[1], swap) - o2.im;
                o2.re = fma(zr, lam, o2.re);
                o2.im = fma(zi, lam, o2.im);
            }
            xr -= o2.re;
            xi -= o2.im;
        }
        if constexpr (NCO) {
            const cd u = cmul(mk(xr, xi), cmul(nco_samp_g[idx % L], nco_blk_g[idx / L]));
            xr = u.re;
            xi = u.im;
        }
        return mk(xr, xi);
    };
    chain_tail(a, L, row, chunk, smem, stp, edge, (const cd *) nullptr);
    if constexpr (!FIX) return;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// IQ correction prefix (read_file.py:64-77: z[i] -= off; off += z[i]*L, off carried across chunks).
// off obeys off' = (1-lambda) off + lambda x, a first-order linear recurrence, so offsets at group starts are a prefix
// scan: iq_local_kernel scans inside each chunk from a zero start (one workgroup per chunk) and records the chunk's
// end value; iq_scan_kernel chains the chunks (and carries the stream state); the chain kernel then adds
// (1-lambda)^position * offset(chunk start) to the local values.
// ---------------------------------------------------------------------------------------------------------------

// One workgroup per chunk, one pass over the data.  Lane = one group of GS consecutive samples (a contiguous 2*GS*item
// bytes, read with the widest loads); 256 groups per sweep, n/(256 GS) sweeps.  Per sweep: the lane's own aggregate
// (GS steps of the recurrence from zero), an inclusive scan of the aggregates over the 256 lanes with multiplier
// (1-lambda)^GS (wave shuffles + 4 wave totals through LDS), and the carry of the previous sweeps.  The exclusive value is
// the offset at the group's first sample when the chunk starts from zero.
namespace {
// v + m * t for a real multiplier
__device__ __forceinline__ cd cfma_r(double m, cd t, cd v) { return mk(fma(m, t.re, v.re), fma(m, t.im, v.im)); }
// DPP move that only the rows of ROWS take (the others read zero)
template <int CTRL, int ROWS>
__device__ __forceinline__ cd dpp_masked(cd v) {
    auto one = [](double x) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWS, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWS, 0xF, false);
        return __hiloint2double(hi, lo);
    };
    return mk(one(v.re), one(v.im));
}
}  // namespace

template <int DT, int GS>
__global__ __launch_bounds__(256) void iq_local_kernel(IqArgs a) {
    constexpr int ITEM = Enc<DT>::size;
    constexpr int BYTES = GS * 2 * ITEM;
    constexpr int NW = BYTES < 4 ? 1 : BYTES / 4;
    __shared__ cd wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long chunk = blockIdx.x;
    const int n = a.n, ngroups = n / GS;
    const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk * (size_t) n * 2 * ITEM;
    const bool swap = a.swap != 0;
    const double lam = a.lambda, om = 1.0 - lam, nbias = a.norm_bias;
    const double ag = a.pow_g;                         // (1-lambda)^GS
    double mp[6];                                      // ag^(1,2,4,...,32)
    mp[0] = ag;
#pragma unroll
    for (int i = 1; i < 6; ++i) mp[i] = mp[i - 1] * mp[i - 1];
    const double ag64 = mp[5] * mp[5];                 // (1-lambda)^(64 GS): one wave
    const double plane = a.pow_tab[lane], ptid = a.pow_tab[tid];          // (1-lambda)^(GS lane), (1-lambda)^(GS tid)
    const double mrow = a.pow_tab[(lane & 15) + 1], mhalf = lane >= 32 ? a.pow_tab[lane - 31] : 0.0;   // distance to lane 15 of the row before / to lane 31
    const double ag256 = a.pow_tab[256];
    cd carry = mk(0.0, 0.0);                           // offset at the first sample of the current sweep
    cd *gout = a.group_off ? (cd *) a.group_off + (size_t) chunk * (size_t) ngroups : nullptr;     // null: chunk aggregates only
    // the records of the next sweep are requested before this sweep's are worked on (a workgroup is otherwise one HBM round
    // trip per sweep long); chunks have at least 256 groups (sdrt_create), so the first load is always inside the chunk
    uint32_t nxt[NW];
    load_words<BYTES>(chunk_raw + (size_t) (tid < ngroups ? tid : 0) * BYTES, nxt);
    for (int g0 = 0; g0 < ngroups; g0 += 256) {
        const int grp = g0 + tid;
        cd s = mk(0.0, 0.0);
        uint32_t wds[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) wds[w] = nxt[w];
        {
            const int gn = grp + 256;
            load_words<BYTES>(chunk_raw + (size_t) (gn < ngroups ? gn : grp < ngroups ? grp : 0) * BYTES, nxt);
        }
        if (grp < ngroups) {
            static_for<0, GS>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const double xr = decode<DT>(get_item<DT, 2 * j>(wds), swap) + nbias;
                const double xi = decode<DT>(get_item<DT, 2 * j + 1>(wds), swap);
                s.re = fma(om, s.re, lam * xr);
                s.im = fma(om, s.im, lam * xi);
            });
        }
        __syncthreads();                               // wtot of the previous sweep has been consumed
        // inclusive scan over the wave without LDS: four shifts inside the 16-lane rows, then the row totals handed on with
        // row_bcast15 (rows 1, 3 take lane 15 of the row before) and row_bcast31 (rows 2, 3 take lane 31)
        s = cfma_r(mp[0], dpp_row<0x111>(s), s);
        s = cfma_r(mp[1], dpp_row<0x112>(s), s);
        s = cfma_r(mp[2], dpp_row<0x114>(s), s);
        s = cfma_r(mp[3], dpp_row<0x118>(s), s);
        s = cfma_r(mrow, dpp_masked<0x142, 0xa>(s), s);
        s = cfma_r(mhalf, dpp_masked<0x143, 0xc>(s), s);
        if (lane == 63) wtot[wave] = s;
        __syncthreads();
        const cd ex = dpp_row<0x138>(s);               // wave_shr:1, lane 0 reads zero
        cd wp = mk(0.0, 0.0);                          // offset at the start of this wave, relative to the sweep start
        for (int wv = 0; wv < wave; ++wv) wp = mk(fma(ag64, wp.re, wtot[wv].re), fma(ag64, wp.im, wtot[wv].im));
        // offset at this group's first sample = (carry propagated over tid groups) + (wave start propagated over lane groups) + ex
        const cd off = mk(fma(ptid, carry.re, fma(plane, wp.re, ex.re)), fma(ptid, carry.im, fma(plane, wp.im, ex.im)));
        // (the wide-column kernel reads the offsets at column starts and at the tail window only: the rest is not written)
        if (gout && grp < ngroups && ((grp & a.wmask) == 0 || grp == a.wtail)) gout[grp] = off;
        // carry to the next sweep: everything of this sweep + the old carry over 256 groups
        cd tot = mk(0.0, 0.0);
        for (int wv = 0; wv < 4; ++wv) tot = mk(fma(ag64, tot.re, wtot[wv].re), fma(ag64, tot.im, wtot[wv].im));
        carry = mk(fma(ag256, carry.re, tot.re), fma(ag256, carry.im, tot.im));
    }
    if (tid == 0) ((cd *) a.chunk_agg)[chunk] = carry;     // offset at the end of the chunk when it starts from zero
}

// The same prefix pass for the wide-column kernel as consumer (round 4): it reads the offsets at the starts of its 256-sample
// columns and at the tail window only, so the pass works in columns too.  A wave takes a column with one coalesced load (lane l:
// samples 4l .. 4l+3), every lane weighs its four samples with lambda (1-lambda)^(255-t) from a table (independent fused
// multiply-adds: the per-sample recurrence of iq_local_kernel is a dependent chain, and its scan over 8- or 16-sample groups was
// most of its work), a DPP reduction gives the column's aggregate, and the 128 (64, 256) column aggregates of the chunk are chained
// by one wave at the end.  One read of the input at streaming rate.
template <int DT>
__global__ __launch_bounds__(256) void iq_cols_kernel(IqArgs a) {
    constexpr int ITEM = Enc<DT>::size, LC = 256;
    constexpr int BYTES = 4 * 2 * ITEM;                // four complex samples per lane
    constexpr int NW = BYTES / 4;
    constexpr int MAXC = 512;                          // columns per chunk this kernel takes (131072 samples)
    __shared__ cd agg[MAXC + 1];
    __shared__ cd tail_part;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long chunk = blockIdx.x;
    const int n = a.n, ncol = n / LC;
    const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk * (size_t) n * 2 * ITEM;
    const bool swap = a.swap != 0;
    const double nbias = a.norm_bias;
    double w[4];                                       // lambda (1-lambda)^(255 - 4 lane - j)
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = a.col_w[4 * lane + j];
    uint32_t nxt[NW];
    load_words<BYTES, true>(chunk_raw + ((size_t) wave * LC + 4 * lane) * 2 * ITEM, nxt);
    for (int c = wave; c < ncol; c += 4) {
        uint32_t wds[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) wds[k] = nxt[k];
        const int cn = c + 4 < ncol ? c + 4 : c;
        load_words<BYTES, true>(chunk_raw + ((size_t) cn * LC + 4 * lane) * 2 * ITEM, nxt);
        double sr = 0.0, si = 0.0;
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            sr = fma(w[j], decode<DT>(get_item<DT, 2 * j>(wds), swap) + nbias, sr);
            si = fma(w[j], decode<DT>(get_item<DT, 2 * j + 1>(wds), swap), si);
        });
        cd s = mk(sr, si);
        // sum over the wave: four shifts inside the 16-lane rows (lane 15 of a row ends up with the row's sum), then the rows
        s = mk(s.re + dpp_row<0x111>(s.re), s.im + dpp_row<0x111>(s.im));
        s = mk(s.re + dpp_row<0x112>(s.re), s.im + dpp_row<0x112>(s.im));
        s = mk(s.re + dpp_row<0x114>(s.re), s.im + dpp_row<0x114>(s.im));
        s = mk(s.re + dpp_row<0x118>(s.re), s.im + dpp_row<0x118>(s.im));
        auto rl = [](double v, int l) {                 // v_readlane: no trip through the LDS crossbar
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
        };
        const double r0 = rl(s.re, 15), r1 = rl(s.re, 31), r2 = rl(s.re, 47), r3 = rl(s.re, 63);
        const double i0 = rl(s.im, 15), i1 = rl(s.im, 31), i2 = rl(s.im, 47), i3 = rl(s.im, 63);
        if (lane == 0) {
            agg[c] = mk((r0 + r1) + (r2 + r3), (i0 + i1) + (i2 + i3));
            // the tail window starts 224 samples into the last column: lanes 0 .. 55 = rows 0 .. 2 and lanes 48 .. 55 of row 3;
            // the weights there are (1-lambda)^32 too small (col_w[256] = (1-lambda)^-32)
            if (c == ncol - 1) tail_part = mk(r0 + r1 + r2, i0 + i1 + i2);
        }
        if (c == ncol - 1) {
            // row 3, lanes 48 .. 55: the shifted sums hold lanes 48 .. l at lane l
            const double t7r = rl(s.re, 55), t7i = rl(s.im, 55);
            if (lane == 0) { agg[MAXC] = mk(t7r, t7i); }
        }
    }
    __syncthreads();
    if (wave == 0) {
        // offsets at the column starts: off[c+1] = mu off[c] + agg[c], mu = (1-lambda)^256; lane l chains columns [l per, (l+1) per)
        const double mu = a.col_w[257];
        const int per = (ncol + 63) / 64;
        cd gout_local[MAXC / 64];
        cd s = mk(0.0, 0.0);
        double p = 1.0;
        for (int k = 0; k < per; ++k) {
            const int c = lane * per + k;
            gout_local[k] = s;                                         // relative to the lane's own start
            if (c < ncol) { s = mk(fma(mu, s.re, agg[c].re), fma(mu, s.im, agg[c].im)); p *= mu; }
        }
        // exclusive scan of the lanes' affine maps (p, s) over the wave
        cd ex = mk(0.0, 0.0);
        {
            cd acc = s;
            double pp = p;
            for (int d = 1; d < 64; d <<= 1) {
                const double ore = __shfl_up(acc.re, d), oim = __shfl_up(acc.im, d), op = __shfl_up(pp, d);
                if (lane >= d) { acc = mk(fma(pp, ore, acc.re), fma(pp, oim, acc.im)); pp *= op; }
            }
            const double ere = __shfl_up(acc.re, 1), eim = __shfl_up(acc.im, 1);
            if (lane > 0) ex = mk(ere, eim);
            if (lane == 63) ((cd *) a.chunk_agg)[chunk] = acc;         // offset at the end of the chunk when it starts from zero
        }
        cd *gout = a.group_off ? (cd *) a.group_off + (size_t) chunk * (size_t) (n / a.gs) : nullptr;
        double q = 1.0;
        const int gpc = LC / a.gs;                                     // groups per column
        for (int k = 0; k < per; ++k) {
            const int c = lane * per + k;
            if (c < ncol) {
                const cd off = mk(fma(q, ex.re, gout_local[k].re), fma(q, ex.im, gout_local[k].im));
                if (gout) gout[(size_t) c * gpc] = off;
                if (gout && c == ncol - 1) {
                    // offset at sample n - 32: the last column's start carried over 224 samples + its first 224 samples' share
                    const double m224 = a.col_w[258], inv32 = a.col_w[256];
                    const cd tp = mk((tail_part.re + agg[MAXC].re) * inv32, (tail_part.im + agg[MAXC].im) * inv32);
                    gout[a.wtail] = mk(fma(m224, off.re, tp.re), fma(m224, off.im, tp.im));
                }
                q *= mu;
            }
        }
    }
}

// one workgroup: chunk offsets off[c+1] = (1-lambda)^n off[c] + agg[c], off[0] = carried state
constexpr int IQ_SCAN_THREADS = 1024;
__global__ __launch_bounds__(IQ_SCAN_THREADS) void iq_scan_kernel(IqArgs a) {
    __shared__ cd tot[IQ_SCAN_THREADS / 64];
    __shared__ double pw[IQ_SCAN_THREADS / 64];
    const int tid = threadIdx.x;
    const size_t nc = a.n_chunks, per = (nc + IQ_SCAN_THREADS - 1) / IQ_SCAN_THREADS;
    const size_t c0 = (size_t) tid * per < nc ? (size_t) tid * per : nc, c1 = c0 + per < nc ? c0 + per : nc;
    const cd *agg = (const cd *) a.chunk_agg;
    cd *offc = (cd *) a.chunk_off;
    cd s = mk(0.0, 0.0);
    double p = 1.0;
    // (batches of eight loads in flight: the recurrence itself is short, the loop is bound by the latency of its loads)
    constexpr int UB = 8;
    size_t c = c0;
    for (; c + UB <= c1; c += UB) {
        cd v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) v[u] = agg[c + u];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            s = mk(fma(a.pow_n, s.re, v[u].re), fma(a.pow_n, s.im, v[u].im));
            p *= a.pow_n;
        }
    }
    for (; c < c1; ++c) {
        s = mk(fma(a.pow_n, s.re, agg[c].re), fma(a.pow_n, s.im, agg[c].im));
        p *= a.pow_n;
    }
    // a thread's range maps the offset o at its start to p o + s: inclusive scan of these maps over the wave (composition),
    // then over the 16 wave totals
    const int lane = tid & 63, wave = tid >> 6;
    double P = p;
    cd S = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double Pu = __shfl_up(P, d);
        const cd Su = mk(__shfl_up(S.re, d), __shfl_up(S.im, d));
        if (lane >= d) {
            S = mk(fma(P, Su.re, S.re), fma(P, Su.im, S.im));
            P *= Pu;
        }
    }
    if (lane == 63) { tot[wave] = S; pw[wave] = P; }
    __syncthreads();
    const cd run0 = mk(a.state[0], a.state[1]);
    cd base = run0;                                         // offset at the start of this wave's range
    for (int w = 0; w < wave; ++w) base = mk(fma(pw[w], base.re, tot[w].re), fma(pw[w], base.im, tot[w].im));
    double Pe = __shfl_up(P, 1);
    cd Se = mk(__shfl_up(S.re, 1), __shfl_up(S.im, 1));
    if (lane == 0) { Pe = 1.0; Se = mk(0.0, 0.0); }
    cd o = mk(fma(Pe, base.re, Se.re), fma(Pe, base.im, Se.im));      // offset at the start of this thread's range
    __syncthreads();                                        // every thread has read the carried state
    if (tid == IQ_SCAN_THREADS - 1) {
        const cd run = mk(fma(P, base.re, S.re), fma(P, base.im, S.im));
        offc[nc] = run;
        a.state[0] = run.re;
        a.state[1] = run.im;
    }
    c = c0;
    for (; c + UB <= c1; c += UB) {
        cd v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) v[u] = agg[c + u];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            offc[c + u] = o;
            o = mk(fma(a.pow_n, o.re, v[u].re), fma(a.pow_n, o.im, v[u].im));
        }
    }
    for (; c < c1; ++c) {
        offc[c] = o;
        o = mk(fma(a.pow_n, o.re, agg[c].re), fma(a.pow_n, o.im, agg[c].im));
    }
}

// The wide-column kernel corrects every chunk from a zero starting offset (its offsets come out of its own matrix rows, no
// second pass over the input).  The chain is linear in the offset h at the chunk's start: y = y_local - h * D(nco * (1-lambda)^t),
// the second factor a fixed vector R per row (computed once at sdrt_create).  This kernel subtracts that share.
__global__ __launch_bounds__(256) void iq_fix_kernel(double *ydec, const double *__restrict__ R, const double *__restrict__ chunk_off,
                                                     int rows, size_t n_chunks, int m) {
    const size_t total = (size_t) rows * n_chunks * (size_t) m;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t) gridDim.x * 256) {
        const size_t item = i / (size_t) m, k = i - item * (size_t) m, row = item / n_chunks, chunk = item - row * n_chunks;
        const cd h = ((const cd *) chunk_off)[chunk], r = ((const cd *) R)[row * (size_t) m + k];
        cd y = ((cd *) ydec)[i];
        y.re = fma(-h.re, r.re, fma(h.im, r.im, y.re));
        y.im = fma(-h.re, r.im, fma(-h.im, r.re, y.im));
        ((cd *) ydec)[i] = y;
    }
}

hipError_t launch_iq_fix(double *ydec, const double *R, const double *chunk_off, int rows, size_t n_chunks, int m, hipStream_t s) {
    const size_t total = (size_t) rows * n_chunks * (size_t) m, wgs = (total + 255) / 256;
    hipLaunchKernelGGL(iq_fix_kernel, dim3((unsigned) (wgs < 8192 ? wgs : 8192)), dim3(256), 0, s, ydec, R, chunk_off, rows, n_chunks, m);
    return hipGetLastError();
}

template <int DT>
static hipError_t launch_iq_local_dt(const IqArgs &a, dim3 grid, hipStream_t s) {
    switch (a.gs) {
        case 2: hipLaunchKernelGGL((iq_local_kernel<DT, 2>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((iq_local_kernel<DT, 4>), grid, dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((iq_local_kernel<DT, 8>), grid, dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((iq_local_kernel<DT, 16>), grid, dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_iq_local(const IqArgs &a, hipStream_t s) {
    const dim3 grid((unsigned) a.n_chunks);
    if (a.col_w && a.wmask && a.n % 256 == 0 && a.n / 256 <= 512 && (256 / a.gs) == a.wmask + 1) {
        // the wide-column kernel is the consumer: offsets at column starts and at the tail window only
        switch (a.dt) {
            case SDRT_S8: hipLaunchKernelGGL((iq_cols_kernel<SDRT_S8>), grid, dim3(256), 0, s, a); return hipGetLastError();
            case SDRT_U8: hipLaunchKernelGGL((iq_cols_kernel<SDRT_U8>), grid, dim3(256), 0, s, a); return hipGetLastError();
            case SDRT_S16: hipLaunchKernelGGL((iq_cols_kernel<SDRT_S16>), grid, dim3(256), 0, s, a); return hipGetLastError();
            case SDRT_U16: hipLaunchKernelGGL((iq_cols_kernel<SDRT_U16>), grid, dim3(256), 0, s, a); return hipGetLastError();
            case SDRT_F32: hipLaunchKernelGGL((iq_cols_kernel<SDRT_F32>), grid, dim3(256), 0, s, a); return hipGetLastError();
            default: break;
        }
    }
    switch (a.dt) {
        case SDRT_S8: return launch_iq_local_dt<SDRT_S8>(a, grid, s);
        case SDRT_U8: return launch_iq_local_dt<SDRT_U8>(a, grid, s);
        case SDRT_S16: return launch_iq_local_dt<SDRT_S16>(a, grid, s);
        case SDRT_U16: return launch_iq_local_dt<SDRT_U16>(a, grid, s);
        case SDRT_S32: return launch_iq_local_dt<SDRT_S32>(a, grid, s);
        case SDRT_U32: return launch_iq_local_dt<SDRT_U32>(a, grid, s);
        case SDRT_F32: return launch_iq_local_dt<SDRT_F32>(a, grid, s);
        case SDRT_F64: return launch_iq_local_dt<SDRT_F64>(a, grid, s);
    }
    return hipErrorInvalidValue;
}

// The same scan over several workgroups, for batches of many chunks (round 4: one workgroup took 0.23 ms for the 131072 chunks of a
// bench step -- one CU walking 4 MB).  Three launches: every workgroup of 256 threads reduces its IQ_SCAN_SPAN chunks to one map
// o -> P o + S; one workgroup chains the maps from the carried state; every workgroup walks its chunks again from its starting offset.
constexpr int IQ_SCAN_PER = 8, IQ_SCAN_SPAN = 256 * IQ_SCAN_PER;
namespace {
// this thread's chunks [c0, c1) as a map, and the inclusive composition over the workgroup (P, S of the threads up to and including it)
__device__ __forceinline__ void iq_block_maps(const IqArgs &a, size_t c0, size_t c1, double &P, cd &S, double &Pth, cd &Sth, cd *tot, double *pw) {
    const cd *agg = (const cd *) a.chunk_agg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cd v[IQ_SCAN_PER];
#pragma unroll
    for (int u = 0; u < IQ_SCAN_PER; ++u) v[u] = c0 + u < c1 ? agg[c0 + u] : mk(0.0, 0.0);
    cd s = mk(0.0, 0.0);
    double p = 1.0;
#pragma unroll
    for (int u = 0; u < IQ_SCAN_PER; ++u)
        if (c0 + u < c1) { s = mk(fma(a.pow_n, s.re, v[u].re), fma(a.pow_n, s.im, v[u].im)); p *= a.pow_n; }
    Pth = p; Sth = s;
    P = p; S = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double Pu = __shfl_up(P, d);
        const cd Su = mk(__shfl_up(S.re, d), __shfl_up(S.im, d));
        if (lane >= d) { S = mk(fma(P, Su.re, S.re), fma(P, Su.im, S.im)); P *= Pu; }
    }
    if (lane == 63) { tot[wave] = S; pw[wave] = P; }
    __syncthreads();
    // the waves before this one
    double Pb = 1.0;
    cd Sb = mk(0.0, 0.0);
    for (int w = 0; w < wave; ++w) { Sb = mk(fma(pw[w], Sb.re, tot[w].re), fma(pw[w], Sb.im, tot[w].im)); Pb *= pw[w]; }
    S = mk(fma(P, Sb.re, S.re), fma(P, Sb.im, S.im));
    P *= Pb;
}
}  // namespace

__global__ __launch_bounds__(256) void iq_scan_reduce_kernel(IqArgs a) {
    __shared__ cd tot[4];
    __shared__ double pw[4];
    const size_t base = (size_t) blockIdx.x * IQ_SCAN_SPAN, c0 = base + (size_t) threadIdx.x * IQ_SCAN_PER;
    const size_t c0c = c0 < a.n_chunks ? c0 : a.n_chunks;
    double P, Pth;
    cd S, Sth;
    iq_block_maps(a, c0c, a.n_chunks, P, S, Pth, Sth, tot, pw);
    if (threadIdx.x == 255) { a.scan_part[4 * blockIdx.x] = P; a.scan_part[4 * blockIdx.x + 1] = S.re; a.scan_part[4 * blockIdx.x + 2] = S.im; }
}

// one workgroup: the workgroups' maps chained from the carried state (parts <= SDRT_IQ_SCAN_PARTS, two per thread at most)
__global__ __launch_bounds__(256) void iq_scan_top_kernel(IqArgs a, int parts) {
    __shared__ double P_[SDRT_IQ_SCAN_PARTS];
    __shared__ cd S_[SDRT_IQ_SCAN_PARTS];
    for (int b = threadIdx.x; b < parts; b += 256) { P_[b] = a.scan_part[4 * b]; S_[b] = mk(a.scan_part[4 * b + 1], a.scan_part[4 * b + 2]); }
    __syncthreads();
    if (threadIdx.x == 0) {
        cd o = mk(a.state[0], a.state[1]);
        for (int b = 0; b < parts; ++b) {
            const cd S = S_[b];
            S_[b] = o;                                     // the workgroup's starting offset takes the place of its sum
            o = mk(fma(P_[b], o.re, S.re), fma(P_[b], o.im, S.im));
        }
        ((cd *) a.chunk_off)[a.n_chunks] = o;
        a.state[0] = o.re;
        a.state[1] = o.im;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < parts; b += 256) { a.scan_part[4 * b + 1] = S_[b].re; a.scan_part[4 * b + 2] = S_[b].im; }
}

__global__ __launch_bounds__(256) void iq_scan_apply_kernel(IqArgs a) {
    __shared__ cd tot[4];
    __shared__ double pw[4];
    const size_t base = (size_t) blockIdx.x * IQ_SCAN_SPAN, c0 = base + (size_t) threadIdx.x * IQ_SCAN_PER;
    const size_t c0c = c0 < a.n_chunks ? c0 : a.n_chunks;
    double P, Pth;
    cd S, Sth;
    iq_block_maps(a, c0c, a.n_chunks, P, S, Pth, Sth, tot, pw);
    // exclusive: the threads before this one
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double Pe = __shfl_up(P, 1);
    cd Se = mk(__shfl_up(S.re, 1), __shfl_up(S.im, 1));
    if (lane == 0) {
        Pe = 1.0; Se = mk(0.0, 0.0);
        for (int w = 0; w < wave; ++w) { Se = mk(fma(pw[w], Se.re, tot[w].re), fma(pw[w], Se.im, tot[w].im)); Pe *= pw[w]; }
    }
    const cd b0 = mk(a.scan_part[4 * blockIdx.x + 1], a.scan_part[4 * blockIdx.x + 2]);
    cd o = mk(fma(Pe, b0.re, Se.re), fma(Pe, b0.im, Se.im));            // offset at the start of this thread's first chunk
    const cd *agg = (const cd *) a.chunk_agg;
    cd *offc = (cd *) a.chunk_off;
#pragma unroll
    for (int u = 0; u < IQ_SCAN_PER; ++u)
        if (c0c + u < a.n_chunks && c0 + u < base + IQ_SCAN_SPAN) {
            const cd v = agg[c0c + u];
            offc[c0c + u] = o;
            o = mk(fma(a.pow_n, o.re, v.re), fma(a.pow_n, o.im, v.im));
        }
}

hipError_t launch_iq_scan(const IqArgs &a, hipStream_t s) {
    const size_t parts = (a.n_chunks + IQ_SCAN_SPAN - 1) / IQ_SCAN_SPAN;
    const int multi = sdrt_diag_int("iq_scan_multi", 1);
    if (multi && a.scan_part && parts >= 4 && parts <= SDRT_IQ_SCAN_PARTS) {
        hipLaunchKernelGGL(iq_scan_reduce_kernel, dim3((unsigned) parts), dim3(256), 0, s, a);
        hipLaunchKernelGGL(iq_scan_top_kernel, dim3(1), dim3(256), 0, s, a, (int) parts);
        hipLaunchKernelGGL(iq_scan_apply_kernel, dim3((unsigned) parts), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(iq_scan_kernel, dim3(1), dim3(IQ_SCAN_THREADS), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone discriminators on already-decimated samples (demodulation.py:25-68): one workgroup per row.
// mode 0 FM (pair angle + resample m/2 -> m), 1 AM, 2 real part, 3 imaginary part.  Not on the streaming path (the chain
// kernel has them fused); this backs sdrterm_amd.dsp.demodulation when called directly.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void demod_kernel(const double *y, double *out, const double *resample_odd, int m, int mode) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *av = (double *) smem;                 // [m/2] pair angles (FM)
    const cd *yr = (const cd *) y + (size_t) blockIdx.x * m;
    double *o = out + (size_t) blockIdx.x * m;
    const int tid = threadIdx.x;
    if (mode == 0) {
        const int N = m >> 1;
        for (int j = tid; j < N; j += 256) {
            const cd p = yr[2 * j], c = yr[2 * j + 1];
            av[j] = atan2_pi(p.im * c.re - p.re * c.im, p.re * c.re + p.im * c.im);
        }
        __syncthreads();
        for (int j = tid; j < N; j += 256) {
            double s0 = 0.0;
            for (int k = 0; k < N; ++k) {
                int d = j - k;
                if (d < 0) d += N;
                s0 = fma(av[k], resample_odd[d], s0);
            }
            o[2 * j] = av[j];
            o[2 * j + 1] = s0;
        }
    } else {
        for (int k = tid; k < m; k += 256) {
            const cd v = yr[k];
            o[k] = mode == 1 ? v.re * v.re + v.im * v.im : (mode == 2 ? v.re : v.im);
        }
    }
}

hipError_t launch_demod(const double *y, double *out, const double *resample_odd, int rows, int m, int mode, hipStream_t s) {
    hipLaunchKernelGGL(demod_kernel, dim3((unsigned) rows), dim3(256), (size_t) (m / 2 + 1) * sizeof(double), s, y, out,
                       resample_odd, m, mode);
    return hipGetLastError();
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
size_t chain_lds_bytes(int n, int L, int m) { return (size_t) lds_layout(n, L, m, 1).total * 16; }

template <int L, int DT, bool NCO, bool IQ>
static hipError_t launch_chain_k(const DecimArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    // more than 64 KiB of dynamic LDS has to be opted into per kernel (gfx950 has 160 KiB per workgroup)
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) chain_kernel<L, DT, NCO, IQ>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((chain_kernel<L, DT, NCO, IQ>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int L, int DT, bool NCO>
static hipError_t launch_chain_fix(const DecimArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) chain_kernel<L, DT, NCO, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((chain_kernel<L, DT, NCO, false, true>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int L, int DT>
static hipError_t launch_chain_ldt(const DecimArgs &a, bool nco, bool iq, dim3 grid, size_t lds, hipStream_t s) {
    if (a.fix_list) {
        // (only where stage_a = auto looks twice: 8/16-bit integer records at the decimations of the wide kernel, no IQ correction)
        if constexpr ((L == 32 || L == 64) && (DT == SDRT_S8 || DT == SDRT_U8 || DT == SDRT_S16 || DT == SDRT_U16))
            return nco ? launch_chain_fix<L, DT, true>(a, grid, lds, s) : launch_chain_fix<L, DT, false>(a, grid, lds, s);
        else
            return hipErrorInvalidValue;
    }
    if (nco) return iq ? launch_chain_k<L, DT, true, true>(a, grid, lds, s) : launch_chain_k<L, DT, true, false>(a, grid, lds, s);
    return iq ? launch_chain_k<L, DT, false, true>(a, grid, lds, s) : launch_chain_k<L, DT, false, false>(a, grid, lds, s);
}

template <int L>
static hipError_t launch_chain_l(const DecimArgs &a, int dt, bool nco, bool iq, dim3 grid, size_t lds, hipStream_t s) {
    switch (dt) {
        case SDRT_S8: return launch_chain_ldt<L, SDRT_S8>(a, nco, iq, grid, lds, s);
        case SDRT_U8: return launch_chain_ldt<L, SDRT_U8>(a, nco, iq, grid, lds, s);
        case SDRT_S16: return launch_chain_ldt<L, SDRT_S16>(a, nco, iq, grid, lds, s);
        case SDRT_U16: return launch_chain_ldt<L, SDRT_U16>(a, nco, iq, grid, lds, s);
        case SDRT_S32: return launch_chain_ldt<L, SDRT_S32>(a, nco, iq, grid, lds, s);
        case SDRT_U32: return launch_chain_ldt<L, SDRT_U32>(a, nco, iq, grid, lds, s);
        case SDRT_F32: return launch_chain_ldt<L, SDRT_F32>(a, nco, iq, grid, lds, s);
        case SDRT_F64: return launch_chain_ldt<L, SDRT_F64>(a, nco, iq, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_chain(const DecimArgs &a, int L, int dt, bool nco, bool iq, size_t n_chunks, hipStream_t s) {
    // ids are (slot << 3 | xcd): round the chunk count up to a multiple of 8, surplus blocks exit at once
    // (a.fix_list: a fixed number of workgroups walks over the listed items -- usually none: the launch costs a few microseconds)
    const size_t groups = (n_chunks + 7) / 8;
    const size_t all = groups * 8 * (size_t) a.rows;
    const dim3 grid((unsigned) (a.fix_list ? (all < 1024 ? all : 1024) : all));
    const size_t lds = chain_lds_bytes(a.n, L, a.m);
    switch (L) {
        case 8: return launch_chain_l<8>(a, dt, nco, iq, grid, lds, s);
        case 16: return launch_chain_l<16>(a, dt, nco, iq, grid, lds, s);
        case 32: return launch_chain_l<32>(a, dt, nco, iq, grid, lds, s);
        case 64: return launch_chain_l<64>(a, dt, nco, iq, grid, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mfma_selftest(const double *A, const double *B, double *D, hipStream_t s) {
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, s, A, B, D);
    return hipGetLastError();
}
