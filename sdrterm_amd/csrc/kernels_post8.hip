// post_fft8_kernel: the FM demodulation stage (demodulation.py:25-38 fmDemod, dsp_processor.py:32-36,149 output lowpass) with
// LPI = 8 lanes of a wave per (chunk, row) item -- eight items per wave.
//
// Why (profiles/r02_pmc_summary.txt, post_fft_kernel<1,4,true>): with one item per wave the two 256-point complex transforms
// of scipy.signal.resample made six trips through a per-wave LDS buffer (99 LDS instructions per item, 41 % of the LDS
// cycles bank conflicts) and spent half of their arithmetic on the zero imaginary parts of a real sequence.  Here
//   * the N pair angles of an item are packed two to a complex number, z[n] = a[2n] + i a[2n+1]: ONE half-size complex
//     transform each way (H = N/2 = 128 points); the packed spectrum of the half-sample shifted sequence comes out of the packed
//     spectrum of the angles bin by bin, Y[k] = alpha_k Z[k] + beta_k conj(Z[H-k]) (tables: sdrt_build_post8_tables);
//   * H = 16 x 8: every lane holds 16 points, a 16-point transform runs in its registers (constant twiddles, no memory),
//     then ONE exchange through LDS and 8-point transforms; the lane that finishes bin k also holds bin H-k;
//   * the exchange image is laid out so that every ds_write_b128 / ds_read_b128 is conflict free (slot = 17 n2 + k1, items 136
//     slots apart: tools/lds_banks.py);
//   * the output lowpass never leaves the registers: a lane holds 4 consecutive output samples of each block of 32, the
//     blocks' states are scanned over the item's 8 lanes with DPP shifts and chained over the 16 blocks in every lane.
// Every lane group runs the same instructions on its own item only: an item's bits do not depend on its neighbours.
#include <cstdlib>

#include "kernels_common.hpp"

namespace {

// cos(2 pi k / 32), k = 0 .. 8
__device__ constexpr double kCos32[9] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                         0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                                         0.19509032201612826785, 0.0};
constexpr double cos32(int k) {       // k in [0, 16]
    return k <= 8 ? kCos32[k] : -kCos32[16 - k];
}
constexpr double sin32(int k) {       // k in [0, 16]
    return k <= 8 ? kCos32[8 - k] : kCos32[k - 8];
}

__device__ __forceinline__ cd csub8(cd a, cd b) { return mk(a.re - b.re, a.im - b.im); }

// v * exp(-2 pi i K / N_) (forward) or v * exp(+2 pi i K / N_) (INV), 0 <= K < N_/2, N_ <= 32
template <int N_, int K, bool INV>
__device__ __forceinline__ cd tw_mul(cd v) {
    if constexpr (K == 0) return v;
    else if constexpr (4 * K == N_) return INV ? mk(-v.im, v.re) : mk(v.im, -v.re);
    else if constexpr (8 * K == N_) {
        constexpr double r = 0.70710678118654752440;
        return INV ? mk((v.re - v.im) * r, (v.re + v.im) * r) : mk((v.re + v.im) * r, (v.im - v.re) * r);
    } else if constexpr (8 * K == 3 * N_) {
        constexpr double r = 0.70710678118654752440;
        return INV ? mk(-(v.re + v.im) * r, (v.re - v.im) * r) : mk((v.im - v.re) * r, -(v.re + v.im) * r);
    } else {
        constexpr double c = cos32(K * (32 / N_)), s = sin32(K * (32 / N_));
        return INV ? mk(fma(v.re, c, -(v.im * s)), fma(v.im, c, v.re * s)) : mk(fma(v.re, c, v.im * s), fma(v.im, c, -(v.re * s)));
    }
}

// N_-point transform of values held in registers (decimation in time, natural order in and out; every index is a compile-time
// constant, so the arrays never exist in memory)
template <int N_, bool INV>
__device__ __forceinline__ void reg_fft(const cd *in, cd *out) {
    if constexpr (N_ == 2) {
        out[0] = cadd(in[0], in[1]);
        out[1] = csub8(in[0], in[1]);
    } else {
        cd e[N_ / 2], o[N_ / 2], E[N_ / 2], O[N_ / 2];
#pragma unroll
        for (int k = 0; k < N_ / 2; ++k) { e[k] = in[2 * k]; o[k] = in[2 * k + 1]; }
        reg_fft<N_ / 2, INV>(e, E);
        reg_fft<N_ / 2, INV>(o, O);
        static_for<0, N_ / 2>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const cd t = tw_mul<N_, k, INV>(O[k]);
            out[k] = cadd(E[k], t);
            out[k + N_ / 2] = csub8(E[k], t);
        });
    }
}

__device__ __forceinline__ double swz_bcast_last8(double v) {
    // lane 7 of every group of 8 lanes to the whole group (ds_swizzle, bit-mask mode: and 0x18, or 0x07)
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x18 | (0x07 << 5));
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x18 | (0x07 << 5));
    return __hiloint2double(hi, lo);
}

// (atan2_pi: kernels_common.hpp)

// DPP quad permutation (control = the four source lanes of a quad, two bits each); sources never leave the quad
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ cd dpp_quad(cd v) { return mk(dpp_quad<CTRL>(v.re), dpp_quad<CTRL>(v.im)); }
// row_shr:4 into lanes 4-7 and 12-15 of every 16-lane row (banks 1 and 3); lanes 0-3 and 8-11 get 0
__device__ __forceinline__ double dpp_shr4_banks13(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x114, 0xF, 0xA, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x114, 0xF, 0xA, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ cd dpp_shr4_banks13(cd v) { return mk(dpp_shr4_banks13(v.re), dpp_shr4_banks13(v.im)); }

// within-wave ordering of LDS traffic between lanes: the hardware executes a wave's LDS instructions in order; this keeps
// the compiler from moving them across
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

#ifndef SDRT_P8_BLOCKS_AHEAD
#define SDRT_P8_BLOCKS_AHEAD 16
#endif
// Eight lanes per item, pair products in: THREE waves per SIMD (round 4).  What held the kernel at two was 254 registers (every load
// of an item in flight: 128 of them) and 17 KiB of exchange image per wave.  Both halve: the item's pair products arrive in two
// halves, and the two bin series a lane finishes (k1 = l and 16 - l) go through the exchange image one after the other -- the image
// is 8 rows of 9 slots instead of 17 (the same conflict-free residues: 9 = 17 - 8, items 72 = 8 mod 16 slots apart).
#ifndef SDRT_P8_BLOCKS_AHEAD3
#define SDRT_P8_BLOCKS_AHEAD3 2
#endif
#ifndef SDRT_P8_LEAN_SAMPLES
#define SDRT_P8_LEAN_SAMPLES 0
#endif
#ifndef SDRT_P8_KEEPX
#define SDRT_P8_KEEPX 1
#endif
#ifndef SDRT_P8_OCC
#define SDRT_P8_OCC 2
#endif
#ifndef SDRT_P8_WAVES3
#define SDRT_P8_WAVES3 1
#endif
__host__ __device__ constexpr bool p8_lean(int lpi, bool pairs) { return SDRT_P8_WAVES3 && (lpi == 8 || lpi == 16) && (pairs || SDRT_P8_LEAN_SAMPLES); }
__host__ __device__ constexpr bool p8_halves(int lpi, bool pairs) { return p8_lean(lpi, pairs) && lpi == 8; }

// LPI lanes per item: 8 (N = 256 pair angles, 512 outputs per chunk) or 16 (N = 512, 1024 outputs per chunk: a DPP row of
// lanes is one item).  16-byte slots of an item's exchange image: 17 per n2 (8 x 17 = 136 = 8 mod 16, 16 x 17 = 272 = 0 mod 16:
// what keeps the four lane groups of a ds_read_b128 on sixteen different slots in either form, tools/lds_banks.py).
template <int LPI, int WPB, bool PAIRS>
__global__ __launch_bounds__(64 * WPB, LPI == 16 && WPB == 8 ? 1 : (p8_halves(LPI, PAIRS) && WPB == 4 ? SDRT_P8_OCC : 2)) void post_fft8_kernel(PostArgs a) {
    constexpr bool HX = p8_halves(LPI, PAIRS), LEAN = p8_lean(LPI, PAIRS);
    constexpr int ST = HX ? 9 : 17;                  // slots between the rows of an item's exchange image
    constexpr int P = SDRT_P8_PPL, H = P * LPI, N = 2 * H, M = 2 * N, IPW = 64 / LPI, TAB = P * LPI, SO = ST * LPI, NC = P / LPI;
    static_assert(LPI == 4 || LPI == 8 || LPI == 16, "4, 8 or 16 lanes per item");
    extern __shared__ __align__(16) unsigned char smem[];
    cd *tabL = (cd *) smem;                                        // tw1 | tw2 | alpha | beta | lane rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, o = lane / LPI, l = lane % LPI;
    cd *X = tabL + 4 * TAB + 8 * LPI + (size_t) wave * (IPW * SO) + o * SO;  // this item's exchange image
    for (int i = tid; i < 4 * TAB + 8 * LPI; i += 64 * WPB) tabL[i] = ((const cd *) a.fft8_tab)[i];
    __syncthreads();
    const cd *tw1L = tabL, *tw2L = tabL + TAB, *alL = tabL + 2 * TAB, *beL = tabL + 3 * TAB, *lpL = tabL + 4 * TAB;
    // the bins k1 this lane finishes (S_l): 16 lanes: l; 8 lanes: l and 16 - l (lane 0: 0 and 8); 4 lanes: those and 4 + l, 12 - l
    const int cs[4] = {l, LPI <= 8 ? (l ? 16 - l : 8) : 0, 4 + l, 12 - l};
    const bool l0 = l == 0;

    // (a.fix_list: the second look of stage_a = auto -- the listed items only, their number read from the device)
    const size_t n_items = a.fix_list ? (size_t) *a.fix_count : a.items;
    if (a.fix_list && blockIdx.x == 0 && tid == 0 && n_items) atomicAdd(&a.cond_count[1], (unsigned) n_items);      // running total (sdrt_refined_items)
    if (n_items == 0) return;
    const size_t ngroups = (n_items + IPW - 1) / IPW;
    for (size_t grp = (size_t) blockIdx.x * WPB + wave; grp < ngroups; grp += (size_t) gridDim.x * WPB) {
        // (lane groups beyond the last item redo it: the same values to the same places, no branch anywhere in the loop)
        const size_t item = grp * IPW + o, item_c = item < n_items ? item : n_items - 1;
        const size_t it = a.fix_list ? (size_t) a.fix_list[item_c] : post_item(a, item_c);
        // stage_a = auto: which lanes have found half of a batch of their pair products under the threshold / a non-zero one at
        // all (wave masks in scalar registers: nothing of this stays in vector registers through the transforms).  A lane's
        // products are spread evenly over the chunk (n = l + LPI k), so "half of them small" is a statement about the chunk's in-band
        // level that a few large samples -- the filter's edge transients under a strong out-of-band signal -- do not undo
        unsigned long long m_weak = 0, m_any = 0;
        // ---- pair angles: x[k] = a[2n] + i a[2n+1], n = l + LPI k (demodulation.py:25-32) -------------------------------
        cd x[P];
        if constexpr (PAIRS) {
            const cd *y = (const cd *) a.y + it * (size_t) (a.pairs_wide ? M : N);      // d[2i] conj(d[2i+1]) from the chain kernel
            if constexpr (LEAN) {
                // half of the item's loads in flight, their angles, the other half (the third wave covers what that exposes)
                static_for<0, 2>([&](auto hc) {
                    constexpr int k0 = (P / 2) * decltype(hc)::value;
                    cd pp[P / 2][2];
#pragma unroll
                    for (int k = 0; k < P / 2; ++k) { pp[k][0] = y[2 * (l + LPI * (k0 + k))]; pp[k][1] = y[2 * (l + LPI * (k0 + k)) + 1]; }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (LPI != 4) {              // (four lanes per item: float32 records, which stage_a = auto does not look at twice)
                        double big = 0.0;
                        int weak = 0;
#pragma unroll
                        for (int k = 0; k < P / 2; ++k)
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const double mag = fmax(fabs(pp[k][u].re), fabs(pp[k][u].im));
                                weak += mag < a.cond_thr ? 1 : 0;
                                big = fmax(big, mag);
                            }
                        m_weak |= __ballot(weak >= P / 2);
                        m_any |= __ballot(big > 0.0);
                    }
#pragma unroll
                    for (int k = 0; k < P / 2; ++k) x[k0 + k] = mk(atan2_pi(pp[k][0].im, pp[k][0].re), atan2_pi(pp[k][1].im, pp[k][1].re));
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
            cd pp[P][2];                                           // every load of the item in flight before the first angle
#pragma unroll
            for (int k = 0; k < P; ++k) { pp[k][0] = y[2 * (l + LPI * k)]; pp[k][1] = y[2 * (l + LPI * k) + 1]; }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LPI != 4) {
                double big = 0.0;
                int weak = 0;
#pragma unroll
                for (int k = 0; k < P; ++k)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const double mag = fmax(fabs(pp[k][u].re), fabs(pp[k][u].im));
                        weak += mag < a.cond_thr ? 1 : 0;
                        big = fmax(big, mag);
                    }
                m_weak |= __ballot(weak >= P);
                m_any |= __ballot(big > 0.0);
            }
#pragma unroll
            for (int k = 0; k < P; ++k) x[k] = mk(atan2_pi(pp[k][0].im, pp[k][0].re), atan2_pi(pp[k][1].im, pp[k][1].re));
            }

        } else {
            const cd *y = (const cd *) a.y + it * (size_t) M;
            const cd *R = a.iq_R ? (const cd *) a.iq_R + (it / a.n_chunks) * (size_t) M : nullptr;
            const cd h = a.iq_R ? ((const cd *) a.iq_off)[it % a.n_chunks] : mk(0.0, 0.0);
            // four bins at a time: sixteen samples (and their offset shares) in flight, then their eight angles
            static_for<0, P / 4>([&](auto gc) {
                constexpr int k0 = 4 * decltype(gc)::value;
                cd d[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int u = 0; u < 4; ++u) d[k][u] = load_y(y, R, h, 4 * (l + LPI * (k0 + k)) + u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    x[k0 + k] = mk(atan2_pi(d[k][0].im * d[k][1].re - d[k][0].re * d[k][1].im, d[k][0].re * d[k][1].re + d[k][0].im * d[k][1].im),
                                   atan2_pi(d[k][2].im * d[k][3].re - d[k][2].re * d[k][3].im, d[k][2].re * d[k][3].re + d[k][2].im * d[k][3].im));
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // The angles are half of the output samples (before the lowpass) and nothing else needs them after the first transform
        // step: they wait in the item's own place in the output buffer (every lane re-reads what it wrote itself, in one go at
        // the start of the lowpass; the final stores overwrite them while the lines are still in L2) instead of sitting in 64
        // registers through the transforms -- that, not the arithmetic, is what the register allocator spilled for.
        // (a.stash: a region per resident wave instead, rewritten by every group the wave works through: the same few megabytes
        // stay in L2 / the Infinity Cache, where the item's own place in the output buffer is new memory every time)
        cd *stash = a.stash ? (cd *) a.stash + ((size_t) blockIdx.x * WPB + wave) * (size_t) (IPW * H) + (size_t) o * H + l
                            : (cd *) (a.out + it * (size_t) M) + l;
        // KEEPX (round 4, with the halved loads / exchange image / parked results above): the angles stay in 64 registers through
        // the transforms after all -- the stash was 0.27 GB each way per 2^32 samples through the memory counters (rocprofv3:
        // WRITE_SIZE 0.79 GB against 0.54 GB of outputs), a third of this kernel's traffic, and the kernel is bound by that traffic
        constexpr bool KEEPX = LEAN && SDRT_P8_KEEPX;
        if constexpr (!KEEPX) {
#pragma unroll
            for (int k = 0; k < P; ++k) stash[LPI * k] = x[k];
        }
        // ---- forward: 16 points in the lane, twiddle, exchange, LPI points ---------------------------------------------------
        cd Z[NC][LPI];
        {
            cd X1[P];
            reg_fft<P, false>(x, X1);
            static_for<1, P>([&](auto kc) { constexpr int k1 = decltype(kc)::value; X1[k1] = cmul(X1[k1], tw1L[k1 * LPI + l]); });
            if constexpr (HX) {
                // bins 0-7 (series k1 = l), then bins 8-15 (series 16 - l, lane 0: 8) through the same 8 x 9 slots
                static_for<0, 2>([&](auto hc) {
                    constexpr int hh = decltype(hc)::value;
                    if constexpr (hh == 1) wave_lds_fence();       // the first half's reads are done
                    static_for<0, P / 2>([&](auto kc) { constexpr int k1 = decltype(kc)::value; X[ST * l + k1] = X1[8 * hh + k1]; });
                    wave_lds_fence();
                    cd V[LPI];
#pragma unroll
                    for (int j = 0; j < LPI; ++j) V[j] = X[ST * j + (cs[hh] & 7)];
                    reg_fft<LPI, false>(V, Z[hh]);
                });
            } else {
            static_for<0, P>([&](auto kc) { constexpr int k1 = decltype(kc)::value; X[17 * l + k1] = X1[k1]; });
            wave_lds_fence();
            cd V[NC][LPI];
#pragma unroll
            for (int j = 0; j < LPI; ++j)
#pragma unroll
                for (int ci = 0; ci < NC; ++ci) V[ci][j] = X[17 * j + cs[ci]];
            static_for<0, NC>([&](auto cc) { reg_fft<LPI, false>(V[decltype(cc)::value], Z[decltype(cc)::value]); });
            }
        }
        // ---- packed spectrum of the angles -> packed spectrum of the half-sample shifted sequence ----------------------------
        // Y[k] = alpha_k Z[k] + beta_k conj(Z[H - k]).  8 lanes: bin k = c + 16 k2 pairs with H - k = (16 - c) + 16 (7 - k2), the
        // lane's other bin series -- except in lane 0, whose series c = 0 and c = 8 each pair with themselves.  16 lanes: the
        // partner (16 - c) + 16 (15 - k2) sits in lane 16 - l of the item's row of lanes (row_mirror, then row_ror:1); lane 0
        // pairs with itself, k2 with 16 - k2.
        cd U[NC][LPI];
        {
            cd Y[NC][LPI];
            static_for<0, LPI>([&](auto kc) {
                constexpr int k2 = decltype(kc)::value;
                if constexpr (LPI <= 8) {
                    const cd q0 = Z[1][LPI - 1 - k2], q0s = Z[0][(LPI - k2) % LPI], q1 = Z[0][LPI - 1 - k2], q1s = Z[1][LPI - 1 - k2];
                    const cd p0 = mk(l0 ? q0s.re : q0.re, l0 ? q0s.im : q0.im), p1 = mk(l0 ? q1s.re : q1.re, l0 ? q1s.im : q1.im);
                    const cd a0 = alL[(0 * LPI + k2) * LPI + l], b0 = beL[(0 * LPI + k2) * LPI + l];
                    const cd a1 = alL[(1 * LPI + k2) * LPI + l], b1 = beL[(1 * LPI + k2) * LPI + l];
                    Y[0][k2] = cfma(b0, mk(p0.re, -p0.im), cmul(a0, Z[0][k2]));
                    Y[1][k2] = cfma(b1, mk(p1.re, -p1.im), cmul(a1, Z[1][k2]));
                    if constexpr (LPI == 4) {
                        // the second pair of series, 4 + l and 12 - l: partners in every lane
                        const cd p2 = Z[3][LPI - 1 - k2], p3 = Z[2][LPI - 1 - k2];
                        const cd a2 = alL[(2 * LPI + k2) * LPI + l], b2 = beL[(2 * LPI + k2) * LPI + l];
                        const cd a3 = alL[(3 * LPI + k2) * LPI + l], b3 = beL[(3 * LPI + k2) * LPI + l];
                        Y[2][k2] = cfma(b2, mk(p2.re, -p2.im), cmul(a2, Z[2][k2]));
                        Y[3][k2] = cfma(b3, mk(p3.re, -p3.im), cmul(a3, Z[3][k2]));
                    }
                } else {
                    const cd far = dpp_row<0x121>(dpp_row<0x140>(Z[0][15 - k2])), own = Z[0][(16 - k2) % 16];
                    const cd p0 = mk(l0 ? own.re : far.re, l0 ? own.im : far.im);
                    const cd a0 = alL[k2 * LPI + l], b0 = beL[k2 * LPI + l];
                    Y[0][k2] = cfma(b0, mk(p0.re, -p0.im), cmul(a0, Z[0][k2]));
                }
            });
            static_for<0, NC>([&](auto cc) { reg_fft<LPI, true>(Y[decltype(cc)::value], U[decltype(cc)::value]); });
        }
        // ---- back: twiddle, exchange, 16 points in the lane -------------------------------------------------------------------
        cd zz[P];
        {
            static_for<1, LPI>([&](auto kc) {
                constexpr int n2 = decltype(kc)::value;
#pragma unroll
                for (int ci = 0; ci < NC; ++ci) U[ci][n2] = cmul(U[ci][n2], tw2L[(ci * LPI + n2) * LPI + l]);
            });
            cd Rr[P];
            if constexpr (HX) {
                static_for<0, 2>([&](auto hc) {
                    constexpr int hh = decltype(hc)::value;
                    wave_lds_fence();                              // the reads before (forward step, first half) are done
#pragma unroll
                    for (int n2 = 0; n2 < LPI; ++n2) X[ST * n2 + (cs[hh] & 7)] = U[hh][n2];
                    wave_lds_fence();
#pragma unroll
                    for (int c = 0; c < P / 2; ++c) Rr[8 * hh + c] = X[ST * l + c];
                });
            } else {
            wave_lds_fence();                                      // the forward step's reads are done
#pragma unroll
            for (int n2 = 0; n2 < LPI; ++n2)
#pragma unroll
                for (int ci = 0; ci < NC; ++ci) X[17 * n2 + cs[ci]] = U[ci][n2];
            wave_lds_fence();
#pragma unroll
            for (int c = 0; c < P; ++c) Rr[c] = X[17 * l + c];
            }
            wave_lds_fence();                                      // ... before the next group's writes
            if constexpr (LEAN) {
                // the last butterflies by hand: the upper half of the results (blocks 8-15 of the lowpass) waits in the wave's
                // exchange image, which nothing else uses until the next group (64 lanes x 8 slots of its 576) -- in registers
                // they were what the allocator spilled, and a scratch reload in the lowpass waits for every store before it
                cd ev[P / 2], od[P / 2], E[P / 2], O[P / 2];
#pragma unroll
                for (int k = 0; k < P / 2; ++k) { ev[k] = Rr[2 * k]; od[k] = Rr[2 * k + 1]; }
                reg_fft<P / 2, true>(ev, E);
                reg_fft<P / 2, true>(od, O);
                cd *park = tabL + 4 * TAB + 8 * LPI + (size_t) wave * (IPW * SO) + lane;
                static_for<0, P / 2>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const cd t = tw_mul<P, k, true>(O[k]);
                    zz[k] = cadd(E[k], t);
                    park[64 * k] = csub8(E[k], t);
                });
            } else
            reg_fft<P, true>(Rr, zz);                              // zz[n1] = b[2n] + i b[2n+1], n = l + LPI n1
        }
        // ---- output samples 4n .. 4n+3 = a[2n], b[2n], a[2n+1], b[2n+1]; lowpass; store ------------------------------------------
        double *out = a.out + it * (size_t) M + 4 * l;
        __builtin_amdgcn_sched_barrier(0);                         // (not before the last transform has let go of its registers)
        // (three waves per SIMD: four of the stashed angle pairs in flight at a time, the others fetched as the blocks go by)
        constexpr int SD = LEAN ? 4 : P, BA = LEAN ? SDRT_P8_BLOCKS_AHEAD3 : SDRT_P8_BLOCKS_AHEAD;
        cd xs[SD];
        if constexpr (!KEEPX) {
#pragma unroll
            for (int k = 0; k < SD; ++k) xs[k] = stash[LPI * k];
        }
        __builtin_amdgcn_sched_barrier(0);
        // the constants every lane shares are fetched here, not before the loop: some forty doubles that would otherwise sit in
        // scalar registers through the transforms (and be spilled to vector lanes there)
        const Post8Tables *pu = a.p8tab;
        asm volatile("" : "+s"(pu));
        const double h0 = pu->h[0], h1 = pu->h[1], h2 = pu->h[2], h3 = pu->h[3];
        const cd p1 = mk(pu->pcp[0][0], pu->pcp[0][1]), p2 = mk(pu->pcp[1][0], pu->pcp[1][1]), p3 = mk(pu->pcp[2][0], pu->pcp[2][1]);
        const double r1 = pu->prp[0], r2 = pu->prp[1], r3 = pu->prp[2];
        const cd g0 = mk(pu->g[0][0], pu->g[0][1]), g1 = mk(pu->g[1][0], pu->g[1][1]), g2 = mk(pu->g[2][0], pu->g[2][1]), g3 = mk(pu->g[3][0], pu->g[3][1]);
        const double gr0 = pu->gr[0], gr1 = pu->gr[1], gr2 = pu->gr[2], gr3 = pu->gr[3];
        const cd pbc = mk(pu->pc_blk[0], pu->pc_blk[1]);
        const double pbr = pu->pr_blk;
        // ... and the lane's own: scan multipliers inside the quad (zero where the quad permutation hands a lane its own or an
        // earlier lane's value), from the quad before, from the block's start (tables.h: the lane rows of the transform tables)
        const cd mA1_ = lpL[0 * LPI + l], mA2_ = lpL[1 * LPI + l], nA = lpL[2 * LPI + l], mB_ = lpL[3 * LPI + l], mE_ = lpL[4 * LPI + l];
        const cd nBE = lpL[5 * LPI + l], plc_ = lpL[6 * LPI + l], plq = lpL[7 * LPI + l];
        const double nA1_ = nA.re, nA2_ = nA.im, nB_ = nBE.re, nE_ = nBE.im, plr_ = plq.re, q1_ = plq.im;
        // 16 lanes: the scan runs over the item's whole row of lanes, multipliers p^4, p^8, p^16, p^32
        const cd s1 = mk(pu->pc4[0], pu->pc4[1]), s2 = mk(pu->pc8[0], pu->pc8[1]), s4 = mk(pu->pc_q[3][0], pu->pc_q[3][1]), s8 = mk(pu->pc32[0], pu->pc32[1]);
        const double t1 = pu->pr4, t2 = pu->pr8, t4 = pu->pr_q[3], t8 = pu->pr32;
        cd Cc = mk(0.0, 0.0);                                      // state at the start of the block
        double Cr = 0.0;
        static_for<0, P>([&](auto nc) {
            constexpr int n1 = decltype(nc)::value;
            cd xv;
            if constexpr (KEEPX) xv = x[n1];
            else {
                xv = xs[n1 % SD];
                if constexpr (n1 + SD < P) xs[n1 % SD] = stash[LPI * (n1 + SD)];
            }
            // three waves per SIMD: the lane's own multipliers come from the tables in every block instead of sitting in 32
            // registers through all of them (the offset is opaque so that the reads are not merged back into one set)
            cd mA1 = mA1_, mA2 = mA2_, mB = mB_, mE = mE_, plc = plc_;
            double nA1 = nA1_, nA2 = nA2_, nB = nB_, nE = nE_, plr = plr_, q1 = q1_;
            if constexpr (LEAN) {
                int zo = 0;
                asm volatile("" : "+v"(zo));
                const cd *lp = lpL + zo + l;
                mA1 = lp[0 * LPI]; mA2 = lp[1 * LPI]; mB = lp[3 * LPI]; mE = lp[4 * LPI]; plc = lp[6 * LPI];
                const cd nA_ = lp[2 * LPI], nBE_ = lp[5 * LPI], plq_ = lp[7 * LPI];
                nA1 = nA_.re; nA2 = nA_.im; nB = nBE_.re; nE = nBE_.im; plr = plq_.re; q1 = plq_.im;
            }
            cd zv = zz[n1 % (LEAN ? P / 2 : P)];
            if constexpr (LEAN && n1 >= P / 2) {
                const cd *park = tabL + 4 * TAB + 8 * LPI + (size_t) wave * (IPW * SO) + lane;
                zv = park[64 * (n1 - P / 2)];
            }
            const double x0 = xv.re, x1 = zv.re, x2 = xv.im, x3 = zv.im;
            // the lane's run of four samples from a zero state: outputs and end state
            const double y00 = h0 * x0, y01 = fma(h0, x1, h1 * x0), y02 = fma(h0, x2, fma(h1, x1, h2 * x0));
            const double y03 = fma(h0, x3, fma(h1, x2, fma(h2, x1, h3 * x0)));
            const cd se = mk(fma(p3.re, x0, fma(p2.re, x1, fma(p1.re, x2, x3))), fma(p3.im, x0, fma(p2.im, x1, p1.im * x2)));
            const double te = fma(r3, x0, fma(r2, x1, fma(r1, x2, x3)));
            cd sc, s;
            double sr, t;
            if constexpr (LPI == 4) {
                // the item is one quad: the two quad steps are the whole scan
                sc = cfma(mA1, dpp_quad<0x90>(se), se);
                sr = fma(nA1, dpp_quad<0x90>(te), te);
                sc = cfma(mA2, dpp_quad<0x40>(sc), sc);
                sr = fma(nA2, dpp_quad<0x40>(sr), sr);
                const cd eq = dpp_quad<0x90>(sc);
                const double eqr = dpp_quad<0x90>(sr);
                s = cfma(plc, Cc, mk(q1 * eq.re, q1 * eq.im));
                t = fma(plr, Cr, q1 * eqr);
            } else if constexpr (LPI == 8) {
                // inclusive scan of the end states over the item's eight lanes: inside the quads (quad permutations never leave
                // the item) ...
                sc = cfma(mA1, dpp_quad<0x90>(se), se);                // [0,0,1,2]: lane q takes lane q-1
                sr = fma(nA1, dpp_quad<0x90>(te), te);
                sc = cfma(mA2, dpp_quad<0x40>(sc), sc);                // [0,0,0,1]: lane q takes lane q-2
                sr = fma(nA2, dpp_quad<0x40>(sr), sr);
                // ... then the first quad's total into the second: row_shr:4 written to banks 1 and 3 only (lanes 4-7, 12-15; the
                // others keep 0: nothing crosses from the neighbouring item), its lane 3 broadcast over the quad
                const cd cq = dpp_quad<0xFF>(dpp_shr4_banks13(sc));
                const double cqr = dpp_quad<0xFF>(dpp_shr4_banks13(sr));
                // state before the lane's first sample: the lanes before it in its quad, the quad before, the block's starting state
                const cd eq = dpp_quad<0x90>(sc);
                const double eqr = dpp_quad<0x90>(sr);
                s = cfma(plc, Cc, cfma(mE, cq, mk(q1 * eq.re, q1 * eq.im)));
                t = fma(plr, Cr, fma(nE, cqr, q1 * eqr));
                sc = cfma(mB, cq, sc);
                sr = fma(nB, cqr, sr);
            } else {
                // the item is a whole row of lanes: shifts inside the row fill with zeros at its start and never leave it
                sc = cfma(s1, dpp_row<0x111>(se), se);
                sr = fma(t1, dpp_row<0x111>(te), te);
                sc = cfma(s2, dpp_row<0x112>(sc), sc);
                sr = fma(t2, dpp_row<0x112>(sr), sr);
                sc = cfma(s4, dpp_row<0x114>(sc), sc);
                sr = fma(t4, dpp_row<0x114>(sr), sr);
                sc = cfma(s8, dpp_row<0x118>(sc), sc);
                sr = fma(t8, dpp_row<0x118>(sr), sr);
                s = cfma(plc, Cc, dpp_row<0x111>(sc));
                t = fma(plr, Cr, dpp_row<0x111>(sr));
            }
            // the next block starts from this block's end state (the item's last lane)
            const cd Tc = LPI == 4 ? dpp_quad<0xFF>(sc) : (LPI == 8 ? mk(swz_bcast_last8(sc.re), swz_bcast_last8(sc.im)) : dpp_row<0x15F>(sc));
            const double Tr = LPI == 4 ? dpp_quad<0xFF>(sr) : (LPI == 8 ? swz_bcast_last8(sr) : dpp_row<0x15F>(sr));
            Cc = cfma(pbc, Cc, Tc);
            Cr = fma(pbr, Cr, Tr);
            // what the starting state adds to the four outputs (dsp_processor.py:32-36: sosfilt, zero state per chunk)
            const double y0 = fma(g0.re, s.re, fma(-g0.im, s.im, fma(gr0, t, y00)));
            const double y1 = fma(g1.re, s.re, fma(-g1.im, s.im, fma(gr1, t, y01)));
            const double y2 = fma(g2.re, s.re, fma(-g2.im, s.im, fma(gr2, t, y02)));
            const double y3 = fma(g3.re, s.re, fma(-g3.im, s.im, fma(gr3, t, y03)));
            *(double2 *) (out + 4 * LPI * n1) = make_double2(y0, y1);
            *(double2 *) (out + 4 * LPI * n1 + 2) = make_double2(y2, y3);
            // a few blocks at a time: enough independent work for the pipeline, few enough intermediate values for the registers
            if constexpr (n1 % BA == BA - 1) __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (PAIRS && LPI != 4) {
            if (a.cond_list) {
                // (at the end of the item, where nothing else is alive) some lane of the item's LPI has found half of a batch of its
                // pair products under the threshold, and not all were zero (silence is exact on every path): the in-band content of
                // this (chunk, row) is near the operator image of the int8 forms -- listed for the float64 pass
                const unsigned long long grp_mask = ((LPI == 64 ? 0ull : (1ull << LPI)) - 1ull) << (o * LPI);
                if (l0 && grp * IPW + o < n_items && (m_weak & grp_mask) != 0 && (m_any & grp_mask) != 0) {
                    const unsigned slot = atomicAdd(a.cond_count, 1u);
                    a.cond_list[slot] = (int) (a.fix_list ? (size_t) a.fix_list[grp * IPW + o] : post_item(a, grp * IPW + o));
                }
            }
        }
    }
}

bool post_fft8_supported(const PostArgs &a);

size_t post_fft8_lds_bytes(int lpi, int wpb, bool pairs) {
    return ((size_t) 4 * SDRT_P8_PPL * lpi + 8 * lpi + (size_t) wpb * (64 / lpi) * (p8_halves(lpi, pairs) ? 9 : 17) * lpi) * 16;
}

template <int LPI, int WPB, bool PAIRS>
static hipError_t launch_post_fft8_p(const PostArgs &a, hipStream_t s) {
    const size_t lds = post_fft8_lds_bytes(LPI, WPB, PAIRS);
    static thread_local int done_dev = -1;                      // once per device and instantiation
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute((const void *) post_fft8_kernel<LPI, WPB, PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    constexpr size_t IPW = 64 / LPI;
    const size_t groups = (a.items + IPW - 1) / IPW;
    size_t wgs = (groups + WPB - 1) / WPB;
    const size_t cap = 256 * (LPI == 16 ? 2 : 4);               // a few workgroups per CU; waves walk over the rest
    if (wgs > cap) wgs = cap;
    if (a.stash && a.stash_bytes < wgs * WPB * 8 * 128 * 16) return hipErrorInvalidValue;      // 8 x 128 (4 x 256) complex angles per wave
    hipLaunchKernelGGL((post_fft8_kernel<LPI, WPB, PAIRS>), dim3((unsigned) wgs), dim3(64 * WPB), lds, s, a);
    return hipGetLastError();
}

// bytes of the per-wave angle stash a launch over `items` items takes (PostArgs.stash), 0: none needed (not this kernel)
size_t post_fft8_stash_bytes(const PostArgs &a) {
    if (!post_fft8_supported(a)) return 0;
    const size_t ipw = 64 / a.p8_lpi, groups = (a.items + ipw - 1) / ipw;
    const bool small = a.items <= (size_t) (64 / a.p8_lpi) * 256;
    const size_t wpb = small ? 1 : (a.p8_lpi == 16 ? 8 : 4), cap = 256 * (a.p8_lpi == 16 ? 2 : 4);
    size_t wgs = (groups + wpb - 1) / wpb;
    if (wgs > cap) wgs = cap;
    return wgs * wpb * 8 * 128 * 16;
}

bool post_fft8_supported(const PostArgs &a) {
    // with the IQ offset share still to subtract every angle takes four samples and four table entries: at 512 outputs per chunk the
    // one-wave-per-item kernel holds those loads better (0.54 against 0.63 ms per 2^32 samples), at 1024 this one does (0.085
    // against 0.099 ms per 2^29 8-bit samples; profiles/r03_experiments.txt).  SDRTERM_DIAG=post8_iq=0 / 2: never / at both sizes
    if (a.iq_R) {
        const int with_share = sdrt_diag_int("post8_iq", 1);
        if (!(with_share == 2 || (with_share == 1 && a.m == 1024))) return false;
    }
    return a.mode == SDRT_FM && a.fft8_tab && a.p8tab && ((a.m == 256 && a.p8_lpi == 4) || (a.m == 512 && a.p8_lpi == 8) || (a.m == 1024 && a.p8_lpi == 16));
}

hipError_t launch_post_fft8(const PostArgs &a, hipStream_t s) {
    if (a.p8_lpi == 16) {
        // the tables take 18 KiB, a wave's exchange image 17: eight waves share one copy of the tables
        if (a.items <= 4 * 256) return a.pairs ? launch_post_fft8_p<16, 1, true>(a, s) : launch_post_fft8_p<16, 1, false>(a, s);
        return a.pairs ? launch_post_fft8_p<16, 8, true>(a, s) : launch_post_fft8_p<16, 8, false>(a, s);
    }
    if (a.p8_lpi == 4) {
        if (a.items <= 16 * 256) return a.pairs ? launch_post_fft8_p<4, 1, true>(a, s) : launch_post_fft8_p<4, 1, false>(a, s);
        return a.pairs ? launch_post_fft8_p<4, 4, true>(a, s) : launch_post_fft8_p<4, 4, false>(a, s);
    }
    // few items: one wave per workgroup spreads them over more compute units
    if (a.items <= 8 * 256) return a.pairs ? launch_post_fft8_p<8, 1, true>(a, s) : launch_post_fft8_p<8, 1, false>(a, s);
    return a.pairs ? launch_post_fft8_p<8, 4, true>(a, s) : launch_post_fft8_p<8, 4, false>(a, s);
}
