// post_kernel: everything after the decimator, one WAVE per (row, chunk) item, no workgroup barriers after the table
// load: FM pair-angle (demodulation.py:25-32) + scipy.signal.resample m/2 -> m (demodulation.py:35-38) or AM
// (demodulation.py:41-48), then the output lowpass (dsp_processor.py:32-36,149: sosfilt with zero state per chunk) as a
// modal scan along the lanes.  Small registers and 12 m bytes of LDS per wave keep many items in flight per CU.
#include "kernels_common.hpp"


// Output lowpass of one item in modal form, one wave: y[t] = D x[t] + sum_k Re(r_k s_k(t)), s_k(t+1) = p_k s_k(t) + x[t].
// Lane l owns samples [l*E, l*E+E): local end state, inclusive scan along the wave, exclusive shift = state at the lane's
// first sample.  zb: [m] demodulated samples in LDS at positions zpos(i) (one pad double per 8 samples: the lanes' strided
// runs then fall on different banks); the results overwrite zb and leave with coalesced stores.  out: [m] (global).
__device__ __forceinline__ int zpos(int i) { return i + (i >> 3); }
__device__ __forceinline__ int zlen(int m) { return m + (m >> 3) + 1; }

template <int NM>
__device__ __forceinline__ void modal_lowpass_body(const OutTables *__restrict__ ot, double *zb, double *out, const int m, const int lane) {
    const int E = ot->E, t0 = lane * E;
    cd sst[NM > 0 ? NM : 1], pk[NM > 0 ? NM : 1], rk[NM > 0 ? NM : 1];
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        pk[k] = mk(ot->p[k][0], ot->p[k][1]);
        rk[k] = mk(ot->r[k][0], ot->r[k][1]);
        cd s = mk(0.0, 0.0);
        for (int e = 0; e < E; ++e) {
            const double x = (t0 + e < m) ? zb[zpos(t0 + e)] : 0.0;
            s = cfma(pk[k], s, mk(x, 0.0));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int d = 1 << i;
            cd t = mk(__shfl_up(s.re, d), __shfl_up(s.im, d));
            if (lane < d) t = mk(0.0, 0.0);
            s = cfma(mk(ot->pstep[k][i][0], ot->pstep[k][i][1]), t, s);
        }
        cd ex = mk(__shfl_up(s.re, 1), __shfl_up(s.im, 1));
        if (lane == 0) ex = mk(0.0, 0.0);
        sst[k] = ex;
    }
    const double Dd = ot->D;
    for (int e = 0; e < E; ++e) {
        if (t0 + e < m) {
            const double x = zb[zpos(t0 + e)];
            double yv = Dd * x;
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                yv += rk[k].re * sst[k].re - rk[k].im * sst[k].im;
                sst[k] = cfma(pk[k], sst[k], mk(x, 0.0));
            }
            zb[zpos(t0 + e)] = yv;
        }
    }
    for (int i = lane; i < m; i += 64) out[i] = zb[zpos(i)];
}

// The same for the usual case -- one complex pole pair + one real pole (ellip(3, ...), dsp_processor.py:39-45), E = m/64
// samples per lane known at compile time: the lane's samples stay in registers between the two passes, the real pole runs
// in real arithmetic, the first four scan steps are DPP row shifts.
template <int EE>
__device__ __forceinline__ void modal_lowpass_c1r1(const OutTables *__restrict__ ot, double *zb, double *out, const int lane) {
    constexpr int m = 64 * EE;
    const int t0 = lane * EE;
    double xs[EE];
#pragma unroll
    for (int e = 0; e < EE; ++e) xs[e] = zb[zpos(t0 + e)];
    const cd pc = mk(ot->p[0][0], ot->p[0][1]), rc = mk(ot->r[0][0], ot->r[0][1]);
    const double pr = ot->p[1][0], rr = ot->r[1][0];
    cd sc = mk(0.0, 0.0);
    double sr = 0.0;
#pragma unroll
    for (int e = 0; e < EE; ++e) {
        sc = cfma(pc, sc, mk(xs[e], 0.0));
        sr = fma(pr, sr, xs[e]);
    }
    // inclusive scan of the lanes' end states along the wave: inside the 16-lane rows with DPP shifts (multipliers
    // p^(E 2^i)), then the carry of the rows before (their end states chained in scalar registers) into every lane
    static_for<0, 4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sc = cfma(mk(ot->pstep[0][i][0], ot->pstep[0][i][1]), dpp_row<0x110 + (1 << i)>(sc), sc);
        sr = fma(ot->pstep[1][i][0], dpp_row<0x110 + (1 << i)>(sr), sr);
    });
    {
        auto at = [](double v, int l) {
            return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
        };
        const cd P16 = mk(ot->pstep[0][4][0], ot->pstep[0][4][1]);
        const double Q16 = ot->pstep[1][4][0];
        const cd c1 = mk(at(sc.re, 15), at(sc.im, 15));
        const cd c2 = cfma(P16, c1, mk(at(sc.re, 31), at(sc.im, 31)));
        const cd c3 = cfma(P16, c2, mk(at(sc.re, 47), at(sc.im, 47)));
        const double d1 = at(sr, 15), d2 = fma(Q16, d1, at(sr, 31)), d3 = fma(Q16, d2, at(sr, 47));
        const int row = lane >> 4;
        const cd cin = row == 0 ? mk(0.0, 0.0) : (row == 1 ? c1 : (row == 2 ? c2 : c3));
        const double din = row == 0 ? 0.0 : (row == 1 ? d1 : (row == 2 ? d2 : d3));
        sc = cfma(mk(ot->prow[0][lane & 15][0], ot->prow[0][lane & 15][1]), cin, sc);
        sr = fma(ot->prow[1][lane & 15][0], din, sr);
    }
    sc = mk(__shfl_up(sc.re, 1), __shfl_up(sc.im, 1));      // exclusive: the state at the lane's first sample
    sr = __shfl_up(sr, 1);
    if (lane == 0) { sc = mk(0.0, 0.0); sr = 0.0; }
    const double Dd = ot->D;
#pragma unroll
    for (int e = 0; e < EE; ++e) {
        const double x = xs[e];
        zb[zpos(t0 + e)] = fma(Dd, x, fma(rc.re, sc.re, fma(-rc.im, sc.im, rr * sr)));
        sc = cfma(pc, sc, mk(x, 0.0));
        sr = fma(pr, sr, x);
    }
    for (int i = lane; i < m; i += 64) out[i] = zb[zpos(i)];
}

__device__ __forceinline__ void modal_lowpass_wave(const OutTables *__restrict__ ot, double *zb, double *out, const int m, const int lane) {
    if (ot->nmodes == 2 && ot->ncomplex == 1) {
        if (m == 512) { modal_lowpass_c1r1<8>(ot, zb, out, lane); return; }
        if (m == 1024) { modal_lowpass_c1r1<16>(ot, zb, out, lane); return; }
        if (m == 2048) { modal_lowpass_c1r1<32>(ot, zb, out, lane); return; }
    }
    switch (ot->nmodes) {
        case 0: modal_lowpass_body<0>(ot, zb, out, m, lane); break;
        case 1: modal_lowpass_body<1>(ot, zb, out, m, lane); break;
        case 2: modal_lowpass_body<2>(ot, zb, out, m, lane); break;
        case 3: modal_lowpass_body<3>(ot, zb, out, m, lane); break;
        default: modal_lowpass_body<4>(ot, zb, out, m, lane); break;
    }
}

// blockDim.x / 64 items per workgroup (launch_post picks as many waves as fit LDS)
__global__ __launch_bounds__(256) void post_kernel(PostArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wpb = blockDim.x >> 6;
    const bool fm = a.mode == SDRT_FM;
    const int N = m >> 1;
    double *hr = (double *) smem;                              // [N] (FM): hr[i] = h[(-i) mod N]
    double *wbase = hr + (fm ? N : 0) + (size_t) wave * (size_t) (fm ? zlen(m) + N : zlen(m));
    double *av = wbase;                                        // [N] pair angles (FM)
    double *zb = wbase + (fm ? N : 0);                         // [m] demodulated samples at the decimated rate
    if (fm) {
        for (int i = tid; i < N; i += blockDim.x) hr[i] = a.resample_odd[i == 0 ? 0 : N - i];
        __syncthreads();
    }
    const size_t witem = (size_t) blockIdx.x * wpb + wave;
    if (witem >= a.items) return;
    const size_t item = post_item(a, witem);
    const cd *y = (const cd *) a.y + item * (size_t) ((a.pairs && !a.pairs_wide) ? N : m);
    const cd *R = a.iq_R ? (const cd *) a.iq_R + (item / a.n_chunks) * (size_t) m : nullptr;
    const cd h = a.iq_R ? ((const cd *) a.iq_off)[item % a.n_chunks] : mk(0.0, 0.0);

    if (fm && a.pairs) {
        for (int j = lane; j < N; j += 64) {
            const cd z = y[j];                                                      // d[2j] * conj(d[2j+1]) from the chain kernel
            av[j] = atan2_pi(z.im, z.re);
        }
    } else if (fm) {
        for (int j = lane; j < N; j += 64) {
            const cd p = load_y(y, R, h, 2 * j), c = load_y(y, R, h, 2 * j + 1);
            av[j] = atan2_pi(p.im * c.re - p.re * c.im, p.re * c.re + p.im * c.im);    // angle(d[i] * conj(d[i+1]))
        }
    }
    if (fm) {
        // even samples are the input, odd samples the circulant product with the half-sample Dirichlet kernel
        for (int j = lane; j < N; j += 64) zb[zpos(2 * j)] = av[j];
        switch (N >> 6) {
            case 1: resample_wave<1>(av, hr, zb, N, lane); break;
            case 2: resample_wave<2>(av, hr, zb, N, lane); break;
            case 4: resample_wave<4>(av, hr, zb, N, lane); break;
            case 8: resample_wave<8>(av, hr, zb, N, lane); break;
            case 16: resample_wave<16>(av, hr, zb, N, lane); break;
            case 32: resample_wave<32>(av, hr, zb, N, lane); break;
            default:
                for (int j = lane; j < N; j += 64) {
                    double s0 = 0.0;
                    for (int k = 0; k < N; ++k) s0 = fma(av[k], hr[(k - j + N) % N], s0);
                    zb[zpos(2 * j + 1)] = s0;
                }
        }
    } else {
        for (int k = lane; k < m; k += 64) {
            const cd v = load_y(y, R, h, k);
            zb[zpos(k)] = v.re * v.re + v.im * v.im;
        }
    }

    double *out = a.out + item * (size_t) m;
    const OutTables *__restrict__ ot = a.otab;
    const int nmodes = ot->nmodes;
    if (nmodes < 0) {
        // sequential transposed-direct-form-II cascade (any SOS); only when the modal form is not available
        if (lane == 0) {
            double st[2 * SDRT_MAX_OUT_SECTIONS];
            for (int s = 0; s < 2 * SDRT_MAX_OUT_SECTIONS; ++s) st[s] = 0.0;
            for (int i = 0; i < m; ++i) {
                double xc = zb[zpos(i)];
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
        return;
    }
    modal_lowpass_wave(ot, zb, out, m, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// post_fft_kernel: the FM post stage with scipy.signal.resample in the form the reference itself uses (demodulation.py:35-38
// -> scipy: rfft, zero-pad, irfft): odd output samples = IDFT(DFT(a) * half-sample shift), even ones = a.  One wave per
// item (the angle sequence as a complex sequence with zero imaginary part).  Transforms: Stockham radix-4 passes (+ one radix-2
// pass when N is not a power of four) through a per-wave LDS buffer, in place: a wave executes all loads of a pass before
// its stores.  N = 256 B.  O(N log N) instead of the O(N^2) circulant product of post_kernel.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// position of element i in the per-wave transform buffer: one pad element per four, so that the stride-4 stores of the
// first pass fall on different banks
__device__ __forceinline__ int xpos(int i) { return i + (i >> 2); }
__device__ __forceinline__ cd csub(cd a, cd b) { return mk(a.re - b.re, a.im - b.im); }
template <bool INV> __device__ __forceinline__ cd cmul_tw(cd v, cd w) {      // v * w (forward) or v * conj(w) (inverse)
    if constexpr (INV) return mk(v.re * w.re + v.im * w.im, v.im * w.re - v.re * w.im);
    else return mk(v.re * w.re - v.im * w.im, v.re * w.im + v.im * w.re);
}

// One radix-4 Stockham pass over N points held as x[b][k] = X[t + k N/4], t = lane + 64 b; sub-transform length n = N/s.
// Results go to store(index, value).
template <int B, int S_, bool INV, class Store>
__device__ __forceinline__ void radix4_pass(cd (&x)[B][4], const cd *tw, const int lane, Store store) {
    constexpr int N = 256 * B, n = N / S_;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int t = lane + 64 * b, p = t / S_, q = t % S_;
        const cd a0 = x[b][0], a1 = x[b][1], a2 = x[b][2], a3 = x[b][3];
        const cd t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
        // forward: y1 = t1 - i t3, y3 = t1 + i t3; inverse: the other way round
        const cd jt3 = INV ? mk(-t3.im, t3.re) : mk(t3.im, -t3.re);
        cd y0 = cadd(t0, t2), y1 = cadd(t1, jt3), y2 = csub(t0, t2), y3 = csub(t1, jt3);
        if constexpr (n > 4) {
            y1 = cmul_tw<INV>(y1, tw[p * S_]);
            y2 = cmul_tw<INV>(y2, tw[2 * p * S_]);
            y3 = cmul_tw<INV>(y3, tw[3 * p * S_]);
        }
        const int o = q + S_ * 4 * p;
        store(o, y0);
        store(o + S_, y1);
        store(o + 2 * S_, y2);
        store(o + 3 * S_, y3);
    }
}

template <int B, bool INV, class Load, class Store>
__device__ __forceinline__ void fft_wave(cd (&x)[B][4], cd *X, const cd *tw, const int lane, Load load_first, Store store_last) {
    constexpr int N = 256 * B;
    constexpr bool R2 = (B == 2 || B == 8);          // 512 = 2 * 4^4, 2048 = 2 * 4^5: a final radix-2 pass
    static_assert(B == 1 || B == 2 || B == 4 || B == 8, "N = 256, 512, 1024 or 2048");
    auto ldX = [&]() {
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) x[b][k] = X[xpos(lane + 64 * b + k * (N / 4))];
    };
    auto stX = [&](int o, cd v) { X[xpos(o)] = v; };
    load_first();
    radix4_pass<B, 1, INV>(x, tw, lane, stX);
    ldX();
    radix4_pass<B, 4, INV>(x, tw, lane, stX);
    ldX();
    radix4_pass<B, 16, INV>(x, tw, lane, stX);
    ldX();
    if constexpr (B == 1) {
        radix4_pass<B, 64, INV>(x, tw, lane, store_last);
    } else {
        radix4_pass<B, 64, INV>(x, tw, lane, stX);
        if constexpr (B == 4) {
            ldX();
            radix4_pass<B, 256, INV>(x, tw, lane, store_last);
        }
        if constexpr (B == 8) {
            ldX();
            radix4_pass<B, 256, INV>(x, tw, lane, stX);
        }
        if constexpr (R2) {
            // n = 2, s = N/2: y[q] = a + b, y[q + N/2] = a - b for q < N/2; 2B items per lane
            cd u[2 * B][2];
#pragma unroll
            for (int c = 0; c < 2 * B; ++c) { u[c][0] = X[xpos(lane + 64 * c)]; u[c][1] = X[xpos(lane + 64 * c + N / 2)]; }
#pragma unroll
            for (int c = 0; c < 2 * B; ++c) {
                store_last(lane + 64 * c, cadd(u[c][0], u[c][1]));
                store_last(lane + 64 * c + N / 2, csub(u[c][0], u[c][1]));
            }
        }
    }
}
}  // namespace

template <int B, int WPB, bool PAIRS>
__global__ __launch_bounds__(64 * WPB) void post_fft_kernel(PostArgs a) {
    constexpr int N = 256 * B, M = 2 * N;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cd *tw = (cd *) smem;                                       // [N] exp(-2 pi i k/N)
    const cd *__restrict__ sh = (const cd *) a.fft_shift;       // [N] half-sample shift / N: read once per item, from L2
    // per wave: the padded transform buffer (xpos) and, in the same bytes, the demodulated samples for the lowpass (zpos):
    // the last inverse pass has loaded all of X before it stores into zb
    constexpr int XL = N + N / 4 + 1;                           // 16-byte elements; covers zlen(M) doubles as well
    static_assert((size_t) XL * 16 >= (size_t) (M + (M >> 3) + 1) * 8, "zb must fit the transform buffer");
    cd *X = tw + N + (size_t) wave * XL;
    double *zb = (double *) X;
    for (int i = tid; i < N; i += 64 * WPB) tw[i] = ((const cd *) a.fft_tw)[i];
    __syncthreads();
    const OutTables *__restrict__ ot = a.otab;
    // one item per wave and turn: every item's result is a function of that item alone, bit for bit (pairing two items
    // into one complex transform would make the last bits of one depend on the other)
    for (size_t witem = (size_t) blockIdx.x * WPB + wave; witem < a.items; witem += (size_t) gridDim.x * WPB) {
        const size_t item = post_item(a, witem);
        const cd *y = (const cd *) a.y + item * (size_t) ((PAIRS && !a.pairs_wide) ? N : M);
        const cd *R = a.iq_R ? (const cd *) a.iq_R + (item / a.n_chunks) * (size_t) M : nullptr;
        const cd h = a.iq_R ? ((const cd *) a.iq_off)[item % a.n_chunks] : mk(0.0, 0.0);
        cd x[B][4];
        double ev[B][4];
        auto angles = [&]() {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = lane + 64 * b + k * (N / 4);
                    // angle(d[i] * conj(d[i+1])) (demodulation.py:25-32); pairs: the product comes from the chain kernel
                    double an;
                    if constexpr (PAIRS) {
                        const cd z = y[j];
                        an = atan2_pi(z.im, z.re);
                    } else {
                        const cd p = load_y(y, R, h, 2 * j), c = load_y(y, R, h, 2 * j + 1);
                        an = atan2_pi(p.im * c.re - p.re * c.im, p.re * c.re + p.im * c.im);
                    }
                    x[b][k] = mk(an, 0.0);
                    ev[b][k] = an;                              // even output samples are the input
                }
        };
        // forward transform -> X (natural order)
        fft_wave<B, false>(x, X, tw, lane, angles, [&](int o, cd v) { X[xpos(o)] = v; });
        // spectrum * shift, inverse transform; its last pass leaves the odd output samples (the imaginary parts are rounding noise)
        auto shifted = [&]() {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = lane + 64 * b + k * (N / 4);
                    x[b][k] = cmul(X[xpos(f)], sh[f]);
                }
        };
        fft_wave<B, true>(x, X, tw, lane, shifted, [&](int o, cd v) { zb[zpos(2 * o + 1)] = v.re; });
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) zb[zpos(2 * (lane + 64 * b + k * (N / 4)))] = ev[b][k];
        modal_lowpass_wave(ot, zb, a.out + item * (size_t) M, M, lane);
    }
}

// The same stage with the N real pair angles packed two to a complex number (round 4; 2048 / 4096 outputs per chunk: decimation
// 16 / 8 of 16-bit records): z[n] = a[2n] + i a[2n+1] through ONE transform of H = N/2 = 256 B points each way, the packed
// spectrum of the half-sample shifted sequence bin by bin from the packed spectrum of the angles (sdrt_build_fft_resample_packed:
// Wp[k] = alpha_k Z[k] + beta_k conj(Z[H-k])), IDFT_H(Wp) = o[2n] + i o[2n+1] = two odd output samples per point.  Half the
// transform work and half the LDS of post_fft_kernel: three workgroups per CU instead of one.
template <int B, int WPB, bool PAIRS>
__global__ __launch_bounds__(64 * WPB) void post_fftp_kernel(PostArgs a) {
    constexpr int H = 256 * B, N = 2 * H, M = 2 * N;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cd *tw = (cd *) smem;                                       // [H] exp(-2 pi i k/H)
    const double4 *__restrict__ ab = (const double4 *) a.fftp_ab;      // [H] alpha, beta: read once per item, from L2
    // per wave: the padded transform buffer (xpos) and, afterwards, the demodulated samples for the lowpass (zpos) -- M doubles
    // need more room than H complex numbers: the buffer is sized for them
    constexpr int ZL = M + (M >> 3) + 1;                        // doubles
    constexpr int XL = (ZL + 1) / 2;                            // 16-byte elements
    static_assert(XL >= H + H / 4 + 1, "the transform buffer must fit");
    cd *X = tw + H + (size_t) wave * XL;
    double *zb = (double *) X;
    for (int i = tid; i < H; i += 64 * WPB) tw[i] = ((const cd *) a.fftp_tw)[i];
    __syncthreads();
    const OutTables *__restrict__ ot = a.otab;
    for (size_t witem = (size_t) blockIdx.x * WPB + wave; witem < a.items; witem += (size_t) gridDim.x * WPB) {
        const size_t item = post_item(a, witem);
        const cd *y = (const cd *) a.y + item * (size_t) ((PAIRS && !a.pairs_wide) ? N : M);
        const cd *R = a.iq_R ? (const cd *) a.iq_R + (item / a.n_chunks) * (size_t) M : nullptr;
        const cd h = a.iq_R ? ((const cd *) a.iq_off)[item % a.n_chunks] : mk(0.0, 0.0);
        cd x[B][4];
        cd ev[B][4];                                            // the angles themselves: the even output samples
        auto angle_of = [&](int j) {
            // angle(d[i] * conj(d[i+1])) (demodulation.py:25-32); pairs: the product comes from the chain kernel
            if constexpr (PAIRS) {
                const cd z = y[j];
                return atan2_pi(z.im, z.re);
            } else {
                const cd p = load_y(y, R, h, 2 * j), c = load_y(y, R, h, 2 * j + 1);
                return atan2_pi(p.im * c.re - p.re * c.im, p.re * c.re + p.im * c.im);
            }
        };
        auto angles = [&]() {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n = lane + 64 * b + k * (H / 4);
                    x[b][k] = mk(angle_of(2 * n), angle_of(2 * n + 1));
                    ev[b][k] = x[b][k];
                }
        };
        fft_wave<B, false>(x, X, tw, lane, angles, [&](int o, cd v) { X[xpos(o)] = v; });
        auto shifted = [&]() {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = lane + 64 * b + k * (H / 4);
                    const cd Z = X[xpos(f)], Zm = X[xpos((H - f) & (H - 1))];
                    const double4 c = ab[f];
                    // alpha Z + beta conj(Zm)
                    x[b][k] = mk(fma(c.x, Z.re, fma(-c.y, Z.im, fma(c.z, Zm.re, c.w * Zm.im))),
                                 fma(c.x, Z.im, fma(c.y, Z.re, fma(c.w, Zm.re, -(c.z * Zm.im)))));
                }
        };
        // the last inverse pass has loaded all of X before it stores into zb (same bytes)
        fft_wave<B, true>(x, X, tw, lane, shifted, [&](int o, cd v) { zb[zpos(4 * o + 1)] = v.re; zb[zpos(4 * o + 3)] = v.im; });
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = lane + 64 * b + k * (H / 4);
                zb[zpos(4 * n)] = ev[b][k].re;
                zb[zpos(4 * n + 2)] = ev[b][k].im;
            }
        modal_lowpass_wave(ot, zb, a.out + item * (size_t) M, M, lane);
    }
}

template <int B, int WPB, bool PAIRS>
static hipError_t launch_post_fftp_p(const PostArgs &a, hipStream_t s) {
    constexpr int H = 256 * B, M = 4 * H;
    const size_t lds = (size_t) (H + WPB * ((M + (M >> 3) + 2) / 2)) * 16;
    static thread_local int done_dev = -1;                      // once per device and instantiation
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute((const void *) post_fftp_kernel<B, WPB, PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    size_t wgs = (a.items + WPB - 1) / WPB;
    const size_t cap = 256 * 8;
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL((post_fftp_kernel<B, WPB, PAIRS>), dim3((unsigned) wgs), dim3(64 * WPB), lds, s, a);
    return hipGetLastError();
}
template <int B, int WPB>
static hipError_t launch_post_fftp(const PostArgs &a, hipStream_t s) {
    return a.pairs ? launch_post_fftp_p<B, WPB, true>(a, s) : launch_post_fftp_p<B, WPB, false>(a, s);
}

template <int B, int WPB, bool PAIRS>
static hipError_t launch_post_fft_p(const PostArgs &a, hipStream_t s) {
    constexpr int N = 256 * B;
    const size_t lds = (size_t) (N + WPB * (N + N / 4 + 1)) * 16;
    static thread_local int done_dev = -1;                      // once per device and instantiation
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute((const void *) post_fft_kernel<B, WPB, PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    size_t wgs = (a.items + WPB - 1) / WPB;
    const size_t cap = 256 * 8;                                  // a few workgroups per CU; waves walk over the rest
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL((post_fft_kernel<B, WPB, PAIRS>), dim3((unsigned) wgs), dim3(64 * WPB), lds, s, a);
    return hipGetLastError();
}
template <int B, int WPB>
static hipError_t launch_post_fft(const PostArgs &a, hipStream_t s) {
    return a.pairs ? launch_post_fft_p<B, WPB, true>(a, s) : launch_post_fft_p<B, WPB, false>(a, s);
}

static size_t post_lds_bytes_w(int m, int mode, int wpb) {
    const size_t N = (size_t) m / 2, zl = (size_t) m + (size_t) (m >> 3) + 1;
    return (mode == SDRT_FM ? N + wpb * (zl + N) : wpb * zl) * sizeof(double);
}
// waves (= items) per workgroup of post_kernel: as many of 4, 2, 1 as fit the 160 KiB of LDS; 0: not even one
int post_waves_per_block(int m, int mode) {
    for (int w = 4; w >= 1; w >>= 1)
        if (post_lds_bytes_w(m, mode, w) <= 160 * 1024) return w;
    return 0;
}
size_t post_lds_bytes(int m, int mode) {
    const int w = post_waves_per_block(m, mode);
    return post_lds_bytes_w(m, mode, w ? w : 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// FFT taps of the plots (plots/spectrum_analyzer_plot.py:74-88, plots/waterfall_plot.py:47-58,90-100): both take the reader's
// chunk, shift it by the plot's own offset (demodulation.shiftFreq) and transform it.
//   tap_decode_kernel   raw records -> complex128, optional normalisation, NCO                        (one pass, any encoding)
//   tap_fft_kernel      spectrum analyser: log10 |fftshift(fft(y) / n)|^2 over the WHOLE chunk: Stockham radix-2 passes of
//                       one workgroup per chunk through a ping-pong buffer in global memory (n = 8192 .. 65536)
//   tap_stft_kernel     waterfall: 10 log10 |ShortTimeFFT(Kaiser(5), 256, hop 128, mfft 1024, centered, magnitude)|: one wave
//                       per frame on the in-wave 1024-point transform of the resampler
// ---------------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void tap_decode_kernel(TapArgs a) {
    const size_t total = a.chunks * (size_t) a.n;
    typedef typename Enc<DT>::raw raw_t;
    const raw_t *rp = (const raw_t *) a.raw;
    cd *u = (cd *) a.u;
    const cd *tile = (const cd *) a.nco_tile, *phase = (const cd *) a.nco_phase;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t) gridDim.x * 256) {
        double xr = decode<DT>(rp[2 * i], a.swap != 0), xi = decode<DT>(rp[2 * i + 1], a.swap != 0);
        if (a.normalize) xr = fma(xr + a.norm_bias, a.norm_scale, 0.0), xi *= a.norm_scale;
        const size_t t = i % (size_t) a.n;
        if (tile) {
            const cd c = cmul(tile[t & 63], phase[t >> 6]);
            const double tr = xr * c.re - xi * c.im;
            xi = xr * c.im + xi * c.re;
            xr = tr;
        }
        u[i] = mk(xr, xi);
    }
}

__global__ __launch_bounds__(256) void tap_fft_kernel(TapArgs a) {
    const int n = a.n, tid = threadIdx.x;
    const cd *tw = (const cd *) a.tw;                          // exp(-2 pi i k/n), k < n
    cd *x = (cd *) a.u + (size_t) blockIdx.x * n;               // the chunk; overwritten
    cd *y = (cd *) a.scratch + (size_t) blockIdx.x * n;
    int len = n, s = 1;
    while (len > 1) {
        const int half = len >> 1;
        for (int j = tid; j < n / 2; j += 256) {
            const int p = j / s, q = j - p * s;
            const cd va = x[q + s * p], vb = x[q + s * (p + half)], w = tw[(size_t) p * (n / len)];
            const cd d = mk(va.re - vb.re, va.im - vb.im);
            y[q + s * 2 * p] = mk(va.re + vb.re, va.im + vb.im);
            y[q + s * (2 * p + 1)] = mk(d.re * w.re - d.im * w.im, d.re * w.im + d.im * w.re);
        }
        __syncthreads();                                       // the workgroup's global stores are visible to it after the barrier
        cd *t = x; x = y; y = t;
        len = half;
        s <<= 1;
    }
    double *out = a.out + (size_t) blockIdx.x * n;
    const double inv = 1.0 / ((double) n * (double) n);
    for (int k = tid; k < n; k += 256) {
        const cd v = x[k];
        out[(k + n / 2) & (n - 1)] = log10((v.re * v.re + v.im * v.im) * inv);      // fftshift; norm='forward'; amp * amp
    }
}

template <int WPB>
__global__ __launch_bounds__(64 * WPB) void tap_stft_kernel(TapArgs a) {
    constexpr int B = 4, N = 1024;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cd *tw = (cd *) smem;
    constexpr int XL = N + N / 4 + 1;
    cd *X = tw + N + (size_t) wave * XL;
    for (int i = tid; i < N; i += 64 * WPB) tw[i] = ((const cd *) a.tw)[i];
    __syncthreads();
    const int n = a.n, cols = n / a.hop + 1;
    const size_t frames = a.chunks * (size_t) cols;
    for (size_t f = (size_t) blockIdx.x * WPB + wave; f < frames; f += (size_t) gridDim.x * WPB) {
        const size_t chunk = f / cols;
        const int p = (int) (f % cols);
        const cd *u = (const cd *) a.u + chunk * (size_t) n;
        const long start = (long) p * a.hop - a.nperseg / 2;    // frame p is centred on sample p * hop; zeros outside the chunk
        cd x[B][4];
        auto frame = [&]() {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int i = lane + 64 * b;                    // window index (nperseg = 256: element k = 0 of every lane's four)
                const long t = start + i;
                cd v = mk(0.0, 0.0);
                if (i < a.nperseg && t >= 0 && t < n) { const double w = a.window[i]; v = mk(u[t].re * w, u[t].im * w); }
                x[b][0] = v;
                x[b][1] = x[b][2] = x[b][3] = mk(0.0, 0.0);
            }
        };
        double *img = a.out + chunk * (size_t) N * cols + p;
        fft_wave<B, false>(x, X, tw, lane, frame, [&](int o, cd v) {
            img[(size_t) ((o + N / 2) & (N - 1)) * cols] = 5.0 * log10(v.re * v.re + v.im * v.im);     // 10 log10 |S|, fft_mode 'centered'
        });
    }
}

hipError_t launch_tap_decode(const TapArgs &a, int dt, hipStream_t s) {
    const dim3 grid(2048);
    switch (dt) {
        case SDRT_S8: hipLaunchKernelGGL(tap_decode_kernel<SDRT_S8>, grid, dim3(256), 0, s, a); break;
        case SDRT_U8: hipLaunchKernelGGL(tap_decode_kernel<SDRT_U8>, grid, dim3(256), 0, s, a); break;
        case SDRT_S16: hipLaunchKernelGGL(tap_decode_kernel<SDRT_S16>, grid, dim3(256), 0, s, a); break;
        case SDRT_U16: hipLaunchKernelGGL(tap_decode_kernel<SDRT_U16>, grid, dim3(256), 0, s, a); break;
        case SDRT_S32: hipLaunchKernelGGL(tap_decode_kernel<SDRT_S32>, grid, dim3(256), 0, s, a); break;
        case SDRT_U32: hipLaunchKernelGGL(tap_decode_kernel<SDRT_U32>, grid, dim3(256), 0, s, a); break;
        case SDRT_F32: hipLaunchKernelGGL(tap_decode_kernel<SDRT_F32>, grid, dim3(256), 0, s, a); break;
        case SDRT_F64: hipLaunchKernelGGL(tap_decode_kernel<SDRT_F64>, grid, dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_tap_fft(const TapArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(tap_fft_kernel, dim3((unsigned) a.chunks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_tap_stft(const TapArgs &a, hipStream_t s) {
    constexpr int WPB = 2, N = 1024;
    const size_t lds = (size_t) (N + WPB * (N + N / 4 + 1)) * 16;
    hipError_t e = hipFuncSetAttribute((const void *) tap_stft_kernel<WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    const size_t frames = a.chunks * (size_t) (a.n / a.hop + 1);
    size_t wgs = (frames + WPB - 1) / WPB;
    if (wgs > 2048) wgs = 2048;
    hipLaunchKernelGGL(tap_stft_kernel<WPB>, dim3((unsigned) wgs), dim3(64 * WPB), lds, s, a);
    return hipGetLastError();
}

// realOutput / imagOutput (demodulation.py:51-68) of the decimated samples; no output filter follows (dsp_processor.py:130-138)
__global__ __launch_bounds__(256) void part_kernel(const double *__restrict__ y, double *__restrict__ out, size_t count, int imag) {
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t) gridDim.x * 256) out[i] = y[2 * i + imag];
}

bool post_takes_pairs(const PostArgs &a) {
    if (a.mode != SDRT_FM) return false;
    if (post_fft8_supported(a)) return true;
    if (a.fft_tw && a.fft_shift && a.fft_ok && (a.m == 512 || a.m == 1024 || a.m == 2048 || a.m == 4096)) return true;
    if ((a.m == 8192 || a.m == 16384 || a.m == 32768) && a.fft_tw && a.fft_shift && a.fftp_tw && a.fftp_ab && !a.iq_R) return true;    // post_bigfftp_kernel
    if (a.m >= 8192 && post_big_supported(a.m, a.mode)) return false;
    return post_waves_per_block(a.m, a.mode) != 0;
}

bool post_takes_iq_share(const PostArgs &a) {
    if (a.mode == SDRT_RE || a.mode == SDRT_IM) return false;                                // part_kernel
    if (post_fft8_supported(a)) return true;
    if (a.mode == SDRT_FM && a.fft_tw && a.fft_shift && a.fft_ok && (a.m == 512 || a.m == 1024 || a.m == 2048 || a.m == 4096)) return true;
    if (a.m >= 8192 && post_big_supported(a.m, a.mode)) return false;                       // post_big_kernel
    return post_waves_per_block(a.m, a.mode) != 0;                                           // post_kernel
}

hipError_t launch_post(const PostArgs &a, hipStream_t s) {
    if (a.mode == SDRT_RE || a.mode == SDRT_IM) {
        const size_t count = a.items * (size_t) a.m;
        const size_t wgs = (count + 255) / 256;
        hipLaunchKernelGGL(part_kernel, dim3((unsigned) (wgs < 4096 ? wgs : 4096)), dim3(256), 0, s, a.y, a.out, count, a.mode == SDRT_IM ? 1 : 0);
        return hipGetLastError();
    }
    if (post_fft8_supported(a)) return launch_post_fft8(a, s);                                // kernels_post8.hip
    if (post_bigfft_supported(a)) return launch_post_bigfft(a, s);                            // kernels_seq.hip: a workgroup per item, 2048 outputs and up
    // FFT form of the resampler where it applies: FM, N = m/2 in {256, 512, 1024}, output lowpass in modal form
    if (a.mode == SDRT_FM && a.fftp_tw && a.fftp_ab && a.fft_ok) {
        switch (a.m) {                                         // the real angles packed two to a complex number: half-size transforms
            // (a wave's share of LDS is its m output samples for the lowpass, 18 / 37 KB: eight / three waves fill a CU)
            case 2048: return launch_post_fftp<2, 8>(a, s);
            case 4096: return launch_post_fftp<4, 3>(a, s);
        }
    }
    if (a.mode == SDRT_FM && a.fft_tw && a.fft_shift && a.fft_ok) {
        switch (a.m) {
            case 512: return launch_post_fft<1, 4>(a, s);
            case 1024: return launch_post_fft<2, 4>(a, s);
            case 2048: return launch_post_fft<4, 2>(a, s);
            case 4096: return launch_post_fft<8, 1>(a, s);
        }
    }
    if (a.m >= 8192 && post_big_supported(a.m, a.mode)) return launch_post_big(a, s);       // kernels_seq.hip
    const int wpb = post_waves_per_block(a.m, a.mode);
    if (wpb == 0) return hipErrorInvalidValue;                      // sdrt_create refuses such a configuration
    const size_t lds = post_lds_bytes_w(a.m, a.mode, wpb);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(post_kernel, dim3((unsigned) ((a.items + wpb - 1) / wpb)), dim3(64 * wpb), lds, s, a);
    return hipGetLastError();
}

// ---- --smooth-output: Savitzky-Golay operator on the written samples (dsp_processor.py:159-160), one wave per item -------
__global__ __launch_bounds__(256) void smooth_kernel(SmoothArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *z = (double *) smem + (size_t) wave * m;
    const size_t item = (size_t) blockIdx.x * (blockDim.x >> 6) + wave;
    if (item >= a.items) return;
    double *out = a.out + item * (size_t) m;
    for (int i = lane; i < m; i += 64) z[i] = out[i];
    for (int i = lane; i < m; i += 64) {
        double acc = 0.0;
        if (i < a.n_edge) {
            const double *row = a.left + (size_t) i * a.window;
            for (int k = 0; k < a.window; ++k) acc = fma(row[k], z[k], acc);
        } else if (i >= m - a.n_edge) {
            const double *row = a.right + (size_t) (i - (m - a.n_edge)) * a.window;
            for (int k = 0; k < a.window; ++k) acc = fma(row[k], z[m - a.window + k], acc);
        } else {
            for (int k = 0; k < a.ntaps; ++k) {
                const int j = i + a.tap_first + k;
                if (j >= 0 && j < m) acc = fma(a.taps[k], z[j], acc);
            }
        }
        out[i] = acc;
    }
}

int smooth_waves_per_block(int m) {
    for (int w = 4; w >= 1; w >>= 1)
        if ((size_t) w * m * sizeof(double) <= 160 * 1024) return w;
    return 0;
}

hipError_t launch_smooth(const SmoothArgs &a, hipStream_t s) {
    const int wpb = smooth_waves_per_block(a.m);
    if (wpb == 0) return hipErrorInvalidValue;                      // sdrt_set_smoothing refuses such a configuration
    const size_t lds = (size_t) wpb * a.m * sizeof(double);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) smooth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(smooth_kernel, dim3((unsigned) ((a.items + wpb - 1) / wpb)), dim3(64 * wpb), lds, s, a);
    return hipGetLastError();
}
