// post_kernel: everything after the decimator, one WAVE per (row, chunk) item, no workgroup barriers after the table
// load: FM pair-angle (demodulation.py:25-32) + scipy.signal.resample m/2 -> m (demodulation.py:35-38) or AM
// (demodulation.py:41-48), then the output lowpass (dsp_processor.py:32-36,149: sosfilt with zero state per chunk) as a
// modal scan along the lanes.  Small registers and 12 m bytes of LDS per wave keep many items in flight per CU.
#include "kernels_common.hpp"

__global__ __launch_bounds__(256) void post_kernel(PostArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool fm = a.mode == SDRT_FM;
    const int N = m >> 1;
    double *hr = (double *) smem;                              // [N] (FM): hr[i] = h[(-i) mod N]
    double *wbase = hr + (fm ? N : 0) + (size_t) wave * (size_t) (fm ? m + N : m);
    double *av = wbase;                                        // [N] pair angles (FM)
    double *zb = wbase + (fm ? N : 0);                         // [m] demodulated samples at the decimated rate
    if (fm) {
        for (int i = tid; i < N; i += 256) hr[i] = a.resample_odd[i == 0 ? 0 : N - i];
        __syncthreads();
    }
    const size_t item = (size_t) blockIdx.x * 4 + wave;
    if (item >= a.items) return;
    const cd *y = (const cd *) a.y + item * (size_t) m;

    if (fm) {
        for (int j = lane; j < N; j += 64) {
            const cd p = y[2 * j], c = y[2 * j + 1];
            av[j] = atan2(p.im * c.re - p.re * c.im, p.re * c.re + p.im * c.im);    // angle(d[i] * conj(d[i+1]))
        }
        // even samples are the input, odd samples the circulant product with the half-sample Dirichlet kernel
        for (int j = lane; j < N; j += 64) zb[2 * j] = av[j];
        switch (N >> 6) {
            case 1: resample_wave<1>(av, hr, zb, N, lane); break;
            case 2: resample_wave<2>(av, hr, zb, N, lane); break;
            case 4: resample_wave<4>(av, hr, zb, N, lane); break;
            case 8: resample_wave<8>(av, hr, zb, N, lane); break;
            case 16: resample_wave<16>(av, hr, zb, N, lane); break;
            default:
                for (int j = lane; j < N; j += 64) {
                    double s0 = 0.0;
                    for (int k = 0; k < N; ++k) s0 = fma(av[k], hr[(k - j + N) % N], s0);
                    zb[2 * j + 1] = s0;
                }
        }
    } else {
        for (int k = lane; k < m; k += 64) {
            const cd v = y[k];
            zb[k] = v.re * v.re + v.im * v.im;
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
                double xc = zb[i];
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
    // modal form: y[t] = D x[t] + sum_k Re(r_k s_k(t)), s_k(t+1) = p_k s_k(t) + x[t].  Lane l owns samples [l*E, l*E+E):
    // local end state, inclusive scan along the wave, exclusive shift = state at the lane's first sample.
    const int E = ot->E, t0 = lane * E;
    cd sst[SDRT_OUT_MAXMODES];
#pragma unroll
    for (int k = 0; k < SDRT_OUT_MAXMODES; ++k) {
        sst[k] = mk(0.0, 0.0);
        if (k < nmodes) {
            const cd p = mk(ot->p[k][0], ot->p[k][1]);
            cd s = mk(0.0, 0.0);
            for (int e = 0; e < E; ++e) {
                const double x = (t0 + e < m) ? zb[t0 + e] : 0.0;
                s = cfma(p, s, mk(x, 0.0));
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
    }
    const double Dd = ot->D;
    for (int e = 0; e < E; ++e) {
        if (t0 + e < m) {
            const double x = zb[t0 + e];
            double yv = Dd * x;
#pragma unroll
            for (int k = 0; k < SDRT_OUT_MAXMODES; ++k) {
                if (k < nmodes) {
                    const cd r = mk(ot->r[k][0], ot->r[k][1]), p = mk(ot->p[k][0], ot->p[k][1]);
                    yv += r.re * sst[k].re - r.im * sst[k].im;
                    sst[k] = cfma(p, sst[k], mk(x, 0.0));
                }
            }
            out[t0 + e] = yv;
        }
    }
}

size_t post_lds_bytes(int m, int mode) {
    const size_t N = (size_t) m / 2;
    return (mode == SDRT_FM ? N + 4 * ((size_t) m + N) : 4 * (size_t) m) * sizeof(double);
}

hipError_t launch_post(const PostArgs &a, hipStream_t s) {
    const size_t lds = post_lds_bytes(a.m, a.mode);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(post_kernel, dim3((unsigned) ((a.items + 3) / 4)), dim3(256), lds, s, a);
    return hipGetLastError();
}
