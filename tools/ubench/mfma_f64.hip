// Micro-benchmark (diagnostic, not product code): issue rate of v_mfma_f64_16x16x4_f64 on gfx950, alone and mixed with
// f64 VALU work, at 1..4 waves per SIMD.  Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CHAINS, int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void kern(double *out, unsigned long long *cyc, int iters) {
    v4d acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double v[4] = {a, b, a + b, a - b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < VALU_PER_MFMA; ++k) v[k & 3] = fma(v[k & 3], b, a);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = v[0] + v[1] + v[2] + v[3];
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// pure f64 VALU FMA stream: NACC independent accumulators, multiplier from an SGPR (wave-uniform) like a weight
template <int NACC>
__global__ __launch_bounds__(256) void kern_valu(double *out, unsigned long long *cyc, int iters, double wuni) {
    double acc[NACC];
    for (int c = 0; c < NACC; ++c) acc[c] = threadIdx.x * 1e-3 + c;
    double x = 1.0 + threadIdx.x * 1e-6;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[c] = fma(wuni, x, acc[c]);
        x += 1e-9;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < NACC; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run_valu(int wgs_per_cu) {
    int iters = 4000, nblk = 256 * wgs_per_cu;
    double *out; unsigned long long *cyc;
    (void) hipMalloc(&out, nblk * 256 * sizeof(double));
    (void) hipMalloc(&cyc, nblk * 4 * sizeof(unsigned long long));
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    kern_valu<NACC><<<nblk, 256>>>(out, cyc, iters, 1.000001);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0);
    kern_valu<NACC><<<nblk, 256>>>(out, cyc, iters, 1.000001);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk * 4);
    (void) hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += x; mean /= h.size();
    double per = mean / ((double) iters * (NACC + 1));
    double tf = 128.0 * iters * NACC * nblk * 4 / (ms * 1e-3) / 1e12;
    printf("VALU f64 FMA x%-2d accumulators          waves/SIMD=%d  cycles/instr/wave=%6.2f  cycles/instr/SIMD=%5.2f  wall=%.3f ms  TF=%.1f  clk=%.2f GHz\n",
           NACC, wgs_per_cu, per, per / wgs_per_cu, ms, tf, mean / (ms * 1e-3) / 1e9);
    (void) hipFree(out); (void) hipFree(cyc);
}

template <int CHAINS, int VALU>
void run(int wgs_per_cu, const char *name) {
    int iters = 2000, nblk = 256 * wgs_per_cu;
    double *out; unsigned long long *cyc;
    hipMalloc(&out, nblk * 256 * sizeof(double));
    hipMalloc(&cyc, nblk * 4 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<CHAINS, VALU><<<nblk, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<CHAINS, VALU><<<nblk, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += x; mean /= h.size();
    double per_mfma_wave = mean / (iters * CHAINS);
    double tf = 2048.0 * iters * CHAINS * nblk * 4 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD=%d  cycles/MFMA/wave=%7.1f  cycles/MFMA/SIMD=%6.1f  wall=%.3f ms  MFMA TF=%.1f  clk=%.2f GHz\n", name, wgs_per_cu,
           per_mfma_wave, per_mfma_wave / wgs_per_cu, ms, tf, mean / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1, 0>(w, "1 chain, no VALU");
        run<2, 0>(w, "2 chains, no VALU");
        run<4, 0>(w, "4 chains, no VALU");
        run<2, 4>(w, "2 chains + 4 f64 FMA per MFMA");
        run<2, 8>(w, "2 chains + 8 f64 FMA per MFMA");
        run<2, 16>(w, "2 chains + 16 f64 FMA per MFMA");
    }
    for (int w : {1, 2, 4, 8}) { run_valu<8>(w); run_valu<16>(w); }
    return 0;
}
