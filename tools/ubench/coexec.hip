// Micro-benchmark (diagnostic, not product code): how int8 matrix instructions and fp64 vector instructions share one SIMD
// of gfx950 at two waves per SIMD (512-thread workgroups, one per CU), under a chip-wide load like chain_i8w_kernel's.
// For each arrangement it prints wall time per launch, the in-kernel clock (s_memtime / s_memrealtime) and the core cycles
// per loop iteration, so that "cycles" and "clock the chip holds" can be told apart.
//
// One loop iteration = the per-SIMD work of one chain_i8w iteration: 2 x 184 v_mfma_i32_16x16x64_i8 and 2 x NV fp64 FMAs.
//   mode 0  every wave: matrix instructions only
//   mode 1  every wave: fp64 FMAs only
//   mode 2  every wave: 23 groups of [8 matrix instructions][NV/23 FMAs]                (the shipped kernel's shape)
//   mode 3  waves 0-3 matrix only (2 x 184), waves 4-7 FMAs only (2 x NV)                (roles split by wave >= 4)
//   mode 4  as 2, waves 4-7 start half a group later (stagger)
//   mode 5  as 0 on v_mfma_i32_32x32x32_i8 (92 per 184)
//   mode 6  as 3 on v_mfma_i32_32x32x32_i8
//   mode 7  as 2 on v_mfma_i32_32x32x32_i8 (groups of [4][NV/23])
//   mode 8  as 3, the matrix waves fetch every A fragment from LDS (ds_read_b128 per matrix instruction)
//   mode 9  as 0 with LDS fragments
//   mode 10 as 2 with LDS fragments: groups of [8 x (ds_read_b128, matrix instruction)][NV/23 FMAs]   (the shipped kernel with its fetches)
//   mode 11 ONE wave per SIMD (256 threads) working on two tiles: every fragment is read once and feeds two matrix instructions,
//           the FMAs of both tiles interleaved between them at instruction granularity
//   mode 12 as 11, but every matrix instruction fetches its own fragment (what sharing the fragment is worth on its own)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NM = 184;          // 16x16x64 matrix instructions per wave and tile
constexpr int GROUPS = 23;

template <int N>
__device__ __forceinline__ void fmas(double (&x)[8], double c1, double c2) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i & 7] = fma(x[i & 7], c1, c2);
}

template <int MODE, int NV>
__global__ __launch_bounds__(512, 1) void k(const int *__restrict__ src, double *out, unsigned long long *st, int iters) {
    __shared__ v4i lds[64 * 40];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 40; i += (int) blockDim.x) lds[i] = ((const v4i *) src)[(i * 7 + blockIdx.x) & 4095];
    __syncthreads();
    v4i a[8], b[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = ((const v4i *) src)[(lane + 64 * i + wave * 17) & 4095];
    b[0] = ((const v4i *) src)[(lane * 3 + 1) & 4095];
    b[1] = ((const v4i *) src)[(lane * 5 + 2) & 4095];
    v4i acc[8], accB[8];
    v16i acc32[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = accB[i] = (v4i) {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc32[i][j] = 0;
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.0 + 1e-3 * lane + i;
    const double c1 = 0.999999 + 1e-9 * lane, c2 = 1e-7 * (wave + 1);
    v4i F[8], G[8];
    int fi = 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { F[i] = lds[i * 64 + lane]; G[i] = lds[(i + 20) * 64 + lane]; }
    constexpr bool M32 = MODE == 5 || MODE == 6 || MODE == 7;
    constexpr bool SPLIT = MODE == 3 || MODE == 6 || MODE == 8;
    constexpr bool LDSF = MODE == 8 || MODE == 9 || MODE == 10;
    auto mm = [&](auto nc) {                       // n 16x16x64 instructions (or n/2 32x32x32)
        constexpr int n = decltype(nc)::value;
        if constexpr (M32) {
#pragma unroll
            for (int i = 0; i < n / 2; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 7], b[i & 1], acc32[i & 3], 0, 0, 0);
        } else if constexpr (MODE == 10) {
            // as the shipped kernel fetches them: a ring of eight fragment registers, each reloaded for the next group as soon as
            // its matrix instruction has been issued
#pragma unroll
            for (int i = 0; i < n; ++i) {
                acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(F[i & 7], b[i & 1], acc[i & 7], 0, 0, 0);
                F[i & 7] = lds[((i + fi) % 40) * 64 + lane];
            }
            fi = (fi + 8) % 40;
        } else if constexpr (LDSF) {
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const v4i f = lds[(i % 40) * 64 + lane];
                acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f, b[i & 1], acc[i & 7], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < n; ++i) acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 7], b[i & 1], acc[i & 7], 0, 0, 0);
        }
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SPLIT) {
        if (wave < 4) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 2 * GROUPS; ++g) { mm(std::integral_constant<int, 8>{}); __builtin_amdgcn_sched_barrier(0); }
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 2 * GROUPS; ++g) { fmas<NV / GROUPS>(x, c1, c2); __builtin_amdgcn_sched_barrier(0); }
            }
        }
    } else if constexpr (MODE == 11 || MODE == 12) {
        // one wave per SIMD, two tiles: per group 8 fragments x 2 matrix instructions and 2 x NV/23 FMAs, interleaved
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(F[i], b[0], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    fmas<3>(x, c1, c2);
                    __builtin_amdgcn_sched_barrier(0);
                    accB[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(MODE == 12 ? G[i] : F[i], b[1], accB[i], 0, 0, 0);
                    // the registers take the next group's fragments (mode 12: one fetch per matrix instruction)
                    F[i] = lds[((g * 8 + i + 8) % 40) * 64 + lane];
                    if constexpr (MODE == 12) G[i] = lds[((g * 8 + i + 28) % 40) * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                    fmas<2>(x, c1, c2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                fmas<2 * (NV / GROUPS) - 40>(x, c1, c2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if constexpr (MODE == 0 || MODE == 5 || MODE == 9) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) { mm(std::integral_constant<int, 8>{}); __builtin_amdgcn_sched_barrier(0); }
        }
    } else if constexpr (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) { fmas<NV / GROUPS>(x, c1, c2); __builtin_amdgcn_sched_barrier(0); }
        }
    } else {
        if (MODE == 4 && wave >= 4) { fmas<NV / GROUPS>(x, c1, c2); __builtin_amdgcn_sched_barrier(0); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                __builtin_amdgcn_s_setprio(1);
                mm(std::integral_constant<int, 8>{});
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);
                fmas<NV / GROUPS>(x, c1, c2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + (double) (acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]) + (double) (accB[i][0] + accB[i][3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += (double) acc32[i][j];
    out[(size_t) blockIdx.x * 512 + tid] = s;
    if (lane == 0) {
        unsigned long long *p = st + ((size_t) blockIdx.x * 8 + wave) * 2;
        p[0] = t1 - t0;
        p[1] = r1 - r0;
    }
}

template <int MODE, int NV>
void run(const char *name, const int *src, double *out, unsigned long long *st, int iters, double seconds) {
    const int threads = (MODE == 11 || MODE == 12) ? 256 : 512, nwaves = threads / 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = 256;
    // load the chip for `seconds` with back-to-back launches, then time 20 more
    k<MODE, NV><<<wgs, threads>>>(src, out, st, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE, NV><<<wgs, threads>>>(src, out, st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms1 = 0; CK(hipEventElapsedTime(&ms1, e0, e1));
    int warm = std::max(1, (int) (seconds * 1e3 / ms1));
    for (int i = 0; i < warm; ++i) k<MODE, NV><<<wgs, threads>>>(src, out, st, iters);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) k<MODE, NV><<<wgs, threads>>>(src, out, st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(wgs * 8 * 2);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc_lo, cyc_hi;
    for (int w = 0; w < wgs * 8; ++w) {
        if ((w & 7) >= nwaves) continue;
        if (nwaves == 4) cyc_hi.push_back((double) h[2 * w] / iters);
        clk.push_back((double) h[2 * w] / (double) h[2 * w + 1] * 100.0);
        ((w & 7) < 4 ? cyc_lo : cyc_hi).push_back((double) h[2 * w] / iters);
    }
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-44s %8.4f ms/launch  clock %5.0f MHz  cycles/iter waves0-3 %7.0f waves4-7 %7.0f  ns/iter %7.1f\n", name, ms / 20, med(clk),
           med(cyc_lo), med(cyc_hi), ms / 20 * 1e6 / iters);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int iters = 128;
    int *src; double *out; unsigned long long *st;
    CK(hipMalloc(&src, 4096 * 16));
    CK(hipMalloc(&out, 256 * 512 * 8));
    CK(hipMalloc(&st, 256 * 8 * 2 * 8));
    std::vector<int> h(4096 * 4);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (int) s; }
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    constexpr int NV = 23 * 22;      // 506 fp64 FMAs per wave and tile ~ the vector side of chain_i8w (2100 cycles)
    printf("per iteration and SIMD: 368 x 16x16x64 int8 matrix instructions (5888 cycles at 16 each), %d fp64 FMAs (%d cycles at 4 each)\n", 2 * NV, 2 * NV * 4);
    run<0, NV>("0 matrix only, all waves", src, out, st, iters, seconds);
    run<1, NV>("1 fp64 FMA only, all waves", src, out, st, iters, seconds);
    run<2, NV>("2 groups [8 matrix][22 FMA], all waves", src, out, st, iters, seconds);
    run<4, NV>("4 as 2, waves 4-7 staggered", src, out, st, iters, seconds);
    run<3, NV>("3 roles: waves 0-3 matrix, 4-7 FMA", src, out, st, iters, seconds);
    run<5, NV>("5 matrix only, 32x32x32", src, out, st, iters, seconds);
    run<7, NV>("7 groups [4 x 32x32x32][22 FMA]", src, out, st, iters, seconds);
    run<6, NV>("6 roles, 32x32x32", src, out, st, iters, seconds);
    run<9, NV>("9 matrix only, A fragments from LDS", src, out, st, iters, seconds);
    run<8, NV>("8 roles, A fragments from LDS", src, out, st, iters, seconds);
    run<10, NV>("10 as 2 + a fragment fetch per matrix instr.", src, out, st, iters, seconds);
    run<11, NV>("11 one wave/SIMD, 2 tiles share a fragment", src, out, st, iters, seconds);
    run<12, NV>("12 one wave/SIMD, a fetch per matrix instr.", src, out, st, iters, seconds);
    run<10, NV>("10 again", src, out, st, iters, seconds);
    run<11, NV>("11 again", src, out, st, iters, seconds);
    return 0;
}
