// Diagnostic: does hipExtStreamCreateWithCUMask restrict where workgroups run on this box, and how do mask bits map to
// (XCC, SE, CU)?  Launches a census kernel (one wave per workgroup, spins ~20 us so that workgroups spread) on a masked
// stream and prints how many distinct CUs per XCC were used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void census(unsigned *out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long) spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char *name, hipStream_t s, unsigned *d) {
    const int n = 4096;
    CK(hipMemsetAsync(d, 0xff, n * 8, s));
    census<<<n, 512, 0, s>>>(d, 2000);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * n);
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> per[16];
    std::set<unsigned long long> all;
    for (int i = 0; i < n; ++i) {
        const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;     // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
        per[xcc].insert((se << 8) | (sh << 4) | cu);
        all.insert(((unsigned long long) xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    printf("%-28s distinct CUs %3zu; per XCC:", name, all.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", per[x].size());
    printf("\n");
}

int main() {
    unsigned *d;
    CK(hipMalloc(&d, 4096 * 8));
    hipStream_t s0;
    CK(hipStreamCreate(&s0));
    run("unmasked", s0, d);
    unsigned mA[8], mB[8];
    for (int i = 0; i < 8; ++i) { mA[i] = i < 7 ? 0xffffffffu : 0u; mB[i] = i < 7 ? 0u : 0xffffffffu; }
    hipStream_t sa, sb;
    hipError_t e = hipExtStreamCreateWithCUMask(&sa, 8, mA);
    printf("create A: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) run("mask bits 0..223", sa, d);
    e = hipExtStreamCreateWithCUMask(&sb, 8, mB);
    printf("create B: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) run("mask bits 224..255", sb, d);
    unsigned mC[8] = {0xffffu, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t sc;
    e = hipExtStreamCreateWithCUMask(&sc, 8, mC);
    if (e == hipSuccess) run("mask bits 0..15", sc, d);
    return 0;
}
