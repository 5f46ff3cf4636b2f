// float32 records in the reference's OWN arithmetic (sdrt_config.faithful_f32): unpack .. decimate in complex64, bit for bit.
//
// For float32 records the reference's `re + 1j*im` is complex64 (src/misc/read_file.py:100-101), its x / y buffers take that
// dtype (src/dsp/dsp_processor.py:171-177), shiftFreq's complex128 product is stored back as complex64
// (src/dsp/demodulation.py:71-79 into x) and scipy.signal.decimate then runs the whole zero-phase IIR in single precision
// (dsp_processor.py:147: sos cast to x.dtype, sosfiltfilt -> odd extension by 27, sosfilt_zi * x[0], _sosfilt forwards, the same
// backwards, [27:-27][::q]).  Single-precision rounding in the cascade's high-gain sections leaves ~3e-4 (q = 64) of noise on the
// decimated samples, and that noise is a deterministic function of every rounding: the only way to agree with the reference to
// 1e-6 is to perform the same IEEE operations in the same order.  So this file is compiled with -ffp-contract=off and restates
//   numpy's complex128 multiply        re = fma(ar, br, -(ai bi)), im = fma(ar, bi, ai br)   (its AVX2 / AVX-512 loop)
//   scipy's _sosfilt (float complex)   x_new = b0 x + z0;  z0 = (b1 x - a1 x_new) + z1;  z1 = b2 x - a2 x_new   per section
// componentwise (the coefficients' imaginary parts are zero).  The recurrence is sequential in time: one (chunk, row) sequence
// per LANE, 64 sequences per wave, re / im as the two halves of packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32).
//
//   f32c_fused_kernel  (default) forward pass straight from the records and the shift table, its outputs stored [group][t][lane];
//                      backward pass over them, every q-th sample out as complex128 for the demodulation stage
//   f32c_prep_kernel   (SDRTERM_DIAG=f32c_fused=0) records -> shifted complex64 samples, odd-extended, transposed to [group][t][lane]
//   f32c_seq_kernel    (the same switch) forward pass in place over those, backward pass
//
// Everything after the decimator is float64 in the reference as well (the numba kernels' signatures are complex128,
// demodulation.py:25,41): the demodulation stage is the one every other encoding uses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels.h"

#pragma clang fp contract(off)

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kEdge = 27;       // scipy sosfiltfilt: padlen = 3 * (2 * n_sections + 1 - ...) = 27 for four full sections
constexpr int kNS = 4;          // cheby1(8): four second-order sections

__device__ __forceinline__ float f32_record(const uint32_t *p, int swap) {
    uint32_t u = *p;
    if (swap) u = __builtin_bswap32(u);
    return __uint_as_float(u);
}

// shifted sample k of (chunk, row): complex64(complex128(z[k]) * shift[row][k])  -- numpy's multiply, then the store into x
__device__ __forceinline__ f2 shifted(const uint32_t *chunk_raw, const double *shift_row, int k, int swap) {
    const float re = f32_record(chunk_raw + 2 * (size_t) k, swap), im = f32_record(chunk_raw + 2 * (size_t) k + 1, swap);
    f2 v;
    if (shift_row) {
        const double ar = (double) re, ai = (double) im, br = shift_row[2 * (size_t) k], bi = shift_row[2 * (size_t) k + 1];
        const double pi = ai * bi, pr = ai * br;
        v.x = (float) fma(ar, br, -pi);
        v.y = (float) fma(ar, bi, pr);
    } else {
        v.x = re;
        v.y = im;
    }
    return v;
}

// ext index e of the odd extension (scipy _arraytools.odd_ext): 2 x[0] - x[27 - e] | x[e - 27] | 2 x[n-1] - x[n - 2 - (e - 27 - n)]
__device__ __forceinline__ f2 extended(const uint32_t *chunk_raw, const double *shift_row, int e, int n, int swap) {
    if (e < kEdge) {
        const f2 x0 = shifted(chunk_raw, shift_row, 0, swap), xk = shifted(chunk_raw, shift_row, kEdge - e, swap);
        return 2.0f * x0 - xk;
    }
    if (e >= kEdge + n) {
        const f2 xe = shifted(chunk_raw, shift_row, n - 1, swap), xk = shifted(chunk_raw, shift_row, n - 2 - (e - kEdge - n), swap);
        return 2.0f * xe - xk;
    }
    return shifted(chunk_raw, shift_row, e - kEdge, swap);
}

// one workgroup per (group of 64 items, tile of 64 ext samples): wave w takes items w, w + 4, .. of the group, its lanes the
// tile's samples (coalesced reads of the records and of the shift table); the tile leaves transposed, 512 bytes per time step
__global__ __launch_bounds__(256) void f32c_prep_kernel(const F32cArgs a) {
    __shared__ f2 tile[64][65];
    // the group index runs fastest: the workgroups in flight at any time work on the same 64 time steps, so the shift table's
    // columns for those steps (rows x 64 x 16 bytes: 256 KiB at 256 rows) are read from memory once and from L2 by every other
    // group -- the table would otherwise cost 16 bytes per row-sample, as much as everything else the pass moves
    const int N = a.n + 2 * kEdge;
    const size_t groups = (a.n_chunks * (size_t) a.rows + 63) / 64;
    const size_t g = blockIdx.x % groups;
    const int t0 = (int) (blockIdx.x / groups) * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t items = a.n_chunks * (size_t) a.rows;
    for (int j = wv; j < 64; j += 4) {
        const size_t it = g * 64 + j;
        f2 v = {0.0f, 0.0f};
        const int e = t0 + lane;
        if (it < items && e < N) {
            const size_t chunk = it / a.rows;
            const int row = (int) (it - chunk * a.rows);
            const uint32_t *cr = (const uint32_t *) a.raw + chunk * (size_t) a.n * 2;
            const double *sr = a.shift ? a.shift + (size_t) row * a.n * 2 : nullptr;
            v = extended(cr, sr, e, a.n, a.swap);
        }
        tile[lane][j] = v;
    }
    __syncthreads();
    f2 *w = (f2 *) a.work + g * (size_t) N * 64;
    for (int tt = wv; tt < 64; tt += 4)
        if (t0 + tt < N) w[(size_t) (t0 + tt) * 64 + lane] = tile[tt][lane];
}

// The same when the rows are a multiple of 64 (the many-row --simo layouts): a group is 64 consecutive rows of ONE chunk, so a
// workgroup keeps its (row group, time tile) share of the shift table in registers -- sixteen entries per thread -- and walks over
// kCPB chunks with it: the table is read once per kCPB chunks instead of once per chunk (it was half of this pass's traffic, from L2),
// and a chunk's records once per wave instead of once per row.
constexpr int kCPB = 16;
__global__ __launch_bounds__(256) void f32c_prep_rows_kernel(const F32cArgs a) {
    __shared__ f2 tile[64][65];
    const int N = a.n + 2 * kEdge, tiles = (N + 63) / 64, RQ = a.rows / 64;
    const int rq = (int) (blockIdx.x % RQ);
    const size_t rest = blockIdx.x / RQ;
    const int t0 = (int) (rest % tiles) * 64;
    const size_t c0 = (rest / tiles) * kCPB, c1 = c0 + kCPB < a.n_chunks ? c0 + kCPB : a.n_chunks;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = t0 + lane, k = e - kEdge;
    const bool middle = k >= 0 && k < a.n;              // a sample of the chunk itself (the extension takes two: the general way)
    double tr[16], ti[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double *sr = a.shift + ((size_t) (rq * 64 + wv + 4 * i) * a.n + (middle ? k : 0)) * 2;
        tr[i] = sr[0];
        ti[i] = sr[1];
    }
    for (size_t c = c0; c < c1; ++c) {
        const uint32_t *cr = (const uint32_t *) a.raw + c * (size_t) a.n * 2;
        double ar = 0.0, ai = 0.0;
        if (middle) { ar = (double) f32_record(cr + 2 * (size_t) k, a.swap); ai = (double) f32_record(cr + 2 * (size_t) k + 1, a.swap); }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f2 v = {0.0f, 0.0f};
            if (middle) {
                const double pi = ai * ti[i], pr = ai * tr[i];
                v.x = (float) fma(ar, tr[i], -pi);
                v.y = (float) fma(ar, ti[i], pr);
            } else if (e < N) {
                v = extended(cr, a.shift + (size_t) (rq * 64 + wv + 4 * i) * a.n * 2, e, a.n, a.swap);
            }
            tile[lane][wv + 4 * i] = v;
        }
        __syncthreads();
        f2 *w = (f2 *) a.work + (c * RQ + rq) * (size_t) N * 64;
        for (int tt = wv; tt < 64; tt += 4)
            if (t0 + tt < N) w[(size_t) (t0 + tt) * 64 + lane] = tile[tt][lane];
        __syncthreads();
    }
}

struct Sos {
    float b0[kNS], b1[kNS], b2[kNS], a1[kNS], a2[kNS];
};

// scipy/signal/_sosfilt.pyx: _sosfilt_float, one sample through the cascade (both components at once)
__device__ __forceinline__ f2 cascade(const Sos &c, f2 x, f2 (&z0)[kNS], f2 (&z1)[kNS]) {
#pragma unroll
    for (int s = 0; s < kNS; ++s) {
        const f2 xn = c.b0[s] * x + z0[s];
        z0[s] = (c.b1[s] * x - c.a1[s] * xn) + z1[s];
        z1[s] = c.b2[s] * x - c.a2[s] * xn;
        x = xn;
    }
    return x;
}

constexpr int kU = 8;           // time steps per unrolled block: their loads in flight together

// one wave per group of 64 sequences, lane = sequence
__global__ __launch_bounds__(64) void f32c_seq_kernel(const F32cArgs a) {
    const size_t g = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = a.n + 2 * kEdge;
    f2 *w = (f2 *) a.work + g * (size_t) N * 64 + lane;
    Sos c;
#pragma unroll
    for (int s = 0; s < kNS; ++s) {
        c.b0[s] = a.sos[s][0]; c.b1[s] = a.sos[s][1]; c.b2[s] = a.sos[s][2]; c.a1[s] = a.sos[s][4]; c.a2[s] = a.sos[s][5];
    }
    f2 z0[kNS], z1[kNS];
    // forward: zi * ext[0] (sosfiltfilt: zi * x_0), the cascade over the whole extended sequence, in place
    {
        const f2 x0 = w[0];
#pragma unroll
        for (int s = 0; s < kNS; ++s) { z0[s] = a.zi[s][0] * x0; z1[s] = a.zi[s][1] * x0; }
    }
    int t = 0;
    for (; t + kU <= N; t += kU) {
        f2 v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = w[(size_t) (t + u) * 64];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = cascade(c, v[u], z0, z1);
#pragma unroll
        for (int u = 0; u < kU; ++u) w[(size_t) (t + u) * 64] = v[u];
    }
    f2 last = {0.0f, 0.0f};
    for (; t < N; ++t) {
        last = cascade(c, w[(size_t) t * 64], z0, z1);
        w[(size_t) t * 64] = last;
    }
    if (N % kU == 0) last = w[(size_t) (N - 1) * 64];
    // backward: zi * y[-1], the cascade over the reversed forward output; ext index 27 + k q is output k ([27:-27][::q])
    const size_t items = a.n_chunks * (size_t) a.rows, it = g * 64 + lane;
    const bool live = it < items;
    const size_t chunk = live ? it / a.rows : 0, row = live ? it - chunk * a.rows : 0;
    double *yo = a.ydec + (row * a.n_chunks + chunk) * (size_t) a.m * 2;
#pragma unroll
    for (int s = 0; s < kNS; ++s) { z0[s] = a.zi[s][0] * last; z1[s] = a.zi[s][1] * last; }
    const int qm = a.q - 1;         // q is a power of two (sdrt_create)
    auto emit = [&](int tt, f2 v) {
        const int k = tt - kEdge;
        if (k >= 0 && k < a.n && (k & qm) == 0 && live) {
            double2 o;
            o.x = (double) v.x;
            o.y = (double) v.y;
            *(double2 *) (yo + 2 * (size_t) (k / a.q)) = o;
        }
    };
    t = N - 1;
    for (; t - kU + 1 >= 0; t -= kU) {
        f2 v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = w[(size_t) (t - u) * 64];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            v[u] = cascade(c, v[u], z0, z1);
            emit(t - u, v[u]);
        }
    }
    for (; t >= 0; --t) emit(t, cascade(c, w[(size_t) t * 64], z0, z1));
}


// ---- the many-row layouts (rows a multiple of 64) WITHOUT the sequences in memory (round 5) -------------
// The two kernels above move 32 bytes per row-sample (the complex64 sequence written by the preparation pass, read and written by
// the forward pass, read by the backward pass) and were bound by exactly that.  Here a wave is (chunk, group of 64 rows), lane = row:
//   * a shifted sample is recomputed where it is needed from the chunk's record (one address for the whole wave) and the shift table
//     TRANSPOSED to [row group][k][lane] (F32cArgs.shift_t: 1 KiB per wave and time step, coalesced; the waves in flight walk the
//     table at about the same k and blockIdx % 8 = XCD sees one row group when there are four, so the table comes from L2 / MALL);
//   * the forward pass keeps nothing but the cascade's state every kB steps (64 bytes per lane: 4 bytes per row-sample at kB = 16);
//   * the backward pass walks the blocks from the last to the first: the forward recurrence again from the block's checkpoint (the
//     same operations on the same values: the same bits), its kB outputs in registers, then the backward recurrence over them.
// Three recurrences instead of two and two shifted-sample computations instead of one, for 8 bytes per row-sample instead of 32.
constexpr int kB = 16;
static_assert(kB >= 16, "a group's checkpoints (8 values per kB steps) live in its share of the work buffer (one value per step)");

// (TAB: 1 the table is there, 0 it is not, decided once per block by the caller: a test per sample kept a block's loads from being
// issued together -- 303 branches instead of 196 in the many-row form, 20.5 -> 26 ms)
template <bool EDGE, int TAB>
__device__ __forceinline__ f2 fused_sample(const uint32_t *cr, const double2 *tab, const int ts, int e, int n, int swap, f2 x0, f2 xe) {
    int k = e - kEdge;
    if constexpr (EDGE) {
        if (e < kEdge) k = kEdge - e;
        else if (e >= kEdge + n) k = 2 * n + kEdge - 2 - e;
    }
    const float re = f32_record(cr + 2 * (size_t) k, swap), im = f32_record(cr + 2 * (size_t) k + 1, swap);
    f2 v;
    if constexpr (TAB != 0) {
        const double2 b = tab[(size_t) k * ts];
        const double ar = (double) re, ai = (double) im;
        const double pi = ai * b.y, pr = ai * b.x;
        v.x = (float) fma(ar, b.x, -pi);
        v.y = (float) fma(ar, b.y, pr);
    } else {
        v.x = re;
        v.y = im;
    }
    if constexpr (EDGE) {
        if (e < kEdge) v = 2.0f * x0 - v;
        else if (e >= kEdge + n) v = 2.0f * xe - v;
    }
    return v;
}

// CK = false: the forward outputs go to the work buffer instead ([t][lane], 8 bytes per row-sample written and read back) and the
// backward pass reads them: two recurrences instead of three for 16 bytes per row-sample instead of 8 -- packed fp32 instructions
// issue at half the rate the peak figures assume on this chip, so the checkpointed form is bound by its third recurrence.
// GEN = true: any number of rows (and none: no shift) -- a lane's item is (chunk, row) = divmod(g 64 + lane, rows) as in the two
// kernels above, its records and its row of the [rows][n] table read through its own pointers: 8 / 16 bytes per lane and step,
// but whole 128-byte lines per lane over 16 / 8 steps, every byte of them used.
template <bool CK, bool GEN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void f32c_fused_kernel(const F32cArgs a) {
    const size_t g = blockIdx.x;
    const int lane = threadIdx.x;
    const int n = a.n, N = n + 2 * kEdge, nblk = (N + kB - 1) / kB, RQ = a.rows / 64;
    size_t chunk;
    int row;
    bool live = true;
    const double2 *tab;
    if constexpr (GEN) {
        const size_t items = a.n_chunks * (size_t) a.rows, it = g * 64 + lane;
        live = it < items;
        chunk = live ? it / a.rows : 0;                 // (a lane past the last item walks item 0 and stores nothing)
        row = live ? (int) (it - chunk * a.rows) : 0;
        tab = a.shift ? (const double2 *) a.shift + (size_t) row * n : nullptr;
    } else {
        chunk = g / RQ;
        const int rg = (int) (g - chunk * RQ);
        row = rg * 64 + lane;
        tab = (const double2 *) a.shift_t + (size_t) rg * n * 64 + lane;
    }
    const int ts = GEN ? 1 : 64;
    const uint32_t *cr = (const uint32_t *) a.raw + chunk * (size_t) n * 2;
    f2 *ck = (f2 *) a.work + g * (size_t) N * 64 + lane;
    const int swap = a.swap;
    Sos c;
#pragma unroll
    for (int s = 0; s < kNS; ++s) {
        c.b0[s] = a.sos[s][0]; c.b1[s] = a.sos[s][1]; c.b2[s] = a.sos[s][2]; c.a1[s] = a.sos[s][4]; c.a2[s] = a.sos[s][5];
    }
    const f2 zero = {0.0f, 0.0f};
    auto sample = [&](int e) {
        if constexpr (GEN) {                    // (tab: uniform, from a kernel argument)
            return tab ? fused_sample<false, 1>(cr, tab, ts, e, n, swap, zero, zero) : fused_sample<false, 0>(cr, tab, ts, e, n, swap, zero, zero);
        } else
            return fused_sample<false, 1>(cr, tab, ts, e, n, swap, zero, zero);
    };
    const f2 x0 = sample(kEdge), xe = sample(kEdge + n - 1);
    // a block of kB extended samples: the ones inside the chunk without the extension's cases
    auto block_of = [&](auto with_tab, int e0, f2 (&v)[kB]) {
        constexpr int TAB = decltype(with_tab)::value;
        if (e0 >= kEdge && e0 + kB <= kEdge + n) {
#pragma unroll
            for (int u = 0; u < kB; ++u) v[u] = fused_sample<false, TAB>(cr, tab, ts, e0 + u, n, swap, x0, xe);
        } else {
#pragma unroll
            for (int u = 0; u < kB; ++u) v[u] = e0 + u < N ? fused_sample<true, TAB>(cr, tab, ts, e0 + u, n, swap, x0, xe) : zero;
        }
    };
    auto block = [&](int e0, f2 (&v)[kB]) {
        if constexpr (GEN) {
            if (tab) block_of(std::integral_constant<int, 1>{}, e0, v);
            else block_of(std::integral_constant<int, 0>{}, e0, v);
        } else if (e0 >= kEdge && e0 + kB <= kEdge + n) {           // (spelled out: through block_of the compiler schedules this form 15 % slower)
#pragma unroll
            for (int u = 0; u < kB; ++u) v[u] = fused_sample<false, 1>(cr, tab, ts, e0 + u, n, swap, x0, xe);
        } else {
#pragma unroll
            for (int u = 0; u < kB; ++u) v[u] = e0 + u < N ? fused_sample<true, 1>(cr, tab, ts, e0 + u, n, swap, x0, xe) : zero;
        }
    };
    f2 z0[kNS], z1[kNS];
    {
        const f2 e0v = 2.0f * x0 - sample(2 * kEdge);      // ext[0] = 2 x[0] - x[27]
#pragma unroll
        for (int s = 0; s < kNS; ++s) { z0[s] = a.zi[s][0] * e0v; z1[s] = a.zi[s][1] * e0v; }
    }
    f2 last = zero;
    for (int blk = 0; blk < nblk; ++blk) {
        if constexpr (CK) {
            f2 *p = ck + (size_t) blk * (2 * kNS * 64);
#pragma unroll
            for (int s = 0; s < kNS; ++s) { p[s * 64] = z0[s]; p[(kNS + s) * 64] = z1[s]; }
        }
        const int e0 = blk * kB;
        f2 v[kB];
        block(e0, v);
        if (e0 + kB <= N) {                     // (one straight line of kB cascades: the sections of neighbouring steps overlap)
#pragma unroll
            for (int u = 0; u < kB; ++u) v[u] = cascade(c, v[u], z0, z1);
            last = v[kB - 1];
            if constexpr (!CK) {
#pragma unroll
                for (int u = 0; u < kB; ++u) ck[(size_t) (e0 + u) * 64] = v[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < kB; ++u)
                if (e0 + u < N) {
                    last = cascade(c, v[u], z0, z1);
                    if constexpr (!CK) ck[(size_t) (e0 + u) * 64] = last;
                }
        }
    }
    // backward: zi * y[-1]; ext index 27 + k q is output k ([27:-27][::q])
    double *yo = a.ydec + ((size_t) row * a.n_chunks + chunk) * (size_t) a.m * 2;
    f2 y0[kNS], y1[kNS];
#pragma unroll
    for (int s = 0; s < kNS; ++s) { y0[s] = a.zi[s][0] * last; y1[s] = a.zi[s][1] * last; }
    const int qm = a.q - 1, qs = __builtin_ctz(a.q);          // q is a power of two (sdrt_create)
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int e0 = blk * kB;
        f2 v[kB];
        if constexpr (CK) {
            const f2 *p = ck + (size_t) blk * (2 * kNS * 64);
#pragma unroll
            for (int s = 0; s < kNS; ++s) { z0[s] = p[s * 64]; z1[s] = p[(kNS + s) * 64]; }
            block(e0, v);
        } else {
#pragma unroll
            for (int u = 0; u < kB; ++u) v[u] = e0 + u < N ? ck[(size_t) (e0 + u) * 64] : zero;
        }
        if (e0 + kB <= N) {
            if constexpr (CK) {
#pragma unroll
                for (int u = 0; u < kB; ++u) v[u] = cascade(c, v[u], z0, z1);
            }
#pragma unroll
            for (int u = kB - 1; u >= 0; --u) v[u] = cascade(c, v[u], y0, y1);
        } else {
            if constexpr (CK) {
#pragma unroll
                for (int u = 0; u < kB; ++u)
                    if (e0 + u < N) v[u] = cascade(c, v[u], z0, z1);
            }
#pragma unroll
            for (int u = kB - 1; u >= 0; --u)
                if (e0 + u < N) v[u] = cascade(c, v[u], y0, y1);
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int k = e0 + u - kEdge;
            if (k >= 0 && k < n && (k & qm) == 0 && live) {
                double2 o;
                o.x = (double) v[u].x;
                o.y = (double) v[u].y;
                *(double2 *) (yo + 2 * (size_t) (k >> qs)) = o;
            }
        }
    }
}

}  // namespace

size_t f32c_work_bytes(int n, int rows, size_t n_chunks) {
    const size_t groups = (n_chunks * (size_t) rows + 63) / 64;
    return groups * (size_t) (n + 2 * kEdge) * 64 * sizeof(f2);
}

hipError_t launch_f32c(const F32cArgs &a, hipStream_t s) {
    const size_t items = a.n_chunks * (size_t) a.rows, groups = (items + 63) / 64;
    if (!groups) return hipSuccess;
    const size_t tiles = (size_t) (a.n + 2 * kEdge + 63) / 64;
    if (groups * tiles > 0x7fffffffull) return hipErrorInvalidValue;
    const int fused = sdrt_diag_int("f32c_fused", 2);          // 0: the two kernels with the sequences in memory, 1: checkpoints, 2: forward outputs stored
    if (a.shift_t && a.rows % 64 == 0 && fused != 0) {
        if (fused == 1) hipLaunchKernelGGL((f32c_fused_kernel<true, false>), dim3((unsigned) groups), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((f32c_fused_kernel<false, false>), dim3((unsigned) groups), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    if (fused != 0 && sdrt_diag_int("f32c_fused_any", 1) != 0) {          // every other row count, and no shift at all
        hipLaunchKernelGGL((f32c_fused_kernel<false, true>), dim3((unsigned) groups), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    if (a.shift && a.rows % 64 == 0) {
        const size_t blocks = (size_t) (a.rows / 64) * tiles * ((a.n_chunks + kCPB - 1) / kCPB);
        hipLaunchKernelGGL(f32c_prep_rows_kernel, dim3((unsigned) blocks), dim3(256), 0, s, a);
    } else
    hipLaunchKernelGGL(f32c_prep_kernel, dim3((unsigned) (groups * tiles)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(f32c_seq_kernel, dim3((unsigned) groups), dim3(64), 0, s, a);
    return hipGetLastError();
}
