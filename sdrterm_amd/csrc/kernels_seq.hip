// Small decimation factors (q = 2, 4; the CLI default is --decimation 2, src/sdrterm.py:80) and chunks with many outputs.
//
// chain_seq_kernel: with an output every 2 or 4 samples there is nothing to gain from block aggregates on the matrix cores:
// the zero-phase decimator (dsp_processor.py:147 -> scipy decimate) is evaluated in its per-sample modal form
//     F_i(t+1) = p_i F_i(t) + ac_i u[t]        state before sample t (causal),       F_i(0) from the head operator
//     G_i(t)   = p_i G_i(t+1) + aa_i u[t]      inclusive (anticausal),               G_i(n) from the tail operator
//     y[t]     = d0p u[t] + 2 Re sum_i (F_i(t) + G_i(t))                              at t = q k
// (tables.h: DecimTables; the same operator as every other chain kernel).  One workgroup per (chunk, row); every thread
// owns a run of n/256 consecutive samples: pass 1 = the run's aggregates from zero states, a scan over the 256 threads
// (wave shuffles + four wave totals), pass 2a = forward through the run with the true causal states (writes the direct
// and causal part of its outputs), pass 2b = backward with the anticausal states (adds the rest).  The IQ correction
// (read_file.py:64-77) runs forward in 2a and is unwound exactly in 2b: off_t = (off_{t+1} - lambda x_t) / (1 - lambda).
//
// post_big_kernel: the demodulation stage for chunks whose m outputs do not fit one wave's LDS share (m > 4096): one
// workgroup per item, FM resampler as the circulant product with the half-sample kernel (O(N^2), LDS-tiled), output
// lowpass as a modal scan over the workgroup.
#include <cstdlib>

#include "kernels_common.hpp"

namespace {

constexpr int SEQ_NCO_L = 64;        // the NCO phasor is tabulated as exp(i theta j), j < 64, times exp(i theta 64 B)

// suffix form of wave_scan_geometric: S_l = sum_{j >= l} mu^(j-l) x_j
__device__ __forceinline__ cd wave_scan_geometric_rev(cd x, cd mu, int lane) {
    cd mpow = mu;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = 1 << i;
        cd t = mk(__shfl_down(x.re, d), __shfl_down(x.im, d));
        if (lane + d > 63) t = mk(0.0, 0.0);
        x = cfma(mpow, t, x);
        mpow = cmul(mpow, mpow);
    }
    return x;
}

__device__ __forceinline__ cd cpow_u(cd b, unsigned e) {
    cd r = mk(1.0, 0.0);
    while (e) {
        if (e & 1u) r = cmul(r, b);
        b = cmul(b, b);
        e >>= 1;
    }
    return r;
}

}  // namespace

#ifndef SDRT_SEQ_WAVES
#define SDRT_SEQ_WAVES 1         // minimum waves per SIMD the register allocation of chain_seq_kernel is held to
#endif
template <int DT, bool NCO, bool IQ, int QS>
__global__ __launch_bounds__(256, SDRT_SEQ_WAVES) void chain_seq_kernel(DecimArgs a) {
    constexpr int qshift = QS;                             // decimation 2 or 4 (log2): which samples of a block are outputs is known at compile time
    constexpr int ITEM = Enc<DT>::size;
    constexpr int SPG = ITEM >= 8 ? 1 : 8 / ITEM;          // samples per 16-byte load (f64: one 16-byte sample)
    constexpr int GB = SPG * 2 * ITEM;                     // bytes per load
    __shared__ cd wtot[16][4];
    __shared__ cd tileL[SEQ_NCO_L];
    __shared__ cd edgeL[16];                               // [0..7] F0, [8..15] sample part of G_end; index mode*2 + stream
    // Outputs travel through LDS in blocks of 16 samples (round 4).  A thread's run is contiguous in time, so its outputs are
    // contiguous in memory and the 64 runs of a wave lie T / q x 16 bytes apart: written one by one, every store instruction
    // touched 64 cache lines with 16 bytes each, the second pass read them back the same way and wrote them again (counters of the
    // round-3 kernel at decimation 2: 2.88 GB written and 1.8 GB fetched per 2^26 samples against 0.54 + 0.27 algorithmic).  Now a
    // lane drops its outputs of a block into its row of the wave's tile, and the wave moves the tile to or from memory with whole
    // cache lines per run (decimation 2: eight outputs = 128 bytes per run and block).  Row stride 9 elements: the lanes' rows start on
    // different banks.
    constexpr int SEQ_BLK = 16, SEQ_ROW = SEQ_BLK / 2 + 1;
    __shared__ cd stageL[4][64 * SEQ_ROW];
    const DecimTables *__restrict__ tab = a.tab;
    const int n = a.n, T = n / 256;
    constexpr int qmask = (1 << qshift) - 1;
    const unsigned bid = blockIdx.x, xcd = bid & 7u, slot = bid >> 3;
    const long chunk = (long) (slot / (unsigned) a.rows) * 8 + xcd;
    const int row = (int) (slot % (unsigned) a.rows);
    if (chunk >= (long) a.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool swap = a.swap != 0;
    const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk * (size_t) n * 2 * ITEM;
    const double nbias = a.norm_bias, lam = a.iq_lambda, om = 1.0 - lam, rom = 1.0 / om;
    const cd *nco_blk = NCO ? (const cd *) a.nco_phase + (size_t) row * (size_t) (n / SEQ_NCO_L) : nullptr;
    if (NCO && tid < SEQ_NCO_L) tileL[tid] = ((const cd *) a.nco_tile)[(size_t) row * SEQ_NCO_L + tid];
    cd p[4], ac[4], aa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = mk(tab->pole[i][0], tab->pole[i][1]);
        ac[i] = mk(tab->ac[i][0], tab->ac[i][1]);
        aa[i] = mk(tab->aa[i][0], tab->aa[i][1]);
    }
    const double d0p = tab->d0p;
    const long t0 = (long) tid * T;
    // IQ offset at the run's first sample: group table (groups of 16 samples, from a zero offset at the chunk start) + the
    // chunk's true starting offset
    cd off0 = mk(0.0, 0.0);
    if constexpr (IQ) {
        const cd c0 = ((const cd *) a.iq_chunk_off)[chunk];
        const long grp = t0 / a.iq_gs;
        const cd go = ((const cd *) a.iq_group_off)[(size_t) chunk * (size_t) (n / a.iq_gs) + grp];
        const double pw = a.iq_pow[grp];
        off0 = mk(fma(pw, c0.re, go.re), fma(pw, c0.im, go.im));
    }
    __syncthreads();

    // sample t as the decimator sees it; forward: `off` is the offset at t and becomes the offset at t + 1
    auto shifted = [&](double xr, double xi, long t, double &ur, double &ui) {
        if constexpr (NCO) {
            const cd c = cmul(tileL[t & (SEQ_NCO_L - 1)], nco_blk[t / SEQ_NCO_L]);
            ur = xr * c.re - xi * c.im;
            ui = xr * c.im + xi * c.re;
        } else {
            ur = xr;
            ui = xi;
        }
    };
    // The run in blocks of 16 samples (round 4; before: sample by sample, with run-time tests for "is this an output", "is the block
    // complete" around every sample -- as many scalar and branch instructions as fused multiply-adds, and no scheduling across them).
    // begin(tb) / end(tb) run before / after a block (tb = its first sample), fn(jj, t, xr, xi) gets the decoded (biased)
    // components of sample t = tb + jj with jj a compile-time constant, in ascending or descending order.
    constexpr int BLK = 16, GPB = BLK / SPG;
    static_assert(BLK % SPG == 0, "whole load groups per block");
    cd nco_c = mk(1.0, 0.0);                               // exp(i theta 64 B) of the block (a block never straddles a multiple of 64)
    int nco_j0 = 0;
    auto block_phasor = [&](long tb) {
        if constexpr (NCO) { nco_c = nco_blk[tb / SEQ_NCO_L]; nco_j0 = (int) (tb & (SEQ_NCO_L - 1)); }
    };
    auto shifted_blk = [&](auto jc, double xr, double xi, double &ur, double &ui) {
        if constexpr (NCO) {
            const cd c = cmul(tileL[nco_j0 + decltype(jc)::value], nco_c);
            ur = xr * c.re - xi * c.im;
            ui = xr * c.im + xi * c.re;
        } else {
            ur = xr;
            ui = xi;
        }
    };
    auto for_blocks = [&](auto dir, auto begin, auto fn, auto end) {
        constexpr bool FWD = decltype(dir)::value;
        for (int b = 0; b < T / BLK; ++b) {
            const long tb = t0 + (long) (FWD ? b : T / BLK - 1 - b) * BLK;
            uint32_t wds[GPB][GB / 4];
#pragma unroll
            for (int g = 0; g < GPB; ++g) load_words<GB>(chunk_raw + (size_t) (tb + g * SPG) * 2 * ITEM, wds[g]);
            block_phasor(tb);
            begin(tb);
            static_for<0, BLK>([&](auto ic) {
                constexpr int jj = FWD ? decltype(ic)::value : BLK - 1 - decltype(ic)::value;
                constexpr int g = jj / SPG, j = jj % SPG;
                fn(std::integral_constant<int, jj>{}, tb + jj, decode<DT>(get_item<DT, 2 * j>(wds[g]), swap) + nbias,
                   decode<DT>(get_item<DT, 2 * j + 1>(wds[g]), swap));
            });
            end(tb);
        }
    };
    auto nothing = [](long) {};

    // ---- edge operators: F(0) from samples 0..27 (thread 0), the sample part of G(n) from samples n-28..n-1 (thread 255) ----
    // Waves 0 and 3 as a whole (every lane computes thread 0's / thread 255's sums from the same operands, lane 0 / 63 stores them), and
    // before pass 1.  Round 4, while the passes became compile-time blocks: in two of the 64 instantiations thread 255's anticausal
    // aggregate came back overwritten (float32 at decimation 4 and 2, int8 at decimation 2; run 254's outputs wrong, all else exact),
    // moving with the place of this block and of the suffix scan below -- at 450 - 490 registers with values parked in accumulator
    // registers across lane-divergent regions.  Not understood down to the instruction; what is done about it: no two-lane branch, the
    // aggregates' live range ends right after pass 1, and tests/test_gpu_parity.py runs EVERY instantiation against the oracle
    // (tools/seq_instantiations.py: the same at two sizes).
    if (wave == 0 || wave == 3) {
        const bool head = wave == 0;
        cd acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = mk(0.0, 0.0);
        const long base = head ? 0 : (long) n - (SDRT_EDGE + 1);
        const int src = head ? 0 : 63;                     // the lane whose run holds the window
        cd off = mk(__shfl(off0.re, src), __shfl(off0.im, src));
        if constexpr (IQ) {
            if (!head) {                    // walk from the last run's start to the window (the run is at least 32 samples long)
                for (long t = (long) 255 * T; t < base; ++t) {
                    const typename Enc<DT>::raw *rp = (const typename Enc<DT>::raw *) chunk_raw + 2 * t;
                    const double zr = decode<DT>(rp[0], swap) + nbias - off.re, zi = decode<DT>(rp[1], swap) - off.im;
                    off.re = fma(zr, lam, off.re); off.im = fma(zi, lam, off.im);
                }
            }
        }
        for (int l = 0; l <= SDRT_EDGE; ++l) {
            const long t = base + l;
            const typename Enc<DT>::raw *rp = (const typename Enc<DT>::raw *) chunk_raw + 2 * t;
            double xr = decode<DT>(rp[0], swap) + nbias, xi = decode<DT>(rp[1], swap);
            if constexpr (IQ) {
                xr -= off.re; xi -= off.im;
                off.re = fma(xr, lam, off.re); off.im = fma(xi, lam, off.im);
            }
            double ur, ui;
            shifted(xr, xi, t, ur, ui);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double *c = head ? tab->head[i][l] : tab->tailU[i][l];
                acc[2 * i] = mk(fma(c[0], ur, acc[2 * i].re), fma(c[1], ur, acc[2 * i].im));
                acc[2 * i + 1] = mk(fma(c[0], ui, acc[2 * i + 1].re), fma(c[1], ui, acc[2 * i + 1].im));
            }
        }
        if (lane == src) {
#pragma unroll
            for (int o = 0; o < 8; ++o) edgeL[(head ? 0 : 8) + o] = acc[o];
        }
    }
    // ---- pass 1: end states of the run from zero (causal: Horner; anticausal aggregate: running power) ------------------
    cd F[4][2], A[4][2], pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { F[i][0] = F[i][1] = A[i][0] = A[i][1] = mk(0.0, 0.0); pw[i] = aa[i]; }
    {
        cd off = off0;
        for_blocks(std::true_type{}, nothing, [&](auto jc, long, double xr, double xi) {
            if constexpr (IQ) {
                xr -= off.re; xi -= off.im;
                off.re = fma(xr, lam, off.re); off.im = fma(xi, lam, off.im);
            }
            double ur, ui;
            shifted_blk(jc, xr, xi, ur, ui);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                F[i][0] = mk(fma(p[i].re, F[i][0].re, fma(-p[i].im, F[i][0].im, ac[i].re * ur)), fma(p[i].re, F[i][0].im, fma(p[i].im, F[i][0].re, ac[i].im * ur)));
                F[i][1] = mk(fma(p[i].re, F[i][1].re, fma(-p[i].im, F[i][1].im, ac[i].re * ui)), fma(p[i].re, F[i][1].im, fma(p[i].im, F[i][1].re, ac[i].im * ui)));
                A[i][0] = mk(fma(pw[i].re, ur, A[i][0].re), fma(pw[i].im, ur, A[i][0].im));
                A[i][1] = mk(fma(pw[i].re, ui, A[i][1].re), fma(pw[i].im, ui, A[i][1].im));
                pw[i] = cmul(pw[i], p[i]);
            }
        }, nothing);
    }
    // ---- causal scan over the threads: state at the run start ----------------------------------------------------------
    cd MT[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) MT[i] = cpow_u(p[i], (unsigned) T);
    // the suffix scan of the anticausal aggregates first (their registers are then free; the wave totals wait in wtot[8..15])
    cd Sr[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            Sr[i][st] = wave_scan_geometric_rev(A[i][st], MT[i], lane);
            if (lane == 0) wtot[8 + i * 2 + st][wave] = Sr[i][st];
        }
    cd S[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            S[i][st] = wave_scan_geometric(F[i][st], MT[i], lane);
            if (lane == 63) wtot[i * 2 + st][wave] = S[i][st];
        }
    __syncthreads();
    cd Fs[4][2], Fend[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const cd M64 = cpow_u(MT[i], 64u), Ml = cpow_u(MT[i], (unsigned) lane), Mt = cpow_u(MT[i], (unsigned) tid);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            cd wp = mk(0.0, 0.0);                          // inclusive sum of the waves before this one
            for (int wv = 0; wv < wave; ++wv) wp = cfma(M64, wp, wtot[i * 2 + st][wv]);
            cd ex = mk(__shfl_up(S[i][st].re, 1), __shfl_up(S[i][st].im, 1));
            if (lane == 0) ex = mk(0.0, 0.0);
            // exclusive value: previous lanes of this wave + earlier waves carried over `lane` runs + F(0) over `tid` runs
            Fs[i][st] = cadd(ex, cadd(cmul(Ml, wp), cmul(Mt, edgeL[i * 2 + st])));
            cd tot = mk(0.0, 0.0);
            for (int wv = 0; wv < 4; ++wv) tot = cfma(M64, tot, wtot[i * 2 + st][wv]);
            Fend[i][st] = cadd(tot, cmul(cpow_u(MT[i], 256u), edgeL[i * 2 + st]));      // F(n)
        }
    }
    __syncthreads();
    // ---- anticausal: G(n) from F(n) and the tail samples, then the suffix scan: state entering the run from the right ------
    cd Gs[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const cd M64 = cpow_u(MT[i], 64u), Ml = cpow_u(MT[i], (unsigned) (63 - lane)), Mt = cpow_u(MT[i], (unsigned) (255 - tid));
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            cd Gn = edgeL[8 + i * 2 + st];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const cd Fj = Fend[j][st];
                Gn = cfma(mk(tab->tailF[i][j][0], tab->tailF[i][j][1]), Fj, Gn);
                Gn = cfma(mk(tab->tailFc[i][j][0], tab->tailFc[i][j][1]), mk(Fj.re, -Fj.im), Gn);
            }
            cd wp = mk(0.0, 0.0);                          // suffix sum of the waves after this one
            for (int wv = 3; wv > wave; --wv) wp = cfma(M64, wp, wtot[8 + i * 2 + st][wv]);
            cd ex = mk(__shfl_down(Sr[i][st].re, 1), __shfl_down(Sr[i][st].im, 1));
            if (lane == 63) ex = mk(0.0, 0.0);
            Gs[i][st] = cadd(ex, cadd(cmul(Ml, wp), cmul(Mt, Gn)));                   // G(t0 + T)
        }
    }

    // ---- pass 2a: forward with the causal states: direct + causal part of the run's outputs -----------------------------
    double *yd = a.ydec + ((size_t) row * a.n_chunks + (size_t) chunk) * (size_t) a.m * 2;
    cd *myrow = &stageL[wave][lane * SEQ_ROW];
    constexpr int obq = SEQ_BLK >> qshift;                  // outputs per block and run
    const long kstride = (long) T >> qshift;                // outputs per run
    // the wave's tile <-> memory: element e = i * 64 + lane of the tile = output e % obq of run e / obq (this block): consecutive
    // lanes take consecutive outputs of a run, whole 128-byte (decimation 4: 64-byte) pieces per run
    auto tile_to_memory = [&](long trel) {                  // trel: first sample of the block relative to the run start
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const long k0 = trel >> qshift;
        for (int i = 0; i < obq; ++i) {
            const int e = i * 64 + lane, r = e / obq, j = e % obq;
            const cd v = stageL[wave][r * SEQ_ROW + j];
            *(double2 *) (yd + 2 * (size_t) ((long) (wave * 64 + r) * kstride + k0 + j)) = make_double2(v.re, v.im);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto memory_to_tile = [&](long trel) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const long k0 = trel >> qshift;
        for (int i = 0; i < obq; ++i) {
            const int e = i * 64 + lane, r = e / obq, j = e % obq;
            const double2 v = *(const double2 *) (yd + 2 * (size_t) ((long) (wave * 64 + r) * kstride + k0 + j));
            stageL[wave][r * SEQ_ROW + j] = mk(v.x, v.y);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    cd off_end = off0;
    {
        cd off = off0;
        for_blocks(std::true_type{}, nothing, [&](auto jc, long, double xr, double xi) {
            constexpr int jj = decltype(jc)::value;
            if constexpr (IQ) {
                xr -= off.re; xi -= off.im;
                off.re = fma(xr, lam, off.re); off.im = fma(xi, lam, off.im);
            }
            double ur, ui;
            shifted_blk(jc, xr, xi, ur, ui);
            if constexpr ((jj & qmask) == 0) {
                double yr = d0p * ur, yi = d0p * ui;
#pragma unroll
                for (int i = 0; i < 4; ++i) { yr = fma(2.0, Fs[i][0].re, yr); yi = fma(2.0, Fs[i][1].re, yi); }
                myrow[jj >> qshift] = mk(yr, yi);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Fs[i][0] = mk(fma(p[i].re, Fs[i][0].re, fma(-p[i].im, Fs[i][0].im, ac[i].re * ur)), fma(p[i].re, Fs[i][0].im, fma(p[i].im, Fs[i][0].re, ac[i].im * ur)));
                Fs[i][1] = mk(fma(p[i].re, Fs[i][1].re, fma(-p[i].im, Fs[i][1].im, ac[i].re * ui)), fma(p[i].re, Fs[i][1].im, fma(p[i].im, Fs[i][1].re, ac[i].im * ui)));
            }
        }, [&](long tb) { tile_to_memory(tb - t0); });       // the block is complete (every lane at once)
        off_end = off;
    }
    // ---- pass 2b: backward with the anticausal states (inclusive), the IQ offset unwound ---------------------------------
    {
        cd off = off_end;                                   // offset at t0 + T
        for_blocks(std::false_type{}, [&](long tb) { memory_to_tile(tb - t0); }, [&](auto jc, long, double xr, double xi) {
            constexpr int jj = decltype(jc)::value;              // (descending: a block begins with its last sample)
            if constexpr (IQ) {
                off.re = (off.re - lam * xr) * rom;          // offset at t
                off.im = (off.im - lam * xi) * rom;
                xr -= off.re; xi -= off.im;
            }
            double ur, ui;
            shifted_blk(jc, xr, xi, ur, ui);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Gs[i][0] = mk(fma(p[i].re, Gs[i][0].re, fma(-p[i].im, Gs[i][0].im, aa[i].re * ur)), fma(p[i].re, Gs[i][0].im, fma(p[i].im, Gs[i][0].re, aa[i].im * ur)));
                Gs[i][1] = mk(fma(p[i].re, Gs[i][1].re, fma(-p[i].im, Gs[i][1].im, aa[i].re * ui)), fma(p[i].re, Gs[i][1].im, fma(p[i].im, Gs[i][1].re, aa[i].im * ui)));
            }
            if constexpr ((jj & qmask) == 0) {
                cd v = myrow[jj >> qshift];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v.re = fma(2.0, Gs[i][0].re, v.re); v.im = fma(2.0, Gs[i][1].re, v.im); }
                myrow[jj >> qshift] = v;
            }
        }, [&](long tb) { tile_to_memory(tb - t0); });       // the block's first sample was its last one to do
    }
}

template <int DT, int QS>
static hipError_t launch_seq_q(const DecimArgs &a, bool nco, bool iq, dim3 grid, hipStream_t s) {
    if (nco) {
        if (iq) hipLaunchKernelGGL((chain_seq_kernel<DT, true, true, QS>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((chain_seq_kernel<DT, true, false, QS>), grid, dim3(256), 0, s, a);
    } else {
        if (iq) hipLaunchKernelGGL((chain_seq_kernel<DT, false, true, QS>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((chain_seq_kernel<DT, false, false, QS>), grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
// The 64 instantiations take minutes to compile: decimation 4 lives in a translation unit of its own (kernels_seq_b.hip includes this
// file with SDRT_SEQ_PART_B and compiles nothing but them).
template <int QS>
static hipError_t launch_seq_every(const DecimArgs &a, int dt, bool nco, bool iq, dim3 grid, hipStream_t s) {
    switch (dt) {
        case SDRT_S8: return launch_seq_q<SDRT_S8, QS>(a, nco, iq, grid, s);
        case SDRT_U8: return launch_seq_q<SDRT_U8, QS>(a, nco, iq, grid, s);
        case SDRT_S16: return launch_seq_q<SDRT_S16, QS>(a, nco, iq, grid, s);
        case SDRT_U16: return launch_seq_q<SDRT_U16, QS>(a, nco, iq, grid, s);
        case SDRT_S32: return launch_seq_q<SDRT_S32, QS>(a, nco, iq, grid, s);
        case SDRT_U32: return launch_seq_q<SDRT_U32, QS>(a, nco, iq, grid, s);
        case SDRT_F32: return launch_seq_q<SDRT_F32, QS>(a, nco, iq, grid, s);
        case SDRT_F64: return launch_seq_q<SDRT_F64, QS>(a, nco, iq, grid, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_chain_seq_q4(const DecimArgs &a, int dt, bool nco, bool iq, dim3 grid, hipStream_t s);

#ifdef SDRT_SEQ_PART_B
hipError_t launch_chain_seq_q4(const DecimArgs &a, int dt, bool nco, bool iq, dim3 grid, hipStream_t s) {
    return launch_seq_every<2>(a, dt, nco, iq, grid, s);
}
#else
bool chain_seq_supported(int q, int n) {
    // every thread's run: whole load groups, whole IQ groups (16 samples), at least the 28-sample edge windows, whole outputs
    const int T = n / 256;
    return n % 256 == 0 && T >= 32 && T % 16 == 0 && (q == 2 || q == 4) && n % SEQ_NCO_L == 0;
}

hipError_t launch_chain_seq(const DecimArgs &a, int q, int dt, bool nco, bool iq, size_t n_chunks, hipStream_t s) {
    const size_t groups = (n_chunks + 7) / 8;
    const dim3 grid((unsigned) (groups * 8 * (size_t) a.rows));
    if (q == 2) return launch_seq_every<1>(a, dt, nco, iq, grid, s);
    if (q == 4) return launch_chain_seq_q4(a, dt, nco, iq, grid, s);
    return hipErrorInvalidValue;                           // (sdrt_create sends decimation 8 and up to the block kernels)
}

// output lowpass of an item's m samples in place (dsp_processor.py:32-36,149), 256 threads: a modal scan over the workgroup
// (PAD: the samples sit in LDS, one pad element per 64 -- a thread's run of m/256 consecutive samples then starts two banks after
// its neighbour's instead of on the same one)
template <bool PAD = false, int THR = 256, int NW = 4>
__device__ __forceinline__ void workgroup_lowpass(const PostArgs &a, double *out, cd (&ptot)[SDRT_OUT_MAXMODES][NW]) {
    const int m = a.m, tid = threadIdx.x % THR, lane = tid & 63, wave = tid >> 6;
    auto at = [&](int i) -> double & { return out[PAD ? i + (i >> 6) : i]; };
    // ---- output lowpass: y[t] = D x[t] + sum_k Re(r_k s_k(t)), s_k(t+1) = p_k s_k(t) + x[t], s_k(0) = 0 --------------------
    const OutTables *__restrict__ ot = a.otab;
    const int nm = ot->nmodes;
    if (nm < 0) {
        // sequential transposed-direct-form-II cascade (any SOS); only when the modal form is not available
        if (tid == 0) {
            double st[2 * SDRT_MAX_OUT_SECTIONS];
            for (int s = 0; s < 2 * SDRT_MAX_OUT_SECTIONS; ++s) st[s] = 0.0;
            for (int i = 0; i < m; ++i) {
                double xc = at(i);
                for (int s = 0; s < a.out_sections; ++s) {
                    const double *c = a.out_sos + 6 * s;
                    const double xn = fma(c[0], xc, st[2 * s]);
                    st[2 * s] = fma(c[1], xc, fma(-c[4], xn, st[2 * s + 1]));
                    st[2 * s + 1] = fma(c[2], xc, -c[5] * xn);
                    xc = xn;
                }
                at(i) = xc;
            }
        }
        return;
    }
    const int E = m / THR, t0 = tid * E;
    cd pk[SDRT_OUT_MAXMODES], rk[SDRT_OUT_MAXMODES], sst[SDRT_OUT_MAXMODES];
#pragma unroll
    for (int k = 0; k < SDRT_OUT_MAXMODES; ++k) {
        pk[k] = k < nm ? mk(ot->p[k][0], ot->p[k][1]) : mk(0.0, 0.0);
        rk[k] = k < nm ? mk(ot->r[k][0], ot->r[k][1]) : mk(0.0, 0.0);
        cd s = mk(0.0, 0.0);
        if (k < nm)
            for (int e = 0; e < E; ++e) s = cfma(pk[k], s, mk(at(t0 + e), 0.0));
        const cd PE = cpow_u(pk[k], (unsigned) E);
        const cd sc = wave_scan_geometric(s, PE, lane);
        if (lane == 63) ptot[k][wave] = sc;
        __syncthreads();
        cd wp = mk(0.0, 0.0);
        const cd P64 = cpow_u(PE, 64u);
        for (int wv = 0; wv < wave; ++wv) wp = cfma(P64, wp, ptot[k][wv]);
        cd ex = mk(__shfl_up(sc.re, 1), __shfl_up(sc.im, 1));
        if (lane == 0) ex = mk(0.0, 0.0);
        sst[k] = cadd(ex, cmul(cpow_u(PE, (unsigned) lane), wp));     // state at the run's first sample
    }
    const double Dd = ot->D;
    for (int e = 0; e < E; ++e) {
        const double x = at(t0 + e);
        double yv = Dd * x;
#pragma unroll
        for (int k = 0; k < SDRT_OUT_MAXMODES; ++k) {
            if (k < nm) {
                yv += rk[k].re * sst[k].re - rk[k].im * sst[k].im;
                sst[k] = cfma(pk[k], sst[k], mk(x, 0.0));
            }
        }
        at(t0 + e) = yv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// post_big_kernel: FM pair angle (demodulation.py:25-32) + resample m/2 -> m (demodulation.py:35-38: even outputs are the
// angles, odd outputs the circulant product with the half-sample kernel) or AM (demodulation.py:41-48), then the output
// lowpass (dsp_processor.py:32-36,149) as a modal scan.  One workgroup of 256 threads per (chunk, row) item; the item's m
// samples live in the output buffer itself between the steps.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void post_big_kernel(PostArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cd ptot[SDRT_OUT_MAXMODES][4];
    const int m = a.m, tid = threadIdx.x;
    const size_t item = post_item(a, blockIdx.x);
    const cd *y = (const cd *) a.y + item * (size_t) m;
    double *out = a.out + item * (size_t) m;
    if (a.mode == SDRT_FM) {
        const int N = m >> 1, msk = N - 1;
        constexpr int AT = 2048, R = 16;                    // angles per LDS tile; outputs per thread and turn
        double *hr = (double *) smem;                       // [N]: hr[i] = h[(-i) mod N]
        double *at = hr + N;                                // [AT]
        for (int i = tid; i < N; i += 256) hr[i] = a.resample_odd[i == 0 ? 0 : N - i];
        for (int j = tid; j < N; j += 256) {
            const cd pa = y[2 * j], c = y[2 * j + 1];
            out[2 * j] = atan2_pi(pa.im * c.re - pa.re * c.im, pa.re * c.re + pa.im * c.im);    // angle(d[i] * conj(d[i+1]))
        }
        __syncthreads();
        // z_odd[j] = sum_k a[k] hr[(k - j) mod N]: thread owns R consecutive outputs per turn; a sliding register window
        // feeds them from one new value of hr per k (as resample_wave does for one wave)
        for (int j0 = R * tid; j0 < N; j0 += R * 256) {
            double acc[R], B[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { acc[r] = 0.0; B[r] = 0.0; }
            for (int k0 = 0; k0 < N; k0 += AT) {
                __syncthreads();
                for (int i = tid; i < AT; i += 256) at[i] = out[2 * (k0 + i)];
                __syncthreads();
#pragma unroll
                for (int r = 1; r < R; ++r) B[r] = hr[(k0 - j0 - r) & msk];
                for (int k = 0; k < AT; k += R) {
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        B[(R - u) % R] = hr[(k0 + k + u - j0) & msk];
                        const double ak = at[k + u];
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = fma(ak, B[(r - u + R) % R], acc[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) out[2 * (j0 + r) + 1] = acc[r];
        }
    } else if (a.am_in_lds) {
        // AM (demodulation.py:41-48), round 4: the item's samples sit in LDS for the lowpass (one pad element per 64) instead of going
        // through the output buffer as 512-byte-strided thread runs; one coalesced store of the result
        double *zb = (double *) smem;
        for (int k = tid; k < m; k += 256) {
            const cd v = y[k];
            zb[k + (k >> 6)] = v.re * v.re + v.im * v.im;
        }
        __syncthreads();
        workgroup_lowpass<true>(a, zb, ptot);
        __syncthreads();
        for (int k = tid; k < m; k += 256) out[k] = zb[k + (k >> 6)];
        return;
    } else {
        for (int k = tid; k < m; k += 256) {
            const cd v = y[k];
            out[k] = v.re * v.re + v.im * v.im;
        }
    }
    __syncthreads();
    workgroup_lowpass(a, out, ptot);
}

// ---------------------------------------------------------------------------------------------------------------------
// post_bigfft_kernel: the same stage with scipy.signal.resample as the reference computes it (rfft, zero-pad, irfft;
// demodulation.py:35-38) for the sizes the decimation factors 2 and 4 -- the CLI's default among them -- produce: N = m/2 = 4096 or
// 8192 pair angles per chunk.  One workgroup of 256 threads per item; Stockham radix-4 passes (+ one radix-2 pass for 8192 = 2 x 4^6)
// through one LDS image of the sequence, in place: every pass loads all of its points into registers, the workgroup meets, then
// stores.  O(N log N) where post_big_kernel's circulant product is O(N^2): 4.5 -> 0.3 ms per 1024 chunks at decimation 2.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int xq(int i) { return i + (i >> 4); }       // one pad element per sixteen (8192 points: 136 KB)
__device__ __forceinline__ cd csub_(cd a, cd b) { return mk(a.re - b.re, a.im - b.im); }
template <bool INV> __device__ __forceinline__ cd cmul_tw_(cd v, cd w) {     // v * w (forward) or v * conj(w) (inverse)
    if constexpr (INV) return mk(fma(v.re, w.re, v.im * w.im), fma(v.im, w.re, -(v.re * w.im)));
    else return mk(fma(v.re, w.re, -(v.im * w.im)), fma(v.re, w.im, v.im * w.re));
}
// one radix-4 Stockham pass over N = 1024 B points held as x[b][k] = X[t + k N/4], t = tid + 256 b; sub-transform length N / S_
template <int B, int S_, bool INV, class Store>
__device__ __forceinline__ void radix4_pass_wg(cd (&x)[B][4], const cd *__restrict__ tw, const int tid, Store store) {
    constexpr int N = 1024 * B, n = N / S_;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int t = tid + 256 * b, p = t / S_, q = t % S_;
        const cd a0 = x[b][0], a1 = x[b][1], a2 = x[b][2], a3 = x[b][3];
        const cd t0 = cadd(a0, a2), t1 = csub_(a0, a2), t2 = cadd(a1, a3), t3 = csub_(a1, a3);
        const cd jt3 = INV ? mk(-t3.im, t3.re) : mk(t3.im, -t3.re);
        cd y0 = cadd(t0, t2), y1 = cadd(t1, jt3), y2 = csub_(t0, t2), y3 = csub_(t1, jt3);
        if constexpr (n > 4) {
            y1 = cmul_tw_<INV>(y1, tw[p * S_]);
            y2 = cmul_tw_<INV>(y2, tw[2 * p * S_]);
            y3 = cmul_tw_<INV>(y3, tw[3 * p * S_]);
        }
        const int o = q + S_ * 4 * p;
        store(o, y0);
        store(o + S_, y1);
        store(o + 2 * S_, y2);
        store(o + 3 * S_, y3);
    }
}
// a whole transform: x holds the first pass's input; the result goes to store_last(index, value) in natural order.
// N = 1024 B = 4^5 (B = 1), 2 x 4^5 (2), 4^6 (4), 2 x 4^6 (8): five or six radix-4 passes, then a radix-2 pass where a factor 2 is left
template <int B, bool INV, class Store>
__device__ __forceinline__ void fft_workgroup(cd (&x)[B][4], cd *X, const cd *__restrict__ tw, const int tid, Store store_last) {
    constexpr int N = 1024 * B;
    constexpr bool R2 = B == 2 || B == 8, SIX = B >= 4;
    auto stX = [&](int o, cd v) { X[xq(o)] = v; };
    auto reload = [&]() {
        __syncthreads();                             // the pass's stores are there
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) x[b][k] = X[xq(tid + 256 * b + k * (N / 4))];
        __syncthreads();                             // everybody holds its points: the image may be overwritten
    };
    radix4_pass_wg<B, 1, INV>(x, tw, tid, stX);
    reload();
    radix4_pass_wg<B, 4, INV>(x, tw, tid, stX);
    reload();
    radix4_pass_wg<B, 16, INV>(x, tw, tid, stX);
    reload();
    radix4_pass_wg<B, 64, INV>(x, tw, tid, stX);
    reload();
    if constexpr (SIX) {
        radix4_pass_wg<B, 256, INV>(x, tw, tid, stX);
        reload();
    }
    constexpr int SL = SIX ? 1024 : 256;             // the last radix-4 pass
    if constexpr (!R2) {
        radix4_pass_wg<B, SL, INV>(x, tw, tid, store_last);
    } else {
        radix4_pass_wg<B, SL, INV>(x, tw, tid, stX);
        __syncthreads();
        // n = 2: y[q] = a + b, y[q + N/2] = a - b, q < N/2.  Everybody holds its points before the first result is stored:
        // store_last may write over the image
        cd u[N / 512], v[N / 512];
#pragma unroll
        for (int c = 0; c < N / 512; ++c) { u[c] = X[xq(tid + 256 * c)]; v[c] = X[xq(tid + 256 * c + N / 2)]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < N / 512; ++c) {
            const int q = tid + 256 * c;
            store_last(q, cadd(u[c], v[c]));
            store_last(q + N / 2, csub_(u[c], v[c]));
        }
    }
}
// The same transform for THR threads per item and N = 4 THR B points = 4^K or 2 x 4^K: K radix-4 passes, then a radix-2 pass where a
// factor 2 is left -- pass for pass what fft_workgroup does with 256 threads (the butterflies are the same, only their owners differ).
template <int THR, int B, int S_, bool INV, class Store>
__device__ __forceinline__ void radix4_pass_grp(cd (&x)[B][4], const cd *__restrict__ tw, const int tid, Store store) {
    constexpr int N = 4 * THR * B, n = N / S_;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int t = tid + THR * b, p = t / S_, q = t % S_;
        const cd a0 = x[b][0], a1 = x[b][1], a2 = x[b][2], a3 = x[b][3];
        const cd t0 = cadd(a0, a2), t1 = csub_(a0, a2), t2 = cadd(a1, a3), t3 = csub_(a1, a3);
        const cd jt3 = INV ? mk(-t3.im, t3.re) : mk(t3.im, -t3.re);
        cd y0 = cadd(t0, t2), y1 = cadd(t1, jt3), y2 = csub_(t0, t2), y3 = csub_(t1, jt3);
        if constexpr (n > 4) {
            y1 = cmul_tw_<INV>(y1, tw[p * S_]);
            y2 = cmul_tw_<INV>(y2, tw[2 * p * S_]);
            y3 = cmul_tw_<INV>(y3, tw[3 * p * S_]);
        }
        const int o = q + S_ * 4 * p;
        store(o, y0);
        store(o + S_, y1);
        store(o + 2 * S_, y2);
        store(o + 3 * S_, y3);
    }
}
__host__ __device__ constexpr int ilog2_(int v) { return v <= 1 ? 0 : 1 + ilog2_(v / 2); }
__host__ __device__ constexpr int ipow4_(int e) { return e <= 0 ? 1 : 4 * ipow4_(e - 1); }
template <int THR, int B, bool INV, class Store>
__device__ __forceinline__ void fft_group(cd (&x)[B][4], cd *X, const cd *__restrict__ tw, const int tid, Store store_last) {
    constexpr int N = 4 * THR * B, LG = ilog2_(N), K = LG / 2;
    constexpr bool R2 = (LG & 1) != 0;
    auto stX = [&](int o, cd v) { X[xq(o)] = v; };
    static_for<0, K>([&](auto ic) {
        constexpr int i = decltype(ic)::value, S_ = ipow4_(i);
        if constexpr (i + 1 < K || R2) {
            radix4_pass_grp<THR, B, S_, INV>(x, tw, tid, stX);
            __syncthreads();                         // the pass's stores are there
            if constexpr (i + 1 < K) {
#pragma unroll
                for (int b = 0; b < B; ++b)
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[b][k] = X[xq(tid + THR * b + k * (N / 4))];
                __syncthreads();                     // everybody holds its points: the image may be overwritten
            }
        } else {
            radix4_pass_grp<THR, B, S_, INV>(x, tw, tid, store_last);
        }
    });
    if constexpr (R2) {
        // everybody holds its points before the first result is stored: store_last may write over the image
        constexpr int C = N / (2 * THR);
        cd u[C], v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { u[c] = X[xq(tid + THR * c)]; v[c] = X[xq(tid + THR * c + N / 2)]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C; ++c) {
            store_last(tid + THR * c, cadd(u[c], v[c]));
            store_last(tid + THR * c + N / 2, csub_(u[c], v[c]));
        }
    }
}
}  // namespace

template <int B, bool PAIRS>
__global__ __launch_bounds__(256) void post_bigfft_kernel(PostArgs a) {
    constexpr int N = 1024 * B, M = 2 * N;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cd ptot[SDRT_OUT_MAXMODES][4];
    cd *X = (cd *) smem;                                        // [N + N/16]
    const cd *__restrict__ tw = (const cd *) a.fft_tw;          // [N] exp(-2 pi i k/N), from L2
    const cd *__restrict__ sh = (const cd *) a.fft_shift;       // [N] half-sample shift / N
    const int tid = threadIdx.x;
    const size_t item = post_item(a, blockIdx.x);
    const cd *y = (const cd *) a.y + item * (size_t) ((PAIRS && !a.pairs_wide) ? N : M);
    double *out = a.out + item * (size_t) M;
    cd x[B][4];
    // pair angles (demodulation.py:25-32): the even output samples, and the real input of the transform
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = tid + 256 * b + k * (N / 4);
            double an;
            if constexpr (PAIRS) {
                const cd z = y[j];                              // d[2j] conj(d[2j+1]) from the chain kernel
                an = atan2_pi(z.im, z.re);
            } else {
                const cd pa = y[2 * j], c = y[2 * j + 1];
                an = atan2_pi(pa.im * c.re - pa.re * c.im, pa.re * c.re + pa.im * c.im);
            }
            out[2 * j] = an;
            x[b][k] = mk(an, 0.0);
        }
    fft_workgroup<B, false>(x, X, tw, tid, [&](int o, cd v) { X[xq(o)] = v; });
    __syncthreads();
    // spectrum x half-sample shift, inverse transform: the odd output samples (imaginary parts are rounding noise)
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = tid + 256 * b + k * (N / 4);
            x[b][k] = cmul(X[xq(f)], sh[f]);
        }
    __syncthreads();
    fft_workgroup<B, true>(x, X, tw, tid, [&](int o, cd v) { out[2 * o + 1] = v.re; });
    __syncthreads();
    workgroup_lowpass(a, out, ptot);
}

// ... and with the pair angles packed two to a complex number (round 4, as post_fftp_kernel in kernels_post.hip): ONE transform of
// H = N/2 = 1024 B points each way, half the passes' work and half the LDS image (two to four workgroups per CU instead of one or two).
// (THR threads per item: 256, or 512 at 16384 outputs per chunk -- the samples take 133 KB of LDS there, one workgroup per CU: with 256
// threads that was one wave per SIMD)
template <int THR, int B, bool PAIRS>
__global__ __launch_bounds__(THR) void post_bigfftp_kernel(PostArgs a) {
    constexpr int H = 4 * THR * B, N = 2 * H, M = 2 * N;
    // 32768 outputs per chunk (decimation 2 of 8-bit records, round 5): the item's samples do not fit the LDS beside nothing (266 KB);
    // they go through the output buffer for the lowpass, as in post_bigfft_kernel, and the even samples are written at once
    constexpr bool ZL = (size_t) (M + M / 64) * 8 <= 160 * 1024;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ cd ptot[SDRT_OUT_MAXMODES][THR / 64];
    cd *X = (cd *) smem;                                        // [H + H/16]
    const cd *__restrict__ tw = (const cd *) a.fftp_tw;         // [H] exp(-2 pi i k/H), from L2
    const double4 *__restrict__ ab = (const double4 *) a.fftp_ab;      // [H] alpha, beta (sdrt_build_fft_resample_packed)
    const int tid = threadIdx.x;
    const size_t item = post_item(a, blockIdx.x);
    const cd *y = (const cd *) a.y + item * (size_t) ((PAIRS && !a.pairs_wide) ? N : M);
    double *out = a.out + item * (size_t) M;
    cd x[B][4], ev[B][4];
    auto angle_of = [&](int j) {
        if constexpr (PAIRS) {
            const cd z = y[j];                                  // d[2j] conj(d[2j+1]) from the chain kernel
            return atan2_pi(z.im, z.re);
        } else {
            const cd pa = y[2 * j], c = y[2 * j + 1];
            return atan2_pi(pa.im * c.re - pa.re * c.im, pa.re * c.re + pa.im * c.im);
        }
    };
    // pair angles (demodulation.py:25-32): the even output samples, and -- two to a point -- the input of the transform
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = tid + THR * b + k * (H / 4);
            const double a0 = angle_of(2 * n), a1 = angle_of(2 * n + 1);
            x[b][k] = mk(a0, a1);
            if constexpr (ZL) ev[b][k] = x[b][k];               // (64 registers through the transforms: no round trip through memory)
            else { out[4 * n] = a0; out[4 * n + 2] = a1; }
        }
    fft_group<THR, B, false>(x, X, tw, tid, [&](int o, cd v) { X[xq(o)] = v; });
    __syncthreads();
    // packed spectrum of the half-sample shifted sequence, bin by bin: alpha Z[f] + beta conj(Z[H - f])
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = tid + THR * b + k * (H / 4);
            const cd Z = X[xq(f)], Zm = X[xq((H - f) & (H - 1))];
            const double4 c = ab[f];
            x[b][k] = mk(fma(c.x, Z.re, fma(-c.y, Z.im, fma(c.z, Zm.re, c.w * Zm.im))),
                         fma(c.x, Z.im, fma(c.y, Z.re, fma(c.w, Zm.re, -(c.z * Zm.im)))));
        }
    __syncthreads();
    if constexpr (!ZL) {
        fft_group<THR, B, true>(x, X, tw, tid, [&](int o, cd v) { out[4 * o + 1] = v.re; out[4 * o + 3] = v.im; });
        __syncthreads();
        workgroup_lowpass<false, THR, THR / 64>(a, out, ptot);
        return;
    }
    // The item's M output samples for the lowpass live in LDS, over the transform image (whose last reads are done when the last
    // pass stores: fft_workgroup).  Round 4: they went through the output buffer before (8-byte stores 32 bytes apart, thread runs
    // read back twice and written once, 512 bytes apart between lanes): 1.67 GB per 2^26 samples against 0.81 GB algorithmic.
    double *zb = (double *) smem;
    auto zp = [](int i) { return i + (i >> 6); };
    fft_group<THR, B, true>(x, X, tw, tid, [&](int o, cd v) { zb[zp(4 * o + 1)] = v.re; zb[zp(4 * o + 3)] = v.im; });
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = tid + THR * b + k * (H / 4);
            zb[zp(4 * n)] = ev[b][k].re;
            zb[zp(4 * n + 2)] = ev[b][k].im;
        }
    __syncthreads();
    workgroup_lowpass<true, THR>(a, zb, ptot);
    __syncthreads();
    for (int i = tid; i < M; i += THR) out[i] = zb[zp(i)];
}

bool post_bigfft_supported(const PostArgs &a) {
    // (2048 / 4096 outputs per chunk stay with the one-wave kernel of kernels_post.hip: a workgroup per item measured 0.289 against
    // 0.251 ms per 2^29 8-bit samples at decimation 32 there, profiles/r03_experiments.txt)
    if (!(a.mode == SDRT_FM && a.fft_tw && a.fft_shift && !a.iq_R)) return false;
    // round 4: with the samples in LDS for the lowpass the packed workgroup form also takes 4096 outputs per chunk (decimation 8 of
    // 16-bit records: 0.55 -> 0.345 ms per 2^28 samples against the one-wave kernel; SDRTERM_DIAG=post_wg4096=0 keeps that one)
    if (a.m == 4096 && a.fftp_tw && a.fftp_ab && a.fft_ok) {
        const int on = sdrt_diag_int("post_wg4096", 1);
        return on != 0;
    }
    if (a.m == 32768) return a.fftp_tw && a.fftp_ab;                                     // (the packed form only)
    return (!a.pairs || (a.fftp_tw && a.fftp_ab)) && (a.m == 8192 || a.m == 16384);      // (pair products: the packed form only)
}

template <int B, bool PAIRS>
static hipError_t launch_post_bigfft_b(const PostArgs &a, hipStream_t s) {
    const size_t lds = (size_t) (1024 * B + 64 * B) * 16;
    static thread_local int done_dev = -1;
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute((const void *) post_bigfft_kernel<B, PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    hipLaunchKernelGGL((post_bigfft_kernel<B, PAIRS>), dim3((unsigned) a.items), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int B, bool PAIRS = false, int THR = 256>
static hipError_t launch_post_bigfftp_b(const PostArgs &a, hipStream_t s) {
    // the transform image [H + H/16] complex, then (over it) the M = 4 H output samples, one pad element per 64 (B: in units of 1024 points)
    const size_t img = (size_t) (1024 * B + 64 * B) * 16, seq = (size_t) (4096 * B + 64 * B) * 8;
    constexpr int BT = B * 256 / THR;                     // points per thread / 4
    const size_t lds = seq > 160 * 1024 ? img : (img > seq ? img : seq);       // (32768 outputs per chunk: the transform image alone)
    static thread_local int done_dev = -1;
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute((const void *) post_bigfftp_kernel<THR, BT, PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    hipLaunchKernelGGL((post_bigfftp_kernel<THR, BT, PAIRS>), dim3((unsigned) a.items), dim3(THR), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_post_bigfft(const PostArgs &a, hipStream_t s) {
    if (a.fftp_tw && a.fftp_ab) {                    // the packed form where its tables exist
        switch (a.m) {
            case 4096: return a.pairs ? launch_post_bigfftp_b<1, true>(a, s) : launch_post_bigfftp_b<1, false>(a, s);
            case 8192: return a.pairs ? launch_post_bigfftp_b<2, true>(a, s) : launch_post_bigfftp_b<2, false>(a, s);
            case 32768: return a.pairs ? launch_post_bigfftp_b<8, true, 512>(a, s) : launch_post_bigfftp_b<8, false, 512>(a, s);
            case 16384: {
                const int thr = sdrt_diag_int("post_thr", 512);
                if (thr == 256) return a.pairs ? launch_post_bigfftp_b<4, true>(a, s) : launch_post_bigfftp_b<4, false>(a, s);
                // (1024 threads -- four waves per SIMD at 122 registers -- measured 0.352 against 0.327 ms per 2^26 samples: r05_experiments.txt section 11)
                return a.pairs ? launch_post_bigfftp_b<4, true, 512>(a, s) : launch_post_bigfftp_b<4, false, 512>(a, s);
            }
        }
    }
    switch (a.m) {
        case 8192: return launch_post_bigfft_b<4, false>(a, s);
        case 16384: return launch_post_bigfft_b<8, false>(a, s);
    }
    return hipErrorInvalidValue;
}

bool post_big_supported(int m, int mode) {
    if (m % 256) return false;
    if (mode != SDRT_FM) return true;
    const int N = m / 2;
    return (N & (N - 1)) == 0 && N >= 4096 && (size_t) (N + 2048) * sizeof(double) <= 160 * 1024;
}

hipError_t launch_post_big(const PostArgs &a0, hipStream_t s) {
    PostArgs a = a0;
    const size_t seq = (size_t) (a.m + a.m / 64) * sizeof(double);
    a.am_in_lds = a.mode != SDRT_FM && seq <= 144 * 1024;
    const size_t lds = a.mode == SDRT_FM ? (size_t) (a.m / 2 + 2048) * sizeof(double) : (a.am_in_lds ? seq : 0);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *) post_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(post_big_kernel, dim3((unsigned) a.items), dim3(256), lds, s, a);
    return hipGetLastError();
}
#endif  // SDRT_SEQ_PART_B
