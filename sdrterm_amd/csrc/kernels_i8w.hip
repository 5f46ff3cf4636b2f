// chain_i8w_kernel: the chain kernel with WIDE matrix-core columns (256 samples) on v_mfma_i32_16x16x64_i8, persistent
// workgroups and a three-stage software pipeline.
//
// Why: with one block of L <= 64 samples per matrix-core column (kernels_i8t.hip) every 64 samples cost a recombination of
// 8 int32 sums to fp64, four DPP scan steps and an output reduction, and that fp64 vector work bounds the kernel.  Here a
// column is LC = 256 samples: the matrix cores also produce, per column, the column-local part of each of its nout = LC/q
// outputs (a third row tile), so the per-column fp64 work is spent once per 256 samples.  The digit planes for that (up to
// 120 KB) only fit LDS once per CU, hence one persistent 512-thread workgroup per CU that walks over its (chunk, row) items
// with the planes resident, and because the tile-carry and combine phases of an item are short and latency bound they are
// software-pipelined against the matrix-core phase of the following items:
//
//   iteration it:   E(it)  edge operators of item it                      (all threads, 16 outputs x 32 lanes)
//                   M(it)  matrix-core phase: wave w <-> tile w (16 columns = 4096 samples); column aggregates -> LDS,
//                          column-local outputs stay in registers of the lane that will finish them
//                   P2(it-1) tile carries of item it-1: two waves (one per real stream), DPP scans over <= 16 tiles
//                   P3(it-2) add the carried states to the outputs of item it-2 and store them
//   one workgroup barrier per iteration; the LDS carry buffers are triple buffered.
//
// Arithmetic: identical to kernels_i8t.hip (error-free int8 slicing, S digit planes, exact int32 sums, one fp64
// recombination), see DESIGN.md section 3.  The lowest (digit 0 x low slice) product of 16-bit encodings is dropped: it
// is below the quantisation step of the planes.
#include <cstdlib>
#include "kernels_common.hpp"

#ifndef SDRT_PF8
#define SDRT_PF8 1               // eight outputs per column, no IQ correction: k-steps of raw samples in flight
#endif
#ifndef SDRT_DUAL_PF
#define SDRT_DUAL_PF 3           // DUAL form: k-steps of raw samples in flight per tile
#endif
#ifndef SDRT_DUAL_VPM
#define SDRT_DUAL_VPM 3          // DUAL form: vector instructions requested after every matrix instruction
#endif
#ifndef SDRT_RAW_NT
#define SDRT_RAW_NT 0          // 1: non-temporal loads of the raw sample stream (measured: see DESIGN.md section 4)
#endif

// Wave priorities around the matrix-instruction groups.  Production: 1 inside a group, 0 outside, for every wave.  With
// -DSDRT_SCHED_EXP (a diagnostic build, tools/sched_exp.sh) DecimArgs.sched picks other assignments for the younger waves 4-7,
// which lose the issue arbitration against waves 0-3 at equal priority (profiles/r02_i8w_phase_stamps_after_2s_load.txt).
#ifdef SDRT_SCHED_EXP
#define SDRT_PRIO_SET(p) do { if ((p) == 0) __builtin_amdgcn_s_setprio(0); else if ((p) == 1) __builtin_amdgcn_s_setprio(1); else if ((p) == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); } while (0)
#define SDRT_PRIO_HI() SDRT_PRIO_SET(prio_hi)
#define SDRT_PRIO_LO() SDRT_PRIO_SET(prio_lo)
#else
#define SDRT_PRIO_HI() __builtin_amdgcn_s_setprio(1)
#define SDRT_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

namespace {
template <int DT> struct W8Enc;   // slices per component, what each slice XORs with
template <> struct W8Enc<SDRT_S8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0u, xhi = 0u; };
template <> struct W8Enc<SDRT_U8> { static constexpr int slices = 1; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };
template <> struct W8Enc<SDRT_S16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };
template <> struct W8Enc<SDRT_U16> { static constexpr int slices = 2; static constexpr unsigned xlo = 0x80808080u, xhi = 0x80808080u; };
// float32: block fixed point against the exponent of the tile (16 columns), four byte slices (kernels_i8t.hip: I8Enc<SDRT_F32>)
template <> struct W8Enc<SDRT_F32> { static constexpr int slices = 4; static constexpr unsigned xlo = 0x80808080u, xhi = 0u; };

// x' + y' after exchanging halves: lanes 0-31 get x(l) + x(l+32), lanes 32-63 get y(l-32) + y(l)
__device__ __forceinline__ double swap32_add(double x, double y) {
    const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned) __double2loint(x), (unsigned) __double2loint(y), false, false);
    const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned) __double2hiint(x), (unsigned) __double2hiint(y), false, false);
    return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}
// rows (16 lanes) 0 and 2 get x(row) + x(row+1), rows 1 and 3 get y(row-1) + y(row)
__device__ __forceinline__ double swap16_add(double x, double y) {
    const v2u lo = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(x), (unsigned) __double2loint(y), false, false);
    const v2u hi = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(x), (unsigned) __double2hiint(y), false, false);
    return __hiloint2double((int) hi[0], (int) lo[0]) + __hiloint2double((int) hi[1], (int) lo[1]);
}

// float32 -> block fixed point: 16 values (8 complex samples, one lane's k-step) against the exponent e of their tile,
// i = trunc(x 2^(30 - e)), and the four bytes of every i into four slices (dword w of slice s: byte s of values 4w .. 4w+3; the
// lower three made signed by the XOR).  One definition for the chain kernel and for the pre-slicing pass: the same bits.
__device__ __forceinline__ void f32_slices(const uint32_t *cur, const int sh, v4i (&bsl)[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned q_[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q_[u] = (unsigned) (int) ldexpf(__uint_as_float(cur[4 * w + u]), sh);
        const unsigned a01 = __builtin_amdgcn_perm(q_[1], q_[0], 0x05010400u), a23 = __builtin_amdgcn_perm(q_[3], q_[2], 0x05010400u);
        const unsigned c01 = __builtin_amdgcn_perm(q_[1], q_[0], 0x07030602u), c23 = __builtin_amdgcn_perm(q_[3], q_[2], 0x07030602u);
        bsl[0][w] = (int) (__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
        bsl[1][w] = (int) (__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
        bsl[2][w] = (int) (__builtin_amdgcn_perm(c23, c01, 0x05040100u) ^ 0x80808080u);
        bsl[3][w] = (int) __builtin_amdgcn_perm(c23, c01, 0x07060302u);
    }
}
// the exponent a tile is sliced against: that of its largest magnitude, rounded up to a multiple of four
// (an all-zero tile: -993, for which the scale 2^(E - 30) below is the double with exponent field 0, i.e. exactly 0.0 -- a silent
// tile contributes exact zeros, as in the reference, not 2^-154 of rounding residue whose FM angle would be noise)
#define SDRT_F32_SILENT (-993)
__device__ __forceinline__ int f32_tile_exponent(unsigned max_abs_bits) {
    return max_abs_bits == 0 ? SDRT_F32_SILENT : (((int) (max_abs_bits >> 23) - 126) + 3) & ~3;
}

struct WideLds {        // byte offsets
    int wmain, wloc, cF, cG, bp, mup, tpw, tailF, tailFc, psi, gv, aopen, pq, cpow, wiq, buf, buf_stride, total;
    // per buffer (offsets in 16-byte units from its start): carF [SLOTS*8], carG [SLOTS*8], F0 [8 SEG], Gu [8 SEG], IQ: tile totals of
    // the offset [NT] and the weighted sum of the last 32 samples [1]
};
// sixteen tiles per chunk, two per wave: the forms in which Q trails K by one TILE PASS instead of one iteration (see the kernel)
// (the fused IQ form needs the room of a fourth carry buffer beside its offset rows: 8-bit records only, whose packed local tile is
// three fragments instead of five)
#ifdef SDRT_NO_SHQ8      // (A/B builds: eight outputs per column on the round-2 form)
__host__ __device__ constexpr bool wide_shq(int nout, int ntc, int iq, int nsl) { return ntc == 16 && nout == 4 && (iq <= 1 || nsl == 1); }
#else
__host__ __device__ constexpr bool wide_shq(int nout, int ntc, int iq, int nsl) { return ntc == 16 && ((nout == 4 && (iq <= 1 || nsl == 1)) || (nout == 8 && iq == 0)); }
#endif
// eight outputs per column: the fourth buffer only fits beside the larger output tables if the column phasors are kept as two
// factors (tile, column in tile: 32 entries instead of 256, one complex product more per use) and the IQ-only areas are left out
__host__ __device__ constexpr bool wide_psf(int nout, int ntc, int iq) { return wide_shq(nout, ntc, iq, 1) && nout == 8; }
// eight outputs per column without IQ correction -- the forms that may be one of several passes over the chunk (decimation below 32):
// the shares from the column's END for the pass's outputs come from a second half of the table, WideTables.bp2[nout + 1 + b]; the IQ
// forms (never one of several passes, and short of LDS) keep the one table and read bp2[nout - b].  A compile-time property of the
// form: choosing at run time cost these forms 8 % in spills.
__host__ __device__ constexpr bool wide_sepba(int nout, int iq) { return nout == 8 && iq == 0; }
__host__ __device__ inline WideLds wide_lds(int S, int nout, int nt, int iq, int ntc, int nsl) {
    WideLds l;
    int p = 0;
    l.wmain = p; p += 2 * S * 8 * 1024;
    // local tile padded to 16 rows (rows >= 2 nout are zero); four outputs per column and one slice: three packed fragments
    l.wloc = p; p += (nout == 4 && nsl == 1 ? 3 : S) * 8 * 1024;
    l.cF = p; p += SDRT_TB * nout * 4 * 16;
    l.cG = p; p += SDRT_TB * nout * 4 * 16;
    l.bp = p; p += (wide_sepba(nout, iq) ? 2 : 1) * (nout + 1) * 4 * 16;      // WideTables.bp2: [b] from the column start; from its end: [nout - b], or (wide_sepba) [nout + 1 + b]
    l.mup = p; p += 4 * 4 * 16;
    l.tpw = p; p += (SDRT_WIDE_MAXTILES + 1) * 4 * 16;
    l.tailF = p; p += 16 * 16;
    l.tailFc = p; p += 16 * 16;
    const bool psf = wide_psf(nout, ntc, iq);
    l.psi = p; p += (psf ? nt + SDRT_TB : nt * SDRT_TB) * 16;      // column phasors of the row (psf: per tile, then per column in a tile)
    l.gv = p; p += psf ? 0 : 48 * 16;               // IQ, float32: rank-one vectors {g_re, g_im} per matrix row
    l.aopen = p; p += iq != 0 && nsl == 2 ? 48 * 4 : 0;   // IQ, 16-bit records: what the lowest accumulators open at, matrix row order (3 tiles x 16)
    l.pq = p; p += psf ? 0 : 16 * 2 * 16;           // IQ: offset terms of the edge operators
    l.cpow = p; p += iq == 1 ? nt * SDRT_TB * 8 : 0;       // IQ from prefix tables: (1-lambda)^(column start)
    l.wiq = p; p += iq == 2 ? (SDRT_WIDE_LC / 32) * 1024 : 0;     // IQ fused: offset rows, one packed fragment per k-step (sdrt_build_i8_offset_rows)
    l.buf = p;
    const int seg = nt / ntc, slots = seg * (ntc + 1);          // carry slots: ntc + 1 per chunk of the item
    l.buf_stride = (2 * slots * 8 + 16 * seg + (iq == 2 ? nt + 1 : 0)) * 16;      // (IQ fused: the tiles' offset totals and the tail sum)
    p += (wide_shq(nout, ntc, iq, nsl) ? 4 : 3) * l.buf_stride;
    l.total = p;
    return l;
}

}  // namespace

// RS ("real streams"): no frequency shift and no IQ correction.  The real and the imaginary sample stream are then filtered by
// the same real-coefficient operator: ONE 16-row aggregate tile applied to the re bytes and to the im bytes separately, 64
// samples per k-step, and a local tile of NOUT rows per stream in which 16 / NOUT digit planes share one instruction: 96
// (NOUT = 4) or 112 (NOUT = 8) matrix instructions per tile instead of 184 / 216.
// IQ: 0 no correction; 1 offsets at column starts from the prefix kernels' tables (iq_local_kernel / iq_scan_kernel: a second pass
// over the input, shared by all rows); 2 offsets from two more matrix rows of this kernel (no second pass; one row: see api.hip)
// NTC: tiles per chunk.  The workgroup's eight waves take eight (or sixteen) tiles per item: chunks of four tiles (float32
// records, or half-size chunks) go two to an item, SEG = 2, with carries, edge operators and column phasors per chunk.
// PRE (float32, many rows): the records arrive sliced already (f32_preslice_kernel: same bytes per lane and k-step, the four
// slices in place of the sixteen floats; tile exponents in a.f32_exp) -- every row of a chunk used to slice the same samples
// again, the larger part of this form's vector work.  The edge operators read the records themselves (a.raw_orig).
// DUAL: ONE wave per SIMD (256 threads) working on two tiles at once -- every digit-plane fragment is read from LDS once and feeds
// two matrix instructions (the fragment reads are a quarter of the kernel's energy at one read per instruction:
// tools/ubench/coexec.hip, modes 10-12), the vector work of both tiles interleaved with them in the one wave's instruction stream.
template <int DT, int NOUT, int NTC, int S, int IQ, bool RS, bool PRE = false, bool DUAL = false>
__global__ __launch_bounds__(DUAL ? 256 : 512, DUAL ? 1 : 2) void chain_i8w_kernel(DecimArgs a) {
    constexpr int WV = DUAL ? 4 : 8, NTHR = 64 * WV;           // waves, threads per workgroup
    static_assert(!DUAL || (NTC == 8 && IQ == 0 && !RS && DT != SDRT_F32 && !PRE), "two tiles at once: the plain eight-tile form");
    constexpr int NT = NTC < 8 ? 8 : NTC, SEG = NT / NTC, SLOTS = SEG * (NTC + 1);
    static_assert(SEG == 1 || (IQ == 0 && !RS), "two chunks per item: plain form only");
    static_assert(!(RS && IQ != 0), "the real-stream form has no IQ correction");
    constexpr bool IQA = IQ != 0, IQT = IQ == 1, IQF = IQ == 2;
    constexpr int LC = SDRT_WIDE_LC, KS = RS ? LC / 64 : LC / 32, SPK = LC / KS;   // k-steps per column, samples per k-step
    constexpr int NSL = W8Enc<DT>::slices;
    constexpr bool F32 = DT == SDRT_F32;
    static_assert(!F32 || (IQ == 0 && !RS), "float32: plain form only");
    static_assert(!PRE || F32, "pre-sliced input: float32 only");
    // digit d x slice s adds to level d + s; levels below LMIN lie under the accuracy of the planes (integers) or of the slices
    // themselves (float32) and are not computed; accumulator index = level - LMIN
    constexpr int LMIN = F32 ? 3 : (NSL == 2 ? 1 : 0);
    constexpr int ITEM = Enc<DT>::size;
    constexpr int NR = 2 * NOUT;                     // rows of the local tile
    constexpr int OPL = NOUT / 4;                    // outputs a lane finishes per tile
    constexpr int TPW = NT / WV;                     // tiles per wave and item
    constexpr int TP = DUAL ? TPW : 1;               // tiles a pass of the matrix phase works on at once
    // Two tiles per wave (8-bit records: a chunk is sixteen tiles).  With Q(it-1) of both tiles inside iteration it, the states K
    // leaves behind (dC, dA, dl: 24 registers per tile) wait half an iteration for their turn, and with the second set of them
    // the steady-state loop carried 40-80 scratch accesses and 20-45 full vmcnt(0) waits.  SHQ: Q trails K by one tile PASS -- the
    // pass over tile j of item `it` runs Q on the tile the previous pass finished (j = 0: tile 1 of item it-1; j = 1: tile 0 of
    // item it) -- so that what K leaves is consumed at once and only one set is alive at any time, as with one tile per wave.
    // P2 and P3 keep their places; the column-local outputs wait five passes for P3 in a queue of their own; the carry buffers
    // become four (Q of item it writes into its buffer while P3 of item it-3 still reads the one three back).
    constexpr bool SHQ = wide_shq(NOUT, NTC, IQ, W8Enc<DT>::slices) && !RS;
    constexpr int NBUF = SHQ ? 4 : 3;
    constexpr bool PSF = wide_psf(NOUT, NTC, IQ);
    // the sixteen-tile forms with --correct-iq are the shortest of registers: what they keep out of the loop-invariant lane values
    // (LDS and output addresses the allocator would spill, each reload a vmcnt(0) drain): the wave number from scalar registers,
    // the lane number read where an address needs it
    constexpr bool OPQ = (SHQ && IQ != 0) || (NOUT == 8 && IQ == 2);
    constexpr int WPK = (RS ? 8 : 4) * ITEM;         // 32-bit words per lane and k-step (8 complex samples; RS: 16)
    constexpr int TILE = SDRT_TB * LC;               // samples per tile
    constexpr int NSTEP = KS * TPW;                  // k-steps per wave and item
    // NOUT == 4: the local tile has 8 rows; two digit planes share one matrix instruction (rows 0-7 / 8-15), see below
    constexpr bool PACK = NOUT == 4 && !RS;
    // --correct-iq on 16-bit records: the accumulators of the lowest level open at a constant per row instead of zero, which makes
    // an all-zero chunk at offset zero cancel to the last bit (operators.cpp: quantise_planes, "consistent")
    constexpr bool AOPEN = IQ != 0 && NSL == 2;
    constexpr int PPF = RS ? 16 / NOUT : (PACK ? 2 : 1);         // digit planes per local-tile fragment
    constexpr int NHF = (S + PPF - 1) / PPF;                      // fragments used with the high (or only) slice
    constexpr int NLF = (PACK || RS) ? NHF + (NSL >= 2 ? (S - 1 + PPF - 1) / PPF : 0) : S;   // local-tile fragments per k-step (both plane pairings)
    constexpr int NSLOT = (RS || F32) ? 1 : 4;                             // raw-sample register ring (RS: reloaded in place once sliced)
    constexpr int PF = (NOUT == 8 && IQ == 0 && !RS && !F32) ? SDRT_PF8 : ((NOUT == 8 || RS || F32) ? 1 : (DUAL ? SDRT_DUAL_PF : 2));            // k-steps of raw samples in flight (registers are scarcer with 8 outputs per column)
    static_assert(NOUT == 4 || NOUT == 8, "4 or 8 outputs per column");
    static_assert(NTC == 4 || NTC == 8 || NTC == 16, "4, 8 or 16 tiles per chunk");
    typedef typename Enc<DT>::raw raw_t;

    extern __shared__ __align__(16) unsigned char smem[];
    const WideTables *__restrict__ wt = a.wtab;
    const int n = a.n, m = a.m;
    const bool nco = a.nco_tile != nullptr;
    const bool swap = a.swap != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nb = lane & 15;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);     // the same in scalar registers
#ifdef SDRT_SCHED_EXP
    const int young = __builtin_amdgcn_readfirstlane(wave) >= 4, sched = a.sched;
    //            sched:  0 production   1 young waves' groups at 2   2 young waves one level up throughout   3 only the young waves raise
    //                    4 no priorities at all   5 old waves' groups at 2   6 young waves at 3 inside groups, 1 outside
    const int prio_hi = sched == 0 ? 1 : sched == 1 ? (young ? 2 : 1) : sched == 2 ? (young ? 2 : 1) : sched == 3 ? (young ? 1 : 0) : sched == 4 ? 0 : sched == 5 ? (young ? 1 : 2) : (young ? 3 : 1);
    const int prio_lo = sched == 2 ? (young ? 1 : 0) : sched == 6 ? (young ? 1 : 0) : 0;
#endif
    const WideLds lo = wide_lds(S, NOUT, NT, IQ, NTC, NSL);
    const v4i *wmainL = (const v4i *) (smem + lo.wmain);
    const v4i *wlocL = (const v4i *) (smem + lo.wloc);
    cd *cFL = (cd *) (smem + lo.cF), *cGL = (cd *) (smem + lo.cG), *bpL = (cd *) (smem + lo.bp), *mupL = (cd *) (smem + lo.mup), *tpwL = (cd *) (smem + lo.tpw);
    cd *tailFL = (cd *) (smem + lo.tailF), *tailFcL = (cd *) (smem + lo.tailFc), *psiL = (cd *) (smem + lo.psi);
    cd *gvL = (cd *) (smem + lo.gv), *pqL = (cd *) (smem + lo.pq);
    const v4i *aopenL = (const v4i *) (smem + lo.aopen);          // [3 tiles][4 lane groups]: rows 4g .. 4g+3
    const v4i *wiqL = (const v4i *) (smem + lo.wiq);
    double *cpowL = (double *) (smem + lo.cpow);
    auto carF = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride); };
    auto carG = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride) + SLOTS * 8; };
    auto F0L = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride) + 2 * SLOTS * 8; };              // [seg][8]
    auto GuL = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride) + 2 * SLOTS * 8 + 8 * SEG; };
    auto OtL = [&](int b) { return (cd *) (smem + lo.buf + b * lo.buf_stride) + 2 * SLOTS * 8 + 16 * SEG; };       // [NT] + tail sum at [NT]

    // ---- which items: workgroup (xcd, slot) keeps one row (digit planes resident) and walks over chunks of its XCD ------
    const unsigned bid = blockIdx.x, xcd = bid & 7u, slot = bid >> 3, nslot = gridDim.x >> 3;
    unsigned row0, rstep, stream, nstreams;
    if ((unsigned) a.rows <= nslot) {
        nstreams = nslot / (unsigned) a.rows;
        row0 = slot % (unsigned) a.rows;
        stream = slot / (unsigned) a.rows;
        rstep = (unsigned) a.rows;
        if (stream >= nstreams) return;
    } else {
        nstreams = 1; stream = 0; row0 = slot; rstep = nslot;
    }
    // item of step k: ((k * nstreams + stream) * 8 + xcd) (= its chunk when SEG == 1); the groups of 8 items this XCD sees: gi < ngroups
    const long n_items = (long) (a.n_count ? a.n_count : a.n_chunks) / SEG;    // SEG consecutive chunks per item (the host passes a multiple)
    const long ngroups = n_items > (long) xcd ? (n_items - (long) xcd + 7) / 8 : 0;
    const int cnt = ngroups > (long) stream ? (int) ((ngroups - (long) stream + (long) nstreams - 1) / (long) nstreams) : 0;
    if (cnt == 0) return;
    auto chunk_of = [&](int k) -> long { return ((long) k * (long) nstreams + (long) stream) * 8 + (long) xcd; };

    // ---- tables that do not depend on the row -> LDS ------------------------------------------------------------------
    // stored in the order P3 reads them, [e][mode][lane]: lane (g, nb) finishes output b = g + 4e of column nb, pos = nb*NOUT + b
    for (int i = tid; i < OPL * 4 * 64; i += NTHR) {
        const int l = i & 63, md = (i >> 6) & 3, e = i >> 8, pos = (l & 15) * NOUT + (l >> 4) + 4 * e;
        cFL[i] = mk(wt->cF2[pos][md][0], wt->cF2[pos][md][1]);
        cGL[i] = mk(wt->cG2[pos][md][0], wt->cG2[pos][md][1]);
    }
    constexpr bool SEPBA_ = wide_sepba(NOUT, IQ);
    if (tid < (SEPBA_ ? 2 : 1) * (NOUT + 1) * 4) bpL[tid] = mk(wt->bp2[tid >> 2][tid & 3][0], wt->bp2[tid >> 2][tid & 3][1]);
    constexpr bool SEPBA = wide_sepba(NOUT, IQ);
    if (tid >= 64 && tid < 64 + (SDRT_WIDE_MAXTILES + 1) * 4) {
        const int i = tid - 64;
        tpwL[i] = mk(wt->tile_pow[i >> 2][i & 3][0], wt->tile_pow[i >> 2][i & 3][1]);
    }
    if (tid >= 192 && tid < 208) {
        const int i = tid - 192;
        tailFL[i] = mk(wt->tailF[i >> 2][i & 3][0], wt->tailF[i >> 2][i & 3][1]);
        tailFcL[i] = mk(wt->tailFc[i >> 2][i & 3][0], wt->tailFc[i >> 2][i & 3][1]);
    }
    if (tid >= 32 && tid < 48) mupL[tid - 32] = mk(wt->mu_pow[(tid - 32) >> 2][tid & 3][0], wt->mu_pow[(tid - 32) >> 2][tid & 3][1]);
    // edge operator of this thread: output o16 = tid >> 5 (kind: 0 head -> F0, 1 tail -> Gu; chain o = mode*2 + stream), sample l
    const int e_o16 = tid >> 5, e_l = tid & 31, e_kind = e_o16 >> 3;
    // sample l of the 32-sample window of the output: the chunk's first / last 32 samples (the operator spans 28 of them)
    const int e_idx = e_kind ? n - 32 + e_l : e_l;
    const double e_off_re = a.e_off_re, e_off_im = a.e_off_im;   // what the edge operators subtract from the decoded components
    const int cgs = LC / (IQT ? a.iq_gs : LC);                   // IQ prefix groups per column
    // IQ fused: the offset recurrence off' = (1-lambda) off + lambda x over whole columns (a.iq_consts: sdrt_build_i8_offset_rows)
    const double iq_wsum = IQF ? a.iq_consts[0] : 0.0, iq_m1 = IQF ? a.iq_consts[1] : 0.0, iq_m2 = IQF ? a.iq_consts[2] : 0.0;
    const double iq_m4 = IQF ? a.iq_consts[3] : 0.0, iq_m8 = IQF ? a.iq_consts[4] : 0.0, iq_m16 = IQF ? a.iq_consts[5] : 0.0;
    const double iq_mnb = IQF ? a.iq_consts[39 + nb] : 0.0;                 // (1-lambda)^(LC nb): tile start -> this lane's column
    const double iq_tw = IQF ? a.iq_consts[7 + (tid & 31)] : 0.0;           // weight of this thread's sample of the tail window
    const double iq_cs = a.iq_combine;
    const double cs_main0 = a.w8_combine * (LMIN == 3 ? 16777216.0 : (LMIN == 1 ? 256.0 : 1.0)), cs_loc0 = a.w8mid_combine * (LMIN == 3 ? 16777216.0 : (LMIN == 1 ? 256.0 : 1.0));
    const size_t chunk_bytes = (size_t) n * 2 * ITEM * SEG;      // bytes of one item
    // this lane's bytes of k-step s of a tile: column nb, samples SPK s + (SPK/4) g .. + SPK/4 - 1
    const size_t lane_off = ((size_t) nb * LC + (size_t) g * (SPK / 4)) * 2 * ITEM;

    for (unsigned row = row0; row < (unsigned) a.rows; row += rstep) {
        __syncthreads();                             // nobody still reads the previous row's planes
        // this row's items leave as samples or as pair products (per-row modes: FM rows products, AM rows samples)
        const bool pr_row = a.row_pair ? a.row_pair[row] != 0 : a.pair_out != 0;
        {
            constexpr int MT = RS ? 1 : 2, LR = RS ? NOUT : NR;       // aggregate tiles, rows of the compact local planes
            const v4i *wg = (const v4i *) (a.w8 + (size_t) row * (size_t) (MT * S * KS * 1024));
            v4i *dst = (v4i *) (smem + lo.wmain);
            for (int i = tid; i < MT * S * KS * 64; i += NTHR) dst[i] = wg[i];
            const v4i *wl = (const v4i *) (a.w8mid + (size_t) row * (size_t) (S * KS * 4 * LR * 16));
            v4i *dl = (v4i *) (smem + lo.wloc);
            // global layout is compact ([digit][kstep][kg][rows]); in LDS every (fragment, kstep, kg) group has 16 rows
            if constexpr (RS) {
                // NOUT rows (row b, the same for both streams): PPF = 16 / NOUT digit planes per fragment, row group rg holds
                // plane base + rg; fragments f < NHF (high / only slice): base = f PPF, the others (low slice): 1 + (f - NHF) PPF.
                // Accumulator a = fragment a (+ fragment NHF + a): row group rg at level a PPF + rg (+1).
                for (int i = tid; i < NLF * KS * 64; i += NTHR) {
                    const int f = i / (KS * 64), s_ = (i / 64) % KS, ln = i & 63, kg = ln >> 4, rowp = ln & 15;
                    const int d = (f < NHF ? f * PPF : 1 + (f - NHF) * PPF) + rowp / NOUT;
                    dl[i] = d < S ? wl[((d * KS + s_) * 4 + kg) * LR + rowp % NOUT] : (v4i) {0, 0, 0, 0};
                }
            } else if constexpr (PACK) {
                // 8 local rows (row 2b + stream): one fragment carries TWO digit planes, the lower one in rows 0-7 and the
                // higher one in rows 8-15, so that one matrix instruction does both products with the same slice.  Fragments
                // 0..2 = planes (0|1), (2|3), (4|-) for the high slice (16-bit) or the only slice (8-bit); 3, 4 = planes
                // (1|2), (3|4) for the low slice.  Accumulator a = fragment a (+ fragment 3+a): rows 0-7 hold level 2a(+1),
                // rows 8-15 the level above it.
                for (int i = tid; i < NLF * KS * 64; i += NTHR) {
                    const int f = i / (KS * 64), s_ = (i / 64) % KS, ln = i & 63, kg = ln >> 4, row = ln & 15;
                    const int lo_pl = f < 3 ? 2 * f : 2 * (f - 3) + 1;
                    const int d = row < 8 ? lo_pl : lo_pl + 1;
                    dl[i] = d < S ? wl[((d * KS + s_) * 4 + kg) * NR + (row & 7)] : (v4i) {0, 0, 0, 0};
                }
            } else {
                // 16 local rows, placed so that the matrix result lands in the lane that finishes the output: output
                // b = g + 4e, stream st <-> row 4g + 2e + st (lane g holds rows 4g .. 4g+3)
                for (int i = tid; i < S * KS * 64; i += NTHR) {
                    const int pr = i & 15, b = (pr >> 2) + 4 * ((pr >> 1) & 1), st = pr & 1;
                    dl[i] = b < NOUT ? wl[(i >> 4) * NR + 2 * b + st] : (v4i) {0, 0, 0, 0};
                }
            }
        }
        const cd *nco_blk_g = nco ? (const cd *) a.nco_phase + (size_t) row * (size_t) (n / LC) : nullptr;  // exp(i theta LC c)
        // coefficients of this thread's sample in its edge operator (NCO phasor and IQ correction of the window folded in)
        const cd *e_ab = (const cd *) a.edge_ab + (((size_t) row * 16 + e_o16) * 32 + e_l) * 2;
        cd e_A = e_ab[0], e_B = e_ab[1];             // float32 (registers are short) reloads them in every iteration instead
        // DUAL: 256 threads for 16 outputs x 32 samples: this thread's second pair, the tail output 8 + (tid >> 5)
        const cd *e_ab2 = e_ab + (DUAL ? 8 * 32 * 2 : 0);
        const cd e_A2 = DUAL ? e_ab2[0] : mk(0.0, 0.0), e_B2 = DUAL ? e_ab2[1] : mk(0.0, 0.0);
        if constexpr (F32) { if (tid < 48) gvL[tid] = ((const cd *) a.gvec)[(size_t) row * SDRT_WIDE_GV + tid]; }
        if constexpr (IQA) {
            if (tid < 48) gvL[tid] = ((const cd *) a.gvec)[(size_t) row * SDRT_WIDE_GV + tid];
            if constexpr (AOPEN) {
                // minus half the sum of every row's lowest digits (tables.h, sdrt_build_i8_weights_wide_pass), in the order the
                // matrix results land in: aggregate rows as they are, local rows as the local tile above places them
                if (tid >= 96 && tid < 144) {
                    const int i = tid - 96, pr = i & 15;
                    const int *src = (const int *) ((const cd *) a.gvec + (size_t) row * SDRT_WIDE_GV + 48);
                    int from = i;
                    if (i >= 32) {
                        if constexpr (PACK) from = pr < 8 ? 32 + pr : -1;      // rows 8-15: the level above
                        else { const int b = (pr >> 2) + 4 * ((pr >> 1) & 1); from = b < NOUT ? 32 + 2 * b + (pr & 1) : -1; }
                    }
                    ((int *) (smem + lo.aopen))[i] = from >= 0 ? src[from] : 0;
                }
            }
            if (tid >= 64 && tid < 96) pqL[tid - 64] = ((const cd *) a.edge_pq)[(size_t) row * 32 + (tid - 64)];
            if constexpr (IQT) for (int i = tid; i < NT * SDRT_TB; i += NTHR) cpowL[i] = a.iq_pow[i * cgs];
            if constexpr (IQF) for (int i = tid; i < KS * 64; i += NTHR) ((v4i *) (smem + lo.wiq))[i] = ((const v4i *) a.w8iq)[i];
        }
        if constexpr (PSF) {
            // phasor of column c = 16 t + nb as the product of the table's entries 16 t and nb
            if (tid < NT + SDRT_TB) psiL[tid] = nco ? nco_blk_g[tid < NT ? SDRT_TB * tid : tid - NT] : mk(1.0, 0.0);
        } else
        for (int i = tid; i < NT * SDRT_TB; i += NTHR) psiL[i] = nco ? nco_blk_g[i % (NTC * SDRT_TB)] : mk(1.0, 0.0);      // the phase restarts with every chunk
        const cd *ybias = (const cd *) a.ybias + (size_t) row * (NOUT == 8 ? a.m_local : m);      // always present for this kernel (zeros if there is no offset)
        __syncthreads();

        uint32_t raw[NSLOT][WPK];
        uint32_t raw2[DUAL ? NSLOT : 1][DUAL ? WPK : 1];      // DUAL: the ring of the wave's second tile
        int e_cur[TPW];                              // float32: exponent K slices this wave's tile against (before K: the guess)
#pragma unroll
        for (int j = 0; j < TPW; ++j) e_cur[j] = 0;
        auto issue = [&](const unsigned char *chunk_raw, auto sc) {
            // k-step st (0 .. NSTEP-1) of the item: tile wave + 8 (st / KS), k-step st % KS
            constexpr int st = decltype(sc)::value;
            if constexpr (DUAL) {
                // both tiles are in the same k-step at the same time: a ring per tile
                const unsigned char *p = chunk_raw + ((size_t) (wave + WV * (st / KS)) * TILE + (size_t) (st % KS) * SPK) * 2 * ITEM + lane_off;
                if constexpr (st / KS == 0) load_words<WPK * 4, SDRT_RAW_NT != 0>(p, raw[(st % KS) % NSLOT]);
                else load_words<WPK * 4, SDRT_RAW_NT != 0>(p, raw2[(st % KS) % NSLOT]);
            } else
            if constexpr (SHQ) {
                // a scalar base and a 32-bit lane offset (as one 64-bit value per lane the address is a loop invariant the register
                // allocator spills; its reload waits with vmcnt(0))
                const unsigned char *p = chunk_raw + ((size_t) (wave_s + WV * (st / KS)) * TILE + (size_t) (st % KS) * SPK) * 2 * ITEM;
                load_words<WPK * 4, SDRT_RAW_NT != 0>(p + (unsigned) lane_off, raw[st % NSLOT]);
            } else {
            const unsigned char *p = chunk_raw + ((size_t) (wave + WV * (st / KS)) * TILE + (size_t) (st % KS) * SPK) * 2 * ITEM + lane_off;
            load_words<WPK * 4, SDRT_RAW_NT != 0>(p, raw[st % NSLOT]);
            }
        };
        // float32: the exponent of a tile (|x| < 2^e for all of its samples) is not known before the tile has been read, and a
        // scan ahead of the matrix phase would read the stream twice (measured: the fabric, not the CU, then sets the rate).
        // Instead the tile is sliced against a guess -- the exponent its predecessor (same wave, previous item) used -- while
        // the largest magnitude is tracked beside the slicing; the exponent to use is the tile's own, rounded up to a multiple
        // of four (27 to 30 bits below the largest sample; adjacent tiles of a stream of even level agree on it), and a guess
        // that differs sends the wave through the matrix phase of that tile once more.
        auto wave_max = [&](unsigned v) {
            // all-reduce without LDS: rotations inside the 16-lane rows (DPP row_ror), then the rows exchanged by permlane swaps
            v = max(v, (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x121, 0xf, 0xf, false));
            v = max(v, (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x122, 0xf, 0xf, false));
            v = max(v, (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x124, 0xf, 0xf, false));
            v = max(v, (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x128, 0xf, 0xf, false));
            const v2u a16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
            v = max(a16[0], a16[1]);
            const v2u a32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
            return max(a32[0], a32[1]);
        };
        {
            const unsigned char *c0 = (const unsigned char *) a.raw + (size_t) chunk_of(0) * chunk_bytes;
            static_for<0, PF>([&](auto sc) { issue(c0, sc); });
            if constexpr (DUAL) static_for<0, PF>([&](auto sc) { issue(c0, std::integral_constant<int, KS + decltype(sc)::value>{}); });
        }
        // Software pipeline over the items k = 0 .. cnt-1 of this workgroup; in iteration `it`:
        //   K(it)     matrix-core phase of item it (all waves; wave w <-> tiles w, w+8): int32 sums -> fp64 column aggregates,
        //             kept in registers (dC, dA, dl) for the next iteration
        //   Q(it-1)   phasor, column scans, tile aggregates -> LDS, column-local outputs -> registers (yq)
        //   P2(it-2)  tile carries (two waves, one per real stream)
        //   P3(it-3)  carried states -> outputs, store
        //   E(it)     edge operators -> LDS
        // Q, P3 and E are cut into short pieces that run between the groups of matrix instructions of K, so that the wave
        // sharing the SIMD finds the matrix pipe free meanwhile (and vice versa).  LDS carry buffers: item k uses k % 3.
        cd yq[2][TPW][OPL], ynew[TPW][OPL];          // column-local outputs: [0] two iterations old (P3 finishes them now)
        cd yqs[SHQ ? 5 : 1][OPL];                    // SHQ: the same as a queue over tile passes ([0]: five passes old)
        cd dC[TPW][2], dA[TPW][2];                   // [stream]: causal / anticausal states of (mode g, column nb) of the tile
        double dli[RS ? TPW : 1][4];                 // RS: local rows of the imaginary stream (dl: of the real stream)
        double dl[TPW][4];                           // local rows of this lane: 8 outputs per column: its own outputs g + 4e,
                                                     // [2e + stream]; 4 per column: outputs 2 (g & 1) + (v >> 1), stream v & 1,
                                                     // lower levels in lanes g < 2, higher levels in lanes g >= 2
        cd pv[TPW][OPL];                             // P3 accumulators
        double ccon[TPW];                            // float32: the constant the slices leave out, in the tile's scale (item of the last K)
        cd oex[TPW];                                 // IQ: offset at this lane's column relative to its tile's start (item of the last K)
        double part[DUAL ? 2 : 1][4][2];             // (DUAL: the pieces of both tiles alternate, each tile keeps its own)
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
#pragma unroll
            for (int e = 0; e < OPL; ++e) yq[0][j][e] = yq[1][j][e] = ynew[j][e] = pv[j][e] = mk(0.0, 0.0);
#pragma unroll
            for (int e = 0; e < OPL; ++e)
#pragma unroll
                for (int d = 0; d < (SHQ ? 5 : 1); ++d) yqs[d][e] = mk(0.0, 0.0);
            dC[j][0] = dC[j][1] = dA[j][0] = dA[j][1] = mk(0.0, 0.0);
            oex[j] = mk(0.0, 0.0);
            ccon[j] = 0.0;
#pragma unroll
            for (int v = 0; v < 4; ++v) dl[j][v] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < (RS ? TPW : 1); ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v) dli[j][v] = 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) part[0][b][0] = part[0][b][1] = part[DUAL ? 1 : 0][b][0] = part[DUAL ? 1 : 0][b][1] = 0.0;

        int bcur = 0;                                // it % NBUF
        // One iteration; HAVE_K = there is an item whose matrix-core phase runs now (false only in the three drain iterations).
        // With HAVE_K every global load and store below is issued unconditionally, so that the compiler can count them and
        // wait for exactly the k-step it needs (s_waitcnt vmcnt(N)) instead of draining the whole prefetch ring.
        // FULL: the pipeline is full (3 <= it < cnt), every phase has its item: the steady-state copy of the iteration carries no
        // branch on what exists (each cost three scalar instructions and cut the iteration into separate scheduling regions; this
        // kernel is bound by the instructions it issues, scalar ones included: profiles/r03_experiments.txt)
        auto iteration = [&](auto hk, auto fullc, const int it) {
            constexpr bool HAVE_K = decltype(hk)::value, FULL = decltype(fullc)::value;
            static_assert(HAVE_K || !FULL, "the steady state has a matrix-core phase");
            // (it-1), (it-2), (it-3) % NBUF
            const int bQ = NBUF == 3 ? (bcur == 0 ? 2 : bcur - 1) : ((bcur + 3) & 3), bP2 = NBUF == 3 ? (bcur == 2 ? 0 : bcur + 1) : ((bcur + 2) & 3);
            const int bP3 = NBUF == 3 ? bcur : ((bcur + 1) & 3);
            const bool have_q = FULL || (it >= 1 && it - 1 < cnt), have_p2 = FULL || (it >= 2 && it - 2 < cnt), have_p3 = FULL || it >= 3;
#ifdef SDRT_STAMPS
            // diagnostic build (tools/stamps_wide.py): phase stamps of iteration 6
            unsigned long long *stp = (a.stamps && it == 6 && (int) bid * 2 < a.stamp_wgs && lane == 0) ? a.stamps + ((size_t) bid * 8 + wave) * 16 : nullptr;
#define SDRT_STAMP(i) do { if (stp) stp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SDRT_STAMP(i) do { } while (0)
#endif
            SDRT_STAMP(0);
            // small global loads first: everything issued later (the raw stream) is younger, so waiting for these never
            // waits for the prefetch ring
            const unsigned char *chunk_raw = (const unsigned char *) a.raw + (size_t) chunk_of(HAVE_K ? it : 0) * chunk_bytes;
            const unsigned char *edge_raw = PRE ? (const unsigned char *) a.raw_orig + (size_t) chunk_of(HAVE_K ? it : 0) * chunk_bytes : chunk_raw;
            if constexpr (F32) { e_A = e_ab[0]; e_B = e_ab[1]; }
            if constexpr (PRE && HAVE_K) {
#pragma unroll
                for (int j = 0; j < TPW; ++j) e_cur[j] = a.f32_exp[(size_t) chunk_of(it) * NT + wave + WV * j];
            }
            raw_t e_re[SEG], e_im[SEG];              // raw edge sample of E(it) per chunk of the item (lanes beyond the pad read sample 27, weight 0)
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg) {
                e_re[sg] = e_im[sg] = 0;
                if constexpr (HAVE_K) {
                    // (a 32-bit lane offset on the chunk pointer: a 64-bit one is a loop invariant the register allocator spills and
                    // reloads in the loop's preheader, and its first use inside the loop then waits with s_waitcnt vmcnt(0) in EVERY
                    // iteration -- a drain of the prefetch ring at the top of each one)
                    const raw_t *rp = (const raw_t *) (edge_raw + (unsigned) (2 * (e_idx + sg * n)) * (unsigned) sizeof(raw_t));
                    e_re[sg] = rp[0];
                    e_im[sg] = rp[1];
                }
            }
            raw_t e_re2 = 0, e_im2 = 0;              // DUAL: sample e_l of the tail window
            if constexpr (DUAL && HAVE_K) {
                const raw_t *rp = (const raw_t *) (edge_raw + (unsigned) (2 * (n - 32 + e_l)) * (unsigned) sizeof(raw_t));
                e_re2 = rp[0];
                e_im2 = rp[1];
            }
            cd yb[TPW][OPL];
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int e = 0; e < OPL; ++e) {
                    // float32: no constant left out; --correct-iq: the offset term carries it (the registers of those forms do not hold a
                    // ybias besides: their residue for a silent stream with a zero offset, ~2^-39 of the offset, stays -- DESIGN section 1).
                    // (+0.0 still turns a -0.0 into +0.0: angle(-0) would be pi where the reference's exact zeros give 0)
                    if constexpr (IQA || F32) yb[j][e] = mk(0.0, 0.0);
                    else yb[j][e] = ybias[(((wave + WV * j) % NTC) * SDRT_TB + nb) * NOUT + g + 4 * e];
                }
            cd off_head = mk(0.0, 0.0), off_tail = mk(0.0, 0.0), off_q = mk(0.0, 0.0), gq[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) gq[j] = mk(0.0, 0.0);
            if constexpr (IQT) {
                // offsets from the prefix kernels: at the chunk start, at the tail window and at this lane's columns of item it-1
                const long ck = chunk_of(HAVE_K ? it : 0), cq = chunk_of(have_q ? it - 1 : 0);
                const int ngrp = n / a.iq_gs, gt = (n - 32) / a.iq_gs;
                off_head = ((const cd *) a.iq_chunk_off)[ck];
                const cd gtl = ((const cd *) a.iq_group_off)[(size_t) ck * ngrp + gt];
                const double ptl = a.iq_pow[gt];
                off_tail = mk(fma(ptl, off_head.re, gtl.re), fma(ptl, off_head.im, gtl.im));
                off_q = ((const cd *) a.iq_chunk_off)[cq];
                // the same value in every lane: keep them in scalar registers
                auto uni = [](double v) {
                    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
                };
                off_head = mk(uni(off_head.re), uni(off_head.im));
                off_tail = mk(uni(off_tail.re), uni(off_tail.im));
                off_q = mk(uni(off_q.re), uni(off_q.im));
                // (SHQ: tile slot 1 is tile 1 of item it-1, tile slot 0 is tile 0 of item it)
#pragma unroll
                for (int j = 0; j < TPW; ++j)
                    gq[j] = ((const cd *) a.iq_group_off)[(size_t) (SHQ && j == 0 ? ck : cq) * ngrp + (size_t) ((wave + WV * j) * SDRT_TB + nb) * cgs];
            }
            if constexpr (IQF) {
                // Offsets of item it-1, relative to a zero offset at the start of its chunk (the chunk's true starting offset
                // enters after the kernel: the chain is linear in it, api.hip).  K(it-1) left the offset at this lane's column
                // relative to the tile start (oex) and every tile's total in LDS: chain the tiles before this one.
                if (have_q) {
                    const cd *ot = OtL(bQ);
                    // (SHQ: tile slot 1 of item it-1 here; tile slot 0 of item it follows the barrier between the two passes)
#pragma unroll
                    for (int j = SHQ ? 1 : 0; j < TPW; ++j) {
                        cd ts = mk(0.0, 0.0);
                        for (int t2 = 0; t2 < wave_s + WV * j; ++t2) ts = mk(fma(iq_m16, ts.re, ot[t2].re), fma(iq_m16, ts.im, ot[t2].im));    // (a scalar trip count)
                        gq[j] = mk(fma(iq_mnb, ts.re, oex[j].re), fma(iq_mnb, ts.im, oex[j].im));
                    }
                    if (wave_s == 0) {
                        // the offset at the chunk's end -> chunk aggregate (the scan over chunks follows the kernel), and the
                        // offset at the first sample of the tail window, unwound over the window's 32 samples: its share of the
                        // tail edge operators (8 outputs) goes into Gu before P2 reads it
                        cd te = mk(0.0, 0.0);
                        for (int t2 = 0; t2 < NT; ++t2) te = mk(fma(iq_m16, te.re, ot[t2].re), fma(iq_m16, te.im, ot[t2].im));
                        const cd tsum = ot[NT];
                        const cd ow = mk((te.re - tsum.re) * a.iq_consts[6], (te.im - tsum.im) * a.iq_consts[6]);
                        ((cd *) a.iq_chunk_agg)[chunk_of(it - 1)] = te;        // (every lane of wave 0, the same value: no lane mask around a store)
                        if (lane < 8) {
                            const cd P = pqL[2 * (8 + lane)], Q = pqL[2 * (8 + lane) + 1];
                            cd gu = GuL(bQ)[lane];
                            gu.re = fma(-P.re, ow.re, fma(-Q.re, ow.im, gu.re));
                            gu.im = fma(-P.im, ow.re, fma(-Q.im, ow.im, gu.im));
                            GuL(bQ)[lane] = gu;
                        }
                    }
                }
            }
            // ---- P2(it-2): tile carries, one wave per real stream; DPP rows = modes, lanes of a row = tiles ------------------
            if (have_p2 && (wave >> 1) == (it & 1)) {     // waves 0-3 win the issue arbitration on their SIMDs and have the slack
                // lane t <-> tile t of the item = tile tin of its chunk sg; carry slot sl: NTC + 1 slots per chunk.  The scans run
                // inside a chunk only: a step that would reach across the chunk boundary gets a zero multiplier
                const int st = wave & 1, o = g * 2 + st, t = nb, sg = t < NT ? t / NTC : 0, tin = t - sg * NTC, sl = t + sg;
                cd *cf = carF(bP2), *cg = carG(bP2);
                auto mstep = [&](int d, bool ok) { return (d < NTC && ok) ? tpwL[d * 4 + g] : mk(0.0, 0.0); };
                cd x = t < NT ? cf[(sl + 1) * 8 + o] : mk(0.0, 0.0);
                x = cfma(mstep(1, tin >= 1), dpp_row<0x111>(x), x);
                x = cfma(mstep(2, tin >= 2), dpp_row<0x112>(x), x);
                if constexpr (NTC > 4) x = cfma(mstep(4, tin >= 4), dpp_row<0x114>(x), x);
                if constexpr (NTC > 8) x = cfma(mstep(8, tin >= 8), dpp_row<0x118>(x), x);
                const cd f0 = F0L(bP2)[sg * 8 + o];
                x = cfma(tpwL[(t < NT ? tin + 1 : 0) * 4 + g], f0, x);
                if (t < NT) cf[(sl + 1) * 8 + o] = x;
                if (t < NT && tin == 0) cf[sl * 8 + o] = f0;
                // anticausal state at the chunk's right edge from the causal states there (all four modes of this stream) + edge samples
                cd G = GuL(bP2)[sg * 8 + o];
                const int send = sg * (NTC + 1) + NTC;          // slot of the chunk's end
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const cd Fj = cf[send * 8 + j * 2 + st];
                    G = cfma(tailFL[g * 4 + j], Fj, G);
                    G = cfma(tailFcL[g * 4 + j], mk(Fj.re, -Fj.im), G);
                }
                cd y = t < NT ? cg[sl * 8 + o] : mk(0.0, 0.0);
                y = cfma(mstep(1, tin + 1 < NTC), dpp_row<0x101>(y), y);
                y = cfma(mstep(2, tin + 2 < NTC), dpp_row<0x102>(y), y);
                if constexpr (NTC > 4) y = cfma(mstep(4, tin + 4 < NTC), dpp_row<0x104>(y), y);
                if constexpr (NTC > 8) y = cfma(mstep(8, tin + 8 < NTC), dpp_row<0x108>(y), y);
                y = cfma(tpwL[(t < NT ? NTC - tin : 0) * 4 + g], G, y);
                if (t < NT) cg[sl * 8 + o] = y;
                if (t < NT && tin == 0) cg[send * 8 + o] = G;
            }
            SDRT_STAMP(1);

            // ---- the pieces of Q(it-1), P3(it-3) and E(it) that run after matrix-instruction group i of tile slot j ------------
            constexpr int P3_0 = 12 + 3 * OPL;       // first P3 piece; the E piece comes after the last one
            auto col_phasor = [&](int t) -> cd {
                if constexpr (PSF) {
                    const cd pa = psiL[t], pb = psiL[NT + nb];
                    return mk(fma(pa.re, pb.re, -(pa.im * pb.im)), fma(pa.re, pb.im, pa.im * pb.re));
                } else {
                    return psiL[t * SDRT_TB + nb];
                }
            };
            auto lane_id = [&]() -> int {
                if constexpr (OPQ) {
                    int l;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
                    return l;
                } else {
                    return lane;
                }
            };
            auto piece = [&](auto jc, auto ic) {
                constexpr int j = decltype(jc)::value, i = decltype(ic)::value;
                const int wv = OPQ ? wave_s : wave;
                constexpr int pj = DUAL ? j : 0;
                const int tile = wv + WV * j;
                // what Q works on in this pass: tile j of item it-1, or (SHQ) the tile the previous pass finished
                constexpr int jq = SHQ ? (j ^ 1) : j;
                const int tileq = wv + WV * jq, bQq = SHQ && j == 1 ? bcur : bQ;
                const bool have_qq = SHQ && j == 1 ? HAVE_K : have_q;
                if (have_qq) {
                    if constexpr ((IQA || F32) && i == 0) {
                        // IQ correction, the part that is linear in the offset at the column's first sample:
                        // row value += (c - off).re g_re[row] + (c - off).im g_im[row]
                        const double cp = IQT ? cpowL[tileq * SDRT_TB + nb] : 0.0;        // fused: zero offset at the chunk start
                        // float32: no offset, only the constant the XOR left out of the three lower slices, in the tile's scale
                        const cd offc = SHQ && j == 1 ? off_head : off_q;             // offset at the start of the tile's chunk
                        const double dr = F32 ? ccon[jq] : a.c_re - fma(cp, offc.re, gq[jq].re), di = F32 ? ccon[jq] : a.c_im - fma(cp, offc.im, gq[jq].im);
                        const cd g0 = gvL[4 * g + 0], g1 = gvL[4 * g + 1], g2 = gvL[4 * g + 2], g3 = gvL[4 * g + 3];
                        dC[jq][0] = mk(fma(dr, g0.re, fma(di, g0.im, dC[jq][0].re)), fma(dr, g1.re, fma(di, g1.im, dC[jq][0].im)));
                        dC[jq][1] = mk(fma(dr, g2.re, fma(di, g2.im, dC[jq][1].re)), fma(dr, g3.re, fma(di, g3.im, dC[jq][1].im)));
                        const cd h0 = gvL[16 + 4 * g + 0], h1 = gvL[16 + 4 * g + 1], h2 = gvL[16 + 4 * g + 2], h3 = gvL[16 + 4 * g + 3];
                        dA[jq][0] = mk(fma(dr, h0.re, fma(di, h0.im, dA[jq][0].re)), fma(dr, h1.re, fma(di, h1.im, dA[jq][0].im)));
                        dA[jq][1] = mk(fma(dr, h2.re, fma(di, h2.im, dA[jq][1].re)), fma(dr, h3.re, fma(di, h3.im, dA[jq][1].im)));
                        if constexpr (PACK) {
                            // local values of output 2 (g & 1) + (v >> 1), stream v & 1, split over the lanes g and g + 2: once
                            const double lowhalf = g < 2 ? 1.0 : 0.0, dr2 = dr * lowhalf, di2 = di * lowhalf;
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const cd l0 = gvL[32 + 4 * (g & 1) + v];
                                dl[jq][v] = fma(dr2, l0.re, fma(di2, l0.im, dl[jq][v]));
                            }
                        } else {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {             // local rows of this lane: output g + 4 (v >> 1), stream v & 1
                                const cd l0 = gvL[32 + 2 * (g + 4 * (v >> 1)) + (v & 1)];
                                dl[jq][v] = fma(dr, l0.re, fma(di, l0.im, dl[jq][v]));
                            }
                        }
                    } else if constexpr (i == 1 && !RS) {
                        const cd ps = col_phasor(tileq);               // the column's common phasor multiplies the complex DATA
                        // (explicit fma everywhere in the pieces: they exist in two compiled copies, steady state and drain,
                        // and both must round identically -- an item's bits must not depend on the batch it arrived in)
                        cd t;
                        t = mk(fma(ps.re, dC[jq][0].re, -(ps.im * dC[jq][1].re)), fma(ps.re, dC[jq][0].im, -(ps.im * dC[jq][1].im)));
                        dC[jq][1] = mk(fma(ps.im, dC[jq][0].re, ps.re * dC[jq][1].re), fma(ps.im, dC[jq][0].im, ps.re * dC[jq][1].im));
                        dC[jq][0] = t;
                        t = mk(fma(ps.re, dA[jq][0].re, -(ps.im * dA[jq][1].re)), fma(ps.re, dA[jq][0].im, -(ps.im * dA[jq][1].im)));
                        dA[jq][1] = mk(fma(ps.im, dA[jq][0].re, ps.re * dA[jq][1].re), fma(ps.im, dA[jq][0].im, ps.re * dA[jq][1].im));
                        dA[jq][0] = t;
                    } else if constexpr (i == 2 && !RS) {
                        const cd ps = col_phasor(tileq);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const double t2 = fma(ps.re, dl[jq][2 * h], -(ps.im * dl[jq][2 * h + 1]));
                            dl[jq][2 * h + 1] = fma(ps.im, dl[jq][2 * h], ps.re * dl[jq][2 * h + 1]);
                            dl[jq][2 * h] = t2;
                        }
                    } else if constexpr (i >= 3 && i <= 10) {
                        // inclusive scans over the 16 columns of the tile (lanes of one DPP row): step (i-3)/2, causal then anticausal
                        constexpr int stp_ = (i - 3) / 2;
                        const cd mu = mupL[stp_ * 4 + g];
                        if constexpr (((i - 3) & 1) == 0) {
                            dC[jq][0] = cfma(mu, dpp_row<0x110 + (1 << stp_)>(dC[jq][0]), dC[jq][0]);
                            dC[jq][1] = cfma(mu, dpp_row<0x110 + (1 << stp_)>(dC[jq][1]), dC[jq][1]);
                        } else {
                            dA[jq][0] = cfma(mu, dpp_row<0x100 + (1 << stp_)>(dA[jq][0]), dA[jq][0]);
                            dA[jq][1] = cfma(mu, dpp_row<0x100 + (1 << stp_)>(dA[jq][1]), dA[jq][1]);
                        }
                    } else if constexpr (i == 11) {
                        // tile aggregates: causal inclusive at the last column -> slot tile+1, anticausal inclusive at the first -> slot tile
                        cd *cf = carF(bQq), *cg = carG(bQq);
                        const int sl = tileq + tileq / NTC;          // carry slot of the tile's start (NTC + 1 slots per chunk)
                        const int ga = lane_id() >> 4;            // (= g)
                        if (nb == 15) { cf[(sl + 1) * 8 + ga * 2 + 0] = dC[jq][0]; cf[(sl + 1) * 8 + ga * 2 + 1] = dC[jq][1]; }
                        if (nb == 0) { cg[sl * 8 + ga * 2 + 0] = dA[jq][0]; cg[sl * 8 + ga * 2 + 1] = dA[jq][1]; }
                        // from here on: states at the column's edges from inside the tile (causal at its start, anticausal at its end)
                        dC[jq][0] = dpp_row<0x111>(dC[jq][0]);
                        dC[jq][1] = dpp_row<0x111>(dC[jq][1]);
                        dA[jq][0] = dpp_row<0x101>(dA[jq][0]);
                        dA[jq][1] = dpp_row<0x101>(dA[jq][1]);
                    } else if constexpr (i >= 12 && i < P3_0) {
                        constexpr int e = (i - 12) / 3, sub = (i - 12) % 3;
                        if constexpr (sub < 2) {
                            // this mode's share of outputs b of the column
#pragma unroll
                            for (int bb = 2 * sub; bb < 2 * sub + 2; ++bb) {
                                const int b = 4 * e + bb;
                                const cd bc = bpL[b * 4 + g], ba = bpL[(SEPBA ? NOUT + 1 + b : NOUT - b) * 4 + g];
                                part[pj][bb][0] = fma(bc.re, dC[jq][0].re, fma(-bc.im, dC[jq][0].im, fma(ba.re, dA[jq][0].re, -(ba.im * dA[jq][0].im))));
                                part[pj][bb][1] = fma(bc.re, dC[jq][1].re, fma(-bc.im, dC[jq][1].im, fma(ba.re, dA[jq][1].re, -(ba.im * dA[jq][1].im))));
                                if constexpr (PACK) {
                                    // the column-local part: lanes g and g + 2 hold its lower and higher levels for outputs
                                    // 2 (g & 1), 2 (g & 1) + 1; the sum over the mode lanes below adds them up
                                    const double mine = ((g & 1) == (b >> 1)) ? 1.0 : 0.0;
                                    part[pj][bb][0] = fma(mine, dl[jq][2 * (b & 1)], part[pj][bb][0]);
                                    part[pj][bb][1] = fma(mine, dl[jq][2 * (b & 1) + 1], part[pj][bb][1]);
                                } else if constexpr (RS && NOUT == 4) {
                                    // every lane holds one level group of all four outputs (dl: real stream, dli: imaginary)
                                    part[pj][bb][0] += dl[jq][bb];
                                    part[pj][bb][1] += dli[jq][bb];
                                } else if constexpr (RS) {
                                    // lanes g hold outputs 4 (g & 1) .. + 3, level groups g >> 1
                                    const double mine = ((g & 1) == e) ? 1.0 : 0.0;
                                    part[pj][bb][0] = fma(mine, dl[jq][bb], part[pj][bb][0]);
                                    part[pj][bb][1] = fma(mine, dli[jq][bb], part[pj][bb][1]);
                                }
                            }
                        } else {
                            // sum over the four modes (lanes 16 apart) as a reduce-scatter: lane g ends up with output g + 4e
                            const double r0 = swap32_add(part[pj][0][0], part[pj][2][0]), i0 = swap32_add(part[pj][0][1], part[pj][2][1]);
                            const double r1 = swap32_add(part[pj][1][0], part[pj][3][0]), i1 = swap32_add(part[pj][1][1], part[pj][3][1]);
                            if constexpr (PACK || RS) ynew[jq][e] = mk(swap16_add(r0, r1), swap16_add(i0, i1));
                            // ... plus the column-local part, which the matrix cores left in this very lane
                            else ynew[jq][e] = mk(swap16_add(r0, r1) + dl[jq][2 * e], swap16_add(i0, i1) + dl[jq][2 * e + 1]);
                        }
                    }
                }
                if constexpr (i >= P3_0 && i < P3_0 + 4) {
                    if (have_p3) {
                        constexpr int mode = i - P3_0;
                        const cd *cf = carF(bP3), *cg = carG(bP3);
                        const int sl = tile + tile / NTC;
                        const cd Fa = cf[sl * 8 + mode * 2 + 0], Fb = cf[sl * 8 + mode * 2 + 1];
                        const cd Ga = cg[(sl + 1) * 8 + mode * 2 + 0], Gb = cg[(sl + 1) * 8 + mode * 2 + 1];
#pragma unroll
                        for (int e = 0; e < OPL; ++e) {
                            const cd f = cFL[(e * 4 + mode) * 64 + lane], gq = cGL[(e * 4 + mode) * 64 + lane];
                            if constexpr (mode == 0) pv[j][e] = SHQ ? yqs[0][e] : yq[0][j][e];
                            pv[j][e].re = fma(f.re, Fa.re, fma(-f.im, Fa.im, fma(gq.re, Ga.re, fma(-gq.im, Ga.im, pv[j][e].re))));
                            pv[j][e].im = fma(f.re, Fb.re, fma(-f.im, Fb.im, fma(gq.re, Gb.re, fma(-gq.im, Gb.im, pv[j][e].im))));
                            // the response to the constant the slices leave out goes in LAST: it is minus what this very sequence of
                            // operations makes of an all-zero chunk (api.hip: calibrate_zero_response), so a silent chunk cancels to the bit
                            if constexpr (mode == 3) pv[j][e] = mk(pv[j][e].re + yb[j][e].re, pv[j][e].im + yb[j][e].im);
                        }
                    }
                } else if constexpr (i == P3_0 + 4) {
                    // the decimated samples leave as they are, or (pair_out: FM without a debug buffer) as the products
                    // y[2i] conj(y[2i+1]) the discriminator takes the angle of (demodulation.py:25-32): half the bytes written here
                    // and read by the demodulation stage.  Outputs g and g + 1 sit in lanes 16 apart: one permlane swap per dword;
                    // one store instruction either way, and no lane masked off: for pairs the lanes of odd g form the same product
                    // from the same operands as their partners and write it to the same place (a store under a branch costs the
                    // exact vmcnt accounting of everything after it)
                    const bool pr = pr_row;
                    // outputs of item it-3 (the first three iterations store to item 0's place, which P3(0) rewrites in iteration 3).
                    // The address is put together here, from scalars, where it is used: held from the top of the iteration it is
                    // a 64-bit value the register allocator spills, and its reload waits with vmcnt(0) -- a drain of the raw-sample
                    // prefetch ring in every iteration
                    // (and it is the same in every lane: in scalar registers, the lanes add 32-bit byte offsets)
                    const uint64_t ydv = (uint64_t) (a.ydec + ((size_t) row * a.n_chunks + (size_t) chunk_of(have_p3 ? it - 3 : 0) * SEG) * (size_t) m * ((pr && !a.pair_wide) ? 1 : 2));
                    // (an address built from integers must say that it is global memory, or the store becomes a flat one)
                    typedef __attribute__((address_space(1))) char gchar;
                    typedef double v2d_ __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(1))) v2d_ gdouble2;
                    gchar *yd = (gchar *) (((uint64_t) (unsigned) __builtin_amdgcn_readfirstlane((int) (ydv >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int) ydv));
#pragma unroll
                    for (int e = 0; e < OPL; ++e) {
                        const int la = lane_id();
                        // (column nb, output g + 4e); eight outputs per column: possibly one pass of several over the chunk (decimation
                        // below 32: a column has out_mul = 256 / q outputs, this pass finishes out_off .. out_off + 7 of them)
                        const int k = NOUT == 8 ? (tile * SDRT_TB + (la & 15)) * a.out_mul + a.out_off + (la >> 4) + 4 * e
                                                : (tile * SDRT_TB + (la & 15)) * NOUT + (la >> 4) + 4 * e;
                        const cd own = pv[j][e];
                        const v2u s0 = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(own.re), (unsigned) __double2loint(own.re), false, false);
                        const v2u s1 = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(own.re), (unsigned) __double2hiint(own.re), false, false);
                        const v2u s2 = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(own.im), (unsigned) __double2loint(own.im), false, false);
                        const v2u s3 = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(own.im), (unsigned) __double2hiint(own.im), false, false);
                        // the swap of x with itself leaves {row 0, row 0, row 2, row 2} in its first result and {row 1, row 1, row 3,
                        // row 3} in its second: every lane of a row pair holds the even output in the one, the odd output in the other
                        const cd za = mk(__hiloint2double((int) s1[0], (int) s0[0]), __hiloint2double((int) s3[0], (int) s2[0]));     // output k & ~1
                        const cd zb = mk(__hiloint2double((int) s1[1], (int) s0[1]), __hiloint2double((int) s3[1], (int) s2[1]));     // output (k & ~1) + 1
                        const cd z = mk(fma(za.re, zb.re, za.im * zb.im), fma(za.im, zb.re, -(za.re * zb.im)));
                        const cd o = pr ? z : own;
                        *(gdouble2 *) (yd + (unsigned) (pr ? 8 * (k & ~1) : 16 * k)) = (v2d_) {o.re, o.im};
                    }
                }
                if constexpr (HAVE_K && j == 0 && i == P3_0 + 5) {
                    // E(it): this thread's term of its edge operator, summed over the 32 lanes of the output: inside the DPP
                    // rows, then across the two rows of each half
#pragma unroll
                    for (int sg = 0; sg < SEG; ++sg) {
                    const double xr = decode<DT>(e_re[sg], swap) - e_off_re, xi = decode<DT>(e_im[sg], swap) - e_off_im;
                    cd f = mk(fma(e_A.re, xr, e_B.re * xi), fma(e_A.im, xr, e_B.im * xi));
                    // IQ fused: the offset at the head window's first sample is zero in chunk-local terms; the tail window's share is
                    // added one iteration later, when the chunk's offsets are known.  What that needs from here: the weighted
                    // sum of the window's 32 samples (the lanes of output 8 store it)
                    cd tsw = mk(iq_tw * xr, iq_tw * xi);
                    if constexpr (IQT) {
                        if (e_l == 0) {                          // the offset at the window's first sample, once per output
                            const cd ow = e_kind ? off_tail : off_head, P = pqL[2 * e_o16], Q = pqL[2 * e_o16 + 1];
                            f.re = fma(-P.re, ow.re, fma(-Q.re, ow.im, f.re));
                            f.im = fma(-P.im, ow.re, fma(-Q.im, ow.im, f.im));
                        }
                    }
                    if constexpr (IQF) {
                        tsw = cadd(tsw, dpp_row<0x111>(tsw));
                        tsw = cadd(tsw, dpp_row<0x112>(tsw));
                        tsw = cadd(tsw, dpp_row<0x114>(tsw));
                        tsw = cadd(tsw, dpp_row<0x118>(tsw));
                        const double tot2 = swap16_add(tsw.re, tsw.im);
                        if (nb == 15 && e_o16 == 8) ((double *) (OtL(bcur) + NT))[(lane_id() >> 4) & 1] = tot2;
                    }
                    f = cadd(f, dpp_row<0x111>(f));
                    f = cadd(f, dpp_row<0x112>(f));
                    f = cadd(f, dpp_row<0x114>(f));
                    f = cadd(f, dpp_row<0x118>(f));     // lane 15 of every row: the row's sum
                    const double tot = swap16_add(f.re, f.im);   // even rows: re(row) + re(row+1); odd rows: im(row-1) + im(row)
                    if (nb == 15) {
                        // (the lane's share of the address is put together here, from the lane number read on the spot: as a loop
                        // invariant it -- or the thread id it comes from -- is spilled in the forms that are short of registers, and the
                        // reload waits with vmcnt(0): a drain of the prefetch ring)
                        int t2;
                        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t2));
                        t2 |= wave_s << 6;
                        double *dst = (double *) (((t2 >> 8) ? GuL(bcur) : F0L(bcur)) + sg * 8 + ((t2 >> 5) & 7));
                        dst[(t2 >> 4) & 1] = tot;
                    }
                    }
                    if constexpr (DUAL) {
                        // the tail outputs (with eight waves the threads 256 .. 511 have them)
                        const double xr = decode<DT>(e_re2, swap) - e_off_re, xi = decode<DT>(e_im2, swap) - e_off_im;
                        cd f = mk(fma(e_A2.re, xr, e_B2.re * xi), fma(e_A2.im, xr, e_B2.im * xi));
                        f = cadd(f, dpp_row<0x111>(f));
                        f = cadd(f, dpp_row<0x112>(f));
                        f = cadd(f, dpp_row<0x114>(f));
                        f = cadd(f, dpp_row<0x118>(f));
                        const double tot = swap16_add(f.re, f.im);
                        if (nb == 15) {
                            double *dst = (double *) (GuL(bcur) + e_o16);
                            dst[g & 1] = tot;
                        }
                    }
                }
            };

            // ---- K(it) with the pieces in between ----------------------------------------------------------------------------
            const bool more = it + 1 < cnt;
            const unsigned char *next_raw = (const unsigned char *) a.raw + (size_t) chunk_of(more ? it + 1 : it) * chunk_bytes;
            static_for<0, TPW / TP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (SHQ && IQF && j == 1) {
                    // Q of tile `wave` of item it comes next: its offset at the tile start chains the totals of the tiles before it,
                    // which the other waves left at the end of their first pass -- the one place where SHQ needs the workgroup
                    // to meet in mid-iteration
                    __syncthreads();
                    const cd *ot = OtL(bcur);
                    cd ts = mk(0.0, 0.0);
                    for (int t2 = 0; t2 < wave_s; ++t2) ts = mk(fma(iq_m16, ts.re, ot[t2].re), fma(iq_m16, ts.im, ot[t2].im));
                    gq[0] = mk(fma(iq_mnb, ts.re, oex[0].re), fma(iq_mnb, ts.im, oex[0].im));
                }
                if constexpr (DUAL) {
                    // two tiles (slots 0 and 1) through the matrix phase together: a digit-plane fragment is fetched once and multiplies
                    // the slices of both; the pieces of both tiles follow every group
                    static_assert(NSL == 2, "16-bit records");
                    v4i acc[2][3][S];
                    v4i F[S];
                    auto frag = [&](auto ic, auto dc) -> v4i {
                        constexpr int i = decltype(ic)::value, d = decltype(dc)::value, s = i / 3, rt = i % 3;
                        if constexpr (rt < 2) return wmainL[((rt * S + d) * KS + s) * 64 + lane];
                        else return wlocL[(d * KS + s) * 64 + lane];
                    };
                    v4i blo[2], bhi[2];
                    if constexpr (HAVE_K) static_for<0, S>([&](auto dc) { F[decltype(dc)::value] = frag(std::integral_constant<int, 0>{}, dc); });
                    static_for<0, KS * 3>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, s = i / 3, rt = i % 3;
                        if constexpr (rt == 0 && s % 2 == 0) SDRT_STAMP(5 + s / 2);      // (diagnostic builds: k-steps 0, 2, 4, 6 -> slots 5 .. 8)
                        if constexpr (HAVE_K) {
                            if constexpr (rt == 0) {
                                const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
                                static_for<0, 2>([&](auto tc) {
                                    constexpr int t = decltype(tc)::value;
                                    const uint32_t *cur = t == 0 ? raw[s % NSLOT] : raw2[s % NSLOT];
#pragma unroll
                                    for (int w = 0; w < 4; ++w) {
                                        blo[t][w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], sel_lo) ^ W8Enc<DT>::xlo);
                                        bhi[t][w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], sel_hi) ^ W8Enc<DT>::xhi);
                                    }
                                    if constexpr (s + PF < KS) issue(chunk_raw, std::integral_constant<int, t * KS + s + PF>{});
                                    else issue(next_raw, std::integral_constant<int, t * KS + s + PF - KS>{});
                                });
                            }
                            static_for<0, S>([&](auto dc) {
                                constexpr int d = decltype(dc)::value;
                                const v4i A = F[d];
                                static_for<0, 2>([&](auto tc) {
                                    constexpr int t = decltype(tc)::value;
                                    if constexpr (PACK && rt == 2) {
                                        if constexpr (d < 3) {
                                            if constexpr (s == 0) acc[t][2][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[t], (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                            else acc[t][2][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[t], acc[t][2][d], 0, 0, 0);
                                        } else if constexpr (d < NLF) {
                                            acc[t][2][d - 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo[t], acc[t][2][d - 3], 0, 0, 0);
                                        }
                                    } else {
                                        if constexpr (s == 0) acc[t][rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[t], (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                        else acc[t][rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi[t], acc[t][rt][d], 0, 0, 0);
                                        if constexpr (d >= 1) acc[t][rt][d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo[t], acc[t][rt][d - 1], 0, 0, 0);
                                    }
                                });
                                if constexpr (i + 1 < KS * 3) {
                                    if constexpr (!(PACK && (i + 1) % 3 == 2 && d >= NLF)) F[d] = frag(std::integral_constant<int, i + 1>{}, dc);
                                }
                            });
                        }
                        piece(std::integral_constant<int, 0>{}, ic);
                        piece(std::integral_constant<int, 1>{}, ic);
                        // (-DSDRT_DUAL_PACE: ask the scheduler for one matrix instruction, three vector instructions, ... over each group
                        // and its pieces -- measured slower, 4.62 against 4.34 ms: the pieces' operands are not there yet when their
                        // turn comes and the matrix instructions bunch up; profiles/r03_experiments.txt)
#ifdef SDRT_DUAL_PACE
                        if constexpr (HAVE_K) {
                            constexpr int NMG = 2 * ((PACK && rt == 2) ? NLF : 2 * S - 1);
                            static_for<0, NMG>([&](auto mc) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, SDRT_DUAL_VPM, 0);
                                if constexpr (decltype(mc)::value % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            });
                        }
                        __builtin_amdgcn_sched_barrier(0);
#endif
                    });
                    SDRT_STAMP(2);
                    if constexpr (HAVE_K) {
                        static_for<0, 2>([&](auto tc) {
                            constexpr int t = decltype(tc)::value;
                            double val[8];
                            static_for<0, 3>([&](auto rc) {
                                constexpr int rt = decltype(rc)::value;
                                if constexpr (PACK && rt == 2) {
                                    const double csl = g >= 2 ? cs_loc0 * 256.0 : cs_loc0;
#pragma unroll
                                    for (int v = 0; v < 4; ++v) {
                                        double dsum = (double) acc[t][2][2][v];
                                        dsum = fma(dsum, 65536.0, (double) acc[t][2][1][v]);
                                        dsum = fma(dsum, 65536.0, (double) acc[t][2][0][v]);
                                        dl[t][v] = dsum * csl;
                                    }
                                } else {
#pragma unroll
                                    for (int v = 0; v < 4; ++v) {
                                        double dsum = (double) acc[t][rt][S - 1][v];
#pragma unroll
                                        for (int q = S - 2; q >= 0; --q) dsum = fma(dsum, 256.0, (double) acc[t][rt][q][v]);
                                        if constexpr (rt < 2) val[4 * rt + v] = dsum * cs_main0; else dl[t][v] = dsum * cs_loc0;
                                    }
                                }
                            });
                            dC[t][0] = mk(val[0], val[1]);
                            dC[t][1] = mk(val[2], val[3]);
                            dA[t][0] = mk(val[4], val[5]);
                            dA[t][1] = mk(val[6], val[7]);
                        });
                    }
                } else if constexpr (RS) {
                    v4i accR[S], accI[S], accLR[NHF], accLI[NHF];
                    v4i A[S], L[NLF];
                    auto fragA = [&](auto sc, auto dc) -> v4i {
                        return wmainL[(decltype(dc)::value * KS + decltype(sc)::value) * 64 + lane];
                    };
                    auto fragL = [&](auto sc, auto fc) -> v4i {
                        return wlocL[(decltype(fc)::value * KS + decltype(sc)::value) * 64 + lane];
                    };
                    if constexpr (HAVE_K) {
                        static_for<0, S>([&](auto dc) { A[decltype(dc)::value] = fragA(std::integral_constant<int, 0>{}, dc); });
                    }
                    v4i br_lo, br_hi, bi_lo, bi_hi;
                    static_for<0, KS>([&](auto sc) {
                        constexpr int s = decltype(sc)::value, stg = j * KS + s;
                        if constexpr (HAVE_K) {
                            // slices of this k-step's 16 complex samples, de-interleaved: re bytes and im bytes
                            const uint32_t *cur = raw[stg % NSLOT];
                            if constexpr (ITEM == 2) {
                                const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
#pragma unroll
                                for (int w = 0; w < 4; ++w) {      // 4 samples -> one dword per (component, slice)
                                    const unsigned xl = __builtin_amdgcn_perm(cur[4 * w + 1], cur[4 * w], sel_lo);      // [re0 im0 re1 im1] low bytes
                                    const unsigned yl = __builtin_amdgcn_perm(cur[4 * w + 3], cur[4 * w + 2], sel_lo);
                                    const unsigned xh = __builtin_amdgcn_perm(cur[4 * w + 1], cur[4 * w], sel_hi);
                                    const unsigned yh = __builtin_amdgcn_perm(cur[4 * w + 3], cur[4 * w + 2], sel_hi);
                                    br_lo[w] = (int) (__builtin_amdgcn_perm(yl, xl, 0x06040200u) ^ W8Enc<DT>::xlo);
                                    bi_lo[w] = (int) (__builtin_amdgcn_perm(yl, xl, 0x07050301u) ^ W8Enc<DT>::xlo);
                                    br_hi[w] = (int) (__builtin_amdgcn_perm(yh, xh, 0x06040200u) ^ W8Enc<DT>::xhi);
                                    bi_hi[w] = (int) (__builtin_amdgcn_perm(yh, xh, 0x07050301u) ^ W8Enc<DT>::xhi);
                                }
                            } else {
#pragma unroll
                                for (int w = 0; w < 4; ++w) {      // a word holds [re0 im0 re1 im1]
                                    br_lo[w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], 0x06040200u) ^ W8Enc<DT>::xlo);
                                    bi_lo[w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], 0x07050301u) ^ W8Enc<DT>::xlo);
                                    br_hi[w] = 0;
                                    bi_hi[w] = 0;
                                }
                            }
                            if constexpr (stg + PF < NSTEP) issue(chunk_raw, std::integral_constant<int, stg + PF>{});
                            else issue(next_raw, std::integral_constant<int, stg + PF - NSTEP>{});
                            // group 0: aggregates of the real stream
                            __builtin_amdgcn_s_setprio(1);
                            static_for<0, S>([&](auto dc) {
                                constexpr int d = decltype(dc)::value;
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (NSL == 2) {
                                    if constexpr (s == 0) accR[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], br_hi, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                    else accR[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], br_hi, accR[d], 0, 0, 0);
                                    if constexpr (d >= 1) accR[d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], br_lo, accR[d - 1], 0, 0, 0);
                                } else {
                                    if constexpr (s == 0) accR[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], br_lo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                    else accR[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], br_lo, accR[d], 0, 0, 0);
                                }
                            });
                            __builtin_amdgcn_sched_barrier(0);
                            __builtin_amdgcn_s_setprio(0);
                        }
                        piece(jc, std::integral_constant<int, 6 * s + 0>{});
                        piece(jc, std::integral_constant<int, 6 * s + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (HAVE_K) {
                            // group 1: the same planes on the imaginary stream; each fragment register then takes the next k-step's
                            __builtin_amdgcn_s_setprio(1);
                            static_for<0, S>([&](auto dc) {
                                constexpr int d = decltype(dc)::value;
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (NSL == 2) {
                                    if constexpr (s == 0) accI[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], bi_hi, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                    else accI[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], bi_hi, accI[d], 0, 0, 0);
                                    if constexpr (d >= 1) accI[d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], bi_lo, accI[d - 1], 0, 0, 0);
                                } else {
                                    if constexpr (s == 0) accI[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], bi_lo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                    else accI[d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[d], bi_lo, accI[d], 0, 0, 0);
                                }
                                if constexpr (s + 1 < KS) A[d] = fragA(std::integral_constant<int, s + 1>{}, dc);
                            });
                            __builtin_amdgcn_sched_barrier(0);
                            __builtin_amdgcn_s_setprio(0);
                        }
                        piece(jc, std::integral_constant<int, 6 * s + 2>{});
                        piece(jc, std::integral_constant<int, 6 * s + 3>{});
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (HAVE_K) {
                            // group 2: the local tile, both streams (its fragments are fetched only now: registers are scarce and
                            // the other wave of the SIMD covers the LDS latency)
                            static_for<0, NLF>([&](auto fc) { L[decltype(fc)::value] = fragL(sc, fc); });
                            __builtin_amdgcn_sched_barrier(0);
                            __builtin_amdgcn_s_setprio(1);
                            static_for<0, NLF>([&](auto fc) {
                                constexpr int f = decltype(fc)::value;
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (f < NHF) {
                                    if constexpr (s == 0) {
                                        accLR[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[f], NSL == 2 ? br_hi : br_lo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                        accLI[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[f], NSL == 2 ? bi_hi : bi_lo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                    } else {
                                        accLR[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[f], NSL == 2 ? br_hi : br_lo, accLR[f], 0, 0, 0);
                                        accLI[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[f], NSL == 2 ? bi_hi : bi_lo, accLI[f], 0, 0, 0);
                                    }
                                } else {
                                    accLR[f - NHF] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[f], br_lo, accLR[f - NHF], 0, 0, 0);
                                    accLI[f - NHF] = __builtin_amdgcn_mfma_i32_16x16x64_i8(L[f], bi_lo, accLI[f - NHF], 0, 0, 0);
                                }
                            });
                            __builtin_amdgcn_sched_barrier(0);
                            __builtin_amdgcn_s_setprio(0);
                        }
                        piece(jc, std::integral_constant<int, 6 * s + 4>{});
                        piece(jc, std::integral_constant<int, 6 * s + 5>{});
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    if constexpr (j == 0) SDRT_STAMP(2);
                    if constexpr (HAVE_K) {
                        // int32 levels -> fp64.  Aggregates: lane (g, nb) = mode g of column nb, rows v = 2*kind + part, per stream.
                        double vr[4], vi[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            double sr_ = (double) accR[S - 1][v], si_ = (double) accI[S - 1][v];
#pragma unroll
                            for (int q = S - 2; q >= 0; --q) {
                                sr_ = fma(sr_, 256.0, (double) accR[q][v]);
                                si_ = fma(si_, 256.0, (double) accI[q][v]);
                            }
                            vr[v] = sr_ * cs_main0;
                            vi[v] = si_ * cs_main0;
                        }
                        dC[j][0] = mk(vr[0], vr[1]);
                        dA[j][0] = mk(vr[2], vr[3]);
                        dC[j][1] = mk(vi[0], vi[1]);
                        dA[j][1] = mk(vi[2], vi[3]);
                        // local tile: this lane holds level group rg of outputs (NOUT == 4: b = v; 8: b = 4 (g & 1) + v)
                        const int rg = PPF == 4 ? g : (g >> 1);
                        const double csl = cs_loc0 * (rg == 0 ? 1.0 : (rg == 1 ? 256.0 : (rg == 2 ? 65536.0 : 16777216.0)));
                        constexpr double STEP = PPF == 4 ? 4294967296.0 : 65536.0;      // 256^PPF between accumulators
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            double sr_ = (double) accLR[NHF - 1][v], si_ = (double) accLI[NHF - 1][v];
#pragma unroll
                            for (int q = NHF - 2; q >= 0; --q) {
                                sr_ = fma(sr_, STEP, (double) accLR[q][v]);
                                si_ = fma(si_, STEP, (double) accLI[q][v]);
                            }
                            dl[j][v] = sr_ * csl;
                            dli[j][v] = si_ * csl;
                        }
                    }
                } else {
                    v4i acc[3][S];
                    v4i accO[2];                         // IQ: offset rows x (high, low) slice
                    v4i F[S];                            // digit-plane fragments of the current group; each is reloaded for the next
                                                         // group as soon as its last matrix instruction has been issued
                    auto frag = [&](auto ic, auto dc) -> v4i {
                        constexpr int i = decltype(ic)::value, d = decltype(dc)::value, s = i / 3, rt = i % 3;
                        if constexpr (rt < 2) return wmainL[((rt * S + d) * KS + s) * 64 + lane];
                        else return wlocL[(d * KS + s) * 64 + lane];
                    };
                    // float32 (registers are short): a rolling window of three fragments over the flattened (group, digit) sequence
                    constexpr int FWN = 3;
                    v4i Fw[F32 ? FWN : 1];
                    auto frag_flat = [&](auto fc) -> v4i {
                        constexpr int f = decltype(fc)::value;
                        return frag(std::integral_constant<int, f / S>{}, std::integral_constant<int, f % S>{});
                    };
                    v4i blo, bhi;
                    v4i bsl[F32 ? 4 : 1];                // float32: the four slices of the k-step
                    float mxf = 0.f;                     // float32: largest magnitude of the tile, per lane
                    // the matrix phase of the tile; pcs: with the pieces of the other phases between its groups (the float32
                    // second pass runs without them)
                    auto kpass = [&](auto pcs) {
                        if constexpr (HAVE_K && F32) static_for<0, FWN>([&](auto fc) { Fw[decltype(fc)::value] = frag_flat(fc); });
                        if constexpr (HAVE_K && !F32) static_for<0, S>([&](auto dc) { F[decltype(dc)::value] = frag(std::integral_constant<int, 0>{}, dc); });
                        static_for<0, KS * 3>([&](auto ic) {
                            constexpr int i = decltype(ic)::value, s = i / 3, rt = i % 3, stg = j * KS + s;
                            if constexpr (HAVE_K) {
                                if constexpr (rt == 0) {
                                    // slices of this k-step's 8 complex samples; then its raw registers take the k-step PF ahead
                                    const uint32_t *cur = raw[stg % NSLOT];
                                    if constexpr (PRE) {
                                        // sliced already: dword w of slice sl sits at word 4 sl + w of the lane's k-step
#pragma unroll
                                        for (int sl_ = 0; sl_ < 4; ++sl_)
#pragma unroll
                                            for (int w = 0; w < 4; ++w) bsl[sl_][w] = (int) cur[4 * sl_ + w];
                                    } else if constexpr (F32) {
                                        // 16 values (8 samples) -> fixed point against the tile's exponent -> 4 x 4 byte transposes
                                        const int sh = 30 - e_cur[j];
                                        uint32_t *curw = raw[stg % NSLOT];
                                        if (swap) {
#pragma unroll
                                            for (int w = 0; w < 16; ++w) curw[w] = __builtin_bswap32(curw[w]);
                                        }
                                        // the largest magnitude so far (before anything else, so that nothing has to be kept for it)
#pragma unroll
                                        for (int w = 0; w < 16; w += 2)
                                            asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mxf) : "v"(curw[w]), "v"(curw[w + 1]));
                                        __builtin_amdgcn_sched_barrier(0);
                                        f32_slices(cur, sh, bsl);
                                    } else if constexpr (ITEM == 2) {
                                        const unsigned sel_lo = swap ? 0x07050301u : 0x06040200u, sel_hi = swap ? 0x06040200u : 0x07050301u;
    #pragma unroll
                                        for (int w = 0; w < 4; ++w) {
                                            blo[w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], sel_lo) ^ W8Enc<DT>::xlo);
                                            bhi[w] = (int) (__builtin_amdgcn_perm(cur[2 * w + 1], cur[2 * w], sel_hi) ^ W8Enc<DT>::xhi);
                                        }
                                    } else {
    #pragma unroll
                                        for (int w = 0; w < 4; ++w) { blo[w] = (int) (cur[w] ^ W8Enc<DT>::xlo); bhi[w] = 0; }
                                    }
                                    if constexpr (stg + PF < NSTEP) issue(chunk_raw, std::integral_constant<int, stg + PF>{});
                                    // the next item's first k-steps follow in the same ring (in the last iteration: this item's again, unused)
                                    else issue(next_raw, std::integral_constant<int, stg + PF - NSTEP>{});
                                }
                                SDRT_PRIO_HI();           // matrix instructions first: the other wave's vector work fills the gaps
                                static_for<0, S>([&](auto dc) {
                                    constexpr int d = decltype(dc)::value;
                                    __builtin_amdgcn_sched_barrier(0);
                                    v4i A;
                                    if constexpr (F32) A = Fw[(i * S + d) % FWN]; else A = F[d];
                                    if constexpr (F32 && PACK && rt == 2) {
                                        // packed local tile, fragments (0|1) (2|3) (4|-) (1|2) (3|4): accumulator a holds levels (2a+3 | 2a+4);
                                        // the first product into every accumulator is the one with the top slice
                                        auto mm = [&](auto ac, auto slc, auto first) {
                                            constexpr int a_ = decltype(ac)::value, sl_ = decltype(slc)::value;
                                            if constexpr (decltype(first)::value && s == 0) acc[2][a_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bsl[sl_], (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                            else acc[2][a_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bsl[sl_], acc[2][a_], 0, 0, 0);
                                        };
                                        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
                                        if constexpr (d == 0) mm(I0{}, I3{}, std::true_type{});
                                        if constexpr (d == 1) { mm(I1{}, I3{}, std::true_type{}); mm(I0{}, I1{}, std::false_type{}); }
                                        if constexpr (d == 2) { mm(I2{}, I3{}, std::true_type{}); mm(I1{}, I1{}, std::false_type{}); }
                                        if constexpr (d == 3) mm(I0{}, I2{}, std::false_type{});
                                        if constexpr (d == 4) { mm(I1{}, I2{}, std::false_type{}); mm(I0{}, I0{}, std::false_type{}); }
                                    } else if constexpr (F32) {
                                        static_for<0, 4>([&](auto slc) {
                                            constexpr int sl_ = 3 - decltype(slc)::value;      // top slice first: it opens its accumulator
                                            if constexpr (d + sl_ >= LMIN) {
                                                if constexpr (sl_ == 3 && s == 0) acc[rt][d + sl_ - LMIN] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bsl[sl_], (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                                else acc[rt][d + sl_ - LMIN] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bsl[sl_], acc[rt][d + sl_ - LMIN], 0, 0, 0);
                                            }
                                        });
                                    } else if constexpr (PACK && rt == 2) {
                                        // packed local tile: fragment d < 3 with the high (or only) slice starts accumulator d,
                                        // fragment 3 + a with the low slice adds to accumulator a
                                        if constexpr (d < 3) {
                                            if constexpr (s == 0 && d == 0 && AOPEN) acc[2][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi, aopenL[8 + g], 0, 0, 0);
                                            else if constexpr (s == 0) acc[2][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, NSL == 2 ? bhi : blo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                            else acc[2][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, NSL == 2 ? bhi : blo, acc[2][d], 0, 0, 0);
                                        } else if constexpr (d < NLF) {
                                            acc[2][d - 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, acc[2][d - 3], 0, 0, 0);
                                        }
                                    } else if constexpr (NSL == 2) {
                                        // digit d x high slice -> level d+1 (accumulator d); digit d x low slice -> level d (accumulator d-1)
                                        if constexpr (s == 0 && d == 0 && AOPEN) acc[rt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi, aopenL[4 * rt + g], 0, 0, 0);
                                        else if constexpr (s == 0) acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                        else acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi, acc[rt][d], 0, 0, 0);
                                        if constexpr (d >= 1) acc[rt][d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, acc[rt][d - 1], 0, 0, 0);
                                    } else {
                                        if constexpr (s == 0) acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                        else acc[rt][d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, acc[rt][d], 0, 0, 0);
                                    }
                                    if constexpr (F32) {
                                        if constexpr (i * S + d + FWN < KS * 3 * S) Fw[(i * S + d) % FWN] = frag_flat(std::integral_constant<int, i * S + d + FWN>{});
                                    } else
                                    // the fragment's register takes the next group's fragment (none for the unused slots of a packed tile)
                                    if constexpr (i + 1 < KS * 3) {
                                        if constexpr (!(PACK && (i + 1) % 3 == 2 && d >= NLF)) F[d] = frag(std::integral_constant<int, i + 1>{}, dc);
                                    }
                                });
                                if constexpr (IQF && rt == 2) {
                                    // the packed offset rows of this k-step against both slices
                                    const v4i Fq = wiqL[s * 64 + lane];
                                    if constexpr (NSL == 2) {
                                        if constexpr (s == 0) accO[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fq, bhi, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                        else accO[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fq, bhi, accO[0], 0, 0, 0);
                                    }
                                    if constexpr (s == 0) accO[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fq, blo, (v4i) {0, 0, 0, 0}, 0, 0, 0);
                                    else accO[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Fq, blo, accO[1], 0, 0, 0);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                SDRT_PRIO_LO();
                            }
                            if constexpr (decltype(pcs)::value) piece(jc, ic);
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    };
                    kpass(std::true_type{});
                    if constexpr (SHQ) {
                        // the queue of column-local outputs moves on by one pass
#pragma unroll
                        for (int e = 0; e < OPL; ++e) {
#pragma unroll
                            for (int d = 0; d < 4; ++d) yqs[d][e] = yqs[d + 1][e];
                            yqs[4][e] = ynew[j ^ 1][e];
                        }
                    }
                    int e_true = 0;
                    if constexpr (F32 && HAVE_K && !PRE) {
                        // the exponent in use is a function of the tile's own samples (its exponent rounded up to a multiple of
                        // four), never of the guess: an item's bits do not depend on what the workgroup saw before it
                        e_true = f32_tile_exponent((unsigned) __builtin_amdgcn_readfirstlane((int) wave_max(__float_as_uint(mxf))));
                        if (e_true != e_cur[j]) {
                            e_cur[j] = e_true;
                            static_for<0, PF>([&](auto pc) { issue(chunk_raw, std::integral_constant<int, j * KS + decltype(pc)::value>{}); });
                            kpass(std::false_type{});
                        }
                    }
                    if constexpr (j == 0) SDRT_STAMP(2);
                    if constexpr (HAVE_K) {
                        // int32 levels -> fp64: value = cs * sum_q 256^q acc[q]; lane (g, nb): rows 4g+v of column nb (v = 2*stream + part)
                        double val[8];
                        // float32: everything of this tile in units of 2^(E - 30)
                        const double esc = F32 ? __hiloint2double((1023 + e_cur[j] - 30) << 20, 0) : 1.0;
                        const double cs_main = cs_main0 * esc, cs_loc = cs_loc0 * esc;
                        if constexpr (F32) ccon[j] = 8421504.0 * esc;      // (e_cur[j] stays: the next item's guess)
                        static_for<0, 3>([&](auto rc) {
                            constexpr int rt = decltype(rc)::value;
                            if constexpr (PACK && rt == 2) {
                                // accumulator a: rows 0-7 level 2a (+1), rows 8-15 one level up: lanes g >= 2 scale by 256
                                const double csl = g >= 2 ? cs_loc * 256.0 : cs_loc;
    #pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    double dsum = (double) acc[2][2][v];
                                    dsum = fma(dsum, 65536.0, (double) acc[2][1][v]);
                                    dsum = fma(dsum, 65536.0, (double) acc[2][0][v]);
                                    dl[j][v] = dsum * csl;
                                }
                            } else {
    #pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    double dsum = (double) acc[rt][S - 1][v];
    #pragma unroll
                                    for (int q = S - 2; q >= 0; --q) dsum = fma(dsum, 256.0, (double) acc[rt][q][v]);
                                    if constexpr (rt < 2) val[4 * rt + v] = dsum * cs_main; else dl[j][v] = dsum * cs_loc;
                                }
                            }
                        });
                        dC[j][0] = mk(val[0], val[1]);
                        dC[j][1] = mk(val[2], val[3]);
                        dA[j][0] = mk(val[4], val[5]);
                        dA[j][1] = mk(val[6], val[7]);
                        if constexpr (IQF) {
                            // what the column adds to the offset: lane (g, nb) holds digit planes 2g, 2g+1 of both sums (rows 4g + v:
                            // plane 2g + (v >> 1), component v & 1; g == 3: empty rows)
                            double tq[4];
#pragma unroll
                            for (int v = 0; v < 4; ++v) tq[v] = NSL == 2 ? fma((double) accO[0][v], 256.0, (double) accO[1][v]) : (double) accO[1][v];
                            const double sc = iq_cs * (g == 0 ? 1.0 : (g == 1 ? 65536.0 : 4294967296.0));
                            double ar = fma(tq[2], 256.0, tq[0]) * sc, ai = fma(tq[3], 256.0, tq[1]) * sc;
                            ar = swap16_add(ar, ar);                                  // all-reduce over the four lanes 16 apart
                            ai = swap16_add(ai, ai);
                            ar = swap32_add(ar, ar);
                            ai = swap32_add(ai, ai);
                            // + the constant the slices leave out; then the inclusive scan over the tile's 16 columns
                            cd o = mk(fma(a.c_re, iq_wsum, ar), fma(a.c_im, iq_wsum, ai));
                            o = mk(fma(iq_m1, dpp_row<0x111>(o.re), o.re), fma(iq_m1, dpp_row<0x111>(o.im), o.im));
                            o = mk(fma(iq_m2, dpp_row<0x112>(o.re), o.re), fma(iq_m2, dpp_row<0x112>(o.im), o.im));
                            o = mk(fma(iq_m4, dpp_row<0x114>(o.re), o.re), fma(iq_m4, dpp_row<0x114>(o.im), o.im));
                            o = mk(fma(iq_m8, dpp_row<0x118>(o.re), o.re), fma(iq_m8, dpp_row<0x118>(o.im), o.im));
                            if (lane == 15) OtL(bcur)[(OPQ ? wave_s : wave) + WV * j] = o;              // the tile's total
                            oex[j] = dpp_row<0x111>(o);                               // offset at the column's first sample
                        }
                    }
                }
            });
            // the queue of column-local outputs moves on
            if constexpr (!SHQ) {
#pragma unroll
                for (int j = 0; j < TPW; ++j)
#pragma unroll
                    for (int e = 0; e < OPL; ++e) { yq[0][j][e] = yq[1][j][e]; yq[1][j][e] = ynew[j][e]; }
            }
            SDRT_STAMP(3);
            __syncthreads();
            SDRT_STAMP(4);
#undef SDRT_STAMP
            bcur = bcur == NBUF - 1 ? 0 : bcur + 1;
        };
        // diagnostic: core-clock and 100 MHz real-time stamps around the steady-state iterations -> in-kernel clock
#ifdef SDRT_STAMPS
        unsigned long long *clk = (a.stamps && (int) bid * 2 < a.stamp_wgs && lane == 0) ? a.stamps + ((size_t) bid * 8 + wave) * 16 : nullptr;
#else
        unsigned long long *clk = nullptr;
#endif
        if (clk) { clk[10] = __builtin_amdgcn_s_memtime(); clk[11] = __builtin_amdgcn_s_memrealtime(); }
        // The steady state as a copy of its own (no branch on which phases have an item) where the register allocator copes with a
        // second loop -- the forms with four outputs per column, eight tiles per chunk and no IQ correction, the headline one among
        // them: no spill and no vmcnt(0) in either loop (tools/loop_scratch.sh).  The others (eight outputs per column, IQ
        // correction, float32, sixteen tiles) answer a second loop with 3 to 4 times the spills, reloads inside the loop included:
        // they keep the one general loop.
        constexpr bool SPLIT = NOUT == 4 && NTC == 8 && IQ == 0 && !F32;
        if constexpr (SPLIT) {
            for (int it = 0; it < cnt && it < 3; ++it) iteration(std::true_type{}, std::false_type{}, it);
            for (int it = 3; it < cnt; ++it) iteration(std::true_type{}, std::true_type{}, it);
        } else {
            for (int it = 0; it < cnt; ++it) iteration(std::true_type{}, std::false_type{}, it);
        }
        if (clk) { clk[12] = __builtin_amdgcn_s_memtime(); clk[13] = __builtin_amdgcn_s_memrealtime(); clk[14] = (unsigned long long) cnt; }
        for (int it = cnt; it < cnt + 3; ++it) iteration(std::false_type{}, std::false_type{}, it);
    }
}

#ifdef SDRT_I8W_PROBE
// tools/cc_i8w.sh: one instantiation alone (the headline one unless SDRT_I8W_PROBE_ARGS says otherwise), for a quick look at its code
#ifndef SDRT_I8W_PROBE_ARGS
#define SDRT_I8W_PROBE_ARGS SDRT_S16, 4, 8, SDRT_I8_DIGITS, 0, false
#endif
template __global__ void chain_i8w_kernel<SDRT_I8W_PROBE_ARGS>(DecimArgs);
#else
// The instantiations are spread over four translation units that compile in parallel: this file holds the signed 16-bit
// encoding and the dispatch; kernels_i8w_b.hip / _c.hip / _d.hip include it with SDRT_I8W_BYTES / SDRT_I8W_U16 / SDRT_I8W_F32
// defined for the 8-bit encodings, the unsigned 16-bit one and float32.
hipError_t launch_chain_i8w_bytes(const DecimArgs &a, int nout, int nt, int dt, dim3 grid, size_t lds, hipStream_t s);
hipError_t launch_chain_i8w_u16(const DecimArgs &a, int nout, int nt, dim3 grid, size_t lds, hipStream_t s);
hipError_t launch_chain_i8w_f32(const DecimArgs &a, int nout, int nt, dim3 grid, size_t lds, hipStream_t s);

template <int DT, int NOUT, int NTC, int IQ, bool RS, bool PRE = false, bool DUAL = false>
static hipError_t launch_i8w_q(const DecimArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    // once per device and instantiation (a driver call per launch costs more than the gap between two kernels)
    static thread_local int done_dev = -1;
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute((const void *) chain_i8w_kernel<DT, NOUT, NTC, SDRT_I8_DIGITS, IQ, RS, PRE, DUAL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    hipLaunchKernelGGL((chain_i8w_kernel<DT, NOUT, NTC, SDRT_I8_DIGITS, IQ, RS, PRE, DUAL>), grid, dim3(DUAL ? 256 : 512), lds, s, a);
    return hipGetLastError();
}
template <int DT, int NOUT, int NT>
static hipError_t launch_i8w_k(const DecimArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    // four tiles per chunk (two chunks per item): the plain form only
    if constexpr (NT == 4) {
        if (a.rs || a.w8iq || a.iq_group_off) return hipErrorInvalidValue;
        if constexpr (DT == SDRT_F32) {
            if (a.f32_exp) return launch_i8w_q<DT, NOUT, 4, 0, false, true>(a, grid, lds, s);
        }
        return launch_i8w_q<DT, NOUT, 4, 0, false>(a, grid, lds, s);
    } else {
        // no shift, no IQ correction: real streams (instantiated for the 16-bit encodings at 8 tiles per chunk only)
        if constexpr (Enc<DT>::size == 2 && NT == 8) {
            if (a.rs) return launch_i8w_q<DT, NOUT, NT, 0, true>(a, grid, lds, s);
        } else if (a.rs) return hipErrorInvalidValue;
        if (a.w8iq) return launch_i8w_q<DT, NOUT, NT, 2, false>(a, grid, lds, s);
        // one wave per SIMD on two tiles at once (signed 16-bit records, four outputs per column, no IQ correction)
        if constexpr (DT == SDRT_S16 && NT == 8 && NOUT == 4) {
            if (a.dual && !a.iq_group_off) return launch_i8w_q<DT, NOUT, NT, 0, false, false, true>(a, grid, lds, s);
        }
        return a.iq_group_off ? launch_i8w_q<DT, NOUT, NT, 1, false>(a, grid, lds, s) : launch_i8w_q<DT, NOUT, NT, 0, false>(a, grid, lds, s);
    }
}

template <int DT>
static hipError_t launch_i8w_d(const DecimArgs &a, int nout, int nt, dim3 grid, size_t lds, hipStream_t s) {
    if constexpr (DT == SDRT_F32 || DT == SDRT_S16 || DT == SDRT_U16) {
        if (nout == 4 && nt == 4) return launch_i8w_k<DT, 4, 4>(a, grid, lds, s);
        if (nout == 8 && nt == 4) return launch_i8w_k<DT, 8, 4>(a, grid, lds, s);
    }
    if constexpr (DT == SDRT_F32) return hipErrorInvalidValue;
    else {
    if (nout == 4 && nt == 8) return launch_i8w_k<DT, 4, 8>(a, grid, lds, s);
    if (nout == 4 && nt == 16) return launch_i8w_k<DT, 4, 16>(a, grid, lds, s);
    if (nout == 8 && nt == 8) return launch_i8w_k<DT, 8, 8>(a, grid, lds, s);
    if (nout == 8 && nt == 16) return launch_i8w_k<DT, 8, 16>(a, grid, lds, s);
    return hipErrorInvalidValue;
    }
}

#if defined(SDRT_I8W_BYTES)
hipError_t launch_chain_i8w_bytes(const DecimArgs &a, int nout, int nt, int dt, dim3 grid, size_t lds, hipStream_t s) {
    return dt == SDRT_S8 ? launch_i8w_d<SDRT_S8>(a, nout, nt, grid, lds, s) : launch_i8w_d<SDRT_U8>(a, nout, nt, grid, lds, s);
}
#elif defined(SDRT_I8W_U16)
hipError_t launch_chain_i8w_u16(const DecimArgs &a, int nout, int nt, dim3 grid, size_t lds, hipStream_t s) {
    return launch_i8w_d<SDRT_U16>(a, nout, nt, grid, lds, s);
}
#elif defined(SDRT_I8W_F32)
hipError_t launch_chain_i8w_f32(const DecimArgs &a, int nout, int nt, dim3 grid, size_t lds, hipStream_t s) {
    return launch_i8w_d<SDRT_F32>(a, nout, nt, grid, lds, s);
}

// Pre-slicing pass for many-row float32 contexts: one workgroup per tile of 4096 samples (32 KiB).  A lane's k-step of the chain
// kernel is an aligned 64-byte piece of the stream (eight complex samples); here every piece is replaced, in a buffer of the same
// shape, by its four slices against the tile's exponent -- computed as the chain kernel computes it (f32_tile_exponent of the
// largest magnitude, f32_slices).  Reads 8 B and writes 8 B per sample, once per chunk instead of once per (chunk, row).
__global__ __launch_bounds__(256) void f32_preslice_kernel(const uint32_t *__restrict__ raw, uint32_t *__restrict__ out, int *__restrict__ exps, int swap) {
    __shared__ unsigned wmax[4];
    const size_t tile = blockIdx.x;
    const int tid = threadIdx.x;
    const uint4 *src = (const uint4 *) (raw + tile * (size_t) (SDRT_TB * SDRT_WIDE_LC * 2));
    uint32_t w0[16], w1[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 p = src[(size_t) tid * 4 + i], r = src[(size_t) (tid + 256) * 4 + i];
        w0[4 * i] = p.x; w0[4 * i + 1] = p.y; w0[4 * i + 2] = p.z; w0[4 * i + 3] = p.w;
        w1[4 * i] = r.x; w1[4 * i + 1] = r.y; w1[4 * i + 2] = r.z; w1[4 * i + 3] = r.w;
    }
    if (swap) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { w0[i] = __builtin_bswap32(w0[i]); w1[i] = __builtin_bswap32(w1[i]); }
    }
    float mxf = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mxf) : "v"(w0[i]), "v"(w0[i + 1]));
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mxf) : "v"(w1[i]), "v"(w1[i + 1]));
    }
    unsigned v = __float_as_uint(mxf);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (unsigned) __shfl_xor((int) v, d));
    if ((tid & 63) == 0) wmax[tid >> 6] = v;
    __syncthreads();
    v = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    const int e = f32_tile_exponent(v);
    if (tid == 0) exps[tile] = e;
    v4i b0[4], b1[4];
    f32_slices(w0, 30 - e, b0);
    f32_slices(w1, 30 - e, b1);
    uint4 *dst = (uint4 *) (out + tile * (size_t) (SDRT_TB * SDRT_WIDE_LC * 2));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[(size_t) tid * 4 + i] = make_uint4((unsigned) b0[i][0], (unsigned) b0[i][1], (unsigned) b0[i][2], (unsigned) b0[i][3]);
        dst[(size_t) (tid + 256) * 4 + i] = make_uint4((unsigned) b1[i][0], (unsigned) b1[i][1], (unsigned) b1[i][2], (unsigned) b1[i][3]);
    }
}

hipError_t launch_f32_preslice(const void *raw, void *out, int *exps, size_t tiles, int swap, hipStream_t s) {
    hipLaunchKernelGGL(f32_preslice_kernel, dim3((unsigned) tiles), dim3(256), 0, s, (const uint32_t *) raw, (uint32_t *) out, exps, swap);
    return hipGetLastError();
}
#else
// permlane swap self-test: out[0][l] = swap32_add(x[l], y[l]), out[1][l] = swap16_add(x[l], y[l])
__global__ void swap_selftest_kernel(const double *x, const double *y, double *out) {
    const int l = threadIdx.x & 63;
    out[l] = swap32_add(x[l], y[l]);
    out[64 + l] = swap16_add(x[l], y[l]);
}

// ---- launch helpers -----------------------------------------------------------------------------------------------
bool chain_i8w_rs_supported(int dt, int n) {
    return (dt == SDRT_S16 || dt == SDRT_U16) && n == 8 * SDRT_TB * SDRT_WIDE_LC;
}

bool chain_i8w_supported(int q, int dt, int n, bool iq) {
    const int nt = n / (SDRT_TB * SDRT_WIDE_LC);
    if (n % (SDRT_TB * SDRT_WIDE_LC)) return false;
    // decimation 8 and 16 (round 4): 32 / 16 outputs per column, eight per pass -- 4 / 2 passes over the chunk (api.hip: WidePass);
    // plain integer forms (no IQ correction, no normalisation bias at decimation 4) at 8 or 16 tiles per chunk
    // ... and decimation 4 as 8 passes (measured against the per-sample kernel: chain 1.245 -> 0.765 ms per 2^26 samples); decimation 2
    // would be 16 passes and ties with it (0.78 / 0.80 ms per 2^25): it stays there unless SDRTERM_DIAG=wide_smallq=1 (diagnostics)
    const char *sq = sdrt_diag("wide_smallq");
    if (q == 4 && sq && atoi(sq) == 0) return false;
    if (q == 4 || q == 8 || q == 16 || (q == 2 && sq && atoi(sq) != 0)) return !iq && (nt == 8 || nt == 16) && (dt == SDRT_S8 || dt == SDRT_U8 || dt == SDRT_S16 || dt == SDRT_U16);
    if (!(q == 32 || q == 64)) return false;
    // chunks of four tiles go two to an item (float32 records; 16-bit records in half-size chunks): plain form only
    if (nt == 4) return !iq && (dt == SDRT_F32 || dt == SDRT_S16 || dt == SDRT_U16);
    if (!(dt == SDRT_S8 || dt == SDRT_U8 || dt == SDRT_S16 || dt == SDRT_U16)) return false;
    return nt == 8 || nt == 16;
}

size_t chain_i8w_lds_bytes(int q, int n, int iq, int dt) {
    const int ntc = n / (SDRT_TB * SDRT_WIDE_LC);
    const int nsl = dt == SDRT_S8 || dt == SDRT_U8 ? 1 : (dt == SDRT_F32 ? 4 : 2);
    const int nout = SDRT_WIDE_LC / q > SDRT_WIDE_MAXOUT ? SDRT_WIDE_MAXOUT : SDRT_WIDE_LC / q;
    return (size_t) wide_lds(SDRT_I8_DIGITS, nout, ntc < 8 ? 8 : ntc, iq, ntc, nsl).total ;
}

hipError_t launch_chain_i8w(const DecimArgs &a, int q, int dt, size_t n_chunks, int n_cus, hipStream_t s) {
    const int nout = SDRT_WIDE_LC / q > SDRT_WIDE_MAXOUT ? SDRT_WIDE_MAXOUT : SDRT_WIDE_LC / q, nt = a.n / (SDRT_TB * SDRT_WIDE_LC);
    // one workgroup per CU, in multiples of 8 (one per XCD); fewer when there is less work than CUs
    size_t items = n_chunks * (size_t) a.rows;
    size_t wgs = (size_t) (n_cus > 8 ? n_cus : 8);
    if (items < wgs) wgs = items;
    wgs = (wgs + 7) / 8 * 8;
    const dim3 grid((unsigned) wgs);
    const size_t lds = chain_i8w_lds_bytes(q, a.n, a.w8iq ? 2 : (a.iq_group_off ? 1 : 0), dt);
    switch (dt) {
        case SDRT_S8: case SDRT_U8: return launch_chain_i8w_bytes(a, nout, nt, dt, grid, lds, s);     // kernels_i8w_b.hip
        case SDRT_S16: return launch_i8w_d<SDRT_S16>(a, nout, nt, grid, lds, s);
        case SDRT_U16: return launch_chain_i8w_u16(a, nout, nt, grid, lds, s);        // kernels_i8w_c.hip
        case SDRT_F32: return launch_chain_i8w_f32(a, nout, nt, grid, lds, s);        // kernels_i8w_d.hip
    }
    return hipErrorInvalidValue;
}

hipError_t launch_swap_selftest(const double *x, const double *y, double *out, hipStream_t s) {
    hipLaunchKernelGGL(swap_selftest_kernel, dim3(1), dim3(64), 0, s, x, y, out);
    return hipGetLastError();
}
#endif
#endif  // SDRT_I8W_PROBE
