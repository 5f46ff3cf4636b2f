// Host-built operator tables shared between operators.cpp (extended-precision construction, host only)
// and the HIP translation unit (device kernels + C ABI).  Plain-old-data, copied to the device verbatim.
#pragma once
#include <stdint.h>

#define SDRT_MAXL 64        // samples per block (K dimension of the stage-A MFMA contraction)
#define SDRT_TB 16          // columns per tile (one DPP row of lanes); a column is G consecutive blocks
#define SDRT_GMAX 2         // blocks per column: 1 (f64 kernel), 1 or 2 (int8 kernel)
#define SDRT_EDGE 27        // sosfiltfilt odd-extension length for the order-8 decimator (3 * (2*4+1))
#define SDRT_NMODE 4        // conjugate pole pairs of the order-8 decimator
#define SDRT_TPOW 64        // tile-carry chains up to this many tiles are summed directly (powers tabulated), longer ones scanned

// Zero-phase decimator in modal block form (see DESIGN.md "Decimator").  All complex numbers are {re, im}.
struct DecimTables {
    int32_t L;              // block length in samples: min(q, 64)
    int32_t ksteps;         // L / 4 MFMA k-steps per block
    int32_t out_stride;     // one output every out_stride blocks: q / L
    int32_t edge;           // == SDRT_EDGE
    int32_t G;              // blocks per column (tile = 16 columns = 16 G blocks)
    int32_t reserved;
    // Stage-A weights, row r = mode (r & 3), kind (r >> 2): 0 Re causal, 1 Im causal, 2 Re anticausal, 3 Im anticausal.
    // causal row i:  alpha_i * p_i^(L-1-j);  anticausal row i: (alpha_i / p_i) * p_i^j.
    double W[16][SDRT_MAXL];
    double mu_pow[5][SDRT_NMODE][2];    // mu_i^(1,2,4,8,16), mu_i = p_i^(L G)   (column-to-column multiplier)
    double cF[SDRT_TB * SDRT_GMAX][SDRT_NMODE][2];  // p_i^(L nb)          carry-in of the causal state to block nb of a tile
    double cG[SDRT_TB * SDRT_GMAX][SDRT_NMODE][2];  // p_i^(L (16 G - nb)) carry-in of the anticausal state
    double tpow[SDRT_TPOW + 1][SDRT_NMODE][2];      // (p_i^(16 G L))^k: tile-to-tile multiplier powers
    double d0p;                          // direct term
    double head[SDRT_NMODE][SDRT_EDGE + 1][2];   // F_0,i = sum_l head[i][l] * u[l]            (per real stream)
    double tailF[SDRT_NMODE][SDRT_NMODE][2];     // G_end,i = sum_j tailF[i][j] F_j + tailFc[i][j] conj(F_j)
    double tailFc[SDRT_NMODE][SDRT_NMODE][2];    //          + sum_l tailU[i][l] * u[n-28+l]
    double tailU[SDRT_NMODE][SDRT_EDGE + 1][2];
    // per-sample form (chain_seq_kernel, decimation < 8): F_i(t+1) = p_i F_i(t) + ac_i u[t] (state before sample t),
    // G_i(t) = p_i G_i(t+1) + aa_i u[t] (inclusive), y[t] = d0p u[t] + 2 Re sum_i (F_i(t) + G_i(t))
    double pole[SDRT_NMODE][2], ac[SDRT_NMODE][2], aa[SDRT_NMODE][2];
};

// Tables of the wide-column int8 kernel (chain_i8w_kernel, kernels_i8w.hip).  A matrix-core column is LC = 256 consecutive
// samples holding nout = LC/q outputs ("boundaries" b = 0..nout-1 at sample b*q of the column); a tile is 16 columns.
// States are the alpha-scaled modal states of DecimTables; factors that only ever appear inside 2 Re(.) carry the 2.
#define SDRT_WIDE_LC 256
#define SDRT_WIDE_MAXOUT 8
#define SDRT_WIDE_MAXTILES 16
#define SDRT_WIDE_GV 60          // complex-sized entries per row of the IQ vectors: 48 pairs {g_re, g_im}, then 48 ints (accumulator openings)
struct WideTables {
    int32_t q, nout, boff, nout_total;                       // outputs boff .. boff + nout - 1 of the nout_total = LC / q of a column
    double mu_pow[4][SDRT_NMODE][2];                         // (p_i^LC)^(1,2,4,8): column-to-column scan inside a tile
    double tile_pow[SDRT_WIDE_MAXTILES + 1][SDRT_NMODE][2];  // (p_i^(16 LC))^k: tile-to-tile carries
    double bp2[2 * (SDRT_WIDE_MAXOUT + 1)][SDRT_NMODE][2];   // [b] 2 p_i^((boff + b) q): state at the column start -> output b of the
                                                             // pass; [nout + 1 + b] 2 p_i^(LC - (boff + b) q): state at the column end
    double cF2[SDRT_TB * SDRT_WIDE_MAXOUT][SDRT_NMODE][2];   // 2 p_i^(s): causal state at the tile start -> the output at sample s of the
                                                             // tile, s = column * LC + (boff + b) q for pos = column * nout + b
    double cG2[SDRT_TB * SDRT_WIDE_MAXOUT][SDRT_NMODE][2];   // 2 p_i^(16 LC - s): anticausal state at the tile end
    double head[SDRT_NMODE][SDRT_EDGE + 1][2];               // edge operators, as in DecimTables
    double tailF[SDRT_NMODE][SDRT_NMODE][2];
    double tailFc[SDRT_NMODE][SDRT_NMODE][2];
    double tailU[SDRT_NMODE][SDRT_EDGE + 1][2];
};

// One-pass form for decimation 2 .. 16 on the int8 matrix cores (chain_sq_kernel, kernels_sq.hip).  A column
// is 256 samples with all of its nout = 256 / q outputs; the part of output b (at sample s = b q of the column) that depends on the
// column's own samples is a convolution, row b = ONE master sequence h[j - s] e^(i theta (j - s)) shifted by s, so the local operator
// is 2 x S x ncp x 1 KB of digit bytes instead of nout x 2 x 512 bytes per digit; the column states are the modal states of the
// per-sample form of DecimTables (pole / ac / aa) at the column boundaries.
#define SDRT_SQ_LC 256
#define SDRT_SQ_MASTER 1024          // bytes per (part, digit, copy) of the master sequence: position 2 (d + 256) + component
// Where the copies sit in LDS, in 16-byte units: a lane reads one unit per matrix instruction, sixteen lanes (one row tile: 8 outputs x
// 2 parts) per LDS cycle -- their units must fall on sixteen different bank groups (unit index mod 16).  A lane's unit is
// part * PS + digit * ncp * CS + copy * CS + (k-step, lane group) - (s >> 3): the strides are padded so that copy and part move the
// residue by what (s >> 3) leaves free (unpadded, decimation 2 read eight ways conflicted).
#define SDRT_SQ_CS(q) (SDRT_SQ_MASTER / 16 + ((q) == 2 ? 2 : ((q) == 4 ? 4 : 0)))
#define SDRT_SQ_PS_BASE(q, S, ncp) ((S) * (ncp) * SDRT_SQ_CS(q))
#define SDRT_SQ_PS(q, S, ncp) (SDRT_SQ_PS_BASE(q, S, ncp) + ((((q) == 16 ? 1 : 8) - SDRT_SQ_PS_BASE(q, S, ncp) % 16 + 16) % 16))
#define SDRT_SQ_MASTER_BYTES(q, S, ncp) (2 * SDRT_SQ_PS(q, S, ncp) * 16)
struct SqTables {
    int32_t q, nout, ncp, nrt;                   // outputs per column; byte-alignment copies of the master (16 / (2 q), at least 1); row tiles of 16
    double mu[SDRT_NMODE][2];                    // p_i^256: column to column
    double head[SDRT_NMODE][SDRT_EDGE + 1][2];   // edge operators, as in DecimTables
    double tailF[SDRT_NMODE][SDRT_NMODE][2];
    double tailFc[SDRT_NMODE][SDRT_NMODE][2];
    double tailU[SDRT_NMODE][SDRT_EDGE + 1][2];
};

// Output lowpass (scipy sosfilt, zero initial state per chunk) in parallel modal form for a 64-lane (one wave) scan:
//   y[t] = D x[t] + sum_k Re(r_k s_k(t)),   s_k(t+1) = p_k s_k(t) + x[t],   s_k(0) = 0
// r_k already carries the weight 2 for a complex pole that stands for its conjugate pair too.
#define SDRT_OUT_MAXMODES 4
struct OutTables {
    int32_t nmodes;          // 0..SDRT_OUT_MAXMODES; -1: not representable -> kernel uses the sequential SOS form
    int32_t E;               // consecutive samples per lane: max(1, m / 64)
    int32_t ncomplex;        // modes [0, ncomplex) are complex poles (each standing for its conjugate too), the rest real
    int32_t reserved;
    double D;
    double r[SDRT_OUT_MAXMODES][2];
    double p[SDRT_OUT_MAXMODES][2];
    double pstep[SDRT_OUT_MAXMODES][6][2];   // p^(E * 2^i): wave scan multipliers
    double prow[SDRT_OUT_MAXMODES][16][2];   // p^(E * (j+1)), j < 16: from the end of the previous 16-lane row to lane j of a row
};

// post_fft8_kernel (kernels_post8.hip): the FM demodulation stage with LPI lanes of a wave per (chunk, row) item.
// N = m/2 pair angles per item, H = N/2 = 16 LPI points of the half-size complex transform: 16 per lane.
//   fft tables (separate device array, [4][16 LPI][2] + [8][LPI][2] doubles, laid out as the lanes read them):
//     tw1  [k1][l]      exp(-2 pi i l k1 / H)                       between the 16-point and the LPI-point forward steps
//     tw2  [ci][n2][l]  exp(+2 pi i n2 c / H), c = S_l[ci]          the same on the way back
//     alpha[ci][k2][l], beta[ci][k2][l]  for bin k = c + 16 k2:  Y[k] = alpha Z[k] + beta conj(Z[H - k]) turns the packed
//        spectrum of the angles into the packed spectrum of their half-sample shifted copy (scipy.signal.resample's odd outputs)
//     lane rows [8][l]  the output lowpass constants that depend on the lane (operators.cpp)
//   S_l: LPI = 8: {l, 16 - l} (lane 0: {0, 8});  LPI = 16: {l}
// Output lowpass (one complex pole pair + one real pole, dsp_processor.py:39-45): every lane holds 4 consecutive samples of
// each block of 4 LPI samples.
#define SDRT_P8_PPL 16
struct Post8Tables {
    int32_t N, LPI, reserved[2];
    // y[t] = D x[t] + Re(rc sc(t)) + rr sr(t);  sc(t+1) = pc sc(t) + x[t], sr(t+1) = pr sr(t) + x[t]  (OutTables' modes 0 and 1)
    double h[4];                            // a run of four samples from a zero state: y0[j] = sum_{i<=j} h[j-i] x[i]
    double pcp[3][2], prp[3];               // pc^1..3, pr^1..3: the run's end state x3 + p x2 + p^2 x1 + p^3 x0
    double g[4][2], gr[4];                  // rc pc^j, rr pr^j: what the state at the run's start adds to y[j]
    double pc4[2], pr4, pc8[2], pr8;        // p^4, p^8: scan inside a quad of lanes
    double pc32[2], pr32;                   // p^32 (with pc_q[3] = p^16: the scan over a row of sixteen lanes)
    double pc_q[4][2], pr_q[4];             // p^(4 (q+1)): the quads before -> lane q of a quad
    double pc_lane[16][2], pr_lane[16];     // p^(4 l):   block start -> first sample of lane l
    double pc_blk[2], pr_blk;               // p^(4 LPI): block to block
};

#ifdef __cplusplus
extern "C" {
#endif
// Tables of post_fft8_kernel for N pair angles per item (N = 32 LPI, LPI = 8 or 16).  fft_tab: [4][16 LPI][2] + [8][LPI][2].  Returns 0, or
// a negative code when the output lowpass is not (one complex pole pair + one real pole) or N does not fit.
int sdrt_build_post8_tables(const double *out_sos, int sections, int N, int LPI, struct Post8Tables *out, double *fft_tab);

// Modal form of an SOS cascade for m outputs per chunk.  Returns 0; out->nmodes == -1 when the cascade has
// repeated poles, FIR sections or more than SDRT_OUT_MAXMODES modes (the caller then uses the sequential form).
int sdrt_build_out_tables(const double *sos, int sections, int m, struct OutTables *out);

// Builds the tables from the decimator SOS (scipy layout [sections][6]).  Returns 0 or a negative code;
// on failure *err (if non-null) points to a static message.
int sdrt_build_decim_tables(const double *sos, int sections, int q, int G, struct DecimTables *out, const char **err);

// NCO tables for one row: tile[t] = exp(-2*pi*i*(f/fs)*t), t < tile_len; tilephase[T] = exp(-2*pi*i*(f/fs)*tile_len*T).
void sdrt_build_nco(double f, double fs, int tile_len, int n_tiles, double *tile /*[tile_len][2]*/,
                    double *tilephase /*[n_tiles][2]*/);

// Odd-sample kernel of scipy.signal.resample(a, 2N) for real a of length N: y[2j+1] = sum_k a[k] h[(j-k) mod N].
void sdrt_build_resample_odd(int N, double *h /*[N]*/);

// Tables of the FFT form of scipy.signal.resample(a, 2N) for real a of length N (power of two): tw[k] = exp(-2 pi i k/N),
// shift[f] = exp(+i pi fs/N)/N with fs the signed frequency of DFT bin f (0 at the Nyquist bin): the odd output samples are
// IDFT(DFT(a) * shift) without further scaling, the even ones are a itself.
void sdrt_build_fft_resample(int N, double *tw /*[N][2]*/, double *shift /*[N][2]*/);

// Digit planes for the 16x16x64 int8 matrix-core kernel (chain_i8t_kernel).  A column is LC = G*L samples.  Two row tiles
// of 16 rows: tile 0 = causal, tile 1 = anticausal aggregates of the column; row = 4*mode + 2*stream + part.  For G > 1 a
// third, compact tile holds per internal block boundary g and stream the column-local part of that boundary's output
// (row stream + 2*(g-1), NR = 2*(G-1) rows).  Layouts (MFMA A-operand order, KS = LC/32 k-steps):
//   out [(((tile*S + digit)*KS + kstep)*64 + lane)*16 + j],  lane = row + 16*kg
//   mid [((((digit*KS + kstep)*4 + kg)*NR) + row)*16 + j]
//   K position (kstep, kg, j) <-> sample kg*(LC/4) + 8*kstep + (j>>1), component j&1.
// One power-of-two scale for tiles 0/1 (*combine) and one for the mid tile (*mid_combine).  nco_samp: [LC][2] or NULL.
// `tab` must have been built with the same G.
int sdrt_build_i8_weights16(const double *sos, int sections, const struct DecimTables *tab, double data_scale,
                            const double *nco_samp, int S, signed char *out, double *combine, signed char *mid_out,
                            double *mid_combine);

// Wide-column kernel: tables for decimation q (32 <= q <= 256, power of two); data_scale multiplies the edge operators
// (the digit planes take it through sdrt_build_i8_weights_wide).
int sdrt_build_wide_tables(const double *sos, int sections, int q, double data_scale, struct WideTables *out, const char **err);
// Decimations below LC / SDRT_WIDE_MAXOUT (round 4): a column has nout_total = LC / q > 8 outputs, the kernel's local tile holds 8 --
// the chunk goes through the kernel nout_total / 8 times, pass p finishing outputs boff = 8 p .. 8 p + 7 of every column (the column
// aggregates are the same in every pass; tables and local planes are the pass's own).  nout_pass * k = nout_total, boff a multiple
// of nout_pass; nout_pass = LC / q, boff = 0 is the single pass above.
int sdrt_build_wide_tables_pass(const double *sos, int sections, int q, double data_scale, int nout_pass, int boff, struct WideTables *out,
                                const char **err);

// Digit planes of the wide-column kernel.  Column = SDRT_WIDE_LC samples, KS = 8 k-steps of 64; K position (kstep, kg, j)
// <-> sample 32*kstep + 8*kg + (j>>1), component j&1.  Row tiles 0/1 (causal / anticausal column aggregates, row =
// 4*mode + 2*stream + part) go to  main[(((tile*S + digit)*KS + kstep)*64 + row + 16*kg)*16 + j];  the local tile (row =
// 2*b + stream: the part of output b of the column that depends on this column's samples only, direct term included,
// NR = 2*nout rows) goes to  loc[((((digit*KS + kstep)*4 + kg)*NR) + row)*16 + j].  One power-of-two scale each.
// With IQ correction (iq_lambda > 0; read_file.py:64-77: z[i] -= off; off += z[i]*lambda) the correction inside a column is
// linear in the column's samples and in the offset at its first sample: z = (I - T) x - g off, T[i][j] = lambda
// (1-lambda)^(i-1-j) (j < i), g[i] = (1-lambda)^i.  The planes then hold W (I - T), and gvec[row] = {sum_t w_re[t] g[t],
// sum_t w_im[t] g[t]} (rows 0..31 main, 32..32+NR-1 local; w_re / w_im = the row's coefficients of the real / imaginary
// sample components) gives the rank-one rest: row value += (c - off).re * gvec[row][0] + (c - off).im * gvec[row][1], c =
// the constant the int8 slices leave out.  gvec may be NULL when iq_lambda == 0.  With iq_lambda > 0 the vectors are the sums of the
// QUANTISED weights, and 48 ints follow the 48 pairs: minus half the sum of every row's lowest digits (main rows 0-31, local rows
// from 32), the value the 16-bit forms open their lowest accumulators at -- an all-zero chunk at offset 0 then cancels exactly
// (operators.cpp: quantise_planes).
int sdrt_build_i8_weights_wide(const double *sos, int sections, int q, double data_scale, const double *nco_samp, int S,
                               double iq_lambda, signed char *main_out, double *main_combine, signed char *loc_out,
                               double *loc_combine, double *gvec /*[SDRT_WIDE_GV][2]*/);
// ... of one pass (see sdrt_build_wide_tables_pass): the local rows are outputs boff .. boff + nout_pass - 1 of the column
int sdrt_build_i8_weights_wide_pass(const double *sos, int sections, int q, double data_scale, const double *nco_samp, int S,
                                    double iq_lambda, int nout_pass, int boff, signed char *main_out, double *main_combine,
                                    signed char *loc_out, double *loc_combine, double *gvec /*[SDRT_WIDE_GV][2]*/);

// Offset rows of the wide-column kernel with IQ correction (iq_lambda > 0): per column the two sums  sum_j w_j x_re[j],
// sum_j w_j x_im[j], w_j = lambda (1-lambda)^(LC-1-j) -- what the column adds to the carried offset.  All S digit planes of
// both rows share ONE 16-row fragment per k-step (row 2 d + component):  out[(kstep*64 + row + 16*kg)*16 + j], K positions as
// for the other planes.  consts[55]: [0] sum_j w_j, [1..5] (1-lambda)^(LC 2^k) (k = 0..4), [6] (1-lambda)^-32,
// [7..38] lambda (1-lambda)^(31-l) (weights of the last 32 samples), [39..54] (1-lambda)^(LC nb), nb < 16.
#define SDRT_IQROW_CONSTS 55
int sdrt_build_i8_offset_rows(double lambda, int S, signed char *out /*[LC/32][64][16]*/, double *combine, double *consts);

// Digit planes of the wide-column kernel WITHOUT frequency shift ("real streams"): the real and the imaginary sample stream
// are then filtered by the same real-coefficient operator, so one 16-row tile (row = 4*mode + 2*kind + part, kind 0 causal /
// 1 anticausal column aggregate) serves both, applied to the re bytes and to the im bytes separately, 64 samples per
// k-step: KS = 4, K position (kstep, kg, j) <-> sample 64*kstep + 16*kg + j.
//   main[((digit*KS + kstep)*64 + row + 16*kg)*16 + j]
//   loc [((((digit*KS + kstep)*4 + kg)*nout) + b)*16 + j]      row b: the column-local part of output b (per stream)
int sdrt_build_i8_weights_wide_rs(const double *sos, int sections, int q, double data_scale, int S, signed char *main_out,
                                  double *main_combine, signed char *loc_out, double *loc_combine);

// Edge operators of the wide-column kernel as per-sample coefficients: output o16 = 8*kind + 2*mode + stream (kind 0: causal
// state F0 at the chunk start from samples 0..31, kind 1: the sample part Gu of the anticausal state at the chunk end from
// samples n-32..n-1) is  sum_l ab[o16][l][0..1] * x_re[l] + ab[o16][l][2..3] * x_im[l]  -  pq[o16][0..1] * off_re -
// pq[o16][2..3] * off_im, all complex {re, im}, x the true samples of the window, off the IQ offset at its first sample
// (lambda == 0: no correction, pq = 0).  NCO phasors as for the planes (NULL = no shift).
void sdrt_build_wide_edge(const struct WideTables *wt, const double *nco_samp, const double *nco_blk, int n, double iq_lambda,
                          double *ab /*[16][32][4]*/, double *pq /*[16][4]*/);

// Packed (real-input) form of the FM resampler's transform tables: see operators.cpp.  tw [N/2][2], ab [N/2][4].
void sdrt_build_fft_resample_packed(int N, double *tw, double *ab);
// chain_sq_kernel: `tab` is the context's DecimTables (data scale applied).  share: [256 / q][8][2] = 2 p_i^(b q) (i < 4), then
// 2 p_i^(256 - b q): what the states at the column's start / end add to output b (inside 2 Re).
int sdrt_build_sq_tables(const struct DecimTables *tab, int q, struct SqTables *out, double *share);
// One row's digit planes: agg [2 tiles][S][8 k-steps][64 lanes][16] (A-operand order; tile 0 causal, 1 anticausal column aggregates,
// row 4 mode + 2 stream + part, the within-column NCO phasor folded in) and master (SDRT_SQ_MASTER_BYTES; unit layout above; part 0 / 1:
// what the real / imaginary part of an output takes of (re, im) of the samples; copy c = the sequence delayed by 2 c q bytes).
// nco_samp: [256][2] exp(i theta j) or NULL.  Returns 0, or -1 when a plane set does not fit S digits.
int sdrt_build_sq_row(const struct DecimTables *tab, int q, const double *nco_samp, int S, signed char *agg, double *agg_combine,
                      signed char *master, double *loc_combine);

// IQ-correction geometric weights: pw[j] = (1-lambda)^j, j <= count.
void sdrt_build_iq_pow(double lambda, int count, double *pw);
#ifdef __cplusplus
}
#endif
