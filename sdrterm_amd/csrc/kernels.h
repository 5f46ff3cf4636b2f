// Kernel argument blocks and launch helpers (kernels.hip <-> api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/sdrterm_hip.h"
#include "tables.h"

#define SDRT_MAX_OUT_SECTIONS 4

// diagnostic switches: the value of `key` in SDRTERM_DIAG="key=value,key=value" or nullptr (api.hip); valid until the next call
const char *sdrt_diag(const char *key);
int sdrt_diag_int(const char *key, int dflt);

struct DecimArgs {
    const void *raw;            // [n_chunks][n][2] records of the input encoding
    const DecimTables *tab;     // device copy
    const OutTables *otab;      // device copy
    const double *nco_tile;     // [rows][L][2]     exp(i theta j), j < L            (sample within block)
    const double *nco_phase;    // [rows][n/L][2]   exp(i theta L B), B < n/L        (block within chunk)
    double *ydec;               // [row][chunk][m][2] decimated complex samples (input of the post kernel)
    unsigned long long *stamps; // optional diagnostic: [wg < stamp_wgs][wave][8] s_memtime at phase boundaries
    int stamp_wgs;
    int pair_out;               // wide-column kernel: ydec takes [row][chunk][m/2] products y[2i] conj(y[2i+1]) instead of the samples
    const int *row_pair;        //   per-row modes: [rows] 1 = this row's items leave as pair products (FM), 0 = as samples (AM); NULL: pair_out for all
    int pair_wide;              //   with row_pair: every item keeps the place the samples would take (products fill its first half)
    size_t n_chunks;            // chunks in the batch = row stride of ydec
    size_t n_count;             // wide-column kernel: chunks to process (0 = n_chunks); raw / ydec then point at the first of them
    int n, m, n_tiles, rows, swap;
    double norm_bias;           // added to Re before everything else (normalisation offset / scale; the scale itself is
                                // folded into the operator tables on the host)
    // IQ correction (read_file.py:64-77), only read when the IQ template flag is set:
    const double *iq_group_off; // [chunk][n/GS][2] offset at the start of every GS-sample group, relative to a zero offset
                                //                  at the start of its chunk (GS = L/4)
    const double *iq_chunk_off; // [chunk+1][2]     true offset at the start of every chunk
    const double *iq_pow;       // [n/GS + 1]       (1-lambda)^(GS*k)
    double iq_lambda;
    // int8-sliced stage A (kernels_i8t.hip, kernels_i8w.hip) only:
    const signed char *w8;      // [rows][S][LC/16][64][16] main digit planes in MFMA A-operand order (LC = G*L)
    const signed char *w8mid;   // [rows][S][LC/16][2][NR][16] mid-boundary digit planes (G > 1)
    double w8mid_combine;
    const double *ybias;        // [rows][m][2] decimator response to the constant the slices leave out (or NULL)
    double w8_combine;          // scale * 2^-(8S-1)
    double x_offset;            // what the slices leave out of every component: x = x' + x_offset
    const double *wsum;         // float32 (kernels_i8t.hip): [rows][32] sum over K of every weight row of the two row tiles
    const WideTables *wtab;     // wide-column kernel (kernels_i8w.hip) only; there w8mid holds the local-row planes
    const double *edge_ab;      //   [rows][16][32][4] edge operators as per-sample coefficients (sdrt_build_wide_edge)
    const double *edge_pq;      //   [rows][16][4]     their IQ-offset terms
    const double *gvec;         //   [rows][48][2]     IQ rank-one vectors (sdrt_build_i8_weights_wide)
    double e_off_re, e_off_im;  //   what the edge operators subtract from the decoded components
    double c_re, c_im;          //   IQ: the constant the slices leave out of (re, im)
    int iq_gs;                  //   IQ: samples per prefix group
    const signed char *w8iq;    //   IQ: packed offset rows [LC/32][64][16] (sdrt_build_i8_offset_rows); non-null selects the IQ form
    const double *iq_consts;    //   IQ: [SDRT_IQROW_CONSTS]
    double iq_combine;          //   IQ: combine factor of the offset rows
    double *iq_chunk_agg;       //   IQ out: [chunk][2] offset at the chunk's end when it starts from zero
    const void *raw_orig;       //   float32, pre-sliced input (f32_exp != NULL): the records themselves, for the edge operators
    const int *f32_exp;         //   float32, pre-sliced input: [chunk][4] tile exponents; raw then holds the slices (launch_f32_preslice)
    int sched;                  //   diagnostic builds only (-DSDRT_SCHED_EXP): wave priority assignment, SDRTERM_DIAG=sched
    int dual;                   //   chain_i8w_kernel: one wave per SIMD on two tiles at once (DUAL form)
    int out_mul, out_off;       //   eight outputs per column: output b of column c goes to index c * out_mul + out_off + b (one pass: 8, 0)
    int m_local;                //   ... and the pass's ybias holds m_local entries per row (one pass: m)
    int rs;                     //   real-stream form (no shift, no IQ correction): w8 / w8mid hold sdrt_build_i8_weights_wide_rs planes
    // second look of stage_a = auto (chain_kernel only): the (row, chunk) items listed by the demodulation stage are done again in
    // float64, each result written over the item's first try
    const int *fix_list;        //   [*fix_count] item = row * n_chunks + chunk (NULL: every item of the grid, the normal launch)
    const unsigned *fix_count;
    int fix_pairs;              //   the items' place in ydec holds [m/2] products y[2i] conj(y[2i+1]) (as pair_out of the wide kernel)
};

struct IqArgs {
    const void *raw;
    double *group_off;          // out: [chunk][n/GS][2]
    double *chunk_agg;          // out: [chunk][2]  offset at the end of the chunk when it starts from zero
    double *chunk_off;          // [chunk+1][2]     (scan kernel) true offsets at chunk starts; [n_chunks] = carried out
    double *state;              // [2] carried offset (in/out, scan kernel)
    size_t n_chunks;
    int n, gs, dt, swap;
    int wmask, wtail;           // groups the consumer reads: those with (group & wmask) == 0 and group wtail (wmask 0: all of them)
    const double *pow_tab;      // [n/GS + 1] (1-lambda)^(GS k)  (same table the chain kernel uses)
    double norm_bias, lambda, pow_g /* (1-lambda)^GS */, pow_n /* (1-lambda)^n */;
    const double *col_w;        // [259] column form (iq_cols_kernel): lambda (1-lambda)^(255-t), t < 256; (1-lambda)^-32; ^256; ^224 (or NULL)
    double *scan_part;          // [SDRT_IQ_SCAN_PARTS][4] scratch of the several-workgroup chunk scan (or NULL: one workgroup)
};
#define SDRT_IQ_SCAN_PARTS 512
hipError_t launch_iq_local(const IqArgs &a, hipStream_t s);
hipError_t launch_iq_scan(const IqArgs &a, hipStream_t s);
// y[row][chunk][k] -= off[chunk] * R[row][k]: the share of the offset at the chunk's start (wide-column kernel, api.hip)
hipError_t launch_iq_fix(double *ydec, const double *R, const double *chunk_off, int rows, size_t n_chunks, int m, hipStream_t s);

size_t chain_lds_bytes(int n, int L, int m);

// demodulate + FM resample + output lowpass of decimated samples: one wave per (row, chunk) item
struct PostArgs {
    const double *y;            // [items][m][2]
    const double *resample_odd; // [m/2] (FM)
    const double *fft_tw;       // [m/2][2] exp(-2 pi i k/N)          (FM, FFT form of the resampler; may be NULL)
    const double *fft_shift;    // [m/2][2] half-sample shift / N
    const OutTables *otab;
    const double *out_sos;      // [out_sections][6], used only by the sequential fallback
    double *out;                // [items][m]
    size_t items;
    int m, mode, out_sections;
    int fft_ok;                 // host check: modal lowpass available (OutTables.nmodes >= 0) and not disabled
    const double *fftp_tw, *fftp_ab;   // packed form (post_fftp_kernel: 2048 / 4096 outputs per chunk): [m/4][2] twiddles, [m/4][4] alpha / beta (or NULL)
    // --correct-iq computed inside the wide-column kernel: the share of the offset at each chunk's start is still to be
    // subtracted, y[row][chunk][k] -= iq_off[chunk] * iq_R[row][k] (NULL: nothing to subtract); see iq_fix_kernel
    const double *iq_R;         // [rows][m][2]
    const double *iq_off;       // [n_chunks][2]
    size_t n_chunks;            // items = rows * n_chunks, row-major
    const int *row_map;         // per-row demodulation modes: work item w = (k, chunk) stands for item (row_map[k], chunk) of the
                                // batch; items = (rows of this mode) * n_chunks.  NULL: every row, item w is item w
    int pairs;                  // FM: y holds [items][m/2] products y[2i] conj(y[2i+1]) (DecimArgs.pair_out)
    int pairs_wide;             //   ... each item in the place its m samples would take (DecimArgs.pair_wide: per-row modes)
    int am_in_lds;              // post_big_kernel, AM: the item's samples fit the LDS for the lowpass (set by launch_post_big)
    // eight / sixteen lanes per item (kernels_post8.hip; FM, m = 512 / 1024, lowpass = one complex pole pair + one real pole), or NULL
    const double *fft8_tab;     // [4][16 LPI][2] + [8][LPI][2]: sdrt_build_post8_tables
    const Post8Tables *p8tab;
    int p8_lpi;                 // lanes per item the tables were built for: 8 (m = 512) or 16 (m = 1024)
    double *stash;              // scratch of post_fft8_stash_bytes() bytes: the pair angles of the items in flight (or NULL: they wait in `out`)
    size_t stash_bytes;
    // stage_a = auto (post_fft8_kernel on pair products): an item of whose pair products half lie under cond_thr (in some lane's
    // evenly spread share; not exactly zero: silence is exact on every path) has its in-band content near the 40-bit operator image
    // of the int8 forms -- it is appended to cond_list for a second, float64 pass.  fix_list / fix_count: this launch IS that second pass: work item w stands
    // for item fix_list[w], w < *fix_count.
    int *cond_list;             // [items] (or NULL: no second look)
    unsigned *cond_count;       // [2]: items listed by this launch; [1] running total (diagnostics: sdrt_refined_items)
    double cond_thr;
    const int *fix_list;
    const unsigned *fix_count;
};
bool post_fft8_supported(const PostArgs &a);
size_t post_fft8_stash_bytes(const PostArgs &a);
hipError_t launch_post_fft8(const PostArgs &a, hipStream_t s);
// does launch_post pick a kernel that reads pair products (PostArgs.pairs)?
bool post_takes_pairs(const PostArgs &a);
// does launch_post pick a kernel that applies iq_R / iq_off while it loads the decimated samples?
bool post_takes_iq_share(const PostArgs &a);
struct SmoothArgs {
    double *out;                // [items][m], smoothed in place
    size_t items;
    int m, window, ntaps, tap_first, n_edge;
    const double *taps, *left, *right;
};
hipError_t launch_smooth(const SmoothArgs &a, hipStream_t s);
// FFT taps of the plots (kernels_post.hip)
struct TapArgs {
    const void *raw;            // [chunks][n][2] records
    double *u;                  // [chunks][n][2] decoded, normalised, shifted samples (overwritten by the spectrum transform)
    double *scratch;            // [chunks][n][2] second buffer of the spectrum transform
    double *out;                // spectrum: [chunks][n]; waterfall: [chunks][1024][n/hop + 1]
    const double *tw;           // exp(-2 pi i k/N): N = n (spectrum) or 1024 (waterfall)
    const double *nco_tile, *nco_phase;   // [64][2], [n/64][2] or NULL
    const double *window;       // [nperseg] (waterfall)
    size_t chunks;
    int n, swap, normalize, nperseg, hop;
    double norm_bias, norm_scale;
};
hipError_t launch_tap_decode(const TapArgs &a, int dt, hipStream_t s);
hipError_t launch_tap_fft(const TapArgs &a, hipStream_t s);
hipError_t launch_tap_stft(const TapArgs &a, hipStream_t s);
int smooth_waves_per_block(int m);                // 0: one item does not fit LDS
size_t post_lds_bytes(int m, int mode);
int post_waves_per_block(int m, int mode);       // 0: the post stage cannot hold one item in LDS
hipError_t launch_post(const PostArgs &a, hipStream_t s);
hipError_t launch_chain(const DecimArgs &a, int L, int dt, bool nco, bool iq, size_t n_chunks, hipStream_t s);
#define SDRT_I8_DIGITS 5
#define SDRT_I8T_G 1            // blocks per matrix-core column in the 16x16x64 int8 kernel (the default path)
hipError_t launch_mfma_selftest(const double *A, const double *B, double *D, hipStream_t s);
hipError_t launch_demod(const double *y, double *out, const double *resample_odd, int rows, int m, int mode, hipStream_t s);
bool chain_i8t_supported(int L, int dt, int n);
size_t chain_i8t_lds_bytes(int n, int L, int m);
hipError_t launch_chain_i8t(const DecimArgs &a, int L, int dt, size_t n_chunks, hipStream_t s);
hipError_t launch_i8t_selftest(const int *A, const int *B, int *D, hipStream_t s);
// wide-column persistent int8 kernel (kernels_i8w.hip): q in {32, 64}, 8/16-bit integer encodings, 8 or 16 tiles per chunk
bool chain_i8w_supported(int q, int dt, int n, bool iq);
bool chain_i8w_rs_supported(int dt, int n);       // its real-stream form (no shift, no IQ correction)
size_t chain_i8w_lds_bytes(int q, int n, int iq, int dt);      // iq: 0 none, 1 prefix tables, 2 fused (chain_i8w_kernel)
hipError_t launch_chain_i8w(const DecimArgs &a, int q, int dt, size_t n_chunks, int n_cus, hipStream_t s);
// float32 records of `tiles` tiles (4096 samples each) -> their four byte slices in place + the tile exponents (many-row contexts)
hipError_t launch_f32_preslice(const void *raw, void *out, int *exps, size_t tiles, int swap, hipStream_t s);
// per-sample modal form for decimation < 8 and the workgroup-per-item demodulation stage for m > 4096 (kernels_seq.hip)
bool chain_seq_supported(int q, int n);
hipError_t launch_chain_seq(const DecimArgs &a, int q, int dt, bool nco, bool iq, size_t n_chunks, hipStream_t s);
bool post_big_supported(int m, int mode);
hipError_t launch_post_big(const PostArgs &a, hipStream_t s);
bool post_bigfft_supported(const PostArgs &a);        // FM at 4096 .. 16384 outputs per chunk with the packed FFT tables: a workgroup transform
hipError_t launch_post_bigfft(const PostArgs &a, hipStream_t s);
// decimation 2 .. 16 of 16-bit records in one pass on the int8 matrix cores (kernels_sq.hip)
struct SqArgs {
    const void *raw;            // [n_chunks][n][2] 16-bit records
    double *ydec;               // [row][chunk][m][2] decimated samples, or with pair_out [row][chunk][m/2][2] products y[2i] conj(y[2i+1])
    size_t n_chunks;
    int n, m, rows, swap, pair_out;
    const SqTables *tab;        // device copy
    const double *share;        // [256 / q][8][2]   (sdrt_build_sq_tables)
    const signed char *agg;     // [rows][2][S][8][64][16]      (sdrt_build_sq_row)
    const signed char *master;  // [rows][2][S][ncp][1024]
    const double *nco_samp;     // [rows][256][2] exp(i theta j) (ones without a shift)
    const double *nco_col;      // [rows][n / 256][2] exp(i theta 256 c)
    const double *ybias;        // [rows][m][2] (zeros while it is being calibrated)
    double comb_agg, comb_loc;  // combine factors of the two plane sets (the same for every row)
    int touch;                  // != 0: the next chunk's lines are touched during the local phase
    int stagger;                // > 0: workgroup b starts (b / 8) x stagger sleeps of 64 cycles late (its XCD's workgroups apart in phase)
    unsigned long long *stamps; // optional diagnostic (-DSDRT_STAMPS builds, tools/stamps_sq.py): [wg < stamp_wgs][wave][16] s_memtime
    int stamp_wgs;
};
bool chain_sq_supported(int q, int dt, int n);
hipError_t launch_chain_sq(const SqArgs &a, int q, int dt, int n_cus, hipStream_t s);
// float32 records in the reference's own complex64 arithmetic (kernels_f32c.hip; sdrt_config.faithful_f32)
struct F32cArgs {
    const void *raw;            // [n_chunks][n][2] float32 records
    const double *shift;        // [rows][n][2] the shift table exactly as the reference builds it with numpy (or NULL: no shift)
    const double *shift_t;      // the same table as [rows / 64][n][64][2] where rows % 64 == 0 (f32c_fused_kernel), else NULL
    float *work;                // [groups of 64 items][n + 54][64][2] complex64: odd-extended shifted samples, filtered in place
    double *ydec;               // [row][chunk][m][2]
    size_t n_chunks;            // chunks in the batch = row stride of ydec
    int n, m, q, rows, swap;
    float sos[4][6];            // the decimator's sections as scipy casts them to the data's dtype (real parts; the imaginary are 0)
    float zi[4][2];             // scipy.signal.sosfilt_zi of that table
};
size_t f32c_work_bytes(int n, int rows, size_t n_chunks);
hipError_t launch_f32c(const F32cArgs &a, hipStream_t s);
hipError_t launch_swap_selftest(const double *x, const double *y, double *out, hipStream_t s);
