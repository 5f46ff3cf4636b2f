// C ABI of libsdrterm_hip.so (see include/sdrterm_hip.h).  Host side only: context management, table
// construction/upload, launches.  There is deliberately no CPU compute path here: every processing entry point
// needs a working HIP device and fails with SDRT_ENODEVICE / SDRT_EHIP otherwise.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "kernels.h"

#include <atomic>
#include <mutex>

namespace {

// every device allocation of this file goes through here: sdrt_alloc_count() lets a caller (and the tests) check that the
// steady state allocates nothing (sdrt_reserve)
std::atomic<unsigned long long> g_dev_allocs{0};
template <typename T> hipError_t counted_malloc(T **p, size_t bytes) {
    g_dev_allocs.fetch_add(1, std::memory_order_relaxed);
    return ::hipMalloc(p, bytes);
}
#define hipMalloc(p, bytes) counted_malloc((p), (bytes))

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

// Diagnostic switches (kernel-variant A/B measurements, the tests that force a form): ONE environment variable,
// SDRTERM_DIAG="key=value,key=value", read where a context is created / a kernel is chosen.  Configuration proper goes through
// sdrt_config (stage_a, faithful_f32, ...); nothing a caller needs is only reachable from here.
const char *sdrt_diag(const char *key) {
    static thread_local char val[64];
    const char *e = getenv("SDRTERM_DIAG");
    if (!e || !key) return nullptr;
    const size_t kl = strlen(key);
    for (const char *p = e; *p;) {
        const char *end = strchr(p, ',');
        const size_t len = end ? (size_t) (end - p) : strlen(p);
        if (len > kl && strncmp(p, key, kl) == 0 && p[kl] == '=') {
            const size_t vl = len - kl - 1 < sizeof(val) - 1 ? len - kl - 1 : sizeof(val) - 1;
            memcpy(val, p + kl + 1, vl);
            val[vl] = 0;
            return val;
        }
        if (!end) break;
        p = end + 1;
    }
    return nullptr;
}
int sdrt_diag_int(const char *key, int dflt) {
    const char *v = sdrt_diag(key);
    return v ? atoi(v) : dflt;
}

// used by the other translation units of the library (comm.hip)
int sdrt_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(SDRT_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const int kItemSize[8] = {1, 1, 2, 2, 4, 4, 4, 8};

struct Pending {
    int which;
    hipEvent_t a, b;
};

}  // namespace

#define SDRT_HOST_SLOTS 3
struct HostSlot {
    void *d_raw = nullptr;
    double *d_out = nullptr, *h_out = nullptr, *out_host = nullptr;
    size_t cap_chunks = 0, out_bytes = 0;
    hipEvent_t e_in = nullptr, e_run = nullptr, e_out = nullptr;
    bool busy = false, direct_out = false;
    int64_t ticket = -1;
};

struct sdrt_ctx {
    sdrt_config cfg;
    int L = 0, m = 0, n_tiles = 0, rows = 1;
    bool nco = false;
    double norm_scale = 1.0, norm_bias = 0.0;
    DecimTables *d_tab = nullptr;
    OutTables *d_otab = nullptr;
    double *d_nco_tile = nullptr, *d_nco_phase = nullptr, *d_resample = nullptr, *d_out_sos = nullptr;
    double *d_fft_tw = nullptr, *d_fft_shift = nullptr;
    double *d_fftp_tw = nullptr, *d_fftp_ab = nullptr;     // packed form of the resampler's transform (post_fftp_kernel)
    double *d_fft8 = nullptr;         // eight-lanes-per-item demodulation stage (kernels_post8.hip): transform tables ...
    Post8Tables *d_p8tab = nullptr;   // ... and its lowpass constants
    int p8_lpi = 0;
    double *d_stash = nullptr;        // ... and the angles of the items in flight (grow-only)
    size_t stash_cap = 0;
    double *d_smooth = nullptr;       // taps | left | right of --smooth-output (sdrt_set_smoothing)
    int sm_window = 0, sm_ntaps = 0, sm_first = 0, sm_edge = 0;
    bool fft_ok = false, out_modal = false;
    double *d_y = nullptr;            // [row][chunk][m][2] decimated samples between the two kernels
    size_t y_cap_chunks = 0;
    HostSlot hs[SDRT_HOST_SLOTS];     // host path: batches in flight (sdrt_submit_host / sdrt_wait)
    int64_t hs_next = 0;
    hipStream_t hs_h2d = nullptr, hs_run = nullptr, hs_d2h = nullptr;
    bool use_i8 = false;              // stage A on the int8 matrix cores (8/16-bit integer encodings): use_i8w || use_i8t
    bool use_i8t = false;             //   ... one block per column (16x16x64), no IQ correction
    bool use_i8w = false;             //   ... wide columns, persistent workgroups (default where supported)
    bool use_seq = false;             // decimation < 8: per-sample modal form (chain_seq_kernel)
    bool refine = false;              // stage_a = auto with a second look: (chunk, row) items whose in-band content is near the int8 forms' operator
                                      // image are listed by the demodulation stage and done again in float64 (enqueue_post)
    double refine_thr = 0.0;          //   threshold on an item's largest pair product
    int *d_cond_list = nullptr;       //   [refine_cap] listed items
    unsigned *d_cond_count = nullptr; //   [2] listed by the launch in flight; running total
    size_t refine_cap = 0;
    int stage = SDRT_STAGE_AUTO;      // enum sdrt_stage_a in effect (sdrt_config.stage_a unless SDRTERM_STAGE_A overrides it)
    bool use_sq = false;              // decimation 2 .. 16 of 16-bit records in one pass on the int8 matrix cores (kernels_sq.hip)
    SqTables *d_sqtab = nullptr;
    double *d_sq_share = nullptr;
    signed char *d_sq_agg = nullptr, *d_sq_master = nullptr;
    double *d_sq_samp = nullptr, *d_sq_col = nullptr;      // exp(i theta j), exp(i theta 256 c) per row (ones without a shift)
    double sq_comb_agg = 0.0, sq_comb_loc = 0.0;
    bool f32c = false;                // float32 records in the reference's complex64 arithmetic (sdrt_config.faithful_f32, kernels_f32c.hip)
    F32cArgs f32c_args;               //   coefficient tables
    double *d_shift_tab = nullptr;    //   [rows][n][2] the host's shift table
    double *d_shift_tab_t = nullptr;  //   the same as [rows / 64][n][64][2] where rows % 64 == 0 (f32c_fused_kernel)
    float *d_f32c_work = nullptr;     //   the extended sequences of a batch (grow-only)
    size_t f32c_cap = 0;
    int iq_gs = 0;                    // samples per group of the IQ prefix tables
    WideTables *d_wtab = nullptr;
    double *d_edge_ab = nullptr, *d_edge_pq = nullptr, *d_gvec = nullptr;
    bool wide_rs = false;             // wide kernel in its real-stream form (no shift, no IQ correction)
    // decimation 8 / 16 on the wide kernel: a column has 32 / 16 outputs, the kernel finishes eight per pass (tables.h:
    // sdrt_build_wide_tables_pass).  Pass 0 lives in the fields above / below (d_wtab, d_w8mid, d_gvec, d_ybias, w8mid_combine);
    // further passes here
    struct WidePass { WideTables *d_wtab = nullptr; signed char *d_w8mid = nullptr; double *d_gvec = nullptr, *d_ybias = nullptr; double w8mid_combine = 0.0; };
    std::vector<WidePass> more_passes;
    int nout_pass = 0;                // outputs per column and pass (0: not the wide kernel)
    int n_cus = 256;
    // per-row demodulation modes (sdrt_config.row_modes): the rows of each mode, on the device, for the demodulation stage
    bool mixed = false;
    int *d_rows_of[2] = {nullptr, nullptr};      // [SDRT_FM], [SDRT_AM]
    int *d_row_pair = nullptr;                   // per-row modes: [rows] 1 = FM row (its items can leave the chain kernel as pair products)
    int n_rows_of[2] = {0, 0};
    int sched = 0;                    // diagnostic builds only (SDRTERM_DIAG=sched, kernels_i8w.hip: SDRT_SCHED_EXP)
    int dual = 0;                     // wide kernel: one wave per SIMD on two tiles at once (SDRTERM_DIAG=wide_dual)
    signed char *d_w8 = nullptr, *d_w8mid = nullptr;
    DecimTables *d_tab8 = nullptr;    // tables for columns of SDRT_I8T_G blocks
    double *d_nco_tile8 = nullptr, *d_nco_phase8 = nullptr;
    double *d_ybias = nullptr;
    double *d_wsum = nullptr;
    bool f32_pre = false;             // float32 with several rows on the wide kernel: the records are sliced once per chunk, not per row
    void *d_pre = nullptr;            //   [chunks][n][2] slices in place of the floats (launch_f32_preslice)
    int *d_pre_exp = nullptr;         //   [chunks][4] tile exponents
    size_t pre_cap = 0;
    void *d_pair_raw = nullptr;       // wide kernel, chunks of four tiles: a single chunk doubled into a pair
    double *d_pair_y = nullptr;
    signed char *d_w8iq = nullptr;    // wide kernel with IQ correction: packed offset rows, their constants, the offset response R
    double *d_iq_consts = nullptr, *d_iq_R = nullptr;
    double iq_combine = 0.0;
    bool iq_fused = false;            // wide kernel computes the offsets itself (one row; several rows share the prefix pass instead)         // float32 on the int8 matrix cores: weight-row sums (DecimArgs.wsum)
    double w8_combine = 0.0, w8mid_combine = 0.0, x_offset = 0.0;
    int n_tiles8 = 0;
    bool iq = false;
    double *d_iq_state = nullptr;     // [2] carried offset, lives on the device
    double *d_iq_pow = nullptr;       // [n/GS + 1]
    double *d_iq_colw = nullptr;      // [259] iq_cols_kernel's column weights (IqArgs.col_w)
    double *d_iq_group = nullptr, *d_iq_agg = nullptr, *d_iq_chunk = nullptr;
    size_t iq_cap_chunks = 0;
    double *d_iq_range = nullptr;     // scratch of sdrt_iq_range_device: [n] aggregates, [n + 1] offsets, [1] state
    size_t iq_range_cap = 0;
    double iq_pow_e = 1.0, iq_pow_n = 1.0;
    unsigned long long *stamps_dev = nullptr;   // diagnostic only (sdrt_debug_stamps)
    int stamp_wgs = 0;
    bool profile = false;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    double ms[3] = {0, 0, 0};
    int64_t launches[3] = {0, 0, 0};
};

namespace {

bool want_i8_fallback_t(int L, int dt, int n, int m) {
    return chain_i8t_supported(L, dt, n) && chain_i8t_lds_bytes(n, L, m) <= 160 * 1024;
}

hipEvent_t get_event(sdrt_ctx *c) {
    if (!c->free_events.empty()) {
        hipEvent_t e = c->free_events.back();
        c->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

int resolve_pending(sdrt_ctx *c) {
    for (auto &p : c->pending) {
        HIP_TRY(hipEventSynchronize(p.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, p.a, p.b));
        c->ms[p.which] += (double) t;
        c->launches[p.which] += 1;
        c->free_events.push_back(p.a);
        c->free_events.push_back(p.b);
    }
    c->pending.clear();
    return SDRT_OK;
}

struct Timed {
    sdrt_ctx *c;
    hipStream_t s;
    int which;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(sdrt_ctx *c_, hipStream_t s_, int w) : c(c_), s(s_), which(w) {
        if (c->profile) {
            a = get_event(c);
            b = get_event(c);
            if (a && b) (void) hipEventRecord(a, s);
        }
    }
    ~Timed() {
        if (c->profile && a && b) {
            (void) hipEventRecord(b, s);
            c->pending.push_back({which, a, b});
        }
    }
};

template <class T>
int upload(T **dst, const void *src, size_t bytes) {
    HIP_TRY(hipMalloc(dst, bytes));
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return SDRT_OK;
}


// ---- stage A on the int8 matrix cores: table construction per form ----------------------------------------------------
void free_stage_a(sdrt_ctx *c) {
    auto drop = [](auto *&p) { if (p) (void) hipFree(p); p = nullptr; };
    drop(c->d_wtab); drop(c->d_edge_ab); drop(c->d_edge_pq); drop(c->d_gvec);
    drop(c->d_w8); drop(c->d_w8mid); drop(c->d_tab8); drop(c->d_nco_tile8); drop(c->d_nco_phase8); drop(c->d_wsum);
    drop(c->d_w8iq); drop(c->d_iq_consts); drop(c->d_iq_R);
    for (auto &wp : c->more_passes) { drop(wp.d_wtab); drop(wp.d_w8mid); drop(wp.d_gvec); drop(wp.d_ybias); }
    c->more_passes.clear();
    c->nout_pass = 0;
    c->use_i8w = c->use_i8t = c->wide_rs = false;
    c->w8_combine = c->w8mid_combine = 0.0;
}

// wide columns (256 samples, chain_i8w_kernel): tables, NCO phasors per column, digit planes per row (NCO folded in).
// Leaves c->use_i8w set on success; returns SDRT_OK with use_i8w clear when this form does not apply.
template <class ScaleFn>
int setup_wide(sdrt_ctx *c, const sdrt_config *cfg, ScaleFn &) {
    const int q = cfg->decimation, n = cfg->n_samples;
    const int S = SDRT_I8_DIGITS, LC = SDRT_WIDE_LC, KS = LC / 32;
    const int nout_total = LC / q, nout_pass = nout_total > SDRT_WIDE_MAXOUT ? SDRT_WIDE_MAXOUT : nout_total, passes = nout_total / nout_pass;
    const int NR = 2 * nout_pass;
    const char *msg = nullptr;
    WideTables wtab;
    if (sdrt_build_wide_tables_pass(cfg->dec_sos, cfg->dec_sections, q, c->norm_scale, nout_pass, 0, &wtab, &msg) != 0)
        return fail(SDRT_EINVAL, "wide-column tables: %s", msg ? msg : "failed");
    const char *rsenv = sdrt_diag("wide_rs");               // diagnostics: "0" keeps the general form
    c->wide_rs = passes == 1 && !c->nco && !c->iq && chain_i8w_rs_supported(cfg->dtype, n) && !(rsenv && strcmp(rsenv, "0") == 0);
    const size_t per_row = c->wide_rs ? (size_t) S * (LC / 64) * 64 * 16 : (size_t) 2 * S * KS * 64 * 16;
    const size_t per_row_loc = c->wide_rs ? (size_t) S * (LC / 64) * 4 * (NR / 2) * 16 : (size_t) S * KS * 4 * NR * 16;
    std::vector<signed char> w8(per_row * c->rows), w8l(per_row_loc * c->rows);
    std::vector<double> samp((size_t) c->rows * LC * 2), blk((size_t) c->rows * (n / LC) * 2);
    std::vector<double> eab((size_t) c->rows * 16 * 32 * 4), epq((size_t) c->rows * 16 * 4), gv((size_t) c->rows * SDRT_WIDE_GV * 2);
    const double lam = c->iq ? cfg->iq_inductance : 0.0;
    bool ok = true;
    for (int r = 0; r < c->rows && ok; ++r) {
        double comb = 0.0, combl = 0.0;
        double *sp = &samp[(size_t) r * LC * 2], *bp = &blk[(size_t) r * (n / LC) * 2];
        if (c->nco) sdrt_build_nco(cfg->row_freqs[r], cfg->fs, LC, n / LC, sp, bp);
        const int brc = c->wide_rs
            ? sdrt_build_i8_weights_wide_rs(cfg->dec_sos, cfg->dec_sections, q, c->norm_scale, S, w8.data() + per_row * r, &comb,
                                            w8l.data() + per_row_loc * r, &combl)
            : sdrt_build_i8_weights_wide_pass(cfg->dec_sos, cfg->dec_sections, q, c->norm_scale, c->nco ? sp : nullptr, S, lam, nout_pass, 0,
                                              w8.data() + per_row * r, &comb, w8l.data() + per_row_loc * r, &combl,
                                              &gv[(size_t) r * 2 * SDRT_WIDE_GV]);
        sdrt_build_wide_edge(&wtab, c->nco ? sp : nullptr, c->nco ? bp : nullptr, n, lam, &eab[(size_t) r * 2048],
                             &epq[(size_t) r * 64]);
        // one power-of-two scale per context (rows differ by a rotation of the weights only); a digit overflow or unequal
        // scales: this form does not apply
        if (brc != 0 || comb == 0.0 || combl == 0.0) ok = false;
        else if (r == 0) { c->w8_combine = comb; c->w8mid_combine = combl; }
        else if (comb != c->w8_combine || combl != c->w8mid_combine) ok = false;
    }
    if (!ok) { free_stage_a(c); return SDRT_OK; }
    int rc = upload(&c->d_wtab, &wtab, sizeof(wtab));
    if (rc == SDRT_OK && c->nco) {
        rc = upload(&c->d_nco_tile8, samp.data(), samp.size() * sizeof(double));
        if (rc == SDRT_OK) rc = upload(&c->d_nco_phase8, blk.data(), blk.size() * sizeof(double));
    }
    if (rc == SDRT_OK) rc = upload(&c->d_w8, w8.data(), w8.size());
    if (rc == SDRT_OK) rc = upload(&c->d_w8mid, w8l.data(), w8l.size());
    if (rc == SDRT_OK) rc = upload(&c->d_edge_ab, eab.data(), eab.size() * sizeof(double));
    if (rc == SDRT_OK) rc = upload(&c->d_edge_pq, epq.data(), epq.size() * sizeof(double));
    if (rc == SDRT_OK) rc = upload(&c->d_gvec, gv.data(), gv.size() * sizeof(double));
    if (rc == SDRT_OK && c->iq_fused) {
        // the offsets come out of two more matrix rows (no second pass over the input) ...
        std::vector<signed char> wq((size_t) KS * 64 * 16);
        std::vector<double> consts(SDRT_IQROW_CONSTS);
        if (sdrt_build_i8_offset_rows(lam, S, wq.data(), &c->iq_combine, consts.data()) != 0) { free_stage_a(c); return SDRT_OK; }
        rc = upload(&c->d_w8iq, wq.data(), wq.size());
        if (rc == SDRT_OK) rc = upload(&c->d_iq_consts, consts.data(), consts.size() * sizeof(double));
        // ... relative to a zero offset at the chunk's start; the true one enters linearly: R = D(nco (1-lambda)^t) per row, from
        // the fp64 kernel on that sequence
        std::vector<double> gz((size_t) n * 2, 0.0), pw((size_t) n + 1);
        sdrt_build_iq_pow(lam, n, pw.data());
        for (int i = 0; i < n; ++i) gz[2 * i] = pw[i];
        double *d_gz = nullptr;
        if (rc == SDRT_OK) rc = upload(&d_gz, gz.data(), gz.size() * sizeof(double));
        if (rc == SDRT_OK && hipMalloc(&c->d_iq_R, (size_t) c->rows * c->m * 2 * sizeof(double)) != hipSuccess)
            rc = fail(SDRT_ENOMEM, "hipMalloc failed");
        if (rc == SDRT_OK) {
            DecimArgs da;
            memset(&da, 0, sizeof(da));
            da.raw = d_gz;
            da.tab = c->d_tab;
            da.nco_tile = c->d_nco_tile;
            da.nco_phase = c->d_nco_phase;
            da.ydec = c->d_iq_R;
            da.n_chunks = 1;
            da.n = n;
            da.m = c->m;
            da.n_tiles = c->n_tiles;
            da.rows = c->rows;
            hipError_t e = launch_chain(da, c->L, SDRT_F64, c->nco, false, 1, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) rc = fail(SDRT_EHIP, "offset response kernel failed: %s", hipGetErrorString(e));
        }
        (void) hipFree(d_gz);
    }
    // the further passes of a small decimation: their own tables, local planes and rank-one vectors (the main planes are the same)
    for (int p = 1; p < passes && rc == SDRT_OK && ok; ++p) {
        sdrt_ctx::WidePass wp;
        WideTables wt2;
        if (sdrt_build_wide_tables_pass(cfg->dec_sos, cfg->dec_sections, q, c->norm_scale, nout_pass, p * nout_pass, &wt2, &msg) != 0) { ok = false; break; }
        std::vector<signed char> w8x(per_row), w8l2(per_row_loc * c->rows);
        std::vector<double> gv2((size_t) c->rows * SDRT_WIDE_GV * 2);
        for (int r = 0; r < c->rows && ok; ++r) {
            double comb = 0.0, combl = 0.0;
            const double *sp = &samp[(size_t) r * LC * 2];
            const int brc = sdrt_build_i8_weights_wide_pass(cfg->dec_sos, cfg->dec_sections, q, c->norm_scale, c->nco ? sp : nullptr, S, lam, nout_pass,
                                                            p * nout_pass, w8x.data(), &comb, w8l2.data() + per_row_loc * r, &combl, &gv2[(size_t) r * 2 * SDRT_WIDE_GV]);
            if (brc != 0 || combl == 0.0) ok = false;
            else if (r == 0) wp.w8mid_combine = combl;
            else if (combl != wp.w8mid_combine) ok = false;
        }
        if (!ok) break;
        rc = upload(&wp.d_wtab, &wt2, sizeof(wt2));
        if (rc == SDRT_OK) rc = upload(&wp.d_w8mid, w8l2.data(), w8l2.size());
        if (rc == SDRT_OK) rc = upload(&wp.d_gvec, gv2.data(), gv2.size() * sizeof(double));
        c->more_passes.push_back(wp);
    }
    if (!ok) { free_stage_a(c); return SDRT_OK; }
    c->nout_pass = nout_pass;
    c->use_i8w = rc == SDRT_OK;
    return rc;
}

// one block of L <= 64 samples per column (chain_i8t_kernel): tables for columns of G blocks, digit planes per row
template <class ScaleFn>
int setup_i8t(sdrt_ctx *c, const sdrt_config *cfg, ScaleFn &scale_tables) {
    const int q = cfg->decimation, n = cfg->n_samples, L = c->L;
    const int S = SDRT_I8_DIGITS, G = SDRT_I8T_G, LC = L * G, KS = LC / 32, NR = 2 * (G - 1);
    const char *msg = nullptr;
    DecimTables tab8;
    if (sdrt_build_decim_tables(cfg->dec_sos, cfg->dec_sections, q, G, &tab8, &msg) != 0)
        return fail(SDRT_EINVAL, "decimator tables: %s", msg ? msg : "failed");
    scale_tables(tab8);
    c->n_tiles8 = n / (SDRT_TB * LC);
    const size_t per_row = (size_t) 2 * S * KS * 64 * 16, per_row_mid = (size_t) S * KS * 4 * NR * 16;
    std::vector<signed char> w8(per_row * c->rows), w8m(per_row_mid * c->rows + 16);
    std::vector<double> samp((size_t) c->rows * LC * 2), blk((size_t) c->rows * (n / LC) * 2);
    bool ok = true;
    for (int r = 0; r < c->rows && ok; ++r) {
        double comb = 0.0, combm = 0.0;
        double *sp = &samp[(size_t) r * LC * 2], *bp = &blk[(size_t) r * (n / LC) * 2];
        if (c->nco) sdrt_build_nco(cfg->row_freqs[r], cfg->fs, LC, n / LC, sp, bp);
        const int brc = sdrt_build_i8_weights16(cfg->dec_sos, cfg->dec_sections, &tab8, c->norm_scale, c->nco ? sp : nullptr, S,
                                                w8.data() + per_row * r, &comb, w8m.data() + per_row_mid * r, &combm);
        if (brc != 0 || comb == 0.0) ok = false;
        else if (r == 0) { c->w8_combine = comb; c->w8mid_combine = combm; }
        else if (comb != c->w8_combine || combm != c->w8mid_combine) ok = false;
    }
    if (!ok) { free_stage_a(c); return SDRT_OK; }          // the fp64 kernel takes over
    int rc = upload(&c->d_tab8, &tab8, sizeof(tab8));
    if (rc == SDRT_OK && c->nco) {
        rc = upload(&c->d_nco_tile8, samp.data(), samp.size() * sizeof(double));
        if (rc == SDRT_OK) rc = upload(&c->d_nco_phase8, blk.data(), blk.size() * sizeof(double));
    }
    if (rc == SDRT_OK) rc = upload(&c->d_w8, w8.data(), w8.size());
    if (rc == SDRT_OK && per_row_mid) rc = upload(&c->d_w8mid, w8m.data(), per_row_mid * c->rows);
    if (rc == SDRT_OK && cfg->dtype == SDRT_F32) {
        // sum over K of every weight row, from the digit planes as the matrix cores see them: what the constant 0x808080 that
        // the XOR leaves out of the three lower slices contributes per column (kernels_i8t.hip: I8Enc<SDRT_F32>)
        std::vector<double> ws((size_t) c->rows * 32, 0.0);
        for (int r = 0; r < c->rows; ++r)
            for (int tile = 0; tile < 2; ++tile)
                for (int d = 0; d < S; ++d) {
                    const double wd = std::ldexp(c->w8_combine, 8 * d);
                    for (int ks = 0; ks < KS; ++ks)
                        for (int lane = 0; lane < 64; ++lane) {
                            const signed char *p = &w8[per_row * r + ((((size_t) tile * S + d) * KS + ks) * 64 + lane) * 16];
                            long sum = 0;
                            for (int j = 0; j < 16; ++j) sum += p[j];
                            ws[(size_t) r * 32 + tile * 16 + (lane & 15)] += wd * (double) sum;
                        }
                }
        rc = upload(&c->d_wsum, ws.data(), ws.size() * sizeof(double));
    }
    c->use_i8t = rc == SDRT_OK;
    return rc;
}

// The int8 kernels filter x - x_offset (the slices' range); the decimator is linear, so its response to the constant the
// slices leave out (plus the normalisation bias) is one vector per row.  Two parts (round 4):
//  * the normalisation bias (re only): the fp64 kernel once on that constant chunk;
//  * the slices' offset: MINUS what the int8 kernel itself makes of an all-zero chunk (calibrate_zero_response, below) -- the
//    reference turns a zero chunk into exact zeros (angle(0) = 0, demodulation.py:30-32), and only the kernel's own response,
//    quantised weights and rounding included, cancels to the last bit.  (The fp64 response to the constant, used before,
//    left 128 x 2^-39 of the weights behind: harmless beside any signal, but the FM angle of a silent chunk was noise.)
int setup_ybias(sdrt_ctx *c, const sdrt_config *cfg) {
    static const double kOff[8] = {0.0, 128.0, 128.0, 32896.0, 0, 0, 0, 0};
    const int n = cfg->n_samples, m = c->m;
    c->x_offset = kOff[cfg->dtype];
    const size_t bytes = (size_t) c->rows * m * 2 * sizeof(double);
    if (!c->iq && (c->x_offset != 0.0 || c->norm_bias != 0.0)) {
        if (hipMalloc(&c->d_ybias, bytes) != hipSuccess) return fail(SDRT_ENOMEM, "hipMalloc failed");
        if (hipMemset(c->d_ybias, 0, bytes) != hipSuccess) return fail(SDRT_EHIP, "hipMemset failed");
        if (c->norm_bias == 0.0) return SDRT_OK;
        std::vector<double> cz((size_t) n * 2, 0.0);
        for (int i = 0; i < n; ++i) cz[2 * i] = c->norm_bias;
        double *d_cz = nullptr;
        int rc = upload(&d_cz, cz.data(), cz.size() * sizeof(double));
        if (rc == SDRT_OK) {
            DecimArgs da;
            memset(&da, 0, sizeof(da));
            da.raw = d_cz;
            da.tab = c->d_tab;
            da.nco_tile = c->d_nco_tile;
            da.nco_phase = c->d_nco_phase;
            da.ydec = c->d_ybias;
            da.n_chunks = 1;
            da.n = n;
            da.m = m;
            da.n_tiles = c->n_tiles;
            da.rows = c->rows;
            hipError_t e = launch_chain(da, c->L, SDRT_F64, c->nco, false, 1, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) rc = fail(SDRT_EHIP, "ybias kernel failed: %s", hipGetErrorString(e));
        }
        (void) hipFree(d_cz);
        return rc;
    }
    if (c->use_i8w) {
        // the wide-column kernel always adds its ybias operand (no branch in its inner loop): zeros here
        if (hipMalloc(&c->d_ybias, bytes) != hipSuccess) return fail(SDRT_ENOMEM, "hipMalloc failed");
        if (hipMemset(c->d_ybias, 0, bytes) != hipSuccess) return fail(SDRT_EHIP, "hipMemset failed");
    }
    return SDRT_OK;
}

int calibrate_zero_response(sdrt_ctx *c);       // after enqueue_chain

// several passes (decimation 8 / 16 on the wide kernel): every pass adds its own slice of the constant response, [row][m / passes]
int setup_pass_ybias(sdrt_ctx *c) {
    const size_t bytes = (size_t) c->rows * (c->m / (1 + c->more_passes.size())) * 2 * sizeof(double);
    for (auto &wp : c->more_passes) {
        if (hipMalloc(&wp.d_ybias, bytes) != hipSuccess) return fail(SDRT_ENOMEM, "hipMalloc failed");
        if (hipMemset(wp.d_ybias, 0, bytes) != hipSuccess) return fail(SDRT_EHIP, "hipMemset failed");
    }
    return SDRT_OK;
}

int enqueue_chain(sdrt_ctx *c, const void *raw_dev, size_t n_chunks, double *ybuf, hipStream_t s, int n_cus, bool fix_later = false, bool pair_out = false,
                  bool row_pairs = false);

// Decimation 2 .. 16 of 16-bit records in one pass (kernels_sq.hip): the context's tables, every row's planes (NCO folded in), and
// ybias = - (the kernel's own response to an all-zero chunk) + normalisation bias x (its response to a unit chunk - that): the
// constant the byte slices leave out and the bias of --normalize-input come back exactly as the kernel itself sees constants, so a
// silent chunk cancels to the bit (section 1 of DESIGN.md).  `tab`: the context's DecimTables with the data scale applied.
int setup_sq(sdrt_ctx *c, const sdrt_config *cfg, const DecimTables &tab) {
    const int q = cfg->decimation, n = cfg->n_samples, S = SDRT_I8_DIGITS, LC = SDRT_SQ_LC, m = c->m;
    const int nout = LC / q, ncp = 2 * q >= 16 ? 1 : 16 / (2 * q);
    SqTables st;
    std::vector<double> share((size_t) nout * 8 * 2);
    if (sdrt_build_sq_tables(&tab, q, &st, share.data()) != 0) return SDRT_OK;       // (not this form: the others take over)
    const size_t per_agg = (size_t) 2 * S * (LC / 32) * 64 * 16, per_master = (size_t) SDRT_SQ_MASTER_BYTES(q, S, ncp);
    std::vector<signed char> agg(per_agg * c->rows), master(per_master * c->rows);
    std::vector<double> samp((size_t) c->rows * LC * 2), col((size_t) c->rows * (n / LC) * 2);
    for (int r = 0; r < c->rows; ++r) {
        double ca = 0.0, cl = 0.0;
        double *sp = &samp[(size_t) r * LC * 2];
        if (c->nco) sdrt_build_nco(cfg->row_freqs[r], cfg->fs, LC, n / LC, sp, &col[(size_t) r * (n / LC) * 2]);
        else {                                                           // (no shift: the kernel reads the same tables, holding ones)
            for (int i = 0; i < LC; ++i) { sp[2 * i] = 1.0; sp[2 * i + 1] = 0.0; }
            for (int i = 0; i < n / LC; ++i) { col[((size_t) r * (n / LC) + i) * 2] = 1.0; col[((size_t) r * (n / LC) + i) * 2 + 1] = 0.0; }
        }
        if (sdrt_build_sq_row(&tab, q, c->nco ? sp : nullptr, S, agg.data() + per_agg * r, &ca, master.data() + per_master * r, &cl) != 0) return SDRT_OK;
        if (r == 0) { c->sq_comb_agg = ca; c->sq_comb_loc = cl; }
        else if (ca != c->sq_comb_agg || cl != c->sq_comb_loc) return SDRT_OK;      // one power-of-two scale per context
    }
    int rc = upload(&c->d_sqtab, &st, sizeof(st));
    if (rc == SDRT_OK) rc = upload(&c->d_sq_share, share.data(), share.size() * sizeof(double));
    if (rc == SDRT_OK) rc = upload(&c->d_sq_agg, agg.data(), agg.size());
    if (rc == SDRT_OK) rc = upload(&c->d_sq_master, master.data(), master.size());
    if (rc == SDRT_OK) rc = upload(&c->d_sq_samp, samp.data(), samp.size() * sizeof(double));
    if (rc == SDRT_OK) rc = upload(&c->d_sq_col, col.data(), col.size() * sizeof(double));
    if (rc != SDRT_OK) return rc;
    c->use_sq = true;
    // ybias from the kernel's own responses
    const int bpc = (cfg->dtype == SDRT_S8 || cfg->dtype == SDRT_U8) ? 1 : 2;          // bytes per component
    const size_t cnt = (size_t) c->rows * m * 2, cb = (size_t) n * 2 * bpc;
    if (hipMalloc(&c->d_ybias, cnt * sizeof(double)) != hipSuccess) return fail(SDRT_ENOMEM, "hipMalloc failed");
    double *d_bias = c->d_ybias;
    if (hipMemset(d_bias, 0, cnt * sizeof(double)) != hipSuccess) return fail(SDRT_EHIP, "hipMemset failed");    // (the calibration runs with zeros)
    void *d_in = nullptr;
    double *d_r = nullptr;
    if (hipMalloc(&d_in, cb) != hipSuccess || hipMalloc(&d_r, cnt * sizeof(double)) != hipSuccess) rc = fail(SDRT_ENOMEM, "hipMalloc failed");
    std::vector<double> r0(cnt), r1(cnt), bias(cnt);
    std::vector<unsigned char> unit(cb, 0);
    auto respond = [&](std::vector<double> &dst) {
        if (rc == SDRT_OK && hipMemcpy(d_in, unit.data(), cb, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemcpy failed");
        if (rc == SDRT_OK) rc = enqueue_chain(c, d_in, 1, d_r, nullptr, c->n_cus);
        if (rc == SDRT_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(SDRT_EHIP, "calibration kernel failed");
        if (rc == SDRT_OK && hipMemcpy(dst.data(), d_r, cnt * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemcpy failed");
    };
    respond(r0);                                                     // all-zero records
    if (rc == SDRT_OK && c->norm_bias != 0.0) {
        // records with re = 1, im = 0 (in the records' byte order)
        const bool be = cfg->swap_endianness != 0;
        for (int i = 0; i < n; ++i) unit[(size_t) i * 2 * bpc + (bpc == 2 && be ? 1 : 0)] = 1;
        respond(r1);
    }
    if (rc == SDRT_OK) {
        for (size_t i = 0; i < cnt; ++i) bias[i] = -r0[i] + (c->norm_bias != 0.0 ? c->norm_bias * (r1[i] - r0[i]) : 0.0);
        if (hipMemcpy(d_bias, bias.data(), cnt * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemcpy failed");
    }
    (void) hipFree(d_in);
    (void) hipFree(d_r);
    return rc;
}

}  // namespace

extern "C" {

int sdrt_abi_version(void) { return SDRT_ABI_VERSION; }

size_t sdrt_config_size(void) { return sizeof(sdrt_config); }

const char *sdrt_last_error(void) { return g_err; }

int sdrt_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        if (count) *count = 0;
        return fail(SDRT_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    if (count) *count = n;
    return SDRT_OK;
}

int sdrt_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return SDRT_OK;
}

int sdrt_create(const sdrt_config *cfg, sdrt_ctx **out) {
    if (!cfg || !out) return fail(SDRT_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(sdrt_config))
        return fail(SDRT_EINVAL, "sdrt_config size mismatch (got %u, built with %zu)", cfg->struct_size, sizeof(sdrt_config));
    if (cfg->decimation < 2) return fail(SDRT_EINVAL, "Decimation must be at least 2.");  // dsp_processor.py:94-95
    if (cfg->dtype < 0 || cfg->dtype > 7) return fail(SDRT_EINVAL, "unknown dtype %d", cfg->dtype);
    if (cfg->mode < SDRT_FM || cfg->mode > SDRT_IM) return fail(SDRT_EINVAL, "unknown mode %d", cfg->mode);
    // per-row modes: FM / AM only; all rows alike is the plain case with that mode
    bool mixed = false, any_fm = cfg->mode == SDRT_FM;
    int uniform_mode = cfg->mode;
    if (cfg->row_modes) {
        if (cfg->n_rows < 1) return fail(SDRT_EINVAL, "n_rows must be >= 1");
        int cnt[2] = {0, 0};
        for (int r = 0; r < cfg->n_rows; ++r) {
            if (cfg->row_modes[r] != SDRT_FM && cfg->row_modes[r] != SDRT_AM) return fail(SDRT_EINVAL, "row_modes[%d] = %d: per-row modes are SDRT_FM or SDRT_AM", r, cfg->row_modes[r]);
            cnt[cfg->row_modes[r]]++;
        }
        mixed = cnt[0] > 0 && cnt[1] > 0;
        any_fm = cnt[0] > 0;
        uniform_mode = cnt[0] ? SDRT_FM : SDRT_AM;
    }
    const bool filtered = mixed || uniform_mode == SDRT_FM || uniform_mode == SDRT_AM;      // real / imaginary output: no output lowpass
    if (!cfg->dec_sos || (filtered && !cfg->out_sos)) return fail(SDRT_EINVAL, "dec_sos and out_sos are required");
    if (filtered && (cfg->out_sections < 1 || cfg->out_sections > SDRT_MAX_OUT_SECTIONS))
        return fail(SDRT_EINVAL, "out_sections must be 1..%d", SDRT_MAX_OUT_SECTIONS);
    if (cfg->n_rows < 1) return fail(SDRT_EINVAL, "n_rows must be >= 1");
    if (cfg->apply_shift && !cfg->row_freqs) return fail(SDRT_EINVAL, "row_freqs required when apply_shift is set");
    if (!(cfg->fs > 0)) return fail(SDRT_EINVAL, "fs must be positive");
    const int q = cfg->decimation, n = cfg->n_samples;
    // the reference's chunks hold a power-of-two number of samples (131072 bytes / record size), so every decimation factor
    // that divides them (dsp_processor.py:147,173) is a power of two
    if (q & (q - 1)) return fail(SDRT_EINVAL, "this build supports decimation factors that are powers of two (got %d)", q);
    if (n <= SDRT_EDGE) return fail(SDRT_EINVAL, "n_samples must exceed the filter pad length %d", SDRT_EDGE);
    // float32 records as the reference computes them (complex64 throughout the decimator): a path of its own
    const bool f32c = cfg->faithful_f32 != 0 && cfg->dtype == SDRT_F32;
    if (f32c) {
        if (cfg->dec_sections != 4 || !cfg->f32_sos || !cfg->f32_zi)
            return fail(SDRT_EINVAL, "faithful_f32 needs the four-section decimator of scipy.signal.decimate as f32_sos / f32_zi");
        if (cfg->apply_shift && !cfg->shift_table) return fail(SDRT_EINVAL, "faithful_f32 with apply_shift needs shift_table");
        if (cfg->correct_iq) return fail(SDRT_EINVAL, "faithful_f32 does not take correct_iq (the reference's complex128 gufunc leaves a complex64 buffer as it was)");
        if (cfg->stage_a == SDRT_STAGE_F64) return fail(SDRT_EINVAL, "faithful_f32 and stage_a = f64 contradict each other");
    }
    if (cfg->stage_a < SDRT_STAGE_AUTO || cfg->stage_a > SDRT_STAGE_F64) return fail(SDRT_EINVAL, "unknown stage_a %d", cfg->stage_a);
    const bool seq = q < 8 && !f32c;                          // per-sample form; block forms from 8 up
    const int L = q < SDRT_MAXL ? q : SDRT_MAXL;
    if (seq && !chain_seq_supported(q, n))
        return fail(SDRT_EINVAL, "decimation %d needs chunks of a multiple of %d samples (got %d)", q, 256 * (q > 16 ? q : 16), n);
    if (!seq && n % (SDRT_TB * L)) return fail(SDRT_EINVAL, "n_samples (%d) must be a multiple of %d", n, SDRT_TB * L);
    if (n % q) return fail(SDRT_EINVAL, "decimation must divide n_samples");   // dsp_processor.py:147,173
    const int m = n / q;
    if (any_fm && (m < 2 || (m & 1))) return fail(SDRT_EINVAL, "FM needs an even number of outputs per chunk");
    // the post stage keeps one (chunk, row) item per wave in LDS: FFT resampler for m/2 in {256 .. 2048}, else the direct forms
    {
        const int Nh = m / 2;
        const bool fft_fm = any_fm && (Nh == 256 || Nh == 512 || Nh == 1024 || Nh == 2048);
        for (int md = SDRT_FM; md <= SDRT_AM; ++md)
            if (filtered && (mixed || md == uniform_mode) && !(md == SDRT_FM && fft_fm) && post_waves_per_block(m, md) == 0 && !(m >= 8192 && post_big_supported(m, md)))
            return fail(SDRT_EINVAL, "%d outputs per chunk (n_samples / decimation) exceed what the demodulation stage holds in LDS", m);
    }
    const int iq_gs = seq ? 16 : L / 4;                      // samples per group of the IQ prefix tables
    if (cfg->correct_iq && n / iq_gs < 256)
        return fail(SDRT_EINVAL, "correct_iq needs at least %d samples per chunk", 256 * iq_gs);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(SDRT_ENODEVICE, "no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SDRT_EINVAL, "device %d out of range (%d found)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    sdrt_ctx *c = new (std::nothrow) sdrt_ctx();
    if (!c) return fail(SDRT_ENOMEM, "out of host memory");
    c->cfg = *cfg;
    c->cfg.row_freqs = nullptr;
    c->cfg.dec_sos = nullptr;
    c->cfg.out_sos = nullptr;
    c->cfg.row_modes = nullptr;
    c->cfg.f32_sos = nullptr;
    c->cfg.f32_zi = nullptr;
    c->cfg.shift_table = nullptr;
    c->cfg.mode = uniform_mode;
    c->mixed = mixed;
    c->L = L;
    c->m = m;
    c->n_tiles = n / (SDRT_TB * L);
    c->rows = cfg->n_rows;
    c->nco = cfg->apply_shift != 0;
    if (cfg->normalize) {
        // read_file.py:96: 1.6*(z - xmin)*d - 0.8 with complex z: only Re is offset.  Written as scale*(x + bias):
        // the bias is added per sample, the scale is folded into the (linear) operator tables below.
        c->norm_scale = 1.6 * cfg->norm_inv_range;
        c->norm_bias = -cfg->norm_xmin - 0.5 / cfg->norm_inv_range;
    }
    c->iq = cfg->correct_iq != 0;
    c->use_seq = seq;
    c->iq_gs = iq_gs;
    // the fp64 block kernel keeps a chunk's block states in LDS: too many blocks (8-bit records at decimation 8: 8192 of them) do
    // not fit.  The wide int8 kernel takes such a chunk where it applies (in passes, section 3c of DESIGN.md) -- without the
    // normalisation bias, whose response comes from the fp64 kernel; settled after the stage-A form has been chosen
    c->f32c = f32c;
    const size_t lds = seq || f32c ? 0 : chain_lds_bytes(n, L, m);
    const bool f64_fits = lds <= 160 * 1024;
    if (!f64_fits && !(chain_i8w_supported(q, cfg->dtype, n, cfg->correct_iq != 0) && c->norm_bias == 0.0) && !(chain_sq_supported(q, cfg->dtype, n) && !cfg->correct_iq)) {
        delete c;
        return fail(SDRT_EINVAL, "chunk needs %zu bytes of LDS (> 160 KiB): n_samples/decimation combination unsupported", lds);
    }

    int rc = SDRT_OK;
    DecimTables tab;
    const char *msg = nullptr;
    if (sdrt_build_decim_tables(cfg->dec_sos, cfg->dec_sections, q, 1, &tab, &msg) != 0) {
        delete c;
        return fail(SDRT_EINVAL, "decimator tables: %s", msg ? msg : "failed");
    }
    auto scale_tables = [&](DecimTables &t) {
        // everything up to the decimated samples is linear in the data: scaling the data == scaling these tables
        const double k = c->norm_scale;
        if (k == 1.0) return;
        for (int r = 0; r < 16; ++r) for (int j = 0; j < SDRT_MAXL; ++j) t.W[r][j] *= k;
        t.d0p *= k;
        for (int i = 0; i < SDRT_NMODE; ++i)
            for (int l = 0; l <= SDRT_EDGE; ++l)
                for (int z = 0; z < 2; ++z) { t.head[i][l][z] *= k; t.tailU[i][l][z] *= k; }
        for (int i = 0; i < SDRT_NMODE; ++i)
            for (int z = 0; z < 2; ++z) { t.ac[i][z] *= k; t.aa[i][z] *= k; }
    };
    scale_tables(tab);
    rc = upload(&c->d_tab, &tab, sizeof(tab));
    if (rc == SDRT_OK && c->iq) {
        // with normalisation the reference corrects the NORMALISED samples (read_file.py:102-103): the offset lives in
        // normalised units, while the kernels work on x + bias (un-scaled) and scale at the end; off' = off / scale obeys
        // the same recurrence, so the device state is kept in un-scaled units and converted in get/set.
        const int gs = iq_gs, ng = n / gs;
        std::vector<double> pw((size_t) ng + 1);
        sdrt_build_iq_pow(cfg->iq_inductance, ng, pw.data());          // (1-lambda)^k, k <= ng
        std::vector<double> pwg((size_t) ng + 1);
        std::vector<double> tmp((size_t) n + 1);
        sdrt_build_iq_pow(cfg->iq_inductance, n, tmp.data());
        for (int k = 0; k <= ng; ++k) pwg[k] = tmp[(size_t) k * gs];
        c->iq_pow_e = tmp[gs];                 // (1-lambda)^GS
        c->iq_pow_n = tmp[n];
        rc = upload(&c->d_iq_pow, pwg.data(), pwg.size() * sizeof(double));
        if (rc == SDRT_OK && n >= 256) {
            // the column form of the prefix pass (iq_cols_kernel): weights of a 256-sample column and three powers
            std::vector<double> cw(259);
            for (int t = 0; t < 256; ++t) cw[t] = cfg->iq_inductance * tmp[255 - t];
            cw[256] = 1.0 / tmp[32];
            cw[257] = tmp[256];
            cw[258] = tmp[224];
            rc = upload(&c->d_iq_colw, cw.data(), cw.size() * sizeof(double));
        }
        if (rc == SDRT_OK) {
            const double zero[2] = {0.0, 0.0};
            rc = upload(&c->d_iq_state, zero, sizeof(zero));
        }
    }
    if (rc == SDRT_OK && filtered) rc = upload(&c->d_out_sos, cfg->out_sos, sizeof(double) * 6 * cfg->out_sections);
    if (rc == SDRT_OK && filtered) {
        OutTables otab;
        sdrt_build_out_tables(cfg->out_sos, cfg->out_sections, m, &otab);
        rc = upload(&c->d_otab, &otab, sizeof(otab));
        c->out_modal = otab.nmodes >= 0;
    }
    if (rc == SDRT_OK && c->nco) {
        // exp(i theta t) is factored as (sample within block) x (block within chunk): L + n/L table entries per row
        const int Ln = seq ? 64 : L, nblk = n / Ln;           // chain_seq_kernel: 64 samples per table block
        std::vector<double> tile((size_t) c->rows * Ln * 2), phase((size_t) c->rows * nblk * 2);
        for (int r = 0; r < c->rows; ++r)
            sdrt_build_nco(cfg->row_freqs[r], cfg->fs, Ln, nblk, &tile[(size_t) r * Ln * 2],
                           &phase[(size_t) r * nblk * 2]);
        rc = upload(&c->d_nco_tile, tile.data(), tile.size() * sizeof(double));
        if (rc == SDRT_OK) rc = upload(&c->d_nco_phase, phase.data(), phase.size() * sizeof(double));
    }
    if (rc == SDRT_OK && mixed) {
        for (int md = SDRT_FM; md <= SDRT_AM && rc == SDRT_OK; ++md) {
            std::vector<int> rows;
            for (int r = 0; r < cfg->n_rows; ++r) if (cfg->row_modes[r] == md) rows.push_back(r);
            c->n_rows_of[md] = (int) rows.size();
            rc = upload(&c->d_rows_of[md], rows.data(), rows.size() * sizeof(int));
        }
        if (rc == SDRT_OK) {
            std::vector<int> rp((size_t) cfg->n_rows);
            for (int r = 0; r < cfg->n_rows; ++r) rp[r] = cfg->row_modes[r] == SDRT_FM ? 1 : 0;
            rc = upload(&c->d_row_pair, rp.data(), rp.size() * sizeof(int));
        }
    }
    if (rc == SDRT_OK && any_fm) {
        std::vector<double> h((size_t) m / 2);
        sdrt_build_resample_odd(m / 2, h.data());
        rc = upload(&c->d_resample, h.data(), h.size() * sizeof(double));
        const int N = m / 2;
        const bool resample_direct = [] { const char *v = sdrt_diag("resample"); return v && strcmp(v, "direct") == 0; }();     // diagnostics: the circulant product
        if (rc == SDRT_OK && (N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096 || N == 8192 || N == 16384) && !resample_direct) {
            std::vector<double> tw((size_t) N * 2), sh((size_t) N * 2);
            sdrt_build_fft_resample(N, tw.data(), sh.data());
            rc = upload(&c->d_fft_tw, tw.data(), tw.size() * sizeof(double));
            if (rc == SDRT_OK) rc = upload(&c->d_fft_shift, sh.data(), sh.size() * sizeof(double));
            c->fft_ok = rc == SDRT_OK;
            // (diagnostics: post_packed=0 keeps the full-size complex transforms)
            if (rc == SDRT_OK && (N == 1024 || N == 2048 || N == 4096 || N == 8192 || N == 16384) && sdrt_diag_int("post_packed", 1) != 0) {
                // 2048 / 4096 outputs per chunk (decimation 16 / 8 of 16-bit records): the real angles packed two to a complex number
                std::vector<double> twp((size_t) N), ab((size_t) N * 2);
                sdrt_build_fft_resample_packed(N, twp.data(), ab.data());
                rc = upload(&c->d_fftp_tw, twp.data(), twp.size() * sizeof(double));
                if (rc == SDRT_OK) rc = upload(&c->d_fftp_ab, ab.data(), ab.size() * sizeof(double));
            }
        }
        // FM with 512 outputs per chunk (decimation 64 of 16-bit records, 32 of float32 records) and the usual output lowpass
        // (one complex pole pair + one real pole): eight lanes per item (diagnostics: SDRTERM_DIAG=post=wave keeps one wave per item)
        const bool post_wave = [] { const char *v = sdrt_diag("post"); return v && strcmp(v, "wave") == 0; }();
        const bool fft_wanted = !resample_direct;      // (256 outputs per chunk: this kernel only, the one-wave FFT form starts at 512)
        if (rc == SDRT_OK && fft_wanted && (m == 256 || m == 512 || m == 1024) && !post_wave) {
            const int lpi = m / 64;                            // 4, 8 or 16 lanes per item
            Post8Tables p8;
            std::vector<double> ft((size_t) (4 * SDRT_P8_PPL * lpi + 8 * lpi) * 2);
            if (sdrt_build_post8_tables(cfg->out_sos, cfg->out_sections, N, lpi, &p8, ft.data()) == 0) {
                rc = upload(&c->d_fft8, ft.data(), ft.size() * sizeof(double));
                if (rc == SDRT_OK) rc = upload(&c->d_p8tab, &p8, sizeof(p8));
                c->p8_lpi = lpi;
            }
        }
    }
    if (rc == SDRT_OK && f32c) {
        F32cArgs &fa = c->f32c_args;
        memset(&fa, 0, sizeof(fa));
        memcpy(fa.sos, cfg->f32_sos, sizeof(fa.sos));
        memcpy(fa.zi, cfg->f32_zi, sizeof(fa.zi));
        if (c->nco) rc = upload(&c->d_shift_tab, cfg->shift_table, (size_t) c->rows * n * 2 * sizeof(double));
        if (rc == SDRT_OK && c->nco && c->rows % 64 == 0) {
            // lane = row for the kernel that recomputes the shifted samples: 64 rows' entries of one time step side by side
            std::vector<double> t((size_t) c->rows * n * 2);
            const double *src = cfg->shift_table;
            for (int r = 0; r < c->rows; ++r)
                for (int k = 0; k < n; ++k) {
                    const size_t o = (((size_t) (r / 64) * n + k) * 64 + (r % 64)) * 2, i = ((size_t) r * n + k) * 2;
                    t[o] = src[i];
                    t[o + 1] = src[i + 1];
                }
            rc = upload(&c->d_shift_tab_t, t.data(), t.size() * sizeof(double));
        }
    }
    if (rc == SDRT_OK && !f32c) {
        // sdrt_config.stage_a; the environment variable overrides it (diagnostics: auto / i8 / f64, and i8t = the
        // one-block-per-column 16x16x64 kernel where it applies)
        int stage = cfg->stage_a;
        const char *force = getenv("SDRTERM_STAGE_A");
        if (force && strcmp(force, "f64") == 0) stage = SDRT_STAGE_F64;
        else if (force && (strcmp(force, "i8") == 0 || strcmp(force, "i8t") == 0)) stage = SDRT_STAGE_I8;
        else if (force && strcmp(force, "auto") == 0) stage = SDRT_STAGE_AUTO;
        c->stage = stage;
        const bool want_i8 = stage != SDRT_STAGE_F64;
        const bool want_t = force && strcmp(force, "i8t") == 0;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) c->n_cus = cus;
        if (const char *e = sdrt_diag("sched")) c->sched = atoi(e);
        if (const char *e = sdrt_diag("wide_dual")) c->dual = atoi(e);
        if (const char *e = sdrt_diag("wide_cus")) { const int v = atoi(e); if (v >= 8 && v <= c->n_cus) c->n_cus = v / 8 * 8; }   // diagnostics: fewer persistent workgroups
        // int8 matrix-core forms in order of preference; a form that cannot represent this configuration (unequal
        // power-of-two scales between rows, a digit overflow) releases what it uploaded and the next one is tried; the
        // fp64 matrix-core kernel takes whatever is left
        // one row: the offsets come out of the wide kernel's own matrix rows (no second pass over the input); several rows share
        // one prefix pass instead of each recomputing them (diagnostics: SDRTERM_DIAG=iq_fused=0 / 1 forces either)
        c->iq_fused = c->iq && c->rows == 1;
        if (const char *e = sdrt_diag("iq_fused")) c->iq_fused = c->iq && atoi(e) != 0;
        // decimation 2 .. 16 of 16-bit records: ONE pass on the matrix cores (kernels_sq.hip; diagnostics: sq=0 keeps the passes of the
        // wide kernel / the per-sample kernel)
        if (want_i8 && !want_t && !c->iq && chain_sq_supported(q, cfg->dtype, n) && sdrt_diag_int("sq", 1) != 0) {
            rc = setup_sq(c, cfg, tab);
            if (c->use_sq) c->use_seq = false;
        }
        if (c->use_sq || rc != SDRT_OK) {
            // (set up above)
        } else if (seq && !(want_i8 && !c->iq && c->norm_bias == 0.0 && chain_i8w_supported(q, cfg->dtype, n, c->iq))) {
            // nothing to set up: the per-sample form reads DecimTables only
        } else if (want_i8 && !want_t && chain_i8w_supported(q, cfg->dtype, n, c->iq) && chain_i8w_lds_bytes(q, n, c->iq ? (c->iq_fused ? 2 : 1) : 0, cfg->dtype) <= 160 * 1024)
            rc = setup_wide(c, cfg, scale_tables);
        if (rc == SDRT_OK && !seq && !c->use_i8w && want_i8 && !c->iq && want_i8_fallback_t(L, cfg->dtype, n, m))
            rc = setup_i8t(c, cfg, scale_tables);
        c->use_i8 = c->use_i8w || c->use_i8t;
        if (seq && c->use_i8w) c->use_seq = false;               // decimation 4 (and, SDRTERM_DIAG=wide_smallq=1, 2) as 8 (16) passes of the wide kernel
        // float32 on the wide kernel with several rows: every row of a chunk would slice the same records again (the larger part
        // of that form's vector work); from four rows up one pass slices them once (diagnostics: SDRTERM_DIAG=f32_pre=0 / 1)
        c->f32_pre = c->use_i8w && cfg->dtype == SDRT_F32 && c->rows >= 4;
        if (const char *e = sdrt_diag("f32_pre")) c->f32_pre = c->use_i8w && cfg->dtype == SDRT_F32 && atoi(e) != 0;
    }
    if (rc == SDRT_OK && !f64_fits && !c->use_i8w && !c->use_sq)
        rc = fail(SDRT_EINVAL, "chunk needs %zu bytes of LDS (> 160 KiB) on the fp64 kernel and the int8 form does not apply: n_samples/decimation combination unsupported", lds);
    if (rc == SDRT_OK && c->use_i8w && c->stage == SDRT_STAGE_AUTO && !c->iq && !mixed && f64_fits && c->more_passes.empty() && any_fm
        && (cfg->dtype == SDRT_S8 || cfg->dtype == SDRT_U8 || cfg->dtype == SDRT_S16 || cfg->dtype == SDRT_U16)) {
        // an item is looked at again when half of its pair products lie under (2^-15 of the encoding's full scale)^2 -- an in-band
        // level of ~1.4 LSB at 16 bits: there the 2^-38-of-the-peak image of the operator can reach 1e-6 of an angle (DESIGN section
        // 1, "Conditioning"); anything a real receiver delivers is far above it and never listed
        const double t = std::ldexp(1.0, 8 * kItemSize[cfg->dtype] - 16) * c->norm_scale;
        c->refine_thr = 2.0 * t * t;
        c->refine = sdrt_diag_int("refine", 1) != 0;
    }
    if (rc == SDRT_OK && c->use_i8) rc = setup_ybias(c, cfg);
    if (rc == SDRT_OK && c->use_i8 && !c->more_passes.empty()) rc = setup_pass_ybias(c);
    if (rc == SDRT_OK && c->use_i8) rc = calibrate_zero_response(c);
    if (rc != SDRT_OK) {
        sdrt_destroy(c);
        return rc;
    }
    *out = c;
    return SDRT_OK;
}

void sdrt_destroy(sdrt_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->cfg.device);
    (void) hipDeviceSynchronize();
    for (auto &p : c->pending) { (void) hipEventDestroy(p.a); (void) hipEventDestroy(p.b); }
    for (auto &e : c->free_events) (void) hipEventDestroy(e);
    (void) hipFree(c->d_tab);
    (void) hipFree(c->d_nco_tile);
    (void) hipFree(c->d_nco_phase);
    (void) hipFree(c->d_resample);
    (void) hipFree(c->d_smooth);
    (void) hipFree(c->d_fft_tw);
    (void) hipFree(c->d_fft_shift);
    (void) hipFree(c->d_fftp_tw);
    (void) hipFree(c->d_fftp_ab);
    (void) hipFree(c->d_fft8);
    (void) hipFree(c->d_p8tab);
    (void) hipFree(c->d_stash);
    (void) hipFree(c->d_rows_of[0]);
    (void) hipFree(c->d_rows_of[1]);
    (void) hipFree(c->d_row_pair);
    (void) hipFree(c->d_out_sos);
    (void) hipFree(c->d_otab);
    (void) hipFree(c->d_y);
    (void) hipFree(c->d_w8);
    (void) hipFree(c->d_w8mid);
    (void) hipFree(c->d_tab8);
    (void) hipFree(c->d_wtab);
    (void) hipFree(c->d_edge_ab);
    (void) hipFree(c->d_edge_pq);
    (void) hipFree(c->d_gvec);
    (void) hipFree(c->d_nco_tile8);
    (void) hipFree(c->d_nco_phase8);
    (void) hipFree(c->d_ybias);
    for (auto &wp : c->more_passes) { (void) hipFree(wp.d_wtab); (void) hipFree(wp.d_w8mid); (void) hipFree(wp.d_gvec); (void) hipFree(wp.d_ybias); }
    (void) hipFree(c->d_wsum);
    (void) hipFree(c->d_pre);
    (void) hipFree(c->d_pre_exp);
    (void) hipFree(c->d_pair_raw);
    (void) hipFree(c->d_pair_y);
    (void) hipFree(c->d_w8iq);
    (void) hipFree(c->d_iq_consts);
    (void) hipFree(c->d_iq_R);
    (void) hipFree(c->d_iq_state);
    (void) hipFree(c->d_iq_pow);
    (void) hipFree(c->d_iq_colw);
    (void) hipFree(c->d_iq_group);
    (void) hipFree(c->d_iq_agg);
    (void) hipFree(c->d_iq_chunk);
    (void) hipFree(c->d_iq_range);
    (void) hipFree(c->d_sqtab);
    (void) hipFree(c->d_sq_share);
    (void) hipFree(c->d_sq_agg);
    (void) hipFree(c->d_sq_master);
    (void) hipFree(c->d_sq_samp);
    (void) hipFree(c->d_sq_col);
    (void) hipFree(c->d_cond_list);
    (void) hipFree(c->d_cond_count);
    (void) hipFree(c->d_shift_tab);
    (void) hipFree(c->d_shift_tab_t);
    (void) hipFree(c->d_f32c_work);
    for (auto &sl : c->hs) {
        (void) hipFree(sl.d_raw);
        (void) hipFree(sl.d_out);
        if (sl.h_out) (void) hipHostFree(sl.h_out);
        if (sl.e_in) { (void) hipEventDestroy(sl.e_in); (void) hipEventDestroy(sl.e_run); (void) hipEventDestroy(sl.e_out); }
    }
    if (c->hs_h2d) { (void) hipStreamDestroy(c->hs_h2d); (void) hipStreamDestroy(c->hs_run); (void) hipStreamDestroy(c->hs_d2h); }
    delete c;
}

int sdrt_set_smoothing(sdrt_ctx *c, int window, int ntaps, int tap_first, const double *taps, int n_edge, const double *left,
                       const double *right) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    (void) hipFree(c->d_smooth);
    c->d_smooth = nullptr;
    c->sm_window = 0;
    if (window == 0) return SDRT_OK;
    if (window < 0 || ntaps < 1 || n_edge < 0 || !taps || (n_edge > 0 && (!left || !right)))
        return fail(SDRT_EINVAL, "bad smoothing operator");
    // scipy: "If mode is 'interp', window_length must be less than or equal to the size of x."
    if (window > c->m || 2 * n_edge > c->m)
        return fail(SDRT_EINVAL, "If mode is 'interp', window_length must be less than or equal to the size of x.");
    if (smooth_waves_per_block(c->m) == 0)
        return fail(SDRT_EINVAL, "--smooth-output: %d outputs per chunk exceed what the smoothing stage holds in LDS", c->m);
    const size_t cnt = (size_t) ntaps + 2 * (size_t) n_edge * window;
    std::vector<double> all(cnt);
    memcpy(all.data(), taps, sizeof(double) * ntaps);
    if (n_edge) {
        memcpy(all.data() + ntaps, left, sizeof(double) * n_edge * window);
        memcpy(all.data() + ntaps + (size_t) n_edge * window, right, sizeof(double) * n_edge * window);
    }
    int rc = upload(&c->d_smooth, all.data(), cnt * sizeof(double));
    if (rc != SDRT_OK) return rc;
    c->sm_window = window;
    c->sm_ntaps = ntaps;
    c->sm_first = tap_first;
    c->sm_edge = n_edge;
    return SDRT_OK;
}

int sdrt_out_per_chunk(const sdrt_ctx *c) { return c ? c->m : -1; }
int sdrt_rows(const sdrt_ctx *c) { return c ? c->rows : -1; }

namespace {

// scratch for the decimated samples between the two stages
int ensure_ybuf(sdrt_ctx *c, size_t n_chunks) {
    if (n_chunks > c->y_cap_chunks) {
        (void) hipFree(c->d_y);
        c->d_y = nullptr;
        c->y_cap_chunks = 0;
        HIP_TRY(hipMalloc(&c->d_y, n_chunks * (size_t) c->rows * c->m * 2 * sizeof(double)));
        c->y_cap_chunks = n_chunks;
    }
    return SDRT_OK;
}

// the IQ prefix tables of a batch (grow-only, like the scratch above)
int ensure_iqbufs(sdrt_ctx *c, size_t n_chunks) {
    if (!c->iq || n_chunks <= c->iq_cap_chunks) return SDRT_OK;
    (void) hipFree(c->d_iq_group);
    (void) hipFree(c->d_iq_agg);
    (void) hipFree(c->d_iq_chunk);
    c->d_iq_group = c->d_iq_agg = c->d_iq_chunk = nullptr;
    c->iq_cap_chunks = 0;
    if (!(c->use_i8w && c->iq_fused))
        HIP_TRY(hipMalloc(&c->d_iq_group, n_chunks * (size_t) (c->cfg.n_samples / c->iq_gs) * 2 * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_iq_agg, n_chunks * 2 * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_iq_chunk, ((n_chunks + 1) * 2 + SDRT_IQ_SCAN_PARTS * 4) * sizeof(double)));      // (+ the scan's scratch)
    c->iq_cap_chunks = n_chunks;
    return SDRT_OK;
}

int ensure_iqrange(sdrt_ctx *c, size_t n_chunks) {
    if (n_chunks <= c->iq_range_cap) return SDRT_OK;
    (void) hipFree(c->d_iq_range);
    c->d_iq_range = nullptr;
    c->iq_range_cap = 0;
    HIP_TRY(hipMalloc(&c->d_iq_range, (2 * n_chunks + 2) * 2 * sizeof(double)));
    c->iq_range_cap = n_chunks;
    return SDRT_OK;
}

// stage_a = auto: the list of items to look at again (one entry per (chunk, row) item at most)
int ensure_refine(sdrt_ctx *c, size_t n_chunks) {
    if (!c->refine || n_chunks <= c->refine_cap) return SDRT_OK;
    (void) hipFree(c->d_cond_list);
    c->d_cond_list = nullptr;
    c->refine_cap = 0;
    HIP_TRY(hipMalloc(&c->d_cond_list, n_chunks * (size_t) c->rows * sizeof(int)));
    if (!c->d_cond_count) {
        HIP_TRY(hipMalloc(&c->d_cond_count, 2 * sizeof(unsigned)));
        HIP_TRY(hipMemset(c->d_cond_count, 0, 2 * sizeof(unsigned)));
    }
    c->refine_cap = n_chunks;
    return SDRT_OK;
}

// float32 in the reference's arithmetic: the odd-extended complex64 sequences of a batch, 64 to a group
int ensure_f32c(sdrt_ctx *c, size_t n_chunks) {
    if (!c->f32c || n_chunks <= c->f32c_cap) return SDRT_OK;
    (void) hipFree(c->d_f32c_work);
    c->d_f32c_work = nullptr;
    c->f32c_cap = 0;
    HIP_TRY(hipMalloc(&c->d_f32c_work, f32c_work_bytes(c->cfg.n_samples, c->rows, n_chunks)));
    c->f32c_cap = n_chunks;
    return SDRT_OK;
}

// float32, several rows: the pre-sliced copy of a batch
int ensure_prebufs(sdrt_ctx *c, size_t n_chunks) {
    if (!c->f32_pre || n_chunks <= c->pre_cap) return SDRT_OK;
    (void) hipFree(c->d_pre);
    (void) hipFree(c->d_pre_exp);
    c->d_pre = nullptr;
    c->d_pre_exp = nullptr;
    c->pre_cap = 0;
    HIP_TRY(hipMalloc(&c->d_pre, n_chunks * (size_t) c->cfg.n_samples * 2 * sizeof(float)));
    HIP_TRY(hipMalloc(&c->d_pre_exp, n_chunks * 4 * sizeof(int)));
    c->pre_cap = n_chunks;
    return SDRT_OK;
}

// the scratch pair of the wide kernel's four-tile chunks (a single chunk is doubled into a pair)
int ensure_pairbufs(sdrt_ctx *c) {
    if (c->d_pair_raw || !(c->use_i8w && c->cfg.n_samples == 4 * SDRT_TB * SDRT_WIDE_LC)) return SDRT_OK;
    const size_t cb = (size_t) c->cfg.n_samples * 2 * kItemSize[c->cfg.dtype];
    HIP_TRY(hipMalloc(&c->d_pair_raw, 2 * cb));
    HIP_TRY(hipMalloc(&c->d_pair_y, (size_t) c->rows * 2 * (size_t) c->m * 2 * sizeof(double)));
    return SDRT_OK;
}

// unpack .. decimate (feedBuffers' per-sample steps, shiftFreq, decimate: read_file.py:100-103, dsp_processor.py:144-147)
// fix_later: the share of the IQ offset at each chunk's start (wide-column kernel, fused form) is subtracted by the post
// kernel while it loads the decimated samples (enqueue_post), not by a pass of its own
// pair_out: the wide-column kernel writes the products y[2i] conj(y[2i+1]) instead of the samples (FM, PostArgs.pairs)
int enqueue_chain(sdrt_ctx *c, const void *raw_dev, size_t n_chunks, double *ybuf, hipStream_t s, int n_cus, bool fix_later, bool pair_out,
                  bool row_pairs) {
    if (c->use_sq) {
        SqArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.raw = raw_dev;
        sa.ydec = ybuf;
        sa.n_chunks = n_chunks;
        sa.n = c->cfg.n_samples;
        sa.m = c->m;
        sa.rows = c->rows;
        sa.swap = c->cfg.swap_endianness;
        sa.pair_out = pair_out ? 1 : 0;
        sa.tab = c->d_sqtab;
        sa.share = c->d_sq_share;
        sa.agg = c->d_sq_agg;
        sa.master = c->d_sq_master;
        sa.nco_samp = c->d_sq_samp;
        sa.nco_col = c->d_sq_col;
        sa.ybias = c->d_ybias;
        sa.comb_agg = c->sq_comb_agg;
        sa.comb_loc = c->sq_comb_loc;
        sa.stamps = c->stamps_dev;
        sa.stamp_wgs = c->stamp_wgs;
        {
            // workgroups of an XCD started apart in phase (kernels_sq.hip): the last one starts 31 steps late, so the step is sized to
            // keep that under ~3 % of what a workgroup has to do in this launch (a chunk costs it ~110 / 60 / 45 / 35 thousand cycles
            // at decimation 2 / 4 / 8 / 16), and 16 sleeps of 64 cycles at most (8 .. 32 measure alike: r05_experiments.txt section 9)
            const char *e = sdrt_diag("sq_stagger");
            const int q = c->cfg.decimation;
            const size_t wpr = std::max<size_t>(1, (size_t) n_cus / (size_t) c->rows);
            const double per_chunk = q == 2 ? 110e3 : q == 4 ? 60e3 : q == 8 ? 45e3 : 35e3;
            const double fit = 0.03 * per_chunk * (double) (n_chunks / wpr) / (31.0 * 64.0);
            sa.stagger = e ? atoi(e) : (int) std::min(16.0, fit);
            // the touches of the next chunk double the input's HBM reads (an XCD's L2 does not keep a round of chunks: counters in
            // r05_experiments.txt section 11) and buy 2 % at decimation 16, nothing at 2 .. 8
            sa.touch = sdrt_diag_int("sq_touch", c->cfg.decimation >= 16 ? 1 : 0);
        }
        Timed t(c, s, 0);
        HIP_TRY(launch_chain_sq(sa, c->cfg.decimation, c->cfg.dtype, n_cus, s));
        return SDRT_OK;
    }
    if (c->f32c) {
        int rcf = ensure_f32c(c, n_chunks);
        if (rcf != SDRT_OK) return rcf;
        F32cArgs fa = c->f32c_args;
        fa.raw = raw_dev;
        fa.shift = c->nco ? c->d_shift_tab : nullptr;
        fa.shift_t = c->nco ? c->d_shift_tab_t : nullptr;
        fa.work = c->d_f32c_work;
        fa.ydec = ybuf;
        fa.n_chunks = n_chunks;
        fa.n = c->cfg.n_samples;
        fa.m = c->m;
        fa.q = c->cfg.decimation;
        fa.rows = c->rows;
        fa.swap = c->cfg.swap_endianness;
        Timed t(c, s, 0);
        HIP_TRY(launch_f32c(fa, s));
        return SDRT_OK;
    }
    DecimArgs da;
    memset(&da, 0, sizeof(da));
    da.raw = raw_dev;
    da.n_count = 0;
    da.tab = c->d_tab;
    da.nco_tile = c->d_nco_tile;
    da.nco_phase = c->d_nco_phase;
    da.ydec = ybuf;
    da.stamps = c->stamps_dev;
    da.stamp_wgs = c->stamp_wgs;
    da.pair_out = pair_out ? 1 : 0;
    da.row_pair = row_pairs ? c->d_row_pair : nullptr;      // per-row modes: FM rows leave as pair products, each item in its samples' place
    da.pair_wide = row_pairs ? 1 : 0;
    da.n_chunks = n_chunks;
    da.n = c->cfg.n_samples;
    da.m = c->m;
    da.n_tiles = c->n_tiles;
    da.rows = c->rows;
    da.swap = c->cfg.swap_endianness && kItemSize[c->cfg.dtype] > 1;
    da.norm_bias = c->norm_bias;
    da.iq_group_off = nullptr;
    da.iq_chunk_off = nullptr;
    da.iq_pow = c->d_iq_pow;
    da.iq_lambda = c->cfg.iq_inductance;
    IqArgs iq_after;
    memset(&iq_after, 0, sizeof(iq_after));
    if (c->iq) {
        const int gs = c->iq_gs;
        int rcq = ensure_iqbufs(c, n_chunks);
        if (rcq != SDRT_OK) return rcq;
        IqArgs ia;
        ia.raw = raw_dev;
        ia.group_off = c->d_iq_group;
        ia.chunk_agg = c->d_iq_agg;
        ia.chunk_off = c->d_iq_chunk;
        ia.state = c->d_iq_state;
        ia.scan_part = c->d_iq_chunk + (c->iq_cap_chunks + 1) * 2;
        ia.n_chunks = n_chunks;
        ia.n = c->cfg.n_samples;
        ia.gs = gs;
        ia.dt = c->cfg.dtype;
        ia.swap = da.swap;
        // wide-column kernel: offsets at the starts of its 256-sample columns and at the tail window
        ia.wmask = c->use_i8w ? SDRT_WIDE_LC / gs - 1 : 0;
        ia.wtail = (c->cfg.n_samples - 32) / gs;
        ia.norm_bias = c->norm_bias;
        ia.lambda = c->cfg.iq_inductance;
        ia.pow_tab = c->d_iq_pow;
        ia.pow_g = c->iq_pow_e;
        ia.pow_n = c->iq_pow_n;
        const char *colenv = sdrt_diag("iq_cols");            // diagnostics: "0" keeps the per-group prefix kernel
        ia.col_w = c->use_i8w && !(colenv && strcmp(colenv, "0") == 0) ? c->d_iq_colw : nullptr;
        if (!(c->use_i8w && c->iq_fused)) {
            // prefix pass over the input, then the chunks chained: the kernels below take offsets from these tables
            Timed t(c, s, 2);
            HIP_TRY(launch_iq_local(ia, s));
            HIP_TRY(launch_iq_scan(ia, s));
            da.iq_group_off = c->d_iq_group;
            da.iq_chunk_off = c->d_iq_chunk;
        }
        iq_after = ia;            // wide-column kernel: it produces the chunk aggregates itself; scan + offset share follow it
    }
    da.w8iq = c->iq && c->use_i8w && c->iq_fused ? c->d_w8iq : nullptr;
    da.iq_consts = c->d_iq_consts;
    da.iq_combine = c->iq_combine;
    da.iq_chunk_agg = c->d_iq_agg;
    da.w8 = c->d_w8;
    da.w8mid = c->d_w8mid;
    da.ybias = c->d_ybias;
    da.w8_combine = c->w8_combine;
    da.w8mid_combine = c->w8mid_combine;
    da.x_offset = c->x_offset;
    da.wsum = c->d_wsum;
    da.wtab = c->d_wtab;
    da.edge_ab = c->d_edge_ab;
    da.edge_pq = c->d_edge_pq;
    da.gvec = c->d_gvec;
    da.e_off_re = c->iq ? -c->norm_bias : c->x_offset;      // IQ: the edge operators see the true samples
    da.e_off_im = c->iq ? 0.0 : c->x_offset;
    da.c_re = c->x_offset + c->norm_bias;
    da.c_im = c->x_offset;
    da.iq_gs = c->iq_gs;
    da.rs = c->use_i8w && c->wide_rs;
    da.raw_orig = nullptr;
    da.f32_exp = nullptr;
    da.sched = c->sched;
    da.dual = c->dual;
    da.out_mul = c->use_i8w ? SDRT_WIDE_LC / c->cfg.decimation : 0;       // outputs per column (all passes)
    da.out_off = 0;
    da.m_local = c->m / (1 + (int) c->more_passes.size());
    if (c->use_i8) {
        da.tab = c->d_tab8;
        da.n_tiles = c->n_tiles8;
        da.nco_tile = c->d_nco_tile8;
        da.nco_phase = c->d_nco_phase8;
    }
    {
        Timed t(c, s, 0);
        if (c->use_seq) {
            HIP_TRY(launch_chain_seq(da, c->cfg.decimation, c->cfg.dtype, c->nco, c->iq, n_chunks, s));
        } else if (c->use_i8w && c->cfg.n_samples == 4 * SDRT_TB * SDRT_WIDE_LC) {
            // chunks of four tiles go two to an item: an odd batch ends with the pair (n-2, n-1) once more (chunk n-2 comes out the
            // same twice); a single chunk is doubled into a scratch pair
            const size_t cb = (size_t) c->cfg.n_samples * 2 * kItemSize[c->cfg.dtype], yb = (size_t) c->m * (pair_out ? 1 : 2);
            if (n_chunks == 1) {
                int rcp = ensure_pairbufs(c);
                if (rcp != SDRT_OK) return rcp;
                HIP_TRY(hipMemcpyAsync(c->d_pair_raw, raw_dev, cb, hipMemcpyDeviceToDevice, s));
                HIP_TRY(hipMemcpyAsync((char *) c->d_pair_raw + cb, raw_dev, cb, hipMemcpyDeviceToDevice, s));
                DecimArgs d2 = da;
                d2.raw = c->d_pair_raw;
                d2.ydec = c->d_pair_y;
                d2.n_chunks = 2;
                HIP_TRY(launch_chain_i8w(d2, c->cfg.decimation, c->cfg.dtype, 2, n_cus, s));
                HIP_TRY(hipMemcpy2DAsync(ybuf, yb * sizeof(double), c->d_pair_y, 2 * yb * sizeof(double), yb * sizeof(double), (size_t) c->rows,
                                         hipMemcpyDeviceToDevice, s));
            } else {
                if (c->f32_pre) {
                    int rcp = ensure_prebufs(c, n_chunks);
                    if (rcp != SDRT_OK) return rcp;
                    HIP_TRY(launch_f32_preslice(raw_dev, c->d_pre, c->d_pre_exp, n_chunks * 4, da.swap, s));
                    da.raw = c->d_pre;
                    da.raw_orig = raw_dev;
                    da.f32_exp = c->d_pre_exp;
                }
                da.n_count = n_chunks & ~(size_t) 1;
                HIP_TRY(launch_chain_i8w(da, c->cfg.decimation, c->cfg.dtype, da.n_count, n_cus, s));
                if (n_chunks & 1) {
                    DecimArgs d2 = da;
                    d2.raw = (const char *) da.raw + (n_chunks - 2) * cb;
                    if (da.f32_exp) { d2.raw_orig = (const char *) raw_dev + (n_chunks - 2) * cb; d2.f32_exp = da.f32_exp + (n_chunks - 2) * 4; }
                    d2.ydec = ybuf + (n_chunks - 2) * yb;
                    d2.n_count = 2;
                    HIP_TRY(launch_chain_i8w(d2, c->cfg.decimation, c->cfg.dtype, 2, n_cus, s));
                }
            }
        } else if (c->use_i8w) {
            HIP_TRY(launch_chain_i8w(da, c->cfg.decimation, c->cfg.dtype, n_chunks, n_cus, s));
            // decimation 8 / 16: the further passes, eight more outputs of every column each
            for (size_t p = 0; p < c->more_passes.size(); ++p) {
                const sdrt_ctx::WidePass &wp = c->more_passes[p];
                DecimArgs dp = da;
                dp.wtab = wp.d_wtab;
                dp.w8mid = wp.d_w8mid;
                dp.gvec = wp.d_gvec;
                dp.ybias = wp.d_ybias;
                dp.w8mid_combine = wp.w8mid_combine;
                dp.out_off = (int) (p + 1) * c->nout_pass;
                HIP_TRY(launch_chain_i8w(dp, c->cfg.decimation, c->cfg.dtype, n_chunks, n_cus, s));
            }
        } else if (c->use_i8t) {
            // the int8 kernels feed x - x_offset to the matrix cores; normalisation bias and offset come back via ybias
            HIP_TRY(launch_chain_i8t(da, c->L, c->cfg.dtype, n_chunks, s));
        } else {
            HIP_TRY(launch_chain(da, c->L, c->cfg.dtype, c->nco, c->iq, n_chunks, s));
        }
    }
    if (c->iq && c->use_i8w && c->iq_fused) {
        Timed t(c, s, 2);
        HIP_TRY(launch_iq_scan(iq_after, s));
        if (!fix_later) HIP_TRY(launch_iq_fix(ybuf, c->d_iq_R, c->d_iq_chunk, c->rows, n_chunks, c->m, s));
    }
    return SDRT_OK;
}

// ybias -= (the stage-A kernel this context will run, applied to one all-zero chunk): see setup_ybias.  The kernel's bits for a
// chunk do not depend on the batch around it (tests: the batch-size and determinism cases, zero chunks scattered over a batch).
int calibrate_zero_response(sdrt_ctx *c) {
    // (--correct-iq: the offset term carries the constant and those forms take no ybias)
    // Signed 8-bit records have no slice offset, but with --normalize-input their ybias holds the bias response in the whole
    // chunk's layout: several passes still need their own slices of it (b, normalised, decimation 8 / 16 read pass 0's
    // slice with the wrong stride and zeros in the further passes)
    if (!c->use_i8 || c->iq || !c->d_ybias || (c->x_offset == 0.0 && c->more_passes.empty())) return SDRT_OK;
    const size_t cb = (size_t) c->cfg.n_samples * 2 * kItemSize[c->cfg.dtype];
    const size_t cnt = (size_t) c->rows * c->m * 2;
    void *d_zero = nullptr;
    double *d_r0 = nullptr;
    int rc = SDRT_OK;
    if (hipMalloc(&d_zero, cb) != hipSuccess || hipMalloc(&d_r0, cnt * sizeof(double)) != hipSuccess) rc = fail(SDRT_ENOMEM, "hipMalloc failed");
    std::vector<double> bias(cnt), r0(cnt);
    if (rc == SDRT_OK && hipMemset(d_zero, 0, cb) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemset failed");
    if (rc == SDRT_OK && hipMemcpy(bias.data(), c->d_ybias, cnt * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemcpy failed");
    if (rc == SDRT_OK && hipMemset(c->d_ybias, 0, cnt * sizeof(double)) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemset failed");
    if (rc == SDRT_OK) rc = enqueue_chain(c, d_zero, 1, d_r0, nullptr, c->n_cus);
    if (rc == SDRT_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(SDRT_EHIP, "zero-response kernel failed");
    if (rc == SDRT_OK && hipMemcpy(r0.data(), d_r0, cnt * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemcpy failed");
    if (rc == SDRT_OK) {
        for (size_t i = 0; i < cnt; ++i) bias[i] -= r0[i];
        const size_t np = 1 + c->more_passes.size();
        if (np == 1) {
            if (hipMemcpy(c->d_ybias, bias.data(), cnt * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemcpy failed");
        } else {
            // output k of the chunk = column k / nout_total, output b = k % nout_total of it: pass b / nout_pass holds it at
            // column * nout_pass + b % nout_pass of its own slice
            const size_t mt = (size_t) c->m, ml = mt / np, no = (size_t) c->nout_pass, nt = no * np;
            std::vector<double> part((size_t) c->rows * ml * 2);
            for (size_t p = 0; p < np && rc == SDRT_OK; ++p) {
                for (size_t r = 0; r < (size_t) c->rows; ++r)
                    for (size_t k = 0; k < ml; ++k) {
                        const size_t col = k / no, b = k % no, kt = col * nt + p * no + b;
                        part[(r * ml + k) * 2] = bias[(r * mt + kt) * 2];
                        part[(r * ml + k) * 2 + 1] = bias[(r * mt + kt) * 2 + 1];
                    }
                double *dst = p == 0 ? c->d_ybias : c->more_passes[p - 1].d_ybias;
                if (hipMemcpy(dst, part.data(), part.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(SDRT_EHIP, "hipMemcpy failed");
            }
        }
    }
    (void) hipFree(d_zero);
    (void) hipFree(d_r0);
    return rc;
}

// demodulate, resample, output lowpass, --smooth-output (dsp_processor.py:148-149,159-160)
// the demodulation stage's arguments; iq_share: it also subtracts the offset share the fused IQ form leaves to it
PostArgs post_args(sdrt_ctx *c, const double *ybuf, size_t n_chunks, double *out_dev, bool iq_share, bool pairs = false) {
    PostArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.y = ybuf;
    pa.resample_odd = c->d_resample;
    pa.fft_tw = c->d_fft_tw;
    pa.fft_shift = c->d_fft_shift;
    pa.fft_ok = c->fft_ok && c->out_modal;
    pa.fftp_tw = c->d_fftp_tw;
    pa.fftp_ab = c->d_fftp_ab;
    pa.otab = c->d_otab;
    pa.out_sos = c->d_out_sos;
    pa.out = out_dev;
    pa.items = n_chunks * (size_t) c->rows;
    pa.m = c->m;
    pa.mode = c->cfg.mode;
    pa.out_sections = c->cfg.out_sections;
    pa.iq_R = iq_share ? c->d_iq_R : nullptr;
    pa.iq_off = iq_share ? c->d_iq_chunk : nullptr;
    pa.n_chunks = n_chunks;
    pa.pairs = pairs ? 1 : 0;
    pa.row_map = nullptr;
    pa.fft8_tab = c->d_fft8;
    pa.p8tab = c->d_p8tab;
    pa.p8_lpi = c->p8_lpi;
    pa.stash = nullptr;
    pa.stash_bytes = 0;
    return pa;
}

// the angle stash of the eight / sixteen-lanes-per-item demodulation stage for a batch of this size
int ensure_stash(sdrt_ctx *c, PostArgs &pa) {
    const char *e = sdrt_diag("post_stash");              // diagnostics: "out" keeps the angles in the output buffer
    if (e && strcmp(e, "out") == 0) return SDRT_OK;
    const size_t need = post_fft8_stash_bytes(pa);
    if (need > c->stash_cap) {
        (void) hipFree(c->d_stash);
        c->d_stash = nullptr;
        c->stash_cap = 0;
        HIP_TRY(hipMalloc(&c->d_stash, need));
        c->stash_cap = need;
    }
    if (need) { pa.stash = c->d_stash; pa.stash_bytes = c->stash_cap; }
    return SDRT_OK;
}

int enqueue_post(sdrt_ctx *c, const double *ybuf, size_t n_chunks, double *out_dev, hipStream_t s, bool iq_share = false, bool pairs = false,
                 bool row_pairs = false, const void *raw_for_refine = nullptr) {
    PostArgs pa = post_args(c, ybuf, n_chunks, out_dev, iq_share, pairs);
    if (!c->mixed) { int rcs = ensure_stash(c, pa); if (rcs != SDRT_OK) return rcs; }
    // stage_a = auto: the pair-product form of the eight / sixteen-lanes-per-item stage lists what deserves a second look
    const bool refine = c->refine && pairs && !iq_share && !row_pairs && post_fft8_supported(pa);
    if (refine) {
        int rcr = ensure_refine(c, n_chunks);
        if (rcr != SDRT_OK) return rcr;
        HIP_TRY(hipMemsetAsync(c->d_cond_count, 0, sizeof(unsigned), s));
        pa.cond_list = c->d_cond_list;
        pa.cond_count = c->d_cond_count;
        pa.cond_thr = c->refine_thr;
    }
    {
        Timed t(c, s, 1);
        if (c->mixed) {
            // per-row modes: one launch per mode over that mode's rows (PostArgs.row_map); the rows keep their places in y / out
            for (int md = SDRT_FM; md <= SDRT_AM; ++md) {
                PostArgs pm = pa;
                pm.mode = md;
                pm.row_map = c->d_rows_of[md];
                pm.items = n_chunks * (size_t) c->n_rows_of[md];
                if (md == SDRT_FM) {
                    if (row_pairs) { pm.pairs = 1; pm.pairs_wide = 1; }
                    int rcs = ensure_stash(c, pm);
                    if (rcs != SDRT_OK) return rcs;
                }
                HIP_TRY(launch_post(pm, s));
            }
        } else
        HIP_TRY(launch_post(pa, s));
        if (c->sm_window > 0) {
            SmoothArgs sa;
            sa.out = out_dev;
            sa.items = pa.items;
            sa.m = c->m;
            sa.window = c->sm_window;
            sa.ntaps = c->sm_ntaps;
            sa.tap_first = c->sm_first;
            sa.n_edge = c->sm_edge;
            sa.taps = c->d_smooth;
            sa.left = c->d_smooth + c->sm_ntaps;
            sa.right = sa.left + (size_t) c->sm_edge * c->sm_window;
            HIP_TRY(launch_smooth(sa, s));
        }
    }
    if (refine) {
        // the listed items again: unpack .. decimate in float64 (the fp64 matrix cores), pair products over the first try's, then
        // the same demodulation stage over them.  Usually nothing is listed and the two launches return at once.
        DecimArgs da;
        memset(&da, 0, sizeof(da));
        da.raw = raw_for_refine;
        da.tab = c->d_tab;
        da.nco_tile = c->d_nco_tile;
        da.nco_phase = c->d_nco_phase;
        da.ydec = const_cast<double *>(ybuf);
        da.n_chunks = n_chunks;
        da.n = c->cfg.n_samples;
        da.m = c->m;
        da.n_tiles = c->n_tiles;
        da.rows = c->rows;
        da.swap = c->cfg.swap_endianness && kItemSize[c->cfg.dtype] > 1;
        da.norm_bias = c->norm_bias;
        da.fix_list = c->d_cond_list;
        da.fix_count = c->d_cond_count;
        da.fix_pairs = 1;
        // (not counted among the timed kernels: with nothing listed they are two empty launches, and a launch count that doubled
        // would halve the averages bench.py reports)
        HIP_TRY(launch_chain(da, c->L, c->cfg.dtype, c->nco, false, n_chunks, s));
        PostArgs pf = pa;
        pf.cond_list = nullptr;
        pf.fix_list = c->d_cond_list;
        pf.fix_count = c->d_cond_count;
        pf.items = pa.items < 2048 ? pa.items : 2048;      // launch geometry only: the kernel reads the count itself
        HIP_TRY(launch_post(pf, s));
    }
    return SDRT_OK;
}

int check_process_args(sdrt_ctx *c, const void *raw_dev, size_t n_chunks, const double *out_dev) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    if (!raw_dev || !out_dev) return fail(SDRT_EINVAL, "null buffer");
    if (((uintptr_t) raw_dev & 15) != 0) return fail(SDRT_EINVAL, "raw_dev must be 16-byte aligned");
    if (n_chunks * (size_t) c->rows > 0x7fffffffull) return fail(SDRT_EINVAL, "too many chunks in one call");
    return SDRT_OK;
}

}  // namespace

int sdrt_process_device_debug(sdrt_ctx *c, const void *raw_dev, size_t n_chunks, double *out_dev, double *ydec_dev,
                              void *stream) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    if (n_chunks == 0) return SDRT_OK;
    int rc = check_process_args(c, raw_dev, n_chunks, out_dev);
    if (rc != SDRT_OK) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    hipStream_t s = (hipStream_t) stream;
    double *ybuf = ydec_dev;
    if (!ybuf) {
        rc = ensure_ybuf(c, n_chunks);
        if (rc != SDRT_OK) return rc;
        ybuf = c->d_y;
    }
    // fused IQ form: when the caller does not ask for the decimated samples, the demodulation stage subtracts the offset
    // share as it loads them (one pass over them less)
    bool later = !ydec_dev && c->iq && c->use_i8w && c->iq_fused && post_takes_iq_share(post_args(c, ybuf, n_chunks, out_dev, true));
    if (later && c->mixed) {                                  // per-row modes: both modes' kernels must take the share
        PostArgs pm = post_args(c, ybuf, n_chunks, out_dev, true);
        pm.mode = SDRT_AM;
        later = post_takes_iq_share(pm);
    }
    // FM on the wide-column kernel, likewise without a debug buffer: the pair products instead of the samples (half the bytes
    // between the two stages); not with the offset share still to come
    // (several passes over the chunk: a pass finishes eight consecutive outputs of every column, whole pairs -- SDRTERM_DIAG=pass_pairs=0: samples)
    const bool pass_pairs = sdrt_diag_int("pass_pairs", 1) != 0;
    const bool pairs = !ydec_dev && (c->use_i8w || c->use_sq) && (c->more_passes.empty() || pass_pairs) && !c->stamps_dev && !later && !c->mixed && post_takes_pairs(post_args(c, ybuf, n_chunks, out_dev, false, true));
    // per-row modes: the FM rows' items as pair products, every item in the place its samples would take (eight- and sixteen-tile
    // chunks; the two-chunks-per-item forms keep samples)
    bool row_pairs = false;
    if (!ydec_dev && c->use_i8w && c->more_passes.empty() && !c->stamps_dev && !later && c->mixed && c->d_row_pair && c->n_rows_of[SDRT_FM] > 0
        && c->cfg.n_samples != 4 * SDRT_TB * SDRT_WIDE_LC) {
        PostArgs pm = post_args(c, ybuf, n_chunks, out_dev, false, true);
        pm.mode = SDRT_FM;
        row_pairs = post_takes_pairs(pm);
    }
    rc = enqueue_chain(c, raw_dev, n_chunks, ybuf, s, c->n_cus, later, pairs, row_pairs);
    if (rc != SDRT_OK) return rc;
    return enqueue_post(c, ybuf, n_chunks, out_dev, s, later, pairs, row_pairs, raw_dev);
}

int sdrt_reserve(sdrt_ctx *c, size_t max_chunks) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    if (max_chunks == 0) return SDRT_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    int rc = ensure_ybuf(c, max_chunks);
    if (rc == SDRT_OK) rc = ensure_iqbufs(c, max_chunks);
    if (rc == SDRT_OK && c->iq) rc = ensure_iqrange(c, max_chunks);
    if (rc == SDRT_OK) rc = ensure_pairbufs(c);
    if (rc == SDRT_OK) rc = ensure_prebufs(c, max_chunks);
    if (rc == SDRT_OK) rc = ensure_f32c(c, max_chunks);
    if (rc == SDRT_OK) rc = ensure_refine(c, max_chunks);
    if (rc == SDRT_OK) {
        PostArgs pa = post_args(c, nullptr, max_chunks, nullptr, false, true);
        if (c->mixed) {                                       // per-row modes: the FM rows' launch is the one that keeps a stash
            pa.mode = SDRT_FM;
            pa.items = max_chunks * (size_t) c->n_rows_of[SDRT_FM];
        }
        rc = ensure_stash(c, pa);
    }
    return rc;
}

int sdrt_process_device(sdrt_ctx *c, const void *raw_dev, size_t n_chunks, double *out_dev, void *stream) {
    return sdrt_process_device_debug(c, raw_dev, n_chunks, out_dev, nullptr, stream);
}

/*
 * Host path (read_file.py:119-125 -> Queue -> dsp_processor.py:164-183): three slots of device buffers + page-locked
 * output staging and three streams, so that the copy-in of batch k+1 runs beside the kernels of batch k and the copy-out of
 * batch k-1.  Kernels of consecutive batches stay in order on one stream (the IQ offset and the per-context scratch are
 * sequential).  Slots grow when a larger batch arrives and are reused otherwise: no allocation on the steady-state path.
 */
namespace {

int host_slot_prepare(sdrt_ctx *c, HostSlot &sl, size_t n_chunks, size_t in_bytes, size_t out_bytes) {
    if (!c->hs_h2d) {
        HIP_TRY(hipStreamCreateWithFlags(&c->hs_h2d, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&c->hs_run, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&c->hs_d2h, hipStreamNonBlocking));
    }
    if (!sl.e_in) {
        HIP_TRY(hipEventCreateWithFlags(&sl.e_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.e_run, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.e_out, hipEventDisableTiming));
    }
    if (n_chunks > sl.cap_chunks) {
        if (sl.d_raw) HIP_TRY(hipFree(sl.d_raw));
        if (sl.d_out) HIP_TRY(hipFree(sl.d_out));
        if (sl.h_out) HIP_TRY(hipHostFree(sl.h_out));
        sl.d_raw = nullptr; sl.d_out = nullptr; sl.h_out = nullptr; sl.cap_chunks = 0;
        HIP_TRY(hipMalloc(&sl.d_raw, n_chunks * in_bytes));
        HIP_TRY(hipMalloc(&sl.d_out, n_chunks * out_bytes));
        HIP_TRY(hipHostMalloc((void **) &sl.h_out, n_chunks * out_bytes, hipHostMallocDefault));
        sl.cap_chunks = n_chunks;
    }
    return SDRT_OK;
}

// the slot's batch is finished: results into the caller's buffer (unless the copy-out went there directly)
int host_slot_finish(HostSlot &sl) {
    if (!sl.busy) return SDRT_OK;
    HIP_TRY(hipEventSynchronize(sl.e_out));
    if (!sl.direct_out) memcpy(sl.out_host, sl.h_out, sl.out_bytes);
    sl.busy = false;
    return SDRT_OK;
}

bool is_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void) hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

}  // namespace

int sdrt_submit_host(sdrt_ctx *c, const void *raw_host, size_t n_chunks, double *out_host, int64_t *ticket) {
    if (!c || !ticket) return fail(SDRT_EINVAL, "null argument");
    if (!raw_host || !out_host || n_chunks == 0) return fail(SDRT_EINVAL, "null or empty buffer");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t in_bytes = (size_t) c->cfg.n_samples * 2 * kItemSize[c->cfg.dtype];
    const size_t out_bytes = (size_t) c->rows * c->m * sizeof(double);
    const int64_t t = c->hs_next++;
    HostSlot &sl = c->hs[t % SDRT_HOST_SLOTS];
    int rc = host_slot_finish(sl);                      // the ring has come round: that batch's results are due now
    if (rc == SDRT_OK) rc = host_slot_prepare(c, sl, n_chunks, in_bytes, out_bytes);
    if (rc != SDRT_OK) return rc;
    // copy-in: asynchronous from page-locked memory (sdrt_host_alloc); from pageable memory the runtime stages the copy and
    // this call returns when it is done -- the previous batch's kernels run meanwhile either way
    HIP_TRY(hipMemcpyAsync(sl.d_raw, raw_host, n_chunks * in_bytes, hipMemcpyHostToDevice, c->hs_h2d));
    HIP_TRY(hipEventRecord(sl.e_in, c->hs_h2d));
    HIP_TRY(hipStreamWaitEvent(c->hs_run, sl.e_in, 0));
    rc = sdrt_process_device(c, sl.d_raw, n_chunks, sl.d_out, c->hs_run);
    if (rc != SDRT_OK) return rc;
    HIP_TRY(hipEventRecord(sl.e_run, c->hs_run));
    HIP_TRY(hipStreamWaitEvent(c->hs_d2h, sl.e_run, 0));
    sl.direct_out = is_pinned(out_host);
    sl.out_host = out_host;
    sl.out_bytes = n_chunks * out_bytes;
    HIP_TRY(hipMemcpyAsync(sl.direct_out ? (void *) out_host : (void *) sl.h_out, sl.d_out, sl.out_bytes, hipMemcpyDeviceToHost, c->hs_d2h));
    HIP_TRY(hipEventRecord(sl.e_out, c->hs_d2h));
    sl.busy = true;
    sl.ticket = t;
    *ticket = t;
    return SDRT_OK;
}

int sdrt_wait(sdrt_ctx *c, int64_t ticket) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    if (ticket < 0 || ticket >= c->hs_next) return fail(SDRT_EINVAL, "unknown ticket");
    HostSlot &sl = c->hs[ticket % SDRT_HOST_SLOTS];
    if (!sl.busy || sl.ticket != ticket) return SDRT_OK;          // already delivered (the ring came round, or waited before)
    HIP_TRY(hipSetDevice(c->cfg.device));
    return host_slot_finish(sl);
}

int sdrt_process_host(sdrt_ctx *c, const void *raw_host, size_t n_chunks, double *out_host) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    if (n_chunks == 0) return SDRT_OK;
    int64_t t = -1;
    int rc = sdrt_submit_host(c, raw_host, n_chunks, out_host, &t);
    if (rc != SDRT_OK) return rc;
    return sdrt_wait(c, t);
}

int sdrt_get_iq_offset(sdrt_ctx *c, double *re, double *im) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    double v[2] = {0.0, 0.0};
    if (c->d_iq_state) {
        HIP_TRY(hipSetDevice(c->cfg.device));
        HIP_TRY(hipMemcpy(v, c->d_iq_state, sizeof(v), hipMemcpyDeviceToHost));
    }
    if (re) *re = v[0] * c->norm_scale;      // device state is in un-scaled units (see sdrt_create)
    if (im) *im = v[1] * c->norm_scale;
    return SDRT_OK;
}

int sdrt_set_iq_offset(sdrt_ctx *c, double re, double im) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    if (!c->d_iq_state) return fail(SDRT_EINVAL, "context was created without correct_iq");
    const double v[2] = {re / c->norm_scale, im / c->norm_scale};
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpy(c->d_iq_state, v, sizeof(v), hipMemcpyHostToDevice));
    return SDRT_OK;
}

int sdrt_iq_range_device(sdrt_ctx *c, const void *raw_dev, size_t n_chunks, double *a, double *b_re, double *b_im, void *stream) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    if (!c->iq) return fail(SDRT_EINVAL, "context was created without correct_iq");
    if (!raw_dev && n_chunks) return fail(SDRT_EINVAL, "null input");
    HIP_TRY(hipSetDevice(c->cfg.device));
    hipStream_t s = (hipStream_t) stream;
    double v[2] = {0.0, 0.0};
    if (n_chunks) {
        // chunk aggregates from a zero start (no group table), chained from a zero state: the end value is b
        int rcr = ensure_iqrange(c, n_chunks);                 // [n_chunks] aggregates, [n_chunks + 1] offsets, [1] state
        if (rcr != SDRT_OK) return rcr;
        double *d_tmp = c->d_iq_range;
        IqArgs ia;
        memset(&ia, 0, sizeof(ia));
        ia.raw = raw_dev;
        ia.group_off = nullptr;
        ia.chunk_agg = d_tmp;
        ia.chunk_off = d_tmp + 2 * n_chunks;
        ia.state = d_tmp + 2 * (2 * n_chunks + 1);
        ia.n_chunks = n_chunks;
        ia.n = c->cfg.n_samples;
        ia.gs = c->iq_gs;
        ia.dt = c->cfg.dtype;
        ia.swap = c->cfg.swap_endianness && kItemSize[c->cfg.dtype] > 1;
        ia.norm_bias = c->norm_bias;
        ia.lambda = c->cfg.iq_inductance;
        ia.pow_tab = c->d_iq_pow;
        ia.pow_g = c->iq_pow_e;
        ia.pow_n = c->iq_pow_n;
        hipError_t e = hipMemsetAsync(ia.state, 0, 2 * sizeof(double), s);
        if (e == hipSuccess) e = launch_iq_local(ia, s);
        if (e == hipSuccess) e = launch_iq_scan(ia, s);
        if (e == hipSuccess) e = hipMemcpyAsync(v, ia.state, sizeof(v), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(SDRT_EHIP, "offset range kernels failed: %s", hipGetErrorString(e));
    }
    if (a) *a = exp((double) n_chunks * (double) c->cfg.n_samples * log1p(-c->cfg.iq_inductance));
    if (b_re) *b_re = v[0] * c->norm_scale;
    if (b_im) *b_im = v[1] * c->norm_scale;
    return SDRT_OK;
}

int sdrt_profile_enable(sdrt_ctx *c, int enable) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    c->profile = enable != 0;
    return SDRT_OK;
}

int sdrt_kernel_time(sdrt_ctx *c, int which, double *total_ms, int64_t *launches) {
    if (!c || which < 0 || which > 2) return fail(SDRT_EINVAL, "bad argument");
    int rc = resolve_pending(c);
    if (rc != SDRT_OK) return rc;
    if (total_ms) *total_ms = c->ms[which];
    if (launches) *launches = c->launches[which];
    return SDRT_OK;
}

int sdrt_profile_reset(sdrt_ctx *c) {
    if (!c) return fail(SDRT_EINVAL, "null context");
    int rc = resolve_pending(c);
    for (int i = 0; i < 3; ++i) { c->ms[i] = 0; c->launches[i] = 0; }
    return rc;
}

const char *sdrt_kernel_name(int which) {
    switch (which) {
        case 0: return "chain_kernel / chain_i8_kernel (unpack..decimate)";
        case 1: return "post_kernel (demodulate, resample, output lowpass)";
        case 2: return "iq_prefix_kernels";
    }
    return "";
}

int sdrt_malloc(void **p, size_t bytes) {
    if (!p) return fail(SDRT_EINVAL, "null argument");
    HIP_TRY(hipMalloc(p, bytes ? bytes : 1));
    return SDRT_OK;
}
int sdrt_free(void *p) {
    HIP_TRY(hipFree(p));
    return SDRT_OK;
}
int sdrt_memcpy_h2d(void *d, const void *s, size_t b) {
    HIP_TRY(hipMemcpy(d, s, b, hipMemcpyHostToDevice));
    return SDRT_OK;
}
int sdrt_memcpy_d2h(void *d, const void *s, size_t b) {
    HIP_TRY(hipMemcpy(d, s, b, hipMemcpyDeviceToHost));
    return SDRT_OK;
}
int sdrt_memcpy_d2d(void *d, const void *s, size_t b) {
    HIP_TRY(hipMemcpy(d, s, b, hipMemcpyDeviceToDevice));
    return SDRT_OK;
}
unsigned long long sdrt_alloc_count(void) { return g_dev_allocs.load(std::memory_order_relaxed); }
int sdrt_mem_info(size_t *free_bytes, size_t *total_bytes) {
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return SDRT_OK;
}
int sdrt_device_synchronize(void) {
    HIP_TRY(hipDeviceSynchronize());
    return SDRT_OK;
}

int sdrt_stream_create(void **stream) {
    if (!stream) return fail(SDRT_EINVAL, "null argument");
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *) s;
    return SDRT_OK;
}
int sdrt_stream_destroy(void *stream) {
    HIP_TRY(hipStreamDestroy((hipStream_t) stream));
    return SDRT_OK;
}
int sdrt_stream_synchronize(void *stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    return SDRT_OK;
}
int sdrt_host_alloc(void **p, size_t bytes) {
    if (!p) return fail(SDRT_EINVAL, "null argument");
    HIP_TRY(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return SDRT_OK;
}
int sdrt_host_free(void *p) {
    HIP_TRY(hipHostFree(p));
    return SDRT_OK;
}
int sdrt_memcpy_h2d_async(void *d, const void *s, size_t b, void *stream) {
    HIP_TRY(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, (hipStream_t) stream));
    return SDRT_OK;
}
int sdrt_memcpy_d2h_async(void *d, const void *s, size_t b, void *stream) {
    HIP_TRY(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, (hipStream_t) stream));
    return SDRT_OK;
}
int sdrt_stream_wait_stream(void *waiter, void *signaller) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t r = hipEventRecord(e, (hipStream_t) signaller);
    if (r == hipSuccess) r = hipStreamWaitEvent((hipStream_t) waiter, e, 0);
    (void) hipEventDestroy(e);          // released once the recorded work has completed
    if (r != hipSuccess) return fail(SDRT_EHIP, "stream dependency failed: %s", hipGetErrorString(r));
    return SDRT_OK;
}

int sdrt_event_create(void **event) {
    if (!event) return fail(SDRT_EINVAL, "null argument");
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *) e;
    return SDRT_OK;
}
int sdrt_event_create_timed(void **event) {
    if (!event) return fail(SDRT_EINVAL, "null argument");
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreate(&e));
    *event = (void *) e;
    return SDRT_OK;
}
int sdrt_event_elapsed_ms(void *start, void *stop, double *ms) {
    if (!start || !stop || !ms) return fail(SDRT_EINVAL, "null argument");
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, (hipEvent_t) start, (hipEvent_t) stop));
    *ms = (double) t;
    return SDRT_OK;
}
int sdrt_memcpy_d2d_async(void *dst_dev, const void *src_dev, size_t bytes, void *stream) {
    if (bytes == 0) return SDRT_OK;
    if (!dst_dev || !src_dev) return fail(SDRT_EINVAL, "null buffer");
    HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    return SDRT_OK;
}
int sdrt_event_destroy(void *event) {
    HIP_TRY(hipEventDestroy((hipEvent_t) event));
    return SDRT_OK;
}
int sdrt_event_record(void *event, void *stream) {
    HIP_TRY(hipEventRecord((hipEvent_t) event, (hipStream_t) stream));
    return SDRT_OK;
}
int sdrt_stream_wait_event(void *stream, void *event) {
    HIP_TRY(hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0));
    return SDRT_OK;
}
int sdrt_event_query(void *event, int *done) {
    if (!done) return fail(SDRT_EINVAL, "null argument");
    const hipError_t e = hipEventQuery((hipEvent_t) event);
    if (e == hipSuccess) { *done = 1; return SDRT_OK; }
    if (e == hipErrorNotReady) { (void) hipGetLastError(); *done = 0; return SDRT_OK; }
    return fail(SDRT_EHIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}
int sdrt_event_synchronize(void *event) {
    HIP_TRY(hipEventSynchronize((hipEvent_t) event));
    return SDRT_OK;
}

int sdrt_demodulate_host(int mode, const double *y_host, int rows, int m, double *out_host) {
    if (!y_host || !out_host || rows < 1 || m < 1 || mode < 0 || mode > 3) return fail(SDRT_EINVAL, "bad argument");
    if (mode == 0 && (m & 1)) return fail(SDRT_EINVAL, "FM needs an even number of samples per row");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SDRT_ENODEVICE, "no HIP device available");
    double *dy = nullptr, *dout = nullptr, *dh = nullptr;
    const size_t ybytes = (size_t) rows * m * 2 * sizeof(double), obytes = (size_t) rows * m * sizeof(double);
    HIP_TRY(hipMalloc(&dy, ybytes));
    HIP_TRY(hipMalloc(&dout, obytes));
    HIP_TRY(hipMemcpy(dy, y_host, ybytes, hipMemcpyHostToDevice));
    if (mode == 0) {
        std::vector<double> h((size_t) m / 2);
        sdrt_build_resample_odd(m / 2, h.data());
        HIP_TRY(hipMalloc(&dh, h.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(dh, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    hipError_t e = launch_demod(dy, dout, dh, rows, m, mode, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, dout, obytes, hipMemcpyDeviceToHost);
    (void) hipFree(dy);
    (void) hipFree(dout);
    (void) hipFree(dh);
    if (e != hipSuccess) return fail(SDRT_EHIP, "demodulate failed: %s", hipGetErrorString(e));
    return SDRT_OK;
}

namespace {

// device buffers of the tap entry points: kept from call to call and only ever grown (the plots call once per frame,
// plots/abstract_plot.py: every 17 ms), per calling thread and device; sdrt_taps_release frees them
struct TapSlot {
    void *p = nullptr;
    size_t cap = 0;
    int need(size_t bytes) {
        if (bytes <= cap) return SDRT_OK;
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
        HIP_TRY(hipMalloc(&p, bytes));
        cap = bytes;
        return SDRT_OK;
    }
    int put(const void *src, size_t bytes) {
        int rc = need(bytes);
        if (rc != SDRT_OK) return rc;
        HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return SDRT_OK;
    }
    void drop() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
};
// A thread's tap buffers go with the thread (a plot thread that ends must not leave its device memory behind: ADVICE r3) -- but
// not at process exit, where thread-local destructors can run after the HIP runtime has torn itself down: an atexit handler,
// registered with the first buffers, raises a flag and the memory is left to the process's end.
std::atomic<bool> g_process_exiting{false};
struct TapBufs {
    TapSlot raw, u, scratch, out, tw, tile, phase, win;
    int dev = -1;
    TapBufs() {
        static std::once_flag once;
        std::call_once(once, [] { std::atexit([] { g_process_exiting.store(true); }); });
    }
    ~TapBufs() {
        if (g_process_exiting.load()) return;
        for (TapSlot *t : {&raw, &u, &scratch, &out, &tw, &tile, &phase, &win}) t->drop();
    }
};
thread_local TapBufs g_taps;

int tap_common(int dtype, int swap, const void *raw_host, int n, int chunks, double shift_hz, double fs, int normalize, double norm_xmin,
               double norm_inv_range, TapBufs &b, TapArgs &ta) {
    if (dtype < 0 || dtype > 7 || !raw_host || n < 256 || (n & (n - 1)) || chunks < 1 || !(fs > 0))
        return fail(SDRT_EINVAL, "bad argument (n must be a power of two >= 256)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(SDRT_ENODEVICE, "no HIP device available");
    const size_t in_bytes = (size_t) chunks * n * 2 * kItemSize[dtype], cbytes = (size_t) chunks * n * 2 * sizeof(double);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (b.dev != dev) {                     // buffers belong to the device they were allocated on
        for (TapSlot *t : {&b.raw, &b.u, &b.scratch, &b.out, &b.tw, &b.tile, &b.phase, &b.win}) t->drop();
        b.dev = dev;
    }
    int rc0 = b.raw.put(raw_host, in_bytes);
    if (rc0 == SDRT_OK) rc0 = b.u.need(cbytes);
    if (rc0 != SDRT_OK) return rc0;
    memset(&ta, 0, sizeof(ta));
    ta.raw = b.raw.p;
    ta.u = (double *) b.u.p;
    ta.chunks = (size_t) chunks;
    ta.n = n;
    ta.swap = swap && kItemSize[dtype] > 1;
    ta.normalize = normalize;
    if (normalize) {                        // read_file.py:96: 1.6 (z - xmin) d - 0.8 with complex z: only Re is offset
        ta.norm_scale = 1.6 * norm_inv_range;
        ta.norm_bias = -norm_xmin - 0.5 / norm_inv_range;
    }
    if (shift_hz != 0.0) {
        std::vector<double> tile(64 * 2), phase((size_t) (n / 64) * 2);
        sdrt_build_nco(shift_hz, fs, 64, n / 64, tile.data(), phase.data());
        int rc = b.tile.put(tile.data(), tile.size() * sizeof(double));
        if (rc == SDRT_OK) rc = b.phase.put(phase.data(), phase.size() * sizeof(double));
        if (rc != SDRT_OK) return rc;
        ta.nco_tile = (const double *) b.tile.p;
        ta.nco_phase = (const double *) b.phase.p;
    }
    HIP_TRY(launch_tap_decode(ta, dtype, nullptr));
    return SDRT_OK;
}

int tap_twiddles(int N, TapBufs &b) {
    std::vector<double> tw((size_t) N * 2), unused((size_t) N * 2);
    sdrt_build_fft_resample(N, tw.data(), unused.data());
    return b.tw.put(tw.data(), tw.size() * sizeof(double));
}

}  // namespace

int sdrt_spectrum_host(int dtype, int swap_endianness, const void *raw_host, int n_samples, int n_chunks, double shift_hz, double fs,
                       int normalize, double norm_xmin, double norm_inv_range, double *out_host) {
    if (!out_host) return fail(SDRT_EINVAL, "null buffer");
    TapBufs &b = g_taps;
    TapArgs ta;
    int rc = tap_common(dtype, swap_endianness, raw_host, n_samples, n_chunks, shift_hz, fs, normalize, norm_xmin, norm_inv_range, b, ta);
    if (rc != SDRT_OK) return rc;
    const size_t cbytes = (size_t) n_chunks * n_samples * 2 * sizeof(double), obytes = (size_t) n_chunks * n_samples * sizeof(double);
    rc = b.scratch.need(cbytes);
    if (rc == SDRT_OK) rc = b.out.need(obytes);
    if (rc == SDRT_OK) rc = tap_twiddles(n_samples, b);
    if (rc != SDRT_OK) return rc;
    ta.scratch = (double *) b.scratch.p;
    ta.out = (double *) b.out.p;
    ta.tw = (const double *) b.tw.p;
    HIP_TRY(launch_tap_fft(ta, nullptr));
    HIP_TRY(hipMemcpy(out_host, b.out.p, obytes, hipMemcpyDeviceToHost));
    return SDRT_OK;
}

int sdrt_stft_host(int dtype, int swap_endianness, const void *raw_host, int n_samples, int n_chunks, double shift_hz, double fs,
                   int normalize, double norm_xmin, double norm_inv_range, int nperseg, int hop, int mfft, const double *window,
                   double *out_host) {
    if (!out_host || !window) return fail(SDRT_EINVAL, "null buffer");
    if (mfft != 1024 || nperseg < 2 || nperseg > 256 || hop < 1 || n_samples % hop)
        return fail(SDRT_EINVAL, "supported: mfft 1024, nperseg <= 256, hop dividing the chunk (the waterfall plot's 256 / 128 / 1024)");
    TapBufs &b = g_taps;
    TapArgs ta;
    int rc = tap_common(dtype, swap_endianness, raw_host, n_samples, n_chunks, shift_hz, fs, normalize, norm_xmin, norm_inv_range, b, ta);
    if (rc != SDRT_OK) return rc;
    const size_t obytes = (size_t) n_chunks * mfft * (size_t) (n_samples / hop + 1) * sizeof(double);
    rc = b.out.need(obytes);
    if (rc == SDRT_OK) rc = tap_twiddles(mfft, b);
    if (rc == SDRT_OK) rc = b.win.put(window, (size_t) nperseg * sizeof(double));
    if (rc != SDRT_OK) return rc;
    ta.out = (double *) b.out.p;
    ta.tw = (const double *) b.tw.p;
    ta.window = (const double *) b.win.p;
    ta.nperseg = nperseg;
    ta.hop = hop;
    HIP_TRY(launch_tap_stft(ta, nullptr));
    HIP_TRY(hipMemcpy(out_host, b.out.p, obytes, hipMemcpyDeviceToHost));
    return SDRT_OK;
}

int sdrt_taps_release(void) {
    TapBufs &b = g_taps;
    for (TapSlot *t : {&b.raw, &b.u, &b.scratch, &b.out, &b.tw, &b.tile, &b.phase, &b.win}) t->drop();
    b.dev = -1;
    return SDRT_OK;
}

int sdrt_debug_stamps(sdrt_ctx *c, const void *raw_dev, size_t n_chunks, double *out_dev, unsigned long long *stamps_host,
                      int n_wgs) {
    if (!c || !stamps_host || n_wgs < 1) return fail(SDRT_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t bytes = (size_t) n_wgs * 4 * 16 * sizeof(unsigned long long);
    HIP_TRY(hipMalloc(&c->stamps_dev, bytes));
    HIP_TRY(hipMemset(c->stamps_dev, 0, bytes));
    c->stamp_wgs = n_wgs;
    int rc = sdrt_process_device(c, raw_dev, n_chunks, out_dev, nullptr);
    hipError_t e = hipMemcpy(stamps_host, c->stamps_dev, bytes, hipMemcpyDeviceToHost);
    (void) hipFree(c->stamps_dev);
    c->stamps_dev = nullptr;
    c->stamp_wgs = 0;
    if (rc != SDRT_OK) return rc;
    if (e != hipSuccess) return fail(SDRT_EHIP, "stamp copy failed: %s", hipGetErrorString(e));
    return SDRT_OK;
}

int sdrt_refined_items(sdrt_ctx *c, unsigned long long *total) {
    if (!c || !total) return fail(SDRT_EINVAL, "null argument");
    *total = 0;
    if (!c->d_cond_count) return SDRT_OK;
    unsigned v[2] = {0, 0};
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(v, c->d_cond_count, sizeof(v), hipMemcpyDeviceToHost));
    *total = v[1];
    return SDRT_OK;
}

int sdrt_stage_a_kind(const sdrt_ctx *c) { return c ? (c->use_sq ? 6 : c->f32c ? 5 : c->use_seq ? 4 : c->use_i8w ? 3 : (c->use_i8t ? 2 : (c->use_i8 ? 1 : 0))) : -1; }

int sdrt_selftest_mfma_i8(int *mismatches) {
    int bad = 0;
    // 16x16x64 form
    {
        static int A2[16 * 64], B2[64 * 16], D2[256];
        int *eA = nullptr, *eB = nullptr, *eD = nullptr;
        for (int i = 0; i < 16; ++i)
            for (int k = 0; k < 64; ++k) {
                A2[i * 64 + k] = ((i * 11 + k * 3) % 29) - 14;
                B2[k * 16 + i] = ((k * 7 - i * 5 + 128) % 23) - 11;
            }
        HIP_TRY(hipMalloc(&eA, sizeof(A2)));
        HIP_TRY(hipMalloc(&eB, sizeof(B2)));
        HIP_TRY(hipMalloc(&eD, sizeof(D2)));
        HIP_TRY(hipMemcpy(eA, A2, sizeof(A2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(eB, B2, sizeof(B2), hipMemcpyHostToDevice));
        HIP_TRY(launch_i8t_selftest(eA, eB, eD, nullptr));
        HIP_TRY(hipMemcpy(D2, eD, sizeof(D2), hipMemcpyDeviceToHost));
        (void) hipFree(eA);
        (void) hipFree(eB);
        (void) hipFree(eD);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                int ref = 0;
                for (int k = 0; k < 64; ++k) ref += A2[i * 64 + k] * B2[k * 16 + j];
                if (ref != D2[i * 16 + j]) ++bad;
            }
    }
    // lane exchanges of the wide-column kernel's reduce-scatter (v_permlane32_swap / v_permlane16_swap)
    {
        double X[64], Y[64], O[128], *dX = nullptr, *dY = nullptr, *dO = nullptr;
        for (int l = 0; l < 64; ++l) { X[l] = 1.0 + l; Y[l] = 1000.0 + 3.0 * l; }
        HIP_TRY(hipMalloc(&dX, sizeof(X)));
        HIP_TRY(hipMalloc(&dY, sizeof(Y)));
        HIP_TRY(hipMalloc(&dO, sizeof(O)));
        HIP_TRY(hipMemcpy(dX, X, sizeof(X), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dY, Y, sizeof(Y), hipMemcpyHostToDevice));
        HIP_TRY(launch_swap_selftest(dX, dY, dO, nullptr));
        HIP_TRY(hipMemcpy(O, dO, sizeof(O), hipMemcpyDeviceToHost));
        (void) hipFree(dX);
        (void) hipFree(dY);
        (void) hipFree(dO);
        for (int l = 0; l < 64; ++l) {
            const double e32 = l < 32 ? X[l] + X[l + 32] : Y[l - 32] + Y[l];
            const int r = l >> 4;
            const double e16 = (r & 1) == 0 ? X[l] + X[l + 16] : Y[l - 16] + Y[l];
            if (O[l] != e32) ++bad;
            if (O[64 + l] != e16) ++bad;
        }
    }
    if (mismatches) *mismatches = bad;
    return bad == 0 ? SDRT_OK : fail(SDRT_EHIP, "int8 MFMA operand layout / lane swap mismatch (%d results wrong)", bad);
}

int sdrt_selftest_mfma(double *max_abs_err) {
    double A[64], B[64], D[256], *dA = nullptr, *dB = nullptr, *dD = nullptr;
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 4; ++k) A[i * 4 + k] = (double) (1 + i * 5 + k * 3);     // asymmetric integer data
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) B[k * 16 + j] = (double) (2 + k * 7 - j * 2);
    HIP_TRY(hipMalloc(&dA, sizeof(A)));
    HIP_TRY(hipMalloc(&dB, sizeof(B)));
    HIP_TRY(hipMalloc(&dD, sizeof(D)));
    HIP_TRY(hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice));
    HIP_TRY(launch_mfma_selftest(dA, dB, dD, nullptr));
    HIP_TRY(hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost));
    (void) hipFree(dA);
    (void) hipFree(dB);
    (void) hipFree(dD);
    double worst = 0.0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0.0;
            for (int k = 0; k < 4; ++k) ref += A[i * 4 + k] * B[k * 16 + j];
            worst = std::fmax(worst, std::fabs(ref - D[i * 16 + j]));
        }
    if (max_abs_err) *max_abs_err = worst;
    return worst == 0.0 ? SDRT_OK : fail(SDRT_EHIP, "MFMA f64 16x16x4 operand layout mismatch (max err %g)", worst);
}

}  // extern "C"
