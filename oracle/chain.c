/*
 * ORACLE (test infrastructure, NOT product code): plain-C fp64 restatement of the sdrterm DSP hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product (sdrterm_amd/) never links or calls it.  Each function restates one step of the reference
 * (peads/sdrterm @ 2024_08_07; paths relative to /root/reference) or of the scipy 1.15.3 routine the
 * reference calls at that step, and cites it.  Filter coefficients (cheby1 / ellip SOS) are inputs:
 * tests feed the tables captured from scipy in tests/golden/reference_vectors.npz.
 *
 * Pinned by tests/test_oracle_golden.py against golden vectors produced by running the real
 * reference (tests/golden/make_golden.py).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;

enum { ORC_S8 = 0, ORC_U8, ORC_S16, ORC_U16, ORC_S32, ORC_U32, ORC_F32, ORC_F64 };
enum { ORC_FM = 0, ORC_AM = 1 };

static const int k_item_size[8] = {1, 1, 2, 2, 4, 4, 4, 8};

int orc_item_size(int code) { return (code >= 0 && code < 8) ? k_item_size[code] : -1; }

/* src/misc/read_file.py:51,100-101 -- z = y['re'] + 1j*y['im'] from interleaved records. */
static double load_scalar(const unsigned char *p, int code, int swap) {
    unsigned char b[8];
    int sz = k_item_size[code];
    if (swap) {
        for (int i = 0; i < sz; ++i) b[i] = p[sz - 1 - i];
    } else {
        memcpy(b, p, (size_t) sz);
    }
    switch (code) {
        case ORC_S8: return (double) *(int8_t *) b;
        case ORC_U8: return (double) *(uint8_t *) b;
        case ORC_S16: { int16_t v; memcpy(&v, b, 2); return (double) v; }
        case ORC_U16: { uint16_t v; memcpy(&v, b, 2); return (double) v; }
        case ORC_S32: { int32_t v; memcpy(&v, b, 4); return (double) v; }
        case ORC_U32: { uint32_t v; memcpy(&v, b, 4); return (double) v; }
        case ORC_F32: { float v; memcpy(&v, b, 4); return (double) v; }
        default: { double v; memcpy(&v, b, 8); return v; }
    }
}

void orc_unpack(const void *raw, int code, int swap, size_t n, cplx *z) {
    const unsigned char *p = (const unsigned char *) raw;
    int sz = k_item_size[code];
    for (size_t i = 0; i < n; ++i) {
        double re = load_scalar(p + (2 * i) * sz, code, swap);
        double im = load_scalar(p + (2 * i + 1) * sz, code, swap);
        z[i] = re + im * I;
    }
}

/* src/misc/read_file.py:82-96 -- res = 1.6*(z - xmin)*xMaxMinDiff - 0.8 on the complex value. */
void orc_normalize(cplx *z, size_t n, double xmin, double inv_range) {
    for (size_t i = 0; i < n; ++i) {
        double re = 1.6 * (creal(z[i]) - xmin) * inv_range - 0.8;
        double im = 1.6 * cimag(z[i]) * inv_range;
        z[i] = re + im * I;
    }
}

/* src/misc/read_file.py:64-77 == extra/src/iq_correction.pyx:45-52.  off carries across chunks. */
void orc_correct_iq(cplx *z, size_t n, cplx *off, double inductance) {
    cplx o = *off;
    for (size_t i = 0; i < n; ++i) {
        z[i] -= o;
        o += z[i] * inductance;
    }
    *off = o;
}

/* dsp_processor.py:185-187 / vfo_processor.py:48,74 -- exp(w*arange(n)), w = -2j*pi*(f/fs). */
void orc_shift_table(double f, double fs, size_t n, cplx *tbl) {
    double wi = -2.0 * M_PI * (f / fs);
    for (size_t k = 0; k < n; ++k) {
        double a = wi * (double) k;
        tbl[k] = cos(a) + sin(a) * I;
    }
}

/* scipy/signal/_sosfilt.pyx (_sosfilt_float): transposed direct form II, per sample across sections. */
void orc_sosfilt_c(const double *sos, int nsec, cplx *x, size_t n, cplx *state /* nsec*2 */) {
    for (size_t i = 0; i < n; ++i) {
        cplx xc = x[i];
        for (int s = 0; s < nsec; ++s) {
            const double *c = sos + 6 * s;
            cplx xn = c[0] * xc + state[2 * s];
            state[2 * s] = c[1] * xc - c[4] * xn + state[2 * s + 1];
            state[2 * s + 1] = c[2] * xc - c[5] * xn;
            xc = xn;
        }
        x[i] = xc;
    }
}

void orc_sosfilt_r(const double *sos, int nsec, double *x, size_t n, double *state) {
    for (size_t i = 0; i < n; ++i) {
        double xc = x[i];
        for (int s = 0; s < nsec; ++s) {
            const double *c = sos + 6 * s;
            double xn = c[0] * xc + state[2 * s];
            state[2 * s] = c[1] * xc - c[4] * xn + state[2 * s + 1];
            state[2 * s + 1] = c[2] * xc - c[5] * xn;
            xc = xn;
        }
        x[i] = xc;
    }
}

/* scipy.signal.sosfilt_zi / lfilter_zi: steady-state step-response state, cascaded with section gains. */
void orc_sosfilt_zi(const double *sos, int nsec, double *zi /* nsec*2 */) {
    double scale = 1.0;
    for (int s = 0; s < nsec; ++s) {
        const double *c = sos + 6 * s;
        double b0 = c[0] / c[3], b1 = c[1] / c[3], b2 = c[2] / c[3], a1 = c[4] / c[3], a2 = c[5] / c[3];
        double B0 = b1 - a1 * b0, B1 = b2 - a2 * b0;
        double z0 = (B0 + B1) / (1.0 + a1 + a2);
        double z1 = (1.0 + a1) * z0 - B0;
        zi[2 * s] = scale * z0;
        zi[2 * s + 1] = scale * z1;
        scale *= (b0 + b1 + b2) / (1.0 + a1 + a2);
    }
}

/* scipy.signal.sosfiltfilt defaults: padtype='odd', padlen = 3*ntaps, ntaps = 2*nsec+1 - min(#b2==0, #a2==0). */
int orc_sosfiltfilt_edge(const double *sos, int nsec) {
    int zb = 0, za = 0;
    for (int s = 0; s < nsec; ++s) {
        if (sos[6 * s + 2] == 0.0) ++zb;
        if (sos[6 * s + 5] == 0.0) ++za;
    }
    int ntaps = 2 * nsec + 1 - (zb < za ? zb : za);
    return 3 * ntaps;
}

/* Workspace variant: ext must hold n + 2*edge values.  y must hold n values.
 * Returns 0, or -1 if n <= edge (scipy raises ValueError there). */
static int sosfiltfilt_ws(const double *sos, int nsec, const cplx *x, size_t n, cplx *y, cplx *ext) {
    int edge = orc_sosfiltfilt_edge(sos, nsec);
    if (n <= (size_t) edge) return -1;
    size_t N = n + 2 * (size_t) edge;
    double zi[32];
    cplx st[32];
    if (nsec > 16) return -2;
    /* odd_ext (scipy/signal/_arraytools.py): 2*x[0] - x[edge:0:-1] | x | 2*x[-1] - x[-2:-(edge+2):-1] */
    for (int i = 0; i < edge; ++i) ext[i] = 2.0 * x[0] - x[edge - i];
    memcpy(ext + edge, x, n * sizeof(cplx));
    for (int i = 0; i < edge; ++i) ext[edge + n + i] = 2.0 * x[n - 1] - x[n - 2 - i];
    orc_sosfilt_zi(sos, nsec, zi);
    for (int k = 0; k < 2 * nsec; ++k) st[k] = zi[k] * ext[0];
    orc_sosfilt_c(sos, nsec, ext, N, st);
    /* reverse, filter again with zi * y_0, reverse back */
    for (size_t i = 0; i < N / 2; ++i) { cplx t = ext[i]; ext[i] = ext[N - 1 - i]; ext[N - 1 - i] = t; }
    for (int k = 0; k < 2 * nsec; ++k) st[k] = zi[k] * ext[0];
    orc_sosfilt_c(sos, nsec, ext, N, st);
    for (size_t i = 0; i < n; ++i) y[i] = ext[N - 1 - edge - i];
    return 0;
}

int orc_sosfiltfilt(const double *sos, int nsec, const cplx *x, size_t n, cplx *y) {
    cplx *ext = (cplx *) malloc((n + 2 * 3 * (2 * (size_t) nsec + 1)) * sizeof(cplx));
    if (!ext) return -2;
    int rc = sosfiltfilt_ws(sos, nsec, x, n, y, ext);
    free(ext);
    return rc;
}

/* scipy.signal.decimate(x, q) defaults (dsp_processor.py:147): sosfiltfilt(cheby1 sos) then y[::q]. */
int orc_decimate(const double *sos, int nsec, const cplx *x, size_t n, int q, cplx *y /* ceil(n/q) */) {
    cplx *full = (cplx *) malloc(n * sizeof(cplx));
    if (!full) return -2;
    int rc = orc_sosfiltfilt(sos, nsec, x, n, full);
    if (rc == 0) {
        size_t k = 0;
        for (size_t i = 0; i < n; i += (size_t) q) y[k++] = full[i];
    }
    free(full);
    return rc;
}

/* --- FFT used only to restate scipy.signal.resample (demodulation.py:35-38) --------------------------- */
static void dft_any(const cplx *in, cplx *out, size_t n, int sign) {
    if (n == 1) { out[0] = in[0]; return; }
    if ((n & 1) == 0) {
        size_t h = n / 2;
        cplx *ev = (cplx *) malloc(n * sizeof(cplx)), *od = ev + h;
        cplx *t = (cplx *) calloc(n, sizeof(cplx));
        for (size_t i = 0; i < h; ++i) { t[i] = in[2 * i]; t[h + i] = in[2 * i + 1]; }
        dft_any(t, ev, h, sign);
        dft_any(t + h, od, h, sign);
        for (size_t k = 0; k < h; ++k) {
            double a = sign * 2.0 * M_PI * (double) k / (double) n;
            cplx w = cos(a) + sin(a) * I;
            out[k] = ev[k] + w * od[k];
            out[k + h] = ev[k] - w * od[k];
        }
        free(ev);
        free(t);
        return;
    }
    for (size_t k = 0; k < n; ++k) {
        cplx acc = 0;
        for (size_t j = 0; j < n; ++j) {
            double a = sign * 2.0 * M_PI * (double) ((k * j) % n) / (double) n;
            acc += in[j] * (cos(a) + sin(a) * I);
        }
        out[k] = acc;
    }
}

/* scipy.signal.resample(x, num) for real x, window=None, domain='time' (scipy/signal/_signaltools.py). */
int orc_resample(const double *x, size_t N, size_t num, double *out) {
    size_t big = N > num ? N : num;
    cplx *a = (cplx *) calloc(big, sizeof(cplx)), *X = (cplx *) calloc(big, sizeof(cplx));
    cplx *Y = (cplx *) calloc(big, sizeof(cplx)), *yy = (cplx *) calloc(big, sizeof(cplx));
    if (!a || !X || !Y || !yy) { free(a); free(X); free(Y); free(yy); return -2; }
    for (size_t i = 0; i < N; ++i) a[i] = x[i];
    dft_any(a, X, N, -1);                      /* X = rfft(x): bins 0..N/2 used */
    size_t m = (num < N ? num : N);
    size_t nyq = m / 2 + 1;                    /* Y[:nyq] = X[:nyq] */
    size_t half = num / 2 + 1;                 /* rfft length of the output */
    cplx *Yh = (cplx *) calloc(half, sizeof(cplx));
    for (size_t k = 0; k < nyq && k < half; ++k) Yh[k] = X[k];
    if ((m & 1) == 0) {
        if (num < N) {                         /* downsampling: fold the bin at m/2 */
            Yh[m / 2] *= 2.0;                  /* real input: X[-m/2] = conj(X[m/2]); scipy adds it -> 2*Re handled by irfft */
            Yh[m / 2] = creal(Yh[m / 2]);
        } else if (N < num) {                  /* upsampling: halve the bin at N/2 */
            Yh[m / 2] *= 0.5;
        }
    }
    /* irfft(Yh, num): build the Hermitian spectrum, inverse DFT, take the real part */
    for (size_t k = 0; k < half; ++k) Y[k] = Yh[k];
    for (size_t k = half; k < num; ++k) Y[k] = conj(Yh[num - k]);
    if ((num & 1) == 0) Y[num / 2] = creal(Yh[num / 2]);
    Y[0] = creal(Yh[0]);
    dft_any(Y, yy, num, +1);
    double scale = (1.0 / (double) num) * ((double) num / (double) N);
    for (size_t i = 0; i < num; ++i) out[i] = creal(yy[i]) * scale;
    free(a); free(X); free(Y); free(yy); free(Yh);
    return 0;
}

/* demodulation.py:25-38 -- res[j] = angle(d[2j]*conj(d[2j+1])), then resample(res[:m/2], m). */
int orc_fm(const cplx *y, size_t m, double *out) {
    size_t h = m >> 1;
    double *a = (double *) malloc((h ? h : 1) * sizeof(double));
    if (!a) return -2;
    for (size_t j = 0; j < h; ++j) {
        cplx p = y[2 * j] * conj(y[2 * j + 1]);
        a[j] = atan2(cimag(p), creal(p));
    }
    int rc = orc_resample(a, h, m, out);
    free(a);
    return rc;
}

/* demodulation.py:41-48 -- abs(square(d)). */
void orc_am(const cplx *y, size_t m, double *out) {
    for (size_t i = 0; i < m; ++i) out[i] = cabs(y[i] * y[i]);
}

/*
 * dsp_processor.py:140-149 for one chunk and all rows:
 *   x[k,:] = z * shift[k,:]   (shift == NULL -> no shift, single row)
 *   y = decimate(x, q);  demod;  sosfilt(out_sos) with zero state.
 * out: rows*m doubles (row-major).  ydec (optional): rows*m complex.
 */
/* scratch for one chunk: x[n], full[n], ext[n + 2*edge], y[m+1] */
static size_t chunk_ws_elems(size_t n, int q, int dec_nsec) {
    return 3 * n + 6 * (2 * (size_t) dec_nsec + 1) + n / (size_t) q + 2;
}

static int process_chunk_ws(const cplx *z, size_t n, int q, int mode, const cplx *shift, int rows,
                            const double *dec_sos, int dec_nsec, const double *out_sos, int out_nsec,
                            double *out, cplx *ydec, cplx *ws) {
    size_t m = n / (size_t) q;
    cplx *x = ws, *full = ws + n, *y = full + n, *ext = y + m + 2;
    int rc = 0;
    for (int r = 0; r < rows && rc == 0; ++r) {
        if (shift) {
            /* plain complex product (no C99 Annex G inf/nan recovery: inputs are finite) */
            const cplx *sh = shift + (size_t) r * n;
            for (size_t i = 0; i < n; ++i) {
                double a = creal(z[i]), b = cimag(z[i]), c = creal(sh[i]), d = cimag(sh[i]);
                x[i] = (a * c - b * d) + (a * d + b * c) * I;
            }
        } else {
            memcpy(x, z, n * sizeof(cplx));
        }
        rc = sosfiltfilt_ws(dec_sos, dec_nsec, x, n, full, ext);
        if (rc) break;
        { size_t k = 0; for (size_t i = 0; i < n; i += (size_t) q) y[k++] = full[i]; }
        if (ydec) memcpy(ydec + (size_t) r * m, y, m * sizeof(cplx));
        double *o = out + (size_t) r * m;
        if (mode == ORC_FM) rc = orc_fm(y, m, o); else orc_am(y, m, o);
        double st[32] = {0};
        orc_sosfilt_r(out_sos, out_nsec, o, m, st);
    }
    return rc;
}

int orc_process_chunk(const cplx *z, size_t n, int q, int mode, const cplx *shift, int rows,
                      const double *dec_sos, int dec_nsec, const double *out_sos, int out_nsec,
                      double *out, cplx *ydec) {
    cplx *ws = (cplx *) malloc(chunk_ws_elems(n, q, dec_nsec) * sizeof(cplx));
    if (!ws) return -2;
    int rc = process_chunk_ws(z, n, q, mode, shift, rows, dec_sos, dec_nsec, out_sos, out_nsec, out, ydec, ws);
    free(ws);
    return rc;
}

/*
 * Whole stream (reader steps + processor steps), chunks in parallel with OpenMP when available.
 * raw: n_chunks * n * 2 items of `code`.  out layout: [chunk][row][m].  off: carried IQ offset (in/out).
 * Returns 0 on success.  Used by tests for multi-chunk parity and by bench.py as the CPU baseline.
 */
int orc_process_stream(const void *raw, int code, int swap, size_t n, size_t n_chunks, int q, int mode,
                       const cplx *shift, int rows, const double *dec_sos, int dec_nsec,
                       const double *out_sos, int out_nsec, int do_norm, double xmin, double inv_range,
                       int do_iq, double inductance, cplx *off, double *out, int nthreads) {
    size_t m = n / (size_t) q;
    int sz = k_item_size[code];
    int rc_all = 0;
    cplx *zall = NULL;
    if (do_iq) {
        /* the recurrence is sequential over the whole stream: do the reader steps first */
        zall = (cplx *) malloc(n * n_chunks * sizeof(cplx));
        if (!zall) return -2;
        for (size_t c = 0; c < n_chunks; ++c) {
            cplx *z = zall + c * n;
            orc_unpack((const unsigned char *) raw + c * n * 2 * (size_t) sz, code, swap, n, z);
            if (do_norm) orc_normalize(z, n, xmin, inv_range);
            orc_correct_iq(z, n, off, inductance);
        }
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        /* one scratch block per thread, allocated once (per-chunk malloc of MiB-sized blocks serialises on mmap) */
        cplx *ws = (cplx *) malloc((chunk_ws_elems(n, q, dec_nsec) + n) * sizeof(cplx));
        cplx *zbuf = ws ? ws + chunk_ws_elems(n, q, dec_nsec) : NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long c = 0; c < (long) n_chunks; ++c) {
            int rc = -2;
            if (ws) {
                const cplx *z;
                if (zall) {
                    z = zall + (size_t) c * n;
                } else {
                    orc_unpack((const unsigned char *) raw + (size_t) c * n * 2 * (size_t) sz, code, swap, n, zbuf);
                    if (do_norm) orc_normalize(zbuf, n, xmin, inv_range);
                    z = zbuf;
                }
                rc = process_chunk_ws(z, n, q, mode, shift, rows, dec_sos, dec_nsec, out_sos, out_nsec,
                                      out + (size_t) c * (size_t) rows * m, NULL, ws);
            }
            if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
                rc_all = rc;
            }
        }
        free(ws);
    }
    free(zall);
    return rc_all;
}

/*
 * ---- float32 records as the reference computes them: complex64 through the decimator ---------------------------------------
 * read_file.py:100-101 makes `re + 1j*im` complex64 for float32 records; dsp_processor.py:171-177 gives x and y that dtype;
 * shiftFreq's complex128 product is stored into x as complex64 (demodulation.py:71-79); scipy.signal.decimate keeps the dtype:
 * sos = np.asarray(sos, dtype=complex64), sosfiltfilt = odd extension by 27, zi = sosfilt_zi(sos) * x[0], _sosfilt forwards,
 * zi * y[-1], _sosfilt over the reversed sequence, [27:-27][::q] (scipy/signal/_signaltools.py, _sosfilt.pyx: _sosfilt_float
 * with DTYPE = float complex).  Every operation below is the IEEE single-precision operation scipy's compiled loop performs,
 * in its order; the coefficients' imaginary parts are zero, so a complex product is the two real products.  Bit-identical to
 * scipy 1.15.3 on x86-64 (tests/test_oracle_golden.py pins it to the reference's complex64 `y0` vectors with array_equal).
 * Everything after the decimator is float64 in the reference too (the numba kernels take complex128).
 */
typedef struct { float re, im; } c32;

/* complex64(complex128(z) * shift): numpy's complex multiply loop on AVX2 / AVX-512 hosts is re = fma(ar, br, -(ai bi)),
 * im = fma(ar, bi, ai br) (numpy/_core/src/umath/loops_arithm_fp.dispatch.c.src); the store into x rounds to single */
static c32 shift_c64(float re, float im, const cplx *sh) {
    c32 v;
    if (sh) {
        const double ar = re, ai = im, br = creal(*sh), bi = cimag(*sh);
        const double pi = ai * bi, pr = ai * br;
        v.re = (float) fma(ar, br, -pi);
        v.im = (float) fma(ar, bi, pr);
    } else {
        v.re = re;
        v.im = im;
    }
    return v;
}

static float load_f32(const unsigned char *p, int swap) {
    uint32_t u;
    memcpy(&u, p, 4);
    if (swap) u = __builtin_bswap32(u);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* work: n + 54 elements.  y: n / q elements. */
int orc_decimate_c64(const float *sos /* [nsec][6] */, const float *zi /* [nsec][2] */, int nsec, const c32 *x, size_t n, int q,
                     c32 *y, c32 *work) {
    enum { E = 27 };
    if (nsec != 4 || n <= E) return -1;
    const size_t N = n + 2 * E;
    for (size_t e = 0; e < E; ++e) {
        work[e].re = 2.0f * x[0].re - x[E - e].re;
        work[e].im = 2.0f * x[0].im - x[E - e].im;
        work[E + n + e].re = 2.0f * x[n - 1].re - x[n - 2 - e].re;
        work[E + n + e].im = 2.0f * x[n - 1].im - x[n - 2 - e].im;
    }
    memcpy(work + E, x, n * sizeof(c32));
    float z[4][2][2];
    for (int pass = 0; pass < 2; ++pass) {
        const c32 x0 = pass == 0 ? work[0] : work[N - 1];
        for (int s = 0; s < nsec; ++s)
            for (int k = 0; k < 2; ++k) { z[s][k][0] = zi[2 * s + k] * x0.re; z[s][k][1] = zi[2 * s + k] * x0.im; }
        for (size_t t = 0; t < N; ++t) {
            const size_t idx = pass == 0 ? t : N - 1 - t;
            float xr = work[idx].re, xi = work[idx].im;
            for (int s = 0; s < nsec; ++s) {
                const float *c = sos + 6 * s;
                const float nr = c[0] * xr + z[s][0][0], ni = c[0] * xi + z[s][0][1];
                z[s][0][0] = (c[1] * xr - c[4] * nr) + z[s][1][0];
                z[s][0][1] = (c[1] * xi - c[4] * ni) + z[s][1][1];
                z[s][1][0] = c[2] * xr - c[5] * nr;
                z[s][1][1] = c[2] * xi - c[5] * ni;
                xr = nr;
                xi = ni;
            }
            work[idx].re = xr;
            work[idx].im = xi;
        }
    }
    for (size_t k = 0; k < n / (size_t) q; ++k) y[k] = work[E + k * (size_t) q];
    return 0;
}

/* float32 records, the whole chain per chunk and row as the reference runs it.  out: [chunk][row][m]; ydec (optional):
 * [chunk][row][m] complex64, the value of `y` after dsp_processor.py:147. */
int orc_process_stream_f32c(const void *raw, int swap, size_t n, size_t n_chunks, int q, int mode, const cplx *shift, int rows,
                            const float *sos32, const float *zi32, int dec_nsec, const double *out_sos, int out_nsec,
                            double *out, c32 *ydec, int nthreads) {
    const size_t m = n / (size_t) q;
    int rc_all = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        c32 *x = (c32 *) malloc((2 * n + 54 + m) * sizeof(c32));
        cplx *yc = (cplx *) malloc((m + 2) * sizeof(cplx));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long c = 0; c < (long) n_chunks; ++c) {
            int rc = (x && yc) ? 0 : -2;
            const unsigned char *cr = (const unsigned char *) raw + (size_t) c * n * 8;
            for (int r = 0; r < rows && rc == 0; ++r) {
                c32 *work = x + n, *y = work + n + 54;
                for (size_t i = 0; i < n; ++i)
                    x[i] = shift_c64(load_f32(cr + 8 * i, swap), load_f32(cr + 8 * i + 4, swap), shift ? shift + (size_t) r * n + i : NULL);
                rc = orc_decimate_c64(sos32, zi32, dec_nsec, x, n, q, y, work);
                if (rc) break;
                if (ydec) memcpy(ydec + ((size_t) c * rows + r) * m, y, m * sizeof(c32));
                for (size_t k = 0; k < m; ++k) yc[k] = (double) y[k].re + (double) y[k].im * I;
                double *o = out + ((size_t) c * rows + r) * m;
                if (mode == ORC_FM) rc = orc_fm(yc, m, o); else orc_am(yc, m, o);
                double st[32] = {0};
                orc_sosfilt_r(out_sos, out_nsec, o, m, st);
            }
            if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
                rc_all = rc;
            }
        }
        free(x);
        free(yc);
    }
    return rc_all;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- --smooth-output: scipy.signal.savgol_filter(z, window, polyorder), mode='interp' (dsp_processor.py:159-160) ------------
 * Restated from scipy's published algorithm (scipy/signal/_savitzky_golay.py): interior = correlation with the least-squares
 * smoothing coefficients (minimum-norm solution of A c = e0, A[k][j] = x_j^k, x = pos .. pos-window+1, pos = window/2 or
 * window/2 - 1/2 for even windows); the first / last window/2 outputs = a degree-`polyorder` polynomial fitted to the first /
 * last `window` samples, evaluated at their positions.  Small dense systems, solved in long double. */
static int orc_solve(int k, long double *M /* k x k */, long double *rhs) {
    for (int c = 0; c < k; ++c) {
        int piv = c;
        for (int r = c + 1; r < k; ++r) if (fabsl(M[r * k + c]) > fabsl(M[piv * k + c])) piv = r;
        if (M[piv * k + c] == 0.0L) return -1;
        if (piv != c) {
            for (int j = 0; j < k; ++j) { long double t = M[c * k + j]; M[c * k + j] = M[piv * k + j]; M[piv * k + j] = t; }
            long double t = rhs[c]; rhs[c] = rhs[piv]; rhs[piv] = t;
        }
        for (int r = c + 1; r < k; ++r) {
            const long double f = M[r * k + c] / M[c * k + c];
            for (int j = c; j < k; ++j) M[r * k + j] -= f * M[c * k + j];
            rhs[r] -= f * rhs[c];
        }
    }
    for (int c = k - 1; c >= 0; --c) {
        long double s = rhs[c];
        for (int j = c + 1; j < k; ++j) s -= M[c * k + j] * rhs[j];
        rhs[c] = s / M[c * k + c];
    }
    return 0;
}

int orc_savgol(double *z, size_t m, int window, int polyorder) {
    const int k = polyorder + 1, half = window / 2;
    if (window < 1 || polyorder < 0 || polyorder >= window || (size_t) window > m || k > 8) return -1;
    double *out = (double *) malloc(m * sizeof(double));
    long double *coef = (long double *) malloc((size_t) window * sizeof(long double));
    if (!out || !coef) { free(out); free(coef); return -2; }
    /* interior coefficients: c = A^T (A A^T)^-1 e0 with x_j = pos - j (the 'conv' ordering of savgol_coeffs) */
    const long double pos = (window % 2) ? (long double) half : (long double) half - 0.5L;
    long double G[64], lam[8];
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) {
            long double s = 0;
            for (int j = 0; j < window; ++j) s += powl(pos - j, a) * powl(pos - j, b);
            G[a * k + b] = s;
        }
    for (int a = 0; a < k; ++a) lam[a] = a == 0 ? 1.0L : 0.0L;
    if (orc_solve(k, G, lam)) { free(out); free(coef); return -3; }
    for (int j = 0; j < window; ++j) {
        long double s = 0;
        for (int a = 0; a < k; ++a) s += lam[a] * powl(pos - j, a);
        coef[j] = s;
    }
    /* scipy.ndimage.convolve1d(z, coef, mode='constant') with the (symmetric) coefficients: samples i - first .. i - first +
     * window - 1, first = window/2 for odd windows and window/2 - 1 for even ones (convolve1d's origin rule); zeros outside */
    const int first = (window % 2) ? half : half - 1;
    for (size_t i = 0; i < m; ++i) {
        long double s = 0;
        for (int j = 0; j < window; ++j) {
            const long long t = (long long) i + j - first;
            if (t >= 0 && t < (long long) m) s += coef[j] * z[t];
        }
        out[i] = (double) s;
    }
    /* edges: least-squares polynomial over the first / last `window` samples (normal equations on x = 0 .. window-1) */
    for (int side = 0; side < 2; ++side) {
        const size_t w0 = side ? m - (size_t) window : 0;
        long double N[64], rhs[8];
        for (int a = 0; a < k; ++a) {
            for (int b = 0; b < k; ++b) {
                long double s = 0;
                for (int j = 0; j < window; ++j) s += powl(j, a + b);
                N[a * k + b] = s;
            }
            long double s = 0;
            for (int j = 0; j < window; ++j) s += powl(j, a) * z[w0 + j];
            rhs[a] = s;
        }
        if (orc_solve(k, N, rhs)) { free(out); free(coef); return -3; }
        for (int i = 0; i < half; ++i) {
            const int x = side ? window - half + i : i;
            long double s = 0;
            for (int a = k - 1; a >= 0; --a) s = s * x + rhs[a];
            out[w0 + x] = (double) s;
        }
    }
    memcpy(z, out, m * sizeof(double));
    free(out);
    free(coef);
    return 0;
}
