/*
 * sdrterm_hip.h -- C ABI of the MI355X-native sdrterm DSP hot path (libsdrterm_hip.so).
 *
 * This is the drop-in boundary for peads/sdrterm's src/dsp streaming chain.  Every entry point uses
 * plain C types and caller-owned buffers; nothing throws across it.  Each function cites the
 * reference interface it replaces (paths relative to the reference repository root).
 *
 * One context == one reference processor instance:
 *     DspProcessor(fs, center, omegaOut, dec, ...)            src/dsp/dsp_processor.py:48-77
 *     VfoProcessor(fs, vfoHost, vfos, ...)                    src/dsp/vfo_processor.py:36-57
 * plus the reader's per-sample steps that precede it in the stream
 *     unpack / _normalize / _correctIq                        src/misc/read_file.py:55-103
 *
 * A "chunk" is what the reference's reader enqueues per read: readSize = 131072 bytes of interleaved
 * (re, im) records (src/misc/read_file.py:38,51,119-124).  Every chunk is filtered independently, as
 * the reference does (dsp_processor.py:140-149); the only cross-chunk state is the IQ-correction
 * offset (read_file.py:53,103), which the context carries.
 */
#ifndef SDRTERM_HIP_H
#define SDRTERM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRT_ABI_VERSION 7

/* Input encodings: src/misc/file_util.py:44-56 (DataType b,B,h,H,i,I,f,d). */
enum sdrt_dtype {
    SDRT_S8 = 0, SDRT_U8 = 1, SDRT_S16 = 2, SDRT_U16 = 3, SDRT_S32 = 4, SDRT_U32 = 5, SDRT_F32 = 6, SDRT_F64 = 7
};

/* Demodulation: DspProcessor.selectOutputFm / selectOutputAm (dsp_processor.py:116-128) with the output lowpass, and
 * selectOutputReal / selectOutputImag (dsp_processor.py:130-138): the real / imaginary part of the decimated samples
 * (demodulation.py:51-68) with no output filter (out_sos is ignored and may be NULL). */
enum sdrt_mode { SDRT_FM = 0, SDRT_AM = 1, SDRT_RE = 2, SDRT_IM = 3 };

enum sdrt_status {
    SDRT_OK = 0,
    SDRT_EINVAL = 1,      /* bad argument / unsupported configuration (message in sdrt_last_error) */
    SDRT_EHIP = 2,        /* a HIP runtime call failed */
    SDRT_ENOMEM = 3,
    SDRT_ENODEVICE = 4    /* no usable GPU: the library never falls back to the CPU */
};

/*
 * Which arithmetic the decimator's block products ("stage A") run in -- the reference has one arithmetic,
 * float64 (scipy.signal.decimate on complex128, dsp_processor.py:147); this build's default for 8/16-bit integer and float32
 * records is exact integer slicing on the int8 matrix cores against a 40-bit image of the operator, which meets 1e-6 while the
 * in-band signal is above about -100 dB of the chunk's peak ("Conditioning").
 *   SDRT_STAGE_AUTO  the int8 forms where they apply, and every chunk whose decimated samples come out below 2^-33 of its
 *                    input peak (where the 40-bit image could show) is done again in float64 before the demodulation stage
 *   SDRT_STAGE_I8    the int8 forms where they apply, no second look
 *   SDRT_STAGE_F64   float64 throughout (the fp64 matrix cores), whatever the encoding
 * (Environment SDRTERM_STAGE_A=auto|i8|f64|i8t overrides the field: diagnostics.)
 */
enum sdrt_stage_a { SDRT_STAGE_AUTO = 0, SDRT_STAGE_I8 = 1, SDRT_STAGE_F64 = 2 };

typedef struct sdrt_ctx sdrt_ctx;

typedef struct sdrt_config {
    uint32_t struct_size;     /* sizeof(sdrt_config), for ABI evolution */
    int32_t device;           /* HIP device ordinal */
    double fs;                /* input sample rate, Hz (DspProcessor.fs) */
    int32_t decimation;       /* DspProcessor.decimation, >= 2 (dsp_processor.py:92-97) */
    int32_t mode;             /* enum sdrt_mode */
    int32_t dtype;            /* enum sdrt_dtype of each of re and im */
    int32_t swap_endianness;  /* records are opposite to host byte order (read_file.py:48-49) */
    int32_t n_samples;        /* complex samples per chunk (131072 / record size in the reference) */
    int32_t n_rows;           /* 1 for DspProcessor; len(vfos)+1 for VfoProcessor (vfo_processor.py:45-47) */
    const double *row_freqs;  /* Hz per row, as VfoProcessor builds them (csv+center ..., center LAST);
                                 for DspProcessor: {center}.  May be NULL when apply_shift == 0 */
    int32_t apply_shift;      /* 0 == the reference's _shift is None (center == 0, no --simo) */
    int32_t normalize;        /* --normalize-input (read_file.py:82-96) */
    double norm_xmin;         /* generateDomain(): xmin      (read_file.py:177-196) */
    double norm_inv_range;    /* generateDomain(): 1/(xmax - xmin) */
    int32_t correct_iq;       /* --correct-iq (read_file.py:57-77) */
    int32_t reserved0;
    double iq_inductance;     /* impedance / fs (read_file.py:65) */
    const double *dec_sos;    /* decimator SOS [dec_sections][6]: scipy cheby1(8, 0.05, 0.8/q, 'sos'),
                                 what scipy.signal.decimate builds at dsp_processor.py:147 */
    int32_t dec_sections;
    int32_t out_sections;
    const double *out_sos;    /* output lowpass SOS [out_sections][6]: generateEllipFilter, dsp_processor.py:39-45 */
    const int32_t *row_modes; /* NULL: every row demodulates as `mode`, as in the reference, whose mode is processor-wide
                                 (io_args.py:118).  Else [n_rows] of SDRT_FM / SDRT_AM: a build extension for --simo with FM
                                 and AM channels in one pass over the input (BASELINE configs[2]); row r then equals row r of a
                                 context created with mode = row_modes[r] and otherwise the same settings. */
    /* ---- ABI 7 ---- */
    int32_t stage_a;          /* enum sdrt_stage_a */
    int32_t faithful_f32;     /* float32 records only (ignored for the other encodings, which the reference computes in float64
                                 anyway): reproduce the reference's single-precision chain bit for bit.  `re + 1j*im` is complex64
                                 there (read_file.py:100-101), the x / y buffers take that dtype (dsp_processor.py:171-177) and
                                 scipy.signal.decimate runs its zero-phase IIR in complex64 (dsp_processor.py:147): the outputs carry
                                 ~3e-4 of single-precision noise that only the same operations in the same order reproduce.
                                 0: float32 records are filtered in float64 / exactly (more accurate than the reference, and several
                                 times faster); 1: one (chunk, row) sequence per GPU lane runs scipy's recurrence as scipy runs it.
                                 Needs f32_sos / f32_zi (and shift_table when apply_shift); not with correct_iq (the reference's
                                 gufunc takes complex128 and leaves a complex64 buffer uncorrected, read_file.py:66-77). */
    const float *f32_sos;     /* [dec_sections][6]: real parts of np.asarray(cheby1(8, 0.05, 0.8/q, output='sos'), complex64), the
                                 cast scipy.signal.decimate makes (sos = np.asarray(sos, dtype=result_type)) */
    const float *f32_zi;      /* [dec_sections][2]: real parts of scipy.signal.sosfilt_zi of that table (sosfiltfilt) */
    const double *shift_table;/* [n_rows][n_samples][2]: the reference's shift table itself, exp(-2j pi (f/fs) arange(n)) as numpy
                                 evaluates it (dsp_processor.py:185-187, vfo_processor.py:71-74); NULL when apply_shift == 0 */
} sdrt_config;

/* Library / device discovery. */
int sdrt_abi_version(void);
size_t sdrt_config_size(void);       /* sizeof(sdrt_config) the library was built with (binding self-check) */
int sdrt_device_count(int *count);
const char *sdrt_last_error(void);   /* thread-local, valid until the next call on the same thread */

/*
 * Construct / destroy: replaces DspProcessor.__init__ + selectOutput*() (dsp_processor.py:51-128) or
 * VfoProcessor.__init__ (vfo_processor.py:38-57) together with the lazy buffer/shift allocation of
 * _processData/_generateShift (dsp_processor.py:164-187, vfo_processor.py:71-78).
 */
int sdrt_create(const sdrt_config *cfg, sdrt_ctx **out);
void sdrt_destroy(sdrt_ctx *ctx);

/* Output samples per chunk per row: n_samples / decimation (dsp_processor.py:173). */
int sdrt_out_per_chunk(const sdrt_ctx *ctx);
int sdrt_rows(const sdrt_ctx *ctx);

/*
 * Process n_chunks consecutive chunks: the body of DspProcessor._processData/_processChunk
 * (dsp_processor.py:140-149,164-183) preceded by feedBuffers' per-sample steps (read_file.py:100-103).
 *   raw : n_chunks * n_samples interleaved (re, im) records of cfg.dtype
 *   out : rows * n_chunks * m float64, row-major [row][chunk][m] -- row r is exactly the byte stream the
 *         reference writes for that row (file for DspProcessor, client socket r for VfoProcessor).
 * The *_device form is asynchronous on `stream` (a hipStream_t, NULL = default stream) and takes
 * device pointers; the *_host form copies in, runs, copies out and synchronises.
 */
int sdrt_process_device(sdrt_ctx *ctx, const void *raw_dev, size_t n_chunks, double *out_dev, void *stream);
int sdrt_process_host(sdrt_ctx *ctx, const void *raw_host, size_t n_chunks, double *out_host);

/*
 * Everything a call with up to max_chunks chunks needs in device scratch (the decimated samples between the two stages, the
 * IQ prefix tables, the scratch of sdrt_iq_range_device, the chunk pair of half-size chunks): allocated here, so that no
 * sdrt_process_* / sdrt_iq_range_device call up to that size allocates.  Without it the scratch grows on the first larger batch
 * (the reference allocates its x / y / z buffers on the first chunk as well, dsp_processor.py:171-178) and never shrinks.
 */
int sdrt_reserve(sdrt_ctx *ctx, size_t max_chunks);

/*
 * Host buffers, batches in flight: sdrt_submit_host starts copy-in, kernels and copy-out of one batch and returns a ticket;
 * sdrt_wait(ticket) returns when that batch's `out_host` is filled.  Up to three batches are in flight (the copy-in of
 * batch k+1 beside the kernels of batch k and the copy-out of batch k-1); a fourth submit first completes the oldest.
 * `raw_host` may be reused once the call has returned if it is pageable, after sdrt_wait if it came from sdrt_host_alloc;
 * `out_host` must stay valid until sdrt_wait.  Batches are processed in submission order (the IQ offset carries over).
 * sdrt_process_host is submit + wait.
 */
int sdrt_submit_host(sdrt_ctx *ctx, const void *raw_host, size_t n_chunks, double *out_host, int64_t *ticket);
int sdrt_wait(sdrt_ctx *ctx, int64_t ticket);

/* Same as sdrt_process_device, additionally storing the decimated complex samples (the value of `y` after
 * dsp_processor.py:147) as interleaved float64 (re, im), layout [row][chunk][m][2].  For parity tests. */
int sdrt_process_device_debug(sdrt_ctx *ctx, const void *raw_dev, size_t n_chunks, double *out_dev,
                              double *ydec_dev, void *stream);

/*
 * --smooth-output: scipy.signal.savgol_filter(z, window, 3) on every chunk's output samples before they are written
 * (DspProcessor._transformData, dsp_processor.py:159-160; single-VFO processors only, VfoProcessor never smooths).
 * The filter is linear; the caller designs it with scipy where the reference does and passes the operator:
 *   interior outputs  i in [n_edge, m - n_edge):  out[i] = sum_k taps[k] * z[i + tap_first + k]   (samples outside the chunk count 0)
 *   first n_edge outputs:  out[i] = sum_k left[i][k] * z[k],  last n_edge:  out[m-n_edge+i] = sum_k right[i][k] * z[m-window+k]
 * (mode='interp': polynomial fits over the first / last `window` samples).  window == 0 switches smoothing off.
 */
int sdrt_set_smoothing(sdrt_ctx *ctx, int window, int ntaps, int tap_first, const double *taps, int n_edge,
                       const double *left /*[n_edge][window]*/, const double *right /*[n_edge][window]*/);

/* The carried IQ-correction offset (read_file.py:53): get/set as (re, im). */
int sdrt_get_iq_offset(sdrt_ctx *ctx, double *re, double *im);
int sdrt_set_iq_offset(sdrt_ctx *ctx, double re, double im);
/*
 * What a range of chunks does to the carried offset, without processing it: off_end = a * off_start + (b_re + i b_im)
 * (the recursion of iq_correction.py:19-26 is affine in its state).  One pass over the input, the context's own state is
 * left alone.  For a stream whose chunks are split over several GPUs ("by chunk"): every rank calls this on
 * its range, the (a, b) pairs are exchanged (48 bytes per rank), each rank folds the pairs of the ranks before it into
 * its starting offset (sdrt_set_iq_offset) and then processes its range.
 */
int sdrt_iq_range_device(sdrt_ctx *ctx, const void *raw_dev, size_t n_chunks, double *a, double *b_re, double *b_im, void *stream);

/*
 * Per-kernel device timing with HIP events recorded on the launch stream (for bench.py's roofline line).
 * enable != 0 makes every sdrt_process_* call bracket each kernel with events; sdrt_kernel_time returns the
 * accumulated milliseconds and launch count for kernel `which` (0 = chain kernel: unpack..decimate, 1 = post kernel: demodulate/resample/output lowpass,
 * 2 = IQ-correction prefix kernels) and resets nothing; sdrt_profile_reset clears the accumulators.
 */
int sdrt_profile_enable(sdrt_ctx *ctx, int enable);
int sdrt_kernel_time(sdrt_ctx *ctx, int which, double *total_ms, int64_t *launches);
int sdrt_profile_reset(sdrt_ctx *ctx);
const char *sdrt_kernel_name(int which);

/* Thin device-memory helpers so a ctypes host needs no other GPU runtime binding. */
int sdrt_malloc(void **dev_ptr, size_t bytes);
int sdrt_free(void *dev_ptr);
int sdrt_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int sdrt_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int sdrt_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes);
int sdrt_device_synchronize(void);
int sdrt_set_device(int device);
int sdrt_mem_info(size_t *free_bytes, size_t *total_bytes);     /* of the current device (sizing batches) */
unsigned long long sdrt_alloc_count(void);                      /* device allocations the library has made so far: constant across
                                                                   the processing calls of a reserved context (sdrt_reserve) */

/*
 * Streams and pinned host memory, for hosts that overlap transfers with computation themselves (the multi-GPU --simo
 * driver, sdrterm_amd/parallel.py: broadcast of block k+1 beside the kernels of block k).
 */
int sdrt_stream_create(void **stream);
int sdrt_stream_destroy(void *stream);
int sdrt_stream_synchronize(void *stream);
int sdrt_host_alloc(void **host_ptr, size_t bytes);      /* page-locked: async copies, full PCIe rate */
int sdrt_host_free(void *host_ptr);
int sdrt_memcpy_h2d_async(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int sdrt_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, void *stream);
/* `waiter` waits (on the device) for everything submitted to `signaller` so far. */
int sdrt_stream_wait_stream(void *waiter, void *signaller);
/* Events: a point in a stream's work that other streams can wait for on the device (sdrt_stream_wait_event) and the host can
 * poll without blocking (sdrt_event_query: *done = 1 once everything before the record has completed).  The multi-GPU driver
 * uses them for its two block buffers -- the broadcast of block k+1 waits for the kernels of block k-1 only, not for those of
 * block k -- and to bound its waits: a rank polls with a deadline and watches the halt flag instead of sitting in
 * sdrt_device_synchronize behind a collective whose peer has died (vfo_processor.py:113-126 is the reference's teardown). */
int sdrt_event_create(void **event);
int sdrt_event_destroy(void *event);
int sdrt_event_record(void *event, void *stream);
int sdrt_stream_wait_event(void *stream, void *event);
int sdrt_event_query(void *event, int *done);
int sdrt_event_synchronize(void *event);
/* ABI 6: events that carry a time stamp (sdrt_event_create's do not: cheaper to record), the milliseconds between two of them once
 * both have completed, and a device-to-device copy on a stream -- what a host needs to time steps and to stage blocks without any
 * other GPU runtime binding (bench.py times its steps with these; the reference has no counterpart, it times with `time`,
 * example.sh:43,52). */
int sdrt_event_create_timed(void **event);
int sdrt_event_elapsed_ms(void *start, void *stop, double *ms);
int sdrt_memcpy_d2d_async(void *dst_dev, const void *src_dev, size_t bytes, void *stream);

/*
 * Multi-GPU --simo: one process per GPU runs VfoProcessor (vfo_processor.py:38-57,80-126) on its
 * contiguous share of the rows; the process that holds the reader's Queue hands every raw block to the others.  That is
 * the only exchange on the path, and it runs over RCCL (librccl, loaded on first use) between device buffers:
 *   sdrt_comm_unique_id   one rank creates the 128-byte id, the host side passes it to the others (pipe, socket, file)
 *   sdrt_comm_init        every rank, collectively; binds the calling process to `device`
 *   sdrt_comm_bcast       the raw block from `root` to every rank, in place, asynchronous on `stream`
 *   sdrt_comm_allgather   alternative when every rank already holds 1/world of the block (each host link fed one shard):
 *                         all seven xGMI links of every GPU carry one shard each way instead of the root's links carrying all
 */
typedef struct sdrt_comm sdrt_comm;
int sdrt_comm_unique_id(void *id128);
int sdrt_comm_init(const void *id128, int world, int rank, int device, sdrt_comm **out);
int sdrt_comm_bcast(sdrt_comm *comm, void *dev_buf, size_t bytes, int root, void *stream);
int sdrt_comm_allgather(sdrt_comm *comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *stream);
/* sdrt_comm_bcast's result built for point-to-point links: the root sends every rank its 1/world of the block, the ranks then
 * all-gather the shards in place -- every link carries one shard instead of every hop the whole block.  bytes % world != 0, a
 * single rank, less than 1 MiB, or a librccl without ncclSend / ncclRecv / ncclGroup*: plain sdrt_comm_bcast. */
int sdrt_comm_bcast_scatter(sdrt_comm *comm, void *dev_buf, size_t bytes, int root, void *stream);
int sdrt_comm_world(const sdrt_comm *comm);
int sdrt_comm_rank(const sdrt_comm *comm);
void sdrt_comm_destroy(sdrt_comm *comm);

/*
 * Stand-alone discriminators on already-decimated samples: fmDemod / amDemod / realOutput / imagOutput of
 * src/dsp/demodulation.py:25-68 (mode 0 FM: pair angle + FFT-equivalent resample m/2 -> m; 1 AM: re^2+im^2; 2 re; 3 im).
 *   y_host : rows * m interleaved (re, im) float64;  out_host : rows * m float64.  Host buffers, synchronous.
 */
int sdrt_demodulate_host(int mode, const double *y_host, int rows, int m, double *out_host);

/*
 * FFT taps of the plots, on raw chunks as the reader hands them on (host buffers, synchronous).  Both decode the records (and
 * normalise them when asked, read_file.py:82-96), shift by shift_hz (AbstractPlot._omega = -2j pi offset/fs applied with
 * shiftFreq, plots/abstract_plot.py:76, spectrum_analyzer_plot.py:77, waterfall_plot.py:97) and transform:
 *   sdrt_spectrum_host  SpectrumAnalyzerPlot.update (plots/spectrum_analyzer_plot.py:74-88): log10(|fftshift(fft(y) / n)|^2) of each
 *                       whole chunk -> out_host [n_chunks][n_samples]
 *   sdrt_stft_host      WaterfallPlot.update (plots/waterfall_plot.py:47-58,90-100): 10 log10 |ShortTimeFFT.stft(y)| with the
 *                       window the caller passes already scaled (scipy scale_to='magnitude'), fft_mode 'centered', frames centred
 *                       on multiples of hop, zeros beyond the chunk -> out_host [n_chunks][mfft][n_samples / hop + 1]
 */
int sdrt_spectrum_host(int dtype, int swap_endianness, const void *raw_host, int n_samples, int n_chunks, double shift_hz, double fs,
                       int normalize, double norm_xmin, double norm_inv_range, double *out_host);
int sdrt_stft_host(int dtype, int swap_endianness, const void *raw_host, int n_samples, int n_chunks, double shift_hz, double fs,
                   int normalize, double norm_xmin, double norm_inv_range, int nperseg, int hop, int mfft, const double *window,
                   double *out_host);

/* The tap entry points keep their device buffers from call to call (per calling thread; the plots transform a chunk per frame):
 * this frees them. */
int sdrt_taps_release(void);

/* Diagnostic: one sdrt_process_device call in which the first n_wgs workgroups record s_memtime at their phase
 * boundaries; stamps_host receives n_wgs*4*16 uint64 ([workgroup][wave][16]).  Not for production use. */
int sdrt_debug_stamps(sdrt_ctx *ctx, const void *raw_dev, size_t n_chunks, double *out_dev,
                      unsigned long long *stamps_host, int n_wgs);

/* Self-tests of the matrix-core operand layouts the decimator relies on (one 16x16x4 f64 / 16x16x64 int8 product, the lane swaps). */
int sdrt_selftest_mfma(double *max_abs_err);
int sdrt_selftest_mfma_i8(int *mismatches);

/* Which stage-A kernel the context uses: 0 = fp64 matrix cores; 2 = int8 matrix cores with exact integer slicing, one
 * block per column (16x16x64); 3 = the same arithmetic in the wide-column persistent kernel (default for 8/16-bit integer
 * encodings at decimation 32 / 64); 4 = the per-sample modal form in fp64 (decimation 2 and 4); 5 = the reference's complex64
 * recurrence for float32 records (sdrt_config.faithful_f32); 6 = decimation 2 .. 16 of 16-bit records in one pass on the int8 matrix
 * cores.  0 - 4 and 6 meet the same tolerance; sdrt_config.stage_a chooses among them. */
int sdrt_stage_a_kind(const sdrt_ctx *ctx);
/* stage_a = SDRT_STAGE_AUTO: how many (chunk, row) items this context has done a second time in float64 so far (their in-band
 * content lay near the int8 forms' operator image; DESIGN.md section 1, "Conditioning").  Synchronises the device. */
int sdrt_refined_items(sdrt_ctx *ctx, unsigned long long *total);

#ifdef __cplusplus
}
#endif
#endif /* SDRTERM_HIP_H */
