"""Host-side wrapper around one ``sdrt_ctx`` (include/sdrterm_hip.h): coefficient design + configuration.

Filter coefficients are designed on the host exactly where the reference designs them, with the same
scipy calls, and handed to the library as plain SOS tables:
  * decimator  -- scipy.signal.decimate's default IIR, ``cheby1(8, 0.05, 0.8/q, output='sos')``
                  (what the reference invokes at src/dsp/dsp_processor.py:147)
  * output     -- ``ellip(3, 1, 30, omegaOut, 'lowpass', output='sos', fs=decimatedFs)``
                  (generateEllipFilter, src/dsp/dsp_processor.py:39-45)
All per-sample work runs on the GPU; nothing here touches sample data except to copy it.
"""
import ctypes as C
import sys

import numpy as np
from scipy.signal import cheby1, ellip

from . import _native as nat

READ_SIZE = 131072  # bytes per chunk the reference's reader produces (src/misc/read_file.py:38)


def generate_domain(code: str):
    """Value range table of src/misc/read_file.py:177-196 (generateDomain)."""
    table = {'B': (0, 255), 'h': (-32768, 32767), 'b': (-128, 127), 'i': (-2147483648, 2147483647),
             'H': (0, 65536), 'I': (0, 4294967295)}
    if code not in table:
        return None
    xmin, xmax = table[code]
    return xmin, 1 / (-xmin + xmax)


def decimator_sos(q: int) -> np.ndarray:
    return np.ascontiguousarray(cheby1(8, 0.05, 0.8 / q, output='sos'), dtype=np.float64)


def output_sos(fs_dec: int, omega_out, degree: int = 3) -> np.ndarray:
    return np.ascontiguousarray(
        ellip(degree, 1, 30, omega_out, btype='lowpass', analog=False, output='sos', fs=fs_dec), dtype=np.float64)


def faithful_f32_tables(q: int):
    """What scipy.signal.decimate builds for complex64 input (the reference's float32 records, dsp_processor.py:147,171-177):
    ``sos = np.asarray(cheby1(8, 0.05, 0.8/q, output='sos'), dtype=x.dtype)`` and, inside sosfiltfilt, ``sosfilt_zi(sos)`` --
    the same scipy calls on the same dtype, so the kernel starts from the reference's own single-precision tables."""
    from scipy.signal import sosfilt_zi
    sos = np.asarray(cheby1(8, 0.05, 0.8 / q, output='sos'), dtype=np.complex64)
    zi = sosfilt_zi(sos)
    if sos.imag.any() or zi.imag.any():
        raise ValueError('decimator tables with non-zero imaginary parts are not supported')
    return np.ascontiguousarray(sos.real, dtype=np.float32), np.ascontiguousarray(zi.real, dtype=np.float32)


def reference_shift_table(freqs, fs, n: int) -> np.ndarray:
    """The reference's shift table, evaluated as the reference evaluates it: ``exp(-2j*pi*(f/fs) * arange(n))`` per row
    (dsp_processor.py:185-187, vfo_processor.py:48,71-74).  Only the faithful float32 mode reads it: there the rounding of every
    shifted sample to complex64 must agree with the reference's, so the table has to be numpy's, not a better one."""
    return np.ascontiguousarray(np.stack([np.exp((-2j * np.pi * (f / fs)) * np.arange(n)) for f in freqs]), dtype=np.complex128)


def savgol_operator(window: int, polyorder: int = 3):
    """The linear operator of ``scipy.signal.savgol_filter(z, window, polyorder)`` (mode='interp'), which is what
    --smooth-output applies (src/dsp/dsp_processor.py:159-160), read off scipy's own result on unit impulses:
    (taps, tap_first, n_edge, left, right) as include/sdrterm_hip.h:sdrt_set_smoothing takes them."""
    from scipy.signal import savgol_filter
    size = 4 * window + 5
    op = savgol_filter(np.eye(size), window, polyorder, axis=0)     # op[i, j] = d out[i] / d in[j]
    n_edge = window // 2                                            # scipy _fit_edges_polyfit: halflen outputs per side
    left = np.ascontiguousarray(op[:n_edge, :window])
    right = np.ascontiguousarray(op[size - n_edge:, size - window:])
    mid = size // 2
    nz = np.nonzero(op[mid])[0]
    taps = np.ascontiguousarray(op[mid, nz[0]:nz[-1] + 1])
    return taps, int(nz[0] - mid), n_edge, left, right


_DEMOD_MODES = {'fm': 0, 'am': 1, 're': 2, 'im': 3}


def demodulate(mode: str, y: np.ndarray) -> np.ndarray:
    """fmDemod / amDemod / realOutput / imagOutput (src/dsp/demodulation.py:25-68) of decimated samples
    y (rows, m) on the GPU -> (rows, m) float64."""
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.complex128)
    out = np.empty(y.shape, dtype=np.float64)
    nat.check(nat.lib().sdrt_demodulate_host(_DEMOD_MODES[mode], y.ctypes.data, y.shape[0], y.shape[1], out.ctypes.data))
    return out


class Engine:
    """One GPU context == one reference processor (rows = VFO channels)."""

    def __init__(self, fs, decimation, mode, code='h', n_samples=None, row_freqs=None, apply_shift=None,
                 out_sos=None, omega_out=12500, normalize=False, correct_iq=False, impedance=50,
                 big_endian=False, device=0, smooth=0, row_modes=None, reserve_chunks=0, stage_a='auto', faithful_f32=False):
        """``stage_a``: 'auto' / 'i8' / 'f64' (include/sdrterm_hip.h: enum sdrt_stage_a).  ``faithful_f32``: float32 records in
        the reference's own complex64 arithmetic (sdrt_config.faithful_f32); no effect on the other encodings.
        ``row_modes``: None (every row demodulates as ``mode``, as the reference does, io_args.py:118) or one of 'fm' / 'am' per
        row -- FM and AM channels of a --simo run in one pass over the input (BASELINE configs[2])."""
        if mode not in nat.MODES:
            raise ValueError(f'unsupported demodulation mode {mode!r}')
        if code not in nat.DTYPES:
            raise ValueError(f'unsupported encoding {code!r}')
        self.fs = fs
        self.q = int(decimation)
        self.mode = mode
        self.code = code
        self.n = int(n_samples) if n_samples else READ_SIZE // (2 * nat.ITEM_SIZE[code])
        freqs = [0.0] if row_freqs is None else [float(f) for f in row_freqs]
        self.rows = len(freqs)
        if apply_shift is None:
            apply_shift = self.rows > 1 or any(f != 0 for f in freqs)
        self._freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        self._dec_sos = decimator_sos(self.q)
        if mode in ('re', 'im'):
            self._out_sos = np.zeros((0, 6))         # realOutput / imagOutput: no output filter (dsp_processor.py:130-138)
        else:
            self._out_sos = np.ascontiguousarray(
                out_sos if out_sos is not None else output_sos(int(fs) // self.q, omega_out), dtype=np.float64)
        dom = generate_domain(code) if normalize else None
        swap = bool(big_endian) != (sys.byteorder == 'big') and nat.ITEM_SIZE[code] > 1

        cfg = nat.Config()
        cfg.struct_size = C.sizeof(nat.Config)
        cfg.device = device
        cfg.fs = float(fs)
        cfg.decimation = self.q
        cfg.mode = nat.MODES[mode]
        cfg.dtype = nat.DTYPES[code]
        cfg.swap_endianness = int(swap)
        cfg.n_samples = self.n
        cfg.n_rows = self.rows
        cfg.row_freqs = self._freqs.ctypes.data_as(C.POINTER(C.c_double))
        cfg.apply_shift = int(bool(apply_shift))
        cfg.normalize = int(dom is not None)
        cfg.norm_xmin = float(dom[0]) if dom else 0.0
        cfg.norm_inv_range = float(dom[1]) if dom else 0.0
        cfg.correct_iq = int(bool(correct_iq))
        cfg.iq_inductance = impedance / fs
        cfg.dec_sos = self._dec_sos.ctypes.data_as(C.POINTER(C.c_double))
        cfg.dec_sections = self._dec_sos.shape[0]
        cfg.out_sos = self._out_sos.ctypes.data_as(C.POINTER(C.c_double))
        cfg.out_sections = self._out_sos.shape[0]
        self.row_modes = None
        if row_modes is not None:
            if len(row_modes) != self.rows or any(md not in ('fm', 'am') for md in row_modes):
                raise ValueError("row_modes: one of 'fm' / 'am' per row")
            self.row_modes = list(row_modes)
            self._row_modes = np.ascontiguousarray([nat.MODES[md] for md in row_modes], dtype=np.int32)
            cfg.row_modes = self._row_modes.ctypes.data_as(C.POINTER(C.c_int32))
        if stage_a not in nat.STAGE_A:
            raise ValueError(f"stage_a: one of {sorted(nat.STAGE_A)} (got {stage_a!r})")
        cfg.stage_a = nat.STAGE_A[stage_a]
        self.faithful_f32 = bool(faithful_f32) and code == 'f'
        if self.faithful_f32:
            self._f32_sos, self._f32_zi = faithful_f32_tables(self.q)
            cfg.faithful_f32 = 1
            cfg.f32_sos = self._f32_sos.ctypes.data_as(C.POINTER(C.c_float))
            cfg.f32_zi = self._f32_zi.ctypes.data_as(C.POINTER(C.c_float))
            if apply_shift:
                self._shift_table = reference_shift_table(freqs, fs, self.n)
                cfg.shift_table = self._shift_table.ctypes.data_as(C.POINTER(C.c_double))
        self._ctx = C.c_void_p()
        nat.check(nat.lib().sdrt_create(C.byref(cfg), C.byref(self._ctx)))
        self.m = nat.lib().sdrt_out_per_chunk(self._ctx)
        self.stage_a = {0: 'f64', 2: 'i8t', 3: 'i8w', 4: 'seq', 5: 'f32c', 6: 'sq'}[nat.lib().sdrt_stage_a_kind(self._ctx)]   # which stage-A kernel runs
        self.chunk_bytes = self.n * 2 * nat.ITEM_SIZE[code]
        if reserve_chunks:
            # device scratch for batches up to this size now, none inside the processing calls (include/sdrterm_hip.h: sdrt_reserve)
            nat.check(nat.lib().sdrt_reserve(self._ctx, int(reserve_chunks)))
        self.smooth = int(smooth) if smooth else 0
        if smooth:
            taps, first, n_edge, left, right = savgol_operator(int(smooth))
            nat.check(nat.lib().sdrt_set_smoothing(self._ctx, int(smooth), taps.size, first, taps.ctypes.data, n_edge,
                                                   left.ctypes.data, right.ctypes.data))

    # -- processing ---------------------------------------------------------------------------------
    def process(self, raw, n_chunks=None) -> np.ndarray:
        """Host buffers in, host array out: (rows, n_chunks*m) float64 (row r == stream of VFO r)."""
        buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else \
            np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        if n_chunks is None:
            if buf.size % self.chunk_bytes:
                raise ValueError(f'input is not a whole number of {self.chunk_bytes}-byte chunks')
            n_chunks = buf.size // self.chunk_bytes
        elif buf.size < n_chunks * self.chunk_bytes:
            raise ValueError(f'{n_chunks} chunks need {n_chunks * self.chunk_bytes} bytes, got {buf.size}')
        out = np.empty((self.rows, n_chunks * self.m), dtype=np.float64)
        nat.check(nat.lib().sdrt_process_host(self._ctx, buf.ctypes.data, n_chunks, out.ctypes.data))
        return out

    def submit(self, raw, n_chunks=None):
        """Start a batch from host memory (copy-in, kernels, copy-out all asynchronous) and return a handle for
        ``collect``; up to three batches overlap (include/sdrterm_hip.h: sdrt_submit_host)."""
        buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else \
            np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        if n_chunks is None:
            if buf.size % self.chunk_bytes:
                raise ValueError(f'input is not a whole number of {self.chunk_bytes}-byte chunks')
            n_chunks = buf.size // self.chunk_bytes
        elif buf.size < n_chunks * self.chunk_bytes:
            raise ValueError(f'{n_chunks} chunks need {n_chunks * self.chunk_bytes} bytes, got {buf.size}')
        out = np.empty((self.rows, n_chunks * self.m), dtype=np.float64)
        t = C.c_int64(-1)
        nat.check(nat.lib().sdrt_submit_host(self._ctx, buf.ctypes.data, n_chunks, out.ctypes.data, C.byref(t)))
        return t.value, out, buf            # buf: page-locked input must outlive the copy

    def collect(self, handle) -> np.ndarray:
        ticket, out, _buf = handle
        nat.check(nat.lib().sdrt_wait(self._ctx, ticket))
        return out

    def process_device(self, raw_ptr: int, n_chunks: int, out_ptr: int, stream: int = 0, ydec_ptr: int = 0):
        """Device pointers in/out, asynchronous on ``stream`` (hipStream_t handle, 0 = default).  One stream per context:
        the decimated samples and the IQ prefix tables of a call live in per-context scratch buffers."""
        if ydec_ptr:
            nat.check(nat.lib().sdrt_process_device_debug(self._ctx, raw_ptr, n_chunks, out_ptr, ydec_ptr, stream or None))
        else:
            nat.check(nat.lib().sdrt_process_device(self._ctx, raw_ptr, n_chunks, out_ptr, stream or None))

    def process_debug(self, raw, n_chunks=None):
        """(out, ydec): as process(), plus the decimated complex samples (rows, n_chunks*m) complex128."""
        buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else \
            np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        if n_chunks is None:
            n_chunks = buf.size // self.chunk_bytes
        elif buf.size < n_chunks * self.chunk_bytes:
            raise ValueError(f'{n_chunks} chunks need {n_chunks * self.chunk_bytes} bytes, got {buf.size}')
        d_raw = nat.DeviceBuffer(n_chunks * self.chunk_bytes).upload(buf[:n_chunks * self.chunk_bytes])
        d_out = nat.DeviceBuffer(self.rows * n_chunks * self.m * 8)
        d_y = nat.DeviceBuffer(self.rows * n_chunks * self.m * 16)
        self.process_device(d_raw.ptr, n_chunks, d_out.ptr, 0, d_y.ptr)
        nat.check(nat.lib().sdrt_device_synchronize())
        out = d_out.download(np.empty((self.rows, n_chunks * self.m), dtype=np.float64))
        y = d_y.download(np.empty((self.rows, n_chunks * self.m), dtype=np.complex128))
        for b in (d_raw, d_out, d_y):
            b.free()
        return out, y

    # -- state / profiling --------------------------------------------------------------------------
    @property
    def iq_offset(self) -> complex:
        re, im = C.c_double(), C.c_double()
        nat.check(nat.lib().sdrt_get_iq_offset(self._ctx, C.byref(re), C.byref(im)))
        return complex(re.value, im.value)

    @iq_offset.setter
    def iq_offset(self, v: complex):
        nat.check(nat.lib().sdrt_set_iq_offset(self._ctx, float(v.real), float(v.imag)))

    def iq_range_device(self, d_raw: int, n_chunks: int, stream: int = 0):
        """(a, b): what these chunks do to the carried IQ offset, ``off_end = a * off_start + b``, without processing them and
        without touching this engine's own state (chunk-sharded streams: ``parallel.chain_iq_offsets``)."""
        a, re, im = C.c_double(), C.c_double(), C.c_double()
        nat.check(nat.lib().sdrt_iq_range_device(self._ctx, d_raw, n_chunks, C.byref(a), C.byref(re), C.byref(im), stream or None))
        return a.value, complex(re.value, im.value)

    def iq_range(self, raw, n_chunks=None):
        """As ``iq_range_device`` for a host buffer."""
        buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else \
            np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        if n_chunks is None:
            n_chunks = buf.size // self.chunk_bytes
        elif buf.size < n_chunks * self.chunk_bytes:
            raise ValueError(f'{n_chunks} chunks need {n_chunks * self.chunk_bytes} bytes, got {buf.size}')
        d_raw = nat.DeviceBuffer(max(n_chunks * self.chunk_bytes, 16)).upload(buf[:n_chunks * self.chunk_bytes])
        try:
            return self.iq_range_device(d_raw.ptr, n_chunks)
        finally:
            d_raw.free()

    @property
    def refined_items(self) -> int:
        """stage_a='auto': (chunk, row) items done a second time in float64 so far (include/sdrterm_hip.h: sdrt_refined_items)."""
        n = C.c_ulonglong(0)
        nat.check(nat.lib().sdrt_refined_items(self._ctx, C.byref(n)))
        return int(n.value)

    def profile(self, enable: bool = True):
        nat.check(nat.lib().sdrt_profile_enable(self._ctx, int(enable)))

    def profile_reset(self):
        nat.check(nat.lib().sdrt_profile_reset(self._ctx))

    def kernel_time(self, which: int):
        ms, cnt = C.c_double(), C.c_int64()
        nat.check(nat.lib().sdrt_kernel_time(self._ctx, which, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            nat.lib().sdrt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
