"""ORACLE (test infrastructure, not product code): numpy/scipy restatement of the sdrterm DSP hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (sdrterm_amd/) never does.  Every function restates one step of the reference
(peads/sdrterm @ 2024_08_07, paths relative to /root/reference) and cites the lines it follows.
The third-party arithmetic (scipy.signal.decimate / resample / sosfilt / ellip / cheby1) is called
exactly as the reference calls it, so on the same scipy this module IS the reference arithmetic.

Pinned by tests/test_oracle_golden.py against tests/golden/reference_vectors.npz, which was
produced by importing and running the real reference (tests/golden/make_golden.py).
"""
import numpy as np
from scipy import signal

READ_SIZE = 131072  # src/misc/read_file.py:38

_NP = {'b': 'i1', 'B': 'u1', 'h': 'i2', 'H': 'u2', 'i': 'i4', 'I': 'u4', 'f': 'f4', 'd': 'f8'}


def np_dtype(code: str, big_endian: bool = False) -> np.dtype:
    """src/misc/file_util.py:44-56 (DataType) with the byte order the header/socket implies."""
    return np.dtype(('>' if big_endian else '<') + _NP[code]) if _NP[code][1] != '1' else np.dtype(_NP[code])


def samples_per_chunk(code: str) -> int:
    """readSize bytes of interleaved (re, im) records, src/misc/read_file.py:51-52,119-124."""
    return READ_SIZE // (2 * np.dtype(_NP[code]).itemsize)


def generate_domain(code: str):
    """src/misc/read_file.py:177-196 -- (xmin, 1/(xmax - xmin)) or None for float encodings."""
    table = {'B': (0, 255), 'h': (-32768, 32767), 'b': (-128, 127), 'i': (-2147483648, 2147483647),
             'H': (0, 65536), 'I': (0, 4294967295), 'L': (0, 18446744073709551615),
             'l': (-9223372036854775808, 9223372036854775807)}
    if code not in table:
        return None
    xmin, xmax = table[code]
    return xmin, 1 / (-xmin + xmax)


def unpack(raw: bytes, code: str, big_endian: bool = False) -> np.ndarray:
    """src/misc/read_file.py:51,100-101 -- structured (re, im) records -> complex128."""
    dt = np_dtype(code, big_endian)
    y = np.frombuffer(raw, dtype=np.dtype([('re', dt), ('im', dt)]))
    return y['re'] + 1j * y['im']


def normalize(z: np.ndarray, code: str) -> np.ndarray:
    """src/misc/read_file.py:82-96 -- 1.6*(z - xmin)*d - 0.8 on the COMPLEX value (only Re is offset)."""
    dom = generate_domain(code)
    if dom is None:
        return z
    xmin, d = dom
    return 1.6 * (z - xmin) * d - 0.8


def correct_iq(z: np.ndarray, off: complex, fs: float, impedance: float = 50):
    """src/misc/read_file.py:64-77 (== extra/src/iq_correction.pyx:45-52): first-order DC tracker.

    Returns (corrected z, carried offset).  Sequential by definition.
    """
    L = impedance / fs
    zz = np.asarray(z, dtype=np.complex128)            # the gufunc's signature is complex128 (read_file.py:66)
    out = np.empty(zz.shape, dtype=np.complex128)
    off = complex(off)
    for i in range(zz.shape[0]):
        v = complex(zz[i]) - off
        out[i] = v
        off = off + v * L
    return out, off


def correct_iq_fast(z: np.ndarray, off: complex, fs: float, impedance: float = 50):
    """Same recurrence through scipy.signal.lfilter (off' = (1-L) off + L z); used for long streams."""
    L = impedance / fs
    # off_i = (1-L)^i off_0 + L * sum_{j<i} (1-L)^(i-1-j) z_j  == lfilter([0, L], [1, -(1-L)], z) + decay
    n = z.shape[0]
    offs, zf = signal.lfilter([0.0, L], [1.0, -(1.0 - L)], z, zi=np.array([off], dtype=np.complex128))
    # offs[i] is the offset BEFORE sample i
    out = z - offs
    off_end = (1.0 - L) * offs[-1] + L * z[-1]
    return out, complex(off_end)


def shift_table(freqs, fs: float, n: int) -> np.ndarray:
    """dsp_processor.py:185-187 and vfo_processor.py:48,71-74 -- exp(-2j*pi*(f/fs)*arange(n)) per row."""
    freqs = np.atleast_1d(np.asarray(freqs))
    omega = -2j * np.pi * (freqs / fs)
    return np.stack([np.exp(w * np.arange(n)) for w in omega])


def vfo_rows(center: int, vfos_csv):
    """vfo_processor.py:42-47 -- rows are int(v)+center for each csv entry, then center LAST."""
    if not vfos_csv:
        return [center]
    rows = [int(v) + center for v in vfos_csv.split(',')]
    rows.append(center)
    return rows


def ellip_sos(fs_dec: int, omega_out: float) -> np.ndarray:
    """dsp_processor.py:39-45,116-128 -- ellip(3, 1, 30, omegaOut, 'lowpass', fs=decimatedFs) as SOS."""
    return signal.ellip(3, 1, 30, omega_out, btype='lowpass', analog=False, output='sos', fs=fs_dec)


def cheby_sos(q: int) -> np.ndarray:
    """scipy.signal.decimate default IIR (called at dsp_processor.py:147): cheby1(8, 0.05, 0.8/q)."""
    return signal.cheby1(8, 0.05, 0.8 / q, output='sos')


def fm_demod(y: np.ndarray) -> np.ndarray:
    """demodulation.py:25-38 -- angle(y[2j] * conj(y[2j+1])) for pairs, then FFT-resample m/2 -> m."""
    m = y.shape[-1]
    a = np.angle(y[..., 0::2] * np.conj(y[..., 1::2]))
    return np.stack([signal.resample(row[:m >> 1], m) for row in a.reshape(-1, m >> 1)]).reshape(y.shape)


def am_demod(y: np.ndarray) -> np.ndarray:
    """demodulation.py:41-48 -- abs(square(d))."""
    return np.abs(np.square(y))


def process_chunk(z: np.ndarray, q: int, mode: str, shift, out_sos, return_decimated: bool = False,
                  faithful_dtype: bool = True, smooth: int = 0):
    """dsp_processor.py:140-149: shift -> decimate -> demod -> output sosfilt, one chunk, all rows.

    ``smooth``: window length of --smooth-output (0 = off), applied after the output lowpass (_transformData).

    ``faithful_dtype``: the reference allocates x and y with the dtype of the first Queue item
    (dsp_processor.py:172-177).  ``re + 1j*im`` is complex64 for float32 input (numpy promotion,
    read_file.py:101), so for that encoding the reference stores the shifted samples in complex64
    and scipy runs the whole zero-phase IIR in SINGLE precision (decimate keeps x.dtype).  With
    faithful_dtype=False the chain is evaluated in complex128 for every encoding (what the fp64
    build computes; for float32 input it differs from the reference by the reference's own
    single-precision rounding, ~2e-4 relative on the decimated samples).
    """
    wdt = z.dtype if (faithful_dtype and z.dtype == np.complex64) else np.complex128
    zz = z.astype(np.complex128)
    x = (zz[None, :] * shift if shift is not None else zz[None, :]).astype(wdt)
    y = signal.decimate(x, q)
    y = y.astype(wdt).astype(np.complex128)  # numba kernels take complex128 (demodulation.py:25,41)
    if mode == 'fm':
        d = fm_demod(y)
    elif mode == 'am':
        d = am_demod(y)
    else:
        raise ValueError(mode)
    out = signal.sosfilt(out_sos, d)
    if smooth:
        # --smooth-output, dsp_processor.py:159-160: savgol_filter(z, smooth, 3) on what is about to be written
        out = signal.savgol_filter(out, smooth, 3)
    return (out, y) if return_decimated else out


class Chain:
    """Reader steps + processor steps for a stream of raw chunks (state: the IQ offset only)."""

    def __init__(self, fs, q, mode, code, center=0, vfos=None, omega_out=12500, normalize=False,
                 correct_iq=False, big_endian=False, impedance=50):
        self.fs, self.q, self.mode, self.code = fs, q, mode, code
        self.big_endian = big_endian
        self.normalize, self.correct = normalize, correct_iq
        self.impedance = impedance
        self.rows = vfo_rows(center, vfos)
        self.n = samples_per_chunk(code)
        self.simo = bool(vfos)
        if self.simo or center:
            self.shift = shift_table(self.rows, fs, self.n)
        else:
            self.shift = None
        self.out_sos = ellip_sos(fs // q, omega_out)
        self.off = 0j

    def read(self, raw: bytes) -> np.ndarray:
        z = unpack(raw, self.code, self.big_endian)
        if self.normalize:
            z = normalize(z, self.code)
        if self.correct:
            # the literal recurrence (a Python loop, ~20 ms per chunk): correct_iq_fast rounds differently by ~1e-11 relative
            z, self.off = correct_iq(z, self.off, self.fs, self.impedance)
        return z

    def process(self, raw: bytes, return_decimated=False, faithful_dtype=True):
        return process_chunk(self.read(raw), self.q, self.mode, self.shift, self.out_sos, return_decimated,
                             faithful_dtype)


def spectrum_tap(z: np.ndarray, fs: float, offset: float) -> np.ndarray:
    """plots/spectrum_analyzer_plot.py:74-88 -- shiftFreq with exp(-2j*pi*(offset/fs)*t) (abstract_plot.py:76,151-153), then
    amp = abs(fftshift(fftn(y, norm='forward'))); log10(amp * amp)."""
    from scipy.fft import fftn, fftshift
    y = np.array([z], dtype=np.complex128)
    y = y * np.array([np.exp(-2j * np.pi * (offset / fs) * np.arange(y.size))])
    amp = np.abs(fftshift(fftn(y, norm='forward')))
    return np.log10(amp * amp)[0]


def waterfall_tap(z: np.ndarray, fs: int, offset: float) -> np.ndarray:
    """plots/waterfall_plot.py:47-58,90-100 -- ShortTimeFFT(kaiser 5, 256, hop 128, mfft 1024, centered, magnitude) of the
    shifted chunk; image = 10 log10 |stft|, shape (1024, n/128 + 1)."""
    y = np.asarray(z, dtype=np.complex128) * np.exp(-2j * np.pi * (offset / fs) * np.arange(z.size))
    sft = signal.ShortTimeFFT.from_window(('kaiser', 5), fs, 256, 128, mfft=1024, fft_mode='centered', scale_to='magnitude',
                                          phase_shift=None)
    return 10.0 * np.log10(np.abs(sft.stft(y)))
