"""ctypes front-end for oracle/chain.c (ORACLE -- test infrastructure, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'liboracle.so')

CODES = {'b': 0, 'B': 1, 'h': 2, 'H': 3, 'i': 4, 'I': 5, 'f': 6, 'd': 7}
MODES = {'fm': 0, 'am': 1}
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'chain.c')
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.run(['make', '-C', _HERE, '-s'], check=True)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_process_stream.restype = C.c_int
        _lib.orc_process_chunk.restype = C.c_int
        _lib.orc_resample.restype = C.c_int
        _lib.orc_decimate.restype = C.c_int
        _lib.orc_fm.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
        _lib.orc_process_stream_f32c.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_threads() -> int:
    return lib().orc_max_threads()


def resample(x: np.ndarray, num: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(num)
    rc = lib().orc_resample(_p(x), C.c_size_t(x.size), C.c_size_t(num), _p(out))
    assert rc == 0
    return out


def decimate(x: np.ndarray, q: int, sos: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.complex128)
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    y = np.empty((x.size + q - 1) // q, dtype=np.complex128)
    rc = lib().orc_decimate(_p(sos), C.c_int(sos.shape[0]), _p(x), C.c_size_t(x.size), C.c_int(q), _p(y))
    if rc:
        raise ValueError(f'orc_decimate rc={rc}')
    return y


def fm(y: np.ndarray) -> np.ndarray:
    y = np.ascontiguousarray(y, dtype=np.complex128)
    out = np.empty(y.size)
    assert lib().orc_fm(_p(y), C.c_size_t(y.size), _p(out)) == 0
    return out


def am(y: np.ndarray) -> np.ndarray:
    y = np.ascontiguousarray(y, dtype=np.complex128)
    out = np.empty(y.size)
    lib().orc_am(_p(y), C.c_size_t(y.size), _p(out))
    return out


def shift_table(freqs, fs: float, n: int) -> np.ndarray:
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    tbl = np.empty((freqs.size, n), dtype=np.complex128)
    for k, f in enumerate(freqs):
        lib().orc_shift_table(C.c_double(f), C.c_double(fs), C.c_size_t(n), _p(tbl[k]))
    return tbl


def process_stream(raw: bytes, code: str, n: int, q: int, mode: str, shift, dec_sos, out_sos, *,
                   big_endian=False, norm=None, iq_inductance=None, off=0j, threads=1, smooth=0):
    """Returns (out[chunks, rows, m], carried offset).  ``norm`` = (xmin, 1/(xmax-xmin)) or None."""
    import sys
    item = np.dtype({'b': 'i1', 'B': 'u1', 'h': 'i2', 'H': 'u2', 'i': 'i4', 'I': 'u4', 'f': 'f4', 'd': 'f8'}[code]).itemsize
    buf = np.frombuffer(raw, dtype=np.uint8)
    n_chunks = buf.size // (2 * item * n)
    rows = 1 if shift is None else shift.shape[0]
    m = n // q
    dec_sos = np.ascontiguousarray(dec_sos, dtype=np.float64)
    out_sos = np.ascontiguousarray(out_sos, dtype=np.float64)
    if shift is not None:
        shift = np.ascontiguousarray(shift, dtype=np.complex128)
    out = np.empty((n_chunks, rows, m))
    offv = np.array([off], dtype=np.complex128)
    swap = int(bool(big_endian) != (sys.byteorder == 'big')) if item > 1 else 0
    rc = lib().orc_process_stream(
        _p(buf), C.c_int(CODES[code]), C.c_int(swap), C.c_size_t(n), C.c_size_t(n_chunks), C.c_int(q),
        C.c_int(MODES[mode]), _p(shift) if shift is not None else None, C.c_int(rows),
        _p(dec_sos), C.c_int(dec_sos.shape[0]), _p(out_sos), C.c_int(out_sos.shape[0]),
        C.c_int(0 if norm is None else 1), C.c_double(0.0 if norm is None else norm[0]),
        C.c_double(0.0 if norm is None else norm[1]),
        C.c_int(0 if iq_inductance is None else 1), C.c_double(iq_inductance or 0.0), _p(offv), _p(out),
        C.c_int(threads))
    if rc:
        raise ValueError(f'orc_process_stream rc={rc}')
    if smooth:
        # --smooth-output, dsp_processor.py:159-160: savgol_filter(z, smooth, 3) on every chunk's samples (single VFO)
        for c in range(n_chunks):
            for r in range(rows):
                row = np.ascontiguousarray(out[c, r])
                rc = lib().orc_savgol(_p(row), C.c_size_t(m), C.c_int(int(smooth)), C.c_int(3))
                if rc:
                    raise ValueError(f'orc_savgol rc={rc}')
                out[c, r] = row
    return out, complex(offv[0])


def faithful_f32_tables(q: int):
    """The tables scipy.signal.decimate works with when its input is complex64 (dsp_processor.py:147 on float32 records):
    the sos cast to the data's dtype and sosfilt_zi of THAT table -- the same scipy calls the reference's call makes."""
    from scipy import signal
    sos = np.asarray(signal.cheby1(8, 0.05, 0.8 / q, output='sos'), dtype=np.complex64)
    zi = signal.sosfilt_zi(sos)
    assert not sos.imag.any() and not zi.imag.any()
    return np.ascontiguousarray(sos.real, dtype=np.float32), np.ascontiguousarray(zi.real, dtype=np.float32)


def process_stream_f32c(raw: bytes, n: int, q: int, mode: str, shift, out_sos, *, big_endian=False, threads=1):
    """float32 records in the reference's own arithmetic (complex64 through the decimator, chain.c: orc_process_stream_f32c).
    ``shift`` must be the table numpy builds (ref_chain.shift_table), not c_chain.shift_table's.  Returns
    (out[chunks, rows, m] float64, ydec[chunks, rows, m] complex64)."""
    import sys
    buf = np.frombuffer(raw, dtype=np.uint8)
    n_chunks = buf.size // (8 * n)
    rows = 1 if shift is None else shift.shape[0]
    m = n // q
    sos32, zi32 = faithful_f32_tables(q)
    out_sos = np.ascontiguousarray(out_sos, dtype=np.float64)
    if shift is not None:
        shift = np.ascontiguousarray(shift, dtype=np.complex128)
    out = np.empty((n_chunks, rows, m))
    ydec = np.empty((n_chunks, rows, m), dtype=np.complex64)
    swap = int(bool(big_endian) != (sys.byteorder == 'big'))
    rc = lib().orc_process_stream_f32c(
        _p(buf), C.c_int(swap), C.c_size_t(n), C.c_size_t(n_chunks), C.c_int(q), C.c_int(MODES[mode]),
        _p(shift) if shift is not None else None, C.c_int(rows), _p(sos32), _p(zi32), C.c_int(4),
        _p(out_sos), C.c_int(out_sos.shape[0]), _p(out), _p(ydec), C.c_int(threads))
    if rc:
        raise ValueError(f'orc_process_stream_f32c rc={rc}')
    return out, ydec
