"""FFT taps: what SpectrumAnalyzerPlot.update and WaterfallPlot.update compute per chunk (src/plots/spectrum_analyzer_plot.py:74-88,
src/plots/waterfall_plot.py:47-58,90-100), through the C ABI (include/sdrterm_hip.h: sdrt_spectrum_host, sdrt_stft_host).

Both plots receive the reader's chunks, shift them by ``center`` (AbstractPlot: ``_omega = -2j*pi*(offset/fs)`` applied with
shiftFreq, abstract_plot.py:73-76) and transform them.  Chunks may be the reference reader's complex arrays or this
package's ``RawChunk`` (raw bytes: decode and --normalize-input run on the GPU as well).
"""
import sys

import numpy as np

from .. import _native as nat
from ..engine import generate_domain


def _describe(chunk):
    """(dtype code, swap, bytes, samples, normalize domain) of a Queue item."""
    if type(chunk).__name__ == 'RawChunk':
        swap = bool(chunk.bigEndian) != (sys.byteorder == 'big') and nat.ITEM_SIZE[chunk.code] > 1
        return chunk.code, swap, np.frombuffer(chunk.data, dtype=np.uint8), chunk.size, generate_domain(chunk.code) if chunk.normalize else None
    z = np.ascontiguousarray(chunk)
    code = {np.dtype(np.complex128): 'd', np.dtype(np.complex64): 'f'}[z.dtype]
    return code, False, z.view(np.uint8).reshape(-1), z.size, None


class SpectrumTap:
    """``amp`` of SpectrumAnalyzerPlot: log10(abs(fftshift(fftn(y, norm='forward')))**2) per chunk; ``freq`` as the plot builds it."""

    def __init__(self, fs: int, center: int = 0, **_):
        self.fs, self.offset = fs, center

    def __call__(self, chunk) -> np.ndarray:
        code, swap, buf, n, dom = _describe(chunk)
        out = np.empty(n, dtype=np.float64)
        nat.check(nat.lib().sdrt_spectrum_host(nat.DTYPES[code], int(swap), buf.ctypes.data, n, 1, float(self.offset), float(self.fs),
                                               int(dom is not None), float(dom[0]) if dom else 0.0, float(dom[1]) if dom else 0.0,
                                               out.ctypes.data))
        return out

    def freq(self, n: int) -> np.ndarray:
        return np.fft.fftshift(np.fft.fftfreq(n, 1 / self.fs))


class WaterfallTap:
    """``image`` of WaterfallPlot: 10 log10 |ShortTimeFFT.stft(y)|, shape (NFFT, n / NOOVERLAP + 1)."""
    _NPERSEG = 256
    _NOOVERLAP = _NPERSEG >> 1
    _NFFT = 1024

    def __init__(self, fs: int, center: int = 0, **_):
        from scipy.signal import ShortTimeFFT
        self.fs, self.offset = fs, center
        # the window exactly as the plot's transform scales it (waterfall_plot.py:47-54)
        sft = ShortTimeFFT.from_window(('kaiser', 5), fs, self._NPERSEG, self._NOOVERLAP, mfft=self._NFFT, fft_mode='centered',
                                       scale_to='magnitude', phase_shift=None)
        self._win = np.ascontiguousarray(sft.win, dtype=np.float64)

    def __call__(self, chunk) -> np.ndarray:
        code, swap, buf, n, dom = _describe(chunk)
        out = np.empty((self._NFFT, n // self._NOOVERLAP + 1), dtype=np.float64)
        nat.check(nat.lib().sdrt_stft_host(nat.DTYPES[code], int(swap), buf.ctypes.data, n, 1, float(self.offset), float(self.fs),
                                           int(dom is not None), float(dom[0]) if dom else 0.0, float(dom[1]) if dom else 0.0,
                                           self._NPERSEG, self._NOOVERLAP, self._NFFT, self._win.ctypes.data, out.ctypes.data))
        return out
