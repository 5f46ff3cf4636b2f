"""Demodulation selectors (mirror of the names in src/dsp/demodulation.py:25-68).

In the reference these are numba kernels that ``DspProcessor._setDemod`` installs as ``self.demod`` and that
``_processChunk`` calls on the decimated samples.  Here the discriminators run inside the fused GPU chain
kernel, so these objects are picklable *selectors*: ``DspProcessor`` maps the installed one to the kernel's
mode.  Calling one directly runs the same GPU code on already-decimated samples (see ``_standalone``).
"""
import numpy as np


class _Demod:
    def __init__(self, name, mode):
        self.__name__ = name
        self.mode = mode

    def __call__(self, data, out):
        return _standalone(self.mode, data, out)

    def __repr__(self):
        return f'<gpu demodulator {self.__name__}>'

    def __reduce__(self):
        return (_lookup, (self.__name__,))


def _lookup(name):
    return globals()[name]


def _standalone(mode, data, out):
    """Run the discriminator of the chain kernel on decimated samples (rows, m) -> out (rows, m)."""
    from ..engine import demodulate
    data = np.atleast_2d(np.asarray(data, dtype=np.complex128))
    res = demodulate(mode, data)
    np.asarray(out)[...] = res.reshape(np.asarray(out).shape)
    return out


fmDemod = _Demod('fmDemod', 'fm')        # demodulation.py:25-38
amDemod = _Demod('amDemod', 'am')        # demodulation.py:41-48
realOutput = _Demod('realOutput', 're')  # demodulation.py:51-58
imagOutput = _Demod('imagOutput', 'im')  # demodulation.py:61-68
