"""GPU taps for the reference's plots (src/plots): the arrays the spectrum analyser and the waterfall draw, computed on the
GPU from raw chunks.  The windows themselves (pyqtgraph) are outside this package's scope."""
from .taps import SpectrumTap, WaterfallTap  # noqa: F401
