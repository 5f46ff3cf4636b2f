"""Drop-in mirrors of the reference's src/dsp package, backed by the HIP chain kernel."""
from .data_processor import DataProcessor  # noqa: F401
from .dsp_processor import DspProcessor  # noqa: F401
from .vfo_processor import VfoProcessor  # noqa: F401
