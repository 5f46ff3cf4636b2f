"""sdrterm_amd -- MI355X-native implementation of peads/sdrterm's src/dsp streaming hot path.

The compute lives in ``libsdrterm_hip.so`` (hand-written HIP for gfx950 behind the C ABI declared in
``include/sdrterm_hip.h``); this package is the thin Python host: a ctypes binding (``_native``) and
drop-in mirrors of the reference's ``DspProcessor`` / ``VfoProcessor`` (``sdrterm_amd.dsp``).
There is no CPU compute path: without the shared library and a GPU the processors raise.
"""
__version__ = '0.1.0'
