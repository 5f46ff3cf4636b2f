"""DspProcessor -- GPU-backed stand-in for the reference's src/dsp/dsp_processor.py.

The public surface is the reference's (dsp_processor.py:48-220): the constructor keywords that IOArgs passes
(io_args.py:111-118), the ``fs`` / ``decimation`` / ``decimatedFs`` properties with their error behaviour, the
``selectOutput*`` methods, ``_setDemod``, ``_generateShift`` and the ``processData(isDead, buffer, f)`` consumer.
What differs is the body: instead of numpy/scipy calls per chunk (dsp_processor.py:140-149) every queued chunk goes
through one launch of the fused HIP chain kernel behind the C ABI (include/sdrterm_hip.h).

Spawn safety: the CLI pickles the processor into a child process (sdrterm.py:182-188, io_args.py:92-95), so no
ctypes or HIP handle exists until ``processData`` runs there; the GPU context is created from the first chunk,
where the reference allocates its x/y/z buffers (dsp_processor.py:171-178).
"""
import json
import os
import sys
from queue import Empty

import numpy as np

from . import demodulation
from .data_processor import DataProcessor
from ..misc.general_util import eprint, vprint


class _EndOfStream(ValueError):
    """Raised where the reference's ``x[0, :] = b''`` raises ValueError (dsp_processor.py:170) to end the loop."""


def generateEllipFilter(fs: int, deg: int, Wn, btype: str):
    """Output lowpass design: the same scipy call as dsp_processor.py:39-45 (1 dB ripple, 30 dB stop band, SOS)."""
    from scipy.signal import ellip
    return ellip(deg, 1, 30, Wn, btype=btype, analog=False, output='sos', fs=fs)


#                 selector                 log line            bandwidth            has lowpass
_SELECTIONS = {
    'Fm': (demodulation.fmDemod, 'NFM Selected', 12500, True),          # dsp_processor.py:116-121
    'Am': (demodulation.amDemod, 'AM Selected', 10000, True),           # dsp_processor.py:123-128
    'Real': (demodulation.realOutput, 'I output Selected', None, False),  # dsp_processor.py:130-133
    'Imag': (demodulation.imagOutput, 'Q output Selected', None, False),  # dsp_processor.py:135-138
}


class DspProcessor(DataProcessor):
    _FILTER_DEGREE = 3
    MAX_BATCH = 64      # chunks per GPU call when that many are already queued (1 == the reference's cadence)

    def __init__(self, fs: int, center: int = 0, omegaOut: int = 0, tuned: int = 0, dec: int = 2,
                 smooth: bool = False, fileInfo: dict = None, **kwargs):
        self._decimationFactor = dec
        self._fs = None
        self._decimatedFs = None
        self.fs = fs
        self.centerFreq = center
        self.tunedFreq = tuned
        self.omegaOut = omegaOut
        self.smooth = smooth
        self.bandwidth = None
        self.tmp = None
        self._fileInfo = fileInfo
        self._shift = None
        self._isDead = False
        self._outputFilters = []
        self._nFreq = 1
        self._engine = None
        self._engineReader = {}
        self._device = int(kwargs.get('device') or 0)
        self._workingReIm = bool(kwargs.get('workingReIm') or os.environ.get('SDRTERM_WORKING_REIM') == '1')
        # which arithmetic the decimator's block products run in (include/sdrterm_hip.h: enum sdrt_stage_a): 'auto' / 'i8' / 'f64'
        self._stageA = str(kwargs.get('stageA') or 'auto')
        if self._stageA not in ('auto', 'i8', 'f64'):
            raise ValueError(f"stageA must be 'auto', 'i8' or 'f64', not {self._stageA!r}")
        # float32 input exactly as the reference computes it -- complex64 through scipy.signal.decimate (read_file.py:100-101,
        # dsp_processor.py:147,171-177): sdrt_config.faithful_f32.  Off: float32 input is filtered in float64 / exactly, which is
        # closer to the true result than the reference's own output and several times faster, but not identical to it
        self._faithfulF32 = bool(kwargs.get('faithfulF32') or os.environ.get('SDRTERM_FAITHFUL_F32') == '1')
        self._pinned = None           # page-locked input ring of _submitBatch (created in the process that runs processData)
        self._pinnedNext = 0

    # ---- rates ----------------------------------------------------------------------------------------
    @property
    def fs(self) -> int:
        return self._fs

    @fs.setter
    def fs(self, value: int) -> None:
        self._fs = value
        self._decimatedFs = value // self._decimationFactor     # integer division, as dsp_processor.py:86

    @property
    def decimation(self) -> int:
        return self._decimationFactor

    @decimation.setter
    def decimation(self, value: int) -> None:
        if value < 2:
            raise ValueError("Decimation must be at least 2.")
        self._decimationFactor = value
        self.fs = self._fs

    @property
    def decimatedFs(self) -> int:       # read-only: assignment raises AttributeError like the reference property
        return self._decimatedFs

    # ---- demodulation selection -------------------------------------------------------------------------
    def demod(self, *_, **__):
        """Placeholder until a selectOutput*() installs a selector (dsp_processor.py:103-104)."""

    def _setDemod(self, fun, *filters):
        """Installs ``fun`` as ``self.demod`` and replaces the output filters (dsp_processor.py:106-114):
        ValueError when ``fun`` is None, TypeError when the filters argument is not iterable."""
        if fun is None:
            raise ValueError("Demodulation function, or filters not defined")
        self._outputFilters.clear()
        if filters:
            self._outputFilters.extend(*filters)
        self.demod = fun
        self._engine = None
        return self.demod

    def _select(self, key: str) -> None:
        selector, message, bandwidth, lowpass = _SELECTIONS[key]
        vprint(message)
        self.bandwidth = bandwidth if bandwidth is not None else self.decimatedFs
        if lowpass:
            self._setDemod(selector, generateEllipFilter(self._decimatedFs, self._FILTER_DEGREE, self.omegaOut, 'lowpass'))
        else:
            self._setDemod(selector)

    def selectOutputFm(self):
        self._select('Fm')

    def selectOutputAm(self):
        self._select('Am')

    def selectOutputReal(self):
        self._select('Real')

    def selectOutputImag(self):
        self._select('Imag')

    # ---- GPU context --------------------------------------------------------------------------------------
    def _rowFreqs(self):
        return [self.centerFreq]

    def _applyShift(self) -> bool:
        return bool(self.centerFreq)

    def _rowModes(self):
        return None

    @staticmethod
    def _encodingOf(chunk) -> str:
        """Queue items are what the reference reader's ``re + 1j*im`` produced (complex128, or complex64 for float32 input),
        or the raw reads of this package's reader (misc/read_file.py: RawChunk), whose per-sample steps the GPU does."""
        from ..misc.read_file import RawChunk
        if isinstance(chunk, RawChunk):
            return chunk.code
        if chunk.dtype == np.complex128:
            return 'd'
        if chunk.dtype == np.complex64:
            return 'f'
        raise TypeError(f'chunks must be complex64 or complex128 arrays, not {chunk.dtype}')

    @staticmethod
    def _readerSteps(chunk) -> dict:
        """Engine options for the reader's per-sample steps (read_file.py:100-103) when the chunk is raw."""
        from ..misc.read_file import RawChunk
        if not isinstance(chunk, RawChunk):
            return {}
        return dict(normalize=chunk.normalize, correct_iq=chunk.correctIq, impedance=chunk.impedance, big_endian=chunk.bigEndian)

    def _getEngine(self, n: int, code: str, smoothed: bool = True, **reader):
        """The GPU context for chunks of n samples of this encoding.  ``smoothed``: with --smooth-output applied
        (_transformData, dsp_processor.py:159-160); _processChunk itself never smooths in the reference.  ``reader``: the
        reader's per-sample steps for raw chunks (_readerSteps)."""
        window = self._smoothing() if smoothed else 0
        eng = self._engine
        if eng is not None and eng.n == n and eng.code == code and eng.smooth == window and self._engineReader == reader:
            return eng
        mode = getattr(self.demod, 'mode', None)
        if mode not in ('fm', 'am', 're', 'im'):
            raise TypeError(f'{self.demod!r} is not a demodulation selector of this package')
        if not len(self._outputFilters) and not (mode in ('re', 'im') and self._workingReIm):
            # the reference gets here with an empty filter list and scipy.signal.sosfilt([]) raises
            # (dsp_processor.py:130-138,149); same exception type and text.  workingReIm=True (keyword or
            # SDRTERM_WORKING_REIM=1) writes the real / imaginary part of the decimated samples instead, which is what
            # selectOutputReal / selectOutputImag set out to do
            raise ValueError('sos array must be 2D')
        from ..engine import Engine
        sos = np.asarray(self._outputFilters, dtype=np.float64).reshape(-1, 6) if len(self._outputFilters) else None
        self._engine = Engine(self._fs, self._decimationFactor, mode, code, n_samples=n, row_freqs=self._rowFreqs(),
                              apply_shift=self._applyShift(), out_sos=sos, device=self._device, smooth=window,
                              row_modes=self._rowModes(), reserve_chunks=self.MAX_BATCH, stage_a=self._stageA,
                              faithful_f32=self._faithfulF32, **reader)
        self._engineReader = reader
        return self._engine

    def _smoothing(self) -> int:
        """--smooth-output window (savgol_filter(z, smooth, 3) in _transformData, dsp_processor.py:159-160); 0 = off."""
        return int(self.smooth) if self.smooth else 0

    # ---- per-chunk API kept for callers/tests written against the reference -----------------------------------
    def _processChunk(self, x: np.ndarray, y: np.ndarray, z: np.ndarray) -> None:
        """x (nFreq, n) complex with the chunk in row 0; fills y (nFreq, n/q) decimated samples and z (nFreq, n/q) output."""
        eng = self._getEngine(x.shape[1], self._encodingOf(x), smoothed=False)
        out, dec = eng.process_debug(np.ascontiguousarray(x[0]), 1)
        y[:] = dec
        z[:] = out

    def _transformData(self, x, y, z, file) -> None:
        self._processChunk(x, y, z)
        if self._smoothing():
            # same operator the batched path applies on the GPU (Engine(smooth=...)), here on one chunk
            eng = self._getEngine(x.shape[1], self._encodingOf(x), smoothed=True)
            z[:] = eng.process(np.ascontiguousarray(x[0]), 1).reshape(z.shape)
        file.write(np.ascontiguousarray(z, dtype='=f8').tobytes())     # == pack('@Nd', *z.flat), dsp_processor.py:162

    # ---- consumer loop ----------------------------------------------------------------------------------------
    def _takeBatch(self, buffer):
        """One blocking get, then whatever is already queued (at most MAX_BATCH).  Returns (chunks, saw_eof).
        The reader signals EOF with b'' (read_file.py:169-171)."""
        got = [buffer.get()]
        while self._isChunk(got[-1]) and len(got) < self.MAX_BATCH:
            try:
                got.append(buffer.get_nowait())
            except (Empty, ValueError, OSError):
                break
        eof = not self._isChunk(got[-1])
        return (got[:-1] if eof else got), eof

    @staticmethod
    def _isChunk(item) -> bool:
        return isinstance(item, np.ndarray) or type(item).__name__ == 'RawChunk'

    def _submitBatch(self, chunks):
        """Starts the batch on the GPU (copy-in, kernels and copy-out are asynchronous: Engine.submit) and returns a handle for
        ``_collectBatch``: the next batch's copy-in then overlaps this one's kernels."""
        eng = self._getEngine(chunks[0].size, self._encodingOf(chunks[0]), **self._readerSteps(chunks[0]))
        raw = self._stage(eng, chunks)
        return eng, eng.submit(raw, len(chunks)), len(chunks)

    PINNED_RING = 4     # > the batches the library keeps in flight (include/sdrterm_hip.h: sdrt_submit_host, three slots)

    def _stage(self, eng, chunks) -> np.ndarray:
        """The batch's bytes, chunk after chunk, in one of PINNED_RING page-locked buffers (sdrt_host_alloc): the copy to the GPU
        then runs asynchronously at the full PCIe rate straight from where the chunks were put, instead of through a pageable
        ``b''.join`` that the runtime has to stage once more (pinned double buffering on the input
        side).  A buffer is reused PINNED_RING submits later, when the library has long completed the batch that read it (a
        fourth submit first finishes the oldest of its three slots)."""
        from .. import _native as nat
        cb = eng.chunk_bytes
        if self._pinned is None or self._pinned[0].size < self.MAX_BATCH * cb:
            self._pinned = [nat.pinned_array(self.MAX_BATCH * cb) for _ in range(self.PINNED_RING)]
            self._pinnedNext = 0
        buf = self._pinned[self._pinnedNext]
        self._pinnedNext = (self._pinnedNext + 1) % self.PINNED_RING
        raw_items = bool(self._readerSteps(chunks[0]))
        for k, c in enumerate(chunks):
            src = np.frombuffer(c.data, dtype=np.uint8) if raw_items else np.ascontiguousarray(c).view(np.uint8).reshape(-1)
            if src.size != cb:
                raise ValueError(f'chunk {k} of the batch has {src.size} bytes, the batch was opened with {cb}-byte chunks')
            buf[k * cb:(k + 1) * cb] = src
        return buf[:len(chunks) * cb]

    def _collectBatch(self, pending) -> np.ndarray:
        eng, handle, n = pending
        return eng.collect(handle).reshape(self._nFreq, n, eng.m)

    def _runBatch(self, chunks) -> np.ndarray:
        return self._collectBatch(self._submitBatch(chunks))

    def _emit(self, z: np.ndarray, file) -> None:
        """z (nFreq, chunks, m): the reference writes chunk after chunk, native-endian doubles."""
        for c in range(z.shape[1]):
            file.write(np.ascontiguousarray(z[:, c, :], dtype='=f8').tobytes())

    @staticmethod
    def _moreQueued(buffer) -> bool:
        """Is another Queue item waiting already?  (Anything that cannot tell -- a closed Queue, no Queue at all -- says no:
        the batch in flight is then written out at once.)"""
        if buffer is None:
            return False
        try:
            return not buffer.empty()
        except (OSError, ValueError, NotImplementedError, AttributeError):
            return False

    def _halted(self, isDead) -> bool:
        """The reference's loop condition (dsp_processor.py:168): the processor's own flag or the shared halt flag."""
        return bool(self._isDead or isDead.value)

    def _processData(self, isDead, buffer, file=None) -> None:
        pending = None              # a batch whose kernels run while the next one is taken from the Queue and copied in

        def flush():
            # the batch leaves `pending` before it is written: a failing write (broken pipe, a HIP error surfacing in
            # sdrt_wait) must not see the same batch collected and written a second time on the way out
            nonlocal pending
            p, pending = pending, None
            if p is not None:
                self._emit(self._collectBatch(p), file)

        try:
            while not self._halted(isDead):
                chunks, eof = self._takeBatch(buffer)
                if chunks:
                    if self._shift is None:
                        self._generateShift(chunks[0].size)
                    started = self._submitBatch(chunks)
                    flush()
                    pending = started
                    if eof or not self._moreQueued(buffer):      # nothing to overlap with: write it out now (latency)
                        flush()
                if eof:
                    flush()             # (the end marker alone in this batch: the one before it is still in flight)
                    # the reference's loop ends on the ValueError numpy raises for ``x[0, :] = b''``
                    # (dsp_processor.py:170, swallowed by processData's finally/return)
                    raise _EndOfStream('end of stream')
        except KeyboardInterrupt:
            # Ctrl-C while waiting for the Queue (or a halt seen there): the reference has written every chunk it processed
            # before it leaves (dsp_processor.py:162-183) -- the batch in flight goes out too.  (flush() empties `pending`
            # first, so an interrupt raised by the write itself does not write it twice.)
            flush()
            raise
        flush()                     # halt flag: what is in flight still goes out; an error leaves it behind

    def _generateShift(self, c: int) -> None:
        """Host copy of the NCO row for callers that inspect ``_shift`` (dsp_processor.py:185-187: only when
        center != 0).  The kernel applies its own extended-precision table of the same sequence."""
        if self.centerFreq:
            k = np.arange(c)
            self._shift = np.array([np.exp(-2j * np.pi * (self.centerFreq / self._fs) * k)])

    def processData(self, isDead, buffer, f: str, *args, **kwargs) -> None:
        sink = open(f, 'wb') if f is not None else open(sys.stdout.fileno(), 'wb', closefd=False)
        with sink as file:
            try:
                self._processData(isDead, buffer, file)
                file.flush()
            except KeyboardInterrupt:
                pass
            except _EndOfStream:
                file.flush()
            except Exception:
                # the reference swallows everything here; a GPU/library failure must at least be visible
                import traceback
                eprint(f'{self} stopped on an error:')
                traceback.print_exc(file=sys.stderr)
            finally:
                # like the reference, whatever ended the loop is swallowed here -- unless closing the queue fails
                buffer.close()
                buffer.join_thread()
                vprint('Standard writer halted')
                return

    # ---- reporting -----------------------------------------------------------------------------------------------
    def __repr__(self):
        """JSON of the public settings; example_simo.sh parses it, so the keys follow dsp_processor.py:206-217."""
        shown = {}
        for key, value in self.__dict__.items():
            if key.startswith('_') or value is None or callable(value) or isinstance(value, np.ndarray):
                continue
            shown[key[:-3] if 'Str' in key else key] = value
        shown['encoding'] = str(self._fileInfo['bitsPerSample'])
        shown['fs'] = self._fs
        shown['decimatedFs'] = self._decimatedFs
        return json.dumps(shown, indent=2)

    def __str__(self):
        return type(self).__name__

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_engine'] = None       # ctypes / HIP handles never cross the spawn boundary
        state['_pinned'] = None       # ... nor does page-locked memory
        return state
