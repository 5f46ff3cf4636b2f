"""VfoProcessor -- the ``--simo`` multi-channel processor, GPU-backed mirror of src/dsp/vfo_processor.py.

Interface kept from the reference class: constructor ``VfoProcessor(fs, vfoHost='localhost', vfos=csv, **kw)``,
``ValueError`` without vfos, ``host:port`` parsing (a free port is picked when none is given), attributes
``vfos`` / ``vfosStr`` / ``host`` / ``port``, and ``processData(isDead, buffer)`` that serves one TCP client per row.

Rows are laid out as the reference lays them out (vfo_processor.py:42-47): one per csv entry at
``int(v) + center``, then ``center`` itself last.  All rows of all queued chunks come from one launch of the GPU
chain kernel (the NCO rows live in the kernel's tables); the socket side is ``fanout.RowFanOut``.
"""
import numpy as np

from .dsp_processor import DspProcessor, _EndOfStream
from .fanout import RowFanOut
from ..misc.general_util import eprint, findPort, vprint


def _parse_host(spec: str):
    if ':' in spec:
        host, port = spec.split(':')
        return host, int(port)
    return spec, findPort(spec)


class VfoProcessor(DspProcessor):

    def __init__(self, fs, vfoHost: str = 'localhost', vfos: str = None, **kwargs):
        super().__init__(fs, **kwargs)
        if vfos is None or len(vfos) < 1:
            raise ValueError('simo mode cannot be used without the vfos option')
        self.vfosStr = vfos + ',0'                      # what the reference reports in its JSON repr
        offsets = [int(v) for v in vfos.split(',')]
        self.vfos = np.array([o + self.centerFreq for o in offsets] + [self.centerFreq])
        self._nFreq = len(self.vfos)
        self.host, self.port = _parse_host(vfoHost)
        self._fanout = None

    # every row goes through the NCO, the center row included (the reference multiplies it by exp(0) = 1 when
    # center == 0, vfo_processor.py:72-74)
    def _rowFreqs(self):
        return [float(v) for v in self.vfos]

    def _applyShift(self) -> bool:
        return True

    @property
    def clients(self):
        return self._fanout.clients if self._fanout is not None else None

    def _generateShift(self, c: int) -> None:
        """The reference builds its (rows, c) phasor table here and then blocks until every row has a client
        (vfo_processor.py:71-78).  The table itself lives in the GPU context; the blocking handshake is kept."""
        self._shift = self.vfos            # marks "done" for the consumer loop; the per-sample table is on the GPU
        if self._fanout is not None:
            if not self._fanout.wait_for_clients(lambda: self._isDead or bool(self._halt.value)):
                raise KeyboardInterrupt

    def _emit(self, z: np.ndarray, _file=None) -> None:
        for c in range(z.shape[1]):
            self._fanout.send_rows(z[:, c, :])

    def processData(self, isDead, buffer, *args, **kwargs) -> None:
        self._halt = isDead
        self._fanout = RowFanOut(self.host, self.port, self._nFreq)
        try:
            self._fanout.start()
            self._processData(isDead, buffer)
        except (KeyboardInterrupt, _EndOfStream):
            pass
        except Exception:
            import sys
            import traceback
            eprint(f'{self} stopped on an error:')
            traceback.print_exc(file=sys.stderr)
        finally:
            self._isDead = True
            self._fanout.close()
            vprint('Multi-VFO writer halted')
            return

    def __getstate__(self):
        state = super().__getstate__()
        state['_fanout'] = None
        state.pop('_halt', None)
        return state
