"""VfoProcessor -- the ``--simo`` multi-channel processor, GPU-backed mirror of src/dsp/vfo_processor.py.

Interface kept from the reference class: constructor ``VfoProcessor(fs, vfoHost='localhost', vfos=csv, **kw)``,
``ValueError`` without vfos, ``host:port`` parsing (a free port is picked when none is given), attributes
``vfos`` / ``vfosStr`` / ``host`` / ``port``, and ``processData(isDead, buffer)`` that serves one TCP client per row.

Rows are laid out as the reference lays them out (vfo_processor.py:42-47): one per csv entry at
``int(v) + center``, then ``center`` itself last.  All rows of all queued chunks come from one launch of the GPU
chain kernel (the NCO rows live in the kernel's tables); the socket side is ``fanout.RowFanOut``.

Several GPUs (``gpus=N`` keyword, or ``SDRTERM_GPUS``): the rows are split contiguously over N
processes, one per GPU.  The process the CLI started (io_args.py:92-95) is rank 0: it owns the reader's Queue, starts
the N-1 others, and hands every raw block to them (``parallel.BlockChannel``: RCCL broadcast between device buffers).
Rank r computes rows [lo_r, hi_r) only and serves them on TCP port ``port + r`` -- client k of that port gets row
lo_r + k -- with the reference's handshake per port (no processing before every row of every rank has its client: the
block exchange is collective, so one waiting rank holds all of them).
"""
import os

import numpy as np

from .dsp_processor import DspProcessor, _EndOfStream
from .fanout import RowFanOut
from ..misc.general_util import eprint, findPort, vprint


def _parse_host(spec: str):
    if ':' in spec:
        host, port = spec.split(':')
        return host, int(port)
    return spec, findPort(spec)


def _rank_main(processor, rank, world, uid, isDead):
    """Entry point of the helper processes (ranks 1 .. world-1): the same processor, pickled, on another GPU."""
    processor._enterRank(rank, world, uid)
    processor.processData(isDead, None)


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
        self._gpus = max(1, int(kwargs.get('gpus') or os.environ.get('SDRTERM_GPUS') or 1))
        self._rank, self._world, self._uid = 0, 1, None
        self._transportKind = kwargs.get('transport') or 'rccl'       # 'torch': an initialised torch.distributed group (tests)
        self._channel = None
        self._helpers = []

    # ---- this rank's share -------------------------------------------------------------------------------------
    def _enterRank(self, rank: int, world: int, uid=None) -> None:
        self._rank, self._world, self._uid = rank, world, uid
        if self._transportKind == 'rccl':
            self._device = rank                  # one process per GPU of the node
        self._engine = None

    def _shard(self):
        from ..parallel import SimoShard
        return SimoShard(len(self.vfos), self._rank, self._world)

    # every row goes through the NCO, the center row included (the reference multiplies it by exp(0) = 1 when
    # center == 0, vfo_processor.py:72-74)
    def _rowFreqs(self):
        sh = self._shard()
        return [float(v) for v in self.vfos[sh.lo:sh.hi]]

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

    # ---- several GPUs: block exchange ------------------------------------------------------------------------------
    def _openChannel(self, chunk_bytes: int):
        from .. import parallel
        if self._transportKind == 'torch':
            t = parallel.TorchTransport()
        else:
            t = parallel.RcclTransport(self._rank, self._world, self._device, self._uid)
        self._channel = parallel.BlockChannel(t, chunk_bytes)

    def _takeBatch(self, buffer):
        """Rank 0 takes the batch from the reader's Queue as the single-GPU processor does and sends it on; the other
        ranks receive it.  Blocks travel as raw bytes + what the engine needs to know about them."""
        if self._world == 1:
            return super()._takeBatch(buffer)
        if self._rank == 0:
            chunks, eof = super()._takeBatch(buffer)
            self._announce(chunks[0] if chunks else None)
            payload = b''.join(self._bytesOf(c) for c in chunks)
            blk = self._channel.exchange(payload, len(chunks), eof)
            self._eofSent = eof
        else:
            if self._channel is None:
                self._openChannel(1)
                import ast
                self._blockKind = ast.literal_eval(self._channel.t.bcast_bytes(None, 512).decode().strip())
                if self._blockKind:
                    self._channel.chunk_bytes = self._blockKind['chunk_bytes']
            blk = self._channel.exchange(None)
        if blk.n:
            blk.size = self._blockKind['samples']
        return ([blk] if blk.n else []), blk.eof

    def _submitBatch(self, chunks):
        if self._world == 1:
            return super()._submitBatch(chunks)
        return None, self._runBlock(chunks), 0             # several GPUs: the block exchange is synchronous already

    def _collectBatch(self, pending) -> np.ndarray:
        return super()._collectBatch(pending) if pending[0] is not None else pending[1]

    def _runBlock(self, chunks) -> np.ndarray:
        blk, k = chunks[0], self._blockKind
        reader = dict(normalize=k['normalize'], correct_iq=k['correctIq'], impedance=k['impedance'],
                      big_endian=k['bigEndian']) if k['raw'] else {}
        eng = self._getEngine(k['samples'], k['code'], **reader)
        if blk.dev_ptr:
            from .. import _native as nat
            if getattr(self, '_outDev', None) is None or self._outDev.nbytes < self._nFreq * blk.n * eng.m * 8:
                self._outDev = nat.DeviceBuffer(self._nFreq * self.MAX_BATCH * eng.m * 8)
            eng.process_device(blk.dev_ptr, blk.n, self._outDev.ptr)
            nat.check(nat.lib().sdrt_device_synchronize())
            out = self._outDev.download(np.empty((self._nFreq, blk.n * eng.m), dtype=np.float64))
        else:
            out = eng.process(blk.host, blk.n)
        return out.reshape(self._nFreq, blk.n, eng.m)

    def _announce(self, first_chunk) -> None:
        """Rank 0, once: open the channel (collective) and tell the others what the blocks hold (None: nothing will come)."""
        if self._channel is not None:
            return
        self._blockKind = self._describe(first_chunk) if first_chunk is not None else None
        self._openChannel(self._blockKind['chunk_bytes'] if self._blockKind else 1)
        self._channel.t.bcast_bytes(repr(self._blockKind).encode().ljust(512), 512)

    def _releaseRanks(self) -> None:
        """Rank 0 is leaving (end of stream already sent, halt flag, error): the other ranks wait in a collective and need
        an end-of-stream block to leave as well."""
        if self._world > 1 and self._rank == 0 and not getattr(self, '_eofSent', False):
            try:
                self._announce(None)
                self._channel.exchange(b'', 0, True)
            except Exception:
                pass
            self._eofSent = True

    @staticmethod
    def _bytesOf(chunk) -> bytes:
        return chunk.data if type(chunk).__name__ == 'RawChunk' else np.ascontiguousarray(chunk).tobytes()

    def _describe(self, chunk) -> dict:
        """What a receiving rank needs to rebuild Queue items from block bytes: raw reads (RawChunk) with their reader
        options, or the reference reader's complex arrays."""
        if type(chunk).__name__ == 'RawChunk':
            return dict(raw=True, code=chunk.code, bigEndian=chunk.bigEndian, normalize=chunk.normalize,
                        correctIq=chunk.correctIq, impedance=chunk.impedance, chunk_bytes=len(chunk.data), samples=chunk.size)
        return dict(raw=False, code=self._encodingOf(chunk), chunk_bytes=chunk.nbytes, samples=chunk.size)

    def _startHelpers(self, isDead) -> None:
        """Rank 0 of ``gpus`` ranks: one more process per further GPU, started with 'spawn' before this process touches
        its GPU, each running this very processor on its share of the rows."""
        import multiprocessing as mp
        from ..parallel import RcclTransport
        uid = RcclTransport.unique_id() if self._transportKind == 'rccl' else None
        ctx = mp.get_context('spawn')
        self._helpers = [ctx.Process(target=_rank_main, args=(self, r, self._gpus, uid, isDead), daemon=True)
                         for r in range(1, self._gpus)]
        for p in self._helpers:
            p.start()
        self._enterRank(0, self._gpus, uid)

    def processData(self, isDead, buffer, *args, **kwargs) -> None:
        self._halt = isDead
        try:
            if self._world == 1 and self._gpus > 1:
                self._startHelpers(isDead)
            sh = self._shard()
            self._nFreq = sh.rows
            self._fanout = RowFanOut(self.host, sh.port(self.port), sh.rows)
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
            self._releaseRanks()
            if self._fanout is not None:
                self._fanout.close()
            if self._channel is not None:
                self._channel.t.close()
            for p in self._helpers:
                p.join(timeout=5)
            vprint('Multi-VFO writer halted')
            return

    def __getstate__(self):
        state = super().__getstate__()
        state['_fanout'] = None
        state['_channel'] = None
        state['_outDev'] = None
        state['_helpers'] = []
        state.pop('_halt', None)
        return state
