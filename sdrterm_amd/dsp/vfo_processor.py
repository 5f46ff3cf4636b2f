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
Rank r computes rows [lo_r, hi_r) only.  Every row is served on the ONE port the reference serves it on
(vfo_processor.py:99-110): rank 0 listens, the k-th connection is row k of the reference's order, and a connection whose row
belongs to another rank is handed to that process (its file descriptor travels over a Unix socket, ``fanout.py``), which then
writes to the client directly -- ``example_simo.sh``-style clients work unchanged.  The reference's handshake holds for the job
as a whole (no processing before every row of every rank has its client: the block exchange is collective, so one waiting rank
holds all of them).  Block k+1 is uploaded and broadcast beside the
kernels of block k (two device buffers per rank, a communication stream, events: ``parallel.RcclTransport``), as the
single-GPU loop overlaps its copy-in.  Every wait on a collective is bounded and watches the halt flag and the helper
processes: a rank that dies takes the job down (halt flag up, non-zero exit) instead of leaving the others in a collective.
"""
import os

import numpy as np

from .dsp_processor import DspProcessor, _EndOfStream
from .fanout import make_fanout
from ..misc.general_util import eprint, findPort, vprint


def _parse_host(spec: str):
    if ':' in spec:
        host, port = spec.split(':')
        return host, int(port)
    return spec, findPort(spec)


def _rank_main(processor, rank, world, uid, isDead):
    """Entry point of the helper processes (ranks 1 .. world-1): the same processor, pickled, on another GPU.  A helper that
    cannot go on raises the shared halt flag: rank 0 and the reader watch it."""
    try:
        processor._parentPid = os.getppid()          # rank 0: while it lives, this rank waits for its blocks
        processor._enterRank(rank, world, uid)
        processor.processData(isDead, None)
    except BaseException:
        isDead.value = 1
        raise


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
        self._deviceBase = self._device
        self._transportKind = kwargs.get('transport') or 'rccl'       # or a picklable factory of a transport object (the CPU tests)
        # build extension (the reference's demodulation is processor-wide, io_args.py:118): one of 'fm' / 'am' per row, in the
        # row order above (csv entries, then the center), as a list or a csv string -- FM and AM channels from one pass
        modes = kwargs.get('rowModes')
        if isinstance(modes, str):
            modes = [m.strip().lower() for m in modes.split(',')]
        if modes is not None and (len(modes) != self._nFreq or any(m not in ('fm', 'am') for m in modes)):
            raise ValueError(f"rowModes needs one of 'fm' / 'am' for each of the {self._nFreq} rows (csv entries, then the center)")
        self.rowModes = list(modes) if modes is not None else None
        self._channel = None
        self._helpers = []
        self._stream, self._slots, self._moreComing, self._announced = None, [None, None], False, False
        # diagnostics / tests: the block channel with a single rank too (a 1-GPU box then runs the whole multi-GPU data path:
        # header, upload + broadcast into two buffers, kernels behind events, bounded waits)
        self._forceChannel = bool(kwargs.get('forceChannel') or os.environ.get('SDRTERM_SIMO_CHANNEL') == '1')

    # ---- this rank's share -------------------------------------------------------------------------------------
    def _enterRank(self, rank: int, world: int, uid=None) -> None:
        self._rank, self._world, self._uid = rank, world, uid
        if self._transportKind == 'rccl':
            from ..parallel import device_for
            self._device = device_for(rank, self._deviceBase)    # one process per GPU of the node, from ``device=`` upwards
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

    def _rowModes(self):
        if self.rowModes is None:
            return None
        sh = self._shard()
        return self.rowModes[sh.lo:sh.hi]

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
        if callable(self._transportKind):
            t = self._transportKind()        # tests: a transport on a gloo group (tests/torch_transport.py)
        else:
            t = parallel.RcclTransport(self._rank, self._world, self._device, self._uid, alive=self._jobAlive)
        self._channel = parallel.BlockChannel(t, chunk_bytes)

    def _jobAlive(self) -> bool:
        """False once the job is going down.  Rank 0: the halt flag is up or a helper process has exited.  The other ranks do
        NOT watch the halt flag: when it goes up (Ctrl-C, the reader's end) rank 0 still sends them an end-of-stream block, and
        they must be in the collective to take it -- a helper that left on the flag alone would leave rank 0 with a broadcast no
        peer completes.  They watch rank 0 itself (their parent process) and the transport's timeout."""
        if self._isDead:
            return False
        if self._rank > 0:
            parent = getattr(self, '_parentPid', None)
            return parent is None or os.getppid() == parent
        if bool(self._halt.value):
            return False
        return all(p.is_alive() for p in self._helpers)

    def _halted(self, isDead) -> bool:
        # ranks 1 .. N-1 leave on the end-of-stream block rank 0 sends them when IT leaves (halt flag included), not on the flag
        return bool(self._isDead) if (self._channelled() and self._rank > 0) else super()._halted(isDead)

    def _takeBatch(self, buffer):
        """Rank 0 takes the batch from the reader's Queue as the single-GPU processor does and sends it on; the other
        ranks receive it.  Blocks travel as raw bytes + what the engine needs to know about them."""
        if not self._channelled():
            return super()._takeBatch(buffer)
        if self._rank == 0:
            chunks, eof = super()._takeBatch(buffer)
            if self._isDead or self._halt.value:
                raise KeyboardInterrupt                      # halted while waiting for the Queue: leave as the loop would
            if not self._jobAlive():
                raise RuntimeError('a helper rank has exited')
            self._announce(chunks[0] if chunks else None)
            payload = b''.join(self._bytesOf(c) for c in chunks)
            # "more input is queued behind this block": every rank then keeps this block's kernels in flight and takes the
            # next one, exactly as this rank does (DspProcessor._processData)
            blk = self._channel.exchange(payload, len(chunks), eof, not eof and super()._moreQueued(buffer))
            self._eofSent = eof
        else:
            if self._channel is None:
                self._openChannel(1)
                import ast
                t = self._channel.t                          # (the first block may be long in coming: wait while the job lives)
                first = t.bcast_bytes(None, 512, timeout=0) if hasattr(t, 'wait_event') else t.bcast_bytes(None, 512)
                self._blockKind = ast.literal_eval(first.decode().strip())
                if self._blockKind:
                    self._channel.chunk_bytes = self._blockKind['chunk_bytes']
            blk = self._channel.exchange(None)
        self._moreComing = blk.more
        if blk.n:
            blk.size = self._blockKind['samples']
        return ([blk] if blk.n else []), blk.eof

    def _channelled(self) -> bool:
        return self._world > 1 or self._forceChannel

    def _moreQueued(self, buffer) -> bool:
        return self._moreComing if self._channelled() else super()._moreQueued(buffer)

    def _submitBatch(self, chunks):
        if not self._channelled():
            return super()._submitBatch(chunks)
        return None, self._startBlock(chunks[0]), 0

    def _collectBatch(self, pending) -> np.ndarray:
        return super()._collectBatch(pending) if pending[0] is not None else self._finishBlock(pending[1])

    def _blockEngine(self):
        k = self._blockKind
        reader = dict(normalize=k['normalize'], correct_iq=k['correctIq'], impedance=k['impedance'],
                      big_endian=k['bigEndian']) if k['raw'] else {}
        return self._getEngine(k['samples'], k['code'], **reader)

    def _startBlock(self, blk):
        """Kernels and copy-out of one exchanged block, asynchronous on this rank's stream: they wait (on the device) for the
        block's broadcast, and the next broadcast into the same buffer waits for them."""
        eng = self._blockEngine()
        if not blk.dev_ptr:
            return eng.process(blk.host, blk.n).reshape(self._nFreq, blk.n, eng.m)      # host transport (tests): synchronous
        from .. import _native as nat
        import ctypes as C
        lib, t = nat.lib(), self._channel.t
        if self._stream is None:
            p = C.c_void_p()
            nat.check(lib.sdrt_stream_create(C.byref(p)))
            self._stream = p.value
        sl = self._slots[blk.slot]
        need = self._nFreq * blk.n * eng.m * 8
        if sl is None or sl['cap'] < need:
            cap = max(need, self._nFreq * self.MAX_BATCH * eng.m * 8)
            ev = C.c_void_p()
            nat.check(lib.sdrt_event_create(C.byref(ev)))
            sl = self._slots[blk.slot] = dict(cap=cap, dev=nat.DeviceBuffer(cap), host=nat.pinned_array(cap), done=ev.value)
        t.block_ready(blk.slot, self._stream)
        eng.process_device(blk.dev_ptr, blk.n, sl['dev'].ptr, self._stream)
        t.block_consumed(blk.slot, self._stream)
        nat.check(lib.sdrt_memcpy_d2h_async(sl['host'].ctypes.data, sl['dev'].ptr, need, self._stream))
        nat.check(lib.sdrt_event_record(sl['done'], self._stream))
        return blk.slot, blk.n, eng.m

    def _finishBlock(self, started) -> np.ndarray:
        if isinstance(started, np.ndarray):
            return started
        slot, n, m = started
        sl = self._slots[slot]
        self._channel.t.wait_event(sl['done'], 'kernels of a block')
        return sl['host'][:self._nFreq * n * m * 8].view(np.float64).reshape(self._nFreq, n, m).copy()

    def _announce(self, first_chunk) -> None:
        """Rank 0, once: tell the others what the blocks hold (None: nothing will come).  The channel itself is opened when the
        ranks start (processData): joining the communicator is collective and bounded in time, the first chunk of a live source
        may be long in coming -- the others wait for THIS message for as long as the job lives."""
        if self._announced:
            return
        self._announced = True
        self._blockKind = self._describe(first_chunk) if first_chunk is not None else None
        if self._channel is None:
            self._openChannel(1)
        if self._blockKind:
            self._channel.chunk_bytes = self._blockKind['chunk_bytes']
        self._channel.t.bcast_bytes(repr(self._blockKind).encode().ljust(512), 512)

    def _releaseRanks(self) -> None:
        """Rank 0 is leaving (end of stream already sent, halt flag, error): the other ranks wait in a collective and need
        an end-of-stream block to leave as well.  Only while they are all there to take it; the wait is bounded either way."""
        if self._channelled() and self._rank == 0 and not getattr(self, '_eofSent', False):
            self._eofSent = True
            if not all(p.is_alive() for p in self._helpers):
                self._abandon = True             # a collective may still sit in the stream with nobody to complete it
                return
            try:
                if self._channel is not None and hasattr(self._channel.t, 'alive'):
                    self._channel.t.alive = lambda: all(p.is_alive() for p in self._helpers)     # (the halt flag may be up already)
                self._announce(None)
                self._channel.exchange(b'', 0, True)
            except Exception:
                self._abandon = True             # the end-of-stream block did not get through: nothing behind it may be waited for

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
        import socket
        from ..parallel import RcclTransport
        # the one port of the job must be free before anybody starts (rank 0 failing to bind after the helpers are up would leave
        # them in their first collective)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as probe:
            probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                probe.bind((self.host, self.port))
            except OSError as e:
                raise OSError(f'{self.host}:{self.port} (--simo on {self._gpus} GPUs) is not free: {e}') from e
        uid = RcclTransport.unique_id() if self._transportKind == 'rccl' else None
        ctx = mp.get_context('spawn')
        self._helpers = [ctx.Process(target=_rank_main, args=(self, r, self._gpus, uid, isDead), daemon=True)
                         for r in range(1, self._gpus)]
        for p in self._helpers:
            p.start()
        self._enterRank(0, self._gpus, uid)

    def processData(self, isDead, buffer, *args, **kwargs) -> None:
        self._halt = isDead
        failed = False
        try:
            if self._world == 1 and self._gpus > 1:
                self._startHelpers(isDead)
            sh = self._shard()
            self._nFreq = sh.rows
            self._fanout = make_fanout(self.host, self.port, self._rank, sh.bounds)
            self._fanout.start()
            if self._world > 1 and self._rank == 0 and self._transportKind == 'rccl':
                self._openChannel(1)               # every rank joins the communicator now (the helpers are in theirs already)
            self._processData(isDead, buffer)
        except (KeyboardInterrupt, _EndOfStream):
            pass
        except Exception:
            import sys
            import traceback
            failed = True
            eprint(f'{self} stopped on an error:')
            traceback.print_exc(file=sys.stderr)
        finally:
            self._isDead = True
            if failed and self._world > 1:
                isDead.value = 1                 # one rank down takes the job down: the others leave their waits, the reader stops
            self._releaseRanks()
            if self._fanout is not None:
                self._fanout.close()
            if self._channel is not None:
                try:
                    abandon = failed or getattr(self, '_abandon', False)
                    self._channel.t.close(abandon=True) if abandon and hasattr(self._channel.t, 'wait_event') else self._channel.t.close()
                except Exception:
                    pass
            for p in self._helpers:
                p.join(timeout=5)
                if p.is_alive():
                    p.terminate()
            vprint('Multi-VFO writer halted')
            if failed and self._world > 1 and self._rank > 0:
                raise SystemExit(1)              # a helper: non-zero exit code for rank 0's liveness check
            return

    def __getstate__(self):
        state = super().__getstate__()
        state['_fanout'] = None
        state['_channel'] = None
        state['_stream'], state['_slots'] = None, [None, None]
        state['_helpers'] = []
        state.pop('_halt', None)
        state.pop('_abandon', None)
        return state
