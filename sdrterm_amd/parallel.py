"""Multi-GPU operation of the hot path: one process per GPU, no reduction anywhere.

* single VFO: chunks are independent (dsp_processor.py:140-149 keeps no state between them), so a stream of chunks is
  split into contiguous ranges, one per rank, with no collective at all (``shard_chunks``; bench.py's default workload);
* ``--simo``: rows (VFO channels) are independent given the raw block (vfo_processor.py:80-84), so the rows are split
  contiguously across the ranks (``shard_rows``) and the rank that owns the reader's Queue hands every raw block to the
  others -- the only exchange on the path.  ``BlockChannel`` is that exchange: a fixed-size header (chunks in the block,
  end-of-stream flag) followed by the block's bytes, over one of two transports:

    ``RcclTransport``   device buffers, RCCL broadcast over xGMI through the C ABI (``sdrt_comm_*``, include/sdrterm_hip.h);
                        the 128-byte unique id travels over a TCP socket at MASTER_ADDR:MASTER_PORT
  or any object with the same five methods (``bcast_bytes``, ``bcast_block``, ``block_consumed``, ``allgather_bytes``, ``close``
  and ``rank`` / ``world`` / ``on_device``): the CPU tests pass one built on a gloo group (tests/torch_transport.py).  Nothing in
  this package imports torch.

``SimoShard`` is what ``VfoProcessor`` uses: which rows this rank serves, which TCP port, and the channel.
``chain_iq_offsets`` is the one exchange a chunk-sharded stream needs when ``--correct-iq`` is on (the carried offset).
"""
import ctypes as C
import os
import socket
import struct
import time
from typing import List, Optional, Tuple

import numpy as np

HEADER = struct.Struct('<qqq')       # chunks in the block, flags, sequence number
F_EOF, F_MORE = 1, 2                 # flags: end of stream; more input was queued behind this block when it was sent
SLOTS = 2                            # block buffers per rank: block k+1 arrives beside the kernels of block k


class PeerLost(RuntimeError):
    """A collective did not complete in time, or the halt flag went up while waiting for it."""


def split_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) of ``total`` items for ``rank`` of ``world`` (first ranks get the extras)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('bad world/rank')
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def device_for(local_rank: int, base: int = 0) -> int:
    """The device a rank of this node runs on: ``base + local_rank`` (one process per GPU), unless SDRTERM_DEVICE_MAP lists the
    devices rank by rank ("4,5,6,7": GPUs 4-7; "0,0": two ranks on one GPU -- only with a transport that allows it, which RCCL
    does not: tests/standin_rccl)."""
    spec = os.environ.get('SDRTERM_DEVICE_MAP')
    if spec:
        devs = [int(v) for v in spec.split(',')]
        if local_rank >= len(devs):
            raise ValueError(f'SDRTERM_DEVICE_MAP={spec!r} names {len(devs)} device(s), rank {local_rank} has none')
        return devs[local_rank]
    return base + local_rank


def shard_rows(row_freqs: List[float], world: int, rank: int) -> Tuple[List[float], Tuple[int, int]]:
    """Rows (VFO channels) this rank computes, and their [start, stop) in the reference's row order."""
    lo, hi = split_range(len(row_freqs), world, rank)
    return list(row_freqs[lo:hi]), (lo, hi)


def shard_chunks(n_chunks: int, world: int, rank: int) -> Tuple[int, int]:
    return split_range(n_chunks, world, rank)


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world) as torch.distributed.run and ``spawn_ranks`` export them; (0, 0, 1) when unset."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0'))),
            int(os.environ.get('WORLD_SIZE', '1')))


# ---- transports ------------------------------------------------------------------------------------------------------
class RcclTransport:
    """RCCL broadcast between device buffers through the C ABI; no torch.  ``uid``: the 128-byte id (rank 0 creates it with
    ``RcclTransport.unique_id()`` and the launcher passes it on), or None to exchange it over TCP at MASTER_ADDR:MASTER_PORT."""

    on_device = True

    def __init__(self, rank: int, world: int, device: int, uid: Optional[bytes] = None, alive=None):
        from . import _native as nat
        self._nat = nat
        self.rank, self.world, self.device_index = rank, world, device
        if uid is None:
            uid = self._exchange_uid(rank, world)
        self._comm = C.c_void_p()
        buf = C.create_string_buffer(uid, 128)
        # joining the communicator is collective and cannot be interrupted: it runs beside a watch on the job, so that a peer
        # that never arrives fails this rank instead of holding it for ever
        import threading
        box = {}

        def join_comm():
            try:
                box['rc'] = nat.lib().sdrt_comm_init(buf, world, rank, device, C.byref(self._comm))
                box['err'] = nat.lib().sdrt_last_error().decode(errors='replace') if box['rc'] else ''
            except BaseException as e:                       # noqa: BLE001 -- reported by the waiting thread
                box['exc'] = e
        th = threading.Thread(target=join_comm, daemon=True)
        th.start()
        deadline = time.monotonic() + float(os.environ.get('SDRTERM_COMM_INIT_TIMEOUT', '600'))
        while th.is_alive():
            th.join(0.05)
            if th.is_alive() and ((alive is not None and not alive()) or time.monotonic() > deadline):
                raise PeerLost('communicator: not every rank arrived')
        if 'exc' in box:
            raise box['exc']
        if box.get('rc'):
            raise nat.SdrtError(box['rc'], box['err'])
        nat.check(nat.lib().sdrt_set_device(device))         # (the communicator was joined on another thread: bind this one too)
        self.timeout = float(os.environ.get('SDRTERM_COMM_TIMEOUT', '300'))      # seconds a collective may take
        self.alive = alive          # callable: False once the job is going down (halt flag, a dead helper process)
        p = C.c_void_p()
        nat.check(nat.lib().sdrt_stream_create(C.byref(p)))
        self._s_comm = p.value      # every collective of this transport runs on this stream
        self._ev_sync = self._event()
        self._stage = None          # device + page-locked staging buffers of bcast_bytes
        self._stage_host = None
        self._stage_cap = 0
        self._block = [None] * SLOTS          # device buffers of bcast_block
        self._block_host = [None] * SLOTS     # the root's page-locked copies
        self._block_cap = [0] * SLOTS
        self._ev_ready = [None] * SLOTS       # broadcast into the buffer done
        self._ev_free = [None] * SLOTS        # kernels reading the buffer done

    @staticmethod
    def unique_id() -> bytes:
        from . import _native as nat
        buf = C.create_string_buffer(128)
        nat.check(nat.lib().sdrt_comm_unique_id(buf))
        return buf.raw

    UID_MAGIC = b'SDRTuid1'
    UID_PORTS = 16            # rank 0 listens on the first free one of this many ports from the base on

    @staticmethod
    def _exchange_uid(rank: int, world: int, job: Optional[str] = None) -> bytes:
        """The 128-byte id from rank 0 to the others over TCP at MASTER_ADDR, ports MASTER_PORT + 17 .. + 32 (or from
        SDRTERM_UID_PORT on): rank 0 takes the first free one, the others knock until one of them answers with the magic word
        and this world size (a launcher's own store or a stranger's service on one of the ports is passed over)."""
        if world == 1:
            return RcclTransport.unique_id()
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        base = int(os.environ.get('SDRTERM_UID_PORT', int(os.environ.get('MASTER_PORT', '29500')) + 17))
        # the hello names the job as well: two jobs on one host whose port ranges overlap (and whose world sizes agree) must not
        # take each other's id -- SDRTERM_JOB_ID if the launcher gives one, else MASTER_ADDR:MASTER_PORT, which differ per job
        import hashlib
        job = job or os.environ.get('SDRTERM_JOB_ID') or f"{addr}:{os.environ.get('MASTER_PORT', '29500')}"
        token = hashlib.sha256(job.encode()).digest()[:8]
        hello = RcclTransport.UID_MAGIC + struct.pack('<q', world) + token
        if rank == 0:
            uid = RcclTransport.unique_id()
            srv = None
            for port in range(base, base + RcclTransport.UID_PORTS):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((addr, port))
                except OSError:
                    s.close()
                    continue
                srv = s
                break
            if srv is None:
                raise OSError(f'no free port for the communicator id in {addr}:{base}..{base + RcclTransport.UID_PORTS - 1}')
            with srv:
                srv.listen(world)
                srv.settimeout(float(os.environ.get('SDRTERM_COMM_INIT_TIMEOUT', '600')))
                served = set()                               # distinct ranks: a rank that asks twice (a retry) counts once
                while len(served) < world - 1:
                    conn, _peer = srv.accept()
                    with conn:
                        conn.settimeout(10)
                        try:
                            got = b''
                            while len(got) < len(hello) + 4:
                                part = conn.recv(len(hello) + 4 - len(got))
                                if not part:
                                    break
                                got += part
                            if len(got) != len(hello) + 4 or got[:len(hello)] != hello:
                                continue                     # not one of ours
                            peer_rank = struct.unpack('<i', got[len(hello):])[0]
                            if not 0 < peer_rank < world:
                                continue
                            conn.sendall(hello + uid)
                            served.add(peer_rank)
                        except OSError:
                            continue
            return uid
        deadline = time.time() + float(os.environ.get('SDRTERM_COMM_INIT_TIMEOUT', '600'))
        while True:
            for port in range(base, base + RcclTransport.UID_PORTS):
                try:
                    with socket.create_connection((addr, port), timeout=2) as s:
                        s.settimeout(5)
                        s.sendall(hello + struct.pack('<i', rank))
                        got = b''
                        while len(got) < len(hello) + 128:
                            part = s.recv(len(hello) + 128 - len(got))
                            if not part:
                                break
                            got += part
                        if len(got) == len(hello) + 128 and got[:len(hello)] == hello:
                            return got[len(hello):]
                except OSError:
                    pass
            if time.time() > deadline:
                raise PeerLost('communicator id: rank 0 did not answer')
            time.sleep(0.1)

    def bcast_device(self, dev_ptr: int, nbytes: int, root: int = 0, stream: int = 0, scatter: Optional[bool] = None) -> None:
        """In place on device memory, asynchronous on ``stream``.  ``scatter`` (the default whenever the block divides by the number
        of ranks; SDRTERM_BCAST=ring in the environment or scatter=False selects ncclBroadcast): the root sends every rank its
        1/world of the block and the ranks all-gather the shards -- every xGMI link carries one shard instead of every hop of a ring
        the whole block (sdrt_comm_bcast_scatter).  xGMI is point to point: a whole-block broadcast is bound by one link per hop
        (DESIGN section 5), which is why the form that uses all seven links of the root at once is the default."""
        if scatter is None:
            scatter = self.world > 1 and nbytes % self.world == 0 and os.environ.get('SDRTERM_BCAST', 'scatter') != 'ring'
        fn = self._nat.lib().sdrt_comm_bcast_scatter if scatter else self._nat.lib().sdrt_comm_bcast
        self._nat.check(fn(self._comm, dev_ptr, nbytes, root, stream or None))

    def allgather_device(self, send_ptr: int, recv_ptr: int, bytes_per_rank: int, stream: int = 0) -> None:
        self._nat.check(self._nat.lib().sdrt_comm_allgather(self._comm, send_ptr, recv_ptr, bytes_per_rank, stream or None))

    # ---- waiting with a deadline ---------------------------------------------------------------------------------------------
    def _event(self):
        p = C.c_void_p()
        self._nat.check(self._nat.lib().sdrt_event_create(C.byref(p)))
        return p.value

    def wait_event(self, event: int, what: str, timeout: Optional[float] = None) -> None:
        """Host wait for ``event`` by polling: returns when it has completed, raises ``PeerLost`` after ``timeout`` seconds
        (None: this transport's) or as soon as ``self.alive()`` says the job is going down (halt flag, a dead helper)."""
        nat, done = self._nat, C.c_int(0)
        limit = timeout if timeout is not None else self.timeout
        deadline = None if limit is None else time.monotonic() + limit
        pause = 0.0
        while True:
            nat.check(nat.lib().sdrt_event_query(event, C.byref(done)))
            if done.value:
                return
            if self.alive is not None and not self.alive():
                raise PeerLost(f'{what}: the job is going down')
            if deadline is not None and time.monotonic() > deadline:
                raise PeerLost(f'{what}: no completion within {limit:g} s (a peer rank is gone?)')
            time.sleep(pause)
            pause = min(0.002, pause + 0.0001)              # spin first: a block broadcast takes tens of microseconds

    def _finish(self, what: str, timeout: Optional[float] = None) -> None:
        """Everything submitted to the communication stream so far."""
        self._nat.check(self._nat.lib().sdrt_event_record(self._ev_sync, self._s_comm))
        self.wait_event(self._ev_sync, what, timeout)

    def bcast_bytes(self, data: Optional[bytes], nbytes: int, root: int = 0, timeout: Optional[float] = None) -> bytes:
        """Small host messages (the block header) through a page-locked staging buffer; returns when the message is here.
        ``timeout``: None = this transport's; 0 = wait for as long as the job is alive (a rank waiting for the next block of
        a live stream)."""
        nat = self._nat
        if nbytes > self._stage_cap:
            if self._stage is not None:
                self._stage.free()
            self._stage, self._stage_cap = nat.DeviceBuffer(max(nbytes, 4096)), max(nbytes, 4096)
            self._stage_host = nat.pinned_array(self._stage_cap)
        if self.rank == root:
            self._stage_host[:nbytes] = np.frombuffer(data, dtype=np.uint8)
            nat.check(nat.lib().sdrt_memcpy_h2d_async(self._stage.ptr, self._stage_host.ctypes.data, nbytes, self._s_comm))
        self.bcast_device(self._stage.ptr, nbytes, root, self._s_comm, scatter=False)      # (a header: one small message)
        if self.rank != root:
            nat.check(nat.lib().sdrt_memcpy_d2h_async(self._stage_host.ctypes.data, self._stage.ptr, nbytes, self._s_comm))
        self._finish('header broadcast', None if timeout is None else (timeout or 1e18))
        return self._stage_host[:nbytes].tobytes()

    def allgather_bytes(self, data: bytes) -> List[bytes]:
        """Every rank's ``data`` (same length everywhere), in rank order; through device staging buffers."""
        nat, nb = self._nat, len(data)
        send, recv = nat.DeviceBuffer(max(nb, 16)), nat.DeviceBuffer(max(nb * self.world, 16))
        try:
            send.upload(np.frombuffer(data, dtype=np.uint8))
            self.allgather_device(send.ptr, recv.ptr, nb, self._s_comm)
            self._finish('all-gather')
            allb = recv.download(np.empty(nb * self.world, dtype=np.uint8)).tobytes()
        finally:
            send.free()
            recv.free()
        return [allb[r * nb:(r + 1) * nb] for r in range(self.world)]

    def selftest(self, nbytes: int = 8 << 20) -> dict:
        """Pre-flight of the block feeds, before anything depends on them: a block of ``nbytes`` patterned bytes goes from rank 0
        to every rank once with ncclBroadcast and once as 1/N shards + all-gather (sdrt_comm_bcast_scatter), and once as an
        all-gather of every rank's own shard; every rank checks what arrived against the pattern (which it can compute itself) and
        the verdicts are gathered.  Returns {'world', 'ranks_seen', 'bcast_ok', 'scatter_ok', 'allgather_ok', 'selftest_ok',
        'bytes'}: ``selftest_ok`` only when every rank saw every feed deliver the right bytes.  Collective: every rank calls it.
        (The scatter form has only ever met RCCL on real hardware with a world of one: DESIGN section 5.)"""
        nat = self._nat
        unit = 16 * self.world                                   # divides by the ranks, shards 16-byte aligned
        nbytes = (max(int(nbytes), 1 << 20) + unit - 1) // unit * unit       # (the scatter form starts at 1 MiB: comm.hip)
        pattern = ((np.arange(nbytes, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) & np.uint64(0xff)).astype(np.uint8)
        buf, out = nat.DeviceBuffer(nbytes), nat.DeviceBuffer(nbytes)
        got = np.empty(nbytes, dtype=np.uint8)
        res = {}
        try:
            for name, scatter in (('bcast_ok', False), ('scatter_ok', True)):
                buf.upload(pattern if self.rank == 0 else np.zeros(nbytes, dtype=np.uint8))
                self.bcast_device(buf.ptr, nbytes, 0, self._s_comm, scatter=scatter)
                self._finish(name)
                res[name] = bool(np.array_equal(buf.download(got), pattern))
            shard = nbytes // self.world
            mine = np.zeros(nbytes, dtype=np.uint8)
            mine[:shard] = pattern[self.rank * shard:(self.rank + 1) * shard]
            buf.upload(mine)
            self.allgather_device(buf.ptr, out.ptr, shard, self._s_comm)
            self._finish('allgather_ok')
            res['allgather_ok'] = bool(np.array_equal(out.download(got), pattern))
        finally:
            buf.free()
            out.free()
        mine_ok = bytes([int(res['bcast_ok']), int(res['scatter_ok']), int(res['allgather_ok'])])
        votes = self.allgather_bytes(mine_ok + bytes(13))
        for k, name in enumerate(('bcast_ok', 'scatter_ok', 'allgather_ok')):
            res[name] = all(v[k] == 1 for v in votes)
        res.update(world=self.world, ranks_seen=len(votes), bytes=int(nbytes),
                   selftest_ok=bool(res['bcast_ok'] and res['scatter_ok'] and res['allgather_ok'] and len(votes) == self.world))
        return res

    def bcast_block(self, data: Optional[bytes], nbytes: int, root: int = 0, slot: int = 0):
        """('dev', pointer): the block stays in device memory on every rank (one upload on the root, RCCL broadcast,
        sdrt_process_device on the spot).  Asynchronous: upload and broadcast are queued on the communication stream behind
        the kernels that last read buffer ``slot`` (``block_consumed``); the caller's stream must wait for them with
        ``block_ready`` before it reads the block."""
        nat, lib = self._nat, self._nat.lib()
        if nbytes > self._block_cap[slot]:
            if self._block[slot] is not None:
                self._finish('block buffer growth')           # bounded, and watching the job: never an open-ended stream wait
                self._block[slot].free()
            self._block[slot], self._block_cap[slot] = nat.DeviceBuffer(nbytes), nbytes
            if self.rank == root:
                self._block_host[slot] = nat.pinned_array(nbytes)
        if self._ev_free[slot] is not None:
            nat.check(lib.sdrt_stream_wait_event(self._s_comm, self._ev_free[slot]))
        if self.rank == root:
            self._block_host[slot][:nbytes] = np.frombuffer(data, dtype=np.uint8)
            nat.check(lib.sdrt_memcpy_h2d_async(self._block[slot].ptr, self._block_host[slot].ctypes.data, nbytes, self._s_comm))
        self.bcast_device(self._block[slot].ptr, nbytes, root, self._s_comm)
        if self._ev_ready[slot] is None:
            self._ev_ready[slot] = self._event()
        nat.check(lib.sdrt_event_record(self._ev_ready[slot], self._s_comm))
        return 'dev', self._block[slot].ptr

    def block_ready(self, slot: int, stream: int) -> None:
        """``stream`` waits (on the device) for the block in buffer ``slot``."""
        self._nat.check(self._nat.lib().sdrt_stream_wait_event(stream or None, self._ev_ready[slot]))

    def block_consumed(self, slot: int, stream: int = 0) -> None:
        """Everything submitted to ``stream`` so far reads buffer ``slot``: the next broadcast into it waits for that."""
        if self._ev_free[slot] is None:
            self._ev_free[slot] = self._event()
        self._nat.check(self._nat.lib().sdrt_event_record(self._ev_free[slot], stream or None))

    def close(self, abandon: bool = False):
        """``abandon``: a peer is gone and collectives may still sit in the communication stream -- nothing is waited for and
        the communicator is left to the process exit."""
        lib = self._nat.lib()
        if not abandon:
            for b in self._block:
                if b is not None:
                    b.free()
            if getattr(self, '_comm', None) is not None and self._comm.value:
                lib.sdrt_comm_destroy(self._comm)
            if self._stage is not None:
                self._stage.free()
            for e in [self._ev_sync] + self._ev_ready + self._ev_free:
                if e is not None:
                    lib.sdrt_event_destroy(e)
            if self._s_comm:
                lib.sdrt_stream_destroy(self._s_comm)
        self._block, self._stage, self._s_comm = [None] * SLOTS, None, None
        self._ev_sync, self._ev_ready, self._ev_free = None, [None] * SLOTS, [None] * SLOTS
        self._comm = C.c_void_p()


class Block:
    """One exchanged block: ``n`` chunks either in host memory (``host``: bytes) or in device memory (``dev_ptr``, buffer
    ``slot``); ``more``: the sender had more input queued behind it (the receivers then keep this block's kernels in flight
    and take the next one, as the single-GPU loop does with its Queue); ``seq``: position in the stream."""
    __slots__ = ('n', 'eof', 'more', 'seq', 'host', 'dev_ptr', 'slot', 'size')

    def __init__(self, n, eof, more=False, seq=0, host=None, dev_ptr=0, slot=0, size=0):
        self.n, self.eof, self.more, self.seq = n, eof, more, seq
        self.host, self.dev_ptr, self.slot, self.size = host, dev_ptr, slot, size


class BlockChannel:
    """The reader-side rank sends (chunks in the block, flags, sequence number, block bytes); ``exchange`` returns the same
    ``Block`` on every rank.  The header is waited for; the block itself (device transports) is only queued, into buffer
    ``seq % SLOTS``: see ``RcclTransport.bcast_block``."""

    def __init__(self, transport, chunk_bytes: int, root: int = 0):
        self.t, self.chunk_bytes, self.root = transport, chunk_bytes, root
        self.seq = 0

    def exchange(self, block: Optional[bytes], n_chunks: int = 0, eof: bool = False, more: bool = False) -> Block:
        is_root = self.t.rank == self.root
        hdr = HEADER.pack(n_chunks, (F_EOF if eof else 0) | (F_MORE if more else 0), self.seq) if is_root else None
        # the root only gets here with a block in its hands: every peer answers at once or is gone; the others wait for as long
        # as the stream is silent
        if isinstance(self.t, RcclTransport):
            raw = self.t.bcast_bytes(hdr, HEADER.size, self.root, timeout=None if is_root else 0)
        else:
            raw = self.t.bcast_bytes(hdr, HEADER.size, self.root)
        n_chunks, flags, seq = HEADER.unpack(raw)
        if seq != self.seq:
            raise PeerLost(f'block {seq} arrived where block {self.seq} was due')
        out = Block(int(n_chunks), bool(flags & F_EOF), bool(flags & F_MORE), int(seq), slot=int(seq % SLOTS))
        if n_chunks:
            where, what = self.t.bcast_block(block, n_chunks * self.chunk_bytes, self.root, out.slot)
            if where == 'dev':
                out.dev_ptr = what
            else:
                out.host = what
        self.seq += 1
        return out


IQ_PAIR = struct.Struct('<ddd')       # a, Re b, Im b of a chunk range: off_end = a * off_start + b


def fold_iq_pairs(pairs, rank: int, off0: complex = 0j) -> Tuple[complex, complex]:
    """(offset at the start of ``rank``'s range, offset after the last range) from every range's (a, b) in stream order."""
    off, mine = complex(off0), complex(off0)
    for r, (a, b) in enumerate(pairs):
        if r == rank:
            mine = off
        off = a * off + b
    return mine, off


def chain_iq_offsets(transport, a: float, b: complex, off0: complex = 0j) -> Tuple[complex, complex]:
    """Chunk-sharded stream with ``--correct-iq`` ("by chunk"): the carried offset
    (read_file.py:53, iq_correction.py:19-26) is the only thing one chunk hands to the next, and the recursion is affine in it.
    Rank r holds the r-th contiguous range of the stream (``shard_chunks``) and passes what its range does to the offset
    (``Engine.iq_range_device``); one all-gather of 24 bytes per rank later every rank knows the offset its range starts
    from (set ``Engine.iq_offset`` to it, then process) and the offset the whole stream ends with (the next batch's
    ``off0``).  No other collective on that path."""
    got = transport.allgather_bytes(IQ_PAIR.pack(float(a), float(b.real), float(b.imag)))
    pairs = [(v[0], complex(v[1], v[2])) for v in (IQ_PAIR.unpack(g) for g in got)]
    return fold_iq_pairs(pairs, transport.rank, off0)


class SimoShard:
    """This rank's share of a --simo processor: rows [lo, hi) of the reference's row order (vfo_processor.py:43-47).  Every
    rank's rows are served on the ONE port the reference serves them on (vfo_processor.py:106): rank 0 listens, the k-th
    connection is row k, and a connection whose row belongs to another rank is handed to that process (dsp/fanout.py)."""

    def __init__(self, n_rows: int, rank: int, world: int):
        self.rank, self.world = rank, world
        self.lo, self.hi = split_range(n_rows, world, rank)
        self.bounds = [split_range(n_rows, world, r) for r in range(world)]

    @property
    def rows(self) -> int:
        return self.hi - self.lo

    def port(self, base_port: int) -> int:
        return base_port
