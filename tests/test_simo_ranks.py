"""CPU test of the multi-GPU --simo driver: two ranks under gloo run the sharded VfoProcessor end to
end with a stub in place of the GPU engine.  Checked: which rows each rank serves (vfo_processor.py:43-47 order, split
contiguously), that ONE port serves them all as in the reference (vfo_processor.py:99-110) -- the k-th client to connect
receives row k of every chunk as big-endian doubles (vfo_processor.py:83-84) whichever rank computes it, the socket having
been handed to that rank's process --, that every rank sees every block in order and that end of stream reaches all of them;
a halt flag raised in mid-stream; the hand-off of client sockets between processes on its own."""
import os
import queue
import socket
import threading
import time
from tests.torch_transport import TorchTransport, max_over_ranks  # noqa: F401

import numpy as np
import pytest

from sdrterm_amd.dsp.vfo_processor import VfoProcessor

FS, Q, N, CHUNKS = 1024000, 64, 2048, 5
VFOS = '100,200,300,400'            # 5 rows: csv entries + center, then center last


class _StubEngine:
    """Stands in for sdrterm_amd.engine.Engine: sample j of chunk c of a row is  row frequency + checksum(c) + j/1024."""

    def __init__(self, freqs, n, q):
        self.freqs, self.n, self.m = list(freqs), n, n // q
        self.code, self.smooth = 'd', 0

    def process(self, raw, n_chunks):
        buf = np.frombuffer(raw, dtype=np.uint8).reshape(n_chunks, -1)
        check = buf.astype(np.int64).sum(axis=1) % 251
        j = np.arange(self.m) / 1024.0
        out = np.empty((len(self.freqs), n_chunks, self.m))
        for r, f in enumerate(self.freqs):
            out[r] = f + check[:, None] + j[None, :]
        return out.reshape(len(self.freqs), n_chunks * self.m)


class StubVfo(VfoProcessor):
    def _getEngine(self, n, code, smoothed=True, **reader):
        if self._engine is None:
            self._engine = _StubEngine(self._rowFreqs(), n, self._decimationFactor)
        return self._engine


class _Flag:
    value = 0


def chunks_for_test():
    rng = np.random.default_rng(7)
    return [(rng.normal(size=N) + 1j * rng.normal(size=N)).astype(np.complex128) for _ in range(CHUNKS)]


def _client(host, port, nbytes, box, idx):
    deadline = time.time() + 60
    while True:
        try:
            s = socket.create_connection((host, port), timeout=30)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    data = b''
    with s:
        while len(data) < nbytes:
            part = s.recv(nbytes - len(data))
            if not part:
                break
            data += part
    box[idx] = data


def _rank_worker(rank, world, master_port, base_port, q, feed='all at once', flag=None):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(master_port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = StubVfo(FS, vfoHost=f'127.0.0.1:{base_port}', vfos=VFOS, center=1000, dec=Q, omegaOut=5000, transport=TorchTransport)
    p.selectOutputFm()
    p._enterRank(rank, world)
    seen = []                                   # (chunks, end of stream, more queued) of every block this rank took
    take = p._takeBatch

    def spy(buffer):
        chunks, eof = take(buffer)
        seen.append((sum(c.n for c in chunks), eof, p._moreComing))
        if flag is not None and rank == 0 and len(seen) == 3:
            flag.value = 1                      # 'halt': the flag goes up in mid-stream, with the other rank waiting for a block
        return chunks, eof
    p._takeBatch = spy
    if feed != 'all at once':
        p.MAX_BATCH = 2
    sh = p._shard()
    m = N // Q
    total = sh.bounds[-1][1]
    box = [None] * total
    ts = []
    starter = None
    if rank == 0:
        # every client connects to the ONE port, one after the other: the k-th connection is row k (the reference's mapping
        # follows connection order too, example_simo.sh:161), whichever rank computes that row
        def connect_all():
            for k in range(total):
                t = threading.Thread(target=_client, args=('127.0.0.1', base_port, CHUNKS * m * 8, box, k), daemon=True)
                t.start()
                ts.append(t)
                time.sleep(0.3)
        starter = threading.Thread(target=connect_all, daemon=True)
        starter.start()
    buf = None
    if rank == 0:
        buf = queue.Queue()
        if feed == 'all at once':
            for c in chunks_for_test():
                buf.put(c)
            buf.put(b'')
        elif feed == 'halt':
            def forever():
                cs = chunks_for_test()
                while not flag.value:
                    buf.put(cs[0])
                    time.sleep(0.3)
            threading.Thread(target=forever, daemon=True).start()
        else:
            # a live source: the first three chunks are there when the clients have connected, the rest trickle in one by
            # one, then the end marker -- several blocks that are not the last one cross the channel, some with more queued
            # behind them (kept in flight while the next is taken), some without
            def trickle():
                cs = chunks_for_test()
                for c in cs[:3]:
                    buf.put(c)
                for c in cs[3:]:
                    time.sleep(0.7)
                    buf.put(c)
                time.sleep(0.7)
                buf.put(b'')
            threading.Thread(target=trickle, daemon=True).start()
    p.processData(flag if flag is not None else _Flag(), buf)
    if starter is not None:
        starter.join(30)
    for t in ts:
        t.join(30)
    q.put((rank, (sh.lo, sh.hi), sh.port(base_port), [np.frombuffer(b, dtype='>f8').astype(np.float64) if b else None for b in box], seen))
    dist.destroy_process_group()


@pytest.mark.parametrize('feed', ['all at once', 'trickle'])
def test_sharded_vfo_processor_under_gloo(feed):
    import torch.multiprocessing as mp
    from sdrterm_amd.misc.general_util import findPort
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world, master, base = 2, findPort('127.0.0.1'), findPort('127.0.0.1')
    procs = [ctx.Process(target=_rank_worker, args=(r, world, master, base, q, feed)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rows_all = [1100.0, 1200.0, 1300.0, 1400.0, 1000.0]        # csv + center, center LAST
    assert [g[1] for g in got] == [(0, 3), (3, 5)] and [g[2] for g in got] == [base, base]      # one port, as in the reference
    m = N // Q
    check = [int(np.frombuffer(c.tobytes(), dtype=np.uint8).astype(np.int64).sum() % 251) for c in chunks_for_test()]
    blocks = [g[4] for g in got]
    assert blocks[0] == blocks[1]                                  # every rank saw the same blocks, flags included
    assert sum(b[0] for b in blocks[0]) == CHUNKS and blocks[0][-1][1] and not any(b[1] for b in blocks[0][:-1])
    if feed == 'trickle':
        # the helper rank runs processData(isDead, None): blocks that are not the last one must get through it (round 2: the
        # helper died on None.empty() after the first of them), in order, with and without more queued behind them
        assert len(blocks[0]) >= 4 and any(b[2] for b in blocks[0]) and any(not b[2] and not b[1] for b in blocks[0])
    streams = got[0][3]                                            # the clients all sat in rank 0's process: client k <-> row k
    assert len(streams) == 5
    for k, s in enumerate(streams):
        assert s is not None and s.size == CHUNKS * m, f'row {k}'
        exp = np.concatenate([rows_all[k] + check[c] + np.arange(m) / 1024.0 for c in range(CHUNKS)])
        assert np.array_equal(s, exp), f'row {k}'


def test_halt_flag_in_mid_stream_releases_the_other_rank():
    """the halt flag goes up while rank 1 waits for its next block.  Rank 1 must not leave on the flag (rank 0 would
    then queue an end-of-stream broadcast no peer completes, and hang in its teardown behind it): it stays for the
    end-of-stream block rank 0 sends on its way out, and both leave with exit code 0."""
    import torch.multiprocessing as mp
    from sdrterm_amd.misc.general_util import findPort
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    flag = ctx.Value('b', 0)
    world, master, base = 2, findPort('127.0.0.1'), findPort('127.0.0.1')
    procs = [ctx.Process(target=_rank_worker, args=(r, world, master, base, q, 'halt', flag)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    blocks = [g[4] for g in got]
    assert flag.value == 1
    assert blocks[1][-1][1] and blocks[1][-1][0] == 0            # the last thing rank 1 took: the end-of-stream block
    assert [b[:2] for b in blocks[1][:-1]] == [b[:2] for b in blocks[0][:len(blocks[1]) - 1]]


def test_client_sockets_are_handed_to_the_rank_that_owns_the_row():
    """dsp/fanout.py on its own: rank 0 listens on the one port; connections 0, 1 stay with it, 2, 3, 4 belong to rank 1 -- a
    separate process, which receives the sockets themselves (SCM_RIGHTS) and writes to the clients directly; a sixth client is
    refused.  The owning process may knock later than its clients connect."""
    import multiprocessing as mp
    from sdrterm_amd.dsp import fanout
    from sdrterm_amd.misc.general_util import findPort
    port = findPort('127.0.0.1')
    bounds = [(0, 2), (2, 5)]
    f0 = fanout.make_fanout('127.0.0.1', port, 0, bounds)
    f0.start()
    socks = []
    for k in range(5):
        socks.append(socket.create_connection(('127.0.0.1', port), timeout=30))
        time.sleep(0.1)
    ctx = mp.get_context('spawn')
    helper = ctx.Process(target=_handed_rank, args=(port, bounds))
    helper.start()                                                   # after its clients: rank 0 holds them until it knocks
    assert f0.wait_for_clients(lambda: False)
    f0.send_rows(np.array([[10.0, 0.5], [11.0, 0.5]]))
    for k, s in enumerate(socks):
        s.settimeout(30)
        data = b''
        while len(data) < 16:
            part = s.recv(16 - len(data))
            assert part, f'client {k}: connection closed'
            data += part
        row, tag = np.frombuffer(data, dtype='>f8')
        assert (row, tag) == ((10.0 + k, 0.5) if k < 2 else (100.0 + k - 2, float(helper.pid))), f'client {k}'
    extra = socket.create_connection(('127.0.0.1', port), timeout=30)
    extra.settimeout(10)
    assert extra.recv(8) == b''                                      # more clients than rows: refused
    helper.join(30)
    assert helper.exitcode == 0
    f0.close()
    for s in socks + [extra]:
        s.close()


def _handed_rank(port, bounds):
    from sdrterm_amd.dsp import fanout
    f1 = fanout.make_fanout('127.0.0.1', port, 1, bounds)
    f1.start()
    assert f1.wait_for_clients(lambda: False)
    f1.send_rows(np.array([[100.0 + k, float(os.getpid())] for k in range(3)]))
    time.sleep(0.5)
    f1.close()


def test_rank_zero_releases_the_others_when_halted_before_any_data():
    """The halt flag set before the first block: rank 0 must still hand the others an end-of-stream block (they wait in a
    collective), here with world == 1 semantics checked on the bookkeeping only."""
    p = StubVfo(FS, vfoHost='127.0.0.1', vfos=VFOS, dec=Q, omegaOut=5000, transport=TorchTransport)
    p._enterRank(0, 1)
    p._releaseRanks()                      # world 1: nothing to do, must not raise
    assert not getattr(p, '_eofSent', False)


class _FakeLib:
    """sdrt_event_query that never completes (a collective whose peer is gone)."""

    def sdrt_event_query(self, event, done):
        done._obj.value = 0
        return 0

    def sdrt_event_record(self, event, stream):
        return 0


class _FakeNat:
    def lib(self):
        return _FakeLib()

    @staticmethod
    def check(rc):
        assert rc == 0


def test_collective_waits_are_bounded_and_watch_the_job():
    """A rank never sits in a device synchronisation behind a collective: it polls an event with a deadline and asks whether
    the job is still alive (halt flag, helper processes), so a dead peer ends the wait (rank 0 hung for ever)."""
    from sdrterm_amd import parallel
    t = object.__new__(parallel.RcclTransport)
    t._nat, t.timeout, t.alive = _FakeNat(), 0.3, None
    t0 = time.monotonic()
    with pytest.raises(parallel.PeerLost, match='no completion'):
        t.wait_event(1, 'header broadcast')
    assert 0.25 < time.monotonic() - t0 < 5
    calls = []
    t.timeout, t.alive = 60, lambda: calls.append(1) or len(calls) < 3
    with pytest.raises(parallel.PeerLost, match='going down'):
        t.wait_event(1, 'header broadcast')
    assert len(calls) == 3


def test_the_port_is_probed_before_helpers_start():
    """The job serves ONE port (rank 0 listens, the others get their sockets from it): a busy one must stop the start-up in rank 0,
    before any helper waits in a collective."""
    import socket
    from sdrterm_amd.misc.general_util import findPort
    base = findPort('127.0.0.1')
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as busy:
        busy.bind(('127.0.0.1', base))
        busy.listen(1)
        p = StubVfo(FS, vfoHost=f'127.0.0.1:{base}', vfos=VFOS, dec=Q, omegaOut=5000, gpus=2, device=3)
        assert p._deviceBase == 3
        with pytest.raises(OSError, match='is not free'):
            p._startHelpers(_Flag())
        assert p._helpers == []


def test_communicator_id_exchange_passes_over_busy_ports(monkeypatch):
    """The 128-byte id travels from rank 0 to the others over TCP near MASTER_PORT; a port in that range that is taken (the
    launcher's own store, somebody else's service) must neither stop rank 0 nor fool the others."""
    import socket
    from sdrterm_amd import parallel
    from sdrterm_amd.misc.general_util import findPort
    base = findPort('127.0.0.1')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('SDRTERM_UID_PORT', str(base))
    monkeypatch.setattr(parallel.RcclTransport, 'unique_id', staticmethod(lambda: bytes(range(128))))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as stranger:
        stranger.bind(('127.0.0.1', base))           # the first port of the range belongs to somebody who answers nothing useful
        stranger.listen(4)
        got = {}

        def run(rank):
            got[rank] = parallel.RcclTransport._exchange_uid(rank, 3)
        ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in (1, 0, 2)]
        for t in ts:
            t.start()
            time.sleep(0.2)
        for t in ts:
            t.join(60)
        assert got == {0: bytes(range(128)), 1: bytes(range(128)), 2: bytes(range(128))}


def test_communicator_id_exchange_keeps_two_jobs_apart(monkeypatch):
    """two jobs on one host with overlapping port ranges and equal world size.  The hello carries a token of the job
    (SDRTERM_JOB_ID or MASTER_ADDR:MASTER_PORT) and the asking rank: job B's rank knocking on job A's port is passed over, and a
    rank that asks twice counts once."""
    from sdrterm_amd import parallel
    from sdrterm_amd.misc.general_util import findPort
    base = findPort('127.0.0.1')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('SDRTERM_UID_PORT', str(base))
    ids = {'A': bytes([1] * 128), 'B': bytes([2] * 128)}
    got = {}

    def run(job, rank):
        if rank == 0:
            # (unique_id is looked up when rank 0 runs: give each job's rank 0 its own)
            uid = ids[job]
            orig = parallel.RcclTransport.unique_id
            parallel.RcclTransport.unique_id = staticmethod(lambda: uid)
            try:
                got[job, rank] = parallel.RcclTransport._exchange_uid(rank, 2, job=job)
            finally:
                parallel.RcclTransport.unique_id = orig
        else:
            got[job, rank] = parallel.RcclTransport._exchange_uid(rank, 2, job=job)
    ta = threading.Thread(target=run, args=('A', 0), daemon=True)
    ta.start()
    time.sleep(0.5)                                   # job A's rank 0 holds the first port of the range ...
    tb = threading.Thread(target=run, args=('B', 0), daemon=True)
    tb.start()
    time.sleep(0.5)                                   # ... job B's the second
    others = [threading.Thread(target=run, args=(j, 1), daemon=True) for j in ('B', 'A')]
    for t in others:
        t.start()
    for t in [ta, tb] + others:
        t.join(60)
    assert got == {('A', 0): ids['A'], ('A', 1): ids['A'], ('B', 0): ids['B'], ('B', 1): ids['B']}


def test_a_failed_hand_off_fails_the_fan_out_instead_of_hanging_it():
    """a helper that has died takes its rows' clients with it.  Rank 0 must not count such a row as served (it would
    enter the first collective while the owner never gets its socket) nor lose its acceptor thread to the OSError: the fan-out is
    marked failed and wait_for_clients returns False.  A local process that connects to the hand-off socket and says nothing
    must not hold up the TCP accepts either."""
    import struct
    from sdrterm_amd.dsp import fanout
    fo = fanout.RowFanOut('127.0.0.1', 0, 1, bounds=[(0, 1), (1, 2)])
    fo.start()
    port = fo.address[1]
    try:
        silent = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        silent.connect(fanout.handoff_name(port))                 # never says which rank it is
        t0 = time.monotonic()
        c0 = socket.create_connection(('127.0.0.1', port))        # row 0: rank 0's own
        deadline = time.monotonic() + 2
        while not fo.clients and time.monotonic() < deadline:
            time.sleep(0.01)
        assert fo.clients and time.monotonic() - t0 < 2.0         # (the old blocking recv held every accept for five seconds)
        helper = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        helper.connect(fanout.handoff_name(port))
        helper.sendall(struct.pack('<i', 1))
        deadline = time.monotonic() + 2
        while 1 not in fo._peers and time.monotonic() < deadline:
            time.sleep(0.01)
        assert 1 in fo._peers
        helper.close()                                            # the helper dies
        time.sleep(0.05)
        c1 = socket.create_connection(('127.0.0.1', port))        # row 1: cannot reach its rank any more
        assert fo.wait_for_clients(lambda: False, poll=0.05) is False
        assert fo.failed and 'rank 1' in fo.failed
        for s in (silent, c0, c1):
            s.close()
    finally:
        fo.close()
