"""CPU test of the multi-GPU --simo driver: two ranks under gloo run the sharded VfoProcessor end to
end with a stub in place of the GPU engine.  Checked: which rows each rank serves (vfo_processor.py:43-47 order, split
contiguously), which TCP port, that client k of a rank's port receives row lo + k of every chunk as big-endian doubles
(vfo_processor.py:83-84), that every rank sees every block in order and that end of stream reaches all of them."""
import os
import queue
import socket
import threading
import time

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


def _rank_worker(rank, world, master_port, base_port, q, feed='all at once'):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(master_port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = StubVfo(FS, vfoHost=f'127.0.0.1:{base_port}', vfos=VFOS, center=1000, dec=Q, omegaOut=5000, transport='torch')
    p.selectOutputFm()
    p._enterRank(rank, world)
    seen = []                                   # (chunks, end of stream, more queued) of every block this rank took
    take = p._takeBatch

    def spy(buffer):
        chunks, eof = take(buffer)
        seen.append((sum(c.n for c in chunks), eof, p._moreComing))
        return chunks, eof
    p._takeBatch = spy
    if feed != 'all at once':
        p.MAX_BATCH = 2
    sh = p._shard()
    m = N // Q
    box = [None] * sh.rows
    # clients connect one after the other, so that client k is registered k-th (the reference's mapping follows
    # connection order too, example_simo.sh:161)
    def connect_all():
        for k in range(sh.rows):
            t = threading.Thread(target=_client, args=('127.0.0.1', sh.port(base_port), CHUNKS * m * 8, box, k), daemon=True)
            t.start()
            ts.append(t)
            time.sleep(0.3)
    ts = []
    starter = threading.Thread(target=connect_all, daemon=True)
    starter.start()
    buf = None
    if rank == 0:
        buf = queue.Queue()
        if feed == 'all at once':
            for c in chunks_for_test():
                buf.put(c)
            buf.put(b'')
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
    p.processData(_Flag(), buf)
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
    assert [g[1] for g in got] == [(0, 3), (3, 5)] and [g[2] for g in got] == [base, base + 1]
    m = N // Q
    check = [int(np.frombuffer(c.tobytes(), dtype=np.uint8).astype(np.int64).sum() % 251) for c in chunks_for_test()]
    blocks = [g[4] for g in got]
    assert blocks[0] == blocks[1]                                  # every rank saw the same blocks, flags included
    assert sum(b[0] for b in blocks[0]) == CHUNKS and blocks[0][-1][1] and not any(b[1] for b in blocks[0][:-1])
    if feed == 'trickle':
        # the helper rank runs processData(isDead, None): blocks that are not the last one must get through it (round 2: the
        # helper died on None.empty() after the first of them), in order, with and without more queued behind them
        assert len(blocks[0]) >= 4 and any(b[2] for b in blocks[0]) and any(not b[2] and not b[1] for b in blocks[0])
    for _rank, (lo, hi), _port, streams, _seen in got:
        assert len(streams) == hi - lo
        for k, s in enumerate(streams):
            assert s is not None and s.size == CHUNKS * m
            exp = np.concatenate([rows_all[lo + k] + check[c] + np.arange(m) / 1024.0 for c in range(CHUNKS)])
            assert np.array_equal(s, exp), f'row {lo + k}'


def test_rank_zero_releases_the_others_when_halted_before_any_data():
    """The halt flag set before the first block: rank 0 must still hand the others an end-of-stream block (they wait in a
    collective), here with world == 1 semantics checked on the bookkeeping only."""
    p = StubVfo(FS, vfoHost='127.0.0.1', vfos=VFOS, dec=Q, omegaOut=5000, transport='torch')
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


def test_ports_of_every_rank_are_probed_before_helpers_start():
    """Rank r serves port + r: a busy one must stop the start-up in rank 0, before any helper waits in a collective."""
    import socket
    from sdrterm_amd.misc.general_util import findPort
    base = findPort('127.0.0.1')
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as busy:
        busy.bind(('127.0.0.1', base + 1))
        busy.listen(1)
        p = StubVfo(FS, vfoHost=f'127.0.0.1:{base}', vfos=VFOS, dec=Q, omegaOut=5000, gpus=2, device=3)
        assert p._deviceBase == 3
        with pytest.raises(OSError, match='rank 1'):
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
