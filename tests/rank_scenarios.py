"""Scenarios of tests/test_gpu_ranks.py that put rank 0 of a VfoProcessor into the calling process: each runs in a fresh
interpreter (python -m tests.rank_scenarios NAME) so that the transport named by SDRTERM_RCCL_LIB is the one this process loads
(a test process that has used RCCL before keeps it).  Exit code 0 = the scenario's assertions held."""
import sys

import numpy as np


def simo_rows_over_rank_processes(gpus=2):
    """VfoProcessor(gpus=2): rank 0 in this process, one helper process, three rows split 2 + 1, every client on the ONE port of
    the job (the k-th connection is row k; the helper's row reaches it as a handed-over socket); nine chunks in batches of two
    (several blocks in flight), every row's stream against oracle/chain.c, in order."""
    import multiprocessing as mp
    import queue
    import socket
    import threading
    import time
    from oracle import c_chain, ref_chain
    from sdrterm_amd.dsp.vfo_processor import VfoProcessor
    from sdrterm_amd.misc.general_util import findPort
    from sdrterm_amd.misc.read_file import RawChunk
    from tests.golden import prng
    fs, q, n, nch = 1024000, 64, 32768, 9
    raw = prng.raw_bytes('h', n * nch, seed=34, amp=4000)
    port = findPort('127.0.0.1')
    vfos = '-100000,50000' if gpus == 2 else '-100000,50000,120000,-30000'
    p = VfoProcessor(fs, vfoHost=f'127.0.0.1:{port}', vfos=vfos, center=15700, dec=q, omegaOut=5000, gpus=gpus)
    p.selectOutputFm()
    p.MAX_BATCH = 2
    buf = queue.Queue()
    for c in range(nch):
        buf.put(RawChunk(raw[c * n * 4:(c + 1) * n * 4], 'h', False, False, False, 50))
    buf.put(b'')
    buf.close = lambda: None
    buf.join_thread = lambda: None
    m = n // q
    # one port, as in the reference (vfo_processor.py:106): connection k is row k; rank 0 computes rows 0, 1, the other ranks one each
    where = [(port, k) for k in range(gpus + 1)]
    box = [None] * len(where)

    def client(prt, k):
        deadline = time.time() + 120
        while True:
            try:
                s = socket.create_connection(('127.0.0.1', prt), timeout=60)
                break
            except OSError:
                if time.time() > deadline:
                    raise
                time.sleep(0.05)
        data = b''
        with s:
            while len(data) < nch * m * 8:
                part = s.recv(nch * m * 8 - len(data))
                if not part:
                    break
                data += part
        box[k] = np.frombuffer(data, dtype='>f8').astype(np.float64)

    def connect_all():
        for prt, k in where:
            threading.Thread(target=client, args=(prt, k), daemon=True).start()
            time.sleep(0.5)
    threading.Thread(target=connect_all, daemon=True).start()
    flag = mp.get_context('spawn').Value('b', 0)
    p.processData(flag, buf)
    deadline = time.time() + 60
    while any(b is None for b in box) and time.time() < deadline:
        time.sleep(0.05)
    assert flag.value == 0, 'a rank raised the halt flag'
    rows = ref_chain.vfo_rows(15700, vfos)
    exp, _ = c_chain.process_stream(raw, 'h', n, q, 'fm', c_chain.shift_table(rows, fs, n), ref_chain.cheby_sos(q),
                                    ref_chain.ellip_sos(fs // q, 5000), threads=4)
    for k in range(len(where)):
        assert box[k] is not None and box[k].size == nch * m, f'row {k}: {None if box[k] is None else box[k].size} samples'
        err = np.abs(box[k].reshape(nch, m) - exp[:, k, :]).max()
        assert err <= 1e-6, f'row {k}: {err:.3e}'


def a_helper_rank_that_dies():
    """a rank that fails must not leave the others in a collective.  Two ranks, a source that trickles; the helper
    process is killed after the first blocks: rank 0 comes back from processData within the deadlines, with the halt flag up."""
    import multiprocessing as mp
    import queue
    import threading
    import time
    from sdrterm_amd.dsp.vfo_processor import VfoProcessor
    from sdrterm_amd.misc.general_util import findPort
    from sdrterm_amd.misc.read_file import RawChunk
    from tests.golden import prng
    fs, q, n = 1024000, 64, 32768
    raw = prng.raw_bytes('h', n * 2, seed=35, amp=4000)
    port = findPort('127.0.0.1')
    p = VfoProcessor(fs, vfoHost=f'127.0.0.1:{port}', vfos='-100000,50000', center=15700, dec=q, omegaOut=5000, gpus=2)
    p.selectOutputFm()
    p.MAX_BATCH = 1
    buf = queue.Queue()
    buf.close = lambda: None
    buf.join_thread = lambda: None
    stop = threading.Event()

    def feeder():
        while not stop.is_set():
            buf.put(RawChunk(raw[:n * 4], 'h', False, False, False, 50))
            time.sleep(0.2)

    def clients():
        import socket
        socks = []
        for prt in (port, port, port):
            deadline = time.time() + 60
            while time.time() < deadline:
                try:
                    socks.append(socket.create_connection(('127.0.0.1', prt), timeout=60))
                    break
                except OSError:
                    time.sleep(0.05)
            time.sleep(0.5)
        while not stop.is_set():
            for s in socks:
                try:
                    s.settimeout(0.05)
                    s.recv(1 << 20)
                except OSError:
                    pass

    def killer():
        deadline = time.time() + 60
        while not p._helpers and time.time() < deadline:
            time.sleep(0.05)
        time.sleep(6.0)                                   # clients connected, a few blocks through
        for h in p._helpers:
            h.kill()
    for fn in (feeder, clients, killer):
        threading.Thread(target=fn, daemon=True).start()
    flag = mp.get_context('spawn').Value('b', 0)
    t0 = time.time()
    p.processData(flag, buf)
    took = time.time() - t0
    stop.set()
    assert flag.value == 1, 'the halt flag stayed down'
    assert took < 6.0 + 45.0, f'rank 0 took {took:.1f} s to give up'


def halt_flag_with_live_helpers():
    """the halt flag goes up in mid-stream while the helper rank waits in a collective.  The helper must stay for the
    end-of-stream block rank 0 sends on its way out and exit with code 0; rank 0 must come back promptly (no teardown behind a
    collective nobody completes)."""
    import multiprocessing as mp
    import queue
    import threading
    import time
    from sdrterm_amd.dsp.vfo_processor import VfoProcessor
    from sdrterm_amd.misc.general_util import findPort
    from sdrterm_amd.misc.read_file import RawChunk
    from tests.golden import prng
    fs, q, n = 1024000, 64, 32768
    raw = prng.raw_bytes('h', n * 2, seed=36, amp=4000)
    port = findPort('127.0.0.1')
    p = VfoProcessor(fs, vfoHost=f'127.0.0.1:{port}', vfos='-100000,50000', center=15700, dec=q, omegaOut=5000, gpus=2)
    p.selectOutputFm()
    p.MAX_BATCH = 1
    buf = queue.Queue()
    buf.close = lambda: None
    buf.join_thread = lambda: None
    stop = threading.Event()
    flag = mp.get_context('spawn').Value('b', 0)
    codes = []

    def feeder():
        while not stop.is_set():
            buf.put(RawChunk(raw[:n * 4], 'h', False, False, False, 50))
            time.sleep(0.2)

    def clients():
        import socket
        socks = []
        for _ in range(3):
            deadline = time.time() + 60
            while time.time() < deadline:
                try:
                    socks.append(socket.create_connection(('127.0.0.1', port), timeout=60))
                    break
                except OSError:
                    time.sleep(0.05)
            time.sleep(0.5)
        while not stop.is_set():
            for s in socks:
                try:
                    s.settimeout(0.05)
                    s.recv(1 << 20)
                except OSError:
                    pass

    def halter():
        deadline = time.time() + 60
        while not p._helpers and time.time() < deadline:
            time.sleep(0.05)
        time.sleep(6.0)                                   # clients connected, a few blocks through
        helpers = list(p._helpers)
        flag.value = 1
        for h in helpers:
            h.join(40)
            # (rank 0's own thread joins its helpers as well: whichever thread reaps the child stores the exit code, the other one
            # may look a moment too early)
            for _ in range(100):
                if h.exitcode is not None:
                    break
                time.sleep(0.05)
            codes.append(h.exitcode)
    ths = [threading.Thread(target=fn, daemon=True) for fn in (feeder, clients, halter)]
    for t in ths:
        t.start()
    t0 = time.time()
    p.processData(flag, buf)
    took = time.time() - t0
    stop.set()
    ths[2].join(60)
    assert took < 6.0 + 20.0, f'rank 0 took {took:.1f} s to leave'
    assert codes == [0], f'helper exit codes {codes}: it did not leave on the end-of-stream block'


def chunk_sharded_stream_with_correct_iq():
    """SURVEY 8e, "by chunk", with --correct-iq: ONE stream of chunks split into contiguous ranges over the ranks.  The carried
    offset (read_file.py:53,64-77) is the only thing a chunk hands to the next; every rank asks what its range does to it
    (sdrt_iq_range_device), the 24-byte pairs are all-gathered through the communicator, every rank folds the ranges before it into
    its starting offset and processes its range.  Every rank's outputs against oracle/chain.c run over the WHOLE stream; the
    verdicts are gathered and rank 0 reports.  Started once per rank (RANK / WORLD_SIZE in the environment); without them this
    process starts the ranks."""
    import os
    import struct
    import subprocess
    if 'WORLD_SIZE' not in os.environ:
        import bench
        world = len(os.environ['SDRTERM_DEVICE_MAP'].split(','))
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, '-m', 'tests.rank_scenarios', 'iqshard'], env=env) for env in bench.rank_environments(world, port)]
        codes = [p.wait(timeout=500) for p in procs]
        assert codes == [0] * world, f'rank exit codes {codes}'
        return
    from oracle import c_chain, ref_chain
    from sdrterm_amd import parallel
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    rank, local, world = parallel.env_rank_world()
    dev = parallel.device_for(local)
    fs, q, n, nch = 1024000, 64, 32768, 7 * world + 3                  # uneven ranges
    raw = prng.raw_bytes('h', n * nch, seed=58, amp=3000, dc=(700, -450))
    lo, hi = parallel.shard_chunks(nch, world, rank)
    t = parallel.RcclTransport(rank, world, dev)
    eng = Engine(fs, q, 'fm', 'h', row_freqs=[15700], omega_out=5000, correct_iq=True, device=dev)
    mine = raw[lo * n * 4:hi * n * 4]
    a, b = eng.iq_range(mine)
    start, end = parallel.chain_iq_offsets(t, a, b, off0=0j)
    eng.iq_offset = start
    out = eng.process(mine).reshape(hi - lo, n // q)
    exp, off_end = c_chain.process_stream(raw, 'h', n, q, 'fm', c_chain.shift_table([15700], fs, n), ref_chain.cheby_sos(q),
                                          ref_chain.ellip_sos(fs // q, 5000), iq_inductance=50 / fs, threads=2)
    err = float(np.abs(out - exp[lo:hi, 0, :]).max())
    ok = err <= 1e-6 and abs(end - off_end) <= 1e-9 * max(1.0, abs(off_end))
    votes = [struct.unpack('<dd', v) for v in t.allgather_bytes(struct.pack('<dd', 1.0 if ok else 0.0, err))]
    t.close()
    print(f'rank {rank}: chunks [{lo}, {hi}), start offset {start:.6g}, max abs err {err:.3e}', flush=True)
    assert all(v[0] == 1.0 for v in votes), f'per-rank (ok, err): {votes}'


if __name__ == '__main__':
    {'iqshard': chunk_sharded_stream_with_correct_iq, 'rows': simo_rows_over_rank_processes, 'rows4': lambda: simo_rows_over_rank_processes(4), 'kill': a_helper_rank_that_dies,
     'halt': halt_flag_with_live_helpers}[sys.argv[1]]()
    print('scenario ok')
