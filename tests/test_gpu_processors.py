"""GPU end-to-end tests of the processor mirrors, the stand-alone discriminators and stream-level properties."""
import os
import socket
import tempfile
import threading
import time

import numpy as np
import pytest

from tests.util import BY_NAME, case_raw, samples_per_chunk, tol_for

pytestmark = pytest.mark.gpu

from tests.util import set_diag as _diag  # noqa: E402


class _Flag:
    value = 0


class _ListQueue:
    """Queue stand-in fed with the reader's chunks then b'' (read_file.py:169-171)."""

    def __init__(self, items):
        import queue
        self.q = queue.Queue()
        for it in items:
            self.q.put(it)

    def get(self):
        return self.q.get()

    def get_nowait(self):
        return self.q.get_nowait()

    def close(self):
        pass

    def join_thread(self):
        pass


def _reader_chunks(case):
    """What the reference's reader would enqueue for this case (unpack/normalize/correctIq), via the oracle."""
    from oracle import ref_chain
    ch = ref_chain.Chain(case['fs'], case['q'], case['mode'], case['code'], center=case['center'],
                         vfos=case.get('vfos'), omega_out=case['omega_out'], normalize=bool(case.get('normalize')),
                         correct_iq=bool(case.get('correct_iq')), big_endian=bool(case.get('big_endian')))
    raw = case_raw(case)
    n = samples_per_chunk(case['code'])
    item = len(raw) // (2 * n * case['n_chunks'])
    return [ch.read(raw[c * 2 * n * item:(c + 1) * 2 * n * item]) for c in range(case['n_chunks'])]


def test_dsp_processor_process_data_matches_reference(golden):
    from sdrterm_amd.dsp import DspProcessor
    z, _ = golden
    case = BY_NAME['h_q64_fm_c']
    p = DspProcessor(case['fs'], center=case['center'], omegaOut=case['omega_out'], dec=case['q'],
                     fileInfo={'bitsPerSample': 'int16'})
    p.selectOutputFm()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'out.bin')
        p.processData(_Flag(), _ListQueue(_reader_chunks(case) + [b'']), path)
        got = np.fromfile(path, dtype='=f8')
    exp = z['h_q64_fm_c/z']
    assert got.size == exp.size
    assert np.abs(got.reshape(exp.shape) - exp).max() <= tol_for(exp, case['mode'])


def test_dsp_processor_smooth_output_matches_reference(golden):
    """--smooth-output: processData writes savgol_filter(z, smooth, 3) (dsp_processor.py:159-160); _processChunk itself
    stays unsmoothed, as in the reference."""
    from sdrterm_amd.dsp import DspProcessor
    z, _ = golden
    case = BY_NAME['h_q64_fm_c_smooth11']
    p = DspProcessor(case['fs'], center=case['center'], omegaOut=case['omega_out'], dec=case['q'], smooth=case['smooth'],
                     fileInfo={'bitsPerSample': 'int16'})
    p.selectOutputFm()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'out.bin')
        p.processData(_Flag(), _ListQueue(_reader_chunks(case) + [b'']), path)
        got = np.fromfile(path, dtype='=f8')
    exp = z['h_q64_fm_c_smooth11/z']
    assert got.size == exp.size
    assert np.abs(got.reshape(exp.shape) - exp).max() <= tol_for(exp, case['mode'])
    chunks = _reader_chunks(case)
    n = chunks[0].size
    x = np.empty((1, n), complex)
    y = np.empty((1, n // case['q']), complex)
    out = np.empty((1, n // case['q']))
    x[0, :] = chunks[0]
    p._processChunk(x, y, out)
    from scipy.signal import savgol_filter
    assert np.abs(savgol_filter(out, case['smooth'], 3) - exp[0]).max() <= tol_for(exp, case['mode'])
    assert np.abs(out - exp[0]).max() > 1e-4          # i.e. _processChunk did not smooth


def test_process_chunk_signature_matches_reference(golden):
    """_processChunk(x, y, z) as the reference's tests/tools call it (dsp_processor.py:140-149)."""
    from sdrterm_amd.dsp import DspProcessor
    z, _ = golden
    case = BY_NAME['h_q64_am_norm_c']
    chunks = _reader_chunks(case)
    p = DspProcessor(case['fs'], center=case['center'], omegaOut=case['omega_out'], dec=case['q'])
    p.selectOutputAm()
    n = chunks[0].size
    x = np.empty((1, n), complex)
    y = np.empty((1, n // case['q']), complex)
    out = np.empty((1, n // case['q']))
    x[0, :] = chunks[0]
    p._processChunk(x, y, out)
    y0 = z['h_q64_am_norm_c/y0']
    assert np.abs(y - y0).max() <= 1e-9 * np.abs(y0).max()
    assert np.abs(out - z['h_q64_am_norm_c/z'][0]).max() <= tol_for(z['h_q64_am_norm_c/z'][0], case['mode'])


def _client(host, port, nbytes, sink, idx):
    s = None
    for _ in range(400):                             # the acceptor may not be listening yet: retry instead of giving up
        try:
            s = socket.create_connection((host, port), timeout=30)
            break
        except OSError:
            time.sleep(0.05)
    if s is None:
        return
    buf = bytearray()
    while len(buf) < nbytes:
        b = s.recv(1 << 16)
        if not b:
            break
        buf += b
    sink[idx] = bytes(buf)
    s.close()


def test_vfo_processor_simo_fanout_matches_reference(golden):
    """--simo: one TCP client per row, big-endian doubles, rows in the reference's order (vfo_processor.py:80-84)."""
    from sdrterm_amd.dsp import VfoProcessor
    from sdrterm_amd.misc.general_util import findPort
    z, _ = golden
    case = BY_NAME['h_q64_fm_vfo9']
    exp = z['h_q64_fm_vfo9/z']                       # (chunks, 9, m)
    nch, rows, m = exp.shape
    port = findPort('127.0.0.1')
    p = VfoProcessor(case['fs'], vfoHost=f'127.0.0.1:{port}', vfos=case['vfos'], center=case['center'],
                     omegaOut=case['omega_out'], dec=case['q'], fileInfo={'bitsPerSample': 'int16'})
    p.selectOutputFm()
    q = _ListQueue(_reader_chunks(case) + [b''])
    t = threading.Thread(target=p.processData, args=(_Flag(), q), daemon=True)
    t.start()
    sink = [None] * rows
    clients = []
    for k in range(rows):                            # connect in row order, like example_simo.sh does with a delay
        c = threading.Thread(target=_client, args=('127.0.0.1', port, nch * m * 8, sink, k), daemon=True)
        c.start()
        clients.append(c)
        time.sleep(0.15)
    for c in clients:
        c.join(60)
    t.join(60)
    assert not t.is_alive()
    for k in range(rows):
        got = np.frombuffer(sink[k], dtype='>f8')
        assert got.size == nch * m
        assert np.abs(got.reshape(nch, m) - exp[:, k, :]).max() <= tol_for(exp, case['mode'])


def test_standalone_discriminators_known_answers(golden):
    """The reference's 8-sample KATs (test/dsp/demodulation_test.py:19-50) against the GPU discriminators."""
    from sdrterm_amd.dsp import demodulation as dem
    z, _ = golden
    inp = z['kat_in']
    for fn, key in ((dem.fmDemod, 'kat_fm'), (dem.amDemod, 'kat_am'), (dem.realOutput, 'kat_re'), (dem.imagOutput, 'kat_im')):
        out = np.empty((1, 8))
        fn(np.array(inp), out)
        assert np.abs(out - z[key]).max() < 1e-12, key


def test_stream_properties_at_scale():
    """512 chunks (16.8 M samples): batching invariance (bitwise), chunk independence, and oracle parity on a sample."""
    from sdrterm_amd.engine import Engine
    from oracle import c_chain, ref_chain
    from tests.golden import prng
    fs, q, n, nch = 256_000_000, 64, 32768, 512
    raw = prng.raw_bytes('h', n * nch, seed=77, amp=8000)
    eng = Engine(fs, q, 'fm', 'h', row_freqs=[15000], omega_out=12500)
    full = eng.process(raw)                                              # one call
    assert np.isfinite(full).all()
    cb = eng.chunk_bytes
    parts = [eng.process(raw[i * cb:(i + 64) * cb]) for i in range(0, nch, 64)]   # eight calls of 64 chunks
    assert np.array_equal(full, np.concatenate(parts, axis=1))          # chunks are independent: identical bits
    # permuting chunks permutes outputs
    perm = np.random.default_rng(0).permutation(nch)
    rawp = b''.join(raw[i * cb:(i + 1) * cb] for i in perm)
    outp = eng.process(rawp).reshape(nch, eng.m)
    assert np.array_equal(outp, full.reshape(nch, eng.m)[perm])
    # oracle on 16 sampled chunks
    pick = [0, 1, 2, 3, 100, 101, 255, 256, 300, 400, 500, 508, 509, 510, 511, 257]
    sub = b''.join(raw[i * cb:(i + 1) * cb] for i in pick)
    shift = c_chain.shift_table([15000], fs, n)
    exp, _ = c_chain.process_stream(sub, 'h', n, q, 'fm', shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, 12500),
                                    threads=4)
    got = full.reshape(nch, eng.m)[pick]
    assert np.abs(got - exp[:, 0, :]).max() < 1e-6


def test_iq_offset_carries_across_calls():
    """--correct-iq keeps one scalar across chunks (read_file.py:53,103): 1 call of 6 chunks == 6 calls of 1 chunk."""
    from sdrterm_amd.engine import Engine
    from oracle import c_chain, ref_chain
    from tests.golden import prng
    fs, q, n, nch = 1024000, 64, 32768, 6
    raw = prng.raw_bytes('h', n * nch, seed=5, amp=3000, dc=(700, -400))
    a = Engine(fs, q, 'fm', 'h', row_freqs=[15700], omega_out=5000, correct_iq=True)
    b = Engine(fs, q, 'fm', 'h', row_freqs=[15700], omega_out=5000, correct_iq=True)
    one = a.process(raw)
    cb = a.chunk_bytes
    six = np.concatenate([b.process(raw[i * cb:(i + 1) * cb]) for i in range(nch)], axis=1)
    assert np.abs(one - six).max() < 1e-9
    assert abs(a.iq_offset - b.iq_offset) < 1e-9
    shift = c_chain.shift_table([15700], fs, n)
    exp, off = c_chain.process_stream(raw, 'h', n, q, 'fm', shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, 5000),
                                      iq_inductance=50 / fs)
    assert np.abs(one - exp[:, 0, :].reshape(1, -1)).max() < 1e-6
    assert abs(a.iq_offset - off) < 1e-8 * max(1.0, abs(off))
    a.iq_offset = 0j
    again = a.process(raw)
    assert np.abs(again - one).max() < 1e-9


def test_linearity_of_the_decimator():
    """decimate is linear: y(a*x1 + x2) == a*y(x1) + y(x2) on integer inputs (checked on the debug output)."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    n = 32768
    x1 = np.frombuffer(prng.raw_bytes('h', n, 1, 2000), dtype='<i2').astype(np.int32)
    x2 = np.frombuffer(prng.raw_bytes('h', n, 2, 2000), dtype='<i2').astype(np.int32)
    eng = Engine(1024000, 64, 'am', 'h', row_freqs=[15700], omega_out=5000)
    _, y1 = eng.process_debug((x1).astype('<i2').tobytes())
    _, y2 = eng.process_debug((x2).astype('<i2').tobytes())
    _, y3 = eng.process_debug((3 * x1 + x2).astype('<i2').tobytes())
    assert np.abs(y3 - (3 * y1 + y2)).max() <= 1e-9 * np.abs(y3).max()


@pytest.mark.parametrize('name', ['h_q64_fm_iq', 'H_q64_am_norm_c', 'h_be_q64_fm_c', 'B_q64_fm_c', 'f_q64_fm_c'])
def test_raw_reader_feeds_the_gpu_processor(golden, name):
    """File -> this package's readFile (raw bytes on the Queue) -> DspProcessor: unpack / --normalize-input / --correct-iq
    run on the GPU (read_file.py:100-103), outputs equal the reference's for the same file."""
    import queue
    from sdrterm_amd.dsp import DspProcessor
    from sdrterm_amd.misc.file_util import DataType
    from sdrterm_amd.misc.read_file import readFile
    z, _ = golden
    case = BY_NAME[name]
    sample = DataType[case['code']].value
    if case.get('big_endian'):
        sample = sample.newbyteorder('>')

    class Q(_ListQueue):
        def put_nowait(self, x):
            self.q.put(x)

    class P:
        exitcode, name = None, 'writer'

    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, 'in.raw'), os.path.join(d, 'out.bin')
        with open(src, 'wb') as f:
            f.write(case_raw(case))
        buf = Q([])
        readFile(bitsPerSample=sample, fs=case['fs'], buffers=[buf], processes=[P()], isDead=_Flag(), inFile=src,
                 correctIq=bool(case.get('correct_iq')), normalize=bool(case.get('normalize')))
        # (float32 records: the reference's complex64 arithmetic is a mode of the processor, faithfulF32)
        p = DspProcessor(case['fs'], center=case['center'], omegaOut=case['omega_out'], dec=case['q'],
                         fileInfo={'bitsPerSample': str(sample)}, faithfulF32=True)
        {'fm': p.selectOutputFm, 'am': p.selectOutputAm}[case['mode']]()
        p.processData(_Flag(), buf, dst)
        got = np.fromfile(dst, dtype='=f8')
    exp = z[f'{name}/z']
    assert got.size == exp.size
    assert np.abs(got.reshape(exp.shape) - exp).max() <= tol_for(exp, case['mode'])


@pytest.mark.parametrize('code', ['h', 'B'])
def test_full_size_workload_properties(code, monkeypatch):
    """BASELINE.json configs[1] at bench size (2^30 samples = 32768 chunks resident on the device; 'B': the same as uint8 records,
    16384 chunks of sixteen tiles -- two tiles per wave, the trailing column phase of kernels_i8w.hip): size-independent
    properties instead of an oracle run -- (1) two independently written kernels (wide-column persistent int8 kernel vs
    the one-block-per-column int8 kernel) agree on every output, (2) splitting the call does not change a bit, (3) a
    sample of chunks equals the C oracle.  (No torch here: its bundled HIP runtime and the system one this library links
    do not initialise in either order inside one process.)"""
    from oracle import c_chain, ref_chain
    from sdrterm_amd import _native as nat
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    fs, q, center, omega = 256_000_000, 64, 15000, 12500
    n = 131072 // (2 * (2 if code == 'h' else 1))
    n_chunks, base = (1 << 30) // n, 256
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    wide = Engine(fs, q, 'fm', code, row_freqs=[center], omega_out=omega)
    assert wide.stage_a == 'i8w'
    block = np.frombuffer(prng.raw_bytes(code, n * base, seed=99, amp=8000 if code == 'h' else 100), dtype=np.uint8)      # 256 distinct chunks
    d_raw = nat.DeviceBuffer(n_chunks * wide.chunk_bytes)
    for i in range(0, n_chunks, base):
        nat.check(nat.lib().sdrt_memcpy_h2d(d_raw.ptr + i * wide.chunk_bytes, block.ctypes.data, block.size))
    d_a = nat.DeviceBuffer(n_chunks * wide.m * 8)
    d_b = nat.DeviceBuffer(n_chunks * wide.m * 8)

    def fetch(buf):
        nat.check(nat.lib().sdrt_device_synchronize())
        return buf.download(np.empty(n_chunks * wide.m, dtype=np.float64))

    wide.process_device(d_raw.ptr, n_chunks, d_a.ptr)
    out_w = fetch(d_a)
    assert np.isfinite(out_w).all()
    # (2) two calls of unequal size
    cut = 12345 if code == 'h' else 6173
    wide.process_device(d_raw.ptr, cut, d_b.ptr)
    wide.process_device(d_raw.ptr + cut * wide.chunk_bytes, n_chunks - cut, d_b.ptr + cut * wide.m * 8)
    assert np.array_equal(out_w, fetch(d_b))
    # the stream repeats every 256 chunks, so must the output, bit for bit (every workgroup / pipeline slot agrees)
    rep = out_w.reshape(n_chunks // base, base * wide.m)
    assert np.array_equal(rep, np.broadcast_to(rep[0], rep.shape))
    # (1) the other int8 kernel
    monkeypatch.setenv('SDRTERM_STAGE_A', 'i8t')
    narrow = Engine(fs, q, 'fm', code, row_freqs=[center], omega_out=omega)
    assert narrow.stage_a == 'i8t'
    narrow.process_device(d_raw.ptr, n_chunks, d_b.ptr)
    assert np.abs(out_w - fetch(d_b)).max() < (1e-8 if code == 'h' else 1e-7)      # (8-bit white noise: a weaker in-band signal)
    # (3) oracle on a few chunks
    pick = [0, 1, 100, 255]
    sub = b''.join(block[c * wide.chunk_bytes:(c + 1) * wide.chunk_bytes].tobytes() for c in pick)
    shift = c_chain.shift_table([center], fs, n)
    exp, _ = c_chain.process_stream(sub, code, n, q, 'fm', shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, omega), threads=4)
    got = out_w.reshape(n_chunks, wide.m)[[c + 256 * (n_chunks // 256 - 3) for c in pick]]
    assert np.abs(got - exp[:, 0, :]).max() < 1e-6
    for b in (d_raw, d_a, d_b):
        b.free()


def test_edge_cases_empty_ragged_and_small_batches(golden):
    """Empty input, input that is not a whole number of chunks, a stream that ends at once, and every batch size around
    the depth of the wide kernel's software pipeline (1..5 chunks, then odd sizes): same outputs as one call."""
    from sdrterm_amd.dsp import DspProcessor
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    eng = Engine(1024000, 64, 'fm', 'h', row_freqs=[15700], omega_out=5000)
    assert eng.stage_a == 'i8w'
    assert eng.process(b'', 0).shape == (1, 0)                            # nothing in, nothing out
    with pytest.raises(ValueError):
        eng.process(b'\x00' * (eng.chunk_bytes + 6))                      # ragged: the caller must pad, as the reader does
    nch = 23
    raw = prng.raw_bytes('h', eng.n * nch, seed=5, amp=6000)
    full = eng.process(raw).reshape(nch, eng.m)
    cb = eng.chunk_bytes
    pos, sizes = 0, [1, 2, 3, 4, 5, 7, 1]
    for k in sizes:
        part = eng.process(raw[pos * cb:(pos + k) * cb]).reshape(k, eng.m)
        assert np.array_equal(part, full[pos:pos + k])
        pos += k
    assert pos == nch
    # a stream that is only EOF: the writer produces an empty file and returns
    p = DspProcessor(1024000, center=15700, omegaOut=5000, dec=64, fileInfo={'bitsPerSample': 'int16'})
    p.selectOutputFm()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'out.bin')
        p.processData(_Flag(), _ListQueue([b'']), path)
        assert os.path.getsize(path) == 0


def test_rccl_transport_single_rank_roundtrip():
    """sdrt_comm_* through the C ABI with a world of one (what a 1-GPU box can run): librccl loads, the communicator
    initialises on device 0, broadcast and all-gather of device buffers return the bytes, BlockChannel frames a block."""
    from sdrterm_amd import _native as nat, parallel
    t = parallel.RcclTransport(0, 1, 0, parallel.RcclTransport.unique_id())
    try:
        data = np.arange(4096, dtype=np.uint8).tobytes() * 3
        assert t.bcast_bytes(data, len(data)) == data
        ch = parallel.BlockChannel(t, chunk_bytes=4096)
        blk = ch.exchange(data, 3, False, more=True)
        assert (blk.n, blk.eof, blk.more, blk.seq, blk.slot) == (3, False, True, 0, 0) and blk.dev_ptr
        t.block_ready(blk.slot, 0)                       # the block is only queued: the reading stream waits for it
        back = np.empty(len(data), dtype=np.uint8)
        nat.check(nat.lib().sdrt_memcpy_d2h(back.ctypes.data, blk.dev_ptr, len(data)))
        assert back.tobytes() == data
        t.block_consumed(blk.slot, 0)
        nxt = ch.exchange(data[::-1], 3, False)
        assert (nxt.seq, nxt.slot, nxt.more) == (1, 1, False) and nxt.dev_ptr != blk.dev_ptr      # the other buffer
        t.block_ready(nxt.slot, 0)
        nat.check(nat.lib().sdrt_memcpy_d2h(back.ctypes.data, nxt.dev_ptr, len(data)))
        assert back.tobytes() == data[::-1]
        end = ch.exchange(b'', 0, True)
        assert (end.n, end.eof, end.seq) == (0, True, 2)
        src = nat.DeviceBuffer(1024).upload(np.full(1024, 7, dtype=np.uint8))
        dst = nat.DeviceBuffer(1024)
        t.allgather_device(src.ptr, dst.ptr, 1024)
        nat.check(nat.lib().sdrt_device_synchronize())
        assert (dst.download(np.empty(1024, dtype=np.uint8)) == 7).all()
    finally:
        t.close()


def test_simo_block_channel_path_on_one_gpu():
    """The multi-GPU --simo data path with a world of one (all a 1-GPU box can run): every raw block goes through the header
    exchange, the page-locked upload and the RCCL broadcast into one of two device buffers on the communication stream, the
    kernels wait for it behind an event on the processor's stream, the next broadcast into the same buffer waits for them, the
    copy-out is collected with a bounded wait.  Five blocks (batches of two chunks, so that several are in flight), three rows,
    three socket clients: every row's stream must equal the oracle's, in order."""
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
    raw = prng.raw_bytes('h', n * nch, seed=33, amp=4000)
    port = findPort('127.0.0.1')
    p = VfoProcessor(fs, vfoHost=f'127.0.0.1:{port}', vfos='-100000,50000', center=15700, dec=q, omegaOut=5000, forceChannel=True)
    p.selectOutputFm()
    p.MAX_BATCH = 2
    buf = queue.Queue()
    for c in range(nch):
        buf.put(RawChunk(raw[c * n * 4:(c + 1) * n * 4], 'h', False, False, False, 50))
    buf.put(b'')
    buf.close = lambda: None
    buf.join_thread = lambda: None
    m = n // q
    box = [None] * 3

    def client(k):
        deadline = time.time() + 60
        while True:
            try:
                s = socket.create_connection(('127.0.0.1', port), timeout=30)
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
        for k in range(3):
            threading.Thread(target=client, args=(k,), daemon=True).start()
            time.sleep(0.3)
    threading.Thread(target=connect_all, daemon=True).start()

    class _Flag:
        value = 0
    seen = []
    take = p._takeBatch

    def spy(b):
        chunks, eof = take(b)
        seen.append((sum(c.n for c in chunks), [c.slot for c in chunks], p._moreComing, eof))
        return chunks, eof
    p._takeBatch = spy
    p.processData(_Flag(), buf)
    deadline = time.time() + 30
    while any(b is None for b in box) and time.time() < deadline:
        time.sleep(0.05)
    assert sum(s[0] for s in seen) == nch and len(seen) >= 5 and seen[-1][3]
    assert [s[1] for s in seen if s[1]] == [[k % 2] for k in range(len([s for s in seen if s[1]]))]      # the two buffers in turn
    assert any(s[2] for s in seen)                                                                  # blocks kept in flight
    rows = ref_chain.vfo_rows(15700, '-100000,50000')
    shift = c_chain.shift_table(rows, fs, n)
    exp, _ = c_chain.process_stream(raw, 'h', n, q, 'fm', shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, 5000), threads=4)
    for k in range(3):
        assert box[k] is not None and box[k].size == nch * m
        assert np.abs(box[k] - exp[:, k, :].reshape(-1)).max() <= 1e-6, f'row {k}'


@pytest.mark.parametrize('mode', ['re', 'im'])
def test_working_real_and_imaginary_streaming_modes(mode, tmp_path):
    """selectOutputReal / selectOutputImag (dsp_processor.py:130-138): the reference itself raises there (sosfilt of an empty
    filter list); with workingReIm the processor writes the real / imaginary part of the decimated samples
    (demodulation.py:51-68), checked against the oracle's decimated samples."""
    import queue
    from oracle import ref_chain
    from sdrterm_amd.dsp.dsp_processor import DspProcessor
    from tests.golden import prng
    fs, q, n, nch = 1024000, 64, 32768, 3
    raw = prng.raw_bytes('h', n * nch, seed=91, amp=4000)
    ch = ref_chain.Chain(fs, q, 'fm', 'h', center=15700, omega_out=5000)
    p = DspProcessor(fs, center=15700, omegaOut=5000, dec=q, fileInfo={'bitsPerSample': 'int16'}, workingReIm=True)
    (p.selectOutputReal if mode == 're' else p.selectOutputImag)()
    buf = queue.Queue()
    exp = []
    for c in range(nch):
        z = ch.read(raw[c * n * 4:(c + 1) * n * 4])
        buf.put(z)
        _, y = ref_chain.process_chunk(z, q, 'fm', ch.shift, ch.out_sos, return_decimated=True)
        exp.append(y[0].real if mode == 're' else y[0].imag)
    buf.put(b'')
    buf.close = lambda: None
    buf.join_thread = lambda: None

    class _Flag:
        value = 0
    f = tmp_path / 'out.bin'
    p.processData(_Flag(), buf, str(f))
    got = np.frombuffer(f.read_bytes(), dtype='=f8')
    exp = np.concatenate(exp)
    assert got.size == exp.size
    assert np.abs(got - exp).max() <= 1e-6 * max(1.0, np.abs(exp).max())


@pytest.mark.parametrize('code,normalize', [('h', False), ('B', True), ('f', False)])
def test_plot_fft_taps_match_the_reference_plots_arithmetic(code, normalize):
    """Spectrum analyser and waterfall arrays (plots/spectrum_analyzer_plot.py:74-88, plots/waterfall_plot.py:90-100) from raw
    chunks on the GPU against the numpy/scipy restatement of those lines.  Compared as amplitudes relative to the largest bin
    (the plots take logarithms: empty bins are rounding noise in both)."""
    from oracle import ref_chain
    from sdrterm_amd.misc.read_file import RawChunk
    from sdrterm_amd.plots import SpectrumTap, WaterfallTap
    from tests.golden import prng
    fs, off = 1024000, 15700
    n = samples_per_chunk_of(code)
    raw = prng.raw_bytes(code, n, seed=5, amp=100 if code == 'B' else 3000)
    z = ref_chain.unpack(raw, code, False)
    if normalize:
        z = ref_chain.normalize(z, code)
    chunk = RawChunk(raw, code, False, normalize, False, 50)
    got = SpectrumTap(fs, center=off)(chunk)
    exp = ref_chain.spectrum_tap(z, fs, off)
    ga, ea = 10.0 ** (got / 2), 10.0 ** (exp / 2)
    assert got.shape == exp.shape and np.abs(ga - ea).max() <= 1e-9 * ea.max()
    img = WaterfallTap(fs, center=off)(chunk)
    ref = ref_chain.waterfall_tap(z, fs, off)
    gi, ri = 10.0 ** (img / 10), 10.0 ** (ref / 10)
    assert img.shape == ref.shape == (1024, n // 128 + 1) and np.abs(gi - ri).max() <= 1e-9 * ri.max()
    # the reference reader's complex arrays go the same way
    got2 = SpectrumTap(fs, center=off)(z.astype(np.complex128))
    assert np.abs(10.0 ** (got2 / 2) - ea).max() <= 1e-9 * ea.max()
    # ... and against what the reference's own plots computed in update() (tests/golden/make_golden.py: plot_taps; float32
    # records: the reference transforms its complex64 chunk in single precision)
    from tests.util import load_golden
    from tests.golden.cases import TAP_FS, TAP_OFFSET, TAP_SEED
    assert (fs, off, 5) == (TAP_FS, TAP_OFFSET, TAP_SEED)
    gz, _ = load_golden()
    tol = 2e-6 if code == 'f' else 1e-9
    ra = 10.0 ** (gz[f'tap/{code}/amp'].astype(np.float64) / 2)
    assert np.abs(ga - ra).max() <= tol * ra.max()
    cols = gz[f'tap/{code}/image_cols']
    rimg = 10.0 ** (gz[f'tap/{code}/image'].astype(np.float64) / 10)
    assert np.abs(gi[:, cols] - rimg).max() <= tol * rimg.max()


def samples_per_chunk_of(code):
    return 131072 // (2 * {'b': 1, 'B': 1, 'h': 2, 'H': 2, 'i': 4, 'I': 4, 'f': 4, 'd': 8}[code])


@pytest.mark.gpu
@pytest.mark.parametrize('code,fused', [('h', '1'), ('h', '0'), ('f', '0')])
def test_iq_offset_of_a_chunk_range_and_a_stream_processed_in_two_parts(code, fused, monkeypatch):
    """sdrt_iq_range_device: what a range of chunks does to the carried offset (off_end = a off_start + b) equals the oracle's
    sequential recursion; and a stream processed as two ranges, the second engine started from the folded offset
    (parallel.fold_iq_pairs: what two GPUs sharing the stream by chunk do), gives the outputs of the stream processed whole."""
    from sdrterm_amd.engine import Engine
    from sdrterm_amd.parallel import fold_iq_pairs
    from oracle import ref_chain
    from tests.golden import prng
    _diag(monkeypatch, iq_fused=fused)
    fs, q, nch, split = 1024000, 64, 7, 3
    kw = dict(row_freqs=[15700.0], apply_shift=True, omega_out=5000, correct_iq=True)
    whole = Engine(fs, q, 'fm', code, **kw)
    n = whole.n
    raw = prng.raw_bytes(code, n * nch, seed=5, amp=3000)
    cb = whole.chunk_bytes
    exp_out = whole.process(raw)
    end_whole = whole.iq_offset
    a0, b0 = whole.iq_range(raw[:split * cb])
    a1, b1 = whole.iq_range(raw[split * cb:])
    assert whole.iq_offset == end_whole                      # the engine's own state is left alone
    start1, end = fold_iq_pairs([(a0, b0), (a1, b1)], 1, 0j)
    assert abs(end - end_whole) <= 1e-10 * abs(end_whole)    # the fused form takes its chunk aggregates from the int8 planes (2^-39)
    first, second = Engine(fs, q, 'fm', code, **kw), Engine(fs, q, 'fm', code, **kw)
    out0 = first.process(raw[:split * cb])
    assert abs(first.iq_offset - b0) <= 1e-10 * abs(b0)
    second.iq_offset = start1
    out1 = second.process(raw[split * cb:])
    m = n // q
    got = np.concatenate([out0.reshape(1, split, m), out1.reshape(1, nch - split, m)], axis=1).reshape(exp_out.shape)
    err = np.abs(got - exp_out).max()
    print(f'{code} fused={fused}: two ranges against the whole stream, max abs difference {err:.3e}')
    assert err <= 1e-9
    # the oracle's recursion on the same samples
    chain = ref_chain.Chain(fs, q, 'fm', code, center=15700, omega_out=5000, correct_iq=True)
    for c in range(nch):
        chain.process(raw[c * cb:(c + 1) * cb])
    assert abs(chain.off - end_whole) <= 1e-9 * abs(end_whole)


@pytest.mark.parametrize('argv', [['--workload', 'simo64', '--feed', 'bcast', '--chunks', '32'],
                                  ['--workload', 'simo64', '--feed', 'allgather', '--chunks', '32'],
                                  ['--workload', 'simo8', '--chunks', '64'],
                                  ['--workload', 'fm64', '--chunks', '256'],
                                  ['--workload', 'fm64', '--chunks', '256', '--correct-iq']])
def test_bench_rank_plumbing_on_one_gpu(argv):
    """bench.py's multi-rank plumbing with a single rank (--force-comm): the RCCL communicator behind the C ABI comes up inside
    the bench process (no torch in it: buffers, streams and events are the C ABI's), barrier and maximum over ranks go through it,
    blocks reach the rank by broadcast or all-gather on the communication stream beside the kernels -- and the line it prints
    carries what the contract asks for, every number measured by that run, with the sampled comparison against the oracle."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--force-comm', '--steps', '3', '--warmup', '1', '--settle-seconds', '0.2',
                          '--long-seconds', '0', '--no-cpu-baseline'] + argv, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0 and d['unit'] == 'MSamples/s' and d['higher_is_better'] is True
    assert d['roofline']['bound'] in ('hbm', 'mfma') and 0 < d['roofline']['frac'] < 1
    assert d['vs_baseline'] is None and d['data'] == 'synthetic' and 'workload' in d['config']
    par = d['parity']
    print(argv, 'parity:', par)
    assert par['ok'] is True and par['max_abs_err'] <= 1e-6 and len(par['chunks']) >= 4 and 'oracle/chain.c' in par['checker']
    if argv[1] == 'simo8':
        assert d['roofline']['bound'] == 'mfma' and 'hbm' in d['roofline'] and par['am_max_err_rel_to_peak'] <= 1e-6
        assert 'FM+AM mixed' in d['config']['workload'] and '--correct-iq' in d['config']['workload'] and d['config']['rows_per_gpu'] == 8
    if argv[1] == 'fm64' and '--correct-iq' not in argv:
        fp = d['roofline']['from_profile']                      # another batch size than the committed counter record's: quoted scaled, and said so
        assert fp is None or (fp['traffic_scaled_from_chunks'] == 131072 and d['roofline']['traffic'] == fp['traffic'])
        assert d['same_arithmetic']['stage_a'] == 'f64' and 0 < d['same_arithmetic']['path_frac'] < 1      # (tiny batch: no ranking of the two)
        assert 0 < d['roofline']['read_frac'] < d['roofline']['path_frac']


def test_bench_py_uses_no_torch():
    """north_star: "no PyTorch".  bench.py and everything it imports on the measured path stay clear of torch: a run with an
    import hook that refuses torch still prints its line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hook = ("import sys, runpy\n"
            "class _No:\n"
            "    def find_spec(self, name, path=None, target=None):\n"
            "        if name == 'torch' or name.startswith('torch.'):\n"
            "            raise ImportError('torch is not to be imported by bench.py')\n"
            "sys.meta_path.insert(0, _No())\n"
            f"sys.argv = ['bench.py', '--steps', '2', '--warmup', '1', '--settle-seconds', '0.1', '--long-seconds', '0', '--no-cpu-baseline', '--no-same-arithmetic', '--chunks', '128']\n"
            f"runpy.run_path({os.path.join(root, 'bench.py')!r}, run_name='__main__')\n")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['SDRTERM_SYSTEM_HIP'] = '1'                      # and the library binds the system HIP runtime, not the one a torch wheel bundles
    out = subprocess.run([sys.executable, '-c', hook], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d['value'] > 0 and d['parity']['ok'] is True


def test_reserve_and_error_paths_of_the_abi():
    """sdrt_reserve: every scratch buffer a batch of that size needs exists afterwards, and a processing call of that size leaves the
    library's count of device allocations where it was (no allocation inside sdrt_process_*); row_modes outside FM / AM are refused with a message;
    the tap entry points keep their buffers until sdrt_taps_release."""
    import ctypes as C
    from sdrterm_amd import _native as nat
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    fs, q, n, nch = 1024000, 64, 32768, 48
    raw = prng.raw_bytes('h', n * nch, seed=12, amp=3000)
    eng = Engine(fs, q, 'fm', 'h', row_freqs=[15700.0, -1000.0], apply_shift=True, omega_out=5000, correct_iq=True, reserve_chunks=nch)
    d_raw = nat.DeviceBuffer(len(raw)).upload(np.frombuffer(raw, dtype=np.uint8))
    d_out = nat.DeviceBuffer(2 * nch * eng.m * 8)
    lib = nat.lib()
    lib.sdrt_alloc_count.restype = C.c_ulonglong
    lib.sdrt_mem_info.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    free, tot = C.c_size_t(), C.c_size_t()
    nat.check(lib.sdrt_mem_info(C.byref(free), C.byref(tot)))
    assert 0 < free.value <= tot.value
    n0 = lib.sdrt_alloc_count()
    eng.process_device(d_raw.ptr, nch, d_out.ptr)
    a, b = eng.iq_range_device(d_raw.ptr, nch)
    nat.check(lib.sdrt_device_synchronize())
    assert lib.sdrt_alloc_count() == n0, 'a processing call within the reserved size allocated device memory'
    # ... also with per-row modes (the FM rows' demodulation launch keeps a stash of its own)
    mixed = Engine(fs, q, 'fm', 'h', row_freqs=[15700.0, -1000.0], row_modes=['fm', 'am'], apply_shift=True, omega_out=5000, correct_iq=True,
                   reserve_chunks=nch)
    mixed.process_device(d_raw.ptr, 1, d_out.ptr)
    nat.check(lib.sdrt_device_synchronize())
    nm = lib.sdrt_alloc_count()
    mixed.process_device(d_raw.ptr, nch, d_out.ptr)
    nat.check(lib.sdrt_device_synchronize())
    assert lib.sdrt_alloc_count() == nm, 'a reserved context with per-row modes allocated device memory while processing'
    small = Engine(fs, q, 'fm', 'h', row_freqs=[15700.0, -1000.0], apply_shift=True, omega_out=5000, correct_iq=True)
    n1 = lib.sdrt_alloc_count()
    small.process_device(d_raw.ptr, nch, d_out.ptr)
    nat.check(lib.sdrt_device_synchronize())
    assert lib.sdrt_alloc_count() > n1                      # (the counter does see what an unreserved context does)
    assert 0 < a < 1 and np.isfinite(b.real)
    with pytest.raises(ValueError):
        Engine(fs, q, 'fm', 'h', row_freqs=[0.0, 1.0], row_modes=['fm', 're'], omega_out=5000)
    # at the ABI itself: a real / imaginary entry among the row modes is refused with a message
    good = Engine(fs, q, 'fm', 'h', row_freqs=[0.0, 1.0], row_modes=['fm', 'am'], omega_out=5000)
    assert good.row_modes == ['fm', 'am']
    bad = np.ascontiguousarray([0, 3], dtype=np.int32)
    import scipy.signal as sg
    dec = np.ascontiguousarray(sg.cheby1(8, 0.05, 0.8 / q, output='sos'))
    outs = np.ascontiguousarray(sg.ellip(3, 1, 30, 5000, output='sos', fs=fs // q))
    freqs = np.ascontiguousarray([0.0, 1.0])
    cfg = nat.Config(struct_size=C.sizeof(nat.Config), device=0, fs=float(fs), decimation=q, mode=0, dtype=2, n_samples=n, n_rows=2,
                     apply_shift=1, dec_sections=4, out_sections=outs.shape[0])
    cfg.row_freqs = freqs.ctypes.data_as(C.POINTER(C.c_double))
    cfg.dec_sos = dec.ctypes.data_as(C.POINTER(C.c_double))
    cfg.out_sos = outs.ctypes.data_as(C.POINTER(C.c_double))
    cfg.row_modes = bad.ctypes.data_as(C.POINTER(C.c_int32))
    ctx = C.c_void_p()
    assert lib.sdrt_create(C.byref(cfg), C.byref(ctx)) == nat.EINVAL and b'row_modes[1]' in lib.sdrt_last_error()
    assert lib.sdrt_taps_release() == 0
