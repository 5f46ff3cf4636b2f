"""CPU-only tests: host logic of the processor mirrors (modelled on the reference's test/dsp/dsp_processor_test.py
and test/misc/*), the C-ABI surface of the shared library, and the multi-GPU partitioning under gloo."""
import ctypes
import json
import math
import os
import pickle
import re
from tests.torch_transport import TorchTransport, max_over_ranks  # noqa: F401

import numpy as np
import pytest

from sdrterm_amd import _native
from sdrterm_amd.dsp import demodulation as dem
from sdrterm_amd.dsp.data_processor import DataProcessor
from sdrterm_amd.dsp.dsp_processor import DspProcessor
from sdrterm_amd.dsp.vfo_processor import VfoProcessor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_FS = 48000
DEFAULT_CENTER = -1000


@pytest.fixture
def processor():
    return DspProcessor(DEFAULT_FS, omegaOut=250)


def test_processor_properties_and_errors(processor):
    """Same checks as the reference's test_init (test/dsp/dsp_processor_test.py:21-75)."""
    assert processor.fs == DEFAULT_FS
    assert processor.decimation == 2
    assert processor.decimatedFs == DEFAULT_FS >> 1
    processor.selectOutputFm()
    assert processor.demod == dem.fmDemod
    processor.selectOutputAm()
    assert processor.demod == dem.amDemod
    assert np.asarray(processor._outputFilters).shape == (2, 6)
    with pytest.raises(ValueError):
        processor.decimation = 1
    with pytest.raises(AttributeError):
        processor.decimatedFs = 1
    with pytest.raises(ValueError):
        processor._setDemod(None, ())
    with pytest.raises(TypeError):
        processor._setDemod("asdf", None)
    processor.decimation = 3
    assert processor.decimation == 3
    assert processor.decimatedFs == DEFAULT_FS / 3
    processor._generateShift(8)
    assert processor._shift is None
    processor.centerFreq = DEFAULT_CENTER
    processor._generateShift(8)
    assert processor._shift.size == 8
    for k in range(8):
        assert processor._shift[0][k] == np.pow(math.e, -2j * math.pi * (DEFAULT_CENTER / DEFAULT_FS) * k)
    with pytest.raises(FileNotFoundError):
        processor.processData(None, None, '')
    with pytest.raises(AttributeError):
        processor.processData(None, None, None)
    with pytest.raises(TypeError):
        DataProcessor()
    DataProcessor.processData(None)


def test_processor_is_spawn_safe_and_reports_json():
    p = DspProcessor(1024000, center=15000, omegaOut=5000, dec=64, fileInfo={'bitsPerSample': 'int16'})
    p.selectOutputFm()
    q = pickle.loads(pickle.dumps(p))          # what multiprocessing 'spawn' does (sdrterm.py:182-188)
    assert q.fs == 1024000 and q.decimation == 64 and q.demod == dem.fmDemod and q._engine is None
    d = json.loads(repr(p))
    assert d['fs'] == 1024000 and d['decimatedFs'] == 16000 and d['encoding'] == 'int16'
    assert d['centerFreq'] == 15000 and d['omegaOut'] == 5000 and d['bandwidth'] == 12500
    assert str(p) == 'DspProcessor'


def test_vfo_processor_rows_and_host_parsing():
    with pytest.raises(ValueError):
        VfoProcessor(1024000)
    with pytest.raises(ValueError):
        VfoProcessor(1024000, vfos='')
    v = VfoProcessor(1024000, vfoHost='localhost:12345', vfos='-200000,50000', center=15700, omegaOut=5000, dec=64,
                     fileInfo={'bitsPerSample': 'int16'})
    assert (v.host, v.port) == ('localhost', 12345)
    assert list(v.vfos) == [-200000 + 15700, 50000 + 15700, 15700]      # csv rows, then center LAST
    assert v._nFreq == 3 and v._rowFreqs() == [-184300.0, 65700.0, 15700.0] and v._applyShift()
    auto = VfoProcessor(1024000, vfoHost='localhost', vfos='0')
    assert isinstance(auto.port, int) and auto.port > 0
    d = json.loads(repr(v))
    assert d['vfos'] == '-200000,50000,0' and d['host'] == 'localhost' and d['port'] == 12345
    w = pickle.loads(pickle.dumps(v))
    assert w._fanout is None and list(w.vfos) == list(v.vfos)
    assert str(v) == 'VfoProcessor'


def test_unsupported_selections_fail_like_the_reference():
    p = DspProcessor(1024000, omegaOut=5000, dec=64)
    p.selectOutputReal()
    assert p.bandwidth == p.decimatedFs
    with pytest.raises(ValueError, match='sos array must be 2D'):       # scipy's message for sosfilt([])
        p._getEngine(32768, 'd')
    p.selectOutputFm()
    p.smooth = 5
    assert p._smoothing() == 5                                          # --smooth-output reaches the engine as its window
    p.smooth = False
    assert p._smoothing() == 0


def test_shared_library_exports_every_declared_symbol():
    """include/sdrterm_hip.h is the contract: every function it declares must be exported (no compute calls here)."""
    hdr = open(os.path.join(ROOT, 'include', 'sdrterm_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(sdrt_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) >= 25
    lib = ctypes.CDLL(_native.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f'{n} declared in the header but not exported'
    assert set(_native.EXPORTS) <= names
    assert _native.lib().sdrt_abi_version() == _native.ABI_VERSION == 7
    assert ctypes.sizeof(_native.Config) == _native.lib().sdrt_config_size()


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    if _native.device_count() > 0:
        pytest.skip('a GPU is present')
    from sdrterm_amd.engine import Engine
    with pytest.raises(_native.SdrtError) as e:
        Engine(1024000, 64, 'fm', 'h', row_freqs=[0], omega_out=5000)
    assert e.value.code == _native.ENODEVICE


def test_config_validation_messages():
    """Argument checks of sdrt_create that do not need a device."""
    from sdrterm_amd.engine import Engine
    with pytest.raises(_native.SdrtError, match='Decimation must be at least 2'):
        Engine(1024000, 1, 'fm', 'h', omega_out=5000)
    with pytest.raises(_native.SdrtError, match='powers of two'):
        Engine(1024000, 48, 'fm', 'h', omega_out=5000)
    with pytest.raises(ValueError):
        Engine(1024000, 64, 'usb', 'h', omega_out=5000)


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from sdrterm_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rows = [float(100 * i) for i in range(9)]
    mine, (lo, hi) = parallel.shard_rows(rows, world, rank)
    c0, c1 = parallel.shard_chunks(37, world, rank)
    # only rank 0 holds the IQ block; the channel hands every rank the same (chunks, eof, bytes)
    ch = parallel.BlockChannel(TorchTransport(), chunk_bytes=2048)
    src = torch.arange(4096, dtype=torch.int16).numpy().tobytes() if rank == 0 else None
    blk = ch.exchange(src, 4, False)
    end = ch.exchange(b'' if rank == 0 else None, 0, True)
    assert (blk.n, blk.eof, end.n, end.eof) == (4, False, 0, True)
    block = torch.frombuffer(bytearray(blk.host), dtype=torch.int16)
    gathered = [None] * world
    dist.all_gather_object(gathered, {'rows': (lo, hi), 'chunks': (c0, c1), 'sum': int(block.sum())})
    slow = max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((gathered, slow))
    dist.destroy_process_group()


def test_two_rank_partition_and_broadcast_under_gloo():
    import torch.multiprocessing as mp
    from sdrterm_amd.misc.general_util import findPort
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world, port = 2, findPort('127.0.0.1')
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, slow = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g['rows'] for g in gathered] == [(0, 5), (5, 9)]
    assert [g['chunks'] for g in gathered] == [(0, 19), (19, 37)]
    assert gathered[0]['sum'] == gathered[1]['sum'] == int(np.arange(4096).sum())
    assert slow == 2.0


def _iq_pair(z, fs):
    """(a, b) of a sample range for the offset recursion of read_file.py:64-77, from the oracle's restatement."""
    from oracle import ref_chain
    lam = 50 / fs
    _z, b = ref_chain.correct_iq_fast(z, 0j, fs)
    return float(np.exp(z.size * np.log1p(-lam))), complex(b)


def _gloo_iq_worker(rank, world, port, q):
    import torch.distributed as dist
    from sdrterm_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    fs, n, chunks = 48000.0, 512, 11
    rng = np.random.default_rng(12)
    z = (rng.normal(size=n * chunks) + 1j * rng.normal(size=n * chunks)) * 100 + (37 - 12j)
    c0, c1 = parallel.shard_chunks(chunks, world, rank)
    a, b = _iq_pair(z[c0 * n:c1 * n], fs)
    start, end = parallel.chain_iq_offsets(TorchTransport(), a, b, off0=3 + 4j)
    q.put((rank, (c0, c1), start, end))
    dist.destroy_process_group()


def test_chunk_sharded_iq_offsets_under_gloo():
    """A stream split by chunk over two ranks with --correct-iq: each rank's starting offset, from one all-gather of the
    ranges' (a, b) pairs, is the offset the sequential recursion (oracle: read_file.py:64-77) has at that chunk."""
    import torch.multiprocessing as mp
    from oracle import ref_chain
    from sdrterm_amd.misc.general_util import findPort
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world, port = 2, findPort('127.0.0.1')
    procs = [ctx.Process(target=_gloo_iq_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    fs, n, chunks = 48000.0, 512, 11
    rng = np.random.default_rng(12)
    z = (rng.normal(size=n * chunks) + 1j * rng.normal(size=n * chunks)) * 100 + (37 - 12j)
    assert [g[1] for g in got] == [(0, 6), (6, 11)]
    _c, at_split = ref_chain.correct_iq(z[:6 * n], 3 + 4j, fs)
    _c, at_end = ref_chain.correct_iq(z[6 * n:], at_split, fs)
    assert got[0][2] == 3 + 4j
    assert abs(got[1][2] - at_split) <= 1e-12 * abs(at_split)
    assert all(abs(g[3] - at_end) <= 1e-12 * abs(at_end) for g in got)


def test_fold_iq_pairs():
    from sdrterm_amd.parallel import fold_iq_pairs
    pairs = [(0.5, 1 + 1j), (0.25, 2j), (0.1, -1)]
    assert fold_iq_pairs(pairs, 0, 8j) == (8j, 0.1 * (0.25 * (0.5 * 8j + 1 + 1j) + 2j) - 1)
    assert fold_iq_pairs(pairs, 2, 8j)[0] == 0.25 * (0.5 * 8j + 1 + 1j) + 2j


def test_split_range_covers_everything():
    from sdrterm_amd.parallel import split_range
    for total in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [split_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_savgol_operator_reproduces_scipy():
    """The operator handed to sdrt_set_smoothing is scipy's savgol_filter(z, w, 3) for odd and even windows."""
    from scipy.signal import savgol_filter
    from sdrterm_amd.engine import savgol_operator
    rng = np.random.default_rng(3)
    for w, m in ((5, 64), (11, 512), (8, 1024), (51, 512), (4, 16)):
        taps, first, n_edge, left, right = savgol_operator(w)
        z = rng.normal(size=m)
        out = np.zeros(m)
        for i in range(m):
            if i < n_edge:
                out[i] = left[i] @ z[:w]
            elif i >= m - n_edge:
                out[i] = right[i - (m - n_edge)] @ z[m - w:]
            else:
                for k in range(taps.size):
                    j = i + first + k
                    if 0 <= j < m:
                        out[i] += taps[k] * z[j]
        assert np.abs(out - savgol_filter(z, w, 3)).max() < 1e-12


def test_vfo_processor_row_modes_extension():
    """rowModes (build extension; the reference's demodulation is processor-wide, io_args.py:118): one of fm / am per row in the
    reference's row order (csv entries, then the center), as a list or a csv string; a rank's share follows its rows."""
    p = VfoProcessor(DEFAULT_FS, vfoHost='127.0.0.1:1', vfos='100,200,300', dec=4, omegaOut=250, rowModes='FM, am,fm,am')
    assert p.rowModes == ['fm', 'am', 'fm', 'am'] and p._rowModes() == ['fm', 'am', 'fm', 'am']
    p._enterRank(1, 2)
    assert p._rowModes() == ['fm', 'am']
    assert VfoProcessor(DEFAULT_FS, vfoHost='127.0.0.1:1', vfos='100', dec=4, omegaOut=250)._rowModes() is None
    for bad in ('fm', ['fm', 'am', 'fm'], 'fm,am,re,im'):
        with pytest.raises(ValueError):
            VfoProcessor(DEFAULT_FS, vfoHost='127.0.0.1:1', vfos='100,200,300', dec=4, omegaOut=250, rowModes=bad)
    q = pickle.loads(pickle.dumps(p))
    assert q.rowModes == p.rowModes


def test_bench_launcher_dry_run():
    """bench.py --gpus N started without a launcher spawns the ranks itself; --dry-run prints what it would start: one rank per
    GPU, the variables torch.distributed.run would export, loopback rendezvous, dmabuf IPC, rank 0's output relayed."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '7', '--workload', 'simo64', '--feed',
                          'allgather', '--dry-run'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ranks = [json.loads(line) for line in out.stdout.strip().splitlines()]
    assert [r['rank'] for r in ranks] == [0, 1, 2, 3]
    ports = {r['env']['MASTER_PORT'] for r in ranks}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r in ranks:
        e = r['env']
        assert (e['RANK'], e['LOCAL_RANK'], e['WORLD_SIZE']) == (str(r['rank']), str(r['rank']), '4')
        assert e['MASTER_ADDR'] == '127.0.0.1' and e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
        assert r['cmd'][1].endswith('bench.py') and '--dry-run' not in r['cmd'] and r['cmd'][2:] == ['--gpus', '4', '--steps', '7', '--workload', 'simo64', '--feed', 'allgather']
        assert r['stdout'] == ('relayed' if r['rank'] == 0 else 'discarded')
    sys.path.insert(0, ROOT)
    import bench
    envs = bench.rank_environments(2, 12345, base={'HSA_ENABLE_IPC_MODE_LEGACY': '1', 'X': 'y'})
    assert envs[1]['RANK'] == '1' and envs[1]['X'] == 'y' and envs[1]['HSA_ENABLE_IPC_MODE_LEGACY'] == '1'
