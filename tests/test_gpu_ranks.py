"""Several rank processes on ONE GPU (all a development box has), with tests/standin_rccl in place of RCCL -- which refuses two
ranks on one device.  What runs is the real thing otherwise: bench.py's self-launched ranks and VfoProcessor's helper processes,
the unique-id exchange, communicator set-up, the block exchange on the communication stream beside the kernels, barrier and
maximum over the ranks, every rank's sequence of collectives, the HIP kernels on every rank's share.  (RCCL itself is exercised
with a world of one in tests/test_gpu_processors.py; N > 1 on N GPUs is the driver's scaling run.)"""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _env(n_ranks):
    from tests import standin_rccl
    env = dict(os.environ, SDRTERM_RCCL_LIB=standin_rccl.library(), SDRTERM_DEVICE_MAP=','.join(['0'] * n_ranks),
               SDRTERM_COMM_TIMEOUT='120', SDRTERM_COMM_INIT_TIMEOUT='120', SDRT_STANDIN_TIMEOUT='120')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def _bench(n_ranks, *flags):
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n_ranks), '--steps', '5', '--warmup', '2', '--settle-seconds', '0.4',
           '--long-seconds', '0.3', '--no-cpu-baseline', '--no-same-arithmetic', *flags]
    r = subprocess.run(cmd, env=_env(n_ranks), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints the one line
    return json.loads(lines[0])


@pytest.mark.parametrize('n_ranks,flags', [(2, ('--workload', 'simo64', '--chunks', '64')),
                                           (3, ('--workload', 'simo64', '--chunks', '48', '--feed', 'allgather')),
                                           (4, ('--workload', 'simo64', '--chunks', '64', '--feed', 'bcast')),
                                           (4, ('--workload', 'simo64', '--chunks', '64', '--feed', 'scatter')),
                                           (3, ('--workload', 'simo64', '--chunks', '64', '--feed', 'scatter')),
                                           (2, ('--workload', 'f32x256', '--chunks', '32')),
                                           (2, ('--workload', 'fm64', '--chunks', '1024'))],
                         ids=['simo64-default-2', 'simo64-allgather-3', 'simo64-bcast-4', 'simo64-scatter-4', 'simo64-scatter-3-falls-back', 'f32x256-default-2', 'fm64-chunks-2'])
def test_bench_with_several_ranks_on_one_gpu(n_ranks, flags):
    """bench.py --gpus N without a launcher: N rank processes, rows (or chunks) sharded, one exchange per step; the line rank 0
    prints counts every rank's work and the configuration says how the rows were split."""
    line = _bench(n_ranks, *flags)
    assert line['n_gpus'] == n_ranks and line['steps'] == 5 and line['value'] > 0 and line['cpu_baseline'] is None
    wl = flags[1]
    rows_total = {'simo64': 64, 'f32x256': 256, 'fm64': 1}[wl]
    if rows_total > 1:
        assert line['scaling'] == 'strong' and line['config']['rows_per_gpu'] == -(-rows_total // n_ranks)      # rank 0 takes the larger share
        # the default feed: scatter + all-gather whenever the block divides by the number of ranks (else ncclBroadcast)
        default = 'scatter' if (int(flags[3]) * 131072) % n_ranks == 0 else 'bcast'
        assert line['config']['parallelism'] == f'rows-sharded x{n_ranks} + ' + (flags[flags.index('--feed') + 1] if '--feed' in flags else default)
        samples = int(flags[3]) * line['config']['samples_per_chunk']
    else:
        assert line['scaling'] == 'weak' and line['config']['parallelism'] == f'chunk-sharded x{n_ranks}'
        samples = int(flags[3]) * line['config']['samples_per_chunk'] * n_ranks
    assert line['value'] == pytest.approx(samples / (line['ms_per_step'] * 1e-3) / 1e6, rel=1e-3)
    assert line['steady_state']['steps'] >= 16 and line['kernels_ms']['chain_kernel'] > 0


def test_five_rank_processes_rows_and_chunks_with_every_rank_checked():
    """The driver's 8-GPU run in the largest rehearsal a one-GPU box allows (at most six processes may hold the card at once, and
    the test runner is one of them): five rank processes on the stand-in transport.  simo64 (configs[3]): 64 rows split
    13 + 13 + 13 + 13 + 12, the default feed (1/5 shards from rank 0 + all-gather), the block scaled with the ranks so that a
    rank's step keeps its one-GPU duration; fm64 --correct-iq: five chunk-sharded batches.  The pre-flight of the feeds must have passed on every rank ('comm'), and EVERY
    rank compares its rows / chunks with the oracle ('parity.ranks_ok'), not rank 0 alone."""
    line = _bench(5, '--workload', 'simo64', '--chunks', '80')
    assert line['comm'] == dict(line['comm'], world=5, ranks_seen=5, bcast_ok=True, scatter_ok=True, allgather_ok=True, selftest_ok=True, feed='scatter')
    assert line['config']['rows_per_gpu'] == 13 and line['config']['parallelism'] == 'rows-sharded x5 + scatter'
    assert line['parity']['ranks_ok'] == [True] * 5 and line['parity']['max_abs_err_over_ranks'] <= 1e-6
    line = _bench(5, '--workload', 'simo64')              # no --chunks: the block grows with the ranks (512 x 5 chunks), per-GPU work fixed
    assert line['scaling'] == 'weak' and line['config']['chunks_per_step_per_gpu'] == 512 * 5 and line['parity']['ranks_ok'] == [True] * 5
    line = _bench(5, '--workload', 'fm64', '--chunks', '2048', '--correct-iq')
    assert line['comm']['selftest_ok'] and line['comm']['feed'].startswith('none')
    assert line['parity']['ranks_ok'] == [True] * 5 and line['scaling'] == 'weak'


def test_chunk_sharded_stream_with_correct_iq_over_four_rank_processes():
    """One stream, contiguous chunk ranges over four rank processes, --correct-iq: the affine fold of the carried offset through the
    communicator (sdrt_iq_range_device + an all-gather of 24 bytes per rank), every rank's range against the oracle's run over the
    whole stream (tests/rank_scenarios.py: iqshard)."""
    _scenario('iqshard', 4)


def _scenario(name, n_ranks=2, **extra):
    r = subprocess.run([sys.executable, '-m', 'tests.rank_scenarios', name], env=dict(_env(n_ranks), **extra), cwd=REPO, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'scenario ok' in r.stdout, (r.stdout[-2000:] + r.stderr[-4000:])


@pytest.mark.parametrize('name,n_ranks,bcast', [('rows', 2, 'ring'), ('rows', 2, 'scatter'), ('rows4', 4, 'scatter')])
def test_simo_rows_over_rank_processes_on_one_gpu(name, n_ranks, bcast):
    """VfoProcessor(gpus=2 / 4): rank 0 and its helper processes, three rows split 2 + 1 (five: 2 + 1 + 1 + 1), every client on the
    ONE port the reference serves (connection k = row k; the other ranks' sockets are handed to them); nine chunks in batches of
    two (several blocks in flight), every row's stream against oracle/chain.c, in order (tests/rank_scenarios.py).  The block goes
    out as 1/N shards from rank 0 that the ranks all-gather in place (sdrt_comm_bcast_scatter, the default) or as one broadcast
    (SDRTERM_BCAST=ring)."""
    _scenario(name, n_ranks, SDRTERM_BCAST=bcast)


def test_bench_under_the_distributed_launcher_on_one_gpu():
    """The driver's own launch line: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 ...: the ranks find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment and
    exchange the communicator id beside the launcher's own store on MASTER_PORT."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1',
           '--workload', 'simo64', '--chunks', '64', '--settle-seconds', '0.3', '--long-seconds', '0.2', '--no-same-arithmetic']
    r = subprocess.run(cmd, env=_env(2), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 4 and line['warmup'] == 1 and line['config']['rows_per_gpu'] == 32
    assert line['cpu_baseline'] is None                      # rank 0 at N = 1 only


def test_a_helper_rank_that_dies_takes_the_job_down_in_bounded_time():
    """a rank that fails must not leave the others in a collective.  Two ranks, a source that trickles; the helper
    process is killed after the first blocks: rank 0 comes back from processData within the deadlines, with the halt flag up
    (tests/rank_scenarios.py)."""
    _scenario('kill', SDRTERM_COMM_TIMEOUT='8', SDRT_STANDIN_TIMEOUT='8')


def test_halt_flag_with_live_helpers_leaves_in_bounded_time():
    """the halt flag raised in mid-stream while the helper waits in a collective: the helper stays for rank 0's
    end-of-stream block and exits 0, rank 0 does not tear down behind a collective nobody completes (tests/rank_scenarios.py)."""
    _scenario('halt')
