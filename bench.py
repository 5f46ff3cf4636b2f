#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sdrterm DSP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 256 MS/s synthetic int16 IQ, FM demodulation, decimation 64, one VFO,
processed in the reference's 131072-byte chunks (32768 complex samples).  One "step" is one pass of the whole
hot path (unpack -> NCO -> zero-phase decimator -> FM discriminator -> resample -> output lowpass) over a batch
of chunks that is already resident in HBM.  With N > 1 each rank owns its own batch (chunks are independent,
SURVEY.md section 8e): weak scaling, no data-path collective.

torch is used here only as plumbing (device buffers, synthetic data, torch.distributed barrier / max); the
product path is the C ABI of libsdrterm_hip.so called through ctypes.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 256_000_000
Q = 64
CENTER = 15_000
OMEGA_OUT = 12_500
N = 32768                      # complex samples per chunk for int16 (131072 B)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # MI355X vector/matrix fp64 peak (spec)
ALG_BYTES_PER_SAMPLE = 4.0 + 8.0 / Q      # int16 IQ in + float64 out every q samples
FP64_FLOP_PER_SAMPLE = 64.0    # stage A: 16 real rows x 2 streams x 2 flop per input sample


def synth_iq_torch(n_samples, seed, device):
    """FM tone at +CENTER with 1 kHz modulation, amplitude 8000, Gaussian noise sigma 300, int16."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_samples, 2), dtype=torch.int16, device=device)
    step = 1 << 24
    fm, dev = 1000.0, 2500.0
    for s in range(0, n_samples, step):
        e = min(n_samples, s + step)
        t = torch.arange(s, e, device=device, dtype=torch.float64)
        ph = 2 * math.pi * torch.remainder(t * (CENTER / FS), 1.0) + (dev / fm) * torch.sin(
            2 * math.pi * torch.remainder(t * (fm / FS), 1.0))
        re = 8000.0 * torch.cos(ph) + 300.0 * torch.randn(e - s, device=device, generator=g, dtype=torch.float64)
        im = 8000.0 * torch.sin(ph) + 300.0 * torch.randn(e - s, device=device, generator=g, dtype=torch.float64)
        out[s:e, 0] = torch.round(re).clamp_(-32768, 32767).to(torch.int16)
        out[s:e, 1] = torch.round(im).clamp_(-32768, 32767).to(torch.int16)
    return out


def cpu_baseline(raw_bytes, budget_s=12.0):
    """The plain-C oracle (oracle/chain.c) timed on this host's cores on a bounded sample of the same workload."""
    import numpy as np
    from oracle import c_chain, ref_chain
    threads = min(c_chain.max_threads(), os.cpu_count() or 1)
    shift = c_chain.shift_table([CENTER], FS, N)
    dec, outs = ref_chain.cheby_sos(Q), ref_chain.ellip_sos(FS // Q, OMEGA_OUT)
    chunk_b = N * 4
    avail = len(raw_bytes) // chunk_b
    want = max(threads, (avail // threads) * threads)
    # warm (thread pool, page faults), then a timed probe pass to size the bounded sample to ~budget_s of CPU work
    c_chain.process_stream(raw_bytes[:want * chunk_b], 'h', N, Q, 'fm', shift, dec, outs, threads=threads)
    t0 = time.perf_counter()
    c_chain.process_stream(raw_bytes[:want * chunk_b], 'h', N, Q, 'fm', shift, dec, outs, threads=threads)
    dt1 = max(time.perf_counter() - t0, 1e-6)
    reps = max(1, min(200, int(round(budget_s / dt1))))
    t0 = time.perf_counter()
    for _ in range(reps):
        c_chain.process_stream(raw_bytes[:want * chunk_b], 'h', N, Q, 'fm', shift, dec, outs, threads=threads)
    dt = time.perf_counter() - t0
    return {'value': round(reps * want * N / dt / 1e6, 3), 'unit': 'MSamples/s', 'cores': threads, 'kind': 'port',
            'sample': f'{reps} pass(es) over {want} chunks x {N} samples of the same int16 FM stream, oracle/chain.c '
                      f'(fp64 C port of the reference chain), {threads} OpenMP threads, {dt:.1f} s'}


def measured_traffic(kernel_name, n_chunks):
    """HBM bytes per launch of the chain kernel from the rocprofv3 PMC passes (tools/pmc.sh; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE), if a measurement of this workload is committed."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if rec.get('chunks_per_step_per_gpu') != n_chunks or rec.get('kernel') not in kernel_name:
        return None
    return rec.get('hbm_bytes_per_launch')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--chunks', type=int, default=32768, help='chunks per step per GPU (32768 = 2^30 samples)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from sdrterm_amd.engine import Engine
    eng = Engine(FS, Q, 'fm', 'h', row_freqs=[CENTER], omega_out=OMEGA_OUT, device=local_rank)
    n_chunks = args.chunks
    n_samples = n_chunks * N
    raw = synth_iq_torch(n_samples, seed=1234 + rank, device=device)
    out = torch.empty((1, n_chunks * eng.m), dtype=torch.float64, device=device)
    torch.cuda.synchronize()

    def step():
        eng.process_device(raw.data_ptr(), n_chunks, out.data_ptr(), 0)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.profile(True)
    eng.profile_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    k_ms, k_n = eng.kernel_time(0)
    p_ms, p_n = eng.kernel_time(1)
    eng.profile(False)
    assert bool(torch.isfinite(out).all()), 'non-finite output'

    KERNEL_OF = {'i8w': 'chain_i8w_kernel', 'i8t': 'chain_i8t_kernel', 'i8x32': 'chain_i8_kernel', 'f64': 'chain_kernel'}
    if rank == 0:
        total_samples = float(n_samples) * args.steps * world
        value = total_samples / elapsed / 1e6
        k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
        achieved = ALG_BYTES_PER_SAMPLE * n_samples / k_avg_s / 1e9
        line = {
            'metric': 'input MSamples/s IQ->FM demod, dec=64',
            'value': round(value, 1), 'unit': 'MSamples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'configs[1]: {FS // 1_000_000} MS/s synthetic int16 IQ (FM tone + noise), FM demod, '
                                   f'dec={Q}, single VFO, center={CENTER} Hz, {n_chunks} chunks x {N} samples per GPU per step, '
                                   'inputs resident in HBM',
                       'chunks_per_step_per_gpu': n_chunks, 'samples_per_chunk': N, 'parallelism': f'chunk-sharded x{world}'},
            'roofline': {'bound': 'hbm', 'kernel': KERNEL_OF[eng.stage_a], 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': measured_traffic(KERNEL_OF[eng.stage_a], n_chunks),
                         'avg_kernel_ms': round(k_avg_s * 1e3, 4), 'launches': k_n,
                         'alg_bytes_per_sample': ALG_BYTES_PER_SAMPLE,
                         'fp64_frac': round(FP64_FLOP_PER_SAMPLE * n_samples / k_avg_s / 1e12 / FP64_PEAK_TFLOPS, 4)},
            'kernels_ms': {'chain_kernel': round(k_ms / max(k_n, 1), 4), 'post_kernel': round(p_ms / max(p_n, 1), 4)},
            'stage_a': eng.stage_a,
        }
        if not args.no_cpu_baseline and world == 1:
            sample = raw[:min(n_samples, 4096 * N)].cpu().numpy().tobytes()
            line['cpu_baseline'] = cpu_baseline(sample)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
