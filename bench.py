#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sdrterm DSP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload fm64|fm64f|simo64|f32x256]

fm64 (default; BASELINE.json configs[1]): 256 MS/s synthetic int16 IQ, FM demodulation, decimation 64, one VFO, processed in
the reference's 131072-byte chunks (32768 complex samples).  One "step" is one pass of the whole hot path (unpack -> NCO
-> zero-phase decimator -> FM discriminator -> resample -> output lowpass) over a batch of chunks that is already resident
in HBM.  With N > 1 each rank owns its own batch (chunks are independent): weak scaling, no
data-path collective.

simo64 (configs[3]): 64 VFO rows, decimation 32, FM; the rows are sharded contiguously over the ranks and every step's raw
block reaches every rank over RCCL (the only collective), block k+1 beside the kernels of block k: --feed bcast (the stream
lives on rank 0, where the reader's Queue ends: sdrt_comm_bcast) or --feed allgather (every rank holds 1/N of each block, as
when every GPU is fed its share over its own host link: sdrt_comm_allgather, each xGMI link then carries 1/N of the block
instead of the root's links carrying all of it).  Total work is fixed: strong scaling; value = input samples per second.

Launch: under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this file is one rank.
Started plainly with --gpus N > 1 it starts the N ranks itself as child processes -- before anything touches a GPU -- and
relays rank 0's line (--dry-run prints what it would start).  The ranks talk through ONE RCCL communicator behind the C ABI
(sdrt_comm_*: barrier, maximum over ranks and the block exchange; the 128-byte id travels over TCP at MASTER_ADDR); there
is no torch.distributed process group.

Timing: W warm-up steps, then back-to-back steps until --settle-seconds have passed (the chip lowers its clock under this
load; a cold start would flatter the number), then EXACTLY K timed steps between barriers, maximum over ranks; after that
a --long-seconds run without per-kernel events gives the steady-state figure next to it ("steady_state").

torch is plumbing here (device buffers, synthetic data); the product path is the C ABI of libsdrterm_hip.so called through
ctypes.  Every number on the line is produced by the run that prints it, except what sits under "from_profile" (hardware
counters need rocprofv3 passes of their own: tools/pmc.sh; the committed record of the same workload is quoted with its file).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 256_000_000
CENTER = 15_000
OMEGA_OUT = 12_500
N = 32768                      # complex samples per chunk for int16 (131072 B)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
I8_PEAK_POPS = 5.0             # dense int8 matrix peak, Pop/s (2 x bf16 per clock: MI355X_MICROARCH.md, Matrix cores)
FP64_PEAK_TFLOPS = 78.6

WORKLOADS = {
    # name: (decimation, total rows, default chunks per step, scaling, encoding, sample rate)
    'fm64': (64, 1, 131072, 'weak', 'h', 256_000_000),
    'fm64f': (64, 1, 131072, 'weak', 'f', 256_000_000),          # the same stream as float32 records (16384-sample chunks)
    'fm64b': (64, 1, 65536, 'weak', 'B', 256_000_000),           # the same stream as uint8 records (rtl_tcp's format; 65536-sample chunks)
    'simo64': (32, 64, 512, 'strong', 'h', 256_000_000),
    'f32x256': (64, 256, 128, 'strong', 'f', 1_000_000_000),
}
ITEM = {'h': 2, 'f': 4, 'B': 1}
TORCH_DTYPE = {'h': 'int16', 'f': 'float32', 'B': 'uint8'}
ENC_NAME = {'h': 'int16', 'f': 'float32', 'B': 'uint8'}


def simo_rows(n_rows):
    """Row frequencies as VfoProcessor lays them out (vfo_processor.py:43-47): csv offsets + center, then center last."""
    step = 3_000_000
    offs = [(i - (n_rows - 1) // 2) * step for i in range(n_rows - 1)]
    offs = [o if o < 0 else o + step for o in offs]                       # none of the csv rows sits on the center itself
    return [float(o + CENTER) for o in offs] + [float(CENTER)]


def synth_iq_torch(n_samples, seed, device, code='h', fs=FS):
    """FM tone at +CENTER with 1 kHz modulation, amplitude 8000, Gaussian noise sigma 300, int16; float32: the
    same stream scaled by 2^-15, not rounded; uint8: scaled by 2^-7 (tone amplitude 62.5, noise sigma 2.3) around 128."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    FS = fs
    out = torch.empty((n_samples, 2), dtype=getattr(torch, TORCH_DTYPE[code]), device=device)
    step = 1 << 24
    fm, dev = 1000.0, 2500.0
    for s in range(0, n_samples, step):
        e = min(n_samples, s + step)
        t = torch.arange(s, e, device=device, dtype=torch.float64)
        ph = 2 * math.pi * torch.remainder(t * (CENTER / FS), 1.0) + (dev / fm) * torch.sin(
            2 * math.pi * torch.remainder(t * (fm / FS), 1.0))
        re = 8000.0 * torch.cos(ph) + 300.0 * torch.randn(e - s, device=device, generator=g, dtype=torch.float64)
        im = 8000.0 * torch.sin(ph) + 300.0 * torch.randn(e - s, device=device, generator=g, dtype=torch.float64)
        if code == 'h':
            out[s:e, 0] = torch.round(re).clamp_(-32768, 32767).to(torch.int16)
            out[s:e, 1] = torch.round(im).clamp_(-32768, 32767).to(torch.int16)
        elif code == 'B':
            out[s:e, 0] = torch.round(re / 128.0 + 128.0).clamp_(0, 255).to(torch.uint8)
            out[s:e, 1] = torch.round(im / 128.0 + 128.0).clamp_(0, 255).to(torch.uint8)
        else:
            out[s:e, 0] = (re / 32768.0).to(torch.float32)
            out[s:e, 1] = (im / 32768.0).to(torch.float32)
    return out


def cpu_baseline(raw_bytes, q, freqs, budget_s=10.0, code='h', N=N, FS=FS):
    """The plain-C oracle (oracle/chain.c) timed on this host's cores on a bounded sample of the same workload: all cores
    (OpenMP over chunks) and one core."""
    from oracle import c_chain, ref_chain
    shift = c_chain.shift_table(freqs, FS, N)
    dec, outs = ref_chain.cheby_sos(q), ref_chain.ellip_sos(FS // q, OMEGA_OUT)
    chunk_b = N * 2 * ITEM[code]
    avail = len(raw_bytes) // chunk_b

    def timed(threads, budget):
        want = max(threads, min(avail, (avail // threads) * threads))
        if threads == 1:
            want = min(avail, 8)
        run = lambda: c_chain.process_stream(raw_bytes[:want * chunk_b], code, N, q, 'fm', shift, dec, outs, threads=threads)
        run()                                                   # warm (thread pool, page faults)
        t0 = time.perf_counter()
        run()
        dt1 = max(time.perf_counter() - t0, 1e-6)
        reps = max(1, min(200, int(round(budget / dt1))))
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        dt = time.perf_counter() - t0
        return reps * want * N / dt / 1e6, reps, want, dt

    threads = min(c_chain.max_threads(), os.cpu_count() or 1)
    v_all, reps, want, dt = timed(threads, budget_s)
    v_one, reps1, want1, dt1 = timed(1, budget_s / 2)
    return {'value': round(v_all, 3), 'unit': 'MSamples/s', 'cores': threads, 'kind': 'port',
            'sample': f'{reps} pass(es) over {want} chunks x {N} samples of the same stream ({len(freqs)} row(s), dec={q}), '
                      f'oracle/chain.c (fp64 C port of the reference chain), {threads} OpenMP threads, {dt:.1f} s',
            'single_core': {'value': round(v_one, 3), 'unit': 'MSamples/s', 'cores': 1,
                            'sample': f'{reps1} pass(es) over {want1} chunks, 1 thread, {dt1:.1f} s'}}


def committed_record(name):
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', name)))
    except (OSError, ValueError):
        return None


def rank_environments(n_gpus, port, base=None):
    """What each rank of a self-launched run finds in its environment (the variables torch.distributed.run would set)."""
    base = dict(os.environ if base is None else base)
    return [dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                 HSA_ENABLE_IPC_MODE_LEGACY=base.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')) for r in range(n_gpus)]


def launch_ranks(args, argv, dry_run=False):
    """--gpus N without a launcher: start the N ranks as fresh child processes (nothing in this process has touched a GPU).
    dry_run: print one JSON line per rank (its command and the variables above) instead."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != '--dry-run']
    envs = rank_environments(args.gpus, port)
    if dry_run:
        keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')
        for r, env in enumerate(envs):
            print(json.dumps({'rank': r, 'cmd': cmd, 'env': {k: env[k] for k in keys}, 'stdout': 'relayed' if r == 0 else 'discarded'}))
        return 0
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL) for r, env in enumerate(envs)]
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = max(rc, p.wait())
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def main():
    global FS, N
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='fm64')
    ap.add_argument('--correct-iq', action='store_true', help='the workload with --correct-iq (single-VFO workloads)')
    ap.add_argument('--chunks', type=int, default=0, help='chunks per step (fm64: per GPU, default 131072 = 2^32 samples = 16 s of the 256 MS/s stream; simo64: 512)')
    ap.add_argument('--settle-seconds', type=float, default=2.0, help='untimed back-to-back steps before the timed region')
    ap.add_argument('--long-seconds', type=float, default=2.0, help='steady-state run after the timed region (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='diagnostic: no per-kernel HIP events in the timed region')
    ap.add_argument('--feed', choices=('bcast', 'scatter', 'allgather'), default='bcast', help='many-row workloads: how a block reaches the ranks (bcast: ncclBroadcast from rank 0; scatter: from rank 0 as 1/N shards + all-gather; allgather: every rank holds 1/N)')
    ap.add_argument('--no-same-arithmetic', action='store_true', help="skip the fp64 kernel's side figure")
    ap.add_argument('--force-comm', action='store_true', help='diagnostic: the RCCL communicator with a single rank too (barrier / maximum / block exchange path on one GPU)')
    ap.add_argument('--dry-run', action='store_true', help='with --gpus N > 1 and no launcher: print the ranks that would be started')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:], dry_run=args.dry_run))

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    from sdrterm_amd import _native as nat, parallel
    from sdrterm_amd.engine import Engine
    local_dev = parallel.device_for(local_rank)       # = LOCAL_RANK unless SDRTERM_DEVICE_MAP says otherwise
    torch.cuda.set_device(local_dev)
    device = torch.device('cuda', local_dev)

    # one communicator for everything the ranks say to each other (barrier, maximum, block exchange): RCCL behind the C ABI
    comm = None
    if world > 1 or args.force_comm:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        comm = parallel.RcclTransport(rank, world, local_dev)        # the id travels over TCP at MASTER_ADDR:MASTER_PORT+17
    q, rows_total, default_chunks, scaling, code, FS = WORKLOADS[args.workload]
    N = 131072 // (2 * ITEM[code])            # complex samples per 131072-byte chunk
    n_chunks = args.chunks or default_chunks
    n_samples = n_chunks * N
    single = rows_total == 1                  # one VFO: chunks sharded over the ranks; else rows sharded, blocks broadcast
    if single:
        freqs_all = [float(CENTER)]
        freqs, (lo, hi) = freqs_all, (0, 1)
        raw = synth_iq_torch(n_samples, seed=1234 + rank, device=device, code=code, fs=FS)
        bufs = [raw]
        src = None
    else:
        freqs_all = simo_rows(rows_total)
        freqs, (lo, hi) = parallel.shard_rows(freqs_all, world, rank)
        # two block buffers per rank: block k+1 arrives beside the kernels of block k.  bcast: the stream lives on rank 0 (where
        # the reader's Queue ends).  allgather: every rank holds the same 1/N of every block (the whole stream is synthesised
        # everywhere from one seed, a rank keeps its slice)
        allg = args.feed == 'allgather' and comm is not None
        if n_samples % world and allg:
            raise SystemExit('--feed allgather needs a block that divides by the number of ranks')
        src = synth_iq_torch(2 * n_samples, seed=1234, device=device, code=code, fs=FS) if (rank == 0 or allg) else None
        if allg:
            per = n_samples // world
            src = torch.stack([src[b * n_samples + rank * per:b * n_samples + (rank + 1) * per] for b in range(2)]).contiguous()
        bufs = [torch.empty((n_samples, 2), dtype=getattr(torch, TORCH_DTYPE[code]), device=device) for _ in range(2)]
    eng = Engine(FS, q, "fm", code, row_freqs=freqs, apply_shift=True, omega_out=OMEGA_OUT, device=local_dev, correct_iq=args.correct_iq,
                 reserve_chunks=n_chunks)         # no allocation inside the timed calls
    rows = len(freqs)
    out = torch.empty((rows, n_chunks * eng.m), dtype=torch.float64, device=device)
    lib = nat.lib()
    # everything is submitted to one non-blocking stream of this process (not the legacy default stream, which orders itself against
    # every other stream)
    st = torch.cuda.Stream(device=device)
    S = st.cuda_stream
    s_comm = None
    if comm is not None and not single:
        p = nat.C.c_void_p()
        nat.check(lib.sdrt_stream_create(nat.C.byref(p)))
        s_comm = p.value
    torch.cuda.synchronize()

    block_bytes = n_samples * 2 * ITEM[code]
    state = {'k': 0}

    def fetch(k):
        """Block k of the stream into buffer k % 2 on every rank (many-row workloads)."""
        dst = bufs[k % 2]
        if s_comm is not None and args.feed == 'allgather':
            nat.check(lib.sdrt_stream_wait_stream(s_comm, S))        # after the kernels still reading dst
            comm.allgather_device(src[k % 2].data_ptr(), dst.data_ptr(), block_bytes // world, s_comm)
            return
        if rank == 0:
            with torch.cuda.stream(st):
                dst.copy_(src[(k % 2) * n_samples:(k % 2 + 1) * n_samples], non_blocking=True)
        if s_comm is not None:
            nat.check(lib.sdrt_stream_wait_stream(s_comm, S))        # after the copy above / the kernels still reading dst
            comm.bcast_device(dst.data_ptr(), block_bytes, 0, s_comm, scatter=args.feed == 'scatter')

    run = eng.process_device

    def step():
        k = state['k']
        if not single:
            fetch(k + 1)                                               # beside this step's kernels (other buffer)
        run(bufs[k % len(bufs)].data_ptr(), n_chunks, out.data_ptr(), S)
        if s_comm is not None:
            nat.check(lib.sdrt_stream_wait_stream(S, s_comm))         # the next step's kernels need that block
        state['k'] = k + 1

    def over_ranks(value):
        """Every rank's value (an all-gather of 8 bytes per rank through the communicator: also the barrier)."""
        if comm is None:
            return [float(value)]
        import struct
        return [struct.unpack('<d', b)[0] for b in comm.allgather_bytes(struct.pack('<d', float(value)))]

    def barrier():
        over_ranks(0.0)

    if not single:
        fetch(0)
        if s_comm is not None:
            nat.check(lib.sdrt_stream_wait_stream(S, s_comm))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t_settle = time.perf_counter()
    settle_steps = 0
    # (the ranks decide together when the settling is over: with a collective in every step a rank that left the loop one round
    # before the others would leave them waiting in it)
    while max(over_ranks(time.perf_counter() - t_settle)) < args.settle_seconds:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
        settle_steps += 16

    # ---- the timed region: exactly K steps ---------------------------------------------------------------------------
    eng.profile(not args.no_kernel_events)
    eng.profile_reset()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # recorded on the submission stream
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record(st)
    for i in range(args.steps):
        step()
        evs[i + 1].record(st)
    torch.cuda.synchronize()
    barrier()
    elapsed = max(over_ranks(time.perf_counter() - t0))
    k_ms, k_n = eng.kernel_time(0)
    p_ms, p_n = eng.kernel_time(1)
    eng.profile(False)
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    assert bool(torch.isfinite(out).all()), 'non-finite output'

    # ---- steady state: a longer run, no per-kernel events ---------------------------------------------------------------
    steady = None
    if args.long_seconds > 0:
        per = max(elapsed / args.steps, 1e-6)
        n_long = max(16, int(args.long_seconds / per))
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_long):
            step()
        torch.cuda.synchronize()
        barrier()
        dt_long = max(over_ranks(time.perf_counter() - t1))
        steady = {'steps': n_long, 'seconds': round(dt_long, 3), 'ms_per_step': round(dt_long / n_long * 1e3, 4)}

    KERNEL_OF = {'i8w': 'chain_i8w_kernel', 'i8t': 'chain_i8t_kernel', 'f64': 'chain_kernel'}
    if rank == 0:
        per_gpu = 1 if scaling == 'strong' else world
        total_samples = float(n_samples) * args.steps * per_gpu
        value = total_samples / elapsed / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        # algorithmic bytes per input sample on ONE GPU: 4 B of int16 IQ in + 8 B per row and q samples out
        alg_bps = 2.0 * ITEM[code] + 8.0 * rows / q
        k_avg_ms = k_ms / max(k_n, 1) if k_n else None
        p_avg_ms = p_ms / max(p_n, 1) if p_n else None
        # the chain kernel's own algorithmic bytes: input + the decimated complex128 samples it hands to the post kernel
        achieved = alg_bps * n_samples / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms else None
        path_achieved = alg_bps * n_samples / (ms_per_step * 1e-3) / 1e9
        # int8 matrix work of the dominant kernel: matrix instructions per 4096-sample tile x 16*16*64*2 operations
        mfma_per_tile = ({'h': {64: 184, 32: 216}, 'f': {64: 288, 32: 336}, 'B': {64: 104, 32: 120}}[code][q] + (16 if args.correct_iq and rows == 1 else 0)) if eng.stage_a == 'i8w' else ({'h': 160, 'f': 224}.get(code) if eng.stage_a == 'i8t' else None)
        i8_ops = mfma_per_tile * 32768.0 * rows * n_samples / 4096 if mfma_per_tile else None
        # hardware counters cannot be read from inside this run: the committed record of rocprofv3 passes over this very workload
        # (tools/pmc.sh) is quoted under "from_profile", with the file it comes from and the commit it was taken at
        pmc_file = 'r03_pmc_record.json'
        pmc = committed_record(pmc_file) if args.workload == 'fm64' and not args.correct_iq else None
        pmc_ok = bool(pmc) and pmc.get('chunks_per_step_per_gpu') == n_chunks and pmc.get('kernel', '') in KERNEL_OF[eng.stage_a]
        from_profile = {'file': 'profiles/' + pmc_file, 'commit': pmc.get('commit'), 'traffic': pmc.get('hbm_bytes_per_launch'),
                        'mfma_i8_busy': pmc.get('mfma_busy'), 'clock_mhz': pmc.get('clock_mhz'),
                        'note': 'rocprofv3 --pmc passes of this workload (tools/pmc.sh), not measured by this run'} if pmc_ok else None
        read_bps = 2.0 * ITEM[code]
        hbm = {'bound': 'hbm', 'kernel': KERNEL_OF[eng.stage_a],
               'achieved': round(achieved, 1) if achieved else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
               'frac': round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
               'traffic': from_profile['traffic'] if from_profile else None,
               'avg_kernel_ms': round(k_avg_ms, 4) if k_avg_ms else None, 'launches': k_n,
               'alg_bytes_per_sample': alg_bps,
               # the whole step (chain + post + gaps) against the same bytes: what north_star's 50 % is asked of; read_frac: the
               # input bytes alone (its "HBM-read roofline")
               'path_achieved': round(path_achieved, 1), 'path_frac': round(path_achieved / HBM_PEAK_GBS, 4),
               'read_frac': round(read_bps * n_samples / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               'from_profile': from_profile}
        i8 = {'int8_pops': round(i8_ops / (k_avg_ms * 1e-3) / 1e15, 3) if (i8_ops and k_avg_ms) else None,
              'int8_frac': round(i8_ops / (k_avg_ms * 1e-3) / 1e15 / I8_PEAK_POPS, 4) if (i8_ops and k_avg_ms) else None}
        if rows == 1 or not i8['int8_pops']:
            roof = dict(hbm, **i8)
        else:
            # many rows per input sample: the int8 matrix work, not the input stream, is what bounds the kernel
            roof = {'bound': 'mfma', 'kernel': KERNEL_OF[eng.stage_a], 'achieved': round(i8['int8_pops'] * 1e3, 1),
                    'peak': I8_PEAK_POPS * 1e3, 'unit': 'TFLOP/s', 'frac': i8['int8_frac'], 'traffic': None,
                    'avg_kernel_ms': hbm['avg_kernel_ms'], 'launches': k_n, 'ops': 'int8 multiply-adds x 2 (dense int8 matrix peak)',
                    'hbm': {k: hbm[k] for k in ('achieved', 'frac', 'alg_bytes_per_sample', 'path_achieved', 'path_frac', 'read_frac')}}
        line = {
            'metric': {'fm64': 'input MSamples/s IQ->FM demod, dec=64', 'fm64f': 'input MSamples/s IQ->FM demod, float32 IQ, dec=64', 'fm64b': 'input MSamples/s IQ->FM demod, uint8 IQ, dec=64', 'simo64': 'input MSamples/s 64-VFO --simo FM, dec=32',
                       'f32x256': 'input MSamples/s 256-VFO --simo FM, float32 IQ, dec=64'}[args.workload],
            'value': round(value, 1), 'unit': 'MSamples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None,
            # what the arithmetic is: int8 x int8 -> int32 matrix products of 5 base-256 weight digits x 2 byte slices (exact),
            # recombined and finished in fp64; 'f64' when the fp64 matrix-core kernel runs
            'dtype': ('f32->4 x i8 block fixed point, i8x5-split+f64' if code == 'f' else 'i8x5-split+f64') if eng.stage_a in ('i8w', 'i8t') else 'f64', 'data': 'synthetic',
            'config': {'workload': (f'configs[1]{"" if code == "h" else f" as {ENC_NAME[code]} records"}: {FS // 1_000_000} MS/s synthetic {ENC_NAME[code]} IQ (FM tone + noise), FM demod, dec={q}, single VFO, '
                                    f'center={CENTER} Hz{", --correct-iq" if args.correct_iq else ""}, {n_chunks} chunks x {N} samples per GPU per step, inputs resident in HBM')
                       if single else
                       (f'configs[{3 if code == "h" else 4}]: --simo {rows_total} VFO rows sharded {rows}/GPU, {FS // 1_000_000} MS/s synthetic {"int16" if code == "h" else "float32"} IQ, FM, dec={q}, '
                        f'{n_chunks} chunks x {N} samples per step ' + {'bcast': 'resident on rank 0, RCCL broadcast per step', 'scatter': 'resident on rank 0, sent as 1/N shards + RCCL all-gather per step', 'allgather': 'resident as 1/N shards per rank, RCCL all-gather per step'}[args.feed]),
                       'chunks_per_step_per_gpu': n_chunks, 'samples_per_chunk': N, 'rows_per_gpu': rows,
                       'parallelism': f'chunk-sharded x{world}' if single else f'rows-sharded x{world} + {args.feed}',
                       'settle_steps': settle_steps + args.warmup},
            'step_ms': {'median': round(step_ms[len(step_ms) // 2], 4), 'min': round(step_ms[0], 4), 'max': round(step_ms[-1], 4),
                        'timed_region_s': round(elapsed, 3)},
            'steady_state': steady,
            'roofline': roof,
            'kernels_ms': {'chain_kernel': round(k_avg_ms, 4) if k_avg_ms else None,
                           'post_kernel': round(p_avg_ms, 4) if p_avg_ms else None},
            'stage_a': eng.stage_a,
        }
        # the same workload in the reference's own arithmetic: stage A on the fp64 matrix cores (SDRTERM_STAGE_A=f64), same
        # buffers, same process, a few steps
        if not args.no_same_arithmetic and eng.stage_a != 'f64':
            os.environ['SDRTERM_STAGE_A'] = 'f64'
            try:
                e64 = Engine(FS, q, 'fm', code, row_freqs=freqs, apply_shift=True, omega_out=OMEGA_OUT, device=local_dev, correct_iq=args.correct_iq)
            finally:
                del os.environ['SDRTERM_STAGE_A']
            n64 = 4 if single else 2
            e64.process_device(bufs[0].data_ptr(), n_chunks, out.data_ptr(), S)
            torch.cuda.synchronize()
            t64 = time.perf_counter()
            for _ in range(n64):
                e64.process_device(bufs[0].data_ptr(), n_chunks, out.data_ptr(), S)
            torch.cuda.synchronize()
            ms64 = (time.perf_counter() - t64) / n64 * 1e3
            line['same_arithmetic'] = {'stage_a': e64.stage_a, 'dtype': 'f64', 'steps': n64, 'ms_per_step': round(ms64, 4),
                                       'value': round(n_samples / (ms64 * 1e-3) / 1e6, 1), 'unit': 'MSamples/s per GPU',
                                       'path_frac': round(alg_bps * n_samples / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            e64.close()
        if not args.no_cpu_baseline and world == 1:
            sample = (bufs[0] if single else src)[:min(n_samples, 2048 * N)].cpu().numpy().tobytes()
            line['cpu_baseline'] = cpu_baseline(sample, q, freqs_all, code=code, N=N, FS=FS)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if comm is not None:
        barrier()
        comm.close()


if __name__ == '__main__':
    main()
