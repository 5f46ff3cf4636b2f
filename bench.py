#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sdrterm DSP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload fm64|fm64f|fm64b|simo8|simo64|f32x256]

fm64 (default; BASELINE.json configs[1]): 256 MS/s synthetic int16 IQ, FM demodulation, decimation 64, one VFO, processed in
the reference's 131072-byte chunks (32768 complex samples).  One "step" is one pass of the whole hot path (unpack -> NCO
-> zero-phase decimator -> FM discriminator -> resample -> output lowpass) over a batch of chunks that is already resident
in HBM.  With N > 1 each rank owns its own batch (chunks are independent): weak scaling, no
data-path collective.

simo8 (configs[2]): 8 VFO rows (7 csv offsets + the centre row, vfo_processor.py:43-47), decimation 32, --correct-iq, FM and AM rows
mixed in one pass (sdrt_config.row_modes).  simo64 (configs[3]): 64 VFO rows, decimation 32, FM.  f32x256 (configs[4]): 256 rows of
float32 IQ.  The rows are sharded contiguously over the ranks and every step's raw block reaches every rank over RCCL (the only
collective), block k+1 beside the kernels of block k: --feed scatter (the default with N > 1: the stream lives on rank 0, where the
reader's Queue ends; rank 0 sends 1/N of the block to each rank over all of its links at once and the ranks all-gather the shards:
sdrt_comm_bcast_scatter), --feed bcast (ncclBroadcast of the whole block) or --feed allgather (every rank holds 1/N of each block, as
when every GPU is fed its share over its own host link).  Total work is fixed: strong scaling; value = input samples per second.

Launch: under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this file is one rank.
Started plainly with --gpus N > 1 it starts the N ranks itself as child processes -- before anything touches a GPU -- and
relays rank 0's line (--dry-run prints what it would start).  The ranks talk through ONE RCCL communicator behind the C ABI
(sdrt_comm_*: barrier, maximum over ranks and the block exchange; the 128-byte id travels over TCP at MASTER_ADDR); there
is no torch.distributed process group.

Timing: W warm-up steps, then back-to-back steps until --settle-seconds have passed (the chip lowers its clock under this
load; a cold start would flatter the number), then EXACTLY K timed steps between barriers, maximum over ranks; after that
a --long-seconds run without per-kernel events gives the steady-state figure next to it ("steady_state").

No torch anywhere (round 4): device buffers, streams, events and copies are the C ABI's own (sdrt_malloc, sdrt_stream_*,
sdrt_event_*: include/sdrterm_hip.h); the stream is synthesised on the host with numpy in blocks, uploaded, and tiled on the
device to the step's size (2^26 distinct samples per GPU for the single-VFO workloads: every chunk is filtered on its own, so a
repeated chunk costs what a fresh one costs; nothing is cached between them, 256 MiB of Infinity Cache against 17 GB per step).

"parity": after the timed region, the outputs of sampled chunks of the very buffer that was timed are compared with
oracle/chain.c (the CPU restatement of the reference chain, pinned to the reference's own vectors) on the same raw bytes.
Every number on the line is produced by the run that prints it, except what sits under "from_profile" (hardware counters need
rocprofv3 passes of their own: tools/pmc.sh; the committed record of the same workload is quoted with its file).
"""
import argparse
import ctypes as C
import json
import math
import os
import struct
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
UNIQUE_SAMPLES = 1 << 26       # distinct synthetic samples per buffer; longer buffers repeat them (see the docstring)

WORKLOADS = {
    # name: (decimation, total rows, default chunks per step, scaling, encoding, sample rate, --correct-iq, per-row modes)
    'fm64': (64, 1, 131072, 'weak', 'h', 256_000_000, False, None),
    'fm64f': (64, 1, 131072, 'weak', 'f', 256_000_000, False, None),     # the same stream as float32 records (16384-sample chunks)
    'fm64b': (64, 1, 65536, 'weak', 'B', 256_000_000, False, None),      # the same stream as uint8 records (rtl_tcp's format; 65536-sample chunks)
    'fm16': (16, 1, 16384, 'weak', 'h', 256_000_000, False, None),       # decimation 2 .. 16: the one-pass kernel (kernels_sq.hip)
    'fm8': (8, 1, 8192, 'weak', 'h', 256_000_000, False, None),
    'fm4': (4, 1, 4096, 'weak', 'h', 256_000_000, False, None),
    'fm2': (2, 1, 2048, 'weak', 'h', 256_000_000, False, None),          # decimation 2, the CLI's default (src/sdrterm.py:80)
    'fm16b': (16, 1, 8192, 'weak', 'B', 256_000_000, False, None),       # the same as uint8 records (65536-sample chunks)
    'fm8b': (8, 1, 4096, 'weak', 'B', 256_000_000, False, None),
    'fm4b': (4, 1, 2048, 'weak', 'B', 256_000_000, False, None),
    'fm2b': (2, 1, 1024, 'weak', 'B', 256_000_000, False, None),
    'simo8': (32, 8, 4096, 'strong', 'h', 256_000_000, True, ('fm', 'am', 'fm', 'am', 'fm', 'am', 'am', 'fm')),
    'simo64': (32, 64, 512, 'strong', 'h', 256_000_000, False, None),
    'f32x256': (64, 256, 128, 'strong', 'f', 1_000_000_000, False, None),
}
METRIC = {'fm64': 'input MSamples/s IQ->FM demod, dec=64', 'fm64f': 'input MSamples/s IQ->FM demod, float32 IQ, dec=64',
          'fm64b': 'input MSamples/s IQ->FM demod, uint8 IQ, dec=64',
          'fm16': 'input MSamples/s IQ->FM demod, dec=16', 'fm8': 'input MSamples/s IQ->FM demod, dec=8', 'fm4': 'input MSamples/s IQ->FM demod, dec=4', 'fm2': 'input MSamples/s IQ->FM demod, dec=2',
          'fm16b': 'input MSamples/s IQ->FM demod, uint8 IQ, dec=16', 'fm2b': 'input MSamples/s IQ->FM demod, uint8 IQ, dec=2',
          'fm8b': 'input MSamples/s IQ->FM demod, uint8 IQ, dec=8', 'fm4b': 'input MSamples/s IQ->FM demod, uint8 IQ, dec=4',
          'simo8': 'input MSamples/s 8-VFO --simo FM+AM mixed, --correct-iq, dec=32',
          'simo64': 'input MSamples/s 64-VFO --simo FM, dec=32', 'f32x256': 'input MSamples/s 256-VFO --simo FM, float32 IQ, dec=64'}
CONFIG_NO = {'fm64': 1, 'fm64f': 1, 'fm64b': 1, 'fm16': 1, 'fm8': 1, 'fm4': 1, 'fm2': 1, 'fm16b': 1, 'fm8b': 1, 'fm4b': 1, 'fm2b': 1, 'simo8': 2, 'simo64': 3, 'f32x256': 4}
ITEM = {'h': 2, 'f': 4, 'B': 1}
NP_DTYPE = {'h': 'int16', 'f': 'float32', 'B': 'uint8'}
ENC_NAME = {'h': 'int16', 'f': 'float32', 'B': 'uint8'}


def simo_rows(n_rows):
    """Row frequencies as VfoProcessor lays them out (vfo_processor.py:43-47): csv offsets + center, then center last."""
    step = 3_000_000
    offs = [(i - (n_rows - 1) // 2) * step for i in range(n_rows - 1)]
    offs = [o if o < 0 else o + step for o in offs]                       # none of the csv rows sits on the center itself
    return [float(o + CENTER) for o in offs] + [float(CENTER)]


def synth_iq_host(n_samples, seed, code='h', fs=FS, start=0):
    """FM tone at +CENTER with 1 kHz modulation, amplitude 8000, Gaussian noise sigma 300, int16; float32: the
    same stream scaled by 2^-15, not rounded; uint8: scaled by 2^-7 (tone amplitude 62.5, noise sigma 2.3) around 128.
    numpy, on the host: (n_samples, 2) array of the encoding's dtype."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = np.empty((n_samples, 2), dtype=NP_DTYPE[code])
    blk = 1 << 22
    fm, dev = 1000.0, 2500.0
    for s in range(0, n_samples, blk):
        e = min(n_samples, s + blk)
        t = np.arange(start + s, start + e, dtype=np.float64)
        ph = 2 * math.pi * np.remainder(t * (CENTER / fs), 1.0) + (dev / fm) * np.sin(2 * math.pi * np.remainder(t * (fm / fs), 1.0))
        re = 8000.0 * np.cos(ph) + 300.0 * rng.standard_normal(e - s)
        im = 8000.0 * np.sin(ph) + 300.0 * rng.standard_normal(e - s)
        if code == 'h':
            out[s:e, 0] = np.clip(np.round(re), -32768, 32767)
            out[s:e, 1] = np.clip(np.round(im), -32768, 32767)
        elif code == 'B':
            out[s:e, 0] = np.clip(np.round(re / 128.0 + 128.0), 0, 255)
            out[s:e, 1] = np.clip(np.round(im / 128.0 + 128.0), 0, 255)
        else:
            out[s:e, 0] = re / 32768.0
            out[s:e, 1] = im / 32768.0
    return out


def cpu_baseline(raw_bytes, q, freqs, budget_s=10.0, code='h', N=N, FS=FS, correct_iq=False, faithful=False):
    """The plain-C oracle (oracle/chain.c) timed on this host's cores on a bounded sample of the same workload: all cores
    (OpenMP over chunks) and one core."""
    from oracle import c_chain, ref_chain
    shift = c_chain.shift_table(freqs, FS, N)
    dec, outs = ref_chain.cheby_sos(q), ref_chain.ellip_sos(FS // q, OMEGA_OUT)
    chunk_b = N * 2 * ITEM[code]
    avail = len(raw_bytes) // chunk_b
    L = 50.0 / FS if correct_iq else None

    def timed(threads, budget):
        want = max(threads, min(avail, (avail // threads) * threads))
        if threads == 1:
            want = min(avail, 8)
        if faithful:
            nshift = ref_chain.shift_table(freqs, FS, N)
            run = lambda: c_chain.process_stream_f32c(raw_bytes[:want * chunk_b], N, q, 'fm', nshift, outs, threads=threads)
        else:
            run = lambda: c_chain.process_stream(raw_bytes[:want * chunk_b], code, N, q, 'fm', shift, dec, outs, threads=threads, iq_inductance=L)
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
            'sample': f'{reps} pass(es) over {want} chunks x {N} samples of the same stream ({len(freqs)} row(s), dec={q}'
                      f'{", --correct-iq" if correct_iq else ""}), '
                      + ("oracle/chain.c (C port of the reference's complex64 chain for float32 records)" if faithful else 'oracle/chain.c (fp64 C port of the reference chain)')
                      + f', {threads} OpenMP threads, {dt:.1f} s',
            'single_core': {'value': round(v_one, 3), 'unit': 'MSamples/s', 'cores': 1,
                            'sample': f'{reps1} pass(es) over {want1} chunks, 1 thread, {dt1:.1f} s'}}


def committed_record(workload, correct_iq, faithful=False):
    """The newest committed counter record of this workload (tools/pmc.sh -> profiles/): (file name, record) or (None, None)."""
    tag = workload + ('_iq' if correct_iq and not WORKLOADS[workload][6] else '') + ('_faithful' if faithful else '')
    names = [f'r05_pmc_record_{tag}.json', f'r04_pmc_record_{tag}.json'] + (['r03_pmc_record.json'] if tag == 'fm64' else [])
    for name in names:
        try:
            return name, json.load(open(os.path.join(ROOT, 'profiles', name)))
        except (OSError, ValueError):
            continue
    return None, None


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


class Gpu:
    """The handful of runtime calls the harness needs, all through the C ABI (include/sdrterm_hip.h)."""

    def __init__(self, nat, device):
        self.nat, self.lib = nat, nat.lib()
        nat.check(self.lib.sdrt_set_device(device))

    def malloc(self, nbytes):
        return self.nat.DeviceBuffer(max(int(nbytes), 16))

    def stream(self):
        p = C.c_void_p()
        self.nat.check(self.lib.sdrt_stream_create(C.byref(p)))
        return p.value

    def event(self):
        p = C.c_void_p()
        self.nat.check(self.lib.sdrt_event_create_timed(C.byref(p)))
        return p.value

    def record(self, ev, stream):
        self.nat.check(self.lib.sdrt_event_record(ev, stream))

    def elapsed_ms(self, a, b):
        ms = C.c_double()
        self.nat.check(self.lib.sdrt_event_elapsed_ms(a, b, C.byref(ms)))
        return ms.value

    def sync(self):
        self.nat.check(self.lib.sdrt_device_synchronize())

    def d2d(self, dst, src, nbytes, stream=None):
        if stream is None:
            self.nat.check(self.lib.sdrt_memcpy_d2d(dst, src, nbytes))
        else:
            self.nat.check(self.lib.sdrt_memcpy_d2d_async(dst, src, nbytes, stream))

    def fill_stream(self, buf, n_samples, seed, code, fs, chunk_samples):
        """Synthesise min(n_samples, UNIQUE_SAMPLES) samples on the host, block by block, and tile them over the buffer on the device."""
        sb = 2 * ITEM[code]
        uniq = min(n_samples, max(UNIQUE_SAMPLES // chunk_samples, 1) * chunk_samples)
        blk = 1 << 22
        for s in range(0, uniq, blk):
            e = min(uniq, s + blk)
            buf.upload_at(s * sb, synth_iq_host(e - s, seed + (s // blk) * 7919, code, fs, start=s))
        done = uniq
        while done < n_samples:                       # doubling copies: log2(n / uniq) of them
            cnt = min(done, n_samples - done)
            self.d2d(buf.ptr + done * sb, buf.ptr, cnt * sb)
            done += cnt
        self.sync()
        return uniq


def parity_check(gpu, eng, raw_buf, out_buf, n_chunks, code, q, fs, freqs, row_modes, correct_iq, chunk_samples, raw_offset=0, faithful=False):
    """Outputs of sampled chunks of the buffer that was just timed against oracle/chain.c on the same raw bytes (outside the timed
    region).  --correct-iq: the timed steps carry the offset from step to step, so the check is an extra pass from a zero offset
    over the first eight chunks of the same buffer instead."""
    import numpy as np
    from oracle import c_chain, ref_chain
    nat = gpu.nat
    m, rows, cb = eng.m, len(freqs), chunk_samples * 2 * ITEM[code]
    if correct_iq:
        idx = list(range(min(8, n_chunks)))
        d_o = gpu.malloc(rows * len(idx) * m * 8)
        eng.iq_offset = 0j
        eng.process_device(raw_buf.ptr + raw_offset, len(idx), d_o.ptr, 0)
        gpu.sync()
        got = d_o.download(np.empty((rows, len(idx), m)))
        d_o.free()
        how = 'extra untimed pass from a zero IQ offset over the first chunks of the timed buffer'
    else:
        idx = sorted({0, 1, n_chunks // 3, n_chunks // 2 + 1, max(n_chunks - 2, 0), n_chunks - 1})
        got = np.empty((rows, len(idx), m))
        for r in range(rows):
            for k, c in enumerate(idx):
                out_buf.download_at(((r * n_chunks) + c) * m * 8, got[r, k])
        how = 'outputs of the last timed step'
    raw = np.empty((len(idx), cb), dtype=np.uint8)
    for k, c in enumerate(idx):
        raw_buf.download_at(raw_offset + c * cb, raw[k])
    shift = c_chain.shift_table(freqs, fs, chunk_samples)
    dec, outs = ref_chain.cheby_sos(q), ref_chain.ellip_sos(int(fs) // q, OMEGA_OUT)
    L = 50.0 / fs if correct_iq else None
    modes = sorted(set(row_modes)) if row_modes else ['fm']
    if faithful:
        # the reference's complex64 chain (oracle/chain.c: orc_process_stream_f32c, pinned bit for bit to the reference's vectors),
        # with the shift table numpy builds, as the reference does
        nshift = ref_chain.shift_table(freqs, fs, chunk_samples)
        exp = {md: c_chain.process_stream_f32c(raw.tobytes(), chunk_samples, q, md, nshift, outs, threads=4)[0] for md in modes}
    else:
        exp = {md: c_chain.process_stream(raw.tobytes(), code, chunk_samples, q, md, shift, dec, outs, threads=4, iq_inductance=L)[0] for md in modes}
    fm_err, am_rel = 0.0, 0.0
    for r in range(rows):
        md = row_modes[r] if row_modes else 'fm'
        e = exp[md][:, r, :]
        d = float(np.abs(got[r] - e).max())
        if md == 'fm':
            fm_err = max(fm_err, d)
        else:
            am_rel = max(am_rel, d / max(1.0, float(np.abs(e).max())))
    ok = bool(np.isfinite(got).all()) and fm_err <= 1e-6 and am_rel <= 1e-6
    res = {'max_abs_err': fm_err, 'tol': 1e-6, 'ok': ok, 'chunks': idx, 'rows': rows,
           'checker': "oracle/chain.c, the reference's complex64 chain for float32 records (bit-pinned to the reference's vectors)" if faithful
                      else 'oracle/chain.c (fp64 CPU restatement of the reference chain)',
           'what': how}
    if row_modes and 'am' in row_modes:
        res['am_max_err_rel_to_peak'] = am_rel
    return res


def main():
    global FS, N
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1500)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='fm64')
    ap.add_argument('--correct-iq', action='store_true', help='the workload with --correct-iq (single-VFO workloads; simo8 always has it)')
    ap.add_argument('--chunks', type=int, default=0, help='chunks per step (fm64: per GPU, default 131072 = 2^32 samples = 16 s of the 256 MS/s stream; simo64: 512)')
    ap.add_argument('--settle-seconds', type=float, default=2.0, help='untimed back-to-back steps before the timed region')
    ap.add_argument('--long-seconds', type=float, default=2.0, help='steady-state run after the timed region (0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the sampled comparison with the oracle after the timed region')
    ap.add_argument('--no-kernel-events', action='store_true', help='diagnostic: no per-kernel HIP events in the timed region')
    ap.add_argument('--feed', choices=('auto', 'bcast', 'scatter', 'allgather'), default='auto',
                    help='many-row workloads: how a block reaches the ranks (scatter: from rank 0 as 1/N shards + all-gather, the default when the block divides '
                         'by N; bcast: ncclBroadcast from rank 0; allgather: every rank holds 1/N)')
    ap.add_argument('--no-same-arithmetic', action='store_true', help="skip the fp64 kernel's side figure")
    ap.add_argument('--force-comm', action='store_true', help='diagnostic: the RCCL communicator with a single rank too (barrier / maximum / block exchange path on one GPU)')
    ap.add_argument('--faithful', action='store_true', help="float32 workloads (fm64f, f32x256) in the reference's own complex64 arithmetic (sdrt_config.faithful_f32); "
                                                           'parity is then checked against the complex64 restatement in oracle/chain.c, bit-pinned to the reference')
    ap.add_argument('--no-comm-selftest', action='store_true', help='skip the pre-flight of the block feeds (N > 1)')
    ap.add_argument('--dry-run', action='store_true', help='with --gpus N > 1 and no launcher: print the ranks that would be started')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:], dry_run=args.dry_run))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    from sdrterm_amd import _native as nat, parallel
    from sdrterm_amd.engine import Engine
    if nat.device_count() < 1:
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    local_dev = parallel.device_for(local_rank)       # = LOCAL_RANK unless SDRTERM_DEVICE_MAP says otherwise
    gpu = Gpu(nat, local_dev)
    lib = gpu.lib

    # one communicator for everything the ranks say to each other (barrier, maximum, block exchange): RCCL behind the C ABI
    comm = None
    if world > 1 or args.force_comm:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        comm = parallel.RcclTransport(rank, world, local_dev)        # the id travels over TCP at MASTER_ADDR:MASTER_PORT+17
    q, rows_total, default_chunks, scaling, code, FS, wl_iq, wl_modes = WORKLOADS[args.workload]
    correct_iq = bool(args.correct_iq or wl_iq)
    faithful = bool(args.faithful)
    if faithful and code != 'f':
        raise SystemExit('--faithful is for the float32 workloads (fm64f, f32x256): the other encodings are float64 in the reference too')
    N = 131072 // (2 * ITEM[code])            # complex samples per 131072-byte chunk
    n_chunks = args.chunks or default_chunks
    if rows_total > 1 and world > 1 and not args.chunks:
        # rows are sharded: a rank's share of a step shrinks with N, and at the one-GPU block size a step at N = 8 would be ~0.15 ms
        # against two collectives' launch latency -- the run would measure RCCL latency, not the design.  The block grows with the
        # ranks instead, so that a rank's step keeps the one-GPU duration (>= 1 ms) and the exchange is what a long stream costs per
        # sample: per-GPU work fixed -> "weak" on the line; value is samples per second either way
        n_chunks *= world
        scaling = 'weak'
    if faithful and not args.chunks:
        # one (chunk, row) sequence per lane: the faithful kernel wants >= 64 x 4 SIMDs x 256 CUs x a few sequences in flight
        n_chunks = {'fm64f': 131072, 'f32x256': 1024}[args.workload]
    n_samples = n_chunks * N
    sb = 2 * ITEM[code]
    block_bytes = n_samples * sb
    single = rows_total == 1                  # one VFO: chunks sharded over the ranks; else rows sharded, blocks broadcast
    feed = args.feed
    if feed == 'auto':
        feed = 'scatter' if block_bytes % max(world, 1) == 0 else 'bcast'
    # pre-flight of the feeds: one small block by every route, checked on every rank, before anything is timed (the default feed
    # has never met RCCL with N > 1 on this builder's boxes: the run must not bet the whole curve on it)
    comm_report = None
    if comm is not None and not args.no_comm_selftest:
        try:
            comm_report = comm.selftest()
        except Exception as e:                       # an RCCL error: say which, leave non-zero (a fresh process per run)
            raise SystemExit(f'comm selftest failed on rank {rank}: {type(e).__name__}: {e}')
        if not single:
            needed = {'scatter': 'scatter_ok', 'bcast': 'bcast_ok', 'allgather': 'allgather_ok'}[feed]
            if not comm_report[needed]:
                if args.feed == 'auto' and comm_report['bcast_ok']:
                    comm_report['feed_fallback'] = f'{feed} delivered wrong bytes in the pre-flight: ncclBroadcast instead'
                    feed = 'bcast'
                else:
                    raise SystemExit(f'comm selftest: feed {feed!r} delivered wrong bytes ({json.dumps(comm_report)})')
        comm_report['feed'] = feed if not single else 'none (chunk-sharded: no data-path collective)'
    src = None
    if single:
        freqs_all = [float(CENTER)]
        freqs, (lo, hi) = freqs_all, (0, 1)
        bufs = [gpu.malloc(block_bytes)]
        gpu.fill_stream(bufs[0], n_samples, 1234 + rank, code, FS, N)
    else:
        freqs_all = simo_rows(rows_total)
        freqs, (lo, hi) = parallel.shard_rows(freqs_all, world, rank)
        # two block buffers per rank: block k+1 arrives beside the kernels of block k.  bcast / scatter: the stream lives on rank 0
        # (where the reader's Queue ends).  allgather: every rank holds the same 1/N of every block (the whole stream is
        # synthesised everywhere from one seed, a rank keeps its slice)
        allg = feed == 'allgather' and comm is not None
        if n_samples % world and allg:
            raise SystemExit('--feed allgather needs a block that divides by the number of ranks')
        if rank == 0 or allg:
            whole = gpu.malloc(2 * block_bytes)
            gpu.fill_stream(whole, 2 * n_samples, 1234, code, FS, N)
            if allg:
                per = block_bytes // world
                src = gpu.malloc(2 * per)
                for b in range(2):
                    gpu.d2d(src.ptr + b * per, whole.ptr + b * block_bytes + rank * per, per)
                gpu.sync()
                whole.free()
            else:
                src = whole
        if comm is None:
            # one GPU, no exchange: the two resident blocks are processed where they lie ("inputs already resident in HBM when the
            # timed region starts"); with a communicator every step's block first travels into one of two block buffers
            class _View:
                def __init__(self, base, off):
                    self.ptr, self._base, self._off = base.ptr + off, base, off

                def download_at(self, offset, out):
                    return self._base.download_at(self._off + offset, out)
            bufs = [_View(src, b * block_bytes) for b in range(2)]
        else:
            bufs = [gpu.malloc(block_bytes) for _ in range(2)]
    modes = list(wl_modes[lo:hi]) if wl_modes else None
    eng = Engine(FS, q, 'fm', code, row_freqs=freqs, apply_shift=True, omega_out=OMEGA_OUT, device=local_dev, correct_iq=correct_iq,
                 row_modes=modes, reserve_chunks=n_chunks, faithful_f32=faithful)         # no allocation inside the timed calls
    rows = len(freqs)
    out = gpu.malloc(rows * n_chunks * eng.m * 8)
    # everything is submitted to one non-blocking stream of this process (not the legacy default stream, which orders itself against
    # every other stream)
    S = gpu.stream()
    s_comm = gpu.stream() if (comm is not None and not single) else None
    gpu.sync()

    state = {'k': 0}

    def fetch(k):
        """Block k of the stream into buffer k % 2 on every rank (many-row workloads with a communicator)."""
        if comm is None:
            return
        dst = bufs[k % 2]
        if s_comm is not None and feed == 'allgather':
            nat.check(lib.sdrt_stream_wait_stream(s_comm, S))        # after the kernels still reading dst
            comm.allgather_device(src.ptr + (k % 2) * (block_bytes // world), dst.ptr, block_bytes // world, s_comm)
            return
        if rank == 0:
            gpu.d2d(dst.ptr, src.ptr + (k % 2) * block_bytes, block_bytes, S)
        if s_comm is not None:
            nat.check(lib.sdrt_stream_wait_stream(s_comm, S))        # after the copy above / the kernels still reading dst
            comm.bcast_device(dst.ptr, block_bytes, 0, s_comm, scatter=feed == 'scatter')

    run = eng.process_device

    def step():
        k = state['k']
        if not single:
            fetch(k + 1)                                               # beside this step's kernels (other buffer)
        run(bufs[k % len(bufs)].ptr, n_chunks, out.ptr, S)
        if s_comm is not None:
            nat.check(lib.sdrt_stream_wait_stream(S, s_comm))         # the next step's kernels need that block
        state['k'] = k + 1

    def over_ranks(value):
        """Every rank's value (an all-gather of 8 bytes per rank through the communicator: also the barrier)."""
        if comm is None:
            return [float(value)]
        return [struct.unpack('<d', b)[0] for b in comm.allgather_bytes(struct.pack('<d', float(value)))]

    def barrier():
        over_ranks(0.0)

    if not single:
        fetch(0)
        if s_comm is not None:
            nat.check(lib.sdrt_stream_wait_stream(S, s_comm))
    for _ in range(args.warmup):
        step()
    gpu.sync()
    t_settle = time.perf_counter()
    settle_steps = 0
    # (the ranks decide together when the settling is over: with a collective in every step a rank that left the loop one round
    # before the others would leave them waiting in it)
    while max(over_ranks(time.perf_counter() - t_settle)) < args.settle_seconds:
        for _ in range(16):
            step()
        gpu.sync()
        settle_steps += 16

    # ---- the timed region: exactly K steps ---------------------------------------------------------------------------
    eng.profile(not args.no_kernel_events)
    eng.profile_reset()
    evs = [gpu.event() for _ in range(args.steps + 1)]     # recorded on the submission stream
    barrier()
    gpu.sync()
    t0 = time.perf_counter()
    gpu.record(evs[0], S)
    for i in range(args.steps):
        step()
        gpu.record(evs[i + 1], S)
    gpu.sync()
    barrier()
    elapsed = max(over_ranks(time.perf_counter() - t0))
    k_ms, k_n = eng.kernel_time(0)
    p_ms, p_n = eng.kernel_time(1)
    q_ms, q_n = eng.kernel_time(2)
    eng.profile(False)
    step_ms = sorted(gpu.elapsed_ms(evs[i], evs[i + 1]) for i in range(args.steps))
    last_buf = bufs[(state['k'] - 1) % len(bufs)]
    parity = None
    if not args.no_parity:
        # every rank checks ITS rows / chunks; rank 0 prints its own record with the other ranks' verdicts folded in
        parity = parity_check(gpu, eng, last_buf, out, n_chunks, code, q, FS, freqs, modes, correct_iq, N, faithful=faithful)
        if comm is not None:
            votes = comm.allgather_bytes(struct.pack('<dd', 1.0 if parity['ok'] else 0.0, float(parity['max_abs_err'])))
            votes = [struct.unpack('<dd', v) for v in votes]
            parity['ranks_ok'] = [bool(v[0]) for v in votes]
            parity['max_abs_err_over_ranks'] = max(v[1] for v in votes)
            parity['ok'] = all(parity['ranks_ok'])

    # ---- steady state: a longer run, no per-kernel events ---------------------------------------------------------------
    steady = None
    if args.long_seconds > 0:
        per = max(elapsed / args.steps, 1e-6)
        n_long = max(16, int(args.long_seconds / per))
        barrier()
        gpu.sync()
        t1 = time.perf_counter()
        for _ in range(n_long):
            step()
        gpu.sync()
        barrier()
        dt_long = max(over_ranks(time.perf_counter() - t1))
        steady = {'steps': n_long, 'seconds': round(dt_long, 3), 'ms_per_step': round(dt_long / n_long * 1e3, 4)}

    KERNEL_OF = {'i8w': 'chain_i8w_kernel', 'i8t': 'chain_i8t_kernel', 'f64': 'chain_kernel', 'seq': 'chain_seq_kernel', 'f32c': 'f32c_seq_kernel', 'sq': 'chain_sq_kernel'}
    if rank == 0:
        per_gpu = 1 if scaling == 'strong' else world
        total_samples = float(n_samples) * args.steps * per_gpu
        value = total_samples / elapsed / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        # algorithmic bytes per input sample on ONE GPU: 4 B of int16 IQ in + 8 B per row and q samples out
        alg_bps = 2.0 * ITEM[code] + 8.0 * rows / q
        k_avg_ms = k_ms / max(k_n, 1) if k_n else None
        p_avg_ms = p_ms / max(p_n, 1) if p_n else None
        q_avg_ms = q_ms / max(q_n, 1) if q_n else None
        # the chain kernel's own algorithmic bytes: input + the decimated complex128 samples it hands to the post kernel
        achieved = alg_bps * n_samples / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms else None
        path_achieved = alg_bps * n_samples / (ms_per_step * 1e-3) / 1e9
        # int8 matrix work of the dominant kernel: matrix instructions per 4096-sample tile x 16*16*64*2 operations
        fused_iq = correct_iq and rows == 1
        # (decimation below 32: 256 / q / 8 passes of the eight-outputs-per-column form)
        mfma_table = {'h': {64: 184, 32: 216, 16: 2 * 216, 8: 4 * 216, 4: 8 * 216}, 'f': {64: 288, 32: 336}, 'B': {64: 104, 32: 120, 16: 2 * 120, 8: 4 * 120}}
        mfma_per_tile = (mfma_table[code].get(q, 0) + (16 if fused_iq else 0) or None) if eng.stage_a == 'i8w' else ({'h': 160, 'f': 224}.get(code) if eng.stage_a == 'i8t' else None)
        if eng.stage_a == 'sq':
            # one pass: two aggregate row tiles + 2 x (256 / q) / 16 local row tiles, 72 matrix instructions each per 4096-sample tile
            mfma_per_tile = 72 * (2 + 32 // q)
        i8_ops = mfma_per_tile * 32768.0 * rows * n_samples / 4096 if mfma_per_tile else None
        # hardware counters cannot be read from inside this run: the committed record of rocprofv3 passes over this very workload
        # (tools/pmc.sh) is quoted under "from_profile", with the file it comes from and the commit it was taken at; traffic is per
        # launch and scales with the chunks of a launch, so a run with another --chunks quotes it scaled (and says so)
        pmc_file, pmc = committed_record(args.workload, correct_iq, faithful)
        if eng.stage_a == 'f32c':
            KERNEL_OF['f32c'] = 'f32c_fused_kernel'            # kernels_f32c.hip: one kernel, no preparation pass
        pmc_ok = bool(pmc) and pmc.get('kernel', '') in KERNEL_OF[eng.stage_a] and pmc.get('chunks_per_step_per_gpu')
        from_profile = None
        if pmc_ok:
            scale = n_chunks / float(pmc['chunks_per_step_per_gpu'])
            from_profile = {'file': 'profiles/' + pmc_file, 'commit': pmc.get('commit'),
                            'traffic': int(pmc.get('hbm_bytes_per_launch') * scale) if pmc.get('hbm_bytes_per_launch') else None,
                            'traffic_scaled_from_chunks': None if scale == 1.0 else pmc['chunks_per_step_per_gpu'],
                            'mfma_i8_busy': pmc.get('mfma_busy'), 'mfma_valu_coexec': pmc.get('mfma_valu_coexec'), 'clock_mhz': pmc.get('clock_mhz'),
                            'note': 'rocprofv3 --pmc passes of this workload (tools/pmc.sh), not measured by this run'}
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
        if eng.stage_a == 'f32c':
            # the reference's recurrence, one sequence per lane: 2 passes x 4 sections x 9 packed single-precision operations (no
            # fused multiply-adds: every rounding is the reference's) per row-sample; the vector units' packed fp32 rate is its
            # other roof (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)
            ops = 2 * 4 * 9 * 2 * float(rows) * n_samples
            peak = 256 * 4 * 16 * 2 * 2.4e9 / 1e12
            roof = dict(hbm, f32_valu={'achieved': round(ops / (k_avg_ms * 1e-3) / 1e12, 2) if k_avg_ms else None, 'peak': round(peak, 1),
                                        'unit': 'Top/s (packed fp32 multiply / add, unfused)',
                                        'frac': round(ops / (k_avg_ms * 1e-3) / 1e12 / peak, 4) if k_avg_ms else None},
                        kernel_traffic_model=('f32c_fused_kernel (rows % 64 == 0): 8 B complex64 per row-sample written by the forward pass and read by the backward pass; '
                                              'the shifted samples computed from the records and the transposed shift table (L2 / MALL) as the forward pass goes') if rows % 64 == 0 else
                                             'f32c_fused_kernel, a lane reading its own records: 8 B per row-sample of records read (whole 128-byte lines per lane), 8 B of complex64 written by the forward pass and read by the backward pass')
        elif rows == 1 or not i8['int8_pops']:
            roof = dict(hbm, **i8)
        else:
            # many rows per input sample: the int8 matrix work, not the input stream, is what bounds the kernel
            roof = {'bound': 'mfma', 'kernel': KERNEL_OF[eng.stage_a], 'achieved': round(i8['int8_pops'] * 1e3, 1),
                    'peak': I8_PEAK_POPS * 1e3, 'unit': 'TFLOP/s', 'frac': i8['int8_frac'], 'traffic': hbm['traffic'],
                    'avg_kernel_ms': hbm['avg_kernel_ms'], 'launches': k_n, 'ops': 'int8 multiply-adds x 2 (dense int8 matrix peak)',
                    'hbm': {k: hbm[k] for k in ('achieved', 'frac', 'alg_bytes_per_sample', 'path_achieved', 'path_frac', 'read_frac')},
                    'from_profile': from_profile}
        feed_txt = {'bcast': 'resident on rank 0, RCCL broadcast per step', 'scatter': 'resident on rank 0, sent as 1/N shards + RCCL all-gather per step',
                    'allgather': 'resident as 1/N shards per rank, RCCL all-gather per step'}[feed]
        mode_txt = 'FM' if not wl_modes else 'FM+AM mixed (' + ','.join(wl_modes) + ')'
        line = {
            'metric': METRIC[args.workload],
            'value': round(value, 1), 'unit': 'MSamples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': scaling, 'vs_baseline': None,
            # what the arithmetic is: int8 x int8 -> int32 matrix products of 5 base-256 weight digits x 2 byte slices (exact),
            # recombined and finished in fp64; 'f64' when the fp64 matrix-core kernel runs
            'dtype': ('f32->4 x i8 block fixed point, i8x5-split+f64' if code == 'f' else 'i8x5-split+f64') if eng.stage_a in ('i8w', 'i8t', 'sq')
                     else ("c64 (the reference's own arithmetic for float32 records) + f64" if eng.stage_a == 'f32c' else 'f64'), 'data': 'synthetic',
            'config': {'workload': (f'configs[1]{"" if code == "h" and q == 64 else (f" as {ENC_NAME[code]} records" if code != "h" else f" at decimation {q}")}: {FS // 1_000_000} MS/s synthetic {ENC_NAME[code]} IQ (FM tone + noise), FM demod, dec={q}, single VFO, '
                                    f'center={CENTER} Hz{", --correct-iq" if correct_iq else ""}, {n_chunks} chunks x {N} samples per GPU per step, inputs resident in HBM')
                       if single else
                       (f'configs[{CONFIG_NO[args.workload]}]: --simo {rows_total} VFO rows sharded {rows}/GPU, {FS // 1_000_000} MS/s synthetic {ENC_NAME[code]} IQ, {mode_txt}, dec={q}'
                        f'{", --correct-iq" if correct_iq else ""}, {n_chunks} chunks x {N} samples per step, ' + (feed_txt if world > 1 else 'inputs resident in HBM')),
                       'chunks_per_step_per_gpu': n_chunks, 'samples_per_chunk': N, 'rows_per_gpu': rows,
                       'parallelism': f'chunk-sharded x{world}' if single else f'rows-sharded x{world} + {feed}',
                       'faithful_f32': faithful,
                       #  one GPU, many rows: the resident blocks are processed where they lie; N > 1 stages every block
                       # into one of two block buffers (a device copy on rank 0 + the exchange) -- the N = 1 point carries no staging
                       'staging': None if single else ('none: resident blocks processed in place' if comm is None else f'device copy on rank 0 + {feed} per step'),
                       'settle_steps': settle_steps + args.warmup,
                       'distinct_samples_per_buffer': min(n_samples * (1 if single else 2), max(UNIQUE_SAMPLES // N, 1) * N)},
            'step_ms': {'median': round(step_ms[len(step_ms) // 2], 4), 'min': round(step_ms[0], 4), 'max': round(step_ms[-1], 4),
                        'timed_region_s': round(elapsed, 3)},
            'steady_state': steady,
            'roofline': roof,
            'kernels_ms': {'chain_kernel': round(k_avg_ms, 4) if k_avg_ms else None,
                           'post_kernel': round(p_avg_ms, 4) if p_avg_ms else None,
                           'iq_prefix_kernels': round(q_avg_ms, 4) if q_avg_ms else None},
            'stage_a': eng.stage_a,
            'parity': parity,
            'comm': comm_report,
        }
        # the same workload in the reference's own arithmetic: stage A on the fp64 matrix cores (SDRTERM_STAGE_A=f64), same
        # buffers, same process, a few steps
        e64 = None
        if not args.no_same_arithmetic and eng.stage_a not in ('f64', 'f32c'):
            try:
                e64 = Engine(FS, q, 'fm', code, row_freqs=freqs, apply_shift=True, omega_out=OMEGA_OUT, device=local_dev, correct_iq=correct_iq, row_modes=modes,
                             stage_a='f64')
            except nat.SdrtError as err:
                # (uint8 records at decimation 8: the float64 kernel's 8192 blocks per 65536-sample chunk do not fit its LDS)
                line['same_arithmetic'] = {'unsupported': str(err)}
        if e64 is not None:
            n64 = 4 if single else 2
            e64.process_device(bufs[0].ptr, n_chunks, out.ptr, S)
            gpu.sync()
            t64 = time.perf_counter()
            for _ in range(n64):
                e64.process_device(bufs[0].ptr, n_chunks, out.ptr, S)
            gpu.sync()
            ms64 = (time.perf_counter() - t64) / n64 * 1e3
            line['same_arithmetic'] = {'stage_a': e64.stage_a, 'dtype': 'f64', 'steps': n64, 'ms_per_step': round(ms64, 4),
                                       'value': round(n_samples / (ms64 * 1e-3) / 1e6, 1), 'unit': 'MSamples/s per GPU',
                                       'path_frac': round(alg_bps * n_samples / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            e64.close()
        if not args.no_cpu_baseline and world == 1:
            import numpy as np
            take = min(n_chunks, 2048 if single else 128) * N * sb
            sample = np.empty(take, dtype=np.uint8)
            bufs[0].download_at(0, sample)
            line['cpu_baseline'] = cpu_baseline(sample.tobytes(), q, freqs_all, code=code, N=N, FS=FS, correct_iq=correct_iq, faithful=faithful)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if comm is not None:
        barrier()
        comm.close()
    if parity is not None and not parity['ok']:
        raise SystemExit('parity check failed: ' + json.dumps(parity))


if __name__ == '__main__':
    main()
