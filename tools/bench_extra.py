#!/usr/bin/env python3
"""Secondary workloads (BASELINE.json configs[2] and neighbours) through the C ABI, device-resident buffers, no torch.
Prints one JSON line per workload: input MSamples/s and the per-kernel HIP-event times.
usage: tools/bench_extra.py [substring of the workload names to run]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrterm_amd import _native as nat
from sdrterm_amd.engine import Engine
from tests.golden import prng

FS = 256_000_000
VFOS8 = [-40e6 + 15000, -30e6 + 15000, -20e6 + 15000, -10e6 + 15000, 10e6 + 15000, 20e6 + 15000, 30e6 + 15000, 15000]


ONLY = sys.argv[1] if len(sys.argv) > 1 else None       # substring of the workload names to run


def run(name, n_chunks, steps=5, **kw):
    if ONLY and ONLY not in name:
        return
    eng = Engine(FS, **kw)
    BLK = 512        # distinct chunks before the stream repeats (a short period keeps every workgroup on the same few L2 lines)
    one = np.frombuffer(prng.raw_bytes(kw.get('code', 'h'), eng.n * BLK, 9, 100 if kw.get('code', 'h') in 'bB' else 8000), dtype=np.uint8)
    d_raw = nat.DeviceBuffer(n_chunks * eng.chunk_bytes)
    for i in range(0, n_chunks, BLK):
        k = min(BLK, n_chunks - i)
        nat.check(nat.lib().sdrt_memcpy_h2d(d_raw.ptr + i * eng.chunk_bytes, one.ctypes.data, k * eng.chunk_bytes))
    d_out = nat.DeviceBuffer(eng.rows * n_chunks * eng.m * 8)
    for _ in range(2):
        eng.process_device(d_raw.ptr, n_chunks, d_out.ptr)
    nat.check(nat.lib().sdrt_device_synchronize())
    eng.profile(True); eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.process_device(d_raw.ptr, n_chunks, d_out.ptr)
    nat.check(nat.lib().sdrt_device_synchronize())
    dt = time.perf_counter() - t0
    k0, n0 = eng.kernel_time(0); k1, n1 = eng.kernel_time(1); k2, n2 = eng.kernel_time(2)
    print(json.dumps({'workload': name, 'stage_a': eng.stage_a, 'rows': eng.rows, 'chunks': n_chunks,
                      'input_MSamples_per_s': round(n_chunks * eng.n * steps / dt / 1e6, 1),
                      'row_MSamples_per_s': round(eng.rows * n_chunks * eng.n * steps / dt / 1e6, 1),
                      'ms_per_step': round(dt / steps * 1e3, 3),
                      'kernel_ms': {'chain': round(k0 / max(n0, 1), 3), 'post': round(k1 / max(n1, 1), 3), 'iq_prefix': round(k2 / max(n2, 1), 3)}}), flush=True)
    for b in (d_raw, d_out):
        b.free()
    eng.close()


if __name__ == '__main__':
    run('configs[1] single VFO FM dec=64 int16', 16384, decimation=64, mode='fm', code='h', row_freqs=[15000], omega_out=12500)
    run('configs[1] variant center=0 (no shift)', 16384, decimation=64, mode='fm', code='h', row_freqs=[0], apply_shift=False, omega_out=12500)
    run('configs[2]-like 8 VFOs FM dec=32 int16', 4096, decimation=32, mode='fm', code='h', row_freqs=VFOS8, omega_out=12500)
    run('configs[2] 8 VFOs FM dec=32 int16 --correct-iq', 4096, decimation=32, mode='fm', code='h', row_freqs=VFOS8, omega_out=12500, correct_iq=True)
    run('configs[2] 8 VFOs FM+AM mixed dec=32 int16 --correct-iq', 4096, decimation=32, mode='fm', code='h', row_freqs=VFOS8, omega_out=12500, correct_iq=True,
        row_modes=['fm', 'am'] * 4)
    run('configs[2] 8 VFOs AM dec=32 int16 --correct-iq --normalize', 4096, decimation=32, mode='am', code='h', row_freqs=VFOS8, omega_out=12500, correct_iq=True, normalize=True)
    run('single VFO FM dec=64 uint8', 8192, decimation=64, mode='fm', code='B', row_freqs=[15000], omega_out=12500)
    run('single VFO FM dec=32 uint8', 8192, decimation=32, mode='fm', code='B', row_freqs=[15000], omega_out=12500)
    run('single VFO FM dec=64 uint8 --correct-iq', 8192, decimation=64, mode='fm', code='B', row_freqs=[15000], omega_out=12500, correct_iq=True)
    run('8 VFOs FM dec=64 uint8 --correct-iq', 2048, decimation=64, mode='fm', code='B', row_freqs=VFOS8, omega_out=12500, correct_iq=True)
    run('single VFO FM dec=64 float32', 32768, decimation=64, mode='fm', code='f', row_freqs=[15000], omega_out=12500)
    run('single VFO FM dec=64 int16 --correct-iq', 8192, decimation=64, mode='fm', code='h', row_freqs=[15000], omega_out=12500, correct_iq=True)
    run('single VFO FM dec=32 int16 --correct-iq', 8192, decimation=32, mode='fm', code='h', row_freqs=[15000], omega_out=12500, correct_iq=True)
    run('single VFO FM dec=16 int16', 8192, decimation=16, mode='fm', code='h', row_freqs=[15000], omega_out=12500)
    run('single VFO FM dec=8 int16', 8192, decimation=8, mode='fm', code='h', row_freqs=[15000], omega_out=12500)
    run('single VFO FM dec=16 uint8', 4096, decimation=16, mode='fm', code='B', row_freqs=[15000], omega_out=12500)
    run('8 VFOs FM dec=16 int16', 1024, decimation=16, mode='fm', code='h', row_freqs=VFOS8, omega_out=12500)
    run('single VFO FM dec=4 int16', 2048, decimation=4, mode='fm', code='h', row_freqs=[15000], omega_out=12500)
    run('single VFO FM dec=2 int16 (the CLI default)', 1024, decimation=2, mode='fm', code='h', row_freqs=[15000], omega_out=12500)
    if os.environ.get('SDRTERM_STAGE_A') is None:
        os.environ['SDRTERM_STAGE_A'] = 'i8t'       # the same int8 workloads on the one-block-per-column kernel, for comparison
        run('[i8t] 8 VFOs FM dec=32 int16', 4096, decimation=32, mode='fm', code='h', row_freqs=VFOS8, omega_out=12500)
        run('[i8t] single VFO FM dec=64 uint8', 8192, decimation=64, mode='fm', code='B', row_freqs=[15000], omega_out=12500)
        run('[i8t] single VFO FM dec=64 float32', 32768, decimation=64, mode='fm', code='f', row_freqs=[15000], omega_out=12500)
