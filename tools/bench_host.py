#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry point (sdrt_process_host): pageable numpy buffers in and out."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrterm_amd.engine import Engine
from tests.golden import prng

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = Engine(256_000_000, 64, 'fm', 'h', row_freqs=[15000], omega_out=12500)
one = np.frombuffer(prng.raw_bytes('h', eng.n * 32, 9, 8000), dtype=np.uint8)
raw = np.tile(one, n_chunks // 32)
eng.process(raw, n_chunks)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    out = eng.process(raw, n_chunks)
dt = (time.perf_counter() - t0) / reps
print(json.dumps({'workload': 'sdrt_process_host, int16 FM dec=64, pageable host buffers', 'chunks': n_chunks,
                  'input_MSamples_per_s': round(n_chunks * eng.n / dt / 1e6, 1), 'ms_per_call': round(dt * 1e3, 2),
                  'input_GB_per_s': round(raw.size / dt / 1e9, 2)}))
