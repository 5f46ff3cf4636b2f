#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry points: sdrt_process_host one batch at a time (pageable buffers) and
sdrt_submit_host / sdrt_wait with batches in flight (pageable and page-locked buffers), next to the plain host-to-device copy
rate of this box (the ceiling of the path: 4 bytes in per sample)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrterm_amd import _native as nat
from sdrterm_amd.engine import Engine
from tests.golden import prng


def main():
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    eng = Engine(256_000_000, 64, 'fm', 'h', row_freqs=[15000], omega_out=12500)
    one = np.frombuffer(prng.raw_bytes('h', eng.n * 32, 9, 8000), dtype=np.uint8)
    raw = np.tile(one, max(1, n_chunks // 32))[:n_chunks * eng.chunk_bytes].copy()
    nbytes = raw.size
    lib = nat.lib()
    # ceiling: page-locked -> device copies of the same size, back to back
    pin = [nat.pinned_array(nbytes) for _ in range(3)]
    for p in pin:
        p[:] = raw
    dev = nat.DeviceBuffer(nbytes)
    nat.check(lib.sdrt_memcpy_h2d(dev.ptr, pin[0].ctypes.data, nbytes))
    t0 = time.perf_counter()
    for i in range(calls):
        nat.check(lib.sdrt_memcpy_h2d(dev.ptr, pin[i % 3].ctypes.data, nbytes))
    h2d = nbytes * calls / (time.perf_counter() - t0)

    def rate(label, fn):
        fn(8)
        t0 = time.perf_counter()
        fn(calls)
        dt = (time.perf_counter() - t0) / calls
        print(json.dumps({'workload': label, 'chunks_per_call': n_chunks, 'input_MSamples_per_s': round(n_chunks * eng.n / dt / 1e6, 1),
                          'us_per_call': round(dt * 1e6, 1), 'input_GB_per_s': round(nbytes / dt / 1e9, 2),
                          'fraction_of_h2d_rate': round(nbytes / dt / h2d, 3), 'h2d_GB_per_s': round(h2d / 1e9, 2)}), flush=True)

    def sync_calls(k):
        for _ in range(k):
            eng.process(raw, n_chunks)

    def in_flight(bufs):
        def run(k):
            pend = []
            for i in range(k):
                pend.append(eng.submit(bufs[i % len(bufs)], n_chunks))
                if len(pend) == 3:
                    eng.collect(pend.pop(0))
            for h in pend:
                eng.collect(h)
        return run

    rate('sdrt_process_host, one batch at a time, pageable buffers', sync_calls)
    rate('sdrt_submit_host / sdrt_wait, 3 batches in flight, pageable input', in_flight([raw, raw.copy(), raw.copy()]))
    rate('sdrt_submit_host / sdrt_wait, 3 batches in flight, page-locked input', in_flight(pin))


if __name__ == '__main__':
    main()
