import os, sys, hashlib
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from sdrterm_amd import _native as nat
from sdrterm_amd.engine import Engine
from tests.golden import prng
for kw, nch in ((dict(decimation=64, mode='fm', code='h', row_freqs=[15000.0]), 4096),
                (dict(decimation=32, mode='am', code='h', row_freqs=[1000.0 * i for i in range(8)], correct_iq=True, normalize=True), 1024),
                (dict(decimation=64, mode='fm', code='h', row_freqs=[0.0], apply_shift=False), 4096),
                (dict(decimation=64, mode='fm', code='B', row_freqs=[15000.0]), 2048),
                (dict(decimation=64, mode='fm', code='f', row_freqs=[15000.0]), 4096),
                (dict(decimation=64, mode='fm', code='h', row_freqs=[15000.0], correct_iq=True), 4096),
                # the one-pass kernel of decimation 2 .. 16 (kernels_sq.hip; it has no barrier between a chunk's local phase and the next
                # chunk's first phases): many chunks per workgroup, 16- and 8-bit records, several rows, AM and FM
                (dict(decimation=2, mode='fm', code='h', row_freqs=[15000.0]), 3000),
                (dict(decimation=16, mode='fm', code='h', row_freqs=[15000.0, -40000.0, 7000.0]), 4099),
                (dict(decimation=8, mode='am', code='h', row_freqs=[15000.0]), 5000),
                (dict(decimation=16, mode='fm', code='B', row_freqs=[15000.0]), 3001),
                (dict(decimation=4, mode='am', code='B', row_freqs=[15000.0, 90000.0]), 1500)):
    eng = Engine(256_000_000, omega_out=12500, **kw)
    block = np.frombuffer(prng.raw_bytes(kw['code'], eng.n * 256, 5, 100 if kw['code'] == 'B' else 8000), dtype=np.uint8)
    d_raw = nat.DeviceBuffer(nch * eng.chunk_bytes)
    for i in range(0, nch, 256):
        nat.check(nat.lib().sdrt_memcpy_h2d(d_raw.ptr + i * eng.chunk_bytes, block.ctypes.data, min(256, nch - i) * eng.chunk_bytes))
    d_out = nat.DeviceBuffer(eng.rows * nch * eng.m * 8)
    digests = set()
    for rep in range(40):
        if kw.get('correct_iq'):
            eng.iq_offset = 0j
        eng.process_device(d_raw.ptr, nch, d_out.ptr)
        nat.check(nat.lib().sdrt_device_synchronize())
        out = d_out.download(np.empty(eng.rows * nch * eng.m, dtype=np.float64))
        digests.add(hashlib.sha1(out.tobytes()).hexdigest())
    print(eng.stage_a, kw.get('code'), kw['decimation'], eng.rows, 'distinct outputs over 40 runs:', len(digests))
    assert len(digests) == 1
print('deterministic')
