"""Shared helpers for the parity tests (inputs regenerated from seeds, golden loader, tolerances)."""
import json
import os

import numpy as np

from tests.golden import prng
from tests.golden.cases import CASES, BY_NAME, READ_SIZE  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden', 'reference_vectors.npz')

_ITEM = {'b': 1, 'B': 1, 'h': 2, 'H': 2, 'i': 4, 'I': 4, 'f': 4, 'd': 8}

# north_star tolerance: 1e-6 max-abs on float64 outputs.  AM on un-normalised integers has outputs
# ~1e6..1e8 where 1e-6 absolute is below fp64 resolution, so AM uses 1e-6 relative to
# the peak of the expected output (never looser than 1e-6 absolute when the peak is <= 1).
ABS_TOL = 1e-6


def tol_for(expected: np.ndarray, mode: str = 'am') -> float:
    """FM outputs are angles: the north-star 1e-6 max-abs applies as is.  AM scales with |x|^2."""
    if mode == 'fm':
        return ABS_TOL
    return ABS_TOL * max(1.0, float(np.abs(expected).max()))


def load_golden():
    z = np.load(GOLDEN)
    meta = json.loads(bytes(z['meta_json']).decode())
    return z, meta


def samples_per_chunk(code: str) -> int:
    return READ_SIZE // (2 * _ITEM[code])


def case_raw(case: dict) -> bytes:
    n = samples_per_chunk(case['code'])
    return prng.case_bytes(case, n)


def set_diag(monkeypatch, **switches) -> None:
    """Set / clear keys of the library's one diagnostic variable, SDRTERM_DIAG="key=value,key=value" (csrc/api.hip: sdrt_diag);
    a value of None removes the key."""
    cur = dict(kv.split('=', 1) for kv in os.environ.get('SDRTERM_DIAG', '').split(',') if '=' in kv)
    for k, v in switches.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv('SDRTERM_DIAG', ','.join(f'{k}={v}' for k, v in cur.items()))
    else:
        monkeypatch.delenv('SDRTERM_DIAG', raising=False)
