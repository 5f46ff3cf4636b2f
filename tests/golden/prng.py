"""Repo-defined counter-based integer PRNG (splitmix64 finaliser) for reproducible test inputs.

Golden fixtures store only (seed, parameters, expected outputs); inputs are regenerated from the
seed with pure 64-bit integer arithmetic, so they are bit-identical on every machine and do not
depend on numpy's generators.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(index: np.ndarray, seed: int) -> np.ndarray:
    """z = mix(seed + (index + 1) * golden), the splitmix64 output function applied to a counter."""
    with np.errstate(over='ignore'):
        z = (index.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def iq_ints(n_complex: int, seed: int, amp: int, dc=(0, 0), start: int = 0) -> np.ndarray:
    """(n_complex, 2) int64 array of interleaved (re, im), uniform in [-amp, amp) plus an integer DC offset.

    ``start`` is the index of the first complex sample, so consecutive chunks of one stream can be
    generated independently.
    """
    idx = np.arange(2 * start, 2 * (start + n_complex), dtype=np.uint64)
    z = splitmix64(idx, seed)
    v = (z >> np.uint64(24)) % np.uint64(2 * amp)
    out = v.astype(np.int64) - amp
    out = out.reshape(n_complex, 2)
    out[:, 0] += dc[0]
    out[:, 1] += dc[1]
    return out


_NP_CODE = {'b': np.int8, 'B': np.uint8, 'h': np.int16, 'H': np.uint16, 'i': np.int32, 'I': np.uint32,
            'f': np.float32, 'd': np.float64}

_UNSIGNED_BIAS = {'B': 128, 'H': 32768, 'I': 2147483648}


def raw_bytes(code: str, n_complex: int, seed: int, amp: int, dc=(0, 0), start: int = 0,
              big_endian: bool = False) -> bytes:
    """Interleaved IQ bytes in the reference's on-disk encoding ``code`` (file_util.DataType names).

    Integer encodings hold the integers directly (unsigned ones biased to mid-scale); float encodings
    hold integer / 2**15, which is exact in float32.
    """
    ints = iq_ints(n_complex, seed, amp, dc, start)
    if code in ('f', 'd'):
        arr = (ints.astype(np.float64) / 32768.0).astype(_NP_CODE[code])
    else:
        arr = (ints + _UNSIGNED_BIAS.get(code, 0)).astype(_NP_CODE[code])
    if big_endian:
        arr = arr.astype(arr.dtype.newbyteorder('>'))
    else:
        arr = arr.astype(arr.dtype.newbyteorder('<'))
    return arr.tobytes()
