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


# ---- structured streams (round 4): every value comes out of 64-bit integer arithmetic and one integer sine table ----------

_SIN_BITS = 16          # table entries per cycle: tone frequencies that are multiples of fs / 2^16 index it exactly
_SIN_SCALE = 30         # table values are round(2^30 sin)
_sin_tab = None


def _pi_fixed(frac_bits: int) -> int:
    """pi * 2^frac_bits from Machin's formula, Python integers only."""
    def atan_inv(x, one):
        total = term = one // x
        x2, k, sign = x * x, 3, -1
        while term:
            term //= x2
            total += sign * (term // k)
            k += 2
            sign = -sign
        return total
    one = 1 << (frac_bits + 16)
    return (4 * (4 * atan_inv(5, one) - atan_inv(239, one))) >> 16


def sin_table() -> np.ndarray:
    """int64[2^16]: round(2^30 sin(2 pi k / 2^16)), built by exact integer rotation (200-bit fixed point): the same on every
    machine, no libm."""
    global _sin_tab
    if _sin_tab is not None:
        return _sin_tab
    F = 200
    one = 1 << F
    n = 1 << _SIN_BITS
    d = (2 * _pi_fixed(F)) >> _SIN_BITS                # the step angle, fixed point

    def series(x, first, k0):                          # Taylor series of sin (first = x, k0 = 2) or cos (first = 1, k0 = 1)
        total = term = first
        k = k0
        x2 = (x * x) >> F
        while term:
            term = -((term * x2) >> F) // (k * (k + 1))
            total += term
            k += 2
        return total
    sd, cd = series(d, d, 2), series(d, one, 1)
    quarter = n // 4
    s, c = 0, one
    q = np.empty(quarter + 1, dtype=np.int64)
    half = 1 << (F - _SIN_SCALE - 1)
    for k in range(quarter + 1):
        q[k] = (s + half) >> (F - _SIN_SCALE)
        s, c = (s * cd + c * sd) >> F, (c * cd - s * sd) >> F
    tab = np.empty(n, dtype=np.int64)
    tab[:quarter + 1] = q
    tab[quarter:2 * quarter + 1] = q[::-1]
    tab[2 * quarter:] = -tab[:2 * quarter]
    _sin_tab = tab
    return tab


def _phase_word(f: float, fs: float) -> int:
    """Frequency as a 32-bit phase increment (cycles * 2^32 per sample), rounded once."""
    return int(round(f / fs * 4294967296.0)) & 0xFFFFFFFF


def _tone(idx: np.ndarray, word: int, amp_q10: int, extra=None):
    """(re, im) = amp * (cos, sin)(2 pi phase) in units of 2^-40 (amp is in 1/1024 LSB); phase = idx * word (+ extra) mod 2^32."""
    tab = sin_table()
    with np.errstate(over='ignore'):
        ph = idx.astype(np.uint64) * np.uint64(word)
        if extra is not None:
            ph = ph + extra.astype(np.uint64)
    ph &= np.uint64(0xFFFFFFFF)
    k = (ph >> np.uint64(32 - _SIN_BITS)).astype(np.int64)
    n = 1 << _SIN_BITS
    return amp_q10 * tab[(k + n // 4) & (n - 1)], amp_q10 * tab[k]


def _noise4(idx2: np.ndarray, seed: int, half: int) -> np.ndarray:
    """Sum of four uniforms in [-half, half): a bell-shaped integer noise of sigma = half * 2 / sqrt(3)."""
    if half <= 0:
        return np.zeros(idx2.shape, dtype=np.int64)
    out = np.zeros(idx2.shape, dtype=np.int64)
    for j in range(4):
        z = splitmix64(idx2, seed + 0x1000 * (j + 1))
        out += ((z >> np.uint64(24)) % np.uint64(2 * half)).astype(np.int64) - half
    return out


_RANGE = {'b': (-128, 127), 'B': (-128, 127), 'h': (-32768, 32767), 'H': (-32768, 32767), 'i': (-2147483648, 2147483647),
          'I': (-2147483648, 2147483647), 'f': (-32768, 32767), 'd': (-32768, 32767)}


def case_ints(case: dict, n: int) -> np.ndarray:
    """(n * n_chunks, 2) int64 samples of a golden case (before the encoding's bias): ``case['sig']`` names the class of
    stream, absent = uniform white noise of amplitude ``amp`` (the round-1 class).

    sig kinds (all on top of, or instead of, that noise):
      rails    per chunk one of 'corners' (every component drawn from the encoding's rails and the byte-slice corner values),
               'const' (re = lowest, im = highest value of the encoding), 'two' (every component one of the two rails),
               'edges' (noise with the first and last 28 samples of the chunk at the rails, alternating)
      tone     FM tone: amplitude a, carrier f, modulation fm / deviation dev (Hz), plus bell noise of half-width ``noise``
      blocker  full-scale tone of amplitude a1 at f1 + weak tone a1 * 10^(db/20) at f2 (both multiples of fs / 65536) + optional noise
      impulse  noise with full-scale impulses at the listed sample positions of every chunk (negative = from the end)
      zero     the listed chunks are all zero
    """
    code, nch, seed, amp = case['code'], case['n_chunks'], case['seed'], case['amp']
    dc = tuple(case.get('dc', (0, 0)))
    sig = case.get('sig')
    lo, hi = _RANGE[code]
    total = n * nch
    if not sig:
        return iq_ints(total, seed, amp, dc)
    kind = sig['kind']
    idx = np.arange(total, dtype=np.uint64)
    idx2 = np.arange(2 * total, dtype=np.uint64)
    if kind in ('zero', 'impulse', 'rails'):
        out = iq_ints(total, seed, amp, dc)
    if kind == 'zero':
        for c in sig['chunks']:
            out[c * n:(c + 1) * n] = -_UNSIGNED_BIAS.get(code, 0)        # the VALUE zero (unsigned encodings are not re-centred)
    elif kind == 'impulse':
        for c in range(nch):
            for j, p in enumerate(sig['at']):
                out[c * n + (p % n)] = (hi, lo) if (j + c) & 1 else (lo, hi)
    elif kind == 'rails':
        corners = np.array([lo, hi, lo, hi, lo, hi, -1, 0, 1, 127, 128, -128, -129, 255, 256, -256, -257,
                            hi - 255, lo + 255, hi - 256, lo + 256], dtype=np.int64)
        corners = np.clip(corners, lo, hi)
        pick = splitmix64(idx2, seed + 77) >> np.uint64(20)
        for c, how in enumerate(sig['chunks']):
            sl = slice(c * n, (c + 1) * n)
            p = pick[2 * c * n:2 * (c + 1) * n]
            if how == 'corners':
                out[sl] = corners[(p % np.uint64(len(corners))).astype(np.int64)].reshape(n, 2)
            elif how == 'two':
                out[sl] = np.where((p & np.uint64(1)).astype(bool), hi, lo).reshape(n, 2)
            elif how == 'const':
                out[sl] = (lo, hi)
            elif how == 'edges':
                e = np.where((np.arange(28) & 1).astype(bool), hi, lo)
                blk = out[sl]
                blk[:28, 0], blk[:28, 1] = e, e[::-1]
                blk[-28:, 0], blk[-28:, 1] = e[::-1], e
            elif how != 'noise':
                raise ValueError(how)
    elif kind == 'tone':
        fs = case['fs']
        mod = None
        if sig.get('dev'):
            # phase deviation (dev / fm) radians -> cycles * 2^32, times sin(2 pi fm t) from the table
            beta = int(round(sig['dev'] / sig['fm'] / (2 * np.pi) * 4294967296.0))
            _, s = _tone(idx, _phase_word(sig['fm'], fs), 1)
            mod = ((beta * s) >> _SIN_SCALE) & 0xFFFFFFFF
        re, im = _tone(idx, _phase_word(sig['f'], fs), sig['a'] << 10, mod)
        nz = _noise4(idx2, seed, sig.get('noise', 0)).reshape(total, 2)
        out = np.stack([(re + (1 << 39)) >> 40, (im + (1 << 39)) >> 40], axis=1) + nz
        out[:, 0] += dc[0]
        out[:, 1] += dc[1]
    elif kind == 'blocker':
        fs = case['fs']
        for f in (sig['f1'], sig['f2']):
            assert (int(f) * 65536) % int(fs) == 0, 'blocker tones must be multiples of fs / 65536 (exact table phase)'
        a1 = sig['a1'] << 10
        a2 = int(round(sig['a1'] * 1024 * 10.0 ** (sig['db'] / 20.0)))
        r1, i1 = _tone(idx, _phase_word(sig['f1'], fs), a1)
        r2, i2 = _tone(idx, _phase_word(sig['f2'], fs), a2)
        nz = _noise4(idx2, seed, sig.get('noise', 0)).reshape(total, 2)
        out = np.stack([(r1 + r2 + (1 << 39)) >> 40, (i1 + i2 + (1 << 39)) >> 40], axis=1) + nz
        out[:, 0] += dc[0]
        out[:, 1] += dc[1]
    else:
        raise ValueError(kind)
    return np.clip(out, lo, hi)


def encode(ints: np.ndarray, code: str, big_endian: bool = False) -> bytes:
    """Integers -> the reference's on-disk encoding (see raw_bytes)."""
    if code in ('f', 'd'):
        arr = (ints.astype(np.float64) / 32768.0).astype(_NP_CODE[code])
    else:
        arr = (ints + _UNSIGNED_BIAS.get(code, 0)).astype(_NP_CODE[code])
    arr = arr.astype(arr.dtype.newbyteorder('>' if big_endian else '<'))
    return arr.tobytes()


def case_bytes(case: dict, n: int) -> bytes:
    """The raw byte stream of a golden case: n complex samples per chunk."""
    return encode(case_ints(case, n), case['code'], bool(case.get('big_endian')))
