"""GPU parity: HIP path (through the C ABI) vs golden vectors of the real reference and vs the oracle."""
import ctypes

import numpy as np
import pytest

from tests.util import ABS_TOL, BY_NAME, CASES, case_raw, samples_per_chunk, tol_for

pytestmark = pytest.mark.gpu

from tests.util import set_diag as _diag  # noqa: E402


def _engine(case, **kw):
    from sdrterm_amd.engine import Engine
    from oracle import ref_chain
    rows = ref_chain.vfo_rows(case['center'], case.get('vfos'))
    return Engine(case['fs'], case['q'], case['mode'], case['code'], row_freqs=rows,
                  apply_shift=bool(case.get('vfos') or case['center']), omega_out=case['omega_out'],
                  normalize=bool(case.get('normalize')), correct_iq=bool(case.get('correct_iq')),
                  big_endian=bool(case.get('big_endian')), smooth=case.get('smooth', 0), **kw)


def test_mfma_layout_selftest():
    from sdrterm_amd import _native
    err = ctypes.c_double(-1)
    _native.check(_native.lib().sdrt_selftest_mfma(ctypes.byref(err)))
    assert err.value == 0.0
    bad = ctypes.c_int(-1)
    _native.check(_native.lib().sdrt_selftest_mfma_i8(ctypes.byref(bad)))
    assert bad.value == 0


@pytest.fixture(params=['auto', 'i8', 'f64', 'i8t'])
def stage_a(request, monkeypatch):
    """Every golden case runs under each value of sdrt_config.stage_a -- auto (the default: the int8 matrix cores where they apply,
    wide-column persistent kernel or the one-block-per-column 16x16x64 form, and a second look in float64 at the items whose
    in-band content is near the int8 forms' operator image), i8 (no second look), f64 -- and with the one-block 16x16x64 int8
    form forced (a diagnostic: SDRTERM_STAGE_A=i8t)."""
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    if request.param == 'i8t':
        monkeypatch.setenv('SDRTERM_STAGE_A', 'i8t')
    return request.param


# Stated limits ("Conditioning"): inputs whose in-band content lies below what an arithmetic other than the
# reference's own operation order can reproduce to 1e-6.  (case, stage-A kernel) -> measured bound.  These stay in the golden set:
# the test asserts the bound (a regression shows), asserts that the error is still above the tolerance (an improvement shows:
# drop the entry) and reports XFAIL.
#  * h_q64_fm_zero_iq_dc_c: a silent chunk under --correct-iq after a DC of 20000, seen from a row 15.7 kHz away: the chunk is the
#    decaying offset 164 dB down the stop band; the reference's FM output there is the angle of its own rounding residue
#    (read_file.py:64-77 -> demodulation.py:30-32).  1.5e-5 on every kernel, float64 included.
#  * h_q64_fm_blk120: in-band content 120 dB under a full-scale tone, a thirtieth of an LSB: past the 40-bit operator image of
#    the wide int8 kernel (1.5e-6); float64 (sdrt_config.stage_a = f64) holds it at 2e-8.
#    Under stage_a = auto that (chunk, row) is recognised and done again in float64: within the tolerance.
STATED_LIMITS = {
    ('h_q64_fm_zero_iq_dc_c', 'auto'): 5e-5, ('h_q64_fm_zero_iq_dc_c', 'i8'): 5e-5, ('h_q64_fm_zero_iq_dc_c', 'f64'): 5e-5,
    ('h_q64_fm_blk120', 'i8'): 5e-6, ('h_q32_fm_blk120', 'i8'): 1e-5, ('h_q64_fm_blk120_vfo2', 'i8'): 5e-6,
}


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_golden_parity(case, golden, stage_a):
    z, _ = golden
    name = case['name']
    if case['code'] == 'f':
        # float32 records: the reference's arithmetic is complex64 (read_file.py:100-101, dsp_processor.py:147); the mode that
        # reproduces it is held to these vectors at the tolerance of every other encoding.  (The default float32 path computes in
        # float64 / exactly and is pinned against the float64 restatements: test_float32_input_against_the_fp64_oracle; how far
        # that is from the reference's complex64 results: test_default_float32_path_deviation_from_the_reference.)
        if stage_a != 'auto':
            pytest.skip('float32 records: one faithful kernel')
        eng = _engine(case, faithful_f32=True)
        assert eng.stage_a == 'f32c'
    else:
        eng = _engine(case, stage_a=stage_a if stage_a in ('auto', 'i8', 'f64') else 'auto')
    if case['code'] == 'f':
        pass
    elif stage_a == 'f64':
        assert eng.stage_a in ('f64', 'seq')         # decimation < 8 always takes the per-sample fp64 form
    elif stage_a in ('auto', 'i8') and eng.stage_a not in ('i8w', 'i8t', 'sq'):
        pytest.skip(f'{stage_a}: the library uses {eng.stage_a} for this case (covered by another run)')
    elif stage_a not in ('auto', 'i8') and eng.stage_a != stage_a:
        pytest.skip(f'{stage_a}: the library uses {eng.stage_a} for this case (covered by another run)')
    raw = case_raw(case)
    out, y = eng.process_debug(raw)
    if stage_a == 'auto' and case['code'] != 'f' and not case.get('correct_iq'):
        # the outputs of the production entry point (no debug buffer: pair products between the stages where the kernels have
        # them, and the second look of stage_a = auto); the decimated samples checked below are the debug call's
        # (--correct-iq: the offset carries from call to call, so that case keeps the debug call's outputs)
        out = _engine(case, stage_a='auto').process(raw)
    exp = z[f'{name}/z']                       # (chunks, N, m)
    nch, rows, m = exp.shape
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    y0 = z[f'{name}/y0'].astype(np.complex128)
    gy0 = y.reshape(rows, nch, m)[:, 0, :]
    peak = float(np.abs(y0).max())
    yerr = np.abs(gy0 - y0).max() / (peak if peak > 0 else 1.0)
    if peak == 0:
        assert float(np.abs(gy0).max()) == 0.0
    err = np.abs(got - exp).max()
    tol = tol_for(exp, case['mode'])
    print(f'{name} [{eng.stage_a}]: decimated rel err {yerr:.3e}, output max abs err {err:.3e} (tol {tol:.1e})')
    assert yerr <= (0.0 if case['code'] == 'f' else 1e-9)
    limit = STATED_LIMITS.get((name, stage_a))
    if limit is not None:
        assert err <= limit, f'{name} [{eng.stage_a}]: {err:.3e} is worse than the stated limit {limit:.1e}'
        assert err > tol, f'{name} [{eng.stage_a}]: {err:.3e} is now within the tolerance -- drop it from STATED_LIMITS'
        pytest.xfail(f'stated limit: {err:.3e} (tolerance {tol:.1e}, bound {limit:.1e})')
    assert err <= tol


def test_auto_looks_twice_where_the_int8_image_ends(golden):
    """sdrt_config.stage_a = auto (the default): the demodulation stage lists every (chunk, row) item whose LARGEST pair product
    lies under (2^-15 of the encoding's full scale)^2 and is not exactly zero; those items go through unpack .. decimate again
    on the fp64 matrix cores and through the demodulation stage again.  h_q64_fm_blk120 -- in-band content 120 dB under a
    full-scale tone, a thirtieth of an LSB -- FAILS the tolerance on the int8 form alone and PASSES on auto; a stream of ordinary
    level is never listed and comes out bit for bit as without the second look; silence is exact on every path and not listed
    either.  (dsp_processor.py:140-149 is the one arithmetic the reference has.)"""
    z, _ = golden
    case = BY_NAME['h_q64_fm_blk120']
    exp = z['h_q64_fm_blk120/z']
    nch, rows, m = exp.shape
    errs = {}
    for st in ('i8', 'auto'):
        eng = _engine(case, stage_a=st)
        assert eng.stage_a == 'i8w'
        out = eng.process(case_raw(case))
        errs[st] = float(np.abs(out.reshape(rows, nch, m).transpose(1, 0, 2) - exp).max())
        refined = eng.refined_items
        print(f'h_q64_fm_blk120 [{st}]: max abs err {errs[st]:.3e}, items done again in float64: {refined}')
        assert refined == (nch * rows if st == 'auto' else 0)
    assert errs['i8'] > 1e-6 and errs['auto'] <= 1e-6
    for name in ('h_q64_fm_bench', 'h_q64_fm_zero_c', 'B_q64_fm_tone_1m', 'h_q32_fm_c', 'h_q64_fm_blk80'):
        c2 = BY_NAME[name]
        a, b = _engine(c2, stage_a='i8'), _engine(c2, stage_a='auto')
        oa, ob = a.process(case_raw(c2)), b.process(case_raw(c2))
        assert b.refined_items == 0, name
        np.testing.assert_array_equal(oa, ob)
    # a batch in which only some items are listed: the loud chunks of another case around the weak ones
    loud, weak = case_raw(BY_NAME['h_q64_fm_c']), case_raw(case)
    cb = len(weak) // nch
    mixed = loud[:cb] + weak[:cb] + loud[cb:2 * cb] + weak[cb:2 * cb]
    e_i8, e_auto, e_f64 = (_engine(case, stage_a=st) for st in ('i8', 'auto', 'f64'))
    o_i8, o_auto, o_f64 = (e.process(mixed).reshape(4, m) for e in (e_i8, e_auto, e_f64))
    assert e_auto.refined_items == 2
    np.testing.assert_array_equal(o_auto[[0, 2]], o_i8[[0, 2]])            # the loud chunks: untouched
    assert np.abs(o_auto[[1, 3]] - o_f64[[1, 3]]).max() <= 1e-9            # the weak ones: the float64 result
    assert np.abs(o_auto[[1, 3]] - exp[:2, 0]).max() <= 1e-6


def test_default_float32_path_deviation_from_the_reference(golden):
    """A stated deviation, measured: with faithful_f32 off, float32 records are filtered in float64 / exactly, which is NOT what
    the reference computes (complex64: ~3e-4 of single-precision noise on its decimated samples).  The default path therefore
    differs from the reference's vectors by the reference's own rounding noise -- bounded here so that it cannot grow unnoticed;
    identity with the reference is what faithful_f32 is for (tests/test_gpu_faithful_f32.py)."""
    z, _ = golden
    for case in (c for c in CASES if c['code'] == 'f'):
        eng = _engine(case)
        out, y = eng.process_debug(case_raw(case))
        exp = z[f"{case['name']}/z"]
        nch, rows, m = exp.shape
        got = out.reshape(rows, nch, m).transpose(1, 0, 2)
        y0 = z[f"{case['name']}/y0"].astype(np.complex128)
        yerr = np.abs(y.reshape(rows, nch, m)[:, 0, :] - y0).max() / max(float(np.abs(y0).max()), 1e-300)
        err = np.abs(got - exp).max() / max(1.0, float(np.abs(exp).max()))
        print(f"{case['name']} [default {eng.stage_a}]: decimated samples {yerr:.2e} of the peak from the reference's complex64 ones, outputs {err:.2e}")
        assert yerr <= 1e-3 and err <= 5e-3


def _oracle_rows(raw, code, n, q, mode, freqs, fs, omega_out, chunks, rows_sel):
    """oracle/chain.c (fp64 C restatement of the reference chain) on selected rows / chunks of a stream: (chunk, row, m)."""
    from oracle import c_chain, ref_chain
    item = {'b': 1, 'B': 1, 'h': 2, 'H': 2, 'i': 4, 'I': 4, 'f': 4, 'd': 8}[code]
    cb = n * 2 * item
    shift = c_chain.shift_table([freqs[r] for r in rows_sel], fs, n)
    buf = b''.join(raw[c * cb:(c + 1) * cb] for c in chunks)
    exp, _ = c_chain.process_stream(buf, code, n, q, mode, shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, omega_out), threads=4)
    return exp


@pytest.mark.parametrize('name', ['f_q64_fm_c', 'f_q32_am_c0'])
def test_float32_input_against_the_fp64_oracle(name, golden):
    """float32 records: the reference itself runs the decimator in complex64 (read_file.py:100-101 -> complex64,
    dsp_processor.py:172-177), so its vectors only bound this build at 5e-3 (test_golden_parity).  The build computes every
    encoding in fp64; that arithmetic is pinned here at the north-star tolerance against the fp64 restatements of the same
    chain: oracle.ref_chain.process_chunk(faithful_dtype=False) and oracle/chain.c -- 1e-6 max-abs (FM), 1e-6 of the peak
    (AM), decimated samples 1e-9 relative."""
    from oracle import ref_chain
    case = BY_NAME[name]
    eng = _engine(case)
    raw = case_raw(case)
    out, y = eng.process_debug(raw)
    n, q = samples_per_chunk('f'), case['q']
    m = n // q
    ch = ref_chain.Chain(case['fs'], q, case['mode'], 'f', center=case['center'], omega_out=case['omega_out'])
    nch = case['n_chunks']
    got = out.reshape(1, nch, m)
    gy = y.reshape(1, nch, m)
    for c in range(nch):
        exp, ydec = ch.process(raw[c * n * 8:(c + 1) * n * 8], return_decimated=True, faithful_dtype=False)
        yerr = np.abs(gy[:, c] - ydec).max() / np.abs(ydec).max()
        err = np.abs(got[:, c] - exp).max()
        print(f'{name} chunk {c}: decimated rel err {yerr:.3e}, output max abs err {err:.3e}')
        assert yerr <= 1e-9
        assert err <= tol_for(exp, case['mode'])
    freqs = [float(case['center'])]
    expc = _oracle_rows(raw, 'f', n, q, case['mode'], freqs, case['fs'], case['omega_out'], range(nch), [0])
    assert np.abs(got.transpose(1, 0, 2) - expc).max() <= tol_for(expc, case['mode'])


def test_float32_nine_rows_against_the_fp64_oracle():
    """9 VFO rows of float32 input (the --simo layout: csv rows + center last) at 1e-6 against oracle/chain.c."""
    from sdrterm_amd.engine import Engine
    from oracle import ref_chain
    from tests.golden import prng
    fs, q, n, nch = 1024000, 64, samples_per_chunk('f'), 2
    freqs = [float(f) for f in ref_chain.vfo_rows(15700, '-200000,-150000,-100000,-50000,50000,100000,150000,200000')]
    raw = prng.raw_bytes('f', n * nch, seed=77, amp=3000)
    out = Engine(fs, q, 'fm', 'f', row_freqs=freqs, apply_shift=True, omega_out=5000).process(raw)
    exp = _oracle_rows(raw, 'f', n, q, 'fm', freqs, fs, 5000, range(nch), range(9))
    got = out.reshape(9, nch, n // q).transpose(1, 0, 2)
    err = np.abs(got - exp).max()
    print(f'float32 x 9 rows: max abs err {err:.3e}')
    assert err <= 1e-6


def test_float32_levels_that_jump_between_tiles_and_chunks(monkeypatch):
    """float32 on the wide-column kernel is block fixed point against each 4096-sample tile's exponent (rounded up to a multiple
    of four), sliced against a guess (the exponent the same wave's previous tile used) and redone when the guess was wrong.
    A stream whose level jumps by up to 2^30 between chunks and 2^6 between the tiles of a chunk, with silent tiles, silent
    chunks and tiles whose only large sample is their last one, long enough that every workgroup works through several
    items.  Block fixed point carries 27 to 30 bits below the tile's largest sample, and the decimator is a weighted mean, so
    every decimated sample must agree with the fp64 oracle to 2^-27 of the largest input sample of its tile and the two next
    to it; a stream of even level stays 1e-9 of the peak (test_float32_input_against_the_fp64_oracle).  The same stream in
    another batching (its second half alone) must give the same bits: the exponent in use depends on the tile only."""
    from sdrterm_amd.engine import Engine
    from oracle import ref_chain
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, q, n, nch = 1024000, 64, 16384, 1101
    rng = np.random.default_rng(2026)
    x = rng.normal(size=(nch, 4, 4096, 2)).astype(np.float32)
    lvl = 2.0 ** (rng.integers(-20, 11, size=(nch, 1, 1, 1)) + rng.integers(0, 7, size=(nch, 4, 1, 1)))
    x *= lvl.astype(np.float32)
    silent_tile = rng.random((nch, 4)) < 0.05
    x[silent_tile] = 0
    x[rng.random(nch) < 0.03] = 0
    spike = rng.random((nch, 4)) < 0.1
    x[spike, -1, :] *= np.float32(2.0 ** 9)              # the tile's exponent is decided by its last sample
    raw = x.tobytes()
    eng = Engine(fs, q, 'fm', 'f', n_samples=n, row_freqs=[15700.0], apply_shift=True, omega_out=5000)
    assert eng.stage_a == 'i8w'
    _out, y = eng.process_debug(raw)
    m = n // q
    y = y.reshape(nch, 4, m // 4)
    top = np.abs(x).max(axis=(2, 3))                     # (chunk, tile)
    scale = top.copy()
    scale[:, 1:] = np.maximum(scale[:, 1:], top[:, :-1])
    scale[:, :-1] = np.maximum(scale[:, :-1], top[:, 1:])
    ch = ref_chain.Chain(fs, q, 'fm', 'f', center=15700, omega_out=5000)
    worst = 0.0
    for c in range(nch):
        _exp, ydec = ch.process(raw[c * n * 8:(c + 1) * n * 8], return_decimated=True, faithful_dtype=False)
        err = np.abs(y[c] - np.asarray(ydec).reshape(4, m // 4)).max(axis=1)
        ok = scale[c] > 0
        if ok.any():
            worst = max(worst, float((err[ok] / scale[c][ok]).max()))
        # silent surroundings: the far tail of a tile further away, or (silent chunk) what is left of the constant the slices leave out
        assert np.all(err[~ok] <= 1e-12 * top[c].max() + 1e-40)
    print(f'float32 with jumping levels: worst error relative to the largest input sample nearby {worst:.3e} (2^-27 = {2.0 ** -27:.3e})')
    assert worst <= 2.0 ** -27
    half = 550
    _o2, y2 = eng.process_debug(raw[half * n * 8:])
    assert np.array_equal(y2.reshape(nch - half, 4, m // 4), y[half:])


_OPTION_MATRIX = [
    # code, q, mode, options -- combinations the golden cases do not hold, against oracle/chain.c
    ('f', 64, 'fm', dict(big_endian=True)),
    ('f', 32, 'am', dict(big_endian=True, normalize=True)),
    ('f', 64, 'am', dict(normalize=True)),
    ('f', 64, 'fm', dict(correct_iq=True)),
    ('f', 32, 'fm', dict(correct_iq=True, normalize=True, big_endian=True)),
    ('H', 64, 'fm', dict(big_endian=True, correct_iq=True)),
    ('B', 32, 'am', dict(normalize=True, correct_iq=True)),
    ('b', 64, 'fm', dict(correct_iq=True)),
    ('h', 32, 'fm', dict(big_endian=True, normalize=True, correct_iq=True)),
    ('i', 64, 'fm', dict(big_endian=True, correct_iq=True)),
    ('d', 32, 'am', dict(big_endian=True, normalize=True)),
]


@pytest.mark.parametrize('code,q,mode,opt', _OPTION_MATRIX, ids=[f"{c}_q{q}_{m}_{'_'.join(sorted(o))}" for c, q, m, o in _OPTION_MATRIX])
def test_reader_options_against_the_c_oracle(code, q, mode, opt, monkeypatch):
    """Byte order, --normalize-input and --correct-iq in combinations (and on encodings) the golden cases leave out: two rows,
    three chunks with a carried offset, against oracle/chain.c on the same bytes."""
    from sdrterm_amd.engine import Engine
    from oracle import c_chain, ref_chain
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, nch = 1024000, 3
    n = samples_per_chunk(code)
    freqs = [15700.0, -203300.0]
    amp = 100 if code in 'bB' else (100000 if code in 'iI' else 3000)
    raw = prng.raw_bytes(code, n * nch, seed=61, amp=amp, dc=(amp // 6, -amp // 10), big_endian=bool(opt.get('big_endian')))
    eng = Engine(fs, q, mode, code, row_freqs=freqs, apply_shift=True, omega_out=5000, normalize=bool(opt.get('normalize')),
                 correct_iq=bool(opt.get('correct_iq')), big_endian=bool(opt.get('big_endian')))
    out = eng.process(raw)
    shift = c_chain.shift_table(freqs, fs, n)
    norm = ref_chain.generate_domain(code) if opt.get('normalize') else None
    exp, off = c_chain.process_stream(raw, code, n, q, mode, shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, 5000),
                                      big_endian=bool(opt.get('big_endian')), norm=norm,
                                      iq_inductance=50 / fs if opt.get('correct_iq') else None, threads=4)
    got = out.reshape(2, nch, n // q).transpose(1, 0, 2)
    err = np.abs(got - exp).max()
    print(f'{code} q={q} {mode} {sorted(opt)} [{eng.stage_a}]: max abs err {err:.3e} (peak {np.abs(exp).max():.3e})')
    assert err <= tol_for(exp, mode)
    if opt.get('correct_iq'):
        assert abs(eng.iq_offset - off) <= 1e-9 * max(abs(off), 1e-30)


@pytest.mark.parametrize('code,q,rows', [('h', 64, 1), ('h', 32, 3), ('B', 64, 1), ('f', 64, 2), ('h', 16, 2), ('h', 8, 1), ('B', 16, 1), ('h', 4, 1)])
def test_pair_products_between_the_stages_equal_the_sample_path(code, q, rows, monkeypatch):
    """FM without a debug buffer: the wide-column kernel hands the demodulation stage the products y[2i] conj(y[2i+1])
    (demodulation.py:25-32) instead of the decimated samples.  Same outputs as the path that writes the samples (taken when
    the caller asks for them), to rounding."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, nch = 1024000, 5
    n = samples_per_chunk(code)
    freqs = [15700.0, -203300.0, 100100.0][:rows]
    raw = prng.raw_bytes(code, n * nch, seed=77, amp=100 if code == 'B' else 3000)
    # (decimation 4 .. 16: several passes of the wide kernel over the chunk, each finishing whole pairs; 16-bit records at decimation
    # 4 .. 16: the one-pass kernel, whose lanes hold both outputs of a pair)
    eng = Engine(fs, q, 'fm', code, row_freqs=freqs, apply_shift=True, omega_out=5000 if q >= 32 else 8000 * (32 // q))
    assert eng.stage_a == ('sq' if code in 'hHbB' and q <= 16 else 'i8w')
    a = eng.process(raw)
    b, _y = eng.process_debug(raw)
    err = np.abs(a - b).max()
    print(f'{code} q={q} rows={rows}: pair path against sample path, max abs difference {err:.3e}')
    assert err <= 1e-12


def test_float32_non_finite_samples_stay_in_their_chunk(monkeypatch):
    """A NaN or an infinity in a float32 record: the reference's IIR passes turn that whole chunk into NaN
    (dsp_processor.py:147); block fixed point has no NaN, the chunk's outputs are unspecified here -- but every other chunk,
    the one sharing its work item included, must come out bit for bit as from the clean stream."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, q, n, nch = 1024000, 64, 16384, 6
    raw = prng.raw_bytes('f', n * nch, seed=91, amp=3000)
    x = np.frombuffer(raw, dtype=np.float32).copy().reshape(nch, n, 2)
    bad = x.copy()
    bad[2, 5000, 0] = np.nan
    bad[5, 16383, 1] = np.inf
    eng = Engine(fs, q, 'fm', 'f', row_freqs=[15700.0], apply_shift=True, omega_out=5000)
    assert eng.stage_a == 'i8w'
    m = n // q
    clean = eng.process(x.tobytes()).reshape(nch, m)
    got = eng.process(bad.tobytes()).reshape(nch, m)
    again = eng.process(x.tobytes()).reshape(nch, m)
    for c in (0, 1, 3, 4):
        assert np.array_equal(got[c], clean[c]), f'chunk {c}'
    assert np.array_equal(again, clean)                  # and the exponent guesses carry no damage into the next call


def test_float32_256_rows_sampled_against_the_fp64_oracle():
    """BASELINE.json configs[4]'s shape on one GPU: 256 VFO rows of float32 records, decimation 64.  Rows sampled across the
    range (first, last = center, and some in between) x 3 chunks against oracle/chain.c at the north-star tolerance."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    fs, q, n, nch = 1_000_000_000, 64, samples_per_chunk('f'), 3
    freqs = [float((i - 127) * 3_000_000 + 15000) for i in range(255)] + [15000.0]
    raw = prng.raw_bytes('f', n * nch, seed=256, amp=8000)
    eng = Engine(fs, q, 'fm', 'f', row_freqs=freqs, apply_shift=True, omega_out=12500)
    out = eng.process(raw)
    pick = [0, 1, 63, 64, 127, 128, 200, 254, 255]
    exp = _oracle_rows(raw, 'f', n, q, 'fm', freqs, fs, 12500, range(nch), pick)
    got = out.reshape(256, nch, n // q)[pick].transpose(1, 0, 2)
    err = np.abs(got - exp).max()
    print(f'float32 x 256 rows [{eng.stage_a}]: max abs err over {len(pick)} sampled rows {err:.3e}')
    assert err <= 1e-6


@pytest.mark.parametrize('rows,nch,q', [(64, 3, 64), (64, 3, 32), (300, 2, 64), (33, 17, 64), (5, 100, 32)])
def test_item_distribution_of_the_persistent_kernel(rows, nch, q, monkeypatch):
    """Many rows x few chunks and few rows x many chunks (rows above and below the number of workgroup slots, chunk counts
    that are not multiples of 8): the wide-column persistent kernel must cover every (chunk, row) item exactly once --
    checked against the independently written one-block-per-column kernel.  (64 rows at decimation 32 is the per-GPU shape
    of bench.py's simo64 workload.)"""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    fs = 1024000
    freqs = [1000.0 * (i - rows // 2) for i in range(rows)]
    raw = prng.raw_bytes('h', 32768 * nch, seed=rows, amp=5000)
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    a = Engine(fs, q, 'fm', 'h', row_freqs=freqs, omega_out=5000)
    out_a = a.process(raw)
    monkeypatch.setenv('SDRTERM_STAGE_A', 'i8t')
    b = Engine(fs, q, 'fm', 'h', row_freqs=freqs, omega_out=5000)
    out_b = b.process(raw)
    assert (a.stage_a, b.stage_a) == ('i8w', 'i8t')
    assert np.isfinite(out_a).all() and np.abs(out_a - out_b).max() < 1e-7     # (two kernels, white-noise FM: each is held to the oracle below)
    # ... and sampled (chunk, row) items against the oracle (oracle/chain.c): first, last and two rows in between, on
    # the first, a middle and the last chunk
    rs = sorted({0, rows // 3, (2 * rows) // 3, rows - 1})
    cs = sorted({0, nch // 2, nch - 1})
    exp = _oracle_rows(raw, 'h', 32768, q, 'fm', freqs, fs, 5000, cs, rs)          # (len(cs), len(rs), m)
    m = 32768 // q
    got = out_a.reshape(rows, nch, m)[np.ix_(rs, cs)].transpose(1, 0, 2)
    err = np.abs(got - exp).max()
    print(f'{rows} rows x {nch} chunks: sampled items vs oracle max abs err {err:.3e}')
    assert err <= 1e-6


@pytest.mark.parametrize('name', ['h_q64_fm_c', 'h_q32_fm_c', 'h_q64_fm_c0'])
def test_resampler_forms_agree(name, golden, monkeypatch):
    """scipy.signal.resample of the FM discriminator (demodulation.py:35-38) in its two GPU forms -- in-wave FFT (default)
    and the circulant product with the half-sample Dirichlet kernel -- and the general vs real-stream form of the wide kernel:
    every combination meets the reference's vectors."""
    z, _ = golden
    case = BY_NAME[name]
    exp = z[f'{name}/z']
    nch, rows, m = exp.shape
    outs = []
    for resample, rs in (('fft', '1'), ('direct', '1'), ('fft', '0')):
        monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
        _diag(monkeypatch, resample=resample)
        _diag(monkeypatch, wide_rs=rs)
        out, _ = _engine(case).process_debug(case_raw(case))
        got = out.reshape(rows, nch, m).transpose(1, 0, 2)
        assert np.abs(got - exp).max() <= tol_for(exp, case['mode'])
        outs.append(got)
    assert np.abs(outs[0] - outs[1]).max() < 1e-9 and np.abs(outs[0] - outs[2]).max() < 1e-9


@pytest.mark.parametrize('name,fused', [('h_q64_fm_iq', '0'), ('h_q32_fm_vfo9_iq', '1'), ('h_q32_am_norm_vfo9_iq', '1')])
def test_both_forms_of_the_iq_correction_on_the_wide_kernel(name, fused, golden, monkeypatch):
    """--correct-iq on the wide-column kernel: offsets from the prefix pass over the input (default for several rows) or from the
    kernel's own matrix rows (default for one row).  The golden sweep runs every case in its default form; here each runs in
    the other one (three chunks, offset carried across them)."""
    z, _ = golden
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    _diag(monkeypatch, iq_fused=fused)
    case = BY_NAME[name]
    eng = _engine(case)
    assert eng.stage_a == 'i8w'
    out, _ = eng.process_debug(case_raw(case))
    exp = z[f'{name}/z']
    nch, rows, m = exp.shape
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    assert np.abs(got - exp).max() <= tol_for(exp, case['mode'])


@pytest.mark.parametrize('code,q,nch', [('h', 64, 6), ('h', 64, 5), ('h', 32, 1), ('f', 64, 7), ('f', 32, 4), ('f', 64, 1)])
def test_two_chunks_per_item_on_the_wide_kernel(code, q, nch, monkeypatch):
    """Chunks of four tiles (float32 records; 16-bit records in 16384-sample chunks) go two to an item of the wide-column
    kernel: carries, edge operators and column phasors per chunk.  Even and odd chunk counts and a single chunk, two rows,
    against oracle/chain.c."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, n = 1024000, 16384
    freqs = [15700.0, -203300.0]
    raw = prng.raw_bytes(code, n * nch, seed=40 + nch, amp=3000)
    eng = Engine(fs, q, 'fm', code, n_samples=n, row_freqs=freqs, apply_shift=True, omega_out=5000)
    assert eng.stage_a == 'i8w'
    out = eng.process(raw)
    exp = _oracle_rows(raw, code, n, q, 'fm', freqs, fs, 5000, range(nch), range(2))
    got = out.reshape(2, nch, n // q).transpose(1, 0, 2)
    err = np.abs(got - exp).max()
    print(f'{code} q={q} {nch} chunk(s): max abs err {err:.3e}')
    assert err <= 1e-6


@pytest.mark.parametrize('fused', ['0', '1', None])
def test_per_row_modes_equal_the_two_single_mode_runs(golden, monkeypatch, fused):
    """BASELINE configs[2] asks for FM and AM channels in one --simo run; the reference's demodulation is processor-wide
    (io_args.py:118), so its own answer is two runs.  A context with per-row modes (sdrt_config.row_modes, one pass over the
    input) must give, row by row, what the reference gave in its FM run and in its AM run of the same nine rows (golden cases
    h_q32_fm_norm_vfo9_iq / h_q32_am_norm_vfo9_iq: same seed, same --normalize-input and --correct-iq)."""
    from sdrterm_amd.engine import Engine
    from oracle import ref_chain
    z, _ = golden
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    if fused is not None:
        _diag(monkeypatch, iq_fused=fused)
    fm, am = BY_NAME['h_q32_fm_norm_vfo9_iq'], BY_NAME['h_q32_am_norm_vfo9_iq']
    assert case_raw(fm) == case_raw(am)
    rows = ref_chain.vfo_rows(fm['center'], fm['vfos'])
    modes = ['fm', 'am'] * 4 + ['fm']
    eng = Engine(fm['fs'], fm['q'], 'fm', 'h', row_freqs=rows, apply_shift=True, omega_out=fm['omega_out'], normalize=True,
                 correct_iq=True, row_modes=modes)
    out = eng.process(case_raw(fm))
    exp = {'fm': z['h_q32_fm_norm_vfo9_iq/z'], 'am': z['h_q32_am_norm_vfo9_iq/z']}      # (chunks, rows, m)
    nch, nrows, m = exp['fm'].shape
    got = out.reshape(nrows, nch, m)
    for r, md in enumerate(modes):
        e = exp[md][:, r, :]
        err = np.abs(got[r] - e).max()
        assert err <= tol_for(e, md), f'row {r} ({md}): {err:.3e}'
    # all rows alike: the plain context
    same = Engine(fm['fs'], fm['q'], 'am', 'h', row_freqs=rows, apply_shift=True, omega_out=fm['omega_out'], normalize=True,
                  correct_iq=True, row_modes=['fm'] * 9).process(case_raw(fm)).reshape(nrows, nch, m)
    assert np.abs(same.transpose(1, 0, 2) - exp['fm']).max() <= ABS_TOL


def test_per_row_modes_on_the_bench_configuration(golden, monkeypatch):
    """The same at configs[2]'s own parameters (256 MS/s, 8 rows, decimation 32, --correct-iq): FM rows against the FM golden
    run; the AM rows (the AM golden run normalises its input, this context does not) against the C oracle."""
    from sdrterm_amd.engine import Engine
    from oracle import ref_chain
    z, _ = golden
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    case = BY_NAME['h_q32_fm_vfo8_iq_bench']
    rows = [float(f) for f in ref_chain.vfo_rows(case['center'], case['vfos'])]
    modes = ['fm', 'am', 'am', 'fm', 'fm', 'am', 'fm', 'am']
    raw = case_raw(case)
    out = Engine(case['fs'], 32, 'fm', 'h', row_freqs=rows, apply_shift=True, omega_out=12500, correct_iq=True, row_modes=modes).process(raw)
    exp = z['h_q32_fm_vfo8_iq_bench/z']
    nch, nrows, m = exp.shape
    got = out.reshape(nrows, nch, m)
    n = samples_per_chunk('h')
    from oracle import c_chain
    shift = c_chain.shift_table(rows, case['fs'], n)
    am, _ = c_chain.process_stream(raw, 'h', n, 32, 'am', shift, ref_chain.cheby_sos(32), ref_chain.ellip_sos(case['fs'] // 32, 12500),
                                   iq_inductance=50 / case['fs'], threads=4)
    for r, md in enumerate(modes):
        e = exp[:, r, :] if md == 'fm' else am[:, r, :]
        assert np.abs(got[r] - e).max() <= tol_for(e, md), f'row {r} ({md})'


@pytest.mark.parametrize('big_endian,nch', [(False, 5), (True, 2), (False, 4)])
def test_float32_rows_sliced_once_per_chunk_give_the_same_bits(big_endian, nch, monkeypatch):
    """float32 records with several rows: from four rows up the records are sliced into their int8 planes once per chunk
    (f32_preslice_kernel) instead of once per (chunk, row) inside the chain kernel.  Same exponent rule, same slicing code:
    the outputs must be bit-identical to the per-row form (odd and even chunk counts, either byte order), and meet the oracle."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, q, n = 1024000, 64, 16384
    freqs = [15700.0, -203300.0, 100100.0, 5.0, -77000.0]
    raw = prng.raw_bytes('f', n * nch, seed=300 + nch, amp=3000, big_endian=big_endian)
    outs = {}
    for pre in ('0', '1'):
        _diag(monkeypatch, f32_pre=pre)
        eng = Engine(fs, q, 'fm', 'f', row_freqs=freqs, apply_shift=True, omega_out=5000, big_endian=big_endian)
        assert eng.stage_a == 'i8w'
        outs[pre] = eng.process(raw)
    assert np.array_equal(outs['0'], outs['1'])
    _diag(monkeypatch, f32_pre=None)
    auto = Engine(fs, q, 'fm', 'f', row_freqs=freqs, apply_shift=True, omega_out=5000, big_endian=big_endian).process(raw)
    assert np.array_equal(auto, outs['1'])                       # five rows: the pre-sliced form is the default
    if not big_endian:
        exp = _oracle_rows(raw, 'f', n, q, 'fm', freqs, fs, 5000, range(nch), range(5))
        assert np.abs(outs['1'].reshape(5, nch, n // q).transpose(1, 0, 2) - exp).max() <= 1e-6


@pytest.mark.parametrize('code,rows,iq,fused,q', [('B', 1, False, None, 64), ('b', 2, False, None, 64), ('B', 1, True, '1', 64), ('b', 1, True, '0', 64),
                                                  ('B', 2, True, None, 64), ('h', 1, False, None, 64), ('B', 2, False, None, 32), ('b', 2, True, None, 32),
                                                  ('h', 1, True, '1', 64), ('h', 2, True, None, 64), ('h', 1, True, '0', 32), ('H', 1, True, '1', 64)])
def test_sixteen_tiles_per_chunk_on_the_wide_kernel(code, rows, iq, fused, q, monkeypatch):
    """Chunks of sixteen tiles (8-bit records; 16-bit records in 65536-sample chunks): two tiles per wave, the column phase of a
    tile running in the pass after its matrix phase (kernels_i8w.hip, SHQ), with and without --correct-iq in both of its forms,
    four and eight outputs per column (decimation 64 / 32).
    Five chunks against oracle/chain.c, offsets carried; then enough chunks that every workgroup reaches the steady state of its
    pipeline (more than four items each) against the independently written one-block-per-column kernel."""
    from sdrterm_amd.engine import Engine
    from oracle import c_chain, ref_chain
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    if fused is not None:
        _diag(monkeypatch, iq_fused=fused)
    fs, n = 1024000, 65536
    freqs = [15700.0, -203300.0][:rows]
    amp = 100 if code in 'bB' else 3000
    kw = dict(n_samples=n, row_freqs=freqs, apply_shift=True, omega_out=5000, correct_iq=iq)
    raw = prng.raw_bytes(code, n * 5, seed=77, amp=amp, dc=(amp // 6, -amp // 10))
    eng = Engine(fs, q, 'fm', code, **kw)
    assert eng.stage_a == 'i8w'
    out = eng.process(raw)
    exp, off = c_chain.process_stream(raw, code, n, q, 'fm', c_chain.shift_table(freqs, fs, n), ref_chain.cheby_sos(q),
                                      ref_chain.ellip_sos(fs // q, 5000), iq_inductance=50 / fs if iq else None, threads=4)
    got = out.reshape(rows, 5, n // q).transpose(1, 0, 2)
    err = np.abs(got - exp).max()
    print(f'{code} x {rows} iq={iq} fused={fused}: 5 chunks vs oracle max abs err {err:.3e}')
    assert err <= 1e-6
    if iq:
        assert abs(eng.iq_offset - off) <= 1e-9 * max(abs(off), 1e-30)
    # the pipeline at every depth: 300 ... 1300 chunks = one to six items per workgroup (256 of them on an MI355X), the last count
    # with every workgroup in its steady state
    block = np.frombuffer(prng.raw_bytes(code, n * 50, seed=78, amp=amp, dc=(amp // 5, amp // 9)), dtype=np.uint8)
    for nch in (300, 550, 800, 1050, 1300):
        big = np.tile(block, nch // 50)
        monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
        a = Engine(fs, q, 'fm', code, **kw)
        out_a = a.process(big)
        monkeypatch.setenv('SDRTERM_STAGE_A', 'i8t')
        b = Engine(fs, q, 'fm', code, **kw)
        out_b = b.process(big)
        assert a.stage_a == 'i8w' and b.stage_a in ('i8t', 'f64')      # (fused --correct-iq exists on the wide and the fp64 kernel)
        d = np.abs(out_a - out_b).max()
        print(f'   {nch} chunks: wide against the {b.stage_a} kernel {d:.3e}')
        assert np.isfinite(out_a).all() and d < 1e-7
        if iq:
            assert abs(a.iq_offset - b.iq_offset) <= 1e-9 * max(abs(b.iq_offset), 1e-30)


@pytest.mark.parametrize('name', ['h_q64_fm_c', 'h_q64_fm_bench', 'h_q64_fm_vfo9', 'h_q64_am_norm_c'])
def test_one_wave_per_simd_form_of_the_wide_kernel(name, golden, monkeypatch):
    """SDRTERM_DIAG=wide_dual=1: the experimental form of chain_i8w_kernel with one wave per SIMD working on two tiles at once (every
    digit-plane fragment read once for two matrix instructions) meets the reference's vectors like the shipped
    form, and agrees with it to rounding."""
    z, _ = golden
    if name not in BY_NAME:
        pytest.skip('no such golden case')
    case = BY_NAME[name]
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    ref_out, _ = _engine(case).process_debug(case_raw(case))
    _diag(monkeypatch, wide_dual='1')
    eng = _engine(case)
    assert eng.stage_a == 'i8w'
    out, _ = eng.process_debug(case_raw(case))
    exp = z[f'{name}/z']
    nch, rows, m = exp.shape
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    assert np.abs(got - exp).max() <= tol_for(exp, case['mode'])
    assert np.abs(out - ref_out).max() < 1e-9


@pytest.mark.parametrize('code,q,center,rows', [('h', 64, 15700, 1), ('h', 64, 0, 1), ('B', 64, 15700, 1), ('H', 64, 15700, 1),
                                                 ('h', 32, 15700, 3), ('B', 32, 0, 1), ('b', 64, 15700, 1), ('f', 64, 15700, 1),
                                                 ('f', 64, 15700, 5), ('h', 128, 15700, 1), ('f', 128, 15700, 1),
                                                 ('h', 16, 15700, 1), ('h', 8, 0, 2), ('B', 16, 15700, 1)])
def test_silent_chunks_anywhere_in_a_batch_come_out_as_exact_zeros(code, q, center, rows, monkeypatch):
    """The reference turns an all-zero chunk into exact zeros (sosfiltfilt of zeros, angle(0) = 0: demodulation.py:30-32,
    dsp_processor.py:147).  The int8 forms filter x - offset and add the response to the offset back: that response is taken
    from the kernel itself (api.hip: calibrate_zero_response), so a silent chunk cancels to the last bit wherever it sits in a
    batch (every workgroup in its steady state, first and last items, odd batch sizes), in FM and in AM."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, n, nch = 1024000, samples_per_chunk(code), 1301
    item = {'b': 1, 'B': 1, 'h': 2, 'H': 2, 'f': 4}[code]
    freqs = [float(center + 50000 * r) for r in range(rows)]
    one = np.frombuffer(prng.raw_bytes(code, n * 4, seed=91, amp=100 if item == 1 else 8000), dtype=np.uint8).reshape(4, -1)
    raw = np.tile(one, (nch // 4 + 1, 1))[:nch].copy()
    silent = np.zeros(nch, dtype=bool)
    silent[[0, 1, 5, 255, 256, 257, 511, 640, 1023, 1024, 1290, 1299, 1300]] = True
    silent[np.arange(300, 1200, 7)] = True
    raw[silent] = 0
    for mode in ('fm', 'am'):
        eng = Engine(fs, q, mode, code, row_freqs=freqs, apply_shift=bool(center or rows > 1), omega_out=5000 if q < 128 else 2500)
        assert eng.stage_a in ('i8w', 'i8t', 'sq')
        out = eng.process(raw.tobytes()).reshape(rows, nch, n // q)
        worst = float(np.abs(out[:, silent]).max())
        loud = float(np.abs(out[:, ~silent]).min(axis=2).max())
        print(f'{code} q={q} center={center} rows={rows} {mode} [{eng.stage_a}]: silent chunks max |out| = {worst:.3e}; loud chunks differ: {loud > 0}')
        assert worst == 0.0
        # and the loud chunks are what they are without the silent ones around them
        ref = eng.process(raw[2:6].tobytes()).reshape(rows, 4, n // q)
        assert np.array_equal(ref[:, :3], out[:, 2:5])
        eng.close()


@pytest.mark.parametrize('code,q,center,rows,fused,n', [('h', 64, 15700, 1, None, 0), ('h', 64, 15700, 1, '0', 0), ('H', 64, -203300, 1, None, 0),
                                                        ('B', 64, 15700, 1, None, 0), ('b', 32, 15700, 1, '0', 0), ('h', 32, 0, 1, None, 0),
                                                        ('h', 128, 15700, 3, None, 0), ('h', 256, 15700, 1, None, 0), ('B', 128, 15700, 2, '0', 0),
                                                        ('h', 32, 15700, 3, None, 0), ('H', 32, 15700, 1, '0', 0),
                                                        ('h', 64, 15700, 1, None, 65536), ('h', 32, 15700, 2, None, 65536), ('H', 64, 15700, 1, '0', 65536)])
def test_a_stream_that_starts_silent_under_correct_iq_comes_out_as_exact_zeros(code, q, center, rows, fused, n, monkeypatch):
    """--correct-iq (read_file.py:64-77): the offset starts at zero, so the silent head of a stream stays x - 0 = 0 and the reference
    writes exact zeros until the first loud sample.  The int8 forms carry the constant their byte slices leave out in a rank-one
    term; the weight planes of those forms are quantised so that this term cancels the matrix cores' share to the last bit
    (operators.cpp: quantise_planes, "consistent").  Fused and prefix forms of the correction, one and several rows; the loud
    chunks that follow must match oracle/chain.c as always."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    if fused is None:
        _diag(monkeypatch, iq_fused=None)
    else:
        _diag(monkeypatch, iq_fused=fused)
    # (n = 65536: chunks of sixteen tiles, two per wave -- the forms shortest of LDS)
    fs, n, nch, head = 1024000, n or samples_per_chunk(code), 37 if not n else 11, 19 if not n else 5
    item = {'b': 1, 'B': 1, 'h': 2, 'H': 2}[code]
    freqs = [float(center + 50000 * r) for r in range(rows)]
    omega = 5000 if q < 128 else (2500 if q < 256 else 1200)
    raw = np.frombuffer(prng.raw_bytes(code, n * nch, seed=93, amp=100 if item == 1 else 8000), dtype=np.uint8).reshape(nch, -1).copy()
    raw[:head] = 0
    for mode in ('fm', 'am'):
        eng = Engine(fs, q, mode, code, n_samples=n, row_freqs=freqs, apply_shift=bool(center or rows > 1), omega_out=omega, correct_iq=True)
        assert eng.stage_a == ('i8w' if q <= 64 else 'f64')       # (decimation 128 and up with --correct-iq: the fp64 kernel)
        m = n // q
        out = eng.process(raw.tobytes()).reshape(rows, nch, m)
        worst = float(np.abs(out[:, :head - 1]).max())
        print(f'{code} q={q} center={center} rows={rows} fused={fused} {mode} [{eng.stage_a}]: silent head max |out| = {worst:.3e}')
        assert worst == 0.0
        eng.close()
        exp = _oracle_rows_iq(raw.tobytes(), code, n, q, mode, freqs, fs, omega, nch)
        got = out.transpose(1, 0, 2)
        # chunk head-1 ends where the loud samples begin (zero-phase filter: its tail already sees them); from there on the usual bound
        err = float(np.abs(got[head - 1:] - exp[head - 1:]).max())
        assert err <= tol_for(exp, mode), err


@pytest.mark.gpu
@pytest.mark.parametrize('code,n,q,mode,rows', [('h', 8192, 2, 'fm', 1), ('h', 4096, 16, 'am', 2), ('h', 8192, 4, 'fm', 3), ('h', 12288, 8, 'am', 1), ('h', 16384, 2, 'fm', 2),
                                                 ('h', 20480, 16, 'am', 1), ('H', 28672, 4, 'am', 2), ('h', 16384, 16, 'fm', 1),
                                                 ('B', 65536, 2, 'fm', 1), ('b', 65536, 4, 'fm', 2), ('B', 65536, 8, 'am', 3), ('B', 36864, 16, 'am', 1), ('b', 20480, 8, 'am', 2),
                                                 ('B', 32768, 2, 'fm', 1), ('b', 4096, 16, 'am', 1), ('B', 49152, 4, 'am', 1)])
def test_one_pass_kernel_on_other_chunk_sizes(code, n, q, mode, rows, monkeypatch):
    """chain_sq_kernel gives each of its four waves the tiles w and w + 4 of a chunk (16 columns of 256 samples each): the reference's
    16-bit chunk has eight (32768 samples, read_file.py:51), its 8-bit chunk sixteen (65536 samples: two pairs of tiles per wave, one
    after the other), the C ABI takes any multiple of 4096 samples up to that (from 8192 at decimation 2 and 4) -- one to fifteen
    tiles: waves without a tile, waves with one, an odd count.  Against oracle/chain.c (dsp_processor.py:140-149), several rows, with
    the chunk's bits independent of the batch it arrives in.  (FM where the chunk's output count is a power of two: the resampler.)"""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    fs, nch = 1024000, 9
    freqs = [15700.0, -203300.0, 100100.0][:rows]
    omega = 100000 if q == 2 else (60000 if q == 4 else (30000 if q == 8 else 15000))
    raw = prng.raw_bytes(code, n * nch, seed=900 + n // 4096 + q, amp=100 if code in 'bB' else 8000)
    eng = Engine(fs, q, mode, code, n_samples=n, row_freqs=freqs, apply_shift=True, omega_out=omega)
    assert eng.stage_a == 'sq'
    out, y = eng.process_debug(raw)
    m = n // q
    exp = _oracle_rows(raw, code, n, q, mode, freqs, fs, omega, range(nch), range(rows))
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    err = float(np.abs(got - exp).max())
    tol = tol_for(exp, mode)
    print(f'n={n} q={q} {mode} x {rows} [{eng.stage_a}]: max abs err {err:.3e} (tol {tol:.1e})')
    assert err <= tol
    whole = eng.process(raw).reshape(rows, nch, m)
    part = eng.process(raw[3 * eng.chunk_bytes:8 * eng.chunk_bytes]).reshape(rows, 5, m)
    assert np.array_equal(part, whole[:, 3:8])
    # a silent chunk between loud ones comes out as exact zeros (demodulation.py:30-32: angle(0) = 0; AM: 0)
    quiet = bytearray(raw)
    quiet[4 * eng.chunk_bytes:5 * eng.chunk_bytes] = bytes(eng.chunk_bytes)
    z = eng.process(bytes(quiet)).reshape(rows, nch, m)
    assert float(np.abs(z[:, 4]).max()) == 0.0
    assert np.array_equal(z[:, :4], whole[:, :4])


def _oracle_rows_iq(raw, code, n, q, mode, freqs, fs, omega_out, nch):
    """oracle/chain.c with --correct-iq carried across the chunks (read_file.py:64-77, inductance 50 / fs): (chunk, row, m)."""
    from oracle import c_chain, ref_chain
    exp, _off = c_chain.process_stream(raw, code, n, q, mode, c_chain.shift_table(freqs, fs, n), ref_chain.cheby_sos(q),
                                       ref_chain.ellip_sos(fs // q, omega_out), iq_inductance=50 / fs, threads=4)
    return exp


@pytest.mark.parametrize('fused', ['1', '0'])
def test_correct_iq_over_a_batch_of_many_chunks(fused, monkeypatch):
    """--correct-iq carries one offset through the whole stream (read_file.py:64-77).  Inside a batch the chunks' starting offsets are a
    scan over the chunk aggregates; from 6145 chunks per call on it runs over several workgroups (kernels.hip: iq_scan_reduce / top /
    apply; one workgroup took 0.23 ms for the 131072 chunks of a bench step).  6200 + 301 chunks in two calls against oracle/chain.c,
    the carried offset included; both forms of the correction."""
    from sdrterm_amd.engine import Engine
    from oracle import c_chain, ref_chain
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    _diag(monkeypatch, iq_fused=fused)
    fs, q, code, n, nch, first = 1024000, 64, 'h', 32768, 6501, 6200
    one = np.frombuffer(prng.raw_bytes(code, n * 13, seed=19, amp=3000, dc=(700, -450)), dtype=np.uint8)
    raw = np.tile(one, nch // 13 + 1)[:nch * n * 4].tobytes()
    eng = Engine(fs, q, 'fm', code, n_samples=n, row_freqs=[15700.0], apply_shift=True, omega_out=5000, correct_iq=True)
    assert eng.stage_a == 'i8w'
    a = eng.process(raw[:first * n * 4])
    b = eng.process(raw[first * n * 4:])
    got = np.concatenate([a.reshape(1, first, n // q), b.reshape(1, nch - first, n // q)], axis=1).transpose(1, 0, 2)
    exp, off = c_chain.process_stream(raw, code, n, q, 'fm', c_chain.shift_table([15700.0], fs, n), ref_chain.cheby_sos(q),
                                      ref_chain.ellip_sos(fs // q, 5000), iq_inductance=50 / fs, threads=4)
    err = float(np.abs(got - exp).max())
    print(f'fused={fused}: {nch} chunks in two calls, max abs err {err:.3e}, carried offset {eng.iq_offset} vs {off}')
    assert err <= 1e-6
    assert abs(eng.iq_offset - off) <= 1e-9 * abs(off)


@pytest.mark.parametrize('code', list('bBhHiIfd'))
def test_every_instantiation_of_the_per_sample_kernel(code, monkeypatch):
    """chain_seq_kernel (decimation 2 and 4) exists once per encoding x decimation x frequency shift x --correct-iq: 64 instantiations
    of a kernel that uses 450 - 490 registers.  Round 4 rebuilt its passes as compile-time blocks of 16 samples, and on the way two of
    the 64 came back with ONE thread's anticausal aggregate overwritten (float32 at decimation 4, int8 at decimation 2: the outputs of
    run 254 wrong, everything else to the last bit) depending on where an unrelated block of the source stood -- nothing the golden
    cases, which exercise five of the 64, would have seen.  Hence every instantiation against oracle/chain.c, three chunks each, the
    run length at its minimum (32 samples) for the 4- and 8-byte encodings and at 64 for the others."""
    from sdrterm_amd.engine import Engine
    from oracle import c_chain, ref_chain
    from tests.golden import prng
    monkeypatch.setenv('SDRTERM_STAGE_A', 'f64')            # (decimation 4 of the 8- and 16-bit encodings otherwise goes to the wide kernel)
    fs, nch = 1024000, 3
    n = {'b': 16384, 'B': 16384, 'h': 16384, 'H': 16384}.get(code, 8192)
    raw = prng.raw_bytes(code, n * nch, seed=5, amp=100 if code in 'bB' else 3000)
    worst = 0.0
    for q in (2, 4):
        for iq in (False, True):
            for center in (0.0, 15700.0):
                eng = Engine(fs, q, 'fm', code, n_samples=n, row_freqs=[center], apply_shift=bool(center), omega_out=60000, correct_iq=iq)
                assert eng.stage_a == 'seq'
                out = eng.process(raw)
                eng.close()
                shift = c_chain.shift_table([center], fs, n) if center else None
                exp, _ = c_chain.process_stream(raw, code, n, q, 'fm', shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, 60000),
                                                iq_inductance=50 / fs if iq else None, threads=4)
                err = float(np.abs(out.reshape(1, nch, n // q).transpose(1, 0, 2) - exp).max())
                print(f'{code} n={n} q={q} iq={iq} center={center}: max abs err {err:.3e}')
                worst = max(worst, err)
    assert worst <= 1e-9


@pytest.mark.parametrize('form', ['one-pass', 'passes'])
@pytest.mark.parametrize('code,q,mode,rows', [('h', 16, 'fm', 3), ('h', 8, 'am', 2), ('H', 16, 'am', 1), ('B', 16, 'fm', 2), ('b', 8, 'fm', 1), ('h', 4, 'fm', 1),
                                              ('h', 2, 'fm', 2), ('H', 2, 'am', 1), ('H', 4, 'fm', 3), ('h', 8, 'fm', 1)])
def test_small_decimations_in_passes_of_the_wide_kernel(code, q, mode, rows, form, monkeypatch):
    """Decimation 4 / 8 / 16 on the wide-column int8 kernel: a 256-sample column has 64 / 32 / 16 outputs and the kernel finishes eight
    per pass (8 / 4 / 2 passes over the chunk, tables and local planes per pass: api.hip WidePass).  Against oracle/chain.c (the
    reference's chain, dsp_processor.py:140-149) on several rows, FM and AM, odd batch sizes; the bits of a chunk must not depend on
    the batch it arrives in; and the fp64 kernel forced on the same input must agree to the usual 1e-9 of the decimated peak.
    Round 5, form 'one-pass': 8- and 16-bit records at decimation 2 .. 16 take chain_sq_kernel instead (kernels_sq.hip: all 256 / q outputs
    of a column from ONE read of the input, the local operator as windows of one master sequence) -- the same checks; 'passes'
    keeps the older forms under test (SDRTERM_DIAG=sq=0: they still serve 8-bit records and are what a context falls back to)."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    if form == 'passes':
        _diag(monkeypatch, sq=0)
        if q == 2:
            pytest.skip('decimation 2 without the one-pass kernel is the per-sample kernel (test_every_instantiation_of_the_per_sample_kernel)')
    fs, n, nch = 1024000, samples_per_chunk(code), 7
    freqs = [15700.0, -203300.0, 100100.0][:rows]
    omega = 60000 if q == 4 else (30000 if q == 8 else 15000)
    raw = prng.raw_bytes(code, n * nch, seed=400 + q, amp=100 if code in 'bB' else 8000)
    omega = 100000 if q == 2 else omega
    eng = Engine(fs, q, mode, code, row_freqs=freqs, apply_shift=True, omega_out=omega)
    assert eng.stage_a == ('sq' if form == 'one-pass' else 'i8w')
    out, y = eng.process_debug(raw)
    m = n // q
    exp = _oracle_rows(raw, code, n, q, mode, freqs, fs, omega, range(nch), range(rows))
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    err = float(np.abs(got - exp).max())
    tol = tol_for(exp, mode)
    print(f'{code} q={q} {mode} x {rows} [{eng.stage_a}]: max abs err {err:.3e} (tol {tol:.1e})')
    assert err <= tol
    # the same chunks in another batching: the same bits (process() hands FM over as pair products, process_debug() as samples:
    # the two differ in the last place, test_pair_products_between_the_stages_equal_the_sample_path)
    whole = eng.process(raw).reshape(rows, nch, m)
    assert np.abs(whole - out.reshape(rows, nch, m)).max() <= 1e-12 * max(1.0, np.abs(whole).max())
    part = eng.process(raw[2 * eng.chunk_bytes:5 * eng.chunk_bytes]).reshape(rows, 3, m)
    assert np.array_equal(part, whole[:, 2:5])
    monkeypatch.setenv('SDRTERM_STAGE_A', 'f64')
    from sdrterm_amd._native import SdrtError
    try:
        e64 = Engine(fs, q, mode, code, row_freqs=freqs, apply_shift=True, omega_out=omega)
    except SdrtError as e:
        # 8-bit records at decimation 8: 8192 blocks per chunk do not fit the fp64 kernel's LDS -- only the int8 form takes them
        assert 'LDS' in str(e) and code in 'bB' and q == 8
        return
    assert e64.stage_a in ('f64', 'seq')
    _o64, y64 = e64.process_debug(raw)
    assert np.abs(y - y64).max() <= 1e-9 * np.abs(y64).max()
