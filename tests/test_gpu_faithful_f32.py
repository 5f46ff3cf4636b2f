"""GPU parity of the faithful float32 mode (sdrt_config.faithful_f32, kernels_f32c.hip): float32 records in the reference's own
complex64 arithmetic (src/misc/read_file.py:100-101, src/dsp/dsp_processor.py:147,171-177), held to the reference's vectors with
no float32 allowance: decimated samples bit for bit, outputs at the north-star tolerance."""
import numpy as np
import pytest

from tests.util import BY_NAME, CASES, case_raw, samples_per_chunk, tol_for

pytestmark = pytest.mark.gpu

F32_CASES = [c for c in CASES if c['code'] == 'f']


def _engine(case, **kw):
    from sdrterm_amd.engine import Engine
    from oracle import ref_chain
    rows = ref_chain.vfo_rows(case['center'], case.get('vfos'))
    return Engine(case['fs'], case['q'], case['mode'], 'f', row_freqs=rows, apply_shift=bool(case.get('vfos') or case['center']),
                  omega_out=case['omega_out'], faithful_f32=True, **kw)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.complex64).view(np.uint32)


@pytest.mark.parametrize('case', F32_CASES, ids=[c['name'] for c in F32_CASES])
def test_float32_golden_cases_at_the_north_star_tolerance(case, golden):
    """Every float32 golden case against the REFERENCE's vectors: decimated samples identical to its complex64 `y0`, FM outputs
    within 1e-6 max-abs, AM within 1e-6 of the peak (tests/util.py: tol_for -- the tolerance of every other encoding)."""
    z, _ = golden
    name = case['name']
    eng = _engine(case)
    assert eng.stage_a == 'f32c'
    out, y = eng.process_debug(case_raw(case))
    exp = z[f'{name}/z']
    nch, rows, m = exp.shape
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    gy0 = y.reshape(rows, nch, m)[:, 0, :]
    y0 = z[f'{name}/y0']
    assert np.array_equal(gy0, gy0.astype(np.complex64).astype(np.complex128))         # complex64 values
    np.testing.assert_array_equal(_bits(gy0), _bits(y0))
    err = np.abs(got - exp).max()
    print(f'{name} [faithful]: decimated samples bit-exact, output max abs err {err:.3e} (tol {tol_for(exp, case["mode"]):.1e})')
    assert err <= tol_for(exp, case['mode'])
    # and the production entry point (no debug buffer) gives the same outputs
    np.testing.assert_array_equal(eng.process(case_raw(case)), out)


def test_big_endian_float32_records(golden):
    z, _ = golden
    case = BY_NAME['f_q64_fm_c']
    raw = np.frombuffer(case_raw(case), dtype='<u4').astype('>u4').tobytes()
    out = _engine(case, big_endian=True).process(raw)
    exp = z['f_q64_fm_c/z']
    nch, rows, m = exp.shape
    assert np.abs(out.reshape(rows, nch, m).transpose(1, 0, 2) - exp).max() <= 1e-6


_ROWS63 = ','.join(str((i - 31) * 7000 + 500) for i in range(63))          # 63 csv rows + the centre = 64: a whole lane group per chunk


_ROWS127 = ','.join(str((i - 63) * 3500 + 250) for i in range(127))       # two lane groups per chunk (the kernel without sequences in memory)


@pytest.mark.parametrize('rows_csv,nch,q,mode', [('-200000,-150000,-100000,50000', 30, 64, 'fm'), (None, 70, 32, 'am'),
                                                  ('-200000,150000', 5, 4, 'fm'), ('100000', 3, 2, 'am'), (_ROWS63, 19, 64, 'fm'),
                                                  (_ROWS127, 5, 32, 'am'), (_ROWS63, 3, 2, 'fm')],
                         ids=['5rows-30', '1row-70', '3rows-q4', '2rows-q2', '64rows-19', '128rows-5', '64rows-q2'])
def test_many_sequences_against_the_faithful_oracle(rows_csv, nch, q, mode):
    """More (chunk, row) sequences than one wave holds (groups of 64 lanes, the last one partly filled), against oracle/chain.c's
    complex64 restatement (itself pinned bit for bit to the reference's vectors): decimated samples identical, outputs 1e-6; the
    same stream in another batching gives the same bits."""
    from sdrterm_amd.engine import Engine
    from oracle import c_chain, ref_chain
    from tests.golden import prng
    fs, center, n = 1024000, 15700, samples_per_chunk('f')
    freqs = ref_chain.vfo_rows(center, rows_csv)
    raw = prng.raw_bytes('f', n * nch, seed=91, amp=3000)
    eng = Engine(fs, q, mode, 'f', row_freqs=freqs, apply_shift=True, omega_out=5000 * (64 // q) ** 0.5, faithful_f32=True)
    out, y = eng.process_debug(raw)
    shift = ref_chain.shift_table(freqs, fs, n)
    exp, ydec = c_chain.process_stream_f32c(raw, n, q, mode, shift, ref_chain.ellip_sos(fs // q, 5000 * (64 // q) ** 0.5), threads=4)
    rows, m = len(freqs), n // q
    np.testing.assert_array_equal(_bits(y.reshape(rows, nch, m).transpose(1, 0, 2)), _bits(ydec))
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    assert np.abs(got - exp).max() <= tol_for(exp, mode)
    half = nch // 2
    out2 = eng.process(raw[half * n * 8:])
    np.testing.assert_array_equal(out2.reshape(rows, nch - half, m), out.reshape(rows, nch, m)[:, half:])


def test_faithful_mode_refuses_what_the_reference_cannot_do():
    from sdrterm_amd import _native as nat
    from sdrterm_amd.engine import Engine
    with pytest.raises(nat.SdrtError, match='correct_iq'):
        Engine(1024000, 64, 'fm', 'f', row_freqs=[15700], omega_out=5000, faithful_f32=True, correct_iq=True)
    with pytest.raises(nat.SdrtError, match='contradict'):
        Engine(1024000, 64, 'fm', 'f', row_freqs=[15700], omega_out=5000, faithful_f32=True, stage_a='f64')
    # other encodings: the reference computes them in float64 anyway, the flag changes nothing
    assert Engine(1024000, 64, 'fm', 'h', row_freqs=[15700], omega_out=5000, faithful_f32=True).stage_a != 'f32c'
