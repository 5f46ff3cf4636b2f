"""GPU parity: HIP path (through the C ABI) vs golden vectors of the real reference and vs the oracle."""
import ctypes

import numpy as np
import pytest

from tests.util import BY_NAME, CASES, case_raw, ref_tol, samples_per_chunk, tol_for

pytestmark = pytest.mark.gpu


def _engine(case):
    from sdrterm_amd.engine import Engine
    from oracle import ref_chain
    rows = ref_chain.vfo_rows(case['center'], case.get('vfos'))
    return Engine(case['fs'], case['q'], case['mode'], case['code'], row_freqs=rows,
                  apply_shift=bool(case.get('vfos') or case['center']), omega_out=case['omega_out'],
                  normalize=bool(case.get('normalize')), correct_iq=bool(case.get('correct_iq')),
                  big_endian=bool(case.get('big_endian')), smooth=case.get('smooth', 0))


def test_mfma_layout_selftest():
    from sdrterm_amd import _native
    err = ctypes.c_double(-1)
    _native.check(_native.lib().sdrt_selftest_mfma(ctypes.byref(err)))
    assert err.value == 0.0
    bad = ctypes.c_int(-1)
    _native.check(_native.lib().sdrt_selftest_mfma_i8(ctypes.byref(bad)))
    assert bad.value == 0


@pytest.fixture(params=['auto', 'f64', 'i8t'])
def stage_a(request, monkeypatch):
    """Every golden case runs with the kernel the library picks (int8 matrix cores: wide-column persistent kernel where it
    applies, else the one-block-per-column 16x16x64 form, for 8/16-bit integer encodings), with the fp64 matrix-core kernel
    forced, and with the one-block 16x16x64 int8 form forced."""
    if request.param == 'auto':
        monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    else:
        monkeypatch.setenv('SDRTERM_STAGE_A', request.param)
    return request.param


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_golden_parity(case, golden, stage_a):
    z, _ = golden
    name = case['name']
    eng = _engine(case)
    if stage_a == 'f64':
        assert eng.stage_a == 'f64'
    elif stage_a == 'auto' and eng.stage_a not in ('i8w', 'i8t'):
        pytest.skip(f'auto: the library uses {eng.stage_a} for this case (covered by another run)')
    elif stage_a != 'auto' and eng.stage_a != stage_a:
        pytest.skip(f'{stage_a}: the library uses {eng.stage_a} for this case (covered by another run)')
    raw = case_raw(case)
    out, y = eng.process_debug(raw)
    exp = z[f'{name}/z']                       # (chunks, N, m)
    nch, rows, m = exp.shape
    got = out.reshape(rows, nch, m).transpose(1, 0, 2)
    y0 = z[f'{name}/y0'].astype(np.complex128)
    gy0 = y.reshape(rows, nch, m)[:, 0, :]
    yerr = np.abs(gy0 - y0).max() / np.abs(y0).max()
    err = np.abs(got - exp).max()
    print(f'{name} [{eng.stage_a}]: decimated rel err {yerr:.3e}, output max abs err {err:.3e} (tol {ref_tol(case, exp):.1e})')
    assert yerr <= (1e-3 if case['code'] == 'f' else 1e-9)
    assert err <= ref_tol(case, exp)


@pytest.mark.parametrize('rows,nch', [(64, 3), (300, 2), (33, 17), (5, 100)])
def test_item_distribution_of_the_persistent_kernel(rows, nch, monkeypatch):
    """Many rows x few chunks and few rows x many chunks (rows above and below the number of workgroup slots, chunk counts
    that are not multiples of 8): the wide-column persistent kernel must cover every (chunk, row) item exactly once --
    checked against the independently written one-block-per-column kernel."""
    from sdrterm_amd.engine import Engine
    from tests.golden import prng
    fs, q = 1024000, 64
    freqs = [1000.0 * (i - rows // 2) for i in range(rows)]
    raw = prng.raw_bytes('h', 32768 * nch, seed=rows, amp=5000)
    monkeypatch.delenv('SDRTERM_STAGE_A', raising=False)
    a = Engine(fs, q, 'fm', 'h', row_freqs=freqs, omega_out=5000)
    out_a = a.process(raw)
    monkeypatch.setenv('SDRTERM_STAGE_A', 'i8t')
    b = Engine(fs, q, 'fm', 'h', row_freqs=freqs, omega_out=5000)
    out_b = b.process(raw)
    assert (a.stage_a, b.stage_a) == ('i8w', 'i8t')
    assert np.isfinite(out_a).all() and np.abs(out_a - out_b).max() < 1e-8


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
        monkeypatch.setenv('SDRTERM_RESAMPLE', resample)
        monkeypatch.setenv('SDRTERM_WIDE_RS', rs)
        out, _ = _engine(case).process_debug(case_raw(case))
        got = out.reshape(rows, nch, m).transpose(1, 0, 2)
        assert np.abs(got - exp).max() <= ref_tol(case, exp)
        outs.append(got)
    assert np.abs(outs[0] - outs[1]).max() < 1e-9 and np.abs(outs[0] - outs[2]).max() < 1e-9
