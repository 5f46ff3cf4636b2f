"""Pins the oracle (oracle/ref_chain.py and oracle/chain.c) to vectors produced by the real reference."""
import numpy as np
import pytest

from oracle import c_chain, ref_chain
from tests.util import CASES, case_raw, samples_per_chunk, tol_for


def _chain(case):
    return ref_chain.Chain(case['fs'], case['q'], case['mode'], case['code'], center=case['center'],
                           vfos=case.get('vfos'), omega_out=case['omega_out'],
                           normalize=bool(case.get('normalize')), correct_iq=bool(case.get('correct_iq')),
                           big_endian=bool(case.get('big_endian')))


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_numpy_oracle_matches_reference(case, golden):
    z, _ = golden
    name = case['name']
    raw = case_raw(case)
    n = samples_per_chunk(case['code'])
    item = len(raw) // (2 * n * case['n_chunks'])
    ch = _chain(case)
    exp = z[f'{name}/z']
    for c in range(case['n_chunks']):
        rawc = raw[c * 2 * n * item:(c + 1) * 2 * n * item]
        zin = ch.read(rawc)
        np.testing.assert_allclose(zin[::case.get('zsub', 251)], z[f'{name}/reader_sub'][c], rtol=1e-12, atol=1e-9)
        assert abs(zin[-1] - z[f'{name}/reader_last'][c]) <= 1e-9 * max(1.0, abs(zin[-1]))
        out, y = ref_chain.process_chunk(zin, case['q'], case['mode'], ch.shift, ch.out_sos, True, smooth=case.get('smooth', 0))
        assert np.abs(out - exp[c]).max() <= tol_for(exp[c]) * 1e-3   # same scipy -> essentially exact
        if c == 0:
            y0 = z[f'{name}/y0']
            assert np.abs(y - y0).max() <= (1e-6 if case['code'] == 'f' else 1e-9) * np.abs(y0).max()
    np.testing.assert_array_equal(ref_chain.cheby_sos(case['q']), z[f'{name}/cheby_sos'])
    np.testing.assert_allclose(ch.out_sos, z[f'{name}/ellip_sos'], rtol=0, atol=0)


# (float32 records: the reference's arithmetic is complex64 -- test_c_oracle_in_the_references_float32_arithmetic below)
@pytest.mark.parametrize('case', [c for c in CASES if c['code'] != 'f'], ids=[c['name'] for c in CASES if c['code'] != 'f'])
def test_c_oracle_matches_reference(case, golden):
    z, _ = golden
    name = case['name']
    raw = case_raw(case)
    n = samples_per_chunk(case['code'])
    rows = ref_chain.vfo_rows(case['center'], case.get('vfos'))
    shift = None
    if case.get('vfos') or case['center']:
        shift = c_chain.shift_table(rows, case['fs'], n)
    norm = ref_chain.generate_domain(case['code']) if case.get('normalize') else None
    L = 50 / case['fs'] if case.get('correct_iq') else None
    out, off = c_chain.process_stream(raw, case['code'], n, case['q'], case['mode'], shift,
                                      z[f'{name}/cheby_sos'], z[f'{name}/ellip_sos'],
                                      big_endian=bool(case.get('big_endian')), norm=norm, iq_inductance=L,
                                      threads=2, smooth=case.get('smooth', 0))
    exp = z[f'{name}/z']
    assert out.shape == exp.shape
    assert np.abs(out - exp).max() <= tol_for(exp, case['mode'])


def test_known_answer_vectors(golden):
    """test/dsp/demodulation_test.py:19-50 (8-sample KATs), oracle vs the reference's own outputs."""
    z, _ = golden
    inp = z['kat_in'][0]
    np.testing.assert_allclose(c_chain.fm(inp), z['kat_fm'][0], atol=1e-12, rtol=0)
    np.testing.assert_allclose(ref_chain.fm_demod(inp[None, :])[0], z['kat_fm'][0], atol=1e-12, rtol=0)
    np.testing.assert_allclose(c_chain.am(inp), z['kat_am'][0], atol=1e-12, rtol=0)
    np.testing.assert_allclose(ref_chain.am_demod(inp), z['kat_am'][0], atol=1e-12, rtol=0)
    # the reference test's own expectation: atan2 of the pair products, FFT-resampled 4 -> 8
    import math
    from scipy.signal import resample
    pairs = [inp[i] * inp[i + 1].conjugate() for i in range(0, 8, 2)]
    expect = resample([math.atan2(p.imag, p.real) for p in pairs], 8)
    np.testing.assert_allclose(c_chain.fm(inp), expect, atol=1e-12, rtol=0)
    np.testing.assert_allclose(z['kat_re'][0], inp.real)
    np.testing.assert_allclose(z['kat_im'][0], inp.imag)


def test_c_oracle_zi_and_edge(golden):
    z, _ = golden
    import ctypes as C
    lib = c_chain.lib()
    for q in (4, 8, 16, 32, 64, 128, 256):
        sos = np.ascontiguousarray(z[f'coef/cheby1_q{q}'])
        zi = np.empty((4, 2))
        lib.orc_sosfilt_zi(sos.ctypes.data_as(C.c_void_p), C.c_int(4), zi.ctypes.data_as(C.c_void_p))
        np.testing.assert_allclose(zi, z[f'coef/cheby1_zi_q{q}'], rtol=1e-9)
        lib.orc_sosfiltfilt_edge.restype = C.c_int
        assert lib.orc_sosfiltfilt_edge(sos.ctypes.data_as(C.c_void_p), C.c_int(4)) == 27


def test_generate_domain_table():
    """test/misc/read_file_test.py:12-22."""
    chars = ('B', 'b', 'H', 'h', 'I', 'i', 'L', 'l')
    vals = (0, 255, -128, 127, 0, 65536, -32768, 32767, 0, 4294967295, -2147483648, 2147483647,
            0, 18446744073709551615, -9223372036854775808, 9223372036854775807)
    for c, (a, b) in zip(chars, [(vals[i], vals[i + 1]) for i in range(0, len(vals), 2)]):
        xmin, diff = ref_chain.generate_domain(c)
        assert xmin == a and diff == 1 / (b - a)
    assert ref_chain.generate_domain('f') is None


@pytest.mark.parametrize('code,normalize', [('h', False), ('B', True), ('f', False)])
def test_plot_tap_restatements_match_the_reference_plots(code, normalize, golden):
    """ref_chain.spectrum_tap / waterfall_tap against the arrays the reference's own plots computed in update()
    (plots/spectrum_analyzer_plot.py:74-88, plots/waterfall_plot.py:90-100; captured by make_golden.py with a pyqtgraph stand-in).
    Compared as amplitudes relative to the largest bin (the plots take logarithms; empty bins are rounding noise).  float32
    records: the reference's chunk is complex64 and scipy transforms it in single precision."""
    from tests.golden import prng
    from tests.golden.cases import TAP_FS, TAP_OFFSET, TAP_SEED
    z, _ = golden
    n = samples_per_chunk(code)
    raw = prng.raw_bytes(code, n, seed=TAP_SEED, amp=100 if code == 'B' else 3000)
    x = ref_chain.unpack(raw, code, False)
    if normalize:
        x = ref_chain.normalize(x, code)
    tol = 2e-6 if code == 'f' else 1e-10
    amp = z[f'tap/{code}/amp']
    mine = ref_chain.spectrum_tap(x, TAP_FS, TAP_OFFSET)
    ea, ma = 10.0 ** (amp.astype(np.float64) / 2), 10.0 ** (mine / 2)
    assert mine.shape == amp.shape and np.abs(ma - ea).max() <= tol * ea.max()
    img, cols = z[f'tap/{code}/image'], z[f'tap/{code}/image_cols']
    full = ref_chain.waterfall_tap(x, TAP_FS, TAP_OFFSET)
    assert tuple(z[f'tap/{code}/image_shape']) == full.shape == (1024, n // 128 + 1)
    ei, mi = 10.0 ** (img.astype(np.float64) / 10), 10.0 ** (full[:, cols] / 10)
    assert np.abs(mi - ei).max() <= tol * ei.max()


F32_CASES = [c for c in CASES if c['code'] == 'f']


@pytest.mark.parametrize('case', F32_CASES, ids=[c['name'] for c in F32_CASES])
def test_c_oracle_in_the_references_float32_arithmetic(case, golden):
    """float32 records: the reference runs its decimator in complex64 (read_file.py:100-101, dsp_processor.py:147,171-177).
    chain.c: orc_process_stream_f32c performs scipy's single-precision operations in scipy's order: its decimated samples are
    the reference's complex64 `y0` BIT FOR BIT, its outputs the reference's at the north-star tolerance (no float32 allowance)."""
    z, _ = golden
    name = case['name']
    raw = case_raw(case)
    n = samples_per_chunk('f')
    rows = ref_chain.vfo_rows(case['center'], case.get('vfos'))
    shift = ref_chain.shift_table(rows, case['fs'], n) if (case.get('vfos') or case['center']) else None
    out, ydec = c_chain.process_stream_f32c(raw, n, case['q'], case['mode'], shift, z[f'{name}/ellip_sos'], threads=2)
    y0 = np.ascontiguousarray(z[f'{name}/y0'])
    assert y0.dtype == np.complex64
    np.testing.assert_array_equal(np.ascontiguousarray(ydec[0]).view(np.uint32), y0.view(np.uint32))
    exp = z[f'{name}/z']
    assert out.shape == exp.shape
    err = np.abs(out - exp).max()
    assert err <= tol_for(exp, case['mode']), err
    if shift is not None:           # the table itself: numpy's, as the reference builds it
        np.testing.assert_array_equal(shift[:, ::1021], z[f'{name}/shift_sub'])


def test_numpy_oracle_float32_is_bit_exact_too(golden):
    """ref_chain.process_chunk(faithful_dtype=True) calls scipy as the reference does: same bits as the reference's y0."""
    z, _ = golden
    for case in F32_CASES:
        raw = case_raw(case)
        n = samples_per_chunk('f')
        ch = _chain(case)
        _, y = ref_chain.process_chunk(ch.read(raw[:n * 8]), case['q'], case['mode'], ch.shift, ch.out_sos, True)
        y0 = np.ascontiguousarray(z[f"{case['name']}/y0"])
        np.testing.assert_array_equal(np.ascontiguousarray(y.astype(np.complex64)).view(np.uint32), y0.view(np.uint32))
