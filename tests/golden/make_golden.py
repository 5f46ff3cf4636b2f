#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (peads/sdrterm @ /root/reference).

Run only in the build container (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

What is captured:
  * the reference reader's per-sample steps (unpack, --normalize-input, --correct-iq with the
    offset carried across chunks) by calling ``misc.read_file.readFile`` on seeded temp files;
  * ``DspProcessor`` / ``VfoProcessor`` ``_processChunk`` outputs for consecutive chunks, plus the
    intermediate ``decimate`` output of the first chunk;
  * the filter coefficient tables scipy produced for those cases (cheby1 SOS, sosfilt_zi, ellip SOS);
  * the 8-sample known-answer vectors of test/dsp/demodulation_test.py;
  * the arrays the plots draw (``amp`` of SpectrumAnalyzerPlot.update, ``image`` of WaterfallPlot.update) from the reference's
    own update() bodies, with tests/golden/_shim/pyqtgraph standing in for the Qt windows.

Inputs come from tests/golden/prng.py (integer PRNG), so only seeds and outputs are stored.
``numba`` is not installed here; tests/golden/_shim/numba is a pure-numpy stand-in for
``guvectorize`` (see its docstring).  Nothing here is imported by the product package.
"""
import json
import os
import queue
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, '_shim'))
sys.path.insert(0, '/root/reference/src')
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
from scipy import signal  # noqa: E402

from tests.golden import prng  # noqa: E402
from tests.golden.cases import CASES, READ_SIZE, TAP_CASES, TAP_FS, TAP_OFFSET, TAP_SEED  # noqa: E402

from dsp.dsp_processor import DspProcessor  # noqa: E402  (reference)
from dsp.vfo_processor import VfoProcessor  # noqa: E402  (reference)
from dsp import demodulation  # noqa: E402  (reference)
from misc.read_file import readFile  # noqa: E402  (reference)
from misc.file_util import DataType  # noqa: E402  (reference)


class _Sink:
    """Stands in for the multiprocessing.Queue the reader feeds (read_file.py:100-117)."""

    def __init__(self):
        self.items = []

    def put_nowait(self, z):
        if isinstance(z, np.ndarray):
            self.items.append(z.copy())

    def close(self):
        pass


class _Proc:
    exitcode = None
    name = 'golden'


class _Flag:
    value = 0


class _NoJoinQueue(queue.Queue):
    """VfoProcessor._generateShift blocks in queue.join() until clients connect (vfo_processor.py:77)."""

    def join(self):
        return


def run_reader(case):
    code = case['code']
    dt = DataType[code].value
    dt = dt.newbyteorder('>' if case.get('big_endian') else '<')
    n = READ_SIZE // (2 * dt.itemsize)
    data = prng.case_bytes(case, n)
    sink = _Sink()
    with tempfile.NamedTemporaryFile(suffix='.raw', delete=False) as f:
        f.write(data)
        path = f.name
    try:
        readFile(bitsPerSample=dt, dataOffset=0, fs=case['fs'], buffers=[sink], processes=[_Proc()],
                 isDead=_Flag(), inFile=path, readSize=READ_SIZE,
                 correctIq=bool(case.get('correct_iq')), normalize=bool(case.get('normalize')))
    finally:
        os.unlink(path)
    assert len(sink.items) == case['n_chunks'], (len(sink.items), case)
    return sink.items, n


def make_processor(case):
    kw = dict(center=case['center'], omegaOut=case['omega_out'], dec=case['q'], smooth=case.get('smooth', False),
              fileInfo={'bitsPerSample': case['code']})
    if case.get('vfos'):
        p = VfoProcessor(case['fs'], vfoHost='localhost:1', vfos=case['vfos'], **kw)
        p._VfoProcessor__queue = _NoJoinQueue()
    else:
        p = DspProcessor(case['fs'], **kw)
    {'fm': p.selectOutputFm, 'am': p.selectOutputAm}[case['mode']]()
    return p


class _Feed:
    """Queue stand-in for DspProcessor._processData (dsp_processor.py:164-183): chunks, then b'' as EOF."""

    def __init__(self, chunks):
        self.items = list(chunks) + [b'']

    def get(self):
        return self.items.pop(0)


class _Capture:
    def __init__(self):
        self.parts = []

    def write(self, b):
        self.parts.append(bytes(b))


def run_case(case):
    """Runs the reference's own consumer loop (_processData) so buffer dtypes follow the reference:
    x/y take the dtype of the first Queue item (complex64 for float32 input, complex128 otherwise)."""
    import struct
    chunks, n = run_reader(case)
    p = make_processor(case)
    nfreq = p._nFreq
    q = case['q']
    m = n // q
    ys = []
    orig = p._processChunk

    def spy(x, y, z):
        orig(x, y, z)
        ys.append(y.copy())

    p._processChunk = spy
    if case.get('vfos'):
        caps = [_Capture() for _ in range(nfreq)]
        p._VfoProcessor__clients = {i: c for i, c in enumerate(caps)}
        fmt, file = '!', None
    else:
        caps = [_Capture()]
        fmt, file = '@', caps[0]
    try:
        p._processData(_Flag(), _Feed(chunks), file)
    except ValueError:
        pass  # x[0, :] = b'' is how the reference's loop ends
    assert len(ys) == case['n_chunks']
    rows = []
    for c in caps:
        data = b''.join(c.parts)
        rows.append(np.array(struct.unpack(fmt + str(len(data) // 8) + 'd', data)).reshape(case['n_chunks'], m))
    outs = np.stack(rows, axis=1)  # (chunks, N, m)
    sub = np.stack([ch[::case.get('zsub', 251)] for ch in chunks])
    last = np.array([ch[-1] for ch in chunks])
    res = {
        'z': outs,                          # (chunks, N, m) float64  -- exactly what the reference writes out
        'y0': ys[0],                        # (N, m) decimate() output of chunk 0 (complex64 for float32 input!)
        'reader_sub': sub,                  # reader output z[::zsub] per chunk (pins unpack/normalize/correctIq)
        'reader_last': last,                # reader output z[-1] per chunk (pins the carried offset)
        'cheby_sos': signal.cheby1(8, 0.05, 0.8 / q, output='sos'),
        'ellip_sos': np.asarray(p._outputFilters),
    }
    if p._shift is not None:
        res['shift_sub'] = np.asarray(p._shift)[:, ::1021].copy()
    return res


def known_answer():
    """test/dsp/demodulation_test.py:12-50 -- the reference's only numerical KATs on this path."""
    inp = np.reshape([0j, 1 + 2j, 2 + 3j, 3 + 4j, 4 + 5j, 5 + 6j, 6 + 7j, 7 + 8j], (1, 8))
    fm = np.empty((1, 8))
    am = np.empty((1, 8))
    re = np.empty((1, 8))
    im = np.empty((1, 8))
    demodulation.fmDemod(np.array(inp), fm)
    demodulation.amDemod(np.array(inp), am)
    demodulation.realOutput(np.array(inp), re)
    demodulation.imagOutput(np.array(inp), im)
    return {'kat_in': inp, 'kat_fm': fm, 'kat_am': am, 'kat_re': re, 'kat_im': im}


def plot_taps():
    """``amp`` of SpectrumAnalyzerPlot.update (plots/spectrum_analyzer_plot.py:74-88) and ``image`` of WaterfallPlot.update
    (plots/waterfall_plot.py:90-100) for one reader chunk per encoding: the reference's own constructors and update() bodies,
    with tests/golden/_shim/pyqtgraph standing in for the windows.  The chunk is what the reference's reader enqueues."""
    from plots.spectrum_analyzer_plot import SpectrumAnalyzerPlot  # noqa: E402  (reference)
    from plots.waterfall_plot import WaterfallPlot  # noqa: E402  (reference)
    res = {}
    for code, normalize in TAP_CASES:
        case = dict(code=code, fs=TAP_FS, n_chunks=1, seed=TAP_SEED, amp=100 if code == 'B' else 3000, normalize=normalize)
        (z,), _n = run_reader(case)

        class _One:
            def __init__(self):
                self.items = [z.copy()]

            def get(self):
                return self.items.pop(0)

            def close(self):
                pass

            def join_thread(self):
                pass
        spec = SpectrumAnalyzerPlot(isDead=_Flag(), fs=TAP_FS, buffer=_One(), center=TAP_OFFSET)
        spec.update()
        res[f'tap/{code}/amp'] = np.asarray(spec.amp)[0].copy()
        res[f'tap/{code}/freq'] = np.asarray(spec.freq)[::1021].copy()
        wf = WaterfallPlot(isDead=_Flag(), fs=TAP_FS, buffer=_One(), center=TAP_OFFSET)
        wf.update()
        # the image is (1024, n/128 + 1) doubles: every 8th column is kept (first, last and interior frames), all 1024 bins
        img = np.asarray(wf.image)
        res[f'tap/{code}/image_shape'] = np.array(img.shape)
        res[f'tap/{code}/image_cols'] = np.arange(0, img.shape[1], 8)
        res[f'tap/{code}/image'] = img[:, ::8].copy()
        print('tap', code, res[f'tap/{code}/amp'].shape, img.shape)
    return res


def main():
    out = {}
    meta = {}
    path = os.path.join(HERE, 'reference_vectors.npz')
    only = sys.argv[1:]                  # names: (re)generate just these cases and keep the rest of the file as it is
    if only:
        old = np.load(path)
        out.update({k: old[k] for k in old.files})
        meta.update(json.loads(bytes(old['meta_json']).decode()))
    for case in CASES:
        if only and case['name'] not in only:
            continue
        res = run_case(case)
        for k, v in res.items():
            out[f"{case['name']}/{k}"] = v
        meta[case['name']] = case
        print(case['name'], res['z'].shape, float(np.abs(res['z']).max()))
    out.update(known_answer())
    if not only or 'taps' in only:
        out.update(plot_taps())
    # coefficient tables for every q / (fs_dec, omega) in use + a few more, straight from scipy
    for q in (4, 8, 16, 32, 64, 128, 256):
        sos = signal.cheby1(8, 0.05, 0.8 / q, output='sos')
        out[f'coef/cheby1_q{q}'] = sos
        out[f'coef/cheby1_zi_q{q}'] = signal.sosfilt_zi(sos)
    for fsd, w in ((16000, 5000), (32000, 5000), (4000000, 12500), (8000000, 12500), (24000, 250)):
        out[f'coef/ellip_{fsd}_{w}'] = signal.ellip(3, 1, 30, w, btype='lowpass', analog=False, output='sos', fs=fsd)
    out['meta_json'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'reference_vectors.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_vectors.npz'),
          os.path.getsize(os.path.join(HERE, 'reference_vectors.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
