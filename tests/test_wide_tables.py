"""CPU check of the host-built operator tables of the wide-column kernel (sdrterm_amd/csrc/operators.cpp): a numpy
emulation of the kernel's arithmetic from those tables (tools/emulate_wide.py: digit planes de-quantised, column
aggregates, in-tile scans, tile carries, edge operators, IQ correction folded into the planes + rank-one term) must
reproduce scipy.signal.decimate, i.e. what the reference computes at dsp_processor.py:147 (and read_file.py:64-77)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.parametrize('q,n,theta,lam,off0', [
    (64, 32768, 0.0, 0.0, 0j),
    (64, 32768, 0.37, 0.0, 0j),
    (32, 32768, 0.1, 0.0, 0j),
    (64, 65536, -1.1, 0.0, 0j),
    (64, 32768, 0.37, 50 / 1024000, 300 - 200j),
    (32, 32768, -0.2, 0.01, -1000 + 50j),
    # decimation 16 / 8 / 4: 16 / 32 / 64 outputs per column in 2 / 4 / 8 passes of eight (sdrt_build_*_pass)
    (16, 32768, 0.3, 0.0, 0j),
    (8, 32768, -0.4, 0.0, 0j),
    (4, 32768, 0.0, 0.0, 0j),
    (16, 65536, 0.2, 0.0, 0j),
])
def test_tables_reproduce_scipy_decimate(q, n, theta, lam, off0):
    import emulate_wide
    assert emulate_wide.run(q, n, theta, lam=lam, off0=off0) < 2e-11
