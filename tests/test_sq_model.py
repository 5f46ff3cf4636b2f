"""The column form the one-pass small-decimation kernel evaluates (csrc/kernels_sq.hip), as numpy, against the oracle's decimate
(scipy.signal.decimate: the reference's dsp_processor.py:147): the local Toeplitz part inside a 256-sample column plus the modal states
at the column boundaries reproduce the zero-phase decimator to rounding, from the tables the library itself builds on the host
(sdrt_build_decim_tables; no GPU involved)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.parametrize('q', [2, 4, 8, 16])
def test_column_form_of_the_decimator_equals_the_oracle(q):
    import model_sq
    assert model_sq.rel_err(q) <= 1e-12
