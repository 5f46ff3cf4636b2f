#!/usr/bin/env python3
"""Development aid: every instantiation of chain_seq_kernel (8 encodings x decimation 2 / 4 x with / without frequency shift x with /
without --correct-iq) at two chunk sizes against oracle/chain.c.  Run on the GPU box with SDRTERM_STAGE_A=f64 (decimation 4 of the
8- and 16-bit encodings otherwise goes to the wide kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrterm_amd.engine import Engine
from tests.golden import prng
from oracle import c_chain, ref_chain
fs = 1024000
worst = 0.0
for code in 'bBhHiIfd':
    for n in (8192, 32768):
        for q in (2, 4):
            for iq in (False, True):
                for center in (0.0, 15700.0):
                    raw = prng.raw_bytes(code, n * 3, seed=5, amp=100 if code in 'bB' else 3000)
                    eng = Engine(fs, q, 'fm', code, n_samples=n, row_freqs=[center], apply_shift=bool(center), omega_out=60000, correct_iq=iq)
                    out = eng.process(raw)
                    shift = c_chain.shift_table([center], fs, n) if center else None
                    exp, _ = c_chain.process_stream(raw, code, n, q, 'fm', shift, ref_chain.cheby_sos(q), ref_chain.ellip_sos(fs // q, 60000), iq_inductance=50 / fs if iq else None, threads=4)
                    got = out.reshape(1, 3, n // q).transpose(1, 0, 2)
                    err = float(np.abs(got - exp).max())
                    worst = max(worst, err)
                    print(code, n, q, 'iq' if iq else '  ', int(center), eng.stage_a, f'{err:.3e}', '' if err < 1e-8 else '   <<<<<<<<')
                    eng.close()
print('worst', worst)
sys.exit(0 if worst < 1e-8 else 1)
