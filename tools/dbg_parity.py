import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import BY_NAME, case_raw, load_golden
from tests.test_gpu_parity import _engine
z, _ = load_golden()
for name in sys.argv[1:]:
    case = BY_NAME[name]
    eng = _engine(case)
    out, y = eng.process_debug(case_raw(case))
    exp = z[f'{name}/z']; nch, rows, m = exp.shape
    y0 = z[f'{name}/y0'].astype(np.complex128)
    gy0 = y.reshape(rows, nch, m)[:, 0, :]
    e = np.abs(gy0 - y0)[0] / np.abs(y0).max()
    print(name, 'max', e.max(), 'argmax', e.argmax())
    print(' err by 16-block tile:', np.array2string(e.reshape(-1, 16).max(axis=1), precision=2))
    print(' err blocks 40..100:', np.array2string(e[40:100], precision=1, max_line_width=200))
