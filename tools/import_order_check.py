"""Diagnostic (GPU box): library first, torch second in one process (one HIP runtime per process, sdrterm_amd/_native.py)."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))


def main():
    from sdrterm_amd.engine import Engine
    import numpy as np
    eng = Engine(1024000, 64, 'fm', 'h', row_freqs=[15000], omega_out=5000)
    out = eng.process(np.zeros(32768 * 2, dtype=np.int16), 1)
    print('engine first ok', eng.stage_a, out.shape)
    import torch
    x = torch.ones(8, device='cuda') * 2
    print('torch after ok', float(x.sum()), torch.cuda.get_device_name(0))


if __name__ == '__main__':
    main()
