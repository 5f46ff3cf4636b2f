"""Test infrastructure: builds tests/standin_rccl/standin_rccl.cpp (see its header) and returns the library's path."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libstandin_rccl.so')


def library() -> str:
    src = os.path.join(HERE, 'standin_rccl.cpp')
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call([hipcc, '-O2', '-shared', '-fPIC', '-std=c++17', '-x', 'hip', '--offload-arch=gfx950', src, '-o', LIB, '-lrt'])
    return LIB
