"""Build tests/emu/libaero_emu.so: the kernels compiled against the CPU emulation of HIP (tests only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, 'aero_amd', 'csrc')
OUT = os.path.join(HERE, 'libaero_emu.so')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False):
    deps = [os.path.join(SRC, f) for f in os.listdir(SRC)] + [os.path.join(HERE, 'hip_emu.h'),
                                                              os.path.join(HERE, 'hip_emu.cpp'),
                                                              os.path.join(ROOT, 'include', 'aero_hip.h')]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(deps):
        return OUT
    cmd = [CLANG, '-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-shared', '-DAERO_EMU', '-I', HERE, '-I', SRC,
           '-o', OUT, os.path.join(SRC, 'aero_hip.hip'), os.path.join(HERE, 'hip_emu.cpp'), '-lpthread']
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    print(build(force=True))
