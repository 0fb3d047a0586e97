"""List every conv launch of Aero.forward (full config) with the kernel aero_conv_kernel_name dispatches it to.
Runs on the CPU emulation (test double) -- the dispatch is a pure function of the descriptor -- at a small batch/length,
with the descriptors rescaled to the bench shape (B, T) before the name query.  python tools/list_convs.py [B] [T]"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import build_model  # noqa: E402
from aero_amd import _lib  # noqa: E402
from aero_amd.engine import HipEngine  # noqa: E402
from emu.build_emu import build  # noqa: E402


def main():
    Bq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    Tq = int(sys.argv[2]) if len(sys.argv) > 2 else 501
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
    lib = _lib.load(build())
    m = build_model(meta, 'full')
    eng = HipEngine(m, lib=lib)
    object.__setattr__(m, '_engine', eng)
    rows = []
    real = lib.cdll.aero_conv_fwd

    def spy(dref, stream):
        d = dref._obj
        b, t = d.B, d.T
        d.B, d.T = Bq, Tq
        buf = C.create_string_buffer(128)
        lib.cdll.aero_conv_kernel_name(C.byref(d), buf, 128)
        d.B, d.T = b, t
        cin = d.C1 + (d.C0 if d.src0 else 0)
        rows.append((buf.value.decode(), d.M, cin, d.ntaps, d.Fin, d.Fout, d.transposed, d.fstride, d.act, d.stat_mode))
        return real(dref, stream)

    class Shim:
        def __getattr__(self, n):
            return spy if n == 'aero_conv_fwd' else getattr(lib.cdll, n)

    orig_call = lib.call

    def call(name, *args):
        if name == 'aero_conv_fwd':
            lib.check(spy(*args), name)
        else:
            orig_call(name, *args)

    lib.call = call
    with torch.no_grad():
        m(torch.randn(1, 1, 1000))
    print(f'{"kernel":42s} {"M":>4s} {"Cin":>4s} taps {"Fin":>4s} {"Fout":>4s} tr fs act sm   GFLOP at B={Bq} T={Tq}')
    for r in rows:
        gf = 2.0 * Bq * r[5] * Tq * r[1] * r[3] * r[2] / 1e9
        print(f'{r[0]:42s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:4d} {r[5]:4d} {r[6]:2d} {r[7]:2d} {r[8]:3d} {r[9]:2d}   {gf:8.2f}')


if __name__ == '__main__':
    main()
