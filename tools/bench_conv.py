"""Micro-benchmark of the decoder 3x3 'rewrite' convolutions (the dominant kernel) through the C ABI.
    python tools/bench_conv.py [--layers d0,d1,d2,d3] [--batch 64] [--iters 20]
Prints per-layer HIP-event time and executed TFLOP/s.  Used for A/B of kernel variants and under rocprofv3 --pmc."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aero_amd import _lib, pack  # noqa: E402
from aero_amd.engine import Ops  # noqa: E402

LAYERS = {  # name: (Cx, Cskip, M, F, null_x, glu)
    'd0': (384, 384, 1536, 4, True, False), 'd1': (192, 192, 768, 8, False, False),
    'd2': (96, 96, 384, 16, False, True), 'd3': (48, 48, 192, 64, False, True),
}
PW = {  # pointwise (1x1) shapes of the path: name: (C, M, F)
    'pw768': (96, 768, 4), 'pw384': (48, 384, 8), 'pw96': (48, 96, 64), 'pw192': (24, 192, 16), 'pw768k': (384, 768, 4),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default=os.environ.get('LAYERS', 'd0,d1,d2,d3'))
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=int(os.environ.get('ITERS', 20)))
    ap.add_argument('--T', type=int, default=501)
    ap.add_argument('--lib', default=None, help='alternative build of the C-ABI library (profiling variants)')
    ap.add_argument('--zeros', action='store_true', help='zero-filled operands: the same instruction stream at lower switching power (DVFS give-back, MI355X_MICROARCH.md)')
    ap.add_argument('--race', type=int, default=0, help='repeat each launch this many times and require bit-identical outputs')
    a = ap.parse_args()
    dev = 'cuda'
    ops = Ops(_lib.load(a.lib))
    for name in a.layers.split(','):
        if name in PW:
            C, M, F = PW[name]
            w = torch.randn(M, C, 1, 1) * 0.05
            taps, df, dt = pack.conv2d_taps(w, 0, 0)
            spec = pack.make_conv_spec(taps, torch.zeros(M), C, 0, df, dt, dev)
            x = torch.randn(a.batch, F, a.T, C, device=dev).half()
            out = torch.empty(a.batch, F, a.T, M, device=dev, dtype=torch.float16)
            for _ in range(3):
                ops.conv(spec, x, None, a.batch, F, F, a.T, dst=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv(spec, x, None, a.batch, F, F, a.T, dst=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            nb = x.numel() * 2 + out.numel() * 2
            print(f'{name}: {ms * 1e3:8.1f} us  {nb / ms / 1e9:6.2f} TB/s  (C={C} M={M} rows={a.batch * F}, {nb / 1e6:.0f} MB)', flush=True)
            continue
        C0, C1, M, F, null0, glu = LAYERS[name]
        w = torch.randn(M, C0 + C1, 3, 3) * (0.0 if a.zeros else 0.02)
        taps, df, dt = pack.conv2d_taps(w, 1, 1)
        spec = pack.make_conv_spec(taps, torch.zeros(M), C0, C1, df, dt, dev, act=_lib.ACT_GLU if glu else _lib.ACT_NONE)
        x = None if null0 else torch.randn(a.batch, F, a.T, C0, device=dev).half()
        sk = torch.randn(a.batch, F, a.T, C1, device=dev).half()
        if a.zeros:
            sk.zero_()
            if x is not None:
                x.zero_()
        out = torch.empty(a.batch, F, a.T, M // 2 if glu else M, device=dev, dtype=torch.float16)
        for _ in range(3):
            ops.conv(spec, x, sk, a.batch, F, F, a.T, dst=out)
        torch.cuda.synchronize()
        if a.race:
            ref = out.clone()
            bad = 0
            for _ in range(a.race):
                out.fill_(float('nan'))
                ops.conv(spec, x, sk, a.batch, F, F, a.T, dst=out)
                bad += int((out != ref).sum())
            print(f'{name}: race screen x{a.race}: {bad} differing elements; finite={bool(torch.isfinite(out).all())}', flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv(spec, x, sk, a.batch, F, F, a.T, dst=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        cin = C1 + (0 if null0 else C0)
        fl = 2.0 * a.batch * F * a.T * M * 9 * cin
        print(f'{name}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s executed  ({fl / 1e9:.1f} GF, M={M} K=9x{cin} rows={a.batch * F})', flush=True)


if __name__ == '__main__':
    main()
