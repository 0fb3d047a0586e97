"""aero_conv_wgrad at the shapes of one training step of BASELINE config 5's per-GPU share (B = 2 clips, T = 1724 frames; the 16 most
expensive of the 92 calls in profiles/r03_train_profile_b2.txt), each alone on an idle chip: time per call (kernel + finish), executed
TF/s, and the HBM floor of reading dy and x once.  python tools/bench_wgrad.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aero_amd import _lib, backward as bw  # noqa: E402
from aero_amd.engine import Ops  # noqa: E402

T3 = ([-1, -1, -1, 0, 0, 0, 1, 1, 1], [-1, 0, 1, -1, 0, 1, -1, 0, 1])
# (dy shape, x shape, (df, dt), fstride, calls per step)
SHAPES = [((2, 8, 1724, 768), (2, 8, 1724, 192), T3, 1, 2),
          ((2, 64, 1724, 192), (2, 64, 1724, 48), T3, 1, 2),
          ((2, 8, 1724, 48), (2, 8, 1724, 48), ([0], [0]), 1, 2),
          ((2, 4, 1724, 1536), (2, 4, 1724, 384), T3, 1, 1),
          ((2, 256, 1724, 48), (2, 256, 1724, 48), ([0], [0]), 1, 2),
          ((144, 1, 200, 384), (144, 1, 200, 96), ([0], [0]), 1, 8),
          ((2, 16, 1724, 384), (2, 16, 1724, 96), T3, 1, 2),
          ((2, 64, 1724, 48), (2, 256, 1724, 48), (list(range(-2, 6)), [0] * 8), 4, 1),
          ((288, 1, 200, 192), (288, 1, 200, 48), ([0], [0]), 1, 8),
          ((2, 64, 1724, 16), (2, 64, 1724, 48), ([0, 0, 0], [-1, 0, 1]), 1, 2),
          ((2, 64, 1724, 96), (2, 256, 1724, 8), (list(range(-2, 6)), [0] * 8), 4, 1),
          ((2, 4, 1724, 768), (2, 14, 1724, 192), (list(range(0, 8)), [0] * 8), 2, 1),
          ((2, 256, 1724, 8), (2, 256, 1724, 48), ([0], [0]), 1, 1),
          ((2, 16, 1724, 192), (2, 64, 1724, 48), (list(range(-2, 6)), [0] * 8), 4, 1),
          ((2, 256, 1724, 8), (2, 256, 1724, 8), ([0], [0]), 1, 1),
          ((2, 64, 1724, 48), (2, 64, 1724, 48), ([0], [0]), 1, 2)]


def main():
    lib = _lib.load()
    ops = Ops(lib)
    total = 0.0
    for dys, xs, (df, dt), fstride, calls in SHAPES:
        dy = torch.randn(*dys, device='cuda').half()
        x = torch.randn(*xs, device='cuda').half()
        fn = lambda: bw.conv_wgrad(ops, dy, x, df, dt, fstride=fstride)      # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        flops = 2.0 * dys[0] * dys[1] * dys[2] * dys[3] * xs[3] * len(df)
        floor = (dy.numel() + x.numel()) * 2 / 5.0e12
        total += best * calls
        print(f'dy{dys} x{xs} taps{len(df)}: {best * 1e3:7.1f} us  {flops / best / 1e9:6.1f} TF/s  floor {floor * 1e6:5.1f} us  x{calls}', flush=True)
    print(f'sum over the step (calls weighted): {total:.3f} ms')


if __name__ == '__main__':
    main()
