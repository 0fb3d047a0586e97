"""Debug: run-to-run determinism of aero_ftb_first_fwd on random inputs; prints where two runs differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aero_amd import _lib  # noqa: E402
from aero_amd.engine import Ops  # noqa: E402

dev = 'cuda'
ops = Ops(_lib.load())
B, F, T, Cc = 64, 256, 501, 48
g = torch.Generator().manual_seed(0)
xn = torch.randn(B, F, T, 2, generator=g).half().to(dev)
u = torch.randn(B, F, T, 2, generator=g).half().to(dev)
gate = torch.rand(B, T, Cc, generator=g).half().to(dev)
img = torch.zeros(128, 64)
img[:Cc, :Cc] = torch.randn(Cc, Cc, generator=g) * 0.1
f32 = lambda n: torch.randn(n, generator=g).float().to(dev)  # noqa: E731
P = dict(C=Cc, w2a=img.half().to(dev), p0=f32(Cc), p1=f32(Cc), pb=f32(Cc), rs=f32(F), a_re=f32(Cc), a_im=f32(Cc), bias=f32(Cc))
outs = []
for _ in range(4):
    o = ops.ftb_first(xn, u, gate, P)
    torch.cuda.synchronize()
    outs.append(o.clone())
for k in range(1, 4):
    diff = (outs[k] != outs[0])
    n = int(diff.sum())
    print(f'run {k} vs 0: {n} differing elements of {outs[0].numel()}')
    if n:
        idx = diff.nonzero()[:12].tolist()
        print('  first differing (b,f,t,c):', idx)
        dd = diff.nonzero()
        print('  t%128 histogram (top):', torch.bincount(dd[:, 2] % 128, minlength=128).topk(5))
        print('  t//128:', torch.bincount(dd[:, 2] // 128, minlength=4).tolist(), ' c:', torch.bincount(dd[:, 3], minlength=Cc).tolist())
        a, b_ = outs[0][diff][:6].tolist(), outs[k][diff][:6].tolist()
        print('  values', a, b_)
