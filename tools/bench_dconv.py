"""Micro-benchmark of the DConv row kernel at the bench shapes (encoder levels 0 and 1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aero_amd import _lib, pack
from aero_amd.engine import Ops
iters = int(os.environ.get('ITERS', 20))
ops = Ops(_lib.load())
dev = 'cuda'
g = torch.Generator().manual_seed(0)
for (Cc, Fq) in ((48, 64), (96, 16)):
    B, T, hid = 64, 501, Cc // 4
    x = torch.randn(B, Fq, T, Cc, generator=g).half().to(dev)
    layers = []
    for l in range(2):
        r = lambda *s: torch.randn(*s, generator=g)
        L = pack.dconv_row_layer(r(hid, Cc, 3) / (3 * Cc) ** 0.5, r(hid), 1 + 0.1 * r(hid), 0.1 * r(hid), r(2 * Cc, hid) / hid ** 0.5,
                                 r(2 * Cc), 1 + 0.1 * r(2 * Cc), 0.1 * r(2 * Cc), r(Cc) * 0.3, 2 ** l, dev)
        L['snake_a'] = (0.5 + torch.rand(Fq, generator=g)).to(dev)
        layers.append(L)
    for _ in range(3):
        ops.dconv_row(x, layers, _lib.ACT_SNAKE, Fq)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.dconv_row(x, layers, _lib.ACT_SNAKE, Fq)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f'dconv_row C={Cc} F={Fq}: {us:.1f} us  ({2 * x.numel() * 2 / us / 1e6:.2f} TB/s)', flush=True)
