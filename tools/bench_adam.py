"""Micro-benchmark of the fused Adam step at the generator's size (19.43 M parameters): 28 bytes of HBM traffic per parameter."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aero_amd.optim import FlatAdam
n = 19_432_958
p = [torch.nn.Parameter(torch.randn(n, device='cuda'))]
opt = FlatAdam(p)
p[0].grad.normal_()
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    opt.step()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f'adam step, {n} parameters: {us:.1f} us  ({28 * n / us / 1e6:.2f} TB/s of 8.0)')
