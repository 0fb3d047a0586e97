"""Micro-benchmark of the LocalState attention core at the model's shapes (A/B via AERO_ATTN_RES)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aero_amd import _lib
from aero_amd.engine import Ops
ops = Ops(_lib.load())
for (R, T, C) in ((512, 501, 48), (256, 501, 96)):
    qkvd = torch.randn(R, T, 3 * C + 16, device='cuda').half()
    for _ in range(3):
        ops.localstate(qkvd, R, T, C, 4, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(int(os.environ.get("ITERS", 20))):
        ops.localstate(qkvd, R, T, C, 4, 4)
    e1.record()
    torch.cuda.synchronize()
    print(f'attn R={R} T={T} C={C}: {e0.elapsed_time(e1) / int(os.environ.get("ITERS", 20)) * 1e3:.1f} us', flush=True)
