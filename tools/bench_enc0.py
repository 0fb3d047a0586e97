"""Micro-benchmark of the fused encoder-0 kernel (aero_enc0_fwd) at the bench shape: B=64, F=256 -> 64, T=501, C=M=48."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from aero_amd import _lib, pack
from aero_amd.engine import Ops
iters = int(os.environ.get('ITERS', 20))
ops = Ops(_lib.load())
B, Fq, T, Cc, M, K, stride, pad = 64, 256, 501, 48, 48, 8, 4, 2
dev = 'cuda'
g = torch.Generator().manual_seed(0)
xn = torch.randn(B, Fq, T, 2, generator=g).half().to(dev)
u = torch.randn(B, Fq, T, 2, generator=g).half().to(dev)
G = (torch.randn(B, 1, T, 3 * Cc, generator=g) * 0.5).half().to(dev)
P = dict(C=Cc, rs=torch.randn(Fq, generator=g).to(dev), a_re=torch.randn(Cc, generator=g).to(dev),
         a_im=torch.randn(Cc, generator=g).to(dev), bias=torch.randn(Cc, generator=g).to(dev))
taps, df, dt = pack.conv2d_taps(torch.randn(M, Cc, K, 1, generator=g) / (Cc * K) ** 0.5, pad, 0)
spec = pack.make_conv_spec(taps, torch.randn(M, generator=g), Cc, 0, df, dt, dev, fstride=stride, act=_lib.ACT_GELU)
Fo = (Fq + 2 * pad - K) // stride + 1
for _ in range(3):
    ops.enc0(xn, u, G, P, spec, Fo, stride, pad, _lib.ACT_GELU)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.enc0(xn, u, G, P, spec, Fo, stride, pad, _lib.ACT_GELU)
e1.record()
torch.cuda.synchronize()
print(f'enc0 B={B} F={Fq}->{Fo} T={T} C={Cc} M={M}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us', flush=True)
