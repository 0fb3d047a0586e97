"""Run-to-run determinism of the kernels that once returned different results on the MI355X only ("hardware-only
wrong results"): each kernel runs N times on the same inputs and every output must be bit-identical to the first run's
(the fp64 statistics sums are accumulated with atomics in arrival order: compared to 1e-12 instead), and the conv
output must also agree with an fp32 torch reference.
    python tools/dbg/determinism.py [--n 50] [--cases enc0,ftb_first,conv_stats8,conv_stats_ring,conv8]
AERO_CONV_RING=0 in the environment sends the wide convs to the 8-wave k_conv.h tiles (the `conv_stats8` case)."""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aero_amd import _lib, pack  # noqa: E402
from aero_amd.engine import Ops  # noqa: E402

dev = 'cuda'


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def repeat(name, fn, n, ref=None, tol=2e-3):
    """fn() -> (tensor to compare bitwise, optional fp64 stats)"""
    y0, s0 = fn()
    torch.cuda.synchronize()
    y0 = y0.clone()
    s0 = None if s0 is None else s0.clone()
    bad, sbad = 0, 0.0
    for _ in range(n - 1):
        y, s = fn()
        torch.cuda.synchronize()
        nd = int((y.view(torch.int16) != y0.view(torch.int16)).sum())
        if nd:
            bad += 1
            if bad == 1:
                idx = (y.view(torch.int16) != y0.view(torch.int16)).nonzero()
                print(f'  {name}: {nd} differing elements, first {idx[:4].tolist()}')
        if s is not None:
            sbad = max(sbad, float(((s - s0).abs() / s0.abs().clamp_min(1e-30)).max()))
    err = None if ref is None else rel(y0.float().cpu(), ref)
    ok = bad == 0 and sbad < 1e-12 and (err is None or err < tol)
    print(f'{name}: {n} runs, {bad} differ from run 0, stats max rel diff {sbad:.1e}, vs reference {err}  -> {"OK" if ok else "FAIL"}',
          flush=True)
    return ok


def conv_case(ops, Cin, Cout, Fq, T, B, G, stats, seed=5):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).half().float()
    b = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, Fq, T, generator=g).half().float()
    taps, df, dt = pack.conv2d_taps(w, 1, 1)
    spec = pack.make_conv_spec(taps, b, Cin, 0, df, dt, dev)
    xcl = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).contiguous()

    def fn():
        if not stats:
            return ops.conv(spec, xcl, None, B, Fq, Fq, T), None
        st = ops.new_stats(B, Fq, G, False, dev)
        y = ops.conv(spec, xcl, None, B, Fq, Fq, T, stat=dict(mode=1, stats=st, G=G, per_row=False))
        return y, st
    return fn, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=50)
    ap.add_argument('--cases', default='enc0,ftb_first,conv_stats,conv')
    a = ap.parse_args()
    ops = Ops(_lib.load())
    ok = True
    g = torch.Generator().manual_seed(0)
    B, Fq, T, Cc = 8, 256, 501, 48
    xn = torch.randn(B, Fq, T, 2, generator=g).half().to(dev)
    u = torch.randn(B, Fq, T, 2, generator=g).half().to(dev)
    f32 = lambda n: torch.randn(n, generator=g).float().to(dev)  # noqa: E731
    for case in a.cases.split(','):
        if case == 'ftb_first':
            gate = torch.rand(B, T, Cc, generator=g).half().to(dev)
            img = torch.zeros(128, 64)
            img[:Cc, :Cc] = torch.randn(Cc, Cc, generator=g) * 0.1
            P = dict(C=Cc, w2a=img.half().to(dev), p0=f32(Cc), p1=f32(Cc), pb=f32(Cc), rs=f32(Fq), a_re=f32(Cc), a_im=f32(Cc),
                     bias=f32(Cc))
            ok &= repeat('ftb_first', lambda: (ops.ftb_first(xn, u, gate, P), None), a.n)
        elif case == 'enc0':
            G3 = (torch.randn(B, 1, T, 3 * Cc, generator=g) * 0.5).half().to(dev)
            P = dict(C=Cc, rs=f32(Fq), a_re=f32(Cc), a_im=f32(Cc), bias=f32(Cc))
            taps, df, dt = pack.conv2d_taps(torch.randn(48, Cc, 8, 1, generator=g) / (Cc * 8) ** 0.5, 2, 0)
            spec = pack.make_conv_spec(taps, torch.randn(48, generator=g), Cc, 0, df, dt, dev, fstride=4, act=_lib.ACT_GELU)
            ok &= repeat('enc0', lambda: (ops.enc0(xn, u, G3, P, spec, 64, 4, 2, _lib.ACT_GELU), None), a.n)
        elif case in ('conv_stats', 'conv'):
            for (Cin, Cout, F2, G) in ((128, 256, 16, 4), (192, 384, 8, 4), (96, 768, 8, 4)):
                fn, ref = conv_case(ops, Cin, Cout, F2, 501, 2, G, case == 'conv_stats')
                fn()
                name = ops.lib.cdll.aero_last_kernel_name().decode()
                ok &= repeat(f'{case} {Cin}->{Cout} [{name}]', fn, a.n, ref)
    print('ALL OK' if ok else 'SOME FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
