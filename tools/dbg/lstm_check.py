"""LSTM ring kernel: run-to-run determinism, row-permutation invariance, and distance to another build (AERO_OLD_LIB)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aero_amd import _lib, pack
from aero_amd.engine import Ops


def run(lib, H, R, x, sd):
    ops = Ops(lib)
    T, W, S, nframes = 501, 200, 100, 6
    nseq = R * nframes
    packs = [pack.pack_lstm_layer(lib, sd, 'l', l, H, 'cuda') for l in range(2)]
    out0 = torch.zeros(nseq, W, 2 * H, device='cuda', dtype=torch.float16)
    out1 = torch.zeros(R, T, 2 * H, device='cuda', dtype=torch.float16)
    ops.lstm(None, None, packs[0][2], H, nseq, W, 1, 0, nframes, S, T, out0, x=x, fused=packs[0][3])
    ops.lstm(None, None, packs[1][2], H, nseq, W, 0, 1, nframes, S, T, out1, x=out0, fused=packs[1][3])
    torch.cuda.synchronize()
    return out0.clone(), out1.clone()


lib = _lib.load()
old = _lib.load(os.environ['AERO_OLD_LIB']) if os.environ.get('AERO_OLD_LIB') else None
for H, R in ((48, 512), (96, 256), (48, 37)):
    g = torch.Generator().manual_seed(H)
    k = 1.0 / H ** 0.5
    sd = {}
    for l in range(2):
        for sfx in ('', '_reverse'):
            inp = H if l == 0 else 2 * H
            sd[f'l.weight_ih_l{l}{sfx}'] = (torch.rand(4 * H, inp, generator=g) * 2 - 1) * k
            sd[f'l.weight_hh_l{l}{sfx}'] = (torch.rand(4 * H, H, generator=g) * 2 - 1) * k
            sd[f'l.bias_ih_l{l}{sfx}'] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
            sd[f'l.bias_hh_l{l}{sfx}'] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
    x = torch.randn(R, 501, H, generator=g).half().cuda()
    a0, a1 = run(lib, H, R, x, sd)
    b0, b1 = run(lib, H, R, x, sd)
    perm = torch.randperm(R, generator=g).cuda()
    p0, p1 = run(lib, H, R, x[perm].contiguous(), sd)
    p0v = p0.view(R, 6, 200, 2 * H)
    a0v = a0.view(R, 6, 200, 2 * H)
    msg = f'H={H} R={R}: repeat equal {torch.equal(a0, b0)} / {torch.equal(a1, b1)}; permuted rows equal {torch.equal(p0v, a0v[perm])} / {torch.equal(p1, a1[perm])}'
    if not torch.equal(p1, a1[perm]):
        d = (p1.float() - a1[perm].float()).abs()
        rows = (d.reshape(R, -1).amax(1) > 0).nonzero().flatten().tolist()
        msg += f'; max diff {float(d.max()):.3e}, rows differing {len(rows)} (first {rows[:6]}), layer-0 rows differing {int(((p0v.float() - a0v[perm].float()).abs().reshape(R, -1).amax(1) > 0).sum())}'
    if old is not None:
        o0, o1 = run(old, H, R, x, sd)
        rel = float((a1.float() - o1.float()).norm() / o1.float().norm())
        msg += f'; vs other build rel {rel:.2e}'
    print(msg, flush=True)
