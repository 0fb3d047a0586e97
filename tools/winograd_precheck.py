"""Zero-GPU pre-check of Winograd F(2,3) ALONG THE TIME AXIS for the four decoder 3x3 'rewrite' convolutions (aero.py:179; 68 % of the
FLOPs): would fp16-rounded TRANSFORMED operands keep the complex spectrogram inside the 1e-3 bar?  

The CPU oracle (oracle/aero_oracle.py) runs the full and the stress-full model with ONLY those four convolutions replaced by
  direct:    operands rounded to fp16, fp32 accumulation, fp16 output          (what the ring kernel computes today)
  winograd:  V = B^T d (from fp16 d, rounded to fp16), U = G g (from fp32 g, rounded to fp16), fp32 accumulation of the four
             component products over (frequency tap, channel), output transform in fp32, fp16 output
and prints the rel-L2 of the spectrogram against the reference's golden for each.  python tools/winograd_precheck.py"""
import json
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import aero_oracle as O  # noqa: E402
from conftest import build_model, load_npz, rel_l2  # noqa: E402


def r16(t):
    return t.half().float()


class Patched:
    def __init__(self, mode):
        self.mode = mode

    def __getattr__(self, name):
        return getattr(TF, name)

    def conv2d(self, x, w, b=None, stride=1, padding=0, **k):
        if self.mode == 'fp32' or tuple(w.shape[2:]) != (3, 3):
            return TF.conv2d(x, w, b, stride=stride, padding=padding, **k)
        if self.mode == 'direct':
            return r16(TF.conv2d(r16(x), r16(w), b, padding=padding))
        # F(2,3) along time; frequency taps stay a plain 3-tap contraction
        B, Cc, Fq, T = x.shape
        d = TF.pad(r16(x), (1, 1 + (T & 1)))                     # time padding 1 (+1 to an even number of outputs)
        n = (T + 1) // 2
        d0, d1, d2, d3 = (d[..., j:j + 2 * n:2] for j in range(4))
        V = [r16(d0 - d2), r16(d1 + d2), r16(d2 - d1), r16(d1 - d3)]
        g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]
        U = [r16(g0), r16((g0 + g1 + g2) * 0.5), r16((g0 - g1 + g2) * 0.5), r16(g2)]
        M = [TF.conv2d(V[j], U[j].unsqueeze(-1), None, padding=(1, 0)) for j in range(4)]
        y = torch.stack([M[0] + M[1] + M[2], M[1] - M[2] - M[3]], -1).reshape(B, w.shape[0], Fq, 2 * n)[..., :T]
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
        return r16(y)


def main():
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
    torch.set_num_threads(8)
    for which, io_name in (('full', 'full_io.npz'), ('stress_full', 'stress_full_io.npz')):
        m = build_model(meta, which)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        cfg = {**O.DEFAULT_CFG, **meta['full_cfg']}
        io = load_npz(io_name)
        x = torch.randn(2, 1, 8000, generator=torch.Generator().manual_seed(0))
        gold = io['spec']
        out = {}
        for mode in ('fp32', 'direct', 'winograd'):
            saved = O.F
            O.F = Patched(mode)
            try:
                with torch.no_grad():
                    _, s = O.aero_forward(sd, cfg, x[:1], return_spec=True, fast=True)
            finally:
                O.F = saved
            out[mode] = rel_l2(s, gold[:1])
        extra = (max(out['winograd'] ** 2 - out['direct'] ** 2, 0.0)) ** 0.5
        print(f'{which:12s} spectrogram rel-L2 vs the reference golden (clip 0): fp32 oracle {out["fp32"]:.2e} | decoder 3x3 convs with fp16 operands, '
              f'direct {out["direct"]:.2e} | Winograd F(2,3) along time {out["winograd"]:.2e}   (excess over direct, in quadrature: {extra:.2e}; bar 1e-3)', flush=True)


if __name__ == '__main__':
    main()
