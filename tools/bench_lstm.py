"""Micro-benchmark of the recurrent LSTM kernel at the model's shapes (for PMC profiling / A-B of variants).
    python tools/bench_lstm.py [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aero_amd import _lib, pack  # noqa: E402
from aero_amd.engine import Ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=int(os.environ.get("ITERS", 10)))
    a = ap.parse_args()
    dev = 'cuda'
    lib = _lib.load()
    ops = Ops(lib)
    for H, R in ((48, 512), (96, 256)):
        T, W, S = 501, 200, 100
        nframes = 6
        nseq = R * nframes
        sd = {}
        k = 1.0 / H ** 0.5
        for l in range(2):
            for sfx in ('', '_reverse'):
                inp = H if l == 0 else 2 * H
                sd[f'l.weight_ih_l{l}{sfx}'] = (torch.rand(4 * H, inp) * 2 - 1) * k
                sd[f'l.weight_hh_l{l}{sfx}'] = (torch.rand(4 * H, H) * 2 - 1) * k
                sd[f'l.bias_ih_l{l}{sfx}'] = (torch.rand(4 * H) * 2 - 1) * k
                sd[f'l.bias_hh_l{l}{sfx}'] = (torch.rand(4 * H) * 2 - 1) * k
        packs = [pack.pack_lstm_layer(lib, sd, 'l', l, H, dev) for l in range(2)]
        x = torch.randn(R, T, H, device=dev).half()
        out0 = torch.empty(nseq, W, 2 * H, device=dev, dtype=torch.float16)
        out1 = torch.empty(R, T, 2 * H, device=dev, dtype=torch.float16)

        def run():
            ops.lstm(None, None, packs[0][2], H, nseq, W, 1, 0, nframes, S, T, out0, x=x, fused=packs[0][3])
            ops.lstm(None, None, packs[1][2], H, nseq, W, 0, 1, nframes, S, T, out1, x=out0, fused=packs[1][3])
        run()
        torch.cuda.synchronize()
        for layer in (0, 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                if layer == 0:
                    ops.lstm(None, None, packs[0][2], H, nseq, W, 1, 0, nframes, S, T, out0, x=x, fused=packs[0][3])
                else:
                    ops.lstm(None, None, packs[1][2], H, nseq, W, 0, 1, nframes, S, T, out1, x=out0, fused=packs[1][3])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            print(f'H={H} layer{layer}: {us:8.1f} us  ({us / W:.3f} us/step, {nseq * 2 // 16} blocks)', flush=True)


if __name__ == '__main__':
    main()
