"""Timing of the backward kernels at the first decoder layer's shapes (B = 64 clips of 2 s): weight gradient of the 3x3 rewrite
conv (aero.py:172), its data gradient on the forward kernel, GroupNorm + GLU backward.  python tools/bench_bwd.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aero_amd import _lib, backward as bw, pack  # noqa: E402
from aero_amd.engine import Ops  # noqa: E402


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    lib = _lib.load()
    ops = Ops(lib)
    dev = 'cuda'
    B = int(os.environ.get('B', 64))
    for name, Cin, Cout, Fr, T in (('D0 rewrite 3x3', 768, 1536, 4, 501), ('D1 rewrite 3x3', 384, 768, 8, 501), ('D3 rewrite 3x3', 96, 192, 64, 501)):
        w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
        x = torch.randn(B, Fr, T, Cin, device=dev).half()
        dy = torch.randn(B, Fr, T, Cout, device=dev).half()
        _, df, dt = pack.conv2d_taps(w, 1, 1)
        flops = 2.0 * B * Fr * T * Cin * Cout * 9
        ms = timeit(lambda: bw.conv_wgrad(ops, dy, x, df, dt))
        print(f'{name}: wgrad {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF/s', flush=True)
        spec = bw.dgrad_conv2d(w, 1, 1, dev)
        ms = timeit(lambda: ops.conv(spec, dy, None, B, Fr, Fr, T))
        print(f'{name}: dgrad {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF/s', flush=True)
        gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        ops.norm_act(dy, 4, 0, gamma, beta, _lib.ACT_GLU)
        stats = ops._last_stats
        g2 = torch.randn(B, Fr, T, Cout // 2, device=dev).half()
        ms = timeit(lambda: bw.norm_bwd(ops, dy, g2, stats, 4, 0, gamma, beta, _lib.ACT_GLU))
        nbytes = dy.numel() * 2 * 3 + g2.numel() * 2 * 2
        print(f'{name}: GroupNorm+GLU backward {ms:8.3f} ms {nbytes / ms / 1e6:7.1f} GB/s', flush=True)

    # iSTFT backward (prep -> forward STFT kernel -> pack) at the model's output geometry, and the FTB's training-mode BatchNorm + ReLU
    nfft, hop, T = 512, 64, 501
    w = torch.hann_window(nfft, device=dev)
    env = torch.zeros(nfft + hop * (T - 1), dtype=torch.float64)
    for t in range(T):
        env[t * hop:t * hop + nfft] += (w.cpu() * w.cpu()).double()
    inv_env = (1 / env).float().to(dev)
    dyw = torch.randn(B, hop * (T - 1), device=dev)
    ms = timeit(lambda: bw.istft_bwd(ops, dyw, nfft, hop, w, inv_env, T))
    print(f'iSTFT backward (B={B}, {hop * (T - 1)} samples): {ms * 1e3:8.1f} us', flush=True)
    xb = torch.randn(B, 64, T, 48, device=dev).half()
    gamma, beta = torch.ones(48, device=dev), torch.zeros(48, device=dev)
    ops.norm_act(xb, 48, 2, gamma, beta, _lib.ACT_RELU)
    stats = ops._last_stats
    gb = torch.randn_like(xb)
    ms = timeit(lambda: bw.norm_bwd(ops, xb, gb, stats, 48, 2, gamma, beta, _lib.ACT_RELU))
    print(f'BatchNorm(train) + ReLU backward [B,48,64,{T}]: {ms:8.3f} ms {xb.numel() * 2 * 5 / ms / 1e6:7.1f} GB/s', flush=True)


if __name__ == '__main__':
    main()
