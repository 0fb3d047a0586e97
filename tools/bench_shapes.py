"""Forward latency / real-time factor of Aero.forward at other batch sizes and configs than bench.py's headline one
(same timing method: warm-up, K timed forwards between synchronisations).  python tools/bench_shapes.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import build_model  # noqa: E402


def run(m, B, L, lr_sr, steps=10, warmup=3):
    x = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        t0 = time.perf_counter()
        n = 0
        while n < warmup or time.perf_counter() - t0 < 0.3:      # also long enough for the clocks to come back up after host-side set-up
            m(x)
            n += 1
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, B * (L / lr_sr) / (ms / 1e3)


def main():
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
    full = build_model(meta, 'full').cuda()                       # aero_4-16_512_64
    eng = full._get_engine()
    for B, L in ((1, 8000), (8, 8000), (64, 8000), (16, 40000), (128, 8000)):
        eng.use_graph = False
        ms, rtf = run(full, B, L, 4000)
        eng.use_graph = True
        msg, rtfg = run(full, B, L, 4000)
        print(f'4->16 kHz nfft 512 hop 64   B={B:4d} L={L:6d} ({L / 4000:4.1f} s clips): eager {ms:8.3f} ms RTF {rtf:8.1f} | HIP graph {msg:8.3f} ms RTF {rtfg:8.1f}', flush=True)
    eng.use_graph = False
    cfg = dict(meta['full_cfg'])
    cfg.update(lr_sr=12000, hr_sr=48000, nfft=1024, hop_length=256)     # BASELINE config 4 (conf/experiment/aero_12-48_1024_256)
    torch.manual_seed(2036)
    from aero_amd import Aero
    wide = Aero(**cfg).eval().cuda()
    for B, L in ((1, 24000), (32, 24000)):
        ms, rtf = run(wide, B, L, 12000)
        print(f'12->48 kHz nfft 1024 hop 256 B={B:4d} L={L:6d} ({L / 12000:4.1f} s clips): {ms:8.3f} ms  RTF {rtf:9.1f}', flush=True)


if __name__ == '__main__':
    main()
