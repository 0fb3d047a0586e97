"""Consecutive batches on alternating HIP streams (a serving loop that does not wait for batch i before it enqueues batch i + 1) against the
engine's own schedule (two half-batches of ONE batch on two streams, joined before the iSTFT).  K forwards of the bench workload are timed
between two device synchronisations in every mode; outputs are compared with the single-stream result.  python tools/bench_pipelined.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import FULL_CFG  # noqa: E402
from aero_amd import Aero  # noqa: E402


def run(model, x, K, outer, inner):
    eng = model._get_engine()
    eng.streams = inner
    cur = torch.cuda.current_stream()
    sts = [torch.cuda.Stream() for _ in range(outer)] if outer > 1 else [cur]
    ys = []
    with torch.no_grad():
        for i in range(6 * max(1, outer)):
            with torch.cuda.stream(sts[i % len(sts)]):
                model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            st = sts[i % len(sts)]
            if outer > 1:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                y = model(x)
            if i >= K - len(sts):
                ys.append(y)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / K * 1e3, ys, host / K * 1e3


def main():
    torch.manual_seed(2036)
    dev = torch.device('cuda', 0)
    model = Aero(**FULL_CFG).eval().to(dev)
    B = int(os.environ.get('B', 64))
    x = torch.randn(B, 1, 8000, generator=torch.Generator().manual_seed(1000)).to(dev)
    K = int(os.environ.get('K', 40))
    _, ref, _ = run(model, x, 4, 1, 1)
    for outer, inner in ((1, 1), (1, 2), (2, 1), (3, 1), (2, 2), (4, 1), (1, 2), (2, 1)):
        ms, ys, host = run(model, x, K, outer, inner)
        same = all(torch.equal(y, ref[0]) for y in ys)
        print(f'{outer} batch stream(s) x {inner} sub-batch stream(s): {ms:7.3f} ms per batch (host enqueue {host:6.3f} ms)   outputs bit-equal to single stream: {same}', flush=True)
    from aero_amd.pipeline import BatchPipeline
    for depth in (3, 3):
        pipe = BatchPipeline(model, depth=depth)
        for _ in range(2 * depth):
            pipe.submit(x)
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            t = pipe.submit(x)
        host = time.perf_counter() - t0
        pipe.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'BatchPipeline depth {depth}: {dt / K * 1e3:7.3f} ms per batch (host enqueue {host / K * 1e3:6.3f} ms)', flush=True)


if __name__ == '__main__':
    main()
