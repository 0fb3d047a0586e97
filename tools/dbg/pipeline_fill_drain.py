"""Where do the ~6 ms go that a timed region of K batches costs beyond K x the steady-state period?  The serving loop from an EMPTY pipeline:
completion time of every batch (event on its stream) relative to the start, for K = 20 and K = 40; optional stagger of the first batches
(STAGGER=stage: batch i + 1 of the first `depth` starts when batch i has passed that stage of its forward).
    python tools/dbg/pipeline_fill_drain.py"""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import FULL_CFG  # noqa: E402
from aero_amd import Aero  # noqa: E402
from aero_amd.pipeline import BatchPipeline  # noqa: E402


def main():
    torch.manual_seed(2036)
    dev = torch.device('cuda', 0)
    model = Aero(**FULL_CFG).eval().to(dev)
    x = torch.randn(64, 1, 8000, generator=torch.Generator().manual_seed(1000)).to(dev)
    kw = {}
    if os.environ.get('STAGGER'):
        kw['stagger'] = float(os.environ['STAGGER'])
    if 'WAITS' in os.environ:                                # e.g. WAITS=0:3,4:8; WAITS= (empty) or WAITS=none: no inter-batch waits; unset: the default
        w = os.environ['WAITS']
        kw['waits'] = [] if w in ('', 'none') else [tuple(float(v) for v in p.split(':')) for p in w.split(',')]
    pipe = BatchPipeline(model, depth=int(os.environ.get('DEPTH', '3')), **kw)
    with torch.no_grad():
        for _ in range(6):
            pipe.submit(x)
        pipe.drain()
        torch.cuda.synchronize()
        for K in (20, 20, 40):
            start = torch.cuda.Event(enable_timing=True)
            start.record()
            t0 = time.perf_counter()
            evs = []
            for _ in range(K):
                t = pipe.submit(x)
                e = torch.cuda.Event(enable_timing=True)
                e.record(t.stream)
                evs.append(e)
            pipe.drain()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            done = [start.elapsed_time(e) for e in evs]
            gaps = [done[0]] + [b - a for a, b in zip(done, done[1:])]
            mid = gaps[8:-4]
            print(f'K={K}: {dt / K:.3f} ms per batch ({dt:.1f} ms); steady-state gap {sum(mid) / len(mid):.3f} ms; first completion at {done[0]:.2f} ms; '
                  f'overhead vs K x steady {dt - K * sum(mid) / len(mid):.2f} ms')
            print('   completion gaps: ' + ' '.join(f'{g:.1f}' for g in gaps), flush=True)


if __name__ == '__main__':
    main()
