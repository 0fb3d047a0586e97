"""Round 6, VERDICT r5 item 1: does SCHEDULING recover the part of the pipelined step that is neither MFMA work nor hidden?
The serving loop (aero_amd/pipeline.py) with the stages of every batch in flight issued on streams of different dispatch priority / CU
masks; every variant is timed over K whole batches between two device synchronisations, interleaved with the plain schedule (A/B on the
same box, same process), and its outputs are compared bit for bit with the plain schedule's.

    python tools/dbg/sched_sweep.py [K] [name=schedule ...]        (schedule: the text form of aero_amd.pipeline.parse_schedule)
"""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', os.environ.get('QUEUES', '8'))
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import FULL_CFG  # noqa: E402
from aero_amd import Aero  # noqa: E402
from aero_amd.pipeline import BatchPipeline  # noqa: E402

DEFAULT = [
    ('plain', 3, ''),
    ('lat-hi', 3, 'stages=mmllmmmm;l=prio:-1'),
    ('lat-fork', 3, 'stages=mmllmmmm;l=plain'),          # control: the same fork / join on a default-priority stream
    ('lstm-hi', 3, 'lstm=l;l=prio:-1'),
    ('lstm-mask128', 3, 'lstm=l;l=mask:0:128'),
    ('lat-hi-shared', 3, 'stages=mmllmmmm;l=prio:-1:shared'),
    ('enc-hi', 3, 'stages=eeeemmmm;e=prio:-1'),
    ('dec-lo', 3, 'stages=mmmmdddd;d=prio:1'),
    ('lat-hi+dec-lo', 3, 'stages=mmlldddd;l=prio:-1;d=prio:1'),
    ('enc-hi+dec-lo', 3, 'stages=eeeedddd;e=prio:-1;d=prio:1'),
    ('lat-mask32', 3, 'stages=mmllmmmm;l=mask:0:32'),
    ('lat-mask64', 3, 'stages=mmllmmmm;l=mask:0:64'),
    ('lat-mask64+dec-mask192', 3, 'stages=mmlldddd;l=mask:0:64;d=mask:64:256'),
    ('lat-mask32+dec-mask224', 3, 'stages=mmlldddd;l=mask:0:32;d=mask:32:256'),
    ('lat-mask96+dec-mask160', 3, 'stages=mmlldddd;l=mask:0:96;d=mask:96:256'),
    ('dec-mask224', 3, 'stages=mmmmdddd;d=mask:32:256'),
    ('lat-hi d4', 4, 'stages=mmllmmmm;l=prio:-1'),
    ('plain d4', 4, ''),
]


def timed(pipe, x, K):
    for _ in range(pipe.depth):
        pipe.submit(x)
    pipe.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        t = pipe.submit(x)
    host = time.perf_counter() - t0
    pipe.drain()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3, host / K * 1e3, pipe.result(t)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
    variants = [a.split('=', 1) for a in sys.argv[1:] if '=' in a and not a.split('=', 1)[0].isdigit()]
    variants = [(n, int(os.environ.get('DEPTH', '3')), s) for n, s in variants] or DEFAULT
    if os.environ.get('ONLY'):                               # ONE variant per process next to the plain schedule: stream objects that merely
        keep = os.environ['ONLY'].split(',')                 # exist shift which streams share a hardware queue
        variants = [v for v in DEFAULT if v[0] in keep or v[0] == 'plain']
    torch.manual_seed(2036)
    dev = torch.device('cuda', 0)
    model = Aero(**FULL_CFG).eval().to(dev)
    x = torch.randn(64, 1, 8000, generator=torch.Generator().manual_seed(1000)).to(dev)
    with torch.no_grad():
        ref = model(x)
    torch.cuda.synchronize()
    if os.environ.get('TRACE'):                              # one schedule, no A/B: the process rocprofv3 --kernel-trace looks at
        name, depth, sched = [v for v in variants if v[0] == os.environ['TRACE']][0]
        pipe = BatchPipeline(model, depth=depth, schedule=sched or None)
        timed(pipe, x, 6)
        ms, host, y = timed(pipe, x, K)
        print(f'{name}: {ms:.3f} ms per batch under the tracer, bit-equal to model(x): {torch.equal(y, ref)}', flush=True)
        return
    pipes = {}
    for name, depth, sched in variants:
        try:
            pipes[name] = BatchPipeline(model, depth=depth, schedule=sched or None)
            timed(pipes[name], x, 6)
        except Exception as e:
            print(f'{name:28s} FAILED to set up: {e!r}'[:300], flush=True)
    print(f'GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES")}  K={K}  (ms per batch of 64 clips; three rounds, variants interleaved)', flush=True)
    res = {n: [] for n in pipes}
    for rnd in range(3):
        for name, pipe in pipes.items():
            ms, host, y = timed(pipe, x, K)
            same = torch.equal(y, ref)
            res[name].append((ms, host, same))
    base = min(r[0] for r in res.get('plain', [(float('nan'),)]))
    for name, rs in res.items():
        ms = [r[0] for r in rs]
        print(f'{name:28s} min {min(ms):7.3f}  runs {" ".join(f"{m:7.3f}" for m in ms)}   host {rs[-1][1]:5.2f}   vs plain {min(ms) - base:+6.3f}   '
              f'bit-equal to model(x): {all(r[2] for r in rs)}', flush=True)


if __name__ == '__main__':
    main()
