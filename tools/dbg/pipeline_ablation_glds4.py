"""follow-up to pipeline_ablation.py: which of the eleven 4-wave LDS-tiled conv launches carry that family's contention penalty (1.16 ms
marginal for 0.62 ms of launches)?  Skips them one shape at a time in the pipelined loop."""
import ctypes as C
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
from conftest import GOLDEN, build_model
from aero_amd.pipeline import BatchPipeline

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
m = build_model(meta, 'full').cuda().eval()
eng = m._get_engine()
lib = eng.lib
x = torch.randn(64, 1, 8000, generator=torch.Generator().manual_seed(1)).cuda()
orig_call = lib.call
state = {'skip': None, 'seen': {}, 'n': 0}


def call(name, *args):
    if name == 'aero_conv_fwd':
        d = args[0]._obj
        buf = C.create_string_buffer(128)
        lib.cdll.aero_conv_kernel_name(args[0], buf, 128)
        kn = buf.value.decode()
        key = (kn, d.M, d.C0 + d.C1, d.ntaps, d.Fin, d.Fout)
        state['seen'][key] = state['seen'].get(key, 0) + 1
        if state['skip'] is not None and key == state['skip']:
            state['n'] += 1
            return
    orig_call(name, *args)


lib.call = call


def timed(depth=3):
    pipe = BatchPipeline(m, depth=depth)
    with torch.no_grad():
        for _ in range(depth + 2):
            pipe.submit(x)
        pipe.drain()
        torch.cuda.synchronize()
        state['n'] = 0
        t0 = time.perf_counter()
        for _ in range(K):
            pipe.submit(x)
        pipe.drain()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3, state['n'] // K


base, _ = timed()
base2, _ = timed()
print(f'nothing skipped: {base:.3f} / {base2:.3f} ms per batch')
keys = [k for k in state['seen'] if 'glds_kernel' in k[0] or 'glds8' in k[0]]
for k in keys:
    state['skip'] = k
    ms, n = timed()
    print(f'{k[0][:42]:42s} M={k[1]:4d} C={k[2]:4d} taps={k[3]} F={k[4]}->{k[5]}: {ms:7.3f} ms ({base2 - ms:+.3f}; {n} launches per batch)', flush=True)
state['skip'] = None
print(f'nothing skipped (again): {timed()[0]:.3f}')
