"""What does each kernel family cost IN THE PRODUCT SCHEDULE (three batches in flight, aero_amd/pipeline.py)?  The per-launch table sums
single-stream durations; with batches overlapping, a latency-bound launch that hides under other batches' work costs the batch less than
its duration, an MFMA-bound one costs all of it.  This tool times the pipelined loop with one family's launches SKIPPED at the C-ABI
boundary (outputs are then garbage: timing only) -- the difference to the full loop is that family's marginal cost per batch.
usage: pipeline_ablation.py [batches]"""
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
skip = {'fns': set(), 'conv': None, 'n': 0}


def call(name, *args):
    if name in skip['fns']:
        skip['n'] += 1
        return
    if name == 'aero_conv_fwd' and skip['conv']:
        buf = C.create_string_buffer(128)
        lib.cdll.aero_conv_kernel_name(args[0], buf, 128)
        if skip['conv'] in buf.value.decode():
            skip['n'] += 1
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
        skip['n'] = 0
        t0 = time.perf_counter()
        for _ in range(K):
            t = pipe.submit(x)
        host = time.perf_counter() - t0
        pipe.drain()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3, host / K * 1e3, skip['n'] // K


cases = [('nothing skipped', set(), None), ('LSTM', {'aero_lstm_fwd'}, None), ('LocalState core', {'aero_localstate_fwd'}, None),
         ('LSTM + LocalState', {'aero_lstm_fwd', 'aero_localstate_fwd'}, None), ('GroupNorm apply / stats', {'aero_norm_apply', 'aero_norm_stats'}, None),
         ('pointwise kernel (k_pw.h)', {'aero_pw_fwd'}, None), ('ring conv 256-row <2, 4, 4, 3', set(), 'ring_kernel<2, 4, 4, 3'),
         ('ring conv 192-row <2, 2|4, 3, 3', set(), ', 3, 3, 0>'), ('8-wave LDS-tiled convs (glds8)', set(), 'glds8'),
         ('4-wave LDS-tiled convs (glds_kernel)', set(), 'glds_kernel'), ('enc0 + dconv rows', {'aero_enc0_fwd', 'aero_dconv_row_fwd'}, None),
         ('freq_fc + squeeze + gram', {'aero_freqfc_fwd', 'aero_squeeze_fwd', 'aero_gram_stats'}, None),
         ('STFT + normalise + iSTFT + tail finish', {'aero_stft_dft_fwd', 'aero_spec_normalize', 'aero_istft_fwd', 'aero_convtr_tail_finish'}, None),
         ('nothing skipped (again)', set(), None)]
base = None
for name, fns, conv in cases:
    skip['fns'], skip['conv'] = fns, conv
    ms, host, n = timed()
    if base is None:
        base = ms
    print(f'{name:44s} {ms:7.3f} ms per batch  ({base - ms:+6.3f} vs full; {n:3d} launches skipped per batch; host enqueue {host:.2f} ms)', flush=True)
skip['fns'], skip['conv'] = set(), None
for d in (1, 2, 4):
    ms, host, _ = timed(d)
    print(f'depth {d}: {ms:7.3f} ms per batch (host {host:.2f})', flush=True)
