"""GPU (-m gpu): the multi-GPU path on hardware -- RCCL ("nccl") process group, clip i -> rank i mod W, forward,
gather back in clip order.  World size 1 always runs; world size 2 when two MI355X are visible."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from aero_amd import distrib
from conftest import build_model
distrib.init_from_env(backend='nccl')
dev = torch.device('cuda', int(os.environ['LOCAL_RANK']))
torch.cuda.set_device(dev)
n = distrib.count_ranks(dev)
meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
m = build_model(meta, 'small').to(dev)
x = torch.randn(5, 1, 2003, generator=torch.Generator().manual_seed(77))
with torch.no_grad():
    y_local = m(distrib.shard_batch(x).to(dev))
    y = distrib.gather_batch(y_local, x.shape[0])
    y_all = m(x.to(dev))
tmax = distrib.max_over_ranks(float(distrib.rank), dev)
distrib.barrier()
if distrib.rank == 0:
    print(json.dumps({'ranks': n, 'same': bool(torch.equal(y, y_all)), 'tmax': tmax, 'backend': torch.distributed.get_backend()}))
distrib.close()
'''


def _run(world, tmp_path):
    from aero_amd import launcher
    w = tmp_path / 'w.py'
    w.write_text(f'ROOT = {ROOT!r}\n' + WORKER)
    port = launcher.free_port()
    procs = [subprocess.Popen([sys.executable, str(w)], env=launcher.rank_env(r, world, port),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True) for r in range(world)]
    out, _ = procs[0].communicate(timeout=600)
    for p in procs:
        assert p.wait(timeout=600) == 0
    return json.loads([ln for ln in out.splitlines() if ln.startswith('{')][-1])


def test_rccl_world_size_1(tmp_path):
    """The nccl backend really initialises (WORLD_SIZE=1 under the launcher contract goes through the same code)."""
    w = tmp_path / 'w1.py'
    w.write_text(f'ROOT = {ROOT!r}\n' + WORKER.replace("distrib.init_from_env(backend='nccl')", '''
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
dist.init_process_group(backend='nccl', init_method='env://', world_size=1, rank=0)
t = torch.ones(1, device='cuda'); dist.all_reduce(t); assert float(t) == 1.0
'''))
    from aero_amd import launcher
    out = subprocess.run([sys.executable, str(w)], env=launcher.rank_env(0, 1, launcher.free_port()),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert r['same'] and r['backend'] == 'nccl'


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two visible MI355X')
def test_rccl_world_size_2_shard_forward_gather(tmp_path):
    r = _run(2, tmp_path)
    assert r == {'ranks': 2, 'same': True, 'tmax': 1.0, 'backend': 'nccl'}


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items()
                                                                           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert out.returncode != 0 and 'visible' in (out.stderr + out.stdout)


def test_comm_entry_points_world_size_1():
    """the RCCL-backed C entry points (include/aero_hip.h: aero_comm_*) with a one-rank communicator on the MI355X: an in-place sum
    over one rank leaves the buffer as it was, an all-gather of one rank is a copy"""
    import ctypes as C
    from aero_amd import _lib
    lib = _lib.load()
    uid = (C.c_char * 128)()
    lib.call('aero_comm_unique_id', C.cast(uid, C.c_void_p))
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    lib.call('aero_comm_init', 0, 1, C.cast(uid, C.c_void_p), C.byref(comm))
    assert comm.value
    st = torch.cuda.current_stream().cuda_stream
    x = torch.arange(1000, dtype=torch.float32, device='cuda') * 0.5
    ref = x.clone()
    lib.call('aero_allreduce_f32', comm, x.data_ptr(), x.numel(), st)
    out = torch.zeros(4000, dtype=torch.uint8, device='cuda')
    lib.call('aero_allgather', comm, ref.data_ptr(), out.data_ptr(), 4000, st)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    assert torch.equal(out.view(torch.float32), ref)
    lib.call('aero_comm_destroy', comm)
