"""CPU, world_size 2 over gloo: the batch-sharding contract that replaces src/ddp:
clip i -> rank i mod W, no data-path collective, results gathered back in clip order; weighted metric
all-reduce; max-over-ranks clock.  The per-clip work is the real Aero.forward through the emulated kernels."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank), AERO_EMU_THREADS='2')
    torch.set_num_threads(1)
    import json
    from aero_amd import _lib, distrib
    from aero_amd.engine import HipEngine
    from conftest import build_model
    from emu.build_emu import build
    distrib.init_from_env(backend='gloo')
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
    m = build_model(meta, 'tiny')
    object.__setattr__(m, '_engine', HipEngine(m, lib=_lib.load(build())))
    x = torch.randn(5, 1, 400, generator=torch.Generator().manual_seed(77))      # global batch, odd size
    mine = distrib.shard_batch(x)
    with torch.no_grad():
        y_local = m(mine)
    y = distrib.gather_batch(y_local, x.shape[0])
    avg = distrib.average([float(rank + 1)], count=mine.shape[0])
    tmax = distrib.max_over_ranks(float(rank))
    distrib.barrier()
    if rank == 0:
        with torch.no_grad():
            y_all = m(x)
        q.put((torch.equal(y, y_all), avg, tmax, distrib.shard_indices(5, 1, 2)))
    distrib.close()


def test_two_rank_batch_sharding():
    from emu.build_emu import build
    build()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, avg, tmax, idx = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert same                                     # sharded + gathered == single-process, bit exact
    assert idx == [1, 3]
    assert tmax == 1.0
    assert avg == pytest.approx([(1 * 3 + 2 * 2) / 5.0])   # rank0 has 3 clips (weight 3), rank1 has 2


def _opt_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from aero_amd import _lib, distrib
    from aero_amd.optim import FlatAdam
    from emu.build_emu import build
    distrib.init_from_env(backend='gloo')
    p = [torch.nn.Parameter(torch.arange(10, dtype=torch.float32)), torch.nn.Parameter(torch.ones(3, 5))]
    opt = FlatAdam(p, lr=1e-2, lib=_lib.load(build()))
    opt.zero_grad()
    for t in p:
        t.grad.fill_(float(rank + 1))                    # rank 0: 1, rank 1: 2 -> mean 1.5 on both
    scale = distrib.sum_gradients(opt.flat_g)            # ONE all-reduce over the flat buffer
    opt.step(grad_scale=scale)
    distrib.barrier()
    q.put((rank, scale, opt.flat_g[:4].tolist(), p[0].detach().numpy().copy()))    # by value: a tensor travels as an fd the exiting worker may close first
    distrib.close()


def test_two_rank_gradient_allreduce_and_fused_step():
    """train.py ddp=true (config 5): gradients summed over the ranks with one collective on FlatAdam's flat buffer,
    the 1 / world factor applied inside the fused optimizer step; both ranks end up with identical parameters, equal to a
    single-process step on the mean gradient."""
    from aero_amd import _lib
    from aero_amd.optim import FlatAdam
    from emu.build_emu import build
    lib = _lib.load(build())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_opt_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == 0.5 and got[0][2] == [3.0] * 4          # summed gradient 1 + 2
    assert (got[0][3] == got[1][3]).all()
    ref = [torch.nn.Parameter(torch.arange(10, dtype=torch.float32)), torch.nn.Parameter(torch.ones(3, 5))]
    o = FlatAdam(ref, lr=1e-2, lib=lib)
    for t in ref:
        t.grad.fill_(1.5)
    o.step()
    assert torch.equal(ref[0].detach(), torch.from_numpy(got[0][3]))
