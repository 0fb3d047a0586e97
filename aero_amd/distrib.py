"""One-process-per-GPU batch sharding over RCCL/xGMI (replaces the reference's src/ddp/distrib.py).

Same public surface as the reference module (`init`, `close`, `average`, `wrap`, `barrier`, `loader`,
module attributes `rank` / `world_size`; distrib.py:16-101) so callers need no change.  The forward
path shards by clip with NO data-path collective: clip i belongs to rank i mod world_size
(distrib.py:100).  Collectives that remain: a weighted metric all-reduce (distrib.py:43-55), barriers,
and `max_over_ranks` for the benchmark clock.  `torch.distributed` backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests.
"""
import logging
import os

import torch

logger = logging.getLogger(__name__)
rank = 0
world_size = 1


def _dist():
    import torch.distributed as dist
    return dist


def is_initialized():
    return world_size > 1 and _dist().is_available() and _dist().is_initialized()


def init(args):
    """Reference-style init from a hydra config (distrib.py:16-34): file:// rendezvous."""
    global rank, world_size
    if args.ddp:
        assert args.rank is not None and args.world_size is not None
        rank, world_size = args.rank, args.world_size
    if world_size == 1:
        return
    backend = args.ddp_backend
    if backend == 'nccl':
        torch.cuda.set_device(rank)
    _dist().init_process_group(backend=backend, init_method='file://' + os.path.abspath(args.rendezvous_file),
                               world_size=world_size, rank=rank)
    logger.debug('Distributed rendezvous went well, rank %d/%d', rank, world_size)


def init_from_env(backend=None):
    """torchrun-style init (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK)."""
    global rank, world_size
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world_size == 1:
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)))
    _dist().init_process_group(backend=backend, init_method='env://', world_size=world_size, rank=rank)


def close():
    global rank, world_size
    if world_size == 1:
        return
    if _dist().is_initialized():
        _dist().destroy_process_group()
    rank, world_size = 0, 1


def barrier():
    if world_size > 1:
        _dist().barrier()


def _device_for_collective():
    if _dist().get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def average(metrics, count=1.):
    """Weighted average of a 1-D float vector over ranks (distrib.py:43-55)."""
    if world_size == 1:
        return metrics
    t = torch.tensor(list(metrics) + [1], device=_device_for_collective(), dtype=torch.float32)
    t *= count
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    return (t[:-1] / t[-1]).cpu().numpy().tolist()


def max_over_ranks(value, device=None):
    if world_size == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_device_for_collective())
    _dist().all_reduce(t, op=_dist().ReduceOp.MAX)
    return float(t.item())


def count_ranks(device=None):
    """Number of ranks that take part in an all-reduce of ones (1 without a process group): proof that the
    collective backend (RCCL over xGMI with "nccl") is up on every rank, not just that WORLD_SIZE was set."""
    if world_size == 1:
        return 1
    t = torch.ones(1, dtype=torch.float32, device=_device_for_collective())
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    return int(round(float(t.item())))


def sum_gradients(flat_grad):
    """One collective for the whole generator: all-reduce (SUM) of FlatAdam's flat gradient buffer over RCCL / xGMI (what
    DistributedDataParallel does bucket by bucket, distrib.py:66).  Returns the factor the optimizer applies to turn the sum into
    the mean (FlatAdam.step(grad_scale=...)): the division rides along in the fused step instead of a separate pass."""
    if world_size == 1 or not is_initialized():
        return 1.0
    t = flat_grad if flat_grad.is_cuda or _dist().get_backend() != 'nccl' else flat_grad.cuda()
    _dist().all_reduce(t, op=_dist().ReduceOp.SUM)
    return 1.0 / world_size


def shard_indices(n, r=None, w=None):
    """Eval sharding rule of the reference: item i -> rank i mod world (distrib.py:100)."""
    r = rank if r is None else r
    w = world_size if w is None else w
    return list(range(r, n, w))


def shard_batch(x, r=None, w=None):
    """The clips of a global batch [N, ...] owned by this rank (clip i -> rank i mod W)."""
    r = rank if r is None else r
    w = world_size if w is None else w
    return x[r::w]


def gather_batch(y_local, n_total):
    """Inverse of shard_batch: all-gather the per-rank results back into global clip order."""
    if world_size == 1:
        return y_local
    dist = _dist()
    per = (n_total + world_size - 1) // world_size
    pad = per - y_local.shape[0]
    if pad:
        y_local = torch.cat([y_local, y_local.new_zeros((pad,) + tuple(y_local.shape[1:]))], 0)
    parts = [torch.empty_like(y_local) for _ in range(world_size)]
    dist.all_gather(parts, y_local.contiguous())
    out = y_local.new_empty((n_total,) + tuple(y_local.shape[1:]))
    for r in range(world_size):
        idx = shard_indices(n_total, r, world_size)
        out[idx] = parts[r][:len(idx)]
    return out


def wrap(model):
    """Inference needs no wrapper (weights are replicated, clips are independent).  Training-time
    gradient all-reduce over RCCL is listed as follow-up work in DESIGN.md."""
    if world_size == 1:
        return model
    from torch.nn.parallel.distributed import DistributedDataParallel
    if next(model.parameters()).is_cuda:
        return DistributedDataParallel(model, device_ids=[torch.cuda.current_device()],
                                       output_device=torch.cuda.current_device())
    return DistributedDataParallel(model)


def loader(dataset, *args, shuffle=False, klass=None, **kwargs):
    """distrib.py:77-101: DistributedSampler for training, strided Subset shard for evaluation."""
    from torch.utils.data import DataLoader, Subset
    from torch.utils.data.distributed import DistributedSampler
    klass = klass or DataLoader
    if world_size == 1:
        return klass(dataset, *args, shuffle=shuffle, **kwargs)
    if shuffle:
        return klass(dataset, *args, **kwargs, sampler=DistributedSampler(dataset, num_replicas=world_size, rank=rank))
    return klass(Subset(dataset, shard_indices(len(dataset))), *args, shuffle=shuffle, **kwargs)
