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
    """One collective for a whole flat gradient buffer (FlatAdam.flat_g) for callers that do not wrap the model: all-reduce (SUM)
    over RCCL / xGMI; returns the factor that turns the sum into the mean (FlatAdam.step(grad_scale=...))."""
    if world_size == 1 or not is_initialized():
        return 1.0
    if not flat_grad.is_cuda and _dist().get_backend() == 'nccl':
        raise RuntimeError('sum_gradients: a CPU buffer cannot be reduced over the nccl (RCCL) backend -- keep the gradients on the device')
    _dist().all_reduce(flat_grad, op=_dist().ReduceOp.SUM)
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


class GradSync:
    """Gradient all-reduce of a wrapped generator (what DistributedDataParallel's reducer does, distrib.py:66-69): the HIP backward
    (aero_amd/train.py) hands over contiguous fp32 segments of its flat gradient buffer as soon as a stage's gradients are final --
    the decoder first, then the encoders from the deepest up -- and each goes out as ONE asynchronous all-reduce on RCCL's stream
    while the rest of the backward runs; the mean's 1 / world-size rides in the un-scaling pass that precedes the reduce."""

    def __init__(self):
        self._work = []
        self.launched = 0

    def active(self):
        return world_size > 1 and is_initialized()

    def mean_factor(self):
        return 1.0 / world_size

    def unscale(self, scale):
        """{S, 1/S} of this rank's backward -> the device factor 1 / (S * world_size)"""
        return scale[1:] * (1.0 / world_size)

    def reduce_async(self, seg):
        if world_size == 1 or not is_initialized():
            return
        self._work.append(_dist().all_reduce(seg, op=_dist().ReduceOp.SUM, async_op=True))
        self.launched += 1

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []


class DataParallel(torch.nn.Module):
    """What `wrap` returns for world_size > 1: the role of DistributedDataParallel around the generator (distrib.py:66-69) without its
    autograd hooks -- `forward` broadcasts the BatchNorm running buffers from rank 0 in training mode (DDP's broadcast_buffers) and runs
    the module; gradients are averaged over ranks by the module's `GradSync` inside the HIP backward.  `.module` as in DDP."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        object.__setattr__(module, '_grad_sync', GradSync())
        params = [p.detach() for p in module.parameters()]
        if is_initialized() and params:                          # DDP's constructor: every rank starts from rank 0's weights
            flat = torch.cat([p.reshape(-1) for p in params])
            _dist().broadcast(flat, 0)
            o = 0
            for p in params:
                p.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            if hasattr(module, 'repack'):
                module.repack()

    def _broadcast_buffers(self):
        bufs = [b for b in self.module.buffers() if b.is_floating_point()]
        if not bufs or not is_initialized():
            return
        flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
        _dist().broadcast(flat, 0)
        o = 0
        with torch.no_grad():
            for b in bufs:
                b.copy_(flat[o:o + b.numel()].view_as(b).to(b.dtype))
                o += b.numel()

    def forward(self, *args, **kwargs):
        if self.module.training and torch.is_grad_enabled():
            self._broadcast_buffers()
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__('module'), name)


def wrap(model):
    """distrib.py:59-69.  Inference needs no wrapper (weights are replicated, clips are independent); for training the generator and the
    critic (solver.py:51 wraps every model) are wrapped in `DataParallel`: their HIP backward passes average the gradients over the
    ranks themselves (`_grad_sync`: flat all-reduce segments overlapped with the generator's backward, one flat all-reduce after the
    critic's), BatchNorm buffers follow rank 0.  A module WITHOUT that protocol (anything that is not one of this package's models)
    gets torch's DistributedDataParallel, as in the reference: its gradients must not silently stay per-rank."""
    if world_size == 1:
        return model
    if getattr(model, '_supports_grad_sync', False):
        return DataParallel(model)
    from torch.nn.parallel import DistributedDataParallel
    if _dist().get_backend() == 'nccl':
        return DistributedDataParallel(model, device_ids=[torch.cuda.current_device()], output_device=torch.cuda.current_device())
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
