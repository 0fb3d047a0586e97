"""`python train.py experiment=aero_11-44_512_256 ddp=true [steps=N]` -- the training entry point of the reference (train.py:20-125) for
the part of it that is on the MI355X path: BASELINE config 5, one forward + backward + optimiser step per batch, one process per GPU.

What it keeps of the reference's flow: the hydra-style command line over `conf/` (`aero_amd.config`), `ddp=true` re-executing this command
once per visible GPU (`start_ddp_workers`, executor.py:50-75), `distrib.init`, the seed, `modelFactory.get_model`, the global batch divided
by the world size (train.py:50-51), Adam for the generator and for the critic (train.py:83-92), every model through `distrib.wrap`
(solver.py:51) and the per-batch step of `Solver._run_one_epoch` (aero_amd/trainer.py).  What it leaves out: the Solver's epochs,
checkpoints, wandb and dataset readers (host code outside the path) -- batches are synthetic white noise of the
experiment's geometry, `steps` of them (default 3).

Rendezvous: workers started by `ddp=true` (or by `python -m torch.distributed.run`) find RANK / WORLD_SIZE / MASTER_* in the environment
(`distrib.init_from_env`: "nccl" = RCCL over xGMI); the reference's own `rank=R world_size=W` + file:// rendezvous is honoured too.
"""
import json
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
logger = logging.getLogger('train')


def run(args):
    import torch

    from aero_amd import distrib, launcher, trainer
    if launcher.under_launcher():
        distrib.init_from_env(backend=args.ddp_backend if int(os.environ.get('WORLD_SIZE', '1')) > 1 else None)
    else:
        distrib.init(args)
    dev = torch.device(args.device)
    if dev.type == 'cuda' and not torch.cuda.is_available():
        raise RuntimeError('train.py: no MI355X visible (the training step has no CPU path)')
    if dev.type == 'cuda':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', distrib.rank)))
    torch.manual_seed(args.seed)
    models = trainer.build_models(args)
    exp = args.experiment
    assert exp.batch_size % distrib.world_size == 0, 'the global batch must divide by the number of ranks (train.py:50)'
    per_rank = exp.batch_size // distrib.world_size
    for m in models.values():
        m.to(dev).train()
    optimizers = trainer.build_optimizers(models, args)
    step = trainer.TrainStep(models, optimizers, args)
    steps = int(args.get('steps', 3))
    hist = []
    for i in range(steps):
        # every rank draws its own clips of the global batch (what DistributedSampler's disjoint shards amount to)
        lr, hr = trainer.synthetic_batch(args, per_rank, dev, seed=1000 * i + distrib.rank)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        distrib.barrier()
        t0 = time.time()
        rec = step(lr, hr)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        dt = distrib.max_over_ranks(time.time() - t0)
        vals = {k: float(v) for k, v in rec.items()}
        keys = sorted(vals)
        avg = distrib.average([vals[k] for k in keys], per_rank)                 # solver.py's logged losses: averaged over ranks
        vals = dict(zip(keys, avg))
        if any(v != v for v in vals.values()):
            raise RuntimeError(f'step {i}: non-finite loss {vals}')
        hist.append(vals)
        if distrib.rank == 0:
            print(json.dumps({'step': i, 'ms': round(1e3 * dt, 2), 'world_size': distrib.world_size, 'batch_per_rank': per_rank,
                              **{k: round(v, 6) for k, v in vals.items()}}), flush=True)
    # DDP's invariant, checked: every rank holds the same weights after the last step
    if distrib.world_size > 1:
        for name, opt in optimizers.items():
            s = torch.stack([opt.flat_p.double().sum(), opt.flat_p.double().square().sum()])
            lo, hi = s.clone(), s.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                raise RuntimeError(f'{name}: the ranks hold different weights after {steps} steps')
        if distrib.rank == 0:
            print(json.dumps({'ranks_in_sync': True, 'world_size': distrib.world_size}), flush=True)
    distrib.close()
    return hist


def main(argv=None):
    from aero_amd import launcher
    from aero_amd.config import load_config
    argv = list(sys.argv[1:] if argv is None else argv)
    args = load_config(os.path.join(ROOT, 'conf'), argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO)
    if args.ddp and args.rank is None and not launcher.under_launcher():
        launcher.start_ddp_workers(args, argv=[os.path.abspath(__file__)] + argv)     # exits with the workers' status
    run(args)


if __name__ == '__main__':
    main()
