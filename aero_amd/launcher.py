"""One process per GPU on one node, supervised (replaces the launcher half of the reference's src/ddp:
executor.py:13-75 `ChildrenManager` / `start_ddp_workers`).

The reference re-executes `sys.argv` once per visible GPU with `world_size=` / `rank=` appended and a file://
rendezvous.  Here the children get the torchrun environment contract instead (RANK, LOCAL_RANK, WORLD_SIZE,
MASTER_ADDR=127.0.0.1, MASTER_PORT) so that `aero_amd.distrib.init_from_env()` brings up RCCL ("nccl" on ROCm) over
xGMI, and the same script also runs unchanged under `python -m torch.distributed.run`.  Supervision is the same
contract as the reference's: if any worker dies with a non-zero status every other worker is terminated and the
launcher reports failure; Ctrl-C terminates all workers.
"""
import logging
import os
import socket
import subprocess as sp
import sys
import time

logger = logging.getLogger(__name__)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def under_launcher():
    """True inside a worker started by `spawn_ranks` or by torchrun (the rendezvous environment is already set)."""
    return 'WORLD_SIZE' in os.environ and 'RANK' in os.environ


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this driver (RCCL needs it)
    return env


class Children:
    """Supervises worker processes: first failure (or KeyboardInterrupt) terminates the rest (executor.py:13-47)."""

    def __init__(self, poll_s=0.1, grace_s=10.0):
        self.procs, self.failed, self.poll_s, self.grace_s = [], False, poll_s, grace_s

    def add(self, proc):
        proc.rank = len(self.procs)
        self.procs.append(proc)

    def wait(self, timeout_s=None):
        alive = list(self.procs)
        t_end = None if timeout_s is None else time.monotonic() + timeout_s
        try:
            while alive and not self.failed:
                for p in list(alive):
                    try:
                        code = p.wait(self.poll_s)
                    except sp.TimeoutExpired:
                        continue
                    alive.remove(p)
                    if code:
                        logger.error('worker %d exited with status %d: stopping the other workers', p.rank, code)
                        self.failed = True
                if t_end is not None and time.monotonic() > t_end and alive:
                    logger.error('workers still running after %.0f s: stopping them', timeout_s)
                    self.failed = True
        except KeyboardInterrupt:
            logger.error('interrupted: stopping all workers')
            self.failed = True
        for p in alive:                                       # exact PIDs we started, never a pattern
            p.terminate()
        t_kill = time.monotonic() + self.grace_s
        for p in alive:
            try:
                p.wait(max(0.0, t_kill - time.monotonic()))
            except sp.TimeoutExpired:
                p.kill()
        return not self.failed


def spawn_ranks(argv, nproc, port=None, quiet_nonzero_ranks=False, timeout_s=None, env=None):
    """Run `python argv...` once per rank (rank r on GPU r).  Returns True when every worker exited 0.
    Rank 0 inherits stdio; the other ranks too unless `quiet_nonzero_ranks` (executor.py:66-69 silences them)."""
    port = port or free_port()
    kids = Children()
    for r in range(nproc):
        kw = {}
        if r > 0 and quiet_nonzero_ranks:
            kw = dict(stdin=sp.DEVNULL, stdout=sp.DEVNULL, stderr=sp.DEVNULL)
        kids.add(sp.Popen([sys.executable] + list(argv), env=rank_env(r, nproc, port, env), **kw))
    return kids.wait(timeout_s)


def start_ddp_workers(args=None, argv=None):
    """Reference entry point name (executor.py:50): one worker per visible GPU re-running this command line."""
    world = visible_gpus()
    if not world:
        logger.error('DDP is only available on GPU: no MI355X visible')
        sys.exit(1)
    logger.info('Starting %d worker processes (one per GPU, RCCL over xGMI).', world)
    ok = spawn_ranks(list(sys.argv if argv is None else argv), world, quiet_nonzero_ranks=True)
    sys.exit(0 if ok else 1)
