"""`src.ddp.distrib` import path of the reference, served by aero_amd.distrib (RCCL over xGMI)."""
from aero_amd.distrib import *  # noqa: F401,F403
from aero_amd.distrib import average, barrier, close, init, loader, wrap  # noqa: F401
