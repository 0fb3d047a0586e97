"""`src.ddp.executor` import path of the reference (executor.py:13-75), served by aero_amd.launcher."""
from aero_amd.launcher import Children as ChildrenManager  # noqa: F401
from aero_amd.launcher import start_ddp_workers  # noqa: F401
