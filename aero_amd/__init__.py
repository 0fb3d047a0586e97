"""aero_amd -- MI355X-native (gfx950) implementation of AERO's spectral forward/inverse path.

Host side: `aero_amd.modules.Aero` (drop-in for the reference's `src.models.aero.Aero`).
Device side: hand-written HIP kernels behind the C-ABI of `include/aero_hip.h`
(`aero_amd/csrc`, built by `__graft_entry__.build()` into `aero_amd/libaero_hip.so`).
"""
from .modules import Aero  # noqa: F401

__all__ = ['Aero']
