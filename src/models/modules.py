"""`src.models.modules` import path of the reference (parameter containers; arithmetic is in aero_amd/csrc)."""
from aero_amd.modules import BLSTM, DConv, FTB, LayerScale, LocalState, ScaledEmbedding  # noqa: F401
