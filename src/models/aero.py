"""`src.models.aero` import path of the reference, served by the MI355X-native implementation."""
from aero_amd.modules import Aero, HDecLayer, HEncLayer, rescale_module  # noqa: F401
