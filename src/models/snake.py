from aero_amd.modules import Snake  # noqa: F401
