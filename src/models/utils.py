from aero_amd.modules import capture_init  # noqa: F401
