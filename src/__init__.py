"""Reference-compatible import paths (`src.models.aero.Aero`, `src.ddp.distrib`) so that checkpoints,
which pickle the generator's class path (reference model_serializer.py:22), and existing callers keep working.
The implementation lives in `aero_amd`."""
