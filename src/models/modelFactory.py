"""get_model(args) as the reference's factory (modelFactory.py:6-12) for the generator.  The GAN critics and the
Seanet baseline are training-only / off the hot path and are not part of this package."""
from aero_amd.modules import Aero


def get_model(args):
    exp = args.experiment
    if exp.model != 'aero':
        raise NotImplementedError(f"model '{exp.model}': only the AERO generator is implemented on MI355X")
    return {'generator': Aero(**exp.aero)}
