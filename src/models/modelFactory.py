"""get_model(args) as the reference's factory (modelFactory.py:6-29): the AERO generator and, for `adversarial: true` experiments, the
MelGAN multi-scale critic (`msd_melgan`, the critic of every aero experiment file).  Seanet and the HiFi-GAN critics are in no aero
config and raise NotImplementedError."""
from aero_amd.trainer import build_models as get_model  # noqa: F401
