"""One training step of BASELINE config 5 as the reference's entry point runs it (`train.py ddp=true`):

    build_models      <- src/models/modelFactory.py:6-29   generator + the `msd_melgan` critic of the experiment file
    build_optimizers  <- train.py:83-92                    Adam(generator), Adam(critics), lr / betas from the config
    TrainStep         <- src/solver.py:51 (every model through distrib.wrap), :296-320 (forward, losses, optimise),
                         :428-470 (which losses), :475-520 (MelGAN hinge / feature matching), :602-611 (the two optimiser steps)

The Solver around it (epochs, checkpoints, logging, evaluation) is host code outside the path and is not rebuilt; this module is the part
of it that touches the device.  Everything runs on the HIP kernels: the generator through `AeroFunction`, the criterion through
`losses.MultiResolutionSTFTLoss`, the critic through `Discriminator.generator_losses / discriminator_loss`, both optimisers as
`FlatAdam`.  With world_size > 1 the gradients of BOTH models are averaged over the ranks inside their backward passes.
"""
import torch

from . import distrib, losses
from .optim import FlatAdam


def build_models(args):
    """modelFactory.py:6-29 for what aero's experiment files use: `model: aero` and, with `adversarial: true`, the MelGAN multi-scale
    critic.  (Seanet and the HiFi-GAN critics appear in no aero config: NotImplementedError, as `src.models.modelFactory`.)"""
    from .modules import Aero
    exp = args.experiment
    if exp.model != 'aero':
        raise NotImplementedError(f"model '{exp.model}': only the AERO generator is implemented on MI355X")
    models = {'generator': Aero(**dict(exp.aero))}
    if exp.get('adversarial'):
        from .discriminators import Discriminator
        for name in exp.discriminator_models:
            if name != 'msd_melgan':
                raise NotImplementedError(f"critic '{name}': only msd_melgan (the critic of every aero experiment file) is implemented")
            models[name] = Discriminator(**dict(exp.melgan_discriminator))
    return models


def build_optimizers(models, args, lib=None):
    """train.py:83-92: Adam(lr, betas=(0.9, beta2)) for the generator and one Adam over the chained critics' parameters."""
    if args.optim != 'adam':
        raise ValueError('Invalid optimizer %s' % args.optim)
    gen = models['generator']
    opts = {'optimizer': FlatAdam(gen.parameters(), lr=args.lr, betas=(0.9, args.beta2), lib=lib, model=gen)}
    critics = [m for k, m in models.items() if k != 'generator']
    if critics:
        if len(critics) != 1:
            raise NotImplementedError('one critic (msd_melgan) per experiment')
        opts['disc_optimizer'] = FlatAdam(critics[0].parameters(), lr=args.lr, betas=(0.9, args.beta2), lib=lib, model=critics[0])
    return opts


class TrainStep:
    def __init__(self, models, optimizers, args):
        self.args = args
        exp = args.experiment
        self.adversarial = bool(exp.get('adversarial'))
        self.models = models
        self.dmodels = {k: distrib.wrap(m) for k, m in models.items()}            # solver.py:51
        self.dmodel = self.dmodels['generator']
        self.optimizer = optimizers['optimizer']
        self.disc_optimizer = optimizers.get('disc_optimizer')
        if self.adversarial and self.disc_optimizer is None:
            raise ValueError('adversarial experiment without a disc_optimizer')
        self.mrstft = None
        if 'stft' in args.losses:
            self.mrstft = losses.MultiResolutionSTFTLoss(factor_sc=args.stft_sc_factor, factor_mag=args.stft_mag_factor)

    def losses_of(self, pr, hr):
        """solver.py:428-470 -> {'generator': {...}, 'discriminator': {...}}"""
        import torch.nn.functional as F
        out = {'generator': {}, 'discriminator': {}}
        if 'l1' in self.args.losses:
            out['generator']['l1'] = F.l1_loss(pr, hr)
        if 'l2' in self.args.losses:
            out['generator']['l2'] = F.mse_loss(pr, hr)
        if self.mrstft is not None:
            sc, mag = self.mrstft(pr.squeeze(1), hr.squeeze(1))
            out['generator']['stft'] = sc + mag
        if self.adversarial:
            exp = self.args.experiment
            critic = self.dmodels['msd_melgan']
            md = exp.melgan_discriminator
            if md.num_D != critic.num_D:
                raise ValueError('melgan_discriminator.num_D does not match the critic')
            adv, feat = critic.generator_losses(pr, hr, n_layers=md.n_layers, features_loss_lambda=exp.features_loss_lambda)
            if not exp.get('only_features_loss'):
                out['generator']['adversarial_melgan'] = adv
            if not exp.get('only_adversarial_loss'):
                out['generator']['features_melgan'] = feat
            # D(fake.detach()), D(real) on the weights the generator's losses just used (solver.py:478-480): the critic keeps that
            # record, so this costs no second forward
            out['discriminator']['msd_melgan'] = critic.discriminator_loss(pr.detach(), hr)
        return out

    def __call__(self, lr, hr):
        """one batch in training mode; returns {'generator_<name>': value, 'discriminator_<name>': value, 'total': value} (device scalars)"""
        pr = self.dmodel(lr)
        ls = self.losses_of(pr, hr)
        total = sum(ls['generator'].values())
        self.optimizer.zero_grad()                                                # solver.py:602-605
        total.backward()
        self.optimizer.step()
        if self.adversarial:                                                      # solver.py:607-611
            d_total = sum(ls['discriminator'].values())
            self.disc_optimizer.zero_grad()
            d_total.backward()
            self.disc_optimizer.step()
        rec = {'total': total.detach()}
        rec.update({'generator_' + k: v.detach() for k, v in ls['generator'].items()})
        rec.update({'discriminator_' + k: v.detach() for k, v in ls['discriminator'].items()})
        return rec


def synthetic_batch(args, batch, device, seed=0):
    """white-noise (lr, hr) pair of the experiment's geometry: `segment` seconds at lr_sr / hr_sr (BASELINE.json: synthetic data; the
    reference's LrHrSet file reader is host code outside the path)"""
    exp = args.experiment
    g = torch.Generator().manual_seed(seed)
    n_lr, n_hr = int(exp.segment * exp.lr_sr), int(exp.segment * exp.hr_sr)
    lr = torch.randn(batch, 1, n_lr, generator=g)
    hr = 0.1 * torch.randn(batch, 1, n_hr, generator=g)
    return lr.to(device), hr.to(device)
