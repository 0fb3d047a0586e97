"""first adversarial step on the HIP path against a dump of the reference's (tools/dbg/gan_ref_step1.pt, built in the container): critic
gradients of the bias / weight-norm gain parameters, logits, and what one Adam step did to them"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import GOLDEN, seeded
from aero_amd import trainer
from aero_amd.config import _wrap
meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
cfgt = meta['train_gan_trajectory']
ref = torch.load(os.path.join(ROOT, 'tools', 'dbg', 'gan_ref_step1.pt'))
args = _wrap(dict(optim='adam', lr=cfgt['lr'], beta2=cfgt['betas'][1], losses=['stft'], stft_sc_factor=0.5, stft_mag_factor=0.5,
                  experiment=dict(model='aero', aero=cfgt['gen_cfg'], adversarial=True, features_loss_lambda=cfgt['features_loss_lambda'],
                                  only_features_loss=False, only_adversarial_loss=False, discriminator_models=['msd_melgan'],
                                  melgan_discriminator=cfgt['disc_cfg'])))
torch.manual_seed(cfgt['seed'])
models = {k: m.cuda().train() for k, m in trainer.build_models(args).items()}
opts = trainer.build_optimizers(models, args)
step = trainer.TrainStep(models, opts, args)
x = seeded((2, 1, cfgt['L']), cfgt['x_seed']).cuda()
hr = (cfgt['hr_scale'] * seeded((2, 1, 4 * cfgt['L']), cfgt['hr_seed'])).cuda()
disc = models['msd_melgan']
before = {n: p.detach().clone() for n, p in disc.named_parameters()}
rec = step(x, hr)
torch.cuda.synchronize()
print({k: float(v) for k, v in rec.items()})
for n, p in disc.named_parameters():
    if n in ref['disc_grad']:
        g, gr = p.grad.detach().cpu().float(), ref['disc_grad'][n]
        d, dr = (p.detach() - before[n]).cpu(), ref['disc_after1'][n] - before[n].cpu()
        print(f'{n:40s} |g| {float(g.norm()):.3e} ref {float(gr.norm()):.3e} rel {float((g - gr).norm() / gr.norm().clamp_min(1e-30)):.2e} | '
              f'update max {float(d.abs().max()):.2e} ref {float(dr.abs().max()):.2e}  diff {float((d - dr).abs().max()):.2e}')
