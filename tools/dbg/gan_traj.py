"""print the HIP adversarial training trajectory next to the reference golden (tests/golden/train_gan_trajectory.npz), term by term"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import GOLDEN, seeded
from aero_amd import trainer
from aero_amd.config import _wrap
meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
cfgt = meta['train_gan_trajectory']
gold = torch.from_numpy(np.load(os.path.join(GOLDEN, 'train_gan_trajectory.npz'))['loss'])
lam = float(os.environ.get('LAMBDA', cfgt['features_loss_lambda']))
args = _wrap(dict(optim='adam', lr=cfgt['lr'], beta2=cfgt['betas'][1], losses=['stft'], stft_sc_factor=0.5, stft_mag_factor=0.5,
                  experiment=dict(model='aero', aero=cfgt['gen_cfg'], adversarial=True, features_loss_lambda=lam,
                                  only_features_loss=False, only_adversarial_loss=False, discriminator_models=['msd_melgan'],
                                  melgan_discriminator=cfgt['disc_cfg'])))
torch.manual_seed(cfgt['seed'])
models = {k: m.cuda().train() for k, m in trainer.build_models(args).items()}
opts = trainer.build_optimizers(models, args)
step = trainer.TrainStep(models, opts, args)
x = seeded((2, 1, cfgt['L']), cfgt['x_seed']).cuda()
hr = (cfgt['hr_scale'] * seeded((2, 1, 4 * cfgt['L']), cfgt['hr_seed'])).cuda()
freeze = os.environ.get('FREEZE', '')
for i in range(cfgt['steps']):
    if freeze == 'disc':
        p0 = opts['disc_optimizer'].flat_p.clone()
    rec = step(x, hr)
    if freeze == 'disc':
        opts['disc_optimizer'].flat_p.copy_(p0); models['msd_melgan'].repack() if hasattr(models['msd_melgan'], 'repack') else None
    g = [float(rec[k]) for k in ('generator_stft', 'generator_adversarial_melgan', 'generator_features_melgan', 'discriminator_msd_melgan')]
    r = gold[i].tolist()
    print(i, ' '.join(f'{a:.5f}/{b:.5f}({abs(a - b) / b:.1e})' for a, b in zip(g, r)), flush=True)
