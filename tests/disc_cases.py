"""MelGAN multi-scale discriminator (aero_amd/discriminators.py, csrc/k_disc.h) against the REFERENCE's critic
(tests/golden/disc_io.npz from oracle/make_golden.py: discriminators.py:14-78 at num_D 3, ndf 16, n_layers 4, factor 4)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN, load_npz, rel_l2, seeded


def case_discriminator(dev, lib=None):
    from aero_amd.discriminators import Discriminator, melgan_losses
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    io = load_npz('disc_io.npz')
    torch.manual_seed(meta['disc_seed'])
    d = Discriminator(**meta['disc_cfg']).eval()
    for k, v in d.state_dict().items():                                   # same seed -> the reference's initial weights, key by key
        cs = meta['disc_checksums'][k]
        assert abs(float(v.double().sum()) - cs[0]) <= 1e-6 * max(1.0, cs[1]) and abs(float(v.double().abs().sum()) - cs[1]) <= 1e-6 * cs[1], k
    assert set(d.state_dict()) == set(meta['disc_checksums'])
    if lib is not None:
        d.use_library(lib)
    d.to(dev)
    xf, xr = (seeded((2, 1, 8192), 72) * 0.3).to(dev), (seeded((2, 1, 8192), 73) * 0.3).to(dev)
    with torch.no_grad():
        of, orr = d(xf), d(xr)
    errs = {}
    assert len(of) == 3 and all(len(s) == 7 for s in of)
    for si, sc in enumerate(of):
        for j, fm in enumerate(sc):
            ref = io[f'fake.{si}.{j}']
            full = fm.float().cpu().numpy()
            assert full.shape[0] == 2 and full.ndim == 3
            got = full[:, ::max(1, full.shape[1] // 16), ::max(1, full.shape[2] // 64)]
            assert got.shape == ref.shape, (si, j, got.shape, ref.shape)
            errs[f'{si}.{j}'] = rel_l2(got, ref)
    for si, sc in enumerate(orr):
        errs[f'real.{si}'] = rel_l2(sc[-1].float().cpu(), io[f'real.{si}.6'])
    dl, ga, gf = melgan_losses(d, of, orr)
    lo = io['losses']
    errs['d_loss'] = abs(float(dl) - lo[0]) / lo[0]
    errs['g_adv'] = abs(float(ga) - lo[1]) / lo[1]
    errs['g_feat'] = abs(float(gf) - lo[2]) / lo[2]
    return errs
