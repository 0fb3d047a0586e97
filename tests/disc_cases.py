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
    xf, xr = (seeded((2, 1, 4096), 72) * 0.3).to(dev), (seeded((2, 1, 4096), 73) * 0.3).to(dev)
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


def _torch_critic(d, x, rounded=False):
    """the same weight-normed nn modules run by torch (fp32): discriminators.py:50-56,72-78"""
    results = []
    for key, disc in d.model.items():
        h = x
        feats = []
        for _, layer in disc.model.items():
            h = layer(h)
            if rounded:                 # the product's storage: fp16 feature maps (straight-through for the gradient)
                h = h + (h.half().float() - h).detach()
            feats.append(h)
        results.append(feats)
        x = d.downsample(x)
        if rounded:
            x = x + (x.half().float() - x).detach()
    return results


def case_critic_backward(dev, lib=None, T=1024, ndf=4, rel_to_g=False):
    """discriminator_loss (gradients of every weight_g / weight_v / bias) and generator_losses (gradient of the fake waveform) on the
    HIP kernels against torch.autograd through the same modules with the feature maps rounded to fp16 where the product stores them
    (a LeakyReLU mask is a discontinuous function of its pre-activation: see tests/train_cases.py)"""
    import torch.nn.functional as Fn
    from aero_amd.discriminators import Discriminator
    torch.manual_seed(5)
    d = Discriminator(num_D=3, ndf=ndf, n_layers=4, downsampling_factor=4)
    with torch.no_grad():
        for p in d.parameters():
            p.copy_(p.half().float())
    xf, xr = (seeded((2, 1, T), 1) * 0.3).half().float(), (seeded((2, 1, T), 2) * 0.3).half().float()
    # torch reference
    ref = {}
    of, orr = _torch_critic(d, xf, True), _torch_critic(d, xr, True)
    # the two hinge terms pull the critic's parameters in opposite directions with similar strength (fake ~ real at initialisation), so
    # their sum is a small difference of large numbers: errors are measured against the size of the TERMS, |g_fake| + |g_real|
    lf, lr_ = sum(Fn.relu(1 + s[-1]).mean() for s in of), sum(Fn.relu(1 - s[-1]).mean() for s in orr)
    dl = lf + lr_
    lf.backward()
    gf = {n: p.grad.clone() for n, p in d.named_parameters()}
    d.zero_grad()
    lr_.backward()
    gr = {n: p.grad.clone() for n, p in d.named_parameters()}
    ref['d'] = {n: gf[n] + gr[n] for n in gf}
    ref['dn'] = {n: float(gf[n].norm() + gr[n].norm()) for n in gf}
    d.zero_grad()
    xg = xf.clone().requires_grad_()
    of, orr = _torch_critic(d, xg, True), _torch_critic(d, xr, True)
    adv = sum(Fn.relu(1 - s[-1]).mean() for s in of)
    wts = (4.0 / 5) * (1.0 / 3)
    feat = 100.0 * sum(wts * Fn.l1_loss(of[i][j], orr[i][j].detach()) for i in range(3) for j in range(6))
    (adv + feat).backward()
    ref['dx'] = xg.grad.clone()
    d.zero_grad()
    if lib is not None:
        d.use_library(lib)
    d.to(dev)
    errs = {}
    loss = d.discriminator_loss(xf.to(dev), xr.to(dev))
    errs["d_loss"] = abs(float(loss.detach()) - float(dl.detach())) / float(dl.detach())
    loss.backward()
    for n, p in d.named_parameters():
        errs['d.' + n] = float((p.grad.cpu().double() - ref['d'][n].double()).norm()) / max(ref['dn'][n], 1e-30)
        if rel_to_g:                                            # ... and against the size of the (nearly cancelling) SUM itself
            errs['g.' + n] = float((p.grad.cpu().double() - ref['d'][n].double()).norm()) / max(float(ref['d'][n].double().norm()), 1e-30)
    xh = xf.to(dev).clone().requires_grad_()
    a2, f2 = d.generator_losses(xh, xr.to(dev))
    errs['adv'] = abs(float(a2) - float(adv)) / float(adv)
    errs['feat'] = abs(float(f2) - float(feat)) / float(feat)
    (a2 + f2).backward()
    errs['dx'] = rel_l2(xh.grad.cpu(), ref['dx'])
    # with FlatAdam attached the backward writes the same gradients straight into the optimizer's flat buffer (no per-parameter
    # tensors through autograd): identical values; a second backward without zero_grad accumulates
    from aero_amd.optim import FlatAdam
    plain = {n: p.grad.detach().clone() for n, p in d.named_parameters()}
    opt = FlatAdam(d.parameters(), lr=1e-4, model=d, lib=lib)
    opt.zero_grad()
    d.discriminator_loss(xf.to(dev), xr.to(dev)).backward()
    errs['sink'] = max(rel_l2(p.grad.cpu(), plain[n].cpu()) for n, p in d.named_parameters())
    assert all(p.grad.data_ptr() >= opt.flat_g.data_ptr() for p in d.parameters())
    (2.0 * d.discriminator_loss(xf.to(dev), xr.to(dev))).backward()
    errs['sink_acc'] = max(rel_l2(p.grad.cpu(), 3.0 * plain[n].cpu()) for n, p in d.named_parameters())
    return errs


def case_critic_replay(dev, lib=None, steps=3, ndf=16, T=2048):
    """the critic's device images (fp16 weights, MFMA images, the dense layer's conv / data-gradient images) after optimizer steps:
    replayed by aero_gather_pack from the weight-normed weights, bit-identical to what the packing closures build"""
    from aero_amd.discriminators import Discriminator
    from aero_amd.optim import FlatAdam
    from aero_amd.repack import _flatten
    torch.manual_seed(11)
    d = Discriminator(num_D=3, ndf=ndf, n_layers=4, downsampling_factor=4)
    if lib is not None:
        d.use_library(lib)
    d.to(dev)
    opt = FlatAdam(d.parameters(), lr=1e-3, model=d, lib=lib)
    xf, xr = (seeded((1, 1, T), 1) * 0.3).to(dev), (seeded((1, 1, T), 2) * 0.3).to(dev)
    for it in range(steps):
        loss = d.discriminator_loss(xf, xr)
        opt.zero_grad()
        loss.backward()
        opt.step()
        packed = d._pack(xf.device)
        if it >= 1:
            assert d._replay is not None and len(d._replay.objects) == 21 and not d._replay.skipped, d._replay and d._replay.skipped
        k = 0
        for si, layers in enumerate(packed):
            for j, ent in enumerate(layers):
                fresh = _flatten(d._builders[f'{si}.{j}'](), [])
                have = _flatten(ent, [])
                assert len(fresh) == len(have)
                for a, b in zip(have, fresh):
                    assert a.dtype == b.dtype and torch.equal(a, b), (it, si, j)
                k += 1
        assert k == 21
    return k
