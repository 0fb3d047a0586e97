"""MelGAN multi-scale discriminator: the HIP forward against the reference's critic -- every feature map and the
logits of the three scales <= 1e-3, the hinge / feature-matching loss values of solver.py:489-520 <= 1e-4."""
import pytest

import disc_cases as dc


def _check(errs):
    maps = {k: v for k, v in errs.items() if k not in ('d_loss', 'g_adv', 'g_feat')}
    assert len(maps) == 24 and max(maps.values()) < 1e-3, maps
    assert errs['d_loss'] < 1e-4 and errs['g_adv'] < 1e-4 and errs['g_feat'] < 1e-4, errs


def test_discriminator_on_the_emulator():
    from aero_amd import _lib
    from emu.build_emu import build
    _check(dc.case_discriminator('cpu', lib=_lib.load(build())))


@pytest.mark.gpu
def test_discriminator_on_the_mi355x():
    _check(dc.case_discriminator('cuda'))


def _check_bwd(errs):
    """loss values <= 1e-4 / 1e-3; the gradient of the fake waveform (the generator's adversarial + feature-matching signal) <= 1e-2;
    every critic parameter gradient <= 3e-2 of |g_fake| + |g_real| (disc_cases.py: the two hinge terms nearly cancel; measured <= 1.6e-2,
    the grouped-conv kernels themselves are exact to fp32 rounding: test_grouped_conv_op)"""
    assert errs['d_loss'] < 1e-4 and errs['adv'] < 1e-4 and errs['feat'] < 1e-3, errs
    assert errs['dx'] < 1e-2, errs['dx']
    # (same kernels, same data; the narrow test critic's grouped layers take the VALU weight-gradient kernel with fp32 atomics: 1.5e-6 run to run)
    assert errs['sink'] < 1e-5 and errs['sink_acc'] < 1e-5, (errs['sink'], errs['sink_acc'])
    bad = {k: v for k, v in errs.items() if k.startswith('d.') and not v < 3e-2}       # (keys 'sink*' are checked above)
    assert not bad, bad


def test_critic_backward_on_the_emulator():
    from aero_amd import _lib
    from emu.build_emu import build
    _check_bwd(dc.case_critic_backward('cpu', lib=_lib.load(build())))


@pytest.mark.gpu
def test_critic_backward_on_the_mi355x():
    _check_bwd(dc.case_critic_backward('cuda', T=8192))


@pytest.mark.gpu
def test_critic_gradients_per_scale_at_full_width_on_the_mi355x():
    """in the first ADVERSARIAL step the critic's gradients were 0.05-0.8 % off at scale 0, ~1 % at scale 2 and 3-4.5 % at scale 1
    (profiles/r05_gan_step1.txt) -- an error that peaks at the MIDDLE scale looks like a defect of that scale's path (average-pool
    backward, data gradient), not like rounding.  There the critic saw the HIP generator's output, which differs from the reference's
    in the last fp16 bits and moves the hinge's active set.  Here the critic alone, at the width and depth the experiments train
    (ndf 16, 3 scales, 4 layers), on IDENTICAL inputs against torch.autograd: every parameter gradient of every scale within 3e-2 of
    |g_fake| + |g_real|, and no scale stands out from the other two."""
    errs = dc.case_critic_backward('cuda', T=16384, ndf=16, rel_to_g=True)
    per = {}
    for k, v in errs.items():
        if k.startswith('d.model.disc_'):
            per.setdefault(k.split('.')[2], []).append(v)
    worst = {s: max(v) for s, v in per.items()}
    mean = {s: sum(v) / len(v) for s, v in per.items()}
    gsum = {s: max(v for k, v in errs.items() if k.startswith('g.model.' + s)) for s in per}
    print('critic gradients per scale, worst tensor (of |g_fake| + |g_real|):', {s: f'{v:.2e}' for s, v in worst.items()},
          '| mean:', {s: f'{v:.2e}' for s, v in mean.items()}, '| worst tensor against |g| of the cancelling sum:', {s: f'{v:.2e}' for s, v in gsum.items()})
    assert sorted(per) == ['disc_0', 'disc_1', 'disc_2'] and all(len(v) >= 14 for v in per.values())
    assert max(worst.values()) < 3e-2, worst                                              # (measured 1.4e-2 / 1.4e-2 / 2.0e-2; the bar of test_critic_backward_*)
    assert worst['disc_1'] < 3 * max(worst['disc_0'], worst['disc_2']) + 2e-3, worst      # the middle scale is not special
    assert errs['dx'] < 1e-2


def test_critic_images_replayed_by_gather_on_the_emulator():
    from aero_amd import _lib
    from emu.build_emu import build
    assert dc.case_critic_replay('cpu', lib=_lib.load(build()), ndf=4, T=1024) == 21       # (a narrow critic: the emulated dense layer is slow)


@pytest.mark.gpu
def test_critic_images_replayed_by_gather_on_the_mi355x():
    assert dc.case_critic_replay('cuda') == 21


def _gconv_case(lib, dev, B, Cin, Cout, groups, K, stride, pad, T, reflect=0, slope=0.2, seed=0):
    """aero_gconv1d_fwd / aero_gconv1d_bwd against torch.nn.functional.conv1d + autograd on fp16-rounded operands"""
    import ctypes as C
    import torch
    import torch.nn.functional as F
    from aero_amd import _lib
    from aero_amd.engine import _ptr
    from conftest import rel_l2
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, T, generator=g).half().float().requires_grad_()
    w = (torch.randn(Cout, Cin // groups, K, generator=g) * 0.1).half().float().requires_grad_()
    b = torch.randn(Cout, generator=g).requires_grad_()
    xp = F.pad(x, (pad, pad), mode='reflect') if reflect else x
    y = F.leaky_relu(F.conv1d(xp, w, b, stride=stride, padding=0 if reflect else pad, groups=groups), slope)
    dy = torch.randn(y.shape, generator=g).half().float()
    y.backward(dy)
    xc = x.detach().permute(0, 2, 1).contiguous().half().to(dev)
    wc = w.detach().permute(0, 2, 1).contiguous().half().to(dev)
    bias = b.detach().to(dev)
    To = y.shape[2]
    yk = torch.empty(B, To, Cout, dtype=torch.float16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream if dev != 'cpu' else 0
    d = _lib.GconvDesc()
    d.x, d.w, d.bias, d.y = _ptr(xc), _ptr(wc), _ptr(bias), _ptr(yk)
    d.B, d.Tin, d.Cin, d.Cout, d.groups, d.K, d.stride, d.pad, d.reflect, d.slope = B, T, Cin, Cout, groups, K, stride, pad, reflect, slope
    lib.call('aero_gconv1d_fwd', C.byref(d), stream)
    assert rel_l2(yk.float().cpu().permute(0, 2, 1), y.detach()) < 5e-4
    dyc = dy.permute(0, 2, 1).contiguous().half().to(dev)
    dx = torch.empty(B, T, Cin, dtype=torch.float16, device=dev)
    dw, db = torch.zeros(Cout, K, Cin // groups, device=dev), torch.zeros(Cout, device=dev)
    bd = _lib.GconvBwdDesc()
    bd.x, bd.w, bd.y, bd.dy, bd.dx, bd.dw, bd.db = _ptr(xc), _ptr(wc), _ptr(yk), _ptr(dyc), _ptr(dx), _ptr(dw), _ptr(db)
    bd.B, bd.Tin, bd.Cin, bd.Cout, bd.groups, bd.K, bd.stride, bd.pad, bd.reflect, bd.slope = B, T, Cin, Cout, groups, K, stride, pad, reflect, slope
    lib.call('aero_gconv1d_bwd', C.byref(bd), stream)
    # (the LeakyReLU derivative is read off the fp16 output: an element whose fp32 pre-activation rounds across zero flips slope 0.2 <-> 1)
    assert rel_l2(dx.float().cpu().permute(0, 2, 1), x.grad) < 2e-3
    # (fp32 atomics over up to ~5e4 positions per output: the sum order is not fixed; and the same flipped LeakyReLU slopes as in dx --
    # measured 6e-4 on the 27 562-step layer, 1e-5 .. 2e-4 on the short ones)
    assert rel_l2(dw.cpu().permute(0, 2, 1), w.grad) < 1e-3 and rel_l2(db.cpu(), b.grad) < 1e-3
    if lib.cdll.aero_gconv1d_mfma_ok(Cin, Cout, groups, K, stride, pad, reflect):
        # the MFMA forms of the same layer (csrc/k_gconv_mfma.h): forward and data gradient from the two weight images
        from aero_amd.discriminators import gconv_mfma_images
        wf, wd = gconv_mfma_images(w.detach(), groups, dev)
        y2 = torch.full_like(yk, float('nan'))
        d.y, d.w_mfma = _ptr(y2), _ptr(wf)
        lib.call('aero_gconv1d_fwd', C.byref(d), stream)
        assert rel_l2(y2.float().cpu().permute(0, 2, 1), y.detach()) < 5e-4 and rel_l2(y2.float().cpu(), yk.float().cpu()) < 5e-4
        dx2 = torch.full_like(dx, float('nan'))
        bd.dx, bd.dw, bd.db, bd.w_dgrad_mfma = _ptr(dx2), None, None, _ptr(wd)
        lib.call('aero_gconv1d_bwd', C.byref(bd), stream)
        assert bool(torch.isfinite(dx2).all())                   # every input step written
        assert rel_l2(dx2.float().cpu().permute(0, 2, 1), x.grad) < 2e-3 and rel_l2(dx2.float().cpu(), dx.float().cpu()) < 1e-3
    # the slab forms of the weight gradient (MFMA for the grouped layers, dedicated kernels for the 1-channel ends of the critic, whose
    # data gradient then also takes its dedicated kernel): per-chunk partial sums added in order
    nsl = lib.cdll.aero_gconv1d_wgrad_slabs(B, T, Cin, Cout, groups, K, stride, pad, reflect)
    if nsl > 0:
        res = []
        for _ in range(2):
            dw2 = torch.zeros(Cout, K, Cin // groups, device=dev)
            db2 = torch.zeros(max(Cout, 4), device=dev)
            dx3 = torch.full_like(dx, float('nan'))
            slabs = torch.full((nsl, Cout * K * (Cin // groups) + max(Cout, 4)), float('nan'), device=dev)
            bd.dx, bd.dw, bd.db, bd.slabs, bd.nslab = _ptr(dx3), _ptr(dw2), _ptr(db2), _ptr(slabs), nsl
            lib.call('aero_gconv1d_bwd', C.byref(bd), stream)
            res.append((dw2.cpu(), db2.cpu()[:Cout], dx3.float().cpu()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])      # deterministic
        assert rel_l2(res[0][0].permute(0, 2, 1), w.grad) < 1e-3 and rel_l2(res[0][1], b.grad) < 1e-3
        assert rel_l2(res[0][0], dw.cpu()) < 2e-4 and rel_l2(res[0][1], db.cpu()) < 2e-4
        assert bool(torch.isfinite(res[0][2]).all()) and rel_l2(res[0][2].permute(0, 2, 1), x.grad) < 2e-3


GCONV = [(2, 16, 64, 4, 41, 4, 20, 300), (2, 256, 256, 64, 41, 4, 20, 16), (1, 64, 256, 16, 41, 4, 20, 139), (2, 1, 16, 1, 15, 1, 7, 200, 1), (2, 128, 1, 1, 3, 1, 1, 9, 0, 1.0),
         (2, 1, 16, 1, 15, 1, 7, 700, 1), (2, 1024, 1, 1, 3, 1, 1, 37, 0, 1.0), (1, 512, 1, 1, 3, 1, 1, 301, 0, 0.2),
         (2, 4, 16, 1, 41, 4, 20, 2048)]


@pytest.mark.parametrize('a', GCONV[:8])
def test_grouped_conv_op(a):
    from aero_amd import _lib
    from emu.build_emu import build
    _gconv_case(_lib.load(build()), 'cpu', *a)


@pytest.mark.gpu
@pytest.mark.parametrize('a', GCONV + [(2, 64, 256, 16, 41, 4, 20, 27562), (1, 1024, 1024, 256, 41, 4, 20, 6890)])
def test_grouped_conv_op_on_the_mi355x(a):
    from aero_amd import _lib
    _gconv_case(_lib.load(), 'cuda', *a)
