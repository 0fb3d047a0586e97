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
