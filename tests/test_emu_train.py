"""CPU: one whole training step of the generator -- `Aero.forward` under autograd -> multi-resolution STFT loss -> HIP backward --
through the CPU emulation of the kernels (test double), against the reference-generated golden and torch.autograd through the
oracle (tests/train_cases.py for what is compared and why).  SURVEY.md 8 f1; solver.py:296-305,560-584,602-605."""
import torch

import train_cases as tc
from aero_amd import _lib


def test_training_step_small_model_on_the_emulator():
    from emu.build_emu import build
    rows = tc.case_training_step_small('cpu', lib=_lib.load(build()))
    assert len(rows) > 250                                    # every parameter with a gradient was compared


def test_partially_frozen_generator_on_the_emulator():
    from emu.build_emu import build
    assert tc.case_partially_frozen('cpu', lib=_lib.load(build())) > 20


def test_weight_images_replayed_by_gather_match_their_closures():
    from emu.build_emu import build
    assert tc.case_weight_replay('cpu', lib=_lib.load(build())) > 50


def test_autograd_needs_the_device_library():
    """no CPU / eager fallback in the product: a CPU model under autograd without the emulator raises"""
    import json
    import os
    import pytest
    from conftest import GOLDEN, build_model
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    m = build_model(meta, 'tiny').train()
    from aero_amd.engine import HipEngine
    with pytest.raises((RuntimeError, ImportError, OSError)):
        m(torch.zeros(1, 1, 400))
