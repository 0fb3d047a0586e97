import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def meta():
    with open(os.path.join(GOLDEN, 'meta.json')) as f:
        return json.load(f)


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def rel_l2(a, b):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    if a.is_complex():
        a, b = torch.view_as_real(a), torch.view_as_real(b)
    a, b = a.double(), b.double()
    return float((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt().clamp_min(1e-30))


def seeded(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def randomize_running_stats(model, seed):
    g = torch.Generator().manual_seed(seed)
    for name, buf in model.named_buffers():
        if name.endswith('running_mean'):
            buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
        elif name.endswith('running_var'):
            buf.copy_(0.5 + torch.rand(buf.shape, generator=g))


def build_model(meta, which):
    """Seed-construct the tiny/small/full/wide model exactly as oracle/make_golden.py did.  `stress_small` /
    `stress_full` are the "trained-like" variants (oracle/stress.py: LayerScale O(1), live attention decay, ...)."""
    from aero_amd import Aero
    torch.manual_seed(meta[f'{which}_seed'])
    if which.startswith('stress_'):
        from oracle.stress import trained_like_
        return trained_like_(Aero(**meta[which[7:] + '_cfg']).eval(), meta[f'{which}_perturb_seed'])
    m = Aero(**meta[f'{which}_cfg']).eval()
    randomize_running_stats(m, meta[f'{which}_bn_seed'])
    return m
