"""Which host-side lines run torch's own ops during one training step (forward + loss + backward + Adam): a TorchDispatchMode
counts every aten call that produces a kernel launch on the device, attributed to the innermost frame under aero_amd/.  Runs on the
CPU emulation of the kernels with the small test model (same host code path as on the GPU).  usage: glue_audit.py [rows]"""
import collections
import json
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

NO_KERNEL = ('view', 'alias', 'detach', 'as_strided', 'reshape', 'permute', 'transpose', 'expand', 'slice', 'select', 'squeeze', 'unsqueeze',
             't.default', 'empty', 'sym_', '_unsafe_view', 'unbind', 'split', 'unfold', 'is_', 'size', 'stride', 'numel', 'dim', 'lift_fresh',
             'result_type', '_local_scalar_dense', 'resize_', 'set_', 'new_empty', 'view_as')


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if not any(name.startswith(p) for p in NO_KERNEL):
            site = 'outside aero_amd'
            for fr in reversed(traceback.extract_stack(limit=24)):
                if '/aero_amd/' in fr.filename:
                    site = f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}'
                    break
            self.sites[site][name] += 1
        return func(*args, **(kwargs or {}))


def critic():
    """the same count for the MelGAN critic: generator_losses (forward + backward to the fake waveform) and the critic's own step"""
    from conftest import seeded
    from emu.build_emu import build
    from aero_amd import _lib
    from aero_amd.discriminators import Discriminator
    from aero_amd.optim import FlatAdam
    lib = _lib.load(build())
    torch.manual_seed(3)
    d = Discriminator(num_D=3, ndf=16, n_layers=4, downsampling_factor=4)
    d.use_library(lib)
    opt = FlatAdam(d.parameters(), lr=1e-4, model=d, lib=lib)
    fake = (seeded((2, 1, 4096), 1) * 0.3).requires_grad_()
    real = seeded((2, 1, 4096), 2) * 0.3

    def step():
        fake.grad = None
        adv, feat = d.generator_losses(fake, real, n_layers=4, features_loss_lambda=100.0)
        (adv + feat).backward()
        dl = d.discriminator_loss(fake.detach(), real)
        opt.zero_grad()
        dl.backward()
        opt.step()
    for _ in range(2):
        step()
    a = Audit()
    with a:
        step()
    tot = sum(sum(c.values()) for c in a.sites.values())
    print(f'critic: {tot} torch ops with a kernel behind them in one generator-loss + critic step')
    for k, c in sorted(a.sites.items(), key=lambda kv: -sum(kv[1].values()))[:45]:
        print(f'{sum(c.values()):5d}  {k:48s} ' + ', '.join(f'{n} x{v}' for n, v in c.most_common(5)))


def main():
    if '--gan' in sys.argv:
        return critic()
    from conftest import GOLDEN, build_model, seeded
    from emu.build_emu import build
    from aero_amd import _lib, losses
    from aero_amd.engine import HipEngine
    from aero_amd.optim import FlatAdam
    lib = _lib.load(build())
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    m = build_model(meta, 'small').train()
    object.__setattr__(m, '_engine', HipEngine(m, lib=lib))
    losses.use_library(lib)
    x, hr = seeded((2, 1, 400), 1), seeded((2, 1, 1600), 2) * 0.1
    crit = losses.MultiResolutionSTFTLoss()
    opt = None
    try:
        opt = FlatAdam(m.parameters(), lr=1e-4, model=m, lib=lib)
    except Exception as e:                                   # noqa: BLE001
        print('(no FlatAdam on the emulator:', e, ')')

    def step():
        y = m(x)
        sc, mg = crit(y.squeeze(1), hr.squeeze(1))
        if opt is not None:
            opt.zero_grad()
        (sc + mg).backward()
        if opt is not None:
            opt.step()
    for _ in range(3):                                      # (the second step compiles the weight replay, aero_amd/repack.py)
        step()
    a = Audit()
    with a:
        step()
    tot = sum(sum(c.values()) for c in a.sites.values())
    print(f'{tot} torch ops with a kernel behind them in one step')
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    for k, c in sorted(a.sites.items(), key=lambda kv: -sum(kv[1].values()))[:rows]:
        print(f'{sum(c.values()):5d}  {k:48s} ' + ', '.join(f'{n} x{v}' for n, v in c.most_common(5)))


if __name__ == '__main__':
    main()
