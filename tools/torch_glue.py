"""Which lines of the host side launch torch's own little kernels (fills, copies, adds) during one config-5 training step:
torch.profiler with Python stacks, device kernels attributed to the innermost frame under aero_amd/.  usage: torch_glue.py [B]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aero_amd import Aero, losses  # noqa: E402
from aero_amd.config import load_config  # noqa: E402
from aero_amd.optim import FlatAdam  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    args = load_config(os.path.join(ROOT, 'conf'), ['experiment=aero_11-44_512_256'])
    torch.manual_seed(2036)
    model = Aero(**dict(args.experiment.aero)).cuda().train()
    opt = FlatAdam(model.parameters(), lr=3e-4, model=model)
    crit = losses.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5)
    g = torch.Generator().manual_seed(0)
    lr = torch.randn(B, 1, 110250, generator=g).cuda()
    hr = (0.1 * torch.randn(B, 1, 441000, generator=g)).cuda()

    def step():
        y = model(lr)
        sc, mg = crit(y.squeeze(1), hr.squeeze(1))
        opt.zero_grad()
        (sc + mg).backward()
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    sites = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    total = [0, 0.0]
    for ev in prof.events():
        if not ev.name.startswith('aten::') or not ev.kernels:
            continue
        dev_us = sum(k.duration for k in ev.kernels)
        site = 'outside aero_amd'
        for fr in ev.stack:
            if 'aero_amd/' in fr:
                site = fr.split('aero_amd/')[-1]
                break
        s = sites[site]
        s[0] += len(ev.kernels)
        s[1] += dev_us
        s[2][ev.name] += len(ev.kernels)
        total[0] += len(ev.kernels)
        total[1] += dev_us
    print(f'B={B}: {total[0]} torch kernels, {total[1] / 1e3:.2f} ms of device time in one step')
    for k, v in sorted(sites.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('GLUE_ROWS', '60'))]:
        ops = ', '.join(f'{n.replace("aten::", "")} x{c}' for n, c in v[2].most_common(4))
        print(f'{v[1] / 1e3:7.3f} ms {v[0]:5d}  {k:60s} {ops}')


if __name__ == '__main__':
    main()
