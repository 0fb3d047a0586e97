"""the config-5 generator step of bench.py (forward + MR-STFT loss + backward + Adam, 2 x 10-s clips) in a loop WITHOUT host synchronisation
between its phases -- the process tools/gpu/r6_train_timeline.sh traces (tools/config5.py synchronises after every phase for its own clock)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import FULL_CFG  # noqa: E402
from aero_amd import Aero, losses  # noqa: E402
from aero_amd.optim import FlatAdam  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(2036)
m = Aero(**dict(FULL_CFG, nfft=512, hop_length=256, lr_sr=11025, hr_sr=44100)).to(dev).train()
opt = FlatAdam(m.parameters(), lr=3e-4, betas=(0.9, 0.999), model=m)
crit = losses.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5)
g = torch.Generator().manual_seed(0)
lr_, hr_ = torch.randn(2, 1, 110250, generator=g).to(dev), (0.1 * torch.randn(2, 1, 441000, generator=g)).to(dev)


def step():
    y = m(lr_)
    sc, mg = crit(y.squeeze(1), hr_.squeeze(1))
    opt.zero_grad()
    (sc + mg).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for _ in range(K):
    step()
torch.cuda.synchronize()
print(f'{(time.perf_counter() - t0) / K * 1e3:.2f} ms per step')
