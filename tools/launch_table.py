"""Per-launch table of one forward at the BASELINE shape: kernel instantiation, shape note, HIP-event time, achieved rates.
    python tools/launch_table.py [--batch 64]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FULL_CFG  # noqa: E402
from aero_amd import Aero  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--config4', action='store_true', help='BASELINE config 4 (12->48 kHz, nfft 1024, hop 256, B = 32) instead of the headline config')
a = ap.parse_args()
torch.manual_seed(2036)
if a.config4:
    m = Aero(**dict(FULL_CFG, nfft=1024, hop_length=256, lr_sr=12000, hr_sr=48000)).eval().cuda()
    x = torch.randn(32 if a.batch == 64 else a.batch, 1, 24000, generator=torch.Generator().manual_seed(1000)).cuda()
else:
    m = Aero(**FULL_CFG).eval().cuda()
    x = torch.randn(a.batch, 1, 8000, generator=torch.Generator().manual_seed(1000)).cuda()
eng = m._get_engine()
with torch.no_grad():
    for _ in range(2):
        m(x)
    R = 3
    eng.ops.prof, eng.ops.prof_shapes = [], []
    for _ in range(R):
        m(x)
torch.cuda.synchronize()
n = len(eng.ops.prof) // R
tot = 0.0
for i in range(n):
    ms = sum(eng.ops.prof[i + r * n][3].elapsed_time(eng.ops.prof[i + r * n][4]) for r in range(R)) / R
    name, fl, nb = eng.ops.prof[i][:3]
    tot += ms
    print(f'{i:3d} {ms * 1e3:8.1f} us  {fl / ms / 1e9 if ms else 0:7.1f} TF/s {nb / ms / 1e6 if ms else 0:7.0f} GB/s  {name.replace("void ", "").split("(")[0]:46s} {eng.ops.prof_shapes[i]}')
print(f'sum of launches {tot:.3f} ms')
