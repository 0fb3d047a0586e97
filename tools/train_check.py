"""Whole-model training-step check on the device: forward -> multi-resolution STFT loss -> HIP backward, every parameter gradient
against torch.autograd through the CPU oracle (fp32) on the same inputs.  usage: train_check.py small|full|stress_full [L] [--vjp]
--vjp: back-propagate the ORACLE's dL/dy through the HIP graph (isolates the backward pass from the loss gradient's sensitivity to
the 1e-3 forward error: d log|X| / dx blows up at near-zero bins)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import build_model, rel_l2, seeded  # noqa: E402
from aero_amd import losses  # noqa: E402
from oracle import aero_oracle as O  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'small'
    L = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 800
    vjp = '--vjp' in sys.argv
    dev = 'cuda'
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
    m = build_model(meta, which).train()
    cfg = meta[(which[7:] if which.startswith('stress_') else which) + '_cfg']
    x = seeded((2, 1, L), 7)
    hr = seeded((2, 1, 4 * L), 8) * 0.1
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    torch.set_num_threads(16)
    t = time.time()
    y_ref = O.aero_forward(sd, cfg, x, train=True, new_stats={})
    y_ref.retain_grad()
    sc, mg = O.mrstft_loss(y_ref.squeeze(1), hr.squeeze(1))
    (sc + mg).backward()
    print(f'oracle loss {float(sc):.6f} {float(mg):.6f}  ({time.time() - t:.1f} s)')
    m.to(dev)
    t = time.time()
    y = m(x.to(dev))
    torch.cuda.synchronize()
    print(f'forward rel-L2 {rel_l2(y.detach().cpu(), y_ref.detach()):.2e}  ({time.time() - t:.2f} s)')
    crit = losses.MultiResolutionSTFTLoss()
    sc2, mg2 = crit(y.squeeze(1), hr.to(dev).squeeze(1))
    print(f'hip loss    {float(sc2):.6f} {float(mg2):.6f}')
    t = time.time()
    if vjp:
        y.backward(y_ref.grad.to(dev))
    else:
        (sc2 + mg2).backward()
    torch.cuda.synchronize()
    print(f'backward {time.time() - t:.2f} s')
    rows = []
    for n, p in m.named_parameters():
        g_ref = sd[n].grad
        rows.append((n, rel_l2(p.grad.cpu(), g_ref), float(g_ref.norm())))
    gmax = max(r[2] for r in rows)
    live = [r for r in rows if r[2] > 1e-6 * gmax]
    for n, r, g in rows:
        print(f'{n:62s} {r:.2e} |g|={g:.2e}{"" if r < 1e-2 else "   <<<" if g > 1e-6 * gmax else "   (dead)"}')
    worst = sorted(live, key=lambda r: -r[1])[:5]
    print('parameters', len(rows), 'live', len(live), 'over 1e-2:', sum(r[1] >= 1e-2 for r in live), 'over 5e-2:', sum(r[1] >= 5e-2 for r in live))
    print('worst', worst)


if __name__ == '__main__':
    main()
