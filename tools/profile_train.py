"""Where one training step of BASELINE config 5 spends its time: HIP-event timing of every C-ABI call of one forward + backward
(`lib.call` is wrapped), grouped by entry point and, for the weight gradients, by shape.  usage: profile_train.py [B]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aero_amd import Aero, _lib, losses  # noqa: E402
from aero_amd import backward as bw  # noqa: E402
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
    lib = _lib.load()
    rec = []
    orig_call = lib.call
    note = ['']

    def timed(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(name, *a)
        e1.record()
        nt = note[0]
        if name == 'aero_conv_fwd':                              # forward convs and data gradients: kernel instantiation + shape
            d = a[0]._obj
            kn = lib.cdll.aero_last_kernel_name().decode().replace('void ', '').split('(')[0]
            nt = f'{kn:40s} B{d.B} F{d.Fin}->{d.Fout} T{d.T} M{d.M} C{d.C0}+{d.C1 if d.src1 else 0} taps{d.ntaps} tr{d.transposed} act{d.act} stat{d.stat_mode}'
        if name.startswith('aero_norm_bwd'):                     # descriptor fields: the shape of this GroupNorm's backward
            d = a[0]._obj
            nt = f'B{d.B} F{d.F} T{d.T} C{d.C} G{d.G} per_row{d.per_row} act{d.act}'
        rec.append((name, nt, e0, e1))
    lib.call = timed
    orig_wgrad = bw.conv_wgrad

    def wgrad(ops, dy, x, df, dt, *a, **k):
        note[0] = f'dy{tuple(dy.shape)} x{tuple(x.shape)} taps{len(df)}'
        try:
            return orig_wgrad(ops, dy, x, df, dt, *a, **k)
        finally:
            note[0] = ''
    bw.conv_wgrad = wgrad
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    step()
    t1.record()
    torch.cuda.synchronize()
    lib.call = orig_call
    by = collections.defaultdict(lambda: [0, 0.0])
    wg = collections.defaultdict(lambda: [0, 0.0])
    for name, nt, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        by[name][0] += 1
        by[name][1] += ms
        if name == 'aero_conv_wgrad':
            wg[nt][0] += 1
            wg[nt][1] += ms
    tot = sum(v[1] for v in by.values())
    print(f'B={B}: one step {t0.elapsed_time(t1):.1f} ms wall (with per-call events), {len(rec)} C-ABI calls, {tot:.1f} ms inside them')
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f'{v[1]:8.2f} ms {v[0]:5d} calls  {k}')
    print('-- aero_conv_wgrad by shape')
    import re
    ideal_tot = 0.0
    for k, v in sorted(wg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('WGRAD_ROWS', '16'))]:
        # memory floor of one call: both operands read once at 4 TB/s (what a streaming kernel reaches here), dw written once
        dims = [int(t) for t in re.findall(r'\d+', k)]
        dyn = dims[0] * dims[1] * dims[2] * dims[3]
        xn = dims[4] * dims[5] * dims[6] * dims[7]
        byt = 2 * (dyn + xn) + 4 * dims[8] * dims[3] * dims[7]
        fl = 2.0 * dyn * dims[7] * dims[8]
        floor = max(byt / 4e12, fl / 1.2e15) * 1e3
        ideal_tot += floor * v[0]
        print(f'{v[1]:8.3f} ms {v[0]:3d} calls  {k}   per call {v[1] / v[0] * 1e3:7.1f} us, floor {floor * 1e3:6.1f} us')
    print(f'   (sum of floors of the rows shown: {ideal_tot:.2f} ms)')
    print('-- aero_conv_fwd (forward convs + data gradients) by kernel and shape: executed TFLOP/s assume every tap valid')
    cf = collections.defaultdict(lambda: [0, 0.0])
    for name, nt, e0, e1 in rec:
        if name == 'aero_conv_fwd':
            cf[nt][0] += 1
            cf[nt][1] += e0.elapsed_time(e1)
    for k, v in sorted(cf.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('CONV_ROWS', '28'))]:
        m = re.search(r'B(\d+) F(\d+)->(\d+) T(\d+) M(\d+) C(\d+)\+(\d+) taps(\d+)', k)
        Bq, Fi, Fo, Tq, Mq, c0, c1, tp = (int(g) for g in m.groups())
        fl = 2.0 * Bq * Fo * Tq * Mq * (c0 + c1) * tp
        print(f'{v[1]:8.3f} ms {v[0]:3d} calls  {k}   {v[1] / v[0] * 1e3:7.1f} us  {fl / (v[1] / v[0]) / 1e9:7.1f} TF/s')
    print('-- GroupNorm backward by shape (reduce | apply; floor: x, dy read twice + dx written once at 5 TB/s)')
    nb = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, nt, e0, e1 in rec:
        if name.startswith('aero_norm_bwd'):
            ms = e0.elapsed_time(e1)
            nb[nt][0] += name.endswith('reduce')
            nb[nt][1 if name.endswith('reduce') else 2] += ms
    for k, v in sorted(nb.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        dims = [int(t) for t in re.findall(r'\d+', k)]
        n = dims[0] * dims[1] * dims[2] * dims[3]
        glu = dims[6] == 3
        floor = (2 * n * 2 + 2 * (n // 2 if glu else n) * 2 + 2 * n) / 5e12 * 1e6
        print(f'{v[1] + v[2]:8.3f} ms {v[0]:3d} norms  {k:46s} per norm: reduce {v[1] / v[0] * 1e3:6.1f} us, apply {v[2] / v[0] * 1e3:6.1f} us, floor {floor:5.1f} us')


if __name__ == '__main__':
    main()
