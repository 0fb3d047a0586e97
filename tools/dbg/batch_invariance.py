"""Which launch makes clip 0's result depend on the batch size?  Runs the same clip at B=1 and inside a batch of B=k and
reports, launch by launch, the first outputs whose clip-0 slice differs (bitwise)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import build_model
from aero_amd.engine import Ops

which, L, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
m = build_model(meta, which).cuda()
x = torch.randn(k, 1, L, generator=torch.Generator().manual_seed(5)).cuda()
log = []
names = ['stft', 'spec_normalize', 'istft', 'conv', 'norm_act', 'gram_stats', 'lstm', 'localstate', 'freqfc', 'ftb_first']
cur = {'B': 1}
orig = {n: getattr(Ops, n) for n in names}
def wrap(n):
    def f(self, *a, **kw):
        out = orig[n](self, *a, **kw)
        t = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(t) and t.numel() % cur['B'] == 0:
            desc = n
            if n == 'conv':
                sp = a[0]; desc = f'conv M={sp.M} C0={sp.C0} C1={sp.C1} taps={len(sp.df)} tr={sp.transposed} Fin={a[4]} Fout={a[5]}'
            elif n == 'norm_act':
                desc = f'norm_act {tuple(a[0].shape[1:])} G={a[1]} per_row={a[2]} act={a[5]}'
            elif n == 'lstm':
                desc = f'lstm H={a[3]} W={a[5]}'
            log.append((desc, t.reshape(cur['B'], -1)[0].float().clone()))
        return out
    return f
for n in names:
    setattr(Ops, n, wrap(n))
runs = []
with torch.no_grad():
    for B in (1, k):
        cur['B'] = B; log.clear()
        m(x[:B]); torch.cuda.synchronize()
        runs.append(list(log))
a, b = runs
print(len(a), len(b))
shown = 0
for i, ((na, ta), (nb, tb)) in enumerate(zip(a, b)):
    if na != nb or ta.shape != tb.shape:
        print(i, 'MISMATCHED LAUNCH', na, nb); continue
    if not torch.equal(ta, tb):
        d = (ta.double() - tb.double()).norm() / tb.double().norm().clamp_min(1e-30)
        print(f'{i:3d} {na}: rel {float(d):.3e}  ndiff {(ta != tb).sum().item()}/{ta.numel()}')
        shown += 1
        if shown > 12: break
print('done')
