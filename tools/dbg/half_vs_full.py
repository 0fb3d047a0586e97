"""Why is the two-half-batch forward (engine.forward, AERO_STREAMS auto) not bit-equal to the one-stream forward at B = 64 while it is at
B = 32?  Runs the SAME clips as one batch of `k` and as a batch of `k // 2` (both on one stream), logs the first tensor every Ops call
returns, and reports launch by launch where the slice of the first k // 2 clips starts to differ; then repeats each run to tell a
batch-size dependent code path (same differences every time) from an order-of-atomics effect (differences come and go).
usage: half_vs_full.py [full|small] [L] [k]"""
import inspect, json, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import build_model
from aero_amd.engine import Ops

which = sys.argv[1] if len(sys.argv) > 1 else 'full'
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'meta.json')))
m = build_model(meta, which).cuda()
eng = m._get_engine()
eng.streams = 1
x = torch.randn(k, 1, L, generator=torch.Generator().manual_seed(5)).cuda()
log, cur = [], {'B': k}
names = [n for n in dir(Ops) if not n.startswith('_') and isinstance(inspect.getattr_static(Ops, n), types.FunctionType)
         and n not in ('stream', 'begin_step', 'new_stats')]


def wrap(n, fn):
    def f(self, *a, **kw):
        out = fn(self, *a, **kw)
        t = out[0] if isinstance(out, (tuple, list)) and out else out
        if torch.is_tensor(t) and t.dim() >= 1 and t.numel() % cur['B'] == 0 and t.numel() >= cur['B']:
            note = ''
            if n == 'conv':
                sp = a[0]
                note = f' M={sp.M} C={sp.C0}+{sp.C1} taps={len(sp.df)} tr={sp.transposed} F={a[4]}->{a[5]}'
            log.append((n + note, t.reshape(cur['B'], -1)[:k // 2].clone()))
        return out
    return f


for n in names:
    setattr(Ops, n, wrap(n, getattr(Ops, n)))


def run(B):
    cur['B'] = B
    log.clear()
    with torch.no_grad():
        y, s = m(x[:B], return_spec=True)
    torch.cuda.synchronize()
    return list(log), y[:k // 2].clone(), s[:k // 2].clone()


full = [run(k) for _ in range(3)]
half = [run(k // 2) for _ in range(3)]
print(f'{which} L={L}: B={k} vs B={k // 2} (first {k // 2} clips); launches logged {len(full[0][0])} / {len(half[0][0])}')
for tag, runs in (('full', full), ('half', half)):
    same = all(torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2]) for r in runs[1:])
    print(f'  {tag}: three repeats bit-equal among themselves: {same}')
print('  y equal:', torch.equal(full[0][1], half[0][1]), ' spec equal:', torch.equal(full[0][2], half[0][2]),
      ' ndiff y', int((full[0][1] != half[0][1]).sum()), 'of', full[0][1].numel())
shown = 0
for i, ((na, ta), (nb, tb)) in enumerate(zip(full[0][0], half[0][0])):
    if na != nb or ta.shape != tb.shape:
        print(f'{i:3d} DIFFERENT LAUNCH: {na} {tuple(ta.shape)} | {nb} {tuple(tb.shape)}')
        shown += 1
    elif not torch.equal(ta, tb):
        d = (ta.double() - tb.double()).norm() / tb.double().norm().clamp_min(1e-30)
        # which clips differ
        clips = (ta != tb).reshape(ta.shape[0], -1).any(1).nonzero().flatten().tolist()
        print(f'{i:3d} {na}: rel {float(d):.3e}  ndiff {int((ta != tb).sum())}/{ta.numel()}  clips {clips[:8]}')
        shown += 1
    if shown > 10:
        break
print('done')
