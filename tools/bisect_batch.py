"""Debug helper: batch-permutation invariance of Aero.forward at B=64 under the current environment flags.
    python tools/bisect_batch.py [--no-ftb0]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import build_model, rel_l2  # noqa: E402


def main():
    meta = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'meta.json')))
    m = build_model(meta, 'full').cuda()
    x = torch.randn(64, 1, 8000, generator=torch.Generator().manual_seed(1))
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(2))

    def fwd(inp):
        with torch.no_grad():
            y, s = m(inp.cuda(), return_spec=True)
        torch.cuda.synchronize()
        return s.cpu()
    s0 = fwd(x)
    if '--no-ftb0' in sys.argv:
        m._get_engine().collapse_first_ftb = False
        s0 = fwd(x)
    s1 = fwd(x)
    s2 = fwd(x)
    s3 = fwd(x)
    print(f'calls: 1v0 {rel_l2(s1, s0):.3e} 2v1 {rel_l2(s2, s1):.3e} 3v2 {rel_l2(s3, s2):.3e} 3v0 {rel_l2(s3, s0):.3e}', flush=True)
    d = (s1 - s0).abs().flatten(2).amax(2).squeeze()
    print('per-clip max diff per freq rows (clip 0):', [round(float(v), 4) for v in (s1 - s0).abs()[0, 0].amax(1)[::32]], flush=True)
    sp = fwd(x[perm])
    per = [(rel_l2(sp[i:i + 1], s0[perm][i:i + 1])) for i in range(64)]
    print(f'rerun {rel_l2(s1, s0):.3e}  perm {rel_l2(sp, s0[perm]):.3e}  worst clips {sorted(per)[-3:]}', flush=True)


if __name__ == '__main__':
    main()
