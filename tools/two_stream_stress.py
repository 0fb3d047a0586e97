"""Two-stream forward (engine.py: AERO_STREAMS auto, B >= 32) against the one-stream order, N times: waveform and spectrogram must be
identical every time.  usage: two_stream_stress.py [N]"""
import sys, os, json, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import GOLDEN, build_model
meta = json.load(open(os.path.join(GOLDEN,'meta.json')))
m = build_model(meta,'full').cuda()
eng = m._get_engine()
x = torch.randn(32,1,8000, generator=torch.Generator().manual_seed(5)).cuda()
def fwd():
    with torch.no_grad():
        y, s, lr = m(x, return_spec=True, return_lr_spec=True)
    torch.cuda.synchronize()
    return y.clone(), s.clone()
eng.streams = 1
y1, s1 = fwd()
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for it in range(N):
    eng.streams = 0 if it % 2 == 0 else 1
    y2, s2 = fwd()
    dy, ds = (y2-y1).abs().max().item(), (torch.view_as_real(s2)-torch.view_as_real(s1)).abs().max().item()
    if dy > 1e-6 or ds > 1e-6:
        bad += 1
        d=(y2-y1).abs()[:,0]
        nz=(d>1e-6).nonzero()
        print('iter', it, 'streams', eng.streams, 'dy', dy, 'ds', ds, 'clips', sorted(set(nz[:,0].tolist()))[:8], 'samples', int(nz[:,1].min()), int(nz[:,1].max()), nz.shape[0])
print('bad', bad, 'of', N)
