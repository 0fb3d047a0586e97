"""Reproduces the open iSTFT concurrency finding: the iSTFT kernel run on one stream while another stream runs a whole
forward (mode istft_vs_forward) now and then returns 512-sample blocks in which one frequency bin of the block's frames was read wrong;
next to another iSTFT (mode istft_vs_istft) it never does.  The product therefore runs the iSTFT after the two streams of a
forward have joined (aero_amd/engine.py).  usage: istft_concurrency.py [istft_vs_forward | istft_vs_istft]"""
import sys, os, json, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import GOLDEN, build_model
meta = json.load(open(os.path.join(GOLDEN,'meta.json')))
m = build_model(meta,'full').cuda()
eng = m._get_engine(); eng.streams = 1
x = torch.randn(32,1,8000, generator=torch.Generator().manual_seed(5)).cuda()
with torch.no_grad():
    y0, s0 = m(x, return_spec=True)
    ref = m._ispec(s0).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    mode = sys.argv[1] if len(sys.argv) > 1 else 'istft_vs_forward'
    for it in range(200):
        sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sb):
            if mode == 'istft_vs_forward':
                m(x[:16])                       # a whole forward of another half-batch keeps the chip busy with other kernels
            else:
                zb = m._ispec(s0[16:].contiguous())
        with torch.cuda.stream(sa):
            za = m._ispec(s0[:16].contiguous())
        torch.cuda.synchronize()
        d = (za - ref[:16]).abs().max().item()
        if d > 1e-7:
            bad += 1
            nz = ((za - ref[:16]).abs()[:,0] > 1e-7).nonzero()
            print('iter', it, 'diff', d, 'clips', sorted(set(nz[:,0].tolist()))[:6], 'samples', int(nz[:,1].min()), int(nz[:,1].max()))
            if bad <= 2:
                c = int(nz[0,0]); s0_ = int(nz[0,1]) // 512 * 512
                dd = (za - ref[:16])[c,0,s0_:s0_+512]
                rr = ref[c,0,s0_:s0_+512]
                print('  block', c, s0_, 'nonzero diffs', int((dd.abs()>1e-7).sum()), 'first idx', (dd.abs()>1e-7).nonzero().flatten()[:8].tolist(), 'last', (dd.abs()>1e-7).nonzero().flatten()[-4:].tolist())
                print('  diff[::32]', [f'{v:.1e}' for v in dd[::32].tolist()])
                print('  ref [::32]', [f'{v:.1e}' for v in rr[::32].tolist()])
    print(mode, 'bad', bad, 'of 200')
