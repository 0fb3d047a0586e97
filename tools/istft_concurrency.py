"""Reproduces the open iSTFT concurrency finding: the iSTFT kernel run on one stream while another stream runs a whole
forward (mode istft_vs_forward) now and then returns 512-sample blocks in which one frequency bin of the block's frames was read wrong;
next to another iSTFT (mode istft_vs_istft) it never does.  The product therefore runs the iSTFT after the two streams of a
forward have joined (aero_amd/engine.py).  usage: istft_concurrency.py [istft_vs_forward | istft_vs_istft | istft_vs_mm | istft_vs_elementwise]"""
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
    mma, mmb = torch.randn(4096, 4096, device='cuda', dtype=torch.float16), torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
    ew = torch.randn(64 << 20, device='cuda')
    from aero_amd import _lib
    lib = _lib.load()
    pat = (torch.arange(1 << 20, dtype=torch.int64, device='cuda') * 2246822519 % (1 << 32)).to(torch.int64)
    pat = (pat - (pat >= (1 << 31)).to(torch.int64) * (1 << 32)).to(torch.int32)
    cnt = torch.zeros(3, dtype=torch.int64, device='cuda')
    bad = 0
    mode = sys.argv[1] if len(sys.argv) > 1 else 'istft_vs_forward'
    for it in range(200):
        sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sb):
            if mode == 'istft_vs_forward':
                m(x[:16])                       # a whole forward of another half-batch keeps the chip busy with other kernels
            elif mode == 'probe':               # handled below: the bystander kernel instead of the iSTFT
                m(x[:16])
            elif mode == 'istft_vs_mm':         # library GEMMs (not this repo's kernels)
                for _ in range(12):
                    mmc = mma @ mmb
            elif mode == 'istft_vs_elementwise':    # torch's streaming kernels: no LDS at all
                for _ in range(40):
                    ew.mul_(1.0001).add_(0.5)
            else:
                zb = m._ispec(s0[16:].contiguous())
        with torch.cuda.stream(sa):
            if mode == 'probe':
                lib.call('aero_debug_probe', pat.data_ptr(), pat.numel(), 1024, 6, cnt.data_ptr(), sa.cuda_stream)
                za = ref[:16]
            else:
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
    print(mode, 'bad', bad, 'of 200', '| probe counters {changed LDS words, wrong loads, wrong twiddles}:', cnt.tolist())
