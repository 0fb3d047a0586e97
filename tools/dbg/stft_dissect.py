"""Round 4: the FFT-form STFT (Aero._spec(scale=True), the loss transforms) came out different in 299 of 300 rounds when the 192-row ring
conv tile ran on another stream (tests/test_gpu_concurrency.py).  This script dissects one such round: which elements differ and what
they hold, whether the baseline / the input were damaged instead, whether the output address matters, and which disturber it takes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import torch
    import concurrency_cases as cc
    from aero_amd import _lib
    from aero_amd.engine import _ptr
    from conftest import GOLDEN, build_model
    meta = json.load(open(os.path.join(GOLDEN, 'meta.json')))
    lib = _lib.load()
    m = build_model(meta, 'full').cuda()
    eng = m._get_engine()
    ops = eng.ops
    dist = cc.RingDisturber(lib, 'cuda')
    hr = (0.05 * torch.randn(16, 1, 32000, generator=torch.Generator().manual_seed(3))).cuda()
    hr_cpu = hr.cpu().clone()
    nfft, hop, win = m.nfft, int(m.hop_length * m.scale), int(m.win_length * m.scale)
    L = hr.shape[-1]
    pad = (m.hop_length - L % m.hop_length) % m.hop_length
    Lp = L + pad
    T = 1 + Lp // hop
    window = eng._window(win, hr.device)
    x2 = hr.reshape(16, L).contiguous()
    print('geometry: nfft', nfft, 'hop', hop, 'win', win, 'L', L, 'Lp', Lp, 'T', T, flush=True)

    def stft_into(buf):
        lib.call('aero_stft_fwd', _ptr(x2), 16, L, Lp, nfft, hop, _ptr(window), nfft // 2, _ptr(buf), T, None, 1, ops.stream(x2))
        return buf

    fixed = torch.zeros(16, nfft // 2, T, 2, device='cuda')
    ref = stft_into(fixed).clone()
    torch.cuda.synchronize()
    ref_cpu = ref.cpu().clone()
    mm_a = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)

    def d_mm(n):
        for _ in range(n):
            mm_a @ mm_a

    def d_none(n):
        pass

    def report(tag, out):
        d = (out - ref).abs()
        nz = (d.reshape(-1, 2).amax(1) > 0).nonzero().flatten()
        if nz.numel() == 0:
            return
        sig, rem = nz // (256 * T), nz % (256 * T)
        k, t = rem // T, rem % T
        print(f'  {tag}: {nz.numel()} complex elements differ; signals {sorted(set(sig.tolist()))[:8]} bins {sorted(set(k.tolist()))[:12]} '
              f'frames min {int(t.min())} max {int(t.max())} distinct {len(set(t.tolist()))}')
        o2, r2 = out.reshape(-1, 2), ref.reshape(-1, 2)
        for j in nz[:6].tolist():
            print(f'    [{j}] sig {j // (256 * T)} bin {(j % (256 * T)) // T} frame {j % T}: got {o2[j].tolist()} ref {r2[j].tolist()}')
        # is a whole block of frames off (one block = 8 frames of one signal)?
        blocks = sorted(set(((sig * ((T + 7) // 8)) + t // 8).tolist()))
        print(f'    distinct (signal, 8-frame block) pairs: {len(blocks)}; first {blocks[:6]}')

    for dname, dfn, nd in (('none', d_none, 0), ('ring192', dist.launch, 4), ('rocblas_mm', d_mm, 4)):
        for where in ('fixed buffer', 'fresh buffers'):
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            bad = 0
            shown = False
            keep = []
            for it in range(100):
                cur = torch.cuda.current_stream()
                sa.wait_stream(cur)
                sb.wait_stream(cur)
                with torch.cuda.stream(sa):
                    dfn(nd)
                with torch.cuda.stream(sb):
                    if where == 'fixed buffer':
                        fixed.zero_()
                        out = stft_into(fixed)
                    else:
                        out = stft_into(torch.empty_like(fixed))
                        keep.append(out)
                        keep = keep[-3:]
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad += 1
                    if not shown:
                        shown = True
                        report(f'{dname} / {where} round {it}', out)
            print(f'disturber {dname:10s} {where:13s}: {bad} of 100 rounds differ', flush=True)
    print('baseline intact:', torch.equal(ref.cpu(), ref_cpu), ' input intact:', torch.equal(hr.cpu(), hr_cpu), flush=True)
    # the same through the engine's own wrapper (allocates the output itself), as the failing test did
    bad, first = cc.overlapped(lambda: m._spec(hr, scale=True), dist.launch, 50, n_disturb=4)
    print('engine wrapper next to ring192:', bad, 'of 50', first)
    bad, first = cc.overlapped(lambda: m._spec(hr, scale=True), d_none, 50, n_disturb=0)
    print('engine wrapper, no disturber (two idle side streams):', bad, 'of 50', first)


if __name__ == '__main__':
    main()
