"""Cross-stream interference cases: a VICTIM runs on one HIP stream while a DISTURBER keeps another
stream busy with the 192-row tile of the software-pipelined ring conv kernel (the decoder-2/3 3x3 shapes) -- the neighbour next to which
the round-3 iSTFT returned wrong 512-sample blocks in 2-14 % of forwards.  Every victim result is compared BIT FOR BIT with the same
call made with the chip to itself.  Shared by tests/test_gpu_concurrency.py (the regression fence) and tools/dbg/istft_probe.py."""
import math

import torch

from aero_amd import _lib, pack
from aero_amd.engine import Ops


class RingDisturber:
    """decoder-3-shaped 3x3 conv (two 48-channel sources -> 192 rows, GLU, F = 64, T = 501): aero_conv_ring_kernel<2, 4, 3, 3, 0>"""

    def __init__(self, lib, dev, B=8, C=48, M=192, Fq=64, T=501, seed=7):
        self.ops = Ops(lib)
        g = torch.Generator().manual_seed(seed)
        w = torch.randn(M, 2 * C, 3, 3, generator=g) / math.sqrt(2 * C * 9)
        b = torch.randn(M, generator=g)
        taps, df, dt = pack.conv2d_taps(w.half().float(), 1, 1)
        self.spec = pack.make_conv_spec(taps, b, C, C, df, dt, dev, fstride=1, act=_lib.ACT_GLU)
        x = torch.randn(B, Fq, T, 2 * C, generator=g).half().to(dev)
        self.s0, self.s1 = x[..., :C].contiguous(), x[..., C:].contiguous()
        self.dst = torch.empty(B, Fq, T, M // 2, dtype=torch.float16, device=dev)
        self.geo = (B, Fq, Fq, T)
        d = _lib.ConvDesc()
        self.name = None

    def kernel_name(self):
        """the instantiation the library picks for this launch (must be the 192-row ring tile for the test to mean anything)"""
        ops = self.ops
        ops.prof, keep = [], ops.prof
        try:
            self.launch(1)
            torch.cuda.synchronize()
            return ops.prof[-1][0]
        finally:
            ops.prof = keep

    def launch(self, n):
        B, Fi, Fo, T = self.geo
        for _ in range(n):
            self.ops.conv(self.spec, self.s0, self.s1, B, Fi, Fo, T, dst=self.dst)


def overlapped(victim, disturber, iters, n_disturb=24, compare=None):
    """victim() -> tensor or tuple of tensors.  Reference = victim() alone; then `iters` rounds of {disturber on stream A, victim on
    stream B}.  Returns (number of rounds whose result differs in any bit, description of the first difference)."""
    def as_tuple(r):
        r = tuple(t for t in (r if isinstance(r, (tuple, list)) else (r,)) if t is not None)
        return tuple(torch.view_as_real(t) if t.is_complex() else t for t in r)
    with torch.no_grad():
        ref = tuple(t.clone() for t in as_tuple(victim()))
        torch.cuda.synchronize()
        again = as_tuple(victim())
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(again, ref)):
            return -1, 'the victim is not bit-reproducible on its own'
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        bad, first = 0, ''
        for it in range(iters):
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                disturber(n_disturb)
            with torch.cuda.stream(sb):
                out = as_tuple(victim())
            torch.cuda.synchronize()
            for k, (a, b) in enumerate(zip(out, ref)):
                if not torch.equal(a, b):
                    bad += 1
                    if not first:
                        d = (a.float() - b.float()).abs()
                        nz = (d.reshape(-1) > 0).nonzero().flatten()
                        first = (f'round {it}, output {k} {tuple(a.shape)}: {int(nz.numel())} elements differ, max {float(d.max()):.3e} '
                                 f'(|ref| max {float(b.float().abs().max()):.3e}), flat index {int(nz[0])}..{int(nz[-1])}')
                    break
        return bad, first
