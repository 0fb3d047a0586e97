"""Op-level parity cases shared by the CPU-emulator tests (tests/test_emu_ops.py) and the GPU tests
(tests/test_gpu_ops.py).  Every case runs one C-ABI entry point through aero_amd.engine.Ops and checks
it against the CPU oracle / plain fp32 torch on the same seeded inputs.

Tolerances: STFT/iSTFT are fp32 -> 2e-6 rel-L2; everything that goes through fp16 operands/storage
-> 2e-3 rel-L2 against the fp32 reference computed on the fp16-ROUNDED inputs (so the residual is only
accumulation order + output rounding; the end-to-end budget of 1e-3 on the spectrogram is tested in
test_*_model.py).
"""
import math

import torch
import torch.nn.functional as F

from aero_amd import _lib, pack
from aero_amd.engine import Ops, _hann_padded
from conftest import rel_l2
from oracle import aero_oracle as O

TOL16 = 2e-3
PW_CASES = [dict(Cin=48, Cout=384, Fq=3, T=70, norm=True, residual=True, scale=True),            # DConv tail, encoder 2 (KS 2, GW 3)
            dict(Cin=96, Cout=768, Fq=2, T=37, norm=True, residual=True, scale=True),            # DConv tail, encoder 3 (KS 3, GW 2, 3 chunks)
            dict(Cin=48, Cout=96, Fq=5, T=50, post=True),                                        # encoder-0 rewrite + GLU + frequency embedding (M padded to a group)
            dict(Cin=96, Cout=192, Fq=2, T=33),                                                  # encoder-1 rewrite + GLU
            dict(Cin=24, Cout=64, Fq=2, T=17, act='relu', residual=True),                        # non-GLU store path, one k-step
            dict(Cin=16, Cout=32, Fq=1, T=5, act='none', B=1),
            dict(Cin=96, Cout=48, Fq=4, T=70, act='relu', split=48),                            # FTB conv2 over cat([att, x]): a k-step spans both sources
            dict(Cin=48, Cout=32, Fq=2, T=21, act='relu', split=24)]
# (round 6: the weights-in-LDS form for 96 < C <= 384, GELU and GroupNorm without GLU are no longer instantiated: nothing launched them)
TOL32 = 2e-6
BLSTM_TOL = 1e-3      # whole BLSTM block / LocalState block (output incl. the skip path) against the oracle: the north-star bar itself
ATTN_TOL = 1e-3


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _rand(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=_g(seed)) * scale


def cl(x):
    """reference layout [B,C,F,T] fp32 -> channels-last fp16 [B,F,T,C]"""
    return x.permute(0, 2, 3, 1).contiguous().half()


def uncl(y):
    return y.float().permute(0, 3, 1, 2)


def q16(x):
    return x.half().float()


# ------------------------------------------------------------------------------------------------
def case_stft(lib, dev, nfft, hop, win, L, B=2, nyquist=False, dft=False):
    """nyquist=True keeps bin n_fft/2 (the loss / metric STFTs of stft_loss.py:22 and metrics.py:50 are one-sided with it);
    dft=True: the short-window GEMM form (aero_stft_dft_fwd) instead of the FFT kernel"""
    ops = Ops(lib)
    ops.dft_stft = dft
    x = _rand((B, L), 1)
    pad = (hop - L % hop) % hop
    stats = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    z = ops.stft(x.to(dev), L, L + pad, nfft, hop, _hann_padded(win, nfft, dev), nfft // 2 + int(nyquist), stats=stats,
                 win_len=win if dft else None)
    if dft:
        assert 'dft' in (ops.lib.cdll.aero_last_kernel_name().decode() or 'dft')
    zr = O.stft(F.pad(x, (0, pad)), nfft, hop, win)
    if not nyquist:
        zr = zr[..., :-1, :]
    assert rel_l2(torch.view_as_complex(z.cpu()), zr) < TOL32
    v = torch.view_as_real(zr).double()
    ref = torch.stack([v.sum(dim=(1, 2, 3)), (v * v).sum(dim=(1, 2, 3))], 1)
    assert torch.allclose(stats.cpu(), ref, rtol=1e-5, atol=1e-3)
    # K2: normalisation
    xn, ms = ops.spec_normalize(z, B, stats)
    vr = torch.view_as_real(zr)
    mean = vr.mean(dim=(1, 2, 3), keepdim=True)
    std = vr.std(dim=(1, 2, 3), keepdim=True)
    assert torch.allclose(ms.cpu(), torch.cat([mean.view(B, 1), std.view(B, 1)], 1), rtol=1e-5, atol=1e-6)
    assert rel_l2(xn.cpu().float(), (vr - mean) / (1e-5 + std)) < 1e-3
    if dft:
        # round 6: the same GEMM run twice (sums, then the normalised fp16 store; aero_stft_dft_norm_fwd) -- no fp32 spectrogram in HBM.
        # Same arithmetic and summation order: BIT-identical to the pair above (index paths exact)
        stats2 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        fused = ops.stft_normalized(x.to(dev), L, L + pad, nfft, hop, _hann_padded(win, nfft, dev), nfft // 2, stats2, 1, win)
        assert fused is not None
        assert torch.equal(fused[0].cpu(), xn.cpu()) and torch.equal(fused[1].cpu(), ms.cpu()) and torch.equal(stats2.cpu(), stats.cpu())


def case_istft(lib, dev, nfft, hop, win, T, crop=5, B=2):
    ops = Ops(lib)
    z = torch.complex(_rand((B, nfft // 2, T), 2), _rand((B, nfft // 2, T), 3))
    w = _hann_padded(win, nfft, 'cpu')
    env = torch.zeros(nfft + hop * (T - 1), dtype=torch.float64)
    for t in range(T):
        env[t * hop:t * hop + nfft] += (w * w).double()
    Lout = hop * (T - 1) - crop
    y = ops.istft(torch.view_as_real(z).contiguous().to(dev), nfft, hop, w.to(dev), (1 / env).float().to(dev), Lout)
    yr = O.istft(F.pad(z, (0, 0, 0, 1)), hop, win)[..., :Lout]
    assert rel_l2(y.cpu(), yr) < TOL32
    # round 6: the pitched layout (rows at a multiple of 16 frames, frame 0 in column t_off; pad columns poisoned): same bits
    pitch, toff = ops.istft_pitch(nfft, hop, T)
    if (pitch, toff) != (T, 0):
        assert pitch % 16 == 0 and pitch >= toff + T
        buf = torch.full((B, nfft // 2, pitch, 2), float('nan'))
        buf[:, :, toff:toff + T] = torch.view_as_real(z)
        buf = buf.to(dev)
        buf._aero_pitched = (T, toff)
        y2 = ops.istft(buf, nfft, hop, w.to(dev), (1 / env).float().to(dev), Lout)
        assert torch.equal(y2.cpu(), y.cpu())


# ------------------------------------------------------------------------------------------------
def case_conv2d(lib, dev, Cin, Cout, kF, kT, stride, padF, padT, Fin, T, act='none', B=2, split=None, null0=False,
                residual=False, seed=10):
    """Conv2d [kF,kT] stride [stride,1] against F.conv2d; optional concat of two sources, NULL first source,
    fused GELU/RELU/GLU, residual add."""
    ops = Ops(lib)
    w = _rand((Cout, Cin, kF, kT), seed, 1.0 / math.sqrt(Cin * kF * kT))
    b = _rand((Cout,), seed + 1)
    x = _rand((B, Cin, Fin, T), seed + 2)
    if null0:
        x[:, :split] = 0
    taps, df, dt = pack.conv2d_taps(q16(w), padF, padT)
    actc = {'none': _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'gelu': _lib.ACT_GELU, 'glu': _lib.ACT_GLU}[act]
    C0 = Cin if split is None else split
    spec = pack.make_conv_spec(taps, b, C0, Cin - C0, df, dt, dev, fstride=stride, act=actc)
    Fout = (Fin + 2 * padF - kF) // stride + 1
    xcl = cl(x).to(dev)
    s0 = None if null0 else xcl[..., :C0]
    s1 = xcl[..., C0:] if split is not None else None
    ref = F.conv2d(q16(x), q16(w), b, stride=(stride, 1), padding=(padF, padT))
    ref = {'none': lambda v: v, 'relu': F.relu, 'gelu': F.gelu, 'glu': lambda v: F.glu(v, 1)}[act](ref)
    res = None
    if residual:
        r = _rand(tuple(ref.shape), seed + 3)
        res = cl(r).to(dev)
        ref = ref + q16(r)
    y = ops.conv(spec, s0, s1, B, Fin, Fout, T, res=res)
    assert y.shape == (B, Fout, T, ref.shape[1])
    assert rel_l2(uncl(y.cpu()), ref) < TOL16


def case_pw(lib, dev, Cin, Cout, Fq, T, B=2, act='glu', norm=False, residual=False, post=False, scale=False, seed=16, split=None):
    """aero_pw_fwd (k_pw.h): 1x1 conv [-> GroupNorm(1 group per (b, f) row) from given sums] -> act [* LayerScale] [+ res] [+ frequency
    embedding row] against fp32 torch on the fp16-rounded operands"""
    ops = Ops(lib)
    g = _g(seed)
    w = torch.randn(Cout, Cin, generator=g) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    actc = {'none': _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'gelu': _lib.ACT_GELU, 'glu': _lib.ACT_GLU}[act]
    spec = pack.make_pw_spec(q16(w), b, actc, lib, dev)
    assert spec is not None
    v = torch.einsum('mc,bcft->bmft', q16(w), q16(x)) + b.view(1, -1, 1, 1)
    kw = {}
    if norm:
        gamma, beta = 1 + 0.3 * torch.randn(Cout, generator=g), 0.2 * torch.randn(Cout, generator=g)
        mu = v.mean(dim=(1, 3), keepdim=True)
        var = v.var(dim=(1, 3), unbiased=False, keepdim=True)
        st = torch.stack([v.double().sum(dim=(1, 3)), v.double().square().sum(dim=(1, 3))], -1).reshape(B * Fq, 2)   # row b*F + f
        v = (v - mu) / torch.sqrt(var + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        gi, bi = (pack.glu_interleave(gamma), pack.glu_interleave(beta)) if act == 'glu' else (gamma, beta)
        kw.update(stats=st.to(dev), count=float(T * Cout), gamma=gi.to(dev).contiguous(), beta=bi.to(dev).contiguous())
    ref = {'none': lambda u: u, 'relu': F.relu, 'gelu': F.gelu, 'glu': lambda u: F.glu(u, 1)}[act](v)
    Mout = ref.shape[1]
    if scale:
        ls = 0.5 + torch.rand(Mout, generator=g)
        ref = ref * ls.view(1, -1, 1, 1)
        kw['layer_scale'] = ls.to(dev)
    if residual:
        r = torch.randn(B, Mout, Fq, T, generator=g)
        ref = ref + q16(r)
        kw['res'] = cl(r).to(dev)
    if post:
        pe = torch.randn(Fq, Mout, generator=g)
        ref = ref + pe.t().reshape(1, Mout, Fq, 1)
        kw['post_add'] = pe.to(dev).contiguous()
    xc = cl(x).to(dev)
    if split:                                                     # two sources: cat([x[:split], x[split:]], channel)
        y = ops.pw(spec, xc[..., :split].contiguous(), B, Fq, T, x1=xc[..., split:].contiguous(), **kw)
    else:
        y = ops.pw(spec, xc, B, Fq, T, **kw)
    assert y.shape == (B, Fq, T, Mout)
    e = rel_l2(uncl(y.cpu()), ref)
    assert e < TOL16, e


def case_squeeze(lib, dev, Cin, Fq, T, M=5, B=2, seed=18):
    """aero_squeeze_fwd: Conv2d(C, M, 1x1) + bias + ReLU written as the [B, T, F*rp] image of the FTB's Conv1d (k_pw.h)"""
    ops = Ops(lib)
    g = _g(seed)
    w = torch.randn(M, Cin, generator=g) / math.sqrt(Cin)
    b = torch.randn(M, generator=g)
    x = torch.randn(B, Cin, Fq, T, generator=g)
    rp = M if (Fq * M) % 8 == 0 else 8
    spec = pack.make_squeeze_spec(q16(w), b, _lib.ACT_RELU, dev)
    assert spec is not None
    dst = torch.zeros(B, T, Fq * rp, dtype=torch.float16, device=dev)
    ops.squeeze(spec, cl(x).to(dev), B, Fq, T, dst, rp)
    ref = F.relu(torch.einsum('mc,bcft->bmft', q16(w), q16(x)) + b.view(1, -1, 1, 1))          # [B, M, F, T]
    got = dst.cpu().float().view(B, T, Fq, rp)[..., :M].permute(0, 3, 2, 1)
    assert rel_l2(got, ref) < TOL16
    if rp > M:
        assert float(dst.cpu().float().view(B, T, Fq, rp)[..., M:].abs().max()) == 0.0


def case_conv_tiny(lib, dev, Cin, Cout, Fq, T, B=2, act='relu', seed=15):
    """Pointwise conv with few channels written into a frequency-major [B, T, F*M] destination (the first FTB's squeeze,
    engine._encode): exercises aero_conv_tiny_kernel."""
    ops = Ops(lib)
    w = _rand((Cout, Cin, 1, 1), seed, 1.0 / math.sqrt(Cin))
    b = _rand((Cout,), seed + 1)
    x = _rand((B, Cin, Fq, T), seed + 2)
    taps, df, dt = pack.conv2d_taps(q16(w), 0, 0)
    actc = {'none': _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'gelu': _lib.ACT_GELU}[act]
    spec = pack.make_conv_spec(taps, b, Cin, 0, df, dt, dev, act=actc)
    dst = torch.zeros(B, T, Fq * Cout, dtype=torch.float16, device=dev)
    ops.conv(spec, cl(x).to(dev), None, B, Fq, Fq, T, dst=dst, dst_strides=(T * Fq * Cout, Cout, Fq * Cout))
    ref = F.conv2d(q16(x), q16(w), b)
    ref = {'none': lambda v: v, 'relu': F.relu, 'gelu': F.gelu}[act](ref)            # [B, M, F, T]
    got = dst.cpu().float().view(B, T, Fq, Cout).permute(0, 3, 2, 1)
    assert rel_l2(got, ref) < TOL16


def case_conv_stats(lib, dev, Cin, Cout, kF, kT, Fq, T, G=1, per_row=False, B=2, seed=17):
    """GroupNorm statistics accumulated by the conv epilogue (stat_mode 1) against sums over the conv's own output."""
    ops = Ops(lib)
    w = _rand((Cout, Cin, kF, kT), seed, 1.0 / math.sqrt(Cin * kF * kT))
    b = _rand((Cout,), seed + 1)
    x = _rand((B, Cin, Fq, T), seed + 2)
    taps, df, dt = pack.conv2d_taps(q16(w), kF // 2, kT // 2)
    spec = pack.make_conv_spec(taps, b, Cin, 0, df, dt, dev)
    st = ops.new_stats(B, Fq, G, per_row, dev)
    y = ops.conv(spec, cl(x).to(dev), None, B, Fq, Fq, T, stat=dict(mode=1, stats=st, G=G, per_row=per_row))
    ref = F.conv2d(q16(x), q16(w), b, padding=(kF // 2, kT // 2)).double()            # [B, M, F, T]
    assert rel_l2(uncl(y.cpu()), ref.float()) < TOL16
    r = ref.view(B, G, Cout // G, Fq, T)
    if per_row:
        s1, s2 = r.sum((2, 4)).permute(0, 2, 1), (r * r).sum((2, 4)).permute(0, 2, 1)          # [B, F, G]
    else:
        s1, s2 = r.sum((2, 3, 4)), (r * r).sum((2, 3, 4))                                      # [B, G]
    got = st.cpu().view(*s1.shape, 2)
    assert torch.allclose(got[..., 1], s2, rtol=2e-4), (got[..., 1].flatten()[:4], s2.flatten()[:4])
    assert torch.allclose(got[..., 0], s1, rtol=1e-3, atol=1e-3 * float(s2.sqrt().mean()))


def case_gram_stats(lib, dev, Cc, M, Fq, T, B=2, pitch=None, seed=19):
    """aero_gram_stats: sum / sum of squares of y = W x + b per (b, f) row from the Gram matrix of x, against the direct
    evaluation in float64 (same fp16-rounded W and x)."""
    ops = Ops(lib)
    w = _rand((M, Cc), seed, 1.0 / math.sqrt(Cc))
    b = _rand((M,), seed + 1)
    x = _rand((B, Fq, T, Cc), seed + 2)
    pitch = pitch or Cc
    buf = torch.zeros(B, Fq, T, pitch, dtype=torch.float16)
    buf[..., :Cc] = x.half()
    xd = buf.to(dev)[..., :Cc]
    tables = pack.gram_tables(w, b, dev)
    st = torch.full((B * Fq, 2), -1.0, dtype=torch.float64, device=dev)
    ops.gram_stats(xd, tables, st)
    y = q16(x).double() @ q16(w).double().t() + b.double()                                  # [B, F, T, M]
    s1, s2 = y.sum((2, 3)).reshape(-1), (y * y).sum((2, 3)).reshape(-1)
    got = st.cpu()
    assert torch.allclose(got[:, 1], s2, rtol=2e-5), (got[:4, 1], s2[:4])
    assert torch.allclose(got[:, 0], s1, rtol=1e-4, atol=1e-4 * float(s2.sqrt().mean()))


def case_conv1d(lib, dev, Cin, Cout, k, dil, R, T, seed=20):
    ops = Ops(lib)
    w = _rand((Cout, Cin, k), seed, 1.0 / math.sqrt(Cin * k))
    b = _rand((Cout,), seed + 1)
    x = _rand((R, Cin, T), seed + 2)
    pad = dil * (k // 2)
    taps, df, dt = pack.conv1d_taps(q16(w), dil, pad)
    spec = pack.make_conv_spec(taps, b, Cin, 0, df, dt, dev)
    xcl = x.permute(0, 2, 1).contiguous().half().view(1, R, T, Cin).to(dev)
    y = ops.conv(spec, xcl, None, 1, R, R, T)
    ref = F.conv1d(q16(x), q16(w), b, dilation=dil, padding=pad)
    assert rel_l2(y.cpu().float()[0].permute(0, 2, 1), ref) < TOL16


def case_convtr(lib, dev, Cin, Cout, K, stride, Fin, T, trim=True, f32_affine=False, B=2, seed=30):
    ops = Ops(lib)
    w = _rand((Cin, Cout, K, 1), seed, 1.0 / math.sqrt(Cin * K / stride))
    b = _rand((Cout,), seed + 1)
    x = _rand((B, Cin, Fin, T), seed + 2)
    taps, df, dt = pack.convtr_taps(q16(w), stride)
    spec = pack.make_conv_spec(taps, b, Cin, 0, df, dt, dev, transposed=1, fstride=stride)
    Fu = (Fin - 1) * stride + K
    pad = (K - stride) // 2 if trim else 0
    ref = F.conv_transpose2d(q16(x), q16(w), b, stride=(stride, 1))
    if pad:
        ref = ref[:, :, pad:-pad]
    kw = {}
    if f32_affine:
        sc, sh = _rand((B,), seed + 3).abs() + 0.5, _rand((B,), seed + 4)
        kw = dict(dst_f32=True, batch_scale=sc.to(dev), batch_shift=sh.to(dev))
        ref = ref * sc.view(B, 1, 1, 1) + sh.view(B, 1, 1, 1)
    y = ops.conv(spec, cl(x).to(dev), None, B, Fin, Fu, T, dst_f_off=pad, dst_F=Fu - 2 * pad, **kw)
    assert y.shape == (B, Fu - 2 * pad, T, Cout)
    if Cin % 32 == 0 and Cin <= 128 and K == 2 * stride and stride * Cout <= 8:      # the carried-tap kernel is the one that ran
        assert 'carry' in ops.lib.cdll.aero_last_kernel_name().decode()
    assert rel_l2(uncl(y.cpu()), ref) < TOL16


def case_conv_tail(lib, dev, Fin, T, B=2, seed=36):
    """the fused last decoder layer (aero_hip.h, aero_conv_desc.tail_w; aero.py:189-215, 497-498): 3x3 rewrite conv over two 48-channel
    sources -> GLU -> ConvTranspose2d(96 -> 2, [8,1] / [4,1]) -> trim -> x * std + mean, as ONE conv launch (the 96-channel activation is
    never stored) + aero_convtr_tail_finish, against (i) torch on fp16-rounded operands with the activation rounded to fp16 where the
    unfused path stores it and (ii) the unfused pair of launches."""
    ops = Ops(lib)
    C, M, K, stride, pad = 48, 192, 8, 4, 2
    w = _rand((M, 2 * C, 3, 3), seed, 1.0 / math.sqrt(2 * C * 9))
    b = _rand((M,), seed + 1)
    wt = _rand((M // 2, 2, K, 1), seed + 2, 1.0 / math.sqrt(M // 2 * K / stride))
    bt = _rand((2,), seed + 3)
    x = _rand((B, 2 * C, Fin, T), seed + 4)
    sc, sh = _rand((B,), seed + 5).abs() + 0.5, _rand((B,), seed + 6)
    taps, df, dt = pack.conv2d_taps(q16(w), 1, 1)
    spec = pack.make_conv_spec(taps, b, C, C, df, dt, dev, act=_lib.ACT_GLU)
    assert spec.tiled_bm == 192, spec.tiled_bm
    timg = pack.convtr_tail_image(q16(wt), stride, dev)
    xc = cl(x).to(dev)
    s0, s1 = xc[..., :C].contiguous(), xc[..., C:].contiguous()
    lo, hi = ops.conv(spec, s0, s1, B, Fin, Fin, T, tail=timg)
    kname = ops.lib.cdll.aero_last_kernel_name().decode()
    assert 'aero_conv_ring_kernel' in kname and ('<2, 2, 3, 3' in kname or '<2, 4, 3, 3' in kname or lib.is_emulator), kname
    y = ops.convtr_tail_finish(lo, hi, bt.to(dev), sc.to(dev), sh.to(dev), 4 * Fin, pad, M // 2)
    # round 6: the same rows at a cache-line pitch (the layout the iSTFT kernel asks for): identical values in columns t_off .. t_off + T
    Tt = lo.shape[2]
    yp = ops.convtr_tail_finish(lo, hi, bt.to(dev), sc.to(dev), sh.to(dev), 4 * Fin, pad, M // 2, pitched=((Tt + 12 + 15) // 16 * 16, 12))
    assert yp.shape[2] % 16 == 0 and yp._aero_pitched == (Tt, 12) and torch.equal(yp[:, :, 12:12 + Tt].cpu(), y.cpu())
    assert y.shape == (B, 4 * Fin, T, 2)
    # reference
    r = F.glu(F.conv2d(q16(x), q16(w), b, padding=1), dim=1)
    ref = F.conv_transpose2d(q16(r), q16(wt), bt, stride=(stride, 1))[:, :, pad:-pad] * sc.view(B, 1, 1, 1) + sh.view(B, 1, 1, 1)
    assert rel_l2(uncl(y.cpu()), ref) < TOL16
    # the unfused pair: same arithmetic up to the summation order of the two taps of an output row
    yy = ops.conv(spec, s0, s1, B, Fin, Fin, T)
    tt, dft, dtt = pack.convtr_taps(q16(wt), stride)
    tspec = pack.make_conv_spec(tt, bt, M // 2, 0, dft, dtt, dev, transposed=1, fstride=stride)
    Fu = (Fin - 1) * stride + K
    y2 = ops.conv(tspec, yy, None, B, Fin, Fu, T, dst_f_off=pad, dst_F=Fu - 2 * pad, dst_f32=True, batch_scale=sc.to(dev), batch_shift=sh.to(dev))
    assert rel_l2(y.cpu(), y2.cpu()) < 2e-6, rel_l2(y.cpu(), y2.cpu())
    # (a descriptor the fused form cannot take fails loudly instead of dropping the tail: 128 rows are not the 192-row tile)
    from aero_amd._lib import AeroHipError
    taps2, df2, dt2 = pack.conv2d_taps(q16(w[:128]), 1, 1)
    bad = pack.make_conv_spec(taps2, b[:128], C, C, df2, dt2, dev, act=_lib.ACT_GLU)
    try:
        ops.conv(bad, s0, s1, B, Fin, Fin, T, tail=timg)
    except AeroHipError as e:
        assert 'fused tail' in str(e)
    else:
        raise AssertionError('a 128-row conv accepted a fused tail')


def case_convtr_stacked(lib, dev, Cin, Cout, K, stride, Fin, T, trim=True, act='none', B=2, seed=33):
    """ConvTranspose2d computed from the input side (stacked residue classes + row scatter, aero_hip.h) against
    F.conv_transpose2d, with and without the frequency trim."""
    from aero_amd.engine import HipEngine
    w = _rand((Cin, Cout, K, 1), seed, 1.0 / math.sqrt(Cin * K / stride))
    b = _rand((Cout,), seed + 1)
    x = _rand((B, Cin, Fin, T), seed + 2)
    actc = {'none': _lib.ACT_NONE, 'gelu': _lib.ACT_GELU}[act]
    spec = pack.convtr_stacked_spec(q16(w), b, stride, dev, act=actc)
    assert spec is not None
    ref = F.conv_transpose2d(q16(x), q16(w), b, stride=(stride, 1))
    if act == 'gelu':
        ref = F.gelu(ref)
    Fu = ref.shape[2]
    pad = (K - stride) // 2 if trim else 0
    if pad:
        ref = ref[:, :, pad:Fu - pad]
    eng = HipEngine.__new__(HipEngine)
    eng.lib, eng.ops = lib, Ops(lib)
    y = eng._convtr_stacked(spec, cl(x).to(dev), B, Fin, T, stride, pad, Fu - 2 * pad)
    assert rel_l2(uncl(y.cpu()), ref) < TOL16


def case_freq_emb_epilogue(lib, dev, seed=40):
    """1x1 rewrite + fused GLU + frequency-embedding post-add (aero.py:133,475-480)."""
    ops = Ops(lib)
    B, Cc, Fq, T = 2, 16, 5, 37
    w = _rand((2 * Cc, Cc, 1, 1), seed, 0.25)
    b = _rand((2 * Cc,), seed + 1)
    x = _rand((B, Cc, Fq, T), seed + 2)
    emb = _rand((Fq, Cc), seed + 3)
    taps, df, dt = pack.conv2d_taps(q16(w), 0, 0)
    spec = pack.make_conv_spec(taps, b, Cc, 0, df, dt, dev, act=_lib.ACT_GLU)
    y = ops.conv(spec, cl(x).to(dev), None, B, Fq, Fq, T, post_add=emb.to(dev))
    ref = F.glu(F.conv2d(q16(x), q16(w), b), 1) + emb.t()[None, :, :, None]
    assert rel_l2(uncl(y.cpu()), ref) < TOL16


# ------------------------------------------------------------------------------------------------
def case_groupnorm(lib, dev, Cc, G, Fq, T, act, per_row=False, B=2, trim=0, seed=50):
    ops = Ops(lib)
    x = _rand((B, Cc, Fq, T), seed) * 2 + 0.3
    gamma, beta = _rand((Cc,), seed + 1) + 1, _rand((Cc,), seed + 2)
    xq = q16(x)
    if per_row:
        rows = xq.permute(0, 2, 1, 3).reshape(B * Fq, Cc, T)
        ref = F.group_norm(rows, G, gamma, beta, 1e-5).view(B, Fq, Cc, T).permute(0, 2, 1, 3)
    else:
        ref = F.group_norm(xq, G, gamma, beta, 1e-5)
    kw = {}
    if act == 'gelu':
        ref, a = F.gelu(ref), _lib.ACT_GELU
    elif act == 'glu':
        ref, a = F.glu(ref, 1), _lib.ACT_GLU
    elif act == 'glu_ls_res':
        ls = _rand((Cc // 2,), seed + 3)
        r = _rand((B, Cc // 2, Fq, T), seed + 4)
        ref, a = q16(r) + ls.view(1, -1, 1, 1) * F.glu(ref, 1), _lib.ACT_GLU
        kw = dict(layer_scale=ls.to(dev), res=cl(r).to(dev))
    elif act == 'snake':
        sa = torch.rand(Fq, generator=_g(seed + 5)) * 3 + 0.2
        ref, a = O.snake(ref.permute(0, 1, 3, 2), sa).permute(0, 1, 3, 2), _lib.ACT_SNAKE
        kw = dict(snake_a=sa.to(dev))
    else:
        a = _lib.ACT_NONE
    if trim:
        ref = ref[:, :, trim:-trim]
        kw.update(f_lo=trim, f_cnt=Fq - 2 * trim)
    y = ops.norm_act(cl(x).to(dev), G, per_row, gamma.to(dev), beta.to(dev), a, **kw)
    assert rel_l2(uncl(y.cpu()), ref) < TOL16


# ------------------------------------------------------------------------------------------------
def _lstm_sd(H, seed):
    sd = {}
    k = 1.0 / math.sqrt(H)
    i = 0
    for l in range(2):
        for sfx in ('', '_reverse'):
            inp = H if l == 0 else 2 * H
            for name, shape in (('weight_ih', (4 * H, inp)), ('weight_hh', (4 * H, H)), ('bias_ih', (4 * H,)),
                                ('bias_hh', (4 * H,))):
                sd[f'b.lstm.{name}_l{l}{sfx}'] = (torch.rand(*shape, generator=_g(seed + i)) * 2 - 1) * k
                i += 1
    sd['b.linear.weight'] = (torch.rand(H, 2 * H, generator=_g(seed + 90)) * 2 - 1) * k
    sd['b.linear.bias'] = (torch.rand(H, generator=_g(seed + 91)) * 2 - 1) * k
    return sd


def case_blstm(lib, dev, H, R, T, seed=60, fuse=True):
    """Whole BLSTM block (input projections fused in the recurrent kernel or as separate convs, 2 recurrent layers,
    linear + skip) against the oracle."""
    from aero_amd.engine import HipEngine

    class _DC:
        hidden = H
    sd = {k: q16(v) if 'weight' in k else v for k, v in _lstm_sd(H, seed).items()}
    x = _rand((R, H, T), seed + 100)
    ref = O.blstm(sd, 'b', q16(x))
    eng = HipEngine.__new__(HipEngine)
    eng.lib, eng.ops = lib, Ops(lib)
    eng.fuse_lstm_proj = fuse
    L = {'lstm': [pack.pack_lstm_layer(lib, sd, 'b.lstm', l, H, dev) for l in range(2)]}
    w = sd['b.linear.weight']
    L['lstm_lin'] = pack.make_conv_spec(w[None, :, None, :], sd['b.linear.bias'], 2 * H, 0, [0], [0], dev)
    h = x.permute(0, 2, 1).contiguous().half().view(1, R, T, H).to(dev)
    y = eng._blstm(_DC, L, h, 1, R, T)
    err = rel_l2(y.cpu().float()[0].permute(0, 2, 1), ref)
    assert err < BLSTM_TOL, err
    return err


def case_blstm_frame_skip(lib, dev, H=8, R=3, seed=62):
    """engine.blstm_frames drops a last frame whose outputs the stitch discards whole: the BLSTM block with it equals, BIT FOR BIT, the
    block with the reference's ceil(T / 100) frames -- for lengths where a frame is dropped (T mod 100 in 1..50) and where none is"""
    from aero_amd import engine as E
    from aero_amd.engine import HipEngine

    class _DC:
        hidden = H
    sd = {k: q16(v) if 'weight' in k else v for k, v in _lstm_sd(H, seed).items()}
    eng = HipEngine.__new__(HipEngine)
    eng.lib, eng.ops, eng.fuse_lstm_proj = lib, Ops(lib), True
    L = {'lstm': [pack.pack_lstm_layer(lib, sd, 'b.lstm', l, H, dev) for l in range(2)]}
    w = sd['b.linear.weight']
    L['lstm_lin'] = pack.make_conv_spec(w[None, :, None, :], sd['b.linear.bias'], 2 * H, 0, [0], [0], dev)
    dropped = 0
    for T in (201, 230, 250, 251, 301, 350, 376, 501):
        x = _rand((R, H, T), seed + T)
        h = x.permute(0, 2, 1).contiguous().half().view(1, R, T, H).to(dev)
        n_ref = math.ceil(T / 100)
        n_eff = E.blstm_frames(T)
        assert n_eff in (n_ref, n_ref - 1) and (n_eff == n_ref - 1) == (T <= (n_ref - 1) * 100 + 50)
        y = eng._blstm(_DC, L, h, 1, R, T)
        saved = E.blstm_frames
        E.blstm_frames = lambda T_, W=200, S=100: math.ceil(T_ / S)
        try:
            y_ref = eng._blstm(_DC, L, h, 1, R, T)
        finally:
            E.blstm_frames = saved
        assert torch.equal(y.cpu(), y_ref.cpu()), T
        # ... and the frame-major sequence order with the stitching layer stopping at its frames' last kept step (aero_lstm_desc.frame_major)
        # against the reference's frame-minor order with every step run; the projections fused and unfused
        for fuse in (True, False):
            eng.fuse_lstm_proj = fuse
            eng.lstm_frame_major = True
            y1 = eng._blstm(_DC, L, h, 1, R, T)
            eng.lstm_frame_major = False
            y0 = eng._blstm(_DC, L, h, 1, R, T)
            assert torch.equal(y1.cpu(), y0.cpu()), (T, fuse)
        eng.fuse_lstm_proj, eng.lstm_frame_major = True, True
        dropped += n_ref - n_eff
        assert rel_l2(y.cpu().float()[0].permute(0, 2, 1), O.blstm(sd, 'b', q16(x))) < BLSTM_TOL
    return dropped


def case_lstm_bitwise(lib, dev, H, R, T=501, seed=55):
    """the recurrent kernel alone (two stacked bidirectional layers with framing, as BLSTM runs them): the same launch twice gives the same
    bits, and a permutation of the rows gives the permuted result bit for bit -- a sequence's arithmetic must not depend on the lane or
    block it lands in.  (Round 5: an unrolled step loop that hipcc compiled with an MFMA -> VALU read hazard on one branch path returned
    run-to-run different hidden states; every tolerance-based model test still passed because LayerScale (1e-3) hides the branch.)"""
    ops = Ops(lib)
    W, S = 200, 100
    nframes = max(1, -(-T // S)) if T > W else 1
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / math.sqrt(H)
    sd = {}
    for l in range(2):
        for sfx in ('', '_reverse'):
            inp = H if l == 0 else 2 * H
            sd[f'l.weight_ih_l{l}{sfx}'] = (torch.rand(4 * H, inp, generator=g) * 2 - 1) * k
            sd[f'l.weight_hh_l{l}{sfx}'] = (torch.rand(4 * H, H, generator=g) * 2 - 1) * k
            sd[f'l.bias_ih_l{l}{sfx}'] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
            sd[f'l.bias_hh_l{l}{sfx}'] = (torch.rand(4 * H, generator=g) * 2 - 1) * k
    packs = [pack.pack_lstm_layer(lib, sd, 'l', l, H, dev) for l in range(2)]
    framed = T > W
    Wk = W if framed else T
    nseq = R * nframes

    def run(x):
        out0 = torch.zeros(nseq, Wk, 2 * H, device=dev, dtype=torch.float16)
        out1 = torch.zeros(R, T, 2 * H, device=dev, dtype=torch.float16)
        ops.lstm(None, None, packs[0][2], H, nseq, Wk, int(framed), 0, nframes, S, T, out0, x=x, fused=packs[0][3])
        ops.lstm(None, None, packs[1][2], H, nseq, Wk, 0, int(framed), nframes, S, T, out1, x=out0, fused=packs[1][3])
        return out0.clone(), out1.clone()
    x = torch.randn(R, T, H, generator=g).half().to(dev)
    a0, a1 = run(x)
    b0, b1 = run(x)
    assert torch.equal(a0, b0) and torch.equal(a1, b1), 'the LSTM kernel is not reproducible run to run'
    perm = torch.randperm(R, generator=g).to(dev)
    p0, p1 = run(x[perm].contiguous())
    assert torch.equal(p1, a1[perm]), float((p1.float() - a1[perm].float()).abs().max())
    assert torch.equal(p0.view(R, nframes, Wk, 2 * H), a0.view(R, nframes, Wk, 2 * H)[perm])
    assert float(a1.float().abs().max()) > 1e-2 and torch.isfinite(a1.float()).all()


def case_localstate(lib, dev, Cc, heads, R, T, seed=70):
    ops = Ops(lib)
    nd = 4
    sd = {}
    for i, (n, m) in enumerate((('query', Cc), ('key', Cc), ('content', Cc), ('query_decay', heads * nd), ('proj', Cc))):
        sd[f'a.{n}.weight'] = q16(_rand((m, Cc, 1), seed + i, 1.0 / math.sqrt(Cc)))
        sd[f'a.{n}.bias'] = _rand((m,), seed + 10 + i, 0.1)
    sd['a.query_decay.bias'] = sd['a.query_decay.bias'] - 1.0
    x = _rand((R, Cc, T), seed + 20)
    ref = O.local_state(sd, 'a', q16(x), heads, nd)
    w = torch.cat([sd[f'a.{n}.weight'][:, :, 0] for n in ('query', 'key', 'content', 'query_decay')], 0)
    b = torch.cat([sd[f'a.{n}.bias'] for n in ('query', 'key', 'content', 'query_decay')], 0)
    qk = pack.make_conv_spec(w[None, :, None, :], b, Cc, 0, [0], [0], dev)
    pj = pack.make_conv_spec(sd['a.proj.weight'][:, :, 0][None, :, None, :], sd['a.proj.bias'], Cc, 0, [0], [0], dev)
    h = x.permute(0, 2, 1).contiguous().half().view(1, R, T, Cc).to(dev)
    qkvd = ops.conv(qk, h, None, 1, R, R, T)
    att = ops.localstate(qkvd, R, T, Cc, heads, nd)
    y = ops.conv(pj, att.view(1, R, T, Cc), None, 1, R, R, T, res=h)
    err = rel_l2(y.cpu().float()[0].permute(0, 2, 1), ref)
    assert err < ATTN_TOL, err
    return err


def case_freqfc(lib, dev, Fq, Cc, T, B=2, seed=80):
    ops = Ops(lib)
    w = q16(_rand((Fq, Fq), seed, 1.0 / math.sqrt(Fq)))
    x = _rand((B, Cc, Fq, T), seed + 1)
    gate = _rand((B, Cc, T), seed + 2).abs()
    img = torch.zeros(pack._round_up(Fq, 128), pack._round_up(Fq, 32))
    img[:Fq, :Fq] = w
    y = ops.freqfc(cl(x).to(dev), img.half().to(dev), gate.permute(0, 2, 1).contiguous().half().to(dev))
    att = q16(gate)[:, :, None, :] * q16(x)
    ref = (att.transpose(2, 3) @ w.t()).transpose(2, 3)
    assert rel_l2(uncl(y.cpu()), ref) < TOL16


def case_enc0(lib, dev, Cc, M, K, stride, pad, Fq, T, B=2, act='gelu', seed=90):
    """aero_enc0_fwd: encoder 0's collapsed FTB (six-term bilinear form of the 2-channel spectrogram, k_enc0.h) fused with the
    layer's strided frequency conv + activation, against the same two steps in fp32 torch (x0 rounded to fp16 in between,
    as the kernel feeds the MFMA)."""
    ops = Ops(lib)
    xn, u = q16(_rand((B, Fq, T, 2), seed)), q16(_rand((B, Fq, T, 2), seed + 1))
    g = q16(_rand((B, T, 3, Cc), seed + 2, 0.5))
    rs, a_re, a_im, bf = (_rand((n,), seed + 3 + i) for i, n in enumerate((Fq, Cc, Cc, Cc)))
    w = _rand((M, Cc, K, 1), seed + 8, 1.0 / math.sqrt(Cc * K))
    b = _rand((M,), seed + 9)
    taps, df, dt = pack.conv2d_taps(q16(w), pad, 0)
    actc = {'none': _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'gelu': _lib.ACT_GELU}[act]
    spec = pack.make_conv_spec(taps, b, Cc, 0, df, dt, dev, fstride=stride, act=actc)
    Fo = (Fq + 2 * pad - K) // stride + 1
    P = dict(C=Cc, rs=rs.to(dev), a_re=a_re.to(dev), a_im=a_im.to(dev), bias=bf.to(dev))
    y = ops.enc0(xn.half().to(dev), u.half().to(dev), g.half().to(dev).view(B, 1, T, 3 * Cc), P, spec, Fo, stride, pad, actc)
    G = g[:, None]                                                        # [B,1,T,3,C]
    x0 = (u[..., 0:1] * G[..., 0, :] + u[..., 1:2] * G[..., 1, :] + rs[None, :, None, None] * G[..., 2, :]
          + xn[..., 0:1] * a_re + xn[..., 1:2] * a_im + bf).relu()        # [B,F,T,C]
    ref = F.conv2d(q16(x0).permute(0, 3, 1, 2), q16(w), b, stride=(stride, 1), padding=(pad, 0))
    ref = {'none': lambda v: v, 'relu': F.relu, 'gelu': F.gelu}[act](ref)
    assert y.shape == (B, Fo, T, M)
    assert rel_l2(uncl(y.cpu()), ref) < TOL16


def case_dconv_row(lib, dev, Cc, T, Fq=3, B=2, depth=2, act='gelu', norm=True, seed=100):
    """aero_dconv_row_fwd: a whole DConv branch (modules.py:221-249 without BLSTM / LocalState) on [B,F,T,C] rows against the
    same layers in fp32 torch (conv1d dilated -> GroupNorm(1) -> act -> conv1d 1x1 -> GroupNorm(1) -> GLU -> LayerScale -> +x)."""
    ops = Ops(lib)
    hid = Cc // 4
    x = q16(_rand((B, Fq, T, Cc), seed))
    actc = {'relu': _lib.ACT_RELU, 'gelu': _lib.ACT_GELU, 'snake': _lib.ACT_SNAKE}[act]
    layers, ref = [], x.reshape(B * Fq, T, Cc).permute(0, 2, 1)                     # [R, C, T]
    for l in range(depth):
        dil = 2 ** l
        w1 = q16(_rand((hid, Cc, 3), seed + 10 * l + 1, 1.0 / math.sqrt(3 * Cc)))
        b1 = _rand((hid,), seed + 10 * l + 2, 0.3)
        w2 = q16(_rand((2 * Cc, hid, 1), seed + 10 * l + 3, 1.0 / math.sqrt(hid)))
        b2 = _rand((2 * Cc,), seed + 10 * l + 4, 0.3)
        g1, be1 = 1 + _rand((hid,), seed + 10 * l + 5, 0.2), _rand((hid,), seed + 10 * l + 6, 0.2)
        g2, be2 = 1 + _rand((2 * Cc,), seed + 10 * l + 7, 0.2), _rand((2 * Cc,), seed + 10 * l + 8, 0.2)
        scale = _rand((Cc,), seed + 10 * l + 9, 0.5)
        sa = 0.5 + torch.rand(Fq, generator=_g(seed + 10 * l + 10)) * 2
        n = (lambda v: v) if norm else (lambda v: None)                               # noqa: E731
        Lr = pack.dconv_row_layer(w1, b1, n(g1), n(be1), w2[:, :, 0], b2, n(g2), n(be2), scale, dil, dev)
        Lr['snake_a'] = sa.float().to(dev).contiguous() if act == 'snake' else None
        layers.append(Lr)
        h = F.conv1d(q16(ref), w1, b1, dilation=dil, padding=dil)
        if norm:
            h = F.group_norm(h, 1, g1, be1, 1e-5)
        if act == 'snake':
            a = sa.repeat(B)[:, None, None]
            h = h + torch.sin(a * h) ** 2 / a
        else:
            h = {'relu': F.relu, 'gelu': F.gelu}[act](h)
        v = F.conv1d(h, w2, b2)
        if norm:
            v = F.group_norm(v, 1, g2, be2, 1e-5)
        ref = ref + scale[None, :, None] * F.glu(v, 1)
    assert ops.dconv_row_fits(T, Cc, hid, 2 ** (depth - 1))
    y = ops.dconv_row(x.half().to(dev), layers, actc, Fq)
    want = ref.permute(0, 2, 1).reshape(B, Fq, T, Cc)
    assert y.shape == x.shape
    err = rel_l2(y.float().cpu(), want)
    assert err < TOL16, err


def case_conv1d_split(lib, dev, Cin, Cout, k, R, T, S=3, seed=25):
    """tap split (aero_conv_desc.tap_split): the same Conv1d with its time taps in S groups summed by aero_split_finish equals the
    one-launch form up to fp32 summation order, and fp32 torch within the usual tolerance; two runs are bit-identical."""
    ops = Ops(lib)
    w = _rand((Cout, Cin, k), seed, 1.0 / math.sqrt(Cin * k))
    b = _rand((Cout,), seed + 1)
    x = _rand((R, Cin, T), seed + 2)
    taps, df, dt = pack.conv1d_taps(q16(w), 1, k // 2)
    spec = pack.make_conv_spec(taps, b, Cin, 0, df, dt, dev, act=_lib.ACT_RELU)
    xc = x.permute(0, 2, 1).contiguous().half().view(1, R, T, Cin).to(dev)
    ref = F.relu(F.conv1d(q16(x), q16(w), b, padding=k // 2))
    y1 = ops.conv(spec, xc, None, 1, R, R, T, tap_split=S)
    y2 = ops.conv(spec, xc, None, 1, R, R, T, tap_split=S)
    y0 = ops.conv(spec, xc, None, 1, R, R, T)
    assert torch.equal(y1, y2)
    assert rel_l2(y1.float().cpu()[0].permute(0, 2, 1), ref) < TOL16
    assert rel_l2(y1.float().cpu(), y0.float().cpu()) < 1e-3


# ---------------------------------------------------------------------------------------------------------------
# backward (aero_amd/backward.py): data gradients on the forward conv kernels, checked against torch.autograd of the
# same op in fp32 on fp16-rounded operands
def _autograd_dx(fn, x, dy):
    xr = q16(x).clone().requires_grad_(True)
    y = fn(xr)
    assert y.shape == dy.shape, (y.shape, dy.shape)
    return torch.autograd.grad(y, xr, q16(dy))[0]


def case_dgrad_conv2d(lib, dev, Cin, Cout, kF, kT, Fr, T, B=2, seed=70):
    from aero_amd import backward as bw
    ops = Ops(lib)
    w = _rand((Cout, Cin, kF, kT), seed, 1.0 / math.sqrt(Cout * kF * kT))
    x = _rand((B, Cin, Fr, T), seed + 1)
    dy = _rand((B, Cout, Fr, T), seed + 2)
    ref = _autograd_dx(lambda v: F.conv2d(v, q16(w), None, padding=(kF // 2, kT // 2)), x, dy)
    spec = bw.dgrad_conv2d(q16(w), kF // 2, kT // 2, dev)
    dx = ops.conv(spec, cl(dy).to(dev), None, B, Fr, Fr, T)
    assert rel_l2(uncl(dx.cpu()), ref) < TOL16


def case_dgrad_conv1d(lib, dev, Cin, Cout, k, dil, R, T, seed=73):
    from aero_amd import backward as bw
    ops = Ops(lib)
    w = _rand((Cout, Cin, k), seed, 1.0 / math.sqrt(Cout * k))
    x = _rand((R, Cin, T), seed + 1)
    dy = _rand((R, Cout, T), seed + 2)
    pad = dil * (k // 2)
    ref = _autograd_dx(lambda v: F.conv1d(v, q16(w), None, dilation=dil, padding=pad), x, dy)
    spec = bw.dgrad_conv1d(q16(w), dil, pad, dev)
    dycl = dy.permute(0, 2, 1).contiguous().half().view(1, R, T, Cout).to(dev)
    dx = ops.conv(spec, dycl, None, 1, R, R, T)
    assert rel_l2(dx.cpu().float()[0].permute(0, 2, 1), ref) < TOL16


def case_dgrad_conv_fstride(lib, dev, Cin, Cout, K, stride, Fin, T, B=2, seed=76):
    from aero_amd import backward as bw
    ops = Ops(lib)
    pad = (K - stride) // 2
    w = _rand((Cout, Cin, K, 1), seed, 1.0 / math.sqrt(Cout * K / stride))
    x = _rand((B, Cin, Fin, T), seed + 1)
    Fo = (Fin + 2 * pad - K) // stride + 1
    dy = _rand((B, Cout, Fo, T), seed + 2)
    ref = _autograd_dx(lambda v: F.conv2d(v, q16(w), None, stride=(stride, 1), padding=(pad, 0)), x, dy)
    spec = bw.dgrad_conv_fstride(q16(w), stride, dev)
    Fu = (Fo - 1) * stride + K
    dx = ops.conv(spec, cl(dy).to(dev), None, B, Fo, Fu, T, dst_f_off=pad, dst_F=Fin)
    assert rel_l2(uncl(dx.cpu()), ref) < TOL16


def case_dgrad_convtr(lib, dev, Cin, Cout, K, stride, Fin, T, B=2, seed=79):
    from aero_amd import backward as bw
    ops = Ops(lib)
    pad = (K - stride) // 2
    w = _rand((Cin, Cout, K, 1), seed, 1.0 / math.sqrt(Cout * K / stride))
    x = _rand((B, Cin, Fin, T), seed + 1)
    Fu = (Fin - 1) * stride + K
    Fy = Fu - 2 * pad
    dy = _rand((B, Cout, Fy, T), seed + 2)
    ref = _autograd_dx(lambda v: F.conv_transpose2d(v, q16(w), None, stride=(stride, 1))[:, :, pad:Fu - pad], x, dy)
    spec = bw.dgrad_convtr(q16(w), stride, pad, dev)
    dx = ops.conv(spec, cl(dy).to(dev), None, B, Fy, Fin, T)
    assert rel_l2(uncl(dx.cpu()), ref) < TOL16


def _autograd_dw(fn, w, b, dy):
    wr = q16(w).clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = fn(wr, br)
    assert y.shape == dy.shape, (y.shape, dy.shape)
    return torch.autograd.grad(y, [wr, br], q16(dy))


def case_wgrad_conv2d(lib, dev, Cin, Cout, kF, kT, Fr, T, B=2, seed=82):
    """weight / bias gradient of a stride-1 Conv2d (aero_conv_wgrad) against torch.autograd"""
    from aero_amd import backward as bw
    ops = Ops(lib)
    w = _rand((Cout, Cin, kF, kT), seed, 1.0 / math.sqrt(Cin * kF * kT))
    b = _rand((Cout,), seed + 3)
    x = _rand((B, Cin, Fr, T), seed + 1)
    dy = _rand((B, Cout, Fr, T), seed + 2)
    gw, gb = _autograd_dw(lambda ww, bb: F.conv2d(q16(x), ww, bb, padding=(kF // 2, kT // 2)), w, b, dy)
    _, df, dt = pack.conv2d_taps(w, kF // 2, kT // 2)
    import os
    os.environ['AERO_WGRAD_256'] = '2'                        # the 256 x 256 tile wherever it is legal (read per call)
    try:
        dwa, _ = bw.conv_wgrad(ops, cl(dy).to(dev), cl(x).to(dev), df, dt, nslab=0)        # chunks added with atomics
        if Cin >= 192 and Cout >= 192:
            assert 'wgrad256' in ops.lib.cdll.aero_last_kernel_name().decode()
        dw, db = bw.conv_wgrad(ops, cl(dy).to(dev), cl(x).to(dev), df, dt)                 # per-chunk slabs, added in order
        dw2, db2 = bw.conv_wgrad(ops, cl(dy).to(dev), cl(x).to(dev), df, dt)
    finally:
        del os.environ['AERO_WGRAD_256']
    assert torch.equal(dw, dw2) and torch.equal(db, db2)         # the slab form is deterministic (bias partials included)
    assert rel_l2(dwa.cpu(), dw.cpu()) < 1e-5
    got = dw.cpu().view(kF, kT, Cout, Cin).permute(2, 3, 0, 1)
    assert rel_l2(got, gw) < TOL16, rel_l2(got, gw)
    assert rel_l2(db.cpu(), gb) < TOL16
    if bw.wgrad_direct_ok(cl(dy), cl(x)):
        # straight into a destination in the nn.Conv2d weight's own layout (a column range of a wider weight, as the two sources of
        # a concatenated input use it) and into a bias destination: added in place, same sums bit for bit
        wfull = torch.full((Cout, Cin + 8, kF, kT), 0.5, device=dev)
        bfull = torch.full((Cout,), 0.25, device=dev)
        bw.conv_wgrad(ops, cl(dy).to(dev), cl(x).to(dev), df, dt, dw_out=wfull, db_out=bfull, layout=1, rowlen=Cin + 8, coff=8)
        dw3, db3 = bw.conv_wgrad(ops, cl(dy).to(dev), cl(x).to(dev), df, dt)           # (same tile choice as the call above)
        got3 = dw3.cpu().view(kF, kT, Cout, Cin).permute(2, 3, 0, 1)
        assert torch.equal(wfull[:, 8:].cpu(), got3 + 0.5) and bool((wfull[:, :8] == 0.5).all())
        assert torch.equal(bfull.cpu(), db3.cpu() + 0.25)


def case_wgrad_conv1d(lib, dev, Cin, Cout, k, dil, R, T, seed=85):
    from aero_amd import backward as bw
    ops = Ops(lib)
    w = _rand((Cout, Cin, k), seed, 1.0 / math.sqrt(Cin * k))
    b = _rand((Cout,), seed + 3)
    x = _rand((R, Cin, T), seed + 1)
    dy = _rand((R, Cout, T), seed + 2)
    pad = dil * (k // 2)
    gw, gb = _autograd_dw(lambda ww, bb: F.conv1d(q16(x), ww, bb, dilation=dil, padding=pad), w, b, dy)
    _, df, dt = pack.conv1d_taps(w, dil, pad)
    to_cl = lambda v: v.permute(0, 2, 1).contiguous().half().view(1, R, T, v.shape[1]).to(dev)   # noqa: E731
    dw, db = bw.conv_wgrad(ops, to_cl(dy), to_cl(x), df, dt)
    assert rel_l2(dw.cpu().permute(1, 2, 0), gw) < TOL16
    assert rel_l2(db.cpu(), gb) < TOL16


def case_wgrad_conv_fstride(lib, dev, Cin, Cout, K, stride, Fin, T, B=2, seed=88):
    from aero_amd import backward as bw
    ops = Ops(lib)
    pad = (K - stride) // 2
    w = _rand((Cout, Cin, K, 1), seed, 1.0 / math.sqrt(Cin * K))
    b = _rand((Cout,), seed + 3)
    x = _rand((B, Cin, Fin, T), seed + 1)
    Fo = (Fin + 2 * pad - K) // stride + 1
    dy = _rand((B, Cout, Fo, T), seed + 2)
    gw, gb = _autograd_dw(lambda ww, bb: F.conv2d(q16(x), ww, bb, stride=(stride, 1), padding=(pad, 0)), w, b, dy)
    _, df, dt = pack.conv2d_taps(w, pad, 0)
    dw, db = bw.conv_wgrad(ops, cl(dy).to(dev), cl(x).to(dev), df, dt, fstride=stride)
    assert rel_l2(dw.cpu().permute(1, 2, 0).unsqueeze(-1), gw) < TOL16
    assert rel_l2(db.cpu(), gb) < TOL16


def case_wgrad_convtr(lib, dev, Cin, Cout, K, stride, Fin, T, B=2, seed=91):
    """ConvTranspose2d weight gradient: the same kernel with the roles of x and dy swapped"""
    from aero_amd import backward as bw
    ops = Ops(lib)
    pad = (K - stride) // 2
    w = _rand((Cin, Cout, K, 1), seed, 1.0 / math.sqrt(Cin * K / stride))
    b = _rand((Cout,), seed + 3)
    x = _rand((B, Cin, Fin, T), seed + 1)
    Fu = (Fin - 1) * stride + K
    dy = _rand((B, Cout, Fu - 2 * pad, T), seed + 2)
    gw, gb = _autograd_dw(lambda ww, bb: F.conv_transpose2d(q16(x), ww, bb, stride=(stride, 1))[:, :, pad:Fu - pad], w, b, dy)
    # dw[ci, co, kk] = sum_q x[q][ci] * dy[q*s + kk - pad][co]
    dw, _ = bw.conv_wgrad(ops, cl(x).to(dev), cl(dy).to(dev), [kk - pad for kk in range(K)], [0] * K, fstride=stride, bias=False)
    assert rel_l2(dw.cpu().permute(1, 2, 0).unsqueeze(-1), gw) < TOL16


def case_norm_bwd(lib, dev, C_, G, per_row, act, Fr, T, B=2, layer_scale=False, seed=94):
    """GroupNorm + activation backward (aero_norm_bwd_*) against torch.autograd of the fp32 composition"""
    from aero_amd import backward as bw
    ops = Ops(lib)
    x = _rand((B, C_, Fr, T), seed, 1.5) + 0.3
    gamma, beta = _rand((C_,), seed + 1) * 0.5 + 1.0, _rand((C_,), seed + 2) * 0.3
    Cout = C_ // 2 if act == 'glu' else C_
    ls = (_rand((Cout,), seed + 4).abs() + 0.2) if layer_scale else None
    dy = _rand((B, Cout, Fr, T), seed + 3)
    xr = q16(x).clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    lr = ls.clone().requires_grad_(True) if ls is not None else None
    if per_row:
        u = F.group_norm(xr.permute(0, 2, 1, 3).reshape(B * Fr, C_, T), G, gr, br).view(B, Fr, C_, T).permute(0, 2, 1, 3)
    else:
        u = F.group_norm(xr, G, gr, br)
    y = {'none': lambda v: v, 'gelu': F.gelu, 'glu': lambda v: F.glu(v, dim=1)}[act](u)
    if lr is not None:
        y = y * lr.view(1, -1, 1, 1)
    grads = torch.autograd.grad(y, [xr, gr, br] + ([lr] if lr is not None else []), q16(dy))
    actc = {'none': _lib.ACT_NONE, 'gelu': _lib.ACT_GELU, 'glu': _lib.ACT_GLU}[act]
    xd = cl(x).to(dev)
    ops.norm_act(xd, G, per_row, gamma.to(dev), beta.to(dev), actc, layer_scale=None if ls is None else ls.to(dev))
    stats = ops._last_stats
    dx, dg, dbt, dls = bw.norm_bwd(ops, xd, cl(dy).to(dev), stats, G, per_row, gamma.to(dev), beta.to(dev), actc,
                                   layer_scale=None if ls is None else ls.to(dev))
    assert rel_l2(uncl(dx.cpu()), grads[0]) < 2 * TOL16, rel_l2(uncl(dx.cpu()), grads[0])
    assert rel_l2(dg.cpu(), grads[1]) < TOL16 and rel_l2(dbt.cpu(), grads[2]) < TOL16
    if lr is not None:
        assert rel_l2(dls.cpu(), grads[3]) < TOL16


def case_block_autograd(lib, dev, kind, Cin, Cout, G, act, Fin, T, B=2, seed=100):
    """conv -> GroupNorm -> GELU / GLU as ONE torch.autograd.Function on the HIP kernels (aero_amd/autograd.py): loss.backward()
    through it gives the same input / parameter gradients as the fp32 torch composition (HEncLayer / HDecLayer blocks)."""
    from aero_amd.autograd import ConvNormAct
    if kind[0] == 'conv2d':
        w = _rand((Cout, Cin, 2 * kind[1] + 1, 2 * kind[2] + 1), seed, 1.0 / math.sqrt(Cin * (2 * kind[1] + 1) * (2 * kind[2] + 1)))
        ref_conv = lambda v, ww, bb: F.conv2d(v, ww, bb, padding=(kind[1], kind[2]))                      # noqa: E731
    elif kind[0] == 'fstride':
        K, s = 2 * kind[1], kind[1]
        w = _rand((Cout, Cin, K, 1), seed, 1.0 / math.sqrt(Cin * K))
        ref_conv = lambda v, ww, bb: F.conv2d(v, ww, bb, stride=(s, 1), padding=((K - s) // 2, 0))        # noqa: E731
    else:
        K, s = 2 * kind[1], kind[1]
        w = _rand((Cin, Cout, K, 1), seed, 1.0 / math.sqrt(Cin * K / s))
        p_ = (K - s) // 2
        ref_conv = lambda v, ww, bb: F.conv_transpose2d(v, ww, bb, stride=(s, 1))[:, :, p_:-p_]         # noqa: E731
    w = q16(w)
    b = _rand((Cout,), seed + 1) * 0.1
    gamma, beta = _rand((Cout,), seed + 2) * 0.3 + 1.0, _rand((Cout,), seed + 3) * 0.2
    x = q16(_rand((B, Cin, Fin, T), seed + 4))
    actf = {'gelu': F.gelu, 'glu': lambda v: F.glu(v, dim=1)}[act]
    # reference: fp32 autograd
    pr = [t.clone().requires_grad_(True) for t in (x, w, b, gamma, beta)]
    yr = actf(F.group_norm(ref_conv(pr[0], pr[1], pr[2]), G, pr[3], pr[4]))
    gy = q16(_rand(tuple(yr.shape), seed + 5))
    (yr * gy).sum().backward()
    # device: the Function
    pd = [cl(x).to(dev).requires_grad_(True)] + [t.clone().to(dev).requires_grad_(True) for t in (w, b, gamma, beta)]
    yd = ConvNormAct.apply(pd[0], pd[1], pd[2], pd[3], pd[4], lib, kind, G, act)
    assert rel_l2(uncl(yd.detach().cpu()), yr.detach()) < TOL16
    (yd.float() * cl(gy).to(dev).float()).sum().backward()
    assert rel_l2(uncl(pd[0].grad.cpu()), pr[0].grad) < 3 * TOL16, rel_l2(uncl(pd[0].grad.cpu()), pr[0].grad)
    for got, ref, nm in zip(pd[1:], pr[1:], ('weight', 'bias', 'gamma', 'beta')):
        assert rel_l2(got.grad.cpu(), ref.grad) < 3 * TOL16, (nm, rel_l2(got.grad.cpu(), ref.grad))


def case_dconv_autograd(lib, dev, Cc, k, dil, Fr, T, B=2, compress=4, seed=110):
    """a residual DConv layer (modules.py:206-244, no LSTM / attention) as one Function on the HIP kernels vs fp32 torch autograd"""
    from aero_amd.autograd import DConvLayer
    H = Cc // compress
    w1 = q16(_rand((H, Cc, k), seed, 1.0 / math.sqrt(Cc * k)))
    w2 = q16(_rand((2 * Cc, H, 1), seed + 1, 1.0 / math.sqrt(H)))
    b1, b2 = _rand((H,), seed + 2) * 0.1, _rand((2 * Cc,), seed + 3) * 0.1
    g1, be1 = _rand((H,), seed + 4) * 0.3 + 1.0, _rand((H,), seed + 5) * 0.2
    g2, be2 = _rand((2 * Cc,), seed + 6) * 0.3 + 1.0, _rand((2 * Cc,), seed + 7) * 0.2
    scale = _rand((Cc,), seed + 8).abs() * 0.5 + 0.1
    x = q16(_rand((B, Cc, Fr, T), seed + 9))
    params = (w1, b1, g1, be1, w2, b2, g2, be2, scale)
    pr = [t.clone().requires_grad_(True) for t in (x,) + params]
    xr = pr[0].permute(0, 2, 1, 3).reshape(B * Fr, Cc, T)
    hr = F.gelu(F.group_norm(F.conv1d(xr, pr[1], pr[2], dilation=dil, padding=dil * (k // 2)), 1, pr[3], pr[4]))
    hr = F.glu(F.group_norm(F.conv1d(hr, pr[5], pr[6]), 1, pr[7], pr[8]), dim=1)
    yr = (xr + pr[9].view(1, -1, 1) * hr).view(B, Fr, Cc, T).permute(0, 2, 1, 3)
    gy = q16(_rand(tuple(yr.shape), seed + 10))
    (yr * gy).sum().backward()
    pd = [cl(x).to(dev).requires_grad_(True)] + [t.clone().to(dev).requires_grad_(True) for t in params]
    yd = DConvLayer.apply(*pd, lib, dil)
    assert rel_l2(uncl(yd.detach().cpu()), yr.detach()) < TOL16
    (yd.float() * cl(gy).to(dev).float()).sum().backward()
    assert rel_l2(uncl(pd[0].grad.cpu()), pr[0].grad) < 3 * TOL16, rel_l2(uncl(pd[0].grad.cpu()), pr[0].grad)
    for got, ref, nm in zip(pd[1:], pr[1:], ('w1', 'b1', 'g1', 'be1', 'w2', 'b2', 'g2', 'be2', 'scale')):
        assert rel_l2(got.grad.cpu(), ref.grad) < 3 * TOL16, (nm, rel_l2(got.grad.cpu(), ref.grad))


def case_train_steps(lib, dev, steps=4, B=2, Fin=16, T=48, seed=120):
    """f1 pieces together: a strided-conv block -> residual DConv layer -> 1x1 rewrite block (an encoder layer of aero.py:86-101
    without FTB / LSTM / attention), trained for a few steps -- forward and backward through aero_amd.autograd on the HIP kernels,
    fused FlatAdam step -- against the fp32 torch modules under torch.optim.Adam: same loss trajectory, same parameters."""
    from aero_amd.autograd import ConvNormAct, DConvLayer
    from aero_amd.optim import FlatAdam
    C0, C1 = 16, 32
    H = C1 // 4
    shapes = dict(w_a=(C1, C0, 8, 1), b_a=(C1,), g_a=(C1,), be_a=(C1,),
                  w1=(H, C1, 3), b1=(H,), g1=(H,), be1=(H,), w2=(2 * C1, H, 1), b2=(2 * C1,), g2=(2 * C1,), be2=(2 * C1,), sc=(C1,),
                  w_r=(2 * C1, C1, 1, 1), b_r=(2 * C1,), g_r=(2 * C1,), be_r=(2 * C1,))
    init = {}
    for i, (k, shp) in enumerate(shapes.items()):
        if k.startswith('g'):
            init[k] = _rand(shp, seed + i) * 0.1 + 1.0
        elif k == 'sc':
            init[k] = _rand(shp, seed + i).abs() * 0.3 + 0.2
        elif k.startswith('w'):
            init[k] = q16(_rand(shp, seed + i, 1.0 / math.sqrt(shp[1] * shp[2])))
        else:
            init[k] = _rand(shp, seed + i) * 0.1
    x = q16(_rand((B, C0, Fin, T), seed + 50))
    tgt = _rand((B, C1, Fin // 4, T), seed + 51)

    def ref_net(P, v):
        h = F.gelu(F.group_norm(F.conv2d(v, P['w_a'], P['b_a'], stride=(4, 1), padding=(2, 0)), 4, P['g_a'], P['be_a']))
        Bq, Cq, Fq, Tq = h.shape
        r = h.permute(0, 2, 1, 3).reshape(Bq * Fq, Cq, Tq)
        u = F.gelu(F.group_norm(F.conv1d(r, P['w1'], P['b1'], padding=1), 1, P['g1'], P['be1']))
        u = F.glu(F.group_norm(F.conv1d(u, P['w2'], P['b2']), 1, P['g2'], P['be2']), dim=1)
        h = (r + P['sc'].view(1, -1, 1) * u).view(Bq, Fq, Cq, Tq).permute(0, 2, 1, 3)
        return F.glu(F.group_norm(F.conv2d(h, P['w_r'], P['b_r']), 4, P['g_r'], P['be_r']), dim=1)

    def dev_net(P, v):
        h = ConvNormAct.apply(v, P['w_a'], P['b_a'], P['g_a'], P['be_a'], lib, ('fstride', 4), 4, 'gelu')
        h = DConvLayer.apply(h, P['w1'], P['b1'], P['g1'], P['be1'], P['w2'], P['b2'], P['g2'], P['be2'], P['sc'], lib, 1)
        return ConvNormAct.apply(h, P['w_r'], P['b_r'], P['g_r'], P['be_r'], lib, ('conv2d', 0, 0), 4, 'glu')

    Pr = {k: torch.nn.Parameter(v.clone()) for k, v in init.items()}
    Pd = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in init.items()}
    opt_r = torch.optim.Adam(list(Pr.values()), lr=2e-3, betas=(0.9, 0.999))
    opt_d = FlatAdam(list(Pd.values()), lr=2e-3, betas=(0.9, 0.999), lib=lib)
    xd, tgd = cl(x).to(dev), cl(tgt).to(dev).float()
    losses = []
    for it in range(steps):
        opt_r.zero_grad()
        lr_ = ((ref_net(Pr, x) - tgt) ** 2).mean()
        lr_.backward()
        opt_r.step()
        opt_d.zero_grad()
        ld = ((dev_net(Pd, xd).float() - tgd) ** 2).mean()          # (the loss itself is host-side code, as in the reference's solver)
        ld.backward()
        opt_d.step()
        losses.append((float(lr_.detach()), float(ld.detach())))
    for a, b_ in losses:
        assert abs(a - b_) <= 2e-2 * abs(a), losses
    assert losses[-1][0] < losses[0][0] and losses[-1][1] < losses[0][1], losses
    for k in init:
        moved = (Pr[k].detach() - init[k]).norm()
        err = (Pd[k].detach().cpu() - Pr[k].detach()).norm()
        assert err <= 0.15 * moved + 1e-6, (k, float(err), float(moved))       # the UPDATE agrees, not just the (mostly unchanged) value


def _snake_ref(u, a):
    """snake.py:67 with the per-frequency-row parameter: u [B, C, F, T], a [F]"""
    av = a.view(1, 1, -1, 1)
    return u + torch.sin(av * u) ** 2 / av


def case_block_autograd_snake(lib, dev, kind, Cin, Cout, G, Fin, T, B=2, seed=130):
    """the flagship config's blocks (act_func snake; G = 0: the layers before norm_starts have no GroupNorm): conv -> [GroupNorm] ->
    Snake through aero_amd.autograd.ConvNormAct, gradients incl. Snake's alpha against fp32 torch autograd"""
    from aero_amd.autograd import ConvNormAct
    if kind[0] == 'fstride':
        K, s_ = 2 * kind[1], kind[1]
        w = q16(_rand((Cout, Cin, K, 1), seed, 1.0 / math.sqrt(Cin * K)))
        ref_conv = lambda v, ww, bb: F.conv2d(v, ww, bb, stride=(s_, 1), padding=((K - s_) // 2, 0))      # noqa: E731
        Fo = Fin // s_
    else:
        K, s_ = 2 * kind[1], kind[1]
        w = q16(_rand((Cin, Cout, K, 1), seed, 1.0 / math.sqrt(Cin * K / s_)))
        p_ = (K - s_) // 2
        ref_conv = lambda v, ww, bb: F.conv_transpose2d(v, ww, bb, stride=(s_, 1))[:, :, p_:-p_]         # noqa: E731
        Fo = Fin * s_
    b = _rand((Cout,), seed + 1) * 0.1
    alpha = _rand((Fo,), seed + 6).abs() * 0.8 + 0.4
    x = q16(_rand((B, Cin, Fin, T), seed + 4))
    names = ['x', 'weight', 'bias']
    pr = [t.clone().requires_grad_(True) for t in (x, w, b)]
    ar = alpha.clone().requires_grad_(True)
    hr = ref_conv(pr[0], pr[1], pr[2])
    if G:
        gamma, beta = _rand((Cout,), seed + 2) * 0.3 + 1.0, _rand((Cout,), seed + 3) * 0.2
        pr += [gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)]
        names += ['gamma', 'beta']
        hr = F.group_norm(hr, G, pr[3], pr[4])
    yr = _snake_ref(hr, ar)
    gy = q16(_rand(tuple(yr.shape), seed + 5))
    (yr * gy).sum().backward()
    pd = [cl(x).to(dev).requires_grad_(True)] + [t.detach().clone().to(dev).requires_grad_(True) for t in pr[1:]]
    ad = alpha.clone().to(dev).requires_grad_(True)
    yd = ConvNormAct.apply(pd[0], pd[1], pd[2], pd[3] if G else None, pd[4] if G else None, lib, kind, G, 'snake', ad)
    assert rel_l2(uncl(yd.detach().cpu()), yr.detach()) < TOL16
    (yd.float() * cl(gy).to(dev).float()).sum().backward()
    assert rel_l2(uncl(pd[0].grad.cpu()), pr[0].grad) < 3 * TOL16, rel_l2(uncl(pd[0].grad.cpu()), pr[0].grad)
    for got, ref, nm in zip(pd[1:], pr[1:], names[1:]):
        assert rel_l2(got.grad.cpu(), ref.grad) < 3 * TOL16, (nm, rel_l2(got.grad.cpu(), ref.grad))
    assert rel_l2(ad.grad.cpu(), ar.grad) < 3 * TOL16, rel_l2(ad.grad.cpu(), ar.grad)


def case_decoder_autograd(lib, dev, C=64, Fin=4, T=40, B=2, seed=140):
    """two HDecLayers as the flagship config builds them (aero.py:148-216): x + skip -> 3x3 rewrite -> [GroupNorm] -> GLU -> ConvTranspose
    -> [GroupNorm] -> Snake, the last one without norm / activation and with 2 output channels -- every gradient (inputs, both skips,
    all parameters) through aero_amd.autograd against fp32 torch autograd."""
    from aero_amd.autograd import ConvNormAct
    P = dict(w_r1=q16(_rand((2 * C, C, 3, 3), seed, 1.0 / math.sqrt(C * 9))), b_r1=_rand((2 * C,), seed + 1) * 0.1,
             g_r1=_rand((2 * C,), seed + 2) * 0.3 + 1.0, be_r1=_rand((2 * C,), seed + 3) * 0.2,
             w_t1=q16(_rand((C, C // 2, 4, 1), seed + 4, 1.0 / math.sqrt(C * 2))), b_t1=_rand((C // 2,), seed + 5) * 0.1,
             g_t1=_rand((C // 2,), seed + 6) * 0.3 + 1.0, be_t1=_rand((C // 2,), seed + 7) * 0.2,
             al1=_rand((2 * Fin,), seed + 8).abs() * 0.8 + 0.4,
             w_r2=q16(_rand((C, C // 2, 3, 3), seed + 9, 1.0 / math.sqrt(C * 4.5))), b_r2=_rand((C,), seed + 10) * 0.1,
             w_t2=q16(_rand((C // 2, 2, 8, 1), seed + 11, 1.0 / math.sqrt(C))), b_t2=_rand((2,), seed + 12) * 0.1)
    x = q16(_rand((B, C, Fin, T), seed + 20))
    s1 = q16(_rand((B, C, Fin, T), seed + 21))
    s2 = q16(_rand((B, C // 2, 2 * Fin, T), seed + 22))

    def ref(P, x, s1, s2):
        h = F.glu(F.group_norm(F.conv2d(x + s1, P['w_r1'], P['b_r1'], padding=1), 4, P['g_r1'], P['be_r1']), dim=1)
        h = F.group_norm(F.conv_transpose2d(h, P['w_t1'], P['b_t1'], stride=(2, 1))[:, :, 1:-1], 4, P['g_t1'], P['be_t1'])
        h = _snake_ref(h, P['al1'])
        h = F.glu(F.conv2d(h + s2, P['w_r2'], P['b_r2'], padding=1), dim=1)
        return F.conv_transpose2d(h, P['w_t2'], P['b_t2'], stride=(4, 1))[:, :, 2:-2]

    def devnet(P, x, s1, s2):
        h = ConvNormAct.apply(x, P['w_r1'], P['b_r1'], P['g_r1'], P['be_r1'], lib, ('conv2d', 1, 1), 4, 'glu', None, s1)
        h = ConvNormAct.apply(h, P['w_t1'], P['b_t1'], P['g_t1'], P['be_t1'], lib, ('convtr', 2), 4, 'snake', P['al1'])
        h = ConvNormAct.apply(h, P['w_r2'], P['b_r2'], None, None, lib, ('conv2d', 1, 1), 0, 'glu', None, s2)
        return ConvNormAct.apply(h, P['w_t2'], P['b_t2'], None, None, lib, ('convtr', 4), 0, 'none')

    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ir = [t.clone().requires_grad_(True) for t in (x, s1, s2)]
    yr = ref(Pr, *ir)
    gy = q16(_rand(tuple(yr.shape), seed + 30))
    (yr * gy).sum().backward()
    Pd = {k: v.clone().to(dev).requires_grad_(True) for k, v in P.items()}
    idv = [cl(t).to(dev).requires_grad_(True) for t in (x, s1, s2)]
    yd = devnet(Pd, *idv)
    assert rel_l2(uncl(yd.detach().cpu()), yr.detach()) < 2 * TOL16
    (yd.float() * cl(gy).to(dev).float()).sum().backward()
    for got, ref_t, nm in zip(idv, ir, ('x', 'skip1', 'skip2')):
        assert rel_l2(uncl(got.grad.cpu()), ref_t.grad) < 4 * TOL16, (nm, rel_l2(uncl(got.grad.cpu()), ref_t.grad))
    for k in P:
        assert rel_l2(Pd[k].grad.cpu(), Pr[k].grad) < 4 * TOL16, (k, rel_l2(Pd[k].grad.cpu(), Pr[k].grad))


def case_batchnorm_bwd(lib, dev, C_, act, Fr, T, B=3, seed=150):
    """training-mode BatchNorm (+ ReLU) backward -- the FTB's norms (modules.py:287,293,300) -- through aero_norm_bwd_* with per_row 2"""
    from aero_amd import backward as bw
    ops = Ops(lib)
    x = _rand((B, C_, Fr, T), seed, 1.5) + 0.3
    gamma, beta = _rand((C_,), seed + 1) * 0.5 + 1.0, _rand((C_,), seed + 2) * 0.3
    dy = _rand((B, C_, Fr, T), seed + 3)
    xr = q16(x).clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    u = F.batch_norm(xr, None, None, gr, br, training=True)
    y = F.relu(u) if act == 'relu' else u
    grads = torch.autograd.grad(y, [xr, gr, br], q16(dy))
    actc = _lib.ACT_RELU if act == 'relu' else _lib.ACT_NONE
    xd = cl(x).to(dev)
    ops.norm_act(xd, C_, 2, gamma.to(dev), beta.to(dev), actc)
    stats = ops._last_stats
    dx, dg, dbt, _ = bw.norm_bwd(ops, xd, cl(dy).to(dev), stats, C_, 2, gamma.to(dev), beta.to(dev), actc)
    assert rel_l2(uncl(dx.cpu()), grads[0]) < 2 * TOL16, rel_l2(uncl(dx.cpu()), grads[0])
    assert rel_l2(dg.cpu(), grads[1]) < TOL16 and rel_l2(dbt.cpu(), grads[2]) < TOL16


def case_istft_bwd(lib, dev, nfft, hop, T, crop=5, B=2, seed=160):
    """iSTFT backward (backward.istft_bwd: prep -> the forward STFT kernel -> pack) against autograd through the oracle's istft"""
    from aero_amd import backward as bw
    ops = Ops(lib)
    z = torch.complex(_rand((B, nfft // 2, T), seed), _rand((B, nfft // 2, T), seed + 1)).requires_grad_(True)
    Lout = hop * (T - 1) - crop
    y = O.istft(F.pad(z, (0, 0, 0, 1)), hop, nfft)[..., :Lout]
    gy = _rand(tuple(y.shape), seed + 2)
    (y * gy).sum().backward()
    w = _hann_padded(nfft, nfft, 'cpu')
    env = torch.zeros(nfft + hop * (T - 1), dtype=torch.float64)
    for t in range(T):
        env[t * hop:t * hop + nfft] += (w * w).double()
    dz = bw.istft_bwd(ops, gy.reshape(B, Lout).contiguous().to(dev), nfft, hop, w.to(dev), (1 / env).float().to(dev), T)
    got = torch.view_as_complex(dz.cpu().contiguous())
    assert rel_l2(torch.view_as_real(got), torch.view_as_real(z.grad)) < 5 * TOL32, rel_l2(torch.view_as_real(got), torch.view_as_real(z.grad))


# ------------------------------------------------------------------------------------------------
# the rest of the training step (csrc/k_train.h) against torch.autograd on fp16-rounded operands
from aero_amd import train_ops as TO  # noqa: E402


def case_freqfc_wgrad(lib, dev, Fq, Cc, T, B=2, seed=200):
    ops = Ops(lib)
    w = q16(_rand((Fq, Fq), seed, 1.0 / math.sqrt(Fq))).requires_grad_()
    x = q16(_rand((B, Cc, Fq, T), seed + 1))
    gate = q16(_rand((B, Cc, T), seed + 2).abs())
    dy = q16(_rand((B, Cc, Fq, T), seed + 3))
    att = gate[:, :, None, :] * x
    ((att.transpose(2, 3) @ w.t()).transpose(2, 3) * dy).sum().backward()
    dw = TO.freqfc_wgrad(ops, cl(dy).to(dev), cl(x).to(dev), gate.permute(0, 2, 1).contiguous().half().to(dev))
    assert rel_l2(dw.cpu(), w.grad) < TOL16


def case_ftb_gate_bwd(lib, dev, Fq, Cc, T, B=2, seed=210):
    ops = Ops(lib)
    v, x, add = (q16(_rand((B, Fq, T, Cc), seed + i)) for i in range(3))
    gate = q16(_rand((B, T, Cc), seed + 4).abs())
    dx, dg = TO.ftb_gate_bwd(ops, v.half().to(dev), x.half().to(dev), gate.half().to(dev), add.half().to(dev))
    assert rel_l2(dx.cpu().float(), add + v * gate[:, None]) < TOL16
    assert rel_l2(dg.cpu().float(), (v * x).sum(1)) < TOL16
    dx2, _ = TO.ftb_gate_bwd(ops, v.half().to(dev), x.half().to(dev), gate.half().to(dev))
    assert rel_l2(dx2.cpu().float(), v * gate[:, None]) < TOL16


def case_sum_bt(lib, dev, Fq, Cc, T, B=3, seed=220):
    ops = Ops(lib)
    x = q16(_rand((B, Fq, T, Cc), seed))
    out = torch.ones(Fq, Cc, dtype=torch.float32, device=dev)
    TO.sum_bt(ops, x.half().to(dev), out, 0.5)
    assert rel_l2(out.cpu(), 1.0 + 0.5 * x.sum((0, 2))) < 1e-5


def case_frames_op(lib, dev, R, T, Cc, seed=230):
    """unfold / stitch against the oracle's index helpers (models/utils.py:22-35, modules.py:49-61) and their adjoints by <Ax, y> = <x, A^T y>"""
    ops = Ops(lib)
    W, S = 200, 100
    nf = math.ceil(T / S)
    x = q16(_rand((R, T, Cc), seed))
    fr = TO.frames_op(ops, x.half().to(dev), 0, R, T, Cc, nf, W, S).cpu().float()
    ref = torch.zeros(R, nf, W, Cc)
    for k in range(nf):
        n = min(W, T - k * S)
        ref[:, k, :n] = x[:, k * S:k * S + n]
    assert torch.equal(fr.view(R, nf, W, Cc), ref)
    y = q16(_rand((R * nf, W, Cc), seed + 1))
    st = TO.frames_op(ops, y.half().to(dev), 2, R, T, Cc, nf, W, S).cpu().float()
    smap = O.stitch_map(T, W, S, nf)                      # (frame, tau) per output step
    yv = y.view(R, nf, W, Cc)
    ref = torch.stack([yv[:, k, tau] for (k, tau) in smap], 1)
    assert torch.equal(st, ref)
    # adjoints (outputs are fp16-rounded sums of at most two terms)
    g = q16(_rand((R, T, Cc), seed + 2))
    ut = TO.frames_op(ops, y.half().to(dev), 1, R, T, Cc, nf, W, S).cpu().float()
    assert abs(float((ut.double() * g.double()).sum() - (y.double() * fr_of(g, R, T, Cc, nf, W, S).double()).sum())) < 2e-3 * float(ut.abs().sum())
    stt = TO.frames_op(ops, g.half().to(dev), 3, R, T, Cc, nf, W, S).cpu().float().view(R, nf, W, Cc)
    ref = torch.zeros(R, nf, W, Cc)
    for t, (k, tau) in enumerate(smap):
        ref[:, k, tau] = g[:, t]
    assert torch.equal(stt, ref)


def fr_of(x, R, T, Cc, nf, W, S):
    out = torch.zeros(R * nf, W, Cc)
    o = out.view(R, nf, W, Cc)
    for k in range(nf):
        n = min(W, T - k * S)
        o[:, k, :n] = x[:, k * S:k * S + n]
    return out


def case_lstm_bwd(lib, dev, H, nseq, W, in_ch=None, seed=240, framed_T=None):
    """one bidirectional nn.LSTM layer: training-mode forward (gates / cell states saved) + aero_lstm_bwd + the GEMMs of the host
    side, against torch.autograd of nn.LSTM on fp16-rounded weights and inputs.  framed_T: dout arrives stitched (out_mode 1)."""
    from aero_amd import backward as bw
    ops = Ops(lib)
    in_ch = H if in_ch is None else in_ch
    torch.manual_seed(seed)
    ref = torch.nn.LSTM(in_ch, H, num_layers=1, bidirectional=True)
    with torch.no_grad():
        for p_ in ref.parameters():
            p_.copy_(q16(p_))
    x = q16(_rand((nseq, W, in_ch), seed + 1)).requires_grad_()
    y = ref(x.permute(1, 0, 2))[0].permute(1, 0, 2)                      # [nseq, W, 2H]
    sd = {'l.' + k: v.detach() for k, v in ref.state_dict().items()}
    spec, xb, whh, fused = pack.pack_lstm_layer(lib, sd, 'l', 0, H, dev)
    out = torch.empty(nseq, W, 2 * H, dtype=torch.float16, device=dev)
    save = TO.lstm_save_buffers(nseq, W, H, dev)
    xd = x.detach().half().to(dev)
    if fused is not None:
        ops.lstm(None, None, whh, H, nseq, W, 0, 0, 1, 1, W, out, x=xd, fused=fused, save=save)
    else:
        xp = ops.conv(spec, xd.view(1, 1, nseq * W, in_ch), None, 1, 1, 1, nseq * W)
        ops.lstm(xp, xb, whh, H, nseq, W, 0, 0, 1, 1, W, out, save=save)
    assert rel_l2(out.cpu().float(), y.detach()) < TOL16
    whh_t = TO.pack_whh_t(lib, [sd['l.weight_hh_l0'], sd['l.weight_hh_l0_reverse']], H, dev)
    if framed_T is None:
        dy = q16(_rand((nseq, W, 2 * H), seed + 2, 0.5))
        y.backward(dy)
        da = TO.lstm_bwd(ops, dy.half().to(dev), whh_t, save[0], save[1], H, nseq, W)
    else:
        T, S = framed_T, W // 2
        nf = math.ceil(T / S)
        R = nseq // nf
        dys = q16(_rand((R, T, 2 * H), seed + 2, 0.5))
        smap = O.stitch_map(T, W, S, nf)
        dyf = torch.zeros(R, nf, W, 2 * H)
        for t, (k, tau) in enumerate(smap):
            dyf[:, k, tau] = dys[:, t]
        y.backward(dyf.view(nseq, W, 2 * H))
        da = TO.lstm_bwd(ops, dys.half().to(dev), whh_t, save[0], save[1], H, nseq, W, out_mode=1, nframes=nf, S=S, T=T)
    g = lstm_layer_grads(ops, da, xd, out, sd, 'l', 0, H, nseq, W, in_ch, dev)
    tol = 3 * TOL16
    for k, v in g.items():
        if k == 'dx':
            assert rel_l2(v.cpu().float(), x.grad) < tol, k
        else:
            assert rel_l2(v.cpu(), dict(ref.named_parameters())[k].grad) < tol, (k, rel_l2(v.cpu(), dict(ref.named_parameters())[k].grad))


def lstm_layer_grads(ops, da, x, out, sd, pre, layer, H, nseq, W, in_ch, dev):
    """parameter gradients (aero_amd.train.lstm_param_grads, scattered into zeroed nn.LSTM-shaped buffers as the engine does into its
    flat buffer) and dx = da W_ih (the engine's `.ih_dgrad` image)"""
    from aero_amd.train import lstm_param_grads
    sdl = {k[len(pre) + 1:]: v for k, v in sd.items()}
    g = {}

    def put(name, rows, perm):
        g.setdefault(name, torch.zeros(sdl[name].shape, dtype=torch.float32, device=dev)).index_copy_(0, perm, rows)
    lstm_param_grads(ops, da, x, out, layer, H, nseq, W, in_ch, dev, put)
    perm = pack.lstm_gate_perm(H, dev)
    wt = torch.cat([sdl[f'weight_ih_l{layer}{sfx}'].detach().float().to(dev)[perm] for sfx in ('', '_reverse')], 0).t().contiguous()
    spec = pack.make_conv_spec(wt[None, :, None, :], None, 8 * H, 0, [0], [0], dev)
    g['dx'] = ops.conv(spec, da.view(1, 1, nseq * W, 8 * H), None, 1, 1, 1, nseq * W).view(nseq, W, in_ch)
    return g


def case_localstate_bwd(lib, dev, Cc, heads, R, T, seed=250):
    ops = Ops(lib)
    nd = 4
    ld = 3 * Cc + heads * nd
    qkvd = q16(_rand((R, T, ld), seed, 0.8))
    qkvd[..., 3 * Cc:] = q16(qkvd[..., 3 * Cc:] * 2 - 1.0)
    qkvd = qkvd.requires_grad_()
    dh = Cc // heads
    q = qkvd[..., :Cc].view(R, T, heads, dh)
    k = qkvd[..., Cc:2 * Cc].view(R, T, heads, dh)
    v = qkvd[..., 2 * Cc:3 * Cc].view(R, T, heads, dh)
    dq = qkvd[..., 3 * Cc:].view(R, T, heads, nd)
    idx = torch.arange(T, dtype=torch.float32)
    delta = idx[:, None] - idx[None, :]
    dots = torch.einsum('rthc,rshc->rhts', k, q) / dh ** 0.5
    decays = torch.arange(1, nd + 1, dtype=torch.float32)
    kern = -decays.view(-1, 1, 1) * delta.abs() / nd ** 0.5
    dots = dots + torch.einsum('fts,rshf->rhts', kern, torch.sigmoid(dq) / 2)
    dots = dots.masked_fill(torch.eye(T, dtype=torch.bool), -100)
    w = torch.softmax(dots, dim=2)
    out = torch.einsum('rhts,rthc->rshc', w, v).reshape(R, T, Cc)
    dout = q16(_rand((R, T, Cc), seed + 1))
    out.backward(dout)
    o16 = out.detach().half().to(dev)
    att = ops.localstate(qkvd.detach().half().to(dev), R, T, Cc, heads, nd)
    assert rel_l2(att.cpu().float(), out.detach()) < 3e-3
    g = TO.localstate_bwd(ops, qkvd.detach().half().to(dev), o16, dout.half().to(dev), R, T, Cc, heads, nd)
    ref = qkvd.grad
    for nm, sl in (('dq', slice(0, Cc)), ('dk', slice(Cc, 2 * Cc)), ('dv', slice(2 * Cc, 3 * Cc)), ('ddecay', slice(3 * Cc, ld))):
        assert rel_l2(g.cpu().float()[..., sl], ref[..., sl]) < TOL16, (nm, rel_l2(g.cpu().float()[..., sl], ref[..., sl]))


def _mag_ref(x, n, h, w):
    z = torch.stft(x, n, h, w, torch.hann_window(w), return_complex=True)
    return torch.sqrt(torch.clamp(z.real ** 2 + z.imag ** 2, min=1e-7)).transpose(2, 1)


def case_stft_loss(lib, dev, n_fft, hop, win, L, B=2, seed=260):
    """one resolution of the multi-resolution STFT loss (stft_loss.py:84-117): values and the gradient w.r.t. the predicted signal
    (aero_stft_fwd -> aero_stft_loss_sums / _bwd -> aero_irfft_frames + aero_stft_adj_fold) against torch.autograd through torch.stft"""
    ops = Ops(lib)
    x = _rand((B, L), seed).requires_grad_()
    y = _rand((B, L), seed + 1)
    xm, ym = _mag_ref(x, n_fft, hop, win), _mag_ref(y, n_fft, hop, win)
    sc = torch.norm(ym - xm, p='fro') / torch.norm(ym, p='fro')
    mg = torch.nn.functional.l1_loss(torch.log(ym), torch.log(xm))
    (0.3 * sc + 0.7 * mg).backward()
    wpad = _hann_padded(win, n_fft, dev)
    T = 1 + L // hop
    zx = ops.stft(x.detach().to(dev), L, L, n_fft, hop, wpad, n_fft // 2 + 1)
    zy = ops.stft(y.to(dev), L, L, n_fft, hop, wpad, n_fft // 2 + 1)
    assert zx.shape[2] == T
    sums = TO.stft_loss_sums(ops, zx, zy, float(n_fft))
    s = sums.cpu()
    n = zx.numel() // 2
    assert abs(float((s[0] / s[1]).sqrt()) - float(sc.detach())) < 1e-4 * float(sc.detach())          # (torch reduces in fp32, the kernel in fp64)
    assert abs(float(s[2] / n) - float(mg.detach())) < 1e-4 * float(mg.detach())
    gout = torch.tensor([0.3, 0.7], dtype=torch.float32, device=dev)
    g = TO.stft_loss_bwd(ops, zx, zy, float(n_fft), sums, 1.0, 1.0, gout)
    dx = TO.stft_adjoint(ops, g, n_fft, hop, wpad, L)
    # fp32 throughout; the L1 term's sign(log xmag - log ymag) is discontinuous, so a handful of near-ties may flip: 1e-3
    assert rel_l2(dx.cpu(), x.grad) < 1e-3, rel_l2(dx.cpu(), x.grad)
    dx2 = TO.stft_adjoint(ops, g, n_fft, hop, wpad, L, dx=dx.clone())
    assert rel_l2(dx2.cpu(), 2 * x.grad) < 1e-3


def case_scale_cast(lib, dev, B=3, n=1000, seed=270):
    ops = Ops(lib)
    x = _rand((B, n), seed, 1e-5)
    sc = torch.tensor([0.5, 2.0, 1.0][:B])
    y, scale = TO.scale_cast(ops, x.to(dev), sc.to(dev), 1024.0)
    am = float((x * sc[:, None]).abs().max())
    S = 2.0 ** math.floor(math.log2(1024.0 / am))
    assert float(scale[0]) == S and float(scale[1]) == 1.0 / S
    assert rel_l2(y.cpu().float(), x * sc[:, None] * S) < 1e-3
    buf = torch.full((777,), 3.0, device=dev)
    TO.scale_f32(ops, buf, scale[1:])
    assert torch.allclose(buf.cpu(), torch.full((777,), 3.0 / S))
    a, b_ = q16(_rand((1001,), seed + 1)), q16(_rand((1001,), seed + 2))
    pad = torch.zeros(7)
    s = TO.add_f16(ops, torch.cat([a, pad]).half().to(dev), torch.cat([b_, pad]).half().to(dev))
    assert torch.equal(s.cpu()[:1001], (a + b_).half())


class _Holder(torch.nn.Module):
    """a one-attribute module tree so that TrainEngine sees state-dict names like '<attr>.conv1.0.weight'"""

    def __init__(self, **mods):
        super().__init__()
        for k, v in mods.items():
            setattr(self, k, v)
    nfft = 512


def _train_engine(lib, holder, dev):
    from aero_amd.train import TrainEngine
    eng = TrainEngine(holder, lib=lib)
    eng._sync_weights(torch.device(dev))
    eng.g = {k: torch.zeros_like(v, dtype=torch.float32, device=dev) for k, v in holder.named_parameters()}
    return eng


def case_ftb_autograd(lib, dev, Cc, Fq, T, B=2, seed=300):
    """the whole FTB in training mode (modules.py:304-325), forward and backward on the HIP kernels (aero_amd/train.py) against
    torch.autograd through the oracle's restatement on fp16-rounded weights / input"""
    from aero_amd.modules import FTB
    torch.manual_seed(seed)
    ftb = FTB(input_dim=Fq, in_channel=Cc).train()
    with torch.no_grad():
        for n_, p_ in ftb.named_parameters():
            if p_.dim() == 1 and 'weight' in n_:
                p_.add_(0.3 * torch.randn_like(p_))             # BatchNorm gammas off 1
            p_.copy_(q16(p_))
    hold = _Holder(fab=ftb).to(dev)
    x = q16(_rand((B, Cc, Fq, T), seed + 1))
    dy = q16(_rand((B, Cc, Fq, T), seed + 2))
    sd = {k: v.detach().clone().cpu().requires_grad_(v.is_floating_point()) for k, v in hold.state_dict().items()}
    xr = x.clone().requires_grad_()
    with O.fp16_storage():              # the conv outputs in front of the BatchNorm + ReLU rounded as the product stores them: a ReLU mask
        O.ftb(sd, 'fab', xr, train=True, new_stats={}).backward(dy)    # is a discontinuous function of them (tests/train_cases.py)
    eng = _train_engine(lib, hold, dev)
    y, r = eng._ftb_fwd('fab', hold.fab, cl(x).to(dev), B, Fq, T)
    assert rel_l2(uncl(y.cpu()), O.ftb({k: v.detach() for k, v in sd.items()}, 'fab', x, train=True)) < TOL16
    dx = eng._ftb_bwd('fab', hold.fab, r, cl(dy).to(dev), B, Fq, T)
    errs = {'dx': rel_l2(uncl(dx.cpu()), xr.grad)}
    for k, p_ in hold.named_parameters():
        gref = sd[k].grad
        if not k.endswith('.0.bias'):                             # (conv biases in front of a batch-statistics BatchNorm have zero gradient)
            errs[k] = rel_l2(eng.g[k].cpu(), gref)
    # same inputs, same rounding points: what is left are pre-activations whose fp32 accumulation order moves them across an fp16
    # rounding boundary next to zero (a flipped ReLU mask): ~1e-4 of the elements at C = 384 -> 1e-2; 5e-4 at the small shapes
    tol = 2e-2 if Cc * Fq * T > 100000 else 3 * TOL16
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, bad
    return errs


# ------------------------------------------------------------------------------------------------
# the reference's own module vectors (tests/golden/modules.npz, oracle/make_golden.py: BLSTM / LocalState / FTB / Snake / DConv /
# HEncLayer / HDecLayer of the REFERENCE on seeded inputs) fed to the HIP kernels -- VERDICT r2 missing #5
def _mod_golden(tag):
    from conftest import load_npz
    g = load_npz('modules.npz')
    w = {k[len(tag) + 3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + '.w.')}
    i = {k[len(tag) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + '.in.')}
    o = {k[len(tag) + 5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + '.out.')}
    return w, i, o


def _engine_for(lib, holder, dev):
    from aero_amd.engine import HipEngine
    eng = HipEngine(holder.to(dev), lib=lib)
    eng._train = False
    eng._prepare(torch.device(dev))
    return eng


def case_module_golden(lib, dev, tag):
    """returns the measured rel-L2 errors {case: error} against the REFERENCE's outputs (fp32 weights and inputs as committed)"""
    from aero_amd import modules as M
    from aero_amd.engine import HipEngine
    ops = Ops(lib)
    w, i, o = _mod_golden(tag)
    errs = {}
    if tag == 'blstm':
        H = 8

        class _DC:
            hidden = H
        sd = {'b.' + k: v for k, v in w.items()}
        eng = HipEngine.__new__(HipEngine)
        eng.lib, eng.ops, eng.fuse_lstm_proj = lib, ops, True
        L = {'lstm': [pack.pack_lstm_layer(lib, sd, 'b.lstm', l, H, dev) for l in range(2)]}
        wl = sd['b.linear.weight']
        L['lstm_lin'] = pack.make_conv_spec(wl[None, :, None, :], sd['b.linear.bias'], 2 * H, 0, [0], [0], dev)
        for case in ('framed', 'unframed'):
            x = i[case]
            R, _, T = x.shape
            h = x.permute(0, 2, 1).contiguous().half().view(1, R, T, H).to(dev)
            y = eng._blstm(_DC, L, h, 1, R, T)
            errs[case] = rel_l2(y.cpu().float()[0].permute(0, 2, 1), o[case])
    elif tag == 'localstate':
        Cc, heads, nd = 16, 4, 4
        wq = torch.cat([w[f'{n}.weight'][:, :, 0] for n in ('query', 'key', 'content', 'query_decay')], 0)
        bq = torch.cat([w[f'{n}.bias'] for n in ('query', 'key', 'content', 'query_decay')], 0)
        qk = pack.make_conv_spec(wq[None, :, None, :], bq, Cc, 0, [0], [0], dev)
        pj = pack.make_conv_spec(w['proj.weight'][:, :, 0][None, :, None, :], w['proj.bias'], Cc, 0, [0], [0], dev)
        x = i['x']
        R, _, T = x.shape
        h = x.permute(0, 2, 1).contiguous().half().view(1, R, T, Cc).to(dev)
        qkvd = ops.conv(qk, h, None, 1, R, R, T)
        att = ops.localstate(qkvd, R, T, Cc, heads, nd)
        y = ops.conv(pj, att.view(1, R, T, Cc), None, 1, R, R, T, res=h)
        errs['y'] = rel_l2(y.cpu().float()[0].permute(0, 2, 1), o['y'])
    elif tag == 'snake':
        x = i['x']                                                       # [B, hid, T, F]: a per frequency bin (modules.py:232-236)
        xc = x.permute(0, 3, 2, 1).contiguous().half().to(dev)           # [B, F, T, hid]
        y = ops.norm_act(xc, 1, True, None, None, _lib.ACT_SNAKE, snake_a=w['a'].reshape(-1).float().to(dev), normalize=False)
        errs['y'] = rel_l2(y.cpu().float().permute(0, 3, 2, 1), o['y'])
    elif tag == 'ftb':
        ftb = M.FTB(input_dim=16, in_channel=8)
        ftb.load_state_dict(w)
        x = i['x']
        B, Cc, Fq, T = x.shape
        # train mode (batch statistics).  The committed state is the one AFTER the reference's train-mode call: the output does not
        # depend on the running statistics, only the update does -- checked by the oracle test (test_ftb_module_golden_eval_and_train)
        hold = _Holder(fab=ftb).to(dev)
        eng = _train_engine(lib, hold, dev)
        y, _ = eng._ftb_fwd('fab', hold.fab.train(), cl(x).to(dev), B, Fq, T)
        errs['train'] = rel_l2(uncl(y.cpu()), o['train'])
        # eval mode: BatchNorm on the running statistics BEFORE that call = what the oracle reproduces; here the folded-conv path of
        # the inference engine on the committed state against the oracle on the same state
        enc = M.HEncLayer(8, 8, kernel_size=8, stride=4, norm_groups=4, dconv=False, freq_attn=True, freq_dim=16, norm=False, rewrite=False)
        enc.freq_attn_block.load_state_dict(w)
        holder = _Holder(encoder=torch.nn.ModuleList([enc]), decoder=torch.nn.ModuleList([]))
        holder.freq_emb = None
        e2 = _engine_for(lib, holder, dev)
        ye = e2._encode_head_unfused(e2.P['encoder.0'], cl(x).to(dev), B, Fq, T)
        ref = O.ftb({'m.' + k: v for k, v in w.items()}, 'm', x)
        errs['eval'] = rel_l2(uncl(ye.cpu()), ref)
    elif tag == 'dconv':
        dc = M.DConv(16, compress=4, depth=2, init=0.5, norm=True, time_attn=True, heads=4, ndecay=4, lstm=True, act_func='snake', freq_dim=4,
                     reshape=True)
        dc.load_state_dict(w)
        x = i['x']
        B, Cc, Fq, T = x.shape
        eng = HipEngine.__new__(HipEngine)
        eng.lib, eng.ops = lib, ops
        eng.fuse_lstm_proj, eng.fuse_dconv_tail, eng.gram_stats, eng.fuse_stats, eng.fuse_dconv_row = True, True, True, 'auto', True
        eng._tables = {}
        layers = eng._pack_dconv({'m.' + k: v.float() for k, v in w.items()}, 'm', dc, dev)
        ops.begin_step(torch.device(dev))
        y = eng._dconv(dc, layers, cl(x).to(dev), B, Fq, T)
        errs['y'] = rel_l2(uncl(y.cpu()), o['y'])
    elif tag == 'henc':
        enc = M.HEncLayer(4, 8, kernel_size=8, stride=4, norm_groups=4, freq=True, dconv=False, is_first=False, freq_attn=False, freq_dim=32,
                          norm=True, context=0, pad=True, rewrite=True)
        enc.load_state_dict(w)
        holder = _Holder(encoder=torch.nn.ModuleList([enc]), decoder=torch.nn.ModuleList([]))
        holder.freq_emb = None
        eng = _engine_for(lib, holder, dev)
        x = i['x']
        B, _, Fq, T = x.shape
        eng.ops.begin_step(torch.device(dev))
        y, Fo = eng._encode(0, enc, eng.P['encoder.0'], cl(x).to(dev), B, Fq, T)
        errs['y'] = rel_l2(uncl(y.cpu()), o['y'])
    elif tag == 'hdec':
        dec = M.HDecLayer(16, 4, last=False, kernel_size=8, stride=4, norm_groups=4, freq=True, dconv=False, norm=True, context=1, pad=True,
                          context_freq=True, rewrite=True)
        dec.load_state_dict(w)
        holder = _Holder(encoder=torch.nn.ModuleList([]), decoder=torch.nn.ModuleList([dec]))
        holder.freq_emb = None
        eng = _engine_for(lib, holder, dev)
        x, sk = i['x'], i['skip']
        B, _, Fq, T = x.shape
        eng.ops.begin_step(torch.device(dev))
        y = eng._decode(0, dec, eng.P['decoder.0'], cl(x).to(dev), cl(sk).to(dev), B, Fq, T, None, None)
        errs['y'] = rel_l2(uncl(y.cpu()), o['y'])
    else:
        raise KeyError(tag)
    return errs


def case_bn_running_update(lib, dev, nc=37, n=1234.0, mom=0.1, seed=400):
    """aero_bn_running_update against the bookkeeping of torch.nn.BatchNorm in training mode"""
    import ctypes as C
    x = _rand((int(n), nc), seed, 1.3) + 0.4
    bn = torch.nn.BatchNorm1d(nc, momentum=mom)
    with torch.no_grad():
        bn.running_mean.copy_(_rand((nc,), seed + 1, 0.2))
        bn.running_var.copy_(_rand((nc,), seed + 2, 0.2).abs() + 0.5)
    rm, rv, nbt = bn.running_mean.clone().to(dev), bn.running_var.clone().to(dev), bn.num_batches_tracked.clone().to(dev)
    bn.train()
    bn(x)
    st = torch.stack([x.double().sum(0), (x.double() ** 2).sum(0)], 1).contiguous().to(dev)
    stream = torch.cuda.current_stream().cuda_stream if str(dev) != 'cpu' else 0
    lib.call('aero_bn_running_update', st.data_ptr(), nc, C.c_double(n), C.c_float(mom), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), stream)
    assert torch.allclose(rm.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(nbt) == int(bn.num_batches_tracked) == 1
