"""torch.autograd glue for the backward kernels: one Function per building block of the generator,
forward AND backward on the HIP kernels -- conv (aero_conv_fwd) -> GroupNorm -> GELU / GLU (aero_norm_stats/apply), then
aero_norm_bwd_* -> data gradient (aero_conv_fwd on re-packed weights) + aero_conv_wgrad.  Tensors are channels-last fp16
[B, F, T, C] on the device; parameters and their gradients are fp32 in the nn.Module layout (aero.py:86-101,172-179).

These per-block Functions are the building blocks the op-level gradient tests drive (tests/op_cases.py); the whole model trains
through aero_amd/train.py (TrainEngine + AeroFunction: one hand-written reverse walk, weight images replayed after optimizer steps),
which is what `Aero.forward` uses under autograd."""
import torch

from . import _lib, backward as bw, pack
from .engine import Ops

_ACT = {'none': _lib.ACT_NONE, 'gelu': _lib.ACT_GELU, 'glu': _lib.ACT_GLU, 'snake': _lib.ACT_SNAKE}


class ConvNormAct(torch.autograd.Function):
    """y = act(GroupNorm_G(conv(x)))   --  HEncLayer conv+norm1+act, rewrite+norm2+glu, HDecLayer conv_tr+norm2+act, with act = GELU
    or Snake (act_func of the config; `alpha` = Snake's per-frequency-row parameter [F_out], snake.py:67) and G = 0 for the layers
    before norm_starts (identity norm; gamma = beta = None); act 'none' with G = 0 is the bare conv (the last decoder layer).
    skip: a second input ADDED to x before the conv (HDecLayer.forward: x = x + skip, aero.py:195) -- run as a two-source conv with
    the weights seen twice, so no separate add pass exists in either direction.
    kind: ('conv2d', pad_f, pad_t) | ('fstride', stride) [kernel [K,1], padding (K-stride)//2] | ('convtr', stride) [cropped]"""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, lib, kind, G, act, alpha=None, skip=None):
        ops = Ops(lib)
        dev = x.device
        B, Fin, T, Cin = x.shape
        w = weight.detach().float().cpu()
        two = skip is not None
        dup = (lambda t: torch.cat([t, t], -1)) if two else (lambda t: t)
        C1 = Cin if two else 0
        if kind[0] == 'conv2d':
            taps, df, dt = pack.conv2d_taps(w, kind[1], kind[2])
            spec = pack.make_conv_spec(dup(taps), bias.detach(), Cin, C1, df, dt, dev)
            Fout, kw = Fin, {}
        elif kind[0] == 'fstride':
            K, s = w.shape[2], kind[1]
            pad = (K - s) // 2
            taps, df, dt = pack.conv2d_taps(w, pad, 0)
            spec = pack.make_conv_spec(dup(taps), bias.detach(), Cin, C1, df, dt, dev, fstride=s)
            Fout, kw = (Fin + 2 * pad - K) // s + 1, {}
        else:
            K, s = w.shape[2], kind[1]
            pad = (K - s) // 2
            taps, df, dt = pack.convtr_taps(w, s)
            spec = pack.make_conv_spec(dup(taps), bias.detach(), Cin, C1, df, dt, dev, transposed=1, fstride=s)
            Fu = (Fin - 1) * s + K
            Fout, kw = Fu, dict(dst_f_off=pad, dst_F=Fu - 2 * pad)
        h = ops.conv(spec, x, skip, B, Fin, Fout, T, **kw)
        sa = None if alpha is None else alpha.detach().float()
        if G:
            y = ops.norm_act(h, G, 0, gamma.detach(), beta.detach(), _ACT[act], snake_a=sa)
            stats = ops._last_stats
        elif act == 'none':
            y, stats = h, None
        else:
            y = ops.norm_act(h, 1, 0, None, None, _ACT[act], snake_a=sa, normalize=False)
            stats = None
        ctx.save_for_backward(x, weight, gamma, beta, h, stats, alpha, skip)
        ctx.cfg = (lib, kind, G, act, df, dt)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, gamma, beta, h, stats, alpha, skip = ctx.saved_tensors
        lib, kind, G, act, df, dt = ctx.cfg
        ops = Ops(lib)
        dev = x.device
        B, Fin, T, Cin = x.shape
        Fh = h.shape[1]
        if not G and act == 'none':
            dh, dgamma, dbeta, dalpha = dy.contiguous(), None, None, None
        else:
            res = bw.norm_bwd(ops, h, dy.contiguous(), stats, G if G else 1, 0, gamma.detach() if G else None, beta.detach() if G else None,
                              _ACT[act], snake_a=None if alpha is None else alpha.detach().float())
            dh, dgamma, dbeta = res[0], res[1], res[2]
            dalpha = res[4] if alpha is not None else None
        w = weight.detach().float().cpu()
        if kind[0] == 'conv2d':
            dx = ops.conv(bw.dgrad_conv2d(w, kind[1], kind[2], dev), dh, None, B, Fh, Fin, T)
            dw, db = bw.conv_wgrad(ops, dh, x, df, dt)
            if skip is not None:
                dw, _ = bw.conv_wgrad(ops, dh, skip, df, dt, bias=False, dw_acc=dw)
            kF, kT = w.shape[2], w.shape[3]
            dweight = dw.view(kF, kT, w.shape[0], Cin).permute(2, 3, 0, 1)
        elif kind[0] == 'fstride':
            K, s = w.shape[2], kind[1]
            pad = (K - s) // 2
            dx = ops.conv(bw.dgrad_conv_fstride(w, s, dev), dh, None, B, Fh, (Fh - 1) * s + K, T, dst_f_off=pad, dst_F=Fin)
            dw, db = bw.conv_wgrad(ops, dh, x, df, dt, fstride=s)
            if skip is not None:
                dw, _ = bw.conv_wgrad(ops, dh, skip, df, dt, fstride=s, bias=False, dw_acc=dw)
            dweight = dw.permute(1, 2, 0).unsqueeze(-1)
        else:
            K, s = w.shape[2], kind[1]
            pad = (K - s) // 2
            dx = ops.conv(bw.dgrad_convtr(w, s, pad, dev), dh, None, B, Fh, Fin, T)
            dw, _ = bw.conv_wgrad(ops, x, dh, [kk - pad for kk in range(K)], [0] * K, fstride=s, bias=False)
            if skip is not None:
                dw, _ = bw.conv_wgrad(ops, skip, dh, [kk - pad for kk in range(K)], [0] * K, fstride=s, bias=False, dw_acc=dw)
            dweight = dw.permute(1, 2, 0).unsqueeze(-1)
            db = _bias_grad(ops, dh)
        return dx, dweight.contiguous(), db, dgamma, dbeta, None, None, None, None, dalpha, (dx if skip is not None else None)


def _bias_grad(ops, dh):
    """db[m] = sum of dh over positions, for the ConvTranspose block whose weight-gradient call has x and dy swapped: the bias
    reduction of aero_conv_wgrad on a 1-tap, 8-channel problem (the [M, 8] dw it also produces is discarded)."""
    _, db = bw.conv_wgrad(ops, dh, dh[..., :8], [0], [0], bias=True)
    return db


def _pad0(t, n, dim=0):
    """zero-pad dimension `dim` of a parameter to n entries"""
    if t.shape[dim] == n:
        return t
    shape = list(t.shape)
    shape[dim] = n - t.shape[dim]
    return torch.cat([t, torch.zeros(shape, dtype=t.dtype, device=t.device)], dim)


class DConvLayer(torch.autograd.Function):
    """One residual DConv layer without LSTM / attention (modules.py:206-244):
        y = x + scale * GLU(GroupNorm1(conv1x1(GELU(GroupNorm1(conv1d_k,dilation(x))))))       on rows [B*F, C, T]
    x fp16 [B,F,T,C]; w1 [H, C, k], w2 [2C, H, 1] (H = C / compress).  The hidden width is zero-padded to a multiple of 8 for the
    16-byte channel vectors of the kernels (zero weights, gamma = beta = 0 there; the statistics count stays T * H)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, scale, lib, dilation):
        ops = Ops(lib)
        dev = x.device
        B, Fr, T, Cc = x.shape
        H, _, k = w1.shape
        Hp = (H + 7) // 8 * 8
        pad = dilation * (k // 2)
        w1p, b1p = _pad0(w1.detach().float().cpu(), Hp), _pad0(b1.detach().float(), Hp)
        g1p, be1p = _pad0(g1.detach().float(), Hp), _pad0(be1.detach().float(), Hp)
        w2p = _pad0(w2.detach().float().cpu(), Hp, 1)
        t1, df1, dt1 = pack.conv1d_taps(w1p, dilation, pad)
        h1 = ops.conv(pack.make_conv_spec(t1, b1p, Cc, 0, df1, dt1, dev), x, None, B, Fr, Fr, T)
        a = ops.norm_act(h1, 1, 1, g1p, be1p, _lib.ACT_GELU, stat_count=T * H)
        st1 = ops._last_stats
        t2, df2, dt2 = pack.conv1d_taps(w2p, 1, 0)
        h2 = ops.conv(pack.make_conv_spec(t2, b2.detach(), Hp, 0, df2, dt2, dev), a, None, B, Fr, Fr, T)
        y = ops.norm_act(h2, 1, 1, g2.detach(), be2.detach(), _lib.ACT_GLU, layer_scale=scale.detach(), res=x)
        st2 = ops._last_stats
        ctx.save_for_backward(x, h1, a, h2, st1, st2, w1p, w2p, g1p, be1p, g2, be2, scale)
        ctx.cfg = (lib, dilation, H, df1, dt1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h1, a, h2, st1, st2, w1p, w2p, g1p, be1p, g2, be2, scale = ctx.saved_tensors
        lib, dilation, H, df1, dt1 = ctx.cfg
        ops = Ops(lib)
        dev = x.device
        B, Fr, T, Cc = x.shape
        k = w1p.shape[2]
        dy = dy.contiguous()
        dh2, dg2, dbe2, dls = bw.norm_bwd(ops, h2, dy, st2, 1, 1, g2.detach(), be2.detach(), _lib.ACT_GLU, layer_scale=scale.detach())
        dw2, db2 = bw.conv_wgrad(ops, dh2, a, [0], [0])
        da = ops.conv(bw.dgrad_conv1d(w2p, 1, 0, dev), dh2, None, B, Fr, Fr, T)
        dh1, dg1, dbe1, _ = bw.norm_bwd(ops, h1, da, st1, 1, 1, g1p, be1p, _lib.ACT_GELU, stat_count=T * H)
        dw1, db1 = bw.conv_wgrad(ops, dh1, x, df1, dt1)
        dx = ops.conv(bw.dgrad_conv1d(w1p, dilation, dilation * (k // 2), dev), dh1, None, B, Fr, Fr, T, res=dy)   # + the skip path
        return (dx, dw1.permute(1, 2, 0)[:H].contiguous(), db1[:H], dg1[:H], dbe1[:H],
                dw2.permute(1, 2, 0)[:, :H].contiguous(), db2, dg2, dbe2, dls, None, None)
