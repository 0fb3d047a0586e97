"""Tensor-level wrappers over the training-step kernels of csrc/k_train.h (include/aero_hip.h, "the rest of one training
step"): host plumbing only -- buffers from the PyTorch allocator, one C-ABI call per kernel, no arithmetic with ATen ops.
Used by aero_amd/train.py (the backward of Aero.forward) and by the op-level tests.  Reference: `loss.backward()` of
src/solver.py:602-605 through src/models/modules.py (FTB :304-325, BLSTM :32-65, LocalState :94-127) and stft_loss.py."""
import ctypes as C

import torch

from . import _lib, pack
from .engine import _ptr


def _f16(shape, like):
    return torch.empty(shape, dtype=torch.float16, device=like.device)


def freqfc_wgrad(ops, dfc, x, gate, nslab=None):
    """dW[f][f'] = sum_{b,t,c} dfc[b,f,t,c] * gate[b,t,c] * x[b,f',t,c]   (fp32 [F, F]; modules.py:296,320)"""
    B, F, T, Cc = x.shape
    assert dfc.is_contiguous() and x.is_contiguous() and gate.is_contiguous()
    if nslab is None:
        chunks = B * ((T * Cc + 31) // 32)
        nslab = max(1, min(256, chunks // 64, 1024 // max(1, ((F + 63) // 64) ** 2) + 1))
    dw = ops.zeros32(F * F, x.device).view(F, F)
    slabs = torch.empty(nslab, F, F, dtype=torch.float32, device=x.device)
    ops.lib.call('aero_freqfc_wgrad', _ptr(dfc), _ptr(x), _ptr(gate), _ptr(dw), _ptr(slabs), nslab, B, F, T, Cc, ops.stream(x))
    return dw


def ftb_gate_bwd(ops, v, x, gate, add=None):
    """(dx = add + v * gate, dgate = sum_f v * x): modules.py:316 with the gate behind freq_fc (k_ftb.h)"""
    B, F, T, Cc = x.shape
    assert v.is_contiguous() and x.is_contiguous() and gate.is_contiguous() and (add is None or add.is_contiguous())
    dx, dgate = _f16(x.shape, x), _f16((B, T, Cc), x)
    ops.lib.call('aero_ftb_gate_bwd', _ptr(v), _ptr(x), _ptr(gate), _ptr(add), _ptr(dx), _ptr(dgate), B, F, T, Cc, ops.stream(x))
    return dx, dgate


def sum_bt(ops, x, out, scale=1.0):
    """out[f][c] += scale * sum_{b,t} x[b,f,t,c]"""
    B, F, T, Cc = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32 and out.numel() == F * Cc
    ops.lib.call('aero_sum_bt', _ptr(x), _ptr(out), B, F, T, Cc, C.c_float(scale), ops.stream(x))
    return out


def frames_op(ops, src, mode, R, T, Cc, nframes, W, S):
    """mode 0 unfold rows->frames, 1 its adjoint, 2 stitch frames->rows, 3 its adjoint (models/utils.py:22-35, modules.py:49-61)"""
    assert src.is_contiguous()
    dst = _f16((R * nframes, W, Cc) if mode in (0, 3) else (R, T, Cc), src)
    ops.lib.call('aero_frames_op', _ptr(src), _ptr(dst), mode, R, T, Cc, nframes, W, S, ops.stream(src))
    return dst


def lstm_save_buffers(nseq, W, H, device):
    nblk = (nseq + 15) // 16
    return (torch.empty(nblk * 2 * W * H * 64, dtype=torch.float16, device=device),
            torch.empty(nblk * 2 * W * H * 16, dtype=torch.float32, device=device))


def pack_whh_t(lib, w_hh_dirs, H, device):
    """[W_hh forward, W_hh reverse] (nn.LSTM layout [4H, H], gate blocks i,f,g,o) -> fp16 [2][HP][K4P] for aero_lstm_bwd"""
    K4P = int(lib.cdll.aero_lstm_bwd_k4p(H))
    if K4P <= 0:
        raise _lib.AeroHipError(f'aero_lstm_bwd: hidden size {H} unsupported')
    HP = (H + 15) // 16 * 16
    perm = pack.lstm_gate_perm(H, w_hh_dirs[0].device)
    img = torch.zeros(2, HP, K4P, dtype=torch.float32, device=w_hh_dirs[0].device)
    for dr in range(2):
        img[dr, :H, :4 * H] = w_hh_dirs[dr].detach().float()[perm].t()
    return img.to(device=device, dtype=torch.float16).contiguous()


def lstm_bwd(ops, dout, whh_t, save_gates, save_c, H, nseq, W, out_mode=0, nframes=1, S=1, T=1):
    """-> da fp16 [nseq*W, 2, 4H] (gate pre-activation gradients, column 4*j + gate)"""
    d = _lib.LstmBwdDesc()
    da = torch.empty(nseq * W, 2, 4 * H, dtype=torch.float16, device=dout.device)
    d.dout, d.whh_t, d.save_gates, d.save_c, d.da = _ptr(dout), _ptr(whh_t), _ptr(save_gates), _ptr(save_c), _ptr(da)
    d.H, d.nseq, d.W, d.out_mode, d.nframes, d.S, d.T = H, nseq, W, out_mode, nframes, S, T
    ops.lib.call('aero_lstm_bwd', C.byref(d), ops.stream(dout))
    return da


def localstate_bwd(ops, qkvd, out, dout, R, T, Cc, heads, ndecay, decay_scale=1.0):
    """-> dqkvd fp16 [R, T, ld] (dQ | dK | dV | d decay pre-activations); modules.py:94-127"""
    assert qkvd.is_contiguous() and out.is_contiguous() and dout.is_contiguous()
    ld = qkvd.shape[-1]
    dq = torch.zeros(R, T, ld, dtype=torch.float16, device=qkvd.device) if ld > 3 * Cc + heads * ndecay else _f16((R, T, ld), qkvd)
    qstats = torch.empty(R, heads, T, 4, dtype=torch.float32, device=qkvd.device)
    d = _lib.AttnBwdDesc()
    d.qkvd, d.ld, d.out, d.dout, d.dqkvd, d.qstats = _ptr(qkvd), ld, _ptr(out), _ptr(dout), _ptr(dq), _ptr(qstats)
    d.R, d.T, d.C, d.heads, d.ndecay, d.decay_scale = R, T, Cc, heads, ndecay, decay_scale
    ops.lib.call('aero_localstate_bwd', C.byref(d), ops.stream(qkvd))
    return dq


def stft_loss_sums(ops, zx, zy, pscale):
    """zx, zy: fp32 [..., 2] STFTs (normalised kernel) -> 3 doubles on the device (stft_loss.py:30-64 reductions)"""
    n = zx.numel() // 2
    npart = min(1024, (n + 255) // 256)
    part = torch.empty(npart * 3, dtype=torch.float64, device=zx.device)
    sums = torch.empty(3, dtype=torch.float64, device=zx.device)
    ops.lib.call('aero_stft_loss_sums', _ptr(zx), _ptr(zy), n, C.c_float(pscale), _ptr(part), npart, _ptr(sums), ops.stream(zx))
    return sums


def stft_loss_bwd(ops, zx, zy, pscale, sums, w_sc, w_mag, gout=None):
    g = torch.empty_like(zx)
    ops.lib.call('aero_stft_loss_bwd', _ptr(zx), _ptr(zy), zx.numel() // 2, C.c_float(pscale), _ptr(sums), C.c_float(w_sc), C.c_float(w_mag),
                 _ptr(gout), _ptr(g), ops.stream(zx))
    return g


def stft_adjoint(ops, g, n_fft, hop, window, L, dx=None):
    """g fp32 [nsig, nb, T, 2]: gradient w.r.t. aero_stft_fwd's output -> dx fp32 [nsig, L] (added to `dx` if given)"""
    nsig, nb, T, _ = g.shape
    assert g.is_contiguous()
    frames = torch.empty(nsig, T, n_fft, dtype=torch.float32, device=g.device)
    ops.lib.call('aero_irfft_frames', _ptr(g), nsig, nb, T, n_fft, _ptr(window), _ptr(frames), ops.stream(g))
    acc = dx is not None
    if dx is None:
        dx = torch.empty(nsig, L, dtype=torch.float32, device=g.device)
    ops.lib.call('aero_stft_adj_fold', _ptr(frames), _ptr(dx), nsig, T, n_fft, hop, L, int(acc), ops.stream(g))
    return dx


def add_f16(ops, a, b, out=None, scale_b=1.0):
    """a + scale_b * b (fp16 tensors, fp32 arithmetic)"""
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype == torch.float16
    out = torch.empty_like(a) if out is None else out
    ops.lib.call('aero_add_f16', _ptr(a), _ptr(b), _ptr(out), a.numel(), C.c_float(scale_b), ops.stream(a))
    return out


def scale_cast(ops, x, item_scale, target):
    """x fp32 [nitems, ...] -> (fp16 of x * item_scale[item] * S, scale fp32 [2] = {S, 1/S}); S = 2^floor(log2(target / amax))"""
    nitems = x.shape[0]
    assert x.is_contiguous()
    dst = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    amax = ops.zeros32(1, x.device, torch.int32)
    scale = torch.empty(2, dtype=torch.float32, device=x.device)
    ops.lib.call('aero_scale_cast', _ptr(x), nitems, x.numel() // nitems, _ptr(item_scale), _ptr(amax), C.c_float(target), _ptr(dst), _ptr(scale),
                 ops.stream(x))
    return dst, scale


def scale_f32(ops, x, scale):
    assert x.is_contiguous() and x.dtype == torch.float32
    ops.lib.call('aero_scale_f32', _ptr(x), x.numel(), _ptr(scale), ops.stream(x))
    return x


def rescale_f16(ops, a, sa, b=None, sb=None, target=4096.0):
    """(a / Sa + b / Sb) * S as fp16 with S = 2^floor(log2(target / amax)); returns (tensor, scale fp32 [2] = {S, 1/S}).
    sa / sb: the {S, 1/S} device pairs the operands carry (None = 1)."""
    assert a.is_contiguous() and (b is None or (b.is_contiguous() and b.shape == a.shape))
    out = torch.empty_like(a)
    amax = ops.zeros32(1, a.device, torch.int32)
    scale = torch.empty(2, dtype=torch.float32, device=a.device)
    ops.lib.call('aero_rescale_f16', _ptr(a), _ptr(sa), _ptr(b), _ptr(sb), a.numel(), _ptr(amax), C.c_float(target), _ptr(out), _ptr(scale), ops.stream(a))
    return out, scale
