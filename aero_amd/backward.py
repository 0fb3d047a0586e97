"""Backward of the generator's building blocks on the MI355X (reference: `loss.backward()` in
src/solver.py:602-605 over the modules of src/models/aero.py / modules.py).

Data gradients of every convolution kind of the path are themselves convolutions of the forward family, so they run on the
forward kernels (`aero_conv_fwd`) with a re-packed weight image -- nothing here touches ATen for arithmetic:

    Conv2d, stride 1 (aero.py:95,172 rewrite; modules.py FTB)        dX = conv2d(dY, W^T flipped, pad k-1-p)
    Conv1d, dilated (modules.py:206-210 DConv)                        dX[t] = sum_j W_j^T dY[t + p - j d]
    Conv2d [K,1] / stride [s,1] (aero.py:86 encoder conv)             dX = conv_transpose2d(dY, W, stride s), rows p.. kept
    ConvTranspose2d [K,1] / stride [s,1] (aero.py:179 decoder)        dX[q] = sum_kk W[:, :, kk] dY[q s + kk - p]   (strided conv)

Weight gradients are a GEMM over positions (`aero_conv_wgrad`, k_bwd.h); GroupNorm + activation backward is
`aero_norm_bwd_*` (k_bwd.h).  LSTM, attention, FTB and STFT backward are not built yet.
"""
import torch

from . import _lib, pack


def dgrad_conv2d(w, pad_f, pad_t, device):
    """w: nn.Conv2d weight [M, C, kF, kT] of a stride-1 conv with zero padding (pad_f, pad_t).  Spec of dY [.., M] -> dX [.., C]."""
    M, Cc, kF, kT = w.shape
    wt = w.detach().float().transpose(0, 1).flip(2, 3)
    taps, df, dt = pack.conv2d_taps(wt, kF - 1 - pad_f, kT - 1 - pad_t)
    return pack.make_conv_spec(taps, None, M, 0, df, dt, device)


def dgrad_conv1d(w, dilation, padding, device):
    """w: nn.Conv1d weight [M, C, k] over time (dilation, zero padding)."""
    M, Cc, k = w.shape
    taps = w.detach().float().permute(1, 2, 0).reshape(1, Cc, k, M)
    return pack.make_conv_spec(taps, None, M, 0, [0] * k, [padding - j * dilation for j in range(k)], device)


def dgrad_conv_fstride(w, stride, device):
    """w: nn.Conv2d weight [M, C, K, 1], stride (s, 1), padding (p, 0): dX is the ConvTranspose of dY by the same weight;
    run with Fin = rows of dY, Fout = (Fin-1)*s + K, dst_f_off = p, dst_F = rows of X."""
    taps, df, dt = pack.convtr_taps(w.detach().float(), stride)
    return pack.make_conv_spec(taps, None, w.shape[0], 0, df, dt, device, transposed=1, fstride=stride)


def dgrad_convtr(w, stride, pad, device):
    """w: nn.ConvTranspose2d weight [Cin, Cout, K, 1], stride (s, 1), output rows cropped by `pad` on both sides: dX is a
    strided conv of dY (zero outside its rows); run with Fin = rows of dY, Fout = rows of X."""
    Cin, Cout, K, kT = w.shape
    assert kT == 1
    taps = w.detach().float()[:, :, :, 0].permute(0, 2, 1).reshape(1, Cin, K, Cout)
    return pack.make_conv_spec(taps, None, Cout, 0, [kk - pad for kk in range(K)], [0] * K, device, fstride=stride)


# ---------------------------------------------------------------------------------------------------------------
# kernel wrappers (k_bwd.h)
import ctypes as C  # noqa: E402

from .engine import _ptr, _strides4  # noqa: E402


def wgrad_direct_ok(dy, x):
    """can conv_wgrad write straight into a caller's destination for these operands?  (8-aligned channel counts: no padded copies)"""
    return dy.shape[-1] % 8 == 0 and x.shape[-1] % 8 == 0


def conv_wgrad(ops, dy, x, df, dt, fstride=1, bias=True, nslab=None, dw_acc=None, dw_out=None, db_out=None, layout=0, rowlen=0, coff=0):
    """dy fp16 [B,Fout,T,M], x fp16 [B,Fin,T,C] (channels-last) -> (dw fp32 [ntaps, M, C], db fp32 [M] or None):
    dw[j][m][c] = sum dy[b,fo,t,m] * x[b, fo*fstride + df[j], t + dt[j], c]  (aero_conv_wgrad).  dw_acc: an earlier result to accumulate
    into (the kernel adds to dw), e.g. the second source of a two-source conv.
    dw_out / db_out: fp32 destinations the result is ADDED to in place (views of the flat gradient buffer; nothing is allocated, filled
    or copied) -- dw_out with layout 0 as [ntaps, M, C], with layout 1 in the layout of the nn.Conv weight itself,
    [M, rowlen, taps] at column offset coff (include/aero_hip.h, aero_wgrad_desc).  Needs 8-aligned M and C (`wgrad_direct_ok`)."""
    B, Fout, T, M = dy.shape
    Bx, Fin, Tx, Cc = x.shape
    assert B == Bx and T == Tx and len(df) == len(dt)
    if M % 8 or Cc % 8:
        assert dw_out is None and db_out is None
        # narrow sides (the last decoder's 2 output channels, the FTB's 5): zero-padded copies to the kernel's 8-channel vectors
        # (data movement only), the padding rows / columns of the result dropped
        pad = lambda t: torch.nn.functional.pad(t, (0, -t.shape[-1] % 8))       # noqa: E731
        dw, db = conv_wgrad(ops, pad(dy), pad(x), df, dt, fstride, bias, nslab, dw_acc=None)
        dw, db = dw[:, :M, :Cc], (None if db is None else db[:M])
        if dw_acc is not None:
            dw_acc += dw
            return dw_acc, db
        return dw.contiguous(), db
    if len(df) == 1 and df[0] == 0 and dt[0] == 0 and fstride == 1 and Fin == Fout and dy.is_contiguous() and x.is_contiguous():
        # a pointwise product has no row structure: one long row (no ragged last 64-step segment per row -- the LSTM / attention
        # layers come as [sequences, 1, 200, .]: 200 = 3 * 64 + 8), as far as the kernel's 32-bit in-row offsets reach
        for lead in (1, B):
            n = B * Fout * T // lead
            if (n + 64) * max(M, Cc) + 512 < 2 ** 31:
                dy, x = dy.view(lead, 1, n, M), x.view(lead, 1, n, Cc)
                B, Fout, Fin, T = lead, 1, 1, n
                break
    d = _lib.WgradDesc()
    d.dy, d.x = _ptr(dy), _ptr(x)
    d.dy_b, d.dy_f, d.dy_t = _strides4(dy)
    d.x_b, d.x_f, d.x_t = _strides4(x)
    plan = nslab if nslab is not None else ops.lib.cdll.aero_conv_wgrad_chunks(M, Cc, len(df), B * Fout, T)
    store = dw_acc is None and dw_out is None and db_out is None and plan > 0   # fresh outputs, slab form: written, not added to
    if dw_out is not None:
        assert dw_acc is None and dw_out.dtype == torch.float32 and dw_out.is_contiguous() and plan > 0
        assert dw_out.numel() == len(df) * M * (rowlen or Cc) and coff + Cc <= (rowlen or Cc)
        dw = dw_out
        d.dw_layout, d.dw_rowlen, d.dw_coff = layout, rowlen, coff
    elif dw_acc is not None:
        dw = dw_acc
        assert dw.shape == (len(df), M, Cc) and dw.is_contiguous()
    else:
        dw = (torch.empty if store else torch.zeros)(len(df), M, Cc, dtype=torch.float32, device=dy.device)
    if not bias:
        db = None
    elif db_out is not None:
        assert db_out.dtype == torch.float32 and db_out.is_contiguous() and db_out.numel() == M and not store
        db = db_out
    else:
        db = (torch.empty if store else torch.zeros)(M, dtype=torch.float32, device=dy.device)
    d.dw, d.db, d.store = _ptr(dw), _ptr(db), int(store)
    d.B, d.Fin, d.Fout, d.T, d.M, d.C, d.ntaps, d.fstride = B, Fin, Fout, T, M, Cc, len(df), fstride
    for i, (a, b_) in enumerate(zip(df, dt)):
        d.df[i], d.dt[i] = a, b_
    if nslab is None:
        # position chunks: the launcher's own plan (enough blocks to fill the chip; a chunk's partial tile -- written, then read back
        # by the finish kernel -- under a quarter of the operand bytes the chunk reads), within a 1-GiB workspace
        nslab = max(1, min(plan, (1 << 28) // (len(df) * M * Cc + M)))
    if nslab:                                   # per-chunk partial slabs added in fixed order (deterministic); nslab = 0: fp32 atomics
        slabs = torch.empty(nslab, len(df) * M * Cc + (M if bias else 0), dtype=torch.float32, device=dy.device)
        d.slabs, d.nslab = _ptr(slabs), nslab
    ops.lib.call('aero_conv_wgrad', C.byref(d), ops.stream(dy))
    return dw, db


def norm_bwd(ops, x, dy, stats, G, per_row, gamma, beta, act, layer_scale=None, eps=1e-5, stat_count=None, snake_a=None, out=None):
    """Backward of aero_norm_apply (GroupNorm + GELU / GLU(+LayerScale) / identity).  x: the norm's input fp16 [B,F,T,C]; stats: the
    forward statistics (fp64 sum / sum of squares per (item, group)); dy: gradient of the output.  Returns
    (dx fp16 [B,F,T,C], dgamma, dbeta fp32 [C], dlayer_scale fp32 [C/2] or None) -- and, for Snake (act 4, snake_a fp32 [F]), the
    gradient of snake_a as a fifth item.  stats None = identity norm (the layers before norm_starts).
    out: optional dict of fp32 destinations {'dgamma', 'dbeta', 'dls', 'dsn'} the kernel ADDS into (views of the flat gradient buffer);
    what is not given comes out of ONE zero-filled scratch allocation (with the fp64 group sums)."""
    B, F, T, Cc = x.shape
    d = _lib.NormBwdDesc()
    d.x, d.dy = _ptr(x), _ptr(dy)
    d.x_b, d.x_f, d.x_t = _strides4(x)
    d.dy_b, d.dy_f, d.dy_t = _strides4(dy)
    dx = torch.empty(B, F, T, Cc, dtype=torch.float16, device=x.device)
    d.dx = _ptr(dx)
    d.dx_b, d.dx_f, d.dx_t = _strides4(dx)
    d.B, d.F, d.T, d.C, d.G, d.per_row, d.eps = B, F, T, Cc, G, int(per_row), eps
    d.stats = _ptr(stats)
    d.stat_count = float((1 if per_row == 1 else (B * F if per_row == 2 else F)) * T * (Cc // G)) if stat_count is None else float(stat_count)
    d.gamma, d.beta, d.layer_scale, d.act = _ptr(gamma), _ptr(beta), _ptr(layer_scale), act
    out = out or {}
    want = {'sums': (0 if (stats is None or per_row == 2) else stats.numel() * 2), 'dgamma': (Cc if gamma is not None else 0),
            'dbeta': (Cc if beta is not None else 0), 'dsn': (F if snake_a is not None else 0),
            'dls': (Cc // 2 if (layer_scale is not None and act == _lib.ACT_GLU) else 0)}        # fp32 words (the sums are fp64 pairs)
    # fp64 staging of the parameter-gradient sums (dgamma | dbeta | dlayer_scale | dsnake_a): order-independent (aero_norm_bwd_desc.psums)
    npar = 3 * Cc + F if any(want[k] for k in ('dgamma', 'dbeta', 'dsn', 'dls')) else 0
    want = dict(psums=2 * npar, **want)                                                          # (first: 8-byte aligned in the scratch)
    out = dict(out, psums=None)
    offs, n = {}, 0
    for k, sz in want.items():
        if sz and out.get(k) is None:
            offs[k] = n
            n += (sz + 3) // 4 * 4
    scratch = ops.zeros32(n, x.device) if n else None
    res = {}
    for k, sz in want.items():
        if not sz:
            res[k] = None
        elif k in offs:
            res[k] = scratch[offs[k]:offs[k] + sz]
        else:
            res[k] = out[k]
            assert res[k].dtype == torch.float32 and res[k].is_contiguous() and res[k].numel() == sz
    sums = None if res['sums'] is None else res['sums'].view(torch.float64).view(stats.shape)
    dgamma, dbeta, dsn, dls = res['dgamma'], res['dbeta'], res['dsn'], res['dls']
    d.snake_a, d.dsnake_a = _ptr(snake_a), _ptr(dsn)
    d.sums, d.dgamma, d.dbeta, d.dlayer_scale = _ptr(sums), _ptr(dgamma), _ptr(dbeta), _ptr(dls)
    d.psums = _ptr(None if res['psums'] is None else res['psums'].view(torch.float64))
    ops.lib.call('aero_norm_bwd_reduce', C.byref(d), ops.stream(x))
    ops.lib.call('aero_norm_bwd_apply', C.byref(d), ops.stream(x))
    if snake_a is not None:
        return dx, dgamma, dbeta, dls, dsn
    return dx, dgamma, dbeta, dls


def istft_bwd(ops, dy, n_fft, hop, window, inv_env, T):
    """Gradient of aero_istft_fwd's input spectrogram: dy fp32 [nsig, Lout] -> dz fp32 [nsig, n_fft/2, T, 2] (interleaved complex, the
    Nyquist bin the forward treats as zero has none).  The adjoint is the forward STFT kernel on dy * inv_env (k_bwd.h)."""
    nsig, L = dy.shape
    off = n_fft // 2 + hop
    Ls = -(-(off + L + n_fft + hop) // hop) * hop
    s = torch.empty(nsig, Ls, dtype=torch.float32, device=dy.device)
    ops.lib.call('aero_istft_bwd_prep', _ptr(dy), _ptr(inv_env), _ptr(s), nsig, L, Ls, off, n_fft // 2, ops.stream(dy))
    spec = ops.stft(s, Ls, Ls, n_fft, hop, window, n_fft // 2)
    Tsrc = spec.shape[2]
    dz = torch.empty(nsig, n_fft // 2, T, 2, dtype=torch.float32, device=dy.device)
    ops.lib.call('aero_istft_bwd_pack', _ptr(spec), _ptr(dz), nsig, n_fft // 2, Tsrc, T, off // hop, ops.stream(dy))
    return dz
