"""Spectral losses / metrics computed on the MI355X STFT kernel (forward values).

    stft_magnitude               <- stft_loss.py:11-27   |STFT| with clamp 1e-7, shape [B, frames, fft/2+1]
    STFTLoss / MultiResolutionSTFTLoss  <- stft_loss.py:84-138 (1024/120/600, 2048/240/1200, 512/50/240; hann)
    lsd                          <- metrics.py:36-70     log-spectral distance, STFT 2048/512

The transforms run through `aero_stft_fwd` (aero_amd/csrc/k_stft.h: n_fft up to 2048, any hop, any window length
zero-padded to n_fft exactly as torch.stft does); the reductions on the magnitude tensors are a few device-side
reductions.  The reference's `torch.stft(..., return_complex unset)` raises on torch >= 2; these restate it.
`STFTLoss` / `MultiResolutionSTFTLoss` are differentiable w.r.t. the predicted signal: their backward runs on the kernels of
csrc/k_train.h (aero_stft_loss_bwd, aero_irfft_frames, aero_stft_adj_fold), so they are the training criterion of solver.py:560-584.
"""
import torch

from . import _lib
from .engine import Ops, _hann_padded

_ops = None
_windows = {}


def _get_ops():
    global _ops
    if _ops is None:
        _ops = Ops(_lib.load())
    return _ops


def use_library(lib):
    """tests: run the losses on an explicitly loaded library (the CPU-emulated test double); product code never calls this"""
    global _ops
    _ops = Ops(lib) if lib is not None else None


def _window(win_length, n_fft, device):
    key = (win_length, n_fft, str(device))
    if key not in _windows:
        _windows[key] = _hann_padded(win_length, n_fft, device)
    return _windows[key]


def stft_power(x, fft_size, hop_size, win_length):
    """|STFT|^2 of x [B, T] (float32 on the MI355X), un-normalised like torch.stft's default: [B, fft/2+1, frames].
    The kernel computes the `normalized=True` transform (aero.py's convention); the factor n_fft restores the scale."""
    ops = _get_ops()
    if not x.is_cuda and not ops.lib.is_emulator:
        raise RuntimeError('aero_amd.losses runs on the MI355X: move the signals to "cuda"')
    x = x.contiguous().float()
    B, L = x.shape
    z = ops.stft(x, L, L, fft_size, hop_size, _window(win_length, fft_size, x.device), fft_size // 2 + 1)
    z = z[:, :, :1 + L // hop_size]                      # torch.stft(center=True): 1 + L // hop frames
    return (z[..., 0].square() + z[..., 1].square()) * float(fft_size)


def stft_magnitude(x, fft_size, hop_size, win_length):
    """stft_loss.py:11-27: sqrt(clamp(re^2 + im^2, 1e-7)) -> [B, frames, fft/2+1]."""
    return torch.sqrt(torch.clamp(stft_power(x, fft_size, hop_size, win_length), min=1e-7)).transpose(2, 1)


class _STFTLossFn(torch.autograd.Function):
    """(spectral convergence, log-magnitude L1) of one resolution with its gradient w.r.t. the predicted signal, all on the HIP
    kernels: aero_stft_fwd -> aero_stft_loss_sums; backward aero_stft_loss_bwd -> aero_irfft_frames + aero_stft_adj_fold (k_train.h)."""

    @staticmethod
    def forward(ctx, x, y, fft_size, hop, win):
        from . import train_ops as TO
        ops = _get_ops()
        if not x.is_cuda and not ops.lib.is_emulator:
            raise RuntimeError('aero_amd.losses runs on the MI355X: move the signals to "cuda"')
        x, y = x.detach().contiguous().float(), y.detach().contiguous().float()
        B, L = x.shape
        wpad = _window(win, fft_size, x.device)
        zx = ops.stft(x, L, L, fft_size, hop, wpad, fft_size // 2 + 1)
        zy = ops.stft(y, L, L, fft_size, hop, wpad, fft_size // 2 + 1)
        sums = TO.stft_loss_sums(ops, zx, zy, float(fft_size))
        ctx.save_for_backward(zx, zy, sums)
        ctx.geom = (fft_size, hop, win, L)
        sc = (sums[0] / sums[1]).sqrt().float()                  # three device scalars: stft_loss.py:47,64
        mag = (sums[2] / (zx.numel() // 2)).float()
        return sc, mag

    @staticmethod
    def backward(ctx, gsc, gmag):
        from . import train_ops as TO
        ops = _get_ops()
        zx, zy, sums = ctx.saved_tensors
        fft_size, hop, win, L = ctx.geom
        gout = torch.stack([gsc.reshape(()), gmag.reshape(())]).float().contiguous()
        g = TO.stft_loss_bwd(ops, zx, zy, float(fft_size), sums, 1.0, 1.0, gout)
        dx = TO.stft_adjoint(ops, g, fft_size, hop, _window(win, fft_size, zx.device), L)
        return dx, None, None, None, None


class STFTLoss(torch.nn.Module):
    """stft_loss.py:84-117: spectral convergence and log-magnitude L1 of one resolution (differentiable w.r.t. x)."""

    def __init__(self, fft_size=1024, shift_size=120, win_length=600):
        super().__init__()
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length

    def forward(self, x, y):
        return _STFTLossFn.apply(x, y, self.fft_size, self.shift_size, self.win_length)


class MultiResolutionSTFTLoss(torch.nn.Module):
    """stft_loss.py:120-161."""

    def __init__(self, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                 factor_sc=0.1, factor_mag=0.1):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.stft_losses = torch.nn.ModuleList(STFTLoss(f, h, w) for f, h, w in zip(fft_sizes, hop_sizes, win_lengths))
        self.factor_sc, self.factor_mag = factor_sc, factor_mag

    def forward(self, x, y):
        sc = mag = 0.0
        for f in self.stft_losses:
            s, m = f(x, y)
            sc, mag = sc + s, mag + m
        n = len(self.stft_losses)
        return self.factor_sc * sc / n, self.factor_mag * mag / n


def lsd(ref_sig, out_sig):
    """metrics.py:58-70 on the device: mean over frames of sqrt(mean over frequency of (log10|R|^2 - log10|O|^2)^2);
    inputs [B, T] on the MI355X; STFT 2048/512 with a periodic hann window of 2048 (metrics.py:37-55)."""
    sp = torch.log10(stft_power(ref_sig, 2048, 512, 2048).clamp(1e-8))
    st = torch.log10(stft_power(out_sig, 2048, 512, 2048).clamp(1e-8))
    return (sp - st).square().mean(dim=1).sqrt().mean()
