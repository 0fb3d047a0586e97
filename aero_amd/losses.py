"""Spectral losses / metrics computed on the MI355X STFT kernel (forward values).

    stft_magnitude               <- stft_loss.py:11-27   |STFT| with clamp 1e-7, shape [B, frames, fft/2+1]
    STFTLoss / MultiResolutionSTFTLoss  <- stft_loss.py:84-138 (1024/120/600, 2048/240/1200, 512/50/240; hann)
    lsd                          <- metrics.py:36-70     log-spectral distance, STFT 2048/512

The transforms run through `aero_stft_fwd` (aero_amd/csrc/k_stft.h: n_fft up to 2048, any hop, any window length
zero-padded to n_fft exactly as torch.stft does); the reductions on the magnitude tensors are a few device-side
reductions.  The reference's `torch.stft(..., return_complex unset)` raises on torch >= 2; these restate it.
No autograd: there are no backward kernels, so these are metric / validation values, not a trainable criterion.
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


def _window(win_length, n_fft, device):
    key = (win_length, n_fft, str(device))
    if key not in _windows:
        _windows[key] = _hann_padded(win_length, n_fft, device)
    return _windows[key]


def stft_power(x, fft_size, hop_size, win_length):
    """|STFT|^2 of x [B, T] (float32 on the MI355X), un-normalised like torch.stft's default: [B, fft/2+1, frames].
    The kernel computes the `normalized=True` transform (aero.py's convention); the factor n_fft restores the scale."""
    if not x.is_cuda:
        raise RuntimeError('aero_amd.losses runs on the MI355X: move the signals to "cuda"')
    x = x.contiguous().float()
    B, L = x.shape
    ops = _get_ops()
    z = ops.stft(x, L, L, fft_size, hop_size, _window(win_length, fft_size, x.device), fft_size // 2 + 1)
    z = z[:, :, :1 + L // hop_size]                      # torch.stft(center=True): 1 + L // hop frames
    return (z[..., 0].square() + z[..., 1].square()) * float(fft_size)


def stft_magnitude(x, fft_size, hop_size, win_length):
    """stft_loss.py:11-27: sqrt(clamp(re^2 + im^2, 1e-7)) -> [B, frames, fft/2+1]."""
    return torch.sqrt(torch.clamp(stft_power(x, fft_size, hop_size, win_length), min=1e-7)).transpose(2, 1)


class STFTLoss(torch.nn.Module):
    """stft_loss.py:84-117: spectral convergence and log-magnitude L1 of one resolution."""

    def __init__(self, fft_size=1024, shift_size=120, win_length=600):
        super().__init__()
        self.fft_size, self.shift_size, self.win_length = fft_size, shift_size, win_length

    def forward(self, x, y):
        with torch.no_grad():
            x_mag = stft_magnitude(x, self.fft_size, self.shift_size, self.win_length)
            y_mag = stft_magnitude(y, self.fft_size, self.shift_size, self.win_length)
            sc = torch.norm(y_mag - x_mag, p='fro') / torch.norm(y_mag, p='fro')
            mag = torch.nn.functional.l1_loss(torch.log(y_mag), torch.log(x_mag))
        return sc, mag


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
