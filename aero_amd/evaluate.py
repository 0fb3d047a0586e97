"""Evaluation helpers mirroring the reference's src/evaluate.py and src/metrics.py (host-side callers of Aero.forward;
the metrics themselves are small CPU computations outside the hot path).

    evaluate_lr_hr  <- evaluate.py:54-70   (pr, pr_spec, lr_spec from ONE forward; hr_spec = model._spec(hr, scale=True))
    lsd             <- metrics.py:36-70    (log-spectral distance, STFT 2048/512, hann, centre-padded)
    evaluate        <- evaluate.py:136-170 (mean LSD over (lr, hr) pairs; rank-sharded like distrib.loader)
ViSQOL needs Google's external binary (metrics.py:73-121) and is reported as 0, exactly as the reference does when
`visqol_path` is not configured (metrics.py:28,32).
"""
import torch

from .enhance import match_signal


def stft_mag(x, nfft=2048, hop=512):
    """metrics.py:36-55: |STFT| with a periodic hann window of nfft, centre (reflect) padding, one-sided -> [B, F, TT]."""
    window = torch.hann_window(nfft, dtype=x.dtype, device=x.device)
    z = torch.stft(x, nfft, hop, window=window, return_complex=True)
    return z.abs()


def lsd(ref_sig, out_sig):
    """metrics.py:58-70: mean over frames of sqrt(mean over FREQUENCY of (log10|R|^2 - log10|O|^2)^2); inputs [B, T].
    Signals on the MI355X go through the HIP STFT (aero_amd.losses.lsd); host tensors (a caller that already moved its
    results off the device, as run_metrics' reference does) are measured with torch.stft on the host."""
    if ref_sig.is_cuda and out_sig.is_cuda:
        from .losses import lsd as lsd_device
        return lsd_device(ref_sig, out_sig)
    sp = torch.log10(stft_mag(ref_sig).square().clamp(1e-8))
    st = torch.log10(stft_mag(out_sig).square().clamp(1e-8))
    return (sp - st).square().mean(dim=1).sqrt().mean()


def evaluate_lr_hr(model, lr, hr):
    """evaluate.py:62-67 for the aero generator: lr [B,1,L] and hr [B,1,L*scale] on the model's device.
    Returns dict(pr, pr_spec, lr_spec, hr_spec); pr is length-matched to hr."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        pr, pr_spec, lr_spec = model(lr, return_spec=True, return_lr_spec=True)
        pr = match_signal(pr, hr.shape[-1])
        hr_spec = model._spec(hr, scale=True)
    if was_training:
        model.train()
    return dict(pr=pr, pr_spec=pr_spec, lr_spec=lr_spec, hr_spec=hr_spec)


def run_metrics(hr, pr):
    """metrics.py:20-33 without the external ViSQOL binary: (lsd, visqol=0); hr, pr are [B,1,T] CPU tensors."""
    if hr.is_cuda and pr.is_cuda:
        return lsd(hr.squeeze(1).float(), pr.squeeze(1).float()).item(), 0         # HIP STFT, tensors stay on the device
    return lsd(hr.squeeze(1).float().cpu(), pr.squeeze(1).float().cpu()).item(), 0


def evaluate(model, pairs, device='cuda', rank=0, world_size=1):
    """Mean LSD over an iterable of (lr [1,L] or [B,1,L], hr) pairs; pair i is handled by rank i mod world_size
    (distrib.py:100), the partial sums are combined by the caller with distrib.average (evaluate.py:160-166).
    Returns (sum of per-file LSD, number of files, per-file list)."""
    total, count, per_file = 0.0, 0, []
    for i, (lr, hr) in enumerate(pairs):
        if i % world_size != rank:
            continue
        lr = lr if lr.dim() == 3 else lr.unsqueeze(0)
        hr = hr if hr.dim() == 3 else hr.unsqueeze(0)
        hr = hr.to(device)
        out = evaluate_lr_hr(model, lr.to(device), hr)
        lsd_i, _ = run_metrics(hr, out['pr'])
        total += lsd_i
        count += 1
        per_file.append(lsd_i)
    return total, count, per_file
