"""Inference helpers mirroring the reference's src/enhance.py and predict.py (host plumbing around Aero.forward)."""
import math
import os

import torch

from . import audio_io

SEGMENT_DURATION_SEC = 10            # predict.py:22


def chunk_ranges(n_samples, sr, segment_sec=SEGMENT_DURATION_SEC):
    """predict.py:61-69: independent [start, end) chunks of `segment_sec` seconds, last one short (integer-exact)."""
    seg = sr * segment_sec
    n_chunks = math.ceil(n_samples / seg)
    return [(i * seg, min((i + 1) * seg, n_samples)) for i in range(n_chunks)]


def get_estimate(model, lr_sig):
    """enhance.py:11-15."""
    with torch.no_grad():
        return model(lr_sig)


def predict_signal(model, lr_sig, sr, device=None, batch_chunks=True):
    """predict.py:61-85 for one file: lr_sig [channels, samples] -> [channels, samples*scale].

    All full-length chunks go through ONE batched forward (they are independent, predict.py:76-80 loops
    over them); the short tail chunk is run separately.  Results are identical per chunk.
    """
    device = device or next(model.parameters()).device
    ranges = chunk_ranges(lr_sig.shape[-1], sr)
    out = [None] * len(ranges)
    model.eval()
    with torch.no_grad():
        full = [i for i, (a, b) in enumerate(ranges) if b - a == sr * SEGMENT_DURATION_SEC]
        if batch_chunks and len(full) > 1:
            x = torch.stack([lr_sig[:, a:b] for a, b in (ranges[i] for i in full)], 0)        # [n, ch, L]
            n, ch, L = x.shape
            y = model(x.reshape(n * ch, 1, L).to(device)).reshape(n, ch, -1).cpu()
            for k, i in enumerate(full):
                out[i] = y[k]
        for i, (a, b) in enumerate(ranges):
            if out[i] is None:
                out[i] = model(lr_sig[:, a:b].unsqueeze(1).to(device)).squeeze(1).cpu()
    return torch.cat(out, dim=-1)


def write(wav, filename, sr):
    """enhance.py:18-21: divide by max(|wav|max, 1) only if it prevents clipping, then save."""
    wav = wav / max(wav.abs().max().item(), 1)
    audio_io.save(filename, wav.cpu(), sr)


def save_wavs(processed_sigs, lr_sigs, hr_sigs, filenames, lr_sr, hr_sr):
    """enhance.py:24-29."""
    for lr, hr, pr, filename in zip(lr_sigs, hr_sigs, processed_sigs, filenames):
        write(lr, filename + '_lr.wav', sr=lr_sr)
        write(hr, filename + '_hr.wav', sr=hr_sr)
        write(pr, filename + '_pr.wav', sr=hr_sr)


def match_signal(signal, ref_len):
    """src/utils.py:211-217."""
    sig_len = signal.shape[-1]
    if sig_len < ref_len:
        signal = torch.nn.functional.pad(signal, (0, ref_len - sig_len))
    elif sig_len > ref_len:
        signal = signal[..., :ref_len]
    return signal


def load_generator(args, device='cuda'):
    """predict.py:24-38 / test.py:25-39: build the generator from the experiment config and load a checkpoint
    written by the reference's serializer ({'models': {'generator': {'state': ...}}}, model_serializer.py:19-48)."""
    from .modules import Aero
    model = Aero(**args.experiment.aero)
    ckpt = args.get('checkpoint_file')
    if ckpt and os.path.exists(str(ckpt)):
        package = torch.load(str(ckpt), map_location='cpu', weights_only=False)
        if args.get('continue_best'):
            best = package['best_states']
            state = best['models']['generator']['state'] if 'models' in best else best['generator']
        else:
            state = package['models']['generator']['state']
        model.load_state_dict(state)
    return model.to(device).eval()
