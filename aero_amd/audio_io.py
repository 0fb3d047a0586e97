"""Minimal WAV I/O (host plumbing) so predict.py / test.py run without torchaudio (absent from the image).

`load(path) -> (float32 tensor [channels, samples] in [-1, 1], sample_rate)` and `save(path, wav, sr)` follow
torchaudio.load / torchaudio.save as used by the reference (predict.py:53, enhance.py:18-21).
PCM16 and IEEE float32 RIFF files are supported; torchaudio is used instead when importable.
"""
import struct

import numpy as np
import torch


def load(path):
    try:
        import torchaudio
        return torchaudio.load(str(path))
    except ImportError:
        pass
    with open(path, 'rb') as f:
        data = f.read()
    if data[:4] != b'RIFF' or data[8:12] != b'WAVE':
        raise ValueError(f'{path}: not a RIFF/WAVE file')
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack('<I', data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            fmt = struct.unpack('<HHIIHH', body[:16])
        elif cid == b'data':
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f'{path}: missing fmt/data chunk')
    tag, nch, sr, _, _, bits = fmt
    if tag == 1 and bits == 16:
        a = np.frombuffer(pcm, dtype='<i2').astype(np.float32) / 32768.0
    elif tag == 3 and bits == 32:
        a = np.frombuffer(pcm, dtype='<f4').astype(np.float32)
    else:
        raise ValueError(f'{path}: unsupported WAV encoding (tag {tag}, {bits} bit)')
    a = a[:len(a) // nch * nch].reshape(-1, nch).T
    return torch.from_numpy(np.ascontiguousarray(a)), sr


def save(path, wav, sr):
    """float32 WAV, [channels, samples]."""
    try:
        import torchaudio
        return torchaudio.save(str(path), wav.cpu(), sr)
    except ImportError:
        pass
    a = wav.detach().cpu().float().numpy()
    if a.ndim == 1:
        a = a[None]
    nch, n = a.shape
    pcm = np.ascontiguousarray(a.T).astype('<f4').tobytes()
    hdr = b'RIFF' + struct.pack('<I', 36 + len(pcm)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 3, nch, sr, sr * nch * 4, nch * 4, 32)
    with open(path, 'wb') as f:
        f.write(hdr + b'data' + struct.pack('<I', len(pcm)) + pcm)


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """`torchaudio.functional.resample(waveform, orig_freq, new_freq)` with its defaults (Hann-windowed sinc interpolation), which the
    reference calls before the model when `experiment.upsample` is set (predict.py:55-57, datasets.py:144).  Host plumbing outside the hot
    path: torchaudio is used when importable; otherwise its published algorithm is restated here (polyphase sinc kernel of
    `new_freq / gcd` phases, width ceil(lowpass_filter_width * orig / (rolloff * min(orig, new))), a strided conv1d, output cropped to
    ceil(new * length / orig) samples).  torchaudio is absent from this image; the restatement is PINNED TO THE PUBLISHED ALGORITHM: a direct
    float64 evaluation of its interpolation formula at three rate ratios (tests/test_callers.py::
    test_resample_against_the_published_interpolation_formula) next to the defining properties (identity, length rule, a tone, linearity)."""
    try:
        from torchaudio.functional import resample as ta_resample
        return ta_resample(waveform, orig_freq, new_freq)
    except ImportError:
        pass
    import math
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError('resample: frequencies must be positive integers')
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    kernels = kernels.to(torch.float32)
    shape = waveform.shape
    w = waveform.reshape(-1, shape[-1]).to(torch.float32)
    length = w.shape[-1]
    w = torch.nn.functional.pad(w, (width, width + orig))
    out = torch.nn.functional.conv1d(w[:, None], kernels, stride=orig)
    out = out.transpose(1, 2).reshape(w.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return out[..., :target].reshape(shape[:-1] + (target,))
