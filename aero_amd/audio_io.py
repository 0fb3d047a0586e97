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
