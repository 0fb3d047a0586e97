"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (it imports /root/reference, which does not exist on
the GPU box).  Run as:  python -B oracle/make_golden.py
Nothing from the reference is copied: the fixtures are seeded inputs, the reference's
outputs on them, and per-tensor checksums of seed-constructed weights.
"""
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')

FULL_CFG = dict(in_channels=1, out_channels=1, channels=48, growth=2, nfft=512, hop_length=64, end_iters=0,
                cac=True, rewrite=True, hybrid=False, hybrid_old=False, freq_emb=0.2, emb_scale=10,
                emb_smooth=True, kernel_size=8, strides=[4, 4, 2, 2], context=1, context_enc=0, freq_ends=4,
                enc_freq_attn=0, norm_starts=2, norm_groups=4, dconv_mode=1, dconv_depth=2, dconv_comp=4,
                dconv_time_attn=2, dconv_lstm=2, dconv_init=1e-3, rescale=0.1, lr_sr=4000, hr_sr=16000,
                spec_upsample=True, act_func='snake', debug=False)
TINY_CFG = dict(channels=4, nfft=128, hop_length=16, lr_sr=4000, hr_sr=16000, enc_freq_attn=0)
SMALL_CFG = dict(channels=16, nfft=256, hop_length=32, lr_sr=4000, hr_sr=16000, enc_freq_attn=0)
WIDE_CFG = dict(FULL_CFG, nfft=1024, hop_length=256, lr_sr=12000, hr_sr=48000)   # BASELINE config 4 geometry


def seeded(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def checksums(sd):
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


def randomize_running_stats(model, seed):
    """BatchNorm running stats are 0/1 at init; make them non-trivial so eval-BN is exercised."""
    g = torch.Generator().manual_seed(seed)
    for name, buf in model.named_buffers():
        if name.endswith('running_mean'):
            buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
        elif name.endswith('running_var'):
            buf.copy_(0.5 + torch.rand(buf.shape, generator=g))


def main():
    sys.path.insert(0, REF)
    from src.models.aero import Aero
    from src.models.spec import spectro, ispectro
    from src.models.utils import unfold
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    # ---- op level: STFT / iSTFT geometries --------------------------------------
    ops = {}
    for tag, (nfft, hop, win, L) in {'a': (512, 16, 128, 1008), 'b': (512, 64, 512, 4032),
                                     'c': (1024, 64, 256, 2048), 'd': (128, 4, 32, 400)}.items():
        x = seeded((2, 1, L), 100 + nfft + hop)
        z = spectro(x, nfft, hop, win_length=win)
        y = ispectro(z, hop, win_length=win)
        ops[f'stft_{tag}_geom'] = np.array([nfft, hop, win, L])
        ops[f'stft_{tag}_x'] = x.numpy()
        ops[f'stft_{tag}_z'] = z.numpy()
        ops[f'stft_{tag}_y'] = y.numpy()
    a = seeded((2, 3, 251), 7)
    ops['unfold_in'] = a.numpy()
    ops['unfold_out'] = unfold(a, 200, 100).contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, 'ops.npz'), **ops)

    # ---- tiny end-to-end model, weights committed --------------------------------
    torch.manual_seed(11)
    tiny = Aero(**TINY_CFG).eval()
    randomize_running_stats(tiny, 12)
    sd = tiny.state_dict()
    np.savez_compressed(os.path.join(OUT, 'tiny_weights.npz'), **{k: v.numpy() for k, v in sd.items()})
    tiny_out = {}
    with torch.no_grad():
        for L in (400, 1000, 999):
            x = seeded((2, 1, L), 1000 + L)
            y, s, lr = tiny(x, return_spec=True, return_lr_spec=True)
            tiny_out[f'x_{L}'] = x.numpy()
            tiny_out[f'y_{L}'] = y.numpy()
            tiny_out[f'spec_{L}'] = s.numpy()
            tiny_out[f'lr_{L}'] = lr.numpy()
    np.savez_compressed(os.path.join(OUT, 'tiny_io.npz'), **tiny_out)

    meta = {'tiny_cfg': TINY_CFG, 'small_cfg': SMALL_CFG, 'full_cfg': FULL_CFG, 'wide_cfg': WIDE_CFG,
            'tiny_seed': 11, 'tiny_bn_seed': 12, 'small_seed': 21, 'small_bn_seed': 22, 'full_seed': 2036,
            'full_bn_seed': 2037, 'wide_seed': 31, 'wide_bn_seed': 32}

    # ---- small model (weights by seed + checksums) -----------------------------------------------
    torch.manual_seed(21)
    small = Aero(**SMALL_CFG).eval()
    randomize_running_stats(small, 22)
    meta['small_checksums'] = checksums(small.state_dict())
    small_out = {}
    with torch.no_grad():
        for L in (800, 2003):
            x = seeded((3, 1, L), 2000 + L)
            y, s, lr = small(x, return_spec=True, return_lr_spec=True)
            small_out[f'x_{L}'] = x.numpy()
            small_out[f'y_{L}'] = y.numpy()
            small_out[f'spec_{L}'] = s.numpy()
            small_out[f'lr_{L}'] = lr.numpy()
    np.savez_compressed(os.path.join(OUT, 'small_io.npz'), **small_out)

    # ---- full-size model: seed 2036, B=2 x 2 s white noise ------------------------
    torch.manual_seed(2036)
    full = Aero(**FULL_CFG).eval()
    meta['full_checksums_init'] = checksums(full.state_dict())
    x = seeded((1, 1, 8000), 0)
    with torch.no_grad():
        y = full(x)
    meta['full_anchor'] = {'y0_3': [float(v) for v in y[0, 0, :3]], 'sum_abs_y': float(y.abs().sum())}
    randomize_running_stats(full, 2037)
    meta['full_checksums'] = checksums(full.state_dict())
    x = seeded((2, 1, 8000), 0)
    taps = {}
    hooks = []
    for i, m in enumerate(full.encoder):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(f'enc{i}', out)))
    for j, m in enumerate(full.decoder):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, j=j: taps.__setitem__(f'dec{j}', out)))
    with torch.no_grad():
        y, s, lr = full(x, return_spec=True, return_lr_spec=True)
    for h in hooks:
        h.remove()
    full_out = {'y': y.numpy(), 'spec': s.numpy().astype(np.complex64), 'lr': lr.numpy()[:, :, ::8, ::5]}
    # NB encoder hook captures the layer output BEFORE the freq-embedding add for layer 0.
    meta['full_layer_rms'] = {k: float(v.pow(2).mean().sqrt()) for k, v in taps.items()}
    np.savez_compressed(os.path.join(OUT, 'full_io.npz'), **full_out)

    # ---- wide-band geometry (BASELINE config 4): 12->48 kHz, nfft 1024, hop 256; short clip -----
    torch.manual_seed(31)
    wide = Aero(**WIDE_CFG).eval()
    randomize_running_stats(wide, 32)
    meta['wide_checksums'] = checksums(wide.state_dict())
    x = seeded((1, 1, 6000), 31)
    with torch.no_grad():
        y, s = wide(x, return_spec=True)
    np.savez_compressed(os.path.join(OUT, 'wide_io.npz'), y=y.numpy(), spec=s.numpy().astype(np.complex64))

    with open(os.path.join(OUT, 'meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print('golden vectors written to', os.path.abspath(OUT))
    for fn in sorted(os.listdir(OUT)):
        print(f'  {fn}: {os.path.getsize(os.path.join(OUT, fn)) / 1024:.0f} KiB')


if __name__ == '__main__':
    main()
