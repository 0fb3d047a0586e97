"""CPU: caller-side rows of the hot-path table: predict.py chunking / concat /
write normalisation, match_signal, wav I/O, config loader.  Integer paths are checked exactly."""
import os

import numpy as np
import pytest
import torch

from aero_amd import _lib, audio_io, enhance
from aero_amd.config import load_config
from aero_amd.engine import HipEngine
from conftest import ROOT, build_model
from oracle import aero_oracle as O


@pytest.mark.parametrize('n', [8000, 40000, 40001, 85000, 1])
def test_chunk_ranges_exact(n):
    r = enhance.chunk_ranges(n, 4000)
    assert r == O.predict_chunks(n, 4000)
    assert r[0][0] == 0 and r[-1][1] == n and all(a2 == b1 for (_, b1), (a2, _) in zip(r, r[1:]))
    assert all(b - a == 40000 for a, b in r[:-1])


def test_integer_index_paths():
    assert O.derive_geometry(512, 64, 4000, 16000) == (4.0, 16, 128)
    assert O.derive_geometry(512, 64, 8000, 24000) == (3.0, 21, 170)         # non-integer-friendly scale (Appendix C)
    assert O.spec_pad_amount(8000, 16) == 0 and O.spec_pad_amount(7999, 16) == 1
    assert O.unfold_geometry(501, 200, 100) == (6, 700)
    sm = O.stitch_map(501, 200, 100, 6)
    assert len(sm) == 501 and sm[0] == (0, 0) and sm[149] == (0, 149) and sm[150] == (1, 50) and sm[500] == (4, 100)
    assert O.output_length(7999, 4.0) == 31996


def test_wav_roundtrip_and_write_normalisation(tmp_path):
    w = torch.randn(2, 1000) * 0.3
    p = str(tmp_path / 'a.wav')
    audio_io.save(p, w, 16000)
    r, sr = audio_io.load(p)
    assert sr == 16000 and torch.equal(r, w)
    loud = torch.tensor([[0.5, -2.0, 1.0]])
    enhance.write(loud, p, 16000)                       # enhance.py:20: divide by max(|w|max, 1)
    r, _ = audio_io.load(p)
    assert torch.allclose(r, loud / 2.0)
    quiet = torch.tensor([[0.5, -0.25]])
    enhance.write(quiet, p, 16000)
    assert torch.equal(audio_io.load(p)[0], quiet)      # not amplified


def test_match_signal():
    x = torch.arange(10.)[None]
    assert enhance.match_signal(x, 12).shape[-1] == 12 and enhance.match_signal(x, 7).shape[-1] == 7
    assert torch.equal(enhance.match_signal(x, 12)[..., :10], x)


def test_config_loader_resolves_the_experiment_group():
    c = load_config(os.path.join(ROOT, 'conf'), ['experiment=aero_4-16_512_64', 'dset=4-16', '+filename=a.wav', 'lr=1e-4'])
    assert c.experiment.name == 'aero-nfft=512-hl=64' and c.experiment.aero.nfft == 512 and c.filename == 'a.wav'
    assert c.experiment.aero.dconv_init == 1e-3 and c.lr == 1e-4 and c.ddp_backend == 'nccl'
    from aero_amd import Aero
    m = Aero(**c.experiment.aero)
    assert (m.scale, m.hop_length, m.win_length) == (4.0, 16, 128) and len(m.state_dict()) == 331
    c4 = load_config(os.path.join(ROOT, 'conf'), ['experiment=aero_12-48_1024_256'])
    assert (c4.experiment.aero.lr_sr, c4.experiment.aero.hr_sr, c4.experiment.aero.nfft) == (12000, 48000, 1024)


def test_predict_signal_equals_per_chunk_forward(meta):
    """predict.py:61-85 on the emulated kernels: batched full chunks + short tail == chunk-by-chunk forward."""
    from emu.build_emu import build
    m = build_model(meta, 'tiny')
    object.__setattr__(m, '_engine', HipEngine(m, lib=_lib.load(build())))
    sr = 40                                              # "10 s" chunk = 400 samples keeps the emulator fast
    sig = torch.randn(1, 2 * 400 + 123, generator=torch.Generator().manual_seed(8))
    pr = enhance.predict_signal(m, sig, sr, device='cpu')
    assert pr.shape == (1, 4 * sig.shape[-1])
    with torch.no_grad():
        ref = torch.cat([m(sig[:, a:b].unsqueeze(0)).squeeze(0) for a, b in enhance.chunk_ranges(sig.shape[-1], sr)], -1)
    assert torch.equal(pr, ref)


def test_src_import_paths_and_checkpoint_format(tmp_path, meta):
    """Reference checkpoints pickle the class path src.models.aero.Aero (model_serializer.py:22)."""
    import pickle
    from src.models.aero import Aero
    from src.models.modelFactory import get_model
    c = load_config(os.path.join(ROOT, 'conf'), ['experiment=aero_4-16_512_64'])
    c.experiment.aero.channels = 4
    c.experiment.aero.nfft = 128
    c.experiment.aero.hop_length = 16
    g = get_model(c)['generator']
    assert isinstance(g, Aero) and g._init_args_kwargs[1]['channels'] == 4
    blob = pickle.dumps({'class': g.__class__, 'args': g._init_args_kwargs[0], 'kwargs': g._init_args_kwargs[1]})
    back = pickle.loads(blob)
    assert back['class'] is Aero
    pkg = {'models': {'generator': {'class': g.__class__, 'args': (), 'kwargs': dict(g._init_args_kwargs[1]),
                                    'state': g.state_dict()}}}
    p = str(tmp_path / 'checkpoint.th')
    torch.save(pkg, p)
    c.checkpoint_file = p
    m2 = enhance.load_generator(c, device='cpu')
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), g.state_dict().values()))


def test_reference_shaped_checkpoint_with_training_only_classes(tmp_path, meta):
    """A package as the reference's serializer writes it (model_serializer.py:19-48): generator AND critic entries whose
    `class` fields are class objects of modules this repo does not ship, plus an omegaconf DictConfig under `args`.
    load_generator must read the generator state out of it (the classes of the other entries become inert stubs)."""
    import pickle
    import sys
    import types
    from src.models.aero import Aero
    c = load_config(os.path.join(ROOT, 'conf'), ['experiment=aero_4-16_512_64'])
    c.experiment.aero.update(channels=4, nfft=128, hop_length=16)
    g = Aero(**c.experiment.aero)
    # fabricate the foreign classes only while pickling, then remove them again (as on a box without the reference)
    fake = {}
    for modname, clsname in (('src.models.discriminators', 'Discriminator'), ('omegaconf.dictconfig', 'DictConfig'),
                             ('omegaconf.base', 'ContainerMetadata')):
        mod = types.ModuleType(modname)
        cls = type(clsname, (), {'__module__': modname, '__init__': lambda self, **kw: self.__dict__.update(kw)})
        setattr(mod, clsname, cls)
        fake[modname] = (mod, cls)
    saved = {k: sys.modules.get(k) for k in list(fake) + ['omegaconf']}
    sys.modules.update({k: v[0] for k, v in fake.items()})
    sys.modules['omegaconf'] = types.ModuleType('omegaconf')
    try:
        Disc, DictConfig, Meta = (fake[k][1] for k in fake)
        pkg = {'models': {'generator': {'class': g.__class__, 'args': (), 'kwargs': dict(g._init_args_kwargs[1]), 'state': g.state_dict()},
                          'msd_melgan': {'class': Disc, 'args': (), 'kwargs': {}, 'state': {'w': torch.ones(3)}}},
               'optimizers': {'optimizer': {'state': {}, 'param_groups': []}}, 'history': [{'train': 1.0}],
               'best_states': {'generator': g.state_dict()},
               'args': DictConfig(_metadata=Meta(flags={'x': 1}), _content={'lr': 3e-4})}
        p = str(tmp_path / 'checkpoint.th')
        torch.save(pkg, p)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    with pytest.raises((ModuleNotFoundError, AttributeError, pickle.UnpicklingError)):
        torch.load(p, map_location='cpu', weights_only=False)        # what a plain load does without the reference's modules
    c.checkpoint_file = p
    m2 = enhance.load_generator(c, device='cpu')
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), g.state_dict().values()))
    c.continue_best = True
    m3 = enhance.load_generator(c, device='cpu')
    assert all(torch.equal(a, b) for a, b in zip(m3.state_dict().values(), g.state_dict().values()))
    pkg2 = enhance.load_package(p)
    assert type(pkg2['models']['msd_melgan']['class']).__name__ == 'type' and pkg2['history'] == [{'train': 1.0}]


def test_missing_checkpoint_is_an_error_unless_random_init_is_requested(tmp_path):
    c = load_config(os.path.join(ROOT, 'conf'), ['experiment=aero_4-16_512_64'])
    c.experiment.aero.update(channels=4, nfft=128, hop_length=16)
    c.checkpoint_file = str(tmp_path / 'nope.th')
    with pytest.raises(FileNotFoundError):
        enhance.load_generator(c, device='cpu')
    c.random_init = True
    assert len(enhance.load_generator(c, device='cpu').state_dict()) > 0


def test_predict_signal_bounds_the_clips_per_forward(meta):
    """Long files are processed in groups of at most `max_clips` chunk-channels; the result does not depend on the bound."""
    from emu.build_emu import build
    m = build_model(meta, 'tiny')
    object.__setattr__(m, '_engine', HipEngine(m, lib=_lib.load(build())))
    sizes = []
    fwd = m.forward
    object.__setattr__(m, 'forward', lambda x, *a, **k: (sizes.append(x.shape[0]), fwd(x, *a, **k))[1])
    sr = 40
    sig = torch.randn(2, 5 * 400 + 123, generator=torch.Generator().manual_seed(9))           # stereo, 5 full chunks + tail
    pr = enhance.predict_signal(m, sig, sr, device='cpu', max_clips=4)
    assert sizes == [4, 4, 2, 2]                                                                # 2+2+1 chunks x 2 channels, tail
    sizes.clear()
    pr1 = enhance.predict_signal(m, sig, sr, device='cpu', batch_chunks=False)
    assert sizes == [2] * 6 and torch.equal(pr, pr1)


def test_lsd_matches_an_independent_numpy_restatement():
    """metrics.py:58-70 (STFT 2048/512, periodic hann, reflect centre padding): numpy restatement vs aero_amd.evaluate."""
    import numpy as np
    from aero_amd import evaluate as ev
    g = torch.Generator().manual_seed(3)
    ref = torch.randn(2, 9000, generator=g)
    est = ref + 0.3 * torch.randn(2, 9000, generator=g)
    assert float(ev.lsd(ref, ref)) == 0.0

    def mag2(x):
        nfft, hop = 2048, 512
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(nfft) / nfft)
        xp = np.pad(x, (nfft // 2, nfft // 2), mode='reflect')
        frames = np.stack([xp[i:i + nfft] * w for i in range(0, len(xp) - nfft + 1, hop)], 1)      # [nfft, TT]
        return np.abs(np.fft.rfft(frames, axis=0)) ** 2
    vals = []
    for b in range(2):
        sp = np.log10(np.maximum(mag2(ref[b].double().numpy()), 1e-8))
        st = np.log10(np.maximum(mag2(est[b].double().numpy()), 1e-8))
        vals.append(np.sqrt(((sp - st) ** 2).mean(0)))
    want = float(np.mean(np.stack(vals)))
    assert abs(float(ev.lsd(ref, est)) - want) < 2e-4 * want


def test_evaluate_lr_hr_contract_with_a_stub_generator():
    """evaluate.py:62-67: one forward with both flags, pr length-matched to hr, hr_spec from _spec(scale=True)."""
    from aero_amd import evaluate as ev

    class Stub(torch.nn.Module):
        scale = 4
        calls = []

        def forward(self, mix, return_spec=False, return_lr_spec=False):
            self.calls.append((return_spec, return_lr_spec, self.training))
            y = mix.repeat_interleave(4, -1)[..., :-3]                      # 3 samples short on purpose
            return y, torch.ones(1, 1, 4, 5, dtype=torch.complex64), torch.zeros(1, 1, 4, 5, dtype=torch.complex64)

        def _spec(self, x, scale=False):
            return torch.full((1, 1, 4, 7), 2.0 if scale else 1.0)
    m = Stub().train()
    lr, hr = torch.randn(1, 1, 1000), torch.randn(1, 1, 4000)
    out = ev.evaluate_lr_hr(m, lr, hr)
    assert Stub.calls == [(True, True, False)] and m.training
    assert out['pr'].shape == hr.shape and float(out['pr'][..., -3:].abs().sum()) == 0.0
    assert float(out['hr_spec'].mean()) == 2.0 and out['lr_spec'].dtype == torch.complex64
    total, count, per_file = ev.evaluate(m, [(lr[0], hr[0])] * 3, device='cpu', rank=1, world_size=2)
    assert count == 1 and len(per_file) == 1 and total == per_file[0] > 0


def test_test_py_pair_listing(tmp_path):
    """test.py pairs lr/hr files by stem (datasets.py:24-37), from json listings or from two folders."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('aero_test_cli', os.path.join(ROOT, 'test.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    (tmp_path / 'lr').mkdir()
    (tmp_path / 'hr').mkdir()
    for n in ('b', 'a', 'c'):
        (tmp_path / 'hr' / f'{n}.wav').write_bytes(b'')
    for n in ('c', 'a'):
        (tmp_path / 'lr' / f'{n}.wav').write_bytes(b'')

    class A(dict):
        __getattr__ = dict.get
    pairs = mod._listing(A(lr_dir=str(tmp_path / 'lr'), hr_dir=str(tmp_path / 'hr')))
    assert [os.path.basename(h) for _, h in pairs] == ['a.wav', 'c.wav'] and all(os.path.basename(l) == os.path.basename(h) for l, h in pairs)
    json.dump([[str(tmp_path / 'lr' / 'a.wav'), 10]], open(tmp_path / 'lr.json', 'w'))
    json.dump([[str(tmp_path / 'hr' / 'a.wav'), 40], [str(tmp_path / 'hr' / 'b.wav'), 40]], open(tmp_path / 'hr.json', 'w'))
    pairs = mod._listing(A(dset=A(test=str(tmp_path))))
    assert len(pairs) == 1 and pairs[0][1].endswith('hr/a.wav')


def test_resample_restatement_properties():
    """`experiment.upsample` (predict.py:55-57): torchaudio.functional.resample restated in aero_amd.audio_io (torchaudio is absent from the
    image, so the restatement is pinned by its defining properties only): identity at equal rates, torchaudio's length rule
    ceil(new * n / orig), a band-limited tone reproduced on the new grid, linearity"""
    import math
    from aero_amd import audio_io
    x = torch.randn(2, 4001, generator=torch.Generator().manual_seed(3))
    assert audio_io.resample(x, 4000, 4000) is x
    for new in (16000, 11025, 3000):
        y = audio_io.resample(x, 4000, new)
        assert y.shape == (2, math.ceil(new * 4001 / 4000)) and y.dtype == torch.float32
    t = torch.arange(4000) / 4000.0
    tone = torch.sin(2 * math.pi * 440 * t)[None]
    up = audio_io.resample(tone, 4000, 16000)
    t2 = torch.arange(up.shape[-1]) / 16000.0
    assert float((up - torch.sin(2 * math.pi * 440 * t2)[None])[:, 200:-200].abs().max()) < 2e-3
    a, b = x[:1], x[1:]
    assert torch.allclose(audio_io.resample(a + 2 * b, 4000, 16000), audio_io.resample(a, 4000, 16000) + 2 * audio_io.resample(b, 4000, 16000), atol=1e-5)


def test_resample_against_the_published_interpolation_formula():
    """Pins `audio_io.resample` to torchaudio's published algorithm (torchaudio.functional.resample, `sinc_interp_hann`, defaults
    lowpass_filter_width = 6, rolloff = 0.99) evaluated DIRECTLY, sample by sample, in float64:
        y[n] = sum_k x[k] h(k / orig - n / new),   h(t) = (base / orig) sinc(pi base t) cos^2(pi base t / (2 lpw))  for |base t| < lpw, else 0,
        base = rolloff * min(orig, new),  orig / new = the rates divided by their gcd,  len(y) = ceil(new len(x) / orig)
    -- no polyphase kernel bank, no padding, no strided convolution: a different evaluation of the same definition.  Three ratios (1:4 as in
    the aero configs, 4:1, and the non-integer 2:3), 32 input samples each."""
    import math
    from aero_amd import audio_io
    lpw, rolloff = 6, 0.99
    for orig_f, new_f in ((4000, 16000), (44100, 11025), (8000, 12000)):
        g = math.gcd(orig_f, new_f)
        orig, new = orig_f // g, new_f // g
        base = rolloff * min(orig, new)
        x = torch.randn(1, 32, generator=torch.Generator().manual_seed(orig_f + new_f), dtype=torch.float64)
        n_out = math.ceil(new * 32 / orig)
        ref = []
        for n in range(n_out):
            acc = 0.0
            for k in range(32):
                bt = base * (k / orig - n / new)
                if abs(bt) >= lpw:
                    continue
                sinc = 1.0 if bt == 0 else math.sin(math.pi * bt) / (math.pi * bt)
                acc += float(x[0, k]) * (base / orig) * sinc * math.cos(math.pi * bt / (2 * lpw)) ** 2
            ref.append(acc)
        y = audio_io.resample(x.float(), orig_f, new_f)
        assert y.shape == (1, n_out)
        err = float((y[0].double() - torch.tensor(ref, dtype=torch.float64)).abs().max())
        assert err < 2e-6, (orig_f, new_f, err)
