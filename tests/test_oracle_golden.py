"""CPU: the oracle restatement against the vectors captured from the reference itself, and the
seed-reproducibility of the parameter tree.  Tolerance: 1e-6 rel-L2 (fp32 noise floor 2e-7)."""
import numpy as np
import pytest
import torch

from conftest import build_model, load_npz, rel_l2
from oracle import aero_oracle as O

TOL = 1e-6


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_stft_istft_golden(tag):
    g = load_npz('ops.npz')
    nfft, hop, win, L = [int(v) for v in g[f'stft_{tag}_geom']]
    x = torch.from_numpy(g[f'stft_{tag}_x'])
    z = O.stft(x, nfft, hop, win)
    assert z.shape == g[f'stft_{tag}_z'].shape
    assert rel_l2(z, g[f'stft_{tag}_z']) < TOL
    y = O.istft(torch.from_numpy(g[f'stft_{tag}_z']), hop, win)
    assert y.shape == g[f'stft_{tag}_y'].shape
    assert rel_l2(y, g[f'stft_{tag}_y']) < TOL


def test_unfold_index_math():
    g = load_npz('ops.npz')
    a = torch.from_numpy(g['unfold_in'])
    n, tgt = O.unfold_geometry(a.shape[-1], 200, 100)
    ap = torch.nn.functional.pad(a, (0, tgt - a.shape[-1]))
    idx = torch.arange(n)[:, None] * 100 + torch.arange(200)[None, :]
    assert np.array_equal(ap[..., idx].numpy(), g['unfold_out'])          # bit exact


def test_tiny_weights_reproduced_by_seed(meta):
    m = build_model(meta, 'tiny')
    w = load_npz('tiny_weights.npz')
    sd = m.state_dict()
    assert set(sd) == set(w)
    for k in w:
        assert np.array_equal(sd[k].numpy(), w[k]), k


@pytest.mark.parametrize('which', ['small', 'full', 'wide'])
def test_weight_checksums(meta, which):
    m = build_model(meta, which)
    ref = meta[f'{which}_checksums']
    sd = m.state_dict()
    assert list(sd) == list(ref)                                           # same keys, same order
    if which == 'full':
        assert len(sd) == 331
    for k, v in sd.items():
        assert float(v.double().sum()) == ref[k][0] and float(v.double().abs().sum()) == ref[k][1], k


@pytest.mark.parametrize('L', [400, 1000, 999])
def test_tiny_forward_golden(meta, L):
    io = load_npz('tiny_io.npz')
    sd = {k: torch.from_numpy(v) for k, v in load_npz('tiny_weights.npz').items()}
    y, s, lr = O.aero_forward(sd, meta['tiny_cfg'], torch.from_numpy(io[f'x_{L}']), True, True)
    assert y.shape == io[f'y_{L}'].shape and y.shape[-1] == 4 * L
    assert rel_l2(lr, io[f'lr_{L}']) < TOL
    assert rel_l2(s, io[f'spec_{L}']) < TOL
    assert rel_l2(y, io[f'y_{L}']) < TOL


def test_tiny_fast_lstm_matches_explicit(meta):
    io = load_npz('tiny_io.npz')
    sd = {k: torch.from_numpy(v) for k, v in load_npz('tiny_weights.npz').items()}
    y = O.aero_forward(sd, meta['tiny_cfg'], torch.from_numpy(io['x_1000']), fast=True)
    assert rel_l2(y, io['y_1000']) < TOL


@pytest.mark.parametrize('L', [800, 2003])
def test_small_forward_golden(meta, L):
    io = load_npz('small_io.npz')
    sd = build_model(meta, 'small').state_dict()
    y, s, lr = O.aero_forward(sd, meta['small_cfg'], torch.from_numpy(io[f'x_{L}']), True, True, fast=True)
    assert rel_l2(s, io[f'spec_{L}']) < TOL
    assert rel_l2(y, io[f'y_{L}']) < TOL


def test_full_forward_golden(meta):
    """aero_4-16_512_64, seed 2036, 2 x 2-s white-noise clips (BASELINE config 1 shape)."""
    io = load_npz('full_io.npz')
    m = build_model(meta, 'full')
    x = torch.randn(2, 1, 8000, generator=torch.Generator().manual_seed(0))
    taps = {}
    y, s, lr = O.aero_forward(m.state_dict(), meta['full_cfg'], x, True, True, fast=True, taps=taps)
    assert y.shape == (2, 1, 32000) and s.shape == (2, 1, 256, 501)
    assert rel_l2(lr[:, :, ::8, ::5], io['lr']) < TOL
    assert rel_l2(s, io['spec']) < TOL
    assert rel_l2(y, io['y']) < TOL
    for k, rms in meta['full_layer_rms'].items():
        if k == 'enc0':
            continue            # the reference hook sees encoder 0 before the frequency-embedding add
        assert abs(float(taps[k].pow(2).mean().sqrt()) - rms) < 1e-5 * max(1.0, rms), k


def test_wide_forward_golden(meta):
    io = load_npz('wide_io.npz')
    m = build_model(meta, 'wide')
    x = torch.randn(1, 1, 6000, generator=torch.Generator().manual_seed(31))
    y, s = O.aero_forward(m.state_dict(), meta['wide_cfg'], x, True, fast=True)
    assert rel_l2(s, io['spec']) < TOL
    assert rel_l2(y, io['y']) < TOL


# ---- op-level vectors captured from the reference's own modules (tests/golden/modules.npz) -------------

def _mod(tag):
    g = load_npz('modules.npz')
    w = {k[len(tag) + 3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + '.w.')}
    i = {k[len(tag) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + '.in.')}
    o = {k[len(tag) + 5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + '.out.')}
    return w, i, o


@pytest.mark.parametrize('fast', [False, True])
def test_blstm_module_golden(fast):
    """modules.py:32-65: framed (T=251 -> 3 overlapping frames of 200, stitched) and unframed (T=150)."""
    w, i, o = _mod('blstm')
    sd = {'m.' + k: v for k, v in w.items()}
    for case in ('framed', 'unframed'):
        assert rel_l2(O.blstm(sd, 'm', i[case], fast=fast), o[case]) < TOL, case


def test_localstate_module_golden():
    w, i, o = _mod('localstate')
    assert rel_l2(O.local_state({'m.' + k: v for k, v in w.items()}, 'm', i['x']), o['y']) < TOL


def test_ftb_module_golden_eval_and_train():
    """modules.py:304-325 in both BatchNorm modes.  The committed weights are the state AFTER the reference's train-mode
    call, so the eval output is reproduced from the state before it: undo the running-stat update with the batch
    statistics the oracle computes itself, then check the update rule against the committed buffers."""
    w, i, o = _mod('ftb')
    after = {'m.' + k: v for k, v in w.items()}
    before = dict(after)
    # batch statistics do not depend on the running statistics: run train mode on any state to get them
    new = {}
    yt = O.ftb(after, 'm', i['x'], train=True, new_stats=new)
    assert rel_l2(yt, o['train']) < TOL
    for k, v in new.items():
        if k.endswith('num_batches_tracked'):
            before[k] = after[k] - 1
        else:                                   # after = 0.9 before + 0.1 batch;  new = 0.9 after + 0.1 batch
            batch = (new[k] - 0.9 * after[k]) / 0.1
            before[k] = (after[k] - 0.1 * batch) / 0.9
    assert rel_l2(O.ftb(before, 'm', i['x']), o['eval']) < 5 * TOL
    again = {}
    O.ftb(before, 'm', i['x'], train=True, new_stats=again)
    for k in again:
        assert torch.allclose(again[k].double(), after[k].double(), rtol=1e-5, atol=1e-6), k


def test_snake_module_golden():
    w, i, o = _mod('snake')
    assert rel_l2(O.snake(i['x'], w['a']), o['y']) < TOL


def test_dconv_module_golden():
    """modules.py:221-249 with Snake, BLSTM and LocalState enabled, LayerScale 0.5 (the branch is NOT scaled away)."""
    w, i, o = _mod('dconv')
    y = O.dconv({'m.' + k: v for k, v in w.items()}, 'm', i['x'], 2, True, True)
    assert rel_l2(y, o['y']) < TOL


def test_henc_hdec_layer_golden():
    """aero.py:108-135 and aero.py:189-215 at tiny channel counts (strided conv + GroupNorm + GELU + rewrite/GLU;
    3x3 rewrite over cat[x, skip] + GLU + transposed conv + norm-before-trim + GELU)."""
    w, i, o = _mod('henc')
    cfg = dict(strides=[4], norm_groups=4, dconv_mode=0, context_enc=0, context=1, dconv_depth=2, dconv_lstm=9, dconv_time_attn=9)
    y = O.enc_layer({'encoder.0.' + k: v for k, v in w.items()}, 0, i['x'], cfg)
    assert y.shape == o['y'].shape and rel_l2(y, o['y']) < TOL
    w, i, o = _mod('hdec')
    z = O.dec_layer({'decoder.0.' + k: v for k, v in w.items()}, 0, i['x'], i['skip'], cfg, last=False)
    assert z.shape == o['y'].shape and rel_l2(z, o['y']) < TOL


# ---- "trained-like" stress models, config 4 at full length, train mode ----------------------------------------------

@pytest.mark.parametrize('which', ['stress_small', 'stress_full'])
def test_stress_weights_reproduced(meta, which):
    sd = build_model(meta, which).state_dict()
    ref = meta[f'{which}_checksums']
    assert list(sd) == list(ref)
    for k, v in sd.items():
        assert float(v.double().sum()) == ref[k][0] and float(v.double().abs().sum()) == ref[k][1], k


@pytest.mark.parametrize('L', [800, 2003])
def test_stress_small_forward_golden(meta, L):
    io = load_npz('stress_small_io.npz')
    sd = build_model(meta, 'stress_small').state_dict()
    x = torch.from_numpy(load_npz('small_io.npz')[f'x_{L}'])
    y, s = O.aero_forward(sd, meta['small_cfg'], x, True, fast=True)
    assert rel_l2(s, io[f'spec_{L}']) < TOL and rel_l2(y, io[f'y_{L}']) < TOL


def test_stress_full_forward_golden(meta):
    io = load_npz('stress_full_io.npz')
    m = build_model(meta, 'stress_full')
    x = torch.randn(2, 1, 8000, generator=torch.Generator().manual_seed(0))
    taps = {}
    y, s = O.aero_forward(m.state_dict(), meta['full_cfg'], x, True, fast=True, taps=taps)
    assert rel_l2(s, io['spec']) < TOL and rel_l2(y, io['y']) < TOL
    for k in ('enc1', 'enc2', 'enc3', 'dec0', 'dec1', 'dec2', 'dec3'):
        assert rel_l2(taps[k][:, ::7, :, ::9], io[k]) < 2 * TOL, k
    # the perturbation did what it is for: the DConv branch now matters (vs 1e-3 LayerScale at init)
    base = O.aero_forward(build_model(meta, 'full').state_dict(), meta['full_cfg'], x[:1], True, fast=True)[1]
    assert rel_l2(s[:1], base) > 0.1


def test_wide_full_length_first_clip_golden(meta):
    """BASELINE config 4 at its full clip length (24000 samples = 2 s at 12 kHz): first clip of the [32,1,24000] input."""
    io = load_npz('wide_full_io.npz')
    m = build_model(meta, 'wide')
    x = torch.randn(32, 1, 24000, generator=torch.Generator().manual_seed(meta['wide_full_input_seed']))[:1]
    y, s = O.aero_forward(m.state_dict(), meta['wide_cfg'], x, True, fast=True)
    assert y.shape == (1, 1, 96000) and s.shape == (1, 1, 512, 376)
    assert rel_l2(s[:, :, ::4, ::3], io['spec']) < TOL and rel_l2(y[..., ::4], io['y']) < TOL


def test_train_mode_forward_golden(meta):
    """Aero.forward in training mode (the FTB's BatchNorms on batch statistics, running statistics updated)."""
    io = load_npz('train_tiny_io.npz')
    m = build_model(meta, 'tiny')
    new = {}
    y, s = O.aero_forward(m.state_dict(), meta['tiny_cfg'], torch.from_numpy(io['x']), True, train=True, new_stats=new)
    assert rel_l2(s, io['spec']) < TOL and rel_l2(y, io['y']) < 3 * TOL      # (waveform: fp32 cancellation in the overlap-add, measured 1.1e-6)
    bufs = {k[4:]: v for k, v in io.items() if k.startswith('buf.')}
    assert set(new) == set(bufs) and len(new) == 4 * 3 * 3
    for k, v in new.items():
        assert torch.allclose(v.double(), torch.from_numpy(bufs[k]).double(), rtol=2e-5, atol=1e-6), k
