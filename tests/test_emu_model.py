"""CPU: the whole Aero.forward through HipEngine with the kernels running on the CPU emulation of
HIP (test double), against the golden vectors of the reference: checks engine plumbing, weight packing
and every kernel's logic end to end without a GPU.  Tolerance 1e-3 rel-L2 on the spectrogram (north_star)."""
import numpy as np
import pytest
import torch

from aero_amd import _lib
from aero_amd.engine import HipEngine
from conftest import build_model, load_npz, rel_l2


@pytest.fixture(scope='module')
def emu():
    from emu.build_emu import build
    return _lib.load(build())


def _with_engine(model, lib):
    object.__setattr__(model, '_engine', HipEngine(model, lib=lib))
    return model


@pytest.mark.parametrize('L', [400, 1000, 999])
def test_tiny_model_golden(emu, meta, L):
    m = _with_engine(build_model(meta, 'tiny'), emu)
    io = load_npz('tiny_io.npz')
    with torch.no_grad():
        y, s, lr = m(torch.from_numpy(io[f'x_{L}']), return_spec=True, return_lr_spec=True)
    assert y.shape == io[f'y_{L}'].shape and s.dtype == torch.complex64
    assert rel_l2(lr, io[f'lr_{L}']) < 2e-6              # STFT is fp32
    assert rel_l2(s, io[f'spec_{L}']) < 1e-3
    assert rel_l2(y, io[f'y_{L}']) < 5e-3


@pytest.mark.parametrize('fuse', [(True, True), (False, True), (False, False)])
def test_small_model_golden_groupnorm_fusion_paths(emu, meta, fuse):
    """GroupNorm statistics accumulated in the conv epilogue (stat_mode 1) and the DConv tail as a recompute pair
    (stat_mode 2 + 3: conv2 -> GN -> GLU -> LayerScale -> +skip without materialising the 2C-channel tensor) versus the
    separate aero_norm_stats / aero_norm_apply kernels: all must reproduce the reference's golden output."""
    m = build_model(meta, 'small')
    eng = HipEngine(m, lib=emu)
    eng.fuse_dconv_tail, eng.fuse_stats = fuse
    object.__setattr__(m, '_engine', eng)
    io = load_npz('small_io.npz')
    with torch.no_grad():
        y, s = m(torch.from_numpy(io['x_2003']), return_spec=True)
    assert rel_l2(s, io['spec_2003']) < 1e-3
    assert rel_l2(y, io['y_2003']) < 5e-3


@pytest.mark.parametrize('collapse,fused', [(True, True), (True, False), (False, False)])
def test_small_model_golden_first_layer_ftb_paths(emu, meta, collapse, fused):
    """channels=16 model: encoder 0 runs as ONE kernel (FTB collapsed onto the 2-channel spectrogram + the strided conv,
    aero_enc0_fwd), with the collapsed FTB on its own (aero_ftb_first_fwd) or layer by layer (pre_conv + aero_freqfc_fwd +
    convs); all three must match the reference's golden output."""
    m = build_model(meta, 'small')
    eng = HipEngine(m, lib=emu)
    eng.collapse_first_ftb, eng.fuse_enc0 = collapse, fused
    names, call = [], eng.ops.lib.call
    eng.ops.lib = type('Spy', (), {'call': staticmethod(lambda fn, *a: (names.append(fn), call(fn, *a))[1]),
                                    '__getattr__': lambda self, k: getattr(emu, k)})()
    object.__setattr__(m, '_engine', eng)
    io = load_npz('small_io.npz')
    with torch.no_grad():
        y, s = m(torch.from_numpy(io['x_800']), return_spec=True)
    assert rel_l2(s, io['spec_800']) < 1e-3
    assert rel_l2(y, io['y_800']) < 5e-3
    assert ('aero_enc0_fwd' in names) == fused and ('aero_ftb_first_fwd' in names) == (collapse and not fused)


def test_spec_ispec_api(emu, meta):
    """Aero._spec / _spec(scale=True) / _ispec (used by evaluate.py:67, solver.py:374) against the oracle."""
    from oracle import aero_oracle as O
    m = _with_engine(build_model(meta, 'tiny'), emu)
    x = torch.randn(2, 1, 1001, generator=torch.Generator().manual_seed(3))
    cfg = {**O.DEFAULT_CFG, **meta['tiny_cfg']}
    assert rel_l2(m._spec(x), O.spec(x, cfg)) < 2e-6
    hr = torch.randn(2, 1, 4004, generator=torch.Generator().manual_seed(4))
    assert rel_l2(m._spec(hr, scale=True), O.spec(hr, cfg, scale=True)) < 2e-6
    z = O.spec(x, cfg)
    assert rel_l2(m._ispec(z), O.ispec(z, cfg)) < 2e-6


def test_cpu_input_without_emulator_fails_loudly(meta):
    """The product has no CPU path: a CPU tensor (or a missing library) must raise, never fall back."""
    m = build_model(meta, 'tiny')
    x = torch.zeros(1, 1, 400)
    with pytest.raises((RuntimeError, ImportError)):
        m(x)


def _train_golden(m, io):
    """Aero.forward in TRAINING mode against the reference's train-mode output (tests/golden/train_tiny_io.npz): the FTB's
    BatchNorms on batch statistics (modules.py:287,293,300) and the running-statistics update of nn.BatchNorm."""
    m.train()
    dev = next(m.parameters()).device
    with torch.no_grad():
        y, s = m(torch.from_numpy(io['x']).to(dev), return_spec=True)
    assert rel_l2(s.cpu(), io['spec']) < 1e-3
    assert rel_l2(y.cpu(), io['y']) < 5e-3
    bufs = {k[4:]: v for k, v in io.items() if k.startswith('buf.')}
    sd = m.state_dict()
    assert len(bufs) == 36
    for k, v in bufs.items():
        got = sd[k].cpu()
        if k.endswith('num_batches_tracked'):
            assert int(got) == int(v), k
        else:                                               # batch statistics of fp16 activations: 2e-3 relative
            assert torch.allclose(got.double(), torch.from_numpy(v).double(), rtol=3e-3, atol=2e-4), k
    # the updated running statistics are what the NEXT eval-mode forward folds into the convs
    m.eval()
    with torch.no_grad():
        y2 = m(torch.from_numpy(io['x']).to(dev))
    assert torch.isfinite(y2).all() and rel_l2(y2.cpu(), y.cpu()) > 1e-4


def test_train_mode_forward_golden(emu, meta):
    _train_golden(_with_engine(build_model(meta, 'tiny'), emu), load_npz('train_tiny_io.npz'))


def test_train_mode_with_autograd_is_refused(emu, meta):
    """The training engine needs 8-channel vectors (hidden sizes that are multiples of 8: every reference config and the small test
    model have them, tests/test_emu_train.py); the TINY model (4 channels, LSTM width 4) does not: a training-mode call that autograd
    would have to differentiate must fail loudly instead of returning a tensor without a graph.  With frozen parameters the same call
    is the inference engine in train mode."""
    m = _with_engine(build_model(meta, 'tiny'), emu).train()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 1, 400))
    for p in m.parameters():
        p.requires_grad_(False)
    assert m(torch.randn(1, 1, 400, generator=torch.Generator().manual_seed(2))).shape == (1, 1, 1600)


def test_weight_update_repacks(emu, meta):
    m = _with_engine(build_model(meta, 'tiny'), emu)
    x = torch.randn(1, 1, 400, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y0 = m(x)
        m.decoder[3].conv_tr.bias.add_(1.0)
        y1 = m(x)
    assert not np.allclose(y0.numpy(), y1.numpy())


def test_data_writes_need_repack_and_shape_cache_is_bounded(emu, meta):
    """`.data` writes do not bump version counters: Aero.repack() makes the engine see them.  Shape-keyed engine caches
    (padded buffers, envelopes, graphs) stay bounded when every forward has a new length (evaluate over a test set)."""
    m = _with_engine(build_model(meta, 'tiny'), emu)
    x = torch.randn(1, 1, 400, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y0 = m(x)
        m.decoder[3].conv_tr.weight.data.mul_(1.5)          # (weights are repacked to fp16 images: a copy, not a view)
        y_stale = m(x)
        m.repack()
        y1 = m(x)
        assert torch.equal(y0, y_stale) and not torch.equal(y0, y1)
        for L in range(400, 400 + 16 * 10, 16):
            m(torch.zeros(1, 1, L))
    eng = m._get_engine()
    kinds = {}
    for k in eng._tables:
        if isinstance(k, tuple):
            kinds[k[0]] = kinds.get(k[0], 0) + 1
    assert all(n <= 8 for kind, n in kinds.items() if kind in ('hidpad', 'ones', 'env', 'graph')), kinds


def test_batch_pipeline_on_cpu_is_the_plain_forward(emu, meta):
    """aero_amd/pipeline.py off the GPU: submit / result / run degrade to one forward at a time with the same results (the stream ring
    itself is covered on the MI355X: tests/test_gpu_model.py::test_batch_pipeline_matches_one_at_a_time)."""
    from aero_amd.pipeline import BatchPipeline
    m = _with_engine(build_model(meta, 'tiny'), emu).eval()
    io = load_npz('tiny_io.npz')
    xs = [torch.from_numpy(io['x_400']), torch.from_numpy(io['x_1000'])]
    pipe = BatchPipeline(m, depth=3)
    outs = pipe.run(xs, return_spec=True)
    with torch.no_grad():
        for x, (y, s) in zip(xs, outs):
            y0, s0 = m(x, return_spec=True)
            assert torch.equal(y, y0) and torch.equal(s, s0)
    m.train()
    with pytest.raises(RuntimeError):
        pipe.submit(xs[0])
