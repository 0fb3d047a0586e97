"""GPU (-m gpu): the cross-stream regression fence of DESIGN.md 5b / VERDICT r3 item 3.  Each kernel family of the path runs as a
VICTIM on one stream while another stream loops the 192-row ring conv tile; results must equal the solo run bit for bit.  The whole
forward (every inference kernel at its real shape, the iSTFT included) is one victim; the two-stream product schedule and the training
step's second-stream weight gradients are the others."""
import pytest
import torch

import concurrency_cases as cc
from conftest import build_model, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rig(meta):
    from aero_amd import _lib
    lib = _lib.load()
    m = build_model(meta, 'full').cuda()
    dist = cc.RingDisturber(lib, 'cuda')
    assert 'aero_conv_ring_kernel<2, 4, 3, 3' in dist.kernel_name(), dist.kernel_name()
    return m, dist


def test_forward_next_to_the_ring_kernel(rig):
    """every inference kernel at its real shape -- STFT, encoders, LSTM, attention, decoders incl. their own ring launches, iSTFT -- with
    a foreign stream running the 192-row ring tile: 60 forwards of 8 clips, all outputs bit-equal to the solo forward"""
    m, dist = rig
    eng = m._get_engine()
    x = seeded((8, 1, 8000), 11).cuda()
    eng.streams = 1
    try:
        bad, first = cc.overlapped(lambda: m(x, return_spec=True, return_lr_spec=True), dist.launch, 60, n_disturb=40)
    finally:
        eng.streams = 0
    assert bad == 0, (bad, first)


def test_istft_and_stft_next_to_the_ring_kernel(rig):
    """the front-end kernels alone, 300 launches each (the round-3 finding was a 2-14 % event of the iSTFT)"""
    m, dist = rig
    x = seeded((16, 1, 8000), 12).cuda()
    with torch.no_grad():
        _, spec = m(x, return_spec=True)
        hr = m(x)
    torch.cuda.synchronize()
    for name, victim in (('istft', lambda: m._ispec(spec)), ('stft_dft_form', lambda: m._spec(x)), ('stft_fft_form', lambda: m._spec(hr, scale=True))):
        bad, first = cc.overlapped(victim, dist.launch, 300, n_disturb=4)
        assert bad == 0, (name, bad, first)


def test_two_stream_forward_equals_one_stream_100_times(rig):
    """the product's own schedule (two half-batches on two streams from 32 clips up, engine.py) against the one-stream order, bit for bit"""
    m, _ = rig
    eng = m._get_engine()
    x = seeded((32, 1, 8000), 5).cuda()
    with torch.no_grad():
        eng.streams = 1
        y1, s1 = m(x, return_spec=True)
        y1, s1 = y1.clone(), s1.clone()
        eng.streams = 0
        bad = []
        for it in range(100):
            y2, s2 = m(x, return_spec=True)
            torch.cuda.synchronize()
            if not (torch.equal(y2, y1) and torch.equal(s2, s1)):
                bad.append((it, float((y2 - y1).abs().max()), float((torch.view_as_real(s2) - torch.view_as_real(s1)).abs().max())))
    assert not bad, (len(bad), bad[:4])


def test_pipelined_batches_300_times_bit_equal(rig):
    """the serving loop (aero_amd/pipeline.py: three whole batches in flight on three streams -- every kernel of one batch, its STFT and
    iSTFT included, next to the kernels of two others; the bench's timed region): 300 batches of two alternating inputs, every output
    bit-equal to the one-stream forward of its input"""
    from aero_amd.pipeline import BatchPipeline
    m, _ = rig
    m.eval()
    eng = m._get_engine()
    xs = [seeded((32, 1, 8000), 5).cuda(), seeded((32, 1, 8000), 6).cuda()]
    with torch.no_grad():
        eng.streams = 1
        refs = [tuple(t.clone() for t in m(x, return_spec=True)) for x in xs]
        eng.streams = 0
    pipe = BatchPipeline(m, depth=3)
    bad = []
    tickets = []
    for it in range(300):
        tickets.append((it, pipe.submit(xs[it % 2], return_spec=True)))
        if len(tickets) > 6:
            i, t = tickets.pop(0)
            y, s = pipe.result(t)
            if not (torch.equal(y, refs[i % 2][0]) and torch.equal(s, refs[i % 2][1])):
                bad.append((i, float((y - refs[i % 2][0]).abs().max())))
    for i, t in tickets:
        y, s = pipe.result(t)
        if not (torch.equal(y, refs[i % 2][0]) and torch.equal(s, refs[i % 2][1])):
            bad.append((i, float((y - refs[i % 2][0]).abs().max())))
    torch.cuda.synchronize()
    assert not bad, (len(bad), bad[:4])


def test_training_step_gradients_do_not_depend_on_the_second_stream(meta):
    """the training backward issues its weight-gradient GEMMs on a second stream beside the main stream's ring / norm / recurrent
    kernels (train.py: on_param_stream).  The backward is not bit-reproducible from run to run even on ONE stream (its loss-scale
    maxima and a few parameter sums go through atomics, and every later stage inherits the last bit), so the fence is statistical:
    per parameter, the distance of a two-stream gradient from a single-stream one (AERO_TRAIN_STREAMS=1) must stay within the
    run-to-run distance of single-stream gradients among themselves (x 4, floor 2e-6) -- a damaged 128-byte line in a weight
    gradient is 1e-3 .. 1e-1 of it, three orders above that floor"""
    import os
    from aero_amd import Aero, losses
    from aero_amd.optim import FlatAdam
    from conftest import rel_l2
    cfg = dict(meta['small_cfg'])
    x, hr = seeded((2, 1, 2003), 1).cuda(), (0.1 * seeded((2, 1, 8012), 2)).cuda()
    grads, names = {}, None
    for mode, reps in (('1', 6), ('2', 20)):
        os.environ['AERO_TRAIN_STREAMS'] = mode
        try:
            torch.manual_seed(3)
            m = Aero(**cfg).cuda().train()
            opt = FlatAdam(m.parameters(), lr=1e-4, model=m)
            crit = losses.MultiResolutionSTFTLoss()
            names = [n for n, _ in m.named_parameters()]
            runs = []
            for _ in range(reps):
                y = m(x)
                sc, mg = crit(y.squeeze(1), hr.squeeze(1))
                opt.zero_grad()
                (sc + mg).backward()
                torch.cuda.synchronize()
                runs.append([p.grad.double().clone() for p in m.parameters()])
            grads[mode] = runs
        finally:
            os.environ.pop('AERO_TRAIN_STREAMS', None)
    solo, two = grads['1'], grads['2']
    gmax = max(float(g.norm()) for g in solo[0])
    bad = []
    for i, n in enumerate(names):
        ref = solo[0][i]
        if float(ref.norm()) < 1e-9 * gmax:
            continue                                             # (mathematically zero gradients: rounding noise only)
        floor = max(rel_l2(r[i], ref) for r in solo[1:])
        worst = max(rel_l2(r[i], ref) for r in two)
        if not worst <= max(4.0 * floor, 2e-6):
            bad.append((n, f'{worst:.2e}', f'solo floor {floor:.2e}'))
    assert not bad, (len(bad), bad[:8])
