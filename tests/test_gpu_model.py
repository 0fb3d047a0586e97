"""GPU (-m gpu): Aero.forward on the MI355X against the golden vectors captured from the reference
(tests/golden, made by oracle/make_golden.py) and against the CPU oracle.
Tolerance (north_star): <= 1e-3 relative L2 on the complex spectrogram; STFT (fp32) <= 2e-6."""
import pytest
import torch

from conftest import build_model, load_npz, rel_l2

pytestmark = pytest.mark.gpu


def _fwd(m, x):
    with torch.no_grad():
        y, s, lr = m(x.cuda(), return_spec=True, return_lr_spec=True)
    torch.cuda.synchronize()
    return y.cpu(), s.cpu(), lr.cpu()


def test_native_library_is_the_gfx950_build():
    from aero_amd import _lib
    lib = _lib.load()
    assert 'gfx950' in lib.version and not lib.is_emulator
    assert 'no-packed-fp32' in lib.version, lib.version           # (the build flag the concurrency fence rests on)


def test_no_aero_switch_is_set_on_the_test_box():
    """the -m gpu suite is evidence for the DEFAULT configuration: aero_version() names every AERO_* variable present in the environment
    (each one is a departure from what is tested), and on the box that runs this suite there must be none (VERDICT r4 hygiene)"""
    import os
    from aero_amd import _lib
    ver = _lib.load().cdll.aero_version().decode()                # (read now: the switches are looked up at call time)
    stray = sorted(k for k in os.environ if k.startswith('AERO_'))
    assert 'non-default switches' not in ver and not stray, (ver, stray)


@pytest.mark.parametrize('L', [400, 1000, 999])
def test_tiny_model_golden(meta, L):
    m = build_model(meta, 'tiny').cuda()
    io = load_npz('tiny_io.npz')
    y, s, lr = _fwd(m, torch.from_numpy(io[f'x_{L}']))
    assert rel_l2(lr, io[f'lr_{L}']) < 2e-6
    assert rel_l2(s, io[f'spec_{L}']) < 1e-3
    assert rel_l2(y, io[f'y_{L}']) < 5e-3


@pytest.mark.parametrize('L', [800, 2003])
def test_small_model_golden(meta, L):
    m = build_model(meta, 'small').cuda()
    io = load_npz('small_io.npz')
    y, s, lr = _fwd(m, torch.from_numpy(io[f'x_{L}']))
    assert rel_l2(lr, io[f'lr_{L}']) < 2e-6
    assert rel_l2(s, io[f'spec_{L}']) < 1e-3
    assert rel_l2(y, io[f'y_{L}']) < 5e-3


def test_full_model_golden(meta):
    """aero_4-16_512_64, seed 2036: the first two clips of BASELINE config 2's input."""
    m = build_model(meta, 'full').cuda()
    io = load_npz('full_io.npz')
    x = torch.randn(2, 1, 8000, generator=torch.Generator().manual_seed(0))
    y, s, lr = _fwd(m, x)
    assert y.shape == (2, 1, 32000) and s.shape == (2, 1, 256, 501)
    assert rel_l2(lr[:, :, ::8, ::5], io['lr']) < 2e-6
    e_spec, e_wav = rel_l2(s, io['spec']), rel_l2(y, io['y'])
    print(f'full model: spectrogram rel-L2 {e_spec:.3e}, waveform rel-L2 {e_wav:.3e}')
    assert e_spec < 1e-3
    assert e_wav < 5e-3


def test_full_model_batch64_matches_golden_and_is_batch_invariant(meta):
    """BASELINE config 2 size (B=64): clips 0,1 are the golden clips; every clip's result must not depend
    on its batch neighbours (clips are independent units -- the property the multi-GPU sharding relies on)."""
    m = build_model(meta, 'full').cuda()
    io = load_npz('full_io.npz')
    g = torch.Generator().manual_seed(0)
    x2 = torch.randn(2, 1, 8000, generator=g)
    rest = torch.randn(62, 1, 8000, generator=torch.Generator().manual_seed(1))
    x = torch.cat([x2, rest], 0)
    y, s, _ = _fwd(m, x)
    assert rel_l2(s[:2], io['spec']) < 1e-3
    # permute the batch: results follow the clips.  Per-clip arithmetic is identical; the only order-dependent step is the
    # fp64 atomic combination of GroupNorm partial sums (rounding at the 1e-16 level), so agreement is to ~1 fp16 ulp
    # on isolated elements rather than bitwise.
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(2))
    yp, sp, _ = _fwd(m, x[perm])
    assert rel_l2(sp, s[perm]) < 1e-5
    assert rel_l2(yp, y[perm]) < 1e-5
    assert torch.isfinite(y).all()


def test_hip_graph_replay_equals_eager(meta):
    """AERO_GRAPH path: the forward captured as a HIP graph (per input shape) and replayed on NEW inputs returns exactly
    what the eager launch sequence returns (same kernels, same order: bit-identical)."""
    m = build_model(meta, 'small').cuda()
    eng = m._get_engine()
    xs = [torch.randn(3, 1, 2003, generator=torch.Generator().manual_seed(i)) for i in (1, 2, 3)]
    eng.use_graph = False
    ref = [_fwd(m, x) for x in xs]
    eng.use_graph = True
    try:
        for x, r in zip(xs, ref):                      # first call captures, the others replay
            got = _fwd(m, x)
            for a, b in zip(got, r):
                assert torch.equal(a, b)
        # new weights: the engine repacks them and must drop the graphs captured over the old packed buffers
        with torch.no_grad():
            m.decoder[0].rewrite.weight.mul_(1.5)
        got = _fwd(m, xs[0])
        eng.use_graph = False
        want = _fwd(m, xs[0])
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert not torch.equal(got[0], ref[0][0])
    finally:
        eng.use_graph = False


def test_wide_band_geometry_golden(meta):
    """BASELINE config 4 geometry (12->48 kHz, nfft 1024, hop 256)."""
    m = build_model(meta, 'wide').cuda()
    io = load_npz('wide_io.npz')
    x = torch.randn(1, 1, 6000, generator=torch.Generator().manual_seed(31))
    y, s, _ = _fwd(m, x)
    assert rel_l2(s, io['spec']) < 1e-3
    assert rel_l2(y, io['y']) < 5e-3


def test_stft_istft_round_trip_full_size():
    """Size-independent property at BASELINE size: iSTFT(STFT(x)) == x for a COLA window (hann, hop = n/8)."""
    from aero_amd import Aero
    m = Aero(nfft=512, hop_length=64, lr_sr=16000, hr_sr=16000).eval().cuda()     # scale 1: same geometry both ways
    x = torch.randn(64, 1, 32000, generator=torch.Generator().manual_seed(9)).cuda()
    z = m._spec(x)
    # Nyquist is dropped by _spec (aero.py:420) so remove it from x first: compare against the oracle instead
    from oracle import aero_oracle as O
    cfg = {**O.DEFAULT_CFG, 'lr_sr': 16000, 'hr_sr': 16000}
    zr = O.spec(x[:2].cpu(), cfg)
    assert rel_l2(z[:2].cpu(), zr) < 2e-6
    y = m._ispec(z)
    yr = O.ispec(zr, cfg)
    assert rel_l2(y[:2].cpu(), yr) < 2e-6
    # linearity of the STFT kernel at full size
    a = m._spec(2.0 * x) - 2.0 * z
    assert float(a.abs().max()) < 1e-4


def test_cpu_tensor_is_refused(meta):
    m = build_model(meta, 'tiny').cuda()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 400))


@pytest.mark.parametrize('L', [800, 2003])
def test_stress_small_model_golden(meta, L):
    """"Trained-like" weights (oracle/stress.py: LayerScale U(0.2,1), live attention decay, perturbed norms, wide Snake
    spread): the DConv branch -- LSTM, LocalState, Snake, both conv1d -- is no longer scaled away by 1e-3, so the
    end-to-end bar of 1e-3 on the complex spectrogram now covers those kernels too."""
    m = build_model(meta, 'stress_small').cuda()
    io = load_npz('stress_small_io.npz')
    y, s, _ = _fwd(m, torch.from_numpy(load_npz('small_io.npz')[f'x_{L}']))
    e_spec, e_wav = rel_l2(s, io[f'spec_{L}']), rel_l2(y, io[f'y_{L}'])
    print(f'stress small L={L}: spectrogram rel-L2 {e_spec:.3e}, waveform rel-L2 {e_wav:.3e}')
    assert e_spec < 1e-3
    assert e_wav < 5e-3


def test_stress_full_model_golden(meta):
    m = build_model(meta, 'stress_full').cuda()
    io = load_npz('stress_full_io.npz')
    x = torch.randn(2, 1, 8000, generator=torch.Generator().manual_seed(0))
    y, s, _ = _fwd(m, x)
    e_spec, e_wav = rel_l2(s, io['spec']), rel_l2(y, io['y'])
    print(f'stress full: spectrogram rel-L2 {e_spec:.3e}, waveform rel-L2 {e_wav:.3e}')
    assert e_spec < 1e-3
    assert e_wav < 5e-3


def test_config4_full_size(meta):
    """BASELINE config 4 at full size: [32, 1, 24000] (12->48 kHz, nfft 1024, hop 256, F=512, T=376).  Clip 0 against the
    reference's output; then size-independent properties over the whole batch: every clip equals its single-clip
    forward (independence), output length int(L*scale), finite."""
    m = build_model(meta, 'wide').cuda()
    io = load_npz('wide_full_io.npz')
    x = torch.randn(32, 1, 24000, generator=torch.Generator().manual_seed(meta['wide_full_input_seed']))
    y, s, lr = _fwd(m, x)
    assert y.shape == (32, 1, 96000) and s.shape == (32, 1, 512, 376) and lr.shape == (32, 1, 512, 376)
    e_spec, e_wav = rel_l2(s[:1, :, ::4, ::3], io['spec']), rel_l2(y[:1, ..., ::4], io['y'])
    print(f'config 4 full size: spectrogram rel-L2 {e_spec:.3e}, waveform rel-L2 {e_wav:.3e}')
    assert e_spec < 1e-3 and e_wav < 5e-3
    assert torch.isfinite(y).all() and torch.isfinite(torch.view_as_real(s)).all()
    for i in (0, 17, 31):
        yi, si, _ = _fwd(m, x[i:i + 1])
        assert rel_l2(si, s[i:i + 1]) < 1e-5 and rel_l2(yi, y[i:i + 1]) < 1e-5
    from oracle import aero_oracle as O
    assert rel_l2(lr[:2], O.spec(x[:2], meta['wide_cfg'])) < 2e-6


def test_predict_path_on_a_wav_file(meta, tmp_path):
    """predict.py's path (BASELINE config 1 plumbing, on the GPU): an 85000-sample 4 kHz wav -> chunks [0,40000),
    [40000,80000), [80000,85000) (exact), two full chunks batched in ONE forward with overlapped copies, tail alone;
    the result equals the chunk-by-chunk forward and has 4x the samples; the written file round-trips."""
    from aero_amd import audio_io, enhance
    m = build_model(meta, 'full').cuda()
    sig = 0.1 * torch.randn(1, 85000, generator=torch.Generator().manual_seed(4))
    p = str(tmp_path / 'in.wav')
    audio_io.save(p, sig, 4000)
    lr_sig, sr = audio_io.load(p)
    assert sr == 4000 and torch.equal(lr_sig, sig)
    assert enhance.chunk_ranges(85000, sr) == [(0, 40000), (40000, 80000), (80000, 85000)]
    pr = enhance.predict_signal(m, lr_sig, sr)
    assert pr.shape == (1, 340000) and torch.isfinite(pr).all()
    with torch.no_grad():
        ref = torch.cat([m(lr_sig[:, a:b].unsqueeze(1).cuda()).squeeze(1).cpu() for a, b in enhance.chunk_ranges(85000, sr)], -1)
    assert rel_l2(pr, ref) < 1e-5
    pr2 = enhance.predict_signal(m, lr_sig, sr, max_clips=1)          # bounded batches: one chunk per forward, same result
    assert rel_l2(pr2, ref) < 1e-5
    out = str(tmp_path / 'out_pr.wav')
    enhance.write(pr, out, 16000)
    back, sr2 = audio_io.load(out)
    assert sr2 == 16000 and torch.allclose(back, pr / max(float(pr.abs().max()), 1.0))


def test_train_mode_forward_golden(meta):
    """Training-mode forward on the MI355X (batch-statistics BatchNorm in the FTBs + running-stat update) against the
    reference's train-mode output."""
    from test_emu_model import _train_golden
    _train_golden(build_model(meta, 'tiny').cuda(), load_npz('train_tiny_io.npz'))


def test_two_stream_forward_equals_one_stream(meta):
    """From 32 clips up a lone model(x) runs as two half-batches on two HIP streams (engine.py: AERO_STREAMS auto).  ONE schedule-independent
    result behind model(x): at the bench size, B = 64, the two-half forward equals the one-stream forward BIT FOR BIT,
    waveform and spectrogram.  (Round 5: the only batch-size dependent arithmetic on the path was the STFT's per-item statistics -- a
    thread's fp32 partial ran over as many time tiles as its block walked, and that number follows the batch size; tools/dbg/half_vs_full.py.)"""
    m = build_model(meta, 'full').cuda()
    eng = m._get_engine()
    x = torch.randn(64, 1, 8000, generator=torch.Generator().manual_seed(5))
    try:
        eng.streams = 1
        y1, s1, _ = _fwd(m, x)
        eng.streams = 0
        y2, s2, _ = _fwd(m, x)
        eng.streams = 1
        ya, sa, _ = _fwd(m, x[:32])
        yb, sb, _ = _fwd(m, x[32:])
    finally:
        eng.streams = 0
    assert torch.equal(s2, s1) and torch.equal(y2, y1)
    # ... and a batch of 64 equals its two halves run as batches of 32 (clips are independent units: what the multi-GPU sharding relies on)
    assert torch.equal(torch.cat([sa, sb]), s1) and torch.equal(torch.cat([ya, yb]), y1)


def test_batch_pipeline_matches_one_at_a_time(meta):
    """aero_amd/pipeline.py: batches enqueued on a ring of HIP streams without waiting for each other (the bench's timed region, the
    enhance loop) give, batch for batch, the BIT-identical output of the single-stream forward -- 30 batches of 3 different inputs and two
    shapes, 3 in flight."""
    from aero_amd.pipeline import BatchPipeline
    m = build_model(meta, 'full').cuda().eval()
    eng = m._get_engine()
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(16, 1, 8000, generator=g).cuda(), torch.randn(16, 1, 8000, generator=g).cuda(), torch.randn(8, 1, 6000, generator=g).cuda()]
    try:
        eng.streams = 1
        with torch.no_grad():
            refs = [m(x, return_spec=True) for x in xs]
    finally:
        eng.streams = 0
    pipe = BatchPipeline(m, depth=3)
    order = [i % 3 for i in range(30)]
    outs = pipe.run([xs[i] for i in order], return_spec=True)
    torch.cuda.synchronize()
    assert eng.streams == 0
    for i, (y, s) in zip(order, outs):
        assert torch.equal(y, refs[i][0]) and torch.equal(s, refs[i][1])
    # host tensors in, pinned host tensors out: upload and download ride on each batch's own stream (the enhance.py loop)
    outs_h = pipe.run([xs[i].cpu() for i in order[:9]], to_host=True, return_spec=True)
    for i, (y, s) in zip(order[:9], outs_h):
        assert not y.is_cuda and y.is_pinned() and torch.equal(y, refs[i][0].cpu()) and torch.equal(s, refs[i][1].cpu())


def test_batch_pipeline_repacks_between_batches(meta):
    """weights edited while batches are in flight: the batches submitted before the edit come out with the old weights, the ones after it
    with the new (the re-pack waits for the batches in flight: their packed images must not be recycled under them)"""
    from aero_amd.pipeline import BatchPipeline
    m = build_model(meta, 'full').cuda().eval()
    eng = m._get_engine()
    x = torch.randn(16, 1, 8000, generator=torch.Generator().manual_seed(12)).cuda()
    w = m.decoder[-1].conv_tr.weight
    try:
        eng.streams = 1
        with torch.no_grad():
            y_old = m(x).clone()
            w.mul_(1.5)
            y_new = m(x).clone()
            w.div_(1.5)
            assert torch.equal(m(x), y_old)
    finally:
        eng.streams = 0
    assert not torch.equal(y_old, y_new)
    pipe = BatchPipeline(m, depth=3)
    with torch.no_grad():
        before = [pipe.submit(x) for _ in range(4)]
        w.mul_(1.5)
        after = [pipe.submit(x) for _ in range(4)]
        outs_b = [pipe.result(t) for t in before]
        outs_a = [pipe.result(t) for t in after]
        torch.cuda.synchronize()
        w.div_(1.5)
    assert all(torch.equal(y, y_old) for y in outs_b)
    assert all(torch.equal(y, y_new) for y in outs_a)


def test_planted_faults_are_caught_on_the_device(meta):
    """tests/sensitivity_cases.py once on the MI355X: with one deliberate fault at a time in the launch sequence (LSTM output, direction
    halves, stitch map, LocalState decay / key index / head map, Snake, DConv dilations / residual / LayerScale / GLU half, frequency
    embedding, FTB gate, freq_fc orientation, ...) the reference's stress golden -- or, for the two one-step index slips, its op-level
    module vector -- must FAIL the 1e-3 bar that the clean run passes.  The goldens see every branch, on the hardware too."""
    import sensitivity_cases as S
    from aero_amd import _lib
    lib = _lib.load()
    e_spec, e_wav = S.run(meta, lib, None, device='cuda')
    assert e_spec < S.BAR and e_wav < 5e-3
    report = {}
    for fault in S.FAULTS:
        e, _ = S.run(meta, lib, fault, device='cuda')
        report[fault] = e
        if fault not in S.SURVIVORS:
            assert e > S.BAR, f'{fault} survived on the device: {e:.3e}'
    print('clean %.3e | ' % e_spec + ' | '.join(f'{k} {v:.2e}' for k, v in report.items()))


def test_batch_pipeline_schedules_are_bit_identical(meta):
    """Round 6: whatever the serving loop does with streams -- the default's two event waits per batch, free-running streams, stages of a
    batch on a high-priority or a CU-masked stream created through the C ABI (aero_stream_create), the recurrent launches alone on one --
    every batch's output is the single-stream forward's, bit for bit: the schedule decides WHEN a launch runs, never what it computes."""
    from aero_amd.pipeline import BatchPipeline
    m = build_model(meta, 'full').cuda().eval()
    eng = m._get_engine()
    g = torch.Generator().manual_seed(21)
    xs = [torch.randn(16, 1, 8000, generator=g).cuda(), torch.randn(8, 1, 6000, generator=g).cuda()]
    try:
        eng.streams = 1
        with torch.no_grad():
            refs = [m(x, return_spec=True) for x in xs]
    finally:
        eng.streams = 0
    order = [i % 2 for i in range(12)]
    for kw in (dict(), dict(waits=[]), dict(stagger=2.5), dict(waits=[(0, 3), (4, 8)], depth=4),
               dict(schedule='stages=mmllmmmm;l=prio:-1'), dict(schedule='lstm=l;l=prio:-1'), dict(schedule='stages=mmlldddd;l=mask:0:128;d=prio:1')):
        pipe = BatchPipeline(m, **{'depth': 3, **kw})
        outs = pipe.run([xs[i] for i in order], return_spec=True)
        torch.cuda.synchronize()
        for i, (y, s) in zip(order, outs):
            assert torch.equal(y, refs[i][0]) and torch.equal(s, refs[i][1]), kw
    assert eng.stage_hook is None and eng.streams == 0
