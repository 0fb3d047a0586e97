"""GPU (-m gpu): the training step of the generator on the MI355X (BASELINE config 5): forward under autograd ->
multi-resolution STFT loss -> HIP backward -> fused Adam.  tests/train_cases.py states what is compared and why."""
import json
import os

import pytest
import torch

import train_cases as tc
from conftest import GOLDEN, build_model, load_npz, rel_l2, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def meta():
    return json.load(open(os.path.join(GOLDEN, 'meta.json')))


def test_partially_frozen_generator():
    assert tc.case_partially_frozen('cuda') > 20


def test_training_step_small_model_vs_reference_golden():
    rows = tc.case_training_step_small('cuda')
    assert len(rows) > 250


@pytest.mark.parametrize('which', ['full', 'stress_full'])
def test_full_model_backward_vs_oracle(meta, which):
    """flagship model (fresh and 'trained-like': LayerScale O(1), live attention decay), B = 2 x 2 s: the oracle's dL/dy of the
    MR-STFT loss pushed back through the HIP graph"""
    m = build_model(meta, which).train()
    cfg = meta['full_cfg']
    x, hr = seeded((2, 1, 8000), 7), seeded((2, 1, 32000), 8) * 0.1
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    y32, dy, g32, _ = tc.oracle_grads(m, cfg, x, hr=hr)
    _, _, gq, _ = tc.oracle_grads(m, cfg, x, dy=dy, rounded=True)
    m.cuda()
    y = m(x.cuda())
    assert rel_l2(y.detach().cpu(), y32) < 4e-3
    y.backward(dy.cuda())
    tc.check_param_grads(m, g32, gq, which)


def test_music_config_forward_golden(meta):
    """BASELINE config 5 geometry (11.025 -> 44.1 kHz, n_fft 512, hop 256, 10-s segments: T = 1724, 18 LSTM frames, the streaming
    attention form inside the model), eval and train mode, against the reference's outputs"""
    io = load_npz('music_io.npz')
    torch.manual_seed(meta['music_seed'])
    from aero_amd import Aero
    from conftest import randomize_running_stats
    m = Aero(**meta['music_cfg']).eval()
    randomize_running_stats(m, meta['music_bn_seed'])
    m.cuda()
    x = seeded((2, 1, 110250), meta['music_input_seed']).cuda()
    with torch.no_grad():
        y, s = m(x, return_spec=True)
        assert y.shape == (2, 1, 441000) and s.shape == (2, 1, 256, 1724)
        assert rel_l2(s.cpu()[:, :, ::4, ::7], io['spec']) < 1e-3
        assert rel_l2(y.cpu()[..., ::16], io['y']) < 5e-3
        m.train()
        yt, st = m(x, return_spec=True)
        assert rel_l2(st.cpu()[:, :, ::4, ::7], io['spec_train']) < 1e-3
        assert rel_l2(yt.cpu()[..., ::16], io['y_train']) < 5e-3


def test_config5_training_steps(meta):
    """BASELINE config 5 per GPU: B = 2 x 10 s at 11.025 -> 44.1 kHz, train mode: forward -> loss -> backward -> FlatAdam.step,
    three steps; the loss of this fixed batch goes down and every parameter receives a finite gradient"""
    from aero_amd import Aero, losses
    from aero_amd.optim import FlatAdam
    torch.manual_seed(2036)
    m = Aero(**meta['music_cfg']).cuda().train()
    opt = FlatAdam(m.parameters(), lr=3e-4, betas=(0.9, 0.999), model=m)
    crit = losses.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5)          # main_config.yaml:64-65
    lr = seeded((2, 1, 110250), 1).cuda()
    hr = (0.1 * seeded((2, 1, 441000), 2)).cuda()
    hist = []
    for _ in range(3):
        y = m(lr)
        sc, mg = crit(y.squeeze(1), hr.squeeze(1))
        loss = sc + mg
        opt.zero_grad()
        loss.backward()
        assert torch.isfinite(opt.flat_g).all() and float(opt.flat_g.abs().max()) > 0
        for n, p in m.named_parameters():
            assert p.grad is not None and p.grad.data_ptr() >= opt.flat_g.data_ptr(), n
        opt.step()
        hist.append(float(loss.detach()))
    assert hist[2] < hist[0], hist


def test_weight_images_replayed_by_gather_match_their_closures():
    """aero_gather_pack on the device: every replayed weight image == its packing closure's output, bit for bit, after each step"""
    assert tc.case_weight_replay('cuda') > 50


def test_config5_weight_replay_covers_the_model(meta):
    """the full-size music model: which image sets the replay declines (and keeps rebuilding with their closures) is a short, known list"""
    from aero_amd import Aero, losses
    from aero_amd.optim import FlatAdam
    from aero_amd.repack import _flatten
    torch.manual_seed(2036)
    m = Aero(**meta['music_cfg']).cuda().train()
    opt = FlatAdam(m.parameters(), lr=3e-4, model=m)
    crit = losses.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5)
    lr, hr = seeded((1, 1, 22050), 1).cuda(), (0.1 * seeded((1, 1, 88200), 2)).cuda()
    for _ in range(2):
        y = m(lr)
        sc, mg = crit(y.squeeze(1), hr.squeeze(1))
        opt.zero_grad()
        (sc + mg).backward()
        opt.step()
    eng = m._get_train_engine()
    eng._sync_weights(lr.device)
    rp = eng._replay
    assert rp is not None and len(rp.objects) > 100
    assert all(k.endswith('lstm.specs') or k.endswith('qkvd_dgrad') for k in rp.skipped), rp.skipped
    for key, obj in rp.objects.items():
        for a, b in zip(_flatten(obj, []), _flatten(eng._builders[key](), [])):
            assert torch.equal(a, b), key


def test_captured_training_step_matches_eager(meta):
    """aero_amd.train.CapturedStep: the whole step (forward, loss, backward, fused Adam with the step count advancing) replayed as one HIP
    graph follows the eager loop on the same data: same losses step by step (to the atomics' rounding), same weights afterwards"""
    from aero_amd import Aero, losses
    from aero_amd.optim import FlatAdam
    from aero_amd.train import CapturedStep
    cfg = dict(meta['small_cfg'])
    x, hr = seeded((2, 1, 2003), 1).cuda(), (0.1 * seeded((2, 1, 8012), 2)).cuda()
    runs = []
    for captured in (False, True):
        torch.manual_seed(3)
        m = Aero(**cfg).cuda().train()
        opt = FlatAdam(m.parameters(), lr=1e-4, model=m)
        crit = losses.MultiResolutionSTFTLoss()

        def step(a, b):
            y = m(a)
            sc, mg = crit(y.squeeze(1), b.squeeze(1))
            loss = sc + mg
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss.detach()
        hist = []
        if captured:
            cap = CapturedStep(step, x, hr, warmup=2, optimizers=[opt])          # two eager warm-up steps, then the capture
            for _ in range(3):
                hist.append(float(cap(x, hr)))
            # the replays changed the weights behind every host-side cache key -- an eval-mode forward right after them
            # must run on the CURRENT weights (== a fresh module that loads them), not on the images of capture time
            m.eval()
            m2 = Aero(**cfg).cuda().eval()
            m2.load_state_dict(m.state_dict())
            with torch.no_grad():
                ya, yb = m(x), m2(x)
            assert torch.equal(ya, yb), float((ya - yb).abs().max())
            m.train()
        else:
            for i in range(5):
                v = float(step(x, hr))
                if i >= 2:
                    hist.append(v)
        runs.append((hist, opt.flat_p.clone(), opt.step_count))
    (h0, p0, n0), (h1, p1, n1) = runs
    assert n0 == n1 == 5
    # (not bit-identical: fp64 / fp32 atomics order differs from run to run, and Adam's first steps turn the sign of a near-zero
    # gradient component into a full +-lr update -- two EAGER runs differ by the same few 1e-3)
    assert all(abs(a - b) < 1e-2 * abs(a) for a, b in zip(h0, h1)), (h0, h1)
    assert h0[2] < h0[0] and h1[2] < h1[0]
    assert rel_l2(p1.cpu(), p0.cpu()) < 1e-2        # (5 sign-like Adam updates of 1e-4 on weights of ~0.05: the bound of what can differ)


def test_training_tracks_the_reference_loss_trajectory(meta):
    """30 steps of the reference's own training loop (fp32 CPU: its Aero in train mode, its MultiResolutionSTFTLoss,
    torch.optim.Adam(lr 3e-4, betas (0.9, 0.999)) as train.py:83 builds it) on one fixed batch are committed as a golden
    (oracle/make_golden_train.py -> tests/golden/train_small_trajectory.npz).  The HIP loop -- fp16 activation / gradient storage, the
    loss on the HIP STFT, the fused FlatAdam -- must follow that trajectory: every step's loss within 2 %, the last within 1 %.
    (The loss falls from 2.07 to 0.70 over these steps: a loop that drifted, stalled or mis-scaled an update would leave that corridor.)"""
    from aero_amd import Aero, losses
    from aero_amd.optim import FlatAdam
    cfgt = meta['train_small_trajectory']
    gold = load_npz('train_small_trajectory.npz')['loss']                # [steps, {sc, mag}]
    torch.manual_seed(cfgt['model_seed'])
    m = Aero(**dict(meta['small_cfg'])).cuda().train()
    opt = FlatAdam(m.parameters(), lr=cfgt['lr'], betas=tuple(cfgt['betas']), model=m)
    crit = losses.MultiResolutionSTFTLoss(factor_sc=cfgt['factor_sc'], factor_mag=cfgt['factor_mag'])
    x = seeded((2, 1, cfgt['L']), cfgt['x_seed']).cuda()
    hr = (cfgt['hr_scale'] * seeded((2, 1, 4 * cfgt['L']), cfgt['hr_seed'])).cuda()
    got = []
    for _ in range(cfgt['steps']):
        y = m(x)
        sc, mg = crit(y.squeeze(1), hr.squeeze(1))
        opt.zero_grad()
        (sc + mg).backward()
        opt.step()
        got.append((float(sc.detach()), float(mg.detach())))
    got = torch.tensor(got, dtype=torch.float64)
    ref = torch.from_numpy(gold)
    tot_g, tot_r = got.sum(1), ref.sum(1)
    rel = ((tot_g - tot_r).abs() / tot_r)
    print('trajectory: max relative deviation %.3e (step %d), last step %.3e; first / last loss %.4f / %.4f (reference %.4f / %.4f)' % (
        float(rel.max()), int(rel.argmax()), float(rel[-1]), float(tot_g[0]), float(tot_g[-1]), float(tot_r[0]), float(tot_r[-1])))
    # The loop is reproducible from run to run since the GroupNorm backward stages its sums in fp64 (k_bwd.h: with fp32 atomics there the
    # same test gave 0.9-2.1 % / 0.1-1.0 % over eight runs -- 30 Adam steps amplify a 1e-7 scatter -- now 0.89 % at step 25 and 0.015 % at
    # the end, every run): the bounds are the ones the round-3 review asked for.
    assert float(rel.max()) < 2e-2, (rel.tolist(), got.tolist())
    assert float(rel[-1]) < 1e-2
    assert float(rel.median()) < 5e-3
    assert float(((got - ref).abs() / ref).max()) < 4e-2                # each term (sc, mag) on its own


def test_training_is_reproducible_run_to_run(meta):
    """two runs of the same eight training steps (same seed, same batch) end with bit-identical losses and parameters: no reduction of the
    step depends on the order in which blocks finish (weight gradients in slabs, GroupNorm backward sums staged in fp64, fixed-order loss
    sums).  Without that the loss trajectories of two runs drift apart by percents within 20 steps."""
    from aero_amd import Aero, losses
    from aero_amd.optim import FlatAdam
    cfgt = meta['train_small_trajectory']

    def run():
        torch.manual_seed(cfgt['model_seed'])
        m = Aero(**dict(meta['small_cfg'])).cuda().train()
        opt = FlatAdam(m.parameters(), lr=cfgt['lr'], betas=tuple(cfgt['betas']), model=m)
        crit = losses.MultiResolutionSTFTLoss(factor_sc=cfgt['factor_sc'], factor_mag=cfgt['factor_mag'])
        x = seeded((2, 1, cfgt['L']), cfgt['x_seed']).cuda()
        hr = (cfgt['hr_scale'] * seeded((2, 1, 4 * cfgt['L']), cfgt['hr_seed'])).cuda()
        ls = []
        for _ in range(8):
            sc, mg = crit(m(x).squeeze(1), hr.squeeze(1))
            opt.zero_grad()
            (sc + mg).backward()
            opt.step()
            ls.append(float((sc + mg).detach()))
        torch.cuda.synchronize()
        return ls, torch.cat([p.detach().flatten() for p in m.parameters()]).clone()

    l1, p1 = run()
    l2, p2 = run()
    assert l1 == l2, (l1, l2)
    assert torch.equal(p1, p2)


def test_adversarial_training_tracks_the_reference_trajectory(meta):
    """the recipe BASELINE config 5's experiment file trains with (`adversarial: true`, msd_melgan).  The reference's
    own loop -- its generator, its MelGAN critic, its MultiResolutionSTFTLoss, hinge and feature-matching losses of solver.py:475-520, two
    torch.optim.Adam, generator step then critic step (solver.py:602-612), fp32 CPU -- over 12 steps on one fixed batch is committed as a
    golden (oracle/make_golden_train_gan.py -> tests/golden/train_gan_trajectory.npz: per step stft, adversarial, features, discriminator
    loss).  `aero_amd.trainer.TrainStep` (HIP generator + critic, fp16 activation storage, both FlatAdam) must follow it term by term."""
    from aero_amd import trainer
    from aero_amd.config import _wrap
    cfgt = meta['train_gan_trajectory']
    gold = torch.from_numpy(load_npz('train_gan_trajectory.npz')['loss'])              # [steps, {stft, adv, feat, disc}]
    args = _wrap(dict(optim='adam', lr=cfgt['lr'], beta2=cfgt['betas'][1], losses=['stft'], stft_sc_factor=0.5, stft_mag_factor=0.5,
                      experiment=dict(model='aero', aero=cfgt['gen_cfg'], adversarial=True, features_loss_lambda=cfgt['features_loss_lambda'],
                                      only_features_loss=False, only_adversarial_loss=False, discriminator_models=['msd_melgan'],
                                      melgan_discriminator=cfgt['disc_cfg'])))
    torch.manual_seed(cfgt['seed'])
    models = {k: m.cuda().train() for k, m in trainer.build_models(args).items()}
    opts = trainer.build_optimizers(models, args)
    step = trainer.TrainStep(models, opts, args)
    x = seeded((2, 1, cfgt['L']), cfgt['x_seed']).cuda()
    hr = (cfgt['hr_scale'] * seeded((2, 1, 4 * cfgt['L']), cfgt['hr_seed'])).cuda()
    got = []
    for _ in range(cfgt['steps']):
        rec = step(x, hr)
        got.append([float(rec[k]) for k in ('generator_stft', 'generator_adversarial_melgan', 'generator_features_melgan', 'discriminator_msd_melgan')])
    got = torch.tensor(got, dtype=torch.float64)
    rel = (got - gold).abs() / gold
    print('adversarial trajectory: worst relative deviation per term (stft, adv, feat, disc) %s, at the last step %s' % (
        [f'{float(v):.2e}' for v in rel.max(0).values], [f'{float(v):.2e}' for v in rel[-1]]))
    # Step 0 is a pure forward comparison (same weights on both sides): 1e-5.  Over the twelve steps the STFT, feature-matching and
    # critic (hinge) terms stay within 2e-3 of the reference (measured 1.6e-4 / 2.0e-3 / 4.7e-4); their bars are 1e-2 / 1e-2 / 5e-3.
    # The generator's ADVERSARIAL term, 3 - sum_scales mean D(fake), is the one ill-conditioned quantity of this recipe at initialisation:
    # it follows the COMMON MODE of the critic's logits, which the critic's loss does not see (relu(1 + f) + relu(1 - r) with |f|, |r| << 1
    # is 2 + f - r: the last bias has an exactly zero gradient on both sides, and every other critic gradient is the small difference of
    # two nearly equal fake / real contributions).  Adam turns each such gradient into a step of +-lr by its SIGN, so rounding decides
    # where the common mode drifts: the HIP critic's first-step gradients agree with the reference's to 0.4-3 % on the first scale and
    # 1-28 % on the bias vectors of the second (tools/dbg/gan_step1.py), and a few % of the elements take their first step the other
    # way.  The separation f - r (the hinge term) is unaffected; the adversarial term drifts 0.1 % (step 1) -> 5.0 % (step 11): bar 8 %,
    # and 1.5 % over the first four steps.
    assert float(rel[0].max()) < 1e-4, rel[0].tolist()
    assert float(rel[:, 0].max()) < 1e-2 and float(rel[:, 2].max()) < 1e-2 and float(rel[:, 3].max()) < 5e-3, rel.max(0).values.tolist()
    assert float(rel[:4, 1].max()) < 1.5e-2 and float(rel[:, 1].max()) < 8e-2, rel[:, 1].tolist()


def test_adversarial_training_is_reproducible_run_to_run():
    """the step the experiment files train with (generator step with MR-STFT + adversarial + feature-matching losses, then the msd_melgan
    critic's step: aero_amd/trainer.py) twice from the same seed: bit-identical generator AND critic parameters after four steps"""
    from aero_amd import trainer
    from aero_amd.config import _wrap
    gen = dict(channels=16, nfft=512, hop_length=256, lr_sr=4000, hr_sr=16000)
    args = _wrap(dict(optim='adam', lr=3e-4, beta2=0.999, losses=['stft'], stft_sc_factor=0.5, stft_mag_factor=0.5,
                      experiment=dict(model='aero', aero=gen, adversarial=True, features_loss_lambda=100, only_features_loss=False,
                                      only_adversarial_loss=False, discriminator_models=['msd_melgan'],
                                      melgan_discriminator=dict(n_layers=4, num_D=3, downsampling_factor=4, ndf=16))))

    def run():
        torch.manual_seed(77)
        models = {k: m.cuda().train() for k, m in trainer.build_models(args).items()}
        opts = trainer.build_optimizers(models, args)
        step = trainer.TrainStep(models, opts, args)
        for i in range(4):
            lr = seeded((2, 1, 8000), 300 + i).cuda()
            hr = (0.1 * seeded((2, 1, 32000), 400 + i)).cuda()
            rec = step(lr, hr)
        torch.cuda.synchronize()
        return ({k: float(v) for k, v in rec.items()}, opts['optimizer'].flat_p.clone(), opts['disc_optimizer'].flat_p.clone())

    r1, g1, d1 = run()
    r2, g2, d2 = run()
    assert r1 == r2, (r1, r2)
    assert torch.equal(g1, g2), float((g1 - g2).abs().max())
    assert torch.equal(d1, d2), float((d1 - d2).abs().max())
