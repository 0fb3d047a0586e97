"""bench.py -- real-time factor of Aero.forward (STFT + U-Net + iSTFT) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts N ranks itself, aero_amd/launcher.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one forward pass of the hot path over one batch of synthetic clips already resident in
HBM (BASELINE.json configs[1]: batch 64 x 2 s white noise, 4->16 kHz, nfft 512, hop 64, random-init
weights, seed 2036).  Clips are independent units: with N ranks every rank processes its own 64
clips (weak scaling, no data-path collective; clip i -> rank i mod N as reference distrib.py:100);
timing is barrier + synchronize on both sides, max over ranks.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL_CFG = dict(in_channels=1, out_channels=1, channels=48, growth=2, nfft=512, hop_length=64, end_iters=0,
                cac=True, rewrite=True, hybrid=False, hybrid_old=False, freq_emb=0.2, emb_scale=10,
                emb_smooth=True, kernel_size=8, strides=[4, 4, 2, 2], context=1, context_enc=0, freq_ends=4,
                enc_freq_attn=0, norm_starts=2, norm_groups=4, dconv_mode=1, dconv_depth=2, dconv_comp=4,
                dconv_time_attn=2, dconv_lstm=2, dconv_init=1e-3, rescale=0.1, lr_sr=4000, hr_sr=16000,
                spec_upsample=True, act_func='snake', debug=False)

PEAK_MFMA_F16_TFLOPS = 2500.0      # dense fp16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0              # HBM3E spec peak, same table


INFERENCE_KERNEL_SOURCES = ('aero_common.h', 'k_attn.h', 'k_conv.h', 'k_conv_ring.h', 'k_dconv.h', 'k_enc0.h', 'k_ftb.h', 'k_gram.h', 'k_lstm.h',
                            'k_norm.h', 'k_pw.h', 'k_stft.h')


def kernels_sha():
    """fingerprint of the kernel sources the forward pass runs: profiles/pmc_traffic.json carries the one of its PMC visit
    (tools/pmc_traffic.py), and `roofline.traffic` is reported only while the two agree"""
    import hashlib
    h = hashlib.sha256()
    for f in INFERENCE_KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, 'aero_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                                    # cgroup v2 CPU quota, if any
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline_worker(seconds_per_clip, lr_sr, budget_s=20.0):
    """The oracle (a port of the reference's CPU path) timed on this box's host cores: a bounded sample.
    Runs in a child process (see cpu_baseline) so that a slow host cannot stall the benchmark."""
    from oracle import aero_oracle as O
    from aero_amd import Aero
    torch.manual_seed(2036)
    model = Aero(**FULL_CFG).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    threads = min(_usable_cores(), 64)                      # more threads than that only adds sync overhead here
    torch.set_num_threads(threads)
    L = int(seconds_per_clip * lr_sr)

    def run(batch):
        x = torch.randn(batch, 1, L, generator=torch.Generator().manual_seed(0))
        t0 = time.perf_counter()
        with torch.no_grad():
            O.aero_forward(sd, FULL_CFG, x, fast=True)
        return time.perf_counter() - t0
    run(1)                                                  # warm-up
    t1 = run(1)
    batch = int(max(1, min(8, budget_s / 3.0 / max(t1, 1e-3))))
    ts = sorted(run(batch) for _ in range(3))
    med = ts[1]
    # the reference's own inference helper pins ONE thread (enhance.py:12): that figure too (one clip, best of 2)
    torch.set_num_threads(1)
    run(1)
    t_one = min(run(1), run(1))
    return {'value': round(batch * seconds_per_clip / med, 3), 'unit': 'audio-sec/wall-sec', 'cores': threads, 'kind': 'port',
            'value_1_thread': round(seconds_per_clip / t_one, 3),
            'sample': f'oracle/aero_oracle.py (fp32 CPU port of the reference path), batch {batch} x {seconds_per_clip:g} s clips, '
                      f'median of 3 after warm-up, {threads} threads of {_usable_cores()} usable cores; value_1_thread: one clip, '
                      f'torch.set_num_threads(1) as the reference forces (enhance.py:12), best of 2'}


def cpu_baseline(seconds_per_clip, lr_sr, timeout_s=240):
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker'], capture_output=True,
                             text=True, timeout=timeout_s, env={**os.environ, 'HIP_VISIBLE_DEVICES': ''})
        for line in out.stdout.splitlines():
            if line.startswith('{'):
                return json.loads(line)
        return {'value': None, 'kind': 'port', 'sample': 'cpu baseline worker failed: ' + out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'kind': 'port', 'sample': f'cpu baseline worker exceeded {timeout_s} s and was stopped'}


def extra_configs(dev, steps, warmup):
    """The other single-GPU configurations of BASELINE.json, same JSON shape, so that they have driver-visible numbers:
    config 4 (12->48 kHz, nfft 1024, hop 256, batch 32, inference) and config 5's per-GPU share (11.025->44.1 kHz, nfft 512, hop 256,
    10-s segments, 2 clips per GPU of the batch of 16: ONE TRAINING STEP = forward + multi-resolution STFT loss + backward + Adam)."""
    from aero_amd import Aero, losses
    from aero_amd.optim import FlatAdam
    out = []
    try:
        torch.manual_seed(31)
        m = Aero(**dict(FULL_CFG, nfft=1024, hop_length=256, lr_sr=12000, hr_sr=48000)).eval().to(dev)
        x = torch.randn(32, 1, 24000, generator=torch.Generator().manual_seed(41)).to(dev)
        from aero_amd.pipeline import BatchPipeline
        depth = int(os.environ.get('AERO_PIPELINE', '3'))
        pipe = BatchPipeline(m, depth=depth)                    # the headline's schedule (see main): `depth` batches in flight
        with torch.no_grad():
            for _ in range(warmup):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                y = m(x)
            torch.cuda.synchronize()
            dt_serial = time.perf_counter() - t0
            for _ in range(depth):
                pipe.submit(x)
            pipe.drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                ticket = pipe.submit(x)
            pipe.drain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            y = pipe.result(ticket)
        out.append({'metric': 'real-time-factor (audio-sec/wall-sec), Aero.forward, 12->48kHz nfft=1024 hop=256 batch=32', 'value': round(32 * 2.0 * steps / dt, 2),
                    'unit': 'audio-sec/wall-sec', 'n_gpus': 1, 'steps': steps, 'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 3),
                    'ms_per_step_one_at_a_time': round(dt_serial / steps * 1e3, 3), 'dtype': 'f16',
                    'data': 'synthetic', 'config': {'workload': 'BASELINE config 4: batch=32 synthetic 2s clips, 12->48 kHz, nfft=1024 hop=256, inference',
                                                    'schedule': f'{depth} batches in flight (aero_amd/pipeline.py)' if depth > 1 else 'one forward at a time',
                                                    'frames': 376, 'output_samples': int(y.shape[-1])}})
        del pipe
        del m, x, y
    except Exception as e:                                       # an extra line must never cost the headline number
        out.append({'config': 'BASELINE config 4', 'error': repr(e)})
    try:
        torch.manual_seed(2036)
        m = Aero(**dict(FULL_CFG, nfft=512, hop_length=256, lr_sr=11025, hr_sr=44100)).to(dev).train()
        opt = FlatAdam(m.parameters(), lr=3e-4, betas=(0.9, 0.999), model=m)
        crit = losses.MultiResolutionSTFTLoss(factor_sc=0.5, factor_mag=0.5)
        g = torch.Generator().manual_seed(0)
        lr_, hr_ = torch.randn(2, 1, 110250, generator=g).to(dev), (0.1 * torch.randn(2, 1, 441000, generator=g)).to(dev)

        def step():
            y = m(lr_)
            sc, mg = crit(y.squeeze(1), hr_.squeeze(1))
            opt.zero_grad()
            (sc + mg).backward()
            opt.step()
            return (sc + mg).detach()
        k5 = max(2, min(steps, 10))                       # (10 training steps: five gave run-to-run 18.0-20.0 ms on one box)
        for _ in range(max(2, min(warmup, 4))):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k5):
            last = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the same step replayed as ONE HIP graph (aero_amd.train.CapturedStep): no host work between the ~500 library launches
        graph_ms = None
        try:
            from aero_amd.train import CapturedStep
            cap = CapturedStep(lambda a, b: step(), lr_, hr_, warmup=1, optimizers=[opt])
            cap(lr_, hr_)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k5):
                cap(lr_, hr_)
            torch.cuda.synchronize()
            graph_ms = round((time.perf_counter() - t0) / k5 * 1e3, 2)
        except Exception as e:
            graph_ms = 'capture failed: ' + repr(e)[:200]
        out.append({'metric': 'training steps per second per GPU (forward + MR-STFT loss + backward + Adam), 11.025->44.1kHz nfft=512 hop=256, 2 x 10-s clips',
                    'value': round(k5 / dt, 3), 'unit': 'steps/s', 'n_gpus': 1, 'steps': k5, 'ms_per_step': round(dt / k5 * 1e3, 2), 'ms_per_step_hip_graph': graph_ms, 'dtype': 'f16',
                    'data': 'synthetic', 'audio_sec_per_wall_sec': round(2 * 10.0 * k5 / dt, 1), 'loss': round(float(last.detach()), 5),
                    'config': {'workload': "BASELINE config 5, one GPU's share (2 of the 16 clips): aero_11-44_512_256, train mode, generator step with "
                                           'losses: [stft]; the msd_melgan critic (tools/config5.py --gan) is not part of this line', 'frames': 1724}})
    except Exception as e:
        out.append({'config': 'BASELINE config 5 (training step)', 'error': repr(e)})
    try:
        # the experiment file's own recipe (conf/experiment/aero_11-44_512_256.yaml: adversarial: true, msd_melgan): solver.py:296-320 +
        # 602-611 -- generator forward, MR-STFT + adversarial + feature-matching losses, backward, Adam; then the critic's hinge step
        from aero_amd.discriminators import Discriminator
        del m, opt
        torch.manual_seed(2036)
        m = Aero(**dict(FULL_CFG, nfft=512, hop_length=256, lr_sr=11025, hr_sr=44100)).to(dev).train()
        opt = FlatAdam(m.parameters(), lr=3e-4, betas=(0.9, 0.999), model=m)
        disc = Discriminator(num_D=3, ndf=16, n_layers=4, downsampling_factor=4).to(dev)
        opt_d = FlatAdam(disc.parameters(), lr=3e-4, betas=(0.9, 0.999), model=disc)

        def gan_step():
            y = m(lr_)
            sc, mg = crit(y.squeeze(1), hr_.squeeze(1))
            adv, feat = disc.generator_losses(y, hr_, n_layers=4, features_loss_lambda=100.0)
            opt.zero_grad()
            (sc + mg + adv + feat).backward()
            opt.step()
            d_loss = disc.discriminator_loss(y.detach(), hr_)
            opt_d.zero_grad()
            d_loss.backward()
            opt_d.step()
            return (sc + mg + adv + feat).detach(), d_loss.detach()
        for _ in range(2):
            gan_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k5):
            lg, ld = gan_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append({'metric': 'adversarial training steps per second per GPU (generator step with MR-STFT + adversarial + feature losses, then the '
                              'msd_melgan critic step), 11.025->44.1kHz nfft=512 hop=256, 2 x 10-s clips',
                    'value': round(k5 / dt, 3), 'unit': 'steps/s', 'n_gpus': 1, 'steps': k5, 'ms_per_step': round(dt / k5 * 1e3, 2), 'dtype': 'f16',
                    'data': 'synthetic', 'audio_sec_per_wall_sec': round(2 * 10.0 * k5 / dt, 1), 'g_loss': round(float(lg), 5), 'd_loss': round(float(ld), 5),
                    'config': {'workload': "BASELINE config 5 as its experiment file trains it (adversarial: true, discriminator_models: [msd_melgan]), "
                                           "one GPU's share (2 of the 16 clips)", 'frames': 1724}})
    except Exception as e:
        out.append({'config': 'BASELINE config 5 (adversarial training step)', 'error': repr(e)})
    return out


def extra_configs_subprocess(steps, warmup, timeout_s=300):
    """run extra_configs() in a child process: whatever happens there (a failed HIP-graph capture can take the process down) cannot
    cost the headline line"""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--extra-configs-worker', '--steps', str(steps), '--warmup', str(warmup)],
                             capture_output=True, text=True, timeout=timeout_s)
        for line in out.stdout.splitlines():
            if line.startswith('['):
                return json.loads(line)
        return [{'error': 'extra-configs worker produced no result', 'stderr_tail': out.stderr[-400:]}]
    except subprocess.TimeoutExpired:
        return [{'error': f'extra-configs worker exceeded {timeout_s} s and was stopped'}]


def main():
    # The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two busy streams on one queue run one
    # after the other: the serving loop's three streams plus the caller's are exactly four, and any stream object created before them --
    # RCCL's, at N > 1 -- would shift which of them share a queue (measured with two idle extra streams: 9.6-9.9 instead of 9.3 ms per
    # batch).  Eight queues leave room; read by the runtime at its first device call, so it has to be set here.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU (BASELINE config 2: 64)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--extra-configs-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-kernel-events', action='store_true', help='skip the per-launch HIP-event pass')
    ap.add_argument('--no-serial-reference', action='store_true', help='skip the one-forward-at-a-time reference loop (ms_per_step_one_at_a_time)')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the BASELINE config 4 / config 5 lines under extra_configs')
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(2.0, FULL_CFG['lr_sr'])))
        return
    if args.extra_configs_worker:
        torch.cuda.set_device(0)
        print(json.dumps(extra_configs(torch.device('cuda', 0), args.steps, args.warmup)))
        return

    from aero_amd import distrib, launcher
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU path in the product)'
    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    if torch.cuda.device_count() < args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible')
    if args.gpus > 1 and not launcher.under_launcher():
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, supervised); rank 0's
        # stdout (the JSON line) is inherited.  Under torchrun the environment is already there and we are a rank.
        ok = launcher.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.exit(0 if ok else 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    distrib.init_from_env()                                   # RCCL process group when WORLD_SIZE > 1
    ranks_verified = distrib.count_ranks(dev)                 # all-reduce of ones over RCCL: every rank really joined
    if ranks_verified != args.gpus:
        sys.exit(f'bench.py: {ranks_verified} rank(s) answered the all-reduce, expected {args.gpus}')

    from aero_amd import Aero
    torch.manual_seed(2036)
    model = Aero(**FULL_CFG).eval().to(dev)
    B, L, secs = args.batch, 8000, 2.0
    # global batch = world*B clips; clip i -> rank i mod world (reference distrib.py:100)
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, 1, L, generator=g).to(dev)

    # The timed region is the SERVING LOOP of the drop-in (aero_amd/pipeline.py, the loop aero_amd/enhance.py runs): step i is enqueued on
    # HIP stream i mod depth without waiting for step i - 1, so up to `depth` steps are in flight and the latency-bound launches of one
    # (LSTM, attention) run under the MFMA / bandwidth-bound launches of the others.  Every step is a complete forward of a whole batch,
    # all K of them finish inside the region (drain + device synchronisation before the clock stops), results are bit-identical to the
    # one-at-a-time forward.  AERO_PIPELINE=1 times the steps one at a time (model(x) in a loop); that figure is reported next to it.
    from aero_amd.pipeline import BatchPipeline
    depth = int(os.environ.get('AERO_PIPELINE', '3'))
    pipe = BatchPipeline(model, depth=depth)
    with torch.no_grad():
        for _ in range(args.warmup):                          # W untimed steps in the timed region's own schedule
            if depth > 1:
                pipe.submit(x)
            else:
                y = model(x)
        pipe.drain()
        # the same K steps one at a time (each forward on the caller's stream, the engine's own two half-batch streams inside it):
        # reported as ms_per_step_one_at_a_time (--no-serial-reference skips it: the rocprofv3 passes want one schedule per trace)
        dt_serial = float('nan')
        if not args.no_serial_reference:
            for _ in range(2):
                y1 = model(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                y1 = model(x)
            torch.cuda.synchronize()
            dt_serial = time.perf_counter() - t1
        for _ in range(depth if depth > 1 and not args.no_serial_reference else 0):   # (back in the pipelined schedule)
            pipe.submit(x)
        pipe.drain()
        distrib.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ticket = pipe.submit(x)
        pipe.drain()
        torch.cuda.synchronize()
        distrib.barrier()
        dt = time.perf_counter() - t0
        y = pipe.result(ticket)
    dt = distrib.max_over_ranks(dt, dev)
    if dt_serial == dt_serial:
        dt_serial = distrib.max_over_ranks(dt_serial, dev)
    assert y.shape == (B, 1, 4 * L) and bool(torch.isfinite(y).all())

    # ---- per-launch HIP events over K more steps: roofline of the dominant kernel ------------------------------
    roof = None
    kernels = {}
    if not args.no_kernel_events:
        eng = model._get_engine()
        eng.ops.prof = []
        with torch.no_grad():
            for _ in range(max(1, min(args.steps, 5))):
                model(x)
        torch.cuda.synchronize()
        nprof = max(1, min(args.steps, 5))
        stack = {'ms': 0.0, 'flops': 0.0}
        for kname, flops, nbytes, e0, e1, tag in eng.ops.prof:
            k = kernels.setdefault(kname, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            ms = e0.elapsed_time(e1)
            k['launches'] += 1
            k['ms'] += ms
            k['flops'] += flops
            k['bytes'] += nbytes
            if tag == 'stack' and 'conv' in kname:
                stack['ms'] += ms
                stack['flops'] += flops
        eng.ops.prof = None
        dom = max(kernels, key=lambda n: kernels[n]['ms'])
        k = kernels[dom]
        avg_ms = k['ms'] / k['launches']
        traffic, traffic_note = None, 'no PMC visit on record (profiles/pmc_traffic.json)'
        pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc):
            try:
                table = json.load(open(pmc))
                stamp = table.get('_meta', {})
                if stamp.get('kernels_sha') != kernels_sha():
                    traffic_note = (f"PMC visit {stamp.get('kernels_sha', '(unstamped)')} predates the current kernel sources {kernels_sha()}: "
                                    'omitted rather than reported stale (re-run tools/gpu/r4_evidence.sh)')
                else:
                    ent = table.get(dom.replace('void ', '').split('(')[0])
                    traffic = ent['bytes'] if ent else None   # HBM-side bytes per launch (rocprofv3 --pmc, see tools/pmc_traffic.py)
                    traffic_note = f"L2-miss-side bytes (served by MALL or HBM): rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, kernel sources {stamp['kernels_sha']}, {stamp.get('date', '')}"
            except Exception as e:
                traffic, traffic_note = None, f'pmc_traffic.json unreadable: {e}'
        if k['flops'] > 0:
            ach = k['flops'] / k['launches'] / (avg_ms * 1e-3) / 1e12
            roof = {'kernel': dom, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_MFMA_F16_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(ach / PEAK_MFMA_F16_TFLOPS, 4), 'traffic': traffic,
                    'avg_launch_ms': round(avg_ms, 4), 'launches_per_step': k['launches'] // max(1, min(args.steps, 5)),
                    'flops_per_launch': k['flops'] / k['launches'], 'traffic_source': traffic_note,
                    'note': 'executed FLOPs (2*MAC; structurally-zero first-decoder input skipped) / HIP-event time; achieved / frac: ONE stream, '
                            'one batch at a time (profiles/*_kernel_stats_1stream.csv); step_mfma_frac: all executed FLOPs of a step / ms_per_step '
                            f'of the timed region ({depth} batches in flight) / peak; traffic: bytes that crossed the L2 -> fabric boundary (TCC EA '
                            'reads x2 + writes, MALL hits included), i.e. an upper bound of the HBM bytes, per launch'}
            # the only in-product quantity that means something while `depth` batches share the chip: every executed FLOP of a step over
            # the step's wall time in the timed region (a per-launch fraction there would divide by a duration that includes waiting for
            # the other batches' launches)
            step_flops = sum(v['flops'] for v in kernels.values()) / max(1, min(args.steps, 5))
            roof.update(step_flops=step_flops, step_mfma_frac=round(step_flops / (dt / args.steps) / 1e12 / PEAK_MFMA_F16_TFLOPS, 4))
        else:
            ach = k['bytes'] / k['launches'] / (avg_ms * 1e-3) / 1e9
            roof = {'kernel': dom, 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_note, 'avg_launch_ms': round(avg_ms, 4)}

        # the conv stack as a whole (Conv2d + ConvTranspose2d of the encoder / decoder layers -- strided convs,
        # rewrite convs, transposed convs; executed FLOPs over the summed HIP-event time of exactly those launches)
        roof_stack = None
        if stack['ms'] > 0:
            ach = stack['flops'] / (stack['ms'] * 1e-3) / 1e12
            roof_stack = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_MFMA_F16_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': round(ach / PEAK_MFMA_F16_TFLOPS, 4), 'ms_per_step': round(stack['ms'] / nprof, 3),
                          'flops_per_step': stack['flops'] / nprof}
        # STFT / iSTFT against the HBM roofline (algorithmic bytes: SURVEY 8d)
        roof_stft = {}
        for kname, v in kernels.items():
            if 'stft' in kname and 'table' not in kname and v['ms'] > 0:
                ach = v['bytes'] / (v['ms'] * 1e-3) / 1e9
                roof_stft[kname] = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                    'frac': round(ach / PEAK_HBM_GBS, 4), 'avg_launch_ms': round(v['ms'] / v['launches'], 4),
                                    'bytes_per_launch': v['bytes'] / v['launches']}
        # north_star's named targets as FLAT scalars inside `roofline` (the driver's record keeps the scalar keys of `roofline` and `config`
        # and drops nested objects and extra top-level keys): conv stack >= 40 % of the MFMA peak, STFT >= 50 % of HBM, the D2 / D3 tile,
        # the latency-bound families, the one-at-a-time step next to the pipelined one
        if roof is not None:
            if roof_stack:
                roof.update(conv_stack_frac=roof_stack['frac'], conv_stack_tflops=roof_stack['achieved'], conv_stack_ms=roof_stack['ms_per_step'],
                            conv_stack_gflop_per_step=round(roof_stack['flops_per_step'] / 1e9, 1))
            for kname, v in roof_stft.items():
                pre = 'istft' if 'istft' in kname else 'stft'
                roof.update({pre + '_frac': v['frac'], pre + '_gbs': v['achieved'], pre + '_us': round(v['avg_launch_ms'] * 1e3, 1),
                             pre + '_mb_per_launch': round(v['bytes_per_launch'] / 1e6, 2)})
                if pre == 'stft' and '<2>' in kname:
                    # the fused pair (two DFT passes, normalised fp16 out) against the algorithmic bytes of the ops it REPLACES (SURVEY 8d:
                    # torch.stft 4 L + 8 F T per clip, the normalisation 4 + 2 B per value): the figure comparable with earlier rounds' stft_frac,
                    # which charged the STFT kernel alone with the fp32 spectrogram it no longer writes
                    repl = B * (L * 4 + 256 * 501 * 8) + B * 256 * 501 * 2 * 6
                    roof['stft_frac_vs_replaced_ops'] = round(repl / (v['avg_launch_ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)

            def fam(pred):
                sel = [v for n, v in kernels.items() if pred(n)]
                return sum(v['ms'] for v in sel), sum(v['flops'] for v in sel)
            ms23, fl23 = fam(lambda n: 'aero_conv_ring_kernel<2, 2, 3, 3' in n or 'aero_conv_ring_kernel<2, 4, 3, 3' in n)
            if ms23 > 0:
                roof.update(d23_tile_frac=round(fl23 / (ms23 * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS, 4), d23_tile_ms=round(ms23 / nprof, 3))
            roof.update(lstm_ms=round(fam(lambda n: 'lstm' in n)[0] / nprof, 3), localstate_ms=round(fam(lambda n: 'attn' in n)[0] / nprof, 3))
            clock = None
            try:                                              # effective clock under the dominant kernel (GRBM_GUI_ACTIVE / duration, tools/pmc_clock.py)
                ck = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_clock.json')))
                if ck.get('_meta', {}).get('kernels_sha') == kernels_sha():
                    clock = ck.get(dom.replace('void ', '').split('(')[0], {}).get('clock_ghz')
            except Exception:
                pass
            roof['clock_ghz'] = clock
            roof['ms_per_step_one_at_a_time'] = None if dt_serial != dt_serial else round(dt_serial / args.steps * 1e3, 3)
            roof['kernel_sum_ms_per_step'] = round(sum(v['ms'] for v in kernels.values()) / nprof, 3)
    else:
        roof_stack, roof_stft = None, None

    if rank != 0:
        distrib.close()
        return
    extra = None
    if world == 1 and not args.no_extra_configs:
        extra = extra_configs_subprocess(args.steps, args.warmup)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(secs, FULL_CFG['lr_sr'])
    audio_s = world * B * secs * args.steps
    # the other single-GPU configurations as FLAT scalars inside `config` (the driver's record drops nested objects and extra top-level keys)
    other = {}
    for e, key in zip(extra or [], ('config4', 'config5_train', 'config5_adv')):
        if not isinstance(e, dict) or 'ms_per_step' not in e:
            other[key + '_error'] = str((e or {}).get('error', 'no result'))[:120] if isinstance(e, dict) else 'no result'
            continue
        if key == 'config4':
            other.update(config4_ms_per_step=e['ms_per_step'], config4_rtf=e['value'], config4_ms_one_at_a_time=e.get('ms_per_step_one_at_a_time'))
        elif key == 'config5_train':
            g_ms = e.get('ms_per_step_hip_graph')
            other.update(config5_train_ms=e['ms_per_step'], config5_train_graph_ms=g_ms if isinstance(g_ms, (int, float)) else None)
        else:
            other.update(config5_adv_ms=e['ms_per_step'])
    out = {
        'metric': 'real-time-factor (audio-sec/wall-sec), Aero.forward STFT+U-Net+iSTFT, 4->16kHz nfft=512 hop=64 batch=64 per GPU',
        'value': round(audio_s / dt, 2), 'unit': 'audio-sec/wall-sec', 'n_gpus': ranks_verified, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'ms_per_step_one_at_a_time': None if dt_serial != dt_serial else round(dt_serial / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
        'config': {'workload': f'batch={B} synthetic 2s white-noise clips per GPU, 4->16 kHz, aero_4-16_512_64 '
                               f'(nfft=512 hop=64), random-init weights seed 2036, inference, inputs resident in HBM',
                   'schedule': (f'serving loop, {depth} batches in flight on {depth} HIP streams, step i+1 starts when step i enters its last encoder layer '
                                '(aero_amd/pipeline.py); all K steps complete inside the timed region; ms_per_step_one_at_a_time = model(x) in a loop')
                               if depth > 1 else 'one forward at a time (two half-batch streams inside each)',
                   'ms_per_step_one_at_a_time': None if dt_serial != dt_serial else round(dt_serial / args.steps * 1e3, 3),
                   **other,
                   'global_batch': world * B, 'clip_samples': L, 'frames': 501, 'parallelism': f'clips sharded over {world} GPU(s), no data-path collective',
                   'precision': 'fp16 operands/storage, fp32 accumulate; STFT/iSTFT/statistics fp32'},
        'roofline': roof, 'roofline_conv_stack': roof_stack, 'roofline_stft': roof_stft, 'cpu_baseline': cpu, 'extra_configs': extra,
        'kernels_ms_per_step': {n: round(v['ms'] / max(1, min(args.steps, 5)), 3) for n, v in sorted(kernels.items(), key=lambda kv: -kv[1]['ms'])},
        # per kernel over the same launches: [executed TFLOP/s, algorithmic GB/s] (HIP-event time; 0 = not applicable)
        'kernels_achieved': {n: [round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1) if v['ms'] > 0 else 0.0,
                                 round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else 0.0]
                             for n, v in sorted(kernels.items(), key=lambda kv: -kv[1]['ms'])},
    }
    print(json.dumps(out))
    distrib.close()


if __name__ == '__main__':
    main()
