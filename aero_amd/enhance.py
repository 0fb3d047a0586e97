"""Inference helpers mirroring the reference's src/enhance.py and predict.py (host plumbing around Aero.forward)."""
import math
import os

import torch

from . import audio_io

SEGMENT_DURATION_SEC = 10            # predict.py:22


def chunk_ranges(n_samples, sr, segment_sec=SEGMENT_DURATION_SEC):
    """predict.py:61-69: independent [start, end) chunks of `segment_sec` seconds, last one short (integer-exact)."""
    seg = sr * segment_sec
    n_chunks = math.ceil(n_samples / seg)
    return [(i * seg, min((i + 1) * seg, n_samples)) for i in range(n_chunks)]


def get_estimate(model, lr_sig):
    """enhance.py:11-15."""
    with torch.no_grad():
        return model(lr_sig)


MAX_CLIPS_PER_FORWARD = 64         # bound on chunk-channels per forward: activation memory stays constant for any file length


PIPELINE_DEPTH = 3                 # forwards in flight on separate HIP streams (aero_amd/pipeline.py)


def _forward_groups(model, groups, device):
    """Run the [n_i, 1, L] host batches of `groups` through the model, one forward each.  On the GPU up to PIPELINE_DEPTH groups are in
    flight, each on its own HIP stream (BatchPipeline): a group's pinned upload, its kernels and its pinned download are ordered on that
    stream and run next to the kernels and copies of the neighbouring groups (predict.py:76-80 moves one chunk at a time, synchronously)."""
    dev = torch.device(device)
    # the pipelined loop needs the generator itself (its engine, eval mode); anything else that is callable -- a distrib.wrap()'ed model, a
    # model left in training mode, a reference-style nn.Module -- takes the plain loop the reference runs (predict.py:76-80)
    core = getattr(model, 'module', model)
    if dev.type != 'cuda' or len(groups) == 0 or not hasattr(core, '_get_engine') or getattr(core, 'training', False):
        return [model(g.to(dev)).cpu() for g in groups]
    try:
        depth = max(1, int(os.environ.get('AERO_PIPELINE', PIPELINE_DEPTH)))
    except ValueError:                                   # (a malformed AERO_PIPELINE must not take the serving path down)
        depth = PIPELINE_DEPTH
    from .pipeline import BatchPipeline
    pipe = BatchPipeline(core, depth=depth)
    return pipe.run(groups, to_host=True)


def predict_signal(model, lr_sig, sr, device=None, batch_chunks=True, max_clips=MAX_CLIPS_PER_FORWARD):
    """predict.py:61-85 for one file: lr_sig [channels, samples] -> [channels, samples*scale].

    The full-length chunks are independent (predict.py:76-80 loops over them): they are batched, at most `max_clips`
    chunk-channels per forward (bounded activation memory whatever the file length), with host<->device copies
    overlapped with compute; the short tail chunk is run separately.  Results are identical per chunk.
    """
    device = device or next(model.parameters()).device
    ranges = chunk_ranges(lr_sig.shape[-1], sr)
    out = [None] * len(ranges)
    model.eval()
    ch = lr_sig.shape[0]
    with torch.no_grad():
        full = [i for i, (a, b) in enumerate(ranges) if b - a == sr * SEGMENT_DURATION_SEC]
        if batch_chunks and len(full) > 1:
            per = max(1, max_clips // ch)                                                     # chunks per forward
            idx_groups = [full[k:k + per] for k in range(0, len(full), per)]
            groups = [torch.stack([lr_sig[:, ranges[i][0]:ranges[i][1]] for i in g], 0).reshape(len(g) * ch, 1, -1)
                      for g in idx_groups]
            for g, y in zip(idx_groups, _forward_groups(model, groups, device)):
                y = y.reshape(len(g), ch, -1)
                for k, i in enumerate(g):
                    out[i] = y[k]
        for i, (a, b) in enumerate(ranges):
            if out[i] is None:
                out[i] = model(lr_sig[:, a:b].unsqueeze(1).to(device)).squeeze(1).cpu()
    return torch.cat(out, dim=-1)


def write(wav, filename, sr):
    """enhance.py:18-21: divide by max(|wav|max, 1) only if it prevents clipping, then save."""
    wav = wav / max(wav.abs().max().item(), 1)
    audio_io.save(filename, wav.cpu(), sr)


def save_wavs(processed_sigs, lr_sigs, hr_sigs, filenames, lr_sr, hr_sr):
    """enhance.py:24-29."""
    for lr, hr, pr, filename in zip(lr_sigs, hr_sigs, processed_sigs, filenames):
        write(lr, filename + '_lr.wav', sr=lr_sr)
        write(hr, filename + '_hr.wav', sr=hr_sr)
        write(pr, filename + '_pr.wav', sr=hr_sr)


def match_signal(signal, ref_len):
    """src/utils.py:211-217."""
    sig_len = signal.shape[-1]
    if sig_len < ref_len:
        signal = torch.nn.functional.pad(signal, (0, ref_len - sig_len))
    elif sig_len > ref_len:
        signal = signal[..., :ref_len]
    return signal


class _Inert:
    """Stand-in for classes a reference checkpoint pickles but the generator path never needs: the GAN critics
    (`src.models.discriminators.*`, every aero config trains with `adversarial: true`), the Seanet baseline and the
    omegaconf / hydra containers of the `args` entry (model_serializer.py:22,47).  Accepts any construction/state."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        self.__dict__['_state'] = state

    def __reduce_ex__(self, protocol):
        return (_Inert, ())


def _tolerant_pickle():
    """A `pickle_module` for torch.load whose Unpickler maps classes that cannot be imported here (or that live in the
    reference's training-only modules) to inert stubs instead of raising ModuleNotFoundError / AttributeError."""
    import importlib
    import pickle
    import types

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                root = module.split('.')[0]
                if root in ('src', 'omegaconf', 'hydra', 'antlr4', 'wandb') or module.startswith('src.'):
                    return type(name, (_Inert,), {'__module__': module})
                raise

    mod = types.ModuleType('aero_amd_tolerant_pickle')
    mod.__dict__.update({k: getattr(pickle, k) for k in dir(pickle) if not k.startswith('__')})
    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    importlib.invalidate_caches()
    return mod


def load_package(path):
    """torch.load of a checkpoint written by the reference's serializer (model_serializer.py:40-54), tolerant of the
    pickled training-only classes."""
    return torch.load(str(path), map_location='cpu', weights_only=False, pickle_module=_tolerant_pickle())


def load_generator(args, device='cuda'):
    """predict.py:24-38 / test.py:25-39: build the generator from the experiment config and load a checkpoint written
    by the reference's serializer ({'models': {'generator': {'state': ...}}, 'best_states': ..., 'args': ...}).
    Like the reference (torch.load on a missing path raises), a configured but absent checkpoint is an error; random
    weights are only used when the caller says so (`+random_init=true`), e.g. for plumbing runs and benchmarks."""
    from .modules import Aero
    model = Aero(**args.experiment.aero)
    ckpt = args.get('checkpoint_file')
    if args.get('random_init'):
        return model.to(device).eval()
    if not ckpt or not os.path.exists(str(ckpt)):
        raise FileNotFoundError(f"checkpoint_file '{ckpt}' not found (pass +random_init=true to run with random-init weights)")
    package = load_package(ckpt)
    if args.get('continue_best'):
        best = package['best_states']
        state = best['models']['generator']['state'] if 'models' in best else best['generator']
    else:
        state = package['models']['generator']['state']
    model.load_state_dict(state)
    return model.to(device).eval()
