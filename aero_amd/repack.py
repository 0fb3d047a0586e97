"""Weight images of the training engine, re-packed after every optimizer step by ONE gather launch per arena.

The training step re-packs every convolution's weights (padded fp16 images, tiled copies for the ring kernel, flipped / transposed
images for the data gradients, ...) after each optimizer step: in torch that was ~700 small layout kernels per step, a quarter of the
device time at BASELINE config 5's per-GPU batch.  All of it is pure data movement -- element i of an image is some element of some
parameter, or zero -- so it can be replayed by a gather: `aero_gather_pack` (csrc/k_train.h) with an index table per arena.

The tables are not written by hand: they are DERIVED from the packing code itself.  Every `TrainEngine.spec(key, build)` closure reads
the parameters through `engine.w(name)`; run on probe parameters whose values are the base-2048 digits of their own flat index
(exact in fp16), the images it returns spell out, digit by digit, where each element came from.  A table is accepted only if replaying
it reproduces the closure's output BIT FOR BIT on the real parameters and on a set of random ones; anything else (scaled or summed
weights, closures that read tensors they captured) keeps being rebuilt by its closure, exactly as before.
"""
import dataclasses

import torch

DIG = 2048


def _flatten(obj, out):
    """tensors of a spec object in a fixed traversal order (dataclasses, tuples, lists, dicts); everything else is static"""
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        for f in dataclasses.fields(obj):
            _flatten(getattr(obj, f.name), out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _flatten(v, out)
    elif isinstance(obj, dict):
        for v in obj.values():
            _flatten(v, out)
    return out


def _rebuild(obj, it):
    """the same object with its tensors replaced, in traversal order, by next(it)"""
    if isinstance(obj, torch.Tensor):
        return next(it)
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return dataclasses.replace(obj, **{f.name: _rebuild(getattr(obj, f.name), it) for f in dataclasses.fields(obj)})
    if isinstance(obj, tuple):
        return tuple(_rebuild(v, it) for v in obj)
    if isinstance(obj, list):
        return [_rebuild(v, it) for v in obj]
    if isinstance(obj, dict):
        return {k: _rebuild(v, it) for k, v in obj.items()}
    return obj


class WeightReplay:
    def __init__(self, lib, stream_of):
        self.lib, self.stream_of = lib, stream_of
        self.objects = {}                                       # key -> spec object whose tensors are views of the arenas
        self.skipped = {}                                       # key -> why its closure stays in charge
        self.arena = {}                                         # dtype -> (arena tensor, int32 table)
        self.names, self.ptr_key = [], None

    # ---------------------------------------------------------------- compile
    def compile(self, sd, builders, set_sd):
        """sd: the engine's current parameter dict (name -> tensor); builders: key -> closure; set_sd(d): make engine.w read d"""
        names = [k for k, v in sd.items() if v.dtype == torch.float32 and v.numel() > 0]
        dev = sd[names[0]].device
        sizes = [sd[k].numel() for k in names]
        starts = [0]
        for n in sizes:
            starts.append(starts[-1] + n)
        total = starts[-1]
        if total + 1 >= 2 ** 31 or len(names) > 1024:
            self.skipped['*'] = 'parameter space too large for int32 tables'
            return False
        fi = torch.arange(1, total + 1, dtype=torch.int64, device=dev)
        digits = [((fi // DIG ** d) % DIG).to(torch.float32) for d in range(3)]
        g = torch.Generator(device='cpu').manual_seed(20360)
        rnd = torch.randn(total, generator=g).to(dev)
        real = torch.cat([sd[k].detach().reshape(-1) for k in names])

        def as_sd(flat):
            d = dict(sd)
            for k, a, n in zip(names, starts, sizes):
                d[k] = flat[a:a + n].view(sd[k].shape)
            return d
        probe_sds = [as_sd(f) for f in digits]
        rnd_sd = as_sd(rnd)
        # a closure that reads a parameter tensor it CAPTURED (instead of asking engine.w) ignores the probes: its output looks like
        # a constant.  One more run with the live parameters themselves overwritten tells the two apart (restored bit for bit below).
        live = {}
        try:
            with torch.no_grad():
                for k, a, n in zip(names, starts, sizes):
                    sd[k].copy_(rnd[a:a + n].view(sd[k].shape))
            set_sd(probe_sds[0])
            for key, build in builders.items():
                try:
                    live[key] = _flatten(build(), [])
                except Exception:                               # noqa: BLE001  (reported by the main pass)
                    live[key] = None
        finally:
            with torch.no_grad():
                for k, a, n in zip(names, starts, sizes):
                    sd[k].copy_(real[a:a + n].view(sd[k].shape))
            set_sd(sd)
        plans = {}
        for key, build in builders.items():
            try:
                set_sd(sd)
                obj = build()
                outs = _flatten(obj, [])
                probes = []
                for psd in probe_sds:
                    set_sd(psd)
                    probes.append(_flatten(build(), []))
                set_sd(rnd_sd)
                outs_rnd = _flatten(build(), [])
            except Exception as e:                              # noqa: BLE001  (a closure that cannot run on substitute parameters)
                self.skipped[key] = f'closure failed on probe parameters: {type(e).__name__}: {e}'
                continue
            finally:
                set_sd(sd)
            why, tables = None, []
            if not outs:
                why = 'no tensors'
            for i, t in enumerate(outs):
                if why:
                    break
                ps = [p[i] if i < len(p) else None for p in probes] + [outs_rnd[i] if i < len(outs_rnd) else None]
                if any(p is None or p.shape != t.shape or p.dtype != t.dtype for p in ps) or any(len(p) != len(outs) for p in probes):
                    why = 'structure changes with the parameters'
                    break
                if all(torch.equal(p, t) for p in ps):
                    lv = live.get(key)
                    if lv is None or i >= len(lv) or lv[i].shape != t.shape or not torch.equal(lv[i], t):
                        why = f'output {i} follows a captured parameter tensor, not engine.w'
                        break
                    tables.append(None)                         # a static tensor (index tables, constants): kept as it is
                    continue
                if t.dtype not in (torch.float16, torch.float32) or t.device != dev:
                    why = f'output {i}: dtype {t.dtype} on {t.device}'
                    break
                code = torch.zeros(t.shape, dtype=torch.int64, device=dev)
                ok = True
                for d, p in enumerate(ps[:3]):
                    pd = p.double()
                    ok = ok and bool(((pd == pd.round()) & (pd >= 0) & (pd < DIG)).all())
                    code += pd.round().to(torch.int64) * DIG ** d
                if not ok or int(code.max()) > total:
                    why = f'output {i} is not a copy of parameter elements'
                    break
                idx = code - 1                                  # -1: zero fill
                take = idx.clamp(min=0)
                for flat, want in ((rnd, outs_rnd[i]), (real, t)):
                    rep = torch.where(idx >= 0, flat[take], torch.zeros((), device=dev)).to(t.dtype)
                    if not torch.equal(rep, want):
                        why = f'output {i}: replay differs from the closure'
                        break
                tables.append(idx.reshape(-1).to(torch.int32).contiguous() if t.is_contiguous() else None)
                if why is None and not t.is_contiguous():
                    why = f'output {i} is not contiguous'
            if why:
                self.skipped[key] = why
            else:
                plans[key] = (obj, outs, tables)
        # arenas: every replayed tensor becomes a 64-byte aligned view
        count = {torch.float16: 0, torch.float32: 0}
        for obj, outs, tables in plans.values():
            for t, tb in zip(outs, tables):
                if tb is not None:
                    count[t.dtype] += (t.numel() + 31) // 32 * 32
        for dt, n in count.items():
            if n:
                self.arena[dt] = (torch.zeros(n, dtype=dt, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev))
        fill = {torch.float16: 0, torch.float32: 0}
        for key, (obj, outs, tables) in plans.items():
            views = []
            for t, tb in zip(outs, tables):
                if tb is None:
                    views.append(t)
                    continue
                ar, tab = self.arena[t.dtype]
                a = fill[t.dtype]
                ar[a:a + t.numel()].copy_(t.reshape(-1))
                tab[a:a + t.numel()].copy_(tb)
                views.append(ar[a:a + t.numel()].view(t.shape))
                fill[t.dtype] += (t.numel() + 31) // 32 * 32
            self.objects[key] = _rebuild(obj, iter(views))
        self.names = names
        self.starts = torch.tensor(starts, dtype=torch.int32, device=dev)
        self.ptrs = torch.zeros(len(names), dtype=torch.int64, device=dev)
        self.ptr_key = None
        self._upload_ptrs(sd)
        return True

    def _upload_ptrs(self, sd):
        key = tuple(sd[k].data_ptr() for k in self.names)
        if key != self.ptr_key:
            if any(not sd[k].is_contiguous() or sd[k].dtype != torch.float32 for k in self.names):
                raise RuntimeError('WeightReplay: parameters must stay contiguous fp32 tensors')
            self.ptrs.copy_(torch.tensor(key, dtype=torch.int64))
            self.ptr_key = key

    def matches(self, sd):
        return all(k in sd and sd[k].numel() > 0 for k in self.names)

    # ---------------------------------------------------------------- every step
    def refresh(self, sd):
        """bring every replayed image up to date with the parameters in sd (two kernel launches)"""
        self._upload_ptrs(sd)
        for dt, (ar, tab) in self.arena.items():
            self.lib.call('aero_gather_pack', self.ptrs.data_ptr(), self.starts.data_ptr(), len(self.names), tab.data_ptr(), ar.data_ptr(),
                          ar.numel(), int(dt == torch.float16), self.stream_of(ar))
        return self.objects
