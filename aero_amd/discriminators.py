"""MelGAN multi-scale discriminator on the MI355X (reference src/models/discriminators.py:14-78 `NLayerDiscriminator` /
`Discriminator`, the critic `msd_melgan` of solver.py:475-520).

Same constructor arguments, module tree and state-dict keys (`model.disc_{i}.model.layer_{n}.{0|1}.{weight_g, weight_v, bias}`),
same RNG stream at construction (so a seed reproduces the reference's initial weights); `forward` runs the layers on the HIP
kernels of csrc/k_disc.h (grouped strided Conv1d with bias + LeakyReLU fused, the dense k = 5 layer on aero_conv_fwd, AvgPool1d
between the scales) and returns the reference's structure: a list (scales) of lists (7 feature maps, the last one the logits),
each [B, C, T] like nn.Conv1d's output (fp16 values, channels-last storage viewed in the reference layout).
The critic's own step (solver.py:607-611) and the generator's adversarial / feature-matching losses are the autograd functions
`discriminator_loss` / `generator_losses` below, whose backward runs on the same kernel family; wrapped by `distrib.wrap` (solver.py:51
wraps every model) the parameter gradients are averaged over the ranks inside that backward (`_grad_sync`, one flat all-reduce)."""
import ctypes as C

import torch
from torch import nn
from torch.nn.utils import weight_norm

from . import _lib, pack
from .engine import Ops, _ptr
from .modules import capture_init


def WNConv1d(*args, **kwargs):
    return weight_norm(nn.Conv1d(*args, **kwargs))          # modules.py:10-11


def weights_init(m):
    """src/models/utils.py:38-44 (the normal_ on a weight-normed conv's derived `weight` only advances the RNG: kept for the stream)"""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find('BatchNorm2d') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


class NLayerDiscriminator(nn.Module):
    """discriminators.py:14-56 (parameters only; the forward is `Discriminator.forward`)"""

    def __init__(self, ndf, n_layers, downsampling_factor):
        super().__init__()
        model = nn.ModuleDict()
        model['layer_0'] = nn.Sequential(nn.ReflectionPad1d(7), WNConv1d(1, ndf, kernel_size=15), nn.LeakyReLU(0.2, True))
        nf = ndf
        stride = downsampling_factor
        max_nf = (stride ** (n_layers - 1)) * ndf
        self.geom = [dict(K=15, stride=1, pad=7, groups=1, reflect=1, slope=0.2)]
        for n in range(1, n_layers + 1):
            nf_prev = nf
            nf = min(nf * stride, max_nf)
            model['layer_%d' % n] = nn.Sequential(
                WNConv1d(nf_prev, nf, kernel_size=stride * 10 + 1, stride=stride, padding=stride * 5, groups=nf_prev // 4),
                nn.LeakyReLU(0.2, True))
            self.geom.append(dict(K=stride * 10 + 1, stride=stride, pad=stride * 5, groups=nf_prev // 4, reflect=0, slope=0.2))
        nf = min(nf * 2, max_nf)
        model['layer_%d' % (n_layers + 1)] = nn.Sequential(WNConv1d(nf_prev, nf, kernel_size=5, stride=1, padding=2), nn.LeakyReLU(0.2, True))
        self.geom.append(dict(K=5, stride=1, pad=2, groups=1, reflect=0, slope=0.2))
        model['layer_%d' % (n_layers + 2)] = WNConv1d(nf, 1, kernel_size=3, stride=1, padding=1)
        self.geom.append(dict(K=3, stride=1, pad=1, groups=1, reflect=0, slope=1.0))
        self.model = model

    def convs(self):
        out = []
        for key, layer in self.model.items():
            out.append(layer[1] if key == 'layer_0' else (layer[0] if isinstance(layer, nn.Sequential) else layer))
        return out

    def forward(self, x):  # pragma: no cover
        raise RuntimeError('run the critic through Discriminator.forward (HIP kernels)')


def gconv_mfma_images(w, groups, dev):
    """weights [Cout, 4, K] of a grouped stride-4 Conv1d (weight norm applied) -> the two fp16 images of the MFMA kernels
    (include/aero_hip.h, aero_gconv_desc.w_mfma / aero_gconv_bwd_desc.w_dgrad_mfma; csrc/k_gconv_mfma.h)"""
    Cout, cig, K = w.shape
    cog = Cout // groups
    assert cig == 4 and cog in (4, 16) and K <= 44
    wg = w.view(groups, cog, 4, K)
    fwd = torch.zeros(groups, 16, 192, dtype=torch.float32, device=w.device)
    fwd[:, :cog, :4 * K] = wg.permute(0, 1, 3, 2).reshape(groups, cog, 4 * K)             # column 4 k + c
    wp = torch.zeros(groups, cog, 4, 48, dtype=torch.float32, device=w.device)
    wp[..., :K] = wg
    t = wp.view(groups, cog, 4, 12, 4).permute(0, 4, 2, 3, 1)                             # [g, r, c, j, o] = W[o][c][r + 4 j]
    if cog == 16:
        dg = t.reshape(groups, 16, 192)                                                   # column 16 j + o
    else:
        t = t.reshape(groups, 16, 6, 2, 4).flip(3)                                        # octet m: (j = 2m + 1, o), (j = 2m, o)
        dg = torch.zeros(groups, 16, 64, dtype=torch.float32, device=w.device)
        dg[:, :, :48] = t.reshape(groups, 16, 48)
    cvt = lambda a: a.to(device=dev, dtype=torch.float16).contiguous()                    # noqa: E731
    return cvt(fwd), cvt(dg)


class Discriminator(nn.Module):
    _supports_grad_sync = True                                   # distrib.wrap: the backward of `discriminator_loss` averages the gradients itself

    @capture_init
    def __init__(self, num_D, ndf, n_layers, downsampling_factor):
        super().__init__()
        self.model = nn.ModuleDict()
        self.num_D = num_D
        for i in range(num_D):
            self.model[f'disc_{i}'] = NLayerDiscriminator(ndf, n_layers, downsampling_factor)
        self.downsample = nn.AvgPool1d(4, stride=2, padding=1, count_include_pad=False)
        self.apply(weights_init)
        self._ops, self._packed, self._key = None, None, None
        self._pair, self._epoch = None, 0
        self._builders = {}

    def repack(self):
        """the weights were edited behind autograd's version counters (FlatAdam's fused step): re-pack on the next forward"""
        self._pair = None
        self._epoch += 1

    def use_library(self, lib):
        """tests: an explicitly loaded library (the CPU-emulated test double)"""
        self._ops = Ops(lib)

    def _get_ops(self):
        if self._ops is None:
            self._ops = Ops(_lib.load())
        return self._ops

    def _pack(self, dev):
        """the layers' device images for the current weights.  Weight norm first (w = g v / |v|, one launch per conv into a persistent
        flat fp32 buffer), then every image -- fp16 [Cout][K][cig], the MFMA images of the grouped layers, the dense layer's conv and
        data-gradient images -- is pure data movement of w and the biases: built by their closures once, and from the first weight
        change on replayed by one gather launch per arena (aero_amd/repack.py), as the generator's training engine does."""
        dev = torch.device(dev)
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        key = (str(dev),) + tuple((p.data_ptr(), p._version) for p in self.parameters()) + (self._epoch,)
        if key == self._key:
            return self._packed
        first = self._packed is None
        ops = self._get_ops()
        convs = [(si, j, conv, g) for si, disc in enumerate(self.model.values()) for j, (conv, g) in enumerate(zip(disc.convs(), disc.geom))]
        if self._wflat is None or self._wflat.device != dev:
            offs, n = [], 0
            for _, _, conv, _ in convs:
                offs.append(n)
                n += (conv.weight_v.numel() + 3) // 4 * 4
            self._wflat, self._woffs = torch.empty(n, dtype=torch.float32, device=dev), offs
            self._replay, self._builders = None, {}
        sd = {}
        for (si, j, conv, g), o in zip(convs, self._woffs):
            v, gg = conv.weight_v.detach(), conv.weight_g.detach()
            w = self._wflat[o:o + v.numel()].view(v.shape)
            ops.lib.call('aero_weightnorm_fwd', _ptr(v.contiguous()), _ptr(gg.contiguous()), _ptr(w), v.shape[0], v.shape[1] * v.shape[2], ops.stream(w))
            sd[f'{si}.{j}.w'], sd[f'{si}.{j}.b'] = w, conv.bias.detach()
        self._sd = sd

        def builder(si, j, g):
            def build():
                w, b = self._sd[f'{si}.{j}.w'], self._sd[f'{si}.{j}.b']
                Cout, cig, K = w.shape
                ent = dict(g, Cout=Cout, Cin=cig * g['groups'], bias=b.float().to(dev).contiguous())
                if g['groups'] == 1 and cig >= 64 and Cout >= 64:                        # the dense k = 5 layer: MFMA conv family
                    from . import backward as bw
                    taps, df, dt = pack.conv1d_taps(w, 1, g['pad'])
                    ent['spec'] = pack.make_conv_spec(taps, b.float(), cig, 0, df, dt, dev)
                    ent['dgrad_spec'] = bw.dgrad_conv1d(w, 1, g['pad'], dev)
                else:
                    ent['w'] = w.permute(0, 2, 1).contiguous().to(device=dev, dtype=torch.float16)      # [Cout][K][Cin/groups]
                    if ops.lib.cdll.aero_gconv1d_mfma_ok(ent['Cin'], Cout, g['groups'], K, g['stride'], g['pad'], int(g['reflect'])):
                        ent['w_mfma'], ent['w_dgrad_mfma'] = gconv_mfma_images(w, g['groups'], dev)
                return ent
            return build
        rp = self._replay
        if rp is None and not first and self._builders and self.replay_enabled and not (
                self._wflat.is_cuda and torch.cuda.is_current_stream_capturing()):
            from .repack import WeightReplay
            rp = WeightReplay(ops.lib, ops.stream)
            rp.compile(sd, self._builders, lambda d: setattr(self, '_sd', d))
            self._replay = rp
            ents = dict(rp.objects)
        elif rp is not None:
            ents = dict(rp.refresh(sd))
        else:
            ents = {}
        packed = []
        for si, j, conv, g in convs:
            if j == 0:
                packed.append([])
            k = f'{si}.{j}'
            if k not in ents:
                if k not in self._builders:
                    self._builders[k] = builder(si, j, g)
                ents[k] = self._builders[k]()
            packed[-1].append(ents[k])
        self._packed, self._key = packed, key
        return packed

    _wflat, _woffs, _replay, _builders, _sd, replay_enabled = None, None, None, {}, None, True

    def _run(self, x):
        """-> per scale: (waveform fp16 [B, T_s], [(layer entry, input h [B,T,Cin], output y [B,T',Cout]) ...])"""
        ops = self._get_ops()
        if not x.is_cuda and not ops.lib.is_emulator:
            raise RuntimeError('aero_amd.discriminators runs on the MI355X: move the signals to "cuda"')
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError('expected a [B, 1, T] waveform')
        dev = x.device
        packed = self._pack(dev)
        B = x.shape[0]
        cur = x.detach().reshape(B, -1).to(torch.float16).contiguous()                   # the fp16 boundary of the critic's input
        scales = []
        for si, layers in enumerate(packed):
            T = cur.shape[1]
            h, Tc = cur.view(B, T, 1), T
            recs = []
            for ent in layers:
                if 'spec' in ent:
                    y = ops.conv(ent['spec'], h.view(B, 1, Tc, ent['Cin']), None, B, 1, 1, Tc).view(B, Tc, ent['Cout'])
                    ops.lib.call('aero_leaky_relu', _ptr(y), y.numel(), C.c_float(ent['slope']), ops.stream(y))
                    To = Tc
                else:
                    To = (Tc + 2 * ent['pad'] - ent['K']) // ent['stride'] + 1
                    y = torch.empty(B, To, ent['Cout'], dtype=torch.float16, device=dev)
                    d = _lib.GconvDesc()
                    d.x, d.w, d.bias, d.y = _ptr(h), _ptr(ent['w']), _ptr(ent['bias']), _ptr(y)
                    d.B, d.Tin, d.Cin, d.Cout, d.groups, d.K, d.stride, d.pad, d.reflect = B, Tc, ent['Cin'], ent['Cout'], ent['groups'], ent['K'], \
                        ent['stride'], ent['pad'], ent['reflect']
                    d.slope = ent['slope']
                    d.w_mfma = _ptr(ent.get('w_mfma'))
                    ops.lib.call('aero_gconv1d_fwd', C.byref(d), ops.stream(y))
                recs.append((ent, h, y))
                h, Tc = y, To
            scales.append((cur, recs))
            if si + 1 < len(packed):
                To = (T + 2 - 4) // 2 + 1
                nxt = torch.empty(B, To, dtype=torch.float16, device=dev)
                ops.lib.call('aero_avgpool1d', _ptr(cur), _ptr(nxt), B, T, ops.stream(cur))
                cur = nxt
        return scales

    def _run_pair(self, fake, real):
        """D(fake) and D(real) as ONE batch of 2B signals (the reference runs the critic twice per loss, solver.py:478-480,505-506: the
        same arithmetic per signal, half the launches), kept until the weights or the signals change: the critic's own step
        (solver.py:607-611) evaluates D on exactly the signals and weights the generator's adversarial / feature losses just used, so
        its forward pass is this record again.  Returns (record of the 2B batch, B)."""
        if fake.shape != real.shape:
            raise ValueError('fake and real must have the same shape')
        key = (fake.data_ptr(), fake._version, real.data_ptr(), real._version, tuple(fake.shape), str(fake.device)) + \
            tuple((p.data_ptr(), p._version) for p in self.parameters()) + (self._epoch,)
        if self._pair is None or self._pair[0] != key:
            # (the record keeps the two signals alive: while it is cached their memory cannot be recycled for other data at the same
            # address and version -- a key built from pointers alone would then hit a stale record, e.g. in a validation loop)
            self._pair = (key, self._run(torch.cat([fake.detach(), real.detach()], 0)), fake.detach(), real.detach())
        return self._pair[1], fake.shape[0]

    @staticmethod
    def _half(runs, lo, hi):
        """the record of batch items [lo, hi) of a run (views)"""
        return [(cur[lo:hi], [(ent, h[lo:hi], y[lo:hi]) for (ent, h, y) in recs]) for (cur, recs) in runs]

    def forward(self, x):
        """x [B, 1, T] float waveform on the device -> list over scales of [fmap_0 .. fmap_5, logits], each [B, C, T'] (values; the
        differentiable entry points are `discriminator_loss` and `generator_losses`)"""
        return [[y.permute(0, 2, 1) for (_, _, y) in recs] for (_, recs) in self._run(x)]

    # ------------------------------------------------------------------ losses with their HIP backward (solver.py:475-520)
    def discriminator_loss(self, fake, real):
        """solver.py:489-496: sum over scales of relu(1 + D(fake)).mean() + relu(1 - D(real)).mean(); differentiable w.r.t. the
        critic's parameters (the generator output is detached, solver.py:479)"""
        names, params = zip(*self.named_parameters())
        return _CriticLoss.apply(self, names, fake.detach(), real.detach(), *params)

    def generator_losses(self, fake, real, n_layers=4, features_loss_lambda=100.0):
        """solver.py:498-520: (adversarial = sum relu(1 - D(fake)).mean(), lambda * feature matching); differentiable w.r.t. `fake`"""
        return _GeneratorLoss.apply(self, fake, real.detach(), n_layers, features_loss_lambda)

    def _backward(self, runs, dtop, dfeat, want_params, want_input, out=None, gl=None):
        """runs: _run() record; dtop[s]: (gradient of scale s's logits fp16 [B,T',1], {S,1/S}); dfeat[s][j]: the same for feature map j
        or None.  Returns ({parameter name: fp32 gradient of weight_g / weight_v / bias}, d waveform fp32 [B,T] or None).
        out: {parameter name: fp32 destination} the parameter gradients are ADDED to (views of a flat gradient buffer) instead of being
        returned as new tensors; gl: 0-dim fp32 device tensor, the upstream factor of the loss (folded into the same kernel)."""
        from . import backward as bw, train_ops as TO
        ops = self._get_ops()
        grads = {}
        dwave = None                                             # (tensor fp16 [B, T_s], scale) flowing from the coarser scales
        for si in reversed(range(len(runs))):
            cur, recs = runs[si]
            B = cur.shape[0]
            disc = self.model[f'disc_{si}']
            keys = list(disc.model.keys())
            g, sc = dtop[si]
            dx = None
            for j in reversed(range(len(recs))):
                ent, h, y = recs[j]
                if j < len(recs) - 1:
                    f = dfeat[si][j] if dfeat is not None else None
                    if f is not None:
                        g, sc = TO.rescale_f16(ops, dx, sc, f[0], f[1])
                    else:
                        g, sc = TO.rescale_f16(ops, dx, sc)
                Tin, To = h.shape[1], y.shape[1]
                conv = disc.convs()[j]
                prefix = f'model.disc_{si}.model.{keys[j]}.' + ('1.' if keys[j] == 'layer_0' else ('0.' if isinstance(disc.model[keys[j]], nn.Sequential) else ''))
                need_dx = want_input or j > 0
                if 'spec' in ent:
                    dyp = torch.empty_like(g)
                    ops.lib.call('aero_loss_grad', _ptr(g), _ptr(y), g.numel(), C.c_float(0.0), C.c_float(ent['slope']), 2, _ptr(dyp), None, ops.stream(g))
                    if want_params:
                        spec = ent['spec']
                        dwk, db = bw.conv_wgrad(ops, dyp.view(B, 1, To, ent['Cout']), h.view(B, 1, Tin, ent['Cin']), spec.df, spec.dt)
                        dw_strides = (ent['Cin'], 1, ent['Cout'] * ent['Cin'])      # [K, Cout, Cin]: element (o, c, k)
                    dx = ops.conv(ent['dgrad_spec'], dyp.view(B, 1, To, ent['Cout']), None, B, 1, 1, To).view(B, Tin, ent['Cin']) \
                        if need_dx else None
                else:
                    d = _lib.GconvBwdDesc()
                    dx = torch.empty(B, Tin, ent['Cin'], dtype=torch.float16, device=g.device) if need_dx else None
                    if want_params:
                        dwk = torch.zeros(ent['Cout'], ent['K'], ent['Cin'] // ent['groups'], dtype=torch.float32, device=g.device)
                        db = torch.zeros(max(ent['Cout'], 4), dtype=torch.float32, device=g.device)[:ent['Cout']]   # (room for a float4)
                    d.x, d.w, d.y, d.dy, d.dx = _ptr(h), _ptr(ent['w']), _ptr(y), _ptr(g), _ptr(dx)
                    d.dw, d.db = (_ptr(dwk), _ptr(db)) if want_params else (None, None)
                    d.B, d.Tin, d.Cin, d.Cout, d.groups, d.K, d.stride, d.pad, d.reflect = B, Tin, ent['Cin'], ent['Cout'], ent['groups'], ent['K'], \
                        ent['stride'], ent['pad'], ent['reflect']
                    d.slope = ent['slope']
                    d.w_dgrad_mfma = _ptr(ent.get('w_dgrad_mfma'))
                    if want_params:                              # per-chunk slabs added in order, where the library has that form
                        nsl = ops.lib.cdll.aero_gconv1d_wgrad_slabs(B, Tin, ent['Cin'], ent['Cout'], ent['groups'], ent['K'], ent['stride'], ent['pad'],
                                                                    int(ent['reflect']))
                        if nsl > 0:
                            slabs = torch.empty(nsl, ent['Cout'] * ent['K'] * (ent['Cin'] // ent['groups']) + max(ent['Cout'], 4),
                                                dtype=torch.float32, device=g.device)
                            d.slabs, d.nslab = _ptr(slabs), nsl
                    ops.lib.call('aero_gconv1d_bwd', C.byref(d), ops.stream(g))
                    if want_params:
                        cig_ = ent['Cin'] // ent['groups']
                        dw_strides = (ent['K'] * cig_, 1, cig_)                      # [Cout, K, cig]: element (o, c, k)
                if want_params:
                    # weight norm (w = g v / |v| per output channel), the 1 / S of the fp16 gradient path and the upstream loss factor in
                    # ONE launch (aero_weightnorm_bwd) -- the same bookkeeping in torch ops was ~20 parameter-sized kernels per conv
                    v, gg = conv.weight_v.detach(), conv.weight_g.detach()
                    assert v.dtype == torch.float32 and v.is_contiguous() and gg.is_contiguous()
                    names3 = (prefix + 'weight_g', prefix + 'weight_v', prefix + 'bias')
                    if out is not None:
                        dg_, dv_, dbias_ = (out[n] for n in names3)
                        acc = 1
                    else:
                        dg_, dv_, dbias_ = torch.empty_like(gg), torch.empty_like(v), torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
                        acc = 0
                        grads[names3[0]], grads[names3[1]], grads[names3[2]] = dg_, dv_, dbias_
                    ops.lib.call('aero_weightnorm_bwd', _ptr(dwk), dw_strides[0], dw_strides[1], dw_strides[2], _ptr(v), _ptr(gg), _ptr(db),
                                 sc[1:].data_ptr(), _ptr(gl), _ptr(dg_), _ptr(dv_), _ptr(dbias_), v.shape[0], v.shape[1], v.shape[2], acc, ops.stream(v))
            if want_input:
                dxw = dx.view(B, -1)                             # gradient of this scale's waveform
                if dwave is not None:                            # + the coarser scales through the AvgPool1d between them
                    up = torch.empty(B, cur.shape[1], dtype=torch.float16, device=g.device)
                    ops.lib.call('aero_avgpool1d_bwd', _ptr(dwave[0]), _ptr(up), B, cur.shape[1], ops.stream(up))
                    dwave = TO.rescale_f16(ops, dxw.contiguous(), sc, up, dwave[1])
                else:
                    dwave = (dxw.contiguous(), sc)
        out = None
        if want_input:
            out = dwave[0].float()
            TO.scale_f32(ops, out, dwave[1][1:])
        return grads, out

    @staticmethod
    def _wn(conv):
        v, gg = conv.weight_v.detach().float(), conv.weight_g.detach().float()
        return v * (gg / v.flatten(1).norm(dim=1).view(-1, 1, 1))


def _scaled_grad(ops, a, b, n_mean, sign, coef, mode, out=None, gl=None):
    """gradient of coef * mean(...) as fp16 with a host-chosen power-of-two scale: returns (tensor, {S, 1/S} on the device)"""
    import math
    c = coef / n_mean
    S = 2.0 ** round(math.log2(32.0 / max(abs(c), 1e-30)))
    g = torch.empty(a.shape, dtype=torch.float16, device=a.device) if out is None else out
    assert a.is_contiguous() and g.is_contiguous() and (b is None or b.is_contiguous())
    ops.lib.call('aero_loss_grad', _ptr(a), _ptr(b), a.numel(), C.c_float(sign), C.c_float(c * S), mode, _ptr(g), _ptr(gl), ops.stream(a))
    return g, _scale_pair(S, a.device)


_SCALES = {}


def _scale_pair(S, dev):
    """{S, 1/S} on the device (cached: host-chosen powers of two, a handful of distinct values)"""
    key = (S, str(dev))
    if key not in _SCALES:
        _SCALES[key] = torch.tensor([S, 1.0 / S], dtype=torch.float32, device=dev)
    return _SCALES[key]


class _CriticLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disc, names, fake, real, *params):
        ops = disc._get_ops()
        runs, B = disc._run_pair(fake, real)
        loss = torch.zeros(1, dtype=torch.float64, device=fake.device)
        for (_, recs) in runs:
            logits = recs[-1][2]
            w = 1.0 / logits[:B].numel()                         # (the means and the sum over scales accumulate in one device scalar)
            _loss_sum(ops, logits[:B], None, 1.0, 0, loss, w)
            _loss_sum(ops, logits[B:], None, -1.0, 0, loss, w)
        ctx.disc, ctx.names, ctx.runs, ctx.B = disc, names, runs, B
        ctx.param_ptrs, ctx.shapes = [p.data_ptr() for p in params], [p.shape for p in params]
        return loss[0].float()

    @staticmethod
    def backward(ctx, gl):
        disc, ops = ctx.disc, ctx.disc._get_ops()
        B = ctx.B
        # one backward pass over the 2B batch: d relu(1 + D(fake)).mean() on the first half, d relu(1 - D(real)).mean() on the second
        dtop = []
        for (_, recs) in ctx.runs:
            logits = recs[-1][2]
            g = torch.empty_like(logits)
            _, sc = _scaled_grad(ops, logits[:B], None, logits[:B].numel(), 1.0, 1.0, 0, out=g[:B])
            _scaled_grad(ops, logits[B:], None, logits[B:].numel(), -1.0, 1.0, 0, out=g[B:])
            dtop.append((g, sc))
        # FlatAdam keeps every parameter's .grad as a view of one flat buffer: write there (freshly zeroed by zero_grad) and hand autograd no
        # per-parameter gradients (its AccumulateGrad nodes were one `grad += g` launch per parameter) -- as aero_amd.train.AeroFunction does
        glf = gl.detach().float().contiguous()
        # distrib.wrap(critic) (solver.py:51): the mean over ranks.  1 / world rides in the upstream factor the weight-norm kernel
        # multiplies in anyway; the sum is ONE all-reduce over the flat gradient range once the pass is done (the critic's backward is a
        # few milliseconds: nothing to overlap it with but the optimizer step that needs its result)
        sync = getattr(disc, '_grad_sync', None)
        if sync is not None and not sync.active():
            sync = None
        if sync is not None:
            glf = glf * sync.mean_factor()
        sink = getattr(disc, '_grad_sink', None)
        sink = sink() if sink is not None else None
        offs, n = [], 0
        for shp in ctx.shapes:
            offs.append(n)
            n += (shp.numel() + 3) // 4 * 4
        params = dict(disc.named_parameters())
        # (a buffer that already holds gradients must not go through the collective a second time: then this pass gets its own tensors)
        if sink is not None and sink.accepts(ctx.param_ptrs, offs, n, glf.device) and (sync is None or sink.fresh):
            out = {nme: params[nme].grad for nme in ctx.names}
            sink.fresh = False
            disc._backward(ctx.runs, dtop, None, True, False, out=out, gl=glf)
            ctx.runs = None
            if sync is not None:
                sync.reduce_async(sink.flat_g)
                sync.wait()
            return (None, None, None, None) + (None,) * len(ctx.names)
        total, _ = disc._backward(ctx.runs, dtop, None, True, False, gl=glf)
        ctx.runs = None
        if sync is not None:
            flat = torch.cat([total[nme].reshape(-1) for nme in ctx.names])
            sync.reduce_async(flat)
            sync.wait()
            o = 0
            for nme in ctx.names:
                k = total[nme].numel()
                total[nme] = flat[o:o + k].view_as(total[nme])
                o += k
        return (None, None, None, None) + tuple(total[n] for n in ctx.names)


class _GeneratorLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disc, fake, real, n_layers, lam):
        ops = disc._get_ops()
        runs, B = disc._run_pair(fake, real)
        rf, rr = disc._half(runs, 0, B), disc._half(runs, B, 2 * B)
        num_D = len(rf)
        w_feat = (4.0 / (n_layers + 1)) * (1.0 / num_D)
        acc = torch.zeros(2, dtype=torch.float64, device=fake.device)        # {adversarial, lambda * feature matching}
        for (_, a), (_, b) in zip(rf, rr):
            _loss_sum(ops, a[-1][2], None, -1.0, 0, acc[0:1], 1.0 / a[-1][2].numel())
            for j in range(len(a) - 1):
                _loss_sum(ops, a[j][2], b[j][2], 0.0, 1, acc[1:2], lam * w_feat / a[j][2].numel())
        ctx.disc, ctx.runs, ctx.cfg, ctx.shape = disc, (rf, rr), (w_feat, lam), fake.shape
        out = acc.float()
        return out[0], out[1]

    @staticmethod
    def backward(ctx, gadv, gfeat):
        from . import train_ops as TO
        disc, ops = ctx.disc, ctx.disc._get_ops()
        rf, rr = ctx.runs
        w_feat, lam = ctx.cfg
        # the upstream factors (1 in solver.py:314-316) stay on the device: the loss-gradient kernel multiplies them in (a float() here
        # would stall the host in the middle of the generator's backward until the device had caught up)
        ga, gf = gadv.detach().float().contiguous(), gfeat.detach().float().contiguous()
        dtop = [_scaled_grad(ops, recs[-1][2], None, recs[-1][2].numel(), -1.0, 1.0, 0, gl=ga) for (_, recs) in rf]
        dfeat = [[_scaled_grad(ops, ra[j][2], rb[j][2], ra[j][2].numel(), 0.0, lam * w_feat, 1, gl=gf) for j in range(len(ra) - 1)]
                 for (_, ra), (_, rb) in zip(rf, rr)]
        _, dx = disc._backward(rf, dtop, dfeat, False, True)
        ctx.runs = None
        return None, dx.view(ctx.shape), None, None, None


def _loss_sum(ops, a, b, sign, mode, out, weight=1.0):
    a = a.contiguous()
    b = None if b is None else b.contiguous()                    # (named: the buffers must outlive the call)
    n = a.numel()
    npart = min(1024, (n + 255) // 256)
    part = torch.empty(npart, dtype=torch.float64, device=a.device)
    ops.lib.call('aero_loss_sum', _ptr(a), _ptr(b), n, C.c_float(sign), mode, _ptr(part), npart, _ptr(out), C.c_double(weight), ops.stream(a))


def melgan_losses(disc, fake, real, n_layers=4, num_D=3, features_loss_lambda=100.0):
    """solver.py:489-520 on the critic's HIP outputs (values): returns (discriminator hinge loss, generator adversarial loss,
    lambda * feature-matching loss) as 0-dim device tensors; fake / real: the lists `Discriminator.forward` returned."""
    ops = disc._get_ops()
    dev = fake[0][-1].device
    acc = torch.zeros(3, dtype=torch.float64, device=dev)
    w_feat = (4.0 / (n_layers + 1)) * (1.0 / num_D)
    d_loss = torch.zeros((), dtype=torch.float64, device=dev)
    g_adv = torch.zeros((), dtype=torch.float64, device=dev)
    g_feat = torch.zeros((), dtype=torch.float64, device=dev)
    for sf, sr in zip(fake, real):
        n = sf[-1].numel()
        acc.zero_()
        _loss_sum(ops, sf[-1], None, 1.0, 0, acc[0:1])            # relu(1 + fake)
        _loss_sum(ops, sr[-1], None, -1.0, 0, acc[1:2])           # relu(1 - real)
        _loss_sum(ops, sf[-1], None, -1.0, 0, acc[2:3])           # relu(1 - fake)
        d_loss = d_loss + (acc[0] + acc[1]) / n
        g_adv = g_adv + acc[2] / n
        for j in range(len(sf) - 1):
            acc.zero_()
            _loss_sum(ops, sf[j], sr[j], 0.0, 1, acc[0:1])
            g_feat = g_feat + w_feat * acc[0] / sf[j].numel()
    return d_loss.float(), g_adv.float(), (features_loss_lambda * g_feat).float()
