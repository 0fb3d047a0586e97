"""MelGAN multi-scale discriminator on the MI355X (reference src/models/discriminators.py:14-78 `NLayerDiscriminator` /
`Discriminator`, the critic `msd_melgan` of solver.py:475-520).

Same constructor arguments, module tree and state-dict keys (`model.disc_{i}.model.layer_{n}.{0|1}.{weight_g, weight_v, bias}`),
same RNG stream at construction (so a seed reproduces the reference's initial weights); `forward` runs the layers on the HIP
kernels of csrc/k_disc.h (grouped strided Conv1d with bias + LeakyReLU fused, the dense k = 5 layer on aero_conv_fwd, AvgPool1d
between the scales) and returns the reference's structure: a list (scales) of lists (7 feature maps, the last one the logits),
each [B, C, T] like nn.Conv1d's output (fp16 values, channels-last storage viewed in the reference layout).
Forward only: the critic's backward (solver.py:607-611) is not built yet."""
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


class Discriminator(nn.Module):
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

    def use_library(self, lib):
        """tests: an explicitly loaded library (the CPU-emulated test double)"""
        self._ops = Ops(lib)

    def _get_ops(self):
        if self._ops is None:
            self._ops = Ops(_lib.load())
        return self._ops

    def _pack(self, dev):
        key = (str(dev),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if key == self._key:
            return self._packed
        packed = []
        for disc in self.model.values():
            layers = []
            for conv, g in zip(disc.convs(), disc.geom):
                v, gg = conv.weight_v.detach().float(), conv.weight_g.detach().float()
                w = v * (gg / v.flatten(1).norm(dim=1).view(-1, 1, 1))                 # weight norm (torch.nn.utils.weight_norm, dim 0)
                Cout, cig, K = w.shape
                ent = dict(g, Cout=Cout, Cin=cig * g['groups'], bias=conv.bias.detach().float().to(dev).contiguous())
                if g['groups'] == 1 and cig >= 64 and Cout >= 64:                        # the dense k = 5 layer: MFMA conv family
                    taps, df, dt = pack.conv1d_taps(w, 1, g['pad'])
                    ent['spec'] = pack.make_conv_spec(taps, conv.bias.detach().float(), cig, 0, df, dt, dev)
                else:
                    ent['w'] = w.permute(0, 2, 1).contiguous().to(device=dev, dtype=torch.float16)      # [Cout][K][Cin/groups]
                layers.append(ent)
            packed.append(layers)
        self._packed, self._key = packed, key
        return packed

    def forward(self, x):
        """x [B, 1, T] float waveform on the device -> list over scales of [fmap_0 .. fmap_5, logits], each [B, C, T']"""
        ops = self._get_ops()
        if not x.is_cuda and not ops.lib.is_emulator:
            raise RuntimeError('aero_amd.discriminators runs on the MI355X: move the signals to "cuda"')
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError('expected a [B, 1, T] waveform')
        dev = x.device
        packed = self._pack(dev)
        B = x.shape[0]
        cur = x.detach().reshape(B, -1).to(torch.float16).contiguous()                   # the fp16 boundary of the critic's input
        results = []
        for si, layers in enumerate(packed):
            T = cur.shape[1]
            h, Tc = cur.view(B, T, 1), T
            feats = []
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
                    ops.lib.call('aero_gconv1d_fwd', C.byref(d), ops.stream(y))
                feats.append(y.permute(0, 2, 1))                                        # [B, C, T'] view, as nn.Conv1d returns
                h, Tc = y, To
            results.append(feats)
            if si + 1 < len(packed):
                To = (T + 2 - 4) // 2 + 1
                nxt = torch.empty(B, To, dtype=torch.float16, device=dev)
                ops.lib.call('aero_avgpool1d', _ptr(cur), _ptr(nxt), B, T, ops.stream(cur))
                cur = nxt
        return results


def _loss_sum(ops, a, b, sign, mode, out):
    a = a.contiguous()
    b = None if b is None else b.contiguous()                    # (named: the buffers must outlive the call)
    n = a.numel()
    npart = min(1024, (n + 255) // 256)
    part = torch.empty(npart, dtype=torch.float64, device=a.device)
    ops.lib.call('aero_loss_sum', _ptr(a), _ptr(b), n, C.c_float(sign), mode, _ptr(part), npart, _ptr(out), ops.stream(a))


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
