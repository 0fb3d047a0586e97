"""One training step of the generator on the MI355X: the training-mode forward of `Aero` with everything the backward needs
kept on the device, and the hand-written reverse pass -- reference `loss.backward()` of src/solver.py:602-605 through
src/models/aero.py:108-135 (HEncLayer), :189-215 (HDecLayer), :446-523 (Aero.forward) and src/models/modules.py:32-65 (BLSTM),
:94-127 (LocalState), :221-249 (DConv), :304-325 (FTB).

Host plumbing only: every arithmetic step is a C-ABI kernel call (include/aero_hip.h) -- the forward family (`aero_conv_fwd`,
`aero_norm_*`, `aero_lstm_fwd`, `aero_localstate_fwd`, `aero_freqfc_fwd`, STFT / iSTFT), the backward kernels of csrc/k_bwd.h
(`aero_conv_wgrad`, `aero_norm_bwd_*`, iSTFT adjoint; data gradients are forward convolutions on re-packed weights,
aero_amd/backward.py) and of csrc/k_train.h (LSTM BPTT, LocalState, FTB gate / freq_fc, frequency embedding, the fp16 boundary).
torch supplies device memory, views / permutes of parameter-sized tensors (layout changes, no arithmetic) and the autograd
hook (`AeroFunction`).  There is no CPU / eager fallback.

Layer by layer rather than on the fused inference kernels: the backward needs the pre-normalisation and pre-activation tensors
the fused forms never write.  Activations and their gradients are fp16 (fp32 accumulation); the gradient enters the fp16 domain
multiplied by a power-of-two loss scale chosen from its own maximum (`aero_scale_cast`) and every parameter gradient is
un-scaled in fp32 at the end (`aero_scale_f32`), so nothing depends on the magnitude of the loss.
"""
import contextlib
import ctypes as C
import math
import os

import torch

from . import _lib, backward as bw, pack, train_ops as TO
from ._lib import ACT_GELU, ACT_GLU, ACT_NONE, ACT_RELU, ACT_SNAKE
from .engine import Ops, _hann_padded, blstm_frames

GRAD_TARGET = 64.0          # a stage's input gradient has its largest magnitude in [32, 64]: three orders of magnitude of headroom
                            # to the fp16 maximum for growth inside the stage, six down to the smallest normal number


def _r8(n):
    return (n + 7) // 8 * 8


def lstm_param_grads(ops, da, x, out, layer, H, nseq, W, in_ch, dev, put):
    """Parameter gradients of one bidirectional nn.LSTM layer from da = d(gate pre-activations) (aero_lstm_bwd): da fp16 [nseq*W, 2, 4H]
    (column 4j + gate), x fp16 [nseq, W, in_ch] the layer input, out fp16 [nseq, W, 2H] its output.  put(name, rows, perm) receives
    the gradient of nn.LSTM parameter `name` as rows in KERNEL order: destination row perm[i] = rows[i]."""
    H4 = 4 * H
    perm = pack.lstm_gate_perm(H, dev)                       # kernel row 4j+g <- nn.LSTM row g*H+j
    dw, db = bw.conv_wgrad(ops, da.view(nseq, 1, W, 2 * H4), x.reshape(nseq, 1, W, in_ch), [0], [0])      # [1, 8H, in_ch], [8H]
    for dr, sfx in enumerate(('', '_reverse')):
        put(f'weight_ih_l{layer}{sfx}', dw[0, dr * H4:(dr + 1) * H4], perm)
        put(f'bias_ih_l{layer}{sfx}', db[dr * H4:(dr + 1) * H4], perm)
        put(f'bias_hh_l{layer}{sfx}', db[dr * H4:(dr + 1) * H4], perm)
        dav = da.view(nseq, W, 2, H4)[:, :, dr, :].unsqueeze(1)                       # [nseq, 1, W, 4H] strided
        hv = out.view(nseq, W, 2 * H)[:, :, dr * H:(dr + 1) * H].unsqueeze(1)         # h of this direction
        if H % 8:
            dav, hv = dav.contiguous(), hv.contiguous()
        dwh, _ = bw.conv_wgrad(ops, dav, hv, [0], [1 if dr else -1], bias=False)      # h_{prev}: the step before in this direction
        put(f'weight_hh_l{layer}{sfx}', dwh[0], perm)


class _Ctx:
    """what one forward keeps for its backward"""
    pass


class TrainEngine:
    def __init__(self, model, lib=None):
        self.model = model
        self.lib = lib if lib is not None else _lib.load()
        self.ops = Ops(self.lib)
        self._cache, self._key = {}, None
        self._tables = {}
        self._epoch = 0
        self._boosts, self._nfwd = {}, 0
        self._builders = {}

    # ------------------------------------------------------------------ weights (packed on the device, per parameter version)
    def invalidate(self):
        self._epoch += 1

    def _sync_weights(self, dev):
        key = (str(dev), self._epoch) + tuple((p.data_ptr(), p._version) for p in self.model.parameters())
        if key == self._key:
            return
        first = self._key is None
        self._key = key
        self.sd = {k: v.detach() for k, v in self.model.state_dict(keep_vars=True).items()}
        # derived entries: b_ih + b_hh of every LSTM layer / direction, in a persistent buffer (stable pointers): with them every weight
        # image of an LSTM is pure data movement too, and is replayed like the rest
        ih = [k for k in self.sd if '.bias_ih_l' in k]
        if ih:
            if self._bias_sums is None or self._bias_sums[0].device != self.sd[ih[0]].device or len(self._bias_sums) != len(ih):
                self._bias_sums = [torch.empty_like(self.sd[k], dtype=torch.float32) for k in ih]
            for k, out in zip(ih, self._bias_sums):
                torch.add(self.sd[k].float(), self.sd[k.replace('.bias_ih_l', '.bias_hh_l')].float(), out=out)
                self.sd[k.replace('.bias_ih_l', '.bias_sum_l')] = out
        rp = self._replay
        sd_dev = str(next(iter(self.sd.values())).device)
        if rp is not None and (self._replay_dev != sd_dev or not rp.matches(self.sd)):
            rp, self._replay, self._builders = None, None, {}
        if rp is None and not first and self._builders and self.replay_enabled and not (
                self.sd and next(iter(self.sd.values())).is_cuda and torch.cuda.is_current_stream_capturing()):
            # the weights changed for the first time since the images were built (an optimizer step): from now on the images are
            # re-packed by one gather launch per arena instead of by their closures (aero_amd/repack.py)
            from .repack import WeightReplay
            rp = WeightReplay(self.lib, self.ops.stream)
            rp.compile(self.sd, self._builders, lambda d: setattr(self, 'sd', d))
            self._replay, self._replay_dev = rp, sd_dev
            if os.environ.get('AERO_REPACK_DEBUG'):
                print(f'[repack] {len(rp.objects)} image sets replayed by gather, {len(rp.skipped)} rebuilt by their closures:')
                for k, why in rp.skipped.items():
                    print(f'[repack]   {k}: {why}')
            self._cache = dict(rp.objects)
        elif rp is not None:
            self._cache = dict(rp.refresh(self.sd))
        else:
            self._cache = {}

    _replay, _replay_dev, replay_enabled = None, None, os.environ.get('AERO_REPACK_GATHER', '1') != '0'
    _bias_sums = None
    _stat_need = 0
    _bwd_need = 0

    def w(self, name):
        return self.sd[name].float()

    def spec(self, key, build):
        s = self._cache.get(key)
        if s is None:
            s = self._cache[key] = build()
            if key not in self._builders:
                self._builders[key] = build
        return s

    def _window(self, win, dev):
        key = ('win', win, str(dev))
        if key not in self._tables:
            self._tables[key] = _hann_padded(win, self.model.nfft, dev)
        return self._tables[key]

    def _inv_env(self, win, hop, T, dev):
        from .engine import HipEngine
        key = ('env', win, hop, T, str(dev))
        if key not in self._tables:
            if len(self._tables) > 32:
                self._tables.clear()
            fake = HipEngine.__new__(HipEngine)
            fake.model, fake._tables = self.model, {}
            self._tables[key] = HipEngine._inv_env(fake, win, hop, T, dev)
        return self._tables[key]

    # ------------------------------------------------------------------ small helpers over Ops
    def _norm(self, x, G, per_row, gamma, beta, act, **kw):
        """aero_norm_stats + aero_norm_apply; returns (y, stats)"""
        y = self.ops.norm_act(x, G, per_row, gamma, beta, act, **kw)
        return y, (self.ops._last_stats if kw.get('normalize', True) else None)

    def _bn_relu(self, y, bnmod, prefix, dst=None, dst_strides=None):
        """training-mode BatchNorm + ReLU (modules.py:287,293,300) and the running-statistics bookkeeping of nn.BatchNorm"""
        Bq, Fy, Ty, Cy = y.shape
        gamma, beta = self.w(prefix + '.weight'), self.w(prefix + '.bias')
        if Cy > gamma.numel():                                  # zero-padded channels (r_channel 5 -> 8): gamma = beta = 0 there
            def b_pad():                                         # (an image like the others: replayed after optimizer steps)
                gm, bt = self.w(prefix + '.weight'), self.w(prefix + '.bias')
                return torch.cat([gm, gm.new_zeros(Cy - gm.numel())]), torch.cat([bt, bt.new_zeros(Cy - bt.numel())])
            gamma, beta = self.spec(prefix + f'.pad{Cy}', b_pad)
        out = self.ops.norm_act(y, Cy, 2, gamma, beta, ACT_RELU, eps=bnmod.eps, dst=dst, dst_strides=dst_strides)
        st = self.ops._last_stats
        n = float(Bq * Fy * Ty)
        nc = bnmod.running_mean.numel()
        rm, rv, nbt = bnmod.running_mean, bnmod.running_var, bnmod.num_batches_tracked
        if rm.dtype == torch.float32 and rv.dtype == torch.float32 and rm.is_contiguous() and rv.is_contiguous() and nbt.dtype == torch.int64 \
                and bnmod.momentum is not None:
            # buffer bookkeeping of nn.BatchNorm (not the data path): one launch
            self.lib.call('aero_bn_running_update', st.data_ptr(), nc, C.c_double(n), C.c_float(bnmod.momentum), rm.data_ptr(), rv.data_ptr(),
                          nbt.data_ptr(), self.ops.stream(y))
        else:
            with torch.no_grad():
                mean = st[:nc, 0] / n
                var = (st[:nc, 1] / n - mean * mean).clamp_min(0)
                mom = bnmod.momentum if bnmod.momentum is not None else 1.0 / float(nbt + 1)
                rm.mul_(1 - mom).add_(mean.to(rm.dtype), alpha=mom)
                rv.mul_(1 - mom).add_((var * (n / max(n - 1.0, 1.0))).to(rv.dtype), alpha=mom)
                nbt.add_(1)
        return out, st, gamma, beta

    # ================================================================== forward
    def forward(self, mix):
        """-> (y fp32 [B,1,Lout], spec_out fp32 [B,F0,T,2], lr_spec complex64, ctx)"""
        m, ops = self.model, self.ops
        if m.in_channels != 1 or m.out_channels != 1:
            raise NotImplementedError('only in_channels = out_channels = 1 (all reference configs)')
        if not self.lib.is_emulator and not mix.is_cuda:
            raise RuntimeError('aero_amd trains on the MI355X only: move the model and the input to "cuda"')
        dev = mix.device
        self._sync_weights(dev)
        self._nfwd += 1
        B, _, L = mix.shape
        mix = mix.contiguous()
        # the GroupNorm / BatchNorm accumulators of this forward are slices of ONE zero-filled buffer (kept alive by the slices the
        # backward holds on to -- not the inference engine's per-stream arena, which the next forward zeroes again)
        ar = [torch.zeros(max(self._stat_need, 1 << 12), dtype=torch.float64, device=dev), 0, 0]
        ops._cur = ar
        ctx = _Ctx()
        hop, win = m.hop_length, m.win_length
        padn = (hop - L % hop) % hop
        stats = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        z = ops.stft(mix.reshape(B, L), L, L + padn, m.nfft, hop, self._window(win, dev), m.nfft // 2, stats=stats, sig_per_item=1, win_len=win)
        F0, T = z.shape[1], z.shape[2]
        x, mean_std = ops.spec_normalize(z, B, stats)
        ctx.mean, ctx.std = mean_std[:, 0].contiguous(), mean_std[:, 1].contiguous()
        ctx.B, ctx.T, ctx.F0, ctx.L = B, T, F0, L
        ctx.enc, ctx.dec = [], []
        saved = []
        Fq = F0
        for i, enc in enumerate(m.encoder):
            x, Fq, rec = self._enc_fwd(i, enc, x, B, Fq, T)
            ctx.enc.append(rec)
            saved.append((x, Fq))
        x = None
        for j, dec in enumerate(m.decoder):
            skip, Fs = saved.pop()
            x, rec = self._dec_fwd(j, dec, x, skip, B, Fs, T, ctx)
            ctx.dec.append(rec)
        spec_out = x
        hop_o, win_o = int(m.hop_length * m.scale), int(m.win_length * m.scale)
        Lout = min(hop_o * (T - 1), int(L * m.scale))
        self._stat_need, ops._cur = ar[2], None
        y = ops.istft(spec_out, m.nfft, hop_o, self._window(win_o, dev), self._inv_env(win_o, hop_o, T, dev), Lout)
        ctx.Lout, ctx.hop_o, ctx.win_o = Lout, hop_o, win_o
        return y.view(B, 1, Lout), spec_out, torch.view_as_complex(z).view(B, 1, F0, T), ctx

    # ------------------------------------------------------------------ encoder layer
    def _enc_fwd(self, i, enc, x, B, Fq, T):
        ops, dev = self.ops, x.device
        p = f'encoder.{i}'
        mk = pack.make_conv_spec
        r = _Ctx()
        r.Fq = Fq
        if enc.is_first:
            r.x_in = x

            def b_pre():
                w, df, dt = pack.conv2d_taps(self.w(f'{p}.pre_conv.weight'), 0, 0)
                return mk(w, self.w(f'{p}.pre_conv.bias'), w.shape[-1], 0, df, dt, dev)
            x = ops.conv(self.spec(p + '.pre', b_pre), x, None, B, Fq, Fq, T)
        r.ftb = None
        if enc.freq_attn:
            x, r.ftb = self._ftb_fwd(p + '.freq_attn_block', enc.freq_attn_block, x, B, Fq, T)
        r.x_conv = x
        Fo = (Fq + 2 * enc.pad - enc.kernel_size) // enc.stride + 1

        def b_conv():
            w, df, dt = pack.conv2d_taps(self.w(f'{p}.conv.weight'), enc.pad, 0)
            return mk(w, self.w(f'{p}.conv.bias'), w.shape[-1], 0, df, dt, dev, fstride=enc.stride)
        r.yc = ops.conv(self.spec(p + '.conv', b_conv), x, None, B, Fq, Fo, T)
        if enc.norm:
            x, r.stc = self._norm(r.yc, enc.norm_groups, 0, self.w(f'{p}.norm1.weight'), self.w(f'{p}.norm1.bias'), ACT_GELU)
        else:
            x, r.stc = self._norm(r.yc, 1, 0, None, None, ACT_GELU, normalize=False)
        r.dconv = []
        if enc.dconv is not None:
            for d_ in range(enc.dconv.depth):
                x, rec = self._dconv_layer_fwd(f'{p}.dconv.layers.{d_}', enc.dconv, d_, x, B, Fo, T)
                r.dconv.append(rec)
        r.x_rw = x
        if enc.rewrite is None:
            raise NotImplementedError('encoder layer without rewrite conv')

        def b_rw():
            w, df, dt = pack.conv2d_taps(self.w(f'{p}.rewrite.weight'), enc.context, enc.context)
            return mk(w, self.w(f'{p}.rewrite.bias'), w.shape[-1], 0, df, dt, dev)
        r.r = ops.conv(self.spec(p + '.rewrite', b_rw), x, None, B, Fo, Fo, T)
        emb = None
        if i == 0 and self.model.freq_emb is not None:
            fe = self.model.freq_emb
            e = (self.w('freq_emb.embedding.weight') * (fe.scale * self.model.freq_emb_scale)).half().contiguous()      # [Fo, C]
            emb = e[None, :, None, :].expand(B, Fo, T, e.shape[1])
        if enc.norm:
            x, r.st_rw = self._norm(r.r, enc.norm_groups, 0, self.w(f'{p}.norm2.weight'), self.w(f'{p}.norm2.bias'), ACT_GLU, res=emb)
        else:
            x, r.st_rw = self._norm(r.r, 1, 0, None, None, ACT_GLU, normalize=False, res=emb)
        r.Fo = Fo
        return x, Fo, r

    def _ftb_fwd(self, q, ftb, x, B, Fq, T):
        """FTB in training mode (modules.py:304-325): conv1 (r_channel 5, run at 8 channels with zero rows) -> BN -> ReLU ->
        [B,T,F*8] image -> Conv1d k=9 -> BN -> ReLU = gate; freq_fc(x * gate); conv2 on cat(att, x) -> BN -> ReLU."""
        ops, dev = self.ops, x.device
        mk = pack.make_conv_spec
        Fd, Cc, rch = ftb.input_dim, ftb.in_channel, ftb.r_channel
        assert Fd == Fq
        rp = _r8(rch)
        r = _Ctx()
        r.x, r.rp = x, rp

        def b_c1():
            w = self.w(f'{q}.conv1.0.weight')[:, :, 0, 0]                       # [r, C]
            wp = torch.zeros(rp, Cc, device=dev)
            wp[:rch] = w
            bp = torch.zeros(rp, device=dev)
            bp[:rch] = self.w(f'{q}.conv1.0.bias')
            return mk(wp[None, :, None, :], bp, Cc, 0, [0], [0], dev)
        r.y1 = ops.conv(self.spec(q + '.c1', b_c1), x, None, B, Fq, Fq, T)                                  # [B,F,T,rp]
        r.c1 = torch.empty(B, T, Fq * rp, dtype=torch.float16, device=dev)
        _, r.st1, r.g1, r.b1 = self._bn_relu(r.y1, ftb.conv1[1], f'{q}.conv1.1', dst=r.c1, dst_strides=(T * Fq * rp, rp, Fq * rp))

        def b_c1d():
            w = self.w(f'{q}.conv1d.0.weight')                                  # [C, r*F, 9], input channel c*F + f (modules.py:311)
            k9 = w.shape[-1]
            w = w.view(Cc, rch, Fd, k9).permute(0, 3, 2, 1)                     # [M, k, F, r]
            wp = torch.zeros(Cc, k9, Fd, rp, device=dev)
            wp[..., :rch] = w
            return mk(wp.reshape(1, Cc, k9, Fd * rp), self.w(f'{q}.conv1d.0.bias'), Fd * rp, 0, [0] * k9, [j - (k9 // 2) for j in range(k9)], dev)
        r.y2 = ops.conv(self.spec(q + '.c1d', b_c1d), r.c1.view(B, 1, T, Fq * rp), None, B, 1, 1, T)          # [B,1,T,C]
        gate, r.st2, r.g2, r.b2 = self._bn_relu(r.y2, ftb.conv1d[1], f'{q}.conv1d.1')
        r.gate = gate.view(B, T, Cc)

        def b_fc():
            wfc = self.w(f'{q}.freq_fc.weight')
            img = torch.zeros(pack._round_up(Fd, 128), pack._round_up(Fd, 32), device=dev)
            img[:Fd, :Fd] = wfc
            imt = torch.zeros_like(img)
            imt[:Fd, :Fd] = wfc.t()
            return img.half().contiguous(), imt.half().contiguous()
        r.fc = ops.freqfc(x, self.spec(q + '.fc', b_fc)[0], r.gate)

        def b_c2():
            w, df, dt = pack.conv2d_taps(self.w(f'{q}.conv2.0.weight'), 0, 0)
            return mk(w, self.w(f'{q}.conv2.0.bias'), Cc, Cc, df, dt, dev)
        r.y3 = ops.conv(self.spec(q + '.c2', b_c2), r.fc, x, B, Fq, Fq, T)
        out, r.st3, r.g3, r.b3 = self._bn_relu(r.y3, ftb.conv2[1], f'{q}.conv2.1')
        return out, r

    # ------------------------------------------------------------------ DConv layer
    def _dconv_layer_fwd(self, q, dc, d_, x, B, Fo, T):
        ops, dev = self.ops, x.device
        mk = pack.make_conv_spec
        Cc, hid = dc.channels, dc.hidden
        hp = _r8(hid)
        dil = 2 ** d_ if dc.dilate else 1
        k = dc.kernel
        r = _Ctx()
        r.x, r.hp, r.dil = x, hp, dil
        if not dc.norm:
            raise NotImplementedError('DConv without GroupNorm is not used by any reference config')

        def b_c1():
            w = self.w(f'{q}.conv1.0.weight')                                   # [hid, C, k]
            wp = torch.zeros(hp, Cc, k, device=dev)
            wp[:hid] = w
            bp = torch.zeros(hp, device=dev)
            bp[:hid] = self.w(f'{q}.conv1.0.bias')
            t, df, dt = pack.conv1d_taps(wp, dil, dil * (k // 2))
            return mk(t, bp, Cc, 0, df, dt, dev), wp
        r.h1 = ops.conv(self.spec(q + '.c1', b_c1)[0], x, None, B, Fo, Fo, T)

        def padded(name):
            v = self.w(name)
            out = torch.zeros(hp, device=dev)
            out[:hid] = v
            return out
        r.g1, r.be1 = self.spec(q + '.gn1pad', lambda: (padded(f'{q}.conv1.1.weight'), padded(f'{q}.conv1.1.bias')))   # (replayed images)
        act = {'snake': ACT_SNAKE, 'gelu': ACT_GELU}.get(dc.act_func, ACT_RELU)
        r.act = act
        r.snake_a = self.w(f'{q}.act.a').reshape(-1).contiguous() if act == ACT_SNAKE else None
        a, r.st1 = self._norm(r.h1, 1, 1, r.g1, r.be1, act, snake_a=r.snake_a, stat_count=T * hid)
        r.lstm = r.attn = None
        if dc.lstm:
            a, r.lstm = self._blstm_fwd(q + '.lstm', hid, hp, a, B, Fo, T)
        if dc.time_attn:
            a, r.attn = self._attn_fwd(q + '.time_attn', dc.layers[d_]['time_attn'], hid, hp, a, B, Fo, T)
        r.a = a

        def b_c2():
            w = self.w(f'{q}.conv2.0.weight')                                   # [2C, hid, 1]
            wp = torch.zeros(2 * Cc, hp, 1, device=dev)
            wp[:, :hid] = w
            t, df, dt = pack.conv1d_taps(wp, 1, 0)
            return mk(t, self.w(f'{q}.conv2.0.bias'), hp, 0, df, dt, dev), wp
        r.h2 = ops.conv(self.spec(q + '.c2', b_c2)[0], a, None, B, Fo, Fo, T)
        y, r.st2 = self._norm(r.h2, 1, 1, self.w(f'{q}.conv2.1.weight'), self.w(f'{q}.conv2.1.bias'), ACT_GLU,
                              layer_scale=self.w(f'{q}.conv2.3.scale'), res=x)

        # LayerScale (init 1e-3, modules.py:130-141) shrinks every gradient inside the residual branch by its magnitude: a power of
        # two of that size is put back where the gradient enters the branch and taken out where it leaves, so the branch's fp16
        # gradients sit in the same range as the main path's.  The factor only positions the fp16 window (any power of two is
        # exact), so it is read from the device once and refreshed every 64 forwards -- never while a HIP graph is being captured.
        ent = self._boosts.get(q)
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
        if ent is None or (not capturing and self._nfwd - ent[1] >= 64):
            if capturing:
                raise RuntimeError('aero_amd: run at least one eager training step before capturing it in a HIP graph')
            mean = float(self.w(f'{q}.conv2.3.scale').abs().mean())
            ent = (float(2.0 ** min(12, max(0, round(-math.log2(max(mean, 2.0 ** -12)))))), self._nfwd)
            self._boosts[q] = ent
        r.boost = ent[0]
        return y, r

    def _lstm_specs(self, q, H, dev):
        def b():
            sd = {k[len(q) + 1:]: v for k, v in self.sd.items() if k.startswith(q + '.')}
            sdf = {k: v.float() for k, v in sd.items()}
            layers = [pack.pack_lstm_layer(self.lib, sdf, 'lstm', l, H, dev) for l in range(2)]
            whh_t = [TO.pack_whh_t(self.lib, [sdf[f'lstm.weight_hh_l{l}'], sdf[f'lstm.weight_hh_l{l}_reverse']], H, dev) for l in range(2)]
            w = sdf['linear.weight']
            lin = pack.make_conv_spec(w[None, :, None, :], sdf['linear.bias'], w.shape[1], 0, [0], [0], dev)
            return layers, whh_t, lin, {k[5:]: v for k, v in sdf.items() if k.startswith('lstm.')}
        return self.spec(q + '.specs', b)

    def _blstm_fwd(self, q, H, hp, a, B, Fo, T):
        """BLSTM (modules.py:32-65): frames by aero_frames_op, two bidirectional layers with their gates / cell states kept, the stitch,
        Linear(2H -> H) + skip.  `a` [B,Fo,T,hp] with H live channels (hp = H unless H % 8)."""
        if hp != H:
            raise NotImplementedError('BLSTM training path needs a hidden size that is a multiple of 8')
        ops, dev = self.ops, a.device
        layers, whh_t, lin, _ = self._lstm_specs(q, H, dev)
        R = B * Fo
        r = _Ctx()
        framed = T > 200
        if framed:
            W, S = 200, 100
            nf = blstm_frames(T, W, S)                       # (engine.blstm_frames: a last frame that the stitch discards whole is not computed)
            fr0 = TO.frames_op(ops, a.view(R, T, H), 0, R, T, H, nf, W, S)
        else:
            W, S, nf = T, 1, 1
            fr0 = a.view(R, T, H)
        nseq = R * nf
        r.framed, r.W, r.S, r.nf, r.nseq, r.a = framed, W, S, nf, nseq, a
        outs, saves, xin = [], [], fr0
        for l in range(2):
            pj, xb, whh, fz = layers[l]
            out = torch.empty(nseq, W, 2 * H, dtype=torch.float16, device=dev)
            save = TO.lstm_save_buffers(nseq, W, H, dev)
            in_ch = xin.shape[-1]
            if fz is not None:
                ops.lstm(None, None, whh, H, nseq, W, 0, 0, 1, 1, W, out, x=xin.reshape(nseq, W, in_ch), fused=fz, save=save)
            else:
                xp = ops.conv(pj, xin.reshape(1, 1, nseq * W, in_ch), None, 1, 1, 1, nseq * W)
                ops.lstm(xp, xb, whh, H, nseq, W, 0, 0, 1, 1, W, out, save=save)
            outs.append(out)
            saves.append(save)
            xin = out
        r.fr0, r.outs, r.saves = fr0, outs, saves
        r.out1s = TO.frames_op(ops, outs[1], 2, R, T, 2 * H, nf, W, S) if framed else outs[1]
        y = ops.conv(lin, r.out1s.view(B, Fo, T, 2 * H), None, B, Fo, Fo, T, res=a)
        return y, r

    def _attn_fwd(self, q, mod, Cc, hp, a, B, Fo, T):
        if hp != Cc:
            raise NotImplementedError('LocalState training path needs a channel count that is a multiple of 8')
        ops, dev = self.ops, a.device

        def b():
            names = ('query', 'key', 'content', 'query_decay')
            w = torch.cat([self.w(f'{q}.{n}.weight')[:, :, 0] for n in names], 0)
            bb = torch.cat([self.w(f'{q}.{n}.bias') for n in names], 0)
            qk = pack.make_conv_spec(w[None, :, None, :], bb, w.shape[1], 0, [0], [0], dev)
            wp = self.w(f'{q}.proj.weight')[:, :, 0]
            pj = pack.make_conv_spec(wp[None, :, None, :], self.w(f'{q}.proj.bias'), wp.shape[1], 0, [0], [0], dev)
            return qk, pj, w, wp
        qk, pj, _, _ = self.spec(q + '.specs', b)
        r = _Ctx()
        r.a = a
        r.heads, r.ndecay = mod.heads, mod.ndecay
        r.qkvd = ops.conv(qk, a, None, B, Fo, Fo, T)
        r.att = ops.localstate(r.qkvd, B * Fo, T, Cc, mod.heads, mod.ndecay)
        y = ops.conv(pj, r.att.view(B, Fo, T, Cc), None, B, Fo, Fo, T, res=a)
        return y, r

    # ------------------------------------------------------------------ decoder layer
    def _dec_fwd(self, j, dec, x, skip, B, Fq, T, ctx):
        ops, dev = self.ops, skip.device
        p = f'decoder.{j}'
        mk = pack.make_conv_spec
        half = dec.chin // 2
        r = _Ctx()
        r.x, r.skip, r.Fq = x, skip, Fq
        if dec.rewrite is None or dec.dconv is not None:
            raise NotImplementedError('decoder without rewrite conv / with DConv is not used by any reference config')

        def b_rw():
            w, df, dt = pack.conv2d_taps(self.w(f'{p}.rewrite.weight'), dec.context, dec.context)
            return mk(w, self.w(f'{p}.rewrite.bias'), half, half, df, dt, dev)
        r.r = ops.conv(self.spec(p + '.rewrite', b_rw), x, skip, B, Fq, Fq, T)
        if dec.norm:
            y, r.st1 = self._norm(r.r, dec.norm_groups, 0, self.w(f'{p}.norm1.weight'), self.w(f'{p}.norm1.bias'), ACT_GLU)
        else:
            y, r.st1 = self._norm(r.r, 1, 0, None, None, ACT_GLU, normalize=False)
        r.y = y
        Fu = (Fq - 1) * dec.stride + dec.kernel_size
        Ft = Fu - 2 * dec.pad
        r.Fu, r.Ft = Fu, Ft

        def b_tr():
            w, df, dt = pack.convtr_taps(self.w(f'{p}.conv_tr.weight'), dec.stride)
            return mk(w, self.w(f'{p}.conv_tr.bias'), w.shape[-1], 0, df, dt, dev, transposed=1, fstride=dec.stride)
        tr = self.spec(p + '.conv_tr', b_tr)
        if dec.last:
            if dec.norm:
                raise NotImplementedError('GroupNorm on the last decoder layer (norm_starts = 0)')
            out = ops.conv(tr, y, None, B, Fq, Fu, T, dst_f32=True, dst_f_off=dec.pad, dst_F=Ft, batch_scale=ctx.std, batch_shift=ctx.mean)
            return out, r
        r.z = ops.conv(tr, y, None, B, Fq, Fu, T)                                                  # untrimmed rows: norm2 sees them (aero.py:206)
        if dec.norm:
            out, r.st2 = self._norm(r.z, dec.norm_groups, 0, self.w(f'{p}.norm2.weight'), self.w(f'{p}.norm2.bias'), ACT_GELU, f_lo=dec.pad, f_cnt=Ft)
        else:
            out, r.st2 = self._norm(r.z, 1, 0, None, None, ACT_GELU, normalize=False, f_lo=dec.pad, f_cnt=Ft)
        return out, r

    # ================================================================== backward
    def backward(self, ctx, dy, grads, stage_done=None):
        """dy fp32 [B,1,Lout]: gradient of the waveform.  `grads`: {state-dict name: fp32 view of the flat gradient buffer} -- every
        parameter gradient is WRITTEN there, still multiplied by its stage's scale; stage_done(prefix, scale) is called when the
        gradients of the parameters `prefix*` are final, with that stage's device pair {S, 1/S}."""
        m, ops = self.model, self.ops
        dev = dy.device
        B, T, F0 = ctx.B, ctx.T, ctx.F0
        self.g = grads
        self._side = self._param_stream(dev)
        # one zero-filled scratch buffer for every small accumulator of this pass (Ops.zeros32), sized from the previous pass's demand
        ar = [torch.zeros(max(self._bwd_need, 1 << 14), dtype=torch.float32, device=dev), 0, 0]
        ops._bwd = ar if os.environ.get('AERO_BWD_ARENA', '1') != '0' else None
        try:
            self._backward(ctx, dy, stage_done, m, ops, dev, B, T, F0)
        finally:
            if self._side is not None:
                torch.cuda.current_stream(dev).wait_stream(self._side)
            self._side = None
            self._bwd_need, ops._bwd = max(self._bwd_need, ar[2]), None

    def _backward(self, ctx, dy, stage_done, m, ops, dev, B, T, F0):
        dz = bw.istft_bwd(ops, dy.reshape(B, ctx.Lout).contiguous().float(), m.nfft, ctx.hop_o, self._window(ctx.win_o, dev),
                          self._inv_env(ctx.win_o, ctx.hop_o, T, dev), T)                               # fp32 [B,F0,T,2]
        # adjoint of x * std + mean (aero.py:497-498) and the fp32 -> fp16 boundary with the loss scale
        d16, sc = TO.scale_cast(ops, dz, ctx.std, GRAD_TARGET)
        # Every gradient tensor that crosses a stage boundary carries its own power-of-two scale {S, 1/S} (device memory): the
        # magnitudes change by orders of magnitude from level to level (LayerScale, GroupNorm, the 0.1-rescaled init), in either
        # direction, so each stage's input is re-normalised to GRAD_TARGET from its own maximum (aero_rescale_f16) and the stage's
        # parameter gradients are un-scaled with that stage's factor (stage_done).
        dx = d16
        dskips = []
        for j in reversed(range(len(m.decoder))):
            dxp, dskip = self._dec_bwd(j, m.decoder[j], ctx.dec[j], dx, B, T)
            dskips.append((dskip, sc))                           # decoder j used the output of encoder len-1-j
            if stage_done is not None:
                stage_done(f'decoder.{j}.', sc)
            if dxp is not None:
                dx, sc = TO.rescale_f16(ops, dxp, sc, target=GRAD_TARGET)
        dskips.reverse()                                         # dskips[j] <-> encoder len-1-j
        dx = None
        n = len(m.encoder)
        for i in reversed(range(n)):
            dsk, ssk = dskips[n - 1 - i]
            if dx is None:
                dout, sc = TO.rescale_f16(ops, dsk, ssk, target=GRAD_TARGET)
            else:
                dout, sc = TO.rescale_f16(ops, dsk, ssk, dx.contiguous(), sc, target=GRAD_TARGET)
            dx = self._enc_bwd(i, m.encoder[i], ctx.enc[i], dout, B, T)
            if stage_done is not None:
                stage_done(f'encoder.{i}.', sc)
        return None

    _unboost = 1.0
    _range_of = None             # AeroFunction: prefix -> contiguous fp32 view of the flat gradient buffer covering those parameters
    _side = None                 # the parameter-gradient stream of the backward pass that is running (None: everything on one stream)
    _side_streams = None

    def _param_stream(self, dev):
        if dev.type != 'cuda' or os.environ.get('AERO_TRAIN_STREAMS', '2') == '1':
            return None
        if self._side_streams is None:
            self._side_streams = {}
        if dev not in self._side_streams:
            from .engine import side_streams
            self._side_streams[dev] = side_streams(dev, 1)[0]   # (the package's shared side streams: engine.side_streams on hardware queues)
        return self._side_streams[dev]

    @contextlib.contextmanager
    def on_param_stream(self, *tensors):
        """Parameter-gradient work (weight gradients, their un-scaling, the gradient all-reduce) goes to a SECOND HIP stream: nothing on the
        activation-gradient chain waits for it, and at BASELINE config 5's per-GPU batch that chain is a string of latency-bound
        launches (LSTM / LocalState backward: 9-18 blocks on 256 CUs) under which the weight-gradient GEMMs fit.  Entering waits for
        everything the main stream has issued so far; `tensors` (allocated on the main stream, read here) are kept from being
        recycled until this stream is done with them; backward() joins the two streams before it returns."""
        side = self._side
        if side is None:
            yield
            return
        side.wait_stream(torch.cuda.current_stream(side.device))
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(side)
        with torch.cuda.stream(side):
            yield

    def _factor(self, f, dev):
        key = ('factor', f, str(dev))
        if key not in self._tables:
            self._tables[key] = torch.full((1,), f, dtype=torch.float32, device=dev)
        return self._tables[key]

    def _put(self, name, t, unboost=True):
        """write one parameter gradient.  Inside a boosted DConv branch (_dconv_layer_fwd) the gradients carry the branch's boost: with the
        flat buffer at hand the whole layer's range is un-boosted in ONE pass at the end of the layer (_dconv_layer_bwd) -- a gradient
        that must not be (LayerScale's own) is pre-multiplied here so that the range pass restores it exactly (powers of two)"""
        dst = self.g[name]
        dst.copy_(t.reshape(dst.shape))
        if self._unboost == 1.0:
            return
        if self._range_of is not None:
            if not unboost:
                TO.scale_f32(self.ops, dst, self._factor(1.0 / self._unboost, dst.device))
        elif unboost:
            TO.scale_f32(self.ops, dst, self._factor(self._unboost, dst.device))

    def _wgrad_to(self, wname, bname, dy, x, df, dt, fstride=1, coff=0):
        """weight (and bias) gradient of a convolution whose weight parameter `wname` is [M, Ctot, taps...] with its taps in the order
        of df / dt: ADDED straight into the flat gradient buffer (zero at the start of the backward), in the parameter's own layout at
        column offset `coff`, when the operands' 8-aligned channel counts are the parameter's; through a temporary otherwise (padded
        channels: the FTB's 5, DConv hidden widths that are not multiples of 8, the 2-channel ends of the net)."""
        gw = self.g[wname]
        gb = self.g[bname] if bname else None
        M, Cx, nt, rowlen = dy.shape[-1], x.shape[-1], len(df), gw.shape[1]
        with self.on_param_stream(dy, x):
            if bw.wgrad_direct_ok(dy, x) and gw.shape[0] == M and gw[0, 0].numel() == nt and coff + Cx <= rowlen:
                bw.conv_wgrad(self.ops, dy, x, df, dt, fstride=fstride, bias=gb is not None, dw_out=gw, db_out=gb, layout=1, rowlen=rowlen, coff=coff)
            else:
                dw, db = bw.conv_wgrad(self.ops, dy, x, df, dt, fstride=fstride, bias=gb is not None)
                Mp, Cp = gw.shape[0], min(Cx, rowlen - coff)
                gw.view(Mp, rowlen, nt)[:, coff:coff + Cp].copy_(dw[:, :Mp, :Cp].permute(1, 2, 0))
                if gb is not None:
                    gb.copy_(db[:Mp])
            if self._unboost != 1.0 and self._range_of is None:
                for t in (gw, gb):
                    if t is not None:
                        TO.scale_f32(self.ops, t, self._factor(self._unboost, t.device))

    def _norm_bwd_to(self, names, x, dy, stats, G, per_row, gamma, beta, act, **kw):
        """bw.norm_bwd with the parameter gradients accumulated straight into the flat gradient buffer: names = the parameters of
        (gamma, beta[, LayerScale[, Snake a]]) -- None to skip one; a parameter narrower than the kernel's (zero-padded) channel count
        goes through a temporary.  Returns dx."""
        keys = ('dgamma', 'dbeta', 'dls', 'dsn')
        want = (x.shape[-1], x.shape[-1], x.shape[-1] // 2, x.shape[1])
        out, late = {}, []
        for k, nme, n in zip(keys, names, want):
            if nme is None:
                continue
            if self.g[nme].numel() == n:
                out[k] = self.g[nme].view(-1)
            else:
                late.append((k, nme))
        res = bw.norm_bwd(self.ops, x, dy, stats, G, per_row, gamma, beta, act, out=out, **kw)
        by_key = dict(zip(('dgamma', 'dbeta', 'dls', 'dsn'), res[1:] + (None,) * 4))
        for k, nme in late:
            self._put(nme, by_key[k][:self.g[nme].numel()], unboost=(k != 'dls'))
        if self._unboost != 1.0:
            for k, t in out.items():
                if self._range_of is not None:
                    if k == 'dls':                               # (see _put: pre-multiplied so that the layer's range pass restores it)
                        TO.scale_f32(self.ops, t, self._factor(1.0 / self._unboost, t.device))
                elif k != 'dls':
                    TO.scale_f32(self.ops, t, self._factor(self._unboost, t.device))
        return res[0]

    # ------------------------------------------------------------------ decoder layer
    def _dec_bwd(self, j, dec, r, dout, B, T):
        ops, dev = self.ops, dout.device
        p = f'decoder.{j}'
        s, K, pad = dec.stride, dec.kernel_size, dec.pad
        w_tr = self.w(f'{p}.conv_tr.weight')
        Fq = r.Fq
        if dec.last:
            dz, pad_eff, Fz = dout.contiguous(), pad, r.Ft
        else:
            M = r.z.shape[-1]
            if dec.norm:
                dfull = torch.zeros(B, r.Fu, T, M, dtype=torch.float16, device=dev)      # dL/d(GELU output) is zero on the trimmed rows
                dfull[:, pad:pad + r.Ft].copy_(dout)
                dz = self._norm_bwd_to((f'{p}.norm2.weight', f'{p}.norm2.bias'), r.z, dfull, r.st2, dec.norm_groups, 0,
                                       self.w(f'{p}.norm2.weight'), self.w(f'{p}.norm2.bias'), ACT_GELU)
                pad_eff, Fz = 0, r.Fu
            else:
                zt = r.z[:, pad:pad + r.Ft]
                dz = bw.norm_bwd(ops, zt, dout.contiguous(), None, 1, 0, None, None, ACT_GELU)[0]
                pad_eff, Fz = pad, r.Ft
        dyv = ops.conv(self.spec(p + f'.tr_dgrad{pad_eff}', lambda: bw.dgrad_convtr(self.w(f'{p}.conv_tr.weight'), s, pad_eff, dev)), dz, None, B, Fz, Fq, T)
        self._wgrad_to(f'{p}.conv_tr.weight', None, r.y, dz, [kk - pad_eff for kk in range(K)], [0] * K, fstride=s)     # [Cin, Cout, K, 1]
        with self.on_param_stream(dz):
            if dz.shape[-1] % 8 == 0:                           # (the bias sum rides on a product with 8 of dz's own channels)
                bw.conv_wgrad(ops, dz, dz[..., :8], [0], [0], bias=True, db_out=self.g[f'{p}.conv_tr.bias'])
            else:
                _, db = bw.conv_wgrad(ops, dz, dz, [0], [0], bias=True)
                self._put(f'{p}.conv_tr.bias', db)
        # norm1 + GLU
        if dec.norm:
            dr = self._norm_bwd_to((f'{p}.norm1.weight', f'{p}.norm1.bias'), r.r, dyv, r.st1, dec.norm_groups, 0,
                                   self.w(f'{p}.norm1.weight'), self.w(f'{p}.norm1.bias'), ACT_GLU)
        else:
            dr = bw.norm_bwd(ops, r.r, dyv, None, 1, 0, None, None, ACT_GLU)[0]
        # rewrite 3x3 over cat(x, skip) (aero.py:195-198)
        w_rw = self.w(f'{p}.rewrite.weight')
        half = dec.chin // 2
        ctxp = dec.context
        spec_rw = self.spec(p + '.rewrite', None)
        # [2chin, chin, kF, kT]: columns [half:] see the skip, [:half] the decoder path (the first decoder's x is zeros, aero.py:484:
        # those columns keep the zero they start with)
        self._wgrad_to(f'{p}.rewrite.weight', f'{p}.rewrite.bias', dr, r.skip, spec_rw.df, spec_rw.dt, coff=half)
        if r.x is not None:
            self._wgrad_to(f'{p}.rewrite.weight', None, dr, r.x, spec_rw.df, spec_rw.dt, coff=0)
        dskip = ops.conv(self.spec(p + '.rw_dgrad_s', lambda: bw.dgrad_conv2d(self.w(f'{p}.rewrite.weight')[:, half:], ctxp, ctxp, dev)), dr, None, B, Fq, Fq, T)
        dxp = None
        if r.x is not None:
            dxp = ops.conv(self.spec(p + '.rw_dgrad_x', lambda: bw.dgrad_conv2d(self.w(f'{p}.rewrite.weight')[:, :half], ctxp, ctxp, dev)), dr, None, B, Fq, Fq, T)
        return dxp, dskip

    # ------------------------------------------------------------------ encoder layer
    def _enc_bwd(self, i, enc, r, dout, B, T):
        ops, dev, m = self.ops, dout.device, self.model
        p = f'encoder.{i}'
        Fq, Fo = r.Fq, r.Fo
        if i == 0 and m.freq_emb is not None:
            ge = self.g['freq_emb.embedding.weight']
            dout = dout.contiguous()
            with self.on_param_stream(dout):
                ge.zero_()
                TO.sum_bt(ops, dout, ge, float(m.freq_emb.scale * m.freq_emb_scale))
        # norm2 + GLU, rewrite
        if enc.norm:
            dr = self._norm_bwd_to((f'{p}.norm2.weight', f'{p}.norm2.bias'), r.r, dout, r.st_rw, enc.norm_groups, 0,
                                   self.w(f'{p}.norm2.weight'), self.w(f'{p}.norm2.bias'), ACT_GLU)
        else:
            dr = bw.norm_bwd(ops, r.r, dout, None, 1, 0, None, None, ACT_GLU)[0]
        w_rw = self.w(f'{p}.rewrite.weight')
        c = enc.context
        spec_rw = self.spec(p + '.rewrite', None)
        self._wgrad_to(f'{p}.rewrite.weight', f'{p}.rewrite.bias', dr, r.x_rw, spec_rw.df, spec_rw.dt)
        dx = ops.conv(self.spec(p + '.rw_dgrad', lambda: bw.dgrad_conv2d(self.w(f'{p}.rewrite.weight'), c, c, dev)), dr, None, B, Fo, Fo, T)
        # DConv residual branch
        if enc.dconv is not None:
            for d_ in reversed(range(enc.dconv.depth)):
                dx = self._dconv_layer_bwd(f'{p}.dconv.layers.{d_}', enc.dconv, d_, r.dconv[d_], dx, B, Fo, T)
        # norm1 + GELU, the strided frequency conv
        if enc.norm:
            dyc = self._norm_bwd_to((f'{p}.norm1.weight', f'{p}.norm1.bias'), r.yc, dx, r.stc, enc.norm_groups, 0,
                                    self.w(f'{p}.norm1.weight'), self.w(f'{p}.norm1.bias'), ACT_GELU)
        else:
            dyc = bw.norm_bwd(ops, r.yc, dx, None, 1, 0, None, None, ACT_GELU)[0]
        w_c = self.w(f'{p}.conv.weight')
        spec_c = self.spec(p + '.conv', None)
        self._wgrad_to(f'{p}.conv.weight', f'{p}.conv.bias', dyc, r.x_conv, spec_c.df, spec_c.dt, fstride=enc.stride)
        need_dx = enc.freq_attn or enc.is_first or i > 0
        if not need_dx:
            return None
        K, s = enc.kernel_size, enc.stride
        dxc = ops.conv(self.spec(p + '.conv_dgrad', lambda: bw.dgrad_conv_fstride(self.w(f'{p}.conv.weight'), s, dev)), dyc, None, B, Fo, (Fo - 1) * s + K, T,
                       dst_f_off=enc.pad, dst_F=Fq)
        if enc.freq_attn:
            dxc = self._ftb_bwd(p + '.freq_attn_block', enc.freq_attn_block, r.ftb, dxc, B, Fq, T)
        if enc.is_first:
            self._wgrad_to(f'{p}.pre_conv.weight', f'{p}.pre_conv.bias', dxc, r.x_in, [0], [0])
            return None
        return dxc

    def _ftb_bwd(self, q, ftb, r, dout, B, Fq, T):
        ops, dev = self.ops, dout.device
        Cc, rch, rp = ftb.in_channel, ftb.r_channel, r.rp
        x = r.x
        # conv2 -> BN -> ReLU
        dy3 = self._norm_bwd_to((f'{q}.conv2.1.weight', f'{q}.conv2.1.bias'), r.y3, dout, r.st3, Cc, 2, r.g3, r.b3, ACT_RELU, eps=ftb.conv2[1].eps)
        w2 = self.w(f'{q}.conv2.0.weight')                                              # [C, 2C, 1, 1]: [att | inputs]
        self._wgrad_to(f'{q}.conv2.0.weight', f'{q}.conv2.0.bias', dy3, r.fc, [0], [0], coff=0)
        self._wgrad_to(f'{q}.conv2.0.weight', None, dy3, x, [0], [0], coff=Cc)
        dfc = ops.conv(self.spec(q + '.c2_dgrad_a', lambda: bw.dgrad_conv2d(self.w(f'{q}.conv2.0.weight')[:, :Cc], 0, 0, dev)), dy3, None, B, Fq, Fq, T)
        dxa = ops.conv(self.spec(q + '.c2_dgrad_b', lambda: bw.dgrad_conv2d(self.w(f'{q}.conv2.0.weight')[:, Cc:], 0, 0, dev)), dy3, None, B, Fq, Fq, T)
        # freq_fc and the gate product
        gfc = self.g[f'{q}.freq_fc.weight']
        with self.on_param_stream(dfc, x, r.gate):
            gfc.copy_(TO.freqfc_wgrad(ops, dfc, x, r.gate))
        okey = ('ones', B, T, Cc, str(dev))
        if okey not in self._tables:
            self._tables[okey] = torch.ones(B, T, Cc, dtype=torch.float16, device=dev)
        v = ops.freqfc(dfc, self.spec(q + '.fc', None)[1], self._tables[okey])
        dxb, dgate = TO.ftb_gate_bwd(ops, v, x, r.gate, add=dxa)
        # conv1d -> BN -> ReLU (the gate)
        dy2 = self._norm_bwd_to((f'{q}.conv1d.1.weight', f'{q}.conv1d.1.bias'), r.y2, dgate.view(B, 1, T, Cc), r.st2, Cc, 2, r.g2, r.b2, ACT_RELU,
                                eps=ftb.conv1d[1].eps)
        spec1d = self.spec(q + '.c1d', None)
        k9 = len(spec1d.dt)
        img = r.c1.view(B, 1, T, Fq * rp)
        with self.on_param_stream(dy2, img):
            dw, db = bw.conv_wgrad(ops, dy2, img, spec1d.df, spec1d.dt)                 # [k9, C, F*rp], ours channel f*rp + c
            dw = dw.view(k9, Cc, Fq, rp)[..., :rch].permute(1, 3, 2, 0)                 # -> [C, r, F, k]: reference channel c*F + f
            self._put(f'{q}.conv1d.0.weight', dw)
            self._put(f'{q}.conv1d.0.bias', db)

        def b_dg1d():
            w = self.w(f'{q}.conv1d.0.weight')
            wv = w.view(Cc, rch, Fq, k9).permute(0, 2, 1, 3)                            # [C, F, r, k]
            wp = torch.zeros(Cc, Fq, rp, k9, device=dev)
            wp[:, :, :rch] = wv
            return bw.dgrad_conv1d(wp.reshape(Cc, Fq * rp, k9), 1, k9 // 2, dev)
        dc1 = ops.conv(self.spec(q + '.c1d_dgrad', b_dg1d), dy2, None, B, 1, 1, T)      # [B,1,T,F*rp]
        # conv1 -> BN -> ReLU (dy arrives in the [B,T,F*rp] image layout)
        dimg = dc1.view(B, T, Fq, rp).permute(0, 2, 1, 3)                               # [B,F,T,rp] strided view
        dy1 = self._norm_bwd_to((f'{q}.conv1.1.weight', f'{q}.conv1.1.bias'), r.y1, dimg, r.st1, rp, 2, r.g1, r.b1, ACT_RELU, eps=ftb.conv1[1].eps)
        self._wgrad_to(f'{q}.conv1.0.weight', f'{q}.conv1.0.bias', dy1, x, [0], [0])

        def b_dg1():
            w = self.w(f'{q}.conv1.0.weight')
            wp = torch.zeros(rp, Cc, 1, 1, device=dev)
            wp[:rch] = w
            return bw.dgrad_conv2d(wp, 0, 0, dev)
        return ops.conv(self.spec(q + '.c1_dgrad', b_dg1), dy1, None, B, Fq, Fq, T, res=dxb)

    def _dconv_layer_bwd(self, q, dc, d_, r, dy, B, Fo, T):
        ops, dev = self.ops, dy.device
        Cc, hid, hp = dc.channels, dc.hidden, r.hp
        k = dc.kernel
        dy = dy.contiguous()
        bst = r.boost
        self._unboost = 1.0 / bst                               # applied by _put to every parameter gradient inside the branch
        try:
            # dh2 = boost * dL/dh2; LayerScale's own gradient (sum dy * GLU(.)) does not pass through the scale
            dh2 = self._norm_bwd_to((f'{q}.conv2.1.weight', f'{q}.conv2.1.bias', f'{q}.conv2.3.scale'), r.h2, dy, r.st2, 1, 1,
                                    self.w(f'{q}.conv2.1.weight'), self.w(f'{q}.conv2.1.bias'), ACT_GLU, layer_scale=self.w(f'{q}.conv2.3.scale') * bst)
            self._wgrad_to(f'{q}.conv2.0.weight', f'{q}.conv2.0.bias', dh2, r.a, [0], [0])
            def b_dg2():                                        # (every image closure reads the parameters through self.w: repack.py)
                wp = torch.zeros(2 * Cc, hp, 1, device=dev)
                wp[:, :hid] = self.w(f'{q}.conv2.0.weight')
                return bw.dgrad_conv1d(wp, 1, 0, dev)
            da = ops.conv(self.spec(q + '.c2_dgrad', b_dg2), dh2, None, B, Fo, Fo, T)
            if r.attn is not None:
                da = self._attn_bwd(q + '.time_attn', hid, r.attn, da, B, Fo, T)
            if r.lstm is not None:
                da = self._blstm_bwd(q + '.lstm', hid, r.lstm, da, B, Fo, T)
            dh1 = self._norm_bwd_to((f'{q}.conv1.1.weight', f'{q}.conv1.1.bias', None, f'{q}.act.a' if r.act == ACT_SNAKE else None),
                                    r.h1, da, r.st1, 1, 1, r.g1, r.be1, r.act, stat_count=T * hid, snake_a=r.snake_a)
            spec1, _ = self.spec(q + '.c1', None)
            self._wgrad_to(f'{q}.conv1.0.weight', f'{q}.conv1.0.bias', dh1, r.x, spec1.df, spec1.dt)
            dil = r.dil

            def b_dg1():
                wp = torch.zeros(hp, Cc, k, device=dev)
                wp[:hid] = self.w(f'{q}.conv1.0.weight')
                return bw.dgrad_conv1d(wp, dil, dil * (k // 2), dev)
            dxb = ops.conv(self.spec(q + '.c1_dgrad', b_dg1), dh1, None, B, Fo, Fo, T)
            if self._range_of is not None and bst != 1.0:
                with self.on_param_stream():                    # (after every gradient of the layer, on either stream)
                    TO.scale_f32(ops, self._range_of(q + '.'), self._factor(1.0 / bst, dev))
        finally:
            self._unboost = 1.0
        return TO.add_f16(ops, dy, dxb, scale_b=1.0 / bst)      # skip path + the branch, un-boosted in fp32

    def _attn_bwd(self, q, Cc, r, dy, B, Fo, T):
        ops, dev = self.ops, dy.device
        qk, pj, wq, wp = self.spec(q + '.specs', None)
        R = B * Fo
        self._wgrad_to(f'{q}.proj.weight', f'{q}.proj.bias', dy, r.att.view(B, Fo, T, Cc), [0], [0])
        datt = ops.conv(self.spec(q + '.proj_dgrad', lambda: bw.dgrad_conv1d(self.w(f'{q}.proj.weight'), 1, 0, dev)), dy, None, B, Fo, Fo, T)
        dsc = 4096.0                                             # own power-of-two scale of the decay columns (~1e-6 below dQ / dK / dV)
        dqkvd = TO.localstate_bwd(ops, r.qkvd.view(R, T, -1), r.att, datt.view(R, T, Cc), R, T, Cc, r.heads, r.ndecay, decay_scale=dsc)
        dqk = dqkvd.view(B, Fo, T, -1)
        nd = r.heads * r.ndecay
        with self.on_param_stream(dqk, r.a):
            dw, db = bw.conv_wgrad(ops, dqk, r.a, [0], [0])
            dw[0, 3 * Cc:3 * Cc + nd] *= 1.0 / dsc                # (parameter-sized fp32 rows: exact power of two)
            db[3 * Cc:3 * Cc + nd] *= 1.0 / dsc
            o = 0
            for nme, n in (('query', Cc), ('key', Cc), ('content', Cc), ('query_decay', nd)):
                self._put(f'{q}.{nme}.weight', dw[0, o:o + n])
                self._put(f'{q}.{nme}.bias', db[o:o + n])
                o += n

        def b_dg():
            w = wq.clone()
            w[3 * Cc:3 * Cc + nd] *= 1.0 / dsc
            return bw.dgrad_conv1d(w[:, :, None], 1, 0, dev)
        return ops.conv(self.spec(q + '.qkvd_dgrad', b_dg), dqk, None, B, Fo, Fo, T, res=dy)

    def _blstm_bwd(self, q, H, r, dy, B, Fo, T):
        ops, dev = self.ops, dy.device
        layers, whh_t, lin, sdl = self._lstm_specs(q, H, dev)
        R = B * Fo
        self._wgrad_to(f'{q}.linear.weight', f'{q}.linear.bias', dy, r.out1s.view(B, Fo, T, 2 * H), [0], [0])
        wl = self.w(f'{q}.linear.weight')
        dout = ops.conv(self.spec(q + '.lin_dgrad', lambda: bw.dgrad_conv1d(self.w(f'{q}.linear.weight')[:, :, None], 1, 0, dev)), dy, None, B, Fo, Fo, T).view(R, T, 2 * H)
        xs = [r.fr0, r.outs[0]]
        for l in (1, 0):
            stitched = r.framed and l == 1
            da = TO.lstm_bwd(ops, dout, whh_t[l], r.saves[l][0], r.saves[l][1], H, r.nseq, r.W, out_mode=1 if stitched else 0,
                             nframes=r.nf, S=r.S, T=T)
            xin = xs[l].reshape(r.nseq, r.W, -1)
            in_ch, npos = xin.shape[-1], r.nseq * r.W

            def b_ih(l=l):                                       # dx = da W_ih: the input projection's data gradient, both directions
                perm = pack.lstm_gate_perm(H, dev)
                wt = torch.cat([self.w(f'{q}.lstm.weight_ih_l{l}{sfx}')[perm] for sfx in ('', '_reverse')], 0).t().contiguous()   # [in_ch, 8H]
                return pack.make_conv_spec(wt[None, :, None, :], None, 8 * H, 0, [0], [0], dev)
            dout = ops.conv(self.spec(q + f'.ih_dgrad{l}', b_ih), da.view(1, 1, npos, 8 * H), None, 1, 1, 1, npos).view(r.nseq, r.W, in_ch)

            def put(kname, rows, perm):                          # scattered straight into the flat gradient buffer (zero before)
                self.g[f'{q}.lstm.{kname}'].index_copy_(0, perm, rows)
            with self.on_param_stream(da, xin, r.outs[l]):
                lstm_param_grads(ops, da, xin, r.outs[l], l, H, r.nseq, r.W, in_ch, dev, put)
        dfr = dout                                               # [nseq, W, H]
        da_in = TO.frames_op(ops, dfr.contiguous(), 1, R, T, H, r.nf, r.W, r.S) if r.framed else dfr
        return TO.add_f16(ops, dy.contiguous(), da_in.reshape(dy.shape).contiguous())


class AeroFunction(torch.autograd.Function):
    """autograd through Aero.forward (solver.py:296-305 forward, :602-605 backward): inputs are the waveform and every parameter;
    the gradient of the waveform output is taken back through the HIP backward pass above."""

    @staticmethod
    def forward(ctx, engine, names, mix, *params):
        with torch.no_grad():
            y, spec_out, lr_spec, c = engine.forward(mix)
        ctx.engine, ctx.c, ctx.names = engine, c, names
        ctx.shapes = [p.shape for p in params]
        ctx.param_ptrs = [p.data_ptr() for p in params]
        ctx.mark_non_differentiable(spec_out, lr_spec)
        return y, spec_out, lr_spec

    @staticmethod
    def backward(ctx, dy, _dspec, _dlr):
        eng = ctx.engine
        dev = dy.device
        offs, n = [], 0
        for s in ctx.shapes:
            offs.append(n)
            n += (s.numel() + 3) // 4 * 4
        # FlatAdam (aero_amd/optim.py) keeps every parameter's .grad as a view of ONE flat buffer with this same layout: the backward
        # then works IN that buffer (freshly zeroed by zero_grad) and hands autograd no per-parameter gradients at all -- its
        # AccumulateGrad nodes were ~300 little `grad += g` launches per step; a buffer that already holds gradients gets one flat add
        sink = getattr(eng.model, '_grad_sink', None)
        sink = sink() if sink is not None else None              # (a weak reference: the optimizer may be gone)
        if sink is not None and not sink.accepts(ctx.param_ptrs, offs, n, dev):
            sink = None
        direct = sink is not None and sink.fresh
        flat = sink.flat_g if direct else torch.zeros(n, dtype=torch.float32, device=dev)
        if sink is not None:
            sink.fresh = False
        views = [flat[o:o + s.numel()].view(s) for o, s in zip(offs, ctx.shapes)]
        grads = dict(zip(ctx.names, views))
        sync = getattr(eng.model, '_grad_sync', None)            # distrib.wrap(): gradient all-reduce over RCCL, overlapped with the backward
        names = ctx.names

        def span(prefix):
            idx = [i for i, nme in enumerate(names) if nme.startswith(prefix)]
            lo, hi = idx[0], idx[-1] + 1
            assert idx == list(range(lo, hi))                    # a module's parameters are contiguous in named_parameters order
            return offs[lo], (offs[hi] if hi < len(offs) else n)

        def stage_done(prefix, scale):
            prefixes = [prefix] + (['freq_emb.'] if prefix == 'encoder.0.' and any(nme.startswith('freq_emb.') for nme in names) else [])
            with eng.on_param_stream(scale):                     # (after every gradient of the stage, whichever stream produced it)
                for pf in prefixes:
                    a, b = span(pf)
                    TO.scale_f32(eng.ops, flat[a:b], sync.unscale(scale) if sync is not None else scale[1:])
                if sync is None:
                    return
                # all-reduce in a few flat segments as stages complete: the whole decoder once its last layer is done (it overlaps with
                # the encoders' backward, the bulk of the time), then each encoder level
                if prefix == 'decoder.0.':
                    a, b = span('decoder.')
                    sync.reduce_async(flat[a:b])
                elif prefix.startswith('encoder.'):
                    for pf in prefixes:
                        a, b = span(pf)
                        sync.reduce_async(flat[a:b])
        def range_of(prefix):
            a, b = span(prefix)
            return flat[a:b]
        with torch.no_grad():
            eng._range_of = range_of
            try:
                eng.backward(ctx.c, dy.contiguous(), grads, stage_done=stage_done)
            finally:
                eng._range_of = None
            if sync is not None:
                sync.wait()
        ctx.c = None
        if sink is not None:
            if not direct:
                sink.flat_g.add_(flat)
            return (None, None, None) + (None,) * len(views)
        return (None, None, None) + tuple(views)


class CapturedStep:
    """One whole training step -- forward, criterion, backward, fused Adam -- captured ONCE as a HIP graph and replayed.

    At BASELINE config 5's per-GPU batch (2 x 10 s) the step is ~500 kernel launches of the library plus a few thousand small
    device-side layout operations (weights re-packed on the device after every optimizer step): a third of the wall time is host work.
    A replay has none.  `step_fn(*inputs)` must run the step on the given STATIC input tensors and return the loss tensor(s); it is
    run eagerly `warmup` times first (lazy tables, the DConv boosts, allocator warm-up), then captured.

        cap = CapturedStep(lambda lr, hr: train_step(lr, hr), lr0, hr0)      # captures on lr0 / hr0's shapes
        loss = cap(lr, hr)                                                    # copies into the static inputs, replays
    """

    def __init__(self, step_fn, *example_inputs, warmup=2, optimizers=(), modules=()):
        self.optimizers = list(optimizers)                       # FlatAdam instances stepped inside step_fn
        # every module whose weights the captured step changes: the optimizers' models plus any named explicitly
        self.modules = [m for m in list(modules) + [getattr(o, 'model', None) for o in self.optimizers] if m is not None and hasattr(m, 'repack')]
        for o in self.optimizers:
            o.prepare_capture()
        self.static_in = [t.clone() for t in example_inputs]
        from .engine import side_streams
        side = side_streams(example_inputs[0].device, 2)[1]       # (not the parameter-gradient stream, which forks from this one inside the capture)
        # the package's side streams are SHARED (engine.side_streams: BatchPipeline's ring, the half-batch streams): whatever they still carry
        # -- pipelined inference batches of the same process -- must be done before warm-up and capture run on them
        torch.cuda.synchronize(example_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                            # (capture must not run on the legacy default stream)
            for _ in range(warmup):
                step_fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = step_fn(*self.static_in)

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            s.copy_(t)
        for o in self.optimizers:
            o.before_replay()
        self.graph.replay()
        # the replay stepped the weights (and re-packed the TRAINING images, captured with the step) behind every cache key kept on the
        # host: bump them, so that an eval-mode forward, an eager step or the critic's cached record re-pack instead of running on the
        # weights of capture time
        for m in self.modules:
            m.repack()
        return self.static_out
