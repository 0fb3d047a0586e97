"""CPU oracle for the AERO spectral forward/inverse path.  TEST INFRASTRUCTURE ONLY.

This file restates, op by op, the algorithm of the reference's ``Aero.forward``
(/root/reference/src/models/aero.py:446-523) as plain functional fp32 code over a
``state_dict``.  It exists so that the HIP path can be checked on the GPU box, where
the reference's Python sources are not available.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; the product path (``aero_amd``) never does.

Pinning status: the reference ships no golden vectors or tests of its own
(SURVEY.md section 4).  The oracle is pinned instead by outputs of the reference
itself: ``oracle/make_golden.py`` imports /root/reference in the build container,
runs it on seeded inputs and commits the vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file against them (<= 1e-6 rel-L2).

The arithmetic of the path lives in PyTorch ATen (pinned by the reference at
torch==1.12.1, requirements.txt:10).  The STFT/iSTFT, LSTM recurrence, attention and
all index arithmetic are written out explicitly here; dense convolutions call
``torch.nn.functional`` on CPU (that *is* the reference's own arithmetic).
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------
# integer / sample-index paths -- must be bit exact
# ----------------------------------------------------------------------------------


def derive_geometry(nfft, hop_length, lr_sr, hr_sr, spec_upsample=True):
    """aero.py:322-328 -- scale, input hop and input window."""
    scale = hr_sr / lr_sr if spec_upsample else 1
    hop_in = int(hop_length // scale)
    win_in = int(nfft // scale)
    return scale, hop_in, win_in


def spec_pad_amount(length, hop_in):
    """aero.py:410-411 -- right zero pad so that length is a multiple of the hop."""
    r = length % hop_in
    return (hop_in - r) if r else 0


def stft_num_frames(length_padded, hop):
    """torch.stft(center=True): 1 + L/hop frames (spec.py:12-20)."""
    return 1 + length_padded // hop


def unfold_geometry(length, kernel_size, stride):
    """models/utils.py:29-31 -- frame count and padded length of `unfold`."""
    n_frames = math.ceil(length / stride)
    tgt_length = (n_frames - 1) * stride + kernel_size
    return n_frames, tgt_length


def stitch_map(T, width, stride, n_frames):
    """modules.py:49-62 -- for each output step t, (frame k, in-frame step tau).

    Frame 0 keeps [0, width-limit), middle frames [limit, width-limit), the last frame
    [limit, width); the concatenation is cropped to T.
    """
    limit = stride // 2
    out = []
    for k in range(n_frames):
        if k == 0:
            lo, hi = 0, width - limit
        elif k == n_frames - 1:
            lo, hi = limit, width
        else:
            lo, hi = limit, width - limit
        # NB: for n_frames == 1 the reference's `k == 0` branch wins: [0, width-limit)
        out += [(k, tau) for tau in range(lo, hi)]
    return out[:T]


def predict_chunks(n_samples, sr, segment_sec=10):
    """predict.py:61-69 -- [start, end) sample ranges of the independent chunks."""
    seg = sr * segment_sec
    n_chunks = math.ceil(n_samples / seg)
    return [(i * seg, min((i + 1) * seg, n_samples)) for i in range(n_chunks)]


def output_length(length, scale):
    """aero.py:513."""
    return int(length * scale)


def match_signal_length(n, target):
    """src/utils.py:211-217 -- pad (>0) or crop (<0) amount to reach `target`."""
    return target - n


# ----------------------------------------------------------------------------------
# STFT / iSTFT written out (spec.py:9-39 over torch.stft / torch.istft semantics)
# ----------------------------------------------------------------------------------


def hann_periodic(n, dtype=torch.float32):
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * math.pi * k / n)).to(dtype)


def _padded_window(win_length, n_fft):
    w = torch.zeros(n_fft, dtype=torch.float32)
    left = (n_fft - win_length) // 2
    w[left:left + win_length] = hann_periodic(win_length)
    return w


def stft(x, n_fft, hop, win_length):
    """x [..., L] real -> [..., n_fft/2+1, T] complex64.  spec.py:9-22 (Appendix B1)."""
    *other, L = x.shape
    x = x.reshape(-1, L)
    p = n_fft // 2
    xp = F.pad(x[:, None, :], (p, p), mode='reflect')[:, 0, :]
    T = 1 + L // hop
    idx = torch.arange(T)[:, None] * hop + torch.arange(n_fft)[None, :]
    frames = xp[:, idx] * _padded_window(win_length, n_fft)           # [N, T, n_fft]
    z = torch.fft.rfft(frames, dim=-1) * (n_fft ** -0.5)              # normalized=True
    z = z.transpose(1, 2)
    return z.reshape(*other, n_fft // 2 + 1, T)


def istft(z, hop, win_length):
    """z [..., n_fft/2+1, T] complex -> [..., hop*(T-1)] real.  spec.py:25-39 (Appendix B10)."""
    *other, freqs, T = z.shape
    n_fft = 2 * freqs - 2
    z = z.reshape(-1, freqs, T)
    w = _padded_window(win_length, n_fft)
    frames = torch.fft.irfft(z.transpose(1, 2) * (n_fft ** 0.5), n=n_fft, dim=-1)  # [N,T,n_fft]
    frames = frames * w
    total = n_fft + hop * (T - 1)
    y = torch.zeros(z.shape[0], total, dtype=frames.dtype)
    env = torch.zeros(total, dtype=frames.dtype)
    w2 = w * w
    for t in range(T):
        y[:, t * hop:t * hop + n_fft] += frames[:, t]
        env[t * hop:t * hop + n_fft] += w2
    p = n_fft // 2
    y = y[:, p:total - p] / env[p:total - p]
    return y.reshape(*other, y.shape[-1])


# ----------------------------------------------------------------------------------
# building blocks (modules.py, snake.py)
# ----------------------------------------------------------------------------------


def snake(x, a):
    """snake.py:67 -- `a` broadcasts on the last dim."""
    return x + (1.0 / a) * torch.sin(x * a) ** 2


def lstm_layer_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.LSTM layer, written out.  x [S, N, I] -> [S, N, H].

    Gate order i, f, g, o; c' = f*c + i*g; h' = o*tanh(c') (modules.py:28, Appendix B6).
    """
    S, N, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(N, H)
    c = x.new_zeros(N, H)
    xp = x @ w_ih.t() + (b_ih + b_hh)
    out = [None] * S
    steps = range(S - 1, -1, -1) if reverse else range(S)
    for s in steps:
        g = xp[s] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[s] = h
    return torch.stack(out, 0)


def blstm(sd, pre, x, max_steps=200, layers=2, fast=False):
    """modules.py:32-65.  x [N, C, T] -> [N, C, T] (skip=True in DConv, modules.py:217)."""
    N, C, T = x.shape
    y = x
    framed = False
    if max_steps is not None and T > max_steps:
        width = max_steps
        stride = width // 2
        n_frames, tgt = unfold_geometry(T, width, stride)
        xp = F.pad(x, (0, tgt - T))
        idx = torch.arange(n_frames)[:, None] * stride + torch.arange(width)[None, :]
        frames = xp[:, :, idx]                                 # [N, C, n_frames, width]
        x = frames.permute(0, 2, 1, 3).reshape(-1, C, width)
        framed = True
    x = x.permute(2, 0, 1)                                     # [S, N', C]
    if fast:
        flat = []
        for l in range(layers):
            for sfx in ('', '_reverse'):
                flat += [sd[f'{pre}.lstm.weight_ih_l{l}{sfx}'], sd[f'{pre}.lstm.weight_hh_l{l}{sfx}'],
                         sd[f'{pre}.lstm.bias_ih_l{l}{sfx}'], sd[f'{pre}.lstm.bias_hh_l{l}{sfx}']]
        hx = x.new_zeros(2 * layers, x.shape[1], C)
        x = torch.lstm(x, (hx, hx.clone()), flat, True, layers, 0.0, False, True, False)[0]
    else:
        for l in range(layers):
            outs = []
            for sfx, rev in (('', False), ('_reverse', True)):
                outs.append(lstm_layer_dir(
                    x, sd[f'{pre}.lstm.weight_ih_l{l}{sfx}'], sd[f'{pre}.lstm.weight_hh_l{l}{sfx}'],
                    sd[f'{pre}.lstm.bias_ih_l{l}{sfx}'], sd[f'{pre}.lstm.bias_hh_l{l}{sfx}'], rev))
            x = torch.cat(outs, dim=2)
    x = x @ sd[f'{pre}.linear.weight'].t() + sd[f'{pre}.linear.bias']
    x = x.permute(1, 2, 0)                                     # [N', C, S]
    if framed:
        frames = x.reshape(N, -1, C, width)
        smap = stitch_map(T, width, stride, n_frames)
        ks = torch.tensor([k for k, _ in smap])
        ts = torch.tensor([t for _, t in smap])
        x = frames[:, ks, :, ts].permute(1, 2, 0)              # [N, C, T]
    return x + y


def local_state(sd, pre, x, heads=4, ndecay=4):
    """modules.py:94-127 (Appendix B7).  x [N, C, T]."""
    N, C, T = x.shape

    def c1(name):
        return F.conv1d(x, sd[f'{pre}.{name}.weight'], sd[f'{pre}.{name}.bias'])
    q = c1('query').view(N, heads, -1, T)
    k = c1('key').view(N, heads, -1, T)
    v = c1('content').view(N, heads, -1, T)
    dots = torch.einsum('bhct,bhcs->bhts', k, q) / (k.shape[2] ** 0.5)
    idx = torch.arange(T, dtype=x.dtype)
    delta = (idx[:, None] - idx[None, :]).abs()                # [t, s]
    dq = torch.sigmoid(c1('query_decay').view(N, heads, ndecay, T)) / 2
    dec = torch.arange(1, ndecay + 1, dtype=x.dtype)
    kern = -dec.view(-1, 1, 1) * delta / (ndecay ** 0.5)       # [f, t, s]
    dots = dots + torch.einsum('fts,bhfs->bhts', kern, dq)
    dots = dots.masked_fill(torch.eye(T, dtype=torch.bool), -100.0)
    w = torch.softmax(dots, dim=2)
    res = torch.einsum('bhts,bhct->bhcs', w, v).reshape(N, -1, T)
    return x + F.conv1d(res, sd[f'{pre}.proj.weight'], sd[f'{pre}.proj.bias'])


def dconv(sd, pre, x, depth, lstm, time_attn, fast=False):
    """modules.py:221-249 (Appendix B5).  x [B, C, Fr, T]."""
    B, C, Fr, T = x.shape
    x = x.permute(0, 2, 1, 3).reshape(-1, C, T)
    for d in range(depth):
        p = f'{pre}.layers.{d}'
        skip = x
        dil = 2 ** d
        h = F.conv1d(x, sd[f'{p}.conv1.0.weight'], sd[f'{p}.conv1.0.bias'], dilation=dil, padding=dil)
        hid = h.shape[1]
        h = F.group_norm(h, 1, sd[f'{p}.conv1.1.weight'], sd[f'{p}.conv1.1.bias'], 1e-5)
        h = h.view(B, Fr, hid, T).permute(0, 2, 3, 1)
        h = snake(h, sd[f'{p}.act.a'])
        h = h.permute(0, 3, 1, 2).reshape(-1, hid, T)
        if lstm:
            h = blstm(sd, f'{p}.lstm', h, fast=fast)
        if time_attn:
            h = local_state(sd, f'{p}.time_attn', h)
        h = F.conv1d(h, sd[f'{p}.conv2.0.weight'], sd[f'{p}.conv2.0.bias'])
        h = F.group_norm(h, 1, sd[f'{p}.conv2.1.weight'], sd[f'{p}.conv2.1.bias'], 1e-5)
        h = F.glu(h, dim=1)
        h = sd[f'{p}.conv2.3.scale'][:, None] * h
        x = skip + h
    return x.view(B, Fr, C, T).permute(0, 2, 1, 3)


def _bn_eval(x, sd, pre):
    shape = [1, -1] + [1] * (x.dim() - 2)
    rm, rv = sd[f'{pre}.running_mean'], sd[f'{pre}.running_var']
    w, b = sd[f'{pre}.weight'], sd[f'{pre}.bias']
    return (x - rm.view(shape)) / torch.sqrt(rv.view(shape) + 1e-5) * w.view(shape) + b.view(shape)


def _bn_train(x, sd, pre, new_stats=None, momentum=0.1):
    """nn.BatchNorm{1,2}d in training mode (modules.py:287,293,300): normalise with the statistics of THIS batch (biased
    variance) and, when `new_stats` is a dict, record the running-statistics update PyTorch performs
    (running = (1-m) running + m batch, with the UNBIASED batch variance; num_batches_tracked + 1)."""
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=dims)
    var = x.var(dim=dims, unbiased=False)
    if new_stats is not None:
        new_stats[f'{pre}.running_mean'] = (1 - momentum) * sd[f'{pre}.running_mean'] + momentum * mean
        new_stats[f'{pre}.running_var'] = (1 - momentum) * sd[f'{pre}.running_var'] + momentum * var * (n / (n - 1))
        new_stats[f'{pre}.num_batches_tracked'] = sd[f'{pre}.num_batches_tracked'] + 1
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + 1e-5) * sd[f'{pre}.weight'].view(shape) + sd[f'{pre}.bias'].view(shape)


def ftb(sd, pre, x, train=False, new_stats=None):
    """modules.py:304-325 (Appendix B4).  eval: BatchNorm on running statistics; train: on batch statistics."""
    bn = (lambda v, q: _bn_train(v, sd, q, new_stats)) if train else (lambda v, q: _bn_eval(v, sd, q))
    B, C, D, T = x.shape
    c1 = F.relu(bn(F.conv2d(x, sd[f'{pre}.conv1.0.weight'], sd[f'{pre}.conv1.0.bias']), f'{pre}.conv1.1'))
    r = c1.shape[1]
    c1 = c1.reshape(B, r * D, T)
    c2 = F.relu(bn(F.conv1d(c1, sd[f'{pre}.conv1d.0.weight'], sd[f'{pre}.conv1d.0.bias'], padding=4), f'{pre}.conv1d.1'))
    att = c2.reshape(B, C, 1, T) * x
    att = (att.transpose(2, 3) @ sd[f'{pre}.freq_fc.weight'].t()).transpose(2, 3)
    cat = torch.cat([att, x], 1)
    return F.relu(bn(F.conv2d(cat, sd[f'{pre}.conv2.0.weight'], sd[f'{pre}.conv2.0.bias']), f'{pre}.conv2.1'))


def _norm(sd, pre, x, groups):
    if f'{pre}.weight' in sd:
        return F.group_norm(x, groups, sd[f'{pre}.weight'], sd[f'{pre}.bias'], 1e-5)
    return x


def enc_layer(sd, i, x, cfg, fast=False, train=False, new_stats=None):
    """aero.py:108-135 (Appendix B3)."""
    p = f'encoder.{i}'
    s = cfg['strides'][i]
    if f'{p}.pre_conv.weight' in sd:
        x = F.conv2d(x, sd[f'{p}.pre_conv.weight'], sd[f'{p}.pre_conv.bias'])
    if f'{p}.freq_attn_block.freq_fc.weight' in sd:
        x = ftb(sd, f'{p}.freq_attn_block', x, train, new_stats)
    w = sd[f'{p}.conv.weight']
    K = w.shape[2]
    x = F.conv2d(x, w, sd[f'{p}.conv.bias'], stride=(s, 1), padding=((K - s) // 2, 0))
    x = F.gelu(_norm(sd, f'{p}.norm1', x, cfg['norm_groups']))
    if cfg['dconv_mode'] & 1:
        x = dconv(sd, f'{p}.dconv', x, cfg['dconv_depth'], i >= cfg['dconv_lstm'], i >= cfg['dconv_time_attn'], fast)
    if f'{p}.rewrite.weight' in sd:
        ctx = cfg['context_enc']
        x = F.conv2d(x, sd[f'{p}.rewrite.weight'], sd[f'{p}.rewrite.bias'], padding=ctx)
        x = F.glu(_norm(sd, f'{p}.norm2', x, cfg['norm_groups']), dim=1)
    return x


def dec_layer(sd, j, x, skip, cfg, last):
    """aero.py:189-215 (Appendix B9)."""
    p = f'decoder.{j}'
    depth = len(cfg['strides'])
    s = cfg['strides'][depth - 1 - j]
    x = torch.cat([x, skip], dim=1)
    y = F.conv2d(x, sd[f'{p}.rewrite.weight'], sd[f'{p}.rewrite.bias'], padding=cfg['context'])
    y = F.glu(_norm(sd, f'{p}.norm1', y, cfg['norm_groups']), dim=1)
    w = sd[f'{p}.conv_tr.weight']
    K = w.shape[2]
    z = F.conv_transpose2d(y, w, sd[f'{p}.conv_tr.bias'], stride=(s, 1))
    z = _norm(sd, f'{p}.norm2', z, cfg['norm_groups'])
    pad = (K - s) // 2
    if pad:
        z = z[..., pad:-pad, :]
    if not last:
        z = F.gelu(z)
    return z


DEFAULT_CFG = dict(in_channels=1, out_channels=1, audio_channels=2, channels=48, growth=2, nfft=512,
                   hop_length=64, end_iters=0, cac=True, rewrite=True, hybrid=False, hybrid_old=False,
                   freq_emb=0.2, emb_scale=10, emb_smooth=True, kernel_size=8, strides=[4, 4, 2, 2],
                   context=1, context_enc=0, freq_ends=4, enc_freq_attn=4, norm_starts=2, norm_groups=4,
                   dconv_mode=1, dconv_depth=2, dconv_comp=4, dconv_time_attn=2, dconv_lstm=2,
                   dconv_init=1e-3, rescale=0.1, lr_sr=4000, hr_sr=16000, spec_upsample=True,
                   act_func='snake', debug=False)


def spec(x, cfg, scale=False):
    """aero.py:409-421."""
    sc, hop_in, win_in = derive_geometry(cfg['nfft'], cfg['hop_length'], cfg['lr_sr'], cfg['hr_sr'],
                                         cfg['spec_upsample'])
    pad = spec_pad_amount(x.shape[-1], hop_in)
    if pad:
        x = F.pad(x, (0, pad))
    hl, wl = hop_in, win_in
    if scale:
        hl, wl = int(hl * sc), int(wl * sc)
    return stft(x, cfg['nfft'], hl, wl)[..., :-1, :]


def ispec(z, cfg):
    """aero.py:423-428."""
    sc, hop_in, win_in = derive_geometry(cfg['nfft'], cfg['hop_length'], cfg['lr_sr'], cfg['hr_sr'],
                                         cfg['spec_upsample'])
    z = F.pad(z, (0, 0, 0, 1))
    return istft(z, int(hop_in * sc), int(win_in * sc))


def aero_forward(sd, cfg, mix, return_spec=False, return_lr_spec=False, fast=False, taps=None, train=False, new_stats=None):
    """aero.py:446-523.  `sd` is a state_dict of fp32 CPU tensors, `cfg` the ctor kwargs.

    `taps`, if a dict, receives intermediate tensors (for drift localisation in tests).
    `train`: the module in training mode -- the only forward-path difference is the FTB's BatchNorm (batch statistics);
    `new_stats` (a dict) then receives the updated running statistics.
    """
    cfg = {**DEFAULT_CFG, **cfg}
    sc, _, _ = derive_geometry(cfg['nfft'], cfg['hop_length'], cfg['lr_sr'], cfg['hr_sr'], cfg['spec_upsample'])
    length = mix.shape[-1]
    z = spec(mix, cfg)
    B, C, Fq, T = z.shape
    x = torch.view_as_real(z).permute(0, 1, 4, 2, 3).reshape(B, C * 2, Fq, T)
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    std = x.std(dim=(1, 2, 3), keepdim=True)
    x = (x - mean) / (1e-5 + std)
    saved = []
    depth = len(cfg['strides'])
    for i in range(depth):
        x = enc_layer(sd, i, x, cfg, fast, train, new_stats)
        if i == 0 and 'freq_emb.embedding.weight' in sd:
            emb = (sd['freq_emb.embedding.weight'] * cfg['emb_scale']).t()[None, :, :, None]
            x = x + cfg['freq_emb'] * emb
        if taps is not None:
            taps[f'enc{i}'] = x
        saved.append(x)
    x = torch.zeros_like(x)
    for j in range(depth):
        x = dec_layer(sd, j, x, saved.pop(-1), cfg, last=(j == depth - 1))
        if taps is not None:
            taps[f'dec{j}'] = x
    x = x.view(B, cfg['out_channels'], -1, Fq, T)
    x = x * std[:, None] + mean[:, None]
    zc = torch.view_as_complex(x.permute(0, 1, 3, 4, 2).contiguous())
    y = ispec(zc, cfg)[..., :output_length(length, sc)]
    if return_spec:
        return (y, zc, z) if return_lr_spec else (y, zc)
    return y


# ------------------------------------------------------------------------------------------------
# multi-resolution STFT loss (src/models/stft_loss.py), restated with return_complex=True: the reference's
# `torch.stft(x, fft_size, hop_size, win_length, window)` (stft_loss.py:22) raises on torch >= 2.
def stft_magnitude(x, fft_size, hop_size, win_length):
    """stft_loss.py:11-27: sqrt(clamp(re^2 + im^2, 1e-7)) of the centred, reflect-padded, un-normalised STFT -> [B, frames, bins]"""
    z = torch.stft(x, fft_size, hop_size, win_length, torch.hann_window(win_length, dtype=x.dtype), return_complex=True)
    return torch.sqrt(torch.clamp(z.real ** 2 + z.imag ** 2, min=1e-7)).transpose(2, 1)


def mrstft_loss(x, y, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240), factor_sc=0.1, factor_mag=0.1):
    """stft_loss.py:84-138: (factor_sc * mean spectral convergence, factor_mag * mean log-magnitude L1); x prediction, y target [B, T]"""
    sc = mag = 0.0
    for n, h, w in zip(fft_sizes, hop_sizes, win_lengths):
        xm, ym = stft_magnitude(x, n, h, w), stft_magnitude(y, n, h, w)
        sc = sc + torch.norm(ym - xm, p='fro') / torch.norm(ym, p='fro')          # stft_loss.py:47
        mag = mag + torch.nn.functional.l1_loss(torch.log(ym), torch.log(xm))    # stft_loss.py:64
    k = len(fft_sizes)
    return factor_sc * sc / k, factor_mag * mag / k


# ------------------------------------------------------------------------------------------------
# The oracle with the PRODUCT's storage precision (test infrastructure for the gradient checks): inside `fp16_storage()` the
# outputs of the convolutions and activations are rounded to fp16 with a straight-through gradient -- the points where
# aero_amd keeps activations in fp16.  Nothing else changes (fp32 arithmetic, statistics, recurrences).
# Gradients of ReLU / BatchNorm-on-batch-statistics layers are discontinuous in the forward activations (a ReLU mask flips
# when a pre-activation within the 1e-3 forward tolerance of zero changes sign), so a backward pass can only be compared
# tightly against a forward that made the same rounding decisions.
class _RoundedF:
    ROUND = ('conv1d', 'conv2d', 'conv_transpose2d', 'gelu', 'glu', 'relu')

    def __getattr__(self, name):
        f = getattr(torch.nn.functional, name)
        if name not in self.ROUND:
            return f

        def g(*a, **k):
            y = f(*a, **k)
            return y + (y.half().float() - y).detach()
        return g


class fp16_storage:
    def __enter__(self):
        global F
        self._F = F
        F = _RoundedF()

    def __exit__(self, *exc):
        global F
        F = self._F
