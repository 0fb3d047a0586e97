"""Parameter tree of the AERO generator (host-side mirror of the reference interface).

The classes below hold *parameters only*: same submodule names, shapes, default inits and
construction order as the reference (`/root/reference/src/models/aero.py:223-407`,
`modules.py:17-325`, `snake.py:37-56`), so that

  * ``state_dict()`` / ``load_state_dict()`` use the reference's 331 keys, and
  * ``torch.manual_seed(s); Aero(**kw)`` draws bit-identical random-init weights.

They carry no forward arithmetic.  ``Aero.forward`` hands the tree to
``aero_amd.engine.HipEngine`` which runs the whole STFT -> U-Net -> iSTFT path through the
hand-written gfx950 kernels behind the C-ABI in ``include/aero_hip.h``.
"""
import functools

import torch
from torch import nn
from torch.distributions.exponential import Exponential


def capture_init(init):
    """Record ctor (args, kwargs) on the instance; the checkpoint serializer reads
    ``_init_args_kwargs`` (reference models/utils.py:7-19, model_serializer.py:20-22)."""
    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        self._init_args_kwargs = (args, kwargs)
        init(self, *args, **kwargs)
    return wrapped


class _ParamsOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f'{type(self).__name__} holds parameters only; run the model through Aero.forward '
                           '(HIP engine)')


class Snake(_ParamsOnly):
    """x + sin^2(a x)/a with one `a` per frequency bin (snake.py:53-54)."""

    def __init__(self, in_features):
        super().__init__()
        self.in_features = in_features if isinstance(in_features, list) else [in_features]
        self.a = nn.Parameter(Exponential(torch.tensor([0.1])).rsample(self.in_features).squeeze())


class LayerScale(_ParamsOnly):
    def __init__(self, channels, init=0.0):
        super().__init__()
        self.scale = nn.Parameter(torch.full((channels,), float(init)))


class BLSTM(_ParamsOnly):
    """modules.py:24-30."""

    def __init__(self, dim, layers=1, max_steps=None, skip=False):
        super().__init__()
        assert max_steps is None or max_steps % 4 == 0
        self.max_steps, self.skip, self.dim, self.layers = max_steps, skip, dim, layers
        self.lstm = nn.LSTM(bidirectional=True, num_layers=layers, hidden_size=dim, input_size=dim)
        self.linear = nn.Linear(2 * dim, dim)


class LocalState(_ParamsOnly):
    """modules.py:74-92."""

    def __init__(self, channels, heads=4, nfreqs=0, ndecay=4):
        super().__init__()
        assert channels % heads == 0, (channels, heads)
        if nfreqs:
            raise NotImplementedError('nfreqs>0 is a dead branch in the reference (modules.py:105-110)')
        self.heads, self.nfreqs, self.ndecay = heads, nfreqs, ndecay
        self.content = nn.Conv1d(channels, channels, 1)
        self.query = nn.Conv1d(channels, channels, 1)
        self.key = nn.Conv1d(channels, channels, 1)
        if ndecay:
            self.query_decay = nn.Conv1d(channels, heads * ndecay, 1)
            self.query_decay.weight.data *= 0.01
            self.query_decay.bias.data[:] = -2
        self.proj = nn.Conv1d(channels, channels, 1)


class DConv(_ParamsOnly):
    """modules.py:152-219."""

    def __init__(self, channels, compress=4, depth=2, init=1e-4, norm=True, time_attn=False, heads=4,
                 ndecay=4, lstm=False, act_func='gelu', freq_dim=None, reshape=False, kernel=3, dilate=True):
        super().__init__()
        assert kernel % 2 == 1
        self.channels, self.compress, self.depth = channels, compress, abs(depth)
        self.dilate = depth > 0
        self.time_attn, self.lstm, self.reshape = time_attn, lstm, reshape
        self.act_func, self.freq_dim, self.kernel, self.norm = act_func, freq_dim, kernel, norm
        self.hidden = int(channels / compress)
        self.layers = nn.ModuleList()
        for d in range(self.depth):
            dilation = 2 ** d if self.dilate else 1
            layer = nn.ModuleDict()
            conv1 = nn.Sequential(
                nn.Conv1d(channels, self.hidden, kernel, dilation=dilation, padding=dilation * (kernel // 2)),
                nn.GroupNorm(1, self.hidden) if norm else nn.Identity())
            if act_func == 'snake':
                act = Snake(freq_dim)
            elif act_func == 'gelu':
                act = nn.GELU()
            else:
                act = nn.ReLU()
            conv2 = nn.Sequential(nn.Conv1d(self.hidden, 2 * channels, 1),
                                  nn.GroupNorm(1, 2 * channels) if norm else nn.Identity(),
                                  nn.GLU(1), LayerScale(channels, init))
            layer.update({'conv1': conv1, 'act': act, 'conv2': conv2})
            if lstm:
                layer.update({'lstm': BLSTM(self.hidden, layers=2, max_steps=200, skip=True)})
            if time_attn:
                layer.update({'time_attn': LocalState(self.hidden, heads=heads, ndecay=ndecay)})
            self.layers.append(layer)


class ScaledEmbedding(_ParamsOnly):
    """modules.py:258-268."""

    def __init__(self, num_embeddings, embedding_dim, scale=10., smooth=False):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings, embedding_dim)
        if smooth:
            w = torch.cumsum(self.embedding.weight.data, dim=0)
            w = w / torch.arange(1, num_embeddings + 1).to(w).sqrt()[:, None]
            self.embedding.weight.data[:] = w
        self.embedding.weight.data /= scale
        self.scale = scale


class FTB(_ParamsOnly):
    """modules.py:281-302."""

    def __init__(self, input_dim=257, in_channel=9, r_channel=5):
        super().__init__()
        self.input_dim, self.in_channel, self.r_channel = input_dim, in_channel, r_channel
        self.conv1 = nn.Sequential(nn.Conv2d(in_channel, r_channel, kernel_size=[1, 1]),
                                   nn.BatchNorm2d(r_channel), nn.ReLU())
        self.conv1d = nn.Sequential(nn.Conv1d(r_channel * input_dim, in_channel, kernel_size=9, padding=4),
                                    nn.BatchNorm1d(in_channel), nn.ReLU())
        self.freq_fc = nn.Linear(input_dim, input_dim, bias=False)
        self.conv2 = nn.Sequential(nn.Conv2d(in_channel * 2, in_channel, kernel_size=[1, 1]),
                                   nn.BatchNorm2d(in_channel), nn.ReLU())


def _layer_geometry(kernel_size, stride, pad):
    if stride == 1 and kernel_size % 2 == 0 and kernel_size > 1:
        kernel_size -= 1
    return kernel_size, ((kernel_size - stride) // 2 if pad else 0)


class HEncLayer(_ParamsOnly):
    """aero.py:32-106."""

    def __init__(self, chin, chout, kernel_size=8, stride=4, norm_groups=1, empty=False, freq=True, dconv=True,
                 is_first=False, freq_attn=False, freq_dim=None, norm=True, context=0, dconv_kw={}, pad=True,
                 rewrite=True):
        super().__init__()
        if not freq:
            raise NotImplementedError('time-axis encoder layers (freq=False) are unreachable from the '
                                      'reference configs (aero.py:349, freq_ends=4)')
        kernel_size, pad = _layer_geometry(kernel_size, stride, pad)
        self.chin, self.chout, self.freq = chin, chout, freq
        self.kernel_size, self.stride, self.empty, self.pad = kernel_size, stride, empty, pad
        self.freq_attn, self.freq_dim, self.norm, self.is_first = freq_attn, freq_dim, norm, is_first
        self.norm_groups, self.context = norm_groups, context

        def norm_fn(d):
            return nn.GroupNorm(norm_groups, d) if norm else nn.Identity()
        if is_first:
            self.pre_conv = nn.Conv2d(chin, chout, [1, 1])
            chin = chout
        if freq_attn:
            self.freq_attn_block = FTB(input_dim=freq_dim, in_channel=chin)
        self.conv = nn.Conv2d(chin, chout, [kernel_size, 1], [stride, 1], [pad, 0] if pad else 0)
        if empty:
            return
        self.norm1 = norm_fn(chout)
        self.rewrite = None
        if rewrite:
            self.rewrite = nn.Conv2d(chout, 2 * chout, 1 + 2 * context, 1, context)
            self.norm2 = norm_fn(2 * chout)
        self.dconv = DConv(chout, **dconv_kw) if dconv else None


class HDecLayer(_ParamsOnly):
    """aero.py:139-187."""

    def __init__(self, chin, chout, last=False, kernel_size=8, stride=4, norm_groups=1, empty=False, freq=True,
                 dconv=True, norm=True, context=1, dconv_kw={}, pad=True, context_freq=True, rewrite=True):
        super().__init__()
        if not freq or not context_freq or empty:
            raise NotImplementedError('only the frequency-axis decoder layer of the reference configs is built')
        kernel_size, pad = _layer_geometry(kernel_size, stride, pad)
        self.pad, self.last, self.freq, self.chin, self.chout = pad, last, freq, chin, chout
        self.empty, self.stride, self.kernel_size, self.norm = empty, stride, kernel_size, norm
        self.norm_groups, self.context, self.context_freq = norm_groups, context, context_freq

        def norm_fn(d):
            return nn.GroupNorm(norm_groups, d) if norm else nn.Identity()
        self.conv_tr = nn.ConvTranspose2d(chin, chout, [kernel_size, 1], [stride, 1])
        self.norm2 = norm_fn(chout)
        self.rewrite = None
        if rewrite:
            self.rewrite = nn.Conv2d(chin, 2 * chin, 1 + 2 * context, 1, context)
            self.norm1 = norm_fn(2 * chin)
        self.dconv = DConv(chin, **dconv_kw) if dconv else None


def rescale_module(module, reference):
    """aero.py:17-28 -- note: only Conv1d / ConvTranspose1d are touched."""
    for sub in module.modules():
        if isinstance(sub, (nn.Conv1d, nn.ConvTranspose1d)):
            std = sub.weight.std().detach()
            scale = (std / reference) ** 0.5
            sub.weight.data /= scale
            if sub.bias is not None:
                sub.bias.data /= scale


class Aero(nn.Module):
    """Drop-in for ``src.models.aero.Aero`` (aero.py:218-523): same 35 ctor kwargs, attributes,
    state_dict keys and ``forward`` signature; the arithmetic runs on gfx950 through the C-ABI."""
    _supports_grad_sync = True                                   # distrib.wrap: AeroFunction.backward averages the gradients over the ranks itself

    @capture_init
    def __init__(self, in_channels=1, out_channels=1, audio_channels=2, channels=48, growth=2, nfft=512,
                 hop_length=64, end_iters=0, cac=True, rewrite=True, hybrid=False, hybrid_old=False,
                 freq_emb=0.2, emb_scale=10, emb_smooth=True, kernel_size=8, strides=[4, 4, 2, 2], context=1,
                 context_enc=0, freq_ends=4, enc_freq_attn=4, norm_starts=2, norm_groups=4, dconv_mode=1,
                 dconv_depth=2, dconv_comp=4, dconv_time_attn=2, dconv_lstm=2, dconv_init=1e-3, rescale=0.1,
                 lr_sr=4000, hr_sr=16000, spec_upsample=True, act_func='snake', debug=False):
        super().__init__()
        self.cac, self.in_channels, self.out_channels = cac, in_channels, out_channels
        self.audio_channels, self.kernel_size, self.context = audio_channels, kernel_size, context
        self.context_enc = context_enc
        self.strides = list(strides)
        self.depth = len(self.strides)
        self.channels, self.lr_sr, self.hr_sr, self.spec_upsample = channels, lr_sr, hr_sr, spec_upsample
        self.scale = hr_sr / lr_sr if spec_upsample else 1
        self.nfft = nfft
        self.hop_length = int(hop_length // self.scale)      # input-signal hop  (aero.py:327)
        self.win_length = int(nfft // self.scale)            # input-signal window (aero.py:328)
        self.end_iters, self.hybrid, self.hybrid_old, self.debug = end_iters, hybrid, hybrid_old, debug
        self.norm_groups, self.dconv_mode, self.dconv_depth = norm_groups, dconv_mode, dconv_depth
        self.act_func = act_func
        if not cac:
            raise NotImplementedError('cac=False is not a configuration the reference ships')
        self.freq_emb = None
        self.encoder = nn.ModuleList()
        self.decoder = nn.ModuleList()

        chin_z = in_channels * 2
        chout_z = channels
        freqs = nfft // 2
        for index in range(self.depth):
            stri = self.strides[index]
            freq = index <= freq_ends
            ker = freqs if (freq and freqs < kernel_size) else kernel_size
            kw = dict(kernel_size=ker, stride=stri, freq=freq, pad=True, norm=index >= norm_starts,
                      rewrite=rewrite, norm_groups=norm_groups,
                      dconv_kw=dict(lstm=index >= dconv_lstm, time_attn=index >= dconv_time_attn,
                                    depth=dconv_depth, compress=dconv_comp, init=dconv_init,
                                    act_func=act_func, reshape=True,
                                    freq_dim=freqs // stri if freq else freqs))
            self.encoder.append(HEncLayer(chin_z, chout_z, dconv=bool(dconv_mode & 1), context=context_enc,
                                          is_first=index == 0, freq_attn=index >= enc_freq_attn,
                                          freq_dim=freqs, **kw))
            if index == 0:
                chin_z = out_channels * 2
            self.decoder.insert(0, HDecLayer(2 * chout_z, chin_z, dconv=bool(dconv_mode & 2),
                                             last=index == 0, context=context, **kw))
            chin_z = chout_z
            chout_z = int(growth * chout_z)
            if freq:
                freqs //= stri
            if index == 0 and freq_emb:
                self.freq_emb = ScaledEmbedding(freqs, chin_z, smooth=emb_smooth, scale=emb_scale)
                self.freq_emb_scale = freq_emb
        if rescale:
            rescale_module(self, reference=rescale)
        self._engine = None

    # ------------------------------------------------------------------ engine plumbing
    def _get_engine(self):
        from .engine import HipEngine
        if self._engine is None:
            object.__setattr__(self, '_engine', HipEngine(self))
        return self._engine

    def _get_train_engine(self):
        from .train import TrainEngine
        if getattr(self, '_train_engine', None) is None:
            object.__setattr__(self, '_train_engine', TrainEngine(self, lib=self._get_engine().lib))
        return self._train_engine

    def repack(self):
        """Tell the device engine that the weights were edited in a way version counters cannot see (writes through
        `.data`, as `rescale_module` and many EMA helpers do): the next forward repacks them."""
        if self._engine is not None:
            self._engine.invalidate()
        if getattr(self, '_train_engine', None) is not None:
            self._train_engine.invalidate()

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_engine'] = None
        st.pop('_train_engine', None)
        st.pop('_grad_sink', None)                               # (a weak reference to the optimizer: FlatAdam re-registers itself)
        return st

    def _spec(self, x, scale=False):
        """STFT of `x` [B, C, L] -> complex64 [B, C, nfft/2, T] (aero.py:409-421)."""
        return self._get_engine().spec(x, scale=scale)

    def _ispec(self, z):
        """iSTFT of complex64 [B, C, nfft/2, T] -> [B, C, hop_out*(T-1)] (aero.py:423-428)."""
        return self._get_engine().ispec(z)

    def forward(self, mix, return_spec=False, return_lr_spec=False):
        """aero.py:446-523.  `mix` [B, in_channels, L] float32 on the MI355X device."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # autograd through forward (solver.py:296-305, 602-605): the layer-by-layer training engine keeps what the HIP backward
            # pass needs (aero_amd/train.py).  The waveform output carries the gradient; the spectrogram outputs are detached views
            # (the reference's losses act on the waveform: solver.py:560-584).
            from .train import AeroFunction
            # (a partially frozen generator -- `requires_grad_(False)` on some parameters, as fine-tuning recipes do -- runs the same backward
            # pass: every gradient is formed, autograd keeps those of the parameters that ask for one; round 6)
            names, params = zip(*self.named_parameters())
            x, spec_r, lr_spec = AeroFunction.apply(self._get_train_engine(), names, mix, *params)
            spec = torch.view_as_complex(spec_r).view(mix.shape[0], 1, spec_r.shape[1], spec_r.shape[2])
            if return_spec:
                return (x, spec, lr_spec) if return_lr_spec else (x, spec)
            return x
        x, spec, lr_spec = self._get_engine().forward(mix, want_spec=return_spec, want_lr_spec=return_lr_spec,
                                                      train=self.training)
        if return_spec:
            return (x, spec, lr_spec) if return_lr_spec else (x, spec)
        return x
