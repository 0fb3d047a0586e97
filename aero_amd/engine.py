"""HipEngine -- runs Aero.forward (reference aero.py:446-523) through the gfx950 kernels.

Host plumbing only: it packs weights once (aero_amd.pack), allocates device buffers with the
PyTorch caching allocator and issues the C-ABI calls of include/aero_hip.h on the current HIP
stream.  No arithmetic of the path is done with ATen ops here, and there is no CPU fallback:
inputs must live on the MI355X (`cuda`) device.

Internal activation layout: fp16 channels-last [B, F, T, C] (see aero_amd/csrc/aero_common.h).
The complex spectrograms handed back to the caller are exactly the reference's complex64
[B, 1, nfft/2, T] tensors (same memory as the fp32 [B, F, T, 2] buffers the kernels write).
"""
import ctypes as C
import math
import os

import torch

from . import _lib, pack
from ._lib import ACT_GELU, ACT_GLU, ACT_NONE, ACT_RELU, ACT_SNAKE


def _ptr(t):
    return None if t is None else t.data_ptr()


def _strides4(t):
    """element strides (b, f, t) of a channels-last [B,F,T,C] tensor (channel stride must be 1)."""
    assert t.dim() == 4 and (t.shape[3] == 1 or t.stride(3) == 1), (t.shape, t.stride())
    return t.stride(0), t.stride(1), t.stride(2)


_SIDE_STREAMS = {}


def side_streams(device, n):
    """The first n of ONE per-device list of HIP streams shared by everything in the package that runs work next to the caller's stream
    (the engine's half-batch streams, BatchPipeline's ring).  The ROCm runtime multiplexes streams onto a few hardware queues (4 by
    default, GPU_MAX_HW_QUEUES): every extra stream object created -- even an idle one -- shifts which streams share a queue, and two
    busy streams on one queue run one after the other.  Measured on the MI355X: three batches in flight 9.3 ms per batch with only
    these streams alive, 9.6-9.9 ms once two more (idle) streams had been created before them."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:             # 'cuda', 'cuda:0', torch.device('cuda', 0): ONE key, hence one list, per device
        dev = torch.device('cuda', torch.cuda.current_device())
    lst = _SIDE_STREAMS.setdefault((dev.type, dev.index), [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:n]


_SPECIAL_STREAMS = {}


def special_stream(device, priority=0, cu_range=None, tag=0):
    """A HIP stream with a dispatch priority (< 0 = ahead of default-priority streams) or restricted to the CUs [lo, hi) of the runtime's
    CU enumeration (round-robin over the XCDs: [0, 8 n) = n CUs of each of the 8 XCDs), created through the C ABI (aero_stream_create)
    and wrapped as a torch ExternalStream.  One object per (device, priority, cu_range, tag), kept for the life of the process."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    key = (dev.index, int(priority), None if cu_range is None else tuple(cu_range), tag)
    st = _SPECIAL_STREAMS.get(key)
    if st is None:
        lib = _lib.load()
        handle = C.c_void_p()
        with torch.cuda.device(dev):
            if cu_range is None:
                lib.call('aero_stream_create', int(priority), None, 0, C.byref(handle))
            else:
                lo, hi = cu_range
                ncu = torch.cuda.get_device_properties(dev).multi_processor_count
                assert 0 <= lo < hi <= ncu, (cu_range, ncu)
                words = (C.c_uint32 * ((ncu + 31) // 32))()
                for i in range(lo, hi):
                    words[i // 32] |= 1 << (i % 32)
                lib.call('aero_stream_create', 0, words, len(words), C.byref(handle))
        st = _SPECIAL_STREAMS[key] = torch.cuda.ExternalStream(handle.value, device=dev)
    return st


class Ops:
    """Tensor-level wrappers over the C ABI (used by the engine and by the op-level tests)."""

    def __init__(self, lib):
        self.lib = lib
        self.prof = None          # bench.py sets this to a list to collect per-launch HIP-event timings
        self.dft_stft = os.environ.get('AERO_STFT_DFT', '1') != '0'      # short-window STFT as a GEMM (k_stft.h)
        self.fused_stft = os.environ.get('AERO_STFT_FUSED', '1') != '0'  # ... run twice (sums, then normalised fp16) instead of stft + spec_normalize
        self._dft_tables = {}
        self.prof_shapes = None   # (tools/launch_table.py) one short shape note per profiled launch
        self.tag = ''             # engine-set label of the launches being issued ('stack' = Conv2d / ConvTranspose2d of the U-Net)
        self._shape_note = ''
        self._arena, self._cur = {}, None

    @staticmethod
    def stream(t):
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def _call(self, fn, kernel, flops, nbytes, *args):
        """One C-ABI call = one kernel launch on the current stream; optionally bracketed by HIP events
        (recorded on that same stream) for the roofline numbers of bench.py."""
        if self.prof is None:
            self._shape_note = ''
            self.lib.call(fn, *args)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.lib.call(fn, *args)
        e1.record()
        name = self.lib.cdll.aero_last_kernel_name().decode() or kernel      # the instantiation as rocprofv3 names it
        self.prof.append((name, flops, nbytes, e0, e1, self.tag))
        if self.prof_shapes is not None:
            self.prof_shapes.append(self._shape_note)
        self._shape_note = ''

    # -- K1/K2/K15 ---------------------------------------------------------------------------
    def stft(self, x, L, Lp, n_fft, hop, window, n_bins, stats=None, sig_per_item=1, win_len=None):
        """win_len: length of the centred window inside `window` [n_fft]; short windows (<= 128 samples, hop % 8 == 0, hop <= 16,
        Nyquist dropped) take the DFT-as-GEMM kernel (aero_stft_dft_fwd) with a table cached per (window, n_fft)."""
        nsig = x.shape[0]
        T = 1 + Lp // hop
        spec = torch.empty(nsig, n_bins, T, 2, dtype=torch.float32, device=x.device)
        if self._dft_applies(nsig, Lp, n_fft, hop, n_bins, win_len):
            win_off = (n_fft - win_len) // 2
            table = self._dft_table(x, window, n_fft, win_len)
            self._call('aero_stft_dft_fwd', 'aero_stft_dft_kernel', 2.0 * nsig * T * n_fft * 128, nsig * (L * 4 + n_bins * T * 8),
                       _ptr(x), nsig, L, Lp, n_fft, hop, win_off, _ptr(table), _ptr(spec), T, _ptr(stats), sig_per_item, self.stream(x))
            return spec
        self._call('aero_stft_fwd', 'aero_stft_kernel', 0, nsig * (L * 4 + n_bins * T * 8),
                   _ptr(x), nsig, L, Lp, n_fft, hop, _ptr(window), n_bins, _ptr(spec), T,
                   _ptr(stats), sig_per_item, self.stream(x))
        return spec

    def _dft_applies(self, nsig, Lp, n_fft, hop, n_bins, win_len):
        return (self.dft_stft and win_len is not None and win_len <= 128 and n_bins == n_fft // 2 and n_fft % 256 == 0 and n_fft <= 4096
                and hop % 8 == 0 and hop <= 16 and nsig <= 65535 and Lp > n_fft // 2)

    def _dft_table(self, x, window, n_fft, win_len):
        win_off = (n_fft - win_len) // 2
        key = (window.data_ptr(), n_fft, win_len, str(x.device))
        table = self._dft_tables.get(key)
        if table is None:
            nbytes = int(self.lib.cdll.aero_stft_dft_table_bytes(n_fft))
            table = torch.empty(nbytes // 2, dtype=torch.float16, device=x.device)
            self.lib.call('aero_stft_dft_table', _ptr(window), n_fft, win_off, _ptr(table), self.stream(x))
            if len(self._dft_tables) >= 8:
                self._dft_tables.pop(next(iter(self._dft_tables)))
            self._dft_tables[key] = (table, window)                    # (the window tensor is kept alive with its table)
            return table
        return table[0]

    def stft_normalized(self, x, L, Lp, n_fft, hop, window, n_bins, stats, sig_per_item, win_len):
        """STFT + per-item normalisation in one call (aero_stft_dft_norm_fwd: the DFT-as-GEMM run twice, sums then normalised fp16 store; no
        fp32 spectrogram in HBM) -> (xn fp16 [nsig, n_bins, T, 2], mean_std fp32 [nitems, 2]), or None when the geometry is not the
        short-window one that kernel takes (the caller then runs stft + spec_normalize)."""
        nsig = x.shape[0]
        if not self.fused_stft or not self._dft_applies(nsig, Lp, n_fft, hop, n_bins, win_len):
            return None
        T = 1 + Lp // hop
        table = self._dft_table(x, window, n_fft, win_len)
        xn = torch.empty(nsig, n_bins, T, 2, dtype=torch.float16, device=x.device)
        mean_std = torch.empty(nsig // sig_per_item, 2, dtype=torch.float32, device=x.device)
        # bytes: what the two launches move (the signal twice, the fp16 result once); FLOPs: the GEMM twice
        self._call('aero_stft_dft_norm_fwd', 'aero_stft_dft_kernel', 2 * 2.0 * nsig * T * n_fft * 128, nsig * (2 * L * 4 + n_bins * T * 4),
                   _ptr(x), nsig, L, Lp, n_fft, hop, (n_fft - win_len) // 2, _ptr(table), T, _ptr(stats), sig_per_item, _ptr(xn), _ptr(mean_std),
                   self.stream(x))
        return xn, mean_std

    def spec_normalize(self, spec, nitems, stats):
        n_per = spec.numel() // nitems
        xn = torch.empty(spec.shape, dtype=torch.float16, device=spec.device)
        mean_std = torch.empty(nitems, 2, dtype=torch.float32, device=spec.device)
        self._call('aero_spec_normalize', 'aero_spec_normalize_kernel', 0, spec.numel() * 6,
                   _ptr(spec), nitems, n_per, _ptr(stats), _ptr(xn), _ptr(mean_std), self.stream(spec))
        return xn, mean_std

    def istft(self, spec, n_fft, hop, window, inv_env, Lout):
        """spec fp32 [nsig, F, T, 2]; or a PITCHED buffer [nsig, F, pitch, 2] marked by `spec._aero_pitched = (T, t_off)` (the layout
        convtr_tail_finish writes for the iSTFT kernel: whole cache lines per 16-frame run, include/aero_hip.h aero_istft_pitch)"""
        nsig, F, T, _ = spec.shape
        y = torch.empty(nsig, Lout, dtype=torch.float32, device=spec.device)
        pitched = getattr(spec, '_aero_pitched', None)
        if pitched is not None:
            pitch, (T, toff) = T, pitched
            self._call('aero_istft_pitched_fwd', 'aero_istft_kernel', 0, nsig * (F * T * 8 + Lout * 4),
                       _ptr(spec), nsig, F, T, pitch, toff, n_fft, hop, _ptr(window), _ptr(inv_env), _ptr(y), Lout, self.stream(spec))
            return y
        self._call('aero_istft_fwd', 'aero_istft_kernel', 0, nsig * (F * T * 8 + Lout * 4),
                   _ptr(spec), nsig, F, T, n_fft, hop, _ptr(window), _ptr(inv_env), _ptr(y), Lout, self.stream(spec))
        return y

    def istft_pitch(self, n_fft, hop, T):
        """(pitch, t_off) of the spectrogram layout the iSTFT kernel reads with whole cache lines; (T, 0): the plain layout only"""
        pitch, toff = C.c_int32(0), C.c_int32(0)
        self.lib.call('aero_istft_pitch', n_fft, hop, T, C.byref(pitch), C.byref(toff))
        return pitch.value, toff.value

    # -- convolution family --------------------------------------------------------------------
    def conv(self, spec, src0, src1, B, Fin, Fout, T, dst=None, dst_f32=False, dst_f_off=0, dst_F=None, res=None,
             post_add=None, batch_scale=None, batch_shift=None, act=None, dst_strides=None, src0_strides=None,
             stat=None, scatter=None, tap_split=1, tail=None):
        """src0/src1: channels-last [B,Fin,T,C] tensors (src0 may be None = zeros).  Returns dst.
        tap_split = S > 1 (aero_hip.h "tap split"): S x the blocks, partial sums into an fp32 accumulator, then aero_split_finish.
        tail = fp16 [16][C] image (pack.convtr_tail_image): the fused transposed-conv tail of aero_hip.h -- the activation is not stored,
        the call returns the two fp32 tap-product tensors (lo, hi) [B, Fout, T, 8] for convtr_tail_finish."""
        dev = spec.weight.device
        act = spec.act if act is None else act
        split_acc = None
        if tap_split > 1:
            assert dst is None and stat is None and res is None and not dst_f32 and act != ACT_GLU
            split_acc = torch.empty(tap_split, B, Fout, T, spec.M, dtype=torch.float32, device=dev)
        Mout = spec.M // 2 if act == ACT_GLU else spec.M
        dst_F = Fout if dst_F is None else dst_F
        no_store = (stat is not None and stat['mode'] == 2) or tail is not None
        tail_lo = tail_hi = None
        if tail is not None:
            assert dst is None and stat is None and res is None and act == ACT_GLU and tap_split == 1
            tail_lo = torch.empty(B, Fout, T, 8, dtype=torch.float32, device=dev)
            tail_hi = torch.empty(B, Fout, T, 8, dtype=torch.float32, device=dev)
        if dst is None and not no_store:
            dst = torch.empty(B, dst_F, T, Mout, dtype=torch.float32 if dst_f32 else torch.float16, device=dev)
        d = _lib.ConvDesc()
        d.src0 = _ptr(src0)
        if src0 is not None:
            d.s0_b, d.s0_f, d.s0_t = src0_strides if src0_strides else _strides4(src0)
        d.C0 = spec.C0
        d.src1 = _ptr(src1)
        if src1 is not None:
            d.s1_b, d.s1_f, d.s1_t = _strides4(src1)
        d.C1 = spec.C1
        d.weight = _ptr(spec.weight)
        d.weight_tiled, d.tiled_bm = _ptr(spec.weight_tiled), spec.tiled_bm
        d.bias = _ptr(spec.bias)
        d.dst = _ptr(dst)
        if dst is not None:
            d.d_b, d.d_f, d.d_t = dst_strides if dst_strides else _strides4(dst)
        d.dst_f32 = int(dst is not None and dst.dtype == torch.float32)
        if stat is not None:
            d.stats, d.stat_count = _ptr(stat['stats']), float(stat.get('count', 0.0))
            d.stat_mode, d.stat_G, d.stat_per_row, d.stat_eps = stat['mode'], stat['G'], int(stat['per_row']), stat.get('eps', 1e-5)
            d.gamma, d.beta, d.layer_scale = _ptr(stat.get('gamma')), _ptr(stat.get('beta')), _ptr(stat.get('layer_scale'))
        d.dst_f_off, d.dst_F = dst_f_off, dst_F
        if scatter is not None:                     # (scatter_M, stride, first kept row, kept rows): aero_hip.h "row scatter"
            d.scatter_M, d.scatter_stride, d.scatter_off, d.scatter_F = scatter
        d.B, d.Fin, d.Fout, d.T, d.M = B, Fin, Fout, T, spec.M
        d.transposed, d.fstride = spec.transposed, spec.fstride
        d.ntaps = len(spec.df)
        for i, (a, b_) in enumerate(zip(spec.df, spec.dt)):
            d.df[i], d.dt[i] = a, b_
        d.act = act
        d.res = _ptr(res)
        if res is not None:
            d.r_b, d.r_f, d.r_t = _strides4(res)
        d.post_add = _ptr(post_add)
        d.batch_scale, d.batch_shift = _ptr(batch_scale), _ptr(batch_shift)
        if split_acc is not None:
            d.tap_split, d.split_acc = tap_split, _ptr(split_acc)
        if tail is not None:
            d.tail_w, d.tail_lo, d.tail_hi, d.tail_cp = _ptr(tail), _ptr(tail_lo), _ptr(tail_hi), tail.shape[1]
        ref_t = dst if dst is not None else spec.weight
        if self.prof is None:
            self.lib.call('aero_conv_fwd', C.byref(d), self.stream(ref_t))
        else:
            buf = C.create_string_buffer(128)
            self.lib.check(self.lib.cdll.aero_conv_kernel_name(C.byref(d), buf, 128), 'aero_conv_kernel_name')
            kname = buf.value.decode()
            self._shape_note = (f'conv M={spec.M} C={spec.C0}+{spec.C1}{"(null0)" if src0 is None and spec.C0 else ""} taps={len(spec.df)} '
                                f'F={Fin}->{Fout} tr={spec.transposed} act={act} res={int(res is not None)} stat={stat["mode"] if stat else 0}')
            pos = B * dst_F * T
            cin_exec = spec.C1 + (spec.C0 if src0 is not None else 0)
            flops = 2.0 * pos * spec.M * len(spec.df) * cin_exec       # executed (NULL source skipped)
            nbytes = pos * (Mout * (dst.element_size() if dst is not None else 0)) + B * Fin * T * cin_exec * 2 + (pos * 64 if tail is not None else 0)
            self._call('aero_conv_fwd', kname, flops, nbytes, C.byref(d), self.stream(ref_t))
        if split_acc is not None:
            self._call('aero_split_finish', 'aero_split_finish_kernel', 0.0, split_acc.numel() * 4 + dst.numel() * 2, _ptr(split_acc), tap_split,
                       _ptr(spec.bias), act, _ptr(dst), B * Fout * T, spec.M, self.stream(dst))
        if tail is not None:
            return tail_lo, tail_hi
        return dst

    def convtr_tail_finish(self, lo, hi, bias, scale, shift, dst_F, pad, cin, pitched=None):
        """second half of the fused last layer (aero_convtr_tail_finish): (lo, hi) [B, Fin, T, 8] -> fp32 [B, dst_F, T, 2];
        pitched = (pitch, t_off): -> [B, dst_F, pitch, 2] with time step t in column t_off + t, marked `_aero_pitched` for Ops.istft"""
        B, Fin, T, _ = lo.shape
        self._shape_note = f'convtr tail finish F={Fin}->{dst_F}'
        if pitched is not None and pitched != (T, 0):
            pitch, toff = pitched
            dst = torch.empty(B, dst_F, pitch, 2, dtype=torch.float32, device=lo.device)
            self._call('aero_convtr_tail_finish_pitched', 'aero_convtr_tail_finish_kernel', 2.0 * B * Fin * T * 16 * cin, lo.numel() * 8 + B * dst_F * T * 8,
                       _ptr(lo), _ptr(hi), _ptr(bias), _ptr(scale), _ptr(shift), _ptr(dst), B, Fin, T, dst_F, pad, pitch, toff, self.stream(lo))
            dst._aero_pitched = (T, toff)
            return dst
        dst = torch.empty(B, dst_F, T, 2, dtype=torch.float32, device=lo.device)
        self._call('aero_convtr_tail_finish', 'aero_convtr_tail_finish_kernel', 2.0 * B * Fin * T * 16 * cin, lo.numel() * 8 + dst.numel() * 4,
                   _ptr(lo), _ptr(hi), _ptr(bias), _ptr(scale), _ptr(shift), _ptr(dst), B, Fin, T, dst_F, pad, self.stream(lo))
        return dst

    def pw(self, spec, x, B, F, T, res=None, post_add=None, stats=None, count=None, gamma=None, beta=None, layer_scale=None,
           eps=1e-5, tag='aero_pw_kernel', x1=None):
        """streaming pointwise conv (aero_pw_fwd, k_pw.h): x fp16 [B,F,T,C] (channels-last view) -> [B,F,T,Mout]; optional GroupNorm
        from per-row sums `stats` [(B*F), 2] (count = elements per row), activation of the spec, LayerScale, residual, frequency-embedding row"""
        Mout = spec.M // 2 if spec.act == ACT_GLU else spec.M
        dst = torch.empty(B, F, T, Mout, dtype=torch.float16, device=x.device)
        d = _lib.PwDesc()
        d.x = _ptr(x)
        d.x_b, d.x_f, d.x_t = _strides4(x)
        d.C = spec.C
        d.wimg, d.bias = _ptr(spec.wimg), _ptr(spec.bias)
        d.stats = _ptr(stats)
        d.stat_count = float(count) if stats is not None else 0.0
        d.stat_eps = eps
        d.gamma, d.beta, d.layer_scale, d.post_add = _ptr(gamma), _ptr(beta), _ptr(layer_scale), _ptr(post_add)
        d.res = _ptr(res)
        if res is not None:
            d.r_b, d.r_f, d.r_t = _strides4(res)
        d.dst = _ptr(dst)
        d.d_b, d.d_f, d.d_t = _strides4(dst)
        d.B, d.F, d.T, d.M, d.act = B, F, T, spec.M, spec.act
        if x1 is not None:                                        # cat([x, x1], channel) in front of the conv
            d.x1 = _ptr(x1)
            d.x1_b, d.x1_f, d.x1_t = _strides4(x1)
            d.C0 = x.shape[-1]
        self._shape_note = f'pw M={spec.M} C={spec.C} F={F} act={spec.act} res={int(res is not None)} norm={int(stats is not None)}'
        nb = x.numel() // x.shape[-1] * (2 * spec.C + 2 * Mout * (2 if res is not None else 1))          # (spec.C counts both sources)
        self._call('aero_pw_fwd', tag, 2.0 * B * F * T * spec.M * spec.C, nb, C.byref(d), self.stream(x))
        return dst

    def squeeze(self, spec, x, B, F, T, dst, rp):
        """the FTB's channel squeeze into the Conv1d image dst [B, T, F*rp] (aero_squeeze_fwd, k_pw.h)"""
        sb, sf, st = _strides4(x)
        self._shape_note = f'squeeze M={spec.M} C={spec.C} F={F}'
        self._call('aero_squeeze_fwd', 'aero_squeeze_kernel', 2.0 * B * F * T * spec.M * spec.C, B * F * T * 2 * (spec.C + spec.M),
                   _ptr(x), sb, sf, st, _ptr(spec.wimg), _ptr(spec.bias), _ptr(dst), B, F, T, spec.C, spec.M, rp, spec.act, self.stream(x))
        return dst

    def begin_step(self, device):
        """Zero the statistics arena of the current stream once: the ~25 GroupNorm accumulators of a forward pass are
        slices of it instead of 25 separate torch.zeros fill launches."""
        key = (str(device), self.stream(torch.empty(0, device=device)))
        ar = self._arena.get(key)
        if ar is None or ar[2] > ar[0].numel():                # first use, or the last pass needed more than we had
            n = max(1 << 16, 2 * (ar[2] if ar else 0))
            ar = [torch.zeros(n, dtype=torch.float64, device=device), 0, 0]
            self._arena[key] = ar
        else:
            ar[0].zero_()
        ar[1] = ar[2] = 0
        self._cur = ar

    _bwd = None           # [fp32 buffer, used, demanded]: the zero-filled scratch of the backward pass that is running (aero_amd/train.py)

    def zeros32(self, n, device, dtype=torch.float32):
        """n zero-initialised 4-byte words for a kernel that accumulates into them: a slice of the running backward pass's ONE zero-filled
        buffer (a single fill launch per backward instead of one per GroupNorm / rescale / weight-gradient call: 55 at the config-5 shape),
        torch.zeros outside a backward pass or when the buffer is exhausted (it is sized from the previous step's demand)."""
        ar = self._bwd
        if ar is not None and ar[0].device == torch.device(device):
            m = (n + 3) // 4 * 4                                 # 16-byte aligned slices
            ar[2] += m
            if ar[1] + m <= ar[0].numel():
                out = ar[0][ar[1]:ar[1] + n]
                ar[1] += m
                return out if dtype == torch.float32 else out.view(dtype)
        return torch.zeros(n, dtype=dtype, device=device)

    def new_stats(self, B, F, G, per_row, device):
        n = (B * F if per_row == 1 else (1 if per_row == 2 else B)) * G * 2
        ar = self._cur
        if ar is not None and ar[0].device == torch.device(device):
            ar[2] += n
            if ar[1] + n <= ar[0].numel():
                out = ar[0][ar[1]:ar[1] + n].view(-1, 2)
                ar[1] += n
                return out
        return torch.zeros(n // 2, 2, dtype=torch.float64, device=device)

    @staticmethod
    def can_fuse_stats(M, G):
        """the conv epilogue can accumulate GroupNorm statistics when a group is 16-row aligned (or there is one group)"""
        return M % 8 == 0 and (G == 1 or (M % G == 0 and (M // G) % 16 == 0))

    # -- GroupNorm + activation ----------------------------------------------------------------
    def norm_act(self, x, G, per_row, gamma, beta, act, snake_a=None, layer_scale=None, res=None, normalize=True,
                 f_lo=0, f_cnt=None, eps=1e-5, dst=None, stats=None, dst_strides=None, stat_count=None):
        """x [B,F,T,C] fp16.  Statistics over all F rows; output only rows [f_lo, f_lo+f_cnt).
        per_row: 0 = per (b, group), 1 = per (b, f) row and group, 2 = per group over the whole batch (BatchNorm, training)."""
        B, F, T, Cc = x.shape
        d = _lib.NormDesc()
        d.src = _ptr(x)
        d.s_b, d.s_f, d.s_t = _strides4(x)
        d.B, d.F, d.T, d.C, d.G, d.per_row, d.eps = B, F, T, Cc, G, int(per_row), eps
        if normalize:
            d.stat_count = float((1 if per_row == 1 else (B * F if per_row == 2 else F)) * T * (Cc // G))
            if stat_count is not None:              # elements per (item, group) when some channels are zero padding
                d.stat_count = float(stat_count)
            if stats is None:                       # not already accumulated by the producing conv's epilogue
                stats = self.new_stats(B, F, G, per_row, x.device)
                d.stats = _ptr(stats)
                self._call('aero_norm_stats', 'aero_norm_stats_kernel', 0, x.numel() * 2, C.byref(d), self.stream(x))
            d.stats = _ptr(stats)
        f_cnt = F if f_cnt is None else f_cnt
        assert not (per_row and (f_lo or f_cnt != F))
        Cout = Cc // 2 if act == ACT_GLU else Cc
        out = dst if dst is not None else torch.empty(B, f_cnt, T, Cout, dtype=torch.float16, device=x.device)
        assert dst_strides is not None or out.shape == (B, f_cnt, T, Cout)
        d.src = x.data_ptr() + f_lo * x.stride(1) * x.element_size()
        d.F = f_cnt
        d.gamma, d.beta = _ptr(gamma), _ptr(beta)
        d.act = act
        d.snake_a, d.layer_scale = _ptr(snake_a), _ptr(layer_scale)
        d.res = _ptr(res)
        if res is not None:
            d.r_b, d.r_f, d.r_t = _strides4(res)
        d.dst = _ptr(out)
        d.d_b, d.d_f, d.d_t = dst_strides if dst_strides is not None else _strides4(out)
        self._last_stats = stats
        self._call('aero_norm_apply', 'aero_norm_apply_kernel', 0,
                   B * f_cnt * T * (Cc + Cout + (Cout if res is not None else 0)) * 2, C.byref(d), self.stream(x))
        return out

    def gram_stats(self, x, tables, stats):
        """GroupNorm(1) sums of a pointwise conv's output from the Gram matrix of its input x [B,F,T,C] (aero_gram_stats)."""
        B, F, T, Cc = x.shape
        d = _lib.GramDesc()
        d.x = _ptr(x)
        d.s_b, d.s_f, d.s_t = _strides4(x)
        d.B, d.F, d.T, d.C = B, F, T, Cc
        d.G, d.g1, d.stats = _ptr(tables[0]), _ptr(tables[1]), _ptr(stats)
        self._call('aero_gram_stats', 'aero_gram_stats_kernel', 2.0 * B * F * T * (Cc + 1) ** 2, x.numel() * 2,
                   C.byref(d), self.stream(x))
        return stats

    # -- LSTM / attention / FTB ----------------------------------------------------------------
    def lstm(self, xproj, xbias, whh, H, nseq, W, in_mode, out_mode, nframes, S, T, out, x=None, fused=None, save=None, frame_major=0):
        """one bidirectional layer.  Either (xproj, xbias) = precomputed input projection, or (x, fused=(wih, bias, in_ch)):
        the projection is computed inside the recurrent kernel from the raw input rows x [npos, in_ch].
        save = (gates fp16, c fp32) buffers (train_ops.lstm_save_buffers): the training-mode forward keeps what BPTT needs."""
        d = _lib.LstmDesc()
        if save is not None:
            d.save_gates, d.save_c = _ptr(save[0]), _ptr(save[1])
        d.xproj, d.xbias, d.whh, d.out = _ptr(xproj), _ptr(xbias), _ptr(whh), _ptr(out)
        d.H, d.nseq, d.W, d.in_mode, d.out_mode, d.nframes, d.S, d.T = H, nseq, W, in_mode, out_mode, nframes, S, T
        d.frame_major = int(frame_major)
        flops, nbytes = 2.0 * nseq * W * 2 * 4 * H * H, out.numel() * 2
        if fused is not None:
            wih, bias, in_ch = fused
            d.x, d.wih, d.bias, d.in_ch, d.x_pitch = _ptr(x), _ptr(wih), _ptr(bias), in_ch, x.shape[-1]
            flops += 2.0 * nseq * W * 2 * 4 * H * in_ch
            nbytes += nseq * W * in_ch * 2
        else:
            nbytes += xproj.numel() * 2
        self._call('aero_lstm_fwd', 'aero_lstm_kernel', flops, nbytes, C.byref(d), self.stream(out))
        return out

    def localstate(self, qkvd, R, T, Cc, heads, ndecay):
        out = torch.empty(R, T, Cc, dtype=torch.float16, device=qkvd.device)
        d = _lib.AttnDesc()
        d.qkvd, d.ld, d.out = _ptr(qkvd), qkvd.shape[-1], _ptr(out)
        d.R, d.T, d.C, d.heads, d.ndecay = R, T, Cc, heads, ndecay
        self._call('aero_localstate_fwd', 'aero_attn_kernel', 4.0 * R * T * T * Cc, qkvd.numel() * 2 + out.numel() * 2,
                   C.byref(d), self.stream(out))
        return out

    def freqfc(self, x, w, gate):
        B, F, T, Cc = x.shape
        assert x.is_contiguous() and gate.is_contiguous()
        out = torch.empty_like(x)
        d = _lib.FreqFcDesc()
        d.x, d.w, d.gate, d.dst = _ptr(x), _ptr(w), _ptr(gate), _ptr(out)
        d.B, d.F, d.T, d.C = B, F, T, Cc
        self._call('aero_freqfc_fwd', 'aero_freqfc_kernel', 2.0 * B * F * F * T * Cc, x.numel() * 4,
                   C.byref(d), self.stream(out))
        return out


def _ftb_first(self, xn, u, gate, P):
    B, F, T, _ = xn.shape
    Cc = P['C']
    out = torch.empty(B, F, T, Cc, dtype=torch.float16, device=xn.device)
    d = _lib.FtbFirstDesc()
    d.xn, d.u, d.gate, d.w2a = _ptr(xn), _ptr(u), _ptr(gate), _ptr(P['w2a'])
    d.p0, d.p1, d.pb, d.rs = _ptr(P['p0']), _ptr(P['p1']), _ptr(P['pb']), _ptr(P['rs'])
    d.a_re, d.a_im, d.bias = _ptr(P['a_re']), _ptr(P['a_im']), _ptr(P['bias'])
    d.dst = _ptr(out)
    d.B, d.F, d.T, d.C = B, F, T, Cc
    self._call('aero_ftb_first_fwd', 'aero_ftb_first_kernel', 2.0 * B * F * T * Cc * Cc, B * F * T * (8 + 2 * Cc),
               C.byref(d), self.stream(out))
    return out


Ops.ftb_first = _ftb_first


def _enc0(self, xn, u, g, P, conv, Fo, stride, pad, act):
    """encoder 0 fused (aero_enc0_fwd): FTB (collapsed, six-term bilinear form) + the strided frequency conv + activation"""
    B, F, T, _ = xn.shape
    Cc, M = P['C'], conv.M
    out = torch.empty(B, Fo, T, M, dtype=torch.float16, device=xn.device)
    d = _lib.Enc0Desc()
    d.xn, d.u, d.g = _ptr(xn), _ptr(u), _ptr(g)
    d.rs, d.a_re, d.a_im, d.bias_f = _ptr(P['rs']), _ptr(P['a_re']), _ptr(P['a_im']), _ptr(P['bias'])
    d.wc, d.bias_c, d.dst = _ptr(conv.weight), _ptr(conv.bias), _ptr(out)
    d.B, d.F, d.T, d.C, d.M, d.Fo = B, F, T, Cc, M, Fo
    d.ktaps, d.stride, d.pad, d.act = len(conv.df), stride, pad, act
    flops = 2.0 * B * Fo * T * M * len(conv.df) * Cc
    self._shape_note = f'enc0 C={Cc} M={M} taps={len(conv.df)} F={F}->{Fo}'
    self._call('aero_enc0_fwd', 'aero_enc0_kernel', flops, B * F * T * 8 + B * Fo * T * M * 2 + g.numel() * 2,
               C.byref(d), self.stream(out))
    return out


Ops.enc0 = _enc0


def _dconv_row(self, x, layers, act, F, eps=1e-5):
    """aero_dconv_row_fwd: every layer of a DConv branch (no BLSTM / LocalState) on x [B,F,T,C] in one launch; returns a new tensor"""
    B, Fq, T, Cc = x.shape
    out = torch.empty_like(x)
    d = _lib.DconvDesc()
    d.x, d.y = _ptr(x), _ptr(out)
    d.R, d.T, d.C, d.hidden, d.depth, d.act, d.F, d.eps = B * Fq, T, Cc, layers[0]['hidden'], len(layers), act, F, eps
    flops = 0.0
    for i, L in enumerate(layers):
        l = d.layer[i]
        l.w1, l.w2, l.consts, l.snake_a = _ptr(L['w1']), _ptr(L['w2']), _ptr(L['consts']), _ptr(L.get('snake_a'))
        l.dilation, l.norm1, l.norm2 = L['dilation'], L['norm1'], L['norm2']
        flops += 2.0 * B * Fq * T * L['hidden'] * (3 * Cc + 2 * Cc)
    self._shape_note = f'dconv rows C={Cc} hidden={layers[0]["hidden"]} depth={len(layers)} F={Fq}'
    self._call('aero_dconv_row_fwd', 'aero_dconv_row_kernel', flops, 2 * x.numel() * 2, C.byref(d), self.stream(x))
    return out


def _dconv_row_fits(self, T, Cc, hidden, maxdil):
    return bool(self.lib.cdll.aero_dconv_row_fits(T, Cc, hidden, maxdil))


Ops.dconv_row = _dconv_row
Ops.dconv_row_fits = _dconv_row_fits


def blstm_frames(T, W=200, S=100):
    """frames of `unfold(x, W, S)` whose outputs survive the BLSTM's stitch (models/utils.py:22-35, modules.py:52-62: frame 0 keeps [0, W - S/2),
    a middle frame [S/2, W - S/2), the last one [S/2, W)).  The reference cuts n = ceil(T / S) frames; when T <= (n - 1) S + S/2 the LAST one
    covers [(n - 1) S + S/2, ...) -- entirely beyond T: every step of it is computed (over zero padding) and discarded.  Dropping it makes
    frame n - 2 the last one, whose kept range then reaches T with the very values it had (a frame's recurrence does not depend on the other
    frames): bit-identical output for 1 / n less recurrent work (T = 501: 6 -> 5 frames, T = 1724: 18 -> 17; T = 376 keeps its 4)."""
    n = math.ceil(T / S)
    if n >= 3 and T <= (n - 1) * S + S // 2:
        n -= 1
    return n


def _hann_padded(win_length, n_fft, device):
    w = torch.zeros(n_fft, dtype=torch.float32)
    left = (n_fft - win_length) // 2
    w[left:left + win_length] = torch.hann_window(win_length, periodic=True, dtype=torch.float32)
    return w.to(device)


class _ShapeCache(dict):
    """Engine-side cache of shape-keyed device buffers / tables / captured graphs.  Entries whose key starts with one of
    BOUNDED are evicted least-recently-used beyond MAX_PER_KIND per kind: test.py / evaluate feed one variable-length
    file per forward, and every distinct frame count T would otherwise pin its own padded buffers, envelopes and graphs
    for the life of the model."""
    BOUNDED = ('hidpad', 'ones', 'env', 'graph')
    MAX_PER_KIND = 8

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(key, tuple) and key and key[0] in self.BOUNDED:
            dict.__delitem__(self, key)                       # re-insert: dict order = recency
            dict.__setitem__(self, key, v)
        return v

    def __setitem__(self, key, value):
        dict.__setitem__(self, key, value)
        if isinstance(key, tuple) and key and key[0] in self.BOUNDED:
            same = [k for k in self if isinstance(k, tuple) and k and k[0] == key[0]]
            for k in same[:max(0, len(same) - self.MAX_PER_KIND)]:
                dict.__delitem__(self, k)


class HipEngine:
    use_pw = os.environ.get('AERO_PW', '1') != '0'              # (class default: tests build bare engines with __new__)
    stagger = 0
    prof_streams = False
    stage_hook = None
    lstm_frame_major = os.environ.get('AERO_LSTM_FRAME_MAJOR', '1') != '0'     # (class default: tests build bare engines with __new__)
    _mark_base = None
    _out_pitch = None                   # (pitch, t_off) of the output spectrogram while a forward that hands it straight to the iSTFT runs

    def __init__(self, model, lib=None):
        self.model = model
        self.lib = lib if lib is not None else _lib.load()
        self.ops = Ops(self.lib)
        self._key = None
        self._params, self._epoch = None, 0
        self._tables = _ShapeCache()
        # sub-batches in flight on separate HIP streams: 0 = auto (two from 32 clips up: the latency-bound recurrent launches of one half
        # overlap with the MFMA / bandwidth-bound launches of the other: B = 64: 11.05 -> 10.72 ms, B = 128: 21.1 -> 20.2; neutral at 16,
        # +1..6 % at <= 8 clips, where it stays off)
        self.streams = int(os.environ.get('AERO_STREAMS', '0'))
        self.stagger = int(os.environ.get('AERO_STAGGER', '0'))          # two-stream forward: start of the second half relative to the first (see forward)
        self.prof_streams = False           # bench.py: keep the sub-batch streams while per-launch HIP events are recorded (each on its launch's stream)
        self.use_graph = os.environ.get('AERO_GRAPH', '0') != '0'  # replay the forward as a captured HIP graph (per input shape)
        # GroupNorm fused into conv epilogues (stat_mode 1-3 of aero_conv_fwd, lean epilogue paths).  Measured on MI355X:
        #  * DConv tail as a recompute pair (statistics pass without stores + normalise/GLU/LayerScale/skip pass): the
        #    2C-channel tensor never reaches HBM; -0.3 ms per forward for the layers with vector-aligned operands -> on.
        #  * statistics of the encoder/decoder norms accumulated by the producing conv: removes 0.8 ms of statistics
        #    passes but the statistics instantiations lose the 8-wave tiles -> slower overall (+0.3 ms) -> off.
        self.fuse_dconv_tail = os.environ.get('AERO_FUSE_DCONV', '1') != '0'       # DConv tail as a recompute pair of conv launches (2C-channel tensor never stored)
        self.gram_stats = os.environ.get('AERO_GRAM_STATS', '1') != '0'    # DConv tail statistics from the Gram matrix of conv2's input (k_gram.h)
        # GroupNorm statistics accumulated in the producing conv's epilogue (k_conv.h tiles): '1' always, '0' never, default
        # 'auto' = where the contraction has several taps (there the extra epilogue work costs 3-8 us against a 9-53 us
        # statistics pass; on the pointwise rewrite convs it costs 45-70 us against 33 us: per-launch table in DESIGN.md)
        self.fuse_stats = {'0': False, '1': True}.get(os.environ.get('AERO_FUSE_STATS', 'auto'), 'auto')
        self.ring_stats = os.environ.get('AERO_RING_STATS', '1') != '0'    # GroupNorm statistics in the ring conv kernel's epilogue
        self.fuse_lstm_proj = True         # W_ih x_t inside the recurrent kernel (no 8H-channel pre-activation tensor in HBM)
        self.collapse_first_ftb = True     # encoder-0 FTB on the 2-channel spectrogram (k_ftb.h); False = layer by layer
        self.split_taps = os.environ.get('AERO_TAP_SPLIT', '1') != '0'        # long thin FTB Conv1d: 3 x the blocks, partial sums summed in fixed order
        self.use_pw = os.environ.get('AERO_PW', '1') != '0'                   # streaming pointwise kernel (k_pw.h) for the DConv tails / rewrite + GLU convs
        self.fuse_dconv_row = os.environ.get('AERO_DCONV_ROW', '1') != '0'    # DConv branches without LSTM / attention: one launch, the row stays in LDS (k_dconv.h)
        self.fuse_enc0 = os.environ.get('AERO_FUSE_ENC0', '1') != '0'     # ... and fused with the layer's strided conv (k_enc0.h)
        self.fuse_tail = os.environ.get('AERO_FUSE_TAIL', '1') != '0'     # last decoder layer: the transposed conv inside the rewrite conv's epilogue (k_conv_ring.h)
        self.pitched_out = os.environ.get('AERO_PITCHED_OUT', '1') != '0' # ... its output rows at the cache-line pitch the iSTFT kernel reads (aero_istft_pitch)

    # ------------------------------------------------------------------ weights
    def _weights_key(self, device):
        """(storage address, version counter) of every parameter and buffer.  The tensor list is collected once (the
        state_dict walk costs more than a small forward); in-place updates through autograd-visible ops bump `_version`,
        a re-assigned parameter changes the address.  Writes through `.data` (e.g. `p.data.mul_()`) bump NEITHER: call
        `invalidate()` after such an update."""
        if self._params is None:
            self._params = list(self.model.state_dict(keep_vars=True).values())
        return (str(device), self._epoch) + tuple((p.data_ptr(), p._version) for p in self._params)

    def invalidate(self):
        """Forget the packed weights (and captured graphs): the next forward repacks from the module's tensors.  Needed
        only after weight edits the version counters cannot see (`.data` writes, load through external memory)."""
        self._epoch += 1
        self._params = None

    def _prepare(self, device):
        key = self._weights_key(device)
        if key != self._key:
            self._params = None                         # re-collect: load_state_dict / new Parameter objects
            self._pack(device)
            self._key = key
            for k in [k for k in self._tables if isinstance(k, tuple) and k and k[0] == 'graph']:
                del self._tables[k]                     # captured graphs point at the previous packed weights

    def _pack(self, device):
        m = self.model
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        P = {}
        mk = pack.make_conv_spec
        for i, enc in enumerate(m.encoder):
            p = f'encoder.{i}'
            L = {}
            if enc.is_first:
                w, df, dt = pack.conv2d_taps(sd[f'{p}.pre_conv.weight'], 0, 0)
                L['pre'] = mk(w, sd[f'{p}.pre_conv.bias'], w.shape[-1], 0, df, dt, device)
            if enc.freq_attn:
                q = f'{p}.freq_attn_block'
                Fd, Cc, r = enc.freq_attn_block.input_dim, enc.freq_attn_block.in_channel, enc.freq_attn_block.r_channel
                rp = r if (Fd * r) % 8 == 0 else 8 * ((r + 7) // 8)    # channel pitch per frequency bin in the [B,T,F*rp] image
                w, b = pack.bn_fold(sd[f'{q}.conv1.0.weight'], sd[f'{q}.conv1.0.bias'], sd[f'{q}.conv1.1.weight'],
                                    sd[f'{q}.conv1.1.bias'], sd[f'{q}.conv1.1.running_mean'], sd[f'{q}.conv1.1.running_var'])
                w, df, dt = pack.conv2d_taps(w, 0, 0)
                L['ftb_c1'] = mk(w, b, Cc, 0, df, dt, device, act=ACT_RELU)
                L['ftb_c1_sq'] = pack.make_squeeze_spec(w[0, :, 0, :], b, ACT_RELU, device) if rp <= 8 else None
                w, b = pack.bn_fold(sd[f'{q}.conv1d.0.weight'], sd[f'{q}.conv1d.0.bias'], sd[f'{q}.conv1d.1.weight'],
                                    sd[f'{q}.conv1d.1.bias'], sd[f'{q}.conv1d.1.running_mean'], sd[f'{q}.conv1d.1.running_var'])
                k9 = w.shape[-1]
                # reference input channel = c*F + f (modules.py:309); ours = f*rp + c (rp = r padded to 8)
                w = w.view(Cc, r, Fd, k9).permute(0, 3, 2, 1)              # [M, k, F, r]
                wp = torch.zeros(Cc, k9, Fd, rp)
                wp[..., :r] = w
                L['ftb_c1d'] = mk(wp.reshape(1, Cc, k9, Fd * rp), b, Fd * rp, 0, [0] * k9,
                                  [j - (k9 // 2) for j in range(k9)], device, act=ACT_RELU)
                L['ftb_rp'] = rp
                wfc = sd[f'{q}.freq_fc.weight']
                img = torch.zeros(pack._round_up(Fd, 128), pack._round_up(Fd, 32))
                img[:Fd, :Fd] = wfc
                L['ftb_fc'] = img.to(device=device, dtype=torch.float16).contiguous()
                w, b = pack.bn_fold(sd[f'{q}.conv2.0.weight'], sd[f'{q}.conv2.0.bias'], sd[f'{q}.conv2.1.weight'],
                                    sd[f'{q}.conv2.1.bias'], sd[f'{q}.conv2.1.running_mean'], sd[f'{q}.conv2.1.running_var'])
                w, df, dt = pack.conv2d_taps(w, 0, 0)
                L['ftb_c2'] = mk(w, b, Cc, Cc, df, dt, device, act=ACT_RELU)
                L['ftb_c2_pw'] = pack.make_pw_spec(w[0, :, 0, :], b, ACT_RELU, self.lib, device) if 2 * Cc <= 96 else None
                if enc.is_first and Cc % 8 == 0 and Cc <= 64:
                    # encoder 0: pre_conv feeds the FTB linearly -> collapse onto the 2 input channels (k_ftb.h)
                    Wp = sd[f'{p}.pre_conv.weight'][:, :, 0, 0]                    # [C, 2]
                    bp = sd[f'{p}.pre_conv.bias']
                    w1, b1 = pack.bn_fold(sd[f'{q}.conv1.0.weight'], sd[f'{q}.conv1.0.bias'], sd[f'{q}.conv1.1.weight'],
                                          sd[f'{q}.conv1.1.bias'], sd[f'{q}.conv1.1.running_mean'], sd[f'{q}.conv1.1.running_var'])
                    w1 = w1[:, :, 0, 0]                                            # [r, C]
                    L['ftb0_c1'] = mk((w1 @ Wp)[None, :, None, :], w1 @ bp + b1, Wp.shape[1], 0, [0], [0], device, act=ACT_RELU)
                    w2, b2 = pack.bn_fold(sd[f'{q}.conv2.0.weight'], sd[f'{q}.conv2.0.bias'], sd[f'{q}.conv2.1.weight'],
                                          sd[f'{q}.conv2.1.bias'], sd[f'{q}.conv2.1.running_mean'], sd[f'{q}.conv2.1.running_var'])
                    w2 = w2[:, :, 0, 0]                                            # [C, 2C]: [attention branch | direct]
                    w2a, w2b = w2[:, :Cc], w2[:, Cc:]
                    img = torch.zeros(128, pack._round_up(Cc, 32))
                    img[:Cc, :Cc] = w2a
                    A = w2b @ Wp
                    f32 = lambda t: t.float().to(device).contiguous()             # noqa: E731
                    # G = (w2a * pq) applied to the gate, q = 0..2: the f-independent factors of the attention branch (k_enc0.h)
                    wg = torch.cat([w2a * Wp[:, 0][None, :], w2a * Wp[:, 1][None, :], w2a * bp[None, :]], 0)      # [3C, C]
                    L['ftb0_g'] = mk(wg[None, :, None, :], None, Cc, 0, [0], [0], device)
                    L['ftb0'] = dict(C=Cc, w2a=img.to(device=device, dtype=torch.float16).contiguous(),
                                     p0=f32(Wp[:, 0]), p1=f32(Wp[:, 1]), pb=f32(bp), rs=f32(wfc.sum(1)),
                                     a_re=f32(A[:, 0]), a_im=f32(A[:, 1]), bias=f32(w2b @ bp + b2))
            w, df, dt = pack.conv2d_taps(sd[f'{p}.conv.weight'], enc.pad, 0)
            L['conv'] = mk(w, sd[f'{p}.conv.bias'], w.shape[-1], 0, df, dt, device, fstride=enc.stride,
                           act=ACT_NONE if enc.norm else ACT_GELU)
            if enc.norm:
                L['norm1'] = (sd[f'{p}.norm1.weight'].to(device), sd[f'{p}.norm1.bias'].to(device))
            if enc.dconv is not None:
                L['dconv'] = self._pack_dconv(sd, f'{p}.dconv', enc.dconv, device)
            if enc.rewrite is not None:
                w, df, dt = pack.conv2d_taps(sd[f'{p}.rewrite.weight'], enc.context, enc.context)
                L['rewrite'] = mk(w, sd[f'{p}.rewrite.bias'], w.shape[-1], 0, df, dt, device,
                                  act=ACT_NONE if enc.norm else ACT_GLU)
                if enc.norm:
                    L['norm2'] = (sd[f'{p}.norm2.weight'].to(device), sd[f'{p}.norm2.bias'].to(device))
                    if len(df) == 1 and w.shape[-1] <= 384:      # pointwise rewrite, GroupNorm + GLU behind it: the conv alone on the streaming kernel
                        L['rewrite_pw'] = pack.make_pw_spec(w[0, :, 0, :], sd[f'{p}.rewrite.bias'], ACT_NONE, self.lib, device)
                elif len(df) == 1:                                 # pointwise rewrite + GLU with no norm between: the streaming kernel (k_pw.h)
                    L['rewrite_pw'] = pack.make_pw_spec(w[0, :, 0, :], sd[f'{p}.rewrite.bias'], ACT_GLU, self.lib, device)
            P[p] = L
        for j, dec in enumerate(m.decoder):
            p = f'decoder.{j}'
            L = {}
            half = dec.chin // 2
            if dec.rewrite is not None:
                w, df, dt = pack.conv2d_taps(sd[f'{p}.rewrite.weight'], dec.context, dec.context)
                L['rewrite'] = mk(w, sd[f'{p}.rewrite.bias'], half, half, df, dt, device,
                                  act=ACT_NONE if dec.norm else ACT_GLU)
                if dec.norm:
                    L['norm1'] = (sd[f'{p}.norm1.weight'].to(device), sd[f'{p}.norm1.bias'].to(device))
            w, df, dt = pack.convtr_taps(sd[f'{p}.conv_tr.weight'], dec.stride)
            act = ACT_NONE if (dec.norm or dec.last) else ACT_GELU
            L['conv_tr'] = mk(w, sd[f'{p}.conv_tr.bias'], w.shape[-1], 0, df, dt, device, transposed=1,
                              fstride=dec.stride, act=act)
            if dec.last and not dec.norm and dec.rewrite is not None:
                # fused tail (aero_hip.h): when the rewrite conv runs on the 192-row ring tile and the transposed conv is C -> 2, [8,1] / [4,1]
                timg = pack.convtr_tail_image(sd[f'{p}.conv_tr.weight'], dec.stride, device)
                rw = L['rewrite']
                if (timg is not None and rw.M == 192 and rw.tiled_bm == 192 and len(rw.df) == 9 and timg.shape[1] == 96 and dec.pad == 2
                        and dec.kernel_size == 8):
                    L['tail'] = (timg, sd[f'{p}.conv_tr.bias'].detach().float().to(device).contiguous())
            if not dec.last and os.environ.get('AERO_CONVTR_STACK', '1') != '0':
                # input-side form (all `stride` residue classes from one pass over the source rows), when Cout % 8 == 0
                st = pack.convtr_stacked_spec(sd[f'{p}.conv_tr.weight'], sd[f'{p}.conv_tr.bias'], dec.stride, device, act=act)
                if st is not None:
                    L['conv_tr_stacked'] = st
            if dec.norm:
                L['norm2'] = (sd[f'{p}.norm2.weight'].to(device), sd[f'{p}.norm2.bias'].to(device))
            if dec.dconv is not None:
                raise NotImplementedError('decoder DConv (dconv_mode & 2) is not used by any reference config')
            P[p] = L
        if m.freq_emb is not None:
            emb = sd['freq_emb.embedding.weight'] * m.freq_emb.scale * m.freq_emb_scale      # [F1, C]
            P['freq_emb'] = emb.to(device).contiguous()
        self.P = P

    def _pack_dconv(self, sd, p, dc, device):
        mk = pack.make_conv_spec
        out = []
        for d in range(dc.depth):
            q = f'{p}.layers.{d}'
            dil = 2 ** d if dc.dilate else 1
            L = {}
            w, df, dt = pack.conv1d_taps(sd[f'{q}.conv1.0.weight'], dil, dil * (dc.kernel // 2))
            L['conv1'] = mk(w, sd[f'{q}.conv1.0.bias'], w.shape[-1], 0, df, dt, device)
            L['gn1'] = (sd[f'{q}.conv1.1.weight'].to(device), sd[f'{q}.conv1.1.bias'].to(device)) if dc.norm else None
            if dc.act_func == 'snake':
                L['snake_a'] = sd[f'{q}.act.a'].reshape(-1).to(device).contiguous()
            if dc.lstm:
                H = dc.hidden
                L['lstm'] = [pack.pack_lstm_layer(self.lib, sd, f'{q}.lstm.lstm', l, H, device) for l in range(2)]
                w = sd[f'{q}.lstm.linear.weight']
                L['lstm_lin'] = mk(w[None, :, None, :], sd[f'{q}.lstm.linear.bias'], w.shape[1], 0, [0], [0], device)
                L['lstm_lin_pw'] = pack.make_pw_spec(w, sd[f'{q}.lstm.linear.bias'], ACT_NONE, self.lib, device)
            if dc.time_attn:
                a = f'{q}.time_attn'
                w = torch.cat([sd[f'{a}.{n}.weight'][:, :, 0] for n in ('query', 'key', 'content', 'query_decay')], 0)
                b = torch.cat([sd[f'{a}.{n}.bias'] for n in ('query', 'key', 'content', 'query_decay')], 0)
                L['attn_qkvd'] = mk(w[None, :, None, :], b, w.shape[1], 0, [0], [0], device)
                L['attn_qkvd_pw'] = pack.make_pw_spec(w, b, ACT_NONE, self.lib, device)
                w = sd[f'{a}.proj.weight'][:, :, 0]
                L['attn_proj'] = mk(w[None, :, None, :], sd[f'{a}.proj.bias'], w.shape[1], 0, [0], [0], device)
                L['attn_proj_pw'] = pack.make_pw_spec(w, sd[f'{a}.proj.bias'], ACT_NONE, self.lib, device)
                mod = dc.layers[d]['time_attn']
                L['attn_geom'] = (mod.heads, mod.ndecay)
            w, df, dt = pack.conv1d_taps(sd[f'{q}.conv2.0.weight'], 1, 0)
            L['conv2'] = mk(w, sd[f'{q}.conv2.0.bias'], w.shape[-1], 0, df, dt, device)
            L['gn2'] = (sd[f'{q}.conv2.1.weight'].to(device), sd[f'{q}.conv2.1.bias'].to(device)) if dc.norm else None
            L['scale'] = sd[f'{q}.conv2.3.scale'].to(device).contiguous()
            # recompute pair for the tail (conv2 -> GroupNorm(1) -> GLU -> LayerScale -> +skip): rows GLU-interleaved
            w, df, dt = pack.conv1d_taps(sd[f'{q}.conv2.0.weight'], 1, 0)
            L['conv2_glu'] = mk(w, sd[f'{q}.conv2.0.bias'], w.shape[-1], 0, df, dt, device, act=ACT_GLU)
            L['pw2'] = pack.make_pw_spec(w[0, :, 0, :], sd[f'{q}.conv2.0.bias'], ACT_GLU, self.lib, device) if (dc.lstm or dc.time_attn) and dc.norm else None
            if dc.norm:
                L['gn2_glu'] = (pack.glu_interleave(sd[f'{q}.conv2.1.weight']).to(device).contiguous(),
                                pack.glu_interleave(sd[f'{q}.conv2.1.bias']).to(device).contiguous())
            hid = w.shape[-1]
            if not dc.lstm and not dc.time_attn and dc.kernel == 3 and hid % 4 == 0 and hid <= 32 and w.shape[1] % 16 == 0:
                # whole-branch row kernel (k_dconv.h): conv1 image [HP][3C] with k = tap*C + c, conv2 image [2C][HP] GLU-interleaved
                g = (lambda n: sd[n]) if dc.norm else (lambda n: None)                       # noqa: E731
                L['row'] = pack.dconv_row_layer(sd[f'{q}.conv1.0.weight'], sd[f'{q}.conv1.0.bias'], g(f'{q}.conv1.1.weight'),
                                                g(f'{q}.conv1.1.bias'), sd[f'{q}.conv2.0.weight'][:, :, 0], sd[f'{q}.conv2.0.bias'],
                                                g(f'{q}.conv2.1.weight'), g(f'{q}.conv2.1.bias'), sd[f'{q}.conv2.3.scale'], dil, device)
            if hid + 1 <= 112:                                      # statistics of conv2's output from the Gram matrix of its input
                L['gram'] = pack.gram_tables(w[0, :, 0, :], sd[f'{q}.conv2.0.bias'], device)
            if hid % 8 and not dc.lstm and not dc.time_attn:
                # hidden width not a multiple of 8 (first layer: 12): the activation is kept with a channel pitch of 16 and
                # zero pad channels so that conv2 reads 16-byte aligned rows (direct-to-LDS pipeline, recompute pair)
                hp = (hid + 7) // 8 * 8
                wp = torch.zeros(w.shape[0], w.shape[1], w.shape[2], hp)
                wp[..., :hid] = w
                L['conv2_glu_pad'] = mk(wp, sd[f'{q}.conv2.0.bias'], hp, 0, df, dt, device, act=ACT_GLU)
                L['hid_pad'] = hp
                L['gram_pad'] = pack.gram_tables(wp[0, :, 0, :], sd[f'{q}.conv2.0.bias'], device)
            out.append(L)
        return out

    # ------------------------------------------------------------------ tables
    def _window(self, win_length, device):
        key = ('win', win_length, self.model.nfft, str(device))
        if key not in self._tables:
            self._tables[key] = _hann_padded(win_length, self.model.nfft, device)
        return self._tables[key]

    def _inv_env(self, win_length, hop, T, device):
        key = ('env', win_length, hop, T, self.model.nfft, str(device))
        if key not in self._tables:
            n_fft = self.model.nfft
            w = _hann_padded(win_length, n_fft, 'cpu')
            w2 = (w * w).double()
            env = torch.zeros(n_fft + hop * (T - 1), dtype=torch.float64)
            for t in range(T):
                env[t * hop:t * hop + n_fft] += w2
            inv = torch.where(env > 1e-11, 1.0 / env, torch.zeros_like(env))
            self._tables[key] = inv.float().to(device)
        return self._tables[key]

    def _check_input(self, x):
        if not self.lib.is_emulator and not x.is_cuda:
            raise RuntimeError('aero_amd runs on the MI355X only: move the model and the input to "cuda" '
                               '(there is no CPU path in the product)')
        if x.dtype != torch.float32:
            raise TypeError('expected a float32 waveform')

    # ------------------------------------------------------------------ public pieces
    def spec(self, x, scale=False, stats=None):
        """Aero._spec (aero.py:409-421): [B, C, L] -> complex64 [B, C, nfft/2, T]."""
        m = self.model
        self._check_input(x)
        B, Cc, L = x.shape
        hop, win = m.hop_length, m.win_length
        pad = (hop - L % hop) % hop                    # aero.py:410-411 uses the INPUT hop for both variants
        Lp = L + pad
        if scale:
            hop, win = int(hop * m.scale), int(win * m.scale)
        z = self.ops.stft(x.reshape(B * Cc, L).contiguous(), L, Lp, m.nfft, hop, self._window(win, x.device),
                          m.nfft // 2, stats=stats, sig_per_item=Cc, win_len=win)
        return torch.view_as_complex(z).view(B, Cc, m.nfft // 2, -1)

    def ispec(self, z):
        """Aero._ispec (aero.py:423-428): complex64 [B, C, nfft/2, T] -> [B, C, hop_out*(T-1)]."""
        m = self.model
        B, Cc, F, T = z.shape
        hop, win = int(m.hop_length * m.scale), int(m.win_length * m.scale)
        zr = torch.view_as_real(z.contiguous()).view(B * Cc, F, T, 2)
        y = self.ops.istft(zr, m.nfft, hop, self._window(win, z.device), self._inv_env(win, hop, T, z.device),
                           hop * (T - 1))
        return y.view(B, Cc, -1)

    # ------------------------------------------------------------------ forward
    def forward(self, mix, want_spec=False, want_lr_spec=False, train=False):
        """`train`: the module is in training mode -- the FTB's BatchNorms use the statistics of this batch and their
        running statistics are updated (modules.py:287,293,300); nothing else in the forward path depends on the mode.
        Clips are independent units: with `self.streams` > 1 the batch is cut into that many sub-batches whose
        kernel sequences are enqueued on separate HIP streams, so latency-bound launches of one sub-batch (the
        recurrent LSTM kernel: one block per CU, 200 dependent steps) overlap with bandwidth/MFMA-bound launches of
        the others.  Results are BIT-identical to the single-stream order (per-clip arithmetic does not depend on the batch a clip sits
        in: tests/test_gpu_model.py::test_two_stream_forward_equals_one_stream)."""
        if train:
            return self._forward_one(mix, want_spec, want_lr_spec, train=True)
        want = self.streams if self.streams > 0 else (2 if mix.shape[0] >= 32 else 1)
        ns = min(want, mix.shape[0]) if mix.is_cuda else 1
        if self.use_graph and mix.is_cuda and self.ops.prof is None and not self.lib.is_emulator:
            return self._forward_graph(mix, want_spec, want_lr_spec)
        if ns <= 1 or (self.ops.prof is not None and not self.prof_streams):
            return self._forward_one(mix, want_spec, want_lr_spec)
        cur = torch.cuda.current_stream(mix.device)
        key = ('streams', ns, str(mix.device))
        if key not in self._tables:
            self._tables[key] = side_streams(mix.device, ns)
        outs = []
        self._prepare(mix.device)                       # weights packed once, on the caller's stream
        # STAGGER: sub-batch k + 1 starts when sub-batch k has finished stage `stagger` (1..4: encoder layers, 5..8: decoder layers).
        # In lockstep the two halves run the same kernel at the same time -- two latency-bound LSTM launches next to each other hide
        # nothing; shifted, the recurrent phase of one half runs under the MFMA / bandwidth-bound phase of the other.
        prev_events = None
        for st, part in zip(self._tables[key], mix.chunk(ns, dim=0)):
            st.wait_stream(cur)
            if prev_events is not None and self.stagger in prev_events:
                st.wait_event(prev_events[self.stagger])
            events = {}

            def stage_cb(i, events=events, st=st):
                if i == self.stagger:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    events[i] = ev
            with torch.cuda.stream(st):
                outs.append(self._forward_one(part, want_spec, want_lr_spec, defer_istft=True, stage_cb=stage_cb if self.stagger else None))
            prev_events = events
        for st in self._tables[key]:
            cur.wait_stream(st)
        # The iSTFT of the halves runs AFTER the join, on the caller's stream.  History: in round 3 an iSTFT that shared the chip with the
        # other half's convolutions now and then returned wrong 512-sample segments; round 4 traced that class to packed-fp32 VALU code and
        # round 5 to the neighbour's LDS-DMA copies as the second ingredient -- the library is built without packed fp32, and
        # BatchPipeline runs every batch's iSTFT beside other batches' kernels by design (tests/test_gpu_concurrency.py fences both).  The
        # join stays here because it costs nothing measurable and keeps one iSTFT launch order for the two halves.
        ys = []
        for o in outs:
            spec_out, hop, win, T, Lout = o[0]
            spec_out.record_stream(cur)
            y = self.ops.istft(spec_out, self.model.nfft, hop, self._window(win, mix.device), self._inv_env(win, hop, T, mix.device), Lout)
            ys.append(y.view(spec_out.shape[0], 1, Lout))
        res = [torch.cat(ys, 0)]
        for k in (1, 2):
            parts = [o[k] for o in outs]
            if parts[0] is None:
                res.append(None)
                continue
            for t in parts:
                t.record_stream(cur)
            res.append(torch.cat(parts, 0))
        return tuple(res)

    def _forward_graph(self, mix, want_spec, want_lr_spec):
        """The whole launch sequence of one forward (~130 kernels through ctypes) captured once per input shape as a HIP
        graph and replayed: at small batch the pass is launch-bound on the host (B = 1: 2.9 ms eager), the graph
        removes the per-launch Python / ctypes / runtime cost.  Inputs are copied into the graph's static buffer;
        outputs are cloned out of it (the graph's memory is reused by the next replay)."""
        key = ('graph', tuple(mix.shape), str(mix.device), want_spec, want_lr_spec,
               self.fuse_dconv_tail, self.fuse_stats, self.collapse_first_ftb, self.fuse_lstm_proj, self.fuse_enc0, self.fuse_dconv_row, self.split_taps)
        self._prepare(mix.device)                       # (drops the captured graphs if the weights changed)
        ent = self._tables.get(key)
        if ent is None:
            static_in = mix.clone()
            for _ in range(2):                          # warm-up outside capture: lazy one-time setup (attributes, tables)
                self._forward_one(static_in, want_spec, want_lr_spec)
            torch.cuda.synchronize(mix.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = self._forward_one(static_in, want_spec, want_lr_spec)
            ent = self._tables[key] = (g, static_in, static_out)
        g, static_in, static_out = ent
        static_in.copy_(mix)
        g.replay()
        return tuple(None if t is None else t.clone() for t in static_out)

    def _forward_one(self, mix, want_spec=False, want_lr_spec=False, train=False, defer_istft=False, stage_cb=None):
        m, ops, P = self.model, self.ops, None
        self._train = train
        if stage_cb is None:
            stage_cb = self.stage_hook                  # (BatchPipeline: stream switch at layer boundaries, aero_amd/pipeline.py)
        self._check_input(mix)
        if m.in_channels != 1 or m.out_channels != 1:
            raise NotImplementedError('only in_channels = out_channels = 1 (all reference configs)')
        dev = mix.device
        self._prepare(dev)
        P = self.P
        B, _, L = mix.shape
        mix = mix.contiguous()
        ops.begin_step(dev)
        stats = ops.new_stats(B, 1, 1, False, dev)
        fused = None
        if not want_lr_spec:                                               # nobody reads the fp32 low-rate spectrogram: normalised fp16 directly
            hop_in, pad_in = m.hop_length, (m.hop_length - L % m.hop_length) % m.hop_length
            fused = ops.stft_normalized(mix.view(B, L), L, L + pad_in, m.nfft, hop_in, self._window(m.win_length, dev), m.nfft // 2, stats, 1,
                                        m.win_length)
        if fused is not None:
            zc = None
            x, mean_std = fused
            F0, T = x.shape[1], x.shape[2]
        else:
            zc = self.spec(mix, stats=stats)                               # complex64 [B,1,F0,T]
            F0, T = zc.shape[2], zc.shape[3]
            z = torch.view_as_real(zc).view(B, F0, T, 2)
            x, mean_std = ops.spec_normalize(z, B, stats)                  # fp16 [B,F0,T,2]
        mean = mean_std[:, 0].contiguous()
        std = mean_std[:, 1].contiguous()

        saved = []
        Fq = F0
        for i, enc in enumerate(m.encoder):
            x, Fq = self._encode(i, enc, P[f'encoder.{i}'], x, B, Fq, T)
            saved.append((x, Fq))
            if stage_cb is not None:
                stage_cb(i + 1)
        x = None
        hop, win = int(m.hop_length * m.scale), int(m.win_length * m.scale)
        # nobody but the iSTFT reads the output spectrogram: its producer writes rows at the line-aligned pitch that kernel asks for
        self._out_pitch = None if (want_spec or not self.pitched_out) else ops.istft_pitch(m.nfft, hop, T)
        try:
            for j, dec in enumerate(m.decoder):
                skip, Fs = saved.pop()
                x = self._decode(j, dec, P[f'decoder.{j}'], x, skip, B, Fs, T, mean, std)
                if stage_cb is not None:
                    stage_cb(len(m.encoder) + j + 1)
        finally:
            self._out_pitch = None
        assert not saved
        spec_out = x                                                       # fp32 [B,F0,T,2] de-normalised (pitched: [B,F0,pitch,2], see Ops.istft)
        Lout = min(hop * (T - 1), int(L * m.scale))
        out_spec = torch.view_as_complex(spec_out).view(B, 1, F0, T) if want_spec else None
        if defer_istft:                                  # (two-stream forward: the caller runs the iSTFT after the streams have joined)
            return (spec_out, hop, win, T, Lout), out_spec, (zc if want_lr_spec else None)
        y = ops.istft(spec_out, m.nfft, hop, self._window(win, dev), self._inv_env(win, hop, T, dev), Lout)
        y = y.view(B, 1, Lout)
        return y, out_spec, (zc if want_lr_spec else None)

    def _encode(self, i, enc, L, x, B, Fq, T):
        ops = self.ops
        if getattr(self, '_train', False) and 'ftb_c1' in L:
            x = self._encode_head_train(i, enc, L, x, B, Fq, T)
            return self._encode_tail(i, enc, L, x, B, Fq, T)
        if 'ftb0' in L and self.collapse_first_ftb:
            Cc, rp = L['ftb0']['C'], L['ftb_rp']
            # pad channels (rp > r) must read as zero; with rp == r the conv writes every element
            c1 = (torch.empty if rp == L['ftb0_c1'].M else torch.zeros)(B, T, Fq * rp, dtype=torch.float16, device=x.device)
            ops.conv(L['ftb0_c1'], x, None, B, Fq, Fq, T, dst=c1, dst_strides=(T * Fq * rp, rp, Fq * rp))
            gate = ops.conv(L['ftb_c1d'], c1.view(B, 1, T, Fq * rp), None, B, 1, 1, T, tap_split=self._tap_split(L['ftb_c1d'], B, T))      # [B,1,T,Cc]
            okey = ('ones', B, T, str(x.device))
            if okey not in self._tables:                # (setdefault would allocate and fill a fresh tensor on every forward)
                self._tables[okey] = torch.ones(B, T, 2, dtype=torch.float16, device=x.device)
            ones = self._tables[okey]
            u = ops.freqfc(x, L['ftb_fc'], ones)                                           # freq_fc on (re, im) only
            conv = L['conv']
            if (self.fuse_enc0 and not enc.norm and Cc % 8 == 0 and conv.M % 16 == 0 and conv.M <= 64 and not conv.transposed
                    and all(t == 0 for t in conv.dt) and list(conv.df) == [j - enc.pad for j in range(len(conv.df))]):
                G = ops.conv(L['ftb0_g'], gate.view(B, 1, T, Cc), None, B, 1, 1, T)          # [B,1,T,3C]
                Fo = (Fq + 2 * enc.pad - enc.kernel_size) // enc.stride + 1
                ops.tag = 'stack'
                y = ops.enc0(x, u, G, L['ftb0'], conv, Fo, enc.stride, enc.pad, conv.act)
                ops.tag = ''
                return self._encode_tail(i, enc, L, None, B, Fq, T, y_pre=y)
            x = ops.ftb_first(x, u, gate.view(B, T, Cc), L['ftb0'])
        elif 'ftb_c1' in L or 'pre' in L:
            x = self._encode_head_unfused(L, x, B, Fq, T)
        return self._encode_tail(i, enc, L, x, B, Fq, T)

    def _encode_head_unfused(self, L, x, B, Fq, T):
        ops = self.ops
        if 'pre' in L:
            x = ops.conv(L['pre'], x, None, B, Fq, Fq, T)
        if 'ftb_c1' in L:
            Cc, rp = L['ftb_c2'].M, L['ftb_rp']
            c1 = (torch.empty if rp == L['ftb_c1'].M else torch.zeros)(B, T, Fq * rp, dtype=torch.float16, device=x.device)
            if self.use_pw and L.get('ftb_c1_sq') is not None and x.is_contiguous():
                ops.squeeze(L['ftb_c1_sq'], x, B, Fq, T, c1, rp)
            else:
                ops.conv(L['ftb_c1'], x, None, B, Fq, Fq, T, dst=c1, dst_strides=(T * Fq * rp, rp, Fq * rp))
            gate = ops.conv(L['ftb_c1d'], c1.view(B, 1, T, Fq * rp), None, B, 1, 1, T, tap_split=self._tap_split(L['ftb_c1d'], B, T))      # [B,1,T,Cc]
            fc = ops.freqfc(x, L['ftb_fc'], gate.view(B, T, Cc))
            if self.use_pw and L.get('ftb_c2_pw') is not None and x.is_contiguous() and fc.is_contiguous():
                x = ops.pw(L['ftb_c2_pw'], fc, B, Fq, T, x1=x)
            else:
                x = ops.conv(L['ftb_c2'], fc, x, B, Fq, Fq, T)
        return x

    def _encode_head_train(self, i, enc, L, x, B, Fq, T):
        """pre_conv + FTB with the BatchNorms in TRAINING mode (modules.py:304-325 under nn.Module.train()): each conv runs
        with its raw weights, the per-channel statistics of its output over the whole batch are accumulated
        (aero_norm_stats, per_row = 2, G = C), BatchNorm + ReLU are applied with them (aero_norm_apply) and the module's
        running statistics are updated as nn.BatchNorm does (momentum 0.1, unbiased variance)."""
        ops, dev = self.ops, x.device
        R = self._train_specs(i, enc, dev)
        ftb = enc.freq_attn_block
        if 'pre' in L:
            x = ops.conv(L['pre'], x, None, B, Fq, Fq, T)
        Cc, rp, r = R['c2'].M, L['ftb_rp'], R['c1'].M

        def bn(y, bnmod, dst=None, dst_strides=None):
            Bq, Fy, Ty, Cy = y.shape
            out = ops.norm_act(y, Cy, 2, bnmod.weight.detach().float(), bnmod.bias.detach().float(), ACT_RELU, eps=bnmod.eps,
                               dst=dst, dst_strides=dst_strides)
            st = ops._last_stats                                     # fp64 [Cy, 2]: sum, sum of squares over (B, F, T)
            n = float(Bq * Fy * Ty)
            with torch.no_grad():                                    # buffer bookkeeping of nn.BatchNorm (not the data path)
                mean = st[:, 0] / n
                var = (st[:, 1] / n - mean * mean).clamp_min(0)
                mom = bnmod.momentum
                bnmod.running_mean.mul_(1 - mom).add_(mean.to(bnmod.running_mean.dtype), alpha=mom)
                bnmod.running_var.mul_(1 - mom).add_((var * (n / max(n - 1.0, 1.0))).to(bnmod.running_var.dtype), alpha=mom)
                bnmod.num_batches_tracked.add_(1)
            return out
        y1 = ops.conv(R['c1'], x, None, B, Fq, Fq, T)                                        # [B,F,T,r], pre-BatchNorm
        c1 = (torch.empty if rp == r else torch.zeros)(B, T, Fq * rp, dtype=torch.float16, device=dev)
        bn(y1, ftb.conv1[1], dst=c1, dst_strides=(T * Fq * rp, rp, Fq * rp))                 # -> [B, T, F*rp] image
        y2 = ops.conv(R['c1d'], c1.view(B, 1, T, Fq * rp), None, B, 1, 1, T)                 # [B,1,T,C]
        gate = bn(y2, ftb.conv1d[1])
        fc = ops.freqfc(x, L['ftb_fc'], gate.view(B, T, Cc))
        y3 = ops.conv(R['c2'], fc, x, B, Fq, Fq, T)
        return bn(y3, ftb.conv2[1])

    def _train_specs(self, i, enc, device):
        """conv specs of the FTB with the RAW weights (no BatchNorm fold, no activation): training-mode forward only"""
        key = ('train_specs', i, self._key)
        if key not in self._tables:
            mk = pack.make_conv_spec
            sd = {k: v.detach().float().cpu() for k, v in enc.freq_attn_block.state_dict().items()}
            ftb = enc.freq_attn_block
            Fd, Cc, r = ftb.input_dim, ftb.in_channel, ftb.r_channel
            rp = r if (Fd * r) % 8 == 0 else 8 * ((r + 7) // 8)
            w, df, dt = pack.conv2d_taps(sd['conv1.0.weight'], 0, 0)
            R = {'c1': mk(w, sd['conv1.0.bias'], Cc, 0, df, dt, device)}
            w = sd['conv1d.0.weight']
            k9 = w.shape[-1]
            w = w.view(Cc, r, Fd, k9).permute(0, 3, 2, 1)
            wp = torch.zeros(Cc, k9, Fd, rp)
            wp[..., :r] = w
            R['c1d'] = mk(wp.reshape(1, Cc, k9, Fd * rp), sd['conv1d.0.bias'], Fd * rp, 0, [0] * k9, [j - (k9 // 2) for j in range(k9)], device)
            w, df, dt = pack.conv2d_taps(sd['conv2.0.weight'], 0, 0)
            R['c2'] = mk(w, sd['conv2.0.bias'], Cc, Cc, df, dt, device)
            for k in [k for k in self._tables if isinstance(k, tuple) and k and k[0] == 'train_specs' and k[1] == i]:
                del self._tables[k]
            self._tables[key] = R
        return self._tables[key]

    def _encode_tail(self, i, enc, L, x, B, Fq, T, y_pre=None):
        ops = self.ops
        Fo = (Fq + 2 * enc.pad - enc.kernel_size) // enc.stride + 1
        if y_pre is not None:
            return self._encode_rest(i, enc, L, y_pre, B, Fo, T)
        ops.tag = 'stack'                               # (SURVEY 8d "conv stack": the Conv2d / ConvTranspose2d of the U-Net proper)
        if enc.norm:
            st = self._stats_for(L['conv'].M, enc.norm_groups, B, Fo, x.device, spec=L['conv'])
            y = ops.conv(L['conv'], x, None, B, Fq, Fo, T, stat=self._acc(st, enc.norm_groups))
            ops.tag = ''
            y = ops.norm_act(y, enc.norm_groups, False, L['norm1'][0], L['norm1'][1], ACT_GELU, stats=st)
        else:
            y = ops.conv(L['conv'], x, None, B, Fq, Fo, T)
        ops.tag = ''
        return self._encode_rest(i, enc, L, y, B, Fo, T)

    def _encode_rest(self, i, enc, L, y, B, Fo, T):
        ops = self.ops
        self._mark(i + 0.25)                                # (stage marks for BatchPipeline's stagger: the layer's conv is out)
        if 'dconv' in L:
            self._mark_base = i
            y = self._dconv(enc.dconv, L['dconv'], y, B, Fo, T)
            self._mark_base = None
            self._mark(i + 0.75)
        if 'rewrite' in L:
            emb = self.P.get('freq_emb') if i == 0 else None
            if enc.norm:
                if emb is not None:
                    raise NotImplementedError('GroupNorm on encoder 0 together with the frequency embedding')
                st = self._stats_for(L['rewrite'].M, enc.norm_groups, B, Fo, y.device, spec=L['rewrite'])
                ops.tag = 'stack'
                if st is None and self.use_pw and L.get('rewrite_pw') is not None and y.is_contiguous():
                    r = ops.pw(L['rewrite_pw'], y, B, Fo, T)
                else:
                    r = ops.conv(L['rewrite'], y, None, B, Fo, Fo, T, stat=self._acc(st, enc.norm_groups))
                ops.tag = ''
                y = ops.norm_act(r, enc.norm_groups, False, L['norm2'][0], L['norm2'][1], ACT_GLU, stats=st)
            else:
                ops.tag = 'stack'
                if self.use_pw and L.get('rewrite_pw') is not None and y.is_contiguous():
                    y = ops.pw(L['rewrite_pw'], y, B, Fo, T, post_add=emb)
                else:
                    y = ops.conv(L['rewrite'], y, None, B, Fo, Fo, T, post_add=emb)
                ops.tag = ''
        elif i == 0 and 'freq_emb' in self.P:
            raise NotImplementedError('frequency embedding without a rewrite conv')
        return y, Fo

    def _mark(self, stage):
        if self.stage_hook is not None and not getattr(self, '_train', False):
            self.stage_hook(stage)

    def _dconv(self, dc, layers, x, B, Fo, T):
        ops = self.ops
        act = {'snake': ACT_SNAKE, 'gelu': ACT_GELU}.get(dc.act_func, ACT_RELU)
        if (self.fuse_dconv_row and all('row' in L for L in layers) and len(layers) <= _lib.DCONV_MAX_DEPTH and x.is_contiguous()
                and ops.dconv_row_fits(T, layers[0]['row']['C'], layers[0]['row']['hidden'], max(L['row']['dilation'] for L in layers))):
            return ops.dconv_row(x, [dict(L['row'], snake_a=L.get('snake_a')) for L in layers], act, Fo)
        for li, L in enumerate(layers):
            if li == 1 and getattr(self, '_mark_base', None) is not None:
                self._mark(self._mark_base + 0.5)            # (between the two DConv layers of an encoder)
            g1 = L['gn1']
            st1 = None
            if g1 is not None and self._want_stats(L['conv1']) and L['conv1'].M > 16 and L['conv1'].M % 8 == 0:       # (M <= 16 runs on the streaming kernel)
                st1 = ops.new_stats(B, Fo, 1, True, x.device)
                h = ops.conv(L['conv1'], x, None, B, Fo, Fo, T, stat=dict(mode=1, stats=st1, G=1, per_row=True))
            else:
                h = ops.conv(L['conv1'], x, None, B, Fo, Fo, T)
            g2 = L['gn2']
            hp = L.get('hid_pad') if (g2 is not None and self.fuse_dconv_tail and L['conv2_glu'].M % 16 == 0) else None
            hdst = None
            if hp:
                hid = h.shape[-1]
                key = ('hidpad', B, Fo, T, hp, str(x.device), self.ops.stream(x))
                if key not in self._tables:                   # pad channels are zero and never written afterwards
                    self._tables[key] = torch.zeros(B, Fo, T, hp, dtype=torch.float16, device=x.device)
                hdst = self._tables[key][..., :hid]
            h = ops.norm_act(h, 1, True, g1[0] if g1 else None, g1[1] if g1 else None, act,
                             snake_a=L.get('snake_a'), normalize=g1 is not None, stats=st1, dst=hdst)
            if hp:
                st2 = ops.new_stats(B, Fo, 1, True, x.device)
                c2, hb = L['conv2_glu_pad'], self._tables[key]
                if self.gram_stats and 'gram_pad' in L:
                    ops.gram_stats(hb, L['gram_pad'], st2)
                else:
                    ops.conv(c2, hb, None, B, Fo, Fo, T, act=ACT_NONE, stat=dict(mode=2, stats=st2, G=1, per_row=True))
                x = ops.conv(c2, hb, None, B, Fo, Fo, T, res=x,
                             stat=dict(mode=3, stats=st2, G=1, per_row=True, count=float(T * c2.M),
                                       gamma=L['gn2_glu'][0], beta=L['gn2_glu'][1], layer_scale=L['scale']))
                continue
            if 'lstm' in L:
                h = self._blstm(dc, L, h, B, Fo, T)
            if 'attn_qkvd' in L:
                heads, ndecay = L['attn_geom']
                pwq, pwp = (L.get('attn_qkvd_pw'), L.get('attn_proj_pw')) if (self.use_pw and h.is_contiguous()) else (None, None)
                qkvd = ops.pw(pwq, h, B, Fo, T) if pwq is not None else ops.conv(L['attn_qkvd'], h, None, B, Fo, Fo, T)
                att = ops.localstate(qkvd, B * Fo, T, dc.hidden, heads, ndecay)
                if pwp is not None:
                    h = ops.pw(pwp, att.view(B, Fo, T, dc.hidden), B, Fo, T, res=h)
                else:
                    h = ops.conv(L['attn_proj'], att.view(B, Fo, T, dc.hidden), None, B, Fo, Fo, T, res=h)
            if g2 is not None and self.fuse_dconv_tail and L['conv2_glu'].M % 16 == 0 and L['conv2_glu'].C0 % 8 == 0:
                # pass 0: statistics of conv2(h) only (nothing stored); pass 1: recompute, normalise, GLU, scale, + skip
                st2 = ops.new_stats(B, Fo, 1, True, x.device)
                c2 = L['conv2_glu']
                if self.gram_stats and 'gram' in L:
                    ops.gram_stats(h, L['gram'], st2)
                else:
                    ops.conv(c2, h, None, B, Fo, Fo, T, act=ACT_NONE, stat=dict(mode=2, stats=st2, G=1, per_row=True))
                if self.use_pw and L.get('pw2') is not None and h.is_contiguous():
                    # the tail as ONE streaming pass: weights resident in registers, no LDS staging, no barrier (k_pw.h)
                    x = ops.pw(L['pw2'], h, B, Fo, T, res=x, stats=st2, count=float(T * c2.M), gamma=L['gn2_glu'][0],
                               beta=L['gn2_glu'][1], layer_scale=L['scale'])
                    continue
                x = ops.conv(c2, h, None, B, Fo, Fo, T, res=x,
                             stat=dict(mode=3, stats=st2, G=1, per_row=True, count=float(T * c2.M),
                                       gamma=L['gn2_glu'][0], beta=L['gn2_glu'][1], layer_scale=L['scale']))
                continue
            st2 = ops.new_stats(B, Fo, 1, True, x.device) if (g2 is not None and self._want_stats(L['conv2']) and L['conv2'].M % 8 == 0) else None
            g = ops.conv(L['conv2'], h, None, B, Fo, Fo, T,
                         stat=None if st2 is None else dict(mode=1, stats=st2, G=1, per_row=True))
            x = ops.norm_act(g, 1, True, g2[0] if g2 else None, g2[1] if g2 else None, ACT_GLU,
                             layer_scale=L['scale'], res=x, normalize=g2 is not None, stats=st2)
        return x

    def _tap_split(self, spec, B, T):
        """3-way split over the time taps for the long thin FTB Conv1d (aero_hip.h "tap split"): K >= 2048 on <= 512 tiles"""
        ktot = spec.weight.shape[-1]                                 # ntaps * Cp
        nT = len(set(spec.dt))
        if (self.split_taps and nT == len(spec.dt) and nT % 3 == 0 and len(set(spec.df)) == 1 and 16 < spec.M <= 128 and ktot >= 2048
                and B * ((T + 127) // 128) <= 512 and spec.act != ACT_GLU and not spec.transposed):
            return 3
        return 1

    def _want_stats(self, spec):
        return self.fuse_stats is True or (self.fuse_stats == 'auto' and spec is not None and len(spec.df) > 1)

    def _stats_for(self, M, G, B, F, device, spec=None):
        """fp64 accumulators for a GroupNorm over [B, M, F, T] if the conv epilogue can fill them, else None.
        The software-pipelined wide-contraction kernel (spec.tiled_bm) accumulates them for free in its epilogue when the
        groups are 32-row aligned; the other tile kernels only on request (AERO_FUSE_STATS: measured slower there)."""
        if spec is not None and spec.tiled_bm and self.ring_stats and M % G == 0 and (M // G) % 32 == 0:
            return self.ops.new_stats(B, F, G, False, device)
        if self._want_stats(spec) and self.ops.can_fuse_stats(M, G):
            return self.ops.new_stats(B, F, G, False, device)
        return None

    @staticmethod
    def _acc(st, G):
        return None if st is None else dict(mode=1, stats=st, G=G, per_row=False)

    def _blstm(self, dc, L, h, B, Fo, T):
        """BLSTM (modules.py:32-65): framing and stitching are index arithmetic inside the LSTM kernel."""
        ops = self.ops
        H = dc.hidden
        R = B * Fo
        max_steps = 200
        framed = T > max_steps
        if framed:
            W, S = max_steps, max_steps // 2
            nframes = blstm_frames(T, W, S)                 # models/utils.py:29 (minus a last frame nothing of which survives the stitch)
        else:
            W, S, nframes = T, 1, 1
        nseq = R * nframes
        # frame-major sequence order: a block of 16 sequences belongs to one frame, so the stitching layer stops where its frame's kept range
        # ends (k_lstm.h: a quarter of a middle frame's steps); the unfused projection path indexes its [B,Fo,T,8H] tensor per row and keeps the
        # reference's order
        fm = 1 if (framed and self.lstm_frame_major) else 0
        (pj0, xb0, whh0, fz0), (pj1, xb1, whh1, fz1) = L['lstm']
        out0 = torch.empty(nseq, W, 2 * H, dtype=torch.float16, device=h.device)
        out1 = torch.empty(R, T, 2 * H, dtype=torch.float16, device=h.device)
        if self.stage_hook is not None and not getattr(self, '_train', False):
            self.stage_hook('lstm+')                         # (BatchPipeline: the two recurrent launches on their own stream kind)
        if fz0 is not None and self.fuse_lstm_proj and h.is_contiguous():
            ops.lstm(None, None, whh0, H, nseq, W, 1 if framed else 0, 0, nframes, S, T, out0, x=h.view(R, T, H), fused=fz0, frame_major=fm)
        else:
            xp0 = ops.conv(pj0, h, None, B, Fo, Fo, T)                              # [B,Fo,T,8H], per position not per frame
            ops.lstm(xp0, xb0, whh0, H, nseq, W, 1 if framed else 0, 0, nframes, S, T, out0, frame_major=fm)
        if fz1 is not None and self.fuse_lstm_proj:
            ops.lstm(None, None, whh1, H, nseq, W, 0, 1 if framed else 0, nframes, S, T, out1, x=out0, fused=fz1, frame_major=fm)
        else:
            xp1 = ops.conv(pj1, out0.view(nseq, 1, W, 2 * H), None, nseq, 1, 1, W)  # [nseq,1,W,8H]
            ops.lstm(xp1, xb1, whh1, H, nseq, W, 0, 1 if framed else 0, nframes, S, T, out1, frame_major=fm)
        if self.stage_hook is not None and not getattr(self, '_train', False):
            self.stage_hook('lstm-')
        if self.use_pw and L.get('lstm_lin_pw') is not None:
            return ops.pw(L['lstm_lin_pw'], out1.view(B, Fo, T, 2 * H), B, Fo, T, res=h)
        return ops.conv(L['lstm_lin'], out1.view(B, Fo, T, 2 * H), None, B, Fo, Fo, T, res=h)

    def _decode(self, j, dec, L, x, skip, B, Fq, T, mean, std):
        self.ops.tag = 'stack'
        try:
            return self._decode_tagged(j, dec, L, x, skip, B, Fq, T, mean, std)
        finally:
            self.ops.tag = ''

    def _decode_tagged(self, j, dec, L, x, skip, B, Fq, T, mean, std):
        ops = self.ops
        if 'rewrite' in L:
            if dec.last and 'tail' in L and self.fuse_tail:
                # the whole last layer in two launches: rewrite 3x3 + GLU + the transposed conv's 16 tap products per time step (the 96-channel
                # activation never reaches memory), then the row combination + bias + x*std + mean (aero.py:189-215, 497-498)
                timg, tbias = L['tail']
                lo, hi = ops.conv(L['rewrite'], x, skip, B, Fq, Fq, T, tail=timg)
                return ops.convtr_tail_finish(lo, hi, tbias, std, mean, (Fq - 1) * dec.stride + dec.kernel_size - 2 * dec.pad, dec.pad, timg.shape[1],
                                              pitched=self._out_pitch)
            if dec.norm:
                st = self._stats_for(L['rewrite'].M, dec.norm_groups, B, Fq, skip.device, spec=L['rewrite'])
                r = ops.conv(L['rewrite'], x, skip, B, Fq, Fq, T, stat=self._acc(st, dec.norm_groups))
                y = ops.norm_act(r, dec.norm_groups, False, L['norm1'][0], L['norm1'][1], ACT_GLU, stats=st)
            else:
                y = ops.conv(L['rewrite'], x, skip, B, Fq, Fq, T)
        else:
            raise NotImplementedError('decoder without rewrite conv')
        Fu = (Fq - 1) * dec.stride + dec.kernel_size          # untrimmed rows of the transposed conv
        Ft = Fu - 2 * dec.pad
        if dec.norm:
            if dec.last:
                raise NotImplementedError('GroupNorm on the last decoder layer (norm_starts = 0)')
            st = self._stats_for(L['conv_tr'].M, dec.norm_groups, B, Fu, y.device, spec=L['conv_tr'])
            if st is None and 'conv_tr_stacked' in L:
                z = self._convtr_stacked(L['conv_tr_stacked'], y, B, Fq, T, dec.stride, 0, Fu)
            else:
                z = ops.conv(L['conv_tr'], y, None, B, Fq, Fu, T, stat=self._acc(st, dec.norm_groups))
            return ops.norm_act(z, dec.norm_groups, False, L['norm2'][0], L['norm2'][1],
                                ACT_NONE if dec.last else ACT_GELU, f_lo=dec.pad, f_cnt=Ft, stats=st)
        if dec.last:
            # aero.py:497-498: x*std + mean fused into the last epilogue; fp32 [B,F0,T,2] == complex64 [B,1,F0,T]
            return ops.conv(L['conv_tr'], y, None, B, Fq, Fu, T, dst_f32=True, dst_f_off=dec.pad, dst_F=Ft,
                            batch_scale=std, batch_shift=mean)
        if 'conv_tr_stacked' in L:
            return self._convtr_stacked(L['conv_tr_stacked'], y, B, Fq, T, dec.stride, dec.pad, Ft)
        return ops.conv(L['conv_tr'], y, None, B, Fq, Fu, T, dst_f_off=dec.pad, dst_F=Ft)

    def _convtr_stacked(self, spec, y, B, Fq, T, stride, first, rows):
        """ConvTranspose2d from the input side: rows q = 0..ceil(Fu/stride)-1, channel block r -> frequency row q*stride+r."""
        Mo = spec.extra['scatter_M']
        NR = Fq - 1 + len(spec.df)                                 # input-aligned rows that touch any output row
        out = torch.empty(B, rows, T, Mo, dtype=torch.float16, device=y.device)
        self.ops.conv(spec, y, None, B, Fq, NR, T, dst=out.view(B, rows, T, Mo), dst_F=NR,
                      dst_strides=_strides4(out), scatter=(Mo, stride, first, rows))
        return out
