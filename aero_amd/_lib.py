"""ctypes binding of the C ABI in include/aero_hip.h (the "reference-side stub" of INTEGRATION.md).

`load()` opens aero_amd/libaero_hip.so, the hipcc/gfx950 build.  There is no fallback: if the
library is missing or a symbol is absent the import fails loudly.  (tests/ may pass an explicit
path to the CPU-emulated test double; product code never does.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, 'libaero_hip.so')

ACT_NONE, ACT_RELU, ACT_GELU, ACT_GLU, ACT_SNAKE = 0, 1, 2, 3, 4

i32, i64, vp, fp, dp = C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [('src0', vp), ('s0_b', i64), ('s0_f', i64), ('s0_t', i64), ('C0', i32),
                ('src1', vp), ('s1_b', i64), ('s1_f', i64), ('s1_t', i64), ('C1', i32),
                ('weight', vp), ('bias', fp),
                ('dst', vp), ('d_b', i64), ('d_f', i64), ('d_t', i64),
                ('dst_f32', i32), ('dst_f_off', i32), ('dst_F', i32),
                ('B', i32), ('Fin', i32), ('Fout', i32), ('T', i32), ('M', i32),
                ('transposed', i32), ('fstride', i32),
                ('ntaps', i32), ('df', i32 * 9), ('dt', i32 * 9),
                ('act', i32),
                ('res', vp), ('r_b', i64), ('r_f', i64), ('r_t', i64),
                ('post_add', fp), ('batch_scale', fp), ('batch_shift', fp),
                ('stats', dp), ('stat_count', C.c_double),
                ('stat_mode', i32), ('stat_G', i32), ('stat_per_row', i32), ('stat_eps', C.c_float),
                ('gamma', fp), ('beta', fp), ('layer_scale', fp),
                ('scatter_M', i32), ('scatter_stride', i32), ('scatter_off', i32), ('scatter_F', i32),
                ('weight_tiled', vp), ('tiled_bm', i32), ('tap_split', i32), ('split_acc', fp),
                ('tail_w', vp), ('tail_lo', fp), ('tail_hi', fp), ('tail_cp', i32)]


class NormDesc(C.Structure):
    _fields_ = [('src', vp), ('s_b', i64), ('s_f', i64), ('s_t', i64),
                ('B', i32), ('F', i32), ('T', i32), ('C', i32), ('G', i32), ('per_row', i32),
                ('eps', C.c_float), ('stats', dp), ('stat_count', C.c_double), ('gamma', fp), ('beta', fp),
                ('act', i32), ('snake_a', fp), ('layer_scale', fp),
                ('res', vp), ('r_b', i64), ('r_f', i64), ('r_t', i64),
                ('dst', vp), ('d_b', i64), ('d_f', i64), ('d_t', i64)]


class WgradDesc(C.Structure):
    _fields_ = [('dy', vp), ('dy_b', i64), ('dy_f', i64), ('dy_t', i64),
                ('x', vp), ('x_b', i64), ('x_f', i64), ('x_t', i64),
                ('dw', fp), ('db', fp),
                ('B', i32), ('Fin', i32), ('Fout', i32), ('T', i32), ('M', i32), ('C', i32), ('ntaps', i32), ('fstride', i32),
                ('df', i32 * 9), ('dt', i32 * 9), ('slabs', fp), ('nslab', i32),
                ('store', i32), ('dw_layout', i32), ('dw_rowlen', i32), ('dw_coff', i32)]


class NormBwdDesc(C.Structure):
    _fields_ = [('x', vp), ('x_b', i64), ('x_f', i64), ('x_t', i64),
                ('dy', vp), ('dy_b', i64), ('dy_f', i64), ('dy_t', i64),
                ('dx', vp), ('dx_b', i64), ('dx_f', i64), ('dx_t', i64),
                ('B', i32), ('F', i32), ('T', i32), ('C', i32), ('G', i32), ('per_row', i32),
                ('eps', C.c_float), ('stats', dp), ('stat_count', C.c_double),
                ('gamma', fp), ('beta', fp), ('layer_scale', fp), ('act', i32),
                ('sums', dp), ('dgamma', fp), ('dbeta', fp), ('dlayer_scale', fp), ('snake_a', fp), ('dsnake_a', fp), ('psums', dp)]


class PwDesc(C.Structure):
    _fields_ = [('x', vp), ('x_b', i64), ('x_f', i64), ('x_t', i64), ('C', i32),
                ('wimg', vp), ('bias', fp),
                ('stats', dp), ('stat_count', C.c_double), ('stat_eps', C.c_float),
                ('gamma', fp), ('beta', fp), ('layer_scale', fp), ('post_add', fp),
                ('res', vp), ('r_b', i64), ('r_f', i64), ('r_t', i64),
                ('dst', vp), ('d_b', i64), ('d_f', i64), ('d_t', i64),
                ('B', i32), ('F', i32), ('T', i32), ('M', i32), ('act', i32),
                ('x1', vp), ('x1_b', i64), ('x1_f', i64), ('x1_t', i64), ('C0', i32)]


class GramDesc(C.Structure):
    _fields_ = [('x', vp), ('s_b', i64), ('s_f', i64), ('s_t', i64),
                ('B', i32), ('F', i32), ('T', i32), ('C', i32),
                ('G', dp), ('g1', dp), ('stats', dp)]


class LstmDesc(C.Structure):
    _fields_ = [('xproj', vp), ('xbias', vp), ('whh', vp), ('out', vp),
                ('H', i32), ('nseq', i32), ('W', i32), ('in_mode', i32), ('out_mode', i32),
                ('nframes', i32), ('S', i32), ('T', i32),
                ('x', vp), ('wih', vp), ('bias', fp), ('in_ch', i32), ('x_pitch', i32),
                ('save_gates', vp), ('save_c', fp), ('frame_major', i32)]


class LstmBwdDesc(C.Structure):
    _fields_ = [('dout', vp), ('whh_t', vp), ('save_gates', vp), ('save_c', fp), ('da', vp),
                ('H', i32), ('nseq', i32), ('W', i32), ('out_mode', i32), ('nframes', i32), ('S', i32), ('T', i32)]


class AttnBwdDesc(C.Structure):
    _fields_ = [('qkvd', vp), ('ld', i64), ('out', vp), ('dout', vp), ('dqkvd', vp), ('qstats', fp),
                ('R', i32), ('T', i32), ('C', i32), ('heads', i32), ('ndecay', i32), ('decay_scale', C.c_float)]


class AttnDesc(C.Structure):
    _fields_ = [('qkvd', vp), ('ld', i64), ('out', vp),
                ('R', i32), ('T', i32), ('C', i32), ('heads', i32), ('ndecay', i32)]


class GconvDesc(C.Structure):
    _fields_ = [('x', vp), ('w', vp), ('bias', fp), ('y', vp),
                ('B', i32), ('Tin', i32), ('Cin', i32), ('Cout', i32), ('groups', i32), ('K', i32), ('stride', i32), ('pad', i32),
                ('reflect', i32), ('slope', C.c_float), ('w_mfma', vp)]


class GconvBwdDesc(C.Structure):
    _fields_ = [('x', vp), ('w', vp), ('y', vp), ('dy', vp), ('dx', vp), ('dw', fp), ('db', fp),
                ('B', i32), ('Tin', i32), ('Cin', i32), ('Cout', i32), ('groups', i32), ('K', i32), ('stride', i32), ('pad', i32),
                ('reflect', i32), ('slope', C.c_float), ('w_dgrad_mfma', vp), ('slabs', fp), ('nslab', i32)]


class FreqFcDesc(C.Structure):
    _fields_ = [('x', vp), ('w', vp), ('gate', vp), ('dst', vp),
                ('B', i32), ('F', i32), ('T', i32), ('C', i32)]


class FtbFirstDesc(C.Structure):
    _fields_ = [('xn', vp), ('u', vp), ('gate', vp), ('w2a', vp),
                ('p0', fp), ('p1', fp), ('pb', fp), ('rs', fp), ('a_re', fp), ('a_im', fp), ('bias', fp),
                ('dst', vp), ('B', i32), ('F', i32), ('T', i32), ('C', i32)]


class Enc0Desc(C.Structure):
    _fields_ = [('xn', vp), ('u', vp), ('g', vp), ('rs', fp), ('a_re', fp), ('a_im', fp), ('bias_f', fp),
                ('wc', vp), ('bias_c', fp), ('dst', vp),
                ('B', i32), ('F', i32), ('T', i32), ('C', i32), ('M', i32), ('Fo', i32), ('ktaps', i32), ('stride', i32),
                ('pad', i32), ('act', i32)]


DCONV_MAX_DEPTH = 4


class DconvLayer(C.Structure):
    _fields_ = [('w1', vp), ('w2', vp), ('consts', fp), ('snake_a', fp),
                ('dilation', i32), ('norm1', i32), ('norm2', i32), ('reserved', i32)]


class DconvDesc(C.Structure):
    _fields_ = [('x', vp), ('y', vp), ('R', i32), ('T', i32), ('C', i32), ('hidden', i32), ('depth', i32), ('act', i32),
                ('F', i32), ('eps', C.c_float), ('layer', DconvLayer * DCONV_MAX_DEPTH)]


_PROTOS = {
    'aero_version': (C.c_char_p, []),
    'aero_last_error': (C.c_char_p, []),
    'aero_last_kernel_name': (C.c_char_p, []),
    'aero_stft_fwd': (i32, [fp, i32, i32, i32, i32, i32, fp, i32, fp, i32, dp, i32, vp]),
    'aero_stft_dft_table_bytes': (i64, [i32]),
    'aero_stft_dft_table': (i32, [fp, i32, i32, vp, vp]),
    'aero_stft_dft_fwd': (i32, [fp, i32, i32, i32, i32, i32, i32, vp, fp, i32, dp, i32, vp]),
    'aero_stft_dft_norm_fwd': (i32, [fp, i32, i32, i32, i32, i32, i32, vp, i32, dp, i32, vp, fp, vp]),
    'aero_spec_normalize': (i32, [fp, i32, i64, dp, vp, fp, vp]),
    'aero_istft_fwd': (i32, [fp, i32, i32, i32, i32, i32, fp, fp, fp, i32, vp]),
    'aero_istft_pitch': (i32, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    'aero_istft_pitched_fwd': (i32, [fp, i32, i32, i32, i32, i32, i32, i32, fp, fp, fp, i32, vp]),
    'aero_conv_fwd': (i32, [C.POINTER(ConvDesc), vp]),
    'aero_split_finish': (i32, [fp, i32, fp, i32, vp, i64, i32, vp]),
    'aero_adam_step': (i32, [fp, fp, fp, fp, i64, C.c_float, C.c_float, C.c_float, C.c_float, i32, C.c_float, vp]),
    'aero_adam_step_dev': (i32, [fp, fp, fp, fp, i64, C.c_float, C.c_float, C.c_float, C.c_float, fp, C.c_float, vp]),
    'aero_conv_wgrad': (i32, [C.POINTER(WgradDesc), vp]),
    'aero_conv_wgrad_chunks': (i32, [i32, i32, i32, i32, i32]),
    'aero_norm_bwd_reduce': (i32, [C.POINTER(NormBwdDesc), vp]),
    'aero_norm_bwd_apply': (i32, [C.POINTER(NormBwdDesc), vp]),
    'aero_istft_bwd_prep': (i32, [fp, fp, fp, i32, i32, i32, i32, i32, vp]),
    'aero_istft_bwd_pack': (i32, [fp, fp, i32, i32, i32, i32, i32, vp]),
    'aero_conv_tile_m': (i32, [i32]),
    'aero_conv_ring_bm': (i32, [i32, i32]),
    'aero_convtr_tail_finish': (i32, [fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, vp]),
    'aero_convtr_tail_finish_pitched': (i32, [fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'aero_conv_kernel_name': (i32, [C.POINTER(ConvDesc), C.c_char_p, i32]),
    'aero_norm_stats': (i32, [C.POINTER(NormDesc), vp]),
    'aero_norm_apply': (i32, [C.POINTER(NormDesc), vp]),
    'aero_gram_stats': (i32, [C.POINTER(GramDesc), vp]),
    'aero_lstm_fwd': (i32, [C.POINTER(LstmDesc), vp]),
    'aero_lstm_geometry': (i32, [i32, C.POINTER(i32), C.POINTER(i32)]),
    'aero_lstm_geometry_in': (i32, [i32, i32, C.POINTER(i32)]),
    'aero_localstate_fwd': (i32, [C.POINTER(AttnDesc), vp]),
    'aero_freqfc_fwd': (i32, [C.POINTER(FreqFcDesc), vp]),
    'aero_ftb_first_fwd': (i32, [C.POINTER(FtbFirstDesc), vp]),
    'aero_enc0_fwd': (i32, [C.POINTER(Enc0Desc), vp]),
    'aero_dconv_row_fwd': (i32, [C.POINTER(DconvDesc), vp]),
    'aero_dconv_row_fits': (i32, [i32, i32, i32, i32]),
    'aero_lstm_bwd': (i32, [C.POINTER(LstmBwdDesc), vp]),
    'aero_lstm_bwd_k4p': (i32, [i32]),
    'aero_localstate_bwd': (i32, [C.POINTER(AttnBwdDesc), vp]),
    'aero_freqfc_wgrad': (i32, [vp, vp, vp, fp, fp, i32, i32, i32, i32, i32, vp]),
    'aero_ftb_gate_bwd': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    'aero_sum_bt': (i32, [vp, fp, i32, i32, i32, i32, C.c_float, vp]),
    'aero_frames_op': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'aero_stft_loss_sums': (i32, [fp, fp, i64, C.c_float, dp, i32, dp, vp]),
    'aero_stft_loss_bwd': (i32, [fp, fp, i64, C.c_float, dp, C.c_float, C.c_float, fp, fp, vp]),
    'aero_irfft_frames': (i32, [fp, i32, i32, i32, i32, fp, fp, vp]),
    'aero_stft_adj_fold': (i32, [fp, fp, i32, i32, i32, i32, i32, i32, vp]),
    'aero_add_f16': (i32, [vp, vp, vp, i64, C.c_float, vp]),
    'aero_scale_cast': (i32, [fp, i32, i64, fp, vp, C.c_float, vp, fp, vp]),
    'aero_scale_f32': (i32, [fp, i64, fp, vp]),
    'aero_gather_pack': (i32, [vp, vp, i32, vp, vp, i64, i32, vp]),
    'aero_debug_probe': (i32, [vp, i32, i32, i32, vp, vp]),
    'aero_bn_running_update': (i32, [dp, i32, C.c_double, C.c_float, fp, fp, vp, vp]),
    'aero_gconv1d_mfma_ok': (i32, [i32, i32, i32, i32, i32, i32, i32]),
    'aero_weightnorm_fwd': (i32, [fp, fp, fp, i32, i32, vp]),
    'aero_weightnorm_bwd': (i32, [fp, i64, i64, i64, fp, fp, fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, vp]),
    'aero_gconv1d_wgrad_slabs': (i32, [i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    'aero_rescale_f16': (i32, [vp, fp, vp, fp, i64, vp, C.c_float, vp, fp, vp]),
    'aero_gconv1d_fwd': (i32, [C.POINTER(GconvDesc), vp]),
    'aero_pw_fwd': (i32, [C.POINTER(PwDesc), vp]),
    'aero_comm_unique_id': (i32, [vp]),
    'aero_comm_init': (i32, [i32, i32, vp, C.POINTER(vp)]),
    'aero_allreduce_f32': (i32, [vp, fp, i64, vp]),
    'aero_allgather': (i32, [vp, vp, vp, i64, vp]),
    'aero_comm_destroy': (i32, [vp]),
    'aero_stream_create': (i32, [i32, C.POINTER(C.c_uint32), i32, C.POINTER(vp)]),
    'aero_stream_destroy': (i32, [vp]),
    'aero_pw_rows': (i32, [i32, i32]),
    'aero_pw_ksteps': (i32, [i32]),
    'aero_squeeze_fwd': (i32, [vp, i64, i64, i64, vp, fp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    'aero_leaky_relu': (i32, [vp, i64, C.c_float, vp]),
    'aero_avgpool1d': (i32, [vp, vp, i32, i32, vp]),
    'aero_loss_sum': (i32, [vp, vp, i64, C.c_float, i32, dp, i32, dp, C.c_double, vp]),
    'aero_gconv1d_bwd': (i32, [C.POINTER(GconvBwdDesc), vp]),
    'aero_loss_grad': (i32, [vp, vp, i64, C.c_float, C.c_float, i32, vp, fp, vp]),
    'aero_avgpool1d_bwd': (i32, [vp, vp, i32, i32, vp]),
}

EXPORTS = tuple(_PROTOS)


class AeroHipError(RuntimeError):
    pass


class Lib:
    """Loaded C-ABI library with checked calls."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise ImportError(
                f'{path} not found: build the gfx950 kernels first (python -c "import __graft_entry__ as g; '
                f'g.build()").  aero_amd has no CPU or eager fallback.')
        self.path = path
        self.cdll = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as e:
                raise ImportError(f'{path} does not export {name}') from e
            fn.restype = res
            fn.argtypes = args
        self.version = self.cdll.aero_version().decode()
        self.is_emulator = 'emulation' in self.version
        # a device library compiled with packed-fp32 instructions returns wrong FFT results next to other streams' MFMA waves:
        # the build reports `no-packed-fp32` in its version; anything else (an out-of-tree build, AERO_HIP_LIB pointing at an experiment
        # build) is refused unless the experiment says so explicitly
        if not self.is_emulator and 'no-packed-fp32' not in self.version and os.environ.get('AERO_ALLOW_PACKED_FP32') != '1':
            raise ImportError(f'{path} was not built with -packed-fp32-ops off / -DAERO_NO_PACKED_FP32 ({self.version}): rebuild it with '
                              f'__graft_entry__.build() (set AERO_ALLOW_PACKED_FP32=1 only for tools/dbg experiment builds)')

    def check(self, rc, what):
        if rc != 0:
            raise AeroHipError(f'{what} failed ({rc}): {self.cdll.aero_last_error().decode()}')

    def call(self, name, *args):
        self.check(getattr(self.cdll, name)(*args), name)

    def lstm_geometry_in(self, H, in_ch):
        """padded W_ih column count for the fused input projection, or None if (H, in_ch) has no instantiation"""
        kpi = i32(0)
        rc = self.cdll.aero_lstm_geometry_in(H, in_ch, C.byref(kpi))
        return kpi.value if rc == 0 else None

    def lstm_geometry(self, H):
        mp, kp = i32(0), i32(0)
        self.check(self.cdll.aero_lstm_geometry(H, C.byref(mp), C.byref(kp)), 'aero_lstm_geometry')
        return mp.value, kp.value


_cached = {}


def load(path=None):
    # AERO_HIP_LIB: an experiment build of the SAME library (tools/dbg: A/B of compiler flags); never a different implementation
    path = os.path.abspath(path or os.environ.get('AERO_HIP_LIB') or DEFAULT_LIB)
    if path not in _cached:
        _cached[path] = Lib(path)
    return _cached[path]
