"""CPU: every kernel's index logic / masks / epilogues exercised through the CPU emulation of the HIP
subset (tests/emu) against the oracle.  This is a test double for the device -- the product never loads it.
Shapes are small (the emulator runs one fiber per GPU thread)."""
import pytest

import op_cases as oc
from aero_amd import _lib


@pytest.fixture(scope='module')
def emu():
    from emu.build_emu import build
    return _lib.load(build())


DEV = 'cpu'


@pytest.mark.parametrize('geom', [(128, 4, 32, 400), (512, 16, 128, 1000), (512, 64, 512, 1536), (1024, 64, 256, 2003),
                                  (256, 8, 64, 799), (64, 8, 32, 300), (32, 4, 32, 123)])
def test_stft(emu, geom):
    oc.case_stft(emu, DEV, *geom)


@pytest.mark.parametrize('geom', [(1024, 120, 600, 3000), (2048, 240, 1200, 5000), (512, 50, 240, 1777), (2048, 512, 2048, 6000)])
def test_stft_loss_and_metric_geometries(emu, geom):
    """The multi-resolution STFT loss (stft_loss.py:120-123: 1024/120/600, 2048/240/1200, 512/50/240) and LSD (metrics.py:58:
    2048/512) geometries: hop does not divide n_fft, windows shorter than n_fft and not powers of two, n_fft 2048, Nyquist kept."""
    oc.case_stft(emu, DEV, *geom, nyquist=True)


@pytest.mark.parametrize('geom', [(128, 16, 128, 26), (512, 64, 512, 33), (1024, 256, 1024, 9), (256, 32, 252, 40), (64, 16, 64, 30), (32, 8, 32, 17),
                                  (512, 64, 512, 150), (512, 256, 512, 70), (1024, 128, 1024, 75)])      # several 64-hop runs of the ring form (k_stft.h, second form)
def test_istft(emu, geom):
    oc.case_istft(emu, DEV, *geom)


@pytest.mark.parametrize('kw', [
    dict(Cin=2, Cout=48, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=5, T=150),                  # pre_conv, scalar loads
    dict(Cin=48, Cout=5, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=3, T=70, act='relu'),      # FTB conv1, M=5
    dict(Cin=16, Cout=24, kF=8, kT=1, stride=4, padF=2, padT=0, Fin=16, T=130, act='gelu'),    # encoder conv
    dict(Cin=24, Cout=32, kF=8, kT=1, stride=2, padF=3, padT=0, Fin=8, T=40),
    dict(Cin=32, Cout=64, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=4, T=140, split=16, act='glu'),   # dec rewrite
    dict(Cin=64, Cout=128, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=3, T=50, split=32, null0=True),  # first decoder
    dict(Cin=96, Cout=192, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=2, T=33, split=48),              # 48+48 chunks
    dict(Cin=12, Cout=96, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=4, T=45, residual=True),
    dict(Cin=4, Cout=2, kF=4, kT=1, stride=2, padF=1, padT=0, Fin=4, T=20),                    # tiny-model shapes
    # 3x3 specialisation (aero_conv3x3_kernel): BM = 128 tiles, slab reuse across the three time taps
    dict(Cin=96, Cout=128, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=3, T=140, split=48),             # chunk spans both sources
    dict(Cin=64, Cout=256, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=2, T=129, split=32, act='glu'),
    dict(Cin=32, Cout=128, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=1, T=37),                        # single source, one row
    # short-K 1x1 shapes of the path
    dict(Cin=96, Cout=384, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=3, T=50),                 # LSTM projection shape
    dict(Cin=96, Cout=48, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=5, T=77, split=48, act='relu'),   # FTB conv2
    dict(Cin=24, Cout=160, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=2, T=131),                # q|k|v|decay, BM=96 x2
    dict(Cin=192, Cout=96, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=2, T=40, residual=True),  # LSTM linear + skip, KT=6
    dict(Cin=48, Cout=96, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=4, T=45, act='glu'),       # encoder rewrite + GLU
    dict(Cin=128, Cout=304, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=1, T=200, B=3),
    # 256-row / 8-wave tiles (aero_conv_glds8_kernel): KC 64 and KC 32, two sources, GLU, ragged time tile
    dict(Cin=128, Cout=256, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=2, T=137, split=64, B=1),
    dict(Cin=160, Cout=512, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=2, T=70, split=96, act='glu', B=1),
    dict(Cin=1024, Cout=256, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=1, T=90, residual=True, B=2),
    dict(Cin=96, Cout=192, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=3, T=150, split=48, act='glu', B=1),   # 192-row tile
    dict(Cin=64, Cout=384, kF=8, kT=1, stride=2, padF=3, padT=0, Fin=6, T=66, B=1, act='gelu'),
    # software-pipelined ring kernel (aero_conv_ring_kernel): 256x256 / 128x512 / 64x512 tiles, ring of 3-4 K-chunks
    dict(Cin=128, Cout=256, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=3, T=300, split=64, null0=True, B=1),   # 2 t-tiles, NULL source
    dict(Cin=112, Cout=256, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=2, T=70, split=48, B=1, act='glu'),    # chunk spans both sources, ragged Cp
    dict(Cin=1056, Cout=256, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=1, T=61, B=2, act='relu'),             # 33 chunks, one tap
    dict(Cin=96, Cout=384, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=2, T=530, split=48, B=1, act='glu'),    # <1,8,4>: 128 x 512, 2 t-tiles
    dict(Cin=96, Cout=128, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=1, T=40, B=1),                          # one row: only 1 f-tap valid
    dict(Cin=96, Cout=192, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=3, T=501, split=48, B=1, act='glu'),    # <1,8,2>: 64 x 512 (decoder 3)
    dict(Cin=32, Cout=64, kF=8, kT=1, stride=2, padF=3, padT=0, Fin=6, T=66, B=1, act='gelu') if False else
    dict(Cin=128, Cout=64, kF=8, kT=1, stride=2, padF=3, padT=0, Fin=6, T=66, B=1, act='gelu'),              # strided taps on <1,8,2>
    # skinny-M streaming kernel (aero_conv_skinny_kernel): two sources, chunk spanning both, NULL first source
    dict(Cin=80, Cout=7, kF=1, kT=3, stride=1, padF=0, padT=1, Fin=3, T=300, split=24, act='gelu'),
    dict(Cin=64, Cout=16, kF=3, kT=1, stride=1, padF=1, padT=0, Fin=4, T=77, split=32, null0=True),
    dict(Cin=8, Cout=6, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=3, T=70, split=4),                  # scalar slice spans both sources
    dict(Cin=2, Cout=5, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=7, T=300, act='relu'),
    dict(Cin=12, Cout=96, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=4, T=45),                           # 8-byte aligned rows (h16x4 loads)
    dict(Cin=20, Cout=40, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=3, T=50, split=12),
])
def test_conv2d(emu, kw):
    oc.case_conv2d(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Fq=64, T=70), dict(Cin=96, Fq=16, T=130, B=1), dict(Cin=192, Fq=8, T=37), dict(Cin=24, Fq=3, T=20)])
def test_squeeze(emu, kw):
    oc.case_squeeze(emu, DEV, **kw)


@pytest.mark.parametrize('kw', oc.PW_CASES)
def test_pw(emu, kw):
    oc.case_pw(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=2, Cout=5, Fq=40, T=70), dict(Cin=2, Cout=5, Fq=64, T=64, act='none'), dict(Cin=4, Cout=6, Fq=33, T=130, B=1, act='gelu'), dict(Cin=3, Cout=3, Fq=7, T=20)])
def test_conv_tiny(emu, kw):
    oc.case_conv_tiny(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=12, Cout=96, kF=1, kT=1, Fq=3, T=70, per_row=True), dict(Cin=32, Cout=64, kF=1, kT=1, Fq=4, T=140, G=4), dict(Cin=128, Cout=256, kF=3, kT=3, Fq=2, T=130, B=1),
                                # ring kernel epilogue statistics: 4 groups of 64 / 96 rows, ragged second time tile, 192-row tile
                                dict(Cin=128, Cout=256, kF=3, kT=3, Fq=2, T=300, B=1, G=4), dict(Cin=96, Cout=384, kF=3, kT=3, Fq=2, T=70, B=2, G=4)])
def test_conv_stats(emu, kw):
    oc.case_conv_stats(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=12, M=96, Fq=3, T=70), dict(Cc=16, M=96, Fq=2, T=300, B=1), dict(Cc=12, M=96, Fq=2, T=133, pitch=16), dict(Cc=96, M=768, Fq=1, T=130, B=1), dict(Cc=5, M=10, Fq=2, T=20)])
def test_gram_stats(emu, kw):
    oc.case_gram_stats(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Cout=12, k=3, dil=1, R=6, T=131), dict(Cin=16, Cout=4, k=3, dil=2, R=3, T=60),
                                dict(Cin=40, Cout=16, k=9, dil=1, R=2, T=50),
                                dict(Cin=256, Cout=48, k=9, dil=1, R=2, T=150),      # long K on a small grid: 16-row tiles
                                dict(Cin=384, Cout=96, k=3, dil=2, R=2, T=300), dict(Cin=384, Cout=96, k=3, dil=1, R=1, T=130)])   # 96-row ring tile, dilated / plain
def test_conv1d(emu, kw):
    oc.case_conv1d(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=32, Cout=16, K=8, stride=2, Fin=4, T=40), dict(Cin=16, Cout=8, K=8, stride=4, Fin=5, T=33),
                                dict(Cin=16, Cout=2, K=8, stride=4, Fin=6, T=70, f32_affine=True),
                                dict(Cin=96, Cout=2, K=8, stride=4, Fin=21, T=70, f32_affine=True),   # carried-tap kernel, row chunks
                                dict(Cin=96, Cout=2, K=8, stride=4, Fin=5, T=64), dict(Cin=32, Cout=1, K=8, stride=4, Fin=9, T=33, trim=False),
                                dict(Cin=64, Cout=2, K=4, stride=2, Fin=18, T=130, B=3),
                                dict(Cin=8, Cout=4, K=4, stride=2, Fin=3, T=20), dict(Cin=8, Cout=4, K=2, stride=2, Fin=2, T=20),
                                dict(Cin=16, Cout=8, K=8, stride=2, Fin=4, T=30, trim=False),
                                dict(Cin=256, Cout=192, K=8, stride=2, Fin=3, T=70)])     # 192-row 8-wave tile, weight sets
def test_convtr(emu, kw):
    oc.case_convtr(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=32, Cout=16, K=8, stride=2, Fin=4, T=40), dict(Cin=64, Cout=8, K=8, stride=4, Fin=5, T=133, act='gelu'), dict(Cin=256, Cout=96, K=8, stride=2, Fin=3, T=70, trim=False, B=1), dict(Cin=16, Cout=24, K=4, stride=2, Fin=3, T=20)])
def test_convtr_stacked(emu, kw):
    oc.case_convtr_stacked(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Fin=3, T=70), dict(Fin=1, T=260, B=1)])
def test_conv_tail_fused_last_layer(emu, kw):
    oc.case_conv_tail(emu, DEV, **kw)


def test_freq_emb_epilogue(emu):
    oc.case_freq_emb_epilogue(emu, DEV)


@pytest.mark.parametrize('kw', [dict(Cc=16, G=4, Fq=3, T=50, act='gelu'), dict(Cc=32, G=4, Fq=2, T=45, act='glu'),
                                dict(Cc=12, G=1, Fq=4, T=33, act='snake', per_row=True),
                                dict(Cc=24, G=1, Fq=3, T=40, act='glu_ls_res', per_row=True),
                                dict(Cc=8, G=4, Fq=10, T=21, act='gelu', trim=2), dict(Cc=2, G=1, Fq=3, T=20, act='snake', per_row=True),
                                dict(Cc=8, G=4, Fq=6, T=17, act='none', trim=1),
                                dict(Cc=192, G=4, Fq=3, T=70, act='gelu'), dict(Cc=64, G=4, Fq=2, T=130, act='glu')])      # narrow groups: statistics from whole positions
def test_groupnorm(emu, kw):
    oc.case_groupnorm(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(H=4, R=3, T=50), dict(H=8, R=2, T=251), dict(H=48, R=2, T=40), dict(H=24, R=18, T=30),
                                dict(H=8, R=2, T=251, fuse=False), dict(H=48, R=2, T=40, fuse=False), dict(H=96, R=1, T=33)])
def test_blstm(emu, kw):
    oc.case_blstm(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=8, heads=4, R=2, T=70), dict(Cc=48, heads=4, R=1, T=300), dict(Cc=4, heads=4, R=3, T=33),
                                dict(Cc=96, heads=4, R=1, T=501), dict(Cc=16, heads=4, R=1, T=530)])    # folded two-pass form (T <= 512) / streaming form
def test_localstate(emu, kw):
    oc.case_localstate(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Fq=16, Cc=8, T=20), dict(Fq=70, Cc=12, T=11), dict(Fq=4, Cc=4, T=9, B=1), dict(Fq=8, Cc=16, T=33), dict(Fq=13, Cc=8, T=7),
                                dict(Fq=40, Cc=2, T=75), dict(Fq=33, Cc=3, T=21)])    # rows 4-byte aligned only (N = 2 T) / odd N
def test_freqfc(emu, kw):
    oc.case_freqfc(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(H=48, R=3, T=251), dict(H=8, R=5, T=150)])
def test_lstm_bitwise_reproducible_and_row_permutation_invariant(emu, kw):
    oc.case_lstm_bitwise(emu, DEV, **kw)


def test_localstate_streaming_form_for_short_rows():
    """AERO_ATTN_FOLD=0: rows with T <= 512 on the streaming kernel (the default for them is the folded two-pass kernel)."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import op_cases as oc\nfrom aero_amd import _lib\nfrom emu.build_emu import build\n"
            "oc.case_localstate(_lib.load(build()), 'cpu', Cc=48, heads=4, R=1, T=300)\nprint('ok')\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], env={**os.environ, 'AERO_ATTN_FOLD': '0'}, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-1500:]


@pytest.mark.parametrize('kw', [dict(Cc=48, M=48, K=8, stride=4, pad=2, Fq=32, T=139),
                                dict(Cc=16, M=16, K=8, stride=4, pad=2, Fq=64, T=130, act='relu'),
                                dict(Cc=24, M=32, K=3, stride=1, pad=1, Fq=9, T=33, B=1, act='none'),
                                dict(Cc=64, M=64, K=8, stride=4, pad=2, Fq=20, T=128, B=1)])     # Fo = 5: a ragged last row group
def test_enc0_fused(emu, kw):
    oc.case_enc0(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=48, T=139, act='snake'), dict(Cc=16, T=33, Fq=2, B=1, depth=1, act='relu'),
                                dict(Cc=96, T=70, Fq=1, depth=3), dict(Cc=32, T=16, Fq=2, norm=False), dict(Cc=128, T=50, Fq=1, B=1)])
def test_dconv_row(emu, kw):
    oc.case_dconv_row(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=64, Cout=48, k=9, R=2, T=150), dict(Cin=32, Cout=24, k=3, R=1, T=70)])
def test_conv1d_tap_split(emu, kw):
    oc.case_conv1d_split(emu, DEV, **kw)


@pytest.mark.parametrize('geom', [(512, 16, 128, 1000), (256, 8, 64, 799), (512, 16, 100, 2003), (1024, 16, 128, 1700)])
def test_stft_short_window_as_gemm(emu, geom):
    """aero_stft_dft_fwd: windows of <= 128 samples (Aero._spec of the low-rate input) as hi/lo-split fp16 MFMAs against a windowed
    DFT table; same 2e-6 bar as the FFT kernel, same statistics."""
    oc.case_stft(emu, DEV, *geom, dft=True)


# ---- backward: data gradients on the forward kernels (aero_amd/backward.py)
@pytest.mark.parametrize('kw', [dict(Cin=32, Cout=64, kF=3, kT=3, Fr=4, T=40), dict(Cin=16, Cout=32, kF=1, kT=1, Fr=3, T=50)])
def test_dgrad_conv2d(emu, kw):
    oc.case_dgrad_conv2d(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Cout=16, k=3, dil=2, R=3, T=70), dict(Cin=16, Cout=96, k=1, dil=1, R=2, T=40)])
def test_dgrad_conv1d(emu, kw):
    oc.case_dgrad_conv1d(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=16, Cout=32, K=8, stride=4, Fin=16, T=40), dict(Cin=32, Cout=32, K=8, stride=2, Fin=8, T=33)])
def test_dgrad_conv_fstride(emu, kw):
    oc.case_dgrad_conv_fstride(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=32, Cout=16, K=8, stride=4, Fin=4, T=40), dict(Cin=64, Cout=32, K=8, stride=2, Fin=3, T=33)])
def test_dgrad_convtr(emu, kw):
    oc.case_dgrad_convtr(emu, DEV, **kw)


# ---- backward: weight gradients and GroupNorm + activation backward (k_bwd.h)
@pytest.mark.parametrize('kw', [dict(Cin=32, Cout=64, kF=3, kT=3, Fr=3, T=70), dict(Cin=136, Cout=144, kF=1, kT=1, Fr=2, T=50),
                                dict(Cin=8, Cout=16, kF=3, kT=1, Fr=5, T=33, B=3),
                                dict(Cin=200, Cout=264, kF=3, kT=1, Fr=3, T=70)])       # 256 x 256 tile, ragged in both directions
def test_wgrad_conv2d(emu, kw):
    oc.case_wgrad_conv2d(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Cout=16, k=3, dil=2, R=5, T=70), dict(Cin=16, Cout=96, k=1, dil=1, R=2, T=40)])
def test_wgrad_conv1d(emu, kw):
    oc.case_wgrad_conv1d(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=16, Cout=32, K=8, stride=4, Fin=16, T=40)])
def test_wgrad_conv_fstride(emu, kw):
    oc.case_wgrad_conv_fstride(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=32, Cout=16, K=8, stride=4, Fin=4, T=40), dict(Cin=64, Cout=32, K=8, stride=2, Fin=3, T=33)])
def test_wgrad_convtr(emu, kw):
    oc.case_wgrad_convtr(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(C_=48, G=4, per_row=0, act='gelu', Fr=4, T=50), dict(C_=96, G=1, per_row=1, act='glu', Fr=3, T=37, layer_scale=True),
                                dict(C_=32, G=1, per_row=1, act='gelu', Fr=2, T=20), dict(C_=64, G=4, per_row=0, act='glu', Fr=3, T=33),
                                dict(C_=48, G=4, per_row=0, act='none', Fr=2, T=300),
                                dict(C_=32, G=1, per_row=1, act='glu', Fr=650, T=8)])     # > 1024 work items: blocks loop over several
def test_norm_bwd(emu, kw):
    oc.case_norm_bwd(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(kind=('conv2d', 1, 1), Cin=16, Cout=32, G=4, act='glu', Fin=4, T=40),
                                dict(kind=('fstride', 4), Cin=16, Cout=32, G=4, act='gelu', Fin=16, T=33),
                                dict(kind=('convtr', 4), Cin=32, Cout=32, G=4, act='gelu', Fin=4, T=33)])
def test_block_autograd(emu, kw):
    oc.case_block_autograd(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=48, k=3, dil=1, Fr=3, T=40), dict(Cc=96, k=3, dil=2, Fr=2, T=33)])
def test_dconv_autograd(emu, kw):
    oc.case_dconv_autograd(emu, DEV, **kw)


def test_train_steps_match_torch(emu):
    oc.case_train_steps(emu, DEV)


@pytest.mark.parametrize('kw', [dict(kind=('fstride', 4), Cin=16, Cout=32, G=0, Fin=16, T=33),
                                dict(kind=('fstride', 2), Cin=16, Cout=32, G=4, Fin=8, T=40),
                                dict(kind=('convtr', 2), Cin=32, Cout=32, G=4, Fin=4, T=33)])
def test_block_autograd_snake(emu, kw):
    oc.case_block_autograd_snake(emu, DEV, **kw)


def test_decoder_autograd(emu):
    oc.case_decoder_autograd(emu, DEV)


@pytest.mark.parametrize('kw', [dict(C_=16, act='relu', Fr=5, T=40), dict(C_=8, act='none', Fr=3, T=33)])
def test_batchnorm_bwd(emu, kw):
    oc.case_batchnorm_bwd(emu, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(nfft=512, hop=64, T=24), dict(nfft=64, hop=16, T=13, crop=0)])
def test_istft_bwd(emu, kw):
    oc.case_istft_bwd(emu, DEV, **kw)


# ---- the rest of the training step (csrc/k_train.h) ------------------------------------------------------------
@pytest.mark.parametrize('a', [(16, 8, 21), (70, 16, 13)])
def test_freqfc_wgrad(emu, a):
    oc.case_freqfc_wgrad(emu, DEV, *a)


def test_ftb_gate_bwd_sum_bt_scale_cast(emu):
    oc.case_ftb_gate_bwd(emu, DEV, 5, 8, 21)
    oc.case_sum_bt(emu, DEV, 4, 16, 37)
    oc.case_sum_bt(emu, DEV, 3, 48, 20)
    oc.case_scale_cast(emu, DEV)


@pytest.mark.parametrize('a', [(2, 251, 8), (1, 430, 4)])
def test_frames_op(emu, a):
    oc.case_frames_op(emu, DEV, *a)


@pytest.mark.parametrize('geom', [(512, 50, 240, 1777), (1024, 120, 600, 3000), (2048, 240, 1200, 5000)])
def test_stft_loss_value_and_gradient(emu, geom):
    oc.case_stft_loss(emu, DEV, *geom)


@pytest.mark.parametrize('a', [(16, 4, 2, 77), (48, 4, 1, 150), (96, 4, 1, 70)])
def test_localstate_bwd(emu, a):
    oc.case_localstate_bwd(emu, DEV, *a)


@pytest.mark.parametrize('kw', [dict(H=8, nseq=5, W=12), dict(H=16, nseq=20, W=9, in_ch=32), dict(H=48, nseq=3, W=7),
                                dict(H=8, nseq=6, W=200, framed_T=251)])
def test_lstm_bwd(emu, kw):
    oc.case_lstm_bwd(emu, DEV, **kw)


@pytest.mark.parametrize('a', [(16, 8, 40), (32, 16, 33)])
def test_ftb_autograd(emu, a):
    oc.case_ftb_autograd(emu, DEV, *a)


def test_blstm_without_the_discarded_last_frame(emu):
    """round 6: T = 501 cuts 6 frames of which the last is stitched away whole -- not computing it changes no bit (8 lengths, 6 with a drop)"""
    assert oc.case_blstm_frame_skip(emu, DEV) == 6


@pytest.mark.parametrize('tag', ['blstm', 'localstate', 'snake', 'ftb', 'dconv', 'henc', 'hdec'])
def test_reference_module_vectors(emu, tag):
    """the REFERENCE's own module outputs (tests/golden/modules.npz) reproduced by the kernels: <= 1e-3, the north-star bar"""
    errs = oc.case_module_golden(emu, DEV, tag)
    assert errs and max(errs.values()) < 1e-3, errs


def test_stft_dft_blocks_walk_several_time_tiles():
    """AERO_STFT_DFT_BLOCKS=1: ONE block per (signal, table quarter) walks all 128-frame tiles with its table slice resident -- the
    form a full batch takes on the device (few signals get one block per tile: that path is what the other DFT cases run)."""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import op_cases as oc\nfrom aero_amd import _lib\nfrom emu.build_emu import build\n"
            "oc.case_stft(_lib.load(build()), 'cpu', 512, 16, 128, 5000, B=2, dft=True)\nprint('ok')\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], env={**os.environ, 'AERO_STFT_DFT_BLOCKS': '1'}, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-1500:]


def test_localstate_bwd_valu_form():
    """AERO_ATTN_BWD_VALU=1: the fp32 VALU form of the LocalState backward (the default is the MFMA form)"""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import op_cases as oc\nfrom aero_amd import _lib\nfrom emu.build_emu import build\n"
            "oc.case_localstate_bwd(_lib.load(build()), 'cpu', 48, 4, 1, 150)\nprint('ok')\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], env={**os.environ, 'AERO_ATTN_BWD_VALU': '1'}, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-1500:]


def test_bn_running_update(emu):
    oc.case_bn_running_update(emu, DEV)
