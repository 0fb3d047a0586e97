"""GPU (-m gpu): op-level parity of every C-ABI entry point on the MI355X against the oracle, at the
shapes of aero_4-16_512_64 (reduced batch) -- same cases as tests/test_emu_ops.py, real kernels."""
import pytest
import torch

import op_cases as oc

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def lib():
    from aero_amd import _lib
    assert torch.cuda.is_available()
    lib = _lib.load()                       # aero_amd/libaero_hip.so -- fails loudly if missing
    assert not lib.is_emulator
    return lib


@pytest.mark.parametrize('geom', [(512, 16, 128, 8000), (512, 64, 512, 32000), (1024, 64, 256, 24000), (128, 4, 32, 999),
                                  (512, 16, 128, 7999), (64, 8, 32, 300), (32, 4, 32, 123), (256, 8, 64, 799)])
def test_stft(lib, geom):
    oc.case_stft(lib, DEV, *geom, B=3)


@pytest.mark.parametrize('geom', [(1024, 120, 600, 32000), (2048, 240, 1200, 32000), (512, 50, 240, 32000), (2048, 512, 2048, 32000), (2048, 512, 2048, 5000)])
def test_stft_loss_and_metric_geometries(lib, geom):
    """stft_loss.py:120-123 and metrics.py:58 geometries on the HIP STFT (<= 2e-6 vs the oracle)."""
    oc.case_stft(lib, DEV, *geom, B=4, nyquist=True)


@pytest.mark.parametrize('geom', [(512, 64, 512, 501), (1024, 256, 1024, 376), (128, 16, 128, 101), (512, 128, 512, 251), (64, 16, 64, 30), (32, 8, 32, 17), (256, 32, 252, 40)])
def test_istft(lib, geom):
    oc.case_istft(lib, DEV, *geom, B=3)


@pytest.mark.parametrize('kw', [
    dict(Cin=2, Cout=48, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=256, T=501),
    dict(Cin=48, Cout=5, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=64, T=501, act='relu'),
    dict(Cin=48, Cout=48, kF=8, kT=1, stride=4, padF=2, padT=0, Fin=256, T=501, act='gelu'),
    dict(Cin=96, Cout=192, kF=8, kT=1, stride=2, padF=3, padT=0, Fin=16, T=501),
    dict(Cin=192, Cout=384, kF=8, kT=1, stride=2, padF=3, padT=0, Fin=8, T=501),
    dict(Cin=768, Cout=1536, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=4, T=501, split=384, null0=True, B=1),
    dict(Cin=384, Cout=768, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=8, T=501, split=192, B=1),
    dict(Cin=192, Cout=384, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=16, T=501, split=96, act='glu', B=1),
    dict(Cin=96, Cout=192, kF=3, kT=3, stride=1, padF=1, padT=1, Fin=64, T=501, split=48, act='glu', B=1),
    dict(Cin=48, Cout=96, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=64, T=501, act='glu'),
    dict(Cin=12, Cout=96, kF=1, kT=1, stride=1, padF=0, padT=0, Fin=64, T=501, residual=True),
    dict(Cin=4, Cout=2, kF=4, kT=1, stride=2, padF=1, padT=0, Fin=4, T=20),
])
def test_conv2d(lib, kw):
    oc.case_conv2d(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Fq=64, T=70), dict(Cin=96, Fq=16, T=130, B=1), dict(Cin=192, Fq=8, T=37), dict(Cin=24, Fq=3, T=20)])
def test_squeeze(lib, kw):
    oc.case_squeeze(lib, DEV, **kw)


@pytest.mark.parametrize('kw', oc.PW_CASES)
def test_pw(lib, kw):
    oc.case_pw(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=2, Cout=5, Fq=256, T=501, B=4), dict(Cin=4, Cout=6, Fq=33, T=130, B=1, act='gelu'), dict(Cin=3, Cout=3, Fq=7, T=20)])
def test_conv_tiny(lib, kw):
    oc.case_conv_tiny(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=12, Cout=96, kF=1, kT=1, Fq=64, T=501, per_row=True), dict(Cin=96, Cout=192, kF=1, kT=1, Fq=16, T=501, G=4), dict(Cin=384, Cout=768, kF=3, kT=3, Fq=8, T=501, B=1), dict(Cin=96, Cout=384, kF=3, kT=3, Fq=16, T=501, B=1, G=4), dict(Cin=48, Cout=384, kF=1, kT=1, Fq=8, T=501, per_row=True),
                                dict(Cin=384, Cout=1536, kF=3, kT=3, Fq=4, T=501, B=2, G=4)])
def test_conv_stats(lib, kw):
    oc.case_conv_stats(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=16, M=96, Fq=64, T=501), dict(Cc=24, M=192, Fq=16, T=501), dict(Cc=48, M=384, Fq=8, T=501), dict(Cc=96, M=768, Fq=4, T=501), dict(Cc=12, M=96, Fq=8, T=501, pitch=16)])
def test_gram_stats(lib, kw):
    oc.case_gram_stats(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Cout=12, k=3, dil=1, R=128, T=501), dict(Cin=384, Cout=96, k=3, dil=2, R=8, T=501),
                                dict(Cin=512, Cout=48, k=9, dil=1, R=2, T=501)])
def test_conv1d(lib, kw):
    oc.case_conv1d(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=768, Cout=192, K=8, stride=2, Fin=4, T=501), dict(Cin=192, Cout=48, K=8, stride=4, Fin=16, T=501),
                                dict(Cin=96, Cout=2, K=8, stride=4, Fin=64, T=501, f32_affine=True),
                                dict(Cin=96, Cout=2, K=8, stride=4, Fin=21, T=70, f32_affine=True),   # carried-tap kernel, row chunks
                                dict(Cin=96, Cout=2, K=8, stride=4, Fin=5, T=64), dict(Cin=32, Cout=1, K=8, stride=4, Fin=9, T=33, trim=False),
                                dict(Cin=64, Cout=2, K=4, stride=2, Fin=18, T=130, B=3),
                                dict(Cin=8, Cout=4, K=2, stride=2, Fin=2, T=20), dict(Cin=384, Cout=96, K=8, stride=2, Fin=8, T=501, trim=False)])
def test_convtr(lib, kw):
    oc.case_convtr(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=768, Cout=192, K=8, stride=2, Fin=4, T=501), dict(Cin=384, Cout=96, K=8, stride=2, Fin=8, T=501, trim=False), dict(Cin=192, Cout=48, K=8, stride=4, Fin=16, T=501, act='gelu'), dict(Cin=16, Cout=24, K=4, stride=2, Fin=3, T=20)])
def test_convtr_stacked(lib, kw):
    oc.case_convtr_stacked(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Fin=64, T=501), dict(Fin=128, T=376, B=1), dict(Fin=3, T=70)])
def test_conv_tail_fused_last_layer(lib, kw):
    oc.case_conv_tail(lib, DEV, **kw)


def test_freq_emb_epilogue(lib):
    oc.case_freq_emb_epilogue(lib, DEV)


@pytest.mark.parametrize('kw', [dict(Cc=192, G=4, Fq=8, T=501, act='gelu'), dict(Cc=768, G=4, Fq=4, T=501, act='glu'),
                                dict(Cc=12, G=1, Fq=64, T=501, act='snake', per_row=True),
                                dict(Cc=768, G=1, Fq=4, T=501, act='glu_ls_res', per_row=True),
                                dict(Cc=96, G=4, Fq=22, T=501, act='gelu', trim=3), dict(Cc=2, G=1, Fq=3, T=20, act='snake', per_row=True)])
def test_groupnorm(lib, kw):
    oc.case_groupnorm(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(H=48, R=16, T=501), dict(H=96, R=8, T=501), dict(H=8, R=3, T=251), dict(H=48, R=5, T=150),
                                dict(H=48, R=16, T=501, fuse=False), dict(H=96, R=8, T=501, fuse=False)])
def test_blstm(lib, kw):
    oc.case_blstm(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(H=48, R=512), dict(H=96, R=256), dict(H=48, R=37), dict(H=96, R=5, T=150), dict(H=8, R=7, T=251)])
def test_lstm_bitwise_reproducible_and_row_permutation_invariant(lib, kw):
    oc.case_lstm_bitwise(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=48, heads=4, R=16, T=501), dict(Cc=96, heads=4, R=8, T=501), dict(Cc=4, heads=4, R=3, T=33),
                                dict(Cc=48, heads=4, R=4, T=1724), dict(Cc=96, heads=4, R=2, T=376)])   # streaming form (10-s segments), config 4's T
def test_localstate(lib, kw):
    oc.case_localstate(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Fq=256, Cc=48, T=501, B=1), dict(Fq=64, Cc=48, T=501), dict(Fq=8, Cc=192, T=501), dict(Fq=16, Cc=96, T=501), dict(Fq=4, Cc=4, T=9, B=1),
                                dict(Fq=256, Cc=2, T=501), dict(Fq=33, Cc=3, T=21)])    # encoder 0's (re, im) rows: 4-byte aligned only
def test_freqfc(lib, kw):
    oc.case_freqfc(lib, DEV, **kw)


def test_spectral_losses_on_the_hip_stft():
    """aero_amd.losses (stft_loss.py:84-161, metrics.py:58-70 restated on aero_stft_fwd) against torch.stft on the host."""
    import torch.nn.functional as F
    from aero_amd import evaluate as ev, losses
    g = torch.Generator().manual_seed(3)
    y = torch.randn(3, 16000, generator=g)
    x = y + 0.3 * torch.randn(3, 16000, generator=g)

    def mag(v, n, h, w):
        z = torch.stft(v, n, h, w, torch.hann_window(w), return_complex=True)
        return torch.sqrt(torch.clamp(z.real ** 2 + z.imag ** 2, min=1e-7)).transpose(2, 1)
    sc_ref = mag_ref = 0.0
    for n, h, w in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
        xm, ym = mag(x, n, h, w), mag(y, n, h, w)
        got = losses.stft_magnitude(x.cuda(), n, h, w).cpu()
        assert got.shape == xm.shape and oc.rel_l2(got, xm) < 1e-5
        sc_ref = sc_ref + torch.norm(ym - xm, p='fro') / torch.norm(ym, p='fro')
        mag_ref = mag_ref + F.l1_loss(torch.log(ym), torch.log(xm))
    sc, mg = losses.MultiResolutionSTFTLoss()(x.cuda(), y.cuda())
    assert abs(float(sc) - 0.1 * float(sc_ref) / 3) < 1e-5 and abs(float(mg) - 0.1 * float(mag_ref) / 3) < 1e-5
    want = float(ev.lsd(y, x))                                   # host path (torch.stft)
    assert abs(float(ev.lsd(y.cuda(), x.cuda())) - want) < 2e-4 * want
    assert float(ev.lsd(y.cuda(), y.cuda())) == 0.0


@pytest.mark.parametrize('kw', [dict(Cc=48, M=48, K=8, stride=4, pad=2, Fq=256, T=501, B=2), dict(Cc=48, M=48, K=8, stride=4, pad=2, Fq=32, T=139),
                                dict(Cc=16, M=16, K=8, stride=4, pad=2, Fq=64, T=130, act='relu'),
                                dict(Cc=24, M=32, K=3, stride=1, pad=1, Fq=9, T=33, B=1, act='none'),
                                dict(Cc=64, M=64, K=8, stride=4, pad=2, Fq=20, T=128, B=1)])     # Fo = 5: a ragged last row group
def test_enc0_fused(lib, kw):
    oc.case_enc0(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=48, T=501, Fq=64, B=2), dict(Cc=96, T=501, Fq=16, B=2, act='snake'), dict(Cc=48, T=139, act='snake'), dict(Cc=16, T=33, Fq=2, B=1, depth=1, act='relu'),
                                dict(Cc=96, T=70, Fq=1, depth=3), dict(Cc=32, T=16, Fq=2, norm=False), dict(Cc=128, T=50, Fq=1, B=1)])
def test_dconv_row(lib, kw):
    oc.case_dconv_row(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=1280, Cout=48, k=9, R=8, T=501), dict(Cin=64, Cout=48, k=9, R=2, T=150), dict(Cin=32, Cout=24, k=3, R=1, T=70)])
def test_conv1d_tap_split(lib, kw):
    oc.case_conv1d_split(lib, DEV, **kw)


@pytest.mark.parametrize('geom', [(512, 16, 128, 8000), (512, 16, 128, 24001), (512, 16, 128, 1000), (256, 8, 64, 799), (512, 16, 100, 2003), (1024, 16, 128, 1700)])
def test_stft_short_window_as_gemm(lib, geom):
    """aero_stft_dft_fwd: windows of <= 128 samples (Aero._spec of the low-rate input) as hi/lo-split fp16 MFMAs against a windowed
    DFT table; same 2e-6 bar as the FFT kernel, same statistics."""
    oc.case_stft(lib, DEV, *geom, dft=True)


# ---- backward (aero_amd/backward.py, k_bwd.h): data gradients on the forward kernels, weight gradients, GroupNorm backward
@pytest.mark.parametrize('kw', [dict(Cin=192, Cout=384, kF=3, kT=3, Fr=4, T=501), dict(Cin=48, Cout=96, kF=1, kT=1, Fr=8, T=501)])
def test_dgrad_conv2d(lib, kw):
    oc.case_dgrad_conv2d(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=192, Cout=48, k=3, dil=2, R=16, T=501), dict(Cin=48, Cout=384, k=1, dil=1, R=8, T=501)])
def test_dgrad_conv1d(lib, kw):
    oc.case_dgrad_conv1d(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Cout=96, K=8, stride=4, Fin=64, T=501), dict(Cin=192, Cout=384, K=8, stride=2, Fin=8, T=501)])
def test_dgrad_conv_fstride(lib, kw):
    oc.case_dgrad_conv_fstride(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=192, Cout=96, K=8, stride=4, Fin=4, T=501), dict(Cin=768, Cout=192, K=8, stride=2, Fin=4, T=501)])
def test_dgrad_convtr(lib, kw):
    oc.case_dgrad_convtr(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=192, Cout=384, kF=3, kT=3, Fr=4, T=501), dict(Cin=136, Cout=144, kF=1, kT=1, Fr=2, T=50),
                                dict(Cin=8, Cout=16, kF=3, kT=1, Fr=5, T=33, B=3),
                                dict(Cin=200, Cout=264, kF=3, kT=1, Fr=3, T=70),        # 256 x 256 tile, ragged in both directions
                                dict(Cin=768, Cout=1536, kF=3, kT=3, Fr=4, T=250, B=1)])  # first decoder's rewrite conv
def test_wgrad_conv2d(lib, kw):
    oc.case_wgrad_conv2d(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=192, Cout=48, k=3, dil=2, R=16, T=501), dict(Cin=16, Cout=96, k=1, dil=1, R=2, T=40)])
def test_wgrad_conv1d(lib, kw):
    oc.case_wgrad_conv1d(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=48, Cout=96, K=8, stride=4, Fin=64, T=501)])
def test_wgrad_conv_fstride(lib, kw):
    oc.case_wgrad_conv_fstride(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cin=192, Cout=96, K=8, stride=4, Fin=4, T=501), dict(Cin=64, Cout=32, K=8, stride=2, Fin=3, T=33)])
def test_wgrad_convtr(lib, kw):
    oc.case_wgrad_convtr(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(C_=48, G=4, per_row=0, act='gelu', Fr=64, T=501), dict(C_=96, G=1, per_row=1, act='glu', Fr=16, T=501, layer_scale=True),
                                dict(C_=384, G=4, per_row=0, act='glu', Fr=4, T=501), dict(C_=48, G=4, per_row=0, act='none', Fr=2, T=300)])
def test_norm_bwd(lib, kw):
    oc.case_norm_bwd(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(kind=('conv2d', 1, 1), Cin=96, Cout=192, G=4, act='glu', Fin=4, T=501),
                                dict(kind=('fstride', 4), Cin=48, Cout=96, G=4, act='gelu', Fin=16, T=501),
                                dict(kind=('convtr', 4), Cin=96, Cout=48, G=4, act='gelu', Fin=4, T=501)])
def test_block_autograd(lib, kw):
    oc.case_block_autograd(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(Cc=48, k=3, dil=1, Fr=16, T=501), dict(Cc=96, k=3, dil=2, Fr=8, T=501)])
def test_dconv_autograd(lib, kw):
    oc.case_dconv_autograd(lib, DEV, **kw)


def test_train_steps_match_torch(lib):
    oc.case_train_steps(lib, DEV)


@pytest.mark.parametrize('kw', [dict(kind=('fstride', 4), Cin=48, Cout=96, G=0, Fin=16, T=501),
                                dict(kind=('fstride', 2), Cin=96, Cout=192, G=4, Fin=8, T=501),
                                dict(kind=('convtr', 2), Cin=192, Cout=96, G=4, Fin=4, T=501)])
def test_block_autograd_snake(lib, kw):
    oc.case_block_autograd_snake(lib, DEV, **kw)


def test_decoder_autograd(lib):
    oc.case_decoder_autograd(lib, DEV, C=96, Fin=8, T=501)


@pytest.mark.parametrize('kw', [dict(C_=48, act='relu', Fr=64, T=501), dict(C_=8, act='none', Fr=3, T=501)])
def test_batchnorm_bwd(lib, kw):
    oc.case_batchnorm_bwd(lib, DEV, **kw)


@pytest.mark.parametrize('kw', [dict(nfft=512, hop=64, T=501, B=8), dict(nfft=64, hop=16, T=13, crop=0)])
def test_istft_bwd(lib, kw):
    oc.case_istft_bwd(lib, DEV, **kw)


# ---- the rest of the training step (csrc/k_train.h) at the flagship / config-5 shapes ---------------------------
@pytest.mark.parametrize('a', [(256, 48, 501), (64, 96, 120), (8, 384, 501)])
def test_freqfc_wgrad(lib, a):
    oc.case_freqfc_wgrad(lib, DEV, *a)


def test_ftb_gate_bwd_sum_bt_scale_cast(lib):
    oc.case_ftb_gate_bwd(lib, DEV, 64, 96, 501)
    oc.case_sum_bt(lib, DEV, 64, 48, 501)
    oc.case_scale_cast(lib, DEV, n=100000)


@pytest.mark.parametrize('a', [(16, 1724, 48), (8, 501, 96), (3, 251, 8)])
def test_frames_op(lib, a):
    oc.case_frames_op(lib, DEV, *a)


@pytest.mark.parametrize('geom', [(512, 50, 240, 32000), (1024, 120, 600, 32000), (2048, 240, 1200, 32000), (2048, 240, 1200, 441000)])
def test_stft_loss_value_and_gradient(lib, geom):
    oc.case_stft_loss(lib, DEV, *geom)


@pytest.mark.parametrize('a', [(48, 4, 4, 501), (96, 4, 2, 501), (48, 4, 2, 1724), (16, 4, 2, 77)])
def test_localstate_bwd(lib, a):
    oc.case_localstate_bwd(lib, DEV, *a)


@pytest.mark.parametrize('kw', [dict(H=48, nseq=96, W=200), dict(H=96, nseq=48, W=200, in_ch=192), dict(H=48, nseq=36, W=200, framed_T=1724),
                                dict(H=96, nseq=24, W=200, framed_T=501), dict(H=16, nseq=20, W=9, in_ch=32)])
def test_lstm_bwd(lib, kw):
    oc.case_lstm_bwd(lib, DEV, **kw)


@pytest.mark.parametrize('a', [(48, 64, 501), (96, 16, 501), (384, 4, 300)])
def test_ftb_autograd(lib, a):
    oc.case_ftb_autograd(lib, DEV, *a)


def test_blstm_without_the_discarded_last_frame(lib):
    assert oc.case_blstm_frame_skip(lib, DEV, H=48, R=5) == 6


@pytest.mark.parametrize('tag', ['blstm', 'localstate', 'snake', 'ftb', 'dconv', 'henc', 'hdec'])
def test_reference_module_vectors(lib, tag):
    """the REFERENCE's own module outputs (tests/golden/modules.npz: BLSTM framed / unframed, LocalState, FTB eval + train, Snake,
    DConv with BLSTM + LocalState, HEncLayer, HDecLayer) reproduced by the HIP kernels: <= 1e-3 (VERDICT r2 missing #5)"""
    errs = oc.case_module_golden(lib, DEV, tag)
    assert errs and max(errs.values()) < 1e-3, errs


def test_bn_running_update(lib):
    oc.case_bn_running_update(lib, DEV)
