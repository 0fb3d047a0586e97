"""Weight packing for the gfx950 kernels (host plumbing; runs once per set of weights).

Turns the reference-layout fp32 parameters (conv [Cout,Cin,kF,kT], conv-transpose
[Cin,Cout,kF,1], LSTM [4H,in] with gate order i,f,g,o -- SURVEY 8b) into the fp16, K-contiguous,
zero-padded images that include/aero_hip.h documents.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib


@dataclass
class ConvSpec:
    weight: torch.Tensor                # fp16 [nwset, Mpad, ntaps*Cp]
    bias: Optional[torch.Tensor]        # fp32 [M]
    M: int
    C0: int
    C1: int
    df: List[int]
    dt: List[int]
    transposed: int = 0
    fstride: int = 1
    act: int = _lib.ACT_NONE
    extra: dict = field(default_factory=dict)
    weight_tiled: Optional[torch.Tensor] = None     # fp16 [nwset, M/bm, Ktot/32, bm*32]: aero_hip.h "weight_tiled"
    tiled_bm: int = 0

    @property
    def Mout(self):
        return self.M // 2 if self.act == _lib.ACT_GLU else self.M


def _round_up(a, b):
    return (a + b - 1) // b * b


def glu_interleave(t):
    """Reorder rows (a_0..a_{n-1}, b_0..b_{n-1}) -> (a_0, b_0, a_1, b_1, ...) so that one MFMA
    accumulator quad holds both halves of two GLU outputs (k_conv.h epilogue)."""
    n = t.shape[0] // 2
    idx = torch.stack([torch.arange(n), torch.arange(n) + n], 1).reshape(-1)
    return t[idx.to(t.device)]


def make_conv_spec(w_taps, bias, C0, C1, df, dt, device, transposed=0, fstride=1, act=_lib.ACT_NONE):
    """w_taps fp32 [nwset, M, ntaps, C0+C1] -> ConvSpec with the padded fp16 image."""
    nw, M, nt, Ct = w_taps.shape
    assert Ct == C0 + C1 and nt == len(df) == len(dt) and 1 <= nt <= 9
    if act == _lib.ACT_GLU:
        w_taps = torch.stack([glu_interleave(w_taps[i]) for i in range(nw)])
        if bias is not None:
            bias = glu_interleave(bias)
    Cp = _round_up(Ct, 32)
    Mpad = _round_up(M, 128)
    img = torch.zeros(nw, Mpad, nt, Cp, dtype=torch.float32, device=w_taps.device)     # (packed where the weights live: the
    img[:, :M, :, :Ct] = w_taps                                                         #  training step re-packs on the device)
    spec = ConvSpec(weight=img.reshape(nw, Mpad, nt * Cp).to(device=device, dtype=torch.float16).contiguous(),
                    bias=None if bias is None else bias.detach().float().to(device).contiguous(),
                    M=M, C0=C0, C1=C1, df=list(df), dt=list(dt), transposed=transposed, fstride=fstride, act=act)
    bm = ring_bm(M, nt * Cp)
    if bm:
        spec.weight_tiled = tile_weights(img.reshape(nw, Mpad, nt * Cp)[:, :M], bm).to(device=device, dtype=torch.float16).contiguous()
        spec.tiled_bm = bm
    return spec


class PwSpec:
    """a pointwise conv in the form aero_pw_fwd takes (k_pw.h)"""
    def __init__(self, wimg, bias, M, C, act):
        self.wimg, self.bias, self.M, self.C, self.act = wimg, bias, M, C, act


def pw_image(w, rows, ks=None):
    """w fp32/fp16 [M, C] (rows already GLU-interleaved where GLU follows) -> the fp16 image aero_pw_fwd reads (include/aero_hip.h):
    [chunk][2][GW][4][KS][64 lanes][8], rows = 128 * GW rows per chunk; lane l of tile j holds logical row 16 * ((l & 15) >> 2) + 4 j +
    (l & 3) of its 64-row group and the k-octet l >> 4: the permutation that leaves a lane's accumulators consecutive rows."""
    M, C = w.shape
    gw = rows // 128
    ks = (C + 31) // 32 if ks is None else ks
    nchunk = (M + rows - 1) // rows
    wp = torch.zeros(nchunk * rows, ks * 32, dtype=torch.float32, device=w.device)
    wp[:M, :C] = w.float()
    lane = torch.arange(64, device=w.device)
    prow = lane & 15                                              # physical tile row of the lane's A fragment
    koct = lane >> 4
    j = torch.arange(4, device=w.device)
    # logical row inside the 64-row group for (j, lane)
    r64 = (16 * (prow >> 2))[None, :] + 4 * j[:, None] + (prow & 3)[None, :]                  # [4, 64]
    grp = torch.arange(nchunk * 2 * gw, device=w.device)                                      # (chunk, wm, g) flattened = 64-row group index
    rows_idx = grp[:, None, None] * 64 + r64[None]                                            # [G, 4, 64]
    kk = torch.arange(ks, device=w.device)[:, None, None] * 32 + (koct * 8)[None, :, None] + torch.arange(8, device=w.device)[None, None, :]   # [KS, 64, 8]
    img = wp[rows_idx[:, :, None, :, None], kk[None, None]]                                   # [G, 4, KS, 64, 8]
    return img.to(torch.float16).contiguous()


class SqueezeSpec:
    def __init__(self, wimg, bias, M, C, act):
        self.wimg, self.bias, self.M, self.C, self.act = wimg, bias, M, C, act


def make_squeeze_spec(w, bias, act, device):
    """w [M <= 8, C] (BatchNorm already folded) -> the fragment image of aero_squeeze_fwd: [KS][64 lanes][8]; None if not served"""
    M, C = w.shape
    if M > 8 or C % 8 or C < 8 or C > 192:
        return None
    ks = (C + 31) // 32
    wp = torch.zeros(16, ks * 32, dtype=torch.float32)
    wp[:M, :C] = w.detach().float().cpu()
    lane = torch.arange(64)
    kk = torch.arange(ks)[:, None, None] * 32 + ((lane >> 4) * 8)[None, :, None] + torch.arange(8)[None, None, :]
    img = wp[(lane & 15)[None, :, None], kk]                       # [KS, 64, 8]
    return SqueezeSpec(img.to(device=device, dtype=torch.float16).contiguous(), None if bias is None else bias.detach().float().to(device).contiguous(), M, C, act)


def make_pw_spec(w, bias, act, lib, device, max_c=96):
    """w [M, C] fp32 in the reference's row order; GLU: rows / bias interleaved (value, gate) as make_conv_spec does.  None if the
    geometry is not served by the streaming pointwise kernel (C <= 96, NONE / RELU / GLU: the weights-in-LDS form for wider inputs was
    15-80% SLOWER than the LDS-tiled conv at the model's widths, profiles/r04_pw_wlds_ab.txt, and is no longer built)."""
    M, C = w.shape
    rows = int(lib.cdll.aero_pw_rows(C, M)) if C <= max_c else 0
    if not rows or act not in (_lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_GLU):
        return None
    if act == _lib.ACT_GLU:
        w = glu_interleave(w)
        bias = None if bias is None else glu_interleave(bias)
    return PwSpec(pw_image(w.to(device), rows, int(lib.cdll.aero_pw_ksteps(C))), None if bias is None else bias.detach().float().to(device).contiguous(), M, C, act)


def ring_bm(M, Ktot):
    """tile height of the software-pipelined kernel for this contraction (0: not taken) -- asks the library"""
    try:
        return int(_lib.load().cdll.aero_conv_ring_bm(M, Ktot))
    except (ImportError, OSError):
        return 0


def tile_weights(w, bm):
    """[nw, M, Ktot] -> [nw, M/bm, Ktot/32, bm*32] in the kernel's LDS tile order (include/aero_hip.h, `weight_tiled`):
    16-byte unit (row, q) of a tile holds channels 8*(q ^ ((-(row >> 2)) & 3)) .. +8 of the 32-channel chunk."""
    nw, M, K = w.shape
    assert M % bm == 0 and K % 32 == 0
    t = w.reshape(nw, M // bm, bm, K // 32, 4, 8).permute(0, 1, 3, 2, 4, 5)          # [nw, mt, kc, row, q_src, 8]
    row = torch.arange(bm, device=w.device)
    q = torch.arange(4, device=w.device)
    src = q[None, :] ^ ((-(row[:, None] >> 2)) & 3)                                    # unit q of `row` reads source slot src
    idx = src[None, None, None, :, :, None].expand(nw, M // bm, K // 32, bm, 4, 8)
    return torch.gather(t, 4, idx).reshape(nw, M // bm, K // 32, bm * 32)


def bn_fold(w, b, bn_w, bn_b, rm, rv, eps=1e-5):
    """Eval-mode BatchNorm folded into the preceding conv (modules.py:287,293,300)."""
    s = bn_w / torch.sqrt(rv + eps)
    shape = [-1] + [1] * (w.dim() - 1)
    return w * s.view(shape), (b - rm) * s + bn_b


def conv2d_taps(w, pad_f, pad_t):
    """nn.Conv2d weight [M, C, kF, kT] -> ([1, M, kF*kT, C], df, dt)."""
    M, Cc, kF, kT = w.shape
    taps = w.permute(0, 2, 3, 1).reshape(1, M, kF * kT, Cc)
    df = [jf - pad_f for jf in range(kF) for _ in range(kT)]
    dt = [jt - pad_t for _ in range(kF) for jt in range(kT)]
    return taps, df, dt


def conv1d_taps(w, dilation, padding):
    """nn.Conv1d weight [M, C, k] (time axis) -> ([1, M, k, C], df, dt)."""
    M, Cc, k = w.shape
    return w.permute(0, 2, 1).reshape(1, M, k, Cc), [0] * k, [j * dilation - padding for j in range(k)]


def convtr_taps(w, stride):
    """nn.ConvTranspose2d weight [Cin, Cout, K, 1] along frequency -> `stride` interleaved convolutions:
    output row fo uses kernel rows kk = fo%stride + j*stride with source row fo//stride - j."""
    Cin, Cout, K, kT = w.shape
    assert kT == 1
    nt = math.ceil(K / stride)
    taps = torch.zeros(stride, Cout, nt, Cin, device=w.device)
    for r in range(stride):
        for j in range(nt):
            kk = r + j * stride
            if kk < K:
                taps[r, :, j, :] = w[:, :, kk, 0].t()
    return taps, [-j for j in range(nt)], [0] * nt


def convtr_tail_image(w, stride, device):
    """the last decoder layer's nn.ConvTranspose2d weight [Cin, 2, 8, 1] (stride 4) as the 16 x Cin matrix of the fused tail
    (include/aero_hip.h, aero_conv_desc.tail_w): row 2 k + co holds W[:, co, k, 0]; fp16 [16][roundup(Cin, 32)].  None for any other shape."""
    Cin, Cout, K, kT = w.shape
    if kT != 1 or Cout != 2 or K != 8 or stride != 4:
        return None
    cp = _round_up(Cin, 32)
    img = torch.zeros(16, cp, dtype=torch.float32, device=w.device)
    img[:, :Cin] = w[:, :, :, 0].permute(2, 1, 0).reshape(16, Cin)           # [k][co][c] -> row 2 k + co
    return img.to(device=device, dtype=torch.float16).contiguous()


def convtr_stacked_spec(w, bias, stride, device, act=_lib.ACT_NONE):
    """nn.ConvTranspose2d [Cin, Cout, K, 1] computed from the input side (aero_hip.h, row scatter): the `stride` residue
    classes are stacked into M = stride*Cout rows (row r*Cout + m), taps df = 0, -1, ...; output channel block r of the
    input-aligned row q belongs to frequency row q*stride + r.  None when Cout is not a multiple of 8."""
    Cin, Cout, K, kT = w.shape
    if Cout % 8 or kT != 1:
        return None
    taps, df, dt = convtr_taps(w, stride)                        # [stride, Cout, nt, Cin]
    stacked = taps.reshape(1, stride * Cout, taps.shape[2], Cin)
    b = None if bias is None else bias.detach().float().repeat(stride)
    spec = make_conv_spec(stacked, b, Cin, 0, df, dt, device, act=act)
    spec.extra.update(scatter_M=Cout, scatter_stride=stride)
    return spec


def gram_tables(w, bias, device):
    """fp64 tables of aero_gram_stats for a pointwise conv y = W x + b (w [M, C] as the kernels see it, i.e. rounded to
    fp16; bias [M] fp32 or None): G = W'^T W' and g1 = sum_m W'[m] with W' = [W b], zero padded to Cp = 16*ceil((C+1)/16)."""
    M, Cc = w.shape
    wq = w.detach().half().double()
    b = torch.zeros(M, dtype=torch.float64) if bias is None else bias.detach().double().cpu()
    wa = torch.cat([wq.cpu(), b[:, None]], 1)                     # [M, C+1]
    Cp = _round_up(Cc + 1, 16)
    G = torch.zeros(Cp, Cp, dtype=torch.float64)
    G[:Cc + 1, :Cc + 1] = wa.t() @ wa
    g1 = torch.zeros(Cp, dtype=torch.float64)
    g1[:Cc + 1] = wa.sum(0)
    return G.to(device).contiguous(), g1.to(device).contiguous()


_PERM = {}


def lstm_gate_perm(H, device='cpu'):
    """row index 4*j+gate of the kernel <- row gate*H + j of nn.LSTM (i,f,g,o blocks); cached per device (no host-to-device copy
    per call: the training step re-packs after every optimizer step, possibly inside a HIP-graph capture)."""
    key = (H, str(device))
    if key not in _PERM:
        j = torch.arange(H)
        _PERM[key] = torch.stack([g * H + j for g in range(4)], 1).reshape(-1).to(device)
    return _PERM[key]


def pack_lstm_layer(lib, sd, prefix, layer, H, device):
    """-> (ConvSpec for the input projection of both directions, xbias fp16 [8H], whh fp16 [2,MP,KP])."""
    perm = lstm_gate_perm(H, sd[f'{prefix}.weight_ih_l{layer}'].device)
    w_ih, b, w_hh = [], [], []
    for sfx in ('', '_reverse'):
        w_ih.append(sd[f'{prefix}.weight_ih_l{layer}{sfx}'].float()[perm])
        bsum = sd.get(f'{prefix}.bias_sum_l{layer}{sfx}')        # (the training engine keeps b_ih + b_hh as a derived entry: pure data movement from there on)
        if bsum is None:
            bsum = sd[f'{prefix}.bias_ih_l{layer}{sfx}'].float() + sd[f'{prefix}.bias_hh_l{layer}{sfx}'].float()
        b.append(bsum.float()[perm])
        w_hh.append(sd[f'{prefix}.weight_hh_l{layer}{sfx}'].float()[perm])
    w_ih = torch.cat(w_ih, 0)                                  # [8H, in]
    b = torch.cat(b, 0)
    spec = make_conv_spec(w_ih[None, :, None, :], b, w_ih.shape[1], 0, [0], [0], device)
    MP, KP = lib.lstm_geometry(H)
    whh = torch.zeros(2, MP, KP, device=w_ih.device)
    for dr in range(2):
        whh[dr, :4 * H, :H] = w_hh[dr]
    fused = None
    in_ch = w_ih.shape[1]
    KPI = lib.lstm_geometry_in(H, in_ch)
    if KPI is not None:                                          # W_ih x_t computed inside the recurrent kernel
        wih = torch.zeros(2, MP, KPI, device=w_ih.device)
        for dr in range(2):
            wih[dr, :4 * H, :in_ch] = w_ih[dr * 4 * H:(dr + 1) * 4 * H]
        fused = (wih.to(device=device, dtype=torch.float16).contiguous(), b.float().to(device).contiguous(), in_ch)
    return spec, b.to(device=device, dtype=torch.float16).contiguous(), \
        whh.to(device=device, dtype=torch.float16).contiguous(), fused


def dconv_row_layer(w1, b1, g1, be1, w2, b2, g2, be2, scale, dilation, device):
    """One DConv layer for aero_dconv_row_fwd (include/aero_hip.h, K14).  w1 [hidden, C, 3], w2 [2C, hidden] (Conv1d weights),
    g*/be* None without GroupNorm.  Returns dict(w1 fp16 [HP][K1p], w2 fp16 [2C][HP] GLU-interleaved, consts fp32, ...)."""
    hid, Cc = w1.shape[0], w1.shape[1]
    HP, K1p = _round_up(hid, 16), _round_up(3 * Cc, 32)
    i1 = torch.zeros(HP, K1p)
    i1[:hid, :3 * Cc] = w1.detach().float().permute(0, 2, 1).reshape(hid, 3 * Cc)
    i2 = torch.zeros(2 * Cc, HP)
    i2[:, :hid] = glu_interleave(w2.detach().float())
    # MFMA 16x16x16 A fragments in lane order: [mf][ks][lane = g*16 + col][e] = W2[mf*16 + col][ks*16 + g*4 + e]
    i2 = i2.reshape(2 * Cc // 16, 16, HP // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()       # [mf, ks, g, col, e]
    c = torch.zeros(3 * HP + 7 * Cc)
    c[:hid] = b1.detach().float()
    c[HP:HP + hid] = 1.0 if g1 is None else g1.detach().float()
    if be1 is not None:
        c[2 * HP:2 * HP + hid] = be1.detach().float()
    o = 3 * HP
    c[o:o + 2 * Cc] = glu_interleave(b2.detach().float())
    c[o + 2 * Cc:o + 4 * Cc] = 1.0 if g2 is None else glu_interleave(g2.detach().float())
    if be2 is not None:
        c[o + 4 * Cc:o + 6 * Cc] = glu_interleave(be2.detach().float())
    c[o + 6 * Cc:] = 1.0 if scale is None else scale.detach().float()
    return dict(w1=i1.to(device=device, dtype=torch.float16).contiguous(), w2=i2.to(device=device, dtype=torch.float16).contiguous(),
                consts=c.to(device).contiguous(), norm1=int(g1 is not None), norm2=int(g2 is not None), dilation=int(dilation),
                C=Cc, hidden=hid)
