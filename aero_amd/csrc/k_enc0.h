// k_enc0.h -- encoder 0 of the U-Net in ONE kernel: pre_conv + FTB (aero.py:119-123, modules.py:304-325, eval-mode
// BatchNorm) + the strided frequency conv with its GELU (aero.py:95,124-127), without the 48-channel tensors in between.
//
// k_ftb.h collapses pre_conv + FTB onto the 2-channel normalised spectrogram v = (re, im):
//     x0[b,f,t,m] = relu( sum_c W2a[m][c] * att[c] + a_re[m]*re + a_im[m]*im + bias[m] ),
//     att[c]      = gate[b,t,c] * (p0[c]*U_re + p1[c]*U_im + pb[c]*rs[f]),      U = freq_fc(v)  (2 channels)
// The contraction over c does not involve f except through the three scalars (U_re, U_im, rs[f]), so it factors:
//     sum_c W2a[m][c] att[c] = U_re * G0[b,t,m] + U_im * G1[b,t,m] + rs[f] * G2[b,t,m],
//     Gq[b,t,m] = sum_c (W2a[m][c] * pq[c]) * gate[b,t,c]                      (one 1x1 conv of the gate: 3C outputs)
// i.e. x0 is a six-term bilinear form  x0 = relu( S(b,f,t) . V(b,t,m) )  with S = (U_re, U_im, rs[f], re, im, 1) and
// V = (G0, G1, G2, a_re, a_im, bias): six FMAs per element instead of a 48-long dot product, and V does not depend on f.
// (The six-term sum runs in packed fp16: |terms| = O(1), the result is rounded to fp16 for the MFMA in any case.)
// A block owns four consecutive OUTPUT rows fo of the strided conv and one 128-step time tile: it keeps its slice of
// G in registers (fp16), walks over the 4*stride + K - stride input rows those outputs touch, evaluates x0 of each row
// DIRECTLY in the MFMA B-fragment layout (lane = position, 8 consecutive channels) and feeds it to the conv MFMAs of the
// (at most two) output rows that use the row.  No LDS traffic, no barrier in that loop; the conv weights sit in LDS.
// Before: aero_ftb_first (788 MB written) + conv (788 MB read, every row twice): 0.71 ms of the 12.8-ms step.
// Bytes: reads 8 B per input (f, t) position (v, U) + the G tile, writes 2*M B per output position.  HBM / VALU bound.
#pragma once
#include "aero_common.h"

static __device__ __forceinline__ h16x8 aero_splat8(h16 v) { return (h16x8){v, v, v, v, v, v, v, v}; }

struct AeroEnc0K {
    aero_enc0_desc d;
    int Cp, Mpad, Ktot;
};

template <int MFC, int KT, int ROWS>        // MFC = M/16 conv row fragments, KT = Cp/32 k-steps, ROWS = output rows per block
__global__ __launch_bounds__(256, 2) void aero_enc0_kernel(AeroEnc0K p) {
    constexpr int BN = 128;
    constexpr int BMC = MFC * 16;
    constexpr int CS = BMC + 8;
    h16* Ws = (h16*)AERO_DYN_SMEM;                              // [ktaps][KT][BMC][32] conv weights (tile-swizzled)
    const aero_enc0_desc& d = p.d;
    const int nrows = (ROWS - 1) * d.stride + d.ktaps;          // input rows the block's outputs touch
    h16* Cs = Ws + d.ktaps * KT * BMC * 32;                      // [BN][CS] output staging
    float* cst = (float*)(Cs + BN * CS);                         // [3][64] a_re | a_im | bias_f, zero above C;  [64] conv bias;  [64] rs of the rows
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int C = d.C, T = d.T, M = d.M;
    const int ntt = (T + BN - 1) / BN;
    const int nfb = (d.Fo + ROWS - 1) / ROWS;
    int id = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int tt = id % ntt;
    id /= ntt;
    const int fb = id % nfb;
    const int b = id / nfb;
    const int fo0 = fb * ROWS, t0 = tt * BN;
    // conv weights -> LDS, once per block: spec image [Mpad][ktaps*Cp] row-major (k = tap*Cp + channel)
    for (int v = tid; v < d.ktaps * KT * BMC * 4; v += 256) {
        const int tile = v / (BMC * 4), rem = v - tile * (BMC * 4);
        const int r = rem >> 2, q = rem & 3;
        const int tap = tile / KT, kt = tile - tap * KT;
        *(h16x8*)&Ws[tile * BMC * 32 + aero_tile_off(r, q)] =
            *(const h16x8*)((const h16*)d.wc + (int64_t)r * p.Ktot + tap * p.Cp + kt * 32 + q * 8);
    }
    if (tid < 64) {
        const bool in = tid < C;
        cst[tid] = in ? d.a_re[tid] : 0.f;
        cst[64 + tid] = in ? d.a_im[tid] : 0.f;
        cst[128 + tid] = in ? d.bias_f[tid] : 0.f;
        cst[192 + tid] = (tid < M && d.bias_c) ? d.bias_c[tid] : 0.f;
        const int fr = fo0 * d.stride - d.pad + tid;
        cst[256 + tid] = (tid < nrows && fr >= 0 && fr < d.F) ? d.rs[fr] : 0.f;
    }
    const int fi_lo = fo0 * d.stride - d.pad;
    // this lane's slice of G: positions pn = t0 + (2*wave + n)*16 + col, channels ks*32 + g*8 .. +7 (the B fragment)
    h16x8 G[3][2][KT];
    int pos[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        pos[n] = t0 + (wave * 2 + n) * 16 + col;
        const int tq = pos[n] < T ? pos[n] : T - 1;
        const h16* gp = (const h16*)d.g + ((int64_t)b * T + tq) * (3 * C);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                const int c = ks * 32 + g * 8;
                G[q][n][ks] = c < C ? *(const h16x8*)(gp + q * C + c) : (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
    }
    f32x4 acc[ROWS][MFC][2];
    __syncthreads();                                             // weights and constants are in LDS
    h16x8 K8[3][KT];                                             // a_re | a_im | bias_f of this lane's channels (zero above C)
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) K8[q][ks][e] = (h16)cst[q * 64 + ks * 32 + g * 8 + e];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < MFC; ++i) {
            acc[r][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[r][i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    const h16* xnb = (const h16*)d.xn + (int64_t)b * d.F * T * 2;
    const h16* ub = (const h16*)d.u + (int64_t)b * d.F * T * 2;
    // S scalars of one input row for this lane's two positions: (U_re, U_im) and (re, im); prefetched one row ahead.
    // (Staging all the block's rows through LDS in one batch instead was slower: 315 us against 288 us.)
    auto load_s = [&](int fi, h16x2 (&uu)[2], h16x2 (&vv)[2]) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            uu[n] = (h16x2){0, 0};
            vv[n] = (h16x2){0, 0};
            if (fi >= 0 && fi < d.F && pos[n] < T) {
                uu[n] = *(const h16x2*)(ub + ((int64_t)fi * T + pos[n]) * 2);
                vv[n] = *(const h16x2*)(xnb + ((int64_t)fi * T + pos[n]) * 2);
            }
        }
    };
    h16x2 un[2], vn[2];
    load_s(fi_lo, un, vn);
#pragma unroll 1
    for (int ri = 0; ri < nrows; ++ri) {
        const int fi = fi_lo + ri;
        h16x2 uu[2], vv[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { uu[n] = un[n]; vv[n] = vn[n]; }
        if (ri + 1 < nrows) load_s(fi + 1, un, vn);
        if (fi < 0 || fi >= d.F) continue;                       // zero padding of the strided conv (block-uniform)
        const h16 rsf = (h16)cst[256 + ri];
        // x0 of row fi in B-fragment layout: xb[ks][n] = 8 channels of position n.  Packed fp16 FMAs (v_pk_fma_f16): the
        // operands and the result are fp16 anyway, the six-term sum in fp32 cost 2.2x the VALU time of the whole kernel
        h16x8 xb[KT][2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const h16x8 ur = aero_splat8(uu[n][0]), ui = aero_splat8(uu[n][1]), re = aero_splat8(vv[n][0]), im = aero_splat8(vv[n][1]);
            const h16x8 rs8 = aero_splat8(rsf);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                const h16x8 v = G[0][n][ks] * ur + (G[1][n][ks] * ui + (G[2][n][ks] * rs8 + (K8[0][ks] * re + (K8[1][ks] * im + K8[2][ks]))));
                xb[ks][n] = __builtin_elementwise_max(v, (h16x8){0, 0, 0, 0, 0, 0, 0, 0});
            }
        }
        // the output rows that use input row fi: tap j = fi - (fo*stride - pad) in [0, ktaps)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int j = ri - r * d.stride;
            if (j < 0 || j >= d.ktaps) continue;                 // block-uniform
            const h16* Wj = Ws + j * KT * BMC * 32;
#pragma unroll
            for (int ks = 0; ks < KT; ++ks)
#pragma unroll
                for (int i = 0; i < MFC; ++i) {
                    const h16x8 a = *(const h16x8*)&Wj[ks * BMC * 32 + aero_tile_off(i * 16 + col, g)];
                    acc[r][i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[ks][0], acc[r][i][0], 0, 0, 0);
                    acc[r][i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[ks][1], acc[r][i][1], 0, 0, 0);
                }
        }
    }
    // epilogue: + bias, activation, transpose through LDS, 16-byte channel vectors out
    constexpr int nvec = BMC / 8;                                // (M == BMC: the launcher takes M % 16 == 0)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int fo = fo0 + r;
        if (fo >= d.Fo) break;                                   // block-uniform
#pragma unroll
        for (int i = 0; i < MFC; ++i) {
            const int m = i * 16 + g * 4;
            const f32x4 bc = *(const f32x4*)&cst[192 + m];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int pl = (wave * 2 + n) * 16 + col;
                const f32x4 v = acc[r][i][n] + bc;
                f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
                if (d.act == AERO_ACT_GELU) { lo = aero_gelu2(lo); hi = aero_gelu2(hi); }
                else if (d.act == AERO_ACT_RELU) { lo = {fmaxf(lo[0], 0.f), fmaxf(lo[1], 0.f)}; hi = {fmaxf(hi[0], 0.f), fmaxf(hi[1], 0.f)}; }
                *(h16x4*)&Cs[pl * CS + m] = (h16x4){(h16)lo[0], (h16)lo[1], (h16)hi[0], (h16)hi[1]};
            }
        }
        aero_lds_barrier();
        h16* drow = (h16*)d.dst + (((int64_t)b * d.Fo + fo) * T) * M;
        for (int idx = tid; idx < BN * nvec; idx += 256) {
            const int pl = idx / nvec, cv = idx - pl * nvec;
            const int t = t0 + pl;
            if (t < T) *(h16x8*)(drow + (int64_t)t * M + cv * 8) = *(const h16x8*)&Cs[pl * CS + cv * 8];
        }
        aero_lds_barrier();
    }
}

static int aero_enc0_launch(const aero_enc0_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->xn || !d->u || !d->g || !d->rs || !d->a_re || !d->a_im || !d->bias_f || !d->wc || !d->dst) { *err = "enc0: null pointer"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->F < 1 || d->T < 1 || d->Fo < 1 || d->ktaps < 1 || d->ktaps > 9 || d->stride < 1 || d->pad < 0) { *err = "enc0: bad geometry"; return AERO_ERR_ARG; }
    if (d->C % 8 || d->C < 8 || d->C > 64) { *err = "enc0: C must be a multiple of 8 in [8,64]"; return AERO_ERR_UNSUPPORTED; }
    if (d->M % 16 || d->M < 16 || d->M > 64) { *err = "enc0: M must be a multiple of 16 in [16,64]"; return AERO_ERR_UNSUPPORTED; }
    if (d->act != AERO_ACT_NONE && d->act != AERO_ACT_RELU && d->act != AERO_ACT_GELU) { *err = "enc0: unsupported act"; return AERO_ERR_UNSUPPORTED; }
    if (d->Fo != (d->F + 2 * d->pad - d->ktaps) / d->stride + 1) { *err = "enc0: Fo inconsistent with F/stride/pad"; return AERO_ERR_ARG; }
    if ((((uintptr_t)d->g | (uintptr_t)d->wc | (uintptr_t)d->dst) & 15) || (((uintptr_t)d->xn | (uintptr_t)d->u) & 3)) { *err = "enc0: unaligned pointer"; return AERO_ERR_ARG; }
    constexpr int ROWS = 4;
    AeroEnc0K p;
    p.d = *d;
    p.Cp = (d->C + 31) / 32 * 32;
    p.Mpad = (d->M + 127) / 128 * 128;
    p.Ktot = d->ktaps * p.Cp;
    const long nwg = (long)d->B * ((d->Fo + ROWS - 1) / ROWS) * ((d->T + 127) / 128);
    if (nwg > 0x7fffffffL) { *err = "enc0: grid too large"; return AERO_ERR_ARG; }
    const int kt = p.Cp / 32, mfc = d->M / 16;
    const int nrows = (ROWS - 1) * d->stride + d->ktaps;
    if (nrows > 64) { *err = "enc0: stride / taps too large"; return AERO_ERR_UNSUPPORTED; }
    const size_t lds = ((size_t)d->ktaps * kt * mfc * 16 * 32 + 128 * (mfc * 16 + 8)) * sizeof(h16) + 320 * sizeof(float);
    if (lds > 160 * 1024) { *err = "enc0: LDS"; return AERO_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)nwg), block(256);
#define AERO_ENC0_GO(MFC_, KT_) AERO_LAUNCH_DYN((aero_enc0_kernel<MFC_, KT_, ROWS>), grid, block, lds, stream, p)
    if (kt == 1) {
        if (mfc == 1) AERO_ENC0_GO(1, 1); else if (mfc == 2) AERO_ENC0_GO(2, 1); else if (mfc == 3) AERO_ENC0_GO(3, 1); else AERO_ENC0_GO(4, 1);
    } else {
        if (mfc == 1) AERO_ENC0_GO(1, 2); else if (mfc == 2) AERO_ENC0_GO(2, 2); else if (mfc == 3) AERO_ENC0_GO(3, 2); else AERO_ENC0_GO(4, 2);
    }
#undef AERO_ENC0_GO
    return AERO_OK;
}
