// k_ftb.h -- FTB frequency mixing (reference modules.py:314-320):
//   dst[b, fo, t, c] = gate[b, t, c] * sum_fi W[fo][fi] * x[b, fi, t, c]
// (the reference multiplies by the gate before the Linear; the gate does not depend on fi, so it
// factors out of the contraction and becomes an epilogue multiply).
// GEMM per batch item: M = fo, K = fi, N = (t, c) flattened.  The contraction axis fi is the SLOW
// axis of the channels-last activation, so the B tile [32 fi][128 n] is transposed while it is
// staged into LDS ([n][k] image, same swizzle as k_conv.h) and then feeds v_mfma_f32_16x16x32_f16.
#pragma once
#include "aero_common.h"

struct AeroFreqFcK {
    aero_freqfc_desc d;
    int Kp, vec;
    int64_t N;
    int nnt, nmt;
};

__global__ __launch_bounds__(256) void aero_freqfc_kernel(AeroFreqFcK p) {
    constexpr int MF = 4, NF = 2, BM = 64, BN = 128;
    __shared__ AERO_LDS_ALIGN h16 As[BM * 32];
    __shared__ AERO_LDS_ALIGN h16 Bs[BN * 32];
    const aero_freqfc_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    int id = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int mt = id % p.nmt;
    id /= p.nmt;
    const int nt = id % p.nnt;
    const int b = id / p.nnt;
    const int m0 = mt * BM;
    const int64_t n0 = (int64_t)nt * BN;
    const int F = d.F;
    const int64_t N = p.N;
    const h16* x = (const h16*)d.x + (int64_t)b * F * N;
    const h16* W = (const h16*)d.w + (int64_t)m0 * p.Kp;

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.Kp / 32;
    h16x8 ra, rb[2];
    auto load_chunk = [&](int kc) {
        ra = *(const h16x8*)(W + (int64_t)(tid >> 2) * p.Kp + kc * 32 + (tid & 3) * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i;
            const int k = v >> 4, nv = v & 15;
            const int fi = kc * 32 + k;
            const int64_t n = n0 + nv * 8;
            h16x8 z = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (fi < F) {
                const h16* src = x + (int64_t)fi * N + n;
                if (p.vec && n + 8 <= N) {
                    z = *(const h16x8*)src;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (n + e < N) z[e] = src[e];
                }
            }
            rb[i] = z;
        }
    };
    load_chunk(0);
    for (int kc = 0; kc < nk; ++kc) {
        *(h16x8*)&As[aero_tile_off(tid >> 2, tid & 3)] = ra;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i;
            const int k = v >> 4, nv = v & 15;
#pragma unroll
            for (int e = 0; e < 8; ++e) Bs[aero_tile_off(nv * 8 + e, k >> 3) + (k & 7)] = rb[i][e];
        }
        __syncthreads();
        if (kc + 1 < nk) load_chunk(kc + 1);
        h16x8 af[MF], bf[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = *(const h16x8*)&As[aero_tile_off(i * 16 + (lane & 15), lane >> 4)];
#pragma unroll
        for (int n = 0; n < NF; ++n) bf[n] = *(const h16x8*)&Bs[aero_tile_off((wave * NF + n) * 16 + (lane & 15), lane >> 4)];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[n], acc[i][n], 0, 0, 0);
        __syncthreads();
    }
    const h16* gate = (const h16*)d.gate + (int64_t)b * N;
    h16* dst = (h16*)d.dst + (int64_t)b * F * N;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const int64_t nn = n0 + (wave * NF + n) * 16 + (lane & 15);
        if (nn >= N) continue;
        const float g = (float)gate[nn];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int fo = m0 + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (fo + r < F) dst[(int64_t)(fo + r) * N + nn] = (h16)(acc[i][n][r] * g);
        }
    }
}

static int aero_freqfc_launch(const aero_freqfc_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->x || !d->w || !d->gate || !d->dst) { *err = "freqfc: null pointer"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->F < 1 || d->T < 1 || d->C < 1) { *err = "freqfc: bad geometry"; return AERO_ERR_ARG; }
    AeroFreqFcK p;
    p.d = *d;
    p.Kp = (d->F + 31) / 32 * 32;
    p.N = (int64_t)d->T * d->C;
    p.vec = (p.N % 8 == 0) && (((uintptr_t)d->x & 15) == 0);
    p.nnt = (int)((p.N + 127) / 128);
    p.nmt = (d->F + 63) / 64;
    const long nwg = (long)d->B * p.nnt * p.nmt;
    if (nwg > 0x7fffffffL) { *err = "freqfc: grid too large"; return AERO_ERR_ARG; }
    dim3 grid((unsigned)nwg), block(256);
    AERO_LAUNCH(aero_freqfc_kernel, grid, block, stream, p);
    return AERO_OK;
}


// ------------------------------------------------------------------------------------------------------
// First-layer FTB, algebraically collapsed (reference aero.py:119-123 + modules.py:304-325 for encoder 0).
// At layer 0 the FTB input is x0 = pre_conv(v), a LINEAR map of the 2-channel normalised spectrogram v = (re, im)
// (aero.py:89,120; no activation in between).  Everything in the FTB that is linear in x0 therefore collapses onto
// the 2 channels of v instead of 48:
//   freq_fc(gate * x0)[c]  = gate[c] * (p0[c]*U_re + p1[c]*U_im + pb[c]*rs[f]),   U = freq_fc applied to v (2 ch),
//                                                                               rs[f] = row sums of W_fc
//   conv2(cat[att, x0])[m] = sum_c W2a[m][c]*att[c] + a_re[m]*re + a_im[m]*im + bias[m]      (BatchNorm folded)
// so the 48-channel tensors x0 and att are never written to HBM: per position this kernel reads 4 B of v, 4 B of U
// and the (L2-resident) gate row, builds att in registers as the MFMA B operand, and writes the 2C-byte output once.
// HBM-bound: algorithmic bytes per position = 8 + 2*C (+ gate reuse).
struct AeroFtbFirstK {
    aero_ftb_first_desc d;
    int Kp;
};

// One block owns a whole (b, f) row and walks over its 128-step tiles: the weight tile and the epilogue coefficients are
// fetched ONCE per block.  The pre_conv coefficients (k0/k1/kb) are deliberately re-read per tile and k-step: with all
// 48 of them hoisted into registers as well the kernel was 25 % faster but returned run-to-run different results on the
// MI355X (lanes 48-63 of the waves building the operand tile; never in the emulator; not cured by draining vmcnt between
// load groups or by extra barriers -- tools/dbg_ftb.py reproduces it).  Root cause not found; this form is bit-reproducible.
template <int MF>
__global__ __launch_bounds__(256) void aero_ftb_first_kernel(AeroFtbFirstK p) {
    constexpr int BM = MF * 16, BN = 128, KT = 2;              // C <= 64: two k-steps of 32 channels
    constexpr int CS = BM + 8;
    constexpr int SB = KT * BN * 32 > BN * CS ? KT * BN * 32 : BN * CS;
    __shared__ AERO_LDS_ALIGN h16 As[KT * BM * 32];            // [KT][BM][32] weights, resident
    __shared__ AERO_LDS_ALIGN h16 Bs[SB];                      // [KT][BN][32] operand tile, then [BN][CS] output staging
    h16* Cs = Bs;
    const aero_ftb_first_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int C = d.C, T = d.T;
    const int row = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int b = row / d.F, f = row % d.F;
    // weights -> LDS
    for (int v = tid; v < KT * BM * 4; v += 256) {
        const int kt = v / (BM * 4), rem = v - kt * (BM * 4);
        const int r = rem >> 2, q = rem & 3;
        h16x8 w = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (kt * 32 < p.Kp) w = *(const h16x8*)((const h16*)d.w2a + (int64_t)r * p.Kp + kt * 32 + q * 8);
        *(h16x8*)&As[kt * BM * 32 + aero_tile_off(r, q)] = w;
    }
    const h16* xn = (const h16*)d.xn + ((int64_t)row * T) * 2;
    const h16* un = (const h16*)d.u + ((int64_t)row * T) * 2;
    const h16* gate = (const h16*)d.gate + (int64_t)b * T * C;
    h16* drow = (h16*)d.dst + ((int64_t)row * T) * C;
    const float rsf = d.rs[f];
    // pre_conv coefficients of the attention branch, staged once per block: [p0 | p1 | pb * rs[f]] (zero above C)
    __shared__ AERO_LDS_ALIGN float kc[3][64];
    if (tid < 64) {
        const bool in = tid < C;
        kc[0][tid] = in ? d.p0[tid] : 0.f;
        kc[1][tid] = in ? d.p1[tid] : 0.f;
        kc[2][tid] = in ? d.pb[tid] * rsf : 0.f;
    }
    __syncthreads();
    // thread tid always builds the same 8-channel slot (q = tid & 3) of each k-step: its per-channel coefficients
    // (pre_conv weights / bias) live in registers, the per-position work is 3 FMA + 1 MUL per channel
    const int qf = tid & 3;
    float ar[MF][4], ai[MF][4], bb[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mi = m + r < C ? m + r : C - 1;
            ar[i][r] = d.a_re[mi];
            ai[i][r] = d.a_im[mi];
            bb[i][r] = d.bias[mi];
        }
    }
    const int nvec = C >> 3;
    const int ntt = (T + BN - 1) / BN;
    for (int tt = 0; tt < ntt; ++tt) {
        const int t0 = tt * BN;
        // attention-branch operand att[pos][c] built on the fly
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int c = kt * 32 + qf * 8;
            float k0[KT][8], k1[KT][8], kb[KT][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                        // (re-read per tile: see DESIGN.md, ftb_first determinism)
                k0[kt][e] = kc[0][c + e];
                k1[kt][e] = kc[1][c + e];
                kb[kt][e] = kc[2][c + e];
            }
#pragma unroll
            for (int i = 0; i < BN * 4 / 256; ++i) {
                const int pos = (tid + 256 * i) >> 2;
                const int t = t0 + pos;
                h16x8 o = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (t < T && c < C) {
                    const h16x2 uu = *(const h16x2*)(un + (int64_t)t * 2);
                    const float ur = (float)uu[0], ui = (float)uu[1];
                    const h16x8 g8 = *(const h16x8*)(gate + (int64_t)t * C + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (h16)((float)g8[e] * (k0[kt][e] * ur + k1[kt][e] * ui + kb[kt][e]));
                }
                *(h16x8*)&Bs[kt * BN * 32 + aero_tile_off(pos, qf)] = o;
            }
        }
        float re[2], im[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int t = t0 + (wave * 2 + n) * 16 + (lane & 15);
            re[n] = im[n] = 0.f;
            if (t < T) {
                const h16x2 vv = *(const h16x2*)(xn + (int64_t)t * 2);
                re[n] = (float)vv[0];
                im[n] = (float)vv[1];
            }
        }
        __syncthreads();
        f32x4 acc[MF][2];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            h16x8 bf[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) bf[n] = *(const h16x8*)&Bs[kt * BN * 32 + aero_tile_off((wave * 2 + n) * 16 + (lane & 15), lane >> 4)];
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const h16x8 a = *(const h16x8*)&As[kt * BM * 32 + aero_tile_off(i * 16 + (lane & 15), lane >> 4)];
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bf[1], acc[i][1], 0, 0, 0);
            }
        }
        __syncthreads();                                       // operand tile consumed: it becomes the output tile
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int pos = (wave * 2 + n) * 16 + (lane & 15);
                h16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (h16)fmaxf(acc[i][n][r] + ar[i][r] * re[n] + ai[i][r] * im[n] + bb[i][r], 0.f);
                *(h16x4*)&Cs[pos * CS + m] = o;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < BN * nvec; idx += 256) {
            const int pos = idx / nvec, cv = idx - pos * nvec;
            const int t = t0 + pos;
            if (t < T) *(h16x8*)(drow + (int64_t)t * C + cv * 8) = *(const h16x8*)&Cs[pos * CS + cv * 8];
        }
        __syncthreads();                                       // staging read out before the next tile overwrites it
    }
}

static int aero_ftb_first_launch(const aero_ftb_first_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->xn || !d->u || !d->gate || !d->w2a || !d->p0 || !d->p1 || !d->pb || !d->rs || !d->a_re || !d->a_im ||
        !d->bias || !d->dst) { *err = "ftb_first: null pointer"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->F < 1 || d->T < 1 || d->C < 8 || d->C > 64 || d->C % 8) { *err = "ftb_first: C must be a multiple of 8 in [8,64]"; return AERO_ERR_UNSUPPORTED; }
    AeroFtbFirstK p;
    p.d = *d;
    p.Kp = (d->C + 31) / 32 * 32;
    const long nwg = (long)d->B * d->F;
    if (nwg > 0x7fffffffL) { *err = "ftb_first: grid too large"; return AERO_ERR_ARG; }
    dim3 grid((unsigned)nwg), block(256);
    const int mf = (d->C + 15) / 16;
    if (mf == 1) AERO_LAUNCH((aero_ftb_first_kernel<1>), grid, block, stream, p);
    else if (mf == 2) AERO_LAUNCH((aero_ftb_first_kernel<2>), grid, block, stream, p);
    else if (mf == 3) AERO_LAUNCH((aero_ftb_first_kernel<3>), grid, block, stream, p);
    else AERO_LAUNCH((aero_ftb_first_kernel<4>), grid, block, stream, p);
    return AERO_OK;
}
