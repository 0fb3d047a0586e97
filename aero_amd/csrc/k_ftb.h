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
                if (p.vec == 1 && n + 8 <= N) {
                    z = *(const h16x8*)src;
                } else if (p.vec == 2 && n + 8 <= N) {            // rows only 4-byte aligned (N even, e.g. encoder 0's (re, im) pairs: N = 2 T): four dwords
                    union { h16x8 h; uint32_t u[4]; } cv;
                    const uint32_t* s32 = (const uint32_t*)src;
#pragma unroll
                    for (int e = 0; e < 4; ++e) cv.u[e] = s32[e];
                    z = cv.h;
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

// Small F (the two deepest encoder levels: 16 and 8 frequency rows): the GEMM above runs one 64 x 128 tile per block with a
// single k-chunk -- 48 000 blocks of staging, transposing and barriers for an 8 x 8 product (133 us for 197 MB).  Here a
// thread owns 8 consecutive (t, c) positions (one 16-byte vector per row), keeps the F input rows in registers as fp32 and
// produces the F output rows with plain FMAs against W (fp32 in LDS, broadcast reads): 16-byte loads and stores, no barrier
// in the loop, 10-19 vector instructions per output.  HBM-bound: 4 bytes per element.
template <int FT>
__global__ __launch_bounds__(256) void aero_freqfc_small_kernel(AeroFreqFcK p) {
    __shared__ AERO_LDS_ALIGN float Wf[FT * FT];
    const aero_freqfc_desc& d = p.d;
    const int F = d.F;
    const int64_t N = p.N;
    for (int i = threadIdx.x; i < FT * FT; i += 256) {
        const int fo = i / FT, fi = i - fo * FT;
        Wf[i] = (fo < F && fi < F) ? (float)((const h16*)d.w)[(int64_t)fo * p.Kp + fi] : 0.f;
    }
    __syncthreads();
    const int64_t nv = N >> 3;                                    // vectors per (b, f) row (N % 8 == 0: host)
    const int64_t total = (int64_t)d.B * nv;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
        const int b = (int)(v / nv);
        const int64_t n = (v - (int64_t)b * nv) * 8;
        const h16* x = (const h16*)d.x + (int64_t)b * F * N + n;
        h16* dst = (h16*)d.dst + (int64_t)b * F * N + n;
        const h16x8 gv = *(const h16x8*)((const h16*)d.gate + (int64_t)b * N + n);
        float xf[FT][8];
#pragma unroll
        for (int fi = 0; fi < FT; ++fi) {
            h16x8 xv = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (fi < F) xv = *(const h16x8*)(x + (int64_t)fi * N);
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[fi][e] = (float)xv[e];
        }
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (float)gv[e];
#pragma unroll
        for (int fo = 0; fo < FT; ++fo) {
            if (fo >= F) break;
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = 0.f;
#pragma unroll
            for (int f4 = 0; f4 < FT; f4 += 4) {
                const f32x4 w = *(const f32x4*)&Wf[fo * FT + f4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = fmaf(w[k], xf[f4 + k][e], a[e]);
            }
            h16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)(a[e] * g[e]);
            *(h16x8*)(dst + (int64_t)fo * N) = o;
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
    if (!p.vec && p.N % 2 == 0 && (((uintptr_t)d->x & 3) == 0)) p.vec = 2;   // (the scalar form of this case: 88 us for encoder 0's 66 MB)
    p.nnt = (int)((p.N + 127) / 128);
    p.nmt = (d->F + 63) / 64;
    const long nwg = (long)d->B * p.nnt * p.nmt;
    if (nwg > 0x7fffffffL) { *err = "freqfc: grid too large"; return AERO_ERR_ARG; }
    dim3 grid((unsigned)nwg), block(256);
    if (d->F <= 16 && p.vec == 1 && (((uintptr_t)d->dst | (uintptr_t)d->gate) & 15) == 0) {
        const int64_t total = (int64_t)d->B * (p.N >> 3);
        const int64_t want = (total + 255) / 256;
        dim3 sgrid((unsigned)(want < 256 * 16 ? want : 256 * 16));
        if (d->F <= 8) AERO_LAUNCH(aero_freqfc_small_kernel<8>, sgrid, block, stream, p);
        else AERO_LAUNCH(aero_freqfc_small_kernel<16>, sgrid, block, stream, p);
        return AERO_OK;
    }
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
    constexpr int NI = MF;                                      // operand items per thread: BN * (C/8) / 256 <= MF
    __shared__ AERO_LDS_ALIGN h16 As[KT * BM * 32];            // [KT][BM][32] weights, resident
    __shared__ AERO_LDS_ALIGN h16 Bs[KT * BN * 32];            // [KT][BN][32] operand tile (slots above C stay zero)
    __shared__ AERO_LDS_ALIGN h16 Cs[BN * CS];                 // [BN][CS] output staging (its own buffer: 2 barriers/tile)
    __shared__ AERO_LDS_ALIGN float kc[3][64];                 // pre_conv coefficients [p0 | p1 | pb * rs[f]], zero above C
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
    for (int v = tid; v < KT * BN * 4; v += 256) *(h16x8*)&Bs[v * 8] = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
    const h16* xn = (const h16*)d.xn + ((int64_t)row * T) * 2;
    const h16* un = (const h16*)d.u + ((int64_t)row * T) * 2;
    const h16* gate = (const h16*)d.gate + (int64_t)b * T * C;
    h16* drow = (h16*)d.dst + ((int64_t)row * T) * C;
    const float rsf = d.rs[f];
    if (tid < 64) {
        const bool in = tid < C;
        kc[0][tid] = in ? d.p0[tid] : 0.f;
        kc[1][tid] = in ? d.p1[tid] : 0.f;
        kc[2][tid] = in ? d.pb[tid] * rsf : 0.f;
    }
    // epilogue coefficients of this lane's output rows, as pairs for the packed-fp32 FMAs
    f32x2 ar[MF][2], ai[MF][2], bb[MF][2];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int m = i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mi = m + r < C ? m + r : C - 1;
            ar[i][r >> 1][r & 1] = d.a_re[mi];
            ai[i][r >> 1][r & 1] = d.a_im[mi];
            bb[i][r >> 1][r & 1] = d.bias[mi];
        }
    }
    // operand items: (position, 8-channel slot) pairs, nvec = C/8 slots per position and no idle lanes: item j = tid +
    // 256 i -> position j / nvec, slot j % nvec.  The mapping does not depend on the tile: computed once.
    const int nvec = C >> 3;
    int it_pos[NI], it_c[NI], it_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = tid + 256 * i;
        const int pos = j / nvec, sl = j - pos * nvec;
        const bool live = j < BN * nvec;
        it_pos[i] = live ? pos : -1;
        it_c[i] = sl * 8;
        it_off[i] = (sl >> 2) * BN * 32 + aero_tile_off(live ? pos : 0, sl & 3);
    }
    __syncthreads();
    const int ntt = (T + BN - 1) / BN;
    for (int tt = 0; tt < ntt; ++tt) {
        const int t0 = tt * BN;
        // attention-branch operand att[pos][c] = gate * (p0*U_re + p1*U_im + pb*rs), built on the fly (the pre_conv
        // coefficients are re-read from LDS per tile: see the note above the kernel)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + it_pos[i];
            if (it_pos[i] < 0) continue;
            h16x8 o = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (t < T) {
                const h16x2 uu = *(const h16x2*)(un + t * 2);
                const f32x2 ur = {(float)uu[0], (float)uu[0]}, ui = {(float)uu[1], (float)uu[1]};
                const h16x8 g8 = *(const h16x8*)(gate + t * C + it_c[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 k0 = *(const f32x2*)&kc[0][it_c[i] + 2 * e];
                    const f32x2 k1 = *(const f32x2*)&kc[1][it_c[i] + 2 * e];
                    const f32x2 kb = *(const f32x2*)&kc[2][it_c[i] + 2 * e];
                    const f32x2 g = {(float)g8[2 * e], (float)g8[2 * e + 1]};
                    const f32x2 v = g * (k0 * ur + (k1 * ui + kb));
                    o[2 * e] = (h16)v[0];
                    o[2 * e + 1] = (h16)v[1];
                }
            }
            *(h16x8*)&Bs[it_off[i]] = o;
        }
        f32x2 re[2], im[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int t = t0 + (wave * 2 + n) * 16 + (lane & 15);
            float r0 = 0.f, i0 = 0.f;
            if (t < T) {
                const h16x2 vv = *(const h16x2*)(xn + t * 2);
                r0 = (float)vv[0];
                i0 = (float)vv[1];
            }
            re[n] = (f32x2){r0, r0};
            im[n] = (f32x2){i0, i0};
        }
        aero_lds_barrier();                                    // operand tile complete (and the previous tile's staging read out)
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
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int pos = (wave * 2 + n) * 16 + (lane & 15);
                const f32x2 lo = ar[i][0] * re[n] + (ai[i][0] * im[n] + ((f32x2){acc[i][n][0], acc[i][n][1]} + bb[i][0]));
                const f32x2 hi = ar[i][1] * re[n] + (ai[i][1] * im[n] + ((f32x2){acc[i][n][2], acc[i][n][3]} + bb[i][1]));
                const h16x4 o = {(h16)fmaxf(lo[0], 0.f), (h16)fmaxf(lo[1], 0.f), (h16)fmaxf(hi[0], 0.f), (h16)fmaxf(hi[1], 0.f)};
                *(h16x4*)&Cs[pos * CS + m] = o;
            }
        }
        aero_lds_barrier();                                    // staging complete; every wave is done reading the operand tile
        for (int idx = tid; idx < BN * nvec; idx += 256) {
            const int pos = idx / nvec, cv = idx - pos * nvec;
            const int t = t0 + pos;
            if (t < T) *(h16x8*)(drow + t * C + cv * 8) = *(const h16x8*)&Cs[pos * CS + cv * 8];
        }
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
    if ((int64_t)d->T * d->C > 0x7fffffffLL) { *err = "ftb_first: row span exceeds 32 bits"; return AERO_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)nwg), block(256);
    const int mf = (d->C + 15) / 16;
    if (mf == 1) AERO_LAUNCH((aero_ftb_first_kernel<1>), grid, block, stream, p);
    else if (mf == 2) AERO_LAUNCH((aero_ftb_first_kernel<2>), grid, block, stream, p);
    else if (mf == 3) AERO_LAUNCH((aero_ftb_first_kernel<3>), grid, block, stream, p);
    else AERO_LAUNCH((aero_ftb_first_kernel<4>), grid, block, stream, p);
    return AERO_OK;
}
