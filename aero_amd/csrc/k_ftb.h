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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
