// k_gram.h -- GroupNorm statistics of a pointwise conv's output from the Gram matrix of its INPUT (include/aero_hip.h,
// aero_gram_stats).  Replaces the statistics-only conv pass of the DConv tail's recompute pair: that pass ran the whole
// 2C x C/4 contraction per step just to sum its outputs; the sums only depend on S = sum_t x'_t x'_t^T (a (C/4+1)^2
// matrix per row), which costs 8x fewer MACs and reads the same C/4-channel rows.
// One block (4 waves) per (b, f) row.  The row is staged 128 steps at a time TRANSPOSED in LDS ([channel][step], plus a
// constant-one channel), so both MFMA operands of S = X'^T X' are K-contiguous fragments of the same tile; each wave
// owns block-rows of S.  The fp32 S blocks are contracted with the caller's fp64 tables in registers.
// Roofline: HBM (2 B per input element), a few hundred MFMAs per row.
#pragma once
#include "aero_common.h"

#define AERO_GRAM_TT 128
#define AERO_GRAM_CMAX 112                    /* channels + 1 <= 112  ->  up to 7 x 7 blocks of 16 */

__global__ __launch_bounds__(256) void aero_gram_stats_kernel(aero_gram_desc d, int Cp) {
    constexpr int TS = AERO_GRAM_TT + 8;                        // padded row: conflict-free b128 fragment reads
    __shared__ AERO_LDS_ALIGN h16 Ht[AERO_GRAM_CMAX * TS];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int row = blockIdx.x;
    const int b = row / d.F, f = row - b * d.F;
    const int C = d.C, T = d.T;
    const int nb = Cp >> 4;                                     // 16 x 16 blocks per side
    const h16* src = (const h16*)d.x + (int64_t)b * d.s_b + (int64_t)f * d.s_f;
    const bool vec = (C % 8 == 0) && (d.s_t % 8 == 0) && ((((uintptr_t)src) & 15) == 0);
    // this wave's block rows bi = wave, wave + 4 (nb <= 7): accumulators for all nb block columns
    f32x4 acc[2][7];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += AERO_GRAM_TT) {
        __syncthreads();                                        // previous tile consumed
        // stage [step][channel] -> Ht[channel][step]; channel C is the constant one (valid steps only), the rest zero
        if (vec) {
            const int cv = C >> 3;
            for (int idx = tid; idx < AERO_GRAM_TT * cv; idx += 256) {
                const int tl = idx / cv, c8 = idx - tl * cv;
                h16x8 v = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (t0 + tl < T) v = *(const h16x8*)(src + (int64_t)(t0 + tl) * d.s_t + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) Ht[(c8 * 8 + e) * TS + tl] = v[e];
            }
        } else {
            for (int idx = tid; idx < AERO_GRAM_TT * C; idx += 256) {
                const int tl = idx / C, c = idx - tl * C;
                Ht[c * TS + tl] = (t0 + tl < T) ? src[(int64_t)(t0 + tl) * d.s_t + c] : (h16)0;
            }
        }
        for (int idx = tid; idx < AERO_GRAM_TT * (Cp - C); idx += 256) {
            const int c = C + idx / AERO_GRAM_TT, tl = idx % AERO_GRAM_TT;
            Ht[c * TS + tl] = (c == C && t0 + tl < T) ? (h16)1.0f : (h16)0;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < AERO_GRAM_TT / 32; ++ks) {
            const int ko = ks * 32 + (lane >> 4) * 8;
            h16x8 bf[7];
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < nb) bf[j] = *(const h16x8*)&Ht[(j * 16 + (lane & 15)) * TS + ko];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int bi = wave + 4 * a;
                if (bi >= nb) continue;
                const h16x8 af = *(const h16x8*)&Ht[(bi * 16 + (lane & 15)) * TS + ko];
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    if (j < nb) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[a][j], 0, 0, 0);
            }
        }
    }
    // contraction with the fp64 tables: D block (bi, bj), lane -> rows i = bi*16 + (lane>>4)*4 + r, column j = bj*16 + (lane&15)
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int bi = wave + 4 * a;
        if (bi >= nb) continue;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (j >= nb) continue;
            const int col = j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = bi * 16 + (lane >> 4) * 4 + r;
                const double sv = (double)acc[a][j][r];
                s2 += d.G[(int64_t)i * Cp + col] * sv;
                if (col == C) s1 += d.g1[i] * sv;               // S[i][C] = sum_t x'_i
            }
        }
    }
    s1 = aero_wave_sum(s1);
    s2 = aero_wave_sum(s2);
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        d.stats[(int64_t)row * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        d.stats[(int64_t)row * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

static int aero_gram_stats_launch(const aero_gram_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->x || !d->G || !d->g1 || !d->stats) { *err = "gram_stats: null pointer"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->F < 1 || d->T < 1 || d->C < 1) { *err = "gram_stats: bad geometry"; return AERO_ERR_ARG; }
    if (d->C + 1 > AERO_GRAM_CMAX) { *err = "gram_stats: more than 111 input channels unsupported"; return AERO_ERR_UNSUPPORTED; }
    const long rows = (long)d->B * d->F;
    if (rows > 0x7fffffffL) { *err = "gram_stats: grid too large"; return AERO_ERR_ARG; }
    const int Cp = (d->C + 1 + 15) / 16 * 16;
    AERO_LAUNCH(aero_gram_stats_kernel, dim3((unsigned)rows), dim3(256), stream, *d, Cp);
    return AERO_OK;
}
