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

// NB = 16 x 16 blocks per side of S (compile time: the block loops unroll without predication and the LDS tile is sized
// to the channel count; with a run-time count this kernel ran VALU-bound at ~3100 vector instructions per wave).
template <int NB>
__global__ __launch_bounds__(256) void aero_gram_stats_kernel(aero_gram_desc d) {
    constexpr int Cp = NB * 16;
    constexpr int TS = AERO_GRAM_TT + 8;                        // padded row: conflict-free b128 fragment reads
    constexpr int NA = NB > 4 ? 2 : 1;                          // block rows per wave: bi = wave, wave + 4
    __shared__ AERO_LDS_ALIGN h16 Ht[Cp * TS];
    __shared__ double red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int row = blockIdx.x;
    const int b = row / d.F, f = row - b * d.F;
    const int C = d.C, T = d.T;
    const h16* src = (const h16*)d.x + (int64_t)b * d.s_b + (int64_t)f * d.s_f;
    const bool vec = (C % 8 == 0) && (d.s_t % 8 == 0) && ((((uintptr_t)src) & 15) == 0);
    const int st = (int)d.s_t;                                  // a row spans < 2^31 elements (checked on the host)
    f32x4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // channels above C are zero for the whole row: written once (channel C, the constant one, is rewritten per tile)
    for (int idx = tid; idx < (Cp - C) * TS; idx += 256) Ht[C * TS + idx] = (h16)0;
    for (int t0 = 0; t0 < T; t0 += AERO_GRAM_TT) {
        __syncthreads();                                        // previous tile consumed
        // stage [step][channel] -> Ht[channel][step].  Vector path: a lane takes 8 channels of TWO adjacent steps and
        // writes 8 dwords (step pairs), a wave's lanes hitting consecutive dwords.
        if (vec) {
            const int cv = C >> 3;
            for (int idx = tid; idx < (AERO_GRAM_TT / 2) * cv; idx += 256) {
                const int tp = idx & (AERO_GRAM_TT / 2 - 1), c8 = idx >> 6;
                const int t = t0 + 2 * tp;
                h16x8 v0 = (h16x8){0, 0, 0, 0, 0, 0, 0, 0}, v1 = v0;
                if (t < T) v0 = *(const h16x8*)(src + t * st + c8 * 8);
                if (t + 1 < T) v1 = *(const h16x8*)(src + (t + 1) * st + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) *(h16x2*)&Ht[(c8 * 8 + e) * TS + 2 * tp] = (h16x2){v0[e], v1[e]};
            }
        } else {
            for (int idx = tid; idx < AERO_GRAM_TT * C; idx += 256) {
                const int tl = idx / C, c = idx - tl * C;
                Ht[c * TS + tl] = (t0 + tl < T) ? src[(int64_t)(t0 + tl) * d.s_t + c] : (h16)0;
            }
        }
        if (tid < AERO_GRAM_TT) Ht[C * TS + tid] = (t0 + tid < T) ? (h16)1.0f : (h16)0;      // valid steps only
        __syncthreads();
        if (wave < NB) {
#pragma unroll
            for (int ks = 0; ks < AERO_GRAM_TT / 32; ++ks) {
                const int ko = ks * 32 + (lane >> 4) * 8;
                h16x8 bf[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) bf[j] = *(const h16x8*)&Ht[(j * 16 + (lane & 15)) * TS + ko];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int bi = wave + 4 * a;
                    if (bi >= NB) continue;
                    const h16x8 af = *(const h16x8*)&Ht[(bi * 16 + (lane & 15)) * TS + ko];
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[a][j], 0, 0, 0);
                }
            }
        }
    }
    // contraction with the fp64 tables: D block (bi, bj), lane -> rows i = bi*16 + (lane>>4)*4 + r, column j = bj*16 + (lane&15)
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int bi = wave + 4 * a;
        if (bi >= NB) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = bi * 16 + (lane >> 4) * 4 + r;
                const double sv = (double)acc[a][j][r];
                s2 += d.G[i * Cp + col] * sv;
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
    if ((int64_t)d->T * d->s_t > 0x7fffffffLL) { *err = "gram_stats: row span exceeds 32 bits"; return AERO_ERR_UNSUPPORTED; }
    const dim3 grid((unsigned)rows), block(256);
    switch ((d->C + 1 + 15) / 16) {
        case 1: AERO_LAUNCH(aero_gram_stats_kernel<1>, grid, block, stream, *d); break;
        case 2: AERO_LAUNCH(aero_gram_stats_kernel<2>, grid, block, stream, *d); break;
        case 3: AERO_LAUNCH(aero_gram_stats_kernel<3>, grid, block, stream, *d); break;
        case 4: AERO_LAUNCH(aero_gram_stats_kernel<4>, grid, block, stream, *d); break;
        case 5: AERO_LAUNCH(aero_gram_stats_kernel<5>, grid, block, stream, *d); break;
        case 6: AERO_LAUNCH(aero_gram_stats_kernel<6>, grid, block, stream, *d); break;
        default: AERO_LAUNCH(aero_gram_stats_kernel<7>, grid, block, stream, *d); break;
    }
    return AERO_OK;
}
