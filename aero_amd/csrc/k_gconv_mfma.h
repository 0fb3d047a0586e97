// k_gconv_mfma.h -- the MelGAN critic's grouped, strided Conv1d layers (discriminators.py:29-38: k = 41, stride 4, groups = Cin / 4)
// on the matrix cores, forward and data gradient.  At BASELINE config 5 (10-s clips at 44.1 kHz, three scales, real + fake, generator
// and critic step) these four layers are 60 GFLOP per step; the VALU forms in k_disc.h ran them at ~0.3 TFLOP/s -- 190 of the 254 ms of
// an adversarial training step.
//
// One group is a small GEMM:  y[o][to] = sum_kk W[o][kk] X[to][kk],  kk = 4 k + c  (K taps x 4 input channels = 164 -> 192).
// On channels-last rows the four channels of a group are 8 contiguous bytes and consecutive taps are consecutive ROWS, so with the
// group's input span staged in LDS as [row][4] the B fragment "8 consecutive kk of output step to" is ONE aligned 16-byte read at
// row 4 to + 2 m: no im2col, no transposes.  W (16 x 192 fp16, zero padded) sits in 24 VGPRs as A fragments.  v_mfma_f32_16x16x32_f16:
// D rows = 16 output channels of the group (4 for the last grouped layer: a quarter of the tile is used), columns = 16 output steps.
// The layers are HBM-bound in this form (0.3 flop/B of padded MFMA work per byte is irrelevant: 164 MACs per output element).
//
// Data gradient: with s = ti + pad = 4 u + r the taps that reach input step ti are k = r + 4 j from output step u - j, so
//   dx[4u + r - pad][c] = sum_{j, o} dy'[u - j][o] W[o][c][r + 4 j]:  D rows = (r, c) = 16, K = (j, o) = 12 x 16 = 192 (or 12 x 4 -> 64),
// columns = 16 values of u; the B fragment is 16 contiguous bytes of row u - j of the staged dy' span (dy' = dy * LeakyReLU'(y),
// applied while staging).  Every dx element is written exactly once (no atomics, no zero fill).
#pragma once
#include "aero_common.h"

struct AeroGconv4K {
    const h16* src; const h16* act; const h16* wimg; const float* bias; h16* dst;
    int B, Ts, Cs, Td, Cd, groups, pad, GPB, NT, ROWS;
    float slope;
};

// forward: src = x [B][Ts][Cs], dst = y [B][Td][Cd]; block = (NT output steps, GPB groups, one batch item)
template <int COG>
__global__ __launch_bounds__(256) void aero_gconv4_fwd_kernel(AeroGconv4K p) {
    h16* slab = (h16*)AERO_DYN_SMEM;                            // [GPB][ROWS][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int q = lane >> 4, col = lane & 15;
    const int b = blockIdx.z, g0 = blockIdx.y * p.GPB, to0 = blockIdx.x * p.NT;
    const int ti0 = to0 * 4 - p.pad;
    const int nvec = p.GPB >> 1;                                // 16-byte pieces (two groups) per row
    const h16* xb = p.src + (int64_t)b * p.Ts * p.Cs + g0 * 4;
    for (int idx = tid; idx < p.ROWS * nvec; idx += 256) {
        const int row = idx / nvec, v = idx - row * nvec;
        const int t = ti0 + row;
        h16x8 val = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (t >= 0 && t < p.Ts) val = *(const h16x8*)(xb + (int64_t)t * p.Cs + v * 8);
        *(h16x4*)(slab + ((2 * v) * p.ROWS + row) * 4) = (h16x4){val[0], val[1], val[2], val[3]};
        *(h16x4*)(slab + ((2 * v + 1) * p.ROWS + row) * 4) = (h16x4){val[4], val[5], val[6], val[7]};
    }
    __syncthreads();
    const int ntile = p.NT >> 4;
    for (int gl = wave; gl < p.GPB; gl += 4) {
        const int g = g0 + gl;
        h16x8 A[6];
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) A[ks] = *(const h16x8*)(p.wimg + ((int64_t)g * 16 + col) * 192 + ks * 32 + q * 8);
        f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias && (COG == 16 || q == 0)) bias4 = *(const f32x4*)(p.bias + g * COG + 4 * q);
        const h16* sg = slab + gl * p.ROWS * 4;
        for (int tile = 0; tile < ntile; ++tile) {
            const int pos = tile * 16 + col;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const h16x8 bf = *(const h16x8*)(sg + (pos * 4 + 2 * (4 * ks + q)) * 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks], bf, acc, 0, 0, 0);
            }
            const int to = to0 + pos;
            if (to < p.Td && (COG == 16 || q == 0)) {
                h16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[i] + bias4[i];
                    o[i] = (h16)(v > 0.f ? v : v * p.slope);
                }
                *(h16x4*)(p.dst + ((int64_t)b * p.Td + to) * p.Cd + g * COG + 4 * q) = o;
            }
        }
    }
}

// data gradient: src = dy, act = y (both [B][Ts = Tout][Cs = Cout]), dst = dx [B][Td = Tin][Cd = Cin]; block = (NT values of u, GPB groups)
template <int COG>
__global__ __launch_bounds__(256) void aero_gconv4_dgrad_kernel(AeroGconv4K p) {
    constexpr int BACK = COG == 16 ? 12 : 16;                   // rows u - j staged below the tile
    constexpr int KS = COG == 16 ? 6 : 2;
    constexpr int KD = KS * 32;
    h16* slab = (h16*)AERO_DYN_SMEM;                            // [GPB][ROWS][COG]
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int q = lane >> 4, col = lane & 15;
    const int b = blockIdx.z, g0 = blockIdx.y * p.GPB, u0 = blockIdx.x * p.NT;
    const int r0 = u0 - BACK;                                   // first staged row of dy
    const int pcs = p.GPB * COG / 8;                            // 16-byte pieces per row
    const h16* dyb = p.src + (int64_t)b * p.Ts * p.Cs + g0 * COG;
    const h16* yb = p.act + (int64_t)b * p.Ts * p.Cs + g0 * COG;
    for (int idx = tid; idx < p.ROWS * pcs; idx += 256) {
        const int row = idx / pcs, v = idx - row * pcs;
        const int t = r0 + row;
        h16x8 val = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (t >= 0 && t < p.Ts) {
            const h16x8 dv = *(const h16x8*)(dyb + (int64_t)t * p.Cs + v * 8);
            const h16x8 yv = *(const h16x8*)(yb + (int64_t)t * p.Cs + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = (float)yv[e] > 0.f ? dv[e] : (h16)((float)dv[e] * p.slope);
        }
        if (COG == 16) {
            const int gl = v >> 1, half = v & 1;
            *(h16x8*)(slab + (gl * p.ROWS + row) * 16 + half * 8) = val;
        } else {
            *(h16x4*)(slab + ((2 * v) * p.ROWS + row) * 4) = (h16x4){val[0], val[1], val[2], val[3]};
            *(h16x4*)(slab + ((2 * v + 1) * p.ROWS + row) * 4) = (h16x4){val[4], val[5], val[6], val[7]};
        }
    }
    __syncthreads();
    const int ntile = p.NT >> 4;
    for (int gl = wave; gl < p.GPB; gl += 4) {
        const int g = g0 + gl;
        h16x8 A[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) A[ks] = *(const h16x8*)(p.wimg + ((int64_t)g * 16 + col) * KD + ks * 32 + q * 8);
        const h16* sg = slab + gl * p.ROWS * COG;
        for (int tile = 0; tile < ntile; ++tile) {
            const int ul = tile * 16 + col + BACK;              // slab row of u
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int m = 4 * ks + q;
                h16x8 bf;
                if (COG == 16) {                                 // kk = 16 j + o: octet m = (j = m >> 1, o half = m & 1)
                    bf = *(const h16x8*)(sg + (ul - (m >> 1)) * 16 + (m & 1) * 8);
                } else {                                         // kk octet m = (j = 2m + 1, o = 0..3), (j = 2m, o = 0..3): two adjacent rows
                    const h16x4 lo = *(const h16x4*)(sg + (ul - 2 * m - 1) * 4);
                    const h16x4 hi = *(const h16x4*)(sg + (ul - 2 * m) * 4);
                    bf = (h16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks], bf, acc, 0, 0, 0);
            }
            const int ti = 4 * (u0 + tile * 16 + col) + q - p.pad;          // D rows 4q .. 4q+3 = (r = q, c = 0..3)
            if (ti >= 0 && ti < p.Td)
                *(h16x4*)(p.dst + ((int64_t)b * p.Td + ti) * p.Cd + g * 4) = (h16x4){(h16)acc[0], (h16)acc[1], (h16)acc[2], (h16)acc[3]};
        }
    }
}

// which geometries the MFMA forms take (4 input channels per group, 16 or 4 output channels per group, stride 4, zero padding)
static int aero_gconv4_ok(int Cin, int Cout, int groups, int K, int stride, int pad, int reflect) {
    if (groups < 4 || (groups & (groups - 1)) || Cin != groups * 4 || reflect || stride != 4 || K < 1 || K > 44 || pad < 0 || pad > 64) return 0;
    const int cog = Cout / groups;
    return (Cout == groups * cog) && (cog == 16 || cog == 4);
}
static void aero_gconv4_tile(int groups, int* GPB, int* NT) {
    *GPB = groups < 16 ? groups : 16;
    *NT = 1024 / *GPB;                                          // 64 steps x 16 groups, 256 x 4
}

// Weight gradient of the same layers.   dw[o][kk] = sum_{b, to} dy'[to][o] X[to][kk],  X[to][4k + c] = x[4 to - pad + k][c]:
// per group a 16 x 164 product whose contraction runs over the output steps -- the slow index of both channels-last operands.  With
// the group's input span in LDS as a flat array xs[row * 4 + c], element (to, kk) is simply xs[16 to + kk]: an MFMA B fragment ("8
// consecutive steps of column kk") is 8 two-byte LDS reads at a stride of 16 elements, an A fragment 8 reads down a column of the
// staged dy' tile; 11 column tiles of 16 share one A fragment.  LDS-read bound (96 two-byte reads per 11 MFMAs) -- at ~20 % of the
// MFMA rate still two orders of magnitude above the VALU kernel with its fp32 atomics (9.7 of the 15.4 ms of a critic step).
// A block owns GPB groups of one batch item and a chunk of 64-step tiles; each wave keeps the 11 accumulator tiles of its groups in
// registers across the chunk and stores them to the chunk's slab (plain stores; aero_wgrad_finish_kernel adds the slabs in order).
struct AeroGconv4WK {
    const h16* x; const h16* dy; const h16* y; float* slabs;
    int B, Tin, Cin, Tout, Cout, groups, K, pad, GPB, ntile, tiles_per_chunk, nchunk, DS, XG;
    float slope;
    int64_t sl_stride, w_n;
};

static __device__ __forceinline__ int aero_gw_skew(int i) { return i + ((i >> 7) << 4); }     // 16 halves of padding per 128: the four k-octets of a fragment read hit disjoint banks

template <int COG, int GPW>
__global__ __launch_bounds__(256) void aero_gconv4_wgrad_kernel(AeroGconv4WK p) {
    constexpr int NP = 64;
    h16* dyps = (h16*)AERO_DYN_SMEM;                            // [NP][DS]
    h16* xs = dyps + NP * p.DS;                                  // [GPB][XG] (skewed flat spans)
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int q = lane >> 4, col = lane & 15;
    const int chunk = blockIdx.x, g0 = blockIdx.y * p.GPB, b = blockIdx.z;
    const int nrow_x = 4 * NP + 44;
    f32x4 acc[GPW][11];
    float bsum[GPW];
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        bsum[gi] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 11; ++nt) acc[gi][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const h16* dyb = p.dy + (int64_t)b * p.Tout * p.Cout + g0 * COG;
    const h16* yb = p.y + (int64_t)b * p.Tout * p.Cout + g0 * COG;
    const h16* xb = p.x + (int64_t)b * p.Tin * p.Cin + g0 * 4;
    const int pcs_d = p.GPB * COG / 8, pcs_x = p.GPB / 2;
    const int t_lo = chunk * p.tiles_per_chunk;
    const int t_hi = t_lo + p.tiles_per_chunk < p.ntile ? t_lo + p.tiles_per_chunk : p.ntile;
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int to0 = tile * NP;
        __syncthreads();
        for (int idx = tid; idx < NP * pcs_d; idx += 256) {
            const int row = idx / pcs_d, v = idx - row * pcs_d;
            const int t = to0 + row;
            h16x8 val = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (t < p.Tout) {
                const h16x8 dv = *(const h16x8*)(dyb + (int64_t)t * p.Cout + v * 8);
                const h16x8 yv = *(const h16x8*)(yb + (int64_t)t * p.Cout + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) val[e] = (float)yv[e] > 0.f ? dv[e] : (h16)((float)dv[e] * p.slope);
            }
            *(h16x4*)(dyps + row * p.DS + v * 8) = (h16x4){val[0], val[1], val[2], val[3]};
            *(h16x4*)(dyps + row * p.DS + v * 8 + 4) = (h16x4){val[4], val[5], val[6], val[7]};
        }
        const int ti0 = 4 * to0 - p.pad;
        for (int idx = tid; idx < nrow_x * pcs_x; idx += 256) {
            const int row = idx / pcs_x, v = idx - row * pcs_x;
            const int t = ti0 + row;
            h16x8 val = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (t >= 0 && t < p.Tin) val = *(const h16x8*)(xb + (int64_t)t * p.Cin + v * 8);
            const int o = aero_gw_skew(4 * row);
            *(h16x4*)(xs + (2 * v) * p.XG + o) = (h16x4){val[0], val[1], val[2], val[3]};
            *(h16x4*)(xs + (2 * v + 1) * p.XG + o) = (h16x4){val[4], val[5], val[6], val[7]};
        }
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            const int gl = wave + 4 * gi;
            const h16* xg = xs + gl * p.XG;
            const bool arow = COG == 16 || col < COG;
#pragma unroll
            for (int ks = 0; ks < NP / 32; ++ks) {
                const int p0 = ks * 32 + 8 * q;
                h16x8 A = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (arow) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) A[e] = dyps[(p0 + e) * p.DS + gl * COG + col];
                }
                float s8 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s8 += (float)A[e];
                bsum[gi] += s8;
#pragma unroll
                for (int nt = 0; nt < 11; ++nt) {
                    const int kk = nt * 16 + col;
                    h16x8 Bf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) Bf[e] = xg[aero_gw_skew(16 * (p0 + e) + kk)];
                    acc[gi][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bf, acc[gi][nt], 0, 0, 0);
                }
            }
        }
    }
    // the chunk's partial sums -> its slab: [Cout][4K] weights, then [Cout] bias sums
    float* sl = p.slabs + ((int64_t)b * p.nchunk + chunk) * p.sl_stride;
    const int K4 = 4 * p.K;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
        const int g = g0 + wave + 4 * gi;
#pragma unroll
        for (int nt = 0; nt < 11; ++nt) {
            const int kk = nt * 16 + col;
            if (kk < K4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * q + i;
                    if (o < COG) sl[(int64_t)(g * COG + o) * K4 + kk] = acc[gi][nt][i];
                }
            }
        }
        float s = bsum[gi];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (q == 0 && col < COG) sl[p.w_n + g * COG + col] = s;
    }
}

// chunks of 64-step tiles per (batch item, group block) of the MFMA weight gradient: ~1024 blocks in all
// (four groups per block, one per wave: 44 accumulator registers.  Sixteen groups per block -- 176 accumulator registers per wave --
// ran 8x slower per MAC than the four-group form of the first grouped layer.)
#define AERO_GCONV4_WGRAD_GPB 4
static void aero_gconv4_wgrad_plan(int B, int Tout, int groups, int* ntile, int* tpc, int* nchunk) {
    const int GPB = AERO_GCONV4_WGRAD_GPB;
    *ntile = (Tout + 63) / 64;
    long want = 1024 / ((long)B * (groups / GPB));
    if (want < 1) want = 1;
    if (want > *ntile) want = *ntile;
    *tpc = (int)((*ntile + want - 1) / want);
    *nchunk = (*ntile + *tpc - 1) / *tpc;
}
