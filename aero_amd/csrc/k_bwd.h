// k_bwd.h -- backward kernels of the generator's building blocks (reference: loss.backward() in
// src/solver.py:602-605 through nn.Conv2d / Conv1d / ConvTranspose2d / GroupNorm + GELU / GLU of src/models/aero.py:86-101,
// 172-179 and modules.py:206-210).  Data gradients of the convolutions run on the FORWARD kernels with re-packed weights
// (aero_amd/backward.py); here are the two things that are not convolutions of that family:
//   * aero_conv_wgrad  -- weight (and bias) gradient: a GEMM whose contraction runs over POSITIONS, the slow index of both
//                         channels-last operands;
//   * aero_norm_bwd    -- GroupNorm + activation backward (two streaming passes: group sums, then dx).
#pragma once
#include "aero_common.h"

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient.   dw[j][m][c] += sum_{b, fo, t} dy[b, fo, t, m] * x[b, fo*fstride + df[j], t + dt[j], c]   (x = 0 outside)
// Both operands are channels-last, so the contraction index (t) is the SLOW one in memory, while an MFMA lane wants 8
// consecutive k of one row.  Transposing through LDS would cost 2-byte scattered accesses; instead:
//   * a thread loads an 8 (positions) x 8 (channels) block -- eight aligned 16-byte loads -- and transposes it in registers
//     (32 dword merges): it now holds, for each of its 8 channels j, the fragment "8 consecutive positions of channel 8*o + j";
//   * MFMA number (j, j') multiplies the fragments of channel set {8*o + j} (16 lanes o) with those of {8*o' + j'}: the matrix
//     rows an MFMA covers are a STRIDED set of channels -- any row permutation is as good as any other for a GEMM -- so the
//     64 (j, j') products tile a 128 x 128 block of dw with no further data movement;
//   * fragments are exchanged between the four waves through LDS in fragment order (16-byte conflict-free writes / reads).
// Block: 256 threads, a 128 (m) x 128 ((tap, c) columns) tile, 64 positions per step (threads 0-127 stage dy, 128-255 stage x),
// wave w owns channels j = 2w, 2w+1 of the m side against all eight j'.  Partial sums of a block's row chunk are added to dw
// with fp32 atomics (the sum order across chunks is not fixed: results differ in the last bits from run to run, as
// torch's own weight gradients do).
struct AeroWgradK {
    aero_wgrad_desc d;
    int nmt, nct, SC, nchunk, noswz; int64_t w_n, sl_stride;                 // SC = 64-position steps per chunk; w_n = ntaps * M * C
    int dt_min, dt_max, ncol;                                                  // range of the time shifts; ncol = ntaps * C (the column index is (tap, c))
};

static __device__ __forceinline__ void aero_transpose8x8(const h16x8* r, h16x8* c) {
    // r[i] = 8 channels of position i  ->  c[j] = 8 positions of channel j
    uint32_t in[8][4], out[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        union { h16x8 h; uint32_t u[4]; } cv;
        cv.h = r[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) in[i][q] = cv.u[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t a = in[2 * e][q], b = in[2 * e + 1][q];
            out[2 * q][e] = (a & 0xffffu) | (b << 16);
            out[2 * q + 1][e] = (a >> 16) | (b & 0xffff0000u);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        union { h16x8 h; uint32_t u[4]; } cv;
#pragma unroll
        for (int e = 0; e < 4; ++e) cv.u[e] = out[j][e];
        c[j] = cv.h;
    }
}

__global__ __launch_bounds__(256, 2) void aero_conv_wgrad_kernel(AeroWgradK p) {
    __shared__ AERO_LDS_ALIGN h16 FR2[2][2][2][8][64 * 8];    // [step parity][operand][k half][j][lane * 8]: 2 x 32 KiB (one barrier per step)
    const aero_wgrad_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    // consecutive tiles of a row chunk share dy / x slices: give each XCD (private L2) a contiguous run of them
    int id = (p.noswz & 1) ? (int)blockIdx.x : aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int mt = id % p.nmt; id /= p.nmt;
    const int ct = id % p.nct;
    const int chunk = id / p.nct;
    // The COLUMN index of the tile is (tap, c) flattened, ncol = ntaps * C (round 4; before: one tile column set per tap, so a 48-channel
    // x side filled 48 of a tile's 128 columns, nine times over): an octet of columns lies inside one tap (C % 8 == 0), so a thread of
    // the x side has ITS tap -- time shift and frequency offset are per lane, everything else is unchanged.
    const int m0 = mt * 128, c0 = ct * 128;
    const int opnd = tid >> 7, o = tid & 15, g8 = (tid >> 4) & 7;
    const int col0 = c0 + 8 * o;                              // (x side) first of this thread's 8 columns
    const int tap_l = opnd ? (col0 < p.ncol ? col0 / d.C : 0) : 0;
    const int dtj = opnd ? d.dt[tap_l] : 0, dfj = opnd ? d.df[tap_l] : 0;
    const int nrows = d.B * d.Fout;
    const int nT = (d.T + 63) >> 6;
    const int it_lo = chunk * p.SC;                           // chunks are runs of the linear step index (row, 64-step segment): a tensor
    const int it_hi = it_lo + p.SC < nrows * nT ? it_lo + p.SC : nrows * nT;   // of few long rows still fills the chip
    const h16* zpv = aero_zero_page;
    const int ch = opnd ? col0 - tap_l * d.C : m0 + 8 * o;    // first of this thread's 8 channels
    const bool ch_ok = opnd ? col0 < p.ncol : ch < d.M;
    f32x4 acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool do_bias = d.db != nullptr && ct == 0 && opnd == 0;
    h16x8 r[8];
    // per-lane element offsets inside a 64-step segment, fixed for the whole kernel: an interior step (all 64 positions and the
    // shifted ones inside [0, T)) then costs no vector address arithmetic at all -- block-uniform row pointer + 32-bit lane offset.
    // (PMC on the first version: 7 vector + 6 scalar instructions per MFMA, the 64-bit per-load address / mask arithmetic twice
    // the transposes; the kernel was VALU-bound at 2x the MFMA time.)
    const int opu = aero_uniform(opnd);
    const int st_e = (int)(opu ? d.x_t : d.dy_t);
    const int coff = ch + 8 * g8 * st_e;                      // (the eight positions of a lane add block-uniform multiples of the step stride)
    auto load = [&](int it) {
        const int row = (it_lo + it) / nT, t0 = (it_lo + it - row * nT) * 64;
        const int b = row / d.Fout, fo = row - b * d.Fout;
        const int fi = fo * d.fstride + dfj;
        const bool row_ok = fi >= 0 && fi < d.Fin;
        const int sh = opu ? dtj : 0;
        if (t0 + 64 <= d.T && (!opu || (t0 + p.dt_min >= 0 && t0 + 64 + p.dt_max <= d.T))) {     // (block-uniform: every tap's shift stays inside the row)
            const bool live = ch_ok && (!opu || row_ok);        // a frequency offset outside the input: this lane's operand is zero
            const h16* rowp = (opu ? (const h16*)d.x + (int64_t)b * d.x_b + (int64_t)(row_ok ? fi : 0) * d.x_f
                                   : (const h16*)d.dy + (int64_t)b * d.dy_b + (int64_t)fo * d.dy_f) + (int64_t)(t0 + sh) * st_e;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = live ? *(const h16x8*)(rowp + (int64_t)i * st_e + coff) : (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            return;
        }
        const h16* base = opnd ? (const h16*)d.x + (int64_t)b * d.x_b + (int64_t)fi * d.x_f + ch
                               : (const h16*)d.dy + (int64_t)b * d.dy_b + (int64_t)fo * d.dy_f + ch;
        const int64_t st = opnd ? d.x_t : d.dy_t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + 8 * g8 + i;
            const int tt = opnd ? t + dtj : t;
            const bool ok = ch_ok && t < d.T && (!opnd || (row_ok && tt >= 0 && tt < d.T));   // (dy unmasked by the tap: x is zero there, and db sums all of dy)
            r[i] = *(const h16x8*)(ok ? base + (int64_t)tt * st : zpv);
        }
    };
    const int nit = it_hi - it_lo;
    if (nit > 0) load(0);
    for (int it = 0; it < nit; ++it) {
        h16x8 c[8];
        aero_transpose8x8(r, c);
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[j] += (float)c[j][e];
        }
        auto& FR = FR2[it & 1];                                // (the readers of this half finished before the previous step's barrier)
#pragma unroll
        for (int j = 0; j < 8; ++j) *(h16x8*)&FR[opnd][g8 >> 2][j][(o + 16 * (g8 & 3)) * 8] = c[j];
        __syncthreads();
        if (it + 1 < nit) load(it + 1);                        // next step's loads fly under the MFMAs
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const h16x8 a0 = *(const h16x8*)&FR[0][kh][2 * wave][lane * 8];
            const h16x8 a1 = *(const h16x8*)&FR[0][kh][2 * wave + 1][lane * 8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const h16x8 bf = *(const h16x8*)&FR[1][kh][j][lane * 8];
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bf, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bf, acc[1][j], 0, 0, 0);
            }
        }
    }
    // accumulator (a, j')[i] of lane l: m = m0 + 8 * ((l >> 4) * 4 + i) + 2 * wave + a,  c = c0 + 8 * (l & 15) + j'
    const int colb = c0 + 8 * (lane & 15);                    // this lane's eight columns: one tap, eight consecutive c
    const int tap_o = colb < p.ncol ? colb / d.C : 0, ccb = colb - tap_o * d.C;
    if (d.slabs) {
        // this chunk's partial tile -> its own slab with plain 16-byte stores (a lane's eight j' are eight consecutive c);
        // aero_wgrad_finish_kernel adds the slabs in chunk order: deterministic, and no scattered 4-byte atomics
        float* sl = d.slabs + (int64_t)chunk * p.sl_stride + (int64_t)tap_o * d.M * d.C;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 8 * ((lane >> 4) * 4 + i) + 2 * wave + a;
                if (m < d.M && colb < p.ncol) {
                    *(f32x4*)(sl + (int64_t)m * d.C + ccb) = (f32x4){acc[a][0][i], acc[a][1][i], acc[a][2][i], acc[a][3][i]};
                    *(f32x4*)(sl + (int64_t)m * d.C + ccb + 4) = (f32x4){acc[a][4][i], acc[a][5][i], acc[a][6][i], acc[a][7][i]};
                }
            }
    } else {
    float* dw = d.dw + (int64_t)tap_o * d.M * d.C;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 8 * ((lane >> 4) * 4 + i) + 2 * wave + a;
                if (m < d.M && colb < p.ncol && !(p.noswz & 2)) atomicAdd(dw + (int64_t)m * d.C + ccb + j, acc[a][j][i]);
            }
    }
    if (d.db != nullptr && ct == 0) {                        // (block-uniform) the eight position octets of a channel: summed in fixed order
        __syncthreads();
        float* bs = (float*)&FR2[0][0][0][0][0];               // [8][128]
        if (opnd == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[g8 * 128 + 8 * o + j] = bsum[j];
        }
        __syncthreads();
        if (tid < 128 && m0 + tid < d.M) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) sacc += bs[g * 128 + tid];
            if (d.slabs) d.slabs[(int64_t)chunk * p.sl_stride + p.w_n + m0 + tid] = sacc;
            else atomicAdd(d.db + m0 + tid, sacc);
        }
    }
}

// 256 (m) x 256 (c) tile on eight waves for the wide layers: the 128 x 128 tile moves 32 KB of operands per 2.1 MFLOP step
// (64 flop/B: the first decoder's weight gradient would pull 42 GB through L2 -- it ran at 6.6 TB/s of fragment traffic, 427 TF/s);
// this one moves 64 KB per 8.4 MFLOP.  Same scheme: 512 threads stage one 8 x 8 block each (threads 0-255 dy, 256-511 x; channel
// half = bit 4 of the octet index), wave w = (m half, c half, j quad) owns j = 4*jq .. 4*jq+3 against all eight j'.
__global__ __launch_bounds__(512, 2) void aero_conv_wgrad256_kernel(AeroWgradK p) {
    h16* FR = (h16*)AERO_DYN_SMEM;                             // [step parity][operand][half][k half][j][lane * 8]: 2 x 64 KiB
    const aero_wgrad_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    // consecutive tiles of a row chunk share dy / x slices: give each XCD (private L2) a contiguous run of them
    int id = (p.noswz & 1) ? (int)blockIdx.x : aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int mt = id % p.nmt; id /= p.nmt;
    const int ct = id % p.nct;
    const int chunk = id / p.nct;
    const int m0 = mt * 256, c0 = ct * 256;                   // (columns = (tap, c) flattened: see the 128 x 128 kernel)
    const int opnd = tid >> 8, o32 = tid & 31, g8 = (tid >> 5) & 7;
    const int half = o32 >> 4, o = o32 & 15;
    const int col0 = c0 + 128 * half + 8 * o;                 // (x side) first of this thread's 8 columns
    const int tap_l = opnd ? (col0 < p.ncol ? col0 / d.C : 0) : 0;
    const int dtj = opnd ? d.dt[tap_l] : 0, dfj = opnd ? d.df[tap_l] : 0;
    const int nrows = d.B * d.Fout;
    const int nT = (d.T + 63) >> 6;
    const int it_lo = chunk * p.SC;                           // chunks are runs of the linear step index (row, 64-step segment): a tensor
    const int it_hi = it_lo + p.SC < nrows * nT ? it_lo + p.SC : nrows * nT;   // of few long rows still fills the chip
    const h16* zpv = aero_zero_page;
    const int ch = opnd ? col0 - tap_l * d.C : m0 + 128 * half + 8 * o;
    const bool ch_ok = opnd ? col0 < p.ncol : ch < d.M;
    const int mh = wave >> 2, chh = (wave >> 1) & 1, jq = wave & 1;
    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool do_bias = d.db != nullptr && ct == 0 && opnd == 0;
    h16x8 r[8];
    // per-lane element offsets inside a 64-step segment, fixed for the whole kernel: an interior step (all 64 positions and the
    // shifted ones inside [0, T)) then costs no vector address arithmetic at all -- block-uniform row pointer + 32-bit lane offset.
    // (PMC on the first version: 7 vector + 6 scalar instructions per MFMA, the 64-bit per-load address / mask arithmetic twice
    // the transposes; the kernel was VALU-bound at 2x the MFMA time.)
    const int opu = aero_uniform(opnd);
    const int st_e = (int)(opu ? d.x_t : d.dy_t);
    const int coff = ch + 8 * g8 * st_e;                      // (the eight positions of a lane add block-uniform multiples of the step stride)
    auto load = [&](int it) {
        const int row = (it_lo + it) / nT, t0 = (it_lo + it - row * nT) * 64;
        const int b = row / d.Fout, fo = row - b * d.Fout;
        const int fi = fo * d.fstride + dfj;
        const bool row_ok = fi >= 0 && fi < d.Fin;
        const int sh = opu ? dtj : 0;
        if (t0 + 64 <= d.T && (!opu || (t0 + p.dt_min >= 0 && t0 + 64 + p.dt_max <= d.T))) {     // (block-uniform: every tap's shift stays inside the row)
            const bool live = ch_ok && (!opu || row_ok);        // a frequency offset outside the input: this lane's operand is zero
            const h16* rowp = (opu ? (const h16*)d.x + (int64_t)b * d.x_b + (int64_t)(row_ok ? fi : 0) * d.x_f
                                   : (const h16*)d.dy + (int64_t)b * d.dy_b + (int64_t)fo * d.dy_f) + (int64_t)(t0 + sh) * st_e;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = live ? *(const h16x8*)(rowp + (int64_t)i * st_e + coff) : (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            return;
        }
        const h16* base = opnd ? (const h16*)d.x + (int64_t)b * d.x_b + (int64_t)fi * d.x_f + ch
                               : (const h16*)d.dy + (int64_t)b * d.dy_b + (int64_t)fo * d.dy_f + ch;
        const int64_t st = opnd ? d.x_t : d.dy_t;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + 8 * g8 + i;
            const int tt = opnd ? t + dtj : t;
            const bool ok = ch_ok && t < d.T && (!opnd || (row_ok && tt >= 0 && tt < d.T));
            r[i] = *(const h16x8*)(ok ? base + (int64_t)tt * st : zpv);
        }
    };
    // fragment (operand, half, kh, j) at FR + ((((operand * 2 + half) * 2 + kh) * 8 + j) * 64 + lane) * 8
    h16* wr = FR + ((((opnd * 2 + half) * 2 + (g8 >> 2)) * 8) * 64 + (o + 16 * (g8 & 3))) * 8;
    const h16* ra = FR + (((0 * 2 + mh) * 2) * 8 * 64 + lane) * 8;
    const h16* rb = FR + (((1 * 2 + chh) * 2) * 8 * 64 + lane) * 8;
    const int nit = it_hi - it_lo;
    if (nit > 0) load(0);
    for (int it = 0; it < nit; ++it) {
        h16x8 c[8];
        aero_transpose8x8(r, c);
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[j] += (float)c[j][e];
        }
        const int par = (it & 1) * 32768;                      // halves: the other 64-KiB image (one barrier per step)
#pragma unroll
        for (int j = 0; j < 8; ++j) *(h16x8*)(wr + par + j * 512) = c[j];
        __syncthreads();
        if (it + 1 < nit && !(p.noswz & 4)) load(it + 1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            h16x8 af[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = *(const h16x8*)(ra + par + (kh * 8 + 4 * jq + a) * 512);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const h16x8 bf = *(const h16x8*)(rb + par + (kh * 8 + j) * 512);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf, acc[a][j], 0, 0, 0);
            }
        }
    }
    const int colb = c0 + 128 * chh + 8 * (lane & 15);
    const int tap_o = colb < p.ncol ? colb / d.C : 0, ccb = colb - tap_o * d.C;
    if (d.slabs) {
        float* sl = d.slabs + (int64_t)chunk * p.sl_stride + (int64_t)tap_o * d.M * d.C;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 128 * mh + 8 * ((lane >> 4) * 4 + i) + 4 * jq + a;
                if (m < d.M && colb < p.ncol) {
                    *(f32x4*)(sl + (int64_t)m * d.C + ccb) = (f32x4){acc[a][0][i], acc[a][1][i], acc[a][2][i], acc[a][3][i]};
                    *(f32x4*)(sl + (int64_t)m * d.C + ccb + 4) = (f32x4){acc[a][4][i], acc[a][5][i], acc[a][6][i], acc[a][7][i]};
                }
            }
    } else {
    float* dw = d.dw + (int64_t)tap_o * d.M * d.C;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 128 * mh + 8 * ((lane >> 4) * 4 + i) + 4 * jq + a;
                if (m < d.M && colb < p.ncol && !(p.noswz & 2)) atomicAdd(dw + (int64_t)m * d.C + ccb + j, acc[a][j][i]);
            }
    }
    if (d.db != nullptr && ct == 0) {
        __syncthreads();
        float* bs = (float*)FR;                                // [8][256]
        if (opnd == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[g8 * 256 + 8 * o32 + j] = bsum[j];
        }
        __syncthreads();
        if (tid < 256 && m0 + tid < d.M) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) sacc += bs[g * 256 + tid];
            if (d.slabs) d.slabs[(int64_t)chunk * p.sl_stride + p.w_n + m0 + tid] = sacc;
            else atomicAdd(d.db + m0 + tid, sacc);
        }
    }
}

// dw[e] += slab_0[e] + slab_1[e] + ... (e < w_n) and db[e - w_n] += ... (the bias partials behind each chunk's tile), in a FIXED
// order: thread (q, s) adds slabs s, s + 8, ... of element quad q (independent loads, four in flight), the eight lane sums are then
// added in lane order.  (One thread walking all slabs of its quad was a chain of nslab dependent L2 / HBM round trips.)
struct AeroWgradFinishK {
    const float* slabs; float* dw; float* db;
    int64_t stride, w_n, n;
    int nslab, store, layout, ntaps, MC, C, rowlen, coff;
};
__global__ __launch_bounds__(256) void aero_wgrad_finish_kernel(AeroWgradFinishK p) {
    __shared__ f32x4 part[8][32];
    const int q = threadIdx.x & 31, s = threadIdx.x >> 5;
    const int64_t e = ((int64_t)blockIdx.x * 32 + q) * 4;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (e < p.n) {
        const float* src = p.slabs + e;
        int k = s;
        for (; k + 24 < p.nslab; k += 32) {
            const f32x4 a0 = *(const f32x4*)(src + (int64_t)k * p.stride);
            const f32x4 a1 = *(const f32x4*)(src + (int64_t)(k + 8) * p.stride);
            const f32x4 a2 = *(const f32x4*)(src + (int64_t)(k + 16) * p.stride);
            const f32x4 a3 = *(const f32x4*)(src + (int64_t)(k + 24) * p.stride);
            v += a0; v += a1; v += a2; v += a3;
        }
        for (; k < p.nslab; k += 8) v += *(const f32x4*)(src + (int64_t)k * p.stride);
    }
    part[s][q] = v;
    __syncthreads();
    if (s == 0 && e < p.n) {
        f32x4 t = part[0][q];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += part[i][q];
        if (e >= p.w_n || p.layout == 0) {
            float* dst = e < p.w_n ? p.dw + e : p.db + (e - p.w_n);
            if (p.store) *(f32x4*)dst = t;
            else *(f32x4*)dst += t;
        } else {                                               // the weight's own layout: [m][c][tap] (C is a multiple of 8: a quad stays in one row)
            const int tap = (int)(e / p.MC);
            const int r = (int)(e - (int64_t)tap * p.MC);
            const int m = r / p.C, c = r - m * p.C;
            float* dst = p.dw + ((int64_t)m * p.rowlen + p.coff + c) * p.ntaps + tap;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (p.store) dst[(int64_t)i * p.ntaps] = t[i];
                else dst[(int64_t)i * p.ntaps] += t[i];
            }
        }
    }
}

// Tile size and number of position chunks of a weight-gradient launch (also exported: the caller sizes `slabs` with it).
// Chunks are runs of 64-position steps.  Enough blocks to fill the chip several times over (a block's steps are a chain of
// load -> transpose -> exchange -> MFMA round trips, ~1.7 us each: parallelism is what hides them), but a chunk's partial tile
// (written, then read back by the finish kernel) stays under ~1/4 of the operand bytes the chunk reads.
static void aero_wgrad_plan(int M, int C, int ntaps, int nrows, int T, bool* big_out, int* nchunk_out, int* SC_out) {
    const char* e256 = getenv("AERO_WGRAD_256");             // (read per call: tests/op_cases.py forces the 256 tile on small shapes)
    const int mode = e256 ? atoi(e256) : 1;
    // AERO_WGRAD_256: 0 never, 2 whenever both sides are >= 192 channels (tests), default: only the widest layers -- measured
    // D0 (1536 x 768) 427 -> 510 TF/s, D1 (768 x 384) 380 -> 366: with one 8-wave block per CU the smaller problem has too few tiles
    const bool big = mode && M >= 192 && C >= 192 && (mode == 2 || (long)M * C >= 768L * 1024);   // (on C, not ntaps * C: the measured crossover)
    const int TS = big ? 256 : 128;
    const long ncol = (long)ntaps * C;                       // the column index of a tile is (tap, c)
    const long tiles = (long)((M + TS - 1) / TS) * ((ncol + TS - 1) / TS);
    const long nsteps = (long)nrows * ((T + 63) / 64);
    const long Mc = M < TS ? M : TS, Cc = ncol < TS ? ncol : TS;
    long min_steps = (Mc * Cc + 4 * (Mc + Cc) - 1) / (4 * (Mc + Cc));
    if (min_steps < 4) min_steps = 4;
    long nchunk = ((big ? 1024 : 2048) + tiles - 1) / tiles;
    if (nchunk > nsteps / min_steps) nchunk = nsteps / min_steps;
    if (nchunk > 512) nchunk = 512;
    if (nchunk < 1) nchunk = 1;
    const long SC = (nsteps + nchunk - 1) / nchunk;
    *big_out = big;
    *SC_out = (int)SC;
    *nchunk_out = (int)((nsteps + SC - 1) / SC);
}

static int aero_conv_wgrad_launch(const aero_wgrad_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->dy || !d->x || !d->dw) { *err = "wgrad: null pointer"; return AERO_ERR_ARG; }
    if (d->ntaps < 1 || d->ntaps > 9 || d->B < 1 || d->Fin < 1 || d->Fout < 1 || d->T < 1 || d->M < 1 || d->C < 1 || d->fstride < 1) {
        *err = "wgrad: bad geometry"; return AERO_ERR_ARG;
    }
    if (d->slabs && (d->nslab < 1 || ((uintptr_t)d->slabs & 15) || ((uintptr_t)d->dw & 15) || ((uintptr_t)d->db & 15))) {
        *err = "wgrad: slabs / dw / db must be 16-byte aligned, nslab >= 1"; return AERO_ERR_ARG;
    }
    if ((d->M % 8) || (d->C % 8) || (d->dy_b % 8) || (d->dy_f % 8) || (d->dy_t % 8) || (d->x_b % 8) || (d->x_f % 8) || (d->x_t % 8) ||
        ((uintptr_t)d->dy & 15) || ((uintptr_t)d->x & 15)) {
        *err = "wgrad: channel counts and strides must be multiples of 8 (16-byte aligned channel vectors)"; return AERO_ERR_UNSUPPORTED;
    }
    if ((int64_t)(d->T + 64) * d->dy_t + d->M + 256 > 0x7fffffffLL || (int64_t)(d->T + 64) * d->x_t + d->C + 256 > 0x7fffffffLL) { *err = "wgrad: rows too long for 32-bit in-row offsets"; return AERO_ERR_UNSUPPORTED; }
    if ((int64_t)d->B * d->Fout * ((d->T + 63) / 64) > 0x3fffffffLL || (int64_t)d->ntaps * d->C > 0x3fffffffLL) { *err = "wgrad: too many positions"; return AERO_ERR_UNSUPPORTED; }
    if (!d->slabs && (d->store || d->dw_layout)) { *err = "wgrad: store / dw_layout need the slab workspace"; return AERO_ERR_ARG; }
    if (d->dw_layout < 0 || d->dw_layout > 1 || d->dw_rowlen < 0 || d->dw_coff < 0 || (d->dw_rowlen && d->dw_coff + d->C > d->dw_rowlen) ||
        (!d->dw_rowlen && d->dw_coff) || (int64_t)d->M * d->C > 0x7fffffffLL) { *err = "wgrad: bad destination layout"; return AERO_ERR_ARG; }
    AeroWgradK p;
    p.d = *d;
    static const int abl = [] { const char* e = getenv("AERO_WGRAD_ABL"); return e ? atoi(e) : 0; }();
    p.noswz = abl;                                            // ablation bits (timing experiments): 1 no XCD swizzle, 2 no atomics, 4 no loads after the first step
    bool big;
    int nchunk, SC;
    const int nrows = d->B * d->Fout;
    aero_wgrad_plan(d->M, d->C, d->ntaps, nrows, d->T, &big, &nchunk, &SC);
    if (d->slabs && nchunk > d->nslab) {                      // a smaller workspace than planned: longer chunks
        const long nsteps = (long)nrows * ((d->T + 63) / 64);
        SC = (int)((nsteps + d->nslab - 1) / d->nslab);
        nchunk = (int)((nsteps + SC - 1) / SC);
    }
    const int TS = big ? 256 : 128;
    p.ncol = d->ntaps * d->C;
    p.nmt = (d->M + TS - 1) / TS;
    p.nct = (p.ncol + TS - 1) / TS;
    p.dt_min = p.dt_max = d->dt[0];
    for (int j = 1; j < d->ntaps; ++j) {
        if (d->dt[j] < p.dt_min) p.dt_min = d->dt[j];
        if (d->dt[j] > p.dt_max) p.dt_max = d->dt[j];
    }
    p.SC = SC;
    p.nchunk = nchunk;
    p.w_n = (int64_t)d->ntaps * d->M * d->C;
    p.sl_stride = p.w_n + (d->db ? d->M : 0);
    const long nb = (long)p.nmt * p.nct * p.nchunk;
    if (nb > 0x7fffffffL) { *err = "wgrad: grid too large"; return AERO_ERR_ARG; }
    if (big) AERO_LAUNCH_DYN(aero_conv_wgrad256_kernel, dim3((unsigned)nb), dim3(512), (size_t)128 * 1024, stream, p);
    else AERO_LAUNCH(aero_conv_wgrad_kernel, dim3((unsigned)nb), dim3(256), stream, p);
    if (d->slabs) {
        AeroWgradFinishK f;
        f.slabs = d->slabs; f.dw = d->dw; f.db = d->db;
        f.stride = p.sl_stride; f.w_n = p.w_n; f.n = p.sl_stride;
        f.nslab = p.nchunk; f.store = d->store; f.layout = d->dw_layout; f.ntaps = d->ntaps; f.MC = d->M * d->C; f.C = d->C;
        f.rowlen = d->dw_rowlen ? d->dw_rowlen : d->C; f.coff = d->dw_coff;
        const int64_t fb = (f.n / 4 + 31) / 32;
        AERO_LAUNCH(aero_wgrad_finish_kernel, dim3((unsigned)fb), dim3(256), stream, f);
    }
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm + activation backward.   forward (aero_norm_apply):  xh = (x - mean) * rstd,  u = xh * gamma + beta,
//   y = GELU(u)   |   y[c] = u[c] * sigmoid(u[c + C/2]) * layer_scale[c]  (GLU)   |   y = u
// backward:  du = dy * act'(u);  dgamma[c] += sum du * xh;  dbeta[c] += sum du;  dxh = du * gamma;
//   dx = rstd * (dxh - S1/N - xh * S2/N)   with   S1 = sum_group dxh,  S2 = sum_group dxh * xh   over the (item, group).
// Two streaming passes over (x, dy): `reduce` adds S1, S2 (fp64) and dgamma / dbeta / dlayer_scale (fp32) with one atomic per
// block and channel; `apply` recomputes du and writes dx (fp16).  Thread (v, ty): channel vector v (8 channels; for GLU also the
// matching 8 gate channels) of time steps ty, ty + TY, ...  Groups must be at least 8 channels wide (a vector then spans at
// most two groups); per_row 0 / 1 as in aero_norm_desc.
struct AeroNormBwdK {
    aero_norm_bwd_desc d;
    int tchunk, apply;
};

template <bool APPLY, int ACT>
__global__ __launch_bounds__(256) void aero_norm_bwd_kernel(AeroNormBwdK p) {
    // block-level sums in fp64: the order in which waves (LDS atomics) and blocks (global atomics) arrive changes an fp64 sum of fp32
    // partials by ~1e-16 relative -- invisible once it is rounded to fp32 -- where fp32 atomics made the parameter gradients and, through
    // the group sums, dx differ from run to run by ~1e-7, which 30 Adam steps amplify to percents of the loss (tests/test_gpu_train.py)
    __shared__ double red_c[3][2048];                          // dgamma, dbeta (all C channels), dlayer_scale (C/2)
    __shared__ double red_g[2][256];                           // S1, S2 per group
    __shared__ double red_a;                                   // d snake_a[f] of the current item
    const aero_norm_bwd_desc& d = p.d;
    constexpr bool glu = ACT == AERO_ACT_GLU;                 // (one instantiation per activation: the unused halves and paths cost registers)
    const int Cout = glu ? d.C / 2 : d.C;
    const int vpp = Cout / 8;
    const int TY = 256 / vpp;
    const int tid = threadIdx.x;
    const int v = tid % vpp, ty = tid / vpp;
    const int gs = d.C / d.G;
    const int nh = glu ? 2 : 1;
    if (!APPLY) {
        for (int i = tid; i < d.C; i += 256) { red_c[0][i] = 0.0; red_c[1][i] = 0.0; }
        for (int i = tid; i < Cout; i += 256) red_c[2][i] = 0.0;
    }
    float dgam[2][8], dbet[2][8], dls[8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) dgam[h][i] = dbet[h][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) dls[i] = 0.f;
    if (APPLY && d.psums && blockIdx.x == 0) {
        // the reduce pass (finished: same stream) left the parameter-gradient sums in fp64: add them, rounded once, to the caller's fp32
        // destinations (one thread per element: nothing else writes these ranges while this kernel runs)
        for (int c = tid; c < d.C; c += 256) {
            if (d.dgamma) d.dgamma[c] += (float)d.psums[c];
            if (d.dbeta) d.dbeta[c] += (float)d.psums[d.C + c];
        }
        if (glu && d.dlayer_scale)
            for (int c = tid; c < Cout; c += 256) d.dlayer_scale[c] += (float)d.psums[2 * (int64_t)d.C + c];
        if (ACT == AERO_ACT_SNAKE && d.dsnake_a)
            for (int f = tid; f < d.F; f += 256) d.dsnake_a[f] += (float)d.psums[3 * (int64_t)d.C + f];
    }
    // work items (b, f, time chunk), grid-strided: the parameter gradients stay in registers across a block's items and reach
    // memory once per block (one block per item made 8192 blocks x 2C same-line atomics: 12 ms for the last decoder's norm)
    const int ntch = (d.T + p.tchunk - 1) / p.tchunk;
    const int nwork = d.B * d.F * ntch;
    for (int work = (int)blockIdx.x; work < nwork; work += (int)gridDim.x) {
    const int tch = work % ntch;
    const int f = (work / ntch) % d.F;
    const int b = work / (ntch * d.F);
    const int item = d.per_row == 1 ? b * d.F + f : (d.per_row == 2 ? 0 : b);
    const bool bn = d.per_row == 2;                            // BatchNorm on batch statistics: group = channel over the whole batch
    if (!APPLY) {
        __syncthreads();                                       // the previous item's group sums have been flushed
        for (int i = tid; i < (bn ? 0 : d.G); i += 256) { red_g[0][i] = 0.0; red_g[1][i] = 0.0; }
        if (tid == 0) red_a = 0.0;
        __syncthreads();
    }
    const float inv_count = (float)(1.0 / d.stat_count);
    // per owned channel (both halves): xh = x * rs + mr,  u = xh * gm + bt;  for APPLY the group terms k1 = S1/N, k2 = S2/N.
    // The statistics become (rstd, -mean * rstd) once per GROUP the vector touches (at most two per half): fp64 only for
    // E[x^2] - mean^2, then a float rsqrt with one Newton step (as aero_norm_apply_kernel)
    float rs[2][8], mr[2][8], gm[2][8], bt[2][8], k1[2][8], k2[2][8], ls[8];
    int grp[2][8];
    if (ty < TY) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h >= nh) continue;
            int g_prev = -1;
            float g_r = 0.f, g_m = 0.f, g_k1 = 0.f, g_k2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = h * Cout + v * 8 + i;
                const int g = c / gs;
                if (!d.stats) {                                 // identity norm (layers before norm_starts, aero.py:56,148)
                    g_r = 1.f;
                } else if (g != g_prev) {
                    g_prev = g;
                    const double* st = d.stats + ((int64_t)item * d.G + g) * 2;
                    const double mean = st[0] / d.stat_count;
                    double var = st[1] / d.stat_count - mean * mean;
                    if (var < 0) var = 0;
                    const float vf = (float)var + d.eps;
                    float r = aero_rsqrt(vf);
                    r = r * (1.5f - 0.5f * vf * r * r);
                    g_r = r;
                    g_m = -(float)mean * r;
                    if (APPLY && !bn) {
                        const double* sm = d.sums + ((int64_t)item * d.G + g) * 2;
                        g_k1 = (float)sm[0] * inv_count;
                        g_k2 = (float)sm[1] * inv_count;
                    } else if (APPLY) {
                        // per-channel groups: S1 = gamma * dbeta, S2 = gamma * dgamma -- the reduce pass's parameter gradients ARE the sums
                        const float sb = d.psums ? (float)d.psums[d.C + c] : d.dbeta[c];
                        const float sg = d.psums ? (float)d.psums[c] : d.dgamma[c];
                        g_k1 = d.gamma[c] * sb * inv_count;
                        g_k2 = d.gamma[c] * sg * inv_count;
                    }
                }
                rs[h][i] = g_r;
                mr[h][i] = g_m;
                k1[h][i] = g_k1;
                k2[h][i] = g_k2;
                gm[h][i] = d.gamma ? d.gamma[c] : 1.f;
                bt[h][i] = d.beta ? d.beta[c] : 0.f;
                grp[h][i] = g;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) ls[i] = (glu && d.layer_scale) ? d.layer_scale[v * 8 + i] : 1.f;
    }
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) s1[h][i] = s2[h][i] = 0.f;
    const int t0 = tch * p.tchunk;
    const int t1 = t0 + p.tchunk < d.T ? t0 + p.tchunk : d.T;
    const h16* xs = (const h16*)d.x + (int64_t)b * d.x_b + (int64_t)f * d.x_f + v * 8;
    const h16* dys = (const h16*)d.dy + (int64_t)b * d.dy_b + (int64_t)f * d.dy_f + v * 8;
    h16* dxs = (h16*)d.dx + (int64_t)b * d.dx_b + (int64_t)f * d.dx_f + v * 8;
    const float sn_a = ACT == AERO_ACT_SNAKE ? d.snake_a[f] : 1.f, sn_ia = 1.f / sn_a;
    float dsn = 0.f;
    // one time step
    auto elem = [&](const h16x8& xa, const h16x8& xg, const h16x8& dyv, h16x8& oa, h16x8& og) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh0 = (float)xa[i] * rs[0][i] + mr[0][i];
            const float u0 = xh0 * gm[0][i] + bt[0][i];
            const float g_out = (float)dyv[i];
            float du0, du1 = 0.f, xh1 = 0.f;
            if (glu) {
                xh1 = (float)xg[i] * rs[1][i] + mr[1][i];
                const float u1 = xh1 * gm[1][i] + bt[1][i];
                const float sg = aero_sigmoid(u1);
                const float gy = g_out * ls[i];
                du0 = gy * sg;
                du1 = gy * u0 * sg * (1.f - sg);
                if (!APPLY) dls[i] += g_out * u0 * sg;
            } else if (ACT == AERO_ACT_GELU) {
                const float cdf = 0.5f * (1.0f + aero_erf(u0 * 0.70710678118654752f));
                const float pdf = 0.3989422804014327f * aero_fast_exp(-0.5f * u0 * u0);
                du0 = g_out * (cdf + u0 * pdf);
            } else if (ACT == AERO_ACT_RELU) {
                du0 = u0 > 0.f ? g_out : 0.f;
            } else if (ACT == AERO_ACT_SNAKE) {               // y = u + sin^2(a u) / a  (snake.py:67), a = snake_a[f]
                const float sn = aero_fast_sin(sn_a * u0), cs = aero_fast_cos(sn_a * u0);     // (hardware sin / cos as in the forward: the libm forms cost 70 registers)
                du0 = g_out * (1.f + 2.f * sn * cs);
                if (!APPLY) dsn += g_out * (2.f * u0 * sn * cs - sn * sn * sn_ia) * sn_ia;
            } else {
                du0 = g_out;
            }
            const float dxh0 = du0 * gm[0][i], dxh1 = du1 * gm[1][i];
            if (!APPLY) {
                dgam[0][i] += du0 * xh0; dbet[0][i] += du0;
                s1[0][i] += dxh0; s2[0][i] += dxh0 * xh0;
                if (glu) {
                    dgam[1][i] += du1 * xh1; dbet[1][i] += du1;
                    s1[1][i] += dxh1; s2[1][i] += dxh1 * xh1;
                }
            } else {
                oa[i] = (h16)(rs[0][i] * (dxh0 - k1[0][i] - xh0 * k2[0][i]));
                if (glu) og[i] = (h16)(rs[1][i] * (dxh1 - k1[1][i] - xh1 * k2[1][i]));
            }
        }
    };
    if (ty < TY) {
        // UNR steps per trip, every load issued before the first use
        constexpr int UNR = 4;
        for (int tb = t0 + ty; tb < t1; tb += UNR * TY) {
            h16x8 xa[UNR], xg[UNR], dyv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int t = tb + u * TY;
                const int tc = t < t1 ? t : t1 - 1;           // clamped address; the result of a step past the end is dropped
                xa[u] = *(const h16x8*)(xs + (int64_t)tc * d.x_t);
                xg[u] = glu ? *(const h16x8*)(xs + (int64_t)tc * d.x_t + Cout) : xa[u];
                dyv[u] = *(const h16x8*)(dys + (int64_t)tc * d.dy_t);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int t = tb + u * TY;
                if (t >= t1) break;
                h16x8 oa, og;
                elem(xa[u], xg[u], dyv[u], oa, og);
                if (APPLY) {
                    *(h16x8*)(dxs + (int64_t)t * d.dx_t) = oa;
                    if (glu) *(h16x8*)(dxs + (int64_t)t * d.dx_t + Cout) = og;
                }
            }
        }
    }
    if (APPLY) continue;
    if (ty < TY) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h >= nh) continue;
            // the vector's group sums: one LDS atomic per group it touches, not per channel
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a1 += s1[h][i];
                a2 += s2[h][i];
                if (!bn && (i == 7 || grp[h][i + 1 < 8 ? i + 1 : 7] != grp[h][i])) {
                    atomicAdd(&red_g[0][grp[h][i]], (double)a1);
                    atomicAdd(&red_g[1][grp[h][i]], (double)a2);
                    a1 = a2 = 0.f;
                }
            }
        }
    }
    if (ACT == AERO_ACT_SNAKE && d.dsnake_a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dsn += __shfl_xor(dsn, o);
        if ((tid & 63) == 0) atomicAdd(&red_a, (double)dsn);
    }
    __syncthreads();
    if (d.stats && !bn)
        for (int g = tid; g < d.G; g += 256) {
            double* sm = d.sums + ((int64_t)item * d.G + g) * 2;
            atomicAdd(sm, red_g[0][g]);
            atomicAdd(sm + 1, red_g[1][g]);
        }
    if (tid == 0 && ACT == AERO_ACT_SNAKE && d.dsnake_a) {
        if (d.psums) atomicAdd(d.psums + 3 * (int64_t)d.C + f, red_a);
        else atomicAdd(d.dsnake_a + f, (float)red_a);
    }
    }   // work items
    if (APPLY) return;
    __syncthreads();
    if (ty < TY) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h >= nh) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = h * Cout + v * 8 + i;
                atomicAdd(&red_c[0][c], (double)dgam[h][i]);
                atomicAdd(&red_c[1][c], (double)dbet[h][i]);
            }
        }
        if (glu && d.dlayer_scale) {
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(&red_c[2][v * 8 + i], (double)dls[i]);
        }
    }
    __syncthreads();
    if (d.psums) {                                             // fp64 staging [dgamma C | dbeta C | dlayer_scale C | dsnake_a F]: the apply pass rounds once
        for (int c = tid; c < d.C; c += 256) {
            if (d.dgamma) atomicAdd(d.psums + c, red_c[0][c]);
            if (d.dbeta) atomicAdd(d.psums + d.C + c, red_c[1][c]);
        }
        if (glu && d.dlayer_scale)
            for (int c = tid; c < Cout; c += 256) atomicAdd(d.psums + 2 * (int64_t)d.C + c, red_c[2][c]);
        return;
    }
    for (int c = tid; c < d.C; c += 256) {
        if (d.dgamma) atomicAdd(d.dgamma + c, (float)red_c[0][c]);
        if (d.dbeta) atomicAdd(d.dbeta + c, (float)red_c[1][c]);
    }
    if (glu && d.dlayer_scale)
        for (int c = tid; c < Cout; c += 256) atomicAdd(d.dlayer_scale + c, (float)red_c[2][c]);
}

static int aero_norm_bwd_launch(const aero_norm_bwd_desc* d, int apply, hipStream_t stream, const char** err) {
    if (!d || !d->x || !d->dy || (d->stats && !d->sums && d->per_row != 2) || (apply && !d->dx)) { *err = "norm_bwd: null pointer"; return AERO_ERR_ARG; }
    if (d->act == AERO_ACT_SNAKE && !d->snake_a) { *err = "norm_bwd: Snake needs snake_a"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->F < 1 || d->T < 1 || d->C < 8 || d->G < 1 || d->C % d->G || (d->stats && d->stat_count <= 0)) { *err = "norm_bwd: bad geometry"; return AERO_ERR_ARG; }
    const bool glu = d->act == AERO_ACT_GLU;
    const int Cout = glu ? d->C / 2 : d->C;
    if (d->per_row < 0 || d->per_row > 2) { *err = "norm_bwd: per_row must be 0, 1 or 2"; return AERO_ERR_ARG; }
    const bool bn = d->per_row == 2;
    if (bn && (d->G != d->C || !d->stats || !d->gamma || !d->dgamma || !d->dbeta || glu || d->act == AERO_ACT_SNAKE)) {
        *err = "norm_bwd: batch statistics (per_row 2) need G == C, stats, gamma and the dgamma / dbeta buffers; no GLU / Snake"; return AERO_ERR_UNSUPPORTED;
    }
    if (d->act != AERO_ACT_NONE && d->act != AERO_ACT_GELU && d->act != AERO_ACT_GLU && d->act != AERO_ACT_RELU && d->act != AERO_ACT_SNAKE) { *err = "norm_bwd: activation not supported"; return AERO_ERR_UNSUPPORTED; }
    if ((Cout % 8) || Cout / 8 > 256 || d->C > 2048 || (!bn && (d->G > 256 || (d->C / d->G) < 8)) || (d->x_b % 8) || (d->x_f % 8) || (d->x_t % 8) || (d->dy_b % 8) || (d->dy_f % 8) ||
        (d->dy_t % 8) || ((uintptr_t)d->x & 15) || ((uintptr_t)d->dy & 15)) {
        *err = "norm_bwd: needs 8-channel aligned fp16 rows, C <= 2048, groups of >= 8 channels"; return AERO_ERR_UNSUPPORTED;
    }
    if (apply && ((d->dx_b % 8) || (d->dx_f % 8) || (d->dx_t % 8) || ((uintptr_t)d->dx & 15))) { *err = "norm_bwd: dx must be 16-byte aligned"; return AERO_ERR_UNSUPPORTED; }
    AeroNormBwdK p;
    p.d = *d;
    p.apply = apply;
    const int TY = 256 / (Cout / 8);
    int tchunk = TY * 16;                                      // 16 time steps per thread (four trips of four)
    if (tchunk > d->T) tchunk = d->T;
    p.tchunk = tchunk;
    const long nwork = (long)((d->T + tchunk - 1) / tchunk) * d->F * d->B;
    if (nwork > 0x7fffffffL) { *err = "norm_bwd: too many work items"; return AERO_ERR_ARG; }
    dim3 grid((unsigned)(apply ? nwork : (nwork < 1024 ? nwork : 1024)));
#define AERO_NB_LAUNCH(A)                                                                               \
    do {                                                                                                \
        if (apply) AERO_LAUNCH((aero_norm_bwd_kernel<true, A>), grid, dim3(256), stream, p);            \
        else AERO_LAUNCH((aero_norm_bwd_kernel<false, A>), grid, dim3(256), stream, p);                 \
    } while (0)
    switch (d->act) {
        case AERO_ACT_RELU: AERO_NB_LAUNCH(AERO_ACT_RELU); break;
        case AERO_ACT_GELU: AERO_NB_LAUNCH(AERO_ACT_GELU); break;
        case AERO_ACT_GLU: AERO_NB_LAUNCH(AERO_ACT_GLU); break;
        case AERO_ACT_SNAKE: AERO_NB_LAUNCH(AERO_ACT_SNAKE); break;
        default: AERO_NB_LAUNCH(AERO_ACT_NONE); break;
    }
#undef AERO_NB_LAUNCH
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// iSTFT backward (the adjoint of aero_istft_fwd; aero.py:423-428, spec.py:30-37).  With g = dy * inv_env the gradient of frame t,
// bin k is  c_k * n_fft^-1/2 * sum_m w[m] g_full[t*hop + m] e^{-2 pi i k m / n_fft}  (c_0 = 1, c_k = 2: the Hermitian half of irfft;
// the imaginary part of DC has no gradient) -- i.e. the EXISTING forward STFT kernel applied to g, zero-padded by n_fft/2 + hop in
// front (the reflect padding of the centred STFT then mirrors zeros) and read at frame t + (n_fft/2 + hop)/hop.  Two small
// streaming kernels bracket that call: `prep` builds the padded g, `pack` applies c_k and the frame shift.
__global__ __launch_bounds__(256) void aero_istft_bwd_prep_kernel(const float* dy, const float* inv_env, float* s, int L, int Ls, int off, int env_off) {
    const int sig = blockIdx.y;
    for (int n = (int)blockIdx.x * 256 + threadIdx.x; n < Ls; n += (int)gridDim.x * 256) {
        const int j = n - off;
        s[(int64_t)sig * Ls + n] = (j >= 0 && j < L) ? dy[(int64_t)sig * L + j] * inv_env[j + env_off] : 0.f;
    }
}

__global__ __launch_bounds__(256) void aero_istft_bwd_pack_kernel(const f32x2* spec, f32x2* dz, int nbins, int Tsrc, int T, int t_off) {
    const int sig = blockIdx.z, k = blockIdx.y;
    const f32x2* src = spec + ((int64_t)sig * nbins + k) * Tsrc + t_off;
    f32x2* dst = dz + ((int64_t)sig * nbins + k) * T;
    for (int t = (int)blockIdx.x * 256 + threadIdx.x; t < T; t += (int)gridDim.x * 256) {
        const f32x2 v = src[t];
        dst[t] = k == 0 ? (f32x2){v[0], 0.f} : (f32x2){2.f * v[0], 2.f * v[1]};
    }
}

static int aero_istft_bwd_prep_launch(const float* dy, const float* inv_env, float* s, int nsig, int L, int Ls, int off, int env_off,
                                      hipStream_t stream, const char** err) {
    if (!dy || !inv_env || !s) { *err = "istft_bwd_prep: null pointer"; return AERO_ERR_ARG; }
    if (nsig < 1 || nsig > 65535 || L < 1 || off < 0 || env_off < 0 || Ls < off + L) { *err = "istft_bwd_prep: bad geometry"; return AERO_ERR_ARG; }
    int nb = (Ls + 255) / 256;
    if (nb > 64) nb = 64;
    AERO_LAUNCH(aero_istft_bwd_prep_kernel, dim3((unsigned)nb, (unsigned)nsig), dim3(256), stream, dy, inv_env, s, L, Ls, off, env_off);
    return AERO_OK;
}

static int aero_istft_bwd_pack_launch(const float* spec, float* dz, int nsig, int nbins, int Tsrc, int T, int t_off, hipStream_t stream, const char** err) {
    if (!spec || !dz) { *err = "istft_bwd_pack: null pointer"; return AERO_ERR_ARG; }
    if (nsig < 1 || nsig > 65535 || nbins < 1 || nbins > 65535 || T < 1 || t_off < 0 || t_off + T > Tsrc) { *err = "istft_bwd_pack: bad geometry"; return AERO_ERR_ARG; }
    int nb = (T + 255) / 256;
    AERO_LAUNCH(aero_istft_bwd_pack_kernel, dim3((unsigned)nb, (unsigned)nbins, (unsigned)nsig), dim3(256), stream, (const f32x2*)spec, (f32x2*)dz, nbins, Tsrc, T, t_off);
    return AERO_OK;
}
