// k_pw.h -- pointwise (1x1) convolution with a SHORT contraction (C <= 96 input channels) and a wide output, HBM-bound:
//   * the tail of a DConv layer that has a BLSTM / LocalState in it (modules.py:240-247 via aero.py:121-131): conv2 (hidden -> 2C)
//     -> GroupNorm(1 group, statistics already known from the Gram matrix of the input, k_gram.h) -> GLU -> LayerScale -> + x;
//   * the encoder's `rewrite` conv + GLU (+ frequency embedding at encoder 0) where no GroupNorm sits between them (aero.py:133).
// On the tiled conv kernels (k_conv.h) these launches ran at 0.9-2 TB/s: a 128-row x 128-step block does ONE K-chunk of MFMAs and then
// an epilogue of coefficient loads, two LDS transposes with four barriers and the residual fetch -- all exposed latency, three M-tiles
// re-reading the input.  Here there is no LDS staging of operands and no barrier in the loop at all:
//   * the weights are MFMA A-fragments RESIDENT IN REGISTERS (a wave owns GW groups of 64 rows: GW*4 tiles x KS k-steps, 16 bytes each);
//   * the B fragment of v_mfma_f32_16x16x32_f16 "lane = (time step n, k-octet q)" is one 16-byte load of x[t0 + n][32 ks + 8 q ..] straight
//     from global memory -- channels-last rows ARE the fragment layout;
//   * rows are PERMUTED when the weight image is packed (host, once per parameter version): tile j, row 4 q + i of a 64-row group holds
//     logical row 16 q + 4 j + i, so the 16 accumulator values of lane (n, q) over the group's four tiles are 16 CONSECUTIVE conv rows of
//     time step n -- after GLU 8 consecutive output channels = ONE 16-byte store (and one 16-byte residual load), no transpose;
//   * per-row coefficients (norm scale / shift, LayerScale, frequency embedding) are computed once per block into LDS (a block works
//     inside one (b, f) row, so GroupNorm's per-row statistics are block constants);
//   * the next unit's loads (input fragments, residual) are issued before the current unit's MFMAs.
// Block = 4 waves = 2 (row halves of the chunk) x 2 (alternate 16-step units); grid = (row, time split) x row chunks of 128*GW rows.
// Roofline: HBM.  Algorithmic bytes per output position: 2 C + 2 Mout (+ 2 Mout residual).
#pragma once
#include "aero_common.h"

struct AeroPwK {
    aero_pw_desc d;
    int nsplit, upb, nchunk;         // time splits per row, 16-step units per split, row chunks (blocks) per (row, split)
    int Mout;                        // stored channels: M / 2 with GLU
};

// WLDS: the weight fragments live in LDS instead of registers (contractions of 128 .. 384 channels: KS = 4 .. 12 k-steps of one 64-row group
// per wave are 64-192 registers); every MFMA's A operand is then one ds_read_b128 of the image in fragment order, everything else is
// unchanged.  8 KS KiB of LDS per block (three / one block per CU at KS 6 / 12).
template <int KS, int GW, int ACT, bool NORM, bool WLDS = false>
__global__ __launch_bounds__(256, 2) void aero_pw_kernel(AeroPwK p) {
    constexpr bool GLU = ACT == AERO_ACT_GLU;
    constexpr int DEPTH = (!WLDS && GW * KS >= 6) ? 1 : 2;        // units requested ahead (96 weight-fragment registers leave room for one)
    constexpr int MC = 128 * GW;                                  // conv rows per chunk (block)
    constexpr int OC = GLU ? MC / 2 : MC;                         // stored channels per chunk
    constexpr int NV = GLU ? 1 : 2;                               // 16-byte vectors a lane stores per group
    __shared__ AERO_LDS_ALIGN float ca[MC], cb[MC], cs[OC], cp[OC];
    const aero_pw_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave & 1, wt = wave >> 1;
    const int n = lane & 15, q = lane >> 4;
    // the chunks of one (row, split) are NEIGHBOURS in an XCD's share of the grid: they read the same input rows, the second and third
    // reader find them in that XCD's L2
    const int lin = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int rs = lin / p.nchunk, chunk = lin - rs * p.nchunk;
    const int row = rs / p.nsplit, split = rs - row * p.nsplit;
    const int b = row / d.F, f = row - b * d.F;
    const int M = d.M, Mout = p.Mout, T = d.T;
    const int mbase = chunk * MC, obase = chunk * OC;

    // ---- block constants: v = acc * a[r] + b[r]  (bias, GroupNorm from the row's sums), LayerScale, frequency embedding row
    {
        float mean = 0.f, rstd = 1.f;
        if constexpr (NORM) {
            const double* sp = d.stats + (int64_t)row * 2;
            const double inv = 1.0 / d.stat_count;
            const double mu = sp[0] * inv;
            double var = sp[1] * inv - mu * mu;
            if (var < 0) var = 0;
            const float vf = (float)var + d.stat_eps;
            float rs = aero_rsqrt(vf);
            rs = rs * (1.5f - 0.5f * vf * rs * rs);
            mean = (float)mu;
            rstd = rs;
        }
        for (int i = tid; i < MC; i += 256) {
            const int r = mbase + i < M ? mbase + i : M - 1;
            const float bias = d.bias ? d.bias[r] : 0.f;
            if constexpr (NORM) {
                const float gm = d.gamma ? d.gamma[r] * rstd : rstd;
                ca[i] = gm;
                cb[i] = (bias - mean) * gm + (d.gamma ? d.beta[r] : 0.f);
            } else {
                ca[i] = 1.f;
                cb[i] = bias;
            }
        }
        for (int i = tid; i < OC; i += 256) {
            const int o = obase + i < Mout ? obase + i : Mout - 1;
            cs[i] = d.layer_scale ? d.layer_scale[o] : 1.f;
            cp[i] = d.post_add ? d.post_add[(int64_t)f * Mout + o] : 0.f;
        }
    }
    // ---- the wave's weight fragments: image [chunk][wm][g][j][ks][lane][8]
    h16x8 A[WLDS ? 1 : GW][WLDS ? 1 : 4][WLDS ? 1 : KS];
    const h16* Wl = nullptr;                                      // WLDS: this wave's fragments in LDS, fragment (g, j, ks) at ((g*4 + j)*KS + ks)*512 + lane*8
    if constexpr (WLDS) {
        h16* Ws = (h16*)AERO_DYN_SMEM;
        const h16* w = (const h16*)d.wimg + ((int64_t)chunk * 2 * GW * 4 * KS) * 512;
        constexpr int NV = 2 * GW * 4 * KS * 64;                  // 16-byte vectors of the chunk's image (both row halves)
        for (int i = tid; i < NV; i += 256) *(h16x8*)(Ws + i * 8) = *(const h16x8*)(w + i * 8);
        Wl = Ws + (wm * GW * 4 * KS) * 512 + lane * 8;
    } else {
        const h16* w = (const h16*)d.wimg + ((int64_t)(chunk * 2 + wm) * GW * 4 * KS) * 512 + lane * 8;
#pragma unroll
        for (int g = 0; g < GW; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) A[g][j][ks] = *(const h16x8*)(w + ((g * 4 + j) * KS + ks) * 512);
    }
    __syncthreads();

    const h16* xr = (const h16*)d.x + (int64_t)b * d.x_b + (int64_t)f * d.x_f;
    // second source (torch.cat([x, x1], 1) in front of the conv: the FTB's conv2, modules.py:322-323): channels C0 .. C-1 come from x1
    const int C0 = d.x1 ? d.C0 : d.C;
    const h16* x1r = d.x1 ? (const h16*)d.x1 + (int64_t)b * d.x1_b + (int64_t)f * d.x1_f - C0 : xr;
    const int x1t = d.x1 ? (int)d.x1_t : (int)d.x_t;
    const h16* rr = d.res ? (const h16*)d.res + (int64_t)b * d.r_b + (int64_t)f * d.r_f : nullptr;
    h16* dr = (h16*)d.dst + (int64_t)b * d.d_b + (int64_t)f * d.d_f;
    const int xt = (int)d.x_t, rt = (int)d.r_t, dt = (int)d.d_t;
    const int u_lo = split * p.upb;
    int u_hi = u_lo + p.upb;
    const int nunit = (T + 15) >> 4;
    if (u_hi > nunit) u_hi = nunit;
    const h16x8 zero8 = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
    // first stored channel of this lane in group g (a lane's 8 (GLU) or 16 consecutive channels)
    const int lane_o = GLU ? 8 * q : 16 * q;
    const int grp_o = wm * GW * (GLU ? 32 : 64);

    // Loads are UNCONDITIONAL and their values are used as they come: a predicated load becomes a branch around it, a select-to-zero
    // behind a load is scheduled right behind it -- either way the compiler waited vmcnt(0) in front of the MFMAs, i.e. for the prefetch
    // it had just issued (1.5 TB/s).  Padding lanes (k >= C) re-read channels 0..7 of their step: their weight columns are zero in the
    // image, and finite x times zero is zero; lanes past Mout read a valid vector they never store.
    const int kq = 8 * q;
    auto load_b = [&](h16x8 (&Bf)[KS], int u) {
        int t = u * 16 + n;
        t = t < T ? t : T - 1;                                    // (masked at the store: the load stays in range)
        const h16* px = xr + t * xt;
        const h16* px1 = x1r + t * x1t;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + kq;
            Bf[ks] = *(const h16x8*)(k0 < C0 ? px + k0 : (k0 < d.C ? px1 + k0 : px));
        }
    };
    auto load_r = [&](h16x8 (&R)[GW][NV], int u) {
        if (!rr) return;                                          // (block-uniform)
        int t = u * 16 + n;
        t = t < T ? t : T - 1;
        const h16* pr = rr + t * rt;
#pragma unroll
        for (int g = 0; g < GW; ++g)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int o = obase + grp_o + g * (GLU ? 32 : 64) + lane_o + 8 * v;
                R[g][v] = *(const h16x8*)(pr + (o < Mout ? o : 0));
            }
    };

    // DEPTH + 1 fragment buffers, used in rotation by an unrolled loop: the unit computed in a trip was requested DEPTH trips (of this
    // wave) earlier.  (First form: two buffers and `Bc = Bn` at the end of the trip -- the copy needs the prefetched registers, so the
    // compiler waited vmcnt(0) there, for the loads AND the trip's store: a prefetch lived for one epilogue, ~0.3 us against ~2 us of
    // memory latency, and with 8-12 waves per CU the kernel ran at 3.0-3.3 TB/s where its epilogue is long (GLU).)
    h16x8 B[DEPTH + 1][KS], R[DEPTH + 1][GW][NV];
#pragma unroll
    for (int s_ = 0; s_ <= DEPTH; ++s_)
#pragma unroll
        for (int g = 0; g < GW; ++g)
#pragma unroll
            for (int v = 0; v < NV; ++v) R[s_][g][v] = zero8;
    int u = u_lo + wt;
    if (u >= u_hi) return;                                           // (no barrier below)
#pragma unroll
    for (int s_ = 0; s_ < DEPTH; ++s_) {
        const int ul = u + 2 * s_ < u_hi ? u + 2 * s_ : u;
        load_b(B[s_], ul);
        load_r(R[s_], ul);
    }
#ifndef AERO_EMU
    // the loop is ENTERED with nothing in flight: otherwise hipcc merges the preheader's pending first loads into the loop header's state
    // and waits there in every trip -- for the next unit's fragments it has just requested (the builtin, not inline asm: an asm wait
    // is invisible to its scoreboard)
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
#endif
    auto trip = [&](h16x8 (&Bc)[KS], h16x8 (&Rc)[GW][NV], h16x8 (&Bl)[KS], h16x8 (&Rl)[GW][NV], int u) {
        {
            const int un = u + 2 * DEPTH < u_hi ? u + 2 * DEPTH : u;   // (past the end: re-read this unit, no branch around the prefetch)
            load_b(Bl, un);
            load_r(Rl, un);
        }
        const int t = u * 16 + n;
        const bool tin = t < T;
#pragma unroll
        for (int g = 0; g < GW; ++g) {
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (WLDS) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const h16x8*)(Wl + ((g * 4 + j) * KS + ks) * 512), Bc[ks], acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[g][j][ks], Bc[ks], acc[j], 0, 0, 0);
                }
            // rows 16 q + 4 j + i of the group: coefficients are 4 + 4 consecutive float4 of the block's tables, applied tile by tile IN the
            // accumulator registers (a group's live set stays small: with everything fetched up front the three groups of a unit
            // spilled 63 registers and every spill reload drained the next unit's prefetch: 1.5 TB/s)
            const int ci = (wm * GW + g) * 64 + 16 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a4 = *(const f32x4*)&ca[ci + 4 * j];
                const f32x4 b4 = *(const f32x4*)&cb[ci + 4 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = acc[j][i] * a4[i] + b4[i];
            }
            const int o0 = obase + grp_o + g * (GLU ? 32 : 64) + lane_o;          // first stored channel of this lane
            const int oi = grp_o + g * (GLU ? 32 : 64) + lane_o;                  // ... inside the chunk (LDS tables)
            if constexpr (GLU) {
                h16x8 y;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {                                  // outputs 4 hh .. 4 hh + 3 = tiles 2 hh, 2 hh + 1
                    const f32x4 s4 = *(const f32x4*)&cs[oi + 4 * hh];
                    const f32x4 p4 = *(const f32x4*)&cp[oi + 4 * hh];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = 2 * hh + (e >> 1), h = e & 1;
                        const float o = acc[j][2 * h] * aero_sigmoid(acc[j][2 * h + 1]);
                        y[4 * hh + e] = (h16)(o * s4[e] + (float)Rc[g][0][4 * hh + e] + p4[e]);
                    }
                }
                if (tin && o0 < Mout) *(h16x8*)(dr + t * dt + o0) = y;
            } else {
#pragma unroll
                for (int vv = 0; vv < 2; ++vv) {
                    h16x8 y;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f32x4 s4 = *(const f32x4*)&cs[oi + 8 * vv + 4 * hh];
                        const f32x4 p4 = *(const f32x4*)&cp[oi + 8 * vv + 4 * hh];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float val = acc[2 * vv + hh][e];
                            if constexpr (ACT == AERO_ACT_RELU) val = fmaxf(val, 0.f);
                            else if constexpr (ACT == AERO_ACT_GELU) val = aero_gelu(val);
                            y[4 * hh + e] = (h16)(val * s4[e] + (float)Rc[g][vv][4 * hh + e] + p4[e]);
                        }
                    }
                    if (tin && o0 + 8 * vv < Mout) *(h16x8*)(dr + t * dt + o0 + 8 * vv) = y;
                }
            }
            aero_sched_fence();                                                   // the next group's MFMAs and reads start after this one's store
        }
    };
#pragma unroll 1
    for (;;) {
        if constexpr (DEPTH == 1) {
            trip(B[0], R[0], B[1], R[1], u);
            if ((u += 2) >= u_hi) break;
            trip(B[1], R[1], B[0], R[0], u);
            if ((u += 2) >= u_hi) break;
        } else {
            trip(B[0], R[0], B[2], R[2], u);
            if ((u += 2) >= u_hi) break;
            trip(B[1], R[1], B[0], R[0], u);
            if ((u += 2) >= u_hi) break;
            trip(B[2], R[2], B[1], R[1], u);
            if ((u += 2) >= u_hi) break;
        }
    }
}

// k-steps of the weight image for C input channels: ceil(C / 32) up to 6, then the 8- and 12-step instantiations (zero columns)
static int aero_pw_ks(int C) {
    const int ks = (C + 31) / 32;
    return ks <= 6 ? ks : (ks <= 8 ? 8 : 12);
}

// rows per chunk (= per block) for a contraction of C channels: the register budget of the resident weight fragments
static int aero_pw_gw(int C, int M) {
    const int ks = aero_pw_ks(C);
    const int groups = (M + 63) / 64;
    if (ks > 12 || ks < 1) return 0;
    if (ks > 3) return 1;                                         // weights in LDS (WLDS): one 64-row group per wave
    int gw = ks == 3 ? 2 : 3;                                     // 96 fragment registers per wave
    while (gw > 1 && 2 * (gw - 1) >= groups) --gw;                // no larger than the layer needs
    return gw;
}

static int aero_pw_ok(const aero_pw_desc* d) {
    if (!d || !d->x || !d->wimg || !d->dst) return 0;
    // Round 6 (kernel coverage, profiles/r06_kernel_coverage.txt): the instantiations nothing launches are gone -- 96 < C <= 384 (weights in
    // LDS: measured 15-80 % slower than the LDS-tiled conv at the model's widths, profiles/r04_pw_wlds_ab.txt; the engine never asked for
    // it), GELU (no pointwise conv of the path has it), GroupNorm with anything but GLU (the DConv tail is the one normalised pointwise
    // conv): 104 -> 32 kernels.  Such descriptors are refused here and the caller takes aero_conv_fwd.
    if (d->C < 8 || d->C % 8 || d->C > 96 || d->M < 16 || d->M % 16) return 0;
    if (d->act != AERO_ACT_NONE && d->act != AERO_ACT_RELU && d->act != AERO_ACT_GLU) return 0;
    if (d->stats && d->act != AERO_ACT_GLU) return 0;
    const int Mout = d->act == AERO_ACT_GLU ? d->M / 2 : d->M;
    if (Mout % 8) return 0;
    auto al = [](int64_t s) { return s % 8 == 0; };
    if (!al(d->x_b) || !al(d->x_f) || !al(d->x_t) || !al(d->d_b) || !al(d->d_f) || !al(d->d_t)) return 0;
    if (((uintptr_t)d->x & 15) || ((uintptr_t)d->dst & 15) || ((uintptr_t)d->wimg & 15)) return 0;
    if (d->res && (!al(d->r_b) || !al(d->r_f) || !al(d->r_t) || ((uintptr_t)d->res & 15))) return 0;
    if (d->x1 && (!al(d->x1_b) || !al(d->x1_f) || !al(d->x1_t) || ((uintptr_t)d->x1 & 15) || d->C0 < 8 || d->C0 % 8 || d->C0 >= d->C ||
                  (int64_t)d->T * d->x1_t >= (1ll << 31))) return 0;
    if ((int64_t)d->T * d->x_t >= (1ll << 31) || (int64_t)d->T * d->d_t >= (1ll << 31) || (d->res && (int64_t)d->T * d->r_t >= (1ll << 31))) return 0;
    if (d->stats && !(d->stat_count > 0)) return 0;
    return aero_pw_gw(d->C, d->M) > 0;
}

template <int KS, int GW>
static void aero_pw_go(const AeroPwK& p, dim3 grid, hipStream_t stream) {
    const bool norm = p.d.stats != nullptr;
    const dim3 block(256);
    constexpr bool WL = KS > 3;
    constexpr size_t lds = WL ? (size_t)2 * GW * 4 * KS * 1024 : 0;
    switch (p.d.act) {                                               // (aero_pw_ok: NONE / RELU / GLU, statistics with GLU only)
        case AERO_ACT_GLU:
            if (norm) AERO_LAUNCH_DYN((aero_pw_kernel<KS, GW, AERO_ACT_GLU, true, WL>), grid, block, lds, stream, p);
            else AERO_LAUNCH_DYN((aero_pw_kernel<KS, GW, AERO_ACT_GLU, false, WL>), grid, block, lds, stream, p);
            break;
        case AERO_ACT_RELU: AERO_LAUNCH_DYN((aero_pw_kernel<KS, GW, AERO_ACT_RELU, false, WL>), grid, block, lds, stream, p); break;
        default: AERO_LAUNCH_DYN((aero_pw_kernel<KS, GW, AERO_ACT_NONE, false, WL>), grid, block, lds, stream, p); break;
    }
}

static int aero_pw_launch(const aero_pw_desc* d, hipStream_t stream, const char** err) {
    if (!aero_pw_ok(d)) { *err = "pw: unsupported descriptor (C <= 96 in steps of 8, 16-byte aligned channels-last rows, M % 16 == 0, NONE / RELU / GLU, statistics with GLU only)"; return AERO_ERR_UNSUPPORTED; }
    if (d->B < 1 || d->F < 1 || d->T < 1) { *err = "pw: empty tensor"; return AERO_ERR_ARG; }
    AeroPwK p;
    p.d = *d;
    p.Mout = d->act == AERO_ACT_GLU ? d->M / 2 : d->M;
    const int ks = aero_pw_ks(d->C), gw = aero_pw_gw(d->C, d->M);
    const int nchunk = (d->M + 128 * gw - 1) / (128 * gw);
    const long rows = (long)d->B * d->F;
    const int nunit = (d->T + 15) / 16;
    // enough blocks for two per CU with something to hide latency behind; a split is a whole number of unit PAIRS (two waves alternate)
    int nsplit = (int)((512 + rows * nchunk - 1) / (rows * nchunk));
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 4) nsplit = 4;
    int upb = (nunit + nsplit - 1) / nsplit;
    upb = (upb + 1) & ~1;
    nsplit = (nunit + upb - 1) / upb;
    p.nsplit = nsplit;
    p.upb = upb;
    p.nchunk = nchunk;
    if (rows * nsplit * nchunk > 0x7fffffffL) { *err = "pw: too many rows for one launch"; return AERO_ERR_ARG; }
    const dim3 grid((unsigned)(rows * nsplit * nchunk));
    if (ks == 1) { if (gw == 1) aero_pw_go<1, 1>(p, grid, stream); else if (gw == 2) aero_pw_go<1, 2>(p, grid, stream); else aero_pw_go<1, 3>(p, grid, stream); }
    else if (ks == 2) { if (gw == 1) aero_pw_go<2, 1>(p, grid, stream); else if (gw == 2) aero_pw_go<2, 2>(p, grid, stream); else aero_pw_go<2, 3>(p, grid, stream); }
    else { if (gw == 1) aero_pw_go<3, 1>(p, grid, stream); else aero_pw_go<3, 2>(p, grid, stream); }      // (ks <= 3: aero_pw_ok holds C <= 96)
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The FTB's channel squeeze (modules.py:284-288 + 307-309: Conv2d(C, r = 5, 1x1) -> BatchNorm -> ReLU, reshaped to [B, r*F, T] for the Conv1d
// over time) as one streaming pass.  dst is the "image" the Conv1d reads: fp16 [B][T][F * rp], element (b, f, t, m) at (b*T + t) * F*rp +
// f*rp + m -- for one (b, f) row that is five halves every 640 bytes, which is what bound the general skinny kernel (137 us for 197 MB).
// Here a block owns 64 time steps x FG = 16 frequency rows: a wave walks the rows for its 16 steps (B fragments straight from global
// memory, the 16 x C weight tile resident in registers, next row's loads in flight), parks its r values per step in an LDS tile
// [64][FG * rp] and the block writes the tile as 16-byte runs of FG * rp halves per time step.
struct AeroSqueezeK {
    const h16* x; int64_t x_b, x_f, x_t;
    const h16* wimg; const float* bias;
    h16* dst;
    int B, F, T, C, M, rp, act;
};

template <int KS>
__global__ __launch_bounds__(256) void aero_squeeze_kernel(AeroSqueezeK p) {
    constexpr int FG = 16, TT = 64;
    __shared__ AERO_LDS_ALIGN h16 Ys[TT * FG * 8];                 // [TT][FG * rp], rp <= 8
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int t0 = blockIdx.x * TT, f0 = blockIdx.y * FG, b = blockIdx.z;
    const int rowlen = FG * p.rp;                                  // halves per time step in the tile
    if (p.rp > p.M) {                                              // pad channels of a slot are written as zeros (the caller's image has them zero)
        for (int i = tid; i < TT * rowlen; i += 256) Ys[i] = (h16)0;
        __syncthreads();
    }
    h16x8 A[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) A[ks] = *(const h16x8*)(p.wimg + ks * 512 + lane * 8);       // [ks][lane][8]: row = lane & 15, k-octet = lane >> 4
    float bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = (p.bias && 4 * q + i < p.M) ? p.bias[4 * q + i] : 0.f;
    int t = t0 + wave * 16 + n;
    const bool tin = t < p.T;
    t = tin ? t : p.T - 1;
    const h16* px = p.x + (int64_t)b * p.x_b + (int64_t)t * p.x_t;
    const int kq = 8 * q;
    const int nf = p.F - f0 < FG ? p.F - f0 : FG;
    auto fetch = [&](h16x8 (&Bf)[KS], int f) {
        const h16* pr = px + (int64_t)(f0 + f) * p.x_f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + kq;
            Bf[ks] = *(const h16x8*)(pr + (k0 < p.C ? k0 : 0));   // (padding lanes: valid data against zero weight columns)
        }
    };
    h16x8 Bc[KS], Bn[KS];
    fetch(Bc, 0);
#ifndef AERO_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
#pragma unroll 1
    for (int f = 0; f < nf; ++f) {
        fetch(Bn, f + 1 < nf ? f + 1 : f);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ks], Bc[ks], acc, 0, 0, 0);
        h16* yr = Ys + (wave * 16 + n) * rowlen + f * p.rp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * q + i;
            float v = acc[i] + bias[i];
            if (p.act == AERO_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == AERO_ACT_GELU) v = aero_gelu(v);
            if (m < p.M) yr[m] = (h16)v;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Bc[ks] = Bn[ks];
    }
    __syncthreads();
    // tile -> image: time step tt holds nf * rp valid halves at Ys[tt * rowlen ..], destined for dst[(b*T + t0 + tt) * F*rp + f0*rp ..]
    const int vl = nf * p.rp;                                       // valid halves per step
    const int64_t Frp = (int64_t)p.F * p.rp;
    if ((vl & 7) == 0 && ((f0 * p.rp) & 7) == 0 && (Frp & 7) == 0 && (rowlen & 7) == 0) {
        const int nv = vl >> 3;
        for (int idx = tid; idx < TT * nv; idx += 256) {
            const int tt = idx / nv, v = idx - tt * nv;
            if (t0 + tt < p.T) *(h16x8*)(p.dst + ((int64_t)b * p.T + t0 + tt) * Frp + f0 * p.rp + v * 8) = *(const h16x8*)&Ys[tt * rowlen + v * 8];
        }
    } else {
        for (int idx = tid; idx < TT * vl; idx += 256) {
            const int tt = idx / vl, e = idx - tt * vl;
            if (t0 + tt < p.T) p.dst[((int64_t)b * p.T + t0 + tt) * Frp + f0 * p.rp + e] = Ys[tt * rowlen + e];
        }
    }
}

static int aero_squeeze_launch(const void* x, int64_t x_b, int64_t x_f, int64_t x_t, const void* wimg, const float* bias, void* dst, int B, int F, int T,
                               int C, int M, int rp, int act, hipStream_t stream, const char** err) {
    if (!x || !wimg || !dst) { *err = "squeeze: null pointer"; return AERO_ERR_ARG; }
    if (B < 1 || F < 1 || T < 1 || M < 1 || M > 16 || rp < M || rp > 8 || C < 8 || C % 8 || C > 192) { *err = "squeeze: 1 <= M <= rp <= 8, C <= 192 in steps of 8"; return AERO_ERR_UNSUPPORTED; }
    if ((x_b % 8) || (x_f % 8) || (x_t % 8) || ((uintptr_t)x & 15) || ((uintptr_t)wimg & 15) || (int64_t)T * x_t >= (1ll << 31)) { *err = "squeeze: 16-byte aligned channels-last rows required"; return AERO_ERR_UNSUPPORTED; }
    if (act != AERO_ACT_NONE && act != AERO_ACT_RELU && act != AERO_ACT_GELU) { *err = "squeeze: activation"; return AERO_ERR_UNSUPPORTED; }
    AeroSqueezeK p;
    p.x = (const h16*)x; p.x_b = x_b; p.x_f = x_f; p.x_t = x_t; p.wimg = (const h16*)wimg; p.bias = bias; p.dst = (h16*)dst;
    p.B = B; p.F = F; p.T = T; p.C = C; p.M = M; p.rp = rp; p.act = act;
    const dim3 grid((unsigned)((T + 63) / 64), (unsigned)((F + 15) / 16), (unsigned)B), block(256);
    switch ((C + 31) / 32) {
        case 1: AERO_LAUNCH(aero_squeeze_kernel<1>, grid, block, stream, p); break;
        case 2: AERO_LAUNCH(aero_squeeze_kernel<2>, grid, block, stream, p); break;
        case 3: AERO_LAUNCH(aero_squeeze_kernel<3>, grid, block, stream, p); break;
        case 4: AERO_LAUNCH(aero_squeeze_kernel<4>, grid, block, stream, p); break;
        case 5: AERO_LAUNCH(aero_squeeze_kernel<5>, grid, block, stream, p); break;
        default: AERO_LAUNCH(aero_squeeze_kernel<6>, grid, block, stream, p); break;
    }
    return AERO_OK;
}
