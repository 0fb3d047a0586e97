// k_dconv.h -- fused tail of a DConv residual layer (reference modules.py:209-210,243-244):
//     g   = conv1x1(h)                       [hidden -> 2C]                 (modules.py:209, nn.Conv1d k=1)
//     g   = GroupNorm(1, 2C)(g)              statistics over (2C, T) per (b, f) row
//     out = x + layer_scale * GLU(g)         (nn.GLU, LayerScale, skip;  modules.py:210,141,244)
// One block owns one (b, f) row.  The 2C-channel tensor g -- 8x larger than h and the largest tensor of the layer --
// never reaches HBM: pass 0 computes g tile by tile on the MFMAs and only accumulates sum / sum of squares, pass 1
// RECOMPUTES g (K is tiny: 12..96 channels, the row of h stays in L2) and applies norm + GLU + scale + residual in
// the epilogue.  Unfused, the same work wrote g (2 B/elt), re-read it for the statistics and again for the
// activation: ~12 B per output element; fused it is 2 B (residual) + 2 B (store) + the small h row.
// Operands stream global -> LDS with global_load_lds (two stages, one barrier per 32-channel chunk), the chunk
// stream runs straight across tile and pass boundaries.
#pragma once
#include "aero_common.h"

struct AeroDconvTailK {
    aero_dconv_tail_desc d;
    int Kp, KT, Mpad, nmt, ntt, M;
};

template <int MF>   // BM = 32*MF output channels (GLU-interleaved rows) per tile: MF = 3 -> 96, MF = 4 -> 128
__global__ __launch_bounds__(256) void aero_dconv_tail_kernel(AeroDconvTailK p) {
    constexpr int WM = 2, WN = 2, NF = 4, BM = 32 * MF, BN = 128;
    constexpr int STAGE = (BM + BN) * 32;
    __shared__ AERO_LDS_ALIGN h16 smem[2 * STAGE];
    __shared__ double red[2][4];
    __shared__ float stat[2];
    const aero_dconv_tail_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int row = blockIdx.x;
    const int T = d.T, M = p.M, C = d.C, hp = d.h_pitch;
    const h16* hrow = (const h16*)d.h + (int64_t)row * T * hp;
    const h16* W = (const h16*)d.weight;
    const h16* zp = aero_zero_page;
    constexpr int NIA = (BM * 4 / 64 + 3) / 4;
    // lane-invariant parts of the copy addresses
    int a_row[NIA], a_q8[NIA], b_pos[2], b_q8[2];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int s = (wave + 4 * i) * 64 + lane;
        a_row[i] = s >> 2;
        a_q8[i] = ((s & 3) ^ ((0 - ((s >> 2) >> 2)) & 3)) * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = (wave + 4 * i) * 64 + lane;
        b_pos[i] = s >> 2;
        b_q8[i] = ((s & 3) ^ ((0 - ((s >> 2) >> 2)) & 3)) * 8;
    }
    const int chunks_per_pass = p.nmt * p.ntt * p.KT;
    const int total = 2 * chunks_per_pass;
    auto issue = [&](int q, int buf) {
        const int qq = q % chunks_per_pass;
        const int kt = qq % p.KT, tile = qq / p.KT;
        const int nt = tile % p.ntt, mt = tile / p.ntt;
        h16* As = smem + buf * STAGE;
        h16* Bs = As + BM * 32;
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            if (wave + 4 * i < BM * 4 / 64)
                aero_glds16(W + (int64_t)(mt * BM + a_row[i]) * p.Kp + kt * 32 + a_q8[i], As + (wave + 4 * i) * 512);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = nt * BN + b_pos[i];
            const int c = kt * 32 + b_q8[i];
            const bool ok = t < T && c < hp;
            aero_glds16(ok ? hrow + (int64_t)t * hp + c : zp, Bs + (wave + 4 * i) * 512);
        }
    };

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    float mean = 0.f, rstd = 1.f;
    h16* dst = (h16*)d.dst + (int64_t)row * T * C;
    const h16* res = (const h16*)d.res + (int64_t)row * T * C;

    issue(0, 0);
    for (int q = 0; q < total; ++q) {
        const int buf = q & 1;
        __syncthreads();                                   // chunk q landed; the other stage is free
        if (q + 1 < total) issue(q + 1, buf ^ 1);
        const h16* As = smem + buf * STAGE;
        const h16* Bs = As + BM * 32;
        h16x8 af[MF], bf[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = *(const h16x8*)&As[aero_tile_off((wm * MF + i) * 16 + (lane & 15), lane >> 4)];
#pragma unroll
        for (int n = 0; n < NF; ++n) bf[n] = *(const h16x8*)&Bs[aero_tile_off((wn * NF + n) * 16 + (lane & 15), lane >> 4)];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[n], acc[i][n], 0, 0, 0);
        const int qq = q % chunks_per_pass;
        if (qq % p.KT != p.KT - 1) continue;
        // ---- a (m-tile, n-tile) of g is complete in the accumulators
        const int pass = q / chunks_per_pass;
        const int tile = qq / p.KT;
        const int nt = tile % p.ntt, mt = tile / p.ntt;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int mbase = mt * BM + (wm * MF + i) * 16 + (lane >> 4) * 4;    // rows (a_u, b_u, a_u+1, b_u+1)
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (mbase < M) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = d.bias[mbase + r];
            }
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int t = nt * BN + (wn * NF + n) * 16 + (lane & 15);
                const bool live = t < T && mbase < M;
                float g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = acc[i][n][r] + bv[r];
                acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!live) continue;
                if (pass == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s1 += g[r]; s2 += g[r] * g[r]; }
                } else {
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = (g[r] - mean) * rstd;
                        if (d.gamma) v = v * d.gamma[mbase + r] + d.beta[mbase + r];
                        y[r] = v;
                    }
                    const int u = mbase >> 1;
                    const int64_t off = (int64_t)t * C + u;
                    const h16x2 rr = *(const h16x2*)(res + off);
                    const float o0 = (float)rr[0] + d.layer_scale[u] * y[0] * aero_sigmoid(y[1]);
                    const float o1 = (float)rr[1] + d.layer_scale[u + 1] * y[2] * aero_sigmoid(y[3]);
                    *(h16x2*)(dst + off) = (h16x2){(h16)o0, (h16)o1};
                }
            }
        }
        if (pass == 0 && q == chunks_per_pass - 1) {
            // ---- row statistics over (2C, T): fp32 per thread, fp64 across the block
            double ds = aero_wave_sum((double)s1), dss = aero_wave_sum((double)s2);
            if (lane == 0) { red[0][wave] = ds; red[1][wave] = dss; }
            __syncthreads();
            if (tid == 0) {
                const double n = (double)M * T;
                const double mu = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n;
                double var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n - mu * mu;
                if (var < 0) var = 0;
                stat[0] = (float)mu;
                stat[1] = (float)(1.0 / sqrt(var + (double)d.eps));
            }
            __syncthreads();
            if (d.gamma) { mean = stat[0]; rstd = stat[1]; }
        }
    }
}

static int aero_dconv_tail_launch(const aero_dconv_tail_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->h || !d->weight || !d->bias || !d->layer_scale || !d->res || !d->dst) { *err = "dconv_tail: null pointer"; return AERO_ERR_ARG; }
    if ((d->gamma == nullptr) != (d->beta == nullptr)) { *err = "dconv_tail: gamma/beta"; return AERO_ERR_ARG; }
    if (d->R < 1 || d->T < 1 || d->C < 2 || (d->C & 1) || d->h_pitch < 8 || d->h_pitch % 8) { *err = "dconv_tail: C even, h_pitch multiple of 8"; return AERO_ERR_ARG; }
    if (d->h_pitch > 128) { *err = "dconv_tail: hidden > 128 unsupported"; return AERO_ERR_UNSUPPORTED; }
    if (((uintptr_t)d->h & 15) || ((uintptr_t)d->res & 3) || ((uintptr_t)d->dst & 3)) { *err = "dconv_tail: alignment"; return AERO_ERR_ARG; }
    AeroDconvTailK p;
    p.d = *d;
    p.M = 2 * d->C;
    p.Kp = (d->h_pitch + 31) / 32 * 32;
    p.KT = p.Kp / 32;
    p.Mpad = (p.M + 127) / 128 * 128;
    const int bm = (p.M % 128 == 0 || p.M % 96 != 0) ? 128 : 96;
    p.nmt = (p.M + bm - 1) / bm;
    if (p.nmt * bm > p.Mpad) { *err = "dconv_tail: tile/padding mismatch"; return AERO_ERR_ARG; }
    p.ntt = (d->T + 127) / 128;
    dim3 grid((unsigned)d->R), block(256);
    if (bm == 96) AERO_LAUNCH((aero_dconv_tail_kernel<3>), grid, block, stream, p);
    else AERO_LAUNCH((aero_dconv_tail_kernel<4>), grid, block, stream, p);
    return AERO_OK;
}
