// k_pw.h -- pointwise (1x1) convolution / Linear as a weight-stationary streaming GEMM.
//
// Most launches of the path are 1x1 contractions with a SHORT K (12..192 channels): DConv conv2, LSTM input
// projections and output Linear, LocalState q|k|v|decay and proj, FTB conv2, encoder rewrites, pre_conv.  They are
// HBM-bound (arithmetic intensity ~K FLOP/B); the tiled kernel of k_conv.h spends their few K-chunks in
// load->LDS->barrier latency.  Here the weight tile [BM][K] is staged into LDS ONCE per block; each wave then
// streams position tiles (32 positions x K) straight from HBM into MFMA B-fragments in registers (16-byte loads,
// next tile prefetched while the current one is multiplied) -- no LDS traffic for activations and no barrier in
// the loop.  Positions are the flattened (b, f, t) index: a 1x1 conv has no row structure.
// Algorithmic bytes per position: 2*(C0+C1) read + 2*Mout written (+2*Mout for a residual).
#pragma once
#include "aero_common.h"

struct AeroPwK {
    aero_conv_desc d;
    int Kp, Mpad, vec_out;
    int64_t P, ntiles;
    int64_t s0_p, s1_p, d_p, r_p;    // elements per position (pitch) of the flat tensors
};

template <int MF, int KT>
__global__ __launch_bounds__(256) void aero_pw_kernel(AeroPwK p) {
    constexpr int BM = MF * 16;
    __shared__ AERO_LDS_ALIGN h16 As[KT * BM * 32];
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, col = lane & 15;
    const int m0 = blockIdx.y * BM;
    const h16* Wp = (const h16*)d.weight + (int64_t)m0 * p.Kp;
    // ---- weights: [BM][Kp] -> LDS, one swizzled [BM][32] image per k-step
    for (int v = tid; v < KT * BM * 4; v += 256) {
        const int kt = v / (BM * 4), rem = v - kt * (BM * 4);
        const int r = rem >> 2, qq = rem & 3;
        h16x8 w = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (kt * 32 < p.Kp) w = *(const h16x8*)(Wp + (int64_t)r * p.Kp + kt * 32 + qq * 8);
        *(h16x8*)&As[kt * BM * 32 + aero_tile_off(r, qq)] = w;
    }
    __syncthreads();

    const h16* s0 = (const h16*)d.src0;
    const h16* s1 = (const h16*)d.src1;
    const int C0 = d.C0, C1 = d.C1;
    auto load_b = [&](int64_t tile, h16x8 (*bf)[2]) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int64_t pos = tile * 128 + wave * 32 + n * 16 + col;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int c = kt * 32 + q * 8;
                h16x8 z = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (pos < p.P) {
                    if (c < C0) {
                        if (s0) z = *(const h16x8*)(s0 + pos * p.s0_p + c);
                    } else if (c - C0 < C1) {
                        z = *(const h16x8*)(s1 + pos * p.s1_p + (c - C0));
                    }
                }
                bf[kt][n] = z;
            }
        }
    };

    const int M = d.M;
    const bool glu = d.act == AERO_ACT_GLU;
    const int Mout = glu ? (M >> 1) : M;
    const int nout = glu ? 2 : 4;
    h16* dst16 = (h16*)d.dst;
    float* dst32 = (float*)d.dst;
    const h16* res = (const h16*)d.res;
    float bv[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + i * 16 + q * 4 + r;
            bv[i][r] = (d.bias && m < M) ? d.bias[m] : 0.f;
        }

    h16x8 bcur[KT][2], bnxt[KT][2];
    int64_t tile = blockIdx.x;
    if (tile < p.ntiles) load_b(tile, bcur);
    for (; tile < p.ntiles; tile += gridDim.x) {
        const int64_t nxt = tile + gridDim.x;
        if (nxt < p.ntiles) load_b(nxt, bnxt);
        f32x4 acc[MF][2];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const h16x8 a = *(const h16x8*)&As[kt * BM * 32 + aero_tile_off(i * 16 + col, q)];
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bcur[kt][0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bcur[kt][1], acc[i][1], 0, 0, 0);
            }
        }
        // ---- epilogue (same order as k_conv.h): +bias, act, +res, +post_add, per-b affine, store
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int64_t pos = tile * 128 + wave * 32 + n * 16 + col;
            if (pos >= p.P) continue;
            int fo = 0;
            float bsc = 1.f, bsh = 0.f;
            if (d.post_add || d.batch_scale) {
                const int64_t rowi = pos / d.T;
                fo = (int)(rowi % d.Fout);
                if (d.batch_scale) {
                    const int b = (int)(rowi / d.Fout);
                    bsc = d.batch_scale[b];
                    bsh = d.batch_shift[b];
                }
            }
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int mbase = m0 + i * 16 + q * 4;
                if (mbase >= M) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[i][n][r] + bv[i][r];
                int cbase = mbase;
                if (glu) {
                    o[0] = o[0] * aero_sigmoid(o[1]);
                    o[1] = o[2] * aero_sigmoid(o[3]);
                    cbase = mbase >> 1;
                } else if (d.act == AERO_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
                } else if (d.act == AERO_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = aero_gelu(o[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r >= nout || cbase + r >= Mout) continue;
                    float x = o[r];
                    if (res) x += (float)res[pos * p.r_p + cbase + r];
                    if (d.post_add) x += d.post_add[(int64_t)fo * Mout + cbase + r];
                    o[r] = x * bsc + bsh;
                }
                const int64_t doff = pos * p.d_p + cbase;
                if (p.vec_out && cbase + nout <= Mout) {
                    if (d.dst_f32) {
                        if (glu) *(f32x2*)(dst32 + doff) = (f32x2){o[0], o[1]};
                        else *(f32x4*)(dst32 + doff) = (f32x4){o[0], o[1], o[2], o[3]};
                    } else {
                        if (glu) *(h16x2*)(dst16 + doff) = (h16x2){(h16)o[0], (h16)o[1]};
                        else *(h16x4*)(dst16 + doff) = (h16x4){(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r >= nout || cbase + r >= Mout) continue;
                        if (d.dst_f32) dst32[doff + r] = o[r];
                        else dst16[doff + r] = (h16)o[r];
                    }
                }
            }
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            bcur[kt][0] = bnxt[kt][0];
            bcur[kt][1] = bnxt[kt][1];
        }
    }
}

// flat = element (b,f,t,c) lives at ((b*F + f)*T + t)*pitch + c
static bool aero_pw_flat(int64_t sb, int64_t sf, int64_t st, int F, int T) {
    return sf == (int64_t)T * st && sb == (int64_t)F * sf;
}

// returns 0 if the descriptor is not eligible for the pointwise kernel, else MF*16 + KT
static int aero_pw_pick(const aero_conv_desc* d, int* mf_out, int* kt_out) {
    if (d->ntaps != 1 || d->transposed || d->fstride != 1 || d->df[0] != 0 || d->dt[0] != 0) return 0;
    if (d->Fin != d->Fout || d->dst_f_off != 0 || d->dst_F != d->Fout) return 0;
    const int Kp = (d->C0 + d->C1 + 31) / 32 * 32;
    if (Kp > 192) return 0;
    if ((d->C0 % 8) || (d->C1 % 8)) return 0;
    auto al8 = [](int64_t v) { return (v & 7) == 0; };
    if (d->src0 && (!aero_pw_flat(d->s0_b, d->s0_f, d->s0_t, d->Fin, d->T) || !al8(d->s0_t) || ((uintptr_t)d->src0 & 15))) return 0;
    if (d->src1 && (!aero_pw_flat(d->s1_b, d->s1_f, d->s1_t, d->Fin, d->T) || !al8(d->s1_t) || ((uintptr_t)d->src1 & 15))) return 0;
    if (!aero_pw_flat(d->d_b, d->d_f, d->d_t, d->Fout, d->T)) return 0;
    if (d->res && !aero_pw_flat(d->r_b, d->r_f, d->r_t, d->Fout, d->T)) return 0;
    const int kt = Kp <= 32 ? 1 : Kp <= 64 ? 2 : Kp <= 96 ? 3 : 6;
    const int M = d->M, Mpad = (M + 127) / 128 * 128;
    int bm;
    if (M <= 128) {
        bm = (M + 15) / 16 * 16;
        if (bm == 80) bm = 96;
        if (bm == 112) bm = 128;
    } else {
        const int cand[3] = {128, 96, 64};
        bm = 128;
        int best = 1 << 30;
        for (int i = 0; i < 3; ++i) {
            const int tot = (M + cand[i] - 1) / cand[i] * cand[i];
            if (tot <= Mpad && tot < best) { best = tot; bm = cand[i]; }
        }
    }
    if (((M + bm - 1) / bm) * bm > Mpad) return 0;
    *mf_out = bm / 16;
    *kt_out = kt;
    return bm + kt;
}

template <int MF>
static void aero_pw_launch_kt(int kt, dim3 grid, hipStream_t stream, const AeroPwK& p) {
    dim3 block(256);
    if (kt == 1) AERO_LAUNCH((aero_pw_kernel<MF, 1>), grid, block, stream, p);
    else if (kt == 2) AERO_LAUNCH((aero_pw_kernel<MF, 2>), grid, block, stream, p);
    else if (kt == 3) AERO_LAUNCH((aero_pw_kernel<MF, 3>), grid, block, stream, p);
    else AERO_LAUNCH((aero_pw_kernel<MF, 6>), grid, block, stream, p);
}

static int aero_pw_launch(const aero_conv_desc* d, int mf, int kt, hipStream_t stream, const char** err) {
    AeroPwK p;
    p.d = *d;
    p.Kp = (d->C0 + d->C1 + 31) / 32 * 32;
    p.Mpad = (d->M + 127) / 128 * 128;
    p.P = (int64_t)d->B * d->Fout * d->T;
    p.ntiles = (p.P + 127) / 128;
    p.s0_p = d->s0_t;
    p.s1_p = d->s1_t;
    p.d_p = d->d_t;
    p.r_p = d->r_t;
    const int Mout = d->act == AERO_ACT_GLU ? d->M / 2 : d->M;
    const int nout = d->act == AERO_ACT_GLU ? 2 : 4;
    const int esz = d->dst_f32 ? 4 : 2;
    p.vec_out = (Mout % nout == 0) && (d->d_t % nout == 0) && (((uintptr_t)d->dst % (uintptr_t)(esz * nout)) == 0);
    const int bm = mf * 16;
    const int nmt = (d->M + bm - 1) / bm;
    int64_t gx = p.ntiles;
    const int64_t cap = 256 * 6;                  // persistent-ish: a few blocks per CU, each streaming many tiles
    if (gx > cap) gx = cap;
    dim3 grid((unsigned)gx, (unsigned)nmt);
    switch (mf) {
        case 1: aero_pw_launch_kt<1>(kt, grid, stream, p); break;
        case 2: aero_pw_launch_kt<2>(kt, grid, stream, p); break;
        case 3: aero_pw_launch_kt<3>(kt, grid, stream, p); break;
        case 4: aero_pw_launch_kt<4>(kt, grid, stream, p); break;
        case 6: aero_pw_launch_kt<6>(kt, grid, stream, p); break;
        case 8: aero_pw_launch_kt<8>(kt, grid, stream, p); break;
        default: *err = "pw: unsupported tile"; return AERO_ERR_UNSUPPORTED;
    }
    return AERO_OK;
}
