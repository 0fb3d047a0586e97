// k_norm.h -- GroupNorm statistics + fused normalise/activation (HBM-bound elementwise work).
//
// Replaces nn.GroupNorm + F.gelu / F.glu / Snake / LayerScale / residual add of the reference
// (aero.py:127,133,198,206-214; modules.py:230-244; snake.py:67).  Statistics are reduced per
// (item, group) in fp32 per thread and fp64 across threads; the apply pass reads each fp16 value
// once and writes the activated result once.  Algorithmic bytes: stats 2 B/element read;
// apply 2 B read + 2 B write per output element (GLU reads 4 B per output) -- DESIGN.md section 4.
#pragma once
#include "aero_common.h"

// one block per (item, group); item = b (per_row == 0) or b*F + f (per_row == 1)
template <int VEC>
__global__ __launch_bounds__(256) void aero_norm_stats_kernel(aero_norm_desc d) {
    __shared__ double red[2][4];
    const int gs = d.C / d.G;
    const int item = blockIdx.x / d.G, g = blockIdx.x % d.G;
    const int b = d.per_row ? item / d.F : item;
    const int f0 = d.per_row ? item % d.F : 0;
    const int nf = d.per_row ? 1 : d.F;
    const h16* base = (const h16*)d.src + (int64_t)b * d.s_b + (int64_t)g * gs;
    const int vpp = gs / VEC;                       // vectors per position
    const int64_t total = (int64_t)nf * d.T * vpp;
    float s = 0.f, ss = 0.f;
    for (int64_t e = threadIdx.x; e < total; e += 256) {
        const int v = (int)(e % vpp);
        const int64_t pos = e / vpp;
        const int t = (int)(pos % d.T);
        const int f = f0 + (int)(pos / d.T);
        const h16* p = base + (int64_t)f * d.s_f + (int64_t)t * d.s_t + v * VEC;
        if (VEC == 8) {
            const h16x8 x = *(const h16x8*)p;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float y = (float)x[i]; s += y; ss += y * y; }
        } else if (VEC == 4) {
            const h16x4 x = *(const h16x4*)p;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float y = (float)x[i]; s += y; ss += y * y; }
        } else {
            const float y = (float)p[0];
            s += y;
            ss += y * y;
        }
    }
    double ds = aero_wave_sum((double)s), dss = aero_wave_sum((double)ss);
    if (aero_lane() == 0) { red[0][aero_wave()] = ds; red[1][aero_wave()] = dss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ds = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        dss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double n = (double)nf * d.T * gs;
        const double mean = ds / n;
        double var = dss / n - mean * mean;         // biased variance, as nn.GroupNorm
        if (var < 0) var = 0;
        d.stats[(int64_t)blockIdx.x * 2 + 0] = (float)mean;
        d.stats[(int64_t)blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)d.eps));
    }
}

// each thread produces VEC consecutive output channels of one position
template <int VEC>
__global__ __launch_bounds__(256) void aero_norm_apply_kernel(aero_norm_desc d, int64_t total) {
    const bool glu = d.act == AERO_ACT_GLU;
    const int Cout = glu ? d.C / 2 : d.C;
    const int vpp = Cout / VEC;
    const int gs = d.C / d.G;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c0 = (int)(e % vpp) * VEC;
        int64_t pos = e / vpp;
        const int t = (int)(pos % d.T);
        pos /= d.T;
        const int f = (int)(pos % d.F);
        const int b = (int)(pos / d.F);
        const int item = d.per_row ? b * d.F + f : b;
        const h16* src = (const h16*)d.src + (int64_t)b * d.s_b + (int64_t)f * d.s_f + (int64_t)t * d.s_t;
        float xa[VEC], xb[VEC];
        if (VEC == 8) {
            const h16x8 v = *(const h16x8*)(src + c0);
#pragma unroll
            for (int i = 0; i < VEC; ++i) xa[i] = (float)v[i];
            if (glu) {
                const h16x8 w = *(const h16x8*)(src + c0 + Cout);
#pragma unroll
                for (int i = 0; i < VEC; ++i) xb[i] = (float)w[i];
            }
        } else if (VEC == 4) {
            const h16x4 v = *(const h16x4*)(src + c0);
#pragma unroll
            for (int i = 0; i < VEC; ++i) xa[i] = (float)v[i];
            if (glu) {
                const h16x4 w = *(const h16x4*)(src + c0 + Cout);
#pragma unroll
                for (int i = 0; i < VEC; ++i) xb[i] = (float)w[i];
            }
        } else {
            xa[0] = (float)src[c0];
            if (glu) xb[0] = (float)src[c0 + Cout];
        }
        float snake_a = 0.f, snake_ia = 0.f;
        if (d.act == AERO_ACT_SNAKE) { snake_a = d.snake_a[f]; snake_ia = 1.0f / snake_a; }
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int c = c0 + i;
            float y = xa[i];
            if (d.stats) {
                const float* st = d.stats + ((int64_t)item * d.G + c / gs) * 2;
                y = (y - st[0]) * st[1];
            }
            if (d.gamma) y = y * d.gamma[c] + d.beta[c];
            if (glu) {
                const int c2 = c + Cout;
                float z = xb[i];
                if (d.stats) {
                    const float* st = d.stats + ((int64_t)item * d.G + c2 / gs) * 2;
                    z = (z - st[0]) * st[1];
                }
                if (d.gamma) z = z * d.gamma[c2] + d.beta[c2];
                y = y * aero_sigmoid(z);
                if (d.layer_scale) y *= d.layer_scale[c];
            } else if (d.act == AERO_ACT_GELU) {
                y = aero_gelu(y);
            } else if (d.act == AERO_ACT_RELU) {
                y = fmaxf(y, 0.f);
            } else if (d.act == AERO_ACT_SNAKE) {
                const float sn = sinf(y * snake_a);
                y = y + snake_ia * sn * sn;
            }
            o[i] = y;
        }
        if (d.res) {
            const h16* r = (const h16*)d.res + (int64_t)b * d.r_b + (int64_t)f * d.r_f + (int64_t)t * d.r_t + c0;
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] += (float)r[i];
        }
        h16* dst = (h16*)d.dst + (int64_t)b * d.d_b + (int64_t)f * d.d_f + (int64_t)t * d.d_t + c0;
        if (VEC == 8) {
            h16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (h16)o[i];
            *(h16x8*)dst = v;
        } else if (VEC == 4) {
            h16x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (h16)o[i];
            *(h16x4*)dst = v;
        } else {
            dst[0] = (h16)o[0];
        }
    }
}

static int aero_norm_pick_vec(int n, const void* p0, const void* p1, const void* p2, int64_t s0, int64_t s1, int64_t s2,
                              int64_t s3, int64_t s4, int64_t s5, int64_t s6, int64_t s7, int64_t s8) {
    const int64_t strides[9] = {s0, s1, s2, s3, s4, s5, s6, s7, s8};
    const void* ptrs[3] = {p0, p1, p2};
    for (int vec = 8; vec >= 4; vec >>= 1) {
        bool ok = (n % vec) == 0;
        for (int i = 0; i < 9 && ok; ++i) ok = (strides[i] % vec) == 0;
        for (int i = 0; i < 3 && ok; ++i) ok = ptrs[i] == nullptr || (((uintptr_t)ptrs[i]) % (uintptr_t)(2 * vec)) == 0;
        if (ok) return vec;
    }
    return 1;
}

static int aero_norm_check(const aero_norm_desc* d, const char** err) {
    if (!d || !d->src) { *err = "norm: null src"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->F < 1 || d->T < 1 || d->C < 1 || d->G < 1 || d->C % d->G) { *err = "norm: bad geometry"; return AERO_ERR_ARG; }
    return AERO_OK;
}

static int aero_norm_stats_launch(const aero_norm_desc* d, hipStream_t stream, const char** err) {
    int rc = aero_norm_check(d, err);
    if (rc) return rc;
    if (!d->stats) { *err = "norm_stats: null stats"; return AERO_ERR_ARG; }
    const int gs = d->C / d->G;
    const int vec = aero_norm_pick_vec(gs, d->src, nullptr, nullptr, d->s_b, d->s_f, d->s_t, 0, 0, 0, 0, 0, 0);
    const int64_t items = d->per_row ? (int64_t)d->B * d->F : d->B;
    dim3 grid((unsigned)(items * d->G)), block(256);
    if (vec == 8) AERO_LAUNCH((aero_norm_stats_kernel<8>), grid, block, stream, *d);
    else if (vec == 4) AERO_LAUNCH((aero_norm_stats_kernel<4>), grid, block, stream, *d);
    else AERO_LAUNCH((aero_norm_stats_kernel<1>), grid, block, stream, *d);
    return AERO_OK;
}

static int aero_norm_apply_launch(const aero_norm_desc* d, hipStream_t stream, const char** err) {
    int rc = aero_norm_check(d, err);
    if (rc) return rc;
    if (!d->dst) { *err = "norm_apply: null dst"; return AERO_ERR_ARG; }
    if (d->act == AERO_ACT_GLU && (d->C & 1)) { *err = "norm_apply: GLU needs even C"; return AERO_ERR_ARG; }
    if (d->act == AERO_ACT_SNAKE && !d->snake_a) { *err = "norm_apply: snake needs a"; return AERO_ERR_ARG; }
    if ((d->gamma == nullptr) != (d->beta == nullptr)) { *err = "norm_apply: gamma/beta"; return AERO_ERR_ARG; }
    const int Cout = d->act == AERO_ACT_GLU ? d->C / 2 : d->C;
    const int vec = aero_norm_pick_vec(Cout, d->src, d->dst, d->res, d->s_b, d->s_f, d->s_t, d->d_b, d->d_f, d->d_t,
                                       d->res ? d->r_b : 0, d->res ? d->r_f : 0, d->res ? d->r_t : 0);
    const int64_t total = (int64_t)d->B * d->F * d->T * (Cout / vec);
    int64_t nb = (total + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    dim3 grid((unsigned)nb), block(256);
    if (vec == 8) AERO_LAUNCH((aero_norm_apply_kernel<8>), grid, block, stream, *d, total);
    else if (vec == 4) AERO_LAUNCH((aero_norm_apply_kernel<4>), grid, block, stream, *d, total);
    else AERO_LAUNCH((aero_norm_apply_kernel<1>), grid, block, stream, *d, total);
    return AERO_OK;
}
