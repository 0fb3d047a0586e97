// k_dconv.h -- a whole DConv residual branch (modules.py:221-249, the layers without BLSTM / LocalState) in ONE kernel:
//     for each layer:  x <- x + LayerScale( GLU( GN2( conv2( act( GN1( conv1_dilated(x) ) ) ) ) ) )
// conv1 is a 3-tap dilated Conv1d along time (C -> hidden = C/4), conv2 a pointwise Conv1d (hidden -> 2C), both GroupNorms
// have ONE group and normalise over (channels, time) of one (b, f) item -- so everything a (b, f) row needs is the row
// itself.  A block owns one row: the [T][C] fp16 slab (48 KB at C = 48, T = 501) is read from HBM ONCE into LDS, all
// `depth` layers run on it in place, and it is written back once.  The layer-by-layer path (conv1, norm statistics, norm
// apply, Gram statistics, conv2 tail; x read three times and written once per layer, five launches) took 655 us for the
// first encoder level and 350 us for the second at the bench shape; this kernel moves 2 x 197 MB and 2 x 98 MB.
//
// Per layer, three passes over the row's 16-step column fragments (waves take fragments round-robin):
//   A  conv1 as MFMA 16x16x32 (A = W1 from LDS, B = 8 consecutive channels of x at t + (tap-1)*dilation straight from the
//      row image), + bias -> raw h, fp16, into an LDS side buffer [T][hidden]; sum / sum of squares -> GN1 statistics
//   B  h -> act(GN1(h)) in place; conv2 as MFMA 16x16x16 -- its B operand (lane = time step, 4 consecutive hidden units)
//      is exactly how the side buffer is read, 8 bytes per lane -- only for the GN2 statistics (the 2C-channel tensor is
//      never stored: recomputing a K = 16 contraction is cheaper than keeping it)
//   C  conv2 again, GN2, GLU (rows of W2 are interleaved (a0, b0, a1, b1, ...) so a lane holds both halves), LayerScale,
//      + x[t][c] read and written in place by the same lane (conv1's neighbours were consumed in pass A)
// Block-wide reductions: wave shuffles + 8 partials in LDS.  HBM-bound: algorithmic bytes = 4*C per (row, time step).
#pragma once
#include "aero_common.h"

struct AeroDconvK {
    aero_dconv_desc d;
    int HP;       // hidden rounded up to 16 (rows of the W1 image, k-extent of the W2 image)
    int hs;       // pitch of the h side buffer in halves (= hidden, a multiple of 4)
    int K1p;      // 3*C rounded up to 32
    int logp;     // 16 pad bytes after every (1 << logp) rows of the row image: b128 reads of 16 consecutive rows conflict-free
    int T16;      // T rounded up to 16
    int PADR;     // zero rows before t = 0 and after T16 (largest dilation)
};

static inline int aero_dconv_logp(int C) {
    int s = (2 * C) & 255, g = 256;                              // gcd(2C mod 256, 256): both powers of two times odd -> lowest set bit
    if (s) g = s & -s;
    int P = 256 / g, l = 0;
    while ((1 << l) < P) ++l;
    return l;
}

// LDS bytes of the kernel for a geometry (0: not representable)
static inline size_t aero_dconv_lds_bytes(int T, int C, int hidden, int maxdil) {
    if (T < 1 || C < 8 || C % 8 || hidden < 4 || hidden % 4 || hidden > 32 || maxdil < 1 || maxdil > 64) return 0;
    const int HP = (hidden + 15) / 16 * 16, K1p = (3 * C + 31) / 32 * 32, T16 = (T + 15) / 16 * 16;
    const int xrows = T16 + 2 * maxdil, logp = aero_dconv_logp(C);
    const size_t xs = (size_t)xrows * C + (size_t)((xrows >> logp) + 1) * 8;
    const size_t hb = (size_t)T16 * hidden;
    const size_t w1 = (size_t)HP * K1p, w2 = (size_t)2 * C * HP;
    const size_t halves = (xs + hb + w1 + w2 + 7) / 8 * 8;
    const size_t floats = 3 * HP + 3 * 2 * C + C + 2 * 8 * 2 + 8;
    return halves * 2 + floats * 4;
}

template <int HM>                                                // HM = HP / 16: M fragments of conv1 = k-steps of conv2
__global__ __launch_bounds__(512) void aero_dconv_row_kernel(AeroDconvK p) {
    const aero_dconv_desc& d = p.d;
    const int C = d.C, T = d.T, HP = p.HP, hs = p.hs, K1p = p.K1p, hidden = d.hidden;
    const int xrows = p.T16 + 2 * p.PADR;
    h16* xs = (h16*)AERO_DYN_SMEM;                                                 // row image, padded (see xoff)
    h16* hb = xs + (size_t)xrows * C + (size_t)((xrows >> p.logp) + 1) * 8;        // [T16][hs]
    h16* w1s = hb + (size_t)p.T16 * hs;                                            // [K1p/32][HP][32] tile-swizzled
    h16* w2s = w1s + (size_t)HP * K1p;                                             // [2C][HP]
    float* c1 = (float*)(xs + ((size_t)(w2s + (size_t)2 * C * HP - xs) + 7) / 8 * 8);   // [3][HP]   b1 | g1 | be1
    float* c2 = c1 + 3 * HP;                                                       // [3][2C]   b2 | g2 | be2 (GLU-interleaved)
    float* sc = c2 + 3 * 2 * C;                                                    // [C]       LayerScale
    float* red = sc + C;                                                           // [2][8][2] reduction partials
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int row = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int nfrag = p.T16 >> 4, cu = C >> 3, nk1 = K1p >> 5, nf2 = (2 * C) >> 4;
    auto xoff = [&](int r) { return r * C + ((r >> p.logp) << 3); };

    // the row (and its zero margins) -> LDS, once
    {
        const h16* src = (const h16*)d.x + (int64_t)row * T * C;
        for (int u = tid; u < xrows * cu; u += 512) {
            const int r = u / cu, s = u - r * cu;
            const int t = r - p.PADR;
            h16x8 v = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (t >= 0 && t < T) v = *(const h16x8*)(src + (int64_t)t * C + s * 8);
            *(h16x8*)&xs[xoff(r) + s * 8] = v;
        }
    }
    float snake_a = 0.f, snake_ia = 0.f;
    for (int l = 0; l < d.depth; ++l) {
        const aero_dconv_layer& L = d.layer[l];
        const int dil = L.dilation;
        const bool norm1 = L.g1 != nullptr, norm2 = L.g2 != nullptr;
        __syncthreads();                                         // previous layer done with the weights / the row is in
        for (int u = tid; u < HP * (K1p >> 3); u += 512) {       // W1 image [HP][K1p] -> k-step tiles
            const int r = u / (K1p >> 3), q8 = u - r * (K1p >> 3);
            const int kk = q8 >> 2, q = q8 & 3;
            *(h16x8*)&w1s[(size_t)kk * HP * 32 + aero_tile_off(r, q)] = *(const h16x8*)((const h16*)L.w1 + (size_t)r * K1p + q8 * 8);
        }
        for (int u = tid; u < 2 * C * HP / 8; u += 512) *(h16x8*)&w2s[u * 8] = *(const h16x8*)((const h16*)L.w2 + u * 8);
        for (int u = tid; u < HP; u += 512) {
            const bool in = u < hidden;
            c1[u] = in ? L.b1[u] : 0.f;
            c1[HP + u] = (in && norm1) ? L.g1[u] : 1.f;
            c1[2 * HP + u] = (in && norm1) ? L.be1[u] : 0.f;
        }
        for (int u = tid; u < 2 * C; u += 512) {
            c2[u] = L.b2[u];
            c2[2 * C + u] = norm2 ? L.g2[u] : 1.f;
            c2[4 * C + u] = norm2 ? L.be2[u] : 0.f;
        }
        for (int u = tid; u < C; u += 512) sc[u] = L.scale ? L.scale[u] : 1.f;
        if (d.act == AERO_ACT_SNAKE) { snake_a = L.snake_a[row % d.F]; snake_ia = 1.0f / snake_a; }
        __syncthreads();

        // ---- pass A: conv1 (+ bias) -> raw h, GN1 statistics
        float s1 = 0.f, s2 = 0.f;
        for (int cf = wave; cf < nfrag; cf += 8) {
            const int t = cf * 16 + col;
            f32x4 acc[HM];
#pragma unroll
            for (int mf = 0; mf < HM; ++mf) acc[mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < nk1; ++kk) {
                const int k = kk * 32 + g * 8;
                const int tap = (k >= C) + (k >= 2 * C);
                const int c = k - tap * C;
                h16x8 bfrag = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (k < 3 * C) bfrag = *(const h16x8*)&xs[xoff(t + p.PADR + (tap - 1) * dil) + c];
#pragma unroll
                for (int mf = 0; mf < HM; ++mf) {
                    const h16x8 a = *(const h16x8*)&w1s[(size_t)kk * HP * 32 + aero_tile_off(mf * 16 + col, g)];
                    acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfrag, acc[mf], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mf = 0; mf < HM; ++mf) {
                const int j = mf * 16 + g * 4;
                if (j < hidden) {                                // (hidden % 4 == 0: the four rows are all real or all padding)
                    const f32x4 hv = acc[mf] + *(const f32x4*)&c1[j];
                    if (t < T) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1 += hv[e]; s2 += hv[e] * hv[e]; }
                    }
                    *(h16x4*)&hb[(size_t)t * hs + j] = (h16x4){(h16)hv[0], (h16)hv[1], (h16)hv[2], (h16)hv[3]};
                }
            }
        }
        float mean1 = 0.f, rstd1 = 1.f;
        if (norm1) {
            s1 = aero_wave_sum(s1);
            s2 = aero_wave_sum(s2);
            if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
            __syncthreads();
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) { a += (double)red[w * 2]; b += (double)red[w * 2 + 1]; }
            const double n = (double)hidden * T, m = a / n;
            double var = b / n - m * m;
            var = var > 0.0 ? var : 0.0;
            mean1 = (float)m;
            rstd1 = (float)(1.0 / sqrt(var + (double)d.eps));
        } else {
            __syncthreads();                                     // every wave is done READING its neighbours' x rows before pass C writes x
        }

        // ---- pass B: h <- act(GN1(h)) in place; conv2 for the GN2 statistics
        s1 = s2 = 0.f;
        for (int cf = wave; cf < nfrag; cf += 8) {
            const int t = cf * 16 + col;
            h16x4 hB[HM];
#pragma unroll
            for (int ks = 0; ks < HM; ++ks) {
                const int j = ks * 16 + g * 4;
                hB[ks] = (h16x4){0, 0, 0, 0};
                if (j < hidden) {
                    const h16x4 raw = *(const h16x4*)&hb[(size_t)t * hs + j];
                    const f32x4 gm = *(const f32x4*)&c1[HP + j], bt = *(const f32x4*)&c1[2 * HP + j];
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = ((float)raw[e] - mean1) * rstd1 * gm[e] + bt[e];
                        if (d.act == AERO_ACT_GELU) y = aero_gelu(y);
                        else if (d.act == AERO_ACT_RELU) y = fmaxf(y, 0.f);
                        else if (d.act == AERO_ACT_SNAKE) { const float sn = aero_fast_sin(y * snake_a); y = y + snake_ia * sn * sn; }
                        o[e] = (h16)y;
                    }
                    *(h16x4*)&hb[(size_t)t * hs + j] = o;
                    hB[ks] = o;
                }
            }
            if (norm2) {
                for (int mf = 0; mf < nf2; ++mf) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < HM; ++ks) {
                        const h16x4 a = *(const h16x4*)&w2s[(size_t)(mf * 16 + col) * HP + ks * 16 + g * 4];
                        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, hB[ks], acc, 0, 0, 0);
                    }
                    const f32x4 v = acc + *(const f32x4*)&c2[mf * 16 + g * 4];
                    if (t < T) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
                    }
                }
            }
        }
        float mean2 = 0.f, rstd2 = 1.f;
        if (norm2) {
            s1 = aero_wave_sum(s1);
            s2 = aero_wave_sum(s2);
            if (lane == 0) { red[16 + wave * 2] = s1; red[16 + wave * 2 + 1] = s2; }
            __syncthreads();
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) { a += (double)red[16 + w * 2]; b += (double)red[16 + w * 2 + 1]; }
            const double n = (double)(2 * C) * T, m = a / n;
            double var = b / n - m * m;
            var = var > 0.0 ? var : 0.0;
            mean2 = (float)m;
            rstd2 = (float)(1.0 / sqrt(var + (double)d.eps));
        }

        // ---- pass C: conv2 again, GN2, GLU, LayerScale, + x in place
        for (int cf = wave; cf < nfrag; cf += 8) {
            const int t = cf * 16 + col;
            h16x4 hB[HM];
#pragma unroll
            for (int ks = 0; ks < HM; ++ks) {
                const int j = ks * 16 + g * 4;
                hB[ks] = j < hidden ? *(const h16x4*)&hb[(size_t)t * hs + j] : (h16x4){0, 0, 0, 0};
            }
            h16* xr = &xs[xoff(t + p.PADR)];
            for (int mf = 0; mf < nf2; ++mf) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < HM; ++ks) {
                    const h16x4 a = *(const h16x4*)&w2s[(size_t)(mf * 16 + col) * HP + ks * 16 + g * 4];
                    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, hB[ks], acc, 0, 0, 0);
                }
                const int m = mf * 16 + g * 4;
                f32x4 v = acc + *(const f32x4*)&c2[m];
                const f32x4 gm = *(const f32x4*)&c2[2 * C + m], bt = *(const f32x4*)&c2[4 * C + m];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean2) * rstd2 * gm[e] + bt[e];
                const int c = mf * 8 + g * 2;
                const f32x2 ls = *(const f32x2*)&sc[c];
                const h16x2 xv = *(const h16x2*)&xr[c];
                const float o0 = (float)xv[0] + v[0] * aero_sigmoid(v[1]) * ls[0];
                const float o1 = (float)xv[1] + v[2] * aero_sigmoid(v[3]) * ls[1];
                if (t < T) *(h16x2*)&xr[c] = (h16x2){(h16)o0, (h16)o1};
            }
        }
    }
    __syncthreads();
    {
        h16* dst = (h16*)d.y + (int64_t)row * T * C;
        for (int u = tid; u < T * cu; u += 512) {
            const int t = u / cu, s = u - t * cu;
            *(h16x8*)(dst + (int64_t)t * C + s * 8) = *(const h16x8*)&xs[xoff(t + p.PADR) + s * 8];
        }
    }
}

static int aero_dconv_row_fits_impl(int T, int C, int hidden, int maxdil) {
    const size_t b = aero_dconv_lds_bytes(T, C, hidden, maxdil);
    return b != 0 && b <= 160 * 1024;
}

static int aero_dconv_launch(const aero_dconv_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->x || !d->y) { *err = "dconv: null pointer"; return AERO_ERR_ARG; }
    if (d->R < 1 || d->T < 1 || d->depth < 1 || d->depth > AERO_DCONV_MAX_DEPTH || d->F < 1) { *err = "dconv: bad geometry"; return AERO_ERR_ARG; }
    if (d->act != AERO_ACT_RELU && d->act != AERO_ACT_GELU && d->act != AERO_ACT_SNAKE && d->act != AERO_ACT_NONE) { *err = "dconv: unsupported act"; return AERO_ERR_UNSUPPORTED; }
    int maxdil = 1;
    for (int l = 0; l < d->depth; ++l) {
        const aero_dconv_layer& L = d->layer[l];
        if (!L.w1 || !L.b1 || !L.w2 || !L.b2) { *err = "dconv: null layer weights"; return AERO_ERR_ARG; }
        if ((L.g1 == nullptr) != (L.be1 == nullptr) || (L.g2 == nullptr) != (L.be2 == nullptr)) { *err = "dconv: gamma/beta"; return AERO_ERR_ARG; }
        if (d->act == AERO_ACT_SNAKE && !L.snake_a) { *err = "dconv: snake needs a"; return AERO_ERR_ARG; }
        if (L.dilation < 1) { *err = "dconv: dilation"; return AERO_ERR_ARG; }
        if ((((uintptr_t)L.w1 | (uintptr_t)L.w2) & 15)) { *err = "dconv: unaligned weights"; return AERO_ERR_ARG; }
        maxdil = L.dilation > maxdil ? L.dilation : maxdil;
    }
    if ((((uintptr_t)d->x | (uintptr_t)d->y) & 15)) { *err = "dconv: unaligned rows"; return AERO_ERR_ARG; }
    if (!aero_dconv_row_fits_impl(d->T, d->C, d->hidden, maxdil)) { *err = "dconv: row does not fit the LDS (or C % 8, hidden % 4, hidden > 32)"; return AERO_ERR_UNSUPPORTED; }
    if (d->R > 0x7fffffff) { *err = "dconv: grid too large"; return AERO_ERR_ARG; }
    AeroDconvK p;
    p.d = *d;
    p.HP = (d->hidden + 15) / 16 * 16;
    p.hs = d->hidden;
    p.K1p = (3 * d->C + 31) / 32 * 32;
    p.logp = aero_dconv_logp(d->C);
    p.T16 = (d->T + 15) / 16 * 16;
    p.PADR = maxdil;
    const size_t lds = aero_dconv_lds_bytes(d->T, d->C, d->hidden, maxdil);
    dim3 grid((unsigned)d->R), block(512);
    if (p.HP == 16) AERO_LAUNCH_DYN((aero_dconv_row_kernel<1>), grid, block, lds, stream, p);
    else AERO_LAUNCH_DYN((aero_dconv_row_kernel<2>), grid, block, lds, stream, p);
    return AERO_OK;
}
