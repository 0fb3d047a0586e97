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
//      row image), bias as the accumulator seed -> raw h in REGISTERS (a wave owns at most MAXF fragments); sum / sum of
//      squares -> GN1 statistics
//   B  h -> act(GN1(h)), fp16, still in registers: the accumulator layout of conv1 (lane = time step, 4 consecutive hidden
//      units) IS the B-operand layout of MFMA 16x16x16, so conv2 consumes it without an LDS round trip; conv2 here only
//      for the GN2 statistics (the 2C-channel tensor is never stored: recomputing a K = 16 contraction is cheaper)
//   C  conv2 again, GN2 (one FMA per value), GLU (rows of W2 are interleaved (a0, b0, a1, b1, ...) so a lane holds both
//      halves), LayerScale, + x[t][c] read and written in place by the same lane (conv1's neighbours were consumed in A)
// Block-wide reductions: wave shuffles + NW partials in LDS.  HBM traffic: 4*C bytes per (row, time step); the kernel
// itself is bound by the LDS pipe and the VALU (PMC of the first version: LDS pipe 74-85 % busy, half of it bank
// conflicts on the W2 fragments and the h side buffer -- hence W2 in lane order and h in registers).
#pragma once
#include "aero_common.h"

struct AeroDconvK {
    aero_dconv_desc d;
    int HP;       // hidden rounded up to 16 (rows of the W1 image, k-extent of the W2 image)  [template HM * 16]
    int logp;     // 16 pad bytes after every (1 << logp) rows of the row image: b128 reads of 16 consecutive rows conflict-free
    int T16;      // T rounded up to 16
    int PADR;     // zero rows before t = 0 and after T16 (largest dilation, rounded up to 8 rows: fragment rows stay bank-aligned)
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
    const int padr = (maxdil + 7) / 8 * 8;
    const int xrows = T16 + 2 * padr, logp = aero_dconv_logp(C);
    const size_t xs = (size_t)xrows * C + (size_t)((xrows >> logp) + 1) * 8;
    const size_t w1 = (size_t)HP * K1p, w2 = (size_t)2 * C * HP;
    const size_t halves = (xs + w1 + w2 + 7) / 8 * 8;
    const size_t floats = 3 * HP + 3 * 2 * C + C + 2 * 16 * 2 + 8;
    return halves * 2 + floats * 4;
}

// (sum, sum of squares) of a row -> mean and 1/sqrt(var + eps).  fp64 only for the cancellation-prone E[x^2] - mean^2; a float rsqrt
// with one Newton step instead of a double division and square root (those expand to ~100 instructions each, and EVERY thread
// of the block runs them twice per layer: the first version spent 15 % of its vector instructions there)
static __device__ __forceinline__ void aero_dconv_moments(float s1, float s2, float inv_n, float eps, float& mean, float& rstd) {
    const float m = s1 * inv_n;
    double var = (double)s2 * (double)inv_n - (double)m * (double)m;
    const float vf = fmaxf((float)var, 0.f) + eps;
    float r = aero_rsqrt(vf);
    r = r * (1.5f - 0.5f * vf * r * r);
    mean = m;
    rstd = r;
}

// 8 waves (two blocks per CU when the row is small) or 16 (one block per CU, or more than 32 fragments); 4 fragments per wave
static inline int aero_dconv_nw(size_t lds_bytes, int T) { return (lds_bytes > 80 * 1024 || T > 512) ? 16 : 8; }

// HM = HP/16: M fragments of conv1 = k-steps of conv2;  NF2 = C/8 = M fragments of conv2;  NW waves, each owning at most MAXF
// 16-step column fragments (cf = wave + f*NW): T <= 16 * NW * MAXF
template <int HM, int NF2, int NW, int MAXF>
__global__ __launch_bounds__(NW * 64, 4) void aero_dconv_row_kernel(AeroDconvK p) {     // 4 waves per SIMD: <= 128 registers (two 8-wave blocks per CU)
    constexpr int C = NF2 * 8, HP = HM * 16, NK1 = (3 * C + 31) / 32, K1p = NK1 * 32, CU = NF2, NT = NW * 64;
    const aero_dconv_desc& d = p.d;
    const int T = d.T, hidden = d.hidden;
    const int xrows = p.T16 + 2 * p.PADR;
    h16* xs = (h16*)AERO_DYN_SMEM;                                                 // row image, padded (see xoff)
    h16* w1s = xs + xrows * C + ((xrows >> p.logp) + 1) * 8;                       // [NK1][HP][32] tile-swizzled
    h16* w2s = w1s + HP * K1p;                                                     // [NF2][HM][64 lanes][4]: fragments in lane order
    float* c1 = (float*)(xs + ((int)(w2s + 2 * C * HP - xs) + 7) / 8 * 8);         // [3][HP]   b1 | g1 | be1
    float* c2 = c1 + 3 * HP;                                                       // [3][2C]   b2 | g2 -> A2 | be2 -> B2 (GLU-interleaved)
    float* sc = c2 + 3 * 2 * C;                                                    // [C]       LayerScale
    float* red = sc + C;                                                           // [2][NW][2] reduction partials (after the constant block)
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int row = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int nfrag = p.T16 >> 4;
    auto xoff = [&](int r) { return r * C + ((r >> p.logp) << 3); };

    // the row (and its zero margins) -> LDS, once.  Direct global->LDS copies: every 16-byte unit of the LINEAR LDS image picks its
    // source (a row unit, or the zero page for margins and pad units), all copies are in flight together and the block waits once.
    {
        const h16* src = (const h16*)d.x + (int64_t)row * T * C;
        const int gsz = (CU << p.logp) + 1;                      // units per group of (1 << logp) rows + the pad unit
        const int total = xrows * CU + (xrows >> p.logp) + 1;
        for (int ub = wave * 64; ub < total; ub += NT) {
            const int U = ub + lane;
            const int grp = U / gsz, rem = U - grp * gsz;
            const int rr = rem / CU, s = rem - rr * CU;
            const int t = (grp << p.logp) + rr - p.PADR;
            const h16* sp = (rem < gsz - 1 && t >= 0 && t < T) ? src + (int64_t)t * C + s * 8 : aero_zero_page;
            if (U < total) aero_glds16(sp, xs + ub * 8);
        }
    }
    float snake_a = 0.f, snake_ia = 0.f;
    constexpr int NCONST = 3 * HP + 7 * C;                       // floats of a layer's constant block
    // per-lane LDS offsets (halves / floats), fixed for the whole kernel
    const int w1lane = aero_tile_off(col, g);                    // + kk*HP*32 + mf*512
    const int w2lane = lane * 4;                                 // + (mf*HM + ks)*256
    const bool hin[2] = {g * 4 < hidden, 16 + g * 4 < hidden};   // this lane's four hidden units of k-step 0 / 1 are real
    for (int l = 0; l < d.depth; ++l) {
        const aero_dconv_layer& L = d.layer[l];
        const int dil = L.dilation;
        const bool norm1 = L.norm1 != 0, norm2 = L.norm2 != 0;
        if (l) __syncthreads();                                  // previous layer done with the weights
        for (int ub = wave * 64; ub < HP * (K1p >> 3); ub += NT) {       // W1 image [HP][K1p] -> k-step tiles (swizzle on the source side)
            const int U = ub + lane;
            const int kk = U / (HP * 4), rem = U - kk * (HP * 4);
            const int r = rem >> 2, q = (rem & 3) ^ ((0 - (r >> 2)) & 3);
            if (U < HP * (K1p >> 3)) aero_glds16((const h16*)L.w1 + r * K1p + kk * 32 + q * 8, w1s + ub * 8);
        }
        for (int ub = wave * 64; ub < 2 * C * HP / 8; ub += NT)
            if (ub + lane < 2 * C * HP / 8) aero_glds16((const h16*)L.w2 + (ub + lane) * 8, w2s + ub * 8);
        for (int ub = wave * 64; ub < NCONST / 4; ub += NT)
            if (ub + lane < NCONST / 4) aero_glds16((const h16*)(L.consts + (ub + lane) * 4), (h16*)c1 + ub * 8);
        if (d.act == AERO_ACT_SNAKE) { snake_a = L.snake_a[row % d.F]; snake_ia = 1.0f / snake_a; }
        __syncthreads();                                         // (drains the copies: vmcnt)

        // ---- pass A: conv1 (bias = accumulator seed) -> raw h (registers), GN1 statistics
        float s1 = 0.f, s2 = 0.f;
        f32x4 hraw[MAXF][HM];
#pragma unroll
        for (int f = 0; f < MAXF; ++f) {
            const int cf = wave + f * NW;                        // (wave-uniform)
#pragma unroll
            for (int mf = 0; mf < HM; ++mf) hraw[f][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (cf >= nfrag) continue;
            const int t = cf * 16 + col;
            f32x4 acc[HM];
#pragma unroll
            for (int mf = 0; mf < HM; ++mf) acc[mf] = *(const f32x4*)&c1[mf * 16 + g * 4];      // (zero above `hidden`)
#pragma unroll
            for (int kk = 0; kk < NK1; ++kk) {
                const int k = kk * 32 + g * 8;
                const int tap = (k >= C) + (k >= 2 * C);
                h16x8 bfrag = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (kk * 32 + 24 < 3 * C || k < 3 * C) bfrag = *(const h16x8*)&xs[xoff(t + p.PADR + (tap - 1) * dil) + k - tap * C];
#pragma unroll
                for (int mf = 0; mf < HM; ++mf) {
                    const h16x8 a = *(const h16x8*)&w1s[kk * HP * 32 + mf * 512 + w1lane];
                    acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfrag, acc[mf], 0, 0, 0);
                }
            }
            const float msk = t < T ? 1.f : 0.f;
#pragma unroll
            for (int mf = 0; mf < HM; ++mf) {
                const f32x4 hv = acc[mf];                        // (rows above `hidden`: zero weights, zero bias -> 0)
                s1 = fmaf(msk, (hv[0] + hv[1]) + (hv[2] + hv[3]), s1);
                s2 = fmaf(msk, fmaf(hv[0], hv[0], hv[1] * hv[1]) + fmaf(hv[2], hv[2], hv[3] * hv[3]), s2);
                hraw[f][mf] = hv;
            }
        }
        float mean1 = 0.f, rstd1 = 1.f;
        if (norm1) {
            s1 = aero_wave_sum(s1);
            s2 = aero_wave_sum(s2);
            if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
            __syncthreads();
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a += red[w * 2]; b += red[w * 2 + 1]; }
            aero_dconv_moments(a, b, 1.0f / ((float)hidden * (float)T), d.eps, mean1, rstd1);
        } else {
            __syncthreads();                                     // every wave is done READING its neighbours' x rows before pass C writes x
        }

        // ---- pass B: h <- act(GN1(h)) in place; conv2 (bias = seed) for the GN2 statistics
        s1 = s2 = 0.f;
        f32x4 ga[HM], gb[HM];                                    // GN1 as one FMA: y = h * ga + gb
#pragma unroll
        for (int ks = 0; ks < HM; ++ks) {
            ga[ks] = *(const f32x4*)&c1[HP + ks * 16 + g * 4] * rstd1;
            gb[ks] = *(const f32x4*)&c1[2 * HP + ks * 16 + g * 4] - ga[ks] * mean1;
        }
        h16x4 hB[MAXF][HM];                                      // act(GN1(h)): conv2's B operand, kept for pass C
        float msk[MAXF];
#pragma unroll
        for (int f = 0; f < MAXF; ++f) {
            msk[f] = ((wave + f * NW) * 16 + col < T) ? 1.f : 0.f;
#pragma unroll
            for (int ks = 0; ks < HM; ++ks) {
                h16x4 o = (h16x4){0, 0, 0, 0};
                if (hin[ks]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = fmaf(hraw[f][ks][e], ga[ks][e], gb[ks][e]);
                        if (d.act == AERO_ACT_GELU) y = aero_gelu(y);
                        else if (d.act == AERO_ACT_RELU) y = fmaxf(y, 0.f);
                        else if (d.act == AERO_ACT_SNAKE) { const float sn = aero_fast_sin(y * snake_a); y = fmaf(snake_ia * sn, sn, y); }
                        o[e] = (h16)y;
                    }
                }
                hB[f][ks] = o;
            }
        }
        // (row fragment of W2 and its constants OUTER, the wave's column fragments inner: one LDS fetch serves MAXF fragments --
        //  with the column fragment outer the broadcast reads of the constants alone kept the LDS pipe 40 % busy)
        if (norm2) {
            // sums per column fragment in PAIRS (v_pk_add_f32 / v_pk_fma_f32: four vector instructions per 4 values); the
            // time mask depends on the fragment only and is applied once at the end
            f32x2 p1[MAXF], p2[MAXF];
#pragma unroll
            for (int f = 0; f < MAXF; ++f) { p1[f] = (f32x2){0.f, 0.f}; p2[f] = (f32x2){0.f, 0.f}; }
#pragma unroll
            for (int mf = 0; mf < NF2; ++mf) {
                const f32x4 bias = *(const f32x4*)&c2[mf * 16 + g * 4];
                h16x4 wf[HM];
#pragma unroll
                for (int ks = 0; ks < HM; ++ks) wf[ks] = *(const h16x4*)&w2s[(mf * HM + ks) * 256 + w2lane];
#pragma unroll
                for (int f = 0; f < MAXF; ++f) {
                    if (wave + f * NW >= nfrag) continue;
                    f32x4 v = bias;
#pragma unroll
                    for (int ks = 0; ks < HM; ++ks) v = __builtin_amdgcn_mfma_f32_16x16x16f16(wf[ks], hB[f][ks], v, 0, 0, 0);
                    const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
                    p1[f] += lo + hi;
                    p2[f] = lo * lo + (hi * hi + p2[f]);
                }
            }
#pragma unroll
            for (int f = 0; f < MAXF; ++f) {
                s1 = fmaf(msk[f], p1[f][0] + p1[f][1], s1);
                s2 = fmaf(msk[f], p2[f][0] + p2[f][1], s2);
            }
        }
        float mean2 = 0.f, rstd2 = 1.f;
        if (norm2) {
            s1 = aero_wave_sum(s1);
            s2 = aero_wave_sum(s2);
            if (lane == 0) { red[2 * NW + wave * 2] = s1; red[2 * NW + wave * 2 + 1] = s2; }
            __syncthreads();
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a += red[2 * NW + w * 2]; b += red[2 * NW + w * 2 + 1]; }
            aero_dconv_moments(a, b, 1.0f / ((float)(2 * C) * (float)T), d.eps, mean2, rstd2);
        }
        // GN2 folded into one FMA per value: v' = acc * A2 + B2, A2 = rstd*g2, B2 = be2 + (b2 - mean) * A2  (in place of g2 / be2)
        for (int m = tid; m < 2 * C; m += NT) {
            const float A2 = rstd2 * c2[2 * C + m];
            c2[4 * C + m] = fmaf(c2[m] - mean2, A2, c2[4 * C + m]);
            c2[2 * C + m] = A2;
        }
        __syncthreads();

        // ---- pass C: conv2 again, GN2, GLU, LayerScale, + x in place
        int xlane[MAXF];
#pragma unroll
        for (int f = 0; f < MAXF; ++f) xlane[f] = xoff((wave + f * NW) * 16 + col + p.PADR) + g * 2;
#pragma unroll
        for (int mf = 0; mf < NF2; ++mf) {
            const f32x4 A2 = *(const f32x4*)&c2[2 * C + mf * 16 + g * 4], B2 = *(const f32x4*)&c2[4 * C + mf * 16 + g * 4];
            const f32x2 ls = *(const f32x2*)&sc[mf * 8 + g * 2];
            h16x4 wf[HM];
#pragma unroll
            for (int ks = 0; ks < HM; ++ks) wf[ks] = *(const h16x4*)&w2s[(mf * HM + ks) * 256 + w2lane];
#pragma unroll
            for (int f = 0; f < MAXF; ++f) {
                if (wave + f * NW >= nfrag) continue;
                h16* xr = &xs[xlane[f] + mf * 8];
                const h16x2 xv = *(const h16x2*)xr;
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < HM; ++ks) v = __builtin_amdgcn_mfma_f32_16x16x16f16(wf[ks], hB[f][ks], v, 0, 0, 0);
                v = v * A2 + B2;
                constexpr float NL2E = -1.4426950408889634f;
                const float g0 = aero_rcp(1.f + aero_exp2(v[1] * NL2E)), g1 = aero_rcp(1.f + aero_exp2(v[3] * NL2E));
                const float o0 = fmaf(v[0] * g0, ls[0], (float)xv[0]);
                const float o1 = fmaf(v[2] * g1, ls[1], (float)xv[1]);
                if (msk[f] != 0.f) *(h16x2*)xr = (h16x2){(h16)o0, (h16)o1};
            }
        }
    }
    __syncthreads();
    {
        h16* dst = (h16*)d.y + (int64_t)row * T * C;
        for (int u = tid; u < T * CU; u += NT) {
            const int t = u / CU, s = u - t * CU;
            *(h16x8*)(dst + (int64_t)t * C + s * 8) = *(const h16x8*)&xs[xoff(t + p.PADR) + s * 8];
        }
    }
}

static int aero_dconv_row_fits_impl(int T, int C, int hidden, int maxdil) {
    const size_t b = aero_dconv_lds_bytes(T, C, hidden, maxdil);
    const int nf2 = C / 8;
    const bool inst = nf2 == 2 || nf2 == 4 || nf2 == 6 || nf2 == 8 || nf2 == 12 || nf2 == 16;      // instantiated widths
    return b != 0 && b <= 160 * 1024 && inst && T <= (nf2 >= 12 && aero_dconv_nw(b, T) == 16 ? 512 : 1024);
}

static int aero_dconv_launch(const aero_dconv_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->x || !d->y) { *err = "dconv: null pointer"; return AERO_ERR_ARG; }
    if (d->R < 1 || d->T < 1 || d->depth < 1 || d->depth > AERO_DCONV_MAX_DEPTH || d->F < 1) { *err = "dconv: bad geometry"; return AERO_ERR_ARG; }
    if (d->act != AERO_ACT_RELU && d->act != AERO_ACT_GELU && d->act != AERO_ACT_SNAKE && d->act != AERO_ACT_NONE) { *err = "dconv: unsupported act"; return AERO_ERR_UNSUPPORTED; }
    int maxdil = 1;
    for (int l = 0; l < d->depth; ++l) {
        const aero_dconv_layer& L = d->layer[l];
        if (!L.w1 || !L.w2 || !L.consts) { *err = "dconv: null layer weights"; return AERO_ERR_ARG; }
        if (d->act == AERO_ACT_SNAKE && !L.snake_a) { *err = "dconv: snake needs a"; return AERO_ERR_ARG; }
        if (L.dilation < 1) { *err = "dconv: dilation"; return AERO_ERR_ARG; }
        if ((((uintptr_t)L.w1 | (uintptr_t)L.w2 | (uintptr_t)L.consts) & 15)) { *err = "dconv: unaligned weights"; return AERO_ERR_ARG; }
        maxdil = L.dilation > maxdil ? L.dilation : maxdil;
    }
    if ((((uintptr_t)d->x | (uintptr_t)d->y) & 15)) { *err = "dconv: unaligned rows"; return AERO_ERR_ARG; }
    if (!aero_dconv_row_fits_impl(d->T, d->C, d->hidden, maxdil)) { *err = "dconv: row does not fit the LDS (or C % 8, hidden % 4, hidden > 32)"; return AERO_ERR_UNSUPPORTED; }
    if (d->R > 0x7fffffff) { *err = "dconv: grid too large"; return AERO_ERR_ARG; }
    AeroDconvK p;
    p.d = *d;
    p.HP = (d->hidden + 15) / 16 * 16;
    p.logp = aero_dconv_logp(d->C);
    p.T16 = (d->T + 15) / 16 * 16;
    p.PADR = (maxdil + 7) / 8 * 8;
    const size_t lds = aero_dconv_lds_bytes(d->T, d->C, d->hidden, maxdil);
    const int nw = aero_dconv_nw(lds, d->T), nf2 = d->C / 8;
    dim3 grid((unsigned)d->R), block((unsigned)nw * 64);
#define AERO_DCONV_GO(NF2_)                                                                                            \
    do {                                                                                                               \
        constexpr int MF16 = NF2_ >= 12 ? 2 : 4;     /* wide rows: 16 waves x 2 fragments (T <= 512; longer rows do not fit the LDS) */ \
        if (nw == 16 && (d->T + 15) / 16 > 16 * MF16) { *err = "dconv: row too long"; return AERO_ERR_UNSUPPORTED; }    \
        if (p.HP == 16 && nw == 8) AERO_LAUNCH_DYN((aero_dconv_row_kernel<1, NF2_, 8, 4>), grid, block, lds, stream, p);  \
        else if (p.HP == 16) AERO_LAUNCH_DYN((aero_dconv_row_kernel<1, NF2_, 16, MF16>), grid, block, lds, stream, p);       \
        else if (nw == 8) AERO_LAUNCH_DYN((aero_dconv_row_kernel<2, NF2_, 8, 4>), grid, block, lds, stream, p);           \
        else AERO_LAUNCH_DYN((aero_dconv_row_kernel<2, NF2_, 16, MF16>), grid, block, lds, stream, p);                       \
    } while (0)
    switch (nf2) {
        case 2: AERO_DCONV_GO(2); break;
        case 4: AERO_DCONV_GO(4); break;
        case 6: AERO_DCONV_GO(6); break;
        case 8: AERO_DCONV_GO(8); break;
        case 12: AERO_DCONV_GO(12); break;
        default: AERO_DCONV_GO(16); break;
    }
#undef AERO_DCONV_GO
    return AERO_OK;
}
