// k_norm.h -- GroupNorm statistics + fused normalise/activation (HBM-bound elementwise work).
//
// Replaces nn.GroupNorm + F.gelu / F.glu / Snake / LayerScale / residual add of the reference
// (aero.py:127,133,198,206-214; modules.py:230-244; snake.py:67).
//   aero_norm_stats : per (item, group) sum and sum of squares, fp32 per thread, fp64 across threads, many
//                     blocks per group combined with fp64 atomics (caller zeroes the 2-double accumulators).
//   aero_norm_apply : mean/rstd are derived from the accumulators in the block preamble and folded with
//                     gamma/beta into one per-channel FMA held in registers; each thread owns a FIXED vector of
//                     8 channels and walks time, so the inner loop is 16-byte load -> 8 FMA -> activation ->
//                     16-byte store with no index arithmetic (no integer division: v1 spent its time there).
// Algorithmic bytes: stats 2 B/element read; apply 2 B read + 2 B write per output element (GLU reads 4 B per
// output; +2 B for a residual) -- DESIGN.md section 4.
#pragma once
#include <stdlib.h>

#include "aero_common.h"

template <int VEC>
struct AeroVecT;
template <>
struct AeroVecT<8> { typedef h16x8 type; };
template <>
struct AeroVecT<4> { typedef h16x4 type; };
template <>
struct AeroVecT<1> { typedef h16 type; };

template <int VEC>
static __device__ __forceinline__ void aero_load_vec(const h16* p, float* out) {
    if (VEC == 8) {
        const h16x8 v = *(const h16x8*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
    } else if (VEC == 4) {
        const h16x4 v = *(const h16x4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = (float)v[i];
    } else {
        out[0] = (float)p[0];
    }
}

// raw (unconverted) vector load and its conversion, so that several loads can be issued before the first is used
template <int VEC>
static __device__ __forceinline__ typename AeroVecT<VEC>::type aero_load_raw(const h16* p) {
    return *(const typename AeroVecT<VEC>::type*)p;
}
template <int VEC>
static __device__ __forceinline__ void aero_cvt_vec(const typename AeroVecT<VEC>::type& v, float* out) {
    if constexpr (VEC == 1) {
        out[0] = (float)v;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = (float)v[i];
    }
}

template <int VEC>
static __device__ __forceinline__ void aero_store_vec(h16* p, const float* in) {
    if (VEC == 8) {
        h16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (h16)in[i];
        *(h16x8*)p = v;
    } else if (VEC == 4) {
        h16x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (h16)in[i];
        *(h16x4*)p = v;
    } else {
        p[0] = (h16)in[0];
    }
}

// grid (t-chunks, nf, items*G), nf = F for per_row == 0 else 1.  Thread (v, ty): channel vector v of the group,
// time phase ty; walks t = t0+ty, t0+ty+TY, ...
template <int VEC>
__global__ __launch_bounds__(256) void aero_norm_stats_kernel(aero_norm_desc d, int tchunk) {
    __shared__ double red[2][4];
    const int gs = d.C / d.G;
    const int item = blockIdx.z / d.G, g = blockIdx.z % d.G;
    const int b = d.per_row == 1 ? item / d.F : item;
    const int f = d.per_row == 1 ? item % d.F : blockIdx.y;
    const int vpp = gs / VEC;
    const int tid = threadIdx.x;
    const bool wide = vpp > 256;
    const int TY = wide ? 1 : 256 / vpp;
    const int v = wide ? tid : tid % vpp;
    const int ty = wide ? 0 : tid / vpp;
    const int vstep = wide ? 256 : vpp;
    const h16* base = (const h16*)d.src + (int64_t)b * d.s_b + (int64_t)f * d.s_f + (int64_t)g * gs;
    const int t0 = blockIdx.x * tchunk;
    const int t1 = (t0 + tchunk < d.T) ? t0 + tchunk : d.T;
    float s = 0.f, ss = 0.f;
    if (ty < TY) {
        if (!wide) {
            // four time steps per trip.  Loads are UNCONDITIONAL (time index clamped into the chunk; a step past the end contributes
            // zeros through a select) and the next trip's are issued before this trip's sums: with `if (t < t1) load` every load sat in
            // a branch and hipcc waited vmcnt(0) in front of the first use (round 4: same finding as k_pw.h / aero_norm_apply_fast).
            // The order of the fp32 additions is unchanged: results are bit-identical to the predicated form.
            const h16* colp = base + v * VEC;
            const int tl = t1 - 1;
            typename AeroVecT<VEC>::type r[4], rn[4];
            auto fetch = [&](typename AeroVecT<VEC>::type (&q)[4], int t) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int tk = t + k * TY;
                    tk = tk < tl ? tk : tl;
                    q[k] = aero_load_raw<VEC>(colp + (int64_t)tk * d.s_t);
                }
            };
            int t = t0 + ty;
            if (t < t1) fetch(r, t);
#ifndef AERO_EMU
            __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): the loop is entered with nothing in flight
#endif
#pragma unroll 1
            for (; t < t1; t += 4 * TY) {
                const int tn = t + 4 * TY;
                fetch(rn, tn < t1 ? tn : t);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool in = t + k * TY < t1;
                    float x[VEC];
                    aero_cvt_vec<VEC>(r[k], x);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float xv = in ? x[i] : 0.f;
                        s += xv;
                        ss += xv * xv;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = rn[k];
            }
        } else {
            for (int t = t0 + ty; t < t1; t += TY) {
                const h16* row = base + (int64_t)t * d.s_t;
                for (int vv = v; vv < vpp; vv += vstep) {
                    float x[VEC];
                    aero_load_vec<VEC>(row + vv * VEC, x);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { s += x[i]; ss += x[i] * x[i]; }
                }
            }
        }
    }
    double ds = aero_wave_sum((double)s), dss = aero_wave_sum((double)ss);
    if (aero_lane() == 0) { red[0][aero_wave()] = ds; red[1][aero_wave()] = dss; }
    __syncthreads();
    if (tid == 0) {
        // per_row == 2: one accumulator pair per group for the WHOLE batch (BatchNorm in training mode, modules.py:287)
        const int64_t slot = d.per_row == 2 ? g : (int64_t)blockIdx.z;
        atomicAdd(d.stats + slot * 2 + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(d.stats + slot * 2 + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// grid (t-chunks, F, B); thread (v, ty) owns output channels [v*VEC, v*VEC+VEC) for all its time steps.
template <int VEC>
__global__ __launch_bounds__(256) void aero_norm_apply_kernel(aero_norm_desc d, int tchunk) {
    const bool glu = d.act == AERO_ACT_GLU;
    const int Cout = glu ? d.C / 2 : d.C;
    const int vpp = Cout / VEC;
    const int TY = 256 / vpp;
    const int tid = threadIdx.x;
    const int v = tid % vpp, ty = tid / vpp;
    if (ty >= TY) return;
    const int b = blockIdx.z, f = blockIdx.y;
    const int item = d.per_row == 1 ? b * d.F + f : (d.per_row == 2 ? 0 : b);
    const int gs = d.C / d.G;
    const int c0 = v * VEC;
    // y = x*A + Bc  (normalisation and affine folded), per owned channel; second half for GLU gates.
    // The statistics are turned into (rstd, -mean*rstd) ONCE per group this thread touches: fp64 only for the
    // cancellation-prone E[x^2] - mean^2, one reciprocal of the count, a float rsqrt with one Newton step.  (Doing a
    // double division and a double sqrt per channel made this preamble ~2400 instructions per thread -- as much as the
    // whole streaming loop -- and the kernel issue-bound at 3.5 TB/s.)
    float A[VEC], Bc[VEC], A2[VEC], B2[VEC], ls[VEC];
    const double inv_count = d.stats ? 1.0 / d.stat_count : 0.0;
    int g_prev = -1;
    float g_a = 1.f, g_b = 0.f;
    auto group_ab = [&](int g) {
        if (g == g_prev) return;
        g_prev = g;
        const double* st = d.stats + ((int64_t)item * d.G + g) * 2;
        const double mean = st[0] * inv_count;
        double var = st[1] * inv_count - mean * mean;              // biased variance, as nn.GroupNorm
        if (var < 0) var = 0;
        const float vf = (float)var + d.eps;
        float r = aero_rsqrt(vf);
        r = r * (1.5f - 0.5f * vf * r * r);                        // Newton step: full fp32 accuracy
        g_a = r;
        g_b = -(float)mean * r;
    };
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int c = c0 + i;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cc = c + half * Cout;
            float a = 1.f, bb = 0.f;
            if (half == 0 || glu) {
                if (d.stats) {
                    group_ab(cc / gs);
                    a = g_a;
                    bb = g_b;
                }
                if (d.gamma) {
                    const float gm = d.gamma[cc], bt = d.beta[cc];
                    a *= gm;
                    bb = bb * gm + bt;
                }
            }
            if (half == 0) { A[i] = a; Bc[i] = bb; } else { A2[i] = a; B2[i] = bb; }
        }
        ls[i] = (glu && d.layer_scale) ? d.layer_scale[c] : 1.f;
    }
    float snake_a = 0.f, snake_ia = 0.f;
    if (d.act == AERO_ACT_SNAKE) { snake_a = d.snake_a[f]; snake_ia = 1.0f / snake_a; }
    const h16* src = (const h16*)d.src + (int64_t)b * d.s_b + (int64_t)f * d.s_f + c0;
    const h16* res = d.res ? (const h16*)d.res + (int64_t)b * d.r_b + (int64_t)f * d.r_f + c0 : nullptr;
    h16* dst = (h16*)d.dst + (int64_t)b * d.d_b + (int64_t)f * d.d_f + c0;
    const int t0 = blockIdx.x * tchunk;
    const int t1 = (t0 + tchunk < d.T) ? t0 + tchunk : d.T;
    // Fold what can be folded into the per-channel FMA coefficients: LayerScale into (A, Bc); -log2(e) into the gate's
    // (A2, B2) so that sigmoid(v) = rcp(1 + exp2(v')) needs no multiply.
    if (glu) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            A[i] *= ls[i]; Bc[i] *= ls[i];
            A2[i] *= -1.4426950408889634f; B2[i] *= -1.4426950408889634f;
        }
    }
    // one time step: x (and the GLU gate z, the residual r) -> o.  The arithmetic runs on PAIRS of channels (f32x2): the
    // affine part, the erf polynomial and the products compile to v_pk_fma/mul_f32, two elements per instruction; only
    // exp2 / rcp / sin stay scalar.  (These kernels are VALU-bound, not HBM-bound: 16 B in -> ~100-200 instructions.)
    auto step = [&](const float* x, const float* z, const float* r, float* o) {
        if constexpr (VEC >= 2) {
#pragma unroll
            for (int j = 0; j < VEC / 2; ++j) {
                const f32x2 xp = {x[2 * j], x[2 * j + 1]};
                const f32x2 Ap = {A[2 * j], A[2 * j + 1]}, Bp = {Bc[2 * j], Bc[2 * j + 1]};
                const f32x2 y = xp * Ap + Bp;
                f32x2 q;
                if (glu) {
                    const f32x2 zp = {z[2 * j], z[2 * j + 1]};
                    const f32x2 A2p = {A2[2 * j], A2[2 * j + 1]}, B2p = {B2[2 * j], B2[2 * j + 1]};
                    const f32x2 v = zp * A2p + B2p;
                    const f32x2 e = {aero_exp2(v[0]), aero_exp2(v[1])};
                    const f32x2 u = e + 1.0f;
                    const f32x2 sg = {aero_rcp(u[0]), aero_rcp(u[1])};
                    q = y * sg;
                } else if (d.act == AERO_ACT_GELU) {
                    q = aero_gelu2(y);
                } else if (d.act == AERO_ACT_RELU) {
                    q = (f32x2){fmaxf(y[0], 0.f), fmaxf(y[1], 0.f)};
                } else if (d.act == AERO_ACT_SNAKE) {
                    const f32x2 ya = y * snake_a;
                    const f32x2 sn = {aero_fast_sin(ya[0]), aero_fast_sin(ya[1])};
                    q = y + (sn * sn) * snake_ia;
                } else {
                    q = y;
                }
                if (res) q += (f32x2){r[2 * j], r[2 * j + 1]};
                o[2 * j] = q[0];
                o[2 * j + 1] = q[1];
            }
        } else {
            const float y = x[0] * A[0] + Bc[0];
            float q;
            if (glu) q = y * aero_rcp(1.0f + aero_exp2(z[0] * A2[0] + B2[0]));
            else if (d.act == AERO_ACT_GELU) q = aero_gelu(y);
            else if (d.act == AERO_ACT_RELU) q = fmaxf(y, 0.f);
            else if (d.act == AERO_ACT_SNAKE) { const float sn = aero_fast_sin(y * snake_a); q = y + snake_ia * sn * sn; }
            else q = y;
            if (res) q += r[0];
            o[0] = q;
        }
    };
    // UNR time steps per trip with every load issued before the first use (a wave keeps UNR KiB in flight; with one
    // load per trip the kernel sat at 3.5 TB/s, latency-bound)
    constexpr int UNR = 4;
    typedef typename AeroVecT<VEC>::type raw_t;
    // (one loop, the last trip predicated per step: a scalar remainder loop -- one load in flight per thread -- ran
    // 23 of the 63 steps of a typical chunk latency-bound)
    for (int t = t0 + ty; t < t1; t += UNR * TY) {
        raw_t rx[UNR], rz[UNR], rr[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int tk = t + k * TY;
            if (tk < t1) {
                rx[k] = aero_load_raw<VEC>(src + (int64_t)tk * d.s_t);
                if (glu) rz[k] = aero_load_raw<VEC>(src + (int64_t)tk * d.s_t + Cout);
                if (res) rr[k] = aero_load_raw<VEC>(res + (int64_t)tk * d.r_t);
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int tk = t + k * TY;
            if (tk < t1) {
                float x[VEC], z[VEC], r[VEC], o[VEC];
                aero_cvt_vec<VEC>(rx[k], x);
                if (glu) aero_cvt_vec<VEC>(rz[k], z);
                if (res) aero_cvt_vec<VEC>(rr[k], r);
                step(x, z, r, o);
                aero_store_vec<VEC>(dst + (int64_t)tk * d.d_t, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// aero_norm_stats for NARROW groups (round 4): with 48-channel groups a block of the kernel above reads a 96-byte piece of every 384-byte
// position -- four blocks (one per group) each pull the same lines (the second decoder's GroupNorm: 74 us at 1.8 TB/s).  Here a block
// reads whole positions (all C channels); a thread's 8-channel vector lies in one group (gs % 8 == 0), its fp32 partial sums are added
// to the block's per-group fp64 pair in LDS and the block adds G pairs to the global sums.  Same per-thread summation order as the
// kernel above (thread = (channel vector, time phase)); the cross-thread order differs only at fp64 rounding level.
__global__ __launch_bounds__(256) void aero_norm_stats_rows_kernel(aero_norm_desc d, int tchunk) {
    __shared__ double red[2][64];
    const int gs = d.C / d.G;
    const int item = blockIdx.z;
    const int b = d.per_row == 1 ? item / d.F : item;
    const int f = d.per_row == 1 ? item % d.F : blockIdx.y;
    const int vpp = d.C / 8;
    const int TY = 256 / vpp;
    const int tid = threadIdx.x;
    const int v = tid % vpp, ty = tid / vpp;
    for (int i = tid; i < 2 * 64; i += 256) (&red[0][0])[i] = 0.0;
    __syncthreads();
    const h16* colp = (const h16*)d.src + (int64_t)b * d.s_b + (int64_t)f * d.s_f + v * 8;
    const int t0 = blockIdx.x * tchunk;
    const int t1 = (t0 + tchunk < d.T) ? t0 + tchunk : d.T;
    float s = 0.f, ss = 0.f;
    if (ty < TY) {
        const int tl = t1 - 1;
        h16x8 r[4], rn[4];
        auto fetch = [&](h16x8 (&q)[4], int t) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int tk = t + k * TY;
                tk = tk < tl ? tk : tl;
                q[k] = *(const h16x8*)(colp + (int64_t)tk * d.s_t);
            }
        };
        int t = t0 + ty;
        if (t < t1) fetch(r, t);
#ifndef AERO_EMU
        __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
#pragma unroll 1
        for (; t < t1; t += 4 * TY) {
            const int tn = t + 4 * TY;
            fetch(rn, tn < t1 ? tn : t);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = t + k * TY < t1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xv = in ? (float)r[k][i] : 0.f;
                    s += xv;
                    ss += xv * xv;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = rn[k];
        }
        const int g = (v * 8) / gs;
        atomicAdd(&red[0][g], (double)s);
        atomicAdd(&red[1][g], (double)ss);
    }
    __syncthreads();
    if (tid < d.G) {
        atomicAdd(d.stats + ((int64_t)item * d.G + tid) * 2 + 0, red[0][tid]);
        atomicAdd(d.stats + ((int64_t)item * d.G + tid) * 2 + 1, red[1][tid]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// aero_norm_apply, fast form (round 4) for 8-channel vectors.  Same arithmetic as aero_norm_apply_kernel<8>; what changes is what the
// streaming pointwise kernel (k_pw.h) taught about hipcc and memory latency:
//   * the activation and the residual are TEMPLATE parameters: nothing is decided per element;
//   * the per-channel coefficients (statistics -> rstd, gamma, beta, LayerScale, the GLU gate's -log2 e) are computed ONCE PER BLOCK into
//     LDS by the first threads instead of by every thread for its own channels (fp64 moments per thread were the whole cost of a block
//     that then streams six time steps);
//   * loads are UNCONDITIONAL (time index clamped into the chunk) and used raw; a predicated load is a branch, and with branches between
//     a prefetch and its use the compiler waits vmcnt(0) -- for the loads it has just issued;
//   * the next trip's loads are issued before the current trip's arithmetic; the loop is entered with nothing in flight.
// grid (t-chunks, F, B) and thread -> (channel vector, time lane) as the general kernel: results are bit-identical to it.
template <int ACT, bool RES>
__global__ __launch_bounds__(256) void aero_norm_apply_fast_kernel(aero_norm_desc d, int tchunk) {
    constexpr int VEC = 8, UNR = 4;
    constexpr bool GLU = ACT == AERO_ACT_GLU;
    float* cA = (float*)AERO_DYN_SMEM;                           // [Cout] x {A, B, A2, B2}
    const int Cout = GLU ? d.C / 2 : d.C;
    float* cB = cA + Cout;
    float* cA2 = cB + Cout;
    float* cB2 = cA2 + Cout;
    const int vpp = Cout / VEC;
    const int TY = 256 / vpp;
    const int tid = threadIdx.x;
    const int b = blockIdx.z, f = blockIdx.y;
    const int item = d.per_row == 1 ? b * d.F + f : (d.per_row == 2 ? 0 : b);
    const int gs = d.C / d.G;
    const double inv_count = d.stats ? 1.0 / d.stat_count : 0.0;
    for (int cc = tid; cc < d.C; cc += 256) {                    // conv channel cc: value half (cc < Cout) or gate half
        float a = 1.f, bb = 0.f;
        if (d.stats) {
            const double* st = d.stats + ((int64_t)item * d.G + cc / gs) * 2;
            const double mean = st[0] * inv_count;
            double var = st[1] * inv_count - mean * mean;
            if (var < 0) var = 0;
            const float vf = (float)var + d.eps;
            float r = aero_rsqrt(vf);
            r = r * (1.5f - 0.5f * vf * r * r);
            a = r;
            bb = -(float)mean * r;
        }
        if (d.gamma) {
            const float gm = d.gamma[cc], bt = d.beta[cc];
            a *= gm;
            bb = bb * gm + bt;
        }
        if (cc < Cout) {
            const float ls = (GLU && d.layer_scale) ? d.layer_scale[cc] : 1.f;
            cA[cc] = GLU ? a * ls : a;
            cB[cc] = GLU ? bb * ls : bb;
        } else {
            cA2[cc - Cout] = a * -1.4426950408889634f;
            cB2[cc - Cout] = bb * -1.4426950408889634f;
        }
    }
    __syncthreads();
    const int v = tid % vpp, ty = tid / vpp;
    if (ty >= TY) return;
    const int c0 = v * VEC;
    float A[VEC], Bc[VEC], A2[VEC], B2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        A[i] = cA[c0 + i];
        Bc[i] = cB[c0 + i];
        A2[i] = GLU ? cA2[c0 + i] : 0.f;
        B2[i] = GLU ? cB2[c0 + i] : 0.f;
    }
    float snake_a = 0.f, snake_ia = 0.f;
    if constexpr (ACT == AERO_ACT_SNAKE) { snake_a = d.snake_a[f]; snake_ia = 1.0f / snake_a; }
    const h16* src = (const h16*)d.src + (int64_t)b * d.s_b + (int64_t)f * d.s_f + c0;
    const h16* res = RES ? (const h16*)d.res + (int64_t)b * d.r_b + (int64_t)f * d.r_f + c0 : nullptr;
    h16* dst = (h16*)d.dst + (int64_t)b * d.d_b + (int64_t)f * d.d_f + c0;
    const int t0 = blockIdx.x * tchunk;
    const int t1 = (t0 + tchunk < d.T) ? t0 + tchunk : d.T;
    const int tl = t1 - 1;
    h16x8 cx[UNR], cz[UNR], cr[UNR], nx[UNR], nz[UNR], nr[UNR];
    auto fetch = [&](h16x8 (&rx)[UNR], h16x8 (&rz)[UNR], h16x8 (&rr)[UNR], int t) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            int tk = t + k * TY;
            tk = tk < tl ? tk : tl;                                // (clamped: the store is what is masked)
            rx[k] = *(const h16x8*)(src + (int64_t)tk * d.s_t);
            if constexpr (GLU) rz[k] = *(const h16x8*)(src + (int64_t)tk * d.s_t + Cout);
            if constexpr (RES) rr[k] = *(const h16x8*)(res + (int64_t)tk * d.r_t);
        }
    };
    int t = t0 + ty;
    if (t < t1) fetch(cx, cz, cr, t);
#ifndef AERO_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): clean scoreboard at the loop header (see k_pw.h)
#endif
#pragma unroll 1
    for (; t < t1; t += UNR * TY) {
        const int tn = t + UNR * TY;
        fetch(nx, nz, nr, tn < t1 ? tn : t);
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int tk = t + k * TY;
            h16x8 o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float y = (float)cx[k][i] * A[i] + Bc[i];
                float qv;
                if constexpr (GLU) {
                    const float g = (float)cz[k][i] * A2[i] + B2[i];
                    qv = y * aero_rcp(aero_exp2(g) + 1.0f);
                } else if constexpr (ACT == AERO_ACT_GELU) {
                    qv = y;                                        // (pairs below: aero_gelu2 as the general kernel)
                } else if constexpr (ACT == AERO_ACT_RELU) {
                    qv = fmaxf(y, 0.f);
                } else if constexpr (ACT == AERO_ACT_SNAKE) {
                    const float sn = aero_fast_sin(y * snake_a);
                    qv = y + (sn * sn) * snake_ia;
                } else {
                    qv = y;
                }
                if constexpr (ACT != AERO_ACT_GELU) {
                    if constexpr (RES) qv += (float)cr[k][i];
                    o[i] = (h16)qv;
                } else {
                    o[i] = (h16)0.f;
                    if (i & 1) {
                        const float y0 = (float)cx[k][i - 1] * A[i - 1] + Bc[i - 1];
                        f32x2 gg = aero_gelu2((f32x2){y0, y});
                        if constexpr (RES) gg += (f32x2){(float)cr[k][i - 1], (float)cr[k][i]};
                        o[i - 1] = (h16)gg[0];
                        o[i] = (h16)gg[1];
                    }
                }
            }
            if (tk < t1) *(h16x8*)(dst + (int64_t)tk * d.d_t) = o;
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            cx[k] = nx[k];
            if constexpr (GLU) cz[k] = nz[k];
            if constexpr (RES) cr[k] = nr[k];
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
    if (d->B > 65535 || d->F > 65535) { *err = "norm: B and F must be <= 65535"; return AERO_ERR_ARG; }
    return AERO_OK;
}

// time steps per block: enough blocks to fill 256 CUs several times over, at least 2 passes per thread
// Blocks are launched in memory order (t-chunks of a row fastest) and each takes a SHORT contiguous piece: the set of
// blocks in flight then covers a few MB of adjacent addresses, like a plain streaming kernel.  (One block per 96-KiB
// row meant ~1300 independent DRAM streams in flight and 3.5 TB/s.)  AERO_NORM_CHUNK_KB overrides the piece size.
static int aero_norm_tchunk(int T, int TY, int64_t rows, int row_bytes_per_step) {
    static int kb = -1;
    if (kb < 0) { const char* e = getenv("AERO_NORM_CHUNK_KB"); kb = e ? atoi(e) : 0; }
    int chunks = (int)((4096 + rows - 1) / rows);
    if (chunks < 1) chunks = 1;
    int tchunk = (T + chunks - 1) / chunks;
    const int unit = TY * 4;                                     // one unrolled trip of every thread
    if (kb > 0) {
        int want = (kb * 1024 + row_bytes_per_step - 1) / row_bytes_per_step;
        want = (want + unit - 1) / unit * unit;
        if (want < tchunk) tchunk = want;
    }
    if (tchunk < unit) tchunk = unit;
    if (tchunk > T) tchunk = T;
    return tchunk;
}

static int aero_norm_stats_launch(const aero_norm_desc* d, hipStream_t stream, const char** err) {
    int rc = aero_norm_check(d, err);
    if (rc) return rc;
    if (!d->stats) { *err = "norm_stats: null stats"; return AERO_ERR_ARG; }
    const int gs = d->C / d->G;
    const int vec = aero_norm_pick_vec(gs, d->src, nullptr, nullptr, d->s_b, d->s_f, d->s_t, 0, 0, 0, 0, 0, 0);
    const int64_t items = d->per_row == 1 ? (int64_t)d->B * d->F : d->B;
    if (items * d->G > 65535) { *err = "norm_stats: more than 65535 (item, group) pairs in one launch"; return AERO_ERR_ARG; }
    const int nf = d->per_row == 1 ? 1 : d->F;
    const int vpp = gs / vec;
    const int TY = vpp > 256 ? 1 : 256 / vpp;
    // (chunking from the rows of ONE item x 64, not from the batch: a clip's partial sums -- fp32 per thread over a chunk --
    // must not depend on how many other clips share the launch (x 64: the batch the chunk sizes were tuned on), or its output would differ between batch sizes / ranks)
    static int rows_on = -1;
    if (rows_on < 0) { const char* e = getenv("AERO_NORM_STATS_ROWS"); rows_on = e ? atoi(e) : 1; }
    if (rows_on && vec == 8 && d->per_row != 2 && d->G > 1 && d->G <= 64 && gs % 8 == 0 && gs * 2 < 256 && d->C / 8 <= 256) {
        const int TYr = 256 / (d->C / 8);
        const int tchunk_r = aero_norm_tchunk(d->T, TYr, (items / d->B) * 64 * nf, d->C * 2);
        dim3 gridr((unsigned)((d->T + tchunk_r - 1) / tchunk_r), (unsigned)nf, (unsigned)items);
        AERO_LAUNCH(aero_norm_stats_rows_kernel, gridr, dim3(256), stream, *d, tchunk_r);
        return AERO_OK;
    }
    const int tchunk = aero_norm_tchunk(d->T, TY, (items / d->B) * 64 * d->G * nf, gs * 2);
    dim3 grid((unsigned)((d->T + tchunk - 1) / tchunk), (unsigned)nf, (unsigned)(items * d->G)), block(256);
    if (vec == 8) AERO_LAUNCH((aero_norm_stats_kernel<8>), grid, block, stream, *d, tchunk);
    else if (vec == 4) AERO_LAUNCH((aero_norm_stats_kernel<4>), grid, block, stream, *d, tchunk);
    else AERO_LAUNCH((aero_norm_stats_kernel<1>), grid, block, stream, *d, tchunk);
    return AERO_OK;
}

static int aero_norm_apply_launch(const aero_norm_desc* d, hipStream_t stream, const char** err) {
    int rc = aero_norm_check(d, err);
    if (rc) return rc;
    if (!d->dst) { *err = "norm_apply: null dst"; return AERO_ERR_ARG; }
    if (d->act == AERO_ACT_GLU && (d->C & 1)) { *err = "norm_apply: GLU needs even C"; return AERO_ERR_ARG; }
    if (d->act == AERO_ACT_SNAKE && !d->snake_a) { *err = "norm_apply: snake needs a"; return AERO_ERR_ARG; }
    if ((d->gamma == nullptr) != (d->beta == nullptr)) { *err = "norm_apply: gamma/beta"; return AERO_ERR_ARG; }
    if (d->stats && !(d->stat_count >= 1.0)) { *err = "norm_apply: stat_count"; return AERO_ERR_ARG; }
    const int Cout = d->act == AERO_ACT_GLU ? d->C / 2 : d->C;
    const int vec = aero_norm_pick_vec(Cout, d->src, d->dst, d->res, d->s_b, d->s_f, d->s_t, d->d_b, d->d_f, d->d_t,
                                       d->res ? d->r_b : 0, d->res ? d->r_f : 0, d->res ? d->r_t : 0);
    const int vpp = Cout / vec;
    if (vpp > 256) { *err = "norm_apply: more than 256 channel vectors per position (C too large)"; return AERO_ERR_UNSUPPORTED; }
    const int TY = 256 / vpp;
    const int tchunk = aero_norm_tchunk(d->T, TY, (int64_t)64 * d->F, d->C * 2);
    dim3 grid((unsigned)((d->T + tchunk - 1) / tchunk), (unsigned)d->F, (unsigned)d->B), block(256);
    static int fast = -1;
    if (fast < 0) { const char* e = getenv("AERO_NORM_FAST"); fast = e ? atoi(e) : 1; }
    if (fast && vec == 8 && (int64_t)d->T * d->s_t < (1ll << 31)) {
        const size_t lds = (size_t)4 * Cout * sizeof(float);
        const bool res = d->res != nullptr;
#define AERO_NORM_FAST_GO(ACT_)                                                                                             \
        do {                                                                                                                \
            if (res) AERO_LAUNCH_DYN((aero_norm_apply_fast_kernel<ACT_, true>), grid, block, lds, stream, *d, tchunk);      \
            else AERO_LAUNCH_DYN((aero_norm_apply_fast_kernel<ACT_, false>), grid, block, lds, stream, *d, tchunk);         \
        } while (0)
        switch (d->act) {
            case AERO_ACT_GLU: AERO_NORM_FAST_GO(AERO_ACT_GLU); break;
            case AERO_ACT_GELU: AERO_NORM_FAST_GO(AERO_ACT_GELU); break;
            case AERO_ACT_RELU: AERO_NORM_FAST_GO(AERO_ACT_RELU); break;
            case AERO_ACT_SNAKE: AERO_NORM_FAST_GO(AERO_ACT_SNAKE); break;
            default: AERO_NORM_FAST_GO(AERO_ACT_NONE); break;
        }
#undef AERO_NORM_FAST_GO
        return AERO_OK;
    }
    if (vec == 8) AERO_LAUNCH((aero_norm_apply_kernel<8>), grid, block, stream, *d, tchunk);
    else if (vec == 4) AERO_LAUNCH((aero_norm_apply_kernel<4>), grid, block, stream, *d, tchunk);
    else AERO_LAUNCH((aero_norm_apply_kernel<1>), grid, block, stream, *d, tchunk);
    return AERO_OK;
}
