// k_gconv_edge.h -- the two ends of the MelGAN critic (discriminators.py:17-19,46-47), whose shapes fit neither the MFMA forms nor the
// generic VALU kernels of k_disc.h well:
//   "c1": Conv1d(1 -> 16, k = 15) behind ReflectionPad1d(7) on the raw waveform (441 000 steps at scale 0): 240 MACs per output step,
//         HBM-bound by its 32-byte output rows.  One thread per step; the waveform span and the 240 weights sit in LDS.
//   "o1": Conv1d(1024 -> 1, k = 3, padding 1): a 3072-term dot product per output step; one wave per step.
// Forward, data gradient and weight gradient of each.  The generic kernels spent 0.3-0.75 ms per launch here (one-sixteenth of
// their channel staging used); these are bound by the bytes they move.  Weight gradients go to per-block slabs added in order by
// aero_wgrad_finish_kernel (deterministic).
#pragma once
#include "aero_common.h"

struct AeroEdgeK {
    const h16* x; const h16* w; const float* bias; const h16* y; const h16* dy; h16* out; float* slabs;
    int B, T, C, K, pad, reflect, tiles_per_block, ntile;
    float slope;
    int64_t sl_stride;
};

static __device__ __forceinline__ int aero_reflect(int t, int T) {
    if (t < 0) t = -t;
    if (t >= T) t = 2 * (T - 1) - t;
    return t;
}

// ---- one input channel: x [B][T], w fp16 [C][K], y [B][T][C], C <= 16 (multiple of 8), K <= 16, stride 1, Tout = T
__global__ __launch_bounds__(256) void aero_gconv_c1_fwd_kernel(AeroEdgeK p) {
    __shared__ float xs[256 + 16];
    __shared__ AERO_LDS_ALIGN float ws[16 * 16];                 // [k][o]
    __shared__ float bs[16];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * 256;
    const h16* xb = p.x + (int64_t)b * p.T;
    for (int i = tid; i < 256 + p.K - 1; i += 256) {
        int t = t0 - p.pad + i;
        if (p.reflect) t = aero_reflect(t, p.T);
        xs[i] = (t >= 0 && t < p.T) ? (float)xb[t] : 0.f;
    }
    for (int i = tid; i < 256; i += 256) {
        const int k = i >> 4, o = i & 15;
        ws[i] = (k < p.K && o < p.C) ? (float)p.w[o * p.K + k] : 0.f;
    }
    if (tid < 16) bs[tid] = (p.bias && tid < p.C) ? p.bias[tid] : 0.f;
    __syncthreads();
    const int t = t0 + tid;
    if (t >= p.T) return;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = bs[o];
    for (int k = 0; k < p.K; ++k) {
        const float xv = xs[tid + k];
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
            const f32x4 w4 = *(const f32x4*)&ws[k * 16 + o4 * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[o4 * 4 + i] = fmaf(xv, w4[i], acc[o4 * 4 + i]);
        }
    }
    h16* yo = p.out + ((int64_t)b * p.T + t) * p.C;
    for (int o8 = 0; o8 < p.C; o8 += 8) {
        h16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float a = acc[o8 + i]; v[i] = (h16)(a > 0.f ? a : a * p.slope); }
        *(h16x8*)(yo + o8) = v;
    }
}

// dx [B][T]: every padded position that aliases step t (t + pad, and the mirror images under ReflectionPad1d) collects
// sum_{k, o} dy'[s - k][o] w[o][k]
__global__ __launch_bounds__(256) void aero_gconv_c1_dgrad_kernel(AeroEdgeK p) {
    __shared__ AERO_LDS_ALIGN float ws[16 * 16];                 // [k][o]
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int i = tid; i < 256; i += 256) {
        const int k = i >> 4, o = i & 15;
        ws[i] = (k < p.K && o < p.C) ? (float)p.w[o * p.K + k] : 0.f;
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + tid;
    if (t >= p.T) return;
    const h16* dyb = p.dy + (int64_t)b * p.T * p.C;
    const h16* yb = p.y + (int64_t)b * p.T * p.C;
    int pp[3], npp = 0;
    pp[npp++] = t + p.pad;
    if (p.reflect) {
        if (t >= 1 && t <= p.pad) pp[npp++] = p.pad - t;
        if (t <= p.T - 2 && t >= p.T - 1 - p.pad) pp[npp++] = p.pad + 2 * (p.T - 1) - t;
    }
    float acc = 0.f;
    for (int qi = 0; qi < npp; ++qi) {
        for (int k = 0; k < p.K; ++k) {
            const int to = pp[qi] - k;
            if (to < 0 || to >= p.T) continue;
            for (int o8 = 0; o8 < p.C; o8 += 8) {
                const h16x8 dv = *(const h16x8*)(dyb + (int64_t)to * p.C + o8);
                const h16x8 yv = *(const h16x8*)(yb + (int64_t)to * p.C + o8);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc = fmaf((float)dv[i] * ((float)yv[i] > 0.f ? 1.f : p.slope), ws[k * 16 + o8 + i], acc);
            }
        }
    }
    p.out[(int64_t)b * p.T + t] = (h16)acc;
}

// dw[o][k] = sum dy'[to][o] xpad[to + k], db[o] = sum dy'[to][o]: thread (o, k) of 16 x 16; a block walks `tiles_per_block` tiles of 256 steps
__global__ __launch_bounds__(256) void aero_gconv_c1_wgrad_kernel(AeroEdgeK p) {
    __shared__ float xs[256 + 16];
    __shared__ float ds[256 * 17];                               // [step][o], odd stride
    const int tid = threadIdx.x, b = blockIdx.y;
    const int o = tid & 15, k = tid >> 4;
    const h16* xb = p.x + (int64_t)b * p.T;
    const h16* dyb = p.dy + (int64_t)b * p.T * p.C;
    const h16* yb = p.y + (int64_t)b * p.T * p.C;
    float acc = 0.f, bacc = 0.f;
    const int tile0 = blockIdx.x * p.tiles_per_block;
    const int tile1 = tile0 + p.tiles_per_block < p.ntile ? tile0 + p.tiles_per_block : p.ntile;
    for (int tile = tile0; tile < tile1; ++tile) {
        const int t0 = tile * 256;
        __syncthreads();
        for (int i = tid; i < 256 + p.K - 1; i += 256) {
            int t = t0 - p.pad + i;
            if (p.reflect) t = aero_reflect(t, p.T);
            xs[i] = (t >= 0 && t < p.T) ? (float)xb[t] : 0.f;
        }
        for (int i = tid; i < 256 * 16; i += 256) {
            const int s = i >> 4, oo = i & 15;
            float d = 0.f;
            if (t0 + s < p.T && oo < p.C) {
                const float yv = (float)yb[(int64_t)(t0 + s) * p.C + oo];
                d = (float)dyb[(int64_t)(t0 + s) * p.C + oo] * (yv > 0.f ? 1.f : p.slope);
            }
            ds[s * 17 + oo] = d;
        }
        __syncthreads();
        if (k < p.K) {
#pragma unroll 8
            for (int s = 0; s < 256; ++s) acc = fmaf(ds[s * 17 + o], xs[s + k], acc);
        } else if (k == 15) {
#pragma unroll 8
            for (int s = 0; s < 256; ++s) bacc += ds[s * 17 + o];
        }
    }
    float* sl = p.slabs + ((int64_t)b * gridDim.x + blockIdx.x) * p.sl_stride;
    if (k < p.K && o < p.C) sl[o * p.K + k] = acc;
    if (k == 15 && o < p.C) sl[p.C * p.K + o] = bacc;            // (K <= 15: row 15 of the thread grid is free for the bias sums)
}

// ---- one output channel: x [B][T][C], w fp16 [1][K][C], y [B][T]; stride 1, zero padding, C a multiple of 512; one wave per step
__global__ __launch_bounds__(256) void aero_gconv_o1_fwd_kernel(AeroEdgeK p) {
    const int lane = aero_lane(), wave = aero_uniform(aero_wave());
    const int b = blockIdx.y;
    const int t = blockIdx.x * 4 + wave;
    if (t >= p.T) return;
    const h16* xb = p.x + (int64_t)b * p.T * p.C;
    float acc = 0.f;
    for (int k = 0; k < p.K; ++k) {
        const int ti = t - p.pad + k;
        if (ti < 0 || ti >= p.T) continue;
        for (int c = lane * 8; c < p.C; c += 512) {
            const h16x8 xv = *(const h16x8*)(xb + (int64_t)ti * p.C + c);
            const h16x8 wv = *(const h16x8*)(p.w + (int64_t)k * p.C + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = fmaf((float)xv[i], (float)wv[i], acc);
        }
    }
    acc = aero_wave_sum(acc);
    if (lane == 0) {
        const float v = acc + (p.bias ? p.bias[0] : 0.f);
        p.out[(int64_t)b * p.T + t] = (h16)(v > 0.f ? v : v * p.slope);
    }
}

// dx[t][c] = sum_k dy'[t + pad - k] w[k][c]: thread = (step, 8 channels)
__global__ __launch_bounds__(256) void aero_gconv_o1_dgrad_kernel(AeroEdgeK p) {
    const int b = blockIdx.y;
    const int per = p.C / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.T * per) return;
    const int t = (int)(idx / per), c = (int)(idx - (int64_t)t * per) * 8;
    const h16* dyb = p.dy + (int64_t)b * p.T;
    const h16* yb = p.y + (int64_t)b * p.T;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = 0; k < p.K; ++k) {
        const int to = t + p.pad - k;
        if (to < 0 || to >= p.T) continue;
        const float d = (float)dyb[to] * ((float)yb[to] > 0.f ? 1.f : p.slope);
        const h16x8 wv = *(const h16x8*)(p.w + (int64_t)k * p.C + c);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(d, (float)wv[i], acc[i]);
    }
    h16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (h16)acc[i];
    *(h16x8*)(p.out + ((int64_t)b * p.T + t) * p.C + c) = o;
}

// dw[k][c] = sum_t dy'[t] x[t - pad + k][c], db = sum dy': a block walks a range of steps; thread = (8 channels, step parity class)
__global__ __launch_bounds__(256) void aero_gconv_o1_wgrad_kernel(AeroEdgeK p) {
    __shared__ float red[256];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int per = p.C / 8;                                     // channel octets; 256 threads = (256 / per) step lanes x per octets when per <= 256
    const int oct = tid % per, lanes = 256 / per, sl_lane = tid / per;
    const h16* xb = p.x + (int64_t)b * p.T * p.C;
    const h16* dyb = p.dy + (int64_t)b * p.T;
    const h16* yb = p.y + (int64_t)b * p.T;
    const int s0 = blockIdx.x * p.tiles_per_block;               // (steps per block)
    const int s1 = s0 + p.tiles_per_block < p.T ? s0 + p.tiles_per_block : p.T;
    float acc[3][8];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[k][i] = 0.f;
    float bacc = 0.f;
    if (sl_lane < lanes) {
        // input row ti feeds dw[k] with dy'[ti + pad - k]
        for (int ti = s0 - (p.K - 1) + sl_lane; ti < s1 + p.K - 1; ti += lanes) {
            if (ti < 0 || ti >= p.T) continue;
            const h16x8 xv = *(const h16x8*)(xb + (int64_t)ti * p.C + oct * 8);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int to = ti + p.pad - k;
                if (k >= p.K || to < s0 || to >= s1) continue;
                const float d = (float)dyb[to] * ((float)yb[to] > 0.f ? 1.f : p.slope);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[k][i] = fmaf(d, (float)xv[i], acc[k][i]);
            }
        }
        if (oct == 0)
            for (int to = s0 + sl_lane; to < s1; to += lanes) bacc += (float)dyb[to] * ((float)yb[to] > 0.f ? 1.f : p.slope);
    }
    float* sl = p.slabs + ((int64_t)b * gridDim.x + blockIdx.x) * p.sl_stride;
    // the step lanes of an octet are added in lane order through LDS
    for (int k = 0; k < p.K; ++k)
        for (int i = 0; i < 8; ++i) {
            __syncthreads();
            red[tid] = acc[k][i];
            __syncthreads();
            if (sl_lane == 0) {
                float s = 0.f;
                for (int l = 0; l < lanes; ++l) s += red[l * per + oct];
                sl[(int64_t)k * p.C + oct * 8 + i] = s;
            }
        }
    __syncthreads();
    red[tid] = bacc;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[l * per];
        sl[(int64_t)p.K * p.C] = s;
        sl[(int64_t)p.K * p.C + 1] = sl[(int64_t)p.K * p.C + 2] = sl[(int64_t)p.K * p.C + 3] = 0.f;     // (slab rows are whole float4s)
    }
}

static int aero_edge_c1_ok(int Cin, int Cout, int groups, int K, int stride, int pad) {
    return Cin == 1 && groups == 1 && stride == 1 && K <= 15 && K == 2 * pad + 1 && (Cout == 8 || Cout == 16);
}
static int aero_edge_o1_ok(int Cin, int Cout, int groups, int K, int stride, int pad, int reflect) {
    return Cout == 1 && groups == 1 && stride == 1 && K <= 3 && K == 2 * pad + 1 && !reflect && Cin % 512 == 0 && Cin <= 2048;
}

// blocks per batch item and steps (o1) / 256-step tiles (c1) per block of the edge-layer weight gradients
static void aero_edge_wgrad_plan(bool c1, int B, int T, int* nb, int* per) {
    long want = (c1 ? 1024 : 512) / (long)B;
    if (want < 1) want = 1;
    const long units = c1 ? (T + 255) / 256 : (T + 31) / 32;     // c1: tiles; o1: at least 32 steps per block
    if (want > units) want = units;
    const long total = c1 ? (T + 255) / 256 : T;
    *per = (int)((total + want - 1) / want);
    *nb = (int)((total + *per - 1) / *per);
}

// Weight-norm chain rule of one convolution (torch.nn.utils.weight_norm, dim 0: w[o] = g[o] v[o] / |v[o]|), with the un-scaling of
// the fp16 gradient path and the upstream loss factor folded in -- one launch instead of ~20 parameter-sized torch kernels per conv:
//   dg[o] (+)= a * <dw[o], v[o]> / |v[o]|,   dv[o] (+)= a * g[o] / |v[o]| * (dw[o] - v[o] <dw[o], v[o]> / |v[o]|^2),   dbias[o] (+)= a * db[o]
// a = inv_scale[0] * (gl ? gl[0] : 1).  dw is addressed by strides (element (o, c, k) at o*so + c*sc + k*sk): the kernels leave it as
// [Cout][K][cig] (grouped / edge layers) or [K][Cout][Cin] (aero_conv_wgrad); v, dv are the parameter's [Cout][cig][K].
struct AeroWnBwdK {
    const float* dw; const float* v; const float* g; const float* db; const float* inv_scale; const float* gl;
    float* dg; float* dv; float* dbias;
    int64_t so, sc, sk;
    int Cout, cig, K, accumulate;
};

__global__ __launch_bounds__(256) void aero_weightnorm_bwd_kernel(AeroWnBwdK p) {
    __shared__ float red[2][4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const int L = p.cig * p.K;
    const float* vr = p.v + (int64_t)o * L;
    float nv2 = 0.f, dot = 0.f;
    for (int i = tid; i < L; i += 256) {
        const int c = i / p.K, k = i - c * p.K;
        const float vv = vr[i], d = p.dw[(int64_t)o * p.so + (int64_t)c * p.sc + (int64_t)k * p.sk];
        nv2 = fmaf(vv, vv, nv2);
        dot = fmaf(d, vv, dot);
    }
    nv2 = aero_wave_sum(nv2);
    dot = aero_wave_sum(dot);
    if (aero_lane() == 0) { red[0][aero_wave()] = nv2; red[1][aero_wave()] = dot; }
    __syncthreads();
    nv2 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    dot = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float a = p.inv_scale[0] * (p.gl ? p.gl[0] : 1.f);
    const float inv = 1.f / sqrtf(nv2), gg = p.g[o];
    const float cg = a * gg * inv, cv = dot / nv2;
    for (int i = tid; i < L; i += 256) {
        const int c = i / p.K, k = i - c * p.K;
        const float d = p.dw[(int64_t)o * p.so + (int64_t)c * p.sc + (int64_t)k * p.sk];
        const float r = cg * (d - vr[i] * cv);
        float* dst = p.dv + (int64_t)o * L + i;
        *dst = p.accumulate ? *dst + r : r;
    }
    if (tid == 0) {
        const float rg = a * dot * inv;
        p.dg[o] = p.accumulate ? p.dg[o] + rg : rg;
        if (p.db && p.dbias) {
            const float rb = a * p.db[o];
            p.dbias[o] = p.accumulate ? p.dbias[o] + rb : rb;
        }
    }
}

static int aero_weightnorm_bwd_launch(const float* dw, int64_t so, int64_t sc, int64_t sk, const float* v, const float* g, const float* db,
                                      const float* inv_scale, const float* gl, float* dg, float* dv, float* dbias, int Cout, int cig, int K,
                                      int accumulate, hipStream_t stream, const char** err) {
    if (!dw || !v || !g || !inv_scale || !dg || !dv || Cout < 1 || cig < 1 || K < 1) { *err = "weightnorm_bwd: bad arguments"; return AERO_ERR_ARG; }
    AeroWnBwdK p;
    p.dw = dw; p.v = v; p.g = g; p.db = db; p.inv_scale = inv_scale; p.gl = gl; p.dg = dg; p.dv = dv; p.dbias = dbias;
    p.so = so; p.sc = sc; p.sk = sk; p.Cout = Cout; p.cig = cig; p.K = K; p.accumulate = accumulate;
    AERO_LAUNCH(aero_weightnorm_bwd_kernel, dim3((unsigned)Cout), dim3(256), stream, p);
    return AERO_OK;
}

// w[o][:] = g[o] * v[o][:] / |v[o]|  (fp32 rows of L elements; one block per output channel): the forward of torch's weight_norm for one conv
__global__ __launch_bounds__(256) void aero_weightnorm_fwd_kernel(const float* v, const float* g, float* w, int L) {
    __shared__ float red[4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const float* vr = v + (int64_t)o * L;
    float nv2 = 0.f;
    for (int i = tid; i < L; i += 256) nv2 = fmaf(vr[i], vr[i], nv2);
    nv2 = aero_wave_sum(nv2);
    if (aero_lane() == 0) red[aero_wave()] = nv2;
    __syncthreads();
    nv2 = red[0] + red[1] + red[2] + red[3];
    const float sc = g[o] / sqrtf(nv2);
    for (int i = tid; i < L; i += 256) w[(int64_t)o * L + i] = vr[i] * sc;
}
