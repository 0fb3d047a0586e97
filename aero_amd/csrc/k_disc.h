// k_disc.h -- MelGAN multi-scale discriminator (reference src/models/discriminators.py:14-78: NLayerDiscriminator / Discriminator),
// the critic of the adversarial half of a training step (solver.py:475-520), SURVEY.md 8 f3.
//
//   aero_gconv1d_fwd    grouped, strided Conv1d over time on channels-last rows [B][T][C] (fp16), bias + LeakyReLU fused
//                       (the k = 41 / stride 4 / groups C/4 layers, the k = 15 input layer behind its ReflectionPad1d(7), the
//                       1024 -> 1 output layer); weights fp16 [Cout][K][Cin/groups] with weight-norm already applied
//   aero_leaky_relu     in place, for the dense k = 5 layer that runs on aero_conv_fwd
//   aero_avgpool1d      AvgPool1d(4, stride 2, padding 1, count_include_pad=False) between the scales (discriminators.py:70)
//   aero_hinge_sum / aero_l1_sum     the reductions of the hinge and feature-matching losses (solver.py:489-512)
//
// A grouped conv with 4 input channels per group is 164 MACs per output: VALU work, bound by LDS reads of the input span.  A block
// owns 64 output steps of one group chunk; the input span and the group's weights sit in LDS as fp32.
#pragma once
#include "aero_common.h"
#include "k_gconv_mfma.h"
#include "k_gconv_edge.h"

struct AeroGconvK {
    const h16* x; const h16* w; const float* bias; h16* y;
    int B, Tin, Tout, Cin, Cout, groups, K, stride, pad, reflect;
    float slope;
    int cig, cog, cob, ncb;                                     // channels per group (in / out), out-channel chunk per block, chunks per group
};

#define AERO_GCONV_TO 64
#define AERO_GCONV_CIC 16                                       /* input channels staged per pass */

__global__ __launch_bounds__(256) void aero_gconv1d_kernel(AeroGconvK p) {
    float* xs = (float*)AERO_DYN_SMEM;                           // [span][cc]
    const int span = (AERO_GCONV_TO - 1) * p.stride + p.K;
    float* ws = xs + span * AERO_GCONV_CIC;                      // [cob][K][cc]
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * AERO_GCONV_TO;
    const int g = blockIdx.y / p.ncb, cb = blockIdx.y % p.ncb;
    const int b = blockIdx.z;
    const int co0 = g * p.cog + cb * p.cob;                      // first output channel of this block
    const int nco = (p.cog - cb * p.cob) < p.cob ? (p.cog - cb * p.cob) : p.cob;
    const int pos = tid & 63, cl = tid >> 6;                     // thread: output step pos, output channels cl, cl + 4, ...
    constexpr int MAXJ = 16;                                     // cob <= 64
    float acc[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) acc[j] = 0.f;
    const h16* xb = p.x + (int64_t)b * p.Tin * p.Cin + g * p.cig;
    for (int c0 = 0; c0 < p.cig; c0 += AERO_GCONV_CIC) {
        const int cc = (p.cig - c0) < AERO_GCONV_CIC ? (p.cig - c0) : AERO_GCONV_CIC;
        __syncthreads();
        for (int idx = tid; idx < span * cc; idx += 256) {
            const int s = idx / cc, c = idx - s * cc;
            int t = t0 * p.stride - p.pad + s;
            if (p.reflect) {
                if (t < 0) t = -t;
                if (t >= p.Tin) t = 2 * (p.Tin - 1) - t;
            }
            xs[s * AERO_GCONV_CIC + c] = (t >= 0 && t < p.Tin) ? (float)xb[(int64_t)t * p.Cin + c0 + c] : 0.f;
        }
        for (int idx = tid; idx < nco * p.K * cc; idx += 256) {
            const int c = idx % cc, k = (idx / cc) % p.K, o = idx / (cc * p.K);
            ws[(o * p.K + k) * AERO_GCONV_CIC + c] = (float)p.w[((int64_t)(co0 + o) * p.K + k) * p.cig + c0 + c];
        }
        __syncthreads();
        for (int k = 0; k < p.K; ++k) {
            const float* xr = xs + (pos * p.stride + k) * AERO_GCONV_CIC;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int o = cl + 4 * j;
                if (o < nco) {
                    const float* wr = ws + (o * p.K + k) * AERO_GCONV_CIC;
                    float s = 0.f;
                    for (int c = 0; c < cc; ++c) s += xr[c] * wr[c];
                    acc[j] += s;
                }
            }
        }
    }
    const int t = t0 + pos;
    if (t >= p.Tout) return;
    h16* yo = p.y + ((int64_t)b * p.Tout + t) * p.Cout + co0;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int o = cl + 4 * j;
        if (o < nco) {
            float v = acc[j] + (p.bias ? p.bias[co0 + o] : 0.f);
            v = v > 0.f ? v : v * p.slope;
            yo[o] = (h16)v;
        }
    }
}

static int aero_gconv1d_launch(const aero_gconv_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->x || !d->w || !d->y) { *err = "gconv1d: null pointer"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->Tin < 1 || d->Cin < 1 || d->Cout < 1 || d->groups < 1 || d->Cin % d->groups || d->Cout % d->groups || d->K < 1 || d->stride < 1 ||
        d->pad < 0 || (d->reflect && d->pad >= d->Tin)) { *err = "gconv1d: bad geometry"; return AERO_ERR_ARG; }
    AeroGconvK p;
    p.x = (const h16*)d->x; p.w = (const h16*)d->w; p.bias = d->bias; p.y = (h16*)d->y;
    p.B = d->B; p.Tin = d->Tin; p.Cin = d->Cin; p.Cout = d->Cout; p.groups = d->groups; p.K = d->K; p.stride = d->stride; p.pad = d->pad;
    p.reflect = d->reflect; p.slope = d->slope;
    p.Tout = (d->Tin + 2 * d->pad - d->K) / d->stride + 1;
    if (p.Tout < 1) { *err = "gconv1d: kernel longer than the padded input"; return AERO_ERR_ARG; }
    if (aero_edge_c1_ok(d->Cin, d->Cout, d->groups, d->K, d->stride, d->pad) || aero_edge_o1_ok(d->Cin, d->Cout, d->groups, d->K, d->stride, d->pad, d->reflect)) {
        if (((uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->w) & 15) { *err = "gconv1d: 16-byte aligned tensors required"; return AERO_ERR_ARG; }
        if (d->B > 65535) { *err = "gconv1d: grid too large"; return AERO_ERR_ARG; }
        AeroEdgeK e = {};
        e.x = (const h16*)d->x; e.w = (const h16*)d->w; e.bias = d->bias; e.out = (h16*)d->y;
        e.B = d->B; e.T = d->Tin; e.K = d->K; e.pad = d->pad; e.reflect = d->reflect; e.slope = d->slope;
        if (d->Cin == 1) {
            e.C = d->Cout;
            AERO_LAUNCH(aero_gconv_c1_fwd_kernel, dim3((unsigned)((d->Tin + 255) / 256), (unsigned)d->B), dim3(256), stream, e);
        } else {
            e.C = d->Cin;
            AERO_LAUNCH(aero_gconv_o1_fwd_kernel, dim3((unsigned)((d->Tin + 3) / 4), (unsigned)d->B), dim3(256), stream, e);
        }
        return AERO_OK;
    }
    if (d->w_mfma && aero_gconv4_ok(d->Cin, d->Cout, d->groups, d->K, d->stride, d->pad, d->reflect) && d->B <= 65535) {
        if (((uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->w_mfma | (uintptr_t)d->bias) & 15) { *err = "gconv1d: 16-byte aligned tensors required"; return AERO_ERR_ARG; }
        AeroGconv4K m;
        m.src = (const h16*)d->x; m.act = nullptr; m.wimg = (const h16*)d->w_mfma; m.bias = d->bias; m.dst = (h16*)d->y;
        m.B = d->B; m.Ts = d->Tin; m.Cs = d->Cin; m.Td = p.Tout; m.Cd = d->Cout; m.groups = d->groups; m.pad = d->pad; m.slope = d->slope;
        aero_gconv4_tile(d->groups, &m.GPB, &m.NT);
        m.ROWS = m.NT * 4 + 48;
        const size_t lds = (size_t)m.GPB * m.ROWS * 4 * sizeof(h16);
        dim3 grid((unsigned)((p.Tout + m.NT - 1) / m.NT), (unsigned)(d->groups / m.GPB), (unsigned)d->B);
        if (d->Cout / d->groups == 16) AERO_LAUNCH_DYN(aero_gconv4_fwd_kernel<16>, grid, dim3(256), lds, stream, m);
        else AERO_LAUNCH_DYN(aero_gconv4_fwd_kernel<4>, grid, dim3(256), lds, stream, m);
        return AERO_OK;
    }
    p.cig = d->Cin / d->groups;
    p.cog = d->Cout / d->groups;
    p.cob = p.cog < 64 ? p.cog : 64;
    p.ncb = (p.cog + p.cob - 1) / p.cob;
    const int span = (AERO_GCONV_TO - 1) * p.stride + p.K;
    const size_t lds = ((size_t)span + (size_t)p.cob * p.K) * AERO_GCONV_CIC * sizeof(float);
    if (lds > 150 * 1024) { *err = "gconv1d: tile exceeds the LDS (kernel / stride too large)"; return AERO_ERR_UNSUPPORTED; }
    const long gy = (long)d->groups * p.ncb;
    if (gy > 65535 || d->B > 65535) { *err = "gconv1d: grid too large"; return AERO_ERR_ARG; }
    dim3 grid((unsigned)((p.Tout + AERO_GCONV_TO - 1) / AERO_GCONV_TO), (unsigned)gy, (unsigned)d->B);
    AERO_LAUNCH_DYN(aero_gconv1d_kernel, grid, dim3(256), lds, stream, p);
    return AERO_OK;
}

__global__ __launch_bounds__(256) void aero_leaky_relu_kernel(h16* x, int64_t n, float slope) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = (float)x[i];
        x[i] = (h16)(v > 0.f ? v : v * slope);
    }
}

static int aero_leaky_relu_launch(void* x, int64_t n, float slope, hipStream_t stream, const char** err) {
    if (!x || n < 1) { *err = "leaky_relu: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    AERO_LAUNCH(aero_leaky_relu_kernel, dim3((unsigned)nb), dim3(256), stream, (h16*)x, n, slope);
    return AERO_OK;
}

// AvgPool1d(kernel 4, stride 2, padding 1, count_include_pad=False): y[t] = mean of x[2t-1 .. 2t+2] over the samples inside [0, T)
__global__ __launch_bounds__(256) void aero_avgpool1d_kernel(const h16* x, h16* y, int T, int To) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= To) return;
    const h16* xs = x + (int64_t)b * T;
    float s = 0.f;
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        const int i = 2 * t - 1 + k;
        if (i >= 0 && i < T) { s += (float)xs[i]; ++n; }
    }
    y[(int64_t)b * To + t] = (h16)(s / (float)n);
}

static int aero_avgpool1d_launch(const void* x, void* y, int B, int T, hipStream_t stream, const char** err) {
    if (!x || !y || B < 1 || T < 2 || B > 65535) { *err = "avgpool1d: bad arguments"; return AERO_ERR_ARG; }
    const int To = (T + 2 - 4) / 2 + 1;
    AERO_LAUNCH(aero_avgpool1d_kernel, dim3((unsigned)((To + 255) / 256), (unsigned)B), dim3(256), stream, (const h16*)x, (h16*)y, T, To);
    return AERO_OK;
}

// out[0] += sum relu(1 + sign * x[i])      (hinge terms of solver.py:489-496,508-509; sign = +1 / -1)
// out[0] += sum |a[i] - b[i]|               (feature matching, solver.py:505; b may alias nothing)
__global__ __launch_bounds__(256) void aero_loss_sum_kernel(const h16* a, const h16* b, int64_t n, float sign, int mode, double* part) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (mode == 0) s += (double)fmaxf(0.f, 1.f + sign * (float)a[i]);
        else s += (double)fabsf((float)a[i] - (float)b[i]);
    }
    s = aero_wave_sum(s);
    if (aero_lane() == 0) red[aero_wave()] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void aero_loss_sum_finish_kernel(const double* part, int nblk, double* out, double weight) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += part[k];
        out[0] += weight * s;
    }
}

static int aero_loss_sum_launch(const void* a, const void* b, int64_t n, float sign, int mode, double* part, int npart, double* out, double weight,
                                hipStream_t stream, const char** err) {
    if (!a || (mode == 1 && !b) || !part || !out || n < 1 || npart < 1 || mode < 0 || mode > 1) { *err = "loss_sum: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > npart) nb = npart;
    AERO_LAUNCH(aero_loss_sum_kernel, dim3((unsigned)nb), dim3(256), stream, (const h16*)a, (const h16*)b, n, sign, mode, part);
    AERO_LAUNCH(aero_loss_sum_finish_kernel, dim3(1), dim3(64), stream, (const double*)part, (int)nb, out, weight);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the critic (solver.py:602-611 through discriminators.py:14-78).  dy arrives as the gradient of the layer's
// POST-activation output y; LeakyReLU is sign preserving, so its derivative is read off y: dyp = dy * (y > 0 ? 1 : slope).
//
//   aero_gconv1d_dgrad   dx[b][t][g*cig + c] = sum_{o in group, k : (t' + pad - k) % stride == 0} dyp[b][(t' + pad - k)/stride][o] w[o][k][c]
//                        summed over the padded positions t' that alias t (ReflectionPad1d: up to three)
//   aero_gconv1d_wgrad   dw[o][k][c] += sum_{b, to} dyp[b][to][o] x[b][to*stride - pad + k][g*cig + c],   db[o] += sum dyp
//   aero_loss_grad       the hinge / L1 loss gradients as fp16 with a power-of-two scale
//   aero_avgpool1d_bwd   adjoint of aero_avgpool1d
struct AeroGconvBwdK {
    const h16* x; const h16* w; const h16* y; const h16* dy; h16* dx; float* dw; float* db;
    int B, Tin, Tout, Cin, Cout, groups, K, stride, pad, reflect;
    float slope;
    int cig, cog, tiles_per_block, ntile;
};

__global__ __launch_bounds__(256) void aero_gconv1d_dgrad_kernel(AeroGconvBwdK p) {
    // block: 256 input steps of one (batch item, group); thread = one step, loops over the group's input channels in chunks
    float* ws = (float*)AERO_DYN_SMEM;                           // [cog][K][cc]
    const int g = blockIdx.y, b = blockIdx.z;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const h16* dyb = p.dy + (int64_t)b * p.Tout * p.Cout + g * p.cog;
    const h16* yb = p.y + (int64_t)b * p.Tout * p.Cout + g * p.cog;
    for (int c0 = 0; c0 < p.cig; c0 += AERO_GCONV_CIC) {
        const int cc = (p.cig - c0) < AERO_GCONV_CIC ? (p.cig - c0) : AERO_GCONV_CIC;
        __syncthreads();
        for (int idx = threadIdx.x; idx < p.cog * p.K * cc; idx += 256) {
            const int c = idx % cc, k = (idx / cc) % p.K, o = idx / (cc * p.K);
            ws[(o * p.K + k) * AERO_GCONV_CIC + c] = (float)p.w[((int64_t)(g * p.cog + o) * p.K + k) * p.cig + c0 + c];
        }
        __syncthreads();
        if (t >= p.Tin) continue;
        float acc[AERO_GCONV_CIC];
#pragma unroll
        for (int c = 0; c < AERO_GCONV_CIC; ++c) acc[c] = 0.f;
        // padded positions that read x[t]: t + pad, and with reflection the mirror images
        int pp[3];
        int npp = 0;
        pp[npp++] = t + p.pad;
        if (p.reflect) {
            if (t >= 1 && t <= p.pad) pp[npp++] = p.pad - t;
            if (t <= p.Tin - 2 && t >= p.Tin - 1 - p.pad) pp[npp++] = p.pad + 2 * (p.Tin - 1) - t;
        }
        for (int q = 0; q < npp; ++q) {
            for (int k = 0; k < p.K; ++k) {
                const int num = pp[q] - k;
                if (num < 0 || num % p.stride) continue;
                const int to = num / p.stride;
                if (to >= p.Tout) continue;
                for (int o = 0; o < p.cog; ++o) {
                    const float yv = (float)yb[(int64_t)to * p.Cout + o];
                    const float d = (float)dyb[(int64_t)to * p.Cout + o] * (yv > 0.f ? 1.f : p.slope);
                    const float* wr = ws + (o * p.K + k) * AERO_GCONV_CIC;
                    for (int c = 0; c < cc; ++c) acc[c] += d * wr[c];
                }
            }
        }
        h16* dxo = p.dx + ((int64_t)b * p.Tin + t) * p.Cin + g * p.cig + c0;
        for (int c = 0; c < cc; ++c) dxo[c] = (h16)acc[c];
    }
}

__global__ __launch_bounds__(256) void aero_gconv1d_wgrad_kernel(AeroGconvBwdK p) {
    // block: (range of 64-step output tiles, group, batch item); a thread owns outputs (o, k, c) = idx, idx + 256, ... of the group chunk
    float* xs = (float*)AERO_DYN_SMEM;                           // [span][cc]
    const int span = (AERO_GCONV_TO - 1) * p.stride + p.K;
    float* ds = xs + span * AERO_GCONV_CIC;                      // [TO][cog]
    const int g = blockIdx.y, b = blockIdx.z;
    const int tile0 = blockIdx.x * p.tiles_per_block;
    int tile1 = tile0 + p.tiles_per_block;
    if (tile1 > p.ntile) tile1 = p.ntile;
    const h16* xb = p.x + (int64_t)b * p.Tin * p.Cin + g * p.cig;
    const h16* dyb = p.dy + (int64_t)b * p.Tout * p.Cout + g * p.cog;
    const h16* yb = p.y + (int64_t)b * p.Tout * p.Cout + g * p.cog;
    constexpr int MAXO = 12;                                     // cog * K * cc <= 12 * 256 outputs per pass
    for (int c0 = 0; c0 < p.cig; c0 += AERO_GCONV_CIC) {
        const int cc = (p.cig - c0) < AERO_GCONV_CIC ? (p.cig - c0) : AERO_GCONV_CIC;
        const int nout = p.cog * p.K * cc;
        for (int o0 = 0; o0 < nout; o0 += MAXO * 256) {
            float acc[MAXO];
#pragma unroll
            for (int j = 0; j < MAXO; ++j) acc[j] = 0.f;
            float dbacc = 0.f;
            for (int tile = tile0; tile < tile1; ++tile) {
                const int t0 = tile * AERO_GCONV_TO;
                __syncthreads();
                for (int idx = threadIdx.x; idx < span * cc; idx += 256) {
                    const int s = idx / cc, c = idx - s * cc;
                    int t = t0 * p.stride - p.pad + s;
                    if (p.reflect) {
                        if (t < 0) t = -t;
                        if (t >= p.Tin) t = 2 * (p.Tin - 1) - t;
                    }
                    xs[s * AERO_GCONV_CIC + c] = (t >= 0 && t < p.Tin) ? (float)xb[(int64_t)t * p.Cin + c0 + c] : 0.f;
                }
                for (int idx = threadIdx.x; idx < AERO_GCONV_TO * p.cog; idx += 256) {
                    const int s = idx / p.cog, o = idx - s * p.cog;
                    float d = 0.f;
                    if (t0 + s < p.Tout) {
                        const float yv = (float)yb[(int64_t)(t0 + s) * p.Cout + o];
                        d = (float)dyb[(int64_t)(t0 + s) * p.Cout + o] * (yv > 0.f ? 1.f : p.slope);
                    }
                    ds[s * p.cog + o] = d;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < MAXO; ++j) {
                    const int idx = o0 + threadIdx.x + j * 256;
                    if (idx < nout) {
                        const int c = idx % cc, k = (idx / cc) % p.K, o = idx / (cc * p.K);
                        float s = 0.f;
                        for (int q = 0; q < AERO_GCONV_TO; ++q) s += ds[q * p.cog + o] * xs[(q * p.stride + k) * AERO_GCONV_CIC + c];
                        acc[j] += s;
                    }
                }
                if (c0 == 0 && o0 == 0 && p.db && (int)threadIdx.x < p.cog) {
                    float s = 0.f;
                    for (int q = 0; q < AERO_GCONV_TO; ++q) s += ds[q * p.cog + threadIdx.x];
                    dbacc += s;
                }
            }
#pragma unroll
            for (int j = 0; j < MAXO; ++j) {
                const int idx = o0 + threadIdx.x + j * 256;
                if (idx < nout) {
                    const int c = idx % cc, k = (idx / cc) % p.K, o = idx / (cc * p.K);
                    atomicAdd(p.dw + ((int64_t)(g * p.cog + o) * p.K + k) * p.cig + c0 + c, acc[j]);
                }
            }
            if (c0 == 0 && o0 == 0 && p.db && (int)threadIdx.x < p.cog) atomicAdd(p.db + g * p.cog + threadIdx.x, dbacc);
        }
    }
}

static int aero_gconv1d_bwd_launch(const aero_gconv_bwd_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->w || !d->y || !d->dy || (!d->dx && !d->dw)) { *err = "gconv1d_bwd: null pointer"; return AERO_ERR_ARG; }
    if (d->dw && !d->x) { *err = "gconv1d_bwd: the weight gradient needs x"; return AERO_ERR_ARG; }
    if (d->B < 1 || d->Tin < 1 || d->Cin < 1 || d->Cout < 1 || d->groups < 1 || d->Cin % d->groups || d->Cout % d->groups || d->K < 1 || d->stride < 1 ||
        d->pad < 0 || d->B > 65535 || d->groups > 65535) { *err = "gconv1d_bwd: bad geometry"; return AERO_ERR_ARG; }
    AeroGconvBwdK p;
    p.x = (const h16*)d->x; p.w = (const h16*)d->w; p.y = (const h16*)d->y; p.dy = (const h16*)d->dy; p.dx = (h16*)d->dx; p.dw = d->dw; p.db = d->db;
    p.B = d->B; p.Tin = d->Tin; p.Cin = d->Cin; p.Cout = d->Cout; p.groups = d->groups; p.K = d->K; p.stride = d->stride; p.pad = d->pad;
    p.reflect = d->reflect; p.slope = d->slope;
    p.Tout = (d->Tin + 2 * d->pad - d->K) / d->stride + 1;
    p.cig = d->Cin / d->groups;
    p.cog = d->Cout / d->groups;
    if (p.cog > 64) { *err = "gconv1d_bwd: more than 64 output channels per group"; return AERO_ERR_UNSUPPORTED; }
    p.ntile = (p.Tout + AERO_GCONV_TO - 1) / AERO_GCONV_TO;
    p.tiles_per_block = 1;
    const bool edge_c1 = aero_edge_c1_ok(d->Cin, d->Cout, d->groups, d->K, d->stride, d->pad);
    const bool edge_o1 = aero_edge_o1_ok(d->Cin, d->Cout, d->groups, d->K, d->stride, d->pad, d->reflect);
    if ((edge_c1 || edge_o1) && (!d->dw || d->slabs)) {
        if (((uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->dy | (uintptr_t)d->dx | (uintptr_t)d->w | (uintptr_t)d->slabs | (uintptr_t)d->dw | (uintptr_t)d->db) & 15) {
            *err = "gconv1d_bwd: 16-byte aligned tensors required"; return AERO_ERR_ARG;
        }
        AeroEdgeK e = {};
        e.x = (const h16*)d->x; e.w = (const h16*)d->w; e.y = (const h16*)d->y; e.dy = (const h16*)d->dy; e.out = (h16*)d->dx; e.slabs = d->slabs;
        e.B = d->B; e.T = d->Tin; e.K = d->K; e.pad = d->pad; e.reflect = d->reflect; e.slope = d->slope;
        e.C = edge_c1 ? d->Cout : d->Cin;
        if (d->dx) {
            if (edge_c1) AERO_LAUNCH(aero_gconv_c1_dgrad_kernel, dim3((unsigned)((d->Tin + 255) / 256), (unsigned)d->B), dim3(256), stream, e);
            else AERO_LAUNCH(aero_gconv_o1_dgrad_kernel, dim3((unsigned)(((int64_t)d->Tin * (d->Cin / 8) + 255) / 256), (unsigned)d->B), dim3(256), stream, e);
        }
        if (d->dw) {
            int nb, per;
            aero_edge_wgrad_plan(edge_c1, d->B, d->Tin, &nb, &per);
            if ((long)d->B * nb > d->nslab) { *err = "gconv1d_bwd: slab workspace smaller than aero_gconv1d_wgrad_slabs()"; return AERO_ERR_ARG; }
            const int64_t w_n = (int64_t)d->Cout * d->K * (d->Cin / d->groups);
            e.sl_stride = w_n + (d->Cout < 4 ? 4 : d->Cout);
            e.tiles_per_block = per;
            e.ntile = (d->Tin + 255) / 256;
            if (edge_c1) AERO_LAUNCH(aero_gconv_c1_wgrad_kernel, dim3((unsigned)nb, (unsigned)d->B), dim3(256), stream, e);
            else AERO_LAUNCH(aero_gconv_o1_wgrad_kernel, dim3((unsigned)nb, (unsigned)d->B), dim3(256), stream, e);
            AeroWgradFinishK f;
            f.slabs = d->slabs; f.dw = d->dw; f.db = d->db;
            f.stride = e.sl_stride; f.w_n = w_n; f.n = d->db ? e.sl_stride : w_n;
            f.nslab = d->B * nb; f.store = 0; f.layout = 0; f.ntaps = 1; f.MC = 0; f.C = 0; f.rowlen = 0; f.coff = 0;
            AERO_LAUNCH(aero_wgrad_finish_kernel, dim3((unsigned)((f.n / 4 + 31) / 32)), dim3(256), stream, f);
        }
        return AERO_OK;
    }
    if (d->dx && d->w_dgrad_mfma && aero_gconv4_ok(d->Cin, d->Cout, d->groups, d->K, d->stride, d->pad, d->reflect)) {
        if (((uintptr_t)d->dx | (uintptr_t)d->dy | (uintptr_t)d->y | (uintptr_t)d->w_dgrad_mfma) & 15) { *err = "gconv1d_bwd: 16-byte aligned tensors required"; return AERO_ERR_ARG; }
        AeroGconv4K m;
        m.src = (const h16*)d->dy; m.act = (const h16*)d->y; m.wimg = (const h16*)d->w_dgrad_mfma; m.bias = nullptr; m.dst = (h16*)d->dx;
        m.B = d->B; m.Ts = p.Tout; m.Cs = d->Cout; m.Td = d->Tin; m.Cd = d->Cin; m.groups = d->groups; m.pad = d->pad; m.slope = d->slope;
        aero_gconv4_tile(d->groups, &m.GPB, &m.NT);
        const int cog = d->Cout / d->groups;
        m.ROWS = m.NT + (cog == 16 ? 12 : 16);
        const size_t lds = (size_t)m.GPB * m.ROWS * cog * sizeof(h16);
        const int umax = ((d->Tin - 1 + d->pad) >> 2) + 1;
        dim3 grid((unsigned)((umax + m.NT - 1) / m.NT), (unsigned)(d->groups / m.GPB), (unsigned)d->B);
        if (cog == 16) AERO_LAUNCH_DYN(aero_gconv4_dgrad_kernel<16>, grid, dim3(256), lds, stream, m);
        else AERO_LAUNCH_DYN(aero_gconv4_dgrad_kernel<4>, grid, dim3(256), lds, stream, m);
    } else if (d->dx) {
        const size_t lds = (size_t)p.cog * p.K * AERO_GCONV_CIC * sizeof(float);
        if (lds > 150 * 1024) { *err = "gconv1d_bwd: weights exceed the LDS"; return AERO_ERR_UNSUPPORTED; }
        AERO_LAUNCH_DYN(aero_gconv1d_dgrad_kernel, dim3((unsigned)((p.Tin + 255) / 256), (unsigned)d->groups, (unsigned)d->B), dim3(256), lds, stream, p);
    }
    if (d->dw && d->slabs && aero_gconv4_ok(d->Cin, d->Cout, d->groups, d->K, d->stride, d->pad, d->reflect)) {
        AeroGconv4WK m;
        m.GPB = AERO_GCONV4_WGRAD_GPB;
        aero_gconv4_wgrad_plan(d->B, p.Tout, d->groups, &m.ntile, &m.tiles_per_chunk, &m.nchunk);
        if ((long)d->B * m.nchunk > d->nslab) { *err = "gconv1d_bwd: slab workspace smaller than aero_gconv1d_wgrad_slabs()"; return AERO_ERR_ARG; }
        if (((uintptr_t)d->x | (uintptr_t)d->dy | (uintptr_t)d->y | (uintptr_t)d->slabs | (uintptr_t)d->dw | (uintptr_t)d->db) & 15) {
            *err = "gconv1d_bwd: 16-byte aligned tensors required"; return AERO_ERR_ARG;
        }
        const int cog = d->Cout / d->groups;
        m.x = (const h16*)d->x; m.dy = (const h16*)d->dy; m.y = (const h16*)d->y; m.slabs = d->slabs;
        m.B = d->B; m.Tin = d->Tin; m.Cin = d->Cin; m.Tout = p.Tout; m.Cout = d->Cout; m.groups = d->groups; m.K = d->K; m.pad = d->pad;
        m.slope = d->slope;
        m.DS = m.GPB * cog + 4;
        m.XG = 1360;                                            // skewed span of 4 * 64 + 44 rows x 4 channels
        m.w_n = (int64_t)d->Cout * d->K * 4;
        m.sl_stride = m.w_n + d->Cout;
        const size_t lds = ((size_t)64 * m.DS + (size_t)m.GPB * m.XG) * sizeof(h16);
        dim3 grid((unsigned)m.nchunk, (unsigned)(d->groups / m.GPB), (unsigned)d->B);
        if (cog == 16) AERO_LAUNCH_DYN((aero_gconv4_wgrad_kernel<16, 1>), grid, dim3(256), lds, stream, m);
        else AERO_LAUNCH_DYN((aero_gconv4_wgrad_kernel<4, 1>), grid, dim3(256), lds, stream, m);
        AeroWgradFinishK f;
        f.slabs = d->slabs; f.dw = d->dw; f.db = d->db;
        f.stride = m.sl_stride; f.w_n = m.w_n; f.n = d->db ? m.sl_stride : m.w_n;
        f.nslab = d->B * m.nchunk; f.store = 0; f.layout = 0; f.ntaps = 1; f.MC = 0; f.C = 0; f.rowlen = 0; f.coff = 0;
        AERO_LAUNCH(aero_wgrad_finish_kernel, dim3((unsigned)((f.n / 4 + 31) / 32)), dim3(256), stream, f);
    } else if (d->dw) {
        const int span = (AERO_GCONV_TO - 1) * p.stride + p.K;
        const size_t lds = ((size_t)span * AERO_GCONV_CIC + (size_t)AERO_GCONV_TO * p.cog) * sizeof(float);
        if (lds > 150 * 1024) { *err = "gconv1d_bwd: tile exceeds the LDS"; return AERO_ERR_UNSUPPORTED; }
        // ~512 blocks: each walks a range of output tiles with its partial sums in registers, then one atomic pass
        long want = 512 / ((long)d->groups * d->B);
        if (want < 1) want = 1;
        int nbx = (int)(want < p.ntile ? want : p.ntile);
        p.tiles_per_block = (p.ntile + nbx - 1) / nbx;
        nbx = (p.ntile + p.tiles_per_block - 1) / p.tiles_per_block;
        AERO_LAUNCH_DYN(aero_gconv1d_wgrad_kernel, dim3((unsigned)nbx, (unsigned)d->groups, (unsigned)d->B), dim3(256), lds, stream, p);
    }
    return AERO_OK;
}

// loss gradients (solver.py:489-512), written as fp16 multiplied by `scale` (a power of two chosen by the caller):
//   mode 0 (hinge)  g[i] = coef * sign * [1 + sign * a[i] > 0]                  d/da of coef * relu(1 + sign * a)
//   mode 1 (L1)     g[i] = coef * sgn(a[i] - b[i])                              d/da of coef * |a - b|
//   mode 2          g[i] = dy[i] * (y[i] > 0 ? 1 : slope)   (a = dy, b = y, coef = slope): LeakyReLU backward for the dense layer
__global__ __launch_bounds__(256) void aero_loss_grad_kernel(const h16* a, const h16* b, int64_t n, float sign, float coef, int mode, h16* g, const float* gl) {
    if (gl && mode < 2) coef *= gl[0];                           // the upstream factor of the loss, a device scalar (no host read)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v;
        if (mode == 0) v = (1.f + sign * (float)a[i] > 0.f) ? coef * sign : 0.f;
        else if (mode == 1) { const float d = (float)a[i] - (float)b[i]; v = d > 0.f ? coef : (d < 0.f ? -coef : 0.f); }
        else v = (float)a[i] * ((float)b[i] > 0.f ? 1.f : coef);
        g[i] = (h16)v;
    }
}

static int aero_loss_grad_launch(const void* a, const void* b, int64_t n, float sign, float coef, int mode, void* g, const float* gl, hipStream_t stream,
                                 const char** err) {
    if (!a || !g || n < 1 || mode < 0 || mode > 2 || (mode >= 1 && !b)) { *err = "loss_grad: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    AERO_LAUNCH(aero_loss_grad_kernel, dim3((unsigned)nb), dim3(256), stream, (const h16*)a, (const h16*)b, n, sign, coef, mode, (h16*)g, gl);
    return AERO_OK;
}

__global__ __launch_bounds__(256) void aero_avgpool1d_bwd_kernel(const h16* dy, h16* dx, int T, int To) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    // windows t with 2t - 1 <= i <= 2t + 2  <=>  (i - 2) / 2 <= t <= (i + 1) / 2
    float s = 0.f;
    for (int t = (i - 2 + 1) / 2 < 0 ? 0 : (i - 1) / 2; t <= (i + 1) / 2; ++t) {
        if (t < 0 || t >= To || 2 * t - 1 > i || 2 * t + 2 < i) continue;
        int lo = 2 * t - 1, hi = 2 * t + 2;
        if (lo < 0) lo = 0;
        if (hi > T - 1) hi = T - 1;
        s += (float)dy[(int64_t)b * To + t] / (float)(hi - lo + 1);
    }
    dx[(int64_t)b * T + i] = (h16)s;
}

static int aero_avgpool1d_bwd_launch(const void* dy, void* dx, int B, int T, hipStream_t stream, const char** err) {
    if (!dy || !dx || B < 1 || T < 2 || B > 65535) { *err = "avgpool1d_bwd: bad arguments"; return AERO_ERR_ARG; }
    const int To = (T + 2 - 4) / 2 + 1;
    AERO_LAUNCH(aero_avgpool1d_bwd_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)B), dim3(256), stream, (const h16*)dy, (h16*)dx, T, To);
    return AERO_OK;
}
