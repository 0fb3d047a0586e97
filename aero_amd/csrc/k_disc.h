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

__global__ void aero_loss_sum_finish_kernel(const double* part, int nblk, double* out) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += part[k];
        out[0] += s;
    }
}

static int aero_loss_sum_launch(const void* a, const void* b, int64_t n, float sign, int mode, double* part, int npart, double* out, hipStream_t stream,
                                const char** err) {
    if (!a || (mode == 1 && !b) || !part || !out || n < 1 || npart < 1 || mode < 0 || mode > 1) { *err = "loss_sum: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > npart) nb = npart;
    AERO_LAUNCH(aero_loss_sum_kernel, dim3((unsigned)nb), dim3(256), stream, (const h16*)a, (const h16*)b, n, sign, mode, part);
    AERO_LAUNCH(aero_loss_sum_finish_kernel, dim3(1), dim3(64), stream, (const double*)part, (int)nb, out);
    return AERO_OK;
}
