// k_attn.h -- LocalState attention core (reference modules.py:101-124), flash-style: the T x T
// score matrix is never materialised.  One thread owns one query s of one (row, head); keys/values
// stream through LDS in tiles of 128 with an online softmax.  The learned distance-decay bias
// (modules.py:112-117) collapses analytically: sum_f -(f+1)|t-s|/sqrt(nd) * sigmoid(d_fs)/2
//   = -|t-s| * D_s,  D_s = sum_f (f+1) sigmoid(d_fs) / (2 sqrt(nd)),
// one scalar per query.  v1 runs the dot products on the vector ALU in fp32 (head dims are 12/24;
// the op is ~1.4 % of the model's FLOPs); DESIGN.md lists the MFMA version as follow-up.
#pragma once
#include "aero_common.h"

template <int DH>
__global__ __launch_bounds__(256) void aero_attn_kernel(aero_attn_desc d) {
    constexpr int KT = 128;
    __shared__ AERO_LDS_ALIGN float Ks[KT * DH];
    __shared__ AERO_LDS_ALIGN float Vs[KT * DH];
    const int tid = threadIdx.x;
    const int s = blockIdx.x * 256 + tid;
    const int h = blockIdx.y;
    const int row = blockIdx.z;
    const int C = d.C, T = d.T;
    const h16* base = (const h16*)d.qkvd + (int64_t)row * T * d.ld;
    const bool live = s < T;
    float qv[DH], acc[DH];
    float Dq = 0.f;
    const float qscale = 1.0f / sqrtf((float)DH);
#pragma unroll
    for (int e = 0; e < DH; ++e) { qv[e] = 0.f; acc[e] = 0.f; }
    if (live) {
        const h16* qp = base + (int64_t)s * d.ld + h * DH;
#pragma unroll
        for (int e = 0; e < DH; ++e) qv[e] = (float)qp[e] * qscale;
        const h16* dp = base + (int64_t)s * d.ld + 3 * C + h * d.ndecay;
        for (int f = 0; f < d.ndecay; ++f) Dq += (float)(f + 1) * aero_sigmoid((float)dp[f]);
        Dq *= 0.5f / sqrtf((float)d.ndecay);
    }
    float m = -1e30f, l = 0.f;
    for (int k0 = 0; k0 < T; k0 += KT) {
        __syncthreads();
        for (int idx = tid; idx < KT * DH; idx += 256) {
            const int kk = idx / DH, e = idx % DH;
            const int t = k0 + kk;
            float kv = 0.f, vv = 0.f;
            if (t < T) {
                const h16* kp = base + (int64_t)t * d.ld + C + h * DH + e;
                kv = (float)kp[0];
                vv = (float)kp[C];
            }
            Ks[idx] = kv;
            Vs[idx] = vv;
        }
        __syncthreads();
        const int kn = (T - k0) < KT ? (T - k0) : KT;
        if (live) {
            for (int kk = 0; kk < kn; ++kk) {
                const int t = k0 + kk;
                float sc = 0.f;
#pragma unroll
                for (int e = 0; e < DH; ++e) sc += qv[e] * Ks[kk * DH + e];
                const int dist = t > s ? t - s : s - t;
                sc -= (float)dist * Dq;
                if (t == s) sc = -100.f;                      // modules.py:120 "kill self reference"
                if (sc > m) {
                    const float a = aero_fast_exp(m - sc);
                    l *= a;
#pragma unroll
                    for (int e = 0; e < DH; ++e) acc[e] *= a;
                    m = sc;
                }
                const float pw = aero_fast_exp(sc - m);
                l += pw;
#pragma unroll
                for (int e = 0; e < DH; ++e) acc[e] += pw * Vs[kk * DH + e];
            }
        }
    }
    if (live) {
        const float inv = 1.0f / l;
        h16* op = (h16*)d.out + ((int64_t)row * T + s) * C + h * DH;
#pragma unroll
        for (int e = 0; e < DH; ++e) op[e] = (h16)(acc[e] * inv);
    }
}

static int aero_attn_launch(const aero_attn_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->qkvd || !d->out) { *err = "localstate: null pointer"; return AERO_ERR_ARG; }
    if (d->R < 1 || d->T < 1 || d->C < 1 || d->heads < 1 || d->C % d->heads || d->ndecay < 0) { *err = "localstate: bad geometry"; return AERO_ERR_ARG; }
    if (d->ld < 3 * d->C + d->heads * d->ndecay) { *err = "localstate: ld too small"; return AERO_ERR_ARG; }
    if (d->R > 65535) { *err = "localstate: too many rows for one launch"; return AERO_ERR_ARG; }
    const int dh = d->C / d->heads;
    dim3 grid((unsigned)((d->T + 255) / 256), (unsigned)d->heads, (unsigned)d->R), block(256);
    switch (dh) {
        case 1: AERO_LAUNCH((aero_attn_kernel<1>), grid, block, stream, *d); break;
        case 2: AERO_LAUNCH((aero_attn_kernel<2>), grid, block, stream, *d); break;
        case 4: AERO_LAUNCH((aero_attn_kernel<4>), grid, block, stream, *d); break;
        case 8: AERO_LAUNCH((aero_attn_kernel<8>), grid, block, stream, *d); break;
        case 12: AERO_LAUNCH((aero_attn_kernel<12>), grid, block, stream, *d); break;
        case 16: AERO_LAUNCH((aero_attn_kernel<16>), grid, block, stream, *d); break;
        case 24: AERO_LAUNCH((aero_attn_kernel<24>), grid, block, stream, *d); break;
        case 32: AERO_LAUNCH((aero_attn_kernel<32>), grid, block, stream, *d); break;
        default: *err = "localstate: head dim not in {1,2,4,8,12,16,24,32}"; return AERO_ERR_UNSUPPORTED;
    }
    return AERO_OK;
}
