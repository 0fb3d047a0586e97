// k_optim.h -- the generator's optimizer step (reference train.py:83: torch.optim.Adam(lr, betas=(0.9, beta2)), solver.py:602-605)
// fused over ONE flat fp32 buffer: every parameter of the model is a view into `p`, every gradient a view into `g` (the layout
// DistributedDataParallel's buckets already have), so a step is a single pass  p, m, v <- f(p, g, m, v)  instead of ~330 small
// launches per tensor.  Same arithmetic, in the same order, as torch's single-tensor Adam (no amsgrad, no weight decay):
//   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 28 bytes per parameter (read p, g, m, v; write p, m, v): 0.54 GB for the 19.4 M-parameter generator.
#pragma once
#include "aero_common.h"

__global__ __launch_bounds__(256) void aero_adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                                                         float b2, float eps, float bc1, float bc2_sqrt, float grad_scale, const float* bc_dev) {
    if (bc_dev) { bc1 = bc_dev[0]; bc2_sqrt = bc_dev[1]; }                  // bias corrections from device memory (a replayed HIP graph: see aero_adam_step_dev)
    const float step_size = lr / bc1;
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * grad_scale;
            mm[e] = mm[e] + (gr - mm[e]) * (1.0f - b1);                     // torch: exp_avg.lerp_(grad, 1 - beta1)
            vv[e] = vv[e] * b2 + (1.0f - b2) * gr * gr;                     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pp[e] = pp[e] - step_size * (mm[e] / denom);                    // param.addcdiv_(exp_avg, denom, value=-step_size)
        }
        ((f32x4*)p)[i] = pp;
        ((f32x4*)m)[i] = mm;
        ((f32x4*)v)[i] = vv;
    }
    if (blockIdx.x == 0) {                                                   // the < 4 tail elements
        for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += 256) {
            const float gr = g[i] * grad_scale;
            const float mi = m[i] + (gr - m[i]) * (1.0f - b1);
            const float vi = v[i] * b2 + (1.0f - b2) * gr * gr;
            m[i] = mi;
            v[i] = vi;
            p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        }
    }
}

 static int aero_adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int32_t step,
                            float grad_scale, hipStream_t stream, const char** err, const float* bc_dev = nullptr) {
    if (!p || !g || !m || !v || n < 1) { *err = "adam: null pointer / empty buffer"; return AERO_ERR_ARG; }
    if (bc_dev) step = 1;
    if (step < 1 || !(b1 >= 0.f && b1 < 1.f) || !(b2 >= 0.f && b2 < 1.f) || !(eps >= 0.f)) { *err = "adam: bad hyper-parameters"; return AERO_ERR_ARG; }
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) { *err = "adam: buffers must be 16-byte aligned"; return AERO_ERR_ARG; }
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    const int64_t want = ((n >> 2) + 255) / 256;
    const unsigned nb = (unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    AERO_LAUNCH(aero_adam_kernel, dim3(nb), dim3(256), stream, p, g, m, v, n, lr, b1, b2, eps, (float)bc1, (float)sqrt(bc2), grad_scale, bc_dev);
    return AERO_OK;
}
