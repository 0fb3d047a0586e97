// pk_probe.hip -- round 4: which VALU instruction classes come out wrong when MFMA-heavy kernels of ANOTHER stream share the chip?
// (The FFT-form STFT returned wrong real parts in lanes 48-63 next to the ring conv tile; the same source compiled without
// packed-fp32 instructions never did: tools/dbg/variants.py.)  Register-only dependent chains of ONE instruction class per kernel, no LDS,
// no memory traffic but the final store; the host compares every launch bit for bit with a launch that had the chip to itself.
// Built on its own (tools/dbg/pk_probe.py --build); never part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// MODE 0: v_fma_f32 (control)   1: v_pk_fma_f32   2: v_pk_fma_f32 with op_sel (the complex-multiply form hipcc emits for cmul)
//      3: v_pk_mul_f32 + v_pk_add_f32   4: v_pk_fma_f16   5: v_pk_fma_f32 issued in pairs with a v_exp_f32 (transcendental unit) between
template <int MODE>
__global__ __launch_bounds__(256) void pk_probe_kernel(float* out, int iters) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    f32x2 a = {1.0f + 1e-3f * lane, 0.5f - 1e-3f * lane};
    const f32x2 b = {0.999f - 1e-5f * (gid & 1023), 0.9985f + 1e-5f * (gid & 511)};
    const f32x2 c = {1e-3f * (1 + (gid & 7)), -2e-3f};
    h16x2 ha = {(_Float16)(1.0f + 0.01f * lane), (_Float16)(0.5f - 0.005f * lane)};
    const h16x2 hb = {(_Float16)0.99f, (_Float16)0.98f}, hc = {(_Float16)0.01f, (_Float16)-0.02f};
    float t = 0.25f;
    const double sc2 = __builtin_bit_cast(double, (unsigned long long)0x3f7fbe773f7fbe77ull);      // {0.999, 0.999} as an SGPR pair
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 13) {                // control for 12: the same EXEC write after two plain v_fma_f32
            unsigned long long save;
            asm volatile("v_fma_f32 %0, %0, %4, %6\n\tv_fma_f32 %1, %1, %5, %7\n\ts_mov_b64 %2, exec\n\ts_mov_b64 exec, 1\n\tv_add_f32 %3, 1.0, %3\n\ts_mov_b64 exec, %2"
                         : "+v"(a[0]), "+v"(a[1]), "=&s"(save), "+v"(t) : "v"(b[0]), "v"(b[1]), "v"(c[0]), "v"(c[1]));
        } else if constexpr (MODE == 0) {
            asm volatile("v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5" : "+v"(a[0]), "+v"(a[1]) : "v"(b[0]), "v"(b[1]), "v"(c[0]), "v"(c[1]));
        }
#ifndef NO_PK_ASM
        else if constexpr (MODE == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (MODE == 2) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (MODE == 3) {
            asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (MODE == 8) {          // SGPR-pair operand, low half broadcast (X * scale in the STFT's unpack)
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %2" : "+v"(a) : "s"(sc2), "v"(c));
        } else if constexpr (MODE == 9) {          // inline constant operand
            asm volatile("v_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
        } else if constexpr (MODE == 10) {         // negation through the packed add with an inline zero
            asm volatile("v_pk_add_f32 %0, %0, 0 neg_lo:[1,1] neg_hi:[1,1]\n\tv_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (MODE == 11) {         // fma with an inline constant in the middle
            asm volatile("v_pk_fma_f32 %0, %0, 0.5, %1 op_sel_hi:[1,0,1]" : "+v"(a) : "v"(c));
        } else if constexpr (MODE == 12) {         // packed op IMMEDIATELY followed by an EXEC write that turns lanes 1..63 off (the shape of a divergent
            unsigned long long save;               // if / else after the op: s_andn2_saveexec_b64 in the STFT's unpack), then a VALU op, then restore
            asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tv_add_f32 %2, 1.0, %2\n\ts_mov_b64 exec, %1"
                         : "+v"(a), "=&s"(save), "+v"(t) : "v"(b), "v"(c));
        } else if constexpr (MODE == 5) {
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_exp_f32 %1, %1\n\tv_pk_fma_f32 %0, %0, %2, %3" : "+v"(a), "+v"(t) : "v"(b), "v"(c));
            t = t * 0.5f - 1.0f;
        }
#endif
        else if constexpr (MODE == 4) {
            asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(ha) : "v"(hb), "v"(hc));
        }
    }
    if constexpr (MODE == 4) { out[2 * gid] = (float)ha[0]; out[2 * gid + 1] = (float)ha[1]; }
    else { out[2 * gid] = a[0]; out[2 * gid + 1] = a[1] + (MODE == 5 ? t : 0.f); }
}

// MODE 6 (plain C++, whatever the compiler selects -- built twice, with and without packed-fp32 instructions): butterflies through LDS
// like a Stockham pass: each lane reads two complex values other lanes wrote, multiplies by a twiddle from LDS, writes two results
__global__ __launch_bounds__(256) void pk_lds_kernel(float* out, int iters) {
    __shared__ f32x2 buf[2][4][256];
    __shared__ f32x2 tw[256];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tw[threadIdx.x] = (f32x2){__cosf(0.0245f * threadIdx.x), -__sinf(0.0245f * threadIdx.x)};
    for (int m = lane; m < 256; m += 64) buf[0][wave][m] = (f32x2){1.0f + 1e-3f * ((gid + m) & 255), 0.5f - 1e-3f * (m & 127)};
    __syncthreads();
    int cur = 0;
    for (int i = 0; i < iters; ++i) {
        const f32x2* src = buf[cur][wave];
        f32x2* dst = buf[cur ^ 1][wave];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = lane + 64 * it;
            const f32x2 A = src[k], B = src[k + 128];
            const f32x2 w = tw[(k * (1 + (i & 3))) & 255];
            const f32x2 c2 = (f32x2){w[0] * B[0] - w[1] * B[1], w[0] * B[1] + w[1] * B[0]};
            dst[(2 * k) & 255] = (A + c2) * 0.70710678f;
            dst[(2 * k + 1) & 255] = (A - c2) * 0.70710678f;
        }
        __builtin_amdgcn_wave_barrier();
        cur ^= 1;
    }
    out[2 * gid] = buf[cur][wave][lane][0] + buf[cur][wave][lane + 64][0] + buf[cur][wave][lane + 128][0] + buf[cur][wave][lane + 192][0];
    out[2 * gid + 1] = buf[cur][wave][lane][1] + buf[cur][wave][lane + 64][1] + buf[cur][wave][lane + 128][1] + buf[cur][wave][lane + 192][1];
}

extern "C" int pk_probe(int mode, float* out, int nblocks, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case 13: hipLaunchKernelGGL(pk_probe_kernel<13>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 0: hipLaunchKernelGGL(pk_probe_kernel<0>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 1: hipLaunchKernelGGL(pk_probe_kernel<1>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 2: hipLaunchKernelGGL(pk_probe_kernel<2>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 3: hipLaunchKernelGGL(pk_probe_kernel<3>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 4: hipLaunchKernelGGL(pk_probe_kernel<4>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 5: hipLaunchKernelGGL(pk_probe_kernel<5>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
#ifndef NO_PK_ASM
        case 8: hipLaunchKernelGGL(pk_probe_kernel<8>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 9: hipLaunchKernelGGL(pk_probe_kernel<9>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 10: hipLaunchKernelGGL(pk_probe_kernel<10>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 11: hipLaunchKernelGGL(pk_probe_kernel<11>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
        case 12: hipLaunchKernelGGL(pk_probe_kernel<12>, dim3(nblocks), dim3(256), 0, s, out, iters); break;
#endif
        default: hipLaunchKernelGGL(pk_lds_kernel, dim3(nblocks), dim3(256), 0, s, out, iters / 8); break;
    }
    return (int)hipGetLastError();
}
