// aero_common.h -- shared types/helpers for the gfx950 kernels of the AERO spectral path.
//
// Activations are fp16, channels-last: element (b, f, t, c) of a tensor lives at
//   base + b*sb + f*sf + t*st + c            (strides in elements, channel stride 1)
// so one (b, f) "row" is a [T, C] slab that is contiguous along time.  GEMM operands are
// K-contiguous on both sides (weights [M][K], activations [pos][C]) which is what the
// 16x16x32 f16 MFMA fragments want (8 consecutive k per lane = one 16-byte load).
#pragma once
#include <stdint.h>

#ifdef AERO_EMU
#include "hip_emu.h"
#define aero_fast_exp(x) expf(x)
#define aero_rcp(x) (1.0f / (x))
#define aero_fast_sin(x) sinf(x)
#define aero_fast_cos(x) cosf(x)
#define aero_exp2(x) exp2f(x)
#define aero_rsqrt(x) (1.0f / sqrtf(x))
#define aero_med3(x, lo, hi) fminf(fmaxf((x), (lo)), (hi))
#else
#define aero_fast_exp(x) __expf(x)
#define aero_rcp(x) __builtin_amdgcn_rcpf(x)
#define aero_fast_sin(x) __sinf(x)
#define aero_fast_cos(x) __cosf(x)
#define aero_exp2(x) __builtin_amdgcn_exp2f(x)              /* bare v_exp_f32 */
#define aero_rsqrt(x) __builtin_amdgcn_rsqf(x)
#define aero_med3(x, lo, hi) __builtin_amdgcn_fmed3f((x), (lo), (hi))   /* one-instruction clamp */
#include <hip/hip_runtime.h>
// The library must be compiled WITHOUT packed-fp32 instructions (with them the FFT kernels return wrong values next to
// another stream's MFMA waves).  The build passes `-Xclang -target-feature -Xclang -packed-fp32-ops` TOGETHER with -DAERO_NO_PACKED_FP32
// (__graft_entry__.HIPCC_FLAGS); a compile line that lost the pair stops here instead of producing a library that is silently wrong
// under concurrency.  aero_version() reports the define, aero_amd/_lib.py refuses a library that does not, tools/isa_lint.py fails the
// build if a v_pk_{fma,mul,add}_f32 is found in the code object.  tools/dbg experiment builds opt out with -DAERO_ALLOW_PACKED_FP32.
#if !defined(AERO_NO_PACKED_FP32) && !defined(AERO_ALLOW_PACKED_FP32)
#error "aero_hip: compile with -Xclang -target-feature -Xclang -packed-fp32-ops -DAERO_NO_PACKED_FP32 (see __graft_entry__.HIPCC_FLAGS, DESIGN.md 5b)"
#endif
// every launch records which instantiation it was: aero_last_kernel_name() (profiling labels that match rocprofv3)
#ifdef AERO_PART
extern thread_local const void* aero_last_kernel_ptr_;      /* the library is built in parts (aero_hip.hip): defined in part 0 */
#else
static thread_local const void* aero_last_kernel_ptr_ = nullptr;
#endif
#define AERO_LAUNCH(kern, grid, block, stream, ...)                                  \
    do {                                                                             \
        aero_last_kernel_ptr_ = (const void*)(kern);                                 \
        hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);               \
    } while (0)
// dynamic LDS (up to the full 160 KiB of a CU): one extern array per translation unit
extern __shared__ __attribute__((aligned(16))) char aero_dyn_smem_[];
#define AERO_DYN_SMEM aero_dyn_smem_
#define AERO_LAUNCH_DYN(kern, grid, block, dyn_bytes, stream, ...)                                                   \
    do {                                                                                                              \
        static size_t aero_max_dyn_ = 0;                                                                              \
        if ((size_t)(dyn_bytes) > aero_max_dyn_) {                                                                    \
            (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(dyn_bytes)); \
            aero_max_dyn_ = (size_t)(dyn_bytes);                                                                      \
        }                                                                                                             \
        aero_last_kernel_ptr_ = (const void*)(kern);                                                                  \
        hipLaunchKernelGGL(kern, grid, block, dyn_bytes, stream, __VA_ARGS__);                                        \
    } while (0)
#endif

#include "../../include/aero_hip.h"

// The library is ONE source (aero_hip.hip) compiled in parts (-DAERO_PART=k, __graft_entry__.build): a host function one part
// calls in another has external linkage there, and is a file-local static when the whole library is a single unit.
#ifdef AERO_PART
#define AERO_XPART
#else
#define AERO_XPART static
#endif

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define AERO_LDS_ALIGN __attribute__((aligned(16)))

// Block barrier that orders LDS traffic only: it waits for this wave's LDS operations (lgkmcnt) but NOT for its global
// loads/stores (vmcnt).  __syncthreads() is a workgroup fence and drains both, which makes every tile of a streaming
// kernel wait for its own output stores to be acknowledged before the next tile may start.
static __device__ __forceinline__ void aero_lds_barrier() {
#ifdef AERO_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// 64 zero bytes+: source of masked lanes of the direct global->LDS copies (padding, out-of-range rows/channels)
static __device__ h16 aero_zero_page[64];

// Direct global -> LDS copy of 16 bytes per lane (gfx950 `global_load_lds_dwordx4`): no VGPR round trip and no
// ds_write issue cost.  The LDS destination is WAVE-UNIFORM base + lane*16 (linear); a swizzled image is obtained by
// permuting the per-lane SOURCE address.  Completion is tracked by vmcnt; __syncthreads() drains it.
static __device__ __forceinline__ void aero_glds16(const h16* gsrc, h16* lds_wave_base) {
#ifdef AERO_EMU
    memcpy(lds_wave_base + (threadIdx.x & 63) * 8, gsrc, 16);
#elif defined(AERO_DBG_NO_GLDS)                                 /* tools/dbg experiment builds only: the same copy through a VGPR */
    const h16x8 v = *(const h16x8*)gsrc;
    *(h16x8*)(lds_wave_base + (threadIdx.x & 63) * 8) = v;
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// Mark a value the whole wave agrees on as scalar.  Indexing a kernel-argument array (tap tables) with an index the
// compiler cannot prove uniform turns into a VECTOR load from the kernarg segment followed by `s_waitcnt vmcnt(0)`,
// which drains every global load in flight; with a readfirstlane'd index it is an s_load.
static __device__ __forceinline__ int aero_uniform(int x) {
#ifdef AERO_EMU
    return x;
#else
    return __builtin_amdgcn_readfirstlane(x);
#endif
}

// wave-level rendezvous between an LDS write and a read of other lanes' data by the SAME wave: hardware executes a
// wave's LDS instructions in order, so this is only a compiler scheduling fence (and a fiber rendezvous in the emulator)
static __device__ __forceinline__ void aero_wave_sync() {
#ifdef AERO_EMU
    emu::wave_barrier();
#elif defined(AERO_DBG_WAVE_SYNC_WAIT)                          /* tools/dbg experiment builds only */
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#else
    __builtin_amdgcn_wave_barrier();
#endif
}

// the values exist in registers HERE (an empty asm that "modifies" them): what computes them cannot be sunk below this point
static __device__ __forceinline__ void aero_pin(float& a, float& b) {
#ifndef AERO_EMU
    asm volatile("" : "+v"(a), "+v"(b));
#endif
}
static __device__ __forceinline__ void aero_sched_fence() {
#ifndef AERO_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int N>
static __device__ __forceinline__ void aero_wait_vm() {
#ifndef AERO_EMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// end of a phase: this wave's LDS reads have returned (their slot may be refilled) and every wave's landed copies are
// visible to the others.  Not __syncthreads(): that would also drain the copies still in flight (vmcnt).
static __device__ __forceinline__ void aero_phase_barrier() {
#ifdef AERO_EMU
    __syncthreads();
#else
    // the BUILTIN wait (not inline asm) so that hipcc's own scoreboard knows the operand registers fetched during this
    // phase are ready: with an asm wait it re-waits `lgkmcnt(0)` in front of the next phase's first MFMA, i.e. also for
    // the fragment reads just issued for the phase after -- the prefetch would never overlap the MFMAs.
    // sched_barrier(0): nothing moves across -- hipcc otherwise hoists register-only MFMAs of the next phase over the
    // s_barrier (legal, but it then waits for this phase's prefetch reads in front of them)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0); vmcnt / expcnt fields at their maxima
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}

static __device__ __forceinline__ int aero_lane() { return threadIdx.x & 63; }
static __device__ __forceinline__ int aero_wave() { return threadIdx.x >> 6; }

// Bijective XCD-aware remap (MI355X: block b runs on XCD b % 8, each XCD has a private L2):
// give every XCD a contiguous chunk of the logical grid so neighbouring tiles share L2.
static __device__ __forceinline__ int aero_xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static __device__ __forceinline__ float aero_sigmoid(float x) { return aero_rcp(1.0f + aero_fast_exp(-x)); }
static __device__ __forceinline__ float aero_tanh(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); saturates cleanly for |x| large
    float e = aero_fast_exp(2.0f * x);
    return 1.0f - 2.0f * aero_rcp(e + 1.0f);
}
// exact-erf GELU (F.gelu default).  erf by Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 (far below fp16 output
// rounding), ~3x cheaper than erff on the vector ALU -- the norm/activation kernels are HBM-bound only if the
// per-element instruction count stays small.
static __device__ __forceinline__ float aero_erf(float x) {
    const float ax = fabsf(x);
    const float t = aero_rcp(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float r = 1.0f - poly * aero_fast_exp(-ax * ax);
    return x < 0.f ? -r : r;
}
static __device__ __forceinline__ float aero_gelu(float x) { return 0.5f * x * (1.0f + aero_erf(x * 0.70710678118654752f)); }
// Two GELUs at once (packed fp32: the affine parts compile to v_pk_fma_f32 / v_pk_mul_f32).  Same A&S 7.1.26 erf,
// rearranged so that no sign select is needed:  gelu(y) = max(y, 0) - |y| * (poly(t)/2) * exp(-y^2/2),
// t = 1/(1 + p|y|/sqrt2);  the 1/2 and the 1/sqrt2 are folded into the constants, exp(-y^2/2) = exp2(-0.7213475 y^2).
static __device__ __forceinline__ f32x2 aero_gelu2(f32x2 y) {
    const f32x2 ay = {fabsf(y[0]), fabsf(y[1])};
    const f32x2 u = ay * 0.23164189678f + 1.0f;
    const f32x2 t = {aero_rcp(u[0]), aero_rcp(u[1])};
    const f32x2 a = y * y * -0.72134752044f;
    const f32x2 E = {aero_exp2(a[0]), aero_exp2(a[1])};
    const f32x2 p = ((((0.5307027145f * t - 0.7265760135f) * t + 0.7107068705f) * t - 0.142248368f) * t + 0.127414796f) * t;
    const f32x2 g = {fmaxf(y[0], 0.f), fmaxf(y[1], 0.f)};
    return g - ay * (p * E);
}

// true if the predicate holds in any lane of the wave (one s_cmp on the vote mask; the emulator sums)
static __device__ __forceinline__ bool aero_wave_any(bool pred);

template <class T>
static __device__ __forceinline__ T aero_wave_sum(T v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

static __device__ __forceinline__ bool aero_wave_any(bool pred) {
#ifdef AERO_EMU
    return aero_wave_sum(pred ? 1 : 0) != 0;
#else
    return __builtin_amdgcn_ballot_w64(pred) != 0;
#endif
}

// LDS image of a [rows][32] fp16 tile (64-byte rows).  ds_read_b128 is serviced in four 16-lane
// groups over a 256-byte bank row; fragment reads put lane (l&15) on row (l&15) at 16-byte slot
// q = l>>4, so un-swizzled rows r, r+4, r+8, r+12 collide.  slot' = q ^ ((-(r>>2)) & 3) makes every
// group conflict free (checked against the gfx950 lane-group table in the design notes).
static __device__ __forceinline__ int aero_tile_off(int row, int q) {
    return row * 32 + ((q ^ ((0 - (row >> 2)) & 3)) << 3);
}

static inline int aero_cdiv(int a, int b) { return (a + b - 1) / b; }
