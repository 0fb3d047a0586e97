// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal CPU emulation of the HIP subset used by aero_amd/csrc/*.hip, so that kernel
// index logic, LDS staging, masks and epilogues can be exercised in the build container
// (which has no GPU).  One fiber (ucontext) per GPU thread; __syncthreads(), wave shuffles
// and the gfx950 MFMA intrinsic used by the kernels are emulated with fiber barriers.
// Blocks are distributed over a few OS threads.
//
// This is NOT a fallback: the product library (libaero_hip.so) is built by hipcc for gfx950
// only and aero_amd never loads the emulated build.  tests/ builds it into
// tests/emu/libaero_emu.so and passes it explicitly to the engine.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

namespace emu {

struct Dim3 {
    unsigned x, y, z;
    Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct Fiber {
    ucontext_t ctx;
    Dim3 tid;
    int flat, lane, wave;
    bool done;
    char* stack;
};

typedef _Float16 eh8 __attribute__((ext_vector_type(8)));

struct WaveCtx {
    int count, gen, alive;
    uint64_t slot[64];
    eh8 A[64], B[64];
};

struct BlockCtx {
    Dim3 bid, bdim, gdim;
    int nthreads, alive, cur;
    int bar_count, bar_gen;
    std::vector<Fiber> fibers;
    std::vector<WaveCtx> waves;
    ucontext_t main_ctx;
    const std::function<void()>* body;
    char* dyn_smem;              // dynamic LDS of this block (emulated)
};

extern thread_local BlockCtx* g_blk;

static inline Fiber& me() { return g_blk->fibers[g_blk->cur]; }
static inline void yield() {
    BlockCtx* b = g_blk;
    swapcontext(&b->fibers[b->cur].ctx, &b->main_ctx);
}

static inline void block_barrier() {
    BlockCtx* b = g_blk;
    int gen = b->bar_gen;
    if (++b->bar_count >= b->alive) {
        b->bar_count = 0;
        b->bar_gen++;
    } else {
        while (b->bar_gen == gen) yield();
    }
}

static inline void wave_barrier() {
    BlockCtx* b = g_blk;
    WaveCtx& w = b->waves[me().wave];
    int gen = w.gen;
    if (++w.count >= w.alive) {
        w.count = 0;
        w.gen++;
    } else {
        while (w.gen == gen) yield();
    }
}

template <class T>
static inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl size");
    WaveCtx& w = g_blk->waves[me().wave];
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    w.slot[me().lane] = bits;
    wave_barrier();
    uint64_t r = w.slot[src & 63];
    wave_barrier();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}

void launch(Dim3 grid, Dim3 block, const std::function<void()>& body, size_t dyn_bytes = 0);

template <class T>
static inline T atomic_add_cas(T* addr, T val) {
    typedef typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type U;
    U* p = reinterpret_cast<U*>(addr);
    U old = __atomic_load_n(p, __ATOMIC_RELAXED);
    for (;;) {
        T cur;
        memcpy(&cur, &old, sizeof(T));
        T nv = cur + val;
        U nb;
        memcpy(&nb, &nv, sizeof(T));
        if (__atomic_compare_exchange_n(p, &old, nb, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return cur;
    }
}

}  // namespace emu

#define threadIdx (emu::me().tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)
typedef emu::Dim3 dim3;

static inline void __syncthreads() { emu::block_barrier(); }
template <class T>
static inline T __shfl_xor(T v, int m) { return emu::shfl_idx(v, emu::me().lane ^ m); }
template <class T>
static inline T __shfl_down(T v, int d) { int s = emu::me().lane + d; return emu::shfl_idx(v, s > 63 ? emu::me().lane : s); }
template <class T>
static inline T __shfl(T v, int src) { return emu::shfl_idx(v, src); }

static inline float atomicAdd(float* a, float v) { return emu::atomic_add_cas(a, v); }
static inline double atomicAdd(double* a, double v) { return emu::atomic_add_cas(a, v); }
static inline unsigned long long atomicAdd(unsigned long long* a, unsigned long long v) { return __atomic_fetch_add(a, v, __ATOMIC_RELAXED); }
static inline unsigned int atomicMax(unsigned int* a, unsigned int v) {
    unsigned int old = __atomic_load_n(a, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(a, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }


typedef float emu_f4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_f16: A lane l holds A[i=l&15][k=(l>>4)*8+e]; B lane l holds B[k=(l>>4)*8+e][j=l&15];
// D lane l reg r holds D[i=(l>>4)*4+r][j=l&15].
static inline emu_f4 __builtin_amdgcn_mfma_f32_16x16x32_f16(emu::eh8 a, emu::eh8 b, emu_f4 c, int, int, int) {
    emu::WaveCtx& w = emu::g_blk->waves[emu::me().wave];
    int lane = emu::me().lane;
    w.A[lane] = a;
    w.B[lane] = b;
    emu::wave_barrier();
    emu_f4 d;
    int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int i = (lane >> 4) * 4 + r;
        float s = c[r];
        for (int k = 0; k < 32; ++k) s += (float)w.A[i + 16 * (k >> 3)][k & 7] * (float)w.B[j + 16 * (k >> 3)][k & 7];
        d[r] = s;
    }
    emu::wave_barrier();
    return d;
}

// v_mfma_f32_16x16x16_f16: A lane l holds A[i=l&15][k=(l>>4)*4+e]; B lane l holds B[k=(l>>4)*4+e][j=l&15]; D as above.
typedef _Float16 emu_h4 __attribute__((ext_vector_type(4)));
static inline emu_f4 __builtin_amdgcn_mfma_f32_16x16x16f16(emu_h4 a, emu_h4 b, emu_f4 c, int, int, int) {
    emu::WaveCtx& w = emu::g_blk->waves[emu::me().wave];
    int lane = emu::me().lane;
    for (int e = 0; e < 4; ++e) { w.A[lane][e] = a[e]; w.B[lane][e] = b[e]; }
    emu::wave_barrier();
    emu_f4 d;
    int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int i = (lane >> 4) * 4 + r;
        float s = c[r];
        for (int k = 0; k < 16; ++k) s += (float)w.A[i + 16 * (k >> 2)][k & 3] * (float)w.B[j + 16 * (k >> 2)][k & 3];
        d[r] = s;
    }
    emu::wave_barrier();
    return d;
}

typedef float emu_f16v __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x16_f16: A lane l holds A[i=l&31][k=(l>>5)*8+e]; B lane l holds B[k=(l>>5)*8+e][j=l&31];
// D lane l reg r holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31].
static inline emu_f16v __builtin_amdgcn_mfma_f32_32x32x16_f16(emu::eh8 a, emu::eh8 b, emu_f16v c, int, int, int) {
    emu::WaveCtx& w = emu::g_blk->waves[emu::me().wave];
    int lane = emu::me().lane;
    w.A[lane] = a;
    w.B[lane] = b;
    emu::wave_barrier();
    emu_f16v d;
    int j = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float s = c[r];
        for (int k = 0; k < 16; ++k) s += (float)w.A[i + 32 * (k >> 3)][k & 7] * (float)w.B[j + 32 * (k >> 3)][k & 7];
        d[r] = s;
    }
    emu::wave_barrier();
    return d;
}

static thread_local const char* aero_last_kernel_str_ = "";
#define AERO_LAUNCH(kern, grid, block, stream, ...) \
    do { aero_last_kernel_str_ = #kern; emu::launch(grid, block, [=]() { kern(__VA_ARGS__); }); } while (0)
#define AERO_LAUNCH_DYN(kern, grid, block, dyn_bytes, stream, ...) \
    do { aero_last_kernel_str_ = #kern; emu::launch(grid, block, [=]() { kern(__VA_ARGS__); }, dyn_bytes); } while (0)
#define AERO_DYN_SMEM (emu::g_blk->dyn_smem)
