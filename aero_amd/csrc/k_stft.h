// k_stft.h -- STFT / iSTFT front-end kernels (HBM-bound; fp32 throughout).
//
// STFT  (reference spec.py:9-22 via aero.py:409-421): real FFT of size n_fft computed as a complex
//        Stockham radix-2 FFT of size n = n_fft/2 on the even/odd packed frame, one frame per
//        wavefront, butterflies staged through LDS; a block's frames are transposed through an LDS
//        tile so spectrogram writes are contiguous runs along the frame axis.
// iSTFT (spec.py:25-39 via aero.py:423-428): Hermitian unpack -> complex FFT (conjugate trick) ->
//        window -> output-stationary overlap-add in LDS (every output sample is produced by exactly
//        one thread; no atomics) -> divide by the window envelope -> trim/crop.
// Algorithmic bytes: STFT reads 4 B/sample, writes 8 B/(bin,frame);
// iSTFT reads 8 B/(bin,frame), writes 4 B/sample.
#pragma once
#include "aero_common.h"

#define AERO_FFT_MAX_N 512    /* iSTFT: complex points = n_fft/2  ->  n_fft <= 1024 */
#define AERO_STFT_MAX_N 1024  /* STFT:  n_fft <= 2048 (the loss / metric geometries of stft_loss.py:120-123, metrics.py:58) */
#define AERO_STFT_SPAN 1536   /* default samples of signal a block's frames share through LDS (grown per launch if needed) */

static __device__ __forceinline__ f32x2 aero_cmul(f32x2 a, f32x2 b) {
    return (f32x2){a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]};
}

// tw[k] = exp(-2*pi*i*k/n_fft), k in [0, n_fft/2)
static __device__ __forceinline__ void aero_fft_init_twiddles(f32x2* tw, int n_fft) {
    for (int k = threadIdx.x; k < n_fft / 2; k += 256) {
#ifdef AERO_EMU
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)n_fft;
        tw[k] = (f32x2){(float)cos(a), (float)sin(a)};
#else
        // sincospi on the exactly representable fraction -2k/n_fft: float-accurate without a double sin/cos per block
        float sn, cs;
        sincospif(-2.0f * (float)k / (float)n_fft, &sn, &cs);
        tw[k] = (f32x2){cs, sn};
#endif
    }
}

// Stockham autosort FFT of n points, one problem per wavefront (waves run independently): radix-4 passes (each the
// exact composition of two radix-2 passes, same twiddle table entries, so results equal the radix-2 form bit for bit)
// and one closing radix-2 pass when log2(n) is odd.  n = 256 is 4 passes of one butterfly per lane.
// Returns the buffer (a or b) that holds the natural-order result.
// LDS position of element idx in the buffer a pass with butterfly span p writes: ds_write_b64 is serviced in groups of
// 16 lanes over 32 banks, and the Stockham scatter of the first two radix-4 passes (stride 4, then runs of 4 every 16)
// would put a group on 4 of its 16 element slots (4-way conflicts).  XOR-ing bits 4-5 of idx into the low bits keeps every
// aligned 16-element run a permutation of itself, so the contiguous reads of the next pass stay conflict-free.
static __device__ __forceinline__ int aero_fft_swz(int idx, int p) {
    return p == 1 ? idx ^ ((idx >> 4) & 3) : p == 4 ? idx ^ (((idx >> 4) & 3) << 2) : idx;
}

// LOGN > 0: n = 2^LOGN is a compile-time constant, every pass and lane loop unrolls and strides, twiddle steps and
// swizzles fold (the kernels were instruction-bound: ~1200 VALU + ~800 SALU per wave for two frames with runtime n).
// LOGN = 0: the same code with n at run time (other power-of-two sizes).
template <int LOGN>
static __device__ __forceinline__ f32x2* aero_fft_wave(f32x2* a, f32x2* b, int n_rt, const f32x2* tw) {
    const int n = LOGN ? (1 << LOGN) : n_rt;
    const int n_fft = 2 * n;
    const int lane = aero_lane();
    const int half = n >> 1, quarter = n >> 2;
    const int logn = LOGN ? LOGN : 31 - __builtin_clz(n);
    const int np4 = logn >> 1;                                   // radix-4 passes
    f32x2* src = a;
    f32x2* dst = b;
#pragma unroll(LOGN ? 8 : 1)
    for (int s4 = 0; s4 < np4; ++s4) {
        const int p = 1 << (2 * s4);
        const int tws2 = n_fft / (2 * p), tws4 = n_fft / (4 * p);
        const int pr = p >> 2;                                   // the pass that wrote src (0: linear input)
        const int iters = (quarter + 63) >> 6;
#pragma unroll(LOGN ? 8 : 1)
        for (int it = 0; it < iters; ++it) {
            const int i = lane + 64 * it;
            if (i < quarter) {
                const int k = i & (p - 1);
                const f32x2 w2 = tw[k * tws2], w1 = tw[k * tws4], w3 = tw[(k + p) * tws4];
                const f32x2 A = src[aero_fft_swz(i, pr)], B = src[aero_fft_swz(i + quarter, pr)];
                const f32x2 Cc = src[aero_fft_swz(i + half, pr)], D = src[aero_fft_swz(i + half + quarter, pr)];
                const f32x2 c2 = aero_cmul(w2, Cc), d2 = aero_cmul(w2, D);
                const f32x2 m0 = A + c2, m1 = A - c2, m2 = B + d2, m3 = B - d2;
                const f32x2 e = aero_cmul(w1, m2), f = aero_cmul(w3, m3);
                const int jj = ((i - k) << 2) + k;
                dst[aero_fft_swz(jj, p)] = m0 + e;
                dst[aero_fft_swz(jj + p, p)] = m1 + f;
                dst[aero_fft_swz(jj + 2 * p, p)] = m0 - e;
                dst[aero_fft_swz(jj + 3 * p, p)] = m1 - f;
            }
        }
        aero_wave_sync();              // a wave owns its two buffers: LDS is in-order per wave, no block barrier needed
        f32x2* tmp = src;
        src = dst;
        dst = tmp;
    }
    if (logn & 1) {                    // closing radix-2 pass
        const int p = half;
        const int pr = p >> 2;
        const int iters = (half + 63) >> 6;
#pragma unroll(LOGN ? 8 : 1)
        for (int it = 0; it < iters; ++it) {
            const int i = lane + 64 * it;
            if (i < half) {            // k = i: p = half, tw step n_fft / (2p) = 2
                const f32x2 u0 = src[aero_fft_swz(i, pr)];
                const f32x2 v = aero_cmul(tw[2 * i], src[aero_fft_swz(i + half, pr)]);
                dst[i] = u0 + v;
                dst[i + p] = u0 - v;
            }
        }
        aero_wave_sync();
        f32x2* tmp = src;
        src = dst;
        dst = tmp;
    }
    // (the result is linear: a closing radix-2 pass writes linearly, p >= 16 passes are not swizzled, and the only sizes
    // that end on a p = 1 / p = 4 pass are n = 4 / 16, where idx >> 4 = 0)
    return src;
}

// 256 points with the pass twiddles in REGISTERS (round 4, the ring-resident iSTFT).  Measured there (tools/dbg/istft_ablation.py): the
// transforms are bound by VALU issue and LDS stores together, not by latency -- two interleaved frames per wave changed nothing -- so
// what is taken out is work: a lane's twiddles depend on (pass, lane) only, so the nine of passes 1-3 are loaded once per kernel
// (aero_fft256_twiddles) instead of three LDS reads and their address arithmetic per butterfly, and pass 0 (all twiddles = 1) has no
// complex products at all.  Same butterflies and the same twiddle values as aero_fft_wave<8>: results are bit-identical.
struct AeroFft256Tw { f32x2 w[3][3]; };                        // [pass - 1][w2, w1, w3]
static __device__ __forceinline__ AeroFft256Tw aero_fft256_twiddles(const f32x2* tw) {
    AeroFft256Tw t;
    const int i = aero_lane();
#pragma unroll
    for (int s4 = 1; s4 < 4; ++s4) {
        const int p = 1 << (2 * s4), k = i & (p - 1);
        t.w[s4 - 1][0] = tw[k * (512 / (2 * p))];
        t.w[s4 - 1][1] = tw[k * (512 / (4 * p))];
        t.w[s4 - 1][2] = tw[(k + p) * (512 / (4 * p))];
    }
    return t;
}
static __device__ __forceinline__ void aero_fft256_regtw(f32x2* a, f32x2* b, const AeroFft256Tw& t) {   // in place: four passes a -> b -> a -> b -> a
    constexpr int half = 128, quarter = 64;
    const int i = aero_lane();
    f32x2* src = a;
    f32x2* dst = b;
    {                                                            // pass 0: p = 1, k = 0, linear input
        const f32x2 A = src[i], B = src[i + quarter], Cc = src[i + half], D = src[i + half + quarter];
        const f32x2 m0 = A + Cc, m1 = A - Cc, m2 = B + D, m3 = B - D;
        const f32x2 f = (f32x2){m3[1], -m3[0]};                 // w3 = tw[n_fft / 4] = -i   (cmul(-i, m3); the table holds exactly (0, -1))
        const int jj = i << 2;
        dst[aero_fft_swz(jj, 1)] = m0 + m2;
        dst[aero_fft_swz(jj + 1, 1)] = m1 + f;
        dst[aero_fft_swz(jj + 2, 1)] = m0 - m2;
        dst[aero_fft_swz(jj + 3, 1)] = m1 - f;
        aero_wave_sync();
        f32x2* tmp = src; src = dst; dst = tmp;
    }
#pragma unroll
    for (int s4 = 1; s4 < 4; ++s4) {
        const int p = 1 << (2 * s4), pr = p >> 2;
        const int k = i & (p - 1);
        const f32x2 w2 = t.w[s4 - 1][0], w1 = t.w[s4 - 1][1], w3 = t.w[s4 - 1][2];
        const f32x2 A = src[aero_fft_swz(i, pr)], B = src[aero_fft_swz(i + quarter, pr)];
        const f32x2 Cc = src[aero_fft_swz(i + half, pr)], D = src[aero_fft_swz(i + half + quarter, pr)];
        const f32x2 c2 = aero_cmul(w2, Cc), d2 = aero_cmul(w2, D);
        const f32x2 m0 = A + c2, m1 = A - c2, m2 = B + d2, m3 = B - d2;
        const f32x2 e = aero_cmul(w1, m2), f = aero_cmul(w3, m3);
        const int jj = ((i - k) << 2) + k;
        dst[aero_fft_swz(jj, p)] = m0 + e;
        dst[aero_fft_swz(jj + p, p)] = m1 + f;
        dst[aero_fft_swz(jj + 2 * p, p)] = m0 - e;
        dst[aero_fft_swz(jj + 3 * p, p)] = m1 - f;
        aero_wave_sync();
        f32x2* tmp = src; src = dst; dst = tmp;
    }
}

struct AeroStftK {
    const float* x; const float* window; float* spec; double* stats;
    int nsig, L, Lp, n_fft, hop, n_bins, T, sig_per_item, FPB;
    int span_cap;                                  // floats of LDS reserved for the shared signal span (0: read frames from global)
#ifdef AERO_DBG_ZERO_LDS
    int dbg_lds_bytes;
#endif
};

// dynamic LDS sized by the actual n = n_fft/2 and frames per block (statically sized for n_fft = 1024 it was 62 KiB:
// two blocks per CU):   tw[n] | bufA[4][n] | bufB[4][n] | tile[n_bins*FPB]   (f32x2)   then   wl[n_fft] | xsp[SPAN]   (float)
static inline size_t aero_stft_lds_bytes(int n_fft, int n_bins, int fpb, int span_cap) {
    const size_t n = (size_t)n_fft / 2;
    return (n + 8 * n + (size_t)n_bins * (fpb + 1)) * sizeof(f32x2) + ((size_t)n_fft + (size_t)span_cap) * sizeof(float);
}

static inline int aero_stft_fpb(int n) { int f = 2048 / n; return f > 32 ? 32 : (f < 4 ? 4 : f); }   // frames per block (power of two)

template <int LOGN>
__global__ __launch_bounds__(256) void aero_stft_kernel(AeroStftK p) {
    __shared__ double red[2][4];
    const int n = LOGN ? (1 << LOGN) : (p.n_fft >> 1);
    const int n_fft = 2 * n;
    const int FPB = LOGN ? ((2048 >> LOGN) > 32 ? 32 : ((2048 >> LOGN) < 4 ? 4 : (2048 >> LOGN))) : p.FPB;
    const int TS = FPB + 1;                                      // odd tile row stride: a ds_write group's 16 lanes hit 16 slots
    f32x2* tw = (f32x2*)AERO_DYN_SMEM;
    f32x2* bufA0 = tw + n;
    f32x2* bufB0 = bufA0 + 4 * n;
    f32x2* tile = bufB0 + 4 * n;
    float* wl = (float*)(tile + p.n_bins * TS);
    float* xsp = wl + n_fft;
    const int lane = aero_lane(), wave = aero_uniform(aero_wave());
    const int sig = blockIdx.y;
    const int tbase = blockIdx.x * FPB;
    const float* xs = p.x + (int64_t)sig * p.L;
#ifdef AERO_DBG_ZERO_LDS                                      /* tools/dbg experiment builds only: no stale LDS contents */
    {
        unsigned* z = (unsigned*)AERO_DYN_SMEM;
        for (int i = threadIdx.x; i < (int)(p.dbg_lds_bytes >> 2); i += blockDim.x) z[i] = 0u;
        __syncthreads();
    }
#endif
    const float scale = 1.0f / sqrtf((float)n_fft);
    aero_fft_init_twiddles(tw, n_fft);
    // The block's frames overlap (hop << n_fft): the window and the reflect-padded signal span they share are staged in
    // LDS once, with independent coalesced loads.  (Reading window[ni] and then, if non-zero, x[...] from global memory
    // per element made every frame a chain of ~16 dependent L2 round trips: 134 us for a 68-MB kernel.)
    const int span = (FPB - 1) * p.hop + n_fft;
    const bool staged = span <= p.span_cap;
    const bool vec = staged && !(p.hop & 1);                     // 8-byte LDS reads (the scalar form is a stride-2 bank pattern)
    for (int i = threadIdx.x; i < n_fft; i += 256) wl[i] = p.window[i];
    if (staged) {
        for (int j = threadIdx.x; j < span; j += 256) {
            int xi = tbase * p.hop + j - n;                      // index into the hop-padded signal
            if (xi < 0) xi = -xi;                                // reflect (no edge repeat)
            if (xi >= p.Lp) xi = 2 * (p.Lp - 1) - xi;
            xsp[j] = (xi >= 0 && xi < p.L) ? xs[xi] : 0.f;       // [L, Lp) is the zero pad of aero.py:410
        }
    }
    __syncthreads();
    float s = 0.f, ss = 0.f;
    const int rounds = (FPB + 3) / 4;
    f32x2* bufA = bufA0 + wave * n;
    f32x2* bufB = bufB0 + wave * n;
#pragma unroll(LOGN ? 2 : 1)
    for (int r = 0; r < rounds; ++r) {
        const int fr = r * 4 + wave;
        const int t = tbase + fr;
        const bool live = fr < FPB && t < p.T;
        const int miters = (n + 63) >> 6;
#pragma unroll(LOGN ? 8 : 1)
        for (int it = 0; it < miters; ++it) {
            const int m = lane + 64 * it;
            if (m >= n) break;
            f32x2 g = (f32x2){0.f, 0.f};
            if (live && vec) {
                const f32x2 w = *(const f32x2*)(wl + 2 * m);
                const f32x2 xv = *(const f32x2*)(xsp + fr * p.hop + 2 * m);
                g = w * xv;
            } else if (live) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ni = 2 * m + e;
                    const float w = wl[ni];
                    if (staged) {
                        g[e] = w * xsp[fr * p.hop + ni];
                    } else if (w != 0.f) {
                        int xi = t * p.hop + ni - n;
                        if (xi < 0) xi = -xi;
                        if (xi >= p.Lp) xi = 2 * (p.Lp - 1) - xi;
                        g[e] = (xi < p.L) ? w * xs[xi] : 0.f;
                    }
                }
            }
            bufA[m] = g;
        }
        aero_wave_sync();                                        // a wave packs and transforms its own frame
        const f32x2* R = aero_fft_wave<LOGN>(bufA, bufB, n, tw);
        if (fr < FPB) {
            const int kiters = (n >> 6) + 1;
#pragma unroll(LOGN ? 9 : 1)
            for (int it = 0; it < kiters; ++it) {
                const int k = lane + 64 * it;
                if (k >= p.n_bins) break;
                f32x2 X;
                if (k == 0) {
                    X = (f32x2){R[0][0] + R[0][1], 0.f};
                } else if (k == n) {
                    X = (f32x2){R[0][0] - R[0][1], 0.f};
                } else {
                    const f32x2 zk = R[k];
                    const f32x2 zc = (f32x2){R[n - k][0], -R[n - k][1]};
                    const f32x2 E = (zk + zc) * 0.5f;
                    const f32x2 D = (zk - zc) * 0.5f;
                    const f32x2 O = (f32x2){D[1], -D[0]};
                    X = E + aero_cmul(tw[k], O);
                }
                X = X * scale;
                tile[k * TS + fr] = X;
                if (live) { s += X[0] + X[1]; ss += X[0] * X[0] + X[1] * X[1]; }
            }
        }
        aero_wave_sync();                                        // the next round's pack reuses bufA/bufB of this wave only
    }
    __syncthreads();
    const int total = p.n_bins * FPB;
    f32x2* out = (f32x2*)p.spec + (int64_t)sig * p.n_bins * p.T;
    const int lf = 31 - __builtin_clz(FPB);                      // FPB is a power of two
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int k = idx >> lf, fr = idx & (FPB - 1);
        const int t = tbase + fr;
        if (t < p.T) out[k * p.T + t] = tile[idx + k];
    }
    if (p.stats) {
        double ds = aero_wave_sum((double)s), dss = aero_wave_sum((double)ss);
        if (lane == 0) { red[0][wave] = ds; red[1][wave] = dss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* st = p.stats + 2 * (int64_t)(sig / p.sig_per_item);
            atomicAdd(st, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            atomicAdd(st + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        }
    }
}

__global__ __launch_bounds__(256) void aero_spec_normalize_kernel(const float* spec, int64_t n_per_item, const double* stats,
                                                                  h16* xn, float* mean_std) {
    const int item = blockIdx.y;
    const double N = (double)n_per_item;
    const double S = stats[2 * item], SS = stats[2 * item + 1];
    const double mean = S / N;
    double var = (SS - S * S / N) / (N - 1.0);     // unbiased (torch.std default, aero.py:463)
    if (var < 0) var = 0;
    const float fm = (float)mean, fs = (float)sqrt(var);
    const float inv = 1.0f / (1e-5f + fs);
    if (blockIdx.x == 0 && threadIdx.x == 0) { mean_std[2 * item] = fm; mean_std[2 * item + 1] = fs; }
    const float* src = spec + (int64_t)item * n_per_item;
    h16* dst = xn + (int64_t)item * n_per_item;
    if ((n_per_item & 3) == 0 && ((((uintptr_t)src) & 15) | (((uintptr_t)dst) & 7)) == 0) {
        // four values a trip (16-byte loads, 8-byte stores), several trips per thread: the fp64 square root / divisions above are per
        // thread, and with one 2-value trip each (the first form) they were most of the kernel (30 us for 99 MB)
        for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < n_per_item; e += (int64_t)gridDim.x * 1024) {
            const f32x4 v = *(const f32x4*)(src + e);
            *(h16x4*)(dst + e) = (h16x4){(h16)((v[0] - fm) * inv), (h16)((v[1] - fm) * inv), (h16)((v[2] - fm) * inv), (h16)((v[3] - fm) * inv)};
        }
        return;
    }
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; e < n_per_item; e += (int64_t)gridDim.x * 512) {
        const f32x2 v = *(const f32x2*)(src + e);
        *(h16x2*)(dst + e) = (h16x2){(h16)((v[0] - fm) * inv), (h16)((v[1] - fm) * inv)};
    }
}

struct AeroIstftK {
    const float* spec; const float* window; const float* inv_env; float* y;
    int nsig, F, T, n_fft, hop, hsh, Lout, FPB, SEG;         // hsh = log2(hop) if hop is a power of two, else -1
    int P, toff;                                             // row pitch of `spec` in frames (>= toff + T) and the column of frame 0 (aero_istft2_kernel only)
#ifdef AERO_ISTFT_DEBUG
    // tools/dbg builds only (never the product library): bit 0 = re-read every spectrum value past the caches (sc0 sc1) and count the
    // values that differ from the ordinary load in dbg[0] (first mismatch: dbg[1..5]); bit 1 = take the cache-bypassing loads as THE loads
    unsigned long long* dbg; int dbg_mode;
#endif
};

#ifdef AERO_ISTFT_DEBUG
static unsigned long long* aero_istft_dbg_ptr = nullptr;
static int aero_istft_dbg_mode = 0;
// 8-byte load with system scope: misses in the vector L1 and in this XCD's L2 (the data comes from the memory side)
static __device__ __forceinline__ f32x2 aero_load_bypass(const f32x2* ptr) {
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
#endif

static inline int aero_istft_fpb(int n) { int f = 4096 / n; return f > 32 ? 32 : f; }   // frame ring (power of two)

// dynamic LDS: tw[n] | fbuf[FPB][n + 1] | sbuf[NW][n]   (f32x2)   then   wl[n_fft]   (float)
static inline size_t aero_istft_lds_bytes(int n_fft, int fpb, int nw = 4) {
    return ((size_t)(n_fft / 2) * (1 + nw + fpb) + fpb) * sizeof(f32x2) + (size_t)n_fft * sizeof(float);
}

template <int LOGN, int NW = 4>
__global__ __launch_bounds__(NW * 64) void aero_istft_kernel(AeroIstftK p) {
    constexpr int NT = NW * 64;
    const int n = LOGN ? (1 << LOGN) : (p.n_fft >> 1);
    const int n_fft = 2 * n;
    const int FPB = LOGN ? ((4096 >> LOGN) > 32 ? 32 : (4096 >> LOGN)) : p.FPB;
    const int fs = n + 1;                                  // frame stride: phase 1 writes with the frame as the fast lane index
    f32x2* tw = (f32x2*)AERO_DYN_SMEM;
    f32x2* fbuf = tw + n;                                  // [FPB][n + 1]
    f32x2* sbuf0 = fbuf + FPB * fs;
    float* wl = (float*)(sbuf0 + NW * n);
    const int lane = aero_lane(), wave = aero_uniform(aero_wave());
    // neighbouring segments of a signal share half of their frames: give each XCD (private L2) a contiguous run of (signal, segment)
    // blocks, so the second reader of a frame finds it in L2 (PMC: 259 MB fetched per launch for 66 MB of spectrogram)
    const int lin = aero_xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int sig = lin / (int)gridDim.x;
    const int o0 = (lin - sig * (int)gridDim.x) * p.SEG;
    const int i0 = o0 + n;                                  // first overlap-add index of this block
    auto div_hop = [&](int v) { return p.hsh >= 0 ? v >> p.hsh : v / p.hop; };      // v >= 0
    const int num = i0 - n_fft + p.hop;
    const int t_lo = num > 0 ? div_hop(num) : 0;
    int t_hi = div_hop(i0 + p.SEG - 1);
    if (t_hi > p.T - 1) t_hi = p.T - 1;
    const int nfr = t_hi - t_lo + 1;                        // <= FPB by construction of SEG
    aero_fft_init_twiddles(tw, n_fft);
    for (int i = threadIdx.x; i < n_fft; i += NT) wl[i] = p.window[i];
    __syncthreads();
    const f32x2* X = (const f32x2*)p.spec + (int64_t)sig * p.F * p.T + t_lo;
    // phase 1: Hermitian unpack of each frame into conj(Z), Z = E + iO  (frames are the fast index: runs of FPB frames
    // per bin).  Bins k and n-k need the same two spectrum values, so one thread loads the pair once and writes both.
    const int lf = 31 - __builtin_clz(FPB);                // FPB is a power of two
    auto unpack = [&](f32x2 xa, f32x2 xb, int k) -> f32x2 {
        const f32x2 E = (xa + xb) * 0.5f;
        const f32x2 D = (xa - xb) * 0.5f;
        const f32x2 O = aero_cmul(D, (f32x2){tw[k][0], -tw[k][1]});
        return (f32x2){E[0] - O[1], -(E[1] + O[0])};
    };
    const int npair = ((n >> 1) + 1) << lf;
#pragma unroll 4
    for (int idx = threadIdx.x; idx < npair; idx += NT) {
        const int fr = idx & (FPB - 1), k = idx >> lf;
        const bool livef = fr < nfr;
        const int t = livef ? fr : 0;
#ifdef AERO_ISTFT_DEBUG
        f32x2 xa = (p.dbg_mode & 2) ? aero_load_bypass(X + k * p.T + t) : X[k * p.T + t];
        if (p.dbg_mode & 1) {
            const f32x2 chk = aero_load_bypass(X + k * p.T + t);
            if (__float_as_uint(chk[0]) != __float_as_uint(xa[0]) || __float_as_uint(chk[1]) != __float_as_uint(xa[1])) {
                if (atomicAdd(p.dbg, 1ull) == 0) {
                    p.dbg[1] = (unsigned long long)(uintptr_t)(X + k * p.T + t);
                    p.dbg[2] = ((unsigned long long)__float_as_uint(xa[0]) << 32) | __float_as_uint(xa[1]);
                    p.dbg[3] = ((unsigned long long)__float_as_uint(chk[0]) << 32) | __float_as_uint(chk[1]);
                    p.dbg[4] = ((unsigned long long)blockIdx.x << 32) | ((unsigned long long)blockIdx.y << 16) | threadIdx.x;
                    p.dbg[5] = ((unsigned long long)k << 32) | (unsigned)fr;
                }
            }
        }
#else
        f32x2 xa = X[k * p.T + t];
#endif
        if (k == 0) {                                       // pairs with the implicit zero Nyquist bin X[n]
            xa[1] = 0.f;                                    // irfft ignores the imaginary part of DC
            const f32x2 z = unpack(xa, (f32x2){0.f, 0.f}, 0);
            fbuf[fr * fs] = livef ? z : (f32x2){0.f, 0.f};
        } else {
#ifdef AERO_ISTFT_DEBUG
            const f32x2 xq = (p.dbg_mode & 2) ? aero_load_bypass(X + (n - k) * p.T + t) : X[(n - k) * p.T + t];
            if (p.dbg_mode & 1) {
                const f32x2 chk = aero_load_bypass(X + (n - k) * p.T + t);
                if (__float_as_uint(chk[0]) != __float_as_uint(xq[0]) || __float_as_uint(chk[1]) != __float_as_uint(xq[1])) {
                    if (atomicAdd(p.dbg, 1ull) == 0) {
                        p.dbg[1] = (unsigned long long)(uintptr_t)(X + (n - k) * p.T + t);
                        p.dbg[2] = ((unsigned long long)__float_as_uint(xq[0]) << 32) | __float_as_uint(xq[1]);
                        p.dbg[3] = ((unsigned long long)__float_as_uint(chk[0]) << 32) | __float_as_uint(chk[1]);
                        p.dbg[4] = ((unsigned long long)blockIdx.x << 32) | ((unsigned long long)blockIdx.y << 16) | threadIdx.x;
                        p.dbg[5] = ((unsigned long long)(n - k) << 32) | (unsigned)fr;
                    }
                }
            }
#else
            const f32x2 xq = X[(n - k) * p.T + t];
#endif
            const f32x2 z0 = unpack(xa, (f32x2){xq[0], -xq[1]}, k);
            const f32x2 z1 = unpack(xq, (f32x2){xa[0], -xa[1]}, n - k);
            fbuf[fr * fs + k] = livef ? z0 : (f32x2){0.f, 0.f};
            fbuf[fr * fs + n - k] = livef ? z1 : (f32x2){0.f, 0.f};
        }
    }
    __syncthreads();
    // phase 2: one frame per wave per round
    const int rounds = (FPB + NW - 1) / NW;
    f32x2* sb = sbuf0 + wave * n;
#pragma unroll(LOGN ? 2 : 1)
    for (int r = 0; r < rounds; ++r) {
        const int fr = r * NW + wave;
        f32x2* a = fr < FPB ? fbuf + fr * fs : sb;          // (FPB is a multiple of 4 in practice)
        f32x2* R = aero_fft_wave<LOGN>(a, sb, n, tw);
        if (R != a) {
            for (int m = lane; m < n; m += 64) a[m] = R[m];
            aero_wave_sync();
        }
    }
    __syncthreads();
    // phase 3: output-stationary overlap-add
    const float scale = sqrtf((float)n_fft) / (float)n;
    float* ys = p.y + (int64_t)sig * p.Lout;
    for (int o = threadIdx.x; o < p.SEG; o += NT) {
        const int oo = o0 + o;
        if (oo >= p.Lout) break;
        const int i = oo + n;
        const int num2 = i - n_fft + p.hop;
        int ta = num2 > 0 ? div_hop(num2) : 0;
        if (ta < t_lo) ta = t_lo;
        int tb = div_hop(i);
        if (tb > t_hi) tb = t_hi;
        float acc = 0.f;
        for (int t = ta; t <= tb; ++t) {
            const int ni = i - t * p.hop;
            const f32x2 R = fbuf[(t - t_lo) * fs + (ni >> 1)];
            const float g = (ni & 1) ? -R[1] : R[0];
            acc += wl[ni] * g;
        }
        ys[oo] = acc * scale * p.inv_env[i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// iSTFT, second form (round 4): a block walks a RUN of frames of one signal in groups of 16 with the overlap-add ring resident in LDS.
// The first form (above) gives every 512-sample output segment its own block: 16 frames of which 8 are halo, twiddles / window / three
// serial phases per block, 4032 blocks in eight rounds -- 58-65 us for 74 MB (ablations: the per-block skeleton 24 us, the doubled frame
// FFTs 23 us; never the traffic).  Here a block owns SEGF = 64 consecutive hops of output: 72 frame transforms for 64 (an 8-frame halo
// group once per block), setup once per 64 frames, and the spectrum values of group g + 1 are requested before the FFTs of group g and
// unpacked from registers afterwards, so no phase waits for HBM.  512 blocks, two per CU: the whole launch is resident at once.
//   ring:   24 frame slots (16 new + need - 1 <= 8 old), slot = (frame - first frame of the block) mod 24, rows of n + 1 complex;
//   group:  unpack (registers -> conj(Z) rows) | barrier | request next group | FFT: wave w transforms frames 2w, 2w + 1 | barrier |
//           overlap-add of the 16 * hop samples this group completes (each sample: its `need` frames, window, 1 / envelope) | barrier.
// Geometry: n_fft 512 or 1024, hop a power of two dividing n_fft / 2, at most 8 frames over a sample (every configuration of the reference).
#define AERO_ISTFT2_SLOTS 24
#define AERO_ISTFT2_HALO 8
#define AERO_ISTFT2_SEGF 64

static inline size_t aero_istft2_lds_bytes(int n_fft) {
    const size_t n = (size_t)n_fft / 2;
    return (n + AERO_ISTFT2_SLOTS * (n + 1) + 8 * n) * sizeof(f32x2);
}

template <int LOGN, int NEED>
__global__ __launch_bounds__(512) void aero_istft2_kernel(AeroIstftK p) {
    constexpr int n = 1 << LOGN, n_fft = 2 * n, fs = n + 1, NP = n / 2;          // NP (k, n - k) pairs with k < n / 2; bin n / 2 pairs with itself
    constexpr int NIT = NP * 16 / 512;                                             // pair items per thread and 16-frame group
    f32x2* tw = (f32x2*)AERO_DYN_SMEM;
    f32x2* ring = tw + n;                                                          // [SLOTS][n + 1]
    f32x2* sbuf0 = ring + AERO_ISTFT2_SLOTS * fs;                                  // [8][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int nseg = (int)gridDim.x;
    const int lin = aero_xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int sig = lin / nseg, seg = lin - sig * nseg;
    const int hop = p.hop, hsh = p.hsh, T = p.T;
    constexpr int need = NEED;                                                     // frames over a sample = n_fft / hop
    const int h2 = n >> hsh;                                                       // frames before the first kept sample
    const int F0 = seg * AERO_ISTFT2_SEGF + h2 - AERO_ISTFT2_HALO;                 // first frame this block transforms (may be < 0: zeros)
    aero_fft_init_twiddles(tw, n_fft);
    // Rows of `spec` have a pitch of P frames and start at column toff: with P * 8 B a multiple of 128 B and toff chosen so that the
    // 16-frame groups below start on a line (aero_istft_pitch), a group's 128-byte run per bin is ONE cache line instead of a straddle of two
    // (round 5 PMC: 115 MB fetched for 74 MB with rows of 4 008 B).  P = T, toff = 0 is the plain complex64 [F][T] layout.
    const int P = p.P;
    const f32x2* X = (const f32x2*)p.spec + (int64_t)sig * p.F * P + p.toff;
    float* ys = p.y + (int64_t)sig * p.Lout;
    const float scale = sqrtf((float)n_fft) / (float)n;
    const int fr = tid & 15, k0 = tid >> 4;                                         // this thread's frame of the group, first pair index
    auto unpack = [&](f32x2 xa, f32x2 xb, int k) -> f32x2 {
        const f32x2 E = (xa + xb) * 0.5f;
        const f32x2 D = (xa - xb) * 0.5f;
        const f32x2 O = aero_cmul(D, (f32x2){tw[k][0], -tw[k][1]});
        return (f32x2){E[0] - O[1], -(E[1] + O[0])};
    };
    // ---- spectrum values of one group: per thread NIT pairs (k, n - k) of frame tg + fr, plus bin n / 2 for threads < 16.
    // Loads are unconditional (frame index clamped); frames outside [0, T) are zeroed when the values are unpacked.
    f32x2 xa[NIT], xq[NIT], xm;
    f32x2 twh[NIT];                                                     // tw[k] / 2 of this thread's pairs (filled once the twiddle table is in)
    auto request = [&](int tg) {
        int t = tg + fr;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = k0 + 32 * it;
            xa[it] = X[k * P + t];
            xq[it] = X[(k == 0 ? n - 1 : n - k) * P + t];             // (k = 0 pairs with the implicit zero Nyquist bin: value unused)
        }
        xm = X[NP * P + (tid < 16 ? t : 0)];
    };
    auto deposit = [&](int tg, int rel0) {                              // rel0: tg - F0, the group's first ring position
        const int t = tg + fr;
        const bool live = t >= 0 && t < T;
        int slot = rel0 + fr;
        slot -= slot >= AERO_ISTFT2_SLOTS ? AERO_ISTFT2_SLOTS : 0;
        slot -= slot >= AERO_ISTFT2_SLOTS ? AERO_ISTFT2_SLOTS : 0;
        f32x2* row = ring + slot * fs;
        const f32x2 zero = (f32x2){0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = k0 + 32 * it;
            f32x2 a = xa[it];
            if (k == 0) {
                a[1] = 0.f;                                             // irfft ignores the imaginary part of DC
                const f32x2 z = unpack(a, zero, 0);
                row[0] = live ? z : zero;
            } else {
                // bins k and n - k together: with S = a + conj(q), D = a - conj(q) and O = (D / 2) * conj(tw[k]) the two unpacked values are
                // (S / 2 -+ ...) of the SAME S, D, O  (tw[n - k] = -conj(tw[k]) makes the second O the conjugate of the first): 16 VALU a pair
                // against 36 for two independent unpacks; the halves are folded into the twiddle register (exact) and the closing fmas
                const f32x2 q = xq[it];
                const f32x2 w = twh[it];
                const float s0 = a[0] + q[0], s1 = a[1] - q[1], d0 = a[0] - q[0], d1 = a[1] + q[1];
                const float o0 = d0 * w[0] + d1 * w[1], o1 = d1 * w[0] - d0 * w[1];
                const f32x2 z0 = (f32x2){s0 * 0.5f - o1, -(s1 * 0.5f) - o0};
                const f32x2 z1 = (f32x2){s0 * 0.5f + o1, s1 * 0.5f - o0};
                row[k] = live ? z0 : zero;
                row[n - k] = live ? z1 : zero;
            }
        }
        if (tid < 16) {
            const f32x2 z = unpack(xm, (f32x2){xm[0], -xm[1]}, NP);
            row[NP] = live ? z : zero;
        }
    };
    AeroFft256Tw rtw;                                                   // (LOGN == 8: filled once the twiddle table is in)
    auto transform = [&](int slot) {
        slot -= slot >= AERO_ISTFT2_SLOTS ? AERO_ISTFT2_SLOTS : 0;
        slot -= slot >= AERO_ISTFT2_SLOTS ? AERO_ISTFT2_SLOTS : 0;
        f32x2* a = ring + slot * fs;
        f32x2* sb = sbuf0 + wave * n;
        if constexpr (LOGN == 8) {
            aero_fft256_regtw(a, sb, rtw);
        } else {
            f32x2* R = aero_fft_wave<LOGN>(a, sb, n, tw);
            if (R != a) {
                for (int m = lane; m < n; m += 64) a[m] = R[m];
                aero_wave_sync();
            }
        }
    };
#ifdef AERO_ISTFT_ABLATION
    const int abl = p.FPB;                                              // (tools/dbg/istft_ablation.py; p.FPB is unused by this form)
#else
    constexpr int abl = 0;
#endif
    request(F0);                                                        // the halo group: frames F0 .. F0 + 7 (lanes fr >= 8 load and drop)
    // overlap-add constants of this thread: its samples are o = tid + 512 * it of a group's 16 * hop, so r = o mod hop is FIXED (hop
    // divides 512) and with it the NEED window values r + j * hop and the sign of the conjugate ((ni & 1) = (r & 1): hop is even)
    const int r_o = tid & (hop - 1);
    float wreg[NEED];
#pragma unroll
    for (int j = 0; j < NEED; ++j) {
        const float w = p.window[r_o + ((NEED - 1 - j) << hsh)];
        wreg[j] = (r_o & 1) ? -w : w;
    }
    __syncthreads();                                                    // twiddles are in
    if constexpr (LOGN == 8) rtw = aero_fft256_twiddles(tw);
#pragma unroll
    for (int it = 0; it < NIT; ++it) twh[it] = tw[k0 + 32 * it] * 0.5f;
    constexpr int NG = AERO_ISTFT2_SEGF / 16;
    // group -1 = the halo (8 frames, one per wave), groups 0 .. NG-1 of 16 frames
    for (int g = -1; g < NG; ++g) {
        const int rel0 = g < 0 ? 0 : AERO_ISTFT2_HALO + 16 * g;
        const int tg = F0 + rel0;
        const int nf = g < 0 ? AERO_ISTFT2_HALO : 16;
        if (fr < nf && !(abl & 8)) deposit(tg, rel0 % AERO_ISTFT2_SLOTS);
        __syncthreads();
        if (g + 1 < NG && !(abl & 1)) request(F0 + AERO_ISTFT2_HALO + 16 * (g + 1));
        const int rs = rel0 % AERO_ISTFT2_SLOTS;
        if (abl & 2) {
        } else if (g < 0) {
            transform(rs + wave);
        } else {
            transform(rs + 2 * wave);
            transform(rs + 2 * wave + 1);
        }
        __syncthreads();
        if (g >= 0 && !(abl & 4)) {
            // the samples this group completes: hop index q = tg + (0 .. 15), OLA index i = q * hop + r, kept sample oo = i - n.
            // Frames outside [0, T) are zero rows (deposit), so the NEED terms are added without a range test; the row walk is a
            // running slot with one conditional wrap.  (First form: per term a division by 24, two range tests and a window read
            // from LDS -- ~25 VALU per term, 10 of the kernel's 39 us.)
            const float* ringf = (const float*)ring;
            if (hsh >= 6) {
                // hop >= 64: the 64 samples of a wave share their hop index, so the frame rows are WAVE-UNIFORM -- the slot walk runs on the
                // scalar unit and a term is one address add, one LDS read and one fma
                for (int ob = wave * 64; ob < (16 << hsh); ob += 512) {
                    const int qi = ob >> hsh;
                    const int oo = ((tg + qi) << hsh) + r_o - n;
                    const bool ok = oo >= 0 && oo < p.Lout;
                    const float ie = p.inv_env[ok ? oo + n : n];        // (issued before the LDS reads: its latency hides under them)
                    int slot = (rel0 + qi - (NEED - 1)) % AERO_ISTFT2_SLOTS;   // ring position of the oldest frame over these samples
                    const float* col = ringf + r_o;
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NEED; ++j) {
                        acc += wreg[j] * col[slot * (2 * fs) + ((NEED - 1 - j) << hsh)];
                        slot = slot + 1 == AERO_ISTFT2_SLOTS ? 0 : slot + 1;
                    }
                    if (ok) ys[oo] = acc * scale * ie;
                }
            } else {
                for (int o = tid; o < (16 << hsh); o += 512) {
                    const int qi = o >> hsh;
                    const int oo = ((tg + qi) << hsh) + r_o - n;
                    int slot = (rel0 + qi - (NEED - 1)) % AERO_ISTFT2_SLOTS;
                    int idx = slot * (2 * fs) + r_o + ((NEED - 1) << hsh);
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NEED; ++j) {
                        acc += wreg[j] * ringf[idx];
                        ++slot;
                        idx += 2 * fs - hop;
                        if (slot == AERO_ISTFT2_SLOTS) { slot = 0; idx -= AERO_ISTFT2_SLOTS * 2 * fs; }
                    }
                    if (oo >= 0 && oo < p.Lout) ys[oo] = acc * scale * p.inv_env[oo + n];
                }
            }
        }
        __syncthreads();                                                // the next group's rows overwrite frames this one still read
    }
}

static int aero_istft2_ok(int n_fft, int hop, int T) {
    const int n = n_fft / 2;
    if (n != 256 && n != 512) return 0;
    if (hop < 16 || (hop & (hop - 1)) || n % hop || n_fft / hop > AERO_ISTFT2_HALO) return 0;
    return T >= 1;
}

static int aero_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static int aero_stft_launch(const float* x, int nsig, int L, int Lp, int n_fft, int hop, const float* window, int n_bins,
                            float* spec, int T, double* stats, int sig_per_item, hipStream_t stream, const char** err) {
    if (!x || !window || !spec) { *err = "stft: null pointer"; return AERO_ERR_ARG; }
    const int n = n_fft / 2;
    if (n_fft < 16 || (1 << aero_ilog2(n_fft)) != n_fft || n > AERO_STFT_MAX_N) { *err = "stft: n_fft must be a power of two in [16,2048]"; return AERO_ERR_UNSUPPORTED; }
    if (hop < 1 || Lp < L || T != 1 + Lp / hop) { *err = "stft: inconsistent L/Lp/hop/T"; return AERO_ERR_ARG; }
    if (Lp <= n) { *err = "stft: signal shorter than the reflect pad"; return AERO_ERR_ARG; }
    if (n_bins != n && n_bins != n + 1) { *err = "stft: n_bins must be n_fft/2 or n_fft/2+1"; return AERO_ERR_ARG; }
    if (stats && sig_per_item < 1) { *err = "stft: sig_per_item"; return AERO_ERR_ARG; }
    AeroStftK p;
    p.x = x; p.window = window; p.spec = spec; p.stats = stats;
    p.nsig = nsig; p.L = L; p.Lp = Lp; p.n_fft = n_fft; p.hop = hop; p.n_bins = n_bins; p.T = T;
    p.sig_per_item = sig_per_item > 0 ? sig_per_item : 1;
    const int fpb = aero_stft_fpb(n);
    p.FPB = fpb;
    dim3 grid((unsigned)((T + fpb - 1) / fpb), (unsigned)nsig), block(256);
    // the signal span the block's frames share: staged in LDS when it fits next to the FFT buffers (160 KiB per CU)
    const int span = (fpb - 1) * hop + n_fft;
    p.span_cap = span > AERO_STFT_SPAN ? span : AERO_STFT_SPAN;
    if (aero_stft_lds_bytes(n_fft, n_bins, fpb, p.span_cap) > 160 * 1024) p.span_cap = 0;
    const size_t lds = aero_stft_lds_bytes(n_fft, n_bins, fpb, p.span_cap);
    if (lds > 160 * 1024) { *err = "stft: frame buffers exceed the LDS"; return AERO_ERR_UNSUPPORTED; }
#ifdef AERO_DBG_ZERO_LDS
    p.dbg_lds_bytes = (int)lds;
#endif
    switch (n) {                                            // compile-time sizes for the usual n_fft; run-time n otherwise
        case 64: AERO_LAUNCH_DYN(aero_stft_kernel<6>, grid, block, lds, stream, p); break;
        case 128: AERO_LAUNCH_DYN(aero_stft_kernel<7>, grid, block, lds, stream, p); break;
        case 256: AERO_LAUNCH_DYN(aero_stft_kernel<8>, grid, block, lds, stream, p); break;
        case 512: AERO_LAUNCH_DYN(aero_stft_kernel<9>, grid, block, lds, stream, p); break;
        case 1024: AERO_LAUNCH_DYN(aero_stft_kernel<10>, grid, block, lds, stream, p); break;
        default: AERO_LAUNCH_DYN(aero_stft_kernel<0>, grid, block, lds, stream, p); break;
    }
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// STFT as a GEMM for SHORT windows (Aero._spec of the low-rate input: n_fft 512 but a 128-sample window, aero.py:324-328).
// Only 128 of the 512 samples of a frame are non-zero, so  X[f, t] = sum_{k<128} A[f][k] * x[t*hop + k + c],
// A[f][k] = w[k] * n_fft^-1/2 * e^{-2 pi i f (k + win_off) / n_fft}  -- a [2*n_bins x 128] matrix times the [128 x T] frame
// matrix of a signal: 4.2 GFLOP for the whole batch, nothing for the matrix cores, where the FFT kernel above spends its time
// on LDS round trips of a 512-point transform that is three quarters zeros (75 us for a 68-MB output).
// fp32 results from the fp16 MFMA: both operands are split  v = hi + lo  (two fp16, 22 significant bits; the table is
// pre-scaled by 2^10 so that its `lo` halves stay normal) and  hi*hi + hi*lo + lo*hi  is accumulated in fp32 (the dropped
// lo*lo term is 2^-22 relative): measured 3e-7 rel-L2 against the fp64 oracle, like the FFT.
// Block: 512 threads, 128 rows (= 64 bins, re/im interleaved) x 128 frames of one signal; wave tile 32 x 64.  The block's
// slice of the table (both halves, all four 32-column chunks: 64 KiB) is copied into LDS in one go (direct global->LDS copies,
// image already in tile order: one L2 latency; a double-buffered chunk pipeline paid four and ran at 35 us); the frames are
// rows of ONE shared span of the signal in LDS (hi / lo), read at offset frame*hop.
// HBM-bound by its 8-byte-per-bin output; the table (256 KB) is L2 traffic.
#define AERO_DFT_K 128
#define AERO_DFT_SPAN 2176          /* samples of signal the 128 frames of a block may span: hop <= 16 (74 KiB of LDS: two blocks per CU) */
struct AeroStftDftK {
    const float* x; const h16* table; float* spec; double* stats;
    int nsig, L, Lp, n_fft, hop, win_off, T, sig_per_item;
    h16* xn; float* mean_std;                                    // MODE 2: the normalised fp16 spectrogram and (mean, std) per item
};

// table image: fp16 [part hi|lo][chunk 0..3][n_fft rows (2f + {re, im})][32], rows in tile order (aero_tile_off); value * 2^10
__global__ __launch_bounds__(256) void aero_stft_dft_table_kernel(const float* window, int n_fft, int win_off, h16* table) {
    const int total = 4 * n_fft * 32;
    const double scale = 1024.0 / sqrt((double)n_fft);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int kc = idx / (n_fft * 32), rem = idx - kc * (n_fft * 32);
        const int r = rem >> 5, j = rem & 31;
        const int q = (j >> 3) ^ ((0 - (r >> 2)) & 3);            // the slot stored at position j >> 3 of row r (aero_tile_off)
        const int k = kc * 32 + q * 8 + (j & 7);
        const int f = r >> 1, n = win_off + k;
        const double w = n < n_fft ? (double)window[n] : 0.0;
        const long ph = ((long)f * n) % n_fft;                    // exact phase index
        const double ang = -2.0 * 3.14159265358979323846 * (double)ph / (double)n_fft;
        const double v = w * scale * ((r & 1) ? sin(ang) : cos(ang));
        const h16 hi = (h16)(float)v;
        const h16 lo = (h16)(float)(v - (double)(float)hi);
        table[(size_t)kc * n_fft * 32 + rem] = hi;
        table[(size_t)(4 + kc) * n_fft * 32 + rem] = lo;
    }
}

// MODE 0: fp32 spectrogram + per-item sums (the form above).  Round 6: the forward does not need the fp32 spectrogram
// unless the caller asks for it (`return_lr_spec`) -- what the U-Net reads is the per-item NORMALISED fp16 tensor (aero.py:459-464), and the
// pair "DFT writes 66 MB fp32 + sums; aero_spec_normalize re-reads them and writes 33 MB" moved 166 MB for a 33-MB result.  The DFT is 4 GF:
// cheaper to run twice than to round-trip its output.  MODE 1: the sums only (nothing stored; the same code, the same summation order:
// mean / std are BIT-identical to MODE 0's).  MODE 2: recompute and store (v - mean) / (1e-5 + std) as fp16, the arithmetic of
// aero_spec_normalize_kernel on the same fp32 values -- the output equals the unfused pair's bit for bit.
template <int MODE>
__global__ __launch_bounds__(512) void aero_stft_dft_kernel(AeroStftDftK p) {
    __shared__ AERO_LDS_ALIGN h16 As[4][2][128 * 32];             // [chunk][hi | lo][128 rows][32]: the block's whole table slice
    __shared__ AERO_LDS_ALIGN h16 xh[AERO_DFT_SPAN + 8], xl[AERO_DFT_SPAN + 8];
    __shared__ double red[2][8];
    __shared__ float nrm[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int kg = lane >> 4, col = lane & 15;
    const int wr = wave >> 1, wc = wave & 1;                     // wave tile: rows wr*32 .. +32, frames wc*64 .. +64
    const int sig = blockIdx.y, quarter = blockIdx.z;
    const int n_fft = p.n_fft, hop = p.hop, n_bins = n_fft >> 1;
    const float* xs = p.x + (int64_t)sig * p.L;
    // the block's table slice, all four chunks at once (one L2 latency instead of four): 4 x 2 x 128 rows x 4 slots = 4096
    // 16-byte units, 8 per thread; the image is already in tile order, so the copy is linear.  The block then WALKS the 128-frame
    // tiles of its signal (blockIdx.x, + gridDim.x, ...) with the slice resident: the 64-KiB fill and its latency are paid once
    // per block instead of once per 128 x 128 output tile (it equalled the output traffic: 64 MB through L2 for 66 MB written).
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int W = wave + 8 * u;                              // 64-unit piece: (chunk, part) = W >> 3, rows (W & 7) * 16 .. +16
        const int cp = W >> 3, kc = cp >> 1, part = cp & 1;
        const h16* src = p.table + ((size_t)(part * 4 + kc) * n_fft + quarter * 128) * 32 + ((W & 7) * 64 + lane) * 8;
        aero_glds16(src, &As[kc][part][0] + (W & 7) * 512);
    }
    const int span = 127 * hop + AERO_DFT_K;
    const int ntile = (p.T + 127) >> 7;
    // Per-item statistics (sum, sum of squares of the spectrogram: aero.py:462-464): fp32 per thread over the tiles its block walks.  WHICH
    // tiles a block walks must therefore not depend on the batch size -- see the launcher: gridDim.x is a function of T only.
    float s = 0.f, ss = 0.f;
    float* out = MODE == 0 ? p.spec + (int64_t)sig * n_bins * p.T * 2 : nullptr;
    h16* outn = MODE == 2 ? p.xn + (int64_t)sig * n_bins * p.T * 2 : nullptr;
    if (MODE == 2 && tid == 0) {                                 // (aero_spec_normalize_kernel's arithmetic, once per block)
        const int item = sig / p.sig_per_item;
        const double N = (double)p.sig_per_item * n_bins * p.T * 2;
        const double S = p.stats[2 * item], SS = p.stats[2 * item + 1];
        const double mean = S / N;
        double var = (SS - S * S / N) / (N - 1.0);
        if (var < 0) var = 0;
        const float fm = (float)mean, fs = (float)sqrt(var);
        nrm[0] = fm;
        nrm[1] = 1.0f / (1e-5f + fs);
        if (blockIdx.x == 0 && quarter == 0 && sig % p.sig_per_item == 0) { p.mean_std[2 * item] = fm; p.mean_std[2 * item + 1] = fs; }
    }
    float fm = 0.f, inv = 0.f;
    // the span of the (hop-padded, reflect-padded) signal a tile's 128 frames read, one load batch per tile, REQUESTED a tile ahead (their
    // latency runs under the MFMAs and stores of the tile before); the barriers between tiles order LDS only (aero_lds_barrier) -- a
    // __syncthreads() there also waited for the tile's output stores to be acknowledged; only the first one covers the table copies.
    // (hipcc still waits vmcnt(0) where the prefetched values are first used, behind the stores: the stores sit in branches (t < T), so
    // it cannot count them.)
    constexpr int NV = (AERO_DFT_SPAN + 511) / 512;
    float v[NV];
    auto span_index = [&](int tl, int u, bool& ok) {
        const int j = tid + u * 512;
        int xi = tl * 128 * hop + j + p.win_off - (n_fft >> 1);
        if (xi < 0) xi = -xi;
        if (xi >= p.Lp) xi = 2 * (p.Lp - 1) - xi;
        ok = j < span && xi >= 0 && xi < p.L;
        return ok ? xi : 0;
    };
    auto fetch_span = [&](int tl) {                              // unconditional loads, used raw: the zero padding is applied where the
#pragma unroll                                                   // values are consumed (a select right behind a load is a wait right behind it)
        for (int u = 0; u < NV; ++u) {
            bool ok;
            v[u] = xs[span_index(tl, u, ok)];
        }
    };
    if ((int)blockIdx.x < ntile) fetch_span((int)blockIdx.x);
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int t0 = tile * 128;
    {
        if (tile != (int)blockIdx.x) aero_lds_barrier();         // the previous tile's fragment reads of xh / xl are done
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int j = tid + u * 512;
            bool ok;
            (void)span_index(tile, u, ok);
            const float x = ok ? v[u] : 0.f;
            if (j < span) {
                const h16 hi = (h16)x;
                xh[j] = hi;
                xl[j] = (h16)(x - (float)hi);
            }
        }
        if (tile + (int)gridDim.x < ntile) fetch_span(tile + (int)gridDim.x);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tile == (int)blockIdx.x) {
        __syncthreads();                                         // first tile: the table slice's direct copies have landed (vmcnt drained) ...
        if (MODE == 2) { fm = nrm[0]; inv = nrm[1]; }
    } else aero_lds_barrier();                                   // ... later tiles: the span only
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        h16x8 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = aero_tile_off(wr * 32 + i * 16 + col, kg);
            ah[i] = *(const h16x8*)&As[kc][0][off];
            al[i] = *(const h16x8*)&As[kc][1][off];
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int base = (wc * 64 + n * 16 + col) * hop + kc * 32 + kg * 8;
            const h16x8 bh = *(const h16x8*)&xh[base];
            const h16x8 bl = *(const h16x8*)&xl[base];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][n], 0, 0, 0);
            }
        }
    }
    // rows m = quarter*128 + wr*32 + i*16 + kg*4 + r = 2f + {re, im}: registers (0,1) and (2,3) are two complex bins
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int t = t0 + wc * 64 + n * 16 + col;
        if (t >= p.T) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = (quarter * 128 + wr * 32 + i * 16 + kg * 4) >> 1;
            const f32x4 v = acc[i][n] * (1.0f / 1024.0f);
            if (MODE == 0) {
                *(f32x2*)(out + ((int64_t)f * p.T + t) * 2) = (f32x2){v[0], v[1]};
                *(f32x2*)(out + ((int64_t)(f + 1) * p.T + t) * 2) = (f32x2){v[2], v[3]};
            }
            if (MODE == 2) {
                *(h16x2*)(outn + ((int64_t)f * p.T + t) * 2) = (h16x2){(h16)((v[0] - fm) * inv), (h16)((v[1] - fm) * inv)};
                *(h16x2*)(outn + ((int64_t)(f + 1) * p.T + t) * 2) = (h16x2){(h16)((v[2] - fm) * inv), (h16)((v[3] - fm) * inv)};
            } else {
                s += (v[0] + v[1]) + (v[2] + v[3]);
                ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
        }
    }
    }
    if (MODE != 2 && p.stats) {
        const double ds = aero_wave_sum((double)s), dss = aero_wave_sum((double)ss);
        if (lane == 0) { red[0][wave] = ds; red[1][wave] = dss; }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < 8; ++w) { a += red[0][w]; b += red[1][w]; }
            double* st = p.stats + 2 * (int64_t)(sig / p.sig_per_item);
            atomicAdd(st, a);
            atomicAdd(st + 1, b);
        }
    }
}

static size_t aero_stft_dft_tbytes(int n_fft) { return (size_t)2 * 4 * n_fft * 32 * sizeof(h16); }

static int aero_stft_dft_ok(int n_fft, int hop, int win_off) {
    return n_fft >= 128 && n_fft % 128 == 0 && n_fft <= 4096 && hop >= 8 && hop % 8 == 0 && 127 * hop + AERO_DFT_K <= AERO_DFT_SPAN &&
           win_off >= 0 && win_off + 1 <= n_fft;
}

static int aero_stft_dft_table_launch(const float* window, int n_fft, int win_off, void* table, hipStream_t stream, const char** err) {
    if (!window || !table) { *err = "stft_dft_table: null pointer"; return AERO_ERR_ARG; }
    if (n_fft < 256 || n_fft % 256 || n_fft > 4096 || win_off < 0 || win_off >= n_fft || ((uintptr_t)table & 15)) { *err = "stft_dft_table: bad geometry"; return AERO_ERR_ARG; }
    AERO_LAUNCH(aero_stft_dft_table_kernel, dim3((unsigned)((4 * n_fft * 32 + 255) / 256)), dim3(256), stream, window, n_fft, win_off, (h16*)table);
    return AERO_OK;
}

// spec != NULL: MODE 0 (xn / mean_std unused).  spec == NULL: the fused form -- MODE 1 then MODE 2 on the same stream (stats, xn, mean_std required)
static int aero_stft_dft_launch(const float* x, int nsig, int L, int Lp, int n_fft, int hop, int win_off, const void* table, float* spec,
                                int T, double* stats, int sig_per_item, hipStream_t stream, const char** err, void* xn = nullptr,
                                float* mean_std = nullptr) {
    if (!x || !table || (!spec && !(xn && mean_std && stats))) { *err = "stft_dft: null pointer"; return AERO_ERR_ARG; }
    if (!spec && ((uintptr_t)xn & 3)) { *err = "stft_dft: misaligned output"; return AERO_ERR_ARG; }
    if (!aero_stft_dft_ok(n_fft, hop, win_off)) { *err = "stft_dft: n_fft % 256, hop % 8, hop <= 16 and a window of <= 128 samples are required"; return AERO_ERR_UNSUPPORTED; }
    if (Lp < L || T != 1 + Lp / hop || Lp <= n_fft / 2) { *err = "stft_dft: inconsistent L/Lp/hop/T"; return AERO_ERR_ARG; }
    if ((stats && sig_per_item < 1) || ((uintptr_t)table & 15) || ((uintptr_t)spec & 7)) { *err = "stft_dft: bad arguments"; return AERO_ERR_ARG; }
    if (nsig > 65535) { *err = "stft_dft: too many signals for one launch"; return AERO_ERR_ARG; }
    AeroStftDftK p;
    p.x = x; p.table = (const h16*)table; p.spec = spec; p.stats = stats;
    p.nsig = nsig; p.L = L; p.Lp = Lp; p.n_fft = n_fft; p.hop = hop; p.win_off = win_off; p.T = T;
    p.sig_per_item = sig_per_item > 0 ? sig_per_item : 1;
    p.xn = (h16*)xn; p.mean_std = mean_std;
    // blocks walk the time tiles of their (signal, table quarter) with the table slice resident: TWO blocks per (signal, quarter), each
    // walking every second tile (B = 64: 512 blocks = the chip twice over, two 74-KiB blocks fit a CU; measured 26.0 / 23.3 / 28.4 / 26.5 us
    // with 1 / 2 / 3 / 4).  The count must NOT follow the batch size (until round 5 it was "enough blocks for 512": 2 at B = 64, 4 at
    // B = 32): a thread's fp32 partial sums of the per-item statistics run over the tiles its block walks, so mean / std of a clip -- hence
    // its whole output -- differed in the last bit between a batch of 64 and the same clips in two batches of 32 (tools/dbg/half_vs_full.py,
    // profiles/r05_half_vs_full*.txt).
    const int ntile = (T + 127) / 128;
    static int tpb_env = -1;
    if (tpb_env < 0) { const char* e = getenv("AERO_STFT_DFT_BLOCKS"); tpb_env = e ? atoi(e) : 0; }
    int gx = tpb_env > 0 ? tpb_env : 2;
    if (gx > ntile) gx = ntile;
    dim3 grid((unsigned)gx, (unsigned)nsig, (unsigned)(n_fft / 128)), block(512);
    if (spec) {
        AERO_LAUNCH(aero_stft_dft_kernel<0>, grid, block, stream, p);
    } else {
        AERO_LAUNCH(aero_stft_dft_kernel<1>, grid, block, stream, p);
        AERO_LAUNCH(aero_stft_dft_kernel<2>, grid, block, stream, p);
    }
    return AERO_OK;
}

static int aero_spec_normalize_launch(const float* spec, int nitems, int64_t n_per_item, const double* stats, void* xn,
                                      float* mean_std, hipStream_t stream, const char** err) {
    if (!spec || !stats || !xn || !mean_std) { *err = "spec_normalize: null pointer"; return AERO_ERR_ARG; }
    if (n_per_item < 2 || (n_per_item & 1)) { *err = "spec_normalize: n_per_item must be even"; return AERO_ERR_ARG; }
    // ~4 trips of four values per thread, and enough blocks over all items to fill the chip a few times
    int64_t nb = (n_per_item / 16 + 255) / 256;
    const int64_t want = (2048 + nitems - 1) / nitems;
    if (nb < want) nb = want;
    if (nb > (n_per_item / 2 + 255) / 256) nb = (n_per_item / 2 + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    dim3 grid((unsigned)nb, (unsigned)nitems), block(256);
    AERO_LAUNCH(aero_spec_normalize_kernel, grid, block, stream, spec, n_per_item, stats, (h16*)xn, mean_std);
    return AERO_OK;
}

// the layout aero_istft2_kernel reads with whole cache lines: pitch (frames per row) and the column of frame 0; (T, 0) when the geometry
// runs on the other kernel
static void aero_istft_pitch_for(int n_fft, int hop, int T, int* pitch, int* toff) {
    *pitch = T; *toff = 0;
    if (!aero_istft2_ok(n_fft, hop, T)) return;
    const int h2 = (n_fft / 2) / hop;                          // a block's 16-frame groups start at frame seg * 64 + h2 + 16 g
    *toff = (16 - (h2 & 15)) & 15;
    *pitch = (*toff + T + 15) / 16 * 16;
}

static int aero_istft_launch(const float* spec, int nsig, int F, int T, int n_fft, int hop, const float* window,
                             const float* inv_env, float* y, int Lout, hipStream_t stream, const char** err, int pitch = 0, int toff = 0) {
    if (!spec || !window || !inv_env || !y) { *err = "istft: null pointer"; return AERO_ERR_ARG; }
    if (pitch == 0) pitch = T;
    if (toff < 0 || pitch < toff + T) { *err = "istft: pitch < toff + T"; return AERO_ERR_ARG; }
    const int n = n_fft / 2;
    if (n_fft < 16 || (1 << aero_ilog2(n_fft)) != n_fft || n > AERO_FFT_MAX_N) { *err = "istft: n_fft must be a power of two in [16,1024]"; return AERO_ERR_UNSUPPORTED; }
    if (F != n) { *err = "istft: F must be n_fft/2"; return AERO_ERR_ARG; }
    if (hop < 1 || hop > n_fft || T < 1 || Lout < 1 || Lout > hop * (T - 1)) { *err = "istft: bad hop/T/Lout"; return AERO_ERR_ARG; }
    AeroIstftK p;
    p.spec = spec; p.window = window; p.inv_env = inv_env; p.y = y;
    p.nsig = nsig; p.F = F; p.T = T; p.n_fft = n_fft; p.hop = hop; p.Lout = Lout; p.P = pitch; p.toff = toff;
    static const int v2 = [] { const char* e = getenv("AERO_ISTFT_V2"); return e ? atoi(e) : 1; }();
    if (v2 && aero_istft2_ok(n_fft, hop, T)) {
        p.hsh = aero_ilog2(hop);
        p.FPB = 0;
#ifdef AERO_ISTFT_ABLATION
        { const char* e = getenv("AERO_ISTFT_ABL"); p.FPB = e ? atoi(e) : 0; }
#endif
        p.SEG = 0;
        const int segs = hop * AERO_ISTFT2_SEGF;
        dim3 grid2((unsigned)((Lout + segs - 1) / segs), (unsigned)nsig);
        const size_t lds2 = aero_istft2_lds_bytes(n_fft);
        const int need2 = n_fft / hop;
#define AERO_ISTFT2_GO(LOGN_, NEED_) AERO_LAUNCH_DYN((aero_istft2_kernel<LOGN_, NEED_>), grid2, dim3(512), lds2, stream, p)
        if (n == 256) { if (need2 == 8) AERO_ISTFT2_GO(8, 8); else if (need2 == 4) AERO_ISTFT2_GO(8, 4); else AERO_ISTFT2_GO(8, 2); }
        else { if (need2 == 8) AERO_ISTFT2_GO(9, 8); else if (need2 == 4) AERO_ISTFT2_GO(9, 4); else AERO_ISTFT2_GO(9, 2); }
#undef AERO_ISTFT2_GO
        return AERO_OK;
    }
    if (pitch != T || toff != 0) { *err = "istft: a pitched spectrogram needs the geometry aero_istft_pitch reports"; return AERO_ERR_UNSUPPORTED; }
    const int fpb = aero_istft_fpb(n);
    const int need = (n_fft + hop - 1) / hop;              // frames overlapping one sample
    if (fpb <= need) { *err = "istft: hop too small for the LDS frame ring"; return AERO_ERR_UNSUPPORTED; }
    p.FPB = fpb;
    p.SEG = (fpb - need) * hop;
    p.hsh = (hop & (hop - 1)) == 0 ? aero_ilog2(hop) : -1;
#ifdef AERO_ISTFT_DEBUG
    p.dbg = aero_istft_dbg_ptr;
    p.dbg_mode = aero_istft_dbg_ptr ? aero_istft_dbg_mode : (aero_istft_dbg_mode & 2);
#endif
    dim3 grid((unsigned)((Lout + p.SEG - 1) / p.SEG), (unsigned)nsig), block(256);
    // eight waves per block for the two common sizes: two frames per wave instead of four, twice the lanes in the unpack and overlap-add
    // phases (78.9 -> 61.5 us at B = 64 with the XCD-aware block order; sixteen waves: 70 us).  Ablation builds of round 3 at eight
    // waves: empty kernel 7 us, no loads / FFT / overlap-add 24 us, no FFT 41 us, no loads 58 us, all 64 us -- the frame FFTs (23 us) and
    // the per-block skeleton (window, unpack, barriers: 17 us) are what is left, not the bytes it fetches.
    static const int waves = [] { const char* e = getenv("AERO_ISTFT_WAVES"); return e ? atoi(e) : 8; }();
    if (waves == 8 && (n == 256 || n == 512)) {
        const size_t lds8 = aero_istft_lds_bytes(n_fft, fpb, 8);
        if (n == 256) AERO_LAUNCH_DYN((aero_istft_kernel<8, 8>), grid, dim3(512), lds8, stream, p);
        else AERO_LAUNCH_DYN((aero_istft_kernel<9, 8>), grid, dim3(512), lds8, stream, p);
        return AERO_OK;
    }
    if (waves == 16 && n == 256) {
        AERO_LAUNCH_DYN((aero_istft_kernel<8, 16>), grid, dim3(1024), aero_istft_lds_bytes(n_fft, fpb, 16), stream, p);
        return AERO_OK;
    }
    const size_t lds = aero_istft_lds_bytes(n_fft, fpb);
    switch (n) {
        case 64: AERO_LAUNCH_DYN(aero_istft_kernel<6>, grid, block, lds, stream, p); break;
        case 128: AERO_LAUNCH_DYN(aero_istft_kernel<7>, grid, block, lds, stream, p); break;
        case 256: AERO_LAUNCH_DYN(aero_istft_kernel<8>, grid, block, lds, stream, p); break;
        case 512: AERO_LAUNCH_DYN(aero_istft_kernel<9>, grid, block, lds, stream, p); break;
        default: AERO_LAUNCH_DYN(aero_istft_kernel<0>, grid, block, lds, stream, p); break;
    }
    return AERO_OK;
}
