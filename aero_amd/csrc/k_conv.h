// k_conv.h -- implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_f16).
//
// One kernel family serves every dense contraction of the path (see include/aero_hip.h,
// aero_conv_fwd).  GEMM view per output row (b, fo):
//     D[m][n] = sum_k A[m][k] * Bm[k][n],   m = output channel, n = time step, k = (tap, channel)
// A = packed weights [Mpad][ntaps*Cp] (K contiguous), Bm = activations gathered from the source
// row(s) fi(fo, tap) shifted by dt(tap) -- a contiguous [T, C] slab per tap, so HBM reads are
// coalesced along time and every K-chunk of 32 channels is 4 x 16-byte loads per position.
// A block owns BM output channels x 128 time steps of ONE (b, fo) row: padding validity, the
// transposed-conv weight set and the source row are block-uniform; invalid taps and the
// structurally-zero source of the first decoder are skipped entirely.
//
// Roofline: MFMA (fp16 in, fp32 accumulate).  Algorithmic FLOPs = 2*M*ntaps*(C0+C1) per output
// position.  v1 pipeline: register-staged global->LDS, one LDS buffer,
// next K-chunk prefetched into VGPRs while the current one feeds the MFMAs.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "aero_common.h"
#include "k_conv_common.h"


// Shared epilogue of the tiled kernels: D[m = (lane>>4)*4 + r][n = lane&15] per fragment.
//   v = acc + bias
//   stat_mode 1/2: GroupNorm statistics of v are accumulated here (fp32 per thread, fp64 shuffles, one fp64 atomic
//                  pair per (wave, group)) -- the separate statistics pass over the stored tensor disappears;
//                  mode 2 stores nothing (first half of a recompute pair whose wide intermediate never reaches HBM)
//   stat_mode 3:   v = (v - mean) * rstd * gamma + beta from previously accumulated statistics
//   then act (GLU pairs rows 2u,2u+1; optional LayerScale), residual, frequency embedding, per-item affine, store.
// Staged form (fp16 output, 8-channel aligned): the tile is transposed through LDS in two passes of 64 positions so
// that every global store (and residual load) is a full 16-byte channel vector (256-byte runs per position).
// Lean epilogue for the common case (fp16 output through the LDS transpose, no statistics / frequency embedding /
// per-item affine): the activation is a TEMPLATE parameter and nothing is decided per element.  PMC on the pointwise
// convs showed why this matters: the generic epilogue below executes ~3600 scalar+vector instructions per wave for
// 48 MFMAs (every runtime option re-evaluated inside the unrolled fragment loops) and those kernels ran issue-bound at
// 1.5 TB/s.  Rows >= M carry zero weights and are masked at the copy-out, so no fragment is skipped here.
template <int MF, int WM, int NWV, int ACT, int SM = 0>
static __device__ __forceinline__ void aero_conv_epilogue_fast(const AeroConvK& p, f32x4 (&acc)[MF][8 / (NWV / WM)], h16* Cs, int b,
                                                               int fo, int fdst, int m0, int t0) {
    constexpr int WN = NWV / WM;
    constexpr int NF = 8 / WN;
    constexpr int BM = 16 * MF * WM;
    constexpr int CS = BM + 8;
    constexpr int NH = NF / 2, PH = NH * 16;
    constexpr bool GLU = ACT == AERO_ACT_GLU;
    constexpr int BMo = GLU ? BM / 2 : BM;
    constexpr int NVEC = BMo / 8;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int T = d.T, M = d.M;
    const int Mout = GLU ? (M >> 1) : M;
    const int m0o = GLU ? (m0 >> 1) : m0;
    // SM (statistics mode of aero_hip.h): 0 none; 1 accumulate GroupNorm sums of the conv output while storing it;
    // 2 accumulate only (nothing is stored: first half of a recompute pair); 3 normalise with previously accumulated sums:
    // v = (acc + bias - mean) * rstd * gamma + beta folds into ONE FMA per value (av, bv) before the activation.
    h16* dst16 = (h16*)d.dst;
    h16* drow = dst16 + (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + m0o;
    const h16* rrow = d.res ? (const h16*)d.res + (int64_t)b * d.r_b + (int64_t)fdst * d.r_f + m0o : nullptr;
    const float* prow = d.post_add ? d.post_add + (int64_t)fo * Mout + m0o : nullptr;      // frequency embedding row
    // copy-out item `it` of half-tile `pass` of this thread: staged position pc, 8-channel vector cv, time step t
    constexpr int NIT = (64 * NVEC + NWV * 64 - 1) / (NWV * 64);
    auto locate = [&](int pass, int it, int& pc, int& cv, int& t) -> bool {
        const int idx = tid + it * NWV * 64;
        pc = idx / NVEC;
        cv = idx - pc * NVEC;
        const int wq = pc / PH, rr = pc - wq * PH;
        t = t0 + (wq * NF + pass * NH + (rr >> 4)) * 16 + (rr & 15);
        return idx < 64 * NVEC && t < T && m0o + cv * 8 < Mout;
    };
    // The residual tile is fetched HERE, together with the coefficient loads below, not inside the copy-out: a short-K
    // block (the DConv tail: one K chunk, 24 KiB of output) otherwise pays three more dependent memory latencies after
    // its K loop -- coefficients, residual of half 0, residual of half 1.  (4-wave tiles only: the 8-wave tiles run at
    // 128 registers and have no room for it.)
    constexpr bool PRE = (NWV == 4) && SM != 2;
    h16x8 rpre[PRE ? 2 : 1][PRE ? NIT : 1];
    if constexpr (PRE) {
        if (rrow) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    int pc, cv, t;
                    if (locate(pass, it, pc, cv, t)) rpre[pass][it] = *(const h16x8*)(rrow + (int64_t)t * d.r_t + cv * 8);
                }
        }
    }
    float bv[MF][4], av[MF][4], ls[MF][2];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int mbase = m0 + (wm * MF + i) * 16 + (lane >> 4) * 4;
        float mean = 0.f, rstd = 1.f;
        if constexpr (SM == 3) {
            const int gs3 = M / d.stat_G;
            int grp = (m0 + (wm * MF + i) * 16) / gs3;
            grp = grp < d.stat_G ? grp : d.stat_G - 1;
            const double* sp = d.stats + ((int64_t)(d.stat_per_row ? b * d.Fout + fo : b) * d.stat_G + grp) * 2;
            const double inv = 1.0 / d.stat_count;
            const double mu = sp[0] * inv;
            double var = sp[1] * inv - mu * mu;
            if (var < 0) var = 0;
            const float vf = (float)var + d.stat_eps;
            float rs = aero_rsqrt(vf);
            rs = rs * (1.5f - 0.5f * vf * rs * rs);
            mean = (float)mu;
            rstd = rs;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mi = mbase + r < M ? mbase + r : M - 1;
            const float bias = d.bias ? d.bias[mi] : 0.f;
            if constexpr (SM == 3) {
                const float gm = d.gamma ? d.gamma[mi] * rstd : rstd;
                av[i][r] = gm;
                bv[i][r] = (bias - mean) * gm + (d.gamma ? d.beta[mi] : 0.f);
            } else {
                av[i][r] = 1.f;
                bv[i][r] = bias;
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ci = (mbase >> 1) + r < Mout ? (mbase >> 1) + r : Mout - 1;
            ls[i][r] = (GLU && d.layer_scale) ? d.layer_scale[ci] : 1.f;
        }
    }
    // stat_mode 1: GroupNorm statistics of the conv output (bias included, before the activation) ride along: two FMAs per
    // value on a VALU that idles under the MFMAs, one fp64 atomic pair per (wave, group) -- the separate read-only pass
    // over the stored tensor (0.8 ms per forward) disappears
    float st1[MF], st2[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) st1[i] = st2[i] = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int cl = (wm * MF + i) * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int nn = 0; nn < NH; ++nn) {
                const int n = pass * NH + nn;
                const int pc = wn * PH + nn * 16 + (lane & 15);
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (SM == 3) ? acc[i][n][r] * av[i][r] + bv[i][r] : acc[i][n][r] + bv[i][r];
                if constexpr (SM == 1 || SM == 2) {
                    const bool tin = t0 + (wn * NF + n) * 16 + (lane & 15) < T;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = (tin && m0 + cl + r < M) ? o[r] : 0.f;
                        st1[i] += v;
                        st2[i] += v * v;
                    }
                }
                if constexpr (SM == 2) continue;                 // statistics only
                if (GLU) {
                    const float g0 = o[0] * aero_sigmoid(o[1]) * ls[i][0];
                    const float g1 = o[2] * aero_sigmoid(o[3]) * ls[i][1];
                    *(h16x2*)&Cs[pc * CS + (cl >> 1)] = (h16x2){(h16)g0, (h16)g1};
                } else {
                    if (ACT == AERO_ACT_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
                    } else if (ACT == AERO_ACT_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = aero_gelu(o[r]);
                    }
                    *(h16x4*)&Cs[pc * CS + cl] = (h16x4){(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
                }
                // 8-wave tiles live at 128 registers: hipcc otherwise SINKS the statistics FMAs of a pass below the copy-out (they are not
                // needed before the end), keeps every `o` alive in between and spills three of them -- a private segment, and a
                // launch of a kernel that has one was measured ~40 us slower whenever the stream's scratch had to grow
                if constexpr ((SM == 1 || SM == 2) && NWV == 8) aero_pin(st1[i], st2[i]);
            }
        }
        if constexpr (SM == 2) continue;
        aero_lds_barrier();                                 // (LDS only: no wait for the other half's global stores)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int pc, cv, t;
            if (locate(pass, it, pc, cv, t)) {
                h16x8 v = *(const h16x8*)&Cs[pc * CS + cv * 8];
                if (rrow) {
                    h16x8 r8;
                    if constexpr (PRE) r8 = rpre[pass][it];
                    else r8 = *(const h16x8*)(rrow + (int64_t)t * d.r_t + cv * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] + (float)r8[e]);
                }
                if (prow) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] + prow[cv * 8 + e]);
                }
                if (d.scatter_M) {                              // transposed conv from the input side: channel block -> row
                    const int cg = m0o + cv * 8;
                    const int r = cg / d.scatter_M;
                    const int frow = fo * d.scatter_stride + r - d.scatter_off;
                    if (frow >= 0 && frow < d.scatter_F)
                        *(h16x8*)(dst16 + (int64_t)b * d.d_b + (int64_t)frow * d.d_f + (int64_t)t * d.d_t + (cg - r * d.scatter_M)) = v;
                    continue;
                }
                *(h16x8*)(drow + (int64_t)t * d.d_t + cv * 8) = v;
            }
        }
        aero_lds_barrier();
    }
    if constexpr (SM == 1 || SM == 2) {
        const int gs = M / d.stat_G;                              // rows per statistics group (16-row aligned, or one group)
        const int64_t sitem = (int64_t)(d.stat_per_row ? b * d.Fout + fo : b) * d.stat_G;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int base_i = m0 + (wm * MF + i) * 16;
            const int grp = base_i / gs;
            if (i + 1 < MF && (base_i + 16) / gs == grp && base_i + 16 < M) {     // same group as the next fragment: fold
                st1[i + 1] += st1[i];
                st2[i + 1] += st2[i];
                continue;
            }
            if (base_i >= M) continue;
            const double a = aero_wave_sum((double)st1[i]);
            const double c = aero_wave_sum((double)st2[i]);
            if (lane == 0) {
                atomicAdd(d.stats + (sitem + grp) * 2, a);
                atomicAdd(d.stats + (sitem + grp) * 2 + 1, c);
            }
        }
    }
}

template <int MF, int WM, bool STATS, int NWV = 4>
static __device__ __forceinline__ void aero_conv_epilogue_generic(const AeroConvK& p, f32x4 (&acc)[MF][8 / (NWV / WM)], h16* Cs, int b, int fo,
                                                          int fdst, int m0, int t0) {
    constexpr int WN = NWV / WM;
    constexpr int NF = 8 / WN;
    constexpr int BM = 16 * MF * WM;
    constexpr int CS = BM + 8;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int T = d.T;
    const int M = d.M;
    const bool glu = d.act == AERO_ACT_GLU;
    const int Mout = glu ? (M >> 1) : M;
    const int nout = glu ? 2 : 4;
    h16* dst16 = (h16*)d.dst;
    float* dst32 = (float*)d.dst;
    const h16* res = (const h16*)d.res;
    const bool res_in_copy = p.staged && res != nullptr;       // residual added with coalesced 16-byte loads
    const float bsc = d.batch_scale ? d.batch_scale[b] : 1.f;
    const float bsh = d.batch_scale ? d.batch_shift[b] : 0.f;
    const int smode = STATS ? d.stat_mode : 0;               // compile-time off: no register/code cost for plain convs
    const int gs = smode ? M / d.stat_G : M;                    // rows per statistics group
    const int64_t sitem = smode ? (int64_t)(d.stat_per_row ? b * d.Fout + fo : b) * d.stat_G : 0;
    float st_mean[MF], st_rstd[MF], st_s1[MF], st_s2[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        st_mean[i] = 0.f; st_rstd[i] = 1.f; st_s1[i] = 0.f; st_s2[i] = 0.f;
        if (smode == 3) {
            const int grp = (m0 + (wm * MF + i) * 16) / gs;
            if (grp < d.stat_G) {
                const double* sp = d.stats + (sitem + grp) * 2;
                const double mu = sp[0] / d.stat_count;
                double var = sp[1] / d.stat_count - mu * mu;
                if (var < 0) var = 0;
                st_mean[i] = (float)mu;
                st_rstd[i] = (float)(1.0 / sqrt(var + (double)d.stat_eps));
            }
        }
    }
    constexpr int NH = NF / 2;                      // n-fragments per wave per pass
    constexpr int PH = NH * 16;                     // positions per wave per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int mbase = m0 + (wm * MF + i) * 16 + (lane >> 4) * 4;
            if (mbase >= M) continue;
            float bv[4], gm[4], bt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool in = mbase + r < M;
                bv[r] = (d.bias && in) ? d.bias[mbase + r] : 0.f;
                gm[r] = (smode == 3 && d.gamma && in) ? d.gamma[mbase + r] * st_rstd[i] : st_rstd[i];
                bt[r] = (smode == 3 && d.gamma && in) ? d.beta[mbase + r] : 0.f;
            }
#pragma unroll
            for (int nn = 0; nn < NH; ++nn) {
                const int n = pass * NH + nn;
                const int t = t0 + (wn * NF + n) * 16 + (lane & 15);
                if (t >= T) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[i][n][r] + bv[r];
                if (smode == 1 || smode == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (mbase + r < M) { st_s1[i] += o[r]; st_s2[i] += o[r] * o[r]; }
                    if (smode == 2) continue;
                } else if (smode == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (o[r] - st_mean[i]) * gm[r] + bt[r];
                }
                int cbase = mbase;
                if (glu) {
                    o[0] = o[0] * aero_sigmoid(o[1]);
                    o[1] = o[2] * aero_sigmoid(o[3]);
                    cbase = mbase >> 1;
                    if (d.layer_scale) {
                        o[0] *= d.layer_scale[cbase];
                        if (cbase + 1 < Mout) o[1] *= d.layer_scale[cbase + 1];
                    }
                } else if (d.act == AERO_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
                } else if (d.act == AERO_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = aero_gelu(o[r]);
                }
                const int64_t doff = (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + (int64_t)t * d.d_t + cbase;
                const int64_t roff = (int64_t)b * d.r_b + (int64_t)fdst * d.r_f + (int64_t)t * d.r_t + cbase;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r >= nout || cbase + r >= Mout) continue;
                    float x = o[r];
                    if (res && !res_in_copy) x += (float)res[roff + r];
                    if (d.post_add) x += d.post_add[(int64_t)fo * Mout + cbase + r];
                    o[r] = x * bsc + bsh;
                }
                if (p.staged) {
                    const int pc = wn * PH + nn * 16 + (lane & 15);
                    const int cl = cbase - (glu ? (m0 >> 1) : m0);
                    if (glu) *(h16x2*)&Cs[pc * CS + cl] = (h16x2){(h16)o[0], (h16)o[1]};
                    else *(h16x4*)&Cs[pc * CS + cl] = (h16x4){(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
                } else if (p.vec_out && cbase + nout <= Mout) {
                    if (d.dst_f32) {
                        if (glu) *(f32x2*)(dst32 + doff) = (f32x2){o[0], o[1]};
                        else *(f32x4*)(dst32 + doff) = (f32x4){o[0], o[1], o[2], o[3]};
                    } else {
                        if (glu) *(h16x2*)(dst16 + doff) = (h16x2){(h16)o[0], (h16)o[1]};
                        else *(h16x4*)(dst16 + doff) = (h16x4){(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r >= nout || cbase + r >= Mout) continue;
                        if (d.dst_f32) dst32[doff + r] = o[r];
                        else dst16[doff + r] = (h16)o[r];
                    }
                }
            }
        }
        if (p.staged && smode != 2) {
            __syncthreads();
            const int BMo = glu ? (BM >> 1) : BM;
            const int nvec = BMo >> 3;
            const int m0o = glu ? (m0 >> 1) : m0;
            h16* drow = dst16 + (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + m0o;
            const h16* rrow = res_in_copy ? res + (int64_t)b * d.r_b + (int64_t)fdst * d.r_f + m0o : nullptr;
            for (int idx = tid; idx < 64 * nvec; idx += NWV * 64) {
                const int pc = idx / nvec, cv = idx - pc * nvec;
                const int wq = pc / PH, rr = pc - wq * PH;
                const int t = t0 + (wq * NF + pass * NH + (rr >> 4)) * 16 + (rr & 15);
                if (t < T && m0o + cv * 8 < Mout) {
                    h16x8 v = *(const h16x8*)&Cs[pc * CS + cv * 8];
                    if (rrow) {
                        const h16x8 r8 = *(const h16x8*)(rrow + (int64_t)t * d.r_t + cv * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (h16)((float)v[e] + (float)r8[e]);
                    }
                    *(h16x8*)(drow + (int64_t)t * d.d_t + cv * 8) = v;
                }
            }
            __syncthreads();
        }
    }
    if (smode == 1 || smode == 2) {
        // fold fragment rows of one group, reduce across the wave in fp64, one atomic pair per (wave, group)
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int base_i = m0 + (wm * MF + i) * 16;
            const int grp = base_i / gs;
            if (i + 1 < MF && (base_i + 16) / gs == grp && base_i + 16 < M) {
                st_s1[i + 1] += st_s1[i];
                st_s2[i + 1] += st_s2[i];
                continue;
            }
            if (base_i >= M) continue;
            const double a = aero_wave_sum((double)st_s1[i]);
            const double c = aero_wave_sum((double)st_s2[i]);
            if (lane == 0) {
                atomicAdd(d.stats + (sitem + grp) * 2, a);
                atomicAdd(d.stats + (sitem + grp) * 2 + 1, c);
            }
        }
    }
}

// Epilogue dispatcher.  STATS instantiations contain ONLY the lean statistics paths (the host refuses anything else), so
// they do not carry the generic epilogue's registers; plain instantiations pick the lean path when they can.
#define AERO_EPI_FAST(ACT_, SM_) aero_conv_epilogue_fast<MF, WM, NWV, ACT_, SM_>(p, acc, Cs, b, fo, fdst, m0, t0)
template <int MF, int WM, bool STATS, int NWV = 4>
static __device__ __forceinline__ void aero_conv_epilogue(const AeroConvK& p, f32x4 (&acc)[MF][8 / (NWV / WM)], h16* Cs, int b, int fo,
                                                          int fdst, int m0, int t0) {
    const aero_conv_desc& d = p.d;
    if constexpr (STATS) {
        if (d.stat_mode == 1) AERO_EPI_FAST(AERO_ACT_NONE, 1);
        else if (d.stat_mode == 2) AERO_EPI_FAST(AERO_ACT_NONE, 2);
        else if (d.act == AERO_ACT_GLU) AERO_EPI_FAST(AERO_ACT_GLU, 3);
        else if (d.act == AERO_ACT_GELU) AERO_EPI_FAST(AERO_ACT_GELU, 3);
        else if (d.act == AERO_ACT_RELU) AERO_EPI_FAST(AERO_ACT_RELU, 3);
        else AERO_EPI_FAST(AERO_ACT_NONE, 3);
    } else {
        if (p.staged && !d.batch_scale) {
            switch (d.act) {
                case AERO_ACT_NONE: AERO_EPI_FAST(AERO_ACT_NONE, 0); break;
                case AERO_ACT_RELU: AERO_EPI_FAST(AERO_ACT_RELU, 0); break;
                case AERO_ACT_GELU: AERO_EPI_FAST(AERO_ACT_GELU, 0); break;
                default: AERO_EPI_FAST(AERO_ACT_GLU, 0); break;
            }
            return;
        }
        aero_conv_epilogue_generic<MF, WM, false, NWV>(p, acc, Cs, b, fo, fdst, m0, t0);
    }
}

template <int MF, int WM, bool STATS>
__global__ __launch_bounds__(256) void aero_conv_kernel(AeroConvK p) {
    constexpr int WN = 4 / WM;
    constexpr int NF = 8 / WN;
    constexpr int BM = 16 * MF * WM;
    constexpr int BN = 128;
    constexpr int AV = (BM * 4 + 255) / 256;
    // one LDS buffer: [A tile | B tile] during the K loop, then a [64][BM+8] fp16 output half-tile in the epilogue
    constexpr int CS = BM + 8;                       // padded row (16 B) -> 2-way ds_write conflicts at most
    constexpr int SMEM = (BM + BN) * 32 > 64 * CS ? (BM + BN) * 32 : 64 * CS;
    __shared__ AERO_LDS_ALIGN h16 smem[SMEM];
    h16* As = smem;
    h16* Bs = smem + BM * 32;
    h16* Cs = smem;

    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int id = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int mt = id % p.nmt;
    id /= p.nmt;
    const int tt = id % p.ntt;
    const int row = id / p.ntt;
    const int b = row / d.Fout, fo = row % d.Fout;
    const int fdst = fo - d.dst_f_off;
    if (fdst < 0 || fdst >= d.dst_F) return;  // trimmed row: block-uniform exit
    const int m0 = mt * BM, t0 = tt * BN;
    const int wset = d.transposed ? (fo % d.fstride) : 0;
    const int fbase = d.transposed ? (fo / d.fstride) : (fo * d.fstride);
    const h16* Wp = (const h16*)d.weight + ((int64_t)wset * p.Mpad + m0) * p.Ktot;
    const h16* s0 = (const h16*)d.src0;
    const h16* s1 = (const h16*)d.src1;
    const int C0 = d.C0, C1 = d.C1, T = d.T;

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    h16x8 ra[AV], rb[2];
    int j = -1, cc = p.cpt - 1, fi = 0;  // chunk iterator state: tap j, channel chunk cc

    auto next_chunk = [&]() -> bool {
        for (;;) {
            ++cc;
            if (cc == p.cpt) {
                cc = 0;
                ++j;
                while (j < d.ntaps) {
                    fi = fbase + d.df[aero_uniform(j)];
                    if (fi >= 0 && fi < d.Fin) break;
                    ++j;
                }
            }
            if (j >= d.ntaps) return false;
            // chunk lies entirely inside a NULL (all-zero) first source: nothing to add
            if (s0 == nullptr && (cc + 1) * 32 <= C0) continue;
            return true;
        }
    };

    auto fetch_b = [&](int t, int c) -> h16x8 {
        h16x8 z = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (t < 0 || t >= T) return z;
        if (p.vec_in) {
            if (c < C0) {
                if (s0) z = *(const h16x8*)(s0 + (int64_t)b * d.s0_b + (int64_t)fi * d.s0_f + (int64_t)t * d.s0_t + c);
            } else if (c - C0 < C1) {
                z = *(const h16x8*)(s1 + (int64_t)b * d.s1_b + (int64_t)fi * d.s1_f + (int64_t)t * d.s1_t + (c - C0));
            }
        } else if (p.vec4) {                                    // channel counts / strides that are only 8-byte aligned (C = 12)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int ch = c + hf * 4;
                h16x4 v = (h16x4){0, 0, 0, 0};
                if (ch < C0) {
                    if (s0) v = *(const h16x4*)(s0 + (int64_t)b * d.s0_b + (int64_t)fi * d.s0_f + (int64_t)t * d.s0_t + ch);
                } else if (ch - C0 < C1) {
                    v = *(const h16x4*)(s1 + (int64_t)b * d.s1_b + (int64_t)fi * d.s1_f + (int64_t)t * d.s1_t + (ch - C0));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) z[hf * 4 + e] = v[e];
            }
        } else {
            const h16* p0 = s0 ? s0 + (int64_t)b * d.s0_b + (int64_t)fi * d.s0_f + (int64_t)t * d.s0_t : nullptr;
            const h16* p1 = s1 ? s1 + (int64_t)b * d.s1_b + (int64_t)fi * d.s1_f + (int64_t)t * d.s1_t : nullptr;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ce = c + e;
                h16 v = (h16)0;
                if (ce < C0) {
                    if (p0) v = p0[ce];
                } else if (ce - C0 < C1) {
                    v = p1[ce - C0];
                }
                z[e] = v;
            }
        }
        return z;
    };

    auto load_chunk = [&]() {
        const int kofs = j * p.Cp + cc * 32;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int v = tid + 256 * i;
            if (v < BM * 4) ra[i] = *(const h16x8*)(Wp + (int64_t)(v >> 2) * p.Ktot + kofs + (v & 3) * 8);
        }
        const int dtj = d.dt[aero_uniform(j)];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i;
            rb[i] = fetch_b(t0 + (v >> 2) + dtj, cc * 32 + (v & 3) * 8);
        }
    };

    bool have = next_chunk();
    if (have) load_chunk();
    while (have) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int v = tid + 256 * i;
            if (v < BM * 4) *(h16x8*)&As[aero_tile_off(v >> 2, v & 3)] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i;
            *(h16x8*)&Bs[aero_tile_off(v >> 2, v & 3)] = rb[i];
        }
        __syncthreads();
        have = next_chunk();
        if (have) load_chunk();  // next chunk's global loads fly while this one feeds the MFMAs
        h16x8 af[MF], bf[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = *(const h16x8*)&As[aero_tile_off((wm * MF + i) * 16 + (lane & 15), lane >> 4)];
#pragma unroll
        for (int n = 0; n < NF; ++n) bf[n] = *(const h16x8*)&Bs[aero_tile_off((wn * NF + n) * 16 + (lane & 15), lane >> 4)];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[n], acc[i][n], 0, 0, 0);
        __syncthreads();
    }

    aero_conv_epilogue<MF, WM, STATS>(p, acc, Cs, b, fo, fdst, m0, t0);
}

// ------------------------------------------------------------------------------------------------------
// Direct-to-LDS pipeline (vector-aligned sources, regular tap grids): operands go HBM/L2 -> LDS with
// `global_load_lds_dwordx4`, two LDS stages, ONE barrier per K-chunk of KC channels; the next chunk's copies are in
// flight while the current one feeds the MFMAs.  No staging VGPRs, no ds_write.  Padding / out-of-range lanes read a
// zero page.  The LDS image keeps an XOR swizzle (conflict-free ds_read_b128); the destination of the copy is
// lane-linear, so the permutation is applied to the SOURCE address.
// PMC on the first version showed the loop was ISSUE-bound by integer overhead (~200 scalar/vector instructions per
// 16 MFMAs), not by LDS or HBM: everything lane-dependent is therefore hoisted out of the loop (per-lane pointers and
// 32-bit offsets), the per-chunk part is a handful of block-uniform scalars, and KC = 64 halves what is left.
// Round 5: NST LDS stages instead of two.  With two stages a chunk's copies have ONE chunk of MFMAs (0.1-0.3 us at 8-16 waves per CU)
// to land before the block-wide wait.  With NST > 2 the copies of chunk i + NST - 1 are issued while chunk i feeds the MFMAs, the wait in
// front of the barrier is a COUNTED `s_waitcnt vmcnt((NST - 2) * copies per wave and chunk)` and the barrier waits for LDS only (the
// k_conv_ring.h rules: a wave's vmcnt wait followed by a barrier every reader has passed orders the LDS-DMA writes; a stage is refilled
// after the barrier that follows its last read).  Measured per instantiation on the B = 64 forward (profiles/r05_glds_stages_ab.txt;
// 2 -> 3 stages): the 8-wave tile 164 -> 116, 166 -> 158, 405 -> 388, 234 -> 219, 207 -> 192 us (strided / transposed convs of the
// U-Net), the 48-row 4-wave tile 62 -> 56 us; but the 96-row 4-wave tile (three blocks per CU, 7.8 TB/s of L2 -> LDS traffic on the second
// encoder's strided conv) 120 -> 160 us and the 64-channel-chunk tiles 80 -> 87 us: those keep two stages.  Four stages of the 8-wave
// tile (80 KiB: one block per CU) lose what three gained.
#ifndef AERO_GLDS_NST
#define AERO_GLDS_NST 3
#endif
template <int MF, int WM, int KC, int NWV>
struct AeroGldsGeom {
    static constexpr int BM = 16 * MF * WM;
    static constexpr int NST = (NWV == 8 || (KC == 32 && WM == 1)) ? AERO_GLDS_NST : 2;
    static constexpr int STAGE = (BM + 128) * KC;
    static constexpr int CS = BM + 8;
    static constexpr int SMEM = NST * STAGE > 64 * CS ? NST * STAGE : 64 * CS;    // h16 elements
};

template <int MF, int WM, int KC, bool STATS, int NWV>
static __device__ __forceinline__ void aero_conv_glds_body(const AeroConvK& p, h16* smem) {
    constexpr int WN = NWV / WM;
    constexpr int NF = 8 / WN;
    constexpr int BM = 16 * MF * WM;
    constexpr int BN = 128;
    constexpr int SLOTS = KC / 8;                   // 16-byte slots per tile row
    constexpr int KS = KC / 32;                     // MFMA k-steps per chunk
    constexpr int STAGE = (BM + BN) * KC;
    constexpr int NIA = (BM * SLOTS / 64 + NWV - 1) / NWV;  // A copy instructions per wave
    constexpr int NIB = BN * SLOTS / 64 / NWV;              // B copy instructions per wave
    h16* Cs = smem;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int id = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int sp = id % p.tsplit;                   // tap group of this block (aero_conv_desc.tap_split; 1 group: sp = 0)
    id /= p.tsplit;
    const int mt = id % p.nmt;
    id /= p.nmt;
    const int tt = id % p.ntt;
    const int row = id / p.ntt;
    const int b = row / d.Fout, fo = row % d.Fout;
    const int fdst = fo - d.dst_f_off;
    if (!d.scatter_M && (fdst < 0 || fdst >= d.dst_F)) return;
    const int m0 = mt * BM, t0 = tt * BN;
    const int wset = d.transposed ? (fo % d.fstride) : 0;
    const int fbase = (d.transposed ? (fo / d.fstride) : (fo * d.fstride)) + p.f_lo;
    const h16* Wp = (const h16*)d.weight + ((int64_t)wset * p.Mpad + m0) * p.Ktot;
    const h16* s0 = (const h16*)d.src0;
    const h16* s1 = (const h16*)d.src1;
    const h16* zp = aero_zero_page;
    const int C0 = d.C0, C01 = d.C0 + d.C1, T = d.T;
    const int st0 = (int)d.s0_t, st1 = (int)d.s1_t;
    const int cpt = p.Cp / KC;
    const int cc_lo = (s0 == nullptr && C0 % KC == 0) ? C0 / KC : 0;
    const int nF = d.ntaps / p.nT;
    const int jt_lo = sp * (p.nT / p.tsplit), jt_hi = jt_lo + p.nT / p.tsplit;     // this block's time taps

    // ---- lane-invariant parts of the copy addresses: full per-lane pointers (item, in-tile position and channel slice
    // folded in), so that a K-chunk only adds ONE block-uniform 32-bit element offset per source.  PMC: the loop used
    // to spend ~85 scalar + ~50 vector instructions per 16 MFMAs rebuilding 64-bit row addresses and re-loading spilled
    // kernel arguments (and the zero-page address) every chunk.
    const h16* a_ptr[NIA];
    const h16* pb0[NIB];
    const h16* pb1[NIB];
    int b_pos[NIB], b_q8[NIB];
    const h16* base0 = s0 ? s0 + (int64_t)b * d.s0_b : zp;
    const h16* base1 = s1 ? s1 + (int64_t)b * d.s1_b - C0 : zp;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int s = (wave + NWV * i) * 64 + lane;
        const int r = s / SLOTS, q = (s % SLOTS) ^ aero_tile_swz<KC>(r);
        a_ptr[i] = Wp + (r * p.Ktot + q * 8);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int s = (wave + NWV * i) * 64 + lane;
        const int pos = s / SLOTS, q = (s % SLOTS) ^ aero_tile_swz<KC>(pos);
        b_pos[i] = pos;
        b_q8[i] = q * 8;
        pb0[i] = base0 + (pos * st0 + q * 8);
        pb1[i] = base1 + (pos * st1 + q * 8);
    }
    // the zero page's address lives in a VGPR pair (opaque to the compiler: no GOT reload inside the loop)
    const h16* zpv = zp;
    int Tv = T;                                                 // likewise T: compared per lane, no SGPR (re)load in the loop
#ifndef AERO_EMU
    asm volatile("" : "+v"(zpv));
    asm volatile("" : "+v"(Tv));
#endif

    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- chunk iterator over (frequency tap jf, time tap jt, channel chunk cc); all state block-uniform.  It carries
    // the weight offset `kofs`, the source time `tsh` of position 0 and the element offsets `off0/off1` of the chunk
    // relative to the per-lane pointers; a plain cc step is three additions.
    const int nT = p.nT, f_step = p.f_step, t_step = p.t_step, Cpk = p.Cp;
    const int s0f = (int)d.s0_f, s1f = (int)d.s1_f;             // in-item row offsets fit 32 bits (checked on the host)
    const int t_base = t0 + p.t_lo;
    int jf = -1, jt = jt_hi - 1, cc = cpt - 1, fi = 0;
    int kofs = 0, tsh = 0, off0 = 0, off1 = 0;
    bool tin[NIB];                                              // source time of this lane's position inside [0, T): per tap
    auto next_chunk = [&]() -> bool {
        if (++cc < cpt) {
            kofs += KC; off0 += KC; off1 += KC;
            return true;
        }
        cc = cc_lo;
        if (++jt >= jt_hi) {
            jt = jt_lo;
            for (;;) {
                if (++jf >= nF) return false;
                fi = fbase + jf * f_step;
                if (fi >= 0 && fi < d.Fin) break;
            }
        }
        kofs = (jf * nT + jt) * Cpk + cc_lo * KC;
        tsh = t_base + jt * t_step;
#pragma unroll
        for (int i = 0; i < NIB; ++i) tin[i] = (unsigned)(b_pos[i] + tsh) < (unsigned)Tv;
        off0 = fi * s0f + tsh * st0 + cc_lo * KC;
        off1 = fi * s1f + tsh * st1 + cc_lo * KC;
        return true;
    };
    const bool has0 = s0 != nullptr;
    auto issue = [&](int buf) {
        h16* As = smem + buf * STAGE;
        h16* Bs = As + BM * KC;
#pragma unroll
        for (int i = 0; i < NIA; ++i)
            if (wave + NWV * i < BM * SLOTS / 64) aero_glds16(a_ptr[i] + kofs, As + (wave + NWV * i) * 512);
        const int c_lo = cc * KC;
        const int lim0 = C0 - c_lo, lim1 = C01 - c_lo;        // channel q8 comes from src0 if q8 < lim0, src1 if q8 < lim1
        // A chunk that comes whole from one source (every chunk but a straddling or ragged one): the select is
        // block-uniform and the copy address is the lane pointer plus one scalar offset.  (PMC: the per-lane form below
        // cost ~22 vector instructions per copy, next to 6-16 MFMAs per chunk.)  Two separate branches on purpose: a
        // select between the two sources' variables makes the compiler keep them in scratch and select their addresses.
        if (lim0 >= KC) {
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                aero_glds16((tin[i] && has0) ? pb0[i] + off0 : zpv, Bs + (wave + NWV * i) * 512);
            return;
        }
        if (lim0 <= 0 && lim1 >= KC) {
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                aero_glds16(tin[i] ? pb1[i] + off1 : zpv, Bs + (wave + NWV * i) * 512);
            return;
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const bool u0 = b_q8[i] < lim0;
            const bool ok = tin[i] && (u0 ? has0 : (b_q8[i] < lim1));
            const h16* ptr = u0 ? pb0[i] + off0 : pb1[i] + off1;
            aero_glds16(ok ? ptr : zpv, Bs + (wave + NWV * i) * 512);
        }
    };

    auto compute = [&](int buf) {
        const h16* As = smem + buf * STAGE;
        const h16* Bs = As + BM * KC;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h16x8 af[MF], bf[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i) af[i] = *(const h16x8*)&As[aero_tile_off_kc<KC>((wm * MF + i) * 16 + (lane & 15), ks * 4 + (lane >> 4))];
#pragma unroll
            for (int n = 0; n < NF; ++n) bf[n] = *(const h16x8*)&Bs[aero_tile_off_kc<KC>((wn * NF + n) * 16 + (lane & 15), ks * 4 + (lane >> 4))];
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int n = 0; n < NF; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[n], acc[i][n], 0, 0, 0);
        }
    };
    constexpr int NST = AeroGldsGeom<MF, WM, KC, NWV>::NST;
    if constexpr (NST == 2) {
        bool have = next_chunk();
        if (have) issue(0);
        int buf = 0;
        while (have) {
            __syncthreads();                   // stage `buf` has landed (vmcnt drained) and stage buf^1 is free again
            have = next_chunk();
            if (have) issue(buf ^ 1);
            compute(buf);
            buf ^= 1;
        }
    } else {
        // copies THIS wave issues per chunk: NIB activation pieces + NIA or NIA - 1 weight pieces (they are dealt out to the waves in
        // rounds; wave-uniform) -- the loop is instantiated for both counts so that the wait's immediate is a compile-time constant
        auto pipeline = [&](auto keep_c) {
            constexpr int KEEP = decltype(keep_c)::value;        // copies of the NST - 2 younger chunks may stay in flight
            bool have = next_chunk();
            int wr = 0, rd = 0, fl = 0;                          // fl: chunks issued and not yet consumed
            for (; fl < NST - 1 && have; ++fl) {
                issue(wr);
                wr = wr + 1 == NST ? 0 : wr + 1;
                have = next_chunk();
            }
            while (have) {                                       // steady state: NST - 1 chunks in flight
                aero_wait_vm<KEEP>();                            // the oldest of them has landed (this wave's part of it)
                aero_phase_barrier();                            // ... every wave's part; and the stage read last iteration is free
                issue(wr);
                wr = wr + 1 == NST ? 0 : wr + 1;
                have = next_chunk();
                compute(rd);
                rd = rd + 1 == NST ? 0 : rd + 1;
            }
            for (; fl; --fl) {                                   // tail (or a contraction shorter than the pipeline)
                aero_wait_vm<0>();
                aero_phase_barrier();
                compute(rd);
                rd = rd + 1 == NST ? 0 : rd + 1;
            }
        };
        constexpr int TOTA = BM * SLOTS / 64;
        if (TOTA % NWV == 0 || wave + NWV * (NIA - 1) < TOTA) pipeline(std::integral_constant<int, (NST - 2) * (NIB + NIA)>());
        else pipeline(std::integral_constant<int, (NST - 2) * (NIB + NIA - 1)>());
    }
    if (p.tsplit > 1) {                        // partial sums of this tap group -> fp32 accumulator (finished by aero_split_finish)
        float* ws = d.split_acc + ((int64_t)((sp * d.B + b) * d.Fout + fo) * T) * d.M;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int t = t0 + (wn * NF + n) * 16 + (lane & 15);
            if (t >= T) continue;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = m0 + (wm * MF + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < d.M) ws[(int64_t)t * d.M + m + r] = acc[i][n][r];
            }
        }
        return;
    }
    __syncthreads();                           // all waves done with the operand stages: smem becomes the output tile
    aero_conv_epilogue<MF, WM, STATS, NWV>(p, acc, Cs, b, fo, fdst, m0, t0);
}

// (KC 32, 128-row tile: ask for 3 blocks per CU -- 168 registers -- instead of the 204 the allocator takes by default)
template <int MF, int WM, int KC, bool STATS>
__global__ __launch_bounds__(256, (KC == 32 && MF * WM >= 6) ? 3 : 1) void aero_conv_glds_kernel(AeroConvK p) {
    aero_conv_glds_body<MF, WM, KC, STATS, 4>(p, (h16*)AERO_DYN_SMEM);      // (dynamic: three 64-channel stages exceed the static 64 KiB)
}

// 256 output channels x 128 steps with EIGHT waves (4 x 2, each 64 x 64 as above): the activation tile is shared by
// twice as many output rows, so the global->LDS traffic per MFMA drops by a quarter and each wave issues 3 copy
// instructions per 32-channel chunk instead of 4.  Dynamic LDS (48 KiB with KC 32).
template <int MF, int KC, bool STATS>
__global__ __launch_bounds__(512, 4) void aero_conv_glds8_kernel(AeroConvK p) {     // 4 waves per SIMD (two blocks per CU): <= 128 registers
    aero_conv_glds_body<MF, 4, KC, STATS, 8>(p, (h16*)AERO_DYN_SMEM);
}

// ------------------------------------------------------------------------------------------------------
// Skinny outputs (M <= 16: the FTB 5-channel squeeze, the 12-channel DConv squeeze of the first layer, the last
// decoder's 2-channel transposed conv): HBM-bound streaming, not GEMMs.  A 128-step tile kernel spends its time in
// per-block set-up for two or three K-chunks; here the blocks are PERSISTENT: the 16 x Ktot weight slab of every
// weight set is staged in LDS once, then each WAVE walks over (row, group of 64-step segments) items on its own:
//   * the loads of up to SIX (segment, 32-channel chunk) slots are issued back to back -- coalesced 16-byte loads,
//     4 adjacent lanes = the 64 bytes of one step -- so an item costs ONE memory latency, not one per chunk;
//   * each slot is transposed through a private 4-KiB LDS tile into MFMA B-fragment order (no block barrier: LDS is
//     in-order per wave) and multiplied against the LDS-resident weights;
//   * the outputs of a segment (64 steps x M channels: one contiguous byte range when the destination is dense) go
//     back through the same LDS tile and leave as full-width coalesced dword stores.
// Earlier forms, measured: fragment-order global loads (adjacent lanes 96+ B apart: 1.4 TB/s); one chunk in flight
// per wave and per-channel 2-byte stores (every chunk waited for the previous item's partial-line store acks: 1.8 TB/s).
// Roofline: HBM, algorithmic bytes = source rows + output rows.
#define AERO_SKINNY_WMAX 832                        /* halves per weight row, all sets together (incl. 8 of padding each) */
#define AERO_SKINNY_SLOTS 6
// VW = halves per load piece (8: one 16-byte load per lane and step; 4/2/1 for channel counts or strides that are
// only 8-/4-/2-byte aligned).  Every load is unconditional (masked lanes read the zero page): no branches between the
// loads of a batch, so the compiler keeps them all in flight instead of waiting after each one.
template <int VW>
__global__ __launch_bounds__(256) void aero_conv_skinny_kernel(AeroConvK p) {
    __shared__ AERO_LDS_ALIGN h16 Ws[16 * AERO_SKINNY_WMAX];
    __shared__ AERO_LDS_ALIGN h16 Xs[4][64 * 32];
    typedef h16 hvw __attribute__((ext_vector_type(VW > 1 ? VW : 2)));
    constexpr int NS = AERO_SKINNY_SLOTS;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int WS = p.Ktot + 8;                                  // padded LDS row: conflict-free fragment reads
    const int nws = d.transposed ? d.fstride : 1;
    const int kv = p.Ktot >> 3;
    for (int v = tid; v < nws * 16 * kv; v += 256) {
        const int ws = v / (16 * kv), rem = v - ws * (16 * kv);
        const int r = rem / kv, q = rem - r * kv;
        *(h16x8*)&Ws[(ws * 16 + r) * WS + q * 8] = *(const h16x8*)((const h16*)d.weight + ((int64_t)ws * p.Mpad + r) * p.Ktot + q * 8);
    }
    __syncthreads();
    const h16* s0 = (const h16*)d.src0;
    const h16* s1 = (const h16*)d.src1;
    const int C0 = d.C0, C01 = d.C0 + d.C1, T = d.T, M = d.M;
    const bool has0 = s0 != nullptr;
    h16* Xw = Xs[wave];
    const int lp = lane >> 2, lq = lane & 3;                    // load role: step lp (+16 i), 8-channel slice lq
    const int mbase = (lane >> 4) * 4;
    const h16x8 zero8 = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (d.bias && mbase + r < M) ? d.bias[mbase + r] : 0.f;
    h16* dst16 = (h16*)d.dst;
    float* dst32 = (float*)d.dst;
    const bool dense = (d.d_t == M) && (((uintptr_t)d.dst & 3) == 0);
    const int nseg = (T + 63) >> 6;
    const int SG = p.nmt;                                       // segments per item (host: 6 / worst-case chunks per segment)
    const int ngrp = (nseg + SG - 1) / SG;
    const long nitems = (long)d.B * d.Fout * ngrp;
    for (long item = (long)blockIdx.x * 4 + wave; item < nitems; item += (long)gridDim.x * 4) {
        const int grp = (int)(item % ngrp);
        const int row = (int)(item / ngrp);
        const int b = row / d.Fout, fo = row - b * d.Fout;
        const int fdst = fo - d.dst_f_off;
        if (fdst < 0 || fdst >= d.dst_F) continue;
        const int wset = d.transposed ? (fo % d.fstride) : 0;
        const int fbase = d.transposed ? (fo / d.fstride) : (fo * d.fstride);
        const h16* Wl = Ws + (wset * 16 + (lane & 15)) * WS + (lane >> 4) * 8;
        const int seg_end = (grp + 1) * SG < nseg ? (grp + 1) * SG : nseg;
        // (segment, tap j, 32-channel chunk cc) iterator, wave-uniform; a segment without any valid chunk yields one
        // empty slot (kofs < 0) so that its bias-only output is still written
        int it_seg = grp * SG, it_j = -1, it_cc = p.cpt - 1, it_fi = 0, it_n = 0;
        auto advance = [&]() -> bool {                          // -> false: item exhausted
            for (;;) {
                bool found = false;
                for (;;) {
                    if (++it_cc == p.cpt) {
                        it_cc = 0;
                        for (++it_j; it_j < d.ntaps; ++it_j) {
                            it_fi = fbase + d.df[aero_uniform(it_j)];
                            if (it_fi >= 0 && it_fi < d.Fin) break;
                        }
                    }
                    if (it_j >= d.ntaps) break;
                    if (!has0 && (it_cc + 1) * 32 <= C0) continue;   // chunk inside the structurally-zero first source
                    found = true;
                    break;
                }
                if (found) { ++it_n; return true; }
                if (it_n == 0) { it_n = 1; it_j = d.ntaps; it_cc = -1; return true; }     // empty segment
                if (++it_seg >= seg_end) return false;
                it_j = -1; it_cc = p.cpt - 1; it_n = 0;
            }
        };
        auto load = [&](h16x8* nb, int seg, int jj, int cc, int fi) {     // global -> registers, coalesced
            const int c = cc * 32 + lq * 8;
            const h16* rb0 = has0 ? s0 + (int64_t)b * d.s0_b + (int64_t)fi * d.s0_f : aero_zero_page;
            const h16* rb1 = s1 ? s1 + (int64_t)b * d.s1_b + (int64_t)fi * d.s1_f : aero_zero_page;
            const int t0 = seg * 64 + lp + d.dt[aero_uniform(jj)];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = t0 + i * 16;
                const bool tok = t >= 0 && t < T;
                h16x8 z;
#pragma unroll
                for (int pc = 0; pc < 8 / VW; ++pc) {
                    const int ce = c + pc * VW;
                    const bool u0 = ce < C0;
                    const bool ok = tok && (u0 ? has0 : (ce < C01));
                    const h16* sp = u0 ? rb0 + (int64_t)t * d.s0_t + ce : rb1 + (int64_t)t * d.s1_t + (ce - C0);
                    sp = ok ? sp : (const h16*)aero_zero_page;
                    if (VW == 1) {
                        z[pc] = *sp;
                    } else {
                        const hvw v = *(const hvw*)sp;
#pragma unroll
                        for (int e = 0; e < VW; ++e) z[pc * VW + e] = v[e];
                    }
                }
                nb[i] = z;
            }
        };
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto mma = [&](const h16x8* nb, int kofs) {             // registers -> private LDS tile -> fragments -> MFMA
            aero_wave_sync();                                   // earlier readers of the tile are done
#pragma unroll
            for (int i = 0; i < 4; ++i) *(h16x8*)&Xw[aero_tile_off(i * 16 + lp, lq)] = nb[i];
            aero_wave_sync();
            const h16x8 af = *(const h16x8*)(Wl + kofs);
            h16x8 bf[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bf[g] = *(const h16x8*)&Xw[aero_tile_off(g * 16 + (lane & 15), lane >> 4)];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[g], acc[g], 0, 0, 0);
        };
        auto epilogue = [&](int seg) {
            const float bsc = d.batch_scale ? d.batch_scale[b] : 1.f;
            const float bsh = d.batch_scale ? d.batch_shift[b] : 0.f;
            const int tw = seg * 64 + (lane & 15);
            if (dense) aero_wave_sync();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int t = tw + g * 16;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[g][r] + bv[r];
                    if (d.act == AERO_ACT_RELU) x = fmaxf(x, 0.f);
                    else if (d.act == AERO_ACT_GELU) x = aero_gelu(x);
                    o[r] = x * bsc + bsh;
                    acc[g][r] = 0.f;
                }
                if (t >= T || mbase >= M) continue;
                if (dense) {                                    // stage [step][channel]: the memory order of the segment
                    const int li = (g * 16 + (lane & 15)) * M + mbase;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mbase + r >= M) continue;
                        if (d.dst_f32) ((float*)Xw)[li + r] = o[r];
                        else Xw[li + r] = (h16)o[r];
                    }
                    continue;
                }
                const int64_t doff = (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + (int64_t)t * d.d_t + mbase;
                if (p.vec_out && mbase + 4 <= M) {
                    if (d.dst_f32) *(f32x4*)(dst32 + doff) = (f32x4){o[0], o[1], o[2], o[3]};
                    else *(h16x4*)(dst16 + doff) = (h16x4){(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mbase + r >= M) continue;
                        if (d.dst_f32) dst32[doff + r] = o[r];
                        else dst16[doff + r] = (h16)o[r];
                    }
                }
            }
            if (dense) {
                aero_wave_sync();
                const int npos = T - seg * 64 < 64 ? T - seg * 64 : 64;
                const int N = npos * M;
                const int64_t E0 = (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + (int64_t)seg * 64 * M;
                if (d.dst_f32) {
                    for (int idx = lane; idx < N; idx += 64) dst32[E0 + idx] = ((const float*)Xw)[idx];
                } else {
                    const int a = (int)(E0 & 1);                // halves between the dword boundary and the segment start
                    const int ndw = (a + N + 1) >> 1;
                    uint32_t* gw = (uint32_t*)(dst16 + (E0 - a));
                    for (int w = lane; w < ndw; w += 64) {
                        const int le = 2 * w - a;
                        const bool lo = le >= 0, hi = le + 1 < N;
                        if (lo && hi) {
                            union { h16 h[2]; uint32_t u; } pk;
                            pk.h[0] = Xw[le];
                            pk.h[1] = Xw[le + 1];
                            gw[w] = pk.u;
                        } else if (lo) {
                            dst16[E0 + le] = Xw[le];
                        } else if (hi) {
                            dst16[E0 + le + 1] = Xw[le + 1];
                        }
                    }
                }
            }
        };
        // batches of up to NS slots: all loads first, then transpose + MFMA slot by slot
        bool more = advance();
        while (more) {
            h16x8 nb[NS][4];
            int s_kofs[NS], s_seg[NS];
            bool s_last[NS];
            int ns = 0;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                if (!more) continue;
                s_seg[sl] = it_seg;
                s_kofs[sl] = it_cc < 0 ? -1 : it_j * p.Cp + it_cc * 32;
                if (it_cc >= 0) load(nb[sl], it_seg, it_j, it_cc, it_fi);
                more = advance();
                s_last[sl] = !more || it_seg != s_seg[sl];
                ns = sl + 1;
            }
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                if (sl >= ns) continue;
                if (s_kofs[sl] >= 0) mma(nb[sl], s_kofs[sl]);
                if (s_last[sl]) epilogue(s_seg[sl]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Lean form of the skinny kernel for the shapes the model actually runs (one 16-byte-aligned source, regular tap grid,
// at most six K-chunks per segment, dense destination).  PMC on the general kernel above: 13 k vector + 14 k scalar
// instructions per wave for 420 MFMAs -- the generic iterator, 64-bit per-lane address arithmetic and per-lane validity
// selects made a streaming kernel issue-bound.  Here:
//   * everything about a slot that does not depend on the row (tap, channel chunk, weight offset) is fixed per kernel;
//   * per-lane addressing is one 32-bit offset computed once, a slot adds a block-uniform 32-bit offset to a
//     block-uniform 64-bit row pointer; interior slots (all 64 steps inside [0, T), all 32 channels present) issue four
//     unconditional loads;
//   * a transposed conv whose `fstride` weight sets fit into the 16 MFMA rows together (fstride * M <= 16: the last
//     decoder, 4 x 2) is computed from the INPUT side: one pass over source rows (q, q-1) yields all `fstride` output
//     rows 4q..4q+3, instead of re-reading the source once per output row (8x less L2 traffic, 4x fewer MFMAs).
#define AERO_STREAM_SLOTS 6
__global__ __launch_bounds__(256) void aero_conv_stream_kernel(AeroConvK p) {
    __shared__ AERO_LDS_ALIGN h16 Ws[16 * AERO_SKINNY_WMAX];
    __shared__ AERO_LDS_ALIGN h16 Xs[4][64 * 32];
    constexpr int NS = AERO_STREAM_SLOTS;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int M = d.M, T = d.T, C0 = d.C0;
    const int stack = d.transposed ? d.fstride : 1;               // weight sets stacked into the MFMA rows
    const int WS = p.Ktot + 8;
    const int kv = p.Ktot >> 3;
    for (int v = tid; v < 16 * kv; v += 256) {
        const int R = v / kv, q = v - R * kv;
        const int r = R / M, m = R - r * M;
        h16x8 w = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (r < stack) w = *(const h16x8*)((const h16*)d.weight + ((int64_t)r * p.Mpad + m) * p.Ktot + q * 8);
        *(h16x8*)&Ws[R * WS + q * 8] = w;
    }
    __syncthreads();
    h16* Xw = Xs[wave];
    const h16* s0 = (const h16*)d.src0;
    const int st = (int)d.s0_t;
    const int lp = lane >> 2, lq = lane & 3;
    const int voff = lp * st + lq * 8;
    const h16* Wl = Ws + (lane & 15) * WS + (lane >> 4) * 8;
    const h16* zpv = aero_zero_page;
    // per-lane output rows: R = (lane>>4)*4 + rr  ->  (weight set r, channel m)
    int o_r[4], o_m[4];
    float bv[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int R = (lane >> 4) * 4 + rr;
        o_r[rr] = R / M;
        o_m[rr] = R - o_r[rr] * M;
        bv[rr] = (d.bias && o_r[rr] < stack) ? d.bias[o_m[rr]] : 0.f;
    }
    // slot table (block-uniform, fixed for the whole kernel): slot s = (segment s / nk, chunk k = s % nk)
    const int cpt = p.cpt, nT = p.nT;
    const int nk = d.ntaps * cpt;
    const int SG = p.nmt;
    int sl_seg[NS], sl_jf[NS], sl_dt[NS], sl_cc[NS], sl_kofs[NS];
    bool sl_last[NS];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
        const int sg = sidx / nk, k = sidx - sg * nk;
        const int tap = k / cpt;
        sl_seg[sidx] = sg < SG ? sg : -1;
        sl_cc[sidx] = k - tap * cpt;
        sl_jf[sidx] = tap / nT;
        sl_dt[sidx] = p.t_lo + (tap - sl_jf[sidx] * nT) * p.t_step;
        sl_kofs[sidx] = k * 32;
        sl_last[sidx] = k == nk - 1;
    }
    h16* dst16 = (h16*)d.dst;
    float* dst32 = (float*)d.dst;
    const int nseg = (T + 63) >> 6;
    const int ngrp = (nseg + SG - 1) / SG;
    const int NR = d.transposed ? (d.Fout + d.fstride - 1) / d.fstride : d.Fout;
    const int nitems = d.B * NR * ngrp;
    for (int item = (int)blockIdx.x * 4 + wave; item < nitems; item += (int)gridDim.x * 4) {
        const int grp = item % ngrp;
        const int row = item / ngrp;
        const int b = row / NR, q = row - b * NR;
        const int fbase = (d.transposed ? q : q * d.fstride) + p.f_lo;
        const h16* sb = s0 + (int64_t)b * d.s0_b;
        // ---- all loads of the item first
        h16x8 nb[NS][4];
        bool ok[NS];
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int seg = grp * SG + sl_seg[sidx];
            const int fi = fbase + sl_jf[sidx] * p.f_step;
            ok[sidx] = sl_seg[sidx] >= 0 && seg < nseg && fi >= 0 && fi < d.Fin;
            if (!ok[sidx]) continue;
            const h16* rb = sb + (int64_t)fi * d.s0_f;
            const int tb = seg * 64 + sl_dt[sidx];
            const int uo = tb * st + sl_cc[sidx] * 32;
            if (tb >= 0 && tb + 64 <= T && (sl_cc[sidx] + 1) * 32 <= C0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) nb[sidx][i] = *(const h16x8*)(rb + (voff + uo + i * 16 * st));
            } else {
                const bool cok = sl_cc[sidx] * 32 + lq * 8 < C0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = tb + lp + i * 16;
                    const h16* sp = rb + (voff + uo + i * 16 * st);
                    nb[sidx][i] = *(const h16x8*)((cok && t >= 0 && t < T) ? sp : zpv);
                }
            }
        }
        // ---- transpose + MFMA slot by slot; a segment's output leaves after its last chunk
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float bsc = d.batch_scale ? d.batch_scale[b] : 1.f;
        const float bsh = d.batch_scale ? d.batch_shift[b] : 0.f;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            if (ok[sidx]) {
                aero_wave_sync();
#pragma unroll
                for (int i = 0; i < 4; ++i) *(h16x8*)&Xw[aero_tile_off(i * 16 + lp, lq)] = nb[sidx][i];
                aero_wave_sync();
                const h16x8 af = *(const h16x8*)(Wl + sl_kofs[sidx]);
                h16x8 bf[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bf[g] = *(const h16x8*)&Xw[aero_tile_off(g * 16 + (lane & 15), lane >> 4)];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[g], acc[g], 0, 0, 0);
            }
            const int seg = grp * SG + sl_seg[sidx];
            if (!sl_last[sidx] || sl_seg[sidx] < 0 || seg >= nseg) continue;
            // epilogue of segment `seg`: stage [set r][step][m] in the wave's LDS tile, then dense coalesced stores
            const int npos = T - seg * 64 < 64 ? T - seg * 64 : 64;
            aero_wave_sync();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int pos = g * 16 + (lane & 15);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float x = acc[g][rr] + bv[rr];
                    if (d.act == AERO_ACT_RELU) x = fmaxf(x, 0.f);
                    else if (d.act == AERO_ACT_GELU) x = aero_gelu(x);
                    x = x * bsc + bsh;
                    acc[g][rr] = 0.f;
                    if (o_r[rr] < stack) {
                        const int li = (o_r[rr] * 64 + pos) * M + o_m[rr];
                        if (d.dst_f32) ((float*)Xw)[li] = x;
                        else Xw[li] = (h16)x;
                    }
                }
            }
            aero_wave_sync();
            const int N = npos * M;
            for (int r = 0; r < stack; ++r) {
                const int fo = d.transposed ? q * d.fstride + r : q;
                const int fdst = fo - d.dst_f_off;
                if (fo >= d.Fout || fdst < 0 || fdst >= d.dst_F) continue;
                const int64_t E0 = (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + (int64_t)seg * 64 * M;
                if (d.dst_f32) {
                    const float* lsrc = (const float*)Xw + r * 64 * M;
                    for (int idx = lane; idx < N; idx += 64) dst32[E0 + idx] = lsrc[idx];
                } else {
                    const h16* lsrc = Xw + r * 64 * M;
                    const int a = (int)(E0 & 1);                // halves between the dword boundary and the segment start
                    const int ndw = (a + N + 1) >> 1;
                    uint32_t* gw = (uint32_t*)(dst16 + (E0 - a));
                    for (int w = lane; w < ndw; w += 64) {
                        const int le = 2 * w - a;
                        const bool lo = le >= 0, hi = le + 1 < N;
                        if (lo && hi) {
                            union { h16 h[2]; uint32_t u; } pk;
                            pk.h[0] = lsrc[le];
                            pk.h[1] = lsrc[le + 1];
                            gw[w] = pk.u;
                        } else if (lo) {
                            dst16[E0 + le] = lsrc[le];
                        } else if (hi) {
                            dst16[E0 + le + 1] = lsrc[le + 1];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Pointwise conv with a handful of channels on both sides and a FREQUENCY-major destination ([b][t][f][m], d_f = M,
// d_t = F*M): the first FTB's 2 -> 5 squeeze, whose output feeds a conv1d over (f, m) (reference modules.py:287-289,
// 309-311).  Per position that is C FMAs per output, so the MFMA tile machinery is pure overhead; what matters is
// that both sides move in full lines although the layout is transposed between them: a block owns a 32-row x 64-step
// tile, reads it along time (256-byte runs), converts in registers, parks the results in LDS as [step][row][m] and
// writes every step's 32*M contiguous values as dwords.  Roofline: HBM, (C + M) * 2 bytes per position.
#define AERO_TINY_TF 32
#define AERO_TINY_TT 64
__global__ __launch_bounds__(256) void aero_conv_tiny_kernel(AeroConvK p) {
    __shared__ AERO_LDS_ALIGN h16 Os[AERO_TINY_TT * AERO_TINY_TF * 8];
    __shared__ float Wt[8 * 8 + 8];
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x;
    const int C = d.C0, M = d.M, T = d.T, F = d.Fout;
    int id = (int)blockIdx.x;
    const int tt = id % p.ntt;
    id /= p.ntt;
    const int ft = id % p.nmt;
    const int b = id / p.nmt;
    const int f0 = ft * AERO_TINY_TF, t0 = tt * AERO_TINY_TT;
    if (tid < M * 8) {
        const int m = tid >> 3, c = tid & 7;
        Wt[tid] = c < C ? (float)((const h16*)d.weight)[(int64_t)m * p.Ktot + c] : 0.f;
    }
    if (tid < 8) Wt[64 + tid] = (d.bias && tid < M) ? d.bias[tid] : 0.f;
    __syncthreads();
    // phase 1: thread -> row fl, 8 consecutive steps
    const int fl = tid >> 3, tl0 = (tid & 7) * 8;
    const int f = f0 + fl;
    const h16* src = (const h16*)d.src0 + (int64_t)b * d.s0_b + (int64_t)f * d.s0_f;
    float w[8][8], bias[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        bias[m] = Wt[64 + m];
#pragma unroll
        for (int c = 0; c < 8; ++c) w[m][c] = Wt[m * 8 + c];
    }
    // two-channel source with a pitch of two (the normalised spectrogram's (re, im) pairs: the first FTB): a step is ONE dword; all eight
    // are requested up front, unconditionally (row / step clamped, zeroed below) -- the general form loads 16 halves one by one, each under
    // its own predicate, i.e. each behind its own wait (70 us for 115 MB)
    const bool pair = C == 2 && d.s0_t == 2 && ((((uintptr_t)d.src0) | (uintptr_t)(d.s0_b * 2) | (uintptr_t)(d.s0_f * 2)) & 3) == 0;
    uint32_t xp[8];
    if (pair) {
        const int fc = f < F ? f : F - 1;
        const uint32_t* s32 = (const uint32_t*)((const h16*)d.src0 + (int64_t)b * d.s0_b + (int64_t)fc * d.s0_f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + tl0 + k;
            xp[k] = s32[t < T ? t : T - 1];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int tl = tl0 + k, t = t0 + tl;
        float x[8];
        if (pair) {
            union { uint32_t u; h16x2 h; } cv;
            cv.u = xp[k];
            const bool live = f < F && t < T;
            x[0] = live ? (float)cv.h[0] : 0.f;
            x[1] = live ? (float)cv.h[1] : 0.f;
#pragma unroll
            for (int c = 2; c < 8; ++c) x[c] = 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = (f < F && t < T && c < C) ? (float)src[(int64_t)t * d.s0_t + c] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m >= M) continue;
            float o = bias[m];
#pragma unroll
            for (int c = 0; c < 8; ++c) o += w[m][c] * x[c];
            if (d.act == AERO_ACT_RELU) o = fmaxf(o, 0.f);
            else if (d.act == AERO_ACT_GELU) o = aero_gelu(o);
            Os[(tl * AERO_TINY_TF + fl) * M + m] = (h16)o;
        }
    }
    __syncthreads();
    // phase 2: every step's [rows][M] run is contiguous in LDS and in the destination
    const int nf = F - f0 < AERO_TINY_TF ? F - f0 : AERO_TINY_TF;
    const int run = nf * M;                                       // halves per step
    const int ndw = (run + 1) >> 1;
    h16* dbase = (h16*)d.dst + (int64_t)b * d.d_b + (int64_t)f0 * d.d_f;
    const bool al = ((((uintptr_t)dbase) | (uintptr_t)(d.d_t * 2)) & 3) == 0 && (run & 1) == 0;
    if (al && (run & 7) == 0 && ((M * AERO_TINY_TF) & 7) == 0 && ((((uintptr_t)dbase) | (uintptr_t)(d.d_t * 2)) & 15) == 0) {
        const int nv = run >> 3;                                  // 16-byte pieces of a step's run (full tiles of the first FTB: 20)
        for (int idx = tid; idx < AERO_TINY_TT * nv; idx += 256) {
            const int tl = idx / nv, v = idx - tl * nv;
            const int t = t0 + tl;
            if (t < T) *(h16x8*)(dbase + (int64_t)t * d.d_t + v * 8) = *(const h16x8*)(Os + tl * AERO_TINY_TF * M + v * 8);
        }
        return;
    }
    for (int idx = tid; idx < AERO_TINY_TT * ndw; idx += 256) {
        const int tl = idx / ndw, wd = idx - tl * ndw;
        const int t = t0 + tl;
        if (t >= T) continue;
        const h16* lp = Os + tl * AERO_TINY_TF * M + wd * 2;
        h16* gp = dbase + (int64_t)t * d.d_t + wd * 2;
        if (al) {
            *(uint32_t*)gp = *(const uint32_t*)lp;
        } else {
            gp[0] = lp[0];
            if (wd * 2 + 1 < run) gp[1] = lp[1];
        }
    }
}

// taps on a regular (frequency x time) grid?  fills the grid parameters of AeroConvK
// ------------------------------------------------------------------------------------------------------
// Last decoder ConvTranspose ([2*fs, 1] kernel, stride [fs, 1], fs * M <= 8 output values per source row and tap half:
// 96 channels -> 4 rows x (re, im)) from the input side with the second tap CARRIED in the accumulator, so that every
// source row is read exactly once (the stream kernel above reads rows q and q-1 per item: 2x the source, 796 MB fetched
// against 393 MB of activations).  A wave owns 64 time steps of one clip and walks source rows upwards:
//   * the 16 MFMA rows hold both taps of the row: 8 "current" rows (tap 0 -> output rows fs*q .. fs*q+fs-1) and 8 "carry"
//     rows (tap 1 -> output rows of q+1); the two weight images swap halves from one step to the next, so the carry rows of
//     step q ARE the current rows of step q+1 -- the MFMA adds the new tap-0 products onto them, no cross-lane movement;
//   * after a step the lanes holding the finished rows store them and clear their accumulators (next step's carry);
//   * a wave stages its 64 steps x C channels of a row with direct global->LDS copies (quads of lanes read whole 64-byte
//     sectors; the swizzle goes on the source address) into its own LDS tile -- no block barriers -- and issues the copy of
//     the NEXT row as soon as this row's fragments are in registers, under this row's MFMAs and stores.
//     (First version: fragments loaded straight from global memory in MFMA layout, lane = step: 16 lanes x 192-byte stride per
//     request, 185 us -- slower than the 2x-traffic stream kernel at 176 us.)
// Rows are cut into chunks of QC per wave for parallelism (one extra row read per chunk to start the carry).
template <int NCH>
__global__ __launch_bounds__(256) void aero_convtr_carry_kernel(AeroConvK p, int QC) {
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int M = d.M, T = d.T, FS = d.fstride;
    const int NR = (d.Fout + FS - 1) / FS;
    const int nseg = (T + 63) >> 6;
    const int nch = (NR + QC - 1) / QC;
    int item = aero_uniform((int)blockIdx.x * 4 + wave);
    if (item >= d.B * nch * nseg) return;
    const int seg = item % nseg;
    item /= nseg;
    const int ch = item % nch;
    const int b = item / nch;
    const int q_lo = ch * QC;
    const int q_hi = q_lo + QC < NR ? q_lo + QC : NR;
    // weight fragments of the two parities: parity 0 = rows 0-7 current (tap 0), rows 8-15 carry (tap 1); parity 1 swapped
    h16x8 W[2][NCH];
    {
        const int row = lane & 15, R8 = row & 7;
        const int r = R8 / M, m = R8 - r * M;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int tap = ((row < 8) == (par == 0)) ? 0 : 1;
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                h16x8 w = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (r < FS) w = *(const h16x8*)((const h16*)d.weight + ((int64_t)r * p.Mpad + m) * p.Ktot + tap * p.Cp + cc * 32 + (lane >> 4) * 8);
                W[par][cc] = w;
            }
        }
    }
    // per-lane output rows of a finished half: R8 = ((lane >> 4) & 1) * 4 + i
    int o_r[4], o_m[4];
    float bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R8 = ((lane >> 4) & 1) * 4 + i;
        o_r[i] = R8 / M;
        o_m[i] = R8 - o_r[i] * M;
        bv[i] = (d.bias && o_r[i] < FS) ? d.bias[o_m[i]] : 0.f;
    }
    const float bsc = d.batch_scale ? d.batch_scale[b] : 1.f;
    const float bsh = d.batch_scale ? d.batch_shift[b] : 0.f;
    const int st = (int)d.s0_t;
    const int tl = seg * 64 + (lane & 15);
    __shared__ AERO_LDS_ALIGN h16 Xs[4][NCH * 2048];
    h16* Xw = Xs[wave];
    const h16* sb = (const h16*)d.src0 + (int64_t)b * d.s0_b;
    const h16* zpv = aero_zero_page;
    const bool full = seg * 64 + 64 <= T;
    h16* dst16 = (h16*)d.dst;
    float* dst32 = (float*)d.dst;
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // copy instruction (cc, i): steps i*16 .. i*16+15, lane -> (step i*16 + lane/4, LDS slot lane%4); aero_tile_off's swizzle
    // ((-(row >> 2)) & 3 = (-(lane >> 4)) & 3 for every i) is applied to the SOURCE slot
    int c_off[4];
    bool c_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int stp = i * 16 + (lane >> 2);
        c_ok[i] = seg * 64 + stp < T;
        c_off[i] = (seg * 64 + stp) * st + (((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 3);
    }
    auto copy = [&](int fi) {
        if (fi < 0 || fi >= d.Fin) return;
        const h16* rb = sb + (int64_t)fi * d.s0_f;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
            for (int i = 0; i < 4; ++i) aero_glds16((full || c_ok[i]) ? rb + (c_off[i] + cc * 32) : zpv, Xw + cc * 2048 + i * 512);
    };
    // fragments of the staged row -> registers, then the copy of row `nxt` may overwrite the tile
    auto fetch = [&](h16x8* nb, int fi, int nxt) {
        if (fi >= 0 && fi < d.Fin) {
#ifndef AERO_EMU
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            aero_wave_sync();
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
                for (int g = 0; g < 4; ++g) nb[g * NCH + cc] = *(const h16x8*)&Xw[cc * 2048 + aero_tile_off(g * 16 + (lane & 15), lane >> 4)];
#ifndef AERO_EMU
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            aero_wave_sync();
        }
        copy(nxt);
    };
    auto step = [&](const h16x8* nb, int fi, const int par) {
        if (fi >= 0 && fi < d.Fin) {
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[par][cc], nb[g * NCH + cc], acc[g], 0, 0, 0);
        }
        const bool mine = (lane >= 32) == (par == 1);                 // this lane holds rows of the finished (current) half
        if (mine) {
            if (fi >= q_lo) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int t = tl + g * 16;
                    if (!full && t >= T) continue;
                    float x[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[g][i] + bv[i];
                        if (d.act == AERO_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (d.act == AERO_ACT_GELU) v = aero_gelu(v);
                        x[i] = v * bsc + bsh;
                    }
                    if (M == 2) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int fo = fi * FS + o_r[2 * h];
                            const int fdst = fo - d.dst_f_off;
                            if (o_r[2 * h] >= FS || fo >= d.Fout || fdst < 0 || fdst >= d.dst_F) continue;
                            const int64_t E = (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + (int64_t)t * 2;
                            if (d.dst_f32) *(f32x2*)(dst32 + E) = (f32x2){x[2 * h], x[2 * h + 1]};
                            else *(h16x2*)(dst16 + E) = (h16x2){(h16)x[2 * h], (h16)x[2 * h + 1]};
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int fo = fi * FS + o_r[i];
                            const int fdst = fo - d.dst_f_off;
                            if (o_r[i] >= FS || fo >= d.Fout || fdst < 0 || fdst >= d.dst_F) continue;
                            const int64_t E = (int64_t)b * d.d_b + (int64_t)fdst * d.d_f + (int64_t)t * M + o_m[i];
                            if (d.dst_f32) dst32[E] = x[i];
                            else dst16[E] = (h16)x[i];
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const int r0 = q_lo - 1, ns = q_hi - r0;                          // source rows r0 .. q_hi-1 (row q_lo-1 only starts the carry)
    h16x8 nb[4 * NCH];
    copy(r0);
    for (int s = 0; s < ns; s += 2) {
        fetch(nb, r0 + s, s + 1 < ns ? r0 + s + 1 : -1);
        step(nb, r0 + s, 0);
        if (s + 1 < ns) {
            fetch(nb, r0 + s + 1, s + 2 < ns ? r0 + s + 2 : -1);
            step(nb, r0 + s + 1, 1);
        }
    }
}

static bool aero_conv_regular_taps(const aero_conv_desc* d, AeroConvK* p) {
    const int n = d->ntaps;
    int nT = 1;
    while (nT < n && d->df[nT] == d->df[0]) ++nT;
    if (n % nT) return false;
    const int nF = n / nT;
    const int f_step = nF > 1 ? d->df[nT] - d->df[0] : 0;
    const int t_step = nT > 1 ? d->dt[1] - d->dt[0] : 0;
    for (int j = 0; j < n; ++j)
        if (d->df[j] != d->df[0] + (j / nT) * f_step || d->dt[j] != d->dt[0] + (j % nT) * t_step) return false;
    if (d->s0_t > 0x3fffff || d->s1_t > 0x3fffff) return false;       // 32-bit in-row offsets
    // 32-bit element offsets inside one batch item (frequency row + time shift + channel chunk)
    auto span = [&](int64_t sf, int64_t st) { return (int64_t)(d->Fin + 1) * (sf < 0 ? -sf : sf) + (int64_t)(d->T + 512) * st; };
    if ((d->src0 && span(d->s0_f, d->s0_t) > 0x7ffffff0LL) || (d->src1 && span(d->s1_f, d->s1_t) > 0x7ffffff0LL)) return false;
    p->nT = nT;
    p->f_lo = d->df[0];
    p->f_step = f_step;
    p->t_lo = d->dt[0];
    p->t_step = t_step;
    return true;
}

// AERO_CONV_GLDS=0 in the environment selects the register-staged pipeline (A/B experiments, bisecting)
static int aero_conv_use_glds() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AERO_CONV_GLDS");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

// AERO_CONV_MODE=1/2 forces 32-/64-channel K-chunks in the glds pipeline (A/B experiments); default 0 = automatic
static int aero_conv_glds_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AERO_CONV_MODE");
        v = e ? atoi(e) : 0;
    }
    return v;
}

static int aero_conv_pick_bm(int M, int Mpad) {
    const int cand[6] = {128, 96, 64, 48, 32, 16};
    int best = 128;
    long best_cost = -1;
    for (int i = 0; i < 6; ++i) {
        const int bm = cand[i];
        const int tiles = (M + bm - 1) / bm;
        if (tiles * bm > Mpad) continue;
        const long cost = (long)tiles * (bm + 40);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = bm;
        }
    }
    return best;
}

// `name` != NULL: dry run -- only report which kernel instantiation would run (profiling labels that match rocprofv3)
#define AERO_CONV_GO_GLDS(A, B, C_)                                                                                             \
    do {                                                                                                                        \
        const size_t dyn = AeroGldsGeom<A, B, C_, 4>::SMEM * sizeof(h16);                                                       \
        if (name) snprintf(name, 96, "aero_conv_glds_kernel<" #A ", " #B ", " #C_ ", %s>", d->stat_mode ? "true" : "false");    \
        else if (d->stat_mode) AERO_LAUNCH_DYN((aero_conv_glds_kernel<A, B, C_, true>), grid, block, dyn, stream, p);           \
        else AERO_LAUNCH_DYN((aero_conv_glds_kernel<A, B, C_, false>), grid, block, dyn, stream, p);                            \
    } while (0)
#define AERO_CONV_GO2(K, A, B)                                                                                 \
    do {                                                                                                       \
        if (name) snprintf(name, 96, #K "<" #A ", " #B ", %s>", d->stat_mode ? "true" : "false");             \
        else if (d->stat_mode) AERO_LAUNCH((K<A, B, true>), grid, block, stream, p);                           \
        else AERO_LAUNCH((K<A, B, false>), grid, block, stream, p);                                            \
    } while (0)

// fp32 partial-sum accumulator of a tap-split conv -> fp16 activation: dst[pos][m] = act(acc[pos][m] + bias[m])
__global__ __launch_bounds__(256) void aero_split_finish_kernel(const float* acc, int nsplit, const float* bias, int act, h16* dst, int64_t n, int M) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = acc[i];
        for (int s = 1; s < nsplit; ++s) v += acc[(int64_t)s * n + i];
        v += bias ? bias[(int)(i % M)] : 0.f;
        if (act == AERO_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == AERO_ACT_GELU) v = aero_gelu(v);
        dst[i] = (h16)v;
    }
}
static int aero_split_finish_launch(const float* acc, int nsplit, const float* bias, int act, void* dst, int64_t npos, int M, hipStream_t stream, const char** err) {
    if (!acc || !dst || npos < 1 || M < 1 || nsplit < 1) { *err = "split_finish: bad arguments"; return AERO_ERR_ARG; }
    if (act != AERO_ACT_NONE && act != AERO_ACT_RELU && act != AERO_ACT_GELU) { *err = "split_finish: unsupported act"; return AERO_ERR_UNSUPPORTED; }
    const int64_t n = npos * M;
    const int64_t want = (n + 255) / 256;
    AERO_LAUNCH(aero_split_finish_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), stream, acc, nsplit, bias, act, (h16*)dst, n, M);
    return AERO_OK;
}

// k_conv_ring.h: the software-pipelined 8-wave kernel for the wide contractions; returns true if it took the launch
AERO_XPART bool aero_conv_ring_try(const aero_conv_desc* d, AeroConvK& p, hipStream_t stream, char* name);

static int aero_conv_launch(const aero_conv_desc* d, hipStream_t stream, const char** err, char* name = nullptr) {
    if (!d || !d->weight || (!d->dst && d->stat_mode != 2 && d->tap_split <= 1 && !d->tail_w)) { *err = "conv: null weight/dst"; return AERO_ERR_ARG; }
    if (d->tail_w) {                                            // fused transposed-conv tail (aero_hip.h): only the 192-row ring tile carries it
        if (!d->tail_lo || !d->tail_hi || ((uintptr_t)d->tail_w & 15) || ((uintptr_t)d->tail_lo & 15) || ((uintptr_t)d->tail_hi & 15)) { *err = "conv: fused tail needs 16-byte aligned tail_w / tail_lo / tail_hi"; return AERO_ERR_ARG; }
        if (d->act != AERO_ACT_GLU || d->M != 192 || d->tail_cp != 96 || d->stat_mode || d->res || d->post_add || d->batch_scale || d->scatter_M ||
            d->tap_split > 1 || d->transposed || d->dst_f_off != 0 || d->dst_F != d->Fout) { *err = "conv: fused tail needs a plain GLU conv with M = 192 (tail_cp 96)"; return AERO_ERR_UNSUPPORTED; }
    }
    if (d->tap_split > 1 && (!d->split_acc || d->M <= 16 || d->stat_mode || d->scatter_M || d->res || d->post_add || d->batch_scale)) {
        *err = "conv: tap split needs split_acc, M > 16 and a plain epilogue";
        return AERO_ERR_UNSUPPORTED;
    }
    if (d->ntaps < 1 || d->ntaps > 9) { *err = "conv: ntaps must be 1..9"; return AERO_ERR_ARG; }
    if (d->C0 < 0 || d->C1 < 0 || d->C0 + d->C1 <= 0 || d->M <= 0) { *err = "conv: bad channel counts"; return AERO_ERR_ARG; }
    if (d->C1 > 0 && !d->src1) { *err = "conv: src1 NULL with C1>0"; return AERO_ERR_ARG; }
    if (d->C1 == 0 && !d->src0) { *err = "conv: no source"; return AERO_ERR_ARG; }
    if (d->fstride < 1 || d->B < 1 || d->Fout < 1 || d->T < 1) { *err = "conv: bad geometry"; return AERO_ERR_ARG; }
    if (d->act == AERO_ACT_GLU && (d->M & 1)) { *err = "conv: GLU needs even M"; return AERO_ERR_ARG; }
    if (d->act < 0 || d->act > AERO_ACT_GLU) { *err = "conv: unsupported act"; return AERO_ERR_UNSUPPORTED; }
    if (d->stat_mode < 0 || d->stat_mode > 3) { *err = "conv: bad stat_mode"; return AERO_ERR_ARG; }
    if (d->stat_mode) {
        if (!d->stats || d->stat_G < 1 || d->M % d->stat_G) { *err = "conv: statistics need stats, stat_G | M"; return AERO_ERR_ARG; }
        if (d->stat_G > 1 && ((d->M / d->stat_G) % 16 || d->act == AERO_ACT_GLU)) { *err = "conv: grouped statistics need 16-row aligned groups and un-interleaved rows"; return AERO_ERR_UNSUPPORTED; }
        if ((d->stat_mode == 1 || d->stat_mode == 2) && d->act != AERO_ACT_NONE) { *err = "conv: statistics are taken before the activation (act must be NONE)"; return AERO_ERR_ARG; }
        if (d->stat_mode == 3 && !(d->stat_count >= 1.0)) { *err = "conv: stat_count"; return AERO_ERR_ARG; }
        if (d->stat_mode == 3 && ((d->gamma == nullptr) != (d->beta == nullptr))) { *err = "conv: gamma/beta"; return AERO_ERR_ARG; }
        if (d->dst_f_off != 0 || d->dst_F != d->Fout) { *err = "conv: statistics and frequency trim do not combine"; return AERO_ERR_UNSUPPORTED; }
    }
    if (d->scatter_M) {
        if (d->scatter_M < 8 || d->scatter_M % 8 || d->M % d->scatter_M || d->scatter_stride < 1 || d->scatter_F < 1 || d->transposed ||
            d->dst_f32 || d->stat_mode || d->res || d->post_add || d->batch_scale || d->act == AERO_ACT_GLU || d->dst_f_off != 0 ||
            d->dst_F != d->Fout) { *err = "conv: row scatter needs a plain fp16 conv with scatter_M % 8 == 0"; return AERO_ERR_UNSUPPORTED; }
    }
    AeroConvK p;
    p.d = *d;
    p.tsplit = 1;
    p.Cp = (d->C0 + d->C1 + 31) / 32 * 32;
    p.cpt = p.Cp / 32;
    p.Ktot = d->ntaps * p.Cp;
    p.Mpad = (d->M + 127) / 128 * 128;
    const int bm = aero_conv_pick_bm(d->M, p.Mpad);
    p.ntt = (d->T + 127) / 128;
    p.nmt = (d->M + bm - 1) / bm;
    auto al8 = [](int64_t v) { return (v & 7) == 0; };
    int vin = (d->C0 % 8 == 0) && (d->C1 % 8 == 0);
    if (d->src0) vin = vin && al8(d->s0_b) && al8(d->s0_f) && al8(d->s0_t) && (((uintptr_t)d->src0 & 15) == 0);
    if (d->src1) vin = vin && al8(d->s1_b) && al8(d->s1_f) && al8(d->s1_t) && (((uintptr_t)d->src1 & 15) == 0);
    p.vec_in = vin;
    auto al4 = [](int64_t v) { return (v & 3) == 0; };
    int v4 = (d->C0 % 4 == 0) && (d->C1 % 4 == 0);
    if (d->src0) v4 = v4 && al4(d->s0_b) && al4(d->s0_f) && al4(d->s0_t) && (((uintptr_t)d->src0 & 7) == 0);
    if (d->src1) v4 = v4 && al4(d->s1_b) && al4(d->s1_f) && al4(d->s1_t) && (((uintptr_t)d->src1 & 7) == 0);
    p.vec4 = v4 && !vin;
    p.glds = aero_conv_use_glds();
    p.nT = 1; p.f_lo = p.f_step = p.t_lo = p.t_step = 0;
    const int Mout = d->act == AERO_ACT_GLU ? d->M / 2 : d->M;
    const int nout = d->act == AERO_ACT_GLU ? 2 : 4;
    const int esz = d->dst_f32 ? 4 : 2;
    p.vec_out = (Mout % nout == 0) && (d->d_b % nout == 0) && (d->d_f % nout == 0) && (d->d_t % nout == 0) &&
                (((uintptr_t)d->dst % (uintptr_t)(esz * nout)) == 0);
    // LDS-staged (transposed) epilogue: fp16 output whose rows can take aligned 16-byte channel vectors
    p.staged = !d->dst_f32 && (Mout % 8 == 0) && (d->d_b % 8 == 0) && (d->d_f % 8 == 0) && (d->d_t % 8 == 0) &&
               (((uintptr_t)d->dst & 15) == 0) && (bm % 16 == 0);
    if (d->res && ((d->r_b % 8) || (d->r_f % 8) || (d->r_t % 8) || ((uintptr_t)d->res & 15))) p.staged = 0;
    if (d->stat_mode == 2) p.staged = 0;
    if ((d->stat_mode == 1 || d->stat_mode == 3) && !p.staged) { *err = "conv: statistics modes need an fp16 destination with 8-channel aligned rows"; return AERO_ERR_UNSUPPORTED; }
    if (d->stat_mode && (d->batch_scale || d->post_add || d->scatter_M)) { *err = "conv: statistics modes do not combine with per-item affine / frequency embedding / row scatter"; return AERO_ERR_UNSUPPORTED; }
    const long nwg = (long)d->B * d->Fout * p.ntt * p.nmt;
    if (nwg <= 0 || nwg > 0x7fffffffL) { *err = "conv: grid too large"; return AERO_ERR_ARG; }
    dim3 grid((unsigned)nwg), block(256);
    static int dbg = -1;
    if (dbg < 0) dbg = getenv("AERO_CONV_DEBUG") ? 1 : 0;
    if (dbg && !name)
        fprintf(stderr, "[aero_conv] M=%d C0=%d C1=%d ntaps=%d B=%d Fin=%d Fout=%d T=%d tr=%d fs=%d act=%d vec_in=%d f32=%d res=%d post=%d s0=%d\n",
                d->M, d->C0, d->C1, d->ntaps, d->B, d->Fin, d->Fout, d->T, d->transposed, d->fstride, d->act, p.vec_in, d->dst_f32,
                d->res != nullptr, d->post_add != nullptr, d->src0 != nullptr);
    // few channels in, few out, frequency-major destination: the transposing pointwise kernel
    if (!d->scatter_M && d->ntaps == 1 && d->df[0] == 0 && d->dt[0] == 0 && !d->transposed && d->fstride == 1 && d->C1 == 0 && d->src0 && d->C0 <= 8 &&
        d->M <= 8 && d->act != AERO_ACT_GLU && !d->res && !d->post_add && !d->batch_scale && !d->stat_mode && !d->dst_f32 &&
        d->d_f == d->M && d->d_t == (int64_t)d->Fout * d->M && d->dst_f_off == 0 && d->dst_F == d->Fout && d->Fin == d->Fout &&
        getenv("AERO_CONV_TINY_OFF") == nullptr) {
        p.ntt = (d->T + AERO_TINY_TT - 1) / AERO_TINY_TT;
        p.nmt = (d->Fout + AERO_TINY_TF - 1) / AERO_TINY_TF;
        const long nb = (long)d->B * p.nmt * p.ntt;
        if (nb > 0x7fffffffL) { *err = "conv: grid too large"; return AERO_ERR_ARG; }
        if (name) snprintf(name, 96, "aero_conv_tiny_kernel");
        else AERO_LAUNCH(aero_conv_tiny_kernel, dim3((unsigned)nb), block, stream, p);
        return AERO_OK;
    }
    static int skinny = -1;
    if (skinny < 0) { const char* e = getenv("AERO_CONV_SKINNY"); skinny = (e && e[0] == '0') ? 0 : 1; }
    // lean streaming form: one aligned source, regular taps, <= 6 chunks per segment, dense destination
    static int stream_on = -1;
    if (stream_on < 0) { const char* e = getenv("AERO_CONV_STREAM"); stream_on = (e && e[0] == '0') ? 0 : 1; }
    if (skinny && stream_on && !d->scatter_M && p.vec_in && d->src0 && d->C1 == 0 && d->act != AERO_ACT_GLU && !d->res && !d->post_add && !d->stat_mode) {
        const int stack = d->transposed ? d->fstride : 1;
        const int nk = d->ntaps * p.cpt;
        const bool dense = d->d_t == d->M && (((uintptr_t)d->dst & 3) == 0);
        const bool small = (int64_t)d->B * d->s0_b < 0x7fffffffLL && (int64_t)(d->T + 64) * d->s0_t + p.Cp < 0x7fffffffLL &&
                           (int64_t)d->B * d->Fout * ((d->T + 63) / 64) < 0x7fffffffLL;
        if (stack * d->M <= 16 && nk <= AERO_STREAM_SLOTS && dense && small && p.Ktot + 8 <= AERO_SKINNY_WMAX &&
            aero_conv_regular_taps(d, &p) && (!d->transposed || (p.f_step == -1 && p.f_lo == 0))) {
            // last decoder ConvTranspose: carried-tap form, every source row read once
            static int carry_on = -1;
            if (carry_on < 0) { const char* e = getenv("AERO_CONVTR_CARRY"); carry_on = (e && e[0] == '0') ? 0 : 1; }
            if (carry_on && d->transposed && d->ntaps == 2 && p.nT == 1 && p.t_lo == 0 && d->fstride * d->M <= 8 && d->C0 % 32 == 0 && d->C0 <= 128 &&
                d->C0 == p.Cp && (d->s0_t % 8) == 0 && (d->s0_f % 8) == 0 && (d->s0_b % 8) == 0 && (((uintptr_t)d->src0 & 15) == 0) &&
                (d->M != 2 || ((d->d_b % 2) == 0 && (d->d_f % 2) == 0 && (((uintptr_t)d->dst & 7) == 0)))) {
                const int NRq = (d->Fout + d->fstride - 1) / d->fstride;
                const int nsegq = (d->T + 63) / 64;
                // rows per wave: as few chunks as give every CU a block (each chunk re-reads one row to start its carry).  Measured at
                // B = 64 (8 segments, 65 row groups): 2 chunks 104 us, 3: 108-114, 5: 121, 9: 107-120 -- more resident waves buy nothing,
                // the kernel sits at ~4.4 TB/s either way, and partial last rounds cost
                int nchq = (int)((256L * 4 + (long)d->B * nsegq - 1) / ((long)d->B * nsegq));
                if (nchq > NRq / 4) nchq = NRq / 4;
                if (nchq < 1) nchq = 1;
                int QC = (NRq + nchq - 1) / nchq;
                { const char* e = getenv("AERO_CARRY_QC"); if (e) QC = atoi(e); }
                const long nitq = (long)d->B * nsegq * ((NRq + QC - 1) / QC);
                const int ncc = d->C0 / 32;
                if (name) snprintf(name, 96, "aero_convtr_carry_kernel<%d>", ncc);
                else if (ncc == 1) AERO_LAUNCH(aero_convtr_carry_kernel<1>, dim3((unsigned)((nitq + 3) / 4)), block, stream, p, QC);
                else if (ncc == 2) AERO_LAUNCH(aero_convtr_carry_kernel<2>, dim3((unsigned)((nitq + 3) / 4)), block, stream, p, QC);
                else if (ncc == 3) AERO_LAUNCH(aero_convtr_carry_kernel<3>, dim3((unsigned)((nitq + 3) / 4)), block, stream, p, QC);
                else AERO_LAUNCH(aero_convtr_carry_kernel<4>, dim3((unsigned)((nitq + 3) / 4)), block, stream, p, QC);
                return AERO_OK;
            }
            p.nmt = AERO_STREAM_SLOTS / nk;                       // segments per wave item
            const int nseg = (d->T + 63) / 64;
            const int NR = d->transposed ? (d->Fout + d->fstride - 1) / d->fstride : d->Fout;
            const long nitems = (long)d->B * NR * ((nseg + p.nmt - 1) / p.nmt);
            const long want = (nitems + 3) / 4;
            const long nb = want < 256 * 2 ? want : 256 * 2;
            if (name) snprintf(name, 96, "aero_conv_stream_kernel");
            else AERO_LAUNCH(aero_conv_stream_kernel, dim3((unsigned)nb), block, stream, p);
            return AERO_OK;
        }
        p.nT = 1; p.f_lo = p.f_step = p.t_lo = p.t_step = 0;
    }
    if (skinny && !d->scatter_M && d->M <= 16 && (d->transposed ? d->fstride : 1) * (p.Ktot + 8) <= AERO_SKINNY_WMAX && d->act != AERO_ACT_GLU &&
        !d->res && !d->post_add && !d->stat_mode) {
        const int nkmax = d->ntaps * p.cpt;                    // worst-case K-chunks per 64-step segment
        p.nmt = nkmax >= AERO_SKINNY_SLOTS ? 1 : AERO_SKINNY_SLOTS / nkmax;      // segments per wave item
        const int nseg = (d->T + 63) / 64;
        const long nitems = (long)d->B * d->Fout * ((nseg + p.nmt - 1) / p.nmt);
        const long want = (nitems + 3) / 4;
        const long nb = want < 256 * 2 ? want : 256 * 2;      // persistent: 2 blocks (8 waves x 24 KiB of loads in flight) per CU
        int vw = 8;                                             // widest load piece the layout allows
        auto fits = [&](int w) {
            bool ok = (d->C0 % w == 0) && (d->C1 % w == 0);
            if (d->src0) ok = ok && d->s0_b % w == 0 && d->s0_f % w == 0 && d->s0_t % w == 0 && ((uintptr_t)d->src0 % (2 * w)) == 0;
            if (d->src1) ok = ok && d->s1_b % w == 0 && d->s1_f % w == 0 && d->s1_t % w == 0 && ((uintptr_t)d->src1 % (2 * w)) == 0;
            return ok;
        };
        while (vw > 1 && !fits(vw)) vw >>= 1;
        if (name) snprintf(name, 96, "aero_conv_skinny_kernel<%d>", vw);
        else if (vw == 8) AERO_LAUNCH(aero_conv_skinny_kernel<8>, dim3((unsigned)nb), block, stream, p);
        else if (vw == 4) AERO_LAUNCH(aero_conv_skinny_kernel<4>, dim3((unsigned)nb), block, stream, p);
        else if (vw == 2) AERO_LAUNCH(aero_conv_skinny_kernel<2>, dim3((unsigned)nb), block, stream, p);
        else AERO_LAUNCH(aero_conv_skinny_kernel<1>, dim3((unsigned)nb), block, stream, p);
        return AERO_OK;
    }
    if (d->scatter_M && !(p.staged && p.vec_in && p.glds)) { *err = "conv: row scatter needs aligned fp16 operands (direct-to-LDS pipeline)"; return AERO_ERR_UNSUPPORTED; }
    if (p.vec_in && p.glds && aero_conv_regular_taps(d, &p)) {
        // 64-channel chunks pay off only for the big compute-bound contractions (measured: +5 % on the decoder 3x3
        // convs, -2 % on the whole model if used everywhere because two 64-KiB stages halve the blocks per CU)
        const int mode = aero_conv_glds_mode();               // 0 auto, 1 = KC 32, 2 = KC 64 where legal
        const bool k64_ok = (p.Cp % 64 == 0);
        // ... and for the long, thin contractions that cannot fill the chip (FTB Conv1d over time: 250 blocks x 360 chunks, one
        // block per CU, each chunk a full copy -> barrier -> MFMA round trip): half as many, twice as long chunks (185 -> 150 us)
        const bool thin = p.Ktot >= 2048 && (long)d->B * d->Fout * p.ntt * ((d->M + bm - 1) / bm) <= 512;
        const bool k64 = mode == 2 ? k64_ok : (mode == 1 ? false : (k64_ok && ((bm >= 96 && p.Ktot >= 1024) || thin)));
        // 256-/192-row tiles (8 waves) for the wide compute-bound contractions; AERO_CONV_BM256=0 disables (A/B),
        // =1 only the 256-row tile.  KC 32 here: two 48-KiB blocks (16 waves) per CU measured 937 TF/s on the first
        // decoder layer vs 872 with one 96-KiB KC-64 block and 860 for the 128-row KC-64 tile.
        if (d->tap_split > 1) {                                 // (4-wave tiles only: the launch is block-starved by construction)
            if (p.nT % d->tap_split) { *err = "conv: tap split must divide the time taps"; return AERO_ERR_UNSUPPORTED; }
            p.tsplit = d->tap_split;
            grid = dim3(grid.x * (unsigned)p.tsplit);
        }
        if (p.tsplit == 1 && aero_conv_ring_try(d, p, stream, name)) return AERO_OK;
        if (d->tail_w) { *err = "conv: fused tail needs the 192-row software-pipelined tile (3x3 taps, tiled weight image, K >= 768)"; return AERO_ERR_UNSUPPORTED; }
        static int wide = -1;
        if (wide < 0) { const char* e = getenv("AERO_CONV_BM256"); wide = e ? atoi(e) : 2; }
        // shortest contraction the 8-wave 192-row tile takes (AERO_CONV_KMIN192, A/B; 768 until round 4): the two-source 1x1 conv of the
        // third encoder's FTB (K = 384) 89 -> 78 us, the second decoder's transposed conv (K = 384) unchanged
        static int kmin192 = -1;
        if (kmin192 < 0) { const char* e = getenv("AERO_CONV_KMIN192"); kmin192 = e ? atoi(e) : 384; }
        const int wbm = p.tsplit > 1 ? 0 : (wide >= 1 && d->M % 256 == 0 && p.Ktot >= 1024) ? 256
                        : (wide >= 2 && d->M % 192 == 0 && p.Ktot >= kmin192) ? 192 : 0;
        if (wbm) {
            p.nmt = d->M / wbm;
            grid = dim3((unsigned)((long)d->B * d->Fout * p.ntt * p.nmt));
            block = dim3(512);
            const bool st = d->stat_mode != 0;
            if (name) snprintf(name, 96, "aero_conv_glds8_kernel<%d, 32, %s>", wbm / 64, st ? "true" : "false");
            else if (wbm == 256) {
                const size_t dyn = AeroGldsGeom<4, 4, 32, 8>::SMEM * sizeof(h16);
                if (st) AERO_LAUNCH_DYN((aero_conv_glds8_kernel<4, 32, true>), grid, block, dyn, stream, p);
                else AERO_LAUNCH_DYN((aero_conv_glds8_kernel<4, 32, false>), grid, block, dyn, stream, p);
            } else {
                const size_t dyn = AeroGldsGeom<3, 4, 32, 8>::SMEM * sizeof(h16);
                if (st) AERO_LAUNCH_DYN((aero_conv_glds8_kernel<3, 32, true>), grid, block, dyn, stream, p);
                else AERO_LAUNCH_DYN((aero_conv_glds8_kernel<3, 32, false>), grid, block, dyn, stream, p);
            }
            return AERO_OK;
        }
        if (k64) {
            switch (bm) {
                case 128: AERO_CONV_GO_GLDS(4, 2, 64); break;
                case 96: AERO_CONV_GO_GLDS(3, 2, 64); break;
                case 64: AERO_CONV_GO_GLDS(4, 1, 64); break;
                case 48: AERO_CONV_GO_GLDS(3, 1, 64); break;
                case 32: AERO_CONV_GO_GLDS(2, 1, 64); break;
                default: AERO_CONV_GO_GLDS(1, 1, 64); break;
            }
        } else {
            switch (bm) {
                case 128: AERO_CONV_GO_GLDS(4, 2, 32); break;
                case 96: AERO_CONV_GO_GLDS(3, 2, 32); break;
                case 64: AERO_CONV_GO_GLDS(4, 1, 32); break;
                case 48: AERO_CONV_GO_GLDS(3, 1, 32); break;
                case 32: AERO_CONV_GO_GLDS(2, 1, 32); break;
                default: AERO_CONV_GO_GLDS(1, 1, 32); break;
            }
        }
        return AERO_OK;
    }
    if (d->scatter_M) { *err = "conv: row scatter needs a regular tap grid"; return AERO_ERR_UNSUPPORTED; }
    if (d->tap_split > 1) { *err = "conv: tap split needs aligned fp16 operands on a regular tap grid"; return AERO_ERR_UNSUPPORTED; }
    if (d->tail_w) { *err = "conv: fused tail needs aligned fp16 operands on a regular tap grid"; return AERO_ERR_UNSUPPORTED; }
    switch (bm) {
        case 128: AERO_CONV_GO2(aero_conv_kernel, 4, 2); break;
        case 96: AERO_CONV_GO2(aero_conv_kernel, 3, 2); break;
        case 64: AERO_CONV_GO2(aero_conv_kernel, 4, 1); break;
        case 48: AERO_CONV_GO2(aero_conv_kernel, 3, 1); break;
        case 32: AERO_CONV_GO2(aero_conv_kernel, 2, 1); break;
        default: AERO_CONV_GO2(aero_conv_kernel, 1, 1); break;
    }
    return AERO_OK;
}
