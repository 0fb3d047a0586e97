// k_lstm.h -- persistent recurrent kernel for one bidirectional LSTM layer (reference
// modules.py:28,46: nn.LSTM inside BLSTM; framing/stitching index math of modules.py:36-62 and
// models/utils.py:22-35 is folded into the addressing, so frames are never materialised).
//
// A block owns 16 sequences of one direction for all W steps.  W_hh lives in VGPRs as MFMA
// A-fragments for the whole kernel (rows permuted to 4*j+gate so that one lane's four accumulator
// registers are the i,f,g,o pre-activations of one (hidden unit, sequence) pair: the cell update is
// lane-local, c stays in fp32 registers).  h_{t-1} is the MFMA B operand, double-buffered in LDS as
// fp16 [16 seq][KP]; one barrier per step.  The input projection x_t W_ih^T + b arrives precomputed
// (fp16) and seeds the accumulator; the next step's slice is prefetched during the current step.
// Latency-bound by construction (W dependent steps); MFMA is used for the 4H x H x 16 step GEMM.
#pragma once
#include <stdlib.h>

#include <type_traits>
#ifndef AERO_LSTM_PRESCALE
#define AERO_LSTM_PRESCALE 1                 /* tools/dbg A/B: 0 = gate constants applied per step */
#endif
#ifndef AERO_LSTM_IMM
#define AERO_LSTM_IMM 1                      /* tools/dbg A/B: 0 = ring slots addressed from the run-time step index */
#endif

#include "aero_common.h"

struct AeroLstmK {
    aero_lstm_desc d;
    int MP, KP;
    int dv;              // the divisor of the sequence -> (row, frame) map: nframes (frame-minor) or rows = nseq / nframes (frame-major)
};

// smallest instantiated (NW waves, TPW gate tiles per wave, KT k-steps) with 16*NW*TPW >= 4H and 32*KT >= H.
// The step is a serial chain (MFMA -> gates -> LDS -> barrier), so the gate rows are spread over up to 8 waves
// (2 per SIMD) to shorten the per-wave work of each step.
static inline int aero_lstm_pick(int H, int* nw, int* tpw, int* kt) {
    const int cfg[6][3] = {{4, 1, 1}, {4, 2, 1}, {6, 2, 2}, {8, 2, 2}, {8, 3, 3}, {8, 4, 4}};
    for (int i = 0; i < 6; ++i) {
        if (16 * cfg[i][0] * cfg[i][1] >= 4 * H && 32 * cfg[i][2] >= H) {
            *nw = cfg[i][0];
            *tpw = cfg[i][1];
            *kt = cfg[i][2];
            return 0;
        }
    }
    return -1;
}

// KTI = 0: the input projection arrives precomputed (xproj).  KTI > 0: the projection W_ih x_t is FUSED: W_ih lives in
// VGPRs next to W_hh, x_t is loaded straight into MFMA B-fragments (16-byte loads, one step ahead) and the KTI extra
// MFMAs per gate tile are issued before the barrier of the previous step -- the 8H-channel pre-activation tensor
// (4x the size of x for layer 2) never exists in HBM and the separate projection launch disappears.
// k-steps of the fused input projection for (H, in_ch), or -1 if that combination is not instantiated
static inline int aero_lstm_kti(int H, int in_ch) {
    int nw, tpw, kt;
    if (aero_lstm_pick(H, &nw, &tpw, &kt)) return -1;
    const int need = (in_ch + 31) / 32;
    if (nw == 4 && tpw == 1) return need <= 1 ? 1 : -1;
    if (nw == 4 && tpw == 2) return need <= 1 ? 1 : (need <= 2 ? 2 : -1);
    if (nw == 6) return need <= 2 ? 2 : (need <= 3 ? 3 : -1);
    if (tpw == 2) return need <= 2 ? 2 : (need <= 4 ? 4 : -1);
    if (tpw == 3) return need <= 3 ? 3 : (need <= 6 ? 6 : -1);
    return -1;
}

// sequence index -> (row r of the [R, T] input, frame k): frame-minor (seq = r * nframes + k: the order `unfold` + reshape gives,
// modules.py:49-51) or, with aero_lstm_desc.frame_major, frame-major (seq = k * R + r): a block's 16 sequences then belong to ONE frame
// (two where R % 16 != 0), which is what lets the stitching layer stop at the last step its frame keeps (round 6)
static __device__ __forceinline__ void aero_lstm_rk(const AeroLstmK& p, int seq, int& r, int& k) {
    const int q = seq / p.dv, rem = seq - q * p.dv;        // (one division by a launch constant, as the frame-minor form always had)
    r = p.d.frame_major ? rem : q;
    k = p.d.frame_major ? q : rem;
}

template <int NW, int TPW, int KT, int KTI>
__global__ __launch_bounds__(NW * 64) void aero_lstm_kernel(AeroLstmK p) {
    constexpr int KP = KT * 32;
    constexpr int MP = NW * TPW * 16;
    constexpr int NT = NW * 64;
    __shared__ AERO_LDS_ALIGN h16 hbuf[2][16 * KP];
    const aero_lstm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * 16;
    const int H = d.H, W = d.W, H4 = 4 * d.H, H8 = 8 * d.H, H2 = 2 * d.H;
    const int q = lane >> 4, col = lane & 15;
    const h16* whh = (const h16*)d.whh + (int64_t)dir * MP * KP;
    const h16* xproj = (const h16*)d.xproj;
    const h16* xbias = (const h16*)d.xbias;
    h16* out = (h16*)d.out;

    h16x8 wf[TPW][KT];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            wf[i][kt] = *(const h16x8*)(whh + (int64_t)((wave * TPW + i) * 16 + col) * KP + kt * 32 + q * 8);

    constexpr int KI = KTI > 0 ? KTI : 1;
    h16x8 wi[TPW][KI];
    f32x4 bias4[TPW];
    if (KTI > 0) {
        const h16* wih = (const h16*)d.wih + (int64_t)dir * MP * (KTI * 32);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
#pragma unroll
            for (int kt = 0; kt < KI; ++kt)
                wi[i][kt] = *(const h16x8*)(wih + (int64_t)((wave * TPW + i) * 16 + col) * (KTI * 32) + kt * 32 + q * 8);
            const int rr = (wave * TPW + i) * 16 + q * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[i][r] = (rr + r < H4) ? d.bias[dir * H4 + rr + r] : 0.f;
        }
    }

    for (int idx = tid; idx < 2 * 16 * KP; idx += NT) (&hbuf[0][0])[idx] = (h16)0;

    float c[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) c[i] = 0.f;

    // this lane's sequence (MFMA column) and its input addressing
    const int seq = seq0 + col;
    const bool seq_ok = seq < d.nseq;
    int64_t in_base = 0;   // position index of step tau = 0
    int t_first = 0;       // in_mode 1: absolute time of tau = 0
    if (d.in_mode == 1) {
        int r, k;
        aero_lstm_rk(p, seq, r, k);
        t_first = k * d.S;
        in_base = (int64_t)r * d.T + t_first;
    } else {
        in_base = (int64_t)seq * W;
    }
    auto load_xp = [&](int tau, h16x4* xp) {
        const bool pad = (d.in_mode == 1) && (t_first + tau >= d.T);
        const h16* rowp = pad ? xbias : xproj + (in_base + tau) * H8;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int rr = (wave * TPW + i) * 16 + q * 4;      // first of this lane's 4 gate rows
            h16x4 v = (h16x4){0, 0, 0, 0};
            if (seq_ok && rr < H4) v = *(const h16x4*)(rowp + dir * H4 + rr);
            xp[i] = v;
        }
    };

    // fused mode: x_tau as MFMA B-fragments (lane: sequence col, channels kt*32 + q*8 .. +7)
    const h16* xin = (const h16*)d.x;
    const bool xvec = (d.in_ch % 8 == 0) && (d.x_pitch % 8 == 0);
    auto load_x = [&](int tau, h16x8* xb) {
        const bool pad = ((d.in_mode == 1) && (t_first + tau >= d.T)) || !seq_ok;
        const h16* rowp = xin + (in_base + tau) * d.x_pitch;
#pragma unroll
        for (int kt = 0; kt < KI; ++kt) {
            const int c = kt * 32 + q * 8;
            h16x8 v = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (!pad && c < d.in_ch) {
                if (xvec) {
                    v = *(const h16x8*)(rowp + c);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (c + e < d.in_ch) v[e] = rowp[c + e];
                }
            }
            xb[kt] = v;
        }
    };
    auto project = [&](const h16x8* xb, f32x4* ax) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) ax[i] = bias4[i];
#pragma unroll
        for (int kt = 0; kt < KI; ++kt)
#pragma unroll
            for (int i = 0; i < TPW; ++i) ax[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wi[i][kt], xb[kt], ax[i], 0, 0, 0);
    };

    // per-thread slots of the cooperative output store: slot idx -> (local sequence sl, element e)
    const int vecs = (H % 8 == 0) ? H / 8 : 0;
    const int per = vecs ? vecs : H;
    constexpr int ST_ITERS = (16 * (KT * 32) + NT - 1) / NT;      // covers per <= KP (scalar path) as well
    bool st_ok[ST_ITERS];
    int st_sl[ST_ITERS], st_e[ST_ITERS], st_lo[ST_ITERS], st_hi[ST_ITERS], st_t0[ST_ITERS];
    int64_t st_base[ST_ITERS];
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
        const int idx = tid + it * NT;
        const int sl = idx / per, e = idx - sl * per;
        const int s2 = seq0 + sl;
        st_ok[it] = idx < 16 * per && s2 < d.nseq;
        st_sl[it] = sl;
        st_e[it] = e;
        st_lo[it] = st_hi[it] = st_t0[it] = 0;
        st_base[it] = 0;
        if (st_ok[it]) {
            if (d.out_mode == 1) {
                int r, k;
                aero_lstm_rk(p, s2, r, k);
                const int lim = d.S / 2;
                st_lo[it] = (k == 0) ? 0 : lim;
                st_hi[it] = (k == d.nframes - 1 && k != 0) ? W : W - lim;
                st_t0[it] = k * d.S;
                st_base[it] = (int64_t)r * d.T;
            } else {
                st_base[it] = (int64_t)s2 * W;
            }
        }
    }

    h16x4 xp_cur[TPW], xp_nxt[TPW];
    h16x8 xb[KI];
    f32x4 accx[TPW];                               // fused mode: bias + W_ih x_tau for the step about to run
    if (KTI > 0) {
        load_x(dir ? W - 1 : 0, xb);
        project(xb, accx);
        if (W > 1) load_x(dir ? W - 2 : 1, xb);
    } else {
        load_xp(dir ? W - 1 : 0, xp_cur);
    }
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < W; ++step) {
        const int tau = dir ? W - 1 - step : step;
        if (KTI == 0 && step + 1 < W) load_xp(dir ? tau - 1 : tau + 1, xp_nxt);
        h16x8 bf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) bf[kt] = *(const h16x8*)&hbuf[cur][col * KP + kt * 32 + q * 8];
        f32x4 accs[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (KTI > 0) accs[i] = accx[i];
            else accs[i] = (f32x4){(float)xp_cur[i][0], (float)xp_cur[i][1], (float)xp_cur[i][2], (float)xp_cur[i][3]};
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)       // k outer: TPW independent accumulator chains interleave on the matrix pipe
#pragma unroll
            for (int i = 0; i < TPW; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i][kt], bf[kt], accs[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const f32x4 acc = accs[i];
            const int j = (wave * TPW + i) * 4 + q;
            const float ig = aero_sigmoid(acc[0]), fg = aero_sigmoid(acc[1]);
            const float gg = aero_tanh(acc[2]), og = aero_sigmoid(acc[3]);
            c[i] = fg * c[i] + ig * gg;
            const float h = og * aero_tanh(c[i]);
            if (j < H) hbuf[cur ^ 1][col * KP + j] = (h16)h;
            if (d.save_gates && j < H) {                       // training-mode forward: what aero_lstm_bwd needs (k_train.h)
                const int64_t sb = ((int64_t)blockIdx.x * 2 + dir) * W + tau;
                *(h16x4*)((h16*)d.save_gates + sb * H * 64 + ((int64_t)j * 16 + col) * 4) = (h16x4){(h16)ig, (h16)fg, (h16)gg, (h16)og};
                d.save_c[sb * H * 16 + j * 16 + col] = c[i];
            }
        }
        if (KTI > 0 && step + 1 < W) {
            // projection of the NEXT step (independent of h): fills the matrix pipe while other waves reach the barrier
            project(xb, accx);
            if (step + 2 < W) load_x(dir ? tau - 2 : tau + 2, xb);
        }
        __syncthreads();
        // cooperative, coalesced store of h_tau for the block's 16 sequences (addresses precomputed above)
#pragma unroll
        for (int it = 0; it < ST_ITERS; ++it) {
            if (!st_ok[it]) continue;
            int64_t opos;
            if (d.out_mode == 1) {
                const int t = st_t0[it] + tau;
                if (tau < st_lo[it] || tau >= st_hi[it] || t >= d.T) continue;
                opos = st_base[it] + t;
            } else {
                opos = st_base[it] + tau;
            }
            if (vecs) *(h16x8*)(out + opos * H2 + dir * H + st_e[it] * 8) = *(const h16x8*)&hbuf[cur ^ 1][st_sl[it] * KP + st_e[it] * 8];
            else out[opos * H2 + dir * H + st_e[it]] = hbuf[cur ^ 1][st_sl[it] * KP + st_e[it]];
        }
        if (KTI == 0) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) xp_cur[i] = xp_nxt[i];
        }
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------
// Ring form of the fused kernel (the one the engine runs).  The ISA of the kernel above showed why a step cost
// 1.1-2.5 us against a ~0.5 us arithmetic floor: x_t was fetched one step ahead and h_t stored every step, and because
// the stores are conditional the compiler can only wait with `s_waitcnt vmcnt(0)` -- every step paid a full HBM round
// trip (load latency AND the previous step's store acknowledgements).  Here all global traffic moves in GROUPS of G
// steps through LDS rings:
//   * x for group g+1 is loaded into a few VGPRs at the start of group g and parked in LDS at its end (G steps of
//     latency cover); the step reads it back as MFMA B-fragments with one ds_read_b128 per k-step;
//   * h_t goes to a 2G-slot LDS ring (it is the recurrence operand anyway); group g-1 is written to HBM in one
//     coalesced burst at the start of group g;
//   * so vmcnt is waited for ONCE per group, G steps after the traffic was issued, and the step loop itself contains
//     only LDS reads, MFMAs, gate math, one ds_write and one barrier.
// (global_load_lds was tried for the x copy: the compiler then waits vmcnt(0) before every barrier.)
// Rows of both rings are padded by 16 bytes: conflict-free ds_read_b128 for every KP/KPI used.
// Gate math: 5 exp + 3 rcp per (unit, sequence) instead of 5 + 5 (shared reciprocals, clamped arguments).
// The gate tiles may be spread over up to 12 waves (3 per SIMD): shorter per-wave chains, and W_ih fits in VGPRs.

template <int KT, int KTI, int G>
struct AeroLstmRingGeom {
    static constexpr int HS = KT * 32 + 8, XS = KTI * 32 + 8;
    static constexpr size_t BYTES = (size_t)(2 * G * 16 * HS + 2 * G * 16 * XS) * sizeof(h16);
};

// SAVE: the training-mode forward -- the gate activations (fp16 [H][16][4]) and the cell state (fp32 [H][16]) of every step go to
// d.save_gates / d.save_c for aero_lstm_bwd (k_train.h).  They are written straight from registers (no extra reciprocal: sigmoid(i),
// tanh(g) and sigmoid(o) fall out of the shared reciprocals of the gate math), and the step barrier becomes LDS-only (lgkmcnt + s_barrier)
// so that the stores stay in flight: with `__syncthreads()` every step waited for their acknowledgement (the step-wise kernel's 2.5 us
// per step).
// (twelve one-tile waves -- the H = 48 launches, 384 blocks on 256 CUs -- must fit TWICE on a CU: six waves per SIMD, at most 80 registers per wave; the second launch-bound argument is waves per SIMD)
template <int NW, int TPW, int KT, int KTI, int G, bool SAVE = false>
__global__ __launch_bounds__(NW * 64, (NW == 12 && TPW == 1 && !SAVE) ? 6 : 1) void aero_lstm_ring_kernel(AeroLstmK p) {
    constexpr int R = 2 * G;
    constexpr int KP = KT * 32, HS = KP + 8, KPI = KTI * 32, XS = KPI + 8, SPR = KTI * 4, NT = NW * 64;
    constexpr int NXV = (G * 16 * SPR + NT - 1) / NT;              // x vectors per thread per group
    h16* hring = (h16*)AERO_DYN_SMEM;                              // [R][16][HS]
    h16* xring = hring + R * 16 * HS;                              // [2][G][16][XS]
    const aero_lstm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * 16;
    const int H = d.H, W = d.W, H4 = 4 * d.H, H2 = 2 * d.H;
    const int q = lane >> 4, col = lane & 15;
    const h16* whh = (const h16*)d.whh + (int64_t)dir * (NW * TPW * 16) * KP;
    const h16* wih = (const h16*)d.wih + (int64_t)dir * (NW * TPW * 16) * KPI;
    const h16* xin = (const h16*)d.x;
    h16* out = (h16*)d.out;

    // Round 5: the gate pre-activations come out of the MFMAs ALREADY multiplied by the constant their exponential wants (-log2 e for the
    // sigmoids of i, f, o; +2 log2 e for tanh(g)): the weight fragments and the bias are scaled once per block while they are loaded (row
    // 4j + gate of the image is gate `row & 3`: a per-lane constant for an A fragment, a per-register one for the bias), which removes
    // four multiplies per (unit, sequence) from every one of the W steps -- the step is VALU-bound on the CUs that host two blocks.
    // The scaled weights are rounded to fp16 a second time (<= 1 ulp on top of the pack's rounding; whole-block error vs the oracle
    // unchanged at 3-5e-4, tests/op_cases.py BLSTM_TOL 1e-3).
    constexpr float L2E = 1.4426950408889634f;
#if AERO_LSTM_PRESCALE
    const float wsc = ((col & 3) == 2) ? 2.f * L2E : -L2E;          // A-fragment lane `col` holds image row 16 * tile + col
    constexpr float GS = 1.f, GG = 1.f, BS = -L2E, BG = 2.f * L2E;  // (gate-math factors left over; bias factors)
#else
    const float wsc = 1.f;
    constexpr float GS = -L2E, GG = 2.f * L2E, BS = 1.f, BG = 1.f;
#endif
    auto scaled = [&](h16x8 v) {
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)((float)v[e] * wsc);
        return o;
    };
    h16x8 wf[TPW][KT], wi[TPW][KTI];
    f32x4 bias4[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int wrow = (wave * TPW + i) * 16 + col;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) wf[i][kt] = scaled(*(const h16x8*)(whh + (int64_t)wrow * KP + kt * 32 + q * 8));
#pragma unroll
        for (int kt = 0; kt < KTI; ++kt) wi[i][kt] = scaled(*(const h16x8*)(wih + (int64_t)wrow * KPI + kt * 32 + q * 8));
        const int rr = (wave * TPW + i) * 16 + q * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[i][r] = (rr + r < H4) ? d.bias[dir * H4 + rr + r] * (r == 2 ? BG : BS) : 0.f;
    }
    for (int idx = tid; idx < R * 16 * HS / 8; idx += NT) ((h16x8*)hring)[idx] = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};

    // cooperative x copy: vector L = tid + v*NT -> (step-in-group, sequence row, 8-channel slot); the addressing is
    // recomputed per group (a handful of integer ops every G steps) rather than held in VGPRs
    // Everything about a thread's share of the group traffic that does not depend on the group is computed ONCE: which
    // (step-in-group, sequence, channel slot) it copies, the frame arithmetic (a division by the run-time frame count), the
    // LDS offsets.  PMC: the step loop ran 54 vector instructions per step and wave, 20 more than the gate math needs --
    // integer divisions and 64-bit address arithmetic of these three helpers, redone every group; on the CUs that host two
    // blocks the kernel is VALU-bound (76 % busy), so they were on the critical path.
    const bool xvec = (d.in_ch % 8 == 0) && (d.x_pitch % 8 == 0) && (((uintptr_t)xin & 15) == 0);
    h16x8 xr[NXV];
    int lx_i[NXV], lx_tmax[NXV], lx_c[NXV], lx_dst[NXV];          // tmax: tau must stay below it (-1: this thread copies nothing)
    int lx_base[NXV];                                             // (row indices: 32 bits, checked on the host -- a 64-bit pair per entry spilled to scratch)
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
        const int L = tid + v * NT;
        const int i = L / (16 * SPR), rem = L - i * (16 * SPR);
        const int row = rem / SPR, slot = rem - row * SPR;
        const int seq = seq0 + row;
        const bool ok = L < G * 16 * SPR && seq < d.nseq && slot * 8 < d.in_ch;
        lx_i[v] = i;
        lx_c[v] = slot * 8;
        lx_dst[v] = (i * 16 + row) * XS + slot * 8;
        if (d.in_mode == 1) {
            int r, k;
            aero_lstm_rk(p, seq, r, k);
            lx_tmax[v] = ok ? d.T - k * d.S : -1;
            lx_base[v] = r * d.T + k * d.S;
        } else {
            lx_tmax[v] = ok ? W : -1;
            lx_base[v] = seq * W;
        }
    }
    auto load_x = [&](int g) {          // global -> VGPR for group g (steps beyond W, padded frames, bad rows: zeros)
#pragma unroll
        for (int v = 0; v < NXV; ++v) {
            const int step = g * G + lx_i[v];
            const int tau = dir ? W - 1 - step : step;
            const bool ok = step < W && tau < lx_tmax[v];
            h16x8 z = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) {
                const h16* sp = xin + (int64_t)(lx_base[v] + tau) * d.x_pitch + lx_c[v];
                if (xvec) {
                    z = *(const h16x8*)sp;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (lx_c[v] + e < d.in_ch) z[e] = sp[e];
                }
            }
            xr[v] = z;
        }
    };
    auto park_x = [&](int buf) {        // VGPR -> LDS ring buffer
#pragma unroll
        for (int v = 0; v < NXV; ++v)
            if (tid + v * NT < G * 16 * SPR) *(h16x8*)(xring + buf * (G * 16 * XS) + lx_dst[v]) = xr[v];
    };
    // write the h rows of group g (LDS ring slots) to the stitched output
    const int vecs = (H % 8 == 0 && (((uintptr_t)out & 15) == 0)) ? H / 8 : 0;
    const int per = vecs ? vecs : H;
    constexpr int NSV = (G * 16 * KT * 4 + NT - 1) / NT;           // precomputed 16-byte pieces per thread (vecs <= KT * 4)
    const bool sg_fast = vecs != 0;
    int sg_pk[NSV], sg_tlo[NSV], sg_thi[NSV];                      // (step-in-group << 20 | LDS offset), valid range of tau (empty: thi = -1)
    int sg_base[NSV];                                             // (element offsets < 2^31: checked on the host)
#pragma unroll
    for (int k2 = 0; k2 < NSV; ++k2) {
        const int idx = tid + k2 * NT;
        const int i = idx / (16 * per), rem = idx - i * (16 * per);
        const int sl = rem / per, e = rem - sl * per;
        const int s2 = seq0 + sl;
        const bool ok = sg_fast && idx < G * 16 * per && s2 < d.nseq;
        sg_pk[k2] = (i << 20) | (sl * HS + e * 8);
        sg_tlo[k2] = 0;
        sg_thi[k2] = ok ? W : -1;
        if (d.out_mode == 1) {
            int r, kf;
            aero_lstm_rk(p, s2, r, kf);
            const int lim = d.S / 2;
            const int hi = (kf == d.nframes - 1 && kf != 0) ? W : W - lim;
            const int tl = d.T - kf * d.S;                          // t = kf*S + tau must stay below T
            sg_tlo[k2] = (kf == 0) ? 0 : lim;
            sg_thi[k2] = ok ? (hi < tl ? hi : tl) : -1;
            sg_base[k2] = (r * d.T + kf * d.S) * H2 + dir * H + e * 8;
        } else {
            sg_base[k2] = (s2 * W) * H2 + dir * H + e * 8;
        }
    }
    // Round 6: the STITCHING layer of an inference forward runs only the steps some sequence of the block keeps.  The stitch (modules.py:52-62)
    // keeps tau in [lo, hi) of a frame -- lo = 0 (first frame) or S/2, hi = W - S/2 or W (last frame), and t = k S + tau < T -- so the forward
    // direction (step = tau) may stop at max hi and the backward one (step = W - 1 - tau) at W - min lo: a quarter of the steps of a middle
    // frame in either direction.  The steps not run are exactly those whose h nothing reads: the result is bit-identical.  Block-uniform,
    // rounded up to whole groups (the unrolled group loop stays straight-line code); only with frame-major sequences (aero_lstm_desc.frame_major):
    // in the reference's frame-minor order every block mixes first, middle and last frames and would keep W anyway.
    int Wrun = W;
    if (!SAVE && d.out_mode == 1 && d.frame_major) {
        // the block's sequences seq0 .. seq0 + 15 lie in frames k0 .. k1 (consecutive; more than two only when a frame has < 16 rows)
        const int s_last = seq0 + 15 < d.nseq ? seq0 + 15 : d.nseq - 1;
        const int k0 = seq0 / p.dv, k1 = s_last / p.dv;
        const int lim = d.S / 2;
        int need = k1 > k0 + 1 ? W : 0;
        for (int kf = k0; kf <= k1 && kf <= k0 + 1; ++kf) {
            const int lo = kf == 0 ? 0 : lim;
            int hi = (kf == d.nframes - 1 && kf != 0) ? W : W - lim;
            const int tl = d.T - kf * d.S;
            hi = hi < tl ? hi : tl;
            const int n = hi <= lo ? 0 : (dir ? W - lo : hi);
            need = n > need ? n : need;
        }
        need = (need + G - 1) / G * G;
        Wrun = need < G ? G : (need < W ? need : W);
    }
    Wrun = aero_uniform(Wrun);
    auto store_group = [&](int g) {
        const int s_base = g * G;
        const int nst = Wrun - s_base < G ? Wrun - s_base : G;
        if (sg_fast) {
#pragma unroll
            for (int k2 = 0; k2 < NSV; ++k2) {
                const int i = sg_pk[k2] >> 20;
                const int step = s_base + i;
                const int tau = dir ? W - 1 - step : step;
                if (i >= nst || tau < sg_tlo[k2] || tau >= sg_thi[k2]) continue;
                *(h16x8*)(out + (sg_base[k2] + tau * H2)) = *(const h16x8*)(hring + (step & (R - 1)) * 16 * HS + (sg_pk[k2] & 0xFFFFF));
            }
            return;
        }
        for (int idx = tid; idx < nst * 16 * per; idx += NT) {
            const int i = idx / (16 * per), rem = idx - i * (16 * per);
            const int sl = rem / per, e = rem - sl * per;
            const int s2 = seq0 + sl;
            if (s2 >= d.nseq) continue;
            const int step = s_base + i;
            const int tau = dir ? W - 1 - step : step;
            int64_t opos;
            if (d.out_mode == 1) {
                int r, k;
                aero_lstm_rk(p, s2, r, k);
                const int lim = d.S / 2;
                const int lo = (k == 0) ? 0 : lim;
                const int hi = (k == d.nframes - 1 && k != 0) ? W : W - lim;
                const int t = k * d.S + tau;
                if (tau < lo || tau >= hi || t >= d.T) continue;
                opos = (int64_t)r * d.T + t;
            } else {
                opos = (int64_t)s2 * W + tau;
            }
            const h16* hp = hring + ((step & (R - 1)) * 16 + sl) * HS;
            if (vecs) *(h16x8*)(out + opos * H2 + dir * H + e * 8) = *(const h16x8*)(hp + e * 8);
            else out[opos * H2 + dir * H + e] = hp[e];
        }
    };

    float c[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) c[i] = 0.f;
    f32x4 accx[TPW];                    // bias + W_ih x for the step about to run
#if AERO_LSTM_IMM
    const int ngroups = (Wrun + G - 1) / G;
    load_x(0);
    park_x(0);
    __syncthreads();
    // Round 5: the ring-slot addressing as IMMEDIATES.  Step s = g G + i uses h-ring slot s mod 2G = (g & 1) G + i and reads slot s - 1; the
    // x ring likewise.  The step loop over i is unrolled (i a compile-time constant), so a slot is the group's parity offset (one scalar per
    // group) plus a constant that goes into the DS instruction's offset field.  Before, every step rebuilt three LDS addresses from the
    // run-time step index: 7 of the 34 vector instructions of a step on a kernel that is VALU-bound where two blocks share a CU.
    // The unrolled body is STRAIGHT-LINE code: no run-time condition around the next step's projection.  A first version kept
    // `if (i + 1 < nst) project(...)`: hipcc then carried accx through phi copies behind a `s_cbranch_execnz`, and on the taken path a
    // `v_mov_b64` read the projection MFMA's destination with ZERO wait states (the hazard recogniser had padded only the fall-through
    // path) -- wrong and run-to-run different results on the MI355X, correct on the emulator (tools/dbg/lstm_check.py).  A ragged last
    // group (W % G != 0: no reference config) takes the run-time loop.
    const int hrd0 = col * HS + q * 8;                             // this lane's B-fragment row inside a slot (h16 elements from hring)
    const int xrd0 = col * XS + q * 8;
    int hwr0[TPW];                                                 // ... its h_t element(s): + unit index
#pragma unroll
    for (int t = 0; t < TPW; ++t) hwr0[t] = col * HS + (wave * TPW + t) * 4 + q;
    bool jok[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) jok[t] = (wave * TPW + t) * 4 + q < H;
    auto project_at = [&](const h16* xs) {                         // xs: lane base incl. q * 8; constant offsets fold into the reads
        h16x8 xf[KTI];
#pragma unroll
        for (int kt = 0; kt < KTI; ++kt) xf[kt] = *(const h16x8*)(xs + kt * 32);
#pragma unroll
        for (int t = 0; t < TPW; ++t) accx[t] = bias4[t];
#pragma unroll
        for (int kt = 0; kt < KTI; ++kt)
#pragma unroll
            for (int t = 0; t < TPW; ++t) accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wi[t][kt], xf[kt], accx[t], 0, 0, 0);
    };
    // one step: h_{s-1} from `hprev` (lane base incl. q * 8), h_s to wdst[t][0]
    auto step_core = [&](const h16* hprev, h16* (&wdst)[TPW], int wofs, int s) {
        h16x8 bf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) bf[kt] = *(const h16x8*)(hprev + kt * 32);
        f32x4 accs[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) accs[t] = accx[t];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int t = 0; t < TPW; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][kt], bf[kt], accs[t], 0, 0, 0);
        // gate math of all the wave's tiles first, the h stores after it: with the store's `if (j < H)` inside the tile loop the
        // compiler sank half of each tile's math into that branch and ran the tiles strictly one after the other -- two
        // dependent exp -> rcp -> exp -> rcp chains back to back instead of interleaved
        float hq[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const f32x4 a = accs[t];                                            // (pre-scaled with AERO_LSTM_PRESCALE: see the weight load)
            constexpr float LIM = 60.f, NOLIM = -3.0e38f;
            const float ei = aero_exp2(a[0] * GS), ef = aero_exp2(a[1] * GS);
            const float eg = aero_exp2(aero_med3(a[2] * GG, NOLIM, LIM));
            const float eo = aero_exp2(a[3] * GS);
            const float r1 = aero_rcp((1.f + ei) * (eg + 1.f));
            const float igg = fmaf(eg, r1, -r1);                                        // sigmoid(i) * tanh(g)
            c[t] = fmaf(aero_rcp(1.f + ef), c[t], igg);
            const float ec = aero_exp2(aero_med3(c[t] * (2.f * L2E), NOLIM, LIM));
            const float r2 = aero_rcp((1.f + eo) * (ec + 1.f));
            hq[t] = fmaf(ec, r2, -r2);                                                  // sigmoid(o) * tanh(c)
            if constexpr (SAVE) {
                const int j = (wave * TPW + t) * 4 + q;
                if (j < H) {
                    const int tau_s = dir ? W - 1 - s : s;
                    const int64_t sb = ((int64_t)blockIdx.x * 2 + dir) * W + tau_s;
                    const float sig_i = r1 * (eg + 1.f), tanh_g = (eg - 1.f) * r1 * (1.f + ei);
                    const float sig_f = aero_rcp(1.f + ef), sig_o = r2 * (ec + 1.f);
                    *(h16x4*)((h16*)d.save_gates + sb * H * 64 + ((int64_t)j * 16 + col) * 4) = (h16x4){(h16)sig_i, (h16)sig_f, (h16)tanh_g, (h16)sig_o};
                    d.save_c[sb * H * 16 + j * 16 + col] = c[t];
                }
            }
        }
#ifndef AERO_EMU
#pragma unroll
        for (int t = 0; t < TPW; ++t) asm volatile("" : "+v"(hq[t]));                   // (keeps the math above out of the store's branch)
#endif
#pragma unroll
        for (int t = 0; t < TPW; ++t)
            if (jok[t]) wdst[t][wofs] = (h16)hq[t];
    };
    auto step_barrier = [&]() {
        if constexpr (SAVE) aero_phase_barrier();
        else __syncthreads();
    };
    const h16* hbase = hring + hrd0;                                // ONE lane base per ring: every slot offset below is a compile-time constant
    const h16* xbase = xring + xrd0;
    h16* wbase[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) wbase[t] = hring + hwr0[t];
    // a FULL group of G steps whose parity (g & 1) is the compile-time constant PAR
    auto group_full = [&](auto parc, int g) {
        constexpr int PAR = decltype(parc)::value;
        constexpr int HC = PAR * G * 16 * HS, HO = (1 - PAR) * G * 16 * HS, XC = PAR * G * 16 * XS;
        if (g + 1 < ngroups) load_x(g + 1);
        if (g > 0) store_group(g - 1);
        project_at(xbase + XC);                                         // first step of a group: x just parked
#pragma unroll
        for (int i = 0; i < G; ++i) {
            step_core(hbase + (i == 0 ? HO + (G - 1) * 16 * HS : HC + (i - 1) * 16 * HS), wbase, HC + i * 16 * HS, g * G + i);
            // input projection of the NEXT step (independent of h): issued before the barrier so its LDS reads and MFMAs fill the pipes
            // while the other waves finish their gate math; at the end of the group the next group's x goes to its ring buffer
            // (pinning the projection's MFMAs in front of the barrier -- hipcc floats them behind it, to the head of the next step's chain -- was
            // measured neutral: 141 / 164 / 166 / 197 us against 141 / 164 / 159 / 199, profiles/r05_lstm_pin_ab.txt)
            if (i + 1 < G) project_at(xbase + XC + (i + 1) * 16 * XS);
            else if (g + 1 < ngroups) park_x((g + 1) & 1);              // loads issued G steps ago
            step_barrier();
        }
    };
    const int nfull = Wrun / G;
    int g = 0;
    for (; g + 2 <= nfull; g += 2) {
        group_full(std::integral_constant<int, 0>{}, g);
        group_full(std::integral_constant<int, 1>{}, g + 1);
    }
    if (g < nfull) {
        group_full(std::integral_constant<int, 0>{}, g);
        ++g;
    }
    if (g < ngroups) {                                                  // ragged last group (W % G steps): run-time slots
        if (g > 0) store_group(g - 1);
        const int nst = Wrun - g * G;
        const int pofs = (g & 1) * (G * 16);
        const h16* hcur = hbase + pofs * HS;
        const h16* hoth = hbase + (G * 16 - pofs) * HS;
        const h16* xcur = xbase + pofs * XS;
        project_at(xcur);
        for (int i = 0; i < nst; ++i) {
            step_core(i == 0 ? hoth + (G - 1) * 16 * HS : hcur + (i - 1) * 16 * HS, wbase, (pofs + i * 16) * HS, g * G + i);
            project_at(xcur + (i + 1 < G ? i + 1 : 0) * 16 * XS);       // (the projection behind the last step is computed and never used)
            step_barrier();
        }
    }
#else
    auto project = [&](const h16* xs) {
        h16x8 xf[KTI];
#pragma unroll
        for (int kt = 0; kt < KTI; ++kt) xf[kt] = *(const h16x8*)(xs + kt * 32 + q * 8);
#pragma unroll
        for (int t = 0; t < TPW; ++t) accx[t] = bias4[t];
#pragma unroll
        for (int kt = 0; kt < KTI; ++kt)
#pragma unroll
            for (int t = 0; t < TPW; ++t) accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wi[t][kt], xf[kt], accx[t], 0, 0, 0);
    };
    const int ngroups = (Wrun + G - 1) / G;
    load_x(0);
    park_x(0);
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        if (g + 1 < ngroups) load_x(g + 1);
        if (g > 0) store_group(g - 1);
        const int nst = Wrun - g * G < G ? Wrun - g * G : G;
        for (int i = 0; i < nst; ++i) {
            const int s = g * G + i;
            const h16* hprev = hring + (((s + R - 1) & (R - 1)) * 16 + col) * HS;
            h16* hnext = hring + ((s & (R - 1)) * 16 + col) * HS;
            if (i == 0) project(xring + (((g & 1) * G) * 16 + col) * XS);      // first step of a group: x just parked
            h16x8 bf[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) bf[kt] = *(const h16x8*)(hprev + kt * 32 + q * 8);
            f32x4 accs[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) accs[t] = accx[t];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int t = 0; t < TPW; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][kt], bf[kt], accs[t], 0, 0, 0);
            // gate math of all the wave's tiles first, the h stores after it: with the store's `if (j < H)` inside the tile loop the
            // compiler sank half of each tile's math into that branch and ran the tiles strictly one after the other -- two
            // dependent exp -> rcp -> exp -> rcp chains back to back instead of interleaved
            float hq[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const f32x4 a = accs[t];
                constexpr float LIM = 60.f, NOLIM = -3.0e38f;
                const float ei = aero_exp2(a[0] * GS), ef = aero_exp2(a[1] * GS);
                const float eg = aero_exp2(aero_med3(a[2] * GG, NOLIM, LIM));
                const float eo = aero_exp2(a[3] * GS);
                const float r1 = aero_rcp((1.f + ei) * (eg + 1.f));
                const float igg = fmaf(eg, r1, -r1);                                    // sigmoid(i) * tanh(g)
                c[t] = fmaf(aero_rcp(1.f + ef), c[t], igg);
                const float ec = aero_exp2(aero_med3(c[t] * (2.f * L2E), NOLIM, LIM));
                const float r2 = aero_rcp((1.f + eo) * (ec + 1.f));
                hq[t] = fmaf(ec, r2, -r2);                                              // sigmoid(o) * tanh(c)
                if constexpr (SAVE) {
                    const int j = (wave * TPW + t) * 4 + q;
                    if (j < H) {
                        const int tau_s = dir ? W - 1 - s : s;
                        const int64_t sb = ((int64_t)blockIdx.x * 2 + dir) * W + tau_s;
                        const float sig_i = r1 * (eg + 1.f), tanh_g = (eg - 1.f) * r1 * (1.f + ei);
                        const float sig_f = aero_rcp(1.f + ef), sig_o = r2 * (ec + 1.f);
                        *(h16x4*)((h16*)d.save_gates + sb * H * 64 + ((int64_t)j * 16 + col) * 4) = (h16x4){(h16)sig_i, (h16)sig_f, (h16)tanh_g, (h16)sig_o};
                        d.save_c[sb * H * 16 + j * 16 + col] = c[t];
                    }
                }
            }
#ifndef AERO_EMU
#pragma unroll
            for (int t = 0; t < TPW; ++t) asm volatile("" : "+v"(hq[t]));               // (keeps the math above out of the store's branch)
#endif
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int j = (wave * TPW + t) * 4 + q;
                if (j < H) hnext[j] = (h16)hq[t];
            }
            // input projection of the NEXT step (independent of h): issued before the barrier so its LDS reads and
            // MFMAs fill the pipes while the other waves finish their gate math
            if (i + 1 < nst) project(xring + (((g & 1) * G + i + 1) * 16 + col) * XS);
            else if (g + 1 < ngroups) park_x((g + 1) & 1);              // loads issued G steps ago
            if constexpr (SAVE) aero_phase_barrier();
            else __syncthreads();
        }
    }
#endif
    store_group(ngroups - 1);
}

static int aero_lstm_launch(const aero_lstm_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->whh || !d->out) { *err = "lstm: null pointer"; return AERO_ERR_ARG; }
    const bool fused = d->wih != nullptr;
    if (fused) {
        if (!d->x || !d->bias || d->in_ch < 1 || d->x_pitch < d->in_ch) { *err = "lstm: fused projection needs x, bias, in_ch, x_pitch"; return AERO_ERR_ARG; }
    } else if (!d->xproj || !d->xbias) {
        *err = "lstm: null xproj/xbias";
        return AERO_ERR_ARG;
    }
    if (d->H < 1 || d->nseq < 1 || d->W < 1) { *err = "lstm: bad geometry"; return AERO_ERR_ARG; }
    if ((d->in_mode == 1 || d->out_mode == 1) && (d->nframes < 1 || d->S < 1 || d->T < 1 || d->nseq % d->nframes)) {
        *err = "lstm: bad framing";
        return AERO_ERR_ARG;
    }
    {   // the ring kernel keeps per-thread row indices / output element offsets in 32 bits
        const int64_t rows_in = d->in_mode == 1 ? (int64_t)(d->nseq / d->nframes) * d->T : (int64_t)d->nseq * d->W;
        const int64_t rows_out = d->out_mode == 1 ? (int64_t)(d->nseq / d->nframes) * d->T : (int64_t)d->nseq * d->W;
        // (padding lanes of the last block form -- and never use -- offsets of up to 15 sequences past the end: the margin covers them too)
        const int64_t slack = (int64_t)16 * d->W + d->W;
        if (rows_in + slack >= (1ll << 31) || (rows_out + slack) * 2 * d->H >= (1ll << 31)) { *err = "lstm: tensor too large for 32-bit offsets"; return AERO_ERR_UNSUPPORTED; }
    }
    int nw, tpw, kt;
    if (aero_lstm_pick(d->H, &nw, &tpw, &kt)) { *err = "lstm: hidden size > 128 unsupported"; return AERO_ERR_UNSUPPORTED; }
    const int kti = fused ? aero_lstm_kti(d->H, d->in_ch) : 0;
    if (fused && kti <= 0) { *err = "lstm: no fused-projection instantiation for this (H, in_ch)"; return AERO_ERR_UNSUPPORTED; }
    AeroLstmK p;
    p.d = *d;
    p.MP = 16 * nw * tpw;
    p.KP = 32 * kt;
    p.dv = 1;
    if (d->in_mode == 1 || d->out_mode == 1) p.dv = d->frame_major ? d->nseq / d->nframes : d->nframes;
    dim3 grid((unsigned)((d->nseq + 15) / 16), 2), block((unsigned)(nw * 64));
    // AERO_LSTM_RING=0: step-wise kernel above (A/B);  AERO_LSTM_WIDE=0: keep the tiles on 6/8 waves instead of 12
    static int ring = -1, wide = -1;
    if (ring < 0) { const char* e = getenv("AERO_LSTM_RING"); ring = (e && e[0] == '0') ? 0 : 1; }
    if (wide < 0) { const char* e = getenv("AERO_LSTM_WIDE"); wide = (e && e[0] == '0') ? 0 : 1; }
#define AERO_LSTM_RING_GO(NW_, TPW_, KT_, KTI_, G_)                                                                     \
    do {                                                                                                                \
        block = dim3(NW_ * 64);                                                                                         \
        AERO_LAUNCH_DYN((aero_lstm_ring_kernel<NW_, TPW_, KT_, KTI_, G_>), grid, block,                                 \
                        (AeroLstmRingGeom<KT_, KTI_, G_>::BYTES), stream, p);                                           \
    } while (0)
    if ((d->save_gates == nullptr) != (d->save_c == nullptr)) { *err = "lstm: save_gates and save_c go together"; return AERO_ERR_ARG; }
    if (fused && ring && d->save_gates && wide && (nw == 6 || (nw == 8 && tpw == 3))) {
        // training-mode forward on the ring kernel (the hidden sizes of the reference configs: 48 -> 12 x 1 tiles, 96 -> 12 x 2)
#define AERO_LSTM_RING_SAVE(NW_, TPW_, KT_, KTI_, G_)                                                                   \
    do {                                                                                                                \
        block = dim3(NW_ * 64);                                                                                         \
        AERO_LAUNCH_DYN((aero_lstm_ring_kernel<NW_, TPW_, KT_, KTI_, G_, true>), grid, block,                           \
                        (AeroLstmRingGeom<KT_, KTI_, G_>::BYTES), stream, p);                                           \
    } while (0)
        if (nw == 6) { if (kti == 2) AERO_LSTM_RING_SAVE(12, 1, 2, 2, 4); else AERO_LSTM_RING_SAVE(12, 1, 2, 3, 4); }
        else { if (kti == 3) AERO_LSTM_RING_SAVE(12, 2, 3, 3, 8); else AERO_LSTM_RING_SAVE(12, 2, 3, 6, 4); }
#undef AERO_LSTM_RING_SAVE
        return AERO_OK;
    }
    if (fused && ring && !d->save_gates) {
        if (nw == 4 && tpw == 1) AERO_LSTM_RING_GO(4, 1, 1, 1, 4);
        else if (nw == 4 && tpw == 2) { if (kti == 1) AERO_LSTM_RING_GO(4, 2, 1, 1, 4); else AERO_LSTM_RING_GO(4, 2, 1, 2, 4); }
        else if (nw == 6) {
            if (wide) { if (kti == 2) AERO_LSTM_RING_GO(12, 1, 2, 2, 4); else AERO_LSTM_RING_GO(12, 1, 2, 3, 4); }
            else { if (kti == 2) AERO_LSTM_RING_GO(6, 2, 2, 2, 4); else AERO_LSTM_RING_GO(6, 2, 2, 3, 4); }
        }
        else if (tpw == 2) { if (kti == 2) AERO_LSTM_RING_GO(8, 2, 2, 2, 4); else AERO_LSTM_RING_GO(8, 2, 2, 4, 4); }
        else {
            if (wide) { if (kti == 3) AERO_LSTM_RING_GO(12, 2, 3, 3, 8); else AERO_LSTM_RING_GO(12, 2, 3, 6, 4); }
            else { if (kti == 3) AERO_LSTM_RING_GO(8, 3, 3, 3, 8); else AERO_LSTM_RING_GO(8, 3, 3, 6, 4); }
        }
        return AERO_OK;
    }
#undef AERO_LSTM_RING_GO
#define AERO_LSTM_GO(NW_, TPW_, KT_, KTI_) AERO_LAUNCH((aero_lstm_kernel<NW_, TPW_, KT_, KTI_>), grid, block, stream, p)
    if (nw == 4 && tpw == 1) { if (kti == 0) AERO_LSTM_GO(4, 1, 1, 0); else AERO_LSTM_GO(4, 1, 1, 1); }
    else if (nw == 4 && tpw == 2) { if (kti == 0) AERO_LSTM_GO(4, 2, 1, 0); else if (kti == 1) AERO_LSTM_GO(4, 2, 1, 1); else AERO_LSTM_GO(4, 2, 1, 2); }
    else if (nw == 6) { if (kti == 0) AERO_LSTM_GO(6, 2, 2, 0); else if (kti == 2) AERO_LSTM_GO(6, 2, 2, 2); else AERO_LSTM_GO(6, 2, 2, 3); }
    else if (tpw == 2) { if (kti == 0) AERO_LSTM_GO(8, 2, 2, 0); else if (kti == 2) AERO_LSTM_GO(8, 2, 2, 2); else AERO_LSTM_GO(8, 2, 2, 4); }
    else if (tpw == 3) { if (kti == 0) AERO_LSTM_GO(8, 3, 3, 0); else if (kti == 3) AERO_LSTM_GO(8, 3, 3, 3); else AERO_LSTM_GO(8, 3, 3, 6); }
    else AERO_LSTM_GO(8, 4, 4, 0);
#undef AERO_LSTM_GO
    return AERO_OK;
}
