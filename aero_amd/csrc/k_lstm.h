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
#include "aero_common.h"

struct AeroLstmK {
    aero_lstm_desc d;
    int MP, KP;
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

template <int NW, int TPW, int KT, int KTI>
__global__ __launch_bounds__(NW * 64) void aero_lstm_kernel(AeroLstmK p) {
    constexpr int KP = KT * 32;
    constexpr int MP = NW * TPW * 16;
    constexpr int NT = NW * 64;
    __shared__ AERO_LDS_ALIGN h16 hbuf[2][16 * KP];
    const aero_lstm_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        const int r = seq / d.nframes, k = seq % d.nframes;
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
                const int r = s2 / d.nframes, k = s2 % d.nframes;
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
    int nw, tpw, kt;
    if (aero_lstm_pick(d->H, &nw, &tpw, &kt)) { *err = "lstm: hidden size > 128 unsupported"; return AERO_ERR_UNSUPPORTED; }
    const int kti = fused ? aero_lstm_kti(d->H, d->in_ch) : 0;
    if (fused && kti <= 0) { *err = "lstm: no fused-projection instantiation for this (H, in_ch)"; return AERO_ERR_UNSUPPORTED; }
    AeroLstmK p;
    p.d = *d;
    p.MP = 16 * nw * tpw;
    p.KP = 32 * kt;
    dim3 grid((unsigned)((d->nseq + 15) / 16), 2), block((unsigned)(nw * 64));
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
