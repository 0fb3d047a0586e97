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

template <int NW, int TPW, int KT>
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
    load_xp(dir ? W - 1 : 0, xp_cur);
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < W; ++step) {
        const int tau = dir ? W - 1 - step : step;
        if (step + 1 < W) load_xp(dir ? tau - 1 : tau + 1, xp_nxt);
        h16x8 bf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) bf[kt] = *(const h16x8*)&hbuf[cur][col * KP + kt * 32 + q * 8];
        f32x4 accs[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i)
            accs[i] = (f32x4){(float)xp_cur[i][0], (float)xp_cur[i][1], (float)xp_cur[i][2], (float)xp_cur[i][3]};
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
#pragma unroll
        for (int i = 0; i < TPW; ++i) xp_cur[i] = xp_nxt[i];
        cur ^= 1;
    }
}

static int aero_lstm_launch(const aero_lstm_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->xproj || !d->xbias || !d->whh || !d->out) { *err = "lstm: null pointer"; return AERO_ERR_ARG; }
    if (d->H < 1 || d->nseq < 1 || d->W < 1) { *err = "lstm: bad geometry"; return AERO_ERR_ARG; }
    if ((d->in_mode == 1 || d->out_mode == 1) && (d->nframes < 1 || d->S < 1 || d->T < 1 || d->nseq % d->nframes)) {
        *err = "lstm: bad framing";
        return AERO_ERR_ARG;
    }
    int nw, tpw, kt;
    if (aero_lstm_pick(d->H, &nw, &tpw, &kt)) { *err = "lstm: hidden size > 128 unsupported"; return AERO_ERR_UNSUPPORTED; }
    AeroLstmK p;
    p.d = *d;
    p.MP = 16 * nw * tpw;
    p.KP = 32 * kt;
    dim3 grid((unsigned)((d->nseq + 15) / 16), 2), block((unsigned)(nw * 64));
    if (nw == 4 && tpw == 1) AERO_LAUNCH((aero_lstm_kernel<4, 1, 1>), grid, block, stream, p);
    else if (nw == 4 && tpw == 2) AERO_LAUNCH((aero_lstm_kernel<4, 2, 1>), grid, block, stream, p);
    else if (nw == 6) AERO_LAUNCH((aero_lstm_kernel<6, 2, 2>), grid, block, stream, p);
    else if (tpw == 2) AERO_LAUNCH((aero_lstm_kernel<8, 2, 2>), grid, block, stream, p);
    else if (tpw == 3) AERO_LAUNCH((aero_lstm_kernel<8, 3, 3>), grid, block, stream, p);
    else AERO_LAUNCH((aero_lstm_kernel<8, 4, 4>), grid, block, stream, p);
    return AERO_OK;
}
