// k_train.h -- the backward kernels that complete one training step of the generator (reference
// src/solver.py:602-605 `loss.backward()` through src/models/aero.py / modules.py, loss = src/models/stft_loss.py):
//
//   aero_freqfc_wgrad      FTB freq_fc weight gradient (modules.py:296,320): an NT GEMM over (b, t, c)
//   aero_ftb_gate_bwd      FTB gate product (modules.py:316): dx += v * gate, dgate = sum_f v * x
//   aero_sum_bt            sum over (b, t) per (f, c): the gradient of the frequency embedding (aero.py:475-480)
//   aero_frames_op         unfold / stitch of BLSTM (models/utils.py:22-35, modules.py:36-62) and their adjoints
//   aero_lstm_bwd          BPTT of one bidirectional nn.LSTM layer (modules.py:28,46) from the gates / cell states
//                          the training-mode forward saved (aero_lstm_fwd with save_gates / save_c)
//   aero_localstate_bwd    LocalState attention backward (modules.py:94-127)
//   aero_stft_loss_sums / aero_stft_loss_bwd     spectral-convergence + log-magnitude loss of one resolution
//                          (stft_loss.py:11-27,30-64) and its gradient w.r.t. the predicted signal's STFT
//   aero_irfft_frames / aero_stft_adj_fold      adjoint of the centred, reflect-padded STFT (torch.stft of stft_loss.py:22)
//   aero_axpy_f16, aero_scale_cast, aero_absmax_f32, aero_scale_f32     element-wise plumbing of the gradient path
//                          (sum of two gradient paths, the fp32 -> fp16 boundary with the loss scale, un-scaling)
//
// State at the end of round 3: the LSTM backward runs in ring form (group-wise loads, LDS-only barriers), the LocalState
// backward on MFMA; the streaming kernels (loss, frames, scaling, gather re-pack) are bound by the bytes they move; what bounds a
// training step at the per-GPU batch is the chain of latency-bound launches, not any one of these.
#pragma once
#include "aero_common.h"
#include "k_stft.h"

// ------------------------------------------------------------------------------------------------------------------
// FTB freq_fc weight gradient.  Forward (k_ftb.h): fc[b,f,n] = gate[b,n] * sum_f' W[f][f'] x[b,f',n], n = (t, c).
//   dW[f][f'] = sum_{b, n} dfc[b,f,n] * gate[b,n] * x[b,f',n]
// Both operands are K-contiguous ([F][N] rows per batch item): an NT GEMM with K = B*N.  A block owns a 64 x 64 tile of dW and a
// slice of the (b, n) range; its partial tile goes to its own slab, added in slice order by aero_wgrad_finish_kernel (k_bwd.h).
struct AeroFfcWgradK {
    const h16* dfc; const h16* x; const h16* gate; float* slabs;
    int B, F, nmt, nslice, chunks_per_b, chunks_per_slice;
    int64_t N;
};

__global__ __launch_bounds__(256) void aero_freqfc_wgrad_kernel(AeroFfcWgradK p) {
    __shared__ AERO_LDS_ALIGN h16 As[64 * 32];
    __shared__ AERO_LDS_ALIGN h16 Bs[64 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int mt = blockIdx.x % p.nmt, nt = blockIdx.x / p.nmt;
    const int slice = blockIdx.y;
    const int m0 = mt * 64, n0 = nt * 64;
    const int row = tid >> 2, slot = tid & 3;
    f32x4 acc[4];                                              // wave w: rows (w & 1) * 32 .. +32 (2 tiles), cols (w >> 1) * 32 .. +32 (2 tiles)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t total = (int64_t)p.B * p.chunks_per_b;
    int64_t c0 = (int64_t)slice * p.chunks_per_slice, c1 = c0 + p.chunks_per_slice;
    if (c1 > total) c1 = total;
    for (int64_t ch = c0; ch < c1; ++ch) {
        const int b = (int)(ch / p.chunks_per_b);
        const int64_t k = (ch - (int64_t)b * p.chunks_per_b) * 32 + slot * 8;
        h16x8 va = (h16x8){0, 0, 0, 0, 0, 0, 0, 0}, vb = va;
        if (k < p.N) {                                          // N % 8 == 0: a vector is inside or outside
            const h16x8 g = *(const h16x8*)(p.gate + (int64_t)b * p.N + k);
            if (m0 + row < p.F) va = *(const h16x8*)(p.dfc + ((int64_t)b * p.F + m0 + row) * p.N + k) * g;
            if (n0 + row < p.F) vb = *(const h16x8*)(p.x + ((int64_t)b * p.F + n0 + row) * p.N + k);
        }
        __syncthreads();
        *(h16x8*)&As[aero_tile_off(row, slot)] = va;
        *(h16x8*)&Bs[aero_tile_off(row, slot)] = vb;
        __syncthreads();
        h16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = *(const h16x8*)&As[aero_tile_off((wave & 1) * 32 + i * 16 + (lane & 15), lane >> 4)];
            bf[i] = *(const h16x8*)&Bs[aero_tile_off((wave >> 1) * 32 + i * 16 + (lane & 15), lane >> 4)];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i * 2 + j], 0, 0, 0);
    }
    float* slab = p.slabs + (int64_t)slice * p.F * p.F;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f2 = n0 + (wave >> 1) * 32 + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f1 = m0 + (wave & 1) * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (f1 < p.F && f2 < p.F) slab[(int64_t)f1 * p.F + f2] = acc[i * 2 + j][r];
            }
        }
}

__global__ __launch_bounds__(256) void aero_slab_sum_kernel(const float* slabs, int nslab, float* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += slabs[(int64_t)k * n + i];
    dst[i] += s;
}

static int aero_freqfc_wgrad_launch(const void* dfc, const void* x, const void* gate, float* dw, float* slabs, int nslab, int B, int F,
                                    int T, int C, hipStream_t stream, const char** err) {
    if (!dfc || !x || !gate || !dw || !slabs) { *err = "freqfc_wgrad: null pointer"; return AERO_ERR_ARG; }
    const int64_t N = (int64_t)T * C;
    if (B < 1 || F < 1 || N < 8 || (N % 8) || nslab < 1) { *err = "freqfc_wgrad: T*C must be a multiple of 8"; return AERO_ERR_UNSUPPORTED; }
    if (((uintptr_t)dfc | (uintptr_t)x | (uintptr_t)gate) & 15) { *err = "freqfc_wgrad: 16-byte aligned tensors required"; return AERO_ERR_ARG; }
    AeroFfcWgradK p;
    p.dfc = (const h16*)dfc; p.x = (const h16*)x; p.gate = (const h16*)gate; p.slabs = slabs;
    p.B = B; p.F = F; p.N = N;
    p.nmt = (F + 63) / 64;
    p.chunks_per_b = (int)((N + 31) / 32);
    const int64_t total = (int64_t)B * p.chunks_per_b;
    p.chunks_per_slice = (int)((total + nslab - 1) / nslab);
    p.nslice = (int)((total + p.chunks_per_slice - 1) / p.chunks_per_slice);
    AERO_LAUNCH(aero_freqfc_wgrad_kernel, dim3((unsigned)(p.nmt * p.nmt), (unsigned)p.nslice), dim3(256), stream, p);
    const int64_t n = (int64_t)F * F;
    AERO_LAUNCH(aero_slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream, (const float*)slabs, p.nslice, dw, n);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// FTB gate product backward (modules.py:316 `att_out = conv1d_out * inputs`, with the gate moved behind freq_fc as in k_ftb.h):
// v = W^T dfc (aero_freqfc_fwd on the transposed weight with a gate of ones) is the gradient of u = gate * x summed over nothing:
//   dx[b,f,n]   = add[b,f,n] + v[b,f,n] * gate[b,n]         (add: the other gradient paths into x, may be NULL)
//   dgate[b,n]  = sum_f v[b,f,n] * x[b,f,n]
// One thread owns 8 consecutive n of one batch item and walks the F rows: 16-byte accesses, every tensor read once.
__global__ __launch_bounds__(256) void aero_ftb_gate_bwd_kernel(const h16* v, const h16* x, const h16* gate, const h16* add, h16* dx, h16* dgate,
                                                                int F, int64_t N) {
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (n >= N) return;
    const int b = blockIdx.y;
    const h16x8 g = *(const h16x8*)(gate + (int64_t)b * N + n);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int f = 0; f < F; ++f) {
        const int64_t o = ((int64_t)b * F + f) * N + n;
        const h16x8 vv = *(const h16x8*)(v + o), xx = *(const h16x8*)(x + o);
        h16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] += (float)vv[e] * (float)xx[e];
            r[e] = (h16)((float)vv[e] * (float)g[e] + (add ? (float)add[o + e] : 0.f));
        }
        *(h16x8*)(dx + o) = r;
    }
    h16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (h16)acc[e];
    *(h16x8*)(dgate + (int64_t)b * N + n) = r;
}

static int aero_ftb_gate_bwd_launch(const void* v, const void* x, const void* gate, const void* add, void* dx, void* dgate, int B, int F, int T,
                                    int C, hipStream_t stream, const char** err) {
    if (!v || !x || !gate || !dx || !dgate) { *err = "ftb_gate_bwd: null pointer"; return AERO_ERR_ARG; }
    const int64_t N = (int64_t)T * C;
    if (B < 1 || F < 1 || (N % 8) || B > 65535) { *err = "ftb_gate_bwd: T*C must be a multiple of 8"; return AERO_ERR_UNSUPPORTED; }
    if (((uintptr_t)v | (uintptr_t)x | (uintptr_t)gate | (uintptr_t)add | (uintptr_t)dx | (uintptr_t)dgate) & 15) { *err = "ftb_gate_bwd: alignment"; return AERO_ERR_ARG; }
    AERO_LAUNCH(aero_ftb_gate_bwd_kernel, dim3((unsigned)((N / 8 + 255) / 256), (unsigned)B), dim3(256), stream, (const h16*)v, (const h16*)x,
                (const h16*)gate, (const h16*)add, (h16*)dx, (h16*)dgate, F, N);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// out[f][c] += scale * sum_{b, t} x[b, f, t, c]   (x fp16 [B,F,T,C] contiguous, out fp32 [F][C]): the gradient of the frequency
// embedding table (aero.py:478-480: emb(frs).t()[None,:,:,None].expand_as(x) added to the encoder-0 output).
__global__ __launch_bounds__(256) void aero_sum_bt_kernel(const h16* x, float* out, int B, int F, int T, int C, float scale) {
    __shared__ float red[256];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int lanes = 256 / C > 0 ? 256 / C : 1;               // time lanes per channel
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + (C >= 256 ? tid : tid % C);
        const int tl = C >= 256 ? 0 : tid / C;
        float s = 0.f;
        if (c < C && tl < lanes) {
            for (int b = 0; b < B; ++b) {
                const h16* row = x + ((int64_t)b * F + f) * T * C;
                for (int t = tl; t < T; t += lanes) s += (float)row[(int64_t)t * C + c];
            }
        }
        red[tid] = s;
        __syncthreads();
        if (tl == 0 && c < C) {
            float tot = 0.f;
            for (int l = 0; l < lanes; ++l) tot += red[l * C + (c - c0)];
            out[(int64_t)f * C + c] += tot * scale;
        }
        __syncthreads();
    }
}

static int aero_sum_bt_launch(const void* x, float* out, int B, int F, int T, int C, float scale, hipStream_t stream, const char** err) {
    if (!x || !out || B < 1 || F < 1 || T < 1 || C < 1) { *err = "sum_bt: bad arguments"; return AERO_ERR_ARG; }
    AERO_LAUNCH(aero_sum_bt_kernel, dim3((unsigned)F), dim3(256), stream, (const h16*)x, out, B, F, T, C, scale);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// BLSTM framing (models/utils.py:22-35 `unfold`, modules.py:36-42) and stitching (modules.py:49-61) as index maps, and their
// adjoints.  rows: [R][T][C] fp16, frames: [R*nframes][W][C] fp16, frame k of row r starts at t = k*S (W = 2*S), the stitch
// keeps tau in [lo_k, hi_k) of frame k: lo = 0 for k = 0 else S/2, hi = W for the last frame else W - S/2.
//   mode 0  unfold        frames[r*nf+k][tau] = rows[r][k*S+tau] (0 beyond T)
//   mode 1  unfold^T      rows[r][t]         = sum_k frames[r*nf+k][t-k*S]
//   mode 2  stitch        rows[r][t]         = frames[r*nf+k(t)][t-k(t)*S]
//   mode 3  stitch^T      frames[r*nf+k][tau] = rows[r][k*S+tau] if tau kept and k*S+tau < T else 0
struct AeroFramesK {
    const h16* src; h16* dst;
    int mode, R, T, C, nframes, W, S;
};

static __device__ __forceinline__ int aero_stitch_frame(int t, int S, int nframes) {
    int k = (t - S / 2) >= 0 ? (t - S / 2) / S : 0;
    return k > nframes - 1 ? nframes - 1 : k;
}

__global__ __launch_bounds__(256) void aero_frames_kernel(AeroFramesK p) {
    const bool to_frames = p.mode == 0 || p.mode == 3;
    const int64_t npos = to_frames ? (int64_t)p.R * p.nframes * p.W : (int64_t)p.R * p.T;
    const int64_t total = npos * p.C;
    const int lim = p.S / 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pos = e / p.C;
        const int c = (int)(e - pos * p.C);
        float v = 0.f;
        if (to_frames) {
            const int tau = (int)(pos % p.W);
            const int64_t seq = pos / p.W;
            const int k = (int)(seq % p.nframes);
            const int64_t r = seq / p.nframes;
            const int t = k * p.S + tau;
            bool ok = t < p.T;
            if (p.mode == 3) {
                const int lo = k == 0 ? 0 : lim, hi = (k == p.nframes - 1) ? p.W : p.W - lim;
                ok = ok && tau >= lo && tau < hi;
            }
            if (ok) v = (float)p.src[(r * p.T + t) * p.C + c];
        } else {
            const int t = (int)(pos % p.T);
            const int64_t r = pos / p.T;
            if (p.mode == 2) {
                const int k = aero_stitch_frame(t, p.S, p.nframes);
                v = (float)p.src[((r * p.nframes + k) * p.W + (t - k * p.S)) * p.C + c];
            } else {
                int k1 = t / p.S;
                if (k1 > p.nframes - 1) k1 = p.nframes - 1;
                for (int k = k1; k >= 0 && t - k * p.S < p.W; --k) v += (float)p.src[((r * p.nframes + k) * p.W + (t - k * p.S)) * p.C + c];
            }
        }
        p.dst[e] = (h16)v;
    }
}

static int aero_frames_launch(const void* src, void* dst, int mode, int R, int T, int C, int nframes, int W, int S, hipStream_t stream, const char** err) {
    if (!src || !dst || mode < 0 || mode > 3 || R < 1 || T < 1 || C < 1 || nframes < 1 || W != 2 * S || S < 2 || (nframes - 1) * S >= T + S) {
        *err = "frames_op: bad arguments (W = 2*S, nframes = ceil(T/S))"; return AERO_ERR_ARG;
    }
    AeroFramesK p;
    p.src = (const h16*)src; p.dst = (h16*)dst; p.mode = mode; p.R = R; p.T = T; p.C = C; p.nframes = nframes; p.W = W; p.S = S;
    const int64_t npos = (mode == 0 || mode == 3) ? (int64_t)R * nframes * W : (int64_t)R * T;
    int64_t nb = (npos * C + 255) / 256;
    if (nb > 65535) nb = 65535;
    AERO_LAUNCH(aero_frames_kernel, dim3((unsigned)nb), dim3(256), stream, p);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// BPTT of one bidirectional LSTM layer.  The training-mode forward (aero_lstm_kernel with save pointers, k_lstm.h) stored, per
// (block of 16 sequences, direction, step tau): the gate activations i, f, g, o as fp16 [H][16][4] and the cell state c_tau as
// fp32 [H][16].  This kernel walks the steps of a direction in reverse and emits the gradient of the gate PRE-activations
//   da[pos][dir][4*j + gate]   (fp16, pos = seq*W + tau)
// from which the host side gets every parameter and input gradient with the existing GEMM kernels (aero_conv_wgrad on (da, x) and
// (da, h_prev); aero_conv_fwd with W_ih^T).  Per step: dh = dout[pos] + W_hh^T da_next (MFMA, W_hh^T resident in registers as
// A fragments: rows = hidden units, k = 4*j' + gate), then the cell algebra lane-locally (a lane holds 4 units of one sequence).
// dout: [.., 2H] gradient of the layer output; out_mode 1 reads it through the stitch map of modules.py:52-61 (zero outside
// the kept range of a frame), exactly as the forward kernel's out_mode 1 writes.
struct AeroLstmBwdK {
    aero_lstm_bwd_desc d;
    int K4P;
};

template <int KT4>
__global__ __launch_bounds__(512) void aero_lstm_bwd_kernel(AeroLstmBwdK p) {
    constexpr int K4P = KT4 * 32;
    __shared__ AERO_LDS_ALIGN h16 dabuf[2][16 * (K4P + 8)];
    constexpr int LD = K4P + 8;
    const aero_lstm_bwd_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int NT = blockDim.x;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * 16;
    const int H = d.H, W = d.W, H4 = 4 * H, H2 = 2 * H;
    const int q = lane >> 4, col = lane & 15;
    const int HP = ((H + 15) / 16) * 16;
    const h16* WT = (const h16*)d.whh_t + (int64_t)dir * HP * K4P;
    h16x8 wf[KT4];
#pragma unroll
    for (int kt = 0; kt < KT4; ++kt) wf[kt] = *(const h16x8*)(WT + (int64_t)(wave * 16 + col) * K4P + kt * 32 + q * 8);
    for (int idx = tid; idx < 2 * 16 * LD; idx += NT) (&dabuf[0][0])[idx] = (h16)0;

    const int seq = seq0 + col;
    const bool seq_ok = seq < d.nseq;
    // dout addressing of this lane's sequence
    int64_t o_base = 0;
    int t0 = 0, lo = 0, hi = W;
    if (d.out_mode == 1) {
        const int r = seq_ok ? seq / d.nframes : 0, k = seq_ok ? seq % d.nframes : 0;
        const int lim = d.S / 2;
        lo = (k == 0) ? 0 : lim;
        hi = (k == d.nframes - 1 && k != 0) ? W : W - lim;
        t0 = k * d.S;
        o_base = (int64_t)r * d.T;
    } else {
        o_base = (int64_t)seq * W;
    }
    const h16* dout = (const h16*)d.dout;
    const h16* gs = (const h16*)d.save_gates + ((int64_t)blockIdx.x * 2 + dir) * W * ((int64_t)H * 64);
    const float* cs = d.save_c + ((int64_t)blockIdx.x * 2 + dir) * W * ((int64_t)H * 16);
    h16* da_out = (h16*)d.da;
    const int j0 = wave * 16 + q * 4;                          // this lane's 4 hidden units
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    // cooperative store of a step's da rows from LDS: 16 sequences x 4H values
    const int vec = (H4 % 8 == 0) ? 8 : 4;                     // H4 is a multiple of 4
    const int per = H4 / vec;
    // the saved activations and dout of a step do not depend on the recurrence: they are fetched ONE STEP AHEAD into registers, so
    // the step's dependent chain is LDS read -> MFMA -> cell algebra -> LDS write -> barrier with no global-memory latency in it
    // (first version: loads after the MFMA, 4.5 us per step; the launch has only nseq/16 x 2 blocks, so latency is all there is)
    struct Pre { h16x4 g4[4]; float ct[4], cp[4]; h16 dy[4]; };
    auto fetch = [&](int step, Pre& pr) {
        const int tau = dir ? step : W - 1 - step;
        const int tau_prev = dir ? tau + 1 : tau - 1;
        bool kept = seq_ok;
        int64_t opos = o_base + tau;
        if (d.out_mode == 1) {
            const int t = t0 + tau;
            kept = kept && tau >= lo && tau < hi && t < d.T;
            opos = o_base + t;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + r;
            pr.g4[r] = (h16x4){0, 0, 0, 0};
            pr.ct[r] = pr.cp[r] = 0.f;
            pr.dy[r] = (h16)0;
            if (j < H && seq_ok) {
                pr.g4[r] = *(const h16x4*)(gs + (int64_t)tau * H * 64 + ((int64_t)j * 16 + col) * 4);
                pr.ct[r] = cs[(int64_t)tau * H * 16 + j * 16 + col];
                if (tau_prev >= 0 && tau_prev < W) pr.cp[r] = cs[(int64_t)tau_prev * H * 16 + j * 16 + col];
                if (kept) pr.dy[r] = dout[opos * H2 + dir * H + j];
            }
        }
    };
    Pre nxt;
    fetch(0, nxt);
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < W; ++step) {
        // forward order of dir 0 is tau = 0..W-1 (dir 1: W-1..0); the backward pass walks it in reverse
        const int tau = dir ? step : W - 1 - step;
        const Pre pre = nxt;
        if (step + 1 < W) fetch(step + 1, nxt);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT4; ++kt) {
            const h16x8 bf = *(const h16x8*)&dabuf[cur][col * LD + kt * 32 + q * 8];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kt], bf, acc, 0, 0, 0);
        }
        h16x4 dav[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + r;
            dav[r] = (h16x4){0, 0, 0, 0};
            if (j < H && seq_ok) {
                const float dh = acc[r] + (float)pre.dy[r];
                const float ig = (float)pre.g4[r][0], fg = (float)pre.g4[r][1], gg = (float)pre.g4[r][2], og = (float)pre.g4[r][3];
                const float th = aero_tanh(pre.ct[r]);
                const float d_o = dh * th;
                const float dct = dc[r] + dh * og * (1.f - th * th);
                const float d_i = dct * gg, d_g = dct * ig, d_f = dct * pre.cp[r];
                dc[r] = dct * fg;
                dav[r] = (h16x4){(h16)(d_i * ig * (1.f - ig)), (h16)(d_f * fg * (1.f - fg)), (h16)(d_g * (1.f - gg * gg)), (h16)(d_o * og * (1.f - og))};
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (j0 + r < H) *(h16x4*)&dabuf[cur ^ 1][col * LD + (j0 + r) * 4] = dav[r];      // (slots >= 4H stay zero)
        __syncthreads();
        for (int idx = tid; idx < 16 * per; idx += NT) {
            const int sl = idx / per, e = idx - sl * per;
            const int s2 = seq0 + sl;
            if (s2 >= d.nseq) continue;
            h16* dst = da_out + (((int64_t)s2 * W + tau) * 2 + dir) * H4 + e * vec;
            if (vec == 8) *(h16x8*)dst = *(const h16x8*)&dabuf[cur ^ 1][sl * LD + e * 8];
            else *(h16x4*)dst = *(const h16x4*)&dabuf[cur ^ 1][sl * LD + e * 4];
        }
        cur ^= 1;
    }
}

// Ring form of the BPTT kernel (what the engine runs for H % 16 == 0).  The step-wise kernel above fetches a step's saved activations
// one step ahead and `__syncthreads()` then waits for those loads (and for the previous step's da stores) at EVERY step: 2.2-2.5 us per
// step, almost all of it one global round trip -- the launch has only nseq/16 x 2 blocks of 3-6 waves, so nothing else hides it.  Here,
// as in the forward ring kernel (k_lstm.h), all global reads move in GROUPS of G steps: the saved gates / cell states / dout rows of group
// g+1 are loaded into registers at the START of group g (15 sixteen-byte loads per lane) and parked in LDS at its END, G steps later;
// the step loop itself contains LDS reads, the MFMA chain (two accumulators), the cell algebra, one LDS write, one LDS-only barrier
// (lgkmcnt + s_barrier: the da stores of the step stay in flight) -- the vector-memory counter is waited for once per group.
// LDS: gates [G][H][16 x 4 fp16, rows padded to 160 B], cell states [G + 1][H][16 fp32, rows padded to 80 B] (the extra row is c of the
// step after the group: c_prev of its last step), dout [G][16][H + 4] fp16; paddings chosen so that the four k-octets of a wave read
// disjoint banks.
template <int KT4, int G>
__global__ __launch_bounds__(512) void aero_lstm_bwd_ring_kernel(AeroLstmBwdK p) {
    constexpr int K4P = KT4 * 32, LD = K4P + 8;
    constexpr int GS = 80, CS = 20;                              // row strides: gates (h16), cell states (float)
    const aero_lstm_bwd_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int NT = blockDim.x;
    const int dir = blockIdx.y;
    const int seq0 = blockIdx.x * 16;
    const int H = d.H, W = d.W, H4 = 4 * H, H2 = 2 * H;
    const int q = lane >> 4, col = lane & 15;
    const int HP = ((H + 15) / 16) * 16;
    const int DYS = H + 4;
    h16* dabuf = (h16*)AERO_DYN_SMEM;                            // [2][16 * LD]
    h16* gring = dabuf + 2 * 16 * LD;                            // [G][H][GS]
    float* cring = (float*)(gring + G * H * GS);                 // [G + 1][H][CS]
    h16* dyring = (h16*)(cring + (G + 1) * H * CS);              // [G][16][DYS]
    const h16* WT = (const h16*)d.whh_t + (int64_t)dir * HP * K4P;
    h16x8 wf[KT4];
#pragma unroll
    for (int kt = 0; kt < KT4; ++kt) wf[kt] = *(const h16x8*)(WT + (int64_t)(wave * 16 + col) * K4P + kt * 32 + q * 8);
    for (int idx = tid; idx < 2 * 16 * LD; idx += NT) dabuf[idx] = (h16)0;

    const h16* dout = (const h16*)d.dout;
    const h16* gs = (const h16*)d.save_gates + ((int64_t)blockIdx.x * 2 + dir) * W * ((int64_t)H * 64);
    const float* cs = d.save_c + ((int64_t)blockIdx.x * 2 + dir) * W * ((int64_t)H * 16);
    h16* da_out = (h16*)d.da;
    // Pieces (16 bytes) of a group, with NT = 4 H threads (H a multiple of 16): per step 8H = 2 NT of gates (two per thread), 4H = NT
    // of cell states (one per thread), 2H = NT / 2 of dout rows (one per thread every other step); plus one more row of cell states (the
    // step after the group).  So a thread's 3.5 G + 1 pieces have compile-time step indices and per-thread constant offsets.
    // Every piece is ONE unconditional load: what must read as zero -- steps past the end, sequences past nseq, dout rows outside a
    // frame's kept range -- reads the zero page (`cond ? load : 0` made the compiler wait for each load where it was issued).
    constexpr int NG = 2 * G, NC = G + 1, ND = G / 2;
    h16x8 rg[NG], rc[NC], rd[ND];
    const h16* zpage = aero_zero_page;
    const int hsel = tid >= (NT >> 1) ? 1 : 0;                   // dout: which of the pair of steps this thread copies
    const int r2 = tid - hsel * (NT >> 1);
    const int dsl = r2 / (H >> 3), de = r2 - dsl * (H >> 3);
    const int dseq = seq0 + dsl;
    int d_lo = 0, d_hi = dseq < d.nseq ? W : -1, d_tmax = 0x7fffffff;
    int64_t d_base = (int64_t)dseq * W * H2 + dir * H + de * 8;
    if (d.out_mode == 1 && dseq < d.nseq) {
        const int r = dseq / d.nframes, k = dseq - r * d.nframes;
        const int lim = d.S / 2;
        d_lo = (k == 0) ? 0 : lim;
        d_hi = (k == d.nframes - 1 && k != 0) ? W : W - lim;
        d_tmax = d.T - k * d.S;                                   // tau must stay below it
        d_base = ((int64_t)r * d.T + k * d.S) * H2 + dir * H + de * 8;
    }
    auto tau_of = [&](int step) { return dir ? step : W - 1 - step; };
    auto load_group = [&](int g) {
#pragma unroll
        for (int v = 0; v < NG; ++v) {
            const int step = g * G + (v >> 1);
            const h16* src = step < W ? gs + (int64_t)tau_of(step) * H * 64 + (tid + (v & 1) * NT) * 8 : zpage;
            rg[v] = *(const h16x8*)src;
        }
#pragma unroll
        for (int v = 0; v < NC; ++v) {
            const int step = g * G + v;
            const h16* src = step < W ? (const h16*)(cs + (int64_t)tau_of(step) * H * 16) + tid * 8 : zpage;
            rc[v] = *(const h16x8*)src;
        }
#pragma unroll
        for (int v = 0; v < ND; ++v) {
            const int step = g * G + 2 * v + hsel;
            const int tau = tau_of(step);
            const bool ok = step < W && tau >= d_lo && tau < d_hi && tau < d_tmax;
            const h16* src = ok ? dout + d_base + (int64_t)tau * H2 : zpage;
            rd[v] = *(const h16x8*)src;
        }
    };
    auto park_group = [&]() {
#pragma unroll
        for (int v = 0; v < NG; ++v) {
            const int rem = tid + (v & 1) * NT;
            *(h16x8*)(gring + ((v >> 1) * H + (rem >> 3)) * GS + (rem & 7) * 8) = rg[v];
        }
#pragma unroll
        for (int v = 0; v < NC; ++v) *(h16x8*)((h16*)(cring + (v * H + (tid >> 2)) * CS) + (tid & 3) * 8) = rc[v];
#pragma unroll
        for (int v = 0; v < ND; ++v) {
            const h16x8 x = rd[v];                              // (row stride H + 4 halves: 8-byte aligned)
            h16* dst = dyring + ((2 * v + hsel) * 16 + dsl) * DYS + de * 8;
            *(h16x4*)dst = (h16x4){x[0], x[1], x[2], x[3]};
            *(h16x4*)(dst + 4) = (h16x4){x[4], x[5], x[6], x[7]};
        }
    };
    const int j0 = wave * 16 + q * 4;                          // this lane's 4 hidden units
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    const int vec = 8, per = H4 / vec;
    const int ngroups = (W + G - 1) / G;
    // Round 5: the step's addressing hoisted.  Everything lane-dependent about a step's LDS reads / writes and its two global stores is a
    // per-thread constant (computed once, here); what changes from step to step is block-uniform (step-in-group i, double-buffer side, tau)
    // and is added as ONE scalar per array.  The ISA of the round-4 step rebuilt (i * H + j) * stride per unit and a 64-bit store address
    // with an integer division per piece: 35 of its 163 vector instructions.
    const int gl = j0 * GS + col * 4;                          // gates ring:  + i * H * GS + r * GS
    const int cl = j0 * CS + col;                              // cell ring:   + i * H * CS + r * CS  (next step's row: + H * CS)
    const int dl = col * DYS + j0;                             // dout ring:   + i * 16 * DYS
    const int bl = col * LD + q * 8;                           // da buffer read:  + side * 16 * LD + kt * 32
    const int wl = col * LD + j0 * 4;                          // da buffer write: + side * 16 * LD + r * 4
    // the block's da rows of a step go out as 16-byte pieces, piece = tid + k * NT: with NT = 4 H threads exactly two per thread
    constexpr int NPC = 2;
    int st_src[NPC];
    int64_t st_dst[NPC];                                       // element offset of tau = 0 (< 0: nothing to store)
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int idx = tid + k * NT;
        const int sl = idx / per, e = idx - sl * per;
        const int s2 = seq0 + sl;
        st_src[k] = sl * LD + e * 8;
        st_dst[k] = (idx < 16 * per && s2 < d.nseq) ? ((int64_t)s2 * W * 2 + dir) * H4 + e * vec : -1;
    }
    const int64_t st_tau = (int64_t)2 * H4;                    // elements per step of one sequence's [W][2][4H] block
    const bool two_pieces = 16 * per <= NPC * NT;              // (always with NT = 4 H; a launch with fewer threads takes the generic loop)
    load_group(0);
    park_group();
    __syncthreads();
    int cur = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (g + 1 < ngroups) load_group(g + 1);
        const int nst = W - g * G < G ? W - g * G : G;
        for (int i = 0; i < nst; ++i) {
            const int step = g * G + i;
            const int tau = tau_of(step);
            const int iH = i * H;                               // (block-uniform: scalar unit)
            const h16* gp = gring + (gl + iH * GS);
            const float* cq = cring + (cl + iH * CS);
            const h16* rb = dabuf + (bl + cur * 16 * LD);
            h16* wb = dabuf + (wl + (cur ^ 1) * 16 * LD);
            // every LDS read of the step first (one latency, not one per MFMA pair / per hidden unit), then the two MFMA chains, then the
            // four units' cell algebra as straight-line code: H is a multiple of 16 here, so every lane's four units exist, and a lane of
            // a sequence past nseq computes on whatever its rows hold -- its column of the MFMA never meets another column and the store
            // below skips it.  (With `if (j < H && seq_ok)` around each unit the compiler emitted four separate blocks, each
            // waiting for its own LDS reads and its own exp -> rcp chain: 1.7 us per step.)
            h16x8 bf[KT4];
#pragma unroll
            for (int kt = 0; kt < KT4; ++kt) bf[kt] = *(const h16x8*)(rb + kt * 32);
            h16x4 g4[4];
            float ct[4], cp[4], dyv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                g4[r] = *(const h16x4*)(gp + r * GS);
                ct[r] = cq[r * CS];
                cp[r] = cq[H * CS + r * CS];
            }
            {
                const h16x4 d4 = *(const h16x4*)(dyring + (dl + i * 16 * DYS));
#pragma unroll
                for (int r = 0; r < 4; ++r) dyv[r] = (float)d4[r];
            }
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int kt = 0; kt < KT4; ++kt) {
                if (kt & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kt], bf[kt], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kt], bf[kt], acc0, 0, 0, 0);
            }
            const f32x4 acc = acc0 + acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dh = acc[r] + dyv[r];
                const float ig = (float)g4[r][0], fg = (float)g4[r][1], gg = (float)g4[r][2], og = (float)g4[r][3];
                const float th = aero_tanh(ct[r]);
                const float d_o = dh * th;
                const float dct = dc[r] + dh * og * (1.f - th * th);
                const float d_i = dct * gg, d_g = dct * ig, d_f = dct * cp[r];
                dc[r] = dct * fg;
                const h16x4 dav = (h16x4){(h16)(d_i * ig * (1.f - ig)), (h16)(d_f * fg * (1.f - fg)), (h16)(d_g * (1.f - gg * gg)), (h16)(d_o * og * (1.f - og))};
                *(h16x4*)(wb + r * 4) = dav;
            }
            aero_phase_barrier();                               // LDS only: the da stores below (and the group loads) stay in flight
            if (two_pieces) {
                const h16* sb = dabuf + (cur ^ 1) * 16 * LD;
#pragma unroll
                for (int k = 0; k < NPC; ++k)
                    if (st_dst[k] >= 0) *(h16x8*)(da_out + st_dst[k] + tau * st_tau) = *(const h16x8*)(sb + st_src[k]);
            } else {
                for (int idx = tid; idx < 16 * per; idx += NT) {
                    const int sl = idx / per, e = idx - sl * per;
                    const int s2 = seq0 + sl;
                    if (s2 >= d.nseq) continue;
                    h16* dst = da_out + (((int64_t)s2 * W + tau) * 2 + dir) * H4 + e * vec;
                    *(h16x8*)dst = *(const h16x8*)&dabuf[(cur ^ 1) * 16 * LD + sl * LD + e * 8];
                }
            }
            cur ^= 1;
        }
        if (g + 1 < ngroups) {
            // every read of this group's rings happened before the last step's barrier; the store loop above reads dabuf only
            park_group();
            aero_phase_barrier();
        }
    }
}

static size_t aero_lstm_bwd_ring_lds(int H, int kt4, int G) {
    const int LD = kt4 * 32 + 8;
    return (size_t)2 * 16 * LD * 2 + (size_t)G * H * 80 * 2 + (size_t)(G + 1) * H * 20 * 4 + (size_t)G * 16 * (H + 4) * 2;
}

static int aero_lstm_bwd_kt4(int H) {
    const int need = (4 * H + 31) / 32;
    const int opts[8] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int i = 0; i < 8; ++i)
        if (opts[i] >= need) return opts[i];
    return -1;
}

static int aero_lstm_bwd_launch(const aero_lstm_bwd_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->dout || !d->whh_t || !d->save_gates || !d->save_c || !d->da) { *err = "lstm_bwd: null pointer"; return AERO_ERR_ARG; }
    if (d->H < 4 || (d->H % 4) || d->H > 128 || d->nseq < 1 || d->W < 1) { *err = "lstm_bwd: H must be a multiple of 4 in [4, 128]"; return AERO_ERR_UNSUPPORTED; }
    if (d->out_mode == 1 && (d->nframes < 1 || d->S < 1 || d->T < 1 || d->nseq % d->nframes)) { *err = "lstm_bwd: bad framing"; return AERO_ERR_ARG; }
    const int kt4 = aero_lstm_bwd_kt4(d->H);
    if (kt4 < 0) { *err = "lstm_bwd: hidden size unsupported"; return AERO_ERR_UNSUPPORTED; }
    AeroLstmBwdK p;
    p.d = *d;
    p.K4P = kt4 * 32;
    const int nw = (d->H + 15) / 16;
    dim3 grid((unsigned)((d->nseq + 15) / 16), 2), block((unsigned)(nw * 64));
    static const int ring = [] { const char* e = getenv("AERO_LSTM_BWD_RING"); return e ? atoi(e) : 1; }();
    if (ring && d->H % 16 == 0 && (((uintptr_t)d->dout | (uintptr_t)d->save_gates | (uintptr_t)d->save_c | (uintptr_t)d->da) & 15) == 0) {
        const size_t l4 = aero_lstm_bwd_ring_lds(d->H, kt4, 4), l2 = aero_lstm_bwd_ring_lds(d->H, kt4, 2);
        const bool g4 = l4 <= 150 * 1024;
        if (g4 || l2 <= 150 * 1024) {
#define AERO_LSTM_BWD_RING_GO(KT_)                                                                                                   \
            do {                                                                                                                     \
                if (g4) AERO_LAUNCH_DYN((aero_lstm_bwd_ring_kernel<KT_, 4>), grid, block, l4, stream, p);                           \
                else AERO_LAUNCH_DYN((aero_lstm_bwd_ring_kernel<KT_, 2>), grid, block, l2, stream, p);                              \
            } while (0)
            switch (kt4) {
                case 1: AERO_LSTM_BWD_RING_GO(1); break;
                case 2: AERO_LSTM_BWD_RING_GO(2); break;
                case 3: AERO_LSTM_BWD_RING_GO(3); break;
                case 4: AERO_LSTM_BWD_RING_GO(4); break;
                case 6: AERO_LSTM_BWD_RING_GO(6); break;
                case 8: AERO_LSTM_BWD_RING_GO(8); break;
                case 12: AERO_LSTM_BWD_RING_GO(12); break;
                default: AERO_LSTM_BWD_RING_GO(16); break;
            }
#undef AERO_LSTM_BWD_RING_GO
            return AERO_OK;
        }
    }
    switch (kt4) {
        case 1: AERO_LAUNCH(aero_lstm_bwd_kernel<1>, grid, block, stream, p); break;
        case 2: AERO_LAUNCH(aero_lstm_bwd_kernel<2>, grid, block, stream, p); break;
        case 3: AERO_LAUNCH(aero_lstm_bwd_kernel<3>, grid, block, stream, p); break;
        case 4: AERO_LAUNCH(aero_lstm_bwd_kernel<4>, grid, block, stream, p); break;
        case 6: AERO_LAUNCH(aero_lstm_bwd_kernel<6>, grid, block, stream, p); break;
        case 8: AERO_LAUNCH(aero_lstm_bwd_kernel<8>, grid, block, stream, p); break;
        case 12: AERO_LAUNCH(aero_lstm_bwd_kernel<12>, grid, block, stream, p); break;
        default: AERO_LAUNCH(aero_lstm_bwd_kernel<16>, grid, block, stream, p); break;
    }
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// LocalState backward (modules.py:94-127).  Per (row r, head h), keys t and queries s (softmax over the keys, dim 2):
//   S[t,s] = K_t.Q_s / sqrt(d) - |t-s| * D_s,  D_s = sum_f (f+1) sigmoid(dq[f,s]) / (2 sqrt(ndecay));  S[s,s] = -100
//   P = softmax_t S;  O_s = sum_t P[t,s] V_t
//   delta_s = O_s.dO_s;  dP[t,s] = V_t.dO_s;  dS = P (dP - delta_s), dS[s,s] = 0 (masked_fill: a constant)
//   dQ_s = sum_t dS K_t / sqrt(d);  dK_t = sum_s dS Q_s / sqrt(d);  dV_t = sum_s P dO_s;  dD_s = -sum_t |t-s| dS
// Two passes, no atomics: pass A gives a thread one QUERY (softmax statistics recomputed online, then dQ, the decay gradient and
// the per-query scalars L = m + log l, delta, D), pass B gives a thread one KEY (dK, dV) and walks the queries with those
// scalars.  The other side is staged through LDS in tiles of 64 as fp32.  VALU fp32 throughout (first version).
struct AeroAttnBwdK {
    aero_attn_bwd_desc d;
    int dh;                                                    // channels per head
};

// The key (pass A) / query (pass B) range of one owner is split over AERO_ATTN_KS lanes that are combined with wave shuffles: at the
// per-GPU batch of BASELINE config 5 a thread per query is only ~7 waves per CU of 3448 dependent iterations each.
#define AERO_ATTN_KS 4
#define AERO_ATTN_OWN (128 / AERO_ATTN_KS)                       /* owners (queries / keys) per 128-thread block */

template <int DP>
__global__ __launch_bounds__(128) void aero_attn_bwd_q_kernel(AeroAttnBwdK p) {
    __shared__ AERO_LDS_ALIGN float Ks[64][DP];
    __shared__ AERO_LDS_ALIGN float Vs[64][DP];
    const aero_attn_bwd_desc& d = p.d;
    const int dh = p.dh, T = d.T, Cc = d.C;
    const int h = blockIdx.y, r = blockIdx.z;
    const int part = threadIdx.x % AERO_ATTN_KS;
    const int s = blockIdx.x * AERO_ATTN_OWN + threadIdx.x / AERO_ATTN_KS;
    const bool live = s < T;
    const h16* base = (const h16*)d.qkvd + (int64_t)r * T * d.ld;
    const float qscale = aero_rsqrt((float)dh);
    float Q[DP], dO[DP], dQ[DP];
    float delta = 0.f, D = 0.f;
    float sg[8];
#pragma unroll
    for (int i = 0; i < DP; ++i) { Q[i] = 0.f; dO[i] = 0.f; dQ[i] = 0.f; }
    if (live) {
        const h16* row = base + (int64_t)s * d.ld;
        const h16* orow = (const h16*)d.out + ((int64_t)r * T + s) * Cc + h * dh;
        const h16* drow = (const h16*)d.dout + ((int64_t)r * T + s) * Cc + h * dh;
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (i < dh) {
                Q[i] = (float)row[h * dh + i] * qscale;
                dO[i] = (float)drow[i];
                delta += (float)orow[i] * dO[i];
            }
        const float dn = 0.5f * aero_rsqrt((float)(d.ndecay > 0 ? d.ndecay : 1));
        for (int f = 0; f < d.ndecay; ++f) {
            sg[f] = aero_sigmoid((float)row[3 * Cc + h * d.ndecay + f]);
            D += (float)(f + 1) * sg[f] * dn;
        }
    }
    float m = -1e30f, l = 0.f, dD = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) {                                              // combine the parts' softmax statistics
#pragma unroll
            for (int k = 1; k < AERO_ATTN_KS; k <<= 1) {
                const float mo = __shfl_xor(m, k), lo = __shfl_xor(l, k);
                const float mn = fmaxf(m, mo);
                l = l * aero_fast_exp(m - mn) + lo * aero_fast_exp(mo - mn);
                m = mn;
            }
        }
        const float inv_l = pass ? 1.f / l : 0.f;
        for (int t0 = 0; t0 < T; t0 += 64) {
            __syncthreads();
            for (int idx = threadIdx.x; idx < 64 * DP; idx += 128) {
                const int tt = idx / DP, i = idx - tt * DP;
                float kv = 0.f, vv = 0.f;
                if (t0 + tt < T && i < dh) {
                    const h16* rowk = base + (int64_t)(t0 + tt) * d.ld;
                    kv = (float)rowk[Cc + h * dh + i];
                    vv = (float)rowk[2 * Cc + h * dh + i];
                }
                Ks[tt][i] = kv;
                Vs[tt][i] = vv;
            }
            __syncthreads();
            if (!live) continue;
            const int tn = T - t0 < 64 ? T - t0 : 64;
            for (int tt = part; tt < tn; tt += AERO_ATTN_KS) {
                const int t = t0 + tt;
                float sv = 0.f;
#pragma unroll
                for (int i = 0; i < DP; ++i) sv += Ks[tt][i] * Q[i];
                const float dist = fabsf((float)(t - s));
                sv = (t == s) ? -100.f : sv - dist * D;
                if (!pass) {
                    const float mn = fmaxf(m, sv);
                    l = l * aero_fast_exp(m - mn) + aero_fast_exp(sv - mn);
                    m = mn;
                } else if (t != s) {
                    const float P = aero_fast_exp(sv - m) * inv_l;
                    float dP = 0.f;
#pragma unroll
                    for (int i = 0; i < DP; ++i) dP += Vs[tt][i] * dO[i];
                    const float dS = P * (dP - delta);
#pragma unroll
                    for (int i = 0; i < DP; ++i) dQ[i] += dS * Ks[tt][i];
                    dD -= dist * dS;
                }
            }
        }
    }
#pragma unroll
    for (int k = 1; k < AERO_ATTN_KS; k <<= 1) {
        dD += __shfl_xor(dD, k);
#pragma unroll
        for (int i = 0; i < DP; ++i) dQ[i] += __shfl_xor(dQ[i], k);
    }
    if (!live || part) return;
    h16* orow = (h16*)d.dqkvd + ((int64_t)r * T + s) * d.ld;
#pragma unroll
    for (int i = 0; i < DP; ++i)
        if (i < dh) orow[h * dh + i] = (h16)(dQ[i] * qscale);
    const float dn = 0.5f * aero_rsqrt((float)(d.ndecay > 0 ? d.ndecay : 1));
    const float dsc = d.decay_scale > 0.f ? d.decay_scale : 1.f;      // (these are orders of magnitude below dQ / dK / dV: own power-of-two scale)
    for (int f = 0; f < d.ndecay; ++f) orow[3 * Cc + h * d.ndecay + f] = (h16)(dD * dsc * (float)(f + 1) * dn * sg[f] * (1.f - sg[f]));
    float* st = d.qstats + (((int64_t)r * d.heads + h) * T + s) * 4;
    st[0] = m + logf(l);
    st[1] = delta;
    st[2] = D;
}

template <int DP>
__global__ __launch_bounds__(128) void aero_attn_bwd_kv_kernel(AeroAttnBwdK p) {
    __shared__ AERO_LDS_ALIGN float Qs[64][DP];
    __shared__ AERO_LDS_ALIGN float Os[64][DP];
    __shared__ AERO_LDS_ALIGN float St[64][4];
    const aero_attn_bwd_desc& d = p.d;
    const int dh = p.dh, T = d.T, Cc = d.C;
    const int h = blockIdx.y, r = blockIdx.z;
    const int part = threadIdx.x % AERO_ATTN_KS;
    const int t = blockIdx.x * AERO_ATTN_OWN + threadIdx.x / AERO_ATTN_KS;
    const bool live = t < T;
    const h16* base = (const h16*)d.qkvd + (int64_t)r * T * d.ld;
    const float qscale = aero_rsqrt((float)dh);
    float K[DP], V[DP], dK[DP], dV[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) { K[i] = 0.f; V[i] = 0.f; dK[i] = 0.f; dV[i] = 0.f; }
    if (live) {
        const h16* row = base + (int64_t)t * d.ld;
#pragma unroll
        for (int i = 0; i < DP; ++i)
            if (i < dh) { K[i] = (float)row[Cc + h * dh + i]; V[i] = (float)row[2 * Cc + h * dh + i]; }
    }
    const float* qst = d.qstats + ((int64_t)r * d.heads + h) * T * 4;
    for (int s0 = 0; s0 < T; s0 += 64) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 64 * DP; idx += 128) {
            const int ss = idx / DP, i = idx - ss * DP;
            float qv = 0.f, ov = 0.f;
            if (s0 + ss < T && i < dh) {
                qv = (float)base[(int64_t)(s0 + ss) * d.ld + h * dh + i] * qscale;
                ov = (float)((const h16*)d.dout)[((int64_t)r * T + s0 + ss) * Cc + h * dh + i];
            }
            Qs[ss][i] = qv;
            Os[ss][i] = ov;
        }
        for (int idx = threadIdx.x; idx < 64 * 4; idx += 128) {
            const int ss = idx >> 2, e = idx & 3;
            St[ss][e] = (s0 + ss < T) ? qst[(int64_t)(s0 + ss) * 4 + e] : 0.f;
        }
        __syncthreads();
        if (!live) continue;
        const int sn = T - s0 < 64 ? T - s0 : 64;
        for (int ss = part; ss < sn; ss += AERO_ATTN_KS) {
            const int s = s0 + ss;
            float sv = 0.f, dP = 0.f;
#pragma unroll
            for (int i = 0; i < DP; ++i) { sv += K[i] * Qs[ss][i]; dP += V[i] * Os[ss][i]; }
            sv = (t == s) ? -100.f : sv - fabsf((float)(t - s)) * St[ss][2];
            const float P = aero_fast_exp(sv - St[ss][0]);
            const float dS = (t == s) ? 0.f : P * (dP - St[ss][1]);
#pragma unroll
            for (int i = 0; i < DP; ++i) { dV[i] += P * Os[ss][i]; dK[i] += dS * Qs[ss][i]; }
        }
    }
#pragma unroll
    for (int k = 1; k < AERO_ATTN_KS; k <<= 1) {
#pragma unroll
        for (int i = 0; i < DP; ++i) { dK[i] += __shfl_xor(dK[i], k); dV[i] += __shfl_xor(dV[i], k); }
    }
    if (!live || part) return;
    h16* orow = (h16*)d.dqkvd + ((int64_t)r * T + t) * d.ld;
#pragma unroll
    for (int i = 0; i < DP; ++i)
        if (i < dh) { orow[Cc + h * dh + i] = (h16)dK[i]; orow[2 * Cc + h * dh + i] = (h16)dV[i]; }
}

// ---- MFMA form of the two passes (d.R*heads*T^2 pairs: 34 ms of a 131-ms step at B = 16 in the VALU form above) --------------
// v_mfma_f32_16x16x16_f16 throughout, ND = ceil(dh / 16) dimension tiles.  The score tile comes out of the MFMA with lane
// (g = lane >> 4, col = lane & 15) holding rows 4g..4g+3 of column col -- which IS the A-operand layout (i = col, k = 4g + e) of the
// transposed tile, so P and dS feed the second MFMA straight from registers (packed to fp16), as the forward kernel does.
//   pass A (a wave = 16 queries, walks key tiles): S^T = K Q^T, dP^T = V dO^T  ->  dS  ->  dQ += dS^T K  (A = dS, B = K^T from LDS)
//   pass B (a wave = 16 keys, walks query tiles):  S = Q K^T, dP = dO V^T      ->  P, dS  ->  dV += P^T dO, dK += dS^T Q
typedef h16 h16x4_t __attribute__((ext_vector_type(4)));
#define AERO_ATTN_TK 64                                          /* rows of the other side staged per LDS tile */

template <int ND>
__global__ __launch_bounds__(256) void aero_attn_bwd_q_mfma_kernel(AeroAttnBwdK p) {
    constexpr int DP = 16 * ND, LT = AERO_ATTN_TK + 4;
    __shared__ AERO_LDS_ALIGN h16 Kr[AERO_ATTN_TK * DP];        // [key][dim]
    __shared__ AERO_LDS_ALIGN h16 Vr[AERO_ATTN_TK * DP];
    __shared__ AERO_LDS_ALIGN h16 Kt[DP * LT];                   // [dim][key]
    const aero_attn_bwd_desc& d = p.d;
    const int dh = p.dh, T = d.T, Cc = d.C;
    const int h = blockIdx.y, r = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int s = blockIdx.x * 64 + wave * 16 + col;             // this lane's query (MFMA column)
    const bool live = s < T;
    const h16* base = (const h16*)d.qkvd + (int64_t)r * T * d.ld;
    const float qscale = aero_rsqrt((float)dh);
    // B operands, constant per wave: Q^T and dO^T, lane = (dims 4g..4g+3 of tile n, query col)
    h16x4_t qb[ND], ob[ND];
    float delta = 0.f, D = 0.f;
    float sg[8];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
        qb[n] = (h16x4_t){0, 0, 0, 0};
        ob[n] = (h16x4_t){0, 0, 0, 0};
    }
    if (live) {
        const h16* row = base + (int64_t)s * d.ld + h * dh;
        const h16* orow = (const h16*)d.out + ((int64_t)r * T + s) * Cc + h * dh;
        const h16* drow = (const h16*)d.dout + ((int64_t)r * T + s) * Cc + h * dh;
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = n * 16 + g * 4 + e;
                if (i < dh) { qb[n][e] = row[i]; ob[n][e] = drow[i]; }
            }
        for (int i = 0; i < dh; ++i) delta += (float)orow[i] * (float)drow[i];
        const h16* rowd = base + (int64_t)s * d.ld + 3 * Cc + h * d.ndecay;
        const float dn = 0.5f * aero_rsqrt((float)(d.ndecay > 0 ? d.ndecay : 1));
        for (int f = 0; f < d.ndecay; ++f) {
            sg[f] = aero_sigmoid((float)rowd[f]);
            D += (float)(f + 1) * sg[f] * dn;
        }
    }
    float m = -1e30f, l = 0.f, dD = 0.f, L = 0.f;
    f32x4 dq[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) dq[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) {
#pragma unroll
            for (int k = 16; k <= 32; k <<= 1) {                 // the four lane groups of a column hold different keys: combine
                const float mo = __shfl_xor(m, k), lo = __shfl_xor(l, k);
                const float mn = fmaxf(m, mo);
                l = l * aero_fast_exp(m - mn) + lo * aero_fast_exp(mo - mn);
                m = mn;
            }
            L = m + logf(l);
        }
        for (int t0 = 0; t0 < T; t0 += AERO_ATTN_TK) {
            __syncthreads();
            for (int idx = tid; idx < AERO_ATTN_TK * DP; idx += 256) {
                const int tt = idx / DP, i = idx - tt * DP;
                h16 kv = (h16)0, vv = (h16)0;
                if (t0 + tt < T && i < dh) {
                    const h16* rowk = base + (int64_t)(t0 + tt) * d.ld + h * dh + i;
                    kv = rowk[Cc];
                    vv = rowk[2 * Cc];
                }
                Kr[tt * DP + i] = kv;
                Vr[tt * DP + i] = vv;
                Kt[i * LT + tt] = kv;
            }
            __syncthreads();
#pragma unroll
            for (int kt = 0; kt < AERO_ATTN_TK / 16; ++kt) {
                if (t0 + kt * 16 >= T) break;
                f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int n = 0; n < ND; ++n) {
                    const h16x4_t ka = *(const h16x4_t*)&Kr[(kt * 16 + col) * DP + n * 16 + g * 4];
                    sc = __builtin_amdgcn_mfma_f32_16x16x16f16(ka, qb[n], sc, 0, 0, 0);
                    if (pass) {
                        const h16x4_t va = *(const h16x4_t*)&Vr[(kt * 16 + col) * DP + n * 16 + g * 4];
                        dp = __builtin_amdgcn_mfma_f32_16x16x16f16(va, ob[n], dp, 0, 0, 0);
                    }
                }
                h16x4_t dsv = (h16x4_t){0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = t0 + kt * 16 + g * 4 + e;
                    const float dist = fabsf((float)(t - s));
                    float sv = (t == s) ? -100.f : sc[e] * qscale - dist * D;
                    if (t >= T) sv = -1e30f;
                    if (!pass) {
                        const float mn = fmaxf(m, sv);
                        l = l * aero_fast_exp(m - mn) + aero_fast_exp(sv - mn);
                        m = mn;
                    } else if (t != s && t < T) {
                        const float P = aero_fast_exp(sv - L);
                        const float dS = P * (dp[e] - delta);
                        dsv[e] = (h16)dS;
                        dD -= dist * dS;
                    }
                }
                if (pass) {
#pragma unroll
                    for (int n = 0; n < ND; ++n) {
                        const h16x4_t kb = *(const h16x4_t*)&Kt[(n * 16 + col) * LT + kt * 16 + g * 4];
                        dq[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(dsv, kb, dq[n], 0, 0, 0);
                    }
                }
            }
        }
    }
    dD += __shfl_xor(dD, 16);
    dD += __shfl_xor(dD, 32);
    // dq[n][e]: query 4g + e of the wave, dimension n*16 + col
    h16* outb = (h16*)d.dqkvd + (int64_t)r * T * d.ld;
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int sq = blockIdx.x * 64 + wave * 16 + g * 4 + e, i = n * 16 + col;
            if (sq < T && i < dh) outb[(int64_t)sq * d.ld + h * dh + i] = (h16)(dq[n][e] * qscale);
        }
    if (!live || g) return;
    h16* orow = outb + (int64_t)s * d.ld;
    const float dn = 0.5f * aero_rsqrt((float)(d.ndecay > 0 ? d.ndecay : 1));
    const float dsc = d.decay_scale > 0.f ? d.decay_scale : 1.f;
    for (int f = 0; f < d.ndecay; ++f) orow[3 * Cc + h * d.ndecay + f] = (h16)(dD * dsc * (float)(f + 1) * dn * sg[f] * (1.f - sg[f]));
    float* st = d.qstats + (((int64_t)r * d.heads + h) * T + s) * 4;
    st[0] = L;
    st[1] = delta;
    st[2] = D;
}

template <int ND>
__global__ __launch_bounds__(256) void aero_attn_bwd_kv_mfma_kernel(AeroAttnBwdK p) {
    constexpr int DP = 16 * ND, LT = AERO_ATTN_TK + 4;
    __shared__ AERO_LDS_ALIGN h16 Qr[AERO_ATTN_TK * DP];        // [query][dim], pre-scaled by 1/sqrt(dh)
    __shared__ AERO_LDS_ALIGN h16 Or[AERO_ATTN_TK * DP];        // dO [query][dim]
    __shared__ AERO_LDS_ALIGN h16 Qt[DP * LT];                   // [dim][query]
    __shared__ AERO_LDS_ALIGN h16 Ot[DP * LT];
    __shared__ AERO_LDS_ALIGN float St[AERO_ATTN_TK][4];
    const aero_attn_bwd_desc& d = p.d;
    const int dh = p.dh, T = d.T, Cc = d.C;
    const int h = blockIdx.y, r = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int t = blockIdx.x * 64 + wave * 16 + col;             // this lane's key (MFMA column)
    const h16* base = (const h16*)d.qkvd + (int64_t)r * T * d.ld;
    const float qscale = aero_rsqrt((float)dh);
    h16x4_t kb[ND], vb[ND];                                      // B operands: K^T, V^T (dims 4g..4g+3 of tile n, key col)
#pragma unroll
    for (int n = 0; n < ND; ++n) {
        kb[n] = (h16x4_t){0, 0, 0, 0};
        vb[n] = (h16x4_t){0, 0, 0, 0};
    }
    if (t < T) {
        const h16* row = base + (int64_t)t * d.ld + h * dh;
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = n * 16 + g * 4 + e;
                if (i < dh) { kb[n][e] = row[Cc + i]; vb[n][e] = row[2 * Cc + i]; }
            }
    }
    f32x4 dk[ND], dv[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) { dk[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float* qst = d.qstats + ((int64_t)r * d.heads + h) * T * 4;
    for (int s0 = 0; s0 < T; s0 += AERO_ATTN_TK) {
        __syncthreads();
        for (int idx = tid; idx < AERO_ATTN_TK * DP; idx += 256) {
            const int ss = idx / DP, i = idx - ss * DP;
            h16 qv = (h16)0, ov = (h16)0;
            if (s0 + ss < T && i < dh) {
                qv = (h16)((float)base[(int64_t)(s0 + ss) * d.ld + h * dh + i] * qscale);
                ov = ((const h16*)d.dout)[((int64_t)r * T + s0 + ss) * Cc + h * dh + i];
            }
            Qr[ss * DP + i] = qv;
            Or[ss * DP + i] = ov;
            Qt[i * LT + ss] = qv;
            Ot[i * LT + ss] = ov;
        }
        for (int idx = tid; idx < AERO_ATTN_TK * 4; idx += 256) {
            const int ss = idx >> 2, e = idx & 3;
            St[ss][e] = (s0 + ss < T) ? qst[(int64_t)(s0 + ss) * 4 + e] : (e == 0 ? 1e30f : 0.f);      // L = +inf: P = 0 beyond T
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < AERO_ATTN_TK / 16; ++qt) {
            if (s0 + qt * 16 >= T) break;
            f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const h16x4_t qa = *(const h16x4_t*)&Qr[(qt * 16 + col) * DP + n * 16 + g * 4];
                const h16x4_t oa = *(const h16x4_t*)&Or[(qt * 16 + col) * DP + n * 16 + g * 4];
                sc = __builtin_amdgcn_mfma_f32_16x16x16f16(qa, kb[n], sc, 0, 0, 0);       // rows: queries 4g + e, column: key col
                dp = __builtin_amdgcn_mfma_f32_16x16x16f16(oa, vb[n], dp, 0, 0, 0);
            }
            h16x4_t pv = (h16x4_t){0, 0, 0, 0}, dsv = (h16x4_t){0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ss = qt * 16 + g * 4 + e, sq = s0 + ss;
                const float sv = (t == sq) ? -100.f : sc[e] - fabsf((float)(t - sq)) * St[ss][2];
                const float P = aero_fast_exp(sv - St[ss][0]);
                pv[e] = (h16)P;
                dsv[e] = (h16)((t == sq) ? 0.f : P * (dp[e] - St[ss][1]));
            }
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const h16x4_t ob = *(const h16x4_t*)&Ot[(n * 16 + col) * LT + qt * 16 + g * 4];
                const h16x4_t qb = *(const h16x4_t*)&Qt[(n * 16 + col) * LT + qt * 16 + g * 4];
                dv[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(pv, ob, dv[n], 0, 0, 0);    // rows: keys 4g + e of the wave, column: dim
                dk[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(dsv, qb, dk[n], 0, 0, 0);
            }
        }
    }
    h16* outb = (h16*)d.dqkvd + (int64_t)r * T * d.ld;
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tk = blockIdx.x * 64 + wave * 16 + g * 4 + e, i = n * 16 + col;
            if (tk < T && i < dh) {
                outb[(int64_t)tk * d.ld + Cc + h * dh + i] = (h16)dk[n][e];
                outb[(int64_t)tk * d.ld + 2 * Cc + h * dh + i] = (h16)dv[n][e];
            }
        }
}

static int aero_attn_bwd_launch(const aero_attn_bwd_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->qkvd || !d->out || !d->dout || !d->dqkvd || !d->qstats) { *err = "localstate_bwd: null pointer"; return AERO_ERR_ARG; }
    if (d->R < 1 || d->T < 1 || d->heads < 1 || d->C % d->heads || d->ndecay < 0 || d->ndecay > 8 || d->ld < 3 * d->C + d->heads * d->ndecay) {
        *err = "localstate_bwd: bad geometry"; return AERO_ERR_ARG;
    }
    if (d->R > 65535 || d->heads > 65535) { *err = "localstate_bwd: too many rows for one launch"; return AERO_ERR_ARG; }
    AeroAttnBwdK p;
    p.d = *d;
    p.dh = d->C / d->heads;
    if (p.dh > 32) { *err = "localstate_bwd: more than 32 channels per head"; return AERO_ERR_UNSUPPORTED; }
    static int valu = -1;                                        // AERO_ATTN_BWD_VALU=1: the fp32 VALU form (A/B, and dh > 32 never reaches here)
    if (valu < 0) { const char* e = getenv("AERO_ATTN_BWD_VALU"); valu = (e && e[0] == '1') ? 1 : 0; }
    if (!valu) {
        dim3 mgrid((unsigned)((d->T + 63) / 64), (unsigned)d->heads, (unsigned)d->R), mblock(256);
        if (p.dh <= 16) {
            AERO_LAUNCH(aero_attn_bwd_q_mfma_kernel<1>, mgrid, mblock, stream, p);
            AERO_LAUNCH(aero_attn_bwd_kv_mfma_kernel<1>, mgrid, mblock, stream, p);
        } else {
            AERO_LAUNCH(aero_attn_bwd_q_mfma_kernel<2>, mgrid, mblock, stream, p);
            AERO_LAUNCH(aero_attn_bwd_kv_mfma_kernel<2>, mgrid, mblock, stream, p);
        }
        return AERO_OK;
    }
    dim3 grid((unsigned)((d->T + AERO_ATTN_OWN - 1) / AERO_ATTN_OWN), (unsigned)d->heads, (unsigned)d->R), block(128);
#define AERO_ATTN_BWD_GO(DP_)                                                         \
    do {                                                                              \
        AERO_LAUNCH(aero_attn_bwd_q_kernel<DP_>, grid, block, stream, p);             \
        AERO_LAUNCH(aero_attn_bwd_kv_kernel<DP_>, grid, block, stream, p);            \
    } while (0)
    if (p.dh <= 4) AERO_ATTN_BWD_GO(4);
    else if (p.dh <= 8) AERO_ATTN_BWD_GO(8);
    else if (p.dh <= 12) AERO_ATTN_BWD_GO(12);
    else if (p.dh <= 16) AERO_ATTN_BWD_GO(16);
    else if (p.dh <= 24) AERO_ATTN_BWD_GO(24);
    else AERO_ATTN_BWD_GO(32);
#undef AERO_ATTN_BWD_GO
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Spectral loss of one resolution (stft_loss.py:11-27: mag = sqrt(clamp(re^2 + im^2, 1e-7)); :30-64: spectral convergence
// ||ymag - xmag||_F / ||ymag||_F and L1 of the log magnitudes) on the complex64 STFTs zx (prediction), zy (target) the
// normalised kernel produced: power = (re^2 + im^2) * pscale (pscale = n_fft restores torch.stft's un-normalised scale).
//   sums[0] = sum (ymag - xmag)^2,  sums[1] = sum ymag^2,  sums[2] = sum |log ymag - log xmag|       (doubles)
// Block partials are written to `part` and added in block order (deterministic).
__global__ __launch_bounds__(256) void aero_stft_loss_sums_kernel(const f32x2* zx, const f32x2* zy, int64_t n, float pscale, double* part) {
    __shared__ double red[3][4];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const f32x2 x = zx[i], y = zy[i];
        const float mx = sqrtf(fmaxf((x[0] * x[0] + x[1] * x[1]) * pscale, 1e-7f));
        const float my = sqrtf(fmaxf((y[0] * y[0] + y[1] * y[1]) * pscale, 1e-7f));
        a += (double)((my - mx) * (my - mx));
        b += (double)(my * my);
        c += (double)fabsf(logf(my) - logf(mx));
    }
    a = aero_wave_sum(a); b = aero_wave_sum(b); c = aero_wave_sum(c);
    const int lane = aero_lane(), wave = aero_wave();
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
    __syncthreads();
    if (threadIdx.x < 3) part[(int64_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ void aero_stft_loss_finish_kernel(const double* part, int nblk, double* sums) {
    if (threadIdx.x < 3) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += part[(int64_t)k * 3 + threadIdx.x];
        sums[threadIdx.x] = s;
    }
}

static int aero_stft_loss_sums_launch(const float* zx, const float* zy, int64_t n, float pscale, double* part, int npart, double* sums,
                                      hipStream_t stream, const char** err) {
    if (!zx || !zy || !part || !sums || n < 1 || npart < 1) { *err = "stft_loss_sums: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > npart) nb = npart;
    AERO_LAUNCH(aero_stft_loss_sums_kernel, dim3((unsigned)nb), dim3(256), stream, (const f32x2*)zx, (const f32x2*)zy, n, pscale, part);
    AERO_LAUNCH(aero_stft_loss_finish_kernel, dim3(1), dim3(64), stream, (const double*)part, (int)nb, sums);
    return AERO_OK;
}

// gradient of  w_sc * sqrt(sums0 / sums1) + w_mag * sums2 / n  w.r.t. the (normalised) STFT of the prediction, times the upstream
// scalar gradients gout[0] (spectral-convergence term) and gout[1] (log-magnitude term) read from device memory:
//   dL/dxmag = w_sc g0 (xmag - ymag) / (sqrt(sums0) sqrt(sums1)) + w_mag g1 sign(log xmag - log ymag) / (n xmag)
//   dxmag/d(re, im) = (re, im) * pscale / xmag  where the clamp is inactive, else 0
__global__ __launch_bounds__(256) void aero_stft_loss_bwd_kernel(const f32x2* zx, const f32x2* zy, int64_t n, float pscale, const double* sums,
                                                                 float w_sc, float w_mag, const float* gout, f32x2* g) {
    const float g0 = gout ? gout[0] : 1.f, g1 = gout ? gout[1] : 1.f;
    const double den = sqrt(sums[0]) * sqrt(sums[1]);
    const float csc = den > 0.0 ? (float)((double)(w_sc * g0) / den) : 0.f;
    const float cmag = w_mag * g1 / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const f32x2 x = zx[i], y = zy[i];
        const float px = (x[0] * x[0] + x[1] * x[1]) * pscale;
        const float mx = sqrtf(fmaxf(px, 1e-7f));
        const float my = sqrtf(fmaxf((y[0] * y[0] + y[1] * y[1]) * pscale, 1e-7f));
        float dm = csc * (mx - my);
        const float dl = logf(mx) - logf(my);
        dm += dl > 0.f ? cmag / mx : (dl < 0.f ? -cmag / mx : 0.f);
        const float k = px > 1e-7f ? dm * pscale / mx : 0.f;
        g[i] = (f32x2){x[0] * k, x[1] * k};
    }
}

static int aero_stft_loss_bwd_launch(const float* zx, const float* zy, int64_t n, float pscale, const double* sums, float w_sc, float w_mag,
                                     const float* gout, float* g, hipStream_t stream, const char** err) {
    if (!zx || !zy || !sums || !g || n < 1) { *err = "stft_loss_bwd: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    AERO_LAUNCH(aero_stft_loss_bwd_kernel, dim3((unsigned)nb), dim3(256), stream, (const f32x2*)zx, (const f32x2*)zy, n, pscale, sums, w_sc,
                w_mag, gout, (f32x2*)g);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Adjoint of the centred, reflect-padded, normalised STFT (aero_stft_fwd; torch.stft of stft_loss.py:22 up to the factor the
// loss kernels carry):  dx = fold( sum_t shift_t( w * Re irdft'(G_t) ) )  with
//   irdft'(G)[m] = n_fft^-1/2 [ G_0.re + sum_{0<k<n} Re(G_k e^{+2 pi i k m / n_fft}) + G_n.re (-1)^m ]        (n = n_fft/2)
// (every bin with weight 1: the forward is one-sided, not Hermitian-doubled).  aero_irfft_frames does the per-frame inverse
// real transform (one frame per wavefront, the Stockham FFT of k_stft.h on the Hermitian-packed half-length problem) and writes
// the windowed frames; aero_stft_adj_fold overlap-adds them and folds the reflect padding back onto the signal.
struct AeroIrfftK {
    const f32x2* g; const float* window; float* frames;
    int nsig, nb, T, n_fft;
    float out_scale;
};

__global__ __launch_bounds__(256) void aero_irfft_frames_kernel(AeroIrfftK p) {
    const int n = p.n_fft >> 1;
    f32x2* tw = (f32x2*)AERO_DYN_SMEM;
    const int lane = aero_lane(), wave = aero_uniform(aero_wave());
    f32x2* a = tw + n + (size_t)wave * 2 * n;
    f32x2* b = a + n;
    aero_fft_init_twiddles(tw, p.n_fft);
    __syncthreads();
    const int sig = blockIdx.y;
    const int t = blockIdx.x * 4 + wave;
    if (t >= p.T) return;
    const f32x2* X = p.g + (int64_t)sig * p.nb * p.T + t;
    auto unpack = [&](f32x2 xa, f32x2 xb, int k) -> f32x2 {
        const f32x2 E = (xa + xb) * 0.5f;
        const f32x2 D = (xa - xb) * 0.5f;
        const f32x2 O = aero_cmul(D, (f32x2){tw[k][0], -tw[k][1]});
        return (f32x2){E[0] - O[1], -(E[1] + O[0])};
    };
    for (int k = lane; k <= (n >> 1); k += 64) {
        if (k == 0) {
            const float dc = X[0][0];
            const float ny = p.nb > n ? X[(int64_t)n * p.T][0] : 0.f;
            a[0] = unpack((f32x2){dc, 0.f}, (f32x2){ny, 0.f}, 0);
        } else {
            const f32x2 xa = X[(int64_t)k * p.T] * 0.5f, xq = X[(int64_t)(n - k) * p.T] * 0.5f;
            const f32x2 z0 = unpack(xa, (f32x2){xq[0], -xq[1]}, k);
            const f32x2 z1 = unpack(xq, (f32x2){xa[0], -xa[1]}, n - k);
            a[k] = z0;
            a[n - k] = z1;
        }
    }
    aero_wave_sync();
    f32x2* R = aero_fft_wave<0>(a, b, n, tw);
    float* out = p.frames + ((int64_t)sig * p.T + t) * p.n_fft;
    for (int m = lane; m < n; m += 64) {
        const f32x2 v = R[m];
        *(f32x2*)(out + 2 * m) = (f32x2){v[0] * p.window[2 * m] * p.out_scale, -v[1] * p.window[2 * m + 1] * p.out_scale};
    }
}

static int aero_irfft_frames_launch(const float* g, int nsig, int nb, int T, int n_fft, const float* window, float* frames, hipStream_t stream,
                                    const char** err) {
    if (!g || !window || !frames) { *err = "irfft_frames: null pointer"; return AERO_ERR_ARG; }
    const int n = n_fft / 2;
    if (n_fft < 16 || (1 << aero_ilog2(n_fft)) != n_fft || n > AERO_STFT_MAX_N) { *err = "irfft_frames: n_fft must be a power of two in [16,2048]"; return AERO_ERR_UNSUPPORTED; }
    if ((nb != n && nb != n + 1) || nsig < 1 || T < 1 || nsig > 65535) { *err = "irfft_frames: bad geometry"; return AERO_ERR_ARG; }
    AeroIrfftK p;
    p.g = (const f32x2*)g; p.window = window; p.frames = frames; p.nsig = nsig; p.nb = nb; p.T = T; p.n_fft = n_fft;
    p.out_scale = 2.0f / sqrtf((float)n_fft);
    const size_t lds = (size_t)n * 9 * sizeof(f32x2);
    AERO_LAUNCH_DYN(aero_irfft_frames_kernel, dim3((unsigned)((T + 3) / 4), (unsigned)nsig), dim3(256), lds, stream, p);
    return AERO_OK;
}

__global__ __launch_bounds__(256) void aero_stft_adj_fold_kernel(const float* frames, float* dx, int T, int n_fft, int hop, int L, int accumulate) {
    const int sig = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= L) return;
    const int P = n_fft >> 1;
    const float* fr = frames + (int64_t)sig * T * n_fft;
    auto padded = [&](int j) -> float {
        int tb = j / hop;
        if (tb > T - 1) tb = T - 1;
        float s = 0.f;
        for (int t = tb; t >= 0 && j - t * hop < n_fft; --t) s += fr[(int64_t)t * n_fft + (j - t * hop)];
        return s;
    };
    float v = padded(m + P);
    if (m >= 1 && m <= P) v += padded(P - m);
    if (m <= L - 2 && m >= L - 1 - P) v += padded(P + 2 * L - 2 - m);
    float* o = dx + (int64_t)sig * L + m;
    *o = accumulate ? *o + v : v;
}

static int aero_stft_adj_fold_launch(const float* frames, float* dx, int nsig, int T, int n_fft, int hop, int L, int accumulate, hipStream_t stream,
                                     const char** err) {
    if (!frames || !dx || nsig < 1 || T < 1 || hop < 1 || L <= n_fft / 2 || nsig > 65535) { *err = "stft_adj_fold: bad arguments"; return AERO_ERR_ARG; }
    AERO_LAUNCH(aero_stft_adj_fold_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)nsig), dim3(256), stream, frames, dx, T, n_fft, hop, L, accumulate);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Element-wise plumbing of the gradient path.
__global__ __launch_bounds__(256) void aero_axpy_f16_kernel(const h16* a, const h16* b, h16* dst, int64_t n, float sb) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const h16x8 va = *(const h16x8*)(a + i), vb = *(const h16x8*)(b + i);
        h16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (h16)((float)va[e] + sb * (float)vb[e]);
        *(h16x8*)(dst + i) = r;
    } else {
        for (int64_t e = i; e < n; ++e) dst[e] = (h16)((float)a[e] + sb * (float)b[e]);
    }
}

static int aero_axpy_f16_launch(const void* a, const void* b, void* dst, int64_t n, float sb, hipStream_t stream, const char** err) {
    if (!a || !b || !dst || n < 1 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)dst) & 15)) { *err = "add_f16: bad arguments"; return AERO_ERR_ARG; }
    AERO_LAUNCH(aero_axpy_f16_kernel, dim3((unsigned)((n / 8 + 256) / 256)), dim3(256), stream, (const h16*)a, (const h16*)b, (h16*)dst, n, sb);
    return AERO_OK;
}

// amax[0] = max |x[i] * item_scale[i / n_per_item]| as the bit pattern of a non-negative float (atomicMax on uint32; the caller zeroes it)
__global__ __launch_bounds__(256) void aero_absmax_f32_kernel(const float* x, int64_t n_per_item, const float* item_scale, unsigned int* amax) {
    __shared__ float red[4];
    const int item = blockIdx.y;
    const float sc = item_scale ? fabsf(item_scale[item]) : 1.f;
    const float* src = x + (int64_t)item * n_per_item;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(src[i]) * sc);
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if (aero_lane() == 0) red[aero_wave()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m == m && m < 3.0e38f) atomicMax(amax, __float_as_uint(m));
        else atomicMax(amax, 0x7f7fffffu);                   // inf / nan upstream: the largest finite pattern (scale collapses to ~0)
    }
}

// dst = fp16(src * item_scale[item] * S),  S = 2^floor(log2(target / amax)) (1 if amax == 0);  scale_out = {S, 1/S}
__global__ __launch_bounds__(256) void aero_scale_cast_kernel(const float* x, int64_t n_per_item, const float* item_scale, const unsigned int* amax,
                                                              float target, h16* dst, float* scale_out) {
    const int item = blockIdx.y;
    const float am = __uint_as_float(amax[0]);
    float S = 1.f;
    if (am > 0.f) S = exp2f(floorf(log2f(target / am)));
    if (!(S > 1e-30f)) S = 1e-30f;
    if (S > 1e30f) S = 1e30f;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { scale_out[0] = S; scale_out[1] = 1.f / S; }
    const float sc = (item_scale ? item_scale[item] : 1.f) * S;
    const float* src = x + (int64_t)item * n_per_item;
    h16* d = dst + (int64_t)item * n_per_item;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_item; i += (int64_t)gridDim.x * 256) d[i] = (h16)(src[i] * sc);
}

static int aero_scale_cast_launch(const float* x, int nitems, int64_t n_per_item, const float* item_scale, unsigned int* amax, float target,
                                  void* dst, float* scale_out, hipStream_t stream, const char** err) {
    if (!x || !amax || !dst || !scale_out || nitems < 1 || n_per_item < 1 || nitems > 65535 || !(target > 0.f)) { *err = "scale_cast: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n_per_item + 255) / 256;
    if (nb > 1024) nb = 1024;
    dim3 grid((unsigned)nb, (unsigned)nitems);
    AERO_LAUNCH(aero_absmax_f32_kernel, grid, dim3(256), stream, x, n_per_item, item_scale, amax);
    AERO_LAUNCH(aero_scale_cast_kernel, grid, dim3(256), stream, x, n_per_item, item_scale, (const unsigned int*)amax, target, (h16*)dst, scale_out);
    return AERO_OK;
}

__global__ __launch_bounds__(256) void aero_scale_f32_kernel(float* x, int64_t n, const float* scale) {
    const float s = scale[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= s;
}

static int aero_scale_f32_launch(float* x, int64_t n, const float* scale, hipStream_t stream, const char** err) {
    if (!x || !scale || n < 1) { *err = "scale_f32: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    AERO_LAUNCH(aero_scale_f32_kernel, dim3((unsigned)nb), dim3(256), stream, x, n, scale);
    return AERO_OK;
}

// Running-statistics bookkeeping of nn.BatchNorm in training mode (torch/nn/modules/batchnorm.py: running = (1 - m) running + m batch,
// the variance unbiased, num_batches_tracked += 1) from the fp64 channel sums the norm kernels accumulated: one launch instead of a
// dozen parameter-sized torch kernels per BatchNorm layer.
__global__ __launch_bounds__(256) void aero_bn_running_kernel(const double* st, int nc, double n, float mom, float* rm, float* rv, long long* nbt) {
    for (int c = threadIdx.x; c < nc; c += 256) {
        const double mean = st[2 * c] / n;
        double var = st[2 * c + 1] / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double unb = var * (n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0));
        rm[c] = rm[c] * (1.f - mom) + mom * (float)mean;
        rv[c] = rv[c] * (1.f - mom) + mom * (float)unb;
    }
    if (threadIdx.x == 0 && nbt) nbt[0] += 1;
}

// Re-packing of the weight images after an optimizer step.  Every packed image of the training engine (padded / tiled / flipped /
// interleaved fp16 or fp32 copies of parameters, aero_amd/pack.py, backward.py) is pure data movement: element i of the arena is
// element table[i] of the parameters laid end to end (or zero, table[i] < 0).  The parameters need not be contiguous in memory:
// `ptrs[p]` is the base of parameter p and `starts[p]` its first flat index (starts[nparam] = total).  One launch per arena replaces
// the ~700 little layout kernels torch ran per step for the same bytes (aero_amd/repack.py builds and checks the tables).
#define AERO_GATHER_MAXP 1024
__global__ __launch_bounds__(256) void aero_gather_pack_kernel(const float* const* ptrs, const int* starts, int nparam, const int* table,
                                                               h16* dst16, float* dst32, int64_t n) {
    __shared__ int st[AERO_GATHER_MAXP + 1];
    __shared__ const float* pp[AERO_GATHER_MAXP];
    for (int i = threadIdx.x; i <= nparam; i += 256) st[i] = starts[i];
    for (int i = threadIdx.x; i < nparam; i += 256) pp[i] = ptrs[i];
    __syncthreads();
    int p = 0;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += (int64_t)gridDim.x * 1024) {
        const int nv = n - i0 < 4 ? (int)(n - i0) : 4;        // (arenas are padded to multiples of 4: nv == 4 in practice)
        int idx[4];
        if (nv == 4) {
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const i32x4 t = *(const i32x4*)(table + i0);
            idx[0] = t[0]; idx[1] = t[1]; idx[2] = t[2]; idx[3] = t[3];
        } else {
            for (int j = 0; j < 4; ++j) idx[j] = j < nv ? table[i0 + j] : -1;
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int fi = idx[j];
            if (fi < 0) { v[j] = 0.f; continue; }
            if (fi < st[p] || fi >= st[p + 1]) {               // neighbours mostly come from the same parameter
                int lo = 0, hi = nparam - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (st[mid] <= fi) lo = mid; else hi = mid - 1;
                }
                p = lo;
            }
            v[j] = pp[p][fi - st[p]];
        }
        if (dst16) {
            for (int j = 0; j < nv; ++j) dst16[i0 + j] = (h16)v[j];
        } else {
            for (int j = 0; j < nv; ++j) dst32[i0 + j] = v[j];
        }
    }
}

static int aero_gather_pack_launch(const float* const* ptrs, const int* starts, int nparam, const int* table, void* dst, int64_t n, int dst_f16,
                                   hipStream_t stream, const char** err) {
    if (!ptrs || !starts || !table || !dst || n < 1 || nparam < 1) { *err = "gather_pack: bad arguments"; return AERO_ERR_ARG; }
    if (nparam > AERO_GATHER_MAXP) { *err = "gather_pack: more than 1024 parameters"; return AERO_ERR_UNSUPPORTED; }
    if (((uintptr_t)table & 15) || ((uintptr_t)dst & 15)) { *err = "gather_pack: table / dst must be 16-byte aligned"; return AERO_ERR_ARG; }
    int64_t nb = (n + 1023) / 1024;
    if (nb > 8192) nb = 8192;
    AERO_LAUNCH(aero_gather_pack_kernel, dim3((unsigned)nb), dim3(256), stream, ptrs, starts, nparam, table, dst_f16 ? (h16*)dst : nullptr,
                dst_f16 ? nullptr : (float*)dst, n);
    return AERO_OK;
}

// Re-normalisation of an fp16 gradient between stages of the backward: v = a / Sa + b / Sb (b optional), S = 2^floor(log2(target / max|v|)),
// out = fp16(v * S), scale_out = {S, 1/S}.  sa / sb: the {S, 1/S} pairs the operands carry (device memory; NULL = 1).  Powers of two: exact
// unless a value leaves the fp16 range, which is what this pass prevents (gradients may grow or shrink by orders of magnitude from
// one encoder level to the next).
__global__ __launch_bounds__(256) void aero_rescale_amax_kernel(const h16* a, const float* sa, const h16* b, const float* sb, int64_t n, unsigned int* amax) {
    __shared__ float red[4];
    const float ia = sa ? sa[1] : 1.f, ib = sb ? sb[1] : 1.f;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = (float)a[i] * ia + (b ? (float)b[i] * ib : 0.f);
        m = fmaxf(m, fabsf(v));
    }
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if (aero_lane() == 0) red[aero_wave()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m == m && m < 3.0e38f) atomicMax(amax, __float_as_uint(m));
        else atomicMax(amax, 0x7f7fffffu);
    }
}

__global__ __launch_bounds__(256) void aero_rescale_apply_kernel(const h16* a, const float* sa, const h16* b, const float* sb, int64_t n, const unsigned int* amax,
                                                                 float target, h16* out, float* scale_out) {
    const float ia = sa ? sa[1] : 1.f, ib = sb ? sb[1] : 1.f;
    const float am = __uint_as_float(amax[0]);
    float S = 1.f;
    if (am > 0.f) S = exp2f(floorf(log2f(target / am)));
    if (!(S > 1e-30f)) S = 1e-30f;
    if (S > 1e30f) S = 1e30f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = S; scale_out[1] = 1.f / S; }
    const float fa = ia * S, fb = ib * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (h16)((float)a[i] * fa + (b ? (float)b[i] * fb : 0.f));
}

static int aero_rescale_f16_launch(const void* a, const float* sa, const void* b, const float* sb, int64_t n, unsigned int* amax, float target, void* out,
                                   float* scale_out, hipStream_t stream, const char** err) {
    if (!a || !amax || !out || !scale_out || n < 1 || !(target > 0.f)) { *err = "rescale_f16: bad arguments"; return AERO_ERR_ARG; }
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    AERO_LAUNCH(aero_rescale_amax_kernel, dim3((unsigned)nb), dim3(256), stream, (const h16*)a, sa, (const h16*)b, sb, n, amax);
    AERO_LAUNCH(aero_rescale_apply_kernel, dim3((unsigned)nb), dim3(256), stream, (const h16*)a, sa, (const h16*)b, sb, n, (const unsigned int*)amax, target,
                (h16*)out, scale_out);
    return AERO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Diagnostic (tools/istft_concurrency.py): a bystander kernel that shares CUs with whatever runs on another stream and
// counts (0) words of its OWN LDS allocation that changed under it and (1) global loads of a constant pattern that came back wrong.
__global__ __launch_bounds__(256) void aero_probe_kernel(const unsigned* pattern, int npat, int rounds, unsigned long long* counters) {
    unsigned* lds = (unsigned*)AERO_DYN_SMEM;
    const int nw = 12 * 1024;                                   // 48 KiB
    for (int i = threadIdx.x; i < nw; i += 256) lds[i] = (unsigned)i * 2654435761u ^ blockIdx.x;
    __syncthreads();
    unsigned long long bad_lds = 0, bad_ld = 0, bad_tw = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < nw; i += 256) bad_lds += lds[i] != ((unsigned)i * 2654435761u ^ blockIdx.x);
        for (int j = 0; j < 8; ++j) {                           // 8-byte loads at an odd 4008-byte row pitch, as the iSTFT's unpack issues them
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const int row = (int)(((long long)blockIdx.x * 977 + (long long)r * 131 + j * 4099) % 500);
            const int idx = row * 1002 + 2 * (threadIdx.x & 15) + 34 * (threadIdx.x >> 4);       // 16 lanes = 128 contiguous bytes
            if (idx + 1 < npat) {
                const u32x2 v = *(const u32x2*)(pattern + idx);
                bad_ld += (v[0] != (unsigned)idx * 2246822519u) + (v[1] != (unsigned)(idx + 1) * 2246822519u);
            }
        }
        __syncthreads();
        // the iSTFT's twiddle table: thread k writes exp(-2 pi i k / 512) computed on the spot, everybody reads entries others wrote
        {
            f32x2* tw = (f32x2*)(lds + nw);                     // 2 KiB behind the pattern words
            float sn, cs;
#ifdef AERO_EMU
            sn = (float)sin(-2.0 * 3.14159265358979323846 * threadIdx.x / 512.0); cs = (float)cos(-2.0 * 3.14159265358979323846 * threadIdx.x / 512.0);
#else
            sincospif(-2.0f * (float)threadIdx.x / 512.f, &sn, &cs);
#endif
            tw[threadIdx.x] = (f32x2){cs, sn};
            __syncthreads();
            const int k2 = (threadIdx.x * 7 + r) & 255;
            float sn2, cs2;
#ifdef AERO_EMU
            sn2 = (float)sin(-2.0 * 3.14159265358979323846 * k2 / 512.0); cs2 = (float)cos(-2.0 * 3.14159265358979323846 * k2 / 512.0);
#else
            sincospif(-2.0f * (float)k2 / 512.f, &sn2, &cs2);
#endif
            const f32x2 t = tw[k2];
            bad_tw += (t[0] != cs2) + (t[1] != sn2);
            __syncthreads();
        }
    }
    if (bad_lds) atomicAdd(counters, bad_lds);
    if (bad_ld) atomicAdd(counters + 1, bad_ld);
    if (bad_tw) atomicAdd(counters + 2, bad_tw);
}
