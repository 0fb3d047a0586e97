// k_attn.h -- LocalState attention core (reference modules.py:101-124) as a flash-style MFMA kernel: the
// T x T score matrix is never materialised.
//
//   S[t][s] = K_t . Q_s / sqrt(dh) - |t-s| * D_s,  S[s][s] = -100,  W = softmax over keys t,  O[:,s] = sum_t W[t][s] V_t
// The learned distance-decay bias (modules.py:112-117) collapses analytically to one slope per query:
//   sum_f -(f+1)|t-s|/sqrt(nd) * sigmoid(d_fs)/2 = -|t-s| * D_s,   D_s = sum_f (f+1) sigmoid(d_fs) / (2 sqrt(nd)).
//
// One block = 128 queries of one (row, head), 8 waves x 16 queries.  Keys are processed 32 at a time:
//   two v_mfma_f32_16x16x32_f16 give S for keys [tb, tb+16) and [tb+16, tb+32): D[i = key][j = query], so a lane
//   holds 4+4 keys of ONE query -> the online softmax is lane-local plus two cross-lane max shuffles, and the
//   exponentials ARE the B fragment of the second MFMA (O += V^T P) after an fp16 pack: P never leaves registers.
//   (The k-slot order of that MFMA is permuted -- slot (g, e) <-> key tb + 16*(e/4) + 4g + e%4 -- identically for
//   the V^T A-fragment, which is legal because a contraction does not care about the order of k.)
// K rows (zero padded to 32 channels) and V^T live in LDS, staged per 256-key chunk.  fp16 operands, fp32
// accumulate/softmax.  Roofline: MFMA-side FLOPs are tiny (4 T^2 C per row); the kernel is bound by the exp/VALU work
// of the softmax and LDS staging -- DESIGN.md section 4.4.
#pragma once
#include "aero_common.h"

#define AERO_ATTN_KC 256                 /* keys per LDS chunk (streaming form) */
#define AERO_ATTN_KRES 512               /* keys held in LDS by the resident form (T <= 512: every reference config at 2-s clips) */

// RES = false is the only instantiation: the streaming form (one block per 128 queries, keys in chunks of 256), used for rows
// longer than AERO_ATTN_FOLD_T.  (RES = true -- K / V^T of a (row, head) staged once for all query blocks -- was measured
// slower with this instruction-bound score loop (328 vs 290 us); the folded kernel below is what made the resident layout pay.)
template <int DT, bool RES>  // DT = number of 16-row output tiles: head dim <= 16*DT
__global__ __launch_bounds__(512) void aero_attn_kernel(aero_attn_desc d) {
    constexpr int KC = RES ? AERO_ATTN_KRES : AERO_ATTN_KC;
    constexpr int VS = KC + 4;                                   // V^T row pitch (halfs): +8 B breaks the ds_read_b64 bank pattern
    __shared__ AERO_LDS_ALIGN h16 Ks[KC * 32];
    __shared__ AERO_LDS_ALIGN h16 Vt[DT * 16 * VS];
    __shared__ AERO_LDS_ALIGN h16 Qs[128 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int h = blockIdx.y, row = blockIdx.z;
    const int C = d.C, T = d.T;
    const int dh = C / d.heads;
    const int nqb = RES ? (T + 127) / 128 : 1;                   // query blocks this CTA walks over
    const h16* base = (const h16*)d.qkvd + (int64_t)row * T * d.ld;
    // the softmax runs in the log2 domain: log2(e) is folded into the query scale, the decay slope and the self-kill value,
    // so every probability is ONE subtraction and one bare v_exp_f32
    constexpr float L2E = 1.4426950408889634f;
    const float qscale = L2E / sqrtf((float)dh);

    // staging moves 4 channels (8 bytes) per step when the head slices are 8-byte aligned, else scalars
    const bool vec4 = (dh % 4 == 0) && (d.ld % 4 == 0) && (C % 4 == 0) && (((uintptr_t)d.qkvd & 7) == 0);
    const int nc4 = (dh + 3) >> 2;                               // 8-byte pieces per head slice: only those are requested
    float koff[8];                                               // position of key slot e inside a 32-key block (lane constant)
#pragma unroll
    for (int e = 0; e < 8; ++e) koff[e] = (float)(((e >> 2) << 4) + g * 4 + (e & 3));
  for (int qb = 0; qb < nqb; ++qb) {
    const int s_blk = (RES ? qb : (int)blockIdx.x) * 128;
    if (RES && qb > 0) __syncthreads();                          // every wave has taken its query fragment of the previous block
    // ---- queries of this block: [128][32], pre-scaled by 1/sqrt(dh), zero padded
    if (vec4) {
        for (int idx = tid; idx < 128 * 8; idx += 512) {
            const int sl = idx >> 3, c4 = idx & 7;
            const int s = s_blk + sl;
            h16x4 v = (h16x4){0, 0, 0, 0};
            if (s < T && c4 * 4 < dh) {
                const h16x4 x = *(const h16x4*)(base + (int64_t)s * d.ld + h * dh + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (h16)((float)x[e] * qscale);
            }
            *(h16x4*)&Qs[aero_tile_off(sl, c4 >> 1) + (c4 & 1) * 4] = v;
        }
    } else {
        for (int idx = tid; idx < 128 * 32; idx += 512) {
            const int sl = idx >> 5, c = idx & 31;
            const int s = s_blk + sl;
            float v = 0.f;
            if (s < T && c < dh) v = (float)base[(int64_t)s * d.ld + h * dh + c] * qscale;
            Qs[aero_tile_off(sl, c >> 3) + (c & 7)] = (h16)v;
        }
    }
    // this lane's query and its decay slope
    const int s = s_blk + wave * 16 + col;
    float Dq = 0.f;
    if (s < T) {
        const h16* dp = base + (int64_t)s * d.ld + 3 * C + h * d.ndecay;
        for (int f = 0; f < d.ndecay; ++f) Dq += (float)(f + 1) * aero_sigmoid((float)dp[f]);
        Dq *= L2E * 0.5f / sqrtf((float)(d.ndecay > 0 ? d.ndecay : 1));
    }
    const int sw_lo = s_blk + wave * 16, sw_hi = sw_lo + 16;  // this wave's queries
    float m = -1e30f, l = 0.f;
    f32x4 O[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) O[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    h16x8 qf = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};

    for (int kc0 = 0; kc0 < T; kc0 += KC) {
        __syncthreads();                                   // previous chunk fully consumed (and Qs written)
        const int kn = (T - kc0) < KC ? (T - kc0) : KC;
        const int kn32 = (kn + 31) & ~31;
        if (RES && qb > 0) {
            // K and V^T of the row are already resident
        } else if (vec4) {
            // zero fill of the padded channels (once), then only the nc4 live 8-byte pieces of every key are requested:
            // consecutive lanes cover the consecutive pieces of one key (24 contiguous bytes for a head of 12)
            for (int idx = tid; idx < kn32 * 8; idx += 512) {
                const int tl = idx >> 3, c4 = idx & 7;
                if (c4 >= nc4 || tl >= kn) *(h16x4*)&Ks[aero_tile_off(tl, c4 >> 1) + (c4 & 1) * 4] = (h16x4){0, 0, 0, 0};
            }
            for (int idx = tid; idx < kn * nc4; idx += 512) {
                const int tl = idx / nc4, c4 = idx - tl * nc4;
                const h16x4 v = *(const h16x4*)(base + (int64_t)(kc0 + tl) * d.ld + C + h * dh + c4 * 4);
                *(h16x4*)&Ks[aero_tile_off(tl, c4 >> 1) + (c4 & 1) * 4] = v;
            }
            for (int idx = tid; idx < kn32 * DT * 4; idx += 512) {
                const int tl = idx / (DT * 4), d4 = idx - tl * (DT * 4);
                h16x4 v = (h16x4){0, 0, 0, 0};
                if (tl < kn && d4 * 4 < dh) v = *(const h16x4*)(base + (int64_t)(kc0 + tl) * d.ld + 2 * C + h * dh + d4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) Vt[(d4 * 4 + e) * VS + tl] = v[e];
            }
        } else {
            for (int idx = tid; idx < kn32 * 32; idx += 512) {
                const int tl = idx >> 5, c = idx & 31;
                h16 v = (h16)0;
                if (tl < kn && c < dh) v = base[(int64_t)(kc0 + tl) * d.ld + C + h * dh + c];
                Ks[aero_tile_off(tl, c >> 3) + (c & 7)] = v;
            }
            for (int idx = tid; idx < kn32 * DT * 16; idx += 512) {
                const int tl = idx / (DT * 16), dd = idx - tl * (DT * 16);
                h16 v = (h16)0;
                if (tl < kn && dd < dh) v = base[(int64_t)(kc0 + tl) * d.ld + 2 * C + h * dh + dd];
                Vt[dd * VS + tl] = v;
            }
        }
        __syncthreads();
        if (kc0 == 0) qf = *(const h16x8*)&Qs[aero_tile_off(wave * 16 + col, g)];
        (void)nqb;
        // Two 32-key blocks per trip: their score -> softmax -> PV chains are independent until the shared running maximum,
        // so one block's MFMA / exp latencies hide behind the other's, and the accumulator is rescaled once per 64 keys.
        auto scores = [&](int tb, float* sc) {
            const h16x8 k0 = *(const h16x8*)&Ks[aero_tile_off(tb + col, g)];
            const h16x8 k1 = *(const h16x8*)&Ks[aero_tile_off(tb + 16 + col, g)];
            const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, qf, z4, 0, 0, 0);
            const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf, z4, 0, 0, 0);
            // key e of this lane sits at t = kc0 + tb + koff[e]; distance to the query = |u + koff[e]| with u = kc0 + tb - s
            const int t_blk = kc0 + tb;
            const float u = (float)(t_blk - s);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = u + koff[e];
                sc[e] = ((e < 4) ? s0[e & 3] : s1[e & 3]) - fabsf(x) * Dq;
            }
            // the self reference (modules.py:120) and the keys beyond T only exist in blocks that overlap the wave's own
            // queries / the end of the row: block-uniform tests keep them out of the common path
            if (t_blk < sw_hi && t_blk + 32 > sw_lo) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (u + koff[e] == 0.f) sc[e] = -100.f * L2E;
            }
            if (t_blk + 32 > T) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (t_blk + (int)koff[e] >= T) sc[e] = -1e30f;
            }
        };
        auto max8 = [](const float* sc) {
            return fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
        };
        auto pv = [&](int tb, const h16x8& pf) {
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const h16* vr = &Vt[(i * 16 + col) * VS + tb + g * 4];
                const h16x4 va = *(const h16x4*)vr;
                const h16x4 vb = *(const h16x4*)(vr + 16);
                const h16x8 vf = (h16x8){va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                O[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, O[i], 0, 0, 0);
            }
        };
        for (int tb = 0; tb < kn32; tb += 64) {
            const bool two = tb + 32 < kn32;                  // block-uniform
            float sa[8], sb[8];
            scores(tb, sa);
            if (two) {
                scores(tb + 32, sb);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) sb[e] = -1e30f;
            }
            float cmax = fmaxf(max8(sa), max8(sb));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            const float mn = fmaxf(m, cmax);
            const float alpha = aero_exp2(m - mn);
            m = mn;
            float psum = 0.f;
            h16x8 pfa, pfb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pa = aero_exp2(sa[e] - mn), pb = aero_exp2(sb[e] - mn);
                psum += pa + pb;
                pfa[e] = (h16)pa;
                pfb[e] = (h16)pb;
            }
            l = l * alpha + psum;
#pragma unroll
            for (int i = 0; i < DT; ++i) O[i] = O[i] * alpha;
            pv(tb, pfa);
            if (two) pv(tb + 32, pfb);
        }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (s < T) {
        const float inv = 1.0f / l;
        h16* op = (h16*)d.out + ((int64_t)row * T + s) * C + h * dh;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dd = i * 16 + g * 4 + r;
                if (dd < dh) op[dd] = (h16)(O[i][r] * inv);
            }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Folded form (T <= 512: every reference config at 2-s clips).  PMC of the kernel above: ~12.7 vector instructions per
// score -- the distance bias (add + fma), the running maximum, the rescale, exp, the row sum and the fp16 pack -- and the
// kernel is bound by exactly that instruction stream.  Here the per-score work outside the MFMAs is ONE exp2 and half a
// pack:
//   * keys and V^T of the (row, head) are resident in LDS, so the scores can be formed twice: pass 1 only takes the
//     maximum per query (one v_max3 per two scores), pass 2 computes the probabilities against that FIXED maximum -- no
//     running maximum, no accumulator rescale;
//   * the head dimension (12 or 24) leaves k-slots of the 16x16x32 score MFMA unused; four of them carry the distance
//     bias and the maximum: key side (t, t, 1, 1), query side (+-D_hi, +-D_lo, c_hi, c_lo) with D = D_hi + D_lo the decay
//     slope split into two fp16 (products are exact in the fp32 accumulator) and c = -+D*s - m, so the MFMA returns
//     S - |t-s|*D - m directly for every key block that lies entirely before or entirely after the wave's 16 queries
//     (the sign of t - s is then fixed); only the <= 2 blocks around the diagonal and the ragged last block take the
//     general path (|t-s|, self reference -100, keys >= T);
//   * V^T carries a row of ones below the head's channels: the softmax denominator falls out of the O += V^T P MFMA;
//   * a fifth slot (key side -30000 for the padded keys of the last block, query side 1) removes the ragged-end test;
//   * one block stages K / V^T once and walks over all query blocks of its (row, head): with the score loop this lean the
//     staging and the per-block setup were 60 % of the instructions (PMC: 1723 vector instructions per wave, ~370 of them
//     in the two score loops).
#define AERO_ATTN_FOLD_T 512
template <int DT>
__global__ __launch_bounds__(512) void aero_attn_fold_kernel(aero_attn_desc d) {
    constexpr int KC = AERO_ATTN_FOLD_T;
    constexpr int VS = KC + 4;
    __shared__ AERO_LDS_ALIGN h16 Ks[KC * 32];
    __shared__ AERO_LDS_ALIGN h16 Vt[DT * 16 * VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    // (row, head) of this block.  Blocks are handed to the 8 XCDs round-robin by linear id, and the heads of one row read the SAME
    // qkvd lines (80 useful bytes out of every 320-byte row each): with the natural order they land on different XCDs and every
    // private L2 fetches those lines again (PMC r02: 476 MB fetched for 107 MB).  Remap so that the heads of a row share an XCD.
    int h = blockIdx.y, row = blockIdx.z;
    if ((d.R & 7) == 0) {
        const int lin = (int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z;
        const int xcd = lin & 7, j = lin >> 3;
        h = j % d.heads;
        row = (j / d.heads) * 8 + xcd;
    }
    const int C = d.C, T = d.T;
    const int dh = C / d.heads;
    const int pos = (dh + 3) & ~3;                               // first of the four bias k-slots
    const int gx = pos >> 3, e0 = pos & 7;                       // lane group / element offset (0 or 4) that holds them
    const h16* base = (const h16*)d.qkvd + (int64_t)row * T * d.ld;
    constexpr float L2E = 1.4426950408889634f;
    const float qscale = L2E / sqrtf((float)dh);
    const int kn32 = (T + 31) & ~31;
    const int nqb = (T + 255) / 256;                             // the block walks over all query blocks (256 queries: 8 waves x 32) of its (row, head)
    // ---- staging: queries (scaled), keys + bias slots, V^T + the row of ones
    const bool vec4 = (dh % 4 == 0) && (d.ld % 4 == 0) && (C % 4 == 0) && (((uintptr_t)d.qkvd & 7) == 0);
    const int nc4 = (dh + 3) >> 2;                               // 8-byte pieces per head slice
    if (vec4) {
        // keys: piece nc4 of every key is the bias quadruple (t, t, 1, 1), the pieces above it zero (no loads in this loop) ...
        for (int idx = tid; idx < kn32 * 8; idx += 512) {
            const int tl = idx >> 3, c4 = idx & 7;
            if (c4 >= nc4 || tl >= T) {
                h16x4 v = (h16x4){0, 0, 0, 0};
                if (tl < T && c4 == nc4) v = (h16x4){(h16)(float)tl, (h16)(float)tl, (h16)1.f, (h16)1.f};
                if (tl >= T && c4 == nc4 + 1) v[0] = (h16)-30000.f;                  // padded key: -30000 * (query slot = 1)
                *(h16x4*)&Ks[aero_tile_off(tl, c4 >> 1) + (c4 & 1) * 4] = v;
            }
        }
        // ... the live 8-byte pieces of K and V come from HBM, FOUR loads in flight per thread before the first LDS store
        // (one load per trip made the staging of a block eight serialized HBM round trips: the kernel ran at half speed)
        const int nk = T * nc4;
        for (int i0 = tid; i0 < nk; i0 += 4 * 512) {
            h16x4 kv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * 512;
                if (idx < nk) {
                    const int tl = idx / nc4, c4 = idx - tl * nc4;
                    kv[u] = *(const h16x4*)(base + (int64_t)tl * d.ld + C + h * dh + c4 * 4);
                    vv[u] = *(const h16x4*)(base + (int64_t)tl * d.ld + 2 * C + h * dh + c4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * 512;
                if (idx < nk) {
                    const int tl = idx / nc4, c4 = idx - tl * nc4;
                    *(h16x4*)&Ks[aero_tile_off(tl, c4 >> 1) + (c4 & 1) * 4] = kv[u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Vt[(c4 * 4 + e) * VS + tl] = vv[u][e];
                }
            }
        }
        // V^T rows from dh up: the row of ones, then zeros; padded keys zero in every row
        for (int idx = tid; idx < (DT * 16 - dh) * (kn32 >> 2); idx += 512) {
            const int rr = idx / (kn32 >> 2), t4 = (idx - rr * (kn32 >> 2)) * 4;
            h16x4 v = (h16x4){0, 0, 0, 0};
            if (rr == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (t4 + e < T) ? (h16)1.f : (h16)0;
            }
            *(h16x4*)&Vt[(dh + rr) * VS + t4] = v;
        }
        for (int idx = tid; idx < dh * (kn32 - T); idx += 512) {
            const int rr = idx / (kn32 - T), tl = T + idx - rr * (kn32 - T);
            Vt[rr * VS + tl] = (h16)0;
        }
    } else {
        for (int idx = tid; idx < kn32 * 32; idx += 512) {
            const int tl = idx >> 5, c = idx & 31;
            h16 v = (h16)0;
            if (tl < T) {
                if (c < dh) v = base[(int64_t)tl * d.ld + C + h * dh + c];
                else if (c == pos || c == pos + 1) v = (h16)(float)tl;               // (integers <= 2048 are exact in fp16)
                else if (c == pos + 2 || c == pos + 3) v = (h16)1.f;
            } else if (c == pos + 4) {
                v = (h16)-30000.f;
            }
            Ks[aero_tile_off(tl, c >> 3) + (c & 7)] = v;
        }
        for (int idx = tid; idx < kn32 * DT * 16; idx += 512) {
            const int tl = idx / (DT * 16), dd = idx - tl * (DT * 16);
            h16 v = (h16)0;
            if (tl < T) {
                if (dd < dh) v = base[(int64_t)tl * d.ld + 2 * C + h * dh + dd];
                else if (dd == dh) v = (h16)1.f;
            }
            Vt[dd * VS + tl] = v;
        }
    }
    __syncthreads();                                             // K and V^T of the (row, head) are in
    float koff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) koff[e] = (float)(((e >> 2) << 4) + g * 4 + (e & 3));
    constexpr int NQ = 2;                                        // query fragments per wave: K / V^T fragments are fetched once for both
  for (int qb = 0; qb < nqb; ++qb) {
    const int sw_lo = qb * 256 + wave * 32;                      // this wave's 32 queries (two fragments of 16): a multiple of 32
    if (sw_lo >= T) continue;                                    // (wave-uniform; nothing below synchronises the block)
    // this lane's queries (fragment j: sw_lo + 16j + col), straight from global memory into MFMA B fragments (dims g*8 .. +7,
    // pre-scaled), and their decay slopes
    int sq[NQ];
    float Dq[NQ];
    h16x8 qf[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        sq[j] = sw_lo + j * 16 + col;
        Dq[j] = 0.f;
        qf[j] = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (sq[j] < T) {
            const h16* qp = base + (int64_t)sq[j] * d.ld + h * dh;
            if (vec4) {
#pragma unroll
                for (int hv = 0; hv < 2; ++hv) {
                    const int c = g * 8 + hv * 4;
                    if (c < dh) {
                        const h16x4 x = *(const h16x4*)(qp + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) qf[j][hv * 4 + e] = (h16)((float)x[e] * qscale);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (g * 8 + e < dh) qf[j][e] = (h16)((float)qp[g * 8 + e] * qscale);
            }
            const h16* dp = base + (int64_t)sq[j] * d.ld + 3 * C + h * d.ndecay;
            for (int f = 0; f < d.ndecay; ++f) Dq[j] += (float)(f + 1) * aero_sigmoid((float)dp[f]);
            Dq[j] *= L2E * 0.5f / sqrtf((float)(d.ndecay > 0 ? d.ndecay : 1));
        }
    }
    // query fragment with the bias slots filled in: sign = +1 for key blocks before the queries, -1 after; cc = c
    auto with_bias = [&](int j, float sign, float cc) {
        h16x8 q = qf[j];
        if (g == gx) {
            const float Ds = sign * Dq[j];
            const h16 d_hi = (h16)Ds, d_lo = (h16)(Ds - (float)d_hi);
            const h16 c_hi = (h16)cc, c_lo = (h16)(cc - (float)c_hi);
            if (e0 == 0) { q[0] = d_hi; q[1] = d_lo; q[2] = c_hi; q[3] = c_lo; }
            else { q[4] = d_hi; q[5] = d_lo; q[6] = c_hi; q[7] = c_lo; }
        }
        if (g == ((pos + 4) >> 3)) {                             // slot pos+4: 1 against the -30000 of the padded keys
            if (e0 == 0) q[4] = (h16)1.f;
            else q[0] = (h16)1.f;
        }
        return q;
    };
    // scores of one 32-key block for both fragments, K fragments fetched once.  GENERAL = the block on the diagonal (and every
    // block when the folded constants would leave fp16): |t - s|, the self reference and the ragged end are applied per score
    auto block = [&](int tb, const h16x8 (&q)[NQ], bool general, float (&sc)[NQ][8]) {
        const h16x8 k0 = *(const h16x8*)&Ks[aero_tile_off(tb + col, g)];
        const h16x8 k1 = *(const h16x8*)&Ks[aero_tile_off(tb + 16 + col, g)];
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, q[j], z4, 0, 0, 0);
            const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, q[j], z4, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[j][e] = s0[e]; sc[j][4 + e] = s1[e]; }
        }
        if (general) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float u = (float)(tb - sq[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = u + koff[e];
                    sc[j][e] -= fabsf(x) * Dq[j];
                    if (x == 0.f) sc[j][e] = -100.f * L2E;                               // self reference (modules.py:120)
                    if (tb + (int)koff[e] >= T) sc[j][e] = -1e30f;
                }
            }
        }
    };
    auto max8 = [](const float* sc) {
        return fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
    };
    // The key range splits at the 32-key block that holds the wave's 32 queries: blocks before it run with the "before" query
    // fragments, the diagonal block takes the general path, blocks after it the "after" fragments -- three straight loops.
    const int tdiag = sw_lo;

    // ---- pass 1: the maximum of every query's scores
    float m[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) m[j] = -1e30f;
    {
        h16x8 qbv[NQ], qav[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) { qbv[j] = with_bias(j, 1.f, -Dq[j] * (float)sq[j]); qav[j] = with_bias(j, -1.f, Dq[j] * (float)sq[j]); }
        auto run = [&](int lo, int hi, const h16x8 (&q)[NQ]) {
            for (int tb = lo; tb < hi; tb += 32) {
                float sc[NQ][8];
                block(tb, q, false, sc);
#pragma unroll
                for (int j = 0; j < NQ; ++j) m[j] = fmaxf(m[j], max8(sc[j]));
            }
        };
        run(0, tdiag < kn32 ? tdiag : kn32, qbv);
        if (tdiag < kn32) {
            float sc[NQ][8];
            block(tdiag, qf, true, sc);
#pragma unroll
            for (int j = 0; j < NQ; ++j) m[j] = fmaxf(m[j], max8(sc[j]));
            run(tdiag + 32, kn32, qav);
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            m[j] = fmaxf(m[j], __shfl_xor(m[j], 16));
            m[j] = fmaxf(m[j], __shfl_xor(m[j], 32));
        }
    }
    // the folded constants must stay inside fp16 (c = -+D*s - m); otherwise every block takes the general path
    const bool fold_ok = !aero_wave_any(!(fabsf(m[0]) < 2.0e4f && fabsf(m[1]) < 2.0e4f));

    // ---- pass 2: probabilities against the fixed maximum, O += V^T P (row dh of V^T = ones: the denominator)
    f32x4 O[NQ][DT];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int i = 0; i < DT; ++i) O[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        h16x8 qbv[NQ], qav[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) { qbv[j] = with_bias(j, 1.f, -Dq[j] * (float)sq[j] - m[j]); qav[j] = with_bias(j, -1.f, Dq[j] * (float)sq[j] - m[j]); }
        auto step = [&](int tb, const h16x8 (&q)[NQ], bool general) {
            float sc[NQ][8];
            block(tb, q, general, sc);
            h16x8 pf[NQ];
#pragma unroll
            for (int j = 0; j < NQ; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[j][e] = (h16)aero_exp2(general ? sc[j][e] - m[j] : sc[j][e]);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const h16* vr = &Vt[(i * 16 + col) * VS + tb + g * 4];
                const h16x4 va = *(const h16x4*)vr;
                const h16x4 vb = *(const h16x4*)(vr + 16);
                const h16x8 vf = (h16x8){va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
                for (int j = 0; j < NQ; ++j) O[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[j], O[j][i], 0, 0, 0);
            }
        };
        if (fold_ok) {
            for (int tb = 0; tb < (tdiag < kn32 ? tdiag : kn32); tb += 32) step(tb, qbv, false);
            if (tdiag < kn32) {
                step(tdiag, qf, true);
                for (int tb = tdiag + 32; tb < kn32; tb += 32) step(tb, qav, false);
            }
        } else {
            for (int tb = 0; tb < kn32; tb += 32) step(tb, qf, true);
        }
    }
    // the denominator: row dh of O, held by lane group (dh % 16) / 4 in register dh % 4 of tile dh / 16
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int li = dh >> 4, lr = dh & 3, lg = (dh & 15) >> 2;
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (i == li && r == lr) mine = O[j][i][r];
        const float lsum = __shfl(mine, lg * 16 + col);
        if (sq[j] < T) {
            const float inv = 1.0f / lsum;
            h16* op = (h16*)d.out + ((int64_t)row * T + sq[j]) * C + h * dh;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const int dd = i * 16 + g * 4;                   // this lane's four consecutive output channels
                if (vec4 && (((uintptr_t)d.out) & 7) == 0) {     // (dh % 4 == 0: all four real or all four padding)
                    if (dd < dh) *(h16x4*)(op + dd) = (h16x4){(h16)(O[j][i][0] * inv), (h16)(O[j][i][1] * inv), (h16)(O[j][i][2] * inv), (h16)(O[j][i][3] * inv)};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (dd + r < dh) op[dd + r] = (h16)(O[j][i][r] * inv);
                }
            }
        }
    }
  }
}

static int aero_attn_launch(const aero_attn_desc* d, hipStream_t stream, const char** err) {
    if (!d || !d->qkvd || !d->out) { *err = "localstate: null pointer"; return AERO_ERR_ARG; }
    if (d->R < 1 || d->T < 1 || d->C < 1 || d->heads < 1 || d->C % d->heads || d->ndecay < 0) { *err = "localstate: bad geometry"; return AERO_ERR_ARG; }
    if (d->ld < 3 * d->C + d->heads * d->ndecay) { *err = "localstate: ld too small"; return AERO_ERR_ARG; }
    if (d->R > 65535 || d->heads > 65535) { *err = "localstate: too many rows/heads for one launch"; return AERO_ERR_ARG; }
    const int dh = d->C / d->heads;
    if (dh > 32) { *err = "localstate: head dim > 32 unsupported"; return AERO_ERR_UNSUPPORTED; }
    dim3 block(512);
    // AERO_ATTN_FOLD=0: the streaming kernel also for short rows (A/B)
    static int fold = -1;
    if (fold < 0) { const char* e = getenv("AERO_ATTN_FOLD"); fold = (e && e[0] == '0') ? 0 : 1; }
    if (fold && d->T <= AERO_ATTN_FOLD_T && ((dh + 3) & ~3) + 5 <= 32 && dh + 1 <= 32) {
        dim3 grid(1, (unsigned)d->heads, (unsigned)d->R);
        if (dh + 1 <= 16) AERO_LAUNCH((aero_attn_fold_kernel<1>), grid, block, stream, *d);
        else AERO_LAUNCH((aero_attn_fold_kernel<2>), grid, block, stream, *d);
        return AERO_OK;
    }
    dim3 grid((unsigned)((d->T + 127) / 128), (unsigned)d->heads, (unsigned)d->R);
    if (dh <= 16) AERO_LAUNCH((aero_attn_kernel<1, false>), grid, block, stream, *d);
    else AERO_LAUNCH((aero_attn_kernel<2, false>), grid, block, stream, *d);
    return AERO_OK;
}
