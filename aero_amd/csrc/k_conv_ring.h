// k_conv_ring.h -- the wide contractions of the path (decoder 3x3 "rewrite" convs, aero.py:179: 68 % of the FLOPs) as a
// software-pipelined implicit GEMM on v_mfma_f32_32x32x16_f16.
//
// Same GEMM view as k_conv.h (one (b, fo) row per block; m = output channel, n = time step, k = (tap, channel)), built
// around what PMC and ablation builds showed the 2-stage / one-__syncthreads-per-chunk kernels wait for:
//   * EIGHT waves per block, ONE block per CU (256 registers per wave): wave tile (NRB*32) x 64, i.e. 128 x 64 for the
//     256-row tiles -- 0.75 LDS operand reads per MFMA-FLOP of the 64 x 64 wave tile;
//   * RINGS of LDS slots filled by `global_load_lds_dwordx4` three chunks ahead; copies stay in flight across barriers:
//     ONE counted `s_waitcnt vmcnt(N)` per K-chunk (never 0 in the loop) and raw `s_barrier`s that wait for LDS only;
//   * operand fragments of phase g+1 are read from LDS while the MFMAs of phase g run, so neither the LDS round trip nor
//     the copy latency sits between two MFMA groups; a phase = 8 x v_mfma_f32_32x32x16_f16 per wave, one barrier per phase;
//   * a branch-free chunk iterator (scalar selects) so that a phase is straight-line code and its bookkeeping is
//     interleaved with the MFMAs (`sched_group_barrier`) instead of running in lockstep in front of them;
//   * the L1/TA request path is what bounds the loop once the above is in place (ablation: same loop, every copy reading
//     one line -> 1.4x faster; TCP busy 80 %).  Hence (i) weights come from a PRE-TILED image (`weight_tiled`: a copy
//     instruction reads 1 KiB of consecutive bytes), and (ii) with NT = 3 unit-stride time taps the activation SLAB
//     [t0-1, t0+BN+1) x 32 channels is copied ONCE per (frequency tap, channel chunk) and read at row offsets 0/1/2 by
//     the three taps: a third of the activation requests.
// Tiles <WM,WN,NRB>: <2,4,4> 256 rows x 256 steps (<2,2,4> 256 x 128 on four waves when T leaves the last 256-step tile half empty); <2,2,3> 192 x 128 on four waves (two blocks per CU; round 5) and <2,4,3> 192 x 256
// (one 12-MFMA phase per chunk); <1,8,4> 128 rows x
// 512 steps (a whole T = 501 row per block); <1,8,2> 64 rows x 512 steps.
// Ordering rules the schedule relies on (LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt followed by a
// barrier the reader has passed): the wait for chunk j sits before the barrier that ENDS the phase preceding the phase in
// which chunk j is first read; a slot is refilled only after the barrier ending the phase of its last read.
// Roofline: MFMA.  Epilogue: + bias, NONE / ReLU / GELU / GLU, fp16 channels-last through an LDS transpose.
#pragma once
#include "k_conv_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NT = time taps that share one activation slab (3: unit-stride 3-tap grids; 1: every tap copies its own tile)
template <int WM, int WN, int NRB, int NT>
struct AeroRingGeom {
    static constexpr int BM = WM * NRB * 32, BN = WN * 64;
    static constexpr int NW = WM * WN;                                            // waves per block: 8, or 4 for the half-height tile <1,4,3>
    static constexpr int RBP = NRB == 3 ? 3 : 2;                                  // row blocks per phase: 4 MFMAs each
    static constexpr int NPH = NRB / RBP;                                         // phases per K-chunk
    // A ring: one [BM][32] tile per slot.  THREE slots for the 256-row tile on four waves (<2,2,4>: 76.8 KiB, two blocks per CU), its weight
    // tile issued two chunks ahead instead of three (a chunk of that block is 16 MFMAs a wave: ~0.9 us with two blocks on the CU)
    static constexpr int NSA = (NW == 4 && NRB == 4) ? 3 : 4;
    static constexpr int AHEAD = NPH == 1 ? 4 : NSA - 1;                          // chunks between an A copy and its use
    static constexpr int SLABI = BN / 16 + (NT > 1 ? 1 : 0);                      // copy instructions per activation slab
    static constexpr int SLABR = SLABI * 16;                                      // slab rows held in LDS
    static constexpr int LB = NT > 1 ? 2 : 3;                                     // groups between a slab copy and its use
    static constexpr int NSB = LB + 1;
    static constexpr int A_SLOT = BM * 32, B_SLOT = SLABR * 32;                   // h16 elements
    static constexpr int NA = (BM / 16 + NW - 1) / NW;                            // A copies per wave and chunk
    static constexpr int NBG = (SLABI + NW - 1) / NW;                             // slab copies per wave and group
    static constexpr int CS = BM + 8;                                             // epilogue staging row (h16), padded
    static constexpr int EPI = WN * 32 * CS;
    static constexpr int RING = NSA * A_SLOT + NSB * B_SLOT;
    static constexpr int SMEM = RING > EPI ? RING : EPI;                          // h16 elements
    // slab copies issued with chunk jt of a group (they are spread over the group's NT chunks)
    static constexpr int nb(int jt) { return (NBG + NT - 1 - jt) / NT; }
    static constexpr int nb_before(int jt) { int n = 0; for (int j = 0; j < jt; ++j) n += nb(j); return n; }
    // copies in flight that the wait guarding the NEXT chunk's operands may leave outstanding.  Issue order inside a chunk:
    // slab pieces (phase 0), then A pieces (phase 1 -- phase 0 when a chunk has one phase).  Two phases: the wait sits at
    // the end of phase 0 of chunk k and guards chunk k+1, whose A went out in phase 1 of chunk k-2; after it: chunk k-1
    // (slab + A) and the slab pieces of chunk k.  One phase: it guards chunk k+2 (A issued in chunk k-2): chunks k-1, k.
    static constexpr int vmw(int jt) {
        const int jp = (jt + NT - 1) % NT;
        // (AHEAD == 2, two phases: chunk k+1's A went out in phase 1 of chunk k-1; after it only the slab pieces of chunk k)
        return NPH == 1 ? 2 * NA + nb(jp) + nb(jt) : (AHEAD == 2 ? nb(jt) : NA + nb(jp) + nb(jt));
    }
};

// SM = 1: GroupNorm statistics (sum, sum of squares of conv + bias, per (item, group)) ride along in fp32 per lane, are
// folded over the wave's row blocks of one group, reduced over the wave in fp64 and added with one fp64 atomic pair per
// (wave, group): the separate read-only statistics pass over the stored tensor (80 us for the first decoder's 394 MB)
// disappears.  Needs 32-row aligned groups (a row block never straddles two groups) and no activation.
template <int WM, int WN, int NRB, int ACT, int SM = 0>
static __device__ __forceinline__ void aero_ring_epilogue(const AeroConvK& p, f32x16 (&acc)[NRB][2], h16* Cs, int b, int fo, int m0, int t0) {
    typedef AeroRingGeom<WM, WN, NRB, 1> G;
    constexpr bool GLU = ACT == AERO_ACT_GLU;
    constexpr int BMo = GLU ? G::BM / 2 : G::BM;
    constexpr int NVEC = BMo / 8;
    constexpr int NPOS = WN * 32;
    constexpr int NTH = G::NW * 64;
    constexpr int NIT = (NPOS * NVEC + NTH - 1) / NTH;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int T = d.T, M = d.M;
    const int Mout = GLU ? (M >> 1) : M;
    const int m0o = GLU ? (m0 >> 1) : m0;
    h16* drow = (h16*)d.dst + (int64_t)b * d.d_b + (int64_t)fo * d.d_f + m0o;
    const int hi4 = (lane >> 5) * 4;
    float st1[NRB], st2[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) st1[rb] = st2[rb] = 0.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int pc = wn * 32 + (lane & 31);
        const bool tin = t0 + wn * 64 + cb * 32 + (lane & 31) < T;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ml = (wm * NRB + rb) * 32 + 8 * j + hi4;               // row of acc regs 4j..4j+3 inside the block
                float o[4];
                if (d.bias) {
                    const f32x4 bv = *(const f32x4*)(d.bias + m0 + ml);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[rb][cb][4 * j + r] + bv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[rb][cb][4 * j + r];
                }
                if constexpr (SM == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = tin ? o[r] : 0.f;
                        st1[rb] += v;
                        st2[rb] = fmaf(v, v, st2[rb]);
                    }
                }
                if constexpr (GLU) {
                    const float g0 = o[0] * aero_sigmoid(o[1]);
                    const float g1 = o[2] * aero_sigmoid(o[3]);
                    *(h16x2*)&Cs[pc * G::CS + (ml >> 1)] = (h16x2){(h16)g0, (h16)g1};
                } else {
                    if constexpr (ACT == AERO_ACT_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
                    } else if constexpr (ACT == AERO_ACT_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = aero_gelu(o[r]);
                    }
                    *(h16x4*)&Cs[pc * G::CS + ml] = (h16x4){(h16)o[0], (h16)o[1], (h16)o[2], (h16)o[3]};
                }
            }
        }
        aero_lds_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NTH;
            const int pos = idx / NVEC, cv = idx - pos * NVEC;
            const int t = t0 + (pos >> 5) * 64 + cb * 32 + (pos & 31);
            if (idx < NPOS * NVEC && t < T && m0o + cv * 8 < Mout)
                *(h16x8*)(drow + (int64_t)t * d.d_t + cv * 8) = *(const h16x8*)&Cs[pos * G::CS + cv * 8];
        }
        aero_lds_barrier();
    }
    if constexpr (SM == 1) {
        const int gs = M / d.stat_G;                              // rows per group: a multiple of 32 (host-checked)
        const int64_t sitem = (int64_t)(d.stat_per_row ? b * d.Fout + fo : b) * d.stat_G;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int base = m0 + (wm * NRB + rb) * 32;
            const int grp = base / gs;
            if (rb + 1 < NRB && (base + 32) / gs == grp) {        // the next row block is in the same group: fold
                st1[rb + 1] += st1[rb];
                st2[rb + 1] += st2[rb];
                continue;
            }
            const double a = aero_wave_sum((double)st1[rb]);
            const double c = aero_wave_sum((double)st2[rb]);
            if (lane == 0) {
                atomicAdd(d.stats + (sitem + grp) * 2, a);
                atomicAdd(d.stats + (sitem + grp) * 2 + 1, c);
            }
        }
    }
}

// Fused transposed-conv tail (aero_conv_desc.tail_w, round 5): the last decoder layer's ConvTranspose2d(C -> 2, [8,1] / [4,1]) applied to the
// GLU'd tile while it sits in the staging area.  The 16 x C tail matrix (rows 2 * tap + out channel) is the A operand of
// v_mfma_f32_16x16x32_f16 (three k-steps for C = 96, fragments fetched once per block), the B operand "8 consecutive channels of one time
// step" is one ds_read_b128 of the staging row the epilogue has just written -- the row stride of 200 halves puts the 16 lanes of a read
// group on 16 different bank quadruples --, and a lane ends up with four consecutive tail rows of one time step: one 16-byte store into
// tail_lo (rows 0..7: taps 0..3) or tail_hi (rows 8..15).  The [B][F][T][C] activation (394 MB at the bench size) is never written or read
// back, and the stand-alone transposed conv over it (aero_convtr_carry_kernel: 102 us) is replaced by aero_convtr_tail_finish_kernel
// (two 32-byte reads and four 8-byte writes per (row, time step)).
template <int WM, int WN, int NRB>
static __device__ __forceinline__ void aero_ring_tail_epilogue(const AeroConvK& p, f32x16 (&acc)[NRB][2], h16* Cs, int b, int fo, int m0, int t0) {
    typedef AeroRingGeom<WM, WN, NRB, 1> G;
    constexpr int BMo = G::BM / 2;                                // channels after GLU: the whole layer (one M-tile)
    constexpr int KS = BMo / 32;
    static_assert(BMo % 32 == 0 && G::NW * 16 == WN * 32, "eight waves x 16 time steps cover one half of the tile's columns");
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int T = d.T;
    const int kq = lane >> 4, col = lane & 15;
    h16x8 wa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wa[ks] = *(const h16x8*)((const h16*)d.tail_w + col * d.tail_cp + ks * 32 + kq * 8);
    const int hi4 = (lane >> 5) * 4;
    float* tdst = (kq < 2 ? d.tail_lo : d.tail_hi) + ((int64_t)b * d.Fout + fo) * (int64_t)T * 8 + (kq & 1) * 4;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int pc = wn * 32 + (lane & 31);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ml = (wm * NRB + rb) * 32 + 8 * j + hi4;
                float o[4];
                if (d.bias) {
                    const f32x4 bv = *(const f32x4*)(d.bias + m0 + ml);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[rb][cb][4 * j + r] + bv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[rb][cb][4 * j + r];
                }
                const float g0 = o[0] * aero_sigmoid(o[1]);
                const float g1 = o[2] * aero_sigmoid(o[3]);
                *(h16x2*)&Cs[pc * G::CS + (ml >> 1)] = (h16x2){(h16)g0, (h16)g1};
            }
        }
        aero_lds_barrier();
        {
            const int pp = wave * 16 + col;                        // staging row = time step (pp >> 5) * 64 + cb * 32 + (pp & 31) of the tile
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const h16x8 bf = *(const h16x8*)&Cs[pp * G::CS + ks * 32 + kq * 8];
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[ks], bf, o, 0, 0, 0);
            }
            const int t = t0 + (pp >> 5) * 64 + cb * 32 + (pp & 31);
            if (t < T) *(f32x4*)(tdst + (int64_t)t * 8) = o;
        }
        aero_lds_barrier();
    }
}

// out[b][fo][t][co] = (lo[b][q][t][2k + co] + hi[b][q - 1][t][2k + co] + bias[co]) * scale[b] + shift[b],  q = (fo + pad) / 4, k = (fo + pad) % 4
__global__ __launch_bounds__(256) void aero_convtr_tail_finish_kernel(const float* lo, const float* hi, const float* bias, const float* scale,
                                                                      const float* shift, float* dst, int Fin, int T, int dstF, int pad, int P, int toff) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int q = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0, h0 = a0, h1 = a0;
    if (q < Fin) {
        const float* pl = lo + (((int64_t)b * Fin + q) * T + t) * 8;
        a0 = *(const f32x4*)pl;
        a1 = *(const f32x4*)(pl + 4);
    }
    if (q >= 1) {
        const float* ph = hi + (((int64_t)b * Fin + (q - 1)) * T + t) * 8;
        h0 = *(const f32x4*)ph;
        h1 = *(const f32x4*)(ph + 4);
    }
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f;
    const float sc = scale ? scale[b] : 1.f, sh = shift ? shift[b] : 0.f;
    const float v[8] = {a0[0] + h0[0], a0[1] + h0[1], a0[2] + h0[2], a0[3] + h0[3], a1[0] + h1[0], a1[1] + h1[1], a1[2] + h1[2], a1[3] + h1[3]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int fo = 4 * q + k - pad;
        if (fo < 0 || fo >= dstF) continue;
        *(f32x2*)(dst + (((int64_t)b * dstF + fo) * P + toff + t) * 2) = (f32x2){fmaf(v[2 * k] + b0, sc, sh), fmaf(v[2 * k + 1] + b1, sc, sh)};
    }
}

static int aero_convtr_tail_finish_launch(const float* lo, const float* hi, const float* bias, const float* scale, const float* shift, float* dst,
                                          int B, int Fin, int T, int dstF, int pad, hipStream_t stream, const char** err, int pitch = 0, int toff = 0) {
    if (!lo || !hi || !dst || B < 1 || Fin < 1 || T < 1 || dstF < 1 || pad < 0 || pad > 3) { *err = "convtr_tail_finish: bad arguments"; return AERO_ERR_ARG; }
    if (pitch == 0) pitch = T;
    if (toff < 0 || pitch < toff + T) { *err = "convtr_tail_finish: pitch < toff + T"; return AERO_ERR_ARG; }
    if (((uintptr_t)lo & 15) || ((uintptr_t)hi & 15) || ((uintptr_t)dst & 7) || B > 65535 || Fin + 1 > 65535) { *err = "convtr_tail_finish: alignment / size"; return AERO_ERR_ARG; }
    AERO_LAUNCH(aero_convtr_tail_finish_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)(Fin + 1), (unsigned)B), dim3(256), stream, lo, hi, bias, scale, shift,
                dst, Fin, T, dstF, pad, pitch, toff);
    return AERO_OK;
}

// ABL: ablation bits for profiling builds (results are WRONG with any bit set): 1 no barriers in the K loop, 2 no copies
// in the loop, 4 no fragment reads in the loop, 8 no interleave hints, 32 no counted vmcnt wait, 64 every copy reads the
// zero page, 128 no K loop at all (prologue + epilogue only)
template <int WM, int WN, int NRB, int NT, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void aero_conv_ring_kernel(AeroConvK p) {
    typedef AeroRingGeom<WM, WN, NRB, NT> G;
    constexpr int BM = G::BM, BN = G::BN, NPH = G::NPH, NA = G::NA, NBG = G::NBG, NSA = G::NSA, NSB = G::NSB, RBP = G::RBP;
    constexpr int KC = 32;
    h16* smem = (h16*)AERO_DYN_SMEM;
    h16* smemB = smem + NSA * G::A_SLOT;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // block order: the M-tile is the SLOWEST index: an XCD works on one or two weight tiles (L2 resident) for its whole
    // share of the grid and streams activations; with the M-tile fastest the 10-20 MB weight images are re-streamed
    // through every XCD's 4-MB L2 (HBM-side traffic of the first decoder conv: 3.8 GB -> 1.6 GB per launch)
    int id = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    // (p.tsplit, unused by this kernel otherwise, carries the block order: 2 = the M-tiles of one (row, time tile) are NEIGHBOURS --
    // for a weight image that stays in L2 anyway it is the activation slab the second M-tile should find there)
    int mt_f = 0;
    if (p.tsplit == 2) {
        mt_f = id % p.nmt;
        id /= p.nmt;
    }
    const int tt = id % p.ntt;
    id /= p.ntt;
    const int nrow = d.B * d.Fout;
    const int row = id % nrow;
    const int mt = p.tsplit == 2 ? mt_f : id / nrow;
    const int b = row / d.Fout, fo = row - b * d.Fout;
    const int m0 = mt * BM, t0 = tt * BN;
    const int wset = d.transposed ? (fo % d.fstride) : 0;
    const int fbase = (d.transposed ? (fo / d.fstride) : (fo * d.fstride)) + p.f_lo;
    const h16* s0 = (const h16*)d.src0;
    const h16* s1 = (const h16*)d.src1;
    const h16* zp = aero_zero_page;
    const int C0 = d.C0, C01 = d.C0 + d.C1, T = d.T;
    const int st0 = (int)d.s0_t, st1 = (int)d.s1_t;
    const int cpt = p.Cp / KC;
    const int cc_lo = (s0 == nullptr && C0 % KC == 0) ? C0 / KC : 0;
    const int nTg = p.nT / NT;                                        // slab groups per (frequency tap, channel chunk): 1 when NT == nT
    const int nF = d.ntaps / p.nT;

    // ---- per-lane copy sources: one 64-bit pointer per copy instruction, a chunk adds one block-uniform 32-bit offset.
    // A comes from the PRE-TILED weight image (include/aero_hip.h, `weight_tiled`): the block of (M-tile mt, chunk kc) is
    // the LDS tile itself.  A slab copy instruction `s` (64 lanes x 16 B) fills 16 slab rows x 4 slots; slab row r holds
    // time step t0 + t_lo + (tap group offset) + r.
    const int nkc = p.Ktot / KC;
    const h16* Wt = (const h16*)d.weight_tiled + ((int64_t)wset * p.nmt + mt) * nkc * (BM * KC);
    const h16* a_ptr[NA];
    int a_slot_off[NA];
    const h16* pb0[NBG];
    const h16* pb1[NBG];
    int b_pos[NBG], b_q8[NBG], b_dst[NBG];
    const h16* base0 = s0 ? s0 + (int64_t)b * d.s0_b : zp;
    const h16* base1 = s1 ? s1 + (int64_t)b * d.s1_b - C0 : zp;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int s = wave + G::NW * i;
        if (s >= BM / 16) s -= BM / 16;                       // uniform copy count per wave: a surplus wave repeats a piece
        a_ptr[i] = Wt + (s * 512 + lane * 8);
        a_slot_off[i] = s * 512;
    }
#pragma unroll
    for (int i = 0; i < NBG; ++i) {
        int s = wave + G::NW * i;
        if (s >= G::SLABI) s = G::SLABI - 1;                  // (the halo piece: every wave copies it, identical bytes)
        const int pos = s * 16 + (lane >> 2), q = (lane & 3) ^ aero_tile_swz<KC>(pos);
        b_pos[i] = pos;
        b_q8[i] = q * 8;
        b_dst[i] = s * 512;
        pb0[i] = base0 + (pos * st0 + q * 8);
        pb1[i] = base1 + (pos * st1 + q * 8);
    }
    const h16* zpv = zp;
    int Tv = T;
#ifndef AERO_EMU
    asm volatile("" : "+v"(zpv));
    asm volatile("" : "+v"(Tv));
#endif

    // ---- chunk sequence: groups (frequency tap jf, channel chunk cc, tap group jg) of NT chunks (time taps).  The valid
    // frequency taps of a regular grid are a contiguous range [jf_lo, jf_hi), so the number of groups is known up front and
    // the iterator is BRANCH-FREE (scalar selects): the loop body is straight-line code.
    const int f_step = p.f_step, t_step = p.t_step;
    const int s0f = (int)d.s0_f, s1f = (int)d.s1_f;
    int jf_lo = 0, jf_hi = 0;
    {
        bool seen = false;
#pragma unroll 1
        for (int j = 0; j < nF; ++j) {
            const int f = fbase + j * f_step;
            if (f >= 0 && f < d.Fin) {
                if (!seen) jf_lo = j;
                seen = true;
                jf_hi = j + 1;
            }
        }
    }
    const int ncc = cpt - cc_lo;
    const int ng = (ABL & 128) ? 0 : (jf_hi - jf_lo) * ncc * nTg;      // groups of this block (block-uniform)
    const bool has0 = s0 != nullptr;
    // A stream: state of the next chunk whose weight tile is issued (AHEAD chunks ahead of the compute)
    int a_jg = 0, a_cc = cc_lo, a_jf = jf_lo, a_n = 0, a_jt = 0, a_slot = 0;
    auto issue_a = [&]() {
        const bool ok = (ABL & 64) ? false : a_n < ng;
        int kofs = ((a_jf * p.nT + a_jg * NT + a_jt) * cpt + a_cc) * (BM * KC);
        kofs = ok ? kofs : 0;                                          // past the end: chunk 0 again, harmless (its B is zero)
        h16* As = smem + a_slot * G::A_SLOT;
#pragma unroll
        for (int i = 0; i < NA; ++i) aero_glds16(a_ptr[i] + kofs, As + a_slot_off[i]);
        a_slot = a_slot + 1 == NSA ? 0 : a_slot + 1;
        // advance: time tap inside the group fastest, then tap group, channel chunk, frequency tap
        const int jt1 = a_jt + 1;
        const bool wt = jt1 == NT;
        a_jt = wt ? 0 : jt1;
        a_n += wt ? 1 : 0;
        const int jg1 = a_jg + (wt ? 1 : 0);
        const bool wg = jg1 == nTg;
        a_jg = wg ? 0 : jg1;
        const int cc1 = a_cc + (wg ? 1 : 0);
        const bool wc = cc1 == cpt;
        a_cc = wc ? cc_lo : cc1;
        a_jf += wc ? 1 : 0;
    };
    // B stream: state of the next GROUP whose activation slab is issued (LB groups ahead), pieces [lo, hi) of it
    int b_jg = 0, b_cc = cc_lo, b_jf = jf_lo, b_n = 0, b_slot = 0;
    auto issue_b = [&](int lo, int hi, bool last) {
        const bool ok = (ABL & 64) ? false : b_n < ng;
        const int fi = fbase + b_jf * f_step;
        const int tsh = t0 + p.t_lo + b_jg * NT * t_step;
        const int c_lo = b_cc * KC;
        const int off0 = fi * s0f + tsh * st0 + c_lo;
        const int off1 = fi * s1f + tsh * st1 + c_lo;
        const int Teff = ok ? Tv : 0;                                  // past the end: every lane reads the zero page
        const int lim0 = C0 - c_lo, lim1 = C01 - c_lo;
        h16* Bs = smemB + b_slot * G::B_SLOT;
#pragma unroll
        for (int i = 0; i < NBG; ++i) {
            if (i < lo || i >= hi) continue;
            const bool tin = (unsigned)(b_pos[i] + tsh) < (unsigned)Teff;
            const bool u0 = b_q8[i] < lim0;
            const bool okl = tin && (u0 ? has0 : (b_q8[i] < lim1));
            const h16* ptr = u0 ? pb0[i] + off0 : pb1[i] + off1;
            aero_glds16(okl ? ptr : zpv, Bs + b_dst[i]);
        }
        if (last) {
            b_slot = b_slot + 1 == NSB ? 0 : b_slot + 1;
            ++b_n;
            const int jg1 = b_jg + 1;
            const bool wg = jg1 == nTg;
            b_jg = wg ? 0 : jg1;
            const int cc1 = b_cc + (wg ? 1 : 0);
            const bool wc = cc1 == cpt;
            b_cc = wc ? cc_lo : cc1;
            b_jf += wc ? 1 : 0;
        }
    };

    f32x16 acc[NRB][2];
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    // ---- fragment addressing: lane l reads row (l & 31), 16-byte slot ks*2 + (l >> 5) of a 64-byte tile row; time tap jt
    // reads the slab jt*t_step rows further down (the swizzle follows the row, so each tap has its own lane offsets)
    int fa_off[2], fb_off[NT][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        fa_off[ks] = aero_tile_off_kc<KC>(wm * NRB * 32 + (lane & 31), ks * 2 + (lane >> 5));
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) fb_off[jt][ks] = aero_tile_off_kc<KC>(wn * 64 + (lane & 31) + jt * t_step, ks * 2 + (lane >> 5));
    }
    // (+32 rows = +1024 h16: the swizzle depends on (row >> 2) & 3 only)
    auto read_a = [&](h16x8 (&A)[RBP][2], int slot, int rb0) {
        const h16* S = smem + slot * G::A_SLOT;
#pragma unroll
        for (int i = 0; i < RBP; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) A[i][ks] = *(const h16x8*)&S[fa_off[ks] + (rb0 + i) * 1024];
    };
    auto read_b = [&](h16x8 (&Bf)[2][2], int slot, int jt) {
        const h16* S = smemB + slot * G::B_SLOT;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) Bf[n][ks] = *(const h16x8*)&S[fb_off[jt][ks] + n * 1024];
    };
    auto mma = [&](const h16x8 (&A)[RBP][2], const h16x8 (&Bf)[2][2], int rb0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < RBP; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[rb0 + i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i][ks], Bf[n][ks], acc[rb0 + i][n], 0, 0, 0);
    };
    // ask the scheduler to spread the phase's bookkeeping (fragment reads, copy issue, iterator) over the gaps between its
    // eight MFMAs instead of running it as a block in front of them.  Masks: 0x8 MFMA, 0x100 DS read, 0x20 VMEM read,
    // 0x4 SALU, 0x2 VALU.
    auto interleave = [&](int n_ds, int n_vmem) {
#ifndef AERO_EMU
        if constexpr ((ABL & 8) == 0) {
#pragma unroll
            for (int g = 0; g < 4 * RBP; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if (g < n_ds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x4, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
                if (g >= 2 && g - 2 < n_vmem) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
            }
        }
#endif
    };

    // ---- prologue: slabs of the first LB groups and the weight tiles of the first AHEAD chunks in flight
#pragma unroll
    for (int g = 0; g < G::LB; ++g) issue_b(0, NBG, true);
#pragma unroll
    for (int a = 0; a < G::AHEAD; ++a) issue_a();
    // every prologue copy has landed before the loop starts: the counted waits inside the loop are derived for the
    // steady-state issue order (slab pieces, then a weight tile, per chunk), which the prologue's order is not
    aero_wait_vm<0>();
    aero_phase_barrier();
    h16x8 A0[RBP][2], A1[RBP][2], B0[2][2], B1[2][2];
    read_a(A0, 0, 0);
    read_b(B0, 0, 0);
#ifndef AERO_EMU
    // (the loop must be ENTERED with hipcc's LDS scoreboard empty: otherwise the merged state at the loop header makes it
    // wait lgkmcnt(0) in front of every phase-0 MFMA group, i.e. also for the prefetch reads issued just before)
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    // one phase per chunk: the first trip already refills ring slot 0, so every wave must have fetched chunk 0's
    // fragments from it first (with two phases the first refill comes after the barrier that ends phase 0)
    if constexpr (NPH == 1) aero_phase_barrier();
    if constexpr ((ABL & 4) != 0) {                         // (profiling builds without fragment reads in the loop)
        read_a(A1, 0, NPH == 2 ? 2 : 0);
        read_b(B1, 0, 0);
    }

    // counted wait of chunk position jt (jt is a constant after unrolling: the chain folds to one s_waitcnt)
    auto wait_next = [&](int jt) {
        if (jt == 0) aero_wait_vm<G::vmw(0)>();
        else if (jt == 1) aero_wait_vm<G::vmw(1 % NT)>();
        else aero_wait_vm<G::vmw(2 % NT)>();
    };
    int ra_slot = 0, rb_slot = 0;                           // ring slots of the chunk / group being computed
    auto next_a = [&](int s) { return s + 1 == NSA ? 0 : s + 1; };
    auto next_b = [&](int s) { return s + 1 == NSB ? 0 : s + 1; };
    // One GROUP per trip, its NT chunks unrolled (static tap offsets and copy counts).  The operands fetched for the next
    // chunk are handed over by register copies (16-32 v_mov per 16 MFMAs): alternating two register sets by NAME over an
    // unrolled pair of chunks made the allocator spill 130+ registers, accumulators included.
#pragma unroll 1
    for (int g = 0; g < ng; ++g) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const int na_slot = next_a(ra_slot);
            // the chunk after this one: same slab shifted one tap further, or the next group's slab at tap 0
            const int nb_slot = jt + 1 < NT ? rb_slot : next_b(rb_slot);
            const int njt = jt + 1 < NT ? jt + 1 : 0;
            constexpr int dummy = 0;
            (void)dummy;
            const int blo = G::nb_before(jt), bhi = G::nb_before(jt) + G::nb(jt);
            if constexpr (NPH == 2) {
                // phase 0: row blocks 0,1; fetch row blocks 2,3 of this chunk; this chunk's share of the slab copies
                if constexpr (!(ABL & 4)) read_a(A1, ra_slot, 2);
                if constexpr (!(ABL & 2)) issue_b(blo, bhi, jt == NT - 1);
                mma(A0, B0, 0);
                interleave(4, G::nb(jt));
                if constexpr (!(ABL & 2) && !(ABL & 32)) wait_next(jt);                    // the next chunk's operands have landed
                if constexpr (!(ABL & 1)) aero_phase_barrier();
                else aero_sched_fence();
                // phase 1: row blocks 2,3; fetch the next chunk's first operands; the weight tile AHEAD chunks ahead
                if constexpr (!(ABL & 4)) {
                    read_a(A0, na_slot, 0);
                    read_b(B1, nb_slot, njt);
                }
                if constexpr (!(ABL & 2)) issue_a();
                mma(A1, B0, 2);
                interleave(8, NA);
                if constexpr (!(ABL & 1)) aero_phase_barrier();
                else aero_sched_fence();
            } else {
                if constexpr (!(ABL & 4)) {
                    read_a(A1, na_slot, 0);
                    read_b(B1, nb_slot, njt);
                }
                if constexpr (!(ABL & 2)) {
                    issue_b(blo, bhi, jt == NT - 1);
                    issue_a();
                }
                mma(A0, B0, 0);
                interleave(2 * RBP + 4, NA + G::nb(jt));
                if constexpr (!(ABL & 2) && !(ABL & 32)) wait_next(jt);                    // chunk k+2 has landed
                if constexpr (!(ABL & 1)) aero_phase_barrier();
                else aero_sched_fence();
            }
            ra_slot = na_slot;
            rb_slot = nb_slot;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int n = 0; n < 2; ++n) B0[n][ks] = B1[n][ks];
                if constexpr (NPH == 1) {
#pragma unroll
                    for (int i = 0; i < RBP; ++i) A0[i][ks] = A1[i][ks];
                }
            }
        }
    }
    aero_wait_vm<0>();
    aero_phase_barrier();                  // every wave is done with the rings: they become the output staging tile
    if constexpr (WM == 2 && (WN == 4 || WN == 2) && NRB == 3 && ABL == 0) {
        if (d.tail_w) {                                    // (block-uniform: the fused transposed-conv tail instead of the activation's store)
            aero_ring_tail_epilogue<WM, WN, NRB>(p, acc, smem, b, fo, m0, t0);
            return;
        }
    }
    if (d.stat_mode == 1) {
        aero_ring_epilogue<WM, WN, NRB, AERO_ACT_NONE, 1>(p, acc, smem, b, fo, m0, t0);
        return;
    }
    switch (d.act) {
        case AERO_ACT_NONE: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_NONE>(p, acc, smem, b, fo, m0, t0); break;
        case AERO_ACT_RELU: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_RELU>(p, acc, smem, b, fo, m0, t0); break;
        case AERO_ACT_GELU: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_GELU>(p, acc, smem, b, fo, m0, t0); break;
        default: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_GLU>(p, acc, smem, b, fo, m0, t0); break;
    }
}

// AERO_CONV_RING=0 keeps every conv on the k_conv.h kernels (A/B experiments, bisecting); 1 = only the 256-row tile
static int aero_conv_ring_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AERO_CONV_RING");
        v = e ? atoi(e) : 2;
    }
    return v;
}

// tile height by shape (also exported: the caller prepares `weight_tiled` for it)
static int aero_conv_ring_pick_bm(int M, int Ktot) {
    const int mode = aero_conv_ring_mode();
    if (!mode || M <= 0 || Ktot % 32) return 0;
    // shortest contraction the 256-row tile takes (AERO_RING_KMIN256, A/B; 1024 until round 4): the deepest encoder's 1x1 rewrite conv
    // (M 768, K 384: twelve chunks) runs 160 us here WITH its GroupNorm sums against 147 + 35 us (LDS-tiled conv + statistics pass)
    static int kmin256 = -1;
    if (kmin256 < 0) { const char* e = getenv("AERO_RING_KMIN256"); kmin256 = e ? atoi(e) : 384; }
    if (M % 256 == 0 && Ktot >= kmin256) return 256;
    static int no192 = -1;                                       // AERO_RING_TILE192=0: 128 / 64-row x 512-step tiles instead of the 192 x 256 one (A/B)
    if (no192 < 0) { const char* e = getenv("AERO_RING_TILE192"); no192 = (e && e[0] == '0') ? 1 : 0; }
    // 96-row x 256-step tiles on FOUR waves (two 77-KiB blocks per CU) for contractions with exactly 96 rows (the dilated Conv1d of the
    // deepest DConv: K = 3 x 384): 66 -> 44 us against the 4-wave glds tile.  AERO_RING_HALF=2 also gives the 192 / 384-row decoder convs
    // this tile (round-4 experiment: 694 -> 666 us and 821 -> 831 us -- neutral: those loops are bound by LDS operand bandwidth, ~107 of
    // 128 B/clk per CU for a 96 x 64 wave tile, not by the per-block prologue / epilogue a second resident block would hide); 0 = off
    static int half = -1;
    if (half < 0) { const char* e = getenv("AERO_RING_HALF"); half = e ? atoi(e) : 1; }
    if (mode >= 2 && half && M == 96 && Ktot >= 768) return 96;
    if (mode >= 2 && half >= 2 && M % 96 == 0 && M <= 384 && Ktot >= 768 && Ktot <= 2048) return 96;
    if (mode >= 2 && !no192 && M % 192 == 0 && Ktot >= 768) return 192;
    if (mode >= 2 && M % 128 == 0 && Ktot >= 768) return 128;
    if (mode >= 2 && M % 64 == 0 && Ktot >= 768) return 64;
    return 0;
}

template <int WM, int WN, int NRB, int NT>
static void aero_conv_ring_go(AeroConvK& p, hipStream_t stream, char* name) {
    typedef AeroRingGeom<WM, WN, NRB, NT> G;
    const aero_conv_desc& d = p.d;
    p.ntt = (d.T + G::BN - 1) / G::BN;
    p.nmt = d.M / G::BM;
    if (name) {
        snprintf(name, 96, "aero_conv_ring_kernel<%d, %d, %d, %d, 0>", WM, WN, NRB, NT);
        return;
    }
    const long nwg = (long)d.B * d.Fout * p.ntt * p.nmt;
    const dim3 grid((unsigned)nwg), block(G::NW * 64);
    const size_t lds = G::SMEM * sizeof(h16);
#ifdef AERO_RING_ABLATION
    if constexpr (WM == 2 && NT == 3) {
        static int abl = -1;
        if (abl < 0) { const char* e = getenv("AERO_RING_ABL"); abl = e ? atoi(e) : 0; }
        switch (abl) {
            case 1: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 1>), grid, block, lds, stream, p); return;
            case 2: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 2>), grid, block, lds, stream, p); return;
            case 4: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 4>), grid, block, lds, stream, p); return;
            case 6: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 6>), grid, block, lds, stream, p); return;
            case 8: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 8>), grid, block, lds, stream, p); return;
            case 32: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 32>), grid, block, lds, stream, p); return;
            case 64: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 64>), grid, block, lds, stream, p); return;
            case 128: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT, 128>), grid, block, lds, stream, p); return;
            default: break;
        }
    }
#endif
    AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, NT>), grid, block, lds, stream, p);
}

#ifndef AERO_RING_ONLY
AERO_XPART bool aero_conv_ring_try(const aero_conv_desc* d, AeroConvK& p, hipStream_t stream, char* name) {
    const int mode = aero_conv_ring_mode();
    if (!mode) return false;
    // what the lean epilogue of this kernel covers: bias, activation, fp16 channels-last rows, no trim / residual /
    // embedding / per-item affine / statistics / scatter
    const bool stats_ok = d->stat_mode == 0 || (d->stat_mode == 1 && d->act == AERO_ACT_NONE && d->stats && d->stat_G >= 1 &&
                                                 d->M % d->stat_G == 0 && (d->M / d->stat_G) % 32 == 0);
    if (!p.staged || !stats_ok || d->res || d->post_add || d->batch_scale || d->scatter_M || d->dst_f32 || d->dst_f_off != 0 ||
        d->dst_F != d->Fout || (d->bias && ((uintptr_t)d->bias & 15)))
        return false;
    const int bm = aero_conv_ring_pick_bm(d->M, p.Ktot);
    if (!bm || !d->weight_tiled || d->tiled_bm != bm || ((uintptr_t)d->weight_tiled & 15)) return false;
    // fused transposed-conv tail: only the 192-row tile with the shared three-tap slab carries it, and the block must hold every channel
    if (d->tail_w && !(bm == 192 && d->M == 192 && d->act == AERO_ACT_GLU && p.nT == 3 && (p.t_step == 1 || p.t_step == 2))) return false;
    if ((long)d->B * d->Fout * ((d->T + 255) / 256) * (d->M / bm) > 0x7fffffffL) return false;
    // three unit-stride time taps share one activation slab; any other tap grid: one tile per tap (256-row tile only:
    // the 512-step tiles have no LDS for four full activation tiles)
    const bool slab3 = p.nT == 3 && (p.t_step == 1 || p.t_step == 2);   // (dilation 2: the taps read rows 0 / 2 / 4 of the slab's 16 spare rows)
    if (bm == 256) {
        // Time tiles: 256 steps on eight waves, or -- when the last 256-step tile would be at most half full, i.e. the number of 128-step
        // tiles is odd -- 128 steps on FOUR waves with a three-slot weight ring (76.8 KiB: two blocks per CU).  At T = 501 (configs 2 / 3:
        // two full tiles) the narrow tile is neither faster nor slower per launch and 1 % slower in the pipelined step
        // (profiles/r05_ring256_narrow_ab.txt); at T = 376 (config 4: 12->48 kHz, hop 256) the wide tile computes 512 columns for 376:
        // D0 1038 -> 820 us, D1 1118 -> 865 us, config 4's pipelined step 7.07 -> 6.8-6.9 ms (profiles/r05_config4_narrow.txt).
        // AERO_RING_256X128=0 / 1 forces wide / narrow (A/B).
        static int narrow256 = -1;
        if (narrow256 < 0) { const char* e = getenv("AERO_RING_256X128"); narrow256 = e ? atoi(e) : 2; }
        const bool narrow = narrow256 == 1 || (narrow256 == 2 && (((d->T + 127) / 128) & 1));
        if (slab3 && narrow) aero_conv_ring_go<2, 2, 4, 3>(p, stream, name);
        else if (slab3) aero_conv_ring_go<2, 4, 4, 3>(p, stream, name);
        else aero_conv_ring_go<2, 4, 4, 1>(p, stream, name);
    } else if (!slab3) {                                         // (round 4: the 192-row tile with one slab per tap, <2,4,3,1>, was tried for the
        // encoder's strided [8,1] conv with M = 384: 173 us against 168 us on the glds8 tile, and the B = 32 forward no longer matched
        // its two B = 16 halves (2.7e-4) -- the one-phase / one-tap-per-slab combination of the pipeline is unvalidated: not used)
        return false;
    } else if (bm == 96) {
        p.tsplit = 2;                                            // M-tiles adjacent (see the kernel's block order)
        aero_conv_ring_go<1, 4, 3, 3>(p, stream, name);
        p.tsplit = 1;
    } else if (bm == 192) {
        // Round 5: the 192-row tile on FOUR waves over 128 time steps, two 77-KiB blocks per CU at FULL height -- one
        // block's prologue / epilogue (22 % of a 27-chunk tile's life) runs under the other block's K loop, at twice the weight-tile
        // traffic per FLOP.  Measured against the 8-wave 192 x 256 tile: D2 701 -> 672-699 us, D3 843 -> 807-822 us, bench 8.90 -> 8.81 ms
        // (profiles/r05_ring_narrow_ab.txt).  AERO_RING_192X128=0 restores the 8-wave tile.
        static int narrow = -1;
        if (narrow < 0) { const char* e = getenv("AERO_RING_192X128"); narrow = e ? atoi(e) : 1; }
        if (narrow) aero_conv_ring_go<2, 2, 3, 3>(p, stream, name);
        else aero_conv_ring_go<2, 4, 3, 3>(p, stream, name);
    } else if (bm == 128) {
        aero_conv_ring_go<1, 8, 4, 3>(p, stream, name);
    } else {
        aero_conv_ring_go<1, 8, 2, 3>(p, stream, name);
    }
    return true;
}
#endif
