// k_conv_ring.h -- the wide contractions of the path (decoder 3x3 "rewrite" convs, aero.py:179: 68 % of the FLOPs) as a
// software-pipelined implicit GEMM on v_mfma_f32_32x32x16_f16.
//
// Same GEMM view as k_conv.h (one (b, fo) row per block; m = output channel, n = time step, k = (tap, channel)), but
// built around what PMC showed the 2-stage / one-__syncthreads-per-chunk kernels wait for:
//   * EIGHT waves per block, ONE block per CU (256 registers per wave): wave tile (NRB*32) x 64, i.e. 128 x 64 for the
//     256-row tiles -- 0.75 LDS operand reads per MFMA-FLOP of the 64 x 64 wave tile, 2/3 of the global->LDS bytes;
//   * a RING of NS = 3-4 LDS slots, one 32-channel K-chunk each, filled by `global_load_lds_dwordx4` AHEAD tiles ahead;
//     copies stay in flight across barriers: ONE counted `s_waitcnt vmcnt(N)` per K-chunk (never 0 in steady state) and
//     raw `s_barrier`s that wait for LDS traffic only;
//   * operand fragments of phase g+1 are read from LDS while the MFMAs of phase g run (two register sets, 192 registers
//     in all), so neither the LDS round trip nor the copy latency sits between two MFMA groups;
//   * a phase = 8 x v_mfma_f32_32x32x16_f16 per wave (two 32-row blocks x 64 steps x 32 channels), one barrier per phase.
// Tiles: <WM,WN,NRB> = <2,4,4>: 256 rows x 256 steps (M % 256 == 0);  <1,8,4>: 128 rows x 512 steps (a whole T = 501
// row per block);  <1,8,2>: 64 rows x 512 steps.
// Ordering rules the schedule relies on (LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt followed by a
// barrier the reader has passed): the wait for tile j sits before the barrier that ENDS the phase preceding the phase
// in which tile j is first read; a slot is refilled only after the barrier ending the phase of its last read.
// Roofline: MFMA.  Epilogue: + bias, NONE / ReLU / GELU / GLU, fp16 channels-last through an LDS transpose.
#pragma once
#include "k_conv_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int NRB>
struct AeroRingGeom {
    static constexpr int BM = WM * NRB * 32, BN = WN * 64;
    static constexpr int SLOT = (BM + BN) * 32;                                   // h16 elements per ring slot
    static constexpr int NS = (4 * SLOT * 2 <= 144 * 1024) ? 4 : 3;
    static constexpr int NPH = NRB / 2;                                           // phases (8 MFMAs each) per K-chunk
    static constexpr int NIA = (BM / 16 + 7) / 8;                                 // A copy instructions per wave and chunk
    static constexpr int NIB = BN / 16 / 8;                                       // B copy instructions per wave and chunk
    static constexpr int NI = NIA + NIB;
    static constexpr int IPP0 = NPH == 1 ? NI : (NI + 1) / 2;                     // copies issued in phase 0 (rest: phase 1)
    static constexpr int AHEAD = NPH == 1 ? NS : NS - 1;                          // tiles between a copy and its use
    // copies issued after tile j's and before the wait that guards tile j's first read (see the schedule above)
    static constexpr int VMW = NPH == 1 ? (NS - 2) * NI : (NS - 3) * NI + IPP0;
    static constexpr int CS = BM + 8;                                             // epilogue staging row (h16), padded
    static constexpr int EPI = WN * 32 * CS;
    static constexpr int SMEM = NS * SLOT > EPI ? NS * SLOT : EPI;                // h16 elements
};

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

template <int WM, int WN, int NRB, int ACT>
static __device__ __forceinline__ void aero_ring_epilogue(const AeroConvK& p, f32x16 (&acc)[NRB][2], h16* Cs, int b, int fo, int m0, int t0) {
    typedef AeroRingGeom<WM, WN, NRB> G;
    constexpr bool GLU = ACT == AERO_ACT_GLU;
    constexpr int BMo = GLU ? G::BM / 2 : G::BM;
    constexpr int NVEC = BMo / 8;
    constexpr int NPOS = WN * 32;
    constexpr int NIT = (NPOS * NVEC + 511) / 512;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int T = d.T, M = d.M;
    const int Mout = GLU ? (M >> 1) : M;
    const int m0o = GLU ? (m0 >> 1) : m0;
    h16* drow = (h16*)d.dst + (int64_t)b * d.d_b + (int64_t)fo * d.d_f + m0o;
    const int hi4 = (lane >> 5) * 4;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int pc = wn * 32 + (lane & 31);
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
            const int idx = tid + it * 512;
            const int pos = idx / NVEC, cv = idx - pos * NVEC;
            const int t = t0 + (pos >> 5) * 64 + cb * 32 + (pos & 31);
            if (idx < NPOS * NVEC && t < T && m0o + cv * 8 < Mout)
                *(h16x8*)(drow + (int64_t)t * d.d_t + cv * 8) = *(const h16x8*)&Cs[pos * G::CS + cv * 8];
        }
        aero_lds_barrier();
    }
}

// ABL: ablation bits for profiling builds (results are WRONG with any bit set): 1 no barriers in the K loop, 2 no copies
// in the loop, 4 no fragment reads in the loop, 8 no interleave hints, 16 frozen chunk iterator, 32 no counted vmcnt wait,
// 64 every copy reads the zero page, 128 no K loop at all (prologue + epilogue only)
template <int WM, int WN, int NRB, int ABL = 0>
__global__ __launch_bounds__(512, 2) void aero_conv_ring_kernel(AeroConvK p) {
    typedef AeroRingGeom<WM, WN, NRB> G;
    constexpr int BM = G::BM, BN = G::BN, NS = G::NS, NPH = G::NPH, NIA = G::NIA, NIB = G::NIB, SLOT = G::SLOT;
    constexpr int KC = 32;
    h16* smem = (h16*)AERO_DYN_SMEM;
    const aero_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = aero_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // block order: the M-tile is the SLOWEST index, so that at any time all CUs share one weight tile (L2 resident) and
    // stream activations; with the M-tile fastest the 10-20 MB weight images are re-streamed through every XCD's 4-MB L2.
    int id = aero_xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
    const int tt = id % p.ntt;
    id /= p.ntt;
    const int nrow = d.B * d.Fout;
    const int row = id % nrow;
    const int mt = id / nrow;
    const int b = row / d.Fout, fo = row - b * d.Fout;
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

    // ---- per-lane copy sources (as in aero_conv_glds_body): one 64-bit pointer per copy instruction, a chunk adds one
    // block-uniform 32-bit element offset.  Copy instruction `s` (64 lanes x 16 B) fills 16 tile rows x 4 slots.
    const h16* a_ptr[NIA];
    const h16* pb0[NIB];
    const h16* pb1[NIB];
    int b_pos[NIB], b_q8[NIB];
    const h16* base0 = s0 ? s0 + (int64_t)b * d.s0_b : zp;
    const h16* base1 = s1 ? s1 + (int64_t)b * d.s1_b - C0 : zp;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        int s = wave + 8 * i;
        if (s >= BM / 16) s -= BM / 16;                       // uniform copy count per wave: a surplus wave repeats a piece
        const int r = s * 16 + (lane >> 2), q = (lane & 3) ^ aero_tile_swz<KC>(r);
        a_ptr[i] = Wp + (r * p.Ktot + q * 8);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int s = wave + 8 * i;
        const int pos = s * 16 + (lane >> 2), q = (lane & 3) ^ aero_tile_swz<KC>(pos);
        b_pos[i] = pos;
        b_q8[i] = q * 8;
        pb0[i] = base0 + (pos * st0 + q * 8);
        pb1[i] = base1 + (pos * st1 + q * 8);
    }
    const h16* zpv = zp;
    int Tv = T;
#ifndef AERO_EMU
    asm volatile("" : "+v"(zpv));
    asm volatile("" : "+v"(Tv));
#endif

    // ---- chunk sequence: (frequency tap jf, channel chunk cc, time tap jt), jt FASTEST so that the three time taps of a
    // 3x3 re-read the same activation lines back to back (L1/L2 hits instead of a reuse distance of a whole row).
    // The valid frequency taps of a regular grid are a contiguous range [jf_lo, jf_hi), so the number of chunks nk is known
    // up front and the iterator is BRANCH-FREE (scalar selects): the loop body is straight-line code that the scheduler can
    // interleave with the MFMAs.  PMC on the first version (branchy iterator, MFMA cluster after it): 6 SALU + 3.6 VALU per
    // MFMA executed in lockstep by all eight waves BEFORE each 8-MFMA group, MFMA pipe 36 % busy, waves parked 33 %.
    const int nT = p.nT, f_step = p.f_step, t_step = p.t_step, Cpk = p.Cp;
    const int s0f = (int)d.s0_f, s1f = (int)d.s1_f;
    const int t_base = t0 + p.t_lo;
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
    const int nk = (ABL & 128) ? 0 : (jf_hi - jf_lo) * ncc * nT;      // K-chunks of this block (block-uniform)
    int it_jt = 0, it_cc = cc_lo, it_jf = jf_lo, it_n = 0;             // the NEXT chunk to issue
    int kofs = 0, off0 = 0, off1 = 0, tsh = 0, c_lo = 0, Teff = 0;
    bool ok = false;
    const bool has0 = s0 != nullptr;
    // state of chunk it_n -> copy parameters, then advance (no branches)
    auto next_chunk = [&]() {
        ok = (ABL & 64) ? false : it_n < nk;
        const int fi = fbase + it_jf * f_step;
        kofs = (it_jf * nT + it_jt) * Cpk + it_cc * KC;
        tsh = t_base + it_jt * t_step;
        c_lo = it_cc * KC;
        off0 = fi * s0f + tsh * st0 + c_lo;
        off1 = fi * s1f + tsh * st1 + c_lo;
        Teff = ok ? T : 0;                                             // past the end: every B lane reads the zero page
        ++it_n;
        const int jt1 = it_jt + 1;
        const bool wt = jt1 == nT;
        it_jt = wt ? 0 : jt1;
        const int cc1 = it_cc + (wt ? 1 : 0);
        const bool wc = cc1 == cpt;
        it_cc = wc ? cc_lo : cc1;
        it_jf += wc ? 1 : 0;
    };
    // copy instructions [lo, hi) of the current chunk into ring slot `slot`; order: A pieces first, then B pieces.
    // Past the last chunk (ok == false) the copies are still issued, from the zero page into a slot nobody reads: the
    // copy counts per phase -- and with them the counted vmcnt waits -- stay the same for every trip of the loop.
    auto issue = [&](int slot, int lo, int hi) {
        h16* As = smem + slot * SLOT;
        h16* Bs = As + BM * KC;
        const int lim0 = C0 - c_lo, lim1 = C01 - c_lo;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            if (i < lo || i >= hi) continue;
            int s = wave + 8 * i;
            if (s >= BM / 16) s -= BM / 16;
            aero_glds16(ok ? a_ptr[i] + kofs : zpv, As + s * 512);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            if (NIA + i < lo || NIA + i >= hi) continue;
            const bool tin = (unsigned)(b_pos[i] + tsh) < (unsigned)Teff;
            const bool u0 = b_q8[i] < lim0;
            const bool okl = tin && (u0 ? has0 : (b_q8[i] < lim1));
            const h16* ptr = u0 ? pb0[i] + off0 : pb1[i] + off1;
            aero_glds16(okl ? ptr : zpv, Bs + (wave + 8 * i) * 512);
        }
    };

    f32x16 acc[NRB][2];
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    // ---- fragment addressing: lane l reads row (l & 31), 16-byte slot ks*2 + (l >> 5) of a 64-byte tile row
    int fa_off[2], fb_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ra = wm * NRB * 32 + (lane & 31), rbn = wn * 64 + (lane & 31);
        fa_off[ks] = aero_tile_off_kc<KC>(ra, ks * 2 + (lane >> 5));
        fb_off[ks] = BM * KC + aero_tile_off_kc<KC>(rbn, ks * 2 + (lane >> 5));
    }
    // A fragments of row blocks rb0, rb0+1 / B fragments of both column blocks, from ring slot `slot` (+32 rows = +1024 h16:
    // the swizzle depends on (row >> 2) & 3 only)
    auto read_a = [&](h16x8 (&A)[2][2], int slot, int rb0) {
        const h16* S = smem + slot * SLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) A[i][ks] = *(const h16x8*)&S[fa_off[ks] + (rb0 + i) * 1024];
    };
    auto read_b = [&](h16x8 (&Bf)[2][2], int slot) {
        const h16* S = smem + slot * SLOT;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) Bf[n][ks] = *(const h16x8*)&S[fb_off[ks] + n * 1024];
    };
    auto mma = [&](const h16x8 (&A)[2][2], const h16x8 (&Bf)[2][2], int rb0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[rb0 + i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i][ks], Bf[n][ks], acc[rb0 + i][n], 0, 0, 0);
    };
    // ask the scheduler to spread the phase's bookkeeping (fragment reads, copy issue, iterator) over the gaps between its
    // eight MFMAs instead of running it as a block in front of them: with two waves per SIMD a wave's MFMA gap is ~64
    // cycles, i.e. ~10 issue slots.  Masks: 0x8 MFMA, 0x100 DS read, 0x20 VMEM read, 0x4 SALU, 0x2 VALU.
    auto interleave = [&](int n_ds, int n_vmem) {
#ifndef AERO_EMU
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (g < n_ds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x4, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
            if (g >= 2 && g - 2 < n_vmem) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
#endif
    };

    // ---- prologue: AHEAD chunks in flight, chunk 0 landed, its first operands in registers
    int wslot = 0;                        // ring slot of the next chunk to issue
    auto next_slot = [&](int s) { return s + 1 == NS ? 0 : s + 1; };
#pragma unroll
    for (int a = 0; a < G::AHEAD; ++a) {
        next_chunk();
        issue(wslot, 0, G::NI);
        wslot = next_slot(wslot);
    }
    aero_wait_vm<(G::AHEAD - 1) * G::NI>();
    aero_phase_barrier();
    h16x8 A0[2][2], A1[2][2], B0[2][2], B1[2][2];
    read_a(A0, 0, 0);
    read_b(B0, 0);
#ifndef AERO_EMU
    // (the loop must be ENTERED with hipcc's LDS scoreboard empty: otherwise the merged state at the loop header makes it
    // wait lgkmcnt(0) in front of every phase-0 MFMA group, i.e. also for the prefetch reads issued just before)
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    if constexpr (NPH == 1) {
        aero_wait_vm<(G::AHEAD - 2) * G::NI>();             // chunk 1 must have landed before phase 0 reads it
        aero_phase_barrier();
    }

    if constexpr ((ABL & 4) != 0) {           // (profiling builds without fragment reads in the loop: defined operands)
        read_a(A1, 0, 2);
        read_b(B1, 0);
    }
    int rslot = 0;                        // ring slot of the chunk being computed
    // One chunk per trip; the operands fetched for the next chunk are handed over by register copies (16-32 v_mov per
    // 16 MFMAs).  Alternating the two register sets by NAME over an unrolled pair of chunks looked free but made the
    // allocator spill 130+ registers, accumulators included (hipcc keeps both role assignments live across the back edge).
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        const int nslot = next_slot(rslot);
        if constexpr (NPH == 2) {
            // phase 0: row blocks 0,1; fetch row blocks 2,3 of this chunk; first half of the copies AHEAD chunks ahead
            if constexpr (!(ABL & 4)) read_a(A1, rslot, 2);
            if constexpr (!(ABL & 16)) next_chunk();
            if constexpr (!(ABL & 2)) issue(wslot, 0, G::IPP0);
            mma(A0, B0, 0);
            if constexpr (!(ABL & 8)) interleave(4, G::IPP0);
            if constexpr (!(ABL & 2) && !(ABL & 32)) aero_wait_vm<G::VMW>();          // chunk k+1 has landed (this wave's part)
            if constexpr (!(ABL & 1)) aero_phase_barrier();
            else aero_sched_fence();
            // phase 1: row blocks 2,3; fetch the next chunk's first operands; second half of the copies
            if constexpr (!(ABL & 4)) {
                read_a(A0, nslot, 0);
                read_b(B1, nslot);
            }
            if constexpr (!(ABL & 2)) issue(wslot, G::IPP0, G::NI);
            wslot = next_slot(wslot);
            mma(A1, B0, 2);
            if constexpr (!(ABL & 8)) interleave(8, G::NI - G::IPP0);
            if constexpr (!(ABL & 1)) aero_phase_barrier();
            else aero_sched_fence();
        } else {
            read_a(A1, nslot, 0);
            read_b(B1, nslot);
            next_chunk();
            issue(wslot, 0, G::NI);
            wslot = next_slot(wslot);
            mma(A0, B0, 0);
            interleave(8, G::NI);
            aero_wait_vm<G::VMW>();                         // chunk k+2 has landed
            aero_phase_barrier();
        }
        rslot = nslot;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                B0[n][ks] = B1[n][ks];
                if constexpr (NPH == 1) A0[n][ks] = A1[n][ks];
            }
    }
    aero_wait_vm<0>();
    aero_phase_barrier();                  // every wave is done with the ring: it becomes the output staging tile
    switch (d.act) {
        case AERO_ACT_NONE: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_NONE>(p, acc, smem, b, fo, m0, t0); break;
        case AERO_ACT_RELU: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_RELU>(p, acc, smem, b, fo, m0, t0); break;
        case AERO_ACT_GELU: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_GELU>(p, acc, smem, b, fo, m0, t0); break;
        default: aero_ring_epilogue<WM, WN, NRB, AERO_ACT_GLU>(p, acc, smem, b, fo, m0, t0); break;
    }
}

// AERO_CONV_RING=0 keeps every conv on the k_conv.h kernels (A/B experiments, bisecting)
static int aero_conv_ring_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("AERO_CONV_RING");
        v = e ? atoi(e) : 2;
    }
    return v;
}

template <int WM, int WN, int NRB>
static void aero_conv_ring_go(AeroConvK& p, hipStream_t stream, char* name) {
    typedef AeroRingGeom<WM, WN, NRB> G;
    const aero_conv_desc& d = p.d;
    p.ntt = (d.T + G::BN - 1) / G::BN;
    p.nmt = d.M / G::BM;
    if (name) {
        snprintf(name, 96, "aero_conv_ring_kernel<%d, %d, %d>", WM, WN, NRB);
        return;
    }
    const long nwg = (long)d.B * d.Fout * p.ntt * p.nmt;
    const dim3 grid((unsigned)nwg), block(512);
    const size_t lds = G::SMEM * sizeof(h16);
#ifdef AERO_RING_ABLATION
    if constexpr (WM == 2) {
        static int abl = -1;
        if (abl < 0) { const char* e = getenv("AERO_RING_ABL"); abl = e ? atoi(e) : 0; }
        switch (abl) {
            case 1: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 1>), grid, block, lds, stream, p); return;
            case 2: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 2>), grid, block, lds, stream, p); return;
            case 4: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 4>), grid, block, lds, stream, p); return;
            case 6: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 6>), grid, block, lds, stream, p); return;
            case 7: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 7>), grid, block, lds, stream, p); return;
            case 8: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 8>), grid, block, lds, stream, p); return;
            case 16: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 16>), grid, block, lds, stream, p); return;
            case 22: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 22>), grid, block, lds, stream, p); return;
            case 23: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 23>), grid, block, lds, stream, p); return;
            case 32: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 32>), grid, block, lds, stream, p); return;
            case 64: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 64>), grid, block, lds, stream, p); return;
            case 96: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 96>), grid, block, lds, stream, p); return;
            case 128: AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB, 128>), grid, block, lds, stream, p); return;
            default: break;
        }
    }
#endif
    AERO_LAUNCH_DYN((aero_conv_ring_kernel<WM, WN, NRB>), grid, block, lds, stream, p);
}

#ifndef AERO_RING_ONLY
static bool aero_conv_ring_try(const aero_conv_desc* d, AeroConvK& p, hipStream_t stream, char* name) {
    const int mode = aero_conv_ring_mode();
    if (!mode) return false;
    // what the lean epilogue of this kernel covers: bias, activation, fp16 channels-last rows, no trim / residual /
    // embedding / per-item affine / statistics / scatter
    if (!p.staged || d->stat_mode || d->res || d->post_add || d->batch_scale || d->scatter_M || d->dst_f32 || d->dst_f_off != 0 ||
        d->dst_F != d->Fout || (d->bias && ((uintptr_t)d->bias & 15)))
        return false;
    const long nrow = (long)d->B * d->Fout;
    if (d->M % 256 == 0 && p.Ktot >= 1024) {
        if (nrow * ((d->T + 255) / 256) * (d->M / 256) > 0x7fffffffL) return false;
        aero_conv_ring_go<2, 4, 4>(p, stream, name);
        return true;
    }
    if (mode >= 2 && d->M % 128 == 0 && p.Ktot >= 768) {
        aero_conv_ring_go<1, 8, 4>(p, stream, name);
        return true;
    }
    if (mode >= 2 && d->M % 64 == 0 && p.Ktot >= 768) {
        aero_conv_ring_go<1, 8, 2>(p, stream, name);
        return true;
    }
    return false;
}
#endif
