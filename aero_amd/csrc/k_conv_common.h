// k_conv_common.h -- launch parameters and LDS tile addressing shared by the convolution kernels (k_conv.h, k_conv_ring.h)
#pragma once
#include "aero_common.h"

struct AeroConvK {
    aero_conv_desc d;
    int Cp, cpt, Ktot, Mpad, nmt, ntt, vec_in, vec4, vec_out, staged, glds;
    int nT, f_lo, f_step, t_lo, t_step;      // regular tap grid: df = f_lo + (j / nT) * f_step, dt = t_lo + (j % nT) * t_step
    int tsplit;                              // >= 1: groups the time taps are split into (aero_conv_desc.tap_split)
};

// LDS image of a [rows][KC] fp16 operand tile: 16-byte slot `slot` of row `row`, XOR-swizzled so that the ds_read_b128
// fragment reads of both MFMA shapes (16x16x32: lane -> row l&15, slot l>>4; 32x32x16: row l&31, slot 2*ks + (l>>5)) are
// bank-conflict free.  The direct global->LDS copies write lane-linear, so the permutation goes on the SOURCE address.
template <int KC>
static __device__ __forceinline__ int aero_tile_off_kc(int row, int slot) {
    if (KC == 32) return row * 32 + ((slot ^ ((0 - (row >> 2)) & 3)) << 3);
    return row * 64 + ((slot ^ ((row >> 1) & 7)) << 3);
}
template <int KC>
static __device__ __forceinline__ int aero_tile_swz(int row) {
    return KC == 32 ? ((0 - (row >> 2)) & 3) : ((row >> 1) & 7);
}

