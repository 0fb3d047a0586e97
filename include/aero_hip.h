/* aero_hip.h -- C ABI of libaero_hip.so: the gfx950 (MI355X) kernels of AERO's spectral path.
 *
 * The reference (slp-rl/aero) has no native/FFI layer: its hot path is `Aero.forward`
 * (src/models/aero.py:446-523) calling ATen ops.  The entry points below sit at those ATen
 * seams; each one cites the reference call site it replaces.
 * The Python binding a maintainer would add is the ctypes stub in INTEGRATION.md
 * (aero_amd/_lib.py is that stub, in product form).
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer (tensor.data_ptr()); nothing is allocated,
 *     freed or synchronised inside the library; scratch is caller provided;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - activations are fp16, channels-last: element (b,f,t,c) at base + b*sb + f*sf + t*st + c
 *     (strides in ELEMENTS); spectrograms are complex64 interleaved [.., F, T] like torch;
 *   - return 0 on success, negative on error; message via aero_last_error() (thread local);
 *     no C++ exception crosses the ABI; all functions are re-entrant.
 *   - process-wide state: none that a call mutates after its first use.  The library reads these environment variables ONCE, at the
 *     first launch of the kernel family concerned (A/B and profiling switches; the defaults are what is tested and benchmarked, and
 *     a value must be set before the first call -- later changes are ignored):
 *       AERO_CONV_RING, AERO_CONV_GLDS, AERO_CONV_MODE, AERO_CONV_BM256, AERO_CONV_SKINNY, AERO_CONV_STREAM, AERO_CONV_TINY_OFF,
 *       AERO_CONVTR_CARRY, AERO_CARRY_QC, AERO_CONV_DEBUG, AERO_RING_ABL, AERO_CONV_KMIN192   (convolution family: kernel selection / ablations)
 *       AERO_LSTM_RING, AERO_LSTM_WIDE                                          (recurrent kernel form)
 *       AERO_ATTN_FOLD                                                          (LocalState: folded vs streaming kernel)
 *       AERO_NORM_CHUNK_KB                                                      (GroupNorm work-item size)
 *       AERO_STFT_DFT_BLOCKS, AERO_ISTFT_WAVES                                  (GEMM-form STFT: blocks per (signal, table quarter);
 *                                                                                iSTFT waves per block)
 *       AERO_WGRAD_ABL                                                          (weight-gradient ablations; AERO_WGRAD_256 -- the tile
 *                                                                                choice -- is the one switch read at every call)
 *       AERO_ATTN_BWD_VALU, AERO_RING_TILE192, AERO_RING_HALF, AERO_RING_KMIN256 (LocalState backward form; ring-tile A/B)
 *       AERO_ISTFT_V2, AERO_NORM_FAST, AERO_NORM_STATS_ROWS                     (round-4 kernel forms: off = the earlier form)
 *     The Python host side has its own AERO_* switches (aero_amd/engine.py); they never reach the library.
 */
#ifndef AERO_HIP_H
#define AERO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AERO_OK 0
#define AERO_ERR_ARG (-1)
#define AERO_ERR_LAUNCH (-2)
#define AERO_ERR_UNSUPPORTED (-3)

enum { AERO_ACT_NONE = 0, AERO_ACT_RELU = 1, AERO_ACT_GELU = 2, AERO_ACT_GLU = 3, AERO_ACT_SNAKE = 4 };

const char* aero_version(void);
const char* aero_last_error(void);
/* the kernel instantiation the calling thread's most recent entry-point call launched, as rocprofv3 prints it
 * (e.g. "void aero_lstm_ring_kernel<12, 2, 3, 6, 4>(AeroLstmK)"); profiling labels only */
const char* aero_last_kernel_name(void);

/* K1 -- torch.stft(center=True, reflect, normalized=True) of spec.py:12-20 as called by
 * Aero._spec (aero.py:409-421).  x [nsig][L] fp32; the signal is treated as right-zero-padded
 * to Lp (a multiple of hop, aero.py:410-411) and then reflect padded by n_fft/2.
 * window: [n_fft] fp32, the analysis window already centred/zero padded to n_fft.
 * spec: complex64 [nsig][n_bins][T], T = 1 + Lp/hop, n_bins = n_fft/2 (Nyquist dropped,
 * aero.py:420) or n_fft/2+1.  If stats != NULL, sum and sum-of-squares of all real/imag
 * values of signal group i/sig_per_item are atomically added to stats[2*item], [2*item+1]
 * (doubles, caller zeroes them) -- the per-item mean/std of aero.py:462-463. */
int aero_stft_fwd(const float* x, int32_t nsig, int32_t L, int32_t Lp, int32_t n_fft, int32_t hop,
                  const float* window, int32_t n_bins, float* spec, int32_t T, double* stats,
                  int32_t sig_per_item, void* stream);

/* K1' -- the same transform for SHORT analysis windows (Aero._spec of the low-rate input: n_fft 512 with a 128-sample window,
 * aero.py:324-328) as a GEMM against a windowed DFT table (k_stft.h): the window's non-zero samples are window[win_off ..
 * win_off + 128) (win_off = (n_fft - win_length) / 2 for the centred window of spec.py:15-16; shorter windows: the rest is the
 * zero padding).  table: fp16, aero_stft_dft_table_bytes(n_fft) bytes, filled once per (window, n_fft) by aero_stft_dft_table.
 * Needs n_fft % 256 == 0, hop % 8 == 0, hop <= 16, n_bins = n_fft/2 (Nyquist dropped).  Same outputs / stats as aero_stft_fwd. */
int64_t aero_stft_dft_table_bytes(int32_t n_fft);
int aero_stft_dft_table(const float* window, int32_t n_fft, int32_t win_off, void* table, void* stream);
int aero_stft_dft_fwd(const float* x, int32_t nsig, int32_t L, int32_t Lp, int32_t n_fft, int32_t hop, int32_t win_off,
                      const void* table, float* spec, int32_t T, double* stats, int32_t sig_per_item, void* stream);

/* K1'+K2 fused (round 6) -- Aero.forward needs the low-rate spectrogram only per-item NORMALISED, as fp16 (aero.py:459-464): the GEMM above
 * run twice instead of once plus a round trip of its fp32 output -- a first pass that only forms the per-item sums (stats: zeroed by the
 * caller, as for aero_stft_dft_fwd), a second that recomputes and stores xn = (v - mean) / (1e-5 + std) as fp16 [nsig][n_fft/2][T][2] and
 * (mean, std) per item.  Same arithmetic, same summation order: xn and mean_std equal aero_stft_dft_fwd + aero_spec_normalize bit for bit. */
int aero_stft_dft_norm_fwd(const float* x, int32_t nsig, int32_t L, int32_t Lp, int32_t n_fft, int32_t hop, int32_t win_off,
                           const void* table, int32_t T, double* stats, int32_t sig_per_item, void* xn, float* mean_std, void* stream);

/* K2 -- aero.py:430-434,462-464: complex -> 2 channels + per-item normalisation.
 * spec viewed as [nitems][n_per_item] fp32; xn fp16 same shape = (v-mean)/(1e-5+std) with the
 * unbiased std; mean_std[2*i], [2*i+1] receive mean and std (used again by K14). */
int aero_spec_normalize(const float* spec, int32_t nitems, int64_t n_per_item, const double* stats,
                        void* xn, float* mean_std, void* stream);

/* K14+K15 -- Aero._ispec / torch.istft (aero.py:423-428, spec.py:30-37): spec complex64
 * [nsig][F][T] with F = n_fft/2 (the Nyquist bin is the implicit zero of aero.py:426, the imaginary
 * part of DC is ignored), window [n_fft] synthesis window centred/zero padded, inv_env
 * [n_fft + hop*(T-1)] = 1 / overlap-added window^2.  y [nsig][Lout] receives samples
 * n_fft/2 .. n_fft/2+Lout-1 of the overlap-add (Lout <= hop*(T-1); crop of aero.py:513 included). */
int aero_istft_fwd(const float* spec, int32_t nsig, int32_t F, int32_t T, int32_t n_fft, int32_t hop,
                   const float* window, const float* inv_env, float* y, int32_t Lout, void* stream);
/* ... with the spectrogram rows at a pitch of `pitch` frames (pitch * 8 bytes per bin row) and frame 0 in column t_off: the layout in which
 * every 16-frame run the kernel reads is one 128-byte cache line.  aero_istft_pitch reports (pitch, t_off) for a geometry -- (T, 0) where
 * the plain layout is the only one supported; a producer (aero_convtr_tail_finish_pitched) writes rows accordingly.  The pad columns are
 * never read. */
int aero_istft_pitch(int32_t n_fft, int32_t hop, int32_t T, int32_t* pitch, int32_t* t_off);
int aero_istft_pitched_fwd(const float* spec, int32_t nsig, int32_t F, int32_t T, int32_t pitch, int32_t t_off, int32_t n_fft, int32_t hop,
                           const float* window, const float* inv_env, float* y, int32_t Lout, void* stream);

/* K3/K4/K5/K6/K9 and every 1x1 -- one implicit-GEMM MFMA kernel family replaces nn.Conv2d
 * (aero.py:89,95,101,179), nn.ConvTranspose2d (aero.py:172), nn.Conv1d (modules.py:206,209,
 * 72-79,291-292) and nn.Linear (modules.py:29).  Output row (b,fo), step t, channel m:
 *   out = epilogue( bias[m] + sum_{tap j} sum_{c} W[wset][m][j][c] * in[b][fi(fo,j)][t+dt[j]][c] )
 * with in = concat(src0, src1) on channels, zero outside [0,Fin)x[0,T);
 *   transposed == 0:  fi = fo*fstride + df[j],            wset = 0
 *   transposed == 1:  fi = fo/fstride + df[j] (floor),    wset = fo % fstride   (conv-transpose
 *                     rewritten as fstride interleaved ordinary convolutions).
 * weight: fp16 [nwset][Mpad][ntaps*Cp], Cp = roundup(C0+C1,32), Mpad = roundup(M,128), zero padded.
 * src0 == NULL with C0 > 0 means "C0 channels of zeros" (first decoder input, aero.py:484).
 * Epilogue order: +bias, [GroupNorm, see stat_mode], act (GLU pairs rows 2u,2u+1 -> channel u, Mout = M/2), +res, +post_add[fo][.],
 * per-b affine v*batch_scale[b]+batch_shift[b]; store fp16 or fp32 at row f = fo - dst_f_off when
 * 0 <= f < dst_F (the trim of aero.py:207-209). */
typedef struct {
    const void* src0; int64_t s0_b, s0_f, s0_t; int32_t C0;
    const void* src1; int64_t s1_b, s1_f, s1_t; int32_t C1;
    const void* weight;
    const float* bias;
    void* dst; int64_t d_b, d_f, d_t;
    int32_t dst_f32, dst_f_off, dst_F;
    int32_t B, Fin, Fout, T, M;
    int32_t transposed, fstride;
    int32_t ntaps; int32_t df[9]; int32_t dt[9];
    int32_t act;
    const void* res; int64_t r_b, r_f, r_t;
    const float* post_add;
    const float* batch_scale; const float* batch_shift;
    /* GroupNorm fused into the epilogue (all optional; stat_mode 0 = off).  Statistics of v = conv + bias per
     * (item, group): item = b (stat_per_row 0) or the (b, fo) row (1); group = m / (M/stat_G).  stats: fp64
     * [items*stat_G][2] sum / sum of squares, same format as aero_norm_stats (caller zeroes it).
     *   1: accumulate, then the usual epilogue/store      2: accumulate only, nothing is stored
     *   3: v <- (v-mean)*rstd*gamma[m]+beta[m] from stats / stat_count, then act; GLU output u is multiplied by
     *      layer_scale[u] when given (DConv tail: modules.py:209-210,141,244 via a recompute pair of launches). */
    double* stats; double stat_count;
    int32_t stat_mode, stat_G, stat_per_row; float stat_eps;
    const float* gamma; const float* beta; const float* layer_scale;
    /* Row scatter (0 = off): a ConvTranspose2d along frequency (aero.py:311, stride s) computed from the INPUT side.  The
     * weight image stacks the s residue classes (M = s * scatter_M rows, row r*scatter_M + m), the launch is a plain conv
     * over the Fout = ceil(rows / s) input-aligned rows q (taps df = 0, -1, ...), and output channel block r of row q goes
     * to destination frequency row q*scatter_stride + r - scatter_off when that lies in [0, scatter_F).  One pass over the
     * source rows feeds all s residue classes.  Needs fp16 output with scatter_M % 8 == 0; no statistics, residual,
     * frequency embedding or per-item affine (AERO_ERR_UNSUPPORTED otherwise). */
    int32_t scatter_M, scatter_stride, scatter_off, scatter_F;
    /* Optional second image of the SAME weights, pre-tiled for the software-pipelined wide-contraction kernel (0 = none):
     * fp16 [nwset][M / tiled_bm][ntaps * Cp / 32][tiled_bm * 32], i.e. one contiguous block per (M-tile, 32-channel
     * K-chunk kc = tap * Cp/32 + cc) holding the rows of that tile in the kernel's LDS order: 16-byte unit u = row * 4 + q
     * carries W[m0 + row][kc*32 + 8*(q ^ ((0 - (row >> 2)) & 3)) .. +8).  The kernel then copies a tile with 1-KiB
     * contiguous reads instead of 16 strided 64-byte pieces per wave instruction.  tiled_bm must equal
     * aero_conv_ring_bm(M, ntaps * Cp) or the image is ignored. */
    const void* weight_tiled; int32_t tiled_bm;
    /* Tap split (0 or 1 = off) for long, thin contractions that cannot fill the chip (FTB Conv1d over time, modules.py:290:
     * M = 48, K = 9 x 1280, 250 tiles): the nT time taps of a regular tap grid are divided into tap_split contiguous groups,
     * the launch has tap_split x the blocks, and every block STORES the partial sums of its group -- no bias, no activation --
     * into its own slab of split_acc, fp32 contiguous [tap_split][B][Fout][T][M] (every element is written; no atomics, so the
     * result does not depend on block order); dst / bias / act are ignored.  nT % tap_split == 0; no statistics / scatter /
     * residual.  aero_split_finish() adds the slabs in order and produces the fp16 activation. */
    int32_t tap_split; float* split_acc;
    /* Fused transposed-conv tail (NULL = off; round 5).  The LAST decoder layer of the U-Net (aero.py:172,189-215: GLU(rewrite 3x3) ->
     * ConvTranspose2d(C -> 2, kernel [8,1], stride [4,1]) -> trim) never needs its C-channel activation in memory: the block that has
     * finished row fi of the rewrite conv applies the 16 x C tail matrix (rows r = 2 * tap + out_channel, tail_w fp16 [16][tail_cp],
     * tail_w[2k + co][c] = W_tr[c][co][k]) to its activated tile while that is still in LDS and stores the 16 tap products of
     * every time step: taps 0..3 -> tail_lo, taps 4..7 -> tail_hi, both fp32 contiguous [B][Fout][T][8].  Output row fo = 4 fi + k - 2
     * of the transposed conv is tail_lo[fi = (fo + 2) / 4] + tail_hi[fi - 1]: aero_convtr_tail_finish() adds the two (fixed order, no
     * atomics), the bias and the per-item affine.  dst is ignored (may be NULL).  Needs act = GLU, M = 192 on the 192-row software-
     * pipelined tile (one M-tile: the block holds every channel), tail_cp = M / 2 rounded to 32, no statistics / residual / embedding /
     * affine / scatter; AERO_ERR_UNSUPPORTED otherwise. */
    const void* tail_w; float* tail_lo; float* tail_hi; int32_t tail_cp;
} aero_conv_desc;
int aero_conv_fwd(const aero_conv_desc* d, void* stream);
/* dst fp16 [npos][M] (contiguous) = act(sum_s acc[s][npos][M] + bias[M]), s = 0..nsplit-1 in that order; act NONE / RELU / GELU;
 * bias may be NULL */
int aero_split_finish(const float* acc, int32_t nsplit, const float* bias, int32_t act, void* dst, int64_t npos, int32_t M, void* stream);
/* rows per block (16/32/48/64/128) of the kernel instantiation aero_conv_fwd picks for M output channels */
/* second half of the fused transposed-conv tail (aero_conv_desc.tail_w): dst fp32 [B][dst_F][T][2] (= complex64 [B][dst_F][T]),
 * dst[b][fo][t][co] = (lo[b][(fo + pad) / 4][t][2 k + co] + hi[b][(fo + pad) / 4 - 1][t][2 k + co] + bias[co]) * scale[b] + shift[b]
 * with k = (fo + pad) % 4, terms whose source row lies outside [0, Fin) dropped (aero.py:207-214 with pad = 2, dst_F = 4 Fin; the
 * per-item affine is the de-normalisation of aero.py:497-498); bias / scale / shift may be NULL */
int aero_convtr_tail_finish(const float* lo, const float* hi, const float* bias, const float* scale, const float* shift, float* dst,
                            int32_t B, int32_t Fin, int32_t T, int32_t dst_F, int32_t pad, void* stream);
/* ... writing dst rows at a pitch of `pitch` time steps starting at column t_off (dst fp32 [B][dst_F][pitch][2]): see aero_istft_pitch */
int aero_convtr_tail_finish_pitched(const float* lo, const float* hi, const float* bias, const float* scale, const float* shift, float* dst,
                                    int32_t B, int32_t Fin, int32_t T, int32_t dst_F, int32_t pad, int32_t pitch, int32_t t_off, void* stream);
int aero_conv_tile_m(int32_t M);
/* rows per block (256/128/64) of the software-pipelined kernel for a contraction with M rows and Ktot = ntaps * Cp
 * columns, or 0 if that kernel does not take the shape: the tile height `weight_tiled` must be prepared for */
int aero_conv_ring_bm(int32_t M, int32_t Ktot);
/* the kernel instantiation aero_conv_fwd would launch for this descriptor, as rocprofv3 prints it (e.g.
 * "aero_conv_glds_kernel<4, 2, 64, false>"); nothing is launched.  name must hold >= 96 bytes.  Profiling labels only. */
int aero_conv_kernel_name(const aero_conv_desc* d, char* name, int32_t cap);

/* K7+K8 -- nn.GroupNorm (aero.py:56,148; modules.py:189) followed by GELU / GLU(+LayerScale
 * +residual) / Snake (aero.py:127,133,198,214; modules.py:141,232-236,244; snake.py:67).
 * per_row == 0: statistics per (b, group) over (f, t, c in group)   [GroupNorm on B,C,F,T]
 * per_row == 1: statistics per (b, f) row and group                 [GroupNorm on B*F,C,T]
 * per_row == 2: statistics per group over the WHOLE batch (b, f, t, c in group): with G == C this is nn.BatchNorm2d /
 *               BatchNorm1d in training mode (the FTB's three BatchNorms, modules.py:287,293,300); stats has G pairs
 * aero_norm_stats ADDS sum and sum of squares to stats[(item*G+g)*2 + {0,1}] (fp64; the caller zeroes them);
 * aero_norm_apply derives mean / biased variance from them with stat_count = elements per (item, group)
 * (lets the statistics cover more rows than are output: the trim of aero.py:206-209) and computes
 *   y = act((x-mean)*rstd*gamma[c]+beta[c]); GLU: y[c] = a[c]*sigmoid(a[c+C/2]) * layer_scale[c];
 *   Snake: y + sin^2(a_f y)/a_f;  then + res.  stats == NULL in apply means identity norm. */
typedef struct {
    const void* src; int64_t s_b, s_f, s_t;
    int32_t B, F, T, C, G, per_row;
    float eps;
    double* stats; double stat_count;
    const float* gamma; const float* beta;
    int32_t act;
    const float* snake_a;
    const float* layer_scale;
    const void* res; int64_t r_b, r_f, r_t;
    void* dst; int64_t d_b, d_f, d_t;
} aero_norm_desc;
int aero_norm_stats(const aero_norm_desc* d, void* stream);
int aero_norm_apply(const aero_norm_desc* d, void* stream);

/* GroupNorm(1, M) statistics of a POINTWISE conv's output without running the conv (DConv tail, modules.py:209-210):
 * for y_t = W x_t + b over the T steps of one (b, f) row,
 *     sum_t sum_m y       = g1 . S[:, C]                 S = sum_t x'_t x'_t^T,  x' = (x, 1)   ((C+1) x (C+1) Gram matrix)
 *     sum_t sum_m y^2     = sum_ij G[i][j] S[i][j]       G = W'^T W',  W' = [W b],  g1 = sum_m W'[m]
 * so the kernel reads only the C-channel input rows (C = M/8 here), forms S with MFMAs and contracts it with the
 * caller's fp64 tables G [Cp][Cp] and g1 [Cp] (Cp = (C+1) rounded up to 16, zero padded).  One block per row, no
 * atomics: stats[(b*F + f)*2 + {0,1}] are WRITTEN (sum, sum of squares), the format aero_norm_apply / aero_conv_fwd
 * stat_mode 3 read with stat_count = T*M.  x fp16 [B,F,T,C] channels-last (channel stride 1, strides in elements). */
typedef struct aero_gram_desc {
    const void* x; int64_t s_b, s_f, s_t;
    int32_t B, F, T, C;
    const double* G; const double* g1;
    double* stats;
} aero_gram_desc;
int aero_gram_stats(const aero_gram_desc* d, void* stream);

/* K9' -- a pointwise (1x1) convolution with a SHORT contraction (C <= 96; activation NONE / RELU / GLU; statistics with GLU only) and a wide output, as ONE streaming pass (k_pw.h): the tail of a
 * DConv layer behind a BLSTM / LocalState (modules.py:240-247: Conv1d(hidden, 2C, 1) -> GroupNorm(1, 2C) -> GLU -> LayerScale, + x) and the
 * encoder's rewrite conv + GLU (+ frequency embedding) where no GroupNorm sits between them (aero.py:133, 475-480).
 *   x fp16 [B][F][T][C] channels-last; dst fp16 [B][F][T][Mout], Mout = M/2 with AERO_ACT_GLU (rows 2u, 2u+1 = value, gate) else M.
 *   v[m] = W[m,:] . x + bias[m];  if stats: v = (v - mean) * rstd * gamma[m] + beta[m] with {sum, sum of squares} of ROW b*F + f at
 *   stats[2*(b*F+f)], mean = sum / stat_count (the sums aero_gram_stats / aero_conv_fwd stat_mode 2 leave);  y = act(v) * layer_scale[c]
 *   + res[b,f,t,c] + post_add[f][c]  (each optional).
 *   wimg: fp16 image of W packed by the host (aero_amd/pack.py: pw_image): [chunk][2][GW][4][KS][64 lanes][8], KS = aero_pw_ksteps(C),
 *   GW = aero_pw_rows(C, M) / 128 rows-per-wave groups; element (chunk, wm, g, j, ks, lane, e) = W[r][k],
 *   r = 128*GW*chunk + 64*(GW*wm + g) + 16*((lane & 15) >> 2) + 4*j + (lane & 3),  k = 32*ks + 8*(lane >> 4) + e  (zero outside M x C):
 *   the row permutation that makes a lane's accumulators 16 consecutive rows of one time step (no transpose before the store). */
typedef struct aero_pw_desc {
    const void* x; int64_t x_b, x_f, x_t; int32_t C;
    const void* wimg; const float* bias;
    const double* stats; double stat_count; float stat_eps;
    const float* gamma; const float* beta; const float* layer_scale;
    const float* post_add;
    const void* res; int64_t r_b, r_f, r_t;
    void* dst; int64_t d_b, d_f, d_t;
    int32_t B, F, T, M, act;
    const void* x1; int64_t x1_b, x1_f, x1_t; int32_t C0;     /* optional second source: input channels C0 .. C-1 are x1's 0 .. C-C0-1 (cat([x, x1], 1)) */
} aero_pw_desc;
int aero_pw_fwd(const aero_pw_desc* d, void* stream);
/* the FTB's channel squeeze (modules.py:284-288, 307-309): y[m] = act(W[m,:] . x[b,f,t,:] + bias[m]) for M <= rp <= 8 output channels, written as the
 * image the FTB's Conv1d over time reads: dst fp16 [B][T][F*rp], element (b, f, t, m) at (b*T + t)*F*rp + f*rp + m (channels m >= M of a slot
 * are not written).  wimg: fp16 [KS = ceil(C/32)][64 lanes][8]: lane l holds W[l & 15][32*ks + 8*(l >> 4) ..] (rows >= M and columns >= C zero). */
int aero_squeeze_fwd(const void* x, int64_t x_b, int64_t x_f, int64_t x_t, const void* wimg, const float* bias, void* dst, int32_t B, int32_t F,
                     int32_t T, int32_t C, int32_t M, int32_t rp, int32_t act, void* stream);
/* conv rows one block covers (128 * GW) for a C -> M pointwise conv, 0 if the geometry is not served (C > 96, C % 8, M % 16) */
int aero_pw_rows(int32_t C, int32_t M);
/* k-steps KS of the weight image for C <= 96 input channels (ceil(C/32): the image is zero padded to 32*KS columns; the fragments stay in
 * registers); 0 for wider inputs (their weights-in-LDS form measured slower than aero_conv_fwd and is no longer built) */
int aero_pw_ksteps(int32_t C);

/* K10 -- the recurrent part of nn.LSTM(bidirectional) inside BLSTM (modules.py:28,46), both
 * directions of ONE layer per call; the input projection is an aero_conv_fwd 1x1.
 * xproj fp16 [npos][8H]: gate pre-activations incl. both biases, channel = dir*4H + 4*j + gate
 * (gate order i,f,g,o).  xbias [8H] fp16: the pre-activation of a zero input (padded steps of
 * `unfold`, models/utils.py:29-31).  whh fp16 [2][MP][KP] rows 4*j+gate, zero padded; MP, KP from
 * aero_lstm_geometry(H).  Sequence s, step tau reads position
 *   in_mode 0: s*W + tau;   in_mode 1 (framed view of [R][T]): r = s/nframes, k = s%nframes,
 *   t = k*S + tau, position r*T + t if t < T else xbias.
 * Output h (fp16, channel dir*H + j):
 *   out_mode 0: out[(s*W + tau)*2H + ..];  out_mode 1: stitched as modules.py:49-62 into
 *   out[(r*T + t)*2H + ..] keeping tau in [0,W-S/2) / [S/2,W-S/2) / [S/2,W) of first/middle/last frame. */
typedef struct {
    const void* xproj; const void* xbias; const void* whh; void* out;
    int32_t H, nseq, W, in_mode, out_mode, nframes, S, T;
    /* fused input projection (wih != NULL; xproj/xbias unused): x fp16 [npos][x_pitch] holds the in_ch input channels of
     * every position (same position indexing as xproj; padded steps are zero inputs), wih fp16 [2][MP][KPI] rows 4*j+gate
     * zero padded (KPI from aero_lstm_geometry_in), bias fp32 [2][4H] = b_ih + b_hh in the same row order. */
    const void* x; const void* wih; const float* bias;
    int32_t in_ch, x_pitch;
    /* training-mode forward (both NULL otherwise): per (block of 16 sequences ib = s/16, direction, step tau) the gate
     * activations i,f,g,o as fp16 [H][16][4] at save_gates + (((ib*2 + dir)*W + tau)*H*64) and the cell state c_tau as fp32
     * [H][16] at save_c + (((ib*2 + dir)*W + tau)*H*16): what aero_lstm_bwd reads.  Runs the step-wise kernel. */
    void* save_gates; float* save_c;
    /* round 6 -- 0: sequence s is frame s % nframes of row s / nframes (the order of `unfold` + reshape, modules.py:49-51); 1: frame-major,
     * s = k * (nseq / nframes) + r.  Both framed layers of a BLSTM must use the same order (the layer-1 output is indexed by s).  With
     * frame-major sequences a block's 16 sequences share a frame, and the stitching layer (out_mode 1, no save buffers) stops at the last
     * step its frame keeps -- a quarter of a middle frame's steps in either direction; the output is bit-identical either way. */
    int32_t frame_major;
} aero_lstm_desc;
/* AERO_ERR_UNSUPPORTED ("tensor too large for 32-bit offsets") when the input holds 2^31 rows of W steps or the output 2^31 elements or
 * more: the kernel keeps per-thread row indices / element offsets in 32 bits (a 64-bit pair per entry spilled registers). */
int aero_lstm_fwd(const aero_lstm_desc* d, void* stream);
/* padded W_hh geometry the kernel instantiation for hidden size H expects: MP rows, KP columns */
int aero_lstm_geometry(int32_t H, int32_t* MP, int32_t* KP);
/* padded column count KPI of wih for the fused projection of in_ch inputs, or a negative code if not supported */
int aero_lstm_geometry_in(int32_t H, int32_t in_ch, int32_t* KPI);

/* K11 -- LocalState attention core (modules.py:101-124).  qkvd fp16 [R][T][ld] holds per position
 * query | key | content (C each) | decay logits (heads*ndecay), produced by one aero_conv_fwd.
 * out fp16 [R][T][C] = softmax_t( K_t.Q_s/sqrt(C/heads) - |t-s| * sum_f (f+1) sigmoid(d_fs)/(2 sqrt(ndecay)),
 * diagonal forced to -100 ) applied to the content; the proj conv + residual (modules.py:127) is
 * an aero_conv_fwd with `res`. */
typedef struct {
    const void* qkvd; int64_t ld; void* out;
    int32_t R, T, C, heads, ndecay;
} aero_attn_desc;
int aero_localstate_fwd(const aero_attn_desc* d, void* stream);

/* K12 (middle) -- FTB frequency mixing (modules.py:314-320): dst[b,fo,t,c] =
 * gate[b,t,c] * sum_fi w[fo][fi] * x[b,fi,t,c]   (the gate does not depend on fi so it factors out).
 * x, dst fp16 contiguous [B][F][T][C]; gate fp16 [B][T][C]; w fp16 [roundup(F,128)][roundup(F,32)]. */
typedef struct {
    const void* x; const void* w; const void* gate; void* dst;
    int32_t B, F, T, C;
} aero_freqfc_desc;
int aero_freqfc_fwd(const aero_freqfc_desc* d, void* stream);

/* K3+K12 fused for encoder 0 -- pre_conv (aero.py:89,120) followed by the FTB (modules.py:304-325) with eval-mode
 * BatchNorm, algebraically collapsed onto the 2-channel normalised spectrogram xn (see aero_amd/csrc/k_ftb.h):
 *   att[c] = gate[b,t,c] * (p0[c]*u_re + p1[c]*u_im + pb[c]*rs[f]),      u = freq_fc applied to xn (aero_freqfc_fwd)
 *   dst[m] = relu( sum_c w2a[m][c]*att[c] + a_re[m]*re + a_im[m]*im + bias[m] )
 * xn, u fp16 [B][F][T][2]; gate fp16 [B][T][C]; w2a fp16 [>=roundup(C,16)][roundup(C,32)] zero padded;
 * p0,p1,pb,a_re,a_im,bias fp32 [C]; rs fp32 [F]; dst fp16 [B][F][T][C].  C multiple of 8, <= 64. */
typedef struct {
    const void* xn; const void* u; const void* gate; const void* w2a;
    const float* p0; const float* p1; const float* pb; const float* rs;
    const float* a_re; const float* a_im; const float* bias;
    void* dst;
    int32_t B, F, T, C;
} aero_ftb_first_desc;
int aero_ftb_first_fwd(const aero_ftb_first_desc* d, void* stream);

/* K3+K12+K3' fused for encoder 0 -- pre_conv + FTB (as aero_ftb_first_fwd) AND the layer's strided frequency conv with
 * its activation (aero.py:95,124-127) in one pass; neither the C-channel FTB output nor anything else between the
 * 2-channel spectrogram and the conv output touches HBM (aero_amd/csrc/k_enc0.h).  The FTB output is evaluated as
 *   x0[b,f,t,c] = relu( u_re*G0 + u_im*G1 + rs[f]*G2 + a_re[c]*re + a_im[c]*im + bias_f[c] ),
 * g fp16 [B][T][3][C] = (G0 | G1 | G2)[b,t,c] = sum_c' (w2a[c][c'] * {p0,p1,pb}[c']) * gate[b,t,c']  (a 1x1 conv of the gate,
 * computed by the caller with aero_conv_fwd), xn / u fp16 [B][F][T][2] as for aero_ftb_first_fwd, rs fp32 [F].
 *   dst[b,fo,t,m] = act( bias_c[m] + sum_{j<ktaps} sum_c wc[m][j*Cp + c] * x0[b, fo*stride - pad + j, t, c] )   (zero outside [0,F))
 * wc: the fp16 image aero_conv_fwd takes for that conv ([>= M rows][ktaps*Cp], Cp = roundup(C,32)); dst fp16 contiguous
 * [B][Fo][T][M].  C multiple of 8 in [8,64], M multiple of 16 in [16,64]. */
typedef struct {
    const void* xn; const void* u; const void* g;
    const float* rs; const float* a_re; const float* a_im; const float* bias_f;
    const void* wc; const float* bias_c;
    void* dst;
    int32_t B, F, T, C, M, Fo, ktaps, stride, pad, act;
} aero_enc0_desc;
int aero_enc0_fwd(const aero_enc0_desc* d, void* stream);

/* K14: a whole DConv residual branch (modules.py:221-249) for the layers WITHOUT BLSTM / LocalState, one launch, the
 * activations read from and written to HBM once (aero_amd/csrc/k_dconv.h).  x, y: fp16 [R][T][C] rows (R = B*F items of
 * the reference's [B*F, C, T] view; y may alias x).  For each of `depth` layers, in place on the row:
 *   h = conv1d(x; w1, b1, kernel 3, dilation, padding = dilation);  h = act(GroupNorm(1, hidden)(h; g1, be1))
 *   v = conv1d(h; w2, b2, kernel 1) [2C];  v = GroupNorm(1, 2C)(v; g2, be2);  x = x + scale * GLU(v)
 * act: RELU / GELU / SNAKE (snake_a fp32 [F], a of frequency row r % F: modules.py:232-236, snake.py:67) / NONE.
 * w1: fp16 [HP][K1p], element (j, tap*C + c), taps t-d, t, t+d, HP = roundup(hidden,16), K1p = roundup(3C,32), zero padded.
 * w2: fp16 [C/8][HP/16][64][4]: conv2 weights with GLU-interleaved rows (a0, b0, a1, b1, ...: row 2i = output channel i, row
 * 2i+1 = gate C+i) as MFMA 16x16x16 A fragments in lane order: element [mf][ks][lane][e] = W2[mf*16 + lane%16][ks*16 + (lane/16)*4 + e].
 * consts: fp32 [3*HP + 7*C] = b1[HP] | g1[HP] | be1[HP] | b2[2C] | g2[2C] | be2[2C] | scale[C]  (entries >= hidden of the first
 * three zero; b2 / g2 / be2 in the row order of w2; without a norm g = 1, be = 0 and norm1 / norm2 = 0: statistics skipped).
 * C in {16, 32, 48, 64, 96, 128}, hidden % 4 == 0, hidden <= 32, T <= 1024 and the row must fit the LDS:
 * aero_dconv_row_fits(T, C, hidden, largest dilation) == 1.  All pointers 16-byte aligned. */
#define AERO_DCONV_MAX_DEPTH 4
typedef struct {
    const void* w1; const void* w2; const float* consts; const float* snake_a;
    int32_t dilation, norm1, norm2, reserved;
} aero_dconv_layer;
typedef struct {
    const void* x; void* y;
    int32_t R, T, C, hidden, depth, act, F;
    float eps;
    aero_dconv_layer layer[AERO_DCONV_MAX_DEPTH];
} aero_dconv_desc;
int aero_dconv_row_fwd(const aero_dconv_desc* d, void* stream);
int aero_dconv_row_fits(int T, int C, int hidden, int max_dilation);

/* Optimizer step of the generator (train.py:83: torch.optim.Adam(params, lr, betas=(0.9, beta2)); solver.py:602-605), fused over a
 * flat buffer: p, g, m, v fp32 [n], 16-byte aligned (parameters and gradients are views into p and g).  step >= 1 is the
 * 1-based step count of the bias corrections; every gradient is multiplied by grad_scale first (1 / world size after a summing
 * all-reduce, or 1).  Same arithmetic and operation order as torch's Adam without amsgrad / weight decay. */
int aero_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   int32_t step, float grad_scale, void* stream);
/* the same step with the bias corrections bc = {1 - beta1^t, sqrt(1 - beta2^t)} read from DEVICE memory: a training step captured as
 * a HIP graph is replayed with the step count advancing (the host uploads the pair before each replay; aero_amd.optim.FlatAdam) */
int aero_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                       const float* bc, float grad_scale, void* stream);

/* ---- backward (loss.backward() of src/solver.py:602-605 through the modules of aero.py / modules.py).
 * Data gradients of the convolutions are aero_conv_fwd calls with re-packed weights (aero_amd/backward.py). */

/* weight gradient of any convolution of the aero_conv_desc family (nn.Conv2d aero.py:86,95,172; nn.Conv1d modules.py:206-210;
 * nn.ConvTranspose2d aero.py:179 with the roles of x and dy swapped):
 *   dw[j][m][c] += sum_{b, fo, t} dy[b, fo, t, m] * x[b, fo*fstride + df[j], t + dt[j], c]      (x = 0 outside its rows / steps)
 *   db[m]       += sum_{b, fo, t} dy[b, fo, t, m]                                                (db may be NULL)
 * dy fp16 [B, Fout, T, M], x fp16 [B, Fin, T, C] (element strides; M, C and strides multiples of 8); dw fp32 [ntaps][M][C] and
 * db fp32 [M] are ACCUMULATED (the caller zeroes them).  The positions (b, fo, t) are cut into chunks -- runs of 64-step segments in
 * (b, fo, t) order -- for parallelism: with slabs == NULL the chunks add their partial tiles (and bias sums) with fp32 atomics (sum
 * order not fixed); with a workspace slabs fp32 [nslab][ntaps*M*C + (db ? M : 0)] each chunk (at most nslab of them) stores its
 * partial tile, followed by its bias partial, with plain stores and a second kernel adds them to dw / db in a fixed order --
 * deterministic, and faster (no scattered 4-byte atomics).  aero_conv_wgrad_chunks: the number of chunks the launch would like for
 * this geometry (nrows = B * Fout): the nslab to allocate; fewer is legal (longer chunks). */
typedef struct {
    const void* dy; int64_t dy_b, dy_f, dy_t;
    const void* x; int64_t x_b, x_f, x_t;
    float* dw; float* db;
    int32_t B, Fin, Fout, T, M, C, ntaps, fstride;
    int32_t df[9], dt[9];
    float* slabs; int32_t nslab;
    /* with slabs only.  store != 0: dw / db are WRITTEN (dw = sum of the chunk partials) instead of added to -- no zero fill needed.
     * dw_layout 1: dw is addressed in the layout of an nn.Conv2d / Conv1d / ConvTranspose2d weight whose taps are in raster order,
     * dw[((m * dw_rowlen + dw_coff + c) * ntaps) + j]  (dw_rowlen 0 = C: the c-extent of the destination rows; dw_coff: first column --
     * the two sources of a concatenated input write the two column ranges of one weight); 0: [ntaps][M][C] as above. */
    int32_t store, dw_layout, dw_rowlen, dw_coff;
} aero_wgrad_desc;
int aero_conv_wgrad(const aero_wgrad_desc* d, void* stream);
int aero_conv_wgrad_chunks(int32_t M, int32_t C, int32_t ntaps, int32_t nrows, int32_t T);

/* nn.GroupNorm + GELU / GLU(+LayerScale) backward (aero.py:56,127,133,148,198,214; modules.py:189,232-236).  x is the saved INPUT
 * of the norm (fp16 [B,F,T,C]), stats / stat_count the forward statistics (aero_norm_stats / the conv epilogues), dy the gradient
 * of the activation output (fp16 [B,F,T,C] or [B,F,T,C/2] for GLU).  aero_norm_bwd_reduce ADDS the group sums
 * (sum dxh, sum dxh*xh: fp64 pairs, laid out as stats) to `sums` and the parameter gradients to dgamma / dbeta [C] /
 * dlayer_scale [C/2] (fp32, may be NULL; the caller zeroes all of them); aero_norm_bwd_apply then writes
 * dx = rstd * (dxh - S1/N - xh * S2/N) as fp16.  per_row 0 / 1 as in aero_norm_desc.  stats == NULL: identity norm (the layers
 * before norm_starts), dx = dxh.  per_row 2 (G == C): nn.BatchNorm on batch statistics (the FTB's three BatchNorms in training mode,
 * modules.py:287,293,300): the per-channel sums are gamma * dbeta and gamma * dgamma, so `sums` is not used and dgamma / dbeta are required.  Snake (snake.py:67): y = u + sin^2(a_f u) / a_f, with d a_f accumulated into dsnake_a. */
typedef struct {
    const void* x; int64_t x_b, x_f, x_t;
    const void* dy; int64_t dy_b, dy_f, dy_t;
    void* dx; int64_t dx_b, dx_f, dx_t;
    int32_t B, F, T, C, G, per_row;
    float eps;
    const double* stats; double stat_count;
    const float* gamma; const float* beta; const float* layer_scale;
    int32_t act;
    double* sums;
    float* dgamma; float* dbeta; float* dlayer_scale;
    const float* snake_a; float* dsnake_a;      /* Snake (act 4): a per frequency row [F], its gradient (accumulated, may be NULL) */
    double* psums;                               /* fp64 [3*C + F], zero-filled by the caller, or NULL.  Given: the reduce pass adds its
                                                  * parameter-gradient sums HERE (dgamma | dbeta | dlayer_scale | dsnake_a) and the apply pass
                                                  * adds them, rounded once, to dgamma / dbeta / dlayer_scale / dsnake_a -- results do not
                                                  * depend on the order in which blocks finish.  NULL: fp32 atomics straight into those. */
} aero_norm_bwd_desc;
int aero_norm_bwd_reduce(const aero_norm_bwd_desc* d, void* stream);
int aero_norm_bwd_apply(const aero_norm_bwd_desc* d, void* stream);

/* iSTFT backward (adjoint of aero_istft_fwd; aero.py:423-428, spec.py:30-37) = aero_stft_fwd on g = dy * inv_env, bracketed by:
 *   aero_istft_bwd_prep: s[sig][off + n] = dy[sig][n] * inv_env[n + env_off] (n < L), zero elsewhere (s: Ls samples per signal;
 *                        off = n_fft/2 + hop so that the centred STFT's reflect padding mirrors zeros, env_off = n_fft/2);
 *   aero_istft_bwd_pack: dz[sig][k][t] = c_k * spec[sig][k][t + t_off]  (c_0 = 1 with zero imaginary part, c_k = 2;
 *                        t_off = off / hop), spec / dz interleaved complex fp32 [nsig][nbins][Tsrc | T].
 * aero_amd/backward.py: istft_bwd(). */
int aero_istft_bwd_prep(const float* dy, const float* inv_env, float* s, int32_t nsig, int32_t L, int32_t Ls, int32_t off, int32_t env_off, void* stream);
int aero_istft_bwd_pack(const float* spec, float* dz, int32_t nsig, int32_t nbins, int32_t Tsrc, int32_t T, int32_t t_off, void* stream);

/* ---- the rest of one training step of the generator (solver.py:602-605; k_train.h) ------------------------------------- */

/* BPTT of one bidirectional nn.LSTM layer (modules.py:28,46) from the activations aero_lstm_fwd saved (save_gates / save_c).
 * dout fp16: gradient of the layer output h, [nseq*W][2H] (out_mode 0) or, out_mode 1, the STITCHED rows [R*T][2H] read through
 * the stitch map of modules.py:52-61 (zero outside a frame's kept range).  whh_t fp16 [2][HP][K4P]: W_hh transposed, row = hidden
 * unit j, column 4*j' + gate = W_hh[gate*H + j'][j]; HP = H rounded up to 16, K4P = aero_lstm_bwd_k4p(H), zero padded.
 * da fp16 [nseq*W][2][4H] (column 4*j + gate): the gradient of the gate PRE-activations -- every parameter / input gradient of the
 * layer is a GEMM of it (aero_conv_wgrad with x_t / h_{t-1}, aero_conv_fwd with W_ih^T): aero_amd/train.py. */
typedef struct {
    const void* dout; const void* whh_t; const void* save_gates; const float* save_c; void* da;
    int32_t H, nseq, W, out_mode, nframes, S, T;
} aero_lstm_bwd_desc;
int aero_lstm_bwd(const aero_lstm_bwd_desc* d, void* stream);
int aero_lstm_bwd_k4p(int32_t H);

/* LocalState backward (modules.py:94-127): qkvd as aero_localstate_fwd read it, out = its output O [R][T][C], dout = dL/dO;
 * dqkvd [R][T][ld] receives dQ | dK | dV | d(decay pre-activations); qstats fp32 [R][heads][T][4] is scratch
 * (log-sum-exp, O.dO, decay slope per query). */
typedef struct {
    const void* qkvd; int64_t ld; const void* out; const void* dout; void* dqkvd; float* qstats;
    int32_t R, T, C, heads, ndecay;
    float decay_scale;      /* the decay columns of dqkvd are written multiplied by this (0 = 1): they sit ~1e-6 below the others */
} aero_attn_bwd_desc;
int aero_localstate_bwd(const aero_attn_bwd_desc* d, void* stream);

/* FTB (modules.py:304-325) pieces without a convolution form.  aero_freqfc_wgrad: dw[f][f'] += sum_{b,t,c} dfc[b,f,t,c] *
 * gate[b,t,c] * x[b,f',t,c] (freq_fc weight, modules.py:296,320; slabs fp32 [nslab][F][F] scratch, partial sums added in slice
 * order).  aero_ftb_gate_bwd: with v = W_fc^T dfc (aero_freqfc_fwd on the transposed weight, gate of ones):
 * dx = add + v * gate (add may be NULL), dgate[b,t,c] = sum_f v * x (the product of modules.py:316).  All fp16 [B,F,T,C]
 * contiguous, gate / dgate [B,T,C]; T*C a multiple of 8. */
int aero_freqfc_wgrad(const void* dfc, const void* x, const void* gate, float* dw, float* slabs, int32_t nslab, int32_t B, int32_t F,
                      int32_t T, int32_t C, void* stream);
int aero_ftb_gate_bwd(const void* v, const void* x, const void* gate, const void* add, void* dx, void* dgate, int32_t B, int32_t F,
                      int32_t T, int32_t C, void* stream);

/* out[f][c] += scale * sum_{b,t} x[b,f,t,c]  (x fp16 [B,F,T,C] contiguous, out fp32): the frequency-embedding gradient (aero.py:475-480) */
int aero_sum_bt(const void* x, float* out, int32_t B, int32_t F, int32_t T, int32_t C, float scale, void* stream);

/* BLSTM framing / stitching as copies (models/utils.py:22-35, modules.py:36-62): mode 0 unfold rows [R][T][C] -> frames
 * [R*nframes][W][C] (zero beyond T), 1 its adjoint (overlap-add), 2 stitch frames -> rows, 3 its adjoint.  W = 2*S. */
int aero_frames_op(const void* src, void* dst, int32_t mode, int32_t R, int32_t T, int32_t C, int32_t nframes, int32_t W, int32_t S,
                   void* stream);

/* Spectral loss of ONE resolution (stft_loss.py:11-27,30-64,84-117) on complex64 STFTs zx (prediction) / zy (target) [n] from
 * aero_stft_fwd; power = |z|^2 * pscale (pscale = n_fft: torch.stft's scale).  sums (3 doubles): sum (ymag-xmag)^2, sum ymag^2,
 * sum |log ymag - log xmag|; part: scratch of 3*npart doubles.  aero_stft_loss_bwd: g = d(w_sc*sqrt(s0/s1)*gout[0] +
 * w_mag*s2/n*gout[1])/d zx (gout: 2 device floats or NULL = 1). */
int aero_stft_loss_sums(const float* zx, const float* zy, int64_t n, float pscale, double* part, int32_t npart, double* sums, void* stream);
int aero_stft_loss_bwd(const float* zx, const float* zy, int64_t n, float pscale, const double* sums, float w_sc, float w_mag,
                       const float* gout, float* g, void* stream);

/* Adjoint of aero_stft_fwd (centred, reflect padded, normalised; n_bins = n_fft/2 or n_fft/2+1): aero_irfft_frames turns the
 * spectrogram gradient g complex64 [nsig][nb][T] into windowed time frames fp32 [nsig][T][n_fft], aero_stft_adj_fold overlap-adds
 * them at t*hop and folds the reflect padding back: dx fp32 [nsig][L] (accumulate != 0: added to dx). */
int aero_irfft_frames(const float* g, int32_t nsig, int32_t nb, int32_t T, int32_t n_fft, const float* window, float* frames, void* stream);
int aero_stft_adj_fold(const float* frames, float* dx, int32_t nsig, int32_t T, int32_t n_fft, int32_t hop, int32_t L, int32_t accumulate,
                       void* stream);

/* Element-wise plumbing of the gradient path: dst = a + scale_b * b (fp16 [n], fp32 arithmetic);  the fp32 -> fp16 boundary with the dynamic loss scale
 * (dst = fp16(x * item_scale[item] * S), S = 2^floor(log2(target / max|x * item_scale|)), scale_out = {S, 1/S}; amax: one
 * zeroed uint32 of scratch) -- the adjoint of the de-normalisation x*std + mean of aero.py:497-498;  x *= scale[0] (fp32). */
int aero_add_f16(const void* a, const void* b, void* dst, int64_t n, float scale_b, void* stream);
int aero_scale_cast(const float* x, int32_t nitems, int64_t n_per_item, const float* item_scale, void* amax, float target, void* dst,
                    float* scale_out, void* stream);
int aero_scale_f32(float* x, int64_t n, const float* scale, void* stream);
/* running statistics of nn.BatchNorm in training mode (the FTB's BatchNorms, modules.py:287,293,300) from the fp64 channel sums
 * {sum, sum of squares} [nc][2] of `count` values per channel: running = (1 - momentum) running + momentum * {mean, unbiased variance};
 * num_batches_tracked += 1 (may be NULL). */
int aero_bn_running_update(const double* stats, int32_t nc, double count, float momentum, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, void* stream);
/* diagnostic bystander kernel (tools/istft_concurrency.py): `blocks` blocks of 256 threads with 50 KiB of LDS each check, `rounds` times,
 * that their own LDS still holds what they wrote and that loads of pattern[i] == i * 2246822519u (uint32 [npat], npat a multiple of 256)
 * return that, and that a twiddle table written by the block (sincospif per thread, as the iSTFT builds it) reads back as recomputed;
 * counters uint64 [3] += {changed LDS words, wrong loads, wrong twiddles}. */
int aero_debug_probe(const void* pattern, int32_t npat, int32_t blocks, int32_t rounds, void* counters, void* stream);
/* Weight images re-packed after an optimizer step (the training engine re-packs every step; torch's layout ops for the same bytes were
 * ~700 launches): dst[i] = table[i] >= 0 ? P[table[i]] : 0 for i < n, converted to fp16 (dst_f16 != 0) or kept fp32, where P is the
 * parameters laid end to end: ptrs = device array of nparam (<= 1024) fp32 base pointers, starts = device int32 [nparam + 1] of their
 * first flat indices.  table / dst 16-byte aligned.  aero_amd/repack.py derives the tables from the packing code itself and verifies them. */
int aero_gather_pack(const void* ptrs, const int32_t* starts, int32_t nparam, const int32_t* table, void* dst, int64_t n, int32_t dst_f16,
                     void* stream);
/* re-normalisation between stages of the backward: v = a / Sa + b / Sb (b may be NULL), S = 2^floor(log2(target / max|v|)),
 * out = fp16(v * S), scale_out = {S, 1/S}; sa / sb: the {S, 1/S} pairs of the operands (device floats, NULL = 1); amax: one zeroed
 * uint32 of scratch.  out may alias a. */
int aero_rescale_f16(const void* a, const float* sa, const void* b, const float* sb, int64_t n, void* amax, float target, void* out,
                     float* scale_out, void* stream);

/* ---- MelGAN multi-scale discriminator (src/models/discriminators.py:14-78; k_disc.h) ---------------------- */

/* grouped, strided nn.Conv1d over time (discriminators.py:17-19,29-38,42-47) on channels-last rows: x fp16 [B][Tin][Cin],
 * w fp16 [Cout][K][Cin/groups] (weight norm w = g v / |v| already applied, modules.py:10-11), bias fp32 [Cout] or NULL,
 * y fp16 [B][Tout][Cout], Tout = (Tin + 2 pad - K) / stride + 1; reflect != 0: ReflectionPad1d(pad) instead of zero padding
 * (discriminators.py:17); y = LeakyReLU_slope(conv + bias) (slope 1 = none). */
typedef struct {
    const void* x; const void* w; const float* bias; void* y;
    int32_t B, Tin, Cin, Cout, groups, K, stride, pad, reflect;
    float slope;
    /* optional MFMA form (layers with 4 input channels and 16 or 4 output channels per group, stride 4, zero padding, K <= 44:
     * aero_gconv1d_mfma_ok): the same weights as fp16 [groups][16][192], row o < Cout/groups, column 4 k + c, zero elsewhere.  NULL (or a
     * geometry the MFMA form does not take): the VALU kernel. */
    const void* w_mfma;
} aero_gconv_desc;
int aero_gconv1d_fwd(const aero_gconv_desc* d, void* stream);
int aero_gconv1d_mfma_ok(int32_t Cin, int32_t Cout, int32_t groups, int32_t K, int32_t stride, int32_t pad, int32_t reflect);
int aero_leaky_relu(void* x, int64_t n, float slope, void* stream);                      /* fp16 [n], in place */
/* nn.AvgPool1d(4, stride=2, padding=1, count_include_pad=False) (discriminators.py:70): x fp16 [B][T] -> y fp16 [B][(T-2)/2+1] */
int aero_avgpool1d(const void* x, void* y, int32_t B, int32_t T, void* stream);
/* reductions of the hinge / feature-matching losses (solver.py:489-512): out[0] += weight * sum relu(1 + sign * a[i]) (mode 0) or
 * weight * sum |a[i] - b[i]| (mode 1) -- the weight carries the 1 / numel of the mean and the loss coefficients, so a whole loss is one scalar; a, b fp16 [n]; part: scratch of npart doubles; block partials added in order (deterministic). */
int aero_loss_sum(const void* a, const void* b, int64_t n, float sign, int32_t mode, double* part, int32_t npart, double* out, double weight,
                  void* stream);

/* Backward of aero_gconv1d_fwd (solver.py:602-611).  y: the layer's post-activation output, dy: its gradient (fp16 [B][Tout][Cout]); the
 * LeakyReLU derivative is read off y.  dx (fp16 [B][Tin][Cin], may be NULL) is WRITTEN -- with reflect != 0 the contributions of the
 * mirrored positions are folded back; dw fp32 [Cout][K][Cin/groups] and db fp32 [Cout] (may be NULL) are ACCUMULATED with atomics
 * (x required for dw). */
typedef struct {
    const void* x; const void* w; const void* y; const void* dy; void* dx; float* dw; float* db;
    int32_t B, Tin, Cin, Cout, groups, K, stride, pad, reflect;
    float slope;
    /* optional MFMA form of the data gradient (same geometries as aero_gconv_desc.w_mfma): fp16 [groups][16][KD], row 4 r + c, with
     * cog = Cout/groups = 16: KD = 192, column 16 j + o; cog = 4: KD = 64, column 8 m + e = (j = 2m + 1, o = e) for e < 4, (j = 2m, o = e - 4)
     * otherwise; value W[g cog + o][c][r + 4 j] (zero where r + 4 j >= K). */
    const void* w_dgrad_mfma;
    /* optional MFMA form of the weight / bias gradient (same geometries): a workspace of nslab >= aero_gconv1d_wgrad_slabs(...) slabs
     * of (Cout * K * Cin/groups + max(Cout, 4)) floats each; position chunks store their partial sums there and a second kernel adds
     * them to dw / db in order (deterministic).  Also taken by the two edge layers (1 -> 8|16 channels, k <= 15, and 512 n -> 1 channel,
     * k <= 3, both stride 1 with pad = (K - 1) / 2), whose forward and data gradient have dedicated kernels as well; db must then have
     * room for 4 floats (added as whole float4s).  NULL: the VALU kernel with fp32 atomics. */
    float* slabs; int32_t nslab;
} aero_gconv_bwd_desc;
int aero_gconv1d_bwd(const aero_gconv_bwd_desc* d, void* stream);
/* weight-norm chain rule of one convolution (torch.nn.utils.weight_norm on dim 0, discriminators.py:10-11: w[o] = g[o] v[o] / |v[o]|):
 *   dg[o] = a <dw[o], v[o]> / |v[o]|,  dv[o] = a g[o] / |v[o]| (dw[o] - v[o] <dw[o], v[o]> / |v[o]|^2),  dbias[o] = a db[o],
 * a = inv_scale[0] * (gl ? gl[0] : 1) (device scalars: the 1/S of the fp16 gradient path, the upstream loss factor).  dw fp32, element
 * (o, c, k) at o*so + c*sc + k*sk; v / dv fp32 [Cout][cig][K]; g / dg, db / dbias fp32 [Cout] (db, dbias may be NULL).  accumulate != 0:
 * added to dg / dv / dbias (views of a flat gradient buffer), else written. */
int aero_weightnorm_bwd(const float* dw, int64_t so, int64_t sc, int64_t sk, const float* v, const float* g, const float* db, const float* inv_scale,
                        const float* gl, float* dg, float* dv, float* dbias, int32_t Cout, int32_t cig, int32_t K, int32_t accumulate, void* stream);
/* ... and its forward: w[o][:] = g[o] v[o][:] / |v[o]| (fp32 rows of L = cig * K elements) */
int aero_weightnorm_fwd(const float* v, const float* g, float* w, int32_t Cout, int32_t L, void* stream);
int aero_gconv1d_wgrad_slabs(int32_t B, int32_t Tin, int32_t Cin, int32_t Cout, int32_t groups, int32_t K, int32_t stride, int32_t pad,
                             int32_t reflect);    /* 0: the MFMA form does not take this layer */
/* gradients of the loss terms as fp16: mode 0 g = coef * sign * [1 + sign a > 0] (hinge, solver.py:489-496,508), mode 1
 * g = coef * sgn(a - b) (L1, solver.py:505), mode 2 g = a * (b > 0 ? 1 : coef) (LeakyReLU backward: a = dy, b = y, coef = slope) */
int aero_loss_grad(const void* a, const void* b, int64_t n, float sign, float coef, int32_t mode, void* g, const float* gl, void* stream);
/* (gl: optional device scalar multiplied into coef for modes 0 / 1 -- the upstream factor of the loss, read on the device) */
int aero_avgpool1d_bwd(const void* dy, void* dx, int32_t B, int32_t T, void* stream);      /* adjoint of aero_avgpool1d */

/* RCCL over xGMI behind the same ABI (replaces the NCCL process group of the reference's src/ddp/distrib.py:16-34 for a
 * host that is not PyTorch).  One communicator per process, one process per GPU (hipSetDevice first).  Rank 0 calls aero_comm_unique_id
 * and hands the 128 bytes to the other ranks by any host channel; every rank then calls aero_comm_init with the same bytes.
 * aero_allreduce_f32: in-place SUM of n floats (a flat gradient buffer; the mean's 1 / world goes into the optimizer step).
 * aero_allgather: recv = concatenation over ranks of `bytes_per_rank` bytes (clip results in rank order, distrib.py:100).
 * librccl.so is dlopen()ed at the first of these calls: no link-time dependency, AERO_ERR_UNSUPPORTED if it is not installed. */
int aero_comm_unique_id(void* id128);
int aero_comm_init(int32_t rank, int32_t world, const void* unique_id_bytes, void** comm);
int aero_allreduce_f32(void* comm, float* buf, int64_t n, void* stream);
int aero_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);
int aero_comm_destroy(void* comm);

/* HIP streams with a dispatch priority or a CU mask (round 6; the reference has no counterpart: its predict.py:76-80 / enhance.py:11-15 run
 * one forward at a time on the default stream).  aero_amd/pipeline.py issues the latency-bound segment of each batch in flight (encoder 2-3:
 * LSTM, LocalState) on such a stream so that its workgroups are dispatched ahead of / beside the other batches' MFMA tiles.
 * priority: 0 default, < 0 higher, > 0 lower (clamped to the device's range).  cu_mask (n_words 32-bit words, NULL / 0 = every CU): bit i
 * enables CU i in the runtime's enumeration -- on gfx950 round-robin over the 8 XCDs, so the low 8 n bits are n CUs of every XCD; a masked
 * stream has the default priority.  The handle is a hipStream_t (usable as the `stream` argument of every entry point above). */
int aero_stream_create(int32_t priority, const uint32_t* cu_mask, int32_t n_words, void** stream);
int aero_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AERO_HIP_H */
