// aero_hip.hip -- the C-ABI shared library (include/aero_hip.h) over the gfx950 kernels.
// gfx950 only: no other offload arch, no CUDA path, no CPU fallback.  (tests/emu builds the same
// translation unit against a CPU emulation of the HIP subset, as a test double -- never loaded by
// the product.)
//
// Built by __graft_entry__.build():  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DAERO_PART=k  for k = 0..7 (no 5), IN PARALLEL,
// then one link into aero_amd/libaero_hip.so.  The library is ONE source file cut into six independently compiled parts (each
// kernel header belongs to exactly one part; a part holds the entry points over its kernels): as a single translation unit it took
// four minutes to compile; now a change to one header rebuilds one part.  Without -DAERO_PART (the emulator's build) the file is the
// whole library in one unit, as before.  EVERY device build carries `-Xclang -target-feature -Xclang -packed-fp32-ops
// -DAERO_NO_PACKED_FP32` (aero_common.h refuses to compile without the define: DESIGN.md 5b).
#include "aero_common.h"
#ifdef AERO_PART
#define AERO_IN(p) (AERO_PART == (p))
#else
#define AERO_IN(p) 1
#endif
#if AERO_IN(0)
#include "k_attn.h"
#include "k_lstm.h"
#include "k_norm.h"
#include "k_gram.h"
#include "k_optim.h"
#endif
#if AERO_IN(1)
#include "k_stft.h"
#include "k_train.h"
#endif
#if AERO_IN(2)
#include "k_conv.h"
#endif
#if AERO_IN(6)
#include "k_conv_ring.h"
#endif
#if AERO_IN(7)
#include "k_pw.h"
#endif
#if AERO_IN(3)
#include "k_ftb.h"
#include "k_enc0.h"
#include "k_dconv.h"
#endif
#if AERO_IN(4)
#include "k_bwd.h"
#include "k_disc.h"
#endif

#include <stdio.h>
#include <string.h>
#ifndef AERO_EMU
#include <cxxabi.h>
#endif

// the last error text of the calling thread: one buffer for the whole library (defined in part 0)
#ifdef AERO_PART
extern thread_local char g_err[512];
#if AERO_IN(0)
thread_local char g_err[512] = "";
#ifndef AERO_EMU
thread_local const void* aero_last_kernel_ptr_ = nullptr;
#endif
#endif
#else
static thread_local char g_err[512] = "";
#endif

static int aero_fail(int rc, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "unknown error");
    return rc;
}

static int aero_finish(int rc, const char* err) {
    if (rc != AERO_OK) return aero_fail(rc, err);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "kernel launch failed: %s", hipGetErrorString(e));
        return AERO_ERR_LAUNCH;
    }
    return AERO_OK;
}

// RCCL (dlopen()ed on first use) for the aero_comm_* entry points of part 0 -- see the end of this file
#if AERO_IN(0)
#ifndef AERO_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct AeroRccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
AeroRccl* aero_rccl() {
    static AeroRccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        r.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) r.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
            r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
        }
    }
    return (r.h && r.GetUniqueId && r.CommInitRank && r.AllReduce && r.AllGather && r.CommDestroy) ? &r : nullptr;
}
int aero_rccl_rc(AeroRccl* r, ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return AERO_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
    return AERO_ERR_LAUNCH;
}
}  // namespace
#endif
#endif

extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// part 0 -- core: version / errors / kernel names; GroupNorm, Gram statistics, LSTM, LocalState, Adam   (k_norm.h k_gram.h k_lstm.h k_attn.h k_optim.h)
#if AERO_IN(0)

// Every AERO_* variable present in the environment is a departure from the tested configuration ("all defaults"): the version string
// names them, so that a stray switch on a user's box shows up in the first line a bug report quotes (hygiene).
extern char** environ;
const char* aero_version(void) {
    static thread_local char ver[768];
#ifdef AERO_EMU
    int n = snprintf(ver, sizeof(ver), "aero_hip 0.1 (CPU emulation build -- tests only)");
#else
#ifdef AERO_NO_PACKED_FP32
    int n = snprintf(ver, sizeof(ver), "aero_hip 0.1 (gfx950; no-packed-fp32)");
#else
    int n = snprintf(ver, sizeof(ver), "aero_hip 0.1 (gfx950; PACKED-FP32 EXPERIMENT BUILD)");
#endif
#endif
    bool first = true;
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "AERO_", 5) != 0) continue;
        if (n >= (int)sizeof(ver) - 8) break;
        n += snprintf(ver + n, sizeof(ver) - n, "%s%.60s", first ? "; non-default switches: " : " ", *e);
        first = false;
    }
    return ver;
}

const char* aero_last_error(void) { return g_err; }

const char* aero_last_kernel_name(void) {
    static thread_local char name[256];
#ifdef AERO_EMU
    snprintf(name, sizeof(name), "%s", aero_last_kernel_str_);
#else
    name[0] = 0;
    if (aero_last_kernel_ptr_) {
        const char* mangled = hipKernelNameRefByPtr(aero_last_kernel_ptr_, nullptr);
        if (mangled) {
            int status = 1;
            char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
            snprintf(name, sizeof(name), "%s", (status == 0 && dem) ? dem : mangled);
            free(dem);
        }
    }
#endif
    return name;
}

int aero_norm_stats(const aero_norm_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_norm_stats_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_gram_stats(const aero_gram_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_gram_stats_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_norm_apply(const aero_norm_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_norm_apply_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_lstm_geometry(int32_t H, int32_t* MP, int32_t* KP) {
    int nw, tpw, kt;
    if (aero_lstm_pick(H, &nw, &tpw, &kt)) return aero_fail(AERO_ERR_UNSUPPORTED, "lstm: hidden size > 128 unsupported");
    if (MP) *MP = 16 * nw * tpw;
    if (KP) *KP = 32 * kt;
    return AERO_OK;
}

int aero_lstm_geometry_in(int32_t H, int32_t in_ch, int32_t* KPI) {
    const int kti = aero_lstm_kti(H, in_ch);
    if (kti <= 0) return aero_fail(AERO_ERR_UNSUPPORTED, "lstm: no fused-projection instantiation for this (H, in_ch)");
    if (KPI) *KPI = 32 * kti;
    return AERO_OK;
}

int aero_lstm_fwd(const aero_lstm_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_lstm_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_localstate_fwd(const aero_attn_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_attn_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int32_t step,
                   float grad_scale, void* stream) {
    const char* err = "";
    int rc = aero_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, const float* bc,
                       float grad_scale, void* stream) {
    const char* err = "";
    if (!bc) return aero_fail(AERO_ERR_ARG, "adam_step_dev: null bias-correction pointer");
    int rc = aero_adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, 1, grad_scale, (hipStream_t)stream, &err, bc);
    return aero_finish(rc, err);
}

#endif  // part 0

// ---------------------------------------------------------------------------------------------------------------
// part 1 -- STFT / iSTFT front-end and the training-step kernels that share its FFT core   (k_stft.h k_train.h)
#if AERO_IN(1)

int aero_stft_fwd(const float* x, int32_t nsig, int32_t L, int32_t Lp, int32_t n_fft, int32_t hop, const float* window,
                  int32_t n_bins, float* spec, int32_t T, double* stats, int32_t sig_per_item, void* stream) {
    const char* err = "";
    int rc = aero_stft_launch(x, nsig, L, Lp, n_fft, hop, window, n_bins, spec, T, stats, sig_per_item,
                              (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int64_t aero_stft_dft_table_bytes(int32_t n_fft) { return (int64_t)aero_stft_dft_tbytes((int)n_fft); }

int aero_stft_dft_table(const float* window, int32_t n_fft, int32_t win_off, void* table, void* stream) {
    const char* err = "";
    int rc = aero_stft_dft_table_launch(window, n_fft, win_off, table, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_stft_dft_fwd(const float* x, int32_t nsig, int32_t L, int32_t Lp, int32_t n_fft, int32_t hop, int32_t win_off, const void* table,
                      float* spec, int32_t T, double* stats, int32_t sig_per_item, void* stream) {
    const char* err = "";
    int rc = aero_stft_dft_launch(x, nsig, L, Lp, n_fft, hop, win_off, table, spec, T, stats, sig_per_item, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_stft_dft_norm_fwd(const float* x, int32_t nsig, int32_t L, int32_t Lp, int32_t n_fft, int32_t hop, int32_t win_off, const void* table,
                           int32_t T, double* stats, int32_t sig_per_item, void* xn, float* mean_std, void* stream) {
    const char* err = "";
    int rc = aero_stft_dft_launch(x, nsig, L, Lp, n_fft, hop, win_off, table, nullptr, T, stats, sig_per_item, (hipStream_t)stream, &err, xn, mean_std);
    return aero_finish(rc, err);
}

int aero_spec_normalize(const float* spec, int32_t nitems, int64_t n_per_item, const double* stats, void* xn,
                        float* mean_std, void* stream) {
    const char* err = "";
    int rc = aero_spec_normalize_launch(spec, nitems, n_per_item, stats, xn, mean_std, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

#ifdef AERO_ISTFT_DEBUG
// tools/dbg builds only: counters (6 x u64 on the device, zeroed by the caller) and mode bits of the iSTFT's load checks (k_stft.h)
int aero_istft_debug_set(void* counters, int32_t mode) {
    aero_istft_dbg_ptr = (unsigned long long*)counters;
    aero_istft_dbg_mode = mode;
    return AERO_OK;
}
#endif

int aero_istft_fwd(const float* spec, int32_t nsig, int32_t F, int32_t T, int32_t n_fft, int32_t hop, const float* window,
                   const float* inv_env, float* y, int32_t Lout, void* stream) {
    const char* err = "";
    int rc = aero_istft_launch(spec, nsig, F, T, n_fft, hop, window, inv_env, y, Lout, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_istft_pitch(int32_t n_fft, int32_t hop, int32_t T, int32_t* pitch, int32_t* t_off) {
    if (!pitch || !t_off || n_fft < 2 || hop < 1 || T < 1) return aero_fail(AERO_ERR_ARG, "istft_pitch: bad arguments");
    int pp, to;
    aero_istft_pitch_for(n_fft, hop, T, &pp, &to);
    *pitch = pp; *t_off = to;
    return AERO_OK;
}

int aero_istft_pitched_fwd(const float* spec, int32_t nsig, int32_t F, int32_t T, int32_t pitch, int32_t t_off, int32_t n_fft, int32_t hop,
                           const float* window, const float* inv_env, float* y, int32_t Lout, void* stream) {
    const char* err = "";
    int rc = aero_istft_launch(spec, nsig, F, T, n_fft, hop, window, inv_env, y, Lout, (hipStream_t)stream, &err, pitch, t_off);
    return aero_finish(rc, err);
}

int aero_lstm_bwd(const aero_lstm_bwd_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_lstm_bwd_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_lstm_bwd_k4p(int32_t H) { const int k = aero_lstm_bwd_kt4(H); return k < 0 ? -1 : 32 * k; }

int aero_localstate_bwd(const aero_attn_bwd_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_attn_bwd_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_freqfc_wgrad(const void* dfc, const void* x, const void* gate, float* dw, float* slabs, int32_t nslab, int32_t B, int32_t F,
                      int32_t T, int32_t C, void* stream) {
    const char* err = "";
    int rc = aero_freqfc_wgrad_launch(dfc, x, gate, dw, slabs, nslab, B, F, T, C, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_ftb_gate_bwd(const void* v, const void* x, const void* gate, const void* add, void* dx, void* dgate, int32_t B, int32_t F,
                      int32_t T, int32_t C, void* stream) {
    const char* err = "";
    int rc = aero_ftb_gate_bwd_launch(v, x, gate, add, dx, dgate, B, F, T, C, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_sum_bt(const void* x, float* out, int32_t B, int32_t F, int32_t T, int32_t C, float scale, void* stream) {
    const char* err = "";
    int rc = aero_sum_bt_launch(x, out, B, F, T, C, scale, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_frames_op(const void* src, void* dst, int32_t mode, int32_t R, int32_t T, int32_t C, int32_t nframes, int32_t W, int32_t S,
                   void* stream) {
    const char* err = "";
    int rc = aero_frames_launch(src, dst, mode, R, T, C, nframes, W, S, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_stft_loss_sums(const float* zx, const float* zy, int64_t n, float pscale, double* part, int32_t npart, double* sums, void* stream) {
    const char* err = "";
    int rc = aero_stft_loss_sums_launch(zx, zy, n, pscale, part, npart, sums, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_stft_loss_bwd(const float* zx, const float* zy, int64_t n, float pscale, const double* sums, float w_sc, float w_mag,
                       const float* gout, float* g, void* stream) {
    const char* err = "";
    int rc = aero_stft_loss_bwd_launch(zx, zy, n, pscale, sums, w_sc, w_mag, gout, g, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_irfft_frames(const float* g, int32_t nsig, int32_t nb, int32_t T, int32_t n_fft, const float* window, float* frames, void* stream) {
    const char* err = "";
    int rc = aero_irfft_frames_launch(g, nsig, nb, T, n_fft, window, frames, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_stft_adj_fold(const float* frames, float* dx, int32_t nsig, int32_t T, int32_t n_fft, int32_t hop, int32_t L, int32_t accumulate,
                       void* stream) {
    const char* err = "";
    int rc = aero_stft_adj_fold_launch(frames, dx, nsig, T, n_fft, hop, L, accumulate, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_add_f16(const void* a, const void* b, void* dst, int64_t n, float scale_b, void* stream) {
    const char* err = "";
    int rc = aero_axpy_f16_launch(a, b, dst, n, scale_b, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_scale_cast(const float* x, int32_t nitems, int64_t n_per_item, const float* item_scale, void* amax, float target, void* dst,
                    float* scale_out, void* stream) {
    const char* err = "";
    int rc = aero_scale_cast_launch(x, nitems, n_per_item, item_scale, (unsigned int*)amax, target, dst, scale_out, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_bn_running_update(const double* stats, int32_t nc, double count, float momentum, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, void* stream) {
    if (!stats || !running_mean || !running_var || nc < 1 || count < 1.0) return aero_fail(AERO_ERR_ARG, "bn_running_update: bad arguments");
    AERO_LAUNCH(aero_bn_running_kernel, dim3(1), dim3(256), (hipStream_t)stream, stats, nc, count, momentum, running_mean, running_var,
                (long long*)num_batches_tracked);
    return aero_finish(AERO_OK, "");
}

int aero_debug_probe(const void* pattern, int32_t npat, int32_t blocks, int32_t rounds, void* counters, void* stream) {
    if (!pattern || !counters || npat < 256 || blocks < 1 || rounds < 1) return aero_fail(AERO_ERR_ARG, "debug_probe: bad arguments");
    AERO_LAUNCH_DYN(aero_probe_kernel, dim3((unsigned)blocks), dim3(256), (size_t)50 * 1024, (hipStream_t)stream, (const unsigned*)pattern, npat, rounds,
                    (unsigned long long*)counters);
    return aero_finish(AERO_OK, "");
}

int aero_gather_pack(const void* ptrs, const int32_t* starts, int32_t nparam, const int32_t* table, void* dst, int64_t n, int32_t dst_f16,
                     void* stream) {
    const char* err = "";
    int rc = aero_gather_pack_launch((const float* const*)ptrs, starts, nparam, table, dst, n, dst_f16, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_scale_f32(float* x, int64_t n, const float* scale, void* stream) {
    const char* err = "";
    int rc = aero_scale_f32_launch(x, n, scale, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_rescale_f16(const void* a, const float* sa, const void* b, const float* sb, int64_t n, void* amax, float target, void* out,
                     float* scale_out, void* stream) {
    const char* err = "";
    int rc = aero_rescale_f16_launch(a, sa, b, sb, n, (unsigned int*)amax, target, out, scale_out, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

#endif  // part 1

// ---------------------------------------------------------------------------------------------------------------
// part 2 -- convolution family incl. the software-pipelined ring kernel   (k_conv.h k_conv_ring.h)
#if AERO_IN(2)

int aero_conv_fwd(const aero_conv_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_conv_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_conv_tile_m(int32_t M) { return aero_conv_pick_bm(M, (M + 127) / 128 * 128); }

int aero_conv_kernel_name(const aero_conv_desc* d, char* name, int32_t cap) {
    if (!name || cap < 96) return aero_fail(AERO_ERR_ARG, "conv_kernel_name: buffer of >= 96 bytes required");
    const char* err = "";
    name[0] = 0;
    int rc = aero_conv_launch(d, nullptr, &err, name);
    return rc == AERO_OK ? AERO_OK : aero_fail(rc, err);
}

int aero_split_finish(const float* acc, int32_t nsplit, const float* bias, int32_t act, void* dst, int64_t npos, int32_t M, void* stream) {
    const char* err = "";
    int rc = aero_split_finish_launch(acc, nsplit, bias, act, dst, npos, M, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

#endif  // part 2

// ---------------------------------------------------------------------------------------------------------------
// part 3 -- FTB, fused encoder 0, row-resident DConv   (k_ftb.h k_enc0.h k_dconv.h)
#if AERO_IN(3)

int aero_freqfc_fwd(const aero_freqfc_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_freqfc_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_dconv_row_fwd(const aero_dconv_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_dconv_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_dconv_row_fits(int T, int C, int hidden, int max_dilation) { return aero_dconv_row_fits_impl(T, C, hidden, max_dilation); }

int aero_enc0_fwd(const aero_enc0_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_enc0_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_ftb_first_fwd(const aero_ftb_first_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_ftb_first_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

#endif  // part 3

// ---------------------------------------------------------------------------------------------------------------
// part 4 -- backward of the conv / norm blocks   (k_bwd.h)
#if AERO_IN(4)

int aero_conv_wgrad(const aero_wgrad_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_conv_wgrad_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_conv_wgrad_chunks(int32_t M, int32_t C, int32_t ntaps, int32_t nrows, int32_t T) {
    if (M < 1 || C < 1 || ntaps < 1 || nrows < 1 || T < 1) return 1;
    bool big;
    int nchunk, SC;
    aero_wgrad_plan(M, C, ntaps, nrows, T, &big, &nchunk, &SC);
    return nchunk;
}

int aero_norm_bwd_reduce(const aero_norm_bwd_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_norm_bwd_launch(d, 0, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_norm_bwd_apply(const aero_norm_bwd_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_norm_bwd_launch(d, 1, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_istft_bwd_prep(const float* dy, const float* inv_env, float* s, int32_t nsig, int32_t L, int32_t Ls, int32_t off, int32_t env_off, void* stream) {
    const char* err = "";
    int rc = aero_istft_bwd_prep_launch(dy, inv_env, s, nsig, L, Ls, off, env_off, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_istft_bwd_pack(const float* spec, float* dz, int32_t nsig, int32_t nbins, int32_t Tsrc, int32_t T, int32_t t_off, void* stream) {
    const char* err = "";
    int rc = aero_istft_bwd_pack_launch(spec, dz, nsig, nbins, Tsrc, T, t_off, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

#endif  // part 4

// ---------------------------------------------------------------------------------------------------------------
// part 5 -- MelGAN critic   (k_disc.h k_gconv_mfma.h k_gconv_edge.h)
#if AERO_IN(4)

int aero_gconv1d_mfma_ok(int32_t Cin, int32_t Cout, int32_t groups, int32_t K, int32_t stride, int32_t pad, int32_t reflect) {
    return aero_gconv4_ok(Cin, Cout, groups, K, stride, pad, reflect);
}

int aero_gconv1d_wgrad_slabs(int32_t B, int32_t Tin, int32_t Cin, int32_t Cout, int32_t groups, int32_t K, int32_t stride, int32_t pad,
                             int32_t reflect) {
    if (B < 1 || Tin < 1) return 0;
    const bool c1 = aero_edge_c1_ok(Cin, Cout, groups, K, stride, pad);
    if (c1 || aero_edge_o1_ok(Cin, Cout, groups, K, stride, pad, reflect)) {
        int nb, per;
        aero_edge_wgrad_plan(c1, B, Tin, &nb, &per);
        return B * nb;
    }
    if (!aero_gconv4_ok(Cin, Cout, groups, K, stride, pad, reflect)) return 0;
    const int Tout = (Tin + 2 * pad - K) / stride + 1;
    if (Tout < 1) return 0;
    int ntile, tpc, nchunk;
    aero_gconv4_wgrad_plan(B, Tout, groups, &ntile, &tpc, &nchunk);
    return B * nchunk;
}

int aero_weightnorm_fwd(const float* v, const float* g, float* w, int32_t Cout, int32_t L, void* stream) {
    if (!v || !g || !w || Cout < 1 || L < 1) return aero_fail(AERO_ERR_ARG, "weightnorm_fwd: bad arguments");
    AERO_LAUNCH(aero_weightnorm_fwd_kernel, dim3((unsigned)Cout), dim3(256), (hipStream_t)stream, v, g, w, L);
    return aero_finish(AERO_OK, "");
}

int aero_weightnorm_bwd(const float* dw, int64_t so, int64_t sc, int64_t sk, const float* v, const float* g, const float* db, const float* inv_scale,
                        const float* gl, float* dg, float* dv, float* dbias, int32_t Cout, int32_t cig, int32_t K, int32_t accumulate, void* stream) {
    const char* err = "";
    int rc = aero_weightnorm_bwd_launch(dw, so, sc, sk, v, g, db, inv_scale, gl, dg, dv, dbias, Cout, cig, K, accumulate, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_gconv1d_fwd(const aero_gconv_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_gconv1d_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_leaky_relu(void* x, int64_t n, float slope, void* stream) {
    const char* err = "";
    int rc = aero_leaky_relu_launch(x, n, slope, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_avgpool1d(const void* x, void* y, int32_t B, int32_t T, void* stream) {
    const char* err = "";
    int rc = aero_avgpool1d_launch(x, y, B, T, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_loss_sum(const void* a, const void* b, int64_t n, float sign, int32_t mode, double* part, int32_t npart, double* out, double weight,
                  void* stream) {
    const char* err = "";
    int rc = aero_loss_sum_launch(a, b, n, sign, mode, part, npart, out, weight, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_gconv1d_bwd(const aero_gconv_bwd_desc* d, void* stream) {
    const char* err = "";
    int rc = aero_gconv1d_bwd_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_loss_grad(const void* a, const void* b, int64_t n, float sign, float coef, int32_t mode, void* g, const float* gl, void* stream) {
    const char* err = "";
    int rc = aero_loss_grad_launch(a, b, n, sign, coef, mode, g, gl, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_avgpool1d_bwd(const void* dy, void* dx, int32_t B, int32_t T, void* stream) {
    const char* err = "";
    int rc = aero_avgpool1d_bwd_launch(dy, dx, B, T, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

#endif  // part 5 (compiled with part 4: the critic's weight gradients end in k_bwd.h's slab-finish kernel)

// ---------------------------------------------------------------------------------------------------------------
// part 6 -- the software-pipelined ring kernel (k_conv_ring.h), entered from part 2's aero_conv_launch through aero_conv_ring_try
#if AERO_IN(6)

int aero_conv_ring_bm(int32_t M, int32_t Ktot) { return aero_conv_ring_pick_bm(M, Ktot); }

int aero_convtr_tail_finish(const float* lo, const float* hi, const float* bias, const float* scale, const float* shift, float* dst,
                            int32_t B, int32_t Fin, int32_t T, int32_t dst_F, int32_t pad, void* stream) {
    const char* err = "";
    int rc = aero_convtr_tail_finish_launch(lo, hi, bias, scale, shift, dst, B, Fin, T, dst_F, pad, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_convtr_tail_finish_pitched(const float* lo, const float* hi, const float* bias, const float* scale, const float* shift, float* dst,
                                    int32_t B, int32_t Fin, int32_t T, int32_t dst_F, int32_t pad, int32_t pitch, int32_t t_off, void* stream) {
    const char* err = "";
    int rc = aero_convtr_tail_finish_launch(lo, hi, bias, scale, shift, dst, B, Fin, T, dst_F, pad, (hipStream_t)stream, &err, pitch, t_off);
    return aero_finish(rc, err);
}

#endif  // part 6

// ---------------------------------------------------------------------------------------------------------------
// part 7 -- streaming pointwise conv for short contractions (k_pw.h)
#if AERO_IN(7)

int aero_pw_fwd(const aero_pw_desc* d, void* stream) {
    const char* err = "";
    if (!d) return aero_fail(AERO_ERR_ARG, "pw: null descriptor");
    int rc = aero_pw_launch(d, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_squeeze_fwd(const void* x, int64_t x_b, int64_t x_f, int64_t x_t, const void* wimg, const float* bias, void* dst, int32_t B, int32_t F,
                     int32_t T, int32_t C, int32_t M, int32_t rp, int32_t act, void* stream) {
    const char* err = "";
    int rc = aero_squeeze_launch(x, x_b, x_f, x_t, wimg, bias, dst, B, F, T, C, M, rp, act, (hipStream_t)stream, &err);
    return aero_finish(rc, err);
}

int aero_pw_rows(int32_t C, int32_t M) {
    if (C < 8 || C % 8 || C > 96 || M < 16 || M % 16) return 0;
    return 128 * aero_pw_gw(C, M);
}

int aero_pw_ksteps(int32_t C) { return (C < 8 || C > 96) ? 0 : aero_pw_ks(C); }

#endif  // part 7

// ---------------------------------------------------------------------------------------------------------------
// part 0 (cont.) -- RCCL over xGMI behind the same C ABI (aero_comm_*): what a non-Python host needs to shard clips over the
// GPUs of a node and average gradients -- a communicator per process (one process per GPU), an in-place fp32 sum and an all-gather.
// librccl.so is opened at the first aero_comm_* call (dlopen): the library has no link-time dependency on it, and a single-GPU user
// never loads it.  The Python host side does not use these entry points: torch.distributed's "nccl" backend IS RCCL (aero_amd/distrib.py).
#if AERO_IN(0)
int aero_comm_unique_id(void* id128) {
#ifdef AERO_EMU
    return aero_fail(AERO_ERR_UNSUPPORTED, "comm: not available in the emulation build");
#else
    AeroRccl* r = aero_rccl();
    if (!r) return aero_fail(AERO_ERR_UNSUPPORTED, "comm: librccl.so could not be opened");
    if (!id128) return aero_fail(AERO_ERR_ARG, "comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return aero_rccl_rc(r, r->GetUniqueId((ncclUniqueId*)id128), "ncclGetUniqueId");
#endif
}

int aero_comm_init(int32_t rank, int32_t world, const void* unique_id_bytes, void** comm) {
#ifdef AERO_EMU
    return aero_fail(AERO_ERR_UNSUPPORTED, "comm: not available in the emulation build");
#else
    AeroRccl* r = aero_rccl();
    if (!r) return aero_fail(AERO_ERR_UNSUPPORTED, "comm: librccl.so could not be opened");
    if (!unique_id_bytes || !comm || world < 1 || rank < 0 || rank >= world) return aero_fail(AERO_ERR_ARG, "comm_init: bad arguments");
    ncclUniqueId id;
    memcpy(&id, unique_id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = aero_rccl_rc(r, r->CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    *comm = (void*)c;
    return rc;
#endif
}

int aero_allreduce_f32(void* comm, float* buf, int64_t n, void* stream) {
#ifdef AERO_EMU
    return aero_fail(AERO_ERR_UNSUPPORTED, "comm: not available in the emulation build");
#else
    AeroRccl* r = aero_rccl();
    if (!r || !comm || !buf || n < 0) return aero_fail(AERO_ERR_ARG, "allreduce_f32: bad arguments (or no communicator)");
    return aero_rccl_rc(r, r->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce");
#endif
}

int aero_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
#ifdef AERO_EMU
    return aero_fail(AERO_ERR_UNSUPPORTED, "comm: not available in the emulation build");
#else
    AeroRccl* r = aero_rccl();
    if (!r || !comm || !send || !recv || bytes_per_rank < 0) return aero_fail(AERO_ERR_ARG, "allgather: bad arguments (or no communicator)");
    return aero_rccl_rc(r, r->AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllGather");
#endif
}

int aero_comm_destroy(void* comm) {
#ifdef AERO_EMU
    return aero_fail(AERO_ERR_UNSUPPORTED, "comm: not available in the emulation build");
#else
    AeroRccl* r = aero_rccl();
    if (!r || !comm) return aero_fail(AERO_ERR_ARG, "comm_destroy: no communicator");
    return aero_rccl_rc(r, r->CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
#endif
}
// Scheduling (round 6): HIP streams with a dispatch PRIORITY or a CU MASK, for hosts that want the latency-bound segment of a forward
// (encoder 2-3: LSTM, LocalState and their small launches) dispatched ahead of / beside other batches' MFMA tiles (aero_amd/pipeline.py).
// priority: 0 = default, < 0 = higher, > 0 = lower (clamped to the device's range).  cu_mask / n_words: NULL / 0 = all CUs; else bit i of the
// mask enables CU i in the runtime's enumeration (round-robin over the 8 XCDs on gfx950: the low 8 n bits = n CUs of every XCD); a masked
// stream takes the default priority (the runtime has no call for both).
int aero_stream_create(int32_t priority, const uint32_t* cu_mask, int32_t n_words, void** stream) {
#ifdef AERO_EMU
    return aero_fail(AERO_ERR_UNSUPPORTED, "stream_create: not available in the emulation build");
#else
    if (!stream || n_words < 0 || (n_words > 0 && !cu_mask)) return aero_fail(AERO_ERR_ARG, "stream_create: bad arguments");
    hipStream_t s = nullptr;
    hipError_t e;
    if (n_words > 0) {
        e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask);
    } else {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);           // (least = numerically largest = lowest priority)
        int pr = priority < greatest ? greatest : (priority > least ? least : priority);
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr);
    }
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "stream_create: %s", hipGetErrorString(e));
        return AERO_ERR_LAUNCH;
    }
    *stream = (void*)s;
    return AERO_OK;
#endif
}

int aero_stream_destroy(void* stream) {
#ifdef AERO_EMU
    return aero_fail(AERO_ERR_UNSUPPORTED, "stream_destroy: not available in the emulation build");
#else
    if (!stream) return aero_fail(AERO_ERR_ARG, "stream_destroy: null stream");
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "stream_destroy: %s", hipGetErrorString(e));
        return AERO_ERR_LAUNCH;
    }
    return AERO_OK;
#endif
}
#endif  // part 0 (cont.)

}  // extern "C"
