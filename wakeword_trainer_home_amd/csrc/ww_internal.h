// Internal declarations shared by the libwwhip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "wwhip.h"

#define WW_NFFT 1024
#define WW_NBINS 513
#define WW_MAX_MELS 128
#define WW_MELQ_MAX_PASSES 4     // WW_MELQ_TAB (wwhip.h) ints: up to 4 passes of 16 blocks, 32 quads = 128 mel bands
#define WW_MAX_MASKS 16
#define WW_FRAMES_PER_BLOCK 16
#define WW_MAX_HOP 512
#define WW_NORM_PARTS 256

void ww_set_error(const char *fmt, ...);

#define WW_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ww_set_error(__VA_ARGS__);       \
            return (code);                   \
        }                                    \
    } while (0)

#define WW_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ww_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                       \
            return WW_E_HIP;                                                              \
        }                                                                                 \
    } while (0)

#define WW_LAUNCH_CHECK()                                                                   \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) {                                                             \
            ww_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                         \
            return WW_E_HIP;                                                                \
        }                                                                                   \
    } while (0)

// Device-resident feature tables for one (sample_rate, n_fft, n_mels, n_mfcc, f_min, f_max).
struct ww_feat_tables {
    ww_feat_cfg cfg;       // key
    float *window;         // (1024) periodic Hann
    float2 *twiddle;       // (1024) exp(-2 pi i j / 1024)
    int32_t *mel_start;    // (n_mels) first bin with non-zero weight
    int32_t *mel_len;      // (n_mels) band length
    int32_t *mel_off;      // (n_mels) offset of the band's weights in mel_w
    float *mel_w;          // compact band weights
    float *dct;            // (n_mfcc, n_mels) or null
    float *dct_t;          // its transpose (n_mels, n_mfcc): k_logmel's lanes take neighbouring coefficients -- neighbouring addresses
    int32_t max_len;
    int32_t n_mel_w;       // number of floats in mel_w
    // the band weights in the form k_logmel's v_mfma_f32_4x4x1 band sums read them (ww_get_feat_tables): melq_tab = WW_MELQ_TAB
    // ints (passes, quads, steps / weight offset per pass, first bin / unit per block, first unit per quad), melq_w = per
    // (pass, step, lane) one weight
    int32_t *melq_tab;
    float *melq_w;
    int32_t n_melq_w;
    ww_feat_tables *next;
};

// ---- opt-in per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
enum {
    WW_K_LOGMEL = 0, WW_K_STEM_FWD, WW_K_DW_FWD, WW_K_PW_FWD, WW_K_GAP_FWD, WW_K_HEAD_LOSS, WW_K_PW_BWD, WW_K_DW_BWD,
    WW_K_STEM_BWD, WW_K_FINALIZE, WW_K_CLIP, WW_K_AUDIO_AUG, WW_K_LINEAR, WW_K_GRU, WW_K_NHWC, WW_K_NCLASS
};
struct ww_prof_rec { int cls; hipEvent_t a, b; };
// A "sum the partials" step of a parameter-gradient kernel, left for ww_deferred_reduce_flush: dst[i] (+)= sum_z part[z*n + i]
struct ww_reduce_item { const float *part; float *dst; long n; int splits; int accumulate; };
struct ww_ctx {
    int device;
    ww_feat_tables *tables;
    float2 *tw16k;                         // (1024) exp(-2 pi i m / 16384), ww_audio.hip's FFT convolution; lazy
    double *norm_partials;                 // (WW_NORM_PARTS) block sums of squares of a large gradient bucket; lazy
    const ww_step_ctl *step_ctl;           // bound device control block (ww_ctx_bind_step_ctl) or NULL
    int logmel_wgs;                        // persistent workgroups of k_logmel (ww_ctx_set_logmel_workgroups); 0 = fill the device
    uint32_t prof_mask;
    std::vector<ww_prof_rec> *prof_recs;   // recorded, not yet collected
    std::vector<ww_prof_rec> *prof_free;   // event pairs ready for reuse
    int defer_on;                          // ww_ctx_set_deferred_reduce: partial-sum steps are queued instead of launched
    std::vector<ww_reduce_item> *deferred; // queued partial sums (ww_deferred_reduce_flush runs them as ONE launch)
};
// queue dst[i] (+)= sum_z part[z*n + i] when the context is deferring (returns true), else leave it to the caller
static inline bool ww_defer(ww_ctx *ctx, const float *part, float *dst, long n, int splits, int accumulate) {
    if (!ctx || !ctx->defer_on) return false;
    ctx->deferred->push_back(ww_reduce_item{part, dst, n, splits, accumulate});
    return true;
}
struct ww_prof_scope {   // RAII: records an event pair around the launches issued in its lifetime
    ww_ctx *ctx; hipStream_t st; ww_prof_rec r; bool on;
    ww_prof_scope(ww_ctx *c, int cls, hipStream_t s) : ctx(c), st(s), on(false) {
        if (!c || !(c->prof_mask & (1u << cls))) return;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;      // event pairs cannot time nodes of a graph being captured
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
        if (!c->prof_free->empty()) { r = c->prof_free->back(); c->prof_free->pop_back(); }
        else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        r.cls = cls;
        on = hipEventRecord(r.a, s) == hipSuccess;
    }
    ~ww_prof_scope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        ctx->prof_recs->push_back(r);
    }
};

int ww_get_feat_tables(ww_ctx *ctx, const ww_feat_cfg *cfg, ww_feat_tables **out);

// ---- reduction slabs (scratch layout of one layer call)
//   [0, 1024*128)            BN-statistics partials  (rows x [sum(64) | sumsq(64)])
//   [1024*128, +256*4096)    weight-gradient partials (rows x up to 4096 cols)
#define WW_STAT_SLAB_FLOATS (WW_MAX_PARTIALS * 128)
#define WW_DW_SLAB_ROWS 768
#define WW_DW_SLAB_FLOATS (WW_DW_SLAB_ROWS * 4096)

// finalize launchers (ww_reduce.hip)
int ww_launch_bn_fwd_finalize(const float *partials, int rows, double count, const ww_bn_t *bn, float *ss_out,
                              float *mr_out, hipStream_t st);
int ww_launch_bn_eval_ss(const ww_bn_t *bn, float *ss_out, float *mr_out, hipStream_t st);
int ww_launch_bn_bwd_finalize(const float *partials, int rows, double count, const float *gamma,
                              const float *mr, float *coef_out, float *dgamma, float *dbeta, hipStream_t st);
int ww_launch_colsum(const float *partials, int rows, int cols, float *out, hipStream_t st);
int ww_launch_bwd_finalize(const float *stat, int rows, double count, const float *gamma, const float *mr,
                           float *coef, float *dgamma, float *dbeta, const float *dwp, int cols, float *dw,
                           hipStream_t st);

// ww_linear.hip: C[i][j] (+)= sum_k A(i,k) B(j,k) + bias[j]; operand element (i,k) = p[i*s_row + k*s_k], one stride must be 1
// defer_ctx (split products only): queue the sum of the partials on it instead of launching it (ww_ctx_set_deferred_reduce)
int ww_gemm(int mode, const float *A, long a_srow, long a_sk, int a_rows, const float *B, long b_srow, long b_sk, int b_rows,
            int K, float *C, long ldc, const float *bias, int accumulate, int splits, float *part, hipStream_t st,
            ww_ctx *defer_ctx = nullptr, int a16 = 0);     // a16 (16-bit modes): A already holds elements of the matrix type
int ww_gemm_seg2(int mode, const float *A, const float *A2, long a_srow, int a_rows, const float *B, const float *B2, long b_sk,
                 int b_rows, int kseg, float *C, long ldc, int accumulate, hipStream_t st, int a16);
int ww_colsum_rows(const float *a, long rows, int cols, float *out, float *part, int chunks, hipStream_t st);
int ww_colsum_pair(const float *a, int rows, int cols, float *out0, float *out1, hipStream_t st);
int ww_colsum_rows_small(const float *a, int rows, int cols, float *out, hipStream_t st);
int ww_occupancy_grid(const void *fn, int block, size_t smem, long want, int cap);
// integer environment knob (tuning / A-B experiments); call sites cache the result in a function-local static
static inline int ww_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
// ww_conv_bwd.hip: ww_conv_stem_bwd with the stem weights (w != NULL: y_out is recomputed from x instead of read)
int ww_stem_bwd_impl(ww_ctx *ctx, int act_dtype, const void *g, const void *y_out, const float *w, const float *coef,
                     const float *x, int B, int Hin, int Win, float *dw, void *scratch, ww_stream_t stream);
// ww_ctx.hip: block sums of squares of g[0..n) into ctx->norm_partials (fixed partition -> deterministic); returns the count
int ww_launch_sumsq_partials(ww_ctx *ctx, const float *g, size_t n, int *parts_out, hipStream_t st);

// ---- Philox4x32-10 (host + device), must match oracle/philox.py bit for bit
#define WW_TAG_SPECAUG 0u
#define WW_TAG_DROPOUT 1u

__host__ __device__ inline void ww_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                          uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Philox step of a launch: the by-value argument, plus the bound control block's counter read at RUN time (so a captured
// HIP graph draws fresh masks on every replay)
__device__ __forceinline__ void ww_step_resolve(const ww_step_ctl *ctl, uint32_t lo, uint32_t hi, uint32_t &out_lo,
                                                uint32_t &out_hi) {
    uint64_t s = ((uint64_t)hi << 32) | lo;
    if (ctl) s += ctl->step;
    out_lo = (uint32_t)s;
    out_hi = (uint32_t)(s >> 32);
}

struct ww_mask_params {  // resolved SpecAugment parameters passed by value to kernels
    int32_t n_f, n_t, f_param, t_param;
    uint64_t f_thresh, t_thresh;
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
    uint64_t sample_offset;
    const ww_step_ctl *ctl;
};

// (start,width) of mask k for sample b; width==0 when the mask is not applied.
__device__ inline void ww_specaug_mask(const ww_mask_params &mp, uint32_t sample, int k, int F, int T, int &s,
                                       int &w) {
    uint32_t r[4], slo, shi;
    ww_step_resolve(mp.ctl, mp.step_lo, mp.step_hi, slo, shi);
    ww_philox(slo, shi, sample, (WW_TAG_SPECAUG << 24) | (uint32_t)k, mp.seed_lo, mp.seed_hi, r);
    const bool is_f = k < mp.n_f;
    const uint32_t dim = is_f ? (uint32_t)F : (uint32_t)T;
    const uint32_t param = is_f ? (uint32_t)mp.f_param : (uint32_t)mp.t_param;
    const uint32_t pmax = param < dim ? param : dim;
    const uint32_t ww = r[0] % (pmax + 1u);
    const uint32_t ss = r[1] % (dim - ww + 1u);
    const uint64_t thr = is_f ? mp.f_thresh : mp.t_thresh;
    s = (int)ss;
    w = ((uint64_t)r[2] < thr) ? (int)ww : 0;
}
