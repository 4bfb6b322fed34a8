// Fused clip + optimizer step on the flat parameter / gradient buckets.
// Replaces, for a HIP-backed model, the reference's per-step glue  clip_gradients -> optimizer.step()
// (src/training/trainer.py:185-193, src/training/optimizer_factory.py:165-199,446-452):
//   norm = ||g||_2 ; g *= min(1, max_norm/(norm+1e-6))           (torch.nn.utils.clip_grad_norm_)
//   skip everything below when stats->found_inf != 0             (non-finite loss / bad target / non-finite norm)
//   Adam / AdamW / SGD-Nesterov update, torch.optim's single-tensor formulas in fp32, bias corrections in double.
// The step counter lives on the device (two slots, read slot[parity], write slot[parity^1]) so that a skipped step does
// not advance it -- as in the reference, where optimizer.step() is simply not called -- without the host knowing.
// Small buckets (<= 4 096 floats) take ONE single-block launch for norm + clip + update with the gradient held in
// registers; larger ones (cnn_small has 20 546) block partial sums of squares followed by a grid-wide update.
#include "ww_internal.h"
#include <algorithm>
#include <stdlib.h>

namespace {

struct OptimArgs {
    int kind;
    float lr, beta1, beta2, eps, wd, momentum, max_norm;
    const ww_step_ctl *ctl;       // bound control block: lr and the step_state slot come from device memory (HIP graph replay)
    ww_loss_scale *ls;            // fp16 mode: gradients arrive times ls->scale[slot]; unscaled here, scale updated for slot^1
};
// torch.amp.GradScaler's update rule (unscale_ -> step -> update, src/training/trainer.py:186-193), decided on the device:
// gradients not finite -> scale *= backoff, tracker = 0; a step that was applied -> tracker + 1, and scale *= growth when it
// reaches the interval; a batch skipped for its loss / targets (before the scaler in the reference) leaves both alone
__device__ __forceinline__ void loss_scale_update(ww_loss_scale *ls, int slot, bool grads_nonfinite, bool skipped) {
    float sc = ls->scale[slot];
    int tr = ls->growth_tracker[slot];
    if (grads_nonfinite) { sc *= ls->backoff_factor; tr = 0; }
    else if (!skipped) {
        if (++tr >= ls->growth_interval) { sc *= ls->growth_factor; tr = 0; }
    }
    ls->scale[slot ^ 1] = sc;
    ls->growth_tracker[slot ^ 1] = tr;
}
// learning rate / slot of this launch: by value, or the bound control block's (read at run time)
__device__ __forceinline__ void optim_resolve(OptimArgs &a, int &parity) {
    if (a.ctl) { a.lr = a.ctl->lr; parity = a.ctl->parity & 1; }
}

__device__ __forceinline__ void optim_update(const OptimArgs &a, float &p, const float g0, float &m, float &v,
                                             const float step_size, const float bc2_sqrt) {
    float g = g0;
    if (a.kind == WW_OPT_SGD) {                       // torch.optim.SGD, nesterov=True, dampening=0
        if (a.wd != 0.f) g = fmaf(a.wd, p, g);
        if (a.momentum != 0.f) {
            m = fmaf(a.momentum, m, g);               // a zero buffer reproduces torch's "first step: buf = g"
            g = fmaf(a.momentum, m, g);
        }
        p = fmaf(-a.lr, g, p);
        return;
    }
    if (a.kind == WW_OPT_ADAMW) p *= 1.0f - a.lr * a.wd;        // decoupled decay
    else if (a.wd != 0.f) g = fmaf(a.wd, p, g);                 // Adam: L2 term in the gradient
    m = m + (1.0f - a.beta1) * (g - m);                         // exp_avg.lerp_(grad, 1 - beta1)
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;                 // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    p = p - step_size * (m / denom);                            // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__device__ __forceinline__ void bias_terms(const OptimArgs &a, long long t, float &step_size, float &bc2_sqrt) {
    if (a.kind == WW_OPT_SGD) { step_size = a.lr; bc2_sqrt = 1.f; return; }
    const double bc1 = 1.0 - pow((double)a.beta1, (double)t), bc2 = 1.0 - pow((double)a.beta2, (double)t);
    step_size = (float)((double)a.lr / bc1);
    bc2_sqrt = (float)sqrt(bc2);
}

// single block, the whole gradient bucket held in registers (n <= 32 * 1024): norm -> clip -> found_inf -> update.
// Loads are issued in unrolled batches so their latencies overlap -- one block has no other way to hide them.
constexpr int OPT_EPT = 32;
__global__ __launch_bounds__(1024) void k_clip_optim_small(OptimArgs a, float *__restrict__ p, float *__restrict__ g,
                                                           float *__restrict__ m, float *__restrict__ v, size_t n,
                                                           long long *__restrict__ step_state, int parity,
                                                           float *__restrict__ norm_out, ww_step_stats *__restrict__ stats,
                                                           ww_step_stats *__restrict__ stats_host,
                                                           ww_step_stats *__restrict__ stats_host_alt,
                                                           const float *__restrict__ found_inf_extra) {
    optim_resolve(a, parity);
    if (parity && stats_host_alt) stats_host = stats_host_alt;
    __shared__ double sh[1024];
    __shared__ float coef_sh, ss_sh, bc_sh;
    __shared__ int skip_sh;
    const float inv_scale = a.ls ? 1.0f / a.ls->scale[parity] : 1.0f;     // GradScaler.unscale_
    float gr[OPT_EPT];
#pragma unroll
    for (int k = 0; k < OPT_EPT; ++k) {
        const size_t i = threadIdx.x + (size_t)k * 1024;
        gr[k] = i < n ? g[i] * inv_scale : 0.f;
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < OPT_EPT; ++k) acc += (double)gr[k] * (double)gr[k];     // same order as k_grad_norm_clip
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(sh[0]);
        if (norm_out) *norm_out = norm;
        bool skip = !isfinite(norm) || (found_inf_extra && *found_inf_extra != 0.0f);   // extra: another rank's bad batch
        const bool loss_bad = (stats && stats->found_inf != 0.0f) || (found_inf_extra && *found_inf_extra != 0.0f);
        if (stats) {
            stats->grad_norm = norm;
            if (skip) stats->found_inf = 1.0f;
            skip = stats->found_inf != 0.0f;
            if (stats_host) *stats_host = *stats;     // the step's one record for the host, written straight to pinned memory
        }
        float c = 1.0f;
        if (a.max_norm > 0.f) { c = a.max_norm / (norm + 1e-6f); if (c > 1.0f) c = 1.0f; }
        coef_sh = c;
        const long long t0 = step_state[parity];
        step_state[parity ^ 1] = skip ? t0 : t0 + 1;
        skip_sh = skip;
        bias_terms(a, t0 + 1, ss_sh, bc_sh);
        if (a.ls) loss_scale_update(a.ls, parity, !loss_bad && !isfinite(norm), skip);
    }
    __syncthreads();
    const float c = coef_sh, step_size = ss_sh, bc2_sqrt = bc_sh;
    const bool skip = skip_sh, clip = a.max_norm > 0.f || a.ls != nullptr;     // unscaled gradients are written back
#pragma unroll
    for (int k0 = 0; k0 < OPT_EPT; k0 += 8) {
        float pr[8], mr[8], vr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t i = threadIdx.x + (size_t)(k0 + k) * 1024;
            const bool ok = i < n && !skip;
            pr[k] = ok ? p[i] : 0.f;
            mr[k] = (ok && m) ? m[i] : 0.f;
            vr[k] = (ok && v) ? v[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t i = threadIdx.x + (size_t)(k0 + k) * 1024;
            if (i >= n) continue;
            float gi = gr[k0 + k];
            if (clip) { gi *= c; g[i] = gi; }
            if (skip) continue;
            optim_update(a, pr[k], gi, mr[k], vr[k], step_size, bc2_sqrt);
            p[i] = pr[k];
            if (m) m[i] = mr[k];
            if (v) v[i] = vr[k];
        }
    }
}

// grid-wide clip + update for large buckets: every block reduces the block partials of the squared norm (k_sumsq_partials)
// in the same order, so all agree on the norm, the clip coefficient and the skip decision without another launch
__global__ __launch_bounds__(256) void k_optim_update(OptimArgs a, float *__restrict__ p, float *__restrict__ g,
                                                      float *__restrict__ m, float *__restrict__ v, size_t n,
                                                      long long *__restrict__ step_state, int parity,
                                                      const double *__restrict__ parts, int nparts,
                                                      float *__restrict__ norm_out, ww_step_stats *__restrict__ stats,
                                                      ww_step_stats *__restrict__ stats_host,
                                                      ww_step_stats *__restrict__ stats_host_alt,
                                                      const float *__restrict__ found_inf_extra, int vec) {
    optim_resolve(a, parity);
    if (parity && stats_host_alt) stats_host = stats_host_alt;
    double t = 0.0;
    for (int i = 0; i < nparts; ++i) t += parts[i];
    const float inv_scale = a.ls ? 1.0f / a.ls->scale[parity] : 1.0f;      // GradScaler.unscale_; the partial sums are of scaled values
    const float norm = (float)sqrt(t) * inv_scale;
    const bool extra = found_inf_extra && *found_inf_extra != 0.0f;
    const bool loss_bad = (stats && stats->found_inf != 0.0f) || extra;
    const bool skip = loss_bad || !isfinite(norm);
    const long long t0 = step_state[parity];
    const bool clip = a.max_norm > 0.f;
    float c = 1.f;
    if (clip) {
        c = a.max_norm / (norm + 1e-6f);
        if (c > 1.f) c = 1.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        step_state[parity ^ 1] = skip ? t0 : t0 + 1;
        if (a.ls) loss_scale_update(a.ls, parity, !loss_bad && !isfinite(norm), skip);
        if (norm_out) *norm_out = norm;
        if (stats) {
            stats->grad_norm = norm;
            if (!isfinite(norm) || extra) stats->found_inf = 1.0f;
            if (stats_host) {
                ww_step_stats s = *stats;
                s.grad_norm = norm;
                if (!isfinite(norm) || extra) s.found_inf = 1.0f;
                *stats_host = s;
            }
        }
    }
    float step_size, bc2_sqrt;
    bias_terms(a, t0 + 1, step_size, bc2_sqrt);
    // float4 body (the buckets are 16-byte aligned: host check) + scalar tail; per element the arithmetic of the scalar form
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 g4 = reinterpret_cast<const float4 *>(g)[i];
        float ge[4] = {g4.x * inv_scale, g4.y * inv_scale, g4.z * inv_scale, g4.w * inv_scale};
        if (clip || a.ls) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ge[e] *= c;
            reinterpret_cast<float4 *>(g)[i] = make_float4(ge[0], ge[1], ge[2], ge[3]);
        }
        if (skip) continue;
        const float4 p4 = reinterpret_cast<const float4 *>(p)[i];
        const float4 m4 = m ? reinterpret_cast<const float4 *>(m)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v4 = v ? reinterpret_cast<const float4 *>(v)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float pe[4] = {p4.x, p4.y, p4.z, p4.w}, me[4] = {m4.x, m4.y, m4.z, m4.w}, ve[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) optim_update(a, pe[e], ge[e], me[e], ve[e], step_size, bc2_sqrt);
        reinterpret_cast<float4 *>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
        if (m) reinterpret_cast<float4 *>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
        if (v) reinterpret_cast<float4 *>(v)[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * inv_scale;
        if (clip || a.ls) { gi *= c; g[i] = gi; }   // clip_grad_norm_ / unscale_ leave the clipped, unscaled gradients behind
        if (skip) continue;
        float pi = p[i], mi = m ? m[i] : 0.f, vi = v ? v[i] : 0.f;
        optim_update(a, pi, gi, mi, vi, step_size, bc2_sqrt);
        p[i] = pi;
        if (m) m[i] = mi;
        if (v) v[i] = vi;
    }
}

}  // namespace

extern "C" int ww_clip_optim_step(ww_ctx *ctx, const ww_optim_cfg *cfg, float *flat_params, float *flat_grads,
                                  float *exp_avg, float *exp_avg_sq, size_t n, int64_t *step_state, int parity,
                                  float *norm_out, ww_step_stats *stats, ww_step_stats *stats_host,
                                  ww_step_stats *stats_host_alt, const float *found_inf_extra, ww_loss_scale *loss_scale,
                                  ww_stream_t stream) {
    WW_REQUIRE(ctx && cfg && flat_params && flat_grads && step_state, WW_E_INVALID, "ww_clip_optim_step: null argument");
    ww_step_stats *stats_host_dev = nullptr, *stats_host_alt_dev = nullptr;
    if (stats_host) {
        WW_REQUIRE(stats != nullptr, WW_E_INVALID, "ww_clip_optim_step: stats_host needs stats");
        WW_HIP(hipHostGetDevicePointer((void **)&stats_host_dev, stats_host, 0));   // fails for pageable memory
    }
    if (stats_host_alt) {
        WW_REQUIRE(stats_host != nullptr && ctx->step_ctl != nullptr, WW_E_INVALID,
                   "ww_clip_optim_step: stats_host_alt needs stats_host and a bound step control block");
        WW_HIP(hipHostGetDevicePointer((void **)&stats_host_alt_dev, stats_host_alt, 0));
    }
    WW_REQUIRE(cfg->kind == WW_OPT_ADAM || cfg->kind == WW_OPT_ADAMW || cfg->kind == WW_OPT_SGD, WW_E_INVALID,
               "ww_clip_optim_step: unknown optimizer kind %d", cfg->kind);
    WW_REQUIRE(parity == 0 || parity == 1, WW_E_INVALID, "ww_clip_optim_step: parity must be 0 or 1");
    WW_REQUIRE(cfg->lr > 0.f || ctx->step_ctl, WW_E_INVALID, "Learning rate must be positive, got %g", (double)cfg->lr);
    WW_REQUIRE(cfg->weight_decay >= 0.f, WW_E_INVALID, "Weight decay must be non-negative, got %g", (double)cfg->weight_decay);
    if (cfg->kind == WW_OPT_SGD) {
        WW_REQUIRE(cfg->momentum >= 0.f && cfg->momentum <= 1.f, WW_E_INVALID, "Momentum must be in [0, 1], got %g",
                   (double)cfg->momentum);
        WW_REQUIRE(cfg->momentum == 0.f || exp_avg, WW_E_INVALID, "ww_clip_optim_step: SGD momentum needs the buffer (exp_avg)");
    } else {
        WW_REQUIRE(cfg->beta1 >= 0.f && cfg->beta1 <= 1.f && cfg->beta2 >= 0.f && cfg->beta2 <= 1.f, WW_E_INVALID,
                   "Betas must be in [0, 1], got (%g, %g)", (double)cfg->beta1, (double)cfg->beta2);
        WW_REQUIRE(exp_avg && exp_avg_sq, WW_E_INVALID, "ww_clip_optim_step: Adam needs exp_avg and exp_avg_sq");
    }
    if (n == 0) return WW_OK;
    OptimArgs a{cfg->kind, cfg->lr, cfg->beta1, cfg->beta2, cfg->eps, cfg->weight_decay, cfg->momentum, cfg->max_norm,
                ctx->step_ctl, loss_scale};
    hipStream_t st = (hipStream_t)stream;
    // Buckets of a few thousand parameters take ONE single-block launch; anything larger two grid-wide ones (block partial
    // sums of squares, then every block of the update reduces them in the same order).  cnn_small's 20 546 parameters used
    // to take the single block too: 33 us of a 1.39 ms step, latency-bound (one block cannot overlap its four dependent
    // rounds of loads); the two-launch form is 14 us (r02: 1.387 -> 1.364 ms per step).  WW_OPTIM_SMALL_MAX overrides.
    static const size_t small_max = []() {
        const char *e = getenv("WW_OPTIM_SMALL_MAX");
        return e ? (size_t)atol(e) : (size_t)4096;
    }();
    if (n <= std::min(small_max, (size_t)OPT_EPT * 1024)) {
        ww_prof_scope ps_(ctx, WW_K_CLIP, st);
        hipLaunchKernelGGL(k_clip_optim_small, dim3(1), dim3(1024), 0, st, a, flat_params, flat_grads, exp_avg, exp_avg_sq,
                           n, (long long *)step_state, parity, norm_out, stats, stats_host_dev, stats_host_alt_dev, found_inf_extra);
        WW_LAUNCH_CHECK();
        return WW_OK;
    }
    ww_prof_scope ps_(ctx, WW_K_CLIP, st);
    int parts = 0;
    const int rc = ww_launch_sumsq_partials(ctx, flat_grads, n, &parts, st);
    if (rc) return rc;
    // float4 body for large buckets (MobileNetV3's 1.5 M parameters: 42 -> ~25 us); a small bucket is latency-bound and
    // runs a little faster spread over more blocks in the scalar form (cnn_small's 20 546: 5.9 vs 7.9 us)
    const int vec = n >= ((size_t)1 << 16) &&
                    (((uintptr_t)flat_params | (uintptr_t)flat_grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
    const int grid = (int)std::min<size_t>(((vec ? n / 4 + 3 : n) + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_optim_update, dim3(grid), dim3(256), 0, st, a, flat_params, flat_grads, exp_avg, exp_avg_sq, n,
                       (long long *)step_state, parity, ctx->norm_partials, parts, norm_out, stats, stats_host_dev,
                       stats_host_alt_dev, found_inf_extra, vec);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
