// 2-class loss + gradient + step counters in one single-block kernel.
//   CE with label smoothing  : src/models/losses.py:66-98  (eps == 0 -> nn.CrossEntropyLoss, :256)
//   focal                    : src/models/losses.py:170-197
// For C == 2 the smoothed CE is exactly BCE-with-logits on d = z1 - z0 with soft target
// t = y(1-eps) + (1-y)eps, so this is the north_star's "BCE loss/grad".
// Also produces what trainer.py:196-200 / metrics.py:105-116 derive from the logits
// (argmax accuracy and confusion counters) so the step needs ONE small D2H read.
#include "ww_internal.h"

namespace {

__global__ __launch_bounds__(1024) void k_ce2_loss(const float *__restrict__ logits, const int64_t *__restrict__ targets,
                                                   int B, int kind, float eps, float alpha, float gamma,
                                                   float *__restrict__ loss_out, float *__restrict__ dlogits,
                                                   ww_step_stats *__restrict__ stats, float *__restrict__ found_inf_out,
                                                   const ww_loss_scale *__restrict__ ls, int ls_slot,
                                                   const ww_step_ctl *__restrict__ ctl) {
    __shared__ double shl[1024];
    __shared__ int shc[6][1024];
    // fp16 storage: dL/dlogits leaves this kernel times the dynamic loss scale (GradScaler.scale(loss).backward(),
    // src/training/trainer.py:182); the loss value itself is reported unscaled
    const float gscale = ls ? ls->scale[(ctl ? ctl->parity : ls_slot) & 1] : 1.0f;
    const float invB = gscale / (float)B;
    const float PT_MIN = 1e-7f, PT_MAX = (float)(1.0 - 1e-7);
    double lsum = 0.0;
    int correct = 0, tp = 0, tn = 0, fp = 0, fn = 0, bad = 0;
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float z0 = logits[(size_t)b * 2], z1 = logits[(size_t)b * 2 + 1];
        long long yy = targets[b];
        if (yy < 0 || yy > 1) {
            bad = 1;
            yy = yy < 0 ? 0 : 1;
        }
        const int y = (int)yy;
        const float m = fmaxf(z0, z1);
        const float e0 = expf(z0 - m), e1 = expf(z1 - m);
        const float se = e0 + e1;
        const float lse = m + logf(se);
        const float lp0 = z0 - lse, lp1 = z1 - lse;
        const float p0 = e0 / se, p1 = e1 / se;
        float lb, d0, d1;
        if (kind == WW_LOSS_CE) {
            const float s0 = y == 0 ? 1.0f - eps : eps;
            const float s1 = y == 1 ? 1.0f - eps : eps;
            lb = -(s0 * lp0 + s1 * lp1);
            d0 = (p0 - s0) * invB;
            d1 = (p1 - s1) * invB;
        } else {
            const float pt_raw = y == 1 ? p1 : p0;
            const float pt = fminf(fmaxf(pt_raw, PT_MIN), PT_MAX);
            const float om = 1.0f - pt;
            const float fw = gamma == 0.f ? 1.0f : powf(om, gamma);
            const float a_t = y == 1 ? alpha : 1.0f - alpha;
            const float ce = -(y == 1 ? lp1 : lp0);
            lb = a_t * fw * ce;
            const bool inside = pt_raw > PT_MIN && pt_raw < PT_MAX;
            const float dfw = (inside && gamma != 0.f) ? -gamma * powf(om, gamma - 1.0f) : 0.f;
            const float oh0 = y == 0 ? 1.f : 0.f, oh1 = 1.f - oh0;
            const float t = dfw * ce * pt_raw;
            d0 = a_t * (fw * (p0 - oh0) + t * (oh0 - p0)) * invB;
            d1 = a_t * (fw * (p1 - oh1) + t * (oh1 - p1)) * invB;
        }
        dlogits[(size_t)b * 2] = d0;
        dlogits[(size_t)b * 2 + 1] = d1;
        lsum += (double)lb;
        const int pred = z1 > z0 ? 1 : 0;  // torch.argmax: first max wins on ties
        correct += pred == y;
        tp += (pred == 1) & (y == 1);
        tn += (pred == 0) & (y == 0);
        fp += (pred == 1) & (y == 0);
        fn += (pred == 0) & (y == 1);
    }
    shl[threadIdx.x] = lsum;
    shc[0][threadIdx.x] = correct; shc[1][threadIdx.x] = tp; shc[2][threadIdx.x] = tn;
    shc[3][threadIdx.x] = fp; shc[4][threadIdx.x] = fn; shc[5][threadIdx.x] = bad;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            shl[threadIdx.x] += shl[threadIdx.x + s];
#pragma unroll
            for (int k = 0; k < 6; ++k) shc[k][threadIdx.x] += shc[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float loss = (float)(shl[0] / (double)B);
        if (loss_out) *loss_out = loss;
        // data parallel: the skip decision travels with the gradients (one extra float at the end of the flat bucket that the
        // all-reduce sums), so every rank skips the step when any rank saw a bad batch
        if (found_inf_out) *found_inf_out = (!isfinite(loss) || shc[5][0]) ? 1.0f : 0.0f;
        if (stats) {
            stats->loss = loss;
            stats->correct = shc[0][0]; stats->tp = shc[1][0]; stats->tn = shc[2][0];
            stats->fp = shc[3][0]; stats->fn = shc[4][0];
            stats->nonfinite = isfinite(loss) ? 0 : 1;
            stats->bad_target = shc[5][0] ? 1 : 0;
            stats->count = B;
            stats->grad_norm = 0.f;
            stats->found_inf = (!isfinite(loss) || shc[5][0]) ? 1.0f : 0.0f;
            stats->reserved = 0;
        }
    }
}

}  // namespace

extern "C" int ww_ce2_loss_fwd_bwd(ww_ctx *ctx, const float *logits, const int64_t *targets, int B, int loss_kind,
                                   float label_smoothing, float focal_alpha, float focal_gamma, float *loss_out,
                                   float *dlogits, ww_step_stats *stats, float *found_inf_out, const ww_loss_scale *loss_scale,
                                   int loss_scale_slot, ww_stream_t stream) {
    WW_REQUIRE(ctx && logits && targets && dlogits, WW_E_INVALID, "ww_ce2_loss_fwd_bwd: null argument");
    WW_REQUIRE(B >= 1, WW_E_INVALID, "ww_ce2_loss_fwd_bwd: B=%d", B);
    WW_REQUIRE(loss_kind == WW_LOSS_CE || loss_kind == WW_LOSS_FOCAL, WW_E_INVALID,
               "ww_ce2_loss_fwd_bwd: unknown loss_kind %d", loss_kind);
    // same range checks as the reference constructors (losses.py:36-37, 132-135)
    WW_REQUIRE(label_smoothing >= 0.f && label_smoothing <= 1.f, WW_E_INVALID,
               "Label smoothing must be in [0, 1], got %g", label_smoothing);
    WW_REQUIRE(focal_alpha >= 0.f && focal_alpha <= 1.f, WW_E_INVALID, "Alpha must be in [0, 1], got %g", focal_alpha);
    WW_REQUIRE(focal_gamma >= 0.f, WW_E_INVALID, "Gamma must be non-negative, got %g", focal_gamma);
    WW_REQUIRE(loss_scale_slot == 0 || loss_scale_slot == 1, WW_E_INVALID, "ww_ce2_loss_fwd_bwd: loss_scale_slot must be 0 or 1");
    ww_prof_scope ps_(ctx, WW_K_HEAD_LOSS, (hipStream_t)stream);
    hipLaunchKernelGGL(k_ce2_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, targets, B, loss_kind,
                       label_smoothing, focal_alpha, focal_gamma, loss_out, dlogits, stats, found_inf_out, loss_scale,
                       loss_scale_slot, ctx->step_ctl);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
