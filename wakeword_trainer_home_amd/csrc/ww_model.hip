// cnn_small forward / backward: workspace layout + layer sequencing on one stream.
// (Model definition: SURVEY.md §8a-M; the reference's create_model has no such entry,
//  src/models/architectures.py:458-509 -- it is added to the factory by this build.)
#include "ww_internal.h"
#include "ww_act.h"
#include <algorithm>

namespace {

constexpr int NL = 9;  // conv layers: 0 stem, 1+2i dw_i, 2+2i pw_i

// all offsets in BYTES from the workspace base
struct Layout {
    int Ho, Wo;
    size_t A;                       // elements per activation tensor
    size_t y[NL], g[2], ss[NL], mr[NL], coef[NL], pool, pd, dpool, scratch, total;
};

Layout make_layout(int B, int F, int T, int act_dtype) {
    Layout L;
    L.Ho = (F + 1) / 2;
    L.Wo = (T + 1) / 2;
    L.A = (size_t)B * L.Ho * L.Wo * 64;
    const size_t esz = act_dtype == WW_ACT_F32 ? 4 : 2;
    size_t o = 0;
    auto takeb = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    auto take = [&](size_t nfloat) { return takeb(nfloat * sizeof(float)); };
    for (int i = 0; i < NL; ++i) L.y[i] = takeb(L.A * esz);
    L.g[0] = takeb(L.A * esz);
    L.g[1] = takeb(L.A * esz);
    for (int i = 0; i < NL; ++i) { L.ss[i] = take(128); L.mr[i] = take(128); L.coef[i] = take(192); }
    L.pool = take((size_t)B * 192);
    L.pd = take((size_t)B * 64);
    L.dpool = take((size_t)B * 64);
    L.scratch = take(WW_STAT_SLAB_FLOATS + WW_DW_SLAB_FLOATS);
    L.total = o;
    return L;
}

// state_dict order (wwhip.h): conv weight index and BatchNorm base index of conv layer l
inline int widx(int l) { return l == 0 ? 0 : (l & 1 ? 5 + 10 * ((l - 1) / 2) : 10 + 10 * ((l - 2) / 2)); }
inline int bnidx(int l) { return widx(l) + 1; }  // gamma, beta, running_mean, running_var

ww_bn_t make_bn(void *const *p, int l, int training, float mom, float eps) {
    ww_bn_t bn;
    const int b = bnidx(l);
    bn.gamma = (const float *)p[b];
    bn.beta = (const float *)p[b + 1];
    bn.running_mean = (float *)p[b + 2];
    bn.running_var = (float *)p[b + 3];
    bn.momentum = mom;
    bn.eps = eps;
    bn.training = training;
    return bn;
}

int check_common(const char *who, ww_ctx *ctx, int act_dtype, void *const *params, const void *x, int B, int F, int T,
                 void *ws, size_t ws_bytes, int nparams = WW_CNN_SMALL_NPTR) {
    WW_REQUIRE(act_dtype == WW_ACT_F32 || act_dtype == WW_ACT_BF16 || act_dtype == WW_ACT_F16, WW_E_INVALID, "%s: unknown act_dtype %d", who,
               act_dtype);
    WW_REQUIRE(ctx && params && x && ws, WW_E_INVALID, "%s: null argument", who);
    WW_REQUIRE(B >= 1 && F >= 1 && T >= 1, WW_E_INVALID, "%s: bad shape (%d,1,%d,%d)", who, B, F, T);
    for (int i = 0; i < nparams; ++i)
        WW_REQUIRE(params[i] != nullptr, WW_E_INVALID, "%s: params[%d] is null", who, i);
    const size_t need = ww_cnn_small_workspace_bytes(B, F, T, act_dtype);
    WW_REQUIRE(ws_bytes >= need, WW_E_WORKSPACE, "%s: workspace %zu B < required %zu B", who, ws_bytes, need);
    WW_REQUIRE(((uintptr_t)ws & 255) == 0, WW_E_INVALID, "%s: workspace must be 256-byte aligned", who);
    return WW_OK;
}

// stem + 4 x (depthwise, pointwise): y_l, scale/shift and mean/rstd of every layer into the workspace
int conv_stack_fwd(ww_ctx *ctx, int act_dtype, void *const *params, const float *x, int B, int F, int T, int training,
                   float bn_momentum, float bn_eps, const Layout &L, char *w, ww_stream_t stream) {
    auto Fp = [&](size_t off) { return (float *)(w + off); };
    void *scratch = w + L.scratch;
    ww_bn_t bn = make_bn(params, 0, training, bn_momentum, bn_eps);
    int rc = ww_conv_stem_fwd(ctx, act_dtype, x, (const float *)params[0], B, F, T, w + L.y[0], &bn, Fp(L.ss[0]), Fp(L.mr[0]),
                              scratch, stream);
    if (rc) return rc;
    for (int i = 0; i < 4; ++i) {
        const int ld = 1 + 2 * i, lp = 2 + 2 * i;
        bn = make_bn(params, ld, training, bn_momentum, bn_eps);
        rc = ww_dwconv3x3_fwd(ctx, act_dtype, w + L.y[ld - 1], Fp(L.ss[ld - 1]), (const float *)params[widx(ld)], B, L.Ho,
                              L.Wo, w + L.y[ld], &bn, Fp(L.ss[ld]), Fp(L.mr[ld]), scratch, stream);
        if (rc) return rc;
        bn = make_bn(params, lp, training, bn_momentum, bn_eps);
        rc = ww_pwconv1x1_fwd(ctx, act_dtype, w + L.y[ld], Fp(L.ss[ld]), (const float *)params[widx(lp)], B, L.Ho, L.Wo,
                              w + L.y[lp], &bn, Fp(L.ss[lp]), Fp(L.mr[lp]), scratch, stream);
        if (rc) return rc;
    }
    return WW_OK;
}

// backward of the conv stack.  from_pool: layer 8's dL/dz is synthesised from the pooled gradient (cnn_small's GAP head);
// otherwise it has been written to g[0] together with coef[8] (the CRNN's frequency pooling)
// part: WW_BWD_ALL, or WW_BWD_LATE (blocks 3, 2) / WW_BWD_EARLY (blocks 1, 0 and the stem) -- the two halves of a
// data-parallel step whose first gradient bucket is all-reduced while the second half runs
int conv_stack_bwd(ww_ctx *ctx, int act_dtype, void *const *params, void *const *grads, const float *x, int B, int F, int T,
                   const Layout &L, char *w, bool from_pool, ww_stream_t stream, int part = WW_BWD_ALL) {
    auto Fp = [&](size_t off) { return (float *)(w + off); };
    auto G = [&](int i) { return (float *)grads[i]; };
    void *scratch = w + L.scratch;
    int rc;
    const int i_hi = part == WW_BWD_EARLY ? 1 : 3, i_lo = part == WW_BWD_LATE ? 2 : 0;
    // g[cur] holds dL/dz of the layer about to be processed (unused for layer 8 when from_pool); every block flips it twice
    int cur = (from_pool && i_hi != 3) ? 1 : 0;
    for (int i = i_hi; i >= i_lo; --i) {
        const int ld = 1 + 2 * i, lp = 2 + 2 * i;
        const bool pooled = from_pool && lp == 8;
        const void *gp = pooled ? nullptr : w + L.g[cur];
        const int nxt = pooled ? 0 : cur ^ 1;
        rc = ww_pwconv1x1_bwd(ctx, act_dtype, gp, Fp(L.dpool), w + L.y[lp], Fp(L.ss[lp]), Fp(L.coef[lp]), w + L.y[ld],
                              Fp(L.ss[ld]), Fp(L.mr[ld]), (const float *)params[bnidx(ld)],
                              (const float *)params[widx(lp)], B, L.Ho, L.Wo, w + L.g[nxt], G(widx(lp)), Fp(L.coef[ld]),
                              G(bnidx(ld)), G(bnidx(ld) + 1), scratch, stream);
        if (rc) return rc;
        cur = nxt;
        const int lprev = ld - 1;
        rc = ww_dwconv3x3_bwd(ctx, act_dtype, w + L.g[cur], w + L.y[ld], Fp(L.coef[ld]), w + L.y[lprev], Fp(L.ss[lprev]),
                              Fp(L.mr[lprev]), (const float *)params[bnidx(lprev)], (const float *)params[widx(ld)], B,
                              L.Ho, L.Wo, w + L.g[cur ^ 1], G(widx(ld)), Fp(L.coef[lprev]), G(bnidx(lprev)),
                              G(bnidx(lprev) + 1), scratch, stream);
        if (rc) return rc;
        cur ^= 1;
    }
    if (part == WW_BWD_LATE) return WW_OK;
    return ww_stem_bwd_impl(ctx, act_dtype, w + L.g[cur], w + L.y[0], (const float *)params[0], Fp(L.coef[0]), x, B, F, T, G(0),
                            scratch, stream);
}

// ---- frequency pooling between the conv stack and a recurrent layer (CRNN): seq[b][w][c] = mean_h relu(bn(y8[b][h][w][c]))
template <typename T>
__global__ __launch_bounds__(256) void k_freqpool_fwd(const T *__restrict__ y, const float *__restrict__ ss, int B, int H, int W,
                                                      float *__restrict__ seq) {
    const int grp = threadIdx.x >> 5, cl = threadIdx.x & 31;
    const float2 sc = *reinterpret_cast<const float2 *>(ss + 2 * cl);
    const float2 sf = *reinterpret_cast<const float2 *>(ss + 64 + 2 * cl);
    const float inv = 1.0f / (float)H;
    for (long it = (long)blockIdx.x * 8 + grp; it < (long)B * W; it += (long)gridDim.x * 8) {
        const long b = it / W;
        const int wc = (int)(it % W);
        const T *p = y + ((size_t)b * H * W + wc) * 64 + 2 * cl;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
        for (int h = 0; h < H; ++h) {
            const float2 v = Act<T>::cvt2(Act<T>::ldraw2(p + (size_t)h * W * 64));
            const float z0 = fmaf(v.x, sc.x, sf.x), z1 = fmaf(v.y, sc.y, sf.y);
            a0 += z0 < 0.f ? 0.f : z0;
            a1 += z1 < 0.f ? 0.f : z1;
        }
        *reinterpret_cast<float2 *>(seq + (size_t)it * 64 + 2 * cl) = make_float2(a0 * inv, a1 * inv);
    }
}
// g8[b][h][w][c] = dseq[b][w][c] / H where z > 0 (stored like every g_l), plus the BatchNorm-backward sums of layer 8
template <typename T>
__global__ __launch_bounds__(256) void k_freqpool_bwd(const float *__restrict__ dseq, const T *__restrict__ y,
                                                      const float *__restrict__ ss, const float *__restrict__ mr, int B, int H,
                                                      int W, T *__restrict__ g, float *__restrict__ stat_partials) {
    __shared__ float sh[8 * 128];
    const int grp = threadIdx.x >> 5, cl = threadIdx.x & 31;
    const float2 sc = *reinterpret_cast<const float2 *>(ss + 2 * cl);
    const float2 sf = *reinterpret_cast<const float2 *>(ss + 64 + 2 * cl);
    const float2 mu = *reinterpret_cast<const float2 *>(mr + 2 * cl);
    const float2 rs = *reinterpret_cast<const float2 *>(mr + 64 + 2 * cl);
    const float inv = 1.0f / (float)H;
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
    for (long it = (long)blockIdx.x * 8 + grp; it < (long)B * W; it += (long)gridDim.x * 8) {
        const long b = it / W;
        const int wc = (int)(it % W);
        const float2 d = *reinterpret_cast<const float2 *>(dseq + (size_t)it * 64 + 2 * cl);
        const size_t base = ((size_t)b * H * W + wc) * 64 + 2 * cl;
#pragma unroll 4
        for (int h = 0; h < H; ++h) {
            const size_t o = base + (size_t)h * W * 64;
            const float2 v = Act<T>::cvt2(Act<T>::ldraw2(y + o));
            const float z0 = fmaf(v.x, sc.x, sf.x), z1 = fmaf(v.y, sc.y, sf.y);
            const float2 gv = Act<T>::round2(make_float2(z0 > 0.f ? d.x * inv : 0.f, z1 > 0.f ? d.y * inv : 0.f));
            Act<T>::st2(g + o, gv);
            s1a += gv.x; s1b += gv.y;
            s2a = fmaf(gv.x, (v.x - mu.x) * rs.x, s2a);
            s2b = fmaf(gv.y, (v.y - mu.y) * rs.y, s2b);
        }
    }
    sh[grp * 128 + 2 * cl] = s1a;       sh[grp * 128 + 2 * cl + 1] = s1b;
    sh[grp * 128 + 64 + 2 * cl] = s2a;  sh[grp * 128 + 64 + 2 * cl + 1] = s2b;
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sh[q * 128 + threadIdx.x];
        stat_partials[(size_t)blockIdx.x * 128 + threadIdx.x] = t;
    }
}

}  // namespace

extern "C" size_t ww_cnn_small_workspace_bytes(int B, int F, int T, int act_dtype) {
    if (B < 1 || F < 1 || T < 1) return 0;
    return make_layout(B, F, T, act_dtype).total;
}

extern "C" int ww_cnn_small_fwd(ww_ctx *ctx, int act_dtype, void *const *params, const float *x, int B, int F, int T, int training,
                                float bn_momentum, float bn_eps, float dropout_p, uint64_t seed, uint64_t step,
                                uint64_t sample_offset, void *ws, size_t ws_bytes, float *logits, ww_stream_t stream) {
    int rc = check_common("ww_cnn_small_fwd", ctx, act_dtype, params, x, B, F, T, ws, ws_bytes);
    if (rc) return rc;
    WW_REQUIRE(logits != nullptr, WW_E_INVALID, "ww_cnn_small_fwd: logits is null");
    const Layout L = make_layout(B, F, T, act_dtype);
    char *w = (char *)ws;
    auto Fp = [&](size_t off) { return (float *)(w + off); };
    rc = conv_stack_fwd(ctx, act_dtype, params, x, B, F, T, training, bn_momentum, bn_eps, L, w, stream);
    if (rc) return rc;
    rc = ww_gap_fwd(ctx, act_dtype, w + L.y[8], Fp(L.ss[8]), Fp(L.mr[8]), B, L.Ho, L.Wo, Fp(L.pool), stream);
    if (rc) return rc;
    return ww_head_fwd(ctx, Fp(L.pool), B, L.Ho * L.Wo, (const float *)params[45], (const float *)params[46], dropout_p,
                       training, seed, step, sample_offset, Fp(L.pd), logits, stream);
}

extern "C" int ww_cnn_small_bwd(ww_ctx *ctx, int act_dtype, void *const *params, void *const *grads, const float *x,
                                const float *dlogits, int B, int F, int T, float dropout_p, uint64_t seed, uint64_t step,
                                uint64_t sample_offset, void *ws, size_t ws_bytes, int part, ww_stream_t stream) {
    int rc = check_common("ww_cnn_small_bwd", ctx, act_dtype, params, x, B, F, T, ws, ws_bytes);
    if (rc) return rc;
    WW_REQUIRE(grads && dlogits, WW_E_INVALID, "ww_cnn_small_bwd: null argument");
    WW_REQUIRE(part == WW_BWD_ALL || part == WW_BWD_LATE || part == WW_BWD_EARLY, WW_E_INVALID,
               "ww_cnn_small_bwd: unknown part %d", part);
    const Layout L = make_layout(B, F, T, act_dtype);
    char *w = (char *)ws;
    auto Fp = [&](size_t off) { return (float *)(w + off); };
    auto G = [&](int i) { return (float *)grads[i]; };
    for (int l = 0; l < NL; ++l) {
        WW_REQUIRE(grads[widx(l)] && grads[bnidx(l)] && grads[bnidx(l) + 1], WW_E_INVALID,
                   "ww_cnn_small_bwd: missing gradient buffer for conv layer %d", l);
    }
    WW_REQUIRE(grads[45] && grads[46], WW_E_INVALID, "ww_cnn_small_bwd: missing classifier gradient buffers");
    const int HW = L.Ho * L.Wo;
    if (part != WW_BWD_EARLY) {
        rc = ww_head_bwd(ctx, dlogits, Fp(L.pd), Fp(L.pool), B, HW, (const float *)params[45], dropout_p, 1, seed, step,
                         sample_offset, (const float *)params[bnidx(8)], Fp(L.mr[8]), G(45), G(46), Fp(L.dpool),
                         Fp(L.coef[8]), G(bnidx(8)), G(bnidx(8) + 1), stream);
        if (rc) return rc;
    }
    return conv_stack_bwd(ctx, act_dtype, params, grads, x, B, F, T, L, w, /*from_pool=*/true, stream, part);
}

// ------------------------------------------------------------------ conv front-end of the CRNN
// The cnn_small conv stack without GAP / classifier: features (B,1,F,T) -> sequence (B, ceil(T/2), 64), the mean over the
// frequency axis of the last layer's activations.  params / grads: the cnn_small pointer table, entries 45/46 unused.
extern "C" int ww_cnn_front_fwd(ww_ctx *ctx, int act_dtype, void *const *params, const float *x, int B, int F, int T, int training,
                                float bn_momentum, float bn_eps, void *ws, size_t ws_bytes, float *seq, ww_stream_t stream) {
    int rc = check_common("ww_cnn_front_fwd", ctx, act_dtype, params, x, B, F, T, ws, ws_bytes, 45);
    if (rc) return rc;
    WW_REQUIRE(seq != nullptr, WW_E_INVALID, "ww_cnn_front_fwd: seq is null");
    const Layout L = make_layout(B, F, T, act_dtype);
    char *w = (char *)ws;
    rc = conv_stack_fwd(ctx, act_dtype, params, x, B, F, T, training, bn_momentum, bn_eps, L, w, stream);
    if (rc) return rc;
    const long items = (long)B * L.Wo;
    const int grid = (int)std::min<long>((items + 7) / 8, 2048);
    hipStream_t st = (hipStream_t)stream;
    if (act_dtype == WW_ACT_BF16)
        hipLaunchKernelGGL(k_freqpool_fwd<ww_bf16>, dim3(grid), dim3(256), 0, st, (const ww_bf16 *)(w + L.y[8]),
                           (const float *)(w + L.ss[8]), B, L.Ho, L.Wo, seq);
    else if (act_dtype == WW_ACT_F16)
        hipLaunchKernelGGL(k_freqpool_fwd<ww_f16>, dim3(grid), dim3(256), 0, st, (const ww_f16 *)(w + L.y[8]),
                           (const float *)(w + L.ss[8]), B, L.Ho, L.Wo, seq);
    else
        hipLaunchKernelGGL(k_freqpool_fwd<float>, dim3(grid), dim3(256), 0, st, (const float *)(w + L.y[8]),
                           (const float *)(w + L.ss[8]), B, L.Ho, L.Wo, seq);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" int ww_cnn_front_bwd(ww_ctx *ctx, int act_dtype, void *const *params, void *const *grads, const float *x,
                                const float *dseq, int B, int F, int T, void *ws, size_t ws_bytes, ww_stream_t stream) {
    int rc = check_common("ww_cnn_front_bwd", ctx, act_dtype, params, x, B, F, T, ws, ws_bytes, 45);
    if (rc) return rc;
    WW_REQUIRE(grads && dseq, WW_E_INVALID, "ww_cnn_front_bwd: null argument");
    for (int l = 0; l < NL; ++l)
        WW_REQUIRE(grads[widx(l)] && grads[bnidx(l)] && grads[bnidx(l) + 1], WW_E_INVALID,
                   "ww_cnn_front_bwd: missing gradient buffer for conv layer %d", l);
    const Layout L = make_layout(B, F, T, act_dtype);
    char *w = (char *)ws;
    hipStream_t st = (hipStream_t)stream;
    float *stat = (float *)(w + L.scratch);
    const long items = (long)B * L.Wo;
    const int grid = (int)std::min<long>((items + 7) / 8, (long)WW_MAX_PARTIALS);
    if (act_dtype == WW_ACT_BF16)
        hipLaunchKernelGGL(k_freqpool_bwd<ww_bf16>, dim3(grid), dim3(256), 0, st, dseq, (const ww_bf16 *)(w + L.y[8]),
                           (const float *)(w + L.ss[8]), (const float *)(w + L.mr[8]), B, L.Ho, L.Wo, (ww_bf16 *)(w + L.g[0]), stat);
    else if (act_dtype == WW_ACT_F16)
        hipLaunchKernelGGL(k_freqpool_bwd<ww_f16>, dim3(grid), dim3(256), 0, st, dseq, (const ww_f16 *)(w + L.y[8]),
                           (const float *)(w + L.ss[8]), (const float *)(w + L.mr[8]), B, L.Ho, L.Wo, (ww_f16 *)(w + L.g[0]), stat);
    else
        hipLaunchKernelGGL(k_freqpool_bwd<float>, dim3(grid), dim3(256), 0, st, dseq, (const float *)(w + L.y[8]),
                           (const float *)(w + L.ss[8]), (const float *)(w + L.mr[8]), B, L.Ho, L.Wo, (float *)(w + L.g[0]), stat);
    WW_LAUNCH_CHECK();
    rc = ww_launch_bn_bwd_finalize(stat, grid, (double)B * L.Ho * L.Wo, (const float *)params[bnidx(8)],
                                   (const float *)(w + L.mr[8]), (float *)(w + L.coef[8]), (float *)grads[bnidx(8)],
                                   (float *)grads[bnidx(8) + 1], st);
    if (rc) return rc;
    return conv_stack_bwd(ctx, act_dtype, params, grads, x, B, F, T, L, w, /*from_pool=*/false, stream);
}
