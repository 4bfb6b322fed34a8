// Context, error channel, constant tables and the small reduction / finalize kernels.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "ww_internal.h"
#include <algorithm>

static thread_local char g_err[512] = "";

void ww_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ww_last_error(void) { return g_err; }
extern "C" int ww_abi_version(void) { return WW_ABI_VERSION; }

extern "C" uint64_t ww_prob_threshold(double p) {
    double t = floor(p * 4294967296.0);
    if (!(t > 0.0)) return 0;  // also catches NaN
    if (t > 4294967296.0) t = 4294967296.0;
    return (uint64_t)t;
}

extern "C" void ww_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    ww_philox(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

extern "C" int ww_feat_num_frames(int n_samples, int hop) {
    if (hop <= 0 || n_samples < 0) return WW_E_INVALID;
    return 1 + n_samples / hop;
}

extern "C" int ww_ctx_create(int device, ww_ctx **out) {
    WW_REQUIRE(out != nullptr, WW_E_INVALID, "ww_ctx_create: out is null");
    int ndev = 0;
    WW_HIP(hipGetDeviceCount(&ndev));
    WW_REQUIRE(device >= 0 && device < ndev, WW_E_INVALID, "ww_ctx_create: device %d not in [0,%d)", device, ndev);
    hipDeviceProp_t prop;
    WW_HIP(hipGetDeviceProperties(&prop, device));
    WW_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, WW_E_UNSUPPORTED,
               "ww_ctx_create: libwwhip is built for gfx950 only, device %d is %s", device, prop.gcnArchName);
    ww_ctx *c = new ww_ctx();
    c->device = device;
    c->tables = nullptr;
    c->prof_mask = 0;
    c->step_ctl = nullptr;
    c->logmel_wgs = 0;
    c->tw16k = nullptr;
    c->norm_partials = nullptr;
    c->defer_on = 0;
    c->deferred = new std::vector<ww_reduce_item>();
    c->prof_recs = new std::vector<ww_prof_rec>();
    c->prof_free = new std::vector<ww_prof_rec>();
    *out = c;
    return WW_OK;
}

// ---- device-resident step control (HIP graph replay)
__global__ void k_step_ctl_advance(ww_step_ctl *ctl) {
    ctl->step += 1;
    ctl->parity ^= 1;
}

extern "C" int ww_ctx_bind_step_ctl(ww_ctx *ctx, const ww_step_ctl *ctl_dev) {
    WW_REQUIRE(ctx != nullptr, WW_E_INVALID, "ww_ctx_bind_step_ctl: ctx is null");
    WW_REQUIRE(((uintptr_t)ctl_dev & 7) == 0, WW_E_INVALID, "ww_ctx_bind_step_ctl: the control block must be 8-byte aligned");
    ctx->step_ctl = ctl_dev;
    return WW_OK;
}

extern "C" int ww_ctx_set_logmel_workgroups(ww_ctx *ctx, int n) {
    WW_REQUIRE(ctx != nullptr, WW_E_INVALID, "ww_ctx_set_logmel_workgroups: ctx is null");
    WW_REQUIRE(n >= 0, WW_E_INVALID, "ww_ctx_set_logmel_workgroups: n=%d must be >= 0", n);
    ctx->logmel_wgs = n;
    return WW_OK;
}

extern "C" int ww_step_ctl_advance(ww_ctx *ctx, ww_stream_t stream) {
    WW_REQUIRE(ctx && ctx->step_ctl, WW_E_INVALID, "ww_step_ctl_advance: no control block is bound");
    hipLaunchKernelGGL(k_step_ctl_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, const_cast<ww_step_ctl *>(ctx->step_ctl));
    WW_LAUNCH_CHECK();
    return WW_OK;
}

static void free_tables(ww_feat_tables *t) {
    while (t) {
        ww_feat_tables *n = t->next;
        (void)hipFree(t->window);
        (void)hipFree(t->twiddle);
        (void)hipFree(t->mel_start);
        (void)hipFree(t->mel_len);
        (void)hipFree(t->mel_off);
        (void)hipFree(t->mel_w);
        if (t->melq_tab) (void)hipFree(t->melq_tab);
        if (t->melq_w) (void)hipFree(t->melq_w);
        if (t->dct) (void)hipFree(t->dct);
        if (t->dct_t) (void)hipFree(t->dct_t);
        delete t;
        t = n;
    }
}

extern "C" int ww_ctx_destroy(ww_ctx *ctx) {
    if (!ctx) return WW_OK;
    free_tables(ctx->tables);
    if (ctx->tw16k) (void)hipFree(ctx->tw16k);
    if (ctx->norm_partials) (void)hipFree(ctx->norm_partials);
    for (auto *v : {ctx->prof_recs, ctx->prof_free}) {
        for (auto &r : *v) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        delete v;
    }
    delete ctx->deferred;
    delete ctx;
    return WW_OK;
}

// ---- deferred partial sums.  The weight-gradient kernels of the generic layers (split-K GEMMs, depthwise / stem weight gradients)
// end in "sum the per-block partials" -- a 4-5 us launch each, ~35 of them in a MobileNetV3 step, every one a pure latency chain
// nothing waits for before the optimizer.  While the context is deferring they are queued, and ONE launch runs them all.
constexpr int RED_MAX = 64;
struct ReduceBatch { ww_reduce_item it[RED_MAX]; int first_block[RED_MAX + 1]; int count; };
// grid = sum over items of ceil(n / 256), block 256 = 64 output float4s x 4 partial-row lanes: lane q sums rows q, q+4, ... in
// double (8 loads in flight), the four lane sums are added in lane order through LDS -- a fixed partition and order, so the
// result is deterministic.  (First form: 1024 outputs per block, one thread walking all rows -- ~500 blocks for ~80 MB of
// partials moved at 1 TB/s, 83 us per MobileNetV3 step.)
constexpr int RED_COLS = 256;
__global__ __launch_bounds__(256) void k_reduce_items(ReduceBatch b) {
    __shared__ double sh[3][64][4];
    int i = 0;
    while (i + 1 < b.count && (int)blockIdx.x >= b.first_block[i + 1]) ++i;
    const float *__restrict__ part = b.it[i].part;
    float *__restrict__ dst = b.it[i].dst;
    const long n = b.it[i].n;
    const int splits = b.it[i].splits, acc = b.it[i].accumulate;
    const int c4 = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i0 = ((long)((int)blockIdx.x - b.first_block[i]) * 64 + c4) * 4;
    const bool vec = (n & 3) == 0 && (((uintptr_t)part | (uintptr_t)dst) & 15) == 0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i0 < n) {
        if (vec) {
            for (int z0 = q; z0 < splits; z0 += 32) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(part + (long)min(z0 + 4 * u, splits - 1) * n + i0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (z0 + 4 * u < splits) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }
            }
        } else {
            for (int z = q; z < splits; z += 4) {
                const float *p = part + (long)z * n + i0;
                s0 += p[0];
                if (i0 + 1 < n) s1 += p[1];
                if (i0 + 2 < n) s2 += p[2];
                if (i0 + 3 < n) s3 += p[3];
            }
        }
    }
    if (q > 0) { sh[q - 1][c4][0] = s0; sh[q - 1][c4][1] = s1; sh[q - 1][c4][2] = s2; sh[q - 1][c4][3] = s3; }
    __syncthreads();
    if (q != 0 || i0 >= n) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) { s0 += sh[k][c4][0]; s1 += sh[k][c4][1]; s2 += sh[k][c4][2]; s3 += sh[k][c4][3]; }
    if (vec) {
        float4 o = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
        if (acc) { const float4 c = *reinterpret_cast<const float4 *>(dst + i0); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
        *reinterpret_cast<float4 *>(dst + i0) = o;
    } else {
        const double sv[4] = {s0, s1, s2, s3};
        for (int k = 0; k < 4 && i0 + k < n; ++k) dst[i0 + k] = acc ? dst[i0 + k] + (float)sv[k] : (float)sv[k];
    }
}
extern "C" int ww_ctx_set_deferred_reduce(ww_ctx *ctx, int on) {
    WW_REQUIRE(ctx != nullptr, WW_E_INVALID, "ww_ctx_set_deferred_reduce: ctx is null");
    ctx->defer_on = on != 0;
    return WW_OK;
}
extern "C" int ww_deferred_reduce_pending(ww_ctx *ctx) { return ctx ? (int)ctx->deferred->size() : 0; }
extern "C" int ww_deferred_reduce_discard(ww_ctx *ctx) {      // drop what is queued without running it (a backward pass that failed midway)
    WW_REQUIRE(ctx != nullptr, WW_E_INVALID, "ww_deferred_reduce_discard: ctx is null");
    ctx->deferred->clear();
    ctx->defer_on = 0;
    return WW_OK;
}
extern "C" int ww_deferred_reduce_flush(ww_ctx *ctx, ww_stream_t stream) {
    WW_REQUIRE(ctx != nullptr, WW_E_INVALID, "ww_deferred_reduce_flush: ctx is null");
    std::vector<ww_reduce_item> &q = *ctx->deferred;
    for (size_t base = 0; base < q.size(); base += RED_MAX) {
        ReduceBatch b;
        b.count = (int)std::min<size_t>(RED_MAX, q.size() - base);
        int blocks = 0;
        for (int i = 0; i < b.count; ++i) {
            b.it[i] = q[base + i];
            b.first_block[i] = blocks;
            blocks += (int)((b.it[i].n + RED_COLS - 1) / RED_COLS);
        }
        for (int i = b.count; i <= RED_MAX; ++i) b.first_block[i] = blocks;
        for (int i = b.count; i < RED_MAX; ++i) b.it[i] = ww_reduce_item{nullptr, nullptr, 0, 0, 0};
        if (blocks > 0) hipLaunchKernelGGL(k_reduce_items, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    }
    q.clear();
    WW_LAUNCH_CHECK();
    return WW_OK;
}

static double hz_to_mel(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double mel_to_hz(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

template <typename T>
static int upload(T **dptr, const std::vector<T> &h) {
    size_t bytes = (h.empty() ? 1 : h.size()) * sizeof(T);
    WW_HIP(hipMalloc((void **)dptr, bytes));
    if (!h.empty()) WW_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return WW_OK;
}

// The mel filterbank of a feature configuration in the two forms the kernels read (host only; DESIGN.md "Feature spec" /
// oracle/features.py, double arithmetic): compact bands (first bin, length, offset into w) and, for n_fft 1024, the
// matrix-pipe form of k_logmel.
struct MelTables {
    std::vector<int32_t> start, len, off, qtab;
    std::vector<float> w, qw;
    int max_len = 0;
};
static void build_mel_tables(const ww_feat_cfg *cfg, MelTables &mt) {
    const int n_fft = cfg->n_fft, n_bins = n_fft / 2 + 1, M = cfg->n_mels;
    const double sr = cfg->sample_rate;
    const double f_min = cfg->f_min, f_max = cfg->f_max > 0.f ? cfg->f_max : sr / 2.0;
    std::vector<int32_t> &qtab = mt.qtab;
    std::vector<float> &w = mt.w, &qw = mt.qw;
    int &max_len = mt.max_len;
    std::vector<double> f_pts(M + 2);
    const double m_lo = hz_to_mel(f_min), m_hi = hz_to_mel(f_max);
    for (int i = 0; i < M + 2; ++i) {
        double m = (i == M + 1) ? m_hi : m_lo + i * ((m_hi - m_lo) / (M + 1));
        f_pts[i] = mel_to_hz(m);
    }
    std::vector<int32_t> &start = mt.start, &len = mt.len, &off = mt.off;
    start.assign(M, 0); len.assign(M, 0); off.assign(M, 0);
    w.clear();
    max_len = 0;
    for (int m = 0; m < M; ++m) {
        int s = -1, e = -1;
        std::vector<float> band;
        for (int k = 0; k < n_bins; ++k) {
            double f = (k == n_bins - 1) ? sr / 2.0 : k * ((sr / 2.0) / (n_bins - 1));
            double down = -(f_pts[m] - f) / (f_pts[m + 1] - f_pts[m]);
            double up = (f_pts[m + 2] - f) / (f_pts[m + 2] - f_pts[m + 1]);
            double v = fmax(0.0, fmin(down, up));
            if (v > 0.0) {
                if (s < 0) s = k;
                e = k;
            }
        }
        if (s < 0) { s = 0; e = -1; }
        start[m] = s;
        len[m] = e - s + 1;
        off[m] = (int32_t)w.size();
        for (int k = s; k <= e; ++k) {
            double f = (k == n_bins - 1) ? sr / 2.0 : k * ((sr / 2.0) / (n_bins - 1));
            double down = -(f_pts[m] - f) / (f_pts[m + 1] - f_pts[m]);
            double up = (f_pts[m + 2] - f) / (f_pts[m + 2] - f_pts[m + 1]);
            w.push_back((float)fmax(0.0, fmin(down, up)));
        }
        if (len[m] > max_len) max_len = len[m];
    }
    // MFMA form of the band sums (k_logmel, n_fft 1024): v_mfma_f32_4x4x1 runs 16 independent 4x4 outer products per
    // instruction -- block = 4 consecutive bands ("quad") x the wave's frames, one spectrum bin per step.  A UNIT is a run of
    // bins of one quad (in groups of 8 bins) handled by one block of one pass; a pass costs its longest unit's steps, so the
    // quads' runs are cut into 16 * P near-equal units (a piece more to whichever quad has the longest pieces) and the units
    // dealt to the passes by length; P = the pass count with the fewest steps in total (40 bands: one pass = 64 steps, two =
    // 24 + 24).  Per (pass, step, lane = 4*block + row) one weight (x 1/4: the kernel's power rows hold 4 |X|^2 -- the two
    // real spectra come out of one complex FFT as sums / differences of X[k] and X[N-k] without their 1/2); a band's sum is
    // the sum of its quad's unit partials in bin order.  Table (ints): [0] passes P, [1] quads NQ, [2+2p] steps of pass p,
    // [3+2p] its weights' offset, [10+32p+2b] first bin (a multiple of 8) of block b, [11+32p+2b] its unit number,
    // [138+q] first unit of quad q (NQ+1 entries).
    qtab.assign(WW_MELQ_TAB, 0);
    qw.clear();
    // Shorter transforms run on the same kernel: a frame of n_fft = 1024 / r samples under its own window, zero-extended to 1024
    // samples, has the n_fft-point spectrum at every r-th bin of its 1024-point one (up to a phase), so bin k's weight sits at
    // bin r k of this table and the bins between weigh nothing.
    if (n_fft <= WW_NFFT) {
        const int r = WW_NFFT / n_fft;
        struct Unit { int quad, g0, g1; };                           // bins [8*g0, 8*g1) of the 1024-point spectrum
        const int NQ = (M + 3) / 4, Pmin = (NQ + 15) / 16;
        std::vector<Unit> quads;
        for (int q = 0; q < NQ; ++q) {
            int s = WW_NFFT / 2 + 1, e = 0;
            for (int m = 4 * q; m < std::min(M, 4 * q + 4); ++m)
                if (len[m] > 0) { s = std::min(s, r * start[m]); e = std::max(e, r * (start[m] + len[m] - 1) + 1); }
            if (e <= s) { s = 0; e = 0; }
            quads.push_back(Unit{q, s / 8, (e + 7) / 8});
        }
        std::vector<Unit> units;
        std::vector<int> order;
        int best = -1;
        for (int P = Pmin; P <= WW_MELQ_MAX_PASSES; ++P) {
            // (at most 8 pieces of a quad hold bins -- the kernel sums a band's partials as 8 loads in flight; any more are empty)
            std::vector<int> pieces(NQ, 1);
            auto longest = [&](int q) { return pieces[q] >= 8 ? 0 : (quads[q].g1 - quads[q].g0 + pieces[q] - 1) / pieces[q]; };
            int fill = 0;
            for (int n = NQ; n < 16 * P; ++n) {
                int big = 0;
                for (int q = 1; q < NQ; ++q)
                    if (longest(q) > longest(big)) big = q;
                if (pieces[big] >= 8) ++fill; else ++pieces[big];
            }
            std::vector<Unit> us;
            for (int q = 0; q < NQ; ++q) {
                const int g = quads[q].g1 - quads[q].g0;
                for (int i = 0; i < pieces[q]; ++i) us.push_back(Unit{q, quads[q].g0 + g * i / pieces[q], quads[q].g0 + g * (i + 1) / pieces[q]});
                if (q == NQ - 1)
                    for (int i = 0; i < fill; ++i) us.push_back(Unit{q, quads[q].g1, quads[q].g1});
            }
            std::vector<int> ord(us.size());
            for (size_t i = 0; i < ord.size(); ++i) ord[i] = (int)i;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return us[x].g1 - us[x].g0 > us[y].g1 - us[y].g0; });
            int total = 0;
            for (int p = 0; p < P; ++p) total += 8 * (us[ord[16 * p]].g1 - us[ord[16 * p]].g0);
            if (best < 0 || total < best) { best = total; units = us; order = ord; }
        }
        const int P = (int)units.size() / 16;
        qtab[0] = P; qtab[1] = NQ;
        for (int p = 0; p < P; ++p) {
            const int steps = 8 * (units[order[16 * p]].g1 - units[order[16 * p]].g0);
            qtab[2 + 2 * p] = steps;
            qtab[3 + 2 * p] = (int32_t)qw.size();
            for (int b = 0; b < 16; ++b) {
                qtab[10 + 32 * p + 2 * b] = 8 * units[order[16 * p + b]].g0;
                qtab[11 + 32 * p + 2 * b] = order[16 * p + b];
            }
            for (int t = 0; t < steps; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    const Unit &u = units[order[16 * p + (lane >> 2)]];
                    const int m = 4 * u.quad + (lane & 3), j = 8 * u.g0 + t;
                    const int k = j / r;
                    const bool in = m < M && j < 8 * u.g1 && j % r == 0 && k >= start[m] && k < start[m] + len[m];
                    qw.push_back(in ? 0.25f * w[off[m] + (k - start[m])] : 0.f);
                }
        }
        for (int q = 0, i = 0; q <= NQ; ++q) {
            while (i < (int)units.size() && units[i].quad < q) ++i;
            qtab[138 + q] = i;
        }
    }
}

extern "C" int ww_feat_mel_tables(const ww_feat_cfg *cfg, int32_t *start, int32_t *len, float *w, int w_cap, int32_t *n_w,
                                  int32_t *melq_tab, float *melq_w, int melq_cap, int32_t *n_melq_w) {
    WW_REQUIRE(cfg && start && len && n_w, WW_E_INVALID, "ww_feat_mel_tables: null argument");
    WW_REQUIRE(cfg->n_fft >= 64 && cfg->n_fft <= 4096 && (cfg->n_fft & (cfg->n_fft - 1)) == 0 && cfg->n_mels >= 1 &&
               cfg->n_mels <= WW_MAX_MELS && cfg->sample_rate > 0, WW_E_INVALID, "ww_feat_mel_tables: bad configuration");
    MelTables mt;
    build_mel_tables(cfg, mt);
    memcpy(start, mt.start.data(), mt.start.size() * sizeof(int32_t));
    memcpy(len, mt.len.data(), mt.len.size() * sizeof(int32_t));
    *n_w = (int32_t)mt.w.size();
    if (w) {
        WW_REQUIRE(w_cap >= (int)mt.w.size(), WW_E_INVALID, "ww_feat_mel_tables: w holds %d floats, %d needed", w_cap, (int)mt.w.size());
        memcpy(w, mt.w.data(), mt.w.size() * sizeof(float));
    }
    if (n_melq_w) *n_melq_w = (int32_t)mt.qw.size();
    if (melq_tab) memcpy(melq_tab, mt.qtab.data(), WW_MELQ_TAB * sizeof(int32_t));
    if (melq_w) {
        WW_REQUIRE(melq_cap >= (int)mt.qw.size(), WW_E_INVALID, "ww_feat_mel_tables: melq_w holds %d floats, %d needed", melq_cap, (int)mt.qw.size());
        memcpy(melq_w, mt.qw.data(), mt.qw.size() * sizeof(float));
    }
    return WW_OK;
}

// Tables follow DESIGN.md "Feature spec" / oracle/features.py exactly (double on the host).
int ww_get_feat_tables(ww_ctx *ctx, const ww_feat_cfg *cfg, ww_feat_tables **out) {
    for (ww_feat_tables *t = ctx->tables; t; t = t->next) {
        if (memcmp(&t->cfg, cfg, sizeof(ww_feat_cfg)) == 0) {
            *out = t;
            return WW_OK;
        }
    }
    const int n_fft = cfg->n_fft, M = cfg->n_mels;
    // transforms up to 1024 points run on k_logmel: the frame's own periodic Hann in the middle of a 1024-sample window of
    // zeros (the frame then starts n_fft/2 before t * hop, as torch.stft(center=True) has it), the 1024-point twiddles
    const int n_tab = n_fft < WW_NFFT ? WW_NFFT : n_fft, lead = (n_tab - n_fft) / 2;
    std::vector<float> win(n_tab, 0.f);
    std::vector<float2> tw(n_tab);
    for (int i = 0; i < n_fft; ++i) win[lead + i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / n_fft));
    for (int i = 0; i < n_tab; ++i)
        tw[i] = make_float2((float)cos(2.0 * M_PI * i / n_tab), (float)(-sin(2.0 * M_PI * i / n_tab)));
    MelTables mt;
    build_mel_tables(cfg, mt);
    const std::vector<int32_t> &start = mt.start, &len = mt.len, &off = mt.off, &qtab = mt.qtab;
    const std::vector<float> &w = mt.w, &qw = mt.qw;
    const int max_len = mt.max_len;
    std::vector<float> dct;
    if (cfg->n_mfcc > 0) {
        dct.resize((size_t)cfg->n_mfcc * M);
        for (int c = 0; c < cfg->n_mfcc; ++c)
            for (int m = 0; m < M; ++m) {
                double v = cos(M_PI / M * (m + 0.5) * c) * sqrt(2.0 / M);
                if (c == 0) v *= 1.0 / sqrt(2.0);
                dct[(size_t)c * M + m] = (float)v;
            }
    }
    ww_feat_tables *t = new ww_feat_tables();
    memset(t, 0, sizeof(*t));
    t->cfg = *cfg;
    t->n_melq_w = (int32_t)qw.size();
    t->max_len = max_len;
    t->n_mel_w = (int32_t)w.size();
    int rc;
    if ((rc = upload(&t->window, win)) || (rc = upload(&t->twiddle, tw)) || (rc = upload(&t->mel_start, start)) ||
        (rc = upload(&t->mel_len, len)) || (rc = upload(&t->mel_off, off)) || (rc = upload(&t->mel_w, w)) ||
        (rc = upload(&t->melq_tab, qtab)) || (rc = upload(&t->melq_w, qw))) {
        free_tables(t);
        return rc;
    }
    std::vector<float> dct_t(dct.size());
    for (int c = 0; c < cfg->n_mfcc; ++c)
        for (int m = 0; m < M; ++m) dct_t[(size_t)m * cfg->n_mfcc + c] = dct[(size_t)c * M + m];
    if (cfg->n_mfcc > 0 && ((rc = upload(&t->dct, dct)) || (rc = upload(&t->dct_t, dct_t)))) {
        free_tables(t);
        return rc;
    }
    t->next = ctx->tables;
    ctx->tables = t;
    *out = t;
    return WW_OK;
}

extern "C" size_t ww_layer_scratch_bytes(void) {
    return (size_t)(WW_STAT_SLAB_FLOATS + WW_DW_SLAB_FLOATS) * sizeof(float);
}

// ------------------------------------------------------------------------------------------
// Finalize kernels.  Slab rows are block partials written by the conv kernels; columns are
// summed in double in a fixed order => bit-reproducible run to run.
// ------------------------------------------------------------------------------------------

// Column sums for ONE group of 8 channels: columns 8*cg..+7 (first statistic) and 64 + 8*cg..+7 (second statistic) of
// partials[rows][128], by one 1024-thread block: thread = (one of the 4 float4 of a row's 16 values, 1 of 256 row parts),
// so a slab of <= 1024 rows is one round of loads; the 256 parts are then summed in a fixed two-level order (16 x 16).
// The finalize kernels run 8 such blocks side by side instead of one block sweeping all 128 columns: the sweep of the
// 512 KB slab by a single CU was the whole cost of a finalize launch (7 us); channels are independent.
// tot[0..15]: [0..7] = first statistic, [8..15] = second, valid after the call in every thread of the 1024.
__device__ __forceinline__ void ww_colgroup_sum(const float *__restrict__ partials, int rows, int cg, double (*sh)[16],
                                                double *tot /*16, shared*/) {
    const int q = threadIdx.x & 3, part = threadIdx.x >> 2;
    {
        const int col = (q < 2 ? 8 * cg + 4 * q : 64 + 8 * cg + 4 * (q - 2));
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        // a thread's (<= 4: rows <= 1024) slab rows in one batch of unconditional, clamped loads: as a guarded loop the rows
        // beyond a multiple of four took the remainder loop -- one dependent L2 round trip each
        for (int r0 = part; r0 < rows; r0 += 4 * 256) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(partials + (size_t)min(r0 + 256 * u, rows - 1) * 128 + col);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r0 + 256 * u < rows) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
        }
        sh[part][4 * q] = a0; sh[part][4 * q + 1] = a1; sh[part][4 * q + 2] = a2; sh[part][4 * q + 3] = a3;
    }
    __syncthreads();
    const int c = threadIdx.x & 15, seg = (threadIdx.x >> 4) & 15;   // threads < 256: column c, parts 16*seg .. +15
    double t2 = 0.0;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int p = 0; p < 16; ++p) t2 += sh[16 * seg + p][c];
    }
    __syncthreads();
    if (threadIdx.x < 256) sh[seg][c] = t2;
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += sh[p][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// BatchNorm2d training forward statistics (torch semantics: biased var for normalisation,
// unbiased for running_var, running = (1-m)*running + m*batch).  grid 8 (channel groups), block 256
__global__ __launch_bounds__(1024) void k_bn_fwd_finalize(const float *__restrict__ partials, int rows,
                                                          double count, ww_bn_t bn, float *__restrict__ ss,
                                                          float *__restrict__ mr) {
    __shared__ double sh[256][16];
    __shared__ double tot[16];
    // the channel's parameters are fetched BEFORE the slab sums (behind them they were two more dependent round trips)
    float g_c = 0.f, b_c = 0.f, rm_c = 0.f, rv_c = 1.f;
    if (threadIdx.x < 8) {
        const int c = 8 * blockIdx.x + threadIdx.x;
        g_c = bn.gamma[c]; b_c = bn.beta[c];
        if (bn.running_mean) { rm_c = bn.running_mean[c]; rv_c = bn.running_var[c]; }
    }
    ww_colgroup_sum(partials, rows, blockIdx.x, sh, tot);
    if (threadIdx.x < 8) {
        const int c = 8 * blockIdx.x + threadIdx.x;
        const double mean = tot[threadIdx.x] / count;
        double var = tot[8 + threadIdx.x] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)bn.eps);
        const double scale = (double)g_c * rstd;
        ss[c] = (float)scale;
        ss[64 + c] = (float)((double)b_c - mean * scale);
        mr[c] = (float)mean;
        mr[64 + c] = (float)rstd;
        if (bn.running_mean) {
            const double m = bn.momentum;
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            bn.running_mean[c] = (float)((1.0 - m) * (double)rm_c + m * mean);
            bn.running_var[c] = (float)((1.0 - m) * (double)rv_c + m * unb);
        }
    }
}

__global__ void k_bn_eval_ss(ww_bn_t bn, float *__restrict__ ss, float *__restrict__ mr) {
    const int c = threadIdx.x;
    if (c < 64) {
        const double rstd = 1.0 / sqrt((double)bn.running_var[c] + (double)bn.eps);
        const double scale = (double)bn.gamma[c] * rstd;
        ss[c] = (float)scale;
        ss[64 + c] = (float)((double)bn.beta[c] - (double)bn.running_mean[c] * scale);
        mr[c] = bn.running_mean[c];
        mr[64 + c] = (float)rstd;
    }
}

// BatchNorm2d backward reduction: partial columns = [sum dz (64) | sum dz*yhat (64)].
//   dgamma = sum dz*yhat, dbeta = sum dz,
//   dy = gamma*rstd*(dz - mean(dz) - yhat*mean(dz*yhat))  ==  A*dz + Bc*y + Cc
// one channel group (8 channels) per 256-thread block
__device__ __forceinline__ void bn_bwd_finalize_group(const float *__restrict__ partials, int rows, double count, int cg,
                                                      const float *__restrict__ gamma, const float *__restrict__ mr,
                                                      float *__restrict__ coef, float *__restrict__ dgamma,
                                                      float *__restrict__ dbeta, double (*sh)[16], double *tot) {
    float mean_c = 0.f, rstd_c = 0.f, g_c = 0.f;        // fetched before the slab sums, not behind them
    if (threadIdx.x < 8) { const int c = 8 * cg + threadIdx.x; mean_c = mr[c]; rstd_c = mr[64 + c]; g_c = gamma[c]; }
    ww_colgroup_sum(partials, rows, cg, sh, tot);
    if (threadIdx.x < 8) {
        const int c = 8 * cg + threadIdx.x;
        const double s1 = tot[threadIdx.x], s2 = tot[8 + threadIdx.x];
        const double mean = mean_c, rstd = rstd_c, g = g_c;
        const double c1 = s1 / count, c2 = s2 / count;
        const double A = g * rstd;
        coef[c] = (float)A;
        coef[64 + c] = (float)(-A * rstd * c2);
        coef[128 + c] = (float)(A * (mean * rstd * c2 - c1));
        dgamma[c] = (float)s2;
        dbeta[c] = (float)s1;
    }
}

__global__ __launch_bounds__(1024) void k_bn_bwd_finalize(const float *__restrict__ partials, int rows,
                                                          double count, const float *__restrict__ gamma,
                                                          const float *__restrict__ mr, float *__restrict__ coef,
                                                          float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ double sh[256][16];
    __shared__ double tot[16];
    bn_bwd_finalize_group(partials, rows, count, blockIdx.x, gamma, mr, coef, dgamma, dbeta, sh, tot);
}

// generic column sum: out[col] = sum_r partials[r][col]; block = 64 columns (16 float4 groups) x 64 row parts.
// cols must be a multiple of 4 (576 and 4096 here).
__device__ __forceinline__ void colsum_body(const float *__restrict__ partials, int rows, int cols,
                                            float *__restrict__ out, int blk, double *sh /*64*64*/) {
    const int c4 = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int col = blk * 64 + 4 * c4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (col < cols)
        for (int r0 = part; r0 < rows; r0 += 8 * 64) {          // batches of eight unconditional, clamped row loads
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(partials + (size_t)min(r0 + 64 * u, rows - 1) * cols + col);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + 64 * u < rows) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
        }
    sh[part * 64 + 4 * c4] = a0; sh[part * 64 + 4 * c4 + 1] = a1;
    sh[part * 64 + 4 * c4 + 2] = a2; sh[part * 64 + 4 * c4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < 64 && blk * 64 + threadIdx.x < cols) {
        double t = 0.0;
#pragma unroll 8
        for (int p = 0; p < 64; ++p) t += sh[p * 64 + threadIdx.x];
        out[blk * 64 + threadIdx.x] = (float)t;
    }
}

__global__ __launch_bounds__(1024) void k_colsum(const float *__restrict__ partials, int rows, int cols,
                                                 float *__restrict__ out) {
    __shared__ double sh[64 * 64];
    colsum_body(partials, rows, cols, out, blockIdx.x, sh);
}

// one launch for a backward layer's two reductions: blocks 0..7 = BatchNorm-backward constants of the input layer (one
// channel group each), blocks 8.. = column sums of the weight-gradient slab
__global__ __launch_bounds__(1024) void k_bwd_finalize(const float *__restrict__ stat, int rows, double count,
                                                       const float *__restrict__ gamma, const float *__restrict__ mr,
                                                       float *__restrict__ coef, float *__restrict__ dgamma,
                                                       float *__restrict__ dbeta, const float *__restrict__ dwp,
                                                       int cols, float *__restrict__ dw) {
    __shared__ double sh[64 * 64];
    __shared__ double tot[16];
    if (blockIdx.x < 8) {
        bn_bwd_finalize_group(stat, rows, count, blockIdx.x, gamma, mr, coef, dgamma, dbeta,
                              reinterpret_cast<double(*)[16]>(sh), tot);
    } else {
        colsum_body(dwp, rows, cols, dw, blockIdx.x - 8, sh);
    }
}

int ww_launch_bn_fwd_finalize(const float *partials, int rows, double count, const ww_bn_t *bn, float *ss_out,
                              float *mr_out, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_fwd_finalize, dim3(8), dim3(1024), 0, st, partials, rows, count, *bn, ss_out, mr_out);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
int ww_launch_bn_eval_ss(const ww_bn_t *bn, float *ss_out, float *mr_out, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_eval_ss, dim3(1), dim3(64), 0, st, *bn, ss_out, mr_out);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
int ww_launch_bn_bwd_finalize(const float *partials, int rows, double count, const float *gamma,
                              const float *mr, float *coef_out, float *dgamma, float *dbeta, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(8), dim3(1024), 0, st, partials, rows, count, gamma, mr, coef_out,
                       dgamma, dbeta);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
int ww_launch_bwd_finalize(const float *stat, int rows, double count, const float *gamma, const float *mr,
                           float *coef, float *dgamma, float *dbeta, const float *dwp, int cols, float *dw,
                           hipStream_t st) {
    hipLaunchKernelGGL(k_bwd_finalize, dim3(8 + (cols + 63) / 64), dim3(1024), 0, st, stat, rows, count, gamma, mr, coef,
                       dgamma, dbeta, dwp, cols, dw);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
int ww_launch_colsum(const float *partials, int rows, int cols, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_colsum, dim3((cols + 63) / 64), dim3(1024), 0, st, partials, rows, cols, out);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

// ------------------------------------------------------------------------------------------
// clip_grad_norm_ on one flat bucket: a single block up to 65536 gradients (cnn_small: ~2.5e4), block partials + a
// grid-wide apply above that; deterministic fixed-order reductions
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_grad_norm_clip(float *__restrict__ g, size_t n, float max_norm,
                                                         float *__restrict__ norm_out,
                                                         ww_step_stats *__restrict__ stats) {
    __shared__ double sh[1024];
    __shared__ float coef_sh;
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const double v = g[i];
        acc += v * v;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(sh[0]);
        if (norm_out) *norm_out = norm;
        if (stats) {
            stats->grad_norm = norm;
            if (!isfinite(norm)) stats->found_inf = 1.0f;
        }
        float c = 1.0f;
        if (max_norm > 0.f) {
            c = max_norm / (norm + 1e-6f);   // torch: clamp(max_norm/(total_norm+1e-6), max=1)
            if (c > 1.0f) c = 1.0f;          // NaN compares false -> NaN propagates like torch
        }
        coef_sh = c;
    }
    __syncthreads();
    const float c = coef_sh;
    if (max_norm > 0.f)
        for (size_t i = threadIdx.x; i < n; i += 1024) g[i] *= c;
}

// ---- large buckets (MobileNetV3-small: 1.5e6 gradients): block partial sums in a fixed partition, then every block of
// the consumer reduces the <= 256 partials in the same order (deterministic, no atomics)
__global__ __launch_bounds__(1024) void k_sumsq_partials(const float *__restrict__ g, size_t n, double *__restrict__ parts) {
    __shared__ double sh[1024];
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double acc = 0.0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const double v = g[i];
        acc += v * v;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) parts[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void k_norm_clip_apply(float *__restrict__ g, size_t n, const double *__restrict__ parts,
                                                         int nparts, float max_norm, float *__restrict__ norm_out,
                                                         ww_step_stats *__restrict__ stats) {
    double t = 0.0;
    for (int i = 0; i < nparts; ++i) t += parts[i];          // same order in every thread
    const float norm = (float)sqrt(t);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (norm_out) *norm_out = norm;
        if (stats) {
            stats->grad_norm = norm;
            if (!isfinite(norm)) stats->found_inf = 1.0f;
        }
    }
    if (!(max_norm > 0.f)) return;
    float c = max_norm / (norm + 1e-6f);
    if (c > 1.0f) c = 1.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] *= c;
}

int ww_launch_sumsq_partials(ww_ctx *ctx, const float *g, size_t n, int *parts_out, hipStream_t st) {
    if (!ctx->norm_partials) WW_HIP(hipMalloc((void **)&ctx->norm_partials, WW_NORM_PARTS * sizeof(double)));
    const int parts = (int)std::min<size_t>(WW_NORM_PARTS, (n + 4095) / 4096);
    hipLaunchKernelGGL(k_sumsq_partials, dim3(parts), dim3(1024), 0, st, g, n, ctx->norm_partials);
    WW_LAUNCH_CHECK();
    *parts_out = parts;
    return WW_OK;
}

extern "C" int ww_grad_norm_clip(ww_ctx *ctx, float *flat_grads, size_t n, float max_norm, float *norm_out,
                                 ww_step_stats *stats, ww_stream_t stream) {
    WW_REQUIRE(ctx && flat_grads, WW_E_INVALID, "ww_grad_norm_clip: null argument");
    if (n == 0) return WW_OK;
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_CLIP, st);
    if (n <= (size_t)1 << 16) {
        hipLaunchKernelGGL(k_grad_norm_clip, dim3(1), dim3(1024), 0, st, flat_grads, n, max_norm, norm_out, stats);
        WW_LAUNCH_CHECK();
        return WW_OK;
    }
    int parts = 0;
    const int rc = ww_launch_sumsq_partials(ctx, flat_grads, n, &parts, st);
    if (rc) return rc;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_norm_clip_apply, dim3(grid), dim3(256), 0, st, flat_grads, n, ctx->norm_partials, parts, max_norm,
                       norm_out, stats);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

// Grid for a persistent (grid-stride) kernel: one full residency wave of blocks, capped by the
// amount of work and by the reduction slab.  Speed only -- no kernel relies on co-residency.
int ww_occupancy_grid(const void *fn, int block, size_t smem, long want, int cap) {
    static int n_cu = 0;
    struct Entry { const void *fn; size_t smem; int per_cu; };
    static Entry cache[32];
    static int n_cache = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        else
            n_cu = 256;
    }
    int per_cu = 0;
    for (int i = 0; i < n_cache; ++i)
        if (cache[i].fn == fn && cache[i].smem == smem) per_cu = cache[i].per_cu;
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, smem) != hipSuccess || per_cu < 1) per_cu = 1;
        if (n_cache < 32) cache[n_cache++] = Entry{fn, smem, per_cu};
    }
    long gsz = (long)per_cu * n_cu;
    if (gsz > cap) gsz = cap;
    if (gsz > want) gsz = want;
    return (int)(gsz < 1 ? 1 : gsz);
}

// ------------------------------------------------------------------------------------------
static const char *k_class_names[WW_K_NCLASS] = {"logmel_specaug", "conv_stem_fwd", "dwconv3x3_fwd", "pwconv1x1_fwd",
                                                 "gap_fwd", "head_loss", "pwconv1x1_bwd", "dwconv3x3_bwd",
                                                 "conv_stem_bwd", "finalize", "grad_norm_clip", "audio_augment", "linear_mfma", "gru", "nhwc_layers"};
extern "C" int ww_prof_num_classes(void) { return WW_K_NCLASS; }
extern "C" const char *ww_prof_class_name(int cls) { return (cls >= 0 && cls < WW_K_NCLASS) ? k_class_names[cls] : ""; }
extern "C" int ww_prof_enable(ww_ctx *ctx, uint32_t class_mask) {
    WW_REQUIRE(ctx != nullptr, WW_E_INVALID, "ww_prof_enable: ctx is null");
    ctx->prof_mask = class_mask;
    return WW_OK;
}
extern "C" int ww_prof_collect(ww_ctx *ctx, float *ms_sum, int32_t *count) {
    WW_REQUIRE(ctx && ms_sum && count, WW_E_INVALID, "ww_prof_collect: null argument");
    for (int i = 0; i < WW_K_NCLASS; ++i) { ms_sum[i] = 0.f; count[i] = 0; }
    for (auto &r : *ctx->prof_recs) {
        WW_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        WW_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        ms_sum[r.cls] += ms;
        count[r.cls] += 1;
        ctx->prof_free->push_back(r);
    }
    ctx->prof_recs->clear();
    return WW_OK;
}
