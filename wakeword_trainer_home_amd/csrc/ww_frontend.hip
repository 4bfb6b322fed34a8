// Batched STFT -> power -> HTK mel -> log (-> DCT-II) with SpecAugment fused into the
// write-out.  One 256-thread workgroup = one clip x 16 consecutive frames.
//
// Data movement (per workgroup): the 16 frames' sample span (15*hop + 1024 samples, reflect
// padded at the clip edges) is read ONCE from HBM with coalesced loads into LDS; every later
// access (6.4x frame overlap at hop 160) is served from LDS.  Each wavefront transforms TWO
// real frames as one 1024-point complex FFT held 16 points/lane, decomposed 16 x 16 x 4:
//   pass 1  radix-16 in registers over n2 (n = lane + 64*n2), twiddle W1024^(lane*kb) -> LDS
//   pass 2  radix-16 in registers over m  (lane = kb*4+q, n1 = 4m+q), twiddle W64^(q*kc) -> LDS
//   pass 3  four radix-4 butterflies per lane -> X[16kc + 256kd + kb] -> LDS (natural order)
// (passes 2 and 3 run IN PLACE: each lane rewrites exactly the 16 LDS slots it read, so one 8.7 KB buffer per
// wavefront suffices and two workgroups fit a CU).  Then the two real spectra are separated (X[k], conj X[N-k]),
// |.|^2 goes to LDS, every mel band is summed by two adjacent lanes (half a band each, weights staged in LDS,
// combined with one shuffle), and the 16 x F tile is transposed through LDS so the (B,1,F,T) output is
// written in 64-byte runs along T.
// Index algebra verified against numpy (tests/test_frontend_index_algebra.py).
#include "ww_internal.h"

namespace {

constexpr int FR = WW_FRAMES_PER_BLOCK;
constexpr int BUF_STRIDE = 68;               // float2 per kb row (64 + 4 pad)
constexpr int BUF_ELEMS = 16 * BUF_STRIDE;   // 1088 float2 per buffer

#include "ww_fft.h"

__device__ __forceinline__ float load_sample(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float load_sample(const int16_t *p, size_t i) { return (float)p[i] * (1.0f / 32768.0f); }

struct FeatArgs {
    int B, N, hop, T, M, F;  // F = output feature rows (n_mfcc or M)
    int use_dct;
    float log_eps;
    const float *window;
    const float2 *twiddle;
    const int32_t *mel_start, *mel_len, *mel_off;
    const float *mel_w;
    const float *dct;
    int span_len;
    int n_mel_w;
};

// order a wave's own LDS traffic (cross-lane hand-off inside one wavefront): LDS processes one wave's operations in
// issue order, so only the compiler has to be kept from reordering.  A wavefront touches only its own FFT / power
// buffers inside a round, so no workgroup barrier is needed there (PMC: 47 % of wave-cycles were parked).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename WaveT>
__global__ __launch_bounds__(256) void k_logmel(const WaveT *__restrict__ wave, FeatArgs a, float *__restrict__ out,
                                                int use_mask, ww_mask_params mp, int32_t *__restrict__ mask_idx) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);                    // 1024
    float2 *fbuf = tw + 1024;                                         // 4 waves x BUF_ELEMS
    float *pb = reinterpret_cast<float *>(fbuf + 4 * BUF_ELEMS);      // 4 waves x 2 x 516 power spectra
    float *lm = pb + 4 * 2 * 516;                                     // FR x M
    float *feat = lm + FR * a.M;                                      // FR x F (== lm when !use_dct)
    int *msk = reinterpret_cast<int *>(feat + (a.use_dct ? FR * a.F : 0));  // 2*WW_MAX_MASKS
    int *mtab = msk + 2 * WW_MAX_MASKS;                               // 3*M : start, len, offset
    float *mw = reinterpret_cast<float *>(mtab + 3 * a.M);            // n_mel_w band weights
    float *span = mw + a.n_mel_w;                                     // span_len
    if (!a.use_dct) feat = lm;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * FR;
    const WaveT *x = wave + (size_t)b * a.N;

    // ---- stage twiddles, SpecAugment masks and the sample span
    for (int i = tid; i < 1024; i += 256) tw[i] = a.twiddle[i];
    for (int i = tid; i < a.M; i += 256) {
        mtab[3 * i] = a.mel_start[i];
        mtab[3 * i + 1] = a.mel_len[i];
        mtab[3 * i + 2] = a.mel_off[i];
    }
    for (int i = tid; i < a.n_mel_w; i += 256) mw[i] = a.mel_w[i];
    const int K = use_mask ? (mp.n_f + mp.n_t) : 0;
    if (tid < K) {
        int s, w;
        ww_specaug_mask(mp, (uint32_t)(mp.sample_offset + (uint64_t)b), tid, a.F, a.T, s, w);
        msk[2 * tid] = s;
        msk[2 * tid + 1] = w;
        if (mask_idx && blockIdx.x == 0) {
            mask_idx[((size_t)b * K + tid) * 2] = s;
            mask_idx[((size_t)b * K + tid) * 2 + 1] = w;
        }
    }
    {
        const long base = (long)t0 * a.hop - WW_NFFT / 2;
        for (int i = tid; i < a.span_len; i += 256) {
            long idx = base + i;
            if (idx < 0) idx = -idx;
            if (idx >= a.N) idx = 2L * (a.N - 1) - idx;
            idx = idx < 0 ? 0 : (idx >= a.N ? a.N - 1 : idx);  // only frames >= T can get here
            span[i] = load_sample(x, (size_t)idx);
        }
    }
    __syncthreads();

    float2 *buf = fbuf + wv * BUF_ELEMS;
    float *pbuf = pb + wv * 2 * 516;

    for (int round = 0; round < 2; ++round) {
        const int fa = round * 8 + 2 * wv, fb = fa + 1;  // local frame indices of this wave's pair
        float re[16], im[16];
        // ---- pass 1: windowed load, radix-16 over n2, twiddle W1024^(lane*kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = lane + 64 * j;
            const float w = a.window[n];
            re[j] = w * span[fa * a.hop + n];
            im[j] = w * span[fb * a.hop + n];
        }
        fft16(re, im);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            float r = re[F16_SLOT(kb)], i = im[F16_SLOT(kb)];
            const float2 t = tw[(lane * kb) & 1023];
            cmul_c(r, i, t.x, t.y);
            buf[kb * BUF_STRIDE + lane] = make_float2(r, i);
        }
        wave_sync();
        // ---- pass 2: lane = kb*4 + q ; radix-16 over m (n1 = 4m + q), twiddle W64^(q*kc)
        const int kb2 = lane >> 2, q = lane & 3;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 v = buf[kb2 * BUF_STRIDE + 4 * m + q];
            re[m] = v.x;
            im[m] = v.y;
        }
        fft16(re, im);
#pragma unroll
        for (int kc = 0; kc < 16; ++kc) {
            float r = re[F16_SLOT(kc)], i = im[F16_SLOT(kc)];
            const float2 t = tw[(16 * q * kc) & 1023];
            cmul_c(r, i, t.x, t.y);
            buf[kb2 * BUF_STRIDE + kc * 4 + q] = make_float2(r, i);
        }
        wave_sync();
        // ---- pass 3 (in place): radix-4 over q for (kb, kc = (lane&3) + 4u); slot [kb][4kc + kd] <- X[16kc + 256kd + kb]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float2 *s4 = buf + kb2 * BUF_STRIDE + ((lane & 3) + 4 * u) * 4;
            const float2 v0 = s4[0], v1 = s4[1], v2 = s4[2], v3 = s4[3];
            float r0 = v0.x, i0 = v0.y, r1 = v1.x, i1 = v1.y, r2 = v2.x, i2 = v2.y, r3 = v3.x, i3 = v3.y;
            fft4(r0, i0, r1, i1, r2, i2, r3, i3);
            s4[0] = make_float2(r0, i0);
            s4[1] = make_float2(r1, i1);
            s4[2] = make_float2(r2, i2);
            s4[3] = make_float2(r3, i3);
        }
        wave_sync();
        // ---- separate the two real spectra, power -> pbuf[frame][k];  X[k] lives at [k&15][4*((k>>4)&15) + (k>>8)]
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int k = lane + 64 * u;
            if (k <= 512) {
                const int kn = (1024 - k) & 1023;
                const float2 A = buf[(k & 15) * BUF_STRIDE + 4 * ((k >> 4) & 15) + (k >> 8)];
                const float2 Bv = buf[(kn & 15) * BUF_STRIDE + 4 * ((kn >> 4) & 15) + (kn >> 8)];
                const float xr = A.x + Bv.x, xi = A.y - Bv.y;   // 2*Xa
                const float yr = A.y + Bv.y, yi = A.x - Bv.x;   // 2*Xb (up to sign of imag)
                pbuf[k] = 0.25f * (xr * xr + xi * xi);
                pbuf[516 + k] = 0.25f * (yr * yr + yi * yi);
            }
        }
        wave_sync();
        // ---- mel band sums: item = ((frame, mel), half); the two halves of a band sit in adjacent lanes
        for (int it0 = 0; it0 < 4 * a.M; it0 += 64) {
            const int it = it0 + lane;
            const bool act = it < 4 * a.M;
            const int pair = act ? it >> 1 : 0, half = it & 1;
            const int fr = pair >= a.M ? 1 : 0;
            const int m = pair - fr * a.M;
            const int s = mtab[3 * m], L = mtab[3 * m + 1];
            const int h0 = (L + 1) >> 1;
            const int j0 = half ? h0 : 0, j1 = half ? L : h0;
            const float *wp = mw + mtab[3 * m + 2];
            const float *pp = pbuf + fr * 516 + s;
            float acc = 0.f;
            if (act) {
#pragma unroll 4
                for (int j = j0; j < j1; ++j) acc = fmaf(wp[j], pp[j], acc);
            }
            acc += __shfl_xor(acc, 1);
            if (act && half == 0) lm[(fa + fr) * a.M + m] = logf(acc + a.log_eps);
        }
        wave_sync();
    }
    __syncthreads();

    if (a.use_dct) {
        for (int it = tid; it < FR * a.F; it += 256) {
            const int fr = it / a.F, c = it - fr * a.F;
            const float *d = a.dct + (size_t)c * a.M;
            const float *l = lm + fr * a.M;
            float acc = 0.f;
            for (int m = 0; m < a.M; ++m) acc = fmaf(d[m], l[m], acc);
            feat[it] = acc;
        }
        __syncthreads();
    }

    // ---- masked, transposed write-out: out[b][0][f][t0 + i]
    for (int it = tid; it < a.F * FR; it += 256) {
        const int f = it / FR, i = it - f * FR;
        const int t = t0 + i;
        if (t < a.T) {
            float v = feat[i * a.F + f];
            for (int k = 0; k < K; ++k) {
                const int s = msk[2 * k], w = msk[2 * k + 1];
                const int pos = k < mp.n_f ? f : t;
                if (pos >= s && pos < s + w) v = 0.f;
            }
            out[((size_t)b * a.F + f) * a.T + t] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_specaug_apply(float *__restrict__ x, int B, int F, int T, ww_mask_params mp,
                                                       int32_t *__restrict__ mask_idx) {
    __shared__ int msk[2 * WW_MAX_MASKS];
    const int b = blockIdx.x, K = mp.n_f + mp.n_t;
    if ((int)threadIdx.x < K) {
        int s, w;
        ww_specaug_mask(mp, (uint32_t)(mp.sample_offset + (uint64_t)b), threadIdx.x, F, T, s, w);
        msk[2 * threadIdx.x] = s;
        msk[2 * threadIdx.x + 1] = w;
        if (mask_idx) {
            mask_idx[((size_t)b * K + threadIdx.x) * 2] = s;
            mask_idx[((size_t)b * K + threadIdx.x) * 2 + 1] = w;
        }
    }
    __syncthreads();
    float *xb = x + (size_t)b * F * T;
    // zero only the masked bands: rows for frequency masks, column runs for time masks
    for (int k = 0; k < K; ++k) {
        const int s = msk[2 * k], w = msk[2 * k + 1];
        if (w == 0) continue;
        if (k < mp.n_f) {
            for (int i = threadIdx.x; i < w * T; i += blockDim.x) xb[(size_t)s * T + i] = 0.f;
        } else {
            for (int i = threadIdx.x; i < F * w; i += blockDim.x) {
                const int f = i / w, j = i - f * w;
                xb[(size_t)f * T + s + j] = 0.f;
            }
        }
    }
}

int resolve_mask(const ww_specaug_cfg *sa, uint64_t seed, uint64_t step, uint64_t sample_offset, ww_mask_params *mp) {
    WW_REQUIRE(sa->n_freq_masks >= 0 && sa->n_time_masks >= 0 &&
                   sa->n_freq_masks + sa->n_time_masks <= WW_MAX_MASKS,
               WW_E_INVALID, "specaug: n_freq_masks + n_time_masks must be in [0,%d]", WW_MAX_MASKS);
    WW_REQUIRE(sa->freq_mask_param >= 0 && sa->time_mask_param >= 0, WW_E_INVALID,
               "specaug: mask params must be >= 0");
    mp->n_f = sa->n_freq_masks;
    mp->n_t = sa->n_time_masks;
    mp->f_param = sa->freq_mask_param;
    mp->t_param = sa->time_mask_param;
    mp->f_thresh = ww_prob_threshold((double)sa->freq_mask_prob);
    mp->t_thresh = ww_prob_threshold((double)sa->time_mask_prob);
    mp->seed_lo = (uint32_t)seed;
    mp->seed_hi = (uint32_t)(seed >> 32);
    mp->step_lo = (uint32_t)step;
    mp->step_hi = (uint32_t)(step >> 32);
    mp->sample_offset = sample_offset;
    return WW_OK;
}

}  // namespace

extern "C" int ww_logmel_fwd(ww_ctx *ctx, const void *wave, int wave_dtype, int B, int N, const ww_feat_cfg *cfg,
                             float *out, const ww_specaug_cfg *sa, uint64_t seed, uint64_t step,
                             uint64_t sample_offset, int32_t *mask_idx, ww_stream_t stream) {
    WW_REQUIRE(ctx && wave && cfg && out, WW_E_INVALID, "ww_logmel_fwd: null argument");
    WW_REQUIRE(B >= 0 && N >= 0, WW_E_INVALID, "ww_logmel_fwd: negative shape");
    WW_REQUIRE(cfg->n_fft == WW_NFFT, WW_E_UNSUPPORTED, "ww_logmel_fwd: n_fft=%d (only %d is implemented)",
               cfg->n_fft, WW_NFFT);
    WW_REQUIRE(cfg->hop >= 1 && cfg->hop <= WW_MAX_HOP, WW_E_UNSUPPORTED, "ww_logmel_fwd: hop=%d not in [1,%d]",
               cfg->hop, WW_MAX_HOP);
    WW_REQUIRE(cfg->n_mels >= 1 && cfg->n_mels <= WW_MAX_MELS, WW_E_UNSUPPORTED,
               "ww_logmel_fwd: n_mels=%d not in [1,%d]", cfg->n_mels, WW_MAX_MELS);
    WW_REQUIRE(cfg->n_mfcc >= 0 && cfg->n_mfcc <= cfg->n_mels, WW_E_INVALID,
               "ww_logmel_fwd: n_mfcc=%d not in [0,n_mels=%d]", cfg->n_mfcc, cfg->n_mels);
    WW_REQUIRE(cfg->sample_rate > 0, WW_E_INVALID, "ww_logmel_fwd: sample_rate must be > 0");
    WW_REQUIRE(wave_dtype == WW_WAVE_F32 || wave_dtype == WW_WAVE_I16, WW_E_INVALID,
               "ww_logmel_fwd: wave_dtype %d", wave_dtype);
    if (B == 0) return WW_OK;
    // reflect padding needs pad < N, as torch.stft(center=True, pad_mode='reflect') does
    WW_REQUIRE(N > WW_NFFT / 2, WW_E_INVALID, "ww_logmel_fwd: N=%d must exceed n_fft/2=%d for reflect padding", N,
               WW_NFFT / 2);
    ww_feat_tables *tb = nullptr;
    int rc = ww_get_feat_tables(ctx, cfg, &tb);
    if (rc) return rc;
    FeatArgs a;
    a.B = B; a.N = N; a.hop = cfg->hop; a.T = 1 + N / cfg->hop; a.M = cfg->n_mels;
    a.use_dct = cfg->n_mfcc > 0;
    a.F = a.use_dct ? cfg->n_mfcc : cfg->n_mels;
    a.log_eps = cfg->log_eps;
    a.window = tb->window; a.twiddle = tb->twiddle;
    a.mel_start = tb->mel_start; a.mel_len = tb->mel_len; a.mel_off = tb->mel_off; a.mel_w = tb->mel_w;
    a.dct = tb->dct;
    a.span_len = (FR - 1) * cfg->hop + WW_NFFT;
    a.n_mel_w = tb->n_mel_w;
    ww_mask_params mp = {};
    int use_mask = 0;
    if (sa) {
        if ((rc = resolve_mask(sa, seed, step, sample_offset, &mp))) return rc;
        use_mask = (mp.n_f + mp.n_t) > 0;
    }
    const size_t smem = 1024 * sizeof(float2) + (size_t)4 * BUF_ELEMS * sizeof(float2) + (size_t)4 * 2 * 516 * sizeof(float) +
                        (size_t)FR * a.M * sizeof(float) + (a.use_dct ? (size_t)FR * a.F * sizeof(float) : 0) +
                        2 * WW_MAX_MASKS * sizeof(int) + (size_t)3 * a.M * sizeof(int) + (size_t)a.n_mel_w * sizeof(float) +
                        (size_t)a.span_len * sizeof(float);
    dim3 grid((a.T + FR - 1) / FR, B);
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_LOGMEL, st);
    if (wave_dtype == WW_WAVE_F32) {
        WW_HIP(hipFuncSetAttribute((const void *)k_logmel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)smem));
        hipLaunchKernelGGL(k_logmel<float>, grid, dim3(256), smem, st, (const float *)wave, a, out, use_mask, mp,
                           mask_idx);
    } else {
        WW_HIP(hipFuncSetAttribute((const void *)k_logmel<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)smem));
        hipLaunchKernelGGL(k_logmel<int16_t>, grid, dim3(256), smem, st, (const int16_t *)wave, a, out, use_mask, mp,
                           mask_idx);
    }
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" int ww_specaug_apply(ww_ctx *ctx, float *x, int B, int F, int T, const ww_specaug_cfg *sa, uint64_t seed,
                                uint64_t step, uint64_t sample_offset, int32_t *mask_idx, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && sa, WW_E_INVALID, "ww_specaug_apply: null argument");
    WW_REQUIRE(B >= 0 && F >= 1 && T >= 1, WW_E_INVALID, "ww_specaug_apply: bad shape (%d,%d,%d)", B, F, T);
    ww_mask_params mp = {};
    int rc = resolve_mask(sa, seed, step, sample_offset, &mp);
    if (rc) return rc;
    if (B == 0 || mp.n_f + mp.n_t == 0) return WW_OK;
    hipLaunchKernelGGL(k_specaug_apply, dim3(B), dim3(256), 0, (hipStream_t)stream, x, B, F, T, mp, mask_idx);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
