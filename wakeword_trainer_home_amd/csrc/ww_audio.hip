// On-GPU waveform augmentation: RIR convolution + background-noise mix at a random SNR.
// Spec: DESIGN.md "Audio augmentation spec" / oracle/audio_augment.py (the reference's AudioAugmentation source is absent).
//
// k_aug_conv : grid (tiles of 2048 samples, clips).  Clips whose Philox draw selects a RIR are convolved directly in the
//              time domain: the tile's input history (2048 + L samples, zero before the clip start) and the RIR are
//              staged in LDS; a thread owns 8 consecutive outputs and consumes 8 taps per step from a 16-sample register
//              window (64 FMAs per 6 ds_read_b128, RIR reads are wave-wide broadcasts).  Other clips are copied.  Per-tile
//              sums of x^2, y^2 and n^2 (noise segment) go to a slab.
// k_aug_mix  : per clip, fixed-order sum of the tile partials -> loudness-preserving scale of the reverberated signal and
//              the noise gain for the drawn SNR; out = clip(y*scale + gain*n, -1, 1).
// (First version: O(N*L) direct form, exact fp32.  An FFT overlap-save form reusing k_logmel's FFT passes is the next step
//  for RIRs beyond a few thousand taps.)
#include "ww_internal.h"

namespace {

constexpr int AUG_TILE = 2048;
constexpr uint32_t TAG_AUDIO = 2u;

struct AugParams {
    int B, N, R, L, Lp, K, Nn, nt;
    uint64_t rir_thresh, noise_thresh;
    float snr_min, snr_max;
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
    uint64_t sample_offset;
};
struct AugChoice { int rir, noise, offset; float snr_db; };

__device__ __forceinline__ AugChoice aug_choice(const AugParams &p, int b) {
    const uint32_t g = (uint32_t)(p.sample_offset + (uint64_t)b);
    uint32_t r0[4], r1[4];
    ww_philox(p.step_lo, p.step_hi, g, (TAG_AUDIO << 24) | 0u, p.seed_lo, p.seed_hi, r0);
    ww_philox(p.step_lo, p.step_hi, g, (TAG_AUDIO << 24) | 1u, p.seed_lo, p.seed_hi, r1);
    AugChoice c;
    c.rir = (p.R > 0 && (uint64_t)r0[0] < p.rir_thresh) ? (int)(r0[1] % (uint32_t)p.R) : -1;
    c.noise = (p.K > 0 && (uint64_t)r1[0] < p.noise_thresh) ? (int)(r1[1] % (uint32_t)p.K) : -1;
    c.offset = p.K > 0 ? (int)(r1[2] % (uint32_t)(p.Nn - p.N + 1)) : 0;
    const float u = (float)(r1[3] >> 8) * 5.9604644775390625e-08f;   // 2^-24, exact
    c.snr_db = fmaf(u, p.snr_max - p.snr_min, p.snr_min);
    return c;
}

__device__ __forceinline__ float block_sum(float v, float *sh) {
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void k_aug_conv(const float *__restrict__ x, const float *__restrict__ rirs,
                                                  const float *__restrict__ noises, AugParams p,
                                                  float *__restrict__ out, float *__restrict__ stats,
                                                  int32_t *__restrict__ choice_out) {
    extern __shared__ __align__(16) float aug_lds[];
    float *xs = aug_lds;                       // AUG_TILE + Lp : xs[i] = x[t0 - Lp + i]
    float *hs = aug_lds + AUG_TILE + p.Lp;     // Lp
    __shared__ float red[4];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * AUG_TILE;
    const AugChoice c = aug_choice(p, b);
    if (choice_out && blockIdx.x == 0 && tid == 0) {
        choice_out[4 * b] = c.rir; choice_out[4 * b + 1] = c.noise; choice_out[4 * b + 2] = c.offset;
        choice_out[4 * b + 3] = __float_as_int(c.snr_db);
    }
    const float *xb = x + (size_t)b * p.N;
    const int o = 8 * tid;
    float y[8], xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = t0 + o + j;
        xv[j] = t < p.N ? xb[t] : 0.f;
        y[j] = xv[j];
    }
    if (c.rir >= 0) {
        const float *h = rirs + (size_t)c.rir * p.L;
        for (int i = tid; i < AUG_TILE + p.Lp; i += 256) {
            const int t = t0 - p.Lp + i;
            xs[i] = (t >= 0 && t < p.N) ? xb[t] : 0.f;
        }
        for (int k = tid; k < p.Lp; k += 256) hs[k] = k < p.L ? h[k] : 0.f;
        __syncthreads();
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < p.Lp; k0 += 8) {
            // window w[i] = x[t0 + o - k0 - 8 + i], i = 0..15 ; tap k0+kk of output j uses w[j + 8 - kk]
            const float4 *wp = reinterpret_cast<const float4 *>(xs + o + p.Lp - k0 - 8);
            const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
            const float w[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
            const float4 h0 = *reinterpret_cast<const float4 *>(hs + k0), h1 = *reinterpret_cast<const float4 *>(hs + k0 + 4);
            const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(hv[kk], w[j + 8 - kk], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = acc[j];
    }
    float sx = 0.f, sy = 0.f, sn = 0.f;
    const float *nb = c.noise >= 0 ? noises + (size_t)c.noise * p.Nn + c.offset : nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = t0 + o + j;
        if (t < p.N) {
            out[(size_t)b * p.N + t] = y[j];
            sx = fmaf(xv[j], xv[j], sx);
            sy = fmaf(y[j], y[j], sy);
            if (nb) { const float nv = nb[t]; sn = fmaf(nv, nv, sn); }
        }
    }
    sx = block_sum(sx, red);
    sy = block_sum(sy, red);
    sn = block_sum(sn, red);
    if (tid == 0) {
        float *s = stats + ((size_t)b * p.nt + blockIdx.x) * 4;
        s[0] = sx; s[1] = sy; s[2] = sn; s[3] = 0.f;
    }
}

__global__ __launch_bounds__(256) void k_aug_mix(const float *__restrict__ noises, AugParams p, float *__restrict__ out,
                                                 const float *__restrict__ stats) {
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * AUG_TILE;
    const AugChoice c = aug_choice(p, b);
    double sx = 0.0, sy = 0.0, sn = 0.0;
    for (int i = 0; i < p.nt; ++i) {          // fixed order -> every tile of the clip derives identical gains
        const float *s = stats + ((size_t)b * p.nt + i) * 4;
        sx += s[0]; sy += s[1]; sn += s[2];
    }
    float scale = 1.f;
    if (c.rir >= 0 && sy > 0.0) scale = (float)sqrt(sx / sy);
    float gain = 0.f;
    if (c.noise >= 0 && sn > 0.0) {
        const double ry = sqrt(sy / p.N) * (double)scale, rn = sqrt(sn / p.N);
        gain = (float)(ry / (rn * pow(10.0, (double)c.snr_db / 20.0)));
    }
    const float *nb = c.noise >= 0 ? noises + (size_t)c.noise * p.Nn + c.offset : nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = t0 + 8 * tid + j;
        if (t < p.N) {
            float v = out[(size_t)b * p.N + t] * scale;
            if (nb) v = fmaf(gain, nb[t], v);
            out[(size_t)b * p.N + t] = fminf(fmaxf(v, -1.f), 1.f);
        }
    }
}

}  // namespace

extern "C" size_t ww_audio_augment_scratch_bytes(int B, int N) {
    if (B < 1 || N < 1) return 0;
    return (size_t)B * ((N + AUG_TILE - 1) / AUG_TILE) * 4 * sizeof(float);
}

extern "C" int ww_audio_augment(ww_ctx *ctx, const float *wave_in, float *wave_out, int B, int N, const float *rirs,
                                int R, int L, const float *noises, int K, int Nn, const ww_audio_aug_cfg *cfg,
                                uint64_t seed, uint64_t step, uint64_t sample_offset, int32_t *choice_out,
                                void *scratch, size_t scratch_bytes, ww_stream_t stream) {
    WW_REQUIRE(ctx && wave_in && wave_out && cfg && scratch, WW_E_INVALID, "ww_audio_augment: null argument");
    WW_REQUIRE(wave_in != wave_out, WW_E_INVALID, "ww_audio_augment: in-place operation is not supported");
    WW_REQUIRE(B >= 1 && N >= 1, WW_E_INVALID, "ww_audio_augment: bad shape (%d,%d)", B, N);
    WW_REQUIRE((R == 0) == (rirs == nullptr) && R >= 0, WW_E_INVALID, "ww_audio_augment: RIR bank pointer/count mismatch");
    WW_REQUIRE((K == 0) == (noises == nullptr) && K >= 0, WW_E_INVALID, "ww_audio_augment: noise bank pointer/count mismatch");
    WW_REQUIRE(R == 0 || (L >= 1 && L <= 8192), WW_E_UNSUPPORTED, "ww_audio_augment: RIR length %d not in [1,8192]", L);
    WW_REQUIRE(K == 0 || Nn >= N, WW_E_INVALID, "ww_audio_augment: noise clips (%d samples) must be at least as long as N=%d", Nn, N);
    WW_REQUIRE(cfg->snr_max_db >= cfg->snr_min_db, WW_E_INVALID, "ww_audio_augment: snr_max < snr_min");
    WW_REQUIRE(scratch_bytes >= ww_audio_augment_scratch_bytes(B, N), WW_E_WORKSPACE, "ww_audio_augment: scratch too small");
    AugParams p;
    p.B = B; p.N = N; p.R = R; p.L = R ? L : 0; p.Lp = R ? (L + 7) / 8 * 8 : 8; p.K = K; p.Nn = Nn;
    p.nt = (N + AUG_TILE - 1) / AUG_TILE;
    p.rir_thresh = ww_prob_threshold((double)cfg->rir_prob);
    p.noise_thresh = ww_prob_threshold((double)cfg->noise_prob);
    p.snr_min = cfg->snr_min_db; p.snr_max = cfg->snr_max_db;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
    p.step_lo = (uint32_t)step; p.step_hi = (uint32_t)(step >> 32);
    p.sample_offset = sample_offset;
    const size_t smem = (size_t)(AUG_TILE + 2 * p.Lp) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    static size_t smem_set = 0;
    if (smem > smem_set) {
        WW_HIP(hipFuncSetAttribute((const void *)k_aug_conv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        smem_set = smem;
    }
    ww_prof_scope ps_(ctx, WW_K_AUDIO_AUG, st);
    dim3 grid(p.nt, B);
    hipLaunchKernelGGL(k_aug_conv, grid, dim3(256), smem, st, wave_in, rirs, noises, p, wave_out, (float *)scratch,
                       choice_out);
    WW_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_aug_mix, grid, dim3(256), 0, st, noises, p, wave_out, (const float *)scratch);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
