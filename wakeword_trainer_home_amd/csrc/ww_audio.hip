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
// k_aug_fftconv : the same convolution for long RIRs (the caller passes spectra from ww_audio_rir_spectra): overlap-save
//              with a 16384-point complex FFT held in LDS (136 KB, one 1024-thread workgroup per CU).  Two consecutive
//              segments of the clip ride in the real and imaginary lanes of ONE transform (h is real, so Re/Im of
//              IFFT(FFT(xa + i xb) H) are xa*h and xb*h); a 1.5 s clip with a 0.25 s RIR is a single block.  The forward
//              transform is decimation-in-frequency (radix 16,16,16,4), the inverse its conjugate transpose, and H is
//              stored in the forward transform's own output order, so no digit reversal is ever materialised; the last
//              radix-4 forward stage, the product with H and the first inverse stage are one register pass.
//              LDS index n lives at n + 4*(n>>6): every stage's ds_read/write_b64 is then bank-conflict free.
#include "ww_internal.h"
#include "ww_fft.h"
#include <vector>

namespace {

constexpr int AUG_TILE = 2048;
constexpr uint32_t TAG_AUDIO = 2u;

struct AugParams {
    int B, N, R, L, Lp, K, Nn, nt, ns;      // nt: 2048-sample tiles of a clip; ns: rows of the stats slab per clip
    uint64_t rir_thresh, noise_thresh;
    float snr_min, snr_max;
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
    uint64_t sample_offset;
    const ww_step_ctl *ctl;
};
struct AugChoice { int rir, noise, offset; float snr_db; };

__device__ __forceinline__ AugChoice aug_choice(const AugParams &p, int b) {
    const uint32_t g = (uint32_t)(p.sample_offset + (uint64_t)b);
    uint32_t r0[4], r1[4], slo, shi;
    ww_step_resolve(p.ctl, p.step_lo, p.step_hi, slo, shi);
    ww_philox(slo, shi, g, (TAG_AUDIO << 24) | 0u, p.seed_lo, p.seed_hi, r0);
    ww_philox(slo, shi, g, (TAG_AUDIO << 24) | 1u, p.seed_lo, p.seed_hi, r1);
    AugChoice c;
    c.rir = (p.R > 0 && (uint64_t)r0[0] < p.rir_thresh) ? (int)(r0[1] % (uint32_t)p.R) : -1;
    c.noise = (p.K > 0 && (uint64_t)r1[0] < p.noise_thresh) ? (int)(r1[1] % (uint32_t)p.K) : -1;
    c.offset = p.K > 0 ? (int)(r1[2] % (uint32_t)(p.Nn - p.N + 1)) : 0;
    const float u = (float)(r1[3] >> 8) * 5.9604644775390625e-08f;   // 2^-24, exact
    c.snr_db = fmaf(u, p.snr_max - p.snr_min, p.snr_min);
    return c;
}

template <int NW>
__device__ __forceinline__ float block_sum(float v, float *sh) {
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += sh[w];
    return s;
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
    sx = block_sum<4>(sx, red);
    sy = block_sum<4>(sy, red);
    sn = block_sum<4>(sn, red);
    if (tid == 0) {
        float *s = stats + ((size_t)b * p.ns + blockIdx.x) * 4;
        s[0] = sx; s[1] = sy; s[2] = sn; s[3] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------ FFT convolution
constexpr int FM = 16384;                         // transform length
constexpr int FM_LDS = FM + (FM >> 6) * 4;        // padded float2 slots
constexpr int FT = 512;                           // threads per block: LDS allows one block per CU, so 8 waves get 256 VGPRs each

__device__ __forceinline__ int fpad(int n) { return n + ((n >> 6) << 2); }
__device__ __forceinline__ float2 cmul2(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// w[k] = w1^k, k = 1..15, by a depth-4 product tree (error ~4 ulp)
__device__ __forceinline__ void tw_powers(const float2 w1, float2 (&w)[16]) {
    w[0] = make_float2(1.f, 0.f); w[1] = w1;
    w[2] = cmul2(w1, w1); w[3] = cmul2(w[2], w1); w[4] = cmul2(w[2], w[2]); w[5] = cmul2(w[4], w1);
    w[6] = cmul2(w[3], w[3]); w[7] = cmul2(w[4], w[3]); w[8] = cmul2(w[4], w[4]);
    w[9] = cmul2(w[8], w1); w[10] = cmul2(w[5], w[5]); w[11] = cmul2(w[8], w[3]); w[12] = cmul2(w[6], w[6]);
    w[13] = cmul2(w[8], w[5]); w[14] = cmul2(w[7], w[7]); w[15] = cmul2(w[8], w[7]);
}

// forward (DIF) stage tail: fft16 output X[k] -> times w1^k -> LDS position base + k*S + m
__device__ __forceinline__ void fwd_store(const float (&re)[16], const float (&im)[16], const float2 w1, float2 *z,
                                          const int pos0, const int S) {
    float2 w[16];
    tw_powers(w1, w);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float2 v = make_float2(re[F16_SLOT(k)], im[F16_SLOT(k)]);
        if (k) v = cmul2(v, w[k]);
        z[fpad(pos0 + k * S)] = v;
    }
}
// inverse stage head: LDS position base + k*S + m -> times conj(w1^k) -> natural-order registers
__device__ __forceinline__ void inv_load(float (&re)[16], float (&im)[16], const float2 w1, const float2 *z,
                                         const int pos0, const int S) {
    float2 w[16];
    tw_powers(make_float2(w1.x, -w1.y), w);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float2 v = z[fpad(pos0 + k * S)];
        if (k) v = cmul2(v, w[k]);
        re[k] = v.x; im[k] = v.y;
    }
}

// Stages 2..3 forward, the fused radix-4/product/radix-4 pass, stages 3..2 inverse.  Stage 1 (stride 1024) is done by the
// caller straight from/to global memory.  SPECTRUM: stop after the forward radix-4 stage and write z*scale to `hout`.
template <bool SPECTRUM>
__device__ __forceinline__ void fft_core(float2 *z, const float2 *__restrict__ tw, const float2 *__restrict__ H,
                                         float2 *__restrict__ hout) {
    float re[16], im[16];
    for (int t = threadIdx.x; t < 1024; t += FT) {   // stage 2: blocks of 1024, stride 64
        const int pos0 = (t >> 6) * 1024 + (t & 63);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float2 v = z[fpad(pos0 + j * 64)]; re[j] = v.x; im[j] = v.y; }
        fft16(re, im);
        fwd_store(re, im, tw[16 * (t & 63)], z, pos0, 64);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += FT) {   // stage 3: blocks of 64, stride 4
        const int pos0 = (t >> 2) * 64 + (t & 3);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float2 v = z[fpad(pos0 + j * 4)]; re[j] = v.x; im[j] = v.y; }
        fft16(re, im);
        fwd_store(re, im, tw[256 * (t & 3)], z, pos0, 4);
    }
    __syncthreads();
#pragma unroll 4
    for (int q = threadIdx.x; q < FM / 4; q += FT) {   // stage 4 (radix 4, stride 1) + product with H + inverse stage 4
        const int n0 = 4 * q;
        float4 *zp = reinterpret_cast<float4 *>(z + fpad(n0));
        float4 a = zp[0], b = zp[1];
        fft4(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
        if (SPECTRUM) {
            const float sc = 1.0f / FM;
            float4 *hp = reinterpret_cast<float4 *>(hout + n0);
            hp[0] = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
            hp[1] = make_float4(b.x * sc, b.y * sc, b.z * sc, b.w * sc);
        } else {
            const float4 *hp = reinterpret_cast<const float4 *>(H + n0);
            const float4 h0 = hp[0], h1 = hp[1];
            float2 y0 = cmul2(make_float2(a.x, a.y), make_float2(h0.x, h0.y));
            float2 y1 = cmul2(make_float2(a.z, a.w), make_float2(h0.z, h0.w));
            float2 y2 = cmul2(make_float2(b.x, b.y), make_float2(h1.x, h1.y));
            float2 y3 = cmul2(make_float2(b.z, b.w), make_float2(h1.z, h1.w));
            fft4(y0.y, y0.x, y1.y, y1.x, y2.y, y2.x, y3.y, y3.x);      // swapped lanes: inverse butterfly
            zp[0] = make_float4(y0.x, y0.y, y1.x, y1.y);
            zp[1] = make_float4(y2.x, y2.y, y3.x, y3.y);
        }
    }
    if (SPECTRUM) return;
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += FT) {   // inverse stage 3
        const int pos0 = (t >> 2) * 64 + (t & 3);
        inv_load(re, im, tw[256 * (t & 3)], z, pos0, 4);
        fft16(im, re);
#pragma unroll
        for (int j = 0; j < 16; ++j) z[fpad(pos0 + j * 4)] = make_float2(re[F16_SLOT(j)], im[F16_SLOT(j)]);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += FT) {   // inverse stage 2
        const int pos0 = (t >> 6) * 1024 + (t & 63);
        inv_load(re, im, tw[16 * (t & 63)], z, pos0, 64);
        fft16(im, re);
#pragma unroll
        for (int j = 0; j < 16; ++j) z[fpad(pos0 + j * 64)] = make_float2(re[F16_SLOT(j)], im[F16_SLOT(j)]);
    }
    __syncthreads();
}

// RIR bank -> spectra in the forward transform's output order, scaled by 1/FM.  grid R, block FT.
__global__ __launch_bounds__(FT) void k_aug_rir_spectrum(const float *__restrict__ rirs, int L,
                                                           const float2 *__restrict__ tw, float2 *__restrict__ spectra) {
    extern __shared__ __align__(16) float aug_lds[];
    float2 *z = reinterpret_cast<float2 *>(aug_lds);
    const float *h = rirs + (size_t)blockIdx.x * L;
    for (int t = threadIdx.x; t < 1024; t += FT) {
        float re[16], im[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int i = j * 1024 + t; re[j] = i < L ? h[i] : 0.f; im[j] = 0.f; }
        fft16(re, im);
        fwd_store(re, im, tw[t], z, t, 1024);
    }
    __syncthreads();
    fft_core<true>(z, tw, nullptr, spectra + (size_t)blockIdx.x * FM);
}

// grid (segment pairs, clips), block FT.  Segment s of a clip yields y[s*V, (s+1)*V), V = FM - L + 1.
__global__ __launch_bounds__(FT) void k_aug_fftconv(const float *__restrict__ x, const float2 *__restrict__ spectra,
                                                      const float *__restrict__ noises, const float2 *__restrict__ tw,
                                                      AugParams p, float *__restrict__ out, float *__restrict__ stats,
                                                      int32_t *__restrict__ choice_out) {
    extern __shared__ __align__(16) float aug_lds[];
    float2 *z = reinterpret_cast<float2 *>(aug_lds);
    __shared__ float red[16];
    const int b = blockIdx.y;
    const AugChoice c = aug_choice(p, b);
    if (choice_out && blockIdx.x == 0 && threadIdx.x == 0) {
        choice_out[4 * b] = c.rir; choice_out[4 * b + 1] = c.noise; choice_out[4 * b + 2] = c.offset;
        choice_out[4 * b + 3] = __float_as_int(c.snr_db);
    }
    const int V = FM - p.L + 1;
    const int ta0 = 2 * blockIdx.x * V;          // first output sample of segment a; segment b starts V later
    const float *xb = x + (size_t)b * p.N;
    float *ob = out + (size_t)b * p.N;
    const float *nb = c.noise >= 0 ? noises + (size_t)c.noise * p.Nn + c.offset : nullptr;
    float sx = 0.f, sy = 0.f, sn = 0.f;
    if (c.rir < 0) {                             // block-uniform: plain copy of this block's output range
        const int hi = min(p.N, ta0 + 2 * V);
        for (int s = ta0 + threadIdx.x; s < hi; s += FT) {
            const float v = xb[s];
            ob[s] = v;
            sx = fmaf(v, v, sx);
            if (nb) { const float nv = nb[s]; sn = fmaf(nv, nv, sn); }
        }
        sy = sx;
    } else {
        float re[16], im[16];
        for (int t = threadIdx.x; t < 1024; t += FT) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {       // stage 1 straight from global: block position i = j*1024 + t
                const int i = j * 1024 + t;
                const int sa = ta0 - (p.L - 1) + i, sb = sa + V;
                re[j] = (sa >= 0 && sa < p.N) ? xb[sa] : 0.f;
                im[j] = (sb >= 0 && sb < p.N) ? xb[sb] : 0.f;
                if (i >= p.L - 1) sx += re[j] * re[j] + im[j] * im[j];
            }
            fft16(re, im);
            fwd_store(re, im, tw[t], z, t, 1024);
        }
        __syncthreads();
        fft_core<false>(z, tw, spectra + (size_t)c.rir * FM, nullptr);
        for (int t = threadIdx.x; t < 1024; t += FT) {
        inv_load(re, im, tw[t], z, t, 1024);     // inverse stage 1 straight to global
        fft16(im, re);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = j * 1024 + t;
            if (i < p.L - 1) continue;           // circularly aliased head of the block
            const int sa = ta0 + i - (p.L - 1), sb = sa + V;
            const float ya = re[F16_SLOT(j)], yb = im[F16_SLOT(j)];
            if (sa < p.N) {
                ob[sa] = ya; sy = fmaf(ya, ya, sy);
                if (nb) { const float nv = nb[sa]; sn = fmaf(nv, nv, sn); }
            }
            if (sb < p.N) {
                ob[sb] = yb; sy = fmaf(yb, yb, sy);
                if (nb) { const float nv = nb[sb]; sn = fmaf(nv, nv, sn); }
            }
        }
        }
    }
    sx = block_sum<FT / 64>(sx, red);
    sy = block_sum<FT / 64>(sy, red);
    sn = block_sum<FT / 64>(sn, red);
    if (threadIdx.x == 0) {
        float *s = stats + ((size_t)b * p.ns + blockIdx.x) * 4;
        s[0] = sx; s[1] = sy; s[2] = sn; s[3] = 0.f;
    }
}

__global__ __launch_bounds__(256) void k_aug_mix(const float *__restrict__ noises, AugParams p, float *__restrict__ out,
                                                 const float *__restrict__ stats) {
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * AUG_TILE;
    const AugChoice c = aug_choice(p, b);
    double sx = 0.0, sy = 0.0, sn = 0.0;
    for (int i = 0; i < p.ns; ++i) {          // fixed order -> every tile of the clip derives identical gains
        const float *s = stats + ((size_t)b * p.ns + i) * 4;
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

static int get_tw16k(ww_ctx *ctx, const float2 **out) {
    if (!ctx->tw16k) {
        std::vector<float2> h(1024);
        for (int m = 0; m < 1024; ++m) {
            const double a = -2.0 * M_PI * m / FM;
            h[m] = make_float2((float)cos(a), (float)sin(a));
        }
        WW_HIP(hipMalloc((void **)&ctx->tw16k, h.size() * sizeof(float2)));
        WW_HIP(hipMemcpy(ctx->tw16k, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    *out = ctx->tw16k;
    return WW_OK;
}
static int fft_smem_attr() {
    static bool done = false;
    if (!done) {
        const int smem = FM_LDS * (int)sizeof(float2);
        WW_HIP(hipFuncSetAttribute((const void *)k_aug_fftconv, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        WW_HIP(hipFuncSetAttribute((const void *)k_aug_rir_spectrum, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        done = true;
    }
    return WW_OK;
}

extern "C" size_t ww_audio_rir_spectra_bytes(int R) { return R < 1 ? 0 : (size_t)R * FM * sizeof(float2); }

extern "C" int ww_audio_rir_spectra(ww_ctx *ctx, const float *rirs, int R, int L, void *spectra, size_t spectra_bytes,
                                    ww_stream_t stream) {
    WW_REQUIRE(ctx && rirs && spectra, WW_E_INVALID, "ww_audio_rir_spectra: null argument");
    WW_REQUIRE(R >= 1 && L >= 1 && L <= 8192, WW_E_UNSUPPORTED, "ww_audio_rir_spectra: bank (%d,%d) not in [1..]x[1,8192]", R, L);
    WW_REQUIRE(spectra_bytes >= ww_audio_rir_spectra_bytes(R), WW_E_WORKSPACE, "ww_audio_rir_spectra: output too small");
    const float2 *tw;
    int rc;
    if ((rc = get_tw16k(ctx, &tw)) || (rc = fft_smem_attr())) return rc;
    hipLaunchKernelGGL(k_aug_rir_spectrum, dim3(R), dim3(FT), FM_LDS * sizeof(float2), (hipStream_t)stream, rirs, L, tw,
                       (float2 *)spectra);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" size_t ww_audio_augment_scratch_bytes(int B, int N) {
    if (B < 1 || N < 1) return 0;
    return (size_t)B * ((N + AUG_TILE - 1) / AUG_TILE) * 4 * sizeof(float);   // >= the FFT path's segment-pair rows
}

extern "C" int ww_audio_augment(ww_ctx *ctx, const float *wave_in, float *wave_out, int B, int N, const float *rirs,
                                int R, int L, const void *rir_spectra, const float *noises, int K, int Nn,
                                const ww_audio_aug_cfg *cfg,
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
    p.ns = p.nt;
    p.rir_thresh = ww_prob_threshold((double)cfg->rir_prob);
    p.noise_thresh = ww_prob_threshold((double)cfg->noise_prob);
    p.snr_min = cfg->snr_min_db; p.snr_max = cfg->snr_max_db;
    p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
    p.step_lo = (uint32_t)step; p.step_hi = (uint32_t)(step >> 32);
    p.ctl = ctx->step_ctl;
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
    if (rir_spectra && R > 0) {                   // long RIRs: overlap-save FFT convolution, two segments per block
        const int V = FM - L + 1;
        p.ns = ((N + V - 1) / V + 1) / 2;
        const float2 *tw;
        int rc;
        if ((rc = get_tw16k(ctx, &tw)) || (rc = fft_smem_attr())) return rc;
        hipLaunchKernelGGL(k_aug_fftconv, dim3(p.ns, B), dim3(FT), FM_LDS * sizeof(float2), st, wave_in,
                           (const float2 *)rir_spectra, noises, tw, p, wave_out, (float *)scratch, choice_out);
    } else {
        hipLaunchKernelGGL(k_aug_conv, grid, dim3(256), smem, st, wave_in, rirs, noises, p, wave_out, (float *)scratch,
                           choice_out);
    }
    WW_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_aug_mix, grid, dim3(256), 0, st, noises, p, wave_out, (const float *)scratch);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
