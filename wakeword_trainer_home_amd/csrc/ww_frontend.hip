// Batched STFT -> power -> HTK mel -> log (-> DCT-II) with SpecAugment fused into the
// write-out.  One 256-thread workgroup = one clip x 16 consecutive frames.
//
// Data movement (per workgroup): the 16 frames' sample span (15*hop + 1024 samples, reflect
// padded at the clip edges) is read ONCE from HBM with coalesced loads into LDS; every later
// access (6.4x frame overlap at hop 160) is served from LDS.  Each wavefront transforms TWO
// real frames as one 1024-point complex FFT held 16 points/lane, decomposed 16 x 16 x 4:
//   pass 1  radix-16 in registers over n2 (n = lane + 64*n2), twiddle W1024^(lane*kb) -> LDS (re plane, then im plane)
//   pass 2  radix-16 in registers over m  (lane = kb*4+q, n1 = 4m+q), twiddle W64^(q*kc) -> LDS (in place: each lane
//           rewrites exactly the 16 slots it read, so one 4.4 KB tile per wavefront suffices)
//   pass 3  radix-4 butterflies X[16kc + 256kd + kb], run in PAIRS that produce X[k] and X[N-k] in the same lane: the two
//           real spectra are separated in registers and |.|^2 goes straight to the power buffers (no return to LDS)
// The mel band sums then run on the matrix pipe (v_mfma_f32_4x4x1: 16 blocks of 4 bands x the pair's frames, one bin per
// step), and the 16 x F tile is transposed through LDS so the (B,1,F,T) output is written in 64-byte runs along T.
// Index algebra verified against numpy (tests/test_frontend_index_algebra.py).
#include "ww_internal.h"

namespace {

// (frames per item: 4 per wavefront of the workgroup, see k_logmel)
constexpr int PB_LD = 552;                   // floats per power spectrum: bin j sits at j + (j >> 4) (bank spreading); the band sums
                                             // read whole groups of 8 bins, the last one [512, 520) -> slots 544 ... 551

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
    const float *dct_t;          // (M, F): k_logmel
    int span_len;
    int n_mel_w;
    const int32_t *melq_tab;     // k_logmel: band sums in v_mfma_f32_4x4x1 form (ww_feat_tables)
    const float *melq_w;
    int n_melq_w;
};

// order a wave's own LDS traffic (cross-lane hand-off inside one wavefront): LDS processes one wave's operations in
// issue order, so only the compiler has to be kept from reordering.  A wavefront touches only its own FFT / power
// buffers inside a round, so no workgroup barrier is needed there (PMC: 47 % of wave-cycles were parked).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v[F16_SLOT(k)] *= W^(base*k), k = 0..15, then store to dst(k).  The 15 factors come from SIX table entries (t1[b] =
// W^(base*b), t4[a] = W^(4*base*a), a, b = 1..3) held in registers for the whole kernel (base depends on the lane only)
// and one complex product each for the nine mixed ones.
struct Tw6 { float2 t1[4], t4[4]; };
__device__ __forceinline__ Tw6 load_tw6(const float2 *__restrict__ tw, int base) {
    Tw6 t;
    t.t1[0] = t.t4[0] = make_float2(1.f, 0.f);
#pragma unroll
    for (int b = 1; b < 4; ++b) {
        t.t1[b] = tw[(base * b) & 1023];
        t.t4[b] = tw[(base * 4 * b) & 1023];
    }
    return t;
}
// Built for a SMALL per-CU footprint: in training the front end runs on a side stream beside the conv stack, and what it
// costs the step there is the registers and LDS its resident workgroups hold (the r01 form of this kernel -- complex
// exchange tiles, window and twiddles in registers: 242 VGPRs, 73 KB -- left a SIMD room for ONE 168-VGPR
// depthwise-backward wave instead of three).  The FFT passes exchange re and im one after the other through ONE 4.4 KB tile
// per wavefront (the power rows reuse it), the window lives in LDS and the twiddle factors of a pass are loaded from the
// L1-resident table right before it: 119 VGPRs, 41 KB LDS per workgroup.  The grid is PERSISTENT (a workgroup walks items
// = (clip, block of FR frames) in steps of gridDim.x; window / mel tables staged once), and its size is the caller's
// choice (ww_ctx_set_logmel_workgroups): the whole device when the front end runs alone, one workgroup per CU beside a
// training step.
constexpr int PROW = 68;                     // floats per kb row of the exchange tile (64 + 4 pad: conflict-free both ways)
constexpr int XB = 2 * PB_LD;                // floats per wavefront tile (>= 16 * PROW = 1088)
static_assert(XB >= 16 * PROW, "tile too small for the exchange planes");

__device__ __forceinline__ void twiddle16(float (&re)[16], float (&im)[16], const Tw6 &t) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const int a = k >> 2, b = k & 3;
        float wr, wi;
        if (a == 0) { wr = t.t1[b].x; wi = t.t1[b].y; }
        else if (b == 0) { wr = t.t4[a].x; wi = t.t4[a].y; }
        else { wr = t.t4[a].x; wi = t.t4[a].y; cmul_c(wr, wi, t.t1[b].x, t.t1[b].y); }
        cmul_c(re[F16_SLOT(k)], im[F16_SLOT(k)], wr, wi);
    }
}

// WAVES = wavefronts per workgroup (4 or 8); an item is FRW = 4 * WAVES frames (two rounds of two frames per wave).  The 8-wave
// form shares the window / mel tables / span of 32 frames among twice the waves: 2 workgroups = 16 waves per CU (4 per SIMD)
// where the 4-wave form's LDS allows 3 workgroups = 12 waves, and a 32-frame span re-reads 864 of 5984 samples instead of 864 of
// 3424 (HBM-side traffic x1.2 instead of x1.38).
#ifdef WW_LOGMEL_STAMPS
// phase stamps (s_memtime) of one wave of one workgroup while it processes its third item: tools/logmel_stamps.py
__device__ unsigned long long ww_logmel_stamps[32];
#define WW_STAMP(i) do { if (stamp_on) ww_logmel_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WW_STAMP(i) do { } while (0)
#endif
template <typename WaveT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 8 ? 4 : 2, 4)))
void k_logmel(const WaveT *__restrict__ wave, FeatArgs a, float *__restrict__ out, int use_mask, ww_mask_params mp,
                     int32_t *__restrict__ mask_idx, int nblk, long nitems) {
    constexpr int NT = 64 * WAVES, FRW = 4 * WAVES;
    extern __shared__ __align__(16) unsigned char smem[];
    float *xball = reinterpret_cast<float *>(smem);                   // WAVES x XB
    float *wl = xball + WAVES * XB;                                       // the 1024 window samples
    float *span = wl + WW_NFFT;                                       // span_len (16-byte aligned: vector staging)
    float *lm = span + ((a.span_len + 3) & ~3);                       // FR x M (raw mel sums, then their logs)
    float *feat = xball;                                              // FR x F cepstra: in the tiles, spent once the rounds are over (== lm when !use_dct)
    int *msk = reinterpret_cast<int *>(lm + FRW * a.M);               // 2*WW_MAX_MASKS
    int *mtab = msk + 2 * WW_MAX_MASKS;                               // WW_MELQ_TAB ints (ww_get_feat_tables)
    float *mw = reinterpret_cast<float *>(mtab + WW_MELQ_TAB);        // n_melq_w band weights, one per (pass, step, lane)
    if (!a.use_dct) feat = lm;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- once per workgroup (persistent grid): window and mel tables
    for (int i = tid; i < WW_NFFT / 4; i += NT) reinterpret_cast<float4 *>(wl)[i] = reinterpret_cast<const float4 *>(a.window)[i];
    // (the tiles' pad slots are never written -- the band sums multiply them by zero weights, so they must not hold NaN patterns)
    for (int i = tid; i < WAVES * XB; i += NT) xball[i] = 0.f;
    for (int i = tid; i < WW_MELQ_TAB; i += NT) mtab[i] = a.melq_tab[i];
    for (int i = tid; i < a.n_melq_w; i += NT) mw[i] = a.melq_w[i];
    const int K = use_mask ? (mp.n_f + mp.n_t) : 0;

    float *xb = xball + wv * XB;
    float *const p1 = xb + lane;                               // pass-1 layout: slot [kb][lane]       -> p1[kb * PROW]
    float *const p2 = xb + (lane >> 2) * PROW + (lane & 3);    // pass-2 layout: slot [kb2][4 m + q]   -> p2[4 m]
    // pass 3: butterfly (kb, kc) turns slots [kb][4kc + q] into X[16kc + 256kd + kb], kd = 0..3; X[1024 - k] comes out of
    // butterfly (16 - kb, 15 - kc) (kb = 0: (0, 16 - kc)) at kd' = 3 - kd, so a lane that runs both has each (X[k], X[N-k])
    // pair in registers.  128 units = 127 butterfly pairs + one unit holding the self-paired butterflies (0,0) and (0,8).
    int offA[2], offB[2], kbase[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = lane + 64 * u;
        int kbA, kcA, kbB, kcB;
        if (idx < 112) { kbA = 1 + (idx >> 4); kcA = idx & 15; kbB = 16 - kbA; kcB = 15 - kcA; }
        else if (idx < 120) { kbA = 8; kcA = idx - 112; kbB = 8; kcB = 15 - kcA; }
        else if (idx < 127) { kbA = 0; kcA = idx - 119; kbB = 0; kcB = 16 - kcA; }
        else { kbA = 0; kcA = 0; kbB = 0; kcB = 8; }
        offA[u] = kbA * PROW + 4 * kcA;
        offB[u] = kbB * PROW + 4 * kcB;
        kbase[u] = 16 * kcA + kbA;
    }
    const bool special = lane == 63;                           // unit 127 (u = 1)
    // The NEXT item's span is fetched into registers while the current item's tail (log pass, write-out) runs, and stored to
    // LDS at the top of the next trip: the ~1.2 us HBM round trip of the staging was exposed once per item (10 % of an item by
    // the s_memtime stamps, tools/logmel_stamps.py).  Interior, 16-byte-aligned fp32 spans of <= 4 float4 per thread only.
    __syncthreads();                                           // the tables are in LDS
    const int npass = __builtin_amdgcn_readfirstlane(mtab[0]), nunit = 16 * npass;
    float4 pre[4];
    bool pre_ok = false;
    auto fast_span = [&](int bb, long base_) {
        return sizeof(WaveT) == 4 && base_ >= 0 && base_ + a.span_len <= a.N && (((size_t)bb * a.N + (size_t)base_) & 3) == 0 &&
               (reinterpret_cast<uintptr_t>(wave) & 15) == 0;
    };

    for (long item = blockIdx.x; item < nitems; item += gridDim.x) {
#ifdef WW_LOGMEL_STAMPS
        const bool stamp_on = blockIdx.x == 5 && tid == 0 && item == 5 + 2L * gridDim.x;      // (an interior block at 256 and 768 workgroups)
#endif
        WW_STAMP(0);
        const int b = (int)(item / nblk), blk = (int)(item - (long)b * nblk), t0 = blk * FRW;
        const WaveT *x = wave + (size_t)b * a.N;
        {
            const long base = (long)t0 * a.hop - WW_NFFT / 2;
            bool fast = false;
            if constexpr (sizeof(WaveT) == 4) {
                // interior span on a 16-byte boundary (hop 160, N 24000: every item but the first and last of a clip)
                fast = fast_span(b, base);
                if (fast && pre_ok) {                                  // fetched during the previous item's tail
                    const int n4 = a.span_len >> 2;
#pragma unroll
                    for (int u = 0; u < 4; ++u) reinterpret_cast<float4 *>(span)[min(tid + NT * u, n4 - 1)] = pre[u];
                    for (int i = 4 * n4 + tid; i < a.span_len; i += NT) span[i] = x[base + i];
                } else if (fast) {
                    // a thread's float4s of the span in batches of four unconditional (clamped) loads: as a plain copy loop
                    // every float4 was its own load -> wait -> LDS store round trip (3-4 dependent HBM latencies per item,
                    // and at one workgroup per CU beside a training step nothing else hides them)
                    const float4 *src = reinterpret_cast<const float4 *>(x + base);
                    const int n4 = a.span_len >> 2;
                    for (int i0 = tid; i0 < n4; i0 += 4 * NT) {
                        float4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + NT * u, n4 - 1)];
                        // all four must be live here (else the register allocator, at this kernel's pressure, funnels them through
                        // ONE register quad: load, wait, store, load ...); the stores are unconditional too (a clamped slot is
                        // rewritten with its own value): behind a guard the compiler sinks each load next to its store again
                        asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y),
                                     "+v"(v[1].z), "+v"(v[1].w), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w),
                                     "+v"(v[3].x), "+v"(v[3].y), "+v"(v[3].z), "+v"(v[3].w));
#pragma unroll
                        for (int u = 0; u < 4; ++u) reinterpret_cast<float4 *>(span)[min(i0 + NT * u, n4 - 1)] = v[u];
                    }
                    for (int i = 4 * n4 + tid; i < a.span_len; i += NT) span[i] = x[base + i];
                }
            }
            if (!fast) {
                for (int i0 = tid; i0 < a.span_len; i0 += 8 * NT) {      // same batching for the reflected / int16 / unaligned spans
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        long idx = base + min(i0 + NT * u, a.span_len - 1);
                        if (idx < 0) idx = -idx;
                        if (idx >= a.N) idx = 2L * (a.N - 1) - idx;
                        idx = idx < 0 ? 0 : (idx >= a.N ? a.N - 1 : idx);  // only frames >= T can get here
                        v[u] = load_sample(x, (size_t)idx);
                    }
                    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
                    for (int u = 0; u < 8; ++u) span[min(i0 + NT * u, a.span_len - 1)] = v[u];
                }
            }
        }
        __syncthreads();
        WW_STAMP(1);

        for (int round = 0; round < 2; ++round) {
            const int fa = round * 2 * WAVES + 2 * wv, fb = fa + 1;  // local frame indices of this wave's pair
            // a clip's last block holds frames past T (T = 151: 7 of its 16 are real): a pair with no real frame is skipped
            // (wave-uniform; nothing below reads what such a pair would have written -- the write-out guards t < T)
            if (t0 + fa >= a.T) continue;
            float re[16], im[16];
            // ---- pass 1: windowed load, radix-16 over n2, twiddle W1024^(lane*kb)
            int tb1 = lane, tb2 = 16 * (lane & 3);
            asm volatile("" : "+v"(tb1), "+v"(tb2));             // keep the table loads inside the round
            {
                const float *sa = span + fa * a.hop + lane, *sb = span + fb * a.hop + lane, *wp = wl + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float w = wp[64 * j];
                    re[j] = w * sa[64 * j];
                    im[j] = w * sb[64 * j];
                }
            }
            WW_STAMP(2 + 8 * round);
            fft16(re, im);
            WW_STAMP(3 + 8 * round);
            {
                const Tw6 tw1 = load_tw6(a.twiddle, tb1);
                twiddle16(re, im, tw1);
            }
            WW_STAMP(4 + 8 * round);
            // exchange 1 (planar): real parts through the tile, then the imaginary parts
#pragma unroll
            for (int k = 0; k < 16; ++k) p1[k * PROW] = re[F16_SLOT(k)];
            wave_sync();
#pragma unroll
            for (int m = 0; m < 16; ++m) re[m] = p2[4 * m];
            wave_sync();
#pragma unroll
            for (int k = 0; k < 16; ++k) p1[k * PROW] = im[F16_SLOT(k)];
            wave_sync();
#pragma unroll
            for (int m = 0; m < 16; ++m) im[m] = p2[4 * m];
            wave_sync();
            // ---- pass 2: lane = kb*4 + q ; radix-16 over m (n1 = 4m + q), twiddle W64^(q*kc); written back in place
            WW_STAMP(5 + 8 * round);
            fft16(re, im);
            {
                const Tw6 tw2 = load_tw6(a.twiddle, tb2);        // (fetching both passes' factors at the top of the round: 128
                twiddle16(re, im, tw2);                          //  VGPRs, 149 us alone either way, 1.260 vs 1.238 ms in the step;
                                                                 //  all 15 factors of a pass straight from the table instead of 6 +
                                                                 //  nine complex products: 133 us alone against 116.  The two
                                                                 //  radix-16 passes on (re, im) register pairs with v_pk_add/mul/
                                                                 //  fma_f32 -- 270 fewer VALU instructions per round of ~1300, but
                                                                 //  pair alignment and dependent mul -> fma pairs: 121 us alone
                                                                 //  against 116, the step equal: profiles/EXPERIMENTS.md)
            }
            // exchange 2 (planar) straight into the pass-3 butterflies' registers
            WW_STAMP(6 + 8 * round);
            float4 ar[2], ai[2], br[2], bi[2];
#pragma unroll
            for (int k = 0; k < 16; ++k) p2[4 * k] = re[F16_SLOT(k)];
            wave_sync();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ar[u] = *reinterpret_cast<const float4 *>(xb + offA[u]);
                br[u] = *reinterpret_cast<const float4 *>(xb + offB[u]);
            }
            wave_sync();
#pragma unroll
            for (int k = 0; k < 16; ++k) p2[4 * k] = im[F16_SLOT(k)];
            wave_sync();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ai[u] = *reinterpret_cast<const float4 *>(xb + offA[u]);
                bi[u] = *reinterpret_cast<const float4 *>(xb + offB[u]);
            }
            wave_sync();
            // ---- pass 3 fused with the separation of the two real spectra; |.|^2 of both frames into the tile's power rows
            WW_STAMP(7 + 8 * round);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                fft4(ar[u].x, ai[u].x, ar[u].y, ai[u].y, ar[u].z, ai[u].z, ar[u].w, ai[u].w);
                fft4(br[u].x, bi[u].x, br[u].y, bi[u].y, br[u].z, bi[u].z, br[u].w, bi[u].w);
                auto emit = [&](float xr_, float xi_, float yr_, float yi_, int k) {   // X = X[k], Y = X[N-k]
                    const int j = k <= 512 ? k : 1024 - k;
                    const int pj = j + (j >> 4);
                    const float pr = xr_ + yr_, pi = xi_ - yi_;     // 2 * spectrum of frame a
                    const float qr = xi_ + yi_, qi = xr_ - yr_;     // 2 * spectrum of frame b (up to the sign of its imaginary part)
                    xb[pj] = pr * pr + pi * pi;                     // 4 |X_a|^2: the 1/4 sits in the band weights (ww_ctx.hip)
                    xb[PB_LD + pj] = qr * qr + qi * qi;
                };
                if (!(u == 1 && special)) {
                    emit(ar[u].x, ai[u].x, br[u].w, bi[u].w, kbase[u]);
                    emit(ar[u].y, ai[u].y, br[u].z, bi[u].z, kbase[u] + 256);
                    emit(ar[u].z, ai[u].z, br[u].y, bi[u].y, kbase[u] + 512);
                    emit(ar[u].w, ai[u].w, br[u].x, bi[u].x, kbase[u] + 768);
                } else {
                    emit(ar[u].x, ai[u].x, ar[u].x, ai[u].x, 0);
                    emit(ar[u].y, ai[u].y, ar[u].w, ai[u].w, 256);
                    emit(ar[u].z, ai[u].z, ar[u].z, ai[u].z, 512);
                    emit(br[u].x, bi[u].x, br[u].w, bi[u].w, 128);
                    emit(br[u].y, bi[u].y, br[u].z, bi[u].z, 384);
                }
            }
            wave_sync();
            WW_STAMP(8 + 8 * round);
            // ---- mel band sums on the matrix pipe: v_mfma_f32_4x4x1 = 16 independent 4x4 outer products, block = (4 consecutive
            // bands) x (this pair's frames: columns 2, 3 repeat 0, 1), one spectrum bin per step -- per step one weight (LDS, a
            // constant offset) and one power value (LDS, consecutive slots inside a group of 8 bins) per lane.  As VALU work
            // (two lanes per band, runs of 16 slots) this phase was 27 trips x (8 x 16-byte LDS loads + 16 FMAs) per round, a
            // third of the kernel (tools/logmel_stamps.py); operand layout checked on the device (tools/probes/mfma4x4x1.hip).
#ifndef WW_LOGMEL_NOMEL
            typedef float f4_ __attribute__((ext_vector_type(4)));
            f4_ acc[WW_MELQ_MAX_PASSES];
            int unit_[WW_MELQ_MAX_PASSES];
#pragma unroll
            for (int p = 0; p < WW_MELQ_MAX_PASSES; ++p) {
                acc[p] = f4_{0.f, 0.f, 0.f, 0.f};
                unit_[p] = 0;
                if (p < npass) {
                    const int steps = __builtin_amdgcn_readfirstlane(mtab[2 + 2 * p]);
                    const float *wq = mw + __builtin_amdgcn_readfirstlane(mtab[3 + 2 * p]) + lane;
                    const int2 blk_ = *reinterpret_cast<const int2 *>(mtab + 10 + 32 * p + 2 * (lane >> 2));   // {first bin, unit}
                    unit_[p] = blk_.y;
                    const float *row = xb + (lane & 1) * PB_LD;
                    // operands of a group of 8 steps; the next group's are fetched before this group's products are issued
                    // (a short unit's idle steps: zero weights, any finite power value)
                    float wa[8], pb[8];
                    auto fetch = [&](int t, float (&w_)[8], float (&p_)[8]) {
                        const int j = min(blk_.x + t, 512);
                        const float *pp = row + j + (j >> 4);
#pragma unroll
                        for (int i = 0; i < 8; ++i) { w_[i] = wq[64 * (t + i)]; p_[i] = pp[i]; }
                    };
                    fetch(0, wa, pb);
                    for (int t = 0; t < steps; t += 16) {                // (two groups per trip: the buffers swap roles, no copies)
                        float wn[8], pn[8];
                        fetch(min(t + 8, steps - 8), wn, pn);
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[i], pb[i], acc[p], 0, 0, 0);
                        fetch(min(t + 16, steps - 8), wa, pb);
                        if (t + 8 < steps) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[p] = __builtin_amdgcn_mfma_f32_4x4x1f32(wn[i], pn[i], acc[p], 0, 0, 0);
                        }
                    }
                }
            }
            // the power rows are spent: the unit partials go through the head of the tile, [frame][unit][4 bands]; a band's sum =
            // its quad's partials in bin order (at most 8 that are not empty, ww_get_feat_tables), fetched together
            wave_sync();
#pragma unroll
            for (int p = 0; p < WW_MELQ_MAX_PASSES; ++p)
                if (p < npass && (lane & 3) < 2)
                    *reinterpret_cast<float4 *>(xb + ((lane & 3) * nunit + unit_[p]) * 4) = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
            wave_sync();
            for (int it = lane; it < 2 * a.M; it += 64) {
                const int fr = it >= a.M ? 1 : 0, m = it - fr * a.M;
                const int2 uu = make_int2(mtab[138 + (m >> 2)], mtab[139 + (m >> 2)]);
                const float *src = xb + (fr * nunit + uu.x) * 4 + (m & 3);
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = src[uu.x + k < uu.y ? 4 * k : 0];
                float sum = v[0];                                      // (every quad has at least one unit)
#pragma unroll
                for (int k = 1; k < 8; ++k) sum += uu.x + k < uu.y ? v[k] : 0.f;
                lm[(fa + fr) * a.M + m] = sum;
            }
#endif
            wave_sync();
            WW_STAMP(9 + 8 * round);
        }
        __syncthreads();
        WW_STAMP(18);
        // the clip's SpecAugment masks (Philox, a handful of lanes): only the write-out needs them, so they are drawn here, beside
        // the log pass, instead of in front of the staging barrier every wave waits at
        if (tid < K) {
            int s, w;
            ww_specaug_mask(mp, (uint32_t)(mp.sample_offset + (uint64_t)b), tid, a.F, a.T, s, w);
            msk[2 * tid] = s;
            msk[2 * tid + 1] = w;
            if (mask_idx && blk == 0) {
                mask_idx[((size_t)b * K + tid) * 2] = s;
                mask_idx[((size_t)b * K + tid) * 2 + 1] = w;
            }
        }
        pre_ok = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);      // (ends the old values' live range before the rounds)
        if constexpr (sizeof(WaveT) == 4) {
            const long nitem = item + gridDim.x;
            const int n4 = a.span_len >> 2;
            if (nitem < nitems && n4 <= 4 * NT) {
                const int nb = (int)(nitem / nblk);
                const long nbase = (long)((int)(nitem - (long)nb * nblk) * FRW) * a.hop - WW_NFFT / 2;
                if (fast_span(nb, nbase)) {
                    const float4 *src = reinterpret_cast<const float4 *>(wave + (size_t)nb * a.N + nbase);
#pragma unroll
                    for (int u = 0; u < 4; ++u) pre[u] = src[min(tid + NT * u, n4 - 1)];      // (no register pin here: an asm
                    pre_ok = true;                                                              //  operand would wait for the loads)
                }
            }
        }
        for (int it = tid; it < FRW * a.M; it += NT) lm[it] = logf(lm[it] + a.log_eps);
        __syncthreads();
        WW_STAMP(19);

        if (a.use_dct) {
            // (FRW * F <= WAVES * XB: F <= M <= 128.)  Neighbouring lanes = neighbouring coefficients of one frame: the
            // transposed table makes their loads one line, eight in flight per lane; with (F, M) rows each lane walked its own
            // row, a dependent L1 round trip per term -- MFCC-13 193 us against the log-mel's 110
            for (int it = tid; it < FRW * a.F; it += NT) {
                const int fr = it / a.F, c = it - fr * a.F;
                const float *d = a.dct_t + c;
                const float *l = lm + fr * a.M;
                float acc = 0.f;
                int m = 0;
                for (; m + 8 <= a.M; m += 8) {
                    float dv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) dv[i] = d[(size_t)(m + i) * a.F];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc = fmaf(dv[i], l[m + i], acc);
                }
                for (; m < a.M; ++m) acc = fmaf(d[(size_t)m * a.F], l[m], acc);
                feat[it] = acc;
            }
            __syncthreads();
        }

        // ---- masked, transposed write-out: out[b][0][f][t0 + i].  A thread's frame index i is the same for all of its elements
        // (FRW divides the workgroup size), so the masks are folded ONCE per item into a time flag and a 128-bit set of masked
        // feature rows (2K LDS reads, one round trip); per element it is then one LDS read and a bit test (it was 2K LDS
        // reads per element: 11 % of an item by the s_memtime stamps)
        {
            const int i = tid % FRW, t = t0 + i;
            bool tmask = false;
            unsigned long long fm0 = 0ull, fm1 = 0ull;
            for (int k = 0; k < K; ++k) {
                const int s_ = msk[2 * k], w_ = msk[2 * k + 1];
                if (k < mp.n_f) {
                    const int lo0 = max(s_, 0), hi0 = min(s_ + w_, 64), lo1 = max(s_, 64) - 64, hi1 = min(s_ + w_, 128) - 64;
                    if (hi0 > lo0) fm0 |= (hi0 - lo0 >= 64 ? ~0ull : ((1ull << (hi0 - lo0)) - 1ull)) << lo0;
                    if (hi1 > lo1) fm1 |= (hi1 - lo1 >= 64 ? ~0ull : ((1ull << (hi1 - lo1)) - 1ull)) << lo1;
                } else {
                    tmask = tmask || (t >= s_ && t < s_ + w_);
                }
            }
            if (t < a.T)
                for (int f = tid / FRW; f < a.F; f += NT / FRW) {
                    const bool fmask = ((f < 64 ? fm0 >> f : fm1 >> (f - 64)) & 1ull) != 0ull;
                    const float v = (tmask || fmask) ? 0.f : feat[i * a.F + f];
                    out[((size_t)b * a.F + f) * a.T + t] = v;
                }
        }
        WW_STAMP(20);
        __syncthreads();   // the next item restages span / masks / lm
        WW_STAMP(21);
    }
}

#ifdef WW_LOGMEL_STAMPS
}  // namespace
extern "C" int ww_debug_logmel_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ww_logmel_stamps), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
namespace {
#endif


// ---- n_fft above 1024 (the reference's validator accepts 256 ... 4096, src/config/validator.py:129; 1024 is its default and
// the size k_logmel is built for; shorter transforms run there too).  Plain and general rather than fast: one workgroup per (clip, frame), the
// windowed frame bit-reversed into LDS as a complex sequence, log2(n_fft) radix-2 stages with the table twiddles, then the
// same power -> mel -> log (-> DCT) -> SpecAugment law as k_logmel.
template <typename WaveT>
__global__ __launch_bounds__(256) void k_logmel_any(const WaveT *__restrict__ wave, FeatArgs a, int n_fft, int log2n,
                                                    float *__restrict__ out, int use_mask, ww_mask_params mp,
                                                    int32_t *__restrict__ mask_idx) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *buf = reinterpret_cast<float2 *>(smem);                   // n_fft
    float *pw = reinterpret_cast<float *>(buf + n_fft);               // n_fft/2 + 1 (+ pad)
    float *lmv = pw + n_fft / 2 + 4;                                  // M
    float *fv = lmv + a.M;                                            // F (MFCC) -- == lmv otherwise
    int *msk = reinterpret_cast<int *>(fv + (a.use_dct ? a.F : 0));   // 2*WW_MAX_MASKS
    if (!a.use_dct) fv = lmv;
    const int tid = threadIdx.x, b = blockIdx.y, t = blockIdx.x;
    const WaveT *x = wave + (size_t)b * a.N;
    const int K = use_mask ? (mp.n_f + mp.n_t) : 0;
    if (tid < K) {
        int s, w;
        ww_specaug_mask(mp, (uint32_t)(mp.sample_offset + (uint64_t)b), tid, a.F, a.T, s, w);
        msk[2 * tid] = s;
        msk[2 * tid + 1] = w;
        if (mask_idx && t == 0) {
            mask_idx[((size_t)b * K + tid) * 2] = s;
            mask_idx[((size_t)b * K + tid) * 2 + 1] = w;
        }
    }
    const long base = (long)t * a.hop - n_fft / 2;
    for (int i = tid; i < n_fft; i += 256) {
        long idx = base + i;
        if (idx < 0) idx = -idx;
        if (idx >= a.N) idx = 2L * (a.N - 1) - idx;
        const float v = a.window[i] * load_sample(x, (size_t)idx);
        const int j = (int)(__brev((unsigned)i) >> (32 - log2n));
        buf[j] = make_float2(v, 0.f);
    }
    __syncthreads();
    for (int s = 0; s < log2n; ++s) {
        const int half = 1 << s, tstep = n_fft >> (s + 1);
        for (int k = tid; k < n_fft / 2; k += 256) {
            const int pos = k & (half - 1), i0 = ((k >> s) << (s + 1)) + pos, i1 = i0 + half;
            const float2 w = a.twiddle[pos * tstep], u = buf[i0], v = buf[i1];
            const float vr = v.x * w.x - v.y * w.y, vi = v.x * w.y + v.y * w.x;
            buf[i0] = make_float2(u.x + vr, u.y + vi);
            buf[i1] = make_float2(u.x - vr, u.y - vi);
        }
        __syncthreads();
    }
    for (int j = tid; j <= n_fft / 2; j += 256) pw[j] = buf[j].x * buf[j].x + buf[j].y * buf[j].y;
    __syncthreads();
    for (int m = tid; m < a.M; m += 256) {
        const int s0 = a.mel_start[m], L = a.mel_len[m];
        const float *wp = a.mel_w + a.mel_off[m];
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(wp[j], pw[s0 + j], acc);
        lmv[m] = logf(acc + a.log_eps);
    }
    __syncthreads();
    if (a.use_dct) {
        for (int c = tid; c < a.F; c += 256) {
            const float *d = a.dct + (size_t)c * a.M;
            float acc = 0.f;
            for (int m = 0; m < a.M; ++m) acc = fmaf(d[m], lmv[m], acc);
            fv[c] = acc;
        }
        __syncthreads();
    }
    for (int f = tid; f < a.F; f += 256) {
        float v = fv[f];
        for (int k = 0; k < K; ++k) {
            const int s = msk[2 * k], w = msk[2 * k + 1];
            const int pos = k < mp.n_f ? f : t;
            if (pos >= s && pos < s + w) v = 0.f;
        }
        out[((size_t)b * a.F + f) * a.T + t] = v;
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

int resolve_mask(const ww_ctx *ctx, const ww_specaug_cfg *sa, uint64_t seed, uint64_t step, uint64_t sample_offset,
                 ww_mask_params *mp) {
    mp->ctl = ctx->step_ctl;
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
    WW_REQUIRE(cfg->n_fft >= 64 && cfg->n_fft <= 4096 && (cfg->n_fft & (cfg->n_fft - 1)) == 0, WW_E_UNSUPPORTED,
               "ww_logmel_fwd: n_fft=%d (a power of two in [64, 4096] is implemented; the reference accepts 256 ... 4096)",
               cfg->n_fft);
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
    WW_REQUIRE(N > cfg->n_fft / 2, WW_E_INVALID, "ww_logmel_fwd: N=%d must exceed n_fft/2=%d for reflect padding", N,
               cfg->n_fft / 2);
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
    a.dct_t = tb->dct_t;
    // workgroup form: 8 waves / 32-frame items when the kernel has the device to itself, 4 waves / 16 frames when it runs
    // beside a training step with a caller-chosen number of workgroups (half the per-CU footprint); WW_LOGMEL_WAVES overrides
    static const int waves_env = ww_env_int("WW_LOGMEL_WAVES", 0);
    const int waves = (waves_env == 4 || waves_env == 8) ? waves_env : (ctx->logmel_wgs > 0 ? 4 : 8);
    const int FRW = 4 * waves;
    a.span_len = (FRW - 1) * cfg->hop + WW_NFFT;
    a.n_mel_w = tb->n_mel_w;
    a.melq_tab = tb->melq_tab; a.melq_w = tb->melq_w; a.n_melq_w = tb->n_melq_w;
    ww_mask_params mp = {};
    int use_mask = 0;
    if (sa) {
        if ((rc = resolve_mask(ctx, sa, seed, step, sample_offset, &mp))) return rc;
        use_mask = (mp.n_f + mp.n_t) > 0;
    }
    // n_fft <= 1024 runs on k_logmel (shorter frames zero-extended inside the 1024-sample window: their spectrum is every
    // (1024 / n_fft)-th bin of the 1024-point one, and the band weights sit there, ww_get_feat_tables); longer ones on the
    // general radix-2 kernel, one workgroup per (clip, frame)
    if (cfg->n_fft > WW_NFFT) {
        int log2n = 0;
        while ((1 << log2n) < cfg->n_fft) ++log2n;
        const size_t smem_any = (size_t)cfg->n_fft * sizeof(float2) + (size_t)(cfg->n_fft / 2 + 4) * sizeof(float) +
                                (size_t)a.M * sizeof(float) + (a.use_dct ? (size_t)a.F * sizeof(float) : 0) +
                                2 * WW_MAX_MASKS * sizeof(int);
        hipStream_t st_any = (hipStream_t)stream;
        ww_prof_scope ps_any(ctx, WW_K_LOGMEL, st_any);
        dim3 grid_any(a.T, B);
        if (wave_dtype == WW_WAVE_F32)
            hipLaunchKernelGGL(k_logmel_any<float>, grid_any, dim3(256), smem_any, st_any, (const float *)wave, a, cfg->n_fft,
                               log2n, out, use_mask, mp, mask_idx);
        else
            hipLaunchKernelGGL(k_logmel_any<int16_t>, grid_any, dim3(256), smem_any, st_any, (const int16_t *)wave, a,
                               cfg->n_fft, log2n, out, use_mask, mp, mask_idx);
        WW_LAUNCH_CHECK();
        return WW_OK;
    }
    // (8-wave form, 40 bands: 81 584 bytes -- two workgroups per CU fit the 160 KB with 336 bytes to spare; the band weights are
    //  12 KB of it.  Reading them from the L1-resident table instead (to keep the second workgroup with more bands) was built:
    //  a pointer chosen at run time is a generic pointer = flat loads, two typed paths cost the 8-wave form its last registers)
    const size_t smem = ((size_t)waves * XB + WW_NFFT + ((a.span_len + 3) & ~3) + (size_t)FRW * a.M +
                         2 * WW_MAX_MASKS + WW_MELQ_TAB + (size_t)((a.n_melq_w + 3) & ~3)) * sizeof(float);
    // Persistent grid: ctx->logmel_wgs workgroups (ww_ctx_set_logmel_workgroups; WW_LOGMEL_WGS overrides it for tuning),
    // 0 = one full residency round of the device.
    const int nblk = (a.T + FRW - 1) / FRW;
    const long nitems = (long)nblk * B;
    static const long wgs_env = [] { const char *e = getenv("WW_LOGMEL_WGS"); return e ? atol(e) : 0L; }();
    const bool f32 = wave_dtype == WW_WAVE_F32;
    const void *fn = waves == 8 ? (f32 ? (const void *)k_logmel<float, 8> : (const void *)k_logmel<int16_t, 8>)
                                : (f32 ? (const void *)k_logmel<float, 4> : (const void *)k_logmel<int16_t, 4>);
    WW_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    long wgs = wgs_env > 0 ? wgs_env : ctx->logmel_wgs;
    if (wgs <= 0) wgs = ww_occupancy_grid(fn, 64 * waves, smem, nitems, 1 << 20);
    dim3 grid((unsigned)(nitems < wgs ? nitems : wgs));
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_LOGMEL, st);
#define WW_LOGMEL_GO(T_, W_) hipLaunchKernelGGL((k_logmel<T_, W_>), grid, dim3(64 * W_), smem, st, (const T_ *)wave, a, out, use_mask, mp, mask_idx, nblk, nitems)
    if (waves == 8) { if (f32) WW_LOGMEL_GO(float, 8); else WW_LOGMEL_GO(int16_t, 8); }
    else { if (f32) WW_LOGMEL_GO(float, 4); else WW_LOGMEL_GO(int16_t, 4); }
#undef WW_LOGMEL_GO
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" int ww_specaug_apply(ww_ctx *ctx, float *x, int B, int F, int T, const ww_specaug_cfg *sa, uint64_t seed,
                                uint64_t step, uint64_t sample_offset, int32_t *mask_idx, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && sa, WW_E_INVALID, "ww_specaug_apply: null argument");
    WW_REQUIRE(B >= 0 && F >= 1 && T >= 1, WW_E_INVALID, "ww_specaug_apply: bad shape (%d,%d,%d)", B, F, T);
    ww_mask_params mp = {};
    int rc = resolve_mask(ctx, sa, seed, step, sample_offset, &mp);
    if (rc) return rc;
    if (B == 0 || mp.n_f + mp.n_t == 0) return WW_OK;
    hipLaunchKernelGGL(k_specaug_apply, dim3(B), dim3(256), 0, (hipStream_t)stream, x, B, F, T, mp, mask_idx);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
