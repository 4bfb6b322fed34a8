// Backward kernels of the depthwise-separable conv stack (channels-last, C = 64).
// y_l / g_l are stored as fp32 or bf16 (ww_act.h); arithmetic, sums and weight gradients are fp32.
//
// Stored tensors per layer l: y_l (PRE-BatchNorm conv output, written by forward) and
// g_l = dL/dz_l (gradient w.r.t. the BatchNorm output, ReLU mask already applied).
// One fused kernel per conv layer l does, in a single pass over HBM:
//     dy_l   = A*g_l + Bc*y_l + Cc              (BatchNorm backward folded into 3 per-channel
//                                                constants produced by the previous finalize)
//     dW_l  += a_{l-1}^T (*) dy_l               a_{l-1} = relu(bn(y_{l-1})) recomputed on load
//     da     = conv^T(dy_l, W_l)
//     g_{l-1} = da * [z_{l-1} > 0]  -> HBM ;  sum g_{l-1}, sum g_{l-1}*yhat_{l-1} -> slab
// i.e. reads g_l, y_l, y_{l-1} and writes g_{l-1}: 4 tensor passes per layer.
// Weight-gradient and statistic partials are per-block slabs summed in double by the
// finalize kernels (bit-reproducible; no float atomics).
#include "ww_internal.h"
#include <stdlib.h>
#include "ww_act.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int PW_LD = 68;

// ------------------------------------------------------------------------------------- head

// grid 8: block g owns channels 8g..8g+7 (the batch loop with its Philox draws is split 128 ways inside each block instead of
// 16 ways inside ONE block: the single-block form took 26 us of a 1.46 ms step)
__global__ __launch_bounds__(1024) void k_head_bwd(const float *__restrict__ dlogits, const float *__restrict__ pd,
                                                  const float *__restrict__ pool, int B, int HW,
                                                  const float *__restrict__ fc_w, float drop_scale,
                                                  uint64_t drop_thresh, int use_dropout, uint32_t seed_lo,
                                                  uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi,
                                                  uint64_t sample_offset, const float *__restrict__ gamma,
                                                  const float *__restrict__ mr, float *__restrict__ dfc_w,
                                                  float *__restrict__ dfc_b, float *__restrict__ dpool,
                                                  float *__restrict__ coef, float *__restrict__ dgamma,
                                                  float *__restrict__ dbeta, const ww_step_ctl *__restrict__ ctl) {
    __shared__ double sh[6][1024];
    __shared__ double sh2[6][16][8];
    ww_step_resolve(ctl, step_lo, step_hi, step_lo, step_hi);
    const int c = 8 * blockIdx.x + (threadIdx.x & 7), part = threadIdx.x >> 3;      // 128 batch parts
    const float w0 = fc_w[c], w1 = fc_w[64 + c];
    float mean_c = 0.f, rstd_c = 0.f, g_c = 0.f;          // for the constants at the end: fetched up front, not behind the sums
    if (threadIdx.x < 8) { mean_c = mr[c]; rstd_c = mr[64 + c]; g_c = gamma[c]; }
    const float inv_hw = 1.0f / (float)HW;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, s1 = 0.0, s2 = 0.0;
    // a part's clips (b = part, part + 128, ...) four at a time: their five loads each are issued as one batch (the guarded loop was
    // one dependent round trip per clip)
    for (int bb = part; bb < B; bb += 4 * 128) {
        float dl0[4], dl1[4], pv[4], pc[4], ph[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t b = (size_t)min(bb + 128 * u, B - 1);
            dl0[u] = dlogits[b * 2]; dl1[u] = dlogits[b * 2 + 1];
            pv[u] = pd[b * 64 + c];
            pc[u] = pool[b * 192 + 128 + c];  // count_{z>0}
            ph[u] = pool[b * 192 + 64 + c];   // sum_{z>0} yhat
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = bb + 128 * u;
            if (b < B) {
                a0 += (double)dl0[u] * pv[u];
                a1 += (double)dl1[u] * pv[u];
                b0 += dl0[u];
                b1 += dl1[u];
                float dp = fmaf(dl0[u], w0, dl1[u] * w1);
                if (use_dropout) {
                    uint32_t rr[4];
                    ww_philox(step_lo, step_hi, (uint32_t)(sample_offset + (uint64_t)b), (WW_TAG_DROPOUT << 24) | (uint32_t)(c >> 2),
                              seed_lo, seed_hi, rr);
                    const int q = c & 3;   // selected with compares: a dynamically indexed local array would live in scratch
                    const uint32_t rv = q == 0 ? rr[0] : q == 1 ? rr[1] : q == 2 ? rr[2] : rr[3];
                    dp = ((uint64_t)rv >= drop_thresh) ? dp * drop_scale : 0.f;
                }
                dp *= inv_hw;  // d mean / d element
                dpool[(size_t)b * 64 + c] = dp;
                s1 += (double)dp * pc[u];
                s2 += (double)dp * ph[u];
            }
        }
    }
    sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1; sh[2][threadIdx.x] = b0;
    sh[3][threadIdx.x] = b1; sh[4][threadIdx.x] = s1; sh[5][threadIdx.x] = s2;
    __syncthreads();
    // two-level fixed-order sum over the 128 parts: 6 quantities x 16 segments x 8 channels by 768 threads (8 parts each), then
    // 16 segments by 48 threads -- eight threads walking 768 LDS values one after the other was half of this kernel's 16 us
    if (threadIdx.x < 768) {
        const int k = threadIdx.x >> 7, seg = (threadIdx.x >> 3) & 15, c8 = threadIdx.x & 7;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) a += sh[k][8 * (8 * seg + q) + c8];
        sh2[k][seg][c8] = a;
    }
    __syncthreads();
    if (threadIdx.x < 48) {
        const int k = threadIdx.x >> 3, c8 = threadIdx.x & 7;
        double a = 0.0;
#pragma unroll
        for (int seg = 0; seg < 16; ++seg) a += sh2[k][seg][c8];
        sh[k][c8] = a;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = sh[k][threadIdx.x];
        dfc_w[c] = (float)t[0];
        dfc_w[64 + c] = (float)t[1];
        if (c == 0) {
            dfc_b[0] = (float)t[2];
            dfc_b[1] = (float)t[3];
        }
        const double count = (double)B * HW;
        const double mean = mean_c, rstd = rstd_c, g = g_c;
        const double c1 = t[4] / count, c2 = t[5] / count, A = g * rstd;
        coef[c] = (float)A;
        coef[64 + c] = (float)(-A * rstd * c2);
        coef[128 + c] = (float)(A * (mean * rstd * c2 - c1));
        dgamma[c] = (float)t[5];
        dbeta[c] = (float)t[4];
    }
}

// -------------------------------------------------------------------------------- pointwise
// LDS: dyt[64][68] | yit[64][68]  (34816 B).
// Wave roles inside a 64-pixel tile (4 waves):
//   dX : wave (rh, n)  -> pixels [32rh, 32rh+32) x input channels [32n, 32n+32)    (32 MFMA)
//   dW : wave (jt, kt) -> the 32x32 quadrant dW[32jt.., 32kt..] over all 64 pixels  (32 MFMA)
// so a lane keeps 32 weight registers and 16 persistent dW accumulators; the next tile's three input
// tensors are prefetched into 48 registers while the 64 MFMAs of this tile run.
constexpr int PWB_TILE = 64;

template <typename T, bool FROM_POOL>
__global__ __launch_bounds__(256, 2) void k_pw_bwd(const T *__restrict__ g, const float *__restrict__ dpool,
                                                   const T *__restrict__ y_out, const float *__restrict__ ss_out,
                                                   const float *__restrict__ coef, const T *__restrict__ y_in,
                                                   const float *__restrict__ ss_in, const float *__restrict__ mr_in,
                                                   const float *__restrict__ w, long M, int HW,
                                                   T *__restrict__ g_in, float *__restrict__ stat_partials,
                                                   float *__restrict__ dw_partials) {
    __shared__ __align__(16) float lds[(Act<T>::is_f32 ? 2 : 3) * PWB_TILE * PW_LD];
    float *dyt = lds, *yit = lds + PWB_TILE * PW_LD;
    float *otile = lds + 2 * PWB_TILE * PW_LD;     // bf16 only: g_in staged for packed stores
    typedef typename Act<T>::raw4 raw4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rh = wv >> 1, n = wv & 1;   // dX role; also (jt, kt) = (rh, n) for the dW role
    // dX B operand of k-step s: B[j = 32h + s][k = 32n + r] = w[j][k]
    float wt[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) wt[s] = w[(size_t)(32 * h + s) * 64 + 32 * n + r];
    const int c4 = tid & 15;
    const float4 cA = *reinterpret_cast<const float4 *>(coef + 4 * c4);
    const float4 cB = *reinterpret_cast<const float4 *>(coef + 64 + 4 * c4);
    const float4 cC = *reinterpret_cast<const float4 *>(coef + 128 + 4 * c4);
    float4 so = make_float4(0.f, 0.f, 0.f, 0.f), to = so;
    if (FROM_POOL) {
        so = *reinterpret_cast<const float4 *>(ss_out + 4 * c4);
        to = *reinterpret_cast<const float4 *>(ss_out + 64 + 4 * c4);
    }
    // input-layer BatchNorm constants of this lane's channel k = 32n + r
    const float sci = ss_in[32 * n + r], sfi = ss_in[64 + 32 * n + r];
    const float mui = mr_in[32 * n + r], rsi = mr_in[64 + 32 * n + r];
    floatx16 dwacc = {0.f};
    float st1 = 0.f, st2 = 0.f;

    const long ntiles = (M + PWB_TILE - 1) / PWB_TILE;
    // software pipeline (see k_pw_fwd): next tile's g / y_out / y_in loads fly during this tile's MFMAs
    raw4 rg[4], ro[4], ri[4];
    auto issue = [&](long ti) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long p = ti * PWB_TILE + (tid >> 4) + 16 * i;
            p = p < M ? p : M - 1;                       // clamped, branch-free; masked when consumed
            ro[i] = Act<T>::ldraw4(y_out + (size_t)p * 64 + 4 * c4);
            ri[i] = Act<T>::ldraw4(y_in + (size_t)p * 64 + 4 * c4);
            if (!FROM_POOL) rg[i] = Act<T>::ldraw4(g + (size_t)p * 64 + 4 * c4);
        }
    };
    if ((long)blockIdx.x < ntiles) issue(blockIdx.x);
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const long p0 = ti * PWB_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i;
            const long p = p0 + row;
            const bool ok = p < M;
            const float4 yo = Act<T>::cvt4(ro[i]);
            const float4 yr = Act<T>::cvt4(ri[i]);
            float4 dz;
            if (FROM_POOL) {
                const float4 dp = *reinterpret_cast<const float4 *>(dpool + (size_t)((uint32_t)(ok ? p : M - 1) / (uint32_t)HW) * 64 + 4 * c4);
                dz.x = fmaf(yo.x, so.x, to.x) > 0.f ? dp.x : 0.f;
                dz.y = fmaf(yo.y, so.y, to.y) > 0.f ? dp.y : 0.f;
                dz.z = fmaf(yo.z, so.z, to.z) > 0.f ? dp.z : 0.f;
                dz.w = fmaf(yo.w, so.w, to.w) > 0.f ? dp.w : 0.f;
            } else {
                dz = Act<T>::cvt4(rg[i]);
            }
            float4 dy, yi;
            dy.x = ok ? fmaf(cA.x, dz.x, fmaf(cB.x, yo.x, cC.x)) : 0.f;
            dy.y = ok ? fmaf(cA.y, dz.y, fmaf(cB.y, yo.y, cC.y)) : 0.f;
            dy.z = ok ? fmaf(cA.z, dz.z, fmaf(cB.z, yo.z, cC.z)) : 0.f;
            dy.w = ok ? fmaf(cA.w, dz.w, fmaf(cB.w, yo.w, cC.w)) : 0.f;
            yi.x = ok ? yr.x : 0.f; yi.y = ok ? yr.y : 0.f; yi.z = ok ? yr.z : 0.f; yi.w = ok ? yr.w : 0.f;
            *reinterpret_cast<float4 *>(dyt + row * PW_LD + 4 * c4) = dy;   // rows past M: dy = 0, y_in = 0
            *reinterpret_cast<float4 *>(yit + row * PW_LD + 4 * c4) = yi;
        }
        __syncthreads();
        if (ti + gridDim.x < ntiles) issue(ti + gridDim.x);
        // ---- dX = dy . W   (rows = pixels, K = output channel j, cols = input channel k)
        {
            const int rbase = 32 * rh;
            floatx16 acc = {0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(dyt + (rbase + r) * PW_LD + 32 * h + 4 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, wt[4 * j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, wt[4 * j + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, wt[4 * j + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, wt[4 * j + 3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int prow = rbase + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const long p = p0 + prow;
                const float yv = yit[prow * PW_LD + 32 * n + r];
                const float d = Act<T>::round1(fmaf(yv, sci, sfi) > 0.f ? acc[reg] : 0.f);
                if constexpr (Act<T>::is_f32) {
                    if (p < M) g_in[(size_t)p * 64 + 32 * n + r] = d;
                } else {
                    otile[prow * PW_LD + 32 * n + r] = d;
                }
                st1 += d;
                st2 = fmaf(d, (yv - mui) * rsi, st2);
            }
        }
        // ---- dW[j][k] += sum_p dy[p][j] * a[p][k] ; K = pixels, lane half h <-> pixel 32h + s
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const int prow = 32 * h + s;
            const float dyv = dyt[prow * PW_LD + 32 * rh + r];
            const float zv = fmaf(yit[prow * PW_LD + 32 * n + r], sci, sfi);
            const float av = zv < 0.f ? 0.f : zv;
            dwacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dyv, av, dwacc, 0, 0, 0);
        }
        __syncthreads();
        if constexpr (!Act<T>::is_f32) {
            // packed 8-byte stores of g_in; otile is rewritten only after the next tile's first barrier
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 4) + 16 * i;
                if (p0 + row < M)
                    Act<T>::st4(g_in + (size_t)(p0 + row) * 64 + 4 * c4,
                                *reinterpret_cast<const float4 *>(otile + row * PW_LD + 4 * c4));
            }
        }
    }
    // ---- block partials: statistics (sum over the two row-halves), dW quadrant straight from registers
    st1 += __shfl_xor(st1, 32);
    st2 += __shfl_xor(st2, 32);
    __syncthreads();
    float *shs = lds;  // [wave][kind][32]
    if (h == 0) {
        shs[wv * 64 + r] = st1;
        shs[wv * 64 + 32 + r] = st2;
    }
    __syncthreads();
    if (tid < 128) {
        const int kind = tid >> 6, c = tid & 63, nn = c >> 5, rr = c & 31;
        stat_partials[(size_t)blockIdx.x * 128 + tid] = shs[nn * 64 + kind * 32 + rr] + shs[(2 + nn) * 64 + kind * 32 + rr];
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int j = 32 * rh + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        dw_partials[(size_t)blockIdx.x * 4096 + j * 64 + 32 * n + r] = dwacc[reg];
    }
}

// ---- bf16 mode: dX and dW on v_mfma_f32_32x32x16_bf16.  Three bf16 [64][72] LDS tiles: dy, raw y_in (ReLU mask
// and yhat of the input layer in the epilogue) and a = relu(bn(y_in)).  dX reads dy rows with ds_read_b128; dW needs
// its operands pixel-major (K = pixels) and takes them from the same [pixel][channel] tiles with the transposing
// read ds_read_b64_tr_b16: a 16-lane group reads 4 pixels x 16 channels and lane i receives channel i's 4 pixels.
typedef short short4v __attribute__((ext_vector_type(4)));
constexpr int PWH_LD = 72;      // 16-bit elements per LDS row (144 B)

// A/B operand of a 32x32x16 MFMA whose K index is the PIXEL: channels c0..c0+31 (lane&31), pixels p0 + 8*(lane>>5) .. +7
template <typename H>
__device__ __forceinline__ typename H16<H>::x8 tr_operand(const H *tile, int p0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const H *base = tile + (p0 + 8 * (g >> 1) + q) * PWH_LD + c0 + 16 * (g & 1) + 4 * pp;
    typedef short4v __attribute__((address_space(3))) * lds_p;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * PWH_LD));
    return __builtin_bit_cast(typename H16<H>::x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// The layer's own output y_out is NOT read back: it is recomputed from the a tile with the forward kernel's exact
// MFMA chain (same operands, same k order, same RNE rounding to bf16 -> bit-identical to the tensor k_pw_fwd_bf16
// stored), which trades one activation-tensor read (97 MB at the full batch) for 4 MFMAs per wave and a third barrier.
// WIDE_IMG: an image has at least one tile of pixels (HW >= 64), so a tile touches at most two images and their pooled
// gradients are two registers; the per-pixel lookup (a division per element) is compiled only into the other variant.
template <typename H, bool FROM_POOL, bool WIDE_IMG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_pw_bwd_bf16(const H *__restrict__ g, const float *__restrict__ dpool,
                                                     const float *__restrict__ ss_out,
                                                     const float *__restrict__ coef, const H *__restrict__ y_in,
                                                     const float *__restrict__ ss_in, const float *__restrict__ mr_in,
                                                     const float *__restrict__ w, long M, int HW,
                                                     H *__restrict__ g_in, float *__restrict__ stat_partials,
                                                     float *__restrict__ dw_partials, int rev) {
    // dy, a = relu(bn(y_in)), and two y_in tiles (even / odd tile of the unrolled loop): g_in overwrites the raw y_in tile
    // in place and is stored from there while the next tile is staged into the other one.  36.9 KB -> 4 workgroups per CU.
    // + the 64x64 weights (bf16): both MFMA B operands come from this one copy (row reads forward, transposing reads for dX)
    __shared__ __align__(16) H tiles[5 * PWB_TILE * PWH_LD];
    H *dyt = tiles, *at = tiles + PWB_TILE * PWH_LD, *yit0 = tiles + 2 * PWB_TILE * PWH_LD, *yit1 = tiles + 3 * PWB_TILE * PWH_LD;
    H *wtile = tiles + 4 * PWB_TILE * PWH_LD;
    typedef Act<H> A16;
    typedef typename H16<H>::x8 bf16x8;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rh = wv >> 1, n = wv & 1;   // y/dX: pixels [32rh,+32) x channels [32n,+32); dW: quadrant (jt,kt) = (rh,n)
    const int ch = 32 * n + r;
    // forward B operand of k-step t: B[k = 16t + 8h + jj][j = ch] = w[ch][k] (ds_read_b128 of a wtile row);
    // dX B operand:                  B[j = 16t + 8h + jj][k = ch] = w[j][ch]  (transposing read of the same tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 4) + 16 * i, c = 4 * (tid & 15);
        const float4 v = *reinterpret_cast<const float4 *>(w + (size_t)row * 64 + c);
        *reinterpret_cast<uint2 *>(wtile + row * PWH_LD + c) = make_uint2(A16::pack2(v.x, v.y), A16::pack2(v.z, v.w));
    }
    const int c4 = tid & 15;
    const float cA = coef[ch], cB = coef[64 + ch], cC = coef[128 + ch];
    const float4 si = *reinterpret_cast<const float4 *>(ss_in + 4 * c4);
    const float4 ti4 = *reinterpret_cast<const float4 *>(ss_in + 64 + 4 * c4);
    float so = 0.f, to = 0.f;
    if (FROM_POOL) {
        so = ss_out[ch];
        to = ss_out[64 + ch];
    }
    const float sci = ss_in[ch], sfi = ss_in[64 + ch];
    const float mui = mr_in[ch], rsi = mr_in[64 + ch];
    floatx16 dwacc = {0.f};
    float st1 = 0.f, st2 = 0.f;

    const long ntiles = (M + PWB_TILE - 1) / PWB_TILE;
    const long last_img = (M - 1) / HW;
    typename A16::raw4 rg0[4], ri0[4];
    float dpA0 = 0.f, dpB0 = 0.f;   // pooled gradient of the (at most two, when HW >= tile) images of a tile
    auto issue = [&](long ti, typename A16::raw4 (&rg)[4], typename A16::raw4 (&ri)[4], float &ndpA, float &ndpB) {
        if (ti >= ntiles) return;
        if (rev) ti = ntiles - 1 - ti;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long p = ti * PWB_TILE + (tid >> 4) + 16 * i;
            p = p < M ? p : M - 1;
            ri[i] = A16::ldraw4(y_in + (size_t)p * 64 + 4 * c4);
            if (!FROM_POOL) rg[i] = A16::ldraw4_nt(g + (size_t)p * 64 + 4 * c4);
        }
        if (FROM_POOL && WIDE_IMG) {
            const long b0 = (long)((uint32_t)(ti * PWB_TILE) / (uint32_t)HW);   // M < 2^31 (checked on the host): 32-bit division
            ndpA = dpool[(size_t)b0 * 64 + ch];
            ndpB = dpool[(size_t)(b0 < last_img ? b0 + 1 : last_img) * 64 + ch];
        }
    };
    auto tile = [&](long ti, H *yit, typename A16::raw4 (&rg)[4], typename A16::raw4 (&ri)[4], float &ndpA, float &ndpB) {
        const long p0 = (rev ? ntiles - 1 - ti : ti) * PWB_TILE;
        const float dpA = ndpA, dpB = ndpB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i;
            const bool ok = p0 + row < M;
            const float4 yr = A16::cvt4(ri[i]);
            const float z0 = fmaf(yr.x, si.x, ti4.x), z1 = fmaf(yr.y, si.y, ti4.y);
            const float z2 = fmaf(yr.z, si.z, ti4.z), z3 = fmaf(yr.w, si.w, ti4.w);
            const float a0 = !ok || z0 < 0.f ? 0.f : z0, a1 = !ok || z1 < 0.f ? 0.f : z1;
            const float a2 = !ok || z2 < 0.f ? 0.f : z2, a3 = !ok || z3 < 0.f ? 0.f : z3;
            *reinterpret_cast<uint2 *>(at + row * PWH_LD + 4 * c4) = make_uint2(A16::pack2(a0, a1), A16::pack2(a2, a3));
            *reinterpret_cast<uint2 *>(yit + row * PWH_LD + 4 * c4) = ok ? ri[i] : make_uint2(0u, 0u);
            if (!FROM_POOL) *reinterpret_cast<uint2 *>(dyt + row * PWH_LD + 4 * c4) = ok ? rg[i] : make_uint2(0u, 0u);   // raw dz; dy in place below
        }
        issue(ti + (long)gridDim.x, rg, ri, ndpA, ndpB);
        __syncthreads();
        const int rbase = 32 * rh;
        // ---- y = a . W^T as the forward computed it, then dy = A dz + B y + C in place (each element is owned by one lane)
        {
            floatx16 yacc = {0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(at + (rbase + r) * PWH_LD + 16 * t + 8 * h);
                const bf16x8 wf = *reinterpret_cast<const bf16x8 *>(wtile + ch * PWH_LD + 16 * t + 8 * h);
                yacc = H16<H>::mfma32(a, wf, yacc);
            }
            const int p0i = (int)p0, Mi = (int)M;
            const int bnd = FROM_POOL ? (int)(((uint32_t)p0 / (uint32_t)HW + 1u) * (uint32_t)HW) : 0;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int prow = rbase + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int p = p0i + prow;                      // M < 2^31 (host check)
                const float okf = p < Mi ? 1.f : 0.f;          // a factor, not a select: keeps the 16 LDS reads branch-free
                const float yv = A16::round1(yacc[reg]);       // (rows >= M hold dz = 0, a = 0 -> y = 0: the product is finite)
                float dz;
                if (FROM_POOL) {
                    const float dp = WIDE_IMG ? (p >= bnd ? dpB : dpA) : dpool[(size_t)((uint32_t)min(p, Mi - 1) / (uint32_t)HW) * 64 + ch];
                    dz = fmaf(yv, so, to) > 0.f ? dp : 0.f;
                } else {
                    dz = (float)dyt[prow * PWH_LD + ch];
                }
                const float d = okf * fmaf(cA, dz, fmaf(cB, yv, cC));
                dyt[prow * PWH_LD + ch] = (H)d;
            }
        }
        __syncthreads();
        // ---- dX = dy . W
        {
            floatx16 acc = {0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(dyt + (rbase + r) * PWH_LD + 16 * t + 8 * h);
                const bf16x8 wt = tr_operand(wtile, 16 * t, 32 * n, lane);
                acc = H16<H>::mfma32(a, wt, acc);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int prow = rbase + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const float yv = (float)yit[prow * PWH_LD + ch];
                const float d = A16::round1(fmaf(yv, sci, sfi) > 0.f ? acc[reg] : 0.f);
                yit[prow * PWH_LD + ch] = (H)d;
                st1 += d;
                st2 = fmaf(d, (yv - mui) * rsi, st2);
            }
        }
        // ---- dW[j][k] += sum_p dy[p][j] * a[p][k]   (K = the tile's 64 pixels, 4 k-steps, transposing LDS reads)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 dyop = tr_operand(dyt, 16 * t, 32 * rh, lane);
            const bf16x8 aop = tr_operand(at, 16 * t, 32 * n, lane);
            dwacc = H16<H>::mfma32(dyop, aop, dwacc);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 4) + 16 * i;
            if (p0 + row < M)
                *reinterpret_cast<uint2 *>(g_in + (size_t)(p0 + row) * 64 + 4 * c4) =
                    *reinterpret_cast<const uint2 *>(yit + row * PWH_LD + 4 * c4);
        }
    };
    issue(blockIdx.x, rg0, ri0, dpA0, dpB0);
    for (long ti = blockIdx.x; ti < ntiles; ti += 2 * (long)gridDim.x) {
        tile(ti, yit0, rg0, ri0, dpA0, dpB0);
        if (ti + gridDim.x < ntiles) tile(ti + gridDim.x, yit1, rg0, ri0, dpA0, dpB0);
    }
    st1 += __shfl_xor(st1, 32);
    st2 += __shfl_xor(st2, 32);
    __syncthreads();
    float *shs = reinterpret_cast<float *>(tiles);  // [wave][kind][32]
    if (h == 0) {
        shs[wv * 64 + r] = st1;
        shs[wv * 64 + 32 + r] = st2;
    }
    __syncthreads();
    if (tid < 128) {
        const int kind = tid >> 6, c = tid & 63, nn = c >> 5, rr = c & 31;
        stat_partials[(size_t)blockIdx.x * 128 + tid] = shs[nn * 64 + kind * 32 + rr] + shs[(2 + nn) * 64 + kind * 32 + rr];
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int j = 32 * rh + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        dw_partials[(size_t)blockIdx.x * 4096 + j * 64 + 32 * n + r] = dwacc[reg];
    }
}

// -------------------------------------------------------------------------------- depthwise
struct DwGeom {
    int B, H, W, ncs, nseg, hs_len;
    long items;
};

constexpr int DW_HS = 20;   // max rows per strip segment (matches ww_conv_fwd.hip)

template <typename T>
__device__ __forceinline__ void dwb_issue(const T *__restrict__ gimg, const T *__restrict__ yimg, int h, int w0, int H,
                                          int W, int cl, typename Act<T>::raw2 (&rg)[6],
                                          typename Act<T>::raw2 (&ry)[6]) {
    const int hh = min(max(h, 0), H - 1);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int wc = min(max(w0 - 1 + i, 0), W - 1);
        const uint32_t o = (uint32_t)((hh * W + wc) * 64 + 2 * cl);   // one image < 2^32 elements: 32-bit offsets
        rg[i] = Act<T>::ldraw2(gimg + o);
        ry[i] = Act<T>::ldraw2(yimg + o);
    }
}
// dy = A*g + Bc*y + Cc inside the image, 0 outside (zero padding of the transposed conv)
template <typename T>
__device__ __forceinline__ void dwb_finish(const typename Act<T>::raw2 (&rg)[6], const typename Act<T>::raw2 (&ry)[6],
                                           int h, int w0, int H, int W, float2 cA, float2 cB, float2 cC,
                                           float2 (&r)[6]) {
    const bool hv = (h >= 0) && (h < H);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int wc = w0 - 1 + i;
        const bool ok = hv && wc >= 0 && wc < W;
        const float2 gv = Act<T>::cvt2(rg[i]), yv = Act<T>::cvt2(ry[i]);
        r[i].x = ok ? fmaf(cA.x, gv.x, fmaf(cB.x, yv.x, cC.x)) : 0.f;
        r[i].y = ok ? fmaf(cA.y, gv.y, fmaf(cB.y, yv.y, cC.y)) : 0.f;
    }
}
template <typename T>
__device__ __forceinline__ void dwb_issue_centre(const T *__restrict__ yin_img, int h, int w0, int H, int W, int cl,
                                                 typename Act<T>::raw2 (&rc)[4]) {
    const int hh = min(max(h, 0), H - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int wc = min(w0 + i, W - 1);
        rc[i] = Act<T>::ldraw2_nt(yin_img + (uint32_t)((hh * W + wc) * 64 + 2 * cl));
    }
}

// centre row h:  rs0 = dy[h+1] (pairs with weight row 0), rs1 = dy[h], rs2 = dy[h-1]; yc = y_in[h][w0..w0+3]
template <typename T>
__device__ __forceinline__ void dwb_row(const typename Act<T>::raw2 (&yc)[4], T *__restrict__ gin_img, int h, int w0,
                                        int W, int cl,
                                        const float2 (&rs0)[6], const float2 (&rs1)[6], const float2 (&rs2)[6],
                                        const float (&wa)[9], const float (&wb)[9], float2 sc, float2 sf,
                                        float (&dwa)[9], float (&dwb)[9], float &s1a, float &s1b, float &s2a,
                                        float &s2b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (w0 + i < W) {
            const float2 yv = Act<T>::cvt2(yc[i]);
            const float z0 = fmaf(yv.x, sc.x, sf.x), z1 = fmaf(yv.y, sc.y, sf.y);
            const float a0 = z0 < 0.f ? 0.f : z0, a1 = z1 < 0.f ? 0.f : z1;
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float2 t0 = rs0[i + 2 - kw], t1 = rs1[i + 2 - kw], t2 = rs2[i + 2 - kw];
                d0 = fmaf(t0.x, wa[kw], d0);     d1 = fmaf(t0.y, wb[kw], d1);
                d0 = fmaf(t1.x, wa[3 + kw], d0); d1 = fmaf(t1.y, wb[3 + kw], d1);
                d0 = fmaf(t2.x, wa[6 + kw], d0); d1 = fmaf(t2.y, wb[6 + kw], d1);
                dwa[kw] = fmaf(a0, t0.x, dwa[kw]);         dwb[kw] = fmaf(a1, t0.y, dwb[kw]);
                dwa[3 + kw] = fmaf(a0, t1.x, dwa[3 + kw]); dwb[3 + kw] = fmaf(a1, t1.y, dwb[3 + kw]);
                dwa[6 + kw] = fmaf(a0, t2.x, dwa[6 + kw]); dwb[6 + kw] = fmaf(a1, t2.y, dwb[6 + kw]);
            }
            const float2 go = Act<T>::round2(make_float2(z0 > 0.f ? d0 : 0.f, z1 > 0.f ? d1 : 0.f));
            const float g0 = go.x, g1 = go.y;
            Act<T>::st2(gin_img + (uint32_t)((h * W + w0 + i) * 64 + 2 * cl), go);
            s1a += g0; s1b += g1;
            s2a = fmaf(g0, yv.x, s2a);      // sum g*y; turned into sum g*yhat once, after the loop (4 VGPRs less in it:
            s2b = fmaf(g1, yv.y, s2b);      // with them the bf16 kernel spilled, and a kernel with scratch pays ~6 us on
                                            // each side of its dispatch)
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256, Act<T>::is_f32 ? 2 : 3) void k_dw_bwd(const T *__restrict__ g, const T *__restrict__ y_out,
                                                const float *__restrict__ coef, const T *__restrict__ y_in,
                                                const float *__restrict__ ss_in, const float *__restrict__ mr_in,
                                                const float *__restrict__ w, DwGeom gm, T *__restrict__ g_in,
                                                float *__restrict__ stat_partials, float *__restrict__ dw_partials) {
    __shared__ float sh[8 * 576];
    typedef typename Act<T>::raw2 raw2;
    const int tid = threadIdx.x, slot = tid >> 5, cl = tid & 31;
    float wa[9], wb[9], dwa[9], dwb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wa[t] = w[(2 * cl) * 9 + t];
        wb[t] = w[(2 * cl + 1) * 9 + t];
        dwa[t] = 0.f;
        dwb[t] = 0.f;
    }
    const float2 cA = *reinterpret_cast<const float2 *>(coef + 2 * cl);
    const float2 cB = *reinterpret_cast<const float2 *>(coef + 64 + 2 * cl);
    const float2 cC = *reinterpret_cast<const float2 *>(coef + 128 + 2 * cl);
    const float2 sc = *reinterpret_cast<const float2 *>(ss_in + 2 * cl);
    const float2 sf = *reinterpret_cast<const float2 *>(ss_in + 64 + 2 * cl);
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
    const size_t img_stride = (size_t)gm.H * gm.W * 64;
    for (long item = (long)blockIdx.x * 8 + slot; item < gm.items; item += (long)gridDim.x * 8) {
        const int cs = (int)(item % gm.ncs);
        const long t = item / gm.ncs;
        const int seg = (int)(t % gm.nseg), b = (int)(t / gm.nseg);
        const int w0 = cs * 4, hs = seg * gm.hs_len;
        const int he = min(gm.H, hs + gm.hs_len);
        const T *gimg = g + (size_t)b * img_stride;
        const T *yoimg = y_out + (size_t)b * img_stride;
        const T *yiimg = y_in + (size_t)b * img_stride;
        T *giimg = g_in + (size_t)b * img_stride;
        // rows[i%3] = dy[h-1], rows[(i+1)%3] = dy[h], rows[(i+2)%3] = dy[h+1]; raw dy row h+2 and the raw centre row
        // h+1 are in flight while row h is processed
        float2 rows[3][6];
        raw2 rg[6], ry[6], ag[6], ay[6], yc[4], ayc[4];
        dwb_issue<T>(gimg, yoimg, hs - 1, w0, gm.H, gm.W, cl, rg, ry);
        dwb_finish<T>(rg, ry, hs - 1, w0, gm.H, gm.W, cA, cB, cC, rows[0]);
        dwb_issue<T>(gimg, yoimg, hs, w0, gm.H, gm.W, cl, rg, ry);
        dwb_finish<T>(rg, ry, hs, w0, gm.H, gm.W, cA, cB, cC, rows[1]);
        dwb_issue<T>(gimg, yoimg, hs + 1, w0, gm.H, gm.W, cl, ag, ay);
        dwb_issue_centre<T>(yiimg, hs, w0, gm.H, gm.W, cl, ayc);
#pragma unroll 1
        for (int i0 = 0; i0 < DW_HS; i0 += 3) {
#pragma unroll
          for (int ii = 0; ii < 3; ++ii) {
            const int i = i0 + ii;           // i % 3 == ii: the row rotation stays compile-time
            const int h = hs + i;
            if (h < he) {
#pragma unroll
                for (int c = 0; c < 6; ++c) { rg[c] = ag[c]; ry[c] = ay[c]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) yc[c] = ayc[c];
                if (h + 1 < he) {
                    dwb_issue<T>(gimg, yoimg, h + 2, w0, gm.H, gm.W, cl, ag, ay);
                    dwb_issue_centre<T>(yiimg, h + 1, w0, gm.H, gm.W, cl, ayc);
                }
                dwb_finish<T>(rg, ry, h + 1, w0, gm.H, gm.W, cA, cB, cC, rows[(ii + 2) % 3]);
                dwb_row<T>(yc, giimg, h, w0, gm.W, cl, rows[(ii + 2) % 3], rows[(ii + 1) % 3], rows[ii % 3], wa, wb, sc, sf,
                        dwa, dwb, s1a, s1b, s2a, s2b);
            }
          }
        }
    }
    // statistics partial: sum g*yhat = rstd * (sum g*y - mean * sum g)
    {
        const float2 mu = *reinterpret_cast<const float2 *>(mr_in + 2 * cl);
        const float2 rsd = *reinterpret_cast<const float2 *>(mr_in + 64 + 2 * cl);
        s2a = rsd.x * (s2a - mu.x * s1a);
        s2b = rsd.y * (s2b - mu.y * s1b);
    }
    sh[slot * 128 + 2 * cl] = s1a;       sh[slot * 128 + 2 * cl + 1] = s1b;
    sh[slot * 128 + 64 + 2 * cl] = s2a;  sh[slot * 128 + 64 + 2 * cl + 1] = s2b;
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) t += sh[s * 128 + tid];
        stat_partials[(size_t)blockIdx.x * 128 + tid] = t;
    }
    __syncthreads();
    // weight-gradient partial, layout [c][tap] like Conv2d.weight (64,1,3,3)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        sh[slot * 576 + (2 * cl) * 9 + t] = dwa[t];
        sh[slot * 576 + (2 * cl + 1) * 9 + t] = dwb[t];
    }
    __syncthreads();
    for (int i = tid; i < 576; i += 256) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) t += sh[s * 576 + i];
        dw_partials[(size_t)blockIdx.x * 576 + i] = t;
    }
}

// ------------------------------------------------------------------------------------- stem
// RECOMP: the layer's own output is recomputed from the input rows that sit in LDS for the weight gradient anyway (the
// forward kernel's nine fmaf in the same order, then its rounding: bit-identical to the stored tensor) instead of read
// back -- the stem reads 2A -> 1A per launch.  Needs the stem weights, which the stand-alone C entry point does not get
// (its signature predates this): the whole-model path passes them, ww_conv_stem_bwd reads y_out.
template <typename T, bool RECOMP>
__global__ __launch_bounds__(256) void k_stem_bwd(const T *__restrict__ g, const T *__restrict__ y_out,
                                                  const float *__restrict__ w, const float *__restrict__ coef,
                                                  const float *__restrict__ x, int B, int Hin, int Win, int Ho, int Wo,
                                                  float *__restrict__ dw_partials, int rev) {
    __shared__ float sh[8 * 576];
    extern __shared__ float xs[];            // [3][Win + 2] zero-padded input rows (see k_stem_fwd)
    const int tid = threadIdx.x, slot = tid >> 5, cl = tid & 31;
    const int ld = Win + 2;
    const float2 cA = *reinterpret_cast<const float2 *>(coef + 2 * cl);
    const float2 cB = *reinterpret_cast<const float2 *>(coef + 64 + 2 * cl);
    const float2 cC = *reinterpret_cast<const float2 *>(coef + 128 + 2 * cl);
    float dwa[9], dwb[9], w0[9], w1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        dwa[t] = dwb[t] = 0.f;
        w0[t] = RECOMP ? w[(2 * cl) * 9 + t] : 0.f;
        w1[t] = RECOMP ? w[(2 * cl + 1) * 9 + t] : 0.f;
    }
    const long nrows = (long)B * Ho;
    // the three input rows of an item are fetched into registers one item AHEAD and written to LDS at the top of the item
    // (3 * ld <= 2 * 256 values; wider inputs are staged in place); the row's gradient pixels of a slot are loaded in ONE
    // batch of up to ten before they are used -- one exposed HBM latency per row instead of one per pixel
    float pre[2];
    auto fetch = [&](long item) {
        if (rev) item = nrows - 1 - item;
        const int fb_ = (int)(item / Ho), foh = (int)(item - (long)fb_ * Ho);
        const float *xb = x + (size_t)fb_ * Hin * Win;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            const int kh = i / ld, c = i - kh * ld;
            const int ih = 2 * foh - 1 + kh, iw = c - 1;
            pre[k] = (i < 3 * ld && ih >= 0 && ih < Hin && iw >= 0 && iw < Win) ? xb[(size_t)ih * Win + iw] : 0.f;
        }
    };
    const bool ahead = 3 * ld <= 2 * 256;
    if (ahead && (long)blockIdx.x < nrows) fetch(blockIdx.x);
    for (long rowi = blockIdx.x; rowi < nrows; rowi += gridDim.x) {
        const long row = rev ? nrows - 1 - rowi : rowi;
        const int b = (int)(row / Ho), oh = (int)(row - (long)b * Ho);
        __syncthreads();
        if (ahead) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (tid + 256 * k < 3 * ld) xs[tid + 256 * k] = pre[k];
        } else {
            const float *xb = x + (size_t)b * Hin * Win;
            for (int i = tid; i < 3 * ld; i += 256) {
                const int kh = i / ld, c = i - kh * ld;
                const int ih = 2 * oh - 1 + kh, iw = c - 1;
                xs[i] = (ih >= 0 && ih < Hin && iw >= 0 && iw < Win) ? xb[(size_t)ih * Win + iw] : 0.f;
            }
        }
        __syncthreads();
        if (ahead && rowi + gridDim.x < nrows) fetch(rowi + gridDim.x);
        for (int ow0 = slot; ow0 < Wo; ow0 += 80) {
            typename Act<T>::raw2 gr[10], yr[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int ow = min(ow0 + 8 * j, Wo - 1);
                const size_t o = ((size_t)row * Wo + ow) * 64 + 2 * cl;
                gr[j] = Act<T>::ldraw2_nt(g + o);
                if constexpr (!RECOMP) yr[j] = Act<T>::ldraw2(y_out + o);
            }
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int ow = ow0 + 8 * j;
                if (ow < Wo) {
                    const float2 gz = Act<T>::cvt2(gr[j]);
                    float v[9];
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) v[kh * 3 + kw] = xs[kh * ld + 2 * ow + kw];
                    float2 yo;
                    if constexpr (RECOMP) {
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int t = 0; t < 9; ++t) {            // k_stem_fwd's accumulation order
                            a0 = fmaf(v[t], w0[t], a0);
                            a1 = fmaf(v[t], w1[t], a1);
                        }
                        yo = Act<T>::round2(make_float2(a0, a1));
                    } else {
                        yo = Act<T>::cvt2(yr[j]);
                    }
                    const float d0 = fmaf(cA.x, gz.x, fmaf(cB.x, yo.x, cC.x));
                    const float d1 = fmaf(cA.y, gz.y, fmaf(cB.y, yo.y, cC.y));
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        dwa[t] = fmaf(d0, v[t], dwa[t]);
                        dwb[t] = fmaf(d1, v[t], dwb[t]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        sh[slot * 576 + (2 * cl) * 9 + t] = dwa[t];
        sh[slot * 576 + (2 * cl + 1) * 9 + t] = dwb[t];
    }
    __syncthreads();
    for (int i = tid; i < 576; i += 256) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) t += sh[s * 576 + i];
        dw_partials[(size_t)blockIdx.x * 576 + i] = t;
    }
}

}  // namespace

extern "C" int ww_head_bwd(ww_ctx *ctx, const float *dlogits, const float *pd, const float *pool, int B, int HW,
                           const float *fc_w, float dropout_p, int training, uint64_t seed, uint64_t step,
                           uint64_t sample_offset, const float *gamma_last, const float *mr_last, float *dfc_w,
                           float *dfc_b, float *dpool, float *coef_last, float *dgamma_last, float *dbeta_last,
                           ww_stream_t stream) {
    WW_REQUIRE(ctx && dlogits && pd && pool && fc_w && gamma_last && mr_last && dfc_w && dfc_b && dpool && coef_last &&
                   dgamma_last && dbeta_last,
               WW_E_INVALID, "ww_head_bwd: null argument");
    WW_REQUIRE(B >= 1 && HW >= 1, WW_E_INVALID, "ww_head_bwd: bad shape (%d,%d)", B, HW);
    WW_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, WW_E_INVALID, "ww_head_bwd: dropout_p=%f not in [0,1)", dropout_p);
    const int use_dropout = training && dropout_p > 0.f;
    const float scale = (float)(1.0 / (1.0 - (double)dropout_p));
    ww_prof_scope ps_(ctx, WW_K_HEAD_LOSS, (hipStream_t)stream);
    hipLaunchKernelGGL(k_head_bwd, dim3(8), dim3(1024), 0, (hipStream_t)stream, dlogits, pd, pool, B, HW, fc_w, scale,
                       ww_prob_threshold((double)dropout_p), use_dropout, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)step, (uint32_t)(step >> 32), sample_offset, gamma_last, mr_last, dfc_w, dfc_b, dpool,
                       coef_last, dgamma_last, dbeta_last, ctx->step_ctl);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

namespace {
int check_act_b(const char *who, int act_dtype) {
    WW_REQUIRE(act_dtype == WW_ACT_F32 || act_dtype == WW_ACT_BF16 || act_dtype == WW_ACT_F16, WW_E_INVALID, "%s: unknown act_dtype %d", who,
               act_dtype);
    return WW_OK;
}

template <typename H>
int launch_pw_bwd_bf16(ww_ctx *ctx, const void *g, const float *dpool, const void *y_out, const float *ss_out,
                       const float *coef, const void *y_in, const float *ss_in, const float *mr_in, const float *w, long M,
                       int HW, void *g_in, float *stat, float *dwp, int *grid_out, hipStream_t st) {
    typedef const H *cp;
    const long ntiles = (M + PWB_TILE - 1) / PWB_TILE;
    int grid;
    ww_prof_scope ps_(ctx, WW_K_PW_BWD, st);
    auto go = [&](auto kern) {
        grid = ww_occupancy_grid((const void *)kern, 256, 0, ntiles, WW_DW_SLAB_ROWS);
        static const int rev = ww_env_int("WW_PW_BWD_REV", 1);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, (cp)g, dpool, ss_out, coef, (cp)y_in, ss_in, mr_in, w, M, HW,
                           (H *)g_in, stat, dwp, rev);
    };
    if (g) go(k_pw_bwd_bf16<H, false, true>);
    else if (HW >= PWB_TILE) go(k_pw_bwd_bf16<H, true, true>);
    else go(k_pw_bwd_bf16<H, true, false>);
    *grid_out = grid;
    return WW_OK;
}

template <typename T>
int launch_pw_bwd(ww_ctx *ctx, const void *g, const float *dpool, const void *y_out, const float *ss_out,
                  const float *coef, const void *y_in, const float *ss_in, const float *mr_in, const float *w, long M,
                  int HW, void *g_in, float *stat, float *dwp, int *grid_out, hipStream_t st) {
    const long ntiles = (M + PWB_TILE - 1) / PWB_TILE;
    int grid;
    ww_prof_scope ps_(ctx, WW_K_PW_BWD, st);
    if (g) {
        grid = ww_occupancy_grid((const void *)k_pw_bwd<T, false>, 256, 0, ntiles, WW_DW_SLAB_ROWS);
        hipLaunchKernelGGL((k_pw_bwd<T, false>), dim3(grid), dim3(256), 0, st, (const T *)g, dpool, (const T *)y_out,
                           ss_out, coef, (const T *)y_in, ss_in, mr_in, w, M, HW, (T *)g_in, stat, dwp);
    } else {
        grid = ww_occupancy_grid((const void *)k_pw_bwd<T, true>, 256, 0, ntiles, WW_DW_SLAB_ROWS);
        hipLaunchKernelGGL((k_pw_bwd<T, true>), dim3(grid), dim3(256), 0, st, (const T *)g, dpool, (const T *)y_out,
                           ss_out, coef, (const T *)y_in, ss_in, mr_in, w, M, HW, (T *)g_in, stat, dwp);
    }
    *grid_out = grid;
    return WW_OK;
}

template <typename T>
int launch_dw_bwd(ww_ctx *ctx, const void *g, const void *y_out, const float *coef, const void *y_in,
                  const float *ss_in, const float *mr_in, const float *w, const DwGeom &gm, void *g_in, float *stat,
                  float *dwp, int *grid_out, hipStream_t st) {
    const long nblk = (gm.items + 7) / 8;
    const int grid = ww_occupancy_grid((const void *)k_dw_bwd<T>, 256, 0, nblk, WW_MAX_PARTIALS);
    ww_prof_scope ps_(ctx, WW_K_DW_BWD, st);
    hipLaunchKernelGGL(k_dw_bwd<T>, dim3(grid), dim3(256), 0, st, (const T *)g, (const T *)y_out, coef, (const T *)y_in,
                       ss_in, mr_in, w, gm, (T *)g_in, stat, dwp);
    *grid_out = grid;
    return WW_OK;
}

template <typename T>
int launch_stem_bwd(ww_ctx *ctx, const void *g, const void *y_out, const float *w, const float *coef, const float *x, int B,
                    int Hin, int Win, float *dwp, int *grid_out, hipStream_t st) {
    const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2;
    const long nrows = (long)B * Ho;
    const size_t smem = (size_t)3 * (Win + 2) * sizeof(float);
    ww_prof_scope ps_(ctx, WW_K_STEM_BWD, st);
    auto go = [&](auto kern) {
        // only the weight-gradient slab is used (576 columns): room for 2048 rows -> full occupancy
        const int grid = ww_occupancy_grid((const void *)kern, 256, smem, nrows, 2048);
        static const int rev = ww_env_int("WW_STEM_BWD_REV", 1);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, (const T *)g, (const T *)y_out, w, coef, x, B, Hin, Win, Ho,
                           Wo, dwp, rev);
        *grid_out = grid;
    };
    if (w) go(k_stem_bwd<T, true>);
    else go(k_stem_bwd<T, false>);
    return WW_OK;
}
}  // namespace

extern "C" int ww_pwconv1x1_bwd(ww_ctx *ctx, int act_dtype, const void *g, const float *dpool, const void *y_out,
                                const float *ss_out, const float *coef, const void *y_in, const float *ss_in,
                                const float *mr_in, const float *gamma_in, const float *w, int B, int H, int W,
                                void *g_in, float *dw, float *coef_in, float *dgamma_in, float *dbeta_in,
                                void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && y_out && coef && y_in && ss_in && mr_in && gamma_in && w && g_in && dw && coef_in && dgamma_in &&
                   dbeta_in && scratch,
               WW_E_INVALID, "ww_pwconv1x1_bwd: null argument");
    WW_REQUIRE(g || (dpool && ss_out), WW_E_INVALID, "ww_pwconv1x1_bwd: need g, or dpool + ss_out for the last layer");
    WW_REQUIRE(B >= 1 && H >= 1 && W >= 1, WW_E_INVALID, "ww_pwconv1x1_bwd: bad shape (%d,%d,%d)", B, H, W);
    WW_REQUIRE((long)B * H * W < (1L << 31), WW_E_UNSUPPORTED, "ww_pwconv1x1_bwd: B*H*W = %ld pixels exceed 2^31",
               (long)B * H * W);
    int rc = check_act_b("ww_pwconv1x1_bwd", act_dtype);
    if (rc) return rc;
    const long M = (long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    float *stat = (float *)scratch, *dwp = stat + WW_STAT_SLAB_FLOATS;
    int grid = 0;
    rc = act_dtype == WW_ACT_BF16
             ? launch_pw_bwd_bf16<ww_bf16>(ctx, g, dpool, y_out, ss_out, coef, y_in, ss_in, mr_in, w, M, H * W, g_in, stat, dwp,
                                           &grid, st)
         : act_dtype == WW_ACT_F16
             ? launch_pw_bwd_bf16<ww_f16>(ctx, g, dpool, y_out, ss_out, coef, y_in, ss_in, mr_in, w, M, H * W, g_in, stat, dwp,
                                          &grid, st)
             : launch_pw_bwd<float>(ctx, g, dpool, y_out, ss_out, coef, y_in, ss_in, mr_in, w, M, H * W, g_in, stat, dwp,
                                    &grid, st);
    if (rc) return rc;
    WW_LAUNCH_CHECK();
    ww_prof_scope pf_(ctx, WW_K_FINALIZE, st);
    return ww_launch_bwd_finalize(stat, grid, (double)M, gamma_in, mr_in, coef_in, dgamma_in, dbeta_in, dwp, 4096, dw, st);
}

extern "C" int ww_dwconv3x3_bwd(ww_ctx *ctx, int act_dtype, const void *g, const void *y_out, const float *coef,
                                const void *y_in, const float *ss_in, const float *mr_in, const float *gamma_in,
                                const float *w, int B, int H, int W, void *g_in, float *dw, float *coef_in,
                                float *dgamma_in, float *dbeta_in, void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && g && y_out && coef && y_in && ss_in && mr_in && gamma_in && w && g_in && dw && coef_in &&
                   dgamma_in && dbeta_in && scratch,
               WW_E_INVALID, "ww_dwconv3x3_bwd: null argument");
    WW_REQUIRE(B >= 1 && H >= 1 && W >= 1, WW_E_INVALID, "ww_dwconv3x3_bwd: bad shape (%d,%d,%d)", B, H, W);
    WW_REQUIRE((long)H * W * 64 < (1L << 31), WW_E_UNSUPPORTED, "ww_dwconv3x3_bwd: one image of %d x %d x 64 exceeds 2^31 elements", H, W);
    int rc = check_act_b("ww_dwconv3x3_bwd", act_dtype);
    if (rc) return rc;
    DwGeom gm;
    gm.B = B; gm.H = H; gm.W = W;
    gm.ncs = (W + 3) / 4;
    gm.nseg = (H + DW_HS - 1) / DW_HS;
    gm.hs_len = (H + gm.nseg - 1) / gm.nseg;   // <= DW_HS
    gm.items = (long)B * gm.nseg * gm.ncs;
    hipStream_t st = (hipStream_t)stream;
    float *stat = (float *)scratch, *dwp = stat + WW_STAT_SLAB_FLOATS;
    int grid = 0;
    rc = act_dtype == WW_ACT_BF16
             ? launch_dw_bwd<ww_bf16>(ctx, g, y_out, coef, y_in, ss_in, mr_in, w, gm, g_in, stat, dwp, &grid, st)
         : act_dtype == WW_ACT_F16
             ? launch_dw_bwd<ww_f16>(ctx, g, y_out, coef, y_in, ss_in, mr_in, w, gm, g_in, stat, dwp, &grid, st)
             : launch_dw_bwd<float>(ctx, g, y_out, coef, y_in, ss_in, mr_in, w, gm, g_in, stat, dwp, &grid, st);
    if (rc) return rc;
    WW_LAUNCH_CHECK();
    ww_prof_scope pf_(ctx, WW_K_FINALIZE, st);
    return ww_launch_bwd_finalize(stat, grid, (double)B * H * W, gamma_in, mr_in, coef_in, dgamma_in, dbeta_in, dwp, 576,
                                  dw, st);
}

// w != NULL: recompute y_out from x (see k_stem_bwd); the whole-model backward (ww_model.hip) comes in here with the weights
int ww_stem_bwd_impl(ww_ctx *ctx, int act_dtype, const void *g, const void *y_out, const float *w, const float *coef,
                     const float *x, int B, int Hin, int Win, float *dw, void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && g && y_out && coef && x && dw && scratch, WW_E_INVALID, "ww_conv_stem_bwd: null argument");
    WW_REQUIRE(B >= 1 && Hin >= 1 && Win >= 1, WW_E_INVALID, "ww_conv_stem_bwd: bad shape (%d,%d,%d)", B, Hin, Win);
    int rc = check_act_b("ww_conv_stem_bwd", act_dtype);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float *dwp = (float *)scratch + WW_STAT_SLAB_FLOATS;
    int grid = 0;
    rc = act_dtype == WW_ACT_BF16  ? launch_stem_bwd<ww_bf16>(ctx, g, y_out, w, coef, x, B, Hin, Win, dwp, &grid, st)
         : act_dtype == WW_ACT_F16 ? launch_stem_bwd<ww_f16>(ctx, g, y_out, w, coef, x, B, Hin, Win, dwp, &grid, st)
                                   : launch_stem_bwd<float>(ctx, g, y_out, w, coef, x, B, Hin, Win, dwp, &grid, st);
    if (rc) return rc;
    WW_LAUNCH_CHECK();
    ww_prof_scope pf_(ctx, WW_K_FINALIZE, st);
    return ww_launch_colsum(dwp, grid, 576, dw, st);
}

extern "C" int ww_conv_stem_bwd(ww_ctx *ctx, int act_dtype, const void *g, const void *y_out, const float *coef,
                                const float *x, int B, int Hin, int Win, float *dw, void *scratch,
                                ww_stream_t stream) {
    return ww_stem_bwd_impl(ctx, act_dtype, g, y_out, nullptr, coef, x, B, Hin, Win, dw, scratch, stream);
}
