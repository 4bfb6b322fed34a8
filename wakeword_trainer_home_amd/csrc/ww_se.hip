// Squeeze-excitation block of the MobileNetV3-small body (torchvision's SqueezeExcitation as
// mobilenet_v3_small instantiates it -- src/models/architectures.py:91-102 builds that model):
//     s = mean_hw(x)   h = relu(W1 s + b1)   g = hardsigmoid(W2 h + b2)   y = x * g[b][c]
// At the per-GPU batch of BASELINE config 3 (256) the unfused form was 15 launches per block and step (pool, two matrix-core
// GEMMs with their pre-activation / bias / split-K helpers, scale; the mirror image backwards), every one of them shorter
// than its own dispatch: 135 of the step's 510 launches for ~0.1 % of its arithmetic.  Here the block is THREE launches:
//   k_se_fwd    a 1024-thread workgroup owns IMG whole images: pooled sums (fixed order, fp64), both FCs (fp32 FMA: the
//               FCs are B x C x C/4 -- too small for a matrix-core tile to pay), gate, scaling;
//   k_se_bwd    the same ownership backwards: dgate = sum_hw dy*x, through hardsigmoid' / W2^T / relu' / W1^T, and
//               dx = dy*g + dpool/HW in one pass;
//   k_se_wgrad  dW1, db1, dW2, db2 = fixed-order sums over the batch of the per-image pre-activation gradients.
// A workgroup streams both weight matrices from L2 once per IMG images, and every product is laid out so that a wave's loads
// are contiguous float4s with MANY in flight (the first form -- 256 threads, one weight row per step -- paid one L2 round
// trip per row and was 4x slower than the 15 launches it replaced): row dots for the forward (a group of 4..32 lanes per
// weight row, up to 10 float4s per lane, xor-butterfly inside the group), row-lane column sums for the transposed products
// of the backward.  Everything is a fixed-order sum: bit-reproducible, so a replayed HIP graph equals the eager step.
#include "ww_internal.h"
#include <algorithm>

namespace {

constexpr int SE_T = 1024;       // threads per workgroup (16 waves)
constexpr int SE_MAXC = 1024;    // channel bounds (LDS vectors, thread maps)
constexpr int SE_MAXCS = 256;
constexpr int SE_Q = 10;         // float4s of a weight row per lane (row dots)

__device__ __forceinline__ float hsig(float z) { return fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f); }
__device__ __forceinline__ float hsig_grad(float z) { return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f; }
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b, float acc) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}

// lanes per weight row: the smallest power of two (>= 8 where the row has 8 float4s: one 128-byte line per group and load)
// that covers the row's K4 float4s with at most SE_Q per lane; K4 <= 256
__host__ __device__ inline int se_group(int K4) {
    int gs = K4 >= 8 ? 8 : 4;
    while (gs * SE_Q < K4) gs <<= 1;
    return gs;
}

// out[i][j] = bias[j] + sum_k W[j*K + k] * v[i][k]   (K % 4 == 0; rows of W contiguous along the reduction).
// 1024/GS rows per round, all of a lane's float4s of its row loaded before the first use.  v: LDS [IMG][K], out: LDS [IMG][N].
template <int IMG>
__device__ __forceinline__ void se_rowdot(const float *__restrict__ W, const float *__restrict__ bias, int N, int K,
                                          const float *v, float *out, int tid) {
    const int K4 = K >> 2, GS = se_group(K4);                     // uniform
    const int grp = tid / GS, gl = tid % GS, ngrp = SE_T / GS;
    for (int j0 = 0; j0 < N; j0 += ngrp) {                        // uniform trip count: whole waves reach the shuffles
        const int j = j0 + grp, jc = min(j, N - 1);
        const float bj = bias[jc];                                // with the weights, not behind the butterfly (a round trip of its own)
        float4 wv[SE_Q];                                          // unconditional (clamped) loads: a guarded load compiles to a
#pragma unroll                                                    // branch and a wait each, i.e. one L2 round trip per float4
        for (int q = 0; q < SE_Q; ++q)
            wv[q] = *reinterpret_cast<const float4 *>(W + (size_t)jc * K + 4 * min(q * GS + gl, K4 - 1));
        float acc[IMG];
#pragma unroll
        for (int i = 0; i < IMG; ++i) acc[i] = 0.f;
#pragma unroll
        for (int q = 0; q < SE_Q; ++q) {
            const int k4 = q * GS + gl;
            if (k4 < K4)
#pragma unroll
                for (int i = 0; i < IMG; ++i) acc[i] = dot4(wv[q], *reinterpret_cast<const float4 *>(v + i * K + 4 * k4), acc[i]);
        }
        for (int m = GS >> 1; m >= 1; m >>= 1)
#pragma unroll
            for (int i = 0; i < IMG; ++i) acc[i] += __shfl_xor(acc[i], m);
        if (gl == 0 && j < N) {
#pragma unroll
            for (int i = 0; i < IMG; ++i) out[i * N + j] = acc[i] + bj;
        }
    }
}

// out[i][col] = sum_row Mx[row*ncols + col] * v[i][row]   (ncols % 4 == 0, ncols <= 1024: the transposed product).
// Thread = (float4 column, row lane): coalesced along the columns, L = 1024 / (ncols/4) row lanes (at most 64) each walking
// rows lane, lane+L, ... with independent loads; the lanes of a column are summed through LDS in fixed order, one image at
// a time.  v: LDS [IMG][nrows], out: LDS [IMG][ncols], red: LDS [1024] float4.
template <int IMG>
__device__ __forceinline__ void se_wcolsum(const float *__restrict__ Mx, int nrows, int ncols, const float *v, float *out,
                                           float4 *red, int tid) {
    const int N4 = ncols >> 2;
    int L = SE_T / N4;
    if (L > 64) L = 64;
    const int c4 = tid % N4, lane = tid / N4;
    const bool live = lane < L;
    float4 acc[IMG];
#pragma unroll
    for (int i = 0; i < IMG; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live)
        for (int r0 = lane; r0 < nrows; r0 += 8 * L) {           // eight (clamped, unconditional) row loads in flight
            float4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                w[u] = *reinterpret_cast<const float4 *>(Mx + (size_t)min(r0 + u * L, nrows - 1) * ncols + 4 * c4);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * L;
                if (r < nrows)
#pragma unroll
                    for (int i = 0; i < IMG; ++i) {
                        const float sc = v[i * nrows + r];
                        acc[i].x = fmaf(w[u].x, sc, acc[i].x); acc[i].y = fmaf(w[u].y, sc, acc[i].y);
                        acc[i].z = fmaf(w[u].z, sc, acc[i].z); acc[i].w = fmaf(w[u].w, sc, acc[i].w);
                    }
            }
        }
#pragma unroll
    for (int i = 0; i < IMG; ++i) {
        if (live) red[lane * N4 + c4] = acc[i];
        __syncthreads();
        if (lane == 0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < L; ++q) {
                const float4 u = red[q * N4 + c4];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *reinterpret_cast<float4 *>(out + i * ncols + 4 * c4) = t;
        }
        __syncthreads();
    }
}

// per-(image, channel) sums over HW of x (dy == nullptr) or dy*x: thread = (image, float4 of channels, row lane), fp64
// accumulation, the row lanes summed through LDS in fixed order.  out: LDS [IMG][C] = float(sum * mul); redd: LDS [1024][4]
// doubles.  IMG * C/4 <= 1024.
template <int IMG>
__device__ __forceinline__ void se_hwsum(const float *__restrict__ x, const float *__restrict__ dy, int nimg, int HW, int C,
                                         double mul, float *out, double *redd, int tid) {
    const int C4 = C >> 2, npairs = IMG * C4;
    int L = SE_T / npairs;
    if (L > 64) L = 64;
    const int p = tid % npairs, lane = tid / npairs, img = p / C4, c4 = p - img * C4;
    const bool live = lane < L;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (live && img < nimg) {
        const size_t o = (size_t)img * HW * C + 4 * c4;
        if (dy) {
            for (int h0 = lane; h0 < HW; h0 += 5 * L) {          // batches of (clamped, unconditional) loads
                float4 xv[5], dv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const size_t a = o + (size_t)min(h0 + u * L, HW - 1) * C;
                    xv[u] = *reinterpret_cast<const float4 *>(x + a);
                    dv[u] = *reinterpret_cast<const float4 *>(dy + a);
                }
#pragma unroll
                for (int u = 0; u < 5; ++u)
                    if (h0 + u * L < HW) {
                        a0 += (double)dv[u].x * (double)xv[u].x; a1 += (double)dv[u].y * (double)xv[u].y;
                        a2 += (double)dv[u].z * (double)xv[u].z; a3 += (double)dv[u].w * (double)xv[u].w;
                    }
            }
        } else {
            for (int h0 = lane; h0 < HW; h0 += 10 * L) {
                float4 xv[10];
#pragma unroll
                for (int u = 0; u < 10; ++u) xv[u] = *reinterpret_cast<const float4 *>(x + o + (size_t)min(h0 + u * L, HW - 1) * C);
#pragma unroll
                for (int u = 0; u < 10; ++u)
                    if (h0 + u * L < HW) { a0 += (double)xv[u].x; a1 += (double)xv[u].y; a2 += (double)xv[u].z; a3 += (double)xv[u].w; }
            }
        }
    }
    if (live) {
        double *r = redd + (size_t)(lane * npairs + p) * 4;
        r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
    }
    __syncthreads();
    if (lane == 0) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int q = 0; q < L; ++q) {
            const double *r = redd + (size_t)(q * npairs + p) * 4;
            t0 += r[0]; t1 += r[1]; t2 += r[2]; t3 += r[3];
        }
        *reinterpret_cast<float4 *>(out + img * C + 4 * c4) = make_float4((float)(t0 * mul), (float)(t1 * mul), (float)(t2 * mul),
                                                                           (float)(t3 * mul));
    }
    __syncthreads();
}

// grid ceil(B / IMG).  LDS: redd [1024][4] doubles (32 KB) | sv [IMG][C] (pooled mean) | gv [IMG][C] (gate) | hv [IMG][Cs]
template <int IMG>
__global__ __launch_bounds__(SE_T) void k_se_fwd(const float *__restrict__ x, int B, int HW, int C, int Cs,
                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                 const float *__restrict__ w2, const float *__restrict__ b2,
                                                 float *__restrict__ y, float *__restrict__ s_out, float *__restrict__ pre1_out,
                                                 float *__restrict__ pre2_out) {
    extern __shared__ __align__(16) unsigned char se_lds[];
    double *redd = reinterpret_cast<double *>(se_lds);
    float *sv = reinterpret_cast<float *>(se_lds + (size_t)SE_T * 4 * sizeof(double));
    float *gv = sv + IMG * C;
    float *hv = gv + IMG * C;
    const int tid = threadIdx.x, b0 = blockIdx.x * IMG, nimg = min(IMG, B - b0);
    const float *xb = x + (size_t)b0 * HW * C;
    se_hwsum<IMG>(xb, nullptr, nimg, HW, C, 1.0 / (double)HW, sv, redd, tid);
    for (int p = tid; p < nimg * C; p += SE_T) s_out[(size_t)b0 * C + p] = sv[p];
    se_rowdot<IMG>(w1, b1, Cs, C, sv, hv, tid);
    __syncthreads();
    for (int p = tid; p < IMG * Cs; p += SE_T) {
        const float pre = hv[p];
        if (p < nimg * Cs) pre1_out[(size_t)b0 * Cs + p] = pre;
        hv[p] = pre < 0.f ? 0.f : pre;
    }
    __syncthreads();
    se_rowdot<IMG>(w2, b2, C, Cs, hv, gv, tid);
    __syncthreads();
    for (int p = tid; p < IMG * C; p += SE_T) {
        const float pre = gv[p];
        if (p < nimg * C) pre2_out[(size_t)b0 * C + p] = pre;
        gv[p] = hsig(pre);
    }
    __syncthreads();
    const uint32_t C4 = (uint32_t)C >> 2, per = (uint32_t)HW * C4, n4 = (uint32_t)nimg * per;
    const float4 *x4 = reinterpret_cast<const float4 *>(xb);
    float4 *y4 = reinterpret_cast<float4 *>(y + (size_t)b0 * HW * C);
    for (uint32_t i0 = tid; i0 < n4; i0 += 6 * SE_T) {
        float4 v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] = x4[min(i0 + u * SE_T, n4 - 1)];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const uint32_t i = i0 + u * SE_T;
            if (i < n4) {
                const uint32_t img = i / per, c = (i % C4) * 4;
                const float4 g = *reinterpret_cast<const float4 *>(gv + img * C + c);
                y4[i] = make_float4(v[u].x * g.x, v[u].y * g.y, v[u].z * g.z, v[u].w * g.w);
            }
        }
    }
}

// LDS: redd [1024][4] doubles (32 KB; its first 16 KB double as the float4 lane partials of the column sums) |
//      dgv [IMG][C] (dgate -> dpre2) | dsv [IMG][C] (dpool) | gv [IMG][C] (gate) | hv [IMG][Cs] (dh -> dpre1)
template <int IMG>
__global__ __launch_bounds__(SE_T) void k_se_bwd(const float *__restrict__ x, const float *__restrict__ dy, int B, int HW, int C,
                                                 int Cs, const float *__restrict__ w1, const float *__restrict__ w2,
                                                 const float *__restrict__ pre1, const float *__restrict__ pre2,
                                                 float *__restrict__ dx, float *__restrict__ dpre1_out,
                                                 float *__restrict__ dpre2_out) {
    extern __shared__ __align__(16) unsigned char se_lds[];
    double *redd = reinterpret_cast<double *>(se_lds);
    float4 *red = reinterpret_cast<float4 *>(se_lds);
    float *dgv = reinterpret_cast<float *>(se_lds + (size_t)SE_T * 4 * sizeof(double));
    float *dsv = dgv + IMG * C;
    float *gv = dsv + IMG * C;
    float *hv = gv + IMG * C;
    const int tid = threadIdx.x, b0 = blockIdx.x * IMG, nimg = min(IMG, B - b0);
    const float *xb = x + (size_t)b0 * HW * C, *dyb = dy + (size_t)b0 * HW * C;
    // the saved pre-activations this thread will need are fetched BEFORE the pooled sums (IMG * C <= 4096, IMG * Cs <= 1024):
    // behind them each was one more dependent round trip of the workgroup's chain
    float z2[4], z1 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) z2[u] = pre2[(size_t)b0 * C + min(tid + u * SE_T, nimg * C - 1)];
    if (tid < nimg * Cs) z1 = pre1[(size_t)b0 * Cs + tid];
    se_hwsum<IMG>(xb, dyb, nimg, HW, C, 1.0, dgv, redd, tid);
#pragma unroll
    for (int u = 0; u < 4; ++u) {                               // dpre2 = dgate * hardsigmoid'(pre2); the gate itself for dx
        const int p = tid + u * SE_T;
        if (p >= IMG * C) break;
        float d = 0.f, g = 0.f;
        if (p < nimg * C) {
            const float z = z2[u];
            d = dgv[p] * hsig_grad(z);
            g = hsig(z);
            dpre2_out[(size_t)b0 * C + p] = d;
        }
        dgv[p] = d;
        gv[p] = g;
    }
    __syncthreads();
    se_wcolsum<IMG>(w2, C, Cs, dgv, hv, red, tid);              // dh[j] = sum_c W2[c][j] dpre2[c]
    if (tid < IMG * Cs) {                                       // dpre1 = dh * relu'(pre1)   (IMG * Cs <= 1024 = one per thread)
        float d = 0.f;
        if (tid < nimg * Cs) {
            d = z1 > 0.f ? hv[tid] : 0.f;
            dpre1_out[(size_t)b0 * Cs + tid] = d;
        }
        hv[tid] = d;
    }
    __syncthreads();
    se_wcolsum<IMG>(w1, Cs, C, hv, dsv, red, tid);              // dpool[c] = sum_j W1[j][c] dpre1[j]
    const uint32_t C4 = (uint32_t)C >> 2, per = (uint32_t)HW * C4, n4 = (uint32_t)nimg * per;
    const float inv = 1.0f / (float)HW;
    const float4 *d4 = reinterpret_cast<const float4 *>(dyb);
    float4 *o4 = reinterpret_cast<float4 *>(dx + (size_t)b0 * HW * C);
    for (uint32_t i0 = tid; i0 < n4; i0 += 6 * SE_T) {
        float4 d[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) d[u] = d4[min(i0 + u * SE_T, n4 - 1)];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const uint32_t i = i0 + u * SE_T;
            if (i < n4) {
                const uint32_t img = i / per, c = (i % C4) * 4;
                const float4 g = *reinterpret_cast<const float4 *>(gv + img * C + c);
                const float4 dp = *reinterpret_cast<const float4 *>(dsv + img * C + c);
                o4[i] = make_float4(fmaf(dp.x, inv, d[u].x * g.x), fmaf(dp.y, inv, d[u].y * g.y), fmaf(dp.z, inv, d[u].z * g.z),
                                    fmaf(dp.w, inv, d[u].w * g.w));
            }
        }
    }
}

// Parameter gradients as register-tiled outer products over the batch: a thread owns a 4 x 4 tile of dW1 (j..j+3, c..c+3) or
// dW2 (c..c+3, j..j+3) and one of 32 batch lanes: per image two float4 loads feed 16 FMAs, and a lane's images are loaded eight
// at a time (unconditional, clamped), so B = 256 is ONE batch of loads per thread -- the kernel is a latency chain, not work: it took
// 14-18 us at every size with four dependent batches.  8 tiles x 32 lanes per workgroup, tiles ordered along the operand that is
// contiguous in memory, the lanes summed through LDS in fixed order.  The bias gradients ride on the tiles of the first
// column block (c == 0 for dW1: db1 = sum_b dpre1; j == 0 for dW2: db2).
//   dW1[j][c] = sum_b dpre1[b][j] * s[b][c]            dW2[c][j] = sum_b dpre2[b][c] * relu(pre1[b][j])
__global__ __launch_bounds__(256) void k_se_wgrad(const float *__restrict__ dpre1, const float *__restrict__ dpre2,
                                                  const float *__restrict__ s, const float *__restrict__ pre1, int B, int C,
                                                  int Cs, float *__restrict__ dw1, float *__restrict__ db1,
                                                  float *__restrict__ dw2, float *__restrict__ db2) {
    __shared__ __align__(16) float red[32][8][20];
    const int C4 = C >> 2, Cs4 = Cs >> 2, ntile = C4 * Cs4;
    const int tl = threadIdx.x & 7, lane = threadIdx.x >> 3;
    int t = blockIdx.x * 8 + tl;
    const bool second = t >= ntile;                 // tiles [0, ntile): dW1, [ntile, 2 ntile): dW2
    if (second) t -= ntile;
    const bool live = t < ntile;
    // dW1: rows j (operand a = dpre1, stride Cs), columns c (operand b = s, stride C, contiguous across tiles)
    // dW2: rows c (a = dpre2, stride C),          columns j (b = pre1 through ReLU, stride Cs, contiguous across tiles)
    const int ncol4 = second ? Cs4 : C4;
    const int r4 = live ? t / ncol4 : 0, q4 = live ? t - r4 * ncol4 : 0;
    const float *pa = (second ? dpre2 : dpre1) + 4 * r4, *pb = (second ? pre1 : s) + 4 * q4;
    const int lda = second ? C : Cs, ldb = second ? Cs : C;
    float acc[4][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
    if (live)
        for (int b0 = lane; b0 < B; b0 += 8 * 32) {
            float4 a[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = min(b0 + 32 * u, B - 1);
                a[u] = *reinterpret_cast<const float4 *>(pa + (size_t)b * lda);
                v[u] = *reinterpret_cast<const float4 *>(pb + (size_t)b * ldb);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + 32 * u < B) {
                    if (second) { v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f); }
                    const float av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        accb[i] += av[i];
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[i][k] = fmaf(av[i], bv[k], acc[i][k]);
                    }
                }
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float4 *>(&red[lane][tl][4 * i]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        red[lane][tl][16 + i] = accb[i];
    }
    __syncthreads();
    // 8 tiles x 20 values, 32 lanes each: thread -> (tile, value), fixed order over the lanes
    if (threadIdx.x < 8 * 20) {
        const int tt = threadIdx.x / 20, v = threadIdx.x - tt * 20;
        int g = blockIdx.x * 8 + tt;
        const bool sec = g >= ntile;
        if (sec) g -= ntile;
        if (g < ntile) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) sum += red[q][tt][v];
            const int nc4 = sec ? Cs4 : C4, rr = g / nc4, qq = g - rr * nc4;
            if (v < 16) {
                const int i = v >> 2, k = v & 3;
                float *dst = sec ? dw2 : dw1;
                dst[(size_t)(4 * rr + i) * (sec ? Cs : C) + 4 * qq + k] = sum;
            } else if (qq == 0) {
                (sec ? db2 : db1)[4 * rr + (v - 16)] = sum;
            }
        }
    }
}

inline size_t se_fwd_lds(int img, int C, int Cs) { return (size_t)SE_T * 4 * sizeof(double) + (size_t)img * (2 * C + Cs) * sizeof(float); }
inline size_t se_bwd_lds(int img, int C, int Cs) { return (size_t)SE_T * 4 * sizeof(double) + (size_t)img * (3 * C + Cs) * sizeof(float); }
// images per workgroup: a workgroup streams both weight matrices once, so more images amortise that; bounded by the 1024
// (image, float4-channel) pairs of the pooled sums' thread map and by 64 KB of LDS
inline int se_img(int B, int C, int Cs) {
    int img = B >= 128 ? 4 : (B >= 32 ? 2 : 1);
    while (img > 1 && (img * (C / 4) > SE_T || se_bwd_lds(img, C, Cs) > 64 * 1024)) img >>= 1;
    return img;
}
int se_check(const char *who, int B, int HW, int C, int Cs) {
    WW_REQUIRE(B >= 1 && HW >= 1 && C >= 4 && Cs >= 4, WW_E_INVALID, "%s: bad shape (B=%d, HW=%d, C=%d, Cs=%d)", who, B, HW, C, Cs);
    WW_REQUIRE((C & 3) == 0 && (Cs & 3) == 0 && C <= SE_MAXC && Cs <= SE_MAXCS, WW_E_UNSUPPORTED,
               "%s: needs C %% 4 == 0, Cs %% 4 == 0, C <= %d, Cs <= %d (got C=%d, Cs=%d)", who, SE_MAXC, SE_MAXCS, C, Cs);
    WW_REQUIRE((long)HW * C < (1L << 28), WW_E_UNSUPPORTED, "%s: image too large for 32-bit indices", who);
    return WW_OK;
}

}  // namespace

extern "C" size_t ww_se_bwd_scratch_bytes(int B, int C, int Cs) {
    return (size_t)std::max(B, 1) * (size_t)(std::max(C, 0) + std::max(Cs, 0)) * sizeof(float);
}

extern "C" int ww_se_fwd(ww_ctx *ctx, const float *x, int B, int HW, int C, int Cs, const float *w1, const float *b1,
                         const float *w2, const float *b2, float *y, float *s, float *pre1, float *pre2, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w1 && b1 && w2 && b2 && y && s && pre1 && pre2, WW_E_INVALID, "ww_se_fwd: null argument");
    int rc = se_check("ww_se_fwd", B, HW, C, Cs);
    if (rc) return rc;
    WW_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0, WW_E_INVALID,
               "ww_se_fwd: x / y / w1 / w2 must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    const int img = se_img(B, C, Cs), grid = (B + img - 1) / img;
    const size_t lds = se_fwd_lds(img, C, Cs);
#define WW_SE_FWD(I) hipLaunchKernelGGL(k_se_fwd<I>, dim3(grid), dim3(SE_T), lds, st, x, B, HW, C, Cs, w1, b1, w2, b2, y, s, pre1, pre2)
    if (img == 4) WW_SE_FWD(4); else if (img == 2) WW_SE_FWD(2); else WW_SE_FWD(1);
#undef WW_SE_FWD
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" int ww_se_bwd(ww_ctx *ctx, const float *x, const float *dy, const float *s, const float *pre1, const float *pre2,
                         const float *w1, const float *w2, int B, int HW, int C, int Cs, float *dx, float *dw1, float *db1,
                         float *dw2, float *db2, void *scratch, size_t scratch_bytes, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && dy && s && pre1 && pre2 && w1 && w2 && dx && dw1 && db1 && dw2 && db2 && scratch, WW_E_INVALID,
               "ww_se_bwd: null argument");
    int rc = se_check("ww_se_bwd", B, HW, C, Cs);
    if (rc) return rc;
    WW_REQUIRE(scratch_bytes >= ww_se_bwd_scratch_bytes(B, C, Cs), WW_E_INVALID, "ww_se_bwd: scratch too small (%zu < %zu)",
               scratch_bytes, ww_se_bwd_scratch_bytes(B, C, Cs));
    WW_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0, WW_E_INVALID,
               "ww_se_bwd: x / dy / dx / w1 / w2 must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    float *dpre1 = (float *)scratch, *dpre2 = dpre1 + (size_t)B * Cs;
    const int img = se_img(B, C, Cs), grid = (B + img - 1) / img;
    const size_t lds = se_bwd_lds(img, C, Cs);
#define WW_SE_BWD(I) hipLaunchKernelGGL(k_se_bwd<I>, dim3(grid), dim3(SE_T), lds, st, x, dy, B, HW, C, Cs, w1, w2, pre1, pre2, dx, dpre1, dpre2)
    if (img == 4) WW_SE_BWD(4); else if (img == 2) WW_SE_BWD(2); else WW_SE_BWD(1);
#undef WW_SE_BWD
    WW_LAUNCH_CHECK();
    const int ntile2 = 2 * (C / 4) * (Cs / 4);
    hipLaunchKernelGGL(k_se_wgrad, dim3((ntile2 + 7) / 8), dim3(256), 0, st, dpre1, dpre2, s, pre1, B, C, Cs, dw1, db1, dw2, db2);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
