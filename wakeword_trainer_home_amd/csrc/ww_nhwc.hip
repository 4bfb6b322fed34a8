// Generic channels-last (NHWC, fp32) layers for bodies whose channel counts vary layer to layer -- the MobileNetV3-small body
// of BASELINE config 3 (torchvision's mobilenet_v3_small as the reference instantiates it,
// src/models/architectures.py:91-102).  Unlike the 64-channel cnn_small kernels (BatchNorm folded into producers and
// consumers) these are plain, unfused building blocks: an activation tensor is a row-major (M = B*H*W, C) matrix, so
//   * every 1x1 convolution and every squeeze-excitation FC is ww_gemm / ww_linear_mfma_* on the matrix cores;
//   * BatchNorm(+activation) is a statistics pass (chunked column sums, fixed-order fp64 finish) and an apply pass;
//   * depthwise k x k (3 or 5, stride 1 or 2) is a per-(pixel, 4-channel) gather;
//   * squeeze-excitation is pool -> FC+ReLU -> FC+Hardsigmoid -> scale.
// First version: correctness and the reference's layer semantics; fusion of these passes is the next step.
#include "ww_internal.h"
#include "ww_layers.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float act_fwd(int act, float z) {
    if (act == WW_LIN_RELU) return z < 0.f ? 0.f : z;
    if (act == WW_LIN_HARDSWISH) return z * fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
    if (act == WW_LIN_HARDSIGMOID) return fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
    return z;
}
__device__ __forceinline__ float act_grad(int act, float z) {
    if (act == WW_LIN_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == WW_LIN_HARDSWISH) return z < -3.f ? 0.f : (z <= 3.f ? z * (1.f / 3.f) + 0.5f : 1.f);
    if (act == WW_LIN_HARDSIGMOID) return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f;
    return 1.f;
}

constexpr int NCHUNK = 256;     // max row chunks of the column reductions (one workgroup each)
constexpr int NCHUNK_DW = 1024; // max pixel chunks of the depthwise weight-gradient kernel

// Column reductions over a tall (M, C) matrix with ANY C <= 1024: a block has C*R threads, thread t owns column t % C and rows
// r0 + t / C (+R, +2R, ...), so consecutive threads read consecutive addresses whatever C is (a 64-column tiling would
// leave 3/4 of the lanes idle at C = 16); one block per row chunk, partials finished in fixed order afterwards.
// partial[chunk][0:C] = sum x, [C:2C] = sum x^2 over the chunk's rows.  grid = chunks, block = C*R, smem = 2*R*C doubles
__global__ __launch_bounds__(1024) void k_colstats(const float *__restrict__ x, long M, int C, int R, long rows_per_chunk,
                                                   float *__restrict__ part) {
    extern __shared__ double red[];
    const int c = threadIdx.x % C, r = threadIdx.x / C;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    double s = 0.0, q = 0.0;
    // a thread's rows in batches of eight (clamped, unconditional) loads: at these sizes a thread has 4-16 rows, so a guarded
    // loop was one L2 round trip per row
    for (long rb = r0 + r; rb < r1; rb += 8L * R) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[min(rb + (long)u * R, r1 - 1) * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rb + (long)u * R < r1) { const double d = v[u]; s += d; q += d * d; }
    }
    red[r * C + c] = s; red[(R + r) * C + c] = q;
    __syncthreads();
    if (r == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < R; ++i) { a += red[i * C + c]; b += red[(R + i) * C + c]; }
        part[(long)blockIdx.x * 2 * C + c] = (float)a;
        part[(long)blockIdx.x * 2 * C + C + c] = (float)b;
    }
}
// fixed-order sum of the chunk partials of 64 columns by 16 row lanes: tot0/tot1 valid in the threads with p == 0.  A lane's
// <= 16 partial rows (chunks <= NCHUNK = 256) are loaded in ONE batch before the first add: as a load -> add loop the finish
// kernels were a chain of 16 dependent L2 round trips (6-7 us between two 5 us kernels); the order of the adds is unchanged.
__device__ __forceinline__ void chunk_sums(const float *__restrict__ part, int chunks, int C, int col, int p, int c,
                                           double (*sh)[16][64], double &tot0, double &tot1) {
    double s = 0.0, q = 0.0;
    if (col < C)
        for (int base = 0; base < chunks; base += NCHUNK) {      // one batch for the layer library's own <= 256 chunks
            float sv[NCHUNK / 16], qv[NCHUNK / 16];
#pragma unroll
            for (int u = 0; u < NCHUNK / 16; ++u) {
                const int i = min(base + p + 16 * u, chunks - 1);    // unconditional loads (a guarded load is a branch + wait each)
                sv[u] = part[(long)i * 2 * C + col];
                qv[u] = part[(long)i * 2 * C + C + col];
            }
#pragma unroll
            for (int u = 0; u < NCHUNK / 16; ++u)
                if (base + p + 16 * u < chunks) { s += sv[u]; q += qv[u]; }
        }
    sh[0][p][c] = s; sh[1][p][c] = q;
    __syncthreads();
    tot0 = tot1 = 0.0;
    if (p == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { tot0 += sh[0][i][c]; tot1 += sh[1][i][c]; }
    }
}
// BatchNorm2d statistics -> scale|shift (ss) and mean|rstd (mr); torch semantics for the running statistics.
// grid ceil(C/64), block 1024 (64 columns x 16 chunk lanes)
__global__ __launch_bounds__(1024) void k_bn_finish(const float *__restrict__ part, int chunks, long M, int C, ww_bn_t bn,
                                                    float *__restrict__ ss, float *__restrict__ mr) {
    __shared__ double sh[2][16][64];
    const int cl = threadIdx.x & 63, p = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    // the per-channel parameters are fetched BEFORE the partial sums (they used to be two more dependent round trips behind them)
    float g_c = 0.f, b_c = 0.f, rm_c = 0.f, rv_c = 1.f;
    if (p == 0 && c < C) {
        g_c = bn.gamma[c]; b_c = bn.beta[c];
        if (bn.running_mean) { rm_c = bn.running_mean[c]; rv_c = bn.running_var[c]; }
    }
    double s = 0.0, q = 0.0;
    if (bn.training) chunk_sums(part, chunks, C, c, p, cl, sh, s, q);
    if (p != 0 || c >= C) return;
    double mean, var;
    if (bn.training) {
        mean = s / (double)M;
        var = q / (double)M - mean * mean;
        if (var < 0.0) var = 0.0;
        if (bn.running_mean) {
            const double m = bn.momentum, unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            bn.running_mean[c] = (float)((1.0 - m) * (double)rm_c + m * mean);
            bn.running_var[c] = (float)((1.0 - m) * (double)rv_c + m * unb);
        }
    } else {
        mean = rm_c;
        var = rv_c;
    }
    const double rstd = 1.0 / sqrt(var + (double)bn.eps), scale = (double)g_c * rstd;
    ss[c] = (float)scale;
    ss[C + c] = (float)((double)b_c - mean * scale);
    mr[c] = (float)mean;
    mr[C + c] = (float)rstd;
}
__global__ __launch_bounds__(256) void k_bn_act_apply(const float *__restrict__ x, const float *__restrict__ ss, long n, int C,
                                                      int act, float *__restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        y[i] = act_fwd(act, fmaf(x[i], ss[c], ss[C + c]));
    }
}
// the elementwise passes for C % 4 == 0 (every MobileNetV3 layer): a thread moves float4s, the channel index comes from a 32-bit
// remainder per FOUR elements (the scalar forms pay a 64-bit one per element); same arithmetic per element, same bits
__global__ __launch_bounds__(256) void k_bn_act_apply4(const float4 *__restrict__ x, const float *__restrict__ ss, uint32_t n4,
                                                       int C, int act, float4 *__restrict__ y) {
    const uint32_t C4 = (uint32_t)C >> 2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        const int c = (int)(i % C4) * 4;
        const float4 v = x[i], sc = *reinterpret_cast<const float4 *>(ss + c), sf = *reinterpret_cast<const float4 *>(ss + C + c);
        y[i] = make_float4(act_fwd(act, fmaf(v.x, sc.x, sf.x)), act_fwd(act, fmaf(v.y, sc.y, sf.y)),
                           act_fwd(act, fmaf(v.z, sc.z, sf.z)), act_fwd(act, fmaf(v.w, sc.w, sf.w)));
    }
}
__device__ __forceinline__ float bnact_bwd_one(float xv, float dav, float sc, float sf, float mu, float rs, float s1, float s2,
                                               float invM, int act, int training) {
    const float dz = dav * act_grad(act, fmaf(xv, sc, sf));
    if (!training) return sc * dz;
    const float xh = (xv - mu) * rs;
    return sc * (dz - s1 * invM - xh * s2 * invM);
}
__global__ __launch_bounds__(256) void k_bnact_bwd_apply4(const float4 *__restrict__ x, const float4 *__restrict__ da,
                                                          const float *__restrict__ ss, const float *__restrict__ mr,
                                                          const float *__restrict__ sums, long M, uint32_t n4, int C, int act,
                                                          int training, float4 *__restrict__ dx) {
    const uint32_t C4 = (uint32_t)C >> 2;
    const float invM = 1.0f / (float)M;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        const int c = (int)(i % C4) * 4;
        const float4 xv = x[i], dv = da[i];
        const float4 sc = *reinterpret_cast<const float4 *>(ss + c), sf = *reinterpret_cast<const float4 *>(ss + C + c);
        const float4 mu = *reinterpret_cast<const float4 *>(mr + c), rs = *reinterpret_cast<const float4 *>(mr + C + c);
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
        if (training) { s1 = *reinterpret_cast<const float4 *>(sums + c); s2 = *reinterpret_cast<const float4 *>(sums + C + c); }
        dx[i] = make_float4(bnact_bwd_one(xv.x, dv.x, sc.x, sf.x, mu.x, rs.x, s1.x, s2.x, invM, act, training),
                            bnact_bwd_one(xv.y, dv.y, sc.y, sf.y, mu.y, rs.y, s1.y, s2.y, invM, act, training),
                            bnact_bwd_one(xv.z, dv.z, sc.z, sf.z, mu.z, rs.z, s1.z, s2.z, invM, act, training),
                            bnact_bwd_one(xv.w, dv.w, sc.w, sf.w, mu.w, rs.w, s1.w, s2.w, invM, act, training));
    }
}
// backward pass 1: dz = da * act'(z); partial sums of dz and dz*xhat (same thread layout as k_colstats)
__global__ __launch_bounds__(1024) void k_bnact_bwd_stats(const float *__restrict__ x, const float *__restrict__ da,
                                                          const float *__restrict__ ss, const float *__restrict__ mr, long M,
                                                          int C, int R, int act, long rows_per_chunk, float *__restrict__ part) {
    extern __shared__ double red[];
    const int c = threadIdx.x % C, r = threadIdx.x / C;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const float sc = ss[c], sf = ss[C + c], mu = mr[c], rs = mr[C + c];
    double s1 = 0.0, s2 = 0.0;
    for (long rb = r0 + r; rb < r1; rb += 8L * R) {              // batches of eight (clamped, unconditional) row loads
        float xv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long o = min(rb + (long)u * R, r1 - 1) * C + c;
            xv[u] = x[o]; dv[u] = da[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rb + (long)u * R < r1) {
                const float dz = dv[u] * act_grad(act, fmaf(xv[u], sc, sf));
                s1 += dz;
                s2 += (double)dz * (double)((xv[u] - mu) * rs);
            }
    }
    red[r * C + c] = s1; red[(R + r) * C + c] = s2;
    __syncthreads();
    if (r == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < R; ++i) { a += red[i * C + c]; b += red[(R + i) * C + c]; }
        part[(long)blockIdx.x * 2 * C + c] = (float)a;
        part[(long)blockIdx.x * 2 * C + C + c] = (float)b;
    }
}
__global__ __launch_bounds__(1024) void k_bnact_bwd_finish(const float *__restrict__ part, int chunks, int C,
                                                           float *__restrict__ sums, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta) {
    __shared__ double sh[2][16][64];
    const int cl = threadIdx.x & 63, p = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    double s1, s2;
    chunk_sums(part, chunks, C, c, p, cl, sh, s1, s2);
    if (p != 0 || c >= C) return;
    sums[c] = (float)s1; sums[C + c] = (float)s2;
    dgamma[c] = (float)s2; dbeta[c] = (float)s1;
}
// backward pass 2: dx = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat))   (training);  eval: dx = scale*dz
__global__ __launch_bounds__(256) void k_bnact_bwd_apply(const float *__restrict__ x, const float *__restrict__ da,
                                                         const float *__restrict__ ss, const float *__restrict__ mr,
                                                         const float *__restrict__ sums, long M, int C, int act, int training,
                                                         float *__restrict__ dx) {
    const long n = M * C;
    const float invM = 1.0f / (float)M;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float xv = x[i], sc = ss[c];
        const float dz = da[i] * act_grad(act, fmaf(xv, sc, ss[C + c]));
        if (training) {
            const float xh = (xv - mr[c]) * mr[C + c];
            dx[i] = sc * (dz - sums[c] * invM - xh * sums[C + c] * invM);
        } else {
            dx[i] = sc * dz;
        }
    }
}

// ---- BatchNorm apply pass that FINISHES the statistics itself, for layers whose PRODUCER (the 1x1-convolution GEMM's epilogue,
// the LDS depthwise kernel) already left per-tile partial sums: conv -> this = 2 launches instead of conv, statistics, finish,
// apply.  A workgroup owns (row chunk, group of CG4 <= 16 channel float4s) and sums the group's partial columns itself --
// float4 loads in batches of eight, the same fixed order in every workgroup, so all of them hold bit-identical totals.  Used
// when that re-read is <= 24 KB per workgroup (the caller picks CG4); with the layer library's own statistics pass (up to 256
// chunks x 64 channels) it was slower than the three-launch form (profiles/r02_mnv3_experiments).
struct BnTile { int CG4, G_c, R; long rows_per_block; };
// totals of the group's 2*CG partial columns -> tot[2*CG] (LDS, double).  part rows are [2C] floats.  256 threads.
__device__ __forceinline__ void bn_group_totals(const float *__restrict__ part, int chunks, int C, int c0, int CG4, double *tot,
                                                double *redd /* [256][4] */) {
    const int nq = 2 * CG4;                         // float4 columns of the group (both statistics)
    const int LA = 256 / nq;                        // chunk lanes (>= 8)
    const int k4 = threadIdx.x % nq, la = threadIdx.x / nq;
    const int comp = k4 / CG4, cq = k4 - comp * CG4, col = comp * C + min(c0 + 4 * cq, C - 4);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (la < LA)
        for (int q0 = la; q0 < chunks; q0 += 8 * LA) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(part + (size_t)min(q0 + u * LA, chunks - 1) * 2 * C + col);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (q0 + u * LA < chunks) { a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w; }
        }
    double *r = redd + (size_t)threadIdx.x * 4;
    r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
    __syncthreads();
    if (threadIdx.x < nq) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int q = 0; q < LA; ++q) {
            const double *u = redd + (size_t)(q * nq + k4) * 4;
            t0 += u[0]; t1 += u[1]; t2 += u[2]; t3 += u[3];
        }
        double *o = tot + comp * 4 * CG4 + 4 * cq;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
    }
    __syncthreads();
}
// grid G_r * G_c, block 256: thread = (float4 of the group's channels, row lane)
__global__ __launch_bounds__(256) void k_bn_act_apply_fin(const float *__restrict__ x, const float *__restrict__ part, int chunks,
                                                          long M, int C, BnTile t, ww_bn_t bn, int act, float *__restrict__ y,
                                                          float *__restrict__ ss, float *__restrict__ mr,
                                                          const float *__restrict__ res /* nullable: y = act(bn(x)) + res */) {
    __shared__ double redd[256 * 4];
    __shared__ double tot[128];
    __shared__ __align__(16) float scsh[128];
    const int gc = blockIdx.x % t.G_c, c0 = gc * 4 * t.CG4, CG = 4 * t.CG4;
    const long rc = blockIdx.x / t.G_c;
    float g_c = 0.f, b_c = 0.f, rm_c = 0.f, rv_c = 1.f;          // fetched before the partial sums, not behind them
    const bool fin = threadIdx.x < CG && c0 + threadIdx.x < C;
    if (fin) {
        const int c = c0 + threadIdx.x;
        g_c = bn.gamma[c]; b_c = bn.beta[c];
        if (rc == 0 && bn.running_mean) { rm_c = bn.running_mean[c]; rv_c = bn.running_var[c]; }
    }
    bn_group_totals(part, chunks, C, c0, t.CG4, tot, redd);
    if (fin) {
        const int c = c0 + threadIdx.x;
        const double mean = tot[threadIdx.x] / (double)M;
        double var = tot[CG + threadIdx.x] / (double)M - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)bn.eps), scale = (double)g_c * rstd;
        const float scf = (float)scale, shf = (float)((double)b_c - mean * scale);
        scsh[threadIdx.x] = scf; scsh[CG + threadIdx.x] = shf;
        if (rc == 0) {
            ss[c] = scf; ss[C + c] = shf;
            mr[c] = (float)mean; mr[C + c] = (float)rstd;
            if (bn.running_mean) {
                const double m = bn.momentum, unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
                bn.running_mean[c] = (float)((1.0 - m) * (double)rm_c + m * mean);
                bn.running_var[c] = (float)((1.0 - m) * (double)rv_c + m * unb);
            }
        }
    }
    __syncthreads();
    const int cq = threadIdx.x % t.CG4, rl = threadIdx.x / t.CG4;
    if (rl >= t.R || c0 + 4 * cq >= C) return;
    const float4 sc = *reinterpret_cast<const float4 *>(scsh + 4 * cq), sf = *reinterpret_cast<const float4 *>(scsh + CG + 4 * cq);
    const long r0 = rc * t.rows_per_block, r1 = min(M, r0 + t.rows_per_block);
    const size_t col = (size_t)c0 + 4 * cq;
    for (long rb = r0 + rl; rb < r1; rb += 4L * t.R) {
        float4 v[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = (size_t)min(rb + (long)u * t.R, r1 - 1) * C + col;
            v[u] = *reinterpret_cast<const float4 *>(x + o);
            rv[u] = res ? *reinterpret_cast<const float4 *>(res + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (rb + (long)u * t.R < r1)
                *reinterpret_cast<float4 *>(y + (size_t)(rb + (long)u * t.R) * C + col) =
                    make_float4(act_fwd(act, fmaf(v[u].x, sc.x, sf.x)) + rv[u].x, act_fwd(act, fmaf(v[u].y, sc.y, sf.y)) + rv[u].y,
                                act_fwd(act, fmaf(v[u].z, sc.z, sf.z)) + rv[u].z, act_fwd(act, fmaf(v[u].w, sc.w, sf.w)) + rv[u].w);
    }
}

// BatchNorm(+activation) backward apply pass that finishes the statistics pass's chunk partials itself (the backward twin of
// k_bn_act_apply_fin): statistics + this = 2 launches instead of statistics, finish, apply.  Every workgroup re-reads the partial
// columns of its channel group (<= 32 KB, L2-resident), so it is taken for SMALL layers only (ww_bn_act_bwd): there the finish
// launch is a 4.8 us latency chain between two other short kernels; on a tall layer the re-read costs more than the launch.
__global__ __launch_bounds__(256) void k_bnact_bwd_apply_fin(const float *__restrict__ x, const float *__restrict__ da,
                                                             const float *__restrict__ part, int chunks, long M, int C, BnTile t,
                                                             const float *__restrict__ ss, const float *__restrict__ mr, int act,
                                                             float *__restrict__ dx, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta) {
    __shared__ double redd[256 * 4];
    __shared__ double tot[128];
    __shared__ __align__(16) float sums[128];
    const int gc = blockIdx.x % t.G_c, c0 = gc * 4 * t.CG4, CG = 4 * t.CG4;
    const long rc = blockIdx.x / t.G_c;
    bn_group_totals(part, chunks, C, c0, t.CG4, tot, redd);
    if (threadIdx.x < CG) {
        const float s1 = (float)tot[threadIdx.x], s2 = (float)tot[CG + threadIdx.x];
        sums[threadIdx.x] = s1; sums[CG + threadIdx.x] = s2;
        if (rc == 0 && c0 + threadIdx.x < C) { dbeta[c0 + threadIdx.x] = s1; dgamma[c0 + threadIdx.x] = s2; }
    }
    __syncthreads();
    const int cq = threadIdx.x % t.CG4, rl = threadIdx.x / t.CG4;
    if (rl >= t.R || c0 + 4 * cq >= C) return;
    const size_t col = (size_t)c0 + 4 * cq;
    const float4 sc = *reinterpret_cast<const float4 *>(ss + col), sf = *reinterpret_cast<const float4 *>(ss + C + col);
    const float4 mu = *reinterpret_cast<const float4 *>(mr + col), rs = *reinterpret_cast<const float4 *>(mr + C + col);
    const float4 s1 = *reinterpret_cast<const float4 *>(sums + 4 * cq), s2 = *reinterpret_cast<const float4 *>(sums + CG + 4 * cq);
    const float invM = 1.0f / (float)M;
    const long r0 = rc * t.rows_per_block, r1 = min(M, r0 + t.rows_per_block);
    for (long rb = r0 + rl; rb < r1; rb += 4L * t.R) {
        float4 xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = (size_t)min(rb + (long)u * t.R, r1 - 1) * C + col;
            xv[u] = *reinterpret_cast<const float4 *>(x + o);
            dv[u] = *reinterpret_cast<const float4 *>(da + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (rb + (long)u * t.R < r1)
                *reinterpret_cast<float4 *>(dx + (size_t)(rb + (long)u * t.R) * C + col) =
                    make_float4(bnact_bwd_one(xv[u].x, dv[u].x, sc.x, sf.x, mu.x, rs.x, s1.x, s2.x, invM, act, 1),
                                bnact_bwd_one(xv[u].y, dv[u].y, sc.y, sf.y, mu.y, rs.y, s1.y, s2.y, invM, act, 1),
                                bnact_bwd_one(xv[u].z, dv[u].z, sc.z, sf.z, mu.z, rs.z, s1.z, s2.z, invM, act, 1),
                                bnact_bwd_one(xv[u].w, dv[u].w, sc.w, sf.w, mu.w, rs.w, s1.w, s2.w, invM, act, 1));
    }
}

// ---- depthwise k x k, stride s, padding k/2.  thread = (output pixel, 4 channels); w (C,1,k,k) as in nn.Conv2d
struct DwG { int B, H, W, C, k, s, Ho, Wo; };
__global__ __launch_bounds__(256) void k_dwg_fwd(const float *__restrict__ x, const float *__restrict__ w, DwG g,
                                                 float *__restrict__ y) {
    extern __shared__ __align__(16) float wl[];      // the layer's weights, transposed to [tap][C]: one ds_read_b128 per tap
    const int c4n = (g.C + 3) / 4, pad = g.k / 2, kk = g.k * g.k;
    const bool vec = (g.C & 3) == 0;
    for (int i = threadIdx.x; i < g.C * kk; i += 256) wl[(i % kk) * g.C + i / kk] = w[i];
    __syncthreads();
    const uint32_t n = (uint32_t)g.B * g.Ho * g.Wo * c4n;      // < 2^31 (host check): 32-bit index arithmetic
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        uint32_t p = i / (uint32_t)c4n;
        const int cq = (int)(i - p * c4n);
        uint32_t q = p / (uint32_t)g.Wo;
        const int wo = (int)(p - q * g.Wo);
        const int b = (int)(q / (uint32_t)g.Ho);
        const int ho = (int)(q - (uint32_t)b * g.Ho);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < g.k; ++kh) {
            const int hi = ho * g.s + kh - pad;
            if (hi < 0 || hi >= g.H) continue;
            for (int kw = 0; kw < g.k; ++kw) {
                const int wi = wo * g.s + kw - pad;
                if (wi < 0 || wi >= g.W) continue;
                const float *xp = x + (((size_t)b * g.H + hi) * g.W + wi) * g.C + 4 * cq;
                if (vec) {
                    const float4 xv = *reinterpret_cast<const float4 *>(xp);
                    const float4 wv = *reinterpret_cast<const float4 *>(wl + (kh * g.k + kw) * g.C + 4 * cq);
                    acc[0] = fmaf(xv.x, wv.x, acc[0]); acc[1] = fmaf(xv.y, wv.y, acc[1]);
                    acc[2] = fmaf(xv.z, wv.z, acc[2]); acc[3] = fmaf(xv.w, wv.w, acc[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * cq + e < g.C) acc[e] = fmaf(xp[e], wl[(kh * g.k + kw) * g.C + 4 * cq + e], acc[e]);
                }
            }
        }
        float *yp = y + (((size_t)b * g.Ho + ho) * g.Wo + wo) * g.C + 4 * cq;
        if (vec) {
            *reinterpret_cast<float4 *>(yp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * cq + e < g.C) yp[e] = acc[e];
        }
    }
}
// dx[b,hi,wi,c] = sum over taps with (hi + pad - kh) divisible by s of w[c,kh,kw] * dy[b,(hi+pad-kh)/s,(wi+pad-kw)/s,c]
__global__ __launch_bounds__(256) void k_dwg_bwd_dx(const float *__restrict__ dy, const float *__restrict__ w, DwG g,
                                                    float *__restrict__ dx) {
    extern __shared__ __align__(16) float wl[];      // weights as [tap][C]
    const int c4n = (g.C + 3) / 4, pad = g.k / 2, kk = g.k * g.k;
    const bool vec = (g.C & 3) == 0;
    for (int i = threadIdx.x; i < g.C * kk; i += 256) wl[(i % kk) * g.C + i / kk] = w[i];
    __syncthreads();
    const uint32_t n = (uint32_t)g.B * g.H * g.W * c4n;        // < 2^31 (host check): 32-bit index arithmetic
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        uint32_t p = i / (uint32_t)c4n;
        const int cq = (int)(i - p * c4n);
        uint32_t q = p / (uint32_t)g.W;
        const int wi = (int)(p - q * g.W);
        const int b = (int)(q / (uint32_t)g.H);
        const int hi = (int)(q - (uint32_t)b * g.H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < g.k; ++kh) {
            const int th = hi + pad - kh;
            if (th < 0 || th % g.s) continue;
            const int ho = th / g.s;
            if (ho >= g.Ho) continue;
            for (int kw = 0; kw < g.k; ++kw) {
                const int tw = wi + pad - kw;
                if (tw < 0 || tw % g.s) continue;
                const int wo = tw / g.s;
                if (wo >= g.Wo) continue;
                const float *dp = dy + (((size_t)b * g.Ho + ho) * g.Wo + wo) * g.C + 4 * cq;
                if (vec) {
                    const float4 dv = *reinterpret_cast<const float4 *>(dp);
                    const float4 wv = *reinterpret_cast<const float4 *>(wl + (kh * g.k + kw) * g.C + 4 * cq);
                    acc[0] = fmaf(dv.x, wv.x, acc[0]); acc[1] = fmaf(dv.y, wv.y, acc[1]);
                    acc[2] = fmaf(dv.z, wv.z, acc[2]); acc[3] = fmaf(dv.w, wv.w, acc[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * cq + e < g.C) acc[e] = fmaf(dp[e], wl[(kh * g.k + kw) * g.C + 4 * cq + e], acc[e]);
                }
            }
        }
        float *xp = dx + (((size_t)b * g.H + hi) * g.W + wi) * g.C + 4 * cq;
        if (vec) {
            *reinterpret_cast<float4 *>(xp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * cq + e < g.C) xp[e] = acc[e];
        }
    }
}
// C % 4 == 0 forms of the two gathers, templated on the kernel size: every tap's float4 is loaded UNCONDITIONALLY from clamped
// coordinates and an out-of-range tap is zeroed by a select afterwards.  The guarded form above (a `continue` per tap) compiles
// to a branch and a wait around every load -- up to 25 dependent L2 round trips per output, 20-40 us for tensors of a few MB;
// this way a thread has all k*k loads in flight at once.  Same products in the same (kh, kw) order: the same bits.
template <int K>
__global__ __launch_bounds__(256) void k_dwg_fwd4(const float *__restrict__ x, const float *__restrict__ w, DwG g,
                                                  float *__restrict__ y) {
    extern __shared__ __align__(16) float wl[];      // [tap][C]
    constexpr int KK = K * K, PAD = K / 2;
    const int c4n = g.C >> 2;
    for (int i = threadIdx.x; i < g.C * KK; i += 256) wl[(i % KK) * g.C + i / KK] = w[i];
    __syncthreads();
    const uint32_t n = (uint32_t)g.B * g.Ho * g.Wo * c4n;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        uint32_t p = i / (uint32_t)c4n;
        const int cq = (int)(i - p * c4n);
        uint32_t q = p / (uint32_t)g.Wo;
        const int wo = (int)(p - q * g.Wo);
        const int b = (int)(q / (uint32_t)g.Ho);
        const int ho = (int)(q - (uint32_t)b * g.Ho);
        const float *xb = x + (size_t)b * g.H * g.W * g.C + 4 * cq;
        float4 xv[KK];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int hi = min(max(ho * g.s + kh - PAD, 0), g.H - 1);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int wi = min(max(wo * g.s + kw - PAD, 0), g.W - 1);
                xv[kh * K + kw] = *reinterpret_cast<const float4 *>(xb + (uint32_t)((hi * g.W + wi) * g.C));
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int hi = ho * g.s + kh - PAD;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int wi = wo * g.s + kw - PAD;
                if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) {
                    const float4 wv = *reinterpret_cast<const float4 *>(wl + (kh * K + kw) * g.C + 4 * cq);
                    const float4 v = xv[kh * K + kw];
                    acc.x = fmaf(v.x, wv.x, acc.x); acc.y = fmaf(v.y, wv.y, acc.y);
                    acc.z = fmaf(v.z, wv.z, acc.z); acc.w = fmaf(v.w, wv.w, acc.w);
                }
            }
        }
        *reinterpret_cast<float4 *>(y + (size_t)p * g.C + 4 * cq) = acc;
    }
}
template <int K>
__global__ __launch_bounds__(256) void k_dwg_bwd_dx4(const float *__restrict__ dy, const float *__restrict__ w, DwG g,
                                                     float *__restrict__ dx) {
    extern __shared__ __align__(16) float wl[];      // [tap][C]
    constexpr int KK = K * K, PAD = K / 2;
    const int c4n = g.C >> 2;
    for (int i = threadIdx.x; i < g.C * KK; i += 256) wl[(i % KK) * g.C + i / KK] = w[i];
    __syncthreads();
    const uint32_t n = (uint32_t)g.B * g.H * g.W * c4n;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        uint32_t p = i / (uint32_t)c4n;
        const int cq = (int)(i - p * c4n);
        uint32_t q = p / (uint32_t)g.W;
        const int wi = (int)(p - q * g.W);
        const int b = (int)(q / (uint32_t)g.H);
        const int hi = (int)(q - (uint32_t)b * g.H);
        const float *db = dy + (size_t)b * g.Ho * g.Wo * g.C + 4 * cq;
        float4 dv[KK];
        uint32_t ok = 0;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int th = hi + PAD - kh, ho = g.s == 2 ? th >> 1 : th;      // stride 1 or 2 (host check)
            const bool okh = th >= 0 && (g.s == 1 || !(th & 1)) && ho < g.Ho;
            const int hoc = min(max(ho, 0), g.Ho - 1);
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int tw = wi + PAD - kw, wo = g.s == 2 ? tw >> 1 : tw;
                const bool okw = tw >= 0 && (g.s == 1 || !(tw & 1)) && wo < g.Wo;
                const int woc = min(max(wo, 0), g.Wo - 1);
                dv[kh * K + kw] = *reinterpret_cast<const float4 *>(db + (uint32_t)((hoc * g.Wo + woc) * g.C));
                ok |= (uint32_t)(okh && okw) << (kh * K + kw);
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < KK; ++t)
            if (ok & (1u << t)) {
                const float4 wv = *reinterpret_cast<const float4 *>(wl + t * g.C + 4 * cq);
                acc.x = fmaf(dv[t].x, wv.x, acc.x); acc.y = fmaf(dv[t].y, wv.y, acc.y);
                acc.z = fmaf(dv[t].z, wv.z, acc.z); acc.w = fmaf(dv[t].w, wv.w, acc.w);
            }
        *reinterpret_cast<float4 *>(dx + (size_t)p * g.C + 4 * cq) = acc;
    }
}
// ---- small feature maps (H*W <= 128 pixels: the 5x19, 3x10 and 2x5 stages of MobileNetV3 at 40 x 151 inputs, 9 of its 11 depthwise
// layers).  There every output needs most of its image, so the gather kernels above re-fetch each input float4 up to k*k times
// through L1 (a 5x5 layer on 2x5 maps moved 147 MB for a 5.9 MB tensor: 37 us).  Here a workgroup stages NIMG whole images x a
// chunk of <= 64 channels in LDS once (coalesced float4 rows, all loads in flight) and gathers from LDS.
struct DwL { int nimg, CC4, G_c; };        // images per workgroup, float4 channels per chunk, channel chunks
// stage rows [0, npix) of images b0 .. b0+nimg-1, channel float4s [c0q, c0q+CC4) of a (B, npix, C) tensor into slab[img][pix][CC4]
__device__ __forceinline__ void dwl_stage(const float *__restrict__ src, int B, int b0, int nimg, int npix, int C, int c0q, int CC4,
                                          float4 *slab) {
    const int C4 = C >> 2, per = npix * CC4, n = nimg * per;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + 256 * u, n - 1), img = i / per, rem = i - img * per, pix = rem / CC4, cq = rem - pix * CC4;
            const int b = min(b0 + img, B - 1), c4 = min(c0q + cq, C4 - 1);
            v[u] = *reinterpret_cast<const float4 *>(src + ((size_t)b * npix + pix) * C + 4 * c4);
        }
        // all four live at once, and the stores unconditional too (a clamped slot is rewritten with its own value): behind a guard
        // the compiler sinks every load next to its store -- load, wait, store, load ... (seen in the ISA: four round trips per batch)
        asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[0].z), "+v"(v[0].w), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[1].z), "+v"(v[1].w),
                     "+v"(v[2].x), "+v"(v[2].y), "+v"(v[2].z), "+v"(v[2].w), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[3].z), "+v"(v[3].w));
#pragma unroll
        for (int u = 0; u < 4; ++u) slab[min(i0 + 256 * u, n - 1)] = v[u];
    }
}
// BWD = false: y = conv(x);  BWD = true: dx = conv^T(dy).  `in` has (Hi, Wi) pixels per image, `out` (Hq, Wq).
// Thread = (channel float4 cq, lane): it walks the workgroup's (image, output pixel) pairs lane, lane + R, ...  STATS (forward):
// the per-channel sum / sum of squares of the workgroup's outputs go to stat_part[image group][2C] (fixed-order LDS sum over
// the lanes) -- the BatchNorm statistics partials of the layer, so no separate statistics pass reads y again.
template <int K, bool BWD, bool STATS>
__global__ __launch_bounds__(256) void k_dwl_conv(const float *__restrict__ in, const float *__restrict__ w, DwG g, DwL l,
                                                  float *__restrict__ out, float *__restrict__ stat_part) {
    extern __shared__ __align__(16) float4 dwl_lds[];          // wl4 [KK][CC4] | slab [nimg][Hi*Wi][CC4] | (STATS) red [2][256]
    constexpr int KK = K * K, PAD = K / 2;
    const int Hi = BWD ? g.Ho : g.H, Wi = BWD ? g.Wo : g.W, Hq = BWD ? g.H : g.Ho, Wq = BWD ? g.W : g.Wo;
    const int gc = blockIdx.x % l.G_c, grp = blockIdx.x / l.G_c, b0 = grp * l.nimg, c0q = gc * l.CC4, C4 = g.C >> 2;
    float4 *wl4 = dwl_lds, *slab = dwl_lds + KK * l.CC4;
    {   // w (C,1,k,k) -> wl4[tap][cq].{x,y,z,w}: a thread's <= 7 weights (k = 5, 64 channels) as ONE batch of clamped loads
        const int nw = KK * l.CC4 * 4;
        float wv[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int i = min((int)threadIdx.x + 256 * u, nw - 1), e = i & 3, cq = (i >> 2) % l.CC4, t = (i >> 2) / l.CC4;
            wv[u] = w[(size_t)min(4 * (c0q + cq) + e, g.C - 1) * KK + t];
        }
        asm volatile("" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(wv[4]), "+v"(wv[5]), "+v"(wv[6]));
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int i = min((int)threadIdx.x + 256 * u, nw - 1), e = i & 3, cq = (i >> 2) % l.CC4;
            reinterpret_cast<float *>(wl4)[i] = 4 * (c0q + cq) + e < g.C ? wv[u] : 0.f;
        }
    }
    const int nimg = min(l.nimg, g.B - b0);
    dwl_stage(in, g.B, b0, nimg, Hi * Wi, g.C, c0q, l.CC4, slab);
    __syncthreads();
    const int cq = threadIdx.x % l.CC4, lane = threadIdx.x / l.CC4, R = 256 / l.CC4, HqWq = Hq * Wq;
    const bool live = lane < R && c0q + cq < C4;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = s4;
    if (live)
        for (int pp = lane; pp < nimg * HqWq; pp += R) {
            const int img = pp / HqWq, pix = pp - img * HqWq, ho = pix / Wq, wo = pix - ho * Wq;
            const float4 *sl = slab + (size_t)img * Hi * Wi * l.CC4 + cq;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                int hi;
                bool okh;
                if (BWD) { const int th = ho + PAD - kh; hi = g.s == 2 ? th >> 1 : th; okh = th >= 0 && (g.s == 1 || !(th & 1)) && hi < Hi; }
                else { hi = ho * g.s + kh - PAD; okh = hi >= 0 && hi < Hi; }
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    int wi;
                    bool okw;
                    if (BWD) { const int tw = wo + PAD - kw; wi = g.s == 2 ? tw >> 1 : tw; okw = tw >= 0 && (g.s == 1 || !(tw & 1)) && wi < Wi; }
                    else { wi = wo * g.s + kw - PAD; okw = wi >= 0 && wi < Wi; }
                    if (okh && okw) {
                        const float4 v = sl[(hi * Wi + wi) * l.CC4], wv = wl4[(kh * K + kw) * l.CC4 + cq];
                        acc.x = fmaf(v.x, wv.x, acc.x); acc.y = fmaf(v.y, wv.y, acc.y);
                        acc.z = fmaf(v.z, wv.z, acc.z); acc.w = fmaf(v.w, wv.w, acc.w);
                    }
                }
            }
            *reinterpret_cast<float4 *>(out + ((size_t)(b0 + img) * HqWq + pix) * g.C + 4 * (c0q + cq)) = acc;
            if (STATS) {
                s4.x += acc.x; s4.y += acc.y; s4.z += acc.z; s4.w += acc.w;
                q4.x = fmaf(acc.x, acc.x, q4.x); q4.y = fmaf(acc.y, acc.y, q4.y);
                q4.z = fmaf(acc.z, acc.z, q4.z); q4.w = fmaf(acc.w, acc.w, q4.w);
            }
        }
    if (STATS) {
        float4 *red = slab + (size_t)l.nimg * Hi * Wi * l.CC4;      // [2][256]
        red[threadIdx.x] = s4; red[256 + threadIdx.x] = q4;
        __syncthreads();
        if (lane == 0 && c0q + cq < C4) {
            double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int q = 0; q < R; ++q) {
                const float4 a = red[q * l.CC4 + cq], b = red[256 + q * l.CC4 + cq];
                t[0] += a.x; t[1] += a.y; t[2] += a.z; t[3] += a.w; t[4] += b.x; t[5] += b.y; t[6] += b.z; t[7] += b.w;
            }
            float *o = stat_part + (size_t)grp * 2 * g.C + 4 * (c0q + cq);
            *reinterpret_cast<float4 *>(o) = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
            *reinterpret_cast<float4 *>(o + g.C) = make_float4((float)t[4], (float)t[5], (float)t[6], (float)t[7]);
        }
    }
}
// weight gradient: x and dy slabs of the workgroup's images in LDS, thread = (tap, channel float4) summing over (image, output
// pixel) in fixed order; part[image group][C*k*k], summed over the groups by the column-sum launch that follows.
template <int K>
__global__ __launch_bounds__(256) void k_dwl_dw(const float *__restrict__ x, const float *__restrict__ dy, DwG g, DwL l,
                                                float *__restrict__ part) {
    extern __shared__ __align__(16) float4 dwl_lds[];          // xs [nimg][H*W][CC4] | ds [nimg][Ho*Wo][CC4]
    constexpr int KK = K * K, PAD = K / 2;
    const int gc = blockIdx.x % l.G_c, grp = blockIdx.x / l.G_c, b0 = grp * l.nimg, c0q = gc * l.CC4, C4 = g.C >> 2;
    const int nimg = min(l.nimg, g.B - b0), HW = g.H * g.W, HoWo = g.Ho * g.Wo;
    float4 *xs = dwl_lds, *ds = dwl_lds + (size_t)l.nimg * HW * l.CC4;
    dwl_stage(x, g.B, b0, nimg, HW, g.C, c0q, l.CC4, xs);
    dwl_stage(dy, g.B, b0, nimg, HoWo, g.C, c0q, l.CC4, ds);
    __syncthreads();
    for (int i = threadIdx.x; i < KK * l.CC4; i += 256) {
        const int t = i / l.CC4, cq = i - t * l.CC4, kh = t / K, kw = t - kh * K;
        if (c0q + cq >= C4) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int img = 0; img < nimg; ++img) {
            const float4 *xi = xs + (size_t)img * HW * l.CC4 + cq, *di = ds + (size_t)img * HoWo * l.CC4 + cq;
            for (int ho = 0; ho < g.Ho; ++ho) {
                const int hi = ho * g.s + kh - PAD;
                if (hi < 0 || hi >= g.H) continue;
                for (int wo = 0; wo < g.Wo; ++wo) {
                    const int wi = wo * g.s + kw - PAD;
                    if (wi < 0 || wi >= g.W) continue;
                    const float4 d = di[(ho * g.Wo + wo) * l.CC4], v = xi[(hi * g.W + wi) * l.CC4];
                    acc.x = fmaf(d.x, v.x, acc.x); acc.y = fmaf(d.y, v.y, acc.y);
                    acc.z = fmaf(d.z, v.z, acc.z); acc.w = fmaf(d.w, v.w, acc.w);
                }
            }
        }
        float *o = part + (size_t)grp * g.C * KK + (size_t)(4 * (c0q + cq)) * KK + t;
        o[0] = acc.x; o[KK] = acc.y; o[2 * KK] = acc.z; o[3 * KK] = acc.w;
    }
}

// dw[c][tap] partials over a chunk of output pixels (thread layout of k_colstats): part[chunk][C*k*k]
__global__ __launch_bounds__(1024) void k_dwg_bwd_dw(const float *__restrict__ x, const float *__restrict__ dy, DwG g, int R,
                                                     long px_per_chunk, float *__restrict__ part) {
    extern __shared__ float redf[];           // [R][C]
    const int c = threadIdx.x % g.C, r = threadIdx.x / g.C, pad = g.k / 2, kk = g.k * g.k;
    const long P = (long)g.B * g.Ho * g.Wo;
    const long p0 = (long)blockIdx.x * px_per_chunk, p1 = min(P, p0 + px_per_chunk);
    float acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = 0.f;
    for (long p = p0 + r; p < p1; p += R) {
        const int wo = (int)(p % g.Wo);
        const long q = p / g.Wo;
        const int ho = (int)(q % g.Ho), b = (int)(q / g.Ho);
        const float d = dy[(size_t)p * g.C + c];
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
            const int hi = ho * g.s + kh - pad;
            if (kh >= g.k || hi < 0 || hi >= g.H) continue;
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const int wi = wo * g.s + kw - pad;
                if (kw >= g.k || wi < 0 || wi >= g.W) continue;
                acc[kh * 5 + kw] = fmaf(d, x[(((size_t)b * g.H + hi) * g.W + wi) * g.C + c], acc[kh * 5 + kw]);
            }
        }
    }
    for (int kh = 0; kh < g.k; ++kh)
        for (int kw = 0; kw < g.k; ++kw) {
            __syncthreads();
            float v = 0.f;                    // acc is indexed with compile-time constants only (no scratch)
#pragma unroll
            for (int a5 = 0; a5 < 25; ++a5) v = (a5 == kh * 5 + kw) ? acc[a5] : v;
            redf[r * g.C + c] = v;
            __syncthreads();
            if (r == 0) {
                float t = 0.f;
                for (int i = 0; i < R; ++i) t += redf[i * g.C + c];
                part[(size_t)blockIdx.x * g.C * kk + (size_t)c * kk + kh * g.k + kw] = t;
            }
        }
}

// the same for C % 4 == 0: a thread owns FOUR channels (float4 loads) and rows r, r+R, ...: block = (C/4)*R <= 512 threads.
// Up to NCHUNK_DW blocks; 32-bit pixel arithmetic; the row lanes of a block are summed through LDS T taps at a time
// (T * R * C floats <= 48 KB), each (tap, channel quad) by one thread in fixed order.  Templated on the kernel size; the k*k
// input float4s of a pixel are loaded unconditionally from clamped coordinates (a guarded tap load is a branch + a wait: 25
// dependent round trips per pixel in the first form) and out-of-range taps are skipped by the accumulation only.
template <int K>
__global__ __launch_bounds__(512) void k_dwg_bwd_dw4(const float *__restrict__ x, const float *__restrict__ dy, DwG g, int R,
                                                     uint32_t px_per_chunk, int T, float *__restrict__ part) {
    extern __shared__ __align__(16) float redf[];           // [T][R][C]
    constexpr int KK = K * K, PAD = K / 2;
    const int C4 = g.C / 4, cq = threadIdx.x % C4, r = threadIdx.x / C4;
    const uint32_t P = (uint32_t)g.B * g.Ho * g.Wo;
    const uint32_t p0 = blockIdx.x * px_per_chunk, p1 = min(P, p0 + px_per_chunk);
    float ax[KK], ay[KK], az[KK], aw[KK];       // plain arrays: an array of HIP float4 structs is not promoted to registers
#pragma unroll
    for (int t = 0; t < KK; ++t) { ax[t] = 0.f; ay[t] = 0.f; az[t] = 0.f; aw[t] = 0.f; }
    for (uint32_t p = p0 + r; p < p1; p += R) {
        const uint32_t q = p / (uint32_t)g.Wo;
        const int wo = (int)(p - q * g.Wo);
        const uint32_t b = q / (uint32_t)g.Ho;
        const int ho = (int)(q - b * g.Ho);
        const float4 d = *reinterpret_cast<const float4 *>(dy + (size_t)p * g.C + 4 * cq);
        const float *xb = x + (size_t)b * g.H * g.W * g.C + 4 * cq;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int hi = ho * g.s + kh - PAD, hic = min(max(hi, 0), g.H - 1);
            float4 xv[K];
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int wic = min(max(wo * g.s + kw - PAD, 0), g.W - 1);
                xv[kw] = *reinterpret_cast<const float4 *>(xb + (uint32_t)((hic * g.W + wic) * g.C));
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int wi = wo * g.s + kw - PAD;
                if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) {
                    const int t = kh * K + kw;
                    ax[t] = fmaf(d.x, xv[kw].x, ax[t]); ay[t] = fmaf(d.y, xv[kw].y, ay[t]);
                    az[t] = fmaf(d.z, xv[kw].z, az[t]); aw[t] = fmaf(d.w, xv[kw].w, aw[t]);
                }
            }
        }
    }
    int slot = 0, base = 0;                     // taps base .. base + slot - 1 (in (kh, kw) order) sit in LDS
    float *outp = part + (size_t)blockIdx.x * g.C * KK + (size_t)(4 * cq) * KK;
#pragma unroll
    for (int t5 = 0; t5 < KK; ++t5) {           // compile-time: the accumulators are read by constant index
        *reinterpret_cast<float4 *>(redf + ((size_t)slot * R + r) * g.C + 4 * cq) = make_float4(ax[t5], ay[t5], az[t5], aw[t5]);
        ++slot;
        if (slot == T || t5 == KK - 1) {
            __syncthreads();
            for (int sl = r; sl < slot; sl += R) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int i = 0; i < R; ++i) {
                    const float4 u = *reinterpret_cast<const float4 *>(redf + ((size_t)sl * R + i) * g.C + 4 * cq);
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                float *o = outp + base + sl;
                o[0] = t.x; o[KK] = t.y; o[2 * KK] = t.z; o[3 * KK] = t.w;
            }
            __syncthreads();
            base += slot;
            slot = 0;
        }
    }
}

// ---- squeeze-excitation / pooling pieces on (B, HW, C)
// one workgroup per (image, group of 64 channels): 16 row lanes walk HW, fixed-order sum in LDS (a thread per (b, c)
// walking HW alone was latency-bound: 16 us per call at B = 256)
__global__ __launch_bounds__(1024) void k_pool_fwd(const float *__restrict__ x, int B, int HW, int C, float *__restrict__ s) {
    __shared__ double sh[16][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6, cg = (C + 63) / 64;
    const int b = blockIdx.x / cg, c = (blockIdx.x % cg) * 64 + cl;
    double a = 0.0;
    if (c < C) {
        const float *p = x + (size_t)b * HW * C + c;
#pragma unroll 8
        for (int h = part; h < HW; h += 16) a += p[(size_t)h * C];
    }
    sh[part][cl] = a;
    __syncthreads();
    if (part == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][cl];
        s[(size_t)b * C + c] = (float)(t / HW);
    }
}
// y = x * g[b][c]
__global__ __launch_bounds__(256) void k_scale_fwd(const float *__restrict__ x, const float *__restrict__ gte, int B, int HW, int C,
                                                   float *__restrict__ y) {
    const long n = (long)B * HW * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long b = i / ((long)HW * C);
        y[i] = x[i] * gte[b * C + c];
    }
}
__global__ __launch_bounds__(256) void k_scale_fwd4(const float4 *__restrict__ x, const float *__restrict__ gte, uint32_t n4,
                                                    uint32_t HWC4, int C, float4 *__restrict__ y) {
    const uint32_t C4 = (uint32_t)C >> 2;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        const uint32_t b = i / HWC4, c = (i % C4) * 4;
        const float4 v = x[i], g = *reinterpret_cast<const float4 *>(gte + (size_t)b * C + c);
        y[i] = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
    }
}
__global__ __launch_bounds__(256) void k_scale_pool_bwd4(const float4 *__restrict__ dy, const float *__restrict__ gte,
                                                         const float *__restrict__ dpool, uint32_t n4, uint32_t HWC4, int HW, int C,
                                                         float4 *__restrict__ dx) {
    const uint32_t C4 = (uint32_t)C >> 2;
    const float inv = 1.0f / (float)HW;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        const uint32_t b = i / HWC4, c = (i % C4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dy) {
            const float4 d = dy[i], g = *reinterpret_cast<const float4 *>(gte + (size_t)b * C + c);
            v = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        }
        if (dpool) {
            const float4 dp = *reinterpret_cast<const float4 *>(dpool + (size_t)b * C + c);
            v = make_float4(fmaf(dp.x, inv, v.x), fmaf(dp.y, inv, v.y), fmaf(dp.z, inv, v.z), fmaf(dp.w, inv, v.w));
        }
        dx[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_add4(const float4 *__restrict__ a, const float4 *__restrict__ b, uint32_t n4,
                                              float4 *__restrict__ y) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
        const float4 u = a[i], v = b[i];
        y[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}
// dg[b][c] = sum_hw dy * x   (same thread map as k_pool_fwd)
__global__ __launch_bounds__(1024) void k_scale_bwd_gate(const float *__restrict__ x, const float *__restrict__ dy, int B, int HW,
                                                         int C, float *__restrict__ dg) {
    __shared__ double sh[16][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6, cg = (C + 63) / 64;
    const int b = blockIdx.x / cg, c = (blockIdx.x % cg) * 64 + cl;
    double a = 0.0;
    if (c < C) {
        const size_t o = (size_t)b * HW * C + c;
#pragma unroll 8
        for (int h = part; h < HW; h += 16) a += (double)dy[o + (size_t)h * C] * (double)x[o + (size_t)h * C];
    }
    sh[part][cl] = a;
    __syncthreads();
    if (part == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][cl];
        dg[(size_t)b * C + c] = (float)t;
    }
}
// dx = dy * g[b][c] + dpool[b][c] / HW     (dy nullable: plain pooling backward; dpool nullable: plain scaling backward)
__global__ __launch_bounds__(256) void k_scale_pool_bwd(const float *__restrict__ dy, const float *__restrict__ gte,
                                                        const float *__restrict__ dpool, int B, int HW, int C,
                                                        float *__restrict__ dx) {
    const long n = (long)B * HW * C;
    const float inv = 1.0f / (float)HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long b = i / ((long)HW * C);
        float v = dy ? dy[i] * gte[b * C + c] : 0.f;
        if (dpool) v = fmaf(dpool[b * C + c], inv, v);
        dx[i] = v;
    }
}
// ---- the one-channel 3x3 stride-2 stem (Conv2d(1, C, 3, 2, 1), the first layer of MobileNetV3Wakeword) as a direct convolution.
// As patches + GEMM (ww_im2col3x3s2 + k_gemm) it was an 11 us gather, a K = 9 matrix product that the 64-wide tiles ran in 39-90 us,
// and a 14 MB patch tensor kept for the backward; a thread here owns (output pixel, 4 channels), reads its nine input samples
// (unconditional, clamped; zeroed by a select outside the image) and keeps its 36 weights in registers.  Persistent grid: a
// thread's channel quad is fixed, so its BatchNorm partial sums / weight-gradient partials accumulate in registers and leave
// through one fixed-order LDS sum per workgroup.
struct StemG { int B, H, W, C, Ho, Wo; };
__device__ __forceinline__ void stem_patch(const float *__restrict__ xb, const StemG &g, int ho, int wo, float (&p)[9]) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = 2 * ho + kh - 1, hic = min(max(hi, 0), g.H - 1);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = 2 * wo + kw - 1, wic = min(max(wi, 0), g.W - 1);
            // zeroed by a 0/1 factor, not a select: a select lets the compiler turn the load back into a guarded one (branch + wait per
            // row, seen in the ISA: 30 us instead of 10).  The clamped sample is always one of the patch's own in-range taps, so a
            // NaN there reaches the output either way.
            const float v = xb[hic * g.W + wic];
            p[kh * 3 + kw] = v * ((hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? 1.f : 0.f);
        }
    }
}
// grid = gridDim.x workgroups of 256 = 64 pixel lanes x (C/4 <= 4) ... generally threads = (pixel lane, channel quad): C4 = C/4 quads,
// 256/C4 pixel lanes.  y (B,Ho,Wo,C); stat_part[block][2C] (nullable).
__global__ __launch_bounds__(256) void k_stem3x3s2_fwd(const float *__restrict__ x, const float *__restrict__ w, StemG g,
                                                       float *__restrict__ y, float *__restrict__ stat_part) {
    __shared__ __align__(16) float4 red[2][256];
    const int C4 = g.C >> 2, cq = threadIdx.x % C4, lane = threadIdx.x / C4, L = 256 / C4;
    float wr[4][9];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[e][t] = w[(size_t)(4 * cq + e) * 9 + t];
    const uint32_t P = (uint32_t)g.B * g.Ho * g.Wo, HoWo = (uint32_t)g.Ho * g.Wo, stride = gridDim.x * L;   // P < 2^31 (host check)
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = s4;
    if (lane < L)
        for (uint32_t p0 = blockIdx.x * L + lane; p0 < P; p0 += 4u * stride) {      // four pixels per trip: 36 loads in flight
            float pt[4][9];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t p = min(p0 + u * stride, P - 1), b = p / HoWo;
                const int r = (int)(p - b * HoWo), ho = r / g.Wo, wo = r - ho * g.Wo;
                stem_patch(x + (size_t)b * g.H * g.W, g, ho, wo, pt[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t p = p0 + u * stride;
                if (p < P) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = 0.f;
#pragma unroll
                        for (int t = 0; t < 9; ++t) a = fmaf(pt[u][t], wr[e][t], a);
                        o[e] = a;
                    }
                    *reinterpret_cast<float4 *>(y + (size_t)p * g.C + 4 * cq) = make_float4(o[0], o[1], o[2], o[3]);
                    s4.x += o[0]; s4.y += o[1]; s4.z += o[2]; s4.w += o[3];
                    q4.x = fmaf(o[0], o[0], q4.x); q4.y = fmaf(o[1], o[1], q4.y); q4.z = fmaf(o[2], o[2], q4.z); q4.w = fmaf(o[3], o[3], q4.w);
                }
            }
        }
    if (!stat_part) return;
    red[0][threadIdx.x] = s4; red[1][threadIdx.x] = q4;
    __syncthreads();
    if (lane == 0) {
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < L; ++q) {
            const float4 a = red[0][q * C4 + cq], b = red[1][q * C4 + cq];
            t[0] += a.x; t[1] += a.y; t[2] += a.z; t[3] += a.w; t[4] += b.x; t[5] += b.y; t[6] += b.z; t[7] += b.w;
        }
        float *o = stat_part + (size_t)blockIdx.x * 2 * g.C + 4 * cq;
        *reinterpret_cast<float4 *>(o) = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        *reinterpret_cast<float4 *>(o + g.C) = make_float4((float)t[4], (float)t[5], (float)t[6], (float)t[7]);
    }
}
// dW[c][t] = sum_p dy[p][c] * patch[p][t]: part[block][C*9] (summed over the blocks by the column-sum launch that follows)
__global__ __launch_bounds__(256) void k_stem3x3s2_dw(const float *__restrict__ x, const float *__restrict__ dy, StemG g,
                                                      float *__restrict__ part) {
    __shared__ float red[256][37];
    const int C4 = g.C >> 2, cq = threadIdx.x % C4, lane = threadIdx.x / C4, L = 256 / C4;
    const uint32_t P = (uint32_t)g.B * g.Ho * g.Wo, HoWo = (uint32_t)g.Ho * g.Wo, stride = gridDim.x * L;   // P < 2^31 (host check)
    float acc[4][9];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[e][t] = 0.f;
    if (lane < L)
        for (uint32_t p0 = blockIdx.x * L + lane; p0 < P; p0 += 4u * stride) {      // four pixels per trip
            float pt[4][9];
            float4 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t p = min(p0 + u * stride, P - 1), b = p / HoWo;
                const int r = (int)(p - b * HoWo), ho = r / g.Wo, wo = r - ho * g.Wo;
                d[u] = *reinterpret_cast<const float4 *>(dy + (size_t)p * g.C + 4 * cq);
                stem_patch(x + (size_t)b * g.H * g.W, g, ho, wo, pt[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p0 + u * stride < P) {
                    const float dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < 9; ++t) acc[e][t] = fmaf(dv[e], pt[u][t], acc[e][t]);
                }
        }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) red[threadIdx.x][e * 9 + t] = acc[e][t];
    __syncthreads();
    // C*9 outputs, each the fixed-order sum over the L pixel lanes of its channel quad
    for (int o = threadIdx.x; o < g.C * 9; o += 256) {
        const int c = o / 9, t = o - c * 9, q4 = c >> 2, e = c & 3;
        float sum = 0.f;
        for (int q = 0; q < L; ++q) sum += red[q * C4 + q4][e * 9 + t];
        part[(size_t)blockIdx.x * g.C * 9 + o] = sum;
    }
}

// 3x3 stride-2 pad-1 patches of a single-channel image: cols (B*Ho*Wo, 9)
__global__ __launch_bounds__(256) void k_im2col3x3s2(const float *__restrict__ x, int B, int H, int W, int Ho, int Wo,
                                                     float *__restrict__ cols) {
    const long n = (long)B * Ho * Wo * 9;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i % 9);
        long p = i / 9;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const int hi = 2 * ho + t / 3 - 1, wi = 2 * wo + t % 3 - 1;
        cols[i] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((size_t)b * H + hi) * W + wi] : 0.f;
    }
}
__global__ __launch_bounds__(256) void k_add(const float *__restrict__ a, const float *__restrict__ b, long n, float *__restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a[i] + b[i];
}

inline int egrid(long n) { return (int)std::min<long>((n + 255) / 256, 256 * 32); }
// float4 form usable: channel count a multiple of 4, every base pointer 16-byte aligned, index fits 32 bits
inline bool vec4_ok(long n, int C, std::initializer_list<const void *> ptrs) {
    if ((C & 3) || (n & 3) || n / 4 >= (1L << 31)) return false;
    for (const void *p : ptrs)
        if ((uintptr_t)p & 15) return false;
    return true;
}
inline int rows_r(int C) { return std::max(1, 1024 / C); }                       // row lanes R of a C*R-thread block
inline int chunks_for(long M, int C) { return (int)std::max<long>(1, std::min<long>(NCHUNK, M / (4L * rows_r(C)))); }
}  // namespace

// scratch of one layer call: chunk partials (BatchNorm: 2C per chunk, depthwise weight gradient: up to 25C per chunk) + 2C sums
extern "C" size_t ww_nhwc_scratch_bytes(int C) { return (size_t)(NCHUNK_DW * 25 + 2) * std::max(C, 1) * sizeof(float); }

extern "C" int ww_bn_act_fwd(ww_ctx *ctx, const float *x, long M, int C, const ww_bn_t *bn, int act, float *y, float *ss,
                             float *mr, void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && bn && bn->gamma && bn->beta && y && ss && mr && scratch, WW_E_INVALID, "ww_bn_act_fwd: null argument");
    WW_REQUIRE(M >= 1 && C >= 1, WW_E_INVALID, "ww_bn_act_fwd: bad shape (%ld,%d)", M, C);
    WW_REQUIRE(bn->training || (bn->running_mean && bn->running_var), WW_E_INVALID, "ww_bn_act_fwd: eval needs running statistics");
    WW_REQUIRE(act >= WW_LIN_NONE && act <= WW_LIN_HARDSIGMOID, WW_E_INVALID, "ww_bn_act_fwd: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    float *part = (float *)scratch;
    WW_REQUIRE(C <= 1024, WW_E_UNSUPPORTED, "ww_bn_act_fwd: C=%d > 1024", C);
    const int R = rows_r(C);
    const int chunks = chunks_for(M, C);
    if (bn->training) {
        hipLaunchKernelGGL(k_colstats, dim3(chunks), dim3(C * R), (size_t)2 * R * C * sizeof(double), st, x, M, C, R,
                           (M + chunks - 1) / chunks, part);
        WW_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_bn_finish, dim3((C + 63) / 64), dim3(1024), 0, st, part, chunks, M, C, *bn, ss, mr);
    WW_LAUNCH_CHECK();
    if (vec4_ok(M * C, C, {x, y, ss}))
        hipLaunchKernelGGL(k_bn_act_apply4, dim3(egrid(M * C / 4)), dim3(256), 0, st, (const float4 *)x, ss, (uint32_t)(M * C / 4), C,
                           act, (float4 *)y);
    else
        hipLaunchKernelGGL(k_bn_act_apply, dim3(egrid(M * C)), dim3(256), 0, st, x, ss, M * C, C, act, y);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

// (row chunk, channel group) tiling of k_bn_act_apply_fin: the widest group (16, 8 or 4 channel float4s) whose re-read of the
// partials stays <= 32 KB per workgroup; CG4 == 0: too many partial rows, take the finish launch instead
static BnTile bn_tile(long M, int C, int chunks) {
    BnTile t = {0, 0, 0, 0};
    const int C4 = C / 4;
    for (int cg4 = 16; cg4 >= 4; cg4 >>= 1) {
        const int G_c = (C4 + cg4 - 1) / cg4, CG4 = (C4 + G_c - 1) / G_c;
        if ((size_t)chunks * 2 * 4 * CG4 * sizeof(float) > 32 * 1024) continue;
        t.G_c = G_c; t.CG4 = CG4; t.R = 256 / CG4;
        const long G_r = std::max<long>(1, std::min<long>(std::max(1, 1024 / G_c), M / (4L * t.R)));
        t.rows_per_block = (M + G_r - 1) / G_r;
        break;
    }
    return t;
}
// training-mode BatchNorm(+activation) of x (M, C) whose statistics partials -- `chunks` rows of [sum (C) | sum of squares (C)]
// -- a producer has already written to `part`
int ww_bn_act_from_partials(ww_ctx *ctx, const float *x, long M, int C, const ww_bn_t *bn, int act, float *y, float *ss, float *mr,
                            const float *part, int chunks, const float *res, hipStream_t st) {
    WW_REQUIRE(C <= 1024 && chunks >= 1, WW_E_UNSUPPORTED, "ww_bn_act_from_partials: C=%d > 1024", C);
    const bool v4 = vec4_ok(M * C, C, {x, y, part, res});
    const BnTile t = v4 ? bn_tile(M, C, chunks) : BnTile{0, 0, 0, 0};
    if (t.CG4) {
        const int blocks = (int)((M + t.rows_per_block - 1) / t.rows_per_block) * t.G_c;
        hipLaunchKernelGGL(k_bn_act_apply_fin, dim3(blocks), dim3(256), 0, st, x, part, chunks, M, C, t, *bn, act, y, ss, mr, res);
        WW_LAUNCH_CHECK();
        return WW_OK;
    }
    hipLaunchKernelGGL(k_bn_finish, dim3((C + 63) / 64), dim3(1024), 0, st, part, chunks, M, C, *bn, ss, mr);
    WW_LAUNCH_CHECK();
    if (v4)
        hipLaunchKernelGGL(k_bn_act_apply4, dim3(egrid(M * C / 4)), dim3(256), 0, st, (const float4 *)x, ss, (uint32_t)(M * C / 4), C,
                           act, (float4 *)y);
    else
        hipLaunchKernelGGL(k_bn_act_apply, dim3(egrid(M * C)), dim3(256), 0, st, x, ss, M * C, C, act, y);
    WW_LAUNCH_CHECK();
    if (res) return ww_add_f32(ctx, y, res, (size_t)(M * C), y, (ww_stream_t)st);      // tall layers: the add stays its own pass (in place)
    return WW_OK;
}

extern "C" int ww_bn_act_bwd(ww_ctx *ctx, const float *x, const float *da, long M, int C, const float *ss, const float *mr, int act,
                             int training, float *dx, float *dgamma, float *dbeta, void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && da && ss && mr && dx && dgamma && dbeta && scratch, WW_E_INVALID, "ww_bn_act_bwd: null argument");
    WW_REQUIRE(M >= 1 && C >= 1, WW_E_INVALID, "ww_bn_act_bwd: bad shape (%ld,%d)", M, C);
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    float *part = (float *)scratch, *sums = part + (size_t)NCHUNK * 2 * C;
    WW_REQUIRE(C <= 1024, WW_E_UNSUPPORTED, "ww_bn_act_bwd: C=%d > 1024", C);
    const int R = rows_r(C);
    const int chunks = chunks_for(M, C);
    hipLaunchKernelGGL(k_bnact_bwd_stats, dim3(chunks), dim3(C * R), (size_t)2 * R * C * sizeof(double), st, x, da, ss, mr, M, C, R,
                       act, (M + chunks - 1) / chunks, part);
    WW_LAUNCH_CHECK();
    // the apply pass finishes the partials itself (2 launches instead of 3) up to WW_BN_BWD_FUSED_MAX_KB of activation tensor:
    // MobileNetV3 B=256 step 2.414 / 2.398 / 2.393 / 2.385 / 2.388 ms at 0 / 1 / 4 / 16 / 64 MB (profiles/r03_p_*)
    static const long fused_max = (long)ww_env_int("WW_BN_BWD_FUSED_MAX_KB", 16384) * 1024;
    if (training && M * C * (long)sizeof(float) <= fused_max && vec4_ok(M * C, C, {x, da, dx, ss, mr, part})) {
        const BnTile t = bn_tile(M, C, chunks);
        if (t.CG4) {
            const int blocks = (int)((M + t.rows_per_block - 1) / t.rows_per_block) * t.G_c;
            hipLaunchKernelGGL(k_bnact_bwd_apply_fin, dim3(blocks), dim3(256), 0, st, x, da, part, chunks, M, C, t, ss, mr, act, dx,
                               dgamma, dbeta);
            WW_LAUNCH_CHECK();
            return WW_OK;
        }
    }
    hipLaunchKernelGGL(k_bnact_bwd_finish, dim3((C + 63) / 64), dim3(1024), 0, st, part, chunks, C, sums, dgamma, dbeta);
    WW_LAUNCH_CHECK();
    if (vec4_ok(M * C, C, {x, da, dx, ss, mr, sums}))
        hipLaunchKernelGGL(k_bnact_bwd_apply4, dim3(egrid(M * C / 4)), dim3(256), 0, st, (const float4 *)x, (const float4 *)da, ss,
                           mr, sums, M, (uint32_t)(M * C / 4), C, act, training, (float4 *)dx);
    else
        hipLaunchKernelGGL(k_bnact_bwd_apply, dim3(egrid(M * C)), dim3(256), 0, st, x, da, ss, mr, sums, M, C, act, training, dx);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

// LDS plan of the small-feature-map depthwise kernels: usable when C % 4 == 0 and an image has <= 128 input pixels; images per
// workgroup so that the slabs stay under `budget` bytes and (for the weight gradient) the image groups fit the partial slab
static bool dwl_plan(const DwG &g, size_t bytes_per_img_cc4, size_t budget, DwL *l) {
    if ((g.C & 3) || g.H * g.W > 128) return false;
    const int C4 = g.C / 4;
    l->G_c = (C4 + 15) / 16;
    l->CC4 = (C4 + l->G_c - 1) / l->G_c;
    const size_t per_img = bytes_per_img_cc4 * l->CC4;
    if (per_img > budget) return false;
    int nimg = (int)std::min<size_t>(8, budget / per_img);
    while (nimg > 1 && (long)((g.B + nimg - 1) / nimg) * l->G_c < 512) --nimg;      // keep the device covered
    l->nimg = nimg;
    return true;
}
static int make_dwg(const char *who, int B, int H, int W, int C, int k, int s, DwG *g) {
    WW_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, WW_E_INVALID, "%s: bad shape", who);
    WW_REQUIRE((k == 3 || k == 5) && (s == 1 || s == 2), WW_E_UNSUPPORTED, "%s: kernel %d stride %d not implemented", who, k, s);
    g->B = B; g->H = H; g->W = W; g->C = C; g->k = k; g->s = s;
    g->Ho = (H + 2 * (k / 2) - k) / s + 1;
    g->Wo = (W + 2 * (k / 2) - k) / s + 1;
    return WW_OK;
}
extern "C" int ww_dwconv_nhwc_fwd(ww_ctx *ctx, const float *x, const float *w, int B, int H, int W, int C, int k, int stride,
                                  float *y, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && y, WW_E_INVALID, "ww_dwconv_nhwc_fwd: null argument");
    DwG g;
    int rc = make_dwg("ww_dwconv_nhwc_fwd", B, H, W, C, k, stride, &g);
    if (rc) return rc;
    ww_prof_scope ps_(ctx, WW_K_NHWC, (hipStream_t)stream);
    const size_t wbytes = (size_t)C * k * k * sizeof(float);
    WW_REQUIRE(wbytes <= 64 * 1024, WW_E_UNSUPPORTED, "ww_dwconv_nhwc_fwd: C*k*k = %d weights do not fit the LDS cache", C * k * k);
    WW_REQUIRE((long)B * H * W * C < (1L << 31), WW_E_UNSUPPORTED, "ww_dwconv_nhwc_fwd: tensor too large for 32-bit indices");
    const dim3 grid(egrid((long)B * g.Ho * g.Wo * ((C + 3) / 4)));
    DwL l;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) == 0 && dwl_plan(g, (size_t)H * W * 16, 40 * 1024, &l)) {
        const dim3 lg((unsigned)((B + l.nimg - 1) / l.nimg) * l.G_c);
        const size_t lds = ((size_t)k * k * l.CC4 + (size_t)l.nimg * H * W * l.CC4) * 16;
        if (k == 3) hipLaunchKernelGGL((k_dwl_conv<3, false, false>), lg, dim3(256), lds, (hipStream_t)stream, x, w, g, l, y, (float *)nullptr);
        else hipLaunchKernelGGL((k_dwl_conv<5, false, false>), lg, dim3(256), lds, (hipStream_t)stream, x, w, g, l, y, (float *)nullptr);
    } else if ((C & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        if (k == 3) hipLaunchKernelGGL(k_dwg_fwd4<3>, grid, dim3(256), wbytes, (hipStream_t)stream, x, w, g, y);
        else hipLaunchKernelGGL(k_dwg_fwd4<5>, grid, dim3(256), wbytes, (hipStream_t)stream, x, w, g, y);
    } else {
        hipLaunchKernelGGL(k_dwg_fwd, grid, dim3(256), wbytes, (hipStream_t)stream, x, w, g, y);
    }
    WW_LAUNCH_CHECK();
    return WW_OK;
}
extern "C" int ww_dwconv_bn_act_fwd(ww_ctx *ctx, const float *x, const float *w, int B, int H, int W, int C, int k, int stride,
                                    const ww_bn_t *bn, int act, float *y, float *a, float *ss, float *mr, void *scratch,
                                    ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && bn && bn->gamma && bn->beta && y && a && ss && mr && scratch, WW_E_INVALID, "ww_dwconv_bn_act_fwd: null argument");
    DwG g;
    int rc = make_dwg("ww_dwconv_bn_act_fwd", B, H, W, C, k, stride, &g);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)B * g.Ho * g.Wo;
    DwL l;
    if (bn->training && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)scratch) & 15) == 0 && dwl_plan(g, (size_t)H * W * 16, 40 * 1024, &l)) {
        // the LDS kernel leaves one row of statistics partials per image group: conv + (finishing) apply = 2 launches
        ww_prof_scope ps_(ctx, WW_K_NHWC, st);
        const int groups = (B + l.nimg - 1) / l.nimg;
        const dim3 lg((unsigned)groups * l.G_c);
        const size_t lds = ((size_t)k * k * l.CC4 + (size_t)l.nimg * H * W * l.CC4 + 512) * 16;
        float *part = (float *)scratch;
        if (k == 3) hipLaunchKernelGGL((k_dwl_conv<3, false, true>), lg, dim3(256), lds, st, x, w, g, l, y, part);
        else hipLaunchKernelGGL((k_dwl_conv<5, false, true>), lg, dim3(256), lds, st, x, w, g, l, y, part);
        WW_LAUNCH_CHECK();
        return ww_bn_act_from_partials(ctx, y, M, C, bn, act, a, ss, mr, part, groups, nullptr, st);
    }
    if ((rc = ww_dwconv_nhwc_fwd(ctx, x, w, B, H, W, C, k, stride, y, stream))) return rc;
    return ww_bn_act_fwd(ctx, y, M, C, bn, act, a, ss, mr, scratch, stream);
}
extern "C" int ww_dwconv_nhwc_bwd(ww_ctx *ctx, const float *x, const float *w, const float *dy, int B, int H, int W, int C, int k,
                                  int stride, float *dx, float *dw, void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && dy && dw && scratch, WW_E_INVALID, "ww_dwconv_nhwc_bwd: null argument");
    DwG g;
    int rc = make_dwg("ww_dwconv_nhwc_bwd", B, H, W, C, k, stride, &g);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    if (dx) {
        const size_t wbytes = (size_t)C * k * k * sizeof(float);
        WW_REQUIRE(wbytes <= 64 * 1024, WW_E_UNSUPPORTED, "ww_dwconv_nhwc_bwd: C*k*k = %d weights do not fit the LDS cache", C * k * k);
        WW_REQUIRE((long)B * H * W * C < (1L << 31), WW_E_UNSUPPORTED, "ww_dwconv_nhwc_bwd: tensor too large for 32-bit indices");
        const dim3 grid(egrid((long)B * H * W * ((C + 3) / 4)));
        DwL l;
        if ((((uintptr_t)dy | (uintptr_t)dx) & 15) == 0 && dwl_plan(g, (size_t)g.Ho * g.Wo * 16, 40 * 1024, &l)) {
            const dim3 lg((unsigned)((B + l.nimg - 1) / l.nimg) * l.G_c);
            const size_t lds = ((size_t)k * k * l.CC4 + (size_t)l.nimg * g.Ho * g.Wo * l.CC4) * 16;
            if (k == 3) hipLaunchKernelGGL((k_dwl_conv<3, true, false>), lg, dim3(256), lds, st, dy, w, g, l, dx, (float *)nullptr);
            else hipLaunchKernelGGL((k_dwl_conv<5, true, false>), lg, dim3(256), lds, st, dy, w, g, l, dx, (float *)nullptr);
        } else if ((C & 3) == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
            if (k == 3) hipLaunchKernelGGL(k_dwg_bwd_dx4<3>, grid, dim3(256), wbytes, st, dy, w, g, dx);
            else hipLaunchKernelGGL(k_dwg_bwd_dx4<5>, grid, dim3(256), wbytes, st, dy, w, g, dx);
        } else {
            hipLaunchKernelGGL(k_dwg_bwd_dx, grid, dim3(256), wbytes, st, dy, w, g, dx);
        }
        WW_LAUNCH_CHECK();
    }
    WW_REQUIRE(C <= 1024, WW_E_UNSUPPORTED, "ww_dwconv_nhwc_bwd: C=%d > 1024", C);
    const long P = (long)B * g.Ho * g.Wo;
    float *part = (float *)scratch;
    int chunks;
    DwL l;
    if ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && dwl_plan(g, (size_t)(H * W + g.Ho * g.Wo) * 16, 56 * 1024, &l) &&
        (B + l.nimg - 1) / l.nimg <= NCHUNK_DW) {
        chunks = (B + l.nimg - 1) / l.nimg;
        const size_t lds = (size_t)l.nimg * (H * W + g.Ho * g.Wo) * l.CC4 * 16;
        if (k == 3) hipLaunchKernelGGL(k_dwl_dw<3>, dim3((unsigned)chunks * l.G_c), dim3(256), lds, st, x, dy, g, l, part);
        else hipLaunchKernelGGL(k_dwl_dw<5>, dim3((unsigned)chunks * l.G_c), dim3(256), lds, st, x, dy, g, l, part);
    } else if ((C & 3) == 0) {
        const int R4 = std::max(1, std::min(32, 512 / (C / 4)));      // <= 512 threads: 100 accumulator registers per thread
        WW_REQUIRE(P < (1L << 31) && (long)H * W * C < (1L << 31), WW_E_UNSUPPORTED, "ww_dwconv_nhwc_bwd: tensor too large for 32-bit pixel indices");
        chunks = (int)std::max<long>(1, std::min<long>(NCHUNK_DW, P / (4L * R4)));
        const int T = std::max(1, std::min(k * k, 12288 / (R4 * C)));   // taps per LDS round: <= 48 KB
        if (k == 3)
            hipLaunchKernelGGL(k_dwg_bwd_dw4<3>, dim3(chunks), dim3((C / 4) * R4), (size_t)T * R4 * C * sizeof(float), st, x, dy, g, R4,
                               (uint32_t)((P + chunks - 1) / chunks), T, part);
        else
            hipLaunchKernelGGL(k_dwg_bwd_dw4<5>, dim3(chunks), dim3((C / 4) * R4), (size_t)T * R4 * C * sizeof(float), st, x, dy, g, R4,
                               (uint32_t)((P + chunks - 1) / chunks), T, part);
    } else {
        const int R = rows_r(C);
        chunks = chunks_for(P, C);
        hipLaunchKernelGGL(k_dwg_bwd_dw, dim3(chunks), dim3(C * R), (size_t)R * C * sizeof(float), st, x, dy, g, R,
                           (P + chunks - 1) / chunks, part);
    }
    WW_LAUNCH_CHECK();
    if (ww_defer(ctx, part, dw, (long)C * k * k, chunks, 0)) return WW_OK;      // (the caller keeps `scratch` until the flush)
    return ww_colsum_rows_small(part, chunks, C * k * k, dw, st);
}

extern "C" int ww_pool_hw_fwd(ww_ctx *ctx, const float *x, int B, int HW, int C, float *s, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && s && B >= 1 && HW >= 1 && C >= 1, WW_E_INVALID, "ww_pool_hw_fwd: bad argument");
    hipLaunchKernelGGL(k_pool_fwd, dim3(B * ((C + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, x, B, HW, C, s);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
extern "C" int ww_scale_bc_fwd(ww_ctx *ctx, const float *x, const float *gate, int B, int HW, int C, float *y, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && gate && y && B >= 1 && HW >= 1 && C >= 1, WW_E_INVALID, "ww_scale_bc_fwd: bad argument");
    const long n = (long)B * HW * C;
    if (vec4_ok(n, C, {x, gate, y}))
        hipLaunchKernelGGL(k_scale_fwd4, dim3(egrid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, gate, (uint32_t)(n / 4),
                           (uint32_t)((long)HW * C / 4), C, (float4 *)y);
    else
        hipLaunchKernelGGL(k_scale_fwd, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, x, gate, B, HW, C, y);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
extern "C" int ww_scale_bc_bwd_gate(ww_ctx *ctx, const float *x, const float *dy, int B, int HW, int C, float *dgate,
                                    ww_stream_t stream) {
    WW_REQUIRE(ctx && x && dy && dgate && B >= 1 && HW >= 1 && C >= 1, WW_E_INVALID, "ww_scale_bc_bwd_gate: bad argument");
    hipLaunchKernelGGL(k_scale_bwd_gate, dim3(B * ((C + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, x, dy, B, HW, C, dgate);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
extern "C" int ww_scale_pool_bwd(ww_ctx *ctx, const float *dy, const float *gate, const float *dpool, int B, int HW, int C,
                                 float *dx, ww_stream_t stream) {
    WW_REQUIRE(ctx && dx && (dy || dpool) && (!dy || gate) && B >= 1 && HW >= 1 && C >= 1, WW_E_INVALID, "ww_scale_pool_bwd: bad argument");
    const long n = (long)B * HW * C;
    if (vec4_ok(n, C, {dy, gate, dpool, dx}))
        hipLaunchKernelGGL(k_scale_pool_bwd4, dim3(egrid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4 *)dy, gate, dpool,
                           (uint32_t)(n / 4), (uint32_t)((long)HW * C / 4), HW, C, (float4 *)dx);
    else
        hipLaunchKernelGGL(k_scale_pool_bwd, dim3(egrid(n)), dim3(256), 0, (hipStream_t)stream, dy, gate, dpool, B, HW, C, dx);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
static int make_stem(const char *who, int B, int H, int W, int C, StemG *g) {
    WW_REQUIRE(B >= 1 && H >= 1 && W >= 1, WW_E_INVALID, "%s: bad shape", who);
    WW_REQUIRE(C >= 4 && (C & 3) == 0 && C <= 64, WW_E_UNSUPPORTED, "%s: C=%d (needs a multiple of 4, at most 64)", who, C);
    g->B = B; g->H = H; g->W = W; g->C = C; g->Ho = (H + 1) / 2; g->Wo = (W + 1) / 2;
    WW_REQUIRE((long)B * g->Ho * g->Wo * C < (1L << 31), WW_E_UNSUPPORTED, "%s: tensor too large for 32-bit indices", who);
    return WW_OK;
}
constexpr int STEM_BLOCKS = 512;        // persistent grid = rows of BatchNorm / weight-gradient partials
// Conv2d(1, C, 3, stride 2, padding 1, bias=False) on (B,H,W) one-channel images + training-mode BatchNorm + activation: the direct
// convolution leaves the statistics partials, ww_bn_act_from_partials finishes and applies them.  scratch: ww_nhwc_scratch_bytes(C).
extern "C" int ww_stem3x3s2_bn_act_fwd(ww_ctx *ctx, const float *x, const float *w, int B, int H, int W, int C, const ww_bn_t *bn,
                                       int act, float *y, float *a, float *ss, float *mr, void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && bn && bn->gamma && bn->beta && y && a && ss && mr && scratch, WW_E_INVALID, "ww_stem3x3s2_bn_act_fwd: null argument");
    StemG g;
    int rc = make_stem("ww_stem3x3s2_bn_act_fwd", B, H, W, C, &g);
    if (rc) return rc;
    WW_REQUIRE(bn->training, WW_E_INVALID, "ww_stem3x3s2_bn_act_fwd: training-mode statistics only");
    WW_REQUIRE((((uintptr_t)y | (uintptr_t)scratch) & 15) == 0, WW_E_INVALID, "ww_stem3x3s2_bn_act_fwd: y / scratch must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    const long P = (long)B * g.Ho * g.Wo;
    const int L = 256 / (C / 4), blocks = (int)std::max<long>(1, std::min<long>(STEM_BLOCKS, (P + L - 1) / L));
    hipLaunchKernelGGL(k_stem3x3s2_fwd, dim3(blocks), dim3(256), 0, st, x, w, g, y, (float *)scratch);
    WW_LAUNCH_CHECK();
    return ww_bn_act_from_partials(ctx, y, P, C, bn, act, a, ss, mr, (const float *)scratch, blocks, nullptr, st);
}
// weight gradient of the same convolution: dw (C,1,3,3) = sum over pixels of dy (B,Ho,Wo,C) x input patches
extern "C" int ww_stem3x3s2_bwd_dw(ww_ctx *ctx, const float *x, const float *dy, int B, int H, int W, int C, float *dw, void *scratch,
                                   ww_stream_t stream) {
    WW_REQUIRE(ctx && x && dy && dw && scratch, WW_E_INVALID, "ww_stem3x3s2_bwd_dw: null argument");
    StemG g;
    int rc = make_stem("ww_stem3x3s2_bwd_dw", B, H, W, C, &g);
    if (rc) return rc;
    WW_REQUIRE(((uintptr_t)dy & 15) == 0, WW_E_INVALID, "ww_stem3x3s2_bwd_dw: dy must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    const long P = (long)B * g.Ho * g.Wo;
    const int L = 256 / (C / 4), blocks = (int)std::max<long>(1, std::min<long>(STEM_BLOCKS, (P + L - 1) / L));
    hipLaunchKernelGGL(k_stem3x3s2_dw, dim3(blocks), dim3(256), 0, st, x, dy, g, (float *)scratch);
    WW_LAUNCH_CHECK();
    if (ww_defer(ctx, (const float *)scratch, dw, (long)C * 9, blocks, 0)) return WW_OK;
    return ww_colsum_rows_small((const float *)scratch, blocks, C * 9, dw, st);
}
extern "C" int ww_im2col3x3s2(ww_ctx *ctx, const float *x, int B, int H, int W, float *cols, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && cols && B >= 1 && H >= 1 && W >= 1, WW_E_INVALID, "ww_im2col3x3s2: bad argument");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(k_im2col3x3s2, dim3(egrid((long)B * Ho * Wo * 9)), dim3(256), 0, (hipStream_t)stream, x, B, H, W, Ho, Wo, cols);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
extern "C" int ww_add_f32(ww_ctx *ctx, const float *a, const float *b, size_t n, float *y, ww_stream_t stream) {
    WW_REQUIRE(ctx && a && b && y, WW_E_INVALID, "ww_add_f32: null argument");
    if (n == 0) return WW_OK;
    if (vec4_ok((long)n, 4, {a, b, y}))
        hipLaunchKernelGGL(k_add4, dim3(egrid((long)n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4 *)a, (const float4 *)b,
                           (uint32_t)(n / 4), (float4 *)y);
    else
        hipLaunchKernelGGL(k_add, dim3(egrid((long)n)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, y);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
