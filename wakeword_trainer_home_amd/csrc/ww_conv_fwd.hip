// Forward kernels of the depthwise-separable conv stack (channels-last, C = 64).
// Activation tensors y_l are stored as fp32 (parity mode) or bf16 (ww_act.h); arithmetic is fp32.
//
// Every conv kernel (a) applies the PRODUCER layer's BatchNorm+ReLU while loading (scale/shift
// per channel), (b) writes its own PRE-BatchNorm output once, and (c) accumulates the
// per-channel sum / sum-of-squares of that output in registers; block partials go to a slab
// that a 1-block finalize kernel sums in double (ww_ctx.hip).  Normalised activations never
// touch HBM: per layer the traffic is one read + one write of the (B,H,W,64) tensor.
//
// Thread maps:
//   stem / depthwise / GAP : 32 lanes x float2 = one pixel's 64 channels (256 B, coalesced);
//                            a wavefront holds 2 pixels; channel statistics stay per-lane.
//   depthwise              : a 32-lane group walks DOWN a 4-column strip keeping a 3x6 pixel
//                            window in registers, so each input pixel is fetched 1.5x from L1/L2
//                            and once from HBM; no LDS, no barriers in the main loop.
//   pointwise              : M=B*H*W rows x K=64 x N=64 GEMM on v_mfma_f32_32x32x2_f32 (exact
//                            fp32, same rate as the fp32 VALU but no LDS broadcast traffic);
//                            128-pixel tiles staged coalesced -> BN+ReLU in registers -> LDS
//                            (272-byte padded rows) -> conflict-free ds_read_b128 fragments; the
//                            64x64 weight lives in 64 VGPRs per lane for the whole kernel.
#include "ww_internal.h"
#include "ww_act.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// BatchNorm + ReLU on load.  ReLU propagates NaN like torch.relu (v_max would silently turn it into 0).
__device__ __forceinline__ float bnrelu(float y, float s, float t) {
    const float z = fmaf(y, s, t);
    return z < 0.f ? 0.f : z;
}

// block-level reduction of per-lane channel statistics -> partial row [sum(64) | sumsq(64)]
// lanes: cl = tid & 31 owns channels 2cl, 2cl+1 ; 8 pixel slots per 256-thread block
__device__ __forceinline__ void reduce_stats_slots(float s0, float s1, float q0, float q1, float *sh /*8*128*/,
                                                   float *__restrict__ partial_row) {
    const int tid = threadIdx.x, slot = tid >> 5, cl = tid & 31;
    __syncthreads();
    sh[slot * 128 + 2 * cl] = s0;
    sh[slot * 128 + 2 * cl + 1] = s1;
    sh[slot * 128 + 64 + 2 * cl] = q0;
    sh[slot * 128 + 64 + 2 * cl + 1] = q1;
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) t += sh[s * 128 + tid];
        partial_row[tid] = t;
    }
}

// ------------------------------------------------------------------------------------- stem
// x (B,Hin,Win) -> y (B,Ho,Wo,64) ; 3x3 stride 2 pad 1 ; TWO (b,oh) output rows per iteration.
// The five input rows are staged zero-padded in LDS (one barrier pair per two rows).  Thread map: 8 lanes = one pixel, a
// lane owns 8 consecutive channels -- 9 LDS broadcast reads feed 72 FMAs, and the result leaves as ONE 16-byte (16-bit
// storage) or two 16-byte (fp32) stores per lane, 1 KB per wave-instruction.  (r01: 2 channels per lane, 4-byte stores, one
// row per barrier pair -- 39.7 us for 100 MB of output, store-issue-bound.)  The nine taps are accumulated in the fixed
// order kh-major / kw-minor: k_stem_bwd recomputes this tensor with the same chain.
template <typename T>
__global__ __launch_bounds__(256) void k_stem_fwd(const float *__restrict__ x, const float *__restrict__ w, int B,
                                                  int Hin, int Win, int Ho, int Wo, T *__restrict__ y,
                                                  float *__restrict__ partials, int rev) {
    __shared__ float sh[32 * 128];
    extern __shared__ float xs[];            // [5][Win + 2] : xs[r][iw + 1], zero outside the image
    const int tid = threadIdx.x, px = tid >> 3, c8 = tid & 7;
    const int ld = Win + 2;
    float wr[8][9];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[j][t] = w[(8 * c8 + j) * 9 + t];
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    const long npairs = (long)B * ((Ho + 1) / 2);
    // the five input rows of an item are fetched into registers one item AHEAD (their HBM latency hides behind the
    // previous item's FMAs) and written to LDS at the top of the item: 5 * ld <= 4 * 256 values
    float pre[4];
    auto fetch = [&](long item) {
        if (rev) item = npairs - 1 - item;
        const int fb_ = (int)(item / ((Ho + 1) / 2)), foh = 2 * (int)(item - (long)fb_ * ((Ho + 1) / 2));
        const float *xb = x + (size_t)fb_ * Hin * Win;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k;
            const int r = i / ld, c = i - r * ld;
            const int ih = 2 * foh - 1 + r, iw = c - 1;
            pre[k] = (i < 5 * ld && ih >= 0 && ih < Hin && iw >= 0 && iw < Win) ? xb[(size_t)ih * Win + iw] : 0.f;
        }
    };
    const bool ahead = 5 * ld <= 4 * 256;    // wider inputs (T > 202 frames) are staged in place
    if (ahead && (long)blockIdx.x < npairs) fetch(blockIdx.x);
    for (long pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const long prm = rev ? npairs - 1 - pr : pr;
        const int b = (int)(prm / ((Ho + 1) / 2)), oh0 = 2 * (int)(prm - (long)b * ((Ho + 1) / 2));
        __syncthreads();                     // previous pair's readers are done
        if (ahead) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tid + 256 * k < 5 * ld) xs[tid + 256 * k] = pre[k];
        } else {
            const float *xb = x + (size_t)b * Hin * Win;
            for (int i = tid; i < 5 * ld; i += 256) {
                const int r = i / ld, c = i - r * ld;
                const int ih = 2 * oh0 - 1 + r, iw = c - 1;
                xs[i] = (ih >= 0 && ih < Hin && iw >= 0 && iw < Win) ? xb[(size_t)ih * Win + iw] : 0.f;
            }
        }
        __syncthreads();
        if (ahead && pr + gridDim.x < npairs) fetch(pr + gridDim.x);
        const int nrow = min(2, Ho - oh0);
        for (int it = px; it < nrow * Wo; it += 32) {
            const int rr = it >= Wo ? 1 : 0, ow = it - rr * Wo;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float v = xs[(2 * rr + kh) * ld + 2 * ow + kw];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wr[j][kh * 3 + kw], acc[j]);
                }
            T *dst = y + (((size_t)b * Ho + oh0 + rr) * Wo + ow) * 64 + 8 * c8;
            float4 o0 = Act<T>::round4(make_float4(acc[0], acc[1], acc[2], acc[3]));
            float4 o1 = Act<T>::round4(make_float4(acc[4], acc[5], acc[6], acc[7]));
            Act<T>::st8(dst, o0, o1);
            s[0] += o0.x; s[1] += o0.y; s[2] += o0.z; s[3] += o0.w; s[4] += o1.x; s[5] += o1.y; s[6] += o1.z; s[7] += o1.w;
            q[0] = fmaf(o0.x, o0.x, q[0]); q[1] = fmaf(o0.y, o0.y, q[1]); q[2] = fmaf(o0.z, o0.z, q[2]); q[3] = fmaf(o0.w, o0.w, q[3]);
            q[4] = fmaf(o1.x, o1.x, q[4]); q[5] = fmaf(o1.y, o1.y, q[5]); q[6] = fmaf(o1.z, o1.z, q[6]); q[7] = fmaf(o1.w, o1.w, q[7]);
        }
    }
    if (partials) {                          // 32 pixel lanes x [sum(64) | sumsq(64)] -> one partial row, fixed order
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sh[px * 128 + 8 * c8 + j] = s[j];
            sh[px * 128 + 64 + 8 * c8 + j] = q[j];
        }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) t += sh[i * 128 + tid];
            partials[(size_t)blockIdx.x * 128 + tid] = t;
        }
    }
}

// -------------------------------------------------------------------------------- depthwise
struct DwGeom {
    int B, H, W, ncs, nseg, hs_len;  // column strips per row, row segments per image, rows per segment
    long items;
};

constexpr int DW_HS = 20;   // max rows per strip segment (compile-time so the row loop fully unrolls)

// raw (pre-BatchNorm) row of 6 pixels around the strip; addresses are clamped so the loads are branch-free
template <typename T>
__device__ __forceinline__ void dw_issue_row(const T *__restrict__ img, int h, int w0, int H, int W, int cl,
                                             typename Act<T>::raw2 (&raw)[6]) {
    const int hh = min(max(h, 0), H - 1);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int wc = min(max(w0 - 1 + i, 0), W - 1);
        raw[i] = Act<T>::ldraw2(img + (uint32_t)((hh * W + wc) * 64 + 2 * cl));   // one image < 2^32 elements: 32-bit offsets
    }
}
// BatchNorm+ReLU of a raw row; positions outside the image are the conv's zero padding (in activation space)
template <typename T>
__device__ __forceinline__ void dw_finish_row(const typename Act<T>::raw2 (&raw)[6], int h, int w0, int H, int W,
                                              float2 sc, float2 sf, float2 (&r)[6]) {
    const bool hv = (h >= 0) && (h < H);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int wc = w0 - 1 + i;
        const bool ok = hv && wc >= 0 && wc < W;
        const float2 v = Act<T>::cvt2(raw[i]);
        r[i].x = ok ? bnrelu(v.x, sc.x, sf.x) : 0.f;
        r[i].y = ok ? bnrelu(v.y, sc.y, sf.y) : 0.f;
    }
}

template <typename T>
__device__ __forceinline__ void dw_out_row(T *__restrict__ oimg, int h, int w0, int W, int cl,
                                           const float2 (&top)[6], const float2 (&mid)[6], const float2 (&bot)[6],
                                           const float (&wa)[9], const float (&wb)[9], float &s0, float &s1, float &q0,
                                           float &q1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            a0 = fmaf(top[i + kw].x, wa[kw], a0);
            a1 = fmaf(top[i + kw].y, wb[kw], a1);
            a0 = fmaf(mid[i + kw].x, wa[3 + kw], a0);
            a1 = fmaf(mid[i + kw].y, wb[3 + kw], a1);
            a0 = fmaf(bot[i + kw].x, wa[6 + kw], a0);
            a1 = fmaf(bot[i + kw].y, wb[6 + kw], a1);
        }
        if (w0 + i < W) {
            const float2 o = Act<T>::round2(make_float2(a0, a1));
            Act<T>::st2(oimg + (uint32_t)((h * W + w0 + i) * 64 + 2 * cl), o);
            s0 += o.x; s1 += o.y;
            q0 = fmaf(o.x, o.x, q0); q1 = fmaf(o.y, o.y, q1);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_dw_fwd(const T *__restrict__ yin, const float *__restrict__ ss,
                                                const float *__restrict__ w, DwGeom g, T *__restrict__ y,
                                                float *__restrict__ partials) {
    __shared__ float sh[8 * 128];
    typedef typename Act<T>::raw2 raw2;
    const int tid = threadIdx.x, slot = tid >> 5, cl = tid & 31;
    float wa[9], wb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wa[t] = w[(2 * cl) * 9 + t];
        wb[t] = w[(2 * cl + 1) * 9 + t];
    }
    const float2 sc = *reinterpret_cast<const float2 *>(ss + 2 * cl);
    const float2 sf = *reinterpret_cast<const float2 *>(ss + 64 + 2 * cl);
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    const size_t img_stride = (size_t)g.H * g.W * 64;
    for (long item = (long)blockIdx.x * 8 + slot; item < g.items; item += (long)gridDim.x * 8) {
        const int cs = (int)(item % g.ncs);
        const long t = item / g.ncs;
        const int seg = (int)(t % g.nseg), b = (int)(t / g.nseg);
        const int w0 = cs * 4, hs = seg * g.hs_len;
        const int he = min(g.H, hs + g.hs_len);
        const T *img = yin + (size_t)b * img_stride;
        T *oimg = y + (size_t)b * img_stride;
        // rows[(i)%3] = row h-1, rows[(i+1)%3] = row h, rows[(i+2)%3] = row h+1 ; `ahead` = raw row h+2 in flight
        float2 rows[3][6];
        raw2 raw[6], ahead[6];
        dw_issue_row<T>(img, hs - 1, w0, g.H, g.W, cl, raw);
        dw_finish_row<T>(raw, hs - 1, w0, g.H, g.W, sc, sf, rows[0]);
        dw_issue_row<T>(img, hs, w0, g.H, g.W, cl, raw);
        dw_finish_row<T>(raw, hs, w0, g.H, g.W, sc, sf, rows[1]);
        dw_issue_row<T>(img, hs + 1, w0, g.H, g.W, cl, ahead);
#pragma unroll 1
        for (int i0 = 0; i0 < DW_HS; i0 += 3) {
#pragma unroll
          for (int ii = 0; ii < 3; ++ii) {
            const int i = i0 + ii;           // i % 3 == ii: the row rotation stays compile-time
            const int h = hs + i;
            if (h < he) {
#pragma unroll
                for (int c = 0; c < 6; ++c) raw[c] = ahead[c];
                if (h + 1 < he) dw_issue_row<T>(img, h + 2, w0, g.H, g.W, cl, ahead);   // one row ahead of its use
                dw_finish_row<T>(raw, h + 1, w0, g.H, g.W, sc, sf, rows[(ii + 2) % 3]);
                dw_out_row<T>(oimg, h, w0, g.W, cl, rows[ii % 3], rows[(ii + 1) % 3], rows[(ii + 2) % 3], wa, wb, s0, s1,
                              q0, q1);
            }
          }
        }
    }
    if (partials) reduce_stats_slots(s0, s1, q0, q1, sh, partials + (size_t)blockIdx.x * 128);
}

// -------------------------------------------------------------------------------- pointwise
constexpr int PW_TILE = 128;    // pixels per workgroup tile (32 per wavefront)
constexpr int PW_LD = 68;       // padded LDS row (floats): 272 B -> conflict-free ds_read_b128 / ds_write_b128

// LDS (dynamic): tile[128][68] (+ otile[128][68] for the bf16 output staging)
template <typename T>
__global__ __launch_bounds__(256) void k_pw_fwd(const T *__restrict__ yin, const float *__restrict__ ss,
                                                const float *__restrict__ w, long M, T *__restrict__ y,
                                                float *__restrict__ partials) {
    extern __shared__ __align__(16) float pw_lds[];
    float *tile = pw_lds;
    float *otile = pw_lds + PW_TILE * PW_LD;     // carved for bf16 only
    typedef typename Act<T>::raw4 raw4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // wave (rh, n) owns pixels [64rh, 64rh+64) x output channels [32n, 32n+32) of the tile
    const int rh = wv >> 1, n = wv & 1;
    // B operand of k-step s: W_kn[k = 32h + s][j = 32n + r] = w[j][k]  (the K order is permuted
    // identically for A and B: lane half h covers input channels 32h .. 32h+31)
    float wreg[32];
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
        const float4 v = *reinterpret_cast<const float4 *>(w + (size_t)(32 * n + r) * 64 + 32 * h + 4 * s4);
        wreg[4 * s4] = v.x; wreg[4 * s4 + 1] = v.y; wreg[4 * s4 + 2] = v.z; wreg[4 * s4 + 3] = v.w;
    }
    const int c4 = tid & 15;
    const float4 sc = *reinterpret_cast<const float4 *>(ss + 4 * c4);
    const float4 sf = *reinterpret_cast<const float4 *>(ss + 64 + 4 * c4);
    float s1 = 0.f, s2 = 0.f;
    const long ntiles = (M + PW_TILE - 1) / PW_TILE;
    // software pipeline: the NEXT tile's global loads are issued right after the LDS barrier, so they are in
    // flight while this tile's MFMAs run (co-resident workgroups otherwise fall into lockstep: all load, then
    // all compute, and time = HBM + MFMA instead of max(HBM, MFMA))
    raw4 raw[8];
    auto issue = [&](long ti) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long p = ti * PW_TILE + (tid >> 4) + 16 * i;
            p = p < M ? p : M - 1;                       // clamped, branch-free; masked when consumed
            raw[i] = Act<T>::ldraw4(yin + (size_t)p * 64 + 4 * c4);
        }
    };
    if ((long)blockIdx.x < ntiles) issue(blockIdx.x);
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const long p0 = ti * PW_TILE;
        // BN+ReLU in registers, padded LDS rows
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 4) + 16 * i;
            const bool ok = p0 + row < M;
            const float4 v = Act<T>::cvt4(raw[i]);
            float4 a;
            a.x = ok ? bnrelu(v.x, sc.x, sf.x) : 0.f; a.y = ok ? bnrelu(v.y, sc.y, sf.y) : 0.f;
            a.z = ok ? bnrelu(v.z, sc.z, sf.z) : 0.f; a.w = ok ? bnrelu(v.w, sc.w, sf.w) : 0.f;
            *reinterpret_cast<float4 *>(tile + row * PW_LD + 4 * c4) = a;
        }
        __syncthreads();
        if (ti + gridDim.x < ntiles) issue(ti + gridDim.x);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int rbase = 64 * rh + 32 * t;
            float a[32];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(tile + (rbase + r) * PW_LD + 32 * h + 4 * j);
                a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
            }
            floatx16 acc = {0.f};
#pragma unroll
            for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[s], acc, 0, 0, 0);
            // D layout: col = lane&31 (output channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int prow = rbase + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const float v = Act<T>::round1(acc[reg]);
                if constexpr (Act<T>::is_f32) {
                    if (p0 + prow < M) y[(size_t)(p0 + prow) * 64 + 32 * n + r] = v;
                } else {
                    otile[prow * PW_LD + 32 * n + r] = v;     // staged: packed 8-byte stores below
                }
                s1 += v;                      // rows past M are exact zeros
                s2 = fmaf(v, v, s2);
            }
        }
        __syncthreads();
        if constexpr (!Act<T>::is_f32) {
            // otile is rewritten only after the next tile's barrier, i.e. after every thread finished this pass
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (tid >> 4) + 16 * i;
                if (p0 + row < M)
                    Act<T>::st4(y + (size_t)(p0 + row) * 64 + 4 * c4,
                                *reinterpret_cast<const float4 *>(otile + row * PW_LD + 4 * c4));
            }
        }
    }
    if (partials) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        __syncthreads();
        float *sh = tile;  // [wave][kind][32]
        if (h == 0) {
            sh[wv * 64 + r] = s1;
            sh[wv * 64 + 32 + r] = s2;
        }
        __syncthreads();
        if (tid < 128) {
            const int kind = tid >> 6, c = tid & 63, nn = c >> 5, rr = c & 31;
            partials[(size_t)blockIdx.x * 128 + tid] = sh[nn * 64 + kind * 32 + rr] + sh[(2 + nn) * 64 + kind * 32 + rr];
        }
    }
}

// ---- bf16 mode: the same GEMM on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate, so the kernel is purely
// HBM-bound).  LDS tile holds relu(bn(y_in)) as bf16 with 144-byte rows (conflict-free ds_read_b128 fragments);
// lane (r,h) of k-step t reads channels 16t+8h .. +7 of pixel r.  Weights are rounded to bf16 once per kernel
// (what autocast does to conv weights); accumulation is fp32.
constexpr int PWH_LD = 72;      // 16-bit elements per LDS row (64 + 8 pad = 144 B)

// H = H | ww_f16 (the same kernel on v_mfma_f32_32x32x16_bf16 / _f16)
template <typename H>
__global__ __launch_bounds__(256) void k_pw_fwd_bf16(const H *__restrict__ yin, const float *__restrict__ ss,
                                                     const float *__restrict__ w, long M, H *__restrict__ y,
                                                     float *__restrict__ partials, int rev) {
    extern __shared__ __align__(16) unsigned char pwh_lds[];
    H *atile = reinterpret_cast<H *>(pwh_lds);                       // [128][72] bf16
    H *otile = atile + PW_TILE * PWH_LD;                                   // [128][72] bf16: the output as it is stored
    // (36.9 KB together: 4 workgroups per CU, what the 96 VGPRs allow; an fp32 staging tile capped it at 3)
    typedef Act<H> A16;
    typedef typename H16<H>::x8 bf16x8;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rh = wv >> 1, n = wv & 1;
    // B operand of k-step t: B[k = 16t + 8h + j][col r] = w[32n + r][16t + 8h + j]
    bf16x8 wb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w[(size_t)(32 * n + r) * 64 + 16 * t + 8 * h + j];
        wb[t] = ww_pack8<H>(v);
    }
    const int c4 = tid & 15;
    const float4 sc = *reinterpret_cast<const float4 *>(ss + 4 * c4);
    const float4 sf = *reinterpret_cast<const float4 *>(ss + 64 + 4 * c4);
    float s1 = 0.f, s2 = 0.f;
    const long ntiles = (M + PW_TILE - 1) / PW_TILE;
    typename A16::raw4 raw[8];
    auto issue = [&](long ti) {
        if (rev) ti = ntiles - 1 - ti;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long p = ti * PW_TILE + (tid >> 4) + 16 * i;
            p = p < M ? p : M - 1;
            raw[i] = A16::ldraw4_nt(yin + (size_t)p * 64 + 4 * c4);
        }
    };
    if ((long)blockIdx.x < ntiles) issue(blockIdx.x);
    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const long p0 = (rev ? ntiles - 1 - ti : ti) * PW_TILE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 4) + 16 * i;
            const bool ok = p0 + row < M;
            const float4 v = A16::cvt4(raw[i]);
            const float a0 = ok ? bnrelu(v.x, sc.x, sf.x) : 0.f, a1 = ok ? bnrelu(v.y, sc.y, sf.y) : 0.f;
            const float a2 = ok ? bnrelu(v.z, sc.z, sf.z) : 0.f, a3 = ok ? bnrelu(v.w, sc.w, sf.w) : 0.f;
            *reinterpret_cast<uint2 *>(atile + row * PWH_LD + 4 * c4) = make_uint2(A16::pack2(a0, a1), A16::pack2(a2, a3));
        }
        __syncthreads();
        if (ti + gridDim.x < ntiles) issue(ti + gridDim.x);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int rbase = 64 * rh + 32 * t2;
            floatx16 acc = {0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(atile + (rbase + r) * PWH_LD + 16 * t + 8 * h);
                acc = H16<H>::mfma32(a, wb[t], acc);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int prow = rbase + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const float v = A16::round1(acc[reg]);
                otile[prow * PWH_LD + 32 * n + r] = (H)v;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 4) + 16 * i;
            if (p0 + row < M)
                *reinterpret_cast<uint2 *>(y + (size_t)(p0 + row) * 64 + 4 * c4) =
                    *reinterpret_cast<const uint2 *>(otile + row * PWH_LD + 4 * c4);
        }
    }
    if (partials) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        __syncthreads();
        float *sh = reinterpret_cast<float *>(otile);  // [wave][kind][32]
        if (h == 0) {
            sh[wv * 64 + r] = s1;
            sh[wv * 64 + 32 + r] = s2;
        }
        __syncthreads();
        if (tid < 128) {
            const int kind = tid >> 6, c = tid & 63, nn = c >> 5, rr = c & 31;
            partials[(size_t)blockIdx.x * 128 + tid] = sh[nn * 64 + kind * 32 + rr] + sh[(2 + nn) * 64 + kind * 32 + rr];
        }
    }
}

// -------------------------------------------------------------------------------------- GAP
// pool[b] = [sum relu(z) (64) | sum_{z>0} yhat (64) | count_{z>0} (64)]
// one 1024-thread workgroup per clip: 64 pixel slots x 16 lanes of FOUR channels (8-byte loads of a 16-bit tensor), a lane's
// pixels in batches of eight unconditional (clamped) loads -- with two channels per lane and four loads per loop trip the
// kernel was a chain of twelve dependent round trips per thread (22 us for 100 MB)
template <typename T>
__global__ __launch_bounds__(1024) void k_gap_fwd(const T *__restrict__ y, const float *__restrict__ ss,
                                                  const float *__restrict__ mr, int HW, float *__restrict__ pool) {
    __shared__ __align__(16) float sh[64 * 192];
    const int tid = threadIdx.x, slot = tid >> 4, cl = tid & 15, b = blockIdx.x;
    const float4 sc = *reinterpret_cast<const float4 *>(ss + 4 * cl);
    const float4 sf = *reinterpret_cast<const float4 *>(ss + 64 + 4 * cl);
    const float4 mu = *reinterpret_cast<const float4 *>(mr + 4 * cl);
    const float4 rs = *reinterpret_cast<const float4 *>(mr + 64 + 4 * cl);
    const T *yb = y + (size_t)b * HW * 64 + 4 * cl;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), h = a, c = a;
    auto one = [](float v, float scv, float sfv, float muv, float rsv, float &av, float &hv, float &cv) {
        const float z = fmaf(v, scv, sfv);
        av += z < 0.f ? 0.f : z;                         // NaN-propagating ReLU, as torch
        if (z > 0.f) { hv += (v - muv) * rsv; cv += 1.f; }
    };
    for (int p0 = slot; p0 < HW; p0 += 8 * 64) {
        typename Act<T>::raw4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = Act<T>::ldraw4_nt(yb + (size_t)min(p0 + 64 * u, HW - 1) * 64);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p0 + 64 * u < HW) {
                const float4 v = Act<T>::cvt4(r[u]);
                one(v.x, sc.x, sf.x, mu.x, rs.x, a.x, h.x, c.x);
                one(v.y, sc.y, sf.y, mu.y, rs.y, a.y, h.y, c.y);
                one(v.z, sc.z, sf.z, mu.z, rs.z, a.z, h.z, c.z);
                one(v.w, sc.w, sf.w, mu.w, rs.w, a.w, h.w, c.w);
            }
    }
    *reinterpret_cast<float4 *>(sh + slot * 192 + 4 * cl) = a;
    *reinterpret_cast<float4 *>(sh + slot * 192 + 64 + 4 * cl) = h;
    *reinterpret_cast<float4 *>(sh + slot * 192 + 128 + 4 * cl) = c;
    __syncthreads();
    if (tid < 192) {
        float t = 0.f;
#pragma unroll 8
        for (int s = 0; s < 64; ++s) t += sh[s * 192 + tid];
        pool[(size_t)b * 192 + tid] = t;
    }
}

// ------------------------------------------------------------------------------------- head
// one wavefront per sample: lane = channel
__global__ __launch_bounds__(256) void k_head_fwd(const float *__restrict__ pool, int B, int HW,
                                                  const float *__restrict__ fc_w, const float *__restrict__ fc_b,
                                                  float drop_scale, uint64_t drop_thresh, int use_dropout,
                                                  uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi,
                                                  uint64_t sample_offset, float *__restrict__ pd,
                                                  float *__restrict__ logits, const ww_step_ctl *__restrict__ ctl) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    ww_step_resolve(ctl, step_lo, step_hi, step_lo, step_hi);
    float v = pool[(size_t)b * 192 + lane] / (float)HW;
    if (use_dropout) {
        uint32_t rr[4];
        ww_philox(step_lo, step_hi, (uint32_t)(sample_offset + (uint64_t)b), (WW_TAG_DROPOUT << 24) | (uint32_t)(lane >> 2),
                  seed_lo, seed_hi, rr);
        const uint32_t d = rr[lane & 3];
        v = ((uint64_t)d >= drop_thresh) ? v * drop_scale : 0.f;
    }
    pd[(size_t)b * 64 + lane] = v;
    float l0 = v * fc_w[lane], l1 = v * fc_w[64 + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l0 += __shfl_xor(l0, o);
        l1 += __shfl_xor(l1, o);
    }
    if (lane == 0) {
        logits[(size_t)b * 2] = l0 + fc_b[0];
        logits[(size_t)b * 2 + 1] = l1 + fc_b[1];
    }
}

int finish_bn(ww_ctx *ctx, const float *partials, int rows, double count, const ww_bn_t *bn, float *ss_out,
              float *mr_out, hipStream_t st) {
    ww_prof_scope ps_(ctx, WW_K_FINALIZE, st);
    if (bn->training) return ww_launch_bn_fwd_finalize(partials, rows, count, bn, ss_out, mr_out, st);
    return ww_launch_bn_eval_ss(bn, ss_out, mr_out, st);
}

int check_bn(const char *who, const ww_bn_t *bn, const float *ss_out, const float *mr_out, const void *scratch) {
    WW_REQUIRE(bn && bn->gamma && bn->beta && ss_out && mr_out, WW_E_INVALID, "%s: null BatchNorm argument", who);
    WW_REQUIRE(bn->training ? scratch != nullptr : (bn->running_mean && bn->running_var), WW_E_INVALID,
               "%s: training needs scratch, eval needs running statistics", who);
    return WW_OK;
}

int check_act(const char *who, int act_dtype) {
    WW_REQUIRE(act_dtype == WW_ACT_F32 || act_dtype == WW_ACT_BF16 || act_dtype == WW_ACT_F16, WW_E_INVALID, "%s: unknown act_dtype %d", who,
               act_dtype);
    return WW_OK;
}

template <typename T>
int launch_stem_fwd(ww_ctx *ctx, const float *x, const float *w, int B, int Hin, int Win, void *y, float *partials,
                    int *grid_out, hipStream_t st) {
    const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2;
    const long nrows = (long)B * ((Ho + 1) / 2);          // work items: pairs of output rows
    const size_t smem = (size_t)5 * (Win + 2) * sizeof(float);
    const int grid = ww_occupancy_grid((const void *)k_stem_fwd<T>, 256, smem, nrows, WW_MAX_PARTIALS);
    {
        ww_prof_scope ps_(ctx, WW_K_STEM_FWD, st);
        static const int rev = ww_env_int("WW_STEM_FWD_REV", 1);
        hipLaunchKernelGGL(k_stem_fwd<T>, dim3(grid), dim3(256), smem, st, x, w, B, Hin, Win, Ho, Wo, (T *)y, partials, rev);
    }
    WW_LAUNCH_CHECK();
    *grid_out = grid;
    return WW_OK;
}

template <typename T>
int launch_dw_fwd(ww_ctx *ctx, const void *y_in, const float *ss_in, const float *w, const DwGeom &g, void *y,
                  float *partials, int *grid_out, hipStream_t st) {
    const long nblk = (g.items + 7) / 8;
    const int grid = ww_occupancy_grid((const void *)k_dw_fwd<T>, 256, 0, nblk, WW_MAX_PARTIALS);
    {
        ww_prof_scope ps_(ctx, WW_K_DW_FWD, st);
        hipLaunchKernelGGL(k_dw_fwd<T>, dim3(grid), dim3(256), 0, st, (const T *)y_in, ss_in, w, g, (T *)y, partials);
    }
    WW_LAUNCH_CHECK();
    *grid_out = grid;
    return WW_OK;
}

template <typename H>
int launch_pw_fwd_bf16(ww_ctx *ctx, const void *y_in, const float *ss_in, const float *w, long M, void *y,
                       float *partials, int *grid_out, hipStream_t st) {
    const long ntiles = (M + PW_TILE - 1) / PW_TILE;
    const size_t smem = (size_t)2 * PW_TILE * PWH_LD * 2;
    const int grid = ww_occupancy_grid((const void *)k_pw_fwd_bf16<H>, 256, smem, ntiles, WW_MAX_PARTIALS);
    {
        ww_prof_scope ps_(ctx, WW_K_PW_FWD, st);
        static const int rev = ww_env_int("WW_PW_FWD_REV", 1);
        hipLaunchKernelGGL(k_pw_fwd_bf16<H>, dim3(grid), dim3(256), smem, st, (const H *)y_in, ss_in, w, M, (H *)y, partials, rev);
    }
    WW_LAUNCH_CHECK();
    *grid_out = grid;
    return WW_OK;
}

template <typename T>
int launch_pw_fwd(ww_ctx *ctx, const void *y_in, const float *ss_in, const float *w, long M, void *y, float *partials,
                  int *grid_out, hipStream_t st) {
    const long ntiles = (M + PW_TILE - 1) / PW_TILE;
    const size_t smem = (size_t)(Act<T>::is_f32 ? 1 : 2) * PW_TILE * PW_LD * sizeof(float);
    WW_HIP(hipFuncSetAttribute((const void *)k_pw_fwd<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = ww_occupancy_grid((const void *)k_pw_fwd<T>, 256, smem, ntiles, WW_MAX_PARTIALS);
    {
        ww_prof_scope ps_(ctx, WW_K_PW_FWD, st);
        hipLaunchKernelGGL(k_pw_fwd<T>, dim3(grid), dim3(256), smem, st, (const T *)y_in, ss_in, w, M, (T *)y, partials);
    }
    WW_LAUNCH_CHECK();
    *grid_out = grid;
    return WW_OK;
}

}  // namespace

extern "C" int ww_conv_stem_fwd(ww_ctx *ctx, int act_dtype, const float *x, const float *w, int B, int Hin, int Win,
                                void *y, const ww_bn_t *bn, float *ss_out, float *mr_out, void *scratch,
                                ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && y, WW_E_INVALID, "ww_conv_stem_fwd: null argument");
    WW_REQUIRE(B >= 1 && Hin >= 1 && Win >= 1, WW_E_INVALID, "ww_conv_stem_fwd: bad shape (%d,%d,%d)", B, Hin, Win);
    int rc = check_bn("ww_conv_stem_fwd", bn, ss_out, mr_out, scratch);
    if (rc || (rc = check_act("ww_conv_stem_fwd", act_dtype))) return rc;
    hipStream_t st = (hipStream_t)stream;
    float *partials = bn->training ? (float *)scratch : nullptr;
    int grid = 0;
    rc = act_dtype == WW_ACT_BF16  ? launch_stem_fwd<ww_bf16>(ctx, x, w, B, Hin, Win, y, partials, &grid, st)
         : act_dtype == WW_ACT_F16 ? launch_stem_fwd<ww_f16>(ctx, x, w, B, Hin, Win, y, partials, &grid, st)
                                   : launch_stem_fwd<float>(ctx, x, w, B, Hin, Win, y, partials, &grid, st);
    if (rc) return rc;
    return finish_bn(ctx, partials, grid, (double)B * ((Hin + 1) / 2) * ((Win + 1) / 2), bn, ss_out, mr_out, st);
}

extern "C" int ww_dwconv3x3_fwd(ww_ctx *ctx, int act_dtype, const void *y_in, const float *ss_in, const float *w,
                                int B, int H, int W, void *y, const ww_bn_t *bn, float *ss_out, float *mr_out,
                                void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && y_in && ss_in && w && y, WW_E_INVALID, "ww_dwconv3x3_fwd: null argument");
    WW_REQUIRE(B >= 1 && H >= 1 && W >= 1, WW_E_INVALID, "ww_dwconv3x3_fwd: bad shape (%d,%d,%d)", B, H, W);
    WW_REQUIRE((long)H * W * 64 < (1L << 31), WW_E_UNSUPPORTED, "ww_dwconv3x3_fwd: one image of %d x %d x 64 exceeds 2^31 elements", H, W);
    int rc = check_bn("ww_dwconv3x3_fwd", bn, ss_out, mr_out, scratch);
    if (rc || (rc = check_act("ww_dwconv3x3_fwd", act_dtype))) return rc;
    DwGeom g;
    g.B = B; g.H = H; g.W = W;
    g.ncs = (W + 3) / 4;
    g.nseg = (H + DW_HS - 1) / DW_HS;
    g.hs_len = (H + g.nseg - 1) / g.nseg;      // <= DW_HS
    g.items = (long)B * g.nseg * g.ncs;
    hipStream_t st = (hipStream_t)stream;
    float *partials = bn->training ? (float *)scratch : nullptr;
    int grid = 0;
    rc = act_dtype == WW_ACT_BF16  ? launch_dw_fwd<ww_bf16>(ctx, y_in, ss_in, w, g, y, partials, &grid, st)
         : act_dtype == WW_ACT_F16 ? launch_dw_fwd<ww_f16>(ctx, y_in, ss_in, w, g, y, partials, &grid, st)
                                   : launch_dw_fwd<float>(ctx, y_in, ss_in, w, g, y, partials, &grid, st);
    if (rc) return rc;
    return finish_bn(ctx, partials, grid, (double)B * H * W, bn, ss_out, mr_out, st);
}

extern "C" int ww_pwconv1x1_fwd(ww_ctx *ctx, int act_dtype, const void *y_in, const float *ss_in, const float *w,
                                int B, int H, int W, void *y, const ww_bn_t *bn, float *ss_out, float *mr_out,
                                void *scratch, ww_stream_t stream) {
    WW_REQUIRE(ctx && y_in && ss_in && w && y, WW_E_INVALID, "ww_pwconv1x1_fwd: null argument");
    WW_REQUIRE(B >= 1 && H >= 1 && W >= 1, WW_E_INVALID, "ww_pwconv1x1_fwd: bad shape (%d,%d,%d)", B, H, W);
    int rc = check_bn("ww_pwconv1x1_fwd", bn, ss_out, mr_out, scratch);
    if (rc || (rc = check_act("ww_pwconv1x1_fwd", act_dtype))) return rc;
    const long M = (long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    float *partials = bn->training ? (float *)scratch : nullptr;
    int grid = 0;
    rc = act_dtype == WW_ACT_BF16  ? launch_pw_fwd_bf16<ww_bf16>(ctx, y_in, ss_in, w, M, y, partials, &grid, st)
         : act_dtype == WW_ACT_F16 ? launch_pw_fwd_bf16<ww_f16>(ctx, y_in, ss_in, w, M, y, partials, &grid, st)
                                   : launch_pw_fwd<float>(ctx, y_in, ss_in, w, M, y, partials, &grid, st);
    if (rc) return rc;
    return finish_bn(ctx, partials, grid, (double)M, bn, ss_out, mr_out, st);
}

extern "C" int ww_gap_fwd(ww_ctx *ctx, int act_dtype, const void *y, const float *ss, const float *mr, int B, int H,
                          int W, float *pool, ww_stream_t stream) {
    WW_REQUIRE(ctx && y && ss && mr && pool, WW_E_INVALID, "ww_gap_fwd: null argument");
    WW_REQUIRE(B >= 1 && H >= 1 && W >= 1, WW_E_INVALID, "ww_gap_fwd: bad shape (%d,%d,%d)", B, H, W);
    int rc = check_act("ww_gap_fwd", act_dtype);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    {
        ww_prof_scope ps_(ctx, WW_K_GAP_FWD, st);
        if (act_dtype == WW_ACT_BF16)
            hipLaunchKernelGGL(k_gap_fwd<ww_bf16>, dim3(B), dim3(1024), 0, st, (const ww_bf16 *)y, ss, mr, H * W, pool);
        else if (act_dtype == WW_ACT_F16)
            hipLaunchKernelGGL(k_gap_fwd<ww_f16>, dim3(B), dim3(1024), 0, st, (const ww_f16 *)y, ss, mr, H * W, pool);
        else
            hipLaunchKernelGGL(k_gap_fwd<float>, dim3(B), dim3(1024), 0, st, (const float *)y, ss, mr, H * W, pool);
    }
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" int ww_head_fwd(ww_ctx *ctx, const float *pool, int B, int HW, const float *fc_w, const float *fc_b,
                           float dropout_p, int training, uint64_t seed, uint64_t step, uint64_t sample_offset,
                           float *pd, float *logits, ww_stream_t stream) {
    WW_REQUIRE(ctx && pool && fc_w && fc_b && pd && logits, WW_E_INVALID, "ww_head_fwd: null argument");
    WW_REQUIRE(B >= 1 && HW >= 1, WW_E_INVALID, "ww_head_fwd: bad shape (%d,%d)", B, HW);
    WW_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, WW_E_INVALID, "ww_head_fwd: dropout_p=%f not in [0,1)", dropout_p);
    const int use_dropout = training && dropout_p > 0.f;
    const float scale = (float)(1.0 / (1.0 - (double)dropout_p));
    ww_prof_scope ps_(ctx, WW_K_HEAD_LOSS, (hipStream_t)stream);
    hipLaunchKernelGGL(k_head_fwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, pool, B, HW, fc_w, fc_b,
                       scale, ww_prob_threshold((double)dropout_p), use_dropout, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)step, (uint32_t)(step >> 32), sample_offset, pd, logits, ctx->step_ctl);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
