// GRU layer (one direction) forward / backward -- the recurrent half of SURVEY.md §8f rank 3 (CRNN); cell maths and
// parameter layout are torch.nn.GRU's, which is what the reference's GRUWakeword wraps
// (src/models/architectures.py:228-235: nn.GRU(input, 128, num_layers=2, batch_first=True, bidirectional=True)):
//   r = s(W_ir x + b_ir + W_hr h + b_hr)   z = s(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))          h' = (1 - z) * n + z * h          gate order r|z|n
// MI355X shape of the computation:
//   * the input projection of ALL time steps is one GEMM  Gi = X W_ih^T + b_ih  (ww_gemm, matrix cores);
//   * the recurrence is ONE persistent kernel per direction: a block owns 16 batch rows for all T steps, keeps h in LDS,
//     and every wavefront keeps ITS slice of W_hh (16 hidden units x 3 gates x 128 = 96 VGPRs) in registers for the whole
//     sequence -- the per-step product h W_hh^T is 96 v_mfma_f32_16x16x4_f32 per wave with no weight traffic at all;
//   * backward mirrors it (dh carried in LDS, W_hh slice by output unit in registers, dGh W_hh per step), and the weight
//     gradients are three large GEMMs over all (batch, time) rows afterwards, split over K with fixed-order sums.
// `mode` selects the matrix type everywhere a matrix core is used: WW_ACT_F32 = exact-fp32 MFMA (parity mode);
// WW_ACT_BF16 = operands rounded to bf16, fp32 accumulation (what autocast does to a GRU) for the big GEMMs (input
// projection, dX, dW_ih, dW_hh) AND for the per-step products of the recurrence (h and W_hh enter the MFMA as bf16; the hidden
// state, the gates and every elementwise step stay fp32).  Hidden size 128 only.
#include "ww_internal.h"
#include "ww_layers.h"
#include "ww_act.h"
#include <algorithm>

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int HB_LD = 128 + 8;       // 16-bit row strides (16-byte aligned 8-element fragments)
constexpr int DGB_LD = 3 * 128 + 8;
// matrix mode of the recurrent kernels: 0 = fp32 MFMA, 1 = bf16, 2 = fp16 operands (state, gates, updates stay fp32)
template <int MODE> struct ModeH { typedef ww_bf16 type; };
template <> struct ModeH<2> { typedef ww_f16 type; };
constexpr int GH = 128;          // hidden size
constexpr int GBT = 16;          // batch rows per block (one 16-row MFMA tile)
constexpr int HS_LD = GH + 4;    // LDS row strides: lane (row i, k) -> bank 4i + k, conflict-free fragment reads
constexpr int DG_LD = 3 * GH + 4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// bf16 matrix mode: v_exp_f32 + v_rcp_f32 forms (1 ulp reciprocal, absolute error ~2e-7 -- far below what the bf16 operands
// of that mode cost); the IEEE division and libm tanhf of the parity mode are ~50 of the ~80 instructions of a gate cell,
// and the recurrence is VALU-issue-bound (2 waves per SIMD, no other work to hide behind)
template <bool FAST> __device__ __forceinline__ float gate_sigmoid(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    else return sigmoidf_(x);
}
template <bool FAST> __device__ __forceinline__ float gate_tanh(float x) {
    if constexpr (FAST) return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)), 1.0f);
    else return tanhf(x);
}

struct GruSaved { float *r, *z, *n, *hn, *hp; };    // (B*T, 128) each: gates, W_hn h + b_hn, h_{t-1}
// per-direction arguments of the recurrent kernels: gridDim.y = 2 runs BOTH directions of a bidirectional layer in one launch
// (blockIdx.y picks the set) -- twice the resident workgroups, no second stream, and a captured HIP graph keeps the concurrency
struct GruFwdDir { const float *gi, *w_hh, *b_hh, *h0; float *y, *hn_out; GruSaved sv; int reverse; };
struct GruBwdDir { const float *w_hh, *dy, *dhn; GruSaved sv; float *dgi, *dgh, *dh0, *bias_part; int reverse; };
__device__ __forceinline__ GruSaved pick(bool second, const GruSaved &a, const GruSaved &b) {
    return GruSaved{second ? b.r : a.r, second ? b.z : a.z, second ? b.n : a.n, second ? b.hn : a.hn, second ? b.hp : a.hp};
}

// grid (ceil(B/16), directions), block 512 = 8 waves; wave w owns hidden units [16w, 16w+16).  BF16: h and W_hh enter the MFMA as bf16
// (v_mfma_f32_16x16x32_bf16, 12 instead of 96 matrix instructions per step); h itself, the gates and the update stay fp32.
// ROWS = batch rows a workgroup owns: 16 (the MFMA tile's height) or 8.  The recurrence is bound by the ELEMENTWISE instructions
// a CU issues per time step (gates, saved-tensor traffic), not by its 12 MFMAs, and at the per-GPU batches of data-parallel
// training B/16 workgroups cover a fraction of the 256 CUs -- so with ROWS = 8 twice as many CUs each do half the elementwise
// work (the MFMA tile keeps 16 rows, the lower 8 unused): the 4 result rows of lanes 0-31 are split with lanes 32-63 by
// v_permlane32_swap, every lane then owns 2 cells instead of 4.
template <int MODE, int ROWS>
__global__ __launch_bounds__(512) void k_gru_fwd(GruFwdDir d0, GruFwdDir d1, int B, int T, long ldy, long bsy, int y_vec) {
    constexpr bool HALF = ROWS == 8;
    constexpr int NC = HALF ? 2 : 4;             // cells (batch rows of its unit) per lane
    const bool second = blockIdx.y != 0;
    const float *__restrict__ gi = second ? d1.gi : d0.gi, *__restrict__ w_hh = second ? d1.w_hh : d0.w_hh;
    const float *__restrict__ b_hh = second ? d1.b_hh : d0.b_hh, *__restrict__ h0 = second ? d1.h0 : d0.h0;
    float *__restrict__ y = second ? d1.y : d0.y, *__restrict__ hn_out = second ? d1.hn_out : d0.hn_out;
    const GruSaved sv = pick(second, d0.sv, d1.sv);
    const int reverse = second ? d1.reverse : d0.reverse;
    constexpr bool BF16 = MODE != 0;
    typedef typename ModeH<MODE>::type H;
    typedef typename H16<H>::x8 bf16x8;
    __shared__ __align__(16) float hs[2][GBT][HS_LD];
    __shared__ __align__(16) H hb[BF16 ? 2 : 1][BF16 ? GBT : 1][HB_LD];
    // what the backward needs of a step (r, z, n, W_hn h + b_hn, h_{t-1}) is parked in LDS in the MFMA result layout (a lane
    // holds 4 ROWS of one unit) and written out one step later as float4 along the UNIT axis by thread (row, 4 units):
    // 6 vector stores per thread and step instead of 24 scalar ones -- the store issue, not the MFMAs, bounded a step.
    // Two parities: a tile is rewritten two steps after it was filled, with a barrier in between.
    extern __shared__ __align__(16) float gru_sav[];           // [2][5][ROWS][HS_LD]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, j = l & 15, kq = l >> 4;
    const int b0 = blockIdx.x * ROWS, u = 16 * w + j;
    const int crow0 = HALF ? 4 * (kq & 1) + 2 * (kq >> 1) : 4 * kq;      // first of this lane's NC consecutive batch rows
    float wreg[BF16 ? 1 : 3][BF16 ? 1 : 32];      // fp32: B operand of k-step kk, gate g: W_hh[g*128 + u][4kk + kq]
    bf16x8 wb[BF16 ? 3 : 1][BF16 ? 4 : 1];        // bf16: W_hh[g*128 + u][32kk + 8kq .. +7]
    if constexpr (BF16) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = w_hh[(size_t)(g * GH + u) * GH + 32 * kk + 8 * kq + e];
                wb[g][kk] = ww_pack8<H>(v);
            }
    } else {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) wreg[g][kk] = w_hh[(size_t)(g * GH + u) * GH + 4 * kk + kq];
    }
    const float bhr = b_hh[u], bhz = b_hh[GH + u], bhn = b_hh[2 * GH + u];
    for (int e = tid; e < GBT * GH; e += 512) {
        const int row = e >> 7, c = e & 127;
        const float hv = (h0 && row < ROWS && b0 + row < B) ? h0[(size_t)(b0 + row) * GH + c] : 0.f;
        hs[0][row][c] = hv;
        if constexpr (BF16) { hb[0][row][c] = (H)hv; hb[1][row][c] = (H)0.f; }     // (rows >= ROWS stay zero operands)
    }
    __syncthreads();
    // input projections of a step are loaded TWO steps ahead (register sets A / B, alternating): a step is ~0.5 us of
    // MFMA + gate work, an HBM/L2 round trip 1-2 us -- loaded at the top of the step that needs them (first version) the
    // latency was most of the step
    const float *gbase[NC];                      // row (b, t = 0) of this lane's batch rows, at its unit
#pragma unroll
    for (int reg = 0; reg < NC; ++reg) gbase[reg] = gi + (size_t)min(b0 + crow0 + reg, B - 1) * T * (3 * GH) + u;
    auto load_gi = [&](int it, float (&gr)[NC], float (&gz)[NC], float (&gn)[NC]) {
        if (it >= T) return;
        const size_t toff = (size_t)(reverse ? T - 1 - it : it) * (3 * GH);      // wave-uniform
#pragma unroll
        for (int reg = 0; reg < NC; ++reg) {
            const float *g3 = gbase[reg] + toff;
            gr[reg] = g3[0]; gz[reg] = g3[GH]; gn[reg] = g3[2 * GH];
        }
    };
    const int frow = tid >> 5, fc0 = 4 * (tid & 31);
    const bool frow_ok = frow < ROWS && b0 + frow < B;           // (ROWS = 8: the upper four waves have nothing to flush)
    const size_t fm0 = (size_t)(b0 + frow) * T * GH + fc0;                 // element (b, t = 0, fc0) of the (B*T, 128) tensors
    float *const fy0 = y + (size_t)(b0 + frow) * bsy + fc0;
    const float *const fs0 = gru_sav + frow * HS_LD + fc0;
    auto flush = [&](int it) {                  // step `it` is complete (barrier passed): its tiles -> HBM
        if (!frow_ok) return;
        const int t = reverse ? T - 1 - it : it, par = it & 1;
        const size_t m = fm0 + (size_t)t * GH;
        const float *sp = fs0 + (size_t)par * 5 * ROWS * HS_LD;
        *reinterpret_cast<float4 *>(sv.r + m) = *reinterpret_cast<const float4 *>(sp);
        *reinterpret_cast<float4 *>(sv.z + m) = *reinterpret_cast<const float4 *>(sp + ROWS * HS_LD);
        *reinterpret_cast<float4 *>(sv.n + m) = *reinterpret_cast<const float4 *>(sp + 2 * ROWS * HS_LD);
        *reinterpret_cast<float4 *>(sv.hn + m) = *reinterpret_cast<const float4 *>(sp + 3 * ROWS * HS_LD);
        *reinterpret_cast<float4 *>(sv.hp + m) = *reinterpret_cast<const float4 *>(sp + 4 * ROWS * HS_LD);
        const float4 h4 = *reinterpret_cast<const float4 *>(&hs[par ^ 1][frow][fc0]);
        float *yo = fy0 + (size_t)t * ldy;
        if (y_vec) *reinterpret_cast<float4 *>(yo) = h4;
        else { yo[0] = h4.x; yo[1] = h4.y; yo[2] = h4.z; yo[3] = h4.w; }
    };
    auto step = [&](int it, float (&sr)[NC], float (&sz)[NC], float (&sn)[NC]) {
        const int cur = it & 1;
        float gir[NC], giz[NC], gin[NC];
#pragma unroll
        for (int reg = 0; reg < NC; ++reg) { gir[reg] = sr[reg]; giz[reg] = sz[reg]; gin[reg] = sn[reg]; }
        load_gi(it + 2, sr, sz, sn);            // the set is free again: refill it for the step after next
        if (it > 0) flush(it - 1);
        floatx4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if constexpr (BF16) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&hb[cur][j][32 * kk + 8 * kq]);
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = H16<H>::mfma16(a, wb[g][kk], acc[g]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float a = hs[cur][j][4 * kk + kq];
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[g][kk], acc[g], 0, 0, 0);
            }
        }
        if constexpr (HALF) {                   // lanes 32-63 take over result rows 2, 3 of lanes 0-31 (rows 8-15 are unused)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    // (plain float copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this clang --
                    //  the generated code swapped acc[g][0] with a copy of itself)
                    const float keep = acc[g][c], give = acc[g][c + 2];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(keep), __float_as_uint(give), false, false);
                    acc[g][c] = __uint_as_float(sw[0]);
                }
        }
#pragma unroll
        for (int reg = 0; reg < NC; ++reg) {    // D[row = crow0 + reg][unit u]
            const int row = crow0 + reg;
            const float hnv = acc[2][reg] + bhn;
            const float r = gate_sigmoid<BF16>(gir[reg] + acc[0][reg] + bhr);
            const float z = gate_sigmoid<BF16>(giz[reg] + acc[1][reg] + bhz);
            const float n = gate_tanh<BF16>(gin[reg] + r * hnv);
            const float hp = hs[cur][row][u];
            const float h = (1.0f - z) * n + z * hp;
            hs[cur ^ 1][row][u] = h;
            if constexpr (BF16) hb[cur ^ 1][row][u] = (H)h;
            float *sp = gru_sav + (size_t)cur * 5 * ROWS * HS_LD + row * HS_LD + u;
            sp[0] = r; sp[ROWS * HS_LD] = z; sp[2 * ROWS * HS_LD] = n; sp[3 * ROWS * HS_LD] = hnv; sp[4 * ROWS * HS_LD] = hp;
        }
        __syncthreads();
    };
    {
        float ar[NC], az[NC], an[NC], br[NC], bz[NC], bn[NC];
        load_gi(0, ar, az, an);
        load_gi(1, br, bz, bn);
        for (int it = 0; it < T; it += 2) {
            step(it, ar, az, an);
            if (it + 1 < T) step(it + 1, br, bz, bn);
        }
        flush(T - 1);
    }
    if (hn_out)
        for (int e = tid; e < ROWS * GH; e += 512) {
            const int row = e >> 7, c = e & 127;
            if (b0 + row < B) hn_out[(size_t)(b0 + row) * GH + c] = hs[T & 1][row][c];
        }
}

// same decomposition; wave w owns OUTPUT units [16w,16w+16) of dh_{t-1} = dh*z + dGh W_hh
template <int MODE, int ROWS>
__global__ __launch_bounds__(512) void k_gru_bwd(GruBwdDir d0, GruBwdDir d1, long ldy, long bsy, int B, int T, int dy_vec) {
    const bool second = blockIdx.y != 0;
    const float *__restrict__ w_hh = second ? d1.w_hh : d0.w_hh, *__restrict__ dy = second ? d1.dy : d0.dy;
    const float *__restrict__ dhn = second ? d1.dhn : d0.dhn;
    const GruSaved sv = pick(second, d0.sv, d1.sv);
    float *__restrict__ dgi = second ? d1.dgi : d0.dgi, *__restrict__ dgh = second ? d1.dgh : d0.dgh;
    float *__restrict__ dh0 = second ? d1.dh0 : d0.dh0, *__restrict__ bias_part = second ? d1.bias_part : d0.bias_part;
    const int reverse = second ? d1.reverse : d0.reverse;
    constexpr bool BF16 = MODE != 0;
    typedef typename ModeH<MODE>::type H;
    typedef typename H16<H>::x8 bf16x8;
    __shared__ __align__(16) float dhs[GBT][HS_LD];
    __shared__ __align__(16) float dg[BF16 ? 1 : GBT][DG_LD];      // fp32 operand tile
    __shared__ __align__(16) H dgb[BF16 ? GBT : 1][DGB_LD];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, j = l & 15, kq = l >> 4;
    const int b0 = blockIdx.x * ROWS, u = 16 * w + j;
    float wreg[BF16 ? 1 : 96];         // fp32: B operand of k-step cc: W_hh[4cc + kq][u]  (contraction over the 384 gate rows)
    bf16x8 wb[BF16 ? 12 : 1];          // bf16: W_hh[32cc + 8kq .. +7][u]
    if constexpr (BF16) {
#pragma unroll
        for (int cc = 0; cc < 12; ++cc) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w_hh[(size_t)(32 * cc + 8 * kq + e) * GH + u];
            wb[cc] = ww_pack8<H>(v);
        }
    } else {
#pragma unroll
        for (int cc = 0; cc < 96; ++cc) wreg[cc] = w_hh[(size_t)(4 * cc + kq) * GH + u];
    }
    for (int e = tid; e < GBT * GH; e += 512) {
        const int row = e >> 7, c = e & 127;
        dhs[row][c] = (dhn && row < ROWS && b0 + row < B) ? dhn[(size_t)(b0 + row) * GH + c] : 0.f;
    }
    if constexpr (ROWS < GBT) {                     // operand rows nobody writes stay zero (their result rows are never read)
        for (int e = tid; e < (GBT - ROWS) * 3 * GH; e += 512) {
            const int row = ROWS + e / (3 * GH), c = e % (3 * GH);
            if constexpr (BF16) dgb[row][c] = (H)0.f; else dg[row][c] = 0.f;
        }
    }
    __syncthreads();
    // elementwise part: thread = (batch row, 4 consecutive units) -> every tensor moves as ONE float4 per thread and step
    // (5 saved gates + dy in, dGi / dGh (3 gates each) out: 12 vector accesses instead of 48 scalar ones; the recurrence
    // is VALU-issue-bound and two thirds of its instructions were address arithmetic and scalar memory operations).
    // The saved gates of a step are loaded TWO steps ahead (register sets A / B, alternating), as in the forward kernel.
    const int erow = tid >> 5, ec0 = 4 * (tid & 31);
    const int eb = min(b0 + erow, B - 1);
    const bool erow_ok = b0 + erow < B;
    const bool ewave = erow < ROWS;                 // wave-uniform (two rows per wave): ROWS = 8 leaves the elementwise part to waves 0-3
    const size_t em0 = (size_t)eb * T * GH + ec0;                        // (b, t = 0, ec0) of the (B*T, 128) tensors
    const float *const edy0 = dy ? dy + (size_t)eb * bsy + ec0 : nullptr;
    struct Saved { float4 r, z, n, hn, hp, dy; };
    auto prefetch = [&](int it, Saved &S) {
        if (it >= T || !ewave) return;
        const int t = reverse ? it : T - 1 - it;        // the forward pass's time order, backwards
        const size_t i = em0 + (size_t)t * GH;
        S.r = *reinterpret_cast<const float4 *>(sv.r + i);
        S.z = *reinterpret_cast<const float4 *>(sv.z + i);
        S.n = *reinterpret_cast<const float4 *>(sv.n + i);
        S.hn = *reinterpret_cast<const float4 *>(sv.hn + i);
        S.hp = *reinterpret_cast<const float4 *>(sv.hp + i);
        if (edy0) {
            const float *p = edy0 + (size_t)t * ldy;
            S.dy = dy_vec ? *reinterpret_cast<const float4 *>(p) : make_float4(p[0], p[1], p[2], p[3]);
        } else {
            S.dy = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float sar[4] = {0.f, 0.f, 0.f, 0.f}, saz[4] = {0.f, 0.f, 0.f, 0.f}, san[4] = {0.f, 0.f, 0.f, 0.f}, shn[4] = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](int it, Saved &S) {
        const int t = reverse ? it : T - 1 - it;
        if (ewave) {
        const float pr[4] = {S.r.x, S.r.y, S.r.z, S.r.w}, pz[4] = {S.z.x, S.z.y, S.z.z, S.z.w};
        const float pn[4] = {S.n.x, S.n.y, S.n.z, S.n.w}, phn[4] = {S.hn.x, S.hn.y, S.hn.z, S.hn.w};
        const float php[4] = {S.hp.x, S.hp.y, S.hp.z, S.hp.w}, pdy[4] = {S.dy.x, S.dy.y, S.dy.z, S.dy.w};
        prefetch(it + 2, S);                            // the set is free again
        const float4 dh4 = *reinterpret_cast<const float4 *>(&dhs[erow][ec0]);
        const float dhv[4] = {dh4.x, dh4.y, dh4.z, dh4.w};
        float dar[4], daz[4], dan[4], dhn_[4], keep[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dh = erow_ok ? dhv[q] + pdy[q] : 0.f;
            const float r = pr[q], z = pz[q], n = pn[q], hnv = phn[q], hp = php[q];
            const float dn = dh * (1.0f - z), dz = dh * (hp - n);
            dan[q] = dn * (1.0f - n * n);
            dar[q] = dan[q] * hnv * r * (1.0f - r);
            daz[q] = dz * z * (1.0f - z);
            dhn_[q] = dan[q] * r;
            keep[q] = dh * z;
            sar[q] += dar[q]; saz[q] += daz[q]; san[q] += dan[q]; shn[q] += dhn_[q];
        }
        const size_t m3 = ((size_t)(b0 + erow) * T + t) * (3 * GH) + ec0;
        if constexpr (BF16) {
            // dGi / dGh leave in the matrix type: the weight-gradient and dX products round their operands to it while staging
            // them anyway (ww_gemm, a16), so the results are bit-identical and those HBM-bound products read half the bytes
            typedef Act<H> A16;
            const uint2 qr = make_uint2(A16::pack2(dar[0], dar[1]), A16::pack2(dar[2], dar[3]));
            const uint2 qz = make_uint2(A16::pack2(daz[0], daz[1]), A16::pack2(daz[2], daz[3]));
            const uint2 qn = make_uint2(A16::pack2(dan[0], dan[1]), A16::pack2(dan[2], dan[3]));
            const uint2 qh = make_uint2(A16::pack2(dhn_[0], dhn_[1]), A16::pack2(dhn_[2], dhn_[3]));
            if (erow_ok) {
                H *gi16 = reinterpret_cast<H *>(dgi) + m3, *gh16 = reinterpret_cast<H *>(dgh) + m3;
                *reinterpret_cast<uint2 *>(gi16) = qr;
                *reinterpret_cast<uint2 *>(gi16 + GH) = qz;
                *reinterpret_cast<uint2 *>(gi16 + 2 * GH) = qn;
                *reinterpret_cast<uint2 *>(gh16) = qr;
                *reinterpret_cast<uint2 *>(gh16 + GH) = qz;
                *reinterpret_cast<uint2 *>(gh16 + 2 * GH) = qh;
            }
            *reinterpret_cast<uint2 *>(&dgb[erow][ec0]) = qr;
            *reinterpret_cast<uint2 *>(&dgb[erow][GH + ec0]) = qz;
            *reinterpret_cast<uint2 *>(&dgb[erow][2 * GH + ec0]) = qh;
        } else {
            if (erow_ok) {
                *reinterpret_cast<float4 *>(dgi + m3) = make_float4(dar[0], dar[1], dar[2], dar[3]);
                *reinterpret_cast<float4 *>(dgi + m3 + GH) = make_float4(daz[0], daz[1], daz[2], daz[3]);
                *reinterpret_cast<float4 *>(dgi + m3 + 2 * GH) = make_float4(dan[0], dan[1], dan[2], dan[3]);
                *reinterpret_cast<float4 *>(dgh + m3) = make_float4(dar[0], dar[1], dar[2], dar[3]);
                *reinterpret_cast<float4 *>(dgh + m3 + GH) = make_float4(daz[0], daz[1], daz[2], daz[3]);
                *reinterpret_cast<float4 *>(dgh + m3 + 2 * GH) = make_float4(dhn_[0], dhn_[1], dhn_[2], dhn_[3]);
            }
            *reinterpret_cast<float4 *>(&dg[erow][ec0]) = make_float4(dar[0], dar[1], dar[2], dar[3]);
            *reinterpret_cast<float4 *>(&dg[erow][GH + ec0]) = make_float4(daz[0], daz[1], daz[2], daz[3]);
            *reinterpret_cast<float4 *>(&dg[erow][2 * GH + ec0]) = make_float4(dhn_[0], dhn_[1], dhn_[2], dhn_[3]);
        }
        *reinterpret_cast<float4 *>(&dhs[erow][ec0]) = make_float4(keep[0], keep[1], keep[2], keep[3]);   // this thread's cells only
        }
        __syncthreads();
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
#pragma unroll
            for (int cc = 0; cc < 12; ++cc)
                acc = H16<H>::mfma16(*reinterpret_cast<const bf16x8 *>(&dgb[j][32 * cc + 8 * kq]), wb[cc], acc);
        } else {
#pragma unroll
            for (int cc = 0; cc < 96; ++cc)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dg[j][4 * cc + kq], wreg[cc], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dhs[4 * kq + reg][u] += acc[reg];
        __syncthreads();
    };
    {
        Saved A, Bs;
        prefetch(0, A);
        prefetch(1, Bs);
        for (int it = 0; it < T; it += 2) {
            step(it, A);
            if (it + 1 < T) step(it + 1, Bs);
        }
    }
    if (dh0)
        for (int e = tid; e < ROWS * GH; e += 512) {
            const int row = e >> 7, c = e & 127;
            if (b0 + row < B) dh0[(size_t)(b0 + row) * GH + c] = dhs[row][c];
        }
    // bias gradients: this block's column sums over its 16 rows and all time steps -> bias_part[block][db_ih(384) | db_hh(384)]
    // (thread (row, 4 columns) holds the time sums of its cells; one pass through dhs per gate, rows added in a fixed order)
    {
        float *o = bias_part + (size_t)blockIdx.x * (6 * GH);
        const float *src[4] = {sar, saz, san, shn};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __syncthreads();
            if (ewave) *reinterpret_cast<float4 *>(&dhs[erow][ec0]) = make_float4(src[k][0], src[k][1], src[k][2], src[k][3]);
            __syncthreads();
            if (tid < 128) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) t += dhs[r][tid];
                if (k == 0) { o[tid] = t; o[3 * GH + tid] = t; }
                else if (k == 1) { o[GH + tid] = t; o[4 * GH + tid] = t; }
                else if (k == 2) o[2 * GH + tid] = t;
                else o[5 * GH + tid] = t;
            }
        }
    }
}

// Philox dropout on a (B,T,C) tensor with row-strided rows: out = keep ? x * scale : 0.  Counter field (24 bits):
// stream_id (4) | t (12) | c>>2 (8); lane c&3; sample = global batch index.  Its own backward (apply to the gradient).
__global__ __launch_bounds__(256) void k_dropout_bt(const float *__restrict__ x, long ldx, int B, int T, int C, float scale,
                                                    uint64_t thresh, uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo,
                                                    uint32_t step_hi, uint64_t sample_offset, uint32_t stream_id,
                                                    float *__restrict__ out, long ldo, const ww_step_ctl *__restrict__ ctl) {
    ww_step_resolve(ctl, step_lo, step_hi, step_lo, step_hi);
    const int cq = (C + 3) / 4;
    const long n = (long)B * T * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int q = (int)(i % cq);
        const long m = i / cq;
        const int t = (int)(m % T), b = (int)(m / T);
        uint32_t rr[4];
        ww_philox(step_lo, step_hi, (uint32_t)(sample_offset + (uint64_t)b),
                  (WW_TAG_DROPOUT << 24) | (stream_id << 20) | ((uint32_t)t << 8) | (uint32_t)q, seed_lo, seed_hi, rr);
        const float *src = x + m * ldx + 4 * q;
        float *dst = out + m * ldo + 4 * q;
        if (4 * q < C) dst[0] = (uint64_t)rr[0] >= thresh ? src[0] * scale : 0.f;
        if (4 * q + 1 < C) dst[1] = (uint64_t)rr[1] >= thresh ? src[1] * scale : 0.f;
        if (4 * q + 2 < C) dst[2] = (uint64_t)rr[2] >= thresh ? src[2] * scale : 0.f;
        if (4 * q + 3 < C) dst[3] = (uint64_t)rr[3] >= thresh ? src[3] * scale : 0.f;
    }
}

// fp32 -> 16-bit copies of the input projection's two operands in ONE launch: x (M rows of I floats, row stride ldx) -> xh (M, I),
// w (Nw*I contiguous) -> wh.  I % 4 == 0; a thread moves float4s in batches of four (unconditional, clamped loads).
template <typename H>
__global__ __launch_bounds__(256) void k_to16_pair(const float *__restrict__ x, long ldx, long M, int I, const float *__restrict__ w,
                                                   long nw, H *__restrict__ xh, H *__restrict__ wh) {
    typedef Act<H> A16;
    const long I4 = I >> 2, nx4 = M * I4, n4 = nx4 + (nw >> 2);
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4L * gridDim.x * 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = min(i0 + (long)u * gridDim.x * 256, n4 - 1);
            const float *src = i < nx4 ? x + (i / I4) * ldx + 4 * (i % I4) : w + 4 * (i - nx4);
            v[u] = *reinterpret_cast<const float4 *>(src);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + (long)u * gridDim.x * 256;
            if (i < n4) {
                H *dst = i < nx4 ? xh + 4 * i : wh + 4 * (i - nx4);
                *reinterpret_cast<uint2 *>(dst) = make_uint2(A16::pack2(v[u].x, v[u].y), A16::pack2(v[u].z, v[u].w));
            }
        }
    }
}

struct WsLayout { size_t gi, dgh, r, z, n, hn, hp, part, total; };
constexpr int GRU_SPLITS = 128;          // workspace bound of the weight-gradient GEMMs' K splits (gru_splits() picks the count)
WsLayout ws_layout(long B, long T, int I) {
    WsLayout L;
    size_t o = 0;
    auto take = [&](size_t nfloat) { size_t r = o; o += (nfloat * sizeof(float) + 255) & ~(size_t)255; return r; };
    const size_t M = (size_t)B * T;
    L.gi = take(M * 3 * GH);        // projections, overwritten by dGi in the backward pass
    L.dgh = take(M * 3 * GH);
    L.r = take(M * GH); L.z = take(M * GH); L.n = take(M * GH); L.hn = take(M * GH); L.hp = take(M * GH);
    L.part = take((size_t)GRU_SPLITS * 3 * GH * std::max(I, GH) + (size_t)((B + 7) / 8) * 6 * GH);      // (8-row workgroups: B/8 bias partials)
    L.total = o;
    return L;
}
int check_gru(const char *who, ww_ctx *ctx, int B, int T, int I, int H, const void *ws, size_t ws_bytes) {
    WW_REQUIRE(ctx && ws, WW_E_INVALID, "%s: null argument", who);
    WW_REQUIRE(B >= 1 && T >= 1 && I >= 1, WW_E_INVALID, "%s: bad shape B=%d T=%d I=%d", who, B, T, I);
    WW_REQUIRE(H == GH, WW_E_UNSUPPORTED, "%s: hidden size %d not implemented (128 only)", who, H);
    WW_REQUIRE(ws_bytes >= ws_layout(B, T, I).total, WW_E_WORKSPACE, "%s: workspace too small", who);
    WW_REQUIRE(((uintptr_t)ws & 255) == 0, WW_E_INVALID, "%s: workspace must be 256-byte aligned", who);
    return WW_OK;
}
GruSaved saved(char *w, const WsLayout &L) {
    return GruSaved{(float *)(w + L.r), (float *)(w + L.z), (float *)(w + L.n), (float *)(w + L.hn), (float *)(w + L.hp)};
}

}  // namespace

extern "C" int ww_dropout_bt(ww_ctx *ctx, const float *x, long ldx, int B, int T, int C, float p, uint64_t seed,
                             uint64_t step, uint64_t sample_offset, int stream_id, float *out, long ldo,
                             ww_stream_t stream) {
    WW_REQUIRE(ctx && x && out, WW_E_INVALID, "ww_dropout_bt: null argument");
    WW_REQUIRE(B >= 1 && T >= 1 && C >= 1 && ldx >= C && ldo >= C, WW_E_INVALID, "ww_dropout_bt: bad shape");
    WW_REQUIRE(T <= 4096 && C <= 1024 && stream_id >= 0 && stream_id < 16, WW_E_UNSUPPORTED,
               "ww_dropout_bt: T <= 4096, C <= 1024, stream_id < 16");
    WW_REQUIRE(p >= 0.f && p < 1.f, WW_E_INVALID, "ww_dropout_bt: p=%f not in [0,1)", (double)p);
    const long n = (long)B * T * ((C + 3) / 4);
    const int grid = (int)std::min<long>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_dropout_bt, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, B, T, C,
                       (float)(1.0 / (1.0 - (double)p)), ww_prob_threshold((double)p), (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)step, (uint32_t)(step >> 32), sample_offset, (uint32_t)stream_id, out, ldo, ctx->step_ctl);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" size_t ww_gru_workspace_bytes(int B, int T, int I, int H) {
    if (B < 1 || T < 1 || I < 1 || H != GH) return 0;
    return ws_layout(B, T, I).total;
}

// ---- host side: a layer = 1 or 2 directions.  Per direction: the input projection GEMM, then ONE recurrent launch for all
// directions (gridDim.y), then (backward) the weight-gradient / bias / dX products per direction on the same stream.
namespace {
// batch rows per workgroup of the recurrent kernels: 8 while 16-row workgroups would leave more than half of the CUs idle
// (WW_GRU_ROWS = 8 | 16 overrides, for A/B measurements)
int gru_rows(int B, int nd) {
    const int forced = ww_env_int("WW_GRU_ROWS", 0);          // (read per call: the tests switch it inside one process)
    if (forced == 8 || forced == 16) return forced;
    return (long)((B + GBT - 1) / GBT) * nd <= 128 ? 8 : GBT;
}
struct FwdDirHost { const float *w_ih, *w_hh, *b_ih, *b_hh, *h0; float *y, *h_n; char *ws; int reverse; };
struct BwdDirHost { const float *w_ih, *w_hh, *dy, *dh_n; char *ws; float *dw_ih, *dw_hh, *db_ih, *db_hh, *dh0; int reverse; };

// *xh_shared: the 16-bit copy of x another direction of the same layer has already made (both directions project the SAME input:
// only the weights are converted then); set to this call's copy when it makes one
int gru_project(ww_ctx *ctx, int mode, const float *x, long ldx, const FwdDirHost &d, const WsLayout &L, int B, int T, int I,
                hipStream_t st, const void **xh_shared) {
    // Gi[(b,t)][3H] = x[(b,t)][:] W_ih^T + b_ih for all time steps at once.  16-bit matrix modes with I a multiple of 64 (the
    // CRNN's 64 conv channels, every second layer's 256): both operands are rounded ONCE into 16-bit copies (the dGh region of
    // the workspace is idle in the forward pass) and the product runs on ww_gemm16_nt's 128 x 128 LDS-DMA tiles with b_ih added
    // in its epilogue -- the same operand roundings as k_gemm's LDS fill, 3-5x faster than its 64 x 64 tiles at these shapes.
    char *w = d.ws;
    const long Mrows = (long)B * T;
    const size_t xh_bytes = ((size_t)Mrows * I * 2 + 255) & ~(size_t)255, wh_bytes = (size_t)3 * GH * I * 2;
    static const int use_gemm16 = ww_env_int("WW_GRU_GEMM16", 1);      // A/B knob: 0 = the k_gemm path for every shape
    if (use_gemm16 && mode != WW_ACT_F32 && I % 64 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)d.w_ih) & 15) == 0 &&
        xh_bytes + wh_bytes <= (size_t)Mrows * 3 * GH * sizeof(float)) {
        void *xh = w + L.dgh, *wh = w + L.dgh + xh_bytes;
        const bool have_x = xh_shared && *xh_shared;
        const long rows = have_x ? 0 : Mrows;                          // rows of x this launch still has to convert
        const long n4 = rows * (I / 4) + 3L * GH * I / 4;
        const int grid = (int)std::min<long>((n4 + 4 * 256 - 1) / (4 * 256), 4096);
        if (mode == WW_ACT_BF16)
            hipLaunchKernelGGL(k_to16_pair<ww_bf16>, dim3(grid), dim3(256), 0, st, x, ldx, rows, I, d.w_ih, 3L * GH * I, (ww_bf16 *)xh, (ww_bf16 *)wh);
        else
            hipLaunchKernelGGL(k_to16_pair<ww_f16>, dim3(grid), dim3(256), 0, st, x, ldx, rows, I, d.w_ih, 3L * GH * I, (ww_f16 *)xh, (ww_f16 *)wh);
        WW_LAUNCH_CHECK();
        const void *xa = have_x ? *xh_shared : xh;
        if (xh_shared && !have_x) *xh_shared = xh;
        return ww_gemm16_nt_bias(ctx, mode, xa, wh, w + L.gi, 1, Mrows, 3 * GH, I, d.b_ih, st);
    }
    return ww_gemm(mode, x, ldx, 1, B * T, d.w_ih, I, 1, 3 * GH, I, (float *)(w + L.gi), 3 * GH, d.b_ih, 0, 1, nullptr, st);
}

int gru_layer_fwd(ww_ctx *ctx, int mode, const float *x, long ldx, const FwdDirHost *d, int nd, int B, int T, int I, long ldy,
                  hipStream_t st) {
    const WsLayout L = ws_layout(B, T, I);
    int rc;
    const void *xh_shared = nullptr;
    for (int k = 0; k < nd; ++k)
        if ((rc = gru_project(ctx, mode, x, ldx, d[k], L, B, T, I, st, &xh_shared))) return rc;
    GruFwdDir a[2];
    int y_vec = ldy % 4 == 0;
    for (int k = 0; k < 2; ++k) {
        const FwdDirHost &h = d[k < nd ? k : 0];
        a[k] = GruFwdDir{(const float *)(h.ws + L.gi), h.w_hh, h.b_hh, h.h0, h.y, h.h_n, saved(h.ws, L), h.reverse};
        y_vec = y_vec && (((uintptr_t)h.y & 15) == 0);
    }
    const int rows = gru_rows(B, nd);
    const size_t smem = (size_t)2 * 5 * rows * HS_LD * sizeof(float);
    auto go = [&](auto kern) -> int {
        WW_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(kern, dim3((B + rows - 1) / rows, nd), dim3(512), smem, st, a[0], a[1], B, T, ldy, (long)T * ldy, y_vec);
        return WW_OK;
    };
    if (rows == 8) rc = mode == WW_ACT_BF16 ? go(k_gru_fwd<1, 8>) : mode == WW_ACT_F16 ? go(k_gru_fwd<2, 8>) : go(k_gru_fwd<0, 8>);
    else rc = mode == WW_ACT_BF16 ? go(k_gru_fwd<1, 16>) : mode == WW_ACT_F16 ? go(k_gru_fwd<2, 16>) : go(k_gru_fwd<0, 16>);
    if (rc) return rc;
    WW_LAUNCH_CHECK();
    return WW_OK;
}

int gru_layer_bwd(ww_ctx *ctx, int mode, const float *x, long ldx, const BwdDirHost *d, int nd, long ldy, int B, int T, int I,
                  float *dx, long lddx, int accumulate_dx, hipStream_t st) {
    const WsLayout L = ws_layout(B, T, I);
    const int rows = gru_rows(B, nd);
    const int nblk = (B + rows - 1) / rows;
    const size_t bpart_off = (size_t)GRU_SPLITS * 3 * GH * std::max(I, GH);
    GruBwdDir a[2];
    int dy_vec = ldy % 4 == 0;
    for (int k = 0; k < 2; ++k) {
        const BwdDirHost &h = d[k < nd ? k : 0];
        a[k] = GruBwdDir{h.w_hh, h.dy, h.dh_n, saved(h.ws, L), (float *)(h.ws + L.gi), (float *)(h.ws + L.dgh), h.dh0,
                         (float *)(h.ws + L.part) + bpart_off, h.reverse};
        dy_vec = dy_vec && (!h.dy || ((uintptr_t)h.dy & 15) == 0);
    }
    const dim3 grid(nblk, nd);
#define WW_GRU_BWD(M_, R_) hipLaunchKernelGGL((k_gru_bwd<M_, R_>), grid, dim3(512), 0, st, a[0], a[1], ldy, (long)T * ldy, B, T, dy_vec)
    if (rows == 8) { if (mode == WW_ACT_BF16) WW_GRU_BWD(1, 8); else if (mode == WW_ACT_F16) WW_GRU_BWD(2, 8); else WW_GRU_BWD(0, 8); }
    else { if (mode == WW_ACT_BF16) WW_GRU_BWD(1, 16); else if (mode == WW_ACT_F16) WW_GRU_BWD(2, 16); else WW_GRU_BWD(0, 16); }
#undef WW_GRU_BWD
    WW_LAUNCH_CHECK();
    const int M = B * T;
    // K splits of the weight-gradient products (contraction over the B*T rows, 12-24 output tiles): as for the 1x1 convolutions
    // a split is a chain of dependent K stages, so more, shallower splits finish sooner -- bounded by the partial traffic
    // (splits x 3H x max(I, H) floats written and re-read): CRNN B=512 step 2.604 / 2.592 / 2.626 ms at 32 / 64 / 128
    // (profiles/r03_i_*; WW_GRU_SPLITS for measurements)
    const int splits = M >= 4096 ? std::min(GRU_SPLITS, std::max(1, ww_env_int("WW_GRU_SPLITS", 64))) : 1;
    int rc;
    // While the context is deferring (ww_ctx_set_deferred_reduce) the three "sum the partials" launches of a direction are
    // queued: the two weight-gradient products then keep their partials apart (dW_hh in the first, dW_ih in the second part of
    // the region sized for GRU_SPLITS splits), and the bias partials are one 768-column item when db_ih | db_hh are adjacent
    // (nn.GRU's parameter order, i.e. their slots of a flat gradient bucket)
    const bool defer = ctx && ctx->defer_on && splits <= GRU_SPLITS / 2;
    for (int k = 0; k < nd; ++k) {
        const BwdDirHost &h = d[k];
        float *dgi = (float *)(h.ws + L.gi), *dgh = (float *)(h.ws + L.dgh), *part = (float *)(h.ws + L.part);
        float *part_ih = defer ? part + (size_t)splits * 3 * GH * GH : part;
        const GruSaved sv = saved(h.ws, L);
        // dW_hh[c][k] = sum_m dGh[m][c] h_prev[m][k]   ;   dW_ih[c][i] = sum_m dGi[m][c] x[m][i]
        // (16-bit modes: the recurrent kernel left dGi / dGh in the matrix type -- a16)
        if ((rc = ww_gemm(mode, dgh, 1, 3 * GH, 3 * GH, sv.hp, 1, GH, GH, M, h.dw_hh, GH, nullptr, 0, splits, part, st, defer ? ctx : nullptr, 1))) return rc;
        if ((rc = ww_gemm(mode, dgi, 1, 3 * GH, 3 * GH, x, 1, ldx, I, M, h.dw_ih, I, nullptr, 0, splits, part_ih, st, defer ? ctx : nullptr, 1))) return rc;
        // db_ih | db_hh: fixed-order sum of the per-block partials the recurrent kernel left (one launch for both: 768 columns)
        if (defer && h.db_hh == h.db_ih + 3 * GH) ww_defer(ctx, part + bpart_off, h.db_ih, 6 * GH, nblk, 0);
        else if ((rc = ww_colsum_pair(part + bpart_off, nblk, 3 * GH, h.db_ih, h.db_hh, st))) return rc;
        // dx[m][i] (+)= sum_c dGi[m][c] W_ih[c][i]   (the second direction adds to the first one's; both directions of a
        // layer: ONE product over the two (dGi, W_ih) pairs below instead)
        if (dx && nd != 2 && (rc = ww_gemm(mode, dgi, 3 * GH, 1, M, h.w_ih, 1, I, I, 3 * GH, dx, lddx, nullptr, accumulate_dx || k > 0, 1, nullptr, st, nullptr, 1)))
            return rc;
    }
    if (dx && nd == 2 && (rc = ww_gemm_seg2(mode, (float *)(d[0].ws + L.gi), (float *)(d[1].ws + L.gi), 3 * GH, M, d[0].w_ih, d[1].w_ih, I, I,
                                            3 * GH, dx, lddx, accumulate_dx, st, 1)))
        return rc;
    return WW_OK;
}
}  // namespace

extern "C" int ww_gru_fwd(ww_ctx *ctx, int mode, const float *x, long ldx, const float *w_ih, const float *w_hh, const float *b_ih,
                          const float *b_hh, const float *h0, int B, int T, int I, int H, int reverse, float *y, long ldy,
                          float *h_n, void *ws, size_t ws_bytes, ww_stream_t stream) {
    int rc = check_gru("ww_gru_fwd", ctx, B, T, I, H, ws, ws_bytes);
    if (rc) return rc;
    WW_REQUIRE(mode == WW_ACT_F32 || mode == WW_ACT_BF16 || mode == WW_ACT_F16, WW_E_INVALID, "ww_gru_fwd: unknown mode %d", mode);
    WW_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && y, WW_E_INVALID, "ww_gru_fwd: null argument");
    WW_REQUIRE(ldx >= I && ldy >= H, WW_E_INVALID, "ww_gru_fwd: row strides smaller than the feature sizes");
    ww_prof_scope ps_(ctx, WW_K_GRU, (hipStream_t)stream);
    const FwdDirHost d{w_ih, w_hh, b_ih, b_hh, h0, y, h_n, (char *)ws, reverse};
    return gru_layer_fwd(ctx, mode, x, ldx, &d, 1, B, T, I, ldy, (hipStream_t)stream);
}

extern "C" int ww_gru_bwd(ww_ctx *ctx, int mode, const float *x, long ldx, const float *w_ih, const float *w_hh, const float *dy,
                          long ldy, const float *dh_n, int B, int T, int I, int H, int reverse, void *ws, size_t ws_bytes,
                          float *dx, long lddx, int accumulate_dx, float *dw_ih, float *dw_hh, float *db_ih, float *db_hh,
                          float *dh0, ww_stream_t stream) {
    int rc = check_gru("ww_gru_bwd", ctx, B, T, I, H, ws, ws_bytes);
    if (rc) return rc;
    WW_REQUIRE(mode == WW_ACT_F32 || mode == WW_ACT_BF16 || mode == WW_ACT_F16, WW_E_INVALID, "ww_gru_bwd: unknown mode %d", mode);
    WW_REQUIRE(x && w_ih && w_hh && dw_ih && dw_hh && db_ih && db_hh, WW_E_INVALID, "ww_gru_bwd: null argument");
    WW_REQUIRE(dy || dh_n, WW_E_INVALID, "ww_gru_bwd: need dy and/or dh_n");
    WW_REQUIRE(!dx || lddx >= I, WW_E_INVALID, "ww_gru_bwd: dx row stride smaller than the input size");
    ww_prof_scope ps_(ctx, WW_K_GRU, (hipStream_t)stream);
    const BwdDirHost d{w_ih, w_hh, dy, dh_n, (char *)ws, dw_ih, dw_hh, db_ih, db_hh, dh0, reverse};
    return gru_layer_bwd(ctx, mode, x, ldx, &d, 1, ldy, B, T, I, dx, lddx, accumulate_dx, (hipStream_t)stream);
}

// Both directions of a bidirectional layer: dir[0] runs t = 0..T-1, dir[1] t = T-1..0; y / dy are (B,T,2H) buffers (row stride
// ldy >= 2H) whose column halves belong to the two directions; ONE recurrent launch (gridDim.y = 2) per pass.
extern "C" int ww_gru_bidir_fwd(ww_ctx *ctx, int mode, const float *x, long ldx, const ww_gru_dir *dir, int B, int T, int I, int H,
                                float *y, long ldy, size_t ws_bytes, ww_stream_t stream) {
    WW_REQUIRE(ctx && dir, WW_E_INVALID, "ww_gru_bidir_fwd: null argument");
    WW_REQUIRE(mode == WW_ACT_F32 || mode == WW_ACT_BF16 || mode == WW_ACT_F16, WW_E_INVALID, "ww_gru_bidir_fwd: unknown mode %d", mode);
    WW_REQUIRE(x && y && ldx >= I && ldy >= 2 * H, WW_E_INVALID, "ww_gru_bidir_fwd: null x / y or row strides too small");
    FwdDirHost d[2];
    for (int k = 0; k < 2; ++k) {
        int rc = check_gru("ww_gru_bidir_fwd", ctx, B, T, I, H, dir[k].ws, ws_bytes);
        if (rc) return rc;
        WW_REQUIRE(dir[k].w_ih && dir[k].w_hh && dir[k].b_ih && dir[k].b_hh, WW_E_INVALID, "ww_gru_bidir_fwd: null parameter");
        d[k] = FwdDirHost{dir[k].w_ih, dir[k].w_hh, dir[k].b_ih, dir[k].b_hh, dir[k].h0, y + (size_t)k * GH, dir[k].h_n,
                          (char *)dir[k].ws, k};
    }
    WW_REQUIRE(d[0].ws != d[1].ws, WW_E_INVALID, "ww_gru_bidir_fwd: the two directions need their own workspaces");
    ww_prof_scope ps_(ctx, WW_K_GRU, (hipStream_t)stream);
    return gru_layer_fwd(ctx, mode, x, ldx, d, 2, B, T, I, ldy, (hipStream_t)stream);
}

extern "C" int ww_gru_bidir_bwd(ww_ctx *ctx, int mode, const float *x, long ldx, const ww_gru_dir *dir, const float *dy, long ldy,
                                int B, int T, int I, int H, size_t ws_bytes, float *dx, long lddx, ww_stream_t stream) {
    WW_REQUIRE(ctx && dir && x, WW_E_INVALID, "ww_gru_bidir_bwd: null argument");
    WW_REQUIRE(mode == WW_ACT_F32 || mode == WW_ACT_BF16 || mode == WW_ACT_F16, WW_E_INVALID, "ww_gru_bidir_bwd: unknown mode %d", mode);
    WW_REQUIRE(!dy || ldy >= 2 * H, WW_E_INVALID, "ww_gru_bidir_bwd: dy row stride smaller than 2H");
    WW_REQUIRE(!dx || lddx >= I, WW_E_INVALID, "ww_gru_bidir_bwd: dx row stride smaller than the input size");
    BwdDirHost d[2];
    for (int k = 0; k < 2; ++k) {
        int rc = check_gru("ww_gru_bidir_bwd", ctx, B, T, I, H, dir[k].ws, ws_bytes);
        if (rc) return rc;
        WW_REQUIRE(dir[k].w_ih && dir[k].w_hh && dir[k].dw_ih && dir[k].dw_hh && dir[k].db_ih && dir[k].db_hh, WW_E_INVALID,
                   "ww_gru_bidir_bwd: null parameter / gradient pointer");
        WW_REQUIRE(dy || dir[k].dh_n, WW_E_INVALID, "ww_gru_bidir_bwd: need dy and/or dh_n");
        d[k] = BwdDirHost{dir[k].w_ih, dir[k].w_hh, dy ? dy + (size_t)k * GH : nullptr, dir[k].dh_n, (char *)dir[k].ws, dir[k].dw_ih,
                          dir[k].dw_hh, dir[k].db_ih, dir[k].db_hh, dir[k].dh0, k};
    }
    ww_prof_scope ps_(ctx, WW_K_GRU, (hipStream_t)stream);
    return gru_layer_bwd(ctx, mode, x, ldx, d, 2, dy ? ldy : 2 * GH, B, T, I, dx, lddx, 0, (hipStream_t)stream);
}
