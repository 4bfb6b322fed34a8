// Dense layers on the matrix cores (ww_linear_mfma_{fwd,bwd}; K8 = the 576 -> 1024 -> num_classes head
// the reference puts on MobileNetV3, src/models/architectures.py:105-111: Linear, Hardswish, Dropout, Linear).
//   fwd : pre = x W^T + b ;  y = dropout(act(pre))
//   bwd : dpre = dy * dropout * act'(pre) ;  dx = dpre W ;  dW = dpre^T x ;  db = colsum(dpre)
// One tiled GEMM kernel serves all three products: C[i][j] = sum_k A(i,k) B(j,k) with either operand read K-contiguous
// (nn.Linear's own layouts for the forward) or transposed on the way into LDS (dx, dW).  64x64 block tile, 4 wavefronts x
// one 32x32 MFMA tile, K staged 32 at a time.  mode WW_ACT_F32: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulation: the parity mode); WW_ACT_BF16: operands rounded to bf16 at LDS-fill time, v_mfma_f32_32x32x16_bf16
// (what fp16/bf16 autocast does to nn.Linear; 16x the fp32 matrix rate).  All tensors are fp32 in HBM.
// Dropout after the activation draws from the same Philox stream as the cnn_small classifier dropout (TAG_DROPOUT,
// 4 features per draw, global sample index), so the oracle reproduces the mask exactly.
#include "ww_internal.h"
#include <type_traits>
#include "ww_layers.h"
#include "ww_act.h"
#include <algorithm>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
// matrix mode of a kernel: 0 = fp32 MFMA, 1 = bf16, 2 = fp16 operands (fp32 accumulation)
template <int MODE> struct ModeH { typedef ww_bf16 type; };
template <> struct ModeH<2> { typedef ww_f16 type; };

// block tiles are 64 or 128 rows/columns (template parameters TM, TN of k_gemm)
constexpr int GK = 32;      // K per LDS stage, fp32 mode
constexpr int GKH = 128;    // K per LDS stage, bf16 mode

struct GemmOperand {
    const float *p;
    long s_row, s_k;    // element (i,k) = p[i*s_row + k*s_k]
    int rows;           // valid rows
    int h16;            // 16-bit modes only: the elements in memory are ALREADY of the kernel's matrix type (p points at them; the
                        // strides count elements) -- half the bytes of an operand the staging would round to that type anyway
    const float *p2;    // SEG kernels: element (i, k) for k >= kseg is p2[i*s_row + (k - kseg)*s_k] -- a contraction over two
    int kseg;           // buffers (the two directions of a recurrent layer) in one product; kseg a multiple of the K stage
};
struct Epilogue {
    const float *bias;          // per column j (nullable)
    float *pre;                 // pre-activation out (nullable)
    int act;                    // WW_LIN_*
    int use_dropout;
    float drop_scale;
    uint64_t drop_thresh;
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
    uint64_t sample_offset;
    const ww_step_ctl *ctl;
    int accumulate;             // C += result (second direction of a bidirectional layer adds into dX)
    float *stat_part;           // nullable: per row tile [sum (N) | sum of squares (N)] of the stored outputs (BatchNorm partials)
};

__device__ __forceinline__ float lin_act(int act, float z) {
    if (act == WW_LIN_HARDSWISH) return z * fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
    if (act == WW_LIN_RELU) return z < 0.f ? 0.f : z;
    if (act == WW_LIN_HARDSIGMOID) return fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
    return z;
}
__device__ __forceinline__ float lin_act_grad(int act, float z) {      // torch's hardswish / hardsigmoid / relu backward
    if (act == WW_LIN_HARDSWISH) return z < -3.f ? 0.f : (z <= 3.f ? z * (1.f / 3.f) + 0.5f : 1.f);
    if (act == WW_LIN_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == WW_LIN_HARDSIGMOID) return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f;
    return 1.f;
}
// The launch-constant Philox inputs of a dropout epilogue (resolved step, seed).  k_gemm keeps them in VECTOR registers on purpose
// (dropout_ctx_vgpr): as wave-uniform values the compiler hoists the whole 10-round key schedule into ~20 SGPRs, which on top of the
// GEMM's live kernel arguments overflowed the scalar file (20-60 spilled SGPRs and a scratch segment in the r02 build).
struct DropCtx { uint32_t slo, shi, klo, khi; };
__device__ __forceinline__ DropCtx dropout_ctx(const Epilogue &e) {
    DropCtx d;
    ww_step_resolve(e.ctl, e.step_lo, e.step_hi, d.slo, d.shi);
    d.klo = e.seed_lo; d.khi = e.seed_hi;
    return d;
}
__device__ __forceinline__ DropCtx dropout_ctx_vgpr(const Epilogue &e) {
    const DropCtx u = dropout_ctx(e);
    DropCtx d;
    asm volatile("v_mov_b32 %0, %1" : "=v"(d.slo) : "s"(u.slo));
    asm volatile("v_mov_b32 %0, %1" : "=v"(d.shi) : "s"(u.shi));
    asm volatile("v_mov_b32 %0, %1" : "=v"(d.klo) : "s"(u.klo));
    asm volatile("v_mov_b32 %0, %1" : "=v"(d.khi) : "s"(u.khi));
    return d;
}
__device__ __forceinline__ bool drop_keep(const Epilogue &e, const DropCtx &d, long row, int col) {
    uint32_t rr[4];
    ww_philox(d.slo, d.shi, (uint32_t)(e.sample_offset + (uint64_t)row), (WW_TAG_DROPOUT << 24) | (uint32_t)(col >> 2), d.klo, d.khi, rr);
    const int q = col & 3;
    const uint32_t rv = q == 0 ? rr[0] : q == 1 ? rr[1] : q == 2 ? rr[2] : rr[3];
    return (uint64_t)rv >= e.drop_thresh;
}

// ---- operand staging.  A tile is ROWS x 32 k (ROWS = 64 or 128).  It is kept in LDS in the operand's own memory orientation,
// so both the global reads (float4) and the LDS writes are contiguous:  KC (k contiguous in memory): [row][k];  otherwise:
// [k][row].  Each thread moves ROWS/32 float4 per tile; the next tile's float4s are fetched before the MFMAs of the current one.
template <typename H, bool KC, int ROWS, int KT, bool SEG = false>
__device__ __forceinline__ void fetch_tile(const GemmOperand &op_, long r0, int k0_, int K_, bool vec_ok,
                                           float4 (&v)[ROWS * KT / 1024]) {
    const int tid = threadIdx.x;
    constexpr bool h16 = sizeof(H) == 2;             // H = the element type in memory: float, or the kernel's 16-bit matrix type
    // SEG: a stage lies in one of the two buffers (kseg is a multiple of the stage depth): pick it once per stage
    GemmOperand op = op_;
    int k0 = k0_, K = K_;
    if constexpr (SEG) {
        if (k0_ >= op_.kseg) { op.p = op_.p2; k0 = k0_ - op_.kseg; K = K_ - op_.kseg; }
        else K = min(K_, op_.kseg);
    }
#pragma unroll
    for (int j = 0; j < ROWS * KT / 1024; ++j) {
        const int q = tid + 256 * j;                 // float4 index within the tile (ROWS * KT / 4 per tile)
        long row; int k;
        if (KC) { row = r0 + q / (KT / 4); k = k0 + 4 * (q % (KT / 4)); }            // KT/4 float4 per row
        else { k = k0 + q / (ROWS / 4); row = r0 + 4 * (q % (ROWS / 4)); }           // ROWS/4 float4 per k-line
        const long lim_c = KC ? K : op.rows, c = KC ? k : row;         // the contiguous coordinate and its bound
        const bool other_ok = KC ? row < op.rows : k < K;
        const long eoff = row * op.s_row + (long)k * op.s_k, st = KC ? op.s_k : op.s_row;
        if constexpr (h16) {
            // the four elements stay in their 16-bit form (as bit patterns in .x / .y): stage_tile<.., RAW> stores them as they are
            const H *src = reinterpret_cast<const H *>(op.p) + eoff;
            uint2 raw;
            if (other_ok && vec_ok && c + 3 < lim_c) {
                raw = Act<H>::ldraw4(src);
            } else {
                const H z = (H)0.f;
                const H e0 = (other_ok && c < lim_c) ? src[0] : z, e1 = (other_ok && c + 1 < lim_c) ? src[st] : z;
                const H e2 = (other_ok && c + 2 < lim_c) ? src[2 * st] : z, e3 = (other_ok && c + 3 < lim_c) ? src[3 * st] : z;
                raw = make_uint2((uint32_t)__builtin_bit_cast(uint16_t, e0) | ((uint32_t)__builtin_bit_cast(uint16_t, e1) << 16),
                                 (uint32_t)__builtin_bit_cast(uint16_t, e2) | ((uint32_t)__builtin_bit_cast(uint16_t, e3) << 16));
            }
            v[j] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), 0.f, 0.f);
            continue;
        }
        const float *src = op.p + eoff;
        if (other_ok && vec_ok && c + 3 < lim_c) {
            v[j] = *reinterpret_cast<const float4 *>(src);
        } else {
            v[j].x = (other_ok && c < lim_c) ? src[0] : 0.f;
            v[j].y = (other_ok && c + 1 < lim_c) ? src[st] : 0.f;
            v[j].z = (other_ok && c + 2 < lim_c) ? src[2 * st] : 0.f;
            v[j].w = (other_ok && c + 3 < lim_c) ? src[3 * st] : 0.f;
        }
    }
}
// LDS strides: [row][k]: KT + 4 (fp32) / KT + 8 (bf16);  [k][row]: ROWS + 4 (fp32) / ROWS + 8 (bf16)
template <int MODE, bool KC, int ROWS, int KT, bool RAW = false>
__device__ __forceinline__ void stage_tile(void *lds, const float4 (&v)[ROWS * KT / 1024]) {
    constexpr bool BF16 = MODE != 0;
    typedef typename ModeH<MODE>::type H;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS * KT / 1024; ++j) {
        const int q = tid + 256 * j;
        const int off = KC ? (q / (KT / 4)) * (BF16 ? KT + 8 : KT + 4) + 4 * (q % (KT / 4))
                           : (q / (ROWS / 4)) * (BF16 ? ROWS + 8 : ROWS + 4) + 4 * (q % (ROWS / 4));
        if (BF16 && RAW) {
            *reinterpret_cast<uint2 *>(reinterpret_cast<H *>(lds) + off) = make_uint2(__float_as_uint(v[j].x), __float_as_uint(v[j].y));
        } else if (BF16) {
            typedef Act<H> A16;
            *reinterpret_cast<uint2 *>(reinterpret_cast<H *>(lds) + off) =
                make_uint2(A16::pack2(v[j].x, v[j].y), A16::pack2(v[j].z, v[j].w));
        } else {
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(lds) + off) = v[j];
        }
    }
}
typedef short short4v __attribute__((ext_vector_type(4)));
// bf16 MFMA operand (rows row0 + lane&31, k = k0 + 8*(lane>>5) .. +7) from a [k][row] tile of stride ld: transposing LDS reads
template <typename H>
__device__ __forceinline__ typename H16<H>::x8 tr_frag(const H *tile, int ld, int k0, int row0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const H *base = tile + (k0 + 8 * (g >> 1) + q) * ld + row0 + 16 * (g & 1) + 4 * pp;
    typedef short4v __attribute__((address_space(3))) * lds_p;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * ld));
    return __builtin_bit_cast(typename H16<H>::x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// Block tile (64*TM) x 64: 4 wavefronts as 2 x 2, each TM 32x32 MFMA tiles stacked along the rows (TM = 2 for tall
// problems: twice the MFMA work per staged B byte).  grid: x = column tiles, y = row tiles, z = K splits (partial
// products go to C + z*split_stride)
// EPI: 0 = plain store; 1 = the dense head's epilogue (bias, pre-activation copy, activation, Philox dropout, accumulate, BatchNorm
// partials); 2 = the light one (bias, accumulate, BatchNorm partials) -- what the 1x1 convolutions and the recurrent layers' input
// projections use.  The full epilogue costs ~60 SGPRs of live kernel arguments; compiled into the 18 conv launches of a MobileNetV3
// step it made the kernel spill SGPRs and carry a scratch segment (r02: 36 B, i.e. scratch set-up at every dispatch).
template <int MODE, bool KCA, bool KCB, int TM, int TN, int KS>
struct GemmLds {
    static constexpr bool BF16 = MODE != 0;
    static constexpr int RA = 64 * TM, RB = 64 * TN;
    static constexpr int KT = KS ? KS : (BF16 ? (TM * TN == 4 ? 64 : GKH) : GK);
    static constexpr int A_BYTES = BF16 ? (KCA ? RA * (KT + 8) : KT * (RA + 8)) * 2 : (KCA ? RA * (KT + 4) : KT * (RA + 4)) * 4;
    static constexpr int B_BYTES = BF16 ? (KCB ? RB * (KT + 8) : KT * (RB + 8)) * 2 : (KCB ? RB * (KT + 4) : KT * (RB + 4)) * 4;
    static constexpr int BYTES = A_BYTES + B_BYTES;
};
// one block tile of the product: block (bx, by) of split bz (of nz); lds >= GemmLds<...>::BYTES, 16-byte aligned
template <int MODE, bool KCA, bool KCB, int EPI, int TM, int TN, int KS, bool A16 = false, bool SEG = false>
__device__ __forceinline__ void gemm_block(const GemmOperand &A, const GemmOperand &B, int K, int k_per_split, float *__restrict__ C,
                                           long ldc, long split_stride, int vecA, int vecB, const Epilogue &e, unsigned char *lds,
                                           int bx, int by, int bz, int nz) {
    constexpr bool BF16 = MODE != 0;
    typedef typename ModeH<MODE>::type H;
    typedef typename std::conditional<BF16 && A16, H, float>::type FA;    // A's element type in memory (A16: already the matrix type)
    typedef typename H16<H>::x8 bf16x8;
    constexpr int RA = 64 * TM, RB = 64 * TN;
    // K per LDS stage: the bf16 MFMA eats 16 k per instruction, so a 32-deep stage is two MFMAs between barrier pairs --
    // 128 gives eight (64 for the 128 x 128 tile, whose prefetch registers double); the fp32 MFMA eats 2 k: 32 is sixteen
    // KS = 32: the shallow-K form for contractions of <= 32 (a 128-deep stage of a K = 9 / 16 / 24 product -- the MobileNetV3 stem and
    // its first expansions -- is 75-93 % zero padding that is still fetched, staged and multiplied: 90 us for the stem's 39 MB)
    constexpr int KT = KS ? KS : (BF16 ? (TM * TN == 4 ? 64 : GKH) : GK);
    constexpr int LDH_KC = KT + 8, LDF_KC = KT + 4;
    constexpr int A_BYTES = BF16 ? (KCA ? RA * LDH_KC : KT * (RA + 8)) * 2 : (KCA ? RA * LDF_KC : KT * (RA + 4)) * 4;
    constexpr int B_BYTES = BF16 ? (KCB ? RB * LDH_KC : KT * (RB + 8)) * 2 : (KCB ? RB * LDF_KC : KT * (RB + 4)) * 4;
    static_assert(A_BYTES + B_BYTES == GemmLds<MODE, KCA, KCB, TM, TN, KS>::BYTES, "LDS plan out of step");
    void *As = lds, *Bs = lds + A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5, rh = wv >> 1, nh = wv & 1;
    const long m0 = (long)by * RA, n0 = (long)bx * RB;
    const int kb = bz * k_per_split, ke = min(K, kb + k_per_split);
    floatx16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = floatx16{0.f};
    float4 va[RA * KT / 1024], vb[RB * KT / 1024];
    if (kb < ke) {
        fetch_tile<FA, KCA, RA, KT, SEG>(A, m0, kb, ke, vecA, va);
        fetch_tile<float, KCB, RB, KT, SEG>(B, n0, kb, ke, vecB, vb);
    }
    for (int k0 = kb; k0 < ke; k0 += KT) {
        __syncthreads();
        stage_tile<MODE, KCA, RA, KT, (BF16 && A16)>(As, va);
        stage_tile<MODE, KCB, RB, KT>(Bs, vb);
        __syncthreads();
        if (k0 + KT < ke) {                        // next tile in flight under the MFMAs
            fetch_tile<FA, KCA, RA, KT, SEG>(A, m0, k0 + KT, ke, vecA, va);
            fetch_tile<float, KCB, RB, KT, SEG>(B, n0, k0 + KT, ke, vecB, vb);
        }
        if (BF16) {
            const H *a = reinterpret_cast<const H *>(As), *b = reinterpret_cast<const H *>(Bs);
#pragma unroll
            for (int t = 0; t < KT / 16; ++t) {
                bf16x8 fb[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col0 = 32 * TN * nh + 32 * tn;
                    fb[tn] = KCB ? *reinterpret_cast<const bf16x8 *>(b + (col0 + r) * LDH_KC + 16 * t + 8 * h)
                                 : tr_frag(b, RB + 8, 16 * t, col0, lane);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int row0 = 32 * TM * rh + 32 * tm;
                    const bf16x8 fa = KCA ? *reinterpret_cast<const bf16x8 *>(a + (row0 + r) * LDH_KC + 16 * t + 8 * h)
                                          : tr_frag(a, RA + 8, 16 * t, row0, lane);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = H16<H>::mfma32(fa, fb[tn], acc[tm][tn]);
                }
            }
        } else {
            const float *a = reinterpret_cast<const float *>(As), *b = reinterpret_cast<const float *>(Bs);
#pragma unroll
            for (int t = 0; t < KT / 2; ++t) {
                float fb[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col0 = 32 * TN * nh + 32 * tn;
                    fb[tn] = KCB ? b[(col0 + r) * LDF_KC + 2 * t + h] : b[(2 * t + h) * (RB + 4) + col0 + r];
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int row0 = 32 * TM * rh + 32 * tm;
                    const float fa = KCA ? a[(row0 + r) * LDF_KC + 2 * t + h] : a[(2 * t + h) * (RA + 4) + row0 + r];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb[tn], acc[tm][tn], 0, 0, 0);
                }
            }
        }
    }
    C += (long)bz * split_stride;
    DropCtx dctx = {};
    if (EPI == 1 && e.use_dropout) dctx = dropout_ctx(e);
    float ssum[TN], qsum[TN];       // EPI && e.stat_part: this lane's column sums of what it stores
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
    ssum[tn] = 0.f; qsum[tn] = 0.f;
    const long col = n0 + 32 * TN * nh + 32 * tn + r;
    if (col >= B.rows) continue;
    const float bias = (EPI && e.bias) ? e.bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        // C += : the tile's 16 old values are loaded as ONE batch (clamped rows, unconditional) before the first store -- as
        // `v += C[..]` inside the guarded store loop every element was its own load -> wait -> add -> store round trip
        // (the accumulating dX product of a bidirectional GRU layer: 116 us against 47 for the plain one)
        float cold[16];
        if (EPI == 2 && e.accumulate && nz == 1) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long row = min(m0 + 32 * TM * rh + 32 * tm + (reg & 3) + 8 * (reg >> 2) + 4 * h, (long)A.rows - 1);
                cold[reg] = C[row * ldc + col];
            }
            // (all 16 live at once: behind the guarded stores the compiler would sink each load next to its use again)
            asm volatile("" : "+v"(cold[0]), "+v"(cold[1]), "+v"(cold[2]), "+v"(cold[3]), "+v"(cold[4]), "+v"(cold[5]), "+v"(cold[6]),
                         "+v"(cold[7]), "+v"(cold[8]), "+v"(cold[9]), "+v"(cold[10]), "+v"(cold[11]), "+v"(cold[12]), "+v"(cold[13]),
                         "+v"(cold[14]), "+v"(cold[15]));
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) cold[reg] = 0.f;
        }
        // dropout decisions of the tile's 16 elements as a bit mask, from a loop that is NOT unrolled: sixteen inlined Philox
        // rounds in the store loop below overflowed the scalar register file (spilled SGPRs and a 36-byte scratch segment,
        // i.e. scratch set-up at every dispatch of the head's GEMM)
        uint32_t keep = 0xFFFFu;
        if (EPI == 1 && e.use_dropout) {
            keep = 0u;
#pragma nounroll
            for (int reg = 0; reg < 16; ++reg) {
                const long row = m0 + 32 * TM * rh + 32 * tm + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                keep |= (drop_keep(e, dctx, row, (int)col) ? 1u : 0u) << reg;
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long row = m0 + 32 * TM * rh + 32 * tm + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row >= A.rows) continue;
            float v = acc[tm][tn][reg];
            if (EPI) v += bias;
            if (EPI == 1) {
                if (e.pre) e.pre[row * ldc + col] = v;
                v = lin_act(e.act, v);
                if (e.use_dropout) v = ((keep >> reg) & 1u) ? v * e.drop_scale : 0.f;
            }
            if (EPI == 1 && e.accumulate && nz == 1) v += C[row * ldc + col];
            if (EPI == 2) v += cold[reg];
            C[row * ldc + col] = v;
            if (EPI == 2) { ssum[tn] += v; qsum[tn] = fmaf(v, v, qsum[tn]); }
        }
    }
    }
    // BatchNorm statistics of a 1x1 convolution ride on its epilogue: the tile's column sums (the two half-waves by a shuffle,
    // the two row waves through LDS, fixed order) -> stat_part[row tile][2N]; the layer needs no pass over y for them
    if (EPI == 2 && e.stat_part) {
        __syncthreads();                               // every wave is done with the operand tiles
        float *sred = reinterpret_cast<float *>(lds);  // [4 waves][TN][32][2]
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const float sv = ssum[tn] + __shfl_xor(ssum[tn], 32), qv = qsum[tn] + __shfl_xor(qsum[tn], 32);
            if (h == 0) { sred[((wv * TN + tn) * 32 + r) * 2] = sv; sred[((wv * TN + tn) * 32 + r) * 2 + 1] = qv; }
        }
        __syncthreads();
        if (tid < RB) {
            const int nh_ = tid / (32 * TN), tn_ = (tid >> 5) % TN, r_ = tid & 31;
            const long col = n0 + tid;
            if (col < B.rows) {
                const float *lo = sred + (((0 * 2 + nh_) * TN + tn_) * 32 + r_) * 2, *hi = sred + (((1 * 2 + nh_) * TN + tn_) * 32 + r_) * 2;
                float *o = e.stat_part + (size_t)by * 2 * B.rows;
                o[col] = lo[0] + hi[0];
                o[B.rows + col] = lo[1] + hi[1];
            }
        }
    }
}

template <int MODE, bool KCA, bool KCB, int EPI, int TM, int TN, int KS = 0, bool A16 = false, bool SEG = false>
__global__ __launch_bounds__(256) void k_gemm(GemmOperand A, GemmOperand B, int K, int k_per_split, float *__restrict__ C,
                                              long ldc, long split_stride, int vecA, int vecB, Epilogue e) {
    __shared__ __align__(16) unsigned char lds[GemmLds<MODE, KCA, KCB, TM, TN, KS>::BYTES];
    gemm_block<MODE, KCA, KCB, EPI, TM, TN, KS, A16, SEG>(A, B, K, k_per_split, C, ldc, split_stride, vecA, vecB, e, lds, blockIdx.x, blockIdx.y,
                                                blockIdx.z, gridDim.z);
}

// dX and dW of one 1x1 convolution / Linear in ONE launch: the two products are independent (both read dy), each alone is a
// 6-14 us chain on a fraction of the CUs, and a kernel boundary between them buys nothing.  Blocks [0, nbw) run the split-K
// weight-gradient tiles (partials: the sum is deferred or follows), blocks [nbw, nbw + nbx) the data-gradient tiles.
struct PairSide { GemmOperand A, B; int K, kps; float *C; long ldc, sstride; int vecA, vecB, gx, gy, nz; };
template <int MODE, int TMX, int KSX>
__global__ __launch_bounds__(256) void k_gemm_pair(PairSide w, PairSide x, int nbw) {
    typedef GemmLds<MODE, false, false, 1, 1, 0> LW;
    typedef GemmLds<MODE, true, false, TMX, 1, KSX> LX;
    __shared__ __align__(16) unsigned char lds[LW::BYTES > LX::BYTES ? LW::BYTES : LX::BYTES];
    const Epilogue none = {};
    int b = blockIdx.x;
    if (b < nbw) {
        const int bx = b % w.gx;
        b /= w.gx;
        gemm_block<MODE, false, false, 0, 1, 1, 0>(w.A, w.B, w.K, w.kps, w.C, w.ldc, w.sstride, w.vecA, w.vecB, none, lds, bx, b % w.gy,
                                                   b / w.gy, w.nz);
    } else {
        b -= nbw;
        gemm_block<MODE, true, false, 0, TMX, 1, KSX>(x.A, x.B, x.K, x.kps, x.C, x.ldc, x.sstride, x.vecA, x.vecB, none, lds, b % x.gx,
                                                      b / x.gx, 0, 1);
    }
}

// C[i] = sum_z P[z][i] in fixed order (split-K partial products)
__global__ __launch_bounds__(256) void k_splitk_sum(const float *__restrict__ P, long n, int splits, float *__restrict__ C,
                                                    int accumulate) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            float4 s = *reinterpret_cast<const float4 *>(P + i);
#pragma unroll 8
            for (int z = 1; z < splits; ++z) {
                const float4 t = *reinterpret_cast<const float4 *>(P + (long)z * n + i);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            if (accumulate) {
                const float4 c = *reinterpret_cast<const float4 *>(C + i);
                s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
            }
            *reinterpret_cast<float4 *>(C + i) = s;
        } else {
            for (long j = i; j < n; ++j) {
                float s = P[j];
                for (int z = 1; z < splits; ++z) s += P[(long)z * n + j];
                C[j] = accumulate ? C[j] + s : s;
            }
        }
    }
}

// dpre = dy * dropout * act'(pre)
__global__ __launch_bounds__(256) void k_linear_dpre(const float *__restrict__ dy, const float *__restrict__ pre, long M, int N,
                                                     Epilogue e, float *__restrict__ dpre) {
    const long n = M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / N;
        const int col = (int)(i - row * N);
        float g = dy[i];
        if (e.use_dropout) g = drop_keep(e, dropout_ctx(e), row, col) ? g * e.drop_scale : 0.f;
        if (e.act != WW_LIN_NONE) g *= lin_act_grad(e.act, pre[i]);
        dpre[i] = g;
    }
}

// db[j] = sum_i dpre[i][j], any N; fixed summation order (16 row parts, then parts in order), fp64
__global__ __launch_bounds__(1024) void k_colsum_any(const float *__restrict__ a, int rows, int cols, float *__restrict__ out) {
    __shared__ double sh[16][64];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
    double acc = 0.0;
    if (col < cols) {
#pragma unroll 8
        for (int r = part; r < rows; r += 16) acc += (double)a[(size_t)r * cols + col];
    }
    sh[part][c] = acc;
    __syncthreads();
    if (part == 0 && col < cols) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += sh[p][c];
        out[col] = (float)t;
    }
}

// partial column sums of a row chunk: grid (column groups of 64, chunks); part[chunk][cols]
__global__ __launch_bounds__(1024) void k_colsum_chunk(const float *__restrict__ a, long rows, int cols, long rows_per_chunk,
                                                       float *__restrict__ part) {
    __shared__ double sh[16][64];
    const int c = threadIdx.x & 63, p = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
    const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    double acc = 0.0;
    if (col < cols) {
#pragma unroll 8
        for (long r = r0 + p; r < r1; r += 16) acc += (double)a[r * cols + col];
    }
    sh[p][c] = acc;
    __syncthreads();
    if (p == 0 && col < cols) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][c];
        part[(long)blockIdx.y * cols + col] = (float)t;
    }
}

int make_epilogue(const ww_ctx *ctx, const ww_linear_epi *epi, const float *bias, float *pre, Epilogue *out) {
    Epilogue e = {};
    e.ctl = ctx ? ctx->step_ctl : nullptr;
    e.bias = bias;
    e.pre = pre;
    if (epi) {
        WW_REQUIRE(epi->act >= WW_LIN_NONE && epi->act <= WW_LIN_HARDSIGMOID, WW_E_INVALID, "ww_linear_mfma: unknown activation %d", epi->act);
        WW_REQUIRE(epi->dropout_p >= 0.f && epi->dropout_p < 1.f, WW_E_INVALID, "ww_linear_mfma: dropout_p=%f not in [0,1)",
                   (double)epi->dropout_p);
        e.act = epi->act;
        e.use_dropout = epi->dropout_p > 0.f;
        e.drop_scale = (float)(1.0 / (1.0 - (double)epi->dropout_p));
        e.drop_thresh = ww_prob_threshold((double)epi->dropout_p);
        e.seed_lo = (uint32_t)epi->seed; e.seed_hi = (uint32_t)(epi->seed >> 32);
        e.step_lo = (uint32_t)epi->step; e.step_hi = (uint32_t)(epi->step >> 32);
        e.sample_offset = epi->sample_offset;
    }
    *out = e;
    return WW_OK;
}

// splits > 1: partial products into `part` (splits x rows x cols), then summed in fixed order into C
template <bool KCA, bool KCB, int EPI, bool A16 = false, bool SEG = false>
int launch_gemm(int mode, const GemmOperand &A, const GemmOperand &B, int K, float *C, long ldc, const Epilogue &e,
                hipStream_t st, int splits = 1, float *part = nullptr, int *row_tile_out = nullptr, ww_ctx *defer_ctx = nullptr) {
    auto aligned = [](const GemmOperand &o, bool kc) {
        const long ld = kc ? o.s_row : o.s_k;
        return (int)((((uintptr_t)o.p | (uintptr_t)o.p2) & (o.h16 ? 7 : 15)) == 0 && (ld & 3) == 0);
    };
    const int vecA = aligned(A, KCA), vecB = aligned(B, KCB);
    int kps = K;
    const int kt = mode != WW_ACT_F32 ? GKH : GK;      // a multiple of every stage depth in use (128, 64, 32)
    if (splits > 1) kps = ((K + splits - 1) / splits + kt - 1) / kt * kt;
    const int nz = (K + kps - 1) / kps;
    // Block tile: 64 x 64 by default; 128 x 128 (each staged byte feeds twice the MFMA work: these GEMMs are bound by operand
    // traffic, not by the matrix cores) when both extents allow it and enough workgroups remain; 128 x 64 for very tall ones
    const long tiles128 = (long)((A.rows + 127) / 128) * ((B.rows + 127) / 128) * nz;
    // (the 128 x 128 form pays in bf16 mode only: in fp32 mode its 64 accumulator + 32 prefetch registers cost occupancy, 72 vs 82 TF)
    // (the dense head's full epilogue -- EPI = 1 -- stays off the 128 x 128 form: with 64 accumulators beside the Philox dropout
    // the kernel spilled scalar registers and carried a scratch segment; the head's GEMMs are far too small to miss that tile)
    const int cfg = (EPI != 1 && mode != WW_ACT_F32 && A.rows >= 128 && B.rows >= 128 && tiles128 >= 256) ? 2 : (A.rows >= 8192 ? 1 : 0);
    const int RA = cfg ? 128 : 64, RBt = cfg == 2 ? 128 : 64;
    if (row_tile_out) *row_tile_out = RA;
    dim3 grid((B.rows + RBt - 1) / RBt, (A.rows + RA - 1) / RA, nz);
    float *dst = nz > 1 ? part : C;
    const long sstride = (long)A.rows * ldc;
#define WW_GEMM_LAUNCH(BF, TM_, TN_) \
    do { if constexpr (!(EPI == 1 && TM_ * TN_ == 4)) \
        hipLaunchKernelGGL((k_gemm<BF, KCA, KCB, EPI, TM_, TN_, 0, (A16 && BF != 0), SEG>), grid, dim3(256), 0, st, A, B, K, kps, dst, ldc, sstride, vecA, vecB, e); } while (0)
#define WW_GEMM_LAUNCH_K32(BF, TM_) \
    hipLaunchKernelGGL((k_gemm<BF, KCA, KCB, EPI, TM_, 1, 32, (A16 && BF != 0), SEG>), grid, dim3(256), 0, st, A, B, K, kps, dst, ldc, sstride, vecA, vecB, e)
    const bool shallow = K <= 32 && nz == 1 && cfg != 2;       // 16-bit modes: one 32-deep stage instead of a 128-deep one
    if (mode == WW_ACT_BF16) {
        if (shallow) { if (cfg == 1) WW_GEMM_LAUNCH_K32(1, 2); else WW_GEMM_LAUNCH_K32(1, 1); }
        else if (cfg == 2) WW_GEMM_LAUNCH(1, 2, 2); else if (cfg == 1) WW_GEMM_LAUNCH(1, 2, 1); else WW_GEMM_LAUNCH(1, 1, 1);
    } else if (mode == WW_ACT_F16) {
        if (shallow) { if (cfg == 1) WW_GEMM_LAUNCH_K32(2, 2); else WW_GEMM_LAUNCH_K32(2, 1); }
        else if (cfg == 2) WW_GEMM_LAUNCH(2, 2, 2); else if (cfg == 1) WW_GEMM_LAUNCH(2, 2, 1); else WW_GEMM_LAUNCH(2, 1, 1);
    } else {
        if (cfg == 2) WW_GEMM_LAUNCH(0, 2, 2); else if (cfg == 1) WW_GEMM_LAUNCH(0, 2, 1); else WW_GEMM_LAUNCH(0, 1, 1);
    }
#undef WW_GEMM_LAUNCH
#undef WW_GEMM_LAUNCH_K32
    WW_LAUNCH_CHECK();
    if (nz > 1 && !(ldc == B.rows && ww_defer(defer_ctx, part, C, sstride, nz, e.accumulate))) {
        const long n = sstride;
        const int g = (int)std::min<long>((n / 4 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL(k_splitk_sum, dim3(g), dim3(256), 0, st, part, n, nz, C, e.accumulate);
        WW_LAUNCH_CHECK();
    }
    return WW_OK;
}

// ---- the paired launch of ww_linear_mfma_bwd: dW (N x K) = dpre^T x split over the M rows, and dX (M x K) = dpre W
struct GemmPlan { int cfg, kps, nz; bool shallow; };
GemmPlan gemm_plan(int mode, const GemmOperand &A, const GemmOperand &B, int K, int splits) {      // launch_gemm's choices
    GemmPlan p;
    p.kps = K;
    const int kt = mode != WW_ACT_F32 ? GKH : GK;
    if (splits > 1) p.kps = ((K + splits - 1) / splits + kt - 1) / kt * kt;
    p.nz = (K + p.kps - 1) / p.kps;
    const long tiles128 = (long)((A.rows + 127) / 128) * ((B.rows + 127) / 128) * p.nz;
    p.cfg = (mode != WW_ACT_F32 && A.rows >= 128 && B.rows >= 128 && tiles128 >= 256) ? 2 : (A.rows >= 8192 ? 1 : 0);
    p.shallow = K <= 32 && p.nz == 1 && p.cfg != 2 && mode != WW_ACT_F32;
    return p;
}
// returns 1 when the pair was launched (dW partials in `part` when nz > 1, their sum queued on ctx or launched), 0 when the
// shapes want tile forms the paired kernel does not carry (the caller then issues the two products separately), < 0 on error
int launch_gemm_pair(ww_ctx *ctx, int mode, const float *dpre, const float *x, const float *w, int M, int K, int N, float *dx,
                     float *dw, int splits, float *part, hipStream_t st) {
    static const int enabled = ww_env_int("WW_GEMM_PAIR", 1);
    if (!enabled) return 0;
    const GemmOperand Ax{dpre, N, 1, M}, Bx{w, 1, K, K};            // dx[m][k] = sum_n dpre[m][n] w[n][k]
    const GemmOperand Aw{dpre, 1, N, N}, Bw{x, 1, K, K};            // dw[n][k] = sum_m dpre[m][n] x[m][k]
    GemmPlan px = gemm_plan(mode, Ax, Bx, N, 1);
    const GemmPlan pw = gemm_plan(mode, Aw, Bw, M, splits);
    if (px.cfg == 2 || pw.cfg != 0 || pw.shallow) return 0;
    // tall dX products (128-row tiles) with a deep K stage: the paired kernel's 128-row form needs 480 registers (one wave per
    // SIMD: 43 us against 31 for the two separate launches at M = 194 560) -- WW_GEMM_PAIR_TALL: 0 = launch them separately,
    // 1 = pair with 64-row dX tiles, 2 = pair with 128-row tiles
    static const int tall = ww_env_int("WW_GEMM_PAIR_TALL", 1);
    if (px.cfg == 1 && !px.shallow) {
        if (tall == 0) return 0;
        if (tall == 1) px.cfg = 0;
    }
    auto aligned = [](const GemmOperand &o, bool kc) {
        const long ld = kc ? o.s_row : o.s_k;
        return (int)((((uintptr_t)o.p | (uintptr_t)o.p2) & (o.h16 ? 7 : 15)) == 0 && (ld & 3) == 0);
    };
    const int RAx = px.cfg ? 128 : 64;
    PairSide sw{Aw, Bw, M, pw.kps, pw.nz > 1 ? part : dw, (long)K, (long)N * K, aligned(Aw, false), aligned(Bw, false),
                (K + 63) / 64, (N + 63) / 64, pw.nz};
    PairSide sx{Ax, Bx, N, N, dx, (long)K, 0, aligned(Ax, true), aligned(Bx, false), (K + 63) / 64, (M + RAx - 1) / RAx, 1};
    const int nbw = sw.gx * sw.gy * sw.nz, nbx = sx.gx * sx.gy;
    const dim3 grid(nbw + nbx);
#define WW_PAIR(MODE_) \
    do { \
        if (px.shallow) { if (px.cfg) hipLaunchKernelGGL((k_gemm_pair<MODE_, 2, 32>), grid, dim3(256), 0, st, sw, sx, nbw); \
                          else hipLaunchKernelGGL((k_gemm_pair<MODE_, 1, 32>), grid, dim3(256), 0, st, sw, sx, nbw); } \
        else if (px.cfg) hipLaunchKernelGGL((k_gemm_pair<MODE_, 2, 0>), grid, dim3(256), 0, st, sw, sx, nbw); \
        else hipLaunchKernelGGL((k_gemm_pair<MODE_, 1, 0>), grid, dim3(256), 0, st, sw, sx, nbw); \
    } while (0)
    if (mode == WW_ACT_BF16) WW_PAIR(1);
    else if (mode == WW_ACT_F16) WW_PAIR(2);
    else { if (px.cfg) hipLaunchKernelGGL((k_gemm_pair<0, 2, 0>), grid, dim3(256), 0, st, sw, sx, nbw);
           else hipLaunchKernelGGL((k_gemm_pair<0, 1, 0>), grid, dim3(256), 0, st, sw, sx, nbw); }
#undef WW_PAIR
    WW_LAUNCH_CHECK();
    if (pw.nz > 1 && !ww_defer(ctx, part, dw, (long)N * K, pw.nz, 0)) {
        const long n = (long)N * K;
        const int g = (int)std::min<long>((n / 4 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL(k_splitk_sum, dim3(g), dim3(256), 0, st, part, n, pw.nz, dw, 0);
        WW_LAUNCH_CHECK();
    }
    return 1;
}

int check_dims(const char *who, int mode, int M, int K, int N) {
    WW_REQUIRE(mode == WW_ACT_F32 || mode == WW_ACT_BF16 || mode == WW_ACT_F16, WW_E_INVALID, "%s: unknown mode %d", who, mode);
    WW_REQUIRE(M >= 1 && K >= 1 && N >= 1, WW_E_INVALID, "%s: bad shape M=%d K=%d N=%d", who, M, K, N);
    return WW_OK;
}

}  // namespace

// internal entry points for the recurrent layers (ww_gru.hip)
int ww_gemm(int mode, const float *A, long a_srow, long a_sk, int a_rows, const float *B, long b_srow, long b_sk, int b_rows,
            int K, float *C, long ldc, const float *bias, int accumulate, int splits, float *part, hipStream_t st, ww_ctx *defer_ctx,
            int a16) {
    Epilogue e = {};
    e.bias = bias;
    e.accumulate = accumulate;
    const GemmOperand a{A, a_srow, a_sk, a_rows, (a16 && mode != WW_ACT_F32) ? 1 : 0}, b{B, b_srow, b_sk, b_rows, 0};
    const bool kca = a_sk == 1, kcb = b_sk == 1;
    if (a.h16) {            // (the recurrent layers' dGi / dGh: never k-contiguous on both sides)
        if (kca && !kcb) return launch_gemm<true, false, 2, true>(mode, a, b, K, C, ldc, e, st, splits, part, nullptr, defer_ctx);
        if (!kca && !kcb) return launch_gemm<false, false, 2, true>(mode, a, b, K, C, ldc, e, st, splits, part, nullptr, defer_ctx);
        ww_set_error("ww_gemm: a 16-bit A operand is built for B with contiguous rows only");
        return WW_E_UNSUPPORTED;
    }
    if (kca && kcb) return launch_gemm<true, true, 2>(mode, a, b, K, C, ldc, e, st, splits, part, nullptr, defer_ctx);
    if (kca && !kcb) return launch_gemm<true, false, 2>(mode, a, b, K, C, ldc, e, st, splits, part, nullptr, defer_ctx);
    if (!kca && kcb) return launch_gemm<false, true, 2>(mode, a, b, K, C, ldc, e, st, splits, part, nullptr, defer_ctx);
    return launch_gemm<false, false, 2>(mode, a, b, K, C, ldc, e, st, splits, part, nullptr, defer_ctx);
}
// C[i][j] (+)= sum_{k < kseg} A(i,k) B(j,k) + sum_{k < kseg} A2(i,k) B2(j,k): one product over two buffer pairs of the same
// shapes and strides (the dX of a bidirectional recurrent layer: both directions' dGi against both W_ih) instead of a product
// and an accumulating one; A k-contiguous, B row-contiguous, kseg a multiple of 128
int ww_gemm_seg2(int mode, const float *A, const float *A2, long a_srow, int a_rows, const float *B, const float *B2, long b_sk,
                 int b_rows, int kseg, float *C, long ldc, int accumulate, hipStream_t st, int a16) {
    WW_REQUIRE(kseg % GKH == 0, WW_E_INVALID, "ww_gemm_seg2: kseg=%d must be a multiple of %d", kseg, GKH);
    Epilogue e = {};
    e.accumulate = accumulate;
    const GemmOperand a{A, a_srow, 1, a_rows, (a16 && mode != WW_ACT_F32) ? 1 : 0, A2, kseg}, b{B, 1, b_sk, b_rows, 0, B2, kseg};
    if (a.h16) return launch_gemm<true, false, 2, true, true>(mode, a, b, 2 * kseg, C, ldc, e, st);
    return launch_gemm<true, false, 2, false, true>(mode, a, b, 2 * kseg, C, ldc, e, st);
}
// a: rows x (2*cols) row-major; out0 = column sums of the left half, out1 of the right half (one launch)
__global__ __launch_bounds__(1024) void k_colsum_pair(const float *__restrict__ a, int rows, int cols, float *__restrict__ out0,
                                                      float *__restrict__ out1) {
    __shared__ double sh[16][64];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
    double acc = 0.0;
    if (col < 2 * cols) {
#pragma unroll 8
        for (int r = part; r < rows; r += 16) acc += (double)a[(size_t)r * (2 * cols) + col];
    }
    sh[part][c] = acc;
    __syncthreads();
    if (part == 0 && col < 2 * cols) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += sh[p][c];
        if (col < cols) out0[col] = (float)t; else out1[col - cols] = (float)t;
    }
}
// out[j] = sum_i a[i][j] for a short matrix (one launch, fixed order)
int ww_colsum_rows_small(const float *a, int rows, int cols, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_colsum_any, dim3((cols + 63) / 64), dim3(1024), 0, st, a, rows, cols, out);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
int ww_colsum_pair(const float *a, int rows, int cols, float *out0, float *out1, hipStream_t st) {
    hipLaunchKernelGGL(k_colsum_pair, dim3((2 * cols + 63) / 64), dim3(1024), 0, st, a, rows, cols, out0, out1);
    WW_LAUNCH_CHECK();
    return WW_OK;
}
// out[j] = sum_i a[i][j] for tall matrices: chunked partial sums (part: chunks x cols floats), then a fixed-order sum
int ww_colsum_rows(const float *a, long rows, int cols, float *out, float *part, int chunks, hipStream_t st) {
    const long rpc = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(k_colsum_chunk, dim3((cols + 63) / 64, chunks), dim3(1024), 0, st, a, rows, cols, rpc, part);
    WW_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_colsum_any, dim3((cols + 63) / 64), dim3(1024), 0, st, part, chunks, cols, out);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

extern "C" int ww_linear_mfma_fwd(ww_ctx *ctx, int mode, const float *x, const float *w, const float *bias, int M, int K,
                                  int N, const ww_linear_epi *epi, float *pre, float *y, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && y, WW_E_INVALID, "ww_linear_mfma_fwd: null argument");
    int rc = check_dims("ww_linear_mfma_fwd", mode, M, K, N);
    if (rc) return rc;
    Epilogue e;
    if ((rc = make_epilogue(ctx, epi, bias, pre, &e))) return rc;
    const GemmOperand A{x, K, 1, M}, B{w, K, 1, N};
    ww_prof_scope ps_(ctx, WW_K_LINEAR, (hipStream_t)stream);
    if (e.pre || e.act != WW_LIN_NONE || e.use_dropout) return launch_gemm<true, true, 1>(mode, A, B, K, y, N, e, (hipStream_t)stream);
    return launch_gemm<true, true, 2>(mode, A, B, K, y, N, e, (hipStream_t)stream);
}

// Conv2dNormActivation with a 1x1 (or im2col'ed) convolution in training mode: y = x W^T on the matrix cores with the BatchNorm
// statistics partials written by the GEMM's own epilogue, then the BatchNorm(+activation) apply pass that finishes them
// (ww_bn_act_from_partials): 2 launches (3 for very tall layers) instead of GEMM + statistics + finish + apply.  residual
// (nullable, (M,N)): the inverted-residual block's input, added to the activated output in the same pass (a = act(bn(y)) + residual).
extern "C" int ww_conv1x1_bn_act_fwd(ww_ctx *ctx, int mode, const float *x, const float *w, int M, int K, int N, const ww_bn_t *bn,
                                     int act, const float *residual, float *y, float *a, float *ss, float *mr, void *scratch,
                                     size_t scratch_bytes, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && bn && bn->gamma && bn->beta && y && a && ss && mr && scratch, WW_E_INVALID, "ww_conv1x1_bn_act_fwd: null argument");
    int rc = check_dims("ww_conv1x1_bn_act_fwd", mode, M, K, N);
    if (rc) return rc;
    WW_REQUIRE(bn->training, WW_E_INVALID, "ww_conv1x1_bn_act_fwd: training-mode statistics only (eval: ww_linear_mfma_fwd + ww_bn_act_fwd)");
    WW_REQUIRE(scratch_bytes >= (size_t)((M + 63) / 64) * 2 * N * sizeof(float), WW_E_WORKSPACE,
               "ww_conv1x1_bn_act_fwd: scratch too small for the statistics partials");
    hipStream_t st = (hipStream_t)stream;
    Epilogue e = {};
    e.act = WW_LIN_NONE;
    e.stat_part = (float *)scratch;
    const GemmOperand A{x, K, 1, M}, B{w, K, 1, N};
    int row_tile = 64;
    {
        ww_prof_scope ps_(ctx, WW_K_LINEAR, st);
        if ((rc = launch_gemm<true, true, 2>(mode, A, B, K, y, N, e, st, 1, nullptr, &row_tile))) return rc;
    }
    ww_prof_scope ps_(ctx, WW_K_NHWC, st);
    return ww_bn_act_from_partials(ctx, y, M, N, bn, act, a, ss, mr, (const float *)scratch, (M + row_tile - 1) / row_tile, residual, st);
}

// K splits of the weight-gradient product dW (N x K) = dpre^T x: its contraction runs over the M rows (batch x pixels), which
// for a 1x1 convolution is hundreds of thousands while N x K is a handful of 64 x 64 tiles -- the splits are what fills the
// chip: enough of them for ~1024 workgroups, each at least 512 rows deep
static int dw_splits(int M, int K, int N) {
    // rows of the contraction per split: ONE 128-deep K stage.  A split's time is a chain of dependent stages (fetch -> LDS ->
    // MFMA, ~3 us each at these sizes), so shallower splits finish sooner and the deferred reduction sums the extra partials for
    // free (MobileNetV3 B=256, WW_DW_MIN_ROWS = 512 / 256 / 128: 2.694 / 2.579 / 2.566 ms per step, profiles/r03_e_*)
    static const int min_rows = std::max(128, ww_env_int("WW_DW_MIN_ROWS", 128));
    const long tiles = (long)((N + 63) / 64) * ((K + 63) / 64);
    long s = (1024 + tiles - 1) / tiles;
    s = std::min<long>(s, M / min_rows);
    return (int)std::max<long>(1, std::min<long>(s, 256));
}
extern "C" size_t ww_linear_mfma_bwd_scratch_bytes(int M, int K, int N) {
    if (M < 1 || K < 1 || N < 1) return 0;
    return ((size_t)M * N + (size_t)dw_splits(M, K, N) * N * K) * sizeof(float);      // dpre + split-K partial products of dW
}

extern "C" int ww_linear_mfma_bwd(ww_ctx *ctx, int mode, const float *x, const float *w, const float *pre, const float *dy,
                                  int M, int K, int N, const ww_linear_epi *epi, float *dx, float *dw, float *db,
                                  void *scratch, size_t scratch_bytes, ww_stream_t stream) {
    WW_REQUIRE(ctx && x && w && dy && dw && scratch, WW_E_INVALID, "ww_linear_mfma_bwd: null argument");
    int rc = check_dims("ww_linear_mfma_bwd", mode, M, K, N);
    if (rc) return rc;
    WW_REQUIRE(scratch_bytes >= ww_linear_mfma_bwd_scratch_bytes(M, K, N), WW_E_WORKSPACE, "ww_linear_mfma_bwd: scratch too small");
    Epilogue e;
    if ((rc = make_epilogue(ctx, epi, nullptr, nullptr, &e))) return rc;
    WW_REQUIRE(e.act == WW_LIN_NONE || pre, WW_E_INVALID, "ww_linear_mfma_bwd: the activation's backward needs `pre`");
    hipStream_t st = (hipStream_t)stream;
    ww_prof_scope ps_(ctx, WW_K_LINEAR, st);
    const float *dpre = dy;
    if (e.act != WW_LIN_NONE || e.use_dropout) {
        const long n = (long)M * N;
        const int grid = (int)std::min<long>((n + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(k_linear_dpre, dim3(grid), dim3(256), 0, st, dy, pre, (long)M, N, e, (float *)scratch);
        WW_LAUNCH_CHECK();
        dpre = (const float *)scratch;
    }
    const Epilogue none = {};
    float *const part0 = (float *)scratch + (size_t)M * N;
    const int paired = dx ? launch_gemm_pair(ctx, mode, dpre, x, w, M, K, N, dx, dw, dw_splits(M, K, N), part0, st) : 0;
    if (paired < 0) return paired;
    if (dx && !paired) {   // dx[m][k] = sum_n dpre[m][n] w[n][k] : A = dpre (n contiguous), B(k, n) = w[n][k] (row index contiguous)
        const GemmOperand A{dpre, N, 1, M}, B{w, 1, K, K};
        if ((rc = launch_gemm<true, false, 0>(mode, A, B, N, dx, K, none, st))) return rc;
    }
    if (!paired) {   // dw[n][k] = sum_m dpre[m][n] x[m][k] : A(n, m) = dpre[m][n], B(k, m) = x[m][k]
        const GemmOperand A{dpre, 1, N, N}, B{x, 1, K, K};
        float *part = (float *)scratch + (size_t)M * N;
        if ((rc = launch_gemm<false, false, 0>(mode, A, B, M, dw, K, none, st, dw_splits(M, K, N), part, nullptr, ctx))) return rc;
    }
    if (db) {
        hipLaunchKernelGGL(k_colsum_any, dim3((N + 63) / 64), dim3(1024), 0, st, dpre, M, N, db);
        WW_LAUNCH_CHECK();
    }
    return WW_OK;
}
