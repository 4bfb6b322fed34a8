// ww_gemm16_nt: C[M][N] = A[M][K] . B[N][K]^T with 16-bit operands (bf16 / fp16) that already LIVE in HBM as 16-bit
// tensors, fp32 accumulation on v_mfma_f32_32x32x16_{bf16,f16}, C as fp32 or as the operand type.  This is the GEMM core
// of VERDICT r01 item 8 / SURVEY.md §8b K8: ww_linear.hip's k_gemm takes fp32 operands and rounds them while it fills LDS
// (the models keep fp32 activations around their dense layers), which caps it near 280 TFLOP/s; a caller that keeps its
// activations and weights in 16 bits (16-bit storage modes) comes here.
//
// Block = 128 x 128 outputs, 4 wavefronts as 2 x 2, each 64 x 64 = four 32x32 accumulators (64 VGPRs); K is walked in steps
// of 64.  Operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), two
// LDS buffers, the DMA of step t+1 in flight under the MFMAs of step t, ONE barrier per step.  An LDS-DMA writes a wave's
// 64 x 16 bytes contiguously, so the image is 8 unpadded 128-byte rows per wave-instruction; bank conflicts are removed on
// the SOURCE side instead: lane l fetches the 16-byte chunk (l & 7) ^ ((row >> 1) & 7) of its row, so that a fragment read
// (ds_read_b128, rows r .. r+31, one chunk) sees 16 different (row & 1, swizzled chunk) bank slots in every 16-lane group.
// blockIdx -> tile is XCD-aware: the 8 column tiles that share an A row panel run on ONE XCD (its L2 fetches the panel once).
#include "ww_internal.h"
#include "ww_layers.h"
#include "ww_act.h"

namespace {

constexpr int GB_M = 128, GB_N = 128, GB_K = 64;
constexpr int TILE_BYTES = GB_M * GB_K * 2;          // one operand tile: 16 KB

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// DMA one 128 x 64 operand tile (rows row0.., k0..k0+63 of a [rows][ld] matrix) into LDS: 16 wave-instructions of 1 KB, 4 per
// wavefront.  Rows past the matrix edge are clamped (their products land in outputs that are never stored).
template <typename H>
__device__ __forceinline__ void dma_tile(const H *__restrict__ src, long row0, long nrows, long ld, long k0,
                                         unsigned char *lds_tile, int wv, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wv * 4 + i;                     // 8 rows each
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        long gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        const H *g = src + gr * ld + k0 + 8 * chunk;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

template <typename H>
__device__ __forceinline__ typename H16<H>::x8 frag(const unsigned char *lds_tile, int row, int chunk) {
    return *reinterpret_cast<const typename H16<H>::x8 *>(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <typename H, typename CT>
__global__ __launch_bounds__(256, 2) void k_gemm16_nt(const H *__restrict__ A, const H *__restrict__ B, CT *__restrict__ C,
                                                      long M, long N, long K, int tiles_m, int tiles_n,
                                                      const float *__restrict__ bias /* per column, nullable */) {
    extern __shared__ __align__(1024) unsigned char lds[];     // [2 buffers][A tile | B tile]
    typedef typename H16<H>::x8 x8;
    typedef typename H16<H>::acc16 acc16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware mapping: consecutive block ids go round-robin over the 8 XCDs; give each XCD a contiguous range of tiles,
    // column tile fastest, so the blocks that share an A panel share an L2
    const int ntiles = tiles_m * tiles_n;
    int tile = blockIdx.x;
    {
        const int per = ntiles / 8;
        if (per * 8 == ntiles) tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const long m0 = (long)tm * GB_M, n0 = (long)tn * GB_N;

    acc16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (int)(K / GB_K);
    dma_tile<H>(A, m0, M, K, 0, lds, wv, lane);
    dma_tile<H>(B, n0, N, K, 0, lds + TILE_BYTES, wv, lane);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        unsigned char *cur = lds + (t & 1) * 2 * TILE_BYTES, *nxt = lds + ((t + 1) & 1) * 2 * TILE_BYTES;
        if (t + 1 < nk) {
            dma_tile<H>(A, m0, M, K, (long)(t + 1) * GB_K, nxt, wv, lane);
            dma_tile<H>(B, n0, N, K, (long)(t + 1) * GB_K, nxt + TILE_BYTES, wv, lane);
        }
        const unsigned char *at = cur, *bt = cur + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const x8 a0 = frag<H>(at, 64 * wm + r, 2 * ks + h), a1 = frag<H>(at, 64 * wm + 32 + r, 2 * ks + h);
            const x8 b0 = frag<H>(bt, 64 * wn + r, 2 * ks + h), b1 = frag<H>(bt, 64 * wn + 32 + r, 2 * ks + h);
            acc[0][0] = H16<H>::mfma32(a0, b0, acc[0][0]);
            acc[0][1] = H16<H>::mfma32(a0, b1, acc[0][1]);
            acc[1][0] = H16<H>::mfma32(a1, b0, acc[1][0]);
            acc[1][1] = H16<H>::mfma32(a1, b1, acc[1][1]);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // the next tile has landed
        __syncthreads();
    }
    // D layout of a 32x32 tile: lane (r, h) holds column r, rows (e & 3) + 8 (e >> 2) + 4 h
    const bool full = m0 + GB_M <= M && n0 + GB_N <= N;       // block-uniform: interior tiles store without bounds checks
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = n0 + 64 * wn + 32 * j + r;
            CT *cp = C + (m0 + 64 * wm + 32 * i + 4 * h) * N + col;
            const float bv = (bias && col < N) ? bias[col] : 0.f;
            if (full) {
#pragma unroll
                for (int e = 0; e < 16; ++e) cp[(long)((e & 3) + 8 * (e >> 2)) * N] = (CT)(acc[i][j][e] + bv);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long row = m0 + 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row < M && col < N) C[row * N + col] = (CT)(acc[i][j][e] + bv);
                }
            }
        }
}

template <typename H, typename CT>
int launch(const void *A, const void *B, void *C, long M, long N, long K, const float *bias, hipStream_t st) {
    const int tiles_m = (int)((M + GB_M - 1) / GB_M), tiles_n = (int)((N + GB_N - 1) / GB_N);
    const size_t smem = 4 * TILE_BYTES;
    // every call: the attribute belongs to the (function, device) pair and a process may drive several devices
    WW_HIP(hipFuncSetAttribute((const void *)k_gemm16_nt<H, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((k_gemm16_nt<H, CT>), dim3(tiles_m * tiles_n), dim3(256), smem, st, (const H *)A, (const H *)B, (CT *)C, M,
                       N, K, tiles_m, tiles_n, bias);
    WW_LAUNCH_CHECK();
    return WW_OK;
}

}  // namespace

// the entry point's body, with the optional per-column bias of the callers inside the library (GRU input projections)
int ww_gemm16_nt_bias(ww_ctx *ctx, int dtype, const void *A, const void *B, void *C, int c_f32, long M, long N, long K,
                      const float *bias, hipStream_t st) {
    WW_REQUIRE(ctx && A && B && C, WW_E_INVALID, "ww_gemm16_nt: null argument");
    WW_REQUIRE(dtype == WW_ACT_BF16 || dtype == WW_ACT_F16, WW_E_INVALID, "ww_gemm16_nt: dtype %d is not a 16-bit type", dtype);
    WW_REQUIRE(M >= 1 && N >= 1 && K >= 1, WW_E_INVALID, "ww_gemm16_nt: bad shape (%ld,%ld,%ld)", M, N, K);
    WW_REQUIRE(K % GB_K == 0, WW_E_UNSUPPORTED, "ww_gemm16_nt: K=%ld must be a multiple of %d", K, GB_K);
    WW_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, WW_E_INVALID, "ww_gemm16_nt: operands must be 16-byte aligned");
    WW_REQUIRE((M + GB_M - 1) / GB_M * ((N + GB_N - 1) / GB_N) < (1L << 31), WW_E_UNSUPPORTED, "ww_gemm16_nt: too many tiles");
    ww_prof_scope ps_(ctx, WW_K_LINEAR, st);
    if (dtype == WW_ACT_BF16)
        return c_f32 ? launch<ww_bf16, float>(A, B, C, M, N, K, bias, st) : launch<ww_bf16, ww_bf16>(A, B, C, M, N, K, bias, st);
    return c_f32 ? launch<ww_f16, float>(A, B, C, M, N, K, bias, st) : launch<ww_f16, ww_f16>(A, B, C, M, N, K, bias, st);
}

extern "C" int ww_gemm16_nt(ww_ctx *ctx, int dtype, const void *A, const void *B, void *C, int c_f32, long M, long N, long K,
                            ww_stream_t stream) {
    return ww_gemm16_nt_bias(ctx, dtype, A, B, C, c_f32, M, N, K, nullptr, (hipStream_t)stream);
}
