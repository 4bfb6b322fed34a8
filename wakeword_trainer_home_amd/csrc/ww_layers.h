// Internal declarations shared by the generic channels-last layer library (ww_nhwc.hip) and the matrix-core GEMMs
// (ww_linear.hip) -- the MobileNetV3 body of SURVEY.md §8f rank 2.
#pragma once
#include "ww_internal.h"

// ww_nhwc.hip: training-mode BatchNorm(+activation) of x (M, C) whose producer already wrote `chunks` rows of statistics partials
// ([sum (C) | sum of squares (C)] each) to `part`: the apply pass finishes them itself when that is cheap, else finish + apply.
// res (nullable): a residual tensor of x's shape added to the activated output (the inverted-residual skip connection).
int ww_bn_act_from_partials(ww_ctx *ctx, const float *x, long M, int C, const ww_bn_t *bn, int act, float *y, float *ss, float *mr,
                            const float *part, int chunks, const float *res, hipStream_t st);

// ww_gemm16.hip: ww_gemm16_nt with an optional per-column fp32 bias added in the epilogue
int ww_gemm16_nt_bias(ww_ctx *ctx, int dtype, const void *A, const void *B, void *C, int c_f32, long M, long N, long K,
                      const float *bias, hipStream_t st);
