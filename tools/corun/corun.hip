// Diagnostic co-runner (tools/corun_probe.py): a persistent kernel with the log-mel kernel's per-CU footprint (256 threads,
// ~120 VGPRs, 41 KB LDS, one workgroup per CU) that exercises ONE resource, to find out which one a conv kernel running
// beside it is sensitive to.  mode 0: hold the footprint and sleep; 1: LDS traffic (ds_read/ds_write b32); 2: VALU (fma chains);
// 3: vector-memory loads that hit L1/L2; 4: LDS + VALU mixed as an FFT round does.  Not part of libwwhip.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_corun(int mode, long iters, const float *__restrict__ src, float *__restrict__ out, unsigned *__restrict__ where) {
    extern __shared__ float lds[];                 // 41 KB
    const int tid = threadIdx.x;
    if (where && tid == 0) {      // which CU this workgroup landed on: HW_ID (wave/simd/pipe/cu/sh/se) and XCC_ID
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xff00);      // XCC | SE SH CU
    }
    float keep[96];                                // holds ~100 VGPRs live across the loop
#pragma unroll
    for (int i = 0; i < 96; ++i) keep[i] = (float)(tid + i);
    for (int i = tid; i < 41 * 256; i += 256) lds[i] = (float)i;
    __syncthreads();
    float acc0 = 0.f, acc1 = 1.f, acc2 = 2.f, acc3 = 3.f;
    for (long it = 0; it < iters; ++it) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(64);
        } else if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = lds[(tid + 64 * j + (int)it) & (41 * 256 - 1) & 8191];
                lds[(tid + 68 * j) & 8191] = v + 1.f;
                acc0 += v;
            }
        } else if (mode == 2) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                acc0 = fmaf(acc0, 1.0001f, 0.5f); acc1 = fmaf(acc1, 0.9999f, 0.25f);
                acc2 = fmaf(acc2, 1.0002f, 0.125f); acc3 = fmaf(acc3, 0.9998f, 0.75f);
            }
        } else if (mode == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc0 += src[(tid + 256 * j + (int)(it & 63) * 2048) & 0xffff];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = lds[(tid + 64 * j + (int)it) & 8191];
                acc0 = fmaf(acc0, 1.0001f, v); acc1 = fmaf(acc1, 0.9999f, v);
                acc2 = fmaf(acc2, 1.0002f, v); acc3 = fmaf(acc3, 0.9998f, v);
                lds[(tid + 68 * j) & 8191] = acc0;
            }
        }
    }
    float s = acc0 + acc1 + acc2 + acc3;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += keep[i];
    asm volatile("" ::"v"(s));
    if (s == 12345.678f) out[tid] = s;
}

extern "C" int corun_launch(int mode, long iters, int grid, const float *src, float *out, void *stream, unsigned *where) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)k_corun, hipFuncAttributeMaxDynamicSharedMemorySize, 41 * 1024) != hipSuccess) return 1;
        attr = true;
    }
    hipLaunchKernelGGL(k_corun, dim3(grid), dim3(256), 41 * 1024, (hipStream_t)stream, mode, iters, src, out, where);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
