// issue cost of the fp32 MFMA shapes on gfx950, one wave alone on its SIMD: s_memtime ticks per instruction for a dependent
// chain (one accumulator) and for four accumulators in rotation
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define N 512
template <int NACC> __global__ void k4x4(float a, float b, float *out, unsigned long long *t) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float x = a + threadIdx.x, y = b - threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[i % NACC], 0, 0, 0);
    f4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *t = t1 - t0;
}
template <int NACC> __global__ void k16(float a, float b, float *out, unsigned long long *t) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float x = a + threadIdx.x, y = b - threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i % NACC], 0, 0, 0);
    f4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *t = t1 - t0;
}
template <int NACC> __global__ void kfma(float a, float b, float *out, unsigned long long *t) {
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    float x = a + threadIdx.x, y = b - threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) { acc[i % NACC] = fmaf(x, y, acc[i % NACC]); asm volatile("" : "+v"(acc[i % NACC])); }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[threadIdx.x] = s;
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *t = t1 - t0;
}
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NACC> __global__ void kpk(float a, float b, float *out, unsigned long long *t) {
    f2 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f2{0.f, 0.f};
    f2 x = {a + threadIdx.x, a - threadIdx.x}, y = {b - threadIdx.x, b + threadIdx.x};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i % NACC]) : "v"(x), "v"(y));
    f2 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[threadIdx.x] = s[0] + s[1];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *t = t1 - t0;
}
template <int NACC> __global__ void kfma2(float a, float b, float *out, unsigned long long *t) {
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    float x = a + threadIdx.x, y = b - threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i)
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i % NACC]) : "v"(x), "v"(y));
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[threadIdx.x] = s;
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *t = t1 - t0;
}
int main() {
    float *o; unsigned long long *t, h;
    (void)hipMalloc(&o, 4096); (void)hipMalloc(&t, 8);
#define RUN(name, K) for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, 1.f, 2.f, o, t); (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); } \
    printf("%-34s %7.1f ticks per instruction\n", name, (double)h / N);
    RUN("v_fma_f32 chain", (kfma<1>)); RUN("v_fma_f32 x4 accumulators", (kfma<4>));
    RUN("v_fma_f32 (asm) chain", (kfma2<1>)); RUN("v_fma_f32 (asm) x8 accumulators", (kfma2<8>));
    RUN("v_pk_fma_f32 chain", (kpk<1>)); RUN("v_pk_fma_f32 x8 accumulators", (kpk<8>));
    RUN("mfma_f32_4x4x1 chain", (k4x4<1>)); RUN("mfma_f32_4x4x1 x4 accumulators", (k4x4<4>));
    RUN("mfma_f32_16x16x4 chain", (k16<1>)); RUN("mfma_f32_16x16x4 x4 accumulators", (k16<4>));
    // the same with 16 waves on the CU (4 per SIMD): SIMD throughput instead of one wave's issue interval
#define RUN16(name, K) for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(K, dim3(1), dim3(1024), 0, 0, 1.f, 2.f, o, t); (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); } \
    printf("%-34s %7.1f ticks per instruction per wave (4 waves per SIMD)\n", name, (double)h / N);
    RUN16("v_fma_f32 (asm) x8, 16 waves", (kfma2<8>)); RUN16("v_pk_fma_f32 x8, 16 waves", (kpk<8>));
    RUN16("mfma_f32_4x4x1 x4, 16 waves", (k4x4<4>));
    // every CU full: 512 blocks of 1024 threads = 8 waves per SIMD where the registers allow -- the SIMD's VALU capacity
    (void)hipFree(o); (void)hipMalloc(&o, 512 * 4096);
#define RUNFULL(name, K) for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(K, dim3(512), dim3(1024), 0, 0, 1.f, 2.f, o, t); (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); } \
    printf("%-34s %7.1f ticks per instruction per wave (8 waves per SIMD, all CUs)\n", name, (double)h / N);
    RUNFULL("v_fma_f32 (asm) x8, full", (kfma2<8>)); RUNFULL("v_pk_fma_f32 x8, full", (kpk<8>));
    return 0;
}
