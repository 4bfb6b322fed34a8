// operand layout probe of v_mfma_f32_4x4x1_16b_f32 on gfx950 (the mel band sums of k_logmel): 16 independent 4x4 outer products.
// Hypothesis checked: lane l = 4*block + i supplies A_block[i] and B_block[i]; D_block[r][c] lands in register r of lane 4*block + c.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *out) {
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + l], b[64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = acc[r];
}
int main() {
    float ha[128], hb[128], ho[256], *a, *b, *o;
    for (int i = 0; i < 128; ++i) { ha[i] = 1.f + 0.37f * i; hb[i] = 2.f - 0.11f * i; }
    (void)hipMalloc(&a, sizeof(ha)); (void)hipMalloc(&b, sizeof(hb)); (void)hipMalloc(&o, sizeof(ho));
    (void)hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
    (void)hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2, c = l & 3;
            const float want = fmaf(ha[64 + 4 * blk + r], hb[64 + 4 * blk + c], ha[4 * blk + r] * hb[4 * blk + c]);
            if (ho[r * 64 + l] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, ho[r * 64 + l], want); ++bad; }
        }
    printf("mfma_f32_4x4x1 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed, bit-exact with fmaf order", bad);
    return bad != 0;
}
