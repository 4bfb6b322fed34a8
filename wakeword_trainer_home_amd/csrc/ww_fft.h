// Register-resident radix-4 / radix-16 DFT building blocks shared by the log-mel front end (ww_frontend.hip) and the
// FFT convolution of the waveform augmentation (ww_audio.hip).
#pragma once
#include <hip/hip_runtime.h>

#define C1 0.92387953251128674f
#define S1 0.38268343236508977f
#define R2 0.70710678118654752f

__device__ __forceinline__ void fft4(float &r0, float &i0, float &r1, float &i1, float &r2, float &i2, float &r3,
                                     float &i3) {
    const float ar = r0 + r2, ai = i0 + i2, br = r0 - r2, bi = i0 - i2;
    const float cr = r1 + r3, ci = i1 + i3, dr = r1 - r3, di = i1 - i3;
    r0 = ar + cr; i0 = ai + ci;
    r2 = ar - cr; i2 = ai - ci;
    r1 = br + di; i1 = bi - dr;
    r3 = br - di; i3 = bi + dr;
}

__device__ __forceinline__ void cmul_c(float &r, float &i, const float wr, const float wi) {
    const float t = r * wr - i * wi;
    i = r * wi + i * wr;
    r = t;
}

// 16-point DFT in registers.  Input natural order v[j]; output X[k] is left at v[(k>>2) + 4*(k&3)].
__device__ __forceinline__ void fft16(float (&re)[16], float (&im)[16]) {
#pragma unroll
    for (int j1 = 0; j1 < 4; ++j1)
        fft4(re[j1], im[j1], re[j1 + 4], im[j1 + 4], re[j1 + 8], im[j1 + 8], re[j1 + 12], im[j1 + 12]);
    // v[j1 + 4*ka] *= W16^(j1*ka)
    cmul_c(re[1 + 4], im[1 + 4], C1, -S1);    // 1*1
    cmul_c(re[1 + 8], im[1 + 8], R2, -R2);    // 1*2
    cmul_c(re[1 + 12], im[1 + 12], S1, -C1);  // 1*3
    cmul_c(re[2 + 4], im[2 + 4], R2, -R2);    // 2*1
    { const float t = re[2 + 8]; re[2 + 8] = im[2 + 8]; im[2 + 8] = -t; }  // 2*2 = 4 -> -i
    cmul_c(re[2 + 12], im[2 + 12], -R2, -R2); // 2*3 = 6
    cmul_c(re[3 + 4], im[3 + 4], S1, -C1);    // 3*1
    cmul_c(re[3 + 8], im[3 + 8], -R2, -R2);   // 3*2 = 6
    cmul_c(re[3 + 12], im[3 + 12], -C1, S1);  // 3*3 = 9
#pragma unroll
    for (int ka = 0; ka < 4; ++ka)
        fft4(re[4 * ka], im[4 * ka], re[4 * ka + 1], im[4 * ka + 1], re[4 * ka + 2], im[4 * ka + 2], re[4 * ka + 3],
             im[4 * ka + 3]);
}
// output slot of X[k]
#define F16_SLOT(k) ((((k) >> 2)) + 4 * ((k) & 3))
