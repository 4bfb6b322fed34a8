// Storage traits of the conv-stack activation tensors (y_l, g_l): fp32 (parity mode), bf16 (BASELINE config 2) or fp16
// (BASELINE config 5; the reference's own reduced precision: fp16 autocast + GradScaler, src/training/trainer.py:172,
// 182-193 -- gradients are then stored times the loss scale).  Arithmetic is always fp32; the 16-bit types affect only
// what is written to / read from HBM (and the matrix-core operands).  Values are rounded (RNE) BEFORE statistics are
// taken, so BatchNorm statistics describe exactly the tensor the consumer will read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 ww_bf16;
typedef __bf16 ww_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ww_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ww_f16;
typedef _Float16 ww_f16x2 __attribute__((ext_vector_type(2)));
typedef float ww_f32x2 __attribute__((ext_vector_type(2)));
typedef float ww_f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Act;


template <> struct Act<float> {
    typedef float2 raw2;
    typedef float4 raw4;
    static constexpr bool is_f32 = true;
    static __device__ __forceinline__ raw2 ldraw2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
    static __device__ __forceinline__ raw4 ldraw4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ raw2 ldraw2_nt(const float *p) { return ldraw2(p); }
    static __device__ __forceinline__ raw4 ldraw4_nt(const float *p) { return ldraw4(p); }
    static __device__ __forceinline__ float2 cvt2(raw2 r) { return r; }
    static __device__ __forceinline__ float4 cvt4(raw4 r) { return r; }
    static __device__ __forceinline__ void st2(float *p, float2 v) { *reinterpret_cast<float2 *>(p) = v; }
    static __device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ float round1(float v) { return v; }
    static __device__ __forceinline__ float2 round2(float2 v) { return v; }
    static __device__ __forceinline__ float4 round4(float4 v) { return v; }
    static __device__ __forceinline__ void st8(float *p, float4 a, float4 b) {      // 8 consecutive elements
        *reinterpret_cast<float4 *>(p) = a;
        *reinterpret_cast<float4 *>(p + 4) = b;
    }
};

template <> struct Act<ww_bf16> {
    typedef uint32_t raw2;
    typedef uint2 raw4;
    static constexpr bool is_f32 = false;
    static __device__ __forceinline__ raw2 ldraw2(const ww_bf16 *p) { return *reinterpret_cast<const uint32_t *>(p); }
    static __device__ __forceinline__ raw4 ldraw4(const ww_bf16 *p) { return *reinterpret_cast<const uint2 *>(p); }
    // single-use streams (read once per step by this kernel, dead afterwards): non-temporal, so they do not displace the
    // lines the NEXT kernel will re-read from the Infinity Cache
    static __device__ __forceinline__ raw2 ldraw2_nt(const ww_bf16 *p) { return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p)); }
    static __device__ __forceinline__ raw4 ldraw4_nt(const ww_bf16 *p) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(q));
        return make_uint2(v.x, v.y);
    }
    static __device__ __forceinline__ float2 cvt2(raw2 r) {
        return make_float2(__uint_as_float(r << 16), __uint_as_float(r & 0xffff0000u));
    }
    static __device__ __forceinline__ float4 cvt4(raw4 r) {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        ww_f32x2 f = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, ww_bf16x2));
    }
    static __device__ __forceinline__ void st2(ww_bf16 *p, float2 v) { *reinterpret_cast<uint32_t *>(p) = pack2(v.x, v.y); }
    static __device__ __forceinline__ void st4(ww_bf16 *p, float4 v) {
        *reinterpret_cast<uint2 *>(p) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
    }
    static __device__ __forceinline__ float2 round2(float2 v) { return cvt2(pack2(v.x, v.y)); }
    static __device__ __forceinline__ float round1(float v) { return __uint_as_float(pack2(v, 0.f) << 16); }
    static __device__ __forceinline__ float4 round4(float4 v) { return cvt4(make_uint2(pack2(v.x, v.y), pack2(v.z, v.w))); }
    static __device__ __forceinline__ void st8(ww_bf16 *p, float4 a, float4 b) {    // 8 consecutive elements: one 16-byte store
        *reinterpret_cast<uint4 *>(p) = make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
    }
};

template <> struct Act<ww_f16> {
    typedef uint32_t raw2;
    typedef uint2 raw4;
    static constexpr bool is_f32 = false;
    static __device__ __forceinline__ raw2 ldraw2(const ww_f16 *p) { return *reinterpret_cast<const uint32_t *>(p); }
    static __device__ __forceinline__ raw4 ldraw4(const ww_f16 *p) { return *reinterpret_cast<const uint2 *>(p); }
    // single-use streams (read once per step by this kernel, dead afterwards): non-temporal, so they do not displace the
    // lines the NEXT kernel will re-read from the Infinity Cache
    static __device__ __forceinline__ raw2 ldraw2_nt(const ww_f16 *p) { return __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p)); }
    static __device__ __forceinline__ raw4 ldraw4_nt(const ww_f16 *p) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(q));
        return make_uint2(v.x, v.y);
    }
    static __device__ __forceinline__ float2 cvt2(raw2 r) {
        const ww_f32x2 f = __builtin_convertvector(__builtin_bit_cast(ww_f16x2, r), ww_f32x2);
        return make_float2(f.x, f.y);
    }
    static __device__ __forceinline__ float4 cvt4(raw4 r) {
        const float2 a = cvt2(r.x), b = cvt2(r.y);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {      // RNE; overflow -> inf (the loss scaler's signal)
        ww_f32x2 f = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, ww_f16x2));
    }
    static __device__ __forceinline__ void st2(ww_f16 *p, float2 v) { *reinterpret_cast<uint32_t *>(p) = pack2(v.x, v.y); }
    static __device__ __forceinline__ void st4(ww_f16 *p, float4 v) {
        *reinterpret_cast<uint2 *>(p) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
    }
    static __device__ __forceinline__ float2 round2(float2 v) { return cvt2(pack2(v.x, v.y)); }
    static __device__ __forceinline__ float round1(float v) { return (float)(ww_f16)v; }
    static __device__ __forceinline__ float4 round4(float4 v) { return cvt4(make_uint2(pack2(v.x, v.y), pack2(v.z, v.w))); }
    static __device__ __forceinline__ void st8(ww_f16 *p, float4 a, float4 b) {
        *reinterpret_cast<uint4 *>(p) = make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
    }
};

// matrix-core forms of the two 16-bit types (8-element operands, fp32 accumulation; same cycles per instruction)
template <typename H> struct H16;
template <> struct H16<ww_bf16> {
    typedef __bf16 x8 __attribute__((ext_vector_type(8)));
    typedef float acc16 __attribute__((ext_vector_type(16)));
    typedef float acc4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc16 mfma32(x8 a, x8 b, acc16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ acc4 mfma16(x8 a, x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct H16<ww_f16> {
    typedef _Float16 x8 __attribute__((ext_vector_type(8)));
    typedef float acc16 __attribute__((ext_vector_type(16)));
    typedef float acc4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc16 mfma32(x8 a, x8 b, acc16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ acc4 mfma16(x8 a, x8 b, acc4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <typename H> __device__ __forceinline__ typename H16<H>::x8 ww_pack8(const float (&v)[8]) {
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    return __builtin_convertvector(f, typename H16<H>::x8);
}
