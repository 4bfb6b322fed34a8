// Storage traits of the conv-stack activation tensors (y_l, g_l): fp32 (parity mode) or bf16 (BASELINE
// config 2).  Arithmetic is always fp32; bf16 affects only what is written to / read from HBM.  Values are
// rounded (RNE, v_cvt_pk_bf16_f32) BEFORE statistics are taken, so BatchNorm statistics describe exactly the
// tensor the consumer will read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 ww_bf16;
typedef __bf16 ww_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ww_bf16x4 __attribute__((ext_vector_type(4)));
typedef float ww_f32x2 __attribute__((ext_vector_type(2)));
typedef float ww_f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Act;

template <> struct Act<float> {
    typedef float2 raw2;
    typedef float4 raw4;
    static constexpr bool is_f32 = true;
    static __device__ __forceinline__ raw2 ldraw2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
    static __device__ __forceinline__ raw4 ldraw4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ float2 cvt2(raw2 r) { return r; }
    static __device__ __forceinline__ float4 cvt4(raw4 r) { return r; }
    static __device__ __forceinline__ void st2(float *p, float2 v) { *reinterpret_cast<float2 *>(p) = v; }
    static __device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ float round1(float v) { return v; }
    static __device__ __forceinline__ float2 round2(float2 v) { return v; }
};

template <> struct Act<ww_bf16> {
    typedef uint32_t raw2;
    typedef uint2 raw4;
    static constexpr bool is_f32 = false;
    static __device__ __forceinline__ raw2 ldraw2(const ww_bf16 *p) { return *reinterpret_cast<const uint32_t *>(p); }
    static __device__ __forceinline__ raw4 ldraw4(const ww_bf16 *p) { return *reinterpret_cast<const uint2 *>(p); }
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
};
