// Shared device/host helpers for the DiffBIR gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dbir.h"

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef unsigned short u16;

// ---- 16-bit element traits: storage is always raw u16; math in f32 ---------------------------
struct F16 {
  typedef f16x8 vec8;
  static __device__ __forceinline__ float to_f32(u16 v) { return (float)__builtin_bit_cast(f16, v); }
  static __device__ __forceinline__ u16 from_f32(float v) { return __builtin_bit_cast(u16, (f16)v); }
  // two f32 -> packed 16-bit pair (a in the low half), round-to-nearest-even: one v_cvt_pk_f16_f32 on gfx950
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
struct BF16 {
  typedef bf16x8 vec8;
  static __device__ __forceinline__ float to_f32(u16 v) { return __builtin_bit_cast(float, ((uint32_t)v) << 16); }
  static __device__ __forceinline__ u16 from_f32(float v) { return __builtin_bit_cast(u16, (__bf16)v); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {  // one v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const u16* h = reinterpret_cast<const u16*>(&v);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = T::to_f32(h[i]);
}
template <typename T>
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(T::pack2(f[0], f[1]), T::pack2(f[2], f[3]), T::pack2(f[4], f[5]), T::pack2(f[6], f[7]));
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host-side error plumbing -----------------------------------------------------------------
void dbir_set_error(const char* fmt, ...);
#define DBIR_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      dbir_set_error(__VA_ARGS__);       \
      return DBIR_ERR_ARG;               \
    }                                    \
  } while (0)
#define DBIR_CHECK_LAUNCH(name)                                             \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      dbir_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return DBIR_ERR_LAUNCH;                                               \
    }                                                                       \
  } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
