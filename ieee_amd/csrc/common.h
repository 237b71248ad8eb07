// Shared declarations for the ieee_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/ieee_amd.h"

namespace ieee {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __bf16 bf16;

// error plumbing (abi.cpp)
void set_error(int code, const char* fmt, ...);
int launch_status(const char* what);

#define IEEE_REQUIRE(cond, ...)                                  \
  do {                                                           \
    if (!(cond)) {                                               \
      ieee::set_error(IEEE_ERR_BAD_ARG, __VA_ARGS__);            \
      return IEEE_ERR_BAD_ARG;                                   \
    }                                                            \
  } while (0)

#define IEEE_HIP(call)                                                        \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess) {                                                   \
      ieee::set_error(IEEE_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
      return IEEE_ERR_HIP;                                                    \
    }                                                                         \
  } while (0)

#define IEEE_TRY(call)            \
  do {                            \
    int s_ = (call);              \
    if (s_ != IEEE_OK) return s_; \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

template <typename T> struct DTypeOf;
template <> struct DTypeOf<float> { static constexpr int value = IEEE_F32; };
template <> struct DTypeOf<bf16> { static constexpr int value = IEEE_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 16-byte vector helpers -----------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
    // bf16 -> f32 is a 16-bit shift
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  __device__ static __forceinline__ uint32_t pk(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 t = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(uint32_t, t);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

}  // namespace ieee
