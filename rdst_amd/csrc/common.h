// Shared helpers for the gfx950 kernels of librdst_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/rdst_hip.h"

typedef __hip_bfloat16 bf16;

// thread-local last-error text behind rdst_last_error()
extern thread_local char g_rdst_err[256];
static inline int rdst_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_rdst_err, sizeof(g_rdst_err), fmt, ap);
  va_end(ap);
  return code;
}
static inline int rdst_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return rdst_fail(-(int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return __float2bfloat16(v); }

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

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  // d/dx [0.5 x (1+erf(x/sqrt2))] = 0.5(1+erf(x/sqrt2)) + x * exp(-x^2/2)/sqrt(2pi)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == RDST_ACT_GELU) return gelu_erf(x);
  if (act == RDST_ACT_LEAKY02) return x > 0.f ? x : 0.2f * x;
  if (act == RDST_ACT_LEAKY001) return x > 0.f ? x : 0.01f * x;
  return x;
}
__device__ __forceinline__ float act_grad(float xpre, int act) {
  if (act == RDST_ACT_GELU) return gelu_erf_grad(xpre);
  if (act == RDST_ACT_LEAKY02) return xpre > 0.f ? 1.f : 0.2f;
  if (act == RDST_ACT_LEAKY001) return xpre > 0.f ? 1.f : 0.01f;
  return 1.f;
}
