// Shared helpers for the gfx950 kernels of librdst_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/rdst_hip.h"

typedef __hip_bfloat16 bf16;

// Ablation switches, in-kernel cycle stamps and environment overrides exist only in a -DRDST_DEBUG build
// (RDST_BUILD_DEBUG=1 python -m rdst_amd.build; tools/ use it).  In the shipped library RDST_DBGV(x) is the
// constant 0 and rdst_dbg_getenv() a constant null pointer, so every `if (RDST_DBGV(p.dbg) & 2)` /
// `if (RDST_DBGV(p.stamps) && ...)` branch and every `e ? atoi(e) : dflt` folds away at compile time: no
// getenv, hipMalloc or host sync in a launcher, no ablation test inside a kernel loop.
#ifdef RDST_DEBUG
#include <stdlib.h>
#define RDST_DBGV(x) (x)
static inline const char* rdst_dbg_getenv(const char* name) { return getenv(name); }
// per-workgroup cycle counters of a kernel (debug build only): `env`=N arms the next N launches; the launcher passes
// the returned device buffer ([grid][n] u64, zeroed) to the kernel and calls rdst_stamps_end() after the launch, which
// synchronises, averages every slot over the workgroups and prints it.
static inline unsigned long long* rdst_stamps_begin(const char* env, int grid, int n, hipStream_t st) {
  const char* e = getenv(env);
  if (!e || atoi(e) <= 0) return nullptr;
  unsigned long long* d = nullptr;
  if (hipMalloc((void**)&d, (size_t)grid * n * 8) != hipSuccess) return nullptr;
  (void)hipMemsetAsync(d, 0, (size_t)grid * n * 8, st);
  return d;
}
static inline void rdst_stamps_end(const char* tag, unsigned long long* d, int grid, int n, hipStream_t st) {
  if (!d) return;
  (void)hipStreamSynchronize(st);
  unsigned long long* h = (unsigned long long*)malloc((size_t)grid * n * 8);
  (void)hipMemcpy(h, d, (size_t)grid * n * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  fprintf(stderr, "[stamps %s grid=%d] mean cycles per workgroup:", tag, grid);
  for (int k = 0; k < n; ++k) {
    double sum = 0;
    for (int w = 0; w < grid; ++w) sum += (double)h[(size_t)w * n + k];
    fprintf(stderr, " %d:%.0f", k, sum / grid);
  }
  fprintf(stderr, "\n");
  free(h);
}
#else
#define RDST_DBGV(x) 0
static inline constexpr const char* rdst_dbg_getenv(const char*) { return nullptr; }
static inline unsigned long long* rdst_stamps_begin(const char*, int, int, hipStream_t) { return nullptr; }
static inline void rdst_stamps_end(const char*, unsigned long long*, int, int, hipStream_t) {}
#endif

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

// RDST_F32X3 on the network entry points: fp32 tensors, GEMMs in the split arithmetic of mfma.h (Mma<float, true>).
// The entry point opens a SplitScope, which turns the dtype into RDST_F32 for everything below it and leaves the mode in
// a thread-local that the launchers of the converted kernels read (kernels without a split form run exact fp32).
extern thread_local int g_rdst_split;
struct SplitScope {
  int prev;
  explicit SplitScope(int& dtype) : prev(g_rdst_split) {
    g_rdst_split = dtype == RDST_F32X3 || (prev && dtype == RDST_F32);   // (an entry point that forwards to another one keeps the mode)
    if (dtype == RDST_F32X3) dtype = RDST_F32;
  }
  ~SplitScope() { g_rdst_split = prev; }
};
static inline bool rdst_split() { return g_rdst_split != 0; }

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
// GELU(erf) for the bf16 throughput mode: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below
// bf16 resolution) on the hardware exp2 / rcp — ~14 vector instructions instead of erff's ~40; the
// Linear kernels around a GELU were vector-issue bound on it (measured).  The fp32 parity mode keeps erff.
__device__ __forceinline__ void erf_as(float z, float& erfz, float& expmz2) {
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  pl *= t;
  expmz2 = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // exp(-z^2)
  erfz = copysignf(fmaf(-pl, expmz2, 1.0f), z);
}
__device__ __forceinline__ float gelu_fast(float x) {
  float e, g;
  erf_as(x * 0.70710678118654752440f, e, g);
  return 0.5f * x * (1.0f + e);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
  float e, g;
  erf_as(x * 0.70710678118654752440f, e, g);                       // g = exp(-x^2/2)
  return fmaf(x * 0.39894228040143267794f, g, 0.5f * (1.0f + e));
}
// GELU through a table in LDS (bf16 throughput mode, the fused Mlp kernels): 512 entries of four fp16 over [-8, 8),
//   entry i = (Phi(u_i), Phi(u_i+1) - Phi(u_i), g(u_i), g(u_i+1) - g(u_i)),  u_i = -8 + i/32,  g = GELU' = Phi + u phi,
// read with ONE ds_read_b64 and interpolated linearly by v_fma_mix_f32 (fp32 weight x fp16 slope + fp16 value, fp32
// result): interpolation error <= 3e-5 in Phi, 1e-4 in GELU', fp16 storage 2.4e-4 / 4.9e-4 — the results are then
// rounded to bf16 (2e-3) anyway — for ~8 vector instructions per element where the erf / exp2 / rcp form takes ~27:
// those kernels' GELU phases run at the vector-issue floor.  Outside [-8, 8) the clamped index
// gives 0 / 1.
#define RDST_GELU_TAB_BYTES 4096
__device__ __forceinline__ void gelu_tab_fill(char* tab, int tid, int nt) {
  for (int i = tid; i < 512; i += nt) {
    const float u0 = -8.f + (float)i * 0.03125f, u1 = u0 + 0.03125f;
    const float c0 = 0.5f * erfcf(-u0 * 0.70710678118654752440f), c1 = 0.5f * erfcf(-u1 * 0.70710678118654752440f);
    const float g0 = fmaf(u0 * 0.39894228040143267794f, expf(-0.5f * u0 * u0), c0);
    const float g1 = fmaf(u1 * 0.39894228040143267794f, expf(-0.5f * u1 * u1), c1);
    const _Float16 h0 = (_Float16)c0, h2 = (_Float16)g0;
    const _Float16 h1 = (_Float16)(c1 - (float)h0), h3 = (_Float16)(g1 - (float)h2);   // slopes from the ROUNDED values
    uint2 w;
    w.x = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
    w.y = (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16);
    reinterpret_cast<uint2*>(tab)[i] = w;
  }
}
// forward only: (Phi, slope) pairs alone, 4 bytes per entry (a b32 read of random entries spreads over all 64 banks)
#define RDST_GELU_TAB4_BYTES 2048
__device__ __forceinline__ void gelu_tab4_fill(char* tab, int tid, int nt) {
  for (int i = tid; i < 512; i += nt) {
    const float u0 = -8.f + (float)i * 0.03125f, u1 = u0 + 0.03125f;
    const float c0 = 0.5f * erfcf(-u0 * 0.70710678118654752440f), c1 = 0.5f * erfcf(-u1 * 0.70710678118654752440f);
    const _Float16 h0 = (_Float16)c0, h1 = (_Float16)(c1 - (float)h0);
    reinterpret_cast<uint32_t*>(tab)[i] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
  }
}
// (interpolation weight, byte offset of the entry)
__device__ __forceinline__ float gelu_tab_index(float u, uint32_t& off) {
  const float t = __builtin_amdgcn_fmed3f(fmaf(u, 32.f, 256.f), 0.f, 511.99f);
  off = (uint32_t)t << 3;
  return __builtin_amdgcn_fractf(t);
}
__device__ __forceinline__ float gelu_tab4_index(float u, uint32_t& off) {
  const float t = __builtin_amdgcn_fmed3f(fmaf(u, 32.f, 256.f), 0.f, 511.99f);
  off = (uint32_t)t << 2;
  return __builtin_amdgcn_fractf(t);
}
// value + weight * slope of one packed (value = low half, slope = high half) fp16 pair
__device__ __forceinline__ float gelu_tab_lerp(float w, uint32_t pair) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w), "v"(pair));
  return r;
}
template <bool FAST = false>
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == RDST_ACT_GELU) return FAST ? gelu_fast(x) : gelu_erf(x);
  if (act == RDST_ACT_LEAKY02) return x > 0.f ? x : 0.2f * x;
  if (act == RDST_ACT_LEAKY001) return x > 0.f ? x : 0.01f * x;
  return x;
}
template <bool FAST = false>
__device__ __forceinline__ float act_grad(float xpre, int act) {
  if (act == RDST_ACT_GELU) return FAST ? gelu_grad_fast(xpre) : gelu_erf_grad(xpre);
  if (act == RDST_ACT_LEAKY02) return xpre > 0.f ? 1.f : 0.2f;
  if (act == RDST_ACT_LEAKY001) return xpre > 0.f ? 1.f : 0.01f;
  return 1.f;
}
