// gfx950 MFMA building blocks shared by the GEMM-shaped kernels (linear / conv / attention).
//
// One abstraction covers both arithmetic modes: a "pack" is the 16 bytes one lane feeds to the matrix
// core per k-step (4 fp32 or 8 bf16).  With lane = (r = lane & 31, h = lane >> 5):
//   bf16: v_mfma_f32_32x32x16_bf16  — lane holds A[r][16t + 8h + j], j = 0..7     (one MFMA per pack)
//   fp32: v_mfma_f32_32x32x2_f32 x4 — lane holds A[r][ 8t + 4h + e], e = 0..3; MFMA e consumes
//         element e of both operands, i.e. physical k = 8t + 4h + e.  The k order is permuted the
//         same way for A and B, so the sum is unchanged; fp32 MFMA is an exact fmaf chain
//         (cdna_hip_programming.md §3), which is what the fp32 parity mode needs.
// In both cases a k-step covers 32 BYTES of a k-contiguous row and lane half h owns bytes
// [32t + 16h, +16) — so LDS tiles and fragment addressing are dtype-agnostic in bytes.
// C/D layout (both): col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16-B load, dword aligned
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

struct alignas(16) Pack16 {
  uint32_t w[4];
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

template <typename E> struct Mma;

template <> struct Mma<float> {
  static constexpr int KP = 8;  // k elements per k-step (both lane halves)
  static constexpr int HP = 4;  // elements per lane pack
  static __device__ __forceinline__ void mma(f32x16& acc, const Pack16& a, const Pack16& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w[e]), __uint_as_float(b.w[e]), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void unpack(const Pack16& p, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(p.w[e]);
  }
  static __device__ __forceinline__ Pack16 pack(const float* f) {
    Pack16 p;
#pragma unroll
    for (int e = 0; e < 4; ++e) p.w[e] = __float_as_uint(f[e]);
    return p;
  }
};

template <> struct Mma<bf16> {
  static constexpr int KP = 16;
  static constexpr int HP = 8;
  static __device__ __forceinline__ void mma(f32x16& acc, const Pack16& a, const Pack16& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc,
                                                  0, 0, 0);
  }
  static __device__ __forceinline__ void unpack(const Pack16& p, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = bf16lo(p.w[e]);
      f[2 * e + 1] = bf16hi(p.w[e]);
    }
  }
  static __device__ __forceinline__ Pack16 pack(const float* f) {
    Pack16 p;
#pragma unroll
    for (int e = 0; e < 4; ++e) p.w[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
    return p;
  }
};

// Guarded load of one lane pack (HP elements starting at element k0 of a row of K valid elements).
// Rows need only dword alignment (the dense-concat slices of an RDSTB start at 60/90/120 elements).
template <typename T>
__device__ __forceinline__ Pack16 load_pack(const T* __restrict__ rowp, int k0, int K, bool valid) {
  constexpr int HP = Mma<T>::HP;
  Pack16 p;
  if (valid && k0 + HP <= K && (reinterpret_cast<uintptr_t>(rowp + k0) & 3) == 0) {
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(rowp + k0);
    p.w[0] = v.x; p.w[1] = v.y; p.w[2] = v.z; p.w[3] = v.w;
    return p;
  }
  float f[HP];
#pragma unroll
  for (int e = 0; e < HP; ++e) f[e] = (valid && k0 + e < K) ? to_f32<T>(rowp[k0 + e]) : 0.f;
  return Mma<T>::pack(f);
}

// Pack HP fp32 parameters (converted to T) starting at src[k0*stride], guarded by k < K.
template <typename T>
__device__ __forceinline__ Pack16 pack_from_f32(const float* __restrict__ src, int k0, int K, int64_t stride, bool valid) {
  constexpr int HP = Mma<T>::HP;
  float f[HP];
#pragma unroll
  for (int e = 0; e < HP; ++e) f[e] = (valid && k0 + e < K) ? src[(int64_t)(k0 + e) * stride] : 0.f;
  return Mma<T>::pack(f);
}

// row of accumulator register v for lane half h (32x32 C/D layout)
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// LDS row stride (bytes) for a k-contiguous tile of `kelems` elements of `esize` bytes read with 16-B
// packs: rounded up to whole 32-B k-steps plus one 16-B slot, so (stride/16) is odd and the 16 rows
// of a ds_read_b128 lane group land on distinct 16-B slots of the 256-B bank row.
static inline int lds_row_bytes(int kelems, int esize) {
  const int b = ((kelems * esize + 31) / 32) * 32;
  return b + 16;
}
