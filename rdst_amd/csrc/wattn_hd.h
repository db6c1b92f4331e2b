// Shared pieces of the compile-time-specialised window-attention kernels (wattn_mfma_hd.hip forward,
// wattn_bwd_mfma_hd.hip backward): bf16, 8x8 windows, HEADS heads of dim D.
#pragma once
#include "common.h"
#include "wattn.h"
#include "mfma.h"

namespace wahd {

constexpr int TSX = 16;  // LDS row stride (floats) of a staged relative-position-table row (15 used)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
template <int GRAN> struct Chunk;
template <> struct Chunk<16> { typedef u32x4_t type; };
template <> struct Chunk<12> { typedef u32x3_a4 type; };  // C = 90: sections are 15 x 12 B, rows only dword aligned
template <> struct Chunk<8> { typedef u32x2_t type; };

#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS char* lds_cp;

template <typename CH> __device__ __forceinline__ void chunk_to_lds(char* dst, const CH& v) { *reinterpret_cast<CH*>(dst) = v; }
template <> __device__ __forceinline__ void chunk_to_lds<u32x3_a4>(char* dst, const u32x3_a4& v) {
  uint32_t* d = reinterpret_cast<uint32_t*>(dst);   // 12-B chunks are not 16-B aligned in LDS
  d[0] = v.x; d[1] = v.y; d[2] = v.z;
}
template <typename CH> __device__ __forceinline__ CH chunk_from_lds(const char* src) { return *reinterpret_cast<const CH*>(src); }
template <> __device__ __forceinline__ u32x3_a4 chunk_from_lds<u32x3_a4>(const char* src) {
  const uint32_t* d = reinterpret_cast<const uint32_t*>(src);
  u32x3_a4 v;
  v.x = d[0]; v.y = d[1]; v.z = d[2];
  return v;
}

__host__ __device__ constexpr uint32_t qmask_bits(int c0, int lo, int hi) {
  return ((c0 >= lo && c0 < hi) ? 0x0000ffffu : 0u) | ((c0 + 1 >= lo && c0 + 1 < hi) ? 0xffff0000u : 0u);
}

__device__ __forceinline__ Pack16 lds_pack(lds_cp p) {
  const u32x4_t v = *reinterpret_cast<const LDS_AS u32x4_t*>(p);
  Pack16 r;
  r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  return r;
}

// One ds_read_b64 per pair of floats (bias pairs of the staged relative-position tables).  hipcc otherwise fuses neighbouring
// pairs into ds_read2_b64, which moves half the bytes per clock of ds_read_b64 (MI355X_MICROARCH.md, LDS table: 128 against
// 256 B/clk) and banks over 32 instead of 64 dwords; a volatile access is not fused and still returns asynchronously (the
// wait is placed at the first use).
__device__ __forceinline__ f32x2 lds_read_f32x2(const LDS_AS f32x2* p) { return *(const volatile LDS_AS f32x2*)p; }

// pack of 8 bf16 read transposed: rows r0..r0+3 and r1..r1+3 of a [row][col] bf16 image (two ds_read_b64_tr_b16)
__device__ __forceinline__ Pack16 lds_tr_pack(lds_cp p0, lds_cp p1) {
  typedef LDS_AS s16x4_t* lds_tr_p;
  const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(p0));
  const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(p1));
  const u32x2_t u0 = __builtin_bit_cast(u32x2_t, b0), u1 = __builtin_bit_cast(u32x2_t, b1);
  Pack16 r;
  r.w[0] = u0.x; r.w[1] = u0.y; r.w[2] = u1.x; r.w[3] = u1.y;
  return r;
}

// max / sum over the two lane halves (lane ^ 32) without touching LDS
__device__ __forceinline__ float half_swap_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Store the rows [c_lo, c_hi) of a transposed accumulator tile (rows = channels og .. og+31 in the
// registers, token on the lane) into the token's LDS row: 4 channels per 8-B store where the group lies
// inside the range, single bf16 stores at its ragged ends.  `rowp` = the lane's row + 8*h bytes.
// one lane half's 4-channel group [c0, c0+4) clipped to [LO, HI): the widest aligned stores that fit
template <int LO, int HI>
__device__ __forceinline__ void store_group(lds_cp rowp, int c0, float a0, float a1, float a2, float a3) {
  const bool in0 = c0 >= LO && c0 < HI, in1 = c0 + 1 >= LO && c0 + 1 < HI, in2 = c0 + 2 >= LO && c0 + 2 < HI,
             in3 = c0 + 3 >= LO && c0 + 3 < HI;
  if (in0 && in1 && in2 && in3) {
    u32x2_t w;
    w.x = pack_bf16x2(a0, a1);
    w.y = pack_bf16x2(a2, a3);
    *reinterpret_cast<LDS_AS u32x2_t*>(rowp + c0 * 2) = w;
    return;
  }
  if (in0 && in1) *reinterpret_cast<LDS_AS uint32_t*>(rowp + c0 * 2) = pack_bf16x2(a0, a1);
  else if (in0) *reinterpret_cast<LDS_AS uint16_t*>(rowp + c0 * 2) = __builtin_bit_cast(uint16_t, (__bf16)a0);
  else if (in1) *reinterpret_cast<LDS_AS uint16_t*>(rowp + c0 * 2 + 2) = __builtin_bit_cast(uint16_t, (__bf16)a1);
  if (in2 && in3) *reinterpret_cast<LDS_AS uint32_t*>(rowp + c0 * 2 + 4) = pack_bf16x2(a2, a3);
  else if (in2) *reinterpret_cast<LDS_AS uint16_t*>(rowp + c0 * 2 + 4) = __builtin_bit_cast(uint16_t, (__bf16)a2);
  else if (in3) *reinterpret_cast<LDS_AS uint16_t*>(rowp + c0 * 2 + 6) = __builtin_bit_cast(uint16_t, (__bf16)a3);
}

// Store the rows [LO, HI) of a transposed accumulator tile (rows = channels OG .. OG+31 in the
// registers, token on the lane) into the token's LDS row.  `rowp` = the lane's row WITHOUT the lane-half
// offset; lane half h owns channels OG + 8*g4 + 4*h .. +3 of register group g4.
template <int OG, int LO, int HI>
__device__ __forceinline__ void store_tile_rows(lds_cp rowp, const f32x16& t, float mul, int h) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int cb = OG + 8 * g4;
    const bool any0 = cb + 3 >= LO && cb < HI, any1 = cb + 7 >= LO && cb + 4 < HI;
    if (!any0 && !any1) continue;
    const float a0 = t[4 * g4] * mul, a1 = t[4 * g4 + 1] * mul, a2 = t[4 * g4 + 2] * mul, a3 = t[4 * g4 + 3] * mul;
    const bool full0 = cb >= LO && cb + 3 < HI, full1 = cb + 4 >= LO && cb + 7 < HI;
    if (full0 && full1) {  // same instruction for both halves
      u32x2_t w;
      w.x = pack_bf16x2(a0, a1);
      w.y = pack_bf16x2(a2, a3);
      *reinterpret_cast<LDS_AS u32x2_t*>(rowp + cb * 2 + 8 * h) = w;
    } else {
      if (any0 && h == 0) store_group<LO, HI>(rowp, cb, a0, a1, a2, a3);
      if (any1 && h != 0) store_group<LO, HI>(rowp, cb + 4, a0, a1, a2, a3);
    }
  }
}

}  // namespace wahd
