// Pieces shared by the window-16 attention kernels (wattn16_mfma.hip: bf16, wattn16_f32.hip: exact fp32): the token map of a
// 16x16 window, the lane-shift helpers of the d(table) diagonal sums and their fixed-order reduction.
#pragma once
#include "wattn_hd.h"

namespace w16c {
using namespace wahd;

__device__ __forceinline__ int64_t win_token16(int b, int wr, int wc, int t, const WinGeom& g) {
  int r = wr * 16 + (t >> 4) + g.shift;
  if (r >= g.H) r -= g.H;
  int c = wc * 16 + (t & 15) + g.shift;
  if (c >= g.W) c -= g.W;
  return ((int64_t)b * g.H + r) * g.W + c;
}

// lane i of every 16-lane row reads lane i - N (shr) / i + N (shl) of its row, 0 outside
template <int N> __device__ __forceinline__ float dpp_row_shr(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + N, 0xf, 0xf, true));
}
template <int N> __device__ __forceinline__ float dpp_row_shl(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float other_half(float x, int h) {   // the value of lane ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(h ? r[0] : r[1]);
}

// fixed-order sum of the row sums of one head: entry (dy, dx) <- sum over yi - yj = dy - 15 of part[yi][yj][dx]
__device__ __forceinline__ void w16_dtable_out(const float* part, float* slab_row, int tid) {
  for (int idx = tid; idx < 961; idx += 512) {
    const int ry = idx / 31, rx = idx - ry * 31;
    float sum = 0.f;
    for (int yi = 0; yi < 16; ++yi) {
      const int yj = yi + 15 - ry;
      if (yj >= 0 && yj < 16) sum += part[(yi * 16 + yj) * 32 + rx];
    }
    slab_row[idx] = sum;
  }
}


}  // namespace w16c
