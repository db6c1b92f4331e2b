// encoder.conv1 of the seg-UNet (smp resnet34: 7x7, stride 2, pad 3, Cin <= 4 -> 64, no bias; loss/seg_unet.py:46,83) and its
// data gradient (rdst_u_stem_fwd / rdst_u_stem_dgrad in segunet.hip; the whole of the shipped ini's 'encoder-L1 [1]' loss).
// One input channel makes this a 49-tap stencil, not a GEMM.  Both kernels put the 64 OUTPUT CHANNELS ON THE LANES of a wave:
//   * every lane keeps the 49 weights of its channel in registers for the whole kernel,
//   * the image values a tap needs are the same for all 64 lanes: the loads are wave-uniform (scalar loads into SGPRs) and
//     feed v_fma_f32 as scalar operands — the inner loop is nothing but FMAs (8 output pixels of one row per wave step),
//   * feature rows are written / read as 256-byte (fp32) or 128-byte (bf16) contiguous rows: fully coalesced.
// The previous form (lane = pixel, one LDS weight read and one global image read per FMA) ran at 4 TFLOP/s: 785 us forward +
// 617 us backward per call at 32 x 256 x 256.  The data gradient sums over the channels = over the lanes: 8 input pixels are
// accumulated per lane and reduced with a 10-step reduce-scatter butterfly (not 8 x 6 steps).
#pragma once
#include "common.h"
#include <type_traits>

namespace stemconv {

constexpr int SC = 64;     // stem channels
struct Geo { int B, H, W, Ho, Wo, Cin; };

// y[(b, ho, wo)][c] = sum_{ci,ky,kx} x[b][ci][2ho+ky-3][2wo+kx-3] w[c][ci][ky][kx];  grid-stride over (row, 8-pixel segment)
template <typename T>
__global__ void __launch_bounds__(256) fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int64_t ldy,
                                                  Geo g) {
  const int c = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nseg = (g.Wo + 7) / 8;
  const int64_t nitems = (int64_t)g.B * g.Ho * nseg;
  for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < nitems; item += (int64_t)gridDim.x * 4) {
    const int seg = (int)(item % nseg), ho = (int)((item / nseg) % g.Ho), b = (int)(item / ((int64_t)nseg * g.Ho));
    const int wo0 = seg * 8;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int ci = 0; ci < g.Cin; ++ci) {
      float wr[49];
      const float* wp = w + ((int64_t)c * g.Cin + ci) * 49;
#pragma unroll
      for (int t = 0; t < 49; ++t) wr[t] = wp[t];
      const float* xp = x + ((int64_t)b * g.Cin + ci) * g.H * g.W;     // wave-uniform from here on
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int iy = 2 * ho + ky - 3;
        if (iy < 0 || iy >= g.H) continue;
        const float* row = xp + (int64_t)iy * g.W;
        float xs[21];
#pragma unroll
        for (int i = 0; i < 21; ++i) {
          const int ix = 2 * wo0 - 3 + i;
          xs[i] = (ix >= 0 && ix < g.W) ? row[ix] : 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[q] = __builtin_fmaf(xs[2 * q + kx], wr[ky * 7 + kx], acc[q]);
        asm volatile("" ::: "memory");   // one image row's 21 scalars at a time (all 7 rows hoisted = 147 SGPRs: spills)
      }
    }
    const int64_t p0 = ((int64_t)b * g.Ho + ho) * g.Wo + wo0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (wo0 + q < g.Wo) y[(p0 + q) * ldy + c] = from_f32<T>(acc[q]);
  }
}

// dimg[b][ci][y][x] = up * sum_{c,ky,kx: (y+3-ky), (x+3-kx) even} dR[b][(y+3-ky)/2][(x+3-kx)/2][c] w[c][ci][ky][kx]
// wave item = 8 consecutive input pixels of one row; lane = channel; sum over lanes at the end
template <typename T>
__global__ void __launch_bounds__(256) dgrad_kernel(const T* __restrict__ dc, int64_t ld, const float* __restrict__ w,
                                                    const float* __restrict__ upstream, float* __restrict__ dx, Geo g) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float up = upstream ? upstream[0] : 1.f;
  const int nseg = (g.W + 7) / 8;
  const int64_t nitems = (int64_t)g.B * g.Cin * g.H * nseg;
  int cur_ci = -1;
  float wr[49];
  for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < nitems; item += (int64_t)gridDim.x * 4) {
    const int seg = (int)(item % nseg), yy = (int)((item / nseg) % g.H);
    const int ci = (int)((item / ((int64_t)nseg * g.H)) % g.Cin), b = (int)(item / ((int64_t)nseg * g.H * g.Cin));
    if (ci != cur_ci) {
      const float* wp = w + ((int64_t)lane * g.Cin + ci) * 49;
#pragma unroll
      for (int t = 0; t < 49; ++t) wr[t] = wp[t];
      cur_ci = ci;
    }
    const int x0 = seg * 8;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    // feature rows ho = (yy + 3 - ky) / 2 for the ky of yy's parity; per row the 8 pixels touch wo in [(x0 + 3 - 6) / 2, (x0 + 7 + 3) / 2]
    // x0 is a multiple of 8, so every index below (tap, feature column, weight) is a compile-time constant once the row
    // parity is fixed: the two parities are separate code paths (a run-time index into wr[] / f[] would go to scratch)
    auto rows = [&](auto parity) {
      constexpr int KYP = decltype(parity)::value;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        constexpr int dummy = 0;
        (void)dummy;
        const int ky = KYP + 2 * kk;
        if (ky > 6) continue;
        const int ho = (yy + 3 - ky) >> 1;
        if (ho < 0 || ho >= g.Ho) continue;
        const T* frow = dc + ((int64_t)b * g.Ho + ho) * g.Wo * ld + lane;
        const int wmin = (x0 >> 1) - 1;              // feature columns wmin .. wmin + 6 serve the 8 pixels
        float f[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int wo = wmin + i;
          f[i] = (wo >= 0 && wo < g.Wo) ? to_f32<T>(frow[(int64_t)wo * ld]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int kx = ((q + 1) & 1) + 2 * j;      // (x0 + q + 3 - kx) even
            if (kx > 6) continue;
            const int i = ((q + 3 - kx) >> 1) + 1;     // = wo - wmin, 0..6
            acc[q] = __builtin_fmaf(f[i], wr[ky * 7 + kx], acc[q]);
          }
      }
    };
    if ((yy + 3) & 1) rows(std::integral_constant<int, 1>{});
    else rows(std::integral_constant<int, 0>{});
    // reduce-scatter over the 64 lanes (channels): halves hand over 4, 2, 1 values, then three plain butterfly steps
    float r4[4], r2[2], r1;
    {
      const bool hi = lane & 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float send = hi ? acc[i] : acc[4 + i];
        const float keep = hi ? acc[4 + i] : acc[i];
        r4[i] = keep + __shfl_xor(send, 32, 64);
      }
    }
    {
      const bool hi = lane & 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float send = hi ? r4[i] : r4[2 + i];
        const float keep = hi ? r4[2 + i] : r4[i];
        r2[i] = keep + __shfl_xor(send, 16, 64);
      }
    }
    {
      const bool hi = lane & 8;
      const float send = hi ? r2[0] : r2[1];
      const float keep = hi ? r2[1] : r2[0];
      r1 = keep + __shfl_xor(send, 8, 64);
    }
    r1 += __shfl_xor(r1, 4, 64);
    r1 += __shfl_xor(r1, 2, 64);
    r1 += __shfl_xor(r1, 1, 64);
    if ((lane & 7) == 0) {
      const int q = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
      if (x0 + q < g.W) dx[(((int64_t)b * g.Cin + ci) * g.H + yy) * g.W + x0 + q] = r1 * up;
    }
  }
}

inline unsigned fwd_grid(const Geo& g) {
  const int64_t n = (int64_t)g.B * g.Ho * ((g.Wo + 7) / 8);
  const int64_t blocks = (n + 3) / 4;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 16 ? 256 * 16 : blocks));
}
inline unsigned dgrad_grid(const Geo& g) {
  const int64_t n = (int64_t)g.B * g.Cin * g.H * ((g.W + 7) / 8);
  const int64_t blocks = (n + 3) / 4;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 16 ? 256 * 16 : blocks));
}

}  // namespace stemconv
