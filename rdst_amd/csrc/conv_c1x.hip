// The one-channel 3x3 convolutions of RDSTSR on fp32 rows (exact fp32 and RDST_F32X3 alike: plain fp32 FMAs, no matrix cores) — the
// tail conv 60 -> 1 on the 256 x 256 output (rdst_variations.py:1314) and its mirror, the head conv 1 -> 60 on the single-channel
// image (:1213).  One output channel is no GEMM (540 multiply-adds against 240 B per pixel); the round-5 fp32 modes ran them on the
// generic paths — the row-stripe MFMA kernel with 31 of 32 output columns empty (637 us forward, 1378 us weight gradient per
// step) and the functor GEMM (244 us data gradient).  conv_c1.hip has the bf16 forms (v_dot2 / packed bf16); here:
//   * wide -> 1 (tail forward): a workgroup owns 8 x 32 output pixels, stages the 10 x 34 halo of input pixels (C floats each: pixel
//     stride 4 C bytes puts 16 consecutive pixels on 16 different 16-byte slots) and the 9 x C weights in LDS; a thread = one pixel:
//     per tap and 4 channels one ds_read_b128 of its pixel and one broadcast ds_read_b128 of the weights, 4 FMAs;
//   * 1 -> wide (tail data gradient with the mirrored kernel; head forward): thread = (4-channel chunk, pixel lane): the chunk's
//     9 x 4 weights in registers for the whole kernel, per pixel 9 LDS reads of the scalar halo, 36 FMAs, + addend / residual, one
//     16-byte store — consecutive threads write consecutive chunks of a pixel row;
//   * weight gradient (both shapes): the same thread mapping with 36 accumulators per thread over its pixels of the staged halo
//     tile, a fixed-order LDS reduction over the pixel lanes, one partial row per workgroup, slab_reduce (deterministic).
#include "conv.h"
#include "gemm_valu.h"
#include "mfma.h"

namespace {

constexpr int TH = 8, TW = 32, HH = TH + 2, HW = TW + 2, NTHR = 256;

struct C1XArgs {
  const float* Xw; int64_t ldw;     // the wide tensor (C channels per pixel)
  const float* X1; int64_t ld1;     // the one-channel tensor
  const float* W; const float* bias;
  const float* Add; int64_t ldadd;  // 1 -> wide: residual / dX_add (output geometry) or null
  float* Y; int64_t ldy;            // output rows
  float* slab;                      // weight gradient: [grid][9 C + C + 1] partial rows
  int B, H, Wd, C;
  int tiles_x, tiles_y; int64_t ntiles;
  float s;
  int mirror;                       // 1 -> wide / weight gradient: 1 = tap index 8 - t (data gradient of the tail; weight gradient of the head)
  int wstride_c, wstride_t;         // weight element (channel c, tap t) at W[c * wstride_c + t * wstride_t]
};

__device__ __forceinline__ void tile_of(const C1XArgs& p, int64_t t, int& b, int& y0, int& x0) {
  const int tx = (int)(t % p.tiles_x);
  const int64_t q = t / p.tiles_x;
  const int ty = (int)(q % p.tiles_y);
  b = (int)(q / p.tiles_y); y0 = ty * TH; x0 = tx * TW;
}

// stage the halo of the wide tensor: [HH x HW pixels][C floats], zeros outside the image
template <int C, int HR = HH>
__device__ __forceinline__ void stage_wide(const C1XArgs& p, float* tile, int b, int y0, int x0, int tid) {
  constexpr int CK = C / 4, NCH = HR * HW * CK, U = 5;
  for (int base = tid; base < NCH; base += NTHR * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + NTHR * u;
      const int px = i / CK, ck = i - px * CK;
      const int hy = px / HW, hx = px - hy * HW;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = i < NCH && y >= 0 && y < p.H && x >= 0 && x < p.Wd;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        const u32x4_a4 q = *reinterpret_cast<const u32x4_a4*>(p.Xw + (((int64_t)b * p.H + y) * p.Wd + x) * p.ldw + 4 * ck);
        v[u] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + NTHR * u;
      if (i < NCH) *reinterpret_cast<float4*>(tile + (size_t)i * 4) = v[u];
    }
  }
}
__device__ __forceinline__ void stage_scalar(const C1XArgs& p, float* tile, int b, int y0, int x0, int tid) {
  for (int i = tid; i < HH * HW; i += NTHR) {
    const int hy = i / HW, hx = i - hy * HW;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    tile[i] = (y >= 0 && y < p.H && x >= 0 && x < p.Wd) ? p.X1[(((int64_t)b * p.H + y) * p.Wd + x) * p.ld1] : 0.f;
  }
}

// ---- wide -> 1: Y[p] = (bias + sum_{t, c} W[c][t] Xw[p + t][c]) s -------------------------------------------------------------------
// EIGHT lanes per output pixel, lane g on channels [8 g, 8 g + 8) with its 9 x 8 weights in registers for the whole kernel: per tap two
// ds_read_b128 of the pixel (the lanes of a pixel read its 4 C contiguous bytes) and 8 FMAs, then a sum over the 8 lanes.  Tiles of
// 4 x 32 pixels: 49 KB of halo, three workgroups per CU (one stages while the others multiply).  The first form — a thread per pixel,
// pixel AND weights from LDS for every 4 FMAs, one 82 KB workgroup of four waves per CU — read 4320 bytes of LDS per pixel where this
// one reads 2304, and nothing hid its staging: 302 us per step at C = 60.
constexpr int TH1 = 4, HH1 = TH1 + 2;
template <int C>
__global__ void __launch_bounds__(NTHR) c1x_fwd_kernel(const C1XArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* tile = sm;                       // [HH1 * HW][C] (+ 8 floats: the last group's second read of the last pixel)
  const int tid = threadIdx.x, g = tid & 7, slot = tid >> 3;
  float w[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) w[t][e] = (8 * g + e < C) ? p.W[(8 * g + e) * p.wstride_c + t * p.wstride_t] : 0.f;
  const float b0 = p.bias ? p.bias[0] : 0.f;
  if (tid < 16) tile[HH1 * HW * C + tid] = 0.f;   // (the pad behind the last pixel: read with zero weights, must be finite)
  for (int64_t t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    const int tx = (int)(t % p.tiles_x);
    const int64_t q = t / p.tiles_x;
    const int ty = (int)(q % p.tiles_y), b = (int)(q / p.tiles_y), y0 = ty * TH1, x0 = tx * TW;
    __syncthreads();
    stage_wide<C, HH1>(p, tile, b, y0, x0, tid);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TH1; ++r) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        // (channels past C: the weights are zero, the bytes read are the next pixel's — finite)
        const float* px = tile + ((r + tp / 3) * HW + slot + tp % 3) * C + 8 * g;
        const float4 xa = *reinterpret_cast<const float4*>(px), xb = *reinterpret_cast<const float4*>(px + 4);
        a0 = fmaf(xa.x, w[tp][0], a0); a1 = fmaf(xa.y, w[tp][1], a1); a0 = fmaf(xa.z, w[tp][2], a0); a1 = fmaf(xa.w, w[tp][3], a1);
        a0 = fmaf(xb.x, w[tp][4], a0); a1 = fmaf(xb.y, w[tp][5], a1); a0 = fmaf(xb.z, w[tp][6], a0); a1 = fmaf(xb.w, w[tp][7], a1);
      }
      float a = a0 + a1;
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 4, 64);
      const int y = y0 + r, x = x0 + slot;
      if (g == 0 && y < p.H && x < p.Wd) p.Y[(((int64_t)b * p.H + y) * p.Wd + x) * p.ldy] = (a + b0) * p.s;
    }
  }
}

// ---- 1 -> wide: Y[p][c] = (bias[c] + sum_t W[c][t'] X1[p + t]) s + Add[p][c], t' = t or 8 - t ---------------------------------------
template <int C>
__global__ void __launch_bounds__(NTHR) c1x_wide_kernel(const C1XArgs p) {
  __shared__ float tile[HH * HW];
  constexpr int CK = C / 4, NL = NTHR / CK;   // pixel lanes
  const int tid = threadIdx.x, ck = tid % CK, pl = tid / CK;
  float w[9][4], bv[4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) w[t][e] = p.W[(4 * ck + e) * p.wstride_c + (p.mirror ? 8 - t : t) * p.wstride_t];
#pragma unroll
  for (int e = 0; e < 4; ++e) bv[e] = p.bias ? p.bias[4 * ck + e] : 0.f;
  for (int64_t t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    int b, y0, x0;
    tile_of(p, t, b, y0, x0);
    __syncthreads();
    stage_scalar(p, tile, b, y0, x0, tid);
    __syncthreads();
    if (pl >= NL) continue;
    for (int px = pl; px < TH * TW; px += NL) {
      const int ly = px >> 5, lx = px & 31;
      const int y = y0 + ly, x = x0 + lx;
      if (y >= p.H || x >= p.Wd) continue;
      float a[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const float v = tile[(ly + tp / 3) * HW + lx + tp % 3];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = fmaf(v, w[tp][e], a[e]);
      }
      const int64_t pix = ((int64_t)b * p.H + y) * p.Wd + x;
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= p.s;
      if (p.Add) {
        const u32x4_a4 q = *reinterpret_cast<const u32x4_a4*>(p.Add + pix * p.ldadd + 4 * ck);
        a[0] += __uint_as_float(q.x); a[1] += __uint_as_float(q.y); a[2] += __uint_as_float(q.z); a[3] += __uint_as_float(q.w);
      }
      u32x4_a4 o;
      o.x = __float_as_uint(a[0]); o.y = __float_as_uint(a[1]); o.z = __float_as_uint(a[2]); o.w = __float_as_uint(a[3]);
      *reinterpret_cast<u32x4_a4*>(p.Y + pix * p.ldy + 4 * ck) = o;
    }
  }
}

// ---- weight gradient: G[t][c] = sum_p X1[p] Xw[p + d(t)][c], d(t) = tap t (or 8 - t: mirror) around the centre; colsum[c] = sum_p Xw[p][c];
// sum1 = sum_p X1[p].  Partial row per workgroup: [9 C | C | 1] ----------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(NTHR) c1x_wgrad_kernel(const C1XArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* tile = sm;                        // [HH * HW][C]
  float* sc = sm + HH * HW * C;            // [TH * TW] scalars of the tile's own pixels (zero outside the image)
  constexpr int CK = C / 4, NL = NTHR / CK, ROW = 9 * C + C + 1;
  const int tid = threadIdx.x, ck = tid % CK, pl = tid / CK;
  float acc[9][4], cs[4] = {0.f, 0.f, 0.f, 0.f}, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
  for (int64_t t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    int b, y0, x0;
    tile_of(p, t, b, y0, x0);
    __syncthreads();
    stage_wide<C>(p, tile, b, y0, x0, tid);
    for (int i = tid; i < TH * TW; i += NTHR) {
      const int y = y0 + (i >> 5), x = x0 + (i & 31);
      sc[i] = (y < p.H && x < p.Wd) ? p.X1[(((int64_t)b * p.H + y) * p.Wd + x) * p.ld1] : 0.f;
    }
    __syncthreads();
    if (pl >= NL) continue;
    for (int px = pl; px < TH * TW; px += NL) {
      const int ly = px >> 5, lx = px & 31;
      const float g = sc[px];
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int tt = p.mirror ? 8 - tp : tp;
        const float4 x4 = *reinterpret_cast<const float4*>(tile + ((ly + tt / 3) * HW + lx + tt % 3) * C + 4 * ck);
        acc[tp][0] = fmaf(g, x4.x, acc[tp][0]); acc[tp][1] = fmaf(g, x4.y, acc[tp][1]);
        acc[tp][2] = fmaf(g, x4.z, acc[tp][2]); acc[tp][3] = fmaf(g, x4.w, acc[tp][3]);
      }
      const float4 c4 = *reinterpret_cast<const float4*>(tile + ((ly + 1) * HW + lx + 1) * C + 4 * ck);
      cs[0] += c4.x; cs[1] += c4.y; cs[2] += c4.z; cs[3] += c4.w;
      if (ck == 0) s1 += g;
    }
  }
  // fixed-order reduction over the pixel lanes through LDS (the tile is dead)
  __syncthreads();
  float* red = sm;   // [NL][CK][41]
  if (pl < NL) {
    float* r = red + (size_t)(pl * CK + ck) * 41;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) r[t * 4 + e] = acc[t][e];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[36 + e] = cs[e];
    r[40] = s1;
  }
  __syncthreads();
  float* out = p.slab + (size_t)blockIdx.x * ROW;
  for (int i = tid; i < ROW; i += NTHR) {
    float a = 0.f;
    if (i < 9 * C) {
      const int t = i / C, c = i - t * C;
      for (int l = 0; l < NL; ++l) a += red[(size_t)(l * CK + c / 4) * 41 + t * 4 + (c & 3)];
    } else if (i < 10 * C) {
      const int c = i - 9 * C;
      for (int l = 0; l < NL; ++l) a += red[(size_t)(l * CK + c / 4) * 41 + 36 + (c & 3)];
    } else {
      for (int l = 0; l < NL; ++l) a += red[(size_t)(l * CK) * 41 + 40];
    }
    out[i] = a;
  }
}

// scatter the reduced row [9 C | C | 1] (x s) into the gradients: dW element (c, t) at dW[c * wsc + t * wst]; dbias = colsum (per channel) or sum1
__global__ void __launch_bounds__(256) c1x_wgrad_finish_kernel(const float* __restrict__ red, int C, float s, int wsc, int wst, float* __restrict__ dW,
                                                               float* __restrict__ dbias, int bias_wide) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 9 * C) {
    const int t = i / C, c = i - t * C;
    if (dW) dW[c * wsc + t * wst] = red[i] * s;
  } else if (dbias) {
    if (bias_wide && i < 10 * C) dbias[i - 9 * C] = red[i] * s;
    else if (!bias_wide && i == 10 * C) dbias[0] = red[i] * s;
  }
}

bool shape_ok(const ConvGeom& g, int in_act, int C) {
  return g.ks == 3 && g.pad == 1 && g.r == 1 && in_act == 0 && (C == 60 || C == 48 || C == 64);
}
void set_tiles(C1XArgs& p, const ConvGeom& g, int C) {
  p.B = g.B; p.H = g.H; p.Wd = g.W; p.C = C;
  p.tiles_x = (g.W + TW - 1) / TW; p.tiles_y = (g.H + TH - 1) / TH;
  p.ntiles = (int64_t)g.B * p.tiles_x * p.tiles_y;
}
int grid_of(const C1XArgs& p, int cap) { return (int)(p.ntiles < cap ? p.ntiles : cap); }

template <int C> size_t fwd_smem() { return sizeof(float) * (HH1 * HW * C + 16); }
template <int C> size_t wg_smem() {
  const size_t a = sizeof(float) * (HH * HW * C + TH * TW), b = sizeof(float) * (size_t)(NTHR / (C / 4)) * (C / 4) * 41;
  return a > b ? a : b;
}

template <int C>
int run_fwd(C1XArgs& p, hipStream_t st) {
  p.tiles_y = (p.H + TH1 - 1) / TH1;   // (this kernel's tiles are 4 rows high)
  p.ntiles = (int64_t)p.B * p.tiles_x * p.tiles_y;
  (void)hipFuncSetAttribute((const void*)c1x_fwd_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_smem<C>());
  hipLaunchKernelGGL(c1x_fwd_kernel<C>, dim3(grid_of(p, 768)), dim3(NTHR), fwd_smem<C>(), st, p);
  return rdst_launch_status("c1x_fwd");
}
template <int C>
int run_wide(C1XArgs& p, hipStream_t st) {
  hipLaunchKernelGGL(c1x_wide_kernel<C>, dim3(grid_of(p, 1024)), dim3(NTHR), 0, st, p);
  return rdst_launch_status("c1x_wide");
}
template <int C>
int run_wgrad(C1XArgs& p, float* dW, float* dbias, int wsc, int wst, int bias_wide, hipStream_t st) {
  const int grid = grid_of(p, 256), ROW = 10 * C + 1;
  (void)hipFuncSetAttribute((const void*)c1x_wgrad_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wg_smem<C>());
  hipLaunchKernelGGL(c1x_wgrad_kernel<C>, dim3(grid), dim3(NTHR), wg_smem<C>(), st, p);
  if (int rc = rdst_launch_status("c1x_wgrad")) return rc;
  float* red = p.slab + (size_t)grid * ROW;
  if (int rc = slab_reduce(p.slab, red, grid, ROW, st)) return rc;
  hipLaunchKernelGGL(c1x_wgrad_finish_kernel, dim3((ROW + 255) / 256), dim3(256), 0, st, red, C, p.s, wsc, wst, dW, dbias, bias_wide);
  return rdst_launch_status("c1x_wgrad_finish");
}
// 1x1 conv with one input and one output channel on fp32 rows (MeanShift on a single-channel image, common.py:151-167): an elementwise
// affine map, out = in (w s) + b s (+ add); four elements per thread where the rows are contiguous.  (On the generic functor GEMM the two
// MeanShifts of an RDST-E1 step and the data gradient of the second took 123 + 123 + 189 us: 0.43 ms of the fp32x3 step.)
__global__ void __launch_bounds__(256) c1x_pw11_kernel(const float* __restrict__ in, int64_t ldi, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* add, int64_t lda, float* out, int64_t ldo,
                                                       int64_t P, float s, int vec) {
  const float w = W[0] * s, b = bias ? bias[0] * s : 0.f;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vec) {
    const int64_t e0 = i * 4;
    if (e0 + 4 <= P) {
      const float4 v = *reinterpret_cast<const float4*>(in + e0);
      float4 o = make_float4(fmaf(v.x, w, b), fmaf(v.y, w, b), fmaf(v.z, w, b), fmaf(v.w, w, b));
      if (add) {
        const float4 a = *reinterpret_cast<const float4*>(add + e0);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      *reinterpret_cast<float4*>(out + e0) = o;
    } else {
      for (int64_t e = e0; e < P; ++e) out[e] = fmaf(in[e], w, b) + (add ? add[e] : 0.f);
    }
    return;
  }
  if (i < P) out[i * ldo] = fmaf(in[i * ldi], w, b) + (add ? add[i * lda] : 0.f);
}
int pw11x_launch(const float* in, int64_t ldi, const float* W, const float* bias, const float* add, int64_t lda, float* out, int64_t ldo,
                 int64_t P, float s, hipStream_t st, const char* what) {
  const bool vec = ldi == 1 && ldo == 1 && (!add || lda == 1) && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)add) & 15) == 0;
  const int64_t n = vec ? (P + 3) / 4 : P;
  hipLaunchKernelGGL(c1x_pw11_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, ldi, W, bias, add, lda, out, ldo, P, s, vec ? 1 : 0);
  return rdst_launch_status(what);
}
}  // namespace

size_t conv_c1x_slab_floats(int C) { return (size_t)(256 + 1) * (10 * C + 1); }

// wide -> 1 forward (the tail conv).  RDST_ENOTSUP for other shapes / a residual.
int conv_c1x_fwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const float* R, int64_t ldr, float* Y,
                     int64_t ldy, const ConvGeom& g, float s, hipStream_t st) {
  if (g.Cin == 1 && g.Cout == 1 && g.ks == 1 && g.r == 1 && in_act == 0)
    return pw11x_launch(X, ldx, Wc, bias, R, ldr, Y, ldy, g.pixels(), s, st, "conv_pw11x_fwd");
  if (g.Cout != 1 || R || !shape_ok(g, in_act, g.Cin) || ((uintptr_t)X & 3)) return RDST_ENOTSUP;
  C1XArgs p{};
  set_tiles(p, g, g.Cin);
  p.Xw = X; p.ldw = ldx; p.W = Wc; p.bias = bias; p.Y = Y; p.ldy = ldy; p.s = s; p.wstride_c = 9; p.wstride_t = 1;
  switch (g.Cin) { case 60: return run_fwd<60>(p, st); case 48: return run_fwd<48>(p, st); case 64: return run_fwd<64>(p, st); }
  return RDST_ENOTSUP;
}
// 1 -> wide forward (the head conv on a single-channel image)
int conv_in1x_fwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const float* R, int64_t ldr, float* Y,
                      int64_t ldy, const ConvGeom& g, float s, hipStream_t st) {
  if (g.Cin != 1 || !shape_ok(g, in_act, g.Cout) || ((uintptr_t)Y & 3) || ((uintptr_t)R & 3)) return RDST_ENOTSUP;
  C1XArgs p{};
  set_tiles(p, g, g.Cout);
  p.X1 = X; p.ld1 = ldx; p.W = Wc; p.bias = bias; p.Add = R; p.ldadd = ldr; p.Y = Y; p.ldy = ldy; p.s = s; p.mirror = 0;
  p.wstride_c = 9; p.wstride_t = 1;   // W (Cout, 1, 3, 3)
  switch (g.Cout) { case 60: return run_wide<60>(p, st); case 48: return run_wide<48>(p, st); case 64: return run_wide<64>(p, st); }
  return RDST_ENOTSUP;
}
// backward of the tail conv (Cout == 1): dX (optional) = s conv^T(dY) + dX_add, dW (1, Cin, 3, 3), dbias (1); slab: conv_c1x_slab_floats(Cin)
int conv_c1x_bwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* dY, int64_t lddy, float* dX, int64_t lddx,
                     const float* acc, int64_t ldacc, float* dW, float* dbias, float* slab, const ConvGeom& g, float s, hipStream_t st) {
  if (g.Cin == 1 && g.Cout == 1 && g.ks == 1 && g.r == 1 && in_act == 0 && !dW && !dbias && dX)   // frozen MeanShift: dX = dY w s (+ dX_add)
    return pw11x_launch(dY, lddy, Wc, nullptr, acc, ldacc, dX, lddx, g.pixels(), s, st, "conv_pw11x_dgrad");
  if (g.Cout != 1 || !shape_ok(g, in_act, g.Cin) || ((uintptr_t)X & 3) || ((uintptr_t)dX & 3) || ((uintptr_t)acc & 3)) return RDST_ENOTSUP;
  C1XArgs p{};
  set_tiles(p, g, g.Cin);
  p.s = s;
  if (dW || dbias) {
    p.Xw = X; p.ldw = ldx; p.X1 = dY; p.ld1 = lddy; p.slab = slab; p.mirror = 0;
    const int C = g.Cin;
    if (C == 60) { if (int rc = run_wgrad<60>(p, dW, dbias, 9, 1, 0, st)) return rc; }
    else if (C == 48) { if (int rc = run_wgrad<48>(p, dW, dbias, 9, 1, 0, st)) return rc; }
    else { if (int rc = run_wgrad<64>(p, dW, dbias, 9, 1, 0, st)) return rc; }
  }
  if (dX) {
    C1XArgs q{};
    set_tiles(q, g, g.Cin);
    q.X1 = dY; q.ld1 = lddy; q.W = Wc; q.bias = nullptr; q.Add = acc; q.ldadd = ldacc; q.Y = dX; q.ldy = lddx; q.s = s; q.mirror = 1;
    q.wstride_c = 9; q.wstride_t = 1;   // W (1, Cin, 3, 3): channel c, tap t at c * 9 + t
    switch (g.Cin) { case 60: return run_wide<60>(q, st); case 48: return run_wide<48>(q, st); case 64: return run_wide<64>(q, st); }
  }
  return 0;
}
// weight gradient of the head conv (Cin == 1, no data gradient: the input is the image): dW (Cout, 1, 3, 3), dbias (Cout)
int conv_in1x_wgrad_f32(const float* X, int64_t ldx, int in_act, const float* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                        const ConvGeom& g, float s, hipStream_t st) {
  if (g.Cin != 1 || !shape_ok(g, in_act, g.Cout) || ((uintptr_t)dY & 3)) return RDST_ENOTSUP;
  C1XArgs p{};
  set_tiles(p, g, g.Cout);
  // dW[c][t] = sum_p dY[p][c] x[p + t - centre] = sum_q x[q] dY[q - (t - centre)][c]: the wide tensor (dY) shifted the other way
  p.Xw = dY; p.ldw = lddy; p.X1 = X; p.ld1 = ldx; p.slab = slab; p.s = s; p.mirror = 1;
  const int C = g.Cout;
  if (C == 60) return run_wgrad<60>(p, dW, dbias, 9, 1, 1, st);
  if (C == 48) return run_wgrad<48>(p, dW, dbias, 9, 1, 1, st);
  return run_wgrad<64>(p, dW, dbias, 9, 1, 1, st);
}
