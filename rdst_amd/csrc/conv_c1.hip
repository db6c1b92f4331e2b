// K6 specialisation: the 3x3 tail conv to ONE output channel (rdst_variations.py:1303, `default_conv(cf, 1, 3)` at
// the 4x resolution: 60 -> 1 on 256x256 pixels), bf16 rows.  With a single output channel there is no GEMM to
// speak of — 540 multiply-adds per pixel against 120 B of input — so the matrix cores are the wrong tool (the
// MFMA weight-gradient kernel ran 559 us with 31 of 32 accumulator columns idle, the generic functor GEMM 200-400
// us); these are plain vector kernels bound by the 252 MB of x / dX:
//   lane = (pixel slot 0..7, channel group 0..7); a lane keeps the 8 x 9 weights of ITS channel group in registers
//   for the whole kernel, a wave handles 8 consecutive pixels per step, a row's 120 B are 8 lanes x 16 B.
//   fwd   : 9 neighbour loads of 16 B, 72 fma, sum over the pixel's 8 lanes (3 shuffles), one bf16 out
//   dgrad : 9 dY neighbours (2 B, shared by the pixel's lanes), 72 fma, one 16-B row-chunk store
//   wgrad : 72 accumulators per lane over all its pixels, reduced over lanes / waves once at the end, per-workgroup
//           partials summed in fixed order by the shared slab reduction (deterministic)
#include "common.h"
#include "gemm_valu.h"
#include "conv.h"
#include "mfma.h"

namespace {

struct C1Args {
  const bf16* X; int64_t ldx;       // (B*H*W, Cin) rows: conv input (fwd, wgrad)
  const float* W;                   // (1, Cin, 3, 3)
  const float* bias;
  const bf16* dY; int64_t lddy;     // (B*H*W, 1)
  bf16* Y; int64_t ldy;             // fwd out (B*H*W, 1)
  bf16* dX; int64_t lddx; const bf16* Acc; int64_t ldacc;
  float* slab;                      // wgrad: [grid][Cin*9 + 1]
  ConvGeom g; float s; int64_t pix_per_wg;
};

constexpr int C1_THREADS = 256;

__device__ __forceinline__ void c1_unpack8(const u32x4_a4& v, float (&f)[8]) {
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
  f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}

// the lane's channel window: 8 channels starting at c0 = min(8*grp, Cin-8); `lo` = first channel of the window that
// belongs to this group (the last group of a row that is not a multiple of 8 overlaps its neighbour)
struct C1Lane {
  int c0, lo;
  bool on;
};
__device__ __forceinline__ C1Lane c1_lane(int grp, int Cin) {
  C1Lane l;
  l.on = 8 * grp < Cin;
  l.c0 = 8 * grp + 8 <= Cin ? 8 * grp : Cin - 8;
  l.lo = 8 * grp - l.c0;
  if (!l.on) { l.c0 = 0; l.lo = 8; }
  return l;
}
__device__ __forceinline__ void c1_advance(const ConvGeom& g, int& b, int& y, int& x, int step) {
  x += step;
  while (x >= g.W) {
    x -= g.W;
    if (++y == g.H) { y = 0; ++b; }
  }
  if (b >= g.B) { b = g.B - 1; }   // past the end (a lane of the last, ragged step): any valid pixel, masked by `valid`
}
__device__ __forceinline__ void c1_weights(const float* W, const C1Lane& l, float (&w)[8][9]) {
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float v = W[(l.c0 + e) * 9 + t];   // unconditional load, masked afterwards
      w[e][t] = (l.on && e >= l.lo) ? v : 0.f;
    }
}

__global__ void __launch_bounds__(C1_THREADS) conv_c1_fwd_kernel(const C1Args p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ps = lane >> 3, grp = lane & 7;
  const ConvGeom g = p.g;
  const C1Lane l = c1_lane(grp, g.Cin);
  float w[8][9];
  c1_weights(p.W, l, w);
  const float b0 = p.bias ? p.bias[0] : 0.f;
  const int64_t P = g.pixels();
  const int64_t rs = (int64_t)g.W * p.ldx, rsy = (int64_t)g.W * p.lddy;
  (void)rs; (void)rsy;
  const int64_t beg = (int64_t)blockIdx.x * p.pix_per_wg, end = beg + p.pix_per_wg < P ? beg + p.pix_per_wg : P;
  // the lane's pixel advances by 32 per step: one decode (integer divisions) up front, carries afterwards
  int b, y, x;
  g.decode(beg + wave * 8 + ps < P ? beg + wave * 8 + ps : P - 1, b, y, x);
  for (int64_t base = beg + wave * 8; base < end; base += (C1_THREADS / 64) * 8, c1_advance(g, b, y, x, (C1_THREADS / 64) * 8)) {
    const int64_t pix = base + ps;
    const bool valid = pix < end;
    u32x4_a4 v[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      ok[t] = valid && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      const int yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= g.W ? g.W - 1 : xx);
      v[t] = *reinterpret_cast<const u32x4_a4*>(p.X + (((int64_t)b * g.H + yc) * g.W + xc) * p.ldx + l.c0);
    }
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float f[8];
      c1_unpack8(v[t], f);
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a = fmaf(f[e], w[e][t], a);
      acc += ok[t] ? a : 0.f;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (valid && grp == 0) p.Y[pix * p.ldy] = __float2bfloat16(fmaf(acc, 1.0f, b0) * p.s);
  }
}

__global__ void __launch_bounds__(C1_THREADS) conv_c1_dgrad_kernel(const C1Args p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ps = lane >> 3, grp = lane & 7;
  const ConvGeom g = p.g;
  const C1Lane l = c1_lane(grp, g.Cin);
  float w[8][9];
  c1_weights(p.W, l, w);
  const int64_t P = g.pixels();
  const int64_t rs = (int64_t)g.W * p.ldx, rsy = (int64_t)g.W * p.lddy;
  (void)rs; (void)rsy;
  const int64_t beg = (int64_t)blockIdx.x * p.pix_per_wg, end = beg + p.pix_per_wg < P ? beg + p.pix_per_wg : P;
  // the lane's pixel advances by 32 per step: one decode (integer divisions) up front, carries afterwards
  int b, y, x;
  g.decode(beg + wave * 8 + ps < P ? beg + wave * 8 + ps : P - 1, b, y, x);
  for (int64_t base = beg + wave * 8; base < end; base += (C1_THREADS / 64) * 8, c1_advance(g, b, y, x, (C1_THREADS / 64) * 8)) {
    const int64_t pix = base + ps;
    const bool valid = pix < end;
    // dX[q] = sum_tap dY[q - (tap offset)] W[tap]
    float dy[9];
    const bf16* yctr = p.dY + (((int64_t)b * g.H + y) * g.W + x) * p.lddy;
    const bool yo[3] = {y + 1 < g.H, true, y > 0}, xo[3] = {x + 1 < g.W, true, x > 0};   // tap t reads the pixel at MINUS its offset
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const bool ok = valid && yo[t / 3] && xo[t % 3];
      const int64_t off = -((int64_t)(t / 3 - 1) * rsy + (int64_t)(t % 3 - 1) * p.lddy);
      const float v = __bfloat162float(*(ok ? yctr + off : yctr));
      dy[t] = ok ? v * p.s : 0.f;
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(dy[t], w[e][t], a);
      o[e] = a;
    }
    if (!valid || !l.on) continue;
    if (p.Acc) {
      float f[8];
      c1_unpack8(*reinterpret_cast<const u32x4_a4*>(p.Acc + pix * p.ldacc + l.c0), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += f[e];
    }
    bf16* dst = p.dX + pix * p.lddx + l.c0;
    if (l.lo == 0) {
      u32x4_a4 u;
      u.x = pack_bf16x2(o[0], o[1]); u.y = pack_bf16x2(o[2], o[3]); u.z = pack_bf16x2(o[4], o[5]); u.w = pack_bf16x2(o[6], o[7]);
      *reinterpret_cast<u32x4_a4*>(dst) = u;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e >= l.lo) dst[e] = __float2bfloat16(o[e]);   // the overlapping last group writes only its own channels
    }
  }
}

__global__ void __launch_bounds__(C1_THREADS) conv_c1_wgrad_kernel(const C1Args p) {
  __shared__ float part[C1_THREADS / 64][8][73];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ps = lane >> 3, grp = lane & 7;
  const ConvGeom g = p.g;
  const C1Lane l = c1_lane(grp, g.Cin);
  float acc[8][9], accb = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[e][t] = 0.f;
  const int64_t P = g.pixels();
  const int64_t rs = (int64_t)g.W * p.ldx, rsy = (int64_t)g.W * p.lddy;
  (void)rs; (void)rsy;
  const int64_t beg = (int64_t)blockIdx.x * p.pix_per_wg, end = beg + p.pix_per_wg < P ? beg + p.pix_per_wg : P;
  // the lane's pixel advances by 32 per step: one decode (integer divisions) up front, carries afterwards
  int b, y, x;
  g.decode(beg + wave * 8 + ps < P ? beg + wave * 8 + ps : P - 1, b, y, x);
  for (int64_t base = beg + wave * 8; base < end; base += (C1_THREADS / 64) * 8, c1_advance(g, b, y, x, (C1_THREADS / 64) * 8)) {
    const int64_t pix = base + ps;
    const bool valid = pix < end;
    // dW[c][tap] += X[q][c] dY[q - (tap offset)]   (q = input pixel)
    const u32x4_a4 xv = *reinterpret_cast<const u32x4_a4*>(p.X + (((int64_t)b * g.H + y) * g.W + x) * p.ldx + l.c0);
    float dy[9];
    const bf16* yctr = p.dY + (((int64_t)b * g.H + y) * g.W + x) * p.lddy;
    const bool yo[3] = {y + 1 < g.H, true, y > 0}, xo[3] = {x + 1 < g.W, true, x > 0};   // tap t reads the pixel at MINUS its offset
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const bool ok = valid && yo[t / 3] && xo[t % 3];
      const int64_t off = -((int64_t)(t / 3 - 1) * rsy + (int64_t)(t % 3 - 1) * p.lddy);
      const float v = __bfloat162float(*(ok ? yctr + off : yctr));
      dy[t] = ok ? v : 0.f;
    }
    float f[8];
    c1_unpack8(xv, f);
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[e][t] = fmaf(f[e], dy[t], acc[e][t]);
    accb += dy[4];   // centre tap = dY at the pixel itself: d(bias)
  }
  // lanes of one channel group (pixel slots: lane bits 3..5), then the waves, then one slab row per workgroup
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float a = acc[e][t];
      a += __shfl_xor(a, 8, 64);
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      acc[e][t] = a;
    }
  accb += __shfl_xor(accb, 8, 64);
  accb += __shfl_xor(accb, 16, 64);
  accb += __shfl_xor(accb, 32, 64);
  if (ps == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 9; ++t) part[wave][grp][e * 9 + t] = acc[e][t];
    part[wave][grp][72] = accb;
  }
  __syncthreads();
  float* my = p.slab + (int64_t)blockIdx.x * (g.Cin * 9 + 1);
  for (int i = tid; i < 8 * 73; i += C1_THREADS) {
    const int gi = i / 73, k = i - gi * 73;
    float a = 0.f;
#pragma unroll
    for (int wv = 0; wv < C1_THREADS / 64; ++wv) a += part[wv][gi][k];
    const C1Lane lg = c1_lane(gi, g.Cin);
    if (k == 72) {
      if (gi == 0) my[g.Cin * 9] = a * p.s;
    } else {
      const int e = k / 9, t = k - e * 9;
      if (lg.on && e >= lg.lo) my[(lg.c0 + e) * 9 + t] = a * p.s;
    }
  }
}

// 1x1 conv with one input and one output channel (MeanShift on a single-channel image, common.py:151-167): an
// elementwise affine map.  out = (in * w) * s (+ b * s) (+ add); 8 elements per thread when the rows are contiguous.
__global__ void __launch_bounds__(256) conv_pw11_kernel(const bf16* __restrict__ in, int64_t ldi, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const bf16* add, int64_t lda, bf16* out,
                                                        int64_t ldo, int64_t P, float s, int vec) {
  const float w = W[0] * s, b = bias ? bias[0] * s : 0.f;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vec) {
    const int64_t e0 = i * 8;
    if (e0 + 8 <= P) {
      float f[8];
      c1_unpack8(*reinterpret_cast<const u32x4_a4*>(in + e0), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], w, b);
      if (add) {
        float a[8];
        c1_unpack8(*reinterpret_cast<const u32x4_a4*>(add + e0), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += a[e];
      }
      u32x4_a4 u;
      u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]); u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
      *reinterpret_cast<u32x4_a4*>(out + e0) = u;
    } else {
      for (int64_t e = e0; e < P; ++e) out[e] = __float2bfloat16(fmaf(__bfloat162float(in[e]), w, b) + (add ? __bfloat162float(add[e]) : 0.f));
    }
    return;
  }
  if (i < P) out[i * ldo] = __float2bfloat16(fmaf(__bfloat162float(in[i * ldi]), w, b) + (add ? __bfloat162float(add[i * lda]) : 0.f));
}

int pw11_launch(const bf16* in, int64_t ldi, const float* W, const float* bias, const bf16* add, int64_t lda, bf16* out,
                int64_t ldo, int64_t P, float s, hipStream_t st, const char* what) {
  const bool vec = ldi == 1 && ldo == 1 && (!add || lda == 1) && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)add) & 3) == 0;
  const int64_t n = vec ? (P + 7) / 8 : P;
  hipLaunchKernelGGL(conv_pw11_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, ldi, W, bias, add, lda, out, ldo, P, s,
                     vec ? 1 : 0);
  return rdst_launch_status(what);
}

bool c1_ok(const ConvGeom& g, int in_act) {
  return g.Cout == 1 && g.ks == 3 && g.r == 1 && in_act == 0 && g.Cin >= 8 && g.Cin <= 64 && (g.Cin & 1) == 0;
}
int c1_grid(const ConvGeom& g, int64_t& ppw, int cap) {
  const int64_t P = g.pixels();
  int64_t grid = (P + 511) / 512;
  if (grid > cap) grid = cap;
  ppw = (((P + grid - 1) / grid) + 7) / 8 * 8;
  return (int)((P + ppw - 1) / ppw);
}

}  // namespace

int conv_c1_fwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const bf16* R, int64_t ldr, bf16* Y,
                     int64_t ldy, const ConvGeom& g, float s, hipStream_t st) {
  if (g.Cin == 1 && g.Cout == 1 && g.ks == 1 && g.r == 1 && in_act == 0)
    return pw11_launch(X, ldx, Wc, bias, R, ldr, Y, ldy, g.pixels(), s, st, "conv_pw11_fwd");
  if (!c1_ok(g, in_act) || R || ((uintptr_t)X & 3) || (ldx & 1)) return RDST_ENOTSUP;
  C1Args p{};
  p.X = X; p.ldx = ldx; p.W = Wc; p.bias = bias; p.Y = Y; p.ldy = ldy; p.g = g; p.s = s;
  const int grid = c1_grid(g, p.pix_per_wg, 4096);
  hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(grid), dim3(C1_THREADS), 0, st, p);
  return rdst_launch_status("conv_c1_fwd");
}

size_t conv_c1_slab_floats(int Cin) { return (size_t)(1024 + 1) * (Cin * 9 + 1); }   // 1024 partial rows + the reduced row

// dW (1, Cin, 3, 3), dbias (1), dX (optional) of the one-output-channel conv; slab >= conv_c1_slab_floats()
int conv_c1_bwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const bf16* dY, int64_t lddy, bf16* dX,
                     int64_t lddx, const bf16* acc, int64_t ldacc, float* dW, float* dbias, float* slab, const ConvGeom& g,
                     float s, hipStream_t st) {
  if (g.Cin == 1 && g.Cout == 1 && g.ks == 1 && g.r == 1 && in_act == 0 && !dW && !dbias && dX)   // frozen MeanShift: dX = dY w s (+ dX_add)
    return pw11_launch(dY, lddy, Wc, nullptr, acc, ldacc, dX, lddx, g.pixels(), s, st, "conv_pw11_dgrad");
  if (!c1_ok(g, in_act) || ((uintptr_t)X & 3) || (ldx & 1) || ((uintptr_t)dX & 3) || (lddx & 1) || ((uintptr_t)acc & 3) || (ldacc & 1))
    return RDST_ENOTSUP;
  C1Args p{};
  p.X = X; p.ldx = ldx; p.W = Wc; p.dY = dY; p.lddy = lddy; p.dX = dX; p.lddx = lddx; p.Acc = acc; p.ldacc = ldacc;
  p.g = g; p.s = s; p.slab = slab;
  if (dW || dbias) {
    const int grid = c1_grid(g, p.pix_per_wg, 1024);
    hipLaunchKernelGGL(conv_c1_wgrad_kernel, dim3(grid), dim3(C1_THREADS), 0, st, p);
    if (int rc = rdst_launch_status("conv_c1_wgrad")) return rc;
    const int n = g.Cin * 9;
    // one reduction over [grid][n + 1]: dW then dbias are contiguous in the slab row; the destinations are not
    float* red = slab + (size_t)grid * (n + 1);
    if (int rc = slab_reduce(slab, red, grid, n + 1, st)) return rc;
    if (dW) (void)hipMemcpyAsync(dW, red, sizeof(float) * n, hipMemcpyDeviceToDevice, st);
    if (dbias) (void)hipMemcpyAsync(dbias, red + n, sizeof(float), hipMemcpyDeviceToDevice, st);
  }
  if (dX) {
    const int grid = c1_grid(g, p.pix_per_wg, 4096);
    hipLaunchKernelGGL(conv_c1_dgrad_kernel, dim3(grid), dim3(C1_THREADS), 0, st, p);
    if (int rc = rdst_launch_status("conv_c1_dgrad")) return rc;
  }
  return 0;
}
