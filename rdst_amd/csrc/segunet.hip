// The vector (non-GEMM) kernels of the seg-UNet perceptual loss (loss/seg_unet.py:46-127):
// training-mode BatchNorm (statistics, apply + residual + ReLU, backward), MaxPool2d(3, 2, 1) and the backward of the
// decoder's nearest upsampling, the 7x7 stem, the feature L1 / MSE losses of the 'encoder' / 'decoder' modes and the
// multiclass Dice loss of the 'label' modes (smp.losses.DiceLoss; 'label-hr' = RDST-HRL, BASELINE.json configs[4]).
// Activations are NHWC rows (pixel-major) of T = bf16 / fp32; every thread owns one 16-byte channel group of a pixel, so
// all traffic is full-width and per-channel coefficients sit in registers.  All of these are HBM-bound streaming passes;
// every reduction is a per-workgroup partial + a fixed-order finish (bit-identical run to run, no atomics).
// The convolutions themselves are uconv_mfma.hip.
#include "mfma.h"
#include "stem_conv.h"

namespace {

constexpr int NPART = 256;                         // partial-sum workgroups of a reduction
constexpr int MAXC = 512;                          // widest BatchNorm of the resnet34 UNet
constexpr size_t OFF_PART = 0;                     // scratch: [NPART][2][MAXC] floats
constexpr size_t OFF_COEF = (size_t)NPART * 2 * MAXC * 4;   // [3][MAXC] floats (BatchNorm backward coefficients)
constexpr size_t SCRATCH = OFF_COEF + (size_t)3 * MAXC * 4 + 1024;

template <typename T> struct V { static constexpr int N = 16 / (int)sizeof(T); };

template <typename T> __device__ __forceinline__ void ldv(const T* p, float* f);
template <> __device__ __forceinline__ void ldv<float>(const float* p, float* f) {
  const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void ldv<bf16>(const bf16* p, float* f) {
  const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
  f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}
template <typename T> __device__ __forceinline__ void stv(T* p, const float* f);
template <> __device__ __forceinline__ void stv<float>(float* p, const float* f) {
  u32x4_a4 v;
  v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]); v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]);
  *reinterpret_cast<u32x4_a4*>(p) = v;
}
template <> __device__ __forceinline__ void stv<bf16>(bf16* p, const float* f) {
  u32x4_a4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]); v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  *reinterpret_cast<u32x4_a4*>(p) = v;
}

// ---- per-channel sums over pixels: the two moments of BatchNorm's forward (MODE 0: sum x, sum x^2) or of its backward
// (MODE 1: sum g, sum g xhat with g = dY [mask > 0]) ------------------------------------------------------------------
template <typename T, int MODE>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ X, int64_t ldx, const T* __restrict__ dY, int64_t lddy,
                                                     const T* __restrict__ Mk, int64_t ldm, const float* __restrict__ coef,
                                                     int64_t P, int C, float* __restrict__ partial) {
  constexpr int VN = V<T>::N;
  __shared__ float red[256][2 * VN + 1];
  const int lpp = C / VN, ppb = 256 / lpp;
  const int cg = threadIdx.x % lpp, q = threadIdx.x / lpp;
  const int64_t per = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
  float s0[VN], s1[VN], mean[VN], rstd[VN], sca[VN], shf[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) {
    s0[e] = s1[e] = 0.f;
    mean[e] = MODE ? coef[2 * C + cg * VN + e] : 0.f;
    rstd[e] = MODE ? coef[3 * C + cg * VN + e] : 0.f;
    sca[e] = MODE ? coef[cg * VN + e] : 0.f;
    shf[e] = MODE ? coef[C + cg * VN + e] : 0.f;
  }
  if (q < ppb)
    for (int64_t p = p0 + q; p < p1; p += ppb) {
      float x[VN];
      ldv<T>(X + p * ldx + cg * VN, x);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < VN; ++e) { s0[e] += x[e]; s1[e] = fmaf(x[e], x[e], s1[e]); }
      } else {
        float g[VN], m[VN];
        ldv<T>(dY + p * lddy + cg * VN, g);
        if (Mk == X) {   // the mask is relu'(BatchNorm(x)) itself: recomputed with bn_apply_kernel's fma, nothing read
#pragma unroll
          for (int e = 0; e < VN; ++e) g[e] = fmaf(sca[e], x[e], shf[e]) > 0.f ? g[e] : 0.f;
        } else if (Mk) {
          ldv<T>(Mk + p * ldm + cg * VN, m);
#pragma unroll
          for (int e = 0; e < VN; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) { s0[e] += g[e]; s1[e] = fmaf(g[e], (x[e] - mean[e]) * rstd[e], s1[e]); }
      }
    }
#pragma unroll
  for (int e = 0; e < VN; ++e) { red[threadIdx.x][e] = s0[e]; red[threadIdx.x][VN + e] = s1[e]; }
  __syncthreads();
  if (threadIdx.x < lpp) {   // fixed order over the pixel phases of this channel group
#pragma unroll
    for (int e = 0; e < 2 * VN; ++e) {
      float t = 0.f;
      for (int k = 0; k < ppb; ++k) t += red[k * lpp + threadIdx.x][e];
      const int which = e / VN, c = threadIdx.x * VN + (e % VN);
      partial[((int64_t)blockIdx.x * 2 + which) * C + c] = t;
    }
  }
}

// One WAVE per channel (4 channels per workgroup): lane l adds partials l, l + 64, ... in double, then a butterfly over the
// lanes — a fixed order, so the statistics are bit-identical run to run.  (One THREAD per channel walking all the partials
// serially cost 63 us per launch: 8.7 ms per training step over the 138 BatchNorm passes of the UNet.)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void channel_sums(const float* __restrict__ partial, int nblk, int C, int c, double& s, double& q) {
  const int lane = threadIdx.x & 63;
  s = 0.0; q = 0.0;
  for (int b = lane; b < nblk; b += 64) {
    s += (double)partial[((int64_t)b * 2 + 0) * C + c];
    q += (double)partial[((int64_t)b * 2 + 1) * C + c];
  }
  s = wave_sum_f64(s);
  q = wave_sum_f64(q);
}

__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ partial, int nblk, int64_t P, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                                          float* __restrict__ coef) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  channel_sums(partial, nblk, C, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  const double mean = s / (double)P;
  double var = q / (double)P - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = gamma[c] * rstd;
  coef[c] = a;
  coef[C + c] = beta[c] - (float)mean * a;
  coef[2 * C + c] = (float)mean;
  coef[3 * C + c] = rstd;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(P > 1 ? var * (double)P / (double)(P - 1) : var);
}

// [nblk][W] partial rows -> [nslice][W]: slice s adds its rows in double, 256 / W row phases per workgroup combined in fixed
// order.  Between a producer that leaves one partial row per workgroup (the convolution's epilogue: up to 16 thousand rows
// of 2 C floats for the 256 x 256 decoder layers) and the one-wave-per-channel finish, which walks rows C floats apart.
__global__ void __launch_bounds__(256) fold_partials_kernel(const float* __restrict__ in, int nblk, int W, float* __restrict__ out) {
  __shared__ double red[256];
  const int per = (nblk + gridDim.x - 1) / gridDim.x, b0 = blockIdx.x * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  const int nph = W < 256 ? 256 / W : 1;
  for (int col0 = 0; col0 < W; col0 += 256) {
    const int col = col0 + (int)(threadIdx.x % (W < 256 ? W : 256)), ph = threadIdx.x / (W < 256 ? W : 256);
    double a = 0.0;
    if (col < W && ph < nph)
      for (int b = b0 + ph; b < b1; b += nph) a += (double)in[(int64_t)b * W + col];
    red[threadIdx.x] = a;
    __syncthreads();
    if (ph == 0 && col < W) {
      double t = 0.0;
      for (int k = 0; k < nph; ++k) t += red[k * (W < 256 ? W : 256) + (threadIdx.x % (W < 256 ? W : 256))];
      out[(int64_t)blockIdx.x * W + col] = (float)t;
    }
    __syncthreads();
  }
}

// (c0, k1, k2) with dX = c0 g + k1 + k2 x  ==  a (g - mean(g) - xhat mean(g xhat))
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int64_t P, int C,
                                                              const float* __restrict__ coef, float* __restrict__ c3) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  channel_sums(partial, nblk, C, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  const float a = coef[c], mean = coef[2 * C + c], rstd = coef[3 * C + c];
  const float m1 = (float)(s / (double)P), m2 = (float)(q / (double)P);
  const float k2 = -a * m2 * rstd;
  c3[c] = a;
  c3[C + c] = -a * m1 - k2 * mean;
  c3[2 * C + c] = k2;
}

template <typename T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ X, int64_t ldx, const float* __restrict__ coef,
                                                       const T* __restrict__ X2, int64_t ldx2, const float* __restrict__ coef2,
                                                       const T* __restrict__ R, int64_t ldr, int relu, T* __restrict__ Y, int64_t ldy,
                                                       int64_t P, int C) {
  constexpr int VN = V<T>::N;
  const int lpp = C / VN, ppb = 256 / lpp;
  const int cg = threadIdx.x % lpp;
  float a[VN], b[VN], a2[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) {
    a[e] = coef[cg * VN + e];
    b[e] = coef[C + cg * VN + e] + (X2 ? coef2[C + cg * VN + e] : 0.f);
    a2[e] = X2 ? coef2[cg * VN + e] : 0.f;
  }
  for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp; p < P; p += (int64_t)gridDim.x * ppb) {
    float x[VN], t[VN];
    ldv<T>(X + p * ldx + cg * VN, x);
#pragma unroll
    for (int e = 0; e < VN; ++e) x[e] = fmaf(a[e], x[e], b[e]);
    if (X2) {
      ldv<T>(X2 + p * ldx2 + cg * VN, t);
#pragma unroll
      for (int e = 0; e < VN; ++e) x[e] = fmaf(a2[e], t[e], x[e]);
    }
    if (R) {
      ldv<T>(R + p * ldr + cg * VN, t);
#pragma unroll
      for (int e = 0; e < VN; ++e) x[e] += t[e];
    }
    if (relu)
#pragma unroll
      for (int e = 0; e < VN; ++e) x[e] = x[e] > 0.f ? x[e] : 0.f;
    stv<T>(Y + p * ldy + cg * VN, x);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dY, int64_t lddy, const T* __restrict__ Mk, int64_t ldm,
                                                           const T* __restrict__ X, int64_t ldx, const float* __restrict__ c3,
                                                           T* __restrict__ dX, int64_t lddx, T* __restrict__ G, int64_t ldg,
                                                           const T* __restrict__ Ga, int64_t ldga, int64_t P, int C,
                                                           const float* __restrict__ coef) {
  constexpr int VN = V<T>::N;
  const int lpp = C / VN, ppb = 256 / lpp;
  const int cg = threadIdx.x % lpp;
  float c0[VN], k1[VN], k2[VN], sca[VN], shf[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) {
    c0[e] = c3[cg * VN + e]; k1[e] = c3[C + cg * VN + e]; k2[e] = c3[2 * C + cg * VN + e];
    sca[e] = coef[cg * VN + e]; shf[e] = coef[C + cg * VN + e];
  }
  for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp; p < P; p += (int64_t)gridDim.x * ppb) {
    float g[VN], x[VN], m[VN];
    ldv<T>(dY + p * lddy + cg * VN, g);
    ldv<T>(X + p * ldx + cg * VN, x);
    if (Mk == X) {   // mask = relu'(BatchNorm(x)), recomputed (see colsum_kernel)
#pragma unroll
      for (int e = 0; e < VN; ++e) g[e] = fmaf(sca[e], x[e], shf[e]) > 0.f ? g[e] : 0.f;
    } else if (Mk) {
      ldv<T>(Mk + p * ldm + cg * VN, m);
#pragma unroll
      for (int e = 0; e < VN; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) x[e] = fmaf(c0[e], g[e], fmaf(k2[e], x[e], k1[e]));
    stv<T>(dX + p * lddx + cg * VN, x);
    if (G) {
      if (Ga) {
        ldv<T>(Ga + p * ldga + cg * VN, m);
#pragma unroll
        for (int e = 0; e < VN; ++e) g[e] += m[e];
      }
      stv<T>(G + p * ldg + cg * VN, g);
    }
  }
}

// ---- MaxPool2d(3, 2, 1) ----------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ X, int64_t ldx, T* __restrict__ Y, int64_t ldy,
                                                          uint8_t* __restrict__ idx, int B, int H, int W, int C) {
  constexpr int VN = V<T>::N;
  const int lpp = C / VN, ppb = 256 / lpp, cg = threadIdx.x % lpp;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t P = (int64_t)B * Ho * Wo;
  for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp; p < P; p += (int64_t)gridDim.x * ppb) {
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const int64_t b = p / ((int64_t)Wo * Ho);
    float best[VN];
    int bi[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool any = false;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
        if (iy < 0 || ix < 0 || iy >= H || ix >= W) continue;
        float x[VN];
        ldv<T>(X + ((b * H + iy) * W + ix) * ldx + cg * VN, x);
#pragma unroll
        for (int e = 0; e < VN; ++e)
          if (!any || x[e] > best[e]) { best[e] = x[e]; bi[e] = ky * 3 + kx; }   // the first maximum in scan order wins (ATen)
        any = true;
      }
    stv<T>(Y + p * ldy + cg * VN, best);
#pragma unroll
    for (int e = 0; e < VN; ++e) idx[p * C + cg * VN + e] = (uint8_t)bi[e];
  }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* __restrict__ dY, int64_t lddy, const uint8_t* __restrict__ idx,
                                                          const T* __restrict__ add, int64_t ld_add, T* __restrict__ dX, int64_t lddx,
                                                          int B, int H, int W, int C) {
  constexpr int VN = V<T>::N;
  const int lpp = C / VN, ppb = 256 / lpp, cg = threadIdx.x % lpp;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t P = (int64_t)B * H * W;
  for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp; p < P; p += (int64_t)gridDim.x * ppb) {
    const int ix = (int)(p % W), iy = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    float acc[VN];
    if (add) ldv<T>(add + p * ld_add + cg * VN, acc);
    else
#pragma unroll
      for (int e = 0; e < VN; ++e) acc[e] = 0.f;
    // windows (oy, ox) with 2 oy - 1 <= iy <= 2 oy + 1
    for (int oy = iy >> 1; oy <= (iy + 1) >> 1; ++oy) {
      if (oy >= Ho) continue;
      for (int ox = ix >> 1; ox <= (ix + 1) >> 1; ++ox) {
        if (ox >= Wo) continue;
        const int k = (iy - 2 * oy + 1) * 3 + (ix - 2 * ox + 1);
        const int64_t o = (b * Ho + oy) * Wo + ox;
        float g[VN];
        ldv<T>(dY + o * lddy + cg * VN, g);
        const uint8_t* ip = idx + o * C + cg * VN;
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] += ip[e] == k ? g[e] : 0.f;
      }
    }
    stv<T>(dX + p * lddx + cg * VN, acc);
  }
}

// backward of F.interpolate(scale_factor=2, mode='nearest'): dX (B, H, W, C) = 2x2 sums of dY (B, 2H, 2W, C) (+ add)
template <typename T>
__global__ void __launch_bounds__(256) sumpool2_kernel(const T* __restrict__ dY, int64_t lddy, const T* __restrict__ add, int64_t ld_add,
                                                       T* __restrict__ dX, int64_t lddx, int B, int H, int W, int C) {
  constexpr int VN = V<T>::N;
  const int lpp = C / VN, ppb = 256 / lpp, cg = threadIdx.x % lpp;
  const int64_t P = (int64_t)B * H * W;
  for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp; p < P; p += (int64_t)gridDim.x * ppb) {
    const int ix = (int)(p % W), iy = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    const T* r0 = dY + ((b * 2 * H + 2 * iy) * 2 * W + 2 * ix) * lddy + cg * VN;
    const T* r1 = r0 + (int64_t)2 * W * lddy;
    float a[VN], t[VN];
    ldv<T>(r0, a);
    ldv<T>(r0 + lddy, t);
#pragma unroll
    for (int e = 0; e < VN; ++e) a[e] += t[e];
    float c[VN];
    ldv<T>(r1, c);
    ldv<T>(r1 + lddy, t);
#pragma unroll
    for (int e = 0; e < VN; ++e) a[e] += c[e] + t[e];
    if (add) {
      ldv<T>(add + p * ld_add + cg * VN, t);
#pragma unroll
      for (int e = 0; e < VN; ++e) a[e] += t[e];
    }
    stv<T>(dX + p * lddx + cg * VN, a);
  }
}

// ---- the 7x7 / 2 stem: stem_conv.h ---------------------------------------------------------------------------------------
using stemconv::Geo;
constexpr int SC = stemconv::SC;

// ---- feature L1 / MSE loss between the SR and the HR features ('encoder' / 'decoder' modes, loss/seg_unet.py:99-111) ------
template <typename T>
__global__ void __launch_bounds__(256) pair_loss_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ Bv, int64_t ldb,
                                                        int64_t P, int C, int mse, float* __restrict__ partial) {
  constexpr int VN = V<T>::N;
  __shared__ float red[256];
  const int lpp = C / VN, ppb = 256 / lpp, cg = threadIdx.x % lpp;
  const int64_t per = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
  float acc = 0.f;
  if (threadIdx.x / lpp < ppb)
    for (int64_t p = p0 + threadIdx.x / lpp; p < p1; p += ppb) {
      float a[VN], b[VN];
      ldv<T>(A + p * lda + cg * VN, a);
      ldv<T>(Bv + p * ldb + cg * VN, b);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float d = a[e] - b[e];
        acc += mse ? d * d : fabsf(d);
      }
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[i];
    partial[blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(64) scalar_finish_kernel(const float* __restrict__ partial, int nblk, double scale, int accumulate,
                                                           float* __restrict__ loss) {
  if (threadIdx.x != 0) return;
  double t = 0.0;
  for (int b = 0; b < nblk; ++b) t += partial[b];
  loss[0] = (accumulate ? loss[0] : 0.f) + (float)(t * scale);
}
template <typename T>
__global__ void __launch_bounds__(256) pair_loss_bwd_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ Bv, int64_t ldb,
                                                            int64_t P, int C, int mse, float scale, const float* __restrict__ upstream,
                                                            const T* __restrict__ add, int64_t ld_add, T* __restrict__ dA, int64_t ldda) {
  constexpr int VN = V<T>::N;
  const int lpp = C / VN, ppb = 256 / lpp, cg = threadIdx.x % lpp;
  const float s = scale * (upstream ? upstream[0] : 1.f);
  for (int64_t p = (int64_t)blockIdx.x * ppb + threadIdx.x / lpp; p < P; p += (int64_t)gridDim.x * ppb) {
    float a[VN], b[VN], o[VN];
    ldv<T>(A + p * lda + cg * VN, a);
    ldv<T>(Bv + p * ldb + cg * VN, b);
    if (add) ldv<T>(add + p * ld_add + cg * VN, o);
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      const float d = a[e] - b[e];
      const float g = mse ? 2.f * d * s : (d > 0.f ? s : (d < 0.f ? -s : 0.f));
      o[e] = add ? o[e] + g : g;
    }
    stv<T>(dA + p * ldda + cg * VN, o);
  }
}

// ---- multiclass Dice (smp.losses.DiceLoss('multiclass', classes), loss/seg_unet.py:71,112-124) ---------------------------
constexpr int MAXCLS = 8;

template <typename T>
__device__ __forceinline__ int softmax_and_target(const T* __restrict__ lg, const T* __restrict__ tl, const int64_t* __restrict__ labels,
                                                  int64_t p, int64_t ld, int64_t ldt, int ncls, float (&prob)[MAXCLS]) {
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) { prob[c] = c < ncls ? to_f32<T>(lg[p * ld + c]) : -INFINITY; mx = fmaxf(mx, prob[c]); }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) { prob[c] = c < ncls ? __expf(prob[c] - mx) : 0.f; s += prob[c]; }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) prob[c] *= inv;
  if (labels) return (int)labels[p];
  int t = 0;
  float best = to_f32<T>(tl[p * ldt]);
#pragma unroll
  for (int c = 1; c < MAXCLS; ++c)            // torch.argmax: the first maximum
    if (c < ncls) {
      const float v = to_f32<T>(tl[p * ldt + c]);
      if (v > best) { best = v; t = c; }
    }
  return t;
}

template <typename T>
__global__ void __launch_bounds__(256) dice_reduce_kernel(const T* __restrict__ lg, int64_t ld, const T* __restrict__ tl, int64_t ldt,
                                                          const int64_t* __restrict__ labels, int64_t P, int ncls, float* __restrict__ partial) {
  __shared__ float red[256][3 * MAXCLS + 1];
  float I[MAXCLS], S[MAXCLS], Tn[MAXCLS];
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) I[c] = S[c] = Tn[c] = 0.f;
  const int64_t per = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
    float prob[MAXCLS];
    const int t = softmax_and_target<T>(lg, tl, labels, p, ld, ldt, ncls, prob);
#pragma unroll
    for (int c = 0; c < MAXCLS; ++c)
      if (c < ncls) {
        S[c] += prob[c];
        if (c == t) { I[c] += prob[c]; Tn[c] += 1.f; }
      }
  }
#pragma unroll
  for (int c = 0; c < MAXCLS; ++c) { red[threadIdx.x][c] = I[c]; red[threadIdx.x][MAXCLS + c] = S[c]; red[threadIdx.x][2 * MAXCLS + c] = Tn[c]; }
  __syncthreads();
  if (threadIdx.x < 3 * MAXCLS) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[i][threadIdx.x];
    partial[(int64_t)blockIdx.x * 3 * MAXCLS + threadIdx.x] = t;
  }
}
// loss = mean over selected classes of [T_c > 0] (1 - 2 I_c / max(S_c + T_c, eps));  coef[c] = A_c, coef[MAXCLS + c] = B_c with
// d loss / d p_c(pixel) = A_c [t == c] + B_c
__global__ void __launch_bounds__(64) dice_finish_kernel(const float* __restrict__ partial, int nblk, int ncls, int class_mask, float eps,
                                                         float weight, int accumulate, float* __restrict__ loss, float* __restrict__ coef) {
  if (threadIdx.x != 0) return;
  int nsel = 0;
  for (int c = 0; c < ncls; ++c) nsel += (class_mask >> c) & 1;
  double total = 0.0;
  for (int c = 0; c < ncls; ++c) {
    double I = 0.0, S = 0.0, Tn = 0.0;
    for (int b = 0; b < nblk; ++b) {
      I += partial[(int64_t)b * 3 * MAXCLS + c];
      S += partial[(int64_t)b * 3 * MAXCLS + MAXCLS + c];
      Tn += partial[(int64_t)b * 3 * MAXCLS + 2 * MAXCLS + c];
    }
    const double card = S + Tn;
    const bool sel = ((class_mask >> c) & 1) && Tn > 0.0 && nsel > 0;
    const double wc = sel ? (double)weight / nsel : 0.0;
    const double den = card > (double)eps ? card : (double)eps;
    total += wc * (1.0 - 2.0 * I / den);
    coef[c] = (float)(-2.0 * wc / den);
    coef[MAXCLS + c] = (float)(card > (double)eps ? 2.0 * wc * I / (den * den) : 0.0);
  }
  loss[0] = (accumulate ? loss[0] : 0.f) + (float)total;
}
template <typename T>
__global__ void __launch_bounds__(256) dice_bwd_kernel(const T* __restrict__ lg, int64_t ld, const T* __restrict__ tl, int64_t ldt,
                                                       const int64_t* __restrict__ labels, int64_t P, int ncls, const float* __restrict__ coef,
                                                       const float* __restrict__ upstream, T* __restrict__ dl, int64_t ldd, int ncls_pad) {
  const float up = upstream ? upstream[0] : 1.f;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
    float prob[MAXCLS], G[MAXCLS];
    const int t = softmax_and_target<T>(lg, tl, labels, p, ld, ldt, ncls, prob);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCLS; ++c) {
      G[c] = c < ncls ? (c == t ? coef[c] : 0.f) + coef[MAXCLS + c] : 0.f;
      dot = fmaf(prob[c], G[c], dot);
    }
#pragma unroll
    for (int c = 0; c < MAXCLS; ++c)
      if (c < ncls) dl[p * ldd + c] = from_f32<T>(up * prob[c] * (G[c] - dot));
    for (int c = ncls; c < ncls_pad; ++c) dl[p * ldd + c] = from_f32<T>(0.f);
  }
}

// ---- launch helpers --------------------------------------------------------------------------------------------------
int vec_ok(const char* who, int C, int dtype) {
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "%s: bad dtype %d", who, dtype);
  const int vn = dtype == RDST_F32 ? 4 : 8;
  if (C <= 0 || C % vn) return rdst_fail(RDST_ENOTSUP, "%s: C = %d must be a multiple of %d", who, C, vn);
  const int lpp = C / vn;
  if (lpp > 256 || 256 % lpp) return rdst_fail(RDST_ENOTSUP, "%s: C = %d: channel groups per pixel must divide 256", who, C);
  return 0;
}
unsigned grid_for(int64_t P, int C, int dtype) {
  const int vn = dtype == RDST_F32 ? 4 : 8;
  const int ppb = 256 / (C / vn);
  int64_t g = (P + ppb - 1) / ppb;
  if (g > 256 * 32) g = 256 * 32;
  return (unsigned)(g < 1 ? 1 : g);
}
int nparts_for(int64_t P, int C, int dtype) {
  const int vn = dtype == RDST_F32 ? 4 : 8;
  const int ppb = 256 / (C / vn);
  int64_t n = P / ((int64_t)ppb * 4);
  if (n > NPART) n = NPART;
  return (int)(n < 1 ? 1 : n);
}

}  // namespace

extern "C" size_t rdst_u_scratch_bytes(void) { return SCRATCH; }

extern "C" int rdst_u_bn_stats(const void* X, int64_t ld, int64_t P, int C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* coef, void* scratch, int dtype,
                               void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!X || !gamma || !beta || !coef || !scratch || P <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_bn_stats: bad argument");
  if (int rc = vec_ok("rdst_u_bn_stats", C, dtype)) return rc;
  if (C > MAXC) return rdst_fail(RDST_ENOTSUP, "rdst_u_bn_stats: C = %d > %d", C, MAXC);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)((char*)scratch + OFF_PART);
  const int nb = nparts_for(P, C, dtype);
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((colsum_kernel<float, 0>), dim3(nb), dim3(256), 0, st, (const float*)X, ld, nullptr, 0, nullptr, 0, nullptr, P, C, part);
  else
    hipLaunchKernelGGL((colsum_kernel<bf16, 0>), dim3(nb), dim3(256), 0, st, (const bf16*)X, ld, nullptr, 0, nullptr, 0, nullptr, P, C, part);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nb, P, C, gamma, beta, eps, momentum, running_mean,
                     running_var, coef);
  return rdst_launch_status("rdst_u_bn_stats");
}

// the same from per-workgroup partials [nblk][2][C] a producer left (rdst_u_conv's `stats`): only the fixed-order finish runs
extern "C" int rdst_u_bn_stats_from(const float* partials, int nblk, int64_t P, int C, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var, float* coef, void* scratch, void* stream) {
  if (!partials || nblk <= 0 || !gamma || !beta || !coef || !scratch || P <= 0)
    return rdst_fail(RDST_EINVAL, "rdst_u_bn_stats_from: bad argument");
  if (C > MAXC) return rdst_fail(RDST_ENOTSUP, "rdst_u_bn_stats_from: C = %d > %d", C, MAXC);
  hipStream_t st = (hipStream_t)stream;
  if (nblk > 256) {   // many short rows: fold them into 64 first (see fold_partials_kernel)
    float* part = (float*)((char*)scratch + OFF_PART);
    constexpr int NS = 64;
    static_assert(NS <= NPART, "scratch");
    hipLaunchKernelGGL(fold_partials_kernel, dim3(NS), dim3(256), 0, st, partials, nblk, 2 * C, part);
    partials = part;
    nblk = NS;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, partials, nblk, P, C, gamma, beta, eps, momentum,
                     running_mean, running_var, coef);
  return rdst_launch_status("rdst_u_bn_stats_from");
}

extern "C" int rdst_u_bn_apply(const void* X, int64_t ldx, const float* coef, const void* X2, int64_t ldx2, const float* coef2,
                               const void* R, int64_t ldr, int relu, void* Y, int64_t ldy, int64_t P, int C, int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!X || !coef || !Y || P <= 0 || (X2 && !coef2)) return rdst_fail(RDST_EINVAL, "rdst_u_bn_apply: bad argument");
  if (int rc = vec_ok("rdst_u_bn_apply", C, dtype)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = grid_for(P, C, dtype);
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)X, ldx, coef, (const float*)X2, ldx2, coef2,
                       (const float*)R, ldr, relu, (float*)Y, ldy, P, C);
  else
    hipLaunchKernelGGL((bn_apply_kernel<bf16>), dim3(g), dim3(256), 0, st, (const bf16*)X, ldx, coef, (const bf16*)X2, ldx2, coef2,
                       (const bf16*)R, ldr, relu, (bf16*)Y, ldy, P, C);
  return rdst_launch_status("rdst_u_bn_apply");
}

extern "C" int rdst_u_bn_bwd(const void* dY, int64_t lddy, const void* Ymask, int64_t ldm, const void* Xraw, int64_t ldx,
                             const float* coef, void* dX, int64_t lddx, void* Gout, int64_t ldg, const void* Gadd, int64_t ldga,
                             int64_t P, int C, void* scratch, int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!dY || !Xraw || !coef || !dX || !scratch || P <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_bn_bwd: bad argument");
  if (int rc = vec_ok("rdst_u_bn_bwd", C, dtype)) return rc;
  if (C > MAXC) return rdst_fail(RDST_ENOTSUP, "rdst_u_bn_bwd: C = %d > %d", C, MAXC);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)((char*)scratch + OFF_PART);
  float* c3 = (float*)((char*)scratch + OFF_COEF);
  const int nb = nparts_for(P, C, dtype);
  const unsigned g = grid_for(P, C, dtype);
  if (dtype == RDST_F32) {
    hipLaunchKernelGGL((colsum_kernel<float, 1>), dim3(nb), dim3(256), 0, st, (const float*)Xraw, ldx, (const float*)dY, lddy,
                       (const float*)Ymask, ldm, coef, P, C, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nb, P, C, coef, c3);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dY, lddy, (const float*)Ymask, ldm,
                       (const float*)Xraw, ldx, c3, (float*)dX, lddx, (float*)Gout, ldg, (const float*)Gadd, ldga, P, C, coef);
  } else {
    hipLaunchKernelGGL((colsum_kernel<bf16, 1>), dim3(nb), dim3(256), 0, st, (const bf16*)Xraw, ldx, (const bf16*)dY, lddy,
                       (const bf16*)Ymask, ldm, coef, P, C, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nb, P, C, coef, c3);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16>), dim3(g), dim3(256), 0, st, (const bf16*)dY, lddy, (const bf16*)Ymask, ldm,
                       (const bf16*)Xraw, ldx, c3, (bf16*)dX, lddx, (bf16*)Gout, ldg, (const bf16*)Gadd, ldga, P, C, coef);
  }
  return rdst_launch_status("rdst_u_bn_bwd");
}

extern "C" int rdst_u_maxpool_fwd(const void* X, int64_t ldx, void* Y, int64_t ldy, uint8_t* idx, int B, int H, int W, int C, int dtype,
                                  void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!X || !Y || !idx || B <= 0 || H <= 0 || W <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_maxpool_fwd: bad argument");
  if (int rc = vec_ok("rdst_u_maxpool_fwd", C, dtype)) return rc;
  const int64_t P = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  const unsigned g = grid_for(P, C, dtype);
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)X, ldx, (float*)Y, ldy, idx, B, H, W, C);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<bf16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16*)X, ldx, (bf16*)Y, ldy, idx, B, H, W, C);
  return rdst_launch_status("rdst_u_maxpool_fwd");
}

extern "C" int rdst_u_maxpool_bwd(const void* dY, int64_t lddy, const uint8_t* idx, const void* add, int64_t ld_add, void* dX,
                                  int64_t lddx, int B, int H, int W, int C, int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!dY || !dX || !idx || B <= 0 || H <= 0 || W <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_maxpool_bwd: bad argument");
  if (int rc = vec_ok("rdst_u_maxpool_bwd", C, dtype)) return rc;
  const int64_t P = (int64_t)B * H * W;
  const unsigned g = grid_for(P, C, dtype);
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dY, lddy, idx, (const float*)add,
                       ld_add, (float*)dX, lddx, B, H, W, C);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<bf16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16*)dY, lddy, idx, (const bf16*)add,
                       ld_add, (bf16*)dX, lddx, B, H, W, C);
  return rdst_launch_status("rdst_u_maxpool_bwd");
}

extern "C" int rdst_u_sumpool2(const void* dY, int64_t lddy, const void* add, int64_t ld_add, void* dX, int64_t lddx, int B, int H, int W,
                               int C, int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!dY || !dX || B <= 0 || H <= 0 || W <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_sumpool2: bad argument");
  if (int rc = vec_ok("rdst_u_sumpool2", C, dtype)) return rc;
  const int64_t P = (int64_t)B * H * W;
  const unsigned g = grid_for(P, C, dtype);
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((sumpool2_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dY, lddy, (const float*)add, ld_add,
                       (float*)dX, lddx, B, H, W, C);
  else
    hipLaunchKernelGGL((sumpool2_kernel<bf16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16*)dY, lddy, (const bf16*)add, ld_add,
                       (bf16*)dX, lddx, B, H, W, C);
  return rdst_launch_status("rdst_u_sumpool2");
}

extern "C" int rdst_u_stem_fwd(const float* img, const float* W, void* Y, int64_t ld_y, int B, int Cin, int H, int Wd, int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!img || !W || !Y || B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cin > 4 || ld_y < SC) return rdst_fail(RDST_EINVAL, "rdst_u_stem_fwd: bad argument");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_u_stem_fwd: bad dtype");
  Geo g{B, H, Wd, (H + 1) / 2, (Wd + 1) / 2, Cin};
  const unsigned grid = stemconv::fwd_grid(g);
  if (dtype == RDST_F32) hipLaunchKernelGGL((stemconv::fwd_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, W, (float*)Y, ld_y, g);
  else hipLaunchKernelGGL((stemconv::fwd_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, W, (bf16*)Y, ld_y, g);
  return rdst_launch_status("rdst_u_stem_fwd");
}

extern "C" int rdst_u_stem_dgrad(const void* dR, int64_t ld, const float* W, const float* upstream, float* dimg, int B, int Cin, int H,
                                 int Wd, int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!dR || !W || !dimg || B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cin > 4 || ld < SC) return rdst_fail(RDST_EINVAL, "rdst_u_stem_dgrad: bad argument");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_u_stem_dgrad: bad dtype");
  Geo g{B, H, Wd, (H + 1) / 2, (Wd + 1) / 2, Cin};
  const unsigned grid = stemconv::dgrad_grid(g);
  if (dtype == RDST_F32) hipLaunchKernelGGL((stemconv::dgrad_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dR, ld, W, upstream, dimg, g);
  else hipLaunchKernelGGL((stemconv::dgrad_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)dR, ld, W, upstream, dimg, g);
  return rdst_launch_status("rdst_u_stem_dgrad");
}

extern "C" int rdst_u_pair_loss_fwd(const void* A, int64_t lda, const void* Bv, int64_t ldb, int64_t P, int C, int mse, float weight,
                                    int accumulate, float* loss, void* scratch, int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!A || !Bv || !loss || !scratch || P <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_pair_loss_fwd: bad argument");
  if (int rc = vec_ok("rdst_u_pair_loss_fwd", C, dtype)) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)((char*)scratch + OFF_PART);
  const int nb = nparts_for(P, C, dtype);
  if (dtype == RDST_F32) hipLaunchKernelGGL((pair_loss_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)A, lda, (const float*)Bv, ldb, P, C, mse, part);
  else hipLaunchKernelGGL((pair_loss_kernel<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)A, lda, (const bf16*)Bv, ldb, P, C, mse, part);
  hipLaunchKernelGGL(scalar_finish_kernel, dim3(1), dim3(64), 0, st, part, nb, (double)weight / ((double)P * C), accumulate, loss);
  return rdst_launch_status("rdst_u_pair_loss_fwd");
}

extern "C" int rdst_u_pair_loss_bwd(const void* A, int64_t lda, const void* Bv, int64_t ldb, int64_t P, int C, int mse, float weight,
                                    const float* upstream, const void* add, int64_t ld_add, void* dA, int64_t ldda, int dtype,
                                    void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!A || !Bv || !dA || P <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_pair_loss_bwd: bad argument");
  if (int rc = vec_ok("rdst_u_pair_loss_bwd", C, dtype)) return rc;
  const float scale = (float)((double)weight / ((double)P * C));
  const unsigned g = grid_for(P, C, dtype);
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((pair_loss_bwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)A, lda, (const float*)Bv, ldb, P, C,
                       mse, scale, upstream, (const float*)add, ld_add, (float*)dA, ldda);
  else
    hipLaunchKernelGGL((pair_loss_bwd_kernel<bf16>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16*)A, lda, (const bf16*)Bv, ldb, P, C,
                       mse, scale, upstream, (const bf16*)add, ld_add, (bf16*)dA, ldda);
  return rdst_launch_status("rdst_u_pair_loss_bwd");
}

extern "C" int rdst_u_dice_fwd(const void* logits, int64_t ld, const void* target_logits, int64_t ldt, const int64_t* labels, int64_t P,
                               int ncls, int class_mask, float eps, float weight, int accumulate, float* loss, float* coef, void* scratch,
                               int dtype, void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!logits || (!target_logits && !labels) || !loss || !coef || !scratch || P <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_dice_fwd: bad argument");
  if (ncls <= 0 || ncls > MAXCLS) return rdst_fail(RDST_ENOTSUP, "rdst_u_dice_fwd: ncls = %d (1..%d)", ncls, MAXCLS);
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_u_dice_fwd: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)((char*)scratch + OFF_PART);
  int64_t nb = P / 1024;
  nb = nb > NPART ? NPART : (nb < 1 ? 1 : nb);
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((dice_reduce_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)logits, ld, (const float*)target_logits, ldt, labels, P, ncls, part);
  else
    hipLaunchKernelGGL((dice_reduce_kernel<bf16>), dim3((unsigned)nb), dim3(256), 0, st, (const bf16*)logits, ld, (const bf16*)target_logits, ldt, labels, P, ncls, part);
  hipLaunchKernelGGL(dice_finish_kernel, dim3(1), dim3(64), 0, st, part, (int)nb, ncls, class_mask, eps, weight, accumulate, loss, coef);
  return rdst_launch_status("rdst_u_dice_fwd");
}

extern "C" int rdst_u_dice_bwd(const void* logits, int64_t ld, const void* target_logits, int64_t ldt, const int64_t* labels, int64_t P,
                               int ncls, const float* coef, const float* upstream, void* dlogits, int64_t ldd, int ncls_pad, int dtype,
                               void* stream) {
  if (dtype == RDST_F32X3) dtype = RDST_F32;
  if (!logits || (!target_logits && !labels) || !coef || !dlogits || P <= 0) return rdst_fail(RDST_EINVAL, "rdst_u_dice_bwd: bad argument");
  if (ncls <= 0 || ncls > MAXCLS || ncls_pad < ncls || ldd < ncls_pad) return rdst_fail(RDST_EINVAL, "rdst_u_dice_bwd: bad class counts");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_u_dice_bwd: bad dtype");
  int64_t g = (P + 255) / 256;
  g = g > 8192 ? 8192 : g;
  if (dtype == RDST_F32)
    hipLaunchKernelGGL((dice_bwd_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)logits, ld, (const float*)target_logits, ldt,
                       labels, P, ncls, coef, upstream, (float*)dlogits, ldd, ncls_pad);
  else
    hipLaunchKernelGGL((dice_bwd_kernel<bf16>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, ld, (const bf16*)target_logits, ldt,
                       labels, P, ncls, coef, upstream, (bf16*)dlogits, ldd, ncls_pad);
  return rdst_launch_status("rdst_u_dice_bwd");
}
