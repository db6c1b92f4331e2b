// N2: the 'encoder' mode of the reference's seg-UNet perceptual loss (loss/seg_unet.py:80-107) for
// loss layer 1 = the stem of the smp.Unet resnet34 encoder: conv 7x7 / stride 2 / pad 3 (in_ch -> 64, no bias) ->
// BatchNorm2d in TRAINING mode (the reference never puts the UNet into eval(): batch statistics, running statistics
// updated on the SR and on the HR batch, in that order) -> ReLU, then MSE ('...L1' selects MSELoss, :73-74) or L1
// ('...L2' and everything else, :75-78) between the SR and the HR features, mean over all elements.
// fp32 throughout (the loss sits behind the network's fp32 NCHW output); every reduction has a fixed order.
//   fwd : conv(sr), conv(hr) -> per-channel batch statistics -> loss; keeps the conv outputs for the backward
//   bwd : d(loss)/d(feature) -> ReLU' -> BatchNorm backward (batch reductions) -> transposed conv -> d(sr)
// Gradients w.r.t. the UNet's own parameters are NOT produced: the reference computes them (its requires_grad = False
// lands on the modules, :59-61) but no optimizer ever reads them.
// Plain vector kernels: one input channel makes the conv a 49-tap stencil, far from any roofline that matters.
#include "common.h"

namespace {

constexpr int SC = 64;            // stem channels
constexpr int NPART = 256;        // partial-sum blocks of the reductions

struct Geo { int B, H, W, Ho, Wo, Cin; };

// y[b][ho][wo][c] = sum_{ci,ky,kx} x[b][ci][2ho+ky-3][2wo+kx-3] w[c][ci][ky][kx]
__global__ void __launch_bounds__(256) stem_conv_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, Geo g) {
  extern __shared__ float wl[];   // [Cin*49][64] transposed: lanes of a pixel read consecutive channels
  for (int i = threadIdx.x; i < SC * g.Cin * 49; i += 256) {
    const int c = i / (g.Cin * 49), t = i - c * (g.Cin * 49);
    wl[t * SC + c] = w[i];
  }
  __syncthreads();
  const int c = threadIdx.x & 63;
  const int64_t P = (int64_t)g.B * g.Ho * g.Wo;
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (int64_t)gridDim.x * 4) {
    const int wo = (int)(p % g.Wo), ho = (int)((p / g.Wo) % g.Ho), b = (int)(p / ((int64_t)g.Wo * g.Ho));
    float a = 0.f;
    for (int ci = 0; ci < g.Cin; ++ci) {
      const float* xp = x + ((int64_t)b * g.Cin + ci) * g.H * g.W;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int yy = 2 * ho + ky - 3;
        if (yy < 0 || yy >= g.H) continue;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const int xx = 2 * wo + kx - 3;
          if (xx < 0 || xx >= g.W) continue;
          a = fmaf(xp[(int64_t)yy * g.W + xx], wl[((ci * 7 + ky) * 7 + kx) * SC + c], a);
        }
      }
    }
    y[p * SC + c] = a;
  }
}

// partial[blk][which][c] : sums over the block's pixel range; which = 0: sum a, 1: sum b  (fixed order per block)
__global__ void __launch_bounds__(256) stem_colsum2(const float* __restrict__ A, const float* __restrict__ Bv, int sq, int64_t P,
                                                    float* __restrict__ partial) {
  __shared__ float s0[4][SC], s1[4][SC];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t per = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
  float a = 0.f, b = 0.f;
  for (int64_t p = p0 + q; p < p1; p += 4) {
    const float va = A[p * SC + c];
    const float vb = Bv ? Bv[p * SC + c] : va;
    a += va;
    b += sq ? vb * vb : vb;
  }
  s0[q][c] = a; s1[q][c] = b;
  __syncthreads();
  if (q == 0) {
    partial[((int64_t)blockIdx.x * 2 + 0) * SC + c] = s0[0][c] + s0[1][c] + s0[2][c] + s0[3][c];
    partial[((int64_t)blockIdx.x * 2 + 1) * SC + c] = s1[0][c] + s1[1][c] + s1[2][c] + s1[3][c];
  }
}

// stats[0][c] = mean, stats[1][c] = rstd (biased variance); running statistics as nn.BatchNorm2d updates them
__global__ void __launch_bounds__(64) stem_stats_finish(const float* __restrict__ partial, int nblk, int64_t P, float eps, float momentum,
                                                        float* __restrict__ stats, float* __restrict__ rmean, float* __restrict__ rvar) {
  const int c = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s += partial[((int64_t)b * 2 + 0) * SC + c];
    q += partial[((int64_t)b * 2 + 1) * SC + c];
  }
  const double mean = s / (double)P;
  double var = q / (double)P - mean * mean;
  var = var > 0.0 ? var : 0.0;
  stats[c] = (float)mean;
  stats[SC + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(P > 1 ? var * (double)P / (double)(P - 1) : var);
}

// features a = relu(gamma xhat_sr + beta), b likewise from hr; per-block partial loss; dy_sr = dL/d(conv_sr's BN input side):
// writes g = dL/da * [a > 0] * 1 (the gradient w.r.t. the BatchNorm OUTPUT) over ysr's storage is NOT possible (xhat is
// needed again), so g goes to `gbuf` (= the hr buffer, dead after this kernel).
__global__ void __launch_bounds__(256) stem_loss_elem(const float* __restrict__ ysr, float* __restrict__ yhr, const float* __restrict__ st_sr,
                                                      const float* __restrict__ st_hr, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int use_mse, int64_t P, float inv_n,
                                                      float* __restrict__ partial) {
  __shared__ float red[256];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const float ms = st_sr[c], rs = st_sr[SC + c], mh = st_hr[c], rh = st_hr[SC + c], gm = gamma[c], bt = beta[c];
  const int64_t per = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
  float acc = 0.f;
  for (int64_t p = p0 + q; p < p1; p += 4) {
    const float a0 = fmaf(gm, (ysr[p * SC + c] - ms) * rs, bt), b0 = fmaf(gm, (yhr[p * SC + c] - mh) * rh, bt);
    const float a = a0 > 0.f ? a0 : 0.f, b = b0 > 0.f ? b0 : 0.f;
    const float d = a - b;
    acc += use_mse ? d * d : fabsf(d);
    const float gl = use_mse ? 2.f * d * inv_n : (d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f));
    yhr[p * SC + c] = a0 > 0.f ? gl : 0.f;     // gradient w.r.t. the BatchNorm output of the SR branch
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 64) {   // fixed order: the four pixel phases of a channel, then the 64 channels by one thread
    red[threadIdx.x] = red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 64; ++i) t += red[i];
    partial[blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(64) stem_loss_finish(const float* __restrict__ partial, int nblk, float inv_n, float* __restrict__ loss) {
  if (threadIdx.x != 0) return;
  double t = 0.0;
  for (int b = 0; b < nblk; ++b) t += partial[b];
  loss[0] = (float)(t * (double)inv_n);
}

// BatchNorm backward partial sums: which 0 = sum g, 1 = sum g * xhat   (g = gradient w.r.t. the BN output)
__global__ void __launch_bounds__(256) stem_bn_bwd_sums(const float* __restrict__ g, const float* __restrict__ ysr, const float* __restrict__ st,
                                                        int64_t P, float* __restrict__ partial) {
  __shared__ float s0[4][SC], s1[4][SC];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const float m = st[c], r = st[SC + c];
  const int64_t per = (P + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
  float a = 0.f, b = 0.f;
  for (int64_t p = p0 + q; p < p1; p += 4) {
    const float gv = g[p * SC + c];
    a += gv;
    b = fmaf(gv, (ysr[p * SC + c] - m) * r, b);
  }
  s0[q][c] = a; s1[q][c] = b;
  __syncthreads();
  if (q == 0) {
    partial[((int64_t)blockIdx.x * 2 + 0) * SC + c] = s0[0][c] + s0[1][c] + s0[2][c] + s0[3][c];
    partial[((int64_t)blockIdx.x * 2 + 1) * SC + c] = s1[0][c] + s1[1][c] + s1[2][c] + s1[3][c];
  }
}
__global__ void __launch_bounds__(64) stem_bn_bwd_finish(const float* __restrict__ partial, int nblk, int64_t P, float* __restrict__ sums) {
  const int c = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s += partial[((int64_t)b * 2 + 0) * SC + c];
    q += partial[((int64_t)b * 2 + 1) * SC + c];
  }
  sums[c] = (float)(s / (double)P);        // mean(g)
  sums[SC + c] = (float)(q / (double)P);   // mean(g xhat)
}
// dconv = gamma rstd (g - mean(g) - xhat mean(g xhat)) * upstream, written over g
__global__ void __launch_bounds__(256) stem_bn_bwd_elem(float* __restrict__ g, const float* __restrict__ ysr, const float* __restrict__ st,
                                                        const float* __restrict__ sums, const float* __restrict__ gamma,
                                                        const float* __restrict__ upstream, int64_t n) {
  const float up = upstream ? upstream[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i & 63);
    const float xh = (ysr[i] - st[c]) * st[SC + c];
    g[i] = gamma[c] * st[SC + c] * (g[i] - sums[c] - xh * sums[SC + c]) * up;
  }
}
// dx[b][ci][y][x] = sum_{c,ky,kx: (y+3-ky) even, (x+3-kx) even} dconv[b][(y+3-ky)/2][(x+3-kx)/2][c] w[c][ci][ky][kx]
__global__ void __launch_bounds__(256) stem_conv_dgrad(const float* __restrict__ dc, const float* __restrict__ w, float* __restrict__ dx, Geo g) {
  extern __shared__ float wl[];   // [Cin*49][64]
  for (int i = threadIdx.x; i < SC * g.Cin * 49; i += 256) {
    const int c = i / (g.Cin * 49), t = i - c * (g.Cin * 49);
    wl[t * SC + c] = w[i];
  }
  __syncthreads();
  const int64_t n = (int64_t)g.B * g.Cin * g.H * g.W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % g.W), y = (int)((i / g.W) % g.H);
    const int ci = (int)((i / ((int64_t)g.W * g.H)) % g.Cin), b = (int)(i / ((int64_t)g.W * g.H * g.Cin));
    float a = 0.f;
    for (int ky = (y + 3) & 1; ky < 7; ky += 2) {
      const int ho = (y + 3 - ky) >> 1;
      if (ho < 0 || ho >= g.Ho) continue;
      for (int kx = (x + 3) & 1; kx < 7; kx += 2) {
        const int wo = (x + 3 - kx) >> 1;
        if (wo < 0 || wo >= g.Wo) continue;
        const float* dp = dc + (((int64_t)b * g.Ho + ho) * g.Wo + wo) * SC;
        const float* wp = wl + ((ci * 7 + ky) * 7 + kx) * SC;
        float t = 0.f;
#pragma unroll 16
        for (int c = 0; c < SC; ++c) t = fmaf(dp[c], wp[c], t);
        a += t;
      }
    }
    dx[i] = a;
  }
}

struct Lay { float* ysr; float* yhr; float* st_sr; float* st_hr; float* sums; float* partial; size_t bytes; };
Lay carve(void* wsp, int B, int Ho, int Wo) {
  Lay l;
  const size_t n = (size_t)B * Ho * Wo * SC;
  float* f = (float*)wsp;
  l.ysr = f; f += n;
  l.yhr = f; f += n;
  l.st_sr = f; f += 2 * SC;
  l.st_hr = f; f += 2 * SC;
  l.sums = f; f += 2 * SC;
  l.partial = f; f += (size_t)NPART * 2 * SC;
  l.bytes = (size_t)((char*)f - (char*)wsp);
  return l;
}

}  // namespace

extern "C" size_t rdst_stem_loss_workspace(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return carve(nullptr, B, (H + 1) / 2, (W + 1) / 2).bytes + 256;
}

extern "C" int rdst_stem_loss_fwd(const float* sr, const float* hr, const float* conv_w, const float* bn_w, const float* bn_b,
                                  float* running_mean, float* running_var, float momentum, float eps, int use_mse, float* loss,
                                  void* workspace, size_t workspace_bytes, int B, int Cin, int H, int W, void* stream) {
  if (!sr || !hr || !conv_w || !bn_w || !bn_b || !loss || !workspace) return rdst_fail(RDST_EINVAL, "rdst_stem_loss_fwd: null pointer");
  if (B <= 0 || Cin <= 0 || Cin > 4 || H <= 0 || W <= 0) return rdst_fail(RDST_EINVAL, "rdst_stem_loss_fwd: bad dimensions");
  if (workspace_bytes < rdst_stem_loss_workspace(B, H, W)) return rdst_fail(RDST_EINVAL, "rdst_stem_loss_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Geo g{B, H, W, (H + 1) / 2, (W + 1) / 2, Cin};   // floor((H + 6 - 7) / 2) + 1
  const Lay l = carve(workspace, B, g.Ho, g.Wo);
  const int64_t P = (int64_t)B * g.Ho * g.Wo;
  const size_t wsm = (size_t)SC * Cin * 49 * 4;
  const int cgrid = (int)((P + 3) / 4 < 2048 ? (P + 3) / 4 : 2048);
  const int nblk = (int)(P < NPART ? P : NPART);
  // SR first, then HR: the order in which the reference's BatchNorm sees (and records) the two batches
  const float* src[2] = {sr, hr};
  float* ys[2] = {l.ysr, l.yhr};
  float* sts[2] = {l.st_sr, l.st_hr};
  for (int k = 0; k < 2; ++k) {
    hipLaunchKernelGGL(stem_conv_fwd, dim3((unsigned)cgrid), dim3(256), wsm, st, src[k], conv_w, ys[k], g);
    hipLaunchKernelGGL(stem_colsum2, dim3((unsigned)nblk), dim3(256), 0, st, ys[k], (const float*)nullptr, 1, P, l.partial);
    hipLaunchKernelGGL(stem_stats_finish, dim3(1), dim3(64), 0, st, l.partial, nblk, P, eps, momentum, sts[k], running_mean, running_var);
  }
  const float inv_n = 1.0f / (float)((double)P * SC);
  hipLaunchKernelGGL(stem_loss_elem, dim3((unsigned)nblk), dim3(256), 0, st, l.ysr, l.yhr, l.st_sr, l.st_hr, bn_w, bn_b, use_mse, P, inv_n,
                     l.partial);
  hipLaunchKernelGGL(stem_loss_finish, dim3(1), dim3(64), 0, st, l.partial, nblk, use_mse ? inv_n : inv_n, loss);
  return rdst_launch_status("stem_loss_fwd");
}

// d(loss)/d(sr) * upstream[0]; `workspace` is the forward's, untouched in between
extern "C" int rdst_stem_loss_bwd(const float* conv_w, const float* bn_w, const float* upstream, float* dsr, void* workspace,
                                  size_t workspace_bytes, int B, int Cin, int H, int W, void* stream) {
  if (!conv_w || !bn_w || !dsr || !workspace) return rdst_fail(RDST_EINVAL, "rdst_stem_loss_bwd: null pointer");
  if (B <= 0 || Cin <= 0 || Cin > 4 || H <= 0 || W <= 0) return rdst_fail(RDST_EINVAL, "rdst_stem_loss_bwd: bad dimensions");
  if (workspace_bytes < rdst_stem_loss_workspace(B, H, W)) return rdst_fail(RDST_EINVAL, "rdst_stem_loss_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Geo g{B, H, W, (H + 1) / 2, (W + 1) / 2, Cin};
  const Lay l = carve(workspace, B, g.Ho, g.Wo);
  const int64_t P = (int64_t)B * g.Ho * g.Wo;
  const int nblk = (int)(P < NPART ? P : NPART);
  hipLaunchKernelGGL(stem_bn_bwd_sums, dim3((unsigned)nblk), dim3(256), 0, st, l.yhr, l.ysr, l.st_sr, P, l.partial);
  hipLaunchKernelGGL(stem_bn_bwd_finish, dim3(1), dim3(64), 0, st, l.partial, nblk, P, l.sums);
  const int64_t n = P * SC;
  hipLaunchKernelGGL(stem_bn_bwd_elem, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st, l.yhr, l.ysr,
                     l.st_sr, l.sums, bn_w, upstream, n);
  const int64_t nin = (int64_t)B * Cin * H * W;
  hipLaunchKernelGGL(stem_conv_dgrad, dim3((unsigned)((nin + 255) / 256 < 8192 ? (nin + 255) / 256 : 8192)), dim3(256),
                     (size_t)SC * Cin * 49 * 4, st, l.yhr, conv_w, dsr, g);
  return rdst_launch_status("stem_loss_bwd");
}
