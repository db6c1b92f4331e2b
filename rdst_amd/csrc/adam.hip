// Fused Adam over the flat parameter / gradient / moment buffers of the data-parallel training step
// (SURVEY.md §8f row N1).  Same update rule and operation order as torch.optim.Adam (amsgrad=False,
// maximize=False), which the reference builds in utils/optim.py:30-53 with the ini's hyper-parameters
// (betas 0.9/0.99, eps 1e-8, lr 1e-4, weight_decay 0):
//   g = grad + wd * p;  m = m + (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g
//   p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// One launch for all 4.46 M parameters (stock foreach Adam: ~15 multi-tensor launches plus ~1500
// scalar kernels per step in its capturable form).  Pure streaming: 16 B read + 12 B written per
// element, 16-B vector accesses, grid-stride.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_bc1,
                                                   float beta1, float beta2, float eps, float wd, float sqrt_bc2) {
  const int64_t n4 = n >> 2;
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = wd != 0.f ? fmaf(wd, pa[e], ga[e]) : ga[e];
      ma[e] = fmaf(gr - ma[e], omb1, ma[e]);
      va[e] = fmaf(omb2 * gr, gr, va[e] * beta2);
      const float denom = sqrtf(va[e]) / sqrt_bc2 + eps;
      pa[e] = pa[e] - lr_bc1 * (ma[e] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n % 4 elements)
  const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && t < n) {
    const float gr = wd != 0.f ? fmaf(wd, p[t], g[t]) : g[t];
    const float mt = fmaf(gr - m[t], omb1, m[t]);
    const float vt = fmaf(omb2 * gr, gr, v[t] * beta2);
    m[t] = mt;
    v[t] = vt;
    p[t] = p[t] - lr_bc1 * (mt / (sqrtf(vt) / sqrt_bc2 + eps));
  }
}

}  // namespace

extern "C" int rdst_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1)
    return rdst_fail(RDST_EINVAL, "rdst_adam_step: null buffer, n < 0 or step < 1");
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
    return rdst_fail(RDST_EINVAL, "rdst_adam_step: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float lr_bc1 = (float)((double)lr / bc1), sqrt_bc2 = (float)sqrt(bc2);
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, lr_bc1, beta1, beta2, eps, weight_decay, sqrt_bc2);
  return rdst_launch_status("rdst_adam_step");
}
