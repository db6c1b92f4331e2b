// Nearest-neighbour x2 upsampling of token-major rows and its backward: the F.interpolate(scale_factor=2, mode='nearest')
// of SwinIR's 'nearest+conv' reconstruction (networks/swin_transformer_sr.py:801-802).  Pure data movement: one thread
// per 16-byte (or element) chunk of an OUTPUT row (forward) / INPUT row (backward, which adds its four children in a
// fixed order: deterministic).  HBM-bound: 5 row-widths per input pixel either way.
#include "common.h"

namespace {

template <typename T, int VEC>
__global__ void __launch_bounds__(256) up2_fwd_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                      int B, int H, int W, int C) {
  const int cpr = (C + VEC - 1) / VEC;   // chunks per row
  const int64_t total = (int64_t)B * 2 * H * 2 * W * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ck = (int)(i % cpr);
    const int64_t pix = i / cpr;
    const int ox = (int)(pix % (2 * W));
    const int64_t t = pix / (2 * W);
    const int oy = (int)(t % (2 * H));
    const int64_t b = t / (2 * H);
    const T* src = x + ((b * H + (oy >> 1)) * W + (ox >> 1)) * ldx + ck * VEC;
    T* dst = y + pix * ldy + ck * VEC;
    if (VEC > 1 && ck * VEC + VEC <= C) {   // VEC elements = 16 bytes
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
      for (int e = 0; e < VEC && ck * VEC + e < C; ++e) dst[e] = src[e];
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) up2_bwd_kernel(const T* __restrict__ dy, int64_t ldy, T* __restrict__ dx, int64_t ldx,
                                                      int B, int H, int W, int C) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int ix = (int)(pix % W);
    const int64_t t = pix / W;
    const int iy = (int)(t % H);
    const int64_t b = t / H;
    const T* r0 = dy + ((b * 2 * H + 2 * iy) * 2 * W + 2 * ix) * ldy + c;
    const T* r1 = r0 + (int64_t)2 * W * ldy;
    const float s = (to_f32<T>(r0[0]) + to_f32<T>(r0[ldy])) + (to_f32<T>(r1[0]) + to_f32<T>(r1[ldy]));
    dx[pix * ldx + c] = from_f32<T>(s);
  }
}

template <typename T>
int launch(bool fwd, const void* src, int64_t lds, void* dst, int64_t ldd, int B, int H, int W, int C, hipStream_t st) {
  const int64_t work = fwd ? (int64_t)B * 4 * H * W * ((C + 7) / 8) : (int64_t)B * H * W * C;
  int64_t grid = (work + 255) / 256;
  if (grid > 256 * 64) grid = 256 * 64;
  if (grid < 1) grid = 1;
  if (fwd) {
    constexpr int V = 16 / (int)sizeof(T);
    const bool vec = (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0 && (lds * sizeof(T)) % 16 == 0 && (ldd * sizeof(T)) % 16 == 0;
    if (vec) hipLaunchKernelGGL((up2_fwd_kernel<T, V>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)src, lds, (T*)dst, ldd, B, H, W, C);
    else hipLaunchKernelGGL((up2_fwd_kernel<T, 1>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)src, lds, (T*)dst, ldd, B, H, W, C);
  } else {
    hipLaunchKernelGGL((up2_bwd_kernel<T>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)src, lds, (T*)dst, ldd, B, H, W, C);
  }
  return rdst_launch_status(fwd ? "rdst_upsample2_fwd" : "rdst_upsample2_bwd");
}

int check(const char* who, const void* a, const void* b, int64_t lda, int64_t ldb, int B, int H, int W, int C, int dtype) {
  if (!a || !b) return rdst_fail(RDST_EINVAL, "%s: null pointer", who);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return rdst_fail(RDST_EINVAL, "%s: bad shape B=%d H=%d W=%d C=%d", who, B, H, W, C);
  if (lda < C || ldb < C) return rdst_fail(RDST_EINVAL, "%s: leading dimension smaller than C=%d", who, C);
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "%s: bad dtype %d", who, dtype);
  return 0;
}

}  // namespace

extern "C" int rdst_upsample2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, int B, int H, int W, int C, int dtype,
                                  void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  if (int rc = check("rdst_upsample2_fwd", x, y, ld_x, ld_y, B, H, W, C, dtype)) return rc;
  return dtype == RDST_F32 ? launch<float>(true, x, ld_x, y, ld_y, B, H, W, C, (hipStream_t)stream)
                           : launch<bf16>(true, x, ld_x, y, ld_y, B, H, W, C, (hipStream_t)stream);
}

extern "C" int rdst_upsample2_bwd(const void* dy, int64_t ld_dy, void* dx, int64_t ld_dx, int B, int H, int W, int C, int dtype,
                                  void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  if (int rc = check("rdst_upsample2_bwd", dy, dx, ld_dy, ld_dx, B, H, W, C, dtype)) return rc;
  return dtype == RDST_F32 ? launch<float>(false, dy, ld_dy, dx, ld_dx, B, H, W, C, (hipStream_t)stream)
                           : launch<bf16>(false, dy, ld_dy, dx, ld_dx, B, H, W, C, (hipStream_t)stream);
}
