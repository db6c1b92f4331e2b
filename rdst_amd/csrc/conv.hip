// K4/K5/K6: k x k convolution (k = 1, 3) on token-major (NHWC) rows with the activation on the input
// side, scale + residual epilogue and PixelShuffle folded into the store; forward, dgrad, wgrad.
// Reference sequences replaced: see include/rdst_hip.h (rdst_conv_fwd / _bwd).
// Shape-generic implementation on gemm_valu.h (implicit GEMM through addressing functors); the MFMA
// fast paths for the 150->60, 60->60 and 60->240 convs live in conv_mfma.hip.
#include "common.h"
#include "gemm_valu.h"
#include "conv.h"

namespace {

template <typename T>
struct ConvA {  // fwd: A(p, k = tap*Cin + ci) = in_act(X)[p + tap][ci]
  static constexpr bool kFast = true;
  const T* X; int64_t ldx; ConvGeom g; int act;
  __device__ __forceinline__ float operator()(int64_t p, int64_t k) const {
    const int tap = (int)(k / g.Cin), ci = (int)(k - (int64_t)tap * g.Cin);
    const int ky = tap / g.ks, kx = tap - ky * g.ks;
    int b, y, x;
    g.decode(p, b, y, x);
    const int yy = y + ky - g.pad, xx = x + kx - g.pad;
    if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) return 0.f;
    return apply_act(to_f32<T>(X[(((int64_t)b * g.H + yy) * g.W + xx) * ldx + ci]), act);
  }
};
template <typename T>
struct ConvAT {  // wgrad B operand: B(p, n = tap*Cin + ci)
  static constexpr bool kFast = false;
  ConvA<T> a;
  __device__ __forceinline__ float operator()(int64_t p, int n) const { return a(p, n); }
};
struct ConvB {  // fwd: B(k = tap*Cin + ci, co) = Wc[co][ci][tap]
  static constexpr bool kFast = true;
  const float* Wc; ConvGeom g;
  __device__ __forceinline__ float operator()(int64_t k, int co) const {
    const int tap = (int)(k / g.Cin), ci = (int)(k - (int64_t)tap * g.Cin);
    return Wc[((int64_t)co * g.Cin + ci) * g.ks * g.ks + tap];
  }
};
struct ConvBd {  // dgrad: B(k' = tap*Cout + co, ci) = Wc[co][ci][tap]
  static constexpr bool kFast = false;
  const float* Wc; ConvGeom g;
  __device__ __forceinline__ float operator()(int64_t k, int ci) const {
    const int tap = (int)(k / g.Cout), co = (int)(k - (int64_t)tap * g.Cout);
    return Wc[((int64_t)co * g.Cin + ci) * g.ks * g.ks + tap];
  }
};
template <typename T>
struct ConvDyA {  // dgrad: A(p, k' = tap*Cout + co) = s * dY[p - tap][co]
  static constexpr bool kFast = true;
  const T* dY; int64_t ld; ConvGeom g; float s;
  __device__ __forceinline__ float operator()(int64_t p, int64_t k) const {
    const int tap = (int)(k / g.Cout), co = (int)(k - (int64_t)tap * g.Cout);
    const int ky = tap / g.ks, kx = tap - ky * g.ks;
    int b, y, x;
    g.decode(p, b, y, x);
    const int yy = y - ky + g.pad, xx = x - kx + g.pad;
    if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) return 0.f;
    int64_t row; int c;
    g.out_rc(b, yy, xx, co, row, c);
    return to_f32<T>(dY[row * ld + c]) * s;
  }
};
template <typename T>
struct ConvDyAT {  // wgrad: A(co, p) = s * dY[p][co]
  static constexpr bool kFast = false;
  const T* dY; int64_t ld; ConvGeom g; float s;
  __device__ __forceinline__ float operator()(int64_t co, int64_t p) const {
    int b, y, x;
    g.decode(p, b, y, x);
    int64_t row; int c;
    g.out_rc(b, y, x, (int)co, row, c);
    return to_f32<T>(dY[row * ld + c]) * s;
  }
};
template <typename T>
struct ConvDyCol {
  const T* dY; int64_t ld; ConvGeom g; float s;
  __device__ __forceinline__ float operator()(int64_t p, int co) const {
    int b, y, x;
    g.decode(p, b, y, x);
    int64_t row; int c;
    g.out_rc(b, y, x, co, row, c);
    return to_f32<T>(dY[row * ld + c]) * s;
  }
};
template <typename T>
struct ConvFwdEp {
  const float* bias; const T* R; int64_t ldr; T* Y; int64_t ldy; ConvGeom g; float s;
  __device__ __forceinline__ void operator()(int64_t p, int co, float acc, int) const {
    int b, y, x;
    g.decode(p, b, y, x);
    int64_t row; int c;
    g.out_rc(b, y, x, co, row, c);
    float v = (acc + (bias ? bias[co] : 0.f)) * s;
    if (R) v += to_f32<T>(R[row * ldr + c]);
    Y[row * ldy + c] = from_f32<T>(v);
  }
};
template <typename T>
struct ConvDxEp {
  const T* X; int64_t ldx; T* dX; int64_t lddx; int act; const T* addp; int64_t ldacc;
  __device__ __forceinline__ void operator()(int64_t p, int ci, float acc, int) const {
    float v = acc;
    if (act) v *= act_grad(to_f32<T>(X[p * ldx + ci]), act);
    if (addp) v += to_f32<T>(addp[p * ldacc + ci]);
    dX[p * lddx + ci] = from_f32<T>(v);
  }
};
struct ConvSlabEp {  // wgrad split-K partials, already in nn.Conv2d weight order [co][ci][tap]
  float* slab; ConvGeom g;
  __device__ __forceinline__ void operator()(int64_t co, int n, float acc, int z) const {
    const int tap = n / g.Cin, ci = n - tap * g.Cin;
    const int64_t total = (int64_t)g.Cout * g.Cin * g.ks * g.ks;
    slab[(int64_t)z * total + (co * g.Cin + ci) * g.ks * g.ks + tap] = acc;
  }
};

constexpr int kSmallBlocks = 512;
constexpr int kMaxSplits = 128;

int wgrad_splits(const ConvGeom& g) {
  const int tiles = ((g.Cout + 63) / 64) * ((g.ks * g.ks * g.Cin + 63) / 64);
  int s = 1024 / tiles;
  if (s < 1) s = 1;
  if (s > kMaxSplits) s = kMaxSplits;
  return s;
}

template <typename T>
int fwd_t(const T* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const T* R, int64_t ldr, T* Y,
          int64_t ldy, const ConvGeom& g, float s, void* wpack, bool prepacked, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (int rc = conv_c1_fwd_bf16(X, ldx, in_act, Wc, bias, R, ldr, Y, ldy, g, s, st); rc != RDST_ENOTSUP) return rc;
    if (int rc = conv_in1_fwd_bf16(X, ldx, in_act, Wc, bias, R, ldr, Y, ldy, g, s, st); rc != RDST_ENOTSUP) return rc;
    if (int rc = conv3_fwd_bf16(X, ldx, in_act, Wc, bias, R, ldr, Y, ldy, g, s, wpack, prepacked, st); rc != RDST_ENOTSUP) return rc;
  }
  if constexpr (sizeof(T) == 4) {   // the one-channel convs: fp32 vector kernels (conv_c1x.hip), both fp32 modes
    if (int rc = conv_c1x_fwd_f32(X, ldx, in_act, Wc, bias, R, ldr, Y, ldy, g, s, st); rc != RDST_ENOTSUP) return rc;
    if (int rc = conv_in1x_fwd_f32(X, ldx, in_act, Wc, bias, R, ldr, Y, ldy, g, s, st); rc != RDST_ENOTSUP) return rc;
  }
  if constexpr (sizeof(T) == 4) {   // RDST_F32X3: the register-stationary kernels on prepacked hi / lo fragments (conv3x_mfma.hip)
    if (rdst_split())
      if (int rc = conv3x_fwd_f32(X, ldx, in_act, Wc, bias, R, ldr, Y, ldy, g, s, wpack, prepacked, st); rc != RDST_ENOTSUP) return rc;
  }
  if (int rc = conv_fwd_mfma<T>(X, ldx, in_act, Wc, bias, R, ldr, Y, ldy, g, s, st); rc != RDST_ENOTSUP) return rc;
  ConvA<T> la{X, ldx, g, in_act};
  ConvB lb{Wc, g};
  ConvFwdEp<T> ep{bias, R, ldr, Y, ldy, g, s};
  return gemm_valu_launch(la, lb, ep, g.pixels(), g.Cout, (int64_t)g.ks * g.ks * g.Cin, 1, st, "conv_fwd");
}

template <typename T>
int bwd_t(const T* X, int64_t ldx, int in_act, const float* Wc, const T* dY, int64_t lddy, T* dX, int64_t lddx,
          const T* acc, int64_t ldacc, float* dW, float* dbias, float* wsp, const ConvGeom& g, float s, hipStream_t st) {
  const int64_t wtotal = (int64_t)g.Cout * g.Cin * g.ks * g.ks;
  // workspace carve: [packed dgrad weights][generic split-K slab][small][MFMA scratch: un-shuffled dY, wgrad slab]
  void* wpack = wsp;
  wsp = reinterpret_cast<float*>(reinterpret_cast<char*>(wsp) + conv3x_pack_bytes(g.Cin, g.Cout));   // (room for the hi / lo image: twice the bf16 one)
  float* w3slab = wsp;
  wsp = reinterpret_cast<float*>(reinterpret_cast<char*>(wsp) + (g.ks == 3 ? conv3_wgrad_slab_bytes(g.Cin, g.Cout) : 0));
  float* slab = wsp;
  float* small = slab + (int64_t)kMaxSplits * wtotal;
  char* mscr = reinterpret_cast<char*>(small + (int64_t)kSmallBlocks * g.Cout + 64);
  if constexpr (sizeof(T) == 2) {   // the one-output-channel tail conv: vector kernels, its own slab at the end of the workspace
    float* c1slab = reinterpret_cast<float*>(mscr + conv_mfma_scratch_bytes(ConvGeom{g.B, g.H, g.W, g.Cin, g.Cout, g.ks, g.pad, 2}));
    if (int rc = conv_c1_bwd_bf16(X, ldx, in_act, Wc, dY, lddy, dX, lddx, acc, ldacc, dW, dbias, c1slab, g, s, st); rc != RDST_ENOTSUP)
      return rc;
    if (!dX && (dW || dbias) && g.r == 1) {   // one input channel (the head conv: no data gradient, the input is the image)
      if (int rc = conv_in1_wgrad_bf16(X, ldx, in_act, dY, lddy, dW, dbias, c1slab, g, s, st); rc != RDST_ENOTSUP) return rc;
    }
  }
  if constexpr (sizeof(T) == 4) {   // the one-channel convs on fp32 rows (conv_c1x.hip)
    float* c1slab = reinterpret_cast<float*>(mscr + conv_mfma_scratch_bytes(ConvGeom{g.B, g.H, g.W, g.Cin, g.Cout, g.ks, g.pad, 2}));
    if (int rc = conv_c1x_bwd_f32(X, ldx, in_act, Wc, dY, lddy, dX, lddx, acc, ldacc, dW, dbias, c1slab, g, s, st); rc != RDST_ENOTSUP)
      return rc;
    if (!dX && (dW || dbias) && g.r == 1) {
      if (int rc = conv_in1x_wgrad_f32(X, ldx, in_act, dY, lddy, dW, dbias, c1slab, g, s, st); rc != RDST_ENOTSUP) return rc;
    }
  }
  bool dxdone = false;
  if constexpr (sizeof(T) == 2) {   // register-stationary dgrad reads the (possibly pixel-shuffled) dY as it lies
    if (dX) {
      const int rc = conv3_dgrad_bf16(Wc, dY, lddy, dX, lddx, acc, ldacc, in_act, g, s, wpack, st);
      if (rc == 0) dxdone = true;
      else if (rc != RDST_ENOTSUP) return rc;
    }
    if (dW || dbias) {
      const int rc = conv3_wgrad_bf16(X, ldx, in_act, dY, lddy, dW, dbias, w3slab, g, s, st);
      if (rc == 0) { dW = nullptr; dbias = nullptr; }
      else if (rc != RDST_ENOTSUP) return rc;
    }
    if ((dxdone || !dX) && !dW && !dbias) return 0;
  }
  if constexpr (sizeof(T) == 4) {   // RDST_F32X3: the register-stationary dgrad (conv3x_mfma.hip)
    if (dX && rdst_split()) {
      const int rc = conv3x_dgrad_f32(Wc, dY, lddy, dX, lddx, acc, ldacc, in_act, g, s, wpack, st);
      if (rc == 0) dxdone = true;
      else if (rc != RDST_ENOTSUP) return rc;
    }
    if ((dW || dbias) && rdst_split()) {   // ... and weight gradient (conv3x_wgrad.hip): the shuffled dY as it lies
      const int rc = conv3x_wgrad_f32(X, ldx, in_act, dY, lddy, dW, dbias, w3slab, g, s, st);
      if (rc == 0) { dW = nullptr; dbias = nullptr; }
      else if (rc != RDST_ENOTSUP) return rc;
    }
    if ((dxdone || !dX) && !dW && !dbias) return 0;
  }
  // MFMA fast paths want dY as plain (B*H*W, Cout) rows
  int prc = 0;
  int64_t ldp = lddy;
  const T* dYp = plain_dy<T>(dY, lddy, g, mscr, ldp, st, prc);
  if (prc) return prc;
  float* mslab = reinterpret_cast<float*>(mscr + (g.r > 1 ? (size_t)g.pixels() * g.Cout * 4 : 0));
  bool wdone = false;
  if (dW || dbias) {
    const int rc = conv_wgrad_mfma<T>(X, ldx, in_act, dYp, ldp, dW, dbias, mslab, g, s, st);
    if (rc == 0) wdone = true;
    else if (rc != RDST_ENOTSUP) return rc;
  }
  if (!wdone && dbias) {
    ConvDyCol<T> f{dY, lddy, g, s};
    if (int rc = colsum_launch(f, g.pixels(), g.Cout, small, kSmallBlocks, dbias, st, "conv_dbias")) return rc;
  }
  if (!wdone && dW) {
    ConvDyAT<T> la{dY, lddy, g, s};
    ConvAT<T> lb{ConvA<T>{X, ldx, g, in_act}};
    ConvSlabEp ep{slab, g};
    const int splits = wgrad_splits(g);
    const int z = gemm_valu_splits(g.pixels(), splits);
    int rc = gemm_valu_launch(la, lb, ep, g.Cout, g.ks * g.ks * g.Cin, g.pixels(), splits, st, "conv_wgrad");
    if (!rc) rc = slab_reduce(slab, dW, z, wtotal, st);
    if (rc) return rc;
  }
  if (dX && !dxdone) {
    int rc = conv_dgrad_mfma<T>(X, ldx, in_act, Wc, dYp, ldp, dX, lddx, acc, ldacc, g, s, st);
    if (rc == RDST_ENOTSUP) {
      ConvDyA<T> la{dY, lddy, g, s};
      ConvBd lb{Wc, g};
      ConvDxEp<T> ep{X, ldx, dX, lddx, in_act, acc, ldacc};
      rc = gemm_valu_launch(la, lb, ep, g.pixels(), g.Cin, (int64_t)g.ks * g.ks * g.Cout, 1, st, "conv_dgrad");
    }
    if (rc) return rc;
  }
  return 0;
}

int make_geom(ConvGeom& g, int B, int H, int W, int Cin, int Cout, int ks, int r, const char* who) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return rdst_fail(RDST_EINVAL, "%s: non-positive dimension", who);
  if (ks != 1 && ks != 3) return rdst_fail(RDST_ENOTSUP, "%s: kernel size %d (1 or 3)", who, ks);
  if (r < 1 || Cout % (r * r)) return rdst_fail(RDST_EINVAL, "%s: Cout=%d not divisible by shuffle^2=%d", who, Cout, r * r);
  if ((int64_t)B * H * W * r * r >= (1ll << 31)) return rdst_fail(RDST_EINVAL, "%s: more than 2^31 output pixels", who);
  g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ks = ks; g.pad = ks / 2; g.r = r;
  return 0;
}

template <typename T>
__global__ void __launch_bounds__(256) nchw_to_rows_kernel(const float* __restrict__ src, T* __restrict__ rows, int64_t ld,
                                                           int B, int C, int64_t HW) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B*HW*C, c fastest
  if (i >= (int64_t)B * HW * C) return;
  const int c = (int)(i % C);
  const int64_t bp = i / C, b = bp / HW, p = bp - b * HW;
  rows[bp * ld + c] = from_f32<T>(src[(b * C + c) * HW + p]);
}
template <typename T>
__global__ void __launch_bounds__(256) rows_to_nchw_kernel(const T* __restrict__ rows, int64_t ld, float* __restrict__ dst,
                                                           int B, int C, int64_t HW) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over B*C*HW, p fastest
  if (i >= (int64_t)B * HW * C) return;
  const int64_t p = i % HW, bc = i / HW, b = bc / C;
  const int c = (int)(bc - b * C);
  dst[i] = to_f32<T>(rows[(b * HW + p) * ld + c]);
}

}  // namespace

extern "C" int rdst_conv_fwd_packable(int Cin, int Cout, int ksize, int shuffle_r, int has_residual, int in_act, int dtype) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  if (dtype == RDST_F32 && rdst_split()) return conv3x_fwd_shape(Cin, Cout, ksize, shuffle_r, has_residual != 0, in_act) != 0;
  if (dtype != RDST_BF16 || ksize != 3 || in_act) return 0;
  return (Cin == 150 && Cout == 60 && shuffle_r == 1) || (Cin == 60 && Cout == 60 && shuffle_r == 1) ||
         (Cin == 60 && Cout == 240 && shuffle_r == 2 && !has_residual);
}

extern "C" size_t rdst_conv_fwd_workspace(int Cin, int Cout, int ksize) {
  if (Cin <= 0 || Cout <= 0 || ksize != 3) return 16;
  return conv3_pack_bytes(Cin, Cout);
}

// ... per compute mode: RDST_F32X3 reads hi / lo fragment pairs (RDST_PACK_CONV3_FWD_X3), twice the bf16 image
extern "C" size_t rdst_conv_fwd_workspace2(int Cin, int Cout, int ksize, int dtype) {
  if (Cin <= 0 || Cout <= 0 || ksize != 3) return 16;
  return dtype == RDST_F32X3 ? conv3x_pack_bytes(Cin, Cout) : conv3_pack_bytes(Cin, Cout);
}

extern "C" int rdst_conv_fwd(const void* X, int64_t ld_x, int in_act, const float* Wc, const float* bias, const void* R,
                             int64_t ld_r, void* Y, int64_t ld_y, void* workspace, size_t workspace_bytes, int B, int H, int W,
                             int Cin, int Cout, int ksize, float out_scale, int shuffle_r, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  ConvGeom g;
  if (int rc = make_geom(g, B, H, W, Cin, Cout, ksize, shuffle_r, "rdst_conv_fwd")) return rc;
  if (!X || !Wc || !Y) return rdst_fail(RDST_EINVAL, "rdst_conv_fwd: null pointer");
  const int cy = Cout / (shuffle_r * shuffle_r);
  if (ld_x < Cin || ld_y < cy || (R && ld_r < cy)) return rdst_fail(RDST_EINVAL, "rdst_conv_fwd: leading dimension too small");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_conv_fwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  void* wpack = (workspace && workspace_bytes >= rdst_conv_fwd_workspace(Cin, Cout, ksize)) ? workspace : nullptr;
  if (dtype == RDST_F32) {
    void* wpx = (rdst_split() && workspace && workspace_bytes >= conv3x_pack_bytes(Cin, Cout)) ? workspace : nullptr;
    return fwd_t<float>((const float*)X, ld_x, in_act, Wc, bias, (const float*)R, ld_r, (float*)Y, ld_y, g, out_scale, wpx,
                        workspace_bytes == RDST_PREPACKED, st);
  }
  return fwd_t<bf16>((const bf16*)X, ld_x, in_act, Wc, bias, (const bf16*)R, ld_r, (bf16*)Y, ld_y, g, out_scale, wpack,
                     workspace_bytes == RDST_PREPACKED, st);
}

extern "C" size_t rdst_conv_bwd_workspace(int B, int H, int W, int Cin, int Cout, int ksize) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksize <= 0) return 0;
  ConvGeom g{B, H, W, Cin, Cout, ksize, ksize / 2, 2};  // r = 2 reserves room for an un-shuffled dY
  return conv3x_pack_bytes(Cin, Cout) + (ksize == 3 ? conv3_wgrad_slab_bytes(Cin, Cout) : 0) + sizeof(float) * ((size_t)kMaxSplits * Cout * Cin * ksize * ksize + (size_t)kSmallBlocks * Cout + 64) +
         conv_mfma_scratch_bytes(g) + sizeof(float) * (Cin == 1 ? conv_in1_slab_floats(Cout) : conv_c1_slab_floats(Cin));
}

extern "C" int rdst_conv_bwd(const void* X, int64_t ld_x, int in_act, const float* Wc, const void* dY, int64_t ld_dy,
                             void* dX, int64_t ld_dx, const void* dX_add, int64_t ld_dx_add, float* dW, float* dbias,
                             void* workspace,
                             size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int ksize, float out_scale,
                             int shuffle_r, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  ConvGeom g;
  if (int rc = make_geom(g, B, H, W, Cin, Cout, ksize, shuffle_r, "rdst_conv_bwd")) return rc;
  if (!X || !Wc || !dY || !workspace) return rdst_fail(RDST_EINVAL, "rdst_conv_bwd: null pointer");
  const int cy = Cout / (shuffle_r * shuffle_r);
  if (ld_x < Cin || ld_dy < cy || (dX && ld_dx < Cin)) return rdst_fail(RDST_EINVAL, "rdst_conv_bwd: leading dimension too small");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_conv_bwd: bad dtype %d", dtype);
  if (workspace_bytes < rdst_conv_bwd_workspace(B, H, W, Cin, Cout, ksize)) return rdst_fail(RDST_EINVAL, "rdst_conv_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RDST_F32)
    return bwd_t<float>((const float*)X, ld_x, in_act, Wc, (const float*)dY, ld_dy, (float*)dX, ld_dx, (const float*)dX_add, ld_dx_add, dW, dbias, (float*)workspace, g, out_scale, st);
  return bwd_t<bf16>((const bf16*)X, ld_x, in_act, Wc, (const bf16*)dY, ld_dy, (bf16*)dX, ld_dx, (const bf16*)dX_add, ld_dx_add, dW, dbias, (float*)workspace, g, out_scale, st);
}

extern "C" int rdst_nchw_to_rows(const float* nchw, void* rows, int64_t ld, int B, int C, int H, int W, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  if (!nchw || !rows || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C) return rdst_fail(RDST_EINVAL, "rdst_nchw_to_rows: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RDST_F32) hipLaunchKernelGGL((nchw_to_rows_kernel<float>), grid, dim3(256), 0, st, nchw, (float*)rows, ld, B, C, (int64_t)H * W);
  else if (dtype == RDST_BF16) hipLaunchKernelGGL((nchw_to_rows_kernel<bf16>), grid, dim3(256), 0, st, nchw, (bf16*)rows, ld, B, C, (int64_t)H * W);
  else return rdst_fail(RDST_EINVAL, "rdst_nchw_to_rows: bad dtype %d", dtype);
  return rdst_launch_status("nchw_to_rows");
}

extern "C" int rdst_rows_to_nchw(const void* rows, int64_t ld, float* nchw, int B, int C, int H, int W, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  if (!nchw || !rows || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C) return rdst_fail(RDST_EINVAL, "rdst_rows_to_nchw: bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == RDST_F32) hipLaunchKernelGGL((rows_to_nchw_kernel<float>), grid, dim3(256), 0, st, (const float*)rows, ld, nchw, B, C, (int64_t)H * W);
  else if (dtype == RDST_BF16) hipLaunchKernelGGL((rows_to_nchw_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)rows, ld, nchw, B, C, (int64_t)H * W);
  else return rdst_fail(RDST_EINVAL, "rdst_rows_to_nchw: bad dtype %d", dtype);
  return rdst_launch_status("rows_to_nchw");
}
