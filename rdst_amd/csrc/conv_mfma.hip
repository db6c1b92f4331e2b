// K4/K5 on the gfx950 matrix cores: 3x3 (and 1x1) convolution on token-major rows as an implicit GEMM
// without im2col: forward, dgrad (= the same kernel with the weights read transposed and the taps
// mirrored) and wgrad.
//
// forward / dgrad: persistent 8-wave workgroups.  A chunk of output channels x ALL taps of the weight
// tensor is converted to the compute type once per workgroup and stays in LDS ([tap][col][ci],
// ci-contiguous, padded so ds_read_b128 is conflict-free); each wave walks 32-pixel slabs: for every
// tap it loads the neighbour pixel's channel row straight into MFMA fragment shape (zero outside the
// image, the activation of the '3conv' variant applied in registers) and multiplies it against every
// column tile of the chunk; accumulators live across the 9 taps.  Epilogue: bias, scale, residual,
// PixelShuffle addressing (fwd) or activation gradient / accumulate (dgrad) on the 32x32 tile.
// A pixel-shuffled dY is un-shuffled once into scratch so dgrad and wgrad read plain rows.
//
// wgrad: contraction over pixels, one workgroup per (pixel range, kernel row ky); stripes of 32 pixels
// of dY and of the three kx-shifted input rows are staged in LDS and read transposed
// (ds_read_b64_tr_b16) / element-wise (fp32).  d(bias) rides on a ones column of the centre tap.
#include "conv.h"
#include "mfma.h"
#include <stdlib.h>

int slab_reduce(const float* slab, float* out, int S, int64_t n, hipStream_t st);

namespace {

bool mfma_disabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("RDST_DISABLE_MFMA");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

constexpr int CMODE_FWD = 0, CMODE_DGRAD = 1;
constexpr int CV_MAXCT = 3;  // column tiles per chunk (96 output channels)

template <typename T>
struct ConvArgs {
  const T* A; int64_t lda; int CA;   // rows contracted per tap: X (fwd, CA = Cin) / dY (dgrad, CA = Cout)
  const float* Wc;                   // (Cout, Cin, ks, ks)
  const float* bias;
  const T* R; int64_t ldr;
  T* Y; int64_t ldy;                 // fwd: Y (output geometry), dgrad: dX
  const T* Xa; int64_t ldxa;         // dgrad: X for act'
  int in_act; const T* Acc; int64_t ldacc;   // dgrad: + dX_add
  ConvGeom g;
  int Nout;                          // fwd: Cout, dgrad: Cin
  float s;
  int Tn, ldw, nch;
  int eps_off;
};

template <typename T, int TMAX, int MODE>
__global__ void __launch_bounds__(512) conv_mfma_kernel(const ConvArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T>;
  constexpr int KP = MM::KP, HP = MM::HP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int Tn = p.Tn;
  const ConvGeom g = p.g;
  const int ntap = g.ks * g.ks;
  const int64_t P = g.pixels();
  const int64_t nslabs = (P + 31) / 32;
  float* eps = reinterpret_cast<float*>(smem + (p.eps_off < 0 ? 0 : p.eps_off)) + wave * 1024;  // wave-private epilogue tile

  for (int n0 = 0; n0 < p.Nout; n0 += p.nch) {
    __syncthreads();
    const int nc = (p.Nout - n0 < p.nch) ? p.Nout - n0 : p.nch;
    const int ncp = ((nc + 31) / 32) * 32;
    // stage W[tap][col][ci-pack] for this chunk
    for (int idx = tid; idx < ntap * ncp * 2 * Tn; idx += 512) {
      const int ph = idx % (2 * Tn);
      const int rest = idx / (2 * Tn);
      const int n = rest % ncp, tap = rest / ncp;
      const bool ok = n < nc;
      Pack16 w;
      if (MODE == CMODE_FWD) {
        // B[k = ci][col = co] = Wc[co][ci][tap]: stride over ci is ks*ks
        w = pack_from_f32<T>(p.Wc + ((int64_t)(n0 + n) * g.Cin) * ntap + tap, ph * HP, p.CA, ntap, ok);
      } else {
        // dgrad: contraction over co, output col = ci, mirrored tap: Wc[co][ci][ntap-1-tap]
        w = pack_from_f32<T>(p.Wc + (int64_t)(n0 + n) * ntap + (ntap - 1 - tap), ph * HP, p.CA, (int64_t)g.Cin * ntap, ok);
      }
      *reinterpret_cast<Pack16*>(smem + ((size_t)tap * ncp + n) * p.ldw + ph * 16) = w;
    }
    __syncthreads();
    const int nct = ncp / 32;

    for (int64_t slab = (int64_t)blockIdx.x * 8 + wave; slab < nslabs; slab += (int64_t)gridDim.x * 8) {
      const int64_t pix = slab * 32 + r;
      const bool pvalid = pix < P;
      int b, y, x;
      g.decode(pvalid ? pix : 0, b, y, x);
      f32x16 acc[CV_MAXCT];
#pragma unroll
      for (int c = 0; c < CV_MAXCT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
      for (int tap = 0; tap < ntap; ++tap) {
        const int ky = tap / g.ks, kx = tap - ky * g.ks;
        const int yy = y + ky - g.pad, xx = x + kx - g.pad;
        const bool valid = pvalid && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        const T* arow = p.A + (valid ? (((int64_t)b * g.H + yy) * g.W + xx) : 0) * p.lda;
        Pack16 a[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
          if (t < Tn) a[t] = load_pack<T>(arow, t * KP + h * HP, p.CA, valid);
        if (MODE == CMODE_FWD && p.in_act) {
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
            if (t < Tn) {
              float f[HP];
              MM::unpack(a[t], f);
#pragma unroll
              for (int e = 0; e < HP; ++e) f[e] = apply_act(f[e], p.in_act);
              a[t] = MM::pack(f);
            }
        }
        const char* wtap = smem + ((size_t)tap * ncp + r) * p.ldw + h * 16;
#pragma unroll
        for (int c = 0; c < CV_MAXCT; ++c)
          if (c < nct) {
            const char* wrow = wtap + (size_t)c * 32 * p.ldw;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
                const Pack16 bb = *reinterpret_cast<const Pack16*>(wrow + t * 32);
                MM::mma(acc[c], a[t], bb);
              }
          }
      }
      // epilogue
#pragma unroll
      for (int c = 0; c < CV_MAXCT; ++c)
        if (c < nct) {
          const int col = n0 + c * 32 + r;
          const float bv = (MODE == CMODE_FWD && p.bias && col < p.Nout) ? p.bias[col] : 0.f;
          if (p.eps_off < 0 || (MODE == CMODE_FWD && g.r > 1)) {
            // element stores: PixelShuffle scatters consecutive channels to different pixels, or no LDS
            // is left for the row-wise bounce (240-channel dgrad)
            if (col < p.Nout) {
#pragma unroll
              for (int v = 0; v < 16; ++v) {
                const int64_t pp = slab * 32 + acc_row(v, h);
                if (pp < P) {
                  if (MODE == CMODE_FWD) {
                    int b2, y2, x2;
                    g.decode(pp, b2, y2, x2);
                    int64_t row; int ch;
                    g.out_rc(b2, y2, x2, col, row, ch);
                    float val = (acc[c][v] + bv) * p.s;
                    if (p.R) val += to_f32<T>(p.R[row * p.ldr + ch]);
                    p.Y[row * p.ldy + ch] = from_f32<T>(val);
                  } else {
                    float val = acc[c][v] * p.s;
                    if (p.in_act) val *= act_grad(to_f32<T>(p.Xa[pp * p.ldxa + col]), p.in_act);
                    if (p.Acc) val += to_f32<T>(p.Acc[pp * p.ldacc + col]);
                    p.Y[pp * p.ldy + col] = from_f32<T>(val);
                  }
                }
              }
            }
          } else {
            float vals[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) vals[v] = (MODE == CMODE_FWD) ? (acc[c][v] + bv) * p.s : acc[c][v] * p.s;
            TileEpilogue ep{};
            if (MODE == CMODE_FWD) {
              ep.R = p.R; ep.ldr = p.ldr; ep.Y = p.Y; ep.ldy = p.ldy;
            } else {
              ep.Xa = p.Xa; ep.ldxa = p.ldxa; ep.act = p.in_act; ep.Y = p.Y; ep.ldy = p.ldy; ep.Acc = p.Acc; ep.ldacc = p.ldacc;
            }
            tile_store_rows<T>(eps, vals, lane, slab * 32, P, n0 + c * 32, p.Nout, ep);
          }
        }
    }
  }
}

template <typename T, int MODE>
int launch_conv(ConvArgs<T>& p, hipStream_t st, const char* what) {
  using MM = Mma<T>;
  p.Tn = (p.CA + MM::KP - 1) / MM::KP;
  if (p.Tn > 16) return RDST_ENOTSUP;
  p.ldw = lds_row_bytes(p.CA, sizeof(T));
  const int ntap = p.g.ks * p.g.ks;
  const int npad = ((p.Nout + 31) / 32) * 32;
  bool rows = true;  // row-wise epilogue needs 8 x 4 KB of LDS besides the weights
  int nch = (int)((124 * 1024) / ((size_t)ntap * p.ldw)) / 32 * 32;
  if (nch < 32) {
    rows = false;
    nch = (int)((156 * 1024) / ((size_t)ntap * p.ldw)) / 32 * 32;
  }
  if (nch > 32 * CV_MAXCT) nch = 32 * CV_MAXCT;
  if (nch < 32) return RDST_ENOTSUP;
  if (nch > npad) nch = npad;
  p.nch = nch;
  p.eps_off = rows ? ntap * nch * p.ldw : -1;
  const size_t smem = (size_t)ntap * nch * p.ldw + (rows ? 8 * 4096 : 0);
  const int64_t nslabs = (p.g.pixels() + 31) / 32;
  int64_t grid = (nslabs + 7) / 8;
  if (grid > 256) grid = 256;
#define RDST_CONV_LAUNCH(TM)                                                                                         \
  {                                                                                                                  \
    auto kern = conv_mfma_kernel<T, TM, MODE>;                                                                       \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);                                          \
  }
  if (p.Tn <= 4) RDST_CONV_LAUNCH(4) else if (p.Tn <= 8) RDST_CONV_LAUNCH(8) else RDST_CONV_LAUNCH(16)
#undef RDST_CONV_LAUNCH
  return rdst_launch_status(what);
}

template <typename T> bool rows_ok(const void*, int64_t) { return true; }  // load_pack checks alignment per access

// dY (B, H*r, W*r, C/r^2) pixel-shuffled rows -> plain (B*H*W, C) rows in nn.PixelShuffle channel order
template <typename T>
__global__ void __launch_bounds__(256) unshuffle_kernel(const T* __restrict__ dY, int64_t ld, T* __restrict__ out,
                                                        ConvGeom g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over pixels * Cout, co fastest
  const int64_t tot = g.pixels() * g.Cout;
  if (i >= tot) return;
  const int co = (int)(i % g.Cout);
  const int64_t pix = i / g.Cout;
  int b, y, x;
  g.decode(pix, b, y, x);
  int64_t row; int c;
  g.out_rc(b, y, x, co, row, c);
  out[i] = dY[row * ld + c];
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[co][ci][ky][kx] = s * sum_p dY[p][co] * in_act(X)[p + (ky,kx) - pad][ci]
// ------------------------------------------------------------------------------------------------
constexpr int CW_MAXT = 6;
constexpr int CW_STRIPE = 32;

template <typename T>
struct ConvWgradArgs {
  const T* X; int64_t ldx; int in_act;
  const T* dY; int64_t lddy;  // plain rows (B*H*W, Cout)
  float* slab;                // [nm][ks][Cout][ks][CinP]
  ConvGeom g;
  int64_t pix_per_wg;
  int ldn, ldk;               // LDS strides (bytes)
  int NT, KT, CinP;           // CinP = KT*32 (padded Cin incl. the ones column)
  int ones_col;               // column of the centre tap carrying 1.0 (d(bias)), or -1
};

template <typename T>
__global__ void __launch_bounds__(512) conv_wgrad_mfma_kernel(const ConvWgradArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T>;
  constexpr int HP = MM::HP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const ConvGeom g = p.g;
  const int ks = g.ks;
  const int ky = blockIdx.y;  // kernel row handled by this workgroup
  char* dYs = smem;
  char* Xs = smem + (size_t)CW_STRIPE * p.ldn;  // [kx][32][ldk]
  f32x16 acc[CW_MAXT];
#pragma unroll
  for (int j = 0; j < CW_MAXT; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int ntiles = p.NT * ks * p.KT;
  const int64_t P = g.pixels();
  const int64_t p_begin = (int64_t)blockIdx.x * p.pix_per_wg;
  const int64_t p_end = (p_begin + p.pix_per_wg < P) ? p_begin + p.pix_per_wg : P;
  const int npk = p.NT * 32 / HP, kpk = p.KT * 32 / HP;

  for (int64_t p0 = p_begin; p0 < p_end; p0 += CW_STRIPE) {
    __syncthreads();
    for (int idx = tid; idx < CW_STRIPE * npk; idx += 512) {
      const int row = idx / npk, pk = idx - row * npk;
      const bool valid = p0 + row < p_end;
      const Pack16 v = load_pack<T>(p.dY + (valid ? (p0 + row) : 0) * p.lddy, pk * HP, g.Cout, valid);
      *reinterpret_cast<Pack16*>(dYs + (size_t)row * p.ldn + pk * 16) = v;
    }
    for (int idx = tid; idx < ks * CW_STRIPE * kpk; idx += 512) {
      const int pk = idx % kpk;
      const int rest = idx / kpk;
      const int row = rest % CW_STRIPE, kx = rest / CW_STRIPE;
      const int64_t pix = p0 + row;
      const bool pvalid = pix < p_end;
      int b, y, x;
      g.decode(pvalid ? pix : 0, b, y, x);
      const int yy = y + ky - g.pad, xx = x + kx - g.pad;
      const bool valid = pvalid && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      const int k0 = pk * HP;
      Pack16 v = load_pack<T>(p.X + (valid ? (((int64_t)b * g.H + yy) * g.W + xx) : 0) * p.ldx, k0, g.Cin, valid);
      float f[HP];
      MM::unpack(v, f);
      if (p.in_act) {
#pragma unroll
        for (int e = 0; e < HP; ++e) f[e] = apply_act(f[e], p.in_act);
      }
      if (p.ones_col >= 0 && ky == g.pad && kx == g.pad && pvalid) {
#pragma unroll
        for (int e = 0; e < HP; ++e)
          if (k0 + e == p.ones_col) f[e] = 1.0f;
      }
      *reinterpret_cast<Pack16*>(Xs + ((size_t)kx * CW_STRIPE + row) * p.ldk + pk * 16) = MM::pack(f);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CW_MAXT; ++j) {
      const int ti = wave + 8 * j;
      if (ti < ntiles) {
        const int nt = ti / (ks * p.KT);
        const int rem = ti - nt * (ks * p.KT);
        const int kx = rem / p.KT, kt = rem - kx * p.KT;
        const char* Xk = Xs + (size_t)kx * CW_STRIPE * p.ldk;
        if constexpr (sizeof(T) == 2) {
          const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
          const int colA = nt * 32 + 16 * (gq & 1) + 4 * pp;
          const int colB = kt * 32 + 16 * (gq & 1) + 4 * pp;
#pragma unroll
          for (int ms = 0; ms < CW_STRIPE / 16; ++ms) {
            const int rowb = ms * 16 + 8 * h + q;
            typedef __attribute__((address_space(3))) s16x4_t* lds_p;
            const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(dYs + (size_t)rowb * p.ldn + colA * 2));
            const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(dYs + (size_t)(rowb + 4) * p.ldn + colA * 2));
            const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xk + (size_t)rowb * p.ldk + colB * 2));
            const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xk + (size_t)(rowb + 4) * p.ldk + colB * 2));
            const uint2 ua0 = __builtin_bit_cast(uint2, a0), ua1 = __builtin_bit_cast(uint2, a1);
            const uint2 ub0 = __builtin_bit_cast(uint2, b0), ub1 = __builtin_bit_cast(uint2, b1);
            Pack16 a, bq;
            a.w[0] = ua0.x; a.w[1] = ua0.y; a.w[2] = ua1.x; a.w[3] = ua1.y;
            bq.w[0] = ub0.x; bq.w[1] = ub0.y; bq.w[2] = ub1.x; bq.w[3] = ub1.y;
            MM::mma(acc[j], a, bq);
          }
        } else {
          const float* dYf = reinterpret_cast<const float*>(dYs);
          const float* Xf = reinterpret_cast<const float*>(Xk);
          const int lda = p.ldn / 4, ldb = p.ldk / 4;
#pragma unroll 8
          for (int s2 = 0; s2 < CW_STRIPE / 2; ++s2) {
            const float av = dYf[(2 * s2 + h) * lda + nt * 32 + r];
            const float bv = Xf[(2 * s2 + h) * ldb + kt * 32 + r];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
  // slab[m-block][ky][co][kx][CinP]
  float* my = p.slab + (((int64_t)blockIdx.x * ks + ky) * g.Cout) * ks * p.CinP;
#pragma unroll
  for (int j = 0; j < CW_MAXT; ++j) {
    const int ti = wave + 8 * j;
    if (ti < ntiles) {
      const int nt = ti / (ks * p.KT);
      const int rem = ti - nt * (ks * p.KT);
      const int kx = rem / p.KT, kt = rem - kx * p.KT;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int co = nt * 32 + acc_row(v, h);
        if (co < g.Cout) my[((int64_t)co * ks + kx) * p.CinP + kt * 32 + r] = acc[j][v];
      }
    }
  }
}

// dW[co][ci][ky][kx] = s * sum_m slab[m][ky][co][kx][ci];  dbias[co] = s * sum_m slab[m][pad][co][pad][ones_col]
__global__ void __launch_bounds__(256) conv_wgrad_reduce_kernel(const float* __restrict__ slab, int nm, ConvGeom g, int CinP,
                                                                int ones_col, float s, float* __restrict__ dW,
                                                                float* __restrict__ dbias) {
  const int ks = g.ks;
  const int64_t per_m = (int64_t)ks * g.Cout * ks * CinP;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_m) return;
  const int ci = (int)(i % CinP);
  int64_t rest = i / CinP;
  const int kx = (int)(rest % ks); rest /= ks;
  const int co = (int)(rest % g.Cout);
  const int ky = (int)(rest / g.Cout);
  const bool is_w = ci < g.Cin;
  const bool is_b = (ci == ones_col && ky == g.pad && kx == g.pad);
  if (!is_w && !is_b) return;
  float a = 0.f;
  for (int m = 0; m < nm; ++m) a += slab[(int64_t)m * per_m + i];
  if (is_w) {
    if (dW) dW[(((int64_t)co * g.Cin + ci) * ks + ky) * ks + kx] = a * s;
  } else if (dbias) {
    dbias[co] = a * s;
  }
}

}  // namespace

template <typename T>
int conv_fwd_mfma(const T* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const T* R, int64_t ldr,
                  T* Y, int64_t ldy, const ConvGeom& g, float s, hipStream_t st) {
  if (mfma_disabled() || !rows_ok<T>(X, ldx) || g.Cin < 8) return RDST_ENOTSUP;
  ConvArgs<T> p{};
  p.A = X; p.lda = ldx; p.CA = g.Cin; p.Wc = Wc; p.bias = bias; p.R = R; p.ldr = ldr; p.Y = Y; p.ldy = ldy;
  p.in_act = in_act; p.g = g; p.Nout = g.Cout; p.s = s;
  return launch_conv<T, CMODE_FWD>(p, st, "conv_fwd_mfma");
}

size_t conv_mfma_scratch_bytes(const ConvGeom& g) {
  // un-shuffled dY (bf16 or fp32) + wgrad slab
  const size_t unsh = g.r > 1 ? (size_t)g.pixels() * g.Cout * 4 : 0;
  const int CinP = ((g.Cin + 1 + 31) / 32) * 32;
  const size_t slab = (size_t)128 * g.ks * g.Cout * g.ks * CinP * sizeof(float);
  return unsh + slab + 256;
}

template <typename T>
const T* plain_dy(const T* dY, int64_t lddy, const ConvGeom& g, void* scratch, int64_t& ld_out, hipStream_t st, int& rc) {
  rc = 0;
  if (g.r == 1) { ld_out = lddy; return dY; }
  T* tmp = reinterpret_cast<T*>(scratch);
  const int64_t tot = g.pixels() * g.Cout;
  hipLaunchKernelGGL((unshuffle_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dY, lddy, tmp, g);
  rc = rdst_launch_status("unshuffle");
  ld_out = g.Cout;
  return tmp;
}

template <typename T>
int conv_dgrad_mfma(const T* X, int64_t ldx, int in_act, const float* Wc, const T* dYp, int64_t lddyp, T* dX,
                    int64_t lddx, const T* acc, int64_t ldacc, const ConvGeom& g, float s, hipStream_t st) {
  // dYp must already be plain rows (B*H*W, Cout)
  if (mfma_disabled() || !rows_ok<T>(dYp, lddyp) || g.Cout < 8) return RDST_ENOTSUP;
  ConvArgs<T> p{};
  p.A = dYp; p.lda = lddyp; p.CA = g.Cout; p.Wc = Wc; p.Y = dX; p.ldy = lddx; p.Xa = X; p.ldxa = ldx;
  p.in_act = in_act; p.Acc = acc; p.ldacc = ldacc; p.g = g; p.Nout = g.Cin; p.s = s;
  return launch_conv<T, CMODE_DGRAD>(p, st, "conv_dgrad_mfma");
}

template <typename T>
int conv_wgrad_mfma(const T* X, int64_t ldx, int in_act, const T* dYp, int64_t lddyp, float* dW, float* dbias,
                    float* slab, const ConvGeom& g, float s, hipStream_t st) {
  if (mfma_disabled() || !rows_ok<T>(X, ldx) || !rows_ok<T>(dYp, lddyp)) return RDST_ENOTSUP;
  ConvWgradArgs<T> p{};
  p.X = X; p.ldx = ldx; p.in_act = in_act; p.dY = dYp; p.lddy = lddyp; p.slab = slab; p.g = g;
  p.NT = (g.Cout + 31) / 32;
  p.KT = (g.Cin + 1 + 31) / 32;   // room for the ones column
  p.CinP = p.KT * 32;
  p.ones_col = g.Cin;
  if (p.NT * g.ks * p.KT > 8 * CW_MAXT) return RDST_ENOTSUP;
  auto stride = [](int elems) {
    const int b = elems * (int)sizeof(T);
    if (sizeof(T) == 4) return b;
    return b <= 64 ? 64 : ((b - 64 + 255) / 256) * 256 + 64;
  };
  p.ldn = stride(p.NT * 32);
  p.ldk = stride(p.KT * 32);
  const size_t smem = (size_t)CW_STRIPE * (p.ldn + (size_t)g.ks * p.ldk);
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  const int64_t P = g.pixels();
  int64_t nm = (P + CW_STRIPE - 1) / CW_STRIPE;
  const int64_t cap = 128;
  if (nm > cap) nm = cap;
  p.pix_per_wg = (((P + nm - 1) / nm + CW_STRIPE - 1) / CW_STRIPE) * CW_STRIPE;
  nm = (P + p.pix_per_wg - 1) / p.pix_per_wg;
  auto kern = conv_wgrad_mfma_kernel<T>;
  if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3((unsigned)nm, (unsigned)g.ks), dim3(512), smem, st, p);
  if (int rc = rdst_launch_status("conv_wgrad_mfma")) return rc;
  const int64_t per_m = (int64_t)g.ks * g.Cout * g.ks * p.CinP;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((per_m + 255) / 256)), dim3(256), 0, st, slab, (int)nm, g,
                     p.CinP, p.ones_col, s, dW, dbias);
  return rdst_launch_status("conv_wgrad_reduce");
}

#define INST(T)                                                                                                       \
  template int conv_fwd_mfma<T>(const T*, int64_t, int, const float*, const float*, const T*, int64_t, T*, int64_t,  \
                                const ConvGeom&, float, hipStream_t);                                                \
  template int conv_dgrad_mfma<T>(const T*, int64_t, int, const float*, const T*, int64_t, T*, int64_t, const T*,    \
                                  int64_t, const ConvGeom&, float, hipStream_t);                                     \
  template int conv_wgrad_mfma<T>(const T*, int64_t, int, const T*, int64_t, float*, float*, float*, const ConvGeom&, \
                                  float, hipStream_t);                                                               \
  template const T* plain_dy<T>(const T*, int64_t, const ConvGeom&, void*, int64_t&, hipStream_t, int&);
INST(float)
INST(bf16)
