// Convolution geometry + hooks for the MFMA fast paths (conv_mfma.hip); a hook returns RDST_ENOTSUP
// when it does not cover the shape and conv.hip falls back to the generic functor GEMM.
#pragma once
#include "common.h"

struct ConvGeom {
  int B, H, W;    // input pixel grid (the conv runs at this resolution)
  int Cin, Cout;  // Cout counts channels BEFORE the pixel shuffle
  int ks, pad;
  int r;          // PixelShuffle factor folded into the output addressing (1 = none)
  __host__ __device__ __forceinline__ int64_t pixels() const { return (int64_t)B * H * W; }
  __device__ __forceinline__ void decode(int64_t p, int& b, int& y, int& x) const {
    const int hw = H * W;
    b = (int)(p / hw);
    const int q = (int)(p - (int64_t)b * hw);
    y = q / W;
    x = q - y * W;
  }
  // output row / channel of conv output channel `co` at input pixel (b,y,x):
  // nn.PixelShuffle(r): channel c*r*r + i*r + j -> channel c at (y*r+i, x*r+j)   (networks/common.py:132)
  __device__ __forceinline__ void out_rc(int b, int y, int x, int co, int64_t& row, int& c) const {
    if (r == 1) {
      row = ((int64_t)b * H + y) * W + x;
      c = co;
      return;
    }
    const int r2 = r * r;
    c = co / r2;
    const int rem = co - c * r2, i = rem / r, j = rem - i * r;
    row = ((int64_t)b * (H * r) + y * r + i) * (int64_t)(W * r) + x * r + j;
  }
};

template <typename T>
int conv_fwd_mfma(const T* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const T* R, int64_t ldr,
                  T* Y, int64_t ldy, const ConvGeom& g, float s, hipStream_t st);
// The two backward hooks take dY as PLAIN rows (B*H*W, Cout): plain_dy() un-shuffles a pixel-shuffled
// dY into scratch once (or returns dY itself when r == 1).
template <typename T>
const T* plain_dy(const T* dY, int64_t lddy, const ConvGeom& g, void* scratch, int64_t& ld_out, hipStream_t st, int& rc);
template <typename T>
int conv_dgrad_mfma(const T* X, int64_t ldx, int in_act, const float* Wc, const T* dYp, int64_t lddyp, T* dX,
                    int64_t lddx, const T* acc, int64_t ldacc, const ConvGeom& g, float s, hipStream_t st);
template <typename T>
int conv_wgrad_mfma(const T* X, int64_t ldx, int in_act, const T* dYp, int64_t lddyp, float* dW, float* dbias,
                    float* slab, const ConvGeom& g, float s, hipStream_t st);
size_t conv_mfma_scratch_bytes(const ConvGeom& g);

// one-output-channel 3x3 conv (the tail conv) and the 1 -> 1 channel 1x1 conv (MeanShift), bf16: plain vector kernels
// (conv_c1.hip); RDST_ENOTSUP for other shapes
int conv_c1_fwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const bf16* R, int64_t ldr, bf16* Y,
                     int64_t ldy, const ConvGeom& g, float s, hipStream_t st);
int conv_c1_bwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const bf16* dY, int64_t lddy, bf16* dX,
                     int64_t lddx, const bf16* acc, int64_t ldacc, float* dW, float* dbias, float* slab, const ConvGeom& g,
                     float s, hipStream_t st);
size_t conv_c1_slab_floats(int Cin);
// the mirror shape, one INPUT channel -> Cout (the head conv on a single-channel image), bf16: forward and weight gradient
// on the same tile kernels; RDST_ENOTSUP for other shapes
int conv_in1_fwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const bf16* R, int64_t ldr, bf16* Y,
                      int64_t ldy, const ConvGeom& g, float s, hipStream_t st);
int conv_in1_wgrad_bf16(const bf16* X, int64_t ldx, int in_act, const bf16* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                        const ConvGeom& g, float s, hipStream_t st);
size_t conv_in1_slab_floats(int Cout);

// register-stationary 3x3 kernels for the E1 shapes, bf16 (conv3_mfma.hip); RDST_ENOTSUP for everything else.
// wpack: conv3_pack_bytes(Cin, Cout) bytes of 16-byte aligned device scratch (NULL -> RDST_ENOTSUP).
size_t conv3_pack_bytes(int Cin, int Cout);
int conv3_fwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const bf16* R, int64_t ldr,
                   bf16* Y, int64_t ldy, const ConvGeom& g, float s, void* wpack, bool prepacked, hipStream_t st);
int conv3_dgrad_bf16(const float* Wc, const bf16* dY, int64_t lddy, bf16* dX, int64_t lddx, const bf16* acc, int64_t ldacc,
                     int in_act, const ConvGeom& g, float s, void* wpack, hipStream_t st);
size_t conv3_wgrad_slab_bytes(int Cin, int Cout);
int conv3_wgrad_bf16(const bf16* X, int64_t ldx, int in_act, const bf16* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                     const ConvGeom& g, float s, hipStream_t st);

// the same register-stationary kernels on fp32 rows in the RDST_F32X3 arithmetic (conv3x_mfma.hip); wpack: conv3x_pack_bytes bytes
size_t conv3x_pack_bytes(int Cin, int Cout);
int conv3x_fwd_shape(int Cin, int Cout, int ks, int r, bool has_res, int in_act);   // 0 = not covered
int conv3x_fwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const float* R, int64_t ldr,
                   float* Y, int64_t ldy, const ConvGeom& g, float s, void* wpack, bool prepacked, hipStream_t st);
int conv3x_dgrad_f32(const float* Wc, const float* dY, int64_t lddy, float* dX, int64_t lddx, const float* acc, int64_t ldacc,
                     int in_act, const ConvGeom& g, float s, void* wpack, hipStream_t st);
int conv3x_wgrad_f32(const float* X, int64_t ldx, int in_act, const float* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                     const ConvGeom& g, float s, hipStream_t st);   // conv3x_wgrad.hip; slab: conv3_wgrad_slab_bytes(Cin, Cout)
// the one-channel 3x3 convolutions on fp32 rows (conv_c1x.hip: plain fp32 FMAs, both fp32 modes): tail conv wide -> 1 and head conv 1 -> wide
size_t conv_c1x_slab_floats(int C);
int conv_c1x_fwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const float* R, int64_t ldr, float* Y,
                     int64_t ldy, const ConvGeom& g, float s, hipStream_t st);
int conv_in1x_fwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const float* R, int64_t ldr, float* Y,
                      int64_t ldy, const ConvGeom& g, float s, hipStream_t st);
int conv_c1x_bwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* dY, int64_t lddy, float* dX, int64_t lddx,
                     const float* acc, int64_t ldacc, float* dW, float* dbias, float* slab, const ConvGeom& g, float s, hipStream_t st);
int conv_in1x_wgrad_f32(const float* X, int64_t ldx, int in_act, const float* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                        const ConvGeom& g, float s, hipStream_t st);
