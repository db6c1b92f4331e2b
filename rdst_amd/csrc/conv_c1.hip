// K6 specialisation: the 3x3 tail conv to ONE output channel (rdst_variations.py:1303, `default_conv(cf, 1, 3)` at
// the 4x resolution: 60 -> 1 on 256x256 pixels), bf16 rows.  With a single output channel there is no GEMM to
// speak of — 540 multiply-adds per pixel against 120 B of input — so the matrix cores are the wrong tool (the
// MFMA weight-gradient kernel ran 559 us with 31 of 32 accumulator columns idle, the generic functor GEMM 200-400
// us); these are plain vector kernels bound by the 252 MB of x / dX:
//   lane = (pixel slot 0..7, channel group 0..7); a lane keeps the 8 x 9 weights of ITS channel group in registers
//   for the whole kernel; a row's 120 B are 8 lanes x 16 B; persistent workgroups over 8 x 32 pixel tiles whose halo
//   (input rows or dY values) is staged in LDS once (see C1_TH below), prefetched one tile ahead in registers.
//   fwd   : sliding 3 x 3 window of 16-B chunks, 36 v_dot2_f32_bf16 per pixel and lane (weights as bf16 pairs), sum over
//           the pixel's 8 lanes on the DPP path, one 2-byte store per lane for the run's 8 pixels       178 -> 86 us
//   dgrad : sliding 3 x 3 window of dY floats, 36 v_pk_fma_f32, one 16-B row-chunk store                113 -> 88 us
//   wgrad : input chunk read ONCE per pixel, 36 packed accumulators per lane over all its pixels, reduced over lanes /
//           waves once at the end, per-workgroup partials summed in fixed order (deterministic)          114 -> 63 us
#include "common.h"
#include "gemm_valu.h"
#include "conv.h"
#include "mfma.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct C1Args {
  const bf16* X; int64_t ldx;       // (B*H*W, Cin) rows: conv input (fwd, wgrad)
  const float* W;                   // (1, Cin, 3, 3)
  const float* bias;
  const bf16* dY; int64_t lddy;     // (B*H*W, 1)
  bf16* Y; int64_t ldy;             // fwd out (B*H*W, 1)
  bf16* dX; int64_t lddx; const bf16* Acc; int64_t ldacc;
  float* slab;                      // wgrad: [grid][Cin*9 + 1]
  ConvGeom g; float s;
  int tiles_x, tiles_y; int64_t ntiles;
};

constexpr int C1_THREADS = 256;
// A workgroup works on 8 x 32 pixel tiles of one image; the tile's 10 x 34 halo is staged in LDS ONCE — the input rows
// (forward: 9 slots of 16 B per pixel, 144 B: two pixels 8 columns apart sit on disjoint bank halves) or the dY values
// (backward: one float per pixel, already scaled) — zero outside the image, so the nine taps need no masks.  The first
// version read the nine neighbours from global memory per pixel: 9x the 252 MB of x through L2 (forward, 178 us) and
// nine 2-byte broadcast loads per pixel and lane (backward, 113 / 114 us: bound by the address unit, not by HBM).
// 32 groups of 8 lanes (lane = channel group) per workgroup; a group walks a run of 8 pixels of one tile row with a
// sliding 3 x 3 window in registers: three new LDS reads per pixel.
constexpr int C1_TH = 8, C1_TW = 32, C1_HH = C1_TH + 2, C1_HW = C1_TW + 2, C1_NPIX = C1_HH * C1_HW, C1_PS = 144;

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
__device__ __forceinline__ void c1_weights(const float* W, const C1Lane& l, float (&w)[8][9]) {
  // the lane's 8 x 9 weights are 72 CONTIGUOUS floats of W (1, Cin, 3, 3): eighteen 16-byte loads, all in flight at once,
  // masked afterwards (72 scalar loads became 72 branches, each with its own wait: ~50 us of prologue in a persistent grid)
  u32x4_a4 q[18];
  const u32x4_a4* src = reinterpret_cast<const u32x4_a4*>(W + l.c0 * 9);
#pragma unroll
  for (int k = 0; k < 18; ++k) q[k] = src[k];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int i = e * 9 + t;
      const uint32_t u = (i & 3) == 0 ? q[i >> 2].x : (i & 3) == 1 ? q[i >> 2].y : (i & 3) == 2 ? q[i >> 2].z : q[i >> 2].w;
      w[e][t] = (l.on && e >= l.lo) ? __uint_as_float(u) : 0.f;
    }
}
// sum over the 8 lanes of a group on the DPP path (no LDS traffic): quad swaps, then the half-row mirror
__device__ __forceinline__ float c1_sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  return v;
}
struct C1Tile { int b, y0, x0; };
// Tiles of a workgroup: a CONTIGUOUS chunk of the row-major tile order, and the chunks of the workgroups of one XCD
// (blockIdx.x mod 8) are neighbours too: the halo rows two tiles share are then found in that XCD's L2
struct C1Range { int64_t t0, t1; };
__device__ __forceinline__ C1Range c1_range(const C1Args& p) {
  const int grid = gridDim.x, bid = blockIdx.x;
  const int v = grid % 8 == 0 ? (bid & 7) * (grid / 8) + (bid >> 3) : bid;
  const int64_t per = (p.ntiles + grid - 1) / grid;
  C1Range r;
  r.t0 = v * per;
  r.t1 = r.t0 + per < p.ntiles ? r.t0 + per : p.ntiles;
  return r;
}
__device__ __forceinline__ C1Tile c1_tile(const C1Args& p, int64_t t) {
  C1Tile q;
  const int per = p.tiles_x * p.tiles_y;
  q.b = (int)(t / per);
  const int r = (int)(t - (int64_t)q.b * per), ty = r / p.tiles_x;
  q.y0 = ty * C1_TH;
  q.x0 = (r - ty * p.tiles_x) * C1_TW;
  return q;
}
// dY halo of a tile as floats (times `mul`), zero outside the image: loaded into registers (for the next tile, while the
// current one is computed), stored to LDS behind the barrier
constexpr int C1_NDY = (C1_NPIX + C1_THREADS - 1) / C1_THREADS;
__device__ __forceinline__ void c1_dy_load(const C1Args& p, const C1Tile& t, float (&v)[C1_NDY], float mul, int tid) {
  const ConvGeom& g = p.g;
#pragma unroll
  for (int k = 0; k < C1_NDY; ++k) {
    const int i0 = tid + k * C1_THREADS, i = i0 < C1_NPIX ? i0 : 0;
    const int hy = i / C1_HW, hx = i - hy * C1_HW, yy = t.y0 + hy - 1, xx = t.x0 + hx - 1;
    const bool inb = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
    const int yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= g.W ? g.W - 1 : xx);
    const float x = __bfloat162float(p.dY[(((int64_t)t.b * g.H + yc) * g.W + xc) * p.lddy]);
    v[k] = inb ? x * mul : 0.f;
  }
}
__device__ __forceinline__ void c1_dy_store(float* dyt, const float (&v)[C1_NDY], int tid) {
#pragma unroll
  for (int k = 0; k < C1_NDY; ++k)
    if (tid + k * C1_THREADS < C1_NPIX) dyt[tid + k * C1_THREADS] = v[k];
}

__global__ void __launch_bounds__(C1_THREADS) conv_c1_fwd_kernel(const C1Args p) {
  __shared__ __attribute__((aligned(16))) char tile[C1_NPIX * C1_PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ps = lane >> 3, grp = lane & 7;
  const ConvGeom g = p.g;
  const C1Lane l = c1_lane(grp, g.Cin);
  // the lane's 8 x 9 weights as bf16 PAIRS of neighbouring channels: a tap of a 16-B chunk is four v_dot2_f32_bf16
  // (exact products, fp32 accumulation; the weights are rounded to bf16 like every weight image of the bf16 mode)
  uint32_t w2[4][9];
  {
    float w[8][9];
    c1_weights(p.W, l, w);
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2)
#pragma unroll
      for (int t = 0; t < 9; ++t) w2[e2][t] = pack_bf16x2(w[2 * e2][t], w[2 * e2 + 1][t]);
  }
  const float b0 = p.bias ? p.bias[0] : 0.f;
  const int gidx = wave * 8 + ps, grow = gidx >> 2, gcol0 = (gidx & 3) * 8;
  // the halo chunks of the NEXT tile are loaded into registers (all NV loads in flight at once) while the current tile
  // is computed from LDS, and written to LDS behind the barrier that ends the computation
  constexpr int NV = (C1_NPIX * 8 + C1_THREADS - 1) / C1_THREADS;
  u32x4_a4 stg[NV];
  auto stage_load = [&](const C1Tile& t) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {   // (i & 7 == grp: the chunk's channel window is the lane's own)
      const int i = tid + k * C1_THREADS, pix = (i < C1_NPIX * 8 ? i : tid) >> 3;
      const int hy = pix / C1_HW, hx = pix - hy * C1_HW, yy = t.y0 + hy - 1, xx = t.x0 + hx - 1;
      const bool inb = l.on && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      const int yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= g.W ? g.W - 1 : xx);
      u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p.X + (((int64_t)t.b * g.H + yc) * g.W + xc) * p.ldx + l.c0);
      v.x = inb ? v.x : 0u; v.y = inb ? v.y : 0u; v.z = inb ? v.z : 0u; v.w = inb ? v.w : 0u;
      stg[k] = v;
    }
  };
  const C1Range rg = c1_range(p);
  if (rg.t0 < rg.t1) stage_load(c1_tile(p, rg.t0));
  for (int64_t tix = rg.t0; tix < rg.t1; ++tix) {
    const C1Tile t = c1_tile(p, tix);
    __syncthreads();   // the previous tile has been read
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tid + k * C1_THREADS;
      if (i < C1_NPIX * 8) {
        Pack16 vv; vv.w[0] = stg[k].x; vv.w[1] = stg[k].y; vv.w[2] = stg[k].z; vv.w[3] = stg[k].w;
        *reinterpret_cast<Pack16*>(tile + (i >> 3) * C1_PS + grp * 16) = vv;
      }
    }
    __syncthreads();
    if (tix + 1 < rg.t1) stage_load(c1_tile(p, tix + 1));
    const char* base = tile + (grow * C1_HW + gcol0) * C1_PS + grp * 16;
    Pack16 win[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < 2; ++c) win[dy][c] = *reinterpret_cast<const Pack16*>(base + (dy * C1_HW + c) * C1_PS);
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) win[dy][(j + 2) % 3] = *reinterpret_cast<const Pack16*>(base + (dy * C1_HW + j + 2) * C1_PS);
      float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const Pack16& v = win[dy][(j + dx) % 3];
          const int tp = dy * 3 + dx;
          a[dy] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, v.w[0]), __builtin_bit_cast(bf16x2_t, w2[0][tp]), a[dy], false);
          a[dy] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, v.w[1]), __builtin_bit_cast(bf16x2_t, w2[1][tp]), a[dy], false);
          a[dy] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, v.w[2]), __builtin_bit_cast(bf16x2_t, w2[2][tp]), a[dy], false);
          a[dy] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, v.w[3]), __builtin_bit_cast(bf16x2_t, w2[3][tp]), a[dy], false);
        }
      const float acc = c1_sum8((a[0] + a[1]) + a[2]);
      mine = grp == j ? acc : mine;   // lane `grp` of the group keeps pixel `grp` of the run: one 2-byte store per lane
    }
    const int y = t.y0 + grow, x = t.x0 + gcol0 + grp;
    if (y < g.H && x < g.W) p.Y[(((int64_t)t.b * g.H + y) * g.W + x) * p.ldy] = __float2bfloat16((mine + b0) * p.s);
  }
}

// dX[q][c] = s * sum_taps dY[q - (tap offset)] W[c][tap] (+ Acc): window element (r, c) of the halo, rows grow + r, columns
// gcol0 + j + c, is the dY of tap (ky, kx) = (2 - r, 2 - c), i.e. tap index 8 - 3r - c.
// HEAD: the MIRROR problem, the forward of a conv from ONE input channel to Cin outputs (the head conv of RDSTSR,
// rdst_variations.py:1213 `default_conv(in_chans, embed_dim, 3)` with in_chans = 1): Y[p][c] = (sum_taps x[p + (tap offset)]
// W[c][tap] + bias[c]) s (+ R) — the same window with tap index 3r + c, a bias per channel.  (The names keep the data-gradient
// reading: dY = the one-channel tensor, dX = the Cin-channel one, Acc = the residual.)
template <bool HEAD>
__global__ void __launch_bounds__(C1_THREADS) conv_c1_dgrad_kernel(const C1Args p) {
  __shared__ float dyt[C1_NPIX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ps = lane >> 3, grp = lane & 7;
  const ConvGeom g = p.g;
  const C1Lane l = c1_lane(grp, g.Cin);
  f32x2 w2[4][9];
  {
    float w[8][9];
    c1_weights(p.W, l, w);
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2)
#pragma unroll
      for (int t = 0; t < 9; ++t) { w2[e2][t].x = w[2 * e2][t]; w2[e2][t].y = w[2 * e2 + 1][t]; }
  }
  const int gidx = wave * 8 + ps, grow = gidx >> 2, gcol0 = (gidx & 3) * 8;
  float bsv[8];   // HEAD: bias[c] * s of the lane's 8 channels
#pragma unroll
  for (int e = 0; e < 8; ++e) bsv[e] = (HEAD && p.bias && l.on && e >= l.lo) ? p.bias[l.c0 + e] * p.s : 0.f;
  float dyv[C1_NDY];
  const C1Range rg = c1_range(p);
  if (rg.t0 < rg.t1) c1_dy_load(p, c1_tile(p, rg.t0), dyv, p.s, tid);
  for (int64_t tix = rg.t0; tix < rg.t1; ++tix) {
    const C1Tile t = c1_tile(p, tix);
    __syncthreads();
    c1_dy_store(dyt, dyv, tid);
    __syncthreads();
    if (tix + 1 < rg.t1) c1_dy_load(p, c1_tile(p, tix + 1), dyv, p.s, tid);
    const float* base = dyt + grow * C1_HW + gcol0;
    float d[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) d[r][c] = base[r * C1_HW + c];
    const int y = t.y0 + grow;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int r = 0; r < 3; ++r) d[r][(j + 2) % 3] = base[r * C1_HW + j + 2];
      f32x2 o[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) o[e2] = (f32x2)(0.f);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x2 dv = (f32x2)(d[r][(j + c) % 3]);
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) o[e2] = __builtin_elementwise_fma(dv, w2[e2][HEAD ? 3 * r + c : 8 - 3 * r - c], o[e2]);
        }
      const int x = t.x0 + gcol0 + j;
      if (!(y < g.H && x < g.W) || !l.on) continue;
      const int64_t pix = ((int64_t)t.b * g.H + y) * g.W + x;
      float of[8] = {o[0].x, o[0].y, o[1].x, o[1].y, o[2].x, o[2].y, o[3].x, o[3].y};
      if (HEAD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] += bsv[e];
      }
      if (p.Acc) {
        float f[8];
        c1_unpack8(*reinterpret_cast<const u32x4_a4*>(p.Acc + pix * p.ldacc + l.c0), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] += f[e];
      }
      bf16* dst = p.dX + pix * p.lddx + l.c0;
      if (l.lo == 0) {
        u32x4_a4 u;
        u.x = pack_bf16x2(of[0], of[1]); u.y = pack_bf16x2(of[2], of[3]); u.z = pack_bf16x2(of[4], of[5]); u.w = pack_bf16x2(of[6], of[7]);
        *reinterpret_cast<u32x4_a4*>(dst) = u;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e >= l.lo) dst[e] = __float2bfloat16(of[e]);   // the overlapping last group writes only its own channels
      }
    }
  }
}

// dW[c][tap] += X[q][c] dY[q - (tap offset)] (q = input pixel, read ONCE): the same window as the data gradient.
// HEAD: the weight gradient of the one-input-channel conv, dW[c][tap] += dYh[p][c] x[p + (tap offset)] (X = the Cin-channel
// gradient dYh, dY = the one-channel input x), tap index 3r + c, and d(bias)[c] = sum_p dYh[p][c] per channel.
template <bool HEAD>
__global__ void __launch_bounds__(C1_THREADS) conv_c1_wgrad_kernel(const C1Args p) {
  __shared__ float dyt[C1_NPIX];
  __shared__ float part[C1_THREADS / 64][8][81];   // 72 weight taps + d(bias): one scalar (column 72) or, HEAD, 8 per-channel sums (73..80)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ps = lane >> 3, grp = lane & 7;
  const ConvGeom g = p.g;
  const C1Lane l = c1_lane(grp, g.Cin);
  f32x2 acc[4][9];
  f32x2 accbh[4] = {(f32x2)(0.f), (f32x2)(0.f), (f32x2)(0.f), (f32x2)(0.f)};   // HEAD: per-channel d(bias)
  float accb = 0.f;
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[e2][t] = (f32x2)(0.f);
  const int gidx = wave * 8 + ps, grow = gidx >> 2, gcol0 = (gidx & 3) * 8;
  const C1Range rg = c1_range(p);
  for (int64_t tix = rg.t0; tix < rg.t1; ++tix) {
    const C1Tile t = c1_tile(p, tix);
    // the run's eight input chunks first: in flight while the dY halo is staged
    const int y = t.y0 + grow, yc = y < g.H ? y : g.H - 1;
    u32x4_a4 xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int x = t.x0 + gcol0 + j, xc = x < g.W ? x : g.W - 1;
      xv[j] = *reinterpret_cast<const u32x4_a4*>(p.X + (((int64_t)t.b * g.H + yc) * g.W + xc) * p.ldx + l.c0);
    }
    float dyv[C1_NDY];
    c1_dy_load(p, t, dyv, 1.0f, tid);
    __syncthreads();
    c1_dy_store(dyt, dyv, tid);
    __syncthreads();
    const float* base = dyt + grow * C1_HW + gcol0;
    float d[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) d[r][c] = base[r * C1_HW + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int r = 0; r < 3; ++r) d[r][(j + 2) % 3] = base[r * C1_HW + j + 2];
      const bool valid = y < g.H && t.x0 + gcol0 + j < g.W;
      float f[8];
      c1_unpack8(xv[j], f);
      f32x2 f2[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) { f2[e2].x = valid ? f[2 * e2] : 0.f; f2[e2].y = valid ? f[2 * e2 + 1] : 0.f; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x2 dv = (f32x2)(d[r][(j + c) % 3]);
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            acc[e2][HEAD ? 3 * r + c : 8 - 3 * r - c] = __builtin_elementwise_fma(f2[e2], dv, acc[e2][HEAD ? 3 * r + c : 8 - 3 * r - c]);
        }
      accb += valid ? d[1][(j + 1) % 3] : 0.f;   // centre tap = dY at the pixel itself: d(bias)
      if (HEAD) {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) accbh[e2] += f2[e2];
      }
    }
  }
  // lanes of one channel group (pixel slots: lane bits 3..5), then the waves, then one slab row per workgroup
  float accs[8][9];
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2)
#pragma unroll
    for (int t = 0; t < 9; ++t) { accs[2 * e2][t] = acc[e2][t].x; accs[2 * e2 + 1][t] = acc[e2][t].y; }
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float a = accs[e][t];
      a += __shfl_xor(a, 8, 64);
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      accs[e][t] = a;
    }
  accb += __shfl_xor(accb, 8, 64);
  accb += __shfl_xor(accb, 16, 64);
  accb += __shfl_xor(accb, 32, 64);
  float bh[8] = {accbh[0].x, accbh[0].y, accbh[1].x, accbh[1].y, accbh[2].x, accbh[2].y, accbh[3].x, accbh[3].y};
  if (HEAD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bh[e] += __shfl_xor(bh[e], 8, 64);
      bh[e] += __shfl_xor(bh[e], 16, 64);
      bh[e] += __shfl_xor(bh[e], 32, 64);
    }
  }
  if (ps == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 9; ++t) part[wave][grp][e * 9 + t] = accs[e][t];
    part[wave][grp][72] = accb;
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave][grp][73 + e] = bh[e];
  }
  __syncthreads();
  // slab row: [Cin * 9 weight taps][d(bias): 1 value, HEAD: Cin values]
  float* my = p.slab + (int64_t)blockIdx.x * (g.Cin * 9 + (HEAD ? g.Cin : 1));
  for (int i = tid; i < 8 * 81; i += C1_THREADS) {
    const int gi = i / 81, k = i - gi * 81;
    float a = 0.f;
#pragma unroll
    for (int wv = 0; wv < C1_THREADS / 64; ++wv) a += part[wv][gi][k];
    const C1Lane lg = c1_lane(gi, g.Cin);
    if (k == 72) {
      if (!HEAD && gi == 0) my[g.Cin * 9] = a * p.s;
    } else if (k > 72) {
      const int e = k - 73;
      if (HEAD && lg.on && e >= lg.lo) my[g.Cin * 9 + lg.c0 + e] = a * p.s;
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
int c1_grid(C1Args& p, int cap) {   // persistent workgroups over the 8 x 32 tiles
  p.tiles_x = (p.g.W + C1_TW - 1) / C1_TW;
  p.tiles_y = (p.g.H + C1_TH - 1) / C1_TH;
  p.ntiles = (int64_t)p.g.B * p.tiles_x * p.tiles_y;
  return (int)(p.ntiles < cap ? p.ntiles : cap);
}

}  // namespace

int conv_c1_fwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const bf16* R, int64_t ldr, bf16* Y,
                     int64_t ldy, const ConvGeom& g, float s, hipStream_t st) {
  if (g.Cin == 1 && g.Cout == 1 && g.ks == 1 && g.r == 1 && in_act == 0)
    return pw11_launch(X, ldx, Wc, bias, R, ldr, Y, ldy, g.pixels(), s, st, "conv_pw11_fwd");
  if (!c1_ok(g, in_act) || R || ((uintptr_t)X & 3) || (ldx & 1)) return RDST_ENOTSUP;
  C1Args p{};
  p.X = X; p.ldx = ldx; p.W = Wc; p.bias = bias; p.Y = Y; p.ldy = ldy; p.g = g; p.s = s;
  const int grid = c1_grid(p, 512);   // two workgroups per CU (173 registers, 49 KB of LDS)
  hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(grid), dim3(C1_THREADS), 0, st, p);
  return rdst_launch_status("conv_c1_fwd");
}

size_t conv_c1_slab_floats(int Cin) { return (size_t)(1024 + 1) * (Cin * 9 + 1); }   // 1024 partial rows + the reduced row
size_t conv_in1_slab_floats(int Cout) { return (size_t)(1024 + 1) * (Cout * 10); }

// The mirror shape: 3x3 conv from ONE input channel to Cout (the head conv of RDSTSR on a single-channel image), bf16 rows.
// Forward = conv_c1_dgrad_kernel<true>, weight gradient = conv_c1_wgrad_kernel<true> (no data gradient: the input is the image).
bool in1_ok(const ConvGeom& g, int in_act) {
  return g.Cin == 1 && g.ks == 3 && g.r == 1 && in_act == 0 && g.Cout >= 8 && g.Cout <= 64 && (g.Cout & 1) == 0;
}
int conv_in1_fwd_bf16(const bf16* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const bf16* R, int64_t ldr, bf16* Y,
                      int64_t ldy, const ConvGeom& g, float s, hipStream_t st) {
  if (!in1_ok(g, in_act) || ((uintptr_t)Y & 3) || (ldy & 1) || ((uintptr_t)R & 3) || (ldr & 1)) return RDST_ENOTSUP;
  C1Args p{};
  ConvGeom gm = g;
  gm.Cin = g.Cout; gm.Cout = 1;            // the kernels' view: Cin = the wide side
  p.W = Wc; p.bias = bias; p.dY = X; p.lddy = ldx; p.dX = Y; p.lddx = ldy; p.Acc = R; p.ldacc = ldr; p.g = gm; p.s = s;
  const int grid = c1_grid(p, 1024);
  hipLaunchKernelGGL(conv_c1_dgrad_kernel<true>, dim3(grid), dim3(C1_THREADS), 0, st, p);
  return rdst_launch_status("conv_in1_fwd");
}
int conv_in1_wgrad_bf16(const bf16* X, int64_t ldx, int in_act, const bf16* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                        const ConvGeom& g, float s, hipStream_t st) {
  if (!in1_ok(g, in_act) || ((uintptr_t)dY & 3) || (lddy & 1)) return RDST_ENOTSUP;
  C1Args p{};
  ConvGeom gm = g;
  gm.Cin = g.Cout; gm.Cout = 1;
  p.X = dY; p.ldx = lddy; p.dY = X; p.lddy = ldx; p.g = gm; p.s = s; p.slab = slab;
  const int grid = c1_grid(p, 1024);
  hipLaunchKernelGGL(conv_c1_wgrad_kernel<true>, dim3(grid), dim3(C1_THREADS), 0, st, p);
  if (int rc = rdst_launch_status("conv_in1_wgrad")) return rc;
  const int n = g.Cout * 9, row = n + g.Cout;
  float* red = slab + (size_t)grid * row;
  if (int rc = slab_reduce(slab, red, grid, row, st)) return rc;
  if (dW) (void)hipMemcpyAsync(dW, red, sizeof(float) * n, hipMemcpyDeviceToDevice, st);
  if (dbias) (void)hipMemcpyAsync(dbias, red + n, sizeof(float) * g.Cout, hipMemcpyDeviceToDevice, st);
  return 0;
}

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
    const int grid = c1_grid(p, 1024);
    hipLaunchKernelGGL(conv_c1_wgrad_kernel<false>, dim3(grid), dim3(C1_THREADS), 0, st, p);
    if (int rc = rdst_launch_status("conv_c1_wgrad")) return rc;
    const int n = g.Cin * 9;
    // one reduction over [grid][n + 1]: dW then dbias are contiguous in the slab row; the destinations are not
    float* red = slab + (size_t)grid * (n + 1);
    if (int rc = slab_reduce(slab, red, grid, n + 1, st)) return rc;
    if (dW) (void)hipMemcpyAsync(dW, red, sizeof(float) * n, hipMemcpyDeviceToDevice, st);
    if (dbias) (void)hipMemcpyAsync(dbias, red + n, sizeof(float), hipMemcpyDeviceToDevice, st);
  }
  if (dX) {
    const int grid = c1_grid(p, 1024);
    hipLaunchKernelGGL(conv_c1_dgrad_kernel<false>, dim3(grid), dim3(C1_THREADS), 0, st, p);
    if (int rc = rdst_launch_status("conv_c1_dgrad")) return rc;
  }
  return 0;
}
