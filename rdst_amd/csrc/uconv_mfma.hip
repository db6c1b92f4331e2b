// rdst_u_conv: implicit-GEMM convolution on the matrix cores for the frozen seg-UNet of the reference's perceptual loss
// (loss/seg_unet.py:46,83-92: smp.Unet resnet34 encoder + decoder + head).  No im2col and no padded copies:
//   D[pixel][cout] = sum over (tap, cin chunk) of X[pixel + tap offset][cin] . W[tap][cout][cin]
// A workgroup (4 waves) owns a BM x BN tile of D; per step it stages a BM x KB-byte slab of input pixels (one tap, one
// channel chunk: each pixel's chunk is KB contiguous bytes of its NHWC row, read as 16-byte pieces; out-of-image taps read
// zeros) and the matching BN x KB slab of the prepacked weights ([tap][cout][cin], cin contiguous) into a double-buffered
// LDS tile (rows KB + 16 bytes: an odd number of 16-byte slots, conflict-free ds_read_b128), global loads of step i + 1 in
// flight under the MFMAs of step i, ONE barrier per step.  The same kernel covers
//   * stride 1 / 2 forward convolutions (3x3, 1x1 downsample),
//   * the decoder's nearest-x2 upsampling + channel concat, folded into the address of the slab loads (source 1 read at
//     (y/2, x/2), source 2 behind it in the channel index): neither F.interpolate nor torch.cat materialises,
//   * `transposed` = the data gradient of all of those (gather form: out-of-phase taps of a stride-2 layer read zeros),
//   * bias and an addend (residual fan-in of gradients) in the epilogue, which bounces each 32x32 accumulator through a
//     wave-private LDS tile and leaves as 8/16-byte row segments (mfma.h: tile_store_rows).
// T = bf16 (v_mfma_f32_32x32x16_bf16) or fp32 (v_mfma_f32_32x32x2_f32: the exact parity mode), same code (mfma.h: Mma<T>).
// SPLIT (RDST_F32X3, the default of the loss network): fp32 rows in HBM, but every staged 16-byte piece is split into
// bf16 (hi, lo = x - hi) halves on its way into LDS and a k-step is THREE bf16 MFMAs (hi.lo + lo.hi + hi.hi; the dropped lo.lo
// term is 2^-16 relative): products good to ~1e-5 at 3/16 of the fp32-MFMA cost.  The (frozen) weights are split once on
// the host: Wp rows are 64-byte groups [16 bf16 hi][16 bf16 lo] per 16 reduction elements (same bytes as fp32).  Why not plain bf16: the loss DIFFERENCES
// the features of SR and HR, and bf16 activations through 50 train-mode BatchNorm layers carry 1e-2..1e-1 relative error —
// as large as that difference late in training (measured: gradient cosine 0.58 against fp32, tools/segunet_bf16_trace.py).
// Roofline: MFMA for the 3x3 layers (K = 9 Cin >= 576: >= 250 FLOP/B), HBM for the 16/32-channel decoder tail.
#include "mfma.h"
#include <type_traits>
#include <stdlib.h>

namespace {

struct UConvP {
  const char* X1; int64_t ld1; int C1; int up1;
  const char* X2; int64_t ld2; int C2;
  const char* Wp;
  const float* bias;
  const void* add; int64_t ld_add;
  void* Y; int64_t ld_y;
  int B, Hin, Win, Hout, Wout, Cin, Cout, Npad, k, stride, transposed;
  int64_t P;   // output pixels
  int dbg;     // ablation switches (RDST_DEBUG builds only)
  int* stats_blocks;  // host: receives gridDim.x when `stats` is written
  float* stats;       // NULL, or [gridDim.x][2][Cout]: per-workgroup sum / sum of squares of the output channels over the tile's
                      // pixels (the BatchNorm statistics of this convolution's output, halo kernel): summed in fixed order by
                      // rdst_u_bn_stats_from — the separate column-sum pass over the output is gone
  const float* bn1;   // [scale C1][shift C1] (fp32) or NULL: the first source is read through relu(scale x + shift) — the
                      // train-mode BatchNorm + ReLU in front of this convolution, applied while the halo is staged (halo kernel)
};

template <typename T, int BM, int BN, int KB, int WM, int WN, bool SPLIT>
__global__ void __launch_bounds__(256) uconv_kernel(const UConvP p) {
  static_assert(!SPLIT || (sizeof(T) == 4 && KB >= 64), "the split mode stages fp32 rows, 16 elements per k-step");
  constexpr int ES = (int)sizeof(T);
  constexpr int RS = KB + 16;               // LDS row stride (bytes)
  constexpr int PIECES = KB / 16;           // 16-byte pieces per row
  constexpr int RPP = 256 / PIECES;         // rows staged per pass of the workgroup
  constexpr int AP = (BM + RPP - 1) / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int EPC = KB / ES;              // elements per chunk
  constexpr int STAGE = (BM + BN) * RS;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int piece = tid % PIECES, r0 = tid / PIECES;
  const int pad = p.k >> 1;
  const int s2 = p.stride == 2 ? 1 : 0;
  const int H1 = p.Hin >> p.up1, W1 = p.Win >> p.up1;

  // The output pixels whose input rows this thread stages.  cy / cx: the tap-0 input coordinate before the tap offset
  // (forward: oy * stride - pad, taps ADD ky; transposed: oy + pad, taps SUBTRACT ky, then the stride divides).
  // Everything a tap needs per row is prepared ONCE: a k*k-bit validity mask and, for the "regular" sources (no
  // upsampling, no stride-2 data gradient), the pixel index of tap (0, 0) — a tap is then a UNIFORM pixel offset, and a
  // row costs four vector instructions per step (add, bit test, select, 64-bit mad).  (Recomputing coordinates, bounds
  // and 64-bit addresses per tap cost ~1000 cycles of vector issue per step: more than the step's MFMAs.)
  const int sgn = p.transposed ? -1 : 1;
  const int sh = p.transposed ? s2 : 0;        // only the data gradient of a stride-2 layer divides (and skips odd phases)
  const bool reg1 = !p.up1 && sh == 0, reg2 = sh == 0;
  int cy[AP], cx[AP], pb1[AP], pb2[AP], pc1[AP], pc2[AP];
  uint32_t vmask[AP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int row = r0 + j * RPP;
    const int64_t m = m0 + row;
    vmask[j] = 0;
    if (row < BM && m < p.P) {
      const int ox = (int)(m % p.Wout);
      const int64_t t = m / p.Wout;
      const int oy = (int)(t % p.Hout);
      const int b = (int)(t / p.Hout);
      cy[j] = p.transposed ? oy + pad : oy * p.stride - pad;
      cx[j] = p.transposed ? ox + pad : ox * p.stride - pad;
      pb1[j] = b * H1 * W1;
      pb2[j] = b * p.Hin * p.Win;
      for (int tp = 0; tp < p.k * p.k; ++tp) {
        const int ky = tp / p.k, kx = tp - ky * p.k;
        const int ty = cy[j] + sgn * ky, tx = cx[j] + sgn * kx;
        const int iy = ty >> sh, ix = tx >> sh;
        const bool ok = (((ty | tx) & sh) == 0) & ((unsigned)iy < (unsigned)p.Hin) & ((unsigned)ix < (unsigned)p.Win);
        vmask[j] |= ok ? (1u << tp) : 0u;
      }
    } else {
      cy[j] = cx[j] = 0;
      pb1[j] = pb2[j] = 0;
    }
    pc1[j] = pb1[j] + cy[j] * W1 + cx[j];          // regular sources: pixel of tap (0, 0) (may lie outside: masked)
    pc2[j] = pb2[j] + cy[j] * p.Win + cx[j];
  }
  // weight rows: byte offset of the thread's piece inside a (tap, chunk) slab, without the uniform part
  uint32_t wof[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int row = r0 + j * RPP;
    wof[j] = (uint32_t)(((n0 + (row < BN ? row : 0)) * p.Cin) * ES + piece * 16);
  }

  const int CH = p.Cin / EPC;                 // chunks per tap
  const int NIT = p.k * p.k * CH;
  u32x4_a4 ra[AP], rb[BP];
  uint32_t okm = 0;   // bit j: row j of the tile in flight is inside the image (the zero is selected when it is STASHED, so
                      // that nothing touches the loaded registers — and waits for them — before the MFMAs of this step)

  // Branch-free: every lane loads (pixel 0 when its tap falls outside the image) and the zero is selected afterwards.
  auto fetch = [&](int tap, int ky, int kx, int ch) {
    const int c0 = ch * EPC;
    const bool first = c0 < p.C1;
    const char* base = (first ? p.X1 : p.X2) + (size_t)((first ? c0 : c0 - p.C1) * ES) + piece * 16;
    const uint32_t ldb = (uint32_t)((first ? p.ld1 : p.ld2) * ES);
    okm = 0;
    if (first ? reg1 : reg2) {
      const int delta = sgn * (ky * (first ? W1 : p.Win) + kx);
#pragma unroll
      for (int j = 0; j < AP; ++j) {
        const bool ok = (vmask[j] >> tap) & 1u;
        const int pix = ok ? (first ? pc1[j] : pc2[j]) + delta : 0;
        okm |= ok ? (1u << j) : 0u;
        if (!(RDST_DBGV(p.dbg) & 1)) ra[j] = *reinterpret_cast<const u32x4_a4*>(base + (uint64_t)(uint32_t)pix * ldb);
      }
    } else {
      const int up = first ? p.up1 : 0;
      const int Ws = first ? W1 : p.Win;
      const int dy = sgn * ky, dx = sgn * kx;
#pragma unroll
      for (int j = 0; j < AP; ++j) {
        const bool ok = (vmask[j] >> tap) & 1u;
        const int iy = (cy[j] + dy) >> sh, ix = (cx[j] + dx) >> sh;
        const int pix = ok ? (first ? pb1[j] : pb2[j]) + (iy >> up) * Ws + (ix >> up) : 0;
        okm |= ok ? (1u << j) : 0u;
        if (!(RDST_DBGV(p.dbg) & 1)) ra[j] = *reinterpret_cast<const u32x4_a4*>(base + (uint64_t)(uint32_t)pix * ldb);
      }
    }
    const char* wb = p.Wp + ((size_t)tap * p.Npad * p.Cin + c0) * ES;
#pragma unroll
    for (int j = 0; j < BP; ++j)
      if (!(RDST_DBGV(p.dbg) & 2)) rb[j] = *reinterpret_cast<const u32x4_a4*>(wb + wof[j]);
  };
  // SPLIT: LDS rows are 64-byte groups [16 bf16 hi][16 bf16 lo]; an activation piece (4 floats) lands as 8 + 8 bytes, the
  // weights arrive ALREADY split in that layout (host-side pack) and are copied as they are.
  auto stash = [&](int buf) {
    if (RDST_DBGV(p.dbg) & 4) return;
    char* A = smem + buf * STAGE;
    char* Bt = A + BM * RS;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int row = r0 + j * RPP;
      if (BM % RPP == 0 || row < BM) {      // compile-time true for every instantiated tile: no exec-mask branch
        const bool ok = (okm >> j) & 1u;
        u32x4_a4 v;
        v.x = ok ? ra[j].x : 0u; v.y = ok ? ra[j].y : 0u; v.z = ok ? ra[j].z : 0u; v.w = ok ? ra[j].w : 0u;
        if (SPLIT) {
          const float f0 = __uint_as_float(v.x), f1 = __uint_as_float(v.y), f2 = __uint_as_float(v.z), f3 = __uint_as_float(v.w);
          u32x2_a4 hi, lo;
          hi.x = pack_bf16x2(f0, f1); hi.y = pack_bf16x2(f2, f3);
          lo.x = pack_bf16x2(f0 - bf16lo(hi.x), f1 - bf16hi(hi.x));
          lo.y = pack_bf16x2(f2 - bf16lo(hi.y), f3 - bf16hi(hi.y));
          char* g = A + row * RS + (piece >> 2) * 64 + (piece & 3) * 8;
          *reinterpret_cast<u32x2_a4*>(g) = hi;
          *reinterpret_cast<u32x2_a4*>(g + 32) = lo;
        } else {
          *reinterpret_cast<u32x4_a4*>(A + row * RS + piece * 16) = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int row = r0 + j * RPP;
      if (BN % RPP == 0 || row < BN) *reinterpret_cast<u32x4_a4*>(Bt + row * RS + piece * 16) = rb[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  int tap = 0, ky = 0, kx = 0, ch = 0;         // (tap, chunk) of the tile being FETCHED
  auto advance = [&]() {
    if (++ch == CH) {
      ch = 0; ++tap;
      if (++kx == p.k) { kx = 0; ++ky; }
    }
  };
  fetch(0, 0, 0, 0);
  advance();
  stash(0);
  __syncthreads();
  for (int it = 0; it < NIT; ++it) {
    const bool more = it + 1 < NIT;
    if (more) {
      fetch(tap, ky, kx, ch);
      advance();
    }
    const char* A = smem + (it & 1) * STAGE + (wm * TM * 32 + r) * RS + h * 16;
    const char* Bt = smem + (it & 1) * STAGE + BM * RS + (wn * TN * 32 + r) * RS + h * 16;
    if (SPLIT) {
#pragma unroll
      for (int kk = 0; kk < KB / 64; ++kk) {   // 16 elements per k-step: a 64-byte group, hi half then lo half
        Pack16 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = *reinterpret_cast<const Pack16*>(A + i * 32 * RS + kk * 64);
          al[i] = *reinterpret_cast<const Pack16*>(A + i * 32 * RS + kk * 64 + 32);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = *reinterpret_cast<const Pack16*>(Bt + j * 32 * RS + kk * 64);
          bl[j] = *reinterpret_cast<const Pack16*>(Bt + j * 32 * RS + kk * 64 + 32);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (RDST_DBGV(p.dbg) & 8) continue;
            Mma<bf16>::mma(acc[i][j], ah[i], bl[j]);
            Mma<bf16>::mma(acc[i][j], al[i], bh[j]);
            Mma<bf16>::mma(acc[i][j], ah[i], bh[j]);
          }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < KB / 32; ++kk) {
        Pack16 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const Pack16*>(A + i * 32 * RS + kk * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const Pack16*>(Bt + j * 32 * RS + kk * 32);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::mma(acc[i][j], a[i], b[j]);
      }
    }
    if (more) stash((it + 1) & 1);
    __syncthreads();
  }

  // epilogue: rows of D leave as 8/16-byte segments through a wave-private 4 KB bounce tile
  float* eps = reinterpret_cast<float*>(smem + wave * 4096);
  TileEpilogue e;
  e.R = p.add; e.ldr = p.ld_add;
  e.Xa = nullptr; e.ldxa = 0; e.act = 0;
  e.Y = p.Y; e.ldy = p.ld_y;
  e.Acc = nullptr; e.ldacc = 0;
  e.Yf32 = nullptr; e.ldf = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col0 = n0 + (wn * TN + j) * 32;
      if (col0 >= p.Cout) continue;
      float vals[16];
      const float bv = (p.bias && col0 + r < p.Cout) ? p.bias[col0 + r] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) vals[v] = acc[i][j][v] + bv;
      tile_store_rows<T>(eps, vals, lane, m0 + (wm * TM + i) * 32, p.P, col0, p.Cout, e);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 (forward and data gradient: 90 % of the UNet's FLOPs): the HALO form.  A workgroup owns a TH x TW patch of
// ONE image and, per channel chunk, stages the (TH + 2) x (TW + 2) halo of input pixels ONCE; the nine taps are row
// offsets of the fragment reads into that tile (compile-time: the tap loop is unrolled), so per tap only the BN x KB weight
// slab is staged (already in its LDS layout: a copy).  Against uconv_kernel — one BM x KB input slab per (tap, chunk),
// ~400 instructions per 24 MFMAs: instruction-issue bound at a third of the MFMA rate — the input staging drops 6.4x
// (180 instead of 9 x 128 rows per chunk) and a tap costs ~100 instructions.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int TH, int TW, int BN, int KB, int WM, int WN, bool SPLIT>
__global__ void __launch_bounds__(256) uconv_halo_kernel(const UConvP p) {
  static_assert(!SPLIT || (sizeof(T) == 4 && KB >= 64), "the split mode stages fp32 rows, 16 elements per k-step");
  constexpr int ES = (int)sizeof(T);
  constexpr int BM = TH * TW;
  constexpr int HH = TH + 2, HW = TW + 2, NH = HH * HW;      // halo pixels
  constexpr int RS = KB + 16;
  constexpr int PIECES = KB / 16;
  constexpr int APN = (NH * PIECES + 255) / 256;             // halo pieces per thread
  constexpr int RPP = 256 / PIECES;
  constexpr int BP = (BN + RPP - 1) / RPP;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int EPC = KB / ES;
  constexpr int ASZ = ((NH * RS + 255) / 256) * 256, BSZ = BN * RS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ah = smem;                      // halo tile (single buffer: rewritten once per channel chunk)
  char* const Bw = smem + ASZ;                // weight slabs, two buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.y * BN;
  const int ntx = (p.Wout + TW - 1) / TW, nty = (p.Hout + TH - 1) / TH;
  const int txi = blockIdx.x % ntx, tyi = (blockIdx.x / ntx) % nty, b = blockIdx.x / (ntx * nty);
  const int y0 = tyi * TH, x0 = txi * TW;
  const int H1 = p.Hin >> p.up1, W1 = p.Win >> p.up1;

  // halo pieces of this thread: source pixel indices (both sources) and validity, prepared once
  int hp1[APN], hp2[APN];
  uint32_t hoff[APN];                          // LDS byte offset of the piece (row * RS + position inside the row)
  uint32_t hok = 0, hin = 0;                   // bit a: inside the image / a real piece of the tile
#pragma unroll
  for (int a = 0; a < APN; ++a) {
    const int idx = tid + a * 256;
    const int hr = idx / PIECES, pc = idx - hr * PIECES;
    const int hy = hr / HW, hx = hr - hy * HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool in = idx < NH * PIECES;
    const bool ok = in && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    hin |= in ? (1u << a) : 0u;
    hok |= ok ? (1u << a) : 0u;
    hp1[a] = ok ? (b * H1 + (iy >> p.up1)) * W1 + (ix >> p.up1) : 0;
    hp2[a] = ok ? (b * p.Hin + iy) * p.Win + ix : 0;
    hoff[a] = (uint32_t)(hr * RS) | ((uint32_t)pc << 24);     // piece index in the top byte
  }
  uint32_t wof[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int row = tid / PIECES + j * RPP;
    wof[j] = (uint32_t)(((n0 + (row < BN ? row : 0)) * p.Cin) * ES + (tid % PIECES) * 16);
  }
  // fragment rows: lane r of M-subtile i is tile pixel m = (wm * TM + i) * 32 + r -> halo row (ty * HW + tx) of tap (0, 0)
  uint32_t arow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = (wm * TM + i) * 32 + r;
    arow[i] = (uint32_t)(((m / TW) * HW + (m % TW)) * RS + h * 16);
  }

  const int CH = p.Cin / EPC;
  u32x4_a4 ra[APN], rb[2][BP];   // weight slabs: two register sets, fetched TWO taps ahead of their use
  // BatchNorm + ReLU of the first source on the way in (p.bn1): every piece of a thread is the same channel quad of the
  // chunk (256 threads are a multiple of the pieces per row), so one (scale, shift) quad per chunk serves all its pieces
  u32x4_a4 bsc = {0u, 0u, 0u, 0u}, bsh = {0u, 0u, 0u, 0u};
  bool bnx = false;   // the chunk in `ra` goes through bsc / bsh (wave-uniform)
  auto fetchA = [&](int ch) {
    const int c0 = ch * EPC;
    const bool first = c0 < p.C1;
    const char* base = (first ? p.X1 : p.X2) + (size_t)((first ? c0 : c0 - p.C1) * ES);
    const uint32_t ldb = (uint32_t)((first ? p.ld1 : p.ld2) * ES);
#pragma unroll
    for (int a = 0; a < APN; ++a)
      ra[a] = *reinterpret_cast<const u32x4_a4*>(base + (uint64_t)(uint32_t)(first ? hp1[a] : hp2[a]) * ldb + (hoff[a] >> 24) * 16);
    if constexpr (ES == 4) {
      bnx = first && p.bn1 != nullptr;
      if (bnx) {
        const int c = c0 + (int)(hoff[0] >> 24) * 4;
        bsc = *reinterpret_cast<const u32x4_a4*>(p.bn1 + c);
        bsh = *reinterpret_cast<const u32x4_a4*>(p.bn1 + p.C1 + c);
      }
    }
  };
  auto stashA = [&]() {
#pragma unroll
    for (int a = 0; a < APN; ++a) {
      if (!((hin >> a) & 1u)) continue;
      const bool ok = (hok >> a) & 1u;
      u32x4_a4 v = ra[a];
      if constexpr (ES == 4) {
        if (bnx) {   // relu(scale x + shift), the very fma of bn_apply_kernel: the same bits as the materialised activation
          { const float t = fmaf(__uint_as_float(bsc.x), __uint_as_float(v.x), __uint_as_float(bsh.x)); v.x = __float_as_uint(t > 0.f ? t : 0.f); }
          { const float t = fmaf(__uint_as_float(bsc.y), __uint_as_float(v.y), __uint_as_float(bsh.y)); v.y = __float_as_uint(t > 0.f ? t : 0.f); }
          { const float t = fmaf(__uint_as_float(bsc.z), __uint_as_float(v.z), __uint_as_float(bsh.z)); v.z = __float_as_uint(t > 0.f ? t : 0.f); }
          { const float t = fmaf(__uint_as_float(bsc.w), __uint_as_float(v.w), __uint_as_float(bsh.w)); v.w = __float_as_uint(t > 0.f ? t : 0.f); }
        }
      }
      v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;   // (zero padding AFTER the activation)
      const uint32_t pc = hoff[a] >> 24;
      char* rowp = Ah + (hoff[a] & 0xffffffu);
      if (SPLIT) {
        const float f0 = __uint_as_float(v.x), f1 = __uint_as_float(v.y), f2 = __uint_as_float(v.z), f3 = __uint_as_float(v.w);
        u32x2_a4 hi, lo;
        hi.x = pack_bf16x2(f0, f1); hi.y = pack_bf16x2(f2, f3);
        lo.x = pack_bf16x2(f0 - bf16lo(hi.x), f1 - bf16hi(hi.x));
        lo.y = pack_bf16x2(f2 - bf16lo(hi.y), f3 - bf16hi(hi.y));
        char* g = rowp + (pc >> 2) * 64 + (pc & 3) * 8;
        *reinterpret_cast<u32x2_a4*>(g) = hi;
        *reinterpret_cast<u32x2_a4*>(g + 32) = lo;
      } else {
        *reinterpret_cast<u32x4_a4*>(rowp + pc * 16) = v;
      }
    }
  };
  auto fetchB = [&](auto slotc, int tap, int ch) {
    constexpr int SL = decltype(slotc)::value;
    const char* wb = p.Wp + ((size_t)tap * p.Npad * p.Cin + (size_t)ch * EPC) * ES;
#pragma unroll
    for (int j = 0; j < BP; ++j) rb[SL][j] = *reinterpret_cast<const u32x4_a4*>(wb + wof[j]);
  };
  auto stashB = [&](auto slotc, int buf) {
    constexpr int SL = decltype(slotc)::value;
    char* Bt = Bw + buf * BSZ;
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int row = tid / PIECES + j * RPP;
      if (BN % RPP == 0 || row < BN) *reinterpret_cast<u32x4_a4*>(Bt + row * RS + (tid % PIECES) * 16) = rb[SL][j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // slab s (= tap-step index ch * 9 + tap) waits in register set s & 1 and is multiplied from LDS buffer s & 1; the loop is
  // unrolled over 18 steps (two chunks) so that both parities are compile-time constants
  auto slab_tap = [&](int sidx, int& t, int& c) { c = sidx / 9; t = sidx - 9 * c; };
  const int NS = 9 * CH;
  fetchA(0);
  fetchB(S0{}, 0, 0);
  stashA();
  stashB(S0{}, 0);
  if (NS > 1) {
    int t, c;
    slab_tap(1, t, c);
    fetchB(S1{}, t, c);
  }
  __syncthreads();
  auto step = [&](int sidx, int ch, auto tapc, auto parc) {
    constexpr int tap = decltype(tapc)::value;
    constexpr int PAR = decltype(parc)::value;         // sidx & 1
    const bool lastc = ch + 1 == CH;
    if (sidx + 2 < NS) {                                // slab sidx + 2 -> the register set slab sidx has left
      int t, c;
      slab_tap(sidx + 2, t, c);
      fetchB(parc, t, c);
    }
    if (tap == 6 && !lastc) fetchA(ch + 1);             // lands under the last three taps
    const int bsel = PAR;
      // the input pixel of tap (ky, kx): forward out + (ky, kx) - 1, data gradient out + 1 - (ky, kx)
      const int ky = tap / 3, kx = tap % 3;
      const int toff = p.transposed ? ((2 - ky) * HW + (2 - kx)) * RS : (ky * HW + kx) * RS;
      const char* Bt = Bw + bsel * BSZ + (wn * TN * 32 + r) * RS + h * 16;
      if (SPLIT) {
#pragma unroll
        for (int kk = 0; kk < KB / 64; ++kk) {
          Pack16 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            ah[i] = *reinterpret_cast<const Pack16*>(Ah + arow[i] + toff + kk * 64);
            al[i] = *reinterpret_cast<const Pack16*>(Ah + arow[i] + toff + kk * 64 + 32);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const Pack16*>(Bt + j * 32 * RS + kk * 64);
            bl[j] = *reinterpret_cast<const Pack16*>(Bt + j * 32 * RS + kk * 64 + 32);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              Mma<bf16>::mma(acc[i][j], ah[i], bl[j]);
              Mma<bf16>::mma(acc[i][j], al[i], bh[j]);
              Mma<bf16>::mma(acc[i][j], ah[i], bh[j]);
            }
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < KB / 32; ++kk) {
          Pack16 a[TM], bq[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const Pack16*>(Ah + arow[i] + toff + kk * 32);
#pragma unroll
          for (int j = 0; j < TN; ++j) bq[j] = *reinterpret_cast<const Pack16*>(Bt + j * 32 * RS + kk * 32);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) Mma<T>::mma(acc[i][j], a[i], bq[j]);
        }
      }
    if (sidx + 1 < NS) stashB(std::integral_constant<int, PAR ^ 1>{}, PAR ^ 1);
    if (tap == 8 && !lastc) {
      __syncthreads();                                  // every wave is done with this chunk's halo
      stashA();
    }
    __syncthreads();
  };
  for (int ch = 0; ch < CH; ch += 2) {
#define RDST_UC_STEP(TAP, PAR) step(ch * 9 + TAP, ch, std::integral_constant<int, TAP>{}, std::integral_constant<int, PAR>{});
    RDST_UC_STEP(0, 0) RDST_UC_STEP(1, 1) RDST_UC_STEP(2, 0) RDST_UC_STEP(3, 1) RDST_UC_STEP(4, 0) RDST_UC_STEP(5, 1) RDST_UC_STEP(6, 0)
    RDST_UC_STEP(7, 1) RDST_UC_STEP(8, 0)
#undef RDST_UC_STEP
    if (ch + 1 < CH) {
#define RDST_UC_STEP(TAP, PAR) step((ch + 1) * 9 + TAP, ch + 1, std::integral_constant<int, TAP>{}, std::integral_constant<int, PAR>{});
      RDST_UC_STEP(0, 1) RDST_UC_STEP(1, 0) RDST_UC_STEP(2, 1) RDST_UC_STEP(3, 0) RDST_UC_STEP(4, 1) RDST_UC_STEP(5, 0) RDST_UC_STEP(6, 1)
      RDST_UC_STEP(7, 0) RDST_UC_STEP(8, 1)
#undef RDST_UC_STEP
    }
  }

  // epilogue: each 32 x 32 accumulator through a wave-private bounce tile, rows leave as 8/16-byte segments; a tile row is
  // a pixel of the patch (not a consecutive pixel index): rows outside the image are dropped
  float* eps = reinterpret_cast<float*>(smem + wave * 4096);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col0 = n0 + (wn * TN + j) * 32;
      if (col0 >= p.Cout) continue;
      const float bv = (p.bias && col0 + r < p.Cout) ? p.bias[col0 + r] : 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) eps[acc_row(v, h) * 32 + r] = acc[i][j][v] + bv;
      __builtin_amdgcn_wave_barrier();
      const int chunk = lane & 7, col = col0 + chunk * 4;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = (lane >> 3) + 8 * pass;
        const int m = (wm * TM + i) * 32 + row;
        const int oy = y0 + m / TW, ox = x0 + m % TW;
        const float4 f4 = *reinterpret_cast<const float4*>(eps + row * 32 + chunk * 4);
        if (oy < p.Hout && ox < p.Wout && col < p.Cout) {
          const int64_t rr = ((int64_t)b * p.Hout + oy) * p.Wout + ox;
          float f[4] = {f4.x, f4.y, f4.z, f4.w};
          T* dst = reinterpret_cast<T*>(p.Y) + rr * p.ld_y + col;
          const bool full = col + 4 <= p.Cout;
          if (p.add) {
            const T* ap = reinterpret_cast<const T*>(p.add) + rr * p.ld_add + col;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (col + q < p.Cout) f[q] += to_f32<T>(ap[q]);
          }
          if (full && (reinterpret_cast<uintptr_t>(dst) & (sizeof(T) == 2 ? 7 : 15)) == 0) {
            if (sizeof(T) == 2) {
              u32x2_a4 u;
              u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
              *reinterpret_cast<u32x2_a4*>(dst) = u;
            } else {
              u32x4_a4 u;
              u.x = __float_as_uint(f[0]); u.y = __float_as_uint(f[1]); u.z = __float_as_uint(f[2]); u.w = __float_as_uint(f[3]);
              *reinterpret_cast<u32x4_a4*>(dst) = u;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (col + q < p.Cout) dst[q] = from_f32<T>(f[q]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  if (p.stats) {   // (no bias, no addend: checked by the launcher) the accumulators ARE the outputs
    float s0[TN], s1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = (wm * TM + i) * 32 + acc_row(v, h);
        const bool ok = y0 + m / TW < p.Hout && x0 + m % TW < p.Wout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float a = ok ? acc[i][j][v] : 0.f;
          s0[j] += a;
          s1[j] = fmaf(a, a, s1[j]);
        }
      }
#pragma unroll
    for (int j = 0; j < TN; ++j) {   // lanes r and r + 32 hold the same channel
      s0[j] += __shfl_xor(s0[j], 32, 64);
      s1[j] += __shfl_xor(s1[j], 32, 64);
    }
    __syncthreads();   // the bounce tiles are done with
    float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]
    if (h == 0)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        red[((wm * BN) + (wn * TN + j) * 32 + r) * 2] = s0[j];
        red[((wm * BN) + (wn * TN + j) * 32 + r) * 2 + 1] = s1[j];
      }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t0 += red[(w * BN + tid) * 2]; t1 += red[(w * BN + tid) * 2 + 1]; }
      p.stats[((int64_t)blockIdx.x * 2) * p.Cout + n0 + tid] = t0;
      p.stats[((int64_t)blockIdx.x * 2 + 1) * p.Cout + n0 + tid] = t1;
    }
  }
}

template <typename T, int TH, int TW, int BN, int KB, int WM, int WN, bool SPLIT>
int launch_halo(const UConvP& p, hipStream_t st) {
  constexpr int RS = KB + 16, NH = (TH + 2) * (TW + 2);
  constexpr size_t ASZ = ((NH * RS + 255) / 256) * 256;
  size_t lds = ASZ + (size_t)2 * BN * RS;
  if (lds < 16384) lds = 16384;
  auto kern = uconv_halo_kernel<T, TH, TW, BN, KB, WM, WN, SPLIT>;
  const int ntx = (p.Wout + TW - 1) / TW, nty = (p.Hout + TH - 1) / TH;
  const dim3 grid((unsigned)((int64_t)p.B * ntx * nty), (unsigned)(p.Npad / BN));
  if (p.stats_blocks) *p.stats_blocks = (int)grid.x;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  return rdst_launch_status("rdst_u_conv");
}

template <typename T, int KB, bool SPLIT>
int pick_halo(const UConvP& p, hipStream_t st) {
  const int64_t big = (int64_t)p.B * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);     // 8 x 16 patches
  if (p.Npad % 128 == 0) {
    if (p.Wout >= 16 && big * (p.Npad / 128) >= 256) return launch_halo<T, 8, 16, 128, KB, 2, 2, SPLIT>(p, st);
    const int64_t small = (int64_t)p.B * ((p.Hout + 7) / 8) * ((p.Wout + 7) / 8);
    if (small * (p.Npad / 128) >= 256) return launch_halo<T, 8, 8, 128, KB, 2, 2, SPLIT>(p, st);
    return launch_halo<T, 8, 8, 64, KB, 2, 2, SPLIT>(p, st);
  }
  if (p.Npad % 64 == 0) {
    if (p.Wout >= 16 && big * (p.Npad / 64) >= 256) return launch_halo<T, 8, 16, 64, KB, 2, 2, SPLIT>(p, st);
    return launch_halo<T, 8, 8, 64, KB, 2, 2, SPLIT>(p, st);
  }
  return launch_halo<T, 8, 16, 32, KB, 4, 1, SPLIT>(p, st);
}

template <typename T, int BM, int BN, int KB, int WM, int WN, bool SPLIT>
int launch(const UConvP& p, hipStream_t st) {
  constexpr int RS = KB + 16;
  size_t lds = (size_t)2 * (BM + BN) * RS;
  if (lds < 16384) lds = 16384;
  const dim3 grid((unsigned)((p.P + BM - 1) / BM), (unsigned)(p.Npad / BN));
  hipLaunchKernelGGL((uconv_kernel<T, BM, BN, KB, WM, WN, SPLIT>), grid, dim3(256), lds, st, p);
  return rdst_launch_status("rdst_u_conv");
}

template <typename T, int KB, bool SPLIT>
int pick_bn(const UConvP& p, hipStream_t st) {
  // the widest tile that still gives every CU a workgroup: the deep layers of the UNet have few pixels (2048 at 8x8 x 32)
  // and long reductions (K = 4608): at 128 x 128 they are 64 workgroups of 144 steps on 256 CUs
  const int64_t mb128 = (p.P + 127) / 128;
  if (p.Npad % 128 == 0) {
    if (mb128 * (p.Npad / 128) >= 256) return launch<T, 128, 128, KB, 2, 2, SPLIT>(p, st);
    if (2 * mb128 * (p.Npad / 128) >= 256) return launch<T, 64, 128, KB, 2, 2, SPLIT>(p, st);
    return launch<T, 64, 64, KB, 2, 2, SPLIT>(p, st);
  }
  if (p.Npad % 64 == 0) {
    if (mb128 * (p.Npad / 64) >= 256) return launch<T, 128, 64, KB, 2, 2, SPLIT>(p, st);
    return launch<T, 64, 64, KB, 2, 2, SPLIT>(p, st);
  }
  return launch<T, 128, 32, KB, 4, 1, SPLIT>(p, st);
}

template <typename T, bool SPLIT>
int pick_kb(const UConvP& p, hipStream_t st) {
  const int es = (int)sizeof(T);
  auto fits = [&](int kb) { return (p.Cin * es) % kb == 0 && (p.C1 * es) % kb == 0; };
  const bool halo = p.k == 3 && p.stride == 1 && !(RDST_DBGV(p.dbg) & 16);
  if (halo) {
    if (fits(128)) return pick_halo<T, 128, SPLIT>(p, st);
    if (fits(64)) return pick_halo<T, 64, SPLIT>(p, st);
  }
  if (p.bn1 || p.stats) return rdst_fail(RDST_ENOTSUP, "rdst_u_conv: bn1 / stats are taken by the halo kernel only");
  if (fits(128)) return pick_bn<T, 128, SPLIT>(p, st);
  if (fits(64)) return pick_bn<T, 64, SPLIT>(p, st);
  if constexpr (!SPLIT)
    if (fits(32)) return pick_bn<T, 32, false>(p, st);
  return rdst_fail(RDST_ENOTSUP, "rdst_u_conv: (C1 + C2) * elementsize = %d must be a multiple of %d bytes (C1 = %d)", p.Cin * es,
                   SPLIT ? 64 : 32, p.C1);
}

}  // namespace

extern "C" int rdst_u_conv(const void* X1, int64_t ld1, int C1, int up1, const void* X2, int64_t ld2, int C2, const void* Wp,
                           const float* bias, const void* add, int64_t ld_add, void* Y, int64_t ld_y, int B, int Hin, int Win,
                           int Hout, int Wout, int Cout, int Npad, int ksize, int stride, int transposed, int dtype, void* stream,
                           const float* bn1, float* stats, int* stats_blocks) {
  if (!X1 || !Wp || !Y) return rdst_fail(RDST_EINVAL, "rdst_u_conv: null pointer");
  if (stats && (!stats_blocks || bias || add || ksize != 3 || stride != 1))
    return rdst_fail(RDST_ENOTSUP, "rdst_u_conv: stats needs the 3x3 stride-1 form without bias / addend (and stats_blocks)");
  if (bn1 && (dtype == RDST_BF16 || ksize != 3 || stride != 1 || transposed || (C1 * 4) % 64 || ((C1 + C2) * 4) % 64 || ((uintptr_t)bn1 & 3)))
    return rdst_fail(RDST_ENOTSUP, "rdst_u_conv: bn1 (BatchNorm + ReLU on the way in) needs the fp32 / fp32x3 3x3 stride-1 forward form");
  if (dtype != RDST_F32 && dtype != RDST_BF16 && dtype != RDST_F32X3) return rdst_fail(RDST_EINVAL, "rdst_u_conv: bad dtype %d", dtype);
  if (B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0)
    return rdst_fail(RDST_EINVAL, "rdst_u_conv: bad shape");
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return rdst_fail(RDST_ENOTSUP, "rdst_u_conv: k = %d stride = %d", ksize, stride);
  if (Npad < Cout || Npad % 32) return rdst_fail(RDST_EINVAL, "rdst_u_conv: Npad = %d (Cout = %d rounded up to 32)", Npad, Cout);
  if (C2 > 0 && !X2) return rdst_fail(RDST_EINVAL, "rdst_u_conv: C2 > 0 without a second source");
  if (up1 && ((Hin | Win) & 1)) return rdst_fail(RDST_EINVAL, "rdst_u_conv: upsampled source needs even Hin, Win");
  if (ld1 < C1 || (C2 > 0 && ld2 < C2) || ld_y < Cout || (add && ld_add < Cout)) return rdst_fail(RDST_EINVAL, "rdst_u_conv: leading dimension too small");
  const int es = dtype == RDST_BF16 ? 2 : 4;
  if (((uintptr_t)X1 | (uintptr_t)X2 | (uintptr_t)Wp) & 3 || (ld1 * es) % 4 || (ld2 * es) % 4)
    return rdst_fail(RDST_EINVAL, "rdst_u_conv: sources must be dword aligned");
  if ((int64_t)B * Hin * Win >= (1ll << 31) || (int64_t)B * Hin * Win * (ld1 > ld2 ? ld1 : ld2) * es >= (1ll << 40))
    return rdst_fail(RDST_ENOTSUP, "rdst_u_conv: tensor too large for 32-bit pixel indices");
  UConvP p;
  p.X1 = (const char*)X1; p.ld1 = ld1; p.C1 = C1; p.up1 = up1 ? 1 : 0;
  p.X2 = (const char*)X2; p.ld2 = ld2; p.C2 = C2;
  p.Wp = (const char*)Wp; p.bias = bias; p.add = add; p.ld_add = ld_add; p.Y = Y; p.ld_y = ld_y;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.Cin = C1 + C2; p.Cout = Cout; p.Npad = Npad;
  p.k = ksize; p.stride = stride; p.transposed = transposed ? 1 : 0;
  p.P = (int64_t)B * Hout * Wout;
  p.bn1 = bn1;
  p.stats = stats; p.stats_blocks = stats ? stats_blocks : nullptr;
  {
    const char* e = rdst_dbg_getenv("RDST_UCONV_DBG");
    p.dbg = e ? atoi(e) : 0;
  }
  if (dtype == RDST_F32X3) return pick_kb<float, true>(p, (hipStream_t)stream);
  return dtype == RDST_F32 ? pick_kb<float, false>(p, (hipStream_t)stream) : pick_kb<bf16, false>(p, (hipStream_t)stream);
}
