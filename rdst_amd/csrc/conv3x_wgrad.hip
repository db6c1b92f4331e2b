// K4/K5 weight gradient for the E1 shapes in the RDST_F32X3 arithmetic: dW[co][ci][ky][kx] = s * sum_p dY[p][co] * X[p + (ky,kx) - 1][ci],
// dbias[co] = s * sum_p dY[p][co], on fp32 ROWS with both operands as two bf16 terms (hi.hi + hi.lo + lo.hi: three
// v_mfma_f32_32x32x16_bf16 per product) and the WHOLE dW ACCUMULATOR OF A WORKGROUP IN REGISTERS — conv3_wgrad.hip (bf16 rows) rebuilt
// for 4-byte pixels; the round-5 split mode ran the stripe kernel of conv_mfma.hip (three workgroups per pixel range, X and dY read
// three times: 254 us per 150 -> 60 launch where the bf16 kernel takes 38):
//   * a workgroup (8 waves) owns a set of 32-pixel-wide strips and ALL 9 taps; wave w keeps its 3-12 output tiles (32 output channels
//     x 32 input channels of one tap) as accumulators from the first pixel to the last;
//   * the RAW fp32 rows of the next step (one X row with its halo, one dY row) arrive by LDS-DMA in a staging buffer while the
//     current output row is multiplied; behind the step's barrier ONE conversion pass, shared by the 8 waves, splits them into hi / lo
//     bf16 PLANES ([pixel][channel], pixel strides = 64 or 192 (mod 256) bytes) of a ring of 4 X rows / 2 dY rows: both operands
//     are ds_read_b64_tr_b16 of either plane (the contraction index — the pixel — is their row), the three kx taps pixel offsets
//     of the SAME X image, the ky taps ring rows;
//   * d(bias) is one more tile whose B operand is a register of ones (hi and lo of dY);
//   * the gradient of conv + PixelShuffle(2) (60 -> 240) runs as FOUR launches of the 60 -> 60 form, one per sub-pixel q = 2 i + j: dY
//     seen through a stride-2 pixel view of the shuffled tensor, output channels 4 c' + q;
//   * fp32 slabs, summed in fixed order by conv3x_wgrad_reduce (deterministic), scale applied, scattered into (Cout, Cin, 3, 3).
#include "conv.h"
#include "mfma.h"

namespace {

constexpr int lds_tr_stride(int bytes) {   // >= bytes, = 64 or 192 (mod 256)
  int s = (bytes + 63) / 64 * 64;
  while ((s % 256) != 64 && (s % 256) != 192) s += 64;
  return s;
}

struct W3XArgs {
  const float* X; int64_t ldx; int x_bytes;
  const float* dY; int64_t lddy; int dy_bytes;
  int ymul, yoff, xmul, xoff;                   // dY pixel (y, x) of the conv's grid = memory pixel (y ymul + yoff, x xmul + xoff)
  float* slab;                                  // [grid][SLABF] accumulator dumps
  int B, H, W;
  int SH, nys, nstrips, npg;                    // strips; npg = pixel groups (= grid)
};

template <int CI, int CO>
struct W3X {
  static constexpr int CT = (CO + 31) / 32;                      // channel tiles (2)
  static constexpr int WPC = 8 / CT;                             // waves per channel tile
  static constexpr int CIT = (CI + 31) / 32;
  static constexpr int TPC = 9 * CIT + 1;                        // tiles per channel tile, + the bias tile
  static constexpr int TPW = (TPC + WPC - 1) / WPC;              // tiles per wave
  static constexpr int XS = lds_tr_stride(2 * CI), YS = lds_tr_stride(2 * CO);
  static constexpr int XPL = 34 * XS, YPL = 32 * YS;             // bytes of one plane of one ring row
  static constexpr int NRX = 4, NRY = 2;
  static constexpr int Y_OFF = NRX * 2 * XPL;
  static constexpr int RAW_OFF = (Y_OFF + NRY * 2 * YPL + 1023) / 1024 * 1024;
  static constexpr int RXSL = (4 * CI + 15) / 16, RYSL = (4 * CO + 15) / 16;   // 16-byte slots of a raw pixel
  static constexpr int RXP = (34 * RXSL + 63) / 64, RYP = (32 * RYSL + 63) / 64;
  static constexpr int RAWY = RAW_OFF + RXP * 1024;
  static constexpr int SMEM = RAWY + RYP * 1024;
  static constexpr int SLABF = 8 * TPW * 1024;                   // floats per workgroup slab
  static_assert(8 % CT == 0, "waves per channel tile");
  static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int CI, int CO>
__global__ void __launch_bounds__(512, 2) conv3x_wgrad_kernel(const W3XArgs p) {
  using CF = W3X<CI, CO>;
  constexpr int TPW = CF::TPW, CIT = CF::CIT, XS = CF::XS, YS = CF::YS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int pg = blockIdx.x;
  const int ct = wave / CF::WPC, wq = wave % CF::WPC;      // channel tile, chunk of its tiles

  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void* ptr, int bytes) {
    u32x4s_t r;
    r.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    r.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
    r.w = 0x00020000u;
    return r;
  };
  const u32x4s_t rx = make_rsrc(p.X, p.x_bytes), ry = make_rsrc(p.dY, p.dy_bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const u32x4s_t& rs, uint32_t ldst, int off) {   // inline asm: see conv3_mfma.hip
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rs) : "memory");
  };
  lds_zero16(smem, CF::RAW_OFF, tid, 512);   // (the planes' pad columns / pixels)
  __syncthreads();

  // ---- the wave's tiles: tile j is n = wq * TPW + j of its channel tile: n < 9 CIT -> (tap, ci tile), n == 9 CIT -> bias
  int boff[TPW];          // lane offset of the B fragment inside an X plane row (without the k-step), per tile
  int bky[TPW];           // its kernel row (wave-uniform)
  const int gq = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int lane_rc = (8 * h + q4) * XS + 32 * (gq & 1) + 8 * pp;   // transposed-read lane part: row 8h + q, 4 columns at 16 (gq&1) + 4 pp
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int n = wq * TPW + j;
    const int tap = n / CIT, cit = n - tap * CIT;
    const int ky = tap / 3, kx = tap - ky * 3;
    bky[j] = ky;
    boff[j] = lane_rc + kx * XS + cit * 64;
  }
  const int aoff = (8 * h + q4) * YS + ct * 64 + 32 * (gq & 1) + 8 * pp;   // A fragment lane offset inside a dY plane row
  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const bf16x8_t ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};

  const int nxs = W / 32;
  for (int strip = pg; strip < p.nstrips; strip += p.npg) {
    const int xs = strip % nxs, tq = strip / nxs;
    const int ysg = tq % p.nys, b = tq / p.nys;
    const int y0 = ysg * p.SH, x0 = xs * 32;
    const int nrows = (H - y0 < p.SH) ? H - y0 : p.SH;

    // X ring: rel row q = input row y0 - 1 + q, slot q % NRX;  dY ring: row q = output row y0 + q, slot q % NRY.
    // Raw rows: [pixel][16-byte slots of its channels], whole 1 KB pieces dealt round-robin to the 8 waves.
    auto load_raw = [&](int xrel, int yrel) {   // xrel / yrel < 0: that operand is not loaded
      const int npx = xrel >= 0 ? CF::RXP : 0, tot = npx + (yrel >= 0 ? CF::RYP : 0);
      for (int q = wave; q < tot; q += 8) {
        if (q < npx) {
          const int sidx = q * 64 + lane, px = sidx / CF::RXSL, sl = sidx - px * CF::RXSL;
          const int x = x0 - 1 + px, y = y0 - 1 + xrel;
          const bool ok = x >= 0 && x < W && px < 34 && y >= 0 && y < H;
          dma(rx, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::RAW_OFF + q * 1024)),
              ok ? (int)((((int64_t)b * H + y) * W + x) * (p.ldx * 4)) + sl * 16 : p.x_bytes);
        } else {
          const int q2 = q - npx;
          const int sidx = q2 * 64 + lane, px = sidx / CF::RYSL, sl = sidx - px * CF::RYSL;
          const int y = y0 + yrel;
          const bool ok = px < 32 && yrel < nrows;
          const int64_t mpix = ((int64_t)b * (H * p.ymul) + (int64_t)y * p.ymul + p.yoff) * ((int64_t)W * p.xmul) + (int64_t)(x0 + px) * p.xmul + p.xoff;
          dma(ry, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::RAWY + q2 * 1024)), ok ? (int)(mpix * (p.lddy * 4)) + sl * 16 : p.dy_bytes);
        }
      }
    };
    // the conversion pass: raw chunk of 4 floats -> hi / lo bf16 pairs into the planes of ring slots (xrel % NRX, yrel % NRY)
    auto convert = [&](int xrel, int yrel) {
      constexpr int CKX = (CI + 3) / 4, CKY = (CO + 3) / 4;
      const int nxc = xrel >= 0 ? 34 * CKX : 0, tot = nxc + (yrel >= 0 ? 32 * CKY : 0);
      for (int idx = tid; idx < tot; idx += 512) {
        const bool isx = idx < nxc;
        const int li = isx ? idx : idx - nxc;
        const int per = isx ? CKX : CKY, width = isx ? CI : CO;
        const int px = li / per, chk = li - px * per;
        int c0 = chk * 4;
        c0 = c0 + 4 <= width ? c0 : width - 4;   // (a ragged last chunk overlaps its neighbour and rewrites the same values)
        const char* src = smem + (isx ? CF::RAW_OFF + px * (CF::RXSL * 16) : CF::RAWY + px * (CF::RYSL * 16)) + c0 * 4;
        float f[4];
        if ((c0 & 3) == 0) {
          const float4 v = *reinterpret_cast<const float4*>(src);
          f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
          const float2 v0 = *reinterpret_cast<const float2*>(src), v1 = *reinterpret_cast<const float2*>(src + 8);
          f[0] = v0.x; f[1] = v0.y; f[2] = v1.x; f[3] = v1.y;
        }
        uint32_t h0 = pack_bf16x2(f[0], f[1]), h1 = pack_bf16x2(f[2], f[3]);
        uint32_t l0 = pack_bf16x2(f[0] - bf16lo(h0), f[1] - bf16hi(h0)), l1 = pack_bf16x2(f[2] - bf16lo(h1), f[3] - bf16hi(h1));
        char* dst = smem + (isx ? (xrel % CF::NRX) * 2 * CF::XPL + px * XS : CF::Y_OFF + (yrel % CF::NRY) * 2 * CF::YPL + px * YS) + c0 * 2;
        uint32_t* dh = reinterpret_cast<uint32_t*>(dst);
        uint32_t* dl = reinterpret_cast<uint32_t*>(dst + (isx ? CF::XPL : CF::YPL));
        dh[0] = h0; dh[1] = h1; dl[0] = l0; dl[1] = l1;
      }
    };
    // strip start: X rows rel 0 .. 2 and dY row 0 through the one staging buffer
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
      load_raw(q, q == 2 ? 0 : -1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      convert(q, q == 2 ? 0 : -1);
      __syncthreads();
    }
    for (int j = 0; j < nrows; ++j) {
      const bool more = j + 1 < nrows;
      if (more) load_raw(j + 3, j + 1);   // (the staging buffer: converted before this step)
      {
        const int yo = j;
        const char* yrow = smem + CF::Y_OFF + (yo % CF::NRY) * 2 * CF::YPL + aoff;
        const char* xrow[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) xrow[ky] = smem + ((yo + ky) % CF::NRX) * 2 * CF::XPL;
        typedef __attribute__((address_space(3))) s16x4_t* lds_p;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Pack16 ah, al;
          {
            const char* a = yrow + ks * 16 * YS;
            const uint2 u0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a));
            const uint2 u1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * YS)));
            const uint2 u2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + CF::YPL)));
            const uint2 u3 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + CF::YPL + 4 * YS)));
            ah.w[0] = u0.x; ah.w[1] = u0.y; ah.w[2] = u1.x; ah.w[3] = u1.y;
            al.w[0] = u2.x; al.w[1] = u2.y; al.w[2] = u3.x; al.w[3] = u3.y;
          }
          const bf16x8_t avh = __builtin_bit_cast(bf16x8_t, ah), avl = __builtin_bit_cast(bf16x8_t, al);
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const int n = wq * TPW + t;                      // wave-uniform
            if (n < 9 * CIT) {
              const char* xr = (bky[t] == 0 ? xrow[0] : (bky[t] == 1 ? xrow[1] : xrow[2])) + boff[t] + ks * 16 * XS;
              const uint2 b0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)xr));
              const uint2 b1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(xr + 4 * XS)));
              const uint2 b2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(xr + CF::XPL)));
              const uint2 b3 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(xr + CF::XPL + 4 * XS)));
              Pack16 bh, bl;
              bh.w[0] = b0.x; bh.w[1] = b0.y; bh.w[2] = b1.x; bh.w[3] = b1.y;
              bl.w[0] = b2.x; bl.w[1] = b2.y; bl.w[2] = b3.x; bl.w[3] = b3.y;
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avl, __builtin_bit_cast(bf16x8_t, bh), acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avh, __builtin_bit_cast(bf16x8_t, bl), acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avh, __builtin_bit_cast(bf16x8_t, bh), acc[t], 0, 0, 0);
            } else if (n == 9 * CIT) {
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avl, ones, acc[t], 0, 0, 0);   // every column = sum over pixels
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(avh, ones, acc[t], 0, 0, 0);
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // the next step's raw rows are published; this step's plane reads are done
      if (more) {
        convert(j + 3, j + 1);
        __syncthreads();
      }
    }
  }
  // ---- dump the accumulators: slab[wg][wave][tile][reg][lane], fp32 ----
  float* my = p.slab + (size_t)blockIdx.x * CF::SLABF + (size_t)wave * TPW * 1024;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) my[(t * 16 + v) * 64 + lane] = acc[t][v];
}

// dW[co][ci][tap] = s * sum_g slab[g][...], dbias[co] likewise (column 0 of the bias tile); one thread per accumulator element;
// output channel co of the launch is channel cmul * co + coff of the convolution (the sub-pixel launches of conv + PixelShuffle)
template <int CI, int CO>
__global__ void __launch_bounds__(256) conv3x_wgrad_reduce_kernel(const float* __restrict__ slab, int npg, float s, int cin_w, int cmul,
                                                                  int coff, float* __restrict__ dW, float* __restrict__ dbias) {
  using CF = W3X<CI, CO>;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= CF::SLABF) return;
  const int lane = e & 63, v = (e >> 6) & 15, t = (e >> 10) % CF::TPW, wave = (e >> 10) / CF::TPW;
  const int ct = wave / CF::WPC, wq = wave % CF::WPC;
  const int n = wq * CF::TPW + t;
  if (n > 9 * CF::CIT) return;
  const int col = lane & 31;
  const bool is_b = n == 9 * CF::CIT;
  const int tap = n / CF::CIT, ci = (n - tap * CF::CIT) * 32 + col;
  if (is_b ? col != 0 : ci >= CI) return;
  const int m = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);          // row of the accumulator tile
  const int co = ct * 32 + m;
  if (co >= CO) return;
  float a = 0.f;
  const float* src = slab + e;
  for (int g0 = 0; g0 < npg; g0 += 16) {   // 16 loads in flight, summed in the same fixed order
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = (g0 + u < npg) ? src[(size_t)(g0 + u) * CF::SLABF] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) a += x[u];
  }
  const int cout = cmul * co + coff;
  if (is_b) {
    if (dbias) dbias[cout] = a * s;
  } else if (dW) {
    dW[((int64_t)cout * cin_w + ci) * 9 + tap] = a * s;
  }
}

template <int CI, int CO>
int launch_w3x(W3XArgs& p, float s, int cmul, int coff, float* dW, float* dbias, hipStream_t st, const char* what) {
  using CF = W3X<CI, CO>;
  int SH = 16;
  auto count = [&](int sh) { return (int64_t)p.B * ((p.H + sh - 1) / sh) * (p.W / 32); };
  while (count(SH) < 256 && SH > 2) SH /= 2;
  p.SH = SH;
  p.nys = (p.H + SH - 1) / SH;
  const int64_t ns = count(SH);
  if (ns >= (1ll << 31)) return RDST_ENOTSUP;
  p.nstrips = (int)ns;
  p.npg = ns < 256 ? (int)ns : 256;
  auto kern = conv3x_wgrad_kernel<CI, CO>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)p.npg), dim3(512), CF::SMEM, st, p);
  if (int rc = rdst_launch_status(what)) return rc;
  hipLaunchKernelGGL((conv3x_wgrad_reduce_kernel<CI, CO>), dim3((unsigned)((CF::SLABF + 255) / 256)), dim3(256), 0, st, p.slab, p.npg, s,
                     CI, cmul, coff, dW, dbias);
  return rdst_launch_status("conv3x_wgrad_reduce");
}

}  // namespace

// RDST_ENOTSUP = not one of the covered shapes.  dY in the output geometry (pixel-shuffled when g.r == 2).  slab: conv3_wgrad_slab_bytes.
int conv3x_wgrad_f32(const float* X, int64_t ldx, int in_act, const float* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                     const ConvGeom& g, float s, hipStream_t st) {
  if (!slab || g.ks != 3 || g.pad != 1 || in_act || g.W % 32 || ((uintptr_t)slab & 15)) return RDST_ENOTSUP;
  if (((uintptr_t)X & 3) || ((uintptr_t)dY & 3)) return RDST_ENOTSUP;
  int shape = 0;
  if (g.Cin == 150 && g.Cout == 60 && g.r == 1) shape = 1;
  else if (g.Cin == 60 && g.Cout == 60 && g.r == 1) shape = 2;
  else if (g.Cin == 60 && g.Cout == 240 && g.r == 2) shape = 3;
  if (!shape) return RDST_ENOTSUP;
  const int64_t xb = ((g.pixels() - 1) * ldx + g.Cin) * 4;
  const int64_t yb = ((g.pixels() * g.r * g.r - 1) * lddy + g.Cout / (g.r * g.r)) * 4;
  if (xb >= (1ll << 31) || yb >= (1ll << 31)) return RDST_ENOTSUP;
  W3XArgs p{};
  p.X = X; p.ldx = ldx; p.x_bytes = (int)xb; p.dY = dY; p.lddy = lddy; p.dy_bytes = (int)yb; p.slab = slab;
  p.B = g.B; p.H = g.H; p.W = g.W; p.ymul = 1; p.yoff = 0; p.xmul = 1; p.xoff = 0;
  if (shape == 1) return launch_w3x<150, 60>(p, s, 1, 0, dW, dbias, st, "conv3x_wgrad_150_60");
  if (shape == 2) return launch_w3x<60, 60>(p, s, 1, 0, dW, dbias, st, "conv3x_wgrad_60_60");
  for (int q = 0; q < 4; ++q) {   // conv channel 4 c' + q is channel c' of output pixel (2y + (q >> 1), 2x + (q & 1))
    W3XArgs pq = p;
    pq.ymul = 2; pq.yoff = q >> 1; pq.xmul = 2; pq.xoff = q & 1;
    if (int rc = launch_w3x<60, 60>(pq, s, 4, q, dW, dbias, st, "conv3x_wgrad_60_240_q")) return rc;
  }
  return 0;
}
