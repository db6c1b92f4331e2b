// K4/K5 weight gradient for the E1 shapes (bf16): dW[co][ci][ky][kx] = s * sum_p dY[p][co] * X[p + (ky,kx) - 1][ci],
// dbias[co] = s * sum_p dY[p][co], with the WHOLE dW ACCUMULATOR OF A WORKGROUP IN REGISTERS.
//
// As a GEMM the contraction runs over pixels (131072 and more) and the output is small: 64 x (9 x 160) fp32 = 368 KB
// for the 150 -> 60 fusion conv — less than a CU's 512 KB register file.  So a workgroup (8 waves, 2 per SIMD) owns a set
// of pixels and ALL 9 taps; wave w keeps its 9-12 output tiles (32 output channels x 32 input channels of one tap) as
// MFMA accumulators from the first pixel to the last and nothing but the two operand streams moves:
//   * pixels are cut into the strips of conv3_mfma.hip (32 wide, SH rows high); X rows (34 pixels with the halo) and dY
//     rows roll through two LDS rings, filled by LDS-DMA (buffer_load ... lds, out-of-range lanes write zeros: image
//     border, pad slots, tensor end), RPS rows per step, one barrier per step, the next step's rows in flight;
//   * both operands have the contraction index (pixel) as their ROW in memory: fragments are read transposed with
//     ds_read_b64_tr_b16 (pixel strides = 64 or 192 mod 256 bytes: the four rows of a transposed read fall on the four
//     64-B quarters of the bank row); the three kx taps are pixel offsets of the SAME X image, the ky taps ring rows;
//   * one A fragment (dY, 32 channels x 16 pixels) serves all tiles of the wave; d(bias) is one more tile whose B
//     operand is a register of ones;
//   * the gradient of conv + PixelShuffle(2) reads the shuffled dY as it lies (channel order k' = 60 q + c' per low-res
//     pixel, q = sub-pixel); the 240 output channels are split over two workgroup roles (4 channel tiles each).
// A workgroup dumps its accumulators as they stand (256-B rows per register) into its slab; conv3_wgrad_reduce sums the
// slabs in fixed order (deterministic), applies s, and scatters into nn.Conv2d's (Cout, Cin, 3, 3) layout.
// The old stripe kernel (conv_mfma.hip) split the taps over three workgroups per pixel range, read X and dY three
// times and staged them through registers: 138 us average per launch for the same work.
#include "conv.h"
#include "mfma.h"

namespace {

constexpr int lds_tr_stride(int bytes) {   // >= bytes, = 64 or 192 (mod 256)
  int s = (bytes + 63) / 64 * 64;
  while ((s % 256) != 64 && (s % 256) != 192) s += 64;
  return s;
}

struct W3Args {
  const bf16* X; int64_t ldx; int x_bytes;
  const bf16* dY; int64_t lddy; int dy_bytes;   // output geometry (pixel-shuffled when UNSHUF)
  float* slab;                                  // [grid][SLABF] accumulator dumps
  int B, H, W;
  int SH, nys, nstrips, npg;                    // strips; npg = pixel groups (grid = npg * roles)
};

template <int CI, int CO, bool UNSHUF>
struct W3Cfg {
  static constexpr int ROLES = CO > 64 ? (CO + 127) / 128 : 1;   // workgroup roles: each owns <= 4 channel tiles
  static constexpr int CT = ((CO + 31) / 32 + ROLES - 1) / ROLES; // channel tiles per workgroup (2 or 4)
  static constexpr int WPC = 8 / CT;                             // waves per channel tile
  static constexpr int CIT = (CI + 31) / 32;
  static constexpr int TPC = 9 * CIT + 1;                        // tiles per channel tile, + the bias tile
  static constexpr int TPW = (TPC + WPC - 1) / WPC;              // tiles per wave
  static constexpr int RPS = 2;
  static constexpr int XS = lds_tr_stride(2 * CI), XSLOTS = XS / 16, XD = (2 * CI + 15) / 16;
  static constexpr int XP = (34 * XSLOTS + 63) / 64, XROWB = XP * 1024, NRX = 2 * RPS + 2;
  static constexpr int YS = lds_tr_stride(2 * CO), YSLOTS = YS / 16, YD = (2 * CO + 15) / 16;
  static constexpr int YP = (32 * YSLOTS + 63) / 64, YROWB = YP * 1024, NRY = 2 * RPS;
  static constexpr int Y_OFF = NRX * XROWB;
  static constexpr int SMEM = Y_OFF + NRY * YROWB + 64;
  static constexpr int SLABF = 8 * TPW * 1024;                   // floats per workgroup slab
  static_assert(8 % CT == 0, "waves per channel tile");
  static_assert(!UNSHUF || CO % 16 == 0, "un-shuffled dY rows: two sub-pixel pairs of whole 16-B slots");
};

template <int CI, int CO, bool UNSHUF>
__global__ void __launch_bounds__(512, 2) conv3_wgrad_kernel(const W3Args p) {
  using CF = W3Cfg<CI, CO, UNSHUF>;
  constexpr int RPS = CF::RPS, TPW = CF::TPW, CIT = CF::CIT, XS = CF::XS, YS = CF::YS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int role = blockIdx.x % CF::ROLES, pg = blockIdx.x / CF::ROLES;
  const int ctl = wave / CF::WPC, wq = wave % CF::WPC;      // channel tile inside the workgroup, chunk of its tiles
  const int ct = role * CF::CT + ctl;                       // channel tile of the (permuted) output channels

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
  lds_zero16(smem, CF::SMEM - 64, tid, 512);
  __syncthreads();

  // ---- the wave's tiles: tile j is n = wq * TPW + j of its channel tile: n < 9 CIT -> (tap, ci tile), n == 9 CIT -> bias
  int boff[TPW];          // lane offset of the B fragment inside an X ring row (without the k-step), per tile
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
  const int aoff = (8 * h + q4) * YS + ct * 64 + 32 * (gq & 1) + 8 * pp;   // A fragment lane offset inside a dY ring row
  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const bf16x8_t ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};

  const int nxs = W / 32;
  // strips of this pixel group: strip = pg + k * npg
  for (int strip = pg; strip < p.nstrips; strip += p.npg) {
    const int xs = strip % nxs, tq = strip / nxs;
    const int ysg = tq % p.nys, b = tq / p.nys;
    const int y0 = ysg * p.SH, x0 = xs * 32;
    const int nrows = (H - y0 < p.SH) ? H - y0 : p.SH;
    const int nsteps = (nrows + RPS - 1) / RPS;

    // X ring: rel row q = input row y0 - 1 + q, slot q % NRX;  dY ring: row q = output row y0 + q, slot q % NRY
    auto x_lane_off = [&](int pi) {
      const int sidx = pi * 64 + lane;
      const int px = sidx / CF::XSLOTS, sl = sidx - px * CF::XSLOTS;
      const int x = x0 - 1 + px;
      return (x >= 0 && x < W && px < 34 && sl < CF::XD) ? x * ((int)p.ldx * 2) + sl * 16 : -1;
    };
    auto y_lane_off = [&](int pi) {
      const int sidx = pi * 64 + lane;
      const int px = sidx / CF::YSLOTS, sl = sidx - px * CF::YSLOTS;
      if constexpr (UNSHUF) {
        constexpr int HS = CF::YD / 2;
        const int i2 = sl >= HS ? 1 : 0;
        return (px < 32 && sl < CF::YD) ? (i2 * 2 * W + 2 * (x0 + px)) * ((int)p.lddy * 2) + (sl - i2 * HS) * 16 : -1;
      } else {
        return (px < 32 && sl < CF::YD) ? (x0 + px) * ((int)p.lddy * 2) + sl * 16 : -1;
      }
    };
    auto x_row = [&](int rel, int pi, int loff) {
      const int y = y0 - 1 + rel;
      const int rowbase = (int)((((int64_t)b * H + y) * W) * (p.ldx * 2));
      dma(rx, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((rel % CF::NRX) * CF::XROWB + pi * 1024)),
          (y >= 0 && y < H && loff >= 0) ? rowbase + loff : p.x_bytes);
    };
    auto y_row = [&](int rel, int pi, int loff) {
      const int y = y0 + rel;
      const int rowbase = UNSHUF ? (int)((((int64_t)b * (2 * H) + 2 * y) * (int64_t)(2 * W)) * (p.lddy * 2))
                                 : (int)((((int64_t)b * H + y) * W) * (p.lddy * 2));
      dma(ry, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::Y_OFF + (rel % CF::NRY) * CF::YROWB + pi * 1024)),
          (rel < nrows && loff >= 0) ? rowbase + loff : p.dy_bytes);
    };
    // a row set = X rows + dY rows of a step, pieces dealt round-robin to the 8 waves
    auto load_set = [&](int xrel0, int nx, int yrel0, int ny) {
      const int npx = nx * CF::XP, tot = npx + ny * CF::YP;
      for (int q = wave; q < tot; q += 8) {
        if (q < npx) {
          const int rr = q / CF::XP, pi = q - rr * CF::XP;
          x_row(xrel0 + rr, pi, x_lane_off(pi));
        } else {
          const int q2 = q - npx;
          const int rr = q2 / CF::YP, pi = q2 - rr * CF::YP;
          y_row(yrel0 + rr, pi, y_lane_off(pi));
        }
      }
    };
    load_set(0, RPS + 2, 0, RPS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int j = 0; j < nsteps; ++j) {
      if (j + 1 < nsteps) load_set((j + 1) * RPS + 2, RPS, (j + 1) * RPS, RPS);   // slots no wave reads in this step
#pragma unroll
      for (int i = 0; i < RPS; ++i) {
        const int yo = j * RPS + i;                          // output row inside the strip (rows past nrows hold zeros)
        const char* yrow = smem + CF::Y_OFF + (yo % CF::NRY) * CF::YROWB + aoff;
        const char* xrow[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) xrow[ky] = smem + ((yo + ky) % CF::NRX) * CF::XROWB;
        typedef __attribute__((address_space(3))) s16x4_t* lds_p;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(yrow + ks * 16 * YS));
          const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(yrow + ks * 16 * YS + 4 * YS));
          const uint2 ua0 = __builtin_bit_cast(uint2, a0), ua1 = __builtin_bit_cast(uint2, a1);
          Pack16 a;
          a.w[0] = ua0.x; a.w[1] = ua0.y; a.w[2] = ua1.x; a.w[3] = ua1.y;
          const bf16x8_t av = __builtin_bit_cast(bf16x8_t, a);
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const int n = wq * TPW + t;                      // wave-uniform
            if (n < 9 * CIT) {
              const char* xr = (bky[t] == 0 ? xrow[0] : (bky[t] == 1 ? xrow[1] : xrow[2])) + boff[t] + ks * 16 * XS;
              const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)xr);
              const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(xr + 4 * XS));
              const uint2 ub0 = __builtin_bit_cast(uint2, b0), ub1 = __builtin_bit_cast(uint2, b1);
              Pack16 bq;
              bq.w[0] = ub0.x; bq.w[1] = ub0.y; bq.w[2] = ub1.x; bq.w[3] = ub1.y;
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, bq), acc[t], 0, 0, 0);
            } else if (n == 9 * CIT) {
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, ones, acc[t], 0, 0, 0);   // every column = sum over pixels
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  // ---- dump the accumulators: slab[wg][wave][tile][reg][lane] ---------------------------------------------------
  // as bf16 pairs (registers v, v + 1 = two consecutive output channels of the lane's column): the partial sums of one
  // workgroup are rounded, the sum over the workgroups runs in fp32 (see reduce_batch.h on the G4 slabs)
  uint32_t* my = reinterpret_cast<uint32_t*>(p.slab) + (size_t)blockIdx.x * (CF::SLABF / 2) + (size_t)wave * TPW * 512;
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int v = 0; v < 16; v += 2) my[(t * 8 + (v >> 1)) * 64 + lane] = pack_bf16x2(acc[t][v], acc[t][v + 1]);
}

// dW[co][ci][tap] = s * sum_g slab[g][...], dbias[co] likewise (column 0 of the bias tile); one thread per accumulator
// element of the role-major dump: i -> (role, wave, tile, reg, lane)
template <int CI, int CO, bool UNSHUF>
__global__ void __launch_bounds__(256) conv3_wgrad_reduce_kernel(const float* __restrict__ slab, int npg, float s,
                                                                 float* __restrict__ dW, float* __restrict__ dbias) {
  using CF = W3Cfg<CI, CO, UNSHUF>;
  // one thread per bf16 PAIR of the role-major dump (registers 2 vp, 2 vp + 1 of one lane): 4-byte loads
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= CF::ROLES * CF::SLABF / 2) return;
  const int role = i / (CF::SLABF / 2), e = i - role * (CF::SLABF / 2);
  const int lane = e & 63, vp = (e >> 6) & 7, t = (e >> 9) % CF::TPW, wave = (e >> 9) / CF::TPW;
  const int ctl = wave / CF::WPC, wq = wave % CF::WPC;
  const int n = wq * CF::TPW + t;
  if (n > 9 * CF::CIT) return;
  const int col = lane & 31;
  const bool is_b = n == 9 * CF::CIT;
  const int tap = n / CF::CIT, ci = (n - tap * CF::CIT) * 32 + col;
  if (is_b ? col != 0 : ci >= CI) return;
  float a0 = 0.f, a1 = 0.f;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(slab) + (size_t)role * (CF::SLABF / 2) + e;
  const size_t gstride = (size_t)CF::ROLES * (CF::SLABF / 2);
  for (int g0 = 0; g0 < npg; g0 += 16) {   // 16 loads in flight, summed in the same fixed order
    uint32_t x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = (g0 + u < npg) ? src[(size_t)(g0 + u) * gstride] : 0u;
#pragma unroll
    for (int u = 0; u < 16; ++u) { a0 += __uint_as_float(x[u] << 16); a1 += __uint_as_float(x[u] & 0xffff0000u); }
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int v = 2 * vp + hf;
    const int m = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);          // row of the accumulator tile
    int co = (role * CF::CT + ctl) * 32 + m;
    if (co >= CO) continue;
    if (UNSHUF) co = 4 * (co % (CO / 4)) + co / (CO / 4);            // k' = (CO/4) q + c'  ->  conv channel 4 c' + q
    const float a = hf ? a1 : a0;
    if (is_b) {
      if (dbias) dbias[co] = a * s;
    } else if (dW) {
      dW[((int64_t)co * CI + ci) * 9 + tap] = a * s;
    }
  }
}

template <int CI, int CO, bool UNSHUF>
int launch_w3(W3Args& p, float s, float* dW, float* dbias, hipStream_t st, const char* what) {
  using CF = W3Cfg<CI, CO, UNSHUF>;
  int SH = 16;
  auto count = [&](int sh) { return (int64_t)p.B * ((p.H + sh - 1) / sh) * (p.W / 32); };
  const int want = 256 / CF::ROLES;
  while (count(SH) < want && SH > CF::RPS) SH /= 2;
  p.SH = SH;
  p.nys = (p.H + SH - 1) / SH;
  const int64_t ns = count(SH);
  if (ns >= (1ll << 31)) return RDST_ENOTSUP;
  p.nstrips = (int)ns;
  p.npg = ns < want ? (int)ns : want;
  const int grid = p.npg * CF::ROLES;
  auto kern = conv3_wgrad_kernel<CI, CO, UNSHUF>;
  // (per launch: the attribute is per DEVICE, a process-wide "done" flag would leave a second GPU without it)
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), CF::SMEM, st, p);
  if (int rc = rdst_launch_status(what)) return rc;
  const int tot = CF::ROLES * CF::SLABF / 2;
  hipLaunchKernelGGL((conv3_wgrad_reduce_kernel<CI, CO, UNSHUF>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, p.slab,
                     p.npg, s, dW, dbias);
  return rdst_launch_status("conv3_wgrad_reduce");
}

bool rows_ok(const void* ptr, int64_t ld) { return ((uintptr_t)ptr & 3) == 0 && (ld & 1) == 0; }

}  // namespace

size_t conv3_wgrad_slab_bytes(int Cin, int Cout) {
  // <= 256 workgroups x 8 waves x TPW tiles x 4 KB (the W3Cfg arithmetic for run-time channel counts)
  const int cot = (Cout + 31) / 32, roles = Cout > 64 ? (Cout + 127) / 128 : 1;
  const int ct = (cot + roles - 1) / roles;
  if (ct <= 0 || 8 % ct) return 256;
  const int wpc = 8 / ct, tpc = 9 * ((Cin + 31) / 32) + 1, tpw = (tpc + wpc - 1) / wpc;
  return (size_t)256 * 8 * tpw * 4096 + 256;
}

// RDST_ENOTSUP = not one of the covered shapes.  dY in the output geometry (pixel-shuffled when g.r == 2).
int conv3_wgrad_bf16(const bf16* X, int64_t ldx, int in_act, const bf16* dY, int64_t lddy, float* dW, float* dbias, float* slab,
                     const ConvGeom& g, float s, hipStream_t st) {
  if (!slab || g.ks != 3 || g.pad != 1 || in_act || g.W % 32 || ((uintptr_t)slab & 15)) return RDST_ENOTSUP;
  if (!rows_ok(X, ldx) || !rows_ok(dY, lddy)) return RDST_ENOTSUP;
  int shape = 0;
  if (g.Cin == 150 && g.Cout == 60 && g.r == 1) shape = 1;
  else if (g.Cin == 60 && g.Cout == 60 && g.r == 1) shape = 2;
  else if (g.Cin == 60 && g.Cout == 240 && g.r == 2 && lddy == 60) shape = 3;
  if (!shape) return RDST_ENOTSUP;
  const int64_t xb = ((g.pixels() - 1) * ldx + g.Cin) * 2;
  const int64_t yb = ((g.pixels() * g.r * g.r - 1) * lddy + g.Cout / (g.r * g.r)) * 2;
  if (xb >= (1ll << 31) || yb >= (1ll << 31)) return RDST_ENOTSUP;
  W3Args p{};
  p.X = X; p.ldx = ldx; p.x_bytes = (int)xb; p.dY = dY; p.lddy = lddy; p.dy_bytes = (int)yb; p.slab = slab;
  p.B = g.B; p.H = g.H; p.W = g.W;
  if (shape == 1) return launch_w3<150, 60, false>(p, s, dW, dbias, st, "conv3_wgrad_150_60");
  if (shape == 2) return launch_w3<60, 60, false>(p, s, dW, dbias, st, "conv3_wgrad_60_60");
  return launch_w3<60, 240, true>(p, s, dW, dbias, st, "conv3_wgrad_60_240_unshuf");
}
