// K7 forward for the E1 shapes (bf16): y = x + fc2(GELU(fc1(LayerNorm(x)))) in one pass, re-cut on the skeleton of
// lin3_mfma.hip (weights in registers as pre-packed bf16 fragments, token tiles by LDS-DMA, counted vmcnt waits):
//   * a workgroup of 8 waves takes 128-token tiles of x into an LDS ring (NBUF tiles in flight);
//   * phase 1, H^T = (W1 gamma) x^T on the RAW rows: wave w owns hidden tile(s) w mod NT1 for its token sub-tiles;
//     LayerNorm enters in the epilogue, h = GELU(rstd (H - mean S1) + b1'), (mean, rstd) from a two-pass sum over the
//     LDS-resident row; h leaves as bf16 rows [token][hidden] of ONE LDS tile (8-B stores: the 4 consecutive hidden
//     units of a register group) — it never touches HBM and is not kept (the backward recomputes it);
//   * phase 2, Y^T = W2 h^T + b2 + x: B fragments are ds_read_b128 of the h rows, the residual x is read back from the
//     x tile (8 B per register group), 16-B row stores after one v_permlane32_swap per register pair.
// Per token the kernel moves 2 C * 2 bytes; what bounds it is the GELU on the vector ALUs (~20 instructions x 2C per
// token): ~16 us at C = 120 against 65 us for the LDS-weight kernel of mlp_mfma.hip it replaces.
// C = 60 / 90 / 120 with hid = 2 C; everything else stays on mlp_mfma.hip.
#include "linear.h"
#include "mfma.h"

#ifndef M3_ABL
#define M3_ABL 0   // compile-time ablations (tools/abl_build.sh): 1 no GELU, 2 no phase-1 MFMAs, 4 no phase-2 MFMAs, 8 no stores
#endif

namespace {

constexpr int M3_TT = 128;

constexpr int m3_gcd(int a, int b) { return b == 0 ? a : m3_gcd(b, a % b); }
constexpr int m3_stride(int K) {  // bytes: covers every k-step, odd number of 16-B slots
  int s = (K + 15) / 16 * 32;
  if (((s / 16) & 1) == 0) s += 16;
  return s;
}

struct M3Args {
  const bf16* X; int64_t ldx; int x_bytes;
  const bf16* W1p; const float* sb1;   // fc1: fragments [NT1][KS1], S1 / b1' (2 x NT1 x 32 floats)
  const bf16* W2p; const float* sb2;   // fc2: fragments [NT2][KS2], (S2 unused) / b2
  bf16* Y; int64_t ldy;
  float* stats;
  int M, ntiles;
  unsigned long long* stamps;   // debug build: [grid][8] cycle counters (RDST_M3_STAMPS), else null
};

#ifndef M3_NW90
#define M3_NW90 12
#endif
// hidden tiles 4 / 6 / 8 -> 8 / 12 / 8 waves.  C = 90 (6 hidden tiles, 105 KB of LDS: one workgroup per CU) ran 6 waves per CU until
// round 5 and took as long as C = 120 (35.9 against 38.6 us in the step); 12 waves = two per hidden tile, one output item each
constexpr int m3_nw(int C) { return ((2 * C + 31) / 32) % 3 == 0 ? M3_NW90 : 8; }

template <int C>
struct M3Cfg {
  static constexpr int HID = 2 * C;
  static constexpr int KS1 = (C + 15) / 16, NT1 = (HID + 31) / 32;
  static constexpr int KS2 = (HID + 15) / 16, NT2 = (C + 31) / 32;
  static constexpr int XS = m3_stride(C), XSLOTS = XS / 16, XD = (2 * C + 15) / 16;
  static constexpr int TP = (M3_TT * XSLOTS + 63) / 64, TILEB = TP * 1024;
  static constexpr int HS = m3_stride(HID), HTILEB = M3_TT * HS;
  // waves: a divisor arrangement in which every wave owns ONE hidden tile and ONE output tile (one fragment set each)
  static constexpr int NW = m3_nw(C), NTHR = 64 * NW;
  static constexpr int NJ1 = (NT1 * 4 + NW - 1) / NW, P1 = NT1 / m3_gcd(NW, NT1), ND1 = NJ1 < P1 ? NJ1 : P1;
  static constexpr int NJ2 = (NT2 * 4 + NW - 1) / NW, P2 = NT2 / m3_gcd(NW, NT2), ND2 = NJ2 < P2 ? NJ2 : P2;
  static constexpr int NBUF = 2;
  static constexpr int CNT = (TP + NW - 1) / NW;
  static constexpr int H_OFF = NBUF * TILEB;
  static constexpr int STAT_OFF = H_OFF + HTILEB;                     // [NBUF][128][2] floats
  static constexpr int SB1_OFF = STAT_OFF + NBUF * M3_TT * 2 * 4;     // [2][NT1*32]
  static constexpr int SB1_B = (2 * NT1 * 32 * 4 + 1023) / 1024 * 1024;
  static constexpr int SB2_OFF = SB1_OFF + SB1_B;                     // [2][NT2*32]
  static constexpr int SB2_B = (2 * NT2 * 32 * 4 + 1023) / 1024 * 1024;
  static constexpr int TAB_OFF = SB2_OFF + SB2_B;                     // GELU table (common.h)
  static constexpr int SMEM = TAB_OFF + RDST_GELU_TAB4_BYTES;
  static constexpr int WGCU = SMEM <= 80 * 1024 ? 2 : 1;
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(TP >= NW, "every wave owns at least one piece of a tile");
  static_assert(CNT + 1 + 2 * ((NT2 * 4) / NW) < 64, "vmcnt is a 6-bit counter");
};

template <int C>
__global__ void __launch_bounds__(64 * m3_nw(C), M3Cfg<C>::WGCU == 2 ? (2 * m3_nw(C) + 3) / 4 : (m3_nw(C) + 3) / 4) mlp3_fwd_kernel(const M3Args p) {
  using CF = M3Cfg<C>;
  constexpr int KS1 = CF::KS1, KS2 = CF::KS2, NT1 = CF::NT1, NT2 = CF::NT2, XS = CF::XS, HS = CF::HS, NBUF = CF::NBUF, NW = CF::NW;
  constexpr int ND1 = CF::ND1, ND2 = CF::ND2, NJ1 = CF::NJ1, NJ2 = CF::NJ2, CNT = CF::CNT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* statL = reinterpret_cast<float*>(smem + CF::STAT_OFF);
  const float* sb1L = reinterpret_cast<const float*>(smem + CF::SB1_OFF);
  const float* sb2L = reinterpret_cast<const float*>(smem + CF::SB2_OFF);
  char* hL = smem + CF::H_OFF;
  gelu_tab4_fill(smem + CF::TAB_OFF, tid, CF::NTHR);   // visible after the first tile's barrier
  const char* gtab = smem + CF::TAB_OFF;

  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void* ptr, uint32_t bytes) {
    u32x4s_t q;
    q.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    q.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32) & 0xffffu);
    q.z = __builtin_amdgcn_readfirstlane(bytes);
    q.w = 0x00020000u;
    return q;
  };
  const u32x4s_t rsx = make_rsrc(p.X, (uint32_t)p.x_bytes), rs1 = make_rsrc(p.sb1, 2 * NT1 * 32 * 4), rs2 = make_rsrc(p.sb2, 2 * NT2 * 32 * 4);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const u32x4s_t& rs, uint32_t ldst, int off) {   // inline asm: see conv3_mfma.hip
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rs) : "memory");
  };
  const int grid = gridDim.x;
  unsigned long long tprev = RDST_DBGV(p.stamps) ? __builtin_readcyclecounter() : 0ull;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_ADD(k)                                                   \
  if (RDST_DBGV(p.stamps)) {                                           \
    const unsigned long long tn_ = __builtin_readcyclecounter();       \
    tacc[k] += tn_ - tprev;                                            \
    tprev = tn_;                                                       \
  }
  // ---- weight fragments (inline-asm loads, issued before the tiles: see lin3_mfma.hip) ---------------------------
  typedef uint32_t u32x4v_t __attribute__((ext_vector_type(4)));
  u32x4v_t w1[ND1][KS1], w2[ND2][KS2];
#pragma unroll
  for (int jd = 0; jd < ND1; ++jd) {
    const int nt = (wave + NW * jd) % NT1;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const char* src = reinterpret_cast<const char*>(p.W1p) + (((int64_t)nt * KS1 + ks) * 64 + lane) * 16;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w1[jd][ks]) : "v"(src) : "memory");
    }
  }
#pragma unroll
  for (int jd = 0; jd < ND2; ++jd) {
    const int nt = (wave + NW * jd) % NT2;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const char* src = reinterpret_cast<const char*>(p.W2p) + (((int64_t)nt * KS2 + ks) * 64 + lane) * 16;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w2[jd][ks]) : "v"(src) : "memory");
    }
  }
  {  // S / b' of both layers by LDS-DMA: every wave one piece of each (duplicates write the same bytes)
    constexpr int N1P = CF::SB1_B / 1024, N2P = CF::SB2_B / 1024;
    const int p1 = wave % N1P, p2 = wave % N2P;
    dma(rs1, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::SB1_OFF + p1 * 1024)), p1 * 1024 + lane * 16);
    dma(rs2, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::SB2_OFF + p2 * 1024)), p2 * 1024 + lane * 16);
  }
  auto issue_tile = [&](int tile, int b) {   // exactly CNT pieces per wave
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      int q = wave + NW * i;
      q = q < CF::TP ? q : q - NW;
      const int sidx = q * 64 + lane;
      const int tok = sidx / CF::XSLOTS, sl = sidx - tok * CF::XSLOTS;
      const int grow = tile * M3_TT + tok;
      const bool ok = tile < p.ntiles && tok < M3_TT && sl < CF::XD && grow < p.M;
      dma(rsx, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(b * CF::TILEB + q * 1024)),
          ok ? grow * ((int)p.ldx * 2) + sl * 16 : p.x_bytes);   // (extent < 2^31 bytes)
    }
  };
  static_assert(NBUF == 2, "the ring below is written for two buffers");
  // tiles of this workgroup: blockIdx.x + k * grid; tile k lives in buffer k & 1; tile k + 1 is in flight while k is computed
  issue_tile(blockIdx.x, 0);
  issue_tile(blockIdx.x + grid, 1);
  // Waits (memory operations retire in issue order; loads, stores and LDS-DMA share one counter): when iteration k
  // starts, the operations YOUNGER than tile k's pieces are tile k + 1's CNT pieces and — from the second iteration on —
  // the previous iteration's statistics store and its phase-2 stores, at least NST of them in every wave.  Waiting for
  // "all but that many" therefore never waits for a store's round trip and never lets tile k slip.
  constexpr int NST = 1 + 2 * ((NT2 * 4) / NW);
  int kk = 0;
  for (int tile = blockIdx.x; tile < p.ntiles; tile += grid, ++kk) {
    const int b = kk & 1;
    if (kk == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + NST) : "memory");
#pragma unroll
    for (int jd = 0; jd < ND1; ++jd)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(w1[jd][ks]));
#pragma unroll
    for (int jd = 0; jd < ND2; ++jd)
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) asm volatile("" : "+v"(w2[jd][ks]));
    __syncthreads();   // B0: tile landed (and: every wave is done with the previous tile's h)
    STAMP_ADD(0);
    const char* tb = smem + b * CF::TILEB;
    float* st = statL + b * M3_TT * 2;
    for (int tok = tid >> 2; tok < M3_TT; tok += CF::NTHR / 4) {  // (mean, rstd): 4 lanes per token, two passes over the row's 16-B slots
      const int part = tid & 3;
      constexpr int NSL = (CF::XD + 3) / 4;
      float xv[NSL][8];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NSL; ++i) {
        const int sl = part + 4 * i;
        const Pack16 q = *reinterpret_cast<const Pack16*>(tb + tok * XS + (sl < CF::XD ? sl : 0) * 16);
        Mma<bf16>::unpack(q, xv[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool valid = sl < CF::XD && sl * 8 + e < C;
          xv[i][e] = valid ? xv[i][e] : 0.f;
          sum += xv[i][e];
        }
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      const float mean = sum * (1.0f / C);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NSL; ++i) {
        const int sl = part + 4 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool valid = sl < CF::XD && sl * 8 + e < C;
          const float d = xv[i][e] - mean;
          sq = valid ? fmaf(d, d, sq) : sq;
        }
      }
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      const float rstd = rsqrtf(sq * (1.0f / C) + 1e-5f);
      if (part == 0) {
        st[tok * 2] = mean;
        st[tok * 2 + 1] = rstd;
        const int grow = tile * M3_TT + tok;
        if (grow < p.M) *reinterpret_cast<float2*>(p.stats + (int64_t)grow * 2) = make_float2(mean, rstd);
      }
    }
    __syncthreads();   // B1: statistics visible
    STAMP_ADD(1);
    // ---- phase 1: h = GELU(LN(x) W1^T + b1) -> LDS -----------------------------------------------------------
#pragma unroll
    for (int jd = 0; jd < ND1; ++jd)
#pragma unroll 1
      for (int j = jd; j < NJ1; j += ND1) {
        const int item = wave + NW * j;
        if (item >= NT1 * 4) break;
        const int nt = item % NT1, tt = item / NT1;
        const int tok = tt * 32 + r;
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const char* brow = tb + tok * XS + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          Pack16 bq = *reinterpret_cast<const Pack16*>(brow + ks * 32);
          if (C % 16 != 0 && ks == KS1 - 1) {   // channels past C inside the last slot: next channels of the memory row, any bits
            constexpr int c0 = C % 16 < 8 ? C % 16 : 8, c1 = C % 16 > 8 ? C % 16 - 8 : 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) bq.w[d] = (2 * d < (h ? c1 : c0)) ? bq.w[d] : 0u;
          }
          if (!(M3_ABL & 2)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w1[jd][ks]), __builtin_bit_cast(bf16x8_t, bq), acc, 0, 0, 0);
          else acc[ks & 15] += __uint_as_float(bq.w[0]);
        }
        const float2 mr = *reinterpret_cast<const float2*>(st + tok * 2);
        const float rstd = mr.y, nrm = -mr.y * mr.x;
        char* hrow = hL + tok * HS;
        // GELU = u Phi(u), Phi from the LDS table (common.h; the erf / exp2 / rcp form kept this phase at the vector-issue
        // rate): all sixteen table reads of the tile are in flight before the first is used
        float u[16], fr[16];
        uint32_t en[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = nt * 32 + 8 * g4 + 4 * h;
          const float4 S4 = *reinterpret_cast<const float4*>(sb1L + n0);
          const float4 B4 = *reinterpret_cast<const float4*>(sb1L + NT1 * 32 + n0);
          u[4 * g4] = fmaf(rstd, acc[4 * g4], fmaf(nrm, S4.x, B4.x));
          u[4 * g4 + 1] = fmaf(rstd, acc[4 * g4 + 1], fmaf(nrm, S4.y, B4.y));
          u[4 * g4 + 2] = fmaf(rstd, acc[4 * g4 + 2], fmaf(nrm, S4.z, B4.z));
          u[4 * g4 + 3] = fmaf(rstd, acc[4 * g4 + 3], fmaf(nrm, S4.w, B4.w));
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          uint32_t off;
          if (M3_ABL & 1) { fr[v] = 0.f; en[v] = 0x3c003c00u; continue; }
          fr[v] = gelu_tab4_index(u[v], off);
          en[v] = *reinterpret_cast<const uint32_t*>(gtab + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = nt * 32 + 8 * g4 + 4 * h;
          const float h0 = u[4 * g4] * gelu_tab_lerp(fr[4 * g4], en[4 * g4]), h1 = u[4 * g4 + 1] * gelu_tab_lerp(fr[4 * g4 + 1], en[4 * g4 + 1]);
          const float h2 = u[4 * g4 + 2] * gelu_tab_lerp(fr[4 * g4 + 2], en[4 * g4 + 2]), h3 = u[4 * g4 + 3] * gelu_tab_lerp(fr[4 * g4 + 3], en[4 * g4 + 3]);
          u32x2_a4 uu;
          uu.x = pack_bf16x2(h0, h1); uu.y = pack_bf16x2(h2, h3);
          if (n0 * 2 + 8 <= HS) *reinterpret_cast<u32x2_a4*>(hrow + n0 * 2) = uu;   // (padded hidden units past the row are dropped)
        }
      }
    STAMP_ADD(2);
    __syncthreads();   // B2: h complete
    STAMP_ADD(3);
    // ---- phase 2: y = x + h W2^T + b2 -------------------------------------------------------------------------
#pragma unroll
    for (int jd = 0; jd < ND2; ++jd)
#pragma unroll 1
      for (int j = jd; j < NJ2; j += ND2) {
        const int item = wave + NW * j;
        if (item >= NT2 * 4) break;
        const int nt = item % NT2, tt = item / NT2;
        const int tok = tt * 32 + r;
        const int grow = tile * M3_TT + tok;
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const char* brow = hL + tok * HS + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          const Pack16 bq = *reinterpret_cast<const Pack16*>(brow + ks * 32);
          if (!(M3_ABL & 4)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w2[jd][ks]), __builtin_bit_cast(bf16x8_t, bq), acc, 0, 0, 0);
          else acc[ks & 15] += __uint_as_float(bq.w[0]);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = nt * 32 + 8 * g4 + 4 * h;
          const float4 B4 = *reinterpret_cast<const float4*>(sb2L + NT2 * 32 + n0);
          uint32_t rx = 0u, ry = 0u;
          if (n0 < C) {   // the residual's 4 channels: 8 B of the token's own row in the x tile (channels past C are never stored)
            const u32x2_a4 rr = *reinterpret_cast<const u32x2_a4*>(tb + tok * XS + n0 * 2);
            rx = rr.x; ry = rr.y;
          }
          acc[4 * g4] += B4.x + bf16lo(rx); acc[4 * g4 + 1] += B4.y + bf16hi(rx);
          acc[4 * g4 + 2] += B4.z + bf16lo(ry); acc[4 * g4 + 3] += B4.w + bf16hi(ry);
        }
        if (grow < p.M && !(M3_ABL & 8)) {
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float c8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * gp + e]), __float_as_uint(acc[8 * gp + 4 + e]),
                                                               false, false);
              c8[e] = __uint_as_float(sw[0]);
              c8[4 + e] = __uint_as_float(sw[1]);
            }
            const int cb = nt * 32 + 8 * (2 * gp + h);
            const int nv = C - cb;                              // valid outputs from cb on (C is a multiple of 4)
            if (nv <= 0) continue;
            bf16* yp = p.Y + (grow * (int)p.ldy + cb);
            if (nv >= 8) {
              u32x4_a4 u;
              u.x = pack_bf16x2(c8[0], c8[1]); u.y = pack_bf16x2(c8[2], c8[3]);
              u.z = pack_bf16x2(c8[4], c8[5]); u.w = pack_bf16x2(c8[6], c8[7]);
              *reinterpret_cast<u32x4_a4*>(yp) = u;
            } else {
#pragma unroll
              for (int d = 0; d < 3; ++d)
                if (2 * d + 2 <= nv) *reinterpret_cast<uint32_t*>(yp + 2 * d) = pack_bf16x2(c8[2 * d], c8[2 * d + 1]);
            }
          }
        }
      }
    STAMP_ADD(4);
    __syncthreads();   // B3: the x tile (residual reads) and h are free: buffer b may take tile k + 2
    issue_tile(tile + 2 * grid, b);
    STAMP_ADD(5);
  }
  if (RDST_DBGV(p.stamps) && tid == 0)
    for (int k = 0; k < 8; ++k) p.stamps[(size_t)blockIdx.x * 8 + k] = tacc[k];
#undef STAMP_ADD
}

template <int C>
int launch_m3(M3Args& p, hipStream_t st) {
  using CF = M3Cfg<C>;
  p.ntiles = (p.M + M3_TT - 1) / M3_TT;
  int grid = (p.ntiles + 1) / 2;             // at least two tiles per workgroup where there are enough
  if (grid > 256 * CF::WGCU) grid = 256 * CF::WGCU;
  if (grid < 1) grid = 1;
  auto kern = mlp3_fwd_kernel<C>;
  // (per launch: the attribute is per DEVICE, a process-wide "done" flag would leave a second GPU without it)
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  p.stamps = rdst_stamps_begin("RDST_M3_STAMPS", grid, 8, st);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CF::NTHR), CF::SMEM, st, p);
  rdst_stamps_end("mlp3_fwd: 0 wait+B0, 1 stats, 2 phase 1, 3 wait+B2, 4 phase 2, 5 B3+issue", p.stamps, grid, 8, st);
  return rdst_launch_status("mlp3_fwd");
}

}  // namespace

size_t mlp3_pack_bytes(int C, int hid) { return lin3_pack_bytes(C, hid) + lin3_pack_bytes(hid, C); }

// packs: [fc1 image: lin3 layout for (N = hid, K = C, gamma / beta folded)][fc2 image: (N = C, K = hid)]
int mlp3_fwd_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* W1, const float* b1, const float* W2,
                  const float* b2, bf16* Y, int64_t ldy, float* stats, int64_t M, int C, int hid, void* wpack, bool prepacked,
                  hipStream_t st) {
  if (!wpack || ((uintptr_t)wpack & 15) || hid != 2 * C || !(C == 60 || C == 90 || C == 120) || M <= 0) return RDST_ENOTSUP;
  if (((uintptr_t)X & 3) || (ldx & 1) || ((uintptr_t)Y & 3) || (ldy & 1)) return RDST_ENOTSUP;
  const int64_t xb = ((M - 1) * ldx + C) * 2;
  if (xb >= (1ll << 31) || M * ldy * 2 >= (1ll << 31)) return RDST_ENOTSUP;
  char* base = reinterpret_cast<char*>(wpack);
  char* base2 = base + lin3_pack_bytes(C, hid);
  if (!prepacked) {
    if (int rc = lin3_pack_launch(W1, ln_w, ln_b, b1, base, hid, C, 1.0f, st)) return rc;
    if (int rc = lin3_pack_launch(W2, nullptr, nullptr, b2, base2, C, hid, 1.0f, st)) return rc;
  }
  const int nt1 = (hid + 31) / 32, ks1 = (C + 15) / 16, nt2 = (C + 31) / 32, ks2 = (hid + 15) / 16;
  M3Args p{};
  p.X = X; p.ldx = ldx; p.x_bytes = (int)xb;
  p.W1p = reinterpret_cast<const bf16*>(base); p.sb1 = reinterpret_cast<const float*>(base + (size_t)nt1 * ks1 * 1024);
  p.W2p = reinterpret_cast<const bf16*>(base2); p.sb2 = reinterpret_cast<const float*>(base2 + (size_t)nt2 * ks2 * 1024);
  p.Y = Y; p.ldy = ldy; p.stats = stats; p.M = (int)M;
  if (C == 60) return launch_m3<60>(p, st);
  if (C == 90) return launch_m3<90>(p, st);
  return launch_m3<120>(p, st);
}
