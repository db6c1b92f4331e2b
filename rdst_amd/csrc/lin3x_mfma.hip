// K3 forward in the RDST_F32X3 arithmetic as an HBM streaming kernel: y = (LayerNorm(x) | GELU(x) | x) @ W^T + b (* scale) (+ residual) on fp32 ROWS,
// the matrix products on the bf16 matrix cores with every operand as two bf16 terms (hi = bf16(v), lo = bf16(v - hi)):
//     a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi          (the dropped a_lo b_lo is <= 2^-18 |a b|, below the 2^-17 of the representation)
// i.e. THREE v_mfma_f32_32x32x16_bf16 per 16 k where the exact mode issues eight v_mfma_f32_32x32x2_f32.  It is lin3_mfma.hip
// (bf16 rows) rebuilt for 4-byte rows — the round-5 split kernels (linear_mfma.hip, `SP`) re-staged and re-split the fp32 weights in
// every workgroup and split every activation fragment once per wave that used it:
//   * weights live in REGISTERS for the whole kernel as PREPACKED hi / lo fragment pairs (pack.h: lin3x_pack_block; LayerNorm gamma,
//     the output scale folded in, b' = (b + W beta) s beside them): a wave owns ONE output tile (8 registers per 16 k);
//   * a workgroup takes 32..128-token tiles of RAW fp32 rows into LDS by LDS-DMA (buffer_load ... lds; rows past M and the pad
//     slots arrive as zeros), two tiles deep, counted vmcnt waits; the residual rows (proj / fc2) come the same way;
//   * ONE conversion pass per tile, shared by all waves: LPT lanes per token read the raw row, form (mean, rstd) by a two-pass sum in
//     registers (written out for the backward), normalise (LayerNorm) or apply the exact-to-1.5e-7 GELU (fc2 reads fc1's
//     pre-activation), split into hi / lo and write the row back IN PLACE as 64-byte groups [8 hi | 8 hi | 8 lo | 8 lo] of 16 k:
//     the B fragments of a k-step are then two ds_read_b128 at the lane's row + immediate (odd number of 16-byte slots per row);
//   * accumulators transposed (output channel in the registers, token on the lane); bias / residual as 16-byte LDS reads, one
//     v_permlane32_swap per register pair, 16-byte fp32 row stores.
// Shapes: K in {60, 90, 120} with N = 3K (norm1 + qkv), N = K + residual (proj), N = 30 (dense tails), N = 2K (norm2 + fc1), and
// K = 2C, N = C with GELU on the way in + residual (fc2).  Everything else stays on linear_mfma.hip.
#include "linear.h"
#include "mfma.h"
#include "pack.h"

#ifndef L3X_TWO_SHAPE
#define L3X_TWO_SHAPE(K, N, RES) (!(RES) && (N) == 2 * (K))   // shapes that run TWO workgroups per CU on half-size tiles (see L3X::TWO):
// norm2 + fc1.  A/B on one box, cold us per call: 26.5 -> 25.1 (C = 60), 42.5 -> 41.0 (90), 56.7 -> 50.6 (120); norm1 + qkv at C = 60 and the
// dense tails at C = 60 are unchanged, the tail at C = 120 is slower (22.4 -> 28.2), N > 256 would need more than 16 waves per CU
#endif
#ifndef L3X_ABL
#define L3X_ABL 0   // compile-time ablations (tools/abl_build.sh): 1 no MFMAs, 2 no conversion pass, 4 no global stores, 8 no tile DMA
#endif

// -DL3X_STAMPS: workgroup 0 prints, per wave, the clock64() ticks it spent in each phase of the tile loop (tools/abl_build.sh)
#ifdef L3X_STAMPS
#define L3X_T0 long long tk_[6] = {0, 0, 0, 0, 0, 0}, tl_ = clock64();
#define L3X_T(i) { const long long n_ = clock64(); tk_[i] += n_ - tl_; tl_ = n_; }
#define L3X_TP if (blockIdx.x == 0 && lane == 0) printf("wave %d: issue %lld  wait %lld  barrier1 %lld  convert %lld  barrier2 %lld  products+stores %lld\n", wave, tk_[0], tk_[1], tk_[2], tk_[3], tk_[4], tk_[5]);
#else
#define L3X_T0
#define L3X_T(i)
#define L3X_TP
#endif

namespace {

constexpr int X3_PLAIN = 0, X3_LN = 1, X3_GELU = 2;

struct L3XArgs {
  const float* X; int64_t ldx; int x_bytes;
  const uint32_t* Wp; const float* bp;
  const float* R; int64_t ldr; int r_bytes;
  float* Y; int64_t ldy;
  float* stats;
  int M, N, ntiles;
};

constexpr int l3x_pow2floor(int v) { return v >= 16 ? 16 : v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : 1; }
constexpr int l3x_pow2ceil(int v) { return v > 8 ? 16 : v > 4 ? 8 : v > 2 ? 4 : v > 1 ? 2 : 1; }
constexpr int l3x_min(int a, int b) { return a < b ? a : b; }
// compute waves: a multiple of NT (a wave keeps ONE output tile's fragments), at most 12, dividing the items evenly where one does
constexpr int l3x_nwc(int NT, int items, int cap = 12) {
  for (int m = (cap / NT) * NT; m >= NT; m -= NT)
    if (m <= items && items % m == 0) return m;
  return NT;
}

template <int K, int N, int MODE, bool RES>
struct L3X {
  static constexpr int KS = (K + 15) / 16, NT = (N + 31) / 32;
  static constexpr int XSLOTS = 4 * KS + 1, XS = 16 * XSLOTS, XD = (4 * K + 15) / 16;     // row: KS groups of 64 B + one pad slot
  static constexpr int RSLOTS = RES ? 8 * NT + 1 : 0, RS = 16 * RSLOTS, RD = (4 * N + 15) / 16;
  static constexpr int NBUF = 2;
  static constexpr int BROWS = 8, BS = 144, BNCB = BROWS * BS;                              // (no residual) a wave's bounce image: 8 rows x 32 floats
  static constexpr int tileb(int tt) { return (tt * XSLOTS + 63) / 64 * 1024 + (RES ? (tt * RSLOTS + 63) / 64 * 1024 : 0); }
  // tokens per tile: as many as two buffers fit (one tile ahead in flight = 30..68 KB per CU)
  // TWO: two workgroups per CU, each with half the LDS — the conversion pass (vector ALU) of one runs beside the products (matrix pipe)
  // and stores of the other; inside ONE workgroup the two phases are separated by barriers and never overlap
  static constexpr bool TWO = L3X_TWO_SHAPE(K, N, RES);
  static constexpr int LIM = TWO ? 62 * 1024 : (RES ? 156 : 144) * 1024;   // (without a residual tile the waves' bounce images come on top)
  static constexpr int TT = NBUF * tileb(128) <= LIM ? 128 : NBUF * tileb(64) <= LIM ? 64 : 32;
  static constexpr int SUB = TT / 32, ITEMS = NT * SUB;
  // waves: NWC compute waves, filled up to 8 with helpers (conversion pass, DMA issue)
  static constexpr int NWC = l3x_nwc(NT, ITEMS, TWO ? 8 : 12);
  static constexpr int NW = NWC < 8 ? 8 : NWC, NTHR = 64 * NW;
  static constexpr int NJ = (ITEMS + NWC - 1) / NWC;
  static constexpr int TP = (TT * XSLOTS + 63) / 64, TILEB = TP * 1024;
  static constexpr int RP = RES ? (TT * RSLOTS + 63) / 64 : 0, RTILEB = RP * 1024;
  static constexpr int CNT = (TP + NW - 1) / NW + (RES ? (RP + NW - 1) / NW : 0);          // DMA pieces per wave and tile
  static constexpr int R_OFF = NBUF * TILEB, P_OFF = R_OFF + NBUF * RTILEB;
  static constexpr int PB = RES ? 0 : (NWC * BNCB + 1023) / 1024 * 1024;
  static constexpr int B_OFF = P_OFF + PB;
  static constexpr int BB = (NT * 32 * 4 + 1023) / 1024 * 1024;
  static constexpr int SMEM = B_OFF + BB;
  static constexpr int LPT = l3x_min(l3x_pow2floor(NTHR / TT), l3x_pow2ceil(KS));         // lanes per token in the conversion pass
  static constexpr int NG = (KS + LPT - 1) / LPT;                                          // 64-byte groups per lane
  static constexpr int WPS = (NW + 3) / 4 * (TWO ? 2 : 1);
  static constexpr int NST = (N % 4 == 0 && ITEMS % NWC == 0 && CNT + 4 * NJ < 64) ? 4 * NJ : 0;   // stores per compute wave and tile, where exact
  static_assert(NWC % NT == 0 && NWC <= 12, "a wave keeps one output tile");
  static_assert(SMEM <= 160 * 1024 / (TWO ? 2 : 1), "LDS");
  static_assert(TP >= NW && (!RES || RP >= NW), "every wave owns at least one piece of a tile");
  static_assert(CNT < 64, "vmcnt is a 6-bit counter");
  static_assert(LPT >= 1 && LPT <= 16 && NTHR % LPT == 0, "lanes per token");
};

// 8 floats -> 8 bf16 hi (4 words) + 8 bf16 lo (4 words)
__device__ __forceinline__ void split8(const float* f, Pack16& hi, Pack16& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi.w[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
    lo.w[e] = pack_bf16x2(f[2 * e] - bf16lo(hi.w[e]), f[2 * e + 1] - bf16hi(hi.w[e]));
  }
}

template <int K, int N, int MODE, bool RES>
__global__ void __launch_bounds__((L3X<K, N, MODE, RES>::NTHR), (L3X<K, N, MODE, RES>::WPS)) lin3x_kernel(const L3XArgs p) {
  using CF = L3X<K, N, MODE, RES>;
  constexpr int KS = CF::KS, NT = CF::NT, XS = CF::XS, NW = CF::NW, NWC = CF::NWC, NTHR = CF::NTHR, NJ = CF::NJ, CNT = CF::CNT;
  constexpr int LPT = CF::LPT, NG = CF::NG, TT = CF::TT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* bL = reinterpret_cast<const float*>(smem + CF::B_OFF);

  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void* ptr, uint32_t bytes) {
    u32x4s_t q;
    q.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    q.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32) & 0xffffu);
    q.z = __builtin_amdgcn_readfirstlane(bytes);
    q.w = 0x00020000u;
    return q;
  };
  const u32x4s_t rsx = make_rsrc(p.X, (uint32_t)p.x_bytes), rsr = make_rsrc(RES ? (const void*)p.R : (const void*)p.X, (uint32_t)p.r_bytes),
                 rsb = make_rsrc(p.bp, NT * 32 * 4);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const u32x4s_t& rs, uint32_t ldst, int off) {   // inline asm: see conv3_mfma.hip
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rs) : "memory");
  };
  const int grid = gridDim.x;
  // ---- this wave's weight fragments (hi, lo per 16 k) of output tile nt = wave % NT: inline-asm loads the compiler does not see, issued
  // before the tiles (memory operations retire in issue order: every later tile wait covers them).  The kernel must not spill.
  typedef uint32_t u32x4v_t __attribute__((ext_vector_type(4)));
  u32x4v_t whi[KS], wlo[KS];
  const int nt = wave % NT;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const char* src = reinterpret_cast<const char*>(p.Wp) + (((int64_t)nt * KS + ks) * 128 + lane) * 16;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(whi[ks]) : "v"(src) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(wlo[ks]) : "v"(src) : "memory");
  }
  {  // b' (NT x 32 floats) by LDS-DMA: piece wave % NBP (duplicates write the same bytes)
    constexpr int NBP = CF::BB / 1024;
    const int pc = wave % NBP;
    dma(rsb, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::B_OFF + pc * 1024)), pc * 1024 + lane * 16);
  }
  // Every wave issues exactly CNT pieces per tile (a wave whose share is one short repeats its last piece; tiles past the end are
  // all-zero pieces): "the tile issued one iteration ago has landed" is the counted wait vmcnt(CNT) behind the NEXT tile's issue.
  auto issue_tile = [&](int tile, int b) {
    if (L3X_ABL & 8) tile = p.ntiles;   // (all-zero pieces: the instruction count stays, no bytes move)
#pragma unroll
    for (int i = 0; i < (CF::TP + NW - 1) / NW; ++i) {
      int q = wave + NW * i;
      q = q < CF::TP ? q : q - NW;
      const int sidx = q * 64 + lane;
      const int tok = sidx / CF::XSLOTS, sl = sidx - tok * CF::XSLOTS;
      const int grow = tile * TT + tok;
      const bool ok = tile < p.ntiles && tok < TT && sl < CF::XD && grow < p.M;
      dma(rsx, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(b * CF::TILEB + q * 1024)),
          ok ? grow * ((int)p.ldx * 4) + sl * 16 : p.x_bytes);   // (extents < 2^31 bytes)
    }
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < (CF::RP + NW - 1) / NW; ++i) {
        int q = wave + NW * i;
        q = q < CF::RP ? q : q - NW;
        const int sidx = q * 64 + lane;
        const int tok = sidx / CF::RSLOTS, sl = sidx - tok * CF::RSLOTS;
        const int grow = tile * TT + tok;
        const bool ok = tile < p.ntiles && tok < TT && sl < CF::RD && grow < p.M;
        dma(rsr, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::R_OFF + b * CF::RTILEB + q * 1024)),
            ok ? grow * ((int)p.ldr * 4) + sl * 16 : p.r_bytes);
      }
    }
  };

  int b = 0;
  if ((int)blockIdx.x < p.ntiles) issue_tile(blockIdx.x, 0);
  L3X_T0
  for (int tile = blockIdx.x; tile < p.ntiles; tile += grid, b ^= 1) {
    issue_tile(tile + grid, b ^ 1);   // (buffer b ^ 1: everybody left it at the barrier that ended the previous iteration)
    L3X_T(0)
    // Wait for THIS tile (issued one iteration ago), not for what was issued since: the CNT pieces of the next tile and — the point —
    // the previous tile's output stores, which then drain beside this tile's conversion and products instead of in front of them
    // (a plain vmcnt(CNT) waits for them: memory operations complete in issue order).  The count must never exceed what really was
    // issued: a compute wave issues exactly 4 stores per item when N % 4 == 0 (no ragged chunk) and all rows lie below M — every
    // tile but the globally last one, which is the last of its workgroup; the first iteration has no stores behind it.
    if (CF::NST > 0 && wave < NWC && tile != (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT + CF::NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(whi[ks]), "+v"(wlo[ks]));   // every use of a fragment is behind a wait
    L3X_T(1)
    __syncthreads();
    L3X_T(2)
    char* tb = smem + b * CF::TILEB;
    char* rb = smem + CF::R_OFF + b * CF::RTILEB;
    // ---- conversion pass: raw fp32 row -> (LayerNorm | GELU | as it is) -> [8 hi | 8 hi | 8 lo | 8 lo] per 16 k, in place ----
    for (int tok = tid / LPT; tok < ((L3X_ABL & 2) ? 0 : TT); tok += NTHR / LPT) {
      const int part = tid % LPT;
      char* row = tb + tok * XS;
      float xv[NG][16];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int g = part + LPT * i;
        const bool gon = g < KS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(row + (gon ? g : 0) * 64 + q * 16);
          const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (columns past K: the tail of the row's last 16-byte slot is the NEXT channels of the same memory row — a dense-buffer
            // slice, possibly not written yet, any bit pattern: select, do not multiply)
            const bool valid = gon && g * 16 + q * 4 + e < K;
            xv[i][q * 4 + e] = valid ? e4[e] : 0.f;
            sum += xv[i][q * 4 + e];
          }
        }
      }
      if constexpr (MODE == X3_LN) {
#pragma unroll
        for (int o = 1; o < LPT; o <<= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * (1.0f / K);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
          const int g = part + LPT * i;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const bool valid = g < KS && g * 16 + e < K;
            const float d = xv[i][e] - mean;
            sq = valid ? fmaf(d, d, sq) : sq;
          }
        }
#pragma unroll
        for (int o = 1; o < LPT; o <<= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq * (1.0f / K) + 1e-5f);
        if (part == 0) {
          const int grow = tile * TT + tok;
          if (grow < p.M) *reinterpret_cast<float2*>(p.stats + (int64_t)grow * 2) = make_float2(mean, rstd);
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
          const int g = part + LPT * i;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const bool valid = g < KS && g * 16 + e < K;
            xv[i][e] = valid ? (xv[i][e] - mean) * rstd : 0.f;
          }
        }
      } else if constexpr (MODE == X3_GELU) {
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) xv[i][e] = gelu_fast(xv[i][e]);   // (GELU(0) = 0: the masked columns stay zero)
      }
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int g = part + LPT * i;
        if (g < KS) {
          Pack16 h0, l0, h1, l1;
          split8(&xv[i][0], h0, l0);
          split8(&xv[i][8], h1, l1);
          Pack16* dst = reinterpret_cast<Pack16*>(row + g * 64);
          dst[0] = h0; dst[1] = h1; dst[2] = l0; dst[3] = l1;
        }
      }
    }
    L3X_T(3)
    __syncthreads();
    L3X_T(4)
    // ---- items of this wave: output tile nt, token sub-tiles tt = (wave + NWC j) / NT ----
    if (wave < NWC) {
#pragma unroll 1
      for (int j = 0; j < NJ; ++j) {
        const int item = wave + NWC * j;
        if (item >= CF::ITEMS) break;                         // wave-uniform
        const int tt = item / NT;
        const int tok = tt * 32 + r;
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const char* brow = tb + tok * XS + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const Pack16 bh = *reinterpret_cast<const Pack16*>(brow + ks * 64);
          const Pack16 bl = *reinterpret_cast<const Pack16*>(brow + ks * 64 + 32);
#if !(L3X_ABL & 1)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wlo[ks]), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, whi[ks]), __builtin_bit_cast(bf16x8_t, bl), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, whi[ks]), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
#else
          acc[ks & 15] += __uint_as_float(bh.w[0] ^ bl.w[1] ^ whi[ks].x ^ wlo[ks].y);
#endif
        }
        // Epilogue.  Register group g4 of lane (r, h) holds outputs n = 32 nt + 8 g4 + 4 h + (0..3) of token r; stored from there
        // (16 bytes of a token per lane, 32 rows per store instruction) the output cost as much as the rest of the kernel together.
        // So the tile goes through LDS and leaves row-wise, 8 rows x 128 contiguous bytes per store instruction:
        //   with a residual: its rows lie in the R tile — acc + b' is added IN PLACE (the item's 32 x 32 block belongs to this wave
        //   alone) and the block is read back by rows; without: through the wave's 8-row bounce image, four passes of 8 tokens.
        // LDS operations of one wave execute in order: no barrier.
        float o16[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 B4 = *reinterpret_cast<const float4*>(bL + nt * 32 + 8 * g4 + 4 * h);
          o16[4 * g4] = acc[4 * g4] + B4.x; o16[4 * g4 + 1] = acc[4 * g4 + 1] + B4.y;
          o16[4 * g4 + 2] = acc[4 * g4 + 2] + B4.z; o16[4 * g4 + 3] = acc[4 * g4 + 3] + B4.w;
        }
        const int row0 = tile * TT + tt * 32;
        const int crow = lane >> 3, cch = lane & 7, col = nt * 32 + 4 * cch;
        if constexpr (RES) {
          char* orow = rb + tok * CF::RS + (nt * 32 + 4 * h) * 4;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            float4 v = *reinterpret_cast<const float4*>(orow + 32 * g4);
            v.x += o16[4 * g4]; v.y += o16[4 * g4 + 1]; v.z += o16[4 * g4 + 2]; v.w += o16[4 * g4 + 3];
            *reinterpret_cast<float4*>(orow + 32 * g4) = v;
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int row = crow + 8 * ps, grow = row0 + row;
            const float4 v = *reinterpret_cast<const float4*>(rb + (tt * 32 + row) * CF::RS + col * 4);
            if (grow < p.M && !(L3X_ABL & 4)) {
              float* yp = p.Y + ((int64_t)grow * p.ldy + col);
              if (col + 4 <= N) {
                u32x4_a4 u;
                u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
                *reinterpret_cast<u32x4_a4*>(yp) = u;
              } else {
                if (col < N) yp[0] = v.x;
                if (col + 1 < N) yp[1] = v.y;
                if (col + 2 < N) yp[2] = v.z;
              }
            }
          }
        } else {
          char* bnc = smem + CF::P_OFF + wave * CF::BNCB;
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            if ((r >> 3) == ps) {   // the 8 tokens of this pass: their lanes (both halves) write their 16 values
              char* orow = bnc + (r & 7) * CF::BS + 16 * h;
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4*>(orow + 32 * g4) = make_float4(o16[4 * g4], o16[4 * g4 + 1], o16[4 * g4 + 2], o16[4 * g4 + 3]);
            }
            __builtin_amdgcn_wave_barrier();
            const int row = crow + 8 * ps, grow = row0 + row;
            const float4 v = *reinterpret_cast<const float4*>(bnc + crow * CF::BS + cch * 16);
            __builtin_amdgcn_wave_barrier();
            if (grow < p.M && !(L3X_ABL & 4)) {
              float* yp = p.Y + ((int64_t)grow * p.ldy + col);
              if (col + 4 <= N) {
                u32x4_a4 u;
                u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
                *reinterpret_cast<u32x4_a4*>(yp) = u;
              } else {
                if (col < N) yp[0] = v.x;
                if (col + 1 < N) yp[1] = v.y;
                if (col + 2 < N) yp[2] = v.z;
              }
            }
          }
        }
      }
    }
    L3X_T(5)
    __syncthreads();   // buffer b may be overwritten (the next iteration's issue targets it)
    L3X_T(2)
  }
  L3X_TP
}

template <int K, int N, int MODE, bool RES>
int launch_l3x(L3XArgs& p, hipStream_t st, const char* what) {
  using CF = L3X<K, N, MODE, RES>;
  p.ntiles = (p.M + CF::TT - 1) / CF::TT;
  const int cap = CF::TWO ? 512 : 256;
  int grid = p.ntiles < cap ? p.ntiles : cap;
  auto kern = lin3x_kernel<K, N, MODE, RES>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CF::NTHR), CF::SMEM, st, p);
  return rdst_launch_status(what);
}

__global__ void __launch_bounds__(256) lin3x_pack_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ bias,
                                                         uint32_t* __restrict__ wp, float* __restrict__ bp, int N, int K, int ksteps,
                                                         int ntiles, float s) {
  lin3x_pack_block((int)blockIdx.x, W, gamma, beta, bias, wp, bp, N, K, ksteps, ntiles, s);
}

}  // namespace

// 0 = not a covered shape; else the kind: 1 qkv (LN, N = 3K), 2 proj (residual, N = K), 3 dense tail (LN, N = 30), 4 fc1 (LN, N = 2K),
// 5 fc2 (GELU on the way in, residual, K = 2N)
int lin3x_kind(int K, int N, bool ln, bool res, int in_act) {
  if (in_act == RDST_ACT_GELU) return (!ln && res && K == 2 * N && (N == 60 || N == 90 || N == 120)) ? 5 : 0;
  if (in_act) return 0;
  if (!(K == 60 || K == 90 || K == 120)) return 0;
  if (ln && !res && N == 3 * K) return 1;
  if (!ln && res && N == K) return 2;
  if (ln && !res && N == 30) return 3;
  if (ln && !res && N == 2 * K) return 4;
  return 0;
}

int lin3x_pack_launch(const float* W, const float* gamma, const float* beta, const float* bias, void* out, int N, int K, float s,
                      hipStream_t st) {
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  uint32_t* wp = reinterpret_cast<uint32_t*>(out);
  float* bp = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (size_t)nt * ks * 2048);
  hipLaunchKernelGGL(lin3x_pack_kernel, dim3((unsigned)lin3x_pack_blocks(K, N)), dim3(256), 0, st, W, gamma, beta, bias, wp, bp, N, K, ks, nt, s);
  return rdst_launch_status("lin3x_pack");
}

// RDST_ENOTSUP = not one of the covered shapes / alignments (the caller falls back to linear_mfma.hip)
int lin3x_fwd_f32(const float* X, int64_t ldx, const float* ln_w, const float* ln_b, int in_act, const float* Wt, const float* bias,
                  const float* R, int64_t ldr, float* Y, int64_t ldy, float* stats, int64_t M, int K, int N, float s, void* wpack,
                  bool prepacked, hipStream_t st) {
  if (!wpack || ((uintptr_t)wpack & 15) || !Wt || M <= 0) return RDST_ENOTSUP;
  const bool ln = ln_w != nullptr;
  const int kind = lin3x_kind(K, N, ln, R != nullptr, in_act);
  if (!kind) return RDST_ENOTSUP;
  if (((uintptr_t)X & 3) || ((uintptr_t)Y & 3) || (R && ((uintptr_t)R & 3))) return RDST_ENOTSUP;
  const int64_t xb = ((M - 1) * ldx + K) * 4;
  if (xb >= (1ll << 31) || M * ldy * 4 >= (1ll << 31) || (R && M * ldr * 4 >= (1ll << 31))) return RDST_ENOTSUP;
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  if (!prepacked)
    if (int rc = lin3x_pack_launch(Wt, ln_w, ln_b, bias, wpack, N, K, s, st)) return rc;
  L3XArgs p{};
  p.X = X; p.ldx = ldx; p.x_bytes = (int)xb; p.Wp = reinterpret_cast<const uint32_t*>(wpack);
  p.bp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + (size_t)nt * ks * 2048);
  p.R = R; p.ldr = ldr; p.r_bytes = R ? (int)(((M - 1) * ldr + N) * 4) : 0; p.Y = Y; p.ldy = ldy; p.stats = stats;
  p.M = (int)M; p.N = N;
#define L3X_CASE(CC)                                                                                          \
  if (kind == 1 && K == CC) return launch_l3x<CC, 3 * CC, X3_LN, false>(p, st, "lin3x_qkv");                  \
  if (kind == 2 && K == CC) return launch_l3x<CC, CC, X3_PLAIN, true>(p, st, "lin3x_proj");                   \
  if (kind == 3 && K == CC) return launch_l3x<CC, 30, X3_LN, false>(p, st, "lin3x_tail");                     \
  if (kind == 4 && K == CC) return launch_l3x<CC, 2 * CC, X3_LN, false>(p, st, "lin3x_fc1");                  \
  if (kind == 5 && N == CC) return launch_l3x<2 * CC, CC, X3_GELU, true>(p, st, "lin3x_fc2");
  L3X_CASE(60) L3X_CASE(90) L3X_CASE(120)
#undef L3X_CASE
  return RDST_ENOTSUP;
}
