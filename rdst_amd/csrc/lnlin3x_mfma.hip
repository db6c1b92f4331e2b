// Backward of a (LayerNorm- / GELU-fronted) Linear in ONE pass over (x, dY) in the RDST_F32X3 arithmetic: fp32 rows in HBM, every
// matrix operand as two bf16 terms (hi = bf16(v), lo = bf16(v - hi)), three v_mfma_f32_32x32x16_bf16 per product
// (hi.hi + hi.lo + lo.hi; the dropped lo.lo is <= 2^-18 relative).  It is lnlin3_mfma.hip (bf16 rows) rebuilt for 4-byte rows; the
// round-5 split mode ran three launches per Linear (lin_wgrad_mfma + lin_dgrad_ln2, the latter twice for the wide shapes), each
// re-reading x and dY and re-splitting the weights per workgroup:
//   * waves are SPECIALISED: weight-gradient waves own TN x TC accumulator tiles of G = dY^T.X' for the whole kernel (X' = x-hat
//     behind a LayerNorm, x for a plain Linear, GELU(x) for fc2 whose input is fc1's pre-activation; column K of X' is a ones
//     column: d(bias)); data-gradient wave d owns DT channel tiles of dA^T = (W gamma)^T.dY^T and keeps its (W gamma)^T fragments —
//     hi AND lo — in registers for the whole kernel;
//   * the RAW fp32 rows of a 32-token tile (dY, x, the forward's statistics) come by LDS-DMA one tile ahead (two raw buffers, counted
//     vmcnt waits: no wave carries prefetch registers — beside 120-190 registers of fragments or accumulators they spilled); ONE
//     conversion pass per tile, shared by all waves, normalises (LayerNorm) / applies the 1.5e-7 GELU (fc2), splits, and writes two
//     bf16 PLANES per operand ([token][channel], the layouts of the bf16 kernel): the weight-gradient operands are
//     ds_read_b64_tr_b16 of either plane, the data-gradient B operand is ds_read_b128 of the dY planes; two barriers per tile;
//   * LayerNorm backward in the accumulators (the two row sums need all channel tiles: partial sums through LDS, rows finished
//     behind the NEXT tile's first barrier — a third barrier per tile measured 1.3x slower), addends and stores in the lane's own
//     8-channel runs; WITHOUT a LayerNorm dA leaves through a per-wave LDS bounce image and is finished ROW-WISE (8 lanes on a row's 128
//     contiguous bytes): dX_add / GELU'(pre-activation) from HBM in the same row-wise chunks, issued ahead of the products,
//     16-byte dX stores (27 of 72 us at C = 120 proj went into dX stored from the accumulators: 64 pieces of 16 bytes in 32 rows);
//   * per-workgroup fp32 slabs G [N][K+1], summed in fixed order (reduce_batch.h; behind a LayerNorm the finish kernel forms
//     dW = gamma G + beta db^T, d(gamma), d(beta) from the summed G).
// Where (W gamma)^T hi + lo and the G tiles do not fit the register file together (norm1 + qkv at C = 120: 736 + 768 registers
// per lane) the host splits N in two launches over halves of the output features — LayerNorm backward is linear in dA — the
// second accumulating onto the first's dX in place.
#include "linear.h"
#include "mfma.h"
#include "reduce_batch.h"
#include "wattn_hd.h"

#ifndef LBX_ABL
#define LBX_ABL 0   // compile-time ablations (tools/abl_build.sh): 1 no weight-gradient MFMAs, 2 no data-gradient MFMAs, 4 no dX stores, 8 no slab dump, 16 no conversion pass, 32 no tile DMA (zero pieces)
#endif

// -DLBX_STAMPS: workgroup 0 prints, per wave, the clock64() ticks it spent in each phase of the tile loop (tools/abl_build.sh)
#ifdef LBX_STAMPS
#define LBX_T0 long long tk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = clock64();
#define LBX_T(i) { const long long n_ = clock64(); tk_[i] += n_ - tl_; tl_ = n_; }
#define LBX_TP(role) if (blockIdx.x == 0 && lane == 0) printf("wave %d %s: issue+wait %lld  barrierA %lld  finish_ln %lld  convert %lld  barrierB %lld  compute %lld (product %lld, rest %lld)\n", wave, role, tk_[0], tk_[1], tk_[2], tk_[3], tk_[4], tk_[5] + tk_[6] + tk_[7], tk_[6], tk_[7]);
#else
#define LBX_T0
#define LBX_T(i)
#define LBX_TP(role)
#endif

namespace {
using namespace wahd;
constexpr int BX_PLAIN = 0, BX_LN = 1, BX_GELU = 2;

struct LBXArgs {
  const float* X; int64_t ldx; const float* stats; const float* lnw; const float* W;
  const float* dY; int64_t lddy; float* dX; int64_t lddx; const float* Acc; int64_t ldacc;
  const float* Acc2; int64_t ldacc2;   // second addend of dX
  float* slab; int64_t slab_stride;    // floats per workgroup
  int64_t M; int64_t ntiles; int tiles_per_wg;
  int x_bytes, y_bytes;                // extents of X / dY (32-bit DMA offsets)
};

__host__ __device__ constexpr int lbx_ld(int cols) {   // plane row stride (bytes): odd 16-B slot count, not 16..47 (mod 256)
  const int b = cols * 2 + 16;
  return (b & 255) < 48 ? b + 64 : b;
}

// DT channel tiles per data-gradient wave, TN x TC accumulator tiles per weight-gradient wave
template <int K_, int N_, int MODE_, int DT_, int TN_, int TC_>
struct LBX {
  static constexpr int K = K_, N = N_, MODE = MODE_, DT = DT_, TN = TN_, TC = TC_;
  static constexpr int NCT = (K + 1 + 31) / 32, NW = (N + 31) / 32, KN = (N + 15) / 16;
  static_assert(NW % TN == 0 && NCT % TC == 0 && NCT % DT == 0, "blocking");
  static constexpr int NGN = NW / TN, NGC = NCT / TC, NWG = NGN * NGC;   // weight-gradient waves
  static constexpr int NDG = NCT / DT;                                    // data-gradient waves
  static constexpr int NWV = NWG + NDG, NT = 64 * NWV;
  static constexpr int CKX = (K + 3) / 4, CKY = (N + 3) / 4;             // 4-float chunks per row
  static constexpr int NXC = 32 * CKX, NYC = 32 * CKY;
  // raw rows in LDS: 16-byte slots, a region = whole 1 KB DMA pieces
  static constexpr int RSX = (4 * K + 15) / 16 * 16, RSY = (4 * N + 15) / 16 * 16;
  static constexpr int TPX = (32 * RSX + 1023) / 1024, TPY = (32 * RSY + 1023) / 1024, TPS = MODE == BX_LN ? 1 : 0;
  static constexpr int TPT = TPY + TPX + TPS, CNT = (TPT + NWG - 1) / NWG;   // pieces per tile / per weight-gradient wave (they issue the DMA)
  // the conversion pass belongs to the weight-gradient waves too (the data-gradient waves are the workgroup's critical path): TPR threads
  // on one token row, thread t of a row takes its 4-float chunks t, t + TPR, ... (dY's CKY chunks first, then x's CKX)
  static constexpr int NCV = 64 * NWG, TPR = NCV / 32, CKR = CKY + CKX, NJ = (CKR + TPR - 1) / TPR;
  static constexpr int RAW_X = TPY * 1024, RAW_S = RAW_X + TPX * 1024, RAWB = RAW_S + TPS * 1024;
  static constexpr int CP = 32 * NCT;
  static constexpr int LDX = lbx_ld(CP), LDY = lbx_ld(32 * NW);
  static constexpr int OFF_XL = 32 * LDX, OFF_YH = 64 * LDX, OFF_YL = OFF_YH + 32 * LDY, OFF_SM = OFF_YL + 32 * LDY, PLB = OFF_SM + 128;   // the planes (one set) + rstd per row
  static constexpr int OFF_RED = (PLB + 15) / 16 * 16;
  // a data-gradient wave's bounce image: BR token rows x 32 floats (+ pad): dA leaves the registers (channel / lane = token) through it
  // and is finished ROW-WISE — 8 lanes on the 128 contiguous bytes of a row: the addend loads and the dX stores of an instruction
  // touch 8 rows, not 64 16-byte pieces in 32 rows (stored from the registers dX cost 27 of 72 us at C = 120 proj).  As many rows as fit
  static constexpr int BSTR = 144;
  static constexpr int OFF_BNC = OFF_RED + 2 * NCT * 256;
  static constexpr int bnc_fit(int br) { return (OFF_BNC + NDG * br * BSTR + 1023) / 1024 * 1024 + 2 * RAWB <= 160 * 1024; }
  static constexpr int BR = bnc_fit(32) ? 32 : bnc_fit(16) ? 16 : 8;
  static constexpr int OFF_RAW = (OFF_BNC + NDG * BR * BSTR + 1023) / 1024 * 1024;
  static constexpr int SMEM = OFF_RAW + 2 * RAWB;
  static constexpr int WPS = (NWV + 3) / 4;
  static constexpr int CVB = TN * TC * 16 < 128 ? 8 : 4;   // chunks of the conversion pass in flight per thread
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(CNT < 64, "pieces per wave");
  static_assert((K % 4 == 0 || K % 4 == 2) && (N % 4 == 0 || N % 4 == 2), "a row's last chunk holds 4 or 2 floats");
  static_assert(MODE != BX_LN || DT == 1, "LayerNorm backward: one channel tile per data-gradient wave");
};

// hi / lo bf16 pairs of 4 floats: (hi.x | hi.y) and (lo.x | lo.y)
__device__ __forceinline__ void split4(const float* f, u32x2_t& hi, u32x2_t& lo) {
  hi.x = pack_bf16x2(f[0], f[1]);
  hi.y = pack_bf16x2(f[2], f[3]);
  lo.x = pack_bf16x2(f[0] - bf16lo(hi.x), f[1] - bf16hi(hi.x));
  lo.y = pack_bf16x2(f[2] - bf16lo(hi.y), f[3] - bf16hi(hi.y));
}
__device__ __forceinline__ void put8(char* dst, const u32x2_t& v) {   // 8 bytes, 4-byte aligned at least
  if ((reinterpret_cast<uintptr_t>(dst) & 7) == 0) *reinterpret_cast<u32x2_t*>(dst) = v;
  else {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    d[0] = v.x; d[1] = v.y;
  }
}

template <class CF>
__global__ void __launch_bounds__((CF::NT), (CF::WPS)) lnlin3x_bwd_kernel(const LBXArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K = CF::K, N = CF::N, MODE = CF::MODE, DT = CF::DT, TN = CF::TN, TC = CF::TC;
  constexpr int NCT = CF::NCT, KN = CF::KN, NGC = CF::NGC, NWG = CF::NWG, NT = CF::NT, LDX = CF::LDX, LDY = CF::LDY;
  constexpr bool LN = MODE == BX_LN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;

  // ---- prologue: zero the planes (pad columns) and the exchange slots, ones column of X' (hi plane; the lo plane keeps its zero) ----
  lds_zero16(smem, CF::OFF_RAW, tid, NT);
  __syncthreads();
  if (tid < 32) *reinterpret_cast<uint16_t*>(smem + tid * LDX + K * 2) = 0x3f80;

  const int q = (lane & 15) >> 2, pp = lane & 3, gq1 = (lane >> 4) & 1;
  const int trc = (16 * gq1 + 4 * pp) * 2;
  const int tro_y = CF::OFF_YH + (8 * hh + q) * LDY + trc;   // transposed-read lane offsets (hi planes)
  const int tro_x = (8 * hh + q) * LDX + trc;

  const bool is_wg = wave < NWG;
  const int gn = wave / NGC, gc = wave - gn * NGC;   // weight-gradient block of this wave
  const int dwv = wave - NWG;                         // data-gradient wave index: channel tiles DT dwv .. DT dwv + DT - 1

  const int64_t t0 = (int64_t)blockIdx.x * p.tiles_per_wg;
  const int64_t t1 = t0 + p.tiles_per_wg < p.ntiles ? t0 + p.tiles_per_wg : p.ntiles;

  // ---- raw rows by LDS-DMA, issued by the weight-gradient waves (CNT 1 KB pieces of a tile each; rows past M and the tail of a region
  // arrive as zeros or are never read).  A lane's offset inside a tile is the same for every tile: computed once.
  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void* ptr, uint32_t bytes) {
    u32x4s_t v;
    v.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    v.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32) & 0xffffu);
    v.z = __builtin_amdgcn_readfirstlane(bytes);
    v.w = 0x00020000u;
    return v;
  };
  const u32x4s_t rsy = make_rsrc(p.dY, (uint32_t)p.y_bytes), rsx = make_rsrc(p.X, (uint32_t)p.x_bytes),
                 rss = make_rsrc(MODE == BX_LN ? (const void*)p.stats : (const void*)p.X, MODE == BX_LN ? (uint32_t)(p.M * 8) : 0u);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const u32x4s_t& rs0, uint32_t ldst, int off) {   // inline asm: see conv3_mfma.hip
    uint32_t keep;
    u32x4s_t rs;   // (scalar again at the point of use: under SGPR pressure the descriptor otherwise arrives in vector registers)
    rs.x = __builtin_amdgcn_readfirstlane(rs0.x); rs.y = __builtin_amdgcn_readfirstlane(rs0.y);
    rs.z = __builtin_amdgcn_readfirstlane(rs0.z); rs.w = __builtin_amdgcn_readfirstlane(rs0.w);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rs) : "memory");
  };
  int dmo[CF::CNT];
  if (is_wg) {
#pragma unroll
    for (int i = 0; i < CF::CNT; ++i) {
      const int pc = wave + NWG * i;   // wave-uniform
      int o = 0x40000000;              // (lanes past the 32 rows: the region's tail, never read)
      if (pc < CF::TPY) {
        const int sidx = pc * 64 + lane, row = sidx / (CF::RSY / 16), sl = sidx - row * (CF::RSY / 16);
        if (row < 32) o = row * (int)(p.lddy * 4) + sl * 16;
      } else if (pc < CF::TPY + CF::TPX) {
        const int sidx = (pc - CF::TPY) * 64 + lane, row = sidx / (CF::RSX / 16), sl = sidx - row * (CF::RSX / 16);
        if (row < 32) o = row * (int)(p.ldx * 4) + sl * 16;
      } else if (lane < 16) {
        o = lane * 16;                 // the tile's 32 (mean, rstd) pairs: two per lane
      }
      dmo[i] = o;
    }
  }
  auto issue_tile = [&](int64_t tile, int b) {
    if (tile >= t1 || (LBX_ABL & 32)) return;
    const int ty = (int)(tile * 32 * (p.lddy * 4)), tx = (int)(tile * 32 * (p.ldx * 4)), ts = (int)(tile * 256);
#pragma unroll
    for (int i = 0; i < CF::CNT; ++i) {
      const int pc = wave + NWG * i;
      if (pc >= CF::TPT) break;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::OFF_RAW + b * CF::RAWB + pc * 1024));
      if (pc < CF::TPY) dma(rsy, dst, dmo[i] + ty);
      else if (pc < CF::TPY + CF::TPX) dma(rsx, dst, dmo[i] + tx);
      else dma(rss, dst, dmo[i] + ts);
    }
  };
  // ---- the conversion pass of one tile (weight-gradient waves): raw chunk of 4 floats -> (x-hat | GELU | as it is) -> hi / lo into the
  // planes.  Every address is a per-thread base + a compile-time distance; CVB chunks in flight (one at a time the pass is a chain of
  // LDS round trips).  A row's last chunk may hold only 2 floats: the other two are forced to the planes' pad values (zeros; the ones
  // column of X' where it falls into that chunk) — what the raw slot holds there is somebody else's data.
  const int cvrow = tid / CF::TPR, cvt = tid - cvrow * CF::TPR;
  auto convert = [&](int b) {
    if (LBX_ABL & 16) return;
    constexpr int TPR = CF::TPR, CKY = CF::CKY, CKR = CF::CKR, NJ = CF::NJ, CB = CF::CVB;
    const char* raw = smem + CF::OFF_RAW + b * CF::RAWB;
    const char* sy = raw + cvrow * CF::RSY + 16 * cvt;
    const char* sx = raw + CF::RAW_X + cvrow * CF::RSX + 16 * (cvt - CKY);
    char* dy = smem + CF::OFF_YH + cvrow * LDY + 8 * cvt;
    char* dx = smem + cvrow * LDX + 8 * (cvt - CKY);
    float rs = 1.f, nm = 0.f;
    if (MODE == BX_LN) {
      const float2 s2 = *reinterpret_cast<const float2*>(raw + CF::RAW_S + cvrow * 8);
      rs = s2.y; nm = -s2.x * s2.y;
      if (cvt == 0) reinterpret_cast<float*>(smem + CF::OFF_SM)[cvrow] = s2.y;   // (kept beside the planes: the raw buffer is the
      // next-but-one tile's DMA target before the data-gradient waves read it)
    }
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += CB) {
      float4 v[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const int j = j0 + u;
        if (j >= NJ) continue;
        const bool ally = TPR * (j + 1) <= CKY, allx = TPR * j >= CKY;
        const int cidx = cvt + TPR * j;
        const bool isy = ally ? true : allx ? false : cidx < CKY;
        const bool on = TPR * (j + 1) <= CKR ? true : cidx < CKR;
        const char* src = (isy ? sy : sx) + 16 * TPR * j;
        if (TPR * (j + 1) > CKR) src = on ? src : sy;   // (a thread without a chunk in the last round reads something valid)
        v[u] = *reinterpret_cast<const float4*>(src);
      }
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const int j = j0 + u;
        if (j >= NJ) continue;
        const bool ally = TPR * (j + 1) <= CKY, allx = TPR * j >= CKY;
        const int cidx = cvt + TPR * j;
        const bool isy = ally ? true : allx ? false : cidx < CKY;
        const bool on = TPR * (j + 1) <= CKR ? true : cidx < CKR;
        float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        if (!ally) {
          float g[4];
          if (MODE == BX_LN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = fmaf(f[e], rs, nm);
          } else if (MODE == BX_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = gelu_fast(f[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = f[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = isy ? f[e] : g[e];
        }
        if (N % 4 == 2 && TPR * j <= CKY - 1 && CKY - 1 < TPR * (j + 1)) {
          if (cidx == CKY - 1) { f[2] = 0.f; f[3] = 0.f; }
        }
        if (K % 4 == 2 && TPR * j <= CKR - 1 && CKR - 1 < TPR * (j + 1)) {
          if (cidx == CKR - 1) { f[2] = 1.f; f[3] = 0.f; }
        }
        u32x2_t hi, lo;
        split4(f, hi, lo);
        if (on) {
          char* dst = (isy ? dy : dx) + 8 * TPR * j;
          *reinterpret_cast<u32x2_t*>(dst) = hi;
          *reinterpret_cast<u32x2_t*>(dst + (isy ? 32 * LDY : CF::OFF_XL)) = lo;
        }
      }
    }
  };
  if (is_wg) issue_tile(t0, 0);

  // The two roles run SEPARATE copies of the tile loop (same barrier count): in one loop body the register allocator would have
  // to keep the G tiles and the W fragments alive side by side in every wave.
  if (is_wg) {
    f32x16 G[TN][TC];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) G[i][j][v] = 0.f;
    int b = 0;
    LBX_T0
    for (int64_t tile = t0; tile < t1; ++tile, b ^= 1) {
      const char* buf = smem;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      LBX_T(0)
      __syncthreads();   // raw tile landed (every wave's pieces); everybody is done with the planes and with raw buffer b ^ 1
      LBX_T(1)
      issue_tile(tile + 1, b ^ 1);
      convert(b);
      LBX_T(3)
      __syncthreads();   // planes staged
      LBX_T(4)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Pack16 yh[TN], yl[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const lds_cp yp = (lds_cp)(buf + tro_y + 16 * s * LDY) + (gn * TN + i) * 64;
          yh[i] = lds_tr_pack(yp, yp + 4 * LDY);
          yl[i] = lds_tr_pack(yp + 32 * LDY, yp + 36 * LDY);
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {   // (the X' operands one channel tile at a time: 8 registers instead of 8 TC)
          const lds_cp xp = (lds_cp)(buf + tro_x + 16 * s * LDX) + (gc * TC + j) * 64;
          const Pack16 xh = lds_tr_pack(xp, xp + 4 * LDX);
          const Pack16 xl = lds_tr_pack(xp + CF::OFF_XL, xp + CF::OFF_XL + 4 * LDX);
#pragma unroll
          for (int i = 0; i < TN; ++i) {
#if !(LBX_ABL & 1)
            // rows = output features n, columns = channels (column K = d(bias))
            G[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, yl[i]), __builtin_bit_cast(bf16x8_t, xh), G[i][j], 0, 0, 0);
            G[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, yh[i]), __builtin_bit_cast(bf16x8_t, xl), G[i][j], 0, 0, 0);
            G[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, yh[i]), __builtin_bit_cast(bf16x8_t, xh), G[i][j], 0, 0, 0);
#else
            G[i][j][0] += __uint_as_float(yh[i].w[0] ^ xh.w[0] ^ yl[i].w[1] ^ xl.w[1]);
#endif
          }
        }
      }
      LBX_T(5)
    }
    LBX_TP("wgrad")
    if (LN) __syncthreads();   // (the data-gradient waves' closing barrier)
    // fp32 slab G [N][K+1] of this workgroup: consecutive lanes = consecutive channels of one row
    float* my = p.slab + (int64_t)blockIdx.x * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const int c = 32 * (gc * TC + j) + r;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int n = 32 * (gn * TN + i) + acc_row(v, hh);
          if (n < N && c <= K && !(LBX_ABL & 8)) my[(int64_t)n * (K + 1) + c] = G[i][j][v];
        }
      }
    return;
  }

  // ---- data-gradient wave: (W gamma)^T fragments hi / lo, lane (r, hh) of k-step ks holds W[16 ks + 8 hh + e][channel] gamma ----
  Pack16 wfh[DT][KN], wfl[DT][KN];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int c = 32 * (DT * dwv + dt) + r;
    const int cc = c < K ? c : K - 1;
    const float gm = (c < K) ? (LN ? p.lnw[cc] : 1.0f) : 0.f;
#pragma unroll
    for (int ks = 0; ks < KN; ++ks) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = 16 * ks + 8 * hh + e;
        const float w = p.W[(size_t)(n < N ? n : N - 1) * K + cc];
        f[e] = n < N ? w * gm : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wfh[dt][ks].w[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
        wfl[dt][ks].w[e] = pack_bf16x2(f[2 * e] - bf16lo(wfh[dt][ks].w[e]), f[2 * e + 1] - bf16hi(wfh[dt][ks].w[e]));
      }
    }
  }
  constexpr float invK = 1.0f / (float)K;
  constexpr int BR = CF::BR, BSTR = CF::BSTR;
  float* red = reinterpret_cast<float*>(smem + CF::OFF_RED);
  char* bnc = smem + CF::OFF_BNC + dwv * (BR * BSTR);
  const bool has_acc = p.Acc != nullptr, has_acc2 = p.Acc2 != nullptr;
  // row-wise layout of a 32 x 32 (token x channel) block: lane -> (row 8 k + lane / 8, k = 0..3; 4-channel chunk lane % 8)
  const int crow = lane >> 3, cch = lane & 7;
  auto load4 = [&](const float* base, int64_t ldb, int64_t grow, int col) {   // 4 floats of a row from HBM: ONE unconditional load
    // (a branch per chunk made the compiler wait for each of a tile's loads in turn).  Chunks past K re-read [K - 4, K) and are
    // dropped; with K % 4 == 2 the chunk at K - 2 is that read shifted by two
    static_assert(K % 4 == 0 || K % 4 == 2, "ragged chunk");
    const int64_t rr = grow < p.M ? grow : p.M - 1;
    const int c0 = col + 4 <= K ? col : K - 4;
    u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(base + rr * ldb + c0);
    if (K % 4 == 2 && col == K - 2) { v.x = v.z; v.y = v.w; }
    return v;
  };
  // the row-wise operands of one channel tile: dX_add (+ dX_add2), or fc2's pre-activation; issued ahead of the products
  auto load_addends = [&](int64_t row0, int ct, u32x4_a4 (&o)[4]) {
    const int col = 32 * ct + 4 * cch;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t grow = row0 + 8 * k + crow;
      if (has_acc) o[k] = load4(p.Acc, p.ldacc, grow, col);
      else if (has_acc2) o[k] = load4(p.Acc2, p.ldacc2, grow, col);
      if (MODE != BX_GELU && has_acc && has_acc2) {
        const u32x4_a4 a2 = load4(p.Acc2, p.ldacc2, grow, col);
        o[k].x = __float_as_uint(__uint_as_float(o[k].x) + __uint_as_float(a2.x));
        o[k].y = __float_as_uint(__uint_as_float(o[k].y) + __uint_as_float(a2.y));
        o[k].z = __float_as_uint(__uint_as_float(o[k].z) + __uint_as_float(a2.z));
        o[k].w = __float_as_uint(__uint_as_float(o[k].w) + __uint_as_float(a2.w));
      }
    }
  };
  // one channel tile of one token tile: values in the accumulator layout (registers: channel, lane: token) -> bounce -> row-wise:
  // + addends / x GELU'(pre-activation) -> dX rows
  auto finish = [&](const float (&o)[16], const u32x4_a4 (&adv)[4], int ct, int64_t row0) {
    const int col = 32 * ct + 4 * cch;
#pragma unroll
    for (int ps = 0; ps < 32 / BR; ++ps) {
      if (BR == 32 || (r / BR) == ps) {   // the tokens of this pass: their lanes (both halves) write their 16 values
        char* orow = bnc + (r % BR) * BSTR + 16 * hh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<float4*>(orow + 32 * g4) = make_float4(o[4 * g4], o[4 * g4 + 1], o[4 * g4 + 2], o[4 * g4 + 3]);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kk = 0; kk < BR / 8; ++kk) {
        const int k = ps * (BR / 8) + kk, trow = 8 * k + crow;
        const int64_t grow = row0 + trow;
        float4 v = *reinterpret_cast<const float4*>(bnc + (8 * kk + crow) * BSTR + cch * 16);
        const float a[4] = {__uint_as_float(adv[k].x), __uint_as_float(adv[k].y), __uint_as_float(adv[k].z), __uint_as_float(adv[k].w)};
        if (MODE == BX_GELU) {
          v.x *= a[0]; v.y *= a[1]; v.z *= a[2]; v.w *= a[3];
        } else if (has_acc || has_acc2) {
          v.x += a[0]; v.y += a[1]; v.z += a[2]; v.w += a[3];
        }
        if (grow < p.M && col < K && !(LBX_ABL & 4)) {
          float* dst = p.dX + grow * p.lddx + col;
          if (col + 4 <= K) {
            u32x4_a4 u;
            u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
            *reinterpret_cast<u32x4_a4*>(dst) = u;
          } else {
            dst[0] = v.x;
            if (col + 1 < K) dst[1] = v.y;
            if (col + 2 < K) dst[2] = v.z;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  };
  auto product = [&](const char* buf, int dt, f32x16& dxv) {
    const lds_cp yrow = (lds_cp)(buf + CF::OFF_YH + r * LDY + hh * 16);
#pragma unroll
    for (int v = 0; v < 16; ++v) dxv[v] = 0.f;
    // the dY packs PD k-steps ahead of the products (read just in time every k-step was an LDS round trip + three MFMAs in a row:
    // 75-80 cycles per MFMA, measured); sched_barrier pins the order
    constexpr int PD = KN < 3 ? KN : 3;
    Pack16 yh[PD], yl[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) { yh[i] = lds_pack(yrow + 32 * i); yl[i] = lds_pack(yrow + 32 * LDY + 32 * i); }
#pragma unroll
    for (int ks = 0; ks < KN; ++ks) {
      const Pack16 ch = yh[ks % PD], cl = yl[ks % PD];
      __builtin_amdgcn_sched_barrier(0);
      if (ks + PD < KN) { yh[ks % PD] = lds_pack(yrow + 32 * (ks + PD)); yl[ks % PD] = lds_pack(yrow + 32 * LDY + 32 * (ks + PD)); }
#if !(LBX_ABL & 2)
      // rows = channels, columns = tokens
      dxv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wfl[dt][ks]), __builtin_bit_cast(bf16x8_t, ch), dxv, 0, 0, 0);
      dxv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wfh[dt][ks]), __builtin_bit_cast(bf16x8_t, cl), dxv, 0, 0, 0);
      dxv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wfh[dt][ks]), __builtin_bit_cast(bf16x8_t, ch), dxv, 0, 0, 0);
#else
      dxv[0] += __uint_as_float(ch.w[0] ^ cl.w[1] ^ wfh[dt][ks].w[0] ^ wfl[dt][ks].w[1]);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // LayerNorm: state of the unfinished tile — dA, x-hat and rstd of the lane's own elements, the addend chunks, the first row —
  // finished behind the NEXT tile's first barrier, when every channel tile's partial row sums are in `red` (two barriers per tile)
  f32x16 dxp;
  u32x4_a4 adp[4];
  float xsp[LN ? 16 : 1], rstdp = 0.f;
  int64_t prow0 = -1;
  // (LayerNorm shapes keep the accumulator-layout finish: through the bounce image they measured 1.15x slower — their
  // data-gradient waves are the workgroup's critical path — although the same change took 10-25 % off the shapes without one.)
  // The lane's 8-channel runs [cb, cb + 8), cb = 32 ct + 8 (2 gp + hh), of the addends: the layout the registers have after one
  // v_permlane32_swap per pair; runs past K re-read [K - 4, K) and are dropped or realigned (K % 4 == 2)
  auto load_runs = [&](int64_t row, u32x4_a4 (&o)[4]) {
    const int64_t rr = row < p.M ? row : p.M - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = 32 * dwv + 8 * (2 * (i >> 1) + hh) + 4 * (i & 1);
      const int c0 = col + 4 <= K ? col : K - 4;
      u32x4_a4 v = {0u, 0u, 0u, 0u};
      if (has_acc) v = *reinterpret_cast<const u32x4_a4*>(p.Acc + rr * p.ldacc + c0);
      if (has_acc2) {
        const u32x4_a4 a2 = *reinterpret_cast<const u32x4_a4*>(p.Acc2 + rr * p.ldacc2 + c0);
        v.x = __float_as_uint(__uint_as_float(v.x) + __uint_as_float(a2.x)); v.y = __float_as_uint(__uint_as_float(v.y) + __uint_as_float(a2.y));
        v.z = __float_as_uint(__uint_as_float(v.z) + __uint_as_float(a2.z)); v.w = __float_as_uint(__uint_as_float(v.w) + __uint_as_float(a2.w));
      }
      if (K % 4 == 2 && col == K - 2) { v.x = v.z; v.y = v.w; }
      o[i] = v;
    }
  };
  auto finish_ln = [&](int pb) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < NCT; ++w) {
      const float2 v = *reinterpret_cast<const float2*>(red + ((pb * NCT + w) * 32 + r) * 2);
      s1 += v.x; s2 += v.y;
    }
    s1 *= invK; s2 *= invK;
    float o[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) o[v] = rstdp * (dxp[v] - s1 - xsp[LN ? v : 0] * s2);
    const int64_t row = prow0 + r;
    if (row >= p.M) return;
    float* drow = p.dX + row * p.lddx;
#pragma unroll
    for (int gp2 = 0; gp2 < 2; ++gp2) {
      float c8[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[8 * gp2 + e]), __float_as_uint(o[8 * gp2 + 4 + e]), false, false);
        c8[e] = __uint_as_float(sw[0]);
        c8[4 + e] = __uint_as_float(sw[1]);
      }
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int c0 = 32 * dwv + 8 * (2 * gp2 + hh) + 4 * qq;
        if (c0 >= K || (LBX_ABL & 4)) continue;
        const u32x4_a4 a4 = adp[2 * gp2 + qq];
        float y[4] = {c8[4 * qq] + __uint_as_float(a4.x), c8[4 * qq + 1] + __uint_as_float(a4.y), c8[4 * qq + 2] + __uint_as_float(a4.z),
                      c8[4 * qq + 3] + __uint_as_float(a4.w)};
        if (c0 + 4 <= K) {
          u32x4_a4 u;
          u.x = __float_as_uint(y[0]); u.y = __float_as_uint(y[1]); u.z = __float_as_uint(y[2]); u.w = __float_as_uint(y[3]);
          *reinterpret_cast<u32x4_a4*>(drow + c0) = u;
        } else {
          drow[c0] = y[0];
          if (c0 + 1 < K) drow[c0 + 1] = y[1];
          if (c0 + 2 < K) drow[c0 + 2] = y[2];
        }
      }
    }
  };
  int b = 0;
  LBX_T0
  for (int64_t tile = t0; tile < t1; ++tile, b ^= 1) {
    const char* buf = smem;
    LBX_T(0)
    __syncthreads();
    LBX_T(1)
    if constexpr (LN) {
      if (prow0 >= 0) finish_ln(b ^ 1);   // the previous tile's rows: every channel tile's partial sums are in `red` now
    }
    LBX_T(2)
    // fc2: GELU'(pre-activation) of the wave's own channel tiles from the raw x tile, in the finish's row-wise layout — while the
    // weight-gradient waves convert (in the finish it was 5.8 of the 8.8 k cycles the data-gradient waves spent per tile)
    u32x4_a4 gp[MODE == BX_GELU ? DT : 1][4];
    if constexpr (MODE == BX_GELU) {
      const char* rawx = smem + CF::OFF_RAW + b * CF::RAWB + CF::RAW_X;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int col = 32 * (DT * dwv + dt) + 4 * cch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 pa = *reinterpret_cast<const float4*>(rawx + (8 * k + crow) * CF::RSX + (col < K ? col : 0) * 4);
          gp[dt][k].x = __float_as_uint(gelu_grad_fast(pa.x)); gp[dt][k].y = __float_as_uint(gelu_grad_fast(pa.y));
          gp[dt][k].z = __float_as_uint(gelu_grad_fast(pa.z)); gp[dt][k].w = __float_as_uint(gelu_grad_fast(pa.w));
          // (pinned here: left alone the compiler sinks the arithmetic behind the barrier, to its use)
          asm volatile("" : "+v"(gp[dt][k].x), "+v"(gp[dt][k].y), "+v"(gp[dt][k].z), "+v"(gp[dt][k].w));
        }
      }
    }
    LBX_T(3)
    __syncthreads();
    LBX_T(4)
    const int64_t row0 = tile * 32;
    if constexpr (LN) {
      load_runs(row0 + r, adp);   // consumed one tile later
      product(buf, 0, dxp);
      LBX_T(6)
      prow0 = row0;
      // x-hat of the lane's own (token, channel) elements = hi + lo of the planes; partial row sums of this channel tile
      const lds_cp xrow = (lds_cp)(buf + r * LDX) + (32 * dwv + 4 * hh) * 2;
      rstdp = reinterpret_cast<const float*>(buf + CF::OFF_SM)[r];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const u32x2_t vh = *reinterpret_cast<const LDS_AS u32x2_t*>(xrow + 16 * g4);
        const u32x2_t vl = *reinterpret_cast<const LDS_AS u32x2_t*>(xrow + CF::OFF_XL + 16 * g4);
        xsp[LN ? 4 * g4 : 0] = bf16lo(vh.x) + bf16lo(vl.x); xsp[LN ? 4 * g4 + 1 : 0] = bf16hi(vh.x) + bf16hi(vl.x);
        xsp[LN ? 4 * g4 + 2 : 0] = bf16lo(vh.y) + bf16lo(vl.y); xsp[LN ? 4 * g4 + 3 : 0] = bf16hi(vh.y) + bf16hi(vl.y);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1 += dxp[4 * g4 + e];
          s2 = fmaf(dxp[4 * g4 + e], xsp[LN ? 4 * g4 + e : 0], s2);
        }
      }
      s1 = half_swap_sum(s1);
      s2 = half_swap_sum(s2);
      if (hh == 0) *reinterpret_cast<float2*>(red + ((b * NCT + dwv) * 32 + r) * 2) = make_float2(s1, s2);
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {   // one channel tile at a time: one accumulator and one set of addend chunks alive
        const int ct = DT * dwv + dt;
        f32x16 dxv;
        u32x4_a4 adv[4];
        if constexpr (MODE == BX_GELU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) adv[k] = gp[dt][k];
        } else {
          load_addends(row0, ct, adv);
        }
        product(buf, dt, dxv);
        LBX_T(6)
        float o[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) o[v] = dxv[v];
        finish(o, adv, ct, row0);
        LBX_T(7)
      }
    }
    LBX_T(7)
  }
  LBX_TP("dgrad")
  if constexpr (LN) {
    __syncthreads();   // (the last tile's partial sums)
    if (prow0 >= 0) finish_ln(b ^ 1);
  }
}

template <int K, int N, int MODE, int DT, int TN, int TC>
int lbx_launch(LBXArgs& p, int64_t max_wgs, int* grid_out, hipStream_t st) {
  using CF = LBX<K, N, MODE, DT, TN, TC>;
  auto kern = lnlin3x_bwd_kernel<CF>;
  int per_cu = CF::NWV <= 6 && 2 * CF::SMEM <= 160 * 1024 ? 2 : 1;
  int64_t cap = 256 * (int64_t)per_cu;
  if (cap > max_wgs) cap = max_wgs;
  int64_t grid = p.ntiles < cap ? p.ntiles : cap;
  p.tiles_per_wg = (int)((p.ntiles + grid - 1) / grid);
  grid = (p.ntiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  p.slab_stride = (int64_t)N * (K + 1);
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CF::NT), CF::SMEM, st, p);
  *grid_out = (int)grid;
  return rdst_launch_status("lnlin3x_bwd");
}

// one launch over output features [n0, n0 + NN) of a Linear with N outputs in all (NN = N: the whole layer)
int lbx_dispatch(LBXArgs& p, int K, int NN, int mode, int64_t max_wgs, int* grid_out, hipStream_t st) {
#define RDST_LBX(KK, NV, MM, DT, TN, TC) if (K == KK && NN == NV && mode == MM) return lbx_launch<KK, NV, MM, DT, TN, TC>(p, max_wgs, grid_out, st);
  // norm1 + qkv (C = 120: halves of 180 output features), proj, dense tails, norm2 + fc1, fc2 (GELU on the way in, K = 2 C)
  RDST_LBX(60, 180, BX_LN, 1, 3, 1) RDST_LBX(90, 270, BX_LN, 1, 3, 3) RDST_LBX(120, 180, BX_LN, 1, 3, 2)
  RDST_LBX(60, 60, BX_PLAIN, 1, 1, 2) RDST_LBX(90, 90, BX_PLAIN, 1, 1, 3) RDST_LBX(120, 120, BX_PLAIN, 1, 2, 2)
  RDST_LBX(60, 30, BX_LN, 1, 1, 2) RDST_LBX(90, 30, BX_LN, 1, 1, 3) RDST_LBX(120, 30, BX_LN, 1, 1, 4)
  RDST_LBX(60, 120, BX_LN, 1, 2, 1) RDST_LBX(90, 180, BX_LN, 1, 2, 3) RDST_LBX(120, 240, BX_LN, 1, 2, 4)
  RDST_LBX(120, 60, BX_GELU, 1, 1, 2) RDST_LBX(180, 90, BX_GELU, 2, 1, 6) RDST_LBX(240, 120, BX_GELU, 2, 1, 8)
#undef RDST_LBX
  return RDST_ENOTSUP;
}

}  // namespace

// 0 = not covered; 1 = one launch; 2 = two launches over halves of the output features
int lnlin3x_bwd_kind(int K, int N, bool ln, int in_act) {
  const int mode = ln ? BX_LN : in_act == RDST_ACT_GELU ? BX_GELU : in_act ? -1 : BX_PLAIN;
  if (mode < 0) return 0;
  if (mode == BX_LN && K == 120 && N == 360) return 2;
  if (mode == BX_LN && (K == 60 || K == 90 || K == 120) && (N == 3 * K || N == 30 || N == 2 * K)) return 1;
  if (mode == BX_PLAIN && (K == 60 || K == 90 || K == 120) && N == K) return 1;
  if (mode == BX_GELU && (N == 60 || N == 90 || N == 120) && K == 2 * N) return 1;
  return 0;
}

// The E1 shapes of the one-pass Linear backward on fp32 rows in the split arithmetic; RDST_ENOTSUP for everything else (the caller
// falls back to the three-launch path of linear_mfma.hip).  Queues / runs the slab sums and the LayerNorm finish itself.
int lnlin3x_bwd_f32(const float* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats, int in_act, const float* Wt,
                    const float* dY, int64_t lddy, float* dX, int64_t lddx, const float* acc, int64_t ldacc, const float* acc2,
                    int64_t ldacc2, float* dW, float* dbias, float* dln_w, float* dln_b, float* slab, float* G, int64_t M, int K, int N,
                    float s, hipStream_t st) {
  const bool ln = ln_w != nullptr;
  const int kind = lnlin3x_bwd_kind(K, N, ln, in_act);
  if (!kind || s != 1.0f || M <= 0 || !dX || !dW || !dbias || (ln && (!ln_b || !dln_w || !dln_b || !stats || !G))) return RDST_ENOTSUP;
  if (((uintptr_t)X & 3) || ((uintptr_t)dY & 3) || ((uintptr_t)dX & 3) || ((uintptr_t)acc & 3) || ((uintptr_t)acc2 & 3)) return RDST_ENOTSUP;
  const int mode = ln ? BX_LN : in_act == RDST_ACT_GELU ? BX_GELU : BX_PLAIN;
  if (mode == BX_GELU && (acc || acc2)) return RDST_ENOTSUP;
  if (((M - 1) * ldx + K) * 4 >= (1ll << 31) || ((M - 1) * lddy + N) * 4 >= (1ll << 31) || M * 8 >= (1ll << 31)) return RDST_ENOTSUP;   // 32-bit DMA offsets
  const int max_wgs = linear_wgrad_max_wgs(N);
  const int NN = kind == 2 ? N / 2 : N;
  for (int half = 0; half < kind; ++half) {
    const int n0 = half * NN;
    LBXArgs p{};
    p.X = X; p.ldx = ldx; p.stats = stats; p.lnw = ln_w; p.W = Wt + (int64_t)n0 * K; p.dY = dY + n0; p.lddy = lddy; p.dX = dX; p.lddx = lddx;
    p.Acc = half == 0 ? acc : dX; p.ldacc = half == 0 ? ldacc : lddx; p.Acc2 = half == 0 ? acc2 : nullptr; p.ldacc2 = ldacc2;
    float* sl = slab + (int64_t)half * max_wgs * NN * (K + 1);   // (the region holds max_wgs slabs of N x (K + 1): half the rows each)
    p.slab = sl; p.M = M; p.ntiles = (M + 31) / 32;
    p.x_bytes = (int)(((M - 1) * ldx + K) * 4); p.y_bytes = (int)(((M - 1) * lddy + NN) * 4);
    int grid = 0;
    if (int rc = lbx_dispatch(p, K, NN, mode, max_wgs, &grid, st)) return rc;
    rbatch::SumJob sj{};
    sj.slab = sl; sj.nwg = grid; sj.stride = (int64_t)NN * (K + 1); sj.g4 = 0; sj.tot = NN * (K + 1);
    if (!ln) {
      sj.map = rbatch::MAP_LINEAR; sj.out = dW; sj.out2 = dbias; sj.a = K; sj.b = K + 1; sj.s = 1.0f;
    } else {
      sj.map = rbatch::MAP_COPY; sj.out = G + (int64_t)n0 * (K + 1);
    }
    if (int rc = rbatch::sum(sj, st)) return rc;
  }
  if (ln) return wgrad_ln_finish_launch(G, Wt, ln_w, ln_b, N, K, 1.0f, dW, dbias, dln_w, dln_b, st);
  return 0;
}
