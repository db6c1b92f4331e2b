// Backward of a (LayerNorm-fused) Linear in ONE pass over (x, dY), re-cut for the E1 shapes (K in {60, 90, 120};
// N = 3K behind norm1, N = K for proj, N = 30 for the dense tails): the round-1 kernel (mlp_mfma.hip:lnlin_bwd_kernel)
// ran the weight-gradient product on all waves, THEN the data-gradient product on K/32 of them, with the W.gamma image
// (up to 110 KB) in LDS, single-buffered tiles and three or four barriers per 32-token tile.  Here
//   * waves are SPECIALISED: weight-gradient waves own a block of TN x TC accumulator tiles of G = dY^T.x-hat for the
//     whole kernel; data-gradient wave d owns channel tile d of dX-hat^T = (W gamma)^T.dY^T and keeps its (W gamma)^T
//     fragments (the MFMA A operand, 4 registers per k-step) in REGISTERS for the whole kernel — the W image is gone
//     from LDS and from the prologue (every wave gathers its own fragments from the fp32 parameter, once);
//   * both kinds work on the same staged tile at the same time; the tiles are double-buffered, so there is ONE barrier
//     per tile; the LayerNorm backward needs the two row sums over all channel tiles: the partial sums of tile i are
//     exchanged through LDS and the rows are finished after the NEXT barrier (dX-hat, x-hat and dX_add of the wave's own
//     channel tile wait in registers);
//   * loads stay register-staged (issued one tile ahead, normalised with the forward's statistics while they are written
//     to LDS), every thread owning up to four 16-byte chunks of [dY | x | dX_add].
// Same slab format (bf16, four rows per 8-byte store) and the same fixed-order reductions as the round-1 kernel.
#include "linear.h"
#include "mfma.h"
#include "wattn_hd.h"
#include <stdlib.h>

#ifndef LB3_LOADW_SHAPE
#define LB3_LOADW_SHAPE(K, N) (((K) == 60 && (N) == 180) || ((K) == 90 && (N) == 270) || ((K) == 120 && (N) == 120))   // which shapes let the
// weight-gradient waves do all the staging (see LB3::LOADW).  A/B on one box, cold, us per call: norm1 + qkv 40.7 -> 37.1 (C = 60), 53.5 -> 51.8 (90),
// proj 38.0 -> 33.5 (C = 120); the other shapes are equal or slower (the dense tails' single weight-gradient wave spills: 41 -> 154 us)
#endif
#ifndef LB3_ABL
#define LB3_ABL 0   // compile-time ablations (tools/abl_build.sh): 1 no weight-gradient MFMAs, 2 no data-gradient MFMAs (the wide chain), 4 no W gather, 8 no slab dump, 16 no LDS zeroing.
                    // Round 4: with BOTH removed the K = 120, N = 360 launch goes from 53.7 to 50.9 us: the kernel is bound by its
                    // load -> stash -> barrier skeleton (one tile of loads in flight per workgroup), not by its arithmetic
#endif

// -DLB3_STAMPS: workgroup 0 prints, per wave, the clock64() ticks it spent in each phase of the tile loop (tools/abl_build.sh)
#ifdef LB3_STAMPS
#define LB3_T0 long long tk_[5] = {0, 0, 0, 0, 0}, tl_ = clock64();
#define LB3_T(i) { const long long n_ = clock64(); tk_[i] += n_ - tl_; tl_ = n_; }
#define LB3_TP(role) if (blockIdx.x == 0 && lane == 0) printf("wave %d %s: stash %lld  fetch %lld  barrier %lld  compute %lld  finish %lld\n", wave, role, tk_[0], tk_[1], tk_[2], tk_[3], tk_[4]);
#else
#define LB3_T0
#define LB3_T(i)
#define LB3_TP(role)
#endif

namespace {
using namespace wahd;
using MM = Mma<bf16>;

struct LB3Args {
  const bf16* X; int64_t ldx; const float* stats; const float* lnw; const float* W;
  const bf16* dY; int64_t lddy; bf16* dX; int64_t lddx; const bf16* Acc; int64_t ldacc;
  const bf16* Acc2; int64_t ldacc2;   // second addend of dX (same chunks as Acc, summed with it on the way into LDS)
  float* slab; int64_t slab_stride;
  int64_t M; int64_t ntiles; int tiles_per_wg;
};

__host__ __device__ constexpr int lb3_ldy(int NP) {   // dY tile row stride: odd 16-B slot count, not 16..47 (mod 256)
  const int b = NP * 2 + 16;
  return (b & 255) < 48 ? b + 64 : b;
}

template <int K_, int N_>
struct LB3 {
  static constexpr int K = K_, N = N_;
  static constexpr int NCT = (K + 1 + 31) / 32, NW = (N + 31) / 32, KN = (N + 15) / 16;
  static constexpr bool WIDE = N >= 180;
  static constexpr int TN = WIDE ? 3 : 1;
  static constexpr int TC = (K == 120 && N == 360) ? 2 : WIDE ? 1 : NCT;
  static_assert(NW % TN == 0 && NCT % TC == 0, "blocking");
  static constexpr int NGN = NW / TN, NGC = NCT / TC, NWG = NGN * NGC;   // weight-gradient waves
  static constexpr int NDG = NCT;                                          // data-gradient waves
  static constexpr int NWV = NWG + NDG, NT = 64 * NWV;
  static constexpr int PKX = (K * 2 + 15) / 16, PKY = (N * 2 + 15) / 16;
  static constexpr int NY = 32 * PKY, NX = 32 * PKX;
  // loader: WIDE shapes split the kinds between the roles (weight-gradient waves stage dY and x, data-gradient waves
  // dX_add: the registers of either role are full); the narrow shapes let every thread take its share of every kind
  // LOADW (round 6, after the per-wave phase stamps of this kernel and of lnlin3x_mfma.hip: the data-gradient waves are the workgroup's
  // critical path — staging, products, LayerNorm finish — while the weight-gradient waves wait at the barrier half of the time): the
  // weight-gradient waves stage EVERYTHING, the data-gradient waves nothing
  static constexpr bool LOADW = LB3_LOADW_SHAPE(K_, N_);
  static constexpr bool SPLITR = WIDE || LOADW;   // the roles load with separate thread sets
  static constexpr int NTW = SPLITR ? 64 * NWG : NT, NTD = SPLITR ? 64 * NDG : NT;
  static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
  static constexpr bool ACC_BY_W = K == 120 && N == 360;   // 92 registers of W fragments: these data-gradient waves stage nothing
  static constexpr int W_SY = cdiv(NY, NTW), W_SX = cdiv(NX, NTW), W_SA = (WIDE && !ACC_BY_W && !LOADW) ? 0 : cdiv(NX, NTW);
  static constexpr int D_SY = SPLITR ? 0 : W_SY, D_SX = SPLITR ? 0 : W_SX, D_SA = (ACC_BY_W || LOADW) ? 0 : cdiv(NX, NTD);
  static constexpr int CP = 32 * NCT;
  // (strides of 64 / 192 (mod 256) bytes — the four rows of a transposed read on disjoint bank quarters — were measured:
  // qkv / proj unchanged within noise, the tails 1-3 us slower (the 8-byte slice reads of the epilogue collide): not the limit)
  static constexpr int LDX = NCT == 4 ? 336 : CP * 2 + 16;
  static constexpr int LDY = lb3_ldy(32 * NW);
  static constexpr int OFF_AC = 32 * LDX, OFF_DY = 64 * LDX, OFF_SM = OFF_DY + 32 * LDY, BUF = OFF_SM + 128;
  static constexpr int OFF_RED = 2 * BUF;
  static constexpr int SMEM = OFF_RED + 2 * NCT * 256;
  // workgroups per CU: at most 2 (measured: up to 4 for the 3-5-wave shapes changes the tails by -1.0 / +2.3 / 0 us and
  // proj 60 by +1.2 us — more slabs and prologues, no more bandwidth)
  static constexpr int PERCU0 = 12 / NWV < 1 ? 1 : 12 / NWV;
  // round 4: ONE workgroup per CU for the wide shapes: K = 60, N = 180 with two (6 waves each) 38.6 us, with one 33.1 us — the
  // steady-state tile loop already streams at 4-5 TB/s (1.5-2.4 us per 32-token tile, measured by the slope over M), what a
  // launch loses is its fixed part (W fragment gather, slab dump, first tile), and that is paid per workgroup
  static constexpr int PERCU1 = WIDE ? 1 : PERCU0 > 2 ? 2 : PERCU0;
  static constexpr int PERCU = PERCU1 * SMEM > 160 * 1024 ? 1 : PERCU1;
  static constexpr int WPS = (NWV * PERCU + 3) / 4;                          // waves per SIMD (the launch bound)
};

// Register-staged loader of one role: SY slots of dY chunks, SX of x, SA of dX_add per thread (fixed kinds: base pointers
// and row strides are wave-uniform); a thread's chunk of a slot is described by two registers.
template <class CF, bool LN, int SY, int SX, int SA>
struct LB3Loader {
  static constexpr int NS = SY + SX + SA;
  int meta[NS > 0 ? NS : 1];   // byte offset in the row | tile row << 16 | on << 24 | aligned << 28 (the LDS position is re-derived
                               // from it at every stash: one more loop-invariant register per slot was what spilled at K = 120, N = 360)
  u32x4_a4 rd[NS > 0 ? NS : 1];
  u32x4_a4 rd2[SA > 0 ? SA : 1];   // the second addend's chunks (dX_add2), summed with dX_add at stash time
  float2 rst[SX > 0 ? SX : 1];
  static __device__ __forceinline__ int kind(int u) { return u < SY ? 0 : u < SY + SX ? 1 : 2; }
  __device__ __forceinline__ void setup(int tid_r, int ntr, bool has_acc) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int kd = kind(u);
      const int idx = tid_r + ntr * (kd == 0 ? u : kd == 1 ? u - SY : u - SY - SX);
      const int per = kd == 0 ? CF::PKY : CF::PKX, rowbytes = kd == 0 ? CF::N * 2 : CF::K * 2;
      const bool on = idx < 32 * per && (kd != 2 || has_acc);
      const int row = on ? idx / per : 0, chk = on ? idx - row * per : 0;
      int o = chk * 16;
      if (o + 16 > rowbytes) o = rowbytes - 16;
      meta[u] = o | (row << 16) | ((on ? 1 : 0) << 24) | (((o & 15) == 0 ? 1 : 0) << 28);
    }
  }
  __device__ __forceinline__ void fetch(const LB3Args& p, int64_t tile) {   // unconditional loads from clamped rows
    const uint32_t Mu = (uint32_t)p.M;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int kd = kind(u);
      uint32_t row = (uint32_t)(tile * 32) + (uint32_t)((meta[u] >> 16) & 31);
      row = row < Mu ? row : Mu - 1;
      const char* base = reinterpret_cast<const char*>(kd == 0 ? p.dY : kd == 1 ? p.X : (p.Acc ? p.Acc : p.X));
      const uint32_t ldb = (uint32_t)((kd == 0 ? p.lddy : kd == 1 ? p.ldx : p.ldacc) * 2);
      rd[u] = *reinterpret_cast<const u32x4_a4*>(base + (size_t)(row * ldb + (uint32_t)(meta[u] & 0xffff)));
      if (kd == 2 && p.Acc2)
        rd2[u - SY - SX] = *reinterpret_cast<const u32x4_a4*>(reinterpret_cast<const char*>(p.Acc2) +
                                                              (size_t)(row * (uint32_t)(p.ldacc2 * 2) + (uint32_t)(meta[u] & 0xffff)));
      if (LN && kd == 1) rst[u - SY] = *reinterpret_cast<const float2*>(p.stats + 2 * (size_t)row);
    }
  }
  __device__ __forceinline__ void stash(const LB3Args& p, int64_t tile, char* buf) {
    const uint32_t Mu = (uint32_t)p.M;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int kd = kind(u);
      if (!((meta[u] >> 24) & 1)) continue;
      const int mrow = (meta[u] >> 16) & 31;
      const bool valid = (uint32_t)(tile * 32) + (uint32_t)mrow < Mu;
      Pack16 v;
      v.w[0] = valid ? rd[u].x : 0u; v.w[1] = valid ? rd[u].y : 0u; v.w[2] = valid ? rd[u].z : 0u; v.w[3] = valid ? rd[u].w : 0u;
      if (kd == 2 && p.Acc2) {   // dX_add + dX_add2 (fp32 sum, one bf16 rounding); dX_add may be absent
        const u32x4_a4 a = rd[u], b2 = rd2[u - SY - SX];
        const bool h1 = p.Acc != nullptr;
        float f[8];
        f[0] = bf16lo(b2.x) + (h1 ? bf16lo(a.x) : 0.f); f[1] = bf16hi(b2.x) + (h1 ? bf16hi(a.x) : 0.f);
        f[2] = bf16lo(b2.y) + (h1 ? bf16lo(a.y) : 0.f); f[3] = bf16hi(b2.y) + (h1 ? bf16hi(a.y) : 0.f);
        f[4] = bf16lo(b2.z) + (h1 ? bf16lo(a.z) : 0.f); f[5] = bf16hi(b2.z) + (h1 ? bf16hi(a.z) : 0.f);
        f[6] = bf16lo(b2.w) + (h1 ? bf16lo(a.w) : 0.f); f[7] = bf16hi(b2.w) + (h1 ? bf16hi(a.w) : 0.f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = valid ? f[e] : 0.f;
        v = MM::pack(f);
      }
      if (LN && kd == 1) {
        const float2 st2 = rst[u - SY];
        float f[8];
        f[0] = bf16lo(rd[u].x); f[1] = bf16hi(rd[u].x); f[2] = bf16lo(rd[u].y); f[3] = bf16hi(rd[u].y);
        f[4] = bf16lo(rd[u].z); f[5] = bf16hi(rd[u].z); f[6] = bf16lo(rd[u].w); f[7] = bf16hi(rd[u].w);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = valid ? (f[e] - st2.x) * st2.y : 0.f;
        v = MM::pack(f);
        if ((meta[u] & 0xffff) == 0) reinterpret_cast<float*>(buf + CF::OFF_SM)[mrow] = st2.y;
      }
      int mt = meta[u];
      asm volatile("" : "+v"(mt));   // (not hoisted out of the tile loop)
      const int lrow = (mt >> 16) & 31, lo = mt & 0xffff;
      char* dst = buf + (kd == 0 ? CF::OFF_DY + lrow * CF::LDY : (kd == 2 ? CF::OFF_AC : 0) + lrow * CF::LDX) + lo;
      if ((meta[u] >> 28) & 1) *reinterpret_cast<Pack16*>(dst) = v;
      else {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        d[0] = v.w[0]; d[1] = v.w[1]; d[2] = v.w[2]; d[3] = v.w[3];
      }
    }
  }
};

template <int K, int N, bool LN>
__global__ void __launch_bounds__((LB3<K, N>::NT), (LB3<K, N>::WPS)) lnlin3_bwd_kernel(const LB3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = LB3<K, N>;
  constexpr int NCT = CF::NCT, NW = CF::NW, KN = CF::KN, TN = CF::TN, TC = CF::TC, NGC = CF::NGC, NWG = CF::NWG, NT = CF::NT;
  constexpr int LDX = CF::LDX, LDY = CF::LDY;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
  const bool has_acc = p.Acc != nullptr || p.Acc2 != nullptr;

  // ---- prologue: zero both tile buffers (pads, absent dX_add), ones column of x-hat ----
  if (!(LB3_ABL & 16)) lds_zero16(smem, CF::SMEM, tid, NT);
  __syncthreads();
  if (tid < 64) *reinterpret_cast<uint16_t*>(smem + (tid >> 5) * CF::BUF + (tid & 31) * LDX + K * 2) = 0x3f80;

  const int q = (lane & 15) >> 2, pp = lane & 3, gq1 = (lane >> 4) & 1;
  const int trc = (16 * gq1 + 4 * pp) * 2;
  const int tro_y = CF::OFF_DY + (8 * hh + q) * LDY + trc;   // transposed-read lane offsets inside a buffer
  const int tro_x = (8 * hh + q) * LDX + trc;

  const bool is_wg = wave < NWG;
  const int gn = wave / NGC, gc = wave - gn * NGC;   // weight-gradient block of this wave
  const int dct = wave - NWG;                         // data-gradient channel tile of this wave

  const int64_t t0 = (int64_t)blockIdx.x * p.tiles_per_wg;
  const int64_t t1 = t0 + p.tiles_per_wg < p.ntiles ? t0 + p.tiles_per_wg : p.ntiles;
  __syncthreads();   // ones columns

  // The two roles run SEPARATE copies of the tile loop (same barrier count): in one loop body the register allocator
  // would have to keep the G tiles and the W fragments alive side by side in every wave.
  if (is_wg) {
    LB3Loader<CF, LN, CF::W_SY, CF::W_SX, CF::W_SA> ld;
    ld.setup(tid, CF::NTW, has_acc);
    if (t0 < t1) ld.fetch(p, t0);
    f32x16 G[TN][TC];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) G[i][j][v] = 0.f;
    int b = 0;
    LB3_T0
    for (int64_t tile = t0; tile < t1; ++tile, b ^= 1) {
      char* buf = smem + b * CF::BUF;
      ld.stash(p, tile, buf);
      LB3_T(0)
      ld.fetch(p, tile + 1 < t1 ? tile + 1 : tile);   // every iteration defines the whole prefetch set
      LB3_T(1)
      __syncthreads();   // the one barrier of the tile: buffer b staged; everybody is done with buffer b^1 and red[b]
      LB3_T(2)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Pack16 ya[TN], xb[TC];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const lds_cp yp = (lds_cp)(buf + tro_y + 16 * s * LDY) + (gn * TN + i) * 64;
          ya[i] = lds_tr_pack(yp, yp + 4 * LDY);
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {
          const lds_cp xp = (lds_cp)(buf + tro_x + 16 * s * LDX) + (gc * TC + j) * 64;
          xb[j] = lds_tr_pack(xp, xp + 4 * LDX);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) {
#if !(LB3_ABL & 1)
            MM::mma(G[i][j], ya[i], xb[j]);   // rows = output features n, columns = channels (column K = d(bias))
#else
            G[i][j][0] += __uint_as_float(ya[i].w[0] ^ xb[j].w[0]);
#endif
          }
      }
      LB3_T(3)
    }
    LB3_TP("wgrad")
    if (LN) __syncthreads();
    // bf16 slab in groups of 4 rows (reduce_batch.h, "G4"): G [N][K+1]; p.slab_stride counts 8-byte groups
    uint2* my = reinterpret_cast<uint2*>(p.slab) + (int64_t)blockIdx.x * p.slab_stride;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const int c = 32 * (gc * TC + j) + r;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = 32 * (gn * TN + i) + 8 * g4 + 4 * hh;
          if (n0 < N && c <= K && !(LB3_ABL & 8))
            my[(int64_t)(n0 >> 2) * (K + 1) + c] =
                make_uint2(pack_bf16x2(G[i][j][4 * g4], G[i][j][4 * g4 + 1]), pack_bf16x2(G[i][j][4 * g4 + 2], G[i][j][4 * g4 + 3]));
        }
      }
    return;
  }

  // ---- data-gradient wave: (W gamma)^T fragments, lane (r, hh) of k-step ks holds W[16 ks + 8 hh + e][32 dct + r] gamma
  LB3Loader<CF, LN, CF::D_SY, CF::D_SX, CF::D_SA> ld;
  ld.setup(CF::SPLITR ? tid - 64 * NWG : tid, CF::NTD, has_acc);
  if (t0 < t1) ld.fetch(p, t0);
  Pack16 wf[KN];
  {
    const int c = 32 * dct + r;
    const int cc = c < K ? c : K - 1;
    const float gm = (c < K) ? (LN ? p.lnw[cc] : 1.0f) : 0.f;
#pragma unroll
    for (int ks = 0; ks < KN; ++ks) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = 16 * ks + 8 * hh + e;
#if LB3_ABL & 4
        const float w = 0.01f * (float)(n + cc);
#else
        const float w = p.W[(size_t)(n < N ? n : N - 1) * K + cc];
#endif
        f[e] = n < N ? w * gm : 0.f;
      }
      wf[ks] = MM::pack(f);
    }
  }
  constexpr float invK = 1.0f / (float)K;
  float* red = reinterpret_cast<float*>(smem + CF::OFF_RED);
  // state of the unfinished tile (LN: finished after the next barrier)
  f32x16 dx;
  u32x2_t xs[4], as[4];
  float rstd = 0.f;
  int64_t prow = -1;
  auto finish = [&](int pb) {
    float s1 = 0.f, s2 = 0.f;
    if (LN) {
#pragma unroll
      for (int w = 0; w < NCT; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(red + ((pb * NCT + w) * 32 + r) * 2);
        s1 += v.x; s2 += v.y;
      }
      s1 *= invK; s2 *= invK;
    }
    float o[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float xh[4] = {bf16lo(xs[g4].x), bf16hi(xs[g4].x), bf16lo(xs[g4].y), bf16hi(xs[g4].y)};
      const float ac[4] = {bf16lo(as[g4].x), bf16hi(as[g4].x), bf16lo(as[g4].y), bf16hi(as[g4].y)};
#pragma unroll
      for (int e = 0; e < 4; ++e) o[4 * g4 + e] = LN ? fmaf(rstd, dx[4 * g4 + e] - s1 - xh[e] * s2, ac[e]) : dx[4 * g4 + e] + ac[e];
    }
    if (prow < p.M) {
      bf16* drow = p.dX + prow * p.lddx;
#pragma unroll
      for (int gp2 = 0; gp2 < 2; ++gp2) {
        float c8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[8 * gp2 + e]), __float_as_uint(o[8 * gp2 + 4 + e]), false, false);
          c8[e] = __uint_as_float(sw[0]);
          c8[4 + e] = __uint_as_float(sw[1]);
        }
        const int cb = 32 * dct + 8 * (2 * gp2 + hh);
        if (cb + 8 <= K) {
          u32x4_a4 u;
          u.x = pack_bf16x2(c8[0], c8[1]); u.y = pack_bf16x2(c8[2], c8[3]);
          u.z = pack_bf16x2(c8[4], c8[5]); u.w = pack_bf16x2(c8[6], c8[7]);
          *reinterpret_cast<u32x4_a4*>(drow + cb) = u;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (cb + e < K) drow[cb + e] = __float2bfloat16(c8[e]);
        }
      }
    }
  };
  int b = 0;
  LB3_T0
  for (int64_t tile = t0; tile < t1; ++tile, b ^= 1) {
    char* buf = smem + b * CF::BUF;
    ld.stash(p, tile, buf);
    LB3_T(0)
    ld.fetch(p, tile + 1 < t1 ? tile + 1 : tile);
    LB3_T(1)
    __syncthreads();
    LB3_T(2)
    if (LN && prow >= 0) finish(b ^ 1);
    LB3_T(4)
    const lds_cp yrow = (lds_cp)(buf + CF::OFF_DY + r * LDY + hh * 16);
#pragma unroll
    for (int v = 0; v < 16; ++v) dx[v] = 0.f;
    if constexpr (KN > 12) {   // one accumulation chain: the W fragments leave no room for a second accumulator
#pragma unroll
      for (int ks = 0; ks < KN; ++ks) {
#if !(LB3_ABL & 2)
        MM::mma(dx, wf[ks], lds_pack(yrow + 32 * ks));   // rows = channels, columns = tokens
#else
        dx[0] += __uint_as_float(lds_pack(yrow + 32 * ks).w[0]);
#endif
      }
    } else {
      f32x16 d2;
#pragma unroll
      for (int v = 0; v < 16; ++v) d2[v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KN; ++ks) {
        const Pack16 yb = lds_pack(yrow + 32 * ks);
        if (ks & 1) MM::mma(d2, wf[ks], yb);
        else MM::mma(dx, wf[ks], yb);
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) dx[v] += d2[v];
    }
    const lds_cp xrow = (lds_cp)(buf + r * LDX) + (32 * dct + 4 * hh) * 2;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      xs[g4] = *reinterpret_cast<const LDS_AS u32x2_t*>(xrow + 16 * g4);
      as[g4] = *reinterpret_cast<const LDS_AS u32x2_t*>(xrow + CF::OFF_AC + 16 * g4);
    }
    prow = tile * 32 + r;
    if (LN) {
      rstd = reinterpret_cast<const float*>(buf + CF::OFF_SM)[r];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float xh[4] = {bf16lo(xs[g4].x), bf16hi(xs[g4].x), bf16lo(xs[g4].y), bf16hi(xs[g4].y)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1 += dx[4 * g4 + e];
          s2 = fmaf(dx[4 * g4 + e], xh[e], s2);
        }
      }
      s1 = half_swap_sum(s1);
      s2 = half_swap_sum(s2);
      if (hh == 0) *reinterpret_cast<float2*>(red + ((b * NCT + dct) * 32 + r) * 2) = make_float2(s1, s2);
    } else {
      LB3_T(3)
      finish(b);
      LB3_T(4)
    }
    LB3_T(3)
  }
  LB3_TP("dgrad")
  if (LN) {
    __syncthreads();
    if (prow >= 0) finish(b ^ 1);
  }
}

template <int K, int N, bool LN>
int lb3_launch(LB3Args& p, int64_t max_wgs, int* grid_out, hipStream_t st) {
  using CF = LB3<K, N>;
  auto kern = lnlin3_bwd_kernel<K, N, LN>;
  static int percu_env = -1;
  if (percu_env < 0) { const char* e = rdst_dbg_getenv("RDST_LB3_PERCU"); percu_env = e ? atoi(e) : 0; }
  int per_cu = CF::PERCU;
  if (percu_env > 0 && percu_env < per_cu) per_cu = percu_env;
  int64_t cap = 256 * (int64_t)per_cu;
  if (cap > max_wgs) cap = max_wgs;
  int64_t grid = p.ntiles < cap ? p.ntiles : cap;
  p.tiles_per_wg = (int)((p.ntiles + grid - 1) / grid);
  grid = (p.ntiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (CF::SMEM > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CF::NT), CF::SMEM, st, p);
  *grid_out = (int)grid;
  return rdst_launch_status("lnlin3_bwd");
}

}  // namespace

// The E1 shapes of the one-pass Linear backward; RDST_ENOTSUP for everything else (the caller falls back to
// lnlin_bwd_kernel).  Leaves one slab row per workgroup (*grid_out of them); the caller queues the reductions.
int lnlin3_bwd_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* stats, const float* Wt, const bf16* dY,
                    int64_t lddy, bf16* dX, int64_t lddx, const bf16* acc, int64_t ldacc, const bf16* acc2, int64_t ldacc2,
                    float* slab, int64_t slab_stride, int64_t M, int K, int N, int64_t max_wgs, int* grid_out, hipStream_t st) {
  static int off = -1;
  if (off < 0) { const char* e = rdst_dbg_getenv("RDST_LB3_OFF"); off = e ? atoi(e) : 0; }
  if (off) return RDST_ENOTSUP;
  const bool ln = ln_w != nullptr;
  LB3Args p{};
  p.X = X; p.ldx = ldx; p.stats = stats; p.lnw = ln_w; p.W = Wt; p.dY = dY; p.lddy = lddy; p.dX = dX; p.lddx = lddx;
  p.Acc = acc; p.ldacc = ldacc; p.Acc2 = acc2; p.ldacc2 = ldacc2; p.slab = slab; p.slab_stride = slab_stride; p.M = M;
  p.ntiles = (M + 31) / 32;
#define RDST_LB3(KK, NN, LL) if (K == KK && N == NN && ln == LL) return lb3_launch<KK, NN, LL>(p, max_wgs, grid_out, st);
  RDST_LB3(60, 180, true) RDST_LB3(90, 270, true) RDST_LB3(120, 360, true)
  RDST_LB3(60, 30, true) RDST_LB3(90, 30, true) RDST_LB3(120, 30, true)
  RDST_LB3(60, 60, false) RDST_LB3(90, 90, false) RDST_LB3(120, 120, false)
#undef RDST_LB3
  return RDST_ENOTSUP;
}
