// K8: the attention half of a Swin block in ONE launch (bf16, 8x8 windows, 6 heads of dim D = 10 / 15 / 20):
//     x1 = x + proj(WindowAttention(qkv(LayerNorm(x))))          networks/swin_transformer_sr.py:240-271, :110-141 inside
// replacing lin3 (LayerNorm + qkv) -> K1 (window attention) -> lin3 (proj + shortcut): those three launches move 11 C bytes
// per token (qkv and the attention output are written and read back); a window's 64 tokens never meet another window before
// the Mlp, so one workgroup can take a window from x to x1: 6 C bytes per token (x in; qkv, the attention output a and x1
// out, the three tensors the backward reads).  The arithmetic is the unfused kernels' arithmetic, operation for operation
// (same bf16 fragments, same k order, same epilogue formulas, same softmax): the outputs are bit-identical to theirs.
//
// One 12-wave workgroup per CU walks windows g, g + G, ...; per window, three workgroup barriers:
//   top   x(n) and its LayerNorm statistics are in LDS (put there during window n - 1, see below)
//   A     qkv = LN(x) Wqkv^T: 6 / 9 / 12 output tiles (q | k | v, each section padded to whole 32-channel tiles of its own:
//         pack.h lin3sec_pack_block) x 2 token halves.  A wave keeps the fragments of ONE tile in registers for the whole
//         kernel (16 / 24 / 32 registers), B operands are ds_read_b128 of the raw token rows, LayerNorm enters in the epilogue
//         (lin3_mfma.hip); the tile goes to the Q / K / V section in LDS (16-byte stores of 8 channels of a token) and, from
//         the same registers, to the qkv rows in HBM (SA_QKV_DIRECT).  Beside it: the PREVIOUS window's x1 and a rows leave
//         for HBM as whole-row copies (see C)
//   B1    waves 0-3 put the next window's rows in flight by LDS-DMA (each its own 16 rows); attention: wave = (head, query
//         half), one of the 12.  K1's data flow (wattn_mfma_hd.hip: S^T = K Q^T with the bias / scale as initial accumulator,
//         the shift mask as one more k-step, in-register softmax, O^T = V^T P^T) for ONE head per wave; O goes to its own tile
//   B2    waves 4-11: proj + bias + x, with the residual read from the x tile in LDS and x1 written back IN PLACE (its row
//         copy to HBM rides on the next window's phase A, like a's out of the O tile); the proj fragments (8 / 18 / 32 KB)
//         live in LDS.  waves 0-3: wait for the next window's rows and compute their LayerNorm statistics (4 lanes per
//         token, two passes: lin3's): neither the row fetch nor the statistics are ever on the critical path.
// No vector register ever holds an input row; the only global loads of the steady state are the LDS-DMA pieces.
// What bounds it: the NUMBER of vector instructions (480 per wave and window on the mean, 4.4
// SIMD cycles each) issued by three waves per SIMD that reach the same dependency at the same time — not bytes (0.33-0.41
// of 8 TB/s), not the matrix pipe (16 % busy).  Seven variants of the store / copy / prefetch structure were measured on one
// box each (tools/sa_ab.sh); the switches below keep the ones that are still a compile-time choice.
#include "wattn_hd.h"
#include "pack.h"
#include "linear.h"
#include <stdlib.h>

namespace {
using namespace wahd;

#ifndef SA_ABL
#define SA_ABL 0   // compile-time ablation switches (tools/abl_build.sh): 1 no qkv stores, 2 no a stores, 4 no x1 stores, 8 no attention,
#endif             // 16 no G1 arithmetic, 32 no proj arithmetic, 64 no row fetch after the first window
#ifndef SA_ACOPY_C
#define SA_ACOPY_C 0      // 1: the a rows leave in phase C on the loader waves (which only have the statistics to do there); 0: in the
#endif                    // next window's phase A on all waves, beside the x1 rows.  Measured equal (+-0.7 us); s_setprio by dispatch
                          // age (the youngest wave of a SIMD is the slowest in every phase) was measured too: +2 .. 3 us
#ifndef SA_QKV_DIRECT
#define SA_QKV_DIRECT 1   // 1: the qkv tiles go to HBM straight from the accumulators (16 bytes of a token per lane); 0: as whole-row
#endif                    // copies out of the sections during the attention phase.  Measured equal within 2 us at every width
constexpr int SA_NLW = 4;   // loader waves (0 .. 3)
#ifndef SA_NW10
#define SA_NW10 12  // waves per workgroup at C = 60: 12 = one workgroup per CU as at C = 90 / 120; 6 = TWO workgroups per CU (79 KB of LDS each),
#endif              // each wave two GEMM items and both query halves of its head — measured: 52 us against 40 (two half-size workgroups
                    // double every wave's dependent chain and pay two prologues; their phases do not interleave usefully)
constexpr int sa_nw(int D) { return D == 10 ? SA_NW10 : 12; }

struct SAArgs {
  const bf16* X; int64_t ldx; uint32_t x_bytes;
  const bf16* Wq; const float* sbq;      // qkv image: fragments [NTQ][KS][64][8], S | b' [2][NTQ * 32]
  const bf16* Wp; const float* sbp;      // proj image: fragments [NTS][KS][64][8], S | b' [2][NTS * 32]
  const float* table;
  bf16* qkv; int64_t ldq;
  bf16* a; int64_t lda;
  bf16* x1; int64_t ld1;
  float* stats;
  WinGeom g;
  float scale;
  int G;
#ifdef SA_STAMPS
  unsigned long long* stamps;   // [grid][12 waves][8] accumulated cycles per phase (diagnostic build only: tools/abl_build.sh -DSA_STAMPS)
#endif
};

template <int D>
struct SA {
  static constexpr int HEADS = 6, C = HEADS * D;
  static constexpr int KS = (C + 15) / 16, NTS = (C + 31) / 32, NTQ = 3 * NTS;   // 4/6/8 k-steps, 2/3/4 tiles per section
  // x tile: [64 tokens][XS bytes], odd number of 16-byte slots (b128 row reads), XD slots carry data
  static constexpr int XS0 = KS * 32, XS = ((XS0 / 16) & 1) ? XS0 : XS0 + 16, XSLOTS = XS / 16, XD = (2 * C + 15) / 16;
  static constexpr int XBUFB = 64 * XS;
  static constexpr int WSLOTS = 16 * XSLOTS, WPIECES = (WSLOTS + 63) / 64;       // a loader wave's 16 rows: slots, DMA pieces
  // Q / K sections and the O tile: row stride XS (b128 row reads only); V: K1's row stride (transposed reads: a 256-byte row
  // gets 80 more bytes, wattn_mfma_hd.hip)
  static constexpr int SEC = C * 2, LDT0 = ((SEC + 31) / 32) * 32;
  static constexpr int LDV = LDT0 % 256 == 0 ? LDT0 + 80 : (LDT0 / 16) % 2 == 0 ? LDT0 + 16 : LDT0;
  static constexpr int LDQ = XS;
  static constexpr int TABF = HEADS * 15 * TSX, TABB = TABF + 8;
  static constexpr int OFF_X = 0;
  static constexpr int OFF_Q = 2 * XBUFB, OFF_K = OFF_Q + 64 * LDQ, OFF_V = OFF_K + 64 * LDQ, OFF_O = OFF_V + 64 * LDV + 256;
  static constexpr int ZEROB = OFF_O + 64 * LDQ - OFF_Q;                          // Q, K, V, 256 spare bytes, O: zeroed once
  static constexpr int OFF_TAB = OFF_O + 64 * LDQ;
  static constexpr int OFF_WP = (OFF_TAB + (TABB + TABF + 8) * 4 + 1023) / 1024 * 1024;
  static constexpr int WPB = NTS * KS * 1024;
  static constexpr int OFF_SBQ = OFF_WP + WPB;                                    // [2][NTQ * 32] floats
  static constexpr int SBQB = (2 * NTQ * 32 * 4 + 1023) / 1024 * 1024;
  static constexpr int OFF_SBP = OFF_SBQ + SBQB;                                  // [2][NTS * 32] floats
  static constexpr int SBPB = (2 * NTS * 32 * 4 + 1023) / 1024 * 1024;
  static constexpr int OFF_ST = OFF_SBP + SBPB;                                   // [2 buffers][64][2] floats
  static constexpr int SMEM = OFF_ST + 2 * 64 * 2 * 4;
  static_assert(SMEM <= 160 * 1024, "LDS");
  // row copies to HBM: 16-byte chunks per row of C channels (a, x1) / of 3 C channels (qkv); bytes of the last chunk
  static constexpr int CPR = (SEC + 15) / 16, TAILB = SEC - 16 * (CPR - 1);
  static constexpr int CPR3 = (3 * SEC + 15) / 16, TAILB3 = 3 * SEC - 16 * (CPR3 - 1);
  static constexpr int sec_off(int sec) { return sec == 0 ? OFF_Q : sec == 1 ? OFF_K : OFF_V; }
  static constexpr int sec_ld(int sec) { return sec == 2 ? LDV : LDQ; }
};

// store the first `nb` (4, 8, 12 or 16) bytes of a 16-byte chunk; dst is dword aligned
__device__ __forceinline__ void store_chunk(char* dst, const u32x4_t& v, int nb) {
  if (nb >= 16) {
    u32x4_a4 o;
    o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
    *reinterpret_cast<u32x4_a4*>(dst) = o;
  } else if (nb == 12) {
    u32x3_a4 o;
    o.x = v.x; o.y = v.y; o.z = v.z;
    *reinterpret_cast<u32x3_a4*>(dst) = o;
  } else if (nb == 8) {
    u32x2_a4 o;
    o.x = v.x; o.y = v.y;
    *reinterpret_cast<u32x2_a4*>(dst) = o;
  } else {
    *reinterpret_cast<uint32_t*>(dst) = v.x;
  }
}

struct SaCtx {
  lds_cp Qp, Kp, Vp, Op;
  const LDS_AS f32x2* tb;
  int h;
  bool masked;
  float scale2;
  Pack16 mK[2], mQ;
};

// one head of one wave: 32 queries (lane & 31) x 64 keys; O (normalised) -> the head's channels of the wave's rows of the O tile
template <int D, int HD, class IO>
__device__ __forceinline__ void sa_head(const SaCtx& c, IO io) {
  using CF = SA<D>;
  constexpr int ldq = CF::LDQ, ldv = CF::LDV;
  constexpr int c_lo = HD * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16;
  constexpr int r_lo = c_lo & ~3;
  const int h = c.h;
  f32x16 X[2];
  const LDS_AS f32x2* tbh = c.tb + HD * (15 * TSX / 2);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = lds_read_f32x2(tbh + ((7 - (kt * 4 + (v >> 2))) * TSX + (v & 3)) / 2);
      X[kt][v] = b2.x;
      X[kt][v + 1] = b2.y;
    }
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    Pack16 qb = lds_pack(c.Qp + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      qb.w[e] &= h ? mB : mA;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const Pack16 ka = lds_pack(c.Kp + kt * 32 * ldq + t * 32);
      Mma<bf16>::mma(X[kt], ka, qb);
    }
  }
  if (c.masked) {   // shifted-window mask as one more k-step (wattn_mfma_hd.hip)
    Mma<bf16>::mma(X[0], c.mK[0], c.mQ);
    Mma<bf16>::mma(X[1], c.mK[1], c.mQ);
  }
  io(0);   // (row copies to HBM trickle out between the stages: a burst in front of the head blocks the wave at the store queue)
  float m = X[0][0];
#pragma unroll
  for (int v = 1; v < 16; ++v) m = __builtin_fmaxf(m, X[0][v]);
#pragma unroll
  for (int v = 0; v < 16; ++v) m = __builtin_fmaxf(m, X[1][v]);
  m = half_swap_max(m);
  const float nm = -c.scale2 * m;
  float l0 = 0.f, l1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v], c.scale2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v + 1], c.scale2, nm));
      l0 += e0;
      l1 += e1;
      X[kt][v] = e0;
      X[kt][v + 1] = e1;
    }
  Pack16 pb[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) pb[kt][s].w[e] = pack_bf16x2(X[kt][8 * s + 2 * e], X[kt][8 * s + 2 * e + 1]);
  const float inv = __builtin_amdgcn_rcpf(half_swap_sum(l0 + l1));
  io(1);
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const lds_cp vb = c.Vp + r_lo * 2 + (kt * 32 + 16 * s) * ldv;
      const Pack16 va = lds_tr_pack(vb, vb + 8 * ldv);
      Mma<bf16>::mma(acc, va, pb[kt][s]);   // rows = channels r_lo .. r_lo + 31 (V^T), cols = queries
    }
  io(2);
  store_tile_rows<r_lo, c_lo, c_hi>(c.Op, acc, inv, h);
}

template <int D>
__global__ void __launch_bounds__(64 * sa_nw(D), 3) swinattn_fwd_kernel(const SAArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = SA<D>;
  constexpr int SA_NW = sa_nw(D), SA_NTH = 64 * SA_NW;
  static_assert(SA_NW == 12 || (SA_NW == 6 && D == 10), "wave roles");
  constexpr int C = CF::C, KS = CF::KS, NTS = CF::NTS, NTQ = CF::NTQ, XS = CF::XS, ldq = CF::LDQ, ldv = CF::LDV;
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv < SA_NLW;
  char* Qs = smem + CF::OFF_Q;
  char* Ks = smem + CF::OFF_K;
  char* Vs = smem + CF::OFF_V;
  char* Os = smem + CF::OFF_O;
  float* tabL = reinterpret_cast<float*>(smem + CF::OFF_TAB);
  const float* sbqL = reinterpret_cast<const float*>(smem + CF::OFF_SBQ);
  const float* sbpL = reinterpret_cast<const float*>(smem + CF::OFF_SBP);
  float* statL = reinterpret_cast<float*>(smem + CF::OFF_ST);

  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void* ptr, uint32_t bytes) {
    u32x4s_t q;
    q.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    q.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32) & 0xffffu);
    q.z = __builtin_amdgcn_readfirstlane(bytes);
    q.w = 0x00020000u;
    return q;
  };
  const u32x4s_t rsx = make_rsrc(p.X, p.x_bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const u32x4s_t& rs, uint32_t ldst, uint32_t off) {   // inline asm: see conv3_mfma.hip
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rs) : "memory");
  };

  // ---- G1 work of this wave: tile tq, token halves [hf0, hf1) ---------------------------------------------------------
  //   D = 20: 12 tiles, wave = tile, both halves;  D = 10: 6 tiles x 2 halves = one item per wave;
  //   D = 15: 9 tiles: waves 0-5 a tile with both halves, waves 6-11 tiles 6-8 with one half each
  int tq, hf0, hf1;
  if constexpr (D == 20) { tq = wv; hf0 = 0; hf1 = 2; }
  else if constexpr (D == 10) {
    if constexpr (SA_NW == 6) { tq = wv; hf0 = 0; hf1 = 2; }
    else { tq = wv % 6; hf0 = wv / 6; hf1 = hf0 + 1; }
  }
  else {
    if (wv < 6) { tq = wv; hf0 = 0; hf1 = 2; }
    else { tq = 6 + (wv - 6) % 3; hf0 = (wv - 6) / 3; hf1 = hf0 + 1; }
  }
  // its weight fragments, loaded by inline asm before anything else is in flight (lin3_mfma.hip)
  typedef uint32_t u32x4v_t __attribute__((ext_vector_type(4)));
  u32x4v_t wfr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const char* src = reinterpret_cast<const char*>(p.Wq) + (((int64_t)tq * KS + ks) * 64 + lane) * 16;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wfr[ks]) : "v"(src) : "memory");
  }
  const int secq = tq / NTS, chq = (tq - secq * NTS) * 32;   // section and first channel of the tile

  const int nW = g.nWh * g.nWw;
  const int nwin = g.B * nW;
  struct WinPos { int b, wr, wc; };
  auto locate = [&](int win) {
    WinPos w;
    w.b = win / nW;
    const int wi = win - w.b * nW;
    w.wr = wi / g.nWw;
    w.wc = wi - w.wr * g.nWw;
    return w;
  };
  auto token = [&](const WinPos& w, int t) {   // row of token t (0 .. 63) of the window in the (B H W) activation
    int rr = w.wr * 8 + (t >> 3) + g.shift;
    if (rr >= g.H) rr -= g.H;
    int cc = w.wc * 8 + (t & 7) + g.shift;
    if (cc >= g.W) cc -= g.W;
    return (w.b * g.H + rr) * g.W + cc;
  };
  // loader wave wv puts rows 16 wv .. 16 wv + 15 of a window in flight: slot s of its region = (row 16 wv + s / XSLOTS, slot)
  const uint32_t ldxb = (uint32_t)p.ldx * 2u;
  auto issue = [&](const WinPos& w, int buf, int ln) {
#pragma unroll
    for (int i = 0; i < CF::WPIECES; ++i) {
      const int s = 64 * i + ln;
      const int rl = s / CF::XSLOTS, sl = s - rl * CF::XSLOTS;
      const uint32_t t = (uint32_t)token(w, (16 * wv + rl) & 63);
      const uint32_t off = sl < CF::XD ? t * ldxb + (uint32_t)(sl * 16) : 0xffffffffu;
      if (s < CF::WSLOTS)
        dma(rsx, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::OFF_X + buf * CF::XBUFB + 16 * wv * XS + i * 1024)), off);
    }
  };
  // LayerNorm statistics of the loader wave's own 16 rows (4 lanes per token, two passes: lin3_mfma.hip) -> LDS + HBM
  auto row_stats = [&](const WinPos& w, int buf, int ln) {
    const int tok = 16 * wv + (ln >> 2), part = ln & 3;
    const char* tb = smem + CF::OFF_X + buf * CF::XBUFB + tok * XS;
    constexpr int NSL = (CF::XD + 3) / 4;
    float xv[NSL][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int sl = part + 4 * i;
      const Pack16 q = *reinterpret_cast<const Pack16*>(tb + (sl < CF::XD ? sl : 0) * 16);
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
      statL[(buf * 64 + tok) * 2] = mean;
      statL[(buf * 64 + tok) * 2 + 1] = rstd;
      *reinterpret_cast<float2*>(p.stats + (int64_t)token(w, tok) * 2) = make_float2(mean, rstd);
    }
  };

  int win = blockIdx.x;
  WinPos cur = locate(win < nwin ? win : 0);
  if (loader) issue(cur, 0, lane);
  // ---- one-time LDS state -----------------------------------------------------------------------------------------------
  for (int i = tid * 16; i < CF::ZEROB; i += SA_NTH * 16) *reinterpret_cast<float4*>(Qs + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  {  // proj fragments, S / b' of both Linears: plain copies (L2-resident, once per workgroup)
    const u32x4_a4* src = reinterpret_cast<const u32x4_a4*>(p.Wp);
    for (int i = tid; i < CF::WPB / 16; i += SA_NTH) *reinterpret_cast<u32x4_a4*>(smem + CF::OFF_WP + i * 16) = src[i];
    for (int i = tid; i < 2 * NTQ * 32; i += SA_NTH) reinterpret_cast<float*>(smem + CF::OFF_SBQ)[i] = p.sbq[i];
    for (int i = tid; i < 2 * NTS * 32; i += SA_NTH) reinterpret_cast<float*>(smem + CF::OFF_SBP)[i] = p.sbp[i];
  }
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  {  // relative-position table / scale, x-reversed, two copies (wattn_mfma_hd.hip)
    constexpr int NT_SRC = 225 * 6, NLD = (NT_SRC + SA_NTH - 1) / SA_NTH;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + SA_NTH * k;
      tv[k] = p.table[j < NT_SRC ? j : NT_SRC - 1];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + SA_NTH * k;
      if (j < NT_SRC) {
        const int rel = j / 6, hd = j - rel * 6;
        const int dy = rel / 15, u = 14 - (rel - dy * 15);
        const float v = tv[k] * rscale;
        tabL[(hd * 15 + dy) * TSX + u] = v;
        if (u >= 1) tabL[CF::TABB + (hd * 15 + dy) * TSX + u - 1] = v;
      }
    }
  }
  // attention role: wave = (head hd, query half qt); with 6 waves a wave takes both query halves of its head, one after the other
  const int hd = SA_NW == 6 ? wv : wv >> 1;
  const int xi = r & 7;
  const int thr = g.ws - g.shift;
  SaCtx c;
  c.h = h;
  c.Kp = (lds_cp)(Ks + r * ldq + h * 16);
  {
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    c.Vp = (lds_cp)(Vs + (4 * h + q) * ldv + (16 * (gq & 1) + 4 * pp) * 2);
  }
  auto set_qt = [&](int qt) {   // the lane's query row: positions of its Q row, its O row and its bias-table row
    const int yi = qt * 4 + (r >> 3);
    c.Qp = (lds_cp)(Qs + (qt * 32 + r) * ldq + h * 16);
    c.Op = (lds_cp)(Os + (qt * 32 + r) * ldq);
    const int u0 = 4 * h - xi + 7;
    const float* tb = (u0 & 1) ? tabL + CF::TABB + yi * TSX + (u0 - 1) : tabL + yi * TSX + u0;
    c.tb = (const LDS_AS f32x2*)tb;
    return yi;
  };
  int yi = set_qt(SA_NW == 6 ? 0 : (wv & 1));
  const uint32_t cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.scale2 = p.scale * LOG2E;

  // every load of the prologue has landed (weights, table, proj image, the loader waves' rows of window 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wfr[ks]));
  if (loader && win < nwin) row_stats(cur, 0, lane);

  // whole-row copies LDS -> HBM: instruction i of a tile moves chunks 64 i .. 64 i + 63 of its 64 x CPR chunks; the waves
  // [w0, w0 + nwv) share the instructions
  auto copy_rows = [&](const char* tile, int ld, bf16* dstp, int64_t ldd, const WinPos& w, int ln, int w0, int nwv) {
    for (int i = wv - w0; i < CF::CPR; i += nwv) {
      const int f = 64 * i + ln;
      const int row = f / CF::CPR, chk = f - row * CF::CPR;
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(tile + row * ld + chk * 16);
      char* dst = reinterpret_cast<char*>(dstp + (int64_t)token(w, row) * ldd) + chk * 16;
      store_chunk(dst, v, (CF::TAILB == 16 || chk < CF::CPR - 1) ? 16 : CF::TAILB);
    }
  };
  // the window's qkv rows (3 C channels: the Q, K and V sections side by side), waves [w0, w0 + nwv)
  auto copy_qkv = [&](const WinPos& w, int ln, int w0, int nwv, int part, int nparts) {
    for (int i = wv - w0 + nwv * part; i < CF::CPR3; i += nwv * nparts) {
      const int f = 64 * i + ln;
      const int row = f / CF::CPR3, chk = f - row * CF::CPR3;
      const int b0 = chk * 16;
      u32x4_t v;
      if constexpr (CF::SEC % 16 == 0) {
        const int sec = (b0 >= CF::SEC ? 1 : 0) + (b0 >= 2 * CF::SEC ? 1 : 0);
        const char* src = smem + (sec == 0 ? CF::OFF_Q : sec == 1 ? CF::OFF_K : CF::OFF_V) + row * (sec == 2 ? ldv : ldq) + (b0 - sec * CF::SEC);
        v = *reinterpret_cast<const u32x4_t*>(src);
      } else {   // a chunk may straddle two sections: dword by dword
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          int b = b0 + 4 * d;
          b = b < 3 * CF::SEC ? b : 3 * CF::SEC - 4;
          const int sec = (b >= CF::SEC ? 1 : 0) + (b >= 2 * CF::SEC ? 1 : 0);
          const char* src = smem + (sec == 0 ? CF::OFF_Q : sec == 1 ? CF::OFF_K : CF::OFF_V) + row * (sec == 2 ? ldv : ldq) + (b - sec * CF::SEC);
          v[d] = *reinterpret_cast<const uint32_t*>(src);
        }
      }
      char* dst = reinterpret_cast<char*>(p.qkv + (int64_t)token(w, row) * p.ldq) + b0;
      store_chunk(dst, v, (CF::TAILB3 == 16 || chk < CF::CPR3 - 1) ? 16 : CF::TAILB3);
    }
  };

  int buf = 0;
  bool have_prev = false;
  WinPos prev = cur;
#ifdef SA_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter(), tn;
#define SA_ST(k) { tn = __builtin_readcyclecounter(); tacc[k] += tn - tl; tl = tn; }
#else
#define SA_ST(k)
#endif
  SA_ST(0)   // 0: prologue
  for (; win < nwin; win += p.G) {
    __syncthreads();   // top: x(n) + statistics in LDS; everybody is done with window n - 1
    SA_ST(1)   // 1: wait at the top barrier
    const WinPos w = cur;
    const int nxt = win + p.G;
    const bool more = nxt < nwin;
    cur = locate(more ? nxt : win);
    int lnw = tid;
    asm volatile("" : "+v"(lnw));
    lnw &= 63;
    char* xb = smem + CF::OFF_X + buf * CF::XBUFB;
    const float* st = statL + buf * 128;

    // ---- A: qkv tile(s) of this wave -> the sections ------------------------------------------------------------------------
    // both token halves' MFMA chains are issued before either epilogue: the second chain runs on the matrix pipe while the
    // first tile is finished on the vector ALUs (in program order chain, epilogue, chain, epilogue the pipe idles)
    // One item = (tile tq, token half hf).  Every LDS operand of the item — the 4 / 6 / 8 B fragments of the token rows, the
    // token's statistics, S and b' of the lane's 16 channels — is requested BEFORE the MFMA chain starts: written as a loop
    // the compiler issues two k-steps of reads, waits for them, issues two MFMAs, and so on: a wave then spends most of the
    // item in LDS round trips (measured with the per-phase stamps of the -DSA_STAMPS build: 5.5 thousand cycles per window in
    // this phase at C = 120, against 1.5 thousand of matrix-pipe time)
    auto g1_item = [&](int hf) {
      const int tok = hf * 32 + r;
      const char* brow = xb + tok * XS + h * 16;
      Pack16 bq[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bq[ks] = *reinterpret_cast<const Pack16*>(brow + ks * 32);
      __builtin_amdgcn_sched_barrier(0);   // (the reads above stay above the chain)
      if (C % 16 != 0) {   // the row's last slot ends with the next channels of the memory row: zero them
        constexpr int c0 = C % 16 < 8 ? C % 16 : 8, c1 = C % 16 > 8 ? C % 16 - 8 : 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) bq[KS - 1].w[d] = (2 * d < (h ? c1 : c0)) ? bq[KS - 1].w[d] : 0u;
      }
      f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < ((SA_ABL & 16) ? 0 : KS); ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wfr[ks]), __builtin_bit_cast(bf16x8_t, bq[ks]), acc, 0, 0, 0);
      // the epilogue's operands are requested while the chain runs
      const float2 mr = *reinterpret_cast<const float2*>(st + tok * 2);
      float4 S4[4], B4[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n0 = tq * 32 + 8 * g4 + 4 * h;
        S4[g4] = *reinterpret_cast<const float4*>(sbqL + n0);
        B4[g4] = *reinterpret_cast<const float4*>(sbqL + NTQ * 32 + n0);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float rstd = mr.y, nrm = -mr.y * mr.x;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        acc[4 * g4] = fmaf(rstd, acc[4 * g4], fmaf(nrm, S4[g4].x, B4[g4].x));
        acc[4 * g4 + 1] = fmaf(rstd, acc[4 * g4 + 1], fmaf(nrm, S4[g4].y, B4[g4].y));
        acc[4 * g4 + 2] = fmaf(rstd, acc[4 * g4 + 2], fmaf(nrm, S4[g4].z, B4[g4].z));
        acc[4 * g4 + 3] = fmaf(rstd, acc[4 * g4 + 3], fmaf(nrm, S4[g4].w, B4[g4].w));
      }
      char* srow = smem + (secq == 0 ? CF::OFF_Q : secq == 1 ? CF::OFF_K : CF::OFF_V) + tok * (secq == 2 ? ldv : ldq);
      bf16* qrow = p.qkv + (int64_t)token(w, tok) * p.ldq + secq * C;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float c8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * gp + e]), __float_as_uint(acc[8 * gp + 4 + e]), false, false);
          c8[e] = __uint_as_float(sw[0]);
          c8[4 + e] = __uint_as_float(sw[1]);
        }
        const int ch = chq + 8 * (2 * gp + h);   // the lane's 8 consecutive channels of the section (zeros from C on)
        u32x4_t u;
        u.x = pack_bf16x2(c8[0], c8[1]); u.y = pack_bf16x2(c8[2], c8[3]);
        u.z = pack_bf16x2(c8[4], c8[5]); u.w = pack_bf16x2(c8[6], c8[7]);
        *reinterpret_cast<u32x4_t*>(srow + ch * 2) = u;
        if (SA_QKV_DIRECT && !(SA_ABL & 1)) {
          const int nv = C - ch;
          if (nv >= 8) store_chunk(reinterpret_cast<char*>(qrow + ch), u, 16);
          else if (nv > 0) store_chunk(reinterpret_cast<char*>(qrow + ch), u, 2 * nv);   // 4 (C = 90) or 8 (C = 60) bytes
        }
      }
    };
    for (int hf = hf0; hf < hf1; ++hf) g1_item(hf);
    // the previous window's x1 rows (written in place of its x tile, phase C) and a rows (the O tile) -> HBM
    if (have_prev && !(SA_ABL & 4)) copy_rows(smem + CF::OFF_X + (buf ^ 1) * CF::XBUFB, XS, p.x1, p.ld1, prev, lnw, 0, SA_NW);
    if (!SA_ACOPY_C && have_prev && !(SA_ABL & 2)) copy_rows(Os, ldq, p.a, p.lda, prev, lnw, 0, SA_NW);
    SA_ST(2)   // 2: phase A work
    __syncthreads();   // B1: the window's q | k | v are in LDS; the other x buffer is free
    SA_ST(3)   // 3: wait at B1

    // ---- B: next rows in flight, qkv rows out, attention -------------------------------------------------------------------
    if (loader && more && !(SA_ABL & 64)) issue(cur, buf ^ 1, lnw);
    auto io = [&](int k) {
      if (!SA_QKV_DIRECT && !loader && !(SA_ABL & 1)) copy_qkv(w, lnw, SA_NLW, SA_NW - SA_NLW, k, 3);
    };
    {
      const bool mrow = g.shift > 0 && w.wr == g.nWh - 1, mcol = g.shift > 0 && w.wc == g.nWw - 1;
      c.masked = __builtin_amdgcn_readfirstlane((int)(mrow || mcol)) != 0;
      auto onehot = [&](int reg, uint32_t v, Pack16& q) {
        q.w[0] = h ? 0u : ((reg == 0 ? v : 0u) | (reg == 1 ? v << 16 : 0u));
        q.w[1] = h ? 0u : ((reg == 2 ? v : 0u) | (reg == 3 ? v << 16 : 0u));
        q.w[2] = 0u;
        q.w[3] = 0u;
      };
      const int rx = (mcol && xi >= thr) ? 1 : 0;
      if (c.masked) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) onehot(2 * ((mrow && kt * 4 + (r >> 3) >= thr) ? 1 : 0) + rx, 0x3f80u, c.mK[kt]);
      }
      for (int pass = 0; pass < (SA_NW == 6 ? 2 : 1); ++pass) {
        if (SA_NW == 6) yi = set_qt(pass);
        if (c.masked) onehot(2 * ((mrow && yi >= thr) ? 1 : 0) + rx, cbits, c.mQ);
        if (!(SA_ABL & 8))
        switch (hd) {
          case 0: sa_head<D, 0>(c, io); break;
          case 1: sa_head<D, 1>(c, io); break;
          case 2: sa_head<D, 2>(c, io); break;
          case 3: sa_head<D, 3>(c, io); break;
          case 4: sa_head<D, 4>(c, io); break;
          default: sa_head<D, 5>(c, io); break;
        }
      }
    }
    SA_ST(4)   // 4: phase B work
    __syncthreads();   // B2: the attention output a is in the O tile
    SA_ST(5)   // 5: wait at B2

    // ---- C: a rows out, proj + shortcut (x1 in place of x); the loader waves: statistics of the next window ------------
    if (loader) {
      // the next window's rows of this wave have landed (issued an attention phase ago)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (more) row_stats(cur, buf ^ 1, lnw);
      if (SA_ACOPY_C && !(SA_ABL & 2)) copy_rows(Os, ldq, p.a, p.lda, w, lnw, 0, SA_NLW);
    }
    {
      // the proj items go to the last waves of the workgroup (12 waves: 4 .. 11, none of them a loader; 6 waves: 2 .. 5)
      constexpr int P0 = SA_NW == 12 ? SA_NLW : SA_NW - 2 * NTS;
      const int sw = wv - P0;
      if (sw >= 0 && sw < 2 * NTS) {   // item sw = (tile j, token half)
        const int j = sw % NTS, hf = sw / NTS;
        const int tok = hf * 32 + r;
        const char* brow = Os + tok * ldq + h * 16;
        const char* wrow = smem + CF::OFF_WP + (j * KS * 64 + lnw) * 16;
        char* xr = xb + tok * XS;
        // LDS operands ahead of the chain (see phase A), in two batches of k-steps: the second batch is requested while the
        // first half of the chain runs; the epilogue's operands while the second half runs
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        constexpr int KH = (KS + 1) / 2;
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
          Pack16 aw[KH], bq[KH];
#pragma unroll
          for (int i = 0; i < KH; ++i) {
            const int ks = bt * KH + i;
            if (ks < KS) {
              aw[i] = *reinterpret_cast<const Pack16*>(wrow + ks * 1024);
              bq[i] = *reinterpret_cast<const Pack16*>(brow + ks * 32);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < KH; ++i)
            if (bt * KH + i < ((SA_ABL & 32) ? 0 : KS)) Mma<bf16>::mma(acc, aw[i], bq[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
        float4 B4[4];
        u32x2_a4 rr[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = j * 32 + 8 * g4 + 4 * h;
          B4[g4] = *reinterpret_cast<const float4*>(sbpL + NTS * 32 + n0);
          rr[g4] = *reinterpret_cast<const u32x2_a4*>(xr + n0 * 2);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = j * 32 + 8 * g4 + 4 * h;
          const float y0 = acc[4 * g4] + B4[g4].x + bf16lo(rr[g4].x), y1 = acc[4 * g4 + 1] + B4[g4].y + bf16hi(rr[g4].x);
          const float y2 = acc[4 * g4 + 2] + B4[g4].z + bf16lo(rr[g4].y), y3 = acc[4 * g4 + 3] + B4[g4].w + bf16hi(rr[g4].y);
          u32x2_a4 o;
          o.x = pack_bf16x2(y0, y1); o.y = pack_bf16x2(y2, y3);
          if (n0 < C) *reinterpret_cast<u32x2_a4*>(xr + n0 * 2) = o;   // x1 where x was (this item's own 4 channels of its own token)
        }
      }
    }
    prev = w;
    have_prev = true;
    buf ^= 1;
    SA_ST(6)   // 6: phase C work
  }
  __syncthreads();
#ifdef SA_STAMPS
  if (p.stamps && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.stamps[((size_t)blockIdx.x * SA_NW + wv) * 8 + k] = tacc[k];
  }
#endif
  if (have_prev && !(SA_ABL & 4)) {   // the last window's x1 rows
    int lnw = tid;
    asm volatile("" : "+v"(lnw));
    lnw &= 63;
    copy_rows(smem + CF::OFF_X + (buf ^ 1) * CF::XBUFB, XS, p.x1, p.ld1, prev, lnw, 0, SA_NW);
    if (!SA_ACOPY_C && !(SA_ABL & 2)) copy_rows(Os, ldq, p.a, p.lda, prev, lnw, 0, SA_NW);
  }
}

template <int D>
int launch_sa(SAArgs& p, hipStream_t st) {
  using CF = SA<D>;
  constexpr int SA_NW = sa_nw(D), SA_NTH = 64 * SA_NW;
  constexpr int WGCU = SA_NW == 6 ? 2 : 1;   // workgroups per CU
  static_assert(WGCU * CF::SMEM <= 160 * 1024, "LDS per CU");
  auto kern = swinattn_fwd_kernel<D>;
  // (per launch: the attribute is per DEVICE, a process-wide "done" flag would leave a second GPU without it)
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  int64_t G = 256 * WGCU;
  if (G > nwin) G = nwin;
  p.G = (int)G;
#ifdef SA_STAMPS
  {
    static int left = -1;
    if (left < 0) { const char* e = getenv("SA_STAMPS_N"); left = e ? atoi(e) : 0; }
    if (left > 0) {
      const size_t n = (size_t)G * SA_NW * 8;
      (void)hipMalloc((void**)&p.stamps, n * 8);
      (void)hipMemsetAsync(p.stamps, 0, n * 8, st);
      hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(SA_NTH), CF::SMEM, st, p);
      (void)hipStreamSynchronize(st);
      unsigned long long* hst = (unsigned long long*)malloc(n * 8);
      (void)hipMemcpy(hst, p.stamps, n * 8, hipMemcpyDeviceToHost);
      (void)hipFree(p.stamps);
      if (--left == 0) {
        static const char* nm[8] = {"prologue", "wait top", "A work", "wait B1", "B work", "wait B2", "C work", "-"};
        fprintf(stderr, "[K8 stamps D=%d grid=%d] mean cycles per workgroup over the launch, by wave:\n", D, (int)G);
        for (int w = 0; w < SA_NW; ++w) {
          fprintf(stderr, "  wave %2d:", w);
          for (int k = 0; k < 7; ++k) {
            double sum = 0;
            for (int b = 0; b < (int)G; ++b) sum += (double)hst[((size_t)b * SA_NW + w) * 8 + k];
            fprintf(stderr, " %s %.0f", nm[k], sum / (double)G);
          }
          fprintf(stderr, "\n");
        }
      }
      free(hst);
      return rdst_launch_status("swinattn_fwd");
    }
    p.stamps = nullptr;
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)G), dim3(SA_NTH), CF::SMEM, st, p);
  return rdst_launch_status("swinattn_fwd");
}

}  // namespace

size_t swinattn_pack_bytes(int C) {
  return lin3sec_pack_bytes(C, 3 * C, 3) + lin3_pack_bytes(C, C);
}
bool swinattn_supported(int C, int heads, int ws) { return ws == 8 && heads == 6 && (C == 60 || C == 90 || C == 120); }

int swinattn_pack_launch(const float* ln_w, const float* ln_b, const float* Wqkv, const float* bqkv, const float* Wproj,
                         const float* bproj, void* out, int C, hipStream_t st);

// bf16, ws 8, 6 heads of dim 10 / 15 / 20, scale > 0; RDST_ENOTSUP otherwise.  wpack: swinattn_pack_bytes(C) bytes = [sectioned qkv
// image][proj image] (pack.h), written here unless `prepacked`.
int swinattn_fwd_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* Wqkv, const float* bqkv,
                      const float* table, const float* Wproj, const float* bproj, bf16* qkv, int64_t ldq, bf16* a, int64_t lda,
                      bf16* x1, int64_t ld1, float* stats, void* wpack, bool prepacked, const WinGeom& g, float scale, hipStream_t st) {
  const int C = g.C;
  if (!swinattn_supported(C, g.heads, g.ws) || g.mask || !(scale > 0.f) || !wpack || ((uintptr_t)wpack & 15)) return RDST_ENOTSUP;
  if (((uintptr_t)X & 3) || (ldx & 1) || ((uintptr_t)qkv & 3) || (ldq & 1) || ((uintptr_t)a & 3) || (lda & 1) || ((uintptr_t)x1 & 3) || (ld1 & 1))
    return RDST_ENOTSUP;
  const int64_t M = (int64_t)g.B * g.H * g.W;
  const int64_t xb = ((M - 1) * ldx + C) * 2;
  if (xb >= (1ll << 31) || M * ldq * 2 >= (1ll << 31)) return RDST_ENOTSUP;
  const int ks = (C + 15) / 16, nts = (C + 31) / 32, ntq = 3 * nts;
  char* img = reinterpret_cast<char*>(wpack);
  char* img2 = img + lin3sec_pack_bytes(C, 3 * C, 3);
  if (!prepacked)
    if (int rc = swinattn_pack_launch(ln_w, ln_b, Wqkv, bqkv, Wproj, bproj, wpack, C, st)) return rc;
  SAArgs p{};
  p.X = X; p.ldx = ldx; p.x_bytes = (uint32_t)xb;
  p.Wq = reinterpret_cast<const bf16*>(img); p.sbq = reinterpret_cast<const float*>(img + (size_t)ntq * ks * 1024);
  p.Wp = reinterpret_cast<const bf16*>(img2); p.sbp = reinterpret_cast<const float*>(img2 + (size_t)nts * ks * 1024);
  p.table = table; p.qkv = qkv; p.ldq = ldq; p.a = a; p.lda = lda; p.x1 = x1; p.ld1 = ld1; p.stats = stats; p.g = g; p.scale = scale;
  if (C == 60) return launch_sa<10>(p, st);
  if (C == 90) return launch_sa<15>(p, st);
  return launch_sa<20>(p, st);
}

namespace {
__global__ void __launch_bounds__(256) swinattn_pack_kernel(const float* ln_w, const float* ln_b, const float* Wqkv, const float* bqkv,
                                                            const float* Wproj, const float* bproj, char* out, int C, int nb_q) {
  const int ks = (C + 15) / 16, nts = (C + 31) / 32, ntq = 3 * nts;
  if ((int)blockIdx.x < nb_q) {
    lin3sec_pack_block((int)blockIdx.x, Wqkv, ln_w, ln_b, bqkv, reinterpret_cast<bf16*>(out),
                       reinterpret_cast<float*>(out + (size_t)ntq * ks * 1024), 3 * C, C, 3, 1.0f);
  } else {
    char* o2 = out + lin3sec_pack_bytes(C, 3 * C, 3);
    lin3_pack_block((int)blockIdx.x - nb_q, Wproj, nullptr, nullptr, bproj, reinterpret_cast<bf16*>(o2),
                    reinterpret_cast<float*>(o2 + (size_t)nts * ks * 1024), C, C, ks, nts, 1.0f);
  }
}
}  // namespace

int swinattn_pack_launch(const float* ln_w, const float* ln_b, const float* Wqkv, const float* bqkv, const float* Wproj,
                         const float* bproj, void* out, int C, hipStream_t st) {
  const int nbq = lin3sec_pack_blocks(C, 3 * C, 3), nbp = lin3_pack_blocks(C, C);
  hipLaunchKernelGGL(swinattn_pack_kernel, dim3((unsigned)(nbq + nbp)), dim3(256), 0, st, ln_w, ln_b, Wqkv, bqkv, Wproj, bproj,
                     reinterpret_cast<char*>(out), C, nbq);
  return rdst_launch_status("swinattn_pack");
}
