// Window attention for 16x16 windows (N = 256 tokens per window) on the matrix cores: bf16, 6 heads of dim
// 10 / 15 / 20 (the RDST widths C = 60 / 90 / 120 with window_size 16 — BASELINE.json configs[3], the SwinIR-style
// large-window configuration; reference: networks/swin_transformer_sr.py:110-141 with N = 256, mask :211-232).
//
// What changes against the 8x8 kernels (wattn_mfma_hd.hip): a query now has 256 keys, so
//   * the Q/K/V rows of a whole window do not fit the LDS for all heads (256 x 3 x 240 B = 184 KB at C = 120):
//     a workgroup owns (window, head PAIR) — 40 / 60 / 80 B pieces of each row section, 37 - 61 KB of LDS — and the
//     three workgroups of a window are placed on the same XCD so the shared cache lines are fetched from HBM once;
//   * a wave owns a query tile (32 queries on the lanes) and ALL 256 keys of a head live in its accumulator
//     registers (8 tiles x 16 registers): S^T = K.Q^T with bias/scale as the initial accumulator, the shift mask as one
//     more k-step of one-hot region vectors, softmax entirely in registers (no online rescaling: every key is
//     there), P^T repacked as the B operand of O^T = V^T.P^T; 4 waves x 2 query tiles, two workgroups per CU;
//   * the relative-position table of a head (31 x 31) is staged REVERSED in both directions, so that the bias of
//     key (yj, xj) for the lane's query sits at lane base + compile-time offset, in two copies shifted by one float
//     (every lane reads aligned 8-byte pairs).
// The kernel is bound by the softmax on the vector ALUs (256 x 256 x 6 exponentials per window), not by HBM.
#include "wattn_hd.h"

namespace {
using namespace wahd;

struct W16Args {
  const bf16* qkv; int64_t ld;
  const float* table;
  bf16* out; int64_t ldo;
  const bf16* dout; int64_t ldd;     // backward only
  bf16* dqkv; int64_t ldq;
  float* slab;                        // backward: [window][heads][961] partial d(table)
  WinGeom g;
  float scale;
};

template <int D>
struct W16 {
  static constexpr int HEADS = 6, HPG = 2, NG = 3;
  static constexpr int C = HEADS * D;
  static constexpr int GC = HPG * D;                 // channels of a head pair
  static constexpr int PB = GC * 2;                  // bytes of its piece of a row section: 40 / 60 / 80
  static constexpr int GRAN = D == 10 ? 8 : D == 15 ? 12 : 16;  // chunk size (D = 15: 12-B chunks, dword aligned)
  static constexpr int CPS = PB / GRAN;              // chunks per piece
  static constexpr int LDT = D == 10 ? 48 : 80;      // LDS row stride: odd number of 16-B slots
  static constexpr int SEC = 256 * LDT;              // bytes of one staged section
  static constexpr int TROW = 32;                    // floats per staged table row (31 used)
  static constexpr int TABF = 31 * TROW;             // floats per head and copy
  static constexpr int NKS = (GC + 15) / 16;         // k-steps over the pair's channels
};

__device__ __forceinline__ int64_t win_token16(int b, int wr, int wc, int t, const WinGeom& g) {
  int r = wr * 16 + (t >> 4) + g.shift;
  if (r >= g.H) r -= g.H;
  int c = wc * 16 + (t & 15) + g.shift;
  if (c >= g.W) c -= g.W;
  return ((int64_t)b * g.H + r) * g.W + c;
}

// workgroup -> (window, head pair): the three pairs of a window run on the same XCD (blockIdx round-robins over 8 XCDs)
__device__ __forceinline__ void w16_locate(int nwin, int& win, int& grp) {
  const int b = blockIdx.x;
  if ((nwin & 7) == 0) {
    const int xcd = b & 7, slot = b >> 3;
    grp = slot % 3;
    win = (slot / 3) * 8 + xcd;
  } else {
    grp = b % 3;
    win = b / 3;
  }
}

struct W16Ctx {
  lds_cp Qp, Kp, Vp, Orow;
  const LDS_AS f32x2* tb;   // lane base into the reversed table copy of its parity (head 0 of the pair)
  int h, r;
  bool masked, mrow, mcol;
  int thr, qt;
  float scale2;
  uint32_t cbits;
};

__device__ __forceinline__ void onehot4(int reg, uint32_t v, int h, Pack16& q) {   // k = 0..3 of lane half 0
  q.w[0] = h ? 0u : ((reg == 0 ? v : 0u) | (reg == 1 ? v << 16 : 0u));
  q.w[1] = h ? 0u : ((reg == 2 ? v : 0u) | (reg == 3 ? v << 16 : 0u));
  q.w[2] = 0u;
  q.w[3] = 0u;
}

// S'^T tiles of one (query tile, head): X[kt][v] = q.k + bias/scale (+ 100/scale where the shift regions agree);
// key j = 32 kt + acc_row(v, h) = (yj = 2 kt + (v >> 3), xj = 8 ((v >> 2) & 1) + 4 h + (v & 3)), query = the lane's.
template <int D, int HL>
__device__ __forceinline__ void w16_scores(f32x16 (&X)[8], const W16Ctx& c) {
  using CF = W16<D>;
  constexpr int ldt = CF::LDT;
  constexpr int c_lo = HL * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16;
  const LDS_AS f32x2* tbh = c.tb + HL * (2 * CF::TABF / 2);
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = tbh[((2 * kt + (v >> 3)) * CF::TROW + 8 * ((v >> 2) & 1) + (v & 3)) / 2];
      X[kt][v] = b2.x;
      X[kt][v + 1] = b2.y;
    }
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    Pack16 qb = lds_pack(c.Qp + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      qb.w[e] &= c.h ? mB : mA;
    }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const Pack16 ka = lds_pack(c.Kp + kt * 32 * ldt + t * 32);
      Mma<bf16>::mma(X[kt], ka, qb);
    }
  }
  if (c.masked) {   // wave-uniform: last window row / column of a shifted block
    const int xi = c.r & 15, yq = 2 * c.qt + (c.r >> 4);
    const int rx = (c.mcol && xi >= c.thr) ? 1 : 0;   // r & 15 is also the key column of row r of a key tile
    Pack16 mQ;
    onehot4(2 * ((c.mrow && yq >= c.thr) ? 1 : 0) + rx, c.cbits, c.h, mQ);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      Pack16 mK;
      onehot4(2 * ((c.mrow && 2 * kt + (c.r >> 4) >= c.thr) ? 1 : 0) + rx, 0x3f80u, c.h, mK);
      Mma<bf16>::mma(X[kt], mK, mQ);
    }
  }
}

// One head of one query tile, forward: softmax over the 256 keys in registers, O^T = V^T.P^T, rows [c_lo, c_hi) of the
// result overwrite the (dead) Q channels of the wave's own query rows.
template <int D, int HL>
__device__ __forceinline__ void w16_head_fwd(const W16Ctx& c) {
  using CF = W16<D>;
  constexpr int ldt = CF::LDT;
  constexpr int c_lo = HL * D, c_hi = c_lo + D;
  constexpr int RL = c_lo & ~3;   // first channel row of the output tile (8-byte aligned for the transposed reads)
  f32x16 X[8];
  w16_scores<D, HL>(X, c);
  float m = X[0][0];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) m = __builtin_fmaxf(m, X[kt][v]);
  m = half_swap_max(m);
  const float nm = -c.scale2 * m;
  float l0 = 0.f, l1 = 0.f;
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  typedef LDS_AS s16x4_t* lds_tr_p;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v], c.scale2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v + 1], c.scale2, nm));
      l0 += e0;
      l1 += e1;
      X[kt][v] = e0;
      X[kt][v + 1] = e1;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {   // un-normalised P^T as the B operand: element jj of lane half h is key 16s + 8(jj>>2) + 4h + (jj&3)
      Pack16 pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) pb.w[e] = pack_bf16x2(X[kt][8 * s + 2 * e], X[kt][8 * s + 2 * e + 1]);
      const lds_cp vb = c.Vp + RL * 2 + (kt * 32 + 16 * s) * ldt;
      const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(vb));
      const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(vb + 8 * ldt));
      const u32x2_t u0 = __builtin_bit_cast(u32x2_t, b0), u1 = __builtin_bit_cast(u32x2_t, b1);
      Pack16 va;
      va.w[0] = u0.x; va.w[1] = u0.y; va.w[2] = u1.x; va.w[3] = u1.y;
      Mma<bf16>::mma(acc, va, pb);   // rows = channels RL.. (V^T), cols = queries
    }
  }
  const float inv = __builtin_amdgcn_rcpf(half_swap_sum(l0 + l1));
  store_tile_rows<RL, c_lo, c_hi>(c.Orow, acc, inv, c.h);
}

template <int D>
__global__ void __launch_bounds__(256, 2) wattn16_fwd_kernel(const W16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = W16<D>;
  using CH = typename Chunk<CF::GRAN>::type;
  constexpr int ldt = CF::LDT, GRAN = CF::GRAN, CPS = CF::CPS, CPR = 3 * CPS, RPI = 64 / CPR, NI = 64 / RPI;
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Qs = smem;
  char* Ks = Qs + CF::SEC;
  char* Vs = Ks + CF::SEC;
  float* tabL = reinterpret_cast<float*>(Vs + CF::SEC + 64);   // [head of the pair][copy A | copy B][31][32]

  const int nW = g.nWh * g.nWw, nwin = g.B * nW;
  int win, grp;
  w16_locate(nwin, win, grp);
  const int b = win / nW, wi = win - b * nW, wr = wi / g.nWw, wc = wi - wr * g.nWw;

  // ---- this wave's 64 token rows -> registers (all in flight), then the table, then LDS
  const int lr0 = lane / CPR, ch = lane - lr0 * CPR;
  const bool act = lr0 < RPI;
  const int lr = act ? lr0 : RPI - 1;
  const int sec = ch / CPS, cw = ch - sec * CPS;
  CH regs[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int t = wv * 64 + i * RPI + lr;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld) + sec * (CF::C * 2) + grp * CF::PB + cw * GRAN;
    regs[i] = *reinterpret_cast<const CH*>(src);
  }
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  {  // table of the pair's two heads, reversed in y and x: A[dy'][u'] = T[30 - dy'][30 - u'] / scale, B[k] = A[k + 1]
    constexpr int NSRC = 2 * 961, NLD = (NSRC + 255) / 256;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 256 * k;
      const int jj = j < NSRC ? j : NSRC - 1;
      const int hl = jj >= 961 ? 1 : 0, rel = jj - 961 * hl;
      tv[k] = p.table[rel * CF::HEADS + grp * 2 + hl];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 256 * k;
      if (j < NSRC) {
        const int hl = j >= 961 ? 1 : 0, rel = j - 961 * hl;
        const int ry = rel / 31, rx = rel - ry * 31;
        const int idx = (30 - ry) * CF::TROW + (30 - rx);
        const float v = tv[k] * rscale;
        float* A = tabL + hl * 2 * CF::TABF;
        A[idx] = v;
        if (idx >= 1) A[CF::TABF + idx - 1] = v;
      }
    }
  }
  if constexpr (ldt > CF::PB) {   // zero the pad bytes of every row: a padded k-step must read zeros (0 x NaN)
    constexpr int padw = (ldt - CF::PB) / 4;
    for (int idx = tid; idx < 3 * 256 * padw; idx += 256) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(Qs + (size_t)row * ldt + CF::PB + 4 * w) = 0u;
    }
  }
  if (tid < 16) *reinterpret_cast<uint32_t*>(Vs + CF::SEC + 4 * tid) = 0u;   // guard behind the last V row
  if (act) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = wv * 64 + i * RPI + lr;
      chunk_to_lds<CH>(smem + sec * CF::SEC + t * ldt + cw * GRAN, regs[i]);
    }
  }
  __syncthreads();

  W16Ctx c;
  c.h = h; c.r = r;
  c.Kp = (lds_cp)(Ks + r * ldt + h * 16);
  {
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    c.Vp = (lds_cp)(Vs + (4 * h + q) * ldt + (16 * (gq & 1) + 4 * pp) * 2);
  }
  c.thr = g.ws - g.shift;
  c.mrow = g.shift > 0 && wr == g.nWh - 1;
  c.mcol = g.shift > 0 && wc == g.nWw - 1;
  c.masked = c.mrow || c.mcol;
  c.cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.scale2 = p.scale * LOG2E;
#pragma unroll 1
  for (int qi = 0; qi < 2; ++qi) {
    const int qt = wv + 4 * qi;
    c.qt = qt;
    c.Qp = (lds_cp)(Qs + (qt * 32 + r) * ldt + h * 16);
    c.Orow = (lds_cp)(Qs + (qt * 32 + r) * ldt);
    {
      const int yi = 2 * qt + (r >> 4), xi = r & 15;
      const int u0 = (15 - yi) * CF::TROW + 15 - xi + 4 * h;
      const float* tb = (u0 & 1) ? tabL + CF::TABF + (u0 - 1) : tabL + u0;
      c.tb = (const LDS_AS f32x2*)tb;
    }
    w16_head_fwd<D, 0>(c);
    w16_head_fwd<D, 1>(c);
  }
  __syncthreads();
  // O (in the Q section) -> global rows: this wave's 64 query rows (tiles wv and wv + 4), CPS chunks per row
#pragma unroll 1
  for (int idx = lane; idx < 64 * CPS; idx += 64) {
    const int row = idx / CPS, k = idx - row * CPS;
    const int t = (row < 32 ? wv * 32 : (wv + 4) * 32 - 32) + row;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    char* dst = reinterpret_cast<char*>(p.out + tok * p.ldo) + grp * CF::PB + k * GRAN;
    *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(Qs + (size_t)t * ldt + k * GRAN);
  }
}

template <int D>
constexpr size_t w16_fwd_smem() { return (size_t)3 * W16<D>::SEC + 64 + (size_t)2 * 2 * W16<D>::TABF * 4; }

template <int D>
int launch_fwd16(const W16Args& p, hipStream_t st) {
  auto kern = wattn16_fwd_kernel<D>;
  constexpr size_t smem = w16_fwd_smem<D>();
  static_assert(smem <= 80 * 1024, "two workgroups per CU");
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  hipLaunchKernelGGL(kern, dim3((unsigned)(3 * nwin)), dim3(256), smem, st, p);
  return rdst_launch_status("wattn16_fwd");
}

bool al(const void* a, int64_t lda_bytes, int gsz) { return (uintptr_t)a % gsz == 0 && lda_bytes % gsz == 0; }

}  // namespace

// bf16, ws 16, 6 heads of dim 10 / 15 / 20, no explicit mask, scale > 0; RDST_ENOTSUP otherwise
int wattn16_fwd_mfma(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                     float scale, hipStream_t st) {
  if (g.ws != 16 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6) return RDST_ENOTSUP;
  if ((int64_t)g.B * g.nWh * g.nWw * 3 > 0x7fffffff) return RDST_ENOTSUP;
  W16Args p{};
  p.qkv = (const bf16*)qkv; p.ld = ld; p.table = table; p.out = (bf16*)out; p.ldo = ldo; p.g = g; p.scale = scale;
  const int d = g.C / 6;
  if (d == 10 && al(qkv, ld * 2, 8) && al(out, ldo * 2, 8)) return launch_fwd16<10>(p, st);
  if (d == 15 && al(qkv, ld * 2, 4) && al(out, ldo * 2, 4)) return launch_fwd16<15>(p, st);
  if (d == 20 && al(qkv, ld * 2, 16) && al(out, ldo * 2, 16)) return launch_fwd16<20>(p, st);
  return RDST_ENOTSUP;
}
