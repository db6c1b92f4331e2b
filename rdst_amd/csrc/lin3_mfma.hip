// K3 forward for the E1 shapes (bf16): y = (LayerNorm(x) | x) @ W^T + b (* scale) (+ residual), written as an HBM
// STREAMING kernel.  The arithmetic is nothing (4 steps of 128 tokens per CU, a few thousand cycles): what a launch
// costs is moving (K + N) * 2 bytes per token, and the old kernel (linear_mfma.hip) spent a third of it restaging and
// converting the fp32 weights in every workgroup and the rest in a slab loop with one 32-token slab in flight per wave.
//   * weights live in REGISTERS for the whole kernel: a tiny pack kernel writes them once per call as bf16 MFMA
//     A-operand fragments (32 outputs x 16 inputs), already multiplied by the LayerNorm gamma and the output scale;
//     a wave loads the 1-5 output tiles it owns with plain 16-B loads (12-30 fragments);
//   * LayerNorm does not touch the tile: with W' = W diag(gamma),  y[n] = rstd (sum_k W'[n][k] x[k] - mean S[n]) + b'[n],
//     S[n] = sum_k W'[n][k] (of the ROUNDED bf16 values), b' = b + W beta — the GEMM runs on the raw rows as they come
//     from HBM and (mean, rstd) enter in the epilogue; they are computed by a two-pass sum over the LDS-resident row
//     (4 lanes per token) and written out for the backward;
//   * a workgroup (8 waves) takes ALL its 128-token tiles (up to 4, 18-30 KB each) into LDS at once with LDS-DMA
//     (buffer_load ... lds; out-of-range lanes — rows past M, pad slots — write zeros), so every CU has its whole input
//     in flight from the first cycle, then walks the tiles: B fragments are ds_read_b128 of the token rows (pixel stride
//     = odd number of 16-B slots), accumulators are transposed (output channel in the registers, token on the lane) and
//     leave as 16-B row stores after one v_permlane32_swap per register pair; the residual rows are read the same way.
// Shapes: K = 60 / 90 / 120 with N = 3K (norm1 + qkv), N = K (proj + shortcut), N = 30 (dense tail: LayerNorm +
// Linear into the dense buffer).  Everything else stays on linear_mfma.hip.
#include "linear.h"
#include "mfma.h"
#include "pack.h"

namespace {

constexpr int L3_TT = 128;   // tokens per tile

constexpr int l3_gcd(int a, int b) { return b == 0 ? a : l3_gcd(b, a % b); }
constexpr int l3_stride(int K) {  // bytes: covers every k-step (no read leaves the token's row), odd number of 16-B slots
  int s = (K + 15) / 16 * 32;
  if (((s / 16) & 1) == 0) s += 16;
  return s;
}

__global__ void __launch_bounds__(256) lin3_pack_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ bias,
                                                        bf16* __restrict__ wp, float* __restrict__ sb, int N, int K, int ksteps,
                                                        int ntiles, float s) {
  lin3_pack_block((int)blockIdx.x, W, gamma, beta, bias, wp, sb, N, K, ksteps, ntiles, s);
}

struct L3Args {
  const bf16* X; int64_t ldx; int x_bytes;
  const bf16* Wp; const float* sb;
  const bf16* R; int64_t ldr; int r_bytes;
  bf16* Y; int64_t ldy;
  float* stats;
  int M, N, ntiles;
};

constexpr int l3_nw(int NT) { return (NT >= 4 && NT <= 12) ? NT : 8; }

template <int K, int NT, bool LN, bool RES>
struct L3Cfg {
  static constexpr int KSTEPS = (K + 15) / 16;
  static constexpr int XS = l3_stride(K), XSLOTS = XS / 16, XD = (2 * K + 15) / 16;
  static constexpr int TP = (L3_TT * XSLOTS + 63) / 64, TILEB = TP * 1024;
  // residual tile (RES): [128 tokens][NT * 32 channels] bf16, 8 slots per 32 channels, + 1 slot: odd slot count
  static constexpr int RS = RES ? NT * 64 + 16 : 0, RSLOTS = RS / 16, RP = RES ? (L3_TT * RSLOTS + 63) / 64 : 0, RTILEB = RP * 1024;
  // tiles in flight per workgroup; up to 8 waves, two workgroups share a CU (more waves to hide the epilogues)
  static constexpr int WGCU = l3_nw(NT) > 8 ? 1 : 2;
  static constexpr int NBUF = RES ? 1 : (WGCU == 1 ? 4 : 2);
  // waves: one output tile per wave where there are 4..12 of them (4 items = the 4 token sub-tiles, one fragment set),
  // else 8 waves dealing the NT * 4 items round-robin
  static constexpr int NW = l3_nw(NT);
  static constexpr int NTHR = 64 * NW;
  static constexpr int ITEMS = NT * 4, NJ = (ITEMS + NW - 1) / NW;
  static constexpr int PERIOD = NT / l3_gcd(NW, NT), ND = NJ < PERIOD ? NJ : PERIOD;
  static constexpr int CNT = (TP + NW - 1) / NW + (RES ? (RP + NW - 1) / NW : 0);   // DMA pieces per wave and tile
  static constexpr int R_OFF = NBUF * TILEB;
  static constexpr int STAT_OFF = R_OFF + NBUF * RTILEB;              // [NBUF][128][2] floats
  static constexpr int SB_OFF = STAT_OFF + NBUF * L3_TT * 2 * 4;      // [2][NT*32] floats
  static constexpr int SMEM = SB_OFF + (2 * NT * 32 * 4 + 1023) / 1024 * 1024;
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(CNT * (NBUF - 1) < 64, "vmcnt is a 6-bit counter");
  static_assert(TP >= NW && (!RES || RP >= NW), "every wave owns at least one piece of a tile");
};

template <int K, int NT, bool LN, bool RES>
__global__ void __launch_bounds__(64 * l3_nw(NT), l3_nw(NT) > 8 ? 1 : (2 * l3_nw(NT) + 3) / 4) lin3_kernel(const L3Args p) {
  using CF = L3Cfg<K, NT, LN, RES>;
  constexpr int KSTEPS = CF::KSTEPS, XS = CF::XS, NBUF = CF::NBUF, ND = CF::ND, NJ = CF::NJ, NW = CF::NW, NTHR = CF::NTHR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* statL = reinterpret_cast<float*>(smem + CF::STAT_OFF);
  const float* sbL = reinterpret_cast<const float*>(smem + CF::SB_OFF);

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
                 rsb = make_rsrc(p.sb, 2 * NT * 32 * 4);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const u32x4s_t& rs, uint32_t ldst, int off) {   // inline asm: see conv3_mfma.hip
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rs) : "memory");
  };
  const int grid = gridDim.x;
  // ---- this wave's weight fragments: item j of the wave is (nt, tt) = ((wave + NW j) % NT, (wave + NW j) / NT) ------
  // Loaded BEFORE the tiles go in flight (memory operations retire in issue order, so every later wait for a tile also
  // covers them) and by inline asm: the compiler must not know about any load here, or it drains the whole queue
  // (vmcnt(0)) at the first use.  The same goes for spill reloads: the kernel must not spill.
  typedef uint32_t u32x4v_t __attribute__((ext_vector_type(4)));
  u32x4v_t wfr[ND][KSTEPS];
#pragma unroll
  for (int jd = 0; jd < ND; ++jd) {
    const int nt = (wave + NW * jd) % NT;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const char* src = reinterpret_cast<const char*>(p.Wp) + (((int64_t)nt * KSTEPS + ks) * 64 + lane) * 16;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wfr[jd][ks]) : "v"(src) : "memory");
    }
  }
  {  // S / b' (2 x NT x 32 floats) by LDS-DMA as well: piece wave % NSBP (duplicates write the same bytes)
    constexpr int NSBP = (2 * NT * 32 * 4 + 1023) / 1024;
    const int pc = wave % NSBP;
    dma(rsb, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::SB_OFF + pc * 1024)), pc * 1024 + lane * 16);
  }
  // ---- a round = NBUF tiles, all in flight at once.  Every wave issues exactly CNT pieces per tile (a wave whose
  // share is one short repeats its last piece; tiles past the end are all-zero pieces), so "tile b has landed" is
  // the COUNTED wait vmcnt(CNT * (NBUF - 1 - b)): the younger tiles stay in flight while tile b is multiplied and
  // stored (reads and writes of a launch overlap instead of forming two bursts).
  constexpr int CNT = CF::CNT;
  auto issue_round = [&](int tile0) {
#pragma unroll
    for (int b = 0; b < NBUF; ++b) {
      const int tile = tile0 + b * grid;
#pragma unroll
      for (int i = 0; i < (CF::TP + NW - 1) / NW; ++i) {
        int q = wave + NW * i;
        q = q < CF::TP ? q : q - NW;
        const int sidx = q * 64 + lane;
        const int tok = sidx / CF::XSLOTS, sl = sidx - tok * CF::XSLOTS;
        const int grow = tile * L3_TT + tok;
        const bool ok = tile < p.ntiles && tok < L3_TT && sl < CF::XD && grow < p.M;
        dma(rsx, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(b * CF::TILEB + q * 1024)),
            ok ? grow * ((int)p.ldx * 2) + sl * 16 : p.x_bytes);   // (extents < 2^31 bytes)
      }
      if constexpr (RES) {   // the residual rows of the tile: 16-B chunks of the N output channels
#pragma unroll
        for (int i = 0; i < (CF::RP + NW - 1) / NW; ++i) {
          int q = wave + NW * i;
          q = q < CF::RP ? q : q - NW;
          const int sidx = q * 64 + lane;
          const int tok = sidx / CF::RSLOTS, sl = sidx - tok * CF::RSLOTS;
          const int grow = tile * L3_TT + tok;
          const bool ok = tile < p.ntiles && tok < L3_TT && sl * 8 < p.N && grow < p.M;
          dma(rsr, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(CF::R_OFF + b * CF::RTILEB + q * 1024)),
              ok ? grow * ((int)p.ldr * 2) + sl * 16 : p.r_bytes);
        }
      }
    }
  };

  for (int tile0 = blockIdx.x; tile0 < p.ntiles; tile0 += grid * NBUF) {
    issue_round(tile0);
#pragma unroll 1
    for (int b = 0; b < NBUF; ++b) {
      const int tile = tile0 + b * grid;
      if (tile >= p.ntiles) break;
      switch (NBUF - 1 - b) {   // the wait's count is an instruction immediate
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT * 3) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT * 2) : "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT * 1) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
#pragma unroll
      for (int jd = 0; jd < ND; ++jd)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(wfr[jd][ks]));   // every use of a fragment is behind a wait
      __syncthreads();
      const char* tb = smem + b * CF::TILEB;
      const char* rb = smem + CF::R_OFF + b * CF::RTILEB;
      float* st = statL + b * L3_TT * 2;
      if constexpr (LN) {
        // (mean, rstd) of every token of the tile: 4 lanes per token, two passes over the row's 16-B slots
        for (int tok = tid >> 2; tok < L3_TT; tok += NTHR / 4) {
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
              const bool valid = sl < CF::XD && sl * 8 + e < K;
              xv[i][e] = valid ? xv[i][e] : 0.f;
              sum += xv[i][e];
            }
          }
          sum += __shfl_xor(sum, 1, 64);
          sum += __shfl_xor(sum, 2, 64);
          const float mean = sum * (1.0f / K);
          float sq = 0.f;
#pragma unroll
          for (int i = 0; i < NSL; ++i) {
            const int sl = part + 4 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const bool valid = sl < CF::XD && sl * 8 + e < K;
              const float d = xv[i][e] - mean;
              sq = valid ? fmaf(d, d, sq) : sq;
            }
          }
          sq += __shfl_xor(sq, 1, 64);
          sq += __shfl_xor(sq, 2, 64);
          const float rstd = rsqrtf(sq * (1.0f / K) + 1e-5f);
          if (part == 0) {
            st[tok * 2] = mean;
            st[tok * 2 + 1] = rstd;
            const int grow = tile * L3_TT + tok;
            if (grow < p.M) *reinterpret_cast<float2*>(p.stats + (int64_t)grow * 2) = make_float2(mean, rstd);
          }
        }
        __syncthreads();
      }
      // items j = jd, jd + ND, ... share the output tile (and the registers) of fragment set jd: one unrolled body per
      // set, a rolled loop over its token sub-tiles
#pragma unroll
      for (int jd = 0; jd < ND; ++jd)
#pragma unroll 1
      for (int j = jd; j < NJ; j += ND) {
        const int item = wave + NW * j;
        if (item >= CF::ITEMS) break;                         // wave-uniform
        const int nt = item % NT, tt = item / NT;
        const int tok = tt * 32 + r;
        const int grow = tile * L3_TT + tok;
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const char* brow = tb + tok * XS + h * 16;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          Pack16 bq = *reinterpret_cast<const Pack16*>(brow + ks * 32);
          if (K % 16 != 0 && ks == KSTEPS - 1) {
            // the row's last 16-B slot ends with the NEXT channels of the same memory row (a dense-buffer slice: possibly
            // not written yet, any bit pattern): zero them, a zero weight does not stop a NaN
            constexpr int c0 = K % 16 < 8 ? K % 16 : 8, c1 = K % 16 > 8 ? K % 16 - 8 : 0;   // valid elements of lane half 0 / 1
#pragma unroll
            for (int d = 0; d < 4; ++d) bq.w[d] = (2 * d < (h ? c1 : c0)) ? bq.w[d] : 0u;
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wfr[jd][ks]), __builtin_bit_cast(bf16x8_t, bq), acc, 0, 0, 0);
        }
        // epilogue: register group g4 holds outputs n = 32 nt + 8 g4 + 4 h + (0..3)
        float mean = 0.f, rstd = 1.f;
        if constexpr (LN) {
          const float2 mr = *reinterpret_cast<const float2*>(st + tok * 2);
          mean = mr.x; rstd = mr.y;
        }
        const float nrm = -rstd * mean;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int n0 = nt * 32 + 8 * g4 + 4 * h;
          const float4 S4 = *reinterpret_cast<const float4*>(sbL + n0);
          const float4 B4 = *reinterpret_cast<const float4*>(sbL + NT * 32 + n0);
          if constexpr (LN) {
            acc[4 * g4] = fmaf(rstd, acc[4 * g4], fmaf(nrm, S4.x, B4.x));
            acc[4 * g4 + 1] = fmaf(rstd, acc[4 * g4 + 1], fmaf(nrm, S4.y, B4.y));
            acc[4 * g4 + 2] = fmaf(rstd, acc[4 * g4 + 2], fmaf(nrm, S4.z, B4.z));
            acc[4 * g4 + 3] = fmaf(rstd, acc[4 * g4 + 3], fmaf(nrm, S4.w, B4.w));
          } else {
            acc[4 * g4] += B4.x; acc[4 * g4 + 1] += B4.y; acc[4 * g4 + 2] += B4.z; acc[4 * g4 + 3] += B4.w;
          }
          if constexpr (RES) {   // + the residual's 4 channels of this group: 8 B of the token's row in the R tile
            const u32x2_a4 rr = *reinterpret_cast<const u32x2_a4*>(rb + tok * CF::RS + n0 * 2);
            acc[4 * g4] += bf16lo(rr.x); acc[4 * g4 + 1] += bf16hi(rr.x);
            acc[4 * g4 + 2] += bf16lo(rr.y); acc[4 * g4 + 3] += bf16hi(rr.y);
          }
        }
        if (grow < p.M) {
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
            const int cb = nt * 32 + 8 * (2 * gp + h);          // the lane's 8 consecutive outputs
            const int nv = p.N - cb;                            // valid outputs from cb on (N is even)
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
    }
    __syncthreads();   // the round's buffers may be overwritten
  }
}

template <int K, int NT, bool LN, bool RES>
int launch_l3(L3Args& p, hipStream_t st, const char* what) {
  using CF = L3Cfg<K, NT, LN, RES>;
  p.ntiles = (p.M + L3_TT - 1) / L3_TT;
  int grid = (p.ntiles + CF::NBUF - 1) / CF::NBUF;
  if (grid > 256 * CF::WGCU) grid = 256 * CF::WGCU;
  auto kern = lin3_kernel<K, NT, LN, RES>;
  // (per launch: the attribute is per DEVICE, a process-wide "done" flag would leave a second GPU without it)
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CF::NTHR), CF::SMEM, st, p);
  return rdst_launch_status(what);
}

}  // namespace

// image of one Linear layer at `out`: [fragments (N/32 tiles x K/16 steps) x 1 KB][S (NP floats)][b' (NP floats)]
int lin3_pack_launch(const float* W, const float* gamma, const float* beta, const float* bias, void* out, int N, int K, float s,
                     hipStream_t st) {
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  bf16* wp = reinterpret_cast<bf16*>(out);
  float* sb = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + (size_t)nt * ks * 1024);
  const int nfr = nt * ks * 64, nb1 = (nfr + 255) / 256, nb2 = (nt * 32 + 3) / 4;
  hipLaunchKernelGGL(lin3_pack_kernel, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, st, W, gamma, beta, bias, wp, sb, N, K, ks, nt, s);
  return rdst_launch_status("lin3_pack");
}

size_t lin3_pack_bytes(int K, int N) {
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  return (size_t)nt * ks * 1024 + (size_t)2 * nt * 32 * 4 + 256;
}

// RDST_ENOTSUP = not one of the covered shapes (the caller falls back to linear_mfma.hip)
int lin3_fwd_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* ln_b, int in_act, const float* Wt, const float* bias,
                  const bf16* R, int64_t ldr, bf16* Y, int64_t ldy, float* stats, int64_t M, int K, int N, float s, void* wpack,
                  bool prepacked, hipStream_t st) {
  if (!wpack || ((uintptr_t)wpack & 15) || in_act || !Wt || M <= 0) return RDST_ENOTSUP;
  const bool ln = ln_w != nullptr;
  if (!(K == 60 || K == 90 || K == 120)) return RDST_ENOTSUP;
  int kind = 0;   // 1 qkv (LN, N = 3K), 2 proj (residual, N = K), 3 dense tail (LN, N = 30)
  if (ln && !R && N == 3 * K) kind = 1;
  else if (!ln && R && N == K) kind = 2;
  else if (ln && !R && N == 30) kind = 3;
  if (!kind) return RDST_ENOTSUP;
  if (((uintptr_t)X & 3) || (ldx & 1) || ((uintptr_t)Y & 3) || (ldy & 1) || (R && (((uintptr_t)R & 3) || (ldr & 1)))) return RDST_ENOTSUP;
  const int64_t xb = ((M - 1) * ldx + K) * 2;
  if (xb >= (1ll << 31) || M * ldy * 2 >= (1ll << 31) || (R && M * ldr * 2 >= (1ll << 31))) return RDST_ENOTSUP;
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  bf16* wp = reinterpret_cast<bf16*>(wpack);
  float* sb = reinterpret_cast<float*>(reinterpret_cast<char*>(wpack) + (size_t)nt * ks * 1024);
  if (!prepacked) {
    const int nfr = nt * ks * 64, nb1 = (nfr + 255) / 256, nb2 = (nt * 32 + 3) / 4;
    hipLaunchKernelGGL(lin3_pack_kernel, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, st, Wt, ln_w, ln_b, bias, wp, sb, N, K, ks, nt, s);
    if (int rc = rdst_launch_status("lin3_pack")) return rc;
  }
  L3Args p{};
  p.X = X; p.ldx = ldx; p.x_bytes = (int)xb; p.Wp = wp; p.sb = sb; p.R = R; p.ldr = ldr;
  p.r_bytes = R ? (int)(((M - 1) * ldr + N) * 2) : 0; p.Y = Y; p.ldy = ldy; p.stats = stats;
  p.M = (int)M; p.N = N;
#define L3_CASE(KK)                                                                                   \
  if (K == KK) {                                                                                      \
    if (kind == 1) return launch_l3<KK, (3 * KK + 31) / 32, true, false>(p, st, "lin3_qkv");           \
    if (kind == 2) return launch_l3<KK, (KK + 31) / 32, false, true>(p, st, "lin3_proj");              \
    return launch_l3<KK, 1, true, false>(p, st, "lin3_tail");                                          \
  }
  L3_CASE(60) L3_CASE(90) L3_CASE(120)
#undef L3_CASE
  return RDST_ENOTSUP;
}
