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
  const bf16* R; int64_t ldr;
  bf16* Y; int64_t ldy;
  float* stats;
  int M, N, ntiles;
};

template <int K, int NT, bool LN, bool RES>
struct L3Cfg {
  static constexpr int KSTEPS = (K + 15) / 16;
  static constexpr int XS = l3_stride(K), XSLOTS = XS / 16, XD = (2 * K + 15) / 16;
  static constexpr int TP = (L3_TT * XSLOTS + 63) / 64, TILEB = TP * 1024;
  static constexpr int NBUF = 4;
  static constexpr int ITEMS = NT * 4, NJ = (ITEMS + 7) / 8;
  static constexpr int PERIOD = NT / l3_gcd(8, NT), ND = NJ < PERIOD ? NJ : PERIOD;
  static constexpr int STAT_OFF = NBUF * TILEB;                       // [NBUF][128][2] floats
  static constexpr int SB_OFF = STAT_OFF + NBUF * L3_TT * 2 * 4;      // [2][NT*32] floats
  static constexpr int SMEM = SB_OFF + 2 * NT * 32 * 4;
  static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int K, int NT, bool LN, bool RES>
__global__ void __launch_bounds__(512, 2) lin3_kernel(const L3Args p) {
  using CF = L3Cfg<K, NT, LN, RES>;
  constexpr int KSTEPS = CF::KSTEPS, XS = CF::XS, NBUF = CF::NBUF, ND = CF::ND, NJ = CF::NJ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* statL = reinterpret_cast<float*>(smem + CF::STAT_OFF);
  float* sbL = reinterpret_cast<float*>(smem + CF::SB_OFF);

  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  u32x4s_t rsrc;
  rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p.X);
  rsrc.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p.X >> 32) & 0xffffu);
  rsrc.z = __builtin_amdgcn_readfirstlane((uint32_t)p.x_bytes);
  rsrc.w = 0x00020000u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](uint32_t ldst, int off) {   // inline asm: see conv3_mfma.hip
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rsrc) : "memory");
  };
  const int grid = gridDim.x;
  // ---- the whole first round of tiles goes in flight before anything else ---------------------------------------
  auto issue_round = [&](int tile0) {
#pragma unroll
    for (int b = 0; b < NBUF; ++b) {
      const int tile = tile0 + b * grid;
      if (tile < p.ntiles) {
        for (int q = wave; q < CF::TP; q += 8) {
          const int sidx = q * 64 + lane;
          const int tok = sidx / CF::XSLOTS, sl = sidx - tok * CF::XSLOTS;
          const int grow = tile * L3_TT + tok;
          const bool ok = tok < L3_TT && sl < CF::XD && grow < p.M;
          const int off = ok ? grow * ((int)p.ldx * 2) + sl * 16 : p.x_bytes;   // (extent < 2^31 bytes)
          dma(__builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(b * CF::TILEB + q * 1024)), off);
        }
      }
    }
  };
  issue_round(blockIdx.x);

  // ---- this wave's weight fragments: item j of the wave is (nt, tt) = ((wave + 8 j) % NT, (wave + 8 j) / NT) ------
  Pack16 wf[ND][KSTEPS];
#pragma unroll
  for (int jd = 0; jd < ND; ++jd) {
    const int nt = (wave + 8 * jd) % NT;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(reinterpret_cast<const char*>(p.Wp) +
                                                             (((int64_t)nt * KSTEPS + ks) * 64 + lane) * 16);
      wf[jd][ks].w[0] = v.x; wf[jd][ks].w[1] = v.y; wf[jd][ks].w[2] = v.z; wf[jd][ks].w[3] = v.w;
    }
  }
  for (int i = tid; i < 2 * NT * 32; i += 512) sbL[i] = p.sb[i];

  for (int tile0 = blockIdx.x; tile0 < p.ntiles; tile0 += grid * NBUF) {
    if (tile0 != (int)blockIdx.x) issue_round(tile0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int b = 0; b < NBUF; ++b) {
      const int tile = tile0 + b * grid;
      if (tile >= p.ntiles) break;
      const char* tb = smem + b * CF::TILEB;
      float* st = statL + b * L3_TT * 2;
      if constexpr (LN) {
        // (mean, rstd) of every token of the tile: 4 lanes per token, two passes over the row's 16-B slots
        const int tok = tid >> 2, part = tid & 3;
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
        __syncthreads();
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int item = wave + 8 * j;
        if (item >= CF::ITEMS) continue;                      // wave-uniform
        const int nt = item % NT, tt = item / NT;
        const int tok = tt * 32 + r;
        const int grow = tile * L3_TT + tok;
        uint32_t rpre[2][4];
        if constexpr (RES) {
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const int cb = nt * 32 + 8 * (2 * gp + h), nv = p.N - cb;
            const bf16* rp = p.R + ((grow < p.M ? grow : 0) * (int)p.ldr + (nv > 0 ? cb : 0));
            if (nv >= 8) {
              const u32x4_a4 q4 = *reinterpret_cast<const u32x4_a4*>(rp);
              rpre[gp][0] = q4.x; rpre[gp][1] = q4.y; rpre[gp][2] = q4.z; rpre[gp][3] = q4.w;
            } else {
#pragma unroll
              for (int d = 0; d < 4; ++d) rpre[gp][d] = (2 * d + 2 <= nv) ? *reinterpret_cast<const uint32_t*>(rp + 2 * d) : 0u;
            }
          }
        }
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
          Mma<bf16>::mma(acc, wf[j % ND][ks], bq);
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
            if constexpr (RES) {
#pragma unroll
              for (int d = 0; d < 4; ++d) { c8[2 * d] += bf16lo(rpre[gp][d]); c8[2 * d + 1] += bf16hi(rpre[gp][d]); }
            }
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
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
  auto kern = lin3_kernel<K, NT, LN, RES>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), CF::SMEM, st, p);
  return rdst_launch_status(what);
}

}  // namespace

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
  p.X = X; p.ldx = ldx; p.x_bytes = (int)xb; p.Wp = wp; p.sb = sb; p.R = R; p.ldr = ldr; p.Y = Y; p.ldy = ldy; p.stats = stats;
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
