// C-ABI entry points of the window-attention path (K1/K2) + the deterministic d(table) reduction.
#include "common.h"
#include "wattn.h"
#include "reduce_batch.h"

thread_local char g_rdst_err[256] = {0};
thread_local int g_rdst_split = 0;

extern "C" int rdst_abi_version(void) { return 11; }
extern "C" const char* rdst_last_error(void) { return g_rdst_err; }

namespace {

int make_geom(WinGeom& g, int B, int H, int W, int C, int heads, int ws, int shift, const float* mask, int mask_nw,
              const char* who) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || ws <= 0)
    return rdst_fail(RDST_EINVAL, "%s: non-positive dimension", who);
  if (C % heads) return rdst_fail(RDST_EINVAL, "%s: C=%d not divisible by heads=%d", who, C, heads);
  if (H % ws || W % ws) return rdst_fail(RDST_EINVAL, "%s: H=%d, W=%d must be multiples of the window size %d", who, H, W, ws);
  if (shift < 0 || shift >= ws) return rdst_fail(RDST_EINVAL, "%s: shift=%d must be in [0, ws=%d)", who, shift, ws);
  if ((int64_t)B * H * W >= (1ll << 31)) return rdst_fail(RDST_EINVAL, "%s: more than 2^31 tokens", who);
  g.B = B; g.H = H; g.W = W; g.C = C; g.heads = heads; g.ws = ws; g.shift = shift;
  g.nWh = H / ws; g.nWw = W / ws; g.N = ws * ws; g.T = (2 * ws - 1) * (2 * ws - 1);
  if (mask && mask_nw <= 0) return rdst_fail(RDST_EINVAL, "%s: mask given with mask_nw=%d", who, mask_nw);
  g.mask = mask; g.mask_nw = mask ? mask_nw : 1;
  g.pdrop = 0.f; g.seed = nullptr;
  return 0;
}

}  // namespace

extern "C" int rdst_wattn_fwd(const void* qkv, int64_t ld_qkv, const float* table, const float* mask, int mask_nw,
                              void* out, int64_t ld_out, int B, int H, int W, int C, int heads, int ws, int shift,
                              float scale, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  WinGeom g;
  if (int rc = make_geom(g, B, H, W, C, heads, ws, shift, mask, mask_nw, "rdst_wattn_fwd")) return rc;
  if (!qkv || !table || !out) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd: null pointer");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd: bad dtype %d", dtype);
  if (ld_qkv < 3 * C || ld_out < C) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd: leading dimension too small");
  if (int rc = wattn_fwd_mfma(qkv, ld_qkv, table, out, ld_out, g, scale, dtype, (hipStream_t)stream); rc != RDST_ENOTSUP)
    return rc;
  return wattn_fwd_generic(qkv, ld_qkv, table, out, ld_out, g, scale, dtype, (hipStream_t)stream);
}

extern "C" size_t rdst_wattn_bwd_workspace(int B, int H, int W, int C, int heads, int ws) {
  if (B <= 0 || H <= 0 || W <= 0 || ws <= 0 || heads <= 0) return 0;
  const size_t nwin = (size_t)B * (H / ws) * (W / ws);
  const size_t T = (size_t)(2 * ws - 1) * (2 * ws - 1);
  (void)C;
  return nwin * heads * T * sizeof(float);
}

extern "C" int rdst_wattn_bwd(const void* qkv, int64_t ld_qkv, const float* table, const float* mask, int mask_nw,
                              const void* dout, int64_t ld_dout, void* dqkv, int64_t ld_dqkv, float* dtable,
                              void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws,
                              int shift, float scale, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  WinGeom g;
  if (int rc = make_geom(g, B, H, W, C, heads, ws, shift, mask, mask_nw, "rdst_wattn_bwd")) return rc;
  if (!qkv || !table || !dout || !dqkv || !dtable || !workspace)
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd: null pointer");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd: bad dtype %d", dtype);
  if (ld_qkv < 3 * C || ld_dqkv < 3 * C || ld_dout < C)
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd: leading dimension too small");
  if (workspace_bytes < rdst_wattn_bwd_workspace(B, H, W, C, heads, ws))
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)workspace;
  int nwin = B * g.nWh * g.nWw;
  int nslab = 0;
  int rc = wattn_bwd_mfma(qkv, ld_qkv, table, dout, ld_dout, dqkv, ld_dqkv, slab, nwin, g, scale, dtype, &nslab, st);
  if (rc == 0) {
    nwin = nslab;  // one slab row per persistent workgroup
  } else {
    if (rc != RDST_ENOTSUP) return rc;
    if (int rc2 = wattn_bwd_generic(qkv, ld_qkv, table, dout, ld_dout, dqkv, ld_dqkv, slab, g, scale, dtype, st)) return rc2;
  }
  rbatch::SumJob sj{};   // dtable[t*heads + h] = sum over the slab rows [heads][T], fixed order
  sj.slab = slab; sj.nwg = nwin; sj.stride = (int64_t)heads * g.T; sj.tot = heads * g.T; sj.map = rbatch::MAP_DTABLE;
  sj.out = dtable; sj.a = heads; sj.b = g.T;
  return rbatch::sum(sj, st);
}

// ---- K8: the attention half of a Swin block in one launch (swinattn_fwd.hip) ---------------------------------------------------
extern "C" int rdst_swin_attn_fwd_supported(int C, int heads, int ws, int dtype) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  return (dtype == RDST_BF16 && swinattn_supported(C, heads, ws)) ? 1 : 0;
}
extern "C" size_t rdst_swin_attn_fwd_workspace(int C) { return C > 0 ? swinattn_pack_bytes(C) : 0; }
extern "C" int rdst_swin_attn_fwd(const void* X, int64_t ld_x, const float* ln_w, const float* ln_b, const float* Wqkv,
                                  const float* bqkv, const float* table, const float* Wproj, const float* bproj, void* qkv,
                                  int64_t ld_qkv, void* a, int64_t ld_a, void* x1, int64_t ld_x1, float* stats, void* workspace,
                                  size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws, int shift, float scale,
                                  int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  WinGeom g;
  if (int rc = make_geom(g, B, H, W, C, heads, ws, shift, nullptr, 0, "rdst_swin_attn_fwd")) return rc;
  if (!X || !ln_w || !ln_b || !Wqkv || !table || !Wproj || !qkv || !a || !x1 || !stats || !workspace)
    return rdst_fail(RDST_EINVAL, "rdst_swin_attn_fwd: null pointer");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_swin_attn_fwd: bad dtype %d", dtype);
  if (ld_x < C || ld_qkv < 3 * C || ld_a < C || ld_x1 < C) return rdst_fail(RDST_EINVAL, "rdst_swin_attn_fwd: leading dimension too small");
  if (dtype != RDST_BF16) return RDST_ENOTSUP;
  const bool prepacked = workspace_bytes == RDST_PREPACKED;
  if (!prepacked && workspace_bytes < swinattn_pack_bytes(C)) return rdst_fail(RDST_EINVAL, "rdst_swin_attn_fwd: workspace too small");
  return swinattn_fwd_bf16((const bf16*)X, ld_x, ln_w, ln_b, Wqkv, bqkv, table, Wproj, bproj, (bf16*)qkv, ld_qkv, (bf16*)a, ld_a,
                           (bf16*)x1, ld_x1, stats, workspace, prepacked, g, scale, (hipStream_t)stream);
}

// ---- window attention with the forward's row statistics kept for the backward (window 16, bf16: wattn16_mfma.hip) ---------------
// nlse[token][head] = -(scale log2e max_j S'_ij + log2 sum_j exp(...)): with it and the forward's output the backward's first pass
// streams its key tiles (no row maximum / sum / normalisation / delta pass).  RDST_ENOTSUP for every shape the plain entry points
// serve with other kernels: the caller then uses rdst_wattn_fwd / rdst_wattn_bwd.
extern "C" int rdst_wattn_fwd_lse(const void* qkv, int64_t ld_qkv, const float* table, void* out, int64_t ld_out, float* nlse,
                                  int B, int H, int W, int C, int heads, int ws, int shift, float scale, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  WinGeom g;
  if (int rc = make_geom(g, B, H, W, C, heads, ws, shift, nullptr, 0, "rdst_wattn_fwd_lse")) return rc;
  if (!qkv || !table || !out || !nlse) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd_lse: null pointer");
  if (ld_qkv < 3 * C || ld_out < C) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd_lse: leading dimension too small");
  if (dtype != RDST_BF16) return RDST_ENOTSUP;
  return wattn16_fwd_mfma(qkv, ld_qkv, table, out, ld_out, g, scale, (hipStream_t)stream, nlse);
}

extern "C" int rdst_wattn_bwd_lse(const void* qkv, int64_t ld_qkv, const float* table, const void* dout, int64_t ld_dout,
                                  const void* out, int64_t ld_out, const float* nlse, void* dqkv, int64_t ld_dqkv, float* dtable,
                                  void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws,
                                  int shift, float scale, int dtype, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  WinGeom g;
  if (int rc = make_geom(g, B, H, W, C, heads, ws, shift, nullptr, 0, "rdst_wattn_bwd_lse")) return rc;
  if (!qkv || !table || !dout || !out || !nlse || !dqkv || !dtable || !workspace)
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd_lse: null pointer");
  if (ld_qkv < 3 * C || ld_dqkv < 3 * C || ld_dout < C || ld_out < C)
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd_lse: leading dimension too small");
  if (workspace_bytes < rdst_wattn_bwd_workspace(B, H, W, C, heads, ws))
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd_lse: workspace too small");
  if (dtype != RDST_BF16) return RDST_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)workspace;
  int nslab = 0;
  const int nwin = B * g.nWh * g.nWw;
  if (int rc = wattn16_bwd_mfma(qkv, ld_qkv, table, dout, ld_dout, dqkv, ld_dqkv, slab, nwin, g, scale, &nslab, st, out, ld_out, nlse))
    return rc;
  rbatch::SumJob sj{};
  sj.slab = slab; sj.nwg = nslab; sj.stride = (int64_t)heads * g.T; sj.tot = heads * g.T; sj.map = rbatch::MAP_DTABLE;
  sj.out = dtable; sj.a = heads; sj.b = g.T;
  return rbatch::sum(sj, st);
}

// ---- attention dropout (WindowAttention.attn_drop > 0 in training, swin_transformer_sr.py:102,136): the generic kernels with
// a counter-based mask; `seed` is a DEVICE pointer to the call's 64-bit seed (drawn by the host from its generator: a HIP graph
// replay then sees a new seed without re-capturing); the backward must get the same attn_drop and seed as its forward.
namespace {
int drop_args(WinGeom& g, float attn_drop, const unsigned long long* seed, const char* who) {
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return rdst_fail(RDST_EINVAL, "%s: attn_drop=%g must be in [0, 1)", who, (double)attn_drop);
  if (attn_drop > 0.f && !seed) return rdst_fail(RDST_EINVAL, "%s: attn_drop > 0 needs a seed", who);
  g.pdrop = attn_drop;
  g.seed = attn_drop > 0.f ? seed : nullptr;
  return 0;
}
}  // namespace

extern "C" int rdst_wattn_fwd_drop(const void* qkv, int64_t ld_qkv, const float* table, const float* mask, int mask_nw,
                                   void* out, int64_t ld_out, int B, int H, int W, int C, int heads, int ws, int shift,
                                   float scale, int dtype, float attn_drop, const unsigned long long* seed, void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  WinGeom g;
  if (int rc = make_geom(g, B, H, W, C, heads, ws, shift, mask, mask_nw, "rdst_wattn_fwd_drop")) return rc;
  if (!qkv || !table || !out) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd_drop: null pointer");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd_drop: bad dtype %d", dtype);
  if (ld_qkv < 3 * C || ld_out < C) return rdst_fail(RDST_EINVAL, "rdst_wattn_fwd_drop: leading dimension too small");
  if (int rc = drop_args(g, attn_drop, seed, "rdst_wattn_fwd_drop")) return rc;
  return wattn_fwd_generic(qkv, ld_qkv, table, out, ld_out, g, scale, dtype, (hipStream_t)stream);
}

extern "C" int rdst_wattn_bwd_drop(const void* qkv, int64_t ld_qkv, const float* table, const float* mask, int mask_nw,
                                   const void* dout, int64_t ld_dout, void* dqkv, int64_t ld_dqkv, float* dtable,
                                   void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws,
                                   int shift, float scale, int dtype, float attn_drop, const unsigned long long* seed,
                                   void* stream) {
  SplitScope split_scope(dtype);   // RDST_F32X3: fp32 rows, split-bf16 GEMMs where a kernel has the form (common.h)
  WinGeom g;
  if (int rc = make_geom(g, B, H, W, C, heads, ws, shift, mask, mask_nw, "rdst_wattn_bwd_drop")) return rc;
  if (!qkv || !table || !dout || !dqkv || !dtable || !workspace)
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd_drop: null pointer");
  if (dtype != RDST_F32 && dtype != RDST_BF16) return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd_drop: bad dtype %d", dtype);
  if (ld_qkv < 3 * C || ld_dqkv < 3 * C || ld_dout < C)
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd_drop: leading dimension too small");
  if (workspace_bytes < rdst_wattn_bwd_workspace(B, H, W, C, heads, ws))
    return rdst_fail(RDST_EINVAL, "rdst_wattn_bwd_drop: workspace too small");
  if (int rc = drop_args(g, attn_drop, seed, "rdst_wattn_bwd_drop")) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)workspace;
  const int nwin = B * g.nWh * g.nWw;
  if (int rc = wattn_bwd_generic(qkv, ld_qkv, table, dout, ld_dout, dqkv, ld_dqkv, slab, g, scale, dtype, st)) return rc;
  rbatch::SumJob sj{};   // one slab row per (window, head) workgroup: [nwin][heads][T]
  sj.slab = slab; sj.nwg = nwin; sj.stride = (int64_t)heads * g.T; sj.tot = heads * g.T; sj.map = rbatch::MAP_DTABLE;
  sj.out = dtable; sj.a = heads; sj.b = g.T;
  return rbatch::sum(sj, st);
}

// The multipliers (0 or 1 / (1 - attn_drop)) the two entry points above apply: out[(window * heads + head)][N][N] floats
// (inspection and tests: the kernels never store the mask).
extern "C" int rdst_wattn_drop_mask(float* out, int B, int H, int W, int heads, int ws, float attn_drop,
                                    const unsigned long long* seed, void* stream) {
  if (!out || !seed || B <= 0 || H <= 0 || W <= 0 || heads <= 0 || ws <= 0 || H % ws || W % ws)
    return rdst_fail(RDST_EINVAL, "rdst_wattn_drop_mask: bad argument");
  if (!(attn_drop > 0.f && attn_drop < 1.f)) return rdst_fail(RDST_EINVAL, "rdst_wattn_drop_mask: attn_drop must be in (0, 1)");
  return wattn_drop_mask(out, (int64_t)B * (H / ws) * (W / ws) * heads, ws * ws, attn_drop, seed, (hipStream_t)stream);
}
