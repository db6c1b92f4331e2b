/* rdst_hip.h — C ABI of librdst_hip.so, the MI355X (gfx950) implementation of the RDST hot path.
 *
 * The reference (GinZhu/RDST) is pure Python/PyTorch and has no FFI of its own: its hot path is
 * sequences of stock torch ops inside nn.Module.forward.  Each entry point below
 * replaces one such sequence; the comment on it cites the reference lines (relative to the
 * reference root).  The Python host side (rdst_amd/ops.py) binds these with ctypes; see
 * INTEGRATION.md for the binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to caller-allocated memory; nothing is allocated, freed or
 *     synchronised inside; work is enqueued on `stream` (a hipStream_t passed as void*), so the
 *     calls are graph-capturable and re-entrant across streams;
 *   - activations are token-major rows: element (row, c) lives at base[row*ld + c], `ld` in
 *     ELEMENTS (this is how the dense concat buffer of an RDSTB is addressed in place);
 *   - `dtype` selects the activation element type: RDST_F32 (parity mode, fp32 I/O and math),
 *     RDST_BF16 (throughput mode: bf16 I/O, fp32 accumulation) or RDST_F32X3 (fast parity mode: fp32
 *     I/O exactly as RDST_F32, but every GEMM-shaped product takes its operands as TWO bf16 terms —
 *     16 mantissa bits — on the bf16 matrix cores with fp32 accumulation; LayerNorm statistics, the
 *     softmax sums and every reduction stay fp32; entry points without GEMMs treat it as RDST_F32).
 *     What RDST_F32X3 changes BESIDES the operand split (so its error model is not "2^-17 per operand"
 *     alone): GELU and GELU' are evaluated as gelu_fast / gelu_grad_fast (erf by Abramowitz-Stegun
 *     7.1.26 on the hardware exp2 / rcp, |error| <= 1.5e-7 absolute) instead of erff; the softmax of
 *     rdst_wattn_* runs in the log2 domain on v_exp_f32 (logits pre-multiplied by log2 e) instead of
 *     expf.  What it does NOT cover and silently runs in exact fp32 (same results as RDST_F32):
 *     rdst_wattn_* with 16x16 windows or an explicit mask, rdst_mlp_fwd / rdst_mlp_bwd (a fused Mlp
 *     call in RDST_F32X3 is refused with RDST_ENOTSUP: the host runs fc1 / fc2 as two rdst_ln_linear_*
 *     calls), LayerNorm-only rows, the one-channel convolutions, rdst_adam.  Parameters and parameter
 *     gradients are ALWAYS fp32;
 *   - return value: 0 on success, a negative hipError_t from the launch, or RDST_EINVAL /
 *     RDST_ENOTSUP for bad or unsupported arguments.  Never throws.  rdst_last_error() returns a
 *     thread-local message for the last non-zero return.
 */
#ifndef RDST_HIP_H
#define RDST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDST_F32 0
#define RDST_BF16 1
#define RDST_F32X3 2   /* fp32 rows, split-bf16 products (~4e-6 .. 1e-5 relative per product).  The E1 shapes of rdst_ln_linear_* and
                          rdst_conv_* (ABI 11: lin3x / lnlin3x / conv3x kernels) and rdst_u_conv run three partial products
                          (hi.hi + hi.lo + lo.hi; the dropped lo.lo is <= 2^-18 relative); rdst_wattn_* with 8x8 windows and the other
                          shapes run all four in two v_mfma_f32_32x32x16_bf16; everything else = RDST_F32 (see the conventions above) */

#define RDST_EINVAL (-10001)
#define RDST_ENOTSUP (-10002)

#define RDST_ACT_NONE 0
#define RDST_ACT_GELU 1       /* exact erf GELU (nn.GELU default), swin_transformer_sr.py:14,19 */
#define RDST_ACT_LEAKY02 2    /* LeakyReLU(0.2), rdst_variations.py:425 ('3conv' variant) */
#define RDST_ACT_LEAKY001 3   /* LeakyReLU(0.01) = nn.LeakyReLU default, swin_transformer_sr.py:752 (SwinIR) */

int rdst_abi_version(void);
const char* rdst_last_error(void);

/* ---- K1: fused window attention forward -------------------------------------------------------
 * Replaces torch.roll (swin_transformer_sr.py:244-247) -> window_partition (:32-43, :250-251) ->
 * per-head split of qkv (:117-118) -> q*scale (:120) -> q@k^T (:121) -> + relative_position_bias
 * gather (:123-126) -> + shifted-window mask (:128-131; mask itself :211-232, computed
 * analytically here, the (nW,N,N) buffer is never read) -> softmax (:132/:134) -> attn@v (:138) ->
 * head merge (:138) -> window_reverse (:46-59, :260-261) -> torch.roll back (:264-267).
 *   qkv   : (B*H*W, 3C) rows, inner order [3][heads][C/heads]   (output of the qkv Linear :117)
 *   table : (2*ws-1)^2 x heads fp32  (relative_position_bias_table, :85-86)
 *   out   : (B*H*W, C) rows          (input of the proj Linear :139)
 *   mask  : NULL (the shifted-window mask is derived from `shift`), or an explicit additive fp32
 *           mask (mask_nw, N, N) applied to window (w % mask_nw) — the `mask` argument of the
 *           reference's standalone WindowAttention.forward(x, mask) (:110-131).
 * H and W must be multiples of ws; 0 <= shift < ws.
 */
int rdst_wattn_fwd(const void* qkv, int64_t ld_qkv, const float* table, const float* mask,
                   int mask_nw, void* out, int64_t ld_out, int B, int H, int W, int C, int heads,
                   int ws, int shift, float scale, int dtype, void* stream);

/* ---- K2: fused window attention backward (autograd of the sequence above) ----------------------
 * Recomputes the softmax from qkv (nothing but qkv is saved by the forward).
 *   dout   : (B*H*W, C) gradient of `out`
 *   dqkv   : (B*H*W, 3C) gradient of `qkv` (overwritten)
 *   dtable : (2*ws-1)^2 x heads fp32 gradient of `table` (overwritten; deterministic two-pass sum)
 *   workspace : >= rdst_wattn_bwd_workspace(...) bytes of scratch
 */
size_t rdst_wattn_bwd_workspace(int B, int H, int W, int C, int heads, int ws);
int rdst_wattn_bwd(const void* qkv, int64_t ld_qkv, const float* table, const float* mask,
                   int mask_nw, const void* dout, int64_t ld_dout, void* dqkv, int64_t ld_dqkv,
                   float* dtable, void* workspace,
                   size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws, int shift,
                   float scale, int dtype, void* stream);

/* ---- K1 / K2 with attention dropout (ABI 8) -----------------------------------------------------------
 * WindowAttention(attn_drop > 0) in training (reference: networks/swin_transformer_sr.py:102, :136 — nn.Dropout on the
 * softmax output): out = (softmax(S) * M / (1 - attn_drop)) @ v with M ~ Bernoulli(1 - attn_drop).  The mask is a pure
 * function of (*seed, window, head, query, key) — a counter-based generator — so nothing is stored between forward and
 * backward; `seed` is a DEVICE pointer (the host draws the value on the stream, e.g. from torch's generator, so a HIP
 * graph replay gets a fresh mask), and rdst_wattn_bwd_drop must be given the attn_drop and *seed of its forward.
 * attn_drop = 0 is rdst_wattn_fwd / _bwd on the shape-generic kernels.  rdst_wattn_drop_mask writes the multipliers
 * (0 or 1 / (1 - attn_drop)) as out[(window * heads + head)][N][N] floats, for inspection and tests. */
int rdst_wattn_fwd_drop(const void* qkv, int64_t ld_qkv, const float* table, const float* mask, int mask_nw,
                        void* out, int64_t ld_out, int B, int H, int W, int C, int heads, int ws, int shift,
                        float scale, int dtype, float attn_drop, const unsigned long long* seed, void* stream);
int rdst_wattn_bwd_drop(const void* qkv, int64_t ld_qkv, const float* table, const float* mask, int mask_nw,
                        const void* dout, int64_t ld_dout, void* dqkv, int64_t ld_dqkv, float* dtable,
                        void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws,
                        int shift, float scale, int dtype, float attn_drop, const unsigned long long* seed,
                        void* stream);
int rdst_wattn_drop_mask(float* out, int B, int H, int W, int heads, int ws, float attn_drop,
                         const unsigned long long* seed, void* stream);

/* ---- K3: (LayerNorm | activation ->) Linear (-> *scale + residual), forward and backward ---------
 * Y[M,N] = ( f(X)[M,K] @ Wt[N,K]^T + bias ) * out_scale + R
 * with f = LayerNorm (ln_w != NULL), or the activation `in_act` applied to X on the fly, or identity.
 * Putting the activation on the INPUT side means the GELU output of the reference's Mlp is never
 * materialised: fc1 writes its pre-activation, fc2 reads it through GELU.
 * Replaces: norm1 + qkv Linear (swin_transformer_sr.py:240, :117); proj + shortcut add (:139,
 * :271); norm2 + fc1 (:272, :24); GELU + fc2 + residual (:25-27, :272); DenseSTLayer tail LN +
 * Linear written straight into its slot of the dense buffer, times dense_scale
 * (rdst_variations.py:310-313, :339-340: the torch.cat disappears); the final norm (:1337) and
 * patch_embed.norm (swin_transformer_sr.py:517-518) with Wt == NULL (LayerNorm only, N == K).
 *   ln_w/ln_b : fp32 (K) or NULL for no LayerNorm (eps 1e-5, biased variance)
 *   Wt, bias  : fp32 (N,K) row-major as nn.Linear stores it / (N) or NULL
 *   R         : optional residual rows (may alias Y), ld_r elements
 *   stats     : fp32 (M,2) {mean, rstd} written by the forward when ln_w != NULL (kept for bwd)
 *   workspace : rdst_ln_linear_fwd_workspace(K, N) bytes of 16-byte aligned device scratch for the bf16 fragment image
 *               of the weights that the streaming kernels read (lin3_mfma.hip); NULL / 0 selects the kernels that
 *               stage the fp32 weights themselves.
 */
size_t rdst_ln_linear_fwd_workspace(int K, int N);
/* ... per compute mode (ABI 11): with dtype = RDST_F32X3 the image is hi / lo bf16 fragment pairs + b' (RDST_PACK_LINEAR_X3), twice
 * the bf16 one; every other dtype gives rdst_ln_linear_fwd_workspace(K, N). */
size_t rdst_ln_linear_fwd_workspace2(int K, int N, int dtype);
int rdst_ln_linear_fwd(const void* X, int64_t ld_x, const float* ln_w, const float* ln_b, int in_act,
                       const float* Wt, const float* bias, const void* R, int64_t ld_r, void* Y,
                       int64_t ld_y, float* stats, void* workspace, size_t workspace_bytes, int64_t M, int K,
                       int N, float out_scale, int dtype, void* stream);

/* Backward of rdst_ln_linear_fwd.  dY (M,N) -> dX (M,K) = dX_add + f'(...) where dX_add is an optional
 * (M,K) tensor added on the way out (NULL = none; it may alias dX for an in-place accumulate) - this is how
 * the gradient of a residual fan-out is summed without a separate add kernel; dW (N,K), dbias (N),
 * dln_w/dln_b (K) [overwritten; any of them may be NULL to skip].  The gradient of the forward's residual
 * operand R is dY itself and is the caller's business. */
size_t rdst_ln_linear_bwd_workspace(int64_t M, int K, int N);
int rdst_ln_linear_bwd(const void* X, int64_t ld_x, const float* ln_w, const float* ln_b,
                       const float* stats, int in_act, const float* Wt, const void* dY,
                       int64_t ld_dy, void* dX, int64_t ld_dx, const void* dX_add, int64_t ld_dx_add,
                       float* dW, float* dbias, float* dln_w, float* dln_b, void* workspace,
                       size_t workspace_bytes, int64_t M, int K, int N, float out_scale, int dtype,
                       void* stream);
/* The same with a SECOND addend of dX, dX = dX_add + dX_add2 + f'(...): the gradient slice a dense concatenation
 * hands back for its prefix (rdst_variations.py:339-340, `torch.cat((x, new), 2)`: x feeds both the DenseSTLayer's
 * body and the concatenation, so autograd would sum the body's dX and a strided slice of the concatenation's gradient
 * in a separate add kernel).  dX_add2 NULL = rdst_ln_linear_bwd.  Taken only by the one-pass kernels of the E1 shapes
 * (RDST_BF16: lnlin3_mfma.hip; RDST_F32X3: lnlin3x_mfma.hip; every gradient requested); RDST_ENOTSUP otherwise, with
 * nothing launched - the caller then adds the slice itself. */
int rdst_ln_linear_bwd2(const void* X, int64_t ld_x, const float* ln_w, const float* ln_b,
                        const float* stats, int in_act, const float* Wt, const void* dY,
                        int64_t ld_dy, void* dX, int64_t ld_dx, const void* dX_add, int64_t ld_dx_add,
                        float* dW, float* dbias, float* dln_w, float* dln_b, void* workspace,
                        size_t workspace_bytes, int64_t M, int K, int N, float out_scale, int dtype,
                        void* stream, const void* dX_add2, int64_t ld_dx_add2);

/* ---- K7: the Mlp half of a Swin block, fused (bf16 throughput mode) ---------------------------------
 * y = x + fc2(GELU(fc1(LayerNorm(x))))   (Mlp.forward swin_transformer_sr.py:23-29 behind norm2 and the
 * second residual add, :272).  rdst_mlp_fused_supported() says whether the fused kernels cover (C, hid,
 * dtype): bf16, C+1 <= 128, hid+1 <= 2*ceil32(C+1); everything else returns RDST_ENOTSUP and the caller
 * composes the same function from rdst_ln_linear_fwd / rdst_ln_linear_bwd (K3).
 * Backward: ONE pass over (x, dY): the hidden activations are recomputed on the matrix cores, never read.
 *   X, stats : the block input rows (M, C) and the forward's LayerNorm statistics (M, 2) {mean, rstd}
 *   W1 (hid, C), b1 (hid) or NULL, W2 (C, hid) : fc1 / fc2 as nn.Linear stores them (fp32)
 *   dY (M, C) -> dX (M, C) = dY + LayerNorm'(fc1'(GELU'(fc2'(dY))));  dW1, db1, dW2, db2, dln_w, dln_b overwritten.
 */
int rdst_mlp_fused_supported(int C, int hid, int dtype);
/* Forward: Y (M, C) = X + fc2(GELU(fc1(LayerNorm(X)))) in one pass; the hidden activations never reach HBM and
 * nothing but `stats` (M, 2) {mean, rstd} is kept for the backward.  b1 / b2 may be NULL.  Y may not alias X. */
/* workspace: rdst_mlp_fwd_workspace(C, hid) bytes (16-byte aligned) for the packed weight images [fc1: Linear image of
 * (W1, ln_w, ln_b, b1)][fc2: Linear image of (W2, b2)] that the streaming kernel (mlp3_mfma.hip) reads; NULL / 0 selects
 * the kernel that stages the fp32 weights itself; RDST_PREPACKED = the two images are already there (rdst_pack_batch
 * with two RDST_PACK_LINEAR jobs whose `out` are workspace and workspace + rdst_ln_linear_fwd_workspace(C, hid)). */
size_t rdst_mlp_fwd_workspace(int C, int hid);
int rdst_mlp_fwd_packable(int C, int hid, int dtype);
int rdst_mlp_fwd(const void* X, int64_t ld_x, const float* ln_w, const float* ln_b, const float* W1,
                 const float* b1, const float* W2, const float* b2, void* Y, int64_t ld_y, float* stats,
                 void* workspace, size_t workspace_bytes, int64_t M, int C, int hid, int dtype, void* stream);
size_t rdst_mlp_bwd_workspace(int64_t M, int C, int hid);
int rdst_mlp_bwd(const void* X, int64_t ld_x, const float* ln_w, const float* ln_b, const float* stats,
                 const float* W1, const float* b1, const float* W2, const void* dY, int64_t ld_dy, void* dX,
                 int64_t ld_dx, float* dW1, float* db1, float* dW2, float* db2, float* dln_w, float* dln_b,
                 void* workspace, size_t workspace_bytes, int64_t M, int C, int hid, int dtype, void* stream);

/* ---- K4/K5/K6: k x k convolution (k = 1 or 3, stride 1, zero padding k/2) on token-major rows ----
 * Y[b,y,x,:] = ( sum_{ky,kx} in_act(X)[b,y+ky-p,x+kx-p,:] @ Wc[:, :, ky, kx]^T + bias ) * out_scale + R
 * with an optional PixelShuffle(r) folded into the store: out channel c*r*r + i*r + j of pixel (y,x)
 * lands in channel c of pixel (y*r+i, x*r+j) of a (B, H*r, W*r, Cout/r^2) image.
 * Replaces: PatchUnEmbed transpose (swin_transformer_sr.py:552-555) -> nn.Conv2d(150,60,3,1,1)
 * (rdst_variations.py:420-421) -> PatchEmbed transpose (:515-516) -> .mul(residual_scale) +
 * shortcut (:444-445); the LeakyReLU(0.2) of the '3conv' variant (:425,:427) as `in_act` of the next
 * conv; conv_after_body (:1285,:1348-1350); head (:1229,:1344); UpSampler conv + nn.PixelShuffle
 * (common.py:125-136) and the last conv (rdst_variations.py:1303); MeanShift (common.py:151-167).
 *   Wc : fp32 (Cout, Cin, k, k) as nn.Conv2d stores it.   R/Y rows are in the OUTPUT geometry.
 */
/*   workspace : rdst_conv_fwd_workspace(Cin, Cout, ksize) bytes of device scratch (16-byte aligned) for the bf16
 *               fragment-major image of the weights that the register-stationary 3x3 kernels read (conv3_mfma.hip);
 *               NULL / 0 selects the kernels that stage the fp32 weights themselves. */
size_t rdst_conv_fwd_workspace(int Cin, int Cout, int ksize);
int rdst_conv_fwd(const void* X, int64_t ld_x, int in_act, const float* Wc, const float* bias,
                  const void* R, int64_t ld_r, void* Y, int64_t ld_y, void* workspace, size_t workspace_bytes,
                  int B, int H, int W, int Cin, int Cout, int ksize, float out_scale, int shuffle_r, int dtype,
                  void* stream);

/* Backward: dY (output geometry) -> dX (B*H*W, Cin) = dX_add + ... (dX_add optional, may alias dX),
 * dW (Cout,Cin,k,k), dbias (Cout) [overwritten; may be NULL]. */
size_t rdst_conv_bwd_workspace(int B, int H, int W, int Cin, int Cout, int ksize);
int rdst_conv_bwd(const void* X, int64_t ld_x, int in_act, const float* Wc, const void* dY,
                  int64_t ld_dy, void* dX, int64_t ld_dx, const void* dX_add, int64_t ld_dx_add, float* dW,
                  float* dbias, void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin, int Cout,
                  int ksize, float out_scale, int shuffle_r, int dtype, void* stream);

/* ---- K1 / K2 with the forward's row statistics kept (window 16, bf16; round 5) --------------------------------------
 * rdst_wattn_fwd_lse = rdst_wattn_fwd (no explicit mask) that also writes nlse (B*H*W, heads) fp32:
 *   nlse[i][h] = -(scale log2(e) max_j S'_ij + log2 sum_j 2^(scale log2(e) (S'_ij - max))),  S' = q.k + bias / scale (+ mask),
 * i.e. 2^(scale log2(e) S'_ij + nlse[i][h]) is the attention weight of (i, j).  rdst_wattn_bwd_lse = rdst_wattn_bwd given
 * those statistics and the forward's output `out` (M, C): delta_i = sum_c dout_ic out_ic replaces the row reduction
 * sum_j P_ij dP_ij, so the first pass of the window-16 backward streams its key tiles (5.5 instead of 8.5 vector
 * instructions per logit, half the matrix instructions).  Same gradients within the bf16 rounding of `out`.
 * Covered: bf16, ws = 16, heads = 6, C/heads in {10, 15, 20}; everything else returns RDST_ENOTSUP (use the plain entry
 * points).  workspace as for rdst_wattn_bwd. */
int rdst_wattn_fwd_lse(const void* qkv, int64_t ld_qkv, const float* table, void* out, int64_t ld_out, float* nlse,
                       int B, int H, int W, int C, int heads, int ws, int shift, float scale, int dtype, void* stream);
int rdst_wattn_bwd_lse(const void* qkv, int64_t ld_qkv, const float* table, const void* dout, int64_t ld_dout,
                       const void* out, int64_t ld_out, const float* nlse, void* dqkv, int64_t ld_dqkv, float* dtable,
                       void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws, int shift,
                       float scale, int dtype, void* stream);

/* ---- K8: the attention half of a Swin block in ONE launch ------------------------------------------------------
 *   x1 = X + proj(WindowAttention(qkv(LayerNorm(X))))     networks/swin_transformer_sr.py:240-271 with :110-141 inside
 * i.e. rdst_ln_linear_fwd (norm1 + qkv) -> rdst_wattn_fwd -> rdst_ln_linear_fwd (proj + shortcut) without the two round
 * trips through HBM in between: 6 C instead of 11 C bytes per token.  Writes everything the backward entry points above
 * read: qkv (M, 3C), a (M, C) = the attention output, x1 (M, C), stats (M, 2) {mean, rstd} of norm1; M = B*H*W.  Results are
 * bit-identical to the three-call sequence.  Covered: bf16, ws = 8, heads = 6, C = 60 / 90 / 120, the analytic
 * shifted-window mask (rdst_swin_attn_fwd_supported); everything else returns RDST_ENOTSUP and the caller composes the
 * three calls.  bqkv / bproj may be NULL.
 *   workspace : rdst_swin_attn_fwd_workspace(C) bytes, 16-byte aligned, for the packed weight images
 *               [RDST_PACK_LINEAR_SEC3 image of (Wqkv, ln_w, ln_b, bqkv)][RDST_PACK_LINEAR image of (Wproj, bproj)];
 *               RDST_PREPACKED = both are already there (rdst_pack_batch, the second at
 *               workspace + rdst_swin_attn_fwd_workspace(C) - rdst_ln_linear_fwd_workspace(C, C)). */
int rdst_swin_attn_fwd_supported(int C, int heads, int ws, int dtype);
size_t rdst_swin_attn_fwd_workspace(int C);
int rdst_swin_attn_fwd(const void* X, int64_t ld_x, const float* ln_w, const float* ln_b, const float* Wqkv,
                       const float* bqkv, const float* table, const float* Wproj, const float* bproj, void* qkv,
                       int64_t ld_qkv, void* a, int64_t ld_a, void* x1, int64_t ld_x1, float* stats, void* workspace,
                       size_t workspace_bytes, int B, int H, int W, int C, int heads, int ws, int shift, float scale,
                       int dtype, void* stream);

/* ---- batched weight packing ------------------------------------------------------------------------------
 * rdst_ln_linear_fwd and rdst_conv_fwd read their weights as bf16 fragment images that a small pack kernel writes into
 * the op's `workspace` on every call.  rdst_pack_batch writes the images of MANY layers in a handful of launches (e.g.
 * once at the start of a network's forward); an op is then called with workspace = that layer's image and
 * workspace_bytes = RDST_PREPACKED and skips its own pack.  `out` needs rdst_ln_linear_fwd_workspace(K, N) /
 * rdst_conv_fwd_workspace(Cin, Cout, 3) bytes, 16-byte aligned.  The weights must not change between the pack and the
 * ops that use it.  Linear: W (N, K), gamma / beta (K) or NULL, bias (N) or NULL, s = out_scale.
 * Conv forward: W (Cout = N, Cin = K, 3, 3), s = out_scale. */
#define RDST_PREPACKED ((size_t)-1)
#define RDST_PACK_LINEAR 0
#define RDST_PACK_CONV3_FWD 1
#define RDST_PACK_LINEAR_SEC3 2   /* a Linear whose N = 3 C outputs are the sections q | k | v, each padded to whole 32-row tiles
                                     (the qkv half of rdst_swin_attn_fwd's workspace; N % 3 == 0) */
#define RDST_PACK_LINEAR_X3 3     /* the RDST_F32X3 image of a Linear: hi / lo bf16 fragment pairs + b' (what rdst_ln_linear_fwd reads with
                                     dtype = RDST_F32X3 on the shapes rdst_ln_linear_fwd_packable reports; `out` needs
                                     rdst_ln_linear_fwd_workspace2(K, N, RDST_F32X3) bytes) */
#define RDST_PACK_CONV3_FWD_X3 4  /* the RDST_F32X3 image of a 3x3 convolution's forward weights (rdst_conv_fwd with dtype = RDST_F32X3 on the
                                     shapes rdst_conv_fwd_packable reports; `out` needs rdst_conv_fwd_workspace2(Cin, Cout, 3, RDST_F32X3)) */
typedef struct rdst_pack_job {
  int kind;
  const float* W; const float* gamma; const float* beta; const float* bias;
  void* out;
  int N, K;
  float s;
} rdst_pack_job;
int rdst_pack_batch(const rdst_pack_job* jobs, int njobs, void* stream);
/* 1 if a call of that shape reads a packed image (so prepacking it is useful), else 0 */
int rdst_ln_linear_fwd_packable(int K, int N, int has_ln, int has_residual, int in_act, int dtype);
size_t rdst_conv_fwd_workspace2(int Cin, int Cout, int ksize, int dtype);   /* rdst_conv_fwd_workspace per compute mode (ABI 11) */
int rdst_conv_fwd_packable(int Cin, int Cout, int ksize, int shuffle_r, int has_residual, int in_act, int dtype);

/* ---- batched fixed-order reductions -------------------------------------------------------------------
 * Every backward entry point above that splits its contraction over workgroups ends with a deterministic sum of
 * per-workgroup partial slabs (and, behind a LayerNorm, a finish kernel).  Between _begin() and _end() those
 * reductions are recorded instead of launched, and _end() runs all of them as two launches on `stream` — the
 * backward of a Swin block (rdst_mlp_bwd + 2 x rdst_ln_linear_bwd + rdst_wattn_bwd) then needs 2 reduction
 * launches instead of 6.  Thread-local; the ops' workspaces and outputs must stay alive until _end() returns and all
 * ops of a batch must be enqueued on `stream`.  Gradients are complete only after _end(). */
int rdst_reduce_batch_begin(void);
int rdst_reduce_batch_end(void* stream);
/* Close an open batch WITHOUT running its queued reductions (a backward that failed half way: the slabs and the
 * outputs the jobs name may already be freed).  No-op without an open batch.  (ABI 7) */
int rdst_reduce_batch_abort(void);

/* ---- layout helpers at the NCHW boundary of the module ------------------------------------------
 * nchw (B,C,H,W) fp32 <-> token rows (B*H*W, C) of `dtype`.  The caller-facing tensors of
 * RDSTSR.forward are fp32 NCHW (rdst_variations.py:1342-1360). */
int rdst_nchw_to_rows(const float* nchw, void* rows, int64_t ld, int B, int C, int H, int W,
                      int dtype, void* stream);
int rdst_rows_to_nchw(const void* rows, int64_t ld, float* nchw, int B, int C, int H, int W,
                      int dtype, void* stream);

/* ---- nearest-neighbour x2 upsampling of token-major rows (SwinIR 'nearest+conv' reconstruction) --------
 * y (B, 2H, 2W, C) = torch.nn.functional.interpolate(x, scale_factor=2, mode='nearest') of x (B, H, W, C) rows
 * (networks/swin_transformer_sr.py:801-802); backward: dx[b, i, j] = sum of the four dy[b, 2i+{0,1}, 2j+{0,1}]
 * (fixed order).  ld_* in elements; dtype RDST_F32 / RDST_BF16. */
int rdst_upsample2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, int B, int H, int W, int C, int dtype,
                       void* stream);
int rdst_upsample2_bwd(const void* dy, int64_t ld_dy, void* dx, int64_t ld_dx, int B, int H, int W, int C, int dtype,
                       void* stream);

/* ---- fused Adam over flat buffers ------------------------------------------------
 * One launch updates all `n` fp32 parameters in place with torch.optim.Adam's rule (amsgrad off),
 * the optimizer utils/optim.py:30-53 builds and models/trans_sr_trainer.py:170-173 steps:
 *   g = grad + weight_decay*p;  m += (g - m)(1 - beta1);  v = v*beta2 + (1 - beta2) g^2;
 *   p -= lr/(1 - beta1^step) * m / (sqrt(v)/sqrt(1 - beta2^step) + eps)
 * `step` is the 1-based count of this update.  All four buffers are 16-byte aligned device fp32. */
int rdst_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                   void* stream);

/* ---- N2 (full): the seg-UNet of the reference's perceptual loss (loss/seg_unet.py:46-127) -----------------------------
 * smp.Unet(in_channels, classes=4) = resnet34 encoder + UNet decoder + 3x3 head, BatchNorm in TRAINING mode, frozen
 * weights: forward on the SR and the HR batch, backward to the SR image ONLY (no weight gradients: the optimizer of the
 * reference holds model_g alone, models/trans_sr_trainer.py:72).  The host (rdst_amd/loss/seg_unet.py) composes the
 * network from the entry points below; activations are NHWC rows (pixel-major, `ld` in elements) of `dtype`
 * (RDST_F32 parity mode / RDST_BF16 throughput mode, fp32 accumulation and statistics).  `scratch` = at least
 * rdst_u_scratch_bytes() bytes of device memory that calls on one stream may share (partial sums, fixed-order reductions).
 */
size_t rdst_u_scratch_bytes(void);

/* Implicit-GEMM convolution on the matrix cores, k x k (1, 3), stride 1 / 2, zero padding k/2.
 *   input  = channel concat of source 1 (C1 channels; nearest-upsampled x2 on the fly when up1 = 1: it then holds
 *            (B, Hin/2, Win/2, C1)) and source 2 (C2 channels, may be NULL / 0) — the UNet decoder's
 *            F.interpolate(x, 2, 'nearest') + torch.cat([x, skip], 1) never materialises;
 *   Wp     = weights in `dtype`, [k*k][Npad][C1 + C2] (reduction index contiguous, Npad = Cout rounded up to 32, zero rows);
 *            RDST_F32X3: the same bytes per row, as 64-byte groups of [16 bf16 hi][16 bf16 lo] per 16 reduction elements
 *            (hi = bf16(w), lo = bf16(w - hi)): the frozen weights are split once, on the host;
 *   Y[b, oy, ox, :Cout] = sum_taps in[b, oy*stride + ky - k/2, ox*stride + kx - k/2, :] . Wp[tap] + bias + add
 *   transposed = 1: the data gradient of such a convolution (input = dY in the FORWARD's output geometry (Hin, Win),
 *            output = dX in the forward's input geometry (Hout, Wout)): in[b, (oy + k/2 - ky) / stride, ...] where divisible,
 *            with Wp = the forward's weights as [k*k][Cin_fwd padded][Cout_fwd].
 * (C1 + C2) * elementsize must be a multiple of 32 bytes and C1 * elementsize a multiple of the reduction chunk.
 *   bn1    = NULL, or the (scale[C1], shift[C1]) head of a rdst_u_bn_stats `coef`: source 1 is then read through
 *            relu(scale x + shift) — the train-mode BatchNorm + ReLU in front of this convolution (smp's Conv2dReLU,
 *            resnet's conv-bn-relu) applied while the input is staged, so that the activation never materialises; zero
 *            padding applies to the ACTIVATION.  fp32 / fp32x3, 3x3, stride 1, forward form only (RDST_ENOTSUP otherwise). */
int rdst_u_conv(const void* X1, int64_t ld1, int C1, int up1, const void* X2, int64_t ld2, int C2, const void* Wp,
                const float* bias, const void* add, int64_t ld_add, void* Y, int64_t ld_y, int B, int Hin, int Win, int Hout,
                int Wout, int Cout, int Npad, int ksize, int stride, int transposed, int dtype, void* stream,
                const float* bn1, float* stats, int* stats_blocks);
/*   stats  = NULL, or room for [blocks][2][Cout] floats, blocks <= B * ceil(Hout / 8) * ceil(Wout / 8): every workgroup leaves the
 *            sum and the sum of squares of its output pixels per channel (3x3 stride 1, no bias / addend), *stats_blocks (host)
 *            receives the number of blocks written; rdst_u_bn_stats_from finishes them — the BatchNorm statistics of the
 *            convolution's output without another pass over it. */

/* encoder.conv1: 7x7 stride 2 pad 3, Cin <= 4 -> 64, from the fp32 NCHW image; and its data gradient back to the image
 * (times the device scalar `upstream`, NULL = 1).  W = fp32 (64, Cin, 7, 7) as nn.Conv2d stores it. */
int rdst_u_stem_fwd(const float* img, const float* W, void* Y, int64_t ld_y, int B, int Cin, int H, int Wd, int dtype,
                    void* stream);
int rdst_u_stem_dgrad(const void* dR, int64_t ld, const float* W, const float* upstream, float* dimg, int B, int Cin, int H,
                      int Wd, int dtype, void* stream);

/* BatchNorm2d in training mode on rows X (P, C): batch statistics (biased variance, fixed-order sums) ->
 * coef[0..C) = a = gamma * rstd, coef[C..2C) = b = beta - mean * a, coef[2C..3C) = mean, coef[3C..4C) = rstd;
 * running_mean / running_var (may be NULL) updated as nn.BatchNorm2d does (momentum, unbiased variance). */
int rdst_u_bn_stats(const void* X, int64_t ld, int64_t P, int C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* coef, void* scratch, int dtype,
                    void* stream);
int rdst_u_bn_stats_from(const float* partials, int nblk, int64_t P, int C, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* coef, void* scratch, void* stream);
/* Y = act(a X + b [+ a2 X2 + b2] [+ R]), act = ReLU when relu != 0: BatchNorm apply, the BasicBlock's identity /
 * downsample branch and the activation in one pass. */
int rdst_u_bn_apply(const void* X, int64_t ldx, const float* coef, const void* X2, int64_t ldx2, const float* coef2,
                    const void* R, int64_t ldr, int relu, void* Y, int64_t ldy, int64_t P, int C, int dtype, void* stream);
/* Backward of Y = act(BatchNorm_train(Xraw) + ...) to Xraw: g = dY * [Ymask > 0] (Ymask NULL: g = dY; Ymask == Xraw, the
 * same pointer: Y = relu(BatchNorm(Xraw)) was never materialised and the mask is recomputed from Xraw and coef),
 * dX = a (g - mean(g) - xhat mean(g xhat)); Gout (may be NULL) receives g (+ Gadd), the gradient of the "+ ..." operand. */
int rdst_u_bn_bwd(const void* dY, int64_t lddy, const void* Ymask, int64_t ldm, const void* Xraw, int64_t ldx,
                  const float* coef, void* dX, int64_t lddx, void* Gout, int64_t ldg, const void* Gadd, int64_t ldga,
                  int64_t P, int C, void* scratch, int dtype, void* stream);

/* MaxPool2d(3, 2, 1) (the first maximum of a window in row-major scan order wins, as ATen), its backward (+ add), and the
 * backward of the nearest x2 upsampling (2x2 sums of dY (B, 2H, 2W, C) (+ add)). */
int rdst_u_maxpool_fwd(const void* X, int64_t ldx, void* Y, int64_t ldy, uint8_t* idx, int B, int H, int W, int C, int dtype,
                       void* stream);
int rdst_u_maxpool_bwd(const void* dY, int64_t lddy, const uint8_t* idx, const void* add, int64_t ld_add, void* dX,
                       int64_t lddx, int B, int H, int W, int C, int dtype, void* stream);
int rdst_u_sumpool2(const void* dY, int64_t lddy, const void* add, int64_t ld_add, void* dX, int64_t lddx, int B, int H,
                    int W, int C, int dtype, void* stream);

/* loss[0] = (accumulate ? loss[0] : 0) + weight * mean((A - B)^2) (mse = 1) or weight * mean(|A - B|) over (P, C) rows;
 * backward: dA = (add +) weight * upstream[0] * d(mean)/dA. */
int rdst_u_pair_loss_fwd(const void* A, int64_t lda, const void* Bv, int64_t ldb, int64_t P, int C, int mse, float weight,
                         int accumulate, float* loss, void* scratch, int dtype, void* stream);
int rdst_u_pair_loss_bwd(const void* A, int64_t lda, const void* Bv, int64_t ldb, int64_t P, int C, int mse, float weight,
                         const float* upstream, const void* add, int64_t ld_add, void* dA, int64_t ldda, int dtype,
                         void* stream);

/* smp.losses.DiceLoss('multiclass', classes) on logits (P, ncls <= 8): targets = argmax of `target_logits` (first
 * maximum; the 'label-hr' mode, loss/seg_unet.py:112-116) or `labels` (int64, 'label-gt'); class_mask bit c = class c is
 * averaged.  coef (2 * ncls floats) carries d(loss)/d(probability) to the backward, which writes dlogits (P, ncls_pad):
 * channels >= ncls are zeroed (the head's data-gradient conv wants a 32-byte reduction). */
int rdst_u_dice_fwd(const void* logits, int64_t ld, const void* target_logits, int64_t ldt, const int64_t* labels, int64_t P,
                    int ncls, int class_mask, float eps, float weight, int accumulate, float* loss, float* coef,
                    void* scratch, int dtype, void* stream);
int rdst_u_dice_bwd(const void* logits, int64_t ld, const void* target_logits, int64_t ldt, const int64_t* labels, int64_t P,
                    int ncls, const float* coef, const float* upstream, void* dlogits, int64_t ldd, int ncls_pad, int dtype,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RDST_HIP_H */
