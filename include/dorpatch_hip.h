/*
 * dorpatch_hip.h — C ABI of libdorpatch_hip.so (gfx950 / MI355X).
 *
 * The DorPatch EOT hot path (reference attack.py:167-342) as hand-written HIP
 * kernels behind plain-C entry points.  Conventions for EVERY entry point:
 *
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch, or any
 *     other HIP allocator); the library allocates nothing and keeps no state;
 *   - tensors are dense, row-major, fp32 unless stated; images are NCHW;
 *     "P" = H*W pixels, W % 4 == 0 (16-byte vector access never straddles a row);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous and
 *     ordered on that stream;
 *   - the return value is a hipError_t as int: 0 = success, 1 =
 *     hipErrorInvalidValue for bad arguments, anything else = launch failure.
 *     dp_error_string() maps it to text.  The Python host raises RuntimeError.
 *
 * The reference has no FFI of its own (it is 4 Python files); each entry point
 * below cites the reference Python lines it replaces.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 */
#ifndef DORPATCH_HIP_H
#define DORPATCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DP_ABI_VERSION 12
#define DP_MAX_RECTS 4 /* occlusion windows per mask-table entry */

typedef void *dp_stream_t; /* hipStream_t */
typedef void *dp_event_t;  /* hipEvent_t  */

/* Per-channel input normalisation fused into the occlusion kernels
 * (reference utils.py:66-78 NormModel: (x - mean) / std, mean = std = 0.5).
 * enable == 0: emit raw [0,1] pixels and fill occluded pixels with `fill`
 * (reference attack.py:206 / PatchCleanser.py:99-100 use 0.5). */
typedef struct {
  int enable;
  float mean[3];
  float std[3];
  float fill;
} dp_norm_t;

int dp_abi_version(void);
const char *dp_error_string(int err);

/* Launch-geometry overrides for tests and A/B measurements — NOT part of the product path (nothing in
 * dorpatch_amd/attack.py calls it).  Every knob is process-wide, defaults to 0 = the product's own choice, changes
 * only how a result is computed, never the result (the tests that use it assert exactly that).  Replaces the
 * DORPATCH_AFFINE_SPB environment variable of ABI 7.  Returns 1 (hipErrorInvalidValue) for an unknown knob / value.
 *   DP_DEBUG_AFFINE_SAMPLES_PER_BLOCK  samples one dp_apply_affine_fwd workgroup walks (1..64; 0: sized from the grid)
 *   DP_DEBUG_UPDATE_VARIANT            dp_project_update: 1 = the 4-byte-lane kernel even where the 16-byte-lane
 *                                      kernel applies (0: 16-byte lanes whenever W % 4 == 0 and pointers are aligned)
 *   DP_DEBUG_APPLY_ORDER               dp_apply_fwd grid walk: 1 = XCD-aware (a tile's samples adjacent on one XCD: 10 %
 *                                      less HBM traffic, measured 19 % slower); 0: tile-fastest 3-D grid (one ascending
 *                                      output stream)
 *   DP_DEBUG_AFFINE_GATHER             dp_apply_affine_bwd: 1 = branch-free gather loop (same candidates, same order, same
 *                                      bits; measured 34 % slower); 2 = hit compaction (round 5: a 64-bit mask of the
 *                                      window's taps, then only the lane's own hits; same bits; measured 22 % slower);
 *                                      0: a branch per candidate
 *   DP_DEBUG_CONV1X1_VARIANT           dp_conv1x1_fwd: bits 0-1 workgroup id -> (pixel tile, channel group): 0 XCD-aware
 *                                      (a tile's channel groups adjacent on one XCD), 1 channel group fastest, 2 tile
 *                                      fastest; bit 2 non-temporal result stores; bit 3 the next chunk goes to LDS in one
 *                                      lump half-way through the chunk (0: one item after each MFMA group); bits 4-6
 *                                      (round 6) force the workgroup's pixel tile: 1 = 448, 2 = 256, 3 = 128, 4 = 64 pixels
 *                                      (0: the launcher picks the largest tile whose grid fills the chip).
 *                                      Same bits out of every variant
 *   DP_DEBUG_CONV3X3_VARIANT           dp_conv3x3_fwd / dp_conv3x3_gn_fwd: 1 = k_conv3x3_mfma (zero rows / columns laid out
 *                                      in LDS) on every side it takes (56 / 28 / 14 / 7), 2 = k_conv3x3_flat (flat LDS
 *                                      image, masked taps) on every side, 0 = the measured winner per side; bits 4-6
 *                                      (round 6) force k_conv3x3_flat's pixel tile on the sides 28 / 14 / 7 / 24 / 12:
 *                                      1 = 448, 3 = 128, 4 = 64 pixels (0: the launcher's rule).  Same bits */
#define DP_DEBUG_AFFINE_SAMPLES_PER_BLOCK 1
#define DP_DEBUG_UPDATE_VARIANT 2
#define DP_DEBUG_APPLY_ORDER 3
#define DP_DEBUG_AFFINE_GATHER 4
#define DP_DEBUG_CONV1X1_VARIANT 5
#define DP_DEBUG_CONV3X3_VARIANT 6
int dp_debug_set(int knob, int value);

/* ---- a-2  utils.clip (utils.py:105-110) + adv_x = delta + x (attack.py:184-185) ---- */

/* Number of per-image partial sums dp_sumsq_partials writes for P pixels. */
int dp_sumsq_nchunk(int P);

/* partials[b*nchunk + k] = sum over chunk k of (mask*(pattern-x))^2
 * (first pass of torch.norm(delta_x, p=2, dim=(1,2,3)), utils.py:107-108).
 * mask (B,1,P) pattern,x (B,3,P); partials (B,nchunk). Deterministic. */
int dp_sumsq_partials(const float *mask, const float *pattern, const float *x,
                      int B, int P, float *partials, dp_stream_t stream);

/* scale[b] = min(eps / sqrt(sum_k partials[b,k]), 1)   (utils.py:109, detached)
 * adv_x    = x + (mask*(pattern-x)) * scale            (utils.py:110, attack.py:185)
 *            (add_x = 0: only the perturbation delta_x * scale, i.e. utils.clip's
 *             return value, for callers that add x themselves — main.py:140-141)
 * l2[b]    = sqrt(sum)                                 (attack.py:323 log value, pre-scale)
 * adv_x (B,3,P); scale,l2 (B). */
int dp_blend(const float *mask, const float *pattern, const float *x,
             const float *partials, float eps, int B, int P, int add_x,
             float *adv_x, float *scale, float *l2, dp_stream_t stream);

/* ---- a-4 / a-10  occlusion apply (attack.py:204-220, PatchCleanser.py:44-59,99-100) ----
 *
 * The reference materialises every mask as a (1,H,W) bool tensor (126 MB for
 * the 2520-mask universe @224).  Here a mask is `R` axis-aligned windows
 * table[m][r] = {row0,row1,col0,col1} (half-open, int32); a pixel is occluded
 * iff it lies in any window.  idx (B,S) picks the mask of every EOT sample;
 * idx_bstride = S for per-image draws, 0 to share one (S,) draw across images.
 * idx2 (same shape, may be NULL) is the reference's `dual` second mask
 * (attack.py:208-218).
 *
 * out[b,s,c,h,w] = occluded ? fill : adv_x[b,c,h,w], then optionally
 * normalised (dp_norm_t).  out is (B,S,3,H,W) == (B*S,3,H,W).
 * Algorithmic HBM traffic: B*S*3*P*4 bytes written, B*3*P*4 read.           */
int dp_apply_fwd(const float *adv_x, const int32_t *table, int R,
                 const int32_t *idx, const int32_t *idx2, int idx_bstride,
                 int B, int S, int H, int W, const dp_norm_t *norm, float *out,
                 dp_stream_t stream);

/* Number of S-slabs dp_apply_bwd writes (so that B*slabs*tiles fills the chip
 * while the S-reduction order stays fixed). */
int dp_apply_bwd_nslab(int B, int S, int P);

/* Backward of dp_apply_fwd w.r.t. adv_x (autograd of attack.py:206-220 +
 * utils.py:77-78):  slabs[z,b,c,p] = sum_{s in slab z} keep(s,p) * G[b,s,c,p] / std_c
 * G (B,S,3,P) is d loss / d out.  slabs (nslab,B,3,P).  The slab count comes
 * from dp_apply_bwd_nslab(); follow with dp_sum_slabs.  Deterministic order. */
int dp_apply_bwd(const float *G, const int32_t *table, int R,
                 const int32_t *idx, const int32_t *idx2, int idx_bstride,
                 int B, int S, int H, int W, const dp_norm_t *norm,
                 float *slabs, dp_stream_t stream);

/* ---- EXTENSION (absent from the reference): per-sample affine placement of the patch, fused with the apply ----
 * BASELINE.json's north_star names "random affine placement"; the reference blends the patch at identity only
 * (attack.py:184-185; its sole trace of transforms is the unused hook of collect_failure, attack.py:384, 395-396).
 * theta (B,S,2,3) fp32 maps OUTPUT pixel coordinates (ox, oy) to SOURCE coordinates in delta:
 *   sx = theta[0]*ox + theta[1]*oy + theta[2],  sy = theta[3]*ox + theta[4]*oy + theta[5]
 * bilinear, zero outside the image — F.grid_sample(delta, F.affine_grid(theta_norm), align_corners=False); the host
 * side (dorpatch_amd/placement.py) converts.  delta (B,3,H,W) = dp_blend(..., add_x = 0).
 *   dp_apply_affine_fwd  out[b,s] = occlude(norm(x[b] + warp(delta[b], theta[b,s])))          (B*S,3,H,W)
 *   dp_apply_affine_bwd  slabs[z,b] = (1/std) sum_{s in slab z} warp^T(keep(s) * G[b,s])      exact adjoint, gather
 *                        form (fixed summation order, no atomics); theta_inv (B,S,2,3) = the inverse maps, used
 *                        only to position the search window; nslab = dp_apply_bwd_nslab(B, S, H*W), then dp_sum_slabs.
 * Identity theta reproduces dp_apply_fwd / dp_apply_bwd bit for bit.  Limits (hipErrorInvalidValue beyond them):
 * 12 * H * W < 2^31 (32-bit byte offsets into one image's delta), H, W < 2^22. */
int dp_apply_affine_fwd(const float *x, const float *delta, const float *theta, const int32_t *table, int R,
                        const int32_t *idx, const int32_t *idx2, int idx_bstride, int B, int S, int H, int W,
                        const dp_norm_t *norm, float *out, dp_stream_t stream);
int dp_apply_affine_bwd(const float *G, const float *theta, const float *theta_inv, const int32_t *table, int R,
                        const int32_t *idx, const int32_t *idx2, int idx_bstride, int B, int S, int H, int W,
                        const dp_norm_t *norm, float *slabs, dp_stream_t stream);

/* out[i] = (accumulate ? out[i] : 0) + sum_{z<nslab} slabs[z*n + i], z ascending. */
int dp_sum_slabs(const float *slabs, int nslab, int64_t n, float *out,
                 int accumulate, dp_stream_t stream);

/* ---- a-7  CW_loss.__call__ + its gradient (attack.py:16-23, 224-230, 247) ----
 * logits (N,C); y (B,) int64 label and targeted (B,) int32 flag of image n / S
 * (a batch is B independent single-image problems; each may have switched to a
 * targeted criterion on its own — attack.py:106-122, 169-176).
 * loss[n]    = max(conf + other - real, 0)  (targeted)  | max(conf + real - other, 0)
 * dlogits    = upstream * d loss[n] / d logits[n,:]     (may be NULL: forward only)
 * pred[n]    = argmax_k logits[n,k] (int32, may be NULL) */
int dp_cw_loss(const float *logits, const int64_t *y, const int32_t *targeted,
               int N, int C, int S, float confidence, float upstream,
               float *loss, float *dlogits, int32_t *pred, dp_stream_t stream);

/* ---- a-5  local_variance / min_var_weighted_variance (attack.py:33-45, 100, 227-228) ---- */

/* lv[b,h,w] = mean_c( |x[h,w]-x[h,w+1]| + |x[h,w]-x[h+1,w]| ), last column /
 * last row keep the raw pixel (attack.py:35-39 slice semantics).  x (B,3,H,W). */
int dp_local_variance(const float *x, int B, int H, int W, float *lv,
                      dp_stream_t stream);

/* Number of per-image tile partials dp_struct_loss writes. */
int dp_struct_ntile(int H, int W);

/* partials[b,t] = sum over tile t of mean_c(L(adv_x)) / (lv_x + 1e-5)
 * loss_struc[b] = sum_t partials[b,t] / P  via dp_reduce_rows(scale = 1/P). */
int dp_struct_loss(const float *adv_x, const float *lv_x, int B, int H, int W,
                   float *partials, dp_stream_t stream);

/* out[b] = scale * sum_k in[b*n + k], k ascending (one wave per row). */
int dp_reduce_rows(const float *in, int B, int n, float scale, float *out,
                   dp_stream_t stream);

/* ---- a-6  density + group-lasso statistics of the mask (attack.py:72-80, 237-245) ----
 * cell_sumsq (B,ncy,ncx): sum of mask^2 per unit x unit cell (conv_group(mask**2))
 * win_sum    (B,nwy,nwx): sum of mask per win x win window  (conv_density(mask))
 * group_lasso[b] = unit * sum_cells sqrt(cell_sumsq)          (attack.py:243-244)
 * density[b]     = unbiased variance of the window sums       (attack.py:237)
 * ncy = (H-unit)/unit+1, nwy = (H-win)/win+1 (stride = kernel, no padding). */
int dp_mask_stats(const float *mask, int B, int H, int W, int unit, int win,
                  float *cell_sumsq, float *win_sum, float *group_lasso,
                  float *density, dp_stream_t stream);

/* ---- a-2 (backward) + a-5 (gradient) + a-6 (gradients) + a-9 (signed update) ----
 * One fused pass over the B images (attack.py:227-247 backward, 333-342 update):
 *
 *   g      = g_adv + d(structured_b * loss_struc)/d adv_x        (a-5 gradient)
 *   g_pat  = g * mask * scale_b                                  (utils.py:107-110 bwd)
 *   g_mask = sum_c g * (pattern - x) * scale_b
 *            + density * 2/(n_win-1) * (win_sum - mean)          (stage 0)
 *            + coeff_gl_b * unit / (2 sqrt(cell_sumsq)) * 2 mask (stage 0; 0*inf = NaN kept)
 *   if save_best[b]: best_pattern = pattern (, best_mask = mask in stage 0)   (attack.py:286-289)
 *   pattern -= lr_b * sign(g_pat);  clamp      (sign(NaN) = 0, as torch.sign)
 *   mask    -= lr_b * sign(g_mask); clamp      (stage 0 only)
 *
 * g_pattern_out / g_mask_out (may be NULL) receive the raw gradients for
 * parity tests.  do_update = 0 leaves pattern/mask untouched.
 * Per-image arrays (B,): scale, structured, coeff_gl, lr; save_best int32 (may be NULL). */
typedef struct {
  int B, H, W;
  int stage;      /* 0: mask is learned; 1: mask frozen */
  int unit, win;  /* group-lasso cell, density window (stage 0) */
  int do_update;
  float density;  /* coefficient (attack.py:239-240), 0 disables */
  float clip_min, clip_max;
} dp_update_cfg_t;

int dp_project_update(const dp_update_cfg_t *cfg, const float *x,
                      const float *adv_x, const float *lv_x, const float *g_adv,
                      const float *scale, const float *structured,
                      const float *coeff_gl, const float *lr,
                      const float *cell_sumsq, const float *win_sum,
                      const int32_t *save_best, float *pattern, float *mask,
                      float *best_pattern, float *best_mask,
                      float *g_pattern_out, float *g_mask_out,
                      dp_stream_t stream);

/* ---- a-8: 3x3 / stride 1 / pad 1 convolutions of the frozen backbone on the matrix cores ----
 * (attack.py:222, 247 through the classifier; MIOpen runs them as fp32 Winograd on the VALUs.)  Direct implicit GEMM on
 * v_mfma_f32_32x32x2_f32: exact f32, y[n][o] = sum_{c,kh,kw} w[o][c][kh][kw] * x[n][c][.+kh-1][.+kw-1], zero padding.
 * Shapes: H = W in {56, 28, 14, 7} (the planes of ResNetV2-50 at 224 x 224) or {96, 48, 24, 12} (at 384 x 384),
 * C % 8 == 0, O % 64 == 0.  Two kernels, same bits: k_conv3x3_mfma (zero rows / columns laid out in LDS; the first four
 * sides) and k_conv3x3_flat (flat LDS image + masked taps; all eight) — the measured winner per side runs.
 * x (N,C,H,W), y (N,O,H,W), dense NCHW.  wt = the weights PRE-PACKED for the kernel's k-walk (frozen: packed once by the
 * host, dorpatch_amd/ops.py pack_conv3x3_weights):
 *   wt[og][chunk][cp][kh][kw][half][o] = w[64 og + o][8 chunk + 2 cp + half][kh][kw],   O/64 x C/8 x 4 x 3 x 3 x 2 x 64.
 * The input gradient of the same convolution is the same entry point on dy with the weights transposed and flipped
 * (w'[c][o][kh][kw] = w[o][c][2-kh][2-kw]).  Deterministic (fixed summation order: channels ascending, taps row-major).
 * Measured against MIOpen by scripts/conv3x3_vs_miopen.py; routed per shape by dorpatch_amd/libconv.py. */
int dp_conv3x3_fwd(const float *x, const float *wt, int N, int C, int O, int H, int W, float *y, dp_stream_t stream);
/* Round 6: the same convolution as Winograd F(2 x 2, 3 x 3) with its 16 position GEMMs on v_mfma_f32_32x32x2_f32 — 16
 * multiplications per 2 x 2 output tile instead of 36 (the route the reference's own library takes for a 3x3 convolution,
 * attack.py:222, 247 through the classifier).  H = W in {56, 28, 14, 7, 96, 48, 24, 12}, C % 8 == 0, O % 64 == 0; ab: NULL, or the (N,C,2)
 * coefficients of dp_gn_stats (the GroupNorm-apply + ReLU folded into the patch loads; the zero padding pads the normalised
 * activation).  wt = the WINOGRAD-DOMAIN filter U = G g G^T (4 x 4 per (o, c), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]),
 * pre-packed in the lanes' OPERAND order (dorpatch_amd/ops.py pack_conv3x3_wino_weights, computed in fp64, rounded once):
 *   wt[og][chunk][4 xi + nu][half][lane][t][f] = U[64 og + 32 f + lane][4 chunk + 2 t + half][xi][nu],
 *   O/64 x C/4 x 16 x 2 x 32 x 2 x 2 — the float4 a lane loads is its four A operands of one (chunk, position).
 * The input gradient is the same entry point on dy with the weights transposed and flipped.  fp32 throughout, fixed
 * summation order (channels ascending): deterministic; NOT the bits of dp_conv3x3_fwd — F(2 x 2, 3 x 3) adds 16 transformed
 * products where the direct form adds 9 plain ones (measured ~1e-6 of the output scale apart; tests hold it to 2e-5). */
int dp_conv3x3_wino_fwd(const float *x, const float *wt, const float *ab, int N, int C, int O, int H, int W, float *y,
                        dp_stream_t stream);
/* The same convolution of relu(group_norm(x)), the GroupNorm-apply + ReLU folded into the operand staging (round 5): x is
 * the RAW tensor, ab (N,C,2) the coefficients dp_gn_stats wrote; bit-identical to dp_gn_relu_fwd followed by dp_conv3x3_fwd
 * (the zero padding pads the normalised activation).  Every side of dp_conv3x3_fwd except 7. */
int dp_conv3x3_gn_fwd(const float *x, const float *wt, const float *ab, int N, int C, int O, int H, int W, float *y,
                      dp_stream_t stream);

/* ---- a-8: the 3x3 / STRIDE 2 / pad 1 convolutions of the frozen backbone on the matrix cores (round 5) ----
 * (attack.py:222 through the classifier: conv2 of the first bottleneck of stages 2-4; MIOpen runs them as NHWC implicit
 * GEMMs between batched_transpose_* kernels, the 56 -> 28 one as a stride-2 Winograd.)  Same exact-f32 MFMA walk, same
 * packed weights (pack_conv3x3_weights) and the same summation order as dp_conv3x3_fwd:
 *   y[n][o][h][w] = sum_{c,kh,kw} w[o][c][kh][kw] * x'[n][c][2h+kh-1][2w+kw-1],   x (N,C,H,W) -> y (N,O,H/2,W/2).
 * H = W in {56, 28, 14} (the INPUT side), C % 8 == 0, O % 64 == 0.  ab == NULL: x' = x; else x' = relu(group_norm(x)) with
 * the (N,C,2) coefficients dp_gn_stats wrote, applied while staging (bit-identical to dp_gn_relu_fwd first). */
int dp_conv3x3s2_fwd(const float *x, const float *wt, const float *ab, int N, int C, int O, int H, int W, float *y,
                     dp_stream_t stream);
/* The INPUT GRADIENT of the same convolution (attack.py:247 through the classifier; MIOpen: NHWC implicit GEMM between
 * batched_transpose_* kernels):  dx[n][c][i][j] = sum_{o,kh,kw : i = 2h+kh-1, j = 2w+kw-1} w[o][c][kh][kw] * dy[n][o][h][w],
 * dy (N,O,Ho,Wo) -> dx (N,C,2Ho,2Wo), Ho = Wo in {28, 14, 7} (224 x 224 inputs) or {48, 24, 12} (384 x 384), O % 16 == 0,
 * C % 64 == 0.  Four parity classes (i & 1, j & 1) of 1 / 2 / 2 / 4 taps over the dy plane, each an exact-f32 MFMA walk over
 * K = (o ascending, taps row-major) — deterministic; one launch.  Two forms, same bits, each with its own packing of the
 * frozen weights (dorpatch_amd/ops.py pack_conv3x3s2_dgrad_weights):
 *   DP_S2BWD_PAIRS    a workgroup owns both column classes of a row parity pr and stores them interleaved (8-byte words);
 *                     wt = row class 1 then 0, each  [og][chunk][cp][th][j][half][c'] = w[8 chunk + 2 cp + half][64 og + c']
 *                     [kh][kw]  with kh = 1 if pr == 0 else (2, 0)[th], kw = (1, 2, 0)[j];   C/64 x O/8 x 4 x (1+pr) x 3 x 2 x 64
 *   DP_S2BWD_CLASSES  one class per workgroup (4-byte words 8 bytes apart); wt = classes (1,1), (0,1), (1,0), (0,0), each
 *                     [og][chunk][cp][th][tw][half][c'] with CH = 16 / ((1+pr)(1+pc)) channels per chunk, kw = 1 if pc == 0
 *                     else (2, 0)[tw];   C/64 x O/CH x CH/2 x (1+pr) x (1+pc) x 2 x 64. */
#define DP_S2BWD_PAIRS 0
#define DP_S2BWD_CLASSES 1
int dp_conv3x3s2_bwd(const float *dy, const float *wt, int N, int O, int C, int Ho, int Wo, float *dx, int form,
                     dp_stream_t stream);

/* ---- a-8: the stem convolution (3 -> 64, 7x7 / stride 2 / pad 3) on the matrix cores (round 5) ----
 * (attack.py:222 through the classifier; MIOpen: stride-2 Winograd at 52 TFLOP/s.)  y[n][o][h][w] = sum_{c,kh,kw}
 * w[o][c][kh][kw] * x[n][c][2h+kh-3][2w+kw-3], zero padding; x (N,3,H,224) -> y (N,64,H/2,112), H even.  Exact f32 on
 * v_mfma_f32_32x32x2_f32, fixed order (deterministic).  wt = the frozen filter packed for the kernel's 77 k-steps
 * (dorpatch_amd/ops.py pack_stem_weights):  wt[7 j + kw][half][o] = w[o][r % 3][r / 3][kw] with r = 2 j + half (0 for r = 21). */
int dp_stem_conv_fwd(const float *x, const float *wt, int N, int H, int W, float *y, dp_stream_t stream);

/* ---- a-8: 1x1 / stride 1 convolutions of the frozen backbone on the matrix cores (round 5) ----
 * (attack.py:222, 247 through the classifier: 33 of ResNetV2-50's 53 convolutions; until round 4 batched library GEMMs /
 * MIOpen NHWC kernels.)  On the NCHW tensors as they lie:  y[n] (O x HW) = W (O x C) x[n] (C x HW), an exact-f32 fmaf chain
 * over the input channels in ascending order on v_mfma_f32_32x32x2_f32 (deterministic).  The input gradient of the same
 * convolution is the same entry point on dy with the TRANSPOSED weights packed.
 * x (N,C,HW), y (N,O,HW) dense; C % 16 == 0, O % 64 == 0; HW % 4 == 0, or HW == 49 (the 7 x 7 planes; no `ab` there).
 * wt = the weights pre-packed for the kernel's k-walk (frozen: packed once, dorpatch_amd/ops.py pack_conv1x1_weights):
 *   wt[og][chunk][k][o] = w[64 og + o][16 chunk + k],     O/64 x C/16 x 16 x 64.
 * ab  (may be NULL) (N,C,2): fold `relu(group_norm(x))` into the operand staging — the kernel reads the RAW x and applies
 *     max(x * ab[n][c][0] + ab[n][c][1], 0) on the way to LDS, with the coefficients dp_gn_stats wrote: the normalised
 *     activation never exists in HBM (saves its 4 B/elem write and 4 B/elem re-read).  Bit-identical to dp_gn_relu_fwd
 *     followed by the plain convolution.
 * res (may be NULL; may alias y) (N,O,HW): y = conv + res — the bottleneck's residual add, or the accumulation of a second
 *     branch's input gradient, in the epilogue. */
int dp_conv1x1_fwd(const float *x, const float *wt, const float *ab, const float *res, int N, int C, int O, int HW,
                   float *y, dp_stream_t stream);

/* ---- next-1  collect_failure (attack.py:384-406) / PatchCleanser (PatchCleanser.py:68-112) ----
 * pred[n] = argmax_k logits[n,k]  (first index on ties). */
int dp_argmax(const float *logits, int N, int C, int32_t *pred,
              dp_stream_t stream);

/* ---- a-8  [residual add +] GroupNorm + ReLU of the frozen backbone, fused (the HBM-bound part) ----
 * timm 0.6.7 GroupNormAct / PreActBottleneck as used by resnetv2_50x1_bit_distilled (reference
 * utils.py:51-63; executed at attack.py:222 forward / :247 backward):
 *     s = x + res   (res == NULL: s = x)          -- the bottleneck's "out + shortcut"
 *     y = relu(group_norm(s, G, gamma, beta, eps))  -- the next block's norm1 (or the final norm)
 * x, res, sum_out, y, dy, dres, dx (N,C,HW) fp32 NCHW; gamma, beta (C); mean, rstd (N*G) saved for
 * the backward.  Requires (C/G)*HW % 4 == 0.  sum_out receives s when res is given.
 * Backward (frozen weights: only the input gradient; `x` is s, the tensor that was normalised):
 *     dx = dres + rstd * (dxh - mean_L(dxh) - xh * mean_L(dxh * xh)),  dxh = dy * [y > 0] * gamma
 * where dres (may be NULL) is the gradient reaching s through the shortcut — d loss/d x and
 * d loss/d res are both dx.
 * Algorithmic HBM traffic per element: forward 4 B read + 4 B write (+ 4 + 4 with res), backward
 * 8 B read + 4 B write (+ 4 with dres); groups up to 7168 float4 stay in registers, larger groups
 * are re-read (+8 B per direction). */
int dp_gn_relu_fwd(const float *x, const float *res, float *sum_out, const float *gamma,
                   const float *beta, int N, int C, int HW, int G, float eps, float *y, float *mean,
                   float *rstd, dp_stream_t stream);
int dp_gn_relu_bwd(const float *dy, const float *dres, const float *x, const float *gamma,
                   const float *beta, const float *mean, const float *rstd, int N, int C, int HW,
                   int G, float *dx, dp_stream_t stream);
/* Statistics-only form of dp_gn_relu_fwd (round 5), for consumers that apply the affine + ReLU themselves while staging
 * their operand (dp_conv1x1_fwd / dp_conv3x3_fwd with `ab`): same reads, same exact two-pass mean / variance, same
 * optional residual add (sum_out = x + res), but y is never written; instead ab (N,C,2) receives, per sample and channel,
 * a = rstd * gamma[c] and b = beta[c] - mean * a — the coefficients dp_gn_relu_fwd's store phase uses.  mean / rstd as
 * dp_gn_relu_fwd (the backward still needs them).  Traffic: 4 B/elem read (+ 4 read + 4 write with res). */
int dp_gn_stats(const float *x, const float *res, float *sum_out, const float *gamma, const float *beta, int N, int C,
                int HW, int G, float eps, float *mean, float *rstd, float *ab, dp_stream_t stream);
/* Gather form of the backward, for a backward pass over only the EOT samples that still carry gradient (the CW hinge
 * of attack.py:16-23 gives an exactly zero logit gradient to every sample whose margin is met, and the frozen,
 * per-sample-normalised backbone then yields an exactly zero input gradient for it): output sample n (of M) takes its
 * x / mean / rstd from SOURCE sample smap[n] (device, M int32, each in [0, n_tabs*tab_rows)).  Source sample s lives
 * in slab x_tabs[s / tab_rows] at row s % tab_rows — x_tabs is a HOST array of n_tabs (<= 8) device pointers, the
 * GroupNorm inputs saved by the micro-batches of one step's forward; mean / rstd are indexed by s*G + g.  dy, dres,
 * dx are dense (M,C,HW).  Same arithmetic and traffic as dp_gn_relu_bwd. */
int dp_gn_relu_bwd_gather(const float *dy, const float *dres, const float *const *x_tabs, int n_tabs,
                          int tab_rows, const int32_t *smap, const float *gamma, const float *beta,
                          const float *mean, const float *rstd, int M, int C, int HW, int G, float *dx,
                          dp_stream_t stream);

/* ---- a-8  ConstantPad2d(1, 0) + MaxPool2d(3, stride 2) of the BiT stem, fused ----
 * (timm 0.6.7 create_resnetv2_stem 'fixed'; reference utils.py:51-63, attack.py:222, 247.)
 * x (NC,Hin,Win) -> y (NC,Hin/2,Win/2); code (NC,Hin/2,Win/2) uint8 = 3*r + c of the winning
 * window position (row-major first max, NaN wins; a padded zero can win, its gradient is dropped).
 * Backward is a gather: dx[h,w] = sum over the <= 4 windows containing (h,w) whose code points at it.
 * Requires Hin even, Win % 8 == 0.  Traffic per input element: forward 4 B read + 1.25 B write,
 * backward 1.25 B read + 4 B write. */
int dp_pad_maxpool_fwd(const float *x, int64_t NC, int Hin, int Win, float *y, uint8_t *code,
                       dp_stream_t stream);
int dp_pad_maxpool_bwd(const float *dy, const uint8_t *code, int64_t NC, int Hin, int Win, float *dx,
                       dp_stream_t stream);

/* ---- a-8  input gradient of the BiT stem convolution (7x7, stride 2, pad 3, 3 input channels) ----
 * (timm 0.6.7 resnetv2 stem.conv; the last convolution of the backward pass, reference
 * attack.py:247 — its result is what dp_apply_bwd reduces over the EOT samples.)
 * dy (N,K,Ho,Wo) = d loss / d conv-out, w (K,3,7,7) the (standardised) filter,
 * dx (N,3,2*Ho,2*Wo) = d loss / d conv-in:
 *   dx[n,c,h,w] = sum_k sum_{i,j} dy[n,k,(h+3-i)/2,(w+3-j)/2] * w[k,c,i,j]   (even differences only)
 * Direct fp32 gather on the VALU (2*K*147 flop per 2x2 output quad); compute-bound. */
int dp_stem_dgrad(const float *dy, const float *w, int N, int K, int Ho, int Wo, float *dx,
                  dp_stream_t stream);

/* ---- a-8 + a-4 backward, fused: stem input gradient, occlusion-masked and reduced over the EOT samples ----
 * dp_stem_dgrad followed by dp_apply_bwd in one launch: dy (B*S,K,Ho,Wo) = d loss / d stem-conv-out of the B*S
 * masked copies (image-major, like dp_apply_fwd's output), w (K,3,7,7); slabs (nslab,B,3,2Ho,2Wo) with
 * nslab = dp_apply_bwd_nslab(B, S, 4*Ho*Wo), to be reduced with dp_sum_slabs (nslab == 1: slabs IS the result):
 *   slabs[z,b,c,h,w] = (1/std_c) * sum_{s in slab z} keep(idx[b,s])[h,w] * stem_dgrad(dy[b*S+s])[c,h,w]
 * The per-sample (B*S,3,H,W) gradient never reaches HBM (602 112 B written + read per sample @224 saved).
 * Same arithmetic and summation order as the two separate entry points: bit-identical results. */
int dp_stem_dgrad_reduce(const float *dy, const float *w, const int32_t *table, int R, const int32_t *idx,
                         const int32_t *idx2, int idx_bstride, int B, int S, int K, int Ho, int Wo,
                         const dp_norm_t *norm, float *slabs, dp_stream_t stream);

/* ---- a-8  stride-2 pixel subsampling around the backbone's strided 1x1 (downsample) convolutions ----
 * (timm 0.6.7 PreActBottleneck.downsample = StdConv2d(1x1, stride 2), first block of stages 1-3; executed at
 * reference attack.py:222, 247.)  A 1x1 convolution with stride 2 reads only the even pixels: it IS the
 * stride-1 1x1 convolution (a GEMM on NCHW memory, dorpatch_amd/conv1x1.py) of the subsampled tensor.
 * Libraries instead transpose the full-resolution activation to NHWC and back, zero-fill the full-resolution
 * gradient and scatter into it, and autograd then adds that to the sibling branch's gradient.
 *   dp_subsample2     y[nc,h,w] = x[nc,2h,2w]                 x (NC,H,W) -> y (NC,H/2,W/2); H, W even
 *   dp_subsample2_add g[nc,2h,2w] += dy[nc,h,w]  (in place)   the adjoint, accumulated into the gradient the
 *                     sibling stride-1 branch already produced (replaces zero-fill + scatter + autograd's add)
 * Traffic: reads the even rows of the large tensor (4 B per output element x 2, half of each line unused),
 * dp_subsample2_add also writes them back. */
int dp_subsample2(const float *x, int64_t NC, int H, int W, float *y, dp_stream_t stream);
int dp_subsample2_add(const float *dy, int64_t NC, int H, int W, float *g, dp_stream_t stream);

/* ---- measurement support (bench.py "roofline"): kernel-precise duration of dp_apply_fwd ----
 * dp_apply_fwd_timed is dp_apply_fwd launched through hipExtLaunchKernelGGL, which stamps `start` /
 * `stop` with the kernel's own begin / end on its stream (the duration rocprofv3 --kernel-trace
 * reports), instead of the completion time of marker packets around it.  dp_event_elapsed_ms
 * blocks until `stop` has happened. */
int dp_event_create(dp_event_t *ev);
int dp_event_destroy(dp_event_t ev);
int dp_event_elapsed_ms(dp_event_t start, dp_event_t stop, float *ms);
int dp_apply_fwd_timed(const float *adv_x, const int32_t *table, int R, const int32_t *idx,
                       const int32_t *idx2, int idx_bstride, int B, int S, int H, int W,
                       const dp_norm_t *norm, float *out, dp_stream_t stream, dp_event_t start,
                       dp_event_t stop);
/* the same for the placement extension's forward (bench.py --placement) */
int dp_apply_affine_fwd_timed(const float *x, const float *delta, const float *theta, const int32_t *table, int R,
                              const int32_t *idx, const int32_t *idx2, int idx_bstride, int B, int S, int H, int W,
                              const dp_norm_t *norm, float *out, dp_stream_t stream, dp_event_t start,
                              dp_event_t stop);

#ifdef __cplusplus
}
#endif
#endif /* DORPATCH_HIP_H */
