/*
 * gqhip.h -- C ABI of libgqhip.so, the MI355X (gfx950) implementation of the
 * reference's Gaussian-quantiser inference hot path.
 *
 * Plain pointers and sizes only: every pointer is a DEVICE pointer (hipMalloc /
 * torch CUDA tensor .data_ptr()) unless its name ends in _host; `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  All entry points
 * are asynchronous on `stream`, never synchronise, never allocate, and return
 * a gqhip_status (0 = success) instead of throwing.  They are re-entrant and
 * hold no global state except the optional profiling recorder.
 *
 * Reference interface each entry point replaces (paths under /root/reference):
 *
 *   gq_scores_f32    gq_cuda_extension/gq_cuda/csrc/cuda/gq_cuda.cu:77-118
 *                    (host launcher `gq_cuda`, schema csrc/gq_cuda.cpp:29-31,
 *                    python gq_cuda/ops.py:7-8) -- fills out[b, n].
 *   gq_argmax_f32    pit/quantization/gaussian.py:124-150 -- the whole
 *                    "score matrix -> argmax -> index_select" block (K1+K2+K3 /
 *                    K4 of SURVEY.md 2.3), fused; indices are those of the
 *                    reference's torch CPU backend (gaussian.py:134-150).
 *   gq_quantize_z_f32 pit/quantization/gaussian.py:61-81,120-160 (GQ1) and
 *                    :273-331 (GQ2.quant_vq) -- also folds the chunk/clamp/exp
 *                    and the group permutes into the kernels.
 *   gq_dequant_f32   pit/quantization/gaussian.py:162-178, :347-362.
 *   gq_quantize_z_gauss_f32  pit/quantization/gaussian.py:211-271,273-345 (GaussianQuantRegularizer2.forward, eval).
 *   vq_argmin_f32    pit/quantization/vq.py:58-73.
 *   vq_quantize_z_f32 pit/quantization/vq.py:39-96 (VQQuantizer.forward, eval).
 *   lfq_pack_f32     pit/quantization/lfq.py:147-158.
 *   lfq_unpack_f32   pit/quantization/lfq.py:210-228 (also BSQ: pit/quantization/bsq.py:85-156).
 *   fsq_quantize_f32 / fsq_dequant_f32  pit/quantization/fsq.py:29-89.
 *   gn_silu_f32      pit/modules/unet.py:49-57 (Normalize + nonlinearity pairs).
 *   gq_index_histogram eval.py:127,137-141,152-154 (stubbed-out histogram).
 */
#ifndef GQHIP_H_
#define GQHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GQHIP_ABI_VERSION 8   /* 2: filter selection, debug plan, NHWC upsample; 3: Winograd / sub-pixel conv transforms, gn_stats, add_bias_stats;
                               * 4: four-launch fused arg-max (no caller-cached max|cb|; noise / zhat_noquant in gq_quantize_z_f32), profile_reserve;
                               * 5: GroupNorm statistics as order-independent fixed-point records (gqhip_gnstat_t), conv3x3_f32 (fp32 matrix
                               *    cores: conv_in / conv_out of the encoder and decoder), gqhip_debug_barrier / gqhip_debug_tail;
                               * 6: three-launch fused arg-max -- undecided rows are finished inside the re-rank, the tail kernel with its grid
                               *    barriers is gone and gqhip_debug_barrier / gqhip_debug_tail with it; gq_step_record_f32, conv3x3_cin_small_f32;
                               * 7: codebook cache (gqhip_cb_cache_bytes; cb_cache arguments of gq_argmax_f32 / gq_quantize_z_f32 / vq_argmin_f32):
                               *    dim 4 runs a pruned exact search over a cached spatial index of the codebook (csrc/gq_grid.h), dims 8 / 16 /
                               *    32 keep the filter's fp16 codebook image there;
                               * 8: module-level eval forwards of GaussianQuantRegularizer2 (gq_quantize_z_gauss_f32: the Gaussian branch's
                               *    statistics and lambda state on the device) and VQQuantizer (vq_quantize_z_f32: layouts, straight-through
                               *    value and codebook loss inside the launches) */

/* GroupNorm statistics of one (image, group): GQHIP_GNSTAT_WORDS int64 words = {sum: 3 limbs, sum of squares: 3 limbs, poison,
 * unused}; value = q0 2^-56 + q1 2^-16 + q2 2^24.  Every kernel that leaves statistics behind adds its threads' fp32 partial
 * sums to these records with INTEGER atomics (exact, associative), so the statistics -- and everything normalised with them --
 * are bit-identical from run to run whatever order the adds land in (csrc/gq_stats.h).  A `stats` argument below is an
 * array of B * groups such records (64 bytes each), zeroed by the producing call. */
typedef int64_t gqhip_gnstat_t;
#define GQHIP_GNSTAT_WORDS 8

typedef enum gqhip_status {
  GQHIP_OK = 0,
  GQHIP_ERR_INVALID_ARG = 1,   /* NULL pointer, negative size, unsupported dim */
  GQHIP_ERR_WORKSPACE = 2,     /* workspace too small / NULL                  */
  GQHIP_ERR_LAUNCH = 3,        /* hipGetLastError() != hipSuccess after launch */
  GQHIP_ERR_NO_DEVICE = 4
} gqhip_status;

int gqhip_abi_version(void);
const char *gqhip_status_string(int status);
/* last hipError_t observed by a failing call on this thread (0 if none). */
int gqhip_last_hip_error(void);

/* Filter kernel of the fused arg-max (gq_argmax_f32 / gq_quantize_z_f32 / vq_argmin_f32).  AUTO: the fp16 main-product
 * filter (ONE fp16 MFMA product per fp32 product, no correction terms; its rounding error is covered by a bound the re-rank
 * derives from the row's own data) at every MFMA dim (4/8/16/32), Gaussian score and VQ; rows whose candidate records are incomplete
 * are finished inside the re-rank kernel by a complete scan of the record sets in question (csrc/gq_rerank.h).  FP32: always the fp32 MFMA filter.  BF16: the split-bf16 filter (three
 * bf16 products per fp32 product).  MIXED: round 2's fp16 + fp8 filter (one fp16 product + block-scaled fp8 corrections) at
 * dim 16 with the Gaussian score, split-bf16 elsewhere.  All feed the same exact re-rank: the indices are identical.
 * Process-wide; initial value from the environment (GQHIP_FILTER=fp32|bf16|mixed).  The workspace size depends on it:
 * query gqhip_workspace_bytes after changing it.  (No reference counterpart.) */
#define GQHIP_FILTER_AUTO 0
#define GQHIP_FILTER_FP32 1
#define GQHIP_FILTER_BF16 2
#define GQHIP_FILTER_MIXED 3
int gqhip_set_filter(int kind);
int gqhip_get_filter(void);

/* Diagnostics: launch plan and workspace layout of the fused arg-max for a shape.  out8 = { byte offset of the
 * candidate records, code splits, tiles per candidate group, tiles per split, filter kernel the plan selects (0 fp32 MFMA, 1 split-bf16, 2 fp16 + fp8, 3 fp16 main product: the default at dims 8 / 16 / 32), coefficient of the
 * filter error bound (E_f = coeff * 2^-24 * T), row tiles per wave, waves per filter block }.
 * The records are 16 bytes per (row, record set): { fp32 m1; fp16 gaps m1 - m2..m4, rounded towards zero; three 16-bit
 * half-group ids relative to the set's split } (csrc/gq_common.h:Rec); "code splits" counts record sets. */
int gqhip_debug_plan(int64_t rows, int64_t n, int64_t dim, int64_t *out8);


/* ---- workspace ------------------------------------------------------------
 * Bytes of scratch gq_argmax_f32 / gq_quantize_z_f32 / vq_argmin_f32 need for
 * `rows` rows against `n` codes of width `dim`.  The caller allocates once
 * (device memory, 256-B aligned) and reuses it; contents are don't-care: every call
 * rebuilds what it keeps there (row operands, bound sums, candidate records, max|cb|,
 * counters) from its arguments.  What is derived from the CODEBOOK and worth keeping
 * across calls lives in the separate codebook cache below, which the library validates
 * against the codebook's content on every call. */
int64_t gqhip_workspace_bytes(int64_t rows, int64_t n, int64_t dim);

/* ---- codebook cache ---------------------------------------------------------
 * Bytes of PERSISTENT device memory (256-B aligned) in which the fused arg-max keeps what it derives from a codebook of `n` codes
 * of width `dim` across calls; 0 when this shape keeps nothing (then pass NULL / 0).  Today: dim 4 with 2^14 <= n <= 2^20
 * -- the codes sorted into 4096 sub-leaves under a tree of bounding boxes (16 -> 256 -> 1024 -> 4096), which lets dim 4 run a pruned
 * exact search (~8 sub-leaves of ~16 codes per row instead of all n codes; csrc/gq_grid.h) in place of filter + re-rank --; dims
 * 8 / 16 / 32: the fp16 operand image of the codebook that the MFMA filter reads (each 1/256 slice stamped with its content hash).
 * Contract: the caller owns the buffer, hands the SAME buffer to every call that uses the same codebook, and never writes it;
 * its initial contents are don't-care.  The library validates it on EVERY call -- the first launch hashes the codebook it is
 * given (it reads it anyway) and compares with the hashes the cache was stamped with -- and rebuilds it in-stream (one extra
 * one-block kernel that otherwise exits at once) when they differ: a codebook edited in place by any route, a different
 * codebook, a fresh buffer or one overwritten as a whole all cost one rebuild and never a wrong index.  What the validation does NOT
 * cover is a write into the BODY of a cache whose 4-KiB header (the stamps) is left intact -- the "never writes it" above is the
 * caller's half of the contract; the dim-4 search still bounds every offset and index it reads from the body by the codebook size,
 * so such a write can cost a wrong index but never an out-of-range access (csrc/gq_grid.h); the fp16 image of dims 8 / 16 / 32 is only
 * ever multiplied -- no address is derived from it.  One cache serves one stream at a time.
 * Without a cache (NULL) every shape runs the filter + re-rank path.  (No reference counterpart: the reference recomputes
 * everything per call, pit/quantization/gaussian.py:136-150.) */
int64_t gqhip_cb_cache_bytes(int64_t n, int64_t dim);
/* 1 when a call with this codebook shape and a cache runs the pruned search (today: dim 4, 2^14 <= n <= 2^20, filter selection
 * AUTO), 0 when it runs filter + re-rank (dims 8 / 16 / 32 then keep the filter's fp16 codebook image in the cache). */
int gqhip_grid_search_applies(int64_t n, int64_t dim);
/* Synchronous (a 4-KiB device-to-host copy; never needed for correctness): what the index builder found when it sorted this codebook.
 * 1: a DEGENERATE book for the search -- some sub-leaf holds more than 255 codes (most of the book in one cell: clustered or collapsed
 * codebooks), so every row that lists it is handed to gq_grid_finish_kernel, a block per row: a multi-millisecond call where the dense
 * filter + re-rank takes ~100 us; a host that sees 1 should stop passing the cache for this codebook (NULL / 0: the dense path).
 * 0: fine.  -1: no current index in `cb_cache` (never built, stale, another shape) or a shape without a search.  pit_hip's Workspace
 * asks once, at the call after the one that built the index.  (No reference counterpart.) */
int gqhip_cb_cache_degenerate(const void *cb_cache, int64_t n, int64_t dim);

/* ---- compat op: the reference's native boundary ---------------------------
 * out[r, j] = sum_i -((cb[j,i]-mu[r,i])/sd[r,i])^2 + cb[j,i]^2 * beta
 * mu, sd [rows, dim]; cb [n, dim]; out [rows, n]; all fp32 contiguous.
 * dims 4 / 8 / 16 / 32: the expanded form [beta - 1/sd^2 | 2 mu/sd^2] x [cb^2 | cb] + const on the matrix cores (fp32 MFMA at
 * dims 4 / 8; three fp16 products of two-term splits with fp32 accumulation at dims 16 / 32 -- error vs an fp64 evaluation
 * <= ~4e-7 of sum_i |terms| per element in either form, the level of the per-pair formula in fp32); other dims, n < 32 or a
 * NaN beta: the per-pair formula.  Environment: GQHIP_SCORES=f32 (fp32 MFMA at dims 16 / 32 too), =direct (per-pair always). */
int gq_scores_f32(const float *mu, const float *sd, const float *cb, float *out,
                  int64_t dim, int64_t rows, int64_t n, double beta,
                  void *stream);

/* ---- fused score + argmax + gather ----------------------------------------
 * idx[r]  = argmax_j of the reference torch-backend score (first max wins,
 *           NaN counts as max), bit-identical to the CPU reference given the
 *           same (mu, sd, logsd).
 * zhat[r] = cb[idx[r]]  (optional, may be NULL).
 * logsd_or_null: log(sd) as the caller computed it; NULL -> the kernel uses
 *           float(log(double(sd))) (correctly rounded).
 * dim: any 1..64 (4, 8, 16, 32 run on the MFMA filter; others exhaustive).
 * Dim 4 with a codebook cache: prep -> index check -> pruned exact search -> finish of the rows it left undecided
 * (csrc/gq_grid.h; four launches, the second and fourth exit at once in the common case).
 * Three launches on `stream` for the other MFMA dims: prep (operand images, bound sums,
 * max|cb|) -> filter -> exact re-rank (rows the filter leaves undecided are
 * finished inside it by a block-wide scan of their record sets). */
int gq_argmax_f32(const float *mu, const float *sd, const float *logsd_or_null,
                  const float *cb, int64_t *idx, float *zhat_or_null,
                  int64_t dim, int64_t rows, int64_t n, double beta,
                  void *workspace, int64_t workspace_bytes,
                  void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream);

/* ---- module-level fused quantiser -----------------------------------------
 * z is the encoder output holding [mu | logvar] along its channel axis.
 * layout: GQHIP_LAYOUT_BCHW  z [B, 2c, L] (L = h*w)   -> idx [B, K, L], zhat [B, c, L]
 *         GQHIP_LAYOUT_BLC   z [B, L, 2c]             -> idx [B, L, K], zhat [B, L, c]
 * grouping: GQHIP_GROUP_STRIDED    (GaussianQuantRegularizer : column g of
 *                                   sub-codebook k <- channel g*K + k)
 *           GQHIP_GROUP_CONTIGUOUS (GaussianQuantRegularizer2: channel k*dim + g)
 * K = c / dim.  logvar is clamped to [lv_min, lv_max]; sd = exp(0.5*logvar) and
 * log(sd) are evaluated in fp64 and rounded once (numerics).
 * mu_out/sd_out (optional, [rows, dim], row = (b*L + l)*K + k) receive the
 * permuted operands so callers/tests can replay them through the oracle.
 * noise_or_null / zhat_noquant_or_null (both in the layout of zhat): when given,
 * zhat_noquant = mu + noise * sd (gaussian.py:121; the caller draws `noise` with
 * its own generator, e.g. torch.randn) is written by the same first launch. */
#define GQHIP_LAYOUT_BCHW 0
#define GQHIP_LAYOUT_BLC 1
#define GQHIP_GROUP_STRIDED 0
#define GQHIP_GROUP_CONTIGUOUS 1
int gq_quantize_z_f32(const float *z, const float *noise_or_null, const float *cb,
                      int64_t *idx, float *zhat_or_null, float *zhat_noquant_or_null,
                      float *mu_out_or_null, float *sd_out_or_null, int64_t B,
                      int64_t L, int64_t c, int64_t dim, int64_t n, int layout,
                      int grouping, double lv_min, double lv_max, double beta,
                      void *workspace, int64_t workspace_bytes,
                      void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream);

/* ---- GaussianQuantRegularizer2.forward in eval (pit/quantization/gaussian.py:211-271 quant_gaussian, :273-331 quant_vq,
 * :333-345 forward) as ONE call: gq_quantize_z_f32's launches, the re-rank launch carrying one extra block for the statistics (behind the
 * dim-4 search and at dims without a filter: one extra one-block launch).  On top of gq_quantize_z_f32:
 *   zhat_noquant = mu + noise * sd (required here: it is the Gaussian branch's sample), sd_out_or_null = sd in the layout of zhat
 *   (info["std"]); use_ste != 0: zhat = (zhat_noquant - zhat_noquant) + code, the value of `zhat_g - zhat_g.detach() + zhat_v`
 *   (gaussian.py:337-338) in the reference's fp32 op order (a non-finite zhat_noquant makes it NaN there too), and
 *   zhat_quant_or_null (layout of zhat) then receives the codewords themselves (info["zhat_quant"]); without use_ste zhat holds them;
 *   per row (= per group) kl2 = sum_i 1.4426 * 0.5 * (mu^2 + var - 1 - logvar) (gaussian.py:225-229; the element in the
 *   reference's fp32 op order with var = float(exp(double(logvar))), the row sum in fp64 rounded once), reduced in a fixed order
 *   (bit-reproducible) to
 *   scalars_out (64 bytes of device memory, 8-byte aligned): float[0..3] = { kl_loss, bits-mean, bits-min, bits-max }
 *       (kl_loss = mean(ge kl2 + eq kl2 + le kl2) * lam with lam / lam_min / lam_max as they were BEFORE this call, gaussian.py:233-241);
 *       double[0..2] at byte 32 = { lam, lam_min, lam_max } AFTER this call's update (what info["lam"], ["lam-min"], ["lam-max"] report).
 *   lam_state (device, double[3] = { lam, lam_min, lam_max }, in / out): the adaptive lambda state machine of gaussian.py:243-257
 *       advanced on the device in fp64 (the same IEEE operations as the reference's Python floats), so the forward reads nothing
 *       back -- the reference pays three host syncs per forward for it.  log2n = int(log2(n_samples)); thresholds are compared as
 *       torch does (the scalar cast to fp32); lam_max_decreases: 1 = gaussian.py:109-112 (GQ1's train branch), 0 = GQ2, whose
 *       decrease at gaussian.py:251 is an expression without effect; clamps: lam_max to [1, lam_hi], lam_min to [lam_lo, 1].
 * Workspace / codebook cache / stream: as gq_quantize_z_f32. */
int gq_quantize_z_gauss_f32(const float *z, const float *noise, const float *cb, int64_t *idx, float *zhat,
                            float *zhat_quant_or_null, float *zhat_noquant, float *sd_out_or_null, void *scalars_out, double *lam_state, int64_t B,
                            int64_t L, int64_t c, int64_t dim, int64_t n, int layout, int grouping, double lv_min,
                            double lv_max, double beta, int use_ste, double log2n, double tolerance, double lam_factor,
                            double lam_lo, double lam_hi, int lam_max_decreases, void *workspace, int64_t workspace_bytes,
                            void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream);

/* zhat from indices (same layouts as above). */
int gq_dequant_f32(const int64_t *idx, const float *cb, float *zhat, int64_t B,
                   int64_t L, int64_t K, int64_t dim, int64_t n, int layout,
                   int grouping, void *stream);

/* ---- VQ: argmin_j |z_r - e_j|^2 (fp64 arbiter, first min wins) ------------- */
int vq_argmin_f32(const float *z, const float *emb, int64_t *idx,
                  float *zq_or_null, int64_t dim, int64_t rows, int64_t n,
                  void *workspace, int64_t workspace_bytes,
                  void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream);

/* ---- VQQuantizer.forward in eval (pit/quantization/vq.py:39-96) as ONE call: the permutes, the per-sub-codebook slices, the
 * embedding lookup, the straight-through value and the two MSE means are inside vq_argmin_f32's launches + one small launch.
 *   z [B, c, L] (GQHIP_LAYOUT_BCHW) or [B, L, c] (GQHIP_LAYOUT_BLC), c = dim * K; sub-codebook k, column d <- channel d * K + k
 *   (vq.py:53), every sub-codebook against the same `emb` [n, dim];
 *   idx [B, K, L] / [B, L, K];  zq (layout of z) = z + (e - z), e = emb[idx], in the reference's fp32 op order (vq.py:89);
 *   loss2_or_null (device float[2]): [0] = codebook_loss = mean + beta * mean (legacy != 0) | beta * mean + mean (vq.py:78-86) in
 *   fp32, [1] = mean = sum((e - z)^2) / (rows * dim): squared differences formed in fp32, summed in fp64 in a fixed order
 *   (bit-reproducible; torch's fp32 mean agrees to rounding).  Arg-min: vq_argmin_f32's (fp64 arbiter, first minimum wins). */
int vq_quantize_z_f32(const float *z, const float *emb, int64_t *idx, float *zq, float *loss2_or_null, int64_t B, int64_t L,
                      int64_t c, int64_t dim, int64_t n, int layout, double beta, int legacy, void *workspace,
                      int64_t workspace_bytes, void *cb_cache_or_null, int64_t cb_cache_bytes, void *stream);

/* ---- LFQ: sign quantisation + big-endian bit pack -------------------------- */
int lfq_pack_f32(const float *x, int64_t *idx, float *q_or_null, int64_t rows,
                 int64_t nbits, void *stream);
int lfq_unpack_f32(const int64_t *idx, float *q, int64_t rows, int64_t nbits,
                   void *stream);

/* ---- FSQ: tanh-bounded rounding + mixed-radix pack (pit/quantization/fsq.py:29-89) ----
 * z, zhat [rows, nlev] fp32; idx [rows] int32 (the reference's dtype); levels_host: nlev (<= 16)
 * positive ints on the HOST.  tanh/atanh are evaluated in fp64 and rounded once. */
int fsq_quantize_f32(const float *z, const int32_t *levels_host, int64_t nlev, float *zhat,
                     int32_t *idx, int64_t rows, void *stream);
int fsq_dequant_f32(const int32_t *idx, const int32_t *levels_host, int64_t nlev, float *zhat,
                    int64_t rows, void *stream);

/* ---- fused GroupNorm (+ SiLU) for the conv stack -----------------------------------
 * y = act( (x - mean_g) * rstd_g * gamma_c + beta_c ), act = SiLU when apply_silu != 0.
 * Replaces the GroupNorm -> swish pairs of pit/modules/unet.py:54-57,49-51,137-153,432-435
 * (three PyTorch kernels, five HBM passes) by a stats pass + one apply pass (three passes).
 * x, y [B, C, HW] fp32 contiguous (NCHW), groups | C; stats_ws: B*groups statistics records
 * (GQHIP_GNSTAT_WORDS int64 each) of caller scratch.  x and y may alias.  pre_bias_or_null [C]: a per-channel bias still pending
 * on x (the producing conv was run without its bias) -- normalises x + pre_bias[c] without a
 * separate bias pass. */
#define GQHIP_LAYOUT_NCHW 0   /* x[b][c][hw] */
#define GQHIP_LAYOUT_NHWC 1   /* x[b][hw][c] (torch channels_last): needs (C/groups) % 4 == 0, (C/4) | 256 */
int gn_silu_f32(const float *x, const float *gamma, const float *beta,
                const float *pre_bias_or_null, float *y, int64_t B, int64_t C, int64_t HW,
                int64_t groups, double eps, int apply_silu, int layout, gqhip_gnstat_t *stats_ws,
                void *stream);
/* y = a + b (+ bias[c]): residual add with the pending conv biases folded in (unet.py:153). */
int add_bias_f32(const float *a, const float *b, const float *bias_or_null, float *y, int64_t B,
                 int64_t C, int64_t HW, int layout, void *stream);

/* NHWC only.  y = a + b (+ bias[c]) AND the GroupNorm statistics of y: record b*groups+g of stats_out = sum and sum of
 * squares over the group (gqhip_gnstat_t records; zeroed here).  The residual add of a ResnetBlock / AttnBlock (unet.py:160, :205) is
 * always followed by a GroupNorm over the same tensor (unet.py:140, :190, :581): gn_apply_f32 then normalises from
 * these statistics without its own statistics pass.  Needs (C/groups) % 4 == 0, 256 % (C/4) == 0, groups <= 64. */
int add_bias_stats_f32(const float *a, const float *b, const float *bias_or_null, float *y, int64_t B,
                       int64_t C, int64_t HW, int64_t groups, gqhip_gnstat_t *stats_out, void *stream);

/* NHWC only.  The apply pass of gn_silu_f32 with statistics computed earlier (by add_bias_stats_f32). */
int gn_apply_f32(const float *x, const float *gamma, const float *beta, float *y, int64_t B, int64_t C,
                 int64_t HW, int64_t groups, double eps, int apply_silu, const gqhip_gnstat_t *stats, void *stream);

/* Content checksums of `count` device tensors in ONE launch: table_dev = count entries {const void *ptr; int64_t words}
 * (32-bit words; 16-byte aligned pointers), sums_dev[t] = a 64-bit position-salted hash sum of tensor t (zeroed here; integer
 * atomics: independent of the order of the adds).  The conv-stack modules use it to notice parameter writes that bump no
 * version counter (`param.data`): their weight-derived caches are rebuilt when a sum differs (pit_hip/modules/unet.py). */
int gqhip_checksum_tensors(const void *table_dev, int64_t count, uint64_t *sums_dev, void *stream);

/* Every entry point that FILLS GroupNorm statistics records zeroes them first (one small memset launch each).  A caller that hands
 * out records which are already zero -- e.g. slices of one arena cleared by a single fill per forward, as pit_hip/modules/unet.py
 * does -- says so with gqhip_stats_prezeroed(1) around those calls; (0) restores the default.  Thread-local.  (ABI 6; no reference
 * counterpart.) */
int gqhip_stats_prezeroed(int on);

/* The encoder's conv_in (pit/modules/unet.py:411-413: 3 -> ch channels): a 3x3 / stride 1 / pad 1 convolution of a channels_last
 * image with Cin <= 4 input channels into Cout = 128 channels as fp32 FMAs in a FIXED order (tap-major, then input channel), + bias,
 * + the statistics of the result for the GroupNorm that follows (stats_out: B * 32 records, zeroed here; groups_out must be 32).
 * wk [9 Cin, 128]: wk[(tap * Cin + ci) * 128 + co] = weight[co][ci][tap / 3][tap % 3].  H % 8 == 0, W % 32 == 0.  Bit-reproducible;
 * replaces the last library convolution of the bench shapes (ABI 6). */
int conv3x3_cin_small_f32(const float *x, const float *wk, const float *bias_or_null, float *y, int64_t *stats_out_or_null,
                          int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t groups_out, void *stream);

/* 3x3 convolution, stride 1, zero padding 1, NHWC fp32, on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: an exact fp32 FMA
 * chain) with a FIXED summation order, so the result is bit-identical from run to run -- for the narrow ends of the conv
 * stack, where the library's pick is a split-K kernel with floating-point atomics: the encoder's conv_out
 * (pit/modules/unet.py:425-436: norm_out -> swish -> conv 512 -> 2 z_channels, the layer that produces z) and the decoder's
 * conv_in (unet.py:487-489: z_channels -> 512).  stats != NULL: the input is act(GroupNorm(x + pre_bias)) applied while the
 * patch is staged (gamma, beta [Cin]; stats = B * groups_in records; Cin <= 1024); act = SiLU when apply_silu != 0.
 * wk: the weights in operand order [ceil(Cout/32)][9 taps][Cin/8][64 lanes][4]: element (t, tap, g, 32 h + j, m) =
 * w[32 t + j][8 g + 4 h + m][tap / 3][tap % 3], zero for output channels >= Cout.  Any H, W (a block owns 32 pixels of a row); Cout % 4 == 0;
 * Cin % 64 == 0 (K split over the four waves of a block, partial sums added in wave order) or, without GroupNorm,
 * Cin in {8, 16, 32} (output channels split over the waves). */
int conv3x3_f32(const float *x, const float *gamma_or_null, const float *beta_or_null, const float *pre_bias_or_null,
                const gqhip_gnstat_t *stats_or_null, int64_t groups_in, double eps, int apply_silu, const float *wk,
                const float *bias_or_null, float *y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, void *stream);

/* Nearest-neighbour x2 upsample of an NHWC fp32 tensor: x [B, H, W, C] -> y [B, 2H, 2W, C], C % 4 == 0
 * (pit/modules/unet.py:69-73, `F.interpolate(scale_factor=2.0, mode="nearest")` in channels_last). */
int upsample2x_nhwc_f32(const float *x, float *y, int64_t B, int64_t H, int64_t W, int64_t C,
                        void *stream);

/* Winograd F(2x2, 3x3) data transforms for a stride-1, padding-1 3x3 convolution of an NHWC tensor (the decoder's wide
 * ResnetBlock convolutions, pit/modules/unet.py:142, :149): V [16, tiles, C] = B^T d B of every 4x4 input tile
 * (tiles = B * H/2 * W/2, H and W even); after the 16 GEMMs M[k] = V[k] x U[k] (U = G g G^T, [16, Cin, Cout]) the
 * output transform writes y [B, H, W, Cout] = A^T M A.  16 instead of 36 multiplies per 2x2 outputs. */
int wino_in_nhwc_f32(const float *x, float *V, int64_t B, int64_t H, int64_t W, int64_t C, void *stream);
/* `mscale` (all three output transforms): y = mscale * (A^T M A) (+ bias, + res): 1 for fp32 operands, the inverse of
 * the operands' power-of-two scales behind wino_in_nhwc_f16x3. */
int wino_out_nhwc_f32(const float *M, float *y, int64_t B, int64_t H, int64_t W, int64_t C, float mscale, void *stream);

/* The input transforms (tile = 2: F(2x2,3x3), 4: F(4x4,3x3)) writing the operand of ONE fp16 GEMM with fp32 accumulation
 * whose K axis carries the three products of two-term fp16 splits: V3 [16|36, tiles, 3C] fp16 = [h | h | l] of
 * (B^T d B) * scale, to be multiplied by U3 [16|36, 3C, Cout] fp16 = [U_h ; U_l ; U_h]: the products h U_h + h U_l + l U_h
 * carry 22-bit operands (error 2.5-2.9e-7 of sum |a||b| measured -- the level of hipBLASLt's own fp32 GEMM, which is a
 * split-bf16 emulation on gfx950 -- at 1.2-2.5x its speed).  `scale`: a power of two with |V| * scale < 65504 (the
 * caller derives it from the GroupNorm that feeds the convolution: |SiLU(GN(x))| <= sqrt(n - 1) max|gamma| + max|beta|). */
int wino_in_nhwc_f16x3(const float *x, void *V3, int64_t B, int64_t H, int64_t W, int64_t C, int tile, float scale,
                       void *stream);
/* ... as V2 [16|36, tiles, 2C] fp16 = [h | l] only (4 instead of 6 bytes per element): the operand of wino_gemm_f16x2, which
 * forms the three products itself. */
int wino_in_nhwc_f16x2(const float *x, void *V2, int64_t B, int64_t H, int64_t W, int64_t C, int tile, float scale,
                       void *stream);
/* The wider levels (Cin % 32 == 0, Cout % 128 == 0, tiles % 256 == 0; the SD3-UNet's 256- and 512-channel convolutions,
 * reference pit/modules/unet.py:142, :149): M [P, tiles, Cout] fp32 = V2 (x) U with Wf [P, Cin/16, Cout/32, 2, 64, 8] fp16 =
 * (U_h, U_l) of U * u_scale in MFMA operand order (lane (c, h) of column tile nt holds k = 16 chunk + 8 h .. + 7 of column
 * 32 nt + c).  Replaces the hipBLASLt GEMM over K' = 3 Cin of V3 = [h | h | l]: same splits, same three products, fp32
 * accumulation; a third less traffic on V. */
int wino_gemm_f16x2(const void *V2, const void *Wf, float *M, int64_t P, int64_t tiles, int64_t Cin, int64_t Cout,
                    void *stream);
/* Direct 3x3 convolution (stride 1, zero padding 1) Cin -> Cout (128 or 256) channels, channels_last, as an implicit
 * GEMM on the fp16 matrix cores with the fp16 x 3 scheme above (two-term splits of both operands, three products, fp32
 * accumulation): the convolutions of the 256 x 256 level and the encoder's 128 x 128 level (reference pit/modules/unet.py:142,
 * :149), where the Winograd routes are HBM-bound on V and M.  Replaces F.conv2d(swish(norm(x)), w, None, 1, 1) + bias +
 * residual (unet.py:140-153).
 *   conv3x3_gn_f16x3     y [B, H, W, Cout] = (SiLU(GroupNorm(x + pre_bias)) (*) w) + bias (+ res) in ONE kernel: x [B, H, W, Cin]
 *                        fp32 is normalised, activated, scaled and split on its way into LDS.  stats_in [B, groups_in, 2] as
 *                        gn_stats_f32 / the statistics outputs of this library; stats_out (optional): GroupNorm statistics of
 *                        y, [B, groups_out] statistics records (4 | Cout / groups_out | 128).  H % 8 == 0, W % 32 == 0, Cin % 32 == 0,
 *                        Cin <= 512.  `scale`: a power of two with |activation| * scale <= 32768; mscale = 1 / (scale * u_scale).
 *                        Wf [Cin/16, 9, Cout/32, 2, 64, 8] fp16: operand-order weights -- chunk, tap ky*3+kx, column tile,
 *                        plane (h, l of w * u_scale), lane (n = 32 tile + lane%32, k-half lane/32), 8 input channels
 *                        16 chunk + 8 (lane/32) + e.
 */
int conv3x3_gn_f16x3(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                     const gqhip_gnstat_t *stats_in, int64_t groups_in, double eps, int apply_silu, float scale, const void *Wf,
                     const float *bias_or_null, const float *res_or_null, float *y, gqhip_gnstat_t *stats_out_or_null, int64_t B,
                     int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t groups_out, float mscale, void *stream);
/* 1x1 convolution Cin -> Cout (128, 256, 512 or 1536 = the attention block's q | k | v) of a channels_last tensor = a GEMM over its B * HW pixels, fp16 x 3 as above
 * with the split of x done inside the kernel: the ResnetBlocks' nin_shortcut (reference pit/modules/unet.py:151-152) and the
 * attention block's proj_out (:203).  x [B * HW, Cin] fp32 (NOT normalised; + pre_bias[c], a bias still pending on it); the
 * power-of-two scale of x + pre_bias comes from
 * scales_dev = {scale, 1 / (scale * u_scale)} in device memory (f16_scales_from_gn_stats) or, when that is NULL, from the
 * `scale` / `mscale` arguments.  Wf [Cin/16, 1, Cout/32, 2, 64, 8] as for conv3x3_gn_f16x3; y = x W * mscale + bias (+ res);
 * stats_out optional.  HW % 256 == 0, Cin % 32 == 0. */
int conv1x1_f16x3(const float *x, const float *pre_bias_or_null, const void *Wf, const float *scales_dev_or_null, float scale,
                  float mscale, const float *bias_or_null, const float *res_or_null, float *y, gqhip_gnstat_t *stats_out_or_null, int64_t B, int64_t HW,
                  int64_t Cin, int64_t Cout, int64_t groups_out, void *stream);
/* 3x3 convolution with stride 2 over x padded by one zero row / column at the bottom / right -- the reference's Downsample
 * (pit/modules/unet.py:76-97: F.pad(x, (0,1,0,1)) + conv stride 2) -- Cin -> Cout (128, 256 or 512), fp16 x 3 on the four
 * phase images of x (nine k-steps per 16 input channels).  x [B, Hin, Win, Cin] fp32 (not normalised; scale as for
 * conv1x1_f16x3), y [B, Hin/2, Win/2, Cout] = conv * mscale + bias; stats_out optional.  Hin % 16 == 0, Win % 64 == 0,
 * Cin % 16 == 0.  Wf [9 Cin/16, Cout/32, 2, 64, 8]: operand-order weights, k-steps in the order (phase (a, b) = (0,0), (0,1),
 * (1,0), (1,1); chunk; tap (dy, dx) of the phase, ky = 2 dy + a, kx = 2 dx + b). */
int conv3x3s2_f16x3(const float *x, const void *Wf, const float *scales_dev_or_null, float scale, float mscale,
                    const float *bias_or_null, float *y, gqhip_gnstat_t *stats_out_or_null, int64_t B, int64_t Hin, int64_t Win,
                    int64_t Cin, int64_t Cout, int64_t groups_out, void *stream);
/* The reference's Upsample (pit/modules/unet.py:60-73: nearest x2, then conv 3x3) Cin -> Cout (128, 256 or 512) as its sub-pixel
 * form computed directly: output phase (a, b) = a 2x2 convolution of the low-resolution input with the phase weights (sums of
 * the 3x3 taps landing on the same source pixel), fp16 x 3, one kernel: no patch matrix, no pixel-shuffle pass.
 * x [B, H, W, Cin] fp32 (not normalised; scale as for conv1x1_f16x3), y [B, 2H, 2W, Cout] = conv * mscale + bias; stats_out
 * optional (GroupNorm statistics of y).  H % 8 == 0, W % 32 == 0, Cin % 16 == 0.  Wf [4, 4 Cin/16, Cout/32, 2, 64, 8]:
 * operand-order phase weights, [phase 2a + b][chunk][tap 2u + v], tap (u, v) reading x[i - 1 + a + u][j - 1 + b + v]. */
int upconv2x_f16x3(const float *x, const void *Wf, const float *scales_dev_or_null, float scale, float mscale,
                   const float *bias_or_null, float *y, gqhip_gnstat_t *stats_out_or_null, int64_t B, int64_t H, int64_t W,
                   int64_t Cin, int64_t Cout, int64_t groups_out, void *stream);
/* 3x3 convolution (stride 1, zero padding 1) into 1..4 channels with GroupNorm (+ SiLU) of the input fused in, fp32 FMAs:
 * the decoder's conv_out(swish(norm_out(h))) (reference pit/modules/unet.py:585-587).  x [B, H, W, Cin] fp32, w_ohwi
 * [Cout, 3, 3, Cin] fp32, y [B, H, W, Cout].  H % 16 == 0, W % 16 == 0, Cin % 32 == 0, Cin <= 512. */
int conv3x3_gn_small_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                         const gqhip_gnstat_t *stats_in, int64_t groups_in, double eps, int apply_silu, const float *w_ohwi,
                         const float *bias_or_null, float *y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                         void *stream);
/* ... with the producer fused in (as wino_in_gn_nhwc_f32 / wino4_in_gn_nhwc_f32): the convolution's input is
 * SiLU(GroupNorm(x + pre_bias)), never written. */
int wino_in_gn_nhwc_f16x3(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                          const gqhip_gnstat_t *stats, void *V3, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                          double eps, int apply_silu, int tile, float scale, void *stream);
/* ... writing the [h | l] operand of wino_gemm_f16x2. */
int wino_in_gn_nhwc_f16x2(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                          const gqhip_gnstat_t *stats, void *V2, int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups,
                          double eps, int apply_silu, int tile, float scale, void *stream);

/* Winograd F(4x4, 3x3): V [36, tiles, C] of the 6x6 input tiles (tiles = B * H/4 * W/4, H and W multiples of 4) and
 * y [B, H, W, Cout] from M [36, tiles, Cout]; U = G g G^T is [36, Cin, Cout].  36 multiplies per 16 outputs and 2.25x
 * (instead of 4x) the activation in V / M, at ~10x the rounding error of F(2x2,3x3): the decoder's convolutions. */
int wino4_in_nhwc_f32(const float *x, float *V, int64_t B, int64_t H, int64_t W, int64_t C, void *stream);
int wino4_out_nhwc_f32(const float *M, float *y, int64_t B, int64_t H, int64_t W, int64_t C, float mscale, void *stream);

/* Output transform (tile = 2: F(2x2,3x3), M [16, tiles, C]; tile = 4: F(4x4,3x3), M [36, tiles, C]) with the ResnetBlock's
 * tail fused in (pit/modules/unet.py:149-153): y = A^T M A + bias[c] (+ res, may be NULL), plus the GroupNorm statistics of y
 * (stats_out as add_bias_stats_f32) for the block that follows.  Needs (C/groups) % 4 == 0, 256 % (C/4) == 0. */
int wino_out_res_nhwc_f32(const float *M, const float *res, const float *bias_or_null, float *y, gqhip_gnstat_t *stats_out,
                          int64_t B, int64_t H, int64_t W, int64_t C, int64_t groups, int tile, float mscale,
                          void *stream);

/* NHWC only.  GroupNorm statistics alone (the first pass of gn_silu_f32): stats_out[2*(b*groups+g)] = sum, [+1] = sum of
 * squares of x (+ pre_bias[c]) over the group (gqhip_gnstat_t records), zeroed here. */
int gn_stats_f32(const float *x, const float *pre_bias_or_null, int64_t B, int64_t C, int64_t HW, int64_t groups,
                 gqhip_gnstat_t *stats_out, void *stream);

/* wino_in_nhwc_f32 with the producer fused in: the convolution's input is SiLU(GroupNorm(x + pre_bias)) (the
 * `Normalize` -> `nonlinearity` -> conv chains of pit/modules/unet.py:140-142, :146-149); the normalised tensor is
 * never written.  `stats` as produced by gn_stats_f32 / add_bias_stats_f32. */
int wino_in_gn_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                        const gqhip_gnstat_t *stats, float *V, int64_t B, int64_t H, int64_t W, int64_t C,
                        int64_t groups, double eps, int apply_silu, void *stream);
/* the same for F(4x4,3x3) (V [36, tiles, C], H and W multiples of 4) */
int wino4_in_gn_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *pre_bias_or_null,
                         const gqhip_gnstat_t *stats, float *V, int64_t B, int64_t H, int64_t W, int64_t C,
                         int64_t groups, double eps, int apply_silu, void *stream);

/* Single-head attention softmax(q k^T C^-1/2) v (reference pit/modules/unet.py:185-206: F.scaled_dot_product_attention on
 * [B, 1, L, C]) as two fp16 GEMMs with fp32 accumulation over K axes of two-term fp16 splits, fp32 results:
 *   attn_split_qkv_f16x3   qkv [B, L, 3C] fp32 (q | k | v per token) -> Q3 [B, L, 3C] = [q_h | q_h | q_l] of q * sq,
 *                          K3 [B, L, 3C] = [k_h | k_l | k_h] of k * sq, V3 [B, 3L, C] = [v_h ; v_l ; v_h] of v * sv
 *                          (sq, sv: powers of two keeping the operands inside fp16's range; C % 4 == 0);
 *   attn_softmax_split_f16x3  S' [rows, L] fp32 (= Q3 K3^T = sq^2 q k^T) -> P3 [rows, 3L] = [p_h | p_h | p_l] of
 *                          softmax(S' * factor) * 2^14, factor = C^-1/2 / sq^2; L % 64 == 0, L <= 4096.
 * The caller multiplies Q3 K3^T and P3 V3 (library fp16 GEMMs, fp32 out) and scales the result by 1 / (2^14 sv). */
int attn_split_qkv_f16x3(const float *qkv, void *Q3, void *K3, void *V3, int64_t B, int64_t L, int64_t C, float sq, float sv,
                         void *stream);
int attn_softmax_split_f16x3(const float *S, void *P3, int64_t rows, int64_t L, float factor, void *stream);
int f16_scales_from_gn_stats(const gqhip_gnstat_t *stats, int64_t n_bg, double amp, double u_scale, float *scales_out,
                             void *stream);

/* ---- index wire format / usage histogram ----------------------------------- */
int gq_index_histogram(const int64_t *idx, int64_t count, int64_t n,
                       int32_t *hist, void *stream);
int gq_indices_to_u16(const int64_t *idx, uint16_t *out, int64_t count,
                      void *stream);
int gq_indices_from_u16(const uint16_t *in, int64_t *idx, int64_t count,
                        void *stream);

/* One rank's per-step record in ONE launch -- what eval.py:165-169 (per-image PSNR: pit/evaluations/psnr.py:17-28 with
 * zero_mean = True, images in [-1, 1]) and eval.py:152-154 (the batch's code indices) publish per batch, in the packed wire
 * format of pit_hip/eval_dist.py:StepRecord with one metric:
 *     rec [B + (n_idx + 1) / 2] int32 = [ B PSNRs as fp32 bits | indices as uint16 pairs, low half first (odd count: zero pad) ].
 * x, x_rec: B images of `per_image` floats each in the SAME dense layout (NCHW or channels_last: the metric is elementwise);
 * idx: n_idx values in [0, 65536) (not checked; a wider value is truncated to its low 16 bits).  The squared differences are
 * formed in the reference's fp32 op order and summed in fp64 in a fixed order (bit-reproducible; within 2e-6 of torch's fp32
 * mean).  workspace: gq_step_record_workspace_bytes(B, per_image) bytes, ZERO when first used; every call leaves it zero
 * again, so one allocation serves a stream of calls (one stream at a time).  Asynchronous on `stream`, no allocation. */
int64_t gq_step_record_workspace_bytes(int64_t B, int64_t per_image);
int gq_step_record_f32(const float *x, const float *x_rec, const int64_t *idx, int32_t *rec, int64_t B, int64_t per_image,
                       int64_t n_idx, void *workspace_zeroed, int64_t workspace_bytes, void *stream);

/* ---- profiling recorder ------------------------------------------------------
 * When enabled, every launch of the MFMA filter kernel is bracketed with
 * hipEvents on its own stream.  gqhip_profile_collect synchronises those
 * events and returns the number of launches and their total/avg milliseconds;
 * it resets the recorder.  Disabled by default (zero overhead). */
int gqhip_profile_enable(int on);
/* Pre-create `pairs` event pairs (outside any timed region): profiled launches only take events from this pool, a
 * launch that finds it empty is not recorded; collected events return to the pool. */
int gqhip_profile_reserve(int pairs);
int gqhip_profile_collect(int *launches_host, double *total_ms_host);

/* Diagnostics of the last fused call on `workspace` (device-side counters,
 * copied out synchronously): rows sent to the exhaustive fallback and total
 * half-tiles re-ranked.  For tests / DESIGN.md statistics only. */
/* on != 0: the re-rank kernel also counts re-ranked half-tiles (adds one
 * contended atomic per row -- keep off when timing). */
int gqhip_debug_enable(int on);
int gqhip_debug_counters(const void *workspace, int64_t *fallback_rows_host,
                         int64_t *reranked_halftiles_host);
/* Grid search (dim 4 with a codebook cache), last call on `workspace`, synchronous copy: out4 = { sub-leaves visited summed
 * over the rows (counted only after gqhip_debug_enable(1); a sub-leaf is ~n / 4096 codes, a whole leaf counts four), codes that
 * received the reference's arithmetic (likewise; a row with ONE code within the margin needs none), rows handed to
 * gq_grid_finish_kernel, 1 if `cb_cache_or_null` holds a current index / 0 if not / -1 without a cache }. */
int gqhip_debug_grid(const void *workspace, const void *cb_cache_or_null, int64_t *out4_host);
#ifdef __cplusplus
}
#endif
#endif /* GQHIP_H_ */
