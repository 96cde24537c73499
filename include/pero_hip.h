/* pero_hip.h - C ABI of libpero_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * per-step hot path of DCGM/pero-pretraining (masked / joint-embedding pre-training over text-line
 * images).
 *
 * The reference is pure Python and has no FFI of its own for this path: every function below
 * replaces a PyTorch library call made by the reference (cited per function, paths relative to
 * /root/reference/pero_pretraining/).  A Python caller binds this header with ctypes (see
 * INTEGRATION.md); nothing in the signatures is a torch type.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (workspaces included); the library never
 *     allocates, frees, copies from the host or synchronises (its only host-side state: pero_set_option's
 *     knobs, and two immutable values initialised once, thread-safely: the device's CU count and each
 *     kernel's dynamic-LDS limit);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - `dtype`: PERO_F32 (parity mode, exact f32 MFMA / VALU arithmetic) or PERO_BF16 (bf16 storage
 *     and MFMA operands, f32 accumulation, f32 statistics);
 *   - return value 0 = ok, negative = PERO_E_*; text via pero_last_error() (thread local);
 *   - re-entrant, callable from any thread (autograd worker threads call the backward entry points).
 */
#ifndef PERO_HIP_H
#define PERO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PERO_F32 0
#define PERO_BF16 1

#define PERO_OK 0
#define PERO_E_INVALID (-1)   /* bad argument (shape, alignment, null pointer) */
#define PERO_E_UNSUPPORTED (-2)
#define PERO_E_LAUNCH (-3)    /* hipLaunch / runtime error; message holds hipGetErrorString */

/* epilogue / mode flags of pero_gemm */
#define PERO_GEMM_RELU 1        /* C = max(C, 0) after bias/residual */
#define PERO_GEMM_ATOMIC 2      /* f32 C only: the product is ADDED into C (split-K safe).  Without a `workspace` the additions are f32
                                 * atomics: several streams may add into one C at the same time.  WITH a `workspace` of
                                 * pero_gemm_workspace_bytes() the slices of a tile are summed in slice order and added to C by a plain
                                 * read-modify-write: bit-reproducible from run to run, but C must then have ONE writer at a time
                                 * (calls that add into the same C must be ordered on one stream or by events) */
#define PERO_GEMM_ACCUM 4       /* f32 C only: C += result (non-atomic) */
#define PERO_GEMM_TRANS_A 8     /* A is stored [K][M] (lda = row pitch of that storage) */
#define PERO_GEMM_TRANS_B 16    /* B is stored [K][N]; default B is stored [N][K] (Linear weight layout) */
#define PERO_GEMM_TILE128 64     /* force the persistent 128x128x64 bf16 kernel */
#define PERO_GEMM_TILE256 128    /* take the eight-phase 256x256x64 kernel whatever the tile count (when the shape allows: M, N % 256, K % 64, K >= 128) */
#define PERO_GEMM_TILE_V 512     /* accepted, no effect (a round-1 tile hint) */
#define PERO_GEMM_ROWDOT 4096   /* bf16 C, N % 128 == 0: `gate` is a bf16 matrix of C's shape that is NOT applied as a gate, and `bias`
                                 * is an OUTPUT (f32 [M][N/128]): bias[m][b] = sum over columns 128b..128b+127 of C[m][c] * gate[m][c],
                                 * with C as stored (bf16).  The attention backward's D = rowsum(dO * O) per head, out of the
                                 * epilogue of the product that writes dO.  No input bias / gate in this mode. */
#define PERO_GEMM_RELU_BITS 8192 /* the ReLU gate as a BIT MASK, bit (n & 7) of byte gate[m * ldg + n / 8] (ldg in bytes).  With
                                   PERO_GEMM_RELU `gate` is an OUTPUT: bit = (stored C[m][n] > 0); without it `gate` is that mask
                                   as INPUT and replaces the bf16 gate matrix (1/16 of its bytes).  bf16 C, batch 1, the 256-row
                                   tile kernels only (M % 256 == 0, N % 128 == 0, K % 32 == 0): anything else is PERO_E_INVALID */
#define PERO_GEMM_MASK_TILED 16384 /* with PERO_GEMM_RELU_BITS, N % 256 == 0: the bit mask is stored per 256-column block - bit (n & 7) of byte
                                   gate[(n / 256) * M * 32 + m * 32 + (n % 256) / 8] (ldg ignored; the same M * N / 8 bytes) - so that a 256 x 256
                                   tile's mask is 8 KiB of contiguous bytes (whole lines) instead of a quarter of a line per row.  Producer and consumer
                                   of a mask must agree. */
#define PERO_GEMM_COLSUM 1024   /* `bias` is an OUTPUT (f32 [N], accumulated atomically): column sums over the M rows of the
                                 * stored result - the bias gradient of the Linear whose output gradient this product
                                 * writes (replaces a separate pero_colsum pass over C).  No input bias in this mode. */
#define PERO_GEMM_FORCE_GENERIC 32 /* testing: take the exact-f32 generic kernel even when the fast bf16 kernel applies */

const char* pero_last_error(void);
int pero_abi_version(void);
/* tuning knobs for benchmarking and tests (defaults are the measured best): "gemm_policy" (0 = auto, 1 / 4 / 7 / 20 = force one tile-kernel
 * family, table in csrc/gemm.hip), "gemm_e256_min" (192: stored products with at least that many 256x256 tiles take the eight-phase
 * kernel; 0 = never), "gemm_e_splitk_min" (4), "gemm_e_var" (diagnostic builds of that kernel), "splitk_items" (512), "splitk_xcd" (1),
 * "splitk_nearest" (0), "attn_bwd_pair" (1: the attention backward with D handed in runs as one launch, csrc/attention.hip), "attn_pipe" (1: attention loops
 * with software-pipelined inline-asm operand reads; 0: the compiler-scheduled loops - same bits), "attn_lh" (0; 1: S = 256 backward as one persistent workgroup
 * per CU and (line, head) - same bits, not faster), "splitk_workspace" (1: the split-K
 * products of the eight-phase kernel leave partial tiles in the caller's `workspace`, summed in slice order by a second kernel - deterministic;
 * 0: f32 atomics even when a workspace is passed), "splitk_table" (1: unaligned slice counts hand their work items out XCD by XCD), "gemm_nw" (0; 1: stored
 * N = 512 products with a bias / residual epilogue run on the row-complete 128 x 512 tile that pero_gemm_resid_layernorm uses - same bits).  Process-wide; not
 * meant to be changed while products are in flight. */
int pero_set_option(const char* name, int value);

/* ---- front end ------------------------------------------------------------------------------
 * images u8 (N,H,W,C) -> patch rows (N*S, C*H*P) ordered (c,h,p), value/255, masked patches replaced
 * by the (C,H,P) noise tile.  Replaces BatchOperator._prepare_batch_images
 * (masked_pretraining/batch_operator.py:17-20) + TransformerEncoder.mask (models/transformers.py:53-68)
 * + the im2col of Conv2d(kernel=stride=(H,P)) (models/transformers.py:99-107).  mask may be null.
 * ld_out >= C*H*P is the row pitch of `patches` in elements; the padding columns are zero-filled (a pitch
 * that is a multiple of 128 lets the weight-gradient GEMM of the patch embedding run on the fast kernel). */
int pero_patches_from_u8(const uint8_t* images, const int64_t* mask, const float* tile, void* patches,
                         int64_t N, int64_t H, int64_t W, int64_t C, int64_t P, int64_t ld_out, int dtype, void* stream);
/* same from float NCHW images (the reference's own model input layout) */
int pero_patches_from_f32(const float* images_nchw, const int64_t* mask, const float* tile, void* patches,
                          int64_t N, int64_t H, int64_t W, int64_t C, int64_t P, int64_t ld_out, int dtype, void* stream);
/* in-place TransformerEncoder.mask on float NCHW images (models/transformers.py:53-68) */
int pero_apply_mask_f32(float* images_nchw, const int64_t* mask, const float* tile,
                        int64_t N, int64_t H, int64_t W, int64_t C, int64_t P, void* stream);

/* ---- GEMM -------------------------------------------------------------------------------------
 * C[b] = alpha * op(A[b]) * op(B[b])^T (+ bias[n]) (+ residual) (relu) (* (gate > 0)),  M x N x K.
 * Replaces torch.nn.Linear / F.linear / torch.matmul / Conv2d-as-GEMM call sites:
 * models/transformers.py:37-43,99 ; masked_pretraining/model.py:102 ; models/autoencoders.py:214 ;
 * joint_embedding_pretraining/losses.py:42,77 and their autograd backward products.
 * Batch b = bo * batch_inner + bi; operand base offset = bo * s?o + bi * s?i (elements).
 * in_dtype: A, B (and residual, gate); out_dtype: C.  bias is f32.  k_split > 1 needs PERO_GEMM_ATOMIC;
 * k_split == 0 with PERO_GEMM_ATOMIC lets the library choose tile size and split.
 * workspace (device, 16-byte aligned, may be null) / workspace_bytes: scratch for this ONE call, owned by the caller and free again
 * when the work enqueued by the call has run (calls on one stream may share it; concurrent streams need one each).  Only the split-K
 * weight-gradient products use it (partial tiles, see PERO_GEMM_ATOMIC); with less than pero_gemm_workspace_bytes() they add with
 * f32 atomics instead - same value up to the order of the additions. */
int pero_gemm(const void* A, const void* B, void* C, const float* bias, const void* residual, const void* gate,
              int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldg,
              int64_t batch, int64_t batch_inner,
              int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi,
              float alpha, int flags, int k_split, int in_dtype, int out_dtype, void* workspace, int64_t workspace_bytes,
              void* stream);
/* bytes of `workspace` with which pero_gemm runs the product of this shape / flags / k_split deterministically (0: it needs none).
 * Depends on the arguments and the pero_set_option knobs only (no device query). */
int64_t pero_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t batch, int flags, int k_split, int in_dtype, int out_dtype);

/* Linear + residual + LayerNorm in ONE launch (bf16; N = 512, M % 128 == 0, K % 64 == 0, K >= 192): Y = A W^T + bias + R stored (the backward
 * reads it), T = LayerNorm(Y) * gamma + beta computed from the rounded rows of Y exactly as pero_layernorm_fwd does, mean / rstd f32 per row.
 * Replaces the pair (Linear with the residual add, torch.nn.LayerNorm) of TransformerEncoderLayer's post-norm blocks
 * (x = norm1(x + out_proj(attn)), x = norm2(x + linear2(...)): models/transformers.py:36-43) - the LayerNorm's read of Y goes away.
 * bias may be null.  PERO_E_INVALID for other shapes: the caller falls back to pero_gemm + pero_layernorm_fwd.
 * Y may be null (round 4): the pre-norm rows are then not stored at all - a backward pass through pero_layernorm_bwd_out needs T and rstd
 * only (16 fewer 16-byte stores per lane and tile, a quarter of the launch's HBM traffic at K = 512). */
int pero_gemm_resid_layernorm(const void* A, const void* W, const float* bias, const void* R, const float* gamma, const float* beta,
                              void* Y, void* T, float* mean, float* rstd, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                              int64_t ldy, int64_t ldr, int64_t ldt, float eps, void* stream);

/* Input gradient of a Linear + residual gradient + LayerNorm BACKWARD of the norm in front of that Linear, in ONE launch (round 4; bf16, N = 512,
 * M % 128 == 0, K % 64 == 0, K >= 192): dt = A Wt^T + R with the rows complete in a workgroup (Wt = the transposed bf16 weight copy, N x K), then
 * pero_layernorm_bwd_out's arithmetic on the rounded dt: DX = rstd (dt gamma - mean(dt gamma) - xhat mean(dt gamma xhat)), xhat = (T - beta) / gamma;
 * dgamma / dbeta / dxsum (dxsum may be null) are ACCUMULATED (+=) from per-workgroup partial column sums.  dt itself never reaches memory: replaces
 * pero_gemm (residual epilogue) + pero_layernorm_bwd_out for the two input-gradient products of TransformerEncoderLayer that feed a norm's backward
 * (linear1 -> norm1, in_proj -> the previous layer's norm2; models/transformers.py:36-43).  work: f32, 3 * PERO_LN_BWD_BLOCKS * 512 elements. */
int pero_gemm_resid_layernorm_bwd(const void* A, const void* Wt, const void* R, const void* T, const float* rstd, const float* gamma,
                                  const float* beta, void* DX, float* dgamma, float* dbeta, float* dxsum, float* work, int64_t M, int64_t N,
                                  int64_t K, int64_t lda, int64_t ldw, int64_t ldr, int64_t ldt, int64_t lddx, void* stream);

/* ---- LayerNorm (+ positional encoding) --------------------------------------------------------------
 * y = (x - mean) * rstd * gamma + beta (+ pe[offsets[row / S] + row % S]) ; rows x d.
 * Replaces torch.nn.LayerNorm (models/transformers.py:28,83-84 and the norm1/norm2 of
 * TransformerEncoderLayer) and PositionalEncoding.forward (models/transformers.py:174-188).
 * pe (f32 [max_len][d]) and offsets (int64 [rows/S]) may be null (offsets null => offset 0). */
int pero_layernorm_fwd(const void* x, const float* gamma, const float* beta, const float* pe, const int64_t* offsets,
                       void* y, float* mean, float* rstd, int64_t rows, int64_t d, int64_t S, float eps,
                       int dtype, void* stream);
/* dx, and dgamma/dbeta/dxsum ACCUMULATED (+=) into f32 [d] buffers (dxsum = column sums of dx, i.e. the bias
 * gradient of the Linear that produced x; may be null).  work: f32 workspace of 3 * PERO_LN_BWD_BLOCKS * d. */
#define PERO_LN_BWD_BLOCKS 512
int pero_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                       void* dx, float* dgamma, float* dbeta, float* dxsum, float* work, int64_t rows, int64_t d,
                       int dtype, void* stream);
/* The same backward from the layer's OUTPUT t = xhat * gamma + beta instead of its input rows: xhat = (t - beta) / gamma, so the
 * forward pass keeps t (which the next Linear reads anyway) and rstd, not x and mean ("memory-efficient" LayerNorm; torch.nn.LayerNorm
 * saves its input - models/transformers.py:28,36-43 - the gradient is the same function of the same quantities).  In f32 the two forms
 * agree to rounding; in bf16 t carries the 2^-9 relative rounding x carried before.  A column with gamma == 0 exactly has no xhat left in
 * t: it is taken as 0 (that column's gamma gradient and its -xhat * c2 share of dx are lost, nothing is NaN, other columns are exact) -
 * keep pero_layernorm_bwd where a scale can be exactly zero.  Same accumulation semantics and workspace as above. */
int pero_layernorm_bwd_out(const void* dy, const void* t, const float* rstd, const float* gamma, const float* beta,
                           void* dx, float* dgamma, float* dbeta, float* dxsum, float* work, int64_t rows, int64_t d,
                           int dtype, void* stream);

/* ---- BatchNorm1d (+ ReLU) over the rows of an (rows, d) matrix: the optional `use_bn=True` layers of the joint-embedding MLPHead
 * (Linear -> torch.nn.BatchNorm1d(hidden) -> ReLU on the (N*S, hidden) rows: joint_embedding_pretraining/model.py:99-103).
 * training != 0: batch statistics (mean, BIASED variance; two passes), save_mean / save_rstd (f32 [d]) for the backward, and - when given -
 * running_mean / running_var updated in place as torch does (momentum; the running variance takes the unbiased batch variance).
 * training == 0: the running statistics.  y = (x - mean) * rstd * weight + bias, then ReLU if relu != 0.  work: f32 [2 * d].  Deterministic
 * (no atomics).  Statistics are per CALL, i.e. per data-parallel rank (torch.nn.BatchNorm1d under DDP without SyncBatchNorm). */
int pero_bn_fwd(const void* x, const float* weight, const float* bias, float* running_mean, float* running_var, void* y,
                float* save_mean, float* save_rstd, float* work, int64_t rows, int64_t d, float eps, float momentum, int training,
                int relu, int dtype, void* stream);
/* dx; dweight / dbias (may be null) ACCUMULATED (+=).  relu != 0: dy is first zeroed where the layer's output y is <= 0.  work: f32 [2 * d]. */
int pero_bn_bwd(const void* dy, const void* x, const void* y, const float* weight, const float* save_mean, const float* save_rstd,
                void* dx, float* dweight, float* dbias, float* work, int64_t rows, int64_t d, int relu, int dtype, void* stream);

/* ---- softmax over the last dim (attention probabilities; torch SDPA inside
 * TransformerEncoderLayer._sa_block, models/transformers.py:86) ---------------------------------------
 * p = softmax(scale * s) row-wise; s is f32 (rows, cols), p has `dtype` */
int pero_softmax_fwd(const float* s, void* p, int64_t rows, int64_t cols, float scale, int dtype, void* stream);
/* ds = scale * p * (dp - sum_j p*dp); dp is f32, p and ds have `dtype` */
int pero_softmax_bwd(const void* p, const float* dp, void* ds, int64_t rows, int64_t cols, float scale, int dtype,
                     void* stream);

/* ---- fused attention (bf16, head_dim 128, S % 128 == 0) on the packed qkv (N*S, 3*nh*128) tensor -----------------
 * out (N*S, nh*128) = softmax(q k^T / sqrt(hd)) v per (line, head); lse (N*nh, S) f32 = base-2 log-sum-exp of the
 * scaled scores (kept for the backward kernels).  Scores never touch memory.  Other shapes / f32: use the
 * batched pero_gemm + pero_softmax_* path. */
int pero_attention_fwd(const void* qkv, void* out, float* lse, int64_t N, int64_t S, int64_t num_heads,
                       int64_t head_dim, int dtype, void* stream);
/* dqkv (N*S, 3d) from dout (N*S, d).  dvec (N*S, nh) f32: D[row][head] = sum over the head's 128 columns of dout*out -
 * computed and stored by the call when `out` is given, or supplied by the caller (out == null; e.g. written by the
 * PERO_GEMM_ROWDOT epilogue of the product that produced dout).  dbias (f32 [3d], may be null):
 * the column sums of dqkv - in_proj's bias gradient - are ACCUMULATED into it: per-workgroup partial rows from the kernels'
 * staged output tiles into work (f32, 3 * N * nh * (S/128) * 128 elements; required with dbias), then one small reduction. */
int pero_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dvec, void* dqkv,
                       float* dbias, float* work, int64_t N, int64_t S, int64_t num_heads, int64_t head_dim, int dtype,
                       void* stream);

/* ---- masked cross entropy (masked_pretraining/model.py:72-95) -------------------------------------------
 * logits (rows, V); labels, mask int64 (rows).  loss_out[0] = mean CE over mask==1 rows
 * (+ unmasked_weight * mean CE over mask==0 & label>=0 rows when unmasked_weight >= 0; pass a negative
 * value for "None").  work: f32 workspace of PERO_CE_WORK(rows) elements, kept by the caller for the backward
 * call (row losses, the two row counts, row logsumexps, partial sums of the final reduction).  Empty selections give NaN
 * like the reference.
 * bwd: dlogits (dtype, rows x V) = dloss[0] * d loss / d logits (dloss: device f32 scalar, null = 1). */
#define PERO_CE_WORK(rows) (2 * (rows) + 8 + 256)
int pero_masked_ce_fwd(const void* logits, const int64_t* labels, const int64_t* mask, float unmasked_weight,
                       float* loss_out, float* work, int64_t rows, int64_t V, int dtype, void* stream);
int pero_masked_ce_bwd(const void* logits, const int64_t* labels, const int64_t* mask, float unmasked_weight,
                       const float* dloss, const float* work, void* dlogits, int64_t rows, int64_t V, int dtype,
                       void* stream);
/* pero_masked_ce_fwd for a caller that knows the rows with mask == 1 (index: int64 device list of n_idx rows) and passes
 * unmasked_weight "None": only those rows are visited (one workgroup each instead of one per position); same loss bits. */
int pero_masked_ce_fwd_rows(const void* logits, const int64_t* labels, const int64_t* mask, const int64_t* index, int64_t n_idx,
                            float* loss_out, float* work, int64_t rows, int64_t V, int dtype, void* stream);
/* The same gradient in COMPACT form: dlogits_rows (dtype, n_rows_out x V) row i = the gradient row of logits row index[i] for
 * i < n_idx (bit-identical to row index[i] of pero_masked_ce_bwd's result), zero rows for n_idx <= i < n_rows_out (padding up to whole
 * GEMM tiles).  With unmasked_weight "None" every row outside mask == 1 of the dense gradient is an exact zero, so the head's
 * backward products (masked_pretraining/model.py:60-61 through autograd) run on the listed rows alone.  index: int64 device list. */
int pero_masked_ce_bwd_rows(const void* logits, const int64_t* labels, const int64_t* mask, float unmasked_weight,
                            const float* dloss, const float* work, const int64_t* index, int64_t n_idx, int64_t n_rows_out,
                            void* dlogits_rows, int64_t rows, int64_t V, int dtype, void* stream);

/* ---- reductions / elementwise ---------------------------------------------------------------------- */
/* out[n] += sum_m x[m][n]  (bias gradients); out f32 */
int pero_colsum(const void* x, float* out, int64_t rows, int64_t cols, int64_t ld, int dtype, void* stream);
/* out[m][b] = sum over columns 128b..128b+127 of x[m][c] * y[m][c]  (bf16 x, y; the pass PERO_GEMM_ROWDOT fuses) */
int pero_rowdot_blocks(const void* x, const void* y, float* out, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy,
                       void* stream);
/* Transposed bf16 copies of every matrix of a flat buffer in one launch (2-byte elements): table (device, int64[n][5]) =
 * {src offset, dst offset, rows, cols, first tile} in elements / 64x64 tiles; dst matrix t is [cols][rows].  The input
 * gradient dX = dY W (torch.nn.Linear backward, models/transformers.py:36-43 layers) reads these K-contiguous copies. */
int pero_transpose_multi(const void* src, void* dst, const int64_t* table, int64_t n_matrices, int64_t total_tiles, void* stream);
/* f32 -> bf16 copy (low-precision weight copies) */
int pero_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int pero_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream);
/* dst[r][0:cols] = bf16(src[r][0:cols]), dst[r][cols:ld_dst] = 0   (K-padded low-precision weight copy) */
int pero_cast_pad_f32_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_dst, void* stream);
/* dst[r][c] += src[r][c] for c < cols (f32, row pitches ld_dst / ld_src) */
int pero_add_rows2d(float* dst, const float* src, int64_t rows, int64_t cols, int64_t ld_dst, int64_t ld_src, void* stream);
/* nbytes of zeros at p */
int pero_zero_fill(void* p, int64_t nbytes, void* stream);
/* y = x * scale (in place allowed), dtype elements */
int pero_scale(void* x, int64_t n, float scale, int dtype, void* stream);

/* ---- Adam (torch.optim.Adam defaults: masked_pretraining/train.py:146) ----------------------------------
 * one launch over a flat f32 parameter / gradient / moment buffer; `step` is 1-based; optionally also
 * writes the bf16 copy of the updated parameters (p_bf16 may be null). */
int pero_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, double lr,
                   double beta1, double beta2, double eps, int64_t step, double grad_scale, void* stream);

/* ---- quantizers (models/autoencoders.py:212-217 ; scripts/produce_kmeans_labels.py:72-76) ---------------
 * indices[m] = argmin_k ( sum(x_m^2) + sum(e_k^2) - 2 x_m . e_k )  in exact f32, first minimum wins.
 * x (M,D) f32, codebook (K,D) f32, indices int64 (M).  best_dist (f32, M) may be null.
 * work: f32 workspace of at least M + K elements (row / code squared norms). */
int pero_vq_argmin(const float* x, const float* codebook, int64_t* indices, float* best_dist, float* work,
                   int64_t M, int64_t K, int64_t D, void* stream);
/* quantized[m] = x[m] + (codebook[indices[m]] - x[m])  (straight-through arithmetic, autoencoders.py:239) */
int pero_vq_gather(const float* x, const float* codebook, const int64_t* indices, float* quantized,
                   int64_t M, int64_t D, void* stream);
/* EMA codebook update of the tokenizer in training mode (models/autoencoders.py:225-237), in place:
 * ema_cluster_size (K) <- Laplace-smoothed decayed counts, ema_w (K,D) <- decayed sums of the rows assigned to each
 * code, codebook (K,D) <- ema_w / ema_cluster_size.  indices are those of pero_vq_argmin on the OLD codebook.
 * work: f32 workspace of K + K*D elements (zeroed by the call). */
int pero_vq_ema_update(const float* x, const int64_t* indices, float* ema_cluster_size, float* ema_w, float* codebook,
                       float* work, int64_t M, int64_t K, int64_t D, double decay, double epsilon, void* stream);

/* ---- row gather / scatter by index (boolean-mask selections of the losses) ----------------------------- */
/* dst[i] = src[index[i]] for i < n_idx, zero rows for n_idx <= i < n_rows_out (padding) */
int pero_gather_rows(const void* src, const int64_t* index, void* dst, int64_t n_idx, int64_t n_rows_out,
                     int64_t d, int dtype, void* stream);
/* dst[index[i]] += src[i] (indices unique) */
int pero_scatter_add_rows(const void* src, const int64_t* index, void* dst, int64_t n_idx, int64_t d,
                          int dtype, void* stream);

/* ---- joint-embedding losses (joint_embedding_pretraining/losses.py) ---------------------------------------
 * The dense products (covariance SYRK z^T z, its backward zc @ G, per-line similarity x y^T and its backward)
 * are pero_gemm calls; these entry points are the surrounding reductions.  `g` arguments are DEVICE f32
 * scalars holding the upstream gradient (null = 1), so backward needs no host synchronisation. */
/* out[0] = scale * sum_rows |x[ix[r]] - y[iy[r]]|^2   (VICReg invariance, losses.py:14-16); partial: f32 [n] */
int pero_sqdiff_rows(const void* x, const int64_t* ix, const void* y, const int64_t* iy, float* partial, float* out,
                     int64_t n, int64_t d, float scale, int dtype, void* stream);
/* dx[ix[r]] += g*coef*(x-y) ; dy[iy[r]] -= g*coef*(x-y) */
int pero_sqdiff_rows_bwd(const void* x, const int64_t* ix, const void* y, const int64_t* iy, void* dx, void* dy,
                         const float* g, float coef, int64_t n, int64_t d, int dtype, void* stream);
/* out[0] = scale * sum(partial[0..n)) in a fixed order */
int pero_sum_scale(const float* partial, float* out, int64_t n, float scale, void* stream);
/* zc = z - colsum/m for rows < m, 0 for padding rows m..m_pad; sumsq[c] += sum_r zc[r][c]^2 (losses.py:38,41) */
int pero_center_cols(const void* z, const float* colsum, void* zc, float* sumsq, int64_t m, int64_t m_pad, int64_t d,
                     int dtype, void* stream);
/* variance hinge loss_var = mean_j relu(threshold - sqrt(sumsq_j/(m-1) + eps)) and its per-column gradient
 * coefficient cvar_j (losses.py:37-38) */
int pero_vicreg_var(const float* sumsq, float* cvar, float* loss_var, int64_t m, int64_t d, float threshold, float eps,
                    void* stream);
/* loss_cov = sum_{i!=j} cov_ij^2 / d (losses.py:40-47) and the backward operand G (dtype, d x d):
 * G_ij = wc*4*cov_ij/(d*(m-1)) (i != j), G_jj = wv*cvar_j, so that d(wv*var + wc*cov)/d zc = zc @ G.
 * rowpart: f32 [d] scratch */
int pero_vicreg_cov(const float* cov, const float* cvar, void* G, float* rowpart, float* loss_cov, int64_t d, int64_t m,
                    float wv, float wc, int dtype, void* stream);
/* dst[index[i]] += g * src[i] */
int pero_scatter_add_rows_scaled(const void* src, const int64_t* index, void* dst, const float* g, int64_t n, int64_t d,
                                 int dtype, void* stream);
/* xn = x / max(|x|_2, 1e-12) per row, inv[r] = 1 / max(|x_r|, 1e-12)   (F.normalize, losses.py:58-59) */
int pero_rownorm_fwd(const void* x, void* xn, float* inv, int64_t rows, int64_t d, int dtype, void* stream);
int pero_rownorm_bwd(const void* xn, const void* dxn, const float* inv, const float* g, void* dx, int64_t rows, int64_t d,
                     int dtype, void* stream);
/* sim (lines, S, S) f32: line_loss[l] = mean_j (logsumexp_r sim[l][r][j] - sim[l][j][j]) (losses.py:80, softmax over
 * dim 0), loss_out[0] = mean_l line_loss[l]; dsim (dtype, may be null) = d loss_out / d sim */
int pero_ntxent_cols(const float* sim, float* line_loss, float* loss_out, void* dsim, int64_t lines, int64_t S, int dtype,
                     void* stream);
/* Cross-rank negatives (NTXentLoss(cross_rank_negatives=True): the data-parallel extension named by BASELINE.json's north_star; the
 * reference's loss is per line and has no such term).  cross (lines*S, L) f32: similarities of every view-2 row with the L pooled
 * embeddings of all ranks' lines (pooled embedding own0 + l belongs to line l itself and is left out).  Column j of line l is
 * normalised over its S own rows AND those L - 1 negatives in one log-sum-exp:
 *   line_loss[l] = mean_j ( log( sum_i exp(sim[l][i][j]) + sum_{l' != own0 + l} exp(cross[l*S + j][l']) ) - sim[l][j][j] ),
 * loss_out[0] = mean_l line_loss[l]; dsim (dtype, lines x S x S) and dcross (dtype, lines*S x L), may be null = d loss_out / d input. */
int pero_ntxent_cols_cross(const float* sim, const float* cross, float* line_loss, float* loss_out, void* dsim, void* dcross,
                           int64_t lines, int64_t S, int64_t L, int64_t own0, int dtype, void* stream);
/* out[l][c] (f32) = mean over the S rows of line l of x[l*S + s][c] (the pooled embedding of a line); d % 8 == 0 */
int pero_line_mean(const void* x, float* out, int64_t lines, int64_t S, int64_t d, int dtype, void* stream);
/* dst[l*S + s][c] += scale * src[l][c] for every row s of line l (the backward of pero_line_mean: scale = 1 / S); dst dtype, src f32 */
int pero_add_line_rows(void* dst, const float* src, int64_t lines, int64_t S, int64_t d, float scale, int dtype, void* stream);

/* ---- evaluation -------------------------------------------------------------
 * replaces masked_pretraining/tester.py:72-113 (_update_errors / _topk / _calculate_errors: host numpy argmax and
 * argsort over the full logit tensor).  For every row with mask == 1: gt = #{j : logit[j] > logit[label]},
 * eq_lo / eq_hi = #{j : logit[j] == logit[label], j < label / j > label}.  counters (u64 [1 + nk], ACCUMULATED, caller
 * zeroes them once per test()): [0] += rows with mask == 1 ("length"), [1 + i] += rows whose label is not in the top
 * ks[i] (k == 1: argmax = first maximum; k > 1: stable-argsort order, see csrc/eval.hip).  ranks (int32 [rows][3],
 * may be null): gt, eq_lo, eq_hi, or -1 where mask != 1.  ks is a DEVICE pointer (nk <= PERO_MAX_TOPK). */
#define PERO_MAX_TOPK 8
int pero_label_rank(const void* logits, int64_t ld, const int64_t* labels, const int64_t* mask, int64_t rows, int64_t V,
                    const int32_t* ks, int32_t nk, uint64_t* counters, int32_t* ranks, int dtype, void* stream);

/* ---- batch collation -------------------------------------------------------------
 * replaces common/dataloader.py:68-155 (BatchCreator.stack_images: host numpy zero-fill + slice copies + mask loops).
 * packed: the ragged uint8 lines back to back, line b = (H, widths[b], C) row-major at byte offsets[b]; the buffer must
 * be readable 8 bytes past its end.  out (B, H, Wt, C) u8, every byte written: line b's pixels at columns
 * [left_px[b], left_px[b] + widths[b]), zero elsewhere.  Wt*C must be a multiple of 16 (Wt % 32 == 0 in the reference). */
int pero_stack_lines(const void* packed, const int64_t* offsets, const int32_t* widths, const int32_t* left_px, void* out,
                     int64_t B, int64_t H, int64_t Wt, int64_t C, void* stream);
/* image masks (B, S) u8 (dataloader.py:92-96), shifts (B) i32 = crop_shifts + left1 - left2 (:126; left* in label
 * positions), three-valued shift masks (:124-138).  Second-view outputs may all be null (unpaired batch);
 * crop_shifts may be null (= 0). */
int pero_line_masks(const int32_t* widths1, const int32_t* widths2, const int32_t* left1, const int32_t* left2,
                    const int32_t* crop_shifts, uint8_t* image_masks1, uint8_t* image_masks2, uint8_t* shift_masks1,
                    uint8_t* shift_masks2, int32_t* shifts, int64_t B, int64_t S, int64_t subsampling, void* stream);

#ifdef __cplusplus
}
#endif
#endif
