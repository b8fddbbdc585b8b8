/* sconf.h — C ABI of libsconf_hip.so: the MI355X (gfx950) SConformerXL hot path.
 *
 * Drop-in boundary.  The reference (robflynnyh/long-context-asr, `lcasr`) has no FFI of its
 * own: its seams are the optional native ops it imports inside try/except.  Each entry point below names
 * the reference interface (file:line under /root/reference) it replaces.  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless stated otherwise;
 *   - all buffers are caller-owned and only borrowed for the enqueued work; kernels are enqueued on `stream`
 *     and never synchronise, allocate or free (graph-capture safe);
 *   - return 0 on success, non-zero on error with a message in sconf_last_error() (thread-local);
 *   - dtype enums: 0 = float32, 1 = bfloat16.  "ACCUMULATED" outputs are += with f32 atomics.
 */
#ifndef SCONF_H
#define SCONF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* sconf_stream_t; /* == hipStream_t */

enum { SCONF_F32 = 0, SCONF_BF16 = 1 };
enum { SCONF_ACT_NONE = 0, SCONF_ACT_GELU = 1, SCONF_ACT_SILU = 2, SCONF_ACT_DGELU = 3, SCONF_ACT_DSILU = 4,
       SCONF_ACT_GELU_DSAVE = 5 /* out = gelu(v), pre = gelu'(v) */, SCONF_ACT_MULAUX = 6 /* out = v * aux */,
       SCONF_ACT_SMAXBWD = 7 /* out = (v - rowv[m]) * aux: softmax backward, sconf_gemm_softmax_bwd only */ };
enum { SCONF_GEMM_NT = 0, SCONF_GEMM_NN = 1, SCONF_GEMM_TN = 2 };
enum { SCONF_NORM_LAYER = 0, SCONF_NORM_RMS = 1, SCONF_NORM_RMS_APEX = 2 };

const char* sconf_last_error(void);
int sconf_version(void);
int sconf_num_cus(void);

/* C[M,N] = resid + alpha * act(A·B + bias), bf16 operands, f32 accumulate (MFMA).
 * layout NT: A[M][K], B[N][K]  (F.linear / fused_dense_cuda.linear_act_forward, fused_dense.py:277-279,329-332,465-469)
 * layout NN: A[M][K], B[K][N]  (dgrad: fused_dense_cuda.bias_act_linear_dgrad_bgrad, fused_dense.py:354-356)
 * layout TN: A[K][M], B[K][N]  (wgrad: fused_dense_cuda.linear_bias_wgrad, fused_dense.py:113-115,338-340,375-378)
 * act DGELU/DSILU multiply by act'(aux[M][N]); pre (nullable) receives A·B+bias in bf16 (save_pre_act);
 * split_k > 1 (plain epilogue only): C must hold sconf_gemm_num_splits(K, split_k) f32 slabs of M*ldc elements; slab s
 * receives the partial sum of K-range s with plain stores (deterministic); combine with sconf_splitk_reduce. */
int sconf_gemm_bf16(int layout, const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc, const float* bias, const float* resid, int64_t ldr,
                    const void* aux, int64_t ldaux, void* pre, int64_t ldpre, float alpha, int act, int out_f32,
                    int split_k, sconf_stream_t stream);

/* qkv projection with the rotary rotation in the GEMM epilogue (attention.py:485,498-507 + rotary_emb.py:61-73 without a pass over
 * the activation): C (M, 3, H, D) bf16 = [q | k | v] = x W^T (+ bias), W (3 H D, K) bf16 in the REGROUPED row order (sconf_cast_shadows
 * with R < 0; bias regrouped the same way), q and k rotated by the NeoX rotary of position (row % seq_len); cos / sin (seq_len, D/2) f32.
 * head_dim 128 with M, 3 H D multiples of 256 runs the fused epilogue; other shapes run sconf_gemm_bf16 + sconf_rotary_inplace. */
int sconf_gemm_qkv_rotary(const void* A, const void* W, void* C, int64_t M, int64_t K, int64_t H, int64_t D,
                          int64_t lda, int64_t ldb, const float* bias, const float* cos_tab, const float* sin_tab,
                          int64_t seq_len, sconf_stream_t stream);

/* Softmax backward inside the dgrad GEMM of the self-conditioning reprojection (sconformer_xl.py:241-243, backward of
 * x + reprojection(softmax(ff(norm(x))))): dl (M, V) bf16 = probs * (dy Wt^T - delta), dy (M, K) bf16, Wt (V, K) bf16 (the transposed
 * reprojection weight), probs (M, V) bf16, delta (M) f32 = sum_v probs * (dy Wt^T) - obtained WITHOUT that product as
 * sconf_rowdot(dy, saved reprojection output before residual, reprojection bias).  colslab (2 M / 256, V) f32 receives column sums of dl
 * per (256-row panel, wave row): sconf_colsum over it gives the decoder bias gradient.  Returns 2 (nothing launched) when the shape does
 * not take the 256-row NT kernel (M % 256, V % 256 or 192, K % 64): the caller then uses sconf_gemm_bf16 + sconf_softmax_bwd. */
int sconf_gemm_softmax_bwd(const void* dy, const void* Wt, const void* probs, const float* delta, void* dl, float* colslab,
                           int64_t M, int64_t V, int64_t K, int64_t lddy, int64_t ldw, int64_t ldp, sconf_stream_t stream);
/* out[m] = sum_c a[m][c] * (b[m][c] - bias[c])   (a, b bf16 (M, d); bias f32 (d) or null; d % 8 == 0) */
int sconf_rowdot(const void* a, const void* b, const float* bias, float* out, int64_t M, int64_t d, int64_t lda, int64_t ldb,
                 sconf_stream_t stream);
int sconf_gemm_num_splits(int64_t K, int split_k);
/* Which kernel sconf_gemm_bf16 runs for a problem (bookkeeping for benchmarks): 0 = 128x128-tile kernel, 1 / 2 = 256-row NT
 * kernel with 256 / 192-wide tiles, 3 = 256x256 TN kernel; -1 = invalid arguments. */
int sconf_gemm_variant(int layout, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int split_k, int act,
                       int has_resid, int has_pre);
int sconf_splitk_reduce(const float* slab, float* out, int64_t splits, int64_t n, int accumulate, sconf_stream_t stream);

/* Row norms over the last dim d <= 2048 (apex FusedLayerNorm/FusedRMSNorm, torch LayerNorm, local RMSNorm:
 * sconformer_xl.py:14-17, normalisation.py:6-47).  mean/rstd: f32 [M] saved statistics. */
int sconf_norm_fwd(int mode, const void* x, int x_dtype, const float* weight, const float* bias, void* y, int y_dtype,
                   float* mean, float* rstd, int64_t M, int64_t d, float eps, sconf_stream_t stream);
/* dx = (dres ? dres : 0) + norm'(x)·dy ; dweight/dbias ACCUMULATED.  workspace: optional scratch of at least
 * sconf_norm_bwd_workspace(M, d) floats (per-workgroup column sums, combined in a fixed order); NULL = f32 atomics. */
int64_t sconf_norm_bwd_workspace(int64_t M, int64_t d);
int sconf_norm_bwd(int mode, const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight,
                   const float* mean, const float* rstd, const float* dres, void* dx, int dx_dtype, float* dweight,
                   float* dbias, float* workspace, int64_t workspace_floats, void* dx_bf16, float* dx_colsum,
                   int64_t M, int64_t d, float eps, sconf_stream_t stream);
/* dx_bf16 / dx_colsum (both or neither; need f32 dx and the workspace): a bf16 copy of dx and its column sums [d] (overwritten),
 * for the block that receives dx as its output gradient - its GEMM operand and its output-projection bias gradient. */

/* Two LayerNorms back to back in one pass (d <= 768): y1 = LN(x; w1, b1) f32 and h2 = LN(y1; w2, b2) bf16 - a ConformerLayer's `norm_out`
 * followed by the decoder norm of the self-conditioning step / head (sconformer_xl.py:371, 241-247; decoder.py:23).  The backward
 * takes the gradient of h2 (bf16) and the gradient reaching y1 directly (dres, f32, nullable), recomputes y1 from x and the saved
 * statistics, and ACCUMULATES the four parameter gradients; dx_bf16 / dx_colsum as in sconf_norm_bwd; the workspace is required. */
int sconf_norm2_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y1, void* h2_bf16,
                    float* mean1, float* rstd1, float* mean2, float* rstd2, float* mean3, float* rstd3, int twice,
                    int64_t M, int64_t d, float eps1, float eps2, sconf_stream_t stream);
int64_t sconf_norm2_bwd_workspace(int64_t M, int64_t d);
int sconf_norm2_bwd(const void* dh2_bf16, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                    const float* mean1, const float* rstd1, const float* mean2, const float* rstd2,
                    const float* mean3, const float* rstd3, int twice, const float* dres,
                    float* dx, float* dw1, float* db1, float* dw2, float* db2, float* workspace, int64_t workspace_floats,
                    void* dx_bf16, float* dx_colsum, int64_t M, int64_t d, sconf_stream_t stream);
/* twice != 0: the second norm is applied two times, h2 = LN(LN(y1; w2, b2); w2, b2) - the head's legacy double norm after the last
 * layer (sconformer_xl.py:246-247, legasee_double_norm); mean3 / rstd3 then hold the statistics of the inner result (else unused). */

int sconf_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, sconf_stream_t stream);
/* dst (C,R) bf16 = transpose(src (R,C) f32): transposed weight shadow so that dgrad GEMMs are NT. */
int sconf_cast_transpose(const float* src, void* dst, int64_t R, int64_t C, sconf_stream_t stream);
/* One launch for all bf16 weight shadows of a model.  table: device array of n_entries + 1 records of six int64
 * {src f32 (R,C), dst bf16 (R,C) or 0, dstT bf16 (C,R) or 0, R (negative: |R| rows, written regrouped 3j + w -> w |R|/3 + j: the
 * "(h d qkv)" rows of the fused qkv projection, attention.py:485, as [q | k | v]), C, first 32x32-tile index}; the last record is the sentinel
 * {0,0,0,0,0,total_tiles}.  Stands in for the per-module weight casts torch.autocast does in the reference
 * (lcasr/utils/general.py, training loop under autocast) plus the transposed copies the NT dgrad GEMMs read. */
int sconf_cast_shadows(const void* table, int64_t n_entries, int64_t total_tiles, sconf_stream_t stream);

/* qkv de-interleave "b n (h d qkv) -> qkv b n h d" + NeoX rotary on q,k (attention.py:485,498-507; rotary_emb.py:61-73).
 * bwd != 0: transpose, (dq,dk,dv) -> dqkv written to `qkv`.  cos/sin: f32 [N][D/2]. */
int sconf_rotary_qkv(int bwd, void* qkv, const float* cos_tab, const float* sin_tab, void* q, void* k, void* v,
                     int64_t B, int64_t N, int64_t H, int64_t D, int use_rotary, sconf_stream_t stream);
/* The fused path of round 2: the qkv weight SHADOW is written regrouped (sconf_cast_shadows, R < 0: row 3j + w -> w * R/3 + j), so
 * the qkv GEMM already emits (B*N, 3, H, D) = [q | k | v] and no de-interleave pass exists; apply_rotary_pos_emb
 * (rotary_emb.py:61-73) is applied in place to the q and k blocks, and its transpose inside sconf_attn_bwd (rot_cos / rot_sin). */
int sconf_rotary_inplace(void* qkv, const float* cos_tab, const float* sin_tab, int64_t B, int64_t N, int64_t H, int64_t D,
                         sconf_stream_t stream);

/* mode 0 softmax (sconformer_xl.py:242), mode 1 log_softmax (decoder.py:25); C <= 8192 classes. */
int sconf_softmax_fwd(int mode, const void* x, int x_dtype, void* y, int y_dtype, int64_t M, int64_t C, sconf_stream_t stream);
int sconf_softmax_bwd(int mode, const void* y, int y_dtype, const void* dy, int dy_dtype, void* dx, int dx_dtype,
                      float* colsum_out, float* workspace, int64_t M, int64_t C, sconf_stream_t stream);
/* colsum_out (optional, f32 [C], accumulated): column sums of dx = the bias gradient of the Linear that produced the logits, from
 * the same pass; needs `workspace` of sconf_softmax_bwd_workspace(M, C) floats. */
int64_t sconf_softmax_bwd_workspace(int64_t M, int64_t C);

/* out[n] += alpha * sum_m x[m][n]  (bias gradients). */
int sconf_colsum(const void* x, int x_dtype, float* out, int64_t M, int64_t N, int64_t ld, float alpha, sconf_stream_t stream);
/* zero rows n >= lengths[b] of x[B][N][d] in place (attention.py:511,546-547; convolution.py:109-110). */
int sconf_mask_rows(void* x, int dtype, const int32_t* lengths, int64_t B, int64_t N, int64_t d, sconf_stream_t stream);

/* Flash attention, bidirectional, head_dim 32 or 128; replaces FlashSelfAttention.forward(qkv[,key_padding_mask],
 * cu_seqlens, max_seqlen) = flash_attn_qkvpacked_func / flash_attn_varlen_qkvpacked_func (attention.py:200-257,
 * 527-535) and F.scaled_dot_product_attention (attention.py:541).  q,k,v,o: bf16 (B,N,H,D) views with element
 * strides {batch, token, head}; lengths: int32 [B] or NULL; window (-1 = unbounded); lse: f32 (B,H,N).
 * sconf_attn_bwd: delta is f32 scratch of 2*B*H*N floats (the dQ kernel, which runs first, leaves the row statistics of the
 * dK/dV kernel there: -rowsum(dO*O) and -lse*log2(e)). */
int sconf_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* lengths,
                   int64_t B, int64_t N, int64_t H, int64_t D, const int64_t* q_strides /*host*/, const int64_t* k_strides /*host*/,
                   const int64_t* v_strides /*host*/, const int64_t* o_strides /*host*/, int win_left, int win_right,
                   float scale, sconf_stream_t stream);
int sconf_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   float* delta, void* dq, void* dk, void* dv, const int32_t* lengths, int64_t B, int64_t N, int64_t H,
                   int64_t D, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                   const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                   const int64_t* dv_strides, int win_left, int win_right, float scale,
                   const float* rot_cos /*nullable: f32 (N, D/2); with rot_sin: dq, dk come back as gradients of the UNROTATED q, k*/,
                   const float* rot_sin, sconf_stream_t stream);

/* Conformer conv module, token-major (convolution.py:103-124; conv1dFunc seam convolution.py:6-22; batchrenorm.py:52-92). */
int sconf_glu_dwconv_fwd(const void* g, const int32_t* lengths, const float* w, const float* bias, void* h, double* stats,
                         int64_t B, int64_t N, int64_t d, int64_t ksize, sconf_stream_t stream);
int sconf_brn_finalize(const double* stats, int64_t count, float* running_mean, float* running_std,
                       int64_t* num_batches_tracked, const float* weight, const float* bias, float* coef, int64_t d,
                       int training, float eps, float momentum, sconf_stream_t stream);
int sconf_affine_silu_fwd(const void* h, const float* coef, void* y, int64_t M, int64_t d, sconf_stream_t stream);
int sconf_convmod_bwd(const void* dy, const void* h, const void* g, const int32_t* lengths, const float* w,
                      const float* brn_weight, const float* coef, double* red, float* bcoef, void* dg, float* dw,
                      float* dbias, float* dbrn_weight, float* dbrn_bias, float* dg_colsum, int64_t B, int64_t N, int64_t d,
                      int64_t ksize, int training, float eps, sconf_stream_t stream);

/* ConvSubsampling 'dw_striding' x8, channels-last (subsampling.py:276-321, 384-428). */
int sconf_sub_conv0_fwd(const void* x, int x_dtype, const float* w, const float* bias, void* y, int64_t B, int64_t F,
                        int64_t T, int64_t C, sconf_stream_t stream);
int sconf_sub_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int64_t B, int64_t Ti, int64_t Fi,
                         int64_t C, sconf_stream_t stream);
int sconf_sub_dwconv_bwd(const void* dout, const float* w, const void* pre_in, void* dpre_in, float* dw, float* dbias,
                         float* dpre_colsum /*nullable, f32 [C], accumulated: column sums of dpre_in = bias gradient of the 1x1 conv before it*/,
                         int64_t B, int64_t Ti, int64_t Fi, int64_t C, sconf_stream_t stream);
int sconf_sub_conv0_bwd(const void* dpre0, const void* x, int x_dtype, float* dw, float* dbias, int64_t B, int64_t F,
                        int64_t T, int64_t C, sconf_stream_t stream);
/* Fused stage 0->1 (conv0 + SiLU + first depthwise conv) without the (B,T/2,F/2,C) intermediate, and its parameter-gradient
 * backward (dw0,db0,dwd,dbd ACCUMULATED) from dd1 (B,T4,F4,C) bf16.  subsampling.py:299-318. */
int sconf_sub_stage01_fwd(const void* x, int x_dtype, const float* w0, const float* b0, const float* wd, const float* bd,
                          void* d1, int64_t B, int64_t F, int64_t T, int64_t C, sconf_stream_t stream);
int sconf_sub_stage01_bwd(const void* dd1, const void* x, int x_dtype, const float* w0, const float* b0, const float* wd,
                          float* dw0, float* db0, float* dwd, float* dbd, int64_t B, int64_t F, int64_t T, int64_t C,
                          sconf_stream_t stream);
int sconf_sub_silu_transpose(int bwd, const void* pre, const void* ds, void* out, int64_t rows, int64_t F8, int64_t C,
                             sconf_stream_t stream);

/* CTC, torch.nn.CTCLoss(blank, reduction='sum') on the (N,B,C) view (exp/train.py:104,249); batch-major input.
 * lpg/alpha/beta: f32 [B][N][2*Smax+1] workspaces; nll: f32 [B]; grad_out: f32 [B] or NULL.
 * offs: f64 [2*B*N + B] workspace written by the forward and read by the backward: alpha and beta rows are stored relative to a
 * per-frame offset (kept in f64, together with the f64 nll), which keeps the f32 lattice exact to ~1e-4 at 16384 frames where the
 * plain log-space recursion (and torch's own f32 op) is 24 % off in the gradient. */
int sconf_ctc_fwd(const float* log_probs, const int32_t* targets, const int32_t* input_lengths,
                  const int32_t* target_lengths, float* lpg, float* alpha, float* beta, double* offs, float* nll, int64_t B, int64_t N,
                  int64_t C, int64_t Smax, int blank, sconf_stream_t stream);
int sconf_ctc_bwd(const float* log_probs, const float* lpg, const float* alpha, const float* beta, const double* offs, const float* nll,
                  const int32_t* targets, const int32_t* input_lengths, const int32_t* target_lengths, const float* grad_out,
                  float* grad, int64_t B, int64_t N, int64_t C, int64_t Smax, int blank, sconf_stream_t stream);
/* The same loss taken from the decoder's LOGITS (f32 (B,N,C)): decoder.py:25 F.log_softmax + torch.nn.CTCLoss (exp/train.py:104,249)
 * as one operator.  The forward folds log_softmax into the emission gather and keeps the row log-sum-exp (lse, f32 (B,N)); the
 * backward returns d nll / d logits in bf16 (the CTC gradient through log_softmax) and, optionally, its column sums ACCUMULATED
 * into colsum_out [C] (the decoder bias gradient; needs sconf_ctc_bwd_logits_workspace(B*N, C) floats of scratch).  Neither the
 * log-probabilities nor their gradient exist as (B,N,C) tensors. */
int sconf_ctc_fwd_logits(const float* logits, const int32_t* targets, const int32_t* input_lengths, const int32_t* target_lengths,
                         float* lse, float* lpg, float* alpha, float* beta, double* offs, float* nll,
                         int64_t B, int64_t N, int64_t C, int64_t Smax, int blank, sconf_stream_t stream);
int64_t sconf_ctc_bwd_logits_workspace(int64_t rows, int64_t C);
int sconf_ctc_bwd_logits(const float* logits, const float* lse, const float* lpg, const float* alpha, const float* beta,
                         const double* offs, const float* nll, const int32_t* targets, const int32_t* input_lengths,
                         const int32_t* target_lengths, const float* grad_out, void* dlogits_bf16, float* colsum_out, float* workspace,
                         int64_t B, int64_t N, int64_t C, int64_t Smax, int blank, sconf_stream_t stream);


/* ---- forward-only inference helpers ------------------------------------------------------------------
 * Overlap-average of sliding-window posteriors, fetch_logits (lcasr/eval/utils.py:45-111): for W equally long windows
 * logp (W,n,C) f32, window w starting at output row pos0 + w*stride:  acc (N,C) += sum of exp(logp) over the windows covering
 * each row, count (N) += how many cover it (both accumulated, so a ragged last window is a second call with W = 1);
 * sconf_overlap_finalize writes out = log(acc / count) for the first N rows (utils.py:107-111). */
int sconf_overlap_add_exp(const float* logp, int64_t W, int64_t n, int64_t C, int64_t stride, int64_t pos0, float* acc,
                          float* count, int64_t N, sconf_stream_t stream);
int sconf_overlap_finalize(const float* acc, const float* count, float* out, int64_t N, int64_t C, sconf_stream_t stream);
/* idx[m] = argmax_c x[m][c], first index on ties: GreedyCTCDecoder.forward (lcasr/decoding/greedy.py:19). */
int sconf_argmax_rows(const float* x, int64_t M, int64_t C, int32_t* idx, sconf_stream_t stream);

/* Fused MADGRAD + global-norm clip over flat f32 buffers (lcasr/optim/madgrad.py:81-212, exp/train.py:46-61). */
int sconf_sumsq(const float* g, int64_t n, double* out, sconf_stream_t stream);
/* k = steps applied so far: from the host, or read from k_dev (device int64) when k_dev != NULL.  At k == 0 the kernel creates
 * x0 := p (the reference creates its state lazily at the first step, madgrad.py:121-125).  A step whose gradient norm is not
 * finite is skipped; sconf_madgrad_advance (once per optimiser step, after the per-group launches) then leaves *k_dev alone,
 * as GradScaler.step skips optimizer.step() in exp/train.py:54-57. */
int sconf_madgrad_step(float* p, const float* g, float* grad_sum_sq, float* s, float* x0, void* bf16_shadow,
                       int64_t n, const double* sumsq, float max_norm, float grad_scale, float lr, float momentum,
                       float eps, float weight_decay, int64_t k, const int64_t* k_dev, sconf_stream_t stream);
int sconf_madgrad_advance(int64_t* k_dev, const double* sumsq, float grad_scale, sconf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SCONF_H */
