/* acmil_hip.h -- C ABI of libacmil_hip.so: MI355X (gfx950) kernels for ACMIL's per-slide
 * gated-attention aggregation path.
 *
 * The reference (dazhangyu123/ACMIL) has no FFI / plugin layer: its replaceable unit is the
 * torch.nn.Module.  These entry points are what a maintainer would bind from
 * `architecture/transformer.py` with ctypes; each one cites the reference code it
 * replaces.  Conventions:
 *   - every pointer is a DEVICE pointer unless marked "host"; the caller (PyTorch's caching allocator in
 *     the shipped host code) owns every buffer including `packed` and `workspace`; the library allocates
 *     nothing, keeps no global state and reads no environment variable, so it is re-entrant across threads
 *     and streams.  ONE documented exception: acmil_transmil_forward owns a per-device side stream + event
 *     pair behind a lock (a convenience wrapper; acmil_transmil_forward_ex takes caller-owned ones and has
 *     no exception).  Measurement switches exist only in the A/B build libacmil_hip_ab.so (csrc/ab_knobs.h);
 *   - `stream` is the caller's hipStream_t (0 = default stream); all work is enqueued on it, nothing
 *     synchronises the device;
 *   - return value: ACMIL_OK, or a negative ACMIL_ERR_* for a bad shape / unsupported configuration /
 *     launch failure.  No exceptions, no aborts.  Asynchronous faults surface at the caller's next sync;
 *   - all matrices are row-major, contiguous, fp32 unless a dtype argument says otherwise.
 */
#ifndef ACMIL_HIP_H
#define ACMIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACMIL_OK 0
#define ACMIL_ERR_SHAPE (-1)       /* a dimension is out of range / not a supported multiple */
#define ACMIL_ERR_UNSUPPORTED (-2) /* valid request this build has no kernel for */
#define ACMIL_ERR_NULL (-3)        /* a required pointer is NULL */
#define ACMIL_ERR_LAUNCH (-4)      /* hipLaunchKernel reported an error */
#define ACMIL_ERR_ARCH (-5)        /* current device is not gfx950 */

/* arithmetic mode of the two projection GEMMs (everything else is always fp32) */
#define ACMIL_MODE_F32 0   /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): bitwise an fp32 fmaf chain        */
#define ACMIL_MODE_F16X3 1 /* fp32-parity split: operands as f16 hi+lo, 3 f16 MFMAs per product, fp32 acc */
#define ACMIL_MODE_F16 2   /* throughput: single f16 MFMA pass (NOT within the 1e-4 fp32 parity bound)     */

/* element type of the bag matrix x[N,D] in HBM */
#define ACMIL_DTYPE_F32 0
#define ACMIL_DTYPE_F16 1
#define ACMIL_DTYPE_BF16 2

#define ACMIL_MAX_TOKENS 16 /* K = n_token supported by this build (Step3_WSI_classification_ACMIL.py:39 --n_token; transformer.py:292-301) */
#define ACMIL_MAX_TOKENS_FUSED 5 /* ... by the single-kernel forward families and the one-call training step; K above runs the composed kernels */
#define ACMIL_MAX_CLASSES 16
#define ACMIL_MAX_BATCH 64   /* bags per acmil_ga_forward_batch / _guarded launch */

/* Library / build identification; returns a static string. */
const char* acmil_version(void);

/* 0 if the current HIP device is gfx950, else ACMIL_ERR_ARCH.  (host-side query; no kernel launch) */
int acmil_check_device(void);

/* Measurement aid of bench.py's roofline line (no reference counterpart): a launch that issues ONLY the matrix instruction of the
 * split-f16 kernels -- iters x 8 independent v_mfma_f32_32x32x16_f16 per wave on pseudo-random f16 operands, two 4-wave workgroups
 * per CU (the headline kernel's residency), no loads, no LDS, no barriers.  Its rate under the socket's power cap is the ceiling any
 * split-f16 kernel of this library can reach on THIS box at THIS moment; bench.py times it with events and reports it next to the
 * nominal 2.5 PF.  sink: >= workgroups * 256 floats (written so that the loop cannot be removed); returns the number of MFMA
 * instructions the launch issues in *mfmas (host pointer, may be NULL). */
int acmil_mfma_probe(int iters, int workgroups, float* sink, long long* mfmas, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight packing.  Rearranges the parameters of ACMIL_GA / ABMIL
 *   dimreduction.fc1.weight            W1 [Di,D]            (architecture/network.py:40, no bias)
 *   attention.attention_V.0.{weight,bias}  Wv [Da,Di], bv [Da]  (architecture/transformer.py:247-250)
 *   attention.attention_U.0.{weight,bias}  Wu [Da,Di], bu [Da]  (:252-255)
 *   attention.attention_weights.{weight,bias}  Ww [K,Da], bw [K] (:257)
 *   classifier.{i}.fc.{weight,bias}    Wc[i] [C,Di], bc[i] [C], i<K   (transformer.py:295-297)
 *   Slide_classifier.fc.{weight,bias}  Ws [C,Di], bs [C]    (:301)   (NULL for ABMIL: no bag head)
 * into the MFMA-fragment-ordered stream the forward kernel consumes.  Must be re-run whenever the
 * parameters change (every optimiser step); it is one small launch.  `Wc`/`bc` are HOST arrays of K
 * device pointers.  Da must be 128 (the reference's fixed ctor default D=128, transformer.py:292).
 * ------------------------------------------------------------------------------------------- */
size_t acmil_ga_packed_bytes(int D, int Di, int Da, int K, int C, int mode);

int acmil_ga_pack_weights(const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu,
                          const float* Ww, const float* bw, const float* const* Wc, const float* const* bc,
                          const float* Ws, const float* bs, int D, int Di, int Da, int K, int C, int mode,
                          void* packed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused forward of one bag (no mask: eval, or the score pass of a training step).  Replaces
 * ACMIL_GA.forward / forward_feature / ABMIL.forward for one slide
 * (architecture/transformer.py:305-330, :332-352, :277-286; DimReduction network.py:49-57;
 * Attention_Gated transformer.py:259-267; Classifier_1fc network.py:14-19):
 *   h = relu(x W1^T); A = ((tanh(h Wv^T+bv) * sigmoid(h Wu^T+bu)) Ww^T + bw)^T          [K,N]
 *   P = softmax_N(A); afeat = P h [K,Di]; sub_preds[k] = Wc[k] afeat[k] + bc[k]          [K,C]
 *   bag_feat = mean_k afeat [Di]; slide_pred = Ws bag_feat + bs                          [C]
 * x: [N,D] of x_dtype (the reference's x[0]; B must be 1).  Outputs, each may be NULL to skip:
 *   A_out [K,N] raw scores (the reference's third return value without its leading 1),
 *   sub_preds [K,C], slide_pred [C] (only if has_bag_head), afeat [K,Di], bag_feat [Di],
 *   h_save [N,Di] fp32 (kept for the masked pooling pass and the backward of a training step).
 * If sub_preds, slide_pred, afeat and bag_feat are all NULL only the scores (and h_save) are produced.
 * ABMIL = K 1, has_bag_head 0: its logits are sub_preds[0].
 * workspace: acmil_ga_workspace_bytes(...) bytes, 256-byte aligned (may be NULL for a scores-only call).  The FIRST 256
 *   bytes of every GA workspace (acmil_ga_forward / _forward_batch / _pool / _train_step) are a control block of 32-bit words:
 *   0 tile counter, 1 range status of the most recent split-f16 launch (0 = every projected feature was finite and inside the
 *   f16 range -- which a bag value outside that range never leaves it: its f16 hi half is inf and poisons every feature of
 *   its patch; non-zero -- repeat the call with ACMIL_MODE_F32), 2.. internal counters.  Zero the block ONCE after allocating the workspace (hipMemset); every launch
 *   leaves its counters at zero, so no memset is needed between launches.  One workspace serves one stream.
 * ------------------------------------------------------------------------------------------- */
size_t acmil_ga_workspace_bytes(int N, int D, int Di, int K, int C, int mode);

/* Zero the control block of a freshly allocated GA workspace (enqueued on `stream`): call ONCE per allocation, before the first
 * acmil_ga_forward / _forward_batch / _guarded / _pool / _train_step on it.  (Equivalent to hipMemsetAsync(workspace, 0, 256, stream);
 * provided so that the one-off initialisation the contract above asks for is an explicit call of this library.) */
int acmil_ga_workspace_init(void* workspace, void* stream);

int acmil_ga_forward(const void* x, int x_dtype, int N, const void* packed, int D, int Di, int Da, int K, int C,
                     int mode, float* A_out, float* sub_preds, float* slide_pred, float* afeat, float* bag_feat,
                     float* h_save, int has_bag_head, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched eval forward: up to ACMIL_MAX_BATCH (64) bags (different N allowed, same weights) in ONE launch of the fused kernel, one
 * merge and one heads launch.  Same maths per bag as acmil_ga_forward; exists because a single 50 000-patch bag
 * only occupies 196 of the 256 CUs (tile quantisation) -- a batch keeps all of them busy.  xs / A_outs: HOST arrays
 * of nbags device pointers (A_outs or its entries may be NULL); Ns: HOST array; sub_preds [B,K,C],
 * slide_pred [B,C], afeat [B,K,Di], bag_feat [B,Di] (each may be NULL).
 * ------------------------------------------------------------------------------------------- */
size_t acmil_ga_batch_workspace_bytes(int nbags, const int* Ns, int D, int Di, int K, int C, int mode);

int acmil_ga_forward_batch(int nbags, const void* const* xs, const int* Ns, int x_dtype, const void* packed, int D, int Di,
                           int Da, int K, int C, int mode, float* const* A_outs, float* sub_preds, float* slide_pred,
                           float* afeat, float* bag_feat, int has_bag_head, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched eval forward with the DEVICE-SIDE range guard (no counterpart in the reference, whose fp32 arithmetic has no
 * range to leave: architecture/transformer.py:305-330 is what both launches compute).  Enqueues, on one stream: the
 * split-f16 fused launch (packed_f16x3 = acmil_ga_pack_weights(..., ACMIL_MODE_F16X3)), then an exact-fp32 launch over the
 * same tiles (packed_fp32 = ... ACMIL_MODE_F32) that exits at once unless the first one's status word is non-zero and
 * otherwise overwrites scores and partials, then merge + heads.  The caller never has to read the status word back:
 * the outputs are the fp32-parity result either way.  fallback_count (device, may be NULL): incremented once per
 * call whose fp32 launch ran.  Same arguments / workspace as acmil_ga_forward_batch otherwise.
 * ------------------------------------------------------------------------------------------- */
int acmil_ga_forward_guarded(int nbags, const void* const* xs, const int* Ns, int x_dtype, const void* packed_f16x3,
                             const void* packed_fp32, int D, int Di, int Da, int K, int C, float* const* A_outs,
                             float* sub_preds, float* slide_pred, float* afeat, float* bag_feat, int has_bag_head,
                             unsigned* fallback_count, void* workspace, void* stream);

/* The same device-side guard for the WIDE families (D_inner 384 / 512: architecture/transformer.py:305-330 on the feature extractors of
 * Step3_WSI_classification_ACMIL.py:78-87), one bag per call: the fused split-f16 launch, then its exact-fp32 repeat OP BY OP
 * ([widen a 16-bit bag] -> h = relu(x W1^T) -> gated scores -> pooling partials), every launch of it predicated on the status word of
 * the fused launch, then merge + heads.  W1 [Di, D] = dimreduction.fc1.weight in raw fp32 (the other operands of the repeat are the raw
 * copies inside `packed_f16x3`); scratch: caller-owned, 256-byte aligned, acmil_ga_forward_guarded_wide_scratch_bytes (need_scores = 1 if
 * A_out is NULL); fallback_count as acmil_ga_forward_guarded. */
size_t acmil_ga_forward_guarded_wide_scratch_bytes(int N, int D, int Di, int K, int x_dtype, int need_scores);

int acmil_ga_forward_guarded_wide(const void* x, int x_dtype, int N, const void* packed_f16x3, const float* W1, int D, int Di, int Da,
                                  int K, int C, float* A_out, float* sub_preds, float* slide_pred, float* afeat, float* bag_feat,
                                  int has_bag_head, unsigned* fallback_count, void* scratch, void* workspace, void* stream);

/* The predicated repeat for the COMPOSED path (D_inner = 768, n_token > 5: acmil_linear_f16x3 -> acmil_gated_scores_packed -> acmil_ga_pool
 * as separate calls; Step3_WSI_classification_ACMIL.py:78-87, :39): if -- and only if -- *cond != 0 (cond = the range status word the
 * projection launch left in its workspace), h [N, Di] and A [K, N] are OVERWRITTEN with their exact-fp32 values (h = relu(x W1^T),
 * A = gated scores on the raw fp32 copies of [Wv; Wu], [bv; bu], Ww, bw inside `packed` = acmil_ga_pack_weights(..., mode)); the caller's
 * pooling / merge / heads then run on whichever values are there.  W1 [Di, D] raw fp32.  Eval forward only.  scratch: 256-byte aligned, acmil_ga_rescore_fp32_cond_scratch_bytes (Da = 128). */
size_t acmil_ga_rescore_fp32_cond_scratch_bytes(int N, int D, int Di, int x_dtype);

int acmil_ga_rescore_fp32_cond(const void* x, int x_dtype, int N, const void* packed, const float* W1, int D, int Di, int Da, int K, int C,
                               int mode, float* h, float* A, const unsigned* cond, unsigned* fallback_count, void* scratch, void* stream);



/* ---------------------------------------------------------------------------------------------
 * Masked pooling pass of a training step.  Replaces transformer.py:318-330 given the scores and h of the
 * score pass: A[k, masked_idx[k,:]] = -1e9 (written in place into A, which then IS the reference's A_out),
 * P = softmax_N(A), afeat = P h, heads as above.  masked_idx [K,n_masked] int64 (device) from
 * acmil_stkim_select; n_masked may be 0 (no mask: identical maths to the fused forward).
 * ------------------------------------------------------------------------------------------- */
int acmil_ga_pool(const float* h, float* A, int N, const void* packed, int D, int Di, int Da, int K, int C, int mode,
                  const int64_t* masked_idx, int n_masked, float* sub_preds, float* slide_pred, float* afeat,
                  float* bag_feat, int has_bag_head, void* workspace, void* stream);

/* The unmasked pooling pass + heads for a GROUP of nbags <= 16 bags whose rows lie back to back (h [N, Di] and the raw scores
 * A [K, N] cover all bags, N = sum rows[b]; rows: HOST array).  Per bag exactly acmil_ga_pool with n_masked = 0 (transformer.py:322-330;
 * the reference is strictly one slide per call, Step3_WSI_classification_ACMIL.py:253-258 -- this is the batched eval of the composed
 * families, where acmil_linear_f16x3 and acmil_gated_scores_packed already ran over all rows at once).  A is not modified.
 * Outputs per bag: sub_preds [nbags, K, C], slide_pred [nbags, C], afeat [nbags, K, Di], bag_feat [nbags, Di] (each may be NULL).
 * workspace: acmil_ga_pool_group_workspace_bytes, initialised once with acmil_ga_workspace_init. */
size_t acmil_ga_pool_group_workspace_bytes(int N, int nbags, int Di, int K);

int acmil_ga_pool_group(const float* h, const float* A, int N, int nbags, const int* rows, const void* packed, int D, int Di, int Da,
                        int K, int C, int mode, float* sub_preds, float* slide_pred, float* afeat, float* bag_feat, int has_bag_head,
                        void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * STKIM selection.  Replaces transformer.py:314-317: top-k of each branch's scores (sorted by descending
 * score, ties broken towards the LOWER index), then the first m = int(k*mask_drop) columns of
 * argsort(uniforms[K,k]) pick which of them are masked.  `uniforms` are the caller's U[0,1) draws (the
 * reference draws torch.rand(K,k)); injecting them keeps the op deterministic and testable.
 *   scores [K,N] fp32, k = min(n_masked_patch, N) <= 64, topk_idx [K,k] int64, masked_idx [K,m] int64.
 * ------------------------------------------------------------------------------------------- */
size_t acmil_stkim_workspace_bytes(int N, int K, int k);

int acmil_stkim_select(const float* scores, int N, int K, int k, int m, const float* uniforms,
                       int64_t* topk_idx, int64_t* masked_idx, void* workspace, void* stream);

/* The same selection with the uniforms drawn ON THE DEVICE when `uniforms` is NULL: Philox4x32-10 keyed on (seed, offset, branch,
 * column), 24 random bits per draw -- the production form of `torch.rand(K, k)` (architecture/transformer.py:316: any iid U[0,1)
 * stream is the same distribution; the caller advances `offset` once per forward).  With `uniforms` given it is acmil_stkim_select. */
int acmil_stkim_select_rng(const float* scores, int N, int K, int k, int m, const float* uniforms, unsigned long long seed,
                           unsigned long long offset, int64_t* topk_idx, int64_t* masked_idx, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Exact-fp32 matrix-core GEMM (v_mfma_f32_32x32x2_f32), row-major, optional batch and transpose:
 *   C[b] = act( alpha * op(A[b]) * op(B[b]) + bias[col] + beta * C[b] ),  op(A) M x K, op(B) K x N.
 * Stands in for the aten::mm / addmm / bmm calls of the reference's backward (autograd of
 * architecture/transformer.py:305-330) and of TransMIL (architecture/transMIL.py, nystrom_attention.py).
 * B may be fp32/fp16/bf16 (b_dtype).  act: 0 none, 1 relu, 2 relu-backward mask by aux (same layout as C),
 * 3 C = beta I - alpha P (beta = diagonal constant), 4 C = alpha P and aux (WRITTEN, same layout as C) = beta I - alpha P.
 * Tall-K products are split along K into `workspace` (acmil_gemm_workspace_bytes) and reduced in a fixed order.
 * ------------------------------------------------------------------------------------------- */
size_t acmil_gemm_workspace_bytes(int M, int N, int K, int batch);

int acmil_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                   long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta, float* C,
                   int ldc, long long strideC, const float* bias, int act, const float* aux, int batch,
                   void* workspace, void* stream);

/* Same contract, split-f16 arithmetic ("f16x3", v_mfma_f32_32x32x16_f16): each fp32 operand is split hi + lo in f16 and
 * the product formed as hi*hi + lo*hi + hi*lo with fp32 accumulation -- relative error ~1e-6 for operands inside the f16
 * range (|v| < 65504; larger values turn into inf), 2-4x the throughput of the exact kernel.  Used for the nn.Linear
 * products x W^T (network.py:49-57 / transMIL.py:51,62 / nystrom_attention.py:55,59) and for the weight / input gradient
 * products of the backward. */
int acmil_gemm_f16x3(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                     long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta, float* C,
                     int ldc, long long strideC, const float* bias, int act, const float* aux, int batch,
                     void* workspace, void* stream);

/* Same contract with bf16 halves ("bf16x3", v_mfma_f32_32x32x16_bf16): 16 mantissa bits (relative error ~1e-5) but the
 * full fp32 exponent range -- for the gradient products of the backward, whose operands reach 1e-8. */
int acmil_gemm_bf16x3(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                      long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta, float* C,
                      int ldc, long long strideC, const float* bias, int act, const float* aux, int batch,
                      void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of one training step of ACMIL_GA (the autograd graph of architecture/transformer.py:305-330,
 * reference has no explicit backward code).  Inputs: the bag x, h [N,Di] and the masked scores A_out [K,N]
 * kept by the forward, afeat [K,Di]; the parameters (raw fp32 tensors); incoming gradients d_sub [K,C],
 * d_slide [C] (NULL without bag head) and d_A [K,N] (gradient w.r.t. the returned attention scores, NULL if
 * unused).  Outputs: every parameter gradient (overwritten, not accumulated):
 *   dW1 [Di,D], dWv [Da,Di], dbv [Da], dWu [Da,Di], dbu [Da], dWw [K,Da], dbw [K],
 *   dWc[k] [C,Di], dbc[k] [C] (HOST arrays of K device pointers), dWs [C,Di], dbs [C].
 * Masked positions (A_out == -1e9) receive zero gradient, as masked_fill does.  No dx (the bag has no grad).
 * ------------------------------------------------------------------------------------------- */
size_t acmil_ga_backward_workspace_bytes(int N, int D, int Di, int K, int C);

int acmil_ga_backward(const void* x, int x_dtype, int N, const float* h, const float* A_out, const float* afeat,
                      const float* Wv, const float* bv, const float* Wu, const float* bu, const float* Ww,
                      const float* const* Wc, const float* Ws, const float* d_sub, const float* d_slide,
                      const float* d_A, float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw,
                      float* dbw, float* const* dWc, float* const* dbc, float* dWs, float* dbs, int D, int Di, int Da,
                      int K, int C, int mode, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One ACMIL_GA training step for one slide in ONE call: forward with STKIM masking (architecture/transformer.py:305-330,
 * training branch), the ACMIL loss (Step3_WSI_classification_ACMIL.py:201-216) and the backward through both (what
 * loss.backward() computes, Step3_WSI_classification_ACMIL.py:217-218) -- 8 launches enqueued by one host call.
 *   packed   acmil_ga_packed_bytes(...) device buffer; rewritten from the raw parameters when repack != 0 (call with
 *            repack = 1 whenever the parameters changed since the last step, e.g. after every optimizer step)
 *   W*, b*   raw fp32 parameters (as acmil_ga_pack_weights); d*: their gradients, OVERWRITTEN (not accumulated).
 *            [dWv; dWu] / [Wv; Wu] adjacent in memory (flat buckets) are used in place
 *   label    [1] int64 on the device; uniforms [K, k_top] fp32 (the torch.rand draw of transformer.py:314), may be
 *            NULL when m_mask == 0; k_top = min(n_masked_patch, N), m_mask = int(k_top * mask_drop); k_top = 0: no masking
 *   outputs  losses [4] = {loss0, loss1, diff_loss, total}; sub_preds [K,C]; slide_pred [C]; A_out [K,N] = the masked
 *            raw scores the reference returns as `attn`; topk_idx [K,k_top], masked_idx [K,m_mask] int64
 *   workspace acmil_ga_train_step_workspace_bytes(...) bytes.  Its first 256 bytes are the GA control block: zero them once
 *            after allocation; word 1 (int32) is the split-f16 range status of this step (0 = in range; otherwise the
 *            caller repeats the step with mode = ACMIL_MODE_F32).
 *   guard_flag (device float, may be NULL): receives 1.0f when that status is non-zero, else 0.0f -- written on the stream, so
 *            it can be handed to acmil_adamw_step's skip_flag (and all-reduced) without the host reading anything.
 * ------------------------------------------------------------------------------------------- */
size_t acmil_ga_train_step_workspace_bytes(int N, int D, int Di, int K, int C, int k_top);

int acmil_ga_train_step(const void* x, int x_dtype, int N, void* packed, int repack,
                        const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu,
                        const float* Ww, const float* bw, const float* const* Wc, const float* const* bc,
                        const float* Ws, const float* bs,
                        float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                        float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                        int D, int Di, int Da, int K, int C, int mode,
                        const int64_t* label, const float* uniforms, int k_top, int m_mask,
                        float* losses, float* sub_preds, float* slide_pred, float* A_out,
                        int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream);

/* acmil_ga_train_step with the STKIM uniforms drawn on the device when `uniforms` is NULL (see acmil_stkim_select_rng): no
 * torch.rand launch per step; a repeat of the step (the fp32 re-run of the range guard) with the same (seed, offset) masks the
 * same patches. */
int acmil_ga_train_step_rng(const void* x, int x_dtype, int N, void* packed, int repack,
                        const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu,
                        const float* Ww, const float* bw, const float* const* Wc, const float* const* bc,
                        const float* Ws, const float* bs,
                        float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                        float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                        int D, int Di, int Da, int K, int C, int mode,
                        const int64_t* label, const float* uniforms, int k_top, int m_mask,
                        float* losses, float* sub_preds, float* slide_pred, float* A_out,
                        int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream,
                            unsigned long long rng_seed, unsigned long long rng_offset);

/* One training step INCLUDING the optimizer, for a single-GPU run: acmil_ga_train_step_rng + torch.optim.AdamW's update
 * (Step3_WSI_classification_ACMIL.py:139 builds it, :219 steps it once per slide) + the re-pack of the updated weights, with the
 * step's LAST launch doing all three closing jobs at once (split-K finish of the weight gradients, AdamW on every parameter, the
 * new values written to every place of `packed` that holds them).  Replaces three launches at the end of a step and the pack
 * launch at the start of the next one; results (gradients, parameters, moments, packed buffer) are bit-identical to
 * acmil_ga_train_step_rng -> acmil_adamw_step_report -> acmil_ga_pack_weights.
 *   W1 .. bs        the parameters, NOT const here: views of one flat fp32 buffer flat_params [n_flat] that they cover exactly
 *   exp_avg, exp_avg_sq [n_flat]  AdamW moments, element i belongs to flat_params[i]
 *   lr .. weight_decay, step, skipped, flag_report   as acmil_adamw_step_report; the skip flag is guard_flag (NULL: never skipped)
 *   repack          1: `packed` is rebuilt first (first step, or the parameters were changed by anything but this entry)
 * A skipped step (range flag set) leaves parameters, moments and `packed` untouched.
 * Returns ACMIL_ERR_UNSUPPORTED before anything is launched when mode != ACMIL_MODE_F16X3 or W1 / Wv / Wu (their gradients,
 * their moments) are not 16-byte aligned; ACMIL_ERR_SHAPE when the parameters do not tile flat_params.  Data-parallel runs keep
 * the separate entries (their all-reduce sits between the gradients and the update). */
int acmil_ga_train_step_adamw(const void* x, int x_dtype, int N, void* packed, int repack,
                        float* W1, float* Wv, float* bv, float* Wu, float* bu, float* Ww, float* bw, float* const* Wc,
                        float* const* bc, float* Ws, float* bs,
                        float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                        float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                        int D, int Di, int Da, int K, int C, int mode,
                        const int64_t* label, const float* uniforms, int k_top, int m_mask,
                        float* losses, float* sub_preds, float* slide_pred, float* A_out,
                        int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream,
                        unsigned long long rng_seed, unsigned long long rng_offset,
                        const float* flat_params, long long n_flat, float* exp_avg, float* exp_avg_sq, float lr, double beta1,
                        double beta2, float eps, float weight_decay, long long step, int* skipped, float* flag_report);

/* Would the closing launch of acmil_ga_train_step_adamw / acmil_ga_train_step_group (adamw != NULL) / acmil_ga_adamw_pack accept this
 * parameter set?  A query: nothing is launched.  ACMIL_OK, or the code those entries return before their first launch
 * (ACMIL_ERR_UNSUPPORTED: mode != ACMIL_MODE_F16X3, or W1 / Wv / Wu, their gradients or moments not 16-byte aligned; ACMIL_ERR_SHAPE: the
 * parameters do not tile flat_params).  Callers ask once per parameter set and treat every error of the step itself as a failure. */
int acmil_ga_adamw_supported(const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu, const float* Ww,
                             const float* bw, const float* const* Wc, const float* const* bc, const float* Ws, const float* bs,
                             const float* dW1, const float* dWv, const float* dWu, int D, int Di, int Da, int K, int C, int mode,
                             const float* flat_params, long long n_flat, const float* exp_avg, const float* exp_avg_sq);

/* A GROUP of bags in ONE training step: the single-GPU twin of slide-level data parallelism.  The reference trains with B = 1
 * (Step3_WSI_classification_ACMIL.py:189-221: one slide, one backward, one optimizer.step()); G data-parallel ranks average the
 * gradients of G slides per step.  This entry computes exactly that average on ONE GPU -- per slide the forward with
 * STKIM, the three losses and the backward of acmil_ga_train_step, the parameter gradients = the MEAN over the bags -- with the
 * patch-parallel work (score pass, both weight-gradient products, the closing launch) batched over all bags: 7 launches per G
 * slides instead of per slide.  Under data parallelism it is `bags per rank`: one all-reduce per G slides.
 *   x          the bags' rows BACK TO BACK: [sum bag_rows][D] (x_dtype as acmil_ga_train_step), bag b = rows
 *              sum(bag_rows[:b]) .. ; bag_rows [nbags] on the HOST, 1 <= nbags <= 16, every bag >= max(1, k_top) rows
 *   labels     [nbags] int64 (device); uniforms [nbags][K][k_top] or NULL (device draw, Philox keyed on (seed, offset, bag, branch, column))
 *   outputs    losses [nbags][4]; sub_preds [nbags][K][C]; slide_pred [nbags][C]; A_out [K][sum bag_rows] (bag b = its column range;
 *              masked raw scores); topk_idx [nbags][K][k_top], masked_idx [nbags][K][m_mask] (indices LOCAL to the bag)
 *   gradients  dW1 .. dbs = mean over the bags of the per-slide gradients
 *   adamw      NULL: the gradients are left for the caller (data parallel: all-reduce, then acmil_ga_adamw_pack / acmil_adamw_step);
 *              else: the step's closing launch applies AdamW and re-packs, as acmil_ga_train_step_adamw does (parameters must
 *              tile adamw->flat_params; same refusals before anything is launched)
 *   workspace  acmil_ga_train_step_group_workspace_bytes(...); control block and range status word as acmil_ga_train_step (ONE status /
 *              guard_flag for the group: a flagged group is repeated bag by bag in ACMIL_MODE_F32 by the caller)
 * mode must be ACMIL_MODE_F16X3 (or _F16), D_inner 128 / 256, K <= 5: ACMIL_ERR_UNSUPPORTED otherwise.  nbags == 1 is
 * acmil_ga_train_step_rng / _adamw bit for bit. */
typedef struct acmil_adamw_args {
    const float* flat_params; long long n_flat; float* exp_avg; float* exp_avg_sq;
    float lr; double beta1, beta2; float eps, weight_decay; long long step; int* skipped; float* flag_report;
} acmil_adamw_args;

size_t acmil_ga_train_step_group_workspace_bytes(int nbags, int N_total, int D, int Di, int K, int C, int k_top);

int acmil_ga_train_step_group(const void* x, int x_dtype, int nbags, const int* bag_rows, void* packed, int repack,
                        float* W1, float* Wv, float* bv, float* Wu, float* bu, float* Ww, float* bw, float* const* Wc,
                        float* const* bc, float* Ws, float* bs,
                        float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                        float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                        int D, int Di, int Da, int K, int C, int mode,
                        const int64_t* labels, const float* uniforms, int k_top, int m_mask,
                        float* losses, float* sub_preds, float* slide_pred, float* A_out,
                        int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream,
                        unsigned long long rng_seed, unsigned long long rng_offset, const acmil_adamw_args* adamw);

/* torch.optim.AdamW's update over the flat parameter buffer of ONE ACMIL_GA / ABMIL module AND the re-pack of its weights in one
 * launch: acmil_adamw_step_report followed by acmil_ga_pack_weights, for steps whose gradients are final when the optimizer runs --
 * data-parallel training (Step3_WSI_classification_ACMIL.py:219's optimizer.step() behind a gradient all-reduce): the next
 * acmil_ga_train_step then passes repack = 0.  Arguments as acmil_ga_train_step_adamw (parameters tile flat_params; moments at the
 * same offsets; skip_flag / skipped / flag_report as acmil_adamw_step_report); mode must be ACMIL_MODE_F16X3.  Same error behaviour:
 * ACMIL_ERR_UNSUPPORTED / ACMIL_ERR_SHAPE before anything is launched. */
int acmil_ga_adamw_pack(float* W1, float* Wv, float* bv, float* Wu, float* bu, float* Ww, float* bw, float* const* Wc, float* const* bc,
                        float* Ws, float* bs,
                        const float* dW1, const float* dWv, const float* dbv, const float* dWu, const float* dbu, const float* dWw,
                        const float* dbw, const float* const* dWc, const float* const* dbc, const float* dWs, const float* dbs,
                        int D, int Di, int Da, int K, int C, int mode, void* packed,
                        const float* flat_params, long long n_flat, float* exp_avg, float* exp_avg_sq, float lr, double beta1,
                        double beta2, float eps, float weight_decay, long long step, const float* skip_flag, int* skipped,
                        float* flag_report, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused ACMIL loss + its gradient w.r.t. the aggregator outputs.  Replaces Step3_WSI_classification_ACMIL.py:201-216
 * (loss0 = CE(sub_preds, label x K) [0 if K == 1], loss1 = CE(slide_pred, label), diff_loss = mean pairwise cosine
 * similarity of softmax_N(attn) rows) and the first autograd step through them.
 *   sub_preds [K,C], slide_pred [C] (NULL without bag head), A_out [K,N] (masked scores), label [1] int64 (device);
 *   losses [4] = {loss0, loss1, diff_loss, total}; d_sub [K,C], d_slide [C], d_A [K,N] = d total / d (.)
 * ------------------------------------------------------------------------------------------- */
size_t acmil_ga_loss_workspace_bytes(int N, int K);

int acmil_ga_loss(const float* sub_preds, const float* slide_pred, const float* A_out, const int64_t* label, int N, int K,
                  int C, float* losses, float* d_sub, float* d_slide, float* d_A, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * TransMIL forward (eval) of one bag.  Replaces TransMIL.forward architecture/transMIL.py:60-91 with its
 * TransLayer :25-28, PPEG :38-45 and NystromAttention (pip nystrom_attention 0.0.12; vendored stand-in
 * architecture/nystrom_attention.py:67-149, pinv :12-27).  heads = 8, dim_head = Di/8, landmarks = Di/2,
 * 6 pinv iterations, residual conv 33 (the reference's fixed ctor arguments, transMIL.py:13-23).
 *   x [N,D] fp32 (B = 1); fc1_w [Di,D], fc1_b [Di]; cls_token [Di];
 *   layer1 / layer2: HOST arrays of 6 device pointers {norm.weight, norm.bias, attn.to_qkv.weight [3Di,Di],
 *                    attn.to_out.0.weight [Di,Di], attn.to_out.0.bias, attn.res_conv.weight [8,33]};
 *   ppeg: HOST array of 6 device pointers {proj.weight [Di,7,7], proj.bias, proj1.weight [Di,5,5], proj1.bias,
 *         proj2.weight [Di,3,3], proj2.bias};  norm_w/norm_b [Di]; fc2_w [C,Di], fc2_b [C].
 * Output logits [C].  dbg_h1 / dbg_hp / dbg_h2: NULL, or [(side^2+1), Di] buffers receiving the token matrix
 * after layer1 / PPEG / layer2 (parity tests).  Dropout (train mode) is not implemented: eval forward only.
 * Streams: everything is ordered on `stream` as far as the caller can tell -- but inside a layer the Moore-Penrose chain runs on ONE
 * library-owned non-blocking stream per device beside the attn3 leg, forked from and joined back into `stream` with events before the
 * call's last launches are enqueued (so `stream` alone orders the outputs; HIP-graph capture of the call works; concurrent callers are
 * serialised over the enqueue by a lock).  Forwards on different streams may overlap on the GPU (tested: tools/stress_transmil.py);
 * the workspace may hold anything on entry.
 * ------------------------------------------------------------------------------------------- */
size_t acmil_transmil_workspace_bytes(int N, int D, int Di, int C);

int acmil_transmil_forward(const float* x, int N, int D, int Di, int C, const float* fc1_w, const float* fc1_b,
                           const float* cls_token, const float* const* layer1, const float* const* layer2,
                           const float* const* ppeg, const float* norm_w, const float* norm_b, const float* fc2_w,
                           const float* fc2_b, float* logits, float* dbg_h1, float* dbg_hp, float* dbg_h2,
                           void* workspace, void* stream);

/* The same forward with CALLER-OWNED concurrency objects -- the ownership contract of every other entry point (the library
 * allocates nothing, keeps no state, takes no lock): side_stream = a second hipStream_t of the same device, fork_event /
 * join_event = two hipEvent_t (timing may be disabled); all three NULL = every launch on `stream`.  The call records / waits on the
 * two events several times and joins back before its last launches, so `stream` alone orders the outputs (HIP-graph capture works).
 * One forward at a time may be ENQUEUED per (side_stream, fork_event, join_event) triple; their GPU work may overlap.
 * acmil_amd.ops.transmil_forward owns one triple per (device, stream) and calls this entry. */
int acmil_transmil_forward_ex(const float* x, int N, int D, int Di, int C, const float* fc1_w, const float* fc1_b,
                              const float* cls_token, const float* const* layer1, const float* const* layer2,
                              const float* const* ppeg, const float* norm_w, const float* norm_b, const float* fc2_w,
                              const float* fc2_b, float* logits, float* dbg_h1, float* dbg_hp, float* dbg_h2,
                              void* workspace, void* stream, void* side_stream, void* fork_event, void* join_event);


/* ---------------------------------------------------------------------------------------------
 * ACMIL_MHA eval forward (architecture/transformer.py:49-83 with MutiHeadAttention :107-185 and
 * MutiHeadAttention_modify :187-236).  x [N,D] fp32; W1 [Di,D]; q [K,Di];
 * branch: K x 10 pointers {q_proj.w [Di,Di], q_proj.b, k_proj.w, k_proj.b, v_proj.w, v_proj.b, out_proj.w, out_proj.b,
 * layer_norm.w, layer_norm.b} of sub_attention[i]; bag: 6 pointers {v_proj.w, v_proj.b, out_proj.w, out_proj.b,
 * layer_norm.w, layer_norm.b} of bag_attention; Wc/bc: K classifier heads [C,Di]/[C]; Ws/bs: Slide_classifier.
 * Outputs: sub_preds [K,C], slide_pred [C], attns [8,K,N] (the reference's third return value, raw scaled scores).
 * mode: ACMIL_MODE_F32 or ACMIL_MODE_F16X3 (arithmetic of the shared projection GEMM; the score GEMM is always exact).
 * Eval only (the reference's train mode draws Dropout(0.1) masks).  8 heads, Di % 64 == 0, Di <= 512, K <= 5.
 * ------------------------------------------------------------------------------------------- */
size_t acmil_mha_workspace_bytes(int N, int D, int Di, int K, int C);

int acmil_mha_forward(const float* x, int N, int D, int Di, int K, int C, const float* W1, const float* q,
                      const float* const* branch, const float* const* bag, const float* const* Wc,
                      const float* const* bc, const float* Ws, const float* bs, int mode, float* sub_preds,
                      float* slide_pred, float* attns, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * nn.Linear with a pre-packed weight stream: y = act(x W^T + bias) + beta * y   (csrc/linear_kernel.h)
 * Replaces the Linear layers around the aggregation kernels: DimReduction.fc1 / TransMIL._fc1 (architecture/network.py:49-57,
 * transMIL.py:51,63), NystromAttention.to_qkv / to_out (nystrom_attention.py:80,139).  Split-f16 arithmetic (as
 * acmil_gemm_f16x3: ~1e-6 relative, operands inside the f16 range), persistent workgroups, LDS-DMA staging.
 *   acmil_linear_pack: W [n_out, K] fp32 (leading dimension ldw) -> fragment stream of acmil_linear_packed_bytes(n_out, K)
 *     bytes (0 = shape not supported); n_out % 128 == 0, K % 16 == 0, K >= 32.  Re-pack when W changes.
 *   acmil_linear_f16x3: x [M, K] of x_dtype (leading dimension ldx elements, rows 16-byte aligned), y [M, n_out] fp32
 *     (leading dimension ldy); bias [n_out] or NULL; act 0 none / 1 relu; beta: residual accumulate.  workspace: 256 bytes.
 * ------------------------------------------------------------------------------------------- */
size_t acmil_linear_packed_bytes(int n_out, int K);

int acmil_linear_pack(const float* W, int ldw, int n_out, int K, void* packed, void* stream);

int acmil_linear_f16x3(const void* x, int x_dtype, int M, int K, long long ldx, const void* packed, int n_out,
                       const float* bias, int act, float beta, float* y, long long ldy, void* workspace, void* stream);
/* (acmil_linear_f16x3 also leaves a range word in workspace word 2 -- byte offset 8 -- of ITS call: non-zero = an output of a valid
 *  row was >= 65504 in magnitude, inf or NaN, i.e. a consumer that splits y into f16 halves again must take the fp32 path; an input
 *  value outside the f16 range poisons its row's outputs and is flagged by the same test.) */

/* Gated-attention scores of a projected bag h [N, L] in ONE pass over h (Attention_Gated.forward, architecture/transformer.py:259-267,
 * attention width 128, K <= ACMIL_MAX_TOKENS, L % 16 == 0, L >= 32): the [Wv; Wu] product with the gate tanh(.) * sigmoid(.) formed in the
 * accumulators and only A [K, N] written -- the [N, 256] pre-activations of acmil_gated_scores never exist.  packed_vu =
 * acmil_linear_pack of the [256, L] matrix with rows [Wv 0..31; Wu 0..31; Wv 32..63; Wu 32..63; ...]; bias_vu [256] in that order.
 * h fp32 / fp16 / bf16 (h_dtype), 16-byte aligned rows.  workspace: 256 bytes. */
int acmil_gated_scores_packed(const void* h, int h_dtype, int N, int L, long long ldh, const void* packed_vu, const float* bias_vu,
                              const float* Ww, const float* bw, int K, float* A, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gated attention on an already projected bag (the other gated-attention consumers).
 *   acmil_gated_scores: A [K,N] = ((tanh(h Wv^T + bv) * sigmoid(h Wu^T + bu)) Ww^T + bw)^T for h [N,L] fp32, Wv/Wu [Da,L],
 *     Ww [K,Da]; any Da, K <= 5.  Replaces Attention_Gated.forward (architecture/Attention.py:47-57, ibmil.py:27-35) and
 *     Attn_Net_Gated.forward (clam.py:62-67).  mode: ACMIL_MODE_F32 / ACMIL_MODE_F16X3 (GEMM arithmetic).
 *   acmil_attn_pool: afeat [K,Di] = softmax_N(A) h  (Attention.py:67-68, ibmil.py:73-74, clam.py:163,190); A is not modified.
 *   acmil_softmax_rows: P = softmax over each row of S [rows, cols] (the normalised attention map these modules return).
 * ------------------------------------------------------------------------------------------- */
size_t acmil_gated_scores_workspace_bytes(int N, int L, int Da, int K);

int acmil_gated_scores(const float* h, int N, int L, int Da, int K, const float* Wv, const float* bv, const float* Wu,
                       const float* bu, const float* Ww, const float* bw, int mode, float* A, void* workspace,
                       void* stream);

size_t acmil_attn_pool_workspace_bytes(int N, int Di, int K);

int acmil_attn_pool(const float* h, const float* A, int N, int Di, int K, float* afeat, void* workspace, void* stream);

int acmil_softmax_rows(const float* S, float* P, int rows, int cols, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Consumers of the raw score map A [K,N] outside the model:
 *   acmil_attn_row_stats: stats [K][4] = per row (max m, L = sum e^{s-m}, T = sum e^{s-m}(s-m), sum_n p log p = T/L - log L).
 *     evaluate()'s div_loss = sum(softmax(A) * log_softmax(A)) / K  (Step3_WSI_classification_ACMIL.py:259) is
 *     sum_k stats[k][3] / K; works for any row count (MHA: rows = 8 K).
 *   acmil_attn_heatmap: probs [N] = scale * mean_k softmax_N(A)[k][n] -- the heat-map scores of
 *     Step4_visualize_heatmap_camelyon.py:117-118 (scale = N * zoom_factor); stats [K][4] is scratch / by-product.
 * ------------------------------------------------------------------------------------------- */
int acmil_attn_row_stats(const float* A, int K, int N, float* stats, void* stream);

int acmil_attn_heatmap(const float* A, int K, int N, float scale, float* probs, float* stats, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Op-level forward / backward kernels for TRAINING the TransMIL / Nystrom path (csrc/transmil_train.hip).  The eval forward
 * is acmil_transmil_forward; training runs op by op as torch.autograd Functions (acmil_amd/autograd.py) over these entry
 * points and the GEMMs above.  All tensors fp32, row-major, caller-owned.
 *   acmil_layernorm_fwd/_bwd   nn.LayerNorm(dim, eps) (transMIL.py:12,27,57,85); stats [rows,2] = (mean, rstd)
 *   acmil_softmax_rows_bwd     dS = P * (dP - sum(dP * P)) per row (backward of the three softmaxes, nystrom_attention.py:115)
 *   acmil_seqconv              out[i][c] = sum_t w[c/d][t] v[i+t-16][c]: Conv2d(8, 8, (33,1), groups=8, no bias) (nystrom_attention.py:38,135-136);
 *                              v may be a column block of a wider matrix (ldv); its input gradient is the same conv with the flipped kernel
 *   acmil_seqconv_bwd_w        dw [8,33]
 *   acmil_dwconv7              depth-wise 7x7 on the [side,side] token grid, channels-last, weff [49,C] (+ beff [C] or NULL):
 *                              the folded PPEG stencil (transMIL.py:33-45)
 *   acmil_dwconv7_bwd_w        dweff [49,C], dbeff [C]
 *   acmil_landmark_mean/_bwd   out [8, n/l, Di/8] = means over l consecutive rows of src [n, Di] (ld) (nystrom_attention.py:95-111)
 * ------------------------------------------------------------------------------------------- */
int acmil_layernorm_fwd(const float* x, long long rows, int dim, const float* gamma, const float* beta, float eps, float* y,
                        float* stats, void* stream);
size_t acmil_layernorm_bwd_workspace_bytes(long long rows, int dim);
int acmil_layernorm_bwd(const float* x, const float* dy, const float* stats, const float* gamma, long long rows, int dim,
                        float* dx, float* dgamma, float* dbeta, void* workspace, void* stream);
int acmil_softmax_rows_bwd(const float* P, const float* dP, float* dS, long long rows, int cols, void* stream);
int acmil_seqconv(const float* v, int ldv, int n, int Di, const float* w, float* out, void* stream);
size_t acmil_seqconv_bwd_w_workspace_bytes(int n, int Di);
int acmil_seqconv_bwd_w(const float* dout, const float* v, int ldv, int n, int Di, float* dw, void* workspace, void* stream);
int acmil_dwconv7(const float* x, int side, int C, const float* weff, const float* beff, float* y, void* stream);
size_t acmil_dwconv7_bwd_w_workspace_bytes(int side, int C);
int acmil_dwconv7_bwd_w(const float* dy, const float* x, int side, int C, float* dweff, float* dbeff, void* workspace,
                        void* stream);
int acmil_landmark_mean(const float* src, int ld, int n, int l, int Di, float* out, void* stream);
int acmil_landmark_mean_bwd(const float* dout, int n, int l, int Di, float* dsrc, void* stream);
/* dx = dy where y > 0 (ReLU backward on the saved output); out[c] = sum_rows x[row][c] (bias gradients, fixed-order reduce) */
int acmil_relu_bwd(const float* dy, const float* y, float* dx, long long total, void* stream);
size_t acmil_colsum_workspace_bytes(long long rows, int cols);
int acmil_colsum(const float* x, long long rows, int cols, float* out, void* workspace, void* stream);
/* y [N,Da] = tanh(G[:, :Da]) * sigmoid(G[:, Da:]) for G [N, 2 Da], and its backward dG [N, 2 Da] (the gate of Attention_Gated,
 * transformer.py:262-264 / Attention.py:49-51 / clam.py:63-65, when the module is trained op by op) */
int acmil_gate_fwd(const float* G, float* y, long long N, int Da, void* stream);
int acmil_gate_bwd(const float* G, const float* dy, float* dG, long long N, int Da, void* stream);

/* AdamW over one flat fp32 buffer (params, grads, both moments contiguous, n elements), torch.optim.AdamW update rule with
 * decoupled weight decay -- the optimizer.step() of Step3_WSI_classification_ACMIL.py:139,219 as ONE launch.
 * step = ordinal of this call (1, 2, ...); the bias corrections 1 - beta^t use t = step - *skipped.
 * skip_flag (device, may be NULL): when it holds a non-zero (or NaN) value at execution time the launch changes nothing
 * except *skipped += 1 (device int, may be NULL) -- the range flag of a split-f16 training step (acmil_ga_train_step's
 * guard_flag, all-reduced with the gradients when data parallel), so that a step whose bag left the f16 range is never
 * applied and the host may look at the flag later. */
int acmil_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr, double beta1,
                     double beta2, float eps, float weight_decay, long long step, const float* skip_flag, int* skipped,
                     void* stream);
/* The same; flag_report (may be NULL) = a DEVICE-VISIBLE address -- pinned host memory -- that receives the value of *skip_flag
 * (0.0f without one) from the launch itself: the host learns whether the step was applied without a copy on the stream. */
int acmil_adamw_step_report(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr, double beta1,
                            double beta2, float eps, float weight_decay, long long step, const float* skip_flag, int* skipped,
                            float* flag_report, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Direct (one-shot) gradient all-reduce fused into the optimizer launch, for slide-level data parallelism on one node
 * (replaces the torch.distributed all_reduce + AdamW pair of acmil_amd.train; the reference itself is single-GPU:
 * Step3_WSI_classification_ACMIL.py:139,219).  Every rank owns two gradient slots [n + 1] fp32 (parity of the step; element n = the
 * range flag) and a flag array [8] uint32 in memory that its peers have mapped (CUDA / HIP IPC: acmil_amd/peer.py).
 *   acmil_peer_publish: bucket [n_total] -> my_slot with system-scope stores, then `step` into entry `rank` of every rank's flag
 *     array (flag_arrays[world], own included).  arrive: one zeroed device word of the caller.
 *   acmil_adamw_step_peer: waits until my_flags[r] >= step for every peer r (at most timeout_s seconds, then *err = 1 and nothing is
 *     changed), sums slots[0..world) element-wise IN RANK ORDER, divides by world and applies acmil_adamw_step's update to n
 *     elements; use_flag: element n of the slots is the averaged range flag (non-zero -> step skipped, *skipped += 1, as
 *     acmil_adamw_step).  reduced_out (may be NULL, [n + 1]): receives the averaged gradients and flag.  step >= 1, increasing.
 * ------------------------------------------------------------------------------------------- */
int acmil_peer_publish(const float* bucket, float* my_slot, long long n_total, void* const* flag_arrays, int world, int rank,
                       unsigned step, unsigned* arrive, void* stream);
int acmil_adamw_step_peer(float* params, float* exp_avg, float* exp_avg_sq, long long n, const void* const* slots,
                          const unsigned* my_flags, int world, int rank, unsigned step, double timeout_s, int* err, float lr,
                          double beta1, double beta2, float eps, float weight_decay, long long launch, int use_flag, int* skipped,
                          float* flag_report, float* reduced_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACMIL_HIP_H */
