// ga_train_internal.h -- pieces of the GA training path shared between translation units of libacmil_hip.so
// (ga_train.hip, ga_backward.hip, ga_step.hip).  Not part of the C ABI.
#pragma once
#include "ga_common.h"
#include "gemm_internal.h"

// ga_step.hip: merge + heads of up to GA_TAIL_MAX_BAGS bags in one launch (the eval forward's finish); arrive = that many zeroed
// control-block words, left zero
#define GA_TAIL_MAX_BAGS 16
struct GaTailBatch { int start[GA_TAIL_MAX_BAGS + 1]; };
int ga_tail_eval(const float* part, const int* tile_start, int nbags, const void* packed, const GaLayout& L, float* sub_preds,
                 float* slide_pred, float* afeat, float* bag_feat, int has_bag_head, unsigned* arrive, hipStream_t st);

// ga_train.hip
// uniforms == null: the kernel draws them itself, Philox4x32-10 keyed on (rng_seed, rng_offset, branch, column)
int stkim_launch(const float* scores, float* A_mask, int N, int K, int k, int m, const float* uniforms, int64_t* topk_idx,
                 int64_t* masked_idx, unsigned long long* cand, unsigned* arrive, hipStream_t st, unsigned long long rng_seed = 0,
                 unsigned long long rng_offset = 0);
// cond (or null): the launch does nothing unless *cond != 0; cond_count (or null) is incremented once by a launch that ran under cond
int ga_pool_launch(const float* h, const float* A, int N, int K, int Di, float* part, float* gram, hipStream_t st,
                   const unsigned* cond = nullptr, unsigned* cond_count = nullptr);

// attn_generic.hip: exact-fp32 gated scores on the concatenated attention weights, predicated on *cond
int ag_gated_scores_cond(const float* h, int N, int L, int Da, int K, const float* wcat, const float* bcat, const float* Ww, const float* bw,
                         float* A, float* G, void* gws, hipStream_t st, const unsigned* cond, unsigned* cond_count = nullptr);

// ga_backward.hip
struct GbWs { size_t G, dpre, d_afeat, ck, stats, part, wcat, bcat, dwcat, gemm, gemm2, wg, total; };
GbWs gb_layout(int N, int D, int Di, int K);

struct GbRun {
    const void* x; int x_dtype; int N;
    const float *h, *A_out, *Wv, *bv, *Wu, *bu, *Ww;
    const float *Wcat, *bcat;   // [Wv; Wu] [2 Da][Di] and [bv; bu] as single operands (packed buffer), or null
    const float* WcatT;         // [Wv; Wu]^T [Di][2 Da] (packed buffer), or null
    const void *w16, *wT16;     // pre-split operands of the fused backward tile kernel (packed buffer: GaLayout::w16_off / wT16_off), or null
    const float* dA_ext;     // [K][N] external gradient of the scores, or null
    const float* coef;       // [KP][KP] diversity-loss coefficients (ga_loss.hip), or null: that term of dA is formed in the gate pass
    const float *d_afeat, *ck, *stats;   // [K][Di], [K], [K][2] = (max, sum exp) of the masked scores
    float *dW1, *dWv, *dbv, *dWu, *dbu, *dWw, *dbw;
    int D, Di, K, mode;
    char* ws;                // acmil_ga_backward_workspace_bytes
    hipStream_t st;
};
// defer (or null): the finishing launch is left to the caller, who receives what it would have worked on -- the training step with
// the optimizer inside (ga_step.hip) closes with ONE launch for finish + AdamW + re-pack (ga_opt_step.hip)
struct GbDefer { GemmArgs g_vu, g_w1; RowSumJob job; };
int gb_run(const GbRun& r, GbDefer* defer = nullptr);

// ga_opt_step.hip: split-K finish + AdamW + weight re-pack of a single-GPU training step in one launch
struct GoTensors {
    float *W1, *Wv, *bv, *Wu, *bu, *Ww, *bw, *Ws, *bs; float* Wc[ACMIL_MAX_TOKENS_FUSED]; float* bc[ACMIL_MAX_TOKENS_FUSED];
    float *dW1, *dWv, *dbv, *dWu, *dbu, *dWw, *dbw, *dWs, *dbs; float* dWc[ACMIL_MAX_TOKENS_FUSED]; float* dbc[ACMIL_MAX_TOKENS_FUSED];
};
int go_check(const GoTensors& t, int D, int Di, int K, int C, int mode, const float* flat, long long n_flat, const float* exp_avg,
             const float* exp_avg_sq);
int go_launch(const GoTensors& t, const GemmArgs* g_vu, const GemmArgs* g_w1, const RowSumJob* job, int KP, void* packed, const GaLayout& L,
              const float* flat, const float* exp_avg, const float* exp_avg_sq, float lr, double beta1, double beta2, float eps, float wd,
              long long step, const float* skip_flag, int* skipped, float* flag_report, hipStream_t st);

// wgrad.hip: both weight-gradient products (contraction over the patches) in one launch; *_bytes == 0 / ACMIL_ERR_UNSUPPORTED
// when the shapes do not fit its 128 x 128 tiles or the bag is tiny (callers then use the generic GEMM)
size_t wgrad_workspace_bytes(int M1, int N1, int M2, int N2, int K);
int wgrad_launch(const float* A1, int lda1, const void* B1, int b1_dtype, int ldb1, int M1, int N1, float* C1,
                 const float* A2, int lda2, const void* B2, int b2_dtype, int ldb2, int M2, int N2, float* C2,
                 int K, void* workspace, hipStream_t st, GemmArgs* g1, GemmArgs* g2);

// ga_bwd_tile.hip: G recompute + gate pass + dpre as one kernel per 64- / 32-patch tile (needs the pre-split operands of the packed
// buffer, with the d_afeat columns of wT16 filled); one partial record per tile (*records; ga_bwd_tile_part_records = upper bound)
size_t ga_bwd_tile_part_records(int N);
int ga_bwd_tile_launch(const float* h, const float* A, const float* stats, const float* ck, const float* coef, const float* Ww,
                       const float* d_afeat, const float* bcat, const void* w16, const void* wT16, float* dS, float* dpre,
                       float* part, int N, int K, int Di, hipStream_t st, int* records);
