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

// A GROUP of bags laid out back to back as one "super bag" (the multi-bag training step, ga_step.hip): rows row0[b] .. row0[b + 1] - 1
// of x / h / dS / dpre and columns of the [K][total rows] score matrix belong to bag b.  Everything per patch (score pass, weight
// gradients) sees one bag of row0[n] rows; the kernels that need a softmax over ONE bag (STKIM, pooling, tail, backward tile kernel)
// cut their tiles per bag: tile t of a kernel with T-row tiles -> ga_seg_find<T>.  Passed by value (kernel argument segment).
#define GA_SEG_MAX GA_TAIL_MAX_BAGS
struct GaSeg { int n; int row0[GA_SEG_MAX + 1]; };
static inline GaSeg ga_seg_single(int N) {
    GaSeg s; s.n = 1; s.row0[0] = 0;
    for (int b = 1; b <= GA_SEG_MAX; ++b) s.row0[b] = N;
    return s;
}
static inline int ga_seg_tiles(const GaSeg& s, int T) {
    int t = 0;
    for (int b = 0; b < s.n; ++b) t += (s.row0[b + 1] - s.row0[b] + T - 1) / T;
    return t;
}
// tile -> (bag, first tile of the bag, first row of the bag, first row of the tile, end row of the bag); the loop is unrolled so that
// the argument struct is only indexed statically (a dynamically indexed kernel-argument array gets a private scratch copy)
struct GaSegTile { int bag, tile0, row0, n0, nend; };
template <int T>
__device__ __forceinline__ GaSegTile ga_seg_find(const GaSeg& s, int tile) {
    GaSegTile r; r.bag = 0; r.tile0 = 0; r.row0 = 0; r.n0 = 0; r.nend = 0;
    int t0 = 0;
#pragma unroll
    for (int b = 0; b < GA_SEG_MAX; ++b) {
        const int a0 = s.row0[b], a1 = s.row0[b + 1];
        const int nt = (b < s.n) ? (a1 - a0 + T - 1) / T : 0;
        if (tile >= t0 && tile < t0 + nt) { r.bag = b; r.tile0 = t0; r.row0 = a0; r.n0 = a0 + (tile - t0) * T; r.nend = a1; }
        t0 += nt;
    }
    return r;
}

// ga_forward.hip: merge + heads for nbags <= GA_MAX_BATCH bags whose partials lie back to back (tile_start[b]: first tile of bag b,
// nbags + 1 entries); outputs [bag][...]; af_scratch: nbags * K * Di floats, used when afeat is null
int ga_finish_batch(const float* part, const int* tile_start, int nbags, const void* packed, const GaLayout& L,
                    float* sub_preds, float* slide_pred, float* afeat, float* bag_feat, int has_bag_head,
                    float* af_scratch, hipStream_t st);

// ga_train.hip
// uniforms == null: the kernel draws them itself, Philox4x32-10 keyed on (rng_seed, rng_offset, branch, column)
// seg (or null = one bag of N rows): a group of bags -- scores / A_mask are [K][ldA] with bag b in columns row0[b].., uniforms
// [bags][K][k], topk_idx [bags][K][k], masked_idx [bags][K][m] (indices local to the bag), arrive = seg->n zeroed words
int stkim_launch(const float* scores, float* A_mask, int N, int K, int k, int m, const float* uniforms, int64_t* topk_idx,
                 int64_t* masked_idx, unsigned long long* cand, unsigned* arrive, hipStream_t st, unsigned long long rng_seed = 0,
                 unsigned long long rng_offset = 0, const GaSeg* seg = nullptr);
size_t stkim_cand_bytes(int N, int nbags, int K, int k);
// cond (or null): the launch does nothing unless *cond != 0; cond_count (or null) is incremented once by a launch that ran under cond
// seg (or null): a group of bags, tiles cut per bag (partials / Gram records at the running tile index), A = [K][N total]
int ga_pool_launch(const float* h, const float* A, int N, int K, int Di, float* part, float* gram, hipStream_t st,
                   const unsigned* cond = nullptr, unsigned* cond_count = nullptr, const GaSeg* seg = nullptr);

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
    const GaSeg* seg = nullptr;      // a group of bags (N = all rows; coef / d_afeat / ck / stats per bag, strides as GaTailArgs): the
    const void* wT_ext = nullptr;    // backward tile kernel only -- ACMIL_ERR_UNSUPPORTED where it has no instance
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
// seg (or null: one bag): a group of bags -- N = all rows, tiles cut per bag, stats [bag][16] / ck [bag][16] / coef [bag][64] /
// d_afeat [bag][K][Di] per bag and wT_ext [bag][Di / 32][2][64][8] bf16 = the d_afeat K slots of the third product's operand
int ga_bwd_tile_launch(const float* h, const float* A, const float* stats, const float* ck, const float* coef, const float* Ww,
                       const float* d_afeat, const float* bcat, const void* w16, const void* wT16, float* dS, float* dpre,
                       float* part, int N, int K, int Di, hipStream_t st, int* records, const GaSeg* seg = nullptr,
                       const void* wT_ext = nullptr);
