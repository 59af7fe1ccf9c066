// ga_group.hip -- pooling + heads of a GROUP of bags whose rows lie back to back: the last stage of the batched eval forward of the
// COMPOSED gated-attention families (D_inner = 768, n_token > 5), whose first two stages -- acmil_linear_f16x3 and
// acmil_gated_scores_packed -- are per patch and run over the rows of all bags at once (ACMIL_GA.forward_group).  Per bag the
// arithmetic of acmil_ga_pool without a mask (reference: architecture/transformer.py:322-330): tiles of ga_pool_kernel are cut per
// bag (GaSeg, as the group training step cuts them), the partials sit at the running tile index, merge + heads of every bag are
// the two batched launches of the fused forward (ga_finish_batch).  The reference itself is one slide per call
// (Step3_WSI_classification_ACMIL.py:253-258).
#include "ga_train_internal.h"

static size_t ga_pool_group_part_bytes(int N, int nbags, int Di, int K) {
    return (((size_t)(ga_pool_tiles(N) + nbags) * K * ga_part_stride(Di) * sizeof(float)) + 255) & ~(size_t)255;
}

extern "C" size_t acmil_ga_pool_group_workspace_bytes(int N, int nbags, int Di, int K) {
    if (N <= 0 || nbags <= 0 || nbags > GA_SEG_MAX || Di <= 0 || K <= 0) return 0;
    return GA_CTRL_BYTES + ga_pool_group_part_bytes(N, nbags, Di, K) + (((size_t)nbags * K * Di * sizeof(float) + 255) & ~(size_t)255);
}

extern "C" int acmil_ga_pool_group(const float* h, const float* A, int N, int nbags, const int* rows, const void* packed, int D, int Di,
                                   int Da, int K, int C, int mode, float* sub_preds, float* slide_pred, float* afeat, float* bag_feat,
                                   int has_bag_head, void* workspace, void* stream) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (N <= 0 || nbags <= 0 || nbags > GA_SEG_MAX) return ACMIL_ERR_SHAPE;
    if (!h || !A || !rows || !packed || !workspace) return ACMIL_ERR_NULL;
    if (Di > 1024) return ACMIL_ERR_UNSUPPORTED;
    GaSeg S; S.n = nbags; S.row0[0] = 0;
    int tile_start[GA_SEG_MAX + 1]; tile_start[0] = 0;
    for (int b = 0; b < GA_SEG_MAX; ++b) {
        if (b < nbags && rows[b] <= 0) return ACMIL_ERR_SHAPE;
        S.row0[b + 1] = S.row0[b] + (b < nbags ? rows[b] : 0);
        tile_start[b + 1] = tile_start[b] + (b < nbags ? ga_pool_tiles(rows[b]) : 0);
    }
    if (S.row0[nbags] != N) return ACMIL_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const GaLayout L = ga_layout(D, Di, K, C, mode);
    float* part = (float*)((char*)workspace + GA_CTRL_BYTES);
    rc = ga_pool_launch(h, A, N, K, Di, part, nullptr, st, nullptr, nullptr, &S);
    if (rc != ACMIL_OK) return rc;
    float* af_scratch = (float*)((char*)part + ga_pool_group_part_bytes(N, nbags, Di, K));
    return ga_finish_batch(part, tile_start, nbags, packed, L, sub_preds, slide_pred, afeat, bag_feat, has_bag_head, af_scratch, st);
}
