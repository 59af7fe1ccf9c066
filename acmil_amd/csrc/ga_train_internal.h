// ga_train_internal.h -- pieces of the GA training path shared between translation units of libacmil_hip.so
// (ga_train.hip, ga_backward.hip, ga_step.hip).  Not part of the C ABI.
#pragma once
#include "ga_common.h"
#include "gemm_internal.h"

// ga_train.hip
int stkim_launch(const float* scores, float* A_mask, int N, int K, int k, int m, const float* uniforms, int64_t* topk_idx,
                 int64_t* masked_idx, unsigned long long* cand, unsigned* arrive, hipStream_t st);
int ga_pool_launch(const float* h, const float* A, int N, int K, int Di, float* part, float* gram, hipStream_t st);

// ga_backward.hip
struct GbWs { size_t G, dpre, d_afeat, ck, stats, part, wcat, bcat, dwcat, gemm, gemm2, total; };
GbWs gb_layout(int N, int D, int Di, int K);

struct GbRun {
    const void* x; int x_dtype; int N;
    const float *h, *A_out, *Wv, *bv, *Wu, *bu, *Ww;
    const float *Wcat, *bcat;   // [Wv; Wu] [2 Da][Di] and [bv; bu] as single operands (packed buffer), or null
    const float* dA_ext;     // [K][N] external gradient of the scores, or null
    const float* coef;       // [KP][KP] diversity-loss coefficients (ga_loss.hip), or null: that term of dA is formed in the gate pass
    const float *d_afeat, *ck, *stats;   // [K][Di], [K], [K][2] = (max, sum exp) of the masked scores
    float *dW1, *dWv, *dbv, *dWu, *dbu, *dWw, *dbw;
    int D, Di, K, mode;
    char* ws;                // acmil_ga_backward_workspace_bytes
    hipStream_t st;
};
int gb_run(const GbRun& r);
