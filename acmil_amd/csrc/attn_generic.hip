// attn_generic.hip -- gated attention on an ALREADY projected bag h [N, L] with free attention width Da: the building
// block of the other gated-attention consumers of the reference, which differ from ACMIL_GA only in
// where the projection sits, in biases and in Da:
//   Attention_Gated.forward(x, isNorm)      architecture/Attention.py:29-57     (DTFD tier-1 / tier-2, L = 512, D = 128)
//   Attention_with_Classifier.forward       architecture/Attention.py:60-70
//   IBMIL.forward (no confounder)           architecture/ibmil.py:69-113
//   CLAM_SB.forward (gated attention net)   architecture/clam.py:159-197       (fc WITH bias, Da = 128 ("small") or 384 ("big"))
// ACMIL_GA itself keeps the fully fused kernel (ga_forward_kernel.h: projection + scores + pooling in one pass, Da = 128);
// here the projection is the caller's GEMM and the scores run as  G = h [Wv;Wu]^T + [bv;bu]  (two MFMA GEMMs into one
// [N, 2 Da] buffer) followed by a row "gate pass":  A[k][n] = sum_u tanh(Gv[n][u]) sigmoid(Gu[n][u]) Ww[k][u] + bw[k].
// Pooling = the same online-softmax tile partials + fixed-order merge as the training path (ga_train.hip / ga_forward.hip).
#include "ga_common.h"
#include "gemm_internal.h"

extern "C" size_t acmil_gemm_workspace_bytes(int M, int N, int K, int batch);

// one wave per patch row; lane covers units u = lane, lane + 64, ...; K <= KP scores per row via shuffle trees
template <int KP>
__global__ __launch_bounds__(256) void ag_gate_scores_kernel(const float* __restrict__ G, int N, int Da, int K,
                                                             const float* __restrict__ Ww, const float* __restrict__ bw,
                                                             float* __restrict__ A, const unsigned* __restrict__ cond = nullptr,
                                                             unsigned* __restrict__ cond_count = nullptr) {
    if (cond) {      // predicated launch (exact-fp32 repeat of a flagged bag); counted once per launch that ran
        if (__builtin_nontemporal_load(cond) == 0u) return;
        if (cond_count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(cond_count, 1u);
    }
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int n = wave; n < N; n += nwaves) {
        const float* g = G + (size_t)n * 2 * Da;
        float s[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) s[k] = 0.0f;
        for (int u = lane; u < Da; u += 64) {
            const float gate = ga_tanh(g[u]) * ga_sigmoid(g[Da + u]);
#pragma unroll
            for (int k = 0; k < KP; ++k) if (k < K) s[k] = fmaf(gate, Ww[(size_t)k * Da + u], s[k]);
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s[k] += __shfl_xor(s[k], o);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) if (k < K) A[(size_t)k * N + n] = s[k] + bw[k];
        }
    }
}

// out-of-place softmax over long rows (same 3-pass scheme as the TransMIL / MHA row softmax)
__global__ __launch_bounds__(1024) void ag_softmax_rows_kernel(const float* __restrict__ S, float* __restrict__ P, int cols) {
    __shared__ float red[16];
    const float* p = S + (size_t)blockIdx.x * cols;
    float* o = P + (size_t)blockIdx.x * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int c = tid; c < cols; c += 1024) mx = fmaxf(mx, p[c]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float sum = 0.0f;
    for (int c = tid; c < cols; c += 1024) { const float e = __expf(p[c] - mx); o[c] = e; sum += e; }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sum += __shfl_xor(sum, s);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float inv = 1.0f / tot;
    for (int c = tid; c < cols; c += 1024) o[c] *= inv;
}

static size_t ag_al(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t acmil_gated_scores_workspace_bytes(int N, int L, int Da, int K) {
    if (N <= 0 || L <= 0 || Da <= 0 || K <= 0) return 0;
    return ag_al((size_t)N * 2 * Da * sizeof(float)) + ag_al(acmil_gemm_workspace_bytes(N, Da, L, 1));
}

extern "C" int acmil_gated_scores(const float* h, int N, int L, int Da, int K, const float* Wv, const float* bv, const float* Wu,
                                  const float* bu, const float* Ww, const float* bw, int mode, float* A, void* workspace,
                                  void* stream) {
    if (N <= 0 || L <= 0 || Da <= 0 || K <= 0) return ACMIL_ERR_SHAPE;
    if (K > ACMIL_MAX_TOKENS) return ACMIL_ERR_UNSUPPORTED;
    if (mode != ACMIL_MODE_F32 && mode != ACMIL_MODE_F16X3) return ACMIL_ERR_UNSUPPORTED;
    if (!h || !Wv || !bv || !Wu || !bu || !Ww || !bw || !A || !workspace) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    float* G = (float*)workspace;
    void* gws = (char*)workspace + ag_al((size_t)N * 2 * Da * sizeof(float));
    auto gemm = (mode == ACMIL_MODE_F32) ? acmil_gemm_f32 : acmil_gemm_f16x3;
    int rc = gemm(0, 1, N, Da, L, 1.0f, h, L, 0, Wv, ACMIL_DTYPE_F32, L, 0, 0.0f, G, 2 * Da, 0, bv, 0, nullptr, 1, gws, st);
    if (rc != ACMIL_OK) return rc;
    rc = gemm(0, 1, N, Da, L, 1.0f, h, L, 0, Wu, ACMIL_DTYPE_F32, L, 0, 0.0f, G + Da, 2 * Da, 0, bu, 0, nullptr, 1, gws, st);
    if (rc != ACMIL_OK) return rc;
    int blocks = (N + 3) / 4; if (blocks > 2048) blocks = 2048;
    if (K == 1) hipLaunchKernelGGL(ag_gate_scores_kernel<1>, dim3(blocks), dim3(256), 0, st, G, N, Da, K, Ww, bw, A, (const unsigned*)nullptr, (unsigned*)nullptr);
    else if (K <= 5) hipLaunchKernelGGL(ag_gate_scores_kernel<5>, dim3(blocks), dim3(256), 0, st, G, N, Da, K, Ww, bw, A, (const unsigned*)nullptr, (unsigned*)nullptr);
    else hipLaunchKernelGGL(ag_gate_scores_kernel<ACMIL_MAX_TOKENS>, dim3(blocks), dim3(256), 0, st, G, N, Da, K, Ww, bw, A, (const unsigned*)nullptr, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// Exact-fp32 gated scores on the CONCATENATED attention weights, predicated on *cond (ga_forward.hip: the repeat of a flagged wide bag):
// G = h wcat^T + bcat ([N, 2 Da], one fp32 MFMA GEMM), then the gate pass.  G: [N, 2 Da] scratch, gws: acmil_gemm_workspace_bytes(N, 2 Da, L, 1).
int ag_gated_scores_cond(const float* h, int N, int L, int Da, int K, const float* wcat, const float* bcat, const float* Ww, const float* bw,
                         float* A, float* G, void* gws, hipStream_t st, const unsigned* cond, unsigned* cond_count) {
    int rc = gemm_f32_cond(0, 1, N, 2 * Da, L, 1.0f, h, L, wcat, ACMIL_DTYPE_F32, L, G, 2 * Da, bcat, 0, gws, st, cond);
    if (rc != ACMIL_OK) return rc;
    int blocks = (N + 3) / 4; if (blocks > 2048) blocks = 2048;
    if (K == 1) hipLaunchKernelGGL(ag_gate_scores_kernel<1>, dim3(blocks), dim3(256), 0, st, G, N, Da, K, Ww, bw, A, cond, cond_count);
    else if (K <= 5) hipLaunchKernelGGL(ag_gate_scores_kernel<5>, dim3(blocks), dim3(256), 0, st, G, N, Da, K, Ww, bw, A, cond, cond_count);
    else hipLaunchKernelGGL(ag_gate_scores_kernel<ACMIL_MAX_TOKENS>, dim3(blocks), dim3(256), 0, st, G, N, Da, K, Ww, bw, A, cond, cond_count);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_softmax_rows(const float* S, float* P, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return ACMIL_ERR_SHAPE;
    if (!S || !P) return ACMIL_ERR_NULL;
    hipLaunchKernelGGL(ag_softmax_rows_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, S, P, cols);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// Consumers of the raw score map A [K, N] outside the model:
//   evaluate():  div_loss = sum(softmax(A) * log_softmax(A)) / K      Step3_WSI_classification_ACMIL.py:259
//   heat maps:   probs = softmax(A, -1).mean(0) * N * zoom_factor     Step4_visualize_heatmap_camelyon.py:117-118
// Both need the per-row softmax statistics; one workgroup per row gathers (max, sum e^{s-m}, sum e^{s-m}(s-m)) with a fixed
// reduction tree (deterministic), since  sum_n p log p = T / L - log L  with T = sum e^{s-m}(s-m), L = sum e^{s-m}.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ag_row_stats_kernel(const float* __restrict__ A, int N, float* __restrict__ stats) {
    __shared__ float red[3][16];
    const float* row = A + (size_t)blockIdx.x * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int n = tid; n < N; n += 1024) m = fmaxf(m, row[n]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[0][wave] = m;
    __syncthreads();
    float M = red[0][0];
#pragma unroll
    for (int w = 1; w < 16; ++w) M = fmaxf(M, red[0][w]);
    float l = 0.0f, t = 0.0f;
    for (int n = tid; n < N; n += 1024) {
        const float d = row[n] - M, e = __expf(d);
        l += e; t = fmaf(e, d, t);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { l += __shfl_xor(l, o); t += __shfl_xor(t, o); }
    if (lane == 0) { red[1][wave] = l; red[2][wave] = t; }
    __syncthreads();
    if (tid == 0) {
        float L = 0.0f, T = 0.0f;
        for (int w = 0; w < 16; ++w) { L += red[1][w]; T += red[2][w]; }
        stats[4 * blockIdx.x + 0] = M; stats[4 * blockIdx.x + 1] = L; stats[4 * blockIdx.x + 2] = T;
        stats[4 * blockIdx.x + 3] = T / L - __logf(L);       // sum_n p log p of this row
    }
}

__global__ __launch_bounds__(256) void ag_heatmap_kernel(const float* __restrict__ A, int K, int N, const float* __restrict__ stats,
                                                         float scale, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += __expf(A[(size_t)k * N + n] - stats[4 * k]) / stats[4 * k + 1];
    out[n] = s * scale / (float)K;
}

extern "C" int acmil_attn_row_stats(const float* A, int K, int N, float* stats, void* stream) {
    if (K <= 0 || N <= 0) return ACMIL_ERR_SHAPE;
    if (!A || !stats) return ACMIL_ERR_NULL;
    hipLaunchKernelGGL(ag_row_stats_kernel, dim3(K), dim3(1024), 0, (hipStream_t)stream, A, N, stats);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_attn_heatmap(const float* A, int K, int N, float scale, float* probs, float* stats, void* stream) {
    if (K <= 0 || N <= 0) return ACMIL_ERR_SHAPE;
    if (!A || !probs || !stats) return ACMIL_ERR_NULL;
    int rc = acmil_attn_row_stats(A, K, N, stats, stream);
    if (rc != ACMIL_OK) return rc;
    hipLaunchKernelGGL(ag_heatmap_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, K, N, stats, scale, probs);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

