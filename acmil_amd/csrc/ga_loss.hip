// ga_loss.hip -- the ACMIL training loss and its gradient w.r.t. the aggregator outputs, fused.
//
// Replaces the trainer-side maths of Step3_WSI_classification_ACMIL.py:201-216 (reference):
//   loss0 = CE(sub_preds [K,C], label x K)   (0 when K == 1)        loss1 = CE(slide_pred [1,C], label)
//   p = softmax_N(attn [K,N]) ; diff_loss = mean_{i<j} cosine_similarity(p_i, p_j)
//   loss = diff_loss + loss0 + loss1
// and the first backward step (what autograd would hand to the aggregator): d_sub, d_slide, d_A.
// With S_ij = p_i . p_j, n_i = sqrt(S_ii), c = 2 / (K (K-1)):
//   d diff / d p_i = c sum_{j != i} [ p_j / (n_i n_j) - S_ij p_i / (n_i^3 n_j) ]   and  (d diff / d p_i) . p_i = 0,
//   so d_A[i][n] = p_i[n] * (d diff / d p_i)[n]  (the softmax Jacobian's rank-one term vanishes identically).
// The reference does this with ~60 small torch kernels forward and ~100 backward per step; here: one statistics
// pass, one Gram pass over [K,N], a scalar kernel, one elementwise pass.
#include "ga_common.h"

#define GL_MAXK ACMIL_MAX_TOKENS
#define GL_BLOCKS 256

// grid K: softmax statistics of each branch (max, sum exp); masked entries (-1e9) contribute exp(-inf) = 0
__global__ __launch_bounds__(1024) void gl_stats_kernel(const float* __restrict__ A, int N, float* __restrict__ stats) {
    __shared__ float red[16];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = A + (size_t)k * N;
    float m = -INFINITY;
    for (int n = tid; n < N; n += 1024) m = fmaxf(m, row[n]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    float M = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) M = fmaxf(M, red[w]);
    __syncthreads();
    float l = 0.0f;
    for (int n = tid; n < N; n += 1024) l += __expf(row[n] - M);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) l += __shfl_xor(l, o);
    if (lane == 0) red[wave] = l;
    __syncthreads();
    if (tid == 0) {
        float L = 0.0f;
        for (int w = 0; w < 16; ++w) L += red[w];
        stats[2 * k] = M; stats[2 * k + 1] = L;
    }
}

// Gram partials: part[block][i*K+j] = sum over the block's patches of p_i p_j   (upper triangle incl. diagonal)
template <int KP>
__global__ __launch_bounds__(256) void gl_gram_kernel(const float* __restrict__ A, int N, int K, const float* __restrict__ stats,
                                                     float* __restrict__ part) {
    __shared__ float red[4][KP * KP];
    float Mk[KP], iL[KP], g[KP * KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { Mk[k] = k < K ? stats[2 * k] : 0.0f; iL[k] = k < K ? 1.0f / stats[2 * k + 1] : 0.0f; }
#pragma unroll
    for (int e = 0; e < KP * KP; ++e) g[e] = 0.0f;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
        float p[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) p[k] = k < K ? __expf(A[(size_t)k * N + n] - Mk[k]) * iL[k] : 0.0f;
#pragma unroll
        for (int i = 0; i < KP; ++i)
#pragma unroll
            for (int j = i; j < KP; ++j) g[i * KP + j] = fmaf(p[i], p[j], g[i * KP + j]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < KP * KP; ++e) {
        if (e / KP > e % KP) { if (lane == 0) red[wave][e] = 0.0f; continue; }      // lower triangle: never accumulated
        float v = g[e];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][e] = v;
    }
    __syncthreads();
    if (threadIdx.x < KP * KP) part[(size_t)blockIdx.x * KP * KP + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one workgroup: reduce Gram partials (fixed order), cross-entropy terms, loss values, coefficient table for the elementwise pass
// coef[i][j] (i != j) = c / (n_i n_j) ; coef[i][i] = -c * sum_{j != i} S_ij / (n_i^3 n_j)
template <int KP>
__global__ __launch_bounds__(1024) void gl_scalar_kernel(const float* __restrict__ part, int nblocks, int K, int C,
                                                       const float* __restrict__ sub, const float* __restrict__ slide,
                                                       const int64_t* __restrict__ label, float* __restrict__ losses,
                                                       float* __restrict__ d_sub, float* __restrict__ d_slide,
                                                       float* __restrict__ coef) {
    __shared__ float S[KP * KP];
    __shared__ float lsub[ACMIL_MAX_TOKENS * ACMIL_MAX_CLASSES], lslide[ACMIL_MAX_CLASSES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // logits -> LDS with one parallel load (thread 0 below would otherwise chain ~40 dependent global loads)
    if (tid < K * C) lsub[tid] = sub[tid];
    if (slide && tid < C) lslide[tid] = slide[tid];
    // wave w reduces Gram entries w, w+16, ...: lanes stride the block partials, then a fixed shuffle tree
    for (int e = wave; e < KP * KP; e += 16) {
        float s = 0.0f;
        for (int b = lane; b < nblocks; b += 64) s += part[(size_t)b * KP * KP + e];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) S[e] = s;
    }
    __syncthreads();
    if (tid != 0) return;
    const int y = (int)label[0];
    // cross entropies
    float loss0 = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float* r = lsub + k * C;
        float mx = r[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(r[c] - mx);
        const float lse = mx + logf(se);
        loss0 += lse - r[y];
        for (int c = 0; c < C; ++c) d_sub[k * C + c] = (K > 1) ? (expf(r[c] - lse) - (c == y ? 1.0f : 0.0f)) / (float)K : 0.0f;
    }
    loss0 = (K > 1) ? loss0 / (float)K : 0.0f;
    float loss1 = 0.0f;
    if (slide) {
        float mx = lslide[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, lslide[c]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(lslide[c] - mx);
        const float lse = mx + logf(se);
        loss1 = lse - lslide[y];
        for (int c = 0; c < C; ++c) d_slide[c] = expf(lslide[c] - lse) - (c == y ? 1.0f : 0.0f);
    }
    // diversity loss
    float diff = 0.0f;
    const float cpair = (K > 1) ? 2.0f / (float)(K * (K - 1)) : 0.0f;
    float nrm[KP];
    for (int i = 0; i < KP; ++i) nrm[i] = i < K ? sqrtf(S[i * KP + i]) : 1.0f;
    for (int i = 0; i < K; ++i) {
        float dsum = 0.0f;
        for (int j = 0; j < K; ++j) {
            if (j == i) continue;
            const float sij = S[(i < j ? i : j) * KP + (i < j ? j : i)];
            const float den = fmaxf(nrm[i] * nrm[j], 1e-8f);          // torch.cosine_similarity eps
            if (i < j) diff += cpair * sij / den;
            coef[i * KP + j] = cpair / den;
            dsum += sij / (nrm[i] * nrm[i] * den);
        }
        coef[i * KP + i] = -cpair * dsum;
    }
    losses[0] = loss0; losses[1] = loss1; losses[2] = diff; losses[3] = loss0 + loss1 + diff;
}

// d_A[i][n] = p_i[n] * sum_j coef[i][j] p_j[n]
template <int KP>
__global__ __launch_bounds__(256) void gl_dA_kernel(const float* __restrict__ A, int N, int K, const float* __restrict__ stats,
                                                   const float* __restrict__ coef, float* __restrict__ dA) {
    float Mk[KP], iL[KP], cf[KP * KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { Mk[k] = k < K ? stats[2 * k] : 0.0f; iL[k] = k < K ? 1.0f / stats[2 * k + 1] : 0.0f; }
#pragma unroll
    for (int e = 0; e < KP * KP; ++e) cf[e] = (e / KP < K && e % KP < K) ? coef[e] : 0.0f;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
        float p[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) p[k] = k < K ? __expf(A[(size_t)k * N + n] - Mk[k]) * iL[k] : 0.0f;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            if (i >= K) break;
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < KP; ++j) s = fmaf(cf[i * KP + j], p[j], s);
            dA[(size_t)i * N + n] = p[i] * s;
        }
    }
}

// workspace: stats [2 K] (256 B) | coef [KP][KP] (1 KiB) | Gram partials [GL_BLOCKS][KP][KP]
extern "C" size_t acmil_ga_loss_workspace_bytes(int N, int K) {
    (void)N;
    if (K <= 0 || K > GL_MAXK) return 0;
    const int KP = ga_kp(K);
    return 256 + 1024 + (((size_t)GL_BLOCKS * KP * KP * 4 + 255) & ~(size_t)255);
}

extern "C" int acmil_ga_loss(const float* sub_preds, const float* slide_pred, const float* A_out, const int64_t* label, int N,
                             int K, int C, float* losses /*[4]: loss0, loss1, diff, total*/, float* d_sub, float* d_slide,
                             float* d_A, void* workspace, void* stream) {
    if (N <= 0 || K <= 0 || C <= 0) return ACMIL_ERR_SHAPE;
    if (K > GL_MAXK || C > ACMIL_MAX_CLASSES) return ACMIL_ERR_UNSUPPORTED;
    if (!sub_preds || !A_out || !label || !losses || !d_sub || !d_A || !workspace) return ACMIL_ERR_NULL;
    if ((slide_pred != nullptr) != (d_slide != nullptr)) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    float* stats = (float*)workspace;
    float* coef = stats + 64;
    float* part = coef + 256;
    const int KP = ga_kp(K);
    int blocks = (N + 255) / 256; if (blocks > GL_BLOCKS) blocks = GL_BLOCKS;
    hipLaunchKernelGGL(gl_stats_kernel, dim3(K), dim3(1024), 0, st, A_out, N, stats);
    if (KP == 1) {
        hipLaunchKernelGGL(gl_gram_kernel<1>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, part);
        hipLaunchKernelGGL(gl_scalar_kernel<1>, dim3(1), dim3(1024), 0, st, part, blocks, K, C, sub_preds, slide_pred, label, losses, d_sub, d_slide, coef);
        hipLaunchKernelGGL(gl_dA_kernel<1>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, coef, d_A);
    } else if (KP == 5) {
        hipLaunchKernelGGL(gl_gram_kernel<5>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, part);
        hipLaunchKernelGGL(gl_scalar_kernel<5>, dim3(1), dim3(1024), 0, st, part, blocks, K, C, sub_preds, slide_pred, label, losses, d_sub, d_slide, coef);
        hipLaunchKernelGGL(gl_dA_kernel<5>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, coef, d_A);
    } else if (KP == 8) {
        hipLaunchKernelGGL(gl_gram_kernel<8>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, part);
        hipLaunchKernelGGL(gl_scalar_kernel<8>, dim3(1), dim3(1024), 0, st, part, blocks, K, C, sub_preds, slide_pred, label, losses, d_sub, d_slide, coef);
        hipLaunchKernelGGL(gl_dA_kernel<8>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, coef, d_A);
    } else if (KP == 16) {      // (register-heavy instances: a generality path, K = 9..16)
        hipLaunchKernelGGL(gl_gram_kernel<16>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, part);
        hipLaunchKernelGGL(gl_scalar_kernel<16>, dim3(1), dim3(1024), 0, st, part, blocks, K, C, sub_preds, slide_pred, label, losses, d_sub, d_slide, coef);
        hipLaunchKernelGGL(gl_dA_kernel<16>, dim3(blocks), dim3(256), 0, st, A_out, N, K, stats, coef, d_A);
    } else return ACMIL_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
