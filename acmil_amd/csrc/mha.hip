// mha.hip -- eval forward of ACMIL_MHA (the `--arch mha` twin of the gated-attention aggregator).
//
// Replaces (reference file:line, /root/reference):
//   ACMIL_MHA.forward               architecture/transformer.py:68-83
//   MutiHeadAttention.forward       architecture/transformer.py:142-185   (n_token single-query attentions, 8 heads)
//   MutiHeadAttention_modify.forward architecture/transformer.py:221-236  (bag attention with a given attention map)
//
// The reference projects every patch twice per branch (k_proj, v_proj: 2 K N Di^2 = 65 GFLOP at N=50 000, Di=256, K=5)
// although each branch has ONE query.  With a single query both projections fold through the attention:
//   score[i,j,n] = (Wk_i h_n + bk_i)_j . q'_ij / sqrt(c)  =  h_n . M[:, (j,i)] + cst[(j,i)],   M = Wk_i[head j]^T q'_ij / sqrt(c)
//   out1[i, head j] = sum_n P[i,j,n] (Wv_i h_n + bv_i)_j   =  Wv_i[head j] (sum_n P[i,j,n] h_n) + bv_i[head j]
// so the whole module is: h = relu(x W1^T) (one GEMM) -> scores = M^T h^T (a [8K, Di] x [Di, N] GEMM) -> softmax over N ->
// pooled = P h (an [8K, N] x [N, Di] split-K GEMM) -> tiny per-branch projections, LayerNorm(eps 1e-6) and heads.
// The bag branch uses mean_i softmax(scores[i,j,:]), i.e. pooled_bag[j] = mean_i pooled[i,j] (linear in P).
// 1.5 GFLOP after the shared projection instead of 65; same result up to fp32 re-association (tests: <= 1e-5).
// Dropout(0.1) after out_proj is identity in eval mode; training of this module is not built.
#include <math.h>
#include "ga_common.h"

#define MHA_HEADS 8
#define MHA_MAXK ACMIL_MAX_TOKENS_FUSED

extern "C" size_t acmil_gemm_workspace_bytes(int M, int N, int K, int batch);

struct MhaFoldArgs {
    const float* q;                       // [K, Di]
    const float* Wq[MHA_MAXK]; const float* bq[MHA_MAXK]; const float* Wk[MHA_MAXK]; const float* bk[MHA_MAXK];
    float* MT;                            // [8K, Di]   row r = j*K + i
    float* cst;                           // [8K]
    int K, Di;
};

// one workgroup per (head j, branch i): q' = Wq q + bq restricted to head j, then M column and constant
__global__ __launch_bounds__(256) void mha_fold_kernel(MhaFoldArgs a) {
    __shared__ float qh[128];             // c = Di / 8 <= 128
    __shared__ float red[4];
    const int j = blockIdx.x / a.K, i = blockIdx.x % a.K, Di = a.Di, c = Di / MHA_HEADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* q = a.q + (size_t)i * Di;
    for (int r = wave; r < c; r += 4) {   // q'[j*c + r] = Wq[j*c + r, :] . q + bq
        const float* w = a.Wq[i] + (size_t)(j * c + r) * Di;
        float s = 0.0f;
        for (int d = lane; d < Di; d += 64) s += w[d] * q[d];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) qh[r] = s + a.bq[i][j * c + r];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)c);
    for (int d = tid; d < Di; d += 256) { // M[d] = sum_r Wk[j*c + r, d] q'[r] / sqrt(c)
        float s = 0.0f;
        for (int r = 0; r < c; ++r) s += a.Wk[i][(size_t)(j * c + r) * Di + d] * qh[r];
        a.MT[(size_t)blockIdx.x * Di + d] = s * scale;
    }
    float s = 0.0f;
    for (int r = tid; r < c; r += 256) s += a.bk[i][j * c + r] * qh[r];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) a.cst[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

// S[r][n] = cst[r] (the GEMM then accumulates with beta = 1)
__global__ __launch_bounds__(256) void mha_fill_kernel(float* __restrict__ S, const float* __restrict__ cst, int N, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < total) S[e] = cst[e / N];
}

// softmax over a long row, out of place (P = softmax(S[row])); 3 passes, the row stays in L2
__global__ __launch_bounds__(1024) void mha_softmax_kernel(const float* __restrict__ S, float* __restrict__ P, int cols) {
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

struct MhaHeadArgs {
    const float* pooled;                  // [8K, Di]  row j*K + i
    const float* Wv[MHA_MAXK + 1]; const float* bv[MHA_MAXK + 1]; const float* Wo[MHA_MAXK + 1]; const float* bo[MHA_MAXK + 1];
    const float* lnw[MHA_MAXK + 1]; const float* lnb[MHA_MAXK + 1];      // index K = bag attention
    const float* Wc[MHA_MAXK + 1]; const float* bc[MHA_MAXK + 1];        // index K = Slide_classifier
    float* sub_preds; float* slide_pred;
    int K, Di, C;
};

// one workgroup per branch (blockIdx.x < K) or the bag (== K): v-projection of the pooled features per head, out_proj,
// LayerNorm(eps = 1e-6, transformer.py:135), classifier
__global__ __launch_bounds__(256) void mha_heads_kernel(MhaHeadArgs a) {
    __shared__ float pin[MHA_HEADS * 512];   // [8][Di] pooled rows of this branch (Di <= 512)
    __shared__ float u[512], o[512];
    __shared__ float red[8];
    const int b = blockIdx.x, K = a.K, Di = a.Di, c = Di / MHA_HEADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < MHA_HEADS * Di; e += 256) {
        const int j = e / Di, d = e % Di;
        float v;
        if (b < K) v = a.pooled[((size_t)j * K + b) * Di + d];
        else { v = 0.0f; for (int i = 0; i < K; ++i) v += a.pooled[((size_t)j * K + i) * Di + d]; v /= (float)K; }
        pin[j * Di + d] = v;
    }
    __syncthreads();
    for (int r = wave; r < Di; r += 4) {      // u[r] = Wv[r, :] . pooled[head(r)] + bv[r]
        const float* w = a.Wv[b] + (size_t)r * Di;
        const float* pj = pin + (r / c) * Di;
        float s = 0.0f;
        for (int d = lane; d < Di; d += 64) s += w[d] * pj[d];
#pragma unroll
        for (int t = 32; t >= 1; t >>= 1) s += __shfl_xor(s, t);
        if (lane == 0) u[r] = s + a.bv[b][r];
    }
    __syncthreads();
    for (int r = wave; r < Di; r += 4) {      // o = Wo u + bo
        const float* w = a.Wo[b] + (size_t)r * Di;
        float s = 0.0f;
        for (int d = lane; d < Di; d += 64) s += w[d] * u[d];
#pragma unroll
        for (int t = 32; t >= 1; t >>= 1) s += __shfl_xor(s, t);
        if (lane == 0) o[r] = s + a.bo[b][r];
    }
    __syncthreads();
    // LayerNorm over Di (biased variance, eps 1e-6)
    float s1 = 0.0f;
    for (int d = tid; d < Di; d += 256) s1 += o[d];
#pragma unroll
    for (int t = 32; t >= 1; t >>= 1) s1 += __shfl_xor(s1, t);
    if (lane == 0) red[wave] = s1;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)Di;
    __syncthreads();
    float s2 = 0.0f;
    for (int d = tid; d < Di; d += 256) { const float t = o[d] - mean; s2 += t * t; }
#pragma unroll
    for (int t = 32; t >= 1; t >>= 1) s2 += __shfl_xor(s2, t);
    if (lane == 0) red[wave] = s2;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)Di + 1e-6f);
    for (int d = tid; d < Di; d += 256) u[d] = (o[d] - mean) * rstd * a.lnw[b][d] + a.lnb[b][d];
    __syncthreads();
    float* dst = (b < K) ? a.sub_preds + (size_t)b * a.C : a.slide_pred;
    for (int cc = wave; cc < a.C; cc += 4) {
        const float* w = a.Wc[b] + (size_t)cc * Di;
        float s = 0.0f;
        for (int d = lane; d < Di; d += 64) s += w[d] * u[d];
#pragma unroll
        for (int t = 32; t >= 1; t >>= 1) s += __shfl_xor(s, t);
        if (lane == 0) dst[cc] = s + a.bc[b][cc];
    }
}

struct MhaWs { size_t H, MT, CST, P, POOL, GEMM, total; };
static size_t mha_al(size_t b) { return (b + 255) & ~(size_t)255; }
static MhaWs mha_ws(int N, int D, int Di, int K) {
    MhaWs w; size_t off = 0;
    const int R = MHA_HEADS * K;
    w.H = off; off += mha_al((size_t)N * Di * 4);
    w.MT = off; off += mha_al((size_t)R * Di * 4);
    w.CST = off; off += mha_al((size_t)R * 4);
    w.P = off; off += mha_al((size_t)R * N * 4);
    w.POOL = off; off += mha_al((size_t)R * Di * 4);
    size_t g = acmil_gemm_workspace_bytes(N, Di, D, 1);
    const size_t g2 = acmil_gemm_workspace_bytes(R, N, Di, 1), g3 = acmil_gemm_workspace_bytes(R, Di, N, 1);
    if (g2 > g) g = g2;
    if (g3 > g) g = g3;
    w.GEMM = off; off += mha_al(g);
    w.total = off;
    return w;
}

static int mha_check(int N, int D, int Di, int K, int C) {
    if (N <= 0 || D <= 0 || Di <= 0 || K <= 0 || C <= 0) return ACMIL_ERR_SHAPE;
    if (Di % (8 * MHA_HEADS) != 0 || Di > 512 || Di / MHA_HEADS > 128) return ACMIL_ERR_UNSUPPORTED;
    if (K > MHA_MAXK || C > ACMIL_MAX_CLASSES) return ACMIL_ERR_UNSUPPORTED;
    return ACMIL_OK;
}

extern "C" size_t acmil_mha_workspace_bytes(int N, int D, int Di, int K, int C) {
    if (mha_check(N, D, Di, K, C) != ACMIL_OK) return 0;
    return mha_ws(N, D, Di, K).total;
}

extern "C" int acmil_mha_forward(const float* x, int N, int D, int Di, int K, int C, const float* W1, const float* q,
                                 const float* const* branch /* K x 10: Wq,bq,Wk,bk,Wv,bv,Wo,bo,ln_w,ln_b */,
                                 const float* const* bag /* 6: Wv,bv,Wo,bo,ln_w,ln_b */, const float* const* Wc,
                                 const float* const* bc, const float* Ws, const float* bs, int mode, float* sub_preds,
                                 float* slide_pred, float* attns /* [8, K, N] */, void* workspace, void* stream) {
    int rc = mha_check(N, D, Di, K, C);
    if (rc != ACMIL_OK) return rc;
    if (mode != ACMIL_MODE_F32 && mode != ACMIL_MODE_F16X3) return ACMIL_ERR_UNSUPPORTED;
    if (!x || !W1 || !q || !branch || !bag || !Wc || !bc || !Ws || !bs || !sub_preds || !slide_pred || !attns || !workspace) return ACMIL_ERR_NULL;
    for (int i = 0; i < K; ++i) {
        if (!Wc[i] || !bc[i]) return ACMIL_ERR_NULL;
        for (int t = 0; t < 10; ++t) if (!branch[i * 10 + t]) return ACMIL_ERR_NULL;
    }
    for (int t = 0; t < 6; ++t) if (!bag[t]) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const MhaWs W = mha_ws(N, D, Di, K);
    char* ws = (char*)workspace;
    float* H = (float*)(ws + W.H); float* MT = (float*)(ws + W.MT); float* cst = (float*)(ws + W.CST);
    float* P = (float*)(ws + W.P); float* pooled = (float*)(ws + W.POOL); void* gws = ws + W.GEMM;
    const int R = MHA_HEADS * K;

    MhaFoldArgs fa; fa.q = q; fa.MT = MT; fa.cst = cst; fa.K = K; fa.Di = Di;
    for (int i = 0; i < MHA_MAXK; ++i) {
        fa.Wq[i] = i < K ? branch[i * 10 + 0] : nullptr; fa.bq[i] = i < K ? branch[i * 10 + 1] : nullptr;
        fa.Wk[i] = i < K ? branch[i * 10 + 2] : nullptr; fa.bk[i] = i < K ? branch[i * 10 + 3] : nullptr;
    }
    hipLaunchKernelGGL(mha_fold_kernel, dim3(R), dim3(256), 0, st, fa);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    // h = relu(x W1^T)   (DimReduction, network.py:49-57)
    rc = (mode == ACMIL_MODE_F32 ? acmil_gemm_f32 : acmil_gemm_f16x3)(0, 1, N, Di, D, 1.0f, x, D, 0, W1, ACMIL_DTYPE_F32, D, 0, 0.0f, H,
                                                                     Di, 0, nullptr, 1, nullptr, 1, gws, st);
    if (rc != ACMIL_OK) return rc;
    // scores [8K, N] = M^T h^T + cst   (exact fp32: these are the returned attention logits)
    const long long total = (long long)R * N;
    hipLaunchKernelGGL(mha_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, attns, cst, N, total);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    rc = acmil_gemm_f32(0, 1, R, N, Di, 1.0f, MT, Di, 0, H, ACMIL_DTYPE_F32, Di, 0, 1.0f, attns, N, 0, nullptr, 0, nullptr, 1, gws, st);
    if (rc != ACMIL_OK) return rc;
    hipLaunchKernelGGL(mha_softmax_kernel, dim3(R), dim3(1024), 0, st, attns, P, N);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    // pooled [8K, Di] = P h   (contraction over the N patches: split-K, fixed-order reduce)
    rc = acmil_gemm_f32(0, 0, R, Di, N, 1.0f, P, N, 0, H, ACMIL_DTYPE_F32, Di, 0, 0.0f, pooled, Di, 0, nullptr, 0, nullptr, 1, gws, st);
    if (rc != ACMIL_OK) return rc;
    MhaHeadArgs ha; ha.pooled = pooled; ha.sub_preds = sub_preds; ha.slide_pred = slide_pred; ha.K = K; ha.Di = Di; ha.C = C;
    for (int i = 0; i <= MHA_MAXK; ++i) { ha.Wv[i] = ha.bv[i] = ha.Wo[i] = ha.bo[i] = ha.lnw[i] = ha.lnb[i] = ha.Wc[i] = ha.bc[i] = nullptr; }
    for (int i = 0; i < K; ++i) {
        ha.Wv[i] = branch[i * 10 + 4]; ha.bv[i] = branch[i * 10 + 5]; ha.Wo[i] = branch[i * 10 + 6]; ha.bo[i] = branch[i * 10 + 7];
        ha.lnw[i] = branch[i * 10 + 8]; ha.lnb[i] = branch[i * 10 + 9]; ha.Wc[i] = Wc[i]; ha.bc[i] = bc[i];
    }
    ha.Wv[K] = bag[0]; ha.bv[K] = bag[1]; ha.Wo[K] = bag[2]; ha.bo[K] = bag[3]; ha.lnw[K] = bag[4]; ha.lnb[K] = bag[5];
    ha.Wc[K] = Ws; ha.bc[K] = bs;
    hipLaunchKernelGGL(mha_heads_kernel, dim3(K + 1), dim3(256), 0, st, ha);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
