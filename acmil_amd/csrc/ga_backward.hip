// ga_backward.hip -- backward of one ACMIL_GA training step (autograd of architecture/transformer.py:305-330;
// the reference has no explicit backward code).
//
// Given x, the saved h = relu(x W1^T) and the (masked) scores A the forward returned, with
//   P = softmax_N(A), afeat = P h, sub_k = Wc_k afeat_k + bc_k, slide = Ws mean_k(afeat_k) + bs
// and incoming dsub [K,C], dslide [C], dA_ext [K,N]:
//   1 heads      d_afeat_k = Wc_k^T dsub_k + Ws^T dslide / K ; c_k = d_afeat_k . afeat_k ; dWc, dbc, dWs, dbs
//   2 stats      M_k = max_n A[k,n], L_k = sum_n exp(A[k,n] - M_k)
//   3 G          = h [Wv;Wu]^T + [bv;bu]                          (fp32 MFMA GEMM, recompute instead of saving)
//   4 gate pass  per row n: P, dP = d_afeat h_n, dA = P (dP - c) + dA_ext (0 where masked),
//                V = tanh, U = sigmoid, g = V U, dg = dA^T Ww, dGv = dg U (1-V^2), dGu = dg V U (1-U)  (G <- dG),
//                dh0_n = sum_k P[k,n] d_afeat_k ; partial sums of dWw, dbw, dbv, dbu per workgroup
//   5 dpre       = (dh0 + dGv Wv + dGu Wu) * [h > 0]              (two GEMMs, beta = 1, relu-mask epilogue)
//   6 dWv, dWu   = dGv^T h, dGu^T h ; dW1 = dpre^T x              (split-K GEMMs over the N patches)
//   7 reduce     the per-workgroup partials of step 4 in a fixed order
// Everything heavy is exact-fp32 MFMA (gemm_f32.hip); the row pass is HBM-streaming with one wave per row.
#include <stdlib.h>

#include "ga_common.h"

#include "ga_train_internal.h"

extern "C" size_t acmil_gemm_workspace_bytes(int M, int N, int K, int batch);

#define GB_MAXK ACMIL_MAX_TOKENS
#define GB_GATE_BLOCKS 1024

struct GaBwdHeadArgs {
    const float* afeat; const float* Ws; const float* d_sub; const float* d_slide;
    const float* Wc[GB_MAXK];
    float* dWc[GB_MAXK]; float* dbc[GB_MAXK];
    float *dWs, *dbs;
    float* d_afeat;   // [K][Di] out
    float* ck;        // [K] out
    int K, C, Di;
};

// grid K + 1: workgroup k < K = branch k, workgroup K = bag head; thread = feature di (looped)
__global__ __launch_bounds__(256) void ga_bwd_heads_kernel(GaBwdHeadArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, C = a.C, Di = a.Di;
    const float invK = 1.0f / (float)K;
    if ((int)blockIdx.x < K) {
        const int k = blockIdx.x;
        float cpart = 0.0f;
        for (int di = tid; di < Di; di += 256) {
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s = fmaf(a.Wc[k][(size_t)c * Di + di], a.d_sub[k * C + c], s);
            if (a.Ws)
                for (int c = 0; c < C; ++c) s = fmaf(a.Ws[(size_t)c * Di + di] * invK, a.d_slide[c], s);
            a.d_afeat[(size_t)k * Di + di] = s;
            const float af = a.afeat[(size_t)k * Di + di];
            cpart = fmaf(s, af, cpart);
            for (int c = 0; c < C; ++c) a.dWc[k][(size_t)c * Di + di] = a.d_sub[k * C + c] * af;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cpart += __shfl_xor(cpart, o);
        __syncthreads();
        if (lane == 0) red[wave] = cpart;
        __syncthreads();
        if (tid == 0) a.ck[k] = (red[0] + red[1]) + (red[2] + red[3]);
        if (tid < C) a.dbc[k][tid] = a.d_sub[k * C + tid];
        return;
    }
    if (a.Ws) {
        for (int di = tid; di < Di; di += 256) {
            float bf = 0.0f;
            for (int k = 0; k < K; ++k) bf += a.afeat[(size_t)k * Di + di];
            bf *= invK;
            for (int c = 0; c < C; ++c) a.dWs[(size_t)c * Di + di] = a.d_slide[c] * bf;
        }
        if (tid < C) a.dbs[tid] = a.d_slide[tid];
    }
}

// grid K; softmax statistics of one branch: stats[2k] = max, stats[2k+1] = sum exp
__global__ __launch_bounds__(1024) void ga_bwd_stats_kernel(const float* __restrict__ A, int N, float* __restrict__ stats) {
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

struct GaBwdGateArgs {
    const float *h, *A, *dA_ext, *coef, *d_afeat, *ck, *stats, *Ww;   // coef [KP][KP] (or null): the diversity-loss term of dA formed here
    float* G;     // [N][256] in: pre-activations (v | u), out: dG
    float* dh0;   // [N][Di] out
    float* part;  // [blocks][KP*128 + KP + 256] partial sums
    int N, K;
};

// one wave per row (grid-stride); lane l: features 4l..4l+3 of h / dh0 (Di = 64*FPL), units 2l, 2l+1 of the gate
// COEF: the instance can form the diversity-loss term itself (a.coef, the one-call step: K <= 5); the K > 5 instances take it
// through dA_ext only and carry no [KP][KP] table
template <int KP, int FPL, bool COEF = true>
__global__ __launch_bounds__(256) void ga_bwd_gate_kernel(GaBwdGateArgs a) {
    constexpr int Di = 64 * FPL;
    constexpr int PREC = KP * GA_DA + KP + 2 * GA_DA;   // floats per workgroup partial record
    __shared__ float sred[4][PREC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, K = a.K;
    constexpr int KC = COEF ? KP : 1;
    float daf[KP][FPL], ww[KP][2], ck[KP], Mk[KP], iL[KP], cf[KC][KC];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const bool on = k < K;
        if constexpr (COEF) {
#pragma unroll
            for (int j = 0; j < KP; ++j) cf[k][j] = (a.coef && on && j < K) ? a.coef[k * KP + j] : 0.0f;
        }
#pragma unroll
        for (int f = 0; f < FPL; ++f) daf[k][f] = on ? a.d_afeat[(size_t)k * Di + FPL * lane + f] : 0.0f;
        ww[k][0] = on ? a.Ww[k * GA_DA + 2 * lane] : 0.0f;
        ww[k][1] = on ? a.Ww[k * GA_DA + 2 * lane + 1] : 0.0f;
        ck[k] = on ? a.ck[k] : 0.0f;
        Mk[k] = on ? a.stats[2 * k] : 0.0f;
        iL[k] = on ? 1.0f / a.stats[2 * k + 1] : 0.0f;
    }
    float aWw[KP][2], abw[KP], abv[2] = {0.f, 0.f}, abu[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KP; ++k) { aWw[k][0] = aWw[k][1] = 0.0f; abw[k] = 0.0f; }

    const int nwaves = gridDim.x * 4;
    for (int n = blockIdx.x * 4 + wave; n < N; n += nwaves) {
        float hv[FPL];
#pragma unroll
        for (int f = 0; f < FPL; ++f) hv[f] = a.h[(size_t)n * Di + FPL * lane + f];
        float dA[KP], P[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            float dp = 0.0f;
#pragma unroll
            for (int f = 0; f < FPL; ++f) dp = fmaf(daf[k][f], hv[f], dp);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) dp += __shfl_xor(dp, o);
            const float s = (k < K) ? a.A[(size_t)k * N + n] : -INFINITY;
            const bool masked = !(s > -5e8f);                 // masked_fill(-1e9) positions (and padded branches)
            P[k] = masked ? 0.0f : __expf(s - Mk[k]) * iL[k];
            const float ext = (a.dA_ext && k < K) ? a.dA_ext[(size_t)k * N + n] : 0.0f;
            dA[k] = masked ? 0.0f : fmaf(P[k], dp - ck[k], ext);
        }
        if constexpr (COEF) if (a.coef) {      // d diff_loss / dA[i][n] = p_i[n] * sum_j coef[i][j] p_j[n]   (ga_loss.hip; zero where masked: p = 0)
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float sdiv = 0.0f;
#pragma unroll
                for (int j = 0; j < KP; ++j) sdiv = fmaf(cf[k][j], P[j], sdiv);
                dA[k] = fmaf(P[k], sdiv, dA[k]);
            }
        }
        const float* grow = a.G + (size_t)n * (2 * GA_DA);
        const float gv0 = grow[2 * lane], gv1 = grow[2 * lane + 1];
        const float gu0 = grow[GA_DA + 2 * lane], gu1 = grow[GA_DA + 2 * lane + 1];
        const float V0 = ga_tanh(gv0), V1 = ga_tanh(gv1), U0 = ga_sigmoid(gu0), U1 = ga_sigmoid(gu1);
        const float g0 = V0 * U0, g1 = V1 * U1;
        float dg0 = 0.0f, dg1 = 0.0f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            dg0 = fmaf(dA[k], ww[k][0], dg0); dg1 = fmaf(dA[k], ww[k][1], dg1);
            aWw[k][0] = fmaf(dA[k], g0, aWw[k][0]); aWw[k][1] = fmaf(dA[k], g1, aWw[k][1]);
            abw[k] += dA[k];
        }
        const float dGv0 = dg0 * U0 * (1.0f - V0 * V0), dGv1 = dg1 * U1 * (1.0f - V1 * V1);
        const float dGu0 = dg0 * V0 * U0 * (1.0f - U0), dGu1 = dg1 * V1 * U1 * (1.0f - U1);
        abv[0] += dGv0; abv[1] += dGv1; abu[0] += dGu0; abu[1] += dGu1;
        float* gout = a.G + (size_t)n * (2 * GA_DA);
        gout[2 * lane] = dGv0; gout[2 * lane + 1] = dGv1;
        gout[GA_DA + 2 * lane] = dGu0; gout[GA_DA + 2 * lane + 1] = dGu1;
#pragma unroll
        for (int f = 0; f < FPL; ++f) {
            float d = 0.0f;
#pragma unroll
            for (int k = 0; k < KP; ++k) d = fmaf(P[k], daf[k][f], d);
            a.dh0[(size_t)n * Di + FPL * lane + f] = d;
        }
    }
    // workgroup partial record: [k][128] dWw, [k] dbw, [128] dbv, [128] dbu
    float* rec = sred[wave];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        rec[k * GA_DA + 2 * lane] = aWw[k][0]; rec[k * GA_DA + 2 * lane + 1] = aWw[k][1];
        if (lane == 0) rec[KP * GA_DA + k] = abw[k];
    }
    rec[KP * GA_DA + KP + 2 * lane] = abv[0]; rec[KP * GA_DA + KP + 2 * lane + 1] = abv[1];
    rec[KP * GA_DA + KP + GA_DA + 2 * lane] = abu[0]; rec[KP * GA_DA + KP + GA_DA + 2 * lane + 1] = abu[1];
    __syncthreads();
    float* out = a.part + (size_t)blockIdx.x * PREC;
    for (int e = tid; e < PREC; e += 256) out[e] = (sred[0][e] + sred[1][e]) + (sred[2][e] + sred[3][e]);
}

static size_t gb_align(size_t b) { return (b + 255) & ~(size_t)255; }

// [Wv; Wu] -> one [2 Da, Di] matrix and [bv; bu] -> [2 Da] (so the three products with the attention weights are single GEMMs),
// and the inverse split of the concatenated weight gradient
__global__ __launch_bounds__(256) void gb_concat_kernel(const float* __restrict__ Wv, const float* __restrict__ Wu, const float* __restrict__ bv,
                                                       const float* __restrict__ bu, int per, float* __restrict__ Wcat, float* __restrict__ bcat) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * per; e += gridDim.x * 256) Wcat[e] = e < per ? Wv[e] : Wu[e - per];
    if (blockIdx.x == 0 && threadIdx.x < 2 * GA_DA) bcat[threadIdx.x] = threadIdx.x < GA_DA ? bv[threadIdx.x] : bu[threadIdx.x - GA_DA];
}
__global__ __launch_bounds__(256) void gb_split_kernel(const float* __restrict__ dWcat, int per, float* __restrict__ dWv, float* __restrict__ dWu) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * per; e += gridDim.x * 256) (e < per ? dWv : dWu)[e < per ? e : e - per] = dWcat[e];
}

GbWs gb_layout(int N, int D, int Di, int K) {
    GbWs w; size_t off = 0;
    const int KP = ga_kp(K);
    w.G = off;       off += gb_align((size_t)N * 2 * GA_DA * 4);
    w.dpre = off;    off += gb_align((size_t)N * Di * 4);
    w.d_afeat = off; off += gb_align((size_t)K * Di * 4);
    w.ck = off;      off += 256;
    w.stats = off;   off += 256;
    const size_t recs = ga_bwd_tile_part_records(N) > GB_GATE_BLOCKS ? ga_bwd_tile_part_records(N) : GB_GATE_BLOCKS;
    w.part = off;    off += gb_align(recs * (KP * GA_DA + KP + 2 * GA_DA) * 4);
    w.wcat = off;    off += gb_align((size_t)2 * GA_DA * Di * 4);     // [Wv; Wu] as one [2 Da, Di] matrix
    w.bcat = off;    off += gb_align((size_t)2 * GA_DA * 4);
    w.dwcat = off;   off += gb_align((size_t)2 * GA_DA * Di * 4);
    const size_t g = acmil_gemm_workspace_bytes(Di, D, N, 1);          // the two split-K products keep their partials until the
    const size_t g2 = acmil_gemm_workspace_bytes(2 * GA_DA, Di, N, 1); // shared finishing launch: separate regions
    w.gemm = off;    off += gb_align(g);
    w.gemm2 = off;   off += gb_align(g2);
    w.wg = off;      off += gb_align(wgrad_workspace_bytes(2 * GA_DA, Di, Di, D, N));   // partials of the dedicated kernel (wgrad.hip)
    w.total = off;
    return w;
}

extern "C" size_t acmil_ga_backward_workspace_bytes(int N, int D, int Di, int K, int C) {
    (void)C;
    if (N <= 0 || D <= 0 || Di <= 0 || K <= 0) return 0;
    return gb_layout(N, D, Di, K).total;
}

// Steps 3-7 of the header: G recompute, gate pass, dpre, the two weight-gradient products and ONE finishing launch (both
// split-K reduces + the gate pass' partial records).  d_afeat, ck, stats are given (ws regions, filled by the caller).
// GEMM arithmetic follows the forward mode: exact fp32 MFMA, or split products -- f16 halves for the recomputed
// pre-activations (forward-sized values), bf16 halves wherever an operand is a gradient (values down to 1e-8).
// ACMIL_GA_BWD_TILE=0 keeps launches 3-5 separate (A/B measurements); read once
static bool gb_use_tile() { static const bool v = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA_BWD_TILE"); return !(e && e[0] == '0'); }(); return v; }

int gb_run(const GbRun& r, GbDefer* defer) {
    const int N = r.N, D = r.D, Di = r.Di, K = r.K;
    if (Di != 128 && Di != 256 && Di != 384 && Di != 512 && Di != 768) return ACMIL_ERR_UNSUPPORTED;   // gate pass instances (FPL = Di/64)
    const int x_fwd = (r.mode == ACMIL_MODE_F32) ? 0 : 1, x_grad = (r.mode == ACMIL_MODE_F32) ? 0 : 2;
    hipStream_t st = r.st;
    char* ws = r.ws;
    const GbWs L = gb_layout(N, D, Di, K);
    float* G = (float*)(ws + L.G);
    float* dpre = (float*)(ws + L.dpre);
    float* part = (float*)(ws + L.part);
    // [Wv; Wu] and [bv; bu] as single operands: the copy inside the packed buffer when the caller has one (training step),
    // the tensors themselves when they are adjacent in memory, a concat copy otherwise.  [dWv; dWu] are written by the
    // finishing launch straight into the two gradient tensors whenever the weight-gradient product is split (always, at
    // bag sizes worth a GPU); only the unsplit product of a tiny bag goes through a scratch matrix + split copy.
    const bool w_adj = (r.Wu == r.Wv + (size_t)GA_DA * Di) && (r.bu == r.bv + GA_DA);
    const bool g_adj = (r.dWu == r.dWv + (size_t)GA_DA * Di);
    const float* Wcat = r.Wcat ? r.Wcat : w_adj ? r.Wv : (const float*)(ws + L.wcat);
    const float* bcat = r.Wcat ? r.bcat : w_adj ? r.bv : (const float*)(ws + L.bcat);
    int rc;
    if (!r.Wcat && !w_adj) {
        hipLaunchKernelGGL(gb_concat_kernel, dim3(64), dim3(256), 0, st, r.Wv, r.Wu, r.bv, r.bu, GA_DA * Di, (float*)(ws + L.wcat), (float*)(ws + L.bcat));
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    }
    const int KP = ga_kp(K);
    if (KP > 5 && r.coef) return ACMIL_ERR_UNSUPPORTED;      // K > 5: the caller forms the diversity term (acmil_ga_loss -> d_A)
    int blocks = 0;
    // 3-5 as ONE kernel per 64-patch tile when the caller holds the pre-split operands (training step, split arithmetic)
    rc = ACMIL_ERR_UNSUPPORTED;
    if (r.w16 && r.wT16 && !r.dA_ext && r.mode != ACMIL_MODE_F32 && (gb_use_tile() || r.seg))
        rc = ga_bwd_tile_launch(r.h, r.A_out, r.stats, r.ck, r.coef, r.Ww, r.d_afeat, bcat, r.w16, r.wT16, G, dpre, part, N, K, Di, st, &blocks,
                                r.seg, r.wT_ext);
    if (rc == ACMIL_OK) {
    } else if (rc != ACMIL_ERR_UNSUPPORTED || r.seg) {       // (a group of bags has no three-launch form)
        return rc;
    } else {
        // 3 G = h [Wv;Wu]^T + [bv;bu]
        GemmArgs gd;
        rc = gemm_run_deferred(x_fwd, 0, 1, N, 2 * GA_DA, Di, 1.0f, r.h, Di, Wcat, ACMIL_DTYPE_F32, Di, 0.0f, G, 2 * GA_DA, bcat, 0, nullptr, ws + L.gemm, st, &gd);
        if (rc != ACMIL_OK) return rc;
        if (gd.splits > 1) return ACMIL_ERR_UNSUPPORTED;     // K = Di: never split
        // 4 gate pass
        GaBwdGateArgs ga;
        ga.h = r.h; ga.A = r.A_out; ga.dA_ext = r.dA_ext; ga.coef = r.coef; ga.d_afeat = r.d_afeat; ga.ck = r.ck; ga.stats = r.stats; ga.Ww = r.Ww;
        ga.G = G; ga.dh0 = dpre; ga.part = part; ga.N = N; ga.K = K;
        const int FPL = Di / 64;
        blocks = (N + 3) / 4 < GB_GATE_BLOCKS ? (N + 3) / 4 : GB_GATE_BLOCKS;
#define GB_LAUNCH_GATE(KP_, FPL_) hipLaunchKernelGGL((ga_bwd_gate_kernel<KP_, FPL_>), dim3(blocks), dim3(256), 0, st, ga)
        if (KP == 1) { if (FPL == 2) GB_LAUNCH_GATE(1, 2); else if (FPL == 4) GB_LAUNCH_GATE(1, 4); else if (FPL == 6) GB_LAUNCH_GATE(1, 6); else if (FPL == 8) GB_LAUNCH_GATE(1, 8); else GB_LAUNCH_GATE(1, 12); }
        else if (KP == 5) { if (FPL == 2) GB_LAUNCH_GATE(5, 2); else if (FPL == 4) GB_LAUNCH_GATE(5, 4); else if (FPL == 6) GB_LAUNCH_GATE(5, 6); else if (FPL == 8) GB_LAUNCH_GATE(5, 8); else GB_LAUNCH_GATE(5, 12); }
#define GB_LAUNCH_GATE_NC(KP_, FPL_) hipLaunchKernelGGL((ga_bwd_gate_kernel<KP_, FPL_, false>), dim3(blocks), dim3(256), 0, st, ga)
        else if (KP == 8) { if (FPL == 2) GB_LAUNCH_GATE_NC(8, 2); else if (FPL == 4) GB_LAUNCH_GATE_NC(8, 4); else if (FPL == 6) GB_LAUNCH_GATE_NC(8, 6); else if (FPL == 8) GB_LAUNCH_GATE_NC(8, 8); else GB_LAUNCH_GATE_NC(8, 12); }
        else if (KP == 16) { if (FPL == 2) GB_LAUNCH_GATE_NC(16, 2); else if (FPL == 4) GB_LAUNCH_GATE_NC(16, 4); else if (FPL == 6) GB_LAUNCH_GATE_NC(16, 6); else if (FPL == 8) GB_LAUNCH_GATE_NC(16, 8); else GB_LAUNCH_GATE_NC(16, 12); }
        else return ACMIL_ERR_UNSUPPORTED;
#undef GB_LAUNCH_GATE_NC
#undef GB_LAUNCH_GATE
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
        // 5 dpre = (dh0 + dS [Wv;Wu]) * [h > 0]   (dS = the gate pass' output in G, K = 2 Da)
        if (r.WcatT)    // K-contiguous copy of the weights: the operand loads are float4 rows instead of 16 strided scalars per thread
            rc = gemm_run_deferred(x_grad, 0, 1, N, Di, 2 * GA_DA, 1.0f, G, 2 * GA_DA, r.WcatT, ACMIL_DTYPE_F32, 2 * GA_DA, 1.0f, dpre, Di, nullptr, 2, r.h, ws + L.gemm, st, &gd);
        else
            rc = gemm_run_deferred(x_grad, 0, 0, N, Di, 2 * GA_DA, 1.0f, G, 2 * GA_DA, Wcat, ACMIL_DTYPE_F32, Di, 1.0f, dpre, Di, nullptr, 2, r.h, ws + L.gemm, st, &gd);
        if (rc != ACMIL_OK) return rc;
        if (gd.splits > 1) return ACMIL_ERR_UNSUPPORTED;
    }
    // 6 weight gradients (contraction over the N patches, split-K): [dWv; dWu] = dS^T h in one product, then dW1
    GemmArgs g1, g2;
    rc = ACMIL_ERR_UNSUPPORTED;
    if (r.mode != ACMIL_MODE_F32)       // dedicated kernel: both products in one launch, hardware-transposed LDS reads (wgrad.hip)
        rc = wgrad_launch(G, 2 * GA_DA, r.h, ACMIL_DTYPE_F32, Di, 2 * GA_DA, Di, r.dWv, dpre, Di, r.x, r.x_dtype, D, Di, D, r.dW1,
                          N, ws + L.wg, st, &g1, &g2);
    if (rc == ACMIL_OK) {
        if (!g_adj) { g1.C2 = r.dWu; g1.split_row = GA_DA; }
    } else if (rc != ACMIL_ERR_UNSUPPORTED) {
        return rc;
    } else {
        const bool will_split = acmil_gemm_workspace_bytes(2 * GA_DA, Di, N, 1) > 256;      // same rule as the launcher's
        float* dWcat = (g_adj || will_split) ? r.dWv : (float*)(ws + L.dwcat);
        rc = gemm_run_deferred(x_grad, 1, 0, 2 * GA_DA, Di, N, 1.0f, G, 2 * GA_DA, r.h, ACMIL_DTYPE_F32, Di, 0.0f, dWcat, Di, nullptr, 0, nullptr, ws + L.gemm2, st, &g1);
        if (rc != ACMIL_OK) return rc;
        if (will_split != (g1.splits > 1)) return ACMIL_ERR_LAUNCH;
        if (g1.splits > 1 && !g_adj) { g1.C2 = r.dWu; g1.split_row = GA_DA; }
        rc = gemm_run_deferred(x_grad, 1, 0, Di, D, N, 1.0f, dpre, Di, r.x, r.x_dtype, D, 0.0f, r.dW1, D, nullptr, 0, nullptr, ws + L.gemm, st, &g2);
        if (rc != ACMIL_OK) return rc;
    }
    // 7 one finishing launch: both split-K reduces and the gate pass' partial records (fixed order)
    RowSumJob job;
    job.part = part; job.records = blocks; job.stride = KP * GA_DA + KP + 2 * GA_DA; job.len = job.stride; job.nseg = 4;
    job.off[0] = 0;                          job.cnt[0] = K * GA_DA; job.dst[0] = r.dWw;
    job.off[1] = KP * GA_DA;                 job.cnt[1] = K;         job.dst[1] = r.dbw;
    job.off[2] = KP * GA_DA + KP;            job.cnt[2] = GA_DA;     job.dst[2] = r.dbv;
    job.off[3] = KP * GA_DA + KP + GA_DA;    job.cnt[3] = GA_DA;     job.dst[3] = r.dbu;
    if (defer) { defer->g_vu = g1; defer->g_w1 = g2; defer->job = job; }
    else {
        rc = gemm_finish(&g1, &g2, &job, st);
        if (rc != ACMIL_OK) return rc;
    }
    if (!g_adj && g1.splits <= 1) {
        hipLaunchKernelGGL(gb_split_kernel, dim3(64), dim3(256), 0, st, (const float*)(ws + L.dwcat), GA_DA * Di, r.dWv, r.dWu);
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    }
    return ACMIL_OK;
}

extern "C" int acmil_ga_backward(const void* x, int x_dtype, int N, const float* h, const float* A_out,
                                 const float* afeat, const float* Wv, const float* bv, const float* Wu, const float* bu,
                                 const float* Ww, const float* const* Wc, const float* Ws, const float* d_sub,
                                 const float* d_slide, const float* d_A, float* dW1, float* dWv, float* dbv, float* dWu,
                                 float* dbu, float* dWw, float* dbw, float* const* dWc, float* const* dbc, float* dWs,
                                 float* dbs, int D, int Di, int Da, int K, int C, int mode, void* workspace, void* stream) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (N <= 0) return ACMIL_ERR_SHAPE;
    if (mode != ACMIL_MODE_F32 && mode != ACMIL_MODE_F16X3 && mode != ACMIL_MODE_F16) return ACMIL_ERR_UNSUPPORTED;
    if (Di != 128 && Di != 256 && Di != 384 && Di != 512 && Di != 768) return ACMIL_ERR_UNSUPPORTED;
    if (!x || !h || !A_out || !afeat || !Wv || !bv || !Wu || !bu || !Ww || !Wc || !d_sub || !workspace) return ACMIL_ERR_NULL;
    if (!dW1 || !dWv || !dbv || !dWu || !dbu || !dWw || !dbw || !dWc || !dbc) return ACMIL_ERR_NULL;
    if ((Ws != nullptr) != (d_slide != nullptr) || (Ws && (!dWs || !dbs))) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const GbWs L = gb_layout(N, D, Di, K);
    float* d_afeat = (float*)(ws + L.d_afeat);
    float* ck = (float*)(ws + L.ck);
    float* stats = (float*)(ws + L.stats);

    // 1 heads
    GaBwdHeadArgs ha;
    ha.afeat = afeat; ha.Ws = Ws; ha.d_sub = d_sub; ha.d_slide = d_slide; ha.dWs = dWs; ha.dbs = dbs;
    ha.d_afeat = d_afeat; ha.ck = ck; ha.K = K; ha.C = C; ha.Di = Di;
    for (int k = 0; k < GB_MAXK; ++k) {
        ha.Wc[k] = k < K ? Wc[k] : nullptr; ha.dWc[k] = k < K ? dWc[k] : nullptr; ha.dbc[k] = k < K ? dbc[k] : nullptr;
        if (k < K && (!ha.Wc[k] || !ha.dWc[k] || !ha.dbc[k])) return ACMIL_ERR_NULL;
    }
    hipLaunchKernelGGL(ga_bwd_heads_kernel, dim3(K + 1), dim3(256), 0, st, ha);
    // 2 stats
    hipLaunchKernelGGL(ga_bwd_stats_kernel, dim3(K), dim3(1024), 0, st, A_out, N, stats);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    GbRun r;
    r.x = x; r.x_dtype = x_dtype; r.N = N; r.h = h; r.A_out = A_out; r.Wv = Wv; r.bv = bv; r.Wu = Wu; r.bu = bu; r.Ww = Ww;
    r.dA_ext = d_A; r.coef = nullptr; r.Wcat = nullptr; r.bcat = nullptr; r.WcatT = nullptr; r.w16 = nullptr; r.wT16 = nullptr; r.d_afeat = d_afeat; r.ck = ck; r.stats = stats;
    r.dW1 = dW1; r.dWv = dWv; r.dbv = dbv; r.dWu = dWu; r.dbu = dbu; r.dWw = dWw; r.dbw = dbw;
    r.D = D; r.Di = Di; r.K = K; r.mode = mode; r.ws = ws; r.st = st;
    return gb_run(r);
}
