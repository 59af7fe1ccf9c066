// ga_step.hip -- one ACMIL_GA training step as ONE call (C ABI: acmil_ga_train_step) and the "tail" kernel that folds the
// small launches after the pooling pass into the merge.
//
// Replaces, for one slide, the reference's
//   forward   architecture/transformer.py:305-330 (training branch: STKIM :311-320, masked softmax + pooling :322-324, heads :325-330)
//   loss      Step3_WSI_classification_ACMIL.py:201-216 (branch CE, bag CE, diversity loss)
//   backward  autograd of the above
// The step is latency-bound at N = 10 000 (each kernel runs 5-50 us), so what matters is the number of launches and the host
// time per launch.  Sequence (8 launches; the reference issues ~350 small torch kernels):
//   1 pack (the parameters changed since the last step)   2 score pass (fused forward, keeps h)   3 STKIM + mask (single launch)
//   4 pooling tiles + Gram partials      5 tail: merge | last workgroup: heads, losses, dL/d(logits), head gradients,
//                                           d_afeat, c_k, softmax statistics, diversity coefficients, range flag
//   6 backward tile kernel (ga_bwd_tile.hip): G recompute + gate pass + dpre per 64-patch tile
//   7 both split-K weight gradients (wgrad.hip)   8 one finishing launch (both reduces + gate partial records)
// All of it is enqueued by one C call: the Python side does one ctypes call instead of ~26 tensor-op wrappers.
#include <string.h>

#include "ga_train_internal.h"

#define GA_MERGE_GROUPS 16
#define GS_MAXK ACMIL_MAX_TOKENS_FUSED

extern "C" size_t acmil_ga_workspace_bytes(int N, int D, int Di, int K, int C, int mode);
extern "C" size_t acmil_ga_backward_workspace_bytes(int N, int D, int Di, int K, int C);
extern "C" size_t acmil_ga_packed_bytes(int D, int Di, int Da, int K, int C, int mode);
extern "C" int acmil_ga_pack_weights(const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu,
                                     const float* Ww, const float* bw, const float* const* Wc, const float* const* bc,
                                     const float* Ws, const float* bs, int D, int Di, int Da, int K, int C, int mode,
                                     void* packed, void* stream);
extern "C" int acmil_ga_forward(const void* x, int x_dtype, int N, const void* packed, int D, int Di, int Da, int K,
                                int C, int mode, float* A_out, float* sub_preds, float* slide_pred, float* afeat,
                                float* bag_feat, float* h_save, int has_bag_head, void* workspace, void* stream);

struct GaTailArgs {
    const float* part; GaTailBatch bt; int K, Di, C, KP;      // tile partials of all bags back to back; bt.start[b] = first tile of bag b
    float* afeat;                 // [bags][K][Di] out
    unsigned* arrive;             // control-block words (one per bag): zero on entry, zero on exit
    const char* packed; GaLayout L; int has_bag_head;
    float *sub_preds, *slide_pred, *bag_feat;                 // [bags][K][C], [bags][C], [bags][Di] (any may be null)
    // ---- training extension
    const int64_t* label; const float* gram_part;
    float *stats, *losses, *d_sub, *d_slide, *coef, *d_afeat, *ck;
    float* dWc[GS_MAXK]; float* dbc[GS_MAXK];
    float *dWs, *dbs;
    const unsigned* status; float* guard_flag;    // range status of the score pass (control block word 1) -> 0 / 1 float flag
    __bf16* wT16;                                 // packed buffer: d_afeat columns of the backward tile kernel's operand (or null)
    // ---- a GROUP of bags in one training step (grid.z = bags > 1; acmil_ga_train_step_group): everything above that depends on the
    // bag is per bag -- label [bag], stats [bag][16], losses [bag][4], d_sub [bag][GS_DSUB_LD], d_slide [bag][16], coef [bag][64],
    // d_afeat [bag][K][Di], ck [bag][16], wT_ext [bag][Di / 32][2][64][8] (instead of wT16) -- every loss gradient carries gscale =
    // 1 / bags (the step's gradient is the MEAN of the bags' gradients), and the head gradients, sums over the bags, are
    // formed by the workgroup that finishes LAST of all bags (arrive_all), in bag order
    int group; float gscale; unsigned* arrive_all; __bf16* wT_ext;
};
#define GS_DSUB_LD 128

// wave-wide sum / max over all 64 lanes
__device__ static inline float gs_wsum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ static inline float gs_wmax(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// grid (K, Di/64, bags), 1024 threads.  Eval (label == nullptr): merge + heads of every bag of a batched forward in ONE launch
// (was ga_merge_kernel + ga_heads_kernel); training: one bag, plus everything listed below.
// Every workgroup: the fixed-order merge of the pooling partials for 64 features of one branch (same arithmetic and order as
// ga_merge_kernel, ga_forward.hip), branch statistics from the c == 0 workgroups.  The workgroup that arrives last (which one
// does not matter: everything below is a function of completed global data, summed in index order) finishes the step's
// small work with 16 waves instead of six further launches.
template <int KP>
__global__ __launch_bounds__(1024) void ga_tail_kernel(GaTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[GA_MERGE_GROUPS][66];
    __shared__ float smx[GA_MERGE_GROUPS];
    __shared__ int is_last;
    const int K = a.K, Di = a.Di, C = a.C;
    const int k = blockIdx.x, c = blockIdx.y, bag = blockIdx.z;
    // (the argument struct is only READ: writing a member or indexing its pointer arrays dynamically makes hipcc keep a private
    // copy of it in scratch memory -- 512 B per lane and a 2x slower kernel)
    const int tile0 = bag == 0 ? 0 : a.bt.start[bag < GA_TAIL_MAX_BAGS ? bag : GA_TAIL_MAX_BAGS];
    const int tiles = a.bt.start[bag + 1 < GA_TAIL_MAX_BAGS ? bag + 1 : GA_TAIL_MAX_BAGS] - tile0;
    const float* const part_b = a.part + (size_t)tile0 * K * (2 + Di);
    float* const afeat_b = a.afeat + (size_t)bag * K * Di;
    unsigned* const arrive_b = a.arrive + bag;
    float* const sub_b = a.sub_preds ? a.sub_preds + (size_t)bag * K * C : nullptr;
    float* const slide_b = a.slide_pred ? a.slide_pred + (size_t)bag * C : nullptr;
    float* const bagf_b = a.bag_feat ? a.bag_feat + (size_t)bag * Di : nullptr;
    float* const stats_b = a.stats ? a.stats + 16 * bag : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const size_t PS = 2 + Di;
    {
        const float* base = part_b + (size_t)k * PS;
        const size_t tstride = (size_t)K * PS;
        float m = -INFINITY;
        for (int t = tid; t < tiles; t += 1024) m = fmaxf(m, base[t * tstride]);
        m = gs_wmax(m);
        if (lane == 0) smx[g] = m;
        __syncthreads();
        float M = smx[0];
#pragma unroll
        for (int w = 1; w < GA_MERGE_GROUPS; ++w) M = fmaxf(M, smx[w]);
        float acc = 0.0f, l = 0.0f;
        const int di = 64 * c + lane;
        // (8 tiles per wave group and iteration in flight: at 391 tiles the 4-tile version was 7 dependent round trips, ~14 us;
        // 16 in flight measured the same)
        constexpr int MU = 8;
        for (int t0 = g; t0 < tiles; t0 += MU * GA_MERGE_GROUPS) {
            float pm[MU], pl[MU], pa[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int t = t0 + u * GA_MERGE_GROUPS;
                const float* p = base + (size_t)(t < tiles ? t : t0) * tstride;
                pm[u] = p[0]; pl[u] = p[1]; pa[u] = p[2 + di];
            }
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                if (t0 + u * GA_MERGE_GROUPS < tiles) {
                    const float f = __expf(pm[u] - M);
                    l = fmaf(f, pl[u], l);
                    acc = fmaf(f, pa[u], acc);
                }
            }
        }
        red[g][lane] = acc;
        if (lane == 0) red[g][64] = l;
        __syncthreads();
        if (g == 0) {
            float A = 0.0f, Ls = 0.0f;
#pragma unroll
            for (int w = 0; w < GA_MERGE_GROUPS; ++w) { A += red[w][lane]; Ls += red[w][64]; }
            // publish: WRITE-THROUGH (sc1) stores, drained, then the arrival ticket.  No release fence: buffer_wbl2 would write
            // back every dirty line of the XCD's L2 -- the 10-50 MB of h the score pass just wrote -- and took 10-35 us here.
            __hip_atomic_store(afeat_b + (size_t)k * Di + di, A / Ls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stats_b && c == 0 && lane == 0) {
                __hip_atomic_store(stats_b + 2 * k, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(stats_b + 2 * k + 1, Ls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned t = atomicAdd(arrive_b, 1u);
        is_last = (t == gridDim.x * gridDim.y - 1u) ? 1 : 0;
        if (is_last) atomicExch(arrive_b, 0u);
    }
    __syncthreads();
    if (!is_last) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drop this CU's L1 (one lane + barrier covers the workgroup)
    __syncthreads();

    // ------------------------------------------------------------------ tail (one workgroup, 16 waves)
    // Every global read the tail needs is issued in ONE batch (a dependent chain of ~10 round trips at ~2 us each was most
    // of this kernel): afeat, branch statistics, label, the tile maxima + Gram partials.  Head weights are read where they
    // are used (two batches: logits, gradients).
    float* af = (float*)smem;                      // [K][Di]
    float* bf = af + (size_t)K * Di;               // [Di]   bag feature = mean_k afeat
    float* lsub = bf + Di;                         // [K*C]  branch logits
    float* lslide = lsub + GS_MAXK * ACMIL_MAX_CLASSES;   // [C]
    float* dsub = lslide + ACMIL_MAX_CLASSES;      // [K*C]
    float* dslide = dsub + GS_MAXK * ACMIL_MAX_CLASSES;   // [C]
    float* S = dslide + ACMIL_MAX_CLASSES;         // [KP*KP] Gram of the softmax rows
    float* sc = S + 64;                            // scalars: [0..K) per-branch CE, [8] bag CE, [16 + 2k] M_k, [17 + 2k] L_k, [32] label
    float* ckp = sc + 64;                          // [16 waves][KP + 3] partial c_k
    float* wcl = ckp + 16 * 16;                    // [K*C][Di] branch-head weights, then [C][Di] bag-head weights: loaded ONCE with the
    float* wsl = wcl + (size_t)K * C * Di;         // first batch (the heads and their gradients both read them: were 2 x 3 round trips)
    const float invK = 1.0f / (float)K;
    const bool train = a.label != nullptr;
    const bool group = a.group != 0;
    const float gsc = group ? a.gscale : 1.0f;
    const float* const gram_b = a.gram_part ? a.gram_part + (size_t)tile0 * KP * KP : nullptr;
    float* const dsub_b = a.d_sub ? a.d_sub + (group ? GS_DSUB_LD * bag : 0) : nullptr;
    float* const dslide_b = a.d_slide ? a.d_slide + (group ? 16 * bag : 0) : nullptr;
    float* const coef_b = a.coef ? a.coef + (group ? 64 * bag : 0) : nullptr;
    float* const daf_b = a.d_afeat ? a.d_afeat + (group ? (size_t)bag * K * Di : 0) : nullptr;
    float* const ck_b = a.ck ? a.ck + (group ? 16 * bag : 0) : nullptr;
    float* const loss_b = a.losses ? a.losses + (group ? 4 * bag : 0) : nullptr;
    if (tid == 0 && a.guard_flag) *a.guard_flag = (a.status && __builtin_nontemporal_load(a.status) != 0u) ? 1.0f : 0.0f;
    for (int e = tid; e < K * Di; e += 1024) af[e] = __hip_atomic_load(afeat_b + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: written sc1 by other CUs
    {
        const f32x4* wsrc = (const f32x4*)(a.packed + a.L.wc_off);
        for (int e = tid; e < K * C * Di / 4; e += 1024) ((f32x4*)wcl)[e] = wsrc[e];
        if (a.has_bag_head) {
            const f32x4* ssrc = (const f32x4*)(a.packed + a.L.ws_off);
            for (int e = tid; e < C * Di / 4; e += 1024) ((f32x4*)wsl)[e] = ssrc[e];
        }
    }
    if (train) {
        if (tid < 2 * K) sc[16 + tid] = __hip_atomic_load(stats_b + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 2 * K) sc[32] = (float)(int)a.label[group ? bag : 0];
    }
    // Gram of the softmax rows from the tile partials: S_ij = sum_t g_t[i][j] f_i(t) f_j(t), f_k(t) = exp(m_t,k - M_k) / L_k.
    // Loads first (raw tile maxima and partials of up to GS_GT tiles per lane), scaling after the barrier that publishes M, L.
    constexpr int GS_GT = 4;
    float gm_i[2][GS_GT], gm_j[2][GS_GT], gv[2][GS_GT];
    if (train) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = g + 16 * q, i = e / KP, j = e % KP;
            const bool on = e < KP * KP && i <= j && j < K;
#pragma unroll
            for (int u = 0; u < GS_GT; ++u) {
                const int t = lane + 64 * u;
                const bool ok = on && t < tiles;
                const float* rec = part_b + (size_t)(ok ? t : 0) * K * PS;
                gm_i[q][u] = ok ? rec[(size_t)i * PS] : 0.0f;
                gm_j[q][u] = ok ? rec[(size_t)j * PS] : 0.0f;
                gv[q][u] = ok ? gram_b[(size_t)t * KP * KP + e] : 0.0f;
            }
        }
    }
    __syncthreads();
    for (int di = tid; di < Di; di += 1024) {
        float s = 0.0f;
        for (int kk = 0; kk < K; ++kk) s += af[kk * Di + di];
        bf[di] = s / (float)K;
        if (bagf_b) bagf_b[di] = bf[di];
    }
    if (train) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = g + 16 * q, i = e / KP, j = e % KP;
            if (e >= KP * KP) continue;
            float s = 0.0f;
            if (i <= j && j < K) {
                const float Mi = sc[16 + 2 * i], Mj = sc[16 + 2 * j], iLi = 1.0f / sc[17 + 2 * i], iLj = 1.0f / sc[17 + 2 * j];
#pragma unroll
                for (int u = 0; u < GS_GT; ++u)
                    if (lane + 64 * u < tiles) s = fmaf(gv[q][u], (__expf(gm_i[q][u] - Mi) * iLi) * (__expf(gm_j[q][u] - Mj) * iLj), s);
                for (int t = lane + 64 * GS_GT; t < tiles; t += 64) {          // bags beyond 64 * GS_GT tiles (N > 32 768)
                    const float* rec = part_b + (size_t)t * K * PS;
                    s = fmaf(gram_b[(size_t)t * KP * KP + e], (__expf(rec[(size_t)i * PS] - Mi) * iLi) * (__expf(rec[(size_t)j * PS] - Mj) * iLj), s);
                }
                s = gs_wsum(s);
            }
            if (lane == 0) S[e] = s;
        }
    }
    __syncthreads();
    // heads (ga_heads_kernel's arithmetic): one wave per output, lanes stride the Di-long dot product
    {
        const float* wc = wcl;
        const float* bc = (const float*)(a.packed + a.L.bc_off);
        const float* ws = wsl;
        const float* bs = (const float*)(a.packed + a.L.bs_off);
        const int nout = K * C + (a.has_bag_head ? C : 0);
        for (int o = g; o < nout; o += 16) {
            const float* w; const float* v; float b;
            if (o < K * C) { w = wc + (size_t)o * Di; v = af + (size_t)(o / C) * Di; b = bc[o]; }
            else { w = ws + (size_t)(o - K * C) * Di; v = bf; b = bs[o - K * C]; }
            float s = 0.0f;
            for (int di = lane; di < Di; di += 64) s = fmaf(w[di], v[di], s);
            s = gs_wsum(s) + b;
            if (lane == 0) {
                if (o < K * C) { lsub[o] = s; if (sub_b) sub_b[o] = s; }
                else { lslide[o - K * C] = s; if (slide_b) slide_b[o - K * C] = s; }
            }
        }
    }
    if (!train) return;
    __syncthreads();
    // cross entropies: wave kk < K = branch kk, wave K = bag head; lane = class
    const int y = (int)sc[32];
    if (g <= K && (g < K || a.has_bag_head)) {
        const float* r = g < K ? lsub + g * C : lslide;
        const float v = lane < C ? r[lane] : -INFINITY;
        const float mx = gs_wmax(v);
        const float se = gs_wsum(lane < C ? expf(v - mx) : 0.0f);
        const float lse = mx + logf(se);
        const float dl = lane < C ? expf(v - lse) - (lane == y ? 1.0f : 0.0f) : 0.0f;
        // (gsc = 1 / bags of a group step -- exactly 1.0f for one bag; the stores are write-through: the workgroup that finishes the
        // group reads every bag's d_sub / d_slide for the head gradients)
        if (g < K) {
            const float d = (K > 1) ? dl * invK * gsc : 0.0f;
            if (lane < C) { dsub[g * C + lane] = d; __hip_atomic_store(dsub_b + g * C + lane, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (lane == 0) sc[g] = lse - r[y];
        } else {
            const float d = dl * gsc;
            if (lane < C) { dslide[lane] = d; __hip_atomic_store(dslide_b + lane, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (lane == 0) sc[8] = lse - r[y];
        }
    }
    __syncthreads();
    if (tid == 0) {
        float loss0 = 0.0f;
        for (int kk = 0; kk < K; ++kk) loss0 += sc[kk];
        loss0 = (K > 1) ? loss0 * invK : 0.0f;
        const float loss1 = a.has_bag_head ? sc[8] : 0.0f;
        // diversity loss and the coefficient table of its gradient (ga_loss.hip, gl_scalar_kernel):
        // coef[i][j] (i != j) = c / (n_i n_j) ; coef[i][i] = -c * sum_{j != i} S_ij / (n_i^3 n_j)
        float diff = 0.0f;
        const float cpair = (K > 1) ? 2.0f / (float)(K * (K - 1)) : 0.0f;
        const float cg = cpair * gsc;                // the gradient table carries the group mean's 1 / bags, the loss value does not
        float nrm[KP];
        for (int i = 0; i < KP; ++i) nrm[i] = i < K ? sqrtf(S[i * KP + i]) : 1.0f;
        for (int i = 0; i < KP; ++i)
            for (int j = 0; j < KP; ++j) coef_b[i * KP + j] = 0.0f;
        for (int i = 0; i < K; ++i) {
            float dsum = 0.0f;
            for (int j = 0; j < K; ++j) {
                if (j == i) continue;
                const float sij = S[(i < j ? i : j) * KP + (i < j ? j : i)];
                const float den = fmaxf(nrm[i] * nrm[j], 1e-8f);          // torch.cosine_similarity eps
                if (i < j) diff += cpair * sij / den;
                coef_b[i * KP + j] = cg / den;
                dsum += sij / (nrm[i] * nrm[i] * den);
            }
            coef_b[i * KP + i] = -cg * dsum;
        }
        loss_b[0] = loss0; loss_b[1] = loss1; loss_b[2] = diff; loss_b[3] = loss0 + loss1 + diff;
    }
    // head gradients, d_afeat and c_k = d_afeat_k . afeat_k   (ga_bwd_heads_kernel's arithmetic); all branches in one sweep:
    // element e = kk * Di + di, partial c_k per wave through LDS, ONE barrier
    {
        const float* wc = wcl;
        const float* ws = wsl;
        float cp[KP];
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) cp[kk] = 0.0f;
        for (int e = tid; e < K * Di; e += 1024) {
            const int kk = e / Di, di = e - kk * Di;
            float s = 0.0f;
            for (int cc = 0; cc < C; ++cc) s = fmaf(wc[((size_t)kk * C + cc) * Di + di], dsub[kk * C + cc], s);
            if (a.has_bag_head)
                for (int cc = 0; cc < C; ++cc) s = fmaf(ws[(size_t)cc * Di + di] * invK, dslide[cc], s);
            daf_b[e] = s;
            if (a.wT16 || a.wT_ext) {      // extension K slot kk of row di of the backward tile kernel's [[Wv;Wu]^T | d_afeat^T] operand (bf16 hi / lo planes)
                const __bf16 dh = (__bf16)s, dl = (__bf16)(s - (float)dh);
                if (group) {               // the bag's own fragment pair of that K step (ga_bwd_tile.hip: wT_ext)
                    __bf16* ext = a.wT_ext + (size_t)bag * (Di / 32) * 1024;
                    ext[ga_frag_off(di, kk, 1, 0)] = dh;
                    ext[ga_frag_off(di, kk, 1, 1)] = dl;
                } else {
                    a.wT16[ga_frag_off(di, 2 * GA_DA + kk, GA_WT_KX / 16, 0)] = dh;
                    a.wT16[ga_frag_off(di, 2 * GA_DA + kk, GA_WT_KX / 16, 1)] = dl;
                }
            }
            const float afv = af[e];
            const float prod = s * afv;
#pragma unroll
            for (int q = 0; q < KP; ++q) cp[q] += (q == kk) ? prod : 0.0f;
            if (!group) {
                float* dw = a.dWc[0];
#pragma unroll
                for (int q = 1; q < GS_MAXK; ++q) dw = (kk == q) ? a.dWc[q] : dw;
                for (int cc = 0; cc < C; ++cc) dw[(size_t)cc * Di + di] = dsub[kk * C + cc] * afv;
            }
        }
        if (group && a.wT_ext) {      // the unused K slots K .. 15 of the bag's fragment pair: zero (the workspace is recycled memory)
            __bf16* ext = a.wT_ext + (size_t)bag * (Di / 32) * 1024;
            for (int e = K * Di + tid; e < 16 * Di; e += 1024) {
                const int kk = e / Di, di = e - kk * Di;
                ext[ga_frag_off(di, kk, 1, 0)] = (__bf16)0.0f;
                ext[ga_frag_off(di, kk, 1, 1)] = (__bf16)0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < KP; ++q) {
            const float v = gs_wsum(cp[q]);
            if (lane == 0) ckp[g * (KP + 3) + q] = v;
        }
        if (!group) {
            if (a.has_bag_head) {
                for (int di = tid; di < Di; di += 1024)
                    for (int cc = 0; cc < C; ++cc) a.dWs[(size_t)cc * Di + di] = dslide[cc] * bf[di];
                if (tid < C) a.dbs[tid] = dslide[tid];
            }
            if (tid < K * C) {
                float* db = a.dbc[0];
#pragma unroll
                for (int q = 1; q < GS_MAXK; ++q) db = (tid / C == q) ? a.dbc[q] : db;
                db[tid % C] = dsub[tid];
            }
        }
        __syncthreads();
        if (tid < K) {
            float s = 0.0f;
            for (int w = 0; w < 16; ++w) s += ckp[w * (KP + 3) + tid];
            ck_b[tid] = s;
        }
    }
    if (!group) return;
    // ------------------------------------------------------------------ group: head gradients = sums over the bags, by whoever finishes last
    // (this bag's d_sub / d_slide went out write-through above; afeat was published by the merge workgroups)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = atomicAdd(a.arrive_all, 1u);
        is_last = (t == gridDim.z - 1u) ? 1 : 0;
        if (is_last) atomicExch(a.arrive_all, 0u);
    }
    __syncthreads();
    if (!is_last) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    {
        const int nb = gridDim.z;
        float* gaf = (float*)smem;                       // [bags][K][Di]
        float* gbf = gaf + (size_t)nb * K * Di;          // [bags][Di]     bag features
        float* gds = gbf + (size_t)nb * Di;              // [bags][K*C]    d_sub (scaled)
        float* gdl = gds + (size_t)nb * K * C;           // [bags][C]      d_slide (scaled)
        for (int e = tid; e < nb * K * Di; e += 1024) gaf[e] = __hip_atomic_load(a.afeat + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int e = tid; e < nb * K * C; e += 1024)
            gds[e] = __hip_atomic_load(a.d_sub + (size_t)(e / (K * C)) * GS_DSUB_LD + e % (K * C), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.has_bag_head)
            for (int e = tid; e < nb * C; e += 1024)
                gdl[e] = __hip_atomic_load(a.d_slide + (size_t)(e / C) * 16 + e % C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        for (int e = tid; e < nb * Di; e += 1024) {      // bag feature exactly as each bag's tail formed it
            const int b = e / Di, di = e - b * Di;
            float s = 0.0f;
            for (int kk = 0; kk < K; ++kk) s += gaf[((size_t)b * K + kk) * Di + di];
            gbf[e] = s / (float)K;
        }
        __syncthreads();
        for (int e = tid; e < K * Di; e += 1024) {
            const int kk = e / Di, di = e - kk * Di;
            float* dw = a.dWc[0];
#pragma unroll
            for (int q = 1; q < GS_MAXK; ++q) dw = (kk == q) ? a.dWc[q] : dw;
            for (int cc = 0; cc < C; ++cc) {
                float s = 0.0f;
                for (int b = 0; b < nb; ++b) s = fmaf(gds[(size_t)b * K * C + kk * C + cc], gaf[(size_t)b * K * Di + e], s);
                dw[(size_t)cc * Di + di] = s;
            }
        }
        if (a.has_bag_head) {
            for (int di = tid; di < Di; di += 1024)
                for (int cc = 0; cc < C; ++cc) {
                    float s = 0.0f;
                    for (int b = 0; b < nb; ++b) s = fmaf(gdl[b * C + cc], gbf[(size_t)b * Di + di], s);
                    a.dWs[(size_t)cc * Di + di] = s;
                }
            if (tid < C) {
                float s = 0.0f;
                for (int b = 0; b < nb; ++b) s += gdl[b * C + tid];
                a.dbs[tid] = s;
            }
        }
        if (tid < K * C) {
            float* db = a.dbc[0];
#pragma unroll
            for (int q = 1; q < GS_MAXK; ++q) db = (tid / C == q) ? a.dbc[q] : db;
            float s = 0.0f;
            for (int b = 0; b < nb; ++b) s += gds[(size_t)b * K * C + tid];
            db[tid % C] = s;
        }
    }
}

static size_t gs_tail_lds(int K, int Di, int C) {
    return ((size_t)K * Di + Di + 2 * (GS_MAXK * ACMIL_MAX_CLASSES + ACMIL_MAX_CLASSES) + 64 + 64 + 16 * 16 + (size_t)(K + 1) * C * Di) * sizeof(float);
}

static int gs_tail_launch(const GaTailArgs& t, int nbags, hipStream_t st) {
    size_t lds = gs_tail_lds(t.K, t.Di, t.C);
    if (t.group) {       // the group's closing section keeps every bag's afeat, bag feature, d_sub and d_slide
        const size_t need = (size_t)nbags * ((size_t)t.K * t.Di + t.Di + (size_t)t.K * t.C + t.C) * sizeof(float);
        if (need > lds) lds = need;
    }
    if (lds > 160 * 1024) return ACMIL_ERR_UNSUPPORTED;
    void (*tail)(GaTailArgs) = t.KP == 1 ? ga_tail_kernel<1> : t.KP == 5 ? ga_tail_kernel<5> : nullptr;
    if (!tail) return ACMIL_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(tail, dim3(t.K, t.Di / 64, nbags), dim3(1024), lds, st, t);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// eval: merge + heads of a batched forward as one launch (ga_forward.hip).  arrive: nbags zeroed control-block words.
int ga_tail_eval(const float* part, const int* tile_start, int nbags, const void* packed, const GaLayout& L, float* sub_preds,
                 float* slide_pred, float* afeat, float* bag_feat, int has_bag_head, unsigned* arrive, hipStream_t st) {
    GaTailArgs t;
    memset(&t, 0, sizeof(t));
    t.part = part;
    for (int b = 0; b <= GA_TAIL_MAX_BAGS; ++b) t.bt.start[b] = tile_start[b <= nbags ? b : nbags];
    t.K = L.K; t.Di = L.Di; t.C = L.C; t.KP = (L.K <= 1) ? 1 : (L.K <= 5) ? 5 : 8;
    t.afeat = afeat; t.arrive = arrive; t.packed = (const char*)packed; t.L = L; t.has_bag_head = has_bag_head;
    t.sub_preds = sub_preds; t.slide_pred = slide_pred; t.bag_feat = bag_feat;
    return gs_tail_launch(t, nbags, st);
}

static size_t gs_align(size_t b) { return (b + 255) & ~(size_t)255; }

struct GsWs { size_t part, h, gram, coef, dsub, dslide, afeat, cand, bwd, g_stats, g_ck, g_daf, g_ext, total; };

// nbags > 1: a group step (N = rows of all bags): every bag may end on a partial tile / chunk, and what depends on the bag's own
// softmax and loss exists once per bag (strides: GaTailArgs)
static GsWs gs_layout(int N, int D, int Di, int K, int C, int k_top, int nbags = 1) {
    GsWs w; size_t off = GA_CTRL_BYTES;
    const int KP = (K <= 1) ? 1 : (K <= 5) ? 5 : 8;
    const int Nt = nbags > 1 ? N + GA_POOL_ROWS * nbags : N;       // tile-count bound of a group: + one tile per bag
    w.part = off;   off += gs_align(acmil_ga_workspace_bytes(Nt, D, Di, K, C, ACMIL_MODE_F16X3) - GA_CTRL_BYTES);
    w.h = off;      off += gs_align((size_t)N * Di * 4);
    w.gram = off;   off += gs_align((size_t)ga_pool_tiles(Nt) * KP * KP * 4);
    w.coef = off;   off += gs_align((size_t)nbags * 64 * 4);
    w.dsub = off;   off += nbags > 1 ? gs_align((size_t)nbags * GS_DSUB_LD * 4) : gs_align((size_t)K * C * 4);
    w.dslide = off; off += gs_align((size_t)nbags * 16 * 4);
    w.afeat = off;  off += gs_align((size_t)nbags * K * Di * 4);
    w.cand = off;   off += nbags > 1 ? stkim_cand_bytes(N, nbags, K, k_top) : gs_align((size_t)K * ((size_t)(N + 4095) / 4096) * (k_top > 0 ? k_top : 1) * 8);
    w.bwd = off;    off += gs_align(acmil_ga_backward_workspace_bytes(N, D, Di, K, C));
    w.g_stats = w.g_ck = w.g_daf = w.g_ext = 0;
    if (nbags > 1) {
        w.g_stats = off; off += gs_align((size_t)nbags * 16 * 4);
        w.g_ck = off;    off += gs_align((size_t)nbags * 16 * 4);
        w.g_daf = off;   off += gs_align((size_t)nbags * K * Di * 4);
        w.g_ext = off;   off += gs_align((size_t)nbags * (Di / 32) * 1024 * 2);
    }
    w.total = off;
    return w;
}

extern "C" size_t acmil_ga_train_step_workspace_bytes(int N, int D, int Di, int K, int C, int k_top) {
    if (N <= 0 || D <= 0 || Di <= 0 || K <= 0 || K > GS_MAXK || C <= 0 || k_top < 0) return 0;
    return gs_layout(N, D, Di, K, C, k_top).total;
}

extern "C" size_t acmil_ga_train_step_group_workspace_bytes(int nbags, int N_total, int D, int Di, int K, int C, int k_top) {
    if (nbags <= 0 || nbags > GA_SEG_MAX || N_total < nbags || D <= 0 || Di <= 0 || K <= 0 || K > GS_MAXK || C <= 0 || k_top < 0) return 0;
    return gs_layout(N_total, D, Di, K, C, k_top, nbags).total;
}

// the optimizer inside the step (acmil_ga_train_step_adamw), or null: the caller's own launch follows
struct GsOpt {
    const float* flat; long long n_flat; float *exp_avg, *exp_avg_sq;
    float lr; double beta1, beta2; float eps, wd; long long step; int* skipped; float* flag_report;
};

static int gs_step(const void* x, int x_dtype, int N, void* packed, int repack,
                   const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu,
                   const float* Ww, const float* bw, const float* const* Wc, const float* const* bc,
                   const float* Ws, const float* bs,
                   float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                   float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                   int D, int Di, int Da, int K, int C, int mode,
                   const int64_t* label, const float* uniforms, int k_top, int m_mask,
                   float* losses, float* sub_preds, float* slide_pred, float* A_out,
                   int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream,
                   unsigned long long rng_seed, unsigned long long rng_offset, const GsOpt* opt, const GaSeg* seg = nullptr) {
    // seg: a GROUP of bags in one step (acmil_ga_train_step_group): x = the bags' rows back to back, N = all rows, label [bags],
    // uniforms [bags][K][k_top], losses [bags][4], sub_preds [bags][K][C], slide_pred [bags][C], A_out [K][N], topk_idx
    // [bags][K][k_top], masked_idx [bags][K][m_mask]; the gradients are the MEAN over the bags
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (K > GS_MAXK) return ACMIL_ERR_UNSUPPORTED;       // the one-call step exists for the fused families (K <= 5); K above: op by op
    if (N <= 0 || k_top < 0 || k_top > 64 || k_top > N || m_mask < 0 || m_mask > k_top) return ACMIL_ERR_SHAPE;
    if (mode != ACMIL_MODE_F32 && mode != ACMIL_MODE_F16X3 && mode != ACMIL_MODE_F16) return ACMIL_ERR_UNSUPPORTED;
    const int nbags = seg ? seg->n : 1;
    if (seg) {
        if (nbags < 1 || nbags > GA_SEG_MAX || seg->row0[0] != 0 || seg->row0[nbags] != N) return ACMIL_ERR_SHAPE;
        for (int b = 0; b < nbags; ++b)
            if (seg->row0[b + 1] - seg->row0[b] < (k_top > 0 ? k_top : 1)) return ACMIL_ERR_SHAPE;
        // the group step runs on the backward tile kernel only (split arithmetic, D_inner 128 / 256); fp32 repeats go bag by bag
        if (mode == ACMIL_MODE_F32 || (Di != 128 && Di != 256)) return ACMIL_ERR_UNSUPPORTED;
    }
    const bool group = seg && nbags > 1;
    if (!x || !packed || !W1 || !Wv || !bv || !Wu || !bu || !Ww || !bw || !Wc || !bc || !label || !workspace) return ACMIL_ERR_NULL;
    if (!dW1 || !dWv || !dbv || !dWu || !dbu || !dWw || !dbw || !dWc || !dbc || !losses || !sub_preds || !A_out) return ACMIL_ERR_NULL;
    const int has_bag_head = Ws != nullptr;
    if (has_bag_head && (!bs || !dWs || !dbs || !slide_pred)) return ACMIL_ERR_NULL;
    if (k_top > 0 && (!topk_idx || (m_mask > 0 && !masked_idx))) return ACMIL_ERR_NULL;      // uniforms null: drawn on the device
    for (int k = 0; k < K; ++k)
        if (!Wc[k] || !bc[k] || !dWc[k] || !dbc[k]) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    // the optimizer inside the step: everything that can refuse does so BEFORE the first launch
    GoTensors gt;
    if (opt) {
        memset(&gt, 0, sizeof(gt));
        gt.W1 = (float*)W1; gt.Wv = (float*)Wv; gt.bv = (float*)bv; gt.Wu = (float*)Wu; gt.bu = (float*)bu; gt.Ww = (float*)Ww; gt.bw = (float*)bw;
        gt.Ws = (float*)Ws; gt.bs = (float*)bs;
        gt.dW1 = dW1; gt.dWv = dWv; gt.dbv = dbv; gt.dWu = dWu; gt.dbu = dbu; gt.dWw = dWw; gt.dbw = dbw; gt.dWs = dWs; gt.dbs = dbs;
        for (int k = 0; k < K; ++k) { gt.Wc[k] = (float*)Wc[k]; gt.bc[k] = (float*)bc[k]; gt.dWc[k] = dWc[k]; gt.dbc[k] = dbc[k]; }
        if (opt->step < 1 || !(opt->beta1 >= 0.0 && opt->beta1 < 1.0) || !(opt->beta2 >= 0.0 && opt->beta2 < 1.0)) return ACMIL_ERR_SHAPE;
        rc = go_check(gt, D, Di, K, C, mode, opt->flat, opt->n_flat, opt->exp_avg, opt->exp_avg_sq);
        if (rc != ACMIL_OK) return rc;
    }
    char* ws = (char*)workspace;
    const GsWs W = gs_layout(N, D, Di, K, C, k_top, group ? nbags : 1);
    unsigned* ctrl = (unsigned*)ws;
    float* h = (float*)(ws + W.h);
    float* part = (float*)(ws + W.part);
    float* gram = (float*)(ws + W.gram);
    float* afeat = (float*)(ws + W.afeat);
    float* coef = (float*)(ws + W.coef);
    char* bws = ws + W.bwd;
    const GbWs BL = gb_layout(N, D, Di, K);
    const int KP = (K <= 1) ? 1 : (K <= 5) ? 5 : 8;

    // 1 pack  2 score pass (the control block sits at the start of this workspace: tile counter, range status)
    if (repack) {
        rc = acmil_ga_pack_weights(W1, Wv, bv, Wu, bu, Ww, bw, Wc, bc, Ws, bs, D, Di, Da, K, C, mode, packed, stream);
        if (rc != ACMIL_OK) return rc;
    }
    rc = acmil_ga_forward(x, x_dtype, N, packed, D, Di, Da, K, C, mode, A_out, nullptr, nullptr, nullptr, nullptr, h, has_bag_head,
                          workspace, stream);
    if (rc != ACMIL_OK) return rc;
    // 3 STKIM + mask
    if (k_top > 0) {
        // (control block words: 4 STKIM arrival of a single bag, 5 .. 20 tail arrivals per bag, 21 .. 36 STKIM arrivals of a group, 37 the group's closing arrival)
        rc = stkim_launch(A_out, m_mask > 0 ? A_out : nullptr, N, K, k_top, m_mask, uniforms, topk_idx, masked_idx,
                          (unsigned long long*)(ws + W.cand), group ? ctrl + 21 : ctrl + 4, st, rng_seed, rng_offset, group ? seg : nullptr);
        if (rc != ACMIL_OK) return rc;
    }
    // 4 pooling tiles + Gram partials
    rc = ga_pool_launch(h, A_out, N, K, Di, part, gram, st, nullptr, nullptr, group ? seg : nullptr);
    if (rc != ACMIL_OK) return rc;
    // 5 tail
    GaTailArgs t;
    memset(&t, 0, sizeof(t));
    t.part = part; t.K = K; t.Di = Di; t.C = C; t.KP = KP;
    for (int b = 1; b <= GA_TAIL_MAX_BAGS; ++b)
        t.bt.start[b] = group ? t.bt.start[b - 1] + (b <= nbags ? ga_pool_tiles(seg->row0[b] - seg->row0[b - 1]) : 0) : ga_pool_tiles(N);
    t.afeat = afeat; t.arrive = ctrl + 5; t.packed = (const char*)packed; t.L = ga_layout(D, Di, K, C, mode); t.has_bag_head = has_bag_head;
    t.sub_preds = sub_preds; t.slide_pred = slide_pred; t.label = label; t.gram_part = gram;
    t.stats = (float*)(bws + BL.stats); t.losses = losses; t.d_sub = (float*)(ws + W.dsub); t.d_slide = (float*)(ws + W.dslide);
    t.coef = coef; t.d_afeat = (float*)(bws + BL.d_afeat); t.ck = (float*)(bws + BL.ck);
    for (int k = 0; k < GS_MAXK; ++k) { t.dWc[k] = k < K ? dWc[k] : nullptr; t.dbc[k] = k < K ? dbc[k] : nullptr; }
    t.dWs = dWs; t.dbs = dbs;
    t.status = (mode == ACMIL_MODE_F16X3) ? ctrl + 1 : nullptr;      // only the split-f16 score pass reports a range status
    t.guard_flag = guard_flag;
    t.wT16 = (__bf16*)((char*)packed + t.L.wT16_off);
    if (group) {
        t.group = 1; t.gscale = 1.0f / (float)nbags; t.arrive_all = ctrl + 37; t.wT_ext = (__bf16*)(ws + W.g_ext);
        t.stats = (float*)(ws + W.g_stats); t.ck = (float*)(ws + W.g_ck); t.d_afeat = (float*)(ws + W.g_daf);
    }
    rc = gs_tail_launch(t, nbags, st);
    if (rc != ACMIL_OK) return rc;
    // 6-11 backward
    GbRun r;
    r.x = x; r.x_dtype = x_dtype; r.N = N; r.h = h; r.A_out = A_out; r.Wv = Wv; r.bv = bv; r.Wu = Wu; r.bu = bu; r.Ww = Ww;
    r.Wcat = (const float*)((const char*)packed + t.L.wcat_off); r.bcat = (const float*)((const char*)packed + t.L.bcat_off);
    r.WcatT = (const float*)((const char*)packed + t.L.wcatT_off);
    r.w16 = (const char*)packed + t.L.w16_off; r.wT16 = (const char*)packed + t.L.wT16_off;
    r.dA_ext = nullptr; r.coef = (K > 1) ? coef : nullptr; r.d_afeat = t.d_afeat; r.ck = t.ck; r.stats = t.stats;
    r.dW1 = dW1; r.dWv = dWv; r.dbv = dbv; r.dWu = dWu; r.dbu = dbu; r.dWw = dWw; r.dbw = dbw;
    r.D = D; r.Di = Di; r.K = K; r.mode = mode; r.ws = bws; r.st = st;
    r.seg = group ? seg : nullptr; r.wT_ext = group ? ws + W.g_ext : nullptr;
    if (!opt) return gb_run(r);
    // 8 (instead of 8, 9 and the next step's 1): finish + AdamW + re-pack of what the update changed, one launch (ga_opt_step.hip)
    GbDefer df;
    rc = gb_run(r, &df);
    if (rc != ACMIL_OK) return rc;
    return go_launch(gt, &df.g_vu, &df.g_w1, &df.job, KP, packed, t.L, opt->flat, opt->exp_avg, opt->exp_avg_sq, opt->lr, opt->beta1, opt->beta2,
                     opt->eps, opt->wd, opt->step, guard_flag, opt->skipped, opt->flag_report, st);
}

extern "C" int acmil_ga_train_step_rng(const void* x, int x_dtype, int N, void* packed, int repack,
                                   const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu,
                                   const float* Ww, const float* bw, const float* const* Wc, const float* const* bc,
                                   const float* Ws, const float* bs,
                                   float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                                   float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                                   int D, int Di, int Da, int K, int C, int mode,
                                   const int64_t* label, const float* uniforms, int k_top, int m_mask,
                                   float* losses, float* sub_preds, float* slide_pred, float* A_out,
                                   int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream,
                                       unsigned long long rng_seed, unsigned long long rng_offset) {
    return gs_step(x, x_dtype, N, packed, repack, W1, Wv, bv, Wu, bu, Ww, bw, Wc, bc, Ws, bs, dW1, dWv, dbv, dWu, dbu, dWw, dbw, dWc, dbc, dWs,
                   dbs, D, Di, Da, K, C, mode, label, uniforms, k_top, m_mask, losses, sub_preds, slide_pred, A_out, topk_idx, masked_idx,
                   guard_flag, workspace, stream, rng_seed, rng_offset, nullptr);
}

// One training step INCLUDING torch.optim.AdamW's update (Step3_WSI_classification_ACMIL.py:200-219 with :139's optimizer) for a
// single-GPU run: the step's last launch finishes the weight gradients, applies AdamW to every parameter (moments at the same
// offsets of exp_avg / exp_avg_sq as the parameters have in flat_params; skipped on the device when the step's range flag is set, as
// acmil_adamw_step_report does) and rewrites `packed` for the updated values -- so the NEXT call passes repack = 0 unless something
// else changed the parameters.  Gradients are left in the gradient tensors as acmil_ga_train_step leaves them.  Results are
// bit-identical to acmil_ga_train_step_rng + acmil_adamw_step_report + acmil_ga_pack_weights.  ACMIL_ERR_UNSUPPORTED (nothing was
// launched): mode != f16x3, or the parameters are not 16-byte aligned inside the flat buffer; ACMIL_ERR_SHAPE: the parameters do
// not cover flat_params exactly.
extern "C" int acmil_ga_train_step_adamw(const void* x, int x_dtype, int N, void* packed, int repack,
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
                                     double beta2, float eps, float weight_decay, long long step, int* skipped, float* flag_report) {
    GsOpt o;
    o.flat = flat_params; o.n_flat = n_flat; o.exp_avg = exp_avg; o.exp_avg_sq = exp_avg_sq; o.lr = lr; o.beta1 = beta1; o.beta2 = beta2;
    o.eps = eps; o.wd = weight_decay; o.step = step; o.skipped = skipped; o.flag_report = flag_report;
    return gs_step(x, x_dtype, N, packed, repack, W1, Wv, bv, Wu, bu, Ww, bw, (const float* const*)Wc, (const float* const*)bc, Ws, bs, dW1, dWv,
                   dbv, dWu, dbu, dWw, dbw, dWc, dbc, dWs, dbs, D, Di, Da, K, C, mode, label, uniforms, k_top, m_mask, losses, sub_preds,
                   slide_pred, A_out, topk_idx, masked_idx, guard_flag, workspace, stream, rng_seed, rng_offset, &o);
}

extern "C" int acmil_ga_train_step(const void* x, int x_dtype, int N, void* packed, int repack,
                                   const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu,
                                   const float* Ww, const float* bw, const float* const* Wc, const float* const* bc,
                                   const float* Ws, const float* bs,
                                   float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                                   float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                                   int D, int Di, int Da, int K, int C, int mode,
                                   const int64_t* label, const float* uniforms, int k_top, int m_mask,
                                   float* losses, float* sub_preds, float* slide_pred, float* A_out,
                                   int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream) {
    // this entry takes the draw from the caller (_rng draws on the device); shape errors keep their precedence over the null check
    const bool shape_ok = ga_check_dims(D, Di, Da, K, C) == ACMIL_OK && N > 0 && k_top >= 0 && k_top <= 64 && k_top <= N && m_mask >= 0 && m_mask <= k_top;
    if (shape_ok && m_mask > 0 && !uniforms) return ACMIL_ERR_NULL;
    return acmil_ga_train_step_rng(x, x_dtype, N, packed, repack, W1, Wv, bv, Wu, bu, Ww, bw, Wc, bc, Ws, bs, dW1, dWv, dbv, dWu, dbu, dWw, dbw, dWc, dbc, dWs, dbs, D, Di, Da, K, C, mode, label, uniforms, k_top, m_mask, losses, sub_preds, slide_pred, A_out, topk_idx, masked_idx, guard_flag, workspace, stream, 0ull, 0ull);
}

// A GROUP of bags in ONE training step -- the single-GPU twin of slide-level data parallelism (G ranks average the
// gradients of G slides per step; here one GPU does the same for G bags, and under data parallelism every rank takes G bags per
// all-reduce).  Per slide everything is what acmil_ga_train_step computes (Step3_WSI_classification_ACMIL.py:189-221: forward with
// STKIM, three losses, backward); the parameters' gradients are the MEAN over the bags.  The bags' rows lie back to back in x, so
// the score pass and both weight-gradient products run as ONE bag of N_total rows (one launch each, one set of split-K partials, one
// closing launch per G slides); STKIM, pooling, tail and the backward tile kernel cut their tiles per bag.
extern "C" int acmil_ga_train_step_group(const void* x, int x_dtype, int nbags, const int* bag_rows, void* packed, int repack,
                                     float* W1, float* Wv, float* bv, float* Wu, float* bu, float* Ww, float* bw, float* const* Wc,
                                     float* const* bc, float* Ws, float* bs,
                                     float* dW1, float* dWv, float* dbv, float* dWu, float* dbu, float* dWw, float* dbw,
                                     float* const* dWc, float* const* dbc, float* dWs, float* dbs,
                                     int D, int Di, int Da, int K, int C, int mode,
                                     const int64_t* labels, const float* uniforms, int k_top, int m_mask,
                                     float* losses, float* sub_preds, float* slide_pred, float* A_out,
                                     int64_t* topk_idx, int64_t* masked_idx, float* guard_flag, void* workspace, void* stream,
                                     unsigned long long rng_seed, unsigned long long rng_offset, const acmil_adamw_args* adamw) {
    if (!bag_rows) return ACMIL_ERR_NULL;
    if (nbags < 1 || nbags > GA_SEG_MAX) return ACMIL_ERR_SHAPE;
    GaSeg seg; seg.n = nbags; seg.row0[0] = 0;
    long long tot = 0;
    for (int b = 0; b < GA_SEG_MAX; ++b) {
        if (b < nbags) { if (bag_rows[b] <= 0) return ACMIL_ERR_SHAPE; tot += bag_rows[b]; }
        if (tot > 0x7fffffffLL / 2) return ACMIL_ERR_SHAPE;
        seg.row0[b + 1] = (int)tot;
    }
    GsOpt o;
    if (adamw) {
        o.flat = adamw->flat_params; o.n_flat = adamw->n_flat; o.exp_avg = adamw->exp_avg; o.exp_avg_sq = adamw->exp_avg_sq; o.lr = adamw->lr;
        o.beta1 = adamw->beta1; o.beta2 = adamw->beta2; o.eps = adamw->eps; o.wd = adamw->weight_decay; o.step = adamw->step;
        o.skipped = adamw->skipped; o.flag_report = adamw->flag_report;
    }
    return gs_step(x, x_dtype, (int)tot, packed, repack, W1, Wv, bv, Wu, bu, Ww, bw, (const float* const*)Wc, (const float* const*)bc, Ws, bs, dW1, dWv,
                   dbv, dWu, dbu, dWw, dbw, dWc, dbc, dWs, dbs, D, Di, Da, K, C, mode, labels, uniforms, k_top, m_mask, losses, sub_preds,
                   slide_pred, A_out, topk_idx, masked_idx, guard_flag, workspace, stream, rng_seed, rng_offset, adamw ? &o : nullptr, &seg);
}
