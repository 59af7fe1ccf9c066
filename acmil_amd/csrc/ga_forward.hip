// ga_forward.hip -- host side of the fused GA forward (C ABI: include/acmil_hip.h) plus the two small
// kernels that finish it: ga_merge_kernel (fixed-order combine of the per-workgroup online-softmax
// partials) and ga_heads_kernel (K branch heads + bag head).  The fused kernel itself is
// ga_forward_kernel.h, instantiated per family in ga_forward_inst.hip.
#include <stdio.h>
#include <stdlib.h>

#include "ga_forward_kernel.h"
#include "ga_train_internal.h"

// ------------------------------------------------------------------------------------------------
// merge: afeat[k][:] = (sum_t e^{m_t-M} acc_t) / (sum_t e^{m_t-M} l_t), fixed summation order.
// grid (K, Di/64), 1024 threads = 16 groups of 64 lanes; lane = feature, group g takes tiles g, g+16, ...
// Loads of up to 8 tiles are issued together (the loop is latency-, not bandwidth-bound).
// ------------------------------------------------------------------------------------------------
#define GA_MERGE_GROUPS 16
struct GaBatchTiles { int start[GA_MAX_BATCH + 1]; };

__global__ __launch_bounds__(1024) void ga_merge_kernel(const float* __restrict__ part_all, GaBatchTiles bt, int K, int Di,
                                                        float* __restrict__ afeat_all) {
    const int bag = blockIdx.z;
    const int tiles = bt.start[bag + 1] - bt.start[bag];
    const float* part = part_all + (size_t)bt.start[bag] * K * (2 + Di);
    float* afeat = afeat_all + (size_t)bag * K * Di;
    __shared__ float red[GA_MERGE_GROUPS][66];
    __shared__ float smx[GA_MERGE_GROUPS];
    const int k = blockIdx.x, c = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const size_t PS = 2 + Di;
    const float* base = part + (size_t)k * PS;
    const size_t tstride = (size_t)K * PS;
    float m = -INFINITY;
    for (int t = tid; t < tiles; t += 1024) m = fmaxf(m, base[t * tstride]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) smx[g] = m;
    __syncthreads();
    float M = smx[0];
#pragma unroll
    for (int w = 1; w < GA_MERGE_GROUPS; ++w) M = fmaxf(M, smx[w]);
    float acc = 0.0f, l = 0.0f;
    const int di = 64 * c + lane;
    // (8 tiles per wave group in flight, as ga_tail_kernel: the loop is a chain of dependent round trips -- 391 tiles were 7 of them
    //  with 4 in flight; the order of the sum is unchanged: g, g + 16, g + 32, ...)
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
        afeat[(size_t)k * Di + di] = A / Ls;
    }
}

// ------------------------------------------------------------------------------------------------
// heads: sub_preds[k] = Wc[k] afeat[k] + bc[k]  (transformer.py:325-327, network.py:14-19)
//        bag_feat = mean_k afeat[k]             (== mm(softmax(A).mean(0), h), transformer.py:328-329)
//        slide_pred = Ws bag_feat + bs          (transformer.py:330)
// one workgroup of 16 waves per bag; each wave takes outputs o = wave, wave+16, ...; lanes stride the Di-long dot product
// (K*C + C outputs: 4 waves made this a 25 us serial chain of dependent global loads at C = 7).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ga_heads_kernel(const float* __restrict__ afeat, const char* __restrict__ packed,
                                                       GaLayout L, int has_bag_head, float* __restrict__ sub_preds,
                                                       float* __restrict__ slide_pred, float* __restrict__ bag_feat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* af = (float*)smem;               // [K][Di]
    float* bf = af + (size_t)L.K * L.Di;    // [Di]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = L.K, C = L.C, Di = L.Di;
    const int bag = blockIdx.x;             // batched launch: one workgroup per bag
    afeat += (size_t)bag * K * Di;
    if (sub_preds) sub_preds += (size_t)bag * K * C;
    if (slide_pred) slide_pred += (size_t)bag * C;
    if (bag_feat) bag_feat += (size_t)bag * Di;
    for (int e = tid; e < K * Di; e += 1024) af[e] = afeat[e];
    __syncthreads();
    for (int di = tid; di < Di; di += 1024) {
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s += af[k * Di + di];
        s = s / (float)K;
        bf[di] = s;
        if (bag_feat) bag_feat[di] = s;
    }
    __syncthreads();
    const float* wc = (const float*)(packed + L.wc_off);
    const float* bc = (const float*)(packed + L.bc_off);
    const float* ws = (const float*)(packed + L.ws_off);
    const float* bs = (const float*)(packed + L.bs_off);
    const int nout = K * C + (has_bag_head ? C : 0);
    for (int o = wave; o < nout; o += 16) {
        const float* w; const float* v; float b; float* dst;
        if (o < K * C) { w = wc + (size_t)o * Di; v = af + (size_t)(o / C) * Di; b = bc[o]; dst = sub_preds ? sub_preds + o : nullptr; }
        else { const int c = o - K * C; w = ws + (size_t)c * Di; v = bf; b = bs[c]; dst = slide_pred ? slide_pred + c : nullptr; }
        float s = 0.0f;
        for (int di = lane; di < Di; di += 64) s = fmaf(w[di], v[di], s);
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1) s += __shfl_xor(s, o2);
        if (lane == 0 && dst) *dst = s + b;
    }
}

// ------------------------------------------------------------------------------------------------ host side
#define GA_FAMILY(ND, KP, MODE) int ga_fwd_family_##ND##_##KP##_##MODE(const GaFwdArgs&, int, bool, int, hipStream_t);
#include "ga_families.inc"
#undef GA_FAMILY
#define GA3_FAMILY(ND, PB, KP) int ga_fwd3_family_##ND##_##PB##_##KP(const GaFwdArgs&, int, bool, hipStream_t);
#include "ga_families3.inc"
#undef GA3_FAMILY

// kernel generation of the split-f16 families: 2 = software-pipelined loops (default), 1 = the first-generation kernel.
// ACMIL_GA_KERNEL=1|2 overrides (A/B measurements); read once.
static int ga_kernel_version() {
    static const int v = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA_KERNEL"); return (e && atoi(e) == 1) ? 1 : 2; }();
    return v;
}

// the persistent split-f16 kernel covers every built family in F16X3 mode
static bool ga_use_v2(int mode) { return mode == ACMIL_MODE_F16X3 && ga_kernel_version() == 2; }

// start delay of the second workgroup of each CU (ga_forward_kernel_v2.h), in s_sleep(127) rounds.  Measured on MI355X: offsets up
// to a full tile change nothing (the two workgroups of a CU drift through every relative phase anyway, the second one
// being ~25 % slower), so the default is 0; ACMIL_GA_DEPHASE keeps the knob for experiments.
static int ga_dephase() {
    static const int env = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA_DEPHASE"); return e ? atoi(e) : 0; }();
    return env > 0 ? env : 0;
}

// A/B knob (timing experiments): ACMIL_GA_MEMSET=1 zeroes the control block with a memset ahead of every launch and the
// kernel skips its end-of-kernel counter reset (the round-2 scheme before the self-resetting block)
static bool ga_memset_mode() { static const bool v = ACMIL_AB_ENV("ACMIL_GA_MEMSET") != nullptr; return v; }

// Tile geometry of the persistent split-f16 kernel, chosen PER LAUNCH (round 5): 4 waves (128-patch tiles, two workgroups per CU) or
// 8 waves (256-patch tiles, one workgroup per CU: the weight stream is staged once per 256 patches).  Measured with
// tools/time_single_bag.py, us per forward incl. merge + heads, 4 / 8 waves: one bag of 313 tiles 78.3 / 72.6, 391 (N = 50 000):
// 85.4 / 79.8, 512: 96.9 / 91.8, 782: 160.6 / 155.9 -- but 256 tiles: 57.1 / 68.7, a single tile: 43.8 / 62.5, and the batched
// launches (6 256 tiles): 872 / 896.  So: 8 waves when the launch holds 257 .. 1 024 tiles of 128 patches (one large slide per
// call -- the reference's own call pattern, Step3_WSI_classification_ACMIL.py:193-200,253-258), 4 waves otherwise.  The tile
// partition is the summation order of the pooled features, so a bag's logits may differ in the last bits (<= 2e-6 measured; the
// contract is 1e-4) with what shares its launch; per-patch scores are bit-identical in both geometries, and a given launch is
// bit-reproducible.  ACMIL_GA2_WAVES=4|8 forces (A/B builds).
static int ga_v2_waves(long long tiles128, int Di) {
    static const int v = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA2_WAVES"); return e ? atoi(e) : 0; }();
    if (v == 4 || v == 8) return v;
    // (D_inner = 256 only: that is where the table above was measured; the 128-wide family keeps its three 4-wave workgroups per CU --
    //  one bf16 bag of 391 tiles: 50.8 us with them, 52.4 with one 8-wave workgroup)
    return (Di == 256 && tiles128 > 256 && tiles128 <= 1024) ? 8 : 4;
}

// wave-pair split of GEMM1 (ga_forward_kernel_v2.h); ACMIL_GA2_PAIR=0|1 overrides (A/B measurements); read once
static int ga_pair_split() {
    static const int v = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA2_PAIR"); return e ? (atoi(e) != 0) : 0; }();
    return v;
}

// A/B builds: two workgroups per CU for the D_inner = 128 family on 16-bit bags (default three)
static int ga_no_tri() { static const int v = ACMIL_AB_ENV("ACMIL_GA2_NO_TRI") != nullptr; return v; }

// The one-wave-per-SIMD kernel (ga_forward_kernel_v3.h, split-f16 arithmetic only): the only fused kernel of the wide families
// (D_inner 384 / 512: 32 patches per wave); at D_inner = 256 its 64-patch wave tile is chosen per launch by ga_pick_geometry().
static bool ga_is_wide(int Di) { return Di == 384 || Di == 512; }
static bool ga_has_v3(int Di, int K, int mode, int x_dtype = ACMIL_DTYPE_F16) {
    return mode == ACMIL_MODE_F16X3 && K <= 5 && (ga_is_wide(Di) || Di == 256 || (Di == 128 && x_dtype != ACMIL_DTYPE_F32));
}
// 32-patch blocks per wave of the v3 family of a width: (16, 1), (12, 1), (8, 2), (4, 3: 16-bit bags only)
static int ga_v3_pb(int Di) { return Di == 256 ? 2 : Di == 128 ? 3 : 1; }

// Geometry of a pooled (eval) launch.  Wide families: always v3.  D_inner = 256: ACMIL_GA3=0|1 forces (A/B builds); default below.
static int ga_pick_v3(int Di, int K, int mode, int x_dtype, int nbags, long long total_patches) {
    if (!ga_has_v3(Di, K, mode, x_dtype)) return 0;
    if (ga_is_wide(Di)) return 1;
    static const int env = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA3"); return e ? atoi(e) : -1; }();
    if (env == 0 || env == 1) return env;
    (void)nbags; (void)total_patches;
    return 0;
}

static int ga_dispatch(const GaFwdArgs& a, int mode, int x_dtype, bool pool, hipStream_t st) {
    const int ND = a.L.ND, K = a.L.K;
    const int KP = (K <= 1) ? 1 : (K <= 5) ? 5 : 8;
    if (a.v3) {
        if (mode != ACMIL_MODE_F16X3) return ACMIL_ERR_UNSUPPORTED;
        const int PB = ga_v3_pb(32 * ND);
#define GA3_FAMILY(ND_, PB_, KP_) \
        if (ND == ND_ && PB == PB_ && KP == KP_) return ga_fwd3_family_##ND_##_##PB_##_##KP_(a, x_dtype, pool, st);
#include "ga_families3.inc"
#undef GA3_FAMILY
        return ACMIL_ERR_UNSUPPORTED;
    }
#define GA_FAMILY(ND_, KP_, MODE_) \
    if (ND == ND_ && KP == KP_ && mode == MODE_) return ga_fwd_family_##ND_##_##KP_##_##MODE_(a, x_dtype, pool, ga_kernel_version(), st);
#include "ga_families.inc"
#undef GA_FAMILY
    return ACMIL_ERR_UNSUPPORTED;
}

// One-off initialisation of a freshly allocated GA workspace: zeroes the 256-byte control block (tile counter, arrival counters,
// range words).  Every launch leaves the counters at zero again, so this is needed ONCE per allocation, not per call; a workspace
// that was never initialised starts with garbage counters and the persistent kernels would skip or repeat tiles.
extern "C" int acmil_ga_workspace_init(void* workspace, void* stream) {
    if (!workspace) return ACMIL_ERR_NULL;
    return hipMemsetAsync(workspace, 0, GA_CTRL_BYTES, (hipStream_t)stream) == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" size_t acmil_ga_workspace_bytes(int N, int D, int Di, int K, int C, int mode) {
    (void)D; (void)C; (void)mode;
    if (N <= 0 || Di <= 0 || K <= 0) return 0;
    size_t b = (size_t)ga_pool_tiles(N) * K * ga_part_stride(Di) * sizeof(float);  // partials (pool tiles >= fused tiles)
    b = (b + 255) & ~(size_t)255;
    b += (size_t)K * Di * sizeof(float);                                          // afeat scratch
    b = (b + 255) & ~(size_t)255;
    return GA_CTRL_BYTES + b;                                                     // control block first (ga_common.h)
}

// merge + heads shared by the fused forward (batched) and the masked pooling pass.
// part: partials of all bags back to back; tile_start[b]: first tile of bag b; afeat scratch lives after the partials.
int ga_finish_batch(const float* part, const int* tile_start, int nbags, const void* packed, const GaLayout& L,
                    float* sub_preds, float* slide_pred, float* afeat, float* bag_feat, int has_bag_head,
                    float* af_scratch, hipStream_t st) {
    const int K = L.K, Di = L.Di;
    GaBatchTiles bt;
    for (int b = 0; b <= GA_MAX_BATCH; ++b) bt.start[b] = b <= nbags ? tile_start[b] : tile_start[nbags];
    float* af = afeat ? afeat : af_scratch;
    hipLaunchKernelGGL(ga_merge_kernel, dim3(K, Di / 64, nbags), dim3(1024), 0, st, part, bt, K, Di, af);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    if (sub_preds || slide_pred || bag_feat) {
        const size_t lds = ((size_t)K * Di + Di) * sizeof(float);
        hipLaunchKernelGGL(ga_heads_kernel, dim3(nbags), dim3(1024), lds, st, af, (const char*)packed, L, has_bag_head,
                           sub_preds, slide_pred, bag_feat);
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    }
    return ACMIL_OK;
}

int ga_finish(const float* part, int tiles, const void* packed, const GaLayout& L, float* sub_preds,
              float* slide_pred, float* afeat, float* bag_feat, int has_bag_head, hipStream_t st) {
    const int ts[2] = {0, tiles};
    size_t poff = ((size_t)tiles * L.K * ga_part_stride(L.Di) * sizeof(float) + 255) & ~(size_t)255;   // afeat scratch follows the partials
    return ga_finish_batch(part, ts, 1, packed, L, sub_preds, slide_pred, afeat, bag_feat, has_bag_head,
                           (float*)((char*)part + poff), st);
}

static int ga_pick_waves(int maxN, long long total_patches = 0) {
    // tile geometry: 8-wave (256-patch) workgroups stream the weights once per 256 patches and give the lowest latency for
    // one large bag; 4-wave (128-patch) workgroups, two per CU, spread small bags over more CUs and -- measured, 8 x 50 000
    // patches per launch: 524 vs 548 us -- balance better once a launch holds several rounds of tiles (>= 1024 of them).
    // ACMIL_GA_WAVES=4|8 overrides (tuning).
    static const int env = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA_WAVES"); return e ? atoi(e) : 0; }();   // read once
    if (env == 4 || env == 8) return env;
    return (maxN >= 32768 && total_patches < 1024LL * 128) ? 8 : 4;
}

extern "C" size_t acmil_ga_batch_workspace_bytes(int nbags, const int* Ns, int D, int Di, int K, int C, int mode) {
    (void)D; (void)C; (void)mode;
    if (nbags <= 0 || nbags > GA_MAX_BATCH || !Ns || Di <= 0 || K <= 0) return 0;
    size_t tiles = 0;
    for (int b = 0; b < nbags; ++b) { if (Ns[b] <= 0) return 0; tiles += ga_pool_tiles(Ns[b]); }
    size_t bytes = (tiles * K * ga_part_stride(Di) * sizeof(float) + 255) & ~(size_t)255;
    bytes += ((size_t)nbags * K * Di * sizeof(float) + 255) & ~(size_t)255;
    return GA_CTRL_BYTES + bytes;
}

// ---------------------------------------------------------------------------------------------------------------
// Device-side range guard of the WIDE families (round 5).  The fused widths have an exact-fp32 twin of the fused kernel (v1) that is
// enqueued behind the split-f16 launch and leaves at once unless the status word is set; D_inner 384 / 512 have no such twin (its 256
// accumulator registers per wave need the one-wave-per-SIMD geometry).  Their repeat is the op-by-op chain in exact fp32 arithmetic --
// [widen a 16-bit bag] -> h = relu(x W1^T) -> G = h [Wv;Wu]^T + b -> gate pass -> pooling partials -- every launch of it predicated on
// the status word of the fused launch (4 - 5 launches that exit at once for an in-range bag: ~3 us each), writing the SAME outputs
// (scores, 128-patch tile partials); merge + heads then finish whichever result is there.  No host read-back on the per-slide path.
// The only operand the packed buffer does not hold in raw fp32 is W1; scratch (caller-owned) holds x32 / h / G / [scores] / GEMM workspace.
struct GaWideRepeat { const float* W1; void* scratch; unsigned* fallback_count; };
struct GaWideScratch { size_t x32, h, A, G, gws, total; };
extern "C" size_t acmil_gemm_workspace_bytes(int M, int N, int K, int batch);
static GaWideScratch ga_wide_scratch(int N, int D, int Di, int K, int x_dtype, bool need_scores) {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    GaWideScratch S;
    size_t off = 0;
    S.x32 = off; off += x_dtype == ACMIL_DTYPE_F32 ? 0 : al((size_t)N * D * 4);
    S.h = off;   off += al((size_t)N * Di * 4);
    S.A = off;   off += need_scores ? al((size_t)K * N * 4) : 0;
    S.G = off;   off += al((size_t)N * 2 * GA_DA * 4);
    const size_t g1 = acmil_gemm_workspace_bytes(N, Di, D, 1), g2 = acmil_gemm_workspace_bytes(N, 2 * GA_DA, Di, 1);
    S.gws = off; off += al(g1 > g2 ? g1 : g2);
    S.total = off;
    return S;
}
__global__ __launch_bounds__(256) void ga_widen_cond_kernel(const void* __restrict__ x, int x_dtype, long long n, float* __restrict__ y,
                                                            const unsigned* __restrict__ cond) {
    if (__builtin_nontemporal_load(cond) == 0u) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = x_dtype == ACMIL_DTYPE_F16 ? (float)((const _Float16*)x)[i] : (float)((const __bf16*)x)[i];
}
static int ga_wide_fp32_repeat(const void* x, int x_dtype, int N, const char* packed, const GaLayout& L, const GaWideRepeat& w,
                               float* A_out, float* part, const unsigned* cond, hipStream_t st) {
    const GaWideScratch S = ga_wide_scratch(N, L.D, L.Di, L.K, x_dtype, A_out == nullptr);
    char* sc = (char*)w.scratch;
    const float* x32 = (const float*)x;
    if (x_dtype != ACMIL_DTYPE_F32) {
        hipLaunchKernelGGL(ga_widen_cond_kernel, dim3(1024), dim3(256), 0, st, x, x_dtype, (long long)N * L.D, (float*)(sc + S.x32), cond);
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
        x32 = (const float*)(sc + S.x32);
    }
    float* h = (float*)(sc + S.h);
    int rc = gemm_f32_cond(0, 1, N, L.Di, L.D, 1.0f, x32, L.D, w.W1, ACMIL_DTYPE_F32, L.D, h, L.Di, nullptr, 1 /* relu */, sc + S.gws, st, cond);
    if (rc != ACMIL_OK) return rc;
    float* A = A_out ? A_out : (float*)(sc + S.A);
    const float* tab = (const float*)(packed + L.tab_off);
    rc = ag_gated_scores_cond(h, N, L.Di, GA_DA, L.K, (const float*)(packed + L.wcat_off), (const float*)(packed + L.bcat_off),
                              tab + 2 * GA_DA, (const float*)(packed + L.bw_off), A, (float*)(sc + S.G), sc + S.gws, st, cond);
    if (rc != ACMIL_OK) return rc;
    return ga_pool_launch(h, A, N, L.K, L.Di, part, nullptr, st, cond, w.fallback_count);
}

// packed_fp32 != null (split-f16 mode only): the DEVICE-SIDE range guard -- right behind the split-f16 launch an exact-fp32 launch of
// the same tiles is enqueued that every workgroup leaves at once unless the status word says a bag left the f16 range; it
// then overwrites the scores and the partials, and merge + heads finish whichever result is there.  No host read-back.
static int ga_forward_batch_impl(int nbags, const void* const* xs, const int* Ns, int x_dtype, const void* packed,
                                 int D, int Di, int Da, int K, int C, int mode, float* const* A_outs,
                                 float* sub_preds, float* slide_pred, float* afeat, float* bag_feat,
                                 int has_bag_head, void* workspace, void* stream, const void* packed_fp32, unsigned* fallback_count,
                                 const GaWideRepeat* wide = nullptr) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (nbags <= 0 || nbags > GA_MAX_BATCH) return ACMIL_ERR_SHAPE;
    if (!xs || !Ns || !packed || !workspace) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    GaFwdArgs a;
    int maxN = 0;
    for (int b = 0; b < nbags; ++b) {
        if (Ns[b] <= 0) return ACMIL_ERR_SHAPE;
        if (!xs[b]) return ACMIL_ERR_NULL;
        if (Ns[b] > maxN) maxN = Ns[b];
    }
    long long total_patches = 0;
    for (int b = 0; b < nbags; ++b) total_patches += Ns[b];
    long long tiles128 = 0;
    for (int b = 0; b < nbags; ++b) tiles128 += (Ns[b] + 127) / 128;
    a.waves = ga_use_v2(mode) ? ga_v2_waves(tiles128, Di) : ga_pick_waves(maxN, total_patches);
    a.dephase = ga_dephase(); a.pair_split = ga_pair_split(); a.no_tri = ga_no_tri();
    a.v3 = ga_pick_v3(Di, K, mode, x_dtype, nbags, total_patches);
    if (packed_fp32 && Di == 128) a.v3 = 0;            // the exact-fp32 repeat (v1 kernel) shares the tile partition: 128- / 256-patch tiles only
    if (a.v3) a.waves = 4 * ga_v3_pb(Di);              // tile rows = 32 * waves: 128 (32 patches per wave) / 256 (64) / 384 (96)
    if (a.v3 && !ga_use_v2(mode)) return ACMIL_ERR_UNSUPPORTED;
    a.tile_start[0] = 0;
    for (int b = 0; b < GA_MAX_BATCH; ++b) {
        a.xs[b] = b < nbags ? xs[b] : nullptr;
        a.Ns[b] = b < nbags ? Ns[b] : 0;
        a.A_outs[b] = (b < nbags && A_outs) ? A_outs[b] : nullptr;
        a.tile_start[b + 1] = a.tile_start[b] + (b < nbags ? (Ns[b] + 32 * a.waves - 1) / (32 * a.waves) : 0);
    }
    a.nbags = nbags; a.packed = (const char*)packed; a.part = (float*)((char*)workspace + GA_CTRL_BYTES); a.h_save = nullptr;
    a.tile_counter = nullptr; a.status = nullptr; a.cond = nullptr; a.cond_count = nullptr;
    if (ga_use_v2(mode)) { a.tile_counter = (unsigned*)workspace; a.status = a.tile_counter + 1; }   // control block, see ga_common.h
    a.self_reset = ga_memset_mode() ? 0 : 1;
    if (a.tile_counter && !a.self_reset && hipMemsetAsync(a.tile_counter, 0, 16, st) != hipSuccess) return ACMIL_ERR_LAUNCH;
    a.L = ga_layout(D, Di, K, C, mode);
    rc = ga_dispatch(a, mode, x_dtype, true, st);
    if (rc != ACMIL_OK) return rc;
    if (packed_fp32) {
        if (!a.status) return ACMIL_ERR_UNSUPPORTED;          // only the persistent split-f16 kernel publishes a status word
        GaFwdArgs b = a;                                       // same bags, same 128-patch tiles, same partial slots
        b.packed = (const char*)packed_fp32; b.L = ga_layout(D, Di, K, C, ACMIL_MODE_F32);
        b.tile_counter = nullptr; b.status = nullptr; b.cond = a.status; b.cond_count = fallback_count; b.v3 = 0;
        rc = ga_dispatch(b, ACMIL_MODE_F32, x_dtype, true, st);
        if (rc != ACMIL_OK) return rc;
    }
    if (wide) {      // the wide families' repeat: op by op in exact fp32, every launch predicated on this launch's status word
        if (nbags != 1 || !a.v3 || !ga_is_wide(Di) || !a.status || Da != GA_DA) return ACMIL_ERR_UNSUPPORTED;
        rc = ga_wide_fp32_repeat(xs[0], x_dtype, Ns[0], (const char*)packed, a.L, *wide, A_outs ? A_outs[0] : nullptr, a.part, a.status, st);
        if (rc != ACMIL_OK) return rc;
    }
    if (!(sub_preds || slide_pred || afeat || bag_feat)) return ACMIL_OK;
    const size_t poff = ((size_t)a.tile_start[nbags] * K * ga_part_stride(Di) * sizeof(float) + 255) & ~(size_t)255;
    // (measured: the single-launch finish of ga_step.hip -- ga_tail_eval -- costs 26 us against 12 us for these two launches at one
    // bag: its release / acquire fences and the one-workgroup heads outweigh the saved launch; it pays off only where it
    // replaces the six further launches of a training step.  ACMIL_GA_TAIL_EVAL=1 selects it for experiments.)
    static const bool tail_eval = ACMIL_AB_ENV("ACMIL_GA_TAIL_EVAL") != nullptr;
    if (tail_eval && K <= 5)
        return ga_tail_eval(a.part, a.tile_start, nbags, packed, a.L, sub_preds, slide_pred, afeat ? afeat : (float*)((char*)a.part + poff),
                            bag_feat, has_bag_head, (unsigned*)workspace + 8, st);
    return ga_finish_batch(a.part, a.tile_start, nbags, packed, a.L, sub_preds, slide_pred, afeat, bag_feat, has_bag_head,
                           (float*)((char*)a.part + poff), st);
}

extern "C" int acmil_ga_forward_batch(int nbags, const void* const* xs, const int* Ns, int x_dtype, const void* packed,
                                      int D, int Di, int Da, int K, int C, int mode, float* const* A_outs,
                                      float* sub_preds, float* slide_pred, float* afeat, float* bag_feat,
                                      int has_bag_head, void* workspace, void* stream) {
    return ga_forward_batch_impl(nbags, xs, Ns, x_dtype, packed, D, Di, Da, K, C, mode, A_outs, sub_preds, slide_pred, afeat, bag_feat,
                                 has_bag_head, workspace, stream, nullptr, nullptr);
}

extern "C" int acmil_ga_forward_guarded(int nbags, const void* const* xs, const int* Ns, int x_dtype, const void* packed_f16x3,
                                        const void* packed_fp32, int D, int Di, int Da, int K, int C, float* const* A_outs,
                                        float* sub_preds, float* slide_pred, float* afeat, float* bag_feat, int has_bag_head,
                                        unsigned* fallback_count, void* workspace, void* stream) {
    if (!packed_fp32) return ACMIL_ERR_NULL;
    if (!ga_use_v2(ACMIL_MODE_F16X3)) return ACMIL_ERR_UNSUPPORTED;
    return ga_forward_batch_impl(nbags, xs, Ns, x_dtype, packed_f16x3, D, Di, Da, K, C, ACMIL_MODE_F16X3, A_outs, sub_preds, slide_pred,
                                 afeat, bag_feat, has_bag_head, workspace, stream, packed_fp32, fallback_count);
}

// The same predicated repeat for the COMPOSED path (D_inner = 768, n_token > 5: projection kernel -> gated scores -> pooling as separate
// calls): after the split-f16 projection (whose launch left `cond`, its range status word) and the gated-score pass, this call OVERWRITES
// h [N, Di] and A [K, N] with their exact-fp32 values if -- and only if -- *cond != 0; the caller's pooling / merge / heads then run on
// whichever values are there.  Eval only: a training step must know on the host which arithmetic its backward has to use.
extern "C" size_t acmil_ga_rescore_fp32_cond_scratch_bytes(int N, int D, int Di, int x_dtype) {
    if (N <= 0 || D <= 0 || Di <= 0) return 0;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t g1 = acmil_gemm_workspace_bytes(N, Di, D, 1), g2 = acmil_gemm_workspace_bytes(N, 2 * GA_DA, Di, 1);
    return (x_dtype == ACMIL_DTYPE_F32 ? 0 : al((size_t)N * D * 4)) + al((size_t)N * 2 * GA_DA * 4) + al(g1 > g2 ? g1 : g2);
}

extern "C" int acmil_ga_rescore_fp32_cond(const void* x, int x_dtype, int N, const void* packed, const float* W1, int D, int Di, int Da, int K,
                                          int C, int mode, float* h, float* A, const unsigned* cond, unsigned* fallback_count, void* scratch,
                                          void* stream) {
    int rc0 = ga_check_dims(D, Di, Da, K, C);
    if (rc0 != ACMIL_OK) return rc0;
    if (N <= 0) return ACMIL_ERR_SHAPE;
    if (!x || !packed || !W1 || !h || !A || !cond || !scratch) return ACMIL_ERR_NULL;
    // the raw fp32 copies inside the packed buffer (ga_common.h: wcat = [Wv; Wu] [2 Da, Di], bcat, the Ww rows of the table, bw)
    const GaLayout L = ga_layout(D, Di, K, C, mode);
    const float* wcat = (const float*)((const char*)packed + L.wcat_off);
    const float* bcat = (const float*)((const char*)packed + L.bcat_off);
    const float* Ww = (const float*)((const char*)packed + L.tab_off) + 2 * GA_DA;
    const float* bw = (const float*)((const char*)packed + L.bw_off);
    if (((size_t)scratch & 255) != 0) return ACMIL_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char* sc = (char*)scratch;
    size_t off = 0;
    const float* x32 = (const float*)x;
    if (x_dtype != ACMIL_DTYPE_F32) {
        hipLaunchKernelGGL(ga_widen_cond_kernel, dim3(1024), dim3(256), 0, st, x, x_dtype, (long long)N * D, (float*)sc, cond);
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
        x32 = (const float*)sc;
        off += al((size_t)N * D * 4);
    }
    float* G = (float*)(sc + off); off += al((size_t)N * 2 * Da * 4);
    void* gws = sc + off;
    int rc = gemm_f32_cond(0, 1, N, Di, D, 1.0f, x32, D, W1, ACMIL_DTYPE_F32, D, h, Di, nullptr, 1 /* relu */, gws, st, cond);
    if (rc != ACMIL_OK) return rc;
    return ag_gated_scores_cond(h, N, Di, Da, K, wcat, bcat, Ww, bw, A, G, gws, st, cond, fallback_count);
}

extern "C" size_t acmil_ga_forward_guarded_wide_scratch_bytes(int N, int D, int Di, int K, int x_dtype, int need_scores) {
    if (N <= 0 || D <= 0 || Di <= 0 || K <= 0) return 0;
    return ga_wide_scratch(N, D, Di, K, x_dtype, need_scores != 0).total;
}

extern "C" int acmil_ga_forward_guarded_wide(const void* x, int x_dtype, int N, const void* packed_f16x3, const float* W1, int D, int Di,
                                             int Da, int K, int C, float* A_out, float* sub_preds, float* slide_pred, float* afeat,
                                             float* bag_feat, int has_bag_head, unsigned* fallback_count, void* scratch, void* workspace,
                                             void* stream) {
    if (!W1 || !scratch || !x) return ACMIL_ERR_NULL;
    if (!ga_is_wide(Di) || K > 5) return ACMIL_ERR_UNSUPPORTED;
    if (((size_t)scratch & 255) != 0) return ACMIL_ERR_SHAPE;
    const GaWideRepeat w = {W1, scratch, fallback_count};
    float* A1[1] = {A_out};
    const void* x1[1] = {x};
    return ga_forward_batch_impl(1, x1, &N, x_dtype, packed_f16x3, D, Di, Da, K, C, ACMIL_MODE_F16X3, A1, sub_preds, slide_pred, afeat,
                                 bag_feat, has_bag_head, workspace, stream, nullptr, nullptr, &w);
}

extern "C" int acmil_ga_forward(const void* x, int x_dtype, int N, const void* packed, int D, int Di, int Da, int K,
                                int C, int mode, float* A_out, float* sub_preds, float* slide_pred, float* afeat,
                                float* bag_feat, float* h_save, int has_bag_head, void* workspace, void* stream) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (N <= 0) return ACMIL_ERR_SHAPE;
    if (!x || !packed) return ACMIL_ERR_NULL;
    const bool pool = sub_preds || slide_pred || afeat || bag_feat;
    if (pool && !workspace) return ACMIL_ERR_NULL;
    if (pool && h_save) return ACMIL_ERR_UNSUPPORTED;  // score pass (h_save) and pooled outputs are separate calls
    if (!pool && !h_save && !A_out) return ACMIL_ERR_NULL;
    if (!h_save) {
        // eval / scores-only: the batched path with one bag (scores-only drops the partials it writes)
        if (!workspace) return ACMIL_ERR_NULL;
        float* A1[1] = {A_out};
        const void* x1[1] = {x};
        return acmil_ga_forward_batch(1, x1, &N, x_dtype, packed, D, Di, Da, K, C, mode, A1, sub_preds, slide_pred, afeat,
                                      bag_feat, has_bag_head, workspace, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    GaFwdArgs a;
    a.waves = ga_use_v2(mode) ? 4 : ga_pick_waves(N);      // score pass: 8 waves gain nothing for it (0.3154 vs 0.3153 ms per training step)
    a.dephase = ga_dephase(); a.pair_split = ga_pair_split(); a.no_tri = ga_no_tri();
    a.v3 = (ga_is_wide(Di) && ga_has_v3(Di, K, mode) && workspace) ? 1 : 0;      // score pass: the wide families only
    if (a.v3) a.waves = 4;
    for (int b = 0; b < GA_MAX_BATCH; ++b) { a.xs[b] = nullptr; a.Ns[b] = 0; a.A_outs[b] = nullptr; a.tile_start[b + 1] = 0; }
    a.xs[0] = x; a.Ns[0] = N; a.A_outs[0] = A_out; a.tile_start[0] = 0;
    for (int b = 1; b <= GA_MAX_BATCH; ++b) a.tile_start[b] = (N + 32 * a.waves - 1) / (32 * a.waves);
    a.nbags = 1; a.packed = (const char*)packed; a.part = workspace ? (float*)((char*)workspace + GA_CTRL_BYTES) : nullptr; a.h_save = h_save;
    a.tile_counter = nullptr; a.status = nullptr; a.cond = nullptr; a.cond_count = nullptr;
    if (ga_use_v2(mode) && workspace) { a.tile_counter = (unsigned*)workspace; a.status = a.tile_counter + 1; }
    a.self_reset = ga_memset_mode() ? 0 : 1;
    if (a.tile_counter && !a.self_reset && hipMemsetAsync(a.tile_counter, 0, 16, st) != hipSuccess) return ACMIL_ERR_LAUNCH;
    a.L = ga_layout(D, Di, K, C, mode);
    return ga_dispatch(a, mode, x_dtype, false, st);
}

// MFMA-only probe (include/acmil_hip.h: acmil_mfma_probe): what the matrix pipe delivers under the power cap, measured by bench.py
// beside every roofline line.  Operands are pseudo-random f16 (zero-filled operands clock ~19 % higher: MI355X_MICROARCH.md, DVFS).
__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(int iters, float* __restrict__ sink) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    f16x8 A[4], B[2];
    unsigned s = t * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; A[i][j] = (_Float16)(((int)(s >> 16) - 32768) * (1.0f / 1048576.0f)); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s = s * 1664525u + 1013904223u; B[i][j] = (_Float16)(((int)(s >> 16) - 32768) * (1.0f / 32768.0f)); }
    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[q & 3], B[q >> 2], acc[q], 0, 0, 0);
    }
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) v += acc[q][r];
    sink[t] = v;
}

extern "C" int acmil_mfma_probe(int iters, int workgroups, float* sink, long long* mfmas, void* stream) {
    if (iters <= 0 || workgroups <= 0) return ACMIL_ERR_SHAPE;
    if (!sink) return ACMIL_ERR_NULL;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink);
    if (mfmas) *mfmas = (long long)workgroups * 4 * iters * 8;
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" const char* acmil_version(void) { return "acmil_hip 0.1 (gfx950)"; }

extern "C" int acmil_check_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ACMIL_ERR_ARCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ACMIL_ERR_ARCH;
    const char* n = prop.gcnArchName;
    return (n[0] == 'g' && n[1] == 'f' && n[2] == 'x' && n[3] == '9' && n[4] == '5' && n[5] == '0') ? ACMIL_OK : ACMIL_ERR_ARCH;
}
