// wgrad.hip -- weight gradients of the GA backward: C = A^T B with the contraction over the N patches of the bag
//   [dWv; dWu] = dS^T h     (2 Da x Di,  A = dS [N, 2 Da], B = h [N, Di])
//   dW1        = dpre^T x   (Di x D,     A = dpre [N, Di], B = x [N, D], fp32 / fp16 / bf16 bag)
// (autograd of architecture/transformer.py:305-330; the reference has no explicit backward code).
//
// Both operands are stored with the contraction index as the SLOW axis ([patch][column]), i.e. transposed with respect to
// what an MFMA fragment wants (8 consecutive k per lane).  The generic GEMM (gemm_f32.hip) transposes through 16 scalar
// loads per thread and K step; here the tiles go into LDS exactly as they lie in memory ([k][column], 16-bit hi / lo planes)
// and the fragments come out through the CDNA4 hardware transpose ds_read_b64_tr_b16 -- no scalar loads, no VALU shuffles.
//   * arithmetic: split-bf16 ("bf16x3", as the generic path of the backward): a = hi + lo in bf16, hi*hi + lo*hi + hi*lo on
//     v_mfma_f32_32x32x16_bf16, fp32 accumulate -- gradients keep the fp32 exponent range without scaling;
//   * tile: 128 x 128 outputs per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 each; two workgroups per CU), K step = 32
//     patches -- or, for bags too large for the Infinity Cache (the operands are then re-read from HBM once per output tile that
//     uses them: 546 MB per launch at N = 50 000, PMC, the kernel sat at 6.3 TB/s) and whenever both N are multiples of 256:
//     128 x 256 outputs per 512-thread workgroup (8 waves as 2 x 4, 64 x 64 each; one workgroup per CU): the B operand is
//     read once per 128 output rows, A once per 256 columns -- 357 MB instead of 546 with the same number of split-K partials
//     (88 -> 76 us at N = 50 000).  Tried and dropped: walking the (tile, split) list XCD by XCD so that the tiles of one split share
//     an L2 (115 us: the same lines requested by 4 - 12 workgroups at once serialise in the L2);
//   * pipeline: global loads are issued three K steps ahead into two register sets, the split + LDS store of step s+1 and
//     the MFMAs of step s share one barrier per step (two LDS stages);
//   * split-K over the patches, partials [split][M][N] finished by gemm_finish_kernel (fixed order, gemm_f32.hip); BOTH
//     products are one launch (the grid is the concatenation of their (tile, split) lists).
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "ga_train_internal.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) h16x4* wg_ltr_t;

// LDS geometry of a T-column plane: bytes per k row = T x 2 B + 32 B pad (conflict-free transposed reads), 32 k rows per plane,
// a stage = A hi, A lo, B hi, B lo, two stages
template <int TM, int TN> struct WgGeom {
    static constexpr int ROW_A = TM * 2 + 32, ROW_B = TN * 2 + 32, PLANE_A = 32 * ROW_A, PLANE_B = 32 * ROW_B;
    static constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B, LDS = 2 * STAGE, THREADS = (TM / 64) * (TN / 64) * 64;
};

struct WgProb {
    const float* A; const void* B; float* ws;
    int lda, ldb, b_dtype, M, N, tiles_n, tiles, splits, kchunk;
};
struct WgArgs { WgProb p[2]; int blocks0, K, xcd; };      // xcd != 0: the tiles of a K split share one XCD (see wgrad_kernel)

__device__ __forceinline__ void wg_split_store(char* hi_plane, char* lo_plane, int off, const f32x4 v) {
    unsigned h0, l0, h1, l1;
    ga_split_pair_bf16(v[0], v[1], h0, l0);
    ga_split_pair_bf16(v[2], v[3], h1, l1);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *(u32x2*)(hi_plane + off) = u32x2{h0, h1};
    *(u32x2*)(lo_plane + off) = u32x2{l0, l1};
}

// fragment of a 32-column block (columns c0 .. c0+31 of the plane), k = kb .. kb+15: lane (i = lane & 31, hi = lane >> 5)
// receives plane[kb + 8 hi + j][c0 + i], j < 8.  Supplier lane p = lane & 15 of 16-lane group g addresses row
// kb + 8 (g >> 1) + (p >> 2) (+4 for the second read), columns c0 + 16 (g & 1) + 4 (p & 3) .. +3.
template <int WG_ROW>
__device__ __forceinline__ bf16x8 wg_frag(const char* plane, int rd_off) {
    const h16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((wg_ltr_t)(plane + rd_off));
    const h16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((wg_ltr_t)(plane + rd_off + 4 * WG_ROW));
    typedef h16x4 h16x4_t;
    struct { h16x4_t a, b; } pr = {a0, a1};
    return __builtin_bit_cast(bf16x8, pr);
}

// TM x TN output tile, one wave per 64 x 64 block: 128 x 128 = 4 waves, 128 x 256 = 8 waves
template <int BDT, int TM, int TN>
__device__ __forceinline__ void wg_body(const WgProb& P, int K, int tile, int split, char* smem) {
    using G = WgGeom<TM, TN>;
    constexpr int ROW_A = G::ROW_A, ROW_B = G::ROW_B, PLANE_A = G::PLANE_A, PLANE_B = G::PLANE_B, WG_STAGE = G::STAGE, THREADS = G::THREADS;
    constexpr int WN = TN / 64;                      // waves along the output columns
    constexpr int AT = 2, BT = 2;                    // 32-row / 32-column blocks of a wave's 64 x 64 sub-tile
    constexpr int CPA = TM / 4, RPA = THREADS / CPA, NPA = 32 / RPA;        // A: float4 pieces per k row, rows per pass, passes
    constexpr int CPB = TN / 4, RPB = THREADS / CPB, NPB = 32 / RPB;        // B fp32
    constexpr int CPH = TN / 8, RPH = THREADS / CPH, NPH = 32 / RPH;        // B 16-bit: 16-byte pieces
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = (tile / P.tiles_n) * TM, n0 = (tile % P.tiles_n) * TN;
    const int kbeg = split * P.kchunk;
    const int kend = min(K, kbeg + P.kchunk);
    const int nsteps = (kend - kbeg + 31) / 32;

    // ---- global -> registers: A float4 pieces (row = RPA j + tid / CPA), B fp32 the same with its own geometry, B 16-bit 16-byte pieces
    const int arow = tid / CPA, ac4 = tid % CPA;
    const int frow = tid / CPB, fc4 = tid % CPB;
    const int brow = tid / CPH, bc8 = tid % CPH;
    const float* Ap = P.A + (size_t)m0 + 4 * ac4;
    f32x4 ra[2][NPA], rb[2][NPB];
    u32x4 rh[2][NPH];
    auto load = [&](auto SET, int step) {
        constexpr int S = decltype(SET)::value;
        const int kb = kbeg + 32 * step;
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int k = kb + RPA * j + arow;
            ra[S][j] = (k < kend) ? *(const f32x4*)(Ap + (size_t)k * P.lda) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        if constexpr (BDT == ACMIL_DTYPE_F32) {
            const float* Bp = (const float*)P.B + (size_t)n0 + 4 * fc4;
#pragma unroll
            for (int j = 0; j < NPB; ++j) {
                const int k = kb + RPB * j + frow;
                rb[S][j] = (k < kend) ? *(const f32x4*)(Bp + (size_t)k * P.ldb) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        } else {
            const uint16_t* Bp = (const uint16_t*)P.B + (size_t)n0 + 8 * bc8;
#pragma unroll
            for (int j = 0; j < NPH; ++j) {
                const int k = kb + RPH * j + brow;
                rh[S][j] = (k < kend) ? *(const u32x4*)(Bp + (size_t)k * P.ldb) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    // ---- registers -> LDS planes of stage `stage` ([k][column] as in memory, split into bf16 hi / lo): A hi, A lo, B hi, B lo
    auto store = [&](auto SET, int stage) {
        constexpr int S = decltype(SET)::value;
        char* base = smem + stage * WG_STAGE;
        char* bb = base + 2 * PLANE_A;
#pragma unroll
        for (int j = 0; j < NPA; ++j) wg_split_store(base, base + PLANE_A, (RPA * j + arow) * ROW_A + ac4 * 8, ra[S][j]);
        if constexpr (BDT == ACMIL_DTYPE_F32) {
#pragma unroll
            for (int j = 0; j < NPB; ++j) wg_split_store(bb, bb + PLANE_B, (RPB * j + frow) * ROW_B + fc4 * 8, rb[S][j]);
        } else {
#pragma unroll
            for (int j = 0; j < NPH; ++j) {
                const int off = (RPH * j + brow) * ROW_B + bc8 * 16;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t word = rh[S][j][2 * w + q];
                        if constexpr (BDT == ACMIL_DTYPE_F16) {
                            v[2 * q] = (float)__builtin_bit_cast(_Float16, (uint16_t)(word & 0xffffu));
                            v[2 * q + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(word >> 16));
                        } else {
                            v[2 * q] = __builtin_bit_cast(float, word << 16);
                            v[2 * q + 1] = __builtin_bit_cast(float, word & 0xffff0000u);
                        }
                    }
                    wg_split_store(bb, bb + PLANE_B, off + 8 * w, v);
                }
            }
        }
    };

    f32x16 acc[AT][BT];
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // transposed-read offset of this lane inside a 32-column block (see wg_frag), for either row pitch
    const int p16 = lane & 15, g4 = lane >> 4;
    const int rk = 8 * (g4 >> 1) + (p16 >> 2), rc = (16 * (g4 & 1) + 4 * (p16 & 3)) * 2;
    const int rda = rk * ROW_A + rc, rdb = rk * ROW_B + rc;
    auto compute = [&](int stage) {
        const char* base = smem + stage * WG_STAGE;
        const char* bb = base + 2 * PLANE_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[AT], al[AT], bh[BT], bl[BT];
#pragma unroll
            for (int t = 0; t < AT; ++t) {
                const int oa = rda + 16 * ks * ROW_A + (64 * wm + 32 * t) * 2;
                ah[t] = wg_frag<ROW_A>(base, oa);
                al[t] = wg_frag<ROW_A>(base + PLANE_A, oa);
            }
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const int ob = rdb + 16 * ks * ROW_B + (64 * wn + 32 * t) * 2;
                bh[t] = wg_frag<ROW_B>(bb, ob);
                bl[t] = wg_frag<ROW_B>(bb + PLANE_B, ob);
            }
#pragma unroll
            for (int a = 0; a < AT; ++a)
#pragma unroll
                for (int b = 0; b < BT; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                }
        }
    };

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    if (nsteps > 0) load(S0{}, 0);
    if (nsteps > 1) load(S1{}, 1);
    if (nsteps > 0) store(S0{}, 0);
    if (nsteps > 2) load(S0{}, 2);
    for (int s = 0; s < nsteps; s += 2) {
        ga_lds_barrier();                    // stage 0 holds step s; stage 1 (step s-1) has been consumed by every wave
        if (s + 1 < nsteps) { store(S1{}, 1); if (s + 3 < nsteps) load(S1{}, s + 3); }
        compute(0);
        if (s + 1 >= nsteps) break;
        ga_lds_barrier();
        if (s + 2 < nsteps) { store(S0{}, 0); if (s + 4 < nsteps) load(S0{}, s + 4); }
        compute(1);
    }
    // ---- partial tile -> ws[split][M][N]; acc[a][b][r]: row = m0 + 64 wm + 32 a + mfma32_row(r, hi), col = n0 + 64 wn + 32 b + lane % 32
    float* out = P.ws + (size_t)split * P.M * P.N;
    const int i31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            const int col = n0 + 64 * wn + 32 * b + i31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wm + 32 * a + mfma32_row(r, hi);
                out[(size_t)row * P.N + col] = acc[a][b][r];
            }
        }
}

template <int TM, int TN>
__global__ __launch_bounds__((WgGeom<TM, TN>::THREADS), (TN == 256 ? 1 : 2)) void wgrad_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int pr, tile, split;
    if (a.xcd) {
        // Workgroups are dealt to the 8 XCDs round robin by their index.  ALL output tiles of one K split (both products) go to one
        // XCD and run there side by side (one workgroup per CU, the whole grid is resident): the two tiles that read the same 256
        // columns of x / h, or the same 128 columns of dpre, walk the same rows at the same pace, so the second request for a
        // line is served by that XCD's L2 instead of HBM (each operand line has exactly two readers; with every tile on its own
        // XCD the operands crossed HBM twice: 358 MB per 50 000 rows where they are 205 MB).
        const int tt = a.p[0].tiles + a.p[1].tiles;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        split = x + 8 * (j / tt);
        const int ti = j % tt;
        pr = ti >= a.p[0].tiles ? 1 : 0;
        tile = pr ? ti - a.p[0].tiles : ti;
        if (split >= a.p[0].splits) return;
    } else {
        pr = (int)blockIdx.x >= a.blocks0 ? 1 : 0;
        const int b = blockIdx.x - (pr ? a.blocks0 : 0);
        tile = b % a.p[pr].tiles; split = b / a.p[pr].tiles;
    }
    const WgProb& P = a.p[pr];
    if (P.b_dtype == ACMIL_DTYPE_F32) wg_body<ACMIL_DTYPE_F32, TM, TN>(P, a.K, tile, split, smem);
    else if (P.b_dtype == ACMIL_DTYPE_F16) wg_body<ACMIL_DTYPE_F16, TM, TN>(P, a.K, tile, split, smem);
    else wg_body<ACMIL_DTYPE_BF16, TM, TN>(P, a.K, tile, split, smem);
}

// Column extent of the output tile for a pair of products: 256 whenever both N allow it (measured 23.1 vs 24.0 us at N = 10 000,
// 76 vs 88 us at N = 50 000); ACMIL_WGRAD_TILE=128 keeps the 128 x 128 tiles (A/B measurements)
static int wg_tile_n(int N1, int N2, int K) {
    static const int forced = [] { const char* e = ACMIL_AB_ENV("ACMIL_WGRAD_TILE"); return e ? atoi(e) : 0; }();
    (void)K;
    return (forced == 128 || N1 % 256 || N2 % 256) ? 128 : 256;
}
// K split shared by both products: two (128 x 128) / one (128 x 256) workgroups per CU over the two tile lists, at least 4 K steps per workgroup
static int wg_target_wgs(int TN) { static const int v = [] { const char* e = ACMIL_AB_ENV("ACMIL_WGRAD_WGS"); return e ? atoi(e) : 0; }(); return v > 0 ? v : (TN == 256 ? 256 : 512); }   // (128 x 128, measured 256..768: 512 is 24 us faster than 256 at N = 50 000)
// the tiles of a split on one XCD (128 x 256 tiles only: one workgroup per CU, the grid resident at once); ACMIL_WGRAD_XCD=0: A/B builds
static int wg_xcd() { static const int v = [] { const char* e = ACMIL_AB_ENV("ACMIL_WGRAD_XCD"); return e ? atoi(e) : 1; }(); return v; }
static int wg_pick_splits(int tiles_total, int K, int TN) {
    const int steps = (K + 31) / 32;
    int s = TN == 256 ? wg_target_wgs(TN) / tiles_total : (wg_target_wgs(TN) + tiles_total - 1) / tiles_total;
    if (TN == 256 && wg_xcd() && s >= 8) s = s / 8 * 8;          // whole splits per XCD: tiles_total * s / 8 workgroups on each (<= 32 CUs)
    if (s > steps / 4) s = steps / 4;
    if (s > 128) s = 128;
    return s < 2 ? 0 : s;           // 0: not worth it (tiny bag) -> the caller keeps the generic path
}

size_t wgrad_workspace_bytes(int M1, int N1, int M2, int N2, int K) {
    if (M1 % 128 || N1 % 128 || M2 % 128 || N2 % 128) return 0;
    size_t best = 0;
    for (int TN = 128; TN <= 256; TN += 128) {      // sized for either tile shape (the choice may be forced through the environment)
        if (TN == 256 && (N1 % 256 || N2 % 256)) continue;
        const int s = wg_pick_splits((M1 / 128) * (N1 / TN) + (M2 / 128) * (N2 / TN), K, TN);
        const size_t b = s ? (((size_t)s * ((size_t)M1 * N1 + (size_t)M2 * N2) * sizeof(float) + 255) & ~(size_t)255) : 0;
        if (b > best) best = b;
    }
    return best;
}

template <int TM, int TN>
static int wg_launch_t(WgArgs& a, int grid, hipStream_t st) {
    using G = WgGeom<TM, TN>;
    void (*kern)(WgArgs) = wgrad_kernel<TM, TN>;
    static bool set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ACMIL_ERR_LAUNCH;
    if (!set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return ACMIL_ERR_LAUNCH;
        set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::THREADS), G::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// Launch both products; g1 / g2 receive what gemm_finish needs (C, ldc, M, N, splits, ws).  Returns ACMIL_ERR_UNSUPPORTED when
// the shapes do not fit (the caller then uses the generic GEMM).
int wgrad_launch(const float* A1, int lda1, const void* B1, int b1_dtype, int ldb1, int M1, int N1, float* C1,
                 const float* A2, int lda2, const void* B2, int b2_dtype, int ldb2, int M2, int N2, float* C2,
                 int K, void* workspace, hipStream_t st, GemmArgs* g1, GemmArgs* g2) {
    if (M1 % 128 || N1 % 128 || M2 % 128 || N2 % 128 || K <= 0) return ACMIL_ERR_UNSUPPORTED;
    if ((lda1 & 3) || (lda2 & 3) || (((size_t)A1 | (size_t)A2) & 15)) return ACMIL_ERR_UNSUPPORTED;
    const int al1 = b1_dtype == ACMIL_DTYPE_F32 ? 3 : 7, al2 = b2_dtype == ACMIL_DTYPE_F32 ? 3 : 7;
    if ((ldb1 & al1) || (ldb2 & al2) || (((size_t)B1 | (size_t)B2) & 15)) return ACMIL_ERR_UNSUPPORTED;
    const int TN = wg_tile_n(N1, N2, K);
    const int t1 = (M1 / 128) * (N1 / TN), t2 = (M2 / 128) * (N2 / TN);
    const int s = wg_pick_splits(t1 + t2, K, TN);
    if (!s || !workspace) return ACMIL_ERR_UNSUPPORTED;
    const int steps = (K + 31) / 32;
    const int kchunk = ((steps + s - 1) / s) * 32;
    const int splits = (K + kchunk - 1) / kchunk;
    WgArgs a;
    a.K = K;
    float* ws1 = (float*)workspace;
    float* ws2 = ws1 + (size_t)splits * M1 * N1;
    a.p[0] = WgProb{A1, B1, ws1, lda1, ldb1, b1_dtype, M1, N1, N1 / TN, t1, splits, kchunk};
    a.p[1] = WgProb{A2, B2, ws2, lda2, ldb2, b2_dtype, M2, N2, N2 / TN, t2, splits, kchunk};
    a.blocks0 = t1 * splits;
    a.xcd = (TN == 256 && wg_xcd() && (t1 + t2) * ((splits + 7) / 8) <= 32) ? 1 : 0;
    const int grid = a.xcd ? 8 * (t1 + t2) * ((splits + 7) / 8) : (t1 + t2) * splits;
    const int rc = TN == 256 ? wg_launch_t<128, 256>(a, grid, st) : wg_launch_t<128, 128>(a, grid, st);
    if (rc != ACMIL_OK) return rc;
    auto fill = [&](GemmArgs* g, float* C, int M, int N, float* ws) {
        memset(g, 0, sizeof(*g));
        g->C = C; g->ws = ws; g->M = M; g->N = N; g->K = K; g->ldc = N; g->splits = splits; g->kchunk = kchunk;
        g->alpha = 1.0f; g->beta = 0.0f;
    };
    fill(g1, C1, M1, N1, ws1);
    fill(g2, C2, M2, N2, ws2);
    return ACMIL_OK;
}
