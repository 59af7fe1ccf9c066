// ga_bwd_tile.hip -- the gate side of the GA backward as ONE kernel per 64-patch tile (replaces three launches of a training
// step: the G recompute GEMM, the gate pass and the dpre GEMM -- 91 of 240 us at N = 10 000, 210 of 490 us at N = 50 000 --
// and the HBM round trips of G [N,256] and dh0 [N,Di] between them).  Autograd of architecture/transformer.py:259-267 and
// :322-324 w.r.t. h (no explicit backward code in the reference):
//
//   1  G   = h [Wv;Wu]^T + [bv;bu]                      64 x 256, K = Di      split-f16 MFMA (forward-sized values)
//   2  gate pass, one wave per patch, G resident in LDS  (same arithmetic as ga_bwd_gate_kernel, ga_backward.hip):
//        P = softmax prob., dA = P (d_afeat . h - c) + diversity term (0 where masked), V = tanh, U = sigmoid,
//        dg = dA^T Ww, dS_v = dg U (1 - V^2), dS_u = dg V U (1 - U)  -> dS overwrites G in LDS and goes to HBM for the
//        weight-gradient kernel; partial sums of dWw, dbw, dbv, dbu per workgroup
//   3  dpre = ([dS | P] [[Wv;Wu]^T | d_afeat^T]^T) * [h > 0]      64 x Di, K = 256 + 16    split-bf16 MFMA
//      (the pooling term dh0 = P d_afeat rides along as 16 extra K slots, so dh0 never exists)
//
// Operands that do not depend on the bag are pre-split once per step: [Wv;Wu] as f16 hi / lo planes [256][Di] and
// [[Wv;Wu]^T | d_afeat^T | 0] as bf16 hi / lo planes [Di][288] (ga_pack.hip; the d_afeat columns are filled by the tail
// kernel, ga_step.hip), so their staging is plain 16-byte copies.  h is split while it is staged; [dS | P] is split from
// the fp32 LDS tile when the A fragments are formed.
// LDS (152 KB, one workgroup of 8 waves per CU): phase 1 = two stages of {h planes 2 x 5 KB, W planes 2 x 20 KB};
// phases 2-3 = the fp32 tile [64][276] (69 KB, aliases the phase-1 stages) + two stages of W^T planes (2 x 20 KB each).
#include <type_traits>

#include "ga_train_internal.h"

typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));

#define BT_ROWS 64
#define BT_LDP 80                         // bytes per row of a staged plane: 32 k x 2 B + 16 B pad
#define BT_GLD 276                        // floats per row of the fp32 tile: 256 + 16 extension + 4 pad
#define BT_KX 288                         // K of the third product, padded to whole 32-wide steps
#define BT_GT_BYTES (BT_ROWS * BT_GLD * 4)

struct GbTileArgs {
    const float *h, *A, *stats, *ck, *coef, *Ww, *d_afeat, *bcat;
    const _Float16* w16;                  // [2][256][Di]   f16 hi / lo planes of [Wv;Wu]
    const __bf16* wT16;                   // [2][Di][288]   bf16 hi / lo planes of [[Wv;Wu]^T | d_afeat^T | 0]
    float *dS, *dpre, *part;
    int N, K;
};

// 512 threads = 8 waves (two per SIMD): with one workgroup per CU (152 KB of LDS) the second wave of a SIMD and loads issued
// two K steps ahead are what hides the global-memory latency -- a first version with 4 waves and a one-step prefetch spent
// ~110 us per tile waiting.
#define BT_THREADS 512
typedef std::integral_constant<int, 0> BtS0;
typedef std::integral_constant<int, 1> BtS1;

template <int KP, int DI>
__global__ __launch_bounds__(BT_THREADS) void ga_bwd_tile_kernel(GbTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FPL = DI / 64;
    constexpr int PREC = KP * GA_DA + KP + 2 * GA_DA;
    constexpr int ST1 = 2 * BT_ROWS * BT_LDP + 2 * 256 * BT_LDP;     // one phase-1 stage: h hi/lo + W hi/lo
    constexpr int ST3 = 2 * DI * BT_LDP;                             // one phase-3 stage: W^T hi/lo
    constexpr int NCT = DI / 32;                                     // 32-column output tiles of the third product
    constexpr int MT3 = NCT == 8 ? 2 : 1;                            // row tiles per wave there (8 waves: 8 x {0,1} or 4 x 2)
    static_assert(NCT == 8 || NCT == 4, "Di = 256 or 128");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i31 = lane & 31, hi = lane >> 5;
    const int N = a.N, K = a.K;
    const int n0 = blockIdx.x * BT_ROWS;
    constexpr int MASK_OFF = (BT_GT_BYTES + 2 * ST3 > BT_GT_BYTES + ST3 + 8 * PREC * 4) ? BT_GT_BYTES + 2 * ST3 : BT_GT_BYTES + ST3 + 8 * PREC * 4;
    static_assert(MASK_OFF >= 2 * ST1, "the mask outlives the first product's stages");
    unsigned char* const mask_lds = (unsigned char*)(smem + MASK_OFF);                  // [64 rows][64 bytes]: 4 relu bits per byte
#ifdef BT_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif

    // ================================================================ 1: G = h W^T + b  (wave w: columns 32 w .. 32 w + 31, all 64 rows)
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    {
        const int hrow = tid >> 3, hq = tid & 7;                      // h: row, 4-float piece of the 32-wide K step
        const bool hok = n0 + hrow < N;
        const float* hp = a.h + (size_t)(hok ? n0 + hrow : 0) * DI + 4 * hq;
        const int wrow = tid >> 1, wh = tid & 1;                      // W planes: row, 32-byte half of the 64-byte step
        const u32x4* wp_hi = (const u32x4*)(a.w16 + (size_t)wrow * DI) + 2 * wh;
        const u32x4* wp_lo = (const u32x4*)(a.w16 + (size_t)256 * DI + (size_t)wrow * DI) + 2 * wh;
        f32x4 rh[2];
        u32x4 rw[2][4];
        auto load = [&](auto SET, int s) {
            constexpr int S = decltype(SET)::value;
            rh[S] = hok ? *(const f32x4*)(hp + 32 * s) : f32x4{0.f, 0.f, 0.f, 0.f};
            rw[S][0] = wp_hi[4 * s]; rw[S][1] = wp_hi[4 * s + 1];
            rw[S][2] = wp_lo[4 * s]; rw[S][3] = wp_lo[4 * s + 1];
        };
        auto store = [&](auto SET, int stage, int step) {
            constexpr int S = decltype(SET)::value;
            char* St = smem + stage * ST1;
            // relu mask of the tile for the third product's epilogue: 4 bits per thread and step, one byte each (kept past both products)
            mask_lds[hrow * 64 + 8 * step + hq] = (unsigned char)((rh[S][0] > 0.0f ? 1 : 0) | (rh[S][1] > 0.0f ? 2 : 0) | (rh[S][2] > 0.0f ? 4 : 0) | (rh[S][3] > 0.0f ? 8 : 0));
            unsigned h0, l0, h1, l1;
            ga_split_pair_f16(rh[S][0], rh[S][1], h0, l0);
            ga_split_pair_f16(rh[S][2], rh[S][3], h1, l1);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *(u32x2*)(St + hrow * BT_LDP + hq * 8) = u32x2{h0, h1};
            *(u32x2*)(St + BT_ROWS * BT_LDP + hrow * BT_LDP + hq * 8) = u32x2{l0, l1};
            char* W = St + 2 * BT_ROWS * BT_LDP;
            *(u32x4*)(W + wrow * BT_LDP + wh * 32) = rw[S][0];
            *(u32x4*)(W + wrow * BT_LDP + wh * 32 + 16) = rw[S][1];
            *(u32x4*)(W + 256 * BT_LDP + wrow * BT_LDP + wh * 32) = rw[S][2];
            *(u32x4*)(W + 256 * BT_LDP + wrow * BT_LDP + wh * 32 + 16) = rw[S][3];
        };
        auto compute = [&](int stage) {
            const char* St = smem + stage * ST1;
            const char* W = St + 2 * BT_ROWS * BT_LDP;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ao = (32 * t + i31) * BT_LDP + ks * 32 + hi * 16;
                    ah[t] = *(const f16x8*)(St + ao);
                    al[t] = *(const f16x8*)(St + BT_ROWS * BT_LDP + ao);
                }
                const int bo = (32 * wave + i31) * BT_LDP + ks * 32 + hi * 16;
                const f16x8 bh = *(const f16x8*)(W + bo), bl = *(const f16x8*)(W + 256 * BT_LDP + bo);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt], 0, 0, 0);
                }
            }
        };
        constexpr int S1 = DI / 32;
        static_assert(S1 % 2 == 0 && S1 >= 4, "the loop below is unrolled by two");
        load(BtS0{}, 0);
        load(BtS1{}, 1);
        store(BtS0{}, 0, 0);
        load(BtS0{}, 2);
        for (int s = 0; s < S1; s += 2) {
            ga_lds_barrier();                     // stage 0 holds step s; stage 1 (step s - 1) consumed by every wave
            store(BtS1{}, 1, s + 1);             // step s + 1
            if (s + 3 < S1) load(BtS1{}, s + 3);
            compute(0);
            ga_lds_barrier();
            if (s + 2 < S1) { store(BtS0{}, 0, s + 2); if (s + 4 < S1) load(BtS0{}, s + 4); }
            compute(1);
        }
    }
    ga_lds_barrier();                                                 // every wave done with the stages: the fp32 tile takes their place
#ifdef BT_PROF
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
    unsigned long long pt1c = 0;
#endif
    float* Gt = (float*)smem;
    {
        const int col = 32 * wave + i31;
        const float b = a.bcat[col];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Gt[(32 * mt + mfma32_row(r, hi)) * BT_GLD + col] = acc[mt][r] + b;
    }
    // third product's operand stream: W^T planes, 4 x 16 B per thread and step, two steps ahead
    char* const st3 = smem + BT_GT_BYTES;
    const int trow = tid >> 1, th = tid & 1;
    const bool tok = trow < DI;
    const u32x4* tp_hi = (const u32x4*)(a.wT16 + (size_t)(tok ? trow : 0) * BT_KX) + 2 * th;
    const u32x4* tp_lo = (const u32x4*)(a.wT16 + (size_t)DI * BT_KX + (size_t)(tok ? trow : 0) * BT_KX) + 2 * th;
    u32x4 rt[2][4];
    auto load3 = [&](auto SET, int t) {
        constexpr int S = decltype(SET)::value;
        if (tok) { rt[S][0] = tp_hi[4 * t]; rt[S][1] = tp_hi[4 * t + 1]; rt[S][2] = tp_lo[4 * t]; rt[S][3] = tp_lo[4 * t + 1]; }
    };
    auto store3 = [&](auto SET, int stage) {
        constexpr int S = decltype(SET)::value;
        char* St = st3 + stage * ST3;
        if (tok) {
            *(u32x4*)(St + trow * BT_LDP + th * 32) = rt[S][0];
            *(u32x4*)(St + trow * BT_LDP + th * 32 + 16) = rt[S][1];
            *(u32x4*)(St + DI * BT_LDP + trow * BT_LDP + th * 32) = rt[S][2];
            *(u32x4*)(St + DI * BT_LDP + trow * BT_LDP + th * 32 + 16) = rt[S][3];
        }
    };
    load3(BtS0{}, 0);
    load3(BtS1{}, 1);

    // ================================================================ 2: gate pass (wave w: rows 8 w .. 8 w + 7)
    {
        // every global value of the wave's 8 rows first: h rows (FPL floats per lane and row) and the K scores of each row
        // (lane 8 k + rr holds A[k][row rr]), then the arithmetic runs from registers / shuffles
        // (unconditional loads from clamped addresses: as `cond ? load : 0` every one of them became its own exec-masked branch
        // with a scalar-width load; rows past the bag and branches >= K end up multiplied by P = 0 / dA = 0 anyway)
        typedef float bt_fv __attribute__((ext_vector_type(FPL)));
        float hv[8][FPL];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int n = n0 + 8 * wave + rr;
            const bt_fv v = *(const bt_fv*)(a.h + (size_t)(n < N ? n : N - 1) * DI + FPL * lane);
#pragma unroll
            for (int f = 0; f < FPL; ++f) hv[rr][f] = v[f];
        }
        float sA = -INFINITY;
        {
            const int k = lane >> 3, n = n0 + 8 * wave + (lane & 7);
            if (k < K && n < N) sA = a.A[(size_t)k * N + n];
        }
        // small per-step tables: ONE vector load each, broadcast by v_readlane (as ~40 scalar loads in unrolled conditional code
        // they were waited for one by one: ~16 k cycles per tile)
        static_assert(KP * KP <= 64, "coefficient table fits a wave");
        const float cfv = (a.coef && lane < KP * KP) ? a.coef[lane] : 0.0f;       // rows / columns >= K are zero in the table
        const float ckv = lane < K ? a.ck[lane] : 0.0f;
        const float stv = lane < 2 * K ? a.stats[lane] : 1.0f;
        float daf[KP][FPL], ww[KP][2], ck[KP], Mk[KP], iL[KP], cf[KP][KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int kc = k < K ? k : 0;
            const bt_fv dv = *(const bt_fv*)(a.d_afeat + (size_t)kc * DI + FPL * lane);
#pragma unroll
            for (int f = 0; f < FPL; ++f) daf[k][f] = dv[f];
            typedef float bt_f2 __attribute__((ext_vector_type(2)));
            const bt_f2 wv = *(const bt_f2*)(a.Ww + kc * GA_DA + 2 * lane);
            ww[k][0] = wv[0]; ww[k][1] = wv[1];
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
#pragma unroll
            for (int j = 0; j < KP; ++j) cf[k][j] = ga_readlane(cfv, k * KP + j);
            ck[k] = ga_readlane(ckv, k);
            Mk[k] = ga_readlane(stv, (2 * k) & 63);
            iL[k] = __builtin_amdgcn_rcpf(ga_readlane(stv, (2 * k + 1) & 63));
        }
        ga_lds_barrier();                                             // G tile complete (all waves' columns)
#ifdef BT_PROF
        pt1c = __builtin_amdgcn_s_memtime();
#endif
        float aWw[KP][2], abw[KP], abv[2] = {0.f, 0.f}, abu[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KP; ++k) { aWw[k][0] = aWw[k][1] = 0.0f; abw[k] = 0.0f; }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = 8 * wave + rr, n = n0 + row;
            float* grow = Gt + row * BT_GLD;
            const bool live = n < N;                                  // rows past the bag: zero operand rows for the third product
            float dA[KP], P[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float dp = 0.0f;
#pragma unroll
                for (int f = 0; f < FPL; ++f) dp = fmaf(daf[k][f], hv[rr][f], dp);
                dp = ga_wave_sum(dp);
                const float s = ga_readlane(sA, (8 * k + rr) & 63);
                const bool masked = !(s > -5e8f);                     // masked_fill(-1e9) positions, padded branches, rows past the bag
                P[k] = masked ? 0.0f : __expf(s - Mk[k]) * iL[k];
                dA[k] = masked ? 0.0f : P[k] * (dp - ck[k]);
            }
            if (a.coef) {      // d diff_loss / dA[i][n] = p_i[n] * sum_j coef[i][j] p_j[n]
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    float sdiv = 0.0f;
#pragma unroll
                    for (int j = 0; j < KP; ++j) sdiv = fmaf(cf[k][j], P[j], sdiv);
                    dA[k] = fmaf(P[k], sdiv, dA[k]);
                }
            }
            const float gv0 = grow[2 * lane], gv1 = grow[2 * lane + 1], gu0 = grow[GA_DA + 2 * lane], gu1 = grow[GA_DA + 2 * lane + 1];
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
            abv[0] += dGv0; abv[1] += dGv1; abu[0] += dGu0; abu[1] += dGu1;       // (all zero for a row past the bag: dA = 0)
            grow[2 * lane] = dGv0; grow[2 * lane + 1] = dGv1; grow[GA_DA + 2 * lane] = dGu0; grow[GA_DA + 2 * lane + 1] = dGu1;
            if (lane < 16) {                                          // extension K slots: P[k] (the pooling term rides in the product)
                float pv = 0.0f;
#pragma unroll
                for (int k = 0; k < KP; ++k) pv = (lane == k) ? P[k] : pv;
                grow[2 * GA_DA + lane] = pv;
            }
            if (live) {
                float* gout = a.dS + (size_t)n * (2 * GA_DA);
                gout[2 * lane] = dGv0; gout[2 * lane + 1] = dGv1;
                gout[GA_DA + 2 * lane] = dGu0; gout[GA_DA + 2 * lane + 1] = dGu1;
            }
        }
        // workgroup partial record: [k][128] dWw, [k] dbw, [128] dbv, [128] dbu  (scratch: the second W^T stage, not in use yet)
        float* sred = (float*)(st3 + ST3);
        // (8 records may reach past that stage when Di = 128: the launcher sizes the LDS for it)
        float* rec = sred + wave * PREC;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            rec[k * GA_DA + 2 * lane] = aWw[k][0]; rec[k * GA_DA + 2 * lane + 1] = aWw[k][1];
            if (lane == 0) rec[KP * GA_DA + k] = abw[k];
        }
        rec[KP * GA_DA + KP + 2 * lane] = abv[0]; rec[KP * GA_DA + KP + 2 * lane + 1] = abv[1];
        rec[KP * GA_DA + KP + GA_DA + 2 * lane] = abu[0]; rec[KP * GA_DA + KP + GA_DA + 2 * lane + 1] = abu[1];
        ga_lds_barrier();
        float* out = a.part + (size_t)blockIdx.x * PREC;
        for (int e = tid; e < PREC; e += BT_THREADS) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += sred[w * PREC + e];
            out[e] = t;
        }
    }
#ifdef BT_PROF
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    store3(BtS0{}, 0);
    load3(BtS0{}, 2);

    // ================================================================ 3: dpre = [dS | P] [W^T | d_afeat^T]^T, masked by h > 0
    // wave w: column tile ct, row tiles mt0 .. mt0 + MT3 - 1
    const int ct = NCT == 8 ? wave : (wave & 3);
    const int mt0 = NCT == 8 ? 0 : (wave >> 2);
    f32x16 acc3[MT3];
#pragma unroll
    for (int m = 0; m < MT3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[m][r] = 0.0f;
    constexpr int S3 = BT_KX / 32;                                    // 9 steps; the last one holds the 16 extension slots (+ 16 zeros)
    auto compute3 = [&](int stage, int t) {
        const char* St = st3 + stage * ST3;
        const int nks = (t == S3 - 1) ? 1 : 2;                        // K = 272: the last step has one 16-wide sub-step
        for (int ks = 0; ks < nks; ++ks) {
            const int bo = (32 * ct + i31) * BT_LDP + ks * 32 + hi * 16;
            const bt_bf16x8 bh = *(const bt_bf16x8*)(St + bo), bl = *(const bt_bf16x8*)(St + DI * BT_LDP + bo);
#pragma unroll
            for (int m = 0; m < MT3; ++m) {
                const float* gp = Gt + (32 * (mt0 + m) + i31) * BT_GLD + 32 * t + 16 * ks + 8 * hi;
                const f32x4 v0 = *(const f32x4*)gp, v1 = *(const f32x4*)(gp + 4);
                u32x4 hw, lw;
                unsigned x, y;
                ga_split_pair_bf16(v0[0], v0[1], x, y); hw[0] = x; lw[0] = y;
                ga_split_pair_bf16(v0[2], v0[3], x, y); hw[1] = x; lw[1] = y;
                ga_split_pair_bf16(v1[0], v1[1], x, y); hw[2] = x; lw[2] = y;
                ga_split_pair_bf16(v1[2], v1[3], x, y); hw[3] = x; lw[3] = y;
                const bt_bf16x8 ah = __builtin_bit_cast(bt_bf16x8, hw), al = __builtin_bit_cast(bt_bf16x8, lw);
                acc3[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc3[m], 0, 0, 0);
                acc3[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc3[m], 0, 0, 0);
                acc3[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc3[m], 0, 0, 0);
            }
        }
    };
    for (int t = 0; t < S3; t += 2) {
        ga_lds_barrier();                     // stage 0 holds step t (and, t = 0: dS / P rows written, partial records read)
        if (t + 1 < S3) { store3(BtS1{}, 1); if (t + 3 < S3) load3(BtS1{}, t + 3); }
        compute3(0, t);
        if (t + 1 >= S3) break;
        ga_lds_barrier();
        if (t + 2 < S3) { store3(BtS0{}, 0); if (t + 4 < S3) load3(BtS0{}, t + 4); }
        compute3(1, t + 1);
    }
#ifdef BT_PROF
    const unsigned long long pt3 = __builtin_amdgcn_s_memtime();
#endif
    // relu mask + store: lane = column, registers = rows (128-byte row segments per half wave)
    {
        const int col = 32 * ct + i31;
#pragma unroll
        for (int m = 0; m < MT3; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (mt0 + m) + mfma32_row(r, hi), n = n0 + row;
                const bool pos = (mask_lds[row * 64 + (col >> 2)] >> (col & 3)) & 1;
                if (n < N) a.dpre[(size_t)n * DI + col] = pos ? acc3[m][r] : 0.0f;
            }
        }
    }
#ifdef BT_PROF
    if (blockIdx.x == 3 && tid == 0)
        printf("BT phases (cycles): gemm1 %llu  prefetch+Gt %llu  gate %llu  gemm3 %llu  epilogue %llu  total %llu\n", pt1 - pt0, pt1c - pt1, pt2 - pt1c, pt3 - pt2,
               __builtin_amdgcn_s_memtime() - pt3, __builtin_amdgcn_s_memtime() - pt0);
#endif
}

size_t ga_bwd_tile_part_records(int N) { return (size_t)(N + BT_ROWS - 1) / BT_ROWS; }

// launches 6-8 of the training step as one kernel; ACMIL_ERR_UNSUPPORTED when no instance fits (the caller keeps the three launches)
int ga_bwd_tile_launch(const float* h, const float* A, const float* stats, const float* ck, const float* coef, const float* Ww,
                       const float* d_afeat, const float* bcat, const void* w16, const void* wT16, float* dS, float* dpre,
                       float* part, int N, int K, int Di, hipStream_t st) {
    const int KP = (K <= 1) ? 1 : (K <= 5) ? 5 : 8;
    if (KP == 8 || (Di != 128 && Di != 256)) return ACMIL_ERR_UNSUPPORTED;
    GbTileArgs a;
    a.h = h; a.A = A; a.stats = stats; a.ck = ck; a.coef = coef; a.Ww = Ww; a.d_afeat = d_afeat; a.bcat = bcat;
    a.w16 = (const _Float16*)w16; a.wT16 = (const __bf16*)wT16; a.dS = dS; a.dpre = dpre; a.part = part; a.N = N; a.K = K;
    void (*kern)(GbTileArgs) = nullptr;
    size_t lds1 = 0, lds3 = 0;
#define BT_PICK(KP_, DI_) { kern = ga_bwd_tile_kernel<KP_, DI_>; lds1 = 2 * (2 * BT_ROWS * BT_LDP + 2 * 256 * BT_LDP); lds3 = BT_GT_BYTES + 2 * (2 * DI_ * BT_LDP); \
                           const size_t rec = BT_GT_BYTES + (2 * DI_ * BT_LDP) + (size_t)8 * (KP_ * GA_DA + KP_ + 2 * GA_DA) * 4; if (rec > lds3) lds3 = rec; \
                           lds3 += 4096 /* relu mask */; }
    if (KP == 1 && Di == 128) BT_PICK(1, 128) else if (KP == 1) BT_PICK(1, 256) else if (Di == 128) BT_PICK(5, 128) else BT_PICK(5, 256)
#undef BT_PICK
    const size_t lds = (lds1 > lds3 ? lds1 : lds3);
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)ga_bwd_tile_part_records(N)), dim3(BT_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
