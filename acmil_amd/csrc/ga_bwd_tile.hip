// ga_bwd_tile.hip -- the gate side of the GA backward as ONE kernel per 64- or 32-patch tile (replaces three launches of a
// training step: the G recompute GEMM, the gate pass and the dpre GEMM, and the HBM round trips of G [N,256] and dh0 [N,Di]
// between them).  Autograd of architecture/transformer.py:259-267 and :322-324 w.r.t. h (no explicit backward code in the
// reference):
//
//   1  G   = h [Wv;Wu]^T + [bv;bu]                      ROWS x 256, K = Di    split-f16 MFMA (forward-sized values)
//   2  gate pass, one wave per patch, G resident in LDS  (same arithmetic as ga_bwd_gate_kernel, ga_backward.hip):
//        P = softmax prob., dA = P (d_afeat . h - c) + diversity term (0 where masked), V = tanh, U = sigmoid,
//        dg = dA^T Ww, dS_v = dg U (1 - V^2), dS_u = dg V U (1 - U)  -> dS goes to HBM for the weight-gradient kernel and
//        replaces its own G row in LDS as bf16 hi / lo halves; partial sums of dWw, dbw, dbv, dbu per workgroup
//   3  dpre = ([dS | P] [[Wv;Wu]^T | d_afeat^T]^T) * [h > 0]      ROWS x Di, K = 256 + 16    split-bf16 MFMA
//      (the pooling term dh0 = P d_afeat rides along as 16 extra K slots, so dh0 never exists)
//
// Operands that do not depend on the bag are pre-split once per step: [Wv;Wu] [256][Di] as f16 hi / lo halves and
// [[Wv;Wu]^T | d_afeat^T | 0] [Di][288] as bf16 hi / lo halves, both stored in MFMA-fragment order (ga_common.h::ga_frag_off;
// ga_pack.hip writes them, the d_afeat columns are filled by the tail kernel, ga_step.hip).
//
// Round 3 layout.  In BOTH products every wave owns its own 32 output columns, i.e. its own rows of the weight operand: staging
// the weights through LDS bought no reuse, only 2 x 40 KB of LDS and a barrier per K step (round 2: 152 KB = one workgroup per CU,
// 66 k cycles per tile of which 13 k are MFMA issue and 17 k VALU issue).  Now the B fragments go global (L2) -> registers, four
// 16-wide K steps ahead, the K loops have no barrier at all, the h tile is staged ONCE as f16 hi / lo planes, and [dS | P] is
// split ONCE by the gate pass (round 2: every wave of the third product split the same fp32 values again: 8 x the VALU work).
// LDS: max(h planes, fp32 G tile = its bf16 planes in place) + relu bits + two partial records = 78 KB at 64 rows, so TWO
// workgroups share a CU (8 waves each, <= 128 VGPRs) and one's gate pass (VALU) overlaps the other's products (MFMA).  Bags that
// do not give every CU a 64-row tile run 32-row tiles.
#include <type_traits>

#include "ga_train_internal.h"

typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));

#define BT_GLD 276                        // floats per row of the fp32 G tile: 256 + 16 extension + 4 pad
#define BT_ROWB (BT_GLD * 4)              // the same row in bytes; after the gate pass: [272 bf16 hi][272 bf16 lo][16 B]
#define BT_LOFF 544                       // byte offset of the lo halves inside a row
#define BT_KX 288                         // K of the third product as stored (row length of wT16), 272 used
#ifndef BT_PF
#define BT_PF 4                           // B fragments in flight: 16-wide K steps ahead
#endif

struct GbTileArgs {
    const float *h, *A, *stats, *ck, *coef, *Ww, *d_afeat, *bcat;
    const _Float16* w16;                  // f16 hi / lo of [Wv;Wu] [256][Di], fragment order (ga_common.h::ga_frag_off)
    const __bf16* wT16;                   // bf16 hi / lo of [[Wv;Wu]^T | d_afeat^T | 0] [Di][288], fragment order
    float *dS, *dpre, *part;
    int N, K;                             // N: rows of ALL bags of the launch (= the row stride of A)
    // a GROUP of bags (multi-bag training step; one bag: seg = ga_seg_single(N), strides unused): tiles are cut per bag, and what
    // depends on the bag's softmax / loss comes per bag: stats [bag][16], ck [bag][16], coef [bag][64], d_afeat [bag][K][Di],
    // and the d_afeat K slots of the third product's B operand: wT_ext [bag][Di / 32][2 planes][64 lanes][8] bf16 (ga_frag_off
    // with one K step; null: the slots inside wT16 itself)
    GaSeg seg;
    const __bf16* wT_ext;
};

#define BT_THREADS 512

template <int KP, int DI, int ROWS>
__global__ __launch_bounds__(BT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ga_bwd_tile_kernel(GbTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FPL = DI / 64;
    constexpr int PREC = KP * GA_DA + KP + 2 * GA_DA;
    constexpr int MT = ROWS / 32;                                    // 32-row tiles
    constexpr int RPW = ROWS / 8;                                    // gate-pass rows per wave
    constexpr int HLD = DI * 2 + 16;                                 // bytes per row of an h plane
    constexpr int NKS1 = DI / 16, NKS3 = 17;                         // 16-wide K steps of the two products (272 = 17 x 16)
    constexpr int NCT = DI / 32;                                     // 32-column output tiles of the third product
    static_assert(NCT == 8 || NCT == 4, "Di = 256 or 128");
    static_assert(ROWS == 64 || ROWS == 32, "tile height");
    constexpr int GT_BYTES = ROWS * BT_ROWB, H_BYTES = 2 * ROWS * HLD;
    constexpr int MASK_OFF = GT_BYTES > H_BYTES ? GT_BYTES : H_BYTES;
    constexpr int MASK_LD = DI / 8;                                  // relu bits of the tile: one bit per element
    constexpr int REC_OFF = MASK_OFF + ROWS * MASK_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i31 = lane & 31, hi = lane >> 5;
    const int K = a.K, ldA = a.N;
    const GaSegTile sg = ga_seg_find<ROWS>(a.seg, blockIdx.x);
    const int n0 = sg.n0, N = sg.nend, bag = sg.bag;          // N: end row of this tile's bag
    const float* const stats_b = a.stats + 16 * bag;
    const float* const ck_b = a.ck + 16 * bag;
    const float* const coef_b = a.coef ? a.coef + 64 * bag : nullptr;
    const float* const daf_b = a.d_afeat + (size_t)bag * K * DI;
    unsigned char* const mask_lds = (unsigned char*)(smem + MASK_OFF);
#ifdef BT_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif

    // ================================================================ 1: G = h W^T + b  (wave w: columns 32 w .. 32 w + 31, all rows)
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    {
        // the wave's B fragments: rows 32 w + i31, 16 bytes per lane, plane and 16-wide K step, 1 KB contiguous per wave-load
        const _Float16* wb = a.w16 + (size_t)wave * NKS1 * 1024 + lane * 8;
        u32x4 bq[BT_PF][2];
#pragma unroll
        for (int j = 0; j < BT_PF; ++j) {
            bq[j][0] = *(const u32x4*)(wb + 1024 * j);
            bq[j][1] = *(const u32x4*)(wb + 1024 * j + 512);
        }
        // the h tile, once: f16 hi / lo planes + the relu bits for the third product's epilogue
        {
            constexpr int QPR = DI / 4, RPP = BT_THREADS / QPR, NP = ROWS / RPP;
            const int hq = tid % QPR, hr = tid / QPR;
            f32x4 rh[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int n = n0 + hr + p * RPP;
                rh[p] = *(const f32x4*)(a.h + (size_t)(n < N ? n : N - 1) * DI + 4 * hq);      // clamped: no exec-masked loads
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int row = hr + p * RPP;
                f32x4 v = rh[p];
                if (n0 + row >= N) v = f32x4{0.f, 0.f, 0.f, 0.f};
                const int m4 = (v[0] > 0.0f ? 1 : 0) | (v[1] > 0.0f ? 2 : 0) | (v[2] > 0.0f ? 4 : 0) | (v[3] > 0.0f ? 8 : 0);
                const int nb = __builtin_amdgcn_mov_dpp(m4, 0xB1, 0xf, 0xf, true);                // the neighbour lane's bits (quad_perm 1,0,3,2)
                if (!(hq & 1)) mask_lds[row * MASK_LD + (hq >> 1)] = (unsigned char)(m4 | (nb << 4));
                unsigned h0, l0, h1, l1;
                ga_split_pair_f16(v[0], v[1], h0, l0);
                ga_split_pair_f16(v[2], v[3], h1, l1);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *(u32x2*)(smem + row * HLD + hq * 8) = u32x2{h0, h1};
                *(u32x2*)(smem + ROWS * HLD + row * HLD + hq * 8) = u32x2{l0, l1};
            }
        }
        ga_lds_barrier();
        const char* HP = smem;
        const char* LP = smem + ROWS * HLD;
#pragma unroll
        for (int ks = 0; ks < NKS1; ++ks) {
            f16x8 ah[MT], al[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int ao = (32 * t + i31) * HLD + ks * 32 + hi * 16;
                ah[t] = *(const f16x8*)(HP + ao);
                al[t] = *(const f16x8*)(LP + ao);
            }
            const f16x8 bh = __builtin_bit_cast(f16x8, bq[ks % BT_PF][0]), bl = __builtin_bit_cast(f16x8, bq[ks % BT_PF][1]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt], 0, 0, 0);
            }
            if (ks + BT_PF < NKS1) {
                bq[ks % BT_PF][0] = *(const u32x4*)(wb + 1024 * (ks + BT_PF));
                bq[ks % BT_PF][1] = *(const u32x4*)(wb + 1024 * (ks + BT_PF) + 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    ga_lds_barrier();                                                 // every wave done with the h planes: the fp32 tile takes their place
#ifdef BT_PROF
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
    unsigned long long pt1c = 0;
#endif
    float* Gt = (float*)smem;
    {
        const int col = 32 * wave + i31;
        const float b = a.bcat[col];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Gt[(32 * mt + mfma32_row(r, hi)) * BT_GLD + col] = acc[mt][r] + b;
    }
#ifdef BT_PROF
    const unsigned long long ptA = __builtin_amdgcn_s_memtime();
    unsigned long long ptB = 0, ptR = 0;
#endif

    // ================================================================ 2: gate pass (wave w: rows RPW w .. RPW w + RPW - 1)
    {
        // every global value of the wave's rows first: h rows (FPL floats per lane and row) and the K scores of each row
        // (lane RPW k + rr holds A[k][row rr]), then the arithmetic runs from registers / shuffles
        // (unconditional loads from clamped addresses: as `cond ? load : 0` every one of them became its own exec-masked branch
        // with a scalar-width load; rows past the bag and branches >= K end up multiplied by P = 0 / dA = 0 anyway)
        typedef float bt_fv __attribute__((ext_vector_type(FPL)));
        // (the row loop below is NOT unrolled: eight independent rows interleaved by the scheduler spilled at the 128 registers two
        // workgroups per CU leave a wave; the next row's h values are fetched while the current row is worked on)
        auto hrow_load = [&](int rr) {
            const int n = n0 + RPW * wave + (rr < RPW ? rr : RPW - 1);
            return *(const bt_fv*)(a.h + (size_t)(n < N ? n : N - 1) * DI + FPL * lane);
        };
        bt_fv hcur = hrow_load(0);
        // everything that depends only on (branch k, row rr) is formed ONCE per wave, one (k, rr) pair per lane (lane RPW k + rr):
        // P = softmax probability (0 where masked / padded / past the bag) and Q = P sum_j coef[k][j] P_j (the diversity term of dA);
        // the row loop reads them back with v_readlane (as uniform per-row arithmetic they were ~16 VALU instructions per branch and
        // row on all 64 lanes -- the gate pass is VALU-issue bound)
        static_assert(KP * RPW <= 64, "the (branch, row) pairs of a wave fit one register");
        const int lk = lane / RPW, lr = lane % RPW;
        float P_all = 0.0f, Q_all = 0.0f;
        {
            const int n = n0 + RPW * wave + lr;
            const bool ok = lk < K && n < N;
            const int kc = lk < K ? lk : 0;
            const float s = a.A[(size_t)kc * ldA + (n < N ? n : N - 1)];
            const float M = stats_b[2 * kc], Lsum = stats_b[2 * kc + 1];
            const bool masked = !ok || !(s > -5e8f);                  // masked_fill(-1e9) positions, padded branches, rows past the bag
            P_all = masked ? 0.0f : __expf(s - M) * __builtin_amdgcn_rcpf(Lsum);
            if (coef_b) {      // d diff_loss / dA[i][n] = p_i[n] * sum_j coef[i][j] p_j[n]   (rows / columns >= K are zero in the table)
                float sdiv = 0.0f;
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const float Pj = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (RPW * j + lr), __builtin_bit_cast(int, P_all)));
                    sdiv = fmaf(lk < KP ? coef_b[lk * KP + j] : 0.0f, Pj, sdiv);
                }
                Q_all = P_all * sdiv;
            }
        }
        const float ckv = lane < K ? ck_b[lane] : 0.0f;
        float daf[KP][FPL], ww[KP][2], ck[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int kc = k < K ? k : 0;
            const bt_fv dv = *(const bt_fv*)(daf_b + (size_t)kc * DI + FPL * lane);
#pragma unroll
            for (int f = 0; f < FPL; ++f) daf[k][f] = dv[f];
            typedef float bt_f2 __attribute__((ext_vector_type(2)));
            const bt_f2 wv = *(const bt_f2*)(a.Ww + kc * GA_DA + 2 * lane);
            ww[k][0] = wv[0]; ww[k][1] = wv[1];
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) ck[k] = ga_readlane(ckv, k);
#ifdef BT_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ptB = __builtin_amdgcn_s_memtime();
#endif
        ga_lds_barrier();                                             // G tile complete (all waves' columns)
#ifdef BT_PROF
        pt1c = __builtin_amdgcn_s_memtime();
#endif
        float aWw[KP][2], abw[KP], abv[2] = {0.f, 0.f}, abu[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KP; ++k) { aWw[k][0] = aWw[k][1] = 0.0f; abw[k] = 0.0f; }
#pragma unroll 1
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = RPW * wave + rr, n = n0 + row;
            char* grow = smem + row * BT_ROWB;
            const bt_fv hnext = hrow_load(rr + 1);
            const bool live = n < N;                                  // rows past the bag: zero operand rows for the third product
            float dA[KP], P[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float dp = 0.0f;
#pragma unroll
                for (int f = 0; f < FPL; ++f) dp = fmaf(daf[k][f], hcur[f], dp);
                dp = ga_wave_sum(dp);
                P[k] = ga_readlane(P_all, (RPW * k + rr) & 63);
                dA[k] = fmaf(P[k], dp - ck[k], ga_readlane(Q_all, (RPW * k + rr) & 63));     // P = 0 (masked) makes Q = 0 too
            }
            typedef float bt_f2 __attribute__((ext_vector_type(2)));
            const bt_f2 gv = *(const bt_f2*)(grow + 8 * lane), gu = *(const bt_f2*)(grow + 4 * GA_DA + 8 * lane);
            const float V0 = ga_tanh(gv[0]), V1 = ga_tanh(gv[1]), U0 = ga_sigmoid(gu[0]), U1 = ga_sigmoid(gu[1]);
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
            // the row's operand of the third product, split once: bf16 hi / lo halves IN PLACE of the fp32 row (the wave has read
            // all of it above; LDS operations of a wave complete in order)
            unsigned vh, vl, uh, ul;
            ga_split_pair_bf16(dGv0, dGv1, vh, vl);
            ga_split_pair_bf16(dGu0, dGu1, uh, ul);
            float p0 = 0.0f, p1 = 0.0f;                               // extension K slots 256 .. 271: P[k] (the pooling term rides along)
#pragma unroll
            for (int k = 0; k < KP; ++k) { p0 = (2 * lane == k) ? P[k] : p0; p1 = (2 * lane + 1 == k) ? P[k] : p1; }
            unsigned ph, pl;
            ga_split_pair_bf16(p0, p1, ph, pl);
            *(unsigned*)(grow + 4 * lane) = vh;                     *(unsigned*)(grow + BT_LOFF + 4 * lane) = vl;
            *(unsigned*)(grow + 2 * GA_DA + 4 * lane) = uh;         *(unsigned*)(grow + BT_LOFF + 2 * GA_DA + 4 * lane) = ul;
            if (lane < 8) { *(unsigned*)(grow + 4 * GA_DA + 4 * lane) = ph; *(unsigned*)(grow + BT_LOFF + 4 * GA_DA + 4 * lane) = pl; }
            if (live) {
                float* gout = a.dS + (size_t)n * (2 * GA_DA);
                *(bt_f2*)(gout + 2 * lane) = bt_f2{dGv0, dGv1};
                *(bt_f2*)(gout + GA_DA + 2 * lane) = bt_f2{dGu0, dGu1};
            }
            hcur = hnext;
        }
#ifdef BT_PROF
        ptR = __builtin_amdgcn_s_memtime();
#endif
        // workgroup partial record: [k][128] dWw, [k] dbw, [128] dbv, [128] dbu -- the 8 waves fold into two slots in a fixed
        // order (waves 0 / 1 write, 2 / 3 add, ...: four rounds), so the record is bitwise reproducible
        float* const slots = (float*)(smem + REC_OFF);
#pragma unroll 1
        for (int round = 0; round < 4; ++round) {
            if ((wave >> 1) == round) {
                float* rec = slots + (wave & 1) * PREC;
                typedef float bt_f2 __attribute__((ext_vector_type(2)));
                auto put2 = [&](float* p, float x, float y) {
                    bt_f2 v = {x, y};
                    if (round) { const bt_f2 o = *(const bt_f2*)p; v[0] += o[0]; v[1] += o[1]; }
                    *(bt_f2*)p = v;
                };
#pragma unroll
                for (int k = 0; k < KP; ++k) put2(rec + k * GA_DA + 2 * lane, aWw[k][0], aWw[k][1]);
                put2(rec + KP * GA_DA + KP + 2 * lane, abv[0], abv[1]);
                put2(rec + KP * GA_DA + KP + GA_DA + 2 * lane, abu[0], abu[1]);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < KP; ++k) rec[KP * GA_DA + k] = (round ? rec[KP * GA_DA + k] : 0.0f) + abw[k];
                }
            }
            ga_lds_barrier();
        }
        float* out = a.part + (size_t)blockIdx.x * PREC;
        for (int e = tid; e < PREC; e += BT_THREADS) out[e] = slots[e] + slots[PREC + e];
    }
#ifdef BT_PROF
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif

    // ================================================================ 3: dpre = [dS | P] [W^T | d_afeat^T]^T, masked by h > 0
    // wave w: column tile ct, row tiles mt0 .. mt0 + MT3 - 1 (Di = 128: four column tiles -- two row tiles go to the two wave
    // groups, one row tile leaves waves 4 .. 7 without work).  No barrier from here on.
    constexpr int MT3 = NCT == 8 ? MT : 1;
    const int ct = NCT == 8 ? wave : (wave & 3);
    const int mt0 = (NCT == 8 || MT == 1) ? 0 : (wave >> 2);
    if (NCT == 8 || MT == 2 || wave < 4) {
        f32x16 acc3[MT3];
#pragma unroll
        for (int m = 0; m < MT3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[m][r] = 0.0f;
        const __bf16* tb = a.wT16 + (size_t)ct * (BT_KX / 16) * 1024 + lane * 8;
        // the last K step (d_afeat slots) of a group launch comes from the bag's own fragment pair
        const __bf16* te = a.wT_ext ? a.wT_ext + ((size_t)bag * NCT + ct) * 1024 + lane * 8 : tb + 1024 * (NKS3 - 1);
        u32x4 bq[BT_PF][2];
#pragma unroll
        for (int j = 0; j < BT_PF; ++j) {
            bq[j][0] = *(const u32x4*)(tb + 1024 * j);
            bq[j][1] = *(const u32x4*)(tb + 1024 * j + 512);
        }
        __builtin_amdgcn_sched_barrier(0);          // keep all BT_PF steps in flight (the scheduler sinks the loads to their uses otherwise)
#pragma unroll
        for (int ks = 0; ks < NKS3; ++ks) {
            bt_bf16x8 ah[MT3], al[MT3];
#pragma unroll
            for (int m = 0; m < MT3; ++m) {
                const char* ap = smem + (32 * (mt0 + m) + i31) * BT_ROWB + ks * 32 + hi * 16;
                ah[m] = *(const bt_bf16x8*)ap;
                al[m] = *(const bt_bf16x8*)(ap + BT_LOFF);
            }
            const bt_bf16x8 bh = __builtin_bit_cast(bt_bf16x8, bq[ks % BT_PF][0]), bl = __builtin_bit_cast(bt_bf16x8, bq[ks % BT_PF][1]);
#pragma unroll
            for (int m = 0; m < MT3; ++m) {
                acc3[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc3[m], 0, 0, 0);
                acc3[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc3[m], 0, 0, 0);
                acc3[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc3[m], 0, 0, 0);
            }
            if (ks + BT_PF < NKS3) {
                const __bf16* src = (ks + BT_PF == NKS3 - 1) ? te : tb + 1024 * (ks + BT_PF);
                bq[ks % BT_PF][0] = *(const u32x4*)src;
                bq[ks % BT_PF][1] = *(const u32x4*)(src + 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef BT_PROF
        const unsigned long long pt3 = __builtin_amdgcn_s_memtime();
#endif
        // relu mask + store: lane = column, registers = rows (128-byte row segments per half wave); the 16 mask bytes first, then the
        // stores back to back (one LDS read + wait + branch per store serialised the 32 stores of a wave)
        const int col = 32 * ct + i31;
        const bool full = n0 + ROWS <= N;
#pragma unroll
        for (int m = 0; m < MT3; ++m) {
            unsigned char mb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mb[r] = mask_lds[(32 * (mt0 + m) + mfma32_row(r, hi)) * MASK_LD + (col >> 3)];
            float* dp = a.dpre + (size_t)(n0 + 32 * (mt0 + m)) * DI + col;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[(size_t)mfma32_row(r, hi) * DI] = ((mb[r] >> (col & 7)) & 1) ? acc3[m][r] : 0.0f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (n0 + 32 * (mt0 + m) + mfma32_row(r, hi) < N) dp[(size_t)mfma32_row(r, hi) * DI] = ((mb[r] >> (col & 7)) & 1) ? acc3[m][r] : 0.0f;
            }
        }
#ifdef BT_PROF
        if ((blockIdx.x == 3 || blockIdx.x == 300) && tid == 0)
            printf("BT phases (cycles) blk %d: stage+gemm1 %llu  Gt write %llu  gate loads+pre %llu  barrier %llu  gate rows %llu  records %llu  gemm3 %llu  epilogue %llu  total %llu\n",
                   (int)blockIdx.x, pt1 - pt0, ptA - pt1, ptB - ptA, pt1c - ptB, ptR - pt1c, pt2 - ptR, pt3 - pt2, __builtin_amdgcn_s_memtime() - pt3, __builtin_amdgcn_s_memtime() - pt0);
#endif
    }
}

// tile height for a bag: 64 rows, or 32 when 64-row tiles would leave CUs without one (two workgroups fit a CU either way)
static int bt_rows(int N) {
    static const int forced = [] { const char* e = ACMIL_AB_ENV("ACMIL_GA_BWD_ROWS"); return e ? atoi(e) : 0; }();
    if (forced == 32 || forced == 64) return forced;
    return (N + 63) / 64 <= 320 ? 32 : 64;
}
size_t ga_bwd_tile_part_records(int N) { return (size_t)(N + 31) / 32 + GA_SEG_MAX; }        // upper bound (workspace sizing; every bag of a group may end on a partial tile)

template <int KP, int DI, int ROWS>
static int bt_launch(const GbTileArgs& a, int tiles, hipStream_t st) {
    constexpr int HLD = DI * 2 + 16, GT = ROWS * BT_ROWB, HB = 2 * ROWS * HLD;
    constexpr size_t lds = (GT > HB ? GT : HB) + ROWS * (DI / 8) + 2 * (KP * GA_DA + KP + 2 * GA_DA) * 4;
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    void (*kern)(GbTileArgs) = ga_bwd_tile_kernel<KP, DI, ROWS>;
    int dev = 0;
    static bool set[64] = {};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ACMIL_ERR_LAUNCH;
    if (!set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ACMIL_ERR_LAUNCH;
        set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(BT_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// launches 6-8 of the training step as one kernel; ACMIL_ERR_UNSUPPORTED when no instance fits (the caller keeps the three launches);
// *records = partial records written (one per tile)
int ga_bwd_tile_launch(const float* h, const float* A, const float* stats, const float* ck, const float* coef, const float* Ww,
                       const float* d_afeat, const float* bcat, const void* w16, const void* wT16, float* dS, float* dpre,
                       float* part, int N, int K, int Di, hipStream_t st, int* records, const GaSeg* seg, const void* wT_ext) {
    const int KP = (K <= 1) ? 1 : (K <= 5) ? 5 : 8;
    if (KP == 8 || (Di != 128 && Di != 256)) return ACMIL_ERR_UNSUPPORTED;
    if (seg && (seg->n < 1 || seg->n > GA_SEG_MAX || seg->row0[seg->n] != N || !wT_ext)) return ACMIL_ERR_SHAPE;
    GbTileArgs a;
    a.h = h; a.A = A; a.stats = stats; a.ck = ck; a.coef = coef; a.Ww = Ww; a.d_afeat = d_afeat; a.bcat = bcat;
    a.w16 = (const _Float16*)w16; a.wT16 = (const __bf16*)wT16; a.dS = dS; a.dpre = dpre; a.part = part; a.N = N; a.K = K;
    a.seg = seg ? *seg : ga_seg_single(N); a.wT_ext = seg ? (const __bf16*)wT_ext : nullptr;
    const int rows = bt_rows(N), tiles = ga_seg_tiles(a.seg, rows);
    *records = tiles;
#define BT_PICK(KP_, DI_) (rows == 32 ? bt_launch<KP_, DI_, 32>(a, tiles, st) : bt_launch<KP_, DI_, 64>(a, tiles, st))
    if (KP == 1) return Di == 128 ? BT_PICK(1, 128) : BT_PICK(1, 256);
    return Di == 128 ? BT_PICK(5, 128) : BT_PICK(5, 256);
#undef BT_PICK
}
