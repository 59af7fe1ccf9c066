// linear_kernel.h -- y = act(x W^T + b) (+ beta y) for the nn.Linear layers of the aggregation paths, split-f16 arithmetic
// (3 f16 MFMA products per fp32 product, fp32 accumulate), gfx950.
//
// Replaces, as ONE kernel family: DimReduction.fc1 / TransMIL._fc1 (architecture/network.py:49-57, transMIL.py:51,63),
// NystromAttention.to_qkv / to_out (nystrom_attention.py:80,139), the [Wv;Wu] projection of Attention_Gated outside the
// fused GA kernel (transformer.py:259-267).
//
// It is the GEMM1 loop of ga_fwd2_kernel (ga_forward_kernel_v2.h) made standalone: the weight matrix is consumed as a
// pre-packed f16 hi/lo FRAGMENT STREAM (acmil_linear_pack: 256- or 128-wide output chunks, per chunk K/16 steps x 2 ND
// fragment rows in exactly the order the MFMA A operand wants them), staged with plain linear LDS-DMA; the activations x are
// the B operand: fp32 (or 16-bit) rows DMA'd as they are, split hi/lo in registers inside the MFMA shadow.  Persistent
// workgroups (2 per CU, 4 waves, one 32-row x (32 ND)-column tile per wave) draw (row tile, chunk) pairs from a counter;
// the DMA ring runs across tile boundaries; P3 (Wlo * xhi) of a step is deferred across the next step's barrier where it
// covers the fragment reads.  The epilogue transposes the accumulators through the free ring slot so that a half-wave
// stores 128 contiguous bytes of an output row, adding bias / ReLU / beta * y there.
#pragma once
#include "ga_forward_kernel_v2.h"

#ifndef LIN_XCD_ORDER
#define LIN_XCD_ORDER 1
#endif

#ifndef LIN_LOSKIP
#define LIN_LOSKIP 1          // 0 (A/B builds): always issue the W_hi x_lo products of an fp32 operand
#endif

#define LIN_CTRL_BYTES 96        // control words of one Linear call (linear.hip::lin_f16x3_run)

// timing-only ablations of lin_kernel (tools/build_lin_variants.sh; WRONG results): 1 no x DMA, 2 no weight DMA, 8 no epilogue stores
// (accumulators kept alive).  0 in every product build.
#ifndef LIN_ABL
#define LIN_ABL 0
#endif

// 16-byte store past the L2 (`nt`).  Inline asm: under a run-time condition hipcc merges `__builtin_nontemporal_store` with the plain
// store of the other branch and drops the hint (checked in the ISA: 24 plain stores).  s_nop 1: the data registers must not be
// overwritten before the store has read them (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void lin_store_nt(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

struct LinArgs {
    const void* x;          // [M, K] row-major, leading dimension ldx (elements), fp32 / fp16 / bf16
    const char* packed;     // fragment stream: chunk c at c * (K/16) * 2 ND KiB
    const float* bias;      // [n_out of this launch] or null (indexed by output column - col0)
    float* y;               // [M, ldy] fp32; this launch writes columns col0 .. col0 + nchunks * 32 ND
    unsigned* tile_counter; // 8 zeroed words (dynamic tile drawing: one queue per XCD, see the kernel) or null
    unsigned* done;         // zeroed word or null: finished-workgroup count; the last workgroup puts it and the 8 queue words back to
                            // zero, so a caller that zeroed them once (TransMIL: once per forward) needs no memset between launches
    long long ldx, ldy;
    int M, K, nchunks, col0, act;   // act: 0 none, 1 relu, 2 gated-attention scores (below)
    float beta;                     // y = act(acc + bias) + beta * y_old   (residual adds)
    // act == 2 (ND = 8, one chunk): the 256 output columns are the attention pre-activations in the order [Wv units 32p..32p+31 |
    // Wu units 32p..32p+31], p = 0..3 (so accumulator tiles 2p / 2p+1 hold the tanh / sigmoid branch of the SAME units in the same
    // registers); the epilogue forms gate = tanh(.) * sigmoid(.) in registers and writes only the K raw scores per row:
    // scores[k][row] = sum_u gate[u] ww[k][u] + bw[k]   (Attention_Gated.forward, architecture/transformer.py:259-267) -- no y store
    const float* ww;                // [kb][128]
    const float* bw;                // [kb]
    float* scores;                  // [kb][M]
    int kb;
    unsigned* status;               // or null: bit 1 is OR-ed in when an output of a VALID row is >= 65504 in magnitude, inf or NaN -- the
                                    // range rule of the split-f16 arithmetic (ga_forward_kernel_v2.h) for consumers that split y again
    // FX & 1 (row-affine prologue: LayerNorm folded into the product, transMIL.py:25-28): the B operand is x[r][k] * rowab[r][0] +
    // rowab[r][1] (= (x - mean) * rstd, formed in registers right before the f16 split; gamma is folded into the packed weights and
    // W beta is the bias).  Rows r < zrows are the zero padding of the sequence (rowab = 0, 0): they get NO bias either.
    const float* rowab;             // [M][2]
    int zrows;
    // FX & 4 (with FX & 1; round 6): the row statistics are formed IN this kernel's K loop instead of by a pass of their own over x
    // (tm_rowstats_kernel: 154 MB read per layer at cfg4 for two numbers per row).  W ((x - mean) rstd) = rstd (W x - mean (W 1)):
    // the B operand is the RAW x, every lane adds up sum x and sum x^2 of its row while the values pass through its registers on
    // their way to the f16 split (one add + one FMA per value instead of the normalising FMA), and the epilogue forms
    // rstd * (acc - mean * wsum[col]) before the bias; wsum[c] = sum_k (W o gamma)[c][k].  rowab is not read.
    const float* wsum;              // [n_out of this launch]
    // FX & 2 (landmark partials, nystrom_attention.py:95-111): every wave tile also leaves the column sums of its 32 output rows,
    // split at the landmark boundary that may cross it: lm_part[(m0 / 32) * 2 + part][lm_cols] for the columns < lm_cols (q and k);
    // part 0 = rows of landmark (m0 / lm_l), part 1 = rows of the next one.  Fixed summation order (bitwise reproducible); needs lm_l >= 32.
    float* lm_part;
    int lm_l, lm_cols;
    int nt_store;                   // non-zero: the epilogue's y stores bypass the L2 (outputs beyond the Infinity Cache)
};

// WV = 4: two 4-wave workgroups per CU, 128-row tiles (default).  WV = 8 (round 5, ND = 8 / 4 only): ONE 8-wave workgroup per CU, 256-row
// tiles -- the weight stream crosses L2 -> LDS once per 256 rows (the Linear launches are bound by the CU's LDS-DMA / store path, not by the
// matrix pipe: profiles/r04_lin64_ablation.md), still two waves per SIMD (lin64_kernel has the same bytes but one wave per SIMD).
template <int ND, int XDT, int FX = 0, int WV = 4>
__global__ __launch_bounds__(64 * WV, 2) void lin_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Ga2Geom<ND, 1, XDT, WV>;
    constexpr int NTHR = 64 * WV, PD = G::PD, NB = G::NB;
    // lo plane: fp32 activations, and any LayerNorm-folded operand (an affine image of a 16-bit value is not f16-exact); a bf16
    // operand is f16-exact like an fp16 one (ga_forward_kernel_v2.h) and is only converted
    constexpr bool XLO = (XDT == ACMIL_DTYPE_F32) || (FX & 1);
    constexpr bool XCV = XLO || (XDT != ACMIL_DTYPE_F16);
    constexpr bool LOSKIP = LIN_LOSKIP && (XDT == ACMIL_DTYPE_F32) && FX == 0;      // raw fp32 rows only (a normalised operand has real lo halves)
    constexpr bool NORM = (FX & 1) != 0, LMP = (FX & 2) != 0, INSTAT = (FX & 4) != 0;
    static_assert(!INSTAT || NORM, "FX & 4 extends the LayerNorm fold");
    static_assert(PD == 2 && NB == 3, "wait counts assume a prefetch distance of 2 steps");
    static_assert(G::REGION >= 4608 || !G::SCRATCH_IN_RING, "transposition tile must fit the free slot");
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto opaque_lane = [&]() { int l = tid & 63; asm volatile("" : "+v"(l)); return l; };
    const int K = a.K, M = a.M;
    const int S1 = K / 16;
    const int rtiles = (M + G::ROWS - 1) / G::ROWS;
    // Tile queues.  The column chunks of one row tile read the same rows of x, and each XCD has its own L2: a launch of a multiple of
    // 8 workgroups (the persistent grid: 2 per CU) keeps ONE QUEUE PER XCD -- workgroup b belongs to XCD b % 8 and only ever takes
    // tiles of row tiles rt = 8 i + (b % 8), chunk index fastest, from its XCD's counter -- so the second and later readers of a row
    // tile hit in that L2 for the WHOLE launch (round 3 arranged this for the first gridDim.x draws only: 543 MB fetched for 154 MB of
    // activations by the 4-chunk to_qkv launch).  Queues differ by at most one row tile; nobody steals.  Other grids: one queue.
    const bool xq = a.tile_counter != nullptr && (gridDim.x & 7) == 0;
    const int xcd = xq ? (int)(blockIdx.x & 7) : 0;
    const int nloc = xq ? (int)(gridDim.x >> 3) : (int)gridDim.x;                  // workgroups that share this queue
    const int ntiles = xq ? ((rtiles - xcd + 7) >> 3) * a.nchunks : rtiles * a.nchunks;
    const size_t chunk_bytes = (size_t)S1 * G::WROWS * GA_FRAG_ROW;
    const size_t rowb = (size_t)a.ldx * G::XE;

    struct TileInfo { int m0, rmax, col; const char* xrow0; const char* wreg0; };
    // queue index -> (row tile, column chunk).  Single queue (static or odd grids): the first gridDim.x tiles are laid out so that the
    // chunks of a row tile start on workgroups of one XCD; past the last full group of 8 row tiles the order is plain.
    const int xcd_span = (rtiles / 8) * 8 * a.nchunks;
    auto tile_info = [&](int t) {
        TileInfo ti;
        int rt = t / a.nchunks, c = t - rt * a.nchunks;
        if (xq) rt = 8 * rt + xcd;
        else if (LIN_XCD_ORDER && t < xcd_span) {
            const int xcd = t & 7, q = t >> 3;
            c = q % a.nchunks;
            rt = (q / a.nchunks) * 8 + xcd;
        }
        ti.m0 = rt * G::ROWS + wave * 32;
        const int m0c = ti.m0 < M ? ti.m0 : M - 1;
        ti.rmax = M - 1 - m0c;
        ti.col = c * 32 * ND;
        ti.xrow0 = (const char*)a.x + (size_t)m0c * rowb;
        ti.wreg0 = a.packed + c * chunk_bytes + (size_t)(wave * G::RW + G::RW) * GA_FRAG_ROW;
        return ti;
    };
    auto tile_xoff = [&](const TileInfo& ti, int lane, unsigned (&xo)[G::XG]) {
#pragma unroll
        for (int q = 0; q < G::XG; ++q) {
            int r, piece;
            if constexpr (G::XG == 2) { r = 16 * q + (lane >> 2); piece = (lane & 3) ^ ((r >> 2) & 3); }
            else { r = lane >> 1; piece = (lane & 1) ^ ((r >> 3) & 1); }
            r = r < ti.rmax ? r : ti.rmax;
            xo[q] = (unsigned)r * (unsigned)rowb + piece * 16;
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const unsigned m0w = lds_base + wave * G::REGION + G::RW * 1024;
    auto dma_piece = [&](auto mc, int u, int slot, unsigned woff, const TileInfo& ti, const unsigned (&xo)[G::XG]) {
        constexpr int m = decltype(mc)::value;
        const unsigned m0v = m0w + slot * G::SLOT;
        if constexpr (m < G::RW) { if constexpr (!(LIN_ABL & 2)) ga2_dma<-(G::RW - m) * 1024>(woff, ti.wreg0 + (size_t)u * G::WROWS * GA_FRAG_ROW, m0v); }
        else if constexpr (m < G::NVX) { constexpr int q = m - G::RW; if constexpr (!(LIN_ABL & 1)) ga2_dma<q * 1024>(xo[q], ti.xrow0 + (size_t)u * 16 * G::XE - q * 1024, m0v); }
    };
#define LIN_DMA_AT(d, ...)                                   \
    do {                                                     \
        if ((d) == 0) dma_piece(I0{}, __VA_ARGS__);          \
        if ((d) == 1) dma_piece(I1{}, __VA_ARGS__);          \
        if ((d) == 2) dma_piece(I2{}, __VA_ARGS__);          \
        if ((d) == 3) dma_piece(I3{}, __VA_ARGS__);          \
        if ((d) == 4) dma_piece(I4{}, __VA_ARGS__);          \
        if ((d) == 5) dma_piece(I5{}, __VA_ARGS__);          \
    } while (0)

    // tile drawing (see ga_fwd2_kernel)
    unsigned* const nn_lds = (unsigned*)(smem + G::ML_OFF);
    unsigned draw_raw = 0;
    auto draw_issue = [&]() {
        unsigned long long keep;
        const unsigned zero = 0, one = 1;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                     : "=&v"(draw_raw), "=&s"(keep) : "v"(zero), "v"(one), "s"(a.tile_counter + xcd) : "memory");
    };
    auto draw_publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(draw_raw) :: "memory");
        const unsigned v = __builtin_amdgcn_readfirstlane(draw_raw) + (unsigned)nloc;
        if ((tid & 63) == 0) *nn_lds = v;
    };
    const bool dynamic = a.tile_counter != nullptr;
    int tile = xq ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int ntile = tile + nloc;
    if (tile >= ntiles) {       // (a queue shorter than its share of the grid: nothing to do but to be counted)
        if (dynamic && a.done && tid == 0) {
            const unsigned d = atomicAdd(a.done, 1u);
            if (d == gridDim.x - 1) { atomicExch(a.done, 0u); for (int q = 0; q < 8; ++q) atomicExch(a.tile_counter + q, 0u); }
        }
        return;
    }
    if (dynamic) {
        if (wave == 0) { draw_issue(); draw_publish(); }
        __syncthreads();
        ntile = (int)__builtin_amdgcn_readfirstlane(*nn_lds);
    }
    TileInfo T = tile_info(tile);
    // FX & 1: (a, b) of this lane's row of x (row = lane & 31 of the wave tile, clamped like the DMA), one tile ahead
    auto load_ab = [&](const TileInfo& ti) {
        f32x2 ab = {1.0f, 0.0f};
        if constexpr (NORM && !INSTAT) {
            const int i31 = (int)(tid & 31);
            const int m0c = ti.m0 < M ? ti.m0 : M - 1;
            ab = *(const f32x2*)(a.rowab + 2 * (size_t)(m0c + (i31 < ti.rmax ? i31 : ti.rmax)));
        }
        return ab;
    };
    f32x2 rab = load_ab(T);

    int islot = 0, rslot = 0;
    {
        const int ln = opaque_lane();
        unsigned xo[G::XG];
        tile_xoff(T, ln, xo);
#pragma unroll
        for (int s = 0; s < PD; ++s) {
#pragma unroll
            for (int m = 0; m < G::NVX; ++m) LIN_DMA_AT(m, s, islot, (unsigned)ln * 16, T, xo);
            islot = (islot + 1 == NB) ? 0 : islot + 1;
        }
    }
    auto step_sync = [&]() {
        ga_wait_vm<G::NVX>();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };

    for (;;) {
        const bool has_next = ntile < ntiles;
        const TileInfo TN = tile_info(has_next ? ntile : tile);
        const f32x2 rabn = load_ab(TN);
        if (dynamic && has_next && wave == 0) draw_issue();
        __builtin_amdgcn_s_waitcnt(0xc07f);

        f32x16 acc[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][r] = 0.0f;
        {
            const int ln = opaque_lane();
            const int i31 = ln & 31, hi = ln >> 5;
            const int lane16 = ln * 16;
            unsigned xoff[G::XG], xoffn[G::XG];
            tile_xoff(T, ln, xoff);
            tile_xoff(TN, ln, xoffn);
            const int xrd0 = wave * G::REGION + G::RW * 1024 +
                             ((G::XG == 2) ? (i31 * 64 + (((2 * hi) ^ ((i31 >> 2) & 3)) * 16)) : (i31 * 32 + ((hi ^ ((i31 >> 3) & 1)) * 16)));
            const int xrd1 = wave * G::REGION + G::RW * 1024 + i31 * 64 + (((2 * hi + 1) ^ ((i31 >> 2) & 3)) * 16);
            f32x4 xr0, xr1;
            u32x4 xrw;
            auto read_x = [&](const char* slot) {
                if constexpr (XDT == ACMIL_DTYPE_F32) { xr0 = *(const f32x4*)(slot + xrd0); xr1 = *(const f32x4*)(slot + xrd1); }
                else xrw = *(const u32x4*)(slot + xrd0);
            };
            constexpr int NSP = XCV ? 4 : 0;
            u32x4 xhw, xlw;
            float rsum = 0.0f, rsq = 0.0f;      // INSTAT: sum x, sum x^2 of this lane's K slots of its row
            auto split_piece = [&](int j) {
                float v0, v1;
                if constexpr (XDT == ACMIL_DTYPE_F32) {
                    v0 = j < 2 ? xr0[2 * (j & 1)] : xr1[2 * (j & 1)];
                    v1 = j < 2 ? xr0[2 * (j & 1) + 1] : xr1[2 * (j & 1) + 1];
                } else {
                    v0 = __builtin_bit_cast(float, xrw[j] << 16);
                    v1 = __builtin_bit_cast(float, xrw[j] & 0xffff0000u);
                }
                if constexpr (INSTAT) { rsum += v0 + v1; rsq = fmaf(v0, v0, rsq); rsq = fmaf(v1, v1, rsq); }
                else if constexpr (NORM) { v0 = fmaf(v0, rab[0], rab[1]); v1 = fmaf(v1, rab[0], rab[1]); }
                if constexpr (XLO) {
                    unsigned h, l;
                    ga2_split_pair(v0, v1, h, l);
                    xhw[j] = h; xlw[j] = l;
                } else xhw[j] = ga_cvt_pair_f16(v0, v1);
            };
            auto split_done = [&](f16x8& h8, f16x8& l8) {
                if constexpr (XLO) { h8 = __builtin_bit_cast(f16x8, xhw); l8 = __builtin_bit_cast(f16x8, xlw); }
                else if constexpr (XCV) h8 = __builtin_bit_cast(f16x8, xhw);
                else h8 = __builtin_bit_cast(f16x8, xrw);
            };
            f16x8 WH[ND], WL[ND];
            f16x8 xh, xl, xhp;
#pragma unroll
            for (int d = 0; d < ND; ++d) WL[d] = (f16x8)(_Float16)0.0f;
            xhp = (f16x8)(_Float16)0.0f;
            __builtin_amdgcn_s_setprio(2);
            for (int s = 0; s < S1; ++s) {
                step_sync();
                const char* slot = smem + rslot * G::SLOT;
                read_x(slot);
#pragma unroll
                for (int d = 0; d < ND; ++d) WH[d] = *(const f16x8*)(slot + G::frow(d) + lane16);
                __builtin_amdgcn_sched_barrier(0);
                if (s > 0) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WL[d], xhp, acc[d], 0, 0, 0);
                        if (d < NSP) {
                            __builtin_amdgcn_sched_barrier(0);
                            split_piece(d);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NSP; ++j) split_piece(j);
                }
                split_done(xh, xl);
                // an fp32 operand whose values are f16-exact (bags stored fp16 and up-cast by the loop, Step3_WSI_classification_ACMIL.py:193)
                // has lo halves of exact zeros: skip the W_hi x_lo group of this wave and step (ga_forward_kernel_v2.h; same numbers)
                bool lo_any = true;
                if constexpr (LOSKIP) {
                    const unsigned lo_or = (xlw[0] | xlw[1] | xlw[2] | xlw[3]) & 0x7fff7fffu;
                    lo_any = __builtin_amdgcn_ballot_w64(lo_or != 0u) != 0ull;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < ND; ++d) WL[d] = *(const f16x8*)(slot + G::frow(ND + d) + lane16);
                // step s+2 of this tile, or step s+2-S1 of the next one
                const bool nx = s + PD >= S1;
                const int un = nx ? s + PD - S1 : s + PD;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH[d], xh, acc[d], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (nx) LIN_DMA_AT(d, un, islot, (unsigned)lane16, TN, xoffn); else LIN_DMA_AT(d, un, islot, (unsigned)lane16, T, xoff);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (ND < G::NVX) {   // more pieces than MFMAs in the group (narrow chunks): issue the rest here
#pragma unroll
                    for (int d = ND; d < G::NVX; ++d) { if (nx) LIN_DMA_AT(d, un, islot, (unsigned)lane16, TN, xoffn); else LIN_DMA_AT(d, un, islot, (unsigned)lane16, T, xoff); }
                }
                islot = (islot + 1 == NB) ? 0 : islot + 1;
                if constexpr (XLO) {
                    if (lo_any) {
#pragma unroll
                        for (int d = 0; d < ND; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH[d], xl, acc[d], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                xhp = xh;
                rslot = (rslot + 1 == NB) ? 0 : rslot + 1;
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WL[d], xhp, acc[d], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (INSTAT) {
                // the two lane halves hold complementary K slots of the same row: fold, then mean / rstd of LayerNorm (eps 1e-5,
                // transMIL.py:12) -- and the accumulators become the product with the NORMALISED row right here, in their own layout
                // (lane = row, register = output column), so that everything below (bias, stores, landmark sums) is unchanged
                rsum += __shfl_xor(rsum, 32);
                rsq += __shfl_xor(rsq, 32);
                const float invk = 1.0f / (float)a.K;
                const float mean = rsum * invk;
                const float var = fmaxf(fmaf(rsq, invk, -mean * mean), 0.0f);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                const float mr = -mean * rstd;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 w4 = *(const f32x4*)(a.wsum + T.col + 32 * d + 8 * q + 4 * hi);      // columns of registers 4q .. 4q + 3
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[d][4 * q + e] = fmaf(acc[d][4 * q + e], rstd, mr * w4[e]);
                    }
            }
        }

        // ======================================================= epilogue (gated scores): gate in registers, K scores per row
        bool gated = false;
        if constexpr (ND == 8) gated = a.act == 2;
        if (gated) {
            if constexpr (ND == 8) {
                const int lane = opaque_lane();
                const int i31 = lane & 31, hi = lane >> 5;
                constexpr int KPG = ACMIL_MAX_TOKENS;
                float sc[KPG];
#pragma unroll
                for (int k = 0; k < KPG; ++k) sc[k] = 0.0f;
                const int kb = a.kb;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int ub = 8 * rq + 4 * hi;                        // registers 4rq..4rq+3 <-> units 32p + ub + {0..3}
                        const f32x4 bv = *(const f32x4*)(a.bias + 64 * p + ub);
                        const f32x4 bu = *(const f32x4*)(a.bias + 64 * p + 32 + ub);
                        float gate[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) gate[q] = ga_tanh(acc[2 * p][4 * rq + q] + bv[q]) * ga_sigmoid(acc[2 * p + 1][4 * rq + q] + bu[q]);
#pragma unroll
                        for (int k = 0; k < KPG; ++k) {
                            if (k < kb) {
                                const f32x4 w = *(const f32x4*)(a.ww + k * GA_DA + 32 * p + ub);
                                sc[k] = fmaf(gate[0], w[0], sc[k]); sc[k] = fmaf(gate[1], w[1], sc[k]);
                                sc[k] = fmaf(gate[2], w[2], sc[k]); sc[k] = fmaf(gate[3], w[3], sc[k]);
                            }
                        }
                    }
                // the two lane halves cover complementary units of the same 32 rows: fold, add bw, one half stores the even branches
                // and the other the odd ones (32 consecutive floats of a score row each)
                const int row = T.m0 + i31;
#pragma unroll
                for (int k = 0; k < KPG; ++k) {
                    if (k < kb) {
                        const float tot = sc[k] + __shfl_xor(sc[k], 32) + a.bw[k];
                        if ((k & 1) == hi && row < M) a.scores[(size_t)k * M + row] = tot;
                    }
                }
            }
        } else if constexpr (LIN_ABL & 8) {
            float keep = 0.0f;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[d][r];
            if (keep == 123456.789f) a.y[0] = keep;
        } else
        // ======================================================= epilogue: transpose 32 x 32 tiles through the free slot, store rows
        {
            const int lane = opaque_lane();
            const int i31 = lane & 31, hi = lane >> 5;
            const int fslot = (rslot == 0) ? NB - 1 : rslot - 1;
            float* pool = (float*)(G::SCRATCH_IN_RING ? smem + fslot * G::SLOT + wave * G::REGION : smem + G::SCR_OFF + wave * G::PW);
            // The transposition tile lives in the ring slot the LAST K step read, and every wave reads ALL fragment rows of that slot:
            // a wave that is done with its last step must not write its tile while a slower wave still reads fragments there.  Round 4
            // found exactly that (one wave's accumulators of the three column blocks whose fragment rows another wave had already
            // overwritten with fp32 values: 1e5 / NaN blocks in to_qkv / to_out outputs) -- only with a second TransMIL forward on another
            // stream, whose short Moore-Penrose workgroups share a SIMD with ONE wave of this workgroup and hold it back by more than
            // a K step; the barrier closes it for any skew (all fragment reads of the last step are retired before anyone writes).
            if constexpr (G::SCRATCH_IN_RING) {
                __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's fragment reads have returned
                __builtin_amdgcn_s_barrier();
            }
            const int m0 = T.m0;
            if (a.status) {      // largest |pre-activation| of this lane's row as a bit pattern (finite < inf < NaN), sign shifted out
                unsigned hm = 0u;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned b = __builtin_bit_cast(unsigned, acc[d][r]) << 1;
                        hm = hm > b ? hm : b;
                    }
                const bool bad = (m0 + i31 < M) && hm >= (0x477fe000u << 1);
                if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(a.status, 2u);
            }
            // Per 32-column tile: accumulators -> wave-private tile [32 rows (patches)][36] (lane = row, register = column), read back
            // as float4 = 4 consecutive output columns of one row: 8 lanes store 128 contiguous bytes of an output row with ONE 16-byte
            // store each (4 stores per lane and tile; the first version stored 16 scalars per lane: the epilogue was store-issue-bound)
            const int rsub = lane >> 3, cq = lane & 7;               // row within a group of 8, column quad
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) pool[i31 * 36 + mfma32_row(r, hi)] = acc[c][r];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const int col = T.col + 32 * c + 4 * cq;
                f32x4 b4 = {0.0f, 0.0f, 0.0f, 0.0f};
                if (a.bias) b4 = f32x4{a.bias[col], a.bias[col + 1], a.bias[col + 2], a.bias[col + 3]};      // (no alignment demand on the bias)
                float* yc = a.y + (size_t)(a.col0 + col);
                f32x4 old[4];
                if (a.beta != 0.0f) {      // residual form: all old values first (unconditional loads from clamped rows), then the stores
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = m0 + 8 * it + rsub;
                        old[it] = *(const f32x4*)(yc + (size_t)(row < M ? row : M - 1) * a.ldy);
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = m0 + 8 * it + rsub;
                    f32x4 v = *(const f32x4*)(pool + (8 * it + rsub) * 36 + 4 * cq);
                    if (!NORM || row >= a.zrows) v = v + b4;
                    if (a.act == 1) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
                    if (a.beta != 0.0f) v = v + old[it] * a.beta;
                    if (row < M) {
                        // outputs larger than the Infinity Cache are streamed past the L2 (non-temporal): written once, read by a later
                        // kernel from HBM anyway, and kept out of the L2 they no longer evict the x rows the other column chunks of
                        // the row tile are about to re-read (to_qkv: 910 -> 741 MB per launch, 310 -> 302 us)
                        if (a.nt_store) lin_store_nt(yc + (size_t)row * a.ldy, v);
                        else *(f32x4*)(yc + (size_t)row * a.ldy) = v;
                    }
                }
                if constexpr (LMP) {
                    // column sums of the tile's 32 rows, split at the landmark boundary; the transposition tile still holds the raw
                    // accumulators (this launch has no activation and no residual: accumulator + bias is the stored value)
                    const int colL = a.col0 + T.col + 32 * c;                     // (column of the whole output, not of this launch)
                    if (colL < a.lm_cols && m0 < M) {
                        const int bnd = (m0 / a.lm_l + 1) * a.lm_l - m0;          // first row (tile-relative) of the next landmark
                        const float bcol = a.bias ? a.bias[T.col + 32 * c + i31] : 0.0f;
                        float sA = 0.0f, sB = 0.0f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = 16 * hi + r;
                            float v = pool[rr * 36 + i31];
                            if (!NORM || m0 + rr >= a.zrows) v += bcol;           // the value that was stored for this row
                            if (rr < bnd) sA += v; else sB += v;
                        }
                        const float oA = __shfl_xor(sA, 32), oB = __shfl_xor(sB, 32);
                        // rows 0..15 first, then rows 16..31, on both halves: the same order whichever half stores
                        const float tA = hi ? oA + sA : sA + oA, tB = hi ? oB + sB : sB + oB;
                        if (!hi || bnd < 32)     // (part 1 is only read for tiles a landmark boundary crosses)
                            a.lm_part[((size_t)(m0 >> 5) * 2 + hi) * a.lm_cols + colL + i31] = hi ? tB : tA;
                    }
                }
            }
        }
        if (!has_next) break;
        if (dynamic) {
            if (wave == 0) draw_publish();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        tile = ntile;
        rab = rabn;
        ntile = dynamic ? (int)__builtin_amdgcn_readfirstlane(*nn_lds) : tile + nloc;
        T = TN;
    }
#undef LIN_DMA_AT
    ga_wait_vm<0>();      // (also: every tile draw of this workgroup has returned)
    if (dynamic && a.done) {
        __syncthreads();
        if (tid == 0) {
            const unsigned d = atomicAdd(a.done, 1u);
            if (d == gridDim.x - 1) { atomicExch(a.done, 0u); for (int q = 0; q < 8; ++q) atomicExch(a.tile_counter + q, 0u); }
        }
    }
}
