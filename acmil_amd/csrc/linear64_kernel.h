// linear64_kernel.h -- the packed-weight Linear kernel (linear_kernel.h) re-cut for ONE 512-register wave per SIMD (round 4).
//
// Same mathematics, same packed fragment stream, same epilogues as lin_kernel (y = act(x W^T + b) + beta y, optional LayerNorm-folded
// operand and landmark partials: TransMIL._fc1 / to_qkv / to_out, transMIL.py:51,63, nystrom_attention.py:80,139; DimReduction.fc1
// at the wide D_inner families, network.py:49-57).  What changes is the shape of the work per wave:
//   * a workgroup = 4 waves = ONE per SIMD (amdgpu_waves_per_eu(1,1)): each wave owns the whole 512-entry register file --
//     2 x ND 32x32 accumulator tiles (64 rows x 32 ND columns: up to 256 registers, the AGPR half) + operands in the arch VGPRs;
//   * every weight fragment read from LDS feeds TWO MFMA column blocks (the wave's two 32-row blocks), and the L2 -> LDS weight
//     stream is copied once per 256 rows (lin_kernel: once per 128 rows, twice per CU): half the LDS-DMA pieces and half the
//     fragment reads per MFMA -- the round-3 ablations name that stream as the largest power consumer after the matrix pipe;
//   * the activations x go through a wave-private LDS-DMA ring as in lin_kernel (16 rows x 64 B per piece).  The first cut fetched
//     them global -> registers in the MFMA B layout (16 bytes per lane, every lane another row): 64 uncoalesced requests per
//     instruction cost 73 us of the 347 us to_qkv launch (tools/abl_lin64.py) -- the texture addresser, not HBM;
//   * 4-slot rings, prefetch distance 3; the K loop is unrolled by 4 so slot indices are static;
//   * with one wave per SIMD nothing hides a stall: the LDS reads of a step and the f16 split of its x values sit between the
//     MFMAs of the previous step's deferred group; and the EPILOGUE stores straight from the accumulator layout (a lane owns 4
//     consecutive columns of its row per register quad: 16-byte stores, no LDS round trip -- the transposing epilogue of lin_kernel
//     took 92 of those 347 us here, where no second workgroup computes meanwhile).  CDNA4 counts stores in vmcnt, so the loads of
//     the next tile's first three steps are drained BEFORE the stores are issued and those steps run without a memory wait.
// Needs K % 64 == 0 (four K steps per unrolled round).  Callers: linear.hip (lin_use64).
#pragma once
#include "linear_kernel.h"

// timing-only ablations (tools/build_lin_variants.sh; WRONG results): 1 no x loads, 2 no weight DMA, 4 no MFMAs, 8 no epilogue,
// 16 no step barrier, 32 no LDS fragment reads, 64 all tiles store to rows 0..255 (L2-resident).  0 in every product build.
#ifndef LIN64_ABL
#define LIN64_ABL 0
#endif

template <int ND, int XDT>
struct Lin64Geom {
    static constexpr int NB = 4, PD = 3;                            // ring slots, prefetch distance (steps)
    static constexpr int RW = 2 * ND / 4;                           // fragment rows each wave copies per step
    static_assert((2 * ND) % 4 == 0 && RW >= 1 && RW <= 4, "row immediates 0..3072");
    static constexpr int SLOT = 2 * ND * GA_FRAG_ROW;               // hi rows then lo rows of one K step
    static constexpr int XE = (XDT == ACMIL_DTYPE_F32) ? 4 : 2;
    static constexpr int XG = 64 * 16 * XE / 1024;                  // x pieces (1 KiB) per wave and step: 4 (fp32) / 2 (16-bit)
    static constexpr int XSLOT = XG * 1024;                         // a wave's 64 rows x 16 K values
    static constexpr int NV = RW + XG;                              // LDS-DMA pieces per wave and step
    static constexpr int ROWS = 256;                                // rows per workgroup tile
    static constexpr int RING = NB * SLOT;                          // weight ring (shared by the 4 waves)
    static constexpr int XRING_OFF = RING;                          // x rings: [wave][slot][XSLOT]
    static constexpr int POOL = 4608;                               // wave-private [32][36] fp32 tile (landmark column sums only)
    static constexpr int POOL_OFF = XRING_OFF + 4 * NB * XSLOT;
    static constexpr int BIAS_OFF = POOL_OFF + 4 * POOL;            // bias of the launch's columns (<= 2048 floats)
    static constexpr int BIAS_MAX = 2048;
    static constexpr int NN_OFF = BIAS_OFF + BIAS_MAX * 4;
    static constexpr int LDS = NN_OFF + 64;
    static_assert(LDS <= 160 * 1024, "one workgroup per CU");
};

typedef int LinPlan_cols_t;
#if LIN64_ABL & 4
#define LIN64_MFMA(A, B, C, X, Y, Z) (C)
#else
#define LIN64_MFMA(A, B, C, X, Y, Z) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, X, Y, Z)
#endif
template <int ND, int XDT, int FX = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lin64_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Lin64Geom<ND, XDT>;
    constexpr int NB = G::NB, PD = G::PD, RW = G::RW, XG = G::XG, NV = G::NV;
    constexpr bool XLO = (XDT == ACMIL_DTYPE_F32) || (FX & 1);      // (a bf16 operand is f16-exact like an fp16 one: converted, no lo plane)
    constexpr bool XCV = XLO || (XDT != ACMIL_DTYPE_F16);
    constexpr bool LOSKIP = LIN_LOSKIP && (XDT == ACMIL_DTYPE_F32) && FX == 0;
    constexpr bool NORM = (FX & 1) != 0, LMP = (FX & 2) != 0;
    static_assert(NB == 4 && PD == 3, "the K loop is unrolled by 4: slot and register-set indices are compile-time");
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto opaque_lane = [&]() { int l = tid & 63; asm volatile("" : "+v"(l)); return l; };
    const int K = a.K, M = a.M;
    const int S1 = K / 16;                                          // multiple of 4
    const int rtiles = (M + G::ROWS - 1) / G::ROWS;
    // tile queues: one per XCD when the grid is a multiple of 8 (see lin_kernel)
    const bool xq = a.tile_counter != nullptr && (gridDim.x & 7) == 0;
    const int xcd = xq ? (int)(blockIdx.x & 7) : 0;
    const int nloc = xq ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int ntiles = xq ? ((rtiles - xcd + 7) >> 3) * a.nchunks : rtiles * a.nchunks;
    const size_t chunk_bytes = (size_t)S1 * 2 * ND * GA_FRAG_ROW;
    const unsigned rowb = (unsigned)((size_t)a.ldx * G::XE);

    struct TileInfo { int m0, col; const char* wsrc; };
    auto tile_info = [&](int t) {
        TileInfo ti;
        int rt = t / a.nchunks;
        const int c = t - rt * a.nchunks;
        if (xq) rt = 8 * rt + xcd;
        ti.m0 = rt * G::ROWS + wave * 64;
        ti.col = c * 32 * ND;
        ti.wsrc = a.packed + c * chunk_bytes + (size_t)(wave * RW) * GA_FRAG_ROW;     // this wave's rows of step 0
        return ti;
    };
    // source offsets of the x pieces of a step (rows clamped to the last valid one).  fp32: piece q = rows 16 q .. 16 q + 15, a
    // row's 64 bytes as four 16-byte chunks held by 4 lanes; 16-bit: piece q = rows 32 q .. 32 q + 31, two chunks per row.  The chunk
    // a lane fetches is XOR-swizzled with its row so that the ds_read_b128 of the MFMA B layout below are conflict-free (lin_kernel)
    auto tile_xoff = [&](const TileInfo& ti, int lane, unsigned (&xo)[XG]) {
#pragma unroll
        for (int q = 0; q < XG; ++q) {
            int rl, piece;
            if constexpr (G::XE == 4) { rl = 16 * q + (lane >> 2); piece = (lane & 3) ^ ((rl >> 2) & 3); }
            else { rl = 32 * q + (lane >> 1); piece = (lane & 1) ^ ((rl >> 3) & 1); }
            int r = ti.m0 + rl;
            r = r < M ? r : M - 1;
            xo[q] = (unsigned)r * rowb + piece * 16;
        }
    };
    auto load_ab = [&](const TileInfo& ti, f32x2 (&ab)[2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            ab[b] = f32x2{1.0f, 0.0f};
            if constexpr (NORM) {
                int r = ti.m0 + 32 * b + (int)(tid & 31);
                r = r < M ? r : M - 1;
                ab[b] = *(const f32x2*)(a.rowab + 2 * (size_t)r);
            }
        }
    };

    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const unsigned ring_w = lds_base + wave * RW * GA_FRAG_ROW;     // this wave's rows inside slot 0
    // weight pieces of step u of tile ti into ring slot `slot` (RW pieces, issued one at a time by the caller)
    auto dma_w = [&](auto rc, int u, int slot, unsigned lane16, const TileInfo& ti) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < RW && !(LIN64_ABL & 2)) ga2_dma<r * GA_FRAG_ROW>(lane16, ti.wsrc + (size_t)u * 2 * ND * GA_FRAG_ROW, ring_w + slot * G::SLOT);
    };
    // x pieces of step u into this wave's x ring slot (XG pieces, one per call)
    const unsigned ring_x = lds_base + G::XRING_OFF + wave * NB * G::XSLOT;
    auto dma_x = [&](auto qc, int u, int slot, const unsigned (&xo)[XG]) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < XG && !(LIN64_ABL & 1))
            ga2_dma<q * 1024>(xo[q], (const char*)a.x + (size_t)u * 16 * G::XE - q * 1024, ring_x + slot * G::XSLOT);
    };
    // bias of this launch's columns -> LDS once (the epilogue reads it with broadcast ds_reads: no VMEM traffic between the stores)
    float* const bias_lds = (float*)(smem + G::BIAS_OFF);
    if (a.bias) {
        const LinPlan_cols_t ncols = 32 * ND * a.nchunks;
        for (int c = tid; c < ncols && c < G::BIAS_MAX; c += 256) bias_lds[c] = a.bias[c];
    }

    // tile drawing (as lin_kernel)
    unsigned* const nn_lds = (unsigned*)(smem + G::NN_OFF);
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
    auto count_done = [&]() {
        if (dynamic && a.done && tid == 0) {
            const unsigned d = atomicAdd(a.done, 1u);
            if (d == gridDim.x - 1) { atomicExch(a.done, 0u); for (int q = 0; q < 8; ++q) atomicExch(a.tile_counter + q, 0u); }
        }
    };
    if (tile >= ntiles) { count_done(); return; }
    if (dynamic) {
        if (wave == 0) { draw_issue(); draw_publish(); }
        __syncthreads();
        ntile = (int)__builtin_amdgcn_readfirstlane(*nn_lds);
    }
    TileInfo T = tile_info(tile);
    f32x2 rab[2];
    load_ab(T, rab);

    // (Measured without effect and removed: starting workgroup b (b / 8) % 8 eighths of a tile late so that the CUs do not all store
    //  at once -- 352 vs 346 us on the to_qkv shape: the epilogue's cost is the CU's own store path, not a chip-wide HBM burst.)
    // prologue: steps 0, 1, 2 of the first tile; they are waited for in full (as after every tile, see the epilogue)
    {
        const int ln = opaque_lane();
        unsigned xo[XG];
        tile_xoff(T, ln, xo);
        const unsigned l16 = (unsigned)ln * 16;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            dma_w(I0{}, u, u, l16, T); dma_w(I1{}, u, u, l16, T); dma_w(I2{}, u, u, l16, T); dma_w(I3{}, u, u, l16, T);
            dma_x(I0{}, u, u, xo); dma_x(I1{}, u, u, xo); dma_x(I2{}, u, u, xo); dma_x(I3{}, u, u, xo);
        }
        ga_wait_vm<0>();
        __syncthreads();       // (also publishes the bias table)
    }

    for (;;) {
        const bool has_next = ntile < ntiles;
        const TileInfo TN = tile_info(has_next ? ntile : tile);
        f32x2 rabn[2];
        load_ab(TN, rabn);
        if (dynamic && has_next && wave == 0) draw_issue();

        f32x16 acc[2][ND];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][d][r] = 0.0f;
        {
            const int ln = opaque_lane();
            const unsigned lane16 = (unsigned)ln * 16;
            const int i31 = ln & 31, hi = ln >> 5;
            unsigned xoff[XG], xoffn[XG];
            tile_xoff(T, ln, xoff);
            tile_xoff(TN, ln, xoffn);
            // read offsets of the B operand inside an x slot: block b = rows 32 b + i31
            const int xrd0 = (G::XE == 4) ? (i31 * 64 + (((2 * hi) ^ ((i31 >> 2) & 3)) * 16)) : (i31 * 32 + ((hi ^ ((i31 >> 3) & 1)) * 16));
            const int xrd1 = i31 * 64 + (((2 * hi + 1) ^ ((i31 >> 2) & 3)) * 16);
            constexpr int XBLK = 32 * 16 * G::XE;                    // bytes of one 32-row block inside a slot
            const char* const xring = smem + G::XRING_OFF + wave * NB * G::XSLOT;
            f16x8 WH[ND], WL[ND], WLp[ND];
            f16x8 xh[2], xl[2], xhp[2];
#pragma unroll
            for (int d = 0; d < ND; ++d) WLp[d] = (f16x8)(_Float16)0.0f;
            xhp[0] = xhp[1] = (f16x8)(_Float16)0.0f;

            // one K step; J = s % 4 = ring slot = x register set of this step; the loads issued here are those of step s + 3
            auto step = [&](auto jc, int s, auto waitc) {
                constexpr int J = decltype(jc)::value, JN = (J + 3) & 3;
                // everything of step s has landed: at most the groups of steps s + 1 and s + 2 are still in flight.  (The first three
                // steps of a tile were drained before the previous epilogue's stores: no memory wait, the barrier publishes them.)
                if constexpr (decltype(waitc)::value) ga_wait_vm<2 * NV>();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if constexpr (!(LIN64_ABL & 16)) __builtin_amdgcn_s_barrier();
                const char* slot = smem + J * G::SLOT;
                if constexpr (!(LIN64_ABL & 32)) {
#pragma unroll
                for (int d = 0; d < ND; ++d) WH[d] = *(const f16x8*)(slot + d * GA_FRAG_ROW + lane16);
                }
                f32x4 xr[2][2];
                u32x4 xrw[2];
                {
                    const char* xs = xring + J * G::XSLOT;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        if constexpr (XDT == ACMIL_DTYPE_F32) { xr[b][0] = *(const f32x4*)(xs + b * XBLK + xrd0); xr[b][1] = *(const f32x4*)(xs + b * XBLK + xrd1); }
                        else xrw[b] = *(const u32x4*)(xs + b * XBLK + xrd0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // deferred group of step s - 1 (Wlo * xhi) with the f16 split of this step's x values in its shadow
                u32x4 hw[2], lw[2];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[b][d] = LIN64_MFMA(WLp[d], xhp[b], acc[b][d], 0, 0, 0);
                        const int piece = 2 * d + b;                 // 8 split pairs: (block, pair) = (piece >> 2, piece & 3)
                        if (piece < 8) {
                            __builtin_amdgcn_sched_barrier(0);
                            const int bb = piece >> 2, j = piece & 3;
                            float v0, v1;
                            if constexpr (XDT == ACMIL_DTYPE_F32) {
                                const f32x4 src = xr[bb][j >> 1];
                                v0 = src[2 * (j & 1)]; v1 = src[2 * (j & 1) + 1];
                            } else if constexpr (XDT == ACMIL_DTYPE_BF16) {
                                const unsigned w = xrw[bb][j];
                                v0 = __builtin_bit_cast(float, w << 16); v1 = __builtin_bit_cast(float, w & 0xffff0000u);
                            } else {
                                const unsigned w = xrw[bb][j];
                                v0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
                                v1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
                            }
                            if constexpr (NORM) { v0 = fmaf(v0, rab[bb][0], rab[bb][1]); v1 = fmaf(v1, rab[bb][0], rab[bb][1]); }
                            if constexpr (XLO) {
                                unsigned h, l;
                                ga2_split_pair(v0, v1, h, l);
                                hw[bb][j] = h; lw[bb][j] = l;
                            } else if constexpr (XCV) {
                                hw[bb][j] = ga_cvt_pair_f16(v0, v1);
                            } else {
                                hw[bb][j] = xrw[bb][j];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) { xh[b] = __builtin_bit_cast(f16x8, hw[b]); if constexpr (XLO) xl[b] = __builtin_bit_cast(f16x8, lw[b]); }
                bool lo_any = true;      // fp32 rows of f16-exact values: no W_hi x_lo group for this wave and step (linear_kernel.h)
                if constexpr (LOSKIP) {
                    const unsigned lo_or = (lw[0][0] | lw[0][1] | lw[0][2] | lw[0][3] | lw[1][0] | lw[1][1] | lw[1][2] | lw[1][3]) & 0x7fff7fffu;
                    lo_any = __builtin_amdgcn_ballot_w64(lo_or != 0u) != 0ull;
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(LIN64_ABL & 32)) {
#pragma unroll
                for (int d = 0; d < ND; ++d) WL[d] = *(const f16x8*)(slot + (ND + d) * GA_FRAG_ROW + lane16);
                }
                // Whi * xhi, with the loads of step s + 3 (this tile's, or step s + 3 - S1 of the next) one per MFMA gap
                const bool nx = s + PD >= S1;
                const int un = nx ? s + PD - S1 : s + PD;
                // (selected once per step: a branch per piece cost 10 scalar branches per step)
                TileInfo TS; TS.m0 = 0; TS.col = 0; TS.wsrc = nx ? TN.wsrc : T.wsrc;
                unsigned xsel[XG];
#pragma unroll
                for (int q = 0; q < XG; ++q) xsel[q] = nx ? xoffn[q] : xoff[q];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[b][d] = LIN64_MFMA(WH[d], xh[b], acc[b][d], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        const int g = 2 * d + b;
                        if (g == 0) dma_w(I0{}, un, JN, lane16, TS);
                        if (g == 1) dma_w(I1{}, un, JN, lane16, TS);
                        if (g == 2) dma_w(I2{}, un, JN, lane16, TS);
                        if (g == 3) dma_w(I3{}, un, JN, lane16, TS);
                        if (g == 4) dma_x(I0{}, un, JN, xsel);
                        if (g == 5) dma_x(I1{}, un, JN, xsel);
                        if (g == 6) dma_x(I2{}, un, JN, xsel);
                        if (g == 7) dma_x(I3{}, un, JN, xsel);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if constexpr (XLO) {
                    if (lo_any) {
#pragma unroll
                        for (int d = 0; d < ND; ++d)
#pragma unroll
                            for (int b = 0; b < 2; ++b) acc[b][d] = LIN64_MFMA(WH[d], xl[b], acc[b][d], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < ND; ++d) WLp[d] = WL[d];
                xhp[0] = xh[0]; xhp[1] = xh[1];
            };
            __builtin_amdgcn_s_setprio(2);
            step(I0{}, 0, I0{}); step(I1{}, 1, I0{}); step(I2{}, 2, I0{}); step(I3{}, 3, I1{});
            for (int s = 4; s < S1; s += 4) { step(I0{}, s, I1{}); step(I1{}, s + 1, I1{}); step(I2{}, s + 2, I1{}); step(I3{}, s + 3, I1{}); }
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[b][d] = LIN64_MFMA(WLp[d], xhp[b], acc[b][d], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }

        // ======================================================= epilogue
        // The loads of the next tile's steps 0..2 are in flight: wait for them BEFORE the stores go out (vmcnt counts stores, in order
        // with the loads: a counted wait behind 48+ stores would stall the first K steps for the stores' whole flight).
        ga_wait_vm<0>();
        if constexpr (LIN64_ABL & 8) {      // keep the accumulators (and so the MFMAs) alive without an epilogue
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int d = 0; d < ND; ++d) asm volatile("" :: "a"(acc[b][d]));
        } else {
            // C/D layout of a 32 x 32 tile: lane (i31 = row of x, hi), register r <-> column (r & 3) + 8 (r >> 2) + 4 hi: the quad
            // r = 4 g .. 4 g + 3 is 4 CONSECUTIVE output columns of the lane's own row -> one 16-byte store per quad, bias / ReLU /
            // residual applied in registers; the two halves of a wave write the two 16-byte halves of each 32-byte piece of a row.
            const int lane = opaque_lane();
            const int i31 = lane & 31, hi = lane >> 5;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int m0 = T.m0 + 32 * b;
                const int row = m0 + i31;
                const bool live = row < M;
                if (a.status) {
                    unsigned hm = 0u;
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned bb = __builtin_bit_cast(unsigned, acc[b][d][r]) << 1;
                            hm = hm > bb ? hm : bb;
                        }
                    const bool bad = live && hm >= (0x477fe000u << 1);
                    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(a.status, 2u);
                }
                const bool addb = a.bias != nullptr && (!NORM || row >= a.zrows);
                float* const yrow = a.y + (size_t)((LIN64_ABL & 64) ? (row & 255) : (live ? row : M - 1)) * a.ldy + a.col0 + T.col + 4 * hi;   // (64: timing only)
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    f32x4 old[4];
                    if (!(LIN64_ABL & 128) && a.beta != 0.0f) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) old[g] = *(const f32x4*)(yrow + 32 * c + 8 * g);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[b][c][4 * g], acc[b][c][4 * g + 1], acc[b][c][4 * g + 2], acc[b][c][4 * g + 3]};
                        if constexpr (!(LIN64_ABL & 128)) {
                        if (addb) v = v + *(const f32x4*)(bias_lds + T.col + 32 * c + 8 * g + 4 * hi);
                        if (a.act == 1) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
                        if (a.beta != 0.0f) v = v + old[g] * a.beta;
                        }
                        if (live) {
                            if (a.nt_store) lin_store_nt(yrow + 32 * c + 8 * g, v);
                            else *(f32x4*)(yrow + 32 * c + 8 * g) = v;
                        }
                    }
                    if constexpr (LMP) {
                        // landmark column sums: through the wave-private transposed tile, as lin_kernel (q / k columns only)
                        const int colL = a.col0 + T.col + 32 * c;
                        if (colL < a.lm_cols && m0 < M) {
                            float* pool = (float*)(smem + G::POOL_OFF + wave * G::POOL);
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int r = 0; r < 16; ++r) pool[i31 * 36 + mfma32_row(r, hi)] = acc[b][c][r];
                            __builtin_amdgcn_wave_barrier();
                            const int bnd = (m0 / a.lm_l + 1) * a.lm_l - m0;
                            const float bcol = a.bias ? bias_lds[T.col + 32 * c + i31] : 0.0f;
                            float sA = 0.0f, sB = 0.0f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int rr = 16 * hi + r;
                                float v = pool[rr * 36 + i31];
                                if (!NORM || m0 + rr >= a.zrows) v += bcol;
                                if (rr < bnd) sA += v; else sB += v;
                            }
                            const float oA = __shfl_xor(sA, 32), oB = __shfl_xor(sB, 32);
                            const float tA = hi ? oA + sA : sA + oA, tB = hi ? oB + sB : sB + oB;
                            if (!hi || bnd < 32)
                                a.lm_part[((size_t)(m0 >> 5) * 2 + hi) * a.lm_cols + colL + i31] = hi ? tB : tA;
                        }
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
        rab[0] = rabn[0]; rab[1] = rabn[1];
        ntile = dynamic ? (int)__builtin_amdgcn_readfirstlane(*nn_lds) : tile + nloc;
        T = TN;
    }
    ga_wait_vm<0>();
    if (dynamic && a.done) { __syncthreads(); count_done(); }
}
