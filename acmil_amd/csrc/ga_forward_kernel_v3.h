// ga_forward_kernel_v3.h -- third-generation fused GA forward (split-f16 arithmetic), gfx950: ONE 512-register wave per SIMD (round 5).
//
// Same mathematics, same packed weight stream (ga_pack.hip), same outputs and per-tile partials as ga_fwd2_kernel
// (ga_forward_kernel_v2.h; reference: architecture/network.py:49-57, architecture/transformer.py:259-267, :322-324).  What changes
// is the shape of the work per wave, so that the WIDE D_inner families of the reference's newer feature extractors
// (Step3_WSI_classification_ACMIL.py:78-87: CLIP-L 768 -> 384, UNI 1024 -> 512) get a fully fused kernel -- their `h` never
// touches HBM (the composed path wrote it once and read it twice: 307 MB of a 603 MB forward at UNI) -- and so that the 256-wide
// headline family can run the 64-patch wave tile of lin64_kernel (linear64_kernel.h) with GEMM2 and the epilogue behind it:
//
//   * a workgroup = 4 waves = ONE per SIMD (amdgpu_waves_per_eu(1, 1)): a wave owns the whole 512-entry register file.  It carries
//     PB blocks of 32 patches through the whole chain:  (ND, PB) = (16, 1) UNI, (12, 1) CLIP-L, (8, 2) the D_inner = 256 family.
//     GEMM1 accumulators: PB x ND tiles of 16 registers (256 at all three shapes; hipcc places them in the AGPR half);
//   * PB = 2: every weight fragment read from LDS feeds TWO MFMA column blocks in GEMM1 AND in GEMM2 (both blocks' relu(h), as f16
//     hi / lo fragments, are 256 registers -- exactly what the dead GEMM1 accumulators free), and the L2 -> LDS weight stream is
//     copied once per 256 patches;
//   * relu(h) feeds GEMM2 from registers as in v1 / v2 (the C/D layout of GEMM1 is a B-operand layout of GEMM2 once the K slots are
//     permuted; the permutation lives in the weight packing).  GEMM2 = 4 unit blocks x 4 steps, a step covers DD = ND / 4 feature
//     tiles (2 x DD fragment groups of 4 rows: hi, lo per tile) for both accumulator tiles (tanh / sigmoid branch) of every block;
//   * LDS-DMA ring as v2 (3 slots, prefetch distance 2, counted vmcnt, one raw s_barrier per step).  A wave copies RW = 2 ND / 4
//     consecutive fragment rows (up to 8) plus its own bag rows per step: the 13-bit immediate of global_load_lds reaches 4 rows
//     back, so rows 0 .. RW - 5 go through a second (M0, base) anchor;
//   * nothing co-resident hides a stall here: the LDS reads of a step sit under the deferred MFMA group of the previous step, the
//     f16 split of the bag values and the DMA issue sit in MFMA gaps (one piece per gap).
// Epilogue (gate, DPP softmax statistics, pooling on the matrix pipe through ds_read_b64_tr_b16, fixed-order combine) as v2, per
// 32-patch block; a tile's partial record is the combine of its 4 x PB blocks.
#pragma once
#include "ga_forward_kernel_v2.h"

#ifndef GA3_DMA_LATE
#define GA3_DMA_LATE 1
#endif
#ifndef GA3_DMA_LATE2
#define GA3_DMA_LATE2 1
#endif
// timing-only ablations (tools/build_v3_variant.sh; WRONG results): 1 no bag DMA, 2 no weight DMA, 4 no MFMAs in the two GEMMs,
// 8 no pooling / combine, 16 no step barrier, 32 no LDS fragment reads.  0 in every product build.
#ifndef GA3_ABL
#define GA3_ABL 0
#endif
#if GA3_ABL & 4
#define GA3_MFMA(A, B, C) (C)
#else
#define GA3_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#endif

// D += A B unless the wave-uniform word `cond` is zero: the branch lives inside the asm statement (see the GEMM1 loop).  The accumulator
// tile is read and written in place (exactly the same vDst as the MFMAs before and after it: hardware-interlocked, no software wait states).
#define GA3_MFMA_IF(cond, A, B, C)                                                                                     \
    asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n1:"              \
                 : "+a"(C) : "v"(A), "v"(B), "s"(cond) : "scc")

template <int ND, int PB, int KP, int XDT, bool POOLED = true>
struct Ga3Geom {
    static constexpr int WAVES = 4;
    static constexpr int XE = (XDT == ACMIL_DTYPE_F32) ? 4 : 2;
    static constexpr int WROWS = 2 * ND;                            // fragment rows per step (hi rows then lo rows)
    static constexpr int RW = WROWS / WAVES;                        // rows each wave copies per step: 4 / 6 / 8
    static_assert(WROWS % WAVES == 0 && RW >= 1 && RW <= 8, "two anchors reach 8 rows");
    static constexpr int AN_HI = RW;                                // anchor row of rows RW-4 .. RW-1 and of the bag pieces
    static constexpr int AN_LO = (RW > 4) ? RW - 4 : RW;            // anchor row of rows 0 .. RW-5
    static constexpr int DD = ND / 4;                               // feature tiles per GEMM2 step
    static constexpr int XGB = 32 * 16 * XE / 1024;                 // bag pieces (1 KiB) per 32-patch block and GEMM1 step
    static constexpr int XG = PB * XGB;
    static_assert(XG <= 4, "bag piece immediates 0..3072");
    static constexpr int XBLK = 32 * 16 * XE;                       // bytes of one block's rows inside a slot
    static constexpr int NVX = RW + XG;                             // LDS-DMA pieces per wave, GEMM1 step
    static constexpr int NVW = RW;                                  //                          GEMM2 step
    static_assert(NVX <= 12, "GA3_DMA_AT");
    static constexpr int REGION = NVX * 1024;                       // per-wave region of a slot: RW rows, then the bag rows
    static constexpr int SLOT = WAVES * REGION;
    static constexpr int NB = 3, PD = 2;
    static constexpr int ROWS = 32 * PB * WAVES;                    // patches per tile
    static constexpr int VW = PB * WAVES;                           // 32-patch blocks ("virtual waves") per tile
    static constexpr int Di = 32 * ND;
    static constexpr int RING = NB * SLOT;
    static constexpr int PTILE = 4608;                              // block-private transposition image (f16 planes) or [32][36] fp32
    static constexpr int COMB = POOLED ? KP * Di * 4 : 0;           // a block's combine record [KP][Di] (none in the score pass)
    static constexpr int PW = (PTILE > COMB) ? PTILE : COMB;        // per-block epilogue scratch
    static constexpr int TAB_BYTES = (2 + KP) * GA_DA * 4 + 32;     // bv[128], bu[128], Ww[KP][128], bw[8]
    static constexpr int PL_BYTES = WAVES * KP * 32 * 4;            // softmax numerators [wave][KP][32] (one block at a time)
    static constexpr int ML_BYTES = VW * 8 * 2 * 4 + 16;            // (max, sum) [block][8], then the drawn tile index
    static constexpr bool SCRATCH_IN_RING = (RING + VW * PW + TAB_BYTES + PL_BYTES + ML_BYTES > 160 * 1024);
    static_assert(!SCRATCH_IN_RING || REGION >= PB * PW, "free-slot scratch must hold a wave's records");
    static constexpr int SCR_OFF = RING;
    static constexpr int TAB_OFF = RING + (SCRATCH_IN_RING ? 0 : VW * PW);
    static constexpr int PL_OFF = TAB_OFF + TAB_BYTES;
    static constexpr int ML_OFF = PL_OFF + PL_BYTES;
    static constexpr int LDS = ML_OFF + ML_BYTES;
    static_assert(LDS <= 160 * 1024, "one workgroup per CU");
    static constexpr int frow(int r) { return (r / RW) * REGION + (r % RW) * 1024; }   // fragment row r inside a slot
};

template <int ND, int PB, int KP, int XDT, bool POOL, bool SAVEH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ga_fwd3_kernel(GaFwdArgs a) {
    static_assert(ND % 4 == 0, "D_inner must be a multiple of 128");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Ga3Geom<ND, PB, KP, XDT, POOL>;
    static_assert(POOL || !G::SCRATCH_IN_RING, "the score pass stores h while the ring is full: its transposition tile lives outside the ring");
    constexpr int WAVES = G::WAVES, NTHR = 256, VW = G::VW;
    constexpr bool XLO = (XDT == ACMIL_DTYPE_F32);   // fp16 and bf16 bags are exact in the hi part (ga_forward_kernel_v2.h)
    constexpr bool XCV = (XDT != ACMIL_DTYPE_F16);
    constexpr int Di = G::Di, PD = G::PD, NB = G::NB, DD = G::DD;
    static_assert(PD == 2 && NB == 3, "the wait counts below are written for a prefetch distance of 2 steps");
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
    using I8 = std::integral_constant<int, 8>; using I9 = std::integral_constant<int, 9>;
    using I10 = std::integral_constant<int, 10>; using I11 = std::integral_constant<int, 11>;

    const GaLayout& L = a.L;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto ga3_lane = [&]() { int l = tid & 63; asm volatile("" : "+v"(l)); return l; };
    const int D = L.D, K = L.K;
    const int ntiles = a.tile_start[a.nbags];
    const char* wstream = a.packed + L.g1_off;
    const int S1 = D / 16;                 // GEMM1 steps
    constexpr int S2 = 16;                 // GEMM2 steps: 4 unit blocks x 4

    // ---- tile bookkeeping (wave-uniform)
    struct TileInfo { int N, m0, rmax; const char* xrow0; float* A_out; };
    auto tile_info = [&](int t) {
        int bag = 0, top = a.nbags - 1;
        while (bag < top) {
            const int mid = (bag + top + 1) >> 1;
            if (t >= a.tile_start[mid]) bag = mid; else top = mid - 1;
        }
        TileInfo ti;
        ti.N = a.Ns[bag];
        ti.m0 = (t - a.tile_start[bag]) * G::ROWS + wave * 32 * PB;
        ti.A_out = a.A_outs[bag];
        const int m0c = ti.m0 < ti.N ? ti.m0 : ti.N - 1;          // rows past the bag re-read its last row; results discarded
        ti.xrow0 = (const char*)a.xs[bag] + (size_t)m0c * D * G::XE;
        ti.rmax = ti.N - 1 - m0c;
        return ti;
    };
    // lane offsets of the bag copy (relative to xrow0): piece q covers rows 16 q .. (fp32) / 32 q .. (16-bit) of the wave's 32 PB rows;
    // the 16-byte chunk a lane fetches is XOR-swizzled with its row so that the ds_read_b128 of the MFMA B layout are conflict-free
    auto tile_xoff = [&](const TileInfo& ti, int lane, unsigned (&xo)[G::XG]) {
#pragma unroll
        for (int q = 0; q < G::XG; ++q) {
            int r, piece;
            if constexpr (G::XE == 4) { r = 16 * q + (lane >> 2); piece = (lane & 3) ^ ((r >> 2) & 3); }
            else { r = 32 * q + (lane >> 1); piece = (lane & 1) ^ ((r >> 3) & 1); }
            r = r < ti.rmax ? r : ti.rmax;
            xo[q] = (unsigned)r * (unsigned)(D * G::XE) + piece * 16;
        }
    };

    const char* const wrow0 = wstream + (size_t)(wave * G::RW) * GA_FRAG_ROW;            // this wave's rows of step 0
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const unsigned rbase = lds_base + wave * G::REGION;                                   // this wave's region of slot 0

    // DMA piece m (compile time) of step u into ring slot `slot`: m < RW = weight row m of this wave; m >= RW = bag piece m - RW
    auto dma_piece = [&](auto mc, int u, int slot, unsigned woff, bool with_x, const char* xrow0, const unsigned (&xo)[G::XG]) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < G::RW) {
            constexpr int AN = (m < G::RW - 4) ? G::AN_LO : G::AN_HI;
            if constexpr (!(GA3_ABL & 2)) ga2_dma<(m - AN) * 1024>(woff, wrow0 + (size_t)u * G::WROWS * GA_FRAG_ROW + AN * 1024, rbase + slot * G::SLOT + AN * 1024);
        } else if constexpr (m < G::NVX) {
            constexpr int q = m - G::RW;
            if constexpr (!(GA3_ABL & 1)) if (with_x) ga2_dma<q * 1024>(xo[q], xrow0 + (size_t)u * 16 * G::XE - q * 1024, rbase + slot * G::SLOT + G::RW * 1024);
        }
    };
#define GA3_DMA_AT(d, ...)                                   \
    do {                                                     \
        if ((d) == 0) dma_piece(I0{}, __VA_ARGS__);          \
        if ((d) == 1) dma_piece(I1{}, __VA_ARGS__);          \
        if ((d) == 2) dma_piece(I2{}, __VA_ARGS__);          \
        if ((d) == 3) dma_piece(I3{}, __VA_ARGS__);          \
        if ((d) == 4) dma_piece(I4{}, __VA_ARGS__);          \
        if ((d) == 5) dma_piece(I5{}, __VA_ARGS__);          \
        if ((d) == 6) dma_piece(I6{}, __VA_ARGS__);          \
        if ((d) == 7) dma_piece(I7{}, __VA_ARGS__);          \
        if ((d) == 8) dma_piece(I8{}, __VA_ARGS__);          \
        if ((d) == 9) dma_piece(I9{}, __VA_ARGS__);          \
        if ((d) == 10) dma_piece(I10{}, __VA_ARGS__);        \
        if ((d) == 11) dma_piece(I11{}, __VA_ARGS__);        \
    } while (0)

    // epilogue vectors bv, bu, Ww, bw -> LDS once per workgroup (rows K..KP-1 of Ww zero); visible after the first step barrier
    {
        const float* src = (const float*)(a.packed + L.tab_off);
        float* dst = (float*)(smem + G::TAB_OFF);
        for (int e = tid; e < (2 + KP) * GA_DA; e += NTHR) dst[e] = (e < (2 + K) * GA_DA) ? src[e] : 0.0f;
        if (tid < 8) dst[(2 + KP) * GA_DA + tid] = ((const float*)(a.packed + L.bw_off))[tid];
    }

    // ---- tile drawing (as v2)
    unsigned* const nn_lds = (unsigned*)(smem + G::ML_OFF + VW * 8 * 2 * 4);
    unsigned draw_raw = 0;
    auto draw_issue = [&]() {
        unsigned long long keep;
        const unsigned zero = 0, one = 1;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                     : "=&v"(draw_raw), "=&s"(keep) : "v"(zero), "v"(one), "s"(a.tile_counter) : "memory");
    };
    auto draw_publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(draw_raw) :: "memory");
        const unsigned v = __builtin_amdgcn_readfirstlane(draw_raw) + gridDim.x;
        if ((tid & 63) == 0) *nn_lds = v;
    };
    const bool dynamic = a.tile_counter != nullptr;

    int tile = blockIdx.x;
    int ntile = tile + (int)gridDim.x;
    if (dynamic) {
        if (wave == 0) { draw_issue(); draw_publish(); }
        __syncthreads();
        ntile = (int)__builtin_amdgcn_readfirstlane(*nn_lds);
    }
    TileInfo T = tile_info(tile);

    // prologue: steps 0 and 1 of the first tile
    int islot = 0;
    {
        const int ln = ga3_lane();
        unsigned xo[G::XG];
        tile_xoff(T, ln, xo);
#pragma unroll
        for (int s = 0; s < PD; ++s) {
#pragma unroll
            for (int m = 0; m < G::NVX; ++m) GA3_DMA_AT(m, s, islot, (unsigned)ln * 16, true, T.xrow0, xo);
            islot = (islot + 1 == NB) ? 0 : islot + 1;
        }
    }
    int rslot = 0;

    auto step_sync = [&](bool next_has_x) {
        if (next_has_x) ga_wait_vm<G::NVX>(); else ga_wait_vm<G::NVW>();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        if constexpr (!(GA3_ABL & 16)) __builtin_amdgcn_s_barrier();
    };
    const float* tabf = (const float*)(smem + G::TAB_OFF);
    const float* bwp = tabf + (2 + KP) * GA_DA;

    // =============================================================================================== tile loop
    for (;;) {
        const bool has_next = ntile < ntiles;
        const TileInfo TN = tile_info(has_next ? ntile : tile);
        if (dynamic && has_next && wave == 0) draw_issue();
        __builtin_amdgcn_s_waitcnt(0xc07f);

        unsigned hmax = 0u;
        f16x8 hh[PB][ND][2], hl[PB][ND][2];
        {
            f32x16 acc1[PB][ND];
#pragma unroll
            for (int b = 0; b < PB; ++b)
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[b][d][r] = 0.0f;
            // ======================================================= GEMM1: h^T = W1 * x^T
            {
                const int ln = ga3_lane();
                const int i31 = ln & 31, hi = ln >> 5;
                const int lane16 = ln * 16;
                unsigned xoff[G::XG];
                tile_xoff(T, ln, xoff);
                const int xbase = wave * G::REGION + G::RW * 1024;
                const int xrd0 = xbase + ((G::XE == 4) ? (i31 * 64 + (((2 * hi) ^ ((i31 >> 2) & 3)) * 16)) : (i31 * 32 + ((hi ^ ((i31 >> 3) & 1)) * 16)));
                const int xrd1 = xbase + i31 * 64 + (((2 * hi + 1) ^ ((i31 >> 2) & 3)) * 16);   // fp32 only
                f32x4 xr0[PB], xr1[PB];
                u32x4 xrw[PB];
                auto read_x = [&](const char* slot) {
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        if constexpr (XDT == ACMIL_DTYPE_F32) { xr0[b] = *(const f32x4*)(slot + b * G::XBLK + xrd0); xr1[b] = *(const f32x4*)(slot + b * G::XBLK + xrd1); }
                        else xrw[b] = *(const u32x4*)(slot + b * G::XBLK + xrd0);
                    }
                };
                constexpr int NSP = XCV ? 4 * PB : 0;      // split / convert pieces per step: (block, pair) = (p >> 2, p & 3)
                u32x4 xhw[PB], xlw[PB];
                auto split_piece = [&](int p) {
                    const int b = p >> 2, j = p & 3;
                    float v0, v1;
                    if constexpr (XDT == ACMIL_DTYPE_F32) {
                        v0 = j < 2 ? xr0[b][2 * (j & 1)] : xr1[b][2 * (j & 1)];
                        v1 = j < 2 ? xr0[b][2 * (j & 1) + 1] : xr1[b][2 * (j & 1) + 1];
                    } else {
                        v0 = __builtin_bit_cast(float, xrw[b][j] << 16);
                        v1 = __builtin_bit_cast(float, xrw[b][j] & 0xffff0000u);
                    }
                    if constexpr (XLO) {
                        unsigned h, l;
                        ga2_split_pair(v0, v1, h, l);
                        xhw[b][j] = h; xlw[b][j] = l;
                    } else xhw[b][j] = ga_cvt_pair_f16(v0, v1);
                };
                f16x8 xh[PB], xl[PB], xhp[PB];
                auto split_done = [&]() {
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        if constexpr (XLO) { xh[b] = __builtin_bit_cast(f16x8, xhw[b]); xl[b] = __builtin_bit_cast(f16x8, xlw[b]); }
                        else if constexpr (XCV) xh[b] = __builtin_bit_cast(f16x8, xhw[b]);
                        else xh[b] = __builtin_bit_cast(f16x8, xrw[b]);
                    }
                };
                f16x8 WH[ND], WL[ND];
                auto read_hi = [&](const char* slot) {
                    if constexpr (GA3_ABL & 32) return;
#pragma unroll
                    for (int d = 0; d < ND; ++d) WH[d] = *(const f16x8*)(slot + G::frow(d) + lane16);
                };
                auto read_lo = [&](const char* slot) {
                    if constexpr (GA3_ABL & 32) return;
#pragma unroll
                    for (int d = 0; d < ND; ++d) WL[d] = *(const f16x8*)(slot + G::frow(ND + d) + lane16);
                };
                __builtin_amdgcn_s_setprio(2);
#pragma unroll
                for (int d = 0; d < ND; ++d) WL[d] = (f16x8)(_Float16)0.0f;
#pragma unroll
                for (int b = 0; b < PB; ++b) xhp[b] = (f16x8)(_Float16)0.0f;
                for (int s = 0; s < S1; ++s) {
                    step_sync(s + 1 < S1);
                    const char* slot = smem + rslot * G::SLOT;
                    read_x(slot);
                    read_hi(slot);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s > 0) {
                        // P3(s-1) = Wlo * xhi covers the reads above; the split of x(s) runs in its first MFMA gaps
#pragma unroll
                        for (int d = 0; d < ND; ++d)
#pragma unroll
                            for (int b = 0; b < PB; ++b) {
                                acc1[b][d] = GA3_MFMA(WL[d], xhp[b], acc1[b][d]);
                                if (d * PB + b < NSP) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    split_piece(d * PB + b);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                    } else {
#pragma unroll
                        for (int p = 0; p < NSP; ++p) split_piece(p);
                    }
                    split_done();
                    // fp32 bags of f16-exact values (stored fp16, up-cast by the loop): lo halves of exact zeros -- the W_hi x_lo MFMAs of this
                    // wave and step are branched over one by one INSIDE their asm statement (GA3_MFMA_IF): for the compiler the group stays
                    // straight-line code (a C++ branch around it cost the AGPR pinning of the 256 accumulators: 430 - 700 spilled registers)
                    unsigned lo_any = 1u;
                    if constexpr (XLO && GA2_LOSKIP) {
                        unsigned lo_or = 0u;
#pragma unroll
                        for (int b = 0; b < PB; ++b) lo_or |= xlw[b][0] | xlw[b][1] | xlw[b][2] | xlw[b][3];
                        lo_any = (unsigned)__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64((lo_or & 0x7fff7fffu) != 0u) != 0ull ? 1u : 0u));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    read_lo(slot);
                    // P1(s) = Whi * xhi, P2(s) = Whi * xlo (fp32 bags); the LDS-DMA pieces of step s + PD go out one per MFMA gap (bag rows
                    // only while that is still a GEMM1 step) -- in the LAST gaps of the step's MFMA sequence (GA3_DMA_LATE, default): the
                    // first gaps of P1 still have this step's 2 ND fragment reads in flight, where a piece costs 100 - 185 issue cycles
                    // instead of 25 - 60 (MI355X_MICROARCH, LDS-DMA issue cost), and this wave has no partner to cover them
                    const bool wx = s + PD < S1;
                    constexpr int GAPS = ND * PB * (XLO ? 2 : 1);
                    constexpr int G0 = GA3_DMA_LATE ? GAPS - G::NVX : 0;
                    static_assert(GAPS >= G::NVX, "a gap per piece");
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int b = 0; b < PB; ++b) {
                            acc1[b][d] = GA3_MFMA(WH[d], xh[b], acc1[b][d]);
                            __builtin_amdgcn_sched_barrier(0);
                            GA3_DMA_AT(d * PB + b - G0, s + PD, islot, (unsigned)lane16, wx, T.xrow0, xoff);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    if constexpr (XLO) {
#pragma unroll
                        for (int d = 0; d < ND; ++d)
#pragma unroll
                            for (int b = 0; b < PB; ++b) {
                                if constexpr (GA2_LOSKIP && !(GA3_ABL & 4)) GA3_MFMA_IF(lo_any, WH[d], xl[b], acc1[b][d]);
                                else acc1[b][d] = GA3_MFMA(WH[d], xl[b], acc1[b][d]);
                                __builtin_amdgcn_sched_barrier(0);
                                GA3_DMA_AT(ND * PB + d * PB + b - G0, s + PD, islot, (unsigned)lane16, wx, T.xrow0, xoff);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                    islot = (islot + 1 == NB) ? 0 : islot + 1;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < PB; ++b) xhp[b] = xh[b];
                    rslot = (rslot + 1 == NB) ? 0 : rslot + 1;
                }
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int b = 0; b < PB; ++b) acc1[b][d] = GA3_MFMA(WL[d], xhp[b], acc1[b][d]);
                __builtin_amdgcn_s_setprio(0);
            }
            // ======================================================= range guard, relu, f16 split of h -- tile by tile, so that a tile's
            // 16 accumulator registers die as its 16 fragment registers are born (256 + 256 would not fit beside anything else)
#pragma unroll
            for (int b = 0; b < PB; ++b)
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    f32x16 t = acc1[b][d];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const unsigned b0 = __builtin_bit_cast(unsigned, t[r]) << 1, b1 = __builtin_bit_cast(unsigned, t[r + 1]) << 1;
                        const unsigned m = b0 > b1 ? b0 : b1;
                        hmax = hmax > m ? hmax : m;
                    }
                    if constexpr (SAVEH) {
                        // score pass of the training step: relu(h) goes to HBM as fp32 rows, 32 features at a time through a block-private padded
                        // tile [32 patches][36] (written lane = patch / register = feature, read back as float4 = 4 consecutive features of one
                        // patch: 8 lanes store 128 contiguous bytes of an h row).  From the fp32 accumulators, before they are split.
                        float* pool = (float*)(smem + G::SCR_OFF + (wave * PB + b) * G::PW);
                        const int ln = ga3_lane();
                        const int i31 = ln & 31, hi = ln >> 5, rsub = ln >> 3, cq = ln & 7;
                        const int m0 = T.m0 + 32 * b;
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int r = 0; r < 16; ++r) pool[i31 * 36 + mfma32_row(r, hi)] = fmaxf(t[r], 0.0f);
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int prow = 8 * it + rsub;
                            const f32x4 hv = *(const f32x4*)(pool + prow * 36 + 4 * cq);
                            if (m0 + prow < T.N) *(f32x4*)(a.h_save + (size_t)(m0 + prow) * Di + 32 * d + 4 * cq) = hv;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        u32x4 hw, lw;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            unsigned h, l;
                            ga2_split_pair(fmaxf(t[8 * e + 2 * j], 0.0f), fmaxf(t[8 * e + 2 * j + 1], 0.0f), h, l);
                            hw[j] = h; lw[j] = l;
                        }
                        hh[b][d][e] = __builtin_bit_cast(f16x8, hw);
                        hl[b][d][e] = __builtin_bit_cast(f16x8, lw);
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        asm volatile("" : "+a"(hh[b][d][e]));
                        asm volatile("" : "+a"(hl[b][d][e]));
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);

        // ======================================================= GEMM2 (four unit blocks) + gate + scores
        float sc[PB][KP];
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int k = 0; k < KP; ++k) sc[b][k] = 0.0f;
        {
            // Fragment rows of a step: (dd * 2 + part) * 4 + t, t = e * 2 + al: 2 DD groups of 4 rows, consumed in the order
            // H(0), L(0), H(1), L(1), ...  (H = hi rows x (hh, hl), L = lo rows x hh).  Two 4-fragment buffers ping-pong: the next group
            // is requested while the current one computes, and L(DD-1) of a step is deferred across the next step's barrier, where it
            // covers the request of that step's H(0).
            f16x8 FA[4], FB[4];
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {
                const int ln = ga3_lane();
                const int hi = ln >> 5, lane16 = ln * 16;
                auto readgrp = [&](const char* slot, int grp, f16x8 (&F)[4]) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) F[t] = *(const f16x8*)(slot + G::frow(grp * 4 + t) + lane16);
                };
                f32x16 acc2[PB][2];
#pragma unroll
                for (int al = 0; al < 2; ++al)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        int boff = 32 * g + 8 * rq + 4 * hi;
                        asm volatile("" : "+v"(boff) : "v"(sc[0][0]));   // order after the previous block's epilogue
                        const f32x4 bb = *(const f32x4*)(tabf + al * GA_DA + boff);
#pragma unroll
                        for (int b = 0; b < PB; ++b) {
                            acc2[b][al][4 * rq + 0] = bb[0]; acc2[b][al][4 * rq + 1] = bb[1];
                            acc2[b][al][4 * rq + 2] = bb[2]; acc2[b][al][4 * rq + 3] = bb[3];
                        }
                    }
                __builtin_amdgcn_s_setprio(2);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int j = g * 4 + st;                       // GEMM2 step; tile step u = S1 + j
                    step_sync(j + 1 == S2);
                    const char* slot = smem + rslot * G::SLOT;
                    readgrp(slot, 0, FA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (st > 0) {   // L(DD-1) of the previous step of this block
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int b = 0; b < PB; ++b) acc2[b][t & 1] = GA3_MFMA(FB[t], hh[b][DD * st - 1][t >> 1], acc2[b][t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    readgrp(slot, 1, FB);
                    const bool nx = j + PD >= S2;
                    const int un = nx ? j + PD - S2 : S1 + j + PD;
                    unsigned xoffn[G::XG];
                    tile_xoff(TN, ln, xoffn);
                    // MFMA gaps of this step so far; one LDS-DMA piece per gap, in the step's LAST gaps (GA3_DMA_LATE2, as in GEMM1)
                    constexpr int GAPS2 = 8 * PB + (DD - 1) * 12 * PB;
                    constexpr int G02 = GA3_DMA_LATE2 ? GAPS2 - G::NVX : 0;
                    static_assert(GAPS2 >= G::NVX, "a gap per piece");
                    int gap = -G02;
#pragma unroll
                    for (int dd = 0; dd < DD; ++dd) {
                        const int d = DD * st + dd;
                        if (dd > 0) {
                            readgrp(slot, 2 * dd, FA);
                            // L(dd-1)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
#pragma unroll
                                for (int b = 0; b < PB; ++b) {
                                    acc2[b][t & 1] = GA3_MFMA(FB[t], hh[b][d - 1][t >> 1], acc2[b][t & 1]);
                                    __builtin_amdgcn_sched_barrier(0);
                                    GA3_DMA_AT(gap, un, islot, (unsigned)lane16, nx, TN.xrow0, xoffn);
                                    __builtin_amdgcn_sched_barrier(0);
                                    ++gap;
                                }
                            readgrp(slot, 2 * dd + 1, FB);
                        }
                        // H(dd): hi rows x (hh, hl)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int b = 0; b < PB; ++b) {
                                acc2[b][t & 1] = GA3_MFMA(FA[t], hh[b][d][t >> 1], acc2[b][t & 1]);
                                __builtin_amdgcn_sched_barrier(0);
                                GA3_DMA_AT(gap, un, islot, (unsigned)lane16, nx, TN.xrow0, xoffn);
                                __builtin_amdgcn_sched_barrier(0);
                                ++gap;
                            }
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int b = 0; b < PB; ++b) {
                                acc2[b][t & 1] = GA3_MFMA(FA[t], hl[b][d][t >> 1], acc2[b][t & 1]);
                                __builtin_amdgcn_sched_barrier(0);
                                GA3_DMA_AT(gap, un, islot, (unsigned)lane16, nx, TN.xrow0, xoffn);
                                __builtin_amdgcn_sched_barrier(0);
                                ++gap;
                            }
                    }
                    islot = (islot + 1 == NB) ? 0 : islot + 1;
                    if (st == 3) {   // the block's last step finishes its own L(DD-1): the gate needs the complete accumulators
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int b = 0; b < PB; ++b) acc2[b][t & 1] = GA3_MFMA(FB[t], hh[b][DD * st + DD - 1][t >> 1], acc2[b][t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    rslot = (rslot + 1 == NB) ? 0 : rslot + 1;
                }
                __builtin_amdgcn_s_setprio(0);
                // gate + partial scores for the 32 units of this block (this lane: 16 of them)
#pragma unroll
                for (int b = 0; b < PB; ++b)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        int ubase = 32 * g + 8 * rq + 4 * hi;
                        float gate[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) gate[q] = ga_tanh(acc2[b][0][4 * rq + q]) * ga_sigmoid(acc2[b][1][4 * rq + q]);
                        asm volatile("" : "+v"(ubase) : "v"(gate[3]));
#pragma unroll
                        for (int k = 0; k < KP; ++k) {
                            const f32x4 w = *(const f32x4*)(tabf + (2 + k) * GA_DA + ubase);
                            sc[b][k] = fmaf(gate[0], w[0], sc[b][k]); sc[b][k] = fmaf(gate[1], w[1], sc[b][k]);
                            sc[b][k] = fmaf(gate[2], w[2], sc[b][k]); sc[b][k] = fmaf(gate[3], w[3], sc[b][k]);
                        }
                    }
            }
        }

        // ======================================================= scores, softmax statistics, pooling -- per 32-patch block
        const int lane = ga3_lane();
        const int i31 = lane & 31, hi = lane >> 5;
        const int N = T.N;
        float* A_out = T.A_out;
        if (a.status) {
            bool hbad = false;
#pragma unroll
            for (int b = 0; b < PB; ++b) hbad = hbad || (T.m0 + 32 * b + i31 < N);
            hbad = hbad && hmax >= (0x477fe000u << 1);
            const unsigned bits = __builtin_amdgcn_ballot_w64(hbad) != 0 ? 2u : 0u;
            if (bits != 0 && lane == 0) atomicOr(a.self_reset ? a.status + 2 : a.status, bits);
        }
        // Scratch: the ring slot consumed last (the two others hold the next tile's first steps, in flight) or the separate region.
        // Every wave read ALL fragment rows of that slot in the last GEMM2 step: one barrier before anybody writes there.
        const int fslot = (rslot == 0) ? NB - 1 : rslot - 1;
        char* const scrw = G::SCRATCH_IN_RING ? smem + fslot * G::SLOT + wave * G::REGION : smem + G::SCR_OFF + wave * PB * G::PW;
        if constexpr (G::SCRATCH_IN_RING) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        float* pl = (float*)(smem + G::PL_OFF) + (size_t)wave * KP * 32;
        float* ml = (float*)(smem + G::ML_OFF);
        constexpr int NS = (KP + 1) / 2;
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int m0 = T.m0 + 32 * b;
            const int row = m0 + i31;
            const bool valid = row < N;
            char* const scr = scrw + b * G::PW;
            float smax[KP], lsum[KP], pe[NS];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                float fa = sc[b][2 * sl], fb = (2 * sl + 1 < KP) ? sc[b][2 * sl + 1] : 0.0f;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(fa), "+v"(fb));
                const int kk = 2 * sl + hi;
                float v = fa + fb;
                v += bwp[kk < KP ? kk : 0];
                if (A_out && valid && kk < K) A_out[(size_t)kk * N + row] = v;
                float m = valid ? v : -INFINITY;
                m = fmaxf(m, ga2_dpp<0x111, 0xf>(m, -INFINITY)); m = fmaxf(m, ga2_dpp<0x112, 0xf>(m, -INFINITY));
                m = fmaxf(m, ga2_dpp<0x114, 0xf>(m, -INFINITY)); m = fmaxf(m, ga2_dpp<0x118, 0xf>(m, -INFINITY));
                m = fmaxf(m, ga2_dpp<0x142, 0xa>(m, -INFINITY));
                const float mlo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 31));
                const float mhi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 63));
                const float p = valid ? __expf(v - (hi ? mhi : mlo)) : 0.0f;
                float l = p;
                l += ga2_dpp<0x111, 0xf>(l, 0.0f); l += ga2_dpp<0x112, 0xf>(l, 0.0f);
                l += ga2_dpp<0x114, 0xf>(l, 0.0f); l += ga2_dpp<0x118, 0xf>(l, 0.0f);
                l += ga2_dpp<0x142, 0xa>(l, 0.0f);
                smax[2 * sl] = mlo;
                lsum[2 * sl] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), 31));
                if (2 * sl + 1 < KP) {
                    smax[2 * sl + 1] = mhi;
                    lsum[2 * sl + 1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), 63));
                }
                pe[sl] = p;
            }
            if constexpr (POOL && !(GA3_ABL & 8)) {
                // attention-weighted sum on the matrix pipe (as v2): D[k][f] = sum_n P[k][n] h[n][f] per 32-feature tile, A = P, B = h^T through
                // the hardware transpose read; split arithmetic Ph*Hh + Pl*Hh + Ph*Hl
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const int kk = 2 * sl + hi;
                    if (kk < KP) pl[kk * 32 + i31] = pe[sl];
                }
                __builtin_amdgcn_wave_barrier();
                f16x8 PH[2], PL[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    f32x4 p0 = {0.0f, 0.0f, 0.0f, 0.0f}, p1 = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (i31 < KP) {
                        p0 = *(const f32x4*)(pl + i31 * 32 + 16 * st + 8 * hi);
                        p1 = *(const f32x4*)(pl + i31 * 32 + 16 * st + 8 * hi + 4);
                    }
                    u32x4 hw, lw;
                    unsigned h, l;
                    ga2_split_pair(p0[0], p0[1], h, l); hw[0] = h; lw[0] = l;
                    ga2_split_pair(p0[2], p0[3], h, l); hw[1] = h; lw[1] = l;
                    ga2_split_pair(p1[0], p1[1], h, l); hw[2] = h; lw[2] = l;
                    ga2_split_pair(p1[2], p1[3], h, l); hw[3] = h; lw[3] = l;
                    PH[st] = __builtin_bit_cast(f16x8, hw);
                    PL[st] = __builtin_bit_cast(f16x8, lw);
                }
                const int wr_off = hi * 576 + i31 * 16;
                const int q = lane & 3;
                const int rd_off = ((((lane >> 4) & 1) * 2 + (q & 1)) * 576) + 8 * (q >> 1) + 16 * (8 * hi + ((lane & 15) >> 2));
                typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
                typedef __attribute__((address_space(3))) h16x4* ltr_t;
                float pk[ND][4];
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16 accp;
#pragma unroll
                    for (int r = 0; r < 16; ++r) accp[r] = 0.0f;
                    *(f16x8*)(scr + wr_off) = hh[b][d][0];
                    *(f16x8*)(scr + wr_off + 2 * 576) = hh[b][d][1];
                    *(f16x8*)(scr + 2304 + wr_off) = hl[b][d][0];
                    *(f16x8*)(scr + 2304 + wr_off + 2 * 576) = hl[b][d][1];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const h16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + rd_off + 256 * st));
                        const h16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + rd_off + 256 * st + 64));
                        const h16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + 2304 + rd_off + 256 * st));
                        const h16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + 2304 + rd_off + 256 * st + 64));
                        const f16x8 Bh = __builtin_shufflevector(__builtin_bit_cast(f16x4, a0), __builtin_bit_cast(f16x4, a1), 0, 1, 2, 3, 4, 5, 6, 7);
                        const f16x8 Bl = __builtin_shufflevector(__builtin_bit_cast(f16x4, b0), __builtin_bit_cast(f16x4, b1), 0, 1, 2, 3, 4, 5, 6, 7);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(PH[st], Bh, accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(PL[st], Bh, accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(PH[st], Bl, accp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[d][r] = accp[r];
                    __builtin_amdgcn_wave_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
                // this block's combine record [KP][Di] overlays its (now dead) transposition image
                float* comb = (float*)scr;
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < KP; ++k) { ml[((wave * PB + b) * 8 + k) * 2 + 0] = smax[k]; ml[((wave * PB + b) * 8 + k) * 2 + 1] = lsum[k]; }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = 4 * hi + r;
                    if (kk < KP) {
#pragma unroll
                        for (int d = 0; d < ND; ++d) comb[kk * Di + 32 * d + i31] = pk[d][r];
                    }
                }
            }
        }
        if constexpr (POOL) {
            // =================================================== combine the tile's 4 PB blocks, publish its partial
            if (dynamic && has_next && wave == 0) draw_publish();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            constexpr int PS = 2 + Di;
            float* out = a.part + (size_t)tile * K * PS;
            const int wstride = G::SCRATCH_IN_RING ? G::REGION : PB * G::PW;
            const char* scr0 = G::SCRATCH_IN_RING ? smem + fslot * G::SLOT : smem + G::SCR_OFF;
            for (int k = 0; k < K; ++k) {
                float mw[VW], M = -INFINITY;
#pragma unroll
                for (int w = 0; w < VW; ++w) { mw[w] = ml[(w * 8 + k) * 2]; M = fmaxf(M, mw[w]); }
                float fw[VW];
#pragma unroll
                for (int w = 0; w < VW; ++w) fw[w] = (mw[w] == -INFINITY) ? 0.0f : __expf(mw[w] - M);
                for (int e = tid; e < Di; e += NTHR) {
                    float v = 0.0f;
#pragma unroll
                    for (int w = 0; w < VW; ++w)
                        v = fmaf(fw[w], ((const float*)(scr0 + (w / PB) * wstride + (w % PB) * G::PW))[k * Di + e], v);
                    out[k * PS + 2 + e] = v;
                }
                if (tid == 0) {
                    float lt = 0.0f;
#pragma unroll
                    for (int w = 0; w < VW; ++w) lt = fmaf(fw[w], ml[(w * 8 + k) * 2 + 1], lt);
                    out[k * PS + 0] = M; out[k * PS + 1] = lt;
                }
            }
        }
        if (!has_next) break;
        if constexpr (!POOL) {
            if (dynamic) {
                if (wave == 0) draw_publish();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
            }
        }
        tile = ntile;
        ntile = dynamic ? (int)__builtin_amdgcn_readfirstlane(*nn_lds) : tile + (int)gridDim.x;
        T = TN;
        // the scratch (possibly a ring slot) is released by the next step's barrier: every wave reaches it after its epilogue
    }
#undef GA3_DMA_AT
    ga_wait_vm<0>();
    if (dynamic && a.self_reset) {
        __syncthreads();
        if (tid == 0) {
            const unsigned done = atomicAdd(a.tile_counter + 2, 1u);
            if (done == gridDim.x - 1) {
                atomicExch(a.tile_counter + 2, 0u); atomicExch(a.tile_counter, 0u);
                atomicExch(a.status, atomicExch(a.status + 2, 0u));
            }
        }
    }
}

// persistent launch: one workgroup per CU (or one per tile when there are fewer tiles)
template <int ND, int PB, int KP, int XDT>
int ga_launch_fwd3(const GaFwdArgs& a, bool pool, hipStream_t st) {
    using GP = Ga3Geom<ND, PB, KP, XDT, true>;
    using GS = Ga3Geom<ND, PB, KP, XDT, false>;
    static int cus_of[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ACMIL_ERR_LAUNCH;
    if (cus_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ACMIL_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)ga_fwd3_kernel<ND, PB, KP, XDT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GP::LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)ga_fwd3_kernel<ND, PB, KP, XDT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GS::LDS) != hipSuccess)
            return ACMIL_ERR_LAUNCH;
        cus_of[dev] = prop.multiProcessorCount;
    }
    const int slots = cus_of[dev];
    const int tiles = a.tile_start[a.nbags];
    const dim3 grid(tiles < slots ? tiles : slots), block(256);
    void (*kern)(GaFwdArgs) = pool ? ga_fwd3_kernel<ND, PB, KP, XDT, true, false> : ga_fwd3_kernel<ND, PB, KP, XDT, false, true>;
    hipLaunchKernelGGL(kern, grid, block, pool ? GP::LDS : GS::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
