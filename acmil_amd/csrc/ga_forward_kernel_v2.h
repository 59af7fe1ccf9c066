// ga_forward_kernel_v2.h -- second-generation fused GA forward (split-f16 arithmetic), gfx950: PERSISTENT workgroups.
//
// Same mathematics, same packed weight stream, same outputs and same per-tile partials as ga_fwd_kernel
// (ga_forward_kernel.h; reference: architecture/network.py:49-57, architecture/transformer.py:259-267, :322-324).
// A cycle account of v1 (s_memtime; tools/probe_ga.py, GA2_PROF) showed where a 128-patch tile's ~156 k cycles per SIMD
// went with two 4-wave workgroups per CU: 74 k matrix pipe; ~35 k of workgroup turnover (launch, table staging, cold
// DMA pipeline, drained tail) per tile; 40 k of gate / softmax / pooling phases during which BOTH co-resident
// workgroups idle the matrix pipe, because equal workgroups started together stay in lock step ("convoy"); the rest in
// per-step stalls of the GEMM loops.  This kernel attacks all three:
//
//   * persistent workgroups: grid = 2 per CU; tiles are DRAWN from a global counter (the second workgroup of a CU runs
//     ~25 % slower than the first -- age-based issue arbitration -- so equal shares would leave the first idle at the
//     end); tables are staged once; the LDS-DMA
//     ring never drains -- the last two steps of a tile already prefetch the first steps of the NEXT tile (its weights
//     and its bag rows), and the epilogue works in the ring slot that is free at that moment;
//   * de-phasing: the second workgroup of every CU (HW_REG_LDS_ALLOC.base != 0) delays its start by about half a tile,
//     so one workgroup's gate / softmax / pooling phases run beside the other's GEMM steps;
//   * software pipelining ACROSS the step barrier.  A step's 3 MFMA groups are issued in the order
//         P1 = Whi * xhi,  P2 = Whi * xlo,  P3 = Wlo * xhi
//     and P3 of step s is deferred until AFTER the barrier of step s+1: the barrier is followed immediately by the
//     ds_reads of the next "hi" fragments and the next x tile, and the 8 deferred MFMAs (256 matrix-pipe cycles) cover
//     their LDS latency and the fp32 -> f16 hi/lo split of x.  The "lo" fragments are requested under P1;
//   * LDS-DMA without vector address arithmetic: global_load_lds_dwordx4 in its SGPR-base + 32-bit lane offset +
//     immediate form.  Every wave copies RW CONSECUTIVE fragment rows plus its own x tile into one contiguous per-wave
//     region of the slot, so one M0 value and immediates -4096..+1024 address all of a step's pieces.  Steps that need
//     no bag rows (GEMM2) issue only the weight rows; the counted vmcnt waits follow the schedule.
// Ring discipline (NB = 3 slots, prefetch distance 2): barrier b_s makes slot s readable (every wave waited for its own
// pieces) and slot s-1 writable (every wave drained lgkmcnt before arriving); step s+2 is issued right after b_s.
#pragma once
#include "ga_forward_kernel.h"

// Measurement hooks.  The product build compiles them to nothing: Ga2Probe below is empty and the MFMA / DMA macros are the plain
// instructions.  tools/build_variants.sh builds timing variants with -DGA2_TOOLS, which swaps in tools/ga2_probe.h (s_memtime cycle
// account per phase, ablation of the MFMAs or the DMA streams -- results are WRONG there by construction, never shipped).
// wave priority inside the GEMM step loops / outside (gate, softmax, pooling): the wave that feeds the matrix pipe wins the
// issue arbitration against a co-resident wave that is in a VALU / LDS phase
#ifndef GA2_PRIO_GEMM
#define GA2_PRIO_GEMM 2
#endif
#ifndef GA2_PRIO_REST
#define GA2_PRIO_REST 0
#endif
#ifdef GA2_TOOLS
#include "ga2_probe.h"
#else
#define GA2_MFMA1(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define GA2_MFMA2(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
struct Ga2Probe {
    static constexpr bool DMA_X = true, DMA_W = true;
    __device__ __forceinline__ void sync_begin() {}
    __device__ __forceinline__ void sync_loaded() {}
    __device__ __forceinline__ void sync_released() {}
    __device__ __forceinline__ void tile_begin() {}
    __device__ __forceinline__ void gemm1_end() {}
    __device__ __forceinline__ void gemm2_end() {}
    __device__ __forceinline__ void softmax_end() {}
    __device__ __forceinline__ void pool_end() {}
    __device__ __forceinline__ void tile_end(float*, int, int, int, int) {}
};
#endif

template <int ND, int KP, int XDT, int WV = 4, bool TRI = false>
struct Ga2Geom {
    static constexpr int WAVES = WV;                                // 4: one wave per SIMD, two workgroups per CU; 8: ONE 256-patch workgroup
    // TRI: THREE 4-wave workgroups per CU = three waves per SIMD (the D_inner = 128 family with 16-bit bags: 64 accumulator
    // registers and a 12 KiB ring slot leave room for it; the pooling image then holds the hi and the lo planes one after the other)
    static constexpr int WGS = TRI ? 3 : 8 / WV;                    //    per CU (the weight stream crosses L2 -> LDS once per 256 patches)
    static constexpr int XE = (XDT == ACMIL_DTYPE_F32) ? 4 : 2;     // bytes per bag element
    static constexpr int WROWS = 2 * ND;                            // fragment rows per step ("hi" rows then "lo" rows)
    static_assert(WROWS % WAVES == 0, "every wave copies the same number of consecutive fragment rows");
    static constexpr int RW = WROWS / WAVES;                        // fragment rows per wave per step
    static_assert(RW >= 1 && RW <= 4, "row immediates must fit -4096..-1024");
    static constexpr int DD = ND / 4;                               // h tiles consumed per GEMM2 step
    static constexpr int XG = 32 * 16 * XE / 1024;                  // x pieces (1 KiB) per wave per GEMM1 step
    static constexpr int NVX = RW + XG;                             // LDS-DMA instructions per wave, GEMM1 step
    static constexpr int NVW = RW;                                  //                                 GEMM2 step
    static constexpr int REGION = NVX * 1024;                       // per-wave region of a slot: RW rows, then the x tile
    static constexpr int SLOT = WAVES * REGION;
    static constexpr int NB = 3;                                    // ring slots
    static constexpr int PD = 2;                                    // prefetch distance in steps
    static constexpr int ROWS = 32 * WAVES;                         // patches per tile
    static_assert(ND % 2 == 0, "whole 64-column pairs");        // (the fused GA kernel further needs ND % 4 == 0: asserted there)
    static constexpr int Di = 32 * ND;
    static constexpr int RING = NB * SLOT;
    static constexpr int PTILE = TRI ? 2304 : 4608;                 // wave-private transposition image: 2 x 4 planes x 576 B (f16; TRI: 4 planes, hi then lo) or [32][36] fp32
    static constexpr int COMB = KP * Di * 4;                        // wave's combine record [KP][Di]
    static constexpr int PW = (PTILE > COMB) ? PTILE : COMB;        // per-wave epilogue scratch
    static constexpr int TAB_BYTES = (2 + KP) * GA_DA * 4 + 32;     // bv[128], bu[128], Ww[KP][128], bw[8]
    static constexpr int PL_BYTES = WAVES * KP * 32 * 4;            // softmax numerators [wave][KP][32]
    static constexpr int ML_BYTES = WAVES * 8 * 2 * 4 + 16;         // (max, sum) [wave][8], then the drawn tile index
    // the epilogue scratch lives in the ring slot that is free between two tiles when a separate region would cost the
    // second workgroup of the CU (80 KiB each)
    static constexpr bool SCRATCH_IN_RING = (WGS * (RING + WAVES * PW + TAB_BYTES + PL_BYTES + ML_BYTES) > 160 * 1024);
    static_assert(!SCRATCH_IN_RING || REGION >= PW, "free-slot scratch must hold a wave's pooling tile / combine record");
    static constexpr int SCR_OFF = RING;                            // separate scratch (when not in the ring)
    static constexpr int TAB_OFF = RING + (SCRATCH_IN_RING ? 0 : WAVES * PW);
    static constexpr int PL_OFF = TAB_OFF + TAB_BYTES;
    static constexpr int ML_OFF = PL_OFF + PL_BYTES;
    static constexpr int LDS = ML_OFF + ML_BYTES;
    static_assert(WGS * LDS <= 160 * 1024, "WGS workgroups per CU");
    static constexpr int frow(int r) { return (r / RW) * REGION + (r % RW) * 1024; }   // fragment row r inside a slot
};

// One 1-KiB LDS-DMA piece: lane l's 16 bytes at (sbase + voff + IMM) land at LDS byte (m0v + IMM + 16 l).
// The immediate moves BOTH addresses (that is what lets one M0 / one base serve a whole step).  M0 is
// compiler-reserved: saved and restored inside the statement.  s_nop 2: M0 write -> LDS-DMA needs 1 wait state;
// an SGPR base last written by a VALU (v_readfirstlane) needs 5 before a VMEM instruction reads it.
template <int IMM>
__device__ __forceinline__ void ga2_dma(unsigned voff, const char* sbase, unsigned m0v) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(m0v), "n"(IMM) : "memory");
}

__device__ __forceinline__ void ga2_split_pair(float x0, float x1, unsigned& hi_pk, unsigned& lo_pk) { ga_split_pair_f16(x0, x1, hi_pk, lo_pk); }

// DPP move (no LDS): lanes the control / row mask leaves without a source keep `idv`, the identity of the reduction
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ga2_dpp(float v, float idv) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, idv), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// three workgroups per CU: the pooled (eval) kernel of the D_inner = 128 family on 16-bit bags, 4-wave geometry
template <int ND, int XDT, bool POOL, int WV, bool PAIR>
struct Ga2Tri { static constexpr bool value = POOL && ND == 4 && XDT != ACMIL_DTYPE_F32 && WV == 4 && !PAIR; };

#ifndef GA2_LOSKIP
#define GA2_LOSKIP 1      // 0 (A/B builds, tools/build_variants.sh): always issue the W_hi x_lo products of an fp32 bag
#endif
template <int ND, int KP, int XDT, bool POOL, bool SAVEH, int WV = 4, bool PAIR = false>
__global__ __launch_bounds__(64 * WV, (Ga2Tri<ND, XDT, POOL, WV, PAIR>::value ? 3 : 2)) void ga_fwd2_kernel(GaFwdArgs a) {
    static_assert(ND % 4 == 0, "D_inner must be a multiple of 128");
    static_assert(!PAIR || ND == 8, "the wave-pair split is built for D_inner = 256 (two h tiles per GEMM2 step)");
    static_assert(!PAIR || WV == 4, "exchange buffers: 4 KiB of the free slot's per-wave region (5 - 6 KiB at 4 waves)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Ga2Geom<ND, KP, XDT, WV, Ga2Tri<ND, XDT, POOL, WV, PAIR>::value>;
    constexpr int WAVES = G::WAVES, NTHR = 64 * WAVES;
    // fp16 bags are exact in the hi part.  So are bf16 bags (round 4): 8 significant bits fit the 11 of f16 down to 2^-14, and below
    // that hi = RN_f16(x) is off by at most half an f16 subnormal step (2^-25), a remainder whose own f16 image rounds to ZERO
    // (ties to even) -- the lo plane of a bf16 bag was identically zero and its product a third of GEMM1's MFMAs spent on zeros.
    // (Out-of-range values become inf in hi either way and are caught by the range guard.)  Results are bit-identical.
    constexpr bool XLO = (XDT == ACMIL_DTYPE_F32);
    constexpr bool XCV = (XDT != ACMIL_DTYPE_F16);   // the operand needs a conversion to f16 at all
    constexpr int Di = G::Di, PD = G::PD, NB = G::NB;
    static_assert(PD == 2 && NB == 3, "the wait counts below are written for a prefetch distance of 2 steps");
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    static_assert(G::NVX <= 6, "at most 6 pieces per step");

    const GaLayout& L = a.L;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Lane-derived addresses are re-derived per phase from an OPAQUE copy of the lane id: hoisted out of the tile loop they
    // would stay live through every phase and push the GEMM2 loop (hh/hl 128 + fragments 64 + accumulators 32 registers)
    // over the 256-register budget (spills, and hipcc drains vmcnt -- i.e. the DMA ring -- at every scratch reload).
    auto ga2_lane = [&]() { int l = tid & 63; asm volatile("" : "+v"(l)); return l; };
    const int D = L.D, K = L.K;
    const int ntiles = a.tile_start[a.nbags];
    const char* wstream = a.packed + L.g1_off;
    const int S1 = D / 16;                 // GEMM1 steps
    constexpr int S2 = 16;                 // GEMM2 steps: 4 unit blocks x 4

    // ---- tile bookkeeping (wave-uniform, SGPRs)
    struct TileInfo { int N, m0, rmax; const char* xrow0; float* A_out; };
    auto tile_info = [&](int t) {
        int bag = 0, top = a.nbags - 1;          // largest bag with tile_start[bag] <= t (binary search: up to 64 bags per launch)
        while (bag < top) {
            const int mid = (bag + top + 1) >> 1;
            if (t >= a.tile_start[mid]) bag = mid; else top = mid - 1;
        }
        TileInfo ti;
        ti.N = a.Ns[bag];
        ti.m0 = (t - a.tile_start[bag]) * G::ROWS + wave * 32;
        ti.A_out = a.A_outs[bag];
        // x tile of this wave: rows m0c .. m0c+31 (rows past the bag re-read its last row; their results are discarded)
        const int m0c = ti.m0 < ti.N ? ti.m0 : ti.N - 1;
        ti.xrow0 = (const char*)a.xs[bag] + (size_t)m0c * D * G::XE;
        ti.rmax = ti.N - 1 - m0c;
        return ti;
    };
    // lane offsets of the x copy (relative to xrow0): piece q, lane l -> (row, 16-B piece) with the source-side XOR swizzle
    auto tile_xoff = [&](const TileInfo& ti, int lane, unsigned (&xo)[G::XG]) {
#pragma unroll
        for (int q = 0; q < G::XG; ++q) {
            int r, piece;
            if constexpr (G::XG == 2) { r = 16 * q + (lane >> 2); piece = (lane & 3) ^ ((r >> 2) & 3); }
            else { r = lane >> 1; piece = (lane & 1) ^ ((r >> 3) & 1); }
            r = r < ti.rmax ? r : ti.rmax;
            xo[q] = (unsigned)r * (unsigned)(D * G::XE) + piece * 16;
        }
    };

    const char* const wreg0 = wstream + (size_t)(wave * G::RW + G::RW) * GA_FRAG_ROW;   // wave-uniform, biased by RW rows
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const unsigned m0w = lds_base + wave * G::REGION + G::RW * 1024;                    // M0 of slot 0

    // DMA piece number m (0..NVX-1, compile time) of step u into ring slot `slot`: m < RW = weight rows of step u of the
    // stream (GEMM1 steps, then GEMM2 steps); m >= RW = bag rows of GEMM1 step u of the tile described by (xrow0, xo)
    auto dma_piece = [&](auto mc, int u, int slot, unsigned woff, bool with_x, const char* xrow0, const unsigned (&xo)[G::XG]) {
        constexpr int m = decltype(mc)::value;
        const unsigned m0v = m0w + slot * G::SLOT;
        if constexpr (m < G::RW) {
            if constexpr (Ga2Probe::DMA_W) ga2_dma<-(G::RW - m) * 1024>(woff, wreg0 + (size_t)u * G::WROWS * GA_FRAG_ROW, m0v);
        } else if constexpr (m < G::NVX) {
            constexpr int q = m - G::RW;
            if constexpr (Ga2Probe::DMA_X) { if (with_x) ga2_dma<q * 1024>(xo[q], xrow0 + (size_t)u * 16 * G::XE - q * 1024, m0v); }
        }
    };
    // piece d (a loop index the unroller resolves) -> dma_piece<d>
#define GA2_DMA_AT(d, ...)                                   \
    do {                                                     \
        if ((d) == 0) dma_piece(I0{}, __VA_ARGS__);          \
        if ((d) == 1) dma_piece(I1{}, __VA_ARGS__);          \
        if ((d) == 2) dma_piece(I2{}, __VA_ARGS__);          \
        if ((d) == 3) dma_piece(I3{}, __VA_ARGS__);          \
        if ((d) == 4) dma_piece(I4{}, __VA_ARGS__);          \
        if ((d) == 5) dma_piece(I5{}, __VA_ARGS__);          \
    } while (0)

    // epilogue vectors bv, bu, Ww, bw -> LDS once per workgroup (rows K..KP-1 of Ww zero); visible after the first step barrier
    {
        const float* src = (const float*)(a.packed + L.tab_off);
        float* dst = (float*)(smem + G::TAB_OFF);
        for (int e = tid; e < (2 + KP) * GA_DA; e += NTHR) dst[e] = (e < (2 + K) * GA_DA) ? src[e] : 0.0f;
        if (tid < 8) dst[(2 + KP) * GA_DA + tid] = ((const float*)(a.packed + L.bw_off))[tid];
    }

    // ---- tile drawing.  One lane of wave 0 adds 1 to the global counter (returning atomic, issued early, waited for late);
    // the result travels to the other waves through one LDS word and a barrier that is there anyway.
    unsigned* const nn_lds = (unsigned*)(smem + G::ML_OFF + WAVES * 8 * 2 * 4);
    unsigned draw_raw = 0;
    auto draw_issue = [&]() {      // wave 0 only
        unsigned long long keep;
        const unsigned zero = 0, one = 1;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                     : "=&v"(draw_raw), "=&s"(keep) : "v"(zero), "v"(one), "s"(a.tile_counter) : "memory");
    };
    auto draw_publish = [&]() {    // wave 0 only, at a point where the look-ahead DMAs were issued long ago: vmcnt(0) costs nothing
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(draw_raw) :: "memory");
        const unsigned v = __builtin_amdgcn_readfirstlane(draw_raw) + gridDim.x;
        if ((tid & 63) == 0) *nn_lds = v;
    };
    const bool dynamic = a.tile_counter != nullptr;

    int tile = blockIdx.x;
    int ntile = tile + (int)gridDim.x;       // static striding unless a counter is given
    if (dynamic) {
        if (wave == 0) { draw_issue(); draw_publish(); }
        __syncthreads();                      // nothing is in flight yet
        ntile = (int)__builtin_amdgcn_readfirstlane(*nn_lds);
    }
    TileInfo T = tile_info(tile);

    // De-phase the two workgroups of a CU: HW_REG_LDS_ALLOC[7:0] (LDS base) is non-zero for the second one.
    if (a.dephase > 0 && __builtin_amdgcn_s_getreg(6 | (0 << 6) | (7 << 11)) != 0)
        for (int i = 0; i < a.dephase; ++i) __builtin_amdgcn_s_sleep(127);

    // prologue: steps 0 and 1 of the first tile
    int islot = 0;   // ring slot the next issued step goes to
    {
        const int ln = ga2_lane();
        unsigned xo[G::XG];
        tile_xoff(T, ln, xo);
#pragma unroll
        for (int s = 0; s < PD; ++s) {
#pragma unroll
            for (int m = 0; m < G::NVX; ++m) GA2_DMA_AT(m, s, islot, (unsigned)ln * 16, true, T.xrow0, xo);
            islot = (islot + 1 == NB) ? 0 : islot + 1;
        }
    }
    int rslot = 0;   // ring slot of the step being consumed

    Ga2Probe probe;      // empty unless built with -DGA2_TOOLS (tools/ga2_probe.h)
    // Before reading slot s: this wave's pieces of step s have landed (at most the NEXT step's pieces stay in flight: they
    // were issued one step ago, step s's two steps ago), every read of the slot about to be recycled has returned, barrier.
    auto step_sync = [&](bool next_has_x) {
        probe.sync_begin();
        if (next_has_x) ga_wait_vm<G::NVX>(); else ga_wait_vm<G::NVW>();
        probe.sync_loaded();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        probe.sync_released();
    };
    const float* tabf = (const float*)(smem + G::TAB_OFF);
    const float* bwp = tabf + (2 + KP) * GA_DA;      // bw[8] (LDS copy: no global load inside the tile loop)

    // =============================================================================================== tile loop
    for (;;) {
        probe.tile_begin();
        // the tile after this one (its first steps are prefetched by this tile's last steps); none -> refetch this tile's rows
        const bool has_next = ntile < ntiles;
        const TileInfo TN = tile_info(has_next ? ntile : tile);
        if (dynamic && has_next && wave == 0) draw_issue();      // the tile after the next one
        // Drain the scalar-load counter with the BUILTIN (which hipcc's wait-count pass models): otherwise pending s_loads
        // make lgkmcnt "out of order" in the compiler's model and every first MFMA of a group waits lgkmcnt(0).
        __builtin_amdgcn_s_waitcnt(0xc07f);

        f32x16 acc1[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[d][r] = 0.0f;

        // range guard of the split-f16 arithmetic: largest |bag value| this lane converted (as a bit pattern, so that inf and
        // NaN rank above every finite value), largest feature it produced
        unsigned hmax = 0u;
        // wave-pair parity: own feature tiles are the actual tiles 2t + hp (PAIR), register order r <-> actual tile r ^ hp
        const int hp = PAIR ? (wave & 1) : 0;
        f16x8 hh[ND][2], hl[ND][2];
        if constexpr (PAIR) {
            // ==================================================== GEMM1 with D_inner split over a WAVE PAIR (ND = 8)
            // Waves (w, w^1) share their 64 patches: each computes HALF of the feature tiles -- actual tiles a = 2t + hp, hp = w & 1 --
            // for BOTH 32-patch groups (gamma = 0: its own group, gamma = 1: the partner's), so one weight fragment feeds two
            // MFMAs: per K step a wave reads 4 hi + 4 lo fragment rows (8 KiB instead of 16) and two bag tiles (its own and its
            // partner's DMA region, 4 KiB instead of 2) for the same 24 MFMAs and the same 128 accumulator registers.  Afterwards
            // the partners exchange the halves they computed for each other through the ring slot that is free at that moment
            // (8 rounds of 2 KiB per wave, double-buffered, one LDS barrier each), and every wave continues exactly as in the
            // unsplit kernel with all of h for its own 32 patches -- in the register order r <-> actual tile r ^ hp (own tiles in
            // the even slots, received tiles in the odd ones), which only shifts a few LDS / global ADDRESSES further down.
            const int ln = ga2_lane();
            const int i31 = ln & 31, hi = ln >> 5;
            const int lane16 = ln * 16;
            unsigned xoff[G::XG];
            tile_xoff(T, ln, xoff);
            const int xl0 = (G::XG == 2) ? (i31 * 64 + (((2 * hi) ^ ((i31 >> 2) & 3)) * 16)) : (i31 * 32 + ((hi ^ ((i31 >> 3) & 1)) * 16));
            const int xl1 = i31 * 64 + (((2 * hi + 1) ^ ((i31 >> 2) & 3)) * 16);   // fp32 only
            const int xb[2] = {wave * G::REGION + G::RW * 1024, (wave ^ 1) * G::REGION + G::RW * 1024};
            f32x4 xr0[2], xr1[2];
            u32x4 xrw[2];
            auto read_x = [&](const char* slot) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if constexpr (XDT == ACMIL_DTYPE_F32) { xr0[g] = *(const f32x4*)(slot + xb[g] + xl0); xr1[g] = *(const f32x4*)(slot + xb[g] + xl1); }
                    else xrw[g] = *(const u32x4*)(slot + xb[g] + xl0);
                }
            };
            constexpr int NSP = XCV ? 8 : 0;      // split / convert pieces per step: 2 groups x 4
            u32x4 xhw[2], xlw[2];
            auto split_piece = [&](int q) {
                const int g = q >> 2, j = q & 3;
                float v0, v1;
                if constexpr (XDT == ACMIL_DTYPE_F32) {
                    v0 = j < 2 ? xr0[g][2 * (j & 1)] : xr1[g][2 * (j & 1)];
                    v1 = j < 2 ? xr0[g][2 * (j & 1) + 1] : xr1[g][2 * (j & 1) + 1];
                } else {
                    v0 = __builtin_bit_cast(float, xrw[g][j] << 16);
                    v1 = __builtin_bit_cast(float, xrw[g][j] & 0xffff0000u);
                }
                if constexpr (XLO) {
                    unsigned h, l;
                    ga2_split_pair(v0, v1, h, l);
                    xhw[g][j] = h; xlw[g][j] = l;
                } else xhw[g][j] = ga_cvt_pair_f16(v0, v1);
            };
            f16x8 xh[2], xl[2], xhp[2];
            auto split_done = [&]() {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if constexpr (XLO) { xh[g] = __builtin_bit_cast(f16x8, xhw[g]); xl[g] = __builtin_bit_cast(f16x8, xlw[g]); }
                    else if constexpr (XCV) xh[g] = __builtin_bit_cast(f16x8, xhw[g]);
                    else xh[g] = __builtin_bit_cast(f16x8, xrw[g]);
                }
            };
            // fragment rows of this wave: hi row 2t + hp, lo row ND + 2t + hp (wave-uniform offsets)
            int fro[4], frl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rh = 2 * t + hp, rl = ND + 2 * t + hp;
                fro[t] = (rh / G::RW) * G::REGION + (rh % G::RW) * 1024;
                frl[t] = (rl / G::RW) * G::REGION + (rl % G::RW) * 1024;
            }
            f16x8 WH[4], WL[4];
            auto read_hi = [&](const char* slot) {
#pragma unroll
                for (int t = 0; t < 4; ++t) WH[t] = *(const f16x8*)(slot + fro[t] + lane16);
            };
            auto read_lo = [&](const char* slot) {
#pragma unroll
                for (int t = 0; t < 4; ++t) WL[t] = *(const f16x8*)(slot + frl[t] + lane16);
            };
            __builtin_amdgcn_s_setprio(GA2_PRIO_GEMM);
#pragma unroll
            for (int t = 0; t < 4; ++t) WL[t] = (f16x8)(_Float16)0.0f;
            xhp[0] = xhp[1] = (f16x8)(_Float16)0.0f;
            // accumulator of (tile t, group g): acc1[2 t + g]
            for (int s = 0; s < S1; ++s) {
                step_sync(s + 1 < S1);
                const char* slot = smem + rslot * G::SLOT;
                read_x(slot);
                read_hi(slot);
                __builtin_amdgcn_sched_barrier(0);
                if (s > 0) {
                    // P3(s-1) covers the reads above; the split of x(s) runs in its MFMA gaps
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        acc1[m] = GA2_MFMA1(WL[m >> 1], xhp[m & 1], acc1[m]);
                        if (m < NSP) {
                            __builtin_amdgcn_sched_barrier(0);
                            split_piece(m);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NSP; ++q) split_piece(q);
                }
                split_done();
                __builtin_amdgcn_sched_barrier(0);
                read_lo(slot);
                const bool wx = s + PD < S1;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    acc1[m] = GA2_MFMA1(WH[m >> 1], xh[m & 1], acc1[m]);
                    __builtin_amdgcn_sched_barrier(0);
                    GA2_DMA_AT(m, s + PD, islot, (unsigned)lane16, wx, T.xrow0, xoff);
                    __builtin_amdgcn_sched_barrier(0);
                }
                islot = (islot + 1 == NB) ? 0 : islot + 1;
                if constexpr (XLO) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) acc1[m] = GA2_MFMA1(WH[m >> 1], xl[m & 1], acc1[m]);
                }
                __builtin_amdgcn_sched_barrier(0);
                xhp[0] = xh[0]; xhp[1] = xh[1];
                rslot = (rslot + 1 == NB) ? 0 : rslot + 1;
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) acc1[m] = GA2_MFMA1(WL[m >> 1], xhp[m & 1], acc1[m]);
            __builtin_amdgcn_s_setprio(GA2_PRIO_REST);
            probe.gemm1_end();
            // ---- relu, range guard, f16 split; exchange of the partner-group halves
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const unsigned b0 = __builtin_bit_cast(unsigned, acc1[d][r]) << 1, b1 = __builtin_bit_cast(unsigned, acc1[d][r + 1]) << 1;
                    const unsigned m = b0 > b1 ? b0 : b1;
                    hmax = hmax > m ? hmax : m;
                }
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[d][r] = fmaxf(acc1[d][r], 0.0f);
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("" : "+v"(acc1[d]));
            // Exchange buffers (per wave, lane-linear 1-KiB pieces): the ring slot consumed last is free until GEMM2's second step
            // issues into it -- its per-wave region holds pieces 0..REGION/1024-1 -- and the bag part of the NEXT slot (it carries
            // GEMM2 step 0: weight rows only) two more.  fp32 bags: 8 pieces = two 4-piece buffers, ONE round per feature tile
            // (hi + lo halves together, 5 barriers per tile of 128 patches); 16-bit bags (5 + 1 pieces): two rounds per feature tile.
            const int xslot = (rslot == 0) ? NB - 1 : rslot - 1;
            const int xoffs_mine = xslot * G::SLOT + wave * G::REGION, xoffs_part = xslot * G::SLOT + (wave ^ 1) * G::REGION;
            const int nslot = rslot;      // slot of GEMM2 step 0
            const int xlane16 = ga2_lane() * 16;
            constexpr bool ONE_ROUND = (G::REGION / 1024 + G::XG >= 8);      // fp32 bags at 4 waves (6 + 2 pieces); else two rounds in 4 KiB
            auto piece = [&](int base_wave_off, int wv, int pc) -> int {      // byte offset of exchange piece pc (0..7) of wave wv
                if (pc < G::REGION / 1024) return base_wave_off + pc * 1024;
                return nslot * G::SLOT + wv * G::REGION + G::RW * 1024 + (pc - G::REGION / 1024) * 1024;
            };
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // own group: registers 2t of the final order
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    u32x4 hw, lw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned h, l;
                        ga2_split_pair(acc1[2 * t][8 * e + 2 * j], acc1[2 * t][8 * e + 2 * j + 1], h, l);
                        hw[j] = h; lw[j] = l;
                    }
                    hh[2 * t][e] = __builtin_bit_cast(f16x8, hw);
                    hl[2 * t][e] = __builtin_bit_cast(f16x8, lw);
                }
                // partner's group: split, hand over
                f16x8 sh[2], sl[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    u32x4 hw, lw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned h, l;
                        ga2_split_pair(acc1[2 * t + 1][8 * e + 2 * j], acc1[2 * t + 1][8 * e + 2 * j + 1], h, l);
                        hw[j] = h; lw[j] = l;
                    }
                    sh[e] = __builtin_bit_cast(f16x8, hw);
                    sl[e] = __builtin_bit_cast(f16x8, lw);
                }
                if constexpr (ONE_ROUND) {
                    const int b = 4 * (t & 1);
                    *(f16x8*)(smem + piece(xoffs_mine, wave, b + 0) + xlane16) = sh[0];
                    *(f16x8*)(smem + piece(xoffs_mine, wave, b + 1) + xlane16) = sh[1];
                    *(f16x8*)(smem + piece(xoffs_mine, wave, b + 2) + xlane16) = sl[0];
                    *(f16x8*)(smem + piece(xoffs_mine, wave, b + 3) + xlane16) = sl[1];
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    hh[2 * t + 1][0] = *(const f16x8*)(smem + piece(xoffs_part, wave ^ 1, b + 0) + xlane16);
                    hh[2 * t + 1][1] = *(const f16x8*)(smem + piece(xoffs_part, wave ^ 1, b + 1) + xlane16);
                    hl[2 * t + 1][0] = *(const f16x8*)(smem + piece(xoffs_part, wave ^ 1, b + 2) + xlane16);
                    hl[2 * t + 1][1] = *(const f16x8*)(smem + piece(xoffs_part, wave ^ 1, b + 3) + xlane16);
                } else {
                    *(f16x8*)(smem + xoffs_mine + xlane16) = sh[0];
                    *(f16x8*)(smem + xoffs_mine + 1024 + xlane16) = sh[1];
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    hh[2 * t + 1][0] = *(const f16x8*)(smem + xoffs_part + xlane16);
                    hh[2 * t + 1][1] = *(const f16x8*)(smem + xoffs_part + 1024 + xlane16);
                    *(f16x8*)(smem + xoffs_mine + 2048 + xlane16) = sl[0];
                    *(f16x8*)(smem + xoffs_mine + 3072 + xlane16) = sl[1];
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                    hl[2 * t + 1][0] = *(const f16x8*)(smem + xoffs_part + 2048 + xlane16);
                    hl[2 * t + 1][1] = *(const f16x8*)(smem + xoffs_part + 3072 + xlane16);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    asm volatile("" : "+v"(hh[d][e]));
                    asm volatile("" : "+v"(hl[d][e]));
                }
        } else {
            // ======================================================= GEMM1: h^T = W1 * x^T
            {
                const int ln = ga2_lane();
                const int i31 = ln & 31, hi = ln >> 5;
                const int lane16 = ln * 16;
                unsigned xoff[G::XG];
                tile_xoff(T, ln, xoff);
                // x operand of a step: raw read from the wave's own tile, then f16 hi/lo split
                const int xrd0 = wave * G::REGION + G::RW * 1024 +
                                 ((G::XG == 2) ? (i31 * 64 + (((2 * hi) ^ ((i31 >> 2) & 3)) * 16)) : (i31 * 32 + ((hi ^ ((i31 >> 3) & 1)) * 16)));
                const int xrd1 = wave * G::REGION + G::RW * 1024 + i31 * 64 + (((2 * hi + 1) ^ ((i31 >> 2) & 3)) * 16);   // fp32 only
                f32x4 xr0, xr1;      // raw fp32
                u32x4 xrw;           // raw 16-bit (8 elements)
                auto read_x = [&](const char* slot) {
                    if constexpr (XDT == ACMIL_DTYPE_F32) { xr0 = *(const f32x4*)(slot + xrd0); xr1 = *(const f32x4*)(slot + xrd1); }
                    else xrw = *(const u32x4*)(slot + xrd0);
                };
                // split piece j (< NSP) of the raw tile into word j of the hi / lo operands (two K slots per piece)
                constexpr int NSP = XCV ? 4 : 0;
                u32x4 xhw, xlw;
                auto split_piece = [&](int j) {
                    float v0, v1;
                    if constexpr (XDT == ACMIL_DTYPE_F32) {
                        v0 = j < 2 ? xr0[2 * (j & 1)] : xr1[2 * (j & 1)];
                        v1 = j < 2 ? xr0[2 * (j & 1) + 1] : xr1[2 * (j & 1) + 1];
                    } else {
                        v0 = __builtin_bit_cast(float, xrw[j] << 16);
                        v1 = __builtin_bit_cast(float, xrw[j] & 0xffff0000u);
                    }
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
                auto read_hi = [&](const char* slot) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) WH[d] = *(const f16x8*)(slot + G::frow(d) + lane16);
                };
                auto read_lo = [&](const char* slot) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) WL[d] = *(const f16x8*)(slot + G::frow(ND + d) + lane16);
                };
                __builtin_amdgcn_s_setprio(GA2_PRIO_GEMM);
                f16x8 xh, xl, xhp;
                // defined values at loop entry: otherwise LLVM treats the fragments of the PREVIOUS tile's last step as the
                // incoming values of the step loop and keeps 36 registers alive (spilled) through GEMM2 and the epilogue
#pragma unroll
                for (int d = 0; d < ND; ++d) WL[d] = (f16x8)(_Float16)0.0f;
                xhp = (f16x8)(_Float16)0.0f;
                for (int s = 0; s < S1; ++s) {
                    step_sync(s + 1 < S1);
                    const char* slot = smem + rslot * G::SLOT;
                    read_x(slot);      // first: the LDS returns data in order, so the split can start under the hi-fragment reads
                    read_hi(slot);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s > 0) {
                        // P3(s-1) covers the reads above; the split of x(s) runs in its first MFMA gaps
#pragma unroll
                        for (int d = 0; d < ND; ++d) {
                            acc1[d] = GA2_MFMA1(WL[d], xhp, acc1[d]);
                            if (d < NSP) {
                                __builtin_amdgcn_sched_barrier(0);
                                split_piece(d);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    } else {   // first step of the tile: nothing deferred yet
#pragma unroll
                        for (int j = 0; j < NSP; ++j) split_piece(j);
                    }
                    split_done(xh, xl);
                    // fp32 bags whose values are f16-exact -- what real bags are: stored fp16 (Step2_feature_extract.py:165), up-cast by the
                    // loop (Step3_WSI_classification_ACMIL.py:193) -- have lo halves of exact zeros: the wave looks at its 32 x 16 lo
                    // values of this step (3 VALU ORs + one compare into a wave-uniform mask) and skips the W_hi x_lo group when all
                    // are zero.  The skipped products are +-0: results are the same numbers (tests: equal to the three-product launch).
                    bool lo_any = true;
                    if constexpr (XLO && GA2_LOSKIP) {
                        const unsigned lo_or = (xlw[0] | xlw[1] | xlw[2] | xlw[3]) & 0x7fff7fffu;
                        lo_any = __builtin_amdgcn_ballot_w64(lo_or != 0u) != 0ull;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    read_lo(slot);
                    // P1(s), one LDS-DMA piece of step s+PD per MFMA gap (bag rows only while that step is still a GEMM1 step)
                    const bool wx = s + PD < S1;
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        acc1[d] = GA2_MFMA1(WH[d], xh, acc1[d]);
                        __builtin_amdgcn_sched_barrier(0);
                        GA2_DMA_AT(d, s + PD, islot, (unsigned)lane16, wx, T.xrow0, xoff);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    islot = (islot + 1 == NB) ? 0 : islot + 1;
                    if constexpr (XLO) {
                        if (lo_any) {
#pragma unroll
                            for (int d = 0; d < ND; ++d) acc1[d] = GA2_MFMA1(WH[d], xl, acc1[d]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    xhp = xh;
                    rslot = (rslot + 1 == NB) ? 0 : rslot + 1;
                }
                // P3 of the last GEMM1 step
#pragma unroll
                for (int d = 0; d < ND; ++d) acc1[d] = GA2_MFMA1(WL[d], xhp, acc1[d]);
                __builtin_amdgcn_s_setprio(GA2_PRIO_REST);
            }

            probe.gemm1_end();
            // ======================================================= relu + f16 split of h
            // acc1[d][r] holds h[patch = lane&31][feature = 32d + mfma32_row(r, hi)].  The empty asm statements keep LLVM from
            // sinking the relu / split into the GEMM2 steps (old and new values live together -> hundreds of spills).
            // Range guard, once per tile and OUTSIDE the GEMM loop: the largest pre-activation as a bit pattern (orders finite < inf
            // < NaN).  A bag value outside the f16 range converts to inf in its hi half and turns every feature of its patch into
            // inf / NaN, so this one test covers the bag values too (checking them inside the loop cost ~9 % of the kernel).
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {     // sign shifted out: magnitudes order as unsigned integers, inf / NaN of either sign on top
                    const unsigned b0 = __builtin_bit_cast(unsigned, acc1[d][r]) << 1, b1 = __builtin_bit_cast(unsigned, acc1[d][r + 1]) << 1;
                    const unsigned m = b0 > b1 ? b0 : b1;
                    hmax = hmax > m ? hmax : m;
                }
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[d][r] = fmaxf(acc1[d][r], 0.0f);
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("" : "+v"(acc1[d]));
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    u32x4 hw, lw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned h, l;
                        ga2_split_pair(acc1[d][8 * e + 2 * j], acc1[d][8 * e + 2 * j + 1], h, l);
                        hw[j] = h; lw[j] = l;
                    }
                    hh[d][e] = __builtin_bit_cast(f16x8, hw);
                    hl[d][e] = __builtin_bit_cast(f16x8, lw);
                }
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    asm volatile("" : "+v"(hh[d][e]));
                    asm volatile("" : "+v"(hl[d][e]));
                }

        }
        // ======================================================= GEMM2 (four unit blocks) + gate + scores
        float sc[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) sc[k] = 0.0f;
        {
            // Fragment rows of a step: (dd*2 + part)*4 + t, t = e*2 + al: four groups of 4 rows, consumed in the order
            // G0 = hi(dd 0), G1 = lo(dd 0), G2 = hi(dd 1), G3 = lo(dd 1).  Two 4-fragment buffers ping-pong (hh/hl already hold
            // 128 registers here): group i+1 is requested while group i computes, and G3 of a step is deferred across the next
            // step's barrier, where it covers the request of that step's G0.
            // (Di = 128: one h tile per step, groups G0 / G1 only)
            static_assert(G::DD == 2 || G::DD == 1, "GEMM2 step = one or two h tiles");
            f16x8 FA[4], FB[4];
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {
                const int ln = ga2_lane();
                const int hi = ln >> 5, lane16 = ln * 16;
                // (PAIR: register tile r holds the actual tile r ^ hp, i.e. the two h tiles of a step are swapped for odd waves)
                int goff[4][4];
#pragma unroll
                for (int grp = 0; grp < 4; ++grp)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int row = (PAIR ? (grp ^ (2 * hp)) : grp) * 4 + t;
                        goff[grp][t] = (row / G::RW) * G::REGION + (row % G::RW) * 1024;
                    }
                auto readgrp = [&](const char* slot, int grp, f16x8 (&F)[4]) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) F[t] = *(const f16x8*)(slot + goff[grp][t] + lane16);
                };
                // unit block g = units 32g..32g+31: accumulator tile al = 0 tanh branch, 1 sigmoid branch; init = bias
                f32x16 acc2[2];
#pragma unroll
                for (int al = 0; al < 2; ++al)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        int boff = 32 * g + 8 * rq + 4 * hi;
                        asm volatile("" : "+v"(boff) : "v"(sc[0]));   // order after the previous block's epilogue
                        const f32x4 b = *(const f32x4*)(tabf + al * GA_DA + boff);
                        acc2[al][4 * rq + 0] = b[0]; acc2[al][4 * rq + 1] = b[1];
                        acc2[al][4 * rq + 2] = b[2]; acc2[al][4 * rq + 3] = b[3];
                    }
                __builtin_amdgcn_s_setprio(GA2_PRIO_GEMM);
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int j = g * 4 + st;                       // GEMM2 step; tile step u = S1 + j
                    const int d0 = G::DD * st, d1 = G::DD * st + G::DD - 1;   // h tiles of this step (d1 == d0 when DD == 1)
                    // the step after this one needs bag rows only when it is step 0 of the next tile
                    step_sync(j + 1 == S2);
                    const char* slot = smem + rslot * G::SLOT;
                    readgrp(slot, 0, FA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (st > 0) {   // last group (lo x hh) of the previous step of this block
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t & 1] = GA2_MFMA2(FB[t], hh[d0 - 1][t >> 1], acc2[t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    readgrp(slot, 1, FB);
                    // step j+PD of this tile, or step j+PD-S2 of the NEXT tile (weights restart at the stream head, bag rows of TN)
                    const bool nx = j + PD >= S2;
                    const int un = nx ? j + PD - S2 : S1 + j + PD;
                    unsigned xoffn[G::XG];
                    tile_xoff(TN, ln, xoffn);     // used by 2 steps per tile: recomputed there instead of carried through GEMM2
                    // G0: hi(dd 0) x (hh, hl), one LDS-DMA piece per MFMA gap
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc2[t & 1] = GA2_MFMA2(FA[t], hh[d0][t >> 1], acc2[t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        GA2_DMA_AT(t, un, islot, (unsigned)lane16, nx, TN.xrow0, xoffn);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc2[t & 1] = GA2_MFMA2(FA[t], hl[d0][t >> 1], acc2[t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        GA2_DMA_AT(4 + t, un, islot, (unsigned)lane16, nx, TN.xrow0, xoffn);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    islot = (islot + 1 == NB) ? 0 : islot + 1;
                    if constexpr (G::DD == 2) {
                        readgrp(slot, 2, FA);
                        // G1: lo(dd 0) x hh
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t & 1] = GA2_MFMA2(FB[t], hh[d0][t >> 1], acc2[t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        readgrp(slot, 3, FB);
                        // G2: hi(dd 1) x (hh, hl)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t & 1] = GA2_MFMA2(FA[t], hh[d1][t >> 1], acc2[t & 1]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t & 1] = GA2_MFMA2(FA[t], hl[d1][t >> 1], acc2[t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (st == 3) {   // the block's last step finishes its own G3: the gate needs the complete accumulators
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t & 1] = GA2_MFMA2(FB[t], hh[d1][t >> 1], acc2[t & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    rslot = (rslot + 1 == NB) ? 0 : rslot + 1;
                }
                __builtin_amdgcn_s_setprio(GA2_PRIO_REST);
                // gate + partial scores for the 32 units of this block (this lane: 16 of them)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    int ubase = 32 * g + 8 * rq + 4 * hi;
                    float gate[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) gate[q] = ga_tanh(acc2[0][4 * rq + q]) * ga_sigmoid(acc2[1][4 * rq + q]);
                    // tie the table address to the gate value: keeps each read next to its use (no mass hoisting + spill)
                    asm volatile("" : "+v"(ubase) : "v"(gate[3]));
#pragma unroll
                    for (int k = 0; k < KP; ++k) {
                        const f32x4 w = *(const f32x4*)(tabf + (2 + k) * GA_DA + ubase);
                        sc[k] = fmaf(gate[0], w[0], sc[k]); sc[k] = fmaf(gate[1], w[1], sc[k]);
                        sc[k] = fmaf(gate[2], w[2], sc[k]); sc[k] = fmaf(gate[3], w[3], sc[k]);
                    }
                }
            }
        }
        probe.gemm2_end();

        // ======================================================= scores, wave-level softmax statistics
        // Everything cross-lane here is DPP / permlane (VALU): no ds_bpermute round trips, no LDS traffic beside the GEMM
        // of the co-resident workgroup.  The two lane halves (64 attention units each) are folded pairwise with
        // v_permlane32_swap: afterwards "slot" s holds branch 2s in lanes 0-31 and branch 2s+1 in lanes 32-63.
        const int lane = ga2_lane();
        const int i31 = lane & 31, hi = lane >> 5;
        const int N = T.N, m0 = T.m0;
        float* A_out = T.A_out;
        const int row = m0 + i31;
        const bool valid = row < N;
        if (a.status) {
            // 65504 = largest finite f16 (the features are split into f16 halves for the second GEMM); inf / NaN rank above it
            const bool hbad = valid && hmax >= (0x477fe000u << 1) /* |v| >= 65504.0f, inf, NaN */;
            const unsigned bits = __builtin_amdgcn_ballot_w64(hbad) != 0 ? 2u : 0u;
            if (bits != 0 && lane == 0) atomicOr(a.self_reset ? a.status + 2 : a.status, bits);  // accumulator word (ballots outside the one-lane branch: all lanes vote)
        }
        constexpr int NS = (KP + 1) / 2;
        float smax[KP], lsum[KP], pe[NS];
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            // (inline asm: hipcc 7.2 folds sw[0] + sw[1] of __builtin_amdgcn_permlane32_swap into 2 * sw[0]; the s_nops are the
            //  VALU-write -> permlane-read wait states the compiler would have inserted)
            float fa = sc[2 * sl], fb = (2 * sl + 1 < KP) ? sc[2 * sl + 1] : 0.0f;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(fa), "+v"(fb));   // [a.lo | b.lo], [a.hi | b.hi]
            const int kk = 2 * sl + hi;                                                  // this lane's branch in slot sl
            float v = fa + fb;
            v += bwp[kk < KP ? kk : 0];
            if (A_out && valid && kk < K) A_out[(size_t)kk * N + row] = v;
            // inclusive max-scan inside the 16-lane rows, then row 15 -> next row: lane 31 / 63 hold the maximum of their half
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
        probe.softmax_end();

        // ======================================================= attention-weighted sum  sum_n p[k][n] h[n][:]
        // Scratch: the ring slot consumed last (the two others hold the next tile's first steps, in flight) or the separate
        // region.  No vmcnt / __syncthreads here: the DMA ring stays in flight.
        const int fslot = (rslot == 0) ? NB - 1 : rslot - 1;
        char* const scr = G::SCRATCH_IN_RING ? smem + fslot * G::SLOT + wave * G::REGION : smem + G::SCR_OFF + wave * G::PW;
        // Every wave read ALL fragment rows of that slot in the last GEMM2 step, so nobody may write its scratch there before the slowest
        // wave's reads have returned.  The gate / softmax arithmetic in between (thousands of cycles) kept that true in every run so far,
        // but it is a property of timing, not of the code: lin_kernel, which goes from its last step straight to the same kind of scratch,
        // was caught with one wave a K step behind (linear_kernel.h, round 4).  One barrier per tile; the waves re-align here instead of
        // at the next tile's first step barrier.  (-DGA2_NO_SCRATCH_BARRIER: A/B build without it.)
#ifndef GA2_NO_SCRATCH_BARRIER
        if constexpr (G::SCRATCH_IN_RING) {
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
        }
#endif
        float* pl = (float*)(smem + G::PL_OFF) + (size_t)wave * KP * 32;   // softmax numerators [KP][32 patches]
        if constexpr (POOL) {
            // On the matrix pipe: D[k][f] = sum_n P[k][n] h[n][f] per 32-feature tile d, A = P (rows = branches, K = the wave's
            // 32 patches), B = h^T.  h sits in registers with lane = patch; the B operand wants lane = feature.  The f16
            // fragments hh / hl are written as they are (ds_write_b128, 4 per tile) and read back through the hardware
            // transpose ds_read_b64_tr_b16 (8 per tile): no fp32 transposition, no conversions, no VALU FMAs.  Split
            // arithmetic as in the GEMMs: Ph*Hh + Pl*Hh + Ph*Hl, fp32 accumulate.
            // LDS image of one tile: plane (e, hi) = the 32 lanes' 16-byte words, planes 576 B apart (64 B pad: conflict-free
            // tr reads); hh planes at 0, hl planes at 2304.
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const int kk = 2 * sl + hi;
                if (kk < KP) pl[kk * 32 + i31] = pe[sl];
            }
            __builtin_amdgcn_wave_barrier();
            // A fragments: lane (row k = lane&31, kg = lane>>5) holds P[k][16s + 8kg + j], j < 8, split into f16 hi / lo
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
            // write side: lane (patch i31, hi) word e -> plane (e*2 + hi); read side: supplier lane (j = (lane&15)>>2, q = lane&3)
            // of 16-lane group (G = (lane>>4)&1, kg = lane>>5) addresses patch 8kg + j, features 16G + 4q .. 4q+3
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
                if constexpr (G::PTILE >= 4608) {
                    *(f16x8*)(scr + wr_off) = hh[d][0];
                    *(f16x8*)(scr + wr_off + 2 * 576) = hh[d][1];
                    *(f16x8*)(scr + 2304 + wr_off) = hl[d][0];
                    *(f16x8*)(scr + 2304 + wr_off + 2 * 576) = hl[d][1];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        // K step st = patches 16st .. 16st+15; two transposed reads (4 patches each) per fragment
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
                } else {
                    // one 2304-byte image: the hi planes, then (after they have been read) the lo planes; same products, same order
                    // of accumulation per K step is NOT kept (hi-plane products of both K steps first) -- this variant has its own
                    // tolerance-level identity with the two-image form, not a bitwise one
                    *(f16x8*)(scr + wr_off) = hh[d][0];
                    *(f16x8*)(scr + wr_off + 2 * 576) = hh[d][1];
                    __builtin_amdgcn_wave_barrier();
                    f16x8 Bh[2];
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const h16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + rd_off + 256 * st));
                        const h16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + rd_off + 256 * st + 64));
                        Bh[st] = __builtin_shufflevector(__builtin_bit_cast(f16x4, a0), __builtin_bit_cast(f16x4, a1), 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);          // the hi planes are in registers
                    __builtin_amdgcn_wave_barrier();
                    *(f16x8*)(scr + wr_off) = hl[d][0];
                    *(f16x8*)(scr + wr_off + 2 * 576) = hl[d][1];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        const h16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + rd_off + 256 * st));
                        const h16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((ltr_t)(scr + rd_off + 256 * st + 64));
                        const f16x8 Bl = __builtin_shufflevector(__builtin_bit_cast(f16x4, b0), __builtin_bit_cast(f16x4, b1), 0, 1, 2, 3, 4, 5, 6, 7);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(PH[st], Bh[st], accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(PL[st], Bh[st], accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(PH[st], Bl, accp, 0, 0, 0);
                    }
                }
                // D rows: register r of lane half hi is branch (r&3) + 8(r>>2) + 4hi -> registers 0..3 = branches 4hi .. 4hi+3
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[d][r] = accp[r];
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            probe.pool_end();
            // =================================================== combine the 4 waves, publish the tile's partial
            float* comb = (float*)scr;   // [KP][Di], overlays this wave's (now dead) transposition image
            float* ml = (float*)(smem + G::ML_OFF);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < KP; ++k) { ml[(wave * 8 + k) * 2 + 0] = smax[k]; ml[(wave * 8 + k) * 2 + 1] = lsum[k]; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 4 * hi + r;
                if (kk < KP) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) comb[kk * Di + 32 * (d ^ hp) + i31] = pk[d][r];
                }
            }
            if (dynamic && has_next && wave == 0) draw_publish();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            constexpr int PS = 2 + Di;
            float* out = a.part + (size_t)tile * K * PS;
            const int scr_stride = G::SCRATCH_IN_RING ? G::REGION : G::PW;
            const char* scr0 = G::SCRATCH_IN_RING ? smem + fslot * G::SLOT : smem + G::SCR_OFF;
            for (int k = 0; k < K; ++k) {
                float mw[WAVES], M = -INFINITY;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) { mw[w] = ml[(w * 8 + k) * 2]; M = fmaxf(M, mw[w]); }
                float fw[WAVES];
#pragma unroll
                for (int w = 0; w < WAVES; ++w) fw[w] = (mw[w] == -INFINITY) ? 0.0f : __expf(mw[w] - M);
                for (int e = tid; e < Di; e += NTHR) {
                    float v = 0.0f;
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) v = fmaf(fw[w], ((const float*)(scr0 + w * scr_stride))[k * Di + e], v);
                    out[k * PS + 2 + e] = v;
                }
                if (tid == 0) {
                    float lt = 0.0f;
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) lt = fmaf(fw[w], ml[(w * 8 + k) * 2 + 1], lt);
                    out[k * PS + 0] = M; out[k * PS + 1] = lt;
                }
            }
        } else {
            // score pass of the training step (SAVEH): h goes to HBM as fp32 rows.  Through a wave-private padded tile [32 patches][36], 32
            // features at a time: written lane = patch / register = feature, read back as float4 = 4 consecutive features of one patch,
            // so 8 lanes store 128 contiguous bytes of an h row with one 16-byte store each (4 per lane and tile, not 16 scalars).
            float* pool = (float*)scr;
            const int rsub = lane >> 3, cq = lane & 7;
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hv = (float)hh[c][r >> 3][r & 7] + (float)hl[c][r >> 3][r & 7];
                    pool[i31 * 36 + mfma32_row(r, hi)] = hv;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SAVEH) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int prow = 8 * it + rsub;
                        const f32x4 hv = *(const f32x4*)(pool + prow * 36 + 4 * cq);
                        if (m0 + prow < N) *(f32x4*)(a.h_save + (size_t)(m0 + prow) * Di + 32 * (c ^ hp) + 4 * cq) = hv;
                    }
                }
            }
        }
        probe.tile_end(A_out, lane, m0, N, tile);
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
#undef GA2_DMA_AT
    ga_wait_vm<0>();   // the last tile's look-ahead pieces (re-fetched rows nobody reads) must land before the LDS is released
    // Leave the tile counter at zero for the next launch: every draw of this workgroup has returned (its value was consumed),
    // so the workgroup that arrives last resets both counters -- no memset on the stream between launches.
    // The range flags were OR-ed into the accumulator word; the last workgroup moves them to the status word (= the result
    // of THIS launch, overwritten by the next) and clears the accumulator.  vmcnt(0) above + the barrier: every wave's
    // atomicOr has reached L2 before this workgroup counts itself as finished.
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

// persistent launch: WGS workgroups per CU (or one per tile when there are fewer tiles).  The dynamic-LDS attribute and the CU
// count are looked up per DEVICE (a process may drive several GPUs, or switch device after the first call)
template <int ND, int KP, int XDT, int WV, bool PAIR>
int ga_launch_fwd2_w(const GaFwdArgs& a, bool pool, hipStream_t st) {
    using GP = Ga2Geom<ND, KP, XDT, WV, Ga2Tri<ND, XDT, true, WV, PAIR>::value>;      // the pooled (eval) kernel's geometry
    using GS = Ga2Geom<ND, KP, XDT, WV, false>;                                       // the score pass
    static int cus_of[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ACMIL_ERR_LAUNCH;
    if (cus_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ACMIL_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)ga_fwd2_kernel<ND, KP, XDT, true, false, WV, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, GP::LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)ga_fwd2_kernel<ND, KP, XDT, false, true, WV, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, GS::LDS) != hipSuccess)
            return ACMIL_ERR_LAUNCH;
        cus_of[dev] = prop.multiProcessorCount;
    }
    const int wgs = pool ? ((GP::WGS == 3 && a.no_tri) ? 2 : GP::WGS) : GS::WGS;      // (no_tri: A/B builds, two workgroups per CU for every family)
    const int slots = wgs * cus_of[dev];
    const int tiles = a.tile_start[a.nbags];
    const dim3 grid(tiles < slots ? tiles : slots), block(64 * WV);
    void (*kern)(GaFwdArgs) = pool ? ga_fwd2_kernel<ND, KP, XDT, true, false, WV, PAIR> : ga_fwd2_kernel<ND, KP, XDT, false, true, WV, PAIR>;
    hipLaunchKernelGGL(kern, grid, block, pool ? GP::LDS : GS::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

template <int ND, int KP, int XDT>
int ga_launch_fwd2(const GaFwdArgs& a, bool pool, hipStream_t st) {
    if constexpr (ND == 8) {
        if (a.pair_split && a.waves != 8) return ga_launch_fwd2_w<ND, KP, XDT, 4, true>(a, pool, st);
    }
    return a.waves == 8 ? ga_launch_fwd2_w<ND, KP, XDT, 8, false>(a, pool, st) : ga_launch_fwd2_w<ND, KP, XDT, 4, false>(a, pool, st);
}
