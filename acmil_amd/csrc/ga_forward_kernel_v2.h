// ga_forward_kernel_v2.h -- second-generation fused GA forward (split-f16 arithmetic only), gfx950.
//
// Same mathematics, same packed weight stream, same outputs and same per-workgroup partials as ga_fwd_kernel
// (ga_forward_kernel.h; reference: architecture/network.py:49-57, architecture/transformer.py:259-267, :322-324).
// What changed is the schedule of the two GEMM loops, which ran at ~57 % matrix-pipe efficiency in v1:
//
//   * software pipelining ACROSS the step barrier.  A step's 3 MFMA groups are issued in the order
//         P1 = Whi * xhi,  P2 = Whi * xlo,  P3 = Wlo * xhi
//     and P3 of step s is deferred until AFTER the barrier of step s+1: the barrier is followed immediately by the
//     ds_reads of the next "hi" fragments and the next x tile, and the 8 deferred MFMAs (256 matrix-pipe cycles) cover
//     their LDS latency and the fp32 -> f16 hi/lo conversion of x.  The "lo" fragments are requested under P1.  No
//     MFMA group ever waits for a read that was issued less than 8 MFMAs earlier, and the pipe stays fed across the barrier.
//   * LDS-DMA without vector address arithmetic: global_load_lds_dwordx4 in its SGPR-base + 32-bit lane offset +
//     immediate form.  Every wave copies RW CONSECUTIVE fragment rows plus its own x tile into one contiguous
//     per-wave region of the slot, so one M0 value and immediates -4096..+1024 address all of a step's pieces; the
//     per-step advance is scalar adds.  Pieces are issued one per MFMA gap of P1.
//   * slot s-1 is recycled right after barrier s (every wave drains lgkmcnt before arriving), which gives the same
//     prefetch distance as v1 (NB-1 steps) with the same ring.
// Phases outside the GEMM loops (relu/split, gate, scores, softmax, pooling, combine) are v1's.
#pragma once
#include "ga_forward_kernel.h"

// Timing-only ablations for tools/build_variants.sh (results are WRONG when any bit is set; never set in the product build):
// 1 no x DMA, 2 no W DMA, 4 no GEMM1 MFMAs, 8 no GEMM2 MFMAs, 16 no step barrier, 32 stop after the scores, 64 no gate,
// 128 no W fragment reads, 256 no x read / split.  GA2_LDS_PAD: extra LDS bytes (forces one workgroup per CU).
// GA2_PROF: s_memtime accounting of the waits; the per-wave cycle totals REPLACE the first 8 scores of each 32-patch group of A_out[0].
// GA2_NB4 / GA2_NB8: ring slots of the 4- / 8-wave workgroups (defaults 3 / 4).
// GA2_DEPHASE: the second co-resident workgroup of a CU delays its first tile by that many s_sleep(127) rounds.
#ifndef GA2_ABL
#define GA2_ABL 0
#endif
#define GA2_MFMA1(A, B, C) ((GA2_ABL & 4) ? ga2_keep(A, B, C) : __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0))
#define GA2_MFMA2(A, B, C) ((GA2_ABL & 8) ? ga2_keep(A, B, C) : __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0))
__device__ __forceinline__ f32x16 ga2_keep(f16x8 a, f16x8 b, f32x16 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }

template <int ND, int KP, int XDT, int WAVES>
struct Ga2Geom {
    static constexpr int XE = (XDT == ACMIL_DTYPE_F32) ? 4 : 2;     // bytes per bag element
    static constexpr int WROWS = 2 * ND;                            // fragment rows per step ("hi" rows then "lo" rows)
    static_assert(WROWS % WAVES == 0, "every wave copies the same number of consecutive fragment rows");
    static constexpr int RW = WROWS / WAVES;                        // fragment rows per wave per step
    static_assert(RW >= 1 && RW <= 4, "row immediates must fit -4096..-1024");
    static constexpr int DD = ND / 4;                               // h tiles consumed per GEMM2 step
    static constexpr int XG = 32 * 16 * XE / 1024;                  // x pieces (1 KiB) per wave per step
    static constexpr int NV = RW + XG;                              // LDS-DMA instructions per wave per step
    static constexpr int REGION = NV * 1024;                        // per-wave region of a slot: RW rows, then the x tile
    static constexpr int SLOT = WAVES * REGION;
#ifndef GA2_NB4
#define GA2_NB4 3
#endif
#ifndef GA2_NB8
#define GA2_NB8 4
#endif
    static constexpr int NB = (WAVES == 8) ? GA2_NB8 : GA2_NB4;     // ring slots (8-wave WG: 1 per CU; 4-wave WG: 2 per CU)
    static constexpr int PD = NB - 1;                               // prefetch distance in steps
    static constexpr int ROWS = 32 * WAVES;
    static_assert(ND % 4 == 0, "Di must be a multiple of 128");
    static constexpr int RING = NB * SLOT;
    static constexpr int POOLW = 64 * 36 * 4;
    static constexpr int REGION0 = (RING > WAVES * POOLW) ? RING : WAVES * POOLW;
    static constexpr int TAB_OFF = REGION0;
    static constexpr int TAB_BYTES = (2 + KP) * GA_DA * 4;
    static constexpr int PL_OFF = TAB_OFF + TAB_BYTES;
    static constexpr int PL_BYTES = WAVES * KP * 32 * 4;
#ifdef GA2_LDS_PAD
    static constexpr int LDS = PL_OFF + PL_BYTES + GA2_LDS_PAD;
#else
    static constexpr int LDS = PL_OFF + PL_BYTES;
#endif
    // byte offset of fragment row r inside a slot
    static constexpr int frow(int r) { return (r / RW) * REGION + (r % RW) * 1024; }
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

// f16 hi/lo split of two fp32 values in 3 VALU instructions: hi = rn_f16(x) (packed convert), lo = rn_f16(x - hi) by
// v_fma_mix{lo,hi}_f16 (f16 source * -1.0 + f32 source, ONE rounding to f16; x - hi is exact in fp32, so this equals the
// convert / subtract / convert sequence bit for bit).  The s_nops cover the partial-register-write forwarding hazard.
__device__ __forceinline__ void ga2_split_pair(float x0, float x1, unsigned& hi_pk, unsigned& lo_pk) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
        : "=&v"(hi_pk), "=&v"(lo_pk) : "v"(x0), "v"(x1));
}

template <int ND, int KP, int XDT, int WAVES, bool POOL, bool SAVEH>
__global__ __launch_bounds__(64 * WAVES, 2) void ga_fwd2_kernel(GaFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Ga2Geom<ND, KP, XDT, WAVES>;
    constexpr int NTHR = 64 * WAVES;
    constexpr bool XLO = (XDT != ACMIL_DTYPE_F16);   // fp16 bags are exact in the hi part
    constexpr int NCH = ND / 2;
    constexpr int Di = ND * 32;

    const GaLayout& L = a.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i31 = lane & 31, hi = lane >> 5;
    int bag = 0;
    while (bag + 1 < a.nbags && (int)blockIdx.x >= a.tile_start[bag + 1]) ++bag;
    const int N = a.Ns[bag], D = L.D, K = L.K;
    const char* xbase = (const char*)a.xs[bag];
    float* A_out = a.A_outs[bag];
    const int m0 = ((int)blockIdx.x - a.tile_start[bag]) * G::ROWS + wave * 32;
    const int row = m0 + i31;
    const bool valid = row < N;

    const char* wstream = a.packed + L.g1_off;
    const int S1 = D / 16;
    constexpr int S2 = 16;
    const int SL = S1 + S2 - 1;

    // ---- LDS-DMA addressing.  x tile of this wave: rows m0c .. m0c+31 (rows past the bag re-read its last row; their
    // results are discarded).  Lane offsets are 32-bit and relative to the wave's first row.
    const int m0c = m0 < N ? m0 : N - 1;
    unsigned xoff[G::XG];
#pragma unroll
    for (int q = 0; q < G::XG; ++q) {
        int r, piece;
        if constexpr (G::XG == 2) { r = 16 * q + (lane >> 2); piece = (lane & 3) ^ ((r >> 2) & 3); }
        else { r = lane >> 1; piece = (lane & 1) ^ ((r >> 3) & 1); }
        const int rmax = N - 1 - m0c;
        r = r < rmax ? r : rmax;
        xoff[q] = (unsigned)r * (unsigned)(D * G::XE) + piece * 16;
    }
    const unsigned woff = lane * 16;
    const char* const xrow0 = xbase + (size_t)m0c * D * G::XE;                          // wave-uniform
    const char* const wreg0 = wstream + (size_t)(wave * G::RW + G::RW) * GA_FRAG_ROW;   // wave-uniform, biased by RW rows
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const unsigned m0w = lds_base + wave * G::REGION + G::RW * 1024;                    // M0 of slot 0

    // piece p (compile time) of step u into ring slot `slot`: p < RW weight rows, then the x pieces
    auto dma_piece = [&](auto pc, int u, int slot) {
        constexpr int p = decltype(pc)::value;
        if constexpr (((GA2_ABL & 2) && p < G::RW) || ((GA2_ABL & 1) && p >= G::RW)) return;
        const unsigned m0v = m0w + slot * G::SLOT;
        if constexpr (p < G::RW) {
            const int uw = u < SL ? u : SL;
            ga2_dma<-(G::RW - p) * 1024>(woff, wreg0 + (size_t)uw * G::WROWS * GA_FRAG_ROW, m0v);
        } else {
            constexpr int q = p - G::RW;
            const int ux = u < S1 ? u : S1 - 1;
            ga2_dma<q * 1024>(xoff[q], xrow0 + (size_t)ux * 16 * G::XE - q * 1024, m0v);
        }
    };
    auto issue_all = [&](int u, int slot) {
        dma_piece(std::integral_constant<int, 0>{}, u, slot);
        if constexpr (G::NV > 1) dma_piece(std::integral_constant<int, 1>{}, u, slot);
        if constexpr (G::NV > 2) dma_piece(std::integral_constant<int, 2>{}, u, slot);
        if constexpr (G::NV > 3) dma_piece(std::integral_constant<int, 3>{}, u, slot);
        if constexpr (G::NV > 4) dma_piece(std::integral_constant<int, 4>{}, u, slot);
        if constexpr (G::NV > 5) dma_piece(std::integral_constant<int, 5>{}, u, slot);
    };
    static_assert(G::NV <= 6, "issue_all covers at most 6 pieces");

    // epilogue vectors bv, bu, Ww -> LDS (rows K..KP-1 of Ww zero); visible after the first step barrier
    {
        const float* src = (const float*)(a.packed + L.tab_off);
        float* dst = (float*)(smem + G::TAB_OFF);
        for (int e = tid; e < (2 + KP) * GA_DA; e += NTHR) dst[e] = (e < (2 + K) * GA_DA) ? src[e] : 0.0f;
    }

#ifdef GA2_DEPHASE
    // HW_REG_LDS_ALLOC[7:0] = LDS base of this workgroup: non-zero for the second workgroup resident on a CU
    if (__builtin_amdgcn_s_getreg(6 | (0 << 6) | (7 << 11)) != 0 && blockIdx.x < 1024)
        for (int i = 0; i < GA2_DEPHASE; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    f32x16 acc1[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[d][r] = 0.0f;

    int islot = 0;   // ring slot the next issued step goes to
#pragma unroll
    for (int s = 0; s < G::PD; ++s) { issue_all(s, islot); islot = (islot + 1 == G::NB) ? 0 : islot + 1; }

    // ---- x operand of a step: raw read from the wave's own tile, then f16 hi/lo split
    const int xrd0 = wave * G::REGION + G::RW * 1024 +
                     ((G::XG == 2) ? (i31 * 64 + (((2 * hi) ^ ((i31 >> 2) & 3)) * 16)) : (i31 * 32 + ((hi ^ ((i31 >> 3) & 1)) * 16)));
    const int xrd1 = wave * G::REGION + G::RW * 1024 + i31 * 64 + (((2 * hi + 1) ^ ((i31 >> 2) & 3)) * 16);   // fp32 only
    f32x4 xr0, xr1;      // raw fp32
    u32x4 xrw;           // raw 16-bit (8 elements)
    auto read_x = [&](const char* slot) {
        if constexpr (GA2_ABL & 256) { asm volatile("" : "+v"(xr0), "+v"(xr1), "+v"(xrw)); return; }
        if constexpr (XDT == ACMIL_DTYPE_F32) { xr0 = *(const f32x4*)(slot + xrd0); xr1 = *(const f32x4*)(slot + xrd1); }
        else xrw = *(const u32x4*)(slot + xrd0);
    };
    // split piece j (< NSP) of the raw tile into word j of the hi / lo operands (two K slots per piece)
    constexpr int NSP = XLO ? 4 : 0;
    u32x4 xhw, xlw;
    auto split_piece = [&](int j) {
        if constexpr (GA2_ABL & 256) { asm volatile("" : "+v"(xhw), "+v"(xlw)); return; }
        float v0, v1;
        if constexpr (XDT == ACMIL_DTYPE_F32) {
            v0 = j < 2 ? xr0[2 * (j & 1)] : xr1[2 * (j & 1)];
            v1 = j < 2 ? xr0[2 * (j & 1) + 1] : xr1[2 * (j & 1) + 1];
        } else {
            v0 = __builtin_bit_cast(float, xrw[j] << 16);
            v1 = __builtin_bit_cast(float, xrw[j] & 0xffff0000u);
        }
        unsigned h, l;
        ga2_split_pair(v0, v1, h, l);
        xhw[j] = h; xlw[j] = l;
    };
    auto split_done = [&](f16x8& h8, f16x8& l8) {
        if constexpr (XLO) { h8 = __builtin_bit_cast(f16x8, xhw); l8 = __builtin_bit_cast(f16x8, xlw); }
        else h8 = __builtin_bit_cast(f16x8, xrw);
    };

    // =========================================================== GEMM1: h^T = W1 * x^T
    f16x8 WH[ND], WL[ND];
    f16x8 xh, xl, xhp;
    int rslot = 0;   // ring slot of the step being consumed
    const int lane16 = lane * 16;
    auto read_hi = [&](const char* slot) {
        if constexpr (GA2_ABL & 128) { for (int d = 0; d < ND; ++d) asm volatile("" : "+v"(WH[d])); return; }
#pragma unroll
        for (int d = 0; d < ND; ++d) WH[d] = *(const f16x8*)(slot + G::frow(d) + lane16);
    };
    auto read_lo = [&](const char* slot) {
        if constexpr (GA2_ABL & 128) { for (int d = 0; d < ND; ++d) asm volatile("" : "+v"(WL[d])); return; }
#pragma unroll
        for (int d = 0; d < ND; ++d) WL[d] = *(const f16x8*)(slot + G::frow(ND + d) + lane16);
    };
#ifdef GA2_PROF
    unsigned long long pf_vm = 0, pf_bar = 0, pf_vm1 = 0, pf_bar1 = 0;
    const unsigned long long pf_t0 = __builtin_amdgcn_s_memtime();
#endif
    auto step_sync = [&]() {
#ifdef GA2_PROF
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
        ga_wait_vm<(G::PD - 1) * G::NV>();      // this wave's pieces of the step about to be consumed have landed
#ifdef GA2_PROF
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): every read of the slot about to be recycled has returned
        if constexpr (!(GA2_ABL & 16)) __builtin_amdgcn_s_barrier();
#ifdef GA2_PROF
        const unsigned long long tc = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        pf_vm += tb - ta; pf_bar += tc - tb;
#endif
    };

    // ---- step 0 (nothing deferred yet)
    step_sync();
    {
        const char* slot = smem + rslot * G::SLOT;
        read_x(slot);
        read_hi(slot);
#pragma unroll
        for (int j = 0; j < NSP; ++j) split_piece(j);
        split_done(xh, xl);
        __builtin_amdgcn_sched_barrier(0);
        read_lo(slot);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            acc1[d] = GA2_MFMA1(WH[d], xh, acc1[d]);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_all(G::PD, islot);
        islot = (islot + 1 == G::NB) ? 0 : islot + 1;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (XLO) {
#pragma unroll
            for (int d = 0; d < ND; ++d) acc1[d] = GA2_MFMA1(WH[d], xl, acc1[d]);
        }
        xhp = xh;
        rslot = (rslot + 1 == G::NB) ? 0 : rslot + 1;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int s = 1; s < S1; ++s) {
        step_sync();
        const char* slot = smem + rslot * G::SLOT;
        read_x(slot);      // first: the LDS returns data in order, so the split can start under the hi-fragment reads
        read_hi(slot);
        __builtin_amdgcn_sched_barrier(0);
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
        split_done(xh, xl);
        __builtin_amdgcn_sched_barrier(0);
        read_lo(slot);
        // P1(s), one LDS-DMA piece of step s+PD per MFMA gap
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            acc1[d] = GA2_MFMA1(WH[d], xh, acc1[d]);
            __builtin_amdgcn_sched_barrier(0);
            if (d == 0) dma_piece(std::integral_constant<int, 0>{}, s + G::PD, islot);
            if (d == 1 && G::NV > 1) dma_piece(std::integral_constant<int, (G::NV > 1 ? 1 : 0)>{}, s + G::PD, islot);
            if (d == 2 && G::NV > 2) dma_piece(std::integral_constant<int, (G::NV > 2 ? 2 : 0)>{}, s + G::PD, islot);
            if (d == 3 && G::NV > 3) dma_piece(std::integral_constant<int, (G::NV > 3 ? 3 : 0)>{}, s + G::PD, islot);
            if (d == 4 && G::NV > 4) dma_piece(std::integral_constant<int, (G::NV > 4 ? 4 : 0)>{}, s + G::PD, islot);
            if (d == 5 && G::NV > 5) dma_piece(std::integral_constant<int, (G::NV > 5 ? 5 : 0)>{}, s + G::PD, islot);
            __builtin_amdgcn_sched_barrier(0);
        }
        islot = (islot + 1 == G::NB) ? 0 : islot + 1;
        if constexpr (XLO) {
#pragma unroll
            for (int d = 0; d < ND; ++d) acc1[d] = GA2_MFMA1(WH[d], xl, acc1[d]);
        }
        __builtin_amdgcn_sched_barrier(0);
        xhp = xh;
        rslot = (rslot + 1 == G::NB) ? 0 : rslot + 1;
    }
    // P3 of the last GEMM1 step
#pragma unroll
    for (int d = 0; d < ND; ++d) acc1[d] = GA2_MFMA1(WL[d], xhp, acc1[d]);
#ifdef GA2_PROF
    const unsigned long long pf_t1 = __builtin_amdgcn_s_memtime();
    pf_vm1 = pf_vm; pf_bar1 = pf_bar;
#endif

    // =========================================================== relu + f16 split of h (see v1 for the pinning notes)
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[d][r] = fmaxf(acc1[d][r], 0.0f);
#pragma unroll
    for (int d = 0; d < ND; ++d) asm volatile("" : "+v"(acc1[d]));

    f16x8 hh[ND][2], hl[ND][2];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = acc1[d][8 * e + j];
                const _Float16 h16 = (_Float16)v;
                hh[d][e][j] = h16;
                hl[d][e][j] = (_Float16)(v - (float)h16);
            }
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            asm volatile("" : "+v"(hh[d][e]));
            asm volatile("" : "+v"(hl[d][e]));
        }

    // =========================================================== GEMM2 (four unit blocks) + gate + scores
    // fragment rows of a step: (dd*2 + part)*4 + t, t = e*2 + al; WH/WL hold the part-0 / part-1 rows of both h tiles
    const float* tabf = (const float*)(smem + G::TAB_OFF);
    float sc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) sc[k] = 0.0f;
    static_assert(4 * G::DD == ND, "WH/WL hold the 4*DD fragments of a GEMM2 step part");
    auto read2 = [&](const char* slot, int part, f16x8 (&W)[ND]) {
        if constexpr (GA2_ABL & 128) { for (int d = 0; d < ND; ++d) asm volatile("" : "+v"(W[d])); return; }
#pragma unroll
        for (int dd = 0; dd < G::DD; ++dd)
#pragma unroll
            for (int t = 0; t < 4; ++t) W[dd * 4 + t] = *(const f16x8*)(slot + G::frow((dd * 2 + part) * 4 + t) + lane16);
    };

#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        f32x16 acc2[2];
#pragma unroll
        for (int al = 0; al < 2; ++al)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                int boff = 32 * g + 8 * rq + 4 * hi;
                asm volatile("" : "+v"(boff) : "v"(sc[0]));
                const f32x4 b = *(const f32x4*)(tabf + al * GA_DA + boff);
                acc2[al][4 * rq + 0] = b[0]; acc2[al][4 * rq + 1] = b[1];
                acc2[al][4 * rq + 2] = b[2]; acc2[al][4 * rq + 3] = b[3];
            }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int u = S1 + g * 4 + st;
            step_sync();
            const char* slot = smem + rslot * G::SLOT;
            read2(slot, 0, WH);
            __builtin_amdgcn_sched_barrier(0);
            if (st > 0) {   // P3 of the previous step of this block
#pragma unroll
                for (int dd = 0; dd < G::DD; ++dd)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc2[t & 1] = GA2_MFMA2(WL[dd * 4 + t], hh[G::DD * (st - 1) + dd][t >> 1], acc2[t & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            read2(slot, 1, WL);
#pragma unroll
            for (int dd = 0; dd < G::DD; ++dd)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = dd * 4 + t;
                    acc2[t & 1] = GA2_MFMA2(WH[m], hh[G::DD * st + dd][t >> 1], acc2[t & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m == 0) dma_piece(std::integral_constant<int, 0>{}, u + G::PD, islot);
                    if (m == 1 && G::NV > 1) dma_piece(std::integral_constant<int, (G::NV > 1 ? 1 : 0)>{}, u + G::PD, islot);
                    if (m == 2 && G::NV > 2) dma_piece(std::integral_constant<int, (G::NV > 2 ? 2 : 0)>{}, u + G::PD, islot);
                    if (m == 3 && G::NV > 3) dma_piece(std::integral_constant<int, (G::NV > 3 ? 3 : 0)>{}, u + G::PD, islot);
                    if (m == 4 && G::NV > 4) dma_piece(std::integral_constant<int, (G::NV > 4 ? 4 : 0)>{}, u + G::PD, islot);
                    if (m == 5 && G::NV > 5) dma_piece(std::integral_constant<int, (G::NV > 5 ? 5 : 0)>{}, u + G::PD, islot);
                    __builtin_amdgcn_sched_barrier(0);
                }
            islot = (islot + 1 == G::NB) ? 0 : islot + 1;
#pragma unroll
            for (int dd = 0; dd < G::DD; ++dd)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc2[t & 1] = GA2_MFMA2(WH[dd * 4 + t], hl[G::DD * st + dd][t >> 1], acc2[t & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (st == 3) {   // the block's last step finishes its own P3: the gate needs the complete accumulators
#pragma unroll
                for (int dd = 0; dd < G::DD; ++dd)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc2[t & 1] = GA2_MFMA2(WL[dd * 4 + t], hh[G::DD * 3 + dd][t >> 1], acc2[t & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            rslot = (rslot + 1 == G::NB) ? 0 : rslot + 1;
        }
        if constexpr (GA2_ABL & 64) { asm volatile("" :: "v"(acc2[0]), "v"(acc2[1])); continue; }
        // gate + partial scores for the 32 units of this block (this lane: 16 of them)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            int ubase = 32 * g + 8 * rq + 4 * hi;
            float gate[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gate[q] = ga_tanh(acc2[0][4 * rq + q]) * ga_sigmoid(acc2[1][4 * rq + q]);
            asm volatile("" : "+v"(ubase) : "v"(gate[3]));
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const f32x4 w = *(const f32x4*)(tabf + (2 + k) * GA_DA + ubase);
                sc[k] = fmaf(gate[0], w[0], sc[k]); sc[k] = fmaf(gate[1], w[1], sc[k]);
                sc[k] = fmaf(gate[2], w[2], sc[k]); sc[k] = fmaf(gate[3], w[3], sc[k]);
            }
        }
    }

#ifdef GA2_PROF
    const unsigned long long pf_t2 = __builtin_amdgcn_s_memtime();
#endif
    const float* bwp = (const float*)(a.packed + L.bw_off);
    float smax[KP], lsum[KP], pe[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        sc[k] += __shfl_xor(sc[k], 32);
        sc[k] += bwp[k];
        if (A_out && valid && hi == 0 && k < K) A_out[(size_t)k * N + row] = sc[k];
        float m = valid ? sc[k] : -INFINITY;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        smax[k] = m;
        pe[k] = valid ? __expf(sc[k] - m) : 0.0f;
        float l = pe[k];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) l += __shfl_xor(l, o);
        lsum[k] = l;
    }

    if constexpr (GA2_ABL & 32) return;
    // =========================================================== attention-weighted sum  sum_n p[k][n] h[n][:]
    ga_wait_vm<0>();  // the clamped tail DMAs still target the ring: drain them before it is reused
    __syncthreads();
    float* pool = (float*)(smem + wave * G::POOLW);
    float* pl = (float*)(smem + G::PL_OFF) + (size_t)wave * KP * 32;
    if (POOL && hi == 0) {
#pragma unroll
        for (int k = 0; k < KP; ++k) pl[k * 32 + i31] = pe[k];
    }
    float pacc[NCH][KP];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int k = 0; k < KP; ++k) pacc[c][k] = 0.0f;

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dl = 0; dl < 2; ++dl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = (float)hh[2 * c + dl][r >> 3][r & 7] + (float)hl[2 * c + dl][r >> 3][r & 7];
                pool[(dl * 32 + mfma32_row(r, hi)) * 36 + i31] = hv;
            }
        __builtin_amdgcn_wave_barrier();
        const f32x4* prow = (const f32x4*)(pool + lane * 36);
#pragma unroll 2
        for (int mq = 0; mq < 8; ++mq) {
            const f32x4 hv = prow[mq];
            if constexpr (SAVEH) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (m0 + 4 * mq + e < N) a.h_save[(size_t)(m0 + 4 * mq + e) * Di + 64 * c + lane] = hv[e];
            }
            if constexpr (POOL) {
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const f32x4 p = *(const f32x4*)(pl + k * 32 + 4 * mq);
                    pacc[c][k] = fmaf(p[0], hv[0], pacc[c][k]); pacc[c][k] = fmaf(p[1], hv[1], pacc[c][k]);
                    pacc[c][k] = fmaf(p[2], hv[2], pacc[c][k]); pacc[c][k] = fmaf(p[3], hv[3], pacc[c][k]);
                }
            }
        }
    }
    if constexpr (!POOL) return;

    // =========================================================== combine the waves, publish the partial
    __builtin_amdgcn_wave_barrier();
    constexpr int PS = 2 + Di;
    static_assert(KP * PS * 4 <= G::POOLW, "combine record must fit the wave's pooling tile");
    float* comb = pool;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K) {
            if (lane == 0) { comb[k * PS + 0] = smax[k]; comb[k * PS + 1] = lsum[k]; }
#pragma unroll
            for (int c = 0; c < NCH; ++c) comb[k * PS + 2 + 64 * c + lane] = pacc[c][k];
        }
    }
    __syncthreads();
    float* out = a.part + (size_t)blockIdx.x * K * PS;
    for (int k = 0; k < K; ++k) {
        float mw[WAVES], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            mw[w] = ((const float*)(smem + w * G::POOLW))[k * PS + 0];
            M = fmaxf(M, mw[w]);
        }
        float fw[WAVES];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) fw[w] = (mw[w] == -INFINITY) ? 0.0f : __expf(mw[w] - M);
        for (int e = tid; e < PS; e += NTHR) {
            float v;
            if (e == 0) v = M;
            else {
                v = 0.0f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) v = fmaf(fw[w], ((const float*)(smem + w * G::POOLW))[k * PS + e], v);
            }
            out[k * PS + e] = v;
        }
    }
#ifdef GA2_PROF
    {
        const unsigned long long pf_t3 = __builtin_amdgcn_s_memtime();
        const float pv[8] = {(float)(pf_t3 - pf_t0), (float)pf_vm1, (float)pf_bar1, (float)(pf_vm - pf_vm1), (float)(pf_bar - pf_bar1),
                             (float)(pf_t1 - pf_t0), (float)(pf_t2 - pf_t1), (float)(pf_t3 - pf_t2)};
        __syncthreads();
        if (A_out && lane < 8 && m0 + 8 <= N) A_out[m0 + lane] = pv[lane];
    }
#endif
}

template <int ND, int KP, int XDT, int WAVES>
int ga_launch_fwd2_w(const GaFwdArgs& a, bool pool, hipStream_t st) {
    using G = Ga2Geom<ND, KP, XDT, WAVES>;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
#ifndef GA2_LDS_PAD
    static_assert(WAVES == 8 || 2 * G::LDS <= 160 * 1024, "two 4-wave workgroups must fit one CU");
#endif
    const dim3 grid(a.tile_start[a.nbags]), block(64 * WAVES);
    void (*kern)(GaFwdArgs) = pool ? ga_fwd2_kernel<ND, KP, XDT, WAVES, true, false>
                                   : ga_fwd2_kernel<ND, KP, XDT, WAVES, false, true>;
    static const hipError_t attr[2] = {
        hipFuncSetAttribute((const void*)ga_fwd2_kernel<ND, KP, XDT, WAVES, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS),
        hipFuncSetAttribute((const void*)ga_fwd2_kernel<ND, KP, XDT, WAVES, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS)};
    if (attr[0] != hipSuccess || attr[1] != hipSuccess) return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, block, G::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

template <int ND, int KP, int XDT>
int ga_launch_fwd2(const GaFwdArgs& a, bool pool, hipStream_t st) {
    return a.waves == 4 ? ga_launch_fwd2_w<ND, KP, XDT, 4>(a, pool, st) : ga_launch_fwd2_w<ND, KP, XDT, 8>(a, pool, st);
}
