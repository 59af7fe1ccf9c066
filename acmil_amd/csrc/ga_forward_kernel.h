// ga_forward_kernel.h -- fused forward of ACMIL's gated-attention aggregation for ONE bag on MI355X (gfx950).
//
// Replaces (reference file:line, /root/reference):
//   DimReduction.forward        architecture/network.py:49-57      h = relu(x W1^T)
//   Attention_Gated.forward     architecture/transformer.py:259-267 A = ((tanh(hWv^T+bv)*sigmoid(hWu^T+bu))Ww^T+bw)^T
//   ACMIL_GA.forward            architecture/transformer.py:322-324 softmax over N, P h   (heads: ga_forward.hip)
//
// Design (one pass over x, no intermediate ever touches HBM):
//   * grid = ceil(N/256) workgroups x 8 waves (2 per SIMD, <=256 registers each); every wave owns 32
//     consecutive patches end to end, so there is no inter-wave data dependence in the GEMM chain -- the
//     workgroup only shares the weight stream (one pass of the packed weights serves 256 patches).
//   * both GEMMs are computed TRANSPOSED (D = W * x^T): patches live on the MFMA column index (= lane&31),
//     output features on the accumulator registers.  An accumulator register file in that layout IS the
//     B operand of the next MFMA chain (K index = register index, lane = column), so relu(h) feeds GEMM2
//     straight from registers; the K-slot permutation this implies is folded into the weight packing.
//   * ALL global reads of the main loop are LDS-DMA (global_load_lds, 16 B/lane): the pre-packed weight
//     fragment stream (linear copy) and the x tile (full 64-B / 32-B row segments per lane group, XOR
//     swizzled on the SOURCE address so the B-fragment ds_read_b128 is bank-conflict free).  They land in a
//     4-slot ring; a step (16 K-values of GEMM1, or one 32-feature slice of GEMM2) waits with a COUNTED
//     s_waitcnt vmcnt(N) that leaves the next two steps' loads in flight, then one raw s_barrier.
//   * GEMM2 runs in four blocks of 32 attention units (2 MFMA tiles: tanh / sigmoid branch) so only 32
//     accumulator registers are live next to the 128 that hold h; biases are the accumulator init.
//   * epilogue in registers: tanh/sigmoid gate, K-way score dot with Ww (per-lane partial + one cross-half
//     shuffle), A_out store, wave-level online-softmax statistics; the attention-weighted sum P h transposes
//     h through a wave-private padded LDS tile (fp32) and accumulates on the VALU with lane = feature.
//   * per-workgroup (m, l, acc[Di]) partials go to the workspace; ga_merge_kernel combines them in a fixed
//     order (deterministic) and ga_heads_kernel applies the K branch heads and the bag head.
//
// Arithmetic modes (MODE): F32 = v_mfma_f32_32x32x2_f32 (exact fp32); F16X3 = every fp32 operand split into
// f16 hi + lo, products hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16 with fp32 accumulation (error
// ~1e-6 relative, keeps the reference's top-k order); F16 = single f16 pass (throughput mode).
#pragma once
#include "ga_common.h"

#define GA_MAX_BATCH ACMIL_MAX_BATCH      // bags per launch (kernel-argument arrays); a launch of 64 x 50 000 patches is 48 rounds of tiles on 512 slots

struct GaFwdArgs {
    const void* xs[GA_MAX_BATCH];     // bag matrices [N_b, D]
    float* A_outs[GA_MAX_BATCH];      // [K, N_b] or null
    int Ns[GA_MAX_BATCH];
    int tile_start[GA_MAX_BATCH + 1]; // first workgroup (tile) of each bag; tile_start[nbags] = grid size
    int nbags;
    const char* packed;
    float* part;     // workspace partials [total tiles][K][2+Di]
    float* h_save;   // [N,Di] or null (single-bag score pass only)
    int waves;       // 8 or 4 waves per workgroup (tile = 32 * waves patches); the persistent v2 kernel always uses 4
    unsigned* tile_counter;   // v2: zeroed word the persistent workgroups draw their next tiles from (null = static striding)
    unsigned* status;         // v2: zeroed word; bit 0 = a bag value, bit 1 = a projected feature h outside the f16 range (or NaN):
                              //     the split-f16 result is then NOT the fp32 result and the caller must redo the bag in fp32 mode
    int self_reset;  // v2: 1 = the last workgroup leaves the control block's counters at zero (default); 0 = the host memsets (A/B knob)
    int pair_split;  // v2, D_inner = 256: GEMM1 with the feature tiles split over wave pairs (half the weight-fragment reads per MFMA)
    int dephase;     // v2: start delay of the second workgroup of a CU, in s_sleep(127) rounds (~8 k cycles each); 0 = none
    int no_tri;      // v2, D_inner = 128 family on 16-bit bags: 1 = launch two workgroups per CU instead of three (A/B builds only)
    int v3;          // 1 = the one-wave-per-SIMD kernel (ga_forward_kernel_v3.h): always for D_inner 384 / 512, 64-patch wave tiles at 256
    const unsigned* cond;     // v1 (fp32 repeat of the device-side range guard): run only if *cond != 0 (the status word the preceding
                              //     split-f16 launch on this stream left); null = unconditional
    unsigned* cond_count;     // v1: incremented once per launch that did run under `cond` (the module's fallback counter); may be null
    GaLayout L;
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of 16 B per lane (global_load_lds_dwordx4): lane l's 16 bytes at gsrc land at ldst + 16*l (ldst wave-uniform).
// Issued through inline asm on purpose: with the __builtin_amdgcn_global_load_lds builtin hipcc's wait-count pass
// treats the LDS counter as out of order and guards EVERY ds_read consumer with s_waitcnt lgkmcnt(0) -- a wave then
// waits for all 16 fragment reads of a step before its first MFMA instead of counting down (15, 14, ...).  The asm
// form is invisible to that pass; its completion is tracked by the hand-counted ga_wait_vm<> below.  M0 carries the
// LDS destination and is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void ga_glds16(const char* gsrc, unsigned ldst) {   // ldst: wave-uniform LDS byte address
    unsigned keep;
    const unsigned lds = __builtin_amdgcn_readfirstlane(ldst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
}
// (A non-temporal variant of this copy was measured for the bag and rejected: a step reads one 64-byte segment per patch
//  row and the other half of each 128-byte line is wanted one step later; without L2 retention FETCH_SIZE rose by 79 %.)
#ifdef GA_GLDS_BUILTIN
#define GA_GLDS16(gsrc, ldst) __builtin_amdgcn_global_load_lds((gptr_t)(gsrc), (lptr_t)(size_t)(ldst), 16, 0, 0)
#else
#define GA_GLDS16(gsrc, ldst) ga_glds16((const char*)(gsrc), (unsigned)(ldst))
#endif


template <int N>
__device__ __forceinline__ void ga_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int ND, int KP, int MODE, int XDT, int WAVES>
struct GaGeom {
    static constexpr int XE = (XDT == ACMIL_DTYPE_F32) ? 4 : 2;       // bytes per bag element
    static constexpr int WROWS = (MODE == ACMIL_MODE_F16) ? ND : 2 * ND;   // fragment rows per step (GEMM1 and GEMM2 alike)
    static constexpr int DD = ND / 4;                                 // h tiles consumed per GEMM2 step (4 steps per unit block)
    static constexpr int WG1 = (WROWS + WAVES - 1) / WAVES;           // weight LDS-DMA instructions per wave per step
    static constexpr int XG = 32 * 16 * XE / 1024;                    // x LDS-DMA instructions per wave per step
    static constexpr int N1 = WG1 + XG;                               // VMEM ops per wave per step (every step issues the same)
    // 8-wave workgroups ALTERNATE the LDS-DMA issue between the two waves of each SIMD: waves 0-3 issue the even
    // steps, waves 4-7 the odd ones (for both x tiles of the pair and 1/4 of the weight rows).  While one wave of a SIMD
    // spends ~600 cycles in the vector-memory issue path, its partner has the matrix pipe to itself.
    static constexpr bool ALT = (WAVES == 8);
    static constexpr int WGA = (WROWS + 3) / 4;                       // weight rows per issuing wave (ALT)
    static constexpr int NA = 2 * XG + WGA;                           // VMEM ops per issuing wave per step (ALT)
    static constexpr int WSLOT = (WROWS + WAVES - 1) / WAVES * WAVES * GA_FRAG_ROW;
    static constexpr int XB = 32 * 16 * XE;                           // x bytes per wave per step
    static constexpr int SLOT = WSLOT + WAVES * XB;
    static constexpr int NB = (WAVES == 8) ? 4 : 3;                   // ring slots (8-wave WG: 1 per CU; 4-wave WG: 2 per CU)
    static constexpr int PD = NB - 1;                                 // prefetch distance in steps
    static constexpr int ROWS = 32 * WAVES;                           // patches per workgroup
    static_assert(ND % 4 == 0, "Di must be a multiple of 128");
    static constexpr int RING = NB * SLOT;
    static constexpr int POOLW = 64 * 36 * 4;                         // wave-private [64 di][32 m (+4 pad)] fp32
    static constexpr int REGION0 = (RING > WAVES * POOLW) ? RING : WAVES * POOLW;
    static constexpr int TAB_OFF = REGION0;                           // bv[128], bu[128], Ww[KP][128]
    static constexpr int TAB_BYTES = (2 + KP) * GA_DA * 4;
    static constexpr int PL_OFF = TAB_OFF + TAB_BYTES;                // p_lds: [wave][KP][32 m] fp32
    static constexpr int PL_BYTES = WAVES * KP * 32 * 4;
    static constexpr int LDS = PL_OFF + PL_BYTES;
};

template <int ND, int KP, int MODE, int XDT, int WAVES, bool POOL, bool SAVEH>
__global__ __launch_bounds__(64 * WAVES, 2) void ga_fwd_kernel(GaFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = GaGeom<ND, KP, MODE, XDT, WAVES>;
    constexpr int NTHR = 64 * WAVES;
    constexpr bool F32M = (MODE == ACMIL_MODE_F32);
    constexpr bool SPLIT = (MODE == ACMIL_MODE_F16X3);
    constexpr bool XLO = SPLIT && (XDT != ACMIL_DTYPE_F16);   // fp16 bags are exact in the hi part
    constexpr int PARTS = SPLIT ? 2 : 1;
    constexpr int NCH = ND / 2;                                // pooling chunks of 64 features
    constexpr int Di = ND * 32;

    const GaLayout& L = a.L;
    const int tid = threadIdx.x, lane = tid & 63;
    if (a.cond) {      // predicated repeat: the whole grid leaves at once unless the split-f16 launch before it flagged its bag(s)
        if (__builtin_nontemporal_load(a.cond) == 0u) return;
        if (blockIdx.x == 0 && tid == 0 && a.cond_count) atomicAdd(a.cond_count, 1u);
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i31 = lane & 31, hi = lane >> 5;
    // batched launch: one grid covers the tiles of up to GA_MAX_BATCH bags (fills the CUs a single 50k-patch bag leaves idle)
    int bag = 0;
    while (bag + 1 < a.nbags && (int)blockIdx.x >= a.tile_start[bag + 1]) ++bag;
    const int N = a.Ns[bag], D = L.D, K = L.K;
    const char* xbase = (const char*)a.xs[bag];
    float* A_out = a.A_outs[bag];
    const int m0 = ((int)blockIdx.x - a.tile_start[bag]) * G::ROWS + wave * 32;
    const int row = m0 + i31;
    const bool valid = row < N;

    const char* wstream = a.packed + L.g1_off;   // GEMM1 rows then GEMM2 rows, contiguous: step u starts at u * WROWS rows
    const int S1 = D / 16;                 // GEMM1 steps
    constexpr int S2 = 16;                 // GEMM2 steps: 4 unit blocks x 4 (each consumes DD h tiles)
    const int SL = S1 + S2 - 1;            // last real step

    // ---- per-lane source pointer of the x tile copy.  fp32: 2 instructions/step, lane l -> (row 16q + l/4, 16-B piece
    // (l&3) ^ swz(row)); 16-bit: 1 instruction, lane l -> (row l/2, piece (l&1) ^ swz(row)).  LDS image is lane-linear.
    constexpr int NXT = G::ALT ? 2 : 1;        // x tiles this wave copies: its own, and (ALT) its SIMD partner's (wave ^ 4)
    const char* xsrc[NXT][G::XG];
#pragma unroll
    for (int t = 0; t < NXT; ++t)
#pragma unroll
        for (int q = 0; q < G::XG; ++q) {
            int r, piece;
            if constexpr (G::XG == 2) { r = 16 * q + (lane >> 2); piece = (lane & 3) ^ ((r >> 2) & 3); }
            else { r = lane >> 1; piece = (lane & 1) ^ ((r >> 3) & 1); }
            int gr = m0 + (t ? ((wave ^ 4) - wave) * 32 : 0) + r;
            gr = gr < N ? gr : N - 1;
            gr = gr < 0 ? 0 : gr;
            xsrc[t][q] = xbase + (size_t)gr * D * G::XE + piece * 16;
        }
    const int wrow0 = G::ALT ? (wave & 3) : wave;                 // first weight row this wave copies; stride 4 (ALT) or WAVES
    const char* wsrc = wstream + lane * 16;

    // Every issuing wave issues the SAME number of LDS-DMA instructions per step (so all s_waitcnt counts are constants):
    // steps past the end of a stream re-fetch its last step (clamped source, L2 hits) into the slot the ring discipline
    // assigns -- data nobody reads.
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;   // LDS byte address of the dynamic region
    auto issue_step = [&](int u) {
        if constexpr (G::ALT) { if ((wave >> 2) != (u & 1)) return; }
        const unsigned slot = lds_base + (u % G::NB) * G::SLOT;
        const int ux = u < S1 ? u : S1 - 1;
        const int uw = u < SL ? u : SL;
#pragma unroll
        for (int t = 0; t < NXT; ++t) {                           // x first (see the waits)
            const unsigned xdst = slot + G::WSLOT + (t ? (wave ^ 4) : wave) * G::XB;
#pragma unroll
            for (int q = 0; q < G::XG; ++q) GA_GLDS16(xsrc[t][q] + (size_t)ux * 16 * G::XE, xdst + q * 1024);
        }
        const char* src = wsrc + (size_t)uw * G::WROWS * GA_FRAG_ROW;
        constexpr int RSTRIDE = G::ALT ? 4 : WAVES;
        constexpr int NW = G::ALT ? G::WGA : G::WG1;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int r = wrow0 + q * RSTRIDE;                    // rows beyond WROWS (padding) re-fetch a valid row
            GA_GLDS16(src + (size_t)(r % G::WROWS) * GA_FRAG_ROW, slot + r * GA_FRAG_ROW);
        }
    };

    // epilogue vectors bv, bu, Ww -> LDS (rows K..KP-1 of Ww zero); visible after the first step barrier
    {
        const float* src = (const float*)(a.packed + L.tab_off);
        float* dst = (float*)(smem + G::TAB_OFF);
        for (int e = tid; e < (2 + KP) * GA_DA; e += NTHR) dst[e] = (e < (2 + K) * GA_DA) ? src[e] : 0.0f;
    }

    f32x16 acc1[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[d][r] = 0.0f;

#pragma unroll
    for (int s = 0; s < G::PD; ++s) issue_step(s);

    // =========================================================== GEMM1: h^T = W1 * x^T
    // lane (m = lane&31, hi) owns the K-slots k = 16s + 8hi + j (j<8) of step s
    const int xrd0 = G::WSLOT + wave * G::XB +
                     ((G::XG == 2) ? (i31 * 64 + (((2 * hi) ^ ((i31 >> 2) & 3)) * 16)) : (i31 * 32 + ((hi ^ ((i31 >> 3) & 1)) * 16)));
    const int xrd1 = G::WSLOT + wave * G::XB + i31 * 64 + (((2 * hi + 1) ^ ((i31 >> 2) & 3)) * 16);   // fp32 only
    // x operand of one step: read this wave's own tile from the ring slot and convert (fp32 -> f16 hi/lo split)
    float xv[8];
    f16x8 xh, xl;
    auto load_x = [&](int s, float (&v)[8], f16x8& h8, f16x8& l8) {
        const char* slot = smem + (s % G::NB) * G::SLOT;
        if constexpr (XDT == ACMIL_DTYPE_F32) {
            const f32x4 a0 = *(const f32x4*)(slot + xrd0), a1 = *(const f32x4*)(slot + xrd1);
            v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3];
            v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
        } else if constexpr (XDT == ACMIL_DTYPE_F16) {
            h8 = *(const f16x8*)(slot + xrd0);
            if constexpr (F32M) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (float)h8[j];
            }
        } else {
            const u32x4 w = *(const u32x4*)(slot + xrd0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = __builtin_bit_cast(float, w[e] << 16);
                v[2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xffff0000u);
            }
        }
        if constexpr (!F32M && XDT != ACMIL_DTYPE_F16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const _Float16 h16 = (_Float16)v[j];
                h8[j] = h16;
                if constexpr (XLO) l8[j] = (_Float16)(v[j] - (float)h16);
            }
        }
    };
    if constexpr (G::ALT) {
        // steps 0 and 2 were issued by waves 0-3, step 1 by waves 4-7; x(0) of BOTH waves of a pair was copied by the even
        // group: it waits for it (newer: W(0) and all of step 2), then a barrier publishes it to the partner
        static_assert(G::PD == 3, "alternating issue assumes a prefetch distance of 3 steps");
        if ((wave >> 2) == 0) ga_wait_vm<G::WGA + G::NA>();
        __builtin_amdgcn_s_barrier();
    } else {
        // x(0) is wave-private: only this wave's own DMA has to land (no barrier)
        ga_wait_vm<G::WG1 + (G::PD - 1) * G::N1>();
    }
    load_x(0, xv, xh, xl);
    // Drain the scalar-load / LDS counter once, with the BUILTIN (which hipcc's wait-count pass models): otherwise the
    // kernarg s_loads still pending at the loop header make lgkmcnt "out of order" in the compiler's model and every
    // first MFMA of a step waits lgkmcnt(0) -- for all 16 fragment reads instead of the first one.
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) only
    constexpr int WAIT1 = G::WG1 + (G::PD - 2) * G::N1;    // newer than x(s+1): W(s+1) and all of steps s+2 .. s+PD-1
    constexpr int WAIT2 = (G::PD - 1) * G::N1;             // GEMM2 only needs W(s): all of steps s+1 .. s+PD-1 may be in flight
    for (int s = 0; s < S1; ++s) {
        // wait for W(s) and for this wave's x(s+1); the rest of step s+1 and all of step s+2 stay in flight
        if constexpr (G::ALT) {
            // the group that issued step s (and, one step ago, step s+2) needs all of step s: its step s+2 stays in flight;
            // the other group issued step s+1 (x first) and needs x(s+1): its W(s+1) stays in flight
            if ((wave >> 2) == (s & 1)) ga_wait_vm<G::NA>(); else ga_wait_vm<G::WGA>();
        } else ga_wait_vm<WAIT1>();
        __builtin_amdgcn_s_barrier();
        if constexpr (G::ALT) issue_step(s + G::PD);   // issuing wave: DMA first, while its partner owns the matrix pipe
        // (non-ALT: the next ring slot's LDS-DMA is issued from inside the MFMA stream below: its issue cost -- address
        //  VALU, M0 writes, ~100 cycles per instruction -- then overlaps matrix-core work instead of delaying it)
        const char* slot = smem + (s % G::NB) * G::SLOT;
        float xvn[8];
        f16x8 xhn, xln;
        if constexpr (F32M) {
            const f32x4* wb = (const f32x4*)slot + lane;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1 && !G::ALT) issue_step(s + G::PD);
                if (half == 1) load_x(s + 1, xvn, xhn, xln);   // next step's x (dummy after the last): overlaps this step's MFMAs
                f32x4 wf[ND];
#pragma unroll
                for (int d = 0; d < ND; ++d) wf[d] = wb[(d * 2 + half) * 64];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        acc1[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[d][q], xv[4 * half + q], acc1[d], 0, 0, 0);
            }
        } else {
            const f16x8* wb = (const f16x8*)slot + lane;
            // All A fragments of the step are requested up front, "hi" parts first, and the MFMAs consume them in
            // request order, so the compiler's lgkmcnt waits count down (15, 14, ...) and only the FIRST fragment's
            // LDS latency is exposed after the barrier.  sched_barrier(0) keeps hipcc from re-clustering the phases.
            f16x8 wh[ND], wl[SPLIT ? ND : 1];
#pragma unroll
            for (int d = 0; d < ND; ++d) wh[d] = wb[d * 64];               // packed: ND "hi" rows, then ND "lo" rows
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[d], xh, acc1[d], 0, 0, 0);
                if (SPLIT && d == 1) {                                       // "lo" requests go out under the first MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int d2 = 0; d2 < ND; ++d2) wl[d2] = wb[(ND + d2) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!G::ALT) issue_step(s + G::PD);   // LDS-DMA issue in the shadow of the MFMAs just queued
            if constexpr (SPLIT) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < ND; ++d) acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[d], xh, acc1[d], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            load_x(s + 1, xvn, xhn, xln);                // next step's x (dummy after the last): read + f16 split
            if constexpr (XLO) {
#pragma unroll
                for (int d = 0; d < ND; ++d) acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[d], xl, acc1[d], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = xvn[j];
        xh = xhn; xl = xln;
    }

    // =========================================================== relu
    // acc1[d][r] now holds h[patch = lane&31][feature = 32d + mfma32_row(r, hi)]
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[d][r] = fmaxf(acc1[d][r], 0.0f);
    // keep the relu (and the f16 split below) HERE: LLVM otherwise sinks them into the GEMM2 steps that use
    // them, the old accumulator tuples stay live next to the new values and the kernel spills hundreds of
    // registers.  An empty volatile asm makes each value opaque at this point.
#pragma unroll
    for (int d = 0; d < ND; ++d) asm volatile("" : "+v"(acc1[d]));

    f16x8 hh[F32M ? 1 : ND][2], hl[SPLIT ? ND : 1][2];
    if constexpr (!F32M) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = acc1[d][8 * e + j];
                    const _Float16 h16 = (_Float16)v;
                    hh[d][e][j] = h16;
                    if constexpr (SPLIT) hl[d][e][j] = (_Float16)(v - (float)h16);
                }
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                asm volatile("" : "+v"(hh[d][e]));
                if constexpr (SPLIT) asm volatile("" : "+v"(hl[d][e]));
            }
    }

    // =========================================================== GEMM2 (four unit blocks) + gate + scores
    const float* tabf = (const float*)(smem + G::TAB_OFF);
    float sc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) sc[k] = 0.0f;

    // (rolled on purpose: one copy of the block body keeps hipcc's scheduler from hoisting loads of later blocks)
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        // unit block g = units 32g..32g+31: accumulator tile al = 0 tanh branch, 1 sigmoid branch;
        // register r of lane half hi is unit 32g + mfma32_row(r,hi); init = bias
        f32x16 acc2[2];
#pragma unroll
        for (int al = 0; al < 2; ++al)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                int boff = 32 * g + 8 * rq + 4 * hi;
                asm volatile("" : "+v"(boff) : "v"(sc[0]));   // order after the previous block's epilogue (see below)
                const f32x4 b = *(const f32x4*)(tabf + al * GA_DA + boff);
                acc2[al][4 * rq + 0] = b[0]; acc2[al][4 * rq + 1] = b[1];
                acc2[al][4 * rq + 2] = b[2]; acc2[al][4 * rq + 3] = b[3];
            }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int s = S1 + g * 4 + st;
            if constexpr (G::ALT) {
                // S1 is even (D % 64 == 0), so step parity == (g*4 + st) parity: the group that issued step s drains it
                if ((wave >> 2) == (st & 1)) ga_wait_vm<G::NA>();
            } else ga_wait_vm<WAIT2>();
            __builtin_amdgcn_s_barrier();
            if constexpr (G::ALT) issue_step(s + G::PD);
            const char* slot = smem + (s % G::NB) * G::SLOT;
#pragma unroll
            for (int dd = 0; dd < G::DD; ++dd) {
                const int d = G::DD * st + dd;
                if (dd == G::DD - 1 && !G::ALT) issue_step(s + G::PD);
                if constexpr (F32M) {
                    const f32x4* wb = (const f32x4*)slot + lane;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 w0 = wb[((dd * 4 + r4) * 2 + 0) * 64], w1 = wb[((dd * 4 + r4) * 2 + 1) * 64];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[q], acc1[d][4 * r4 + q], acc2[0], 0, 0, 0);
                            acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[q], acc1[d][4 * r4 + q], acc2[1], 0, 0, 0);
                        }
                    }
                } else {
                    const f16x8* wb = (const f16x8*)slot + lane;
                    // same discipline as GEMM1: request the 4 "hi" then the 4 "lo" fragments of this h tile, consume in order
                    f16x8 wh[4], wl[SPLIT ? 4 : 1];
#pragma unroll
                    for (int t = 0; t < 4; ++t) wh[t] = wb[((dd * PARTS + 0) * 4 + t) * 64];      // t = e*2 + al
                    if constexpr (SPLIT) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) wl[t] = wb[((dd * PARTS + 1) * 4 + t) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc2[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], hh[d][t >> 1], acc2[t & 1], 0, 0, 0);
                    if constexpr (SPLIT) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t], hh[d][t >> 1], acc2[t & 1], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], hl[d][t >> 1], acc2[t & 1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // gate + partial scores for the 32 units of this block (this lane: 16 of them)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            int ubase = 32 * g + 8 * rq + 4 * hi;
            float gate[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gate[q] = ga_tanh(acc2[0][4 * rq + q]) * ga_sigmoid(acc2[1][4 * rq + q]);
            // hipcc otherwise hoists every table read of the epilogue (80+ registers) above the last GEMM2 step and
            // spills them at once; tying the address to the gate value keeps each read next to its use.
            asm volatile("" : "+v"(ubase) : "v"(gate[3]));
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const f32x4 w = *(const f32x4*)(tabf + (2 + k) * GA_DA + ubase);
                sc[k] = fmaf(gate[0], w[0], sc[k]); sc[k] = fmaf(gate[1], w[1], sc[k]);
                sc[k] = fmaf(gate[2], w[2], sc[k]); sc[k] = fmaf(gate[3], w[3], sc[k]);
            }
        }
    }

    const float* bwp = (const float*)(a.packed + L.bw_off);
    float smax[KP], lsum[KP], pe[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        sc[k] += __shfl_xor(sc[k], 32);      // the other lane-half holds the other 64 attention units
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

    // =========================================================== attention-weighted sum  sum_n p[k][n] h[n][:]
    ga_wait_vm<0>();  // the clamped tail DMAs still target the ring: drain them before it is reused
    __syncthreads();  // every wave is done with the ring; region 0 becomes the pooling tiles
    float* pool = (float*)(smem + wave * G::POOLW);
    float* pl = (float*)(smem + G::PL_OFF) + (size_t)wave * KP * 32;   // [KP][32 m]
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
                float hv;
                if constexpr (F32M) hv = acc1[2 * c + dl][r];
                else if constexpr (SPLIT) hv = (float)hh[2 * c + dl][r >> 3][r & 7] + (float)hl[2 * c + dl][r >> 3][r & 7];
                else hv = (float)hh[2 * c + dl][r >> 3][r & 7];
                pool[(dl * 32 + mfma32_row(r, hi)) * 36 + i31] = hv;
            }
        __builtin_amdgcn_wave_barrier();
        const f32x4* prow = (const f32x4*)(pool + lane * 36);     // lane = feature 64c + lane
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

    // =========================================================== combine the 8 waves, publish the partial
    __builtin_amdgcn_wave_barrier();
    constexpr int PS = 2 + Di;
    static_assert(KP * PS * 4 <= G::POOLW, "combine record must fit the wave's pooling tile");
    float* comb = pool;  // overlays this wave's (now dead) pooling tile
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
}

// launcher for one (ND, KP, MODE, XDT) family; pool=true -> eval variant, else the h-saving score pass.
// waves = 8: one 256-patch workgroup per CU; waves = 4: two independent 128-patch workgroups per CU whose
// phases drift apart, so one's VALU/LDS epilogues overlap the other's MFMA steps.
template <int ND, int KP, int MODE, int XDT, int WAVES>
int ga_launch_fwd_w(const GaFwdArgs& a, bool pool, hipStream_t st) {
    using G = GaGeom<ND, KP, MODE, XDT, WAVES>;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    static_assert(WAVES == 8 || 2 * G::LDS <= 160 * 1024, "two 4-wave workgroups must fit one CU");
    const dim3 grid(a.tile_start[a.nbags]), block(64 * WAVES);
    void (*kern)(GaFwdArgs) = pool ? ga_fwd_kernel<ND, KP, MODE, XDT, WAVES, true, false>
                                   : ga_fwd_kernel<ND, KP, MODE, XDT, WAVES, false, true>;
    static bool attr_set[16] = {};      // per DEVICE, once (a process may drive several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ACMIL_ERR_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)ga_fwd_kernel<ND, KP, MODE, XDT, WAVES, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)ga_fwd_kernel<ND, KP, MODE, XDT, WAVES, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
            return ACMIL_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, grid, block, G::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

template <int ND, int KP, int MODE, int XDT>
int ga_launch_fwd(const GaFwdArgs& a, bool pool, hipStream_t st) {
    return a.waves == 4 ? ga_launch_fwd_w<ND, KP, MODE, XDT, 4>(a, pool, st) : ga_launch_fwd_w<ND, KP, MODE, XDT, 8>(a, pool, st);
}
