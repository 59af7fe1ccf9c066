// ga_forward_kernel.h -- fused forward of ACMIL's gated-attention aggregation for ONE bag on MI355X (gfx950).
//
// Replaces (reference file:line, /root/reference):
//   DimReduction.forward        architecture/network.py:49-57      h = relu(x W1^T)
//   Attention_Gated.forward     architecture/transformer.py:259-267 A = ((tanh(hWv^T+bv)*sigmoid(hWu^T+bu))Ww^T+bw)^T
//   ACMIL_GA.forward            architecture/transformer.py:322-324 softmax over N, P h   (heads: ga_forward.hip)
//
// Design (one pass over x, no intermediate ever touches HBM):
//   * grid = ceil(N/128) workgroups x 4 waves; every wave owns 32 consecutive patches end to end, so there
//     is no inter-wave data dependence in the GEMM chain -- the workgroup only shares the weight stream.
//   * both GEMMs are computed TRANSPOSED (D = W * x^T): patches live on the MFMA column index (= lane&31),
//     output features on the accumulator registers.  An accumulator register file in that layout IS the
//     B operand of the next MFMA chain (K index = register index, lane = column), so relu(h) feeds GEMM2
//     straight from registers; the K-slot permutation this implies is folded into the weight packing.
//   * weights arrive as a pre-packed fragment stream (ga_pack.hip): staged into LDS by linear
//     global_load_lds (LDS-DMA, no VGPR round trip), double buffered, one barrier per stage; A operands are
//     read with lane-linear conflict-free ds_read_b128.
//   * x is read exactly once, straight from HBM into the B operand registers (each lane 128 contiguous
//     bytes per 64-wide K macro-step, prefetched one macro-step ahead).
//   * GEMM2 runs in two groups of 64 attention units (4 MFMA tiles: tanh/sigmoid branch x 2) so only 64
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

struct GaFwdArgs {
    const void* x;
    const char* packed;
    float* A_out;    // [K,N] or null
    float* part;     // workspace partials [tiles][K][2+Di]
    float* h_save;   // [N,Di] or null
    int N;
    GaLayout L;
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int ROWS>
__device__ __forceinline__ void ga_stage_copy(const char* gsrc, char* lbuf, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < ROWS / GA_WAVES; ++r) {
        const int row = r * GA_WAVES + wave;
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + (size_t)row * GA_FRAG_ROW + lane * 16),
                                         (lptr_t)(lbuf + row * GA_FRAG_ROW), 16, 0, 0);
    }
}

__device__ __forceinline__ void ga_sync_stage() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- load one 64-wide K macro-step of this lane's patch row: 32 consecutive elements -> fp32 registers
template <int XDT>
__device__ __forceinline__ void ga_load_x(const void* xrow, int t, f32x4 (&dst)[8]) {
    if constexpr (XDT == ACMIL_DTYPE_F32) {
        const f32x4* p = (const f32x4*)((const float*)xrow + 64 * t);
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q] = __builtin_nontemporal_load(p + q);
    } else {
        const u32x4* p = (const u32x4*)((const uint16_t*)xrow + 64 * t);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 w = __builtin_nontemporal_load(p + q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo, hi;
                if constexpr (XDT == ACMIL_DTYPE_F16) {
                    lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[e] & 0xffffu));
                    hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[e] >> 16));
                } else {
                    lo = __builtin_bit_cast(float, w[e] << 16);
                    hi = __builtin_bit_cast(float, w[e] & 0xffff0000u);
                }
                dst[2 * q + (e >> 1)][2 * (e & 1) + 0] = lo;
                dst[2 * q + (e >> 1)][2 * (e & 1) + 1] = hi;
            }
        }
    }
}

template <int ND, int KP, int MODE>
struct GaGeom {
    static constexpr int SPM = (MODE == ACMIL_MODE_F16) ? 1 : 2;     // stages per 64-wide macro-step
    static constexpr int R1 = 4 * ND;                                // fragment rows per GEMM1 stage
    static constexpr int R2 = (MODE == ACMIL_MODE_F16) ? 8 : 16;     // rows per GEMM2 stage (group g, h tile d)
    static constexpr int STAGE_BYTES = (R1 > R2 ? R1 : R2) * GA_FRAG_ROW;
    static constexpr int POOLW = 128 * 33 * 4;                       // wave-private [128 di][32 m (+1 pad)] fp32
    static constexpr int REGION0 = (2 * STAGE_BYTES > GA_WAVES * POOLW) ? 2 * STAGE_BYTES : GA_WAVES * POOLW;
    static constexpr int TAB_OFF = REGION0;                          // bv[128], bu[128], Ww[KP][128]
    static constexpr int TAB_BYTES = (2 + KP) * GA_DA * 4;
    static constexpr int KP4 = (KP + 3) / 4 * 4;
    static constexpr int PL_OFF = TAB_OFF + TAB_BYTES;               // p_lds: [wave][32 m][KP4] fp32
    static constexpr int PL_BYTES = GA_WAVES * 32 * KP4 * 4;
    static constexpr int LDS = PL_OFF + PL_BYTES;
};

template <int ND, int KP, int MODE, int XDT, bool POOL, bool SAVEH>
__global__ __launch_bounds__(256) void ga_fwd_kernel(GaFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = GaGeom<ND, KP, MODE>;
    constexpr bool F32M = (MODE == ACMIL_MODE_F32);
    constexpr bool SPLIT = (MODE == ACMIL_MODE_F16X3);
    constexpr bool XLO = SPLIT && (XDT != ACMIL_DTYPE_F16);   // fp16 bags are exact in the hi part
    constexpr int SPM = G::SPM, R1 = G::R1, R2 = G::R2, STAGE_BYTES = G::STAGE_BYTES, POOLW = G::POOLW;
    constexpr int KP4 = G::KP4;
    constexpr int NCH = ND / 4;                                // pooling chunks of 128 features
    constexpr int PARTS = SPLIT ? 2 : 1;

    const GaLayout& L = a.L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i31 = lane & 31, hi = lane >> 5;
    const int N = a.N, D = L.D, K = L.K;
    constexpr int Di = ND * 32;
    const int T = D / 64;
    const int m0 = blockIdx.x * GA_ROWS_PER_WG + wave * 32;
    const int row = m0 + i31;
    const bool valid = row < N;
    const int rowc = valid ? row : N - 1;
    const size_t xelem = (XDT == ACMIL_DTYPE_F32) ? 4 : 2;
    const char* xrow = (const char*)a.x + ((size_t)rowc * D + 32 * hi) * xelem;

    const char* g1 = a.packed + L.g1_off;
    const char* g2 = a.packed + L.g2_off;
    const int S1 = T * SPM;

    auto issue_stage = [&](int s) {
        char* buf = smem + (s & 1) * STAGE_BYTES;
        if (s < S1) ga_stage_copy<R1>(g1 + (size_t)s * R1 * GA_FRAG_ROW, buf, wave, lane);
        else if (s < S1 + 2 * ND) ga_stage_copy<R2>(g2 + (size_t)(s - S1) * R2 * GA_FRAG_ROW, buf, wave, lane);
    };

    // epilogue vectors bv, bu, Ww -> LDS (rows K..KP-1 of Ww zero); visible after the first stage barrier
    {
        const float* src = (const float*)(a.packed + L.tab_off);
        float* dst = (float*)(smem + G::TAB_OFF);
        for (int e = tid; e < (2 + KP) * GA_DA; e += 256) dst[e] = (e < (2 + K) * GA_DA) ? src[e] : 0.0f;
    }

    f32x16 acc1[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[d][r] = 0.0f;

    f32x4 xr[8], xn[8];
    ga_load_x<XDT>(xrow, 0, xr);
#pragma unroll
    for (int q = 0; q < 8; ++q) xn[q] = xr[q];
    issue_stage(0);

    // =========================================================== GEMM1: h^T = W1 * x^T
    for (int t = 0; t < T; ++t) {
        f16x8 xh[4], xl[4];
        if constexpr (!F32M) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = xr[2 * s + (j >> 2)][j & 3];
                    const _Float16 h16 = (_Float16)v;
                    xh[s][j] = h16;
                    if constexpr (XLO) xl[s][j] = (_Float16)(v - (float)h16);
                }
        }
#pragma unroll
        for (int half = 0; half < SPM; ++half) {
            const int s = t * SPM + half;
            ga_sync_stage();
            issue_stage(s + 1);
            if (half == 0 && t + 1 < T) ga_load_x<XDT>(xrow, t + 1, xn);
            const char* buf = smem + (s & 1) * STAGE_BYTES;
            if constexpr (F32M) {
                const f32x4* wb = (const f32x4*)buf + lane;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 wf[ND];
#pragma unroll
                    for (int d = 0; d < ND; ++d) wf[d] = wb[(g * ND + d) * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int d = 0; d < ND; ++d)
                            acc1[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[d][q], xr[4 * half + g][q], acc1[d], 0, 0, 0);
                }
            } else {
                const f16x8* wb = (const f16x8*)buf + lane;
                constexpr int KS = 4 / SPM;       // 16-wide k-steps per stage
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int s4 = half * KS + ks;
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        const f16x8 wh = wb[((ks * ND + d) * PARTS + 0) * 64];
                        acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[s4], acc1[d], 0, 0, 0);
                        if constexpr (SPLIT) {
                            const f16x8 wl = wb[((ks * ND + d) * PARTS + 1) * 64];
                            acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[s4], acc1[d], 0, 0, 0);
                            if constexpr (XLO) acc1[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[s4], acc1[d], 0, 0, 0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) xr[q] = xn[q];
    }

    // =========================================================== relu
    // acc1[d][r] now holds h[patch = lane&31][feature = 32d + mfma32_row(r, hi)]
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[d][r] = fmaxf(acc1[d][r], 0.0f);

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
    }

    // =========================================================== GEMM2 (two unit groups) + gate + scores
    const float* tabf = (const float*)(smem + G::TAB_OFF);
    float sc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) sc[k] = 0.0f;

#pragma unroll
    for (int g = 0; g < 2; ++g) {
        // accumulator tiles: al = 0..3 -> (tanh, sigmoid) branch of unit pair-block p = 2g + (al>>1);
        // register r of lane half hi is unit 32p + mfma32_row(r,hi); init = bias
        f32x16 acc2[4];
#pragma unroll
        for (int al = 0; al < 4; ++al)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int p = 2 * g + (al >> 1);
                const f32x4 b = *(const f32x4*)(tabf + (al & 1) * GA_DA + 32 * p + 8 * rq + 4 * hi);
                acc2[al][4 * rq + 0] = b[0]; acc2[al][4 * rq + 1] = b[1];
                acc2[al][4 * rq + 2] = b[2]; acc2[al][4 * rq + 3] = b[3];
            }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int s = S1 + g * ND + d;
            ga_sync_stage();
            issue_stage(s + 1);
            const char* buf = smem + (s & 1) * STAGE_BYTES;
            if constexpr (F32M) {
                const f32x4* wb = (const f32x4*)buf + lane;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 wf[4];
#pragma unroll
                    for (int al = 0; al < 4; ++al) wf[al] = wb[(r4 * 4 + al) * 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int al = 0; al < 4; ++al)
                            acc2[al] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[al][q], acc1[d][4 * r4 + q], acc2[al], 0, 0, 0);
                }
            } else {
                const f16x8* wb = (const f16x8*)buf + lane;
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int al = 0; al < 4; ++al) {
                        const f16x8 wh = wb[((e * 4 + al) * PARTS + 0) * 64];
                        acc2[al] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hh[d][e], acc2[al], 0, 0, 0);
                        if constexpr (SPLIT) {
                            const f16x8 wl = wb[((e * 4 + al) * PARTS + 1) * 64];
                            acc2[al] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, hh[d][e], acc2[al], 0, 0, 0);
                            acc2[al] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, hl[d][e], acc2[al], 0, 0, 0);
                        }
                    }
            }
        }
        // gate + partial scores for the 64 units of this group (this lane: 32 of them)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int ubase = 32 * (2 * g + pl) + 8 * rq + 4 * hi;
                float gate[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    gate[q] = ga_tanh(acc2[2 * pl][4 * rq + q]) * ga_sigmoid(acc2[2 * pl + 1][4 * rq + q]);
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
        if (a.A_out && valid && hi == 0 && k < K) a.A_out[(size_t)k * N + row] = sc[k];
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
    __syncthreads();  // every wave is done with the stage buffers; region 0 becomes the pooling tiles
    float* pool = (float*)(smem + wave * POOLW);
    float* pl = (float*)(smem + G::PL_OFF) + (size_t)wave * 32 * KP4;
    if (POOL && hi == 0) {
#pragma unroll
        for (int k = 0; k < KP4; ++k) pl[i31 * KP4 + k] = (k < KP) ? pe[k < KP ? k : 0] : 0.0f;
    }
    float pacc[NCH][2][KP];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int k = 0; k < KP; ++k) pacc[c][ps][k] = 0.0f;

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dl = 0; dl < 4; ++dl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float hv;
                if constexpr (F32M) hv = acc1[4 * c + dl][r];
                else if constexpr (SPLIT) hv = (float)hh[4 * c + dl][r >> 3][r & 7] + (float)hl[4 * c + dl][r >> 3][r & 7];
                else hv = (float)hh[4 * c + dl][r >> 3][r & 7];
                pool[(dl * 32 + mfma32_row(r, hi)) * 33 + i31] = hv;
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int dil = 64 * ps + lane;
            const float* prow = pool + dil * 33;
#pragma unroll 4
            for (int m = 0; m < 32; ++m) {
                const float hv = prow[m];
                if constexpr (SAVEH) {
                    if (m0 + m < N) a.h_save[(size_t)(m0 + m) * Di + 128 * c + dil] = hv;
                }
                if constexpr (POOL) {
                    float e[KP4];
#pragma unroll
                    for (int f = 0; f < KP4 / 4; ++f) {
                        const f32x4 v = *(const f32x4*)(pl + m * KP4 + 4 * f);
                        e[4 * f + 0] = v[0]; e[4 * f + 1] = v[1]; e[4 * f + 2] = v[2]; e[4 * f + 3] = v[3];
                    }
#pragma unroll
                    for (int k = 0; k < KP; ++k) pacc[c][ps][k] = fmaf(e[k], hv, pacc[c][ps][k]);
                }
            }
        }
    }
    if constexpr (!POOL) return;

    // =========================================================== combine the 4 waves, publish the partial
    __builtin_amdgcn_wave_barrier();
    const int PS = 2 + Di;
    float* comb = pool;  // overlays this wave's (now dead) pooling tile
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K) {
            if (lane == 0) { comb[k * PS + 0] = smax[k]; comb[k * PS + 1] = lsum[k]; }
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) comb[k * PS + 2 + 128 * c + 64 * ps + lane] = pacc[c][ps][k];
        }
    }
    __syncthreads();
    float* out = a.part + (size_t)blockIdx.x * K * PS;
    for (int k = 0; k < K; ++k) {
        float mw[GA_WAVES], M = -INFINITY;
#pragma unroll
        for (int w = 0; w < GA_WAVES; ++w) {
            mw[w] = ((const float*)(smem + w * POOLW))[k * PS + 0];
            M = fmaxf(M, mw[w]);
        }
        float fw[GA_WAVES];
#pragma unroll
        for (int w = 0; w < GA_WAVES; ++w) fw[w] = (mw[w] == -INFINITY) ? 0.0f : __expf(mw[w] - M);
        for (int e = tid; e < PS; e += 256) {
            float v;
            if (e == 0) v = M;
            else {
                v = 0.0f;
#pragma unroll
                for (int w = 0; w < GA_WAVES; ++w) v = fmaf(fw[w], ((const float*)(smem + w * POOLW))[k * PS + e], v);
            }
            out[k * PS + e] = v;
        }
    }
}

// launcher for one (ND, KP, MODE, XDT) family; pool=true -> eval variant, else the h-saving score pass
template <int ND, int KP, int MODE, int XDT>
int ga_launch_fwd(const GaFwdArgs& a, bool pool, hipStream_t st) {
    using G = GaGeom<ND, KP, MODE>;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    const dim3 grid(ga_num_tiles(a.N)), block(256);
    void (*kern)(GaFwdArgs) = pool ? ga_fwd_kernel<ND, KP, MODE, XDT, true, false>
                                   : ga_fwd_kernel<ND, KP, MODE, XDT, false, true>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
        return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, grid, block, G::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
