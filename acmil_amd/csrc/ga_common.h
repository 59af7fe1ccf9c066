// ga_common.h -- shared definitions for the gated-attention (GA) kernels of libacmil_hip.so.
// gfx950 / CDNA4 only: wave64, v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_f16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "acmil_hip.h"
#include "ab_knobs.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GA_DA 128           // attention hidden width, fixed by the reference (transformer.py:292)
#define GA_WAVES 8          // waves per workgroup (2 per SIMD); each wave owns 32 consecutive patches
#define GA_ROWS_PER_WG 256  // 8 waves x 32 patches (fused forward tile)
#define GA_POOL_ROWS 128    // rows per workgroup of the h-streaming pooling kernel (ga_train.hip)
#define GA_WT_KX 288        // K of the backward's third product (2 Da + 16 extension slots), padded to 32-wide steps (ga_bwd_tile.hip)
#define GA_FRAG_ROW 1024    // one "fragment row" of the packed weight stream: 64 lanes x 16 B

// Row index inside a 32x32 MFMA C/D tile held by (register r, lane-half hi): the gfx950 C/D map is
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)   (cdna_hip_programming.md section 3).
__host__ __device__ static inline int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Fragment-ordered hi / lo plane pairs (operands of ga_bwd_tile.hip; written by ga_pack.hip and, for the d_afeat columns, by the
// step's tail kernel): element (row, k) of a [rows][16 nks] matrix lives where the MFMA B fragment of 32-row tile row / 32 and
// 16-wide K step k / 16 wants it -- lane = 32 * ((k >> 3) & 1) + (row & 31) holds 8 consecutive k as 16 bytes -- so a wave's
// fragment (1 KB) and its hi / lo pair (2 KB) are contiguous: the kernel loads them global -> registers with fully used cache lines.
__host__ __device__ static inline size_t ga_frag_off(int row, int k, int nks, int plane) {
    return ((((size_t)(row >> 5) * nks + (k >> 4)) * 2 + plane) * 64 + ((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
}

// ---------------------------------------------------------------------------------------------------
// Packed-weight buffer layout (built by ga_pack.hip, consumed by ga_forward.hip).  All offsets in bytes.
//
//  g1 : GEMM1 operand stream, (D/16) steps x 2 x ND fragment rows (ND = Di/32 output tiles)
//  g2 : GEMM2 operand stream, 2 unit-groups x ND x 16 fragment rows   (both: half as many rows in MODE_F16)
//  tab: bv[128], bu[128], Ww[K][128]  raw fp32 (the C/D register quad (4rq..4rq+3) of tile pair p in lane
//       half hi covers the 4 CONSECUTIVE units 32p + 8rq + 4hi + {0..3}, so the epilogue reads plain float4s)
//  bw : ACMIL_MAX_TOKENS floats (bw[K], zero padded)
//  heads: Wc [K][C][Di], bc [K][C], Ws [C][Di], bs [C]   (raw copies, fp32)
//  wcat : [Wv; Wu] [2 Da][Di], bcat [bv; bu] [2 Da], wcatT = [Wv; Wu]^T [Di][2 Da]  (raw fp32; read by the backward of a training step)
// ---------------------------------------------------------------------------------------------------
struct GaLayout {
    int D, Di, K, C, ND, mode;
    size_t g1_off, g1_rows, g2_off, g2_rows, tab_off, bw_off, wc_off, bc_off, ws_off, bs_off, wcat_off, bcat_off, wcatT_off, w16_off, wT16_off, total;
};

__host__ __device__ static inline GaLayout ga_layout(int D, int Di, int K, int C, int mode) {
    GaLayout L;
    L.D = D; L.Di = Di; L.K = K; L.C = C; L.mode = mode;
    L.ND = Di / 32;
    size_t off = 0;
    const int half = (mode == ACMIL_MODE_F16) ? 2 : 1;  // single-pass f16 carries no "lo" rows
    L.g1_off = off; L.g1_rows = (size_t)(D / 64) * 8 * L.ND / half; off += L.g1_rows * GA_FRAG_ROW;
    L.g2_off = off; L.g2_rows = (size_t)L.ND * 32 / half;            off += L.g2_rows * GA_FRAG_ROW;
    L.tab_off = off; off += (size_t)(2 + K) * GA_DA * 4;
    L.bw_off = off;  off += ACMIL_MAX_TOKENS * 4;
    L.wc_off = off;  off += (size_t)K * C * Di * 4;
    L.bc_off = off;  off += (size_t)((K * C + 3) / 4) * 16;
    L.ws_off = off;  off += (size_t)C * Di * 4;
    L.bs_off = off;  off += (size_t)((C + 3) / 4) * 16;
    off = (off + 255) & ~(size_t)255;
    L.wcat_off = off; off += (size_t)2 * GA_DA * Di * 4;     // [Wv; Wu] as one [2 Da, Di] matrix, [bv; bu]: the backward's
    L.bcat_off = off; off += (size_t)2 * GA_DA * 4;          // single-GEMM operands (ga_backward.hip)
    L.wcatT_off = off; off += (size_t)2 * GA_DA * Di * 4;    // [Wv; Wu]^T [Di][2 Da]: K-contiguous operand of the dpre product
    L.w16_off = off;   off += (size_t)2 * 2 * GA_DA * Di * 2;    // f16 hi / lo of [Wv; Wu] [2 Da][Di], fragment order (ga_frag_off) } operands of the fused
    L.wT16_off = off;  off += (size_t)2 * Di * GA_WT_KX * 2;     // bf16 hi / lo of [[Wv;Wu]^T | d_afeat^T | 0] [Di][288], same order } backward tile kernel
    L.total = (off + 255) & ~(size_t)255;
    return L;
}

// Workspace layout of the forward: per-workgroup online-softmax partials
//   part[tile][k][0] = running max m, [1] = sum l, [2 .. 2+Di) = sum_n exp(s-m) h[n][:]
__host__ __device__ static inline size_t ga_part_stride(int Di) { return (size_t)(2 + Di); }
__host__ __device__ static inline int ga_num_tiles(int N) { return (N + GA_ROWS_PER_WG - 1) / GA_ROWS_PER_WG; }
__host__ __device__ static inline int ga_pool_tiles(int N) { return (N + GA_POOL_ROWS - 1) / GA_POOL_ROWS; }

// padded branch count the K-templated kernels are instantiated for
__host__ __device__ static inline int ga_kp(int K) { return (K <= 1) ? 1 : (K <= 5) ? 5 : (K <= 8) ? 8 : 16; }

static inline int ga_check_dims(int D, int Di, int Da, int K, int C) {
    if (Da != GA_DA) return ACMIL_ERR_UNSUPPORTED;
    if (D <= 0 || Di <= 0 || K <= 0 || C <= 0) return ACMIL_ERR_SHAPE;
    if (D % 64 != 0 || Di % 128 != 0) return ACMIL_ERR_SHAPE;
    if (K > ACMIL_MAX_TOKENS || C > ACMIL_MAX_CLASSES) return ACMIL_ERR_UNSUPPORTED;
    return ACMIL_OK;
}

// fast transcendental forms: one v_exp_f32 + one v_rcp_f32 each; absolute error ~1e-7, which is the
// fp32 round-off class of the GEMM that feeds them.
__device__ static inline float ga_sigmoid(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
__device__ static inline float ga_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * v)); }

// Control block = the FIRST GA_CTRL_BYTES of every GA workspace (32-bit words): 0 tile counter of the persistent forward,
// 1 range status of the split-f16 arithmetic = result of the MOST RECENT launch on this workspace, 2 finished-workgroup
// counter, 3 range-flag accumulator of the running launch, 4.. arrival counters of the single-launch STKIM / tail kernels.
// The caller zeroes the block ONCE when it allocates the workspace; every kernel that counts in it leaves its counters at
// zero again (the last workgroup resets them), so no memset sits on the stream between launches.
#define GA_CTRL_BYTES 256

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also waits vmcnt(0),
// i.e. for every global load in flight -- which turns a "loads issued N steps ahead" pipeline into one full memory latency
// per step.  A ds_write of loaded data still waits for exactly its own source registers.
__device__ __forceinline__ void ga_lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Sum over the 64 lanes with DPP row operations (VALU, no LDS round trips; a __shfl_xor butterfly is 6 dependent ds_bpermutes):
// inclusive scan inside each row of 16 (row_shr 1, 2, 4, 8, missing sources read 0), row_bcast15 / row_bcast31 carry the row
// totals across, lane 63 holds the total, v_readlane broadcasts it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ga_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float ga_wave_sum(float v) {
    v = ga_dpp_add<0x111, 0xf>(v);
    v = ga_dpp_add<0x112, 0xf>(v);
    v = ga_dpp_add<0x114, 0xf>(v);
    v = ga_dpp_add<0x118, 0xf>(v);
    v = ga_dpp_add<0x142, 0xa>(v);
    v = ga_dpp_add<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float ga_readlane(float v, int l) {      // l must be wave-uniform (a constant after unrolling)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// ---- hi / lo splits of two fp32 values at a time (packed results: element 0 in the low half-word)
// f16: hi = rn_f16(x) (packed convert), lo = rn_f16(x - hi) by v_fma_mix{lo,hi}_f16 (f16 source * -1.0 + f32 source, ONE rounding
// to f16; x - hi is exact in fp32, so this equals the convert / subtract / convert sequence bit for bit): 3 VALU instructions
// for two elements instead of 8.  The s_nops cover the partial-register-write forwarding hazard.
__device__ __forceinline__ void ga_split_pair_f16(float x0, float x1, unsigned& hi_pk, unsigned& lo_pk) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
        : "=&v"(hi_pk), "=&v"(lo_pk) : "v"(x0), "v"(x1));
}
// hi only: round-to-nearest packed convert (a bf16 value is f16-exact above the f16 subnormals and rounds by at most 2^-25 inside
// them -- where the f16 image of the remainder is 0 anyway: the lo half of a bf16 value is identically zero, see ga_forward_kernel_v2.h)
__device__ __forceinline__ unsigned ga_cvt_pair_f16(float x0, float x1) {
    unsigned hi_pk;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi_pk) : "v"(x0), "v"(x1));
    return hi_pk;
}
// bf16: hi = rn_bf16(x), lo = rn_bf16(x - hi) with the PACKED hardware convert (the compiler's scalar casts cost one
// v_cvt_pk_bf16_f32 per element plus pack operations): 6 VALU instructions for two elements
__device__ __forceinline__ void ga_split_pair_bf16(float x0, float x1, unsigned& hi_pk, unsigned& lo_pk) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi_pk) : "v"(x0), "v"(x1));
    const float d0 = x0 - __builtin_bit_cast(float, hi_pk << 16);
    const float d1 = x1 - __builtin_bit_cast(float, hi_pk & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo_pk) : "v"(d0), "v"(d1));
}

// merge + heads (ga_forward.hip), shared by the fused forward and the masked pooling pass; the afeat scratch follows the partials
int ga_finish(const float* part, int tiles, const void* packed, const GaLayout& L, float* sub_preds,
              float* slide_pred, float* afeat, float* bag_feat, int has_bag_head, hipStream_t st);
