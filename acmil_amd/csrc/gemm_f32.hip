// gemm_f32.hip -- exact-fp32 matrix-core GEMM for gfx950 (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain,
// 157 TFLOP/s peak), the building block of the backward pass of the gated-attention path and of the
// TransMIL / Nystrom path.  Hand-written, no rocBLAS / hipBLASLt.
//
//   C[b] = act( alpha * op(A[b]) * op(B[b]) + bias + beta * C[b] )        b < batch
//   op(A) is M x K, op(B) is K x N, all row-major with leading dimensions; transA/transB select op = transpose.
//   B may be fp32 / fp16 / bf16 in memory (converted to fp32 on load; used for dW1 = dpre^T * x on fp16 bags).
//
// Tiling: 128 x 128 x 32 per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles, 64
// accumulator registers).  Both operand tiles are staged K-major in LDS ([k][m] / [k][n], +4 pad), so the
// one-float-per-lane MFMA fragments A[i = lane&31][k = lane>>5] are conflict-free ds_read_b32 whatever the
// transposition; global loads are float4 along the contiguous axis and register-prefetched one tile ahead.
// Small problems (the m x m products of the Moore-Penrose iteration: 8 x 192^3) use a 64 x 64 x 32 tile (one MFMA tile
// per wave) so that 72 instead of 32 workgroups share the work and a wave's serial MFMA chain is 4x shorter.
// Tall-K problems (weight gradients: K = number of patches) are split along K over gridDim.z into a
// workspace and reduced in a fixed order (deterministic), the epilogue then runs in the reduce kernel.
#include "ga_common.h"
#include "gemm_internal.h"
#include <string.h>

#define GM_BM 128
#define GM_BN 128
#define GM_BK 32


template <int BDT>
__device__ __forceinline__ float gm_ldb(const void* B, long long idx) {
    if constexpr (BDT == ACMIL_DTYPE_F32) return ((const float*)B)[idx];
    else if constexpr (BDT == ACMIL_DTYPE_F16) return (float)((const _Float16*)B)[idx];
    else return __builtin_bit_cast(float, (uint32_t)((const uint16_t*)B)[idx] << 16);
}

// 4 consecutive elements starting at idx (idx % 4 == 0 and the base 16-B aligned when vec) -> fp32
template <bool IS_B, int BDT>
__device__ __forceinline__ void gm_ld4(const void* P, long long idx, bool vec, int nvalid, float (&o)[4]) {
    constexpr bool HALF = IS_B && (BDT != ACMIL_DTYPE_F32);
    if (vec && nvalid == 4) {
        if constexpr (!HALF) {
            const f32x4 v = *(const f32x4*)((const float*)P + idx);
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        } else {
            const uint2 w = *(const uint2*)((const uint16_t*)P + idx);
            if constexpr (BDT == ACMIL_DTYPE_F16) {
                o[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.x & 0xffffu)); o[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.x >> 16));
                o[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.y & 0xffffu)); o[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w.y >> 16));
            } else {
                o[0] = __builtin_bit_cast(float, w.x << 16); o[1] = __builtin_bit_cast(float, w.x & 0xffff0000u);
                o[2] = __builtin_bit_cast(float, w.y << 16); o[3] = __builtin_bit_cast(float, w.y & 0xffff0000u);
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (q < nvalid) ? (IS_B ? gm_ldb<BDT>(P, idx + q) : ((const float*)P)[idx + q]) : 0.0f;
}

// stage one 128 x 32 (or 32 x 128) tile of a row-major matrix into registers; `contig_is_k` tells whether
// the contiguous memory axis is K (operand stored [rows][K]) or the M/N axis (operand stored [K][rows]).
// Each thread owns 4 groups of 4 elements that are contiguous in memory: one 16-B (8-B for 16-bit B) load
// when the matrix is vector-aligned (`vec`) and the group is interior, guarded scalar loads at the edges.
// G = groups of 4 elements per thread: 4 for the 128-row tile, 2 for the 64-row tile (ROWS = 32 G).
template <bool IS_B, int BDT, int G>
__device__ __forceinline__ void gm_load_tile(const void* P, int ld, bool contig_is_k, int r0, int k0, int R, int K,
                                             int kend, int tid, bool vec, float (&reg)[G][4]) {
    if (contig_is_k) {
        // thread -> (row = tid/8 + 32*i, 4 consecutive k at 4*(tid%8))
        const int k = k0 + 4 * (tid & 7);
        const int nv = kend - k < 0 ? 0 : (kend - k > 4 ? 4 : kend - k);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int r = r0 + (tid >> 3) + 32 * i;
            gm_ld4<IS_B, BDT>(P, (long long)r * ld + k, vec, r < R ? nv : 0, reg[i]);
        }
    } else {
        // thread -> (k = tid/(8G) + (32/G)*i, 4 consecutive rows at 4*(tid%(8G)))
        const int r = r0 + 4 * (tid % (8 * G));
        const int nv = R - r < 0 ? 0 : (R - r > 4 ? 4 : R - r);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = k0 + tid / (8 * G) + (32 / G) * i;
            gm_ld4<IS_B, BDT>(P, (long long)k * ld + r, vec, k < kend ? nv : 0, reg[i]);
        }
    }
    (void)K;
}

template <int G>
__device__ __forceinline__ void gm_store_tile(float* S, bool contig_is_k, int tid, const float (&reg)[G][4]) {
    constexpr int LD = 32 * G + 4;
    if (contig_is_k) {
        const int kq = 4 * (tid & 7);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int r = (tid >> 3) + 32 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(kq + q) * LD + r] = reg[i][q];
        }
    } else {
        const int rq = 4 * (tid % (8 * G));
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int k = tid / (8 * G) + (32 / G) * i;
            *(f32x4*)(S + k * LD + rq) = f32x4{reg[i][0], reg[i][1], reg[i][2], reg[i][3]};
        }
    }
}

// act: 0 none, 1 relu, 2 relu-backward mask (v if aux[row][col] > 0 else 0; aux has C's leading dimension),
//      3 "c I - P": out = (row == col ? beta : 0) - alpha * P   (beta is the diagonal constant, not an accumulate factor)
//      4 both: out = alpha * P and aux (written, same layout as C) = beta I - alpha * P
__device__ __forceinline__ float gm_act(float v, int act, const float* aux, long long idx) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return aux[idx] > 0.0f ? v : 0.0f;
    return v;
}

// epilogue / partial store shared by the fp32 and the split-f16 kernels
template <int WT>
__device__ __forceinline__ void gm_epilogue(const GemmArgs& g, const f32x16 (&acc)[WT][WT], int m0, int n0, int wm, int wn, int i31,
                                            int hi, int batch, int split) {
    constexpr int WS = 32 * WT;
    // ---- epilogue / partial store.  acc[a][b][r]: row = m0 + WS wm + 32a + mfma32_row(r,hi), col = n0 + WS wn + 32b + i31
    float* Cb = (g.splits > 1) ? g.ws + ((long long)batch * g.splits + split) * g.M * g.N : g.C + (long long)batch * g.sC;
    const int ldc = (g.splits > 1) ? g.N : g.ldc;
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b) {
            const int col = n0 + WS * wn + 32 * b + i31;
            if (col >= g.N) continue;
            const float bias = (g.splits > 1 || !g.bias) ? 0.0f : g.bias[col];
            // accumulate / mask operands of the whole 32 x 32 tile first (32 independent loads in flight): read one by one
            // between the stores they form a chain of 64 dependent round trips (the stores may alias them for the compiler)
            float oldv[16], auxv[16];
            const bool direct = g.splits <= 1 && g.act != 3 && g.act != 4;
            const bool need_old = direct && g.beta != 0.0f, need_aux = direct && g.act == 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + WS * wm + 32 * a + mfma32_row(r, hi);
                const long long idx = (long long)row * ldc + col;
                oldv[r] = (need_old && row < g.M) ? Cb[idx] : 0.0f;
                auxv[r] = (need_aux && row < g.M) ? g.aux[(long long)batch * g.sC + idx] : 1.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + WS * wm + 32 * a + mfma32_row(r, hi);
                if (row >= g.M) continue;
                float* dst = Cb + (long long)row * ldc + col;
                if (g.splits > 1) *dst = acc[a][b][r];
                else if (g.act == 3) *dst = (row == col ? g.beta : 0.0f) - g.alpha * acc[a][b][r];
                else if (g.act == 4) {   // two outputs: C = alpha P and aux = beta I - alpha P (aux is written, C's layout)
                    const float pv = g.alpha * acc[a][b][r];
                    *dst = pv;
                    const_cast<float*>(g.aux)[(long long)batch * g.sC + (long long)row * ldc + col] = (row == col ? g.beta : 0.0f) - pv;
                }
                else {
                    float v = g.alpha * acc[a][b][r] + bias;
                    if (g.beta != 0.0f) v += g.beta * oldv[r];
                    *dst = g.act == 1 ? fmaxf(v, 0.0f) : g.act == 2 ? (auxv[r] > 0.0f ? v : 0.0f) : v;
                }
            }
        }
}

// WT = MFMA tiles per wave and dimension: 2 -> 128 x 128 workgroup tile, 1 -> 64 x 64.
template <int BDT, int WT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    if (g.cond && __builtin_nontemporal_load(g.cond) == 0u) return;      // predicated launch (gemm_internal.h)
    constexpr int G = 2 * WT, BT = 64 * WT, LD = BT + 4, WS = 32 * WT;   // groups/thread, tile edge, LDS row, wave tile edge
    __shared__ __attribute__((aligned(16))) float As[GM_BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[GM_BK * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
    const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const float* A = g.A + (long long)batch * g.sA;
    const char* B = (const char*)g.B + (long long)batch * g.sB * (BDT == ACMIL_DTYPE_F32 ? 4 : 2);
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    // operand A is stored [M][K] (contiguous K) unless transA; operand B is stored [K][N] unless transB ([N][K])
    const bool a_ck = !g.transA, b_ck = (g.transB != 0);

    f32x16 acc[WT][WT];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // vector loads need 16-B (A, fp32 B) / 8-B (16-bit B) aligned groups: base and leading dimension multiples of 4 elements
    const bool va = ((g.lda & 3) == 0) && ((((size_t)A) & 15) == 0) && ((kbeg & 3) == 0);
    const bool vb = ((g.ldb & 3) == 0) && ((((size_t)B) & (BDT == ACMIL_DTYPE_F32 ? 15 : 7)) == 0) && ((kbeg & 3) == 0);
    float ra[G][4], rb[G][4];
    gm_load_tile<false, ACMIL_DTYPE_F32, G>(A, g.lda, a_ck, m0, kbeg, g.M, g.K, kend, tid, va, ra);
    gm_load_tile<true, BDT, G>(B, g.ldb, b_ck, n0, kbeg, g.N, g.K, kend, tid, vb, rb);
    for (int k0 = kbeg; k0 < kend; k0 += GM_BK) {
        __syncthreads();   // previous tile fully consumed
        gm_store_tile<G>(As, a_ck, tid, ra);
        gm_store_tile<G>(Bs, b_ck, tid, rb);
        __syncthreads();
        if (k0 + GM_BK < kend) {
            gm_load_tile<false, ACMIL_DTYPE_F32, G>(A, g.lda, a_ck, m0, k0 + GM_BK, g.M, g.K, kend, tid, va, ra);
            gm_load_tile<true, BDT, G>(B, g.ldb, b_ck, n0, k0 + GM_BK, g.N, g.K, kend, tid, vb, rb);
        }
        const float* ap = As + hi * LD + WS * wm + i31;
        const float* bp = Bs + hi * LD + WS * wn + i31;
#pragma unroll
        for (int kk = 0; kk < GM_BK / 2; ++kk) {
            float av[WT], bv[WT];
#pragma unroll
            for (int t = 0; t < WT; ++t) { av[t] = ap[2 * kk * LD + 32 * t]; bv[t] = bp[2 * kk * LD + 32 * t]; }
#pragma unroll
            for (int a = 0; a < WT; ++a)
#pragma unroll
                for (int b = 0; b < WT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    gm_epilogue<WT>(g, acc, m0, n0, wm, wn, i31, hi, batch, split);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-f16 variant ("f16x3"): every fp32 operand is split a = hi + lo in f16 while it is staged into LDS and the
// product is formed as hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- 3x the MFMA work on a
// pipe that is 16x faster than the fp32 one.  Relative error ~1e-6 (22 mantissa bits) for operands inside the f16 range
// (|v| < 65504; anything larger becomes inf and shows up loudly), values below 6e-8 flush to zero.  Used for the
// Linear-layer products (activations x weights) and the weight-gradient / input-gradient products of the backward.
// LDS: [plane hi|lo][row][32 k f16 + 8 pad] = 80-B rows, so the fragment ds_read_b128 of 32 consecutive rows is conflict-free.
#define GX_LDB 80
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// bf16 variant ("bf16x3"): same scheme with bf16 halves -- 16 mantissa bits (relative error ~1e-5) but the full fp32
// exponent range, so tiny operands (gradients of a softmax over 50 000 patches ~1e-7) need no scaling.  Used by the backward.
template <bool BF> struct GxT { typedef _Float16 T; typedef f16x8 V8; };
template <> struct GxT<true> { typedef __bf16 T; typedef bf16x8 V8; };
template <bool BF>
__device__ __forceinline__ f32x16 gx_mfma(typename GxT<BF>::V8 a, typename GxT<BF>::V8 b, f32x16 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool BF>
__device__ __forceinline__ void gx_split_pair(float x0, float x1, unsigned& h, unsigned& l) {
    if constexpr (BF) ga_split_pair_bf16(x0, x1, h, l); else ga_split_pair_f16(x0, x1, h, l);
}

template <bool BF>
__device__ __forceinline__ void gx_store_split(char* S, int tid, const float (&reg)[4][4]) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int kq = 4 * (tid & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 3) + 32 * i;
        unsigned h0, l0, h1, l1;
        gx_split_pair<BF>(reg[i][0], reg[i][1], h0, l0);
        gx_split_pair<BF>(reg[i][2], reg[i][3], h1, l1);
        *(u32x2*)(S + r * GX_LDB + kq * 2) = u32x2{h0, h1};
        *(u32x2*)(S + 128 * GX_LDB + r * GX_LDB + kq * 2) = u32x2{l0, l1};
    }
}

// Operands stored [K][rows] (rows contiguous: transposed A, plain B): thread -> row = tid % 128, 16 consecutive k starting
// at 16 * (tid / 128).  The 16 scalar loads are each coalesced across the lanes (consecutive rows); the thread then owns 16
// consecutive k of one row = 32 B of the hi plane and 32 B of the lo plane, written with 2 + 2 ds_write_b128.
template <bool IS_B, int BDT>
__device__ __forceinline__ void gx_load_rows(const void* P, int ld, int r0, int k0, int R, int kend, int tid, bool vec, float (&reg)[4][4]) {
    if constexpr (IS_B && BDT != ACMIL_DTYPE_F32) {
        // 16-bit operand: thread -> row PAIR 2 * (tid % 64), 8 consecutive k at 8 * (tid / 64); one 4-byte load per k
        // (256 B per wave-instruction) when the pair is aligned and interior.  reg index = 8 * (row in pair) + k.
        const int r = r0 + 2 * (tid & 63);
        const int kb = k0 + 8 * (tid >> 6);
        const bool pair = vec && (r + 1 < R);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kb + j;
            float v0 = 0.0f, v1 = 0.0f;
            if (k < kend) {
                const long long idx = (long long)k * ld + r;
                if (pair) {
                    const uint32_t w = *(const uint32_t*)((const uint16_t*)P + idx);
                    if constexpr (BDT == ACMIL_DTYPE_F16) {
                        v0 = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); v1 = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
                    } else { v0 = __builtin_bit_cast(float, w << 16); v1 = __builtin_bit_cast(float, w & 0xffff0000u); }
                } else {
                    if (r < R) v0 = gm_ldb<BDT>(P, idx);
                    if (r + 1 < R) v1 = gm_ldb<BDT>(P, idx + 1);
                }
            }
            reg[j >> 2][j & 3] = v0;
            reg[2 + (j >> 2)][j & 3] = v1;
        }
    } else {
        const int r = r0 + (tid & 127);
        const int kb = k0 + 16 * (tid >> 7);
        const bool rok = r < R;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = kb + j;
            float v = 0.0f;
            if (rok && k < kend) {
                const long long idx = (long long)k * ld + r;
                if constexpr (IS_B) v = gm_ldb<BDT>(P, idx); else v = ((const float*)P)[idx];
            }
            reg[j >> 2][j & 3] = v;
        }
    }
}

template <bool BF, bool PAIRS>
__device__ __forceinline__ void gx_store_split_rows(char* S, int tid, const float (&reg)[4][4]) {
    u32x4 h[2], l[2];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        unsigned hp, lp;
        gx_split_pair<BF>(reg[j >> 2][j & 3], reg[(j + 1) >> 2][(j + 1) & 3], hp, lp);
        h[j >> 3][(j & 7) >> 1] = hp;
        l[j >> 3][(j & 7) >> 1] = lp;
    }
    if constexpr (PAIRS) {       // registers 0-7: row r, k 0..7; 8-15: row r + 1 (see gx_load_rows, 16-bit operand)
        char* d = S + 2 * (tid & 63) * GX_LDB + 8 * (tid >> 6) * 2;
        *(u32x4*)(d) = h[0]; *(u32x4*)(d + GX_LDB) = h[1];
        *(u32x4*)(d + 128 * GX_LDB) = l[0]; *(u32x4*)(d + 128 * GX_LDB + GX_LDB) = l[1];
    } else {
        char* d = S + (tid & 127) * GX_LDB + 16 * (tid >> 7) * 2;
        *(u32x4*)(d) = h[0]; *(u32x4*)(d + 16) = h[1];
        *(u32x4*)(d + 128 * GX_LDB) = l[0]; *(u32x4*)(d + 128 * GX_LDB + 16) = l[1];
    }
}

template <int BDT, bool BF>
__global__ __launch_bounds__(256, 3) void gemm_f16x3_kernel(GemmArgs g) {
    typedef typename GxT<BF>::V8 V8;
    __shared__ __attribute__((aligned(16))) char As[2 * 128 * GX_LDB];
    __shared__ __attribute__((aligned(16))) char Bs[2 * 128 * GX_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const float* A = g.A + (long long)batch * g.sA;
    const char* B = (const char*)g.B + (long long)batch * g.sB * (BDT == ACMIL_DTYPE_F32 ? 4 : 2);
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const bool a_ck = !g.transA, b_ck = (g.transB != 0);   // operand stored with K contiguous?
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const bool va = ((g.lda & 3) == 0) && ((((size_t)A) & 15) == 0) && ((kbeg & 3) == 0);
    const bool vb = ((g.ldb & 3) == 0) && ((((size_t)B) & (BDT == ACMIL_DTYPE_F32 ? 15 : 7)) == 0) && ((kbeg & 3) == 0);
    const bool vb2 = ((g.ldb & 1) == 0) && ((((size_t)B) & 3) == 0) && ((n0 & 1) == 0);   // aligned row pairs of a 16-bit [K][N] operand
    float ra[4][4], rb[4][4];
    auto load_a = [&](int k0) {
        if (a_ck) gm_load_tile<false, ACMIL_DTYPE_F32, 4>(A, g.lda, true, m0, k0, g.M, g.K, kend, tid, va, ra);
        else gx_load_rows<false, ACMIL_DTYPE_F32>(A, g.lda, m0, k0, g.M, kend, tid, false, ra);
    };
    auto load_b = [&](int k0) {
        if (b_ck) gm_load_tile<true, BDT, 4>(B, g.ldb, true, n0, k0, g.N, g.K, kend, tid, vb, rb);
        else gx_load_rows<true, BDT>(B, g.ldb, n0, k0, g.N, kend, tid, vb2, rb);
    };
    load_a(kbeg); load_b(kbeg);
    const char* ap = As + (64 * wm + i31) * GX_LDB + hi * 16;
    const char* bp = Bs + (64 * wn + i31) * GX_LDB + hi * 16;
    for (int k0 = kbeg; k0 < kend; k0 += GM_BK) {
        __syncthreads();   // previous tile fully consumed
        if (a_ck) gx_store_split<BF>(As, tid, ra); else gx_store_split_rows<BF, false>(As, tid, ra);
        if (b_ck) gx_store_split<BF>(Bs, tid, rb); else gx_store_split_rows<BF, (BDT != ACMIL_DTYPE_F32)>(Bs, tid, rb);
        __syncthreads();
        if (k0 + GM_BK < kend) { load_a(k0 + GM_BK); load_b(k0 + GM_BK); }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            V8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *(const V8*)(ap + 32 * t * GX_LDB + kh * 32);
                al[t] = *(const V8*)(ap + 128 * GX_LDB + 32 * t * GX_LDB + kh * 32);
                bh[t] = *(const V8*)(bp + 32 * t * GX_LDB + kh * 32);
                bl[t] = *(const V8*)(bp + 128 * GX_LDB + 32 * t * GX_LDB + kh * 32);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = gx_mfma<BF>(ah[a], bh[b], acc[a][b]);
                    acc[a][b] = gx_mfma<BF>(al[a], bh[b], acc[a][b]);
                    acc[a][b] = gx_mfma<BF>(ah[a], bl[b], acc[a][b]);
                }
        }
    }
    gm_epilogue<2>(g, acc, m0, n0, wm, wn, i31, hi, batch, split);
}

__device__ __forceinline__ void gm_reduce_body(const GemmArgs& g, int nbatch, int block, int nblocks) {
    const long long per = (long long)g.M * g.N, total = per * nbatch;
    for (long long e = (long long)block * 256 + threadIdx.x; e < total; e += (long long)nblocks * 256) {
        const int b = e / per; const long long r = e % per;
        const float* src = g.ws + (long long)b * g.splits * per + r;
        float s = 0.0f;
        int sp = 0;
        for (; sp + 8 <= g.splits; sp += 8) {                                  // 8 loads in flight, summed in index order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(long long)(sp + j) * per];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; sp < g.splits; ++sp) s += src[(long long)sp * per];             // fixed order
        const int row = r / g.N, col = r % g.N;
        float v = g.alpha * s + (g.bias ? g.bias[col] : 0.0f);
        long long idx = (long long)row * g.ldc + col;
        float* dst = g.C + (long long)b * g.sC + idx;
        if (g.C2 && row >= g.split_row) { idx = (long long)(row - g.split_row) * g.ldc + col; dst = g.C2 + idx; }
        if (g.act == 3) { *dst = (row == col ? g.beta : 0.0f) - g.alpha * s; continue; }
        if (g.act == 4) { *dst = g.alpha * s; const_cast<float*>(g.aux)[(long long)b * g.sC + idx] = (row == col ? g.beta : 0.0f) - g.alpha * s; continue; }
        if (g.beta != 0.0f) v += g.beta * *dst;
        *dst = gm_act(v, g.act, g.aux + (long long)b * g.sC, idx);
    }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g, int nbatch) {
    if (g.cond && __builtin_nontemporal_load(g.cond) == 0u) return;
    gm_reduce_body(g, nbatch, blockIdx.x, gridDim.x);
}

// One launch that finishes up to two split-K products (their fixed-order reduces + epilogues) and one "sum of per-workgroup
// partial records" job (the gate pass of the GA backward): blocks [0, b1) reduce g1, [b1, b1 + b2) reduce g2, the rest
// take GM_EPW adjacent record elements each (one wave) -- lanes stride the records, shuffle tree (deterministic).
__global__ __launch_bounds__(256) void gemm_finish_kernel(GemmArgs g1, GemmArgs g2, RowSumJob job, int b1, int b2) {
    const int blk = blockIdx.x;
    if (blk < b1) { gm_reduce_body(g1, 1, blk, b1); return; }
    if (blk < b1 + b2) { gm_reduce_body(g2, 1, blk - b1, b2); return; }
    if (threadIdx.x >= 64) return;                      // one wave per block, neighbouring groups on one XCD (gemm_internal.h)
    const int e0 = gm_rec_group(blk - b1 - b2, job.len), lane = threadIdx.x & 63;
    if (e0 < 0) return;
    float sv[GM_EPW];
    gm_record_sum_n(job.part, job.records, job.stride, e0, job.len, lane, sv);
    float s = sv[0];
#pragma unroll
    for (int j = 1; j < GM_EPW; ++j) s = (lane == j) ? sv[j] : s;
    const int e = e0 + lane;
    if (lane >= GM_EPW || e >= job.len) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q < job.nseg && e >= job.off[q] && e < job.off[q] + job.cnt[q]) job.dst[q][e - job.off[q]] = s;
}

// 64 x 64 tiles when the 128 x 128 grid would leave most CUs idle and K is too short to split
static bool gm_small_tile(int M, int N, int K, int batch) {
    const long long tiles = (long long)((M + GM_BM - 1) / GM_BM) * ((N + GM_BN - 1) / GM_BN) * batch;
    return tiles < 128 && K < 4 * GM_BK * 8;
}

// choose a K split so that (tiles * splits) roughly fills the 256 CUs when the output is small and K is long
static int gm_pick_splits(int M, int N, int K, int batch) {
    const long long tiles = (long long)((M + GM_BM - 1) / GM_BM) * ((N + GM_BN - 1) / GM_BN) * batch;
    if (tiles >= 128 || K < 4 * GM_BK * 8) return 1;
    long long s = (512 + tiles - 1) / tiles;
    const long long maxs = K / (GM_BK * 4);
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    return s < 2 ? 1 : (int)s;
}

extern "C" size_t acmil_gemm_workspace_bytes(int M, int N, int K, int batch) {
    const int s = gm_pick_splits(M, N, K, batch);
    return s > 1 ? (((size_t)s * batch * M * N * sizeof(float) + 255) & ~(size_t)255) : 256;
}

static int gm_run(int x3, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                  long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta,
                  float* C, int ldc, long long strideC, const float* bias, int act, const float* aux,
                  int batch, void* workspace, void* stream, GemmArgs* defer = nullptr, const unsigned* cond = nullptr) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || lda <= 0 || ldb <= 0 || ldc < N) return ACMIL_ERR_SHAPE;
    if (!A || !B || !C) return ACMIL_ERR_NULL;
    if (act < 0 || act > 4) return ACMIL_ERR_UNSUPPORTED;
    if ((act == 2 || act == 4) && !aux) return ACMIL_ERR_NULL;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux; g.ws = (float*)workspace;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.transA = transA; g.transB = transB; g.b_dtype = b_dtype; g.act = act; g.alpha = alpha; g.beta = beta;
    g.C2 = nullptr; g.split_row = 0; g.cond = cond;
    if (cond && x3 != 0) return ACMIL_ERR_UNSUPPORTED;      // only the exact-fp32 kernels carry the predicate
    g.splits = gm_pick_splits(M, N, K, batch);
    if (g.splits > 1 && !workspace) return ACMIL_ERR_NULL;
    const int ktiles = (K + GM_BK - 1) / GM_BK;
    g.kchunk = g.splits > 1 ? ((ktiles + g.splits - 1) / g.splits) * GM_BK : K;
    if (g.splits > 1) g.splits = (K + g.kchunk - 1) / g.kchunk;
    if (g.splits < 1) g.splits = 1;
    hipStream_t st = (hipStream_t)stream;
    const bool use_x3 = x3 != 0;
    const bool small = !use_x3 && gm_small_tile(M, N, K, batch) && g.splits == 1 && b_dtype == ACMIL_DTYPE_F32;
    const int bt = small ? 64 : GM_BM;
    const dim3 grid((N + bt - 1) / bt, (M + bt - 1) / bt, g.splits * batch);
    if (x3 == 1) switch (b_dtype) {
        case ACMIL_DTYPE_F32: hipLaunchKernelGGL((gemm_f16x3_kernel<ACMIL_DTYPE_F32, false>), grid, dim3(256), 0, st, g); break;
        case ACMIL_DTYPE_F16: hipLaunchKernelGGL((gemm_f16x3_kernel<ACMIL_DTYPE_F16, false>), grid, dim3(256), 0, st, g); break;
        case ACMIL_DTYPE_BF16: hipLaunchKernelGGL((gemm_f16x3_kernel<ACMIL_DTYPE_BF16, false>), grid, dim3(256), 0, st, g); break;
        default: return ACMIL_ERR_UNSUPPORTED;
    }
    else if (x3 == 2) switch (b_dtype) {
        case ACMIL_DTYPE_F32: hipLaunchKernelGGL((gemm_f16x3_kernel<ACMIL_DTYPE_F32, true>), grid, dim3(256), 0, st, g); break;
        case ACMIL_DTYPE_F16: hipLaunchKernelGGL((gemm_f16x3_kernel<ACMIL_DTYPE_F16, true>), grid, dim3(256), 0, st, g); break;
        case ACMIL_DTYPE_BF16: hipLaunchKernelGGL((gemm_f16x3_kernel<ACMIL_DTYPE_BF16, true>), grid, dim3(256), 0, st, g); break;
        default: return ACMIL_ERR_UNSUPPORTED;
    }
    else if (small) hipLaunchKernelGGL((gemm_f32_kernel<ACMIL_DTYPE_F32, 1>), grid, dim3(256), 0, st, g);
    else switch (b_dtype) {
        case ACMIL_DTYPE_F32: hipLaunchKernelGGL((gemm_f32_kernel<ACMIL_DTYPE_F32, 2>), grid, dim3(256), 0, st, g); break;
        case ACMIL_DTYPE_F16: hipLaunchKernelGGL((gemm_f32_kernel<ACMIL_DTYPE_F16, 2>), grid, dim3(256), 0, st, g); break;
        case ACMIL_DTYPE_BF16: hipLaunchKernelGGL((gemm_f32_kernel<ACMIL_DTYPE_BF16, 2>), grid, dim3(256), 0, st, g); break;
        default: return ACMIL_ERR_UNSUPPORTED;
    }
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    if (defer) { *defer = g; return ACMIL_OK; }       // the caller finishes it with gemm_finish (splits may be 1: nothing left to do)
    if (g.splits > 1) {
        const long long total = (long long)M * N * batch;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, g, batch);
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    }
    return ACMIL_OK;
}

extern "C" int acmil_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                              long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta,
                              float* C, int ldc, long long strideC, const float* bias, int act, const float* aux,
                              int batch, void* workspace, void* stream) {
    return gm_run(0, transA, transB, M, N, K, alpha, A, lda, strideA, B, b_dtype, ldb, strideB, beta, C, ldc, strideC, bias, act,
                  aux, batch, workspace, stream);
}

extern "C" int acmil_gemm_f16x3(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                                long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta,
                                float* C, int ldc, long long strideC, const float* bias, int act, const float* aux,
                                int batch, void* workspace, void* stream) {
    return gm_run(1, transA, transB, M, N, K, alpha, A, lda, strideA, B, b_dtype, ldb, strideB, beta, C, ldc, strideC, bias, act,
                  aux, batch, workspace, stream);
}

extern "C" int acmil_gemm_bf16x3(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                                 long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta,
                                 float* C, int ldc, long long strideC, const float* bias, int act, const float* aux,
                                 int batch, void* workspace, void* stream) {
    return gm_run(2, transA, transB, M, N, K, alpha, A, lda, strideA, B, b_dtype, ldb, strideB, beta, C, ldc, strideC, bias, act,
                  aux, batch, workspace, stream);
}

// ---- internal (gemm_internal.h): a split-K product whose reduce is left to gemm_finish, so that several products and a
// record sum share ONE finishing launch (the GA training step).  batch = 1.
int gemm_run_deferred(int x3, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const void* B,
                      int b_dtype, int ldb, float beta, float* C, int ldc, const float* bias, int act, const float* aux,
                      void* workspace, hipStream_t stream, GemmArgs* out) {
    if (!out) return ACMIL_ERR_NULL;
    return gm_run(x3, transA, transB, M, N, K, alpha, A, lda, 0, B, b_dtype, ldb, 0, beta, C, ldc, 0, bias, act, aux, 1, workspace,
                  (void*)stream, out);
}

int gemm_f32_cond(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const void* B, int b_dtype, int ldb,
                  float* C, int ldc, const float* bias, int act, void* workspace, hipStream_t stream, const unsigned* cond) {
    return gm_run(0, transA, transB, M, N, K, alpha, A, lda, 0, B, b_dtype, ldb, 0, 0.0f, C, ldc, 0, bias, act, nullptr, 1, workspace,
                  (void*)stream, nullptr, cond);
}

int gemm_finish(const GemmArgs* g1, const GemmArgs* g2, const RowSumJob* job, hipStream_t st) {
    GemmArgs z; memset(&z, 0, sizeof(z)); z.splits = 1;
    RowSumJob zj; memset(&zj, 0, sizeof(zj));
    auto nblk = [](const GemmArgs* g) {
        if (!g || g->splits <= 1) return 0;
        const long long total = (long long)g->M * g->N;
        return (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    };
    const int b1 = nblk(g1), b2 = nblk(g2), b3 = (job && job->len > 0) ? gm_rec_blocks(job->len) : 0;
    if (b1 + b2 + b3 == 0) return ACMIL_OK;
    hipLaunchKernelGGL(gemm_finish_kernel, dim3(b1 + b2 + b3), dim3(256), 0, st, g1 ? *g1 : z, g2 ? *g2 : z, job ? *job : zj, b1, b2);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
