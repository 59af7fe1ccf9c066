// transmil_pinv.hip -- the Moore-Penrose iteration of the Nystrom attention as ONE launch: one workgroup per head.
//
// Reference: moore_penrose_iter_pinv, architecture/nystrom_attention.py:12-27:
//     z0 = x^T / (max_i sum_j |x_ij| * max_j sum_i |x_ij|)            (maxima over ALL heads: tm_pinv_maxsum_kernel, transmil.hip)
//     6 x:  xz = x z ;  z <- 1/4 z (13 I - xz (15 I - xz (7 I - xz)))
// The launch-per-product chain (24 batched [m x m x m] products + 2 per layer, acmil_gemm_f32) spent 12.6 us per product on
// launch + fill latency for 14 MFLOP per head: 0.33 ms per layer, 18 % of the TransMIL forward.  Here the heads are
// independent after the global maxima, so each head is ONE 8-wave workgroup that runs all 24 products back to back:
//   * operands stay in global memory (5 matrices of m x m fp32 per head: L2-resident) and are staged per 32-wide K chunk
//     into LDS as f16 hi / lo planes (double-buffered, one LDS barrier per chunk);
//   * every product is C = A B with B given TRANSPOSED, so both operand chunks are K-contiguous rows: the epilogue of the
//     producing product writes whichever layouts its consumers need (row-major for A operands, transposed -- one float4 per
//     lane -- for B operands), so no product ever transposes anything;
//   * arithmetic: split-f16, 3 MFMA products (v_mfma_f32_16x16x32_f16), fp32 accumulate, each operand matrix pre-scaled by a
//     power of two taken from its own running max |.| (tracked by the epilogues), so the halves sit in the f16 normal range
//     whatever the conditioning; the iteration is self-correcting, the fixtures bound the end result.
#include "ga_common.h"

#define TP_HEADS 8
typedef float tp_f32x4 __attribute__((ext_vector_type(4)));

enum { TP_EPI_XZ = 0, TP_EPI_C15 = 1, TP_EPI_C13 = 2, TP_EPI_Z = 3 };

__device__ __forceinline__ float tp_scale_of(unsigned amax_bits) {      // 2^(14 - floor(log2 amax)): scaled max in [2^14, 2^15)
    const int e = (int)((amax_bits >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;
    int se = 268 - e;
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}

template <int M>
struct TpGeom {
    static constexpr int NTHR = 512;            // 8 waves (2 per SIMD, 256 VGPRs each) as 2 x 4 output blocks
    static constexpr int WBR = M / 2, WBC = M / 4;      // rows / columns of a wave's output block
    static constexpr int NTR = WBR / 16, NTC = WBC / 16; // 16 x 16 tiles per block side
    static constexpr int PLANE = M * 64;        // one f16 plane of a chunk: M rows x 32 k
    static constexpr int BUF = 4 * PLANE;       // A hi, A lo, B hi, B lo
    static constexpr int LDS = 2 * BUF;
    static constexpr int F4 = 16 * M / NTHR;    // float4 loads per thread per chunk (A and B together)
    static_assert(M % 64 == 0 && M <= 192, "built for m = 64, 128, 192 (D_inner 128, 256, 384)");
};

// byte offset of (row, 16-byte K group kg) inside a plane; XOR swizzle: rows 4 apart share a bank group otherwise
__device__ __forceinline__ int tp_off(int row, int kg) { return row * 64 + ((kg ^ ((row >> 2) & 3)) << 4); }

// C = alpha * A * B with BT = B^T given; EPI selects the outputs (see the kernel).  amax slots: index of A's, B's and the outputs'.
template <int M, int EPI>
__device__ __forceinline__ void tp_gemm(const float* __restrict__ A, const float* __restrict__ BT, float* __restrict__ O0,
                                        float* __restrict__ O1, unsigned* amax, int ia, int ib, int io0, int io1, char* smem) {
    using G = TpGeom<M>;
    // an OPAQUE copy of the thread id per product: without it LLVM hoists every address of all four products of an iteration
    // (staging pointers, LDS offsets, 18 tiles x 5 epilogue addresses each) out of the iteration loop and spills hundreds of registers
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int r0 = wr * G::WBR, c0 = wc * G::WBC;
    const float sa = tp_scale_of(amax[ia]), sb = tp_scale_of(amax[ib]);
    __syncthreads();                                    // everybody has read the operand maxima
    if (tid == 0) { amax[io0] = 0u; if (io1 >= 0) amax[io1] = 0u; }
    tp_f32x4 acc[G::NTR][G::NTC];
#pragma unroll
    for (int i = 0; i < G::NTR; ++i)
#pragma unroll
        for (int j = 0; j < G::NTC; ++j) acc[i][j] = tp_f32x4{0.f, 0.f, 0.f, 0.f};
    // staging map: float4 q of this thread: e = q * 1024 + tid -> matrix (e >= 8M: B), row (e % 8M) / 8, float4 k4 = e % 8
    tp_f32x4 st[G::F4];
    auto gload = [&](int c) {
#pragma unroll
        for (int q = 0; q < G::F4; ++q) {
            const int e = q * G::NTHR + tid;
            const bool isb = e >= 8 * M;
            const int r = (isb ? e - 8 * M : e) >> 3, k4 = e & 7;
            st[q] = *(const tp_f32x4*)((isb ? BT : A) + (size_t)r * M + 32 * c + 4 * k4);
        }
    };
    auto lstore = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < G::F4; ++q) {
            const int e = q * G::NTHR + tid;
            const bool isb = e >= 8 * M;
            const int r = (isb ? e - 8 * M : e) >> 3, k4 = e & 7;
            const float s = isb ? sb : sa;
            unsigned h0, l0, h1, l1;
            ga_split_pair_f16(st[q][0] * s, st[q][1] * s, h0, l0);
            ga_split_pair_f16(st[q][2] * s, st[q][3] * s, h1, l1);
            char* p = buf + (isb ? 2 * G::PLANE : 0) + tp_off(r, k4 >> 1) + (k4 & 1) * 8;
            *(uint2*)p = make_uint2(h0, h1);
            *(uint2*)(p + G::PLANE) = make_uint2(l0, l1);
        }
    };
    constexpr int NC = M / 32;
    gload(0);
    lstore(smem);
    ga_lds_barrier();
    const int fr = lane & 15, kg = lane >> 4;
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) gload(c + 1);
        const char* buf = smem + (c & 1) * G::BUF;
        f16x8 bh[G::NTC], bl[G::NTC];
#pragma unroll
        for (int t = 0; t < G::NTC; ++t) {
            const int ob = tp_off(c0 + 16 * t + fr, kg);
            bh[t] = *(const f16x8*)(buf + 2 * G::PLANE + ob);
            bl[t] = *(const f16x8*)(buf + 3 * G::PLANE + ob);
        }
#pragma unroll
        for (int i = 0; i < G::NTR; ++i) {
            const int oa = tp_off(r0 + 16 * i + fr, kg);
            const f16x8 ah = *(const f16x8*)(buf + oa);
            const f16x8 al = *(const f16x8*)(buf + G::PLANE + oa);
#pragma unroll
            for (int j = 0; j < G::NTC; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], acc[i][j], 0, 0, 0);
            }
        }
        if (c + 1 < NC) lstore(smem + ((c + 1) & 1) * G::BUF);
        ga_lds_barrier();
    }
    // ---- epilogue.  Lane holds C[row0 + i][col], i < 4, row0 = r0 + 16 ti + 4 (lane >> 4), col = c0 + 16 tj + (lane & 15).
    // (all global reads of A / BT are complete: the last chunk was staged before the last barrier -- in-place outputs are safe)
    const float inv = 1.0f / (sa * sb);
    float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
    for (int ti = 0; ti < G::NTR; ++ti)
#pragma unroll
        for (int tj = 0; tj < G::NTC; ++tj) {
            __builtin_amdgcn_sched_barrier(0);      // one tile at a time: otherwise every tile's addresses and temporaries are live together
            const int row0 = r0 + 16 * ti + 4 * kg, col = c0 + 16 * tj + fr;
            tp_f32x4 v = acc[ti][tj] * inv;
            if constexpr (EPI == TP_EPI_XZ) {            // O0 = xz (row-major: the A operand of the next two products), O1 = (7 I - xz)^T
                tp_f32x4 t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    O0[(size_t)(row0 + i) * M + col] = v[i];
                    t[i] = ((row0 + i == col) ? 7.0f : 0.0f) - v[i];
                    m0 = fmaxf(m0, fabsf(v[i])); m1 = fmaxf(m1, fabsf(t[i]));
                }
                *(tp_f32x4*)(O1 + (size_t)col * M + row0) = t;
            } else if constexpr (EPI == TP_EPI_C15 || EPI == TP_EPI_C13) {      // O0 = (c I - P)^T only (consumed as a B operand)
                constexpr float cI = (EPI == TP_EPI_C15) ? 15.0f : 13.0f;
                tp_f32x4 t;
#pragma unroll
                for (int i = 0; i < 4; ++i) { t[i] = ((row0 + i == col) ? cI : 0.0f) - v[i]; m0 = fmaxf(m0, fabsf(t[i])); }
                *(tp_f32x4*)(O0 + (size_t)col * M + row0) = t;
            } else {                                      // z' = 1/4 z t: O0 = z' (row-major, in place over A), O1 = z'^T
                v = v * 0.25f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { O0[(size_t)(row0 + i) * M + col] = v[i]; m0 = fmaxf(m0, fabsf(v[i])); }
                *(tp_f32x4*)(O1 + (size_t)col * M + row0) = v;
            }
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
    if (lane == 0) {
        atomicMax(&amax[io0], __float_as_uint(m0));
        if (EPI == TP_EPI_XZ) atomicMax(&amax[io1], __float_as_uint(m1));
        if (EPI == TP_EPI_Z) atomicMax(&amax[io1], __float_as_uint(m0));
    }
    __syncthreads();      // outputs (global) and maxima (LDS) visible to the whole workgroup
}

// grid = heads.  X = softmax(sim2) [H][m][m]; Z (row-major) receives the pseudo-inverse; ZT, XZ, T1T, ST: scratch of the same size.
template <int M>
__global__ __launch_bounds__(512) void tm_pinv_fused_kernel(const float* __restrict__ X_all, float* __restrict__ Z_all,
                                                             float* __restrict__ ZT_all, float* __restrict__ XZ_all,
                                                             float* __restrict__ T1T_all, float* __restrict__ ST_all,
                                                             const unsigned* __restrict__ scal, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned amax[8];        // 0 x, 1 z (= z^T), 2 xz, 3 t1, 4 s
    const size_t hoff = (size_t)blockIdx.x * M * M;
    const float* X = X_all + hoff;
    float *Z = Z_all + hoff, *ZT = ZT_all + hoff, *XZ = XZ_all + hoff, *T1T = T1T_all + hoff, *ST = ST_all + hoff;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 8) amax[tid] = 0u;
    __syncthreads();
    // z0 = x^T / (max row sum * max column sum), both layouts; max |x| on the way
    const float inv = 1.0f / (__uint_as_float(scal[0]) * __uint_as_float(scal[1]));
    float mx = 0.0f;
    for (int e = tid; e < M * M; e += 512) {
        const int i = e / M, j = e - i * M;
        const float x = X[e];
        mx = fmaxf(mx, fabsf(x));
        ZT[e] = x * inv;
        Z[(size_t)j * M + i] = x * inv;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) { atomicMax(&amax[0], __float_as_uint(mx)); atomicMax(&amax[1], __float_as_uint(mx * inv)); }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        tp_gemm<M, TP_EPI_XZ>(X, ZT, XZ, T1T, amax, 0, 1, 2, 3, smem);          // xz = x z ; t1 = 7 I - xz
        tp_gemm<M, TP_EPI_C15>(XZ, T1T, ST, nullptr, amax, 2, 3, 4, -1, smem);  // s = 15 I - xz t1
        tp_gemm<M, TP_EPI_C13>(XZ, ST, T1T, nullptr, amax, 2, 4, 3, -1, smem);  // t1 = 13 I - xz s
        tp_gemm<M, TP_EPI_Z>(Z, T1T, Z, ZT, amax, 1, 3, 1, 1, smem);            // z = 1/4 z t1
    }
}

// launcher (transmil.hip): false = this m has no fused instance (the caller keeps its product chain)
bool tm_pinv_fused_supported(int m) { return m == 64 || m == 128 || m == 192; }

template <int M>
static int tm_pinv_launch(const float* X, float* Z, float* ZT, float* XZ, float* T1T, float* ST, const unsigned* scal, int iters, hipStream_t st) {
    static bool attr_done[16] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ACMIL_ERR_LAUNCH;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)tm_pinv_fused_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, TpGeom<M>::LDS) != hipSuccess)
            return ACMIL_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(tm_pinv_fused_kernel<M>, dim3(TP_HEADS), dim3(512), TpGeom<M>::LDS, st, X, Z, ZT, XZ, T1T, ST, scal, iters);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

int tm_pinv_fused(const float* X, float* Z, float* ZT, float* XZ, float* T1T, float* ST, const unsigned* scal, int m, int iters, hipStream_t st) {
    switch (m) {
        case 64: return tm_pinv_launch<64>(X, Z, ZT, XZ, T1T, ST, scal, iters, st);
        case 128: return tm_pinv_launch<128>(X, Z, ZT, XZ, T1T, ST, scal, iters, st);
        case 192: return tm_pinv_launch<192>(X, Z, ZT, XZ, T1T, ST, scal, iters, st);
    }
    return ACMIL_ERR_UNSUPPORTED;
}


// =====================================================================================================================
// The default form of the chain: ONE WAVE PER 16 x 16 OUTPUT TILE, the whole K extent in registers.
// A product of the chain is 14 MFLOP per head; what it costs is latency.  The generic GEMM (gemm_f32.hip: 64 x 64 tiles, K loop of
// 32-wide steps, two __syncthreads per step each waiting for the prefetched loads) took 12.4 us per product.  Here a wave loads
// its 16 rows of A and its 16 rows of B^T -- both K-contiguous, 2 x M/16 float4 per lane, ALL in flight at once -- and runs
// M/4 exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32: lane (r = lane & 15, kq = lane >> 4) supplies A[r][k], so with float4 loads at
// k = 16 j + 4 kq the four elements feed four consecutive MFMAs -- a permutation of the K order that A and B share).  No LDS,
// no barrier: one L2 round trip + 0.7 us of MFMA per product, 1152 waves at m = 192.  The epilogues write the layouts the next
// products read (row-major for A operands, transposed -- one float4 per lane -- for B operands), as in the one-launch kernel above.
// Arithmetic: exact fp32 products, fp32 accumulate (the class of the chain it replaces; reference nystrom_attention.py:12-27).
// (Measured floor: a dispatch of ANY kernel reports >= 4.6 us here; a product launch takes 8.2 us, the generic GEMM took 12.4.)
// The iteration in THREE dependent launches instead of four.  With y = x z:
//     z' = 1/4 z (13 I - y (15 I - y (7 I - y))) = 1/4 (13 z - 15 a + (7 z - a) b),   a = z y,  b = y y
// (the same polynomial, re-associated): a and b depend on y only, so they share ONE launch (twice the blocks), and with
// l = 7 z - a the last product is z' = w + 1/4 l b, w = 1/4 (15 l - 92 z) formed by the second launch's epilogue:
//   P1  y = x z                      -> y (row-major), y^T
//   P2  a = z y  -> l, w (row-major)        |  b = y y -> b^T          (one launch: first / second half of the grid)
//   P3  z' = w + 1/4 l b             -> z, z^T (in place: P3 reads neither)
// Round 4: TWO dependent launches per iteration.  x a = x z y = y y = b, so multiplying the update by x from the left gives the next y
// from y and b alone:  y' = x z' = 1/4 (13 y - 15 b + (7 y - b) b)  -- the same form as z' with (y, b) in place of (z, a).  P2's second
// half therefore also leaves ly = 7 y - b and wy = 1/4 (15 ly - 92 y), P3 becomes a dual launch (z' = w + 1/4 l b | y' = wy + 1/4 ly b,
// both with b^T as the shared B operand) and P1 runs once, before the first iteration: 14 launches per layer instead of 19.
// (y' is the exact image of the iteration's own map t -> t (13 - 15 t + 7 t^2 - t^3) / 4 on y, whose fixed point 1 is
//  super-attracting; rounding differs from recomputing x z' by ~1e-7 relative per iteration.  ACMIL_TM_PINV_Y1=1 keeps P1 per iteration.)
#ifndef TP_HEAD_XCD
#define TP_HEAD_XCD 1
#endif
enum { TQ_Y = 0, TQ_DUAL = 1, TQ_ZF = 2, TQ_DUAL2 = 3, TQ_ZF2 = 4 };

// TQ_DUAL2 / TQ_ZF2 (second half of the grid = the y side): aux1 = its elementwise operand, O3 / O4 = its outputs
// Workgroups of TWO waves (round 4): the chain runs on a side stream beside the attn3 leg, whose two 6-wave workgroups of 128 VGPRs
// leave room on two of a CU's four SIMDs only -- a four-wave workgroup waited for that launch to retire (tools/coresident_probe.hip).
// A 32 x 32 block is two workgroups (tile rows).  The launch bound stays 256 (launched with 128 threads): under it hipcc keeps the
// m = 192 instances at 128 registers per wave -- two of them fit a half-free SIMD -- and the wider ones free of scratch.
template <int M, int EPI>
__global__ __launch_bounds__(256) void tm_pinv_prod_kernel(const float* __restrict__ A0_all, const float* __restrict__ BT0_all,
                                                            const float* __restrict__ A1_all, const float* __restrict__ aux_all,
                                                            float* __restrict__ O0_all, float* __restrict__ O1_all, float* __restrict__ O2_all,
                                                            const float* __restrict__ aux1_all = nullptr, float* __restrict__ O3_all = nullptr,
                                                            float* __restrict__ O4_all = nullptr) {
    constexpr bool DUAL = (EPI == TQ_DUAL || EPI == TQ_DUAL2 || EPI == TQ_ZF2);
    constexpr int NJ = M / 16, TB = M / 32, NBLK = TB * TB * TP_HEADS;
    // Block -> (head, 32-row band by, 32-column band bx) such that the TB blocks of one (head, by) -- which read the SAME rows of A --
    // sit on one XCD (block b runs on XCD b % 8: used for speed only)
    int L = blockIdx.x;
    bool second = false;                                    // TQ_DUAL: blocks 2 NBLK.. compute b = y y
    if (DUAL && L >= 2 * NBLK) { L -= 2 * NBLK; second = true; }
    const int xcd = L & 7, half = (L >> 3) & 1, q = L >> 4;
#if TP_HEAD_XCD
    // ALL blocks of a head on one XCD (8 heads, 8 XCDs): its A, B^T and elementwise operands cross the fabric once per launch instead of
    // once per XCD that hosts one of its (head, by) groups
    static_assert(TP_HEADS == 8, "head -> XCD");
    const int head = xcd, by = q / TB, bx = q - by * TB;
#else
    const int grp = (q / TB) * 8 + xcd;                     // (head, by) pair, TP_HEADS * TB of them
    const int head = grp / TB, by = grp - head * TB, bx = q - (q / TB) * TB;
#endif
    const size_t hoff = (size_t)head * M * M;
    const float* A = (second ? A1_all : A0_all) + hoff;
    const float* BT = BT0_all + hoff;
    const int lane = threadIdx.x & 63, wave = 2 * half + (threadIdx.x >> 6);
    const int ti = 2 * by + (wave >> 1), tj = 2 * bx + (wave & 1);
    const int r = lane & 15, kq = lane >> 4;
    const tp_f32x4* ap = (const tp_f32x4*)(A + (size_t)(16 * ti + r) * M + 4 * kq);
    const tp_f32x4* bp = (const tp_f32x4*)(BT + (size_t)(16 * tj + r) * M + 4 * kq);
    tp_f32x4 a[NJ], b[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { a[j] = ap[4 * j]; b[j] = bp[4 * j]; }
    // lane holds C[row0 + i][col], i < 4; the elementwise operands of the epilogue are requested with the panels
    const int row0 = 16 * ti + 4 * kq, col = 16 * tj + r;
    tp_f32x4 ax = {0.f, 0.f, 0.f, 0.f};
    if ((EPI == TQ_DUAL && !second) || EPI == TQ_ZF || EPI == TQ_DUAL2 || EPI == TQ_ZF2) {
        const float* X = ((second && (EPI == TQ_DUAL2 || EPI == TQ_ZF2)) ? aux1_all : aux_all) + hoff;
#pragma unroll
        for (int i = 0; i < 4; ++i) ax[i] = X[(size_t)(row0 + i) * M + col];
    }
    __builtin_amdgcn_sched_barrier(0);      // every load is issued before the first MFMA (left alone, hipcc sinks each load to its use: 30 VGPRs, serial latencies)
    tp_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two chains: the dependent-issue latency is 40 cycles
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], b[j][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][1], b[j][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][2], b[j][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][3], b[j][3], acc1, 0, 0, 0);
    }
    const tp_f32x4 v = acc0 + acc1;
    if constexpr (EPI == TQ_Y) {                             // y row-major + y^T
        float* O0 = O0_all + hoff;
#pragma unroll
        for (int i = 0; i < 4; ++i) O0[(size_t)(row0 + i) * M + col] = v[i];
        *(tp_f32x4*)(O1_all + hoff + (size_t)col * M + row0) = v;
    } else if constexpr (EPI == TQ_DUAL || EPI == TQ_DUAL2) {
        if (second) {                                        // b^T (consumed as a B operand)
            *(tp_f32x4*)(O2_all + hoff + (size_t)col * M + row0) = v;
            if constexpr (EPI == TQ_DUAL2) {                 // + ly = 7 y - b ; wy = 1/4 (15 ly - 92 y)      (ax = y)
                float* O3 = O3_all + hoff; float* O4 = O4_all + hoff;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float l = 7.0f * ax[i] - v[i];
                    O3[(size_t)(row0 + i) * M + col] = l;
                    O4[(size_t)(row0 + i) * M + col] = 0.25f * (15.0f * l - 92.0f * ax[i]);
                }
            }
        } else {                                             // l = 7 z - a ; w = 1/4 (15 l - 92 z)      (ax = z)
            float* O0 = O0_all + hoff; float* O1 = O1_all + hoff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float l = 7.0f * ax[i] - v[i];
                O0[(size_t)(row0 + i) * M + col] = l;
                O1[(size_t)(row0 + i) * M + col] = 0.25f * (15.0f * l - 92.0f * ax[i]);
            }
        }
    } else {                                                 // z' = w + 1/4 l b      (ax = w): row-major + transposed; second half: y'
        float* O0 = (second ? O3_all : O0_all) + hoff;
        float* const OT = (second ? O4_all : O1_all);
        tp_f32x4 z;
#pragma unroll
        for (int i = 0; i < 4; ++i) { z[i] = fmaf(0.25f, v[i], ax[i]); O0[(size_t)(row0 + i) * M + col] = z[i]; }
        *(tp_f32x4*)(OT + hoff + (size_t)col * M + row0) = z;
    }
}

// z0 = x^T / (scal0 * scal1) in both layouts
__global__ __launch_bounds__(128) void tm_pinv_init2_kernel(const float* __restrict__ x, int m, const unsigned* __restrict__ scal,
                                                             float* __restrict__ z, float* __restrict__ zt) {
    const float inv = 1.0f / (__uint_as_float(scal[0]) * __uint_as_float(scal[1]));
    const size_t per = (size_t)m * m, total = per * TP_HEADS;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t h = e / per, rr = e % per; const int i = (int)(rr / m), j = (int)(rr % m);
        const float v = x[e] * inv;
        zt[e] = v;
        z[h * per + (size_t)j * m + i] = v;
    }
}

bool tm_pinv_tiles_supported(int m) { return m == 64 || m == 128 || m == 192 || m == 256 || m == 384; }

template <int M>
static int tm_pinv_tiles_run(const float* X, float* Z, float* ZT, float* Wp, float* BT, float* Y, float* YT, float* Lb,
                             const unsigned* scal, int iters, float** z_final, hipStream_t st, float* LY, float* WY) {
    hipLaunchKernelGGL(tm_pinv_init2_kernel, dim3(512), dim3(128), 0, st, X, M, scal, Z, ZT);
    const unsigned nblk = 2 * (M / 32) * (M / 32) * TP_HEADS;      // two-wave workgroups: two per 32 x 32 block
    const dim3 block(128);
    static const bool y_each = ACMIL_AB_ENV("ACMIL_TM_PINV_Y1") != nullptr;      // A/B knob: recompute y = x z in every iteration (round 3)
    if (LY && WY && !y_each) {
        hipLaunchKernelGGL((tm_pinv_prod_kernel<M, TQ_Y>), dim3(nblk), block, 0, st, X, ZT, (const float*)nullptr, (const float*)nullptr, Y, YT, (float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, (float*)nullptr);
        for (int it = 0; it < iters; ++it) {
            const bool last = it + 1 == iters;
            // a = z y -> l, w | b = y y -> b^T, ly, wy
            hipLaunchKernelGGL((tm_pinv_prod_kernel<M, TQ_DUAL2>), dim3(2 * nblk), block, 0, st, Z, YT, Y, Z, Lb, Wp, BT, (const float*)Y, LY, WY);
            // z' = w + 1/4 l b -> z, z^T | y' = wy + 1/4 ly b -> y, y^T   (the last iteration needs no y')
            if (last) hipLaunchKernelGGL((tm_pinv_prod_kernel<M, TQ_ZF>), dim3(nblk), block, 0, st, Lb, BT, (const float*)nullptr, Wp, Z, ZT, (float*)nullptr,
                                         (const float*)nullptr, (float*)nullptr, (float*)nullptr);
            else hipLaunchKernelGGL((tm_pinv_prod_kernel<M, TQ_ZF2>), dim3(2 * nblk), block, 0, st, Lb, BT, LY, Wp, Z, ZT, (float*)nullptr, (const float*)WY, Y, YT);
        }
        *z_final = Z;
        return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
    }
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL((tm_pinv_prod_kernel<M, TQ_Y>), dim3(nblk), block, 0, st, X, ZT, (const float*)nullptr, (const float*)nullptr, Y, YT, (float*)nullptr);
        hipLaunchKernelGGL((tm_pinv_prod_kernel<M, TQ_DUAL>), dim3(2 * nblk), block, 0, st, Z, YT, Y, Z, Lb, Wp, BT);
        hipLaunchKernelGGL((tm_pinv_prod_kernel<M, TQ_ZF>), dim3(nblk), block, 0, st, Lb, BT, (const float*)nullptr, Wp, Z, ZT, (float*)nullptr);
    }
    *z_final = Z;
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// X [H][m][m] -> *z_final = the buffer (Za or Zb) that holds the pseudo-inverse, row-major
int tm_pinv_tiles(const float* X, float* Za, float* ZTa, float* Zb, float* ZTb, float* XZ, float* T1T, float* ST, const unsigned* scal,
                  int m, int iters, float** z_final, hipStream_t st, float* LY, float* WY) {
    switch (m) {
        case 64: return tm_pinv_tiles_run<64>(X, Za, ZTa, Zb, ZTb, XZ, T1T, ST, scal, iters, z_final, st, LY, WY);
        case 128: return tm_pinv_tiles_run<128>(X, Za, ZTa, Zb, ZTb, XZ, T1T, ST, scal, iters, z_final, st, LY, WY);
        case 192: return tm_pinv_tiles_run<192>(X, Za, ZTa, Zb, ZTb, XZ, T1T, ST, scal, iters, z_final, st, LY, WY);
        case 256: return tm_pinv_tiles_run<256>(X, Za, ZTa, Zb, ZTb, XZ, T1T, ST, scal, iters, z_final, st, LY, WY);
        case 384: return tm_pinv_tiles_run<384>(X, Za, ZTa, Zb, ZTb, XZ, T1T, ST, scal, iters, z_final, st, LY, WY);
    }
    return ACMIL_ERR_UNSUPPORTED;
}
