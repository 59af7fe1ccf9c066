// gemm_internal.h -- library-internal view of the GEMM launcher (gemm_f32.hip): products whose split-K reduce is deferred so
// that one finishing launch serves several of them (ga_backward.hip / ga_step.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

struct GemmArgs {
    const float* A; const void* B; float* C; const float* bias; const float* aux; float* ws;
    int M, N, K, lda, ldb, ldc;
    long long sA, sB, sC;   // batch strides (elements)
    int transA, transB, b_dtype, act, splits, kchunk;
    float alpha, beta;
    float* C2; int split_row;   // deferred split-K products only: rows >= split_row go to C2 (row - split_row); C2 null: one destination
    const unsigned* cond;       // or null: the launch does nothing unless *cond != 0 (device-side predicate: the exact-fp32 repeat of a
                                // split-f16 launch that flagged its bag, ga_forward.hip)
};

// sum `records` partial records of `len` floats (record r at part + r * stride) in a fixed order; element e goes to the
// segment q with off[q] <= e < off[q] + cnt[q] (elements outside every segment are dropped: padded branches)
struct RowSumJob {
    const float* part; int records, stride, len, nseg;
    int off[4], cnt[4];
    float* dst[4];
};

// One element of the record sum by one wave: lanes stride the records (lane l adds records l, l + 64, ... in that order), then a
// shuffle tree -- the summation order gemm_finish_kernel and ga_opt_step_kernel share (bit-identical results).  Eight loads are in
// flight per lane: a group step has thousands of records (one per backward tile of every bag), and one dependent round trip per
// record made the closing launch 61 us at 8 x 50 000 rows.
// The same sums for GM_EPW ADJACENT elements e0 .. e0 + GM_EPW - 1 by one wave (element e0 + j comes back in s[j] on every lane; per element
// the order above, so the results are the same bits): a lane then reads GM_EPW consecutive words of a record and the four waves of a
// block 64 consecutive bytes (one 16-byte load per lane and record) -- with one element per wave every 4-byte word cost a sector of its own on whichever XCD the block sat
// (the closing launch of an 8 x 50 000 step read 250 MB for 22.5 MB of records).  Elements >= len read element len - 1 (ignored by the caller).
#define GM_EPW 4
// Placement of the record sums: ONE wave per block (a wave's loads of a step go to 64 different lines: the loop lives off the memory
// parallelism of many CUs -- four such waves per CU took 52 - 56 us at 6 250 records where one per CU takes 36), and the blocks that read
// neighbouring 16-byte groups of the same lines sit on ONE XCD (block b runs on XCD b % 8: XCD x takes the groups [x per, (x + 1) per)),
// so that a line crosses the fabric once instead of once per group.  gm_rec_blocks(len) blocks; gm_rec_group(block) -> first element or -1.
__host__ __device__ static inline int gm_rec_blocks(int len) { return 8 * (((len + GM_EPW - 1) / GM_EPW + 7) / 8); }
__device__ __forceinline__ int gm_rec_group(int rb, int len) {
    const int ng = (len + GM_EPW - 1) / GM_EPW, per = (ng + 7) / 8;
    const int g = (rb & 7) * per + (rb >> 3);
    return ((rb >> 3) < per && g < ng) ? g * GM_EPW : -1;
}
typedef float gm_f4 __attribute__((ext_vector_type(4)));
typedef gm_f4 gm_f4u __attribute__((aligned(4)));      // a record's words are only 4-byte aligned (stride 901): hipcc still emits ONE global_load_dwordx4
__device__ __forceinline__ void gm_record_sum_n(const float* part, int records, int stride, int e0, int len, int lane, float (&s)[GM_EPW]) {
    static_assert(GM_EPW == 4, "one 16-byte load per record");
#pragma unroll
    for (int q = 0; q < GM_EPW; ++q) s[q] = 0.0f;
    const float* base = part + e0;
    int r = lane;
    if (e0 + GM_EPW <= len) {
        for (; r + 7 * 64 < records; r += 8 * 64) {
            gm_f4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const gm_f4u*)(base + (size_t)(r + 64 * j) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < GM_EPW; ++q) s[q] += v[j][q];
        }
        for (; r < records; r += 64) {
            const gm_f4 v = *(const gm_f4u*)(base + (size_t)r * stride);
#pragma unroll
            for (int q = 0; q < GM_EPW; ++q) s[q] += v[q];
        }
    } else {      // the last, partial group of a record: word by word, elements >= len read element len - 1 (ignored by the caller)
        for (; r < records; r += 64)
#pragma unroll
            for (int q = 0; q < GM_EPW; ++q) {
                const int e = e0 + q < len ? e0 + q : len - 1;
                s[q] += part[(size_t)r * stride + e];
            }
    }
#pragma unroll
    for (int q = 0; q < GM_EPW; ++q)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s[q] += __shfl_xor(s[q], o);
}

__device__ __forceinline__ float gm_record_sum(const float* part, int records, int stride, int e, int lane) {
    float s = 0.0f;
    int r = lane;
    for (; r + 7 * 64 < records; r += 8 * 64) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(r + 64 * j) * stride + e];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; r < records; r += 64) s += part[(size_t)r * stride + e];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// x3: 0 exact fp32 MFMA, 1 split-f16, 2 split-bf16 (as acmil_gemm_f32 / _f16x3 / _bf16x3).  Launches the product only; `out`
// receives what gemm_finish needs (out->splits == 1: the product already wrote C with its epilogue).
int gemm_run_deferred(int x3, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const void* B,
                      int b_dtype, int ldb, float beta, float* C, int ldc, const float* bias, int act, const float* aux,
                      void* workspace, hipStream_t stream, GemmArgs* out);
// (set out->C2 / out->split_row after the call to scatter the reduced rows over two tensors, e.g. [dWv; dWu])
int gemm_finish(const GemmArgs* g1, const GemmArgs* g2, const RowSumJob* job, hipStream_t stream);
// exact-fp32 MFMA product C = act(alpha op(A) op(B) + bias), batch 1, that runs only if *cond != 0 (cond null: always)
int gemm_f32_cond(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const void* B, int b_dtype, int ldb,
                  float* C, int ldc, const float* bias, int act, void* workspace, hipStream_t stream, const unsigned* cond);
