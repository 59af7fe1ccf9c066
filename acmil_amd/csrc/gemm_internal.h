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
