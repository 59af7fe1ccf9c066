// transmil.hip -- TransMIL / Nystrom-attention forward (eval) of one bag on MI355X (gfx950).
//
// Replaces (reference file:line, /root/reference):
//   TransMIL.forward           architecture/transMIL.py:60-91
//   TransLayer.forward         architecture/transMIL.py:25-28
//   PPEG.forward               architecture/transMIL.py:38-45
//   NystromAttention.forward   architecture/nystrom_attention.py:67-149 (mask=None; stand-in for pip nystrom_attention 0.0.12)
//   moore_penrose_iter_pinv    architecture/nystrom_attention.py:12-27
//
// All matrix products run on the exact-fp32 MFMA GEMM (gemm_f32.hip) with strided / batched views (no head
// split or merge copies: q, k, v are column slices of the [n', 3Di] projection, the attention output is written
// straight into the merged-head layout).  The rest are small HBM-bound kernels below: LayerNorm, landmark means,
// row softmax (short rows: one wave per row; long rows over n': one workgroup per row), pinv initialisation with
// the reference's GLOBAL max over heads, the 33-tap depth-wise residual conv along the sequence, and PPEG as ONE
// depth-wise 7x7 conv whose kernel is w7 + pad(w5) + pad(w3) + identity (same zero padding, same centre).
// The product is re-associated as attn1 (attn2^+ (attn3 v)) (saves 59 GFLOP per layer at N = 1e5; differs from
// the reference's association only by fp32 round-off).  Reference quirks kept: FRONT zero padding to a multiple
// of m landmarks (padded rows take part in all three softmaxes), repeat-padding of the token grid to a square.
#include <math.h>
#include <stdlib.h>

#include <mutex>

#include "ga_common.h"

extern "C" int acmil_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                              long long strideA, const void* B, int b_dtype, int ldb, long long strideB, float beta,
                              float* C, int ldc, long long strideC, const float* bias, int act, const float* aux,
                              int batch, void* workspace, void* stream);
extern "C" size_t acmil_gemm_workspace_bytes(int M, int N, int K, int batch);

#define TM_HEADS 8
#define TM_RES 33
#define TM_QKV_GUARD 16      // zero rows kept before and after QKV [npad, 3 Di]: the convolution folded into the attn1 leg reads +-16 rows

// ------------------------------------------------------------------------------------------------ small kernels
// rows [0, zero_rows) of out are zeroed; rows [zero_rows, zero_rows + rows) = LayerNorm(in row) (eps 1e-5).  One wave per row.
__global__ __launch_bounds__(256) void tm_layernorm_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                          int dim, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int zero_rows) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long long)rows + zero_rows) return;
    float* o = out + r * dim;
    if (r < zero_rows) { for (int c = lane; c < dim; c += 64) o[c] = 0.0f; return; }
    const float* x = in + r * dim;
    // the row is read ONCE into registers (dim <= 1024: 16 values per lane), statistics and output come from there
    float v[16];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; v[i] = c < dim ? x[c] : 0.0f; s += v[i]; }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) s += __shfl_xor(s, o2);
    const float mean = s / dim;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = (lane + 64 * i < dim) ? v[i] - mean : 0.0f; q = fmaf(d, d, q); }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) q += __shfl_xor(q, o2);
    const float rstd = 1.0f / sqrtf(q / dim + 1e-5f);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < dim) o[c] = (v[i] - mean) * rstd * gamma[c] + beta[c]; }
}

// LayerNorm folded into the next product (lin_qkv_norm_run, linear.hip): only the row statistics are formed here, exactly as
// tm_layernorm_kernel forms them (row read once into registers, mean, then the centred sum of squares): ab[r] = (rstd, -mean * rstd),
// (0, 0) for the zero padding rows in front -- the normalised activations [npad, Di] (154 MB at cfg4, written and read once per
// layer) never exist.  Also zeroes the two atomic-max words of this layer's Moore-Penrose scaling.  One wave per row.
__global__ __launch_bounds__(256) void tm_rowstats_kernel(const float* __restrict__ in, float* __restrict__ ab, int rows, int dim, int zero_rows,
                                                         unsigned* __restrict__ scal_zero) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (scal_zero && blockIdx.x == 0 && threadIdx.x < 2) scal_zero[threadIdx.x] = 0u;
    if (r >= (long long)rows + zero_rows) return;
    if (r < zero_rows) { if (lane == 0) *(f32x2*)(ab + 2 * r) = f32x2{0.0f, 0.0f}; return; }
    const float* x = in + r * dim;
    float v[16];
    float s = 0.0f;
    if ((dim & 3) == 0 && dim <= 1024) {      // one float4 per lane and 256 columns
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * lane + 256 * i;
            const f32x4 t = c < dim ? *(const f32x4*)(x + c) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
            s += (t[0] + t[1]) + (t[2] + t[3]);
        }
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1) s += __shfl_xor(s, o2);
        const float mean = s / dim;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = (4 * lane + 256 * i + j < dim) ? v[4 * i + j] - mean : 0.0f; q = fmaf(d, d, q); }
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1) q += __shfl_xor(q, o2);
        const float rstd = 1.0f / sqrtf(q / dim + 1e-5f);
        if (lane == 0) *(f32x2*)(ab + 2 * r) = f32x2{rstd, -mean * rstd};
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; v[i] = c < dim ? x[c] : 0.0f; s += v[i]; }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) s += __shfl_xor(s, o2);
    const float mean = s / dim;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = (lane + 64 * i < dim) ? v[i] - mean : 0.0f; q = fmaf(d, d, q); }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) q += __shfl_xor(q, o2);
    const float rstd = 1.0f / sqrtf(q / dim + 1e-5f);
    if (lane == 0) *(f32x2*)(ab + 2 * r) = f32x2{rstd, -mean * rstd};
}

// bias of the LayerNorm-folded to_qkv: wb[job][c] = sum_k W[c][k] beta[k], and (round 6) the row sums of the folded weights
// wsum[job][c] = sum_k W[c][k] gamma[k] -- what the in-kernel row statistics multiply the mean with (LinArgs::wsum)
// (one wave per output column; both layers in one launch)
struct TmWbJobs { const float* W[2]; const float* beta[2]; float* out[2]; const float* gamma[2]; float* wsum[2]; };
__global__ __launch_bounds__(256) void tm_wbeta_kernel(TmWbJobs J, int n_out, int K) {
    const int lane = threadIdx.x & 63, job = blockIdx.y;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_out) return;
    const float* w = J.W[job] + (size_t)c * K;
    const float* b = J.beta[job];
    const float* gm = J.gamma[job];
    float t = 0.0f, u = 0.0f;
    for (int k = lane; k < K; k += 64) { t = fmaf(w[k], b[k], t); u = fmaf(w[k], gm[k], u); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { t += __shfl_xor(t, o); u += __shfl_xor(u, o); }
    if (lane == 0) { J.out[job][c] = t; J.wsum[job][c] = u; }
}

// landmark means from the per-wave-tile column sums the to_qkv launch left (LinArgs::lm_part): QL / KL [h][m][d] = (1 / l) * sum of the
// partials of the 32-row tiles that overlap rows [j l, (j + 1) l), tiles in index order (fixed order: bitwise reproducible).
// Tile t (rows 32 t ..) belongs with part 0 to landmark (32 t) / l and with part 1 to the next one (l >= 32: at most two).
__global__ __launch_bounds__(256) void tm_landmark_reduce_kernel(const float* __restrict__ part, int l, int m, int Di, float* __restrict__ QL,
                                                                float* __restrict__ KL, unsigned* __restrict__ scal_zero) {
    const int j = blockIdx.x, cols = 2 * Di, d = Di / TM_HEADS;
    const int c = blockIdx.y * 256 + threadIdx.x;
    // (the two atomic-max words of this layer's Moore-Penrose scaling, where no statistics pass is left to zero them)
    if (scal_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) scal_zero[threadIdx.x] = 0u;
    if (c >= cols) return;
    const int t0 = (int)(((long long)j * l) >> 5), t1 = (int)((((long long)(j + 1) * l) - 1) >> 5);
    const float inv = 1.0f / (float)l;
    float s = 0.0f;
    int t = t0;
    for (; t + 7 <= t1; t += 8) {          // 8 independent loads in flight, added in tile order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(t + u) * 2 + (((32 * (t + u)) / l == j) ? 0 : 1)) * cols + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; t <= t1; ++t) s += part[((size_t)t * 2 + (((32 * t) / l == j) ? 0 : 1)) * cols + c];
    const int cc = c < Di ? c : c - Di;
    (c < Di ? QL : KL)[((size_t)(cc / d) * m + j) * d + cc % d] = s * inv;
}

// token assembly after fc1: cls row, wrap-around rows (repeat the first tokens), zero front padding
// X2: the second token buffer (PPEG output = layer 2's input): its front padding rows are zeroed here too -- nothing else ever writes
// them, and the LayerNorm-folded to_qkv reads them as x * 0 + 0, which is NaN for whatever NaN / inf bit patterns a recycled
// workspace holds (found in round 4 by running the forward on a workspace pre-filled with 0xFF bytes: layer 2 and the logits were NaN)
__global__ void tm_assemble_kernel(float* __restrict__ X, float* __restrict__ X2, int pad, int N, int nsq, int dim, const float* __restrict__ cls,
                                   float* __restrict__ guard_lo, float* __restrict__ guard_hi, int guard_n) {
    // the zero guard rows of QKV (once per forward: nothing else writes them)
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < guard_n; e += gridDim.x * blockDim.x) { guard_lo[e] = 0.0f; guard_hi[e] = 0.0f; }
    const long long total = (long long)(pad + 1 + (nsq - N)) * dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / dim; const int c = e % dim;
        if (r < pad) { X[r * dim + c] = 0.0f; X2[r * dim + c] = 0.0f; }
        else if (r == pad) X[r * dim + c] = cls[c];
        else { const long long j = r - pad - 1; X[((long long)pad + 1 + N + j) * dim + c] = X[((long long)pad + 1 + j) * dim + c]; }
    }
}

// landmark means of q and k: QL/KL [h][m][d] = mean over l consecutive rows of the q / k column slices of QKV [npad][3Di].
// grid (m, 2): workgroup = (landmark j, q or k half); thread = (float4 column, row phase): `phases` threads share a column and
// take rows phase, phase + phases, ... (float4 loads, 4 in flight), then a fixed-order LDS reduce over the phases.
__global__ __launch_bounds__(1024) void tm_landmark_kernel(const float* __restrict__ qkv, int l, int m, int Di, int phases,
                                                          float* __restrict__ QL, float* __restrict__ KL, unsigned* __restrict__ scal_zero) {
    extern __shared__ __attribute__((aligned(16))) float lm_red[];   // [phases][Di]
    const int j = blockIdx.x, half = blockIdx.y;
    // the two atomic-max words of this layer's Moore-Penrose scaling (tm_pinv_maxsum_kernel, two launches further down the stream)
    if (scal_zero && j == 0 && half == 0 && threadIdx.x < 2) scal_zero[threadIdx.x] = 0u;
    const int c4n = Di / 4;
    const int c4 = threadIdx.x % c4n, ph = threadIdx.x / c4n;
    const int d = Di / TM_HEADS;
    if (ph < phases) {
        const f32x4* src = (const f32x4*)(qkv + (size_t)j * l * 3 * Di + (size_t)half * Di) + c4;
        const size_t rs = (size_t)3 * Di / 4;      // row stride in float4
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        int t = ph;
        for (; t + 3 * phases < l; t += 4 * phases) {
            const f32x4 v0 = src[(size_t)t * rs], v1 = src[(size_t)(t + phases) * rs], v2 = src[(size_t)(t + 2 * phases) * rs],
                        v3 = src[(size_t)(t + 3 * phases) * rs];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; t < l; t += phases) s += src[(size_t)t * rs];
        *(f32x4*)(lm_red + (size_t)ph * Di + 4 * c4) = s;
    }
    __syncthreads();
    const float inv = 1.0f / (float)l;
    for (int c = threadIdx.x; c < Di; c += blockDim.x) {
        float s = 0.0f;
        for (int p2 = 0; p2 < phases; ++p2) s += lm_red[(size_t)p2 * Di + c];
        (half ? KL : QL)[((size_t)(c / d) * m + j) * d + c % d] = s * inv;
    }
}

// in-place softmax over rows of length cols <= 1024 (one wave per row, values kept in registers)
template <int VPL>
__global__ __launch_bounds__(256) void tm_softmax_short_kernel(float* __restrict__ x, long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* p = x + r * cols;
    float v[VPL], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { const int c = lane + 64 * i; v[i] = c < cols ? p[c] : -INFINITY; mx = fmaxf(mx, v[i]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i] = (lane + 64 * i < cols) ? __expf(v[i] - mx) : 0.0f; s += v[i]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { const int c = lane + 64 * i; if (c < cols) p[c] = v[i] * inv; }
}

// sim2 = softmax_j(scale q_l k_l^T) [H][m][m] in one launch: one wave per (head, row); lane j + 64 c forms its dot product over the
// d features with plain fp32 FMAs (K = d = 48 at cfg4: 144 FMAs per lane), then the row softmax as tm_softmax_short_kernel.  Was a
// batched GEMM launch (14.5 us of launch latency for 28 MFLOP) + a softmax launch.
template <int VPL>
__global__ __launch_bounds__(256) void tm_sim2_softmax_kernel(const float* __restrict__ QL, const float* __restrict__ KL, float* __restrict__ S2,
                                                             int m, int d, float scale) {
    __shared__ float qs[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + wave;      // row index over [H][m]; 2 or 4 rows per workgroup
    const int h = (int)(r / m);
    const bool live = r < (long long)TM_HEADS * m;
    const float* q = QL + (size_t)(live ? r : 0) * d;
    for (int k = lane; k < d; k += 64) qs[wave][k] = q[k];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (!live) return;
    const float* Kh = KL + (size_t)h * m * d;
    float v[VPL], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
        const int j = lane + 64 * c;
        float acc = 0.0f;
        if (j < m) {
            const f32x4* kr = (const f32x4*)(Kh + (size_t)j * d);
            for (int k4 = 0; k4 < d / 4; ++k4) {
                const f32x4 kv = kr[k4];
                acc = fmaf(qs[wave][4 * k4], kv[0], acc); acc = fmaf(qs[wave][4 * k4 + 1], kv[1], acc);
                acc = fmaf(qs[wave][4 * k4 + 2], kv[2], acc); acc = fmaf(qs[wave][4 * k4 + 3], kv[3], acc);
            }
        }
        v[c] = j < m ? acc * scale : -INFINITY;
        mx = fmaxf(mx, v[c]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) { v[c] = (lane + 64 * c < m) ? __expf(v[c] - mx) : 0.0f; sum += v[c]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    float* out = S2 + (size_t)r * m;
#pragma unroll
    for (int c = 0; c < VPL; ++c) { const int j = lane + 64 * c; if (j < m) out[j] = v[c] * inv; }
}

// W2 = attn2^+ (attn3 v) per head: [m x m] x [m x d] with exact fp32 products (v_mfma_f32_16x16x4_f32), one wave per 16 x 16 output
// tile, operands straight from L2 (both are a few hundred KB).  Was a batched generic-GEMM launch: 14.5 us of latency for 28 MFLOP.
typedef float tm_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void tm_w2_kernel(const float* __restrict__ Z, const float* __restrict__ AV, float* __restrict__ W2, int m, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.y;
    const int ct = d / 16, t = blockIdx.x * 4 + wave;
    if (t >= (m / 16) * ct) return;
    const int i0 = (t / ct) * 16, e0 = (t % ct) * 16;
    const int li = lane & 15, lk = lane >> 4;
    const float* zr = Z + ((size_t)h * m + i0 + li) * m + lk;            // A[i][k]: lane (i = lane & 15, k = lane >> 4)
    const float* av = AV + ((size_t)h * m + lk) * d + e0 + li;           // B[k][j]: lane (k = lane >> 4, j = lane & 15)
    tm_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = 0; s0 < m / 4; s0 += 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int s = s0 + u; a[u] = s < m / 4 ? zr[4 * s] : 0.0f; b[u] = s < m / 4 ? av[(size_t)4 * s * d] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    // C/D: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
    for (int r = 0; r < 4; ++r) W2[((size_t)h * m + i0 + 4 * lk + r) * d + e0 + li] = acc[r];
}

// in-place softmax over long rows (one workgroup of 1024 threads per row; 3 passes, the row stays in L2)
__global__ __launch_bounds__(1024) void tm_softmax_long_kernel(float* __restrict__ x, int cols) {
    __shared__ float red[16];
    float* p = x + (size_t)blockIdx.x * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int c = tid; c < cols; c += 1024) mx = fmaxf(mx, p[c]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float s = 0.0f;
    for (int c = tid; c < cols; c += 1024) { const float e = __expf(p[c] - mx); p[c] = e; s += e; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float inv = 1.0f / tot;
    for (int c = tid; c < cols; c += 1024) p[c] *= inv;
}

// pinv init, pass 1: global max over heads of the row sums and of the column sums of |x| (x >= 0 after softmax).
// scal[0] = max_i sum_j |x_ij| ("col" in the reference), scal[1] = max_j sum_i |x_ij| ("row"); uint-ordered atomics.
// One workgroup per head: column sums with one thread per column (coalesced row reads), row sums with one wave per row; the
// first version walked columns with a stride of m floats per lane and took 37 us for 1.2 MB.
__global__ __launch_bounds__(1024) void tm_pinv_maxsum_kernel(const float* __restrict__ x, int m, unsigned* __restrict__ scal) {
    __shared__ float red[2][16];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* X = x + (size_t)h * m * m;
    float cs = 0.0f;                                   // column tid
    if (tid < m) {
#pragma unroll 8
        for (int i = 0; i < m; ++i) cs += fabsf(X[(size_t)i * m + tid]);
    }
    float rmax = 0.0f;                                 // rows wave, wave + 16, ...
    for (int i = wave; i < m; i += 16) {
        float rs = 0.0f;
        for (int j = lane; j < m; j += 64) rs += fabsf(X[(size_t)i * m + j]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) rs += __shfl_xor(rs, o);
        rmax = fmaxf(rmax, rs);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cs = fmaxf(cs, __shfl_xor(cs, o));
    if (lane == 0) { red[0][wave] = rmax; red[1][wave] = cs; }
    __syncthreads();
    if (tid == 0) {
        float r = 0.0f, c = 0.0f;
        for (int w = 0; w < 16; ++w) { r = fmaxf(r, red[0][w]); c = fmaxf(c, red[1][w]); }
        atomicMax(scal + 0, __float_as_uint(r)); atomicMax(scal + 1, __float_as_uint(c));
    }
}

// The same two maxima with short workgroups (round 4: the chain runs beside the attn3 leg on the side stream, where only workgroups
// of at most two waves find room -- tools/coresident_probe.hip --, and 192 dependent loads per thread made the kernel above 15 us
// of latency).  grid (ceil(m / 16), H), 128 threads: column sums of 16 columns -- thread (column c = tid & 15, row group g = tid >> 4)
// adds rows g, g + 8, ... in index order, the 8 partials are added in group order (fixed order: reproducible) --, and the row sums
// of 16 rows, one wave per eight rows.
__global__ __launch_bounds__(128) void tm_pinv_maxsum2_kernel(const float* __restrict__ x, int m, unsigned* __restrict__ scal) {
    __shared__ float part[8][17];
    __shared__ float rmx[2];
    const int h = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* X = x + (size_t)h * m * m;
    {
        const int c = 16 * b + (tid & 15), g = tid >> 4;
        float cs = 0.0f;
        if (c < m) {
#pragma unroll 8
            for (int i = g; i < m; i += 8) cs += fabsf(X[(size_t)i * m + c]);
        }
        part[g][tid & 15] = cs;
    }
    float rmax = 0.0f;
    for (int q = 0; q < 8; ++q) {
        const int i = 16 * b + 8 * wave + q;
        float rs = 0.0f;
        if (i < m)
            for (int j = lane; j < m; j += 64) rs += fabsf(X[(size_t)i * m + j]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) rs += __shfl_xor(rs, o);
        rmax = fmaxf(rmax, rs);
    }
    if (lane == 0) rmx[wave] = rmax;
    __syncthreads();
    if (tid < 16) {
        float cs = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) cs += part[g][tid];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) cs = fmaxf(cs, __shfl_xor(cs, o));
        if (tid == 0) { atomicMax(scal + 0, __float_as_uint(fmaxf(rmx[0], rmx[1]))); atomicMax(scal + 1, __float_as_uint(cs)); }
    }
}

// pass 2: z = x^T / (scal0 * scal1)
__global__ void tm_pinv_init_kernel(const float* __restrict__ x, int m, const unsigned* __restrict__ scal, float* __restrict__ z) {
    const float inv = 1.0f / (__uint_as_float(scal[0]) * __uint_as_float(scal[1]));
    const size_t per = (size_t)m * m, total = per * TM_HEADS;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t h = e / per, r = e % per; const int i = r / m, j = r % m;
        z[e] = x[h * per + (size_t)j * m + i] * inv;
    }
}

// out[i][h*d+dd] += sum_t w[h][t] * v[i + t - 16][h*d+dd]   (v = third column block of QKV, zero outside [0, npad))
#define TM_CONV_ROWS 64
__global__ __launch_bounds__(256) void tm_seqconv_kernel(const float* __restrict__ qkv, float* __restrict__ out, int npad, int Di,
                                                        const float* __restrict__ w) {
    __shared__ __attribute__((aligned(16))) float tile[(TM_CONV_ROWS + TM_RES - 1) * 64];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * TM_CONV_ROWS;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int d = Di / TM_HEADS;
    // halo tile -> LDS with float4 loads over the channels (16 lanes per row: one wave instruction = 4 rows x 256 B)
#pragma unroll 4
    for (int idx = threadIdx.x; idx < (TM_CONV_ROWS + TM_RES - 1) * 16; idx += 256) {
        const int rr = idx >> 4, c4 = idx & 15;
        const int r = r0 + rr - TM_RES / 2;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (r >= 0 && r < npad && c0 + 4 * c4 < Di) v = *(const f32x4*)(qkv + (size_t)r * 3 * Di + 2 * Di + c0 + 4 * c4);
        *(f32x4*)(tile + rr * 64 + 4 * c4) = v;
    }
    __syncthreads();
    if (c0 + cl >= Di) return;
    const float* wh = w + (size_t)((c0 + cl) / d) * TM_RES;
    float wr[TM_RES];
#pragma unroll
    for (int t = 0; t < TM_RES; ++t) wr[t] = wh[t];
    // register blocking: 4 consecutive rows per pass share a 36-value window (36 instead of 132 LDS reads per 132 FMAs);
    // wave rg takes rows 16 rg .. 16 rg + 15
#pragma unroll 1
    for (int blk = 0; blk < TM_CONV_ROWS / 16; ++blk) {
        const int rr0 = 16 * rg + 4 * blk;
        if (r0 + rr0 >= npad) break;
        float in[TM_RES + 3];
#pragma unroll
        for (int i = 0; i < TM_RES + 3; ++i) in[i] = tile[(rr0 + i) * 64 + cl];
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < TM_RES; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wr[t], in[j + t], acc[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + rr0 + j < npad) out[(size_t)(r0 + rr0 + j) * Di + c0 + cl] += acc[j];
    }
}

// PPEG weights: weff[tap 0..48][c] = w7 + pad(w5) + pad(w3) + identity at the centre ; beff[c] = b7 + b5 + b3.  One thread per
// (tap, channel) element (a thread per channel walked its 49 + 25 + 9 taps as a chain of dependent loads: 13.8 us); the same launch
// passes the cls row through (transMIL.py:39: cls_token is not convolved) -- it was a launch of its own.
__global__ __launch_bounds__(256) void tm_ppeg_pack_kernel(const float* w7, const float* b7, const float* w5, const float* b5, const float* w3,
                                                          const float* b3, int C, float* weff, float* beff, const float* cls_in, float* cls_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 49 * C) return;
    const int tap = e / C, c = e - tap * C;
    const int ky = tap / 7, kx = tap - 7 * ky;
    float v = w7[(size_t)c * 49 + tap];
    if (ky >= 1 && ky <= 5 && kx >= 1 && kx <= 5) v += w5[(size_t)c * 25 + (ky - 1) * 5 + (kx - 1)];
    if (ky >= 2 && ky <= 4 && kx >= 2 && kx <= 4) v += w3[(size_t)c * 9 + (ky - 2) * 3 + (kx - 2)];
    if (tap == 24) v += 1.0f;
    weff[(size_t)tap * C + c] = v;
    if (tap == 0) { beff[c] = b7[c] + b5[c] + b3[c]; if (cls_in) cls_out[c] = cls_in[c]; }
}

// the end of the forward: LayerNorm of the cls row and the class logits in ONE launch (were a LayerNorm launch + a 1 x C GEMM: 19 us)
__global__ __launch_bounds__(256) void tm_cls_head_kernel(const float* __restrict__ x, int dim, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ W,
                                                         const float* __restrict__ b, int C, float* __restrict__ logits) {
    __shared__ float ln[1024];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[4], s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; v[i] = c < dim ? x[c] : 0.0f; s += v[i]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / dim;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = (tid + 256 * i < dim) ? v[i] - mean : 0.0f; q = fmaf(d, d, q); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / dim + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; if (c < dim) ln[c] = (v[i] - mean) * rstd * gamma[c] + beta[c]; }
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
        float t = 0.0f;
        for (int d = lane; d < dim; d += 64) t = fmaf(ln[d], W[(size_t)c * dim + d], t);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) logits[c] = t + b[c];
    }
}

// depth-wise 7x7 on the [side x side] token grid, channels-last (token p = y*side + x lives at row p of `in`).
// One workgroup = 64 channels x an 8 x 8 pixel tile: the 14 x 14 halo tile is staged in LDS (lane = channel, so
// every LDS access is conflict-free and every global row read is a coalesced 256-B segment); thread (c, g)
// computes rows 2g, 2g+1 of the tile with its 49 taps in registers.
#define TM_PT 8
// cls_in / cls_out (or null): the cls row the stencil passes through (transMIL.py:39-44), copied by the first tile's workgroups
__global__ __launch_bounds__(256) void tm_ppeg_kernel(const float* __restrict__ in, float* __restrict__ out, int side, int C,
                                                     const float* __restrict__ weff, const float* __restrict__ beff,
                                                     const float* __restrict__ cls_in, float* __restrict__ cls_out) {
    __shared__ __attribute__((aligned(16))) float tile[(TM_PT + 6) * (TM_PT + 6) * 64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (cls_in && blockIdx.y == 0 && grp == 0 && c < C) cls_out[c] = cls_in[c];
    const int tiles_x = (side + TM_PT - 1) / TM_PT;
    const int ty0 = (blockIdx.y / tiles_x) * TM_PT, tx0 = (blockIdx.y % tiles_x) * TM_PT;
    // halo tile -> LDS: thread = (pixel e, float4 of channels); 16 lanes cover the 64 channels of a pixel, so one wave
    // instruction fetches 4 pixels x 256 B (4x fewer, 4x wider loads than one channel per lane); C % 4 == 0
    {
        const int cb = blockIdx.x * 64;
#pragma unroll 4
        for (int idx = threadIdx.x; idx < (TM_PT + 6) * (TM_PT + 6) * 16; idx += 256) {
            const int e = idx >> 4, c4 = idx & 15;
            const int yy = ty0 + e / (TM_PT + 6) - 3, xx = tx0 + e % (TM_PT + 6) - 3;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (cb + 4 * c4 < C && yy >= 0 && yy < side && xx >= 0 && xx < side)
                v = *(const f32x4*)(in + ((size_t)yy * side + xx) * C + cb + 4 * c4);
            *(f32x4*)(tile + e * 64 + 4 * c4) = v;
        }
    }
    __syncthreads();
    if (c >= C) return;
    float w[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) w[t] = weff[(size_t)t * C + c];
    const float b = beff[c];
    // register blocking: a thread produces whole rows of 8 pixels (rows 2 grp, 2 grp + 1): per kernel row it reads the 14
    // halo values once and reuses each for up to 7 taps -> 196 instead of 784 LDS reads per 16 outputs
#pragma unroll
    for (int r = 0; r < TM_PT / 4; ++r) {
        const int py = (TM_PT / 4) * grp + r, y = ty0 + py;
        if (y >= side) continue;
        float acc[TM_PT];
#pragma unroll
        for (int px = 0; px < TM_PT; ++px) acc[px] = b;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            float in[TM_PT + 6];
#pragma unroll
            for (int i = 0; i < TM_PT + 6; ++i) in[i] = tile[((py + ky) * (TM_PT + 6) + i) * 64 + cl];
#pragma unroll
            for (int px = 0; px < TM_PT; ++px)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) acc[px] = fmaf(w[ky * 7 + kx], in[px + kx], acc[px]);
        }
#pragma unroll
        for (int px = 0; px < TM_PT; ++px)
            if (tx0 + px < side) out[((size_t)y * side + tx0 + px) * C + c] = acc[px];
    }
}

// The same stencil with PACKED fp32 FMAs (round 4): a thread owns a channel PAIR and one row of the 8 x 8 pixel tile, so every
// operand is a natural 8-byte pair -- the halo values of two adjacent channels (one ds_read_b64), their two weights -- and a tap is one
// v_pk_fma_f32 for two outputs: half the VALU instructions and half the LDS instructions of the scalar kernel above
// (1.9 GFMA per slide were 181 us there: VALU issue, not HBM -- 1.7 TB/s).  C % 64 == 0 (else the scalar kernel).
// (In fact a channel QUAD per thread and half a pixel row: see the store note inside.)
__global__ __launch_bounds__(256, 3) void tm_ppeg2_kernel(const float* __restrict__ in, float* __restrict__ out, int side, int C,
                                                      const float* __restrict__ weff, const float* __restrict__ beff,
                                                      const float* __restrict__ cls_in, float* __restrict__ cls_out) {
    __shared__ __attribute__((aligned(16))) float tile[(TM_PT + 6) * (TM_PT + 6) * 64];
    if (cls_in && blockIdx.y == 0 && threadIdx.x < 64) cls_out[blockIdx.x * 64 + threadIdx.x] = cls_in[blockIdx.x * 64 + threadIdx.x];
    // thread = (channel QUAD cq, tile row py, half of the row's 8 pixels): 4 channels x 4 pixels, so that an output pixel is ONE
    // 16-byte store (a store instruction costs the CU ~75 cycles whatever its width: 8-byte stores were 52 us of this kernel)
    const int cq = threadIdx.x & 15, py = (threadIdx.x >> 4) & 7, ph = threadIdx.x >> 7;
    const int cb = blockIdx.x * 64;
    const int tiles_x = (side + TM_PT - 1) / TM_PT;
    const int ty0 = (blockIdx.y / tiles_x) * TM_PT, tx0 = (blockIdx.y % tiles_x) * TM_PT;
    // ALL of a thread's 13 halo quads in flight at once (round 5).  The loop used to be unrolled by 4: four dependent memory round trips
    // per tile, most of the ~9 us a workgroup spent on a tile with only three workgroups per CU to overlap them -- 136.5 -> 109.0 us
    // per launch on one box; with the next row's weight quads requested before a row is computed (below) 107.3 us.  Halo element
    // e = (tid >> 4) + 16 it: its (row, column) advance by one row + 2 columns per iteration (no division by 14 per element).
    {
        constexpr int HW = TM_PT + 6, HN = HW * HW * 16, HIT = (HN + 255) / 256;
        f32x4 hv[HIT];
        const int c4 = (int)threadIdx.x & 15;
        int hr = (int)(threadIdx.x >> 4) / HW, hc = (int)(threadIdx.x >> 4) % HW;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int idx = (int)threadIdx.x + 256 * it;
            const int yy = ty0 + hr - 3, xx = tx0 + hc - 3;
            hv[it] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (idx < HN && yy >= 0 && yy < side && xx >= 0 && xx < side) hv[it] = *(const f32x4*)(in + ((size_t)yy * side + xx) * C + cb + 4 * c4);
            hr += 1; hc += 16 - HW;
            if (hc >= HW) { hc -= HW; hr += 1; }
        }
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int idx = (int)threadIdx.x + 256 * it;
            if (idx < HN) *(f32x4*)(tile + (idx >> 4) * 64 + 4 * (idx & 15)) = hv[it];
        }
    }
    __syncthreads();
    const int c = cb + 4 * cq, y = ty0 + py, px0 = 4 * ph;
    if (y >= side) return;
    const f32x4 b = *(const f32x4*)(beff + c);
    f32x2 acc[4][2];
#pragma unroll
    for (int px = 0; px < 4; ++px) { acc[px][0] = f32x2{b[0], b[1]}; acc[px][1] = f32x2{b[2], b[3]}; }
    // one kernel row at a time: its 7 weight quads (L1-resident: 49 x C floats per launch) and the 10 halo quads this half row needs.
    // Two register sets: the quads of row ky + 1 are requested before row ky is computed (rolled in pairs -- fully unrolled, hipcc
    // hoisted all 49 loads: 168 VGPRs and 1.3 KB of scratch per lane)
    f32x4 w0[7], w1[7];
    auto load_w = [&](f32x4 (&w)[7], int ky) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) w[kx] = *(const f32x4*)(weff + (size_t)(ky * 7 + kx) * C + c);
    };
    auto row = [&](const f32x4 (&w)[7], int ky) {
        f32x4 v[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = *(const f32x4*)(tile + ((py + ky) * (TM_PT + 6) + px0 + i) * 64 + 4 * cq);
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                acc[px][0] = __builtin_elementwise_fma(f32x2{w[kx][0], w[kx][1]}, f32x2{v[px + kx][0], v[px + kx][1]}, acc[px][0]);
                acc[px][1] = __builtin_elementwise_fma(f32x2{w[kx][2], w[kx][3]}, f32x2{v[px + kx][2], v[px + kx][3]}, acc[px][1]);
            }
    };
    load_w(w0, 0);
#pragma unroll 1
    for (int kp = 0; kp < 3; ++kp) {
        load_w(w1, 2 * kp + 1);
        __builtin_amdgcn_sched_barrier(0);
        row(w0, 2 * kp);
        __builtin_amdgcn_sched_barrier(0);
        load_w(w0, 2 * kp + 2);
        __builtin_amdgcn_sched_barrier(0);
        row(w1, 2 * kp + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    row(w0, 6);
#pragma unroll
    for (int px = 0; px < 4; ++px)
        if (tx0 + px0 + px < side)
            *(f32x4*)(out + ((size_t)y * side + tx0 + px0 + px) * C + c) = f32x4{acc[px][0][0], acc[px][0][1], acc[px][1][0], acc[px][1][1]};
}

__global__ void tm_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) dst[e] = src[e];
}

// ------------------------------------------------------------------------------------------------ host orchestration
struct TmLayerW { const float *norm_w, *norm_b, *qkv_w, *out_w, *out_b, *res_w; };

// fused Nystrom attention legs (transmil_attn.hip)
int tm_attn_fused_supported(int Di);
size_t tm_attn3_partial_bytes(int npad, int Di);
int tm_attn1_fused(const float* QKV, const float* KL, const float* W2, float* OUT, int npad, int Di, float scale, hipStream_t st,
                   const float* convw, int* conv_done);
int tm_attn3_fused(const float* QKV, const float* QL, float* AV, float* part, int npad, int Di, float scale, hipStream_t st, bool shared);

struct TmGeom {
    int N, D, Di, C, side, nsq, n, m, npad, pad, l, d;
};

static TmGeom tm_geom(int N, int D, int Di, int C) {
    TmGeom g; g.N = N; g.D = D; g.Di = Di; g.C = C;
    g.side = (int)ceil(sqrt((double)N));
    while ((long long)g.side * g.side < N) ++g.side;
    while (g.side > 1 && (long long)(g.side - 1) * (g.side - 1) >= N) --g.side;
    g.nsq = g.side * g.side; g.n = g.nsq + 1; g.m = Di / 2;
    g.npad = (g.n + g.m - 1) / g.m * g.m; g.pad = g.npad - g.n; g.l = (g.n + g.m - 1) / g.m; g.d = Di / TM_HEADS;
    return g;
}

static size_t tm_al(size_t b) { return (b + 255) & ~(size_t)255; }

struct TmWs { size_t XA, XB, LN, QKV, S1, S3, OUT, QL, KL, S2, Z, ZT, ZB, ZTB, XZ, T1, T2, LY, WY, AV, W2, WEFF, BEFF, SCAL, PART, PKW, LINWS, AB, LMP, WB, GEMM, total; };
bool tm_pinv_tiles_supported(int m);
int tm_pinv_tiles(const float* X, float* Za, float* ZTa, float* Zb, float* ZTb, float* XZ, float* T1T, float* ST, const unsigned* scal,
                  int m, int iters, float** z_final, hipStream_t st, float* LY, float* WY);

// transmil_pinv.hip: the whole Moore-Penrose iteration of one layer as ONE launch (one workgroup per head); opt-in, see tm_layer
bool tm_pinv_fused_supported(int m);
int tm_pinv_fused(const float* X, float* Z, float* ZT, float* XZ, float* T1T, float* ST, const unsigned* scal, int m, int iters, hipStream_t st);

static TmWs tm_ws(const TmGeom& g) {
    TmWs w; size_t off = 0;
    const size_t tok = tm_al((size_t)g.npad * g.Di * 4), mm = tm_al((size_t)TM_HEADS * g.m * g.m * 4), md = tm_al((size_t)TM_HEADS * g.m * g.d * 4);
    w.XA = off; off += tok; w.XB = off; off += tok; w.LN = off; off += tok; w.OUT = off; off += tok;
    w.QKV = off; off += tm_al((size_t)(g.npad + 2 * TM_QKV_GUARD) * 3 * g.Di * 4);      // (+ the guard rows; QKV itself starts TM_QKV_GUARD rows in)
    w.S1 = off; off += tm_al((size_t)TM_HEADS * g.npad * g.m * 4);
    w.S3 = off; off += tm_al((size_t)TM_HEADS * g.m * g.npad * 4);
    w.QL = off; off += md; w.KL = off; off += md; w.AV = off; off += md; w.W2 = off; off += md;
    w.S2 = off; off += mm; w.Z = off; off += mm; w.ZT = off; off += mm; w.ZB = off; off += mm; w.ZTB = off; off += mm; w.XZ = off; off += mm; w.T1 = off; off += mm; w.T2 = off; off += mm; w.LY = off; off += mm; w.WY = off; off += mm;
    w.WEFF = off; off += tm_al((size_t)49 * g.Di * 4); w.BEFF = off; off += tm_al((size_t)g.Di * 4);
    w.SCAL = off; off += 256;
    w.PART = off; off += tm_al(tm_attn3_partial_bytes(g.npad, g.Di));   // chunk partials of the fused attn3 leg
    // packed-weight Linear kernel (linear.hip): the fragment streams of all five Linear layers (fc1; to_qkv, to_out of both layers:
    // PKW + {0, fc1, fc1 + qkv, fc1 + qkv + out, ...}), packed by ONE launch at the start of a forward, + the kernel's counters
    {
        const size_t pk = tm_al((size_t)g.Di * g.D * 4), q = tm_al((size_t)3 * g.Di * g.Di * 4), o = tm_al((size_t)g.Di * g.Di * 4);
        w.PKW = off; off += pk + 2 * (q + o);
        w.LINWS = off; off += 256;
    }
    // LayerNorm-folded to_qkv: row statistics, landmark partials of the q / k columns per 32-row tile, W beta of both layers
    w.AB = off; off += tm_al((size_t)g.npad * 2 * 4);
    w.LMP = off; off += tm_al((size_t)(g.npad / 32 + 1) * 2 * 2 * g.Di * 4);
    w.WB = off; off += tm_al((size_t)4 * 3 * g.Di * 4);      // W beta of both layers, then the row sums of W o gamma of both layers
    // split-K scratch: the largest need over EVERY product of the forward (a small bag with a wide feature vector splits
    // products that never split at slide scale, e.g. fc1 at N = 400, D = 1536)
    const int H = TM_HEADS, m = g.m, d = g.d, np_ = g.npad, Di = g.Di;
    const int shapes[][4] = {{np_, 3 * Di, Di, 1}, {np_, m, d, H}, {m, m, d, H}, {m, m, m, H}, {m, np_, d, H}, {m, d, np_, H},
                             {m, d, m, H}, {np_, d, m, H}, {g.n, Di, Di, 1}, {g.N, Di, g.D, 1}, {1, g.C, Di, 1}};
    size_t gw = 256;
    for (const auto& sh : shapes) {
        const size_t b = acmil_gemm_workspace_bytes(sh[0], sh[1], sh[2], sh[3]);
        if (b > gw) gw = b;
    }
    w.GEMM = off; off += tm_al(gw);
    w.total = off;
    return w;
}

extern "C" size_t acmil_transmil_workspace_bytes(int N, int D, int Di, int C) {
    if (N <= 0 || D <= 0 || Di <= 0 || C <= 0 || Di % 16 != 0) return 0;
    return tm_ws(tm_geom(N, D, Di, C)).total;
}


// ------------------------------------------------------------------------------------------------ side stream (round 4)
// Two chains of a TransLayer are independent between the landmark means and W2 = attn2^+ (attn3 v): the Moore-Penrose chain
// (sim2 softmax -> max sums -> 14 small launches, each bound by its own latency: ~170 us of a mostly idle GPU) and the attn3 leg
// (one matrix-pipe-bound launch + merge, ~150 us).  They run side by side: the chain goes to a library-owned non-blocking
// stream of HIGH priority (its workgroups are short and latency-critical; the attention leg leaves a wave slot per SIMD free),
// forked from / joined to the caller's stream with events.  One side stream and one event pair per device; the enqueue of a
// forward holds a lock, so concurrent callers cannot interleave each other's record / wait pairs.  ACMIL_TM_SIDE_STREAM=0: serial.
struct TmSide { hipStream_t s; hipEvent_t fork, join; int state; };      // state: 0 = untried, 1 = ready, -1 = unavailable
static std::mutex tm_side_mutex;
static TmSide* tm_side(hipStream_t st) {
    static const bool off = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_SIDE_STREAM"); return e && e[0] == '0'; }();
    if (off) return nullptr;
    static TmSide side[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    TmSide& t = side[dev];
    if (t.state == 0) {
        // never create the stream / events while the caller is CAPTURING (a first call inside a graph capture: creation calls may be
        // refused there and would invalidate the capture): that forward runs serially, a later eager call creates them
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (cs != hipStreamCaptureStatusNone) return nullptr;
        int least = 0, greatest = 0;
        t.state = -1;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
        { const char* e = ACMIL_AB_ENV("ACMIL_TM_SIDE_PRIO"); if (e && e[0] == '0') greatest = least; }      // A/B knob: normal priority
        if (hipStreamCreateWithPriority(&t.s, hipStreamNonBlocking, greatest) == hipSuccess &&
            hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t.join, hipEventDisableTiming) == hipSuccess)
            t.state = 1;
        else (void)hipGetLastError();
    }
    return t.state == 1 ? &t : nullptr;
}
// Opt-in (ACMIL_TM_SERIAL=1): one forward at a time ON THE GPU as well -- a forward waits for the previous forward of the process on
// this device, whichever stream that used (one event record per forward, one wait when the stream changed; never under a capture).
// This was the stop-gap while two forwards on different streams corrupted each other; the cause turned out to be a missing barrier in
// lin_kernel (linear_kernel.h: a wave wrote its epilogue scratch into the ring slot a slower wave was still reading fragments from --
// only a second forward's short Moore-Penrose workgroups ever held a wave back that far) and is fixed there, so forwards overlap
// again by default; tests/test_transmil_gpu.py and tools/stress_transmil.py run the two-stream case.
struct TmSerial { hipEvent_t done; hipStream_t last; int state; };      // state: 0 = no event yet, 1 = event exists, 2 = recorded
static TmSerial* tm_serial(hipStream_t st, bool* capturing) {
    static const bool off = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_SERIAL"); return !(e && e[0] == '1'); }();
    static TmSerial ser[64];
    int dev = 0;
    *capturing = false;
    if (off || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) { *capturing = true; return nullptr; }
    TmSerial& t = ser[dev];
    if (t.state == 0) {
        if (hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        t.state = 1;
    }
    return &t;
}
// side stream continues from what `st` holds now
static bool tm_fork(TmSide* sd, hipStream_t st) {
    return sd && hipEventRecord(sd->fork, st) == hipSuccess && hipStreamWaitEvent(sd->s, sd->fork, 0) == hipSuccess;
}
// the side stream's work so far is marked; the caller waits for sd->join where it needs it
static int tm_join_later(TmSide* sd) { return hipEventRecord(sd->join, sd->s) == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH; }
// `st` continues after what the side stream holds now
static int tm_join(TmSide* sd, hipStream_t st) {
    if (hipEventRecord(sd->join, sd->s) != hipSuccess || hipStreamWaitEvent(st, sd->join, 0) != hipSuccess) return ACMIL_ERR_LAUNCH;
    return ACMIL_OK;
}

#define TM_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH; } while (0)
#define TM_GEMM(...) do { int rc_ = acmil_gemm_f32(__VA_ARGS__); if (rc_ != ACMIL_OK) return rc_; } while (0)
// nn.Linear products (activations x weights, both K-contiguous): split-f16 MFMA, ~1e-6 relative; ACMIL_TM_FP32_GEMM=1 keeps them exact
static bool tm_linear_exact() { static const bool v = ACMIL_AB_ENV("ACMIL_TM_FP32_GEMM") != nullptr; return v; }
// Moore-Penrose products: exact fp32 by default; ACMIL_TM_PINV_X3=1 runs them as split-f16 (experiment)
static bool tm_pinv_x3() { static const bool v = ACMIL_AB_ENV("ACMIL_TM_PINV_X3") != nullptr; return v; }
#define TM_PINV_GEMM(...) do { int rc_ = tm_pinv_x3() ? acmil_gemm_f16x3(__VA_ARGS__) : acmil_gemm_f32(__VA_ARGS__); if (rc_ != ACMIL_OK) return rc_; } while (0)
#define TM_LINEAR(...) do { int rc_ = tm_linear_exact() ? acmil_gemm_f32(__VA_ARGS__) : acmil_gemm_f16x3(__VA_ARGS__); if (rc_ != ACMIL_OK) return rc_; } while (0)

extern "C" size_t acmil_linear_packed_bytes(int n_out, int K);
extern "C" int acmil_linear_pack(const float* W, int ldw, int n_out, int K, void* packed, void* stream);
// linear.hip: the packed Linear kernel on control words the CALLER has zeroed (once per forward here); every launch leaves them zero
int lin_f16x3_run(const void* x, int x_dtype, int M, int K, long long ldx, const void* packed, int n_out, const float* bias, int act,
                  float beta, float* y, long long ldy, void* workspace, hipStream_t st, bool init);
int lin_pack_multi(const float* const* W, const int* ldw, const int* n_out, const int* K, void* const* packed, int n, hipStream_t st,
                   const float* const* colscale);
bool lin_qkv_norm_instat_ok(int M, int K, int n_out);
int lin_qkv_norm_run(const float* x, int M, int K, long long ldx, const float* rowab, int zrows, const void* packed, int n_out,
                     const float* bias, float* y, long long ldy, float* lm_part, int lm_l, int lm_cols, void* workspace, hipStream_t st,
                     const float* wsum, int c_lo, int c_hi);
bool lin_qkv_norm_cols_ok(int M, int K, int n_out, int c_lo, int c_hi);

// y = act(x W^T + b) + beta y for the nn.Linear layers (fc1 / to_qkv / to_out, transMIL.py:51,63, nystrom_attention.py:80,139).
// Split-f16: the packed-weight kernel (linear.hip; fragment stream packed here, per call -- the library keeps no state: one small
// launch, ~4 us) where its shape rules hold -- 216 / 216 / 197 TF on the three cfg4 shapes against 199 / 201 / 186 for the generic
// split GEMM; otherwise, and for ACMIL_TM_FP32_GEMM=1 (exact fp32 MFMA), the generic GEMMs.  ACMIL_TM_GENERIC_GEMM=1: A/B knob.
static int tm_linear(const float* x, int M, int K, long long ldx, const float* W, int n_out, const float* bias, int act, float beta,
                     float* y, long long ldy, char* pkw, void* linws, void* gws, hipStream_t st, bool prepacked = false) {
    static const bool generic = ACMIL_AB_ENV("ACMIL_TM_GENERIC_GEMM") != nullptr;
    const bool lin_ok = !tm_linear_exact() && !generic && acmil_linear_packed_bytes(n_out, K) != 0 && ((size_t)x & 15) == 0 &&
                        ((size_t)ldx * 4) % 16 == 0 && ldy >= n_out && ((size_t)y & 15) == 0 && ldy % 4 == 0;
    if (lin_ok) {
        if (!prepacked) {
            const int rc = acmil_linear_pack(W, K, n_out, K, pkw, st);
            if (rc != ACMIL_OK) return rc;
        }
        return lin_f16x3_run(x, ACMIL_DTYPE_F32, M, K, ldx, pkw, n_out, bias, act, beta, y, ldy, linws, st, false);
    }
    TM_LINEAR(0, 1, M, n_out, K, 1.0f, x, (int)ldx, 0, W, ACMIL_DTYPE_F32, K, 0, beta, y, (int)ldy, 0, bias, act, nullptr, 1, gws, st);
    return ACMIL_OK;
}

static int tm_softmax_short(float* x, long long rows, int cols, hipStream_t st) {
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (cols <= 64) hipLaunchKernelGGL(tm_softmax_short_kernel<1>, dim3(blocks), dim3(256), 0, st, x, rows, cols);
    else if (cols <= 128) hipLaunchKernelGGL(tm_softmax_short_kernel<2>, dim3(blocks), dim3(256), 0, st, x, rows, cols);
    else if (cols <= 256) hipLaunchKernelGGL(tm_softmax_short_kernel<4>, dim3(blocks), dim3(256), 0, st, x, rows, cols);
    else if (cols <= 512) hipLaunchKernelGGL(tm_softmax_short_kernel<8>, dim3(blocks), dim3(256), 0, st, x, rows, cols);
    else if (cols <= 1024) hipLaunchKernelGGL(tm_softmax_short_kernel<16>, dim3(blocks), dim3(256), 0, st, x, rows, cols);
    else return ACMIL_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// one TransLayer in place on X [npad][Di] (token i at row pad + i):  X[pad:] += to_out(attention(LayerNorm(X[pad:])))
static int tm_layer(const TmGeom& g, const TmWs& W, char* ws, float* X, const TmLayerW& p, hipStream_t st, char* pk_qkv = nullptr,
                    char* pk_out = nullptr, const float* wbeta = nullptr, TmSide* side = nullptr, const float* wsum = nullptr) {
    const int Di = g.Di, m = g.m, d = g.d, npad = g.npad, H = TM_HEADS;
    float* LN = (float*)(ws + W.LN); float* QKV = (float*)(ws + W.QKV) + (size_t)TM_QKV_GUARD * 3 * g.Di; float* S1 = (float*)(ws + W.S1);
    float* S3 = (float*)(ws + W.S3); float* OUT = (float*)(ws + W.OUT); float* QL = (float*)(ws + W.QL);
    float* KL = (float*)(ws + W.KL); float* S2 = (float*)(ws + W.S2); float* Z = (float*)(ws + W.Z);
    float* XZ = (float*)(ws + W.XZ); float* T1 = (float*)(ws + W.T1); float* T2 = (float*)(ws + W.T2);
    float* AV = (float*)(ws + W.AV); float* W2 = (float*)(ws + W.W2); unsigned* scal = (unsigned*)(ws + W.SCAL);
    void* gws = ws + W.GEMM;
    const float scale = 1.0f / sqrtf((float)d);
    const long long mm = (long long)m * m, md = (long long)m * d;

    // fused = the two long attention legs run as flash-style kernels (no [H, npad, m] matrices in HBM); the GEMM + softmax
    // chain below stays as the path for other widths and as the A/B reference (ACMIL_TM_UNFUSED=1)
    static const bool force_unfused = ACMIL_AB_ENV("ACMIL_TM_UNFUSED") != nullptr;
    const bool fused = tm_attn_fused_supported(Di) && !force_unfused;
    static const bool pinv_knobs = ACMIL_AB_ENV("ACMIL_TM_PINV_FUSED") != nullptr || ACMIL_AB_ENV("ACMIL_TM_PINV_CHAIN") != nullptr;
    const bool can_fork = fused && d % 4 == 0 && d <= 128 && m <= 512 && tm_pinv_tiles_supported(m) && !pinv_knobs;
    bool forked = false, early = false;      // early (round 6): the side stream was forked after [q | k], see below
    bool instat = false;
    if (wbeta) {
        // LayerNorm folded into to_qkv (round 4): row statistics -> the projection normalises its B operand in registers (gamma is in
        // the packed weights, W beta is the bias) and leaves the landmark column sums of q and k per 32-row tile -> a small reduce.
        // Gone: the LayerNorm pass (read + write of [npad, Di]) and the landmark pass (re-read of the q / k columns of QKV).
        float* AB = (float*)(ws + W.AB); float* LMP = (float*)(ws + W.LMP);
        // round 6: the row statistics come out of to_qkv's own K loop (linear_kernel.h, FX & 4) -- no pass over X for them
        instat = wsum != nullptr && lin_qkv_norm_instat_ok(npad, Di, 3 * Di);
        if (!instat) {
            hipLaunchKernelGGL(tm_rowstats_kernel, dim3((npad + 3) / 4), dim3(256), 0, st, X, AB, g.n, Di, g.pad, scal);
            TM_CHECK_LAUNCH();
        }
        const bool lm = g.l >= 32;
        // round 6: to_qkv as TWO launches, [q | k] then v.  The landmarks, sim2 and the Moore-Penrose chain (a third of a layer's
        // critical path, ~25 latency-bound launches that fill a few CUs) need q and k only: they start on the side stream as soon as the
        // first launch retires and run beside the v projection -- a launch that keeps the whole chip busy by itself and does not mind
        // the company -- instead of beside the attn3 leg, which then has the chip to itself for most of its time (two workgroups per
        // CU: 150 us instead of 190).  Price: x is read twice by to_qkv.  ACMIL_TM_QKV_SPLIT=0: one launch (A/B).
        static const bool split_off = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_QKV_SPLIT"); return !(e && e[0] == '1'); }();
        const bool split = instat && lm && can_fork && side && !split_off && lin_qkv_norm_cols_ok(npad, Di, 3 * Di, 0, 2 * Di) &&
                           lin_qkv_norm_cols_ok(npad, Di, 3 * Di, 2 * Di, 3 * Di);
        if (split) {
            int rq = lin_qkv_norm_run(X, npad, Di, Di, nullptr, g.pad, pk_qkv, 3 * Di, wbeta, QKV, 3 * Di, LMP, g.l, 2 * Di, ws + W.LINWS, st, wsum, 0, 2 * Di);
            if (rq != ACMIL_OK) return rq;
            forked = tm_fork(side, st);
            const hipStream_t sl = forked ? side->s : st;
            hipLaunchKernelGGL(tm_landmark_reduce_kernel, dim3(m, (2 * Di + 255) / 256), dim3(256), 0, sl, LMP, g.l, m, Di, QL, KL, scal);
            TM_CHECK_LAUNCH();
            // (the fork event is free again: the side stream's wait on it is enqueued and refers to the record above; recorded anew on
            // the SIDE stream it now says "q_l and k_l are there", which the attn3 leg on the main stream waits for)
            if (forked && hipEventRecord(side->fork, side->s) != hipSuccess) { (void)tm_join(side, st); return ACMIL_ERR_LAUNCH; }
            early = forked;
            rq = lin_qkv_norm_run(X, npad, Di, Di, nullptr, g.pad, pk_qkv, 3 * Di, wbeta, QKV, 3 * Di, nullptr, g.l, 2 * Di, ws + W.LINWS, st, wsum, 2 * Di, 3 * Di);
            if (rq != ACMIL_OK) { if (forked) (void)tm_join(side, st); return rq; }
        } else {
        const int rq = lin_qkv_norm_run(X, npad, Di, Di, instat ? nullptr : AB, g.pad, pk_qkv, 3 * Di, wbeta, QKV, 3 * Di, lm ? LMP : nullptr, g.l, 2 * Di,
                                        ws + W.LINWS, st, instat ? wsum : nullptr, 0, 3 * Di);
        if (rq != ACMIL_OK) return rq;
        if (lm) {
            hipLaunchKernelGGL(tm_landmark_reduce_kernel, dim3(m, (2 * Di + 255) / 256), dim3(256), 0, st, LMP, g.l, m, Di, QL, KL, instat ? scal : nullptr);
            TM_CHECK_LAUNCH();
        }
        }
    } else {
    hipLaunchKernelGGL(tm_layernorm_kernel, dim3((npad + 3) / 4), dim3(256), 0, st, X, LN, g.n, Di, p.norm_w, p.norm_b, g.pad);
    TM_CHECK_LAUNCH();
    // qkv projection (no bias): [npad, 3Di]
    { const int rq = tm_linear(LN, npad, Di, Di, p.qkv_w, 3 * Di, nullptr, 0, 0.0f, QKV, 3 * Di, pk_qkv ? pk_qkv : ws + W.PKW, ws + W.LINWS, gws, st, pk_qkv != nullptr); if (rq != ACMIL_OK) return rq; }
    }
    if (!wbeta || g.l < 32) {
        int phases = 1024 / (Di / 4); if (phases > 8) phases = 8; if (phases > g.l) phases = g.l; if (phases < 1) phases = 1;
        const int threads = ((Di / 4) * phases + 63) / 64 * 64;
        hipLaunchKernelGGL(tm_landmark_kernel, dim3(m, 2), dim3(threads), (size_t)phases * Di * sizeof(float), st, QKV, g.l, m, Di, phases, QL, KL, (wbeta && !instat) ? nullptr : scal);
    }
    TM_CHECK_LAUNCH();
    int rc = ACMIL_OK;
    if (!fused) {
    // sim1 = scale q k_l^T  [H, npad, m] ; softmax over m
    TM_GEMM(0, 1, npad, m, d, scale, QKV, 3 * Di, d, KL, ACMIL_DTYPE_F32, d, md, 0.0f, S1, m, (long long)npad * m, nullptr, 0, nullptr, H, gws, st);
    rc = tm_softmax_short(S1, (long long)H * npad, m, st); if (rc != ACMIL_OK) return rc;
    }
    // sim2 = scale q_l k_l^T [H, m, m] ; softmax ; Moore-Penrose iteration -- on the side stream beside the attn3 leg where both
    // run on their own kernels and scratch (the generic-GEMM paths share the split-K workspace: serial)
    const hipStream_t st_main = st;
    if (!early) forked = can_fork && tm_fork(side, st);
    static const bool side_swap_knob = ACMIL_AB_ENV("ACMIL_TM_SIDE_SWAP") != nullptr;   // A/B knob: the attn3 leg on the side stream instead
    const bool side_swap = side_swap_knob && !early;
    const hipStream_t st_chain = (forked && !side_swap) ? side->s : st_main, st_leg = (forked && side_swap) ? side->s : st_main;
    st = st_chain;
    if (d % 4 == 0 && d <= 128 && m <= 512) {
        // two-wave workgroups: beside the attn3 leg (two 6-wave workgroups of 128 VGPRs per CU = 4, 4, 2, 2 waves on the four SIMDs) a
        // workgroup of more than two waves finds no room until that launch retires (tools/coresident_probe.hip)
        const unsigned blocks = (unsigned)(((long long)H * m + 1) / 2);
        if (m <= 64) hipLaunchKernelGGL(tm_sim2_softmax_kernel<1>, dim3(blocks), dim3(128), 0, st, QL, KL, S2, m, d, scale);
        else if (m <= 128) hipLaunchKernelGGL(tm_sim2_softmax_kernel<2>, dim3(blocks), dim3(128), 0, st, QL, KL, S2, m, d, scale);
        else if (m <= 256) hipLaunchKernelGGL(tm_sim2_softmax_kernel<4>, dim3(blocks), dim3(128), 0, st, QL, KL, S2, m, d, scale);
        else hipLaunchKernelGGL(tm_sim2_softmax_kernel<8>, dim3(blocks), dim3(128), 0, st, QL, KL, S2, m, d, scale);
        TM_CHECK_LAUNCH();
    } else {
        TM_GEMM(0, 1, m, m, d, scale, QL, d, md, KL, ACMIL_DTYPE_F32, d, md, 0.0f, S2, m, mm, nullptr, 0, nullptr, H, gws, st);
        rc = tm_softmax_short(S2, (long long)H * m, m, st); if (rc != ACMIL_OK) return rc;
    }
    static const bool maxsum_old = ACMIL_AB_ENV("ACMIL_TM_MAXSUM_OLD") != nullptr;       // A/B knob: one 16-wave workgroup per head
    if (maxsum_old) hipLaunchKernelGGL(tm_pinv_maxsum_kernel, dim3(H), dim3(1024), 0, st, S2, m, scal);
    else hipLaunchKernelGGL(tm_pinv_maxsum2_kernel, dim3((m + 15) / 16, H), dim3(128), 0, st, S2, m, scal);
    TM_CHECK_LAUNCH();
    float* zc = Z; float* zn = T2;     // ping-pong z
    // ACMIL_TM_PINV_FUSED=1: ONE launch for the 24 products (transmil_pinv.hip: one 8-wave workgroup per head, operands staged
    // through LDS, split-f16).  Correct (the TransMIL suite passes with it), but OPT-IN: 8 workgroups occupy 8 CUs and a 192^3
    // product costs ~16 us there (MFMA floor 4.3 us on one CU + staging / store / barrier latency) against 12.6 us for the
    // launch-per-product chain over 72 workgroups (exact fp32 MFMA) below: 3.70 vs 3.56 ms per N = 100 000 slide.  Also measured
    // without gain: the chain on a second stream beside the fused attention leg (3.60 vs 3.62 ms).
    static const bool pinv_one = ACMIL_AB_ENV("ACMIL_TM_PINV_FUSED") != nullptr;
    static const bool pinv_chain = ACMIL_AB_ENV("ACMIL_TM_PINV_CHAIN") != nullptr;       // A/B knob: the generic-GEMM chain
    bool pinv_fused = tm_pinv_fused_supported(m) && pinv_one;
    if (!pinv_fused && !pinv_chain && tm_pinv_tiles_supported(m)) {
        // default: one wave per 16 x 16 output tile, whole K in registers, exact fp32 MFMA (transmil_pinv.hip): 24 + 1 launches of ~3 us
        // ACMIL_TM_ABL (A/B build, timing only, WRONG results): bit 0 no attn3 leg, bit 1 no Moore-Penrose products, bit 2 no attn1 leg
        static const int abl = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_ABL"); return e ? atoi(e) : 0; }();
        if (!(abl & 2)) rc = tm_pinv_tiles(S2, Z, (float*)(ws + W.ZT), (float*)(ws + W.ZB), (float*)(ws + W.ZTB), XZ, T1, T2, scal, m, 6, &zc, st,
                           (float*)(ws + W.LY), (float*)(ws + W.WY));
        if (rc != ACMIL_OK) return rc;
        pinv_fused = true;       // (skips the generic chain below)
    } else if (pinv_fused) {
        rc = tm_pinv_fused(S2, Z, (float*)(ws + W.ZT), XZ, T1, T2, scal, m, 6, st);
        if (rc != ACMIL_OK) return rc;
    } else {
        hipLaunchKernelGGL(tm_pinv_init_kernel, dim3(256), dim3(256), 0, st, S2, m, scal, Z);
        TM_CHECK_LAUNCH();
    }
    for (int it = 0; it < (pinv_fused ? 0 : 6); ++it) {
        float* spare = (zc == Z) ? T2 : Z;
        // xz = x z ; t1 = 7I - xz ; t = 15I - xz t1 ; t1 = 13I - xz t ; z' = 0.25 z t1
        TM_PINV_GEMM(0, 0, m, m, m, 1.0f, S2, m, mm, zc, ACMIL_DTYPE_F32, m, mm, 7.0f, XZ, m, mm, nullptr, 4, T1, H, gws, st);   // + t1 = 7I - xz
        TM_PINV_GEMM(0, 0, m, m, m, 1.0f, XZ, m, mm, T1, ACMIL_DTYPE_F32, m, mm, 15.0f, spare, m, mm, nullptr, 3, nullptr, H, gws, st);
        TM_PINV_GEMM(0, 0, m, m, m, 1.0f, XZ, m, mm, spare, ACMIL_DTYPE_F32, m, mm, 13.0f, T1, m, mm, nullptr, 3, nullptr, H, gws, st);
        TM_PINV_GEMM(0, 0, m, m, m, 0.25f, zc, m, mm, T1, ACMIL_DTYPE_F32, m, mm, 0.0f, spare, m, mm, nullptr, 0, nullptr, H, gws, st);
        zn = spare; float* t = zc; zc = zn; zn = t;
    }
    st = st_main;
    if (fused) {
        // AV = softmax_n(scale q_l k^T) v  [H, m, d]   (chunk partials in their own workspace region)
        // early fork: q_l comes from the side stream's landmark launch; most of the chain is over when v is there, the leg runs in
        // the form it has alone on the chip (ACMIL_TM_ATTN3_SHARED=1: the one-workgroup-per-CU form of the late fork, A/B)
        static const bool early_shared = ACMIL_AB_ENV("ACMIL_TM_ATTN3_SHARED") != nullptr;
        if (early && hipStreamWaitEvent(st_leg, side->fork, 0) != hipSuccess) { (void)tm_join(side, st); return ACMIL_ERR_LAUNCH; }
        static const int abl3 = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_ABL"); return e ? atoi(e) : 0; }();
        if (!(abl3 & 1)) rc = tm_attn3_fused(QKV, QL, AV, (float*)(ws + W.PART), npad, Di, scale, st_leg, forked && (!early || early_shared));
        if (forked) { const int rj = tm_join(side, st); if (rc == ACMIL_OK) rc = rj; }       // (joined even after a failed launch)
        if (rc != ACMIL_OK) return rc;
    } else {
    // sim3 = scale q_l k^T [H, m, npad] ; softmax over npad
    TM_GEMM(0, 1, m, npad, d, scale, QL, d, md, QKV + Di, ACMIL_DTYPE_F32, 3 * Di, d, 0.0f, S3, npad, (long long)m * npad, nullptr, 0, nullptr, H, gws, st);
    if (npad <= 1024) { rc = tm_softmax_short(S3, (long long)H * m, npad, st); if (rc != ACMIL_OK) return rc; }
    else { hipLaunchKernelGGL(tm_softmax_long_kernel, dim3(H * m), dim3(1024), 0, st, S3, npad); TM_CHECK_LAUNCH(); }
    // AV = attn3 v [H, m, d]
    TM_GEMM(0, 0, m, d, npad, 1.0f, S3, npad, (long long)m * npad, QKV + 2 * Di, ACMIL_DTYPE_F32, 3 * Di, d, 0.0f, AV, d, md, nullptr, 0, nullptr, H, gws, st);
    }
    // W2 = attn2^+ AV ; OUT = attn1 W2 written into the merged-head layout [npad, H*d]
    if (m % 16 == 0 && d % 16 == 0) {
        hipLaunchKernelGGL(tm_w2_kernel, dim3((unsigned)(((m / 16) * (d / 16) + 3) / 4), H), dim3(256), 0, st, zc, AV, W2, m, d);
        TM_CHECK_LAUNCH();
    } else {
        TM_GEMM(0, 0, m, d, m, 1.0f, zc, m, mm, AV, ACMIL_DTYPE_F32, d, md, 0.0f, W2, d, md, nullptr, 0, nullptr, H, gws, st);
    }
    int conv_done = 0;
    static const int abl1 = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_ABL"); return e ? atoi(e) : 0; }();
    if (fused && (abl1 & 4)) conv_done = 1;
    else if (fused) { rc = tm_attn1_fused(QKV, KL, W2, OUT, npad, Di, scale, st, p.res_w, &conv_done); if (rc != ACMIL_OK) return rc; }
    else TM_GEMM(0, 0, npad, d, m, 1.0f, S1, m, (long long)npad * m, W2, ACMIL_DTYPE_F32, d, md, 0.0f, OUT, Di, d, nullptr, 0, nullptr, H, gws, st);
    // + depth-wise residual conv of v along the sequence (a pass of its own only where the attention leg did not add it)
    if (!conv_done) {
        hipLaunchKernelGGL(tm_seqconv_kernel, dim3((Di + 63) / 64, (npad + TM_CONV_ROWS - 1) / TM_CONV_ROWS), dim3(256), 0, st, QKV, OUT, npad, Di, p.res_w);
        TM_CHECK_LAUNCH();
    }
    // X[pad:] += OUT[pad:] Wout^T + b   (only the last n rows are kept by the reference)
    return tm_linear(OUT + (size_t)g.pad * Di, g.n, Di, Di, p.out_w, Di, p.out_b, 0, 1.0f, X + (size_t)g.pad * Di, Di, pk_out ? pk_out : ws + W.PKW, ws + W.LINWS, gws, st,
                     pk_out != nullptr);
}

// The forward proper.  `side` = the second stream + event pair the Moore-Penrose chain / weight packing use beside `stream` (null:
// every launch on `stream`).  Whoever owns `side` guarantees that one forward at a time records / waits on its events.
static int tm_forward_impl(const float* x, int N, int D, int Di, int C, const float* fc1_w, const float* fc1_b,
                           const float* cls_token, const float* const* layer1 /*6 ptrs*/,
                           const float* const* layer2 /*6 ptrs*/, const float* const* ppeg /*w7,b7,w5,b5,w3,b3*/,
                           const float* norm_w, const float* norm_b, const float* fc2_w, const float* fc2_b,
                           float* logits, float* dbg_h1, float* dbg_hp, float* dbg_h2, void* workspace,
                           hipStream_t st, TmSide* side) {
    if (N <= 0 || D <= 0 || Di <= 0 || C <= 0) return ACMIL_ERR_SHAPE;
    if (Di % 16 != 0 || Di / 2 > 1024 || Di > 1024) return ACMIL_ERR_UNSUPPORTED;      // (Di <= 1024: tm_cls_head_kernel's row buffer)
    if (!x || !fc1_w || !fc1_b || !cls_token || !layer1 || !layer2 || !ppeg || !norm_w || !norm_b || !fc2_w || !fc2_b || !logits || !workspace)
        return ACMIL_ERR_NULL;
    for (int i = 0; i < 6; ++i) if (!layer1[i] || !layer2[i] || !ppeg[i]) return ACMIL_ERR_NULL;
    const TmGeom g = tm_geom(N, D, Di, C);
    const TmWs W = tm_ws(g);
    char* ws = (char*)workspace;
    float* XA = (float*)(ws + W.XA); float* XB = (float*)(ws + W.XB); float* LN = (float*)(ws + W.LN);
    float* weff = (float*)(ws + W.WEFF); float* beff = (float*)(ws + W.BEFF);
    void* gws = ws + W.GEMM;
    const size_t tokbytes = (size_t)g.n * Di;
    // control words of the packed Linear launches: zeroed ONCE per forward (every launch leaves its counters at zero again)
    if (hipMemsetAsync(ws + W.LINWS, 0, 96, st) != hipSuccess) return ACMIL_ERR_LAUNCH;        // (LIN_CTRL_BYTES)
    // the fragment streams of the five Linear layers in ONE launch (the library keeps no state between calls: packed per forward)
    char* pk1 = nullptr; char* pkq[2] = {nullptr, nullptr}; char* pko[2] = {nullptr, nullptr};
    bool fold_ln = false, packs_forked = false;
    TmLayerW l1 = {layer1[0], layer1[1], layer1[2], layer1[3], layer1[4], layer1[5]};
    TmLayerW l2 = {layer2[0], layer2[1], layer2[2], layer2[3], layer2[4], layer2[5]};
    {
        static const bool generic = ACMIL_AB_ENV("ACMIL_TM_GENERIC_GEMM") != nullptr;
        if (!tm_linear_exact() && !generic && acmil_linear_packed_bytes(Di, D) && acmil_linear_packed_bytes(3 * Di, Di) && acmil_linear_packed_bytes(Di, Di)) {
            const size_t pk = tm_al((size_t)Di * D * 4), q = tm_al((size_t)3 * Di * Di * 4), o = tm_al((size_t)Di * Di * 4);
            char* base = ws + W.PKW;
            pk1 = base; pkq[0] = base + pk; pko[0] = pkq[0] + q; pkq[1] = pko[0] + o; pko[1] = pkq[1] + q;
            const float* Wp[5] = {fc1_w, l1.qkv_w, l1.out_w, l2.qkv_w, l2.out_w};
            const int ldw[5] = {D, Di, Di, Di, Di}, no[5] = {Di, 3 * Di, Di, 3 * Di, Di}, Kk[5] = {D, Di, Di, Di, Di};
            void* outp[5] = {pk1, pkq[0], pko[0], pkq[1], pko[1]};
            // ACMIL_TM_LN_PASS=1: the round-3 pipeline (LayerNorm pass, plain to_qkv, landmark pass) -- A/B knob and test reference
            static const bool ln_pass = ACMIL_AB_ENV("ACMIL_TM_LN_PASS") != nullptr;
            fold_ln = !ln_pass;
            const float* cs[5] = {nullptr, fold_ln ? l1.norm_w : nullptr, nullptr, fold_ln ? l2.norm_w : nullptr, nullptr};
            // fc1's stream is all the first product needs; the four layer streams and W beta are formed beside it on the side stream
            packs_forked = tm_fork(side, st);
            const hipStream_t sp = packs_forked ? side->s : st;
            int rp = lin_pack_multi(Wp, ldw, no, Kk, outp, packs_forked ? 1 : 5, st, cs);
            if (rp == ACMIL_OK && packs_forked) rp = lin_pack_multi(Wp + 1, ldw + 1, no + 1, Kk + 1, outp + 1, 4, sp, cs + 1);
            if (rp == ACMIL_OK && fold_ln) {
                float* wb = (float*)(ws + W.WB);
                TmWbJobs J = {{l1.qkv_w, l2.qkv_w}, {l1.norm_b, l2.norm_b}, {wb, wb + 3 * Di}, {l1.norm_w, l2.norm_w}, {wb + 6 * Di, wb + 9 * Di}};
                hipLaunchKernelGGL(tm_wbeta_kernel, dim3((3 * Di + 3) / 4, 2), dim3(256), 0, sp, J, 3 * Di, Di);
                if (hipGetLastError() != hipSuccess) rp = ACMIL_ERR_LAUNCH;
            }
            if (rp == ACMIL_OK && packs_forked) {       // the folded PPEG stencil as well (weights only; the cls row is copied by the stencil launch)
                hipLaunchKernelGGL(tm_ppeg_pack_kernel, dim3((49 * Di + 255) / 256), dim3(256), 0, sp, ppeg[0], ppeg[1], ppeg[2], ppeg[3], ppeg[4], ppeg[5], Di,
                                   weff, beff, (const float*)nullptr, (float*)nullptr);
                if (hipGetLastError() != hipSuccess) rp = ACMIL_ERR_LAUNCH;
            }
            if (packs_forked) { const int rj = tm_join_later(side); if (rp == ACMIL_OK) rp = rj; }
            if (rp != ACMIL_OK) return rp;
        }
    }
    // fc1 + relu straight into the token rows, then cls / wrap-around / front padding
    { const int r1 = tm_linear(x, N, D, D, fc1_w, Di, fc1_b, 1, 0.0f, XA + (size_t)(g.pad + 1) * Di, Di, pk1 ? pk1 : ws + W.PKW, ws + W.LINWS, gws, st, pk1 != nullptr); if (r1 != ACMIL_OK) return r1; }
    {
        float* qkv0 = (float*)(ws + W.QKV);
        hipLaunchKernelGGL(tm_assemble_kernel, dim3(512), dim3(256), 0, st, XA, XB, g.pad, N, g.nsq, Di, cls_token, qkv0,
                           qkv0 + (size_t)(TM_QKV_GUARD + g.npad) * 3 * Di, TM_QKV_GUARD * 3 * Di);
    }
    TM_CHECK_LAUNCH();
    const float* wb = (const float*)(ws + W.WB);
    if (packs_forked && hipStreamWaitEvent(st, side->join, 0) != hipSuccess) return ACMIL_ERR_LAUNCH;      // the layer streams are packed
    int rc = tm_layer(g, W, ws, XA, l1, st, pkq[0], pko[0], fold_ln ? wb : nullptr, side, fold_ln ? wb + 6 * Di : nullptr); if (rc != ACMIL_OK) return rc;
    if (dbg_h1) { hipLaunchKernelGGL(tm_copy_kernel, dim3(512), dim3(256), 0, st, XA + (size_t)g.pad * Di, dbg_h1, tokbytes); TM_CHECK_LAUNCH(); }
    // PPEG: cls passthrough + combined depth-wise 7x7 (the folded stencil was packed beside fc1 when the side stream is in use)
    const float* cls_in = XA + (size_t)g.pad * Di; float* cls_out = XB + (size_t)g.pad * Di;
    if (!packs_forked) {
        hipLaunchKernelGGL(tm_ppeg_pack_kernel, dim3((49 * Di + 255) / 256), dim3(256), 0, st, ppeg[0], ppeg[1], ppeg[2], ppeg[3], ppeg[4], ppeg[5], Di, weff, beff,
                           (const float*)nullptr, (float*)nullptr);
        TM_CHECK_LAUNCH();
    }
    {
        const int tiles = (g.side + TM_PT - 1) / TM_PT;
        static const bool ppeg_scalar = ACMIL_AB_ENV("ACMIL_TM_PPEG_SCALAR") != nullptr;       // A/B knob
        if (Di % 64 == 0 && !ppeg_scalar)
            hipLaunchKernelGGL(tm_ppeg2_kernel, dim3(Di / 64, tiles * tiles), dim3(256), 0, st, XA + (size_t)(g.pad + 1) * Di,
                               XB + (size_t)(g.pad + 1) * Di, g.side, Di, weff, beff, cls_in, cls_out);
        else
        hipLaunchKernelGGL(tm_ppeg_kernel, dim3((Di + 63) / 64, tiles * tiles), dim3(256), 0, st, XA + (size_t)(g.pad + 1) * Di,
                           XB + (size_t)(g.pad + 1) * Di, g.side, Di, weff, beff, cls_in, cls_out);
    }
    TM_CHECK_LAUNCH();
    if (dbg_hp) { hipLaunchKernelGGL(tm_copy_kernel, dim3(512), dim3(256), 0, st, XB + (size_t)g.pad * Di, dbg_hp, tokbytes); TM_CHECK_LAUNCH(); }
    rc = tm_layer(g, W, ws, XB, l2, st, pkq[1], pko[1], fold_ln ? wb + 3 * Di : nullptr, side, fold_ln ? wb + 9 * Di : nullptr); if (rc != ACMIL_OK) return rc;
    if (dbg_h2) { hipLaunchKernelGGL(tm_copy_kernel, dim3(512), dim3(256), 0, st, XB + (size_t)g.pad * Di, dbg_h2, tokbytes); TM_CHECK_LAUNCH(); }
    // final LayerNorm on the cls row only, then fc2 (exact fp32 FMAs)
    hipLaunchKernelGGL(tm_cls_head_kernel, dim3(1), dim3(256), 0, st, XB + (size_t)g.pad * Di, Di, norm_w, norm_b, fc2_w, fc2_b, C, logits);
    TM_CHECK_LAUNCH();
    return ACMIL_OK;
}

// Convenience entry: the side stream and its event pair are the LIBRARY's (one per device, created at first use, never while the
// caller is capturing), the enqueue holds a lock -- the one place where this library keeps state (include/acmil_hip.h says so).
// Callers that want the ownership contract of every other entry point use acmil_transmil_forward_ex.
extern "C" int acmil_transmil_forward(const float* x, int N, int D, int Di, int C, const float* fc1_w, const float* fc1_b,
                                      const float* cls_token, const float* const* layer1, const float* const* layer2,
                                      const float* const* ppeg, const float* norm_w, const float* norm_b, const float* fc2_w,
                                      const float* fc2_b, float* logits, float* dbg_h1, float* dbg_hp, float* dbg_h2, void* workspace,
                                      void* stream) {
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> side_lock(tm_side_mutex);      // one forward at a time records / waits on the side stream's events
    TmSide* side = tm_side(st);
    bool capturing = false;
    TmSerial* ser = tm_serial(st, &capturing);
    if (ser && ser->state == 2 && ser->last != st && hipStreamWaitEvent(st, ser->done, 0) != hipSuccess) return ACMIL_ERR_LAUNCH;
    struct SerialMark {      // the end of this forward, recorded on every return path
        TmSerial* s; hipStream_t st;
        ~SerialMark() { if (s && hipEventRecord(s->done, st) == hipSuccess) { s->state = 2; s->last = st; } else if (s) (void)hipGetLastError(); }
    } serial_mark{ser, st};
    return tm_forward_impl(x, N, D, Di, C, fc1_w, fc1_b, cls_token, layer1, layer2, ppeg, norm_w, norm_b, fc2_w, fc2_b, logits, dbg_h1, dbg_hp,
                           dbg_h2, workspace, st, side);
}

// The same forward with CALLER-OWNED concurrency objects (the library allocates nothing and keeps no state):
// side_stream = a second stream of the same device, fork_event / join_event = two events (timing may be disabled); all three NULL =
// every launch on `stream`.  The call records fork_event on `stream` / join_event on side_stream and makes each stream wait for
// the other's event, several times; when it returns, `stream` alone orders the outputs.  The caller must not use the three objects
// for anything else while a call is being ENQUEUED (one forward at a time per triple; the GPU work of consecutive forwards may overlap).
extern "C" int acmil_transmil_forward_ex(const float* x, int N, int D, int Di, int C, const float* fc1_w, const float* fc1_b,
                                         const float* cls_token, const float* const* layer1, const float* const* layer2,
                                         const float* const* ppeg, const float* norm_w, const float* norm_b, const float* fc2_w,
                                         const float* fc2_b, float* logits, float* dbg_h1, float* dbg_hp, float* dbg_h2, void* workspace,
                                         void* stream, void* side_stream, void* fork_event, void* join_event) {
    const int given = (side_stream != nullptr) + (fork_event != nullptr) + (join_event != nullptr);
    if (given != 0 && given != 3) return ACMIL_ERR_NULL;
    if (given == 3 && side_stream == stream) return ACMIL_ERR_SHAPE;
    TmSide side = {(hipStream_t)side_stream, (hipEvent_t)fork_event, (hipEvent_t)join_event, 1};
    return tm_forward_impl(x, N, D, Di, C, fc1_w, fc1_b, cls_token, layer1, layer2, ppeg, norm_w, norm_b, fc2_w, fc2_b, logits, dbg_h1, dbg_hp,
                           dbg_h2, workspace, (hipStream_t)stream, given == 3 ? &side : nullptr);
}
