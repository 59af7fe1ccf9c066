// transmil_train.hip -- op-level forward/backward kernels for TRAINING the TransMIL / Nystrom path (and any other module
// assembled from them).  The eval forward of TransMIL is the fused pipeline of transmil.hip; training needs gradients, so it
// runs op by op as torch.autograd Functions (acmil_amd/autograd.py) whose forward and backward are these entry points plus
// the GEMMs of gemm_f32.hip -- every O(N) pass is a hand-written kernel here, autograd only sequences them.
//
//   acmil_layernorm_fwd / _bwd        nn.LayerNorm(dim, eps)                       transMIL.py:12,27 / :57,85
//   acmil_softmax_rows_bwd            d softmax(S, dim=-1)                          nystrom_attention.py:115 (attn1..3)
//   acmil_seqconv                     depth-wise Conv2d(heads, heads, (33,1), groups=heads) along the sequence   :135-136
//   acmil_seqconv_bwd_w               its weight gradient (the input gradient is the same conv with the flipped kernel)
//   acmil_dwconv7                     depth-wise 7x7 on the token grid, channels-last (PPEG, transMIL.py:33-45, kernels folded)
//   acmil_dwconv7_bwd_w               its weight / bias gradient
//   acmil_landmark_mean / _bwd        landmark means over l consecutive tokens      nystrom_attention.py:95-111
#include <math.h>
#include "ga_common.h"

#define TT_HEADS 8
#define TT_RES 33

// ---------------------------------------------------------------------------------------------------- LayerNorm
// one wave per row; stats[row] = (mean, rstd) kept for the backward
__global__ __launch_bounds__(256) void tt_ln_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ stats,
                                                       long long rows, int dim, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + r * dim;
    float v[16], s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; v[i] = c < dim ? xr[c] : 0.0f; s += v[i]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / dim;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = (lane + 64 * i < dim) ? v[i] - mean : 0.0f; q = fmaf(d, d, q); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / dim + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < dim) y[r * dim + c] = (v[i] - mean) * rstd * gamma[c] + beta[c]; }
    if (lane == 0 && stats) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  One wave per row; every workgroup also accumulates its
// rows' contribution to dgamma / dbeta into part[block][2][dim] (reduced by tt_colsum_reduce_kernel, fixed order).
__global__ __launch_bounds__(256) void tt_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       long long rows, int dim, float* __restrict__ dx, float* __restrict__ part) {
    extern __shared__ float tt_ln_sm[];          // [4 waves][2][dim]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[16], db[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dg[i] = db[i] = 0.0f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        float xh[16], g[16], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = lane + 64 * i;
            const bool ok = c < dim;
            const float xv = ok ? x[r * dim + c] : 0.0f, dyv = ok ? dy[r * dim + c] : 0.0f;
            xh[i] = ok ? (xv - mean) * rstd : 0.0f;
            g[i] = ok ? dyv * gamma[c] : 0.0f;
            s1 += g[i]; s2 = fmaf(g[i], xh[i], s2);
            dg[i] = fmaf(dyv, xh[i], dg[i]); db[i] += dyv;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float m1 = s1 / dim, m2 = s2 / dim;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < dim) dx[r * dim + c] = rstd * (g[i] - m1 - xh[i] * m2); }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < dim) { tt_ln_sm[(wave * 2 + 0) * dim + c] = dg[i]; tt_ln_sm[(wave * 2 + 1) * dim + c] = db[i]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * dim; e += 256) {
        const int which = e / dim, c = e % dim;
        float s = 0.0f;
        for (int w = 0; w < 4; ++w) s += tt_ln_sm[(w * 2 + which) * dim + c];
        part[((size_t)blockIdx.x * 2 + which) * dim + c] = s;
    }
}

// out[e] = sum_b part[b * stride + e] (fixed order), e < width
__global__ __launch_bounds__(256) void tt_colsum_reduce_kernel(const float* __restrict__ part, int nblocks, size_t stride, int width,
                                                              float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= width) return;
    float s = 0.0f;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * stride + e];
    out[e] = s;
}

// ---------------------------------------------------------------------------------------------------- softmax backward
// dS = P * (dP - sum_j dP_j P_j) per row; one workgroup of 1024 threads per row (rows are 64 .. 100 608 long)
__global__ __launch_bounds__(1024) void tt_softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP,
                                                             float* __restrict__ dS, int cols) {
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.0f;
    for (int c = tid; c < cols; c += 1024) s = fmaf(P[base + c], dP[base + c], s);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];
    for (int c = tid; c < cols; c += 1024) dS[base + c] = P[base + c] * (dP[base + c] - tot);
}

// short rows (cols <= 1024): one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void tt_softmax_bwd_short_kernel(const float* __restrict__ P, const float* __restrict__ dP,
                                                                  float* __restrict__ dS, long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const size_t base = (size_t)r * cols;
    float p[16], d[16], s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        p[i] = c < cols ? P[base + c] : 0.0f; d[i] = c < cols ? dP[base + c] : 0.0f;
        s = fmaf(p[i], d[i], s);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; if (c < cols) dS[base + c] = p[i] * (d[i] - s); }
}

// ---------------------------------------------------------------------------------------------------- sequence conv (33 taps)
// out[i][c] = sum_t w[c / d][t] * v[i + t - 16][c]  (zero outside [0, n)); v has leading dimension ldv (a column block of QKV)
#define TT_CONV_ROWS 64
__global__ __launch_bounds__(256) void tt_seqconv_kernel(const float* __restrict__ v, int ldv, float* __restrict__ out, int n, int Di,
                                                        const float* __restrict__ w) {
    __shared__ __attribute__((aligned(16))) float tile[(TT_CONV_ROWS + TT_RES - 1) * 64];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * TT_CONV_ROWS;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int d = Di / TT_HEADS;
#pragma unroll 4
    for (int idx = threadIdx.x; idx < (TT_CONV_ROWS + TT_RES - 1) * 16; idx += 256) {
        const int rr = idx >> 4, c4 = idx & 15;
        const int r = r0 + rr - TT_RES / 2;
        f32x4 val = {0.0f, 0.0f, 0.0f, 0.0f};
        if (r >= 0 && r < n && c0 + 4 * c4 < Di) val = *(const f32x4*)(v + (size_t)r * ldv + c0 + 4 * c4);
        *(f32x4*)(tile + rr * 64 + 4 * c4) = val;
    }
    __syncthreads();
    if (c0 + cl >= Di) return;
    const float* wh = w + (size_t)((c0 + cl) / d) * TT_RES;
    float wr[TT_RES];
#pragma unroll
    for (int t = 0; t < TT_RES; ++t) wr[t] = wh[t];
#pragma unroll 1
    for (int blk = 0; blk < TT_CONV_ROWS / 16; ++blk) {
        const int rr0 = 16 * rg + 4 * blk;
        if (r0 + rr0 >= n) break;
        float in[TT_RES + 3];
#pragma unroll
        for (int i = 0; i < TT_RES + 3; ++i) in[i] = tile[(rr0 + i) * 64 + cl];
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < TT_RES; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wr[t], in[j + t], acc[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + rr0 + j < n) out[(size_t)(r0 + rr0 + j) * Di + c0 + cl] = acc[j];
    }
}

// dw[h][t] = sum_{i, c in head h} dout[i][c] * v[i + t - 16][c].  grid (heads, row chunks): workgroup accumulates the 33 taps of
// one head over its row chunk (lanes stride the d channels, waves stride rows), partials reduced in a fixed order afterwards.
__global__ __launch_bounds__(256) void tt_seqconv_dw_kernel(const float* __restrict__ dout, const float* __restrict__ v, int ldv, int n,
                                                           int Di, int rows_per_block, float* __restrict__ part) {
    __shared__ float red[4][TT_RES];
    const int h = blockIdx.x, d = Di / TT_HEADS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.y * rows_per_block, re = min(n, rb + rows_per_block);
    float acc[TT_RES];
#pragma unroll
    for (int t = 0; t < TT_RES; ++t) acc[t] = 0.0f;
    for (int c = lane; c < d; c += 64) {
        const int col = h * d + c;
        for (int i = rb + wave; i < re; i += 4) {
            const float g = dout[(size_t)i * Di + col];
#pragma unroll
            for (int t = 0; t < TT_RES; ++t) {
                const int r = i + t - TT_RES / 2;
                if (r >= 0 && r < n) acc[t] = fmaf(g, v[(size_t)r * ldv + col], acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TT_RES; ++t) {
        float s = acc[t];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave][t] = s;
    }
    __syncthreads();
    if (threadIdx.x < TT_RES)
        part[((size_t)blockIdx.y * TT_HEADS + h) * TT_RES + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------- depth-wise 7x7 (PPEG)
#define TT_PT 8
__global__ __launch_bounds__(256) void tt_dwconv7_kernel(const float* __restrict__ in, float* __restrict__ out, int side, int C,
                                                        const float* __restrict__ weff, const float* __restrict__ beff) {
    __shared__ __attribute__((aligned(16))) float tile[(TT_PT + 6) * (TT_PT + 6) * 64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int tiles_x = (side + TT_PT - 1) / TT_PT;
    const int ty0 = (blockIdx.y / tiles_x) * TT_PT, tx0 = (blockIdx.y % tiles_x) * TT_PT;
    {
        const int cb = blockIdx.x * 64;
#pragma unroll 4
        for (int idx = threadIdx.x; idx < (TT_PT + 6) * (TT_PT + 6) * 16; idx += 256) {
            const int e = idx >> 4, c4 = idx & 15;
            const int yy = ty0 + e / (TT_PT + 6) - 3, xx = tx0 + e % (TT_PT + 6) - 3;
            f32x4 val = {0.0f, 0.0f, 0.0f, 0.0f};
            if (cb + 4 * c4 < C && yy >= 0 && yy < side && xx >= 0 && xx < side)
                val = *(const f32x4*)(in + ((size_t)yy * side + xx) * C + cb + 4 * c4);
            *(f32x4*)(tile + e * 64 + 4 * c4) = val;
        }
    }
    __syncthreads();
    if (c >= C) return;
    float w[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) w[t] = weff[(size_t)t * C + c];
    const float b = beff ? beff[c] : 0.0f;
#pragma unroll
    for (int r = 0; r < TT_PT / 4; ++r) {
        const int py = (TT_PT / 4) * grp + r, y = ty0 + py;
        if (y >= side) continue;
        float acc[TT_PT];
#pragma unroll
        for (int px = 0; px < TT_PT; ++px) acc[px] = b;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            float iv[TT_PT + 6];
#pragma unroll
            for (int i = 0; i < TT_PT + 6; ++i) iv[i] = tile[((py + ky) * (TT_PT + 6) + i) * 64 + cl];
#pragma unroll
            for (int px = 0; px < TT_PT; ++px)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) acc[px] = fmaf(w[ky * 7 + kx], iv[px + kx], acc[px]);
        }
#pragma unroll
        for (int px = 0; px < TT_PT; ++px)
            if (tx0 + px < side) out[((size_t)y * side + tx0 + px) * C + c] = acc[px];
    }
}

// dweff[tap][c] = sum_pixels dy[p][c] * x[p shifted by tap][c]; dbeff[c] = sum_pixels dy[p][c].  grid (C/64, row chunks): lane =
// channel, waves stride the pixels of the chunk; partials [chunk][50][C] reduced afterwards in a fixed order.
__global__ __launch_bounds__(256) void tt_dwconv7_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, int side, int C,
                                                           int rows_per_block, float* __restrict__ part) {
    __shared__ float red[4][50][64];
    const int cl = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int y0 = blockIdx.y * rows_per_block, y1 = min(side, y0 + rows_per_block);
    float acc[50];
#pragma unroll
    for (int t = 0; t < 50; ++t) acc[t] = 0.0f;
    if (c < C) {
        for (int p = y0 * side + wave; p < y1 * side; p += 4) {
            const int y = p / side, xx = p % side;
            const float g = dy[(size_t)p * C + c];
            acc[49] += g;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const int yy = y + ky - 3;
                if (yy < 0 || yy >= side) continue;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const int x2 = xx + kx - 3;
                    if (x2 >= 0 && x2 < side) acc[ky * 7 + kx] = fmaf(g, x[((size_t)yy * side + x2) * C + c], acc[ky * 7 + kx]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 50; ++t) red[wave][t][cl] = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < 50 * 64; e += 256) {
        const int t = e / 64, l = e % 64;
        if (blockIdx.x * 64 + l < C)
            part[((size_t)blockIdx.y * 50 + t) * C + blockIdx.x * 64 + l] = (red[0][t][l] + red[1][t][l]) + (red[2][t][l] + red[3][t][l]);
    }
}

// ---------------------------------------------------------------------------------------------------- landmark means
// out[h][j][dd] = mean over rows j*l .. j*l+l-1 of src[row][h*d + dd]   (src leading dimension lds: a q or k column block of QKV)
__global__ __launch_bounds__(1024) void tt_landmark_kernel(const float* __restrict__ src, int lds_, int l, int m, int Di, int phases,
                                                          float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tt_lm_red[];
    const int j = blockIdx.x, c4n = Di / 4;
    const int c4 = threadIdx.x % c4n, ph = threadIdx.x / c4n;
    const int d = Di / TT_HEADS;
    if (ph < phases) {
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int t = ph; t < l; t += phases) s += *(const f32x4*)(src + ((size_t)j * l + t) * lds_ + 4 * c4);
        *(f32x4*)(tt_lm_red + (size_t)ph * Di + 4 * c4) = s;
    }
    __syncthreads();
    const float inv = 1.0f / (float)l;
    for (int c = threadIdx.x; c < Di; c += blockDim.x) {
        float s = 0.0f;
        for (int p2 = 0; p2 < phases; ++p2) s += tt_lm_red[(size_t)p2 * Di + c];
        out[((size_t)(c / d) * m + j) * d + c % d] = s * inv;
    }
}

// dsrc[row][c] = dout[c / d][row / l][c % d] / l    (dense [n, Di] output)
__global__ __launch_bounds__(256) void tt_landmark_bwd_kernel(const float* __restrict__ dout, int l, int m, int Di, long long total,
                                                             float* __restrict__ dsrc) {
    const int d = Di / TT_HEADS;
    const float inv = 1.0f / (float)l;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / Di; const int c = (int)(e % Di);
        dsrc[e] = dout[((size_t)(c / d) * m + row / l) * d + c % d] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------- elementwise / reductions
// dx = dy where y > 0 else 0   (ReLU backward on the saved output)
__global__ __launch_bounds__(256) void tt_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                         long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) dx[e] = y[e] > 0.0f ? dy[e] : 0.0f;
}

// part[b][c] = sum over the rows of chunk b of x[row][c]   (bias gradients); lanes stride columns, waves stride rows
__global__ __launch_bounds__(256) void tt_colsum_kernel(const float* __restrict__ x, long long rows, int cols, int rows_per_block,
                                                       float* __restrict__ part) {
    extern __shared__ float tt_cs_sm[];      // [4][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long rb = (long long)blockIdx.x * rows_per_block, re = rb + rows_per_block < rows ? rb + rows_per_block : rows;
    for (int c0 = 0; c0 < cols; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.0f;
        if (c < cols) for (long long r = rb + wave; r < re; r += 4) s += x[r * cols + c];
        if (c < cols) tt_cs_sm[wave * cols + c] = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256)
        part[(size_t)blockIdx.x * cols + c] = (tt_cs_sm[c] + tt_cs_sm[cols + c]) + (tt_cs_sm[2 * cols + c] + tt_cs_sm[3 * cols + c]);
}

// y[n][u] = tanh(G[n][u]) * sigmoid(G[n][Da + u])   (the gate of Attention_Gated / Attn_Net_Gated); G is [N, 2 Da]
__global__ __launch_bounds__(256) void tt_gate_fwd_kernel(const float* __restrict__ G, float* __restrict__ y, long long N, int Da) {
    const long long total = N * Da;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long n = e / Da; const int u = (int)(e % Da);
        y[e] = tanhf(G[n * 2 * Da + u]) * (1.0f / (1.0f + expf(-G[n * 2 * Da + Da + u])));
    }
}

// dG[n][u] = dy * sig(b) * (1 - tanh(a)^2) ; dG[n][Da + u] = dy * tanh(a) * sig(b) * (1 - sig(b))
__global__ __launch_bounds__(256) void tt_gate_bwd_kernel(const float* __restrict__ G, const float* __restrict__ dy, float* __restrict__ dG,
                                                         long long N, int Da) {
    const long long total = N * Da;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long n = e / Da; const int u = (int)(e % Da);
        const float t = tanhf(G[n * 2 * Da + u]), sg = 1.0f / (1.0f + expf(-G[n * 2 * Da + Da + u])), g = dy[e];
        dG[n * 2 * Da + u] = g * sg * (1.0f - t * t);
        dG[n * 2 * Da + Da + u] = g * t * sg * (1.0f - sg);
    }
}

// ==================================================================================================== C ABI
static size_t tt_al(size_t b) { return (b + 255) & ~(size_t)255; }
#define TT_LAUNCH_OK() (hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH)

extern "C" int acmil_layernorm_fwd(const float* x, long long rows, int dim, const float* gamma, const float* beta, float eps,
                                   float* y, float* stats, void* stream) {
    if (rows <= 0 || dim <= 0) return ACMIL_ERR_SHAPE;
    if (dim > 1024) return ACMIL_ERR_UNSUPPORTED;
    if (!x || !gamma || !beta || !y) return ACMIL_ERR_NULL;
    hipLaunchKernelGGL(tt_ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, stats, rows, dim, gamma, beta, eps);
    return TT_LAUNCH_OK();
}

#define TT_LN_BLOCKS 512
extern "C" size_t acmil_layernorm_bwd_workspace_bytes(long long rows, int dim) {
    if (rows <= 0 || dim <= 0) return 0;
    return tt_al((size_t)TT_LN_BLOCKS * 2 * dim * sizeof(float));
}

extern "C" int acmil_layernorm_bwd(const float* x, const float* dy, const float* stats, const float* gamma, long long rows, int dim,
                                   float* dx, float* dgamma, float* dbeta, void* workspace, void* stream) {
    if (rows <= 0 || dim <= 0) return ACMIL_ERR_SHAPE;
    if (dim > 1024) return ACMIL_ERR_UNSUPPORTED;
    if (!x || !dy || !stats || !gamma || !dx || !dgamma || !dbeta || !workspace) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    int blocks = (int)((rows + 3) / 4 < TT_LN_BLOCKS ? (rows + 3) / 4 : TT_LN_BLOCKS);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(tt_ln_bwd_kernel, dim3(blocks), dim3(256), (size_t)8 * dim * sizeof(float), st, x, dy, stats, gamma, rows, dim, dx, part);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(tt_colsum_reduce_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, part, blocks, (size_t)2 * dim, dim, dgamma);
    hipLaunchKernelGGL(tt_colsum_reduce_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, part + dim, blocks, (size_t)2 * dim, dim, dbeta);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_softmax_rows_bwd(const float* P, const float* dP, float* dS, long long rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return ACMIL_ERR_SHAPE;
    if (!P || !dP || !dS) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (cols <= 1024) hipLaunchKernelGGL(tt_softmax_bwd_short_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, dS, rows, cols);
    else hipLaunchKernelGGL(tt_softmax_bwd_kernel, dim3((unsigned)rows), dim3(1024), 0, st, P, dP, dS, cols);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_seqconv(const float* v, int ldv, int n, int Di, const float* w, float* out, void* stream) {
    if (n <= 0 || Di <= 0 || ldv < Di) return ACMIL_ERR_SHAPE;
    if (Di % (4 * TT_HEADS) != 0 || ldv % 4 != 0) return ACMIL_ERR_UNSUPPORTED;
    if (!v || !w || !out) return ACMIL_ERR_NULL;
    hipLaunchKernelGGL(tt_seqconv_kernel, dim3((Di + 63) / 64, (n + TT_CONV_ROWS - 1) / TT_CONV_ROWS), dim3(256), 0, (hipStream_t)stream, v, ldv, out, n, Di, w);
    return TT_LAUNCH_OK();
}

#define TT_DW_CHUNKS 256
extern "C" size_t acmil_seqconv_bwd_w_workspace_bytes(int n, int Di) {
    (void)n; (void)Di;
    return tt_al((size_t)TT_DW_CHUNKS * TT_HEADS * TT_RES * sizeof(float));
}

extern "C" int acmil_seqconv_bwd_w(const float* dout, const float* v, int ldv, int n, int Di, float* dw, void* workspace, void* stream) {
    if (n <= 0 || Di <= 0 || ldv < Di) return ACMIL_ERR_SHAPE;
    if (Di % TT_HEADS != 0) return ACMIL_ERR_UNSUPPORTED;
    if (!dout || !v || !dw || !workspace) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    int chunks = (n + 255) / 256; if (chunks > TT_DW_CHUNKS) chunks = TT_DW_CHUNKS;
    const int rpb = (n + chunks - 1) / chunks;
    chunks = (n + rpb - 1) / rpb;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(tt_seqconv_dw_kernel, dim3(TT_HEADS, chunks), dim3(256), 0, st, dout, v, ldv, n, Di, rpb, part);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(tt_colsum_reduce_kernel, dim3((TT_HEADS * TT_RES + 255) / 256), dim3(256), 0, st, part, chunks, (size_t)TT_HEADS * TT_RES, TT_HEADS * TT_RES, dw);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_dwconv7(const float* x, int side, int C, const float* weff, const float* beff, float* y, void* stream) {
    if (side <= 0 || C <= 0) return ACMIL_ERR_SHAPE;
    if (C % 4 != 0) return ACMIL_ERR_UNSUPPORTED;
    if (!x || !weff || !y) return ACMIL_ERR_NULL;
    const int tiles = (side + TT_PT - 1) / TT_PT;
    hipLaunchKernelGGL(tt_dwconv7_kernel, dim3((C + 63) / 64, tiles * tiles), dim3(256), 0, (hipStream_t)stream, x, y, side, C, weff, beff);
    return TT_LAUNCH_OK();
}

#define TT_D7_CHUNKS 128
extern "C" size_t acmil_dwconv7_bwd_w_workspace_bytes(int side, int C) {
    if (side <= 0 || C <= 0) return 0;
    return tt_al((size_t)TT_D7_CHUNKS * 50 * C * sizeof(float));
}

extern "C" int acmil_dwconv7_bwd_w(const float* dy, const float* x, int side, int C, float* dweff, float* dbeff, void* workspace, void* stream) {
    if (side <= 0 || C <= 0) return ACMIL_ERR_SHAPE;
    if (!dy || !x || !dweff || !dbeff || !workspace) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    int chunks = side < TT_D7_CHUNKS ? side : TT_D7_CHUNKS;
    const int rpb = (side + chunks - 1) / chunks;
    chunks = (side + rpb - 1) / rpb;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(tt_dwconv7_dw_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, st, dy, x, side, C, rpb, part);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(tt_colsum_reduce_kernel, dim3((49 * C + 255) / 256), dim3(256), 0, st, part, chunks, (size_t)50 * C, 49 * C, dweff);
    hipLaunchKernelGGL(tt_colsum_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, st, part + (size_t)49 * C, chunks, (size_t)50 * C, C, dbeff);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_landmark_mean(const float* src, int ld, int n, int l, int Di, float* out, void* stream) {
    if (n <= 0 || l <= 0 || Di <= 0 || n % l != 0 || ld < Di) return ACMIL_ERR_SHAPE;
    if (Di % (4 * TT_HEADS) != 0 || ld % 4 != 0 || Di > 4096) return ACMIL_ERR_UNSUPPORTED;
    if (!src || !out) return ACMIL_ERR_NULL;
    const int m = n / l;
    int phases = 1024 / (Di / 4); if (phases > 8) phases = 8; if (phases > l) phases = l; if (phases < 1) phases = 1;
    const int threads = ((Di / 4) * phases + 63) / 64 * 64;
    hipLaunchKernelGGL(tt_landmark_kernel, dim3(m), dim3(threads), (size_t)phases * Di * sizeof(float), (hipStream_t)stream, src, ld, l, m, Di, phases, out);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_landmark_mean_bwd(const float* dout, int n, int l, int Di, float* dsrc, void* stream) {
    if (n <= 0 || l <= 0 || Di <= 0 || n % l != 0) return ACMIL_ERR_SHAPE;
    if (Di % TT_HEADS != 0) return ACMIL_ERR_UNSUPPORTED;
    if (!dout || !dsrc) return ACMIL_ERR_NULL;
    const long long total = (long long)n * Di;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(tt_landmark_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, l, n / l, Di, total, dsrc);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_relu_bwd(const float* dy, const float* y, float* dx, long long total, void* stream) {
    if (total <= 0) return ACMIL_ERR_SHAPE;
    if (!dy || !y || !dx) return ACMIL_ERR_NULL;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(tt_relu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, y, dx, total);
    return TT_LAUNCH_OK();
}

#define TT_CS_CHUNKS 256
extern "C" size_t acmil_colsum_workspace_bytes(long long rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return tt_al((size_t)TT_CS_CHUNKS * cols * sizeof(float));
}

extern "C" int acmil_colsum(const float* x, long long rows, int cols, float* out, void* workspace, void* stream) {
    if (rows <= 0 || cols <= 0) return ACMIL_ERR_SHAPE;
    if (cols > 8192) return ACMIL_ERR_UNSUPPORTED;
    if (!x || !out || !workspace) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    int chunks = (int)((rows + 63) / 64 < TT_CS_CHUNKS ? (rows + 63) / 64 : TT_CS_CHUNKS);
    const int rpb = (int)((rows + chunks - 1) / chunks);
    chunks = (int)((rows + rpb - 1) / rpb);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(tt_colsum_kernel, dim3(chunks), dim3(256), (size_t)4 * cols * sizeof(float), st, x, rows, cols, rpb, part);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(tt_colsum_reduce_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, part, chunks, (size_t)cols, cols, out);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_gate_fwd(const float* G, float* y, long long N, int Da, void* stream) {
    if (N <= 0 || Da <= 0) return ACMIL_ERR_SHAPE;
    if (!G || !y) return ACMIL_ERR_NULL;
    const long long total = N * Da;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(tt_gate_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G, y, N, Da);
    return TT_LAUNCH_OK();
}

extern "C" int acmil_gate_bwd(const float* G, const float* dy, float* dG, long long N, int Da, void* stream) {
    if (N <= 0 || Da <= 0) return ACMIL_ERR_SHAPE;
    if (!G || !dy || !dG) return ACMIL_ERR_NULL;
    const long long total = N * Da;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(tt_gate_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G, dy, dG, N, Da);
    return TT_LAUNCH_OK();
}
