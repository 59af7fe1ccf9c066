// transmil_attn.hip -- fused Nystrom-attention kernels of the TransMIL path for gfx950 (exact-fp32 MFMA,
// v_mfma_f32_32x32x2_f32).  They replace, per layer and without ever materialising the [heads, n', m] / [heads, m, n']
// attention matrices (618 MB each at cfg4, written once and read 2-3 times by the unfused GEMM + softmax + GEMM chain):
//
//   tm_attn1_kernel : OUT[n, h*d + e] = sum_l softmax_l( scale q[n] . k_l[l] ) * W2[h][l][e]
//                     = the `attn1 @ (attn2_inv @ (attn3 @ v))` leg      (architecture/nystrom_attention.py:113-133)
//   tm_attn3_kernel : AV[h][l][e]     = sum_n softmax_n( scale q_l[l] . k[n] ) * v[n][e]
//                     = `attn3 @ v` with the softmax over all n' keys    (nystrom_attention.py:115-121,131)
//                     flash-style: keys are split into chunks, each workgroup keeps a running (max, sum, acc) per
//                     landmark; tm_attn3_merge_kernel combines the chunk partials in a fixed order (deterministic).
//
// Both use the register-chaining trick of ga_forward_kernel.h: the first GEMM is computed with the softmax axis on the
// accumulator REGISTERS (32x32 C/D layout: row = (r&3) + 8(r>>2) + 4(lane>>5), col = lane&31), so the soft-maxed
// accumulators are directly the B operand of the second GEMM (K slot pair (r, lane>>5) <-> rows l, l+4); the matching
// K-slot order of the A operand (W2^T resp. v^T) is produced by the LDS addressing.  K = d is consumed in the lane-half
// split order d_idx = (d/2) * (lane>>5) + s so that every lane's operand values are contiguous in memory.
// Geometry is fixed by TransLayer (transMIL.py:13-23): heads = 8, m = Di/2 landmarks, d = Di/8  =>  d = m/4; with
// MT = m/32 the kernels are instantiated for MT in {2,4,6,8} (Di = 128, 256, 384, 512); other widths use the GEMM chain.
#include <stdlib.h>
#include "ga_common.h"

#define TMA_HEADS 8
#define TMA_MAX_CHUNKS 128      // key chunks of the attn3 leg (workgroups per head)
// wave priority inside the matrix phases of the split-f16 legs (0 in the softmax / staging phases): the co-resident wave's VALU
// no longer takes issue slots from a running MFMA chain (attn1 leg 157 -> 149 us; 2 measured the same as 1)
#ifndef TMA_PRIO
#define TMA_PRIO 1
// timing-only ablations of tm_attn1x_kernel (tools/build_file_variant.sh; WRONG results; 0 in every product build): 1 no exp / max / sum,
// 2 no GEMM-S MFMAs, 4 no GEMM-PV MFMAs, 8 no convolution (loads + MFMAs), 16 no OUT stores
#ifndef TMA1_ABL
#define TMA1_ABL 0
#endif
#endif
#ifndef TMA_PRIO3
#define TMA_PRIO3 0
#endif

__device__ __forceinline__ float tma_xor32(float v) { return __shfl_xor(v, 32); }

// ---------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tm_attn1_kernel(const float* __restrict__ QKV, const float* __restrict__ KL,
                                                       const float* __restrict__ W2, float* __restrict__ OUT, int npad, int Di,
                                                       float scale) {
    constexpr int M = 32 * MT, D = 8 * MT, DH = D / 2, ET = (D + 31) / 32, EP = 32 * ET, LDM = M + 4, LDE = EP + 8;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* KLs = sm;                  // [D][LDM]   scale * k_l, transposed: A operand of GEMM-S (lane = landmark)
    float* W2s = sm + D * LDM;        // [M][LDE]   W2 rows, zero padded to EP columns: A operand of GEMM-PV (lane = e)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y;
    for (int idx = tid; idx < M * D; idx += 256) {
        const int l = idx / D, c = idx % D;
        KLs[c * LDM + l] = KL[((size_t)h * M + l) * D + c] * scale;
        W2s[l * LDE + c] = W2[((size_t)h * M + l) * D + c];
    }
    if (EP > D)
        for (int idx = tid; idx < M * (EP - D); idx += 256) W2s[(idx / (EP - D)) * LDE + D + idx % (EP - D)] = 0.0f;
    __syncthreads();

    const int nblk = npad / 32, stride = gridDim.x * 4;
    int rb = blockIdx.x * 4 + wave;
    f32x4 qn[DH / 4];
    auto load_q = [&](int b) {
        const float* qp = QKV + (size_t)(b * 32 + i31) * 3 * Di + h * D + DH * hi;
#pragma unroll
        for (int j = 0; j < DH / 4; ++j) qn[j] = *(const f32x4*)(qp + 4 * j);
    };
    if (rb < nblk) load_q(rb);
    for (; rb < nblk; rb += stride) {
        float q[DH];
#pragma unroll
        for (int j = 0; j < DH / 4; ++j) { q[4 * j] = qn[j][0]; q[4 * j + 1] = qn[j][1]; q[4 * j + 2] = qn[j][2]; q[4 * j + 3] = qn[j][3]; }
        if (rb + stride < nblk) load_q(rb + stride);       // prefetch the next block of queries behind the MFMA work

        // GEMM-S: acc[t][r] = scale * q[n = i31] . k_l[l = 32t + row(r, hi)]
        f32x16 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const float* ap = KLs + (DH * hi) * LDM + i31;
#pragma unroll
        for (int s = 0; s < DH; ++s) {
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s * LDM + 32 * t], q[s], acc[t], 0, 0, 0);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep hipcc from hoisting all LDS reads of the unrolled loop (spills)
        }

        // softmax over the m landmarks of row n: registers of this lane + the partner half (lane ^ 32)
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[t][r]);
        mx = fmaxf(mx, tma_xor32(mx));
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float p = __expf(acc[t][r] - mx); acc[t][r] = p; sum += p; }
        sum += tma_xor32(sum);

        // GEMM-PV: o[et][r] = sum_l W2[l][e = 32et + row(r, hi)] * p[l][n]; K slot pair of (t, r): l = 32t + (r&3) + 8(r>>2) + 4hi
        f32x16 o[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[et][r] = 0.0f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* wp = W2s + (32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi) * LDE + i31;
#pragma unroll
                for (int et = 0; et < ET; ++et) o[et] = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[32 * et], acc[t][r], o[et], 0, 0, 0);
                if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        const float inv = 1.0f / sum;
        float* op = OUT + (size_t)(rb * 32 + i31) * Di + h * D;
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int e = 32 * et + 8 * gq + 4 * hi;
                if (e < D) *(f32x4*)(op + e) = f32x4{o[et][4 * gq] * inv, o[et][4 * gq + 1] * inv, o[et][4 * gq + 2] * inv, o[et][4 * gq + 3] * inv};
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One workgroup = (head, key chunk), MT waves = the MT landmark tiles of the head; 32 keys per step, k / v tiles staged
// through a double-buffered LDS pair (8 D float4 per tile = exactly one float4 of k and one of v per thread).
template <int MT>
__global__ __launch_bounds__(64 * MT) void tm_attn3_kernel(const float* __restrict__ QKV, const float* __restrict__ QL,
                                                          float* __restrict__ part_ms, float* __restrict__ part_o, int npad, int Di,
                                                          float scale, int blocks_per_chunk) {
    constexpr int M = 32 * MT, D = 8 * MT, DH = D / 2, ET = (D + 31) / 32, EP = 32 * ET, LDK = 36, LDE = EP + 8;
    __shared__ __attribute__((aligned(16))) float ks[2][D * LDK];    // [d][key]  A operand of GEMM-S (lane = key)
    __shared__ __attribute__((aligned(16))) float vs[2][32 * LDE];   // [key][e]  A operand of GEMM-PV (lane = e)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int nblk = npad / 32;
    const int kb0 = chunk * blocks_per_chunk, kb1 = min(nblk, kb0 + blocks_per_chunk);

    float ql[DH];                     // B operand of GEMM-S: scale * q_l[l = 32 wave + i31][DH hi + s]
    {
        const float* qp = QL + ((size_t)h * M + 32 * wave + i31) * D + DH * hi;
#pragma unroll
        for (int s = 0; s < DH; ++s) ql[s] = qp[s] * scale;
    }
    if (EP > D) {                     // zero the padded e columns of both v buffers once
        for (int idx = tid; idx < 2 * 32 * (EP - D); idx += 64 * MT) {
            const int b = idx / (32 * (EP - D)), rem = idx % (32 * (EP - D));
            vs[b][(rem / (EP - D)) * LDE + D + rem % (EP - D)] = 0.0f;
        }
    }
    // staging map: thread -> key row = tid / (D/4), float4 column c4 = tid % (D/4)   (64 MT == 8 D threads)
    const int srow = tid / (D / 4), sc4 = tid % (D / 4);
    f32x4 kreg, vreg;
    auto load_kv = [&](int kb) {
        const float* base = QKV + (size_t)(kb * 32 + srow) * 3 * Di + h * D + 4 * sc4;
        kreg = *(const f32x4*)(base + Di);
        vreg = *(const f32x4*)(base + 2 * Di);
    };
    float m_run = -INFINITY, s_run = 0.0f;
    f32x16 o[ET];
#pragma unroll
    for (int et = 0; et < ET; ++et)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[et][r] = 0.0f;

    if (kb0 < kb1) load_kv(kb0);
    for (int kb = kb0; kb < kb1; ++kb) {
        const int buf = (kb - kb0) & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) ks[buf][(4 * sc4 + j) * LDK + srow] = kreg[j];
        *(f32x4*)(&vs[buf][srow * LDE + 4 * sc4]) = vreg;
        __syncthreads();              // one barrier per step: the buffer written two steps later was last read before this barrier
        if (kb + 1 < kb1) load_kv(kb + 1);

        // GEMM-S: S[r] = scale * k[n = row(r, hi)] . q_l[l = i31]
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
        const float* ap = ks[buf] + (DH * hi) * LDK + i31;
#pragma unroll
        for (int s = 0; s < DH; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s * LDK], ql[s], S, 0, 0, 0);

        // online softmax over keys for landmark l (this lane + partner half hold the 32 keys of the step)
        float bm = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) bm = fmaxf(bm, S[r]);
        bm = fmaxf(bm, tma_xor32(bm));
        const float mn = fmaxf(m_run, bm);
        const float alpha = __expf(m_run - mn);      // exp(-inf) = 0 on the first step
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = __expf(S[r] - mn); ps += S[r]; }
        ps += tma_xor32(ps);
        s_run = s_run * alpha + ps;
        m_run = mn;
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[et][r] *= alpha;

        // GEMM-PV: o[et][r] += sum_n v[n][e = 32et + row(r, hi)] * p[n][l]; K slot pair of r: n = (r&3) + 8(r>>2) + 4hi
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vp = vs[buf] + ((r & 3) + 8 * (r >> 2) + 4 * hi) * LDE + i31;
#pragma unroll
            for (int et = 0; et < ET; ++et) o[et] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * et], S[r], o[et], 0, 0, 0);
        }
    }
    // partials: part_ms[h][chunk][l][2], part_o[h][chunk][l][D]
    const size_t row = ((size_t)h * nchunks + chunk) * M + 32 * wave + i31;
    if (hi == 0) { part_ms[row * 2] = m_run; part_ms[row * 2 + 1] = s_run; }
#pragma unroll
    for (int et = 0; et < ET; ++et)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int e = 32 * et + 8 * gq + 4 * hi;
            if (e < D) *(f32x4*)(part_o + row * D + e) = f32x4{o[et][4 * gq], o[et][4 * gq + 1], o[et][4 * gq + 2], o[et][4 * gq + 3]};
        }
}

// AV[h][l][e] = sum_c o_c[e] exp(m_c - M) / sum_c s_c exp(m_c - M), chunks in index order (deterministic).
// One wave per landmark row: lane c first loads the chunk statistics (<= 64 chunks), the wave agrees on M and the scale
// factors, then lane e accumulates its feature over the chunks with the factors read through LDS.
__global__ __launch_bounds__(256) void tm_attn3_merge_kernel(const float* __restrict__ part_ms, const float* __restrict__ part_o,
                                                            float* __restrict__ AV, int M, int D, int nchunks) {
    __shared__ float fac[4][TMA_MAX_CHUNKS];
    const int h = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + wave;
    if (l >= M) return;
    // lane c holds chunks c and c + 64 (up to TMA_MAX_CHUNKS = 128 chunks)
    float mc[2] = {-INFINITY, -INFINITY}, sc[2] = {0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (lane + 64 * u < nchunks) {
            const float* p = part_ms + (((size_t)h * nchunks + lane + 64 * u) * M + l) * 2;
            mc[u] = p[0]; sc[u] = p[1];
        }
    float mx = fmaxf(mc[0], mc[1]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float f = (lane + 64 * u < nchunks) ? __expf(mc[u] - mx) : 0.0f;
        fac[wave][lane + 64 * u] = f;
        den += sc[u] * f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) den += __shfl_xor(den, o);       // fixed shuffle tree: deterministic
    const float inv = 1.0f / den;
    for (int e = lane; e < D; e += 64) {
        float num = 0.0f;
        for (int c = 0; c < nchunks; ++c) num += part_o[(((size_t)h * nchunks + c) * M + l) * D + e] * fac[wave][c];
        AV[((size_t)h * M + l) * D + e] = num * inv;
    }
}

// ===============================================================================================================
// Split-f16 ("f16x3") variants of the two legs: every fp32 operand is split hi + lo in f16 and each product is formed as
// hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16 (fp32 accumulate, ~1e-6 relative) -- 5.3x fewer matrix-pipe cycles than
// the 32x32x2 fp32 form.  Same register chaining: the 16 accumulator registers of a 32x32 tile, taken 8 at a time, are the
// K slots of the next MFMA; K slot (hi, j) of register group g is row 16 g + 4 hi + (j & 3) + 8 (j >> 2), and the LDS
// images of W2^T / v^T are written pre-permuted in that order so a lane reads its 8 slots as one ds_read_b128.
// W2 = attn2^+ (attn3 v) is rescaled by a power of two per head (|W2| <= 1024) before the split: a pseudo-inverse can be
// large, and f16 halves overflow at 65504.
typedef _Float16 tma_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 tma_h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tma_split8(const float (&v)[8], tma_h8& h, tma_h8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const _Float16 x = (_Float16)v[j]; h[j] = x; l[j] = (_Float16)(v[j] - (float)x); }
}
#define TMA_MFMA3(acc, ah, al, bh, bl)                                         \
    do {                                                                      \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);   \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);   \
    } while (0)
// K slot -> position inside the permuted 32-row group: row lr = 16 e2 + 4 hi + (j & 3) + 8 (j >> 2)
__device__ __forceinline__ int tma_perm_index(int lr, int EP, int e) {   // f16 index of (row lr, column e) inside one 32-row group image
    const int e2 = lr >> 4, hi_n = (lr >> 2) & 1, j = (lr & 3) + 4 * ((lr >> 3) & 1);
    return (((e2 * EP + e) * 2 + hi_n) * 8) + j;
}

// convw != null (round 4): the depth-wise residual convolution of v along the sequence (nystrom_attention.py:135-142; 33 taps per
// head, zero padding) is added HERE, on the matrix pipe, instead of by a pass of its own over v and OUT (tm_seqconv_kernel: 540 MB of
// HBM traffic and 125 us per layer at cfg4).  For the 32 output rows n of a block and the 64 input rows n' = 32 rb - 16 .. 32 rb + 47,
//   conv[e][n] = sum_n' v[n'][e] T[n'][n],   T[n'][n] = w[h][n' - n + 16]  (0 outside 0..32):
// a banded Toeplitz factor.  A = v^T comes straight from global memory in the MFMA's A layout (lane = feature e, 8 consecutive rows
// per lane: every load instruction reads 128 contiguous bytes of one row of v), split hi / lo in registers; B = T depends only on
// (lane, K step): a 8 KB table in LDS built once per workgroup.  4 K steps x ET tiles x 3 split products = 24 MFMAs per block next to
// the 126 of the attention leg; the accumulators are the leg's own o[et] after their final scaling.
template <int MT>
__global__ __launch_bounds__(512) void tm_attn1x_kernel(const float* __restrict__ QKV, const float* __restrict__ KL,
                                                        const float* __restrict__ W2, float* __restrict__ OUT, int npad, int Di,
                                                        float scale, const float* __restrict__ convw) {
    constexpr int M = 32 * MT, D = 8 * MT, KS = D / 16, ET = (D + 31) / 32, EP = 32 * ET, LD = D + 8;
    constexpr int KLN = M * LD, W2N = MT * 2 * EP * 16;      // f16 elements per plane
    extern __shared__ __attribute__((aligned(16))) char smx[];
    _Float16* KLh = (_Float16*)smx; _Float16* KLl = KLh + KLN;
    _Float16* W2h = KLl + KLN;      _Float16* W2l = W2h + W2N;
    tma_h8* Tch = (tma_h8*)(W2l + W2N);      // [4 K steps][64 lanes] hi halves of the Toeplitz B operand, then the lo halves
    tma_h8* Tcl = Tch + 4 * 64;
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y;
    // power-of-two scale of W2 (per head)
    float mx = 0.0f;
    for (int idx = tid; idx < M * D; idx += 512) mx = fmaxf(mx, fabsf(W2[(size_t)h * M * D + idx]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) mx = fmaxf(mx, red[w]);
    const float s2 = (mx > 0.0f && mx < INFINITY) ? exp2f(10.0f - ceilf(log2f(mx))) : 1.0f;
    const float inv_s2 = 1.0f / s2;
    for (int idx = tid; idx < M * D; idx += 512) {
        const int l = idx / D, c = idx % D;
        const float v = KL[((size_t)h * M + l) * D + c] * scale;
        const _Float16 x = (_Float16)v;
        KLh[l * LD + c] = x; KLl[l * LD + c] = (_Float16)(v - (float)x);
    }
    for (int idx = tid; idx < M * EP; idx += 512) {
        const int l = idx / EP, e = idx % EP;
        const float v = e < D ? W2[((size_t)h * M + l) * D + e] * s2 : 0.0f;
        const _Float16 x = (_Float16)v;
        const int p = (l >> 5) * 2 * EP * 16 + tma_perm_index(l & 31, EP, e);
        W2h[p] = x; W2l[p] = (_Float16)(v - (float)x);
    }
    if (convw && tid < 4 * 64) {
        // B operand of the convolution: K slot (hi, j) of step kp <-> input row 16 kp + 8 hi + j of the 64-row window, column = output row
        // n = lane & 31: tap = (16 kp + 8 hi + j) - n
        const int kp = tid >> 6, ln = tid & 63, n = ln & 31, hh = ln >> 5;
        tma_h8 th, tl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 16 * kp + 8 * hh + j - n;
            const float w = (tap >= 0 && tap <= 32) ? convw[h * 33 + tap] : 0.0f;
            const _Float16 x = (_Float16)w;
            th[j] = x; tl[j] = (_Float16)(w - (float)x);
        }
        Tch[kp * 64 + ln] = th; Tcl[kp * 64 + ln] = tl;
    }
    __syncthreads();

    const int nblk = npad / 32, stride = gridDim.x * 8;
    int rb = blockIdx.x * 8 + wave;
    // v column block of this head from the first guard row on, as a buffer resource (see the convolution below)
    const unsigned ldv4 = 3u * (unsigned)Di * 4u;
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(QKV + 2 * Di + h * D - (size_t)16 * 3 * Di), 0, (int)(((size_t)npad + 32) * 3 * Di * 4 - (size_t)(2 * Di + h * D) * 4), 0x00020000);
    int lane_off[ET];
#pragma unroll
    for (int et = 0; et < ET; ++et) { const int e = 32 * et + i31; lane_off[et] = (8 * hi * 3 * Di + (e < D ? e : D - 1)) * 4; }
    f32x4 qn[KS][2];
    auto load_q = [&](int b) {
        const float* qp = QKV + (size_t)(b * 32 + i31) * 3 * Di + h * D + 8 * hi;
#pragma unroll
        for (int st = 0; st < KS; ++st) { qn[st][0] = *(const f32x4*)(qp + 16 * st); qn[st][1] = *(const f32x4*)(qp + 16 * st + 4); }
    };
    if (rb < nblk) load_q(rb);
    for (; rb < nblk; rb += stride) {
        tma_h8 qh[KS], ql[KS];
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const float v[8] = {qn[st][0][0], qn[st][0][1], qn[st][0][2], qn[st][0][3], qn[st][1][0], qn[st][1][1], qn[st][1][2], qn[st][1][3]};
            tma_split8(v, qh[st], ql[st]);
        }

        f32x16 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        // LDS fragment addresses = ONE per-lane byte offset, made opaque per block, + compile-time constants that fit the ds_read
        // offset field.  Left to itself hipcc hoisted ~27 separate address registers out of the block loop and parked them in
        // scratch: every reload carries an s_waitcnt vmcnt(0) that drains the q / v prefetches in flight.
        int lbk = (i31 * LD + 8 * hi) * 2, lbw = (i31 * 2 + hi) * 16;
        asm volatile("" : "+v"(lbk), "+v"(lbw));
        const char* klh_b = (const char*)KLh + lbk; const char* kll_b = (const char*)KLl + lbk;
        const char* w2h_b = (const char*)W2h + lbw; const char* w2l_b = (const char*)W2l + lbw;
#if TMA_PRIO
        __builtin_amdgcn_s_setprio(TMA_PRIO);
#endif
#pragma unroll
        for (int st = 0; st < KS; ++st) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                constexpr int dummy = 0; (void)dummy;
                const int offb = (32 * t * LD + 16 * st) * 2;
                const tma_h8 ah = *(const tma_h8*)(klh_b + offb), al = *(const tma_h8*)(kll_b + offb);
#if TMA1_ABL & 2
                acc[t][st] += (float)ah[0] + (float)al[1] + (float)qh[st][0] + (float)ql[st][1];
#else
                TMA_MFMA3(acc[t], ah, al, qh[st], ql[st]);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next block's queries are fetched HERE, into the registers the split q halves have just left (issued at the top of the
        // loop they sat on top of the score accumulators' peak and hipcc parked them in scratch: a store waiting on an HBM load)
        if (rb + stride < nblk) load_q(rb + stride);
        __builtin_amdgcn_sched_barrier(0);
#if TMA_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#if TMA1_ABL & 1
        float sum = 1.0f;
#else
        float m2 = -INFINITY;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) m2 = fmaxf(m2, acc[t][r]);
        m2 = fmaxf(m2, tma_xor32(m2));
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float p = __expf(acc[t][r] - m2); acc[t][r] = p; sum += p; }
        sum += tma_xor32(sum);
#endif

        f32x16 o[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[et][r] = 0.0f;
        // the 64-row window of v for the convolution (4 K steps x ET x 8 rows per lane) is fetched INSIDE the product below, two K
        // steps at a time at the points where a third of the softmax accumulators has just died: the loads fly under ~50 MFMAs
        // and the register peak stays where it was (issued after the leg they cost 87 us per layer: 210 vs 123 us per launch)
#if TMA_PRIO
        __builtin_amdgcn_s_setprio(TMA_PRIO);
#endif
        float vw[4][ET][8];
        // addresses: wave-uniform base (block, K step, row j, e tile) + one 32-bit lane offset (K half, feature); features e >= D of
        // the last e tile read a clamped column -- those accumulator rows are never stored.  The first and the last block reach 16
        // rows beyond the sequence: the caller keeps 16 ZERO guard rows on both sides of QKV (TM_QKV_GUARD, transmil.hip), so there
        // is no checked path (a per-lane row test made hipcc spill 600 B per lane here).
        // Buffer loads: ONE descriptor per kernel, a scalar byte offset per (block, K step, row) and a 32-bit lane offset per e tile
        // (with flat loads hipcc materialised 64 per-lane 64-bit addresses ahead of the loop and spilled 700 B per lane).
        const unsigned sblk = (unsigned)__builtin_amdgcn_readfirstlane(rb) * 32u * ldv4;
        auto load_v = [&](int kp) {
#pragma unroll
            for (int et = 0; et < ET; ++et)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    vw[kp][et][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrsrc, lane_off[et], sblk + (unsigned)(16 * kp + j) * ldv4, 0));
        };
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = acc[t][8 * e2 + j];
                tma_h8 ph, pl;
                tma_split8(pv, ph, pl);
#pragma unroll
                for (int et = 0; et < ET; ++et) {
                    const int offb = (t * 2 * EP * 16 + (e2 * EP + 32 * et) * 16) * 2;
                    const tma_h8 wh = *(const tma_h8*)(w2h_b + offb), wl = *(const tma_h8*)(w2l_b + offb);
#if TMA1_ABL & 4
                    o[et][(2 * t + e2) & 15] += (float)wh[0] + (float)wl[1] + (float)ph[0] + (float)pl[1];
#else
                    TMA_MFMA3(o[et], wh, wl, ph, pl);
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (convw && !(TMA1_ABL & 8)) {
                if (t == (MT >= 4 ? MT - 3 : 0)) { load_v(0); load_v(1); __builtin_amdgcn_sched_barrier(0); }
                if (t == MT - 1) { load_v(2); load_v(3); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        const float inv = inv_s2 / sum;
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[et][r] *= inv;
        if (convw && !(TMA1_ABL & 8)) {
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                const tma_h8 bh = Tch[kp * 64 + lane], bl = Tcl[kp * 64 + lane];
#pragma unroll
                for (int et = 0; et < ET; ++et) {
                    tma_h8 ah, al;
                    tma_split8(vw[kp][et], ah, al);
                    TMA_MFMA3(o[et], ah, al, bh, bl);
                }
            }
        }
#if TMA_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        float* op = OUT + (size_t)(rb * 32 + i31) * Di + h * D;
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int e = 32 * et + 8 * gq + 4 * hi;
#if TMA1_ABL & 16
                asm volatile("" :: "v"(o[et][4 * gq]), "v"(o[et][4 * gq + 1]), "v"(o[et][4 * gq + 2]), "v"(o[et][4 * gq + 3]), "v"(op));
#else
                if (e < D) *(f32x4*)(op + e) = f32x4{o[et][4 * gq], o[et][4 * gq + 1], o[et][4 * gq + 2], o[et][4 * gq + 3]};
#endif
            }
    }
}

// GRP key chunks per workgroup (round 4): MT = 6 waves do not divide over a CU's four SIMDs -- two such workgroups sit as 4, 4, 2, 2
// waves, the leg runs at the pace of the SIMDs that carry four, and nothing else finds room on those (tools/coresident_probe.hip).
// GRP = 2 puts two chunks -- two independent 6-wave groups, each with its own staging buffers and online-softmax state, sharing only
// the step barrier -- into ONE 12-wave workgroup: 3 waves on every SIMD.  Same chunk partition, same partials, same merge.
template <int MT, int GRP>
__global__ __launch_bounds__(64 * MT * GRP) void tm_attn3x_kernel(const float* __restrict__ QKV, const float* __restrict__ QL,
                                                           float* __restrict__ part_ms, float* __restrict__ part_o, int npad, int Di,
                                                           float scale, int blocks_per_chunk, int nchunks) {
    constexpr int M = 32 * MT, D = 8 * MT, KS = D / 16, ET = (D + 31) / 32, EP = 32 * ET, LDK = D + 8, VN = 2 * EP * 16;
    __shared__ __attribute__((aligned(16))) _Float16 kh_[GRP][2][32 * LDK], kl_[GRP][2][32 * LDK];   // [key][d]       A operand of GEMM-S
    __shared__ __attribute__((aligned(16))) _Float16 vh_[GRP][2][VN], vl_[GRP][2][VN];                // permuted v^T  A operand of GEMM-PV
    const int grp = (GRP > 1) ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / (64 * MT)) : 0;
    const int tid = threadIdx.x - grp * 64 * MT, lane = tid & 63, wave = tid >> 6, i31 = lane & 31, hi = lane >> 5;
    _Float16 (*kh)[32 * LDK] = kh_[grp]; _Float16 (*kl)[32 * LDK] = kl_[grp];
    _Float16 (*vh)[VN] = vh_[grp]; _Float16 (*vl)[VN] = vl_[grp];
    const int h = blockIdx.y, chunk = blockIdx.x * GRP + grp;
    const int nblk = npad / 32;
    const bool live = chunk < nchunks;                               // (an odd chunk count leaves the last workgroup's second group idle)
    const int kb0 = chunk * blocks_per_chunk, kb1 = live ? min(nblk, kb0 + blocks_per_chunk) : kb0;

    tma_h8 qh[KS], ql[KS];            // B operand of GEMM-S: scale * q_l[l = 32 wave + i31][16 st + 8 hi + j]
    {
        const float* qp = QL + ((size_t)h * M + 32 * wave + i31) * D + 8 * hi;
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = qp[16 * st + j] * scale;
            tma_split8(v, qh[st], ql[st]);
        }
    }
    for (int idx = tid; idx < 2 * VN; idx += 64 * MT) { vh[idx / VN][idx % VN] = (_Float16)0.0f; vl[idx / VN][idx % VN] = (_Float16)0.0f; }
    __syncthreads();                  // the padded e columns stay zero; real columns are overwritten every step
    const int srow = tid / (D / 4), sc4 = tid % (D / 4);
    f32x4 kreg, vreg;
    auto load_kv = [&](int kb) {
        const float* base = QKV + (size_t)(kb * 32 + srow) * 3 * Di + h * D + 4 * sc4;
        kreg = *(const f32x4*)(base + Di);
        vreg = *(const f32x4*)(base + 2 * Di);
    };
    float m_run = -INFINITY, s_run = 0.0f;
    f32x16 o[ET];
#pragma unroll
    for (int et = 0; et < ET; ++et)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[et][r] = 0.0f;

    if (kb0 < kb1) load_kv(kb0);
    // (GRP > 1: the groups share the barrier, so every group runs blocks_per_chunk steps; steps past its own range do nothing)
    for (int it = 0; it < blocks_per_chunk; ++it) {
        const int kb = kb0 + it;
        const bool on = kb < kb1;
        if (GRP == 1 && !on) break;
        const int buf = it & 1;
        if (on) {
            tma_h4 h4, l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const _Float16 x = (_Float16)kreg[j]; h4[j] = x; l4[j] = (_Float16)(kreg[j] - (float)x); }
            *(tma_h4*)(&kh[buf][srow * LDK + 4 * sc4]) = h4;
            *(tma_h4*)(&kl[buf][srow * LDK + 4 * sc4]) = l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const _Float16 x = (_Float16)vreg[j];
                const int p = tma_perm_index(srow, EP, 4 * sc4 + j);
                vh[buf][p] = x; vl[buf][p] = (_Float16)(vreg[j] - (float)x);
            }
        }
        __syncthreads();      // (an LDS-only barrier with the prefetch issued ahead of it measured the same: 149.3 vs 150.8 us)
        if (!on) continue;
        if (kb + 1 < kb1) load_kv(kb + 1);

        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
#if TMA_PRIO3
        __builtin_amdgcn_s_setprio(TMA_PRIO3);
#endif
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const int off = i31 * LDK + 16 * st + 8 * hi;
            const tma_h8 ah = *(const tma_h8*)(&kh[buf][off]), al = *(const tma_h8*)(&kl[buf][off]);
            TMA_MFMA3(S, ah, al, qh[st], ql[st]);
        }
#if TMA_PRIO3
        __builtin_amdgcn_s_setprio(0);
#endif
        float bm = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) bm = fmaxf(bm, S[r]);
        bm = fmaxf(bm, tma_xor32(bm));
        const float mn = fmaxf(m_run, bm);
        const float alpha = __expf(m_run - mn);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = __expf(S[r] - mn); ps += S[r]; }
        ps += tma_xor32(ps);
        s_run = s_run * alpha + ps;
        m_run = mn;
#pragma unroll
        for (int et = 0; et < ET; ++et)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[et][r] *= alpha;
#if TMA_PRIO3
        __builtin_amdgcn_s_setprio(TMA_PRIO3);
#endif
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = S[8 * e2 + j];
            tma_h8 ph, pl;
            tma_split8(pv, ph, pl);
#pragma unroll
            for (int et = 0; et < ET; ++et) {
                const int off = ((e2 * EP + 32 * et + i31) * 2 + hi) * 8;
                const tma_h8 wh = *(const tma_h8*)(&vh[buf][off]), wl = *(const tma_h8*)(&vl[buf][off]);
                TMA_MFMA3(o[et], wh, wl, ph, pl);
            }
        }
#if TMA_PRIO3
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    if (!live) return;
    const size_t row = ((size_t)h * nchunks + chunk) * M + 32 * wave + i31;
    if (hi == 0) { part_ms[row * 2] = m_run; part_ms[row * 2 + 1] = s_run; }
#pragma unroll
    for (int et = 0; et < ET; ++et)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int e = 32 * et + 8 * gq + 4 * hi;
            if (e < D) *(f32x4*)(part_o + row * D + e) = f32x4{o[et][4 * gq], o[et][4 * gq + 1], o[et][4 * gq + 2], o[et][4 * gq + 3]};
        }
}

// ---------------------------------------------------------------------------------------------------------------
// host side (called from transmil.hip).  Return ACMIL_ERR_UNSUPPORTED when the geometry has no instantiation.
int tm_attn_fused_supported(int Di) { return Di == 128 || Di == 256 || Di == 384 || Di == 512; }

size_t tm_attn3_partial_bytes(int npad, int Di) {
    const int m = Di / 2, d = Di / TMA_HEADS;
    (void)npad;
    return (size_t)TMA_HEADS * TMA_MAX_CHUNKS * m * (2 + d) * sizeof(float) + 1024;   // <= TMA_MAX_CHUNKS chunks (tm_attn3_launch)
}

// ACMIL_TM_ATTN_FP32=1 selects the exact-fp32 MFMA kernels (A/B reference); default = split-f16
static bool tm_attn_exact() { static const bool v = ACMIL_AB_ENV("ACMIL_TM_ATTN_FP32") != nullptr; return v; }

template <int MT>
static int tm_attn1x_launch(const float* QKV, const float* KL, const float* W2, float* OUT, int npad, int Di, float scale, hipStream_t st,
                            const float* convw) {
    constexpr int M = 32 * MT, D = 8 * MT, ET = (D + 31) / 32, EP = 32 * ET;
    const size_t lds = ((size_t)2 * M * (D + 8) + (size_t)2 * MT * 2 * EP * 16) * sizeof(_Float16) + 2 * 4 * 64 * 16;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)tm_attn1x_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ACMIL_ERR_LAUNCH;
        attr_set = true;
    }
    const int nblk = npad / 32;
    int gx = (nblk + 7) / 8; if (gx > 32) gx = 32;
    hipLaunchKernelGGL(tm_attn1x_kernel<MT>, dim3(gx, TMA_HEADS), dim3(512), lds, st, QKV, KL, W2, OUT, npad, Di, scale, convw);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

template <int MT>
static int tm_attn1_launch(const float* QKV, const float* KL, const float* W2, float* OUT, int npad, int Di, float scale, hipStream_t st,
                           const float* convw, int* conv_done) {
    // ACMIL_TM_SEQCONV_PASS=1: keep the residual convolution as a pass of its own (A/B knob); the exact-fp32 leg never folds it
    static const bool conv_pass = ACMIL_AB_ENV("ACMIL_TM_SEQCONV_PASS") != nullptr;
    if (!tm_attn_exact()) {
        const bool fold = convw != nullptr && !conv_pass;
        *conv_done = fold ? 1 : 0;
        return tm_attn1x_launch<MT>(QKV, KL, W2, OUT, npad, Di, scale, st, fold ? convw : nullptr);
    }
    *conv_done = 0;
    constexpr int M = 32 * MT, D = 8 * MT, ET = (D + 31) / 32, EP = 32 * ET;
    const size_t lds = ((size_t)D * (M + 4) + (size_t)M * (EP + 8)) * sizeof(float);
    static bool attr_set = false;      // idempotent attribute; racing callers set the same value
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)tm_attn1_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ACMIL_ERR_LAUNCH;
        attr_set = true;
    }
    const int nblk = npad / 32;
    int gx = (nblk + 3) / 4; if (gx > 32) gx = 32;                 // 8 heads x 32 = 256 workgroups, each wave loops over its row blocks
    hipLaunchKernelGGL(tm_attn1_kernel<MT>, dim3(gx, TMA_HEADS), dim3(256), lds, st, QKV, KL, W2, OUT, npad, Di, scale);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// convw [8][33] (res_conv.weight) or null; *conv_done = 1 when the leg added the residual convolution of v itself
int tm_attn1_fused(const float* QKV, const float* KL, const float* W2, float* OUT, int npad, int Di, float scale, hipStream_t st,
                   const float* convw, int* conv_done) {
    switch (Di) {
        case 128: return tm_attn1_launch<2>(QKV, KL, W2, OUT, npad, Di, scale, st, convw, conv_done);
        case 256: return tm_attn1_launch<4>(QKV, KL, W2, OUT, npad, Di, scale, st, convw, conv_done);
        case 384: return tm_attn1_launch<6>(QKV, KL, W2, OUT, npad, Di, scale, st, convw, conv_done);
        case 512: return tm_attn1_launch<8>(QKV, KL, W2, OUT, npad, Di, scale, st, convw, conv_done);
    }
    return ACMIL_ERR_UNSUPPORTED;
}

template <int MT>
static int tm_attn3_launch(const float* QKV, const float* QL, float* AV, float* part, int npad, int Di, float scale, hipStream_t st, bool shared) {
    constexpr int M = 32 * MT, D = 8 * MT;
    const int nblk = npad / 32;
    // chunks per head: 64 = 512 workgroups = two per CU.  Measured (ACMIL_TM_ATTN3_CHUNKS, whole forward on one box): 64 chunks 2.396 ms,
    // 96 (three workgroups per CU) 2.414, 128 2.475 -- the leg is not occupancy-limited, more chunks only add partials.
    // shared = the Moore-Penrose chain runs beside this launch on the side stream (transmil.hip): 32 chunks = ONE workgroup per CU leave
    // that chain room on every SIMD -- the leg alone takes 173 instead of 150 us, the pair 234 us instead of 266 (forward 2.18 vs 2.21 ms)
    static const int forced = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_ATTN3_CHUNKS"); const int v = e ? atoi(e) : 0; return v < 1 ? 0 : (v > TMA_MAX_CHUNKS ? TMA_MAX_CHUNKS : v); }();
    const int want = forced ? forced : (shared ? 32 : 64);
    int nchunks = nblk < want ? nblk : want;
    const int bpc = (nblk + nchunks - 1) / nchunks;
    nchunks = (nblk + bpc - 1) / bpc;
    float* part_ms = part;
    float* part_o = part + (((size_t)TMA_HEADS * nchunks * M * 2 + 63) & ~(size_t)63);
    // MT = 6 alone on the GPU: two chunks per 12-wave workgroup (3 waves on every SIMD instead of 4, 4, 2, 2: 2.21 vs 2.23 ms per forward);
    // beside the Moore-Penrose chain the 6-wave workgroups stay, one per CU (32 chunks): measured 2.07 ms against 2.12 with the 12-wave
    // form, whose three waves per SIMD leave the chain's waves too few issue slots.  ACMIL_TM_ATTN3_GRP=1 / 2 forces one form (A/B).
    static const int grp_forced = [] { const char* e = ACMIL_AB_ENV("ACMIL_TM_ATTN3_GRP"); return e ? atoi(e) : 0; }();
    constexpr int GRP = (MT == 6) ? 2 : 1;
    const bool grp1 = grp_forced == 1 || (grp_forced != 2 && shared);
    if (tm_attn_exact()) hipLaunchKernelGGL(tm_attn3_kernel<MT>, dim3(nchunks, TMA_HEADS), dim3(64 * MT), 0, st, QKV, QL, part_ms, part_o, npad, Di, scale, bpc);
    else if (GRP == 1 || grp1) hipLaunchKernelGGL((tm_attn3x_kernel<MT, 1>), dim3(nchunks, TMA_HEADS), dim3(64 * MT), 0, st, QKV, QL, part_ms, part_o, npad, Di, scale, bpc, nchunks);
    else hipLaunchKernelGGL((tm_attn3x_kernel<MT, GRP>), dim3((nchunks + GRP - 1) / GRP, TMA_HEADS), dim3(64 * MT * GRP), 0, st, QKV, QL, part_ms, part_o, npad, Di, scale, bpc, nchunks);
    if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    hipLaunchKernelGGL(tm_attn3_merge_kernel, dim3((M + 3) / 4, TMA_HEADS), dim3(256), 0, st, part_ms, part_o, AV, M, D, nchunks);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

int tm_attn3_fused(const float* QKV, const float* QL, float* AV, float* part, int npad, int Di, float scale, hipStream_t st, bool shared) {
    switch (Di) {
        case 128: return tm_attn3_launch<2>(QKV, QL, AV, part, npad, Di, scale, st, shared);
        case 256: return tm_attn3_launch<4>(QKV, QL, AV, part, npad, Di, scale, st, shared);
        case 384: return tm_attn3_launch<6>(QKV, QL, AV, part, npad, Di, scale, st, shared);
        case 512: return tm_attn3_launch<8>(QKV, QL, AV, part, npad, Di, scale, st, shared);
    }
    return ACMIL_ERR_UNSUPPORTED;
}
