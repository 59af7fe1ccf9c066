// ga_train.hip -- the pieces of ACMIL_GA.forward that only run in training mode
//   STKIM top-k + random subset selection     architecture/transformer.py:311-317
//   mask application                          architecture/transformer.py:318-320
//   masked softmax + attention-weighted sum   architecture/transformer.py:322-324 (on the saved h)
// plus their C entry points (include/acmil_hip.h).  All of it is HBM-bound streaming / selection work:
// coalesced float4 row reads, LDS for the per-tile probabilities, wave shuffles for reductions.
#include "ga_common.h"
#include "ga_train_internal.h"

// ------------------------------------------------------------------------------------------------
// top-k.  Order-preserving key: (monotone uint32 image of the fp32 score) << 32 | (0xFFFFFFFF - index),
// so a max over keys picks the largest score and, among equal scores, the LOWEST index.
// ------------------------------------------------------------------------------------------------
#define STKIM_CHUNK 4096
#define STKIM_EPT (STKIM_CHUNK / 256)

__device__ static inline unsigned long long stkim_key(float v, unsigned idx) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// max of a 64-bit key over the wave with DPP row operations (no LDS round trips): row_shr 1,2,4,8 leave each row's max in its
// lane 15, row_bcast15 / row_bcast31 carry it across the rows into lane 63.  Lanes a DPP step has no source for keep their own
// value (old = self), which is harmless for a max.
template <int CTRL, int ROW_MASK>
__device__ static inline unsigned long long stkim_dpp_max(unsigned long long v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned lo2 = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi2 = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)hi2 << 32) | lo2;
    return o > v ? o : v;
}
__device__ static inline unsigned long long wave_max_key(unsigned long long v) {
    v = stkim_dpp_max<0x111, 0xf>(v);
    v = stkim_dpp_max<0x112, 0xf>(v);
    v = stkim_dpp_max<0x114, 0xf>(v);
    v = stkim_dpp_max<0x118, 0xf>(v);
    v = stkim_dpp_max<0x142, 0xa>(v);
    v = stkim_dpp_max<0x143, 0xc>(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// block-wide max: wave maxima through a double-buffered LDS row (ONE barrier per call; `round` alternates the row)
__device__ static inline unsigned long long block_max_key(unsigned long long key, unsigned long long (*red)[4], int round) {
    const unsigned long long wm = wave_max_key(key);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long* row = red[round & 1];
    if (lane == 0) row[wave] = wm;
    __syncthreads();
    unsigned long long m = row[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) m = row[w] > m ? row[w] : m;
    return m;
}

// merge of one branch's candidates by one wave: E keys per lane, k rounds of wave-wide max
template <int E>
__device__ static inline void stkim_merge_wave(const unsigned long long* __restrict__ c, int ncand, int k, int lane, unsigned* sel,
                                               int64_t* __restrict__ topk_row) {
    unsigned long long keys[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * 64 + lane;
        keys[e] = i < ncand ? __hip_atomic_load(c + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    for (int j = 0; j < k; ++j) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int e = 0; e < E; ++e) best = keys[e] > best ? keys[e] : best;
        best = wave_max_key(best);
        if (lane == 0) {
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull);
            sel[j] = idx;
            topk_row[j] = (int64_t)idx;
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (keys[e] == best) keys[e] = 0ull;      // keys are unique (index in the low word); 0 = taken / padding
    }
}

// Single launch, grid (nchunks, K):
//   every block extracts the k best keys of its chunk of row br -> cand[br][chunk][k], then counts itself in `arrive`;
//   the block that arrives last (any block: the merge order is fixed by index, not by arrival) finishes all K branches,
//   one wave per branch: merge the candidates into the sorted top-k, pick the masked subset (the columns with the m smallest
//   uniforms, argsort ascending, first m, in that order) and -- when A_mask is given -- write the -1e9 mask into the scores
//   (architecture/transformer.py:311-320).  `arrive` is left at zero for the next launch.
// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> 4 x 32 random bits.  The production draw of the STKIM
// uniforms (`torch.rand(K, k)` of the reference, architecture/transformer.py:316): any iid U[0,1) stream is the same distribution,
// so the step draws its own on the device -- keyed on (seed, offset = the caller's step number, branch, column) -- instead of paying
// a torch.rand launch per step; injected uniforms remain for the parity tests.
__device__ static inline void stkim_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ static inline float stkim_uniform(unsigned long long seed, unsigned long long offset, unsigned branch, unsigned col) {
    unsigned r[4];
    stkim_philox(col, branch, (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
    return (float)(r[0] >> 8) * (1.0f / 16777216.0f);      // 24 random bits: [0, 1)
}

#define STKIM_MERGE_E 32      // candidates per lane in the merge: nchunks * k <= 64 * 32
// timing-only stamps (tools/stkim_probe.py builds a variant with -DSTKIM_PROF): 100 MHz wall clock into the control block, bytes 64..
#ifdef STKIM_PROF
#define STKIM_STAMP(slot, cond) do { if ((cond) && threadIdx.x == 0) ((unsigned long long*)arrive)[8 + (slot)] = wall_clock64(); } while (0)
#else
#define STKIM_STAMP(slot, cond) do { } while (0)
#endif
// seg: the bags of the launch (one for the stand-alone selection; a group for the multi-bag training step): chunk c of the grid
// belongs to bag seg-find(c); scores / A_mask are [K][ldA] with bag b in columns row0[b] .. (indices in the keys and in the outputs
// are LOCAL to the bag, as the reference's are); one arrival word per bag, the last block of a BAG finishes that bag's K branches.
__global__ __launch_bounds__(256) void stkim_fused_kernel(const float* __restrict__ scores, float* __restrict__ A_mask, GaSeg seg, int ldA, int K,
                                                          int k, int m, const float* __restrict__ uniforms,
                                                          unsigned long long rng_seed, unsigned long long rng_offset,
                                                          unsigned long long* __restrict__ cand, unsigned* __restrict__ arrive,
                                                          int64_t* __restrict__ topk_idx, int64_t* __restrict__ masked_idx) {
    __shared__ unsigned long long red[2][4];
    __shared__ unsigned sel[4][64];
    __shared__ int is_last;
    const int br = blockIdx.y, nch_all = gridDim.x;
    const GaSegTile sg = ga_seg_find<STKIM_CHUNK>(seg, blockIdx.x);
    const int bag = sg.bag, N = sg.nend - sg.row0, chunk = blockIdx.x - sg.tile0, nch = (N + STKIM_CHUNK - 1) / STKIM_CHUNK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0;
    STKIM_STAMP(0, first);
    {
        const float* row = scores + (size_t)br * ldA + sg.row0;
        unsigned long long keys[STKIM_EPT];
#pragma unroll
        for (int e = 0; e < STKIM_EPT; ++e) {
            const unsigned idx = (unsigned)chunk * STKIM_CHUNK + e * 256 + tid;
            keys[e] = idx < (unsigned)N ? stkim_key(row[idx], idx) : 0ull;
        }
        unsigned long long* out = cand + ((size_t)br * nch_all + blockIdx.x) * k;
#ifdef STKIM_PROF
        if (first && keys[0] == 1ull) out[0] = 0ull;      // (the stamp below waits for the loads)
#endif
        STKIM_STAMP(1, first);
        for (int j = 0; j < k; ++j) {
            unsigned long long best = 0ull;
#pragma unroll
            for (int e = 0; e < STKIM_EPT; ++e) best = keys[e] > best ? keys[e] : best;
            const unsigned long long win = block_max_key(best, red, j);
            if (tid == 0) __hip_atomic_store(out + j, win, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through (sc1): see below
#pragma unroll
            for (int e = 0; e < STKIM_EPT; ++e)
                if (keys[e] == win) keys[e] = 0ull;  // 0 = taken / padding
        }
    }
    // publish the candidates: they were stored write-through (sc1) by this lane, so draining its stores is the release (an
    // agent release fence would write back every dirty L2 line -- here the scores and the saved h of the score pass);
    // count this block; the last one reads everybody's candidates with sc1 loads
    STKIM_STAMP(2, first);
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = atomicAdd(arrive + bag, 1u);
        is_last = (t == (unsigned)(nch * K) - 1u) ? 1 : 0;
        if (is_last) atomicExch(arrive + bag, 0u);
    }
    __syncthreads();
    STKIM_STAMP(3, first);
    if (!is_last) return;
    STKIM_STAMP(4, true);
    const int ncand = nch * k;
    for (int b = wave; b < K; b += 4) {
        const unsigned long long* c = cand + ((size_t)b * nch_all + sg.tile0) * k;
        const size_t orow = (size_t)bag * K + b;      // output / uniform row of (bag, branch); the Philox stream is keyed on it as well
        int64_t* trow = topk_idx + orow * k;
        if (ncand <= 64) stkim_merge_wave<1>(c, ncand, k, lane, sel[wave], trow);
        else if (ncand <= 128) stkim_merge_wave<2>(c, ncand, k, lane, sel[wave], trow);
        else if (ncand <= 256) stkim_merge_wave<4>(c, ncand, k, lane, sel[wave], trow);
        else if (ncand <= 512) stkim_merge_wave<8>(c, ncand, k, lane, sel[wave], trow);
        else stkim_merge_wave<STKIM_MERGE_E>(c, ncand, k, lane, sel[wave], trow);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // sel[] written by lane 0 is visible to the wave
        STKIM_STAMP(5, b == 0);
        if (m > 0) {
            // this lane's uniform: injected (parity tests) or drawn here (uniforms == null); rank = position in argsort ascending
            const float mine = lane < k ? (uniforms ? uniforms[orow * k + lane] : stkim_uniform(rng_seed, rng_offset, (unsigned)orow, (unsigned)lane)) : 2.0f;
            int rank = 0;
            for (int j = 0; j < k; ++j) {
                const float uj = __shfl(mine, j);
                rank += (uj < mine || (uj == mine && j < lane)) ? 1 : 0;
            }
            if (lane < k && rank < m) {
                const unsigned n = sel[wave][lane];
                masked_idx[orow * m + rank] = (int64_t)n;
                if (A_mask && n < (unsigned)N) A_mask[(size_t)b * ldA + sg.row0 + n] = -1e9f;
            }
        }
        STKIM_STAMP(6, b == 0);
    }
}

// workspace: [256-byte control block: arrival counter][candidates]
extern "C" size_t acmil_stkim_workspace_bytes(int N, int K, int k) {
    if (N <= 0 || K <= 0 || k <= 0) return 0;
    const size_t nch = (size_t)(N + STKIM_CHUNK - 1) / STKIM_CHUNK;
    return 256 + (((size_t)K * nch * k * sizeof(unsigned long long) + 255) & ~(size_t)255);
}

// candidate scratch of a launch: K rows of (chunks of all bags) x k keys
size_t stkim_cand_bytes(int N, int nbags, int K, int k) {
    const size_t nch = (size_t)(N + STKIM_CHUNK - 1) / STKIM_CHUNK + (size_t)(nbags > 1 ? nbags : 0);      // every bag of a group may end on a partial chunk
    return (((size_t)K * nch * (k > 0 ? k : 1) * sizeof(unsigned long long)) + 255) & ~(size_t)255;
}

// shared by acmil_stkim_select and the fused training step (ga_step.hip): one launch; `arrive` (one word per bag) must be zero and is left zero
int stkim_launch(const float* scores, float* A_mask, int N, int K, int k, int m, const float* uniforms, int64_t* topk_idx,
                 int64_t* masked_idx, unsigned long long* cand, unsigned* arrive, hipStream_t st, unsigned long long rng_seed,
                 unsigned long long rng_offset, const GaSeg* seg) {
    if (N <= 0 || K <= 0 || k <= 0 || k > 64 || k > N || m < 0 || m > k) return ACMIL_ERR_SHAPE;
    if (!scores || !topk_idx || !cand || !arrive || (m > 0 && !masked_idx)) return ACMIL_ERR_NULL;      // uniforms null = device draw
    const GaSeg one = ga_seg_single(N);
    const GaSeg& S = seg ? *seg : one;
    if (S.n < 1 || S.n > GA_SEG_MAX || S.row0[0] != 0 || S.row0[S.n] != N) return ACMIL_ERR_SHAPE;
    for (int b = 0; b < S.n; ++b) {
        const int nb = S.row0[b + 1] - S.row0[b];
        if (nb < k) return ACMIL_ERR_SHAPE;                                            // every bag holds its own top-k
        if ((size_t)((nb + STKIM_CHUNK - 1) / STKIM_CHUNK) * k > 64 * STKIM_MERGE_E) return ACMIL_ERR_UNSUPPORTED;  // a bag up to ~131k rows at k=64, ~800k at k=10
    }
    const int nch = ga_seg_tiles(S, STKIM_CHUNK);
    hipLaunchKernelGGL(stkim_fused_kernel, dim3(nch, K), dim3(256), 0, st, scores, A_mask, S, N, K, k, m, uniforms, rng_seed, rng_offset,
                       cand, arrive, topk_idx, masked_idx);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_stkim_select_rng(const float* scores, int N, int K, int k, int m, const float* uniforms,
                                      unsigned long long seed, unsigned long long offset, int64_t* topk_idx, int64_t* masked_idx,
                                      void* workspace, void* stream) {
    if (N <= 0 || K <= 0 || k <= 0 || k > 64 || k > N || m < 0 || m > k) return ACMIL_ERR_SHAPE;
    if (!scores || !topk_idx || !workspace || (m > 0 && !masked_idx)) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    // stand-alone call on a caller-provided scratch buffer: the arrival counter is zeroed here (the fused step keeps its own)
    if (hipMemsetAsync(workspace, 0, 4, st) != hipSuccess) return ACMIL_ERR_LAUNCH;
    return stkim_launch(scores, nullptr, N, K, k, m, uniforms, topk_idx, masked_idx,
                        (unsigned long long*)((char*)workspace + 256), (unsigned*)workspace, st, seed, offset);
}

extern "C" int acmil_stkim_select(const float* scores, int N, int K, int k, int m, const float* uniforms,
                                  int64_t* topk_idx, int64_t* masked_idx, void* workspace, void* stream) {
    if (m > 0 && !uniforms) return ACMIL_ERR_NULL;      // this entry takes the draw from the caller; _rng draws on the device
    return acmil_stkim_select_rng(scores, N, K, k, m, uniforms, 0ull, 0ull, topk_idx, masked_idx, workspace, stream);
}

// ------------------------------------------------------------------------------------------------
// mask application: A[k, masked_idx[k, j]] = -1e9        (transformer.py:318-320)
// ------------------------------------------------------------------------------------------------
__global__ void ga_apply_mask_kernel(float* __restrict__ A, int N, int K, const int64_t* __restrict__ midx, int m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * m) {
        const int br = i / m;
        const int64_t n = midx[i];
        if (n >= 0 && n < N) A[(size_t)br * N + n] = -1e9f;
    }
}

// ------------------------------------------------------------------------------------------------
// pooling over a saved h: per 128-row tile the online-softmax partial (m, l, sum_n e^{s-m} h[n][:]) in
// the same format the fused forward emits (so ga_merge_kernel / ga_heads_kernel finish both).
// 256 threads; LPR = Di/4 lanes cover one h row with float4 loads, RPI = 256/LPR rows per iteration.
// ------------------------------------------------------------------------------------------------
// gram != nullptr (training step): also the tile's Gram partial of the UNNORMALISED probabilities, gram[tile][i*KP+j] =
// sum_n e_i(n) e_j(n), i <= j, e_k(n) = exp(s_k(n) - m_tile,k) -- rescaled by the tail kernel (ga_step.hip) for the
// diversity loss (Step3_WSI_classification_ACMIL.py:207-212).
// seg: one bag, or a group of bags whose tiles are cut per bag (tile blockIdx.x -> ga_seg_find); A is [K][N] over ALL rows.
template <int KP>
__global__ __launch_bounds__(256) void ga_pool_kernel(const float* __restrict__ h, const float* __restrict__ A, GaSeg seg, int N,
                                                      int K, int Di, float* __restrict__ part, float* __restrict__ gram,
                                                      const unsigned* __restrict__ cond, unsigned* __restrict__ cond_count) {
    if (cond) {      // predicated launch: the exact-fp32 repeat of a flagged bag (ga_forward.hip); counted once per launch that ran
        if (__builtin_nontemporal_load(cond) == 0u) return;
        if (cond_count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(cond_count, 1u);
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* p_lds = (float*)smem;                     // [128][KP]
    float* stat = p_lds + 128 * KP;                  // [KP][2] : m, l
    float* red = stat + 2 * KP;                      // [RPI][KP][Di] cross-row-group reduction
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const GaSegTile sg = ga_seg_find<GA_POOL_ROWS>(seg, blockIdx.x);
    const int n0 = sg.n0;
    const int rows = min(GA_POOL_ROWS, sg.nend - n0);
    // ---- tile statistics: wave w handles branches w, w+4, ...; 64 lanes x 2 rows each
    for (int k = wave; k < KP; k += 4) {
        float s0 = -INFINITY, s1 = -INFINITY;
        if (k < K) {
            if (lane < rows) s0 = A[(size_t)k * N + n0 + lane];
            if (lane + 64 < rows) s1 = A[(size_t)k * N + n0 + lane + 64];
        }
        float m = fmaxf(s0, s1);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e0 = (k < K && lane < rows) ? __expf(s0 - m) : 0.0f;
        const float e1 = (k < K && lane + 64 < rows) ? __expf(s1 - m) : 0.0f;
        float l = e0 + e1;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) l += __shfl_xor(l, o);
        p_lds[lane * KP + k] = e0;
        p_lds[(lane + 64) * KP + k] = e1;
        if (lane == 0) { stat[2 * k] = m; stat[2 * k + 1] = l; }
    }
    __syncthreads();
    if (gram && wave == 0) {
        float* go = gram + (size_t)blockIdx.x * KP * KP;
#pragma unroll
        for (int i = 0; i < KP; ++i)
#pragma unroll
            for (int j = i; j < KP; ++j) {
                float v = fmaf(p_lds[lane * KP + i], p_lds[lane * KP + j], p_lds[(lane + 64) * KP + i] * p_lds[(lane + 64) * KP + j]);
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) go[i * KP + j] = v;
            }
    }
    const int LPR = Di / 4, RPI = 256 / LPR;
    const int rsub = tid / LPR, c4 = tid % LPR;
    float acc[KP][4];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0f;
    if (rsub < RPI) {
        const f32x4* hp = (const f32x4*)(h + (size_t)n0 * Di) + c4;
#pragma unroll 4
        for (int n = rsub; n < rows; n += RPI) {
            const f32x4 v = __builtin_nontemporal_load(hp + (size_t)n * LPR);
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float p = p_lds[n * KP + k];
                acc[k][0] = fmaf(p, v[0], acc[k][0]); acc[k][1] = fmaf(p, v[1], acc[k][1]);
                acc[k][2] = fmaf(p, v[2], acc[k][2]); acc[k][3] = fmaf(p, v[3], acc[k][3]);
            }
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) *(f32x4*)(red + ((size_t)(rsub * KP + k) * Di) + 4 * c4) = f32x4{acc[k][0], acc[k][1], acc[k][2], acc[k][3]};
    }
    __syncthreads();
    const size_t PS = 2 + Di;
    float* out = part + (size_t)blockIdx.x * K * PS;
    for (int k = 0; k < K; ++k) {
        if (tid < 2) out[k * PS + tid] = stat[2 * k + tid];
        for (int di = tid; di < Di; di += 256) {
            float s = 0.0f;
            for (int r = 0; r < RPI; ++r) s += red[(size_t)(r * KP + k) * Di + di];
            out[k * PS + 2 + di] = s;
        }
    }
}

// one workgroup per 128-row tile; gram: see the kernel.  Shared by acmil_ga_pool, acmil_attn_pool and the fused step.
int ga_pool_launch(const float* h, const float* A, int N, int K, int Di, float* part, float* gram, hipStream_t st, const unsigned* cond,
                   unsigned* cond_count, const GaSeg* seg) {
    if (Di % 64 != 0 || Di / 4 > 256 || Di > 1024) return ACMIL_ERR_UNSUPPORTED;
    if (K > ACMIL_MAX_TOKENS) return ACMIL_ERR_UNSUPPORTED;
    const int KP = ga_kp(K);
    const int RPI = 256 / (Di / 4) > 0 ? 256 / (Di / 4) : 1;
    const size_t lds = (size_t)(128 * KP + 2 * KP + (size_t)RPI * KP * Di) * sizeof(float);
    if (lds > 160 * 1024) return ACMIL_ERR_UNSUPPORTED;
    void (*kern)(const float*, const float*, GaSeg, int, int, int, float*, float*, const unsigned*, unsigned*) =
        KP == 1 ? ga_pool_kernel<1> : KP == 5 ? ga_pool_kernel<5> : KP == 8 ? ga_pool_kernel<8> : ga_pool_kernel<16>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ACMIL_ERR_LAUNCH;
    const GaSeg one = ga_seg_single(N);
    const GaSeg& S = seg ? *seg : one;
    if (S.n < 1 || S.n > GA_SEG_MAX || S.row0[S.n] != N) return ACMIL_ERR_SHAPE;
    hipLaunchKernelGGL(kern, dim3(ga_seg_tiles(S, GA_POOL_ROWS)), dim3(256), lds, st, h, A, S, N, K, Di, part, gram, cond, cond_count);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_ga_pool(const float* h, float* A, int N, const void* packed, int D, int Di, int Da, int K, int C,
                             int mode, const int64_t* masked_idx, int n_masked, float* sub_preds, float* slide_pred,
                             float* afeat, float* bag_feat, int has_bag_head, void* workspace, void* stream) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (N <= 0 || n_masked < 0) return ACMIL_ERR_SHAPE;
    if (!h || !A || !packed || !workspace || (n_masked > 0 && !masked_idx)) return ACMIL_ERR_NULL;
    if (Di > 1024) return ACMIL_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n_masked > 0) {
        const int tot = K * n_masked;
        hipLaunchKernelGGL(ga_apply_mask_kernel, dim3((tot + 63) / 64), dim3(64), 0, st, A, N, K, masked_idx, n_masked);
        if (hipGetLastError() != hipSuccess) return ACMIL_ERR_LAUNCH;
    }
    const GaLayout L = ga_layout(D, Di, K, C, mode);
    const int tiles = ga_pool_tiles(N);
    float* part = (float*)((char*)workspace + GA_CTRL_BYTES);     // same layout as acmil_ga_forward: control block first
    rc = ga_pool_launch(h, A, N, K, Di, part, nullptr, st, nullptr, nullptr);
    if (rc != ACMIL_OK) return rc;
    return ga_finish(part, tiles, packed, L, sub_preds, slide_pred, afeat, bag_feat, has_bag_head, st);
}

// Attention pooling without heads: afeat [K, Di] = softmax_N(A) h for raw scores A [K, N] (not modified) -- the
// `F.softmax(A, dim=1); torch.mm(A, h)` pair of the other gated-attention consumers (architecture/Attention.py:67-68,
// ibmil.py:73-74, clam.py:163-190).  Same tile partials + fixed-order merge as acmil_ga_pool.
extern "C" size_t acmil_attn_pool_workspace_bytes(int N, int Di, int K) {
    if (N <= 0 || Di <= 0 || K <= 0) return 0;
    return (((size_t)ga_pool_tiles(N) * K * ga_part_stride(Di) * sizeof(float) + 255) & ~(size_t)255) + (size_t)K * Di * sizeof(float) + 256;
}

extern "C" int acmil_attn_pool(const float* h, const float* A, int N, int Di, int K, float* afeat, void* workspace, void* stream) {
    if (N <= 0 || Di <= 0 || K <= 0) return ACMIL_ERR_SHAPE;
    if (Di % 64 != 0 || Di > 1024 || K > ACMIL_MAX_TOKENS) return ACMIL_ERR_UNSUPPORTED;
    if (!h || !A || !afeat || !workspace) return ACMIL_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = ga_pool_tiles(N);
    float* part = (float*)workspace;
    const int rc = ga_pool_launch(h, A, N, K, Di, part, nullptr, st, nullptr, nullptr);
    if (rc != ACMIL_OK) return rc;
    GaLayout L; L.K = K; L.Di = Di; L.C = 1; L.D = 0; L.ND = Di / 32; L.mode = 0;
    return ga_finish(part, tiles, nullptr, L, nullptr, nullptr, afeat, nullptr, 0, st);
}
