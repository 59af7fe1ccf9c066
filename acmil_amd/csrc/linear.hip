// linear.hip -- host side of the packed-weight Linear kernel (linear_kernel.h): weight packing and the C entry points.
#include "linear_kernel.h"

// Packed stream of W [n_out, K] (row-major, leading dimension ldw) as output-column chunks (lin_plan): n_out / 256 chunks of ND = 8
// (256 columns) then one ND = 4 chunk when n_out % 256 == 128 -- except n_out = 384, which runs as 2 chunks of ND = 6
// (192 columns: no half-rate remainder launch; the TransMIL width and CLIP-L's D_inner: 272 vs 323 us at M = 100 000, K = 768).  Chunk = K/16 steps; step = ND "hi"
// fragment rows then ND "lo" rows; fragment row (step s, tile d): lane (i = lane & 31, hi = lane >> 5) holds the 8 f16 halves of
// W[col0 + 32 d + i][16 s + 8 hi .. + 7].
struct LinPackArgs { const float* W; char* out; int ldw, n_out, K; };
struct LinPlan { int nd, nmain, nd_rem; };      // nmain chunks of 32 * nd columns, then one chunk of 32 * nd_rem columns (0 = none)
__host__ __device__ static inline LinPlan lin_plan(int n_out) {
    if (n_out == 384) return LinPlan{6, 2, 0};
    return LinPlan{8, n_out / 256, (n_out % 256) ? 4 : 0};
}

__device__ __forceinline__ void lin_pack_rows(const LinPackArgs& a, unsigned block) {
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const size_t row = (size_t)block * 4 + (threadIdx.x >> 6);
    const int S1 = a.K / 16;
    const LinPlan P = lin_plan(a.n_out);
    const size_t per = (size_t)S1 * 2 * P.nd;                 // fragment rows of one main chunk
    const size_t rows_full = (size_t)P.nmain * per;
    const size_t rows_all = rows_full + (size_t)S1 * 2 * P.nd_rem;
    if (row >= rows_all) return;
    int ND, col0; size_t r;
    if (row < rows_full) { ND = P.nd; const size_t c = row / per; col0 = (int)c * 32 * P.nd; r = row - c * per; }
    else { ND = P.nd_rem; col0 = P.nmain * 32 * P.nd; r = row - rows_full; }
    const int d = r % ND; r /= ND;
    const int part = r & 1; const int s = (int)(r >> 1);
    const float* src = a.W + (size_t)(col0 + 32 * d + i) * a.ldw + 16 * s + 8 * hi;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)src[j];
        v[j] = part ? (_Float16)(src[j] - (float)h) : h;
    }
    *(f16x8*)(a.out + row * GA_FRAG_ROW + lane * 16) = v;
}
__global__ __launch_bounds__(256) void lin_pack_kernel(LinPackArgs a) { lin_pack_rows(a, blockIdx.x); }

// several weight matrices in ONE launch (grid.y = job): TransMIL packs its five Linear layers at the start of a forward
#define LIN_PACK_MAX_JOBS 8
struct LinPackMulti { LinPackArgs job[LIN_PACK_MAX_JOBS]; };
__global__ __launch_bounds__(256) void lin_pack_multi_kernel(LinPackMulti m) {
    // (the argument struct is only read, with a wave-uniform index)
    lin_pack_rows(m.job[blockIdx.y], blockIdx.x);
}

static bool lin_dims_ok(int n_out, int K) { return n_out > 0 && K > 0 && n_out % 128 == 0 && K % 16 == 0; }

extern "C" size_t acmil_linear_packed_bytes(int n_out, int K) {
    if (!lin_dims_ok(n_out, K)) return 0;
    const LinPlan P = lin_plan(n_out);
    return (size_t)(K / 16) * GA_FRAG_ROW * ((size_t)2 * P.nd * P.nmain + 2 * P.nd_rem);
}

// internal: n <= 8 packs in one launch; every job's shape must satisfy acmil_linear_packed_bytes != 0
int lin_pack_multi(const float* const* W, const int* ldw, const int* n_out, const int* K, void* const* packed, int n, hipStream_t st) {
    if (n <= 0 || n > LIN_PACK_MAX_JOBS) return ACMIL_ERR_SHAPE;
    LinPackMulti m;
    size_t maxrows = 0;
    for (int j = 0; j < LIN_PACK_MAX_JOBS; ++j) {
        const int q = j < n ? j : 0;
        if (!lin_dims_ok(n_out[q], K[q]) || ldw[q] < K[q]) return ACMIL_ERR_SHAPE;
        if (!W[q] || !packed[q]) return ACMIL_ERR_NULL;
        m.job[j] = LinPackArgs{W[q], (char*)packed[q], ldw[q], n_out[q], K[q]};
        const size_t rows = acmil_linear_packed_bytes(n_out[q], K[q]) / GA_FRAG_ROW;
        if (rows > maxrows) maxrows = rows;
    }
    hipLaunchKernelGGL(lin_pack_multi_kernel, dim3((unsigned)((maxrows + 3) / 4), (unsigned)n), dim3(256), 0, st, m);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_linear_pack(const float* W, int ldw, int n_out, int K, void* packed, void* stream) {
    if (!lin_dims_ok(n_out, K) || ldw < K) return ACMIL_ERR_SHAPE;
    if (!W || !packed) return ACMIL_ERR_NULL;
    LinPackArgs a = {W, (char*)packed, ldw, n_out, K};
    const size_t rows = acmil_linear_packed_bytes(n_out, K) / GA_FRAG_ROW;
    hipLaunchKernelGGL(lin_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

template <int ND, int XDT>
static int lin_launch(const LinArgs& a, hipStream_t st) {
    using G = Ga2Geom<ND, 1, XDT>;
    // per DEVICE: the dynamic-LDS attribute and the CU count (a process may drive several GPUs, or switch device after the first call)
    static int slots_of[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ACMIL_ERR_LAUNCH;
    if (slots_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ACMIL_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)lin_kernel<ND, XDT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return ACMIL_ERR_LAUNCH;
        slots_of[dev] = 2 * prop.multiProcessorCount;
    }
    const int slots = slots_of[dev];
    const long long tiles = (long long)((a.M + G::ROWS - 1) / G::ROWS) * a.nchunks;
    const dim3 grid((unsigned)(tiles < slots ? tiles : slots)), block(256);
    hipLaunchKernelGGL((lin_kernel<ND, XDT>), grid, block, G::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

template <int ND>
static int lin_launch_dt(const LinArgs& a, int x_dtype, hipStream_t st) {
    switch (x_dtype) {
        case ACMIL_DTYPE_F32: return lin_launch<ND, ACMIL_DTYPE_F32>(a, st);
        case ACMIL_DTYPE_F16: return lin_launch<ND, ACMIL_DTYPE_F16>(a, st);
        case ACMIL_DTYPE_BF16: return lin_launch<ND, ACMIL_DTYPE_BF16>(a, st);
    }
    return ACMIL_ERR_UNSUPPORTED;
}

// Control words of a call (32-bit, at `workspace`): 0 / 1 tile counters of the main / remainder launch, 2 range status, 4 / 5
// finished-workgroup counters.  init: zero them here (the C entry: any 256-byte scratch will do); !init: the caller zeroed them once
// and every launch leaves the counters at zero (TransMIL: one memset per forward instead of one per Linear layer; the status word
// then accumulates over the forward and is not looked at).
int lin_f16x3_run(const void* x, int x_dtype, int M, int K, long long ldx, const void* packed, int n_out, const float* bias, int act,
                  float beta, float* y, long long ldy, void* workspace, hipStream_t st, bool init) {
    if (M <= 0 || !lin_dims_ok(n_out, K) || ldx < K || ldy < n_out) return ACMIL_ERR_SHAPE;
    if (act != 0 && act != 1) return ACMIL_ERR_UNSUPPORTED;
    if (!x || !packed || !y || !workspace) return ACMIL_ERR_NULL;
    const int xe = (x_dtype == ACMIL_DTYPE_F32) ? 4 : 2;
    if (((size_t)x & 15) != 0 || ((size_t)ldx * xe) % 16 != 0) return ACMIL_ERR_SHAPE;     // 16-byte LDS-DMA pieces
    if (((size_t)y & 15) != 0 || ldy % 4 != 0) return ACMIL_ERR_SHAPE;                     // 16-byte row stores
    unsigned* ctr = (unsigned*)workspace;
    if (init && hipMemsetAsync(ctr, 0, 32, st) != hipSuccess) return ACMIL_ERR_LAUNCH;
    LinArgs a;
    a.x = x; a.ldx = ldx; a.M = M; a.K = K; a.bias = bias; a.act = act; a.beta = beta; a.y = y; a.ldy = ldy;
    a.ww = nullptr; a.bw = nullptr; a.scores = nullptr; a.kb = 0;
    a.status = ctr + 2;          // workspace word 2: range status of this call (zeroed by the memset above)
    const LinPlan P = lin_plan(n_out);
    int rc = ACMIL_OK;
    if (P.nmain > 0) {
        a.packed = (const char*)packed; a.nchunks = P.nmain; a.col0 = 0; a.tile_counter = ctr; a.done = ctr + 4;
        rc = P.nd == 6 ? lin_launch_dt<6>(a, x_dtype, st) : lin_launch_dt<8>(a, x_dtype, st);
        if (rc != ACMIL_OK) return rc;
    }
    if (P.nd_rem) {
        const int c0 = P.nmain * 32 * P.nd;
        a.packed = (const char*)packed + (size_t)P.nmain * (K / 16) * 2 * P.nd * GA_FRAG_ROW; a.nchunks = 1; a.col0 = c0;
        a.bias = bias ? bias + c0 : nullptr; a.tile_counter = ctr + 1; a.done = ctr + 5;
        rc = lin_launch_dt<4>(a, x_dtype, st);
    }
    return rc;
}

extern "C" int acmil_linear_f16x3(const void* x, int x_dtype, int M, int K, long long ldx, const void* packed, int n_out,
                                  const float* bias, int act, float beta, float* y, long long ldy, void* workspace, void* stream) {
    return lin_f16x3_run(x, x_dtype, M, K, ldx, packed, n_out, bias, act, beta, y, ldy, workspace, (hipStream_t)stream, true);
}

// Gated-attention scores of an already projected bag h [N, L] in ONE pass over h (Attention_Gated.forward,
// architecture/transformer.py:259-267; attention width 128): the [Wv; Wu] product runs as the packed-weight Linear kernel above with
// the gate formed in the accumulators -- h is read once, the [N, 256] pre-activations never exist in memory.
// packed_vu = acmil_linear_pack of the [256, L] matrix whose rows are [Wv 0..31; Wu 0..31; Wv 32..63; Wu 32..63; ...], bias_vu [256]
// in the same order.  Replaces the two GEMMs + gate pass of acmil_gated_scores at the widths it covers (K <= 16, L % 16 == 0).
extern "C" int acmil_gated_scores_packed(const void* h, int h_dtype, int N, int L, long long ldh, const void* packed_vu,
                                         const float* bias_vu, const float* Ww, const float* bw, int K, float* A, void* workspace,
                                         void* stream) {
    if (N <= 0 || L <= 0 || L % 16 != 0 || ldh < L || K <= 0) return ACMIL_ERR_SHAPE;
    if (K > ACMIL_MAX_TOKENS) return ACMIL_ERR_UNSUPPORTED;
    if (!h || !packed_vu || !bias_vu || !Ww || !bw || !A || !workspace) return ACMIL_ERR_NULL;
    const int xe = (h_dtype == ACMIL_DTYPE_F32) ? 4 : 2;
    if (((size_t)h & 15) != 0 || ((size_t)ldh * xe) % 16 != 0 || ((size_t)bias_vu & 15) != 0 || ((size_t)Ww & 15) != 0) return ACMIL_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    unsigned* ctr = (unsigned*)workspace;
    if (hipMemsetAsync(ctr, 0, 8, st) != hipSuccess) return ACMIL_ERR_LAUNCH;
    LinArgs a;
    a.x = h; a.ldx = ldh; a.M = N; a.K = L; a.bias = bias_vu; a.act = 2; a.beta = 0.0f; a.y = nullptr; a.ldy = 0;
    a.packed = (const char*)packed_vu; a.nchunks = 1; a.col0 = 0; a.tile_counter = ctr; a.done = nullptr;
    a.ww = Ww; a.bw = bw; a.scores = A; a.kb = K; a.status = nullptr;
    return lin_launch_dt<8>(a, h_dtype, st);
}
