// linear.hip -- host side of the packed-weight Linear kernel (linear_kernel.h): weight packing and the C entry points.
#include "linear64_kernel.h"

// Packed stream of W [n_out, K] (row-major, leading dimension ldw) as output-column chunks (lin_plan): n_out / 256 chunks of ND = 8
// (256 columns) then one ND = 4 chunk when n_out % 256 == 128 -- except n_out = 384, which runs as 2 chunks of ND = 6
// (192 columns: no half-rate remainder launch; the TransMIL width and CLIP-L's D_inner: 272 vs 323 us at M = 100 000, K = 768).  Chunk = K/16 steps; step = ND "hi"
// fragment rows then ND "lo" rows; fragment row (step s, tile d): lane (i = lane & 31, hi = lane >> 5) holds the 8 f16 halves of
// W[col0 + 32 d + i][16 s + 8 hi .. + 7].
struct LinPackArgs { const float* W; char* out; int ldw, n_out, K; const float* colscale; int wide8; };   // colscale [K] or null: pack W[c][k] * colscale[k]
struct LinPlan { int nd, nmain, nd_rem; };      // nmain chunks of 32 * nd columns, then one chunk of 32 * nd_rem columns (0 = none)
// wide8: the 8-wave geometry (256-row tiles) needs whole fragment rows per wave, i.e. ND = 8 / 4 chunks
__host__ __device__ static inline LinPlan lin_plan(int n_out, bool wide8 = false) {
    if (!wide8 && n_out % 192 == 0 && n_out % 256 != 0) return LinPlan{6, n_out / 192, 0};      // 384 (TransMIL width), 1152 (its to_qkv), 576, ...
    return LinPlan{8, n_out / 256, (n_out % 256) ? 4 : 0};
}
// A/B builds: ACMIL_LIN_WAVES=8 runs every packed Linear launch on the 8-wave geometry (pack and launch must agree: read once)
static bool lin_wide8() { static const bool v = [] { const char* e = ACMIL_AB_ENV("ACMIL_LIN_WAVES"); return e && atoi(e) == 8; }(); return v; }

__device__ __forceinline__ void lin_pack_rows(const LinPackArgs& a, unsigned block) {
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const size_t row = (size_t)block * 4 + (threadIdx.x >> 6);
    const int S1 = a.K / 16;
    const LinPlan P = lin_plan(a.n_out, a.wide8 != 0);
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
        const float w = a.colscale ? src[j] * a.colscale[16 * s + 8 * hi + j] : src[j];
        const _Float16 h = (_Float16)w;
        v[j] = part ? (_Float16)(w - (float)h) : h;
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

// K >= 32: the LDS-DMA ring runs 2 K steps ahead and crosses tile boundaries on the assumption that the next tile HAS a step s + 2 - S1;
// with a single K step (K = 16) it fetched one step past the end of the weight stream and of the last rows of x (an out-of-bounds read
// that only faulted when the allocation ended at a page boundary -- round 4's first GPU run) and handed the wrong slot to a workgroup's
// second tile.  Such shapes are refused here; callers use the generic split GEMM.
static bool lin_dims_ok(int n_out, int K) { return n_out > 0 && K >= 32 && n_out % 128 == 0 && K % 16 == 0; }

extern "C" size_t acmil_linear_packed_bytes(int n_out, int K) {
    if (!lin_dims_ok(n_out, K)) return 0;
    const LinPlan P = lin_plan(n_out, lin_wide8());
    return (size_t)(K / 16) * GA_FRAG_ROW * ((size_t)2 * P.nd * P.nmain + 2 * P.nd_rem);
}

// internal: n <= 8 packs in one launch; every job's shape must satisfy acmil_linear_packed_bytes != 0
int lin_pack_multi(const float* const* W, const int* ldw, const int* n_out, const int* K, void* const* packed, int n, hipStream_t st,
                   const float* const* colscale) {
    if (n <= 0 || n > LIN_PACK_MAX_JOBS) return ACMIL_ERR_SHAPE;
    LinPackMulti m;
    size_t maxrows = 0;
    for (int j = 0; j < LIN_PACK_MAX_JOBS; ++j) {
        const int q = j < n ? j : 0;
        if (!lin_dims_ok(n_out[q], K[q]) || ldw[q] < K[q]) return ACMIL_ERR_SHAPE;
        if (!W[q] || !packed[q]) return ACMIL_ERR_NULL;
        m.job[j] = LinPackArgs{W[q], (char*)packed[q], ldw[q], n_out[q], K[q], colscale ? colscale[q] : nullptr, lin_wide8() ? 1 : 0};
        const size_t rows = acmil_linear_packed_bytes(n_out[q], K[q]) / GA_FRAG_ROW;
        if (rows > maxrows) maxrows = rows;
    }
    hipLaunchKernelGGL(lin_pack_multi_kernel, dim3((unsigned)((maxrows + 3) / 4), (unsigned)n), dim3(256), 0, st, m);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_linear_pack(const float* W, int ldw, int n_out, int K, void* packed, void* stream) {
    if (!lin_dims_ok(n_out, K) || ldw < K) return ACMIL_ERR_SHAPE;
    if (!W || !packed) return ACMIL_ERR_NULL;
    LinPackArgs a = {W, (char*)packed, ldw, n_out, K, nullptr, lin_wide8() ? 1 : 0};
    const size_t rows = acmil_linear_packed_bytes(n_out, K) / GA_FRAG_ROW;
    hipLaunchKernelGGL(lin_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

template <int ND, int XDT, int FX = 0, int WV = 4>
static int lin_launch(const LinArgs& a, hipStream_t st) {
    using G = Ga2Geom<ND, 1, XDT, WV>;
    // per DEVICE: the dynamic-LDS attribute and the CU count (a process may drive several GPUs, or switch device after the first call)
    static int slots_of[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ACMIL_ERR_LAUNCH;
    if (slots_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ACMIL_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)lin_kernel<ND, XDT, FX, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return ACMIL_ERR_LAUNCH;
        slots_of[dev] = G::WGS * prop.multiProcessorCount;
    }
    const int slots = slots_of[dev];
    const long long tiles = (long long)((a.M + G::ROWS - 1) / G::ROWS) * a.nchunks;
    const dim3 grid((unsigned)(tiles < slots ? tiles : slots)), block(64 * WV);
    hipLaunchKernelGGL((lin_kernel<ND, XDT, FX, WV>), grid, block, G::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// lin64_kernel (linear64_kernel.h): one 512-register wave per SIMD, 64 rows per wave, one workgroup per CU
template <int ND, int XDT, int FX = 0>
static int lin_launch64(const LinArgs& a, hipStream_t st) {
    using G = Lin64Geom<ND, XDT>;
    // the kernel addresses x rows by 32-bit byte offsets from the matrix base (lin_use64 keeps M <= 65 536; a forced A/B run on a
    // 16 x 50 000-row group of 1536-wide fp32 rows read wrong rows beyond 4 GiB): refuse instead
    if ((unsigned long long)a.M * (unsigned long long)a.ldx * (XDT == ACMIL_DTYPE_F32 ? 4 : 2) > 0xffffffffull) return ACMIL_ERR_UNSUPPORTED;
    static int slots_of[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ACMIL_ERR_LAUNCH;
    if (slots_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return ACMIL_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)lin64_kernel<ND, XDT, FX>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return ACMIL_ERR_LAUNCH;
        slots_of[dev] = prop.multiProcessorCount;
    }
    const int slots = slots_of[dev];
    const long long tiles = (long long)((a.M + G::ROWS - 1) / G::ROWS) * a.nchunks;
    const dim3 grid((unsigned)(tiles < slots ? tiles : slots)), block(256);
    hipLaunchKernelGGL((lin64_kernel<ND, XDT, FX>), grid, block, G::LDS, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
// which kernel: lin64 needs four K steps per unrolled round; it is the default where it measured faster (profiles/r04_lin64_ablation.md:
// long K loops, K >= 768, with at least one 256-row tile per CU -- the wide GA projections, 3 - 7 %); with K = 384 its epilogue, which no
// second workgroup hides, costs what the leaner K loop gains.  ACMIL_LIN64=0 / 1 forces.
static bool lin_use64(int M, int K, int nchunks, int nd) {
    static const char* e = ACMIL_AB_ENV("ACMIL_LIN64");
    if (K % 64 != 0) return false;
    // lin64_kernel keeps the bias of ALL columns of a launch in an LDS table of BIAS_MAX floats (its epilogue and the landmark sums read
    // it up to column 32 nd nchunks): wider launches -- TransMIL's to_qkv at D_inner = 768 is 2 304 columns -- stay on lin_kernel, which
    // reads the bias from global memory (columns 2048.. read past the table and the folded LayerNorm bias W beta was wrong)
    if (32 * nd * nchunks > Lin64Geom<8, ACMIL_DTYPE_F32>::BIAS_MAX) return false;
    if (e) return e[0] == '1';
    // (M <= 65536: TransMIL's fc1 -- 100 000 rows, K = 768, 2 x 192 columns -- measured 223 vs 211 us with lin64, the same product on
    //  50 000 rows 103 - 109 vs 115 us)
    return K >= 768 && M <= 65536 && (long long)((M + 255) / 256) * nchunks >= 256;
}

template <int ND>
static int lin_launch_dt(const LinArgs& a, int x_dtype, hipStream_t st) {
    if (a.act != 2 && !lin_wide8() && lin_use64(a.M, a.K, a.nchunks, ND)) {
        switch (x_dtype) {
            case ACMIL_DTYPE_F32: return lin_launch64<ND, ACMIL_DTYPE_F32>(a, st);
            case ACMIL_DTYPE_F16: return lin_launch64<ND, ACMIL_DTYPE_F16>(a, st);
            case ACMIL_DTYPE_BF16: return lin_launch64<ND, ACMIL_DTYPE_BF16>(a, st);
        }
        return ACMIL_ERR_UNSUPPORTED;
    }
    if constexpr (ND == 8 || ND == 4) {
        if (lin_wide8() && a.act != 2 && x_dtype == ACMIL_DTYPE_F32) return lin_launch<ND, ACMIL_DTYPE_F32, 0, 8>(a, st);
    }
    switch (x_dtype) {
        case ACMIL_DTYPE_F32: return lin_launch<ND, ACMIL_DTYPE_F32>(a, st);
        case ACMIL_DTYPE_F16: return lin_launch<ND, ACMIL_DTYPE_F16>(a, st);
        case ACMIL_DTYPE_BF16: return lin_launch<ND, ACMIL_DTYPE_BF16>(a, st);
    }
    return ACMIL_ERR_UNSUPPORTED;
}

// y stores past the L2 for outputs that exceed the 256 MB Infinity Cache (and are not read back as the residual): see the epilogue
static int lin_nt_store(int M, int n_out, float beta) { return beta == 0.0f && (size_t)M * n_out * 4 > ((size_t)256 << 20); }

// Control words of a call (32-bit, at `workspace`): 2 range status, 4 / 5 finished-workgroup counters of the main / remainder launch,
// 8..15 / 16..23 their per-XCD tile queues (LIN_CTRL_BYTES = 96 bytes in all).  init: zero them here (the C entry: any 256-byte scratch will do); !init: the caller zeroed them once
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
    if (init && hipMemsetAsync(ctr, 0, LIN_CTRL_BYTES, st) != hipSuccess) return ACMIL_ERR_LAUNCH;
    LinArgs a;
    a.x = x; a.ldx = ldx; a.M = M; a.K = K; a.bias = bias; a.act = act; a.beta = beta; a.y = y; a.ldy = ldy;
    a.ww = nullptr; a.bw = nullptr; a.scores = nullptr; a.kb = 0;
    a.rowab = nullptr; a.wsum = nullptr; a.zrows = 0; a.lm_part = nullptr; a.lm_l = 0; a.lm_cols = 0;
    a.nt_store = lin_nt_store(M, n_out, beta);
    a.status = ctr + 2;          // workspace word 2: range status of this call (zeroed by the memset above)
    const LinPlan P = lin_plan(n_out, lin_wide8());
    int rc = ACMIL_OK;
    if (P.nmain > 0) {
        a.packed = (const char*)packed; a.nchunks = P.nmain; a.col0 = 0; a.tile_counter = ctr + 8; a.done = ctr + 4;
        rc = P.nd == 6 ? lin_launch_dt<6>(a, x_dtype, st) : lin_launch_dt<8>(a, x_dtype, st);
        if (rc != ACMIL_OK) return rc;
    }
    if (P.nd_rem) {
        const int c0 = P.nmain * 32 * P.nd;
        a.packed = (const char*)packed + (size_t)P.nmain * (K / 16) * 2 * P.nd * GA_FRAG_ROW; a.nchunks = 1; a.col0 = c0;
        a.bias = bias ? bias + c0 : nullptr; a.tile_counter = ctr + 16; a.done = ctr + 5;
        rc = lin_launch_dt<4>(a, x_dtype, st);
    }
    return rc;
}

// TransLayer's to_qkv with the LayerNorm folded in (transMIL.py:25-28, nystrom_attention.py:80,95-111), internal to the TransMIL forward:
//   y[r] = ((x[r] - mean_r) * rstd_r) (W o gamma)^T + W beta      for rows r >= zrows,   y[r] = 0 for the zero padding rows r < zrows
// rowab [M][2] = (rstd, -mean * rstd) per row (0, 0 for r < zrows) from tm_rowstats_kernel; `packed` = the stream of W o gamma
// (lin_pack_multi with colscale = gamma); bias = W beta.  lm_part (or null): per-wave-tile column sums of the first lm_cols output
// columns, split at the landmark boundary (see LinArgs).  Control words as lin_f16x3_run with init = false.
// Round 6: rowab == null and wsum [n_out] given (wsum[c] = sum_k (W o gamma)[c][k]): the row statistics are formed inside the launch
// (LinArgs::wsum, FX & 4) -- no statistics pass over x at all.  lin_qkv_norm_instat_ok tells whether this launch plan has that form
// (the 4-wave lin_kernel; the 64-row / 8-wave A/B geometries keep the statistics pass).
bool lin_qkv_norm_instat_ok(int M, int K, int n_out) {
    static const bool off = ACMIL_AB_ENV("ACMIL_TM_ROWSTATS") != nullptr;      // A/B knob: keep tm_rowstats_kernel
    if (off || lin_wide8() || !lin_dims_ok(n_out, K)) return false;
    const LinPlan P = lin_plan(n_out, false);
    return !(P.nmain > 0 && lin_use64(M, K, P.nmain, P.nd));
}

// c_lo .. c_hi (round 6; in-launch statistics only): the launch forms only these output columns -- whole chunks of the launch plan
// (lin_qkv_norm_cols_ok) -- so that TransMIL can run [q | k] first and v beside the Moore-Penrose chain that only needs q and k.
bool lin_qkv_norm_cols_ok(int M, int K, int n_out, int c_lo, int c_hi) {
    if (!lin_qkv_norm_instat_ok(M, K, n_out)) return false;
    const LinPlan P = lin_plan(n_out, false);
    const int cw = 32 * P.nd;
    return P.nd_rem == 0 && c_lo >= 0 && c_lo < c_hi && c_hi <= n_out && c_lo % cw == 0 && c_hi % cw == 0;
}

int lin_qkv_norm_run(const float* x, int M, int K, long long ldx, const float* rowab, int zrows, const void* packed, int n_out,
                     const float* bias, float* y, long long ldy, float* lm_part, int lm_l, int lm_cols, void* workspace, hipStream_t st,
                     const float* wsum, int c_lo, int c_hi) {
    if (M <= 0 || !lin_dims_ok(n_out, K) || ldx < K || ldy < n_out || zrows < 0) return ACMIL_ERR_SHAPE;
    const bool cols = !(c_lo == 0 && c_hi == n_out);
    if (cols && (rowab != nullptr || !lin_qkv_norm_cols_ok(M, K, n_out, c_lo, c_hi))) return ACMIL_ERR_UNSUPPORTED;
    if (!x || (!rowab && !wsum) || !packed || !bias || !y || !workspace) return ACMIL_ERR_NULL;
    const bool instat = rowab == nullptr;
    if (instat && (!lin_qkv_norm_instat_ok(M, K, n_out) || ((size_t)wsum & 15) != 0)) return ACMIL_ERR_UNSUPPORTED;
    if (((size_t)x & 15) != 0 || ((size_t)ldx * 4) % 16 != 0 || ((size_t)y & 15) != 0 || ldy % 4 != 0 || ((size_t)rowab & 7) != 0) return ACMIL_ERR_SHAPE;
    if (lm_part && (lm_l < 32 || lm_cols % 32 != 0 || lm_cols > n_out)) return ACMIL_ERR_SHAPE;
    unsigned* ctr = (unsigned*)workspace;
    LinArgs a;
    a.x = x; a.ldx = ldx; a.M = M; a.K = K; a.bias = bias; a.act = 0; a.beta = 0.0f; a.y = y; a.ldy = ldy;
    a.ww = nullptr; a.bw = nullptr; a.scores = nullptr; a.kb = 0; a.status = nullptr;
    a.rowab = rowab; a.wsum = instat ? wsum : nullptr; a.zrows = zrows; a.lm_part = lm_part; a.lm_l = lm_l; a.lm_cols = lm_part ? lm_cols : 0;
    a.nt_store = lin_nt_store(M, n_out, 0.0f);
    const LinPlan P = lin_plan(n_out, lin_wide8());
    int rc = ACMIL_OK;
    if (instat) {
        if (cols) {
            const int cw = 32 * P.nd;
            a.packed = (const char*)packed + (size_t)(c_lo / cw) * (K / 16) * 2 * P.nd * GA_FRAG_ROW; a.nchunks = (c_hi - c_lo) / cw; a.col0 = c_lo;
            a.bias = bias + c_lo; a.wsum = wsum + c_lo; a.tile_counter = ctr + 8; a.done = ctr + 4;
            if (c_lo >= lm_cols) { a.lm_part = nullptr; a.lm_cols = 0; }
            return P.nd == 6 ? lin_launch<6, ACMIL_DTYPE_F32, 7>(a, st) : lin_launch<8, ACMIL_DTYPE_F32, 7>(a, st);
        }
        if (P.nmain > 0) {
            a.packed = (const char*)packed; a.nchunks = P.nmain; a.col0 = 0; a.tile_counter = ctr + 8; a.done = ctr + 4;
            rc = P.nd == 6 ? lin_launch<6, ACMIL_DTYPE_F32, 7>(a, st) : lin_launch<8, ACMIL_DTYPE_F32, 7>(a, st);
            if (rc != ACMIL_OK) return rc;
        }
        if (P.nd_rem) {
            const int c0 = P.nmain * 32 * P.nd;
            a.packed = (const char*)packed + (size_t)P.nmain * (K / 16) * 2 * P.nd * GA_FRAG_ROW; a.nchunks = 1; a.col0 = c0;
            a.bias = bias + c0; a.wsum = wsum + c0; a.tile_counter = ctr + 16; a.done = ctr + 5;
            rc = lin_launch<4, ACMIL_DTYPE_F32, 7>(a, st);
        }
        return rc;
    }
    if (P.nmain > 0) {
        a.packed = (const char*)packed; a.nchunks = P.nmain; a.col0 = 0; a.tile_counter = ctr + 8; a.done = ctr + 4;
        if (lin_wide8()) rc = lin_launch<8, ACMIL_DTYPE_F32, 3, 8>(a, st);
        else if (lin_use64(M, K, P.nmain, P.nd)) rc = P.nd == 6 ? lin_launch64<6, ACMIL_DTYPE_F32, 3>(a, st) : lin_launch64<8, ACMIL_DTYPE_F32, 3>(a, st);
        else rc = P.nd == 6 ? lin_launch<6, ACMIL_DTYPE_F32, 3>(a, st) : lin_launch<8, ACMIL_DTYPE_F32, 3>(a, st);
        if (rc != ACMIL_OK) return rc;
    }
    if (P.nd_rem) {
        const int c0 = P.nmain * 32 * P.nd;
        a.packed = (const char*)packed + (size_t)P.nmain * (K / 16) * 2 * P.nd * GA_FRAG_ROW; a.nchunks = 1; a.col0 = c0;
        a.bias = bias + c0; a.tile_counter = ctr + 16; a.done = ctr + 5;
        rc = lin_wide8() ? lin_launch<4, ACMIL_DTYPE_F32, 3, 8>(a, st) : lin_launch<4, ACMIL_DTYPE_F32, 3>(a, st);
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
    if (K > ACMIL_MAX_TOKENS || L < 32) return ACMIL_ERR_UNSUPPORTED;      // (L >= 32: two K steps, see lin_dims_ok)
    if (!h || !packed_vu || !bias_vu || !Ww || !bw || !A || !workspace) return ACMIL_ERR_NULL;
    const int xe = (h_dtype == ACMIL_DTYPE_F32) ? 4 : 2;
    if (((size_t)h & 15) != 0 || ((size_t)ldh * xe) % 16 != 0 || ((size_t)bias_vu & 15) != 0 || ((size_t)Ww & 15) != 0) return ACMIL_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    unsigned* ctr = (unsigned*)workspace;
    if (hipMemsetAsync(ctr, 0, LIN_CTRL_BYTES, st) != hipSuccess) return ACMIL_ERR_LAUNCH;
    LinArgs a;
    a.rowab = nullptr; a.wsum = nullptr; a.zrows = 0; a.lm_part = nullptr; a.lm_l = 0; a.lm_cols = 0; a.nt_store = 0;
    a.x = h; a.ldx = ldh; a.M = N; a.K = L; a.bias = bias_vu; a.act = 2; a.beta = 0.0f; a.y = nullptr; a.ldy = 0;
    a.packed = (const char*)packed_vu; a.nchunks = 1; a.col0 = 0; a.tile_counter = ctr + 8; a.done = nullptr;
    a.ww = Ww; a.bw = bw; a.scores = A; a.kb = K; a.status = nullptr;
    return lin_launch_dt<8>(a, h_dtype, st);
}
