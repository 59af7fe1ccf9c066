// ga_pack.hip -- rearrange ACMIL_GA / ABMIL parameters into the MFMA-fragment-ordered stream that
// ga_forward.hip consumes (layout: ga_common.h).  One launch per parameter update.
//
// Fragment order is chosen so that (a) a workgroup stages the stream into LDS with plain linear
// global_load_lds copies, (b) every wave reads its MFMA A-operand with lane-linear, conflict-free
// ds_read_b128, and (c) the K-slot permutation of GEMM2 matches the C/D register layout GEMM1 leaves its
// result in, so relu(h) feeds the second MFMA chain straight from registers.
#include "ga_common.h"

struct GaPackArgs {
    const float *W1, *Wv, *bv, *Wu, *bu, *Ww, *bw, *Ws, *bs;
    const float* Wc[ACMIL_MAX_TOKENS];
    const float* bc[ACMIL_MAX_TOKENS];
    char* out;
    GaLayout L;
};

__device__ static inline void split_f16(float w, _Float16& hi, _Float16& lo) {
    hi = (_Float16)w;
    lo = (_Float16)(w - (float)hi);
}

// Row of [Wv;Wu] addressed by (a-tile at in 0..7, row i in 0..31): tiles alternate v,u so that the
// accumulators of tiles 2p and 2p+1 hold tanh- and sigmoid-branch pre-activations of the SAME 32 units.
// GEMM2 runs in two unit-groups g (tiles 4g..4g+3 = units 64g..64g+63) to halve its accumulator footprint.
__device__ static inline const float* vu_row(const GaPackArgs& a, int at, int i) {
    const int unit = 32 * (at >> 1) + i;
    return ((at & 1) ? a.Wu : a.Wv) + (size_t)unit * a.L.Di;
}

__global__ __launch_bounds__(256) void ga_pack_kernel(GaPackArgs a) {
    const GaLayout& L = a.L;
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, hi = lane >> 5;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ND = L.ND;
    if (row < L.g1_rows) {
        char* dst = a.out + L.g1_off + row * GA_FRAG_ROW + lane * 16;
        if (L.mode == ACMIL_MODE_F32) {
            // row = (t*8 + c4)*ND + d ; lane holds W1[32d+i][64t + 32hi + 4c4 + q], q<4
            const int d = row % ND, c4 = (row / ND) % 8, t = row / ND / 8;
            const float* src = a.W1 + (size_t)(32 * d + i) * L.D + 64 * t + 32 * hi + 4 * c4;
            *(f32x4*)dst = *(const f32x4*)src;
        } else {
            // F16X3: row = ((t*4 + s)*ND + d)*2 + part ; F16: row = (t*4 + s)*ND + d
            // lane holds 8 f16: W1[32d+i][64t + 32hi + 8s + j], j<8
            size_t r = row; int part = 0;
            if (L.mode == ACMIL_MODE_F16X3) { part = r & 1; r >>= 1; }
            const int d = r % ND, s = (r / ND) % 4, t = r / ND / 4;
            const float* src = a.W1 + (size_t)(32 * d + i) * L.D + 64 * t + 32 * hi + 8 * s;
            f16x8 v;
            for (int j = 0; j < 8; ++j) { _Float16 h, l; split_f16(src[j], h, l); v[j] = part ? l : h; }
            *(f16x8*)dst = v;
        }
        return;
    }
    size_t r2 = row - L.g1_rows;
    if (r2 < L.g2_rows) {
        char* dst = a.out + L.g2_off + r2 * GA_FRAG_ROW + lane * 16;
        if (L.mode == ACMIL_MODE_F32) {
            // row = ((g*ND + d)*4 + r4)*4 + al, at = 4g + al ; lane holds Wvu[at,i][32d + 8r4 + 4hi + q]
            const int al = r2 % 4, r4 = (r2 / 4) % 4, d = (r2 / 16) % ND, g = r2 / 16 / ND;
            const int at = 4 * g + al;
            const float* src = vu_row(a, at, i) + 32 * d + 8 * r4 + 4 * hi;
            *(f32x4*)dst = *(const f32x4*)src;
        } else {
            // F16X3: row = (((g*ND + d)*2 + e)*4 + al)*2 + part ; F16: row = ((g*ND + d)*2 + e)*4 + al ; at = 4g + al
            // slot j <-> GEMM1 accumulator register 8e+j of tile d: di = 32d + (j&3) + 8(2e + (j>>2)) + 4hi
            size_t r = r2; int part = 0;
            if (L.mode == ACMIL_MODE_F16X3) { part = r & 1; r >>= 1; }
            const int al = r % 4, e = (r / 4) % 2, d = (r / 8) % ND, g = r / 8 / ND;
            const int at = 4 * g + al;
            const float* src = vu_row(a, at, i);
            f16x8 v;
            for (int j = 0; j < 8; ++j) {
                const int di = 32 * d + (j & 3) + 8 * (2 * e + (j >> 2)) + 4 * hi;
                _Float16 h, l; split_f16(src[di], h, l); v[j] = part ? l : h;
            }
            *(f16x8*)dst = v;
        }
        return;
    }
    // ---- tail: epilogue table, biases, classifier heads (handled by the first workgroup past the streams)
    const size_t tail_block = (L.g1_rows + L.g2_rows + 3) / 4;
    if (blockIdx.x != tail_block) return;
    const int tid = threadIdx.x;
    float* tab = (float*)(a.out + L.tab_off);
    for (int e = tid; e < GA_DA; e += 256) {
        tab[e] = a.bv[e];
        tab[GA_DA + e] = a.bu[e];
        for (int k = 0; k < L.K; ++k) tab[(2 + k) * GA_DA + e] = a.Ww[(size_t)k * GA_DA + e];
    }
    float* bw = (float*)(a.out + L.bw_off);
    if (tid < 8) bw[tid] = (tid < L.K) ? a.bw[tid] : 0.0f;
    float* wc = (float*)(a.out + L.wc_off);
    float* bc = (float*)(a.out + L.bc_off);
    const int CD = L.C * L.Di;
    for (int k = 0; k < L.K; ++k) {
        for (int e = tid; e < CD; e += 256) wc[(size_t)k * CD + e] = a.Wc[k][e];
        if (tid < L.C) bc[k * L.C + tid] = a.bc[k][tid];
    }
    float* ws = (float*)(a.out + L.ws_off);
    float* bs = (float*)(a.out + L.bs_off);
    for (int e = tid; e < CD; e += 256) ws[e] = a.Ws ? a.Ws[e] : 0.0f;
    if (tid < L.C) bs[tid] = a.bs ? a.bs[tid] : 0.0f;
}

extern "C" size_t acmil_ga_packed_bytes(int D, int Di, int Da, int K, int C, int mode) {
    if (ga_check_dims(D, Di, Da, K, C) != ACMIL_OK) return 0;
    return ga_layout(D, Di, K, C, mode).total;
}

extern "C" int acmil_ga_pack_weights(const float* W1, const float* Wv, const float* bv, const float* Wu,
                                     const float* bu, const float* Ww, const float* bw, const float* const* Wc,
                                     const float* const* bc, const float* Ws, const float* bs, int D, int Di, int Da,
                                     int K, int C, int mode, void* packed, void* stream) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (mode != ACMIL_MODE_F32 && mode != ACMIL_MODE_F16X3 && mode != ACMIL_MODE_F16) return ACMIL_ERR_UNSUPPORTED;
    if (!W1 || !Wv || !bv || !Wu || !bu || !Ww || !bw || !Wc || !bc || !packed) return ACMIL_ERR_NULL;
    if ((Ws == nullptr) != (bs == nullptr)) return ACMIL_ERR_NULL;
    GaPackArgs a;
    a.W1 = W1; a.Wv = Wv; a.bv = bv; a.Wu = Wu; a.bu = bu; a.Ww = Ww; a.bw = bw; a.Ws = Ws; a.bs = bs;
    for (int k = 0; k < ACMIL_MAX_TOKENS; ++k) {
        a.Wc[k] = k < K ? Wc[k] : nullptr;
        a.bc[k] = k < K ? bc[k] : nullptr;
        if (k < K && (!a.Wc[k] || !a.bc[k])) return ACMIL_ERR_NULL;
    }
    a.out = (char*)packed;
    a.L = ga_layout(D, Di, K, C, mode);
    const unsigned blocks = (unsigned)((a.L.g1_rows + a.L.g2_rows + 3) / 4) + 1;
    hipLaunchKernelGGL(ga_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
