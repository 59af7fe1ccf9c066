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
// GEMM2 runs in four unit blocks g4 (tiles 2*g4, 2*g4+1 = units 32*g4..32*g4+31): 32 accumulator registers live.
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
        // GEMM1 step s covers k = 16s .. 16s+15; lane (i, hi) owns the 8 K-slots k = 16s + 8hi + j, j < 8
        if (L.mode == ACMIL_MODE_F32) {
            // row = (s*ND + d)*2 + half ; lane holds W1[32d+i][16s + 8hi + 4half + q], q<4 (MFMA sub-step c = 4half+q)
            const int half = row & 1, d = (row >> 1) % ND, st = (row >> 1) / ND;
            const float* src = a.W1 + (size_t)(32 * d + i) * L.D + 16 * st + 8 * hi + 4 * half;
            *(f32x4*)dst = *(const f32x4*)src;
        } else {
            // F16X3: row = (s*2 + part)*ND + d (all "hi" fragments of a step, then all "lo") ; F16: row = s*ND + d ;
            // lane holds 8 f16 W1[32d+i][16s + 8hi + j]
            size_t r = row; int part = 0;
            const int d = r % ND; r /= ND;
            if (L.mode == ACMIL_MODE_F16X3) { part = r & 1; r >>= 1; }
            const int st = r;
            const float* src = a.W1 + (size_t)(32 * d + i) * L.D + 16 * st + 8 * hi;
            f16x8 v;
            for (int j = 0; j < 8; ++j) { _Float16 h, l; split_f16(src[j], h, l); v[j] = part ? l : h; }
            *(f16x8*)dst = v;
        }
        return;
    }
    size_t r2 = row - L.g1_rows;
    if (r2 < L.g2_rows) {
        char* dst = a.out + L.g2_off + r2 * GA_FRAG_ROW + lane * 16;
        // GEMM2 step j = 4*g4 + st (st < 4): unit block g4 (tiles at = 2*g4 + al, al = 0 tanh / 1 sigmoid branch),
        // h tiles d = DD*st + dd with DD = ND/4.  K-slot <-> GEMM1 accumulator register of tile d (see header comment).
        const int DD = ND / 4;
        if (L.mode == ACMIL_MODE_F32) {
            // local row = (dd*4 + r4)*2 + al (8*DD per step) ; lane holds Wvu[at,i][32d + 8r4 + 4hi + q], q<4
            const int per = 8 * DD;
            const int loc = r2 % per; const size_t j = r2 / per;
            const int al = loc & 1, r4 = (loc >> 1) & 3, dd = loc >> 3;
            const int g4 = j / 4, d = DD * (j % 4) + dd, at = 2 * g4 + al;
            const float* src = vu_row(a, at, i) + 32 * d + 8 * r4 + 4 * hi;
            *(f32x4*)dst = *(const f32x4*)src;
        } else {
            // F16X3: local row = (dd*2 + part)*4 + e*2 + al (8*DD per step: per h tile 4 "hi" rows then 4 "lo" rows) ;
            // F16: dd*4 + e*2 + al (4*DD per step)
            // slot jj <-> GEMM1 accumulator register 8e+jj of tile d: di = 32d + (jj&3) + 8(2e + (jj>>2)) + 4hi
            const bool split = L.mode == ACMIL_MODE_F16X3;
            const int per = (split ? 8 : 4) * DD;
            const int loc0 = r2 % per; const size_t j = r2 / per;
            const int t = loc0 & 3;
            const int part = split ? ((loc0 >> 2) & 1) : 0;
            const int dd = split ? (loc0 >> 3) : (loc0 >> 2);
            const int al = t & 1, e = t >> 1;
            const int g4 = j / 4, d = DD * (j % 4) + dd, at = 2 * g4 + al;
            const float* src = vu_row(a, at, i);
            f16x8 v;
            for (int jj = 0; jj < 8; ++jj) {
                const int di = 32 * d + (jj & 3) + 8 * (2 * e + (jj >> 2)) + 4 * hi;
                _Float16 h, l; split_f16(src[di], h, l); v[jj] = part ? l : h;
            }
            *(f16x8*)dst = v;
        }
        return;
    }
    // ---- past the streams: one workgroup for the small vectors, then 1024 elements per workgroup of the raw fp32 copies
    // (concatenated attention weights for the backward, classifier heads)
    const size_t tail_block = (L.g1_rows + L.g2_rows + 3) / 4;
    const int tid = threadIdx.x;
    const int CD = L.C * L.Di;
    if (blockIdx.x > tail_block) {
        const size_t ncat = (size_t)2 * GA_DA * L.Di, nwc = (size_t)L.K * CD, nT = (size_t)L.Di * GA_WT_KX;
        const size_t total = ncat + nwc + CD + ncat + ncat + nT;
        float* wcat = (float*)(a.out + L.wcat_off);
        float* wcatT = (float*)(a.out + L.wcatT_off);
        float* wc = (float*)(a.out + L.wc_off);
        float* ws = (float*)(a.out + L.ws_off);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t e = ((size_t)blockIdx.x - tail_block - 1) * 1024 + q * 256 + tid;
            if (e >= total) break;
            if (e < ncat) wcat[e] = e < ncat / 2 ? a.Wv[e] : a.Wu[e - ncat / 2];
            else if (e < ncat + nwc) { const size_t r = e - ncat; wc[r] = a.Wc[r / CD][r % CD]; }
            else if (e < ncat + nwc + CD) { const size_t r = e - ncat - nwc; ws[r] = a.Ws ? a.Ws[r] : 0.0f; }
            else if (e < ncat + nwc + CD + ncat) {      // transposed copy: element (di, u) of [Di][2 Da]
                const size_t r = e - ncat - nwc - CD;
                const int di = (int)(r / (2 * GA_DA)), u = (int)(r % (2 * GA_DA));
                wcatT[r] = u < GA_DA ? a.Wv[(size_t)u * L.Di + di] : a.Wu[(size_t)(u - GA_DA) * L.Di + di];
            } else if (e < ncat + nwc + CD + 2 * ncat) {  // f16 hi / lo of [Wv; Wu] [2 Da][Di] in fragment order
                const size_t r = e - ncat - nwc - CD - ncat;
                const float w = r < ncat / 2 ? a.Wv[r] : a.Wu[r - ncat / 2];
                _Float16 h, l; split_f16(w, h, l);
                _Float16* p16 = (_Float16*)(a.out + L.w16_off);
                const int u = (int)(r / L.Di), k = (int)(r % L.Di);
                p16[ga_frag_off(u, k, L.Di / 16, 0)] = h; p16[ga_frag_off(u, k, L.Di / 16, 1)] = l;
            } else {                                      // bf16 hi / lo planes of [[Wv;Wu]^T | (d_afeat^T: filled by the step's tail kernel) | 0] [Di][288]
                const size_t r = e - ncat - nwc - CD - 2 * ncat;
                const int di = (int)(r / GA_WT_KX), u = (int)(r % GA_WT_KX);
                const float w = u < GA_DA ? a.Wv[(size_t)u * L.Di + di] : u < 2 * GA_DA ? a.Wu[(size_t)(u - GA_DA) * L.Di + di] : 0.0f;
                const __bf16 h = (__bf16)w, l = (__bf16)(w - (float)h);
                __bf16* pT = (__bf16*)(a.out + L.wT16_off);
                pT[ga_frag_off(di, u, GA_WT_KX / 16, 0)] = h; pT[ga_frag_off(di, u, GA_WT_KX / 16, 1)] = l;
            }
        }
        return;
    }
    if (blockIdx.x != tail_block) return;
    float* tab = (float*)(a.out + L.tab_off);
    float* bcat = (float*)(a.out + L.bcat_off);
    for (int e = tid; e < GA_DA; e += 256) {
        tab[e] = a.bv[e];
        tab[GA_DA + e] = a.bu[e];
        bcat[e] = a.bv[e];
        bcat[GA_DA + e] = a.bu[e];
        for (int k = 0; k < L.K; ++k) tab[(2 + k) * GA_DA + e] = a.Ww[(size_t)k * GA_DA + e];
    }
    float* bw = (float*)(a.out + L.bw_off);
    if (tid < ACMIL_MAX_TOKENS) bw[tid] = (tid < L.K) ? a.bw[tid] : 0.0f;
    float* bc = (float*)(a.out + L.bc_off);
    for (int k = 0; k < L.K; ++k)
        if (tid < L.C) bc[k * L.C + tid] = a.bc[k][tid];
    float* bs = (float*)(a.out + L.bs_off);
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
    const size_t aux = (size_t)6 * GA_DA * Di + (size_t)K * C * Di + (size_t)C * Di + (size_t)Di * GA_WT_KX;
    const unsigned blocks = (unsigned)((a.L.g1_rows + a.L.g2_rows + 3) / 4) + 1 + (unsigned)((aux + 1023) / 1024);
    hipLaunchKernelGGL(ga_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
