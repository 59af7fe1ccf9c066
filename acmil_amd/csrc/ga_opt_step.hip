// ga_opt_step.hip -- the closing launch of a single-GPU ACMIL_GA training step: split-K finish + AdamW + weight re-pack in ONE kernel.
//
// The step of Step3_WSI_classification_ACMIL.py:200-219 (forward, loss, backward, optimizer.step()) used to end with three dependent
// launches that each move well under a megabyte of results -- gemm_finish_kernel (fixed-order sums of the weight-gradient partials
// and of the gate-pass records), adamw_flat_kernel (the optimizer) -- and to BEGIN the next step with a fourth, ga_pack_kernel
// (the updated parameters as MFMA-fragment streams and pre-split operands).  At N = 10 000 those four are 19 us of a 158 us step,
// nearly all of it launch latency.  Here the thread that owns four consecutive elements of a parameter
//   1. forms their gradient exactly as gemm_finish_kernel does (partials summed in index order; records: lanes stride the
//      records, shuffle tree) and stores it into the gradient tensor,
//   2. applies the AdamW update (optim_kernel.h: the same inline function as adamw_flat_kernel, incl. the device-side skip on the
//      step's range flag and the skipped-launch count of the bias corrections),
//   3. scatters the new values to every place ga_pack_kernel would have put them (GEMM1 / GEMM2 fragment streams as f16 hi / lo,
//      [Wv;Wu] raw, transposed, f16 and bf16 fragment planes, the epilogue tables, the head copies).
// The results -- gradients, parameters, moments, packed buffer -- are bit-identical to the three-launch sequence followed by a
// re-pack (tests/test_train_opt_gpu.py).  Data-parallel runs keep the separate launches: their all-reduce sits between 1 and 2.
#include <string.h>

#include "ga_train_internal.h"
#include "optim_kernel.h"

#define GO_MAXK ACMIL_MAX_TOKENS_FUSED

struct GoArgs {
    const float *ws_vu, *ws_w1; int splits_vu, splits_w1;     // split-K partials [split][2 Da][Di] / [split][Di][D]; splits <= 1: the gradient tensor already holds the product
    const float* rec; int records, rec_stride, KP;            // gate-pass records (dWw | dbw | dbv | dbu per record); rec == null: the
                                                              // gradient tensors are final (acmil_ga_adamw_pack: after a data-parallel all-reduce)
    float *W1, *Wv, *bv, *Wu, *bu, *Ww, *bw, *Ws, *bs; float* Wc[GO_MAXK]; float* bc[GO_MAXK];
    float *dW1, *dWv, *dbv, *dWu, *dbu, *dWw, *dbw, *dWs, *dbs; float* dWc[GO_MAXK]; float* dbc[GO_MAXK];
    long long m_off, v_off;                                   // the moments of parameter element p live at p + m_off / p + v_off
    float lr, eps, wd; double beta1, beta2; long long launch;
    const float* skip_flag; int* skipped; float* flag_report;
    char* out; GaLayout L;
    int blkA, blkB, blkC;                                     // blocks of W1, of [Wv;Wu], of the records; the rest: heads
};

typedef float go_f4 __attribute__((ext_vector_type(4)));
typedef unsigned go_u2 __attribute__((ext_vector_type(2)));

struct GoCoef { float lr, wd, inv_bc1, inv_sqrt_bc2, eps, b1, b2; };

// sum of `splits` partial tiles, four neighbouring elements at once, each component in the index order gm_reduce_body uses
__device__ __forceinline__ go_f4 go_sum4(const float* src, long long per, int splits) {
    go_f4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    int sp = 0;
    for (; sp + 16 <= splits; sp += 16) {          // 16 loads in flight, summed in index order
        go_f4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *(const go_f4*)(src + (long long)(sp + j) * per);
#pragma unroll
        for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; sp + 4 <= splits; sp += 4) {
        go_f4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *(const go_f4*)(src + (long long)(sp + j) * per);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[j];
    }
    for (; sp < splits; ++sp) s += *(const go_f4*)(src + (long long)sp * per);
    return s;
}

struct GoState4 { go_f4 p, m, v; };
__device__ __forceinline__ GoState4 go_load4(const float* p, long long m_off, long long v_off) {
    GoState4 s;
    s.p = *(const go_f4*)p; s.m = *(const go_f4*)(p + m_off); s.v = *(const go_f4*)(p + v_off);
    return s;
}
__device__ __forceinline__ void go_adamw4(float* p, const GoState4& st, go_f4 g, long long m_off, long long v_off, const GoCoef& c, go_f4& out) {
    go_f4 pv = st.p, mv = st.m, vv = st.v;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float pi = pv[t], mi = mv[t], vi = vv[t];
        adamw_update(pi, g[t], mi, vi, c.lr, c.wd, c.inv_bc1, c.inv_sqrt_bc2, c.eps, c.b1, c.b2);
        pv[t] = pi; mv[t] = mi; vv[t] = vi;
    }
    *(go_f4*)p = pv; *(go_f4*)(p + m_off) = mv; *(go_f4*)(p + v_off) = vv;
    out = pv;
}

__device__ __forceinline__ float go_adamw1(float* p, float g, long long m_off, long long v_off, const GoCoef& c) {
    float pi = *p, mi = p[m_off], vi = p[v_off];
    adamw_update(pi, g, mi, vi, c.lr, c.wd, c.inv_bc1, c.inv_sqrt_bc2, c.eps, c.b1, c.b2);
    *p = pi; p[m_off] = mi; p[v_off] = vi;
    return pi;
}

// four values -> four f16 "hi" and four f16 "lo" halves (ga_pack.hip: hi = RN(w), lo = RN(w - hi)), 8 bytes each
__device__ __forceinline__ void go_split_f16(go_f4 w, go_u2& hi, go_u2& lo) {
    _Float16 h[4], l[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { h[t] = (_Float16)w[t]; l[t] = (_Float16)(w[t] - (float)h[t]); }
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
    hi = __builtin_bit_cast(go_u2, hv); lo = __builtin_bit_cast(go_u2, lv);
}

__global__ __launch_bounds__(256) void ga_opt_step_kernel(GoArgs a) {
    __shared__ float s_bc[2];
    const GaLayout& L = a.L;
    const int tid = threadIdx.x, blk = blockIdx.x;
    const float flag = a.skip_flag ? a.skip_flag[0] : 0.0f;
    if (a.flag_report && blk == 0 && tid == 0) __builtin_nontemporal_store(flag, a.flag_report);
    const bool skip = !(flag == 0.0f);      // (a NaN flag skips as well) -- gradients are still written, parameters / moments / packed stay
    if (skip && a.skipped && blk == 0 && tid == 0) atomicAdd(a.skipped, 1);
    // the bias corrections (two double pow: ~1.5 us of one lane) are formed while the other waves already load; every region
    // below issues its global reads first and meets this lane at ONE barrier (go_coef) before the update
    if (tid == 0 && !skip) {
        const long long t = a.launch - (a.skipped ? (long long)*a.skipped : 0);
        adamw_bias(a.beta1, a.beta2, t, s_bc[0], s_bc[1]);
    }
    auto go_coef = [&]() {
        __syncthreads();
        GoCoef c;
        c.lr = a.lr; c.wd = a.wd; c.eps = a.eps; c.b1 = (float)a.beta1; c.b2 = (float)a.beta2;
        c.inv_bc1 = skip ? 0.0f : s_bc[0]; c.inv_sqrt_bc2 = skip ? 0.0f : s_bc[1];
        return c;
    };
    const int Di = L.Di, D = L.D, ND = L.ND;
    const go_f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    if (blk < a.blkA) {
        // ---- W1 [Di][D]: elements (r, k .. k+3)
        const long long per = (long long)Di * D;
        const long long e = ((long long)blk * 256 + tid) * 4;
        const bool on = e < per;
        GoState4 st = {zero4, zero4, zero4};
        go_f4 g = zero4;
        if (on) {
            if (!skip) st = go_load4(a.W1 + e, a.m_off, a.v_off);
            g = a.splits_w1 > 1 ? go_sum4(a.ws_w1 + e, per, a.splits_w1) : *(const go_f4*)(a.dW1 + e);
        }
        const GoCoef c = go_coef();
        if (!on) return;
        if (a.splits_w1 > 1) *(go_f4*)(a.dW1 + e) = g;
        if (skip) return;
        go_f4 w;
        go_adamw4(a.W1 + e, st, g, a.m_off, a.v_off, c, w);
        // GEMM1 stream (F16X3): row (st*2 + part)*ND + d, lane i + 32 hi holds W1[32d + i][16 st + 8 hi + j], j < 8
        const int r = (int)(e / D), k = (int)(e % D);
        const int d = r >> 5, i = r & 31, stp = k >> 4, hi = (k >> 3) & 1, jh = (k >> 2) & 1;
        go_u2 h2, l2;
        go_split_f16(w, h2, l2);
        char* dst = a.out + L.g1_off + (size_t)(i + 32 * hi) * 16 + jh * 8;
        *(go_u2*)(dst + ((size_t)(stp * 2 + 0) * ND + d) * GA_FRAG_ROW) = h2;
        *(go_u2*)(dst + ((size_t)(stp * 2 + 1) * ND + d) * GA_FRAG_ROW) = l2;
        return;
    }
    if (blk < a.blkA + a.blkB) {
        // ---- [Wv; Wu] [2 Da][Di]: elements (u, di .. di+3), u < Da tanh branch, else sigmoid branch
        const long long per = (long long)2 * GA_DA * Di;
        const long long e = ((long long)(blk - a.blkA) * 256 + tid) * 4;
        const bool on = e < per;
        const int u = on ? (int)(e / Di) : 0, di = on ? (int)(e % Di) : 0;
        const int al = u >= GA_DA ? 1 : 0, unit = u - al * GA_DA;
        const size_t loc = (size_t)unit * Di + di;
        float* p = (al ? a.Wu : a.Wv) + loc;
        float* gd = (al ? a.dWu : a.dWv) + loc;
        GoState4 st = {zero4, zero4, zero4};
        go_f4 g = zero4;
        if (on) {
            if (!skip) st = go_load4(p, a.m_off, a.v_off);
            g = a.splits_vu > 1 ? go_sum4(a.ws_vu + e, per, a.splits_vu) : *(const go_f4*)gd;
        }
        const GoCoef c = go_coef();
        if (!on) return;
        if (a.splits_vu > 1) *(go_f4*)gd = g;
        if (skip) return;
        go_f4 w;
        go_adamw4(p, st, g, a.m_off, a.v_off, c, w);
        go_u2 h2, l2;
        go_split_f16(w, h2, l2);
        {   // GEMM2 stream: step j = 4 g4 + st, local row (dd*2 + part)*4 + e2*2 + al; slot jj of lane (i, hi) <-> di = 32 d + (jj&3) + 8 (2 e2 + (jj>>2)) + 4 hi
            const int DD = ND / 4, per_rows = 8 * DD;
            const int g4 = unit >> 5, i = unit & 31;
            const int d = di >> 5, q = (di >> 3) & 3, hi = (di >> 2) & 1;
            const int e2 = q >> 1, jq = q & 1;
            const int stp = d / DD, dd = d % DD;
            const size_t j = (size_t)4 * g4 + stp;
            char* dst = a.out + L.g2_off + (size_t)(i + 32 * hi) * 16 + jq * 8;
            *(go_u2*)(dst + (j * per_rows + (size_t)(dd * 2 + 0) * 4 + e2 * 2 + al) * GA_FRAG_ROW) = h2;
            *(go_u2*)(dst + (j * per_rows + (size_t)(dd * 2 + 1) * 4 + e2 * 2 + al) * GA_FRAG_ROW) = l2;
        }
        *(go_f4*)((float*)(a.out + L.wcat_off) + e) = w;                                      // raw [2 Da][Di]
        float* wcatT = (float*)(a.out + L.wcatT_off);                                         // [Di][2 Da]
#pragma unroll
        for (int t = 0; t < 4; ++t) wcatT[(size_t)(di + t) * (2 * GA_DA) + u] = w[t];
        _Float16* p16 = (_Float16*)(a.out + L.w16_off);                                       // f16 hi / lo, fragment order: 4 neighbouring k
        *(go_u2*)(p16 + ga_frag_off(u, di, Di / 16, 0)) = h2;
        *(go_u2*)(p16 + ga_frag_off(u, di, Di / 16, 1)) = l2;
        __bf16* pT = (__bf16*)(a.out + L.wT16_off);                                           // bf16 hi / lo of the transpose, fragment order
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const __bf16 bh = (__bf16)w[t], bl = (__bf16)(w[t] - (float)bh);
            pT[ga_frag_off(di + t, u, GA_WT_KX / 16, 0)] = bh;
            pT[ga_frag_off(di + t, u, GA_WT_KX / 16, 1)] = bl;
        }
        return;
    }
    const int K = L.K, C = L.C, KP = a.KP;
    float* tab = (float*)(a.out + L.tab_off);
    if (blk < a.blkA + a.blkB + a.blkC) {
        // ---- Ww [K][Da], bw [K], bv [Da], bu [Da]
        const int o1 = KP * GA_DA, o2 = o1 + KP, o3 = o2 + GA_DA;
        int e; float s = 0.0f; bool mine;
        if (a.rec) {
            // gradients = sums of the gate-pass records: GM_EPW adjacent elements per wave, lanes stride the records (gemm_finish_kernel's
            // order, the same bits as one element per wave); lane j < GM_EPW carries element e0 + j through the update below
            const int lane = tid & 63;
            if (tid >= 64) return;      // one wave per block, neighbouring groups on one XCD (gemm_internal.h)
            const int e0 = gm_rec_group(blk - a.blkA - a.blkB, a.rec_stride);
            if (e0 < 0) return;
            float sv[GM_EPW];
            gm_record_sum_n(a.rec, e0 < a.rec_stride ? a.records : 0, a.rec_stride, e0 < a.rec_stride ? e0 : 0, a.rec_stride, lane, sv);
            s = sv[0];
#pragma unroll
            for (int j = 1; j < GM_EPW; ++j) s = (lane == j) ? sv[j] : s;
            e = e0 + (lane < GM_EPW ? lane : 0);
            mine = lane < GM_EPW && e < a.rec_stride;
        } else {
            // final gradients: one lane per element of the same index space
            e = (blk - a.blkA - a.blkB) * 256 + tid;
            mine = e < o3 + GA_DA;
            const float* g = nullptr;
            if (e < K * GA_DA) g = a.dWw + e;
            else if (e >= o1 && e < o1 + K) g = a.dbw + (e - o1);
            else if (e >= o2 && e < o3) g = a.dbv + (e - o2);
            else if (e >= o3 && e < o3 + GA_DA) g = a.dbu + (e - o3);
            if (mine && g) s = *g;
        }
        const GoCoef c = go_coef();
        if (!mine) return;
        float* bcat = (float*)(a.out + L.bcat_off);
        if (e < K * GA_DA) {
            if (a.rec) a.dWw[e] = s;
            if (!skip) tab[2 * GA_DA + e] = go_adamw1(a.Ww + e, s, a.m_off, a.v_off, c);      // tab[(2 + k) Da + col], e = k Da + col
        } else if (e >= o1 && e < o1 + K) {
            const int k = e - o1;
            if (a.rec) a.dbw[k] = s;
            if (!skip) ((float*)(a.out + L.bw_off))[k] = go_adamw1(a.bw + k, s, a.m_off, a.v_off, c);
        } else if (e >= o2 && e < o3) {
            const int cc = e - o2;
            if (a.rec) a.dbv[cc] = s;
            if (!skip) { const float w = go_adamw1(a.bv + cc, s, a.m_off, a.v_off, c); tab[cc] = w; bcat[cc] = w; }
        } else if (e >= o3 && e < o3 + GA_DA) {
            const int cc = e - o3;
            if (a.rec) a.dbu[cc] = s;
            if (!skip) { const float w = go_adamw1(a.bu + cc, s, a.m_off, a.v_off, c); tab[GA_DA + cc] = w; bcat[GA_DA + cc] = w; }
        }
        return;
    }
    // ---- heads (their gradients were written by the step's tail kernel): Wc [K][C][Di], bc [K][C], Ws [C][Di], bs [C]
    const GoCoef c = go_coef();
    if (skip) return;
    const int CD = C * Di;
    const int n_wc = K * CD, n_bc = K * C, n_ws = a.Ws ? CD : 0, n_bs = a.Ws ? C : 0;
    const int e = (blk - a.blkA - a.blkB - a.blkC) * 256 + tid;
    if (e < n_wc) {
        const int kk = e / CD, r = e - kk * CD;
        float* p = a.Wc[0]; const float* g = a.dWc[0];
#pragma unroll
        for (int q = 1; q < GO_MAXK; ++q) { p = (kk == q) ? a.Wc[q] : p; g = (kk == q) ? a.dWc[q] : g; }
        ((float*)(a.out + L.wc_off))[e] = go_adamw1(p + r, g[r], a.m_off, a.v_off, c);
    } else if (e < n_wc + n_bc) {
        const int x = e - n_wc, kk = x / C, r = x - kk * C;
        float* p = a.bc[0]; const float* g = a.dbc[0];
#pragma unroll
        for (int q = 1; q < GO_MAXK; ++q) { p = (kk == q) ? a.bc[q] : p; g = (kk == q) ? a.dbc[q] : g; }
        ((float*)(a.out + L.bc_off))[x] = go_adamw1(p + r, g[r], a.m_off, a.v_off, c);
    } else if (e < n_wc + n_bc + n_ws) {
        const int r = e - n_wc - n_bc;
        ((float*)(a.out + L.ws_off))[r] = go_adamw1(a.Ws + r, a.dWs[r], a.m_off, a.v_off, c);
    } else if (e < n_wc + n_bc + n_ws + n_bs) {
        const int r = e - n_wc - n_bc - n_ws;
        ((float*)(a.out + L.bs_off))[r] = go_adamw1(a.bs + r, a.dbs[r], a.m_off, a.v_off, c);
    }
}

static bool go_al16(const void* p) { return ((size_t)p & 15) == 0; }

// Can the closing launch serve this parameter set?  (16-byte alignment of the three matrices, their gradients and moments; every
// parameter inside the optimizer's flat buffer and the buffer exactly covered by them.)
int go_check(const GoTensors& t, int D, int Di, int K, int C, int mode, const float* flat, long long n_flat, const float* exp_avg,
             const float* exp_avg_sq) {
    if (mode != ACMIL_MODE_F16X3) return ACMIL_ERR_UNSUPPORTED;
    if (!flat || !exp_avg || !exp_avg_sq || n_flat <= 0) return ACMIL_ERR_NULL;
    if ((D & 15) || (Di & 31) || ((Di / 32) & 3) || K > GO_MAXK) return ACMIL_ERR_UNSUPPORTED;
    const long long m_off = exp_avg - flat, v_off = exp_avg_sq - flat;
    long long covered = 0;
    auto inside = [&](const float* p, long long n) {
        if (!p) return false;
        const long long o = p - flat;
        covered += n;
        return o >= 0 && o + n <= n_flat;
    };
    bool ok = inside(t.W1, (long long)Di * D) && inside(t.Wv, (long long)GA_DA * Di) && inside(t.bv, GA_DA) && inside(t.Wu, (long long)GA_DA * Di) &&
              inside(t.bu, GA_DA) && inside(t.Ww, (long long)K * GA_DA) && inside(t.bw, K);
    for (int k = 0; ok && k < K; ++k) ok = inside(t.Wc[k], (long long)C * Di) && inside(t.bc[k], C);
    if (ok && t.Ws) ok = inside(t.Ws, (long long)C * Di) && inside(t.bs, C);
    if (!ok || covered != n_flat) return ACMIL_ERR_SHAPE;
    float* const mats[3] = {t.W1, t.Wv, t.Wu};
    float* const grads[3] = {t.dW1, t.dWv, t.dWu};
    for (int q = 0; q < 3; ++q)
        if (!go_al16(mats[q]) || !go_al16(grads[q]) || !go_al16(mats[q] + m_off) || !go_al16(mats[q] + v_off)) return ACMIL_ERR_UNSUPPORTED;
    return ACMIL_OK;
}

// g_vu / g_w1 / job null: the gradient tensors are final (AdamW + re-pack only)
int go_launch(const GoTensors& t, const GemmArgs* g_vu, const GemmArgs* g_w1, const RowSumJob* job, int KP, void* packed, const GaLayout& L,
              const float* flat, const float* exp_avg, const float* exp_avg_sq, float lr, double beta1, double beta2, float eps, float wd,
              long long step, const float* skip_flag, int* skipped, float* flag_report, hipStream_t st) {
    GoArgs a;
    memset(&a, 0, sizeof(a));
    if (g_vu) { a.ws_vu = g_vu->ws; a.splits_vu = g_vu->splits; }
    if (g_w1) { a.ws_w1 = g_w1->ws; a.splits_w1 = g_w1->splits; }
    if (job) { a.rec = job->part; a.records = job->records; a.rec_stride = job->stride; }
    a.KP = KP;
    a.W1 = t.W1; a.Wv = t.Wv; a.bv = t.bv; a.Wu = t.Wu; a.bu = t.bu; a.Ww = t.Ww; a.bw = t.bw; a.Ws = t.Ws; a.bs = t.bs;
    a.dW1 = t.dW1; a.dWv = t.dWv; a.dbv = t.dbv; a.dWu = t.dWu; a.dbu = t.dbu; a.dWw = t.dWw; a.dbw = t.dbw; a.dWs = t.dWs; a.dbs = t.dbs;
    for (int k = 0; k < GO_MAXK; ++k) { a.Wc[k] = t.Wc[k]; a.bc[k] = t.bc[k]; a.dWc[k] = t.dWc[k]; a.dbc[k] = t.dbc[k]; }
    a.m_off = exp_avg - flat; a.v_off = exp_avg_sq - flat;
    a.lr = lr; a.eps = eps; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2; a.launch = step;
    a.skip_flag = skip_flag; a.skipped = skipped; a.flag_report = flag_report;
    a.out = (char*)packed; a.L = L;
    const long long nA = (long long)L.Di * L.D / 4, nB = (long long)2 * GA_DA * L.Di / 4;
    a.blkA = (int)((nA + 255) / 256); a.blkB = (int)((nB + 255) / 256); a.blkC = job ? gm_rec_blocks(job->stride) : (KP * GA_DA + KP + 2 * GA_DA + 255) / 256;
    const int nD = L.K * L.C * L.Di + L.K * L.C + (t.Ws ? L.C * L.Di + L.C : 0);
    const int blkD = (nD + 255) / 256;
    hipLaunchKernelGGL(ga_opt_step_kernel, dim3(a.blkA + a.blkB + a.blkC + blkD), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

// torch.optim.AdamW's update over the flat parameter buffer of ONE ACMIL_GA / ABMIL module AND the re-pack of its weights, one launch:
// acmil_adamw_step_report + acmil_ga_pack_weights for callers whose gradients are final when the optimizer runs (data-parallel steps:
// the bucket all-reduce sits between acmil_ga_train_step and this call).  Same arithmetic, bit for bit.
extern "C" int acmil_ga_adamw_pack(float* W1, float* Wv, float* bv, float* Wu, float* bu, float* Ww, float* bw, float* const* Wc, float* const* bc,
                                   float* Ws, float* bs,
                                   const float* dW1, const float* dWv, const float* dbv, const float* dWu, const float* dbu, const float* dWw,
                                   const float* dbw, const float* const* dWc, const float* const* dbc, const float* dWs, const float* dbs,
                                   int D, int Di, int Da, int K, int C, int mode, void* packed,
                                   const float* flat_params, long long n_flat, float* exp_avg, float* exp_avg_sq, float lr, double beta1,
                                   double beta2, float eps, float weight_decay, long long step, const float* skip_flag, int* skipped,
                                   float* flag_report, void* stream) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (K > GO_MAXK) return ACMIL_ERR_UNSUPPORTED;
    if (step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return ACMIL_ERR_SHAPE;
    if (!W1 || !Wv || !bv || !Wu || !bu || !Ww || !bw || !Wc || !bc || !packed) return ACMIL_ERR_NULL;
    if (!dW1 || !dWv || !dbv || !dWu || !dbu || !dWw || !dbw || !dWc || !dbc) return ACMIL_ERR_NULL;
    if ((Ws == nullptr) != (bs == nullptr) || (Ws && (!dWs || !dbs))) return ACMIL_ERR_NULL;
    GoTensors t;
    memset(&t, 0, sizeof(t));
    t.W1 = W1; t.Wv = Wv; t.bv = bv; t.Wu = Wu; t.bu = bu; t.Ww = Ww; t.bw = bw; t.Ws = Ws; t.bs = bs;
    t.dW1 = (float*)dW1; t.dWv = (float*)dWv; t.dbv = (float*)dbv; t.dWu = (float*)dWu; t.dbu = (float*)dbu; t.dWw = (float*)dWw; t.dbw = (float*)dbw;
    t.dWs = (float*)dWs; t.dbs = (float*)dbs;
    for (int k = 0; k < K; ++k) {
        if (!Wc[k] || !bc[k] || !dWc[k] || !dbc[k]) return ACMIL_ERR_NULL;
        t.Wc[k] = Wc[k]; t.bc[k] = bc[k]; t.dWc[k] = (float*)dWc[k]; t.dbc[k] = (float*)dbc[k];
    }
    rc = go_check(t, D, Di, K, C, mode, flat_params, n_flat, exp_avg, exp_avg_sq);
    if (rc != ACMIL_OK) return rc;
    return go_launch(t, nullptr, nullptr, nullptr, ga_kp(K), packed, ga_layout(D, Di, K, C, mode), flat_params, exp_avg, exp_avg_sq, lr, beta1, beta2,
                     eps, weight_decay, step, skip_flag, skipped, flag_report, (hipStream_t)stream);
}

// The refusal of the closing launch as a QUERY (nothing is launched): ACMIL_OK when acmil_ga_train_step_adamw / acmil_ga_train_step_group
// (adamw) / acmil_ga_adamw_pack would accept this parameter set, else the code they would return before their first launch.
extern "C" int acmil_ga_adamw_supported(const float* W1, const float* Wv, const float* bv, const float* Wu, const float* bu, const float* Ww,
                                        const float* bw, const float* const* Wc, const float* const* bc, const float* Ws, const float* bs,
                                        const float* dW1, const float* dWv, const float* dWu, int D, int Di, int Da, int K, int C, int mode,
                                        const float* flat_params, long long n_flat, const float* exp_avg, const float* exp_avg_sq) {
    int rc = ga_check_dims(D, Di, Da, K, C);
    if (rc != ACMIL_OK) return rc;
    if (K > GO_MAXK) return ACMIL_ERR_UNSUPPORTED;
    if (!W1 || !Wv || !bv || !Wu || !bu || !Ww || !bw || !Wc || !bc || !dW1 || !dWv || !dWu) return ACMIL_ERR_NULL;
    if ((Ws == nullptr) != (bs == nullptr)) return ACMIL_ERR_NULL;
    GoTensors t;
    memset(&t, 0, sizeof(t));
    t.W1 = (float*)W1; t.Wv = (float*)Wv; t.bv = (float*)bv; t.Wu = (float*)Wu; t.bu = (float*)bu; t.Ww = (float*)Ww; t.bw = (float*)bw;
    t.Ws = (float*)Ws; t.bs = (float*)bs; t.dW1 = (float*)dW1; t.dWv = (float*)dWv; t.dWu = (float*)dWu;
    for (int k = 0; k < K; ++k) {
        if (!Wc[k] || !bc[k]) return ACMIL_ERR_NULL;
        t.Wc[k] = (float*)Wc[k]; t.bc[k] = (float*)bc[k];
    }
    return go_check(t, D, Di, K, C, mode, flat_params, n_flat, exp_avg, exp_avg_sq);
}
