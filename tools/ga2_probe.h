// tools/ga2_probe.h -- measurement build of the hooks of acmil_amd/csrc/ga_forward_kernel_v2.h.  NOT part of the product: included
// only when tools/build_variants.sh compiles a timing variant with -DGA2_TOOLS (plus -DGA2_PROF and / or -DGA2_ABL=<bits>).
//   GA2_ABL bits: 1 no bag-row DMA, 2 no weight DMA, 4 no GEMM1 MFMAs, 8 no GEMM2 MFMAs.  Results are WRONG with any bit set.
//   GA2_PROF: s_memtime accounting; per-wave cycle totals REPLACE the first 16 scores of each 32-patch group of A_out[0]
//             (read back by tools/probe_ga.py): total, GEMM1 vmcnt wait, GEMM1 barrier wait, softmax, pooling, GEMM1, GEMM2 + gates,
//             combine; then start time (two 24-bit halves), HW_ID, XCC_ID, second-workgroup flag, workgroup id, tile.
#pragma once
#ifndef GA2_ABL
#define GA2_ABL 0
#endif
__device__ __forceinline__ f32x16 ga2_keep(f16x8 a, f16x8 b, f32x16 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }
#define GA2_MFMA1(A, B, C) ((GA2_ABL & 4) ? ga2_keep(A, B, C) : __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0))
#define GA2_MFMA2(A, B, C) ((GA2_ABL & 8) ? ga2_keep(A, B, C) : __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0))

struct Ga2Probe {
    static constexpr bool DMA_X = !(GA2_ABL & 1), DMA_W = !(GA2_ABL & 2);
#ifdef GA2_PROF
    unsigned long long vm = 0, bar = 0, ta = 0, tb = 0;
    unsigned long long t0 = 0, vm0 = 0, bar0 = 0, t1 = 0, vm1 = 0, bar1 = 0, t2 = 0, t2a = 0, tpool = 0;
    __device__ __forceinline__ void sync_begin() { ta = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void sync_loaded() { tb = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void sync_released() {
        const unsigned long long tc = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        vm += tb - ta; bar += tc - tb;
    }
    __device__ __forceinline__ void tile_begin() { t0 = __builtin_amdgcn_s_memtime(); vm0 = vm; bar0 = bar; }
    __device__ __forceinline__ void gemm1_end() { t1 = __builtin_amdgcn_s_memtime(); vm1 = vm; bar1 = bar; }
    __device__ __forceinline__ void gemm2_end() { t2 = __builtin_amdgcn_s_memtime(); tpool = t2; t2a = t2; }
    __device__ __forceinline__ void softmax_end() { t2a = __builtin_amdgcn_s_memtime(); tpool = t2a; }
    __device__ __forceinline__ void pool_end() { tpool = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void tile_end(float* A_out, int lane, int m0, int N, int tile) {
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        const float pv[8] = {(float)(t3 - t0), (float)(vm1 - vm0), (float)(bar1 - bar0), (float)(t2a - t2),
                             (float)(tpool - t2a), (float)(t1 - t0), (float)(t2 - t1), (float)(t3 - tpool)};
        if (A_out && lane < 8 && m0 + 8 <= N) A_out[m0 + lane] = pv[lane];
        const float pw[8] = {(float)(unsigned)(t0 & 0xffffff), (float)(unsigned)((t0 >> 24) & 0xffffff),
                             (float)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11)), (float)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)),
                             (float)__builtin_amdgcn_s_getreg(6 | (0 << 6) | (7 << 11)), (float)blockIdx.x, (float)tile, 0.0f};
        if (A_out && lane >= 8 && lane < 16 && m0 + 16 <= N) A_out[m0 + lane] = pw[lane - 8];
    }
#else
    __device__ __forceinline__ void sync_begin() {}
    __device__ __forceinline__ void sync_loaded() {}
    __device__ __forceinline__ void sync_released() {}
    __device__ __forceinline__ void tile_begin() {}
    __device__ __forceinline__ void gemm1_end() {}
    __device__ __forceinline__ void gemm2_end() {}
    __device__ __forceinline__ void softmax_end() {}
    __device__ __forceinline__ void pool_end() {}
    __device__ __forceinline__ void tile_end(float*, int, int, int, int) {}
#endif
};
