// hbm_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS PATTERN of ga_fwd_kernel.
//
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced stream and is
// "uncalibrated" for other widths -- so this tool reads a KNOWN byte count with exactly the bag-tile pattern of the fused
// forward (per wave and step: 2 LDS-DMA instructions, each 16 patches x one 64-byte row segment, rows D*4 bytes apart)
// and writes a known byte count (fp32, 256 B per wave-instruction, like the A_out / partial stores).  Run it under
//   rocprofv3 --pmc FETCH_SIZE ...   and   rocprofv3 --pmc WRITE_SIZE ...
// and divide: correction = known bytes / reported bytes.  tools/pmc_ga.py does that and applies it to the GA kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/exp/hbm_calib tools/hbm_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const char* gsrc, unsigned ldst) {
    unsigned keep;
    const unsigned lds = __builtin_amdgcn_readfirstlane(ldst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
}

// one workgroup = 256 consecutive rows of x [N, D] fp32; D/16 steps; wave w copies rows 32w .. 32w+31
__global__ __launch_bounds__(512) void calib_read(const char* x, int N, int D, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const int m0 = blockIdx.x * 256 + wave * 32;
    for (int s = 0; s < D / 16; ++s) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int row = m0 + q * 16 + (lane >> 2);
            row = row < N ? row : N - 1;
            glds16(x + ((size_t)row * D + s * 16) * 4 + (lane & 3) * 16, lds + ((s & 3) * 8 + wave) * 2048 + q * 1024);
        }
        if ((s & 3) == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && threadIdx.x == 0 && ((float*)smem)[0] == 123.456f) sink[0] = 1.f;   // keep the loads observable
}

__global__ __launch_bounds__(256) void calib_write(float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)i;
}

int main(int argc, char** argv) {
    const int N = 50000, D = 512, bags = 8, reps = argc > 1 ? atoi(argv[1]) : 10;
    const size_t bag_bytes = (size_t)N * D * 4, wn = (size_t)16 << 20;   // 64 MB of fp32 writes per launch
    char* buf; float* out; float* sink;
    if (hipMalloc(&buf, bag_bytes * bags) != hipSuccess || hipMalloc(&out, wn * 4 * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bag_bytes * bags);
    hipFuncSetAttribute((const void*)calib_read, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(calib_read, dim3((N + 255) / 256), dim3(512), 65536, 0, buf + (size_t)(r % bags) * bag_bytes, N, D, sink);
        hipLaunchKernelGGL(calib_write, dim3(2048), dim3(256), 0, 0, out + (size_t)(r % 4) * wn, wn);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("calib_read: %zu bytes per launch; calib_write: %zu bytes per launch; %d launches each\n", bag_bytes, wn * 4, reps);
    return 0;
}
