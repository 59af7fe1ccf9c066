// coresident_probe.hip -- does a short kernel on a second stream get onto the CUs while a long kernel occupies them?
// Kernel A: grid x block, spins for `us` microseconds (wall clock), with a chosen VGPR allocation and LDS footprint.
// Kernel B: 384 workgroups x 256 threads, trivial.  Reported: B's duration (events on its stream) alone and beside A.
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/coresident_probe tools/coresident_probe.hip && build/exp/coresident_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NV>
__global__ void spin_kernel(float* out, long long ticks, int lds_floats) {
    extern __shared__ float sm[];
    if (NV == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");       // forces the allocation of 128 / 96 / 64 VGPRs
    else if (NV == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    else asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (lds_floats) sm[threadIdx.x % lds_floats] = 1.0f;
    const long long t0 = wall_clock64();
    float acc = 0.0f;
    while (wall_clock64() - t0 < ticks) acc += 1.0f;
    if (acc < 0.0f) out[0] = acc + (lds_floats ? sm[0] : 0.0f);
}
__global__ void small_kernel(float* out) { out[blockIdx.x * blockDim.x + threadIdx.x] = 1.0f; }

int main() {
    float* buf; CK(hipMalloc(&buf, 1 << 22));
    hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    int least, greatest; CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));      // kHz
    const long long ticks = (long long)rate * 300 / 1000;                                     // 300 us
    printf("wall clock %d kHz; priority range least %d greatest %d\n", rate, least, greatest);
    struct Case { const char* name; int nv, block, grid, lds; } cases[] = {
        {"no A", 0, 0, 0, 0},
        {"A 128 vgpr, 384 thr, 512 wg, 30 KB lds", 128, 384, 512, 30720},
        {"A 128 vgpr, 384 thr, 512 wg, no lds", 128, 384, 512, 0},
        {"A  64 vgpr, 384 thr, 512 wg, 30 KB lds", 64, 384, 512, 30720},
        {"A  96 vgpr, 384 thr, 512 wg, 30 KB lds", 96, 384, 512, 30720},
        {"A 128 vgpr, 512 thr, 512 wg, 30 KB lds (full)", 128, 512, 512, 30720},
        {"A 128 vgpr, 256 thr, 512 wg, 30 KB lds", 128, 256, 512, 30720},
        {"A 128 vgpr, 256 thr, 256 wg, 30 KB lds", 128, 256, 256, 30720},
        {"A 128 vgpr, 384 thr, 256 wg, 30 KB lds", 128, 384, 256, 30720},
        {"A 128 vgpr, 512 thr, 256 wg, 64 KB lds", 128, 512, 256, 65536},
        {"A 128 vgpr, 384 thr, 1024 wg (queued), 30 KB lds", 128, 384, 1024, 30720},
    };
    const int bblocks[] = {256, 128, 64};
    for (int pass = 0; pass < 2; ++pass)
    for (auto& c : cases) {
        for (int which = 0; which < 3; ++which) {
            hipStream_t sb = s2;
            const int bb = bblocks[which];
            CK(hipDeviceSynchronize());
            if (c.nv == 128) hipLaunchKernelGGL(spin_kernel<128>, dim3(c.grid), dim3(c.block), c.lds, s1, buf, ticks, c.lds / 4);
            else if (c.nv == 96) hipLaunchKernelGGL(spin_kernel<96>, dim3(c.grid), dim3(c.block), c.lds, s1, buf, ticks, c.lds / 4);
            else if (c.nv == 64) hipLaunchKernelGGL(spin_kernel<64>, dim3(c.grid), dim3(c.block), c.lds, s1, buf, ticks, c.lds / 4);
            // let A get resident (host sleep ~50 us)
            { hipEvent_t t; (void)t; for (volatile int i = 0; i < 200000; ++i) {} }
            CK(hipEventRecord(e0, sb));
            hipLaunchKernelGGL(small_kernel, dim3(384 * 256 / bb), dim3(bb), 0, sb, buf + 4096);
            CK(hipEventRecord(e1, sb));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("%-52s  B with %3d-thread workgroups: %7.1f us\n", c.name, bb, ms * 1e3);
        }
    }
    return 0;
}
