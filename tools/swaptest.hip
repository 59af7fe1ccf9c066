// Hardware check of v_permlane32_swap and the DPP row scan used by ga_forward_kernel_v2.h
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out, float* fo) {
    const int l = threadIdx.x;
    const unsigned a = 1000 + l, b = 2000 + l;
    const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = sw[0]; out[64 + l] = sw[1];
    float m = (float)((l * 37) % 101);
    float v = m;
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1e30f), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1e30f), __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1e30f), __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1e30f), __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1e30f), __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false)));
    fo[l] = v;
}
int main() {
    unsigned* d; float* f; hipMalloc((void**)&d, 128 * 4); hipMalloc((void**)&f, 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, f);
    unsigned h[128]; float hf[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
    printf("sw0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
    printf("sw1: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
    float m0 = -1, m1 = -1;
    for (int l = 0; l < 32; ++l) { float x = (float)((l * 37) % 101); if (x > m0) m0 = x; }
    for (int l = 32; l < 64; ++l) { float x = (float)((l * 37) % 101); if (x > m1) m1 = x; }
    printf("scan: lane31 %g (expect %g) lane63 %g (expect %g) lane15 %g lane47 %g\n", hf[31], m0, hf[63], m1, hf[15], hf[47]);
    return 0;
}
