// Hardware check of ds_read_b64_tr_b16 (gfx950) for ga_forward_kernel_v2.h: with per-lane 8-byte addresses A_l, lane l of a
// 16-lane group receives element (c % 4) of the 8 bytes addressed by lane 4j + c / 4 of its group (c = l & 15), j = 0..3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(s16x4* out) {
    __shared__ __attribute__((aligned(16))) short sm[64 * 4 * 2];
    const int l = threadIdx.x;
    for (int e = l; e < 512; e += 64) sm[e] = (short)e;
    __syncthreads();
    typedef __attribute__((address_space(3))) h16x4* lp;
    const int slot = (l ^ 5) + ((l & 3) == 1 ? 64 : 0);          // arbitrary per-lane 8-byte slot
    h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)((char*)sm + slot * 8));
    out[l] = __builtin_bit_cast(s16x4, v);
}
int main() {
    s16x4* d; hipMalloc(&d, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    s16x4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int c = l & 15, src = (l & 48) + 4 * j + c / 4;
            const int slot = (src ^ 5) + ((src & 3) == 1 ? 64 : 0);
            const int expect = slot * 4 + (c % 4);
            if (h[l][j] != expect) { if (bad < 8) printf("lane %d j %d got %d expect %d\n", l, j, h[l][j], expect); ++bad; }
        }
    printf("TRTEST %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
