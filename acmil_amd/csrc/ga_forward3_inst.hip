// ga_forward3_inst.hip -- one translation unit per (ND, PB, KP) family of the one-wave-per-SIMD fused forward kernel
// (ga_forward_kernel_v3.h), compiled with -DGA3_ND=.. -DGA3_PB=.. -DGA3_KP=.. (see Makefile) so the families build in parallel.
#include "ga_forward_kernel_v3.h"

#define GA3_CAT_(a, b, c, d) a##b##_##c##_##d
#define GA3_CAT(a, b, c, d) GA3_CAT_(a, b, c, d)

int GA3_CAT(ga_fwd3_family_, GA3_ND, GA3_PB, GA3_KP)(const GaFwdArgs& a, int x_dtype, bool pool, hipStream_t st) {
    switch (x_dtype) {
#if GA3_PB <= 2      // (PB = 3: 16-bit bags only -- an fp32 bag would need 6 bag pieces per wave and step)
        case ACMIL_DTYPE_F32: return ga_launch_fwd3<GA3_ND, GA3_PB, GA3_KP, ACMIL_DTYPE_F32>(a, pool, st);
#endif
        case ACMIL_DTYPE_F16: return ga_launch_fwd3<GA3_ND, GA3_PB, GA3_KP, ACMIL_DTYPE_F16>(a, pool, st);
        case ACMIL_DTYPE_BF16: return ga_launch_fwd3<GA3_ND, GA3_PB, GA3_KP, ACMIL_DTYPE_BF16>(a, pool, st);
    }
    return ACMIL_ERR_UNSUPPORTED;
}
