// ga_forward_inst.hip -- one translation unit per (ND, KP, MODE) family of the fused forward kernel,
// compiled with -DGA_ND=.. -DGA_KP=.. -DGA_MODE=.. (see Makefile) so the families build in parallel.
#include "ga_forward_kernel.h"
#if GA_MODE == ACMIL_MODE_F16X3
#include "ga_forward_kernel_v2.h"
#endif

#define GA_CAT_(a, b, c, d) a##b##_##c##_##d
#define GA_CAT(a, b, c, d) GA_CAT_(a, b, c, d)

int GA_CAT(ga_fwd_family_, GA_ND, GA_KP, GA_MODE)(const GaFwdArgs& a, int x_dtype, bool pool, int version, hipStream_t st) {
#if GA_MODE == ACMIL_MODE_F16X3
    if (version == 2) {   // software-pipelined split-f16 kernel (ga_forward_kernel_v2.h)
        switch (x_dtype) {
            case ACMIL_DTYPE_F32: return ga_launch_fwd2<GA_ND, GA_KP, ACMIL_DTYPE_F32>(a, pool, st);
            case ACMIL_DTYPE_F16: return ga_launch_fwd2<GA_ND, GA_KP, ACMIL_DTYPE_F16>(a, pool, st);
            case ACMIL_DTYPE_BF16: return ga_launch_fwd2<GA_ND, GA_KP, ACMIL_DTYPE_BF16>(a, pool, st);
        }
        return ACMIL_ERR_UNSUPPORTED;
    }
#endif
    (void)version;
    switch (x_dtype) {
        case ACMIL_DTYPE_F32: return ga_launch_fwd<GA_ND, GA_KP, GA_MODE, ACMIL_DTYPE_F32>(a, pool, st);
        case ACMIL_DTYPE_F16: return ga_launch_fwd<GA_ND, GA_KP, GA_MODE, ACMIL_DTYPE_F16>(a, pool, st);
        case ACMIL_DTYPE_BF16: return ga_launch_fwd<GA_ND, GA_KP, GA_MODE, ACMIL_DTYPE_BF16>(a, pool, st);
    }
    return ACMIL_ERR_UNSUPPORTED;
}
