// ga_forward_inst.hip -- one translation unit per (ND, KP, MODE) family of the fused forward kernel,
// compiled with -DGA_ND=.. -DGA_KP=.. -DGA_MODE=.. (see Makefile) so the families build in parallel.
#include "ga_forward_kernel.h"

#define GA_CAT_(a, b, c, d) a##b##_##c##_##d
#define GA_CAT(a, b, c, d) GA_CAT_(a, b, c, d)

int GA_CAT(ga_fwd_family_, GA_ND, GA_KP, GA_MODE)(const GaFwdArgs& a, int x_dtype, bool pool, hipStream_t st) {
    switch (x_dtype) {
        case ACMIL_DTYPE_F32: return ga_launch_fwd<GA_ND, GA_KP, GA_MODE, ACMIL_DTYPE_F32>(a, pool, st);
        case ACMIL_DTYPE_F16: return ga_launch_fwd<GA_ND, GA_KP, GA_MODE, ACMIL_DTYPE_F16>(a, pool, st);
        case ACMIL_DTYPE_BF16: return ga_launch_fwd<GA_ND, GA_KP, GA_MODE, ACMIL_DTYPE_BF16>(a, pool, st);
    }
    return ACMIL_ERR_UNSUPPORTED;
}
