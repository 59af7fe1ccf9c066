// optim_kernel.h -- the AdamW update of ONE element, shared by the flat optimizer launch (optim.hip) and by the training step's
// closing launch (ga_opt_step.hip) so that both apply bit-identical arithmetic (torch.optim.AdamW, decoupled weight decay):
//   p *= 1 - lr wd ;  m = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2 ;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
#pragma once
#include <hip/hip_runtime.h>

// 1 / bc1 and 1 / sqrt(bc2) of step t (formed in double: b^t loses nothing for t up to millions)
__device__ __forceinline__ void adamw_bias(double beta1, double beta2, long long t, float& inv_bc1, float& inv_sqrt_bc2) {
    inv_bc1 = (float)(1.0 / (1.0 - pow(beta1, (double)t)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)t)));
}

__device__ __forceinline__ void adamw_update(float& p, float g, float& m, float& v, float lr, float wd, float inv_bc1, float inv_sqrt_bc2,
                                             float eps, float b1, float b2) {
#pragma clang fp contract(off)      // every caller rounds the same way, whatever surrounds the inlined body
    const float mi = b1 * m + (1.0f - b1) * g;
    const float vi = b2 * v + (1.0f - b2) * g * g;
    m = mi; v = vi;
    const float pi = p * (1.0f - lr * wd);
    p = pi - (lr * inv_bc1) * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
}
