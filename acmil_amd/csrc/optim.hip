// optim.hip -- AdamW over ONE flat fp32 parameter buffer (torch.optim.AdamW semantics, decoupled weight decay), one launch.
// Replaces the optimizer.step() of the reference's loop (Step3_WSI_classification_ACMIL.py:139,219: torch.optim.AdamW over
// 26 small tensors = 10 foreach launches, or 38 us in torch's fused variant); the training step is launch-bound at small
// bags, and parameters / gradients / moments already live in flat buffers (slide-level DP all-reduces one flat bucket).
//   p *= 1 - lr * wd ;  m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - b1^t, bc2 = 1 - b2^t computed by the caller (the step count lives on the host).
#include "ga_common.h"

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                        float wd, float inv_bc1, float inv_sqrt_bc2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float pi = p[i] * (1.0f - lr * wd);
        p[i] = pi - (lr * inv_bc1) * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

extern "C" int acmil_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                                float bias_correction2, void* stream) {
    if (n <= 0 || bias_correction1 <= 0.0f || bias_correction2 <= 0.0f) return ACMIL_ERR_SHAPE;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return ACMIL_ERR_NULL;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, 1.0f / bias_correction1, 1.0f / sqrtf(bias_correction2));
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
