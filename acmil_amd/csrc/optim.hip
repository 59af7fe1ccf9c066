// optim.hip -- AdamW over ONE flat fp32 parameter buffer (torch.optim.AdamW semantics, decoupled weight decay), one launch.
// Replaces the optimizer.step() of the reference's loop (Step3_WSI_classification_ACMIL.py:139,219: torch.optim.AdamW over
// 26 small tensors = 10 foreach launches, or 38 us in torch's fused variant); the training step is launch-bound at small
// bags, and parameters / gradients / moments already live in flat buffers (slide-level DP all-reduces one flat bucket).
//   p *= 1 - lr * wd ;  m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - b1^t, bc2 = 1 - b2^t formed on the device in double (t = the caller's launch ordinal minus the skipped launches).
#include "ga_common.h"
#include "optim_kernel.h"

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, float lr, double beta1, double beta2, float eps,
                                                        float wd, long long launch, const float* __restrict__ skip_flag,
                                                        int* __restrict__ skipped, float* __restrict__ flag_report) {
    // device-side skip (like an AMP overflow skip): a non-zero / NaN flag -- the all-reduced range flag of a split-f16 training
    // step, csrc/ga_step.hip -- leaves parameters and moments untouched and is counted in *skipped; the host learns about it
    // later (optim.py).  The step number of the bias corrections is the launch ordinal minus the launches skipped so far, so
    // the update stays torch.optim.AdamW's for the steps that are applied.
    __shared__ float s_bc[2];
    // the flag, reported to the host without a copy on the stream: one lane stores it to a device-visible address (pinned host
    // memory); the host reads it after the event it records behind this launch
    if (flag_report && blockIdx.x == 0 && threadIdx.x == 0) __builtin_nontemporal_store(skip_flag ? skip_flag[0] : 0.0f, flag_report);
    if (skip_flag && !(skip_flag[0] == 0.0f)) {
        if (skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
        return;
    }
    if (threadIdx.x == 0) {
        const long long t = launch - (skipped ? (long long)*skipped : 0);
        adamw_bias(beta1, beta2, t, s_bc[0], s_bc[1]);
    }
    __syncthreads();
    const float inv_bc1 = s_bc[0], inv_sqrt_bc2 = s_bc[1];
    const float b1 = (float)beta1, b2 = (float)beta2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_update(pi, g[i], mi, vi, lr, wd, inv_bc1, inv_sqrt_bc2, eps, b1, b2);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

extern "C" int acmil_adamw_step_report(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                                       double beta1, double beta2, float eps, float weight_decay, long long step, const float* skip_flag,
                                       int* skipped, float* flag_report, void* stream) {
    if (n <= 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return ACMIL_ERR_SHAPE;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return ACMIL_ERR_NULL;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, step, skip_flag, skipped, flag_report);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                                double beta1, double beta2, float eps, float weight_decay, long long step, const float* skip_flag,
                                int* skipped, void* stream) {
    return acmil_adamw_step_report(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, skip_flag, skipped,
                                   nullptr, stream);
}
