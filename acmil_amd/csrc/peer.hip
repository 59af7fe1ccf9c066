// peer.hip -- one-shot direct gradient all-reduce over peer-mapped buffers, fused into the AdamW launch (round 4).
//
// Slide-level data parallelism reduces ONE flat fp32 bucket per step (833 KB at D=512, D_inner=256, 7 classes + the range flag).  A
// ring / tree collective is latency-bound at that size (RCCL: 25-40 us on an 8-GPU xGMI node) and sits serially between the last
// weight gradient and the optimizer.  Here every rank
//   1. publishes its bucket into its own IPC-shared slot with system-scope (write-through) stores, fences, and raises a step flag in
//      every peer's flag array (peer_publish_kernel), and
//   2. runs adamw_peer_kernel: wait for the flags of all peers (bounded spin), read the W buckets -- the 7 remote ones straight over
//      xGMI through the mapped pointers --, add them in RANK ORDER (every rank forms bit-identical sums), divide by W, and apply the
//      AdamW update of optim.hip.  No collective launch, no reduced copy of the gradients in memory.
// Slots are double-buffered by step parity: a rank can overwrite slot (t+1) & 1 only after its optimizer launch of step t, which has
// seen every peer's flag of step t, i.e. every peer has finished reading the slots of step t-1.
// torch.distributed / RCCL stays the fallback and the default (acmil_amd/peer.py decides); the mapping is torch's CUDA IPC.
#include "ga_common.h"
#include "optim_kernel.h"

#define PEER_MAX 8      // ranks of one node

struct PeerSlots { const float* slot[PEER_MAX]; };     // the W buckets of this step's parity (own included), indexed by rank
struct PeerFlags { unsigned* flags[PEER_MAX]; };       // every rank's flag array [PEER_MAX] (own included)

__device__ __forceinline__ void peer_store_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float peer_load_sys(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// own bucket -> own shared slot (write-through), then the step flag into every rank's flag array.  Every workgroup fences after its
// stores (each XCD has its own L2: one thread's fence covers one XCD); the last workgroup to arrive raises the flags.
__global__ __launch_bounds__(256) void peer_publish_kernel(const float* __restrict__ src, float* __restrict__ slot, long long n, PeerFlags pf,
                                                          int world, int rank, unsigned step, unsigned* __restrict__ arrive) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) peer_store_sys(slot + i, src[i]);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(arrive, 1u);
        if (t == gridDim.x - 1) {
            atomicExch(arrive, 0u);
            __threadfence_system();
            for (int r = 0; r < world; ++r) __hip_atomic_store(pf.flags[r] + rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// wait + reduce + AdamW.  err[0] is set to 1 when a peer's flag did not arrive within timeout_ticks (100 MHz wall clock): the launch
// then changes nothing -- the host raises.  Grid <= one workgroup per CU: a spinning launch must leave room for other work on the GPU
// (two ranks sharing one GPU in the test suite).
__global__ __launch_bounds__(256) void adamw_peer_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, long long n,
                                                        PeerSlots ps, const unsigned* __restrict__ myflags, int world, int rank, unsigned step,
                                                        long long timeout_ticks, int* __restrict__ err, float lr, double beta1, double beta2,
                                                        float eps, float wd, long long launch, int use_flag, int* __restrict__ skipped,
                                                        float* __restrict__ flag_report, float* __restrict__ reduced_out) {
    __shared__ int s_ok;
    __shared__ float s_bc[2];
    // (once a wait has timed out every later launch returns at once: a dead peer costs one timeout, not one per step)
    if (threadIdx.x == 0) s_ok = (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) ? 1 : 0;
    __syncthreads();
    // (the host sees a dead reduction through the tracked flag as well: -1 instead of the 0 / 1 of a live step, no extra synchronisation)
    if (!s_ok) { if (flag_report && blockIdx.x == 0 && threadIdx.x == 0) __builtin_nontemporal_store(-1.0f, flag_report); return; }
    if ((int)threadIdx.x < world && (int)threadIdx.x != rank) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(myflags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < step) {
            if (wall_clock64() - t0 > timeout_ticks) { s_ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (!s_ok) {
        if (threadIdx.x == 0) { atomicExch(err, 1); if (flag_report && blockIdx.x == 0) __builtin_nontemporal_store(-1.0f, flag_report); }
        return;
    }
    __threadfence_system();
    // the range flag of the step rides behind the gradients (element n of every bucket): its sum decides for all ranks alike
    float flag = 0.0f;
    if (use_flag) {
        for (int r = 0; r < world; ++r) flag += peer_load_sys(ps.slot[r] + n);
        flag = flag / (float)world;
    }
    if (flag_report && blockIdx.x == 0 && threadIdx.x == 0) __builtin_nontemporal_store(flag, flag_report);
    if (reduced_out && use_flag && blockIdx.x == 0 && threadIdx.x == 0) reduced_out[n] = flag;      // (slot n exists only with the flag)
    const bool skip = use_flag && !(flag == 0.0f);
    if (skip && skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
    if (threadIdx.x == 0) {
        const long long t = launch - (skipped ? (long long)*skipped : 0);      // (only used by launches that are applied)
        adamw_bias(beta1, beta2, t, s_bc[0], s_bc[1]);
    }
    __syncthreads();
    const float inv_bc1 = s_bc[0], inv_sqrt_bc2 = s_bc[1];
    const float b1 = (float)beta1, b2 = (float)beta2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float gi = 0.0f;
        for (int r = 0; r < world; ++r) gi += peer_load_sys(ps.slot[r] + i);      // rank order: the same sum on every rank
        gi = gi / (float)world;
        if (reduced_out) reduced_out[i] = gi;
        if (skip) continue;
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_update(pi, gi, mi, vi, lr, wd, inv_bc1, inv_sqrt_bc2, eps, b1, b2);      // (optim_kernel.h: the flat launch's arithmetic, bit for bit)
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

extern "C" int acmil_peer_publish(const float* bucket, float* my_slot, long long n, void* const* flag_arrays, int world, int rank,
                                  unsigned step, unsigned* arrive, void* stream) {
    if (n <= 0 || world < 1 || world > PEER_MAX || rank < 0 || rank >= world || step == 0) return ACMIL_ERR_SHAPE;
    if (!bucket || !my_slot || !flag_arrays || !arrive) return ACMIL_ERR_NULL;
    PeerFlags pf;
    for (int r = 0; r < PEER_MAX; ++r) { pf.flags[r] = (unsigned*)flag_arrays[r < world ? r : 0]; if (!pf.flags[r]) return ACMIL_ERR_NULL; }
    const unsigned blocks = (unsigned)((n + 1023) / 1024 < 128 ? (n + 1023) / 1024 : 128);
    hipLaunchKernelGGL(peer_publish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bucket, my_slot, n, pf, world, rank, step, arrive);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}

extern "C" int acmil_adamw_step_peer(float* params, float* exp_avg, float* exp_avg_sq, long long n, const void* const* slots,
                                     const unsigned* my_flags, int world, int rank, unsigned step, double timeout_s, int* err, float lr,
                                     double beta1, double beta2, float eps, float weight_decay, long long launch, int use_flag, int* skipped,
                                     float* flag_report, float* reduced_out, void* stream) {
    if (n <= 0 || launch < 1 || world < 1 || world > PEER_MAX || rank < 0 || rank >= world || step == 0) return ACMIL_ERR_SHAPE;
    if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(timeout_s > 0.0)) return ACMIL_ERR_SHAPE;
    if (!params || !exp_avg || !exp_avg_sq || !slots || !my_flags || !err) return ACMIL_ERR_NULL;
    PeerSlots ps;
    for (int r = 0; r < PEER_MAX; ++r) { ps.slot[r] = (const float*)slots[r < world ? r : 0]; if (!ps.slot[r]) return ACMIL_ERR_NULL; }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long blocks = (n + 255) / 256;
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(adamw_peer_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, exp_avg, exp_avg_sq, n, ps, my_flags,
                       world, rank, step, (long long)(timeout_s * 1.0e8), err, lr, beta1, beta2, eps, weight_decay, launch, use_flag, skipped,
                       flag_report, reduced_out);
    return hipGetLastError() == hipSuccess ? ACMIL_OK : ACMIL_ERR_LAUNCH;
}
