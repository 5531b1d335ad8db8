// rs_ppo.hip -- policy-side kernels: MFMA forward of the FF_core MLP, the fused on-device collector
// (rs_rollout) and the fused PPO loss/gradient pass (rs_ppo_grad).  See rs_mlp.hpp for the MFMA mapping.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/radsearch.h"
#include "rs_mlp.hpp"
#include "rs_handle.hpp"

static inline RsMlpParams to_dev(const rs_mlp_params* p) { return RsMlpParams{p->w1, p->b1, p->w2, p->b2, p->w3, p->b3}; }

// load this lane's sample (11 floats, zero padded) and its partner's (lane ^ 32)
__device__ __forceinline__ void rs_exchange_x(const float (&xo)[RS_IN_PAD], float (&xp)[RS_IN_PAD]) {
#pragma unroll
    for (int k = 0; k < RS_IN_PAD; ++k) xp[k] = __shfl_xor(xo[k], 32);
}

// ------------------------------------------------------------------------------------------------
// Policy forward for M samples: one wave per 64 samples, grid-stride over sample groups.
__global__ void __launch_bounds__(64) rs_policy_forward_kernel(RsMlpParams pa, RsMlpParams pc, const float* __restrict__ x, int M,
                                                               float* __restrict__ logits, float* __restrict__ value) {
    extern __shared__ __align__(16) float smem_f[];
    RsMlpLds<8> A; RsMlpLds<1> Cn;
    A.carve(smem_f);
    Cn.carve(smem_f + rs_mlp_lds_floats(8));
    A.fill(pa); Cn.fill(pc);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int groups = (M + 63) / 64;
    for (int gi = blockIdx.x; gi < groups; gi += gridDim.x) {
        const int m = gi * 64 + lane;
        const int mm = min(m, M - 1);
        float xo[RS_IN_PAD], xp[RS_IN_PAD];
#pragma unroll
        for (int k = 0; k < RS_IN; ++k) xo[k] = x[(size_t)mm * RS_IN + k];
        xo[11] = 0.0f;
        rs_exchange_x(xo, xp);
        if (logits) {
            float lo[8];
            rs_mlp_forward<8>(A, xo, xp, lo);
            if (m < M) {
#pragma unroll
                for (int o = 0; o < 8; ++o) logits[(size_t)m * 8 + o] = lo[o];
            }
        }
        if (value) {
            float v[1];
            rs_mlp_forward<1>(Cn, xo, xp, v);
            if (m < M) value[m] = v[0];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K6: fused collector (rs_rollout16.hpp).  Per-episode Welford standardisation state of one env:
struct RsWelford {
    double count, mean, sq, std;
    __device__ __forceinline__ void update(double x) {          // StatisticStandardization.update (RADTEAM_core.py:215-251)
        count += 1.0;
        if (count == 1.0) { mean = x; }
        else {
            double mean_new = mean + (x - mean) / count;
            sq = sq + (x - mean) * (x - mean_new);
            mean = mean_new;
            std = fmax(sqrt(sq / (count - 1.0)), 1.0);
        }
    }
    __device__ __forceinline__ float standardize(float x) const { return (float)(((double)x - mean) / std); }
    __device__ __forceinline__ void reset() { count = 0.0; mean = 0.0; sq = 0.0; std = 1.0; }
};

#include "rs_rollout16.hpp"

// ------------------------------------------------------------------------------------------------
// K7: fused PPO loss + gradients for one FF_core network (NOUT = 8: actor, NOUT = 1: critic): rs_ppo_grad2.hpp.
// Gradient accumulators stay in registers for the whole launch; each workgroup writes one partial slab and
// rs_ppo_reduce_kernel sums the slabs in a fixed order (bitwise reproducible, no float atomics).
__host__ __device__ constexpr int rs_net_params(int nout) { return 64 * 11 + 64 + 64 * 64 + 64 + nout * 64 + nout; }

__device__ __forceinline__ void rs_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#include "rs_ppo_grad2.hpp"

// deterministic reduction of the per-workgroup slabs: a workgroup owns 64 consecutive parameters; 16 thread
// groups each sum a fixed 1/16 of the slabs in order (coalesced 256-byte rows), then the 16 partial sums are
// added in a fixed order through LDS.  Same order every launch -> bitwise reproducible gradients.
__global__ void __launch_bounds__(1024) rs_ppo_reduce_kernel(const float* __restrict__ pa, const float* __restrict__ pc,
                                                             const double* __restrict__ sa, const double* __restrict__ sc, int n_waves,
                                                             float* __restrict__ grads, double* __restrict__ stats, float alpha, float vf,
                                                             const int* __restrict__ stop) {
    __shared__ float part[16][64];
    __shared__ double spart[5][64];
    constexpr int NA = rs_net_params(8), NC = rs_net_params(1);
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + col;
    if (stop && *stop) {
        // early stop already hit: the gradient kernels were no-ops.  Publish zeros so that the data-parallel all-reduce that
        // follows every iteration (the host does not know the stop state) keeps summing finite, idempotent values.
        if (grp == 0 && p < NA + NC + RS_PPO_STATS_TAIL) grads[p] = 0.0f;
        if (blockIdx.x == 0 && threadIdx.x < 5) stats[threadIdx.x] = 0.0;
        return;
    }
    float acc = 0.0f;
    if (p < NA + NC) {
        const float* src = (p < NA) ? pa + p : pc + (p - NA);
        const int stride = (p < NA) ? NA : NC;
        const int per = (n_waves + 15) / 16;
        const int w0 = grp * per, w1 = min(w0 + per, n_waves);
        for (int w = w0; w < w1; ++w) acc += src[(size_t)w * stride];
    }
    part[grp][col] = acc;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // statistics: 64 threads each own a fixed subset of the waves, then a fixed-order sum
        double t[5] = {0, 0, 0, 0, 0};
        for (int w = threadIdx.x; w < n_waves; w += 64) {
            t[0] += sa[w * 5 + 0]; t[1] += sa[w * 5 + 1]; t[2] += sa[w * 5 + 2]; t[4] += sa[w * 5 + 4];
            t[3] += sc[w * 5 + 3];
        }
        for (int q = 0; q < 5; ++q) spart[q][threadIdx.x] = t[q];
    }
    __syncthreads();
    if (grp == 0 && p < NA + NC) {
        float v = part[0][col];
        for (int g2 = 1; g2 < 16; ++g2) v += part[g2][col];
        grads[p] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double t[5];
        for (int q = 0; q < 5; ++q) { double v = 0; for (int i = 0; i < 64; ++i) v += spart[q][i]; t[q] = v; }
        t[4] = -(t[4] - (double)vf * t[3] + (double)alpha * t[1]);           // ppo.py:1221-1225
        // the statistics also ride at the end of the gradient bucket as float32 (hi, lo) pairs, so that ONE all-reduce of the
        // bucket exchanges gradients and statistics between the ranks (xGMI all-reduces of this size are latency bound)
        for (int q = 0; q < 5; ++q) {
            stats[q] = t[q];
            const float hi = (float)t[q];
            grads[NA + NC + 2 * q] = hi;
            grads[NA + NC + 2 * q + 1] = (float)(t[q] - (double)hi);
        }
        for (int q = 10; q < RS_PPO_STATS_TAIL; ++q) grads[NA + NC + q] = 0.0f;
    }
}

#define RS_GRAD_BLOCKS 256

// Adam (torch.optim.Adam semantics) + the KL early-stop decision, on the device.
struct RsParamSeg { float* p[12]; int off[13]; };

__global__ void __launch_bounds__(256) rs_adam_apply_kernel(RsParamSeg S, const float* __restrict__ grads, float* __restrict__ m,
                                                            float* __restrict__ v, const double* __restrict__ stats,
                                                            const rs_update_state* __restrict__ st, float lr, float thr) {
    if (st->stopped) return;
    // stats == nullptr: the (all-reduced) statistics are the (hi, lo) pairs behind the gradients
    const double kl = stats ? stats[0] : (double)grads[S.off[12]] + (double)grads[S.off[12] + 1];
    if (!(kl < (double)thr)) return;                             // kl >= 1.5 * target_kl: no step (ppo.py:1252-1261)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.off[12]) return;
    int seg = 0;
#pragma unroll
    for (int k = 1; k < 12; ++k) seg += (i >= S.off[k]) ? 1 : 0;
    float* p = S.p[seg] + (i - S.off[seg]);
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const int step = st->adam_step + 1;
    const float g = grads[i];
    const float mi = m[i] + (g - m[i]) * (1.0f - b1);            // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.0f - b2) * g * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    const float step_size = lr / bc1;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    *p = *p - step_size * (mi / denom);
}

__global__ void rs_adam_commit_kernel(const double* __restrict__ stats, const float* __restrict__ tail, rs_update_state* st, float thr) {
    if (st->stopped) return;
    st->iters += 1;
    for (int q = 0; q < 5; ++q) st->last_stats[q] = stats ? stats[q] : (double)tail[2 * q] + (double)tail[2 * q + 1];
    if (st->last_stats[0] < (double)thr) st->adam_step += 1; else st->stopped = 1;
}

extern "C" {

int rs_rollout(rs_handle* h, const rs_mlp_params* actor, const rs_mlp_params* critic, const rs_rollout_args* args,
               rs_stream_t stream) {
    if (!h || !actor || !critic || !args) return RS_ERR_INVALID_ARG;
    const RsParams& P = h->P;
    if (P.A != 1 || (P.N % 16) != 0) return RS_ERR_UNSUPPORTED;
    const bool has_obs = P.obstruction_count != 0;
    if (has_obs && P.group != 1) return RS_ERR_UNSUPPORTED;
    if (args->steps_per_epoch < 1 || args->steps_per_episode < 1) return RS_ERR_INVALID_ARG;
    size_t lds = sizeof(float) * (size_t)(rs_mlp16_lds_floats(8) + rs_mlp16_lds_floats(1)) + 16;
    lds += (has_obs ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0);
    lds += RS_WAVE * RS_OBS_DIM * 4 + RS_WAVE * 4 + 2 * RS_WAVE + rs_rollout16_mailbox_bytes();
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (has_obs) hipLaunchKernelGGL(rs_rollout16_kernel<true>, dim3(P.N / 16), dim3(2 * RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
    else hipLaunchKernelGGL(rs_rollout16_kernel<false>, dim3(P.N / 16), dim3(2 * RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

size_t rs_ppo_grad_workspace_bytes(void) {
    const size_t slabs = RS_GRAD_BLOCKS;
    return slabs * (size_t)(rs_net_params(8) + rs_net_params(1)) * sizeof(float) + 2 * slabs * 5 * sizeof(double) + 1024;
}

int rs_adam_step(const rs_mlp_params* actor, const rs_mlp_params* critic, const float* grads, float* m, float* v,
                 const double* stats, rs_update_state* state, float lr, float kl_threshold, rs_stream_t stream) {
    if (!actor || !critic || !grads || !m || !v || !state) return RS_ERR_INVALID_ARG;
    RsParamSeg S;
    const float* ptrs[12] = {actor->w1, actor->b1, actor->w2, actor->b2, actor->w3, actor->b3,
                             critic->w1, critic->b1, critic->w2, critic->b2, critic->w3, critic->b3};
    const int sizes[12] = {704, 64, 4096, 64, 512, 8, 704, 64, 4096, 64, 64, 1};
    int o = 0;
    for (int k = 0; k < 12; ++k) { S.p[k] = const_cast<float*>(ptrs[k]); S.off[k] = o; o += sizes[k]; }
    S.off[12] = o;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(rs_adam_apply_kernel, dim3((o + 255) / 256), dim3(256), 0, s, S, grads, m, v, stats, state, lr, kl_threshold);
    hipLaunchKernelGGL(rs_adam_commit_kernel, dim3(1), dim3(1), 0, s, stats, grads + o, state, kl_threshold);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_ppo_grad(const rs_mlp_params* actor, const rs_mlp_params* critic, const rs_ppo_batch* batch, float* grads,
                double* stats, void* workspace, const int32_t* stop_flag, rs_stream_t stream) {
    if (!actor || !critic || !batch || !grads || !stats || !workspace || batch->M < 1) return RS_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return RS_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int slabs = RS_GRAD_BLOCKS;                                // the 8 waves of a workgroup reduce in LDS: one slab per workgroup
    float* pa = static_cast<float*>(workspace);
    float* pc = pa + (size_t)slabs * rs_net_params(8);
    double* sa = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(pc + (size_t)slabs * rs_net_params(1)) + 255) & ~uintptr_t(255));
    double* sc = sa + (size_t)slabs * 5;
    const size_t lds_a = sizeof(float) * (size_t)rs_grad2_lds_floats(8);
    const size_t lds_c = sizeof(float) * (size_t)rs_grad2_lds_floats(1);
    // > 64 KB of dynamic LDS needs the attribute on every device the library is used on: set it per call (cheap, no sync)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rs_ppo_grad2_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(rs_ppo_grad2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess)
        return RS_ERR_HIP;
    int threads = 512;
#ifdef RS_K7_STAMPS
    // diagnostic build only: RS_K7_THREADS=256 runs ONE wave per SIMD (half of the groups are skipped: results are wrong, the
    // per-phase cycle counts are those of a wave that has its SIMD to itself)
    if (const char* e = getenv("RS_K7_THREADS")) threads = atoi(e) == 256 ? 256 : 512;
#endif
    hipLaunchKernelGGL(rs_ppo_grad2_kernel<8>, dim3(RS_GRAD_BLOCKS), dim3(threads), lds_a, s, to_dev(actor), *batch, pa, sa, stop_flag);
    hipLaunchKernelGGL(rs_ppo_grad2_kernel<1>, dim3(RS_GRAD_BLOCKS), dim3(threads), lds_c, s, to_dev(critic), *batch, pc, sc, stop_flag);
    const int np = rs_net_params(8) + rs_net_params(1);
    hipLaunchKernelGGL(rs_ppo_reduce_kernel, dim3((np + 63) / 64), dim3(1024), 0, s, pa, pc, sa, sc, slabs, grads, stats,
                       batch->alpha, batch->vf_coef, stop_flag);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

#ifdef RS_K7_STAMPS
// diagnostic build only: read (and optionally clear) the phase-cycle table of rs_ppo_grad2_kernel; out[2][16]
int rs_debug_k7_stamps(unsigned long long* out, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return RS_ERR_HIP;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rs_k7_stamp_table), sizeof(unsigned long long) * 2 * RS_K7_NPH) != hipSuccess) return RS_ERR_HIP;
    if (clear) {
        unsigned long long z[2 * RS_K7_NPH] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rs_k7_stamp_table), z, sizeof(z)) != hipSuccess) return RS_ERR_HIP;
    }
    return RS_OK;
}
#endif

int rs_policy_forward(const rs_mlp_params* actor, const rs_mlp_params* critic, const float* x, int32_t M,
                      float* logits, float* value, rs_stream_t stream) {
    if (!actor || !critic || !x || M < 1) return RS_ERR_INVALID_ARG;
    const int groups = (M + 63) / 64;
    const int grid = groups < 2048 ? groups : 2048;
    const size_t lds = sizeof(float) * (size_t)(rs_mlp_lds_floats(8) + rs_mlp_lds_floats(1));
    hipLaunchKernelGGL(rs_policy_forward_kernel, dim3(grid), dim3(64), lds, static_cast<hipStream_t>(stream), to_dev(actor),
                       to_dev(critic), x, M, logits, value);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
