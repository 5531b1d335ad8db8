// rs_welford.hip -- the running observation statistics of the collectors as three one-pass kernels.
//
// Replaces the element-wise composition of StatisticStandardization (NeuralNetworkCores/RADTEAM_core.py:188-277; StatBuff in
// RADA2C_core.py) in radiation_ppo_amd/ppo.py: DeviceWelford -- ~25 float64 element-wise launches per update, ~60 per lock-step
// of the RAD-A2C collector (two updates, a reset, two standardisations), half of its launches.  One stream per (env, agent);
// float64 state; the arithmetic is the reference's, operation by operation (the build disables FMA contraction), so the
// results equal the torch composition bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radsearch.h"

namespace {

__global__ void __launch_bounds__(256) rs_welford_update_kernel(double* __restrict__ count, double* __restrict__ mean, double* __restrict__ sq,
                                                                double* __restrict__ sd, const float* __restrict__ reading, long long stride,
                                                                const uint8_t* __restrict__ mask, int M, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M || (mask && !mask[i / A])) return;
    const double x = (double)reading[(long long)i * stride];
    const double c = count[i] + 1.0, m = mean[i];
    count[i] = c;
    if (c == 1.0) {                                         // first sample: mean = x, sq and std stay (:237-240)
        mean[i] = x;
        return;
    }
    const double mn = m + (x - m) / c;
    const double s = sq[i] + (x - m) * (x - mn);
    mean[i] = mn;
    sq[i] = s;
    sd[i] = fmax(sqrt(s / fmax(c - 1.0, 1.0)), 1.0);
}

__global__ void __launch_bounds__(256) rs_welford_reset_kernel(double* __restrict__ count, double* __restrict__ mean, double* __restrict__ sq,
                                                               double* __restrict__ sd, const uint8_t* __restrict__ mask, int M, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M || (mask && !mask[i / A])) return;
    count[i] = 0.0; mean[i] = 0.0; sq[i] = 0.0; sd[i] = 1.0;
}

__global__ void __launch_bounds__(256) rs_welford_standardize_kernel(const double* __restrict__ mean, const double* __restrict__ sd,
                                                                     const float* __restrict__ reading, long long stride,
                                                                     float* __restrict__ out, long long out_stride, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    out[(long long)i * out_stride] = (float)(((double)reading[(long long)i * stride] - mean[i]) / sd[i]);
}

}  // namespace

extern "C" {

int rs_welford_update(double* count, double* mean, double* sq, double* std, const float* reading, int64_t stride, const uint8_t* mask,
                      int32_t num_envs, int32_t num_agents, rs_stream_t stream) {
    if (!count || !mean || !sq || !std || !reading || num_envs < 1 || num_agents < 1) return RS_ERR_INVALID_ARG;
    const int M = num_envs * num_agents;
    hipLaunchKernelGGL(rs_welford_update_kernel, dim3((M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), count, mean, sq, std,
                       reading, (long long)stride, mask, M, num_agents);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_welford_reset(double* count, double* mean, double* sq, double* std, const uint8_t* mask, int32_t num_envs, int32_t num_agents,
                     rs_stream_t stream) {
    if (!count || !mean || !sq || !std || num_envs < 1 || num_agents < 1) return RS_ERR_INVALID_ARG;
    const int M = num_envs * num_agents;
    hipLaunchKernelGGL(rs_welford_reset_kernel, dim3((M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), count, mean, sq, std,
                       mask, M, num_agents);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_welford_standardize(const double* mean, const double* std, const float* reading, int64_t stride, float* out, int64_t out_stride,
                           int32_t streams, rs_stream_t stream) {
    if (!mean || !std || !reading || !out || streams < 1) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_welford_standardize_kernel, dim3((streams + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), mean, std,
                       reading, (long long)stride, out, (long long)out_stride, streams);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
