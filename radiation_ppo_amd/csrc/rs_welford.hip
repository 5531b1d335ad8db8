// rs_welford.hip -- the running observation statistics of the collectors as three one-pass kernels, and their per-epoch logger
// statistics as one.
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

// The per-epoch logger statistics of the collectors (train.py:386-398, :494-501, :519-526; ppo.py: EpochStats) for one lock-step:
// out-of-bounds and terminal counts per agent id, and count / length / sum / sum of squares / max / min of the returns of the
// episodes that ended.  One workgroup, fixed summation order (lane-private sums over n = lane, lane + 256, ..., then a tree).
constexpr int ES_MAXA = RS_MAX_AGENTS;
__device__ __forceinline__ double es_wave_sum(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}
__device__ __forceinline__ double es_wave_max(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmax(v, __shfl_xor(v, s));
    return v;
}
// ONE pass over the envs: every thread keeps its lane-private partial of all 6 A + 2 quantities (n = tid, tid + 256, ...), then each
// quantity goes through a butterfly over the wave and a fixed-order sum of the four wave partials.  (Round 2 made one pass over the
// envs and one nine-barrier tree per quantity: 40 us per lock-step at 4096 envs, 11 % of the RAD-A2C collector.)
__global__ void __launch_bounds__(256) rs_epoch_stats_kernel(const uint8_t* __restrict__ oob, const uint8_t* __restrict__ done,
                                                             const float* __restrict__ ep_ret, const int32_t* __restrict__ steps,
                                                             const uint8_t* __restrict__ over, double* __restrict__ acc_oob,
                                                             double* __restrict__ acc_done, double* __restrict__ ep_cnt, double* __restrict__ ep_len,
                                                             double* __restrict__ ret_sum, double* __restrict__ ret_sq, double* __restrict__ ret_max,
                                                             double* __restrict__ ret_min, int N, int A) {
    __shared__ double red[6 * ES_MAXA + 2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v_oob[ES_MAXA], v_done[ES_MAXA], v_sum[ES_MAXA], v_sq[ES_MAXA], v_max[ES_MAXA], v_min[ES_MAXA], v_cnt = 0.0, v_len = 0.0;
#pragma unroll
    for (int a = 0; a < ES_MAXA; ++a) { v_oob[a] = 0.0; v_done[a] = 0.0; v_sum[a] = 0.0; v_sq[a] = 0.0; v_max[a] = -INFINITY; v_min[a] = INFINITY; }
    for (int n = tid; n < N; n += 256) {
        const bool ov = over[n] != 0;
        if (ov) { v_cnt += 1.0; v_len += (double)steps[n]; }
#pragma unroll
        for (int a = 0; a < ES_MAXA; ++a) {
            if (a < A) {
                const double r = (double)ep_ret[(size_t)n * A + a];
                v_oob[a] += (double)oob[(size_t)n * A + a];
                v_done[a] += (double)done[(size_t)n * A + a];
                if (ov) { v_sum[a] += r; v_sq[a] += r * r; v_max[a] = fmax(v_max[a], r); v_min[a] = fmin(v_min[a], r); }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < ES_MAXA; ++a) {
        if (a < A) {
            const double t0 = es_wave_sum(v_oob[a]), t1 = es_wave_sum(v_done[a]), t2 = es_wave_sum(v_sum[a]), t3 = es_wave_sum(v_sq[a]);
            const double t4 = es_wave_max(v_max[a]), t5 = -es_wave_max(-v_min[a]);
            if (lane == 0) {
                red[0 * ES_MAXA + a][wave] = t0; red[1 * ES_MAXA + a][wave] = t1; red[2 * ES_MAXA + a][wave] = t2;
                red[3 * ES_MAXA + a][wave] = t3; red[4 * ES_MAXA + a][wave] = t4; red[5 * ES_MAXA + a][wave] = t5;
            }
        }
    }
    {
        const double tc = es_wave_sum(v_cnt), tl = es_wave_sum(v_len);
        if (lane == 0) { red[6 * ES_MAXA][wave] = tc; red[6 * ES_MAXA + 1][wave] = tl; }
    }
    __syncthreads();
    if (tid < 6 * ES_MAXA + 2) {
        const int kind = tid < 6 * ES_MAXA ? tid / ES_MAXA : 6 + (tid - 6 * ES_MAXA), a = tid < 6 * ES_MAXA ? tid % ES_MAXA : 0;
        if (kind >= 6 || a < A) {
            const double* r = red[tid];
            switch (kind) {
                case 0: acc_oob[a] += (r[0] + r[1]) + (r[2] + r[3]); break;
                case 1: acc_done[a] += (r[0] + r[1]) + (r[2] + r[3]); break;
                case 2: ret_sum[a] += (r[0] + r[1]) + (r[2] + r[3]); break;
                case 3: ret_sq[a] += (r[0] + r[1]) + (r[2] + r[3]); break;
                case 4: ret_max[a] = fmax(ret_max[a], fmax(fmax(r[0], r[1]), fmax(r[2], r[3]))); break;
                case 5: ret_min[a] = fmin(ret_min[a], fmin(fmin(r[0], r[1]), fmin(r[2], r[3]))); break;
                case 6: ep_cnt[0] += (r[0] + r[1]) + (r[2] + r[3]); break;
                default: ep_len[0] += (r[0] + r[1]) + (r[2] + r[3]); break;
            }
        }
    }
}

// One lock-step's rows of the rollout buffer (PPOBuffer.store, algos/multiagent/ppo.py:296-389, as the RAD-A2C collector fills it):
// action, log-probability, value, observation, source location, reward, cut flag and bootstrap value of every (env, agent) go to
// row *t of the time-major buffers in one launch (the torch collector did ~19: staging copies, index_copy per column, conversions).
struct StoreArgs {
    const int64_t* t;        // [1] device-side step counter
    const int64_t* act;      // [A][N]
    const float* f;          // [A][3][N]: logp, value, bootstrap value
    const float* x;          // [N][A][11]
    const int32_t* src_x;    // [N]
    const int32_t* src_y;    // [N]
    const float* rew;        // [N][A]
    const uint8_t* cut;      // [N]
    const uint8_t* boot;     // [N]
    int64_t* b_act;          // [T][N][A]
    float* b_logp; float* b_val; float* b_last; float* b_obs; float* b_src; float* b_rew;
    uint8_t* b_cut;
    int N, A, T;
};

__global__ void __launch_bounds__(256) rs_store_rows_kernel(StoreArgs a_) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a_.N, A = a_.A;
    if (i >= N * A) return;
    const long long t = a_.t[0];
    if (t < 0 || t >= a_.T) return;
    const int n = i / A, a = i - n * A;
    const size_t row = ((size_t)t * N + n) * A + a;
    a_.b_act[row] = a_.act[(size_t)a * N + n];
    a_.b_logp[row] = a_.f[((size_t)a * 3 + 0) * N + n];
    a_.b_val[row] = a_.f[((size_t)a * 3 + 1) * N + n];
    a_.b_last[row] = a_.boot[n] ? a_.f[((size_t)a * 3 + 2) * N + n] : 0.0f;
    a_.b_rew[row] = a_.rew[i];
    a_.b_cut[row] = a_.cut[n] ? 1 : 0;
#pragma unroll
    for (int k = 0; k < RS_OBS_DIM; ++k) a_.b_obs[row * RS_OBS_DIM + k] = a_.x[(size_t)i * RS_OBS_DIM + k];
    if (a == 0) {
        a_.b_src[((size_t)t * N + n) * 2] = (float)a_.src_x[n];
        a_.b_src[((size_t)t * N + n) * 2 + 1] = (float)a_.src_y[n];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The element-wise bookkeeping of a collector lock-step (train.py:332-548 between the library calls), three launches instead of
// ~30 (each ~4.6 us inside the replayed graph: half of the RAD-A2C collector's lock-step).  One thread per env; the Welford arithmetic
// is rs_welford_update_kernel's, operation by operation.
__device__ __forceinline__ void cs_welford_update(const rs_collect_state& c, int i, double x) {
    const double cnt = c.w_count[i] + 1.0, m = c.w_mean[i];
    c.w_count[i] = cnt;
    if (cnt == 1.0) { c.w_mean[i] = x; return; }
    const double mn = m + (x - m) / cnt;
    const double s = c.w_sq[i] + (x - m) * (x - mn);
    c.w_mean[i] = mn;
    c.w_sq[i] = s;
    c.w_std[i] = fmax(sqrt(s / fmax(cnt - 1.0, 1.0)), 1.0);
}

__device__ __forceinline__ void cs_standardized_row(const rs_collect_state& c, int i, const float* __restrict__ src, float* __restrict__ dst) {
#pragma unroll
    for (int k = 1; k < RS_OBS_DIM; ++k) dst[k] = src[k];
    dst[0] = c.w_count ? (float)(((double)src[0] - c.w_mean[i]) / c.w_std[i]) : src[0];
}

// x <- obs with the reading standardised by the running statistics (train.py:334-341)
__global__ void __launch_bounds__(256) rs_collect_pre_kernel(rs_collect_state c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.num_envs * c.num_agents) return;
    cs_standardized_row(c, i, c.obs + (size_t)i * RS_OBS_DIM, c.x + (size_t)i * RS_OBS_DIM);
}

// after rs_step: returns, step counters, episode / epoch cut flags (train.py:366-405), Welford update with the new readings (:432-436),
// the new observation and its standardised form for the bootstrap value (:462-480); the predictor's call counter advances
__global__ void __launch_bounds__(256) rs_collect_post_step_kernel(rs_collect_state c, int epoch_ended) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= c.num_envs) return;
    const int A = c.num_agents;
    bool terminal = false;
    for (int a = 0; a < A; ++a) {
        const int i = n * A + a;
        const float r = c.team_reward ? c.env_team[n] : c.env_reward[i];
        if (c.reward_used) c.reward_used[i] = r;
        c.ep_ret[i] += r;
        terminal = terminal || c.env_done[i] != 0;
        if (c.done_copy) { c.done_copy[i] = c.env_done[i]; c.oob_copy[i] = c.env_oob[i]; }
    }
    if (c.done_copy) { c.src_copy[n] = c.env_src_x[n]; c.src_copy[c.num_envs + n] = c.env_src_y[n]; }
    const int st = c.steps_in_ep[n] + 1;
    c.steps_in_ep[n] = st;
    const bool timeout = st == c.steps_per_episode;
    const bool over = terminal || timeout;
    const bool cut = epoch_ended ? true : over;
    c.over[n] = over ? 1 : 0;
    c.cut[n] = cut ? 1 : 0;
    c.boot[n] = (epoch_ended ? cut : timeout) ? 1 : 0;
    if (c.complete_len && over) c.complete_len[n] = c.t[0] + 1;
    for (int a = 0; a < A; ++a) {
        const int i = n * A + a;
        const float* src = c.env_obs + (size_t)i * RS_OBS_DIM;
        float* dst = c.obs + (size_t)i * RS_OBS_DIM;
        if (c.w_count) cs_welford_update(c, i, (double)src[0]);
#pragma unroll
        for (int k = 0; k < RS_OBS_DIM; ++k) dst[k] = src[k];
        if (c.xb) cs_standardized_row(c, i, src, c.xb + (size_t)i * RS_OBS_DIM);
    }
    if (c.pf_calls) c.pf_calls[n] += 1;
}

// after rs_reset(cut): the envs that were cut start an episode -- fresh observation, zero return / step count, restarted statistics
// with the first reading (train.py:504-548), new draw counters for the predictor bank and the GRU's initial state (reset_hidden);
// the device-side step counter advances
__global__ void __launch_bounds__(256) rs_collect_post_reset_kernel(rs_collect_state c, int reset_hidden) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && c.t) c.t[0] += 1;
    if (n >= c.num_envs) return;
    const int A = c.num_agents;
    if (c.pf_calls && c.boot[n]) c.pf_calls[n] += 1;            // the bootstrap round's prediction (masked to these envs)
    if (!c.cut[n]) return;
    for (int a = 0; a < A; ++a) {
        const int i = n * A + a;
        const float* src = c.env_obs + (size_t)i * RS_OBS_DIM;
        float* dst = c.obs + (size_t)i * RS_OBS_DIM;
#pragma unroll
        for (int k = 0; k < RS_OBS_DIM; ++k) dst[k] = src[k];
        c.ep_ret[i] = 0.0f;
        if (c.w_count) { c.w_count[i] = 1.0; c.w_mean[i] = (double)src[0]; c.w_sq[i] = 0.0; c.w_std[i] = 1.0; }     // reset + first update
    }
    c.steps_in_ep[n] = 0;
    if (reset_hidden) {
        if (c.pf_episode) c.pf_episode[n] += 1;
        if (c.pf_calls) c.pf_calls[n] = 0;
        if (c.episodes_begun) c.episodes_begun[n] += 1;
    }
}

}  // namespace

extern "C" {

static bool collect_ok(const rs_collect_state* c) {
    return c && c->num_envs >= 1 && c->num_agents >= 1 && c->num_agents <= RS_MAX_AGENTS && c->obs && c->env_obs && c->env_reward && c->env_done &&
           c->ep_ret && c->steps_in_ep && c->over && c->cut && c->boot && (!c->team_reward || c->env_team) &&
           ((c->w_count != nullptr) == (c->w_mean != nullptr)) && ((c->w_count != nullptr) == (c->w_sq != nullptr)) &&
           ((c->w_count != nullptr) == (c->w_std != nullptr)) &&
           ((c->done_copy != nullptr) == (c->oob_copy != nullptr)) && ((c->done_copy != nullptr) == (c->src_copy != nullptr)) &&
           (c->done_copy == nullptr || (c->env_oob && c->env_src_x && c->env_src_y));
}

int rs_collect_pre(const rs_collect_state* c, rs_stream_t stream) {
    if (!collect_ok(c) || !c->x) return RS_ERR_INVALID_ARG;
    const int M = c->num_envs * c->num_agents;
    hipLaunchKernelGGL(rs_collect_pre_kernel, dim3((M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *c);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_collect_post_step(const rs_collect_state* c, int32_t epoch_ended, rs_stream_t stream) {
    if (!collect_ok(c) || c->steps_per_episode < 1) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_collect_post_step_kernel, dim3((c->num_envs + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *c,
                       epoch_ended ? 1 : 0);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_collect_post_reset(const rs_collect_state* c, int32_t reset_hidden, rs_stream_t stream) {
    if (!collect_ok(c)) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_collect_post_reset_kernel, dim3((c->num_envs + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), *c,
                       reset_hidden ? 1 : 0);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_store_rows(const int64_t* t, const int64_t* act, const float* logp_val_boot, const float* x, const int32_t* src_x, const int32_t* src_y,
                  const float* rew, const uint8_t* cut, const uint8_t* boot, int64_t* buf_act, float* buf_logp, float* buf_val,
                  float* buf_last_val, float* buf_obs, float* buf_source, float* buf_rew, uint8_t* buf_cut, int32_t num_envs,
                  int32_t num_agents, int32_t steps_per_epoch, rs_stream_t stream) {
    if (!t || !act || !logp_val_boot || !x || !src_x || !src_y || !rew || !cut || !boot || !buf_act || !buf_logp || !buf_val || !buf_last_val ||
        !buf_obs || !buf_source || !buf_rew || !buf_cut || num_envs < 1 || num_agents < 1 || steps_per_epoch < 1)
        return RS_ERR_INVALID_ARG;
    StoreArgs a{t, act, logp_val_boot, x, src_x, src_y, rew, cut, boot, buf_act, buf_logp, buf_val, buf_last_val, buf_obs, buf_source, buf_rew,
                buf_cut, num_envs, num_agents, steps_per_epoch};
    const int M = num_envs * num_agents;
    hipLaunchKernelGGL(rs_store_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_epoch_stats(const uint8_t* out_of_bounds, const uint8_t* done, const float* ep_ret, const int32_t* steps_in_ep, const uint8_t* over,
                   double* acc_oob, double* acc_done, double* ep_cnt, double* ep_len, double* ret_sum, double* ret_sq, double* ret_max,
                   double* ret_min, int32_t num_envs, int32_t num_agents, rs_stream_t stream) {
    if (!out_of_bounds || !done || !ep_ret || !steps_in_ep || !over || !acc_oob || !acc_done || !ep_cnt || !ep_len || !ret_sum || !ret_sq ||
        !ret_max || !ret_min || num_envs < 1 || num_agents < 1 || num_agents > ES_MAXA)
        return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_epoch_stats_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), out_of_bounds, done, ep_ret, steps_in_ep,
                       over, acc_oob, acc_done, ep_cnt, ep_len, ret_sum, ret_sq, ret_max, ret_min, num_envs, num_agents);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

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
