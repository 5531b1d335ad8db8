// rs_ppo.hip -- policy-side kernels: MFMA forward of the FF_core MLP, the fused on-device collector
// (rs_rollout) and the fused PPO loss/gradient pass (rs_ppo_grad).  See rs_mlp.hpp for the MFMA mapping.
#include <hip/hip_runtime.h>
#include <stdint.h>
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
// K6: fused collector.  One wave = 64 envs for the whole epoch; weights in LDS in fragment order.
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

template <bool HAS_OBS>
__global__ void __launch_bounds__(64) rs_rollout_kernel(RsParams P, RsMlpParams pa, RsMlpParams pc, rs_rollout_args R) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* wts = reinterpret_cast<float*>(smem);
    unsigned char* p = smem + sizeof(float) * (size_t)(rs_mlp_lds_floats(8) + rs_mlp_lds_floats(1));
    p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~uintptr_t(15));
    int* lds_geo = reinterpret_cast<int*>(p);
    uint32_t* lds_adj = reinterpret_cast<uint32_t*>(p + RS_MAX_VERT * RS_WAVE * 4);
    double* lds_d = reinterpret_cast<double*>(p + 2 * RS_MAX_VERT * RS_WAVE * 4);
    float* tile = reinterpret_cast<float*>(p + (HAS_OBS ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0));
    float* lds_rew = tile + RS_WAVE * RS_OBS_DIM;                       // [64]
    uint8_t* lds_done = reinterpret_cast<uint8_t*>(lds_rew + RS_WAVE);  // [64]
    uint8_t* lds_oob = lds_done + RS_WAVE;                              // [64]

    RsMlpLds<8> ACT; RsMlpLds<1> CRT;
    ACT.carve(wts);
    CRT.carve(wts + rs_mlp_lds_floats(8));
    ACT.fill(pa); CRT.fill(pc);

    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * RS_WAVE + lane;          // N % 64 == 0 (checked by the host)
    const int N = P.N, T = R.steps_per_epoch, L = R.steps_per_episode;
    RsGeo g{lds_geo, 0, 0, 0};
    if (HAS_OBS) rs_load_geo(P, n, true, lds_geo, g);
    __syncthreads();

    // ---- carried collector state
    float oraw[RS_OBS_DIM];
#pragma unroll
    for (int k = 0; k < RS_OBS_DIM; ++k) oraw[k] = R.cur_obs[(size_t)n * RS_OBS_DIM + k];
    RsWelford W{R.w_count[n], R.w_mean[n], R.w_sq[n], R.w_std[n]};
    int steps = R.steps_in_ep[n];
    float ep_ret = R.ep_ret[n];
    int done_count = 0, oob_count = 0, ep_count = 0;
    double ep_ret_sum = 0.0, ep_len_sum = 0.0;
    const uint32_t k0 = P.seed, k1 = P.env_id_base + (uint32_t)n;

    float xo[RS_IN_PAD], xp[RS_IN_PAD];
#pragma unroll
    for (int k = 0; k < RS_OBS_DIM; ++k) xo[k] = oraw[k];
    xo[0] = W.standardize(oraw[0]);
    xo[11] = 0.0f;
    rs_exchange_x(xo, xp);
    float v;
    { float vv[1]; rs_mlp_forward<1>(CRT, xo, xp, vv); v = vv[0]; }

    RsOut O;
    O.obs_row = tile + lane * RS_OBS_DIM;
    O.reward = lds_rew + lane - (size_t)n;              // O.reward[n*A + 0] == lds_rew[lane]
    O.team = nullptr;
    O.done = lds_done + lane - (size_t)n;
    O.oob = lds_oob + lane - (size_t)n;
    O.oobc = nullptr; O.blocked = nullptr; O.collision = nullptr;

    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * N + n;
        // ---- actor forward + inverse-CDF sampling (FF_core.py:95-107 with the Philox uniform of rs_action_uniforms)
        float lg[8];
        rs_mlp_forward<8>(ACT, xo, xp, lg);
        float mx = lg[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, lg[j]);
        float se = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) se += __expf(lg[j] - mx);
        const float lse = __logf(se);
        const uint32_t episode = P.episode[n] - 1u, tenv = P.tstep[n];
        u32x4 ph = philox4x32_10(0u, tenv, episode, RS_STREAM_ACT, k0, k1);
        const float u = (float)(ph.x >> 8) * (1.0f / 16777216.0f);
        int a = 0;
        float cdf = 0.0f, logp = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lpj = (lg[j] - mx) - lse;
            cdf += __expf(lpj);
            if (j < 7) a += (cdf <= u) ? 1 : 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) logp = (a == j) ? ((lg[j] - mx) - lse) : logp;
        // ---- buffer row (PPOBuffer.store, ppo.py:339-381)
#pragma unroll
        for (int k = 0; k < RS_OBS_DIM; ++k) tile[lane * RS_OBS_DIM + k] = xo[k];
        __syncthreads();
        {
            float* dst = R.obs + ((size_t)t * N + (size_t)blockIdx.x * RS_WAVE) * RS_OBS_DIM;
#pragma unroll
            for (int i = 0; i < RS_OBS_DIM; ++i) dst[i * RS_WAVE + lane] = tile[i * RS_WAVE + lane];
        }
        __syncthreads();
        R.act[row] = (int64_t)a;
        R.logp[row] = logp;
        R.val[row] = v;
        R.source_tar[row * 2 + 0] = (float)P.src_x[n];
        R.source_tar[row * 2 + 1] = (float)P.src_y[n];
        // ---- env step (train.py:361-363)
        rs_env_step_lane<HAS_OBS>(P, g, n, [&](int) -> int { return a; }, O);
        const float r = lds_rew[lane];
        const bool terminal = lds_done[lane] != 0;
        oob_count += lds_oob[lane];
        R.rew[row] = r;
        ep_ret += r;
        steps += 1;
        done_count += terminal ? 1 : 0;
        const bool timeout = steps == L;                              // train.py:394-405
        const bool over = terminal || timeout;
        const bool ended = t == T - 1;
        const bool cut = over || ended;
        const bool boot = timeout || ended;
#pragma unroll
        for (int k = 0; k < RS_OBS_DIM; ++k) oraw[k] = tile[lane * RS_OBS_DIM + k];
        W.update((double)oraw[0]);                                    // train.py:432-436
#pragma unroll
        for (int k = 1; k < RS_OBS_DIM; ++k) xo[k] = oraw[k];
        xo[0] = W.standardize(oraw[0]);
        rs_exchange_x(xo, xp);
        float vb;
        { float vv[1]; rs_mlp_forward<1>(CRT, xo, xp, vv); vb = vv[0]; }   // bootstrap value / next step's value
        R.cut[row] = cut ? 1 : 0;
        R.last_val[row] = (cut && boot) ? vb : 0.0f;                  // train.py:462-487
        if (over) { ep_ret_sum += (double)ep_ret; ep_len_sum += (double)steps; ep_count += 1; }
        if (cut) {
            if (ended) P.epoch_end[n] = 1;                            // train.py:482-484
            W.reset();                                                // train.py:504-509
            rs_env_reset_lane<HAS_OBS>(P, g, n, lds_geo, lds_adj, lds_d, tile + lane * RS_OBS_DIM, O);   // train.py:530
#pragma unroll
            for (int k = 0; k < RS_OBS_DIM; ++k) oraw[k] = tile[lane * RS_OBS_DIM + k];
            W.update((double)oraw[0]);                                // train.py:542-548
#pragma unroll
            for (int k = 1; k < RS_OBS_DIM; ++k) xo[k] = oraw[k];
            xo[0] = W.standardize(oraw[0]);
            steps = 0;
            ep_ret = 0.0f;
        }
        v = vb;
        if (__ballot(cut) != 0ull) {                                  // wave-uniform: some env restarted
            rs_exchange_x(xo, xp);
            float vv[1];
            rs_mlp_forward<1>(CRT, xo, xp, vv);
            v = cut ? vv[0] : vb;
        }
    }
    // ---- carry state to the next launch
#pragma unroll
    for (int k = 0; k < RS_OBS_DIM; ++k) R.cur_obs[(size_t)n * RS_OBS_DIM + k] = oraw[k];
    R.w_count[n] = W.count; R.w_mean[n] = W.mean; R.w_sq[n] = W.sq; R.w_std[n] = W.std;
    R.steps_in_ep[n] = steps;
    R.ep_ret[n] = ep_ret;
    R.done_count[n] = done_count; R.oob_count[n] = oob_count; R.ep_count[n] = ep_count;
    R.ep_ret_sum[n] = ep_ret_sum; R.ep_len_sum[n] = ep_len_sum;
}

extern "C" {

int rs_rollout(rs_handle* h, const rs_mlp_params* actor, const rs_mlp_params* critic, const rs_rollout_args* args,
               rs_stream_t stream) {
    if (!h || !actor || !critic || !args) return RS_ERR_INVALID_ARG;
    const RsParams& P = h->P;
    if (P.A != 1 || (P.N % RS_WAVE) != 0) return RS_ERR_UNSUPPORTED;
    const bool has_obs = P.obstruction_count != 0;
    if (has_obs && P.group != 1) return RS_ERR_UNSUPPORTED;
    if (args->steps_per_epoch < 1 || args->steps_per_episode < 1) return RS_ERR_INVALID_ARG;
    size_t lds = sizeof(float) * (size_t)(rs_mlp_lds_floats(8) + rs_mlp_lds_floats(1)) + 16;
    lds += (has_obs ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0);
    lds += RS_WAVE * RS_OBS_DIM * 4 + RS_WAVE * 4 + 2 * RS_WAVE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (has_obs) hipLaunchKernelGGL(rs_rollout_kernel<true>, dim3(P.N / RS_WAVE), dim3(RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
    else hipLaunchKernelGGL(rs_rollout_kernel<false>, dim3(P.N / RS_WAVE), dim3(RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

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
