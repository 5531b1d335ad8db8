// rs_cnn_loss.hip -- the PPO-clip actor loss of the RAD-TEAM update behind the logits, forward and backward in one pass.
//
// Replaces the element-wise tail of AgentPPO.compute_loss_pi (algos/multiagent/ppo.py:966-1003) that radiation_ppo_amd/ppo_cnn.py
// composed from torch ops per 524 288-sample chunk: log_softmax, gather, exp, clamp, min, the weighted sums of the loss and of the
// statistics (kl, entropy, clip fraction) and the autograd of all of it -- ~30 launches, ~1 ms per chunk-iteration of config 4's
// update (800 per PPO iteration).  One sample per lane; the derivative is K7's (csrc/rs_ppo_grad2.hpp), pinned to the reference there:
// d loss / d logits = -w * [ratio in [1 - c, 1 + c] or ratio * adv < clip(ratio) * adv] * adv * ratio * (onehot(a) - softmax).
// HBM bound: 32 B logits + 16 B scalars in, 32 B out per sample.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radsearch.h"

namespace {

constexpr int NA = 8;

__global__ void __launch_bounds__(256) rs_actor_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ act,
                                                            const float* __restrict__ adv, const float* __restrict__ lpo,
                                                            const float* __restrict__ wt, float* __restrict__ dlogits,
                                                            float* __restrict__ stats, long long S, float clip) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < S;
    const long long ic = live ? i : S - 1;
    float lg[NA];
    {
        const float4 a = *reinterpret_cast<const float4*>(logits + ic * NA), b = *reinterpret_cast<const float4*>(logits + ic * NA + 4);
        lg[0] = a.x; lg[1] = a.y; lg[2] = a.z; lg[3] = a.w; lg[4] = b.x; lg[5] = b.y; lg[6] = b.z; lg[7] = b.w;
    }
    const int a = (int)act[ic];
    const float av = adv[ic], lo_p = lpo[ic], wi = live ? wt[ic] : 0.0f;
    float mx = lg[0];
#pragma unroll
    for (int j = 1; j < NA; ++j) mx = fmaxf(mx, lg[j]);
    float se = 0.0f;
#pragma unroll
    for (int j = 0; j < NA; ++j) se += expf(lg[j] - mx);
    const float lse = logf(se);
    float pj[NA], ent = 0.0f, logp = 0.0f;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const float lp = (lg[j] - mx) - lse;
        pj[j] = expf(lp);
        ent -= pj[j] * lp;
        logp = (a == j) ? lp : logp;
    }
    const float ratio = expf(logp - lo_p);
    const float lo = 1.0f - clip, hi = 1.0f + clip;
    const float s1 = ratio * av, s2 = fminf(fmaxf(ratio, lo), hi) * av;
    const bool inside = ratio >= lo && ratio <= hi;
    const float g_lp = -wi * (((inside || s1 < s2) ? av : 0.0f) * ratio);
    if (live) {
        float d[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) d[j] = g_lp * (((a == j) ? 1.0f : 0.0f) - pj[j]);
        *reinterpret_cast<float4*>(dlogits + i * NA) = make_float4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<float4*>(dlogits + i * NA + 4) = make_float4(d[4], d[5], d[6], d[7]);
    }
    // per-wave weighted sums: kl, entropy, clip fraction, loss
    float st[4] = {wi * (lo_p - logp), wi * ent, wi * ((ratio > hi || ratio < lo) ? 1.0f : 0.0f), -wi * fminf(s1, s2)};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) st[q] += __shfl_xor(st[q], sft);
    }
    if ((threadIdx.x & 63) == 0 && live) {                       // a wave past the last sample has no row in stats
        float* so = stats + (i >> 6) * 4;
        so[0] = st[0]; so[1] = st[1]; so[2] = st[2]; so[3] = st[3];
    }
}

// The RAD-TEAM heads behind their first Linear layer, for the collector's select_action (RADTEAM_core.py:1838-1892) on every env at once:
// y1 = Linear(2704, 32)(a2) comes from the BLAS library (pre-activation); here ReLU -> Linear(32, 16) -> ReLU -> Linear(16, OUT) and
//   OUT = 8: log-softmax, the inverse-CDF draw on the env's uniform (Categorical.sample as CNNAgentPPO.act composes it: a = #{j < 7 :
//            cdf_j <= u}), log-probability of the drawn action -- the collector did this with ~14 element-wise launches per agent;
//   OUT = 1: the state value, written `copies` times at a stride (one row per agent that shares a global critic).
// One env per lane; the 16 x 32 and OUT x 16 weights are wave uniform (scalar loads).
typedef const float __attribute__((address_space(4))) * hd_cmem_t;
__device__ __forceinline__ hd_cmem_t hd_cmem(const float* p) { return (hd_cmem_t)(uintptr_t)p; }

template <int OUT>
__global__ void __launch_bounds__(64) rs_cnn_head_kernel(const float* __restrict__ y1, const float* w2_, const float* b2_, const float* w3_,
                                                         const float* b3_, const float* __restrict__ u, int us, int64_t* __restrict__ act,
                                                         float* __restrict__ logp, int8_t* __restrict__ act8, int as,
                                                         float* __restrict__ value, int copies, long long vstride,
                                                         const uint8_t* __restrict__ mask, int N) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const bool live = e < N && (mask == nullptr || mask[e] != 0);
    if (!__any(live)) return;
    const int ec = e < N ? e : N - 1;
    const hd_cmem_t w2 = hd_cmem(w2_), b2 = hd_cmem(b2_), w3 = hd_cmem(w3_), b3 = hd_cmem(b3_);
    float h1[32];
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(y1 + (size_t)ec * 32 + k);
        h1[k] = fmaxf(v.x, 0.0f); h1[k + 1] = fmaxf(v.y, 0.0f); h1[k + 2] = fmaxf(v.z, 0.0f); h1[k + 3] = fmaxf(v.w, 0.0f);
    }
    float h2[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        float s = b2[o];
#pragma unroll
        for (int k = 0; k < 32; ++k) s = __builtin_fmaf(w2[o * 32 + k], h1[k], s);
        h2[o] = fmaxf(s, 0.0f);
    }
    float lg[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        float s = b3[o];
#pragma unroll
        for (int k = 0; k < 16; ++k) s = __builtin_fmaf(w3[o * 16 + k], h2[k], s);
        lg[o] = s;
    }
    if (OUT == 1) {
        if (live)
            for (int c = 0; c < copies; ++c) value[(size_t)c * vstride + e] = lg[0];
        return;
    }
    float mx = lg[0];
#pragma unroll
    for (int o = 1; o < OUT; ++o) mx = fmaxf(mx, lg[o]);
    float se = 0.0f;
#pragma unroll
    for (int o = 0; o < OUT; ++o) se += expf(lg[o] - mx);
    const float lse = logf(se);
    const float uu = u[(size_t)ec * us];
    float cdf = 0.0f, lp_sel = (lg[0] - mx) - lse;
    int a = 0;
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        const float lp = (lg[o] - mx) - lse;
        cdf += expf(lp);
        if (o < OUT - 1 && cdf <= uu) a = o + 1;
    }
#pragma unroll
    for (int o = 1; o < OUT; ++o) if (a == o) lp_sel = (lg[o] - mx) - lse;
    if (live) {
        if (act) act[e] = a;
        if (logp) logp[e] = lp_sel;
        if (act8) act8[(size_t)e * as] = (int8_t)a;
    }
}

}  // namespace

extern "C" {

int rs_cnn_head(const float* y1, const float* w2, const float* b2, const float* w3, const float* b3, int32_t out_dim, const float* u,
                int32_t u_stride, int64_t* act, float* logp, int8_t* act8, int32_t act8_stride, float* value, int32_t value_copies,
                int64_t value_stride, const uint8_t* mask, int32_t num_envs, rs_stream_t stream) {
    if (!y1 || !w2 || !b2 || !w3 || !b3 || num_envs < 1 || (out_dim != 8 && out_dim != 1)) return RS_ERR_INVALID_ARG;
    if (out_dim == 8 && (!u || u_stride < 1 || act8_stride < 1)) return RS_ERR_INVALID_ARG;
    if (out_dim == 1 && (!value || value_copies < 1)) return RS_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dim == 8)
        hipLaunchKernelGGL(rs_cnn_head_kernel<8>, dim3((num_envs + 63) / 64), dim3(64), 0, s, y1, w2, b2, w3, b3, u, u_stride, act, logp, act8,
                           act8_stride, value, value_copies, (long long)value_stride, mask, num_envs);
    else
        hipLaunchKernelGGL(rs_cnn_head_kernel<1>, dim3((num_envs + 63) / 64), dim3(64), 0, s, y1, w2, b2, w3, b3, u, u_stride, act, logp, act8,
                           act8_stride, value, value_copies, (long long)value_stride, mask, num_envs);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_actor_loss(const float* logits, const int64_t* act, const float* adv, const float* logp_old, const float* sample_weight,
                  float* dlogits, float* stats, int64_t samples, double clip_ratio, rs_stream_t stream) {
    if (!logits || !act || !adv || !logp_old || !sample_weight || !dlogits || !stats || samples < 1) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_actor_loss_kernel, dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       act, adv, logp_old, sample_weight, dlogits, stats, (long long)samples, (float)clip_ratio);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
