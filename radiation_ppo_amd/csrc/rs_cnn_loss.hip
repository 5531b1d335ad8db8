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

}  // namespace

extern "C" {

int rs_actor_loss(const float* logits, const int64_t* act, const float* adv, const float* logp_old, const float* sample_weight,
                  float* dlogits, float* stats, int64_t samples, double clip_ratio, rs_stream_t stream) {
    if (!logits || !act || !adv || !logp_old || !sample_weight || !dlogits || !stats || samples < 1) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_actor_loss_kernel, dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       act, adv, logp_old, sample_weight, dlogits, stats, (long long)samples, (float)clip_ratio);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
