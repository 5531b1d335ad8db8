// rs_gru.hip -- K12: the recurrence of the RAD-A2C GRU over an episode-major batch, forward and
// back-propagation through time.
//
// Replaces the time loop of `self.seq_model(x, hidden)` in SeqPt.forward (NeuralNetworkCores/RADA2C_core.py:377-381, a
// torch.nn.GRU(13, 24, 1) run over whole episodes by grad_step :550-566) and of its autograd.  What is NOT sequential stays in
// library GEMMs on the host side (radiation_ppo_amd/rada2c.py: GRUSequence): the input projection gi = X W_ih^T + b_ih for all
// (t, episode) at once, and the four weight gradients, which are sums over (t, episode) of outer products of what these kernels
// write.  torch.nn.GRU semantics (gate order r, z, n):
//     r = sigmoid(gi_r + gh_r)   z = sigmoid(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   h' = (1 - z) * n + z * h,   gh = W_hh h + b_hh
//
// Mapping: one episode per lane, the 24 hidden units of its state in registers; per step a 72 x 24 (forward) or 24 x 72
// (backward) product whose weights are wave-uniform and arrive through the scalar unit (s_load_dwordx16 -> SGPR operand of
// v_fma), exactly as in K11 (rs_pfgru.hip) -- including its two remedies for the scalar-weight code generation (row blocks
// closed by a scheduling barrier, results pinned).  The chain is 120 steps of ~1 900 dependent-ish FMAs per lane: latency
// bound (E / 64 waves), ~1 ms per pass at 4 400 episodes, against ~45 ms for the library GRU's per-step launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radsearch.h"

namespace {

constexpr int GH = RS_GRU_HIDDEN;            // 24
constexpr int G3 = 3 * GH;                   // 72

typedef const float __attribute__((address_space(4))) * cmem_t;
__device__ __forceinline__ cmem_t as_cmem(const float* p) { return (cmem_t)(uintptr_t)p; }

// out[OUTP] += W^T c for a k-major [K][OUTP] block read through the scalar unit (OUTP a multiple of 16), one k row per block
template <int K, int OUTP, typename F>
__device__ __forceinline__ void gru_matvec(cmem_t W, F cval, float (&out)[OUTP]) {
#pragma unroll
    for (int ch = 0; ch < OUTP / 16; ++ch) {
        float acc[16], wa[16], wb[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) { acc[o] = out[16 * ch + o]; wa[o] = W[16 * ch + o]; }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (k + 1 < K) {
#pragma unroll
                for (int o = 0; o < 16; ++o) wb[o] = W[(k + 1) * OUTP + 16 * ch + o];
            }
            const float c = cval(k);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fmaf(wa[o], c, acc[o]);
            if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int o = 0; o < 16; ++o) wa[o] = wb[o];
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            asm volatile("" : "+v"(acc[o]));
            out[16 * ch + o] = acc[o];
        }
    }
}

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// forward: gi [L][E][72], h0 [E][24], whh_t [24][80] (k-major W_hh^T, columns 72..79 zero), bhh [80]
//          -> hs [L][E][24] (h_t), gates [L][E][96] = r | z | n | (W_hn h + b_hn)
__global__ void __launch_bounds__(64) rs_gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ h0, const float* __restrict__ whh_t,
                                                        const float* __restrict__ bhh, float* __restrict__ hs, float* __restrict__ gates,
                                                        int L, int E) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int ec = e < E ? e : E - 1;                      // idle lanes shadow the last episode, store nothing
    const cmem_t W = as_cmem(whh_t), B = as_cmem(bhh);
    float h[GH];
#pragma unroll
    for (int j = 0; j < GH; ++j) h[j] = h0[(size_t)ec * GH + j];
    for (int t = 0; t < L; ++t) {
        float gh[80];
#pragma unroll
        for (int o = 0; o < 80; ++o) gh[o] = B[o];
        gru_matvec<GH, 80>(W, [&](int k) -> float { return h[k]; }, gh);
        const float* g = gi + ((size_t)t * E + ec) * G3;
        float* go = gates + ((size_t)t * E + ec) * (4 * GH);
        float* ho = hs + ((size_t)t * E + ec) * GH;
#pragma unroll
        for (int j = 0; j < GH; ++j) {
            const float r = gru_sigmoid(g[j] + gh[j]);
            const float z = gru_sigmoid(g[GH + j] + gh[GH + j]);
            const float hn = gh[2 * GH + j];
            const float n = tanhf(g[2 * GH + j] + r * hn);
            const float hv = (1.0f - z) * n + z * h[j];
            if (e < E) { go[j] = r; go[GH + j] = z; go[2 * GH + j] = n; go[3 * GH + j] = hn; ho[j] = hv; }
            h[j] = hv;
        }
    }
}

// backward: dhs [L][E][24] (dL/dh_t from the heads), hs, gates, h0, whh [72][32] (k-major W_hh, columns 24..31 zero)
//           -> dgi [L][E][72] (dL/d(gi)), dgh [L][E][72] (dL/d(W_hh h + b_hh))
__global__ void __launch_bounds__(64) rs_gru_bwd_kernel(const float* __restrict__ dhs, const float* __restrict__ hs, const float* __restrict__ gates,
                                                        const float* __restrict__ h0, const float* __restrict__ whh, float* __restrict__ dgi,
                                                        float* __restrict__ dgh, int L, int E) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int ec = e < E ? e : E - 1;
    const cmem_t W = as_cmem(whh);
    float dh[GH];
#pragma unroll
    for (int j = 0; j < GH; ++j) dh[j] = 0.0f;
    for (int t = L - 1; t >= 0; --t) {
        const size_t te = (size_t)t * E + ec;
        const float* go = gates + te * (4 * GH);
        const float* hp = (t > 0) ? hs + ((size_t)(t - 1) * E + ec) * GH : h0 + (size_t)ec * GH;
        float dg[G3];                                     // dL/d(gh): [dr | dz | dn * r]
        float nxt[32];
#pragma unroll
        for (int j = 0; j < GH; ++j) {
            const float d = dh[j] + dhs[te * GH + j];
            const float r = go[j], z = go[GH + j], n = go[2 * GH + j], hn = go[3 * GH + j];
            const float dn = d * (1.0f - z) * (1.0f - n * n);
            const float dz = d * (hp[j] - n) * z * (1.0f - z);
            const float dr = dn * hn * r * (1.0f - r);
            dg[j] = dr; dg[GH + j] = dz; dg[2 * GH + j] = dn * r;
            if (e < E) {
                dgi[te * G3 + j] = dr; dgi[te * G3 + GH + j] = dz; dgi[te * G3 + 2 * GH + j] = dn;
                dgh[te * G3 + j] = dr; dgh[te * G3 + GH + j] = dz; dgh[te * G3 + 2 * GH + j] = dn * r;
            }
            nxt[j] = d * z;                               // the direct path h_{t-1} -> h_t
        }
#pragma unroll
        for (int j = GH; j < 32; ++j) nxt[j] = 0.0f;
        gru_matvec<G3, 32>(W, [&](int k) -> float { return dg[k]; }, nxt);       // + W_hh^T dgh
#pragma unroll
        for (int j = 0; j < GH; ++j) dh[j] = nxt[j];
    }
}

}  // namespace

extern "C" {

int rs_gru_forward(const float* gi, const float* h0, const float* whh_t, const float* bhh, float* hs, float* gates, int32_t steps,
                   int32_t episodes, rs_stream_t stream) {
    if (!gi || !h0 || !whh_t || !bhh || !hs || !gates || steps < 1 || episodes < 1) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_gru_fwd_kernel, dim3((episodes + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), gi, h0, whh_t, bhh,
                       hs, gates, steps, episodes);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_gru_backward(const float* dhs, const float* hs, const float* gates, const float* h0, const float* whh, float* dgi, float* dgh,
                    int32_t steps, int32_t episodes, rs_stream_t stream) {
    if (!dhs || !hs || !gates || !h0 || !whh || !dgi || !dgh || steps < 1 || episodes < 1) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_gru_bwd_kernel, dim3((episodes + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), dhs, hs, gates, h0,
                       whh, dgi, dgh, steps, episodes);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
