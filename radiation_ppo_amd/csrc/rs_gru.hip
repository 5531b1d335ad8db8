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
// bound (E / 64 waves), against ~45 ms for the library GRU's per-step launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radsearch.h"
#include "rs_sstream.hpp"

namespace {

constexpr int GH = RS_GRU_HIDDEN;            // 24
constexpr int G3 = 3 * GH;                   // 72

typedef const float __attribute__((address_space(4))) * cmem_t;
__device__ __forceinline__ cmem_t as_cmem(const float* p) { return (cmem_t)(uintptr_t)p; }

// out[OUTP] += W^T c through the scalar-unit weight stream (csrc/rs_sstream.hpp: wait -> request -> FMA blocks of two rows)
template <int K, int OUTP, int OUTR = OUTP, typename F>
__device__ __forceinline__ void gru_matvec(cmem_t W, F cval, float (&out)[OUTP]) { rs_ss_mv<K, OUTP, OUTR>(W, cval, out); }

// sigmoid / tanh on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each), branch free
__device__ __forceinline__ float gru_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float gru_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x)); }

template <int N>
__device__ __forceinline__ void ld_row(const float* __restrict__ src, float (&dst)[N]) {
#pragma unroll
    for (int u = 0; u < N; u += 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + u);
        dst[u] = v.x; dst[u + 1] = v.y; dst[u + 2] = v.z; dst[u + 3] = v.w;
    }
}
template <int N>
__device__ __forceinline__ void st_row(float* __restrict__ dst, const float (&src)[N]) {
#pragma unroll
    for (int u = 0; u < N; u += 4) *reinterpret_cast<float4*>(dst + u) = make_float4(src[u], src[u + 1], src[u + 2], src[u + 3]);
}

// forward: gi [L][E][72], h0 [E][24], whh_t [24][80] (k-major W_hh^T, columns 72..79 zero), bhh [80]
//          -> hs [L][E][24] (h_t), gates [L][E][96] = r | z | n | (W_hn h + b_hn)
// A step's rows are read and written as whole float4 rows, the next step's gi row is requested before this step's product: the
// first version loaded three words and stored five per hidden unit behind one another (24 memory round trips per step: 18 us
// per step, all of it waiting).
__global__ void __launch_bounds__(64) rs_gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ h0, const float* __restrict__ whh_t,
                                                        const float* __restrict__ bhh, float* __restrict__ hs, float* __restrict__ gates,
                                                        int L, int E) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int ec = e < E ? e : E - 1;                      // idle lanes shadow the last episode, store nothing
    float h[GH], g[G3];
    ld_row(h0 + (size_t)ec * GH, h);
    ld_row(gi + (size_t)ec * G3, g);
    for (int t = 0; t < L; ++t) {
        const float* wp = whh_t; asm volatile("" : "+s"(wp));                 // per step: keeps LICM from hoisting (and spilling) the rows
        const float* bp = bhh; asm volatile("" : "+s"(bp));
        const cmem_t W = as_cmem(wp), B = as_cmem(bp);
        float gn[G3];
        if (t + 1 < L) ld_row(gi + ((size_t)(t + 1) * E + ec) * G3, gn);
        float gh[80];
#pragma unroll
        for (int o = 0; o < 80; ++o) gh[o] = B[o];
        gru_matvec<GH, 80, G3>(W, [&](int k) -> float { return h[k]; }, gh);              // 72 of the 80 columns are real
        float go[4 * GH];
#pragma unroll
        for (int j = 0; j < GH; ++j) {
            const float r = gru_sigmoid(g[j] + gh[j]);
            const float z = gru_sigmoid(g[GH + j] + gh[GH + j]);
            const float hn = gh[2 * GH + j];
            const float n = gru_tanh(g[2 * GH + j] + r * hn);
            h[j] = (1.0f - z) * n + z * h[j];
            go[j] = r; go[GH + j] = z; go[2 * GH + j] = n; go[3 * GH + j] = hn;
        }
        if (e < E) {
            st_row(gates + ((size_t)t * E + ec) * (4 * GH), go);
            st_row(hs + ((size_t)t * E + ec) * GH, h);
        }
        if (t + 1 < L) {
#pragma unroll
            for (int u = 0; u < G3; ++u) g[u] = gn[u];
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
    float dh[GH];
#pragma unroll
    for (int j = 0; j < GH; ++j) dh[j] = 0.0f;
    float dx[GH], go[4 * GH], hp[GH];                     // the step's rows: dL/dh_t from the heads, gates, h_{t-1}
    auto load_step = [&](int t, float (&a)[GH], float (&b)[4 * GH], float (&c)[GH]) {
        const size_t te = (size_t)t * E + ec;
        ld_row(dhs + te * GH, a);
        ld_row(gates + te * (4 * GH), b);
        ld_row(t > 0 ? hs + (te - E) * GH : h0 + (size_t)ec * GH, c);
    };
    load_step(L - 1, dx, go, hp);
    for (int t = L - 1; t >= 0; --t) {
        const float* wp = whh; asm volatile("" : "+s"(wp));
        const cmem_t W = as_cmem(wp);
        const size_t te = (size_t)t * E + ec;
        float dxn[GH], gon[4 * GH], hpn[GH];
        if (t > 0) load_step(t - 1, dxn, gon, hpn);        // requested before this step's arithmetic
        float dg[G3], di[G3];                              // dL/d(gh) = [dr | dz | dn * r], dL/d(gi) = [dr | dz | dn]
        float nxt[32];
#pragma unroll
        for (int j = 0; j < GH; ++j) {
            const float d = dh[j] + dx[j];
            const float r = go[j], z = go[GH + j], n = go[2 * GH + j], hn = go[3 * GH + j];
            const float dn = d * (1.0f - z) * (1.0f - n * n);
            const float dz = d * (hp[j] - n) * z * (1.0f - z);
            const float dr = dn * hn * r * (1.0f - r);
            dg[j] = dr; dg[GH + j] = dz; dg[2 * GH + j] = dn * r;
            di[j] = dr; di[GH + j] = dz; di[2 * GH + j] = dn;
            nxt[j] = d * z;                               // the direct path h_{t-1} -> h_t
        }
        if (e < E) {
            st_row(dgi + te * G3, di);
            st_row(dgh + te * G3, dg);
        }
#pragma unroll
        for (int j = GH; j < 32; ++j) nxt[j] = 0.0f;
        gru_matvec<G3, 32, GH>(W, [&](int k) -> float { return dg[k]; }, nxt);       // + W_hh^T dgh
#pragma unroll
        for (int j = 0; j < GH; ++j) dh[j] = nxt[j];
        if (t > 0) {
#pragma unroll
            for (int j = 0; j < GH; ++j) { dx[j] = dxn[j]; hp[j] = hpn[j]; }
#pragma unroll
            for (int j = 0; j < 4 * GH; ++j) go[j] = gon[j];
        }
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
