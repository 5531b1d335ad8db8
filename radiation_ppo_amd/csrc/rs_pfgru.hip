// rs_pfgru.hip -- K11: one forward step of the PFGRU location predictor for every (owner, env).
//
// Replaces `self.model(obs_tensor, hidden)` of CNNBase.select_action (algos/test_cnn/RADTEAM_core.py:1872-1879), i.e.
// PFGRUCell.forward (:1586-1631) with observation_likelihood (:1633-1641), soft resampling (PFRNNBaseCell.resampling
// :1466-1515) and reparameterize (:1517-1530): 40 particles x 24 hidden units, alpha 0.7, tanh.
//
// Mapping: one wave per (owner, env), one particle per lane (40 of 64 lanes).  A particle's 24 hidden units, its gates and
// its candidate state live in the lane's registers; the three small matrix products (27 -> 48, 27 -> 48, 27 -> 1) are
// per-lane FMA chains whose weights are wave-uniform (one owner per wave): they arrive through the scalar unit
// (s_load_dwordx16 from the constant address space -> SGPR operand of v_fma), costing neither VGPRs nor LDS bandwidth.
// What couples the particles -- log-softmax, the resampling CDF, the gather of resampled particles, the weighted mean --
// goes through wave reductions and a 4.6 KB per-wave LDS tile.  Random draws (reparameterisation noise, resampling
// uniforms) are the counter hash of radiation_ppo_amd/pfgru.py evaluated in the kernel: nothing is read but the
// observation row, the particle set (3.9 KB) and the 13.5 KB of weights (L2 / scalar cache resident).
//
// Bound: VALU issue (2 700 FMAs + 24 hashes + ~120 transcendentals per lane); 16 384 waves per step at config 4.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radsearch.h"
#include "rs_wave.hpp"

namespace {

constexpr int PF_P = RS_PFGRU_PARTICLES, PF_H = RS_PFGRU_HIDDEN, PF_IN = 3, PF_K = PF_H + PF_IN;   // 40, 24, 3, 27
// packed weights of one owner (floats), produced on the host (radiation_ppo_amd/pfgru.py: pack_weights):
//   zr_t [28][48] (k-major, row 27 = 0; outputs 0..23 = fc_z, 24..47 = fc_r) | zr_b [48] | n_t [28][48] (0..23 mu, 24..47 var) | n_b [48]
//   | o_w [27] | o_b [1] | pad to 16 | h0_t [24][24] (k-major) | h0_b [24] | h2_w [2][24] | h2_b [2] | pad
constexpr int PF_KP = 28;                                          // k rows padded to an even count: the products run two rows per block
constexpr int PF_ZR = 0, PF_ZRB = PF_ZR + PF_KP * 48, PF_N = PF_ZRB + 48, PF_NB = PF_N + PF_KP * 48, PF_O = PF_NB + 48,
              PF_OB = PF_O + PF_K, PF_H0 = ((PF_OB + 1 + 15) / 16) * 16, PF_H0B = PF_H0 + PF_H * 24, PF_H2 = PF_H0B + 24,
              PF_H2B = PF_H2 + 48, PF_STRIDE = ((PF_H2B + 2 + 15) / 16) * 16;
static_assert(PF_STRIDE == RS_PFGRU_WEIGHT_FLOATS, "include/radsearch.h: RS_PFGRU_WEIGHT_FLOATS");

typedef const float __attribute__((address_space(4))) * cmem_t;
__device__ __forceinline__ cmem_t as_cmem(const float* p) { return (cmem_t)(uintptr_t)p; }

// splitmix64 finaliser on wrapping 64-bit arithmetic == pfgru.py: hash_bits
__device__ __forceinline__ uint64_t pf_hash(uint64_t key) {
    uint64_t x = key * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float wave_max(float v) { return rs_wave_max(v); }     // csrc/rs_wave.hpp: DPP rows + v_readlane, no LDS
__device__ __forceinline__ float wave_sum(float v) { return rs_wave_sum(v); }

// out[48] = b + W^T c for a [28][48] k-major weight block read through the scalar unit.  Two k rows (2 x s_load_dwordx16 per
// 16-output chunk) form a block, double buffered (64 SGPRs live); a scheduling barrier closes every block: without it the scheduler
// hoists dozens of the (independent) loads, runs out of SGPRs and spills them through v_writelane / v_readlane (measured: 1 389 loads
// and 2 756 v_readlane in 21 k instructions for this kernel, 419 us per step at config 4).  Inside a block the order is
//     wait for this block's rows  ->  request the next block  ->  32 FMAs:
// scalar loads return out of order, the only wait is lgkmcnt(0) and it drains everything in flight, so a request issued BEFORE the wait
// (hipcc's own order) is waited for at once.  The empty asm "reads" the current rows and so pins the wait in front of the requests.
#define PF_ARRIVED(w) asm volatile("" :: "s"((w)[0]), "s"((w)[16]))
template <int WOFF, int BOFF, typename F>
__device__ __forceinline__ void pf_matvec48(cmem_t W, F cval, float (&out)[48]) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float acc[16], wq[2][32];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = W[BOFF + 16 * ch + o];
#pragma unroll
        for (int i = 0; i < 32; ++i) wq[0][i] = W[WOFF + (i >> 4) * 48 + 16 * ch + (i & 15)];
#pragma unroll
        for (int b = 0; b < PF_KP / 2; ++b) {
            float (&cur)[32] = wq[b & 1];
            PF_ARRIVED(cur);
            __builtin_amdgcn_sched_barrier(0);
            if (b + 1 < PF_KP / 2) {
#pragma unroll
                for (int i = 0; i < 32; ++i) wq[(b + 1) & 1][i] = W[WOFF + (2 * (b + 1) + (i >> 4)) * 48 + 16 * ch + (i & 15)];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float c0 = cval(2 * b), c1 = cval(2 * b + 1);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fmaf(cur[o], c0, acc[o]);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fmaf(cur[16 + o], c1, acc[o]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            // materialise the result here: otherwise the compiler sinks each output's FMA chain to its use (to save VGPRs) and
            // re-reads a whole 16-dword row per FMA (seen in the ISA: one s_load_dwordx16 + s_waitcnt per v_fmac)
            asm volatile("" : "+v"(acc[o]));
            out[16 * ch + o] = acc[o];
        }
    }
}

struct PfArgs {
    const float* w;           // [A][PF_STRIDE]
    const float* obs;         // [N][A][11]
    float* h;                 // [A][N][P][H]
    float* p;                 // [A][N][P]
    const int64_t* base;      // [A][N]  per (owner, env) key
    const int64_t* episode;   // [N]
    const int64_t* calls;     // [N]
    const uint8_t* mask;      // [N] or null: envs whose carried state is updated
    float* pred;              // [N][A][2]
    const float* eps_in;      // [A][N][P][H] recorded reparameterisation noise   } REC instantiation only: the draws the reference
    const int32_t* idx_in;    // [A][N][P]    recorded resampling indices         } made (tests/golden/pfgru.npz, rada2c_core.npz)
    int N, A, carry;
    float alpha, floor_;      // soft-resampling alpha and (1 - alpha) / P, rounded to float32 as torch does for scalars
};

constexpr int PF_ROW = PF_H + 1;                                   // odd row stride: conflict-free row writes and column reads
constexpr int PF_LDS_WAVE = PF_P * PF_ROW * 4 + PF_P * 8 + 64 * 4; // h tile, cdf (f64), p1 / mean

template <bool REC>
__global__ void __launch_bounds__(256, 2) rs_pfgru_kernel(PfArgs a_) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wv = (long long)blockIdx.x * 4 + wid;
    if (wv >= (long long)a_.A * a_.N) return;                       // whole waves only: no barrier below
    const int own = __builtin_amdgcn_readfirstlane((int)(wv / a_.N));
    const int n = __builtin_amdgcn_readfirstlane((int)(wv - (long long)own * a_.N));
    // a masked round (the collectors' bootstrap predictions, train.py:462-480) only counts for the masked envs: the others' rows are
    // discarded by the caller, so their waves leave at once (a bootstrap round costs the few envs that time out, not all of them)
    if (a_.mask != nullptr && a_.mask[n] == 0) return;
    unsigned char* base = smem + (size_t)wid * PF_LDS_WAVE;
    float* tile = reinterpret_cast<float*>(base);                   // [P][PF_ROW]
    double* cdf = reinterpret_cast<double*>(base + PF_P * PF_ROW * 4);
    float* vec = reinterpret_cast<float*>(base + PF_P * PF_ROW * 4 + PF_P * 8);   // [64]

    const bool act = lane < PF_P;
    const int pl = act ? lane : PF_P - 1;                           // idle lanes shadow the last particle (values discarded)
    cmem_t W = as_cmem(a_.w + (size_t)own * PF_STRIDE);
    const size_t slot = (size_t)own * a_.N + n;
    const float* hp = a_.h + (slot * PF_P + pl) * PF_H;
    float h0[PF_H];
#pragma unroll
    for (int u = 0; u < PF_H; u += 4) {
        const float4 v = *reinterpret_cast<const float4*>(hp + u);
        h0[u] = v.x; h0[u + 1] = v.y; h0[u + 2] = v.z; h0[u + 3] = v.w;
    }
    const float p0 = a_.p[slot * PF_P + pl];
    float x[PF_IN];
    {
        cmem_t o = as_cmem(a_.obs + ((size_t)n * a_.A + own) * RS_OBS_DIM);
#pragma unroll
        for (int k = 0; k < PF_IN; ++k) x[k] = o[k];
    }
    // keys (pfgru.py: PredictorBank._key): kind 1 = reparameterisation noise, 2 = resampling uniforms
    uint64_t k_res = 0, pk = 0;
    if constexpr (!REC) {
        const uint64_t kb = (uint64_t)a_.base[slot] * 1000003ull;
        const uint64_t ctr8 = ((uint64_t)a_.episode[n] * 100003ull + (uint64_t)a_.calls[n]) * 8ull;
        const uint64_t k_eps = kb ^ ((ctr8 + 1ull) * 0xA24BAED4963EE407ull);
        k_res = kb ^ ((ctr8 + 2ull) * 0xA24BAED4963EE407ull);
        pk = k_eps * 1048583ull + (uint64_t)pl * 4096ull;
    }

    // ---- gates: z | r = sigmoid(W_zr [h0, x] + b)
    float g[48];
    pf_matvec48<PF_ZR, PF_ZRB>(W, [&](int k) -> float { return (k < PF_H) ? h0[k < PF_H ? k : 0] : (k < PF_K ? x[(k >= PF_H && k < PF_K) ? k - PF_H : 0] : 0.0f); }, g);
#pragma unroll
    for (int o = 0; o < 48; ++o) g[o] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * g[o]));   // sigmoid: v_exp_f32, v_rcp_f32 (1 ulp each)
    // ---- candidate: n = tanh(mu + eps * softplus(var)), [mu | var] = W_n [r * h0, x] + b
    float m[48];
    pf_matvec48<PF_N, PF_NB>(W, [&](int k) -> float { return (k < PF_H) ? g[24 + (k < PF_H ? k : 0)] * h0[k < PF_H ? k : 0] : (k < PF_K ? x[(k >= PF_H && k < PF_K) ? k - PF_H : 0] : 0.0f); }, m);
    float h1[PF_H];
#pragma unroll
    for (int u = 0; u < PF_H; ++u) {
        float eps;
        if constexpr (REC) {
            eps = a_.eps_in[(slot * PF_P + pl) * PF_H + u];
        } else {
            const uint64_t hx = pf_hash(pk + (uint64_t)u);
            const float u1 = (float)((uint32_t)(hx >> 40) + 1u) * (1.0f / 16777216.0f);          // (0, 1]
            const float u2 = (float)((uint32_t)(hx >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);   // [0, 1)
            // Box-Muller on the hardware transcendentals (1 ulp each; v_cos_f32 takes revolutions: cos(2 pi u2) is ONE instruction,
            // the library cosf would drag its Payne-Hanek reduction along): |error| ~ 1e-6 on eps, inside the test tolerance
            eps = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
        }
        const float var = m[24 + u];
        const float sp = (var > 20.0f) ? var : 0.69314718f * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(1.44269504f * var));   // F.softplus
        const float y = m[u] + eps * sp;
        const float nv = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * y));                          // tanh
        h1[u] = (1.0f - g[u]) * nv + g[u] * h0[u];
    }
    // ---- observation likelihood, log-softmax over the particles
    float lg = W[PF_OB];
#pragma unroll
    for (int k = 0; k < PF_K; ++k) lg = fmaf(W[PF_O + k], (k < PF_H) ? h1[k < PF_H ? k : 0] : x[k < PF_H ? 0 : k - PF_H], lg);
    lg += p0;
    const float mx = wave_max(act ? lg : -INFINITY);
    const float se = wave_sum(act ? expf(lg - mx) : 0.0f);
    float p1 = (lg - mx) - logf(se);
    // ---- soft resampling: indices by inverse CDF of alpha * w + (1 - alpha) / P
    const float al = a_.alpha, floor_ = a_.floor_;
    {
        double c = act ? (double)(al * expf(p1) + floor_) : 0.0;
        c = rs_wave_scan(c);                                       // inclusive scan over the lanes (float64) on the DPP path
        const double tot = rs_lane_d<PF_P - 1>(c);
        if (act) cdf[lane] = c / tot;
#pragma unroll
        for (int u = 0; u < PF_H; ++u) if (act) tile[lane * PF_ROW + u] = h1[u];
        vec[lane] = p1;
    }
    __builtin_amdgcn_wave_barrier();
    int idx = 0;
    if constexpr (REC) {
        idx = min(max(a_.idx_in[slot * PF_P + pl], 0), PF_P - 1);
    } else {
        const double ru = (double)(pf_hash(k_res * 1048583ull + (uint64_t)pl * 4096ull) >> 11) * (1.0 / 9007199254740992.0);
        for (int q = 0; q < PF_P; ++q) idx += (cdf[q] <= ru) ? 1 : 0;  // searchsorted(..., right=True)
        idx = min(idx, PF_P - 1);
    }
#pragma unroll
    for (int u = 0; u < PF_H; ++u) h1[u] = tile[idx * PF_ROW + u];
    float pn = expf(vec[idx]);
    pn = logf(pn / (al * pn + floor_));
    const float mx2 = wave_max(act ? pn : -INFINITY);
    const float lse = logf(wave_sum(act ? expf(pn - mx2) : 0.0f)) + mx2;
    p1 = pn - lse;
    if (a_.carry && (a_.mask == nullptr || a_.mask[n]) && act) {
        float* hw = a_.h + (slot * PF_P + lane) * PF_H;
#pragma unroll
        for (int u = 0; u < PF_H; u += 4) *reinterpret_cast<float4*>(hw + u) = make_float4(h1[u], h1[u + 1], h1[u + 2], h1[u + 3]);
        a_.p[slot * PF_P + lane] = p1;
    }
    // ---- weighted mean of the particles, then hid_obs: Linear(24, 24)-ReLU-Linear(24, 2)-ReLU
    __builtin_amdgcn_wave_barrier();
    const float wgt = expf(p1);
#pragma unroll
    for (int u = 0; u < PF_H; ++u) if (act) tile[lane * PF_ROW + u] = wgt * h1[u];
    __builtin_amdgcn_wave_barrier();
    const int ul = lane < PF_H ? lane : PF_H - 1;
    float mean = 0.0f;
    for (int q = 0; q < PF_P; ++q) mean += tile[q * PF_ROW + ul];
    __builtin_amdgcn_wave_barrier();
    vec[lane] = mean;                                              // lanes 0..23: mean_hid[lane]
    __builtin_amdgcn_wave_barrier();
    const float* wg = a_.w + (size_t)own * PF_STRIDE;
    float t = wg[PF_H0B + ul];
    for (int k = 0; k < PF_H; ++k) t = fmaf(wg[PF_H0 + k * 24 + ul], vec[k], t);
    t = fmaxf(t, 0.0f);
    const float o0 = wave_sum(lane < PF_H ? wg[PF_H2 + ul] * t : 0.0f) + wg[PF_H2B];
    const float o1 = wave_sum(lane < PF_H ? wg[PF_H2 + 24 + ul] * t : 0.0f) + wg[PF_H2B + 1];
    if (lane == 0) {
        float* out = a_.pred + ((size_t)n * a_.A + own) * 2;
        out[0] = fmaxf(o0, 0.0f); out[1] = fmaxf(o1, 0.0f);
    }
}

// reset_hidden (RADTEAM_core.py:2030-2033) for the masked envs: h0 ~ U[0,1) from the hash (kind 0), p0 = log(1 / P).
// One lane per (owner, env, particle); the caller has already advanced episode[] and zeroed calls[] for those envs.
__global__ void __launch_bounds__(256) rs_pfgru_reset_kernel(float* h, float* p, const int64_t* base, const int64_t* episode,
                                                             const int64_t* calls, const uint8_t* mask, int N, int A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)A * N * PF_P) return;
    const int pl = (int)(i % PF_P);
    const long long slot = i / PF_P;
    const int n = (int)(slot % N);
    if (mask && !mask[n]) return;
    const uint64_t kb = (uint64_t)base[slot] * 1000003ull;
    const uint64_t ctr8 = ((uint64_t)episode[n] * 100003ull + (uint64_t)calls[n]) * 8ull;
    const uint64_t pk = (kb ^ (ctr8 * 0xA24BAED4963EE407ull)) * 1048583ull + (uint64_t)pl * 4096ull;
    float* hw = h + i * PF_H;
#pragma unroll
    for (int u = 0; u < PF_H; u += 4) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (float)((double)(pf_hash(pk + (uint64_t)(u + q)) >> 11) * (1.0 / 9007199254740992.0));
        *reinterpret_cast<float4*>(hw + u) = make_float4(v[0], v[1], v[2], v[3]);
    }
    p[i] = -3.6888794541139363f;                                   // float32(log(1 / 40))
}

// The draws of one training pass over an episode-major batch (rada2c.HashDraws) in one launch instead of ~25 int64 element-wise
// launches per step: key[e] -> h0 [E][P][H] (uniforms, kind 0), eps [L][E][P][H] (standard normals, kind 1, step t) and
// u [L][E][P] (resampling uniforms, kind 2, step t).  One lane per (t, episode, particle).
__global__ void __launch_bounds__(256) rs_pfgru_draws_kernel(const int64_t* __restrict__ key, int E, int L, float* __restrict__ h0,
                                                             float* __restrict__ eps, double* __restrict__ u) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)L * E * PF_P) return;
    const int pl = (int)(i % PF_P);
    const long long te = i / PF_P;
    const int e = (int)(te % E), t = (int)(te / E);
    const uint64_t kb = (uint64_t)key[e] * 1000003ull;
    const uint64_t pu = (uint64_t)pl * 4096ull;
    const uint64_t k1 = (kb ^ ((uint64_t)(t * 8 + 1) * 0xA24BAED4963EE407ull)) * 1048583ull + pu;
    float* ew = eps + i * PF_H;
#pragma unroll
    for (int q = 0; q < PF_H; q += 4) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t hx = pf_hash(k1 + (uint64_t)(q + j));
            const float u1 = (float)((uint32_t)(hx >> 40) + 1u) * (1.0f / 16777216.0f);
            const float u2 = (float)((uint32_t)(hx >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);
            v[j] = sqrtf(-2.0f * logf(u1)) * cosf(6.2831855f * u2);              // pfgru.py: hash_normal (library functions, as torch)
        }
        *reinterpret_cast<float4*>(ew + q) = make_float4(v[0], v[1], v[2], v[3]);
    }
    const uint64_t k2 = (kb ^ ((uint64_t)(t * 8 + 2) * 0xA24BAED4963EE407ull)) * 1048583ull + pu;
    u[i] = (double)(pf_hash(k2) >> 11) * (1.0 / 9007199254740992.0);
    if (t == 0) {
        const uint64_t k0 = kb * 1048583ull + pu;                                 // kind 0, t 0: (0 * 8 + 0) * C = 0
        float* hw = h0 + ((long long)e * PF_P + pl) * PF_H;
#pragma unroll
        for (int q = 0; q < PF_H; ++q) hw[q] = (float)((double)(pf_hash(k0 + (uint64_t)q) >> 11) * (1.0 / 9007199254740992.0));
    }
}

}  // namespace

extern "C" {

int rs_pfgru_draws(const int64_t* keys, int32_t episodes, int32_t steps, float* h0, float* eps, double* u, rs_stream_t stream) {
    if (!keys || !h0 || !eps || !u || episodes < 1 || steps < 1) return RS_ERR_INVALID_ARG;
    const long long lanes = (long long)steps * episodes * PF_P;
    hipLaunchKernelGGL(rs_pfgru_draws_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), keys,
                       episodes, steps, h0, eps, u);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_pfgru_reset(float* h, float* p, const int64_t* base_key, const int64_t* episode, const int64_t* calls, const uint8_t* mask,
                   int32_t num_envs, int32_t num_agents, rs_stream_t stream) {
    if (!h || !p || !base_key || !episode || !calls || num_envs < 1 || num_agents < 1) return RS_ERR_INVALID_ARG;
    const long long lanes = (long long)num_envs * num_agents * PF_P;
    hipLaunchKernelGGL(rs_pfgru_reset_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h, p, base_key, episode, calls, mask, num_envs, num_agents);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_pfgru_step(const float* weights, const float* obs, float* h, float* p, const int64_t* base_key, const int64_t* episode,
                  const int64_t* calls, const uint8_t* mask, int32_t carry_hidden, double alpha, float* pred, int32_t num_envs,
                  int32_t num_agents, rs_stream_t stream) {
    if (!weights || !obs || !h || !p || !base_key || !episode || !calls || !pred || num_envs < 1 || num_agents < 1)
        return RS_ERR_INVALID_ARG;
    PfArgs a{weights, obs, h, p, base_key, episode, calls, mask, pred, nullptr, nullptr, num_envs, num_agents, carry_hidden ? 1 : 0,
             (float)alpha, (float)((1.0 - alpha) / (double)PF_P)};
    const long long waves = (long long)num_envs * num_agents;
    hipLaunchKernelGGL(rs_pfgru_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 4 * PF_LDS_WAVE,
                       static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_pfgru_step_recorded(const float* weights, const float* obs, float* h, float* p, const float* eps, const int32_t* idx,
                           const uint8_t* mask, int32_t carry_hidden, double alpha, float* pred, int32_t num_envs, int32_t num_agents,
                           rs_stream_t stream) {
    if (!weights || !obs || !h || !p || !eps || !idx || !pred || num_envs < 1 || num_agents < 1) return RS_ERR_INVALID_ARG;
    PfArgs a{weights, obs, h, p, nullptr, nullptr, nullptr, mask, pred, eps, idx, num_envs, num_agents, carry_hidden ? 1 : 0,
             (float)alpha, (float)((1.0 - alpha) / (double)PF_P)};
    const long long waves = (long long)num_envs * num_agents;
    hipLaunchKernelGGL(rs_pfgru_kernel<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 4 * PF_LDS_WAVE,
                       static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
