// rs_pfgru.hip -- K11: one forward step of the PFGRU location predictor for every (owner, env).
//
// Replaces `self.model(obs_tensor, hidden)` of CNNBase.select_action (algos/test_cnn/RADTEAM_core.py:1872-1879), i.e.
// PFGRUCell.forward (:1586-1631) with observation_likelihood (:1633-1641), soft resampling (PFRNNBaseCell.resampling
// :1466-1515) and reparameterize (:1517-1530): 40 particles x 24 hidden units, alpha 0.7, tanh.
//
// Mapping: six (owner, env) particle sets per 256-thread workgroup, one particle per lane (240 of 256 lanes carry a particle; one
// owner per workgroup).  A particle's 24 hidden units, its gates and its candidate state live in the lane's registers; the three
// small matrix products (27 -> 48, 27 -> 48, 27 -> 1) are per-lane FMA chains whose weights are workgroup-uniform: they arrive
// through the scalar unit (s_load_dwordx16 from the constant address space -> SGPR operand of v_fma, ordered wait -> request ->
// FMA by rs_sstream.hpp), costing neither VGPRs nor LDS bandwidth.  Both gate products are consumed chunk by chunk (16 live
// accumulators, packed in pairs: 147-155 VGPRs, three waves per SIMD, no scratch).  A set's 40 lanes straddle waves, so what couples its particles --
// two log-softmaxes, the float64 resampling CDF, the gather of resampled particles, the weighted mean, the hid_obs head -- goes
// through LDS and workgroup barriers; every lane reads the set's 40 values back and reduces them in index order (deterministic,
// independent of which sets share a workgroup).  Random draws (reparameterisation noise: one splitmix64 hash per PAIR of units ->
// Box-Muller cos / sin; resampling uniforms) are the counter hash of radiation_ppo_amd/pfgru.py evaluated in the kernel: nothing is
// read but the observation row, the quad-major particle set (3.9 KB) and the 13.9 KB of weights (scalar cache / L2 resident).
//
// Instantiations: <false, 1> the collectors' step (rs_pfgru_step); <false, 4> FOUR time steps per launch as four straight-line
// copies of the step with the particle set in registers in between (rs_pfgru_pass: the policy loop's 120-step passes in 30
// launches); <true, 1> the same arithmetic with noise / resampling indices read from buffers (the reference's recorded runs).
//
// Bound: VALU issue.  Algorithmic work 2 619 multiply-adds per particle-step (2 x 27 x 48 + 27); the kernel issues 1 296 v_pk_fma_f32
// (two multiply-adds each; a SIMD retires one per 4 cycles, a plain v_fma_f32 per 2: the same FLOP rate from half the instructions) +
// 12 hashes + ~300 hardware transcendentals (8 cycles each) per lane and step.  16 384 sets x 40 particles at config 4: 98 us per step =
// 35 TFLOP/s of algorithmic work = 0.22 of the 157.3 TFLOP/s f32 peak (bench.py: roofline_pfgru_step).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radsearch.h"
#include "rs_sstream.hpp"

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


// sigmoid on the hardware transcendentals: v_exp_f32, v_rcp_f32 (1 ulp each)
__device__ __forceinline__ float pf_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v)); }
// the softmax / soft-resampling exponentials, logarithms and the one f32 quotient on the hardware transcendentals too (1 ulp each; the
// library expf / logf / IEEE division were ~110 instructions per lane and step).  The float64 CDF quotient stays IEEE: it decides indices.
__device__ __forceinline__ float pf_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504f * x); }
__device__ __forceinline__ float pf_log(float x) { return 0.69314718f * __builtin_amdgcn_logf(x); }

struct PfArgs {
    const float* w;           // [A][PF_STRIDE]
    const float* obs;         // [N][A][11]
    float* h;                 // [A][N][H / 4][P][4]: quad-major particle sets
    float* p;                 // [A][N][P]
    const int64_t* base;      // [A][N]  per (owner, env) key
    const int64_t* episode;   // [N]
    const int64_t* calls;     // [N]
    const uint8_t* mask;      // [N] or null: envs whose carried state is updated
    float* pred;              // [N][A][2]
    const float* eps_in;      // [A][N][P][H] recorded reparameterisation noise   } REC instantiation only: the draws the reference
    const int32_t* idx_in;    // [A][N][P]    recorded resampling indices         } made (tests/golden/pfgru.npz, rada2c_core.npz)
    int N, A, carry;
    int Ns[5];                // STEPS > 1 (rs_pfgru_pass): the sets still running at the launch's 2nd .. 6th step (prefixes of N, descending)
    long long step_stride;    // STEPS > 1: elements between two steps' rows of calls[]; obs / pred rows are 11 / 2 x that apart
    float alpha, floor_;      // soft-resampling alpha and (1 - alpha) / P, rounded to float32 as torch does for scalars
};

constexpr int PF_ROW = PF_H + 1;                                   // odd row stride: conflict-free row writes and column reads
// Lane packing (round 3): six (owner, env) sets of 40 particles share a 256-thread workgroup -- 240 of 256 lanes carry a particle
// where one set per wave used 40 of 64.  A set's lanes straddle waves, so what couples its particles goes through LDS and workgroup
// barriers instead of wave operations: every lane reads the set's 40 values back (broadcast ds_read_b128) and reduces them itself in
// index order -- deterministic and independent of which sets share the workgroup.  Nine barriers per step cost 1.5 % (measured by
// putting them into the one-set-per-wave kernel); the lanes gained are 1.5x.  Four waves per workgroup = one per SIMD, three workgroups per
// CU at 168 VGPRs (nine sets on 384 threads left the second workgroup without room on two of the SIMDs: 188 us per step).
constexpr int PK_SETS = 6, PK_NT = 256;
#ifndef RS_K11_PASS_STEPS
#define RS_K11_PASS_STEPS 4        // time steps per launch of rs_pfgru_pass (1 .. 6).  A/B, 120-step pass of 16.5 k episodes inside the policy
                                   // loop: 1 = 14.3 ms, 2 = 13.25, 3 = 12.89, 4 = 12.61, 5 = 12.76, 6 = 12.8 (scalar FMAs at 4 waves per SIMD, 127 VGPRs; with the
                                   // packed pairs at 3 waves per SIMD 4 = 11.5 ms); the code of a copy is 35 KB; 8 copies do not compile: "illegal VGPR to SGPR copy")
#endif
// LDS of one set (floats): tile [40][25] | cdf 40 x f64 | va [40] | vb [40] | vc [40] | vm [24] (+ pad); the stride is 12 (mod 32) banks so
// that the up to three sets a wave touches read different banks
constexpr int PK_TILE = 0, PK_CDF = PF_P * PF_ROW, PK_VA = PK_CDF + 2 * PF_P, PK_VB = PK_VA + PF_P, PK_VC = PK_VB + PF_P,
              PK_VM = PK_VC + PF_P, PK_STRIDE = 1228;
static_assert(PK_VM + PF_H <= PK_STRIDE && PK_STRIDE % 4 == 0 && PK_STRIDE % 32 == 12 && PK_CDF % 2 == 0, "LDS layout of a particle set");

__device__ __forceinline__ float pk_max40(const float* v) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < PF_P / 4; ++i) {
        const float4 t = reinterpret_cast<const float4*>(v)[i];
        m = fmaxf(fmaxf(m, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
    }
    return m;
}
__device__ __forceinline__ float pk_sum40(const float* v) {          // index order
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < PF_P / 4; ++i) {
        const float4 t = reinterpret_cast<const float4*>(v)[i];
        s = (((s + t.x) + t.y) + t.z) + t.w;
    }
    return s;
}

// STEPS > 1 (rs_pfgru_pass only, one owner): consecutive time steps of the carried sets from one launch, the particle set staying in
// registers in between -- a launch costs ~17 us before its first round of workgroups is at speed (profiles/r03_k11_rounds.txt), a pass
// has 120 steps.  The steps are explicit COPIES of the step (a generic lambda instantiated per step index): inside a runtime loop -- and
// hipcc keeps a `#pragma unroll` loop over the steps a loop -- the schedule of the weight stream falls apart (412 B of scratch, 204 SGPR
// spills).  Between the copies the weight pointer and the lane indices are laundered together with the log-weight
// the previous copy produced, so that nothing of the next copy is requested, computed or kept early (without: 360 B of scratch).
#ifndef RS_K11_OCC
#define RS_K11_OCC 3        // waves per SIMD the register budget is cut for.  With the weight streams' accumulators as v_pk_fma_f32 pairs (rs_sstream.hpp)
                            // 4 spills 20-72 registers (84-116 B of scratch); 3 = 147-155 VGPRs, none.  A/B on one box, RAD-A2C leg of bench.py
                            // (gpurun_out/r4b_ab1.txt): scalar FMAs at 4: pass 12.8 ms, 1.45 M env steps/s; pairs at 4: 12.0 ms, 1.49 M; pairs at 3: 11.5 ms, 1.54 M
#endif
template <bool REC, int STEPS = 1>
__global__ void __launch_bounds__(PK_NT, RS_K11_OCC) rs_pfgru_kernel(PfArgs a_, int groups) {
    __shared__ __align__(16) float smem[PK_SETS * PK_STRIDE];
    const int tid = threadIdx.x;
    const int own = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (unsigned)groups));       // one owner per workgroup: wave-uniform weights
    const int grp = blockIdx.x - own * groups;
    const int set = tid / PF_P, q = tid - set * PF_P;                // set == PK_SETS: the 16 lanes that carry nothing
    const int n_raw = grp * PK_SETS + set;
    const bool in_range = set < PK_SETS && n_raw < a_.N;
    const int n = in_range ? n_raw : a_.N - 1;                      // lanes without a set shadow the last env (nothing is stored)
    // a masked round (the collectors' bootstrap predictions, train.py:462-480) only counts for the masked envs: the others' rows are
    // discarded by the caller; a workgroup without a masked env leaves at once
    const bool live = in_range && (a_.mask == nullptr || a_.mask[n] != 0);
    if (__syncthreads_or(live ? 1 : 0) == 0) return;
    float* S = smem + (set < PK_SETS ? set : PK_SETS - 1) * PK_STRIDE;      // (the spare lanes read the last set's area and write nothing)
    float* tile = S + PK_TILE;
    double* cdf = reinterpret_cast<double*>(S + PK_CDF);
    float *va = S + PK_VA, *vb = S + PK_VB, *vc = S + PK_VC, *vm = S + PK_VM;
    const bool act = set < PK_SETS;                                 // LDS writes allowed (the set's own area)

    const size_t slot = (size_t)own * a_.N + n;
    // the particle sets are stored QUAD-major ([H / 4][P] float4 per set): a load / store instruction of a set's 40 lanes is 640 contiguous
    // bytes (a particle's 24 floats contiguous meant 16-byte pieces at a 96-byte stride; K13's scratch: 14 % of its forward walk)
    const float4* hp = reinterpret_cast<const float4*>(a_.h + slot * PF_P * PF_H) + q;
    float h0[PF_H];
#pragma unroll
    for (int u = 0; u < PF_H; u += 4) {
        const float4 v = hp[(u / 4) * PF_P];
        h0[u] = v.x; h0[u + 1] = v.y; h0[u + 2] = v.z; h0[u + 3] = v.w;
    }
    float p0 = a_.p[slot * PF_P + q];
    const int q_ = q, n_ = n;
    auto step = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s_ = decltype(sc)::value;
        const float* wl = a_.w + (size_t)own * PF_STRIDE;
        int ql = q_, nl = n_;
        // the second copy re-reads the weights and re-derives its keys and addresses, and not before the first one has produced its
        // log-weights: nothing of it is requested, computed or kept early
        // (the single hashing step takes the laundering too: 125 VGPRs and no scratch instead of 128 + 40 B, 105 -> 98 us with carried sets)
        if constexpr (!REC || STEPS > 1) asm volatile("" : "+s"(wl), "+v"(p0), "+v"(ql), "+v"(nl));
        const int q = ql, n = nl;
        cmem_t W = as_cmem(wl);
        float x[PF_IN];
        {
            const float* o = a_.obs + (size_t)s_ * a_.step_stride * RS_OBS_DIM + ((size_t)n * a_.A + own) * RS_OBS_DIM;
#pragma unroll
            for (int k = 0; k < PF_IN; ++k) x[k] = o[k];
        }
        // keys (pfgru.py: PredictorBank._key): kind 1 = reparameterisation noise, 2 = resampling uniforms
        uint64_t k_res = 0, pk = 0;
        if constexpr (!REC) {
            const uint64_t kb = (uint64_t)a_.base[slot] * 1000003ull;
            const uint64_t ctr8 = ((uint64_t)a_.episode[n] * 100003ull + (uint64_t)a_.calls[n + (size_t)s_ * a_.step_stride]) * 8ull;
            const uint64_t k_eps = kb ^ ((ctr8 + 1ull) * 0xA24BAED4963EE407ull);
            k_res = kb ^ ((ctr8 + 2ull) * 0xA24BAED4963EE407ull);
            pk = k_eps * 1048583ull + (uint64_t)q * 4096ull;
        }

        // ---- gates: z | r = sigmoid(W_zr [h0, x] + b), consumed chunk by chunk (16 accumulators = 8 pairs live, not 48: the kernel
        // fits 3 waves per SIMD without spilling that way): z stays, r becomes r * h0 at once
        float z[PF_H], rh[PF_H];
        auto cv1 = [&](int k) -> float { return (k < PF_H) ? h0[k < PF_H ? k : 0] : x[(k >= PF_H && k < PF_K) ? k - PF_H : 0]; };
        {
            float acc[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = W[PF_ZRB + o];
            rs_ss_mv_cols<PF_K, 48, 0, 16>(W + PF_ZR, cv1, acc);
#pragma unroll
            for (int o = 0; o < 16; ++o) z[o] = pf_sigmoid(acc[o]);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = W[PF_ZRB + 16 + o];
            rs_ss_mv_cols<PF_K, 48, 16, 16>(W + PF_ZR, cv1, acc);
#pragma unroll
            for (int o = 0; o < 8; ++o) { z[16 + o] = pf_sigmoid(acc[o]); rh[o] = pf_sigmoid(acc[8 + o]) * h0[o]; }
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = W[PF_ZRB + 32 + o];
            rs_ss_mv_cols<PF_K, 48, 32, 16>(W + PF_ZR, cv1, acc);
#pragma unroll
            for (int o = 0; o < 16; ++o) rh[8 + o] = pf_sigmoid(acc[o]) * h0[8 + o];
        }
        // ---- candidate: n = tanh(mu + eps * softplus(var)), [mu | var] = W_n [r * h0, x] + b; the columns arrive as three chunks of
        // [mu(8j .. 8j+7) | var(8j .. 8j+7)] (pfgru.py: pack_weights), each finishing eight units of h1
        float h1[PF_H];
        auto cv2 = [&](int k) -> float { return (k < PF_H) ? rh[k < PF_H ? k : 0] : x[(k >= PF_H && k < PF_K) ? k - PF_H : 0]; };
        auto candidate = [&](auto jc, const float (&acc)[16]) {
            constexpr int J = decltype(jc)::value;
            float eps8[8];
#pragma unroll
            for (int w = 0; w < 8; w += 2) {
                const int u = 8 * J + w;
                if constexpr (REC) {
                    eps8[w] = a_.eps_in[(slot * PF_P + q) * PF_H + u];
                    eps8[w + 1] = a_.eps_in[(slot * PF_P + q) * PF_H + u + 1];
                } else {
                    // pfgru.py: hash_normal -- one hash per PAIR of units, Box-Muller's cosine for the even unit and sine for the odd one, on
                    // the hardware transcendentals (1 ulp each; v_cos_f32 / v_sin_f32 take revolutions: no range reduction): |error| ~ 1e-6
                    const uint64_t hx = pf_hash(pk + (uint64_t)u);
                    const float u1 = (float)((uint32_t)(hx >> 40) + 1u) * (1.0f / 16777216.0f);          // (0, 1]
                    const float u2 = (float)((uint32_t)(hx >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);   // [0, 1)
                    const float r = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u1));
                    eps8[w] = r * __builtin_amdgcn_cosf(u2);
                    eps8[w + 1] = r * __builtin_amdgcn_sinf(u2);
                }
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int u = 8 * J + w;
                const float eps = eps8[w];
                const float var = acc[8 + w];
                const float sp = (var > 20.0f) ? var : 0.69314718f * __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(1.44269504f * var));   // F.softplus
                const float y = acc[w] + eps * sp;
                const float nv = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * y));                          // tanh
                h1[u] = (1.0f - z[u]) * nv + z[u] * h0[u];
            }
        };
        {
            float acc[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = W[PF_NB + o];
            rs_ss_mv_cols<PF_K, 48, 0, 16>(W + PF_N, cv2, acc);
            candidate(std::integral_constant<int, 0>{}, acc);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = W[PF_NB + 16 + o];
            rs_ss_mv_cols<PF_K, 48, 16, 16>(W + PF_N, cv2, acc);
            candidate(std::integral_constant<int, 1>{}, acc);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = W[PF_NB + 32 + o];
            rs_ss_mv_cols<PF_K, 48, 32, 16>(W + PF_N, cv2, acc);
            candidate(std::integral_constant<int, 2>{}, acc);
        }
        // ---- observation likelihood, log-softmax over the set's particles
        float lg = W[PF_OB];
#pragma unroll
        for (int k = 0; k < PF_K; ++k) lg = fmaf(W[PF_O + k], (k < PF_H) ? h1[k < PF_H ? k : 0] : x[k < PF_H ? 0 : k - PF_H], lg);
        lg += p0;
        if (act) va[q] = lg;
        __syncthreads();                                                // 1
        const float mx = pk_max40(va);
        const float e1 = pf_exp(lg - mx);
        if (act) vb[q] = e1;
        __syncthreads();                                                // 2
        float p1 = (lg - mx) - pf_log(pk_sum40(vb));
        // ---- soft resampling: indices by inverse CDF of alpha * w + (1 - alpha) / P
        const float al = a_.alpha, floor_ = a_.floor_;
        if (act) {
            va[q] = al * pf_exp(p1) + floor_;
            vc[q] = p1;
#pragma unroll
            for (int u = 0; u < PF_H; ++u) tile[q * PF_ROW + u] = h1[u];
        }
        __syncthreads();                                                // 3
        int idx = 0;
        if constexpr (REC) {
            idx = min(max(a_.idx_in[slot * PF_P + q], 0), PF_P - 1);
        } else {
            double run = 0.0, mine = 0.0;                               // float64 prefix sums in index order (torch.cumsum of the .double() weights)
#pragma unroll
            for (int i = 0; i < PF_P / 4; ++i) {
                const float4 t = reinterpret_cast<const float4*>(va)[i];
                run += (double)t.x; mine = (4 * i == q) ? run : mine;
                run += (double)t.y; mine = (4 * i + 1 == q) ? run : mine;
                run += (double)t.z; mine = (4 * i + 2 == q) ? run : mine;
                run += (double)t.w; mine = (4 * i + 3 == q) ? run : mine;
            }
            if (act) cdf[q] = mine / run;
            __syncthreads();                                            // 4
            const double ru = (double)(pf_hash(k_res * 1048583ull + (uint64_t)q * 4096ull) >> 11) * (1.0 / 9007199254740992.0);
#pragma unroll
            for (int j = 0; j < PF_P / 2; ++j) {                        // searchsorted(..., right=True)
                const double2 c2 = reinterpret_cast<const double2*>(cdf)[j];
                idx += (c2.x <= ru) ? 1 : 0;
                idx += (c2.y <= ru) ? 1 : 0;
            }
            idx = min(idx, PF_P - 1);
        }
#pragma unroll
        for (int u = 0; u < PF_H; ++u) h1[u] = tile[idx * PF_ROW + u];
        float pn = pf_exp(vc[idx]);
        pn = pf_log(pn * __builtin_amdgcn_rcpf(al * pn + floor_));
        if (act) vb[q] = pn;
        __syncthreads();                                                // 5
        const float mx2 = pk_max40(vb);
        const float e2 = pf_exp(pn - mx2);
        if (act) va[q] = e2;
        __syncthreads();                                                // 6
        p1 = pn - (pf_log(pk_sum40(va)) + mx2);
        if (s_ == STEPS - 1 && a_.carry && live) {
            float4* hw = reinterpret_cast<float4*>(a_.h + slot * PF_P * PF_H) + q;
#pragma unroll
            for (int u = 0; u < PF_H; u += 4) hw[(u / 4) * PF_P] = make_float4(h1[u], h1[u + 1], h1[u + 2], h1[u + 3]);
            a_.p[slot * PF_P + q] = p1;
        }
        // ---- weighted mean of the particles, then hid_obs: Linear(24, 24)-ReLU-Linear(24, 2)-ReLU
        const float wgt = pf_exp(p1);
        if (act) {
#pragma unroll
            for (int u = 0; u < PF_H; ++u) tile[q * PF_ROW + u] = wgt * h1[u];      // every lane gathered its row before barrier 5
        }
        __syncthreads();                                                // 7
        const int ul = q < PF_H ? q : PF_H - 1;
        float mean = 0.0f;
        for (int j = 0; j < PF_P; ++j) mean += tile[j * PF_ROW + ul];
        if (act && q < PF_H) vm[q] = mean;
        __syncthreads();                                                // 8
        const float* wg = a_.w + (size_t)own * PF_STRIDE;
        float t = wg[PF_H0B + ul];
        for (int k = 0; k < PF_H; ++k) t = fmaf(wg[PF_H0 + k * 24 + ul], vm[k], t);
        t = fmaxf(t, 0.0f);
        if (act && q < PF_H) { vb[q] = wg[PF_H2 + ul] * t; vc[q] = wg[PF_H2 + 24 + ul] * t; }      // vb / vc: last read before barrier 6
        __syncthreads();                                                // 9
        if (live && q == 0 && (s_ == 0 || n < a_.Ns[s_ > 0 ? s_ - 1 : 0])) {
            float o0 = 0.0f, o1 = 0.0f;
            for (int k = 0; k < PF_H; ++k) { o0 += vb[k]; o1 += vc[k]; }
            float* out = a_.pred + (size_t)s_ * a_.step_stride * 2 + ((size_t)n * a_.A + own) * 2;
            out[0] = fmaxf(o0 + wg[PF_H2B], 0.0f); out[1] = fmaxf(o1 + wg[PF_H2B + 1], 0.0f);
        }
        if (s_ + 1 < STEPS) {                                       // the resampled set is the next step's input (no HBM round trip)
#pragma unroll
            for (int u = 0; u < PF_H; ++u) h0[u] = h1[u];
            p0 = p1;
        }
    };
    step(std::integral_constant<int, 0>{});
    if constexpr (STEPS > 1) step(std::integral_constant<int, 1>{});      // (a `#pragma unroll` loop over the steps is NOT unrolled by hipcc)
    if constexpr (STEPS > 2) step(std::integral_constant<int, 2>{});
    if constexpr (STEPS > 3) step(std::integral_constant<int, 3>{});
    if constexpr (STEPS > 4) step(std::integral_constant<int, 4>{});
    if constexpr (STEPS > 5) step(std::integral_constant<int, 5>{});
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
    float4* hw = reinterpret_cast<float4*>(h + slot * PF_P * PF_H) + pl;           // quad-major, as the step kernel reads it
#pragma unroll
    for (int u = 0; u < PF_H; u += 4) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (float)((double)(pf_hash(pk + (uint64_t)(u + q)) >> 11) * (1.0 / 9007199254740992.0));
        hw[(u / 4) * PF_P] = make_float4(v[0], v[1], v[2], v[3]);
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
        for (int j = 0; j < 4; j += 2) {
            // pfgru.py: hash_normal -- one hash per pair of units (cosine / sine of Box-Muller), on the hardware transcendentals as the
            // step kernel evaluates it (v_log_f32 is log2, v_cos_f32 / v_sin_f32 take revolutions; 1 ulp each): with the library logf /
            // cosf this kernel was instruction bound at a quarter of the HBM fill rate
            const uint64_t hx = pf_hash(k1 + (uint64_t)(q + j));
            const float u1 = (float)((uint32_t)(hx >> 40) + 1u) * (1.0f / 16777216.0f);
            const float u2 = (float)((uint32_t)(hx >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);
            const float r = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u1));
            v[j] = r * __builtin_amdgcn_cosf(u2);
            v[j + 1] = r * __builtin_amdgcn_sinf(u2);
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
    PfArgs a{weights, obs, h, p, base_key, episode, calls, mask, pred, nullptr, nullptr, num_envs, num_agents, carry_hidden ? 1 : 0, {0, 0, 0, 0, 0}, 0,
             (float)alpha, (float)((1.0 - alpha) / (double)PF_P)};
    const int groups = (num_envs + PK_SETS - 1) / PK_SETS;
    hipLaunchKernelGGL(rs_pfgru_kernel<false>, dim3((unsigned)(groups * num_agents)), dim3(PK_NT), 0, static_cast<hipStream_t>(stream), a, groups);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_pfgru_pass(const float* weights, const float* obs, float* h, float* p, const int64_t* base_key, const int64_t* episode,
                  const int64_t* calls, double alpha, float* pred, const int32_t* alive, int32_t steps, int32_t episodes, rs_stream_t stream) {
    if (!weights || !obs || !h || !p || !base_key || !episode || !calls || !pred || !alive || steps < 1 || episodes < 1) return RS_ERR_INVALID_ARG;
    for (int t = 0; t < steps; ++t)
        if (alive[t] < 0 || alive[t] > episodes || (t > 0 && alive[t] > alive[t - 1])) return RS_ERR_INVALID_ARG;
    int rc = rs_pfgru_reset(h, p, base_key, episode, calls, nullptr, episodes, 1, stream);
    int t = 0;
    // steps in groups of PASS_STEPS while the group's last one still has episodes (the multi-step instantiation: the sets alive at t, of
    // which the first alive[t + s] also take -- and report -- step t + s; the others' sets are dead by then, their extra steps are discarded)
    constexpr int PASS_STEPS = RS_K11_PASS_STEPS;
    static_assert(PASS_STEPS >= 1 && PASS_STEPS <= 6, "copies of the step per launch");
    for (; t + PASS_STEPS - 1 < steps && rc == RS_OK && alive[t + PASS_STEPS - 1] > 0; t += PASS_STEPS) {
        PfArgs a{weights, obs + (size_t)t * episodes * RS_OBS_DIM, h, p, base_key, episode, calls + (size_t)t * episodes, nullptr,
                 pred + (size_t)t * episodes * 2, nullptr, nullptr, alive[t], 1, 1, {0, 0, 0, 0, 0}, (long long)episodes,
                 (float)alpha, (float)((1.0 - alpha) / (double)PF_P)};
        for (int s_ = 1; s_ < PASS_STEPS; ++s_) a.Ns[s_ - 1] = alive[t + s_];
        const int groups = (alive[t] + PK_SETS - 1) / PK_SETS;
        hipLaunchKernelGGL((rs_pfgru_kernel<false, PASS_STEPS>), dim3((unsigned)groups), dim3(PK_NT), 0, static_cast<hipStream_t>(stream), a, groups);
        if (hipGetLastError() != hipSuccess) rc = RS_ERR_HIP;
    }
    for (; t < steps && rc == RS_OK && alive[t] > 0; ++t)
        rc = rs_pfgru_step(weights, obs + (size_t)t * episodes * RS_OBS_DIM, h, p, base_key, episode, calls + (size_t)t * episodes, nullptr, 1, alpha,
                           pred + (size_t)t * episodes * 2, alive[t], 1, stream);
    return rc;
}

int rs_pfgru_step_recorded(const float* weights, const float* obs, float* h, float* p, const float* eps, const int32_t* idx,
                           const uint8_t* mask, int32_t carry_hidden, double alpha, float* pred, int32_t num_envs, int32_t num_agents,
                           rs_stream_t stream) {
    if (!weights || !obs || !h || !p || !eps || !idx || !pred || num_envs < 1 || num_agents < 1) return RS_ERR_INVALID_ARG;
    PfArgs a{weights, obs, h, p, nullptr, nullptr, nullptr, mask, pred, eps, idx, num_envs, num_agents, carry_hidden ? 1 : 0, {0, 0, 0, 0, 0}, 0,
             (float)alpha, (float)((1.0 - alpha) / (double)PF_P)};
    const int groups = (num_envs + PK_SETS - 1) / PK_SETS;
    hipLaunchKernelGGL(rs_pfgru_kernel<true>, dim3((unsigned)(groups * num_agents)), dim3(PK_NT), 0, static_cast<hipStream_t>(stream), a, groups);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
