// rs_pfgru_train.hip -- K13: loss and parameter gradients of the PFGRU location predictor over whole episodes, one call (a forward-
// walk launch and a backward-walk launch) per pass of update_model.
//
// Replaces the body of AgentPPO.update_model's iteration (algos/multiagent/ppo.py:1062-1128: the PFGRU unrolled through every
// episode, the regression + ELBO loss on the mean prediction and on every particle's prediction, loss.backward()) that
// radiation_ppo_amd/rada2c.py: RNNAgentPPO.model_loss composes from torch ops (~24 000 small launches per pass through autograd).
// Cell arithmetic: PFGRUCell.forward (NeuralNetworkCores/RADA2C_core.py / algos/test_cnn/RADTEAM_core.py:1586-1631), soft
// resampling :1466-1515, reparameterize :1517-1530, as in K11 (rs_pfgru.hip).
//
// Mapping: one wave per episode, one particle per lane (40 of 64; lane 40 carries the weighted-mean "particle" through hid_obs).
// The forward kernel's wave runs the episode forward, storing every step's resampled particle set (3.9 KB per step) and the step's gates
// z, r, n and eps * softplus'(var) (15 KB per step; round 3 -- round 2 recomputed them from the stored input state, 21 % of the kernel),
// then the backward kernel's wave walks it backwards: each step's gates are reloaded, the loss terms of the step are formed
// and differentiated in registers, and the gradient flows to the previous step's particles through the resampling gather (an
// LDS scatter-add) and the gates.  The forward walk's matrix products use wave-uniform weights through the scalar unit (as K11 / K12);
// the backward walk's (three transposed products and hid_obs' forward product) run on the matrix cores (mvt_fetch / mvt_mfma below);
// the weight gradients are sums over particles of outer products: the per-particle factors are staged transposed in LDS and
// accumulated on the matrix cores (v_mfma_f32_16x16x4_f32, contraction over the particles) in registers for the whole episode,
// the two thin ones (fc_obs, hid_obs[2]) as rows of one more tile each.  Output: one gradient slab and one
// weighted loss per episode; the caller sums them (fixed order).
//
// Resampling indices are constants of the backward pass, as in autograd (torch.searchsorted / multinomial have no gradient).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radsearch.h"
#include "rs_wave.hpp"
#include "rs_sstream.hpp"

// Diagnostic build only (-DRS_K13_STAMPS, scripts/k13_stamps.py): s_memtime stamps at the phase boundaries of a step, summed per wave
// and added to a device table at the end of the episode.  Read the SHARES (the stamps serialise what the product kernel overlaps).
#ifdef RS_K13_STAMPS
__device__ unsigned long long rs_k13_cyc[16];
#define K13_DECL unsigned long long k13_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long k13_t = __builtin_amdgcn_s_memtime();
#define K13_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long k13_n = __builtin_amdgcn_s_memtime(); k13_acc[i] += k13_n - k13_t; k13_t = k13_n; __builtin_amdgcn_sched_barrier(0); }
#define K13_FLUSH if (threadIdx.x == 0) { for (int q = 0; q < 12; ++q) atomicAdd(&rs_k13_cyc[q], k13_acc[q]); }
#else
#define K13_DECL
#define K13_STAMP(i)
#define K13_FLUSH
#endif

namespace {

constexpr int P = RS_PFGRU_PARTICLES, H = RS_PFGRU_HIDDEN, IN = 3;      // 40, 24, 3
// packed weights (floats; packer: radiation_ppo_amd/rada2c.py: pack_train_weights)
constexpr int T_ZR = 0;                    // [28][48] k-major [h | x | 0] -> [z | r]
constexpr int T_ZRB = T_ZR + 28 * 48;      // [48]
constexpr int T_N = T_ZRB + 48;            // [28][48] k-major [r*h | x | 0] -> [mu | var]
constexpr int T_NB = T_N + 28 * 48;        // [48]
constexpr int T_O = T_NB + 48;             // fc_obs: w [27], b [1], pad to 32
constexpr int T_H0 = T_O + 32;             // hid_obs[0] k-major [24][32] (columns 24..31 zero)
constexpr int T_H0B = T_H0 + 24 * 32;      // [32]
constexpr int T_H2 = T_H0B + 32;           // hid_obs[2]: w [2][24], b [2], pad to 64
constexpr int T_STRIDE = T_H2 + 64;        // 14.7 KB: the whole set stays resident in the 16 KB scalar data cache.  A first version kept
                                           // transposed copies for the backward products (30 KB): every s_load then missed and a pass took twice as long
static_assert(T_STRIDE == RS_PFGRU_TRAIN_WEIGHT_FLOATS, "include/radsearch.h: RS_PFGRU_TRAIN_WEIGHT_FLOATS");
// gradient slab (floats): dW_zr [48][28] (column 27 = bias) | dW_n [48][28] | d hid_obs[0] [24][25] | d hid_obs[2] [2][25] | d fc_obs [28]
constexpr int G_ZR = 0, G_N = G_ZR + 48 * 28, G_H0 = G_N + 48 * 28, G_H2 = G_H0 + 24 * 25, G_O = G_H2 + 50, G_END = G_O + 28;
static_assert(G_END <= RS_PFGRU_TRAIN_GRAD_FLOATS, "include/radsearch.h: RS_PFGRU_TRAIN_GRAD_FLOATS");

#ifndef RS_PF_AHEAD
#define RS_PF_AHEAD 2                     // weight rows requested ahead of the row in use (16 SGPRs each)
#endif

typedef const float __attribute__((address_space(4))) * cmem_t;
__device__ __forceinline__ cmem_t as_cmem(const float* p) { return (cmem_t)(uintptr_t)p; }
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_max(float v) { return rs_wave_max(v); }     // csrc/rs_wave.hpp: DPP rows + v_readlane, no LDS
__device__ __forceinline__ float wave_sum(float v) { return rs_wave_sum(v); }

// The scalar-unit weight streams (wait -> request -> FMA blocks): csrc/rs_sstream.hpp
template <int K, int OUTP, int OUTR = OUTP, typename F>
__device__ __forceinline__ void mv(cmem_t W, F cval, float (&out)[OUTP]) { rs_ss_mv<K, OUTP, OUTR>(W, cval, out); }

// value of lane L (a constant) in every lane: v_readlane_b32 (one instruction into an SGPR) instead of __shfl's LDS permute
template <int L>
__device__ __forceinline__ float lane_bcast(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); }

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
// splitmix64 finaliser on wrapping 64-bit arithmetic == pfgru.py: hash_bits == csrc/rs_pfgru.hip: pf_hash
__device__ __forceinline__ uint64_t tr_hash(uint64_t key) {
    uint64_t x = key * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// Both walks' exponentials, logarithms and f32 quotients on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each, as
// K11's cell): the library expf / logf / IEEE division are 10-20 instructions each, ~200 of a step's 5 000 at the lone-wave issue rate
__device__ __forceinline__ float bw_exp(float x) { return __builtin_amdgcn_exp2f(1.44269504f * x); }
__device__ __forceinline__ float bw_log(float x) { return 0.69314718f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float bw_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// gates and candidate state of one particle: z, r, n, es = eps * d softplus(var) / d var, h1 = (1 - z) n + z h0, lg = fc_obs([h1, x])
// One step of the cell for the forward walk: h1 and the observation logit; the gates z | r | n | eps * softplus'(var) go straight to
// `gw` (what the backward walk reloads: 384 B per particle instead of recomputing both gate products and their transcendentals, which
// was 21 % of the fused kernel) as soon as they exist, so that only z and r * h0 stay live across the second product.
__device__ __forceinline__ void pf_cell(cmem_t W, const float (&h0)[H], const float (&x)[IN], const float (&eps)[H], float4* gw, bool store,
                                        float (&h1)[H], float& lg) {
    float g[48];
#pragma unroll
    for (int o = 0; o < 48; ++o) g[o] = W[T_ZRB + o];
    mv<27, 48>(W + T_ZR, [&](int k) -> float { return k < H ? h0[k < H ? k : 0] : (k < H + IN ? x[(k >= H && k < H + IN) ? k - H : 0] : 0.0f); }, g);
    float z[H], rh[H];
#pragma unroll
    for (int u = 0; u < H; u += 4) {
        float r4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { z[u + j] = sigmoidf_(g[u + j]); r4[j] = sigmoidf_(g[H + u + j]); rh[u + j] = r4[j] * h0[u + j]; }
        if (store) {
            gw[(u / 4) * P] = make_float4(z[u], z[u + 1], z[u + 2], z[u + 3]);
            gw[(6 + u / 4) * P] = make_float4(r4[0], r4[1], r4[2], r4[3]);
        }
    }
    float m[48];
#pragma unroll
    for (int o = 0; o < 48; ++o) m[o] = W[T_NB + o];
    mv<27, 48>(W + T_N, [&](int k) -> float { return k < H ? rh[k < H ? k : 0] : (k < H + IN ? x[(k >= H && k < H + IN) ? k - H : 0] : 0.0f); }, m);
    lg = W[T_O + 27];
#pragma unroll
    for (int u = 0; u < H; u += 4) {
        float n4[4], es4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float var = m[H + u + j];
            const float e = __builtin_amdgcn_exp2f(1.44269504f * var);
            const bool big = var > 20.0f;                                                // F.softplus: identity (slope 1) beyond 20
            const float sp = big ? var : 0.69314718f * __builtin_amdgcn_logf(1.0f + e);
            es4[j] = big ? eps[u + j] : eps[u + j] * (1.0f - __builtin_amdgcn_rcpf(1.0f + e));
            const float y = m[u + j] + eps[u + j] * sp;
            n4[j] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * y));      // tanh
            h1[u + j] = (1.0f - z[u + j]) * n4[j] + z[u + j] * h0[u + j];
            lg = fmaf(W[T_O + u + j], h1[u + j], lg);
        }
        if (store) {
            gw[(12 + u / 4) * P] = make_float4(n4[0], n4[1], n4[2], n4[3]);
            gw[(18 + u / 4) * P] = make_float4(es4[0], es4[1], es4[2], es4[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < IN; ++k) lg = fmaf(W[T_O + H + k], x[k], lg);
}

struct TrArgs {
    const float* w;           // [T_STRIDE]
    const float* obs;         // [L][E][11] (columns 0..2 feed the PFGRU)
    const float* tar;         // [L][E][2]  source location / area scale
    const float* bp;          // [L][E]     normalised exp(bp_decay * t) weights of the episode's steps
    const int64_t* lens;      // [E]
    const float* w_ep;        // [E]        weight of the episode's loss
    const float* h0;          // [E][P][H]  initial particles
    const float* eps;         // [L][E][P][H]
    const double* u;          // [L][E][P]  resampling uniforms; NULL: idx[] holds the indices to take (recorded draws)
    const int64_t* keys;      // [E] or NULL.  Given (rs_pfgru_train_keyed): h0 / eps / u are NOT read -- the forward walk evaluates the counter hash
                              // of rs_pfgru_draws itself (same keys, same arithmetic, bit-identical values; csrc/rs_pfgru.hip: rs_pfgru_draws_kernel)
    float* hs;                // [L][E][H / 4][P] float4 scratch: resampled particles after every step, quad-major
    float* ps;                // [L][E][P]    scratch: their log weights
    float* gates;             // [L][E][4 H / 4][P] float4 scratch: z | r | n | eps * softplus'(var) of the forward walk, QUAD-major (quad j of all 40
                              // particles contiguous: the forward's stores are 640 B runs and the backward's LDS reads are conflict free; with a
                              // particle's 96 floats contiguous every lane of the backward read hit the same four banks)
    int32_t* idx;             // [L][E][P]    resampling indices (output; constants of the backward pass)
    float* loss;              // [E]
    float* grads;             // [E][RS_PFGRU_TRAIN_GRAD_FLOATS]
    int L, E;
    float alpha, floor_, l2w, l1w, elbo;
};

constexpr int ROW = H + 1;                 // 25: odd stride (forward walk's particle rows)
constexpr int BROW = 33;                   // backward walk: the matrix-core products hand back 2 x 3 tiles of 16 x 16 = 32 units x 48 particle
constexpr int TILE_F = 48 * BROW;          // columns; [48][33] takes ALL of them, so their 24 stores per product need no predicate (with [44][25]
                                           // each tile's stores sat behind an exec-mask save / branch / restore: ~120 scalar instructions per step)
constexpr int SP = 45;                     // staging row stride: 44 particle columns + 1
constexpr int DT_F = 48 * SP, IT_F = 32 * SP;
constexpr int PRE_F = P * 4 * H;           // the next backward step's gates, fetched by LDS-DMA while the current step computes (15 KB)
constexpr int TAKERS_F = P * P / 4;         // [P][P] bytes: the lanes that resampled a particle (backward walk)
constexpr int LDS_FLOATS = TILE_F + 2 * P /* forward: cdf (f64); backward: taker counts */ + 64 + DT_F + IT_F + PRE_F + TAKERS_F;

// acc[ti][tj] += D^T I over the particles: DT [16 TI][SP] (row = output unit, column = particle), IT [16 TJ][SP]
template <int TI, int TJ, int KS = 11>          // KS k-steps of 4 particle columns: 11 with the mean "particle" (column 40), 10 without
__device__ __forceinline__ void outer_acc(const float* DT, const float* IT, f4 (&acc)[TI][TJ], int lane) {
    const int c = lane & 15, q = lane >> 4;
    // the fragments of k-step s + 1 are read from LDS while the MFMAs of step s run: at one wave per SIMD every read that is waited for
    // right behind its issue costs an LDS round trip (~60 cycles, 55 k-steps per episode-step)
    float an[TI], bn[TJ];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) an[ti] = DT[(16 * ti + c) * SP + q];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) bn[tj] = IT[(16 * tj + c) * SP + q];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float a[TI], b[TJ];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) a[ti] = an[ti];
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) b[tj] = bn[tj];
        if (s + 1 < KS) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) an[ti] = DT[(16 * ti + c) * SP + 4 * (s + 1) + q];
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) bn[tj] = IT[(16 * tj + c) * SP + 4 * (s + 1) + q];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, TI + TJ, 0);      // the next step's DS reads go out first ...
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TI * TJ, 0);      // ... then this step's MFMAs
    }
}

template <int TI, int TJ>
__device__ __forceinline__ void outer_store(const f4 (&acc)[TI][TJ], float* out, int rows, int cols, int lane) {
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * ti + 4 * q + v, col = 16 * tj + c;
                if (row < rows && col < cols) out[row * cols + col] = acc[ti][tj][v];
            }
}

// Transposed products of the backward walk on the matrix cores: out[u] (u < 24, for the lane's particle) = sum over o < 4 KK of
// Wt[u][o] d[o][particle], with d already staged in DT for the weight gradient's outer product (DT [o][SP], column = particle) and the
// weights k-major in the packed set (Wt[u][o] = wt[u * WS + o]).  D [u][particle] = A [u][o] B [o][particle]: A fragments are per-lane
// global loads of the weights (L1 / L2 hits; requested by mvt_fetch BEFORE the outer product that precedes the product, so their latency
// is covered), B fragments are LDS reads of DT, and the 4 consecutive units a lane receives for one particle go through `tile` back to
// the particle's own lane.  As 1 152 scalar-streamed FMAs at the lone-wave issue rate one product took ~5 700 cycles; 72 MFMAs take ~2 300.
// Rows u >= 24 (clamped to row 23) and columns >= 44 (whatever follows DT) are computed and dropped: an element of D depends on its own
// row of A and column of B only.
template <int KK, int WS>
__device__ __forceinline__ void mvt_fetch(const float* wt, float (&wa)[2][KK], int lane) {
    typedef const __attribute__((address_space(1))) float* gptr_t;
    const int c = lane & 15, q = lane >> 4;
    gptr_t r0 = (gptr_t)wt + c * WS + q;
    gptr_t r1 = (gptr_t)wt + (c < 8 ? 16 + c : H - 1) * WS + q;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) { wa[0][kk] = r0[4 * kk]; wa[1][kk] = r1[4 * kk]; }
    __builtin_amdgcn_sched_group_barrier(0x020, 2 * KK, 0);          // all requests go out HERE: left alone, the scheduler sinks each
}                                                                    // load to its MFMA and waits for it there (24 exposed round trips)

// The same for a FORWARD product out[o] = b[o] + sum over k < 4 KK of W[k][o] in[k][particle] (k-major table, row stride WS, 32 real or
// zero columns; `in` staged like DT): A [o][k] = wt[k * WS + o], the bias rides in as the accumulators' initial value (bias4: the 4
// consecutive outputs a lane holds per tile row block).
template <int KK, int WS>
__device__ __forceinline__ void mv_fetch(const float* wt, const float* bias, float (&wa)[2][KK], f4 (&b4)[2], int lane) {
    typedef const __attribute__((address_space(1))) float* gptr_t;
    typedef const __attribute__((address_space(1))) f4* gptr4_t;
    const int c = lane & 15, q = lane >> 4;
    gptr_t r0 = (gptr_t)wt + q * WS + c;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) { wa[0][kk] = r0[4 * kk * WS]; wa[1][kk] = r0[4 * kk * WS + 16]; }
    b4[0] = *((gptr4_t)bias + q);
    b4[1] = *((gptr4_t)bias + 4 + q);
    __builtin_amdgcn_sched_group_barrier(0x020, 2 * KK + 2, 0);
}

template <int KK, bool ACC>
__device__ __forceinline__ void mvt_mfma(const float (&wa)[2][KK], const float* DT, float* tile, int lane, float (&out)[H],
                                         f4 init0 = f4{0, 0, 0, 0}, f4 init1 = f4{0, 0, 0, 0}) {
    const int c = lane & 15, q = lane >> 4;
    f4 acc[2][3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) { acc[0][nt] = init0; acc[1][nt] = init1; }
    float bn[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) bn[nt] = DT[q * SP + 16 * nt + c];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        float b[3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) b[nt] = bn[nt];
        if (kk + 1 < KK) {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) bn[nt] = DT[(4 * (kk + 1) + q) * SP + 16 * nt + c];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][kk], b[nt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int pp = 16 * nt + c, u0 = 16 * mt + 4 * q;
#pragma unroll
            for (int v = 0; v < 4; ++v) tile[pp * BROW + u0 + v] = acc[mt][nt][v];
        }
    __builtin_amdgcn_wave_barrier();
    const float* row = tile + (lane < 48 ? lane : 47) * BROW;
#pragma unroll
    for (int u = 0; u < H; ++u) out[u] = ACC ? out[u] + row[u] : row[u];
    __builtin_amdgcn_wave_barrier();
}

// The forward walk of a training pass as its own launch (round 3).  Inside the 512-register backward kernel it ran at one wave per SIMD,
// where a lone wave issues a VALU instruction only every ~5 cycles (scripts/micro/sload_latency.hip: 32 v_fmac = 180 cycles; two or
// more waves share the SIMD at 2.5): the walk needs no gradient accumulators, fits 2 waves per SIMD without spilling (175 VGPRs; at 3 waves 9
// registers spill and the pass is no faster: 16.9 against 16.5 ms for 16 384 episodes) and stores everything the backward walk reads (particle sets, log-weights, resampling indices, gates) exactly as
// the fused kernel did.  Lane-packed like K11 (csrc/rs_pfgru.hip): six episodes of 40 particles per 256-thread workgroup, what couples
// an episode's particles goes through LDS and workgroup barriers, every lane reducing its episode's 40 values in index order.
constexpr int FW_SETS = 6, FW_NT = 256;
#ifndef K13_FW_OCC
#define K13_FW_OCC 2          // waves per SIMD the register budget is cut for.  A/B at 16 384 episodes: 2 (179 VGPRs) 51.6 ms per pass, 3 (168 + 52 B scratch) 51.9, 4 (128 + 220 B) 55.2
#endif
constexpr int FW_TILE = 0, FW_CDF = P * ROW, FW_VA = FW_CDF + 2 * P, FW_VB = FW_VA + P, FW_VC = FW_VB + P, FW_STRIDE = 1228;
static_assert(FW_VC + P <= FW_STRIDE && FW_STRIDE % 4 == 0 && FW_STRIDE % 32 == 12 && FW_CDF % 2 == 0, "LDS layout of an episode's particle set");

__device__ __forceinline__ float fw_max40(const float* v) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < P / 4; ++i) {
        const float4 t = reinterpret_cast<const float4*>(v)[i];
        m = fmaxf(fmaxf(m, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
    }
    return m;
}
__device__ __forceinline__ float fw_sum40(const float* v) {          // index order
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < P / 4; ++i) {
        const float4 t = reinterpret_cast<const float4*>(v)[i];
        s = (((s + t.x) + t.y) + t.z) + t.w;
    }
    return s;
}

// KEYED: the pass's draws (initial particles, reparameterisation noise, resampling uniforms) are hashed here instead of read: the draws
// launch wrote 8.2 GB per pass at 16.5 k episodes (2.3 ms) for this kernel to read back -- K11 has always hashed in the kernel
template <bool KEYED>
__global__ void __launch_bounds__(FW_NT, K13_FW_OCC) rs_pfgru_train_fwd_kernel(TrArgs a_) {
    __shared__ __align__(16) float smem[FW_SETS * FW_STRIDE];
    __shared__ int lens_s[FW_SETS];
    const int tid = threadIdx.x;
    const int set = tid / P, q = tid - set * P;                       // set == FW_SETS: the 16 lanes that carry nothing
    const int E = a_.E;
    const int e_raw = blockIdx.x * FW_SETS + set;
    const bool has = set < FW_SETS && e_raw < E;
    const int e = has ? e_raw : E - 1;                                // lanes without an episode shadow the last one (nothing is stored)
    const int len = has ? (int)a_.lens[e] : 0;
    if (tid < FW_SETS) lens_s[tid] = 0;
    __syncthreads();
    if (has && q == 0) lens_s[set] = len;
    __syncthreads();
    int lmax = 0;
#pragma unroll
    for (int i = 0; i < FW_SETS; ++i) lmax = max(lmax, lens_s[i]);   // the workgroup walks to its longest episode (sorted batches: equal lengths)
    float* S = smem + (set < FW_SETS ? set : FW_SETS - 1) * FW_STRIDE;
    float* tile = S + FW_TILE;
    double* cdf = reinterpret_cast<double*>(S + FW_CDF);
    float *va = S + FW_VA, *vb = S + FW_VB, *vc = S + FW_VC;
    const bool act = set < FW_SETS;                                   // LDS writes allowed (the set's own area)
    K13_DECL
    auto wptr = [&]() -> cmem_t { const float* w = a_.w; asm volatile("" : "+s"(w)); return as_cmem(w); };
    const float al = a_.alpha, floor_ = a_.floor_;
    const size_t PH = (size_t)P * H;
    auto load24 = [&](const float* src, float (&dst)[H]) {
#pragma unroll
        for (int u = 0; u < H; u += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + u);
            dst[u] = v.x; dst[u + 1] = v.y; dst[u + 2] = v.z; dst[u + 3] = v.w;
        }
    };
    float h0[H];
    uint64_t kb = 0;
    if constexpr (KEYED) {
        kb = (uint64_t)a_.keys[e] * 1000003ull;
        const uint64_t k0 = kb * 1048583ull + (uint64_t)q * 4096ull;                      // kind 0, step 0
#pragma unroll
        for (int u = 0; u < H; ++u) h0[u] = (float)((double)(tr_hash(k0 + (uint64_t)u) >> 11) * (1.0 / 9007199254740992.0));
        if (has) {                                                    // the backward walk's last step reads the initial particles
            float4* dst = reinterpret_cast<float4*>(const_cast<float*>(a_.h0) + ((size_t)e * P + q) * H);
#pragma unroll
            for (int u = 0; u < H; u += 4) dst[u / 4] = make_float4(h0[u], h0[u + 1], h0[u + 2], h0[u + 3]);
        }
    } else {
        load24(a_.h0 + ((size_t)e * P + q) * H, h0);
    }
    float p0 = -3.6888794541139363f;                                 // float32(log(1 / 40))
    for (int t = 0; t < lmax; ++t) {
        const bool on = t < len;                                      // this episode is still running (uniform over its 40 lanes)
        const int tc = on ? t : (len > 0 ? len - 1 : 0);              // finished episodes re-read their last step (nothing is stored)
        const cmem_t W = wptr();
        const size_t te = (size_t)tc * E + e;
        float x[IN], eps[H], h1[H], lg;
        {
            const float* o = a_.obs + te * RS_OBS_DIM;
#pragma unroll
            for (int k = 0; k < IN; ++k) x[k] = o[k];
        }
        if constexpr (KEYED) {
            const uint64_t k1 = (kb ^ ((uint64_t)(tc * 8 + 1) * 0xA24BAED4963EE407ull)) * 1048583ull + (uint64_t)q * 4096ull;
#pragma unroll
            for (int u = 0; u < H; u += 2) {                          // one hash per pair of units: Box-Muller's cosine and sine
                const uint64_t hx = tr_hash(k1 + (uint64_t)u);
                const float u1 = (float)((uint32_t)(hx >> 40) + 1u) * (1.0f / 16777216.0f);
                const float u2 = (float)((uint32_t)(hx >> 16) & 0xFFFFFFu) * (1.0f / 16777216.0f);
                const float r = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u1));
                eps[u] = r * __builtin_amdgcn_cosf(u2);
                eps[u + 1] = r * __builtin_amdgcn_sinf(u2);
            }
        } else {
            load24(a_.eps + te * PH + (size_t)q * H, eps);
        }
        pf_cell(W, h0, x, eps, reinterpret_cast<float4*>(a_.gates + te * (size_t)(P * 4 * H)) + q, on, h1, lg);
        K13_STAMP(0)                                                 // forward: loads + cell
        lg += p0;
        if (act) va[q] = lg;
        __syncthreads();                                             // 1
        const float mx = fw_max40(va);
        const float e1 = bw_exp(lg - mx);
        if (act) vb[q] = e1;
        __syncthreads();                                             // 2
        const float p1 = (lg - mx) - bw_log(fw_sum40(vb));
        if (act) {
            va[q] = al * bw_exp(p1) + floor_;
            vc[q] = p1;
#pragma unroll
            for (int u = 0; u < H; ++u) tile[q * ROW + u] = h1[u];
        }
        __syncthreads();                                             // 3
        int idx = 0;
        if (KEYED || a_.u) {
            double run = 0.0, mine = 0.0;                            // float64 prefix sums in index order
#pragma unroll
            for (int i = 0; i < P / 4; ++i) {
                const float4 w4 = reinterpret_cast<const float4*>(va)[i];
                run += (double)w4.x; mine = (4 * i == q) ? run : mine;
                run += (double)w4.y; mine = (4 * i + 1 == q) ? run : mine;
                run += (double)w4.z; mine = (4 * i + 2 == q) ? run : mine;
                run += (double)w4.w; mine = (4 * i + 3 == q) ? run : mine;
            }
            if (act) cdf[q] = mine / run;
            __syncthreads();                                         // 4 (a_.u is a launch argument: every lane takes this branch or none)
            double ru;
            if constexpr (KEYED) {
                const uint64_t k2 = (kb ^ ((uint64_t)(tc * 8 + 2) * 0xA24BAED4963EE407ull)) * 1048583ull + (uint64_t)q * 4096ull;
                ru = (double)(tr_hash(k2) >> 11) * (1.0 / 9007199254740992.0);
            } else {
                ru = a_.u[te * P + q];
            }
#pragma unroll
            for (int j = 0; j < P / 2; ++j) {                        // searchsorted(..., right=True)
                const double2 c2 = reinterpret_cast<const double2*>(cdf)[j];
                idx += (c2.x <= ru) ? 1 : 0;
                idx += (c2.y <= ru) ? 1 : 0;
            }
            idx = min(idx, P - 1);
        } else {
            idx = min(max(a_.idx[te * P + q], 0), P - 1);            // what torch.multinomial returned in the reference's run
        }
#pragma unroll
        for (int u = 0; u < H; ++u) h0[u] = tile[idx * ROW + u];
        float pn = bw_exp(vc[idx]);
        pn = bw_log(bw_div(pn, al * pn + floor_));
        if (act) vb[q] = pn;
        __syncthreads();                                             // 5
        const float mx2 = fw_max40(vb);
        const float e2 = bw_exp(pn - mx2);
        if (act) va[q] = e2;
        __syncthreads();                                             // 6
        p0 = pn - (bw_log(fw_sum40(va)) + mx2);
        if (on) {
            float4* hw = reinterpret_cast<float4*>(a_.hs + te * PH) + q;          // quad-major, as the gates: quad j of all particles contiguous
#pragma unroll
            for (int u = 0; u < H; u += 4) hw[(u / 4) * P] = make_float4(h0[u], h0[u + 1], h0[u + 2], h0[u + 3]);
            a_.ps[te * P + q] = p0;
            a_.idx[te * P + q] = idx;
        }
        __syncthreads();                                             // 7: va / tile are rewritten at the top of the next step
        K13_STAMP(1)                                                 // forward: softmax, resampling, stores
    }
    K13_FLUSH
}

__global__ void __launch_bounds__(64) rs_pfgru_train_kernel(TrArgs a_) {
    __shared__ __align__(16) float smem[LDS_FLOATS];
    const int lane = threadIdx.x;
    const int e = blockIdx.x;
    float* tile = smem;                                              // [48][BROW]
    float* vec = smem + TILE_F + 2 * P;                              // [64]
    float* DT = vec + 64;                                            // [48][SP]
    float* IT = DT + DT_F;                                           // [32][SP]
    float* pre = IT + IT_F;                                          // [P][4 H] gates of one step, flat as in HBM
    int* cnt = reinterpret_cast<int*>(smem + TILE_F);                // [P] how many lanes resampled particle p
    unsigned char* takers = reinterpret_cast<unsigned char*>(pre + PRE_F);        // [P][P] which ones
    for (int i = lane; i < DT_F + IT_F; i += 64) DT[i] = 0.0f;       // rows 28..31 of IT stay zero for the whole kernel

    const bool act = lane < P;
    const int pl = act ? lane : P - 1;                               // idle lanes shadow the last particle (values discarded)
    K13_DECL
    // the weight pointer is laundered once per time step and again before every transposed product (wptr): the weights are loop
    // invariant and the forward and transposed products read the same rows -- LICM / GVN would otherwise hoist or keep thousands of
    // scalar loads and spill them to VGPR lanes (measured: 3 920 SGPR spills, 24 k v_readlane)
    auto wptr = [&]() -> cmem_t { const float* w = a_.w; asm volatile("" : "+s"(w)); return as_cmem(w); };
    auto wglob = [&]() -> const float* { const float* w = a_.w; asm volatile("" : "+s"(w)); return w; };      // the same, for per-lane (vector) loads
    const int E = a_.E;
    const int len = (int)a_.lens[e];
    const float al = a_.alpha, floor_ = a_.floor_;
    const size_t PH = (size_t)P * H;

    auto load24 = [&](const float* src, float (&dst)[H]) {
#pragma unroll
        for (int u = 0; u < H; u += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + u);
            dst[u] = v.x; dst[u + 1] = v.y; dst[u + 2] = v.z; dst[u + 3] = v.w;
        }
    };
    auto load_x = [&](int t, float (&x)[IN]) {
        const cmem_t o = as_cmem(a_.obs + ((size_t)t * E + e) * RS_OBS_DIM);
#pragma unroll
        for (int k = 0; k < IN; ++k) x[k] = o[k];
    };

    // ------------------------------------------------------------------------------------------ backward through the episode
    f4 accZR[3][2], accN[3][2], accH0[2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accZR[i][j] = f4{0, 0, 0, 0}; accN[i][j] = f4{0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accH0[i][j] = f4{0, 0, 0, 0};
    // the two thin gradients (hid_obs[2]: 2 x 25, fc_obs: 28) ride on the matrix cores as well: row 0 (rows 0, 1) of one more
    // 16-row tile each.  As per-lane FMA accumulators they were 78 registers of a kernel that already holds 512: the compiler spilled
    // exactly that much (316 B of scratch per lane, written and re-read every step: +75 % HBM write traffic in the PMC pass)
    f4 accW2[1][2], accO[1][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { accW2[0][j] = f4{0, 0, 0, 0}; accO[0][j] = f4{0, 0, 0, 0}; }
    float dh[H], dp = 0.0f;
#pragma unroll
    for (int u = 0; u < H; ++u) dh[u] = 0.0f;
    const float G = a_.w_ep[e];
    const float inv_nel = 1.0f / (float)(2 * len);
    const float l2w = a_.l2w, l1w = a_.l1w, elbo = a_.elbo;
    float l2s = 0.0f, l1s = 0.0f, l2ps = 0.0f, l1ps = 0.0f;          // the episode's four loss terms (wave-uniform)
    // staging column of this lane: lanes 44..63 all write the rows' PAD column (44: read by nothing but the transposed products' dropped
    // output columns), so no staging store sits behind an exec-mask save / restore
    const int scol = lane < 44 ? lane : 44;

    // a step's gates (40 x 96 floats, contiguous in HBM) -> `pre`, 15 global_load_lds_dwordx4: issued for step t - 1 as soon as step
    // t's gates are in registers, so the 15 KB arrive under the ~90 k cycles of the step instead of in front of it
    // Four base pairs (global, LDS) x up to four chunks selected by the instruction's immediate offset, which the hardware adds to BOTH
    // addresses: 15 transfers + 8 address instructions.  (One generic -> LDS pointer conversion per transfer came with a null check and a
    // 64-bit add each: ~90 scalar / vector instructions per step at the lone-wave issue rate.)
    typedef __attribute__((address_space(3))) float* lds_f;
    typedef const __attribute__((address_space(1))) float* glb_f;
    const lds_f pre3 = (lds_f)pre;
    auto dma_gates = [&](int t) {
        const float* src = a_.gates + ((size_t)t * E + e) * (size_t)(P * 4 * H) + lane * 4;
        static_assert(PRE_F / 256 == 15, "15 transfers of 256 floats");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const glb_f gp = (glb_f)(src + g * 1024);
            const lds_f lp = pre3 + g * 1024;
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
            if (g < 3) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
        }
    };
    if (len > 0) dma_gates(len - 1);

    for (int t = len - 1; t >= 0; --t) {
        const cmem_t W = wptr();
        const size_t te = (size_t)t * E + e;
        float h0[H], x[IN], z[H], r[H], n[H], es[H], h1[H], lg, p0;
        // everything else the step reads from HBM is requested here, in one round trip
        const int idx = a_.idx[te * P + pl];
        float h1r[H];
        auto load24q = [&](const float* blk, float (&dst)[H]) {             // a particle set stored quad-major ([H / 4][P] float4)
            const float4* src = reinterpret_cast<const float4*>(blk) + pl;
#pragma unroll
            for (int u = 0; u < H; u += 4) {
                const float4 v4 = src[(u / 4) * P];
                dst[u] = v4.x; dst[u + 1] = v4.y; dst[u + 2] = v4.z; dst[u + 3] = v4.w;
            }
        };
        load24q(a_.hs + te * PH, h1r);
        const float ps_t = a_.ps[te * P + pl];
        {
            if (t > 0) {
                load24q(a_.hs + (te - E) * PH, h0);
                p0 = a_.ps[(te - E) * P + pl];
            } else {
                load24(a_.h0 + ((size_t)e * P + pl) * H, h0);      // (a keyed pass: written by the forward walk)
                p0 = -3.6888794541139363f;
            }
            load_x(t, x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the DMA of this step's gates (and the loads above) have landed
            __builtin_amdgcn_wave_barrier();
            {
                const float4* gr = reinterpret_cast<const float4*>(pre) + pl;        // quad j of particle pl: gr[j * P]
#pragma unroll
                for (int u = 0; u < H; u += 4) {
                    const float4 a = gr[(u / 4) * P], b = gr[(6 + u / 4) * P], c = gr[(12 + u / 4) * P], d = gr[(18 + u / 4) * P];
                    z[u] = a.x; z[u + 1] = a.y; z[u + 2] = a.z; z[u + 3] = a.w;
                    r[u] = b.x; r[u + 1] = b.y; r[u + 2] = b.z; r[u + 3] = b.w;
                    n[u] = c.x; n[u + 1] = c.y; n[u + 2] = c.z; n[u + 3] = c.w;
                    es[u] = d.x; es[u + 1] = d.y; es[u + 2] = d.z; es[u + 3] = d.w;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the gates are in registers: `pre` may be overwritten
            __builtin_amdgcn_wave_barrier();
            if (t > 0) dma_gates(t - 1);
            // h1 and the observation logit exactly as pf_cell forms them
            lg = W[T_O + 27];
#pragma unroll
            for (int u = 0; u < H; ++u) {
                h1[u] = (1.0f - z[u]) * n[u] + z[u] * h0[u];
                lg = fmaf(W[T_O + u], h1[u], lg);
            }
#pragma unroll
            for (int k = 0; k < IN; ++k) lg = fmaf(W[T_O + H + k], x[k], lg);
        }
        K13_STAMP(2)                                                 // backward: loads of the step's gates, h1 and logit
        lg += p0;
        const float mx = wave_max(act ? lg : -INFINITY);
        const float se = wave_sum(act ? bw_exp(lg - mx) : 0.0f);
        const float p1 = (lg - mx) - bw_log(se);
        const float pi = act ? bw_exp(ps_t) : 0.0f;

        float wf[2][6];
        f4 bf[2];
        { const float* wg = wglob(); mv_fetch<6, 32>(wg + T_H0, wg + T_H0B, wf, bf, lane); }   // hid_obs[0]'s fragments: requested a phase ahead of their product
        // ---- weighted mean of the resampled particles (unit u summed by lane u)
#pragma unroll
        for (int u = 0; u < H; ++u) if (act) tile[lane * BROW + u] = pi * h1r[u];
        __builtin_amdgcn_wave_barrier();
        {
            const int ul = lane < H ? lane : H - 1;
            float mean = 0.0f;
            for (int q = 0; q < P; ++q) mean += tile[q * BROW + ul];
            __builtin_amdgcn_wave_barrier();
            // staged once for both the forward product below (matrix cores, as the transposed products) and hid_obs[0]'s weight gradient:
            // the particles' columns straight from their lanes, the mean's column (40) from the 24 lanes that summed it.  (Column 40 used to go
            // lane 0..23 -> vec -> lane 40's registers -> IT: 24 LDS reads and 24 selects in every lane.)  Columns 41..43 hold the shadow
            // lanes' particle (finite); their DT columns are zero, see below.
#pragma unroll
            for (int k = 0; k < H; ++k) IT[k * SP + scol] = h1r[k];
            IT[H * SP + scol] = 1.0f;
            if (lane < H) IT[lane * SP + P] = mean;                  // (program order: after lane 40's own store to the same column)
        }
        K13_STAMP(3)                                                 // log-softmax, loads of the resampled set, weighted mean; hid_obs staging
        __builtin_amdgcn_wave_barrier();
        float uu[H];
        mvt_mfma<6, false>(wf, IT, tile, lane, uu, bf[0], bf[1]);
        float out[2] = {W[T_H2 + 48], W[T_H2 + 49]};
#pragma unroll
        for (int k = 0; k < H; ++k) {
            uu[k] = fmaxf(uu[k], 0.0f);
            out[0] = fmaf(W[T_H2 + k], uu[k], out[0]);
            out[1] = fmaf(W[T_H2 + H + k], uu[k], out[1]);
        }
        out[0] = fmaxf(out[0], 0.0f); out[1] = fmaxf(out[1], 0.0f);
        // ---- the step's loss terms and d loss / d out
        const cmem_t tr = as_cmem(a_.tar + te * 2);
        const float bpt = as_cmem(a_.bp + te)[0];
        float dop[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float d = out[c] - tr[c];
            const float e2 = act ? bw_exp(-d * d * bpt) : 0.0f;
            const float y2 = wave_sum(e2) * (1.0f / P);
            float gpart = bw_div(l2w * 2.0f * d * bpt * e2, P * y2);
            l2ps += -bw_log(y2);
            if (l1w != 0.0f) {
                const float e1 = act ? bw_exp(-fabsf(d) * bpt) : 0.0f;
                const float y1 = wave_sum(e1) * (1.0f / P);
                const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
                gpart += bw_div(l1w * 10.0f * sg * bpt * e1, P * y1);
                l1ps += -bw_log(y1);
            }
            const float dm = lane_bcast<P>(d);                           // the mean prediction's error (lane 40)
            const float sgm = dm > 0.0f ? 1.0f : (dm < 0.0f ? -1.0f : 0.0f);
            l2s += dm * dm * bpt;
            l1s += fabsf(dm) * bpt;
            const float gmean = l2w * 2.0f * d * bpt + l1w * 10.0f * sgm * bpt * inv_nel;
            const float dout = act ? G * elbo * gpart * inv_nel : (lane == P ? G * gmean : 0.0f);
            dop[c] = out[c] > 0.0f ? dout : 0.0f;
        }
        K13_STAMP(4)                                                 // hid_obs forward + loss terms
        // ---- hid_obs backwards: thin gradients in per-lane accumulators, hid_obs[0]'s on the matrix cores
        float du[H];
#pragma unroll
        for (int k = 0; k < H; ++k) du[k] = uu[k] > 0.0f ? W[T_H2 + k] * dop[0] + W[T_H2 + H + k] * dop[1] : 0.0f;
        {
#pragma unroll
            for (int k = 0; k < H; ++k) DT[k * SP + scol] = du[k];        // zero in columns 41..43 by construction (dop is); IT still holds [v | 1]
        }
        __builtin_amdgcn_wave_barrier();
        float wh[2][6];
        mvt_fetch<6, 32>(wglob() + T_H0, wh, lane);
        outer_acc<2, 2>(DT, IT, accH0, lane);
        float dv[H];
        mvt_mfma<6, false>(wh, DT, tile, lane, dv);                  // particles and (lane 40) the mean
        // d hid_obs[2] = sum over particles (and the mean) of dop (x) [relu(u) | 1]: rows 0, 1 of a tile (rows 2..15 hold stale du: their
        // products land in accumulator rows that are never stored)
        {
            DT[0 * SP + scol] = dop[0]; DT[1 * SP + scol] = dop[1];                     // dop = 0 beyond lane 40
#pragma unroll
            for (int k = 0; k < H; ++k) IT[k * SP + scol] = uu[k];      // row 24 still holds the ones
        }
        __builtin_amdgcn_wave_barrier();
        outer_acc<1, 2>(DT, IT, accW2, lane);
        __builtin_amdgcn_wave_barrier();
        K13_STAMP(5)                                                 // hid_obs backward: transposed product, two outer products
        // ---- gradient at the resampled particles and their log weights
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const float dmean = lane_bcast<P>(dv[k]);
            dh[k] = dh[k] + dv[k] + pi * dmean;                      // dL / d h1r
            dot = fmaf(dmean, h1r[k], dot);
        }
        const float dp1r = dp + pi * dot;
        const float S = wave_sum(act ? dp1r : 0.0f);
        const float dpn = dp1r - pi * S;                             // through p1r = pn - logsumexp(pn)
        vec[lane] = p1;
        __builtin_amdgcn_wave_barrier();
        const float wj = bw_exp(vec[idx]);
        const float gp = bw_div(dpn * floor_, al * wj + floor_);          // through pn = log(w / (alpha w + floor))
        // ---- back through the gather: every source particle sums the gradients of the lanes that resampled it.  Not by LDS float atomics
        // (25 x ds_add_f32 per lane took ~360 cycles each: 9 000 of the step's 61 000 cycles): one INTEGER atomic per lane builds the list of
        // a source's takers, every lane stores its gradient row, and each source then pulls its takers' rows (reads only, no
        // read-modify-write chain; as many rounds as the most resampled particle has takers)
        if (act) cnt[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (act) {
            const int slot_ = __hip_atomic_fetch_add(&cnt[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            takers[idx * P + slot_] = (unsigned char)lane;
#pragma unroll
            for (int k = 0; k < H; ++k) tile[lane * BROW + k] = dh[k];
            tile[lane * BROW + H] = gp;
        }
        __builtin_amdgcn_wave_barrier();
        float dh1[H];
#pragma unroll
        for (int k = 0; k < H; ++k) dh1[k] = 0.0f;
        float dp1 = 0.0f;
        const int ntk = act ? cnt[pl] : 0;
        for (int s_ = 0; __any(s_ < ntk); ++s_) {
            if (s_ < ntk) {
                const float* row = tile + (int)takers[pl * P + s_] * BROW;
#pragma unroll
                for (int k = 0; k < H; ++k) dh1[k] += row[k];
                dp1 += row[H];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float dlp = dp1 - bw_exp(p1) * wave_sum(dp1);            // through p1 = lp - logsumexp(lp); also d / d p0
        dp = dlp;
        K13_STAMP(6)                                                 // resampling backwards: scatter-add, softmax derivatives
        // ---- fc_obs: d fc_obs = sum over particles of dlp (x) [h1 | x | 1] (row 0 of a tile, as d hid_obs[2] above)
#pragma unroll
        for (int k = 0; k < H; ++k) dh1[k] = fmaf(dlp, W[T_O + k], dh1[k]);
        {
            // (the three particle-only outer products below contract over columns 0..39 only -- KS = 10 -- and a transposed product's
            // output column depends on its own input column alone: columns 40..43 carry whatever the shadow lanes hold, unselected.
            // The selects that zeroed them were ~250 of the step's 5 200 instructions at the lone-wave issue rate.)
            DT[0 * SP + scol] = dlp;
#pragma unroll
            for (int k = 0; k < H; ++k) IT[k * SP + scol] = h1[k];
#pragma unroll
            for (int k = 0; k < IN; ++k) IT[(H + k) * SP + scol] = x[k];
            IT[27 * SP + scol] = 1.0f;
        }
        __builtin_amdgcn_wave_barrier();
        outer_acc<1, 2, 10>(DT, IT, accO, lane);
        __builtin_amdgcn_wave_barrier();
        K13_STAMP(7)                                                 // fc_obs outer product
        // ---- h1 = (1 - z) n + z h0, n = tanh(mu + eps softplus(var))
        // (element-wise chains as explicit pairs -- v_pk_mul_f32 / v_pk_add_f32: a lone wave issues one vector instruction per ~3.4 ns
        // whatever it is, so a packed one is two for the price of one: profiles/r04_valu_cost.txt)
        float dan[48], dz[H];
#pragma unroll
        for (int u = 0; u < H; u += 2) {
            const v2 z2 = {z[u], z[u + 1]}, n2 = {n[u], n[u + 1]}, g2 = {dh1[u], dh1[u + 1]}, h2 = {h0[u], h0[u + 1]}, e2 = {es[u], es[u + 1]};
            const v2 one = {1.0f, 1.0f};
            const v2 dn = g2 * (one - z2);
            const v2 dz2 = g2 * (h2 - n2);
            const v2 dh2 = g2 * z2;                                  // dL / d h0 (direct path)
            const v2 dm = dn * (one - n2 * n2);
            const v2 dv2 = dm * e2;
            dz[u] = dz2.x; dz[u + 1] = dz2.y; dh[u] = dh2.x; dh[u + 1] = dh2.y;
            dan[u] = dm.x; dan[u + 1] = dm.y; dan[H + u] = dv2.x; dan[H + u + 1] = dv2.y;
        }
        {
#pragma unroll
            for (int o = 0; o < 48; ++o) DT[o * SP + scol] = dan[o];
#pragma unroll
            for (int k = 0; k < H; k += 2) {
                const v2 t2 = (v2){r[k], r[k + 1]} * (v2){h0[k], h0[k + 1]};
                IT[k * SP + scol] = t2.x; IT[(k + 1) * SP + scol] = t2.y;
            }
#pragma unroll
            for (int k = 0; k < IN; ++k) IT[(H + k) * SP + scol] = x[k];
            IT[27 * SP + scol] = 1.0f;
        }
        __builtin_amdgcn_wave_barrier();
        float wa[2][12];
        mvt_fetch<12, 48>(wglob() + T_N, wa, lane);                  // the transposed product's weight fragments travel under the outer product
        outer_acc<3, 2, 10>(DT, IT, accN, lane);
        K13_STAMP(8)                                                 // d candidate, staging, outer product N
        float drh[H];
        mvt_mfma<12, false>(wa, DT, tile, lane, drh);
        K13_STAMP(9)                                                 // transposed product N
        // ---- z, r = sigmoid(W_zr [h0, x] + b)
#pragma unroll
        for (int u = 0; u < H; u += 2) {
            const v2 z2 = {z[u], z[u + 1]}, r2 = {r[u], r[u + 1]}, d2 = {drh[u], drh[u + 1]}, h2 = {h0[u], h0[u + 1]}, q2 = {dz[u], dz[u + 1]};
            const v2 one = {1.0f, 1.0f};
            const v2 dh2 = __builtin_elementwise_fma(d2, r2, (v2){dh[u], dh[u + 1]});
            const v2 dr = d2 * h2;
            const v2 az = q2 * z2 * (one - z2);                      // dan now holds d (a_z | a_r)
            const v2 ar = dr * r2 * (one - r2);
            dh[u] = dh2.x; dh[u + 1] = dh2.y;
            dan[u] = az.x; dan[u + 1] = az.y; dan[H + u] = ar.x; dan[H + u + 1] = ar.y;
        }
        {
#pragma unroll
            for (int o = 0; o < 48; ++o) DT[o * SP + scol] = dan[o];
#pragma unroll
            for (int k = 0; k < H; ++k) IT[k * SP + scol] = h0[k];      // rows 24..27 still hold x | 1
        }
        __builtin_amdgcn_wave_barrier();
        mvt_fetch<12, 48>(wglob() + T_ZR, wa, lane);
        outer_acc<3, 2, 10>(DT, IT, accZR, lane);
        K13_STAMP(10)                                                // d gates, staging, outer product ZR
        mvt_mfma<12, true>(wa, DT, tile, lane, dh);                  // dL / d (h1r of step t - 1), particle by particle
        K13_STAMP(11)                                                // transposed product ZR
    }
    K13_FLUSH

    // ------------------------------------------------------------------------------------------ the episode's slab
    float* g = a_.grads + (size_t)e * RS_PFGRU_TRAIN_GRAD_FLOATS;
    outer_store<3, 2>(accZR, g + G_ZR, 48, 28, lane);
    outer_store<3, 2>(accN, g + G_N, 48, 28, lane);
    outer_store<2, 2>(accH0, g + G_H0, 24, 25, lane);
    outer_store<1, 2>(accW2, g + G_H2, 2, 25, lane);
    outer_store<1, 2>(accO, g + G_O, 1, 28, lane);
    if (lane < RS_PFGRU_TRAIN_GRAD_FLOATS - G_END) g[G_END + lane] = 0.0f;        // the slab's padding
    if (lane == 0) {
        const float pred = l2w * l2s + l1w * 10.0f * l1s * inv_nel;
        const float part = (l2w * l2ps + l1w * 10.0f * l1ps) * inv_nel;
        a_.loss[e] = G * (pred + elbo * part);
    }
}

}  // namespace

extern "C" {

#ifdef RS_K13_STAMPS
int rs_debug_k13_stamps(unsigned long long* out, int reset) {
    unsigned long long z[16] = {0};
    if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(rs_k13_cyc), z, sizeof(z)) == hipSuccess ? 0 : 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rs_k13_cyc), sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

static int pfgru_train_launch(const TrArgs& a, int32_t episodes, rs_stream_t stream) {
    if (a.keys) hipLaunchKernelGGL(rs_pfgru_train_fwd_kernel<true>, dim3((unsigned)((episodes + FW_SETS - 1) / FW_SETS)), dim3(FW_NT), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(rs_pfgru_train_fwd_kernel<false>, dim3((unsigned)((episodes + FW_SETS - 1) / FW_SETS)), dim3(FW_NT), 0, static_cast<hipStream_t>(stream), a);
    hipLaunchKernelGGL(rs_pfgru_train_kernel, dim3((unsigned)episodes), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_pfgru_train(const float* weights, const float* obs, const float* target, const float* bp, const int64_t* lens, const float* w_ep,
                   const float* h0, const float* eps, const double* u, float* hs, float* ps, float* gates, int32_t* idx, float* loss,
                   float* grads, int32_t steps, int32_t episodes, double alpha, double l2_weight, double l1_weight, double elbo_weight,
                   rs_stream_t stream) {
    if (!weights || !obs || !target || !bp || !lens || !w_ep || !h0 || !eps || !hs || !ps || !gates || !idx || !loss || !grads || steps < 1 ||
        episodes < 1)
        return RS_ERR_INVALID_ARG;
    TrArgs a{weights, obs, target, bp, lens, w_ep, h0, eps, u, nullptr, hs, ps, gates, idx, loss, grads, steps, episodes, (float)alpha,
             (float)((1.0 - alpha) / (double)P), (float)l2_weight, (float)l1_weight, (float)elbo_weight};
    return pfgru_train_launch(a, episodes, stream);
}

int rs_pfgru_train_keyed(const float* weights, const float* obs, const float* target, const float* bp, const int64_t* lens, const float* w_ep,
                         const int64_t* keys, float* h0, float* hs, float* ps, float* gates, int32_t* idx, float* loss, float* grads, int32_t steps,
                         int32_t episodes, double alpha, double l2_weight, double l1_weight, double elbo_weight, rs_stream_t stream) {
    if (!weights || !obs || !target || !bp || !lens || !w_ep || !keys || !h0 || !hs || !ps || !gates || !idx || !loss || !grads || steps < 1 ||
        episodes < 1)
        return RS_ERR_INVALID_ARG;
    TrArgs a{weights, obs, target, bp, lens, w_ep, h0, nullptr, nullptr, keys, hs, ps, gates, idx, loss, grads, steps, episodes, (float)alpha,
             (float)((1.0 - alpha) / (double)P), (float)l2_weight, (float)l1_weight, (float)elbo_weight};
    return pfgru_train_launch(a, episodes, stream);
}

}  // extern "C"
