// rs_ppo.hip -- policy-side kernels: MFMA forward of the FF_core MLP, the fused on-device collector
// (rs_rollout) and the fused PPO loss/gradient pass (rs_ppo_grad).  See rs_mlp.hpp for the MFMA mapping.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
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


#include "rs_rollout16.hpp"

// ------------------------------------------------------------------------------------------------
// K7: fused PPO loss + gradients for one FF_core network (NOUT = 8: actor, NOUT = 1: critic).
//
// 256-thread workgroups (4 waves, one per SIMD; ~450 registers per lane), one workgroup per CU, grid-stride
// over groups of 64 samples.  Per group and wave: forward (rs_mlp.hpp), per-sample loss derivative on the
// VALU, backward through the output layer on the VALU, then three weight-gradient GEMMs and one
// activation-gradient GEMM on the matrix cores.  The weight-gradient GEMMs contract over SAMPLES, which sit
// on the lanes in accumulator layout, so both operands are transposed through per-wave LDS tiles with row
// stride 65 (conflict-free column writes AND row reads).  Gradient accumulators stay in registers for the
// whole launch; each wave then writes one partial slab and rs_ppo_reduce_kernel sums the slabs in a fixed
// order (bitwise reproducible, no float atomics).
#define RS_TS 65                                  // LDS tile row stride (floats)

__host__ __device__ constexpr int rs_net_params(int nout) { return 64 * 11 + 64 + 64 * 64 + 64 + nout * 64 + nout; }
__host__ __device__ constexpr int rs_grad_lds_floats(int nout) { return rs_mlp_lds_floats(nout) + 2 * 2 * 16 * 64 + 2 * 4 * 64 + 4 * (64 + 32 + 12) * RS_TS; }

__device__ __forceinline__ void rs_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// stage an accumulator-layout half (32 units x 64 samples: v[jt][r]) as T[unit_local][sample]
__device__ __forceinline__ void rs_stage_half(float* T, const f32x16 (&v)[2], int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) T[rs_kappa(r, h) * RS_TS + 32 * jt + j] = v[jt][r];
}

template <int NOUT>
__global__ void __launch_bounds__(256, 1) rs_ppo_grad_kernel(RsMlpParams prm, rs_ppo_batch B, float* __restrict__ partial,
                                                             double* __restrict__ stat_partial, const int* __restrict__ stop) {
    extern __shared__ __align__(16) float smem_f[];
    if (stop && *stop) return;                      // early stop already hit: this iteration is a no-op
    RsMlpLds<NOUT> W;
    W.carve(smem_f);
    float* w2tf = smem_f + rs_mlp_lds_floats(NOUT);                    // [2 it][2 kt][16 r][64]: W2[32kt + kappa][32it + (l&31)]
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    float* w3tf = w2tf + 2 * 2 * 16 * 64;                               // [2 it][4 s][64]: W3[2s + (l>>5)][32it + (l&31)] (actor)
    float* Qt = w3tf + 2 * 4 * 64 + wid * (64 + 32 + 12) * RS_TS;       // [64][65]
    float* Pt = Qt + 64 * RS_TS;                                        // [32][65]
    float* St = Pt + 32 * RS_TS;                                        // [12][65]
    W.fill(prm);
    for (int i = threadIdx.x; i < 2 * 2 * 16 * 64; i += blockDim.x) {
        int l = i & 63, r = (i >> 6) & 15, kt = (i >> 10) & 1, it = i >> 11;
        w2tf[i] = prm.w2[(32 * kt + rs_kappa(r, l >> 5)) * RS_HID + 32 * it + (l & 31)];
    }
    for (int i = threadIdx.x; i < 2 * 4 * 64; i += blockDim.x) {
        int l = i & 63, sq = (i >> 6) & 3, it = i >> 8;
        int o = 2 * sq + (l >> 5);
        w3tf[i] = (o < NOUT) ? prm.w3[o * RS_HID + 32 * it + (l & 31)] : 0.0f;
    }
    __syncthreads();

    const int M = B.M;
    const int groups = (M + 63) / 64;
    const int wave_g = blockIdx.x * 4 + wid, n_waves = gridDim.x * 4;

    // persistent gradient accumulators: dW2 as 2x2 tiles of 32x32 (64 regs); dW1 [64 x 16] and the actor's dW3
    // [NOUT(<=16) x 64] as 4 tiles of 16x16 each (16 + 16 regs, v_mfma_f32_16x16x4_f32); db2 per-lane partial sums (32 regs)
    f32x16 acc2[2][2], db2[2];
    f32x4 acc1[4], acc3[4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[a][0][r] = 0.f; acc2[a][1][r] = 0.f; db2[a][r] = 0.f; }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc1[a][r] = 0.f; acc3[a][r] = 0.f; }
    }
    const int l15 = lane & 15, l4 = lane >> 4;
    float db3[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) db3[o] = 0.f;
    double st_kl = 0.0, st_ent = 0.0, st_cf = 0.0, st_vl = 0.0, st_surr = 0.0;

    for (int gi = wave_g; gi < groups; gi += n_waves) {
        const int m = gi * 64 + lane;
        const bool valid = m < M;
        const int mm = valid ? m : M - 1;
        float xo[RS_IN_PAD], xp[RS_IN_PAD];
#pragma unroll
        for (int k = 0; k < RS_IN; ++k) xo[k] = B.x[(size_t)mm * RS_IN + k];
        xo[11] = 0.0f;
        rs_exchange_x(xo, xp);
        const float wi = valid ? B.w[mm] : 0.0f;

        RsHidden H1, H2;
        rs_mlp_layer1<NOUT>(W, xo, xp, H1);
        rs_mlp_layer2<NOUT>(W, H1, H2);
        float out[NOUT];
        rs_mlp_out<NOUT>(W, H2, out);

        // ---- per-sample loss derivative wrt the network outputs (own sample)
        float dz[NOUT];
        if (NOUT == 8) {
            const int a = (int)B.act[mm];
            const float adv = B.adv[mm], lpo = B.logp_old[mm];
            float mx = out[0];
#pragma unroll
            for (int j = 1; j < NOUT; ++j) mx = fmaxf(mx, out[j]);
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) se += expf(out[j] - mx);
            const float lse = logf(se);
            float lp[NOUT], pj[NOUT], ent = 0.f, logp = 0.f;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                lp[j] = (out[j] - mx) - lse;
                pj[j] = expf(lp[j]);
                ent -= pj[j] * lp[j];
                logp = (a == j) ? lp[j] : logp;
            }
            const float ratio = expf(logp - lpo);
            const float lo = 1.0f - B.clip_ratio, hi = 1.0f + B.clip_ratio;
            const float clipped = fminf(fmaxf(ratio, lo), hi);
            const float s1 = ratio * adv, s2 = clipped * adv;
            const float surr = fminf(s1, s2);
            const bool inside = ratio >= lo && ratio <= hi;
            const float dr = (inside || s1 < s2) ? adv : 0.0f;           // d min(r A, clip(r) A) / dr
            const float g_lp = -wi * dr * ratio;                          // d(-w surr)/d logp
            // alpha * H is a detached scalar in the reference (ppo.py:1216 `.detach().mean().item()`): loss value only
#pragma unroll
            for (int j = 0; j < NOUT; ++j) dz[j] = g_lp * (((a == j) ? 1.0f : 0.0f) - pj[j]);
            st_kl += (double)(wi * (lpo - logp));
            st_ent += (double)(wi * ent);
            st_cf += (double)(wi * ((ratio > hi || ratio < lo) ? 1.0f : 0.0f));
            st_surr += (double)(wi * surr);
        } else {
            const float diff = out[0] - B.ret[mm];
            dz[0] = 2.0f * B.vf_coef * wi * diff;                          // d(vf w (V-R)^2)/dV
            st_vl += (double)(wi * diff * diff);
        }
        float dzp[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) { dzp[o] = __shfl_xor(dz[o], 32); db3[o] += dz[o]; }

        // ================= backward =================
        // Regions R1..R6 are separated by wave-level syncs only where an LDS tile changes hands; inside a
        // region the MFMA chain and the VALU / LDS-staging work are independent, so they overlap (MFMA executes
        // asynchronously; one wave per SIMD has no other wave to hide behind).
        // ---- R1: stage h2^T and dz^T for dW3
        {
            rs_stage_half(Qt, H2.v[0], lane);
            rs_stage_half(Qt + 32 * RS_TS, H2.v[1], lane);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) St[o * RS_TS + lane] = dz[o];
            rs_wave_sync();
        }
        // ---- R2: dW3 += dz . h2^T (matrix cores)  ||  dh2 = W3^T dz, dpre2 = dh2 * (1 - h2^2) (in place of H2)
        {
            // dW3[o][unit] = sum_n dz[o][n] h2[unit][n]: 16x16x4 tiles, A = dz^T tile (rows o < NOUT), B = h2^T tile
            {
                float a_c = (l15 < NOUT) ? St[l15 * RS_TS + l4] : 0.0f;
                float b_c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b_c[u] = Qt[(16 * u + l15) * RS_TS + l4];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    float a_n = 0.f, b_n[4] = {0.f, 0.f, 0.f, 0.f};
                    if (s + 1 < 16) {
                        a_n = (l15 < NOUT) ? St[l15 * RS_TS + 4 * (s + 1) + l4] : 0.0f;
#pragma unroll
                        for (int u = 0; u < 4; ++u) b_n[u] = Qt[(16 * u + l15) * RS_TS + 4 * (s + 1) + l4];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc3[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c, b_c[u], acc3[u], 0, 0, 0);
                    a_c = a_n;
#pragma unroll
                    for (int u = 0; u < 4; ++u) b_c[u] = b_n[u];
                    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        }
        if (NOUT == 8) {
            // dh2[unit][sample] = sum_o W3[o][unit] dz[o][sample]: K = 8 outputs -> 4 k-steps per 32x32 tile
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                f32x16 t0, t1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { t0[r] = 0.f; t1[r] = 0.f; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float own = h ? dz[(2 * s + 1) % NOUT] : dz[(2 * s) % NOUT];
                    const float par = h ? dzp[(2 * s + 1) % NOUT] : dzp[(2 * s) % NOUT];
                    const float a = w3tf[(it * 4 + s) * 64 + lane];
                    t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h ? par : own, t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h ? own : par, t1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float h20 = H2.v[it][0][r], h21 = H2.v[it][1][r];
                    H2.v[it][0][r] = t0[r] * (1.0f - h20 * h20);
                    H2.v[it][1][r] = t1[r] * (1.0f - h21 * h21);
                }
            }
        } else {
            const float z0 = h ? dzp[0] : dz[0];               // sample tile jt = 0 is owned by lanes < 32
            const float z1 = h ? dz[0] : dzp[0];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float wv = W.w3h[h * 32 + kt * 16 + r];
                    const float h20 = H2.v[kt][0][r], h21 = H2.v[kt][1][r];
                    H2.v[kt][0][r] = (wv * z0) * (1.0f - h20 * h20);
                    H2.v[kt][1][r] = (wv * z1) * (1.0f - h21 * h21);
                }
        }
        rs_wave_sync();
        // ---- R3: dh1 = W2^T . dpre2 (register operands)  ||  stage h1^T -> Qt, dpre2[it=0] -> Pt, x -> St
        RsHidden D1;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) { D1.v[it][0][r] = 0.f; D1.v[it][1][r] = 0.f; }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float a_c = w2tf[((it * 2 + 0) * 16 + 0) * 64 + lane];
            float a_n = w2tf[((it * 2 + 0) * 16 + 1) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int kt = q >> 4, r = q & 15;
                float a_nn = 0.f;
                if (q + 2 < 32) a_nn = w2tf[((it * 2 + ((q + 2) >> 4)) * 16 + ((q + 2) & 15)) * 64 + lane];
                D1.v[it][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, H2.v[kt][0][r], D1.v[it][0], 0, 0, 0);
                D1.v[it][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, H2.v[kt][1][r], D1.v[it][1], 0, 0, 0);
                a_c = a_n; a_n = a_nn;
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        rs_stage_half(Qt, H1.v[0], lane);
        rs_stage_half(Qt + 32 * RS_TS, H1.v[1], lane);
        rs_stage_half(Pt, H2.v[0], lane);
#pragma unroll
        for (int k = 0; k < RS_IN; ++k) St[k * RS_TS + lane] = xo[k];
        St[11 * RS_TS + lane] = 1.0f;                           // x[11] := 1 -> column 11 of dW1 is db1
        rs_wave_sync();
        // ---- R4: dW2[it=0] += dpre2[0] . h1^T  ||  dpre1 = dh1 * (1 - h1^2) (in place of D1), db2
        {
            float a_c = Pt[c * RS_TS + h], b0_c = Qt[c * RS_TS + h], b1_c = Qt[(32 + c) * RS_TS + h];
            float a_n = Pt[c * RS_TS + 2 + h], b0_n = Qt[c * RS_TS + 2 + h], b1_n = Qt[(32 + c) * RS_TS + 2 + h];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                float a_nn = 0.f, b0_nn = 0.f, b1_nn = 0.f;
                if (s + 2 < 32) {
                    a_nn = Pt[c * RS_TS + 2 * (s + 2) + h];
                    b0_nn = Qt[c * RS_TS + 2 * (s + 2) + h];
                    b1_nn = Qt[(32 + c) * RS_TS + 2 * (s + 2) + h];
                }
                acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b0_c, acc2[0][0], 0, 0, 0);
                acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b1_c, acc2[0][1], 0, 0, 0);
                a_c = a_n; b0_c = b0_n; b1_c = b1_n;
                a_n = a_nn; b0_n = b0_nn; b1_n = b1_nn;
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                db2[it][r] += H2.v[it][0][r] + H2.v[it][1][r];
                const float h0 = H1.v[it][0][r], h1v = H1.v[it][1][r];
                D1.v[it][0][r] = D1.v[it][0][r] * (1.0f - h0 * h0);
                D1.v[it][1][r] = D1.v[it][1][r] * (1.0f - h1v * h1v);
            }
        rs_wave_sync();
        // ---- R5: dW2[it=1]
        rs_stage_half(Pt, H2.v[1], lane);
        rs_wave_sync();
        {
            float a_c = Pt[c * RS_TS + h], b0_c = Qt[c * RS_TS + h], b1_c = Qt[(32 + c) * RS_TS + h];
            float a_n = Pt[c * RS_TS + 2 + h], b0_n = Qt[c * RS_TS + 2 + h], b1_n = Qt[(32 + c) * RS_TS + 2 + h];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                float a_nn = 0.f, b0_nn = 0.f, b1_nn = 0.f;
                if (s + 2 < 32) {
                    a_nn = Pt[c * RS_TS + 2 * (s + 2) + h];
                    b0_nn = Qt[c * RS_TS + 2 * (s + 2) + h];
                    b1_nn = Qt[(32 + c) * RS_TS + 2 * (s + 2) + h];
                }
                acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b0_c, acc2[1][0], 0, 0, 0);
                acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b1_c, acc2[1][1], 0, 0, 0);
                a_c = a_n; b0_c = b0_n; b1_c = b1_n;
                a_n = a_nn; b0_n = b0_nn; b1_n = b1_nn;
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        // ---- R6: dW1[unit][input] += sum_n dpre1[unit][n] x[input][n]: 16x16x4 tiles (4 unit tiles x 1 input tile)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            rs_wave_sync();
            rs_stage_half(Pt, D1.v[it], lane);
            rs_wave_sync();
            float b_c = (l15 < RS_IN_PAD) ? St[l15 * RS_TS + l4] : 0.0f;
            float a0_c = Pt[l15 * RS_TS + l4], a1_c = Pt[(16 + l15) * RS_TS + l4];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float b_n = 0.f, a0_n = 0.f, a1_n = 0.f;
                if (s + 1 < 16) {
                    b_n = (l15 < RS_IN_PAD) ? St[l15 * RS_TS + 4 * (s + 1) + l4] : 0.0f;
                    a0_n = Pt[l15 * RS_TS + 4 * (s + 1) + l4];
                    a1_n = Pt[(16 + l15) * RS_TS + 4 * (s + 1) + l4];
                }
                acc1[2 * it + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0_c, b_c, acc1[2 * it + 0], 0, 0, 0);
                acc1[2 * it + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1_c, b_c, acc1[2 * it + 1], 0, 0, 0);
                b_c = b_n; a0_c = a0_n; a1_c = a1_n;
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        rs_wave_sync();
    }

    // ---- one partial slab per wave, in the parameter order {w1, b1, w2, b2, w3, b3}
    float* out = partial + (size_t)wave_g * rs_net_params(NOUT);
    float* g_w1 = out, *g_b1 = g_w1 + 64 * 11, *g_w2 = g_b1 + 64, *g_b2 = g_w2 + 64 * 64, *g_w3 = g_b2 + 64, *g_b3 = g_w3 + NOUT * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * it + rs_kappa(r, h);
            g_w2[row * 64 + c] = acc2[it][0][r];
            g_w2[row * 64 + 32 + c] = acc2[it][1][r];
            // db2: sum the 32 lanes that share this (r, h)
            float v = db2[it][r];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            if (c == 0) g_b2[row] = v;
        }
    // dW1 tiles (16x16x4 layout): unit = 16u + 4*(lane>>4) + q, input = lane&15; input 11 carries db1
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * u + 4 * l4 + q;
            if (l15 < RS_IN) g_w1[row * RS_IN + l15] = acc1[u][q];
            if (l15 == RS_IN) g_b1[row] = acc1[u][q];
        }
    // dW3 tiles: output o = 4*(lane>>4) + q (< NOUT), unit = 16u + (lane&15)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = 4 * l4 + q;
            if (o < NOUT) g_w3[o * 64 + 16 * u + l15] = acc3[u][q];
        }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float v = db3[o];
        v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        if (lane == 0) g_b3[o] = v;
    }
    double sv[5] = {st_kl, st_ent, st_cf, st_vl, st_surr};
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double v = sv[q];
        v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        if (lane == 0) stat_partial[(size_t)wave_g * 5 + q] = v;
    }
}

#include "rs_ppo_grad2.hpp"
#include "rs_ppo_grad3.hpp"

// deterministic reduction of the per-wave slabs: a workgroup owns 64 consecutive parameters; 16 thread
// groups each sum a fixed 1/16 of the slabs in order (coalesced 256-byte rows), then the 16 partial sums are
// added in a fixed order through LDS.  Same order every launch -> bitwise reproducible gradients.
__global__ void __launch_bounds__(1024) rs_ppo_reduce_kernel(const float* __restrict__ pa, const float* __restrict__ pc,
                                                             const double* __restrict__ sa, const double* __restrict__ sc, int n_waves,
                                                             float* __restrict__ grads, double* __restrict__ stats, float alpha, float vf,
                                                             const int* __restrict__ stop) {
    __shared__ float part[16][64];
    if (stop && *stop) return;
    __shared__ double spart[5][64];
    constexpr int NA = rs_net_params(8), NC = rs_net_params(1);
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + col;
    float acc = 0.0f;
    if (p < NA + NC) {
        const float* src = (p < NA) ? pa + p : pc + (p - NA);
        const int stride = (p < NA) ? NA : NC;
        const int per = (n_waves + 15) / 16;
        const int w0 = grp * per, w1 = min(w0 + per, n_waves);
        for (int w = w0; w < w1; ++w) acc += src[(size_t)w * stride];
    }
    part[grp][col] = acc;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // statistics: 64 threads each own a fixed subset of the waves, then a fixed-order sum
        double t[5] = {0, 0, 0, 0, 0};
        for (int w = threadIdx.x; w < n_waves; w += 64) {
            t[0] += sa[w * 5 + 0]; t[1] += sa[w * 5 + 1]; t[2] += sa[w * 5 + 2]; t[4] += sa[w * 5 + 4];
            t[3] += sc[w * 5 + 3];
        }
        for (int q = 0; q < 5; ++q) spart[q][threadIdx.x] = t[q];
    }
    __syncthreads();
    if (grp == 0 && p < NA + NC) {
        float v = part[0][col];
        for (int g2 = 1; g2 < 16; ++g2) v += part[g2][col];
        grads[p] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double t[5];
        for (int q = 0; q < 5; ++q) { double v = 0; for (int i = 0; i < 64; ++i) v += spart[q][i]; t[q] = v; }
        stats[0] = t[0]; stats[1] = t[1]; stats[2] = t[2]; stats[3] = t[3];
        stats[4] = -(t[4] - (double)vf * t[3] + (double)alpha * t[1]);       // ppo.py:1221-1225
    }
}

#define RS_GRAD_BLOCKS 256

// Adam (torch.optim.Adam semantics) + the KL early-stop decision, on the device.
struct RsParamSeg { float* p[12]; int off[13]; };

__global__ void __launch_bounds__(256) rs_adam_apply_kernel(RsParamSeg S, const float* __restrict__ grads, float* __restrict__ m,
                                                            float* __restrict__ v, const double* __restrict__ stats,
                                                            const rs_update_state* __restrict__ st, float lr, float thr) {
    if (st->stopped) return;
    if (!(stats[0] < (double)thr)) return;                       // kl >= 1.5 * target_kl: no step (ppo.py:1252-1261)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.off[12]) return;
    int seg = 0;
#pragma unroll
    for (int k = 1; k < 12; ++k) seg += (i >= S.off[k]) ? 1 : 0;
    float* p = S.p[seg] + (i - S.off[seg]);
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const int step = st->adam_step + 1;
    const float g = grads[i];
    const float mi = m[i] + (g - m[i]) * (1.0f - b1);            // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.0f - b2) * g * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    const float step_size = lr / bc1;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    *p = *p - step_size * (mi / denom);
}

__global__ void rs_adam_commit_kernel(const double* __restrict__ stats, rs_update_state* st, float thr) {
    if (st->stopped) return;
    st->iters += 1;
    for (int q = 0; q < 5; ++q) st->last_stats[q] = stats[q];
    if (stats[0] < (double)thr) st->adam_step += 1; else st->stopped = 1;
}

extern "C" {

int rs_rollout(rs_handle* h, const rs_mlp_params* actor, const rs_mlp_params* critic, const rs_rollout_args* args,
               rs_stream_t stream) {
    if (!h || !actor || !critic || !args) return RS_ERR_INVALID_ARG;
    const RsParams& P = h->P;
    if (P.A != 1 || (P.N % 16) != 0) return RS_ERR_UNSUPPORTED;
    const bool has_obs = P.obstruction_count != 0;
    if (has_obs && P.group != 1) return RS_ERR_UNSUPPORTED;
    if (args->steps_per_epoch < 1 || args->steps_per_episode < 1) return RS_ERR_INVALID_ARG;
    static int ver = -1;
    if (ver < 0) { const char* e = getenv("RS_ROLLOUT_V"); ver = (e && e[0] == '1') ? 1 : 2; }
    const bool v2 = ver == 2 || (P.N % RS_WAVE) != 0;          // v2: 16 envs per wave (256 waves at 4096 envs)
    size_t lds = sizeof(float) * (size_t)(v2 ? (rs_mlp16_lds_floats(8) + rs_mlp16_lds_floats(1)) : (rs_mlp_lds_floats(8) + rs_mlp_lds_floats(1))) + 16;
    lds += (has_obs ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0);
    lds += RS_WAVE * RS_OBS_DIM * 4 + RS_WAVE * 4 + 2 * RS_WAVE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (v2) {
        if (has_obs) hipLaunchKernelGGL(rs_rollout16_kernel<true>, dim3(P.N / 16), dim3(RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
        else hipLaunchKernelGGL(rs_rollout16_kernel<false>, dim3(P.N / 16), dim3(RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
    } else {
        if (has_obs) hipLaunchKernelGGL(rs_rollout_kernel<true>, dim3(P.N / RS_WAVE), dim3(RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
        else hipLaunchKernelGGL(rs_rollout_kernel<false>, dim3(P.N / RS_WAVE), dim3(RS_WAVE), lds, s, P, to_dev(actor), to_dev(critic), *args);
    }
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

static int rs_grad_version() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("RS_GRAD_V"); v = (e && e[0] == '1') ? 1 : ((e && e[0] == '3') ? 3 : ((e && e[0] == '4') ? 4 : 2)); }
    return v;
}

size_t rs_ppo_grad_workspace_bytes(void) {
    const size_t waves = RS_GRAD_BLOCKS * 8;
    return waves * (size_t)(rs_net_params(8) + rs_net_params(1)) * sizeof(float) + 2 * waves * 5 * sizeof(double) + 512
           + 2 * (2 * 2 * 4 * 64 * 8 * sizeof(uint16_t)) + 256;          // v4: third bf16 pieces of the layer-2 weights, both nets
}

int rs_adam_step(const rs_mlp_params* actor, const rs_mlp_params* critic, const float* grads, float* m, float* v,
                 const double* stats, rs_update_state* state, float lr, float kl_threshold, rs_stream_t stream) {
    if (!actor || !critic || !grads || !m || !v || !stats || !state) return RS_ERR_INVALID_ARG;
    RsParamSeg S;
    const float* ptrs[12] = {actor->w1, actor->b1, actor->w2, actor->b2, actor->w3, actor->b3,
                             critic->w1, critic->b1, critic->w2, critic->b2, critic->w3, critic->b3};
    const int sizes[12] = {704, 64, 4096, 64, 512, 8, 704, 64, 4096, 64, 64, 1};
    int o = 0;
    for (int k = 0; k < 12; ++k) { S.p[k] = const_cast<float*>(ptrs[k]); S.off[k] = o; o += sizes[k]; }
    S.off[12] = o;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(rs_adam_apply_kernel, dim3((o + 255) / 256), dim3(256), 0, s, S, grads, m, v, stats, state, lr, kl_threshold);
    hipLaunchKernelGGL(rs_adam_commit_kernel, dim3(1), dim3(1), 0, s, stats, state, kl_threshold);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_ppo_grad(const rs_mlp_params* actor, const rs_mlp_params* critic, const rs_ppo_batch* batch, float* grads,
                double* stats, void* workspace, const int32_t* stop_flag, rs_stream_t stream) {
    if (!actor || !critic || !batch || !grads || !stats || !workspace || batch->M < 1) return RS_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return RS_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ver = rs_grad_version();
    const bool v2 = ver >= 2;                                        // v3 / v4 = v2 with split-bf16 matrix instructions (opt-in)
    const int waves = v2 ? RS_GRAD_BLOCKS : RS_GRAD_BLOCKS * 4;      // v2 reduces its 8 waves in LDS: one slab per workgroup
    float* pa = static_cast<float*>(workspace);
    float* pc = pa + (size_t)waves * rs_net_params(8);
    double* sa = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(pc + (size_t)waves * rs_net_params(1)) + 255) & ~uintptr_t(255));
    double* sc = sa + (size_t)waves * 5;
    __bf16* xa = reinterpret_cast<__bf16*>((reinterpret_cast<uintptr_t>(sc + (size_t)waves * 5) + 255) & ~uintptr_t(255));
    __bf16* xc = xa + 2 * 2 * 4 * 64 * 8;
    static bool attr_set = false;
    const size_t lds_a = sizeof(float) * (size_t)(v2 ? rs_grad2_lds_floats(8) : rs_grad_lds_floats(8));
    const size_t lds_c = sizeof(float) * (size_t)(v2 ? rs_grad2_lds_floats(1) : rs_grad_lds_floats(1));
    if (!attr_set) {
        const void* ka = ver == 4 ? reinterpret_cast<const void*>(rs_ppo_grad3_kernel<8, 6>)
                       : ver == 3 ? reinterpret_cast<const void*>(rs_ppo_grad3_kernel<8, 3>)
                       : (v2 ? reinterpret_cast<const void*>(rs_ppo_grad2_kernel<8>) : reinterpret_cast<const void*>(rs_ppo_grad_kernel<8>));
        const void* kc = ver == 4 ? reinterpret_cast<const void*>(rs_ppo_grad3_kernel<1, 6>)
                       : ver == 3 ? reinterpret_cast<const void*>(rs_ppo_grad3_kernel<1, 3>)
                       : (v2 ? reinterpret_cast<const void*>(rs_ppo_grad2_kernel<1>) : reinterpret_cast<const void*>(rs_ppo_grad_kernel<1>));
        if (hipFuncSetAttribute(ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess ||
            hipFuncSetAttribute(kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess)
            return RS_ERR_HIP;
        attr_set = true;
    }
    if (ver == 4) {
        hipLaunchKernelGGL(rs_ppo_prep_kernel, dim3(8), dim3(512), 0, s, to_dev(actor), xa);
        hipLaunchKernelGGL(rs_ppo_prep_kernel, dim3(8), dim3(512), 0, s, to_dev(critic), xc);
        hipLaunchKernelGGL((rs_ppo_grad3_kernel<8, 6>), dim3(RS_GRAD_BLOCKS), dim3(512), lds_a, s, to_dev(actor), *batch, pa, sa, stop_flag, xa);
        hipLaunchKernelGGL((rs_ppo_grad3_kernel<1, 6>), dim3(RS_GRAD_BLOCKS), dim3(512), lds_c, s, to_dev(critic), *batch, pc, sc, stop_flag, xc);
    } else if (ver == 3) {
        hipLaunchKernelGGL((rs_ppo_grad3_kernel<8, 3>), dim3(RS_GRAD_BLOCKS), dim3(512), lds_a, s, to_dev(actor), *batch, pa, sa, stop_flag, xa);
        hipLaunchKernelGGL((rs_ppo_grad3_kernel<1, 3>), dim3(RS_GRAD_BLOCKS), dim3(512), lds_c, s, to_dev(critic), *batch, pc, sc, stop_flag, xc);
    } else if (v2) {
        hipLaunchKernelGGL(rs_ppo_grad2_kernel<8>, dim3(RS_GRAD_BLOCKS), dim3(512), lds_a, s, to_dev(actor), *batch, pa, sa, stop_flag);
        hipLaunchKernelGGL(rs_ppo_grad2_kernel<1>, dim3(RS_GRAD_BLOCKS), dim3(512), lds_c, s, to_dev(critic), *batch, pc, sc, stop_flag);
    } else {
        hipLaunchKernelGGL(rs_ppo_grad_kernel<8>, dim3(RS_GRAD_BLOCKS), dim3(256), lds_a, s, to_dev(actor), *batch, pa, sa, stop_flag);
        hipLaunchKernelGGL(rs_ppo_grad_kernel<1>, dim3(RS_GRAD_BLOCKS), dim3(256), lds_c, s, to_dev(critic), *batch, pc, sc, stop_flag);
    }
    const int np = rs_net_params(8) + rs_net_params(1);
    hipLaunchKernelGGL(rs_ppo_reduce_kernel, dim3((np + 63) / 64), dim3(1024), 0, s, pa, pc, sa, sc, waves, grads, stats,
                       batch->alpha, batch->vf_coef, stop_flag);
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
