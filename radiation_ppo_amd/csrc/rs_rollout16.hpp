// rs_rollout16.hpp -- K6: the fused collector with 16 envs per wave.
//
// A wave owns 16 envs: 4096 envs are 256 waves (every CU busy), the MLP runs on v_mfma_f32_16x16x4_f32 tiles (152
// MFMAs of 32 cycles per lock-step), and the env step -- a latency-bound scalar f64 chain whose cost does not depend on
// how many lanes run it -- is paid once per 16 envs.  (A 64-envs-per-wave layout was measured in round 1: 64 waves on 64
// of the 256 CUs and 304 MFMAs of 64 cycles per lock-step on one SIMD: slower.)
//
// Lane mapping: lane l = (j = l&15, g = l>>4).  Env slot j of the wave is env n = 16*block + j; its state and all
// per-env logic live in lane (j, 0).  For the MLP the four lanes (j, 0..3) share sample j:
//   A lane l: W[16it + (l&15)][k = 4s + (l>>4)]     B lane l: In[k = 4s + (l>>4)][sample l&15]
//   D lane l, reg q: sample l&15, unit 16it + 4*(l>>4) + q
// so the k-step (ut, q) of the next layer consumes unit 16ut + 4g + q straight from the accumulator registers
// (the 16x16 analogue of the kappa trick in rs_mlp.hpp).
//
// Two waves per workgroup (round 4).  A lone wave issues one instruction per ~3.4 ns whatever it is (profiles/r04_valu_cost.txt), and
// a lock-step was one serial stream of ~2 900 of them although only  observation -> actor -> draw -> env step -> observation  is a
// dependence chain: the critic's value is stored, never fed back.  Wave 0 walks the chain; wave 1 (another SIMD of the same CU, idle
// before) evaluates the critic on the observations wave 0 leaves in an LDS mailbox -- the pre-reset one for the bootstrap value, the
// post-reset one for the envs that were cut -- and writes val / last_val itself.  One workgroup barrier per lock-step, mailbox double
// buffered (wave 0 runs a step ahead).  With the critic skipped outright the lock-step took 7.6 instead of 9.6 us: that is the bound.
#pragma once
#include "rs_mlp.hpp"

__host__ __device__ constexpr int rs_mlp16_lds_floats(int nout) { return 4 * 3 * 64 + 4 * 16 * 64 + 64 + 4 * nout * 16 + nout; }

template <int NOUT>
struct RsMlp16 {
    float* w1f;   // [4 it][3 s][64]   2log2e * W1[16it + (l&15)][4s + (l>>4)], column 11 = 2log2e * b1 (x[11] := 1)
    float* w2f;   // [4 it][16 ks][64] 2log2e * W2[16it + (l&15)][16ut + 4(l>>4) + q], ks = 4ut + q
    float* b2;    // [64] 2log2e * b2
    float* w3g;   // [4 g][NOUT][16]   W3[o][16ut + 4g + q]
    float* b3;    // [NOUT]
    __device__ __forceinline__ void carve(float* base) {
        w1f = base; w2f = w1f + 4 * 3 * 64; b2 = w2f + 4 * 16 * 64; w3g = b2 + 64; b3 = w3g + 4 * NOUT * 16;
    }
    __device__ __forceinline__ void fill(const RsMlpParams& p) {
        const int tid = threadIdx.x, nt = blockDim.x;
        for (int i = tid; i < 4 * 3 * 64; i += nt) {
            const int l = i & 63, s = (i >> 6) % 3, it = i / (3 * 64);
            const int row = 16 * it + (l & 15), k = 4 * s + (l >> 4);
            w1f[i] = RS_TANH_PRESCALE * ((k < RS_IN) ? p.w1[row * RS_IN + k] : p.b1[row]);
        }
        for (int i = tid; i < 4 * 16 * 64; i += nt) {
            const int l = i & 63, ks = (i >> 6) & 15, it = i >> 10;
            const int unit = 16 * (ks >> 2) + 4 * (l >> 4) + (ks & 3);
            w2f[i] = RS_TANH_PRESCALE * p.w2[(16 * it + (l & 15)) * RS_HID + unit];
        }
        for (int i = tid; i < 64; i += nt) b2[i] = RS_TANH_PRESCALE * p.b2[i];
        for (int i = tid; i < 4 * NOUT * 16; i += nt) {
            const int q16 = i & 15, o = (i >> 4) % NOUT, g = i / (16 * NOUT);
            w3g[i] = p.w3[o * RS_HID + 16 * (q16 >> 2) + 4 * g + (q16 & 3)];
        }
        for (int i = tid; i < NOUT; i += nt) b3[i] = p.b3[i];
    }
    // forward for the wave's 16 samples; xs[k] = input k of sample (lane&15), already broadcast to the 4 lanes
    __device__ __forceinline__ void forward(const float (&xs)[RS_IN_PAD], float (&out)[NOUT]) const {
        const int lane = threadIdx.x & 63, g = lane >> 4;
        f32x4 H1[4], H2[4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) { H1[it][q] = 0.0f; H2[it][q] = b2[16 * it + 4 * g + q]; }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float b = (g == 0) ? xs[4 * s] : (g == 1) ? xs[4 * s + 1] : (g == 2) ? xs[4 * s + 2] : xs[4 * s + 3];
#pragma unroll
            for (int it = 0; it < 4; ++it)
                H1[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[(it * 3 + s) * 64 + lane], b, H1[it], 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) H1[it][q] = rs_tanh_scaled(H1[it][q]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float b = H1[ks >> 2][ks & 3];
#pragma unroll
            for (int it = 0; it < 4; ++it)
                H2[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[(it * 16 + ks) * 64 + lane], b, H2[it], 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) H2[it][q] = rs_tanh_scaled(H2[it][q]);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float4* w = reinterpret_cast<const float4*>(w3g + (g * NOUT + o) * 16);
            float p = 0.0f;
#pragma unroll
            for (int ut = 0; ut < 4; ++ut) {
                const float4 wv = w[ut];
                p = fmaf(wv.x, H2[ut][0], p); p = fmaf(wv.y, H2[ut][1], p);
                p = fmaf(wv.z, H2[ut][2], p); p = fmaf(wv.w, H2[ut][3], p);
            }
            // fixed-order tree over the four lanes of the sample: (g0 + g1) + (g2 + g3) in every lane
            const float q1 = __shfl_xor(p, 16);
            const float s01 = (g & 1) ? (q1 + p) : (p + q1);
            const float q2 = __shfl_xor(s01, 32);
            out[o] = ((g & 2) ? (q2 + s01) : (s01 + q2)) + b3[o];
        }
    }
};

// broadcast the inputs of sample j (held by lane (j, 0)) to its four lanes
__device__ __forceinline__ void rs_bcast_x16(const float (&xo)[RS_IN_PAD], float (&xs)[RS_IN_PAD]) {
    const int src = threadIdx.x & 15;
#pragma unroll
    for (int k = 0; k < RS_IN_PAD; ++k) xs[k] = __shfl(xo[k], src);
}

constexpr int RS_MB_PRE = 0, RS_MB_POST = 16 * RS_IN_PAD, RS_MB_FLAGS = 2 * 16 * RS_IN_PAD, RS_MB_SLOT = 2 * 16 * RS_IN_PAD + 16;   // words
__host__ __device__ constexpr int rs_rollout16_mailbox_bytes() { return 2 * RS_MB_SLOT * 4; }

template <bool HAS_OBS>
__global__ void __launch_bounds__(128) rs_rollout16_kernel(RsParams P, RsMlpParams pa, RsMlpParams pc, rs_rollout_args R) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* wts = reinterpret_cast<float*>(smem);
    unsigned char* p = smem + sizeof(float) * (size_t)(rs_mlp16_lds_floats(8) + rs_mlp16_lds_floats(1));
    p = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~uintptr_t(15));
    int* lds_geo = reinterpret_cast<int*>(p);
    uint32_t* lds_adj = reinterpret_cast<uint32_t*>(p + RS_MAX_VERT * RS_WAVE * 4);
    double* lds_d = reinterpret_cast<double*>(p + 2 * RS_MAX_VERT * RS_WAVE * 4);
    float* tile = reinterpret_cast<float*>(p + (HAS_OBS ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0));
    float* lds_rew = tile + RS_WAVE * RS_OBS_DIM;
    uint8_t* lds_done = reinterpret_cast<uint8_t*>(lds_rew + RS_WAVE);
    uint8_t* lds_oob = lds_done + RS_WAVE;
    float* mbox = reinterpret_cast<float*>(lds_oob + RS_WAVE);          // [2 slots][pre 16 x 12 | post 16 x 12 | flags 16]

    RsMlp16<8> ACT; RsMlp16<1> CRT;
    ACT.carve(wts);
    CRT.carve(wts + rs_mlp16_lds_floats(8));
    ACT.fill(pa); CRT.fill(pc);

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, j = lane & 15;
    const bool own = lane < 16;                                   // lane (j, 0) of either wave stands for env slot j
    const int n = blockIdx.x * 16 + j;                            // N % 16 == 0 (checked by the host)
    const int N = P.N, T = R.steps_per_epoch, L = R.steps_per_episode;
    // obstacles: the four lanes (j, 0..3) of an env run its step together (rs_env_step_lane<true, 4>): lane (j, g) takes
    // corner g of every rectangle in the shortest-path loop, every 4th rectangle, two of the eight probe directions
    const int cj = lane >> 4;
    RsGeo g{lds_geo, 0, 0, 0};
    if (HAS_OBS && wave == 0) {
        rs_load_geo(P, n, own, lds_geo, g);
        g.off = __shfl(g.off, j); g.stride = __shfl(g.stride, j); g.n = __shfl(g.n, j);
    }
    __syncthreads();

    if (wave == 1) {
        // ---- the critic's wave: message t (t = -1: the epoch's first observation) is written by wave 0 before barrier t
        for (int t = -1; t < T; ++t) {
            __syncthreads();
            const float* M = mbox + (t & 1) * RS_MB_SLOT;
            float xs[RS_IN_PAD];
#pragma unroll
            for (int k = 0; k < RS_IN_PAD; ++k) xs[k] = M[RS_MB_PRE + j * RS_IN_PAD + k];
            float vb;
            { float vv[1]; CRT.forward(xs, vv); vb = vv[0]; }
            const int fl = reinterpret_cast<const int*>(M + RS_MB_FLAGS)[j];          // bit 0: cut, bit 1: boot (both 0 in message -1)
            const bool cut = (fl & 1) != 0, boot = (fl & 2) != 0;
            float v = vb;
            if (__ballot(cut) != 0ull) {                          // the envs that were cut start again from the observation after the reset
#pragma unroll
                for (int k = 0; k < RS_IN_PAD; ++k) xs[k] = M[RS_MB_POST + j * RS_IN_PAD + k];
                float vv[1];
                CRT.forward(xs, vv);
                v = cut ? vv[0] : vb;
            }
            if (own) {
                if (t >= 0) R.last_val[(size_t)t * N + n] = (cut && boot) ? vb : 0.0f;
                if (t + 1 < T) R.val[(size_t)(t + 1) * N + n] = v;
            }
        }
        return;
    }

    float oraw[RS_OBS_DIM];
#pragma unroll
    for (int k = 0; k < RS_OBS_DIM; ++k) oraw[k] = R.cur_obs[(size_t)n * RS_OBS_DIM + k];
    RsWelford W{R.w_count[n], R.w_mean[n], R.w_sq[n], R.w_std[n]};
    int steps = R.steps_in_ep[n];
    float ep_ret = R.ep_ret[n];
    int done_count = 0, oob_count = 0, ep_count = 0;
    double ep_ret_sum = 0.0, ep_len_sum = 0.0, ep_ret_sq = 0.0;
    float ep_ret_max = -INFINITY, ep_ret_min = INFINITY;
    const uint32_t k0 = P.seed, k1 = P.env_id_base + (uint32_t)n;
    // per-episode constants and the step counter the sampler's Philox counter needs are mirrored in registers (re-read after a
    // reset only): a global load per lock-step is a full memory latency for the lone wave of a SIMD
    uint32_t ep_key = P.episode[n] - 1u, t_key = P.tstep[n];
    float srcx = (float)P.src_x[n], srcy = (float)P.src_y[n];

    float xo[RS_IN_PAD], xs[RS_IN_PAD];
#pragma unroll
    for (int k = 0; k < RS_OBS_DIM; ++k) xo[k] = oraw[k];
    xo[0] = W.standardize(oraw[0]);
    xo[11] = 1.0f;                                                // constant input carrying the layer-1 bias
    rs_bcast_x16(xo, xs);
    {   // message -1: the epoch's first observation (no flags)
        float* M = mbox + RS_MB_SLOT;
        if (own) {
#pragma unroll
            for (int k = 0; k < RS_IN_PAD; ++k) M[RS_MB_PRE + j * RS_IN_PAD + k] = xo[k];
            reinterpret_cast<int*>(M + RS_MB_FLAGS)[j] = 0;
        }
    }
    __syncthreads();

    RsOut O;
    O.obs_row = tile + j * RS_OBS_DIM;
    O.reward = lds_rew + j - (size_t)n;
    O.team = nullptr;
    O.done = lds_done + j - (size_t)n;
    O.oob = lds_oob + j - (size_t)n;
    O.oobc = nullptr; O.blocked = nullptr; O.collision = nullptr;

    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * N + n;
        float* M = mbox + (t & 1) * RS_MB_SLOT;
        float lg[8];
        ACT.forward(xs, lg);
        int a = 0;
        float logp = 0.0f;
        if (own) {
            float mx = lg[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) mx = fmaxf(mx, lg[q]);
            float se = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) se += __expf(lg[q] - mx);
            const float lse = __logf(se);
            const uint32_t episode = ep_key, tenv = t_key;
            u32x4 ph = philox4x32_10(0u, tenv, episode, RS_STREAM_ACT, k0, k1);
            const float u = (float)(ph.x >> 8) * (1.0f / 16777216.0f);
            float cdf = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float lpq = (lg[q] - mx) - lse;
                cdf += __expf(lpq);
                if (q < 7) a += (cdf <= u) ? 1 : 0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) logp = (a == q) ? ((lg[q] - mx) - lse) : logp;
#pragma unroll
            for (int k = 0; k < RS_OBS_DIM; ++k) tile[lane * RS_OBS_DIM + k] = xo[k];
        }
        __builtin_amdgcn_wave_barrier();                         // (one wave: its LDS operations are ordered; this only pins the compiler)
        {
            float* dst = R.obs + ((size_t)t * N + (size_t)blockIdx.x * 16) * RS_OBS_DIM;
            for (int i = lane; i < 16 * RS_OBS_DIM; i += RS_WAVE) dst[i] = tile[i];
        }
        __builtin_amdgcn_wave_barrier();
        bool cut = false;
        bool over = false, boot = false, ended = t == T - 1;
        if (own) {
            R.act[row] = (int64_t)a;
            R.logp[row] = logp;
            R.source_tar[row * 2 + 0] = srcx;
            R.source_tar[row * 2 + 1] = srcy;
        }
        if (HAS_OBS) {
            const int a_env = __shfl(a, j);
            rs_env_step_lane<true, 4>(P, g, n, [&](int) -> int { return a_env; }, O, false, cj);
        } else if (own) {
            rs_env_step_lane<false>(P, g, n, [&](int) -> int { return a; }, O);
        }
        if (own) {
            const float r = lds_rew[lane];
            const bool terminal = lds_done[lane] != 0;
            oob_count += lds_oob[lane];
            R.rew[row] = r;
            ep_ret += r;
            steps += 1;
            t_key += 1u;                                         // the env advanced its step counter (rs_env_step_lane)
            done_count += terminal ? 1 : 0;
            const bool timeout = steps == L;
            over = terminal || timeout;
            cut = over || ended;
            boot = timeout || ended;
#pragma unroll
            for (int k = 0; k < RS_OBS_DIM; ++k) oraw[k] = tile[lane * RS_OBS_DIM + k];
            W.update((double)oraw[0]);
#pragma unroll
            for (int k = 1; k < RS_OBS_DIM; ++k) xo[k] = oraw[k];
            xo[0] = W.standardize(oraw[0]);
            // the critic's message: the observation the bootstrap value is taken on, and whether it is
#pragma unroll
            for (int k = 0; k < RS_IN_PAD; ++k) M[RS_MB_PRE + j * RS_IN_PAD + k] = xo[k];
            reinterpret_cast<int*>(M + RS_MB_FLAGS)[j] = (cut ? 1 : 0) | (boot ? 2 : 0);
            R.cut[row] = cut ? 1 : 0;
            if (over) {
                ep_ret_sum += (double)ep_ret; ep_len_sum += (double)steps; ep_count += 1;
                ep_ret_sq += (double)ep_ret * (double)ep_ret;
                ep_ret_max = fmaxf(ep_ret_max, ep_ret); ep_ret_min = fminf(ep_ret_min, ep_ret);
            }
        }
        if (HAS_OBS) {
            if (__shfl((int)cut, j) != 0) {                      // the env's four lanes reset it together
                if (ended) P.epoch_end[n] = 1;
                rs_env_reset_lane<true, 4>(P, g, n, lds_geo, lds_adj, lds_d, tile + j * RS_OBS_DIM, O, j, cj);
            }
        } else if (cut) {
            if (ended) P.epoch_end[n] = 1;
            rs_env_reset_lane<false>(P, g, n, lds_geo, lds_adj, lds_d, tile + lane * RS_OBS_DIM, O);
        }
        if (own) {
            if (cut) {
                W.reset();
#pragma unroll
                for (int k = 0; k < RS_OBS_DIM; ++k) oraw[k] = tile[lane * RS_OBS_DIM + k];
                W.update((double)oraw[0]);
#pragma unroll
                for (int k = 1; k < RS_OBS_DIM; ++k) xo[k] = oraw[k];
                xo[0] = W.standardize(oraw[0]);
                steps = 0;
                ep_ret = 0.0f;
                ep_key = P.episode[n] - 1u; t_key = P.tstep[n];  // new episode: new source, new Philox counters
                srcx = (float)P.src_x[n]; srcy = (float)P.src_y[n];
#pragma unroll
                for (int k = 0; k < RS_IN_PAD; ++k) M[RS_MB_POST + j * RS_IN_PAD + k] = xo[k];
            }
        }
        rs_bcast_x16(xo, xs);
        __syncthreads();                                          // message t is complete; wave 1 takes it from here
    }
    if (own) {
#pragma unroll
        for (int k = 0; k < RS_OBS_DIM; ++k) R.cur_obs[(size_t)n * RS_OBS_DIM + k] = oraw[k];
        R.w_count[n] = W.count; R.w_mean[n] = W.mean; R.w_sq[n] = W.sq; R.w_std[n] = W.std;
        R.steps_in_ep[n] = steps;
        R.ep_ret[n] = ep_ret;
        R.done_count[n] = done_count; R.oob_count[n] = oob_count; R.ep_count[n] = ep_count;
        R.ep_ret_sum[n] = ep_ret_sum; R.ep_len_sum[n] = ep_len_sum;
        R.ep_ret_sq_sum[n] = ep_ret_sq; R.ep_ret_max[n] = ep_ret_max; R.ep_ret_min[n] = ep_ret_min;
    }
}
