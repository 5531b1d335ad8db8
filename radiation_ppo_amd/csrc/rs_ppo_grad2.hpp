// rs_ppo_grad2.hpp -- K7 v2: the fused PPO loss+gradient pass with 32 samples per wave and TWO waves per SIMD.
//
// v1 (rs_ppo_grad_kernel) gives every wave 64 samples and ~500 registers, i.e. one wave per SIMD: the VALU/LDS
// phases of a group (tanh, output layer, loss, tile staging) cannot hide behind the MFMA phases because an
// in-order wave has no second wave to interleave with -- rocprof shows the matrix pipe 55-60 % busy.  v2 halves
// the transient state (one 32-sample tile per wave: H1, H2, dH1 are 32 registers each), keeps the whole kernel
// under 256 registers, and launches 512-thread workgroups = 8 waves per CU = 2 per SIMD, so the hardware
// overlaps one wave's VALU/LDS work with the other's MFMAs.
//
// Lane mapping (v_mfma_f32_32x32x2_f32): lane l = (c = l&31, h = l>>5); both lanes (c, 0) and (c, 1) belong to
// sample c of the group and hold the hidden units 32*it + kappa(r, h) of that sample, so no partner exchange is
// needed for the inputs or for dz; per-sample scalars are computed redundantly in both lanes.
#pragma once
#include "rs_mlp.hpp"

#define RS_T2 33                                   // row stride of the 32-sample LDS tiles (floats)
#define RS_G2_WAVE_FLOATS ((64 + 32 + 12) * RS_T2 + 64)

__host__ __device__ constexpr int rs_grad2_lds_floats(int nout) {
    return rs_mlp_lds_floats(nout) + 2 * 2 * 16 * 64 + 2 * 4 * 64 + 8 * RS_G2_WAVE_FLOATS;
}

__device__ __forceinline__ void rs_stage32(float* T, const f32x16& v, int c, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[rs_kappa(r, h) * RS_T2 + c] = v[r];
}

template <int NOUT>
__global__ void __launch_bounds__(512, 2) rs_ppo_grad2_kernel(RsMlpParams prm, rs_ppo_batch B, float* __restrict__ partial,
                                                              double* __restrict__ stat_partial, const int* __restrict__ stop) {
    extern __shared__ __align__(16) float smem_f[];
    if (stop && *stop) return;
    RsMlpLds<NOUT> W;
    W.carve(smem_f);
    float* w2tf = smem_f + rs_mlp_lds_floats(NOUT);     // [2 it][2 kt][16 r][64]: W2[32kt + kappa][32it + (l&31)]
    float* w3tf = w2tf + 2 * 2 * 16 * 64;               // [2 it][4 s][64]:        W3[2s + (l>>5)][32it + (l&31)]
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int l15 = lane & 15, l4 = lane >> 4;
    float* Qt = w3tf + 2 * 4 * 64 + wid * RS_G2_WAVE_FLOATS;   // [64][33]  h^T tile (h2 for dW3, then h1 for dW2)
    float* Pt = Qt + 64 * RS_T2;                               // [32][33]  dpre^T half tile
    float* St = Pt + 32 * RS_T2;                               // [12][33]  dz^T (rows < NOUT) / x^T (12 rows, row 11 = 1)
    float* dbl = St + 12 * RS_T2;                              // [64]      db2 accumulators of this wave
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
    dbl[lane] = 0.0f;
    __syncthreads();
    // layer-1 fragments with the bias in the padded input column k = 11 (k-step 5, upper lane half)
    float* w1b = W.w1f;
    for (int i = threadIdx.x; i < 2 * 32; i += blockDim.x) {
        const int it = i >> 5, row = i & 31;
        w1b[(it * 6 + 5) * 64 + 32 + row] = W.b1[32 * it + row];
    }
    __syncthreads();

    const int M = B.M;
    const int groups = (M + 31) / 32;
    const int wave_g = blockIdx.x * 8 + wid, n_waves = gridDim.x * 8;

    f32x16 acc2[2][2];
    f32x4 acc1[4], acc3[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[a][0][r] = 0.f; acc2[a][1][r] = 0.f; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc1[a][r] = 0.f; acc3[a][r] = 0.f; }
    float db3[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) db3[o] = 0.f;
    double st0 = 0.0, st1 = 0.0, st2 = 0.0, st3 = 0.0;      // actor: kl, ent, clipfrac, surr; critic: st0 = value loss

    for (int gi = wave_g; gi < groups; gi += n_waves) {
        const int m = gi * 32 + c;
        const bool valid = m < M;
        const int mm = valid ? m : M - 1;
        float x[RS_IN_PAD];
#pragma unroll
        for (int k = 0; k < RS_IN; ++k) x[k] = B.x[(size_t)mm * RS_IN + k];
        x[11] = 1.0f;                     // constant input: column 11 of w1b carries b1, and column 11 of dW1 is db1
        const float wi = valid ? B.w[mm] : 0.0f;
        // x^T goes to its LDS tile right away (R6 reads it at the end of the group), so the 12 input registers die
        // after layer 1 instead of living through the whole backward pass (they were what spilled)
        if (h == 0) {
#pragma unroll
            for (int k = 0; k < RS_IN_PAD; ++k) St[k * RS_T2 + c] = x[k];
        }

        // ---------------- forward ----------------
        f32x16 H1[2], H2[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) H1[it][r] = 0.0f;
        {
            float a0 = w1b[(0 * 6 + 0) * 64 + lane], a1 = w1b[(1 * 6 + 0) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float b = h ? x[2 * s + 1] : x[2 * s];
                float n0 = 0.f, n1 = 0.f;
                if (s + 1 < 6) { n0 = w1b[(0 * 6 + s + 1) * 64 + lane]; n1 = w1b[(1 * 6 + s + 1) * 64 + lane]; }
                H1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, H1[0], 0, 0, 0);
                H1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, H1[1], 0, 0, 0);
                a0 = n0; a1 = n1;
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) H1[it][r] = rs_tanh_scaled(H1[it][r]);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) H2[it][r] = W.b2[32 * it + rs_kappa(r, h)];
        {
            // both output tiles advance together: two independent accumulators per fragment pair
            float a0 = W.w2f[((0 * 2 + 0) * 16 + 0) * 64 + lane], a1 = W.w2f[((1 * 2 + 0) * 16 + 0) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int kt = q >> 4, r = q & 15;
                float n0 = 0.f, n1 = 0.f;
                if (q + 1 < 32) {
                    n0 = W.w2f[((0 * 2 + ((q + 1) >> 4)) * 16 + ((q + 1) & 15)) * 64 + lane];
                    n1 = W.w2f[((1 * 2 + ((q + 1) >> 4)) * 16 + ((q + 1) & 15)) * 64 + lane];
                }
                H2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, H1[kt][r], H2[0], 0, 0, 0);
                H2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, H1[kt][r], H2[1], 0, 0, 0);
                a0 = n0; a1 = n1;
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) H2[it][r] = rs_tanh_scaled(H2[it][r]);
        float out[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float p = 0.0f;
            const float4* w = reinterpret_cast<const float4*>(W.w3h + (h * NOUT + o) * 32);   // 16-byte aligned rows
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 wv = w[kt * 4 + r4];
                    p = fmaf(wv.x, H2[kt][4 * r4 + 0], p);
                    p = fmaf(wv.y, H2[kt][4 * r4 + 1], p);
                    p = fmaf(wv.z, H2[kt][4 * r4 + 2], p);
                    p = fmaf(wv.w, H2[kt][4 * r4 + 3], p);
                }
            // fixed summation order in both lanes: (half 0) + (half 1)
            const float q = __shfl_xor(p, 32);
            out[o] = (h ? (q + p) : (p + q)) + W.b3[o];
        }

        // ---------------- per-sample loss derivative (identical in both lanes of a sample) ----------------
        float dz[NOUT];
        if (NOUT == 8) {
            const int a = (int)B.act[mm];
            const float adv = B.adv[mm], lpo = B.logp_old[mm];
            float mx = out[0];
#pragma unroll
            for (int j = 1; j < NOUT; ++j) mx = fmaxf(mx, out[j]);
            float se = 0.f;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) se += __expf(out[j] - mx);
            const float lse = __logf(se);
            float lp[NOUT], pj[NOUT], ent = 0.f, logp = 0.f;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                lp[j] = (out[j] - mx) - lse;
                pj[j] = __expf(lp[j]);
                ent -= pj[j] * lp[j];
                logp = (a == j) ? lp[j] : logp;
            }
            const float ratio = __expf(logp - lpo);
            const float lo = 1.0f - B.clip_ratio, hi = 1.0f + B.clip_ratio;
            const float clipped = fminf(fmaxf(ratio, lo), hi);
            const float s1 = ratio * adv, s2 = clipped * adv;
            const float surr = fminf(s1, s2);
            const bool inside = ratio >= lo && ratio <= hi;
            const float dr = (inside || s1 < s2) ? adv : 0.0f;
            const float g_lp = -wi * dr * ratio;
            // the entropy bonus is a detached scalar in the reference (`ent = pi.entropy().detach().mean().item()`,
            // ppo.py:1216): alpha * H moves the loss VALUE only, no gradient flows through it
#pragma unroll
            for (int j = 0; j < NOUT; ++j) dz[j] = g_lp * (((a == j) ? 1.0f : 0.0f) - pj[j]);
            if (h == 0) {
                st0 += (double)(wi * (lpo - logp));
                st1 += (double)(wi * ent);
                st2 += (double)(wi * ((ratio > hi || ratio < lo) ? 1.0f : 0.0f));
                st3 += (double)(wi * surr);
            }
        } else {
            const float diff = out[0] - B.ret[mm];
            dz[0] = 2.0f * B.vf_coef * wi * diff;
            if (h == 0) st0 += (double)(wi * diff * diff);
        }
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) db3[o] += dz[o];
        }

        // ---------------- backward ----------------
        // R1: h2^T -> Qt, dz^T -> St
        rs_stage32(Qt, H2[0], c, h);
        rs_stage32(Qt + 32 * RS_T2, H2[1], c, h);
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) Pt[o * RS_T2 + c] = dz[o];      // dz^T borrows Pt (free until R3)
        }
        rs_wave_sync();
        // R2: dW3[o][unit] += sum_n dz[o][n] h2[unit][n]  (16x16x4, 8 k-steps)  ||  dh2 -> dpre2 (in place of H2)
        {
            float a_c = (l15 < NOUT) ? Pt[l15 * RS_T2 + l4] : 0.0f;
            float b_c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b_c[u] = Qt[(16 * u + l15) * RS_T2 + l4];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float a_n = 0.f, b_n[4] = {0.f, 0.f, 0.f, 0.f};
                if (s + 1 < 8) {
                    a_n = (l15 < NOUT) ? Pt[l15 * RS_T2 + 4 * (s + 1) + l4] : 0.0f;
#pragma unroll
                    for (int u = 0; u < 4; ++u) b_n[u] = Qt[(16 * u + l15) * RS_T2 + 4 * (s + 1) + l4];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc3[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c, b_c[u], acc3[u], 0, 0, 0);
                a_c = a_n;
#pragma unroll
                for (int u = 0; u < 4; ++u) b_c[u] = b_n[u];
            }
        }
        if (NOUT == 8) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float b = h ? dz[(2 * s + 1) % NOUT] : dz[(2 * s) % NOUT];
                    t = __builtin_amdgcn_mfma_f32_32x32x2f32(w3tf[(it * 4 + s) * 64 + lane], b, t, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float hv = H2[it][r]; H2[it][r] = fmaf(-(t[r] * hv), hv, t[r]); }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hv = H2[kt][r];
                    const float d = W.w3h[h * 32 + kt * 16 + r] * dz[0];
                    H2[kt][r] = fmaf(-(d * hv), hv, d);
                }
        }
        rs_wave_sync();
        // R3: dh1 = W2^T dpre2 (register operands)  ||  h1^T -> Qt, dpre2[0]^T -> Pt, x^T -> St
        f32x16 D1[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) D1[it][r] = 0.f;
        {
            float a0 = w2tf[((0 * 2 + 0) * 16 + 0) * 64 + lane], a1 = w2tf[((1 * 2 + 0) * 16 + 0) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int kt = q >> 4, r = q & 15;
                float n0 = 0.f, n1 = 0.f;
                if (q + 1 < 32) {
                    n0 = w2tf[((0 * 2 + ((q + 1) >> 4)) * 16 + ((q + 1) & 15)) * 64 + lane];
                    n1 = w2tf[((1 * 2 + ((q + 1) >> 4)) * 16 + ((q + 1) & 15)) * 64 + lane];
                }
                D1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, H2[kt][r], D1[0], 0, 0, 0);
                D1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, H2[kt][r], D1[1], 0, 0, 0);
                a0 = n0; a1 = n1;
            }
        }
        rs_stage32(Qt, H1[0], c, h);
        rs_stage32(Qt + 32 * RS_T2, H1[1], c, h);
        rs_stage32(Pt, H2[0], c, h);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float hv = H1[it][r]; const float d = D1[it][r]; D1[it][r] = fmaf(-(d * hv), hv, d); }
        rs_wave_sync();
        // R4 / R5: dW2[it][kt] += dpre2[it] . h1^T (16 k-steps over the 32 samples); db2 row sums from the staged tile
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (it == 1) {
                rs_wave_sync();
                rs_stage32(Pt, H2[1], c, h);
                rs_wave_sync();
            }
            float a_c = Pt[c * RS_T2 + h], b0_c = Qt[c * RS_T2 + h], b1_c = Qt[(32 + c) * RS_T2 + h];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float a_n = 0.f, b0_n = 0.f, b1_n = 0.f;
                if (s + 1 < 16) {
                    a_n = Pt[c * RS_T2 + 2 * (s + 1) + h];
                    b0_n = Qt[c * RS_T2 + 2 * (s + 1) + h];
                    b1_n = Qt[(32 + c) * RS_T2 + 2 * (s + 1) + h];
                }
                acc2[it][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b0_c, acc2[it][0], 0, 0, 0);
                acc2[it][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b1_c, acc2[it][1], 0, 0, 0);
                a_c = a_n; b0_c = b0_n; b1_c = b1_n;
            }
            // db2[32it + c] += sum over the 32 samples of dpre2: lane (c, h) sums samples 16h .. 16h+15 of row c
            float rs = 0.0f;
#pragma unroll
            for (int n = 0; n < 16; ++n) rs += Pt[c * RS_T2 + 16 * h + n];
            rs += __shfl_xor(rs, 32);
            if (h == 0) dbl[32 * it + c] += rs;
        }
        // R6: dW1[unit][input] += sum_n dpre1[unit][n] x[input][n]  (16x16x4, 8 k-steps per half)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            rs_wave_sync();
            rs_stage32(Pt, D1[it], c, h);
            rs_wave_sync();
            float b_c = (l15 < RS_IN_PAD) ? St[l15 * RS_T2 + l4] : 0.0f;
            float a0_c = Pt[l15 * RS_T2 + l4], a1_c = Pt[(16 + l15) * RS_T2 + l4];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float b_n = 0.f, a0_n = 0.f, a1_n = 0.f;
                if (s + 1 < 8) {
                    b_n = (l15 < RS_IN_PAD) ? St[l15 * RS_T2 + 4 * (s + 1) + l4] : 0.0f;
                    a0_n = Pt[l15 * RS_T2 + 4 * (s + 1) + l4];
                    a1_n = Pt[(16 + l15) * RS_T2 + 4 * (s + 1) + l4];
                }
                acc1[2 * it + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0_c, b_c, acc1[2 * it + 0], 0, 0, 0);
                acc1[2 * it + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1_c, b_c, acc1[2 * it + 1], 0, 0, 0);
                b_c = b_n; a0_c = a0_n; a1_c = a1_n;
            }
        }
        rs_wave_sync();
    }

    // ---- one partial slab per WORKGROUP, parameter order {w1, b1, w2, b2, w3, b3}: the eight waves add their
    // accumulators into one LDS slab in wave order (fixed order -> reproducible), then the block streams it out.
    __syncthreads();                                  // every wave is done with its staging tiles
    float* red = w3tf + 2 * 4 * 64;                   // reuse the staging region: rs_net_params(NOUT) floats
    double* sred = reinterpret_cast<double*>(red + ((rs_net_params(NOUT) + 1) & ~1));
    float* g_w1 = red, *g_b1 = g_w1 + 64 * 11, *g_w2 = g_b1 + 64, *g_b2 = g_w2 + 64 * 64, *g_w3 = g_b2 + 64, *g_b3 = g_w3 + NOUT * 64;
    const float db2v = dbl[lane];                     // (dbl lives in the region being overwritten: read it first)
    float db3r[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float v = db3[o];
        v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        db3r[o] = v;
    }
    double sv[5];
    if (NOUT == 8) { sv[0] = st0; sv[1] = st1; sv[2] = st2; sv[3] = 0.0; sv[4] = st3; }
    else { sv[0] = 0.0; sv[1] = 0.0; sv[2] = 0.0; sv[3] = st0; sv[4] = 0.0; }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double v = sv[q];
        v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        sv[q] = v;
    }
    __syncthreads();                                  // all dbl reads done before the slab is written
    for (int wv = 0; wv < 8; ++wv) {
        if (wid == wv) {
            const bool first = wv == 0;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * it + rs_kappa(r, h);
                    g_w2[row * 64 + c] = (first ? 0.0f : g_w2[row * 64 + c]) + acc2[it][0][r];
                    g_w2[row * 64 + 32 + c] = (first ? 0.0f : g_w2[row * 64 + 32 + c]) + acc2[it][1][r];
                }
            g_b2[lane] = (first ? 0.0f : g_b2[lane]) + db2v;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * u + 4 * l4 + q;
                    if (l15 < RS_IN) g_w1[row * RS_IN + l15] = (first ? 0.0f : g_w1[row * RS_IN + l15]) + acc1[u][q];
                    if (l15 == RS_IN) g_b1[row] = (first ? 0.0f : g_b1[row]) + acc1[u][q];
                    const int o = 4 * l4 + q;
                    if (o < NOUT) g_w3[o * 64 + 16 * u + l15] = (first ? 0.0f : g_w3[o * 64 + 16 * u + l15]) + acc3[u][q];
                }
            if (lane == 0) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) g_b3[o] = (first ? 0.0f : g_b3[o]) + db3r[o];
#pragma unroll
                for (int q = 0; q < 5; ++q) sred[q] = (first ? 0.0 : sred[q]) + sv[q];
            }
        }
        __syncthreads();
    }
    float* outp = partial + (size_t)blockIdx.x * rs_net_params(NOUT);
    for (int i = threadIdx.x; i < rs_net_params(NOUT); i += blockDim.x) outp[i] = red[i];
    if (threadIdx.x < 5) stat_partial[(size_t)blockIdx.x * 5 + threadIdx.x] = sred[threadIdx.x];
}
