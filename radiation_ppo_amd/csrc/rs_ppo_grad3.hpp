// rs_ppo_grad3.hpp -- K7 v3 / v4 (opt-in, RS_GRAD_V=3 / 4): v2 with its three 64x64 GEMMs per sample group -- forward
// layer 2, dh1 = W2^T dpre2, dW2 = dpre2 . h1^T -- moved from v_mfma_f32_32x32x2_f32 to split-bf16 matrix
// instructions (v_mfma_f32_32x32x16_bf16: 32 cycles per 16-deep k-step against 8 x 64 cycles).  Every f32 operand is
// written as a sum of bf16 pieces, x = hi + lo (+ lo2), each the bf16 rounding of what is left, and a product is
// accumulated in float32 from the significant cross terms:
//   TERMS = 3 (v3): hi*hi + hi*lo + lo*hi                      16-bit operands, ~4e-6 relative error
//   TERMS = 6 (v4): + hi*lo2 + lo2*hi + lo*lo                  24-bit operands, as accurate as the f32 instruction
// (scripts/bf16_split_study.py).  The third weight pieces do not fit the 160 KB of LDS next to the sample tiles: they
// are produced once per pass by rs_ppo_prep_kernel into 16 KB of the workspace and read through L1.
// Everything else (layer 1, output layer, dh2, dW3, dW1, LDS tiles, reduction, results layout) is v2 unchanged.
#pragma once
#include "rs_mlp.hpp"
#include "rs_ppo_grad2.hpp"

typedef __bf16 rs_bf16x8 __attribute__((ext_vector_type(8)));

// hidden unit supplied as k-element i of lane half kg in 16-deep step s (4 steps cover the 64 units):
// the 8 accumulator registers 8*(s&1) .. +7 of tile s>>1
__device__ __forceinline__ int rs_unit16(int s, int kg, int i) { return 32 * (s >> 1) + rs_kappa(8 * (s & 1) + i, kg); }
struct RsPieces { rs_bf16x8 p0, p1, p2; };       // hi, lo, lo2 (p2 unused when TERMS == 3)
template <int TERMS>
__device__ __forceinline__ RsPieces rs_split(const float (&x)[8]) {
    RsPieces r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)x[i];
        const float r1 = x[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        r.p0[i] = a; r.p1[i] = b;
        r.p2[i] = (TERMS == 6) ? (__bf16)(r1 - (float)b) : (__bf16)0.0f;
    }
    return r;
}
// smallest terms first into the accumulator would be marginally better; the order is fixed (reproducible)
template <int TERMS>
__device__ __forceinline__ void rs_mfma_split(f32x16& acc, const RsPieces& a, const RsPieces& b) {
    if (TERMS == 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b.p1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b.p2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, b.p0, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b.p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b.p0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p0, b.p0, acc, 0, 0, 0);
}

// third bf16 piece of the layer-2 weight fragments, [fwd 2x4x64x8 | bwd 2x4x64x8] (same order as the LDS fragments)
__global__ void rs_ppo_prep_kernel(RsMlpParams prm, __bf16* __restrict__ w2x) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * 4 * 64 * 8; e += gridDim.x * blockDim.x) {
        const int i = e & 7, l = (e >> 3) & 63, sx = (e >> 9) & 3, tile = e >> 11;
        const int u = rs_unit16(sx, l >> 5, i);
        const float vf = RS_TANH_PRESCALE * prm.w2[(32 * tile + (l & 31)) * RS_HID + u];
        const float vb = prm.w2[u * RS_HID + 32 * tile + (l & 31)];
        const __bf16 f0 = (__bf16)vf, b0 = (__bf16)vb;
        const float fr = vf - (float)f0, br = vb - (float)b0;
        const __bf16 f1 = (__bf16)fr, b1 = (__bf16)br;
        w2x[e] = (__bf16)(fr - (float)f1);
        w2x[2 * 4 * 64 * 8 + e] = (__bf16)(br - (float)b1);
    }
}


template <int NOUT, int TERMS>
__global__ void __launch_bounds__(512, 2) rs_ppo_grad3_kernel(RsMlpParams prm, rs_ppo_batch B, float* __restrict__ partial,
                                                              double* __restrict__ stat_partial, const int* __restrict__ stop,
                                                              const __bf16* __restrict__ w2x) {
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
    __syncthreads();                                   // W.fill wrote the f32 layer-2 fragments: replace them
    // [tile 2][step 4][lane 64][8] bf16, hi array then lo array, in the 4096-float regions v2 uses for w2f / w2tf
    __bf16* w2h = reinterpret_cast<__bf16*>(W.w2f);    // forward:  S * W2[32 ot + (l&31)][unit16(s, l>>5, i)]
    __bf16* w2l = w2h + 2 * 4 * 64 * 8;
    __bf16* w2th = reinterpret_cast<__bf16*>(w2tf);    // backward: W2[unit16(s, l>>5, i)][32 it + (l&31)]
    __bf16* w2tl = w2th + 2 * 4 * 64 * 8;
    for (int e = threadIdx.x; e < 2 * 4 * 64 * 8; e += blockDim.x) {
        const int i = e & 7, l = (e >> 3) & 63, sx = (e >> 9) & 3, tile = e >> 11;
        const int u = rs_unit16(sx, l >> 5, i);
        const float vf = RS_TANH_PRESCALE * prm.w2[(32 * tile + (l & 31)) * RS_HID + u];
        const float vb = prm.w2[u * RS_HID + 32 * tile + (l & 31)];
        const __bf16 fh = (__bf16)vf, bh = (__bf16)vb;
        w2h[e] = fh; w2l[e] = (__bf16)(vf - (float)fh);
        w2th[e] = bh; w2tl[e] = (__bf16)(vb - (float)bh);
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
            // layer 2 on split-bf16 matrix instructions: B = h1 (this lane's accumulator registers, split here),
            // A = the pre-split weight fragments (one 16-byte LDS read per piece)
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = H1[sx >> 1][8 * (sx & 1) + i];
                const RsPieces bp = rs_split<TERMS>(v8);
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    RsPieces ap;
                    ap.p0 = *reinterpret_cast<const rs_bf16x8*>(w2h + ((ot * 4 + sx) * 64 + lane) * 8);
                    ap.p1 = *reinterpret_cast<const rs_bf16x8*>(w2l + ((ot * 4 + sx) * 64 + lane) * 8);
                    if (TERMS == 6) ap.p2 = *reinterpret_cast<const rs_bf16x8*>(w2x + ((ot * 4 + sx) * 64 + lane) * 8);
                    rs_mfma_split<TERMS>(H2[ot], ap, bp);
                }
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
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = H2[sx >> 1][8 * (sx & 1) + i];
                const RsPieces bp = rs_split<TERMS>(v8);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    RsPieces ap;
                    ap.p0 = *reinterpret_cast<const rs_bf16x8*>(w2th + ((it * 4 + sx) * 64 + lane) * 8);
                    ap.p1 = *reinterpret_cast<const rs_bf16x8*>(w2tl + ((it * 4 + sx) * 64 + lane) * 8);
                    if (TERMS == 6) ap.p2 = *reinterpret_cast<const rs_bf16x8*>(w2x + 2 * 4 * 64 * 8 + ((it * 4 + sx) * 64 + lane) * 8);
                    rs_mfma_split<TERMS>(D1[it], ap, bp);
                }
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
            // dW2[it][kt] += dpre2[it] . h1^T: both operands are rows of the transposed f32 tiles (8 consecutive samples
            // per lane and 16-deep step), split on the way to the matrix instruction
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = Pt[c * RS_T2 + 16 * s2 + 8 * h + i];
                const RsPieces ap = rs_split<TERMS>(v8);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v8[i] = Qt[(32 * kt + c) * RS_T2 + 16 * s2 + 8 * h + i];
                    const RsPieces bp = rs_split<TERMS>(v8);
                    rs_mfma_split<TERMS>(acc2[it][kt], ap, bp);
                }
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
