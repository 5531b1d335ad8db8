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

// Diagnostic build only (-DRS_K7_STAMPS, scripts/k7_stamps.py): s_memtime stamps at the phase boundaries of the sample-group
// loop, summed per wave in scalar registers and added to a table no other code reads.  The product build contains none of it.
#ifdef RS_K7_STAMPS
#define RS_K7_NPH 16
__device__ unsigned long long rs_k7_stamp_table[2][RS_K7_NPH];
#define RS_STAMP_DECL unsigned long long st_acc[RS_K7_NPH] = {0}; const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(); \
                      unsigned long long st_last = rs_k7_now();
#define RS_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; RS_STAMP_MARK " #i); const unsigned long long t_ = rs_k7_now(); st_acc[i] += t_ - st_last; st_last = t_; \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ unsigned long long rs_k7_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#else
#define RS_STAMP_DECL
// product build: a phase boundary is a scheduling fence only (A/B switch RS_K7_NO_PHASE_FENCE: hipcc is then free to move LDS
// reads, staging writes and VALU work across the phases)
#ifdef RS_K7_NO_PHASE_FENCE
#define RS_STAMP(i) do {} while (0)
#else
#define RS_STAMP(i) __builtin_amdgcn_sched_barrier(0)
#endif
#endif
#define RS_XRAW 400                                // per-wave landing zone of the group's 32 x 11 sample rows (352 floats, LDS-DMA);
                                                   // between layer 1 and the next DMA it holds the [12][33] dz^T / statistics tile
#define RS_XSC 192                                 // per-wave landing zone of the per-sample scalars: act (32 x int64), adv, logp_old, w, ret
#define RS_G2_WAVE_FLOATS ((64 + 32) * RS_T2 + RS_XRAW + RS_XSC + 64)

__host__ __device__ constexpr int rs_grad2_lds_floats(int nout) {
    return ((rs_mlp_lds_floats(nout) + 3) & ~3) + 2 * 2 * 16 * 64 + 2 * 4 * 64 + 8 * RS_G2_WAVE_FLOATS;
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
    float* w2tf = smem_f + ((rs_mlp_lds_floats(NOUT) + 3) & ~3);   // 16-byte aligned; [2 it][2 kt][4 r4][64 lanes][4]: W2[32kt + kappa(r, l>>5)][32it + (l&31)]
    float* w3tf = w2tf + 2 * 2 * 16 * 64;               // [2 it][4 s][64]:        W3[2s + (l>>5)][32it + (l&31)]
    // wave id as a SCALAR: the per-wave LDS bases below then live in SGPRs (and M0 for the LDS-DMA) instead of VGPRs
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const int l15 = lane & 15, l4 = lane >> 4;
    float* Qt = w3tf + 2 * 4 * 64 + wid * RS_G2_WAVE_FLOATS;   // [64][33]  h^T tile (h2 for dW3, then h1 for dW2)
    float* Pt = Qt + 64 * RS_T2;                               // [32][33]  dpre^T half tile
    float* xraw = Pt + 32 * RS_T2;                             // [32][11]  the group's sample rows as they lie in HBM (LDS-DMA target)
    float* xsc = xraw + RS_XRAW;                               // [192]     the group's per-sample scalars (LDS-DMA target)
    float* dbl = xsc + RS_XSC;                                 // [64]      db2 accumulators of this wave
    W.fill(prm);
    // W2 fragments for the matrix pipe, four consecutive k-steps per lane contiguous ([it][kt][r / 4][lane][r % 4]): one
    // ds_read_b128 feeds four MFMAs (an LDS read instruction costs the wave ~13 issue cycles whatever its width).  The forward
    // copy (W.w2f, filled by W.fill in the layout rs_policy_forward uses) is rewritten in the same form below.
    for (int i = threadIdx.x; i < 2 * 2 * 16 * 64; i += blockDim.x) {
        const int ri = i & 3, l = (i >> 2) & 63, r4 = (i >> 8) & 3, kt = (i >> 10) & 1, it = i >> 11;
        const int r = 4 * r4 + ri;
        w2tf[i] = prm.w2[(32 * kt + rs_kappa(r, l >> 5)) * RS_HID + 32 * it + (l & 31)];
    }
    for (int i = threadIdx.x; i < 2 * 4 * 64; i += blockDim.x) {
        int l = i & 63, sq = (i >> 6) & 3, it = i >> 8;
        int o = 2 * sq + (l >> 5);
        w3tf[i] = (o < NOUT) ? prm.w3[o * RS_HID + 32 * it + (l & 31)] : 0.0f;
    }
    dbl[lane] = 0.0f;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 2 * 16 * 64; i += blockDim.x) {
        const int ri = i & 3, l = (i >> 2) & 63, r4 = (i >> 8) & 3, kt = (i >> 10) & 1, it = i >> 11;
        const int r = 4 * r4 + ri;
        W.w2f[i] = RS_TANH_PRESCALE * prm.w2[(32 * it + (l & 31)) * RS_HID + 32 * kt + rs_kappa(r, l >> 5)];
    }
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
    // db3 and the loss statistics are per-sample scalars summed over samples: they ride in the dz^T tile (rows 0..NOUT-1: dz,
    // rows 8..11: kl / entropy / clip fraction / surrogate terms, or the value-loss term) and are summed by eight extra
    // 16x16x4 MFMAs per group against a B operand of ones -- no per-lane accumulators, no cross-lane reduction
    f32x4 accs;
#pragma unroll
    for (int r = 0; r < 4; ++r) accs[r] = 0.f;

    // ---- sample data of a group: the 32 x 11 rows lie contiguously in HBM (352 floats) and are copied to the wave's LDS
    // landing zone by LDS-DMA (six 256-byte wave-instructions, no registers), the per-sample scalars (action, advantage,
    // old log-prob, weight, return) by three more.  Both are fetched ONE GROUP AHEAD (issued once the landing zones are free,
    // after the dW3 phase), so no HBM latency is exposed and no register is held across the group.
    // Every wave runs the same number of trips; a trip past the last group works on clamped rows with weight 0.
    const int trips = (groups + n_waves - 1) / n_waves;
    const long x_last = (long)M * RS_IN - 1;
    auto dma_rows = [&](int g) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            long idx = (long)g * (32 * RS_IN) + i * 64 + lane;
            idx = idx < x_last ? idx : x_last;                  // tail of the last group / lanes past the 352 floats: stay in bounds
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B.x + idx),
                                             (__attribute__((address_space(3))) void*)(xraw + i * 64), 4, 0, 0);
        }
    };
    auto dma_scal = [&](int g) {
        // three wave-instructions: [act: 64 dwords = 32 int64] [adv | logp_old] [w | ret]; sample index clamped like the rows
        const int m0 = g * 32;
        const int ms = min(m0 + c, M - 1);
        {
            const int mi = min(m0 + (lane >> 1), M - 1);
            const int* src = reinterpret_cast<const int*>(B.act) + 2 * (size_t)mi + (lane & 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xsc), 4, 0, 0);
        }
        {
            const float* src = (h ? B.logp_old : B.adv) + ms;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xsc + 64), 4, 0, 0);
        }
        {
            const float* src = (h ? B.ret : B.w) + ms;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xsc + 128), 4, 0, 0);
        }
    };
    {
        const int g0 = wave_g < groups ? wave_g : groups - 1;
        dma_rows(g0);
        dma_scal(g0);
    }
    RS_STAMP_DECL
    for (int trip = 0; trip < trips; ++trip) {
        RS_STAMP(15);                                   // loop overhead / tail of the previous group
        const int gi_raw = wave_g + trip * n_waves;
        const int gi = gi_raw < groups ? gi_raw : groups - 1;
        const int m = gi * 32 + c;
        const bool valid = gi_raw < groups && m < M;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the group's rows and scalars have landed in xraw / xsc
#if defined(RS_K7_ALIGN) && RS_K7_ALIGN >= 1
        __builtin_amdgcn_s_barrier();                               // A/B: start every group in step (VALU phases of the two waves of a SIMD co-issue)
#endif
        const float wi = valid ? xsc[128 + c] : 0.0f;
        const int s_act = reinterpret_cast<const int*>(xsc)[2 * c];
        const float s_adv = xsc[64 + c], s_lpo = xsc[96 + c], s_ret = xsc[160 + c];
        // layer-1 B operands: lane (c, h) feeds input 2s + h of sample c at k-step s; input 11 is the constant 1 (bias column)
        float xv[6];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) xv[s6] = xraw[c * RS_IN + 2 * s6 + h];
        if (h) xv[5] = 1.0f;
        RS_STAMP(0);                                    // wait for the rows, operand reads
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
                const float b = xv[s];
                float n0 = 0.f, n1 = 0.f;
                if (s + 1 < 6) { n0 = w1b[(0 * 6 + s + 1) * 64 + lane]; n1 = w1b[(1 * 6 + s + 1) * 64 + lane]; }
                H1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, H1[0], 0, 0, 0);
                H1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, H1[1], 0, 0, 0);
                a0 = n0; a1 = n1;
            }
        }
        RS_STAMP(1);                                    // layer 1 MFMAs
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) H1[it][r] = rs_tanh_scaled(H1[it][r]);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) H2[it][r] = W.b2[32 * it + rs_kappa(r, h)];
        RS_STAMP(2);                                    // tanh 1 + bias loads
        {
            // both output tiles advance together: two independent accumulators per fragment pair
            const float4* wq = reinterpret_cast<const float4*>(W.w2f) + lane;       // [(it * 2 + kt) * 4 + r4][64 lanes]
            float4 a0 = wq[0], a1 = wq[(1 * 2 + 0) * 4 * 64];
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {
                const int kt = g4 >> 2, r0 = 4 * (g4 & 3);
                float4 n0 = a0, n1 = a1;
                if (g4 + 1 < 8) {
                    n0 = wq[((0 * 2 + ((g4 + 1) >> 2)) * 4 + ((g4 + 1) & 3)) * 64];
                    n1 = wq[((1 * 2 + ((g4 + 1) >> 2)) * 4 + ((g4 + 1) & 3)) * 64];
                }
                H2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, H1[kt][r0 + 0], H2[0], 0, 0, 0);
                H2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, H1[kt][r0 + 0], H2[1], 0, 0, 0);
                H2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, H1[kt][r0 + 1], H2[0], 0, 0, 0);
                H2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, H1[kt][r0 + 1], H2[1], 0, 0, 0);
                H2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, H1[kt][r0 + 2], H2[0], 0, 0, 0);
                H2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, H1[kt][r0 + 2], H2[1], 0, 0, 0);
                H2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, H1[kt][r0 + 3], H2[0], 0, 0, 0);
                H2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, H1[kt][r0 + 3], H2[1], 0, 0, 0);
                a0 = n0; a1 = n1;
            }
        }
        // h1 has fed its last MFMA: its transpose goes to Qt now (B operand of dW2 later, and the source of the tanh'
        // factor of dpre1), so the 32 registers are free for the rest of the group
        rs_stage32(Qt, H1[0], c, h);
        rs_stage32(Qt + 32 * RS_T2, H1[1], c, h);
        RS_STAMP(3);                                    // layer 2 MFMAs + h1^T staging
#if defined(RS_K7_ALIGN) && RS_K7_ALIGN >= 2
        __builtin_amdgcn_s_barrier();
#endif
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) H2[it][r] = rs_tanh_scaled(H2[it][r]);
        RS_STAMP(4);                                    // tanh 2
        // output layer on the VALU: out[o] = sum over the 32 units this lane holds of W3[o][unit] * h2[unit], halves added across
        // lane pairs.  Weights are broadcast float4 reads; they are fetched one batch (8 outputs x 4 units) ahead of the FMAs
        // that use them so that the LDS latency is paid once, not per output (summation order per output unchanged: ascending unit)
        float out[NOUT];
        {
            float pacc[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) pacc[o] = 0.0f;
            const float4* wrow = reinterpret_cast<const float4*>(W.w3h + h * NOUT * 32);      // [o][8 float4], 16-byte aligned rows
            float4 wb[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) wb[o] = wrow[o * 8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int kt = b >> 2, r4 = b & 3;
                float4 wn[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) wn[o] = (b + 1 < 8) ? wrow[o * 8 + b + 1] : wb[o];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    pacc[o] = fmaf(wb[o].x, H2[kt][4 * r4 + 0], pacc[o]);
                    pacc[o] = fmaf(wb[o].y, H2[kt][4 * r4 + 1], pacc[o]);
                    pacc[o] = fmaf(wb[o].z, H2[kt][4 * r4 + 2], pacc[o]);
                    pacc[o] = fmaf(wb[o].w, H2[kt][4 * r4 + 3], pacc[o]);
                }
                if (b + 1 < 8) {
                    __builtin_amdgcn_sched_group_barrier(0x100, NOUT, 0);        // next batch's weight reads first ...
                    __builtin_amdgcn_sched_group_barrier(0x002, 4 * NOUT, 0);    // ... then this batch's FMAs
                }
#pragma unroll
                for (int o = 0; o < NOUT; ++o) wb[o] = wn[o];
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                // fixed summation order in both lanes: (half 0) + (half 1)
                const float q = __shfl_xor(pacc[o], 32);
                out[o] = (h ? (q + pacc[o]) : (pacc[o] + q)) + W.b3[o];
            }
        }
        RS_STAMP(5);                                    // output layer (VALU)
        // ---------------- per-sample loss derivative (identical in both lanes of a sample) ----------------
        float dz[NOUT], sq[4];
        if (NOUT == 8) {
            const int a = s_act;
            const float adv = s_adv, lpo = s_lpo;
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
            sq[0] = wi * (lpo - logp);
            sq[1] = wi * ent;
            sq[2] = wi * ((ratio > hi || ratio < lo) ? 1.0f : 0.0f);
            sq[3] = wi * surr;
        } else {
            const float diff = out[0] - s_ret;
            dz[0] = 2.0f * B.vf_coef * wi * diff;
            sq[0] = wi * diff * diff; sq[1] = 0.f; sq[2] = 0.f; sq[3] = 0.f;
        }
        RS_STAMP(6);                                    // loss derivative

        // ---------------- backward ----------------
        // dz^T borrows the row landing zone (its rows were consumed by layer 1; the next group's DMA is issued after dW3):
        // rows 0..NOUT-1 = dz, rows 8..11 = the statistics terms (rows NOUT..7 of the critic's tile are never read)
        float* Dz = xraw;                                // [12][33]
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) Dz[o * RS_T2 + c] = dz[o];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) Dz[(8 + q) * RS_T2 + c] = sq[q];
        }
        // R1/R2: dW3[o][unit] += sum_n dz[o][n] h2[unit][n]  (16x16x4 tiles, 8 k-steps), h2^T through Pt one 32-unit half at a time
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            rs_wave_sync();                              // the previous readers of Pt are done
            rs_stage32(Pt, H2[hf], c, h);
            rs_wave_sync();
            // A rows: o < NOUT -> dz, 8..11 -> statistics terms (they meet only the ones operand), everything else 0
            const bool a_row = l15 < NOUT || (l15 >= 8 && l15 < 12);
            float a_c = a_row ? Dz[l15 * RS_T2 + l4] : 0.0f;
            float b_c[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) b_c[u] = Pt[(16 * u + l15) * RS_T2 + l4];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float a_n = 0.f, b_n[2] = {0.f, 0.f};
                if (s + 1 < 8) {
                    a_n = a_row ? Dz[l15 * RS_T2 + 4 * (s + 1) + l4] : 0.0f;
#pragma unroll
                    for (int u = 0; u < 2; ++u) b_n[u] = Pt[(16 * u + l15) * RS_T2 + 4 * (s + 1) + l4];
                }
                const float a_dz = (l15 < NOUT) ? a_c : 0.0f;      // the statistics rows must not leak into dW3
#pragma unroll
                for (int u = 0; u < 2; ++u) acc3[2 * hf + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_dz, b_c[u], acc3[2 * hf + u], 0, 0, 0);
                if (hf == 0) accs = __builtin_amdgcn_mfma_f32_16x16x4f32(a_c, 1.0f, accs, 0, 0, 0);   // row sums: db3 and the statistics
                a_c = a_n;
#pragma unroll
                for (int u = 0; u < 2; ++u) b_c[u] = b_n[u];
            }
        }
        RS_STAMP(7);                                    // dW3
        // dh2 = W3^T dz, dpre2 = dh2 * (1 - h2^2) in place of H2
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
        // the landing zones are free again: fetch the NEXT group's rows and scalars behind the rest of this group
        {
            const int gn_raw = wave_g + (trip + 1) * n_waves;
            const int gn = gn_raw < groups ? gn_raw : groups - 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // every ds_read of Dz / xsc has returned
            dma_rows(gn);
            dma_scal(gn);
        }
        RS_STAMP(8);                                    // dh2 -> dpre2, DMA issue
#if defined(RS_K7_ALIGN) && RS_K7_ALIGN >= 3
        __builtin_amdgcn_s_barrier();
#endif
        // R3: dh1 = W2^T dpre2 (register operands)  ||  dpre2[0]^T -> Pt
        rs_wave_sync();
        rs_stage32(Pt, H2[0], c, h);
        f32x16 D1[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) D1[it][r] = 0.f;
        {
            const float4* wq = reinterpret_cast<const float4*>(w2tf) + lane;
            float4 a0 = wq[0], a1 = wq[(1 * 2 + 0) * 4 * 64];
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {
                const int kt = g4 >> 2, r0 = 4 * (g4 & 3);
                float4 n0 = a0, n1 = a1;
                if (g4 + 1 < 8) {
                    n0 = wq[((0 * 2 + ((g4 + 1) >> 2)) * 4 + ((g4 + 1) & 3)) * 64];
                    n1 = wq[((1 * 2 + ((g4 + 1) >> 2)) * 4 + ((g4 + 1) & 3)) * 64];
                }
                D1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, H2[kt][r0 + 0], D1[0], 0, 0, 0);
                D1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, H2[kt][r0 + 0], D1[1], 0, 0, 0);
                D1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, H2[kt][r0 + 1], D1[0], 0, 0, 0);
                D1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, H2[kt][r0 + 1], D1[1], 0, 0, 0);
                D1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, H2[kt][r0 + 2], D1[0], 0, 0, 0);
                D1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, H2[kt][r0 + 2], D1[1], 0, 0, 0);
                D1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, H2[kt][r0 + 3], D1[0], 0, 0, 0);
                D1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, H2[kt][r0 + 3], D1[1], 0, 0, 0);
                a0 = n0; a1 = n1;
            }
        }
        // R6's B operand: x[sample 4s + l4][input l15] (input 11 := 1 -> column 11 of dW1 is db1), re-read from L2 (the DMA
        // touched the same lines one group ago); issued here, consumed after dW2
        float xb[8];
        {
            int gi_late = gi;
            asm volatile("" : "+v"(gi_late));           // opaque: keeps the eight loads from being hoisted to the top of the group
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                int n = gi_late * 32 + 4 * s8 + l4;
                n = n < M ? n : M - 1;
                xb[s8] = (l15 < RS_IN) ? B.x[(size_t)n * RS_IN + l15] : ((l15 == RS_IN) ? 1.0f : 0.0f);
            }
        }
        // dpre1 = dh1 * (1 - h1^2): h1 comes back from its transposed tile in accumulator layout (conflict-free column reads)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hv = Qt[(32 * it + rs_kappa(r, h)) * RS_T2 + c];
                const float d = D1[it][r];
                D1[it][r] = fmaf(-(d * hv), hv, d);
            }
        rs_wave_sync();
        RS_STAMP(9);                                    // R3 dh1 MFMAs + dpre2^T staging + dpre1
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
        RS_STAMP(10);                                   // R4 / R5 dW2 + db2
        // R6: dW1[unit][input] += sum_n dpre1[unit][n] x[input][n]  (16x16x4, 8 k-steps per half)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            rs_wave_sync();
            rs_stage32(Pt, D1[it], c, h);
            rs_wave_sync();
            float a0_c = Pt[l15 * RS_T2 + l4], a1_c = Pt[(16 + l15) * RS_T2 + l4];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float a0_n = 0.f, a1_n = 0.f;
                if (s + 1 < 8) {
                    a0_n = Pt[l15 * RS_T2 + 4 * (s + 1) + l4];
                    a1_n = Pt[(16 + l15) * RS_T2 + 4 * (s + 1) + l4];
                }
                acc1[2 * it + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0_c, xb[s], acc1[2 * it + 0], 0, 0, 0);
                acc1[2 * it + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1_c, xb[s], acc1[2 * it + 1], 0, 0, 0);
                a0_c = a0_n; a1_c = a1_n;
            }
        }
        rs_wave_sync();
        RS_STAMP(11);                                   // R6 dW1
    }
#ifdef RS_K7_STAMPS
    if (lane == 0) {
        st_acc[14] = __builtin_amdgcn_s_memrealtime() - st_rt0;     // 100 MHz ticks over the same span: in-kernel clock = cycles / ticks * 100 MHz
        for (int q = 0; q < RS_K7_NPH; ++q) atomicAdd(&rs_k7_stamp_table[NOUT == 8 ? 0 : 1][q], st_acc[q]);
    }
#endif

    // ---- one partial slab per WORKGROUP, parameter order {w1, b1, w2, b2, w3, b3}: the eight waves add their
    // accumulators into one LDS slab in wave order (fixed order -> reproducible), then the block streams it out.
    __syncthreads();                                  // every wave is done with its staging tiles
    float* red = w3tf + 2 * 4 * 64;                   // reuse the staging region: rs_net_params(NOUT) floats
    double* sred = reinterpret_cast<double*>(red + ((rs_net_params(NOUT) + 1) & ~1));
    float* g_w1 = red, *g_b1 = g_w1 + 64 * 11, *g_w2 = g_b1 + 64, *g_b2 = g_w2 + 64 * 64, *g_w3 = g_b2 + 64, *g_b3 = g_w3 + NOUT * 64;
    const float db2v = dbl[lane];                     // (dbl lives in the region being overwritten: read it first)
    // accs (16x16 D layout: row 4*(lane>>4) + q, every column identical): rows 0..NOUT-1 = db3, rows 8..11 = statistics sums
    // of this wave; bring them to lane 0 (row r lives in lanes with lane>>4 == r/4, register r%4)
    float db3r[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) db3r[o] = __shfl(accs[o & 3], 16 * (o >> 2));
    double sv[5];
    {
        const float t0 = __shfl(accs[0], 32), t1 = __shfl(accs[1], 32), t2 = __shfl(accs[2], 32), t3 = __shfl(accs[3], 32);
        if (NOUT == 8) { sv[0] = (double)t0; sv[1] = (double)t1; sv[2] = (double)t2; sv[3] = 0.0; sv[4] = (double)t3; }
        else { sv[0] = 0.0; sv[1] = 0.0; sv[2] = 0.0; sv[3] = (double)t0; sv[4] = 0.0; }
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
