// rs_mlp.hpp -- the 2x64 tanh MLP of FF_core.ActorCritic (NeuralNetworkCores/FF_core.py:42-129) on the
// gfx950 matrix cores, for ONE wave that owns 64 samples (lane = sample).
//
// Formulation  Out[unit i][sample j] = sum_k W[i][k] * In[k][j]  with v_mfma_f32_32x32x2_f32 (exact f32,
// k-ordered fmaf chains): weights are the A operand, activations the B operand, samples run along the lanes.
//   A lane l: A[i = l&31][k = l>>5]      B lane l: B[k = l>>5][j = l&31]
//   D lane l, reg r: column j = l&31, row i = (r&3) + 8*(r>>2) + 4*(l>>5)           (MI355X guide, sec. 3)
// Because the summation order over k is free as long as A and B agree, the accumulator registers of one
// layer ARE the B operands of the next (k-step (kt, r) consumes hidden unit 32*kt + kappa(r, h)); no lane
// movement and no LDS round trip between layers.  Only the weights are laid out to match ("fragment
// order", built once per launch in LDS).  The 8-/1-wide output layers are too thin for a 32x32 tile and run
// on the VALU: each lane reduces the 32 hidden units it holds per sample, lane pairs (l, l^32) exchange
// their halves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RS_HID 64
#define RS_IN 11
#define RS_IN_PAD 12
#define RS_TANH_PRESCALE 2.885390081777927f   // 2 log2(e): see rs_tanh_scaled

// torch layout, row-major [out][in]
struct RsMlpParams { const float *w1, *b1, *w2, *b2, *w3, *b3; };

// floats of LDS one net needs (NOUT = 8 actor, 1 critic)
__host__ __device__ constexpr int rs_mlp_lds_floats(int nout) { return 2 * 6 * 64 + 64 + 2 * 2 * 16 * 64 + 64 + 2 * nout * 32 + nout; }

__device__ __forceinline__ int rs_kappa(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int NOUT>
struct RsMlpLds {
    // w1f, b1, w2f, b2 are stored multiplied by 2 log2(e) (see rs_tanh_scaled)
    float* w1f;   // [2 it][6 s][64 lanes]      W1[32it + (l&31)][2s + (l>>5)]   (k = 11 -> 0)
    float* b1;    // [64]
    float* w2f;   // [2 it][2 kt][16 r][64]     W2[32it + (l&31)][32kt + kappa(r, l>>5)]
    float* b2;    // [64]
    float* w3h;   // [2 h][NOUT][32 q]          W3[o][32*(q>>4) + kappa(q&15, h)]
    float* b3;    // [NOUT]

    __device__ __forceinline__ void carve(float* base) {
        w1f = base; b1 = w1f + 2 * 6 * 64; w2f = b1 + 64; b2 = w2f + 2 * 2 * 16 * 64; w3h = b2 + 64; b3 = w3h + 2 * NOUT * 32;
    }
    // cooperative fill by all threads of the block (call, then __syncthreads())
    __device__ __forceinline__ void fill(const RsMlpParams& p) {
        const int tid = threadIdx.x, nt = blockDim.x;
        for (int i = tid; i < 2 * 6 * 64; i += nt) {
            int l = i & 63, s = (i >> 6) % 6, it = i / (6 * 64);
            int row = 32 * it + (l & 31), k = 2 * s + (l >> 5);
            w1f[i] = (k < RS_IN) ? RS_TANH_PRESCALE * p.w1[row * RS_IN + k] : 0.0f;
        }
        for (int i = tid; i < 2 * 2 * 16 * 64; i += nt) {
            int l = i & 63, r = (i >> 6) & 15, kt = (i >> 10) & 1, it = i >> 11;
            w2f[i] = RS_TANH_PRESCALE * p.w2[(32 * it + (l & 31)) * RS_HID + 32 * kt + rs_kappa(r, l >> 5)];
        }
        for (int i = tid; i < 2 * NOUT * 32; i += nt) {
            int q = i & 31, o = (i >> 5) % NOUT, h = i / (32 * NOUT);
            w3h[i] = p.w3[o * RS_HID + 32 * (q >> 4) + rs_kappa(q & 15, h)];
        }
        for (int i = tid; i < 64; i += nt) { b1[i] = RS_TANH_PRESCALE * p.b1[i]; b2[i] = RS_TANH_PRESCALE * p.b2[i]; }
        for (int i = tid; i < NOUT; i += nt) b3[i] = p.b3[i];
    }
};

// tanh from a PRE-SCALED pre-activation y = 2 log2(e) x  (the LDS copies of W1, b1, W2, b2 carry the factor):
//   tanh(x) = 1 - 2 / (1 + 2^y)   ->  v_exp_f32, v_add_f32, v_rcp_f32, v_fma_f32: 4 VALU instructions.
// The f32-input MFMA runs on the same FP32 lanes as the VALU (both 64 FLOP/clk/SIMD), so VALU instructions
// do not hide behind MFMAs -- every instruction saved here is wall time.  abs error < 3e-7.
__device__ __forceinline__ float rs_tanh_scaled(float y) {
    const float e = __builtin_amdgcn_exp2f(y);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// Hidden activations of one net for the wave's 64 samples, kept in accumulator layout:
// h[kt][jt][r] = unit 32kt + kappa(r, lane>>5) of sample 32jt + (lane&31).
struct RsHidden { f32x16 v[2][2]; };

// layer 1 from per-lane inputs.  xo = this lane's sample (12 floats, x[11] = 0), xp = the sample of lane^32.
template <int NOUT>
__device__ __forceinline__ void rs_mlp_layer1(const RsMlpLds<NOUT>& W, const float (&xo)[RS_IN_PAD], const float (&xp)[RS_IN_PAD],
                                              RsHidden& H) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) H.v[it][jt][r] = W.b1[32 * it + rs_kappa(r, h)];
    // weight fragments are fetched one k-step ahead of the MFMAs that consume them (LDS latency hidden)
    float a0 = W.w1f[(0 * 6 + 0) * 64 + lane], a1 = W.w1f[(1 * 6 + 0) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const float own = h ? xo[2 * s + 1] : xo[2 * s];
        const float par = h ? xp[2 * s + 1] : xp[2 * s];
        const float b0 = h ? par : own;      // jt = 0: lanes < 32 own the sample
        const float b1 = h ? own : par;      // jt = 1: lanes >= 32 own the sample
        float n0 = 0.f, n1 = 0.f;
        if (s + 1 < 6) { n0 = W.w1f[(0 * 6 + s + 1) * 64 + lane]; n1 = W.w1f[(1 * 6 + s + 1) * 64 + lane]; }
        H.v[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, H.v[0][0], 0, 0, 0);
        H.v[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, H.v[0][1], 0, 0, 0);
        H.v[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, H.v[1][0], 0, 0, 0);
        H.v[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, H.v[1][1], 0, 0, 0);
        a0 = n0; a1 = n1;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) H.v[it][jt][r] = rs_tanh_scaled(H.v[it][jt][r]);
}

// layer 2: accumulator layout in -> accumulator layout out (pre-activation in Z when KEEP_PRE, tanh in H2)
template <int NOUT>
__device__ __forceinline__ void rs_mlp_layer2(const RsMlpLds<NOUT>& W, const RsHidden& H1, RsHidden& H2) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) H2.v[it][jt][r] = W.b2[32 * it + rs_kappa(r, h)];
    // out-tile major: the tanh of tile it = 0 (VALU) overlaps the MFMA chain of tile it = 1; weight fragments
    // are fetched two k-steps ahead of their MFMAs
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float a_c = W.w2f[((it * 2 + 0) * 16 + 0) * 64 + lane];
        float a_n = W.w2f[((it * 2 + 0) * 16 + 1) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int kt = q >> 4, r = q & 15;
            float a_nn = 0.f;
            if (q + 2 < 32) a_nn = W.w2f[((it * 2 + ((q + 2) >> 4)) * 16 + ((q + 2) & 15)) * 64 + lane];
            H2.v[it][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, H1.v[kt][0][r], H2.v[it][0], 0, 0, 0);
            H2.v[it][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, H1.v[kt][1][r], H2.v[it][1], 0, 0, 0);
            a_c = a_n; a_n = a_nn;
            // pin the issue order: the fragment read for step q+2, then this step's two MFMAs (hipcc otherwise
            // sinks the read next to its use and waits lgkmcnt(0) in front of every MFMA pair)
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) H2.v[it][jt][r] = rs_tanh_scaled(H2.v[it][jt][r]);
    }
}

// output layer on the VALU: out[o] for the lane's OWN sample
template <int NOUT>
__device__ __forceinline__ void rs_mlp_out(const RsMlpLds<NOUT>& W, const RsHidden& H2, float (&out)[NOUT]) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float p0 = 0.0f, p1 = 0.0f;           // partial sums for sample tile jt = 0 / 1
        const float* w = W.w3h + (h * NOUT + o) * 32;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float wv = w[kt * 16 + r];
                p0 = fmaf(wv, H2.v[kt][0][r], p0);
                p1 = fmaf(wv, H2.v[kt][1][r], p1);
            }
        const float mine = h ? p1 : p0;        // the tile this lane owns is jt = h
        const float give = h ? p0 : p1;
        const float recv = __shfl_xor(give, 32);
        out[o] = (mine + recv) + W.b3[o];
    }
}

template <int NOUT>
__device__ __forceinline__ void rs_mlp_forward(const RsMlpLds<NOUT>& W, const float (&xo)[RS_IN_PAD], const float (&xp)[RS_IN_PAD],
                                               float (&out)[NOUT]) {
    RsHidden H1, H2;
    rs_mlp_layer1<NOUT>(W, xo, xp, H1);
    rs_mlp_layer2<NOUT>(W, H1, H2);
    rs_mlp_out<NOUT>(W, H2, out);
}
