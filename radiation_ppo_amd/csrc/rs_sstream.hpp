// rs_sstream.hpp -- per-lane matrix-vector products whose (wave-uniform) weights stream through the scalar unit: K12 - K15.
//
// out[OUTP] (+)= W^T c for a k-major [K][OUTP] block: one lane = one sample / particle, the weights arrive as s_load_dwordx16 rows
// in the constant address space and feed v_fmac's SGPR operand, so they cost neither VGPRs nor LDS bandwidth.
//
// Rows travel in blocks of two (32 SGPRs), double buffered.  Scalar loads return out of order, so the only wait there is,
// s_waitcnt lgkmcnt(0), drains EVERYTHING in flight; the order inside a block therefore has to be
//     wait for this block's rows  ->  request the next block  ->  32 FMAs,
// which gives the next rows the whole FMA run (~80 cycles) to arrive -- one x16 load costs a wave ~44 cycles
// (scripts/micro/sload_latency.hip).  Left to itself hipcc requests the next rows first and waits right behind them: at one wave per
// SIMD (K13) a full scalar round trip was exposed for every second or third row (27.9 -> 24.9 ms per launch when this was fixed).  The
// empty asm that "reads" the current rows pins the wait in front of the requests; scheduling barriers keep the three parts in that
// order and stop the scheduler from hoisting dozens of loads (SGPR spills through v_writelane / v_readlane); the accumulators are
// pinned at every block, otherwise whole FMA chains are sunk to their use and re-read a row per FMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef const float __attribute__((address_space(4))) * rs_cmem_t;
__device__ __forceinline__ rs_cmem_t rs_as_cmem(const float* p) { return (rs_cmem_t)(uintptr_t)p; }

#define RS_SS_ARRIVED(w) asm volatile("" :: "s"((w)[0]), "s"((w)[16]))

template <int K, int OUTP, typename F>
__device__ __forceinline__ void rs_ss_mv(rs_cmem_t W, F cval, float (&out)[OUTP]) {
    static_assert(OUTP % 16 == 0, "16 outputs per chunk");
    constexpr int NB = (K + 1) / 2;                                  // an odd K ends with a one-row block (its second row re-reads row K - 1)
#pragma unroll
    for (int ch = 0; ch < OUTP / 16; ++ch) {
        float acc[16], wq[2][32];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = out[16 * ch + o];
#pragma unroll
        for (int i = 0; i < 32; ++i) wq[0][i] = W[((i >> 4) < K ? (i >> 4) : K - 1) * OUTP + 16 * ch + (i & 15)];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float (&cur)[32] = wq[b & 1];
            RS_SS_ARRIVED(cur);
            __builtin_amdgcn_sched_barrier(0);
            if (b + 1 < NB) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int row = 2 * (b + 1) + (i >> 4);
                    wq[(b + 1) & 1][i] = W[(row < K ? row : K - 1) * OUTP + 16 * ch + (i & 15)];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const float c0 = cval(2 * b);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fmaf(cur[o], c0, acc[o]);
            if (2 * b + 1 < K) {
                const float c1 = cval(2 * b + 1);
#pragma unroll
                for (int o = 0; o < 16; ++o) acc[o] = fmaf(cur[16 + o], c1, acc[o]);
            }
#pragma unroll
            for (int o = 0; o < 16; ++o) asm volatile("" : "+v"(acc[o]));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) out[16 * ch + o] = acc[o];
    }
}

// out[k] += sum_o W[k][o] c(o) on the SAME k-major [K][OUTP] block (the transposed product of a backward pass: a dot product along
// each row).  Two 16-weight pieces per block, the same wait -> request -> FMA order, two partial sums per row.
template <int K, int OUTP, typename F>
__device__ __forceinline__ void rs_ss_mvt(rs_cmem_t W, F cval, float (&out)[K]) {
    static_assert(OUTP % 16 == 0, "16 weights per piece");
    constexpr int CH = OUTP / 16, NP = K * CH, NB = (NP + 1) / 2;
    float wq[2][32];
#pragma unroll
    for (int i = 0; i < 32; ++i) wq[0][i] = W[((i >> 4) < NP ? (i >> 4) : NP - 1) * 16 + (i & 15)];
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int b2 = 0; b2 < NB; ++b2) {
        float (&cur)[32] = wq[b2 & 1];
        RS_SS_ARRIVED(cur);
        __builtin_amdgcn_sched_barrier(0);
        if (b2 + 1 < NB) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int piece = 2 * (b2 + 1) + (i >> 4);
                wq[(b2 + 1) & 1][i] = W[(piece < NP ? piece : NP - 1) * 16 + (i & 15)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int b = 2 * b2 + hb;
            if (b < NP) {
                const int k = b / CH, ch = b % CH;
                if (ch == 0) { a0 = 0.0f; a1 = 0.0f; }
#pragma unroll
                for (int o = 0; o < 16; o += 2) {
                    a0 = fmaf(cur[16 * hb + o], cval(16 * ch + o), a0);
                    a1 = fmaf(cur[16 * hb + o + 1], cval(16 * ch + o + 1), a1);
                }
                asm volatile("" : "+v"(a0), "+v"(a1));
                if (ch == CH - 1) out[k] += a0 + a1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
