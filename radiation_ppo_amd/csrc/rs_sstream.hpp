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
#include <type_traits>

typedef const float __attribute__((address_space(4))) * rs_cmem_t;
typedef float rs_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rs_cmem_t rs_as_cmem(const float* p) { return (rs_cmem_t)(uintptr_t)p; }

#define RS_SS_ARRIVED(w) asm volatile("" :: "s"((w)[0]), "s"((w)[16]))

// NC (16 or 8) output columns starting at column C0 of a k-major block with row stride STRIDE: acc[NC] += W[:, C0 .. C0 + NC)^T c.
// A block is 32 SGPRs: two rows of 16 columns or four rows of 8 (s_load_dwordx8) -- the 8-wide form serves matrices whose width is
// 8 (mod 16): hid_obs' 24 outputs are one 16-wide and one 8-wide chunk instead of two 16-wide ones with 8 columns of padding.
template <int K, int STRIDE, int C0, int NC, typename F>
__device__ __forceinline__ void rs_ss_mv_cols(rs_cmem_t W, F cval, float (&acc)[NC]) {
    static_assert(NC == 16 || NC == 8, "16 or 8 columns per chunk");
    constexpr int RPB = 32 / NC;                                     // rows per block: 2 or 4
    constexpr int NB = (K + RPB - 1) / RPB;                          // a short last block re-reads row K - 1 and skips its FMAs
    float wq[2][32];
    // the accumulators as explicit pairs: v_pk_fma_f32 acc[2q : 2q+1] += s[w : w+1] * (c, c) takes the SGPR pair of two adjacent columns
    // and the lane's value broadcast by op_sel -- two multiply-adds per issue slot.  (Left as scalar fmaf the chain compiled to one
    // v_fmac_f32 per multiply-add: K11 - K15 issued 2 554 / 2 538 / 1 661 / 4 518 of them, half of all their instructions, round 4.)
    rs_v2f a2[NC / 2];
#pragma unroll
    for (int q = 0; q < NC / 2; ++q) a2[q] = (rs_v2f){acc[2 * q], acc[2 * q + 1]};
#pragma unroll
    for (int i = 0; i < 32; ++i) wq[0][i] = W[((i / NC) < K ? (i / NC) : K - 1) * STRIDE + C0 + (i % NC)];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float (&cur)[32] = wq[b & 1];
        RS_SS_ARRIVED(cur);
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < NB) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = RPB * (b + 1) + i / NC;
                wq[(b + 1) & 1][i] = W[(row < K ? row : K - 1) * STRIDE + C0 + (i % NC)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RPB; ++r) {
            if (RPB * b + r < K) {
                const float c = cval(RPB * b + r);
#pragma unroll
                for (int q = 0; q < NC / 2; ++q)
                    a2[q] = __builtin_elementwise_fma((rs_v2f){cur[r * NC + 2 * q], cur[r * NC + 2 * q + 1]}, (rs_v2f){c, c}, a2[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < NC / 2; ++q) asm volatile("" : "+v"(a2[q]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 0; o < NC; ++o) acc[o] = a2[o / 2][o & 1];
}

// out[0 .. OUTR) += W^T c for a k-major [K][OUTP] block (OUTP = the row stride, a multiple of 16; OUTR <= OUTP the columns that are
// real: a multiple of 8 -- columns OUTR .. OUTP - 1 are padding and are neither read nor written)
template <int K, int OUTP, int OUTR = OUTP, typename F>
__device__ __forceinline__ void rs_ss_mv(rs_cmem_t W, F cval, float (&out)[OUTP]) {
    static_assert(OUTP % 16 == 0 && OUTR % 8 == 0 && OUTR <= OUTP && OUTR > OUTP - 16, "16 outputs per chunk, an 8-wide last chunk allowed");
    auto chunk = [&](auto c0, auto nc) {
        constexpr int C0 = decltype(c0)::value, NC = decltype(nc)::value;
        float acc[NC];
#pragma unroll
        for (int o = 0; o < NC; ++o) acc[o] = out[C0 + o];
        rs_ss_mv_cols<K, OUTP, C0, NC>(W, cval, acc);
#pragma unroll
        for (int o = 0; o < NC; ++o) out[C0 + o] = acc[o];
    };
    constexpr int FULL = OUTR / 16;
    if constexpr (FULL >= 1) chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
    if constexpr (FULL >= 2) chunk(std::integral_constant<int, 16>{}, std::integral_constant<int, 16>{});
    if constexpr (FULL >= 3) chunk(std::integral_constant<int, 32>{}, std::integral_constant<int, 16>{});
    if constexpr (FULL >= 4) chunk(std::integral_constant<int, 48>{}, std::integral_constant<int, 16>{});
    if constexpr (FULL >= 5) chunk(std::integral_constant<int, 64>{}, std::integral_constant<int, 16>{});
    static_assert(FULL <= 5, "up to 80 outputs");
    if constexpr (OUTR % 16 == 8) chunk(std::integral_constant<int, 16 * FULL>{}, std::integral_constant<int, 8>{});
}

// out[k] += sum_o W[k][o] c(o), o < OUTR, on the SAME k-major [K][OUTP] block (the transposed product of a backward pass: a dot product
// along each row).  A row is cut into 16-wide pieces and, when OUTR is 8 (mod 16), one 8-wide piece; pieces are streamed two 16-wide
// slots per block (32 SGPRs) in the same wait -> request -> FMA order, two partial sums per row.
template <int K, int OUTP, int OUTR = OUTP, typename F>
__device__ __forceinline__ void rs_ss_mvt(rs_cmem_t W, F cval, float (&out)[K]) {
    static_assert(OUTP % 16 == 0 && OUTR % 8 == 0 && OUTR <= OUTP && OUTR > OUTP - 16, "16 weights per piece, an 8-wide last piece allowed");
    constexpr int CH = (OUTR + 15) / 16, NP = K * CH, NB = (NP + 1) / 2;
    constexpr bool TAIL8 = OUTR % 16 == 8;
    // piece -> (row, first column, width); the slot of an 8-wide piece loads 8 weights (s_load_dwordx8), the rest of the slot is unused
    float wq[2][32];
    auto request = [&](int b2, float (&dst)[32]) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int piece = 2 * b2 + hb < NP ? 2 * b2 + hb : NP - 1;
            const int k = piece / CH, ch = piece % CH;
            const int w = (TAIL8 && ch == CH - 1) ? 8 : 16;
#pragma unroll
            for (int o = 0; o < 16; ++o)
                if (o < w) dst[16 * hb + o] = W[k * OUTP + 16 * ch + o];
        }
    };
    request(0, wq[0]);
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int b2 = 0; b2 < NB; ++b2) {
        float (&cur)[32] = wq[b2 & 1];
        RS_SS_ARRIVED(cur);
        __builtin_amdgcn_sched_barrier(0);
        if (b2 + 1 < NB) request(b2 + 1, wq[(b2 + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int b = 2 * b2 + hb;
            if (b < NP) {
                const int k = b / CH, ch = b % CH;
                const int w = (TAIL8 && ch == CH - 1) ? 8 : 16;
                if (ch == 0) { a0 = 0.0f; a1 = 0.0f; }
#pragma unroll
                for (int o = 0; o < 16; o += 2) {
                    if (o < w) {
                        a0 = fmaf(cur[16 * hb + o], cval(16 * ch + o), a0);
                        a1 = fmaf(cur[16 * hb + o + 1], cval(16 * ch + o + 1), a1);
                    }
                }
                asm volatile("" : "+v"(a0), "+v"(a1));
                if (ch == CH - 1) out[k] += a0 + a1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
