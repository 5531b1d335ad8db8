// rs_wave.hpp -- wave-wide reductions and the float64 prefix sum of the particle filter on the DPP data path (gfx9 row operations).
//
// __shfl_xor / __shfl_up compile to ds_bpermute_b32: every step of a butterfly is an LDS round trip (~60 cycles, measured by
// scripts/micro/sload_latency.hip) in a dependent chain -- ~50 of them per PFGRU step (two log-softmaxes, the CDF scan, the head's
// sums), 13 % of K11's cycles and fully exposed in the one-wave-per-SIMD backward walk of K13.  DPP modifiers move data between the
// lanes of a 16-lane row inside the VALU (no LDS); the four row results are combined through v_readlane.
//     quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: every lane of a row ends up with the row's total;
//     row_shr 1, 2, 4, 8 (zero fill) + row_bcast 15 / 31: inclusive prefix sum over the 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int rs_dpp_i(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false);     // lanes without a source keep `old`
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float rs_dpp_f(float old, float v) {
    return __int_as_float(rs_dpp_i<CTRL, ROW_MASK, BANK_MASK>(__float_as_int(old), __float_as_int(v)));
}
template <int L>
__device__ __forceinline__ float rs_lane_f(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); }

constexpr int RS_DPP_QUAD_XOR1 = 0xB1, RS_DPP_QUAD_XOR2 = 0x4E, RS_DPP_ROW_MIRROR = 0x140, RS_DPP_ROW_HALF_MIRROR = 0x141,
              RS_DPP_ROW_SHR = 0x110, RS_DPP_ROW_BCAST15 = 0x142, RS_DPP_ROW_BCAST31 = 0x143;

// sum over the 64 lanes, the same value in every lane (inactive lanes must carry 0)
__device__ __forceinline__ float rs_wave_sum(float v) {
    v += rs_dpp_f<RS_DPP_QUAD_XOR1>(0.0f, v);
    v += rs_dpp_f<RS_DPP_QUAD_XOR2>(0.0f, v);
    v += rs_dpp_f<RS_DPP_ROW_HALF_MIRROR>(0.0f, v);
    v += rs_dpp_f<RS_DPP_ROW_MIRROR>(0.0f, v);
    return (rs_lane_f<0>(v) + rs_lane_f<16>(v)) + (rs_lane_f<32>(v) + rs_lane_f<48>(v));
}
// maximum over the 64 lanes, the same value in every lane (inactive lanes must carry -inf)
__device__ __forceinline__ float rs_wave_max(float v) {
    v = fmaxf(v, rs_dpp_f<RS_DPP_QUAD_XOR1>(v, v));
    v = fmaxf(v, rs_dpp_f<RS_DPP_QUAD_XOR2>(v, v));
    v = fmaxf(v, rs_dpp_f<RS_DPP_ROW_HALF_MIRROR>(v, v));
    v = fmaxf(v, rs_dpp_f<RS_DPP_ROW_MIRROR>(v, v));
    return fmaxf(fmaxf(rs_lane_f<0>(v), rs_lane_f<16>(v)), fmaxf(rs_lane_f<32>(v), rs_lane_f<48>(v)));
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double rs_dpp_d(double v) {                    // lanes without a source (or masked rows) read 0.0
    const long long b = __double_as_longlong(v);
    const int lo = rs_dpp_i<CTRL, ROW_MASK>(0, (int)(b & 0xFFFFFFFFll)), hi = rs_dpp_i<CTRL, ROW_MASK>(0, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// inclusive prefix sum over the lanes in float64 (lane i: c[0] + ... + c[i])
__device__ __forceinline__ double rs_wave_scan(double c) {
    c += rs_dpp_d<RS_DPP_ROW_SHR + 1>(c);
    c += rs_dpp_d<RS_DPP_ROW_SHR + 2>(c);
    c += rs_dpp_d<RS_DPP_ROW_SHR + 4>(c);
    c += rs_dpp_d<RS_DPP_ROW_SHR + 8>(c);
    c += rs_dpp_d<RS_DPP_ROW_BCAST15, 0xA>(c);                            // rows 1 and 3 take the total of the row before them
    c += rs_dpp_d<RS_DPP_ROW_BCAST31, 0xC>(c);                            // rows 2 and 3 take the total of the first two rows
    return c;
}
template <int L>
__device__ __forceinline__ double rs_lane_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), L), hi = __builtin_amdgcn_readlane((int)(b >> 32), L);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
