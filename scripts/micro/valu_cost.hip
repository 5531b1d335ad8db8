// valu_cost.hip -- what one vector instruction costs a SIMD on gfx950, by kind, at 1 / 2 / 4 waves per SIMD.
// Every kernel runs `iters` trips of 32 copies of one instruction (independent destinations) on 256 x 4 x W waves; the time per
// instruction and SIMD is reported relative to v_fma_f32 at four waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP32(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(name, body, ...)                                                                           \
    __global__ void __launch_bounds__(256) name(float* out, int iters) {                                    \
        float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;        \
        float b0 = 0.5f, b1 = 0.25f;                                                                        \
        for (int it = 0; it < iters; ++it) {                                                                \
            asm volatile(REP4(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : __VA_ARGS__); \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                        \
    }

// each body = 8 instructions on a0..a7 (%0..%7), b0 = %8, b1 = %9
KERNEL(k_fma, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n", "memory")
KERNEL(k_fmac_s, "v_fmac_f32 %0, s4, %8\n v_fmac_f32 %1, s5, %8\n v_fmac_f32 %2, s6, %8\n v_fmac_f32 %3, s7, %8\n v_fmac_f32 %4, s4, %9\n v_fmac_f32 %5, s5, %9\n v_fmac_f32 %6, s6, %9\n v_fmac_f32 %7, s7, %9\n", "memory")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n", "memory")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %9\n v_mov_b32 %3, %9\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n", "memory")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n", "memory")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %8, %9, s[4:5]\n v_cndmask_b32 %1, %8, %9, s[4:5]\n v_cndmask_b32 %2, %8, %9, s[6:7]\n v_cndmask_b32 %3, %8, %9, s[6:7]\n v_cndmask_b32 %4, %8, %9, s[4:5]\n v_cndmask_b32 %5, %8, %9, s[4:5]\n v_cndmask_b32 %6, %8, %9, s[6:7]\n v_cndmask_b32 %7, %8, %9, s[6:7]\n", "memory")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n", "vcc")
KERNEL(k_readlane, "v_readlane_b32 s4, %0, 3\n v_readlane_b32 s5, %1, 3\n v_readlane_b32 s6, %2, 3\n v_readlane_b32 s7, %3, 3\n v_readlane_b32 s4, %4, 3\n v_readlane_b32 s5, %5, 3\n v_readlane_b32 s6, %6, 3\n v_readlane_b32 s7, %7, 3\n", "s4", "s5", "s6", "s7")
KERNEL(k_writelane, "v_writelane_b32 %0, s4, 3\n v_writelane_b32 %1, s5, 3\n v_writelane_b32 %2, s6, 3\n v_writelane_b32 %3, s7, 3\n v_writelane_b32 %4, s4, 5\n v_writelane_b32 %5, s5, 5\n v_writelane_b32 %6, s6, 5\n v_writelane_b32 %7, s7, 5\n", "memory")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n", "memory")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n", "memory")
KERNEL(k_max, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n", "memory")
KERNEL(k_snop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n", "memory")
KERNEL(k_salu, "s_add_u32 s4, s4, 1\n s_add_u32 s5, s5, 1\n s_add_u32 s6, s6, 1\n s_add_u32 s7, s7, 1\n s_add_u32 s4, s4, 1\n s_add_u32 s5, s5, 1\n s_add_u32 s6, s6, 1\n s_add_u32 s7, s7, 1\n", "s4", "s5", "s6", "s7", "scc")

KERNEL(k_cnd_fma, "v_cndmask_b32 %0, %8, %9, vcc\n v_fma_f32 %1, %8, %9, %1\n v_cndmask_b32 %2, %8, %9, vcc\n v_fma_f32 %3, %8, %9, %3\n v_cndmask_b32 %4, %8, %9, vcc\n v_fma_f32 %5, %8, %9, %5\n v_cndmask_b32 %6, %8, %9, vcc\n v_fma_f32 %7, %8, %9, %7\n", "memory")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %8, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %8, %9, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %8, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %8, %9, vcc\n", "vcc")
KERNEL(k_cmp64_cnd, "v_cmp_lt_f32 s[4:5], %0, %8\n v_cndmask_b32 %1, %8, %9, s[4:5]\n v_cmp_lt_f32 s[6:7], %2, %8\n v_cndmask_b32 %3, %8, %9, s[6:7]\n v_cmp_lt_f32 s[4:5], %4, %8\n v_cndmask_b32 %5, %8, %9, s[4:5]\n v_cmp_lt_f32 s[6:7], %6, %8\n v_cndmask_b32 %7, %8, %9, s[6:7]\n", "s4", "s5", "s6", "s7")
KERNEL(k_cnd_dep, "v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n", "memory")
KERNEL(k_ashr_and, "v_ashrrev_i32 %0, 31, %0\n v_and_b32 %1, %1, %8\n v_ashrrev_i32 %2, 31, %2\n v_and_b32 %3, %3, %8\n v_ashrrev_i32 %4, 31, %4\n v_and_b32 %5, %5, %8\n v_ashrrev_i32 %6, 31, %6\n v_and_b32 %7, %7, %8\n", "memory")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n", "memory")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n", "memory")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n", "memory")

// packed FMA: 64-bit operands
#define KERNEL2(name, body)                                                                                 \
    __global__ void __launch_bounds__(256) name(float* out, int iters) {                                    \
        typedef float v2f __attribute__((ext_vector_type(2)));                                              \
        v2f a0 = {(float)threadIdx.x, 1.f}, a1 = {1.f, 2.f}, a2 = {2.f, 3.f}, a3 = {3.f, 4.f};               \
        v2f b0 = {0.5f, 0.25f};                                                                             \
        for (int it = 0; it < iters; ++it) {                                                                \
            asm volatile(REP4(REP4(body) ) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "memory"); \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y;        \
    }
// body = 4 instructions; REP4(REP4()) = 64... keep the count in the table below
KERNEL2(k_pk_v, "v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3\n")
KERNEL2(k_pk_s, "v_pk_fma_f32 %0, s[4:5], %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, s[6:7], %4, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, s[4:5], %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, s[6:7], %4, %3 op_sel_hi:[1,0,1]\n")
KERNEL2(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n")
KERNEL2(k_mov64, "v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4\n")
KERNEL2(k_pk_add, "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")

#define KERNEL2V(name, body)                                                                                \
    __global__ void __launch_bounds__(256) name(float* out, int iters) {                                    \
        typedef float v2f __attribute__((ext_vector_type(2)));                                              \
        v2f a0 = {(float)threadIdx.x, 1.f}, a1 = {1.f, 2.f}, a2 = {2.f, 3.f}, a3 = {3.f, 4.f};               \
        v2f b0 = {0.5f, 0.25f};                                                                             \
        for (int it = 0; it < iters; ++it) {                                                                \
            asm volatile(REP4(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "memory");        \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y;        \
    }
KERNEL2V(k_pk_s16, "v_pk_fma_f32 %0, s[4:5], %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, s[6:7], %4, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, s[8:9], %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, s[10:11], %4, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, s[12:13], %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, s[14:15], %4, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, s[16:17], %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, s[18:19], %4, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, s[20:21], %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, s[22:23], %4, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, s[24:25], %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, s[26:27], %4, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, s[28:29], %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, s[30:31], %4, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, s[32:33], %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, s[34:35], %4, %3 op_sel_hi:[1,0,1]\n ")

struct Case { const char* name; void (*k)(float*, int); int per_trip; };

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 4 * 256 * sizeof(float));
    std::vector<Case> cases = {
        {"v_fma_f32 (VGPR operands)", k_fma, 32}, {"v_fmac_f32 with an SGPR operand", k_fmac_s, 32}, {"v_pk_fma_f32 (VGPR pairs)", k_pk_v, 64},
        {"v_pk_fma_f32 (SGPR pair x splat)", k_pk_s, 64}, {"v_pk_fma_f32 (16 different SGPR pairs x splat)", k_pk_s16, 64}, {"v_pk_add_f32", k_pk_add, 64}, {"v_add_u32", k_add_u32, 32}, {"v_mov_b32", k_mov, 32}, {"v_mov_b64", k_mov64, 64},
        {"v_max_f32", k_max, 32}, {"v_cndmask_b32 (vcc)", k_cndmask, 32}, {"v_cndmask_b32 (SGPR pair mask)", k_cndmask_s, 32}, {"v_cmp_lt_f32 -> vcc", k_cmp, 32},
        {"v_lshl_add_u64", k_lshl_add_u64, 64}, {"v_mul_lo_u32", k_mul_lo, 32}, {"v_exp_f32", k_exp, 32}, {"v_readlane_b32", k_readlane, 32}, {"v_writelane_b32", k_writelane, 32},
        {"v_cndmask(vcc) / v_fma alternating", k_cnd_fma, 32}, {"v_cmp -> vcc, v_cndmask(vcc) pairs", k_cmp_cnd, 32}, {"v_cmp -> s[a:b], v_cndmask(s[a:b]) pairs", k_cmp64_cnd, 32}, {"v_cndmask(vcc), dst = src0", k_cnd_dep, 32}, {"v_ashrrev_i32 / v_and_b32", k_ashr_and, 32}, {"v_add_f32", k_add_f32, 32}, {"v_mul_f32", k_mul_f32, 32}, {"v_rcp_f32", k_rcp, 32}, {"s_nop 0", k_snop, 32}, {"s_add_u32", k_salu, 32}};
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double ref = 0;
    for (int W : {4, 3, 1}) {
        printf("---- %d wave(s) per SIMD (256 CUs x %d workgroups of 256 threads)\n", W, W);
        for (auto& c : cases) {
            c.k<<<256 * W, 256>>>(out, 10);
            hipEventRecord(e0);
            c.k<<<256 * W, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per = (double)ms * 1e6 / ((double)iters * c.per_trip * W);      // ns per instruction and SIMD
            if (ref == 0) ref = per;
            printf("  %-36s %7.3f ms  %6.3f ns per instruction and SIMD = %5.2f x v_fma_f32@4\n", c.name, ms, per, per / ref);
        }
    }
    return 0;
}
