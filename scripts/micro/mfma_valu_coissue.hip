// mfma_valu_coissue.hip -- does non-matrix work overlap with the f32-input MFMA on gfx950?
//
// K7 (rs_ppo_grad2_kernel) runs two waves per SIMD and sits at 0.60 of the f32 MFMA roofline.  DESIGN.md (round 1) assumed
// "the f32-input MFMA shares the FP32 lanes with the VALU, so nothing overlaps".  This measures it:
//   * one wave per SIMD issuing back-to-back v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: cycles per MFMA;
//   * one wave per SIMD issuing v_fma_f32 / v_exp_f32 / ds_read_b32 streams: cycles per instruction;
//   * TWO waves on the same SIMD (waves w and w+4 of a 512-thread workgroup): one MFMA-only, the partner VALU-/LDS-only,
//     each timed with s_memtime; overlap = (t_mfma_alone + t_other_alone) / max(t_mfma_co, t_other_co);
//   * ONE wave interleaving k VALU instructions per MFMA: cycles per MFMA as a function of k (what a single in-order wave
//     can hide behind its own matrix instructions).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coissue mfma_valu_coissue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { M_IDLE = 0, M_MFMA32 = 1, M_FMA = 2, M_EXP = 3, M_LDS = 4, M_MFMA16 = 5, M_MIX = 6, M_MFMA4 = 7 };

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// body of one role; iters = loop trips; every trip issues UNR instructions of the kind
template <int K_MIX>
__device__ __forceinline__ float run_role(int mode, int iters, float seed, const float* lds) {
    float r = 0.f;
    if (mode == M_MFMA32 || mode == M_MIX) {
        f32x16 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] = seed * (a + 1);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = seed + j;
        const float av = seed + 1.f, bv = seed - 1.f;
        if (mode == M_MIX) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < K_MIX; ++k) f[k & 7] = fmaf(f[k & 7], 1.0001f, 0.5f);
                    // keep the k VALU instructions next to their MFMA
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, K_MIX, 0);
                }
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) r += acc[a][0] + acc[a][7];
#pragma unroll
        for (int j = 0; j < 8; ++j) r += f[j];
    } else if (mode == M_MFMA16) {
        f32x4 acc[8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][i] = seed * (a + 1);
        const float av = seed + 1.f, bv = seed - 1.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) r += acc[a][0];
    } else if (mode == M_MFMA4) {
        // v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks, K = 1 (thin products at 100 % fill)
        f32x4 acc[8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[a][i] = seed * (a + 1);
        const float av = seed + 1.f, bv = seed - 1.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) r += acc[a][0];
    } else if (mode == M_FMA) {
        float f[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) f[j] = seed + j;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = fmaf(f[j], 1.0001f, 0.5f);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r += f[j];
    } else if (mode == M_EXP) {
        float f[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) f[j] = seed * 0.01f + j * 0.001f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = __builtin_amdgcn_exp2f(f[j]) * 0.25f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r += f[j];
    } else if (mode == M_LDS) {
        const int lane = threadIdx.x & 63;
        float s = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) s += lds[(j * 64 + lane + it) & 4095];
        }
        r = s;
    }
    return r;
}

template <int K_MIX>
__global__ void __launch_bounds__(512, 2) coissue_kernel(int mode_lo, int mode_hi, int iters_lo, int iters_hi, float seed,
                                                         unsigned long long* cyc, float* sink) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const bool hi = wave >= 4;
    const int mode = __builtin_amdgcn_readfirstlane(hi ? mode_hi : mode_lo), iters = __builtin_amdgcn_readfirstlane(hi ? iters_hi : iters_lo);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = stamp();
    __builtin_amdgcn_sched_barrier(0);
    const float r = run_role<K_MIX>(mode, iters, seed, lds);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(r));
    const unsigned long long t1 = stamp();
    __builtin_amdgcn_sched_barrier(0);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678f) sink[0] = r;
}

struct Res { double lo, hi; };

template <int K_MIX>
static Res run(int mode_lo, int mode_hi, int iters_lo, int iters_hi, unsigned long long* d_cyc, float* d_sink, int threads) {
    const int blocks = 256;
    std::vector<unsigned long long> h(blocks * 8);
    std::vector<double> lo, hi;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(d_cyc, 0, sizeof(unsigned long long) * blocks * 8));
        hipLaunchKernelGGL(coissue_kernel<K_MIX>, dim3(blocks), dim3(threads), 0, 0, mode_lo, mode_hi, iters_lo, iters_hi, 1.0f + rep, d_cyc, d_sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost));
        if (rep == 0) continue;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < threads / 64; ++w) (w < 4 ? lo : hi).push_back((double)h[b * 8 + w]);
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    return Res{med(lo), med(hi)};
}

int main() {
    unsigned long long* d_cyc; float* d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 256 * 8));
    CHECK(hipMalloc(&d_sink, 64));
    const int IT = 2000;
    printf("# s_memtime cycles, median over 256 CUs x waves; 'per' = cycles per instruction of that role\n");
    // ---- alone, one wave per SIMD (256 threads)
    struct { const char* name; int mode; int per_iter; } roles[] = {
        {"mfma_f32_32x32x2", M_MFMA32, 4}, {"mfma_f32_16x16x4", M_MFMA16, 8}, {"mfma_f32_4x4x1_16b", M_MFMA4, 8}, {"v_fma_f32", M_FMA, 16},
        {"v_exp_f32(+mul)", M_EXP, 16}, {"ds_read_b32", M_LDS, 16}};
    double alone[8] = {0};
    for (auto& r : roles) {
        Res a = run<0>(r.mode, M_IDLE, IT, 0, d_cyc, d_sink, 256);
        alone[r.mode] = a.lo;
        printf("alone  1 wave/SIMD  %-18s %9.0f cyc  per %.2f\n", r.name, a.lo, a.lo / (IT * r.per_iter));
    }
    // ---- same role on both waves of a SIMD
    for (auto& r : roles) {
        Res a = run<0>(r.mode, r.mode, IT, IT, d_cyc, d_sink, 512);
        printf("same   2 waves/SIMD %-18s lo %9.0f hi %9.0f  per(pair) %.2f\n", r.name, a.lo, a.hi,
               std::max(a.lo, a.hi) / (2.0 * IT * r.per_iter));
    }
    // ---- MFMA wave + partner with other work, iteration counts chosen so both last about equally long alone
    for (int mm : {M_MFMA32, M_MFMA16}) {
        for (auto& r : roles) {
            if (r.mode == M_MFMA32 || r.mode == M_MFMA16 || r.mode == M_MFMA4) continue;
            const int it_other = (int)(IT * alone[mm] / alone[r.mode]);
            Res solo_o = run<0>(M_IDLE, r.mode, 0, it_other, d_cyc, d_sink, 512);
            Res solo_m = run<0>(mm, M_IDLE, IT, 0, d_cyc, d_sink, 512);
            Res co = run<0>(mm, r.mode, IT, it_other, d_cyc, d_sink, 512);
            const double serial = solo_m.lo + solo_o.hi, both = std::max(co.lo, co.hi);
            printf("co-run %-16s + %-16s alone %8.0f / %8.0f   together mfma %8.0f other %8.0f   overlap %.2f (1.0 = serial, 2.0 = free)\n",
                   mm == M_MFMA32 ? "mfma_32x32x2" : "mfma_16x16x4", r.name, solo_m.lo, solo_o.hi, co.lo, co.hi, serial / both);
        }
    }
    // ---- one wave: k dependent-free v_fma_f32 per MFMA
    {
        Res b0 = run<0>(M_MFMA32, M_IDLE, IT, 0, d_cyc, d_sink, 256);
        printf("mix    1 wave/SIMD  k= 0 fma per mfma_32x32x2: %.1f cyc per MFMA\n", b0.lo / (IT * 4.0));
#define MIXRUN(K) { Res m = run<K>(M_MIX, M_IDLE, IT, 0, d_cyc, d_sink, 256); \
        printf("mix    1 wave/SIMD  k=%2d fma per mfma_32x32x2: %.1f cyc per MFMA\n", K, m.lo / (IT * 4.0)); }
        MIXRUN(4) MIXRUN(8) MIXRUN(12) MIXRUN(16) MIXRUN(24) MIXRUN(32)
#define MIXRUN2(K) { Res m = run<K>(M_MIX, M_MIX, IT, IT, d_cyc, d_sink, 512); \
        printf("mix    2 waves/SIMD k=%2d fma per mfma_32x32x2: %.1f cyc per MFMA (pair issues 2 MFMAs)\n", K, std::max(m.lo, m.hi) / (IT * 4.0 * 2.0)); }
        MIXRUN2(4) MIXRUN2(8) MIXRUN2(16) MIXRUN2(32)
    }
    return 0;
}
