// rs_cnn.hip -- K9/K10: the convolutional trunk of the RAD-TEAM CNN actor / critic
// (algos/multiagent/NeuralNetworkCores/RADTEAM_core.py:962-1023 Actor, :1211-1271 Critic):
//     conv3x3(Cin->8, pad 1) - ReLU - maxpool 2x2/2 - conv3x3(8->16, pad 1) - ReLU - flatten(2704)
// forward and backward, straight from the resident heat maps.  The channel counts (6/4 -> 8 -> 16) are far too
// small for an implicit-GEMM library convolution (MIOpen spends minutes tuning and then runs at ~5 % of the f32
// peak on these shapes), so the trunk is written directly:
//   * whole images in LDS (zero-padded planes), workgroups stride over the batch (persistent grid): forward three images per
//     512-thread workgroup at a time (507 lanes own a pooled cell), backward one image per 256-thread workgroup;
//   * the input stack is never materialised in HBM: the 4 shared maps of a sample are read once (11.7 KB).  Of the actor's six
//     channels (CNNBase.get_map_stack, RADTEAM_core.py:1791-1836) only these four are dense; the owner's prediction and location
//     maps are ONE-HOT and `others = combined - location`, so conv1 runs over the 4 dense planes (`combined` with the weights of
//     `others`) and the one-hots enter as 3x3 weight STAMPS on the <= 9 output pixels around the owner's cell
//     (w1[:, pred] at the prediction cell, w1[:, loc] - w1[:, others] at the location cell): 2 of 6 channels of conv1 and of dW1 are
//     not convolved at all.  Backward, the same channels' weight gradients are 9-element gathers of dZ1 (through the pool's arg-max);
//   * conv1 is evaluated per POOLED cell (thread = cell: a 4x4 input window per channel in registers feeds the
//     2x2 block of conv outputs, 8 channels each -> 32 accumulators), so ReLU + max-pool happen in registers;
//     row/column 26 of the conv1 output never reach the pool (27 = 2*13 + 1) and are not computed;
//   * conv2: thread = output pixel, 16 accumulators;
//   * weights are staged in LDS once per workgroup and read as wave-uniform (broadcast) float4s;
//   * backward: dP1 = conv2^T(dZ2) on the VALU (thread = pixel), dW2 = dZ2 x patches(P1) on the matrix cores
//     (v_mfma_f32_16x16x4_f32, contraction over the 169 pixels, accumulators live across all images of the
//     workgroup), dW1 through the pool's argmax: only one of the four conv1 pixels of a cell carries gradient, so
//     dW1 costs 8*169*Cin*9 MACs per image instead of 8*676*Cin*9; per-workgroup partial sums go to a slab that
//     the caller reduces (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/radsearch.h"

// Diagnostic build only (-DRS_CNN_STAMPS, scripts/cnn_stamps.py): s_memtime stamps at the phase boundaries of an image round, summed per
// wave and added to a device table [kernel][wave][phase] when the workgroup leaves.  Read the SHARES (the stamps pin the schedule).
#ifdef RS_CNN_STAMPS
__device__ unsigned long long rs_cnn_cyc[2][8][8];
#define CNN_DECL unsigned long long cs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long cs_t = __builtin_amdgcn_s_memtime();
#define CNN_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long cs_n = __builtin_amdgcn_s_memtime(); cs_acc[i] += cs_n - cs_t; cs_t = cs_n; __builtin_amdgcn_sched_barrier(0); }
#define CNN_FLUSH(k) if ((threadIdx.x & 63) == 0) { for (int q = 0; q < 8; ++q) atomicAdd(&rs_cnn_cyc[k][threadIdx.x >> 6][q], cs_acc[q]); }
#else
#define CNN_DECL
#define CNN_STAMP(i)
#define CNN_FLUSH(k)
#endif

namespace {

#ifndef CNN_BWD_WAVES
#define CNN_BWD_WAVES 3
#endif
#ifndef CNN_BWD_PREFETCH
#define CNN_BWD_PREFETCH 1
#endif
#ifndef CNN_NTB
#define CNN_NTB 256
#endif
constexpr int CNN_NT_BWD = CNN_NTB;         // backward: a 4th wave that owns no pixel shares the staging, the dW2 MFMA steps and
                                            // the dW1 gathers -- the kernel is wait-bound (56 % of its wave cycles parked), not VALU-bound
constexpr int MAPW = 27, MAPC = 729;        // heat-map side and cells
constexpr int PW = 13, PC = 169;            // pooled side and cells
constexpr int DP = 4;                       // dense input planes in LDS: combined, readings, visits, obstacles (actor and critic alike)
constexpr int XP_RS = 28, XP_PLANE = 28 * 28 + 6;   // padded input plane: xp[r][c] = x[r-1][c-1], r,c in [0,28): the bottom/right
                                                // border only feeds conv1 row/column 26, which the pool drops; even stride (b64 reads)
constexpr int XP_PLANE_B = 28 * 28 + 7;         // backward: ODD plane stride -- the dW1 gather's two channel halves (2 planes apart) must
                                                // land in the other 16 LDS banks (see the bank map in rs_cnn_bwd_kernel)
constexpr int PP_RS = 15, PP_PLANE = 15 * 15;   // padded pooled plane (forward)
// Backward: K10 is LDS-bandwidth bound (profiles/r04_cnn_sq_counters.txt: the LDS pipe busy ~90 % of the time, more than half of it bank
// conflicts), so the padded P1 / dZ2 planes get the strides that make the matrix-core operand reads conflict free -- row stride 18, plane
// stride 278 (= 22 mod 32): the 16 taps x 2 pixels a half-wave reads as B operands fall into 32 different banks (2.8-way on average
// with 15 / 225), the 16 channels x 2 pixels of the A operand too -- and dP1's pixel threads take their pixels from BWD_SLOT_PIXEL,
// a table that gives every half-wave 28-29 pixels whose addresses 18 py + px are distinct mod 32 (with pixel = thread index two of
// the three 13-pixel rows a half-wave touches overlap: every one of the 144 neighbourhood reads took two passes).
constexpr int PPB_RS = 18, PPB_PLANE = 278;
// slot (thread of the three pixel waves; pass k, lane of the helper wave) -> pixel, 255 = none; generated for row stride 18
__device__ const unsigned char BWD_SLOT_PIXEL[192] = {
    0, 1, 2, 4, 6, 8, 9, 10, 12, 15, 19, 21, 22, 23, 25, 36, 39, 59, 60, 74, 76, 89, 92, 107, 122, 135, 148, 154, 161, 255, 255, 255,
    13, 16, 26, 28, 30, 31, 32, 34, 35, 41, 43, 44, 45, 47, 49, 50, 51, 58, 64, 81, 82, 88, 96, 98, 111, 141, 144, 157, 255, 255, 255, 255,
    11, 24, 52, 53, 54, 56, 57, 62, 65, 66, 67, 69, 71, 72, 73, 75, 77, 80, 86, 87, 90, 104, 110, 118, 120, 133, 152, 163, 255, 255, 255, 255,
    7, 14, 20, 33, 46, 61, 78, 79, 84, 91, 93, 94, 95, 97, 99, 101, 102, 103, 108, 109, 112, 114, 115, 116, 129, 132, 153, 167, 255, 255, 255, 255,
    5, 18, 29, 42, 48, 55, 63, 68, 83, 106, 117, 119, 121, 123, 124, 125, 127, 128, 130, 131, 134, 136, 137, 138, 140, 142, 151, 165, 255, 255, 255, 255,
    3, 17, 27, 37, 38, 40, 70, 85, 100, 105, 113, 126, 139, 143, 145, 146, 147, 149, 150, 155, 156, 158, 159, 160, 162, 164, 166, 168, 255, 255, 255, 255};
constexpr int C1 = 8, C2 = 16, FLAT = C2 * PC;  // 2704
constexpr int DP1_SPLIT = 13;                   // backward, dP1: output channels [0, 13) on the pixel threads, [13, 16) on the helper wave (measured: 16 -> 707, 13 -> 672, 12 -> 676, 10 -> 703 us per 32768 images)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// Weights are read through the scalar unit: a wave-uniform s_load_dwordx8/x16 from the constant address space feeds
// v_fma's SGPR operand, so they cost neither LDS return bandwidth nor VGPRs.  (Broadcast ds_reads still return
// 64 x 16 B per wave and made conv2 / dP1 LDS-bound.)  The scalar loads want [tap][channel] rows, produced once per
// call by rs_cnn_prep_kernel into a caller-provided scratch:
//   [w1t (Cin*9) x 8 | b1 8 | w2t 72 x 16 | b2 16 | w2b (co,ky,kx)=144 x 8 ci | stamps 2 x 9 x 8 (actor: prediction, location)]
typedef const float __attribute__((address_space(4))) * cmem_t;
__device__ __forceinline__ cmem_t as_cmem(const float* p) { return (cmem_t)(uintptr_t)p; }
constexpr int WT_W1(int) { return 0; }
constexpr int WT_B1(int cin) { return cin * 9 * 8; }
constexpr int WT_W2(int cin) { return cin * 9 * 8 + 8; }
constexpr int WT_B2(int cin) { return cin * 9 * 8 + 8 + 72 * 16; }
constexpr int WT_W2B(int cin) { return cin * 9 * 8 + 8 + 72 * 16 + 16; }
constexpr int WT_ST(int cin) { return cin * 9 * 8 + 8 + 72 * 16 + 16 + 144 * 8; }
constexpr int WT_TOTAL(int cin) { return WT_ST(cin) + 2 * 9 * 8; }

__global__ void rs_cnn_prep_kernel(int cin, const float* __restrict__ w1, const float* __restrict__ b1,
                                   const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ wt) {
    const int K1 = cin * 9;
    for (int e = threadIdx.x; w1 && e < 8 * K1; e += blockDim.x) {       // torch [co][ci][ky][kx] -> [(ci,ky,kx)][co]
        const int co = e / K1, k = e - co * K1;
        wt[WT_W1(cin) + k * 8 + co] = w1[e];
    }
    for (int e = threadIdx.x; e < 16 * 72; e += blockDim.x) {
        const int co = e / 72, k = e - co * 72;
        wt[WT_W2(cin) + k * 16 + co] = w2[e];                             // [(ci,ky,kx)][co]
        const int ci = k / 9, kk = k - ci * 9;
        wt[WT_W2B(cin) + (co * 9 + kk) * 8 + ci] = w2[e];                 // [(co,ky,kx)][ci]
    }
    if (w1 && cin == 6)
        for (int e = threadIdx.x; e < 72; e += blockDim.x) {                 // [tap][co]: what a one-hot input adds to conv1's output
            const int kk = e / 8, co = e - kk * 8;
            wt[WT_ST(cin) + e] = w1[(co * 6 + 0) * 9 + kk];                                   // prediction map (channel 0)
            wt[WT_ST(cin) + 72 + e] = w1[(co * 6 + 1) * 9 + kk] - w1[(co * 6 + 2) * 9 + kk];  // location (1) minus its share of `others` (2)
        }
    if (threadIdx.x < 8) wt[WT_B1(cin) + threadIdx.x] = b1 ? b1[threadIdx.x] : 0.0f;
    if (threadIdx.x < 16) wt[WT_B2(cin) + threadIdx.x] = b2 ? b2[threadIdx.x] : 0.0f;
}

// Weight rows through the scalar unit in the order  wait for this block -> request the next block -> this block's FMAs  (csrc/rs_sstream.hpp
// explains why: scalar loads return out of order, s_waitcnt lgkmcnt(0) drains everything in flight, and hipcc on its own requests a
// block and waits right behind it -- in K9 / K10 every tap group exposed a full scalar round trip, ~40 % of the SIMD time idle at
// 3-4 waves per SIMD, profiles/r04_cnn_phase_cycles.txt).  ROWS rows of RW floats (RW = 8 or 16) from W, RPB = 32 / RW rows per
// block, double buffered in 64 SGPRs; fully unrolled.  want(b): the caller requests whatever else block b needs (LDS values) next to
// the block's weights; use(row, w): the row's FMAs; pin(): the caller pins its accumulators ("+v") so that no FMA chain is sunk.
template <int ROWS, int RW, typename Want, typename Use, typename Pin>
__device__ __forceinline__ void cnn_wstream(cmem_t W, Want want, Use use, Pin pin) {
    static_assert(RW == 8 || RW == 16, "8 or 16 weights per row");
    constexpr int RPB = 32 / RW, NB = (ROWS + RPB - 1) / RPB;
    float wq[2][32];
    // the table's address as a value of its own: as a member of the kernel-argument block hipcc spilled the whole 8-SGPR tuple to VGPR
    // lanes and restored it with 16 v_readlane in front of every block's two loads
    asm volatile("" : "+s"(W));
#pragma unroll
    for (int i = 0; i < 32; ++i) wq[0][i] = W[((i / RW) < ROWS ? (i / RW) : ROWS - 1) * RW + i % RW];
    want(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float (&cur)[32] = wq[b & 1];
        asm volatile("" :: "s"(cur[0]), "s"(cur[8]), "s"(cur[16]), "s"(cur[24]));
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < NB) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = RPB * (b + 1) + i / RW;
                wq[(b + 1) & 1][i] = W[(row < ROWS ? row : ROWS - 1) * RW + i % RW];
            }
            want(b + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RPB; ++r)
            if (RPB * b + r < ROWS) use(RPB * b + r, &cur[r * RW]);
        pin();
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct CnnIn {
    const float* maps;        // [S][4][729]: combined, readings, visits, obstacles
    const int64_t* cells;     // [S][A] owner cells (actor) or nullptr (critic)
    const int64_t* pcells;    // [S][A] prediction cell or -1
    int A, agent;
    long long S;
};

// ---- one sample's inputs travel HBM -> registers -> LDS; the registers of sample s+grid are filled while sample s
// is being computed (software prefetch), so the per-image memory latency is off the critical path.
// Mapping: the 4 x 27 map rows of a sample are 108 contiguous rows of 27 floats.  G = NT / 27 thread groups of 27 threads
// (thread = column c of its group's rows p = g, g + G, ...): the global index is tid + i * 27 G (contiguous across the active
// threads), and the padded LDS address advances by a constant per row with one wrap test per element -- a first version
// decomposed every element index with four divisions by constants, 19 % of K9's instructions.
template <int NT>
struct CnnFetch {
    static constexpr int G = NT / MAPW;                                 // 9 (NT = 256, the backward kernel)
    static constexpr int PER_THREAD = (4 * MAPW + G - 1) / G;           // 12 rows per thread
    float m[PER_THREAD];
    int loc, pc;
};

// Backward (NT = 256): G = 9 groups of 27 threads, thread (g, c) takes map rows p = g + 9 i, i < 12.  Because 27 = 3 x 9, row p lies in
// plane i / 3 at row g + 9 (i % 3) -- no carry for any g < 9 -- so every global offset (243 i floats) and every padded LDS offset
// ((i / 3) PLANE + 9 (i % 3) rows) is a compile-time constant behind ONE per-thread base and ONE predicate (tid < 243).  The first
// version tested `e < 4 * MAPC` per element and walked the LDS address with a wrap test: a branch + exec mask per load and ~6 vector
// instructions per store, a third of K10's vector instructions (profiles/r04_cnn_phase_cycles.txt).
template <int CIN, int NT>
__device__ __forceinline__ void cnn_fetch(const CnnIn& in, long long s, CnnFetch<NT>& f) {
    static_assert(NT == 256 && CnnFetch<NT>::G == 9 && CnnFetch<NT>::PER_THREAD == 12, "the carry-free row walk assumes 9 groups x 12 rows");
    const float* src = in.maps + (size_t)s * 4 * MAPC + threadIdx.x;
    if (threadIdx.x < 9 * MAPW) {
#pragma unroll
        for (int i = 0; i < 12; ++i) f.m[i] = src[i * 9 * MAPW];
    }
    f.loc = -1; f.pc = -1;
    if (CIN == 6) {
        f.loc = (int)in.cells[(size_t)s * in.A + in.agent];
        f.pc = (int)in.pcells[(size_t)s * in.A + in.agent];
    }
}

template <int PLANE, int NT>
__device__ __forceinline__ void cnn_stage(const CnnFetch<NT>& f, float* xp) {
    const int g = threadIdx.x / MAPW, c = threadIdx.x - g * MAPW;
    if (threadIdx.x < 9 * MAPW) {
        float* d = xp + (g + 1) * XP_RS + c + 1;                        // row g of plane 0
#pragma unroll
        for (int i = 0; i < 12; ++i) d[(i / 3) * PLANE + (i % 3) * 9 * XP_RS] = f.m[i];
    }
}

// ------------------------------------------------------------------------------------------------
// K9 forward: maps -> a2 [S][2704] (post-ReLU conv2 output, torch Flatten order c*169 + y*13 + x), and for
// training p1 [S][8][169] (pooled activations) + amax [S][8][169] (which pixel of the 2x2 window won, 0..3 in
// row-major order; first maximum wins like torch's CPU max_pool2d).
// THREE consecutive images per 512-thread workgroup (round 3): 3 x 169 = 507 of 512 lanes own a pooled cell / pixel (one image per
// 192 threads left 12 % of the lanes idle in a VALU-bound kernel); the three images' 12 planes are one contiguous block of 324 map
// rows in HBM = 18 rows for each of the 18 staging groups of 27 threads.
constexpr int FW_IMG = 3, FW_NT = 512;
constexpr int FW_G = FW_NT / MAPW;                                  // 18 staging groups
constexpr int FW_ROWS = FW_IMG * 4 * MAPW;                          // 324 map rows per workgroup round
constexpr int FW_PER = (FW_ROWS + FW_G - 1) / FW_G;                 // 18 rows per staging thread
constexpr int FW_OWN = FW_IMG * PC;                                 // 507 owning threads

struct FwFetch {
    float m[FW_PER];
    int loc, pc;
};

template <int CIN>
__device__ __forceinline__ void fw_fetch(const CnnIn& in, long long s0, int img, FwFetch& f) {
    const long long left = in.S - s0;                               // images left from s0 on (>= 1)
    const int lim = (int)(left < FW_IMG ? left : FW_IMG) * 4 * MAPC;
    const float* src = in.maps + (size_t)s0 * 4 * MAPC;
    const bool active = threadIdx.x < FW_G * MAPW;
#pragma unroll
    for (int i = 0; i < FW_PER; ++i) {
        const int e = threadIdx.x + i * FW_G * MAPW;
        f.m[i] = (active && e < lim) ? src[e] : 0.0f;
    }
    f.loc = -1; f.pc = -1;
    if (CIN == 6 && img < FW_IMG && s0 + img < in.S) {
        f.loc = (int)in.cells[(size_t)(s0 + img) * in.A + in.agent];
        f.pc = (int)in.pcells[(size_t)(s0 + img) * in.A + in.agent];
    }
}

__device__ __forceinline__ void fw_stage(const FwFetch& f, float* xp) {
    const int g = threadIdx.x / MAPW, c = threadIdx.x - g * MAPW;
    const bool active = g < FW_G;
    int r = g, dst = (g + 1) * XP_RS + c + 1;                           // map row g of plane 0; the planes of all images are equally strided
#pragma unroll
    for (int i = 0; i < FW_PER; ++i) {
        if (active && g + i * FW_G < FW_ROWS) xp[dst] = f.m[i];
        r += FW_G; dst += FW_G * XP_RS;
        if (r >= MAPW) { r -= MAPW; dst += XP_PLANE - MAPW * XP_RS; }  // into the next plane (FW_G < 27: one wrap at most)
    }
}

template <int CIN>
__global__ void __launch_bounds__(FW_NT, 4) rs_cnn_fwd_kernel(CnnIn in, const float* __restrict__ wt, float* __restrict__ a2,
                                                            float* __restrict__ p1g, uint8_t* __restrict__ amax,
                                                            uint16_t* __restrict__ relu_mask) {
    extern __shared__ __align__(16) float smem[];
    float* xp = smem;                               // [FW_IMG][4 dense planes][28][28]
    float* pp = xp + FW_IMG * DP * XP_PLANE;        // [FW_IMG][8][15][15]
    float* stl = pp + FW_IMG * C1 * PP_PLANE;       // [2][9][8] actor: the one-hot channels' weight stamps
    constexpr int CH0 = (CIN == 6) ? 2 : 0;         // the logical channel of dense plane 0 (actor: `others`, convolved as `combined`)
    const cmem_t w1c = as_cmem(wt + WT_W1(CIN) + CH0 * 9 * C1), b1c = as_cmem(wt + WT_B1(CIN));
    const cmem_t w2c = as_cmem(wt + WT_W2(CIN)), b2c = as_cmem(wt + WT_B2(CIN));
    const int tid = threadIdx.x;
    for (int e = tid; e < FW_IMG * (DP * XP_PLANE + C1 * PP_PLANE); e += FW_NT) smem[e] = 0.0f;      // borders stay zero
    if (CIN == 6 && tid < 2 * 9 * C1) stl[tid] = wt[WT_ST(CIN) + tid];
    __syncthreads();
    const int img = tid / PC, cell = tid - img * PC;                 // img == FW_IMG: the five threads that own nothing
    const int py = cell / PW, px = cell - py * PW;
    const float* xpi = xp + (img < FW_IMG ? img : 0) * DP * XP_PLANE;
    float* ppi = pp + (img < FW_IMG ? img : 0) * C1 * PP_PLANE;
    const long long rounds = (in.S + FW_IMG - 1) / FW_IMG;
    FwFetch f;
    if ((long long)blockIdx.x < rounds) fw_fetch<CIN>(in, (long long)blockIdx.x * FW_IMG, img, f);
    CNN_DECL
    for (long long rd = blockIdx.x; rd < rounds; rd += gridDim.x) {
        const long long s = rd * FW_IMG + img;
        const bool own = tid < FW_OWN && s < in.S;
        const int loc = f.loc, pc = f.pc;
        fw_stage(f, xp);
        if (rd + gridDim.x < rounds) fw_fetch<CIN>(in, (rd + gridDim.x) * FW_IMG, img, f);      // next round: in flight during the compute
        CNN_STAMP(0)
        __syncthreads();
        CNN_STAMP(1)
        if (own) {
            // ---- conv1 on the 2x2 block of the cell + bias + ReLU + max-pool.  The 36 weight rows (plane, ky, kx) x 8 output channels stream
            // through the scalar unit four at a time (cnn_wstream); a plane's 4 x 4 input window is read from LDS one block before its
            // first tap is used (double buffered); accumulators as explicit pairs (v_pk_fma_f32)
            v2f accv[4][C1 / 2];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int co = 0; co < C1 / 2; ++co) accv[p][co] = (v2f){0.0f, 0.0f};
            {
                float win[2][4][4];
                const float* xw = xpi + 2 * py * XP_RS + 2 * px;
                cnn_wstream<DP * 9, C1>(
                    w1c,
                    [&](int b) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 4 * b + r;
                            if (row < DP * 9 && row % 9 == 0) {
                                const int ci = row / 9;
#pragma unroll
                                for (int rr = 0; rr < 4; ++rr) {
                                    const float2 a = *reinterpret_cast<const float2*>(&xw[ci * XP_PLANE + rr * XP_RS]);
                                    const float2 c = *reinterpret_cast<const float2*>(&xw[ci * XP_PLANE + rr * XP_RS + 2]);
                                    win[ci & 1][rr][0] = a.x; win[ci & 1][rr][1] = a.y; win[ci & 1][rr][2] = c.x; win[ci & 1][rr][3] = c.y;
                                }
                            }
                        }
                    },
                    [&](int row, const float* w) {
                        const int ci = row / 9, kk = row - ci * 9, ky = kk / 3, kx = kk - ky * 3;
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float v = win[ci & 1][i + ky][j + kx];
#pragma unroll
                                for (int q = 0; q < C1 / 2; ++q)
                                    accv[i * 2 + j][q] = __builtin_elementwise_fma((v2f){w[2 * q], w[2 * q + 1]}, (v2f){v, v}, accv[i * 2 + j][q]);
                            }
                    },
                    [&]() {
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int q = 0; q < C1 / 2; ++q) asm volatile("" : "+v"(accv[p][q]));
                    });
            }
            float acc[4][C1];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int co = 0; co < C1; ++co) acc[p][co] = accv[p][co / 2][co & 1];
            if (CIN == 6) {
                // the one-hot channels: a 1 at input (r, c) adds w[ky][kx] to conv output (r - ky + 1, c - kx + 1); at most four cells
                // of an image own such a pixel
                auto stamp = [&](int at, const float* ws) {
                    const int r = at / MAPW, c = at - r * MAPW;
                    const int dy = r - 2 * py, dx = c - 2 * px;
                    if (dy >= -1 && dy <= 2 && dx >= -1 && dx <= 2) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int ky = dy - i + 1, kx = dx - j + 1;
                                if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) {
                                    const v4f wa = *reinterpret_cast<const v4f*>(ws + (ky * 3 + kx) * C1);
                                    const v4f wb = *reinterpret_cast<const v4f*>(ws + (ky * 3 + kx) * C1 + 4);
#pragma unroll
                                    for (int q = 0; q < 4; ++q) { acc[i * 2 + j][q] += wa[q]; acc[i * 2 + j][4 + q] += wb[q]; }
                                }
                            }
                    }
                };
                if (loc >= 0) stamp(loc, stl + 72);      // -1 = no position recorded yet (fresh maps): contributes nothing, as in K10's gather
                if (pc >= 0) stamp(pc, stl);
            }
            float pbest[C1];
            uint32_t pidx[2] = {0u, 0u};
#pragma unroll
            for (int co = 0; co < C1; ++co) {
                const float bb = b1c[co];
                float best = fmaxf(acc[0][co] + bb, 0.0f);
                int idx = 0;
#pragma unroll
                for (int p = 1; p < 4; ++p) {
                    const float v = fmaxf(acc[p][co] + bb, 0.0f);
                    if (v > best) { best = v; idx = p; }
                }
                ppi[co * PP_PLANE + (py + 1) * PP_RS + px + 1] = best;
                pbest[co] = best;
                pidx[co >> 2] |= (uint32_t)idx << (8 * (co & 3));
            }
            if (p1g) {                  // cell-major rows: 32 B of p1 and 8 B of amax per thread, contiguous across the wave
                v4f* dst = reinterpret_cast<v4f*>(p1g + ((size_t)s * PC + cell) * C1);
                dst[0] = (v4f){pbest[0], pbest[1], pbest[2], pbest[3]};
                dst[1] = (v4f){pbest[4], pbest[5], pbest[6], pbest[7]};
                *reinterpret_cast<uint2*>(amax + ((size_t)s * PC + cell) * C1) = make_uint2(pidx[0], pidx[1]);
            }
        }
        CNN_STAMP(2)
        __syncthreads();
        CNN_STAMP(3)
        if (own) {
            // ---- conv2 + bias + ReLU: 72 weight rows (ci, ky, kx) x 16 output channels, two per block; a block's two P1 values are read
            // from LDS one block ahead
            v2f acc2v[C2 / 2];
#pragma unroll
            for (int co = 0; co < C2 / 2; ++co) acc2v[co] = (v2f){0.0f, 0.0f};
            {
                float vq[2][2];
                const float* ppq = ppi + py * PP_RS + px;
                cnn_wstream<C1 * 9, C2>(
                    w2c,
                    [&](int b) {
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int row = 2 * b + r < C1 * 9 ? 2 * b + r : C1 * 9 - 1;
                            const int ci = row / 9, kk = row - ci * 9, ky = kk / 3, kx = kk - ky * 3;
                            vq[b & 1][r] = ppq[ci * PP_PLANE + ky * PP_RS + kx];
                        }
                    },
                    [&](int row, const float* w) {
                        const float v = vq[(row / 2) & 1][row & 1];
#pragma unroll
                        for (int q = 0; q < C2 / 2; ++q) acc2v[q] = __builtin_elementwise_fma((v2f){w[2 * q], w[2 * q + 1]}, (v2f){v, v}, acc2v[q]);
                    },
                    [&]() {
#pragma unroll
                        for (int q = 0; q < C2 / 2; ++q) asm volatile("" : "+v"(acc2v[q]));
                    });
            }
            float acc2[C2];
#pragma unroll
            for (int co = 0; co < C2; ++co) acc2[co] = acc2v[co / 2][co & 1];
            uint32_t live = 0u;
#pragma unroll
            for (int co = 0; co < C2; ++co) {
                const float o = fmaxf(acc2[co] + b2c[co], 0.0f);
                a2[(size_t)s * FLAT + co * PC + cell] = o;
                live |= (o > 0.0f ? 1u : 0u) << co;
            }
            if (relu_mask) relu_mask[(size_t)s * PC + cell] = (uint16_t)live;       // what backward needs of a2: its sign
        }
        CNN_STAMP(4)
        __syncthreads();
        CNN_STAMP(5)
    }
    CNN_FLUSH(0)
}

// ------------------------------------------------------------------------------------------------
// K10 backward: dL/d(a2) -> per-workgroup partial sums of dW1, db1, dW2, db2 (slab row layout:
// [dW1 8*CIN*9 | db1 8 | dW2 16*72 | db2 16], torch weight order).
template <int CIN>
__global__ void __launch_bounds__(CNN_NT_BWD, CNN_BWD_WAVES) rs_cnn_bwd_kernel(CnnIn in, const float* __restrict__ wt, const float* __restrict__ da2,
                                                            const uint16_t* __restrict__ relu_mask, const float* __restrict__ p1g,
                                                            const uint8_t* __restrict__ amax, float* __restrict__ slab) {
    extern __shared__ __align__(16) float smem[];
    constexpr int K1 = CIN * 9;
    float* xp = smem;                               // [4 dense planes][28][28] (plane stride XP_PLANE_B)
    float* pp = xp + DP * XP_PLANE_B;               // [8][15][15]   padded P1
    float* dzp = pp + C1 * PPB_PLANE;                // [16][15][15]  padded dZ2
    float* gbuf = dzp + C2 * PPB_PLANE;              // [8][169]      dL/d(pooled) after the ReLU gate: output channels 0..12 of conv2
    float* gbufb = gbuf + C1 * PC;                  // [8][169]      ... and 13..15 (the helper wave's share)
    uint8_t* ambuf = reinterpret_cast<uint8_t*>(gbufb + C1 * PC);     // [169][8]
    const cmem_t w2b = as_cmem(wt + WT_W2B(CIN));   // [(co,ky,kx)][8 ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave id as a scalar
    constexpr int NWAVE = CNN_NT_BWD / 64;
    for (int e = tid; e < DP * XP_PLANE_B + C1 * PPB_PLANE + C2 * PPB_PLANE; e += CNN_NT_BWD) smem[e] = 0.0f;
    __syncthreads();
    const int py = tid / PW, px = tid - py * PW;
    const bool own = tid < PC;
    const int dp1_cell = tid < 192 ? (int)BWD_SLOT_PIXEL[tid] : 255;      // the pixel this thread takes in the dP1 phase (bank-conflict free)
    // dW2 (+ db2 in column 72) accumulators: 5 column tiles of the [16 co] x [80] product, this wave's share of the pixels
    v4f accw[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) accw[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    // dW1 of the DENSE planes: thread = (output channel co1, half of the 4 planes) x one of 16 cell groups: per cell three index reads
    // (gradient, arg-max, cell offset) feed CG*9 window reads + FMAs (+ db1 on the first-half threads).  The actor's one-hot
    // channels take their weight gradients from 9-element gathers instead (gst below).
    constexpr int CG = DP / 2, NCOMBO = C1 * 2, NGRP = CNN_NT_BWD / NCOMBO, AW = CG * 9 + 1;     // 16 combos x (threads / 16) groups
    float gst = 0.0f;                               // actor: thread (which, co, tap) of the 2 x 8 x 9 stamp gradients, tid < 144
    int st_pack;                                    // which | co << 1 | (2 - ky) << 4 | (2 - kx) << 6: one register across the image loop
    {
        const int st_which = tid / (C1 * 9), st_co = (tid - st_which * (C1 * 9)) / 9, st_kk = tid % 9;
        st_pack = (st_which & 1) | (st_co << 1) | ((2 - st_kk / 3) << 4) | ((2 - st_kk % 3) << 6);
    }
    float aw1[AW];
#pragma unroll
    for (int k = 0; k < AW; ++k) aw1[k] = 0.0f;
    const int combo = tid % NCOMBO, grp = tid / NCOMBO;
    const int co1 = combo & 7, c0g = combo >> 3;
    // Cells are walked column-major (walk index e -> cell (e % 13, e / 13)), so the 4 cells a wave gathers at once share cx and
    // differ in cy: their window offsets 56*cy + {0,1,28,29} fall into 16 distinct LDS banks (bank mod 4 in {0,1}), and the second
    // channel half (2 planes = 1582 floats = +14 banks: bank mod 4 in {2,3}) takes the other 16 -> the 9-tap gathers are
    // conflict-free (row-major neighbours collided 2-3 way).
    // MFMA operand coordinates of this lane
    const int mrow = lane & 15, mk = lane >> 4;
    // float indices (into smem) of this lane's six operands for its first pixel p0 = 4 wave + mk (< 16): A = dZ2[mrow][p], B column
    // n = 16 tile + mrow = tap (ci, ky, kx) of the padded P1 planes at p.  Tile 4: columns 64..71 are taps, column 72 is the ones column
    // (db2) and 73..79 are zero -- those lanes read a cell holding 1.0 / a border cell (0.0) and do not advance (m4 = 0)
    const int p0 = 4 * wave + mk, wy0 = p0 >= PW ? 1 : 0, wx0 = p0 - wy0 * PW;
    const int off0 = wy0 * PPB_RS + wx0;
    const int ia0 = (int)(dzp - smem) + mrow * PPB_PLANE + PPB_RS + 1 + off0;
    int ib0[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int n = 16 * t + mrow, nn = (n < 72) ? n : 0;
        ib0[t] = (int)(pp - smem) + (nn / 9) * PPB_PLANE + ((nn % 9) / 3) * PPB_RS + nn % 3 + off0;
    }
    const int m4 = (mrow < 8) ? -1 : 0;
    float* onec = reinterpret_cast<float*>(ambuf + C1 * PC);          // [1] = 1.0f
    if (mrow == 8) ib0[4] = (int)(onec - smem);
    if (mrow > 8) ib0[4] = (int)(pp - smem);                          // a border cell of the padded plane: never written, 0.0
    if (tid == 0) *onec = 1.0f;
    CnnFetch<CNN_NT_BWD> f;
    v4f fp1a = (v4f){0.f, 0.f, 0.f, 0.f}, fp1b = fp1a;
    float fda2[C2];
    uint2 fam = make_uint2(0u, 0u);
    uint32_t fmask = 0u;
    auto fetch_acts = [&](long long s) {
        if (own) {
            const v4f* src = reinterpret_cast<const v4f*>(p1g + ((size_t)s * PC + tid) * C1);
            fp1a = src[0]; fp1b = src[1];
            fam = *reinterpret_cast<const uint2*>(amax + ((size_t)s * PC + tid) * C1);
            fmask = relu_mask[(size_t)s * PC + tid];
#pragma unroll
            for (int co = 0; co < C2; ++co) fda2[co] = da2[(size_t)s * FLAT + co * PC + tid];
        }
    };
    if ((long long)blockIdx.x < in.S) { cnn_fetch<CIN, CNN_NT_BWD>(in, blockIdx.x, f); if (CNN_BWD_PREFETCH) fetch_acts(blockIdx.x); }
    CNN_DECL
    for (long long s = blockIdx.x; s < in.S; s += gridDim.x) {
        const int loc = f.loc, pc = f.pc;
        if (!CNN_BWD_PREFETCH) fetch_acts(s);
        cnn_stage<XP_PLANE_B, CNN_NT_BWD>(f, xp);
        if (own) {
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                pp[co * PPB_PLANE + (py + 1) * PPB_RS + px + 1] = fp1a[co];
                pp[(4 + co) * PPB_PLANE + (py + 1) * PPB_RS + px + 1] = fp1b[co];
            }
            {   // arg-max codes 0..3 (row << 1 | column of the 2 x 2 pool window) -> the window's byte offset (row * 28 + column) * 4, per byte
                const uint32_t e0 = ((fam.x >> 1) & 0x01010101u) * (4u * XP_RS) + (fam.x & 0x01010101u) * 4u;
                const uint32_t e1 = ((fam.y >> 1) & 0x01010101u) * (4u * XP_RS) + (fam.y & 0x01010101u) * 4u;
                *reinterpret_cast<uint2*>(ambuf + tid * C1) = make_uint2(e0, e1);
            }
#pragma unroll
            for (int co = 0; co < C2; ++co)
                dzp[co * PPB_PLANE + (py + 1) * PPB_RS + px + 1] = ((fmask >> co) & 1u) ? fda2[co] : 0.0f;     // ReLU gate
        }
        if (s + gridDim.x < in.S) { cnn_fetch<CIN, CNN_NT_BWD>(in, s + gridDim.x, f); if (CNN_BWD_PREFETCH) fetch_acts(s + gridDim.x); }   // in flight during the compute
        CNN_STAMP(0)
        __syncthreads();
        CNN_STAMP(1)
        // ---- dW2[co][n] += sum_px dZ2[co][px] * P1patch[px][n]   (matrix cores; k-steps interleaved over the 3 waves)
#if !(defined(CNN_ABL) && CNN_ABL == 3)
        {
            // operands of step st + NWAVE are read from LDS while the five MFMAs of step st run (one exposed LDS round trip per wave
            // and image instead of two per step).  The lane's pixel 4 st + mk advances by 16 = one row + 3 columns per step, so the six
            // operand addresses are carried and bumped by one per-lane delta (row wrap: + another row - 13 columns) -- recomputing
            // y = pixel / 13, x, and six addresses from them cost 43 vector instructions per k-step, a third of the kernel's (round 4).
            typedef const __attribute__((address_space(3))) char* lptr_t;      // LDS byte addresses, bumped in bytes
            const lptr_t lbase = (lptr_t)smem;
            lptr_t qa = lbase + 4 * ia0, qb[5];
            int wx = wx0, pix = 4 * wave + mk;
#pragma unroll
            for (int t = 0; t < 5; ++t) qb[t] = lbase + 4 * ib0[t];
            auto operands = [&](float& av, float (&bv)[5]) {
                const float a = *(const __attribute__((address_space(3))) float*)qa;
                av = pix < PC ? a : 0.0f;
#pragma unroll
                for (int t = 0; t < 5; ++t) bv[t] = *(const __attribute__((address_space(3))) float*)qb[t];
                wx += 3;
                const bool wrap = wx >= PW;
                const int d = wrap ? 4 * (2 * PPB_RS + 3 - PW) : 4 * (PPB_RS + 3);
                wx = wrap ? wx - PW : wx;
                pix += 16;
                qa += d;
#pragma unroll
                for (int t = 0; t < 4; ++t) qb[t] += d;
                qb[4] += d & m4;
            };
            // two steps per trip (registers ping-pong, no moves); waves 0-2 take 11 steps, wave 3 ten
            float av0, bv0[5], av1 = 0.0f, bv1[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            operands(av0, bv0);
            const int nst = (43 - wave + NWAVE - 1) / NWAVE;
#pragma unroll 1
            for (int j = 0; j < nst; j += 2) {
                if (j + 1 < nst) operands(av1, bv1);
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);      // the next step's six DS reads go out first ...
#pragma unroll
                for (int t = 0; t < 5; ++t) accw[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv0[t], accw[t], 0, 0, 0);
                if (j + 1 < nst) {
                    if (j + 2 < nst) operands(av0, bv0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                    for (int t = 0; t < 5; ++t) accw[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv1[t], accw[t], 0, 0, 0);
                }
            }
        }
#endif
        CNN_STAMP(2)
        // ---- dP1 = conv2^T(dZ2), gated by the pool's ReLU (P1 > 0).  The sum over the 16 output channels is split between the pixel
        // threads (waves 0-2: co 0..12, 117 taps each) and the helper wave (co 13..15 for up to three pixels per lane, 81 taps):
        // with all 144 taps on the 169 pixel threads the fourth wave sat idle for the longest phase of the kernel (28 % of it).  The
        // two partial sums land in gbuf / gbufb and are added where they are read (dW1 gathers below).
#if defined(CNN_ABL) && CNN_ABL == 2
        if (own) for (int ci = 0; ci < C1; ++ci) { gbuf[ci * PC + tid] = dzp[ci * PPB_PLANE + (py + 1) * PPB_RS + px + 1]; gbufb[ci * PC + tid] = 0.0f; }
        if (false) {
#else
        {
#endif
            // rows (co, ky, kx) of 8 weights (ci) stream through the scalar unit, four rows per block; the four dZ2 values a block
            // multiplies them with are read from LDS one block ahead, next to the block's weights
            auto dp1 = [&](int qy, int qx, int cell_, auto co_lo_c, auto co_hi_c, float* dst) {
                constexpr int CO_LO = decltype(co_lo_c)::value, CO_HI = decltype(co_hi_c)::value, ROWS = (CO_HI - CO_LO) * 9, NBLK = (ROWS + 3) / 4;
                v2f g2[C1 / 2];                              // explicit pairs: v_pk_fma_f32 (the pins below would otherwise split them)
#pragma unroll
                for (int ci = 0; ci < C1 / 2; ++ci) g2[ci] = (v2f){0.0f, 0.0f};
                float vq[2][4];
                const float* dzq = dzp + CO_LO * PPB_PLANE + (qy + 2) * PPB_RS + qx + 2;
                auto value_at = [&](int row) -> float {            // row = (co - CO_LO) * 9 + ky * 3 + kx, compile-time after unrolling
                    const int co = row / 9, kk = row - co * 9, ky = kk / 3, kx = kk - ky * 3;
                    return dzq[co * PPB_PLANE - ky * PPB_RS - kx];
                };
                cnn_wstream<ROWS, C1>(
                    w2b + CO_LO * 9 * C1,
                    [&](int b) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) vq[b & 1][r] = value_at(4 * b + r < ROWS ? 4 * b + r : ROWS - 1);
                    },
                    [&](int row, const float* w) {
                        const float v = vq[(row / 4) & 1][row & 3];
#pragma unroll
                        for (int q = 0; q < C1 / 2; ++q) g2[q] = __builtin_elementwise_fma((v2f){w[2 * q], w[2 * q + 1]}, (v2f){v, v}, g2[q]);
                    },
                    [&]() {
#pragma unroll
                        for (int q = 0; q < C1 / 2; ++q) asm volatile("" : "+v"(g2[q]));
                    });
                (void)NBLK;
                float g[C1];
#pragma unroll
                for (int ci = 0; ci < C1; ++ci) g[ci] = g2[ci / 2][ci & 1];
                float gate[C1];                         // all eight reads before the first write: the compiler cannot tell dst from pp
#pragma unroll
                for (int ci = 0; ci < C1; ++ci) gate[ci] = pp[ci * PPB_PLANE + (qy + 1) * PPB_RS + qx + 1];
#pragma unroll
                for (int ci = 0; ci < C1; ++ci) dst[ci * PC + cell_] = (gate[ci] > 0.0f) ? g[ci] : 0.0f;
            };
            if (wave < 3) {
                if (dp1_cell < PC) {
                    const int qy = dp1_cell / PW;
                    dp1(qy, dp1_cell - qy * PW, dp1_cell, std::integral_constant<int, 0>{}, std::integral_constant<int, DP1_SPLIT>{}, gbuf);
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < 3; ++k) {
                    const int cell_ = BWD_SLOT_PIXEL[lane + 64 * k];
                    if (cell_ < PC) {
                        const int qy = cell_ / PW;
                        dp1(qy, cell_ - qy * PW, cell_, std::integral_constant<int, DP1_SPLIT>{}, std::integral_constant<int, C2>{}, gbufb);
                    }
                }
            }
        }
        CNN_STAMP(3)
        __syncthreads();
        CNN_STAMP(4)
        // ---- dW1[co1][c0][:] += g[co1][cell] * x[c0] window at the cell's arg-max pixel (no branch on g == 0: a dead
        // cell adds zeros; the phase is instruction-issue bound, so the index work is shared by CG channels)
#if !(defined(CNN_ABL) && CNN_ABL == 1)
        {
            // One ROW of pooled cells per group (cy = grp < 13; groups 13..15 idle): consecutive cells are one gradient float, eight
            // arg-max bytes and two input columns apart, so the walk is three pointer bumps, and the arg-max is stored as the window's byte offset (ambuf, encoded at staging).  Per cell: 21 LDS reads, 9 packed FMAs
            // and ~8 other vector instructions.  (Rolled loop with three pointer bumps: fully unrolled, hipcc hoisted the loads of several cells and spilled 284 B.  Round 3 walked the cells column-major with stride 16 and rebuilt cell index, row,
            // column, window offset and the arg-max decode per cell: ~30 vector instructions around the same 9 FMAs, a third of the
            // kernel's vector instructions -- profiles/r04_cnn_phase_cycles.txt.)  Software-pipelined by hand as before: the next
            // cell's gradient / arg-max reads are in flight while this cell's 18 window values arrive in one round trip.
            if (grp < PW) {
                typedef const __attribute__((address_space(3))) char* lptr_t;
                typedef const __attribute__((address_space(3))) float* lfp_t;
                typedef const __attribute__((address_space(3))) uint8_t* lbp_t;
                lfp_t gq = (lfp_t)(gbuf + co1 * PC + grp * PW);             // gbufb lies C1 * PC floats behind gbuf: an immediate offset
                lbp_t aq = (lbp_t)(ambuf + grp * PW * C1 + co1);
                lptr_t wq = (lptr_t)(xp + (c0g * CG) * XP_PLANE_B + 2 * grp * XP_RS);
                float gv_n = gq[0] + gq[C1 * PC];
                int am_n = aq[0];
#pragma unroll 1
                for (int cx = 0; cx < PW; ++cx) {
                    const float gv = gv_n;
                    const lfp_t base = (lfp_t)(wq + am_n);
                    float wv[CG * 9];
#pragma unroll
                    for (int j = 0; j < CG; ++j)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) wv[j * 9 + ky * 3 + kx] = base[j * XP_PLANE_B + ky * XP_RS + kx];
                    gq += 1; aq += C1; wq += 8;
                    if (cx + 1 < PW) { gv_n = gq[0] + gq[C1 * PC]; am_n = aq[0]; }
                    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);     // all DS reads of the round first ...
                    __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);     // ... then the VALU work
#pragma unroll
                    for (int k = 0; k < CG * 9; ++k) aw1[k] = __builtin_fmaf(gv, wv[k], aw1[k]);
                    aw1[AW - 1] += gv;
                }
            }
        }
#endif
        CNN_STAMP(5)
        if (CIN == 6 && tid < 2 * C1 * 9) {
            // one-hot input at (r, c): tap (ky, kx) of conv1 meets it at output pixel (r - ky + 1, c - kx + 1); that pixel carries
            // gradient iff it is its pool window's arg-max for channel co
            const int st_co = (st_pack >> 1) & 7;
            const int cell = (st_pack & 1) ? loc : pc;
            if (cell >= 0) {
                const int r = cell / MAPW, c = cell - r * MAPW;
                const int y = r + ((st_pack >> 4) & 3) - 1, x = c + ((st_pack >> 6) & 3) - 1;
                if (y >= 0 && y < 2 * PW && x >= 0 && x < 2 * PW) {
                    const int pcell = (y >> 1) * PW + (x >> 1);
                    if (ambuf[pcell * C1 + st_co] == (uint8_t)((y & 1) * (4 * XP_RS) + (x & 1) * 4)) gst += gbuf[st_co * PC + pcell] + gbufb[st_co * PC + pcell];
                }
            }
        }
        CNN_STAMP(6)
        __syncthreads();
        CNN_STAMP(7)
    }
    CNN_FLUSH(1)
    // ---- workgroup reduction (fixed order) -> slab row
    __syncthreads();
    float* red = smem;                               // aliases the image buffers: [NGRP][NCOMBO][AW], [NWAVE][16][80], [2][8][9]
    float* red2 = red + CNN_NT_BWD * AW;
    float* gred = red2 + NWAVE * 16 * 80;
    if (tid < 2 * C1 * 9) gred[tid] = gst;
#pragma unroll
    for (int k = 0; k < AW; ++k) red[tid * AW + k] = aw1[k];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) red2[(wave * 16 + 4 * mk + j) * 80 + 16 * t + mrow] = accw[t][j];
    __syncthreads();
    constexpr int ROW = C1 * K1 + C1 + C2 * 72 + C2;
    float* out = slab + (size_t)blockIdx.x * ROW;
    for (int e = tid; e < C1 * K1 + C1; e += CNN_NT_BWD) {
        int o1, cb, slot;
        float sum = 0.0f;
        if (e < C1 * K1) {
            o1 = e / K1;
            const int k = e - o1 * K1, i0 = k / 9, kk = k - i0 * 9;      // i0: the LOGICAL input channel (torch weight order)
            constexpr int CH0 = (CIN == 6) ? 2 : 0;
            if (i0 < CH0) {                                               // prediction (0) / location (1): the stamp gathers
                out[e] = gred[(i0 * C1 + o1) * 9 + kk];
                continue;
            }
            const int d = i0 - CH0;                                       // dense plane
            cb = o1 + 8 * (d / CG);
            slot = (d % CG) * 9 + kk;
            if (CIN == 6 && d == 0) sum = -gred[(C1 + o1) * 9 + kk];      // others = combined - location
        } else {
            o1 = e - C1 * K1; cb = o1; slot = AW - 1;           // db1 lives on the first-half threads
        }
#pragma unroll
        for (int gi = 0; gi < NGRP; ++gi) sum += red[(gi * NCOMBO + cb) * AW + slot];
        out[e] = sum;
    }
    for (int e = tid; e < C2 * 73; e += CNN_NT_BWD) {
        const int co = e / 73, n = e - co * 73;
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) sum += red2[(w * 16 + co) * 80 + n];
        if (n < 72) out[C1 * K1 + C1 + co * 72 + n] = sum; else out[C1 * K1 + C1 + C2 * 72 + co] = sum;
    }
}

inline size_t fwd_lds(int) { return sizeof(float) * (size_t)(FW_IMG * (DP * XP_PLANE + C1 * PP_PLANE) + 2 * 9 * C1); }
inline size_t bwd_lds(int) {
    size_t img = (size_t)(DP * XP_PLANE_B + C1 * PPB_PLANE + C2 * PPB_PLANE + 2 * C1 * PC) * 4 + C1 * PC + 16;     // + the ones cell
    size_t red = (size_t)(CNN_NT_BWD * ((DP / 2) * 9 + 1) + (CNN_NT_BWD / 64) * 16 * 80 + 2 * C1 * 9) * 4;
    return ((img > red ? img : red) + 15) & ~(size_t)15;
}
// persistent grid: exactly as many workgroups as are resident at once (occupancy query x CU count), so no second,
// half-empty round of workgroups trails the first; each workgroup strides over the batch
template <typename K>
inline int cnn_grid(K kernel, int threads, size_t lds, long long S) {
    int per_cu = 0, dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const long long cap = (long long)cus * per_cu;
    return (int)(S < cap ? S : cap);
}

}  // namespace

extern "C" {

#ifdef RS_CNN_STAMPS
int rs_debug_cnn_stamps(unsigned long long* out, int reset) {
    unsigned long long z[2 * 8 * 8] = {0};
    if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(rs_cnn_cyc), z, sizeof(z)) == hipSuccess ? 0 : 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rs_cnn_cyc), sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif

int32_t rs_cnn_trunk_slab_row(int32_t in_channels) { return C1 * in_channels * 9 + C1 + C2 * 72 + C2; }
int32_t rs_cnn_trunk_slab_rows(int64_t num_samples, int32_t in_channels) {
    return in_channels == 6 ? cnn_grid(rs_cnn_bwd_kernel<6>, CNN_NT_BWD, bwd_lds(6), num_samples) : cnn_grid(rs_cnn_bwd_kernel<4>, CNN_NT_BWD, bwd_lds(4), num_samples);
}
int32_t rs_cnn_trunk_scratch_floats(int32_t in_channels) { return WT_TOTAL(in_channels); }

int rs_cnn_trunk_forward(const float* maps, const int64_t* cells, const int64_t* pcells, int32_t num_agents, int32_t agent,
                         int64_t num_samples, const float* w1, const float* b1, const float* w2, const float* b2, float* a2,
                         float* p1, uint8_t* amax, uint16_t* relu_mask, float* wscratch, rs_stream_t stream) {
    if (!maps || !w1 || !b1 || !w2 || !b2 || !a2 || !wscratch || num_samples < 0) return RS_ERR_INVALID_ARG;
    if ((p1 == nullptr) != (amax == nullptr) || (p1 == nullptr) != (relu_mask == nullptr)) return RS_ERR_INVALID_ARG;
    if (agent >= 0 && (!cells || !pcells || agent >= num_agents)) return RS_ERR_INVALID_ARG;
    if (num_samples == 0) return RS_OK;
    CnnIn in{maps, cells, pcells, num_agents, agent, (long long)num_samples};
    hipStream_t s = (hipStream_t)stream;
    const int cin = agent >= 0 ? 6 : 4;
    hipLaunchKernelGGL(rs_cnn_prep_kernel, dim3(1), dim3(256), 0, s, cin, w1, b1, w2, b2, wscratch);
    if (agent >= 0) hipLaunchKernelGGL(rs_cnn_fwd_kernel<6>, dim3(cnn_grid(rs_cnn_fwd_kernel<6>, FW_NT, fwd_lds(6), (num_samples + FW_IMG - 1) / FW_IMG)), dim3(FW_NT),
                                       fwd_lds(6), s, in, wscratch, a2, p1, amax, relu_mask);
    else hipLaunchKernelGGL(rs_cnn_fwd_kernel<4>, dim3(cnn_grid(rs_cnn_fwd_kernel<4>, FW_NT, fwd_lds(4), (num_samples + FW_IMG - 1) / FW_IMG)), dim3(FW_NT),
                            fwd_lds(4), s, in, wscratch, a2, p1, amax, relu_mask);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

/* The forward trunk for a collector's lock-steps: the weights are re-arranged ONCE per epoch (rs_cnn_trunk_prepare -> wscratch) and every
 * select_action round is one launch (rs_cnn_trunk_infer; no activations for a backward pass are kept). */
int rs_cnn_trunk_prepare(int32_t in_channels, const float* w1, const float* b1, const float* w2, const float* b2, float* wscratch,
                         rs_stream_t stream) {
    if ((in_channels != 6 && in_channels != 4) || !w1 || !b1 || !w2 || !b2 || !wscratch) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_cnn_prep_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in_channels, w1, b1, w2, b2, wscratch);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_cnn_trunk_infer(const float* maps, const int64_t* cells, const int64_t* pcells, int32_t num_agents, int32_t agent, int64_t num_samples,
                       const float* wscratch, float* a2, rs_stream_t stream) {
    if (!maps || !wscratch || !a2 || num_samples < 0) return RS_ERR_INVALID_ARG;
    if (agent >= 0 && (!cells || !pcells || agent >= num_agents)) return RS_ERR_INVALID_ARG;
    if (num_samples == 0) return RS_OK;
    CnnIn in{maps, cells, pcells, num_agents, agent, (long long)num_samples};
    hipStream_t s = (hipStream_t)stream;
    const long long rounds = (num_samples + FW_IMG - 1) / FW_IMG;
    if (agent >= 0) hipLaunchKernelGGL(rs_cnn_fwd_kernel<6>, dim3(cnn_grid(rs_cnn_fwd_kernel<6>, FW_NT, fwd_lds(6), rounds)), dim3(FW_NT), fwd_lds(6), s,
                                       in, wscratch, a2, (float*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr);
    else hipLaunchKernelGGL(rs_cnn_fwd_kernel<4>, dim3(cnn_grid(rs_cnn_fwd_kernel<4>, FW_NT, fwd_lds(4), rounds)), dim3(FW_NT), fwd_lds(4), s, in,
                            wscratch, a2, (float*)nullptr, (uint8_t*)nullptr, (uint16_t*)nullptr);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_cnn_trunk_backward(const float* maps, const int64_t* cells, const int64_t* pcells, int32_t num_agents, int32_t agent,
                          int64_t num_samples, const float* w2, const float* da2, const uint16_t* relu_mask, const float* p1,
                          const uint8_t* amax, float* slab, float* wscratch, rs_stream_t stream) {
    if (!maps || !w2 || !da2 || !relu_mask || !p1 || !amax || !slab || !wscratch || num_samples <= 0) return RS_ERR_INVALID_ARG;
    if (agent >= 0 && (!cells || !pcells || agent >= num_agents)) return RS_ERR_INVALID_ARG;
    CnnIn in{maps, cells, pcells, num_agents, agent, (long long)num_samples};
    hipStream_t s = (hipStream_t)stream;
    const int cin = agent >= 0 ? 6 : 4;
    hipLaunchKernelGGL(rs_cnn_prep_kernel, dim3(1), dim3(256), 0, s, cin, (const float*)nullptr, (const float*)nullptr, w2,
                       (const float*)nullptr, wscratch);
    if (agent >= 0) hipLaunchKernelGGL(rs_cnn_bwd_kernel<6>, dim3(cnn_grid(rs_cnn_bwd_kernel<6>, CNN_NT_BWD, bwd_lds(6), num_samples)), dim3(CNN_NT_BWD),
                                       bwd_lds(6), s, in, wscratch, da2, relu_mask, p1, amax, slab);
    else hipLaunchKernelGGL(rs_cnn_bwd_kernel<4>, dim3(cnn_grid(rs_cnn_bwd_kernel<4>, CNN_NT_BWD, bwd_lds(4), num_samples)), dim3(CNN_NT_BWD),
                            bwd_lds(4), s, in, wscratch, da2, relu_mask, p1, amax, slab);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
