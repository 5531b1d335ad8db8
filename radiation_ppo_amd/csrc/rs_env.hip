// rs_env.hip -- env kernels (K1 step, K2 reset, K2a shared-geometry resample, K4 GAE) and the C ABI
// declared in include/radsearch.h.  gfx950 only; one environment per wavefront lane; 64-thread
// workgroups (one wave) so that even the 4096-env configuration spreads over 64 CUs and all eight
// XCDs; the blockIdx -> env mapping is identical in every kernel, so an env's state stays in the
// L2 of the XCD its block lands on from launch to launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>

#include "../../include/radsearch.h"
#include "rs_device.hpp"
#include "rs_handle.hpp"

// ------------------------------------------------------------------------------------------------
// K1: RadSearch.step for N envs (rad_search_env.py:443-728)
template <bool HAS_OBS>
__global__ void __launch_bounds__(64, (HAS_OBS ? 4 : 1)) rs_step_kernel(RsParams P, const int8_t* __restrict__ actions, float* obs, RsOut O) {
    extern __shared__ __align__(16) unsigned char smem[];
    int* lds_geo = reinterpret_cast<int*>(smem);
    float* tile = reinterpret_cast<float*>(smem + (HAS_OBS ? RS_MAX_VERT * RS_WAVE * 4 : 0));
    int* flags = reinterpret_cast<int*>(tile + RS_WAVE * rs_tile_stride(P.A));
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = n < P.N;
    RsGeo g{lds_geo, 0, 0, 0};
    if (HAS_OBS) rs_load_geo(P, n, active, lds_geo, g);
    flags[lane] = active ? 1 : 0;
    __syncthreads();
    if (active) {
        RsOut o = O;
        o.obs_row = tile + lane * rs_tile_stride(P.A);
        const int8_t* arow = actions + (size_t)n * P.A;
        uint32_t bad = 0;
        bool single_int = false;                   // 16 + a: RadSearch.step(int) for several agents, no collision rule (:676-690)
        for (int a = 0; a < P.A; ++a) single_int |= arow[a] >= 16;
        auto act_of = [&](int a) -> int {
            int v = arow[a];
            if (v >= 16) v -= 16;
            if (v == -1) return RS_IDLE;
            if (v < 0 || v > RS_ACT_NONE) { bad = RS_ENVERR_BAD_ACTION; return RS_IDLE; }
            return v;                               // 0..8, or RS_ACT_NONE = step(None) (:528-567)
        };
        rs_env_step_lane<HAS_OBS>(P, g, n, act_of, o, single_int);
        if (bad) P.err[n] |= bad;
    }
    __syncthreads();
    rs_copy_out(P, obs, tile, flags, blockIdx.x * blockDim.x);
}

// K1 with obstacles: 16 envs per wave, four lanes per env (lane = slot + 16*cj).  The step of an env behind an obstacle is a
// long serial chain (up to 28 rectangle vertices x 7 rectangles of exact segment tests, 8 probe directions); with one env
// per lane a wave waits for its slowest lane while 8192 envs are only 128 waves on 256 CUs.  Here the four lanes split
// those loops (rs_env_step_lane<true, 4>), and 8192 envs are 512 waves.  Bit-identical to the one-lane form.
__global__ void __launch_bounds__(64, 2) rs_step4_kernel(RsParams P, const int8_t* __restrict__ actions, float* obs, RsOut O) {
    extern __shared__ __align__(16) unsigned char smem[];
    int* lds_geo = reinterpret_cast<int*>(smem);
    float* tile = reinterpret_cast<float*>(smem + RS_MAX_VERT * RS_WAVE * 4);
    const int lane = threadIdx.x & 63, slot = lane & 15, cj = lane >> 4;
    const int n = blockIdx.x * 16 + slot;
    const bool active = n < P.N;
    const int gi = active ? n / P.group : 0;
    int no = active ? P.num_obs[gi] : 0;
    for (int w = cj; w < 4 * no; w += 4) lds_geo[w * RS_WAVE + slot] = P.rect[(size_t)w * P.G + gi];
    if (P.uniform_nobs > 0 && active) no = P.uniform_nobs;
    RsGeo g{lds_geo, RS_WAVE, slot, no};
    __syncthreads();
    const int S = rs_tile_stride(P.A);
    if (active) {
        RsOut o = O;
        o.obs_row = tile + slot * S;
        const int8_t* arow = actions + (size_t)n * P.A;
        uint32_t bad = 0;
        bool single_int = false;
        for (int a = 0; a < P.A; ++a) single_int |= arow[a] >= 16;
        auto act_of = [&](int a) -> int {
            int v = arow[a];
            if (v >= 16) v -= 16;
            if (v == -1) return RS_IDLE;
            if (v < 0 || v > RS_ACT_NONE) { bad = RS_ENVERR_BAD_ACTION; return RS_IDLE; }
            return v;
        };
        rs_env_step_lane<true, 4>(P, g, n, act_of, o, single_int, cj);
        if (bad && cj == 0) P.err[n] |= bad;
    }
    __syncthreads();
    if (obs) {
        const int row = P.A * RS_OBS_DIM;
        const int rows = min(16, P.N - blockIdx.x * 16);
        float* dst = obs + (size_t)blockIdx.x * 16 * row;
        for (int i = lane; i < rows * row; i += RS_WAVE) {
            const int l = i / row, k = i - l * row;
            dst[i] = tile[l * S + k];
        }
    }
}

// K2a: obstacle layouts shared by a group of envs (geom_group_size > 1): one lane per group.
__global__ void __launch_bounds__(64) rs_geom_kernel(RsParams P, const uint8_t* __restrict__ mask) {
    extern __shared__ __align__(16) unsigned char smem[];
    int* lds_geo = reinterpret_cast<int*>(smem);
    const int lane = threadIdx.x & 63;
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= P.G) return;
    const int n0 = gi * P.group;                       // the group's first env decides
    if (!P.epoch_end[n0] || (mask && !mask[n0])) return;
    RsDrawSeq seq{P.seed, P.env_id_base + (uint32_t)n0, P.geom_epoch[gi], RS_STREAM_GEOM, 0u};
    int num;
    do num = rs_create_obs(P, seq, lds_geo, RS_WAVE, lane);      // redrawn until world.is_valid (:788) holds
    while (!rs_layout_valid(lds_geo, RS_WAVE, lane, num));
    P.num_obs[gi] = num;
    for (int w = 0; w < 4 * num; ++w) P.rect[(size_t)w * P.G + gi] = lds_geo[w * RS_WAVE + lane];
    P.geom_epoch[gi] += 1;
}

// K2: RadSearch.reset for the masked envs (rad_search_env.py:730-797)
template <bool HAS_OBS>
__global__ void __launch_bounds__(64) rs_reset_kernel(RsParams P, const uint8_t* __restrict__ mask, float* obs, RsOut O) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool has_obs = HAS_OBS;
    int* lds_geo = reinterpret_cast<int*>(smem);
    uint32_t* lds_adj = reinterpret_cast<uint32_t*>(smem + RS_MAX_VERT * RS_WAVE * 4);
    double* lds_d = reinterpret_cast<double*>(smem + 2 * RS_MAX_VERT * RS_WAVE * 4);
    float* tile = reinterpret_cast<float*>(smem + (has_obs ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0));
    int* flags = reinterpret_cast<int*>(tile + RS_WAVE * rs_tile_stride(P.A));
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = P.N, A = P.A;
    const bool active = n < N && (mask == nullptr || mask[n] != 0);
    RsGeo g{lds_geo, 0, 0, 0};
    if (HAS_OBS) rs_load_geo(P, n, active && P.group > 1, lds_geo, g);   // group > 1: layout resampled by rs_geom_kernel
    flags[lane] = active ? 1 : 0;
    __syncthreads();
    if (active) {
        rs_env_reset_lane<HAS_OBS>(P, g, n, lds_geo, lds_adj, lds_d, tile + lane * rs_tile_stride(A), O);
    }
    __syncthreads();
    rs_copy_out(P, obs, tile, flags, blockIdx.x * blockDim.x);
}

// K2 with obstacles, four lanes per env (see rs_step4_kernel): the visibility graph behind the geodesic cache (up to 28 x 27 / 2
// vertex pairs x 7 rectangles of exact segment tests) and its relaxation are split over the group.
__global__ void __launch_bounds__(64) rs_reset4_kernel(RsParams P, const uint8_t* __restrict__ mask, float* obs, RsOut O) {
    extern __shared__ __align__(16) unsigned char smem[];
    int* lds_geo = reinterpret_cast<int*>(smem);
    uint32_t* lds_adj = reinterpret_cast<uint32_t*>(smem + RS_MAX_VERT * RS_WAVE * 4);
    double* lds_d = reinterpret_cast<double*>(smem + 2 * RS_MAX_VERT * RS_WAVE * 4);
    float* tile = reinterpret_cast<float*>(smem + 2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8);
    const int S = rs_tile_stride(P.A);
    int* flags = reinterpret_cast<int*>(tile + RS_WAVE * S);
    const int lane = threadIdx.x & 63, slot = lane & 15, cj = lane >> 4;
    const int n = blockIdx.x * 16 + slot;
    const bool active = n < P.N && (mask == nullptr || mask[n] != 0);
    RsGeo g{lds_geo, RS_WAVE, slot, 0};
    if (P.group > 1 && active) {                       // shared layouts were resampled by rs_geom_kernel
        const int gi = n / P.group;
        g.n = P.num_obs[gi];
        for (int w = cj; w < 4 * g.n; w += 4) lds_geo[w * RS_WAVE + slot] = P.rect[(size_t)w * P.G + gi];
        if (P.uniform_nobs > 0) g.n = P.uniform_nobs;
    }
    flags[slot] = active ? 1 : 0;
    __syncthreads();
    if (active) rs_env_reset_lane<true, 4>(P, g, n, lds_geo, lds_adj, lds_d, tile + slot * S, O, slot, cj);
    __syncthreads();
    if (obs) {
        const int row = P.A * RS_OBS_DIM;
        const int rows = min(16, P.N - blockIdx.x * 16);
        float* dst = obs + (size_t)blockIdx.x * 16 * row;
        for (int i = lane; i < rows * row; i += RS_WAVE) {
            const int l = i / row, k = i - l * row;
            if (flags[l]) dst[i] = tile[l * S + k];
        }
    }
}

// K2b: RadSearch.refresh_environment for the masked envs (rad_search_env.py:799-874)
template <bool HAS_OBS>
__global__ void __launch_bounds__(64) rs_refresh_kernel(RsParams P, const uint8_t* __restrict__ mask, RsRefresh R, float* obs, RsOut O) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool has_obs = HAS_OBS;
    int* lds_geo = reinterpret_cast<int*>(smem);
    uint32_t* lds_adj = reinterpret_cast<uint32_t*>(smem + RS_MAX_VERT * RS_WAVE * 4);
    double* lds_d = reinterpret_cast<double*>(smem + 2 * RS_MAX_VERT * RS_WAVE * 4);
    float* tile = reinterpret_cast<float*>(smem + (has_obs ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0));
    int* flags = reinterpret_cast<int*>(tile + RS_WAVE * rs_tile_stride(P.A));
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = n < P.N && (mask == nullptr || mask[n] != 0);
    RsGeo g{lds_geo, 0, 0, 0};
    flags[lane] = active ? 1 : 0;
    __syncthreads();
    if (active) rs_env_refresh_lane<HAS_OBS>(P, g, n, R, lds_geo, lds_adj, lds_d, tile + lane * rs_tile_stride(P.A), O);
    __syncthreads();
    rs_copy_out(P, obs, tile, flags, blockIdx.x * blockDim.x);
}

__global__ void __launch_bounds__(64) rs_action_uniform_kernel(RsParams P, float* __restrict__ u) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= P.N) return;
    const uint32_t episode = P.episode[n] - 1u, t = P.tstep[n];
    for (int a = 0; a < P.A; ++a) {
        u32x4 o = philox4x32_10(0u, t, episode, RS_STREAM_ACT + (uint32_t)a, P.seed, P.env_id_base + (uint32_t)n);
        u[(size_t)n * P.A + a] = (float)(o.x >> 8) * (1.0f / 16777216.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// K4: GAE(lambda) + rewards-to-go, PPOBuffer.GAE_advantage_and_rewardsToGO (ppo.py:391-423) for the
// whole time-major buffer.  One column (env x agent trajectory stream) per lane; a reverse scan in
// time with a restart wherever a trajectory was cut.  Every row access is a coalesced 256-byte row.
// float64 recurrences y = x + (d * y_next) reproduce scipy.signal.lfilter (ppo.py:85) bit for bit.
//
// Parallel in TIME as well, without changing a single operation: the scan state is reset at every cut, so a
// lane that owns the time chunk [t_lo, t_hi] first finds the nearest cut at or after t_hi (episodes are at
// most steps_per_episode long), replays the scan from there down to t_hi without storing, and then continues
// through its own chunk storing adv / ret.  480 steps x 4096 columns become 8 chunks x 4096 lanes = 512 waves
// with a critical path of <= chunk + episode steps instead of 480.  The rows of eight steps are loaded
// before they are consumed, so the HBM latency is paid once per eight steps.
#define RS_GAE_U 8
__global__ void __launch_bounds__(64) rs_gae_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                    const uint8_t* __restrict__ cut, const float* __restrict__ last_val,
                                                    float* __restrict__ adv, float* __restrict__ ret, int T, int M,
                                                    double gamma, double gl, int chunk) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int t_lo = blockIdx.y * chunk;
    const int t_hi = min(T, t_lo + chunk) - 1;
    if (t_lo >= T) return;
    // nearest cut at or after t_hi (none: the scan starts at T-1 from a zero state, as the serial scan does)
    // sixteen rows per probe: the flags of a probe are independent loads (one latency), where a row-by-row search paid one
    // HBM latency per step -- up to an episode length of them, which was most of this kernel's time
    int t_start = t_hi;
    for (;;) {
        uint8_t cc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) cc[u] = cut[(size_t)min(t_start + u, T - 1) * M + m];
        int found = -1;
#pragma unroll
        for (int u = 15; u >= 0; --u)
            if (cc[u] != 0 || t_start + u >= T - 1) found = u;
        if (found >= 0) { t_start = min(t_start + found, T - 1); break; }
        t_start += 16;
    }
    double a_acc = 0.0, r_acc = 0.0, v_next = 0.0;
    float r8[RS_GAE_U], v8[RS_GAE_U], l8[RS_GAE_U], rn[RS_GAE_U], vn[RS_GAE_U], ln[RS_GAE_U];
    uint8_t c8[RS_GAE_U], cn[RS_GAE_U];
    auto load_block = [&](int tb, float (&r)[RS_GAE_U], float (&v)[RS_GAE_U], float (&l)[RS_GAE_U], uint8_t (&c)[RS_GAE_U]) {
#pragma unroll
        for (int u = 0; u < RS_GAE_U; ++u) {
            const int t = max(tb - u, t_lo);                       // past the chunk's first row: re-read row t_lo (never used)
            const size_t i = (size_t)t * M + m;
            r[u] = rew[i]; v[u] = val[i]; c[u] = cut[i]; l[u] = last_val[i];
        }
    };
    load_block(t_start, r8, v8, l8, c8);
    for (int tb = t_start; tb >= t_lo; tb -= RS_GAE_U) {
        load_block(tb - RS_GAE_U, rn, vn, ln, cn);                 // the next eight rows travel while these are consumed
#pragma unroll
        for (int u = 0; u < RS_GAE_U; ++u) {
            const int t = tb - u;
            if (t >= t_lo) {
                const double r = (double)r8[u], v = (double)v8[u];
                if (c8[u]) {
                    const double lv = (double)l8[u];
                    // rews = [..., last_val], vals = [..., last_val]: the appended element seeds both scans
                    v_next = lv;
                    r_acc = lv;          // discount_cumsum(rews)[-1] = last_val
                    a_acc = 0.0;
                }
                const double delta = r + gamma * v_next - v;
                a_acc = delta + gl * a_acc;
                r_acc = r + gamma * r_acc;
                if (t <= t_hi) {
                    const size_t i = (size_t)t * M + m;
                    adv[i] = (float)a_acc;
                    ret[i] = (float)r_acc;
                }
                v_next = v;
            }
        }
#pragma unroll
        for (int u = 0; u < RS_GAE_U; ++u) { r8[u] = rn[u]; v8[u] = vn[u]; l8[u] = ln[u]; c8[u] = cn[u]; }
    }
}

// ================================================================================================
// Host side of the C ABI
// ================================================================================================
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carver {
    unsigned char* base; size_t off; rs_handle* h;
    template <typename T> T* take(const char* name, int rows, int cols) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        if (h) { h->fields[h->n_fields++] = rs_field{name, (void*)p, (int)sizeof(T), rows, cols}; }
        off += sizeof(T) * (size_t)rows * (size_t)cols;
        return p;
    }
};

static bool cfg_ok(const rs_config* c) {
    if (!c) return false;
    if (c->num_envs < 1 || c->num_agents < 1 || c->num_agents > RS_MAX_AGENTS) return false;
    if (c->obstruction_count < -1 || c->obstruction_count > RS_MAX_OBS) return false;
    if (c->geom_group_size < 1) return false;
    if (c->bbox[2] <= c->bbox[0] || c->bbox[3] <= c->bbox[1]) return false;
    if (c->bbox[2] > 16000 || c->bbox[3] > 16000 || c->bbox[0] < -16000 || c->bbox[1] < -16000) return false;   // int32 predicates
    if (c->observation_area[0] < 0 || c->observation_area[1] <= c->observation_area[0]) return false;
    return true;
}

static size_t carve(const rs_config* c, unsigned char* base, rs_handle* h) {
    const int N = c->num_envs, A = c->num_agents;
    const int G = (N + c->geom_group_size - 1) / c->geom_group_size;
    Carver cv{base, 0, h};
    RsParams P;
    memset(&P, 0, sizeof(P));
    P.src_x = cv.take<int>("src_x", 1, N);
    P.src_y = cv.take<int>("src_y", 1, N);
    P.intensity = cv.take<int>("intensity", 1, N);
    P.bkg = cv.take<int>("bkg", 1, N);
    P.iter_count = cv.take<int>("iter_count", 1, N);
    P.episode = cv.take<uint32_t>("episode", 1, N);
    P.tstep = cv.take<uint32_t>("tstep", 1, N);
    P.err = cv.take<uint32_t>("err", 1, N);
    P.done = cv.take<uint8_t>("done", 1, N);
    P.epoch_end = cv.take<uint8_t>("epoch_end", 1, N);
    P.num_obs = cv.take<int>("num_obs", 1, G);
    P.rect = cv.take<int>("rect", RS_MAX_VERT, G);
    P.geom_epoch = cv.take<uint32_t>("geom_epoch", 1, G);
    P.dsrc = cv.take<double>("dsrc", RS_MAX_VERT, N);
    P.ax = cv.take<int>("x", A, N);
    P.ay = cv.take<int>("y", A, N);
    P.oobc = cv.take<int>("oob_count", A, N);
    P.sp = cv.take<double>("sp", A, N);
    P.prev = cv.take<double>("prev", A, N);
    P.aflags = cv.take<uint8_t>("aflags", A, N);
    if (h) {
        P.N = N; P.A = A; P.G = G;
        P.obstruction_count = c->obstruction_count; P.enforce = c->enforce_grid_boundaries ? 1 : 0;
        P.uniform_nobs = c->obstruction_count > 0 ? c->obstruction_count : 0;
        P.falloff = c->falloff ? 1 : 0; P.group = c->geom_group_size;
        P.coord_noise = c->coord_noise ? 1 : 0; P.debug = c->debug_spawn ? 1 : 0;
        P.bx0 = c->bbox[0]; P.by0 = c->bbox[1]; P.bx1 = c->bbox[2]; P.by1 = c->bbox[3];
        P.oa_lo = c->observation_area[0]; P.oa_hi = c->observation_area[1];
        P.sa_x0 = P.bx0 + P.oa_lo; P.sa_y0 = P.by0 + P.oa_lo;       // rad_search_env.py:393-420
        P.sa_x1 = P.bx1 - P.oa_hi; P.sa_y1 = P.by1 - P.oa_hi;
        P.obs_hi_x = (int)((double)P.sa_x1 * 0.9); P.obs_hi_y = (int)((double)P.sa_y1 * 0.9);   // :961-966
        double ddx = (double)(P.sa_x1 - P.sa_x1), ddy = (double)(P.sa_y1 - P.sa_y0);            // :423-425
        P.max_dist = sqrt(ddx * ddx + ddy * ddy);
        P.scale = 1.0 / (double)P.sa_y1;                                                        // :435
        P.seed = c->seed; P.env_id_base = c->env_id_base;
        h->P = P;
    }
    return align_up(cv.off, 256);
}

extern "C" {

const char* rs_strerror(int code) {
    switch (code) {
        case RS_OK: return "ok";
        case RS_ERR_INVALID_ARG: return "invalid argument";
        case RS_ERR_HIP: return "HIP runtime error";
        case RS_ERR_WORKSPACE: return "workspace too small or misaligned";
        case RS_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

int rs_abi_version(void) { return RS_ABI_VERSION; }

size_t rs_state_bytes(const rs_config* cfg) {
    if (!cfg_ok(cfg)) return 0;
    return carve(cfg, nullptr, nullptr);
}

int rs_create(const rs_config* cfg, void* workspace, size_t workspace_bytes, rs_stream_t stream, rs_handle** out) {
    if (!out) return RS_ERR_INVALID_ARG;
    *out = nullptr;
    if (!cfg_ok(cfg) || !workspace) return RS_ERR_INVALID_ARG;
    const int sa_w = cfg->bbox[2] - cfg->observation_area[1] - (cfg->bbox[0] + cfg->observation_area[0]);
    const int sa_h = cfg->bbox[3] - cfg->observation_area[1] - (cfg->bbox[1] + cfg->observation_area[0]);
    if (sa_h <= 1000 || sa_w <= 0) return RS_ERR_UNSUPPORTED;     // assert max_dist > 1000 (:431-433)
    size_t need = carve(cfg, nullptr, nullptr);
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return RS_ERR_WORKSPACE;
    rs_handle* h = new (std::nothrow) rs_handle;
    if (!h) return RS_ERR_HIP;
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    h->bytes = carve(cfg, static_cast<unsigned char*>(workspace), h);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(workspace, 0, need, s) != hipSuccess) { delete h; return RS_ERR_HIP; }
    if (hipMemsetAsync(h->P.epoch_end, 1, (size_t)cfg->num_envs, s) != hipSuccess) { delete h; return RS_ERR_HIP; }
    *out = h;
    return RS_OK;
}

void rs_destroy(rs_handle* h) { delete h; }

int rs_state_field(rs_handle* h, const char* name, void** dev_ptr, int32_t* elem_bytes, int32_t* rows, int32_t* cols) {
    if (!h || !name) return RS_ERR_INVALID_ARG;
    for (int i = 0; i < h->n_fields; ++i) {
        if (strcmp(h->fields[i].name, name) == 0) {
            if (dev_ptr) *dev_ptr = h->fields[i].ptr;
            if (elem_bytes) *elem_bytes = h->fields[i].elem;
            if (rows) *rows = h->fields[i].rows;
            if (cols) *cols = h->fields[i].cols;
            return RS_OK;
        }
    }
    return RS_ERR_INVALID_ARG;
}

int rs_set_epoch_end(rs_handle* h, rs_stream_t stream) {
    if (!h) return RS_ERR_INVALID_ARG;
    if (hipMemsetAsync(h->P.epoch_end, 1, (size_t)h->P.N, static_cast<hipStream_t>(stream)) != hipSuccess) return RS_ERR_HIP;
    return RS_OK;
}

static RsOut make_out(float* reward, float* team, uint8_t* done, const rs_info* info) {
    RsOut o;
    memset(&o, 0, sizeof(o));
    o.reward = reward; o.team = team; o.done = done;
    if (info) { o.oob = info->out_of_bounds; o.oobc = info->out_of_bounds_count; o.blocked = info->blocked; o.collision = info->collision; }
    return o;
}

static size_t tile_bytes(int A) { int s = (A * RS_OBS_DIM) | 1; return (size_t)RS_WAVE * s * 4 + RS_WAVE * 4; }

int rs_reset(rs_handle* h, const uint8_t* mask, float* obs, float* reward, float* team, uint8_t* done,
             const rs_info* info, rs_stream_t stream) {
    if (!h) return RS_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RsParams& P = h->P;
    const bool has_obs = P.obstruction_count != 0;
    if (has_obs && P.group > 1) {
        hipLaunchKernelGGL(rs_geom_kernel, dim3((P.G + 63) / 64), dim3(64), RS_MAX_VERT * RS_WAVE * 4, s, P, mask);
    }
    size_t lds = tile_bytes(P.A) + (has_obs ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0);
    if (has_obs) hipLaunchKernelGGL(rs_reset4_kernel, dim3((P.N + 15) / 16), dim3(64), lds, s, P, mask, obs, make_out(reward, team, done, info));
    else hipLaunchKernelGGL(rs_reset_kernel<false>, dim3((P.N + 63) / 64), dim3(64), lds, s, P, mask, obs, make_out(reward, team, done, info));
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_refresh(rs_handle* h, const uint8_t* mask, const int32_t* src_xy, const int32_t* det_xy, const int32_t* intensity,
               const int32_t* bkg, const int32_t* num_obs, const int32_t* rects, float* obs, float* reward, float* team,
               uint8_t* done, const rs_info* info, rs_stream_t stream) {
    if (!h || !src_xy || !det_xy || !intensity || !bkg || ((num_obs == nullptr) != (rects == nullptr))) return RS_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RsParams& P = h->P;
    const bool has_obs = P.obstruction_count != 0;
    if (num_obs && (!has_obs || P.group > 1)) return RS_ERR_UNSUPPORTED;     // layouts are per env; obstacle-free handles hold none
    if (num_obs) h->P.uniform_nobs = 0;                  // saved layouts may hold any number of rectangles from now on
    RsRefresh R{src_xy, det_xy, intensity, bkg, num_obs, rects};
    size_t lds = tile_bytes(P.A) + (has_obs ? (2 * RS_MAX_VERT * RS_WAVE * 4 + RS_MAX_VERT * RS_WAVE * 8) : 0);
    if (has_obs) hipLaunchKernelGGL(rs_refresh_kernel<true>, dim3((P.N + 63) / 64), dim3(64), lds, s, P, mask, R, obs, make_out(reward, team, done, info));
    else hipLaunchKernelGGL(rs_refresh_kernel<false>, dim3((P.N + 63) / 64), dim3(64), lds, s, P, mask, R, obs, make_out(reward, team, done, info));
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_step(rs_handle* h, const int8_t* actions, float* obs, float* reward, float* team, uint8_t* done,
            const rs_info* info, rs_stream_t stream) {
    if (!h || !actions) return RS_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RsParams& P = h->P;
    size_t lds = tile_bytes(P.A) + (P.obstruction_count != 0 ? RS_MAX_VERT * RS_WAVE * 4 : 0);
    if (P.obstruction_count != 0) hipLaunchKernelGGL(rs_step4_kernel, dim3((P.N + 15) / 16), dim3(64), lds, s, P, actions, obs, make_out(reward, team, done, info));
    else hipLaunchKernelGGL(rs_step_kernel<false>, dim3((P.N + 63) / 64), dim3(64), lds, s, P, actions, obs, make_out(reward, team, done, info));
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_action_uniforms(rs_handle* h, float* u, rs_stream_t stream) {
    if (!h || !u) return RS_ERR_INVALID_ARG;
    const RsParams& P = h->P;
    hipLaunchKernelGGL(rs_action_uniform_kernel, dim3((P.N + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), P, u);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_error_flags(rs_handle* h, rs_stream_t stream, uint32_t* flags_out) {
    if (!h || !flags_out) return RS_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = h->P.N;
    uint32_t* host = new (std::nothrow) uint32_t[N];
    if (!host) return RS_ERR_HIP;
    int rc = RS_OK;
    if (hipMemcpyAsync(host, h->P.err, sizeof(uint32_t) * (size_t)N, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) rc = RS_ERR_HIP;
    uint32_t acc = 0;
    if (rc == RS_OK) for (int i = 0; i < N; ++i) acc |= host[i];
    delete[] host;
    *flags_out = acc;
    return rc;
}

int rs_gae(const float* rew, const float* val, const uint8_t* cut, const float* last_val, float* adv, float* ret,
           int32_t T, int32_t M, double gamma, double lam, rs_stream_t stream) {
    if (!rew || !val || !cut || !last_val || !adv || !ret || T < 1 || M < 1) return RS_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // time chunks: enough lanes to fill the chip (>= ~512 waves) but chunks no shorter than 32 steps (the replay from the
    // next cut costs up to one episode per chunk)
    const int col_waves = (M + 63) / 64;
    int chunks = (512 + col_waves - 1) / col_waves;
    chunks = chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks);
    int chunk = (T + chunks - 1) / chunks;
    if (chunk < 32) chunk = 32;
    chunks = (T + chunk - 1) / chunk;
    hipLaunchKernelGGL(rs_gae_kernel, dim3(col_waves, chunks), dim3(64), 0, s, rew, val, cut, last_val, adv, ret, T, M,
                       gamma, gamma * lam, chunk);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

#ifdef RS_STEP_STAMPS
// diagnostic build only: read (and optionally clear) the phase-cycle table of the env step; out[16]
int rs_debug_step_stamps(unsigned long long* out, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return RS_ERR_HIP;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rs_step_stamp_table), sizeof(unsigned long long) * 16) != hipSuccess) return RS_ERR_HIP;
    if (clear) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rs_step_stamp_table), z, sizeof(z)) != hipSuccess) return RS_ERR_HIP;
    }
    return RS_OK;
}
#endif

}  // extern "C"
