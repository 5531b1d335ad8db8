// rs_env.hip -- env kernels (K1 step, K2 reset, K2a shared-geometry resample, K4 GAE) and the C ABI
// declared in include/radsearch.h.  gfx950 only; one environment per wavefront lane; 64-thread
// workgroups (one wave) so that even the 4096-env configuration spreads over 64 CUs and all eight
// XCDs; the blockIdx -> env mapping is identical in every kernel, so an env's state stays in the
// L2 of the XCD its block lands on from launch to launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <new>

#include "../../include/radsearch.h"
#include "rs_device.hpp"

// ------------------------------------------------------------------------------------------------
// LDS carve-up of one wave.  geo: obstacle rectangles (per-lane or shared layout, see RsGeo);
// tile: the wave's [64][A*11] observation rows (+ odd padding -> conflict-free row writes) that are
// copied out as whole 256-byte rows; flags: which lanes produced a row.
struct WaveLds {
    int* geo;       // 28*64 ints
    float* tile;    // 64 * S floats
    int* flags;     // 64 ints
};
__device__ __forceinline__ int rs_tile_stride(int A) { int s = A * RS_OBS_DIM; return s | 1; }

__device__ __forceinline__ void rs_load_geo(const RsParams& P, int n, bool active, int* lds_geo, RsGeo& g) {
    const int lane = threadIdx.x & 63;
    if (P.obstruction_count == 0) { g.r = lds_geo; g.stride = 0; g.off = 0; g.n = 0; return; }
    const bool shared = (P.group % RS_WAVE) == 0;
    if (shared) {
        const int grp = (blockIdx.x * blockDim.x + (threadIdx.x & ~63)) / P.group;   // wave-uniform
        const int gi = min(grp, P.G - 1);
        if (lane < RS_MAX_VERT) lds_geo[lane] = P.rect[(size_t)lane * P.G + gi];
        g.r = lds_geo; g.stride = 1; g.off = 0; g.n = P.num_obs[gi];
    } else {
        const int gi = active ? n / P.group : 0;
        const int no = active ? P.num_obs[gi] : 0;
        for (int w = 0; w < 4 * no; ++w) lds_geo[w * RS_WAVE + lane] = P.rect[(size_t)w * P.G + gi];
        g.r = lds_geo; g.stride = RS_WAVE; g.off = lane; g.n = no;
    }
}

__device__ __forceinline__ void rs_copy_out(const RsParams& P, float* obs, const float* tile, const int* flags, int wave_env0) {
    if (!obs) return;
    const int lane = threadIdx.x & 63;
    const int row = P.A * RS_OBS_DIM, S = rs_tile_stride(P.A);
    const int total = RS_WAVE * row;
    float* dst = obs + (size_t)wave_env0 * row;
    for (int i = lane; i < total; i += RS_WAVE) {
        int l = i / row, k = i - l * row;
        if (flags[l]) dst[i] = tile[l * S + k];
    }
}

// ------------------------------------------------------------------------------------------------
// K1: RadSearch.step for N envs (rad_search_env.py:443-728)
template <bool HAS_OBS>
__global__ void __launch_bounds__(64) rs_step_kernel(RsParams P, const int8_t* __restrict__ actions, float* obs, RsOut O) {
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
        auto act_of = [&](int a) -> int {
            int v = arow[a];
            if (v == -1) return RS_IDLE;
            if (v < 0 || v > 8) { bad = RS_ENVERR_BAD_ACTION; return RS_IDLE; }
            return v;
        };
        rs_env_step_lane<HAS_OBS>(P, g, n, act_of, o);
        if (bad) P.err[n] |= bad;
    }
    __syncthreads();
    rs_copy_out(P, obs, tile, flags, blockIdx.x * blockDim.x);
}

// ------------------------------------------------------------------------------------------------
// rectangles' boundaries share a point  <=>  isclose(boundary_distance(poly1, poly2), 0) (:988)
__device__ __forceinline__ bool rs_rects_touch(int ax0, int ay0, int ax1, int ay1, int bx0, int by0, int bx1, int by1) {
    bool overlap = ax0 <= bx1 && bx0 <= ax1 && ay0 <= by1 && by0 <= ay1;
    bool a_in_b = bx0 < ax0 && ax1 < bx1 && by0 < ay0 && ay1 < by1;
    bool b_in_a = ax0 < bx0 && bx1 < ax1 && ay0 < by0 && by1 < ay1;
    return overlap && !a_in_b && !b_in_a;
}

// create_obs (:948-1011) into a per-lane LDS layout (stride 64) -- draws continue the caller's sequence
__device__ __forceinline__ int rs_create_obs(const RsParams& P, RsDrawSeq& seq, int* lds_geo, int stride, int off) {
    int num = P.obstruction_count;
    if (num == -1) num = seq.integers(1, 6);
    int ii = 0;
    while (ii < num) {
        int sx = seq.integers(P.sa_x0, P.obs_hi_x);
        int sy = seq.integers(P.sa_y0, P.obs_hi_y);
        int ex = seq.integers(P.oa_lo, P.oa_hi);
        int ey = seq.integers(P.oa_lo, P.oa_hi);
        bool touch = false;
        for (int kk = 0; kk < ii && !touch; ++kk) {
            int x0 = lds_geo[(kk * 4 + 0) * stride + off], y0 = lds_geo[(kk * 4 + 1) * stride + off];
            int x1 = lds_geo[(kk * 4 + 2) * stride + off], y1 = lds_geo[(kk * 4 + 3) * stride + off];
            touch = rs_rects_touch(x0, y0, x1, y1, sx, sy, sx + ex, sy + ey);
        }
        if (!touch) {
            lds_geo[(ii * 4 + 0) * stride + off] = sx;
            lds_geo[(ii * 4 + 1) * stride + off] = sy;
            lds_geo[(ii * 4 + 2) * stride + off] = sx + ex;
            lds_geo[(ii * 4 + 3) * stride + off] = sy + ey;
            ii += 1;
        }
    }
    return num;
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
    int num = rs_create_obs(P, seq, lds_geo, RS_WAVE, lane);
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
        const uint32_t episode = P.episode[n];
        RsDrawSeq seq{P.seed, P.env_id_base + (uint32_t)n, episode, RS_STREAM_RESET, 0u};
        // ---- per-env obstacle layout (geom_group_size == 1): resample when epoch_end is set (:744-762)
        if (has_obs && P.group == 1) {
            g.r = lds_geo; g.stride = RS_WAVE; g.off = lane;
            if (P.epoch_end[n]) {
                int num = rs_create_obs(P, seq, lds_geo, RS_WAVE, lane);
                P.num_obs[n] = num;
                for (int w = 0; w < 4 * num; ++w) P.rect[(size_t)w * P.G + n] = lds_geo[w * RS_WAVE + lane];
                g.n = num;
            } else {
                g.n = P.num_obs[n];
                for (int w = 0; w < 4 * g.n; ++w) lds_geo[w * RS_WAVE + lane] = P.rect[(size_t)w * P.G + n];
            }
        }
        P.epoch_end[n] = 0;
        // ---- sample_source_loc_pos (:1013-1131); rand_point uses the x-range for both axes (:1033)
        int srx = seq.integers(P.sa_x0, P.sa_x1), sry = seq.integers(P.sa_x0, P.sa_x1);
        int dtx = seq.integers(P.sa_x0, P.sa_x1), dty = seq.integers(P.sa_x0, P.sa_x1);
        for (;;) {
            bool inside = false;
            for (int o = 0; o < g.n && !inside; ++o) {
                int x0, y0, x1, y1; g.rect(o, x0, y0, x1, y1);
                inside = (x0 <= dtx && dtx <= x1 && y0 <= dty && dty <= y1);
            }
            if (!inside) break;
            dtx = seq.integers(P.sa_x0, P.sa_x1); dty = seq.integers(P.sa_x0, P.sa_x1);
        }
        int num_retry = 0;
        for (;;) {
            while (rs_dist_i(dtx, dty, srx, sry) < 1000.0) { srx = seq.integers(P.sa_x0, P.sa_x1); sry = seq.integers(P.sa_x0, P.sa_x1); }
            bool resamp = false, inter = false;
            for (int o = 0; o < g.n && !resamp; ++o) {
                int x0, y0, x1, y1; g.rect(o, x0, y0, x1, y1);
                if (x0 <= srx && srx <= x1 && y0 <= sry && sry <= y1) resamp = true;
                if (!resamp && rs_seg_rect_close(dtx, dty, srx, sry, x0, y0, x1, y1)) inter = true;
            }
            if (g.n == 0 || (num_retry > 20 && !resamp)) break;
            else if (resamp || !inter) { srx = seq.integers(P.sa_x0, P.sa_x1); sry = seq.integers(P.sa_x0, P.sa_x1); num_retry += 1; }
            else break;
        }
        // ---- geodesic distances source -> rectangle vertices (visibility graph relaxation)
        const int V = HAS_OBS ? 4 * g.n : 0;
        for (int v = 0; v < V; ++v) {
            int vx, vy; g.vertex(v, vx, vy);
            lds_d[v * RS_WAVE + lane] = rs_visible(g, srx, sry, vx, vy) ? rs_dist_i(srx, sry, vx, vy) : INFINITY;
            uint32_t m = 0;
            for (int u = 0; u < V; ++u) {
                if (u == v) continue;
                int ux, uy; g.vertex(u, ux, uy);
                if (rs_visible(g, ux, uy, vx, vy)) m |= 1u << u;
            }
            lds_adj[v * RS_WAVE + lane] = m;
        }
        bool changed = V > 0;
        while (changed) {
            changed = false;
            for (int v = 0; v < V; ++v) {
                int vx, vy; g.vertex(v, vx, vy);
                uint32_t m = lds_adj[v * RS_WAVE + lane];
                double dv = lds_d[v * RS_WAVE + lane];
                for (int u = 0; u < V; ++u) {
                    if (!(m >> u & 1u)) continue;
                    int ux, uy; g.vertex(u, ux, uy);
                    double c = lds_d[u * RS_WAVE + lane] + rs_dist_i(ux, uy, vx, vy);
                    if (c < dv) { dv = c; changed = true; }
                }
                lds_d[v * RS_WAVE + lane] = dv;
            }
        }
        for (int v = 0; v < V; ++v) P.dsrc[(size_t)v * N + n] = lds_d[v * RS_WAVE + lane];
        double prev = (HAS_OBS && g.n > 0) ? rs_shortest_path(g, P.dsrc, N, n, srx, sry, dtx, dty) : rs_dist_i(srx, sry, dtx, dty);
        int intensity = seq.integers(1000000, 10000000);    // :778
        int bkg = seq.integers(10, 51);                     // :779
        // ---- state write (Agent.reset :292-301, reset :736-742, :771-776)
        P.src_x[n] = srx; P.src_y[n] = sry; P.intensity[n] = intensity; P.bkg[n] = bkg;
        P.done[n] = 0; P.iter_count[n] = 0; P.tstep[n] = 0;
        for (int a = 0; a < A; ++a) {
            size_t ia = (size_t)a * N + n;
            P.ax[ia] = dtx; P.ay[ia] = dty; P.sp[ia] = prev; P.prev[ia] = prev; P.oobc[ia] = 0; P.aflags[ia] = 0;
        }
        // ---- initial observation: step(None) (:794-797)
        RsOut o = O;
        o.obs_row = tile + lane * rs_tile_stride(A);
        P.episode[n] = episode + 1;       // draws of this episode are keyed by `episode`
        rs_env_step_lane<HAS_OBS>(P, g, n, [](int) -> int { return RS_ACT_NONE; }, o);
        P.iter_count[n] = 0;
    }
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
__global__ void __launch_bounds__(256) rs_gae_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                     const uint8_t* __restrict__ cut, const float* __restrict__ last_val,
                                                     float* __restrict__ adv, float* __restrict__ ret, int T, int M,
                                                     double gamma, double gl) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    double a_acc = 0.0, r_acc = 0.0, v_next = 0.0;
    for (int t = T - 1; t >= 0; --t) {
        const size_t i = (size_t)t * M + m;
        const double r = (double)rew[i], v = (double)val[i];
        if (cut[i]) {
            const double lv = (double)last_val[i];
            // rews = [..., last_val], vals = [..., last_val]: the appended element seeds both scans
            v_next = lv;
            r_acc = lv;          // discount_cumsum(rews)[-1] = last_val
            a_acc = 0.0;
        }
        const double delta = r + gamma * v_next - v;
        a_acc = delta + gl * a_acc;
        r_acc = r + gamma * r_acc;
        adv[i] = (float)a_acc;
        ret[i] = (float)r_acc;
        v_next = v;
    }
}

// ================================================================================================
// Host side of the C ABI
// ================================================================================================
struct rs_field { const char* name; void* ptr; int elem, rows, cols; };

struct rs_handle {
    rs_config cfg;
    RsParams P;
    int device;
    size_t bytes;
    int n_fields;
    rs_field fields[32];
};

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
    if (c->bbox[2] > 20000 || c->bbox[3] > 20000 || c->bbox[0] < -20000 || c->bbox[1] < -20000) return false;   // int32 predicates
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
        P.falloff = c->falloff ? 1 : 0; P.group = c->geom_group_size;
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
    if (has_obs) hipLaunchKernelGGL(rs_reset_kernel<true>, dim3((P.N + 63) / 64), dim3(64), lds, s, P, mask, obs, make_out(reward, team, done, info));
    else hipLaunchKernelGGL(rs_reset_kernel<false>, dim3((P.N + 63) / 64), dim3(64), lds, s, P, mask, obs, make_out(reward, team, done, info));
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_step(rs_handle* h, const int8_t* actions, float* obs, float* reward, float* team, uint8_t* done,
            const rs_info* info, rs_stream_t stream) {
    if (!h || !actions) return RS_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const RsParams& P = h->P;
    size_t lds = tile_bytes(P.A) + (P.obstruction_count != 0 ? RS_MAX_VERT * RS_WAVE * 4 : 0);
    if (P.obstruction_count != 0) hipLaunchKernelGGL(rs_step_kernel<true>, dim3((P.N + 63) / 64), dim3(64), lds, s, P, actions, obs, make_out(reward, team, done, info));
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
    const int threads = (M >= 256 * 256) ? 256 : 64;
    hipLaunchKernelGGL(rs_gae_kernel, dim3((M + threads - 1) / threads), dim3(threads), 0, s, rew, val, cut, last_val, adv, ret, T, M,
                       gamma, gamma * lam);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

}  // extern "C"
