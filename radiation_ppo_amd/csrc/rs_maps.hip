// rs_maps.hip -- K5: the RAD-TEAM heat-map builder (MapsBuffer.observation_to_map,
// algos/multiagent/NeuralNetworkCores/RADTEAM_core.py:532-616) for N envs, one env per lane, plus the map-stack
// builder that feeds the CNN actor / critic.  Maps stay resident in HBM ([N][X*Y] per map); a step touches only
// the cells the agents stand on.  The readings estimator (median of all readings taken in a cell, :160-166) keeps
// a per-env ring of (value, previous entry of the same cell) records chained per cell, so the median of a cell
// costs O(entries in that cell).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <new>

#include "../../include/radsearch.h"
#include "rs_handle.hpp"

struct RsMapsParams {
    int N, A, L, X, Y, C, cap, base;     // C = X*Y cells, cap = ring capacity, base = (L+1)*A
    double ra;
    float *comb, *read, *visit, *obst;   // [N][C]
    uint16_t *shadow;                    // [N][C] visit count (number of visits so far)
    uint16_t *head;                      // [N][C] index+1 of the newest ring entry of the cell (0 = none)
    float *ring_val;                     // [N][cap]
    uint16_t *ring_prev;                 // [N][cap] index+1 of the previous entry of the same cell
    int *ring_n;                         // [N]
    int *cell, *pred_cell;               // [N][A] last_coords (-1 none), last prediction cell per owner (-1 none)
    int *wcount; double *wmean, *wsq, *wstd;   // StatisticStandardization per env
    const float* visit_table;            // [base + 1]
    uint32_t* err;                       // [N]
};

struct rs_maps {
    RsMapsParams P;
    size_t bytes;
    int n_fields;
    rs_field fields[24];
};

#define RS_MAPERR_RING_FULL 1u
#define RS_MAPERR_VISIT_OVERFLOW 2u
#define RS_MAPERR_OFF_MAP 4u            // a coordinate beyond the map (the reference's numpy indexing raises IndexError there)

// ------------------------------------------------------------------------------------------------
// median of the readings recorded in `cell` (statistics.median: mean of the two middle values when even).
// The chain of the cell is walked ONCE (a pointer chase: one memory latency per entry) into the lane's column of an LDS buffer;
// the rank counting then runs over that column.  (The first version chased the chain again inside both rank loops: O(m^2)
// dependent loads -- with a trained policy, whose agents revisit the cells around the source, the collector step of config 4
// slowed down 2.6x over 30 epochs, scripts/epoch_times_cnn.py.)  The buffer holds min(ring capacity, 512) entries per lane (the ring
// capacity is steps_per_episode x agents = 480 for the reference's settings: 120 KB of LDS per 64-env block -- the grid has only N / 64
// blocks, so one block per CU is all it needs); chains longer than the buffer take the chasing path.
#define RS_MED_CAP_MAX 512
__device__ __forceinline__ double rs_cell_median(const RsMapsParams& M, int n, int cellidx, float (*buf)[64], int medcap) {
    const float* val = M.ring_val + (size_t)n * M.cap;
    const uint16_t* prv = M.ring_prev + (size_t)n * M.cap;
    const int lane = threadIdx.x & 63;
    const int head = M.head[(size_t)n * M.C + cellidx];
    int m = 0;
    for (int j = head; j != 0; j = prv[j - 1]) {
        if (m < medcap) buf[m][lane] = val[j - 1];
        ++m;
    }
    const int k_lo = (m - 1) / 2, k_hi = m / 2;
    double v_lo = 0.0, v_hi = 0.0;
    if (m <= medcap) {
        // Hoare's selection of the k_hi-th smallest inside the lane's LDS column (expected O(m)); afterwards everything left of
        // k_hi is <= it, so for an even count the other middle value is the maximum of that part.  Only VALUES matter for a
        // median, so this equals the rank-counting result below bit for bit.
        int lo = 0, hi = m - 1;
        while (lo < hi) {
            const float pivot = buf[(lo + hi) >> 1][lane];
            int i = lo, j = hi;
            while (i <= j) {
                while (buf[i][lane] < pivot) ++i;
                while (buf[j][lane] > pivot) --j;
                if (i <= j) {
                    const float t = buf[i][lane]; buf[i][lane] = buf[j][lane]; buf[j][lane] = t;
                    ++i; --j;
                }
            }
            if (k_hi <= j) hi = j; else if (k_hi >= i) lo = i; else break;
        }
        const float vh = buf[k_hi][lane];
        float vl = vh;
        if (k_lo != k_hi) {
            vl = buf[0][lane];
            for (int q = 1; q < k_hi; ++q) vl = fmaxf(vl, buf[q][lane]);
        }
        return (k_lo == k_hi) ? (double)vh : ((double)vl + (double)vh) / 2.0;
    }
    // chains longer than the buffer: k-th smallest by rank counting along the chain; ties broken by chain position
    int ia = 0;
    for (int a = head; a != 0; a = prv[a - 1], ++ia) {
        const float va = val[a - 1];
        int less = 0, eq_before = 0, ib = 0;
        for (int b = head; b != 0; b = prv[b - 1], ++ib) {
            const float vb = val[b - 1];
            less += (vb < va) ? 1 : 0;
            eq_before += (vb == va && ib < ia) ? 1 : 0;
        }
        const int rank = less + eq_before;
        if (rank == k_lo) v_lo = (double)va;
        if (rank == k_hi) v_hi = (double)va;
    }
    return (k_lo == k_hi) ? v_lo : (v_lo + v_hi) / 2.0;
}

__global__ void __launch_bounds__(64) rs_maps_update_kernel(RsMapsParams M, RsParams E, const float* __restrict__ obs,
                                                            const float* __restrict__ pred, const uint8_t* __restrict__ mask, int medcap) {
    extern __shared__ __align__(16) float med_dyn[];
    float (*med_buf)[64] = reinterpret_cast<float (*)[64]>(med_dyn);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= M.N || (mask && !mask[n])) return;
    const int A = M.A, C = M.C;
    float* comb = M.comb + (size_t)n * C;
    float* readm = M.read + (size_t)n * C;
    float* visit = M.visit + (size_t)n * C;
    float* obst = M.obst + (size_t)n * C;
    uint16_t* shadow = M.shadow + (size_t)n * C;
    uint16_t* head = M.head + (size_t)n * C;
    float* rval = M.ring_val + (size_t)n * M.cap;
    uint16_t* rprev = M.ring_prev + (size_t)n * M.cap;
    uint32_t err = 0;
    // inflate coordinates (:692-715): int(obs[1] * resolution_accuracy) with obs[1] = x * scale in float64
    int cur[RS_MAX_AGENTS];
#pragma unroll
    for (int a = 0; a < RS_MAX_AGENTS; ++a) {
        cur[a] = 0;
        if (a < A) {
            int cx = (int)(((double)E.ax[(size_t)a * E.N + n] * E.scale) * M.ra);
            int cy = (int)(((double)E.ay[(size_t)a * E.N + n] * E.scale) * M.ra);
            if (E.coord_noise) {
                // the reference bins what the agent OBSERVES, int(single_observation[1] * resolution_accuracy) (RADTEAM_core.py:705-711),
                // and with coord_noise the observed coordinates carry N(0, 5 cm) (rad_search_env.py:569-580): take the cell from the
                // observation row (its float32; the exact position is used otherwise so that no rounding of the row can move a cell)
                const float* o = obs + ((size_t)n * A + a) * RS_OBS_DIM;
                cx = (int)((double)o[1] * M.ra);
                cy = (int)((double)o[2] * M.ra);
            }
            // a NEGATIVE index is legal numpy: map[-k] is the k-th cell from the end.  Without enforced walls a detector left of /
            // below the area lands there (what the 147 x 147 map's extra cells absorb, RADTEAM_core.py:1727-1738)
            if (cx < 0) cx += M.X;
            if (cy < 0) cy += M.Y;
            if (cx < 0 || cx >= M.X || cy < 0 || cy >= M.Y) err |= RS_MAPERR_OFF_MAP;
            cur[a] = min(max(cx, 0), M.X - 1) * M.Y + min(max(cy, 0), M.Y - 1);
        }
    }
    // readings buffer first, for every agent (:541-545)
    int rn = M.ring_n[n];
#pragma unroll
    for (int a = 0; a < RS_MAX_AGENTS; ++a) {
        if (a < A) {
            if (rn < M.cap) {
                rval[rn] = obs[((size_t)n * A + a) * RS_OBS_DIM];
                rprev[rn] = head[cur[a]];
                head[cur[a]] = (uint16_t)(rn + 1);
                rn += 1;
            } else err |= RS_MAPERR_RING_FULL;
        }
    }
    M.ring_n[n] = rn;
    int wc = M.wcount[n];
    double wmean = M.wmean[n], wsq = M.wsq[n], wstd = M.wstd[n];
#pragma unroll
    for (int a = 0; a < RS_MAX_AGENTS; ++a) {
        if (a >= A) continue;
        const int c = cur[a];
        const int last = M.cell[(size_t)n * A + a];
        // combined locations (:768-790)
        if (last >= 0) comb[last] -= 1.0f;
        comb[c] += 1.0f;
        // readings: median estimate -> Welford -> z-score (:844-872, StatisticStandardization :215-265)
        const double est = rs_cell_median(M, n, c, med_buf, medcap);
        wc += 1;
        if (wc == 1) wmean = est;
        else {
            const double mean_new = wmean + (est - wmean) / (double)wc;
            wsq = wsq + (est - wmean) * (est - mean_new);
            wmean = mean_new;
            wstd = fmax(sqrt(wsq / (double)(wc - 1)), 1.0);
        }
        readm[c] = (float)((est - wmean) / wstd);
        // visit counts (:874-908): table[c] = log(2 + 2c, base) / log(2 base, base)
        const int vc = shadow[c];
        if (vc > M.base) err |= RS_MAPERR_VISIT_OVERFLOW;
        visit[c] = M.visit_table[min(vc, M.base)];
        shadow[c] = (uint16_t)(vc + 1);
        // obstacles (:910-932): the last non-zero detection wins
        const float* det = obs + ((size_t)n * A + a) * RS_OBS_DIM + 3;
        float dv = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dv = (det[k] != 0.0f) ? det[k] : dv;
        if (dv != 0.0f) obst[c] = dv;
        M.cell[(size_t)n * A + a] = c;
        // prediction cell of owner a (the prediction map is a one-hot at the last prediction, :747-766)
        if (pred) {
            const int px = (int)((double)pred[((size_t)n * A + a) * 2 + 0] * M.ra);
            const int py = (int)((double)pred[((size_t)n * A + a) * 2 + 1] * M.ra);
            if (px >= 0 && px < M.X && py >= 0 && py < M.Y) M.pred_cell[(size_t)n * A + a] = px * M.Y + py;
        }
    }
    M.wcount[n] = wc; M.wmean[n] = wmean; M.wsq[n] = wsq; M.wstd[n] = wstd;
    if (err) M.err[n] |= err;
}

// one workgroup per env: zero the maps of the masked envs with coalesced stores
__global__ void __launch_bounds__(256) rs_maps_reset_kernel(RsMapsParams M, const uint8_t* __restrict__ mask) {
    const int n = blockIdx.x;
    if (mask && !mask[n]) return;
    const size_t o = (size_t)n * M.C;
    for (int i = threadIdx.x; i < M.C; i += blockDim.x) {
        M.comb[o + i] = 0.f; M.read[o + i] = 0.f; M.visit[o + i] = 0.f; M.obst[o + i] = 0.f;
        M.shadow[o + i] = 0; M.head[o + i] = 0;
    }
    if (threadIdx.x < M.A) { M.cell[(size_t)n * M.A + threadIdx.x] = -1; M.pred_cell[(size_t)n * M.A + threadIdx.x] = -1; }
    if (threadIdx.x == 0) { M.ring_n[n] = 0; M.wcount[n] = 0; M.wmean[n] = 0.0; M.wsq[n] = 0.0; M.wstd[n] = 1.0; }
}

// one workgroup per env: critic stack [N][4][C], actor stacks [N][A][6][C]; every row is a coalesced stream
__global__ void __launch_bounds__(256) rs_maps_stack_kernel(RsMapsParams M, float* __restrict__ actor, float* __restrict__ critic) {
    const int n = blockIdx.x, C = M.C, A = M.A;
    const size_t o = (size_t)n * C;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        const float cb = M.comb[o + i], rd = M.read[o + i], vs = M.visit[o + i], ob = M.obst[o + i];
        if (critic) {
            float* c = critic + (size_t)n * 4 * C;
            c[i] = cb; c[C + i] = rd; c[2 * C + i] = vs; c[3 * C + i] = ob;
        }
        if (actor) {
            for (int a = 0; a < A; ++a) {
                float* s = actor + ((size_t)n * A + a) * 6 * C;
                const float loc = (M.cell[(size_t)n * A + a] == i) ? 1.0f : 0.0f;
                s[i] = (M.pred_cell[(size_t)n * A + a] == i) ? 1.0f : 0.0f;
                s[C + i] = loc;
                s[2 * C + i] = cb - loc;
                s[3 * C + i] = rd; s[4 * C + i] = vs; s[5 * C + i] = ob;
            }
        }
    }
}

// ================================================================================================
static inline size_t m_align(size_t x) { return (x + 255) / 256 * 256; }

struct MCarver {
    unsigned char* base; size_t off; rs_maps* h;
    template <typename T> T* take(const char* name, int rows, int cols) {
        off = m_align(off);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        if (h && name) h->fields[h->n_fields++] = rs_field{name, (void*)p, (int)sizeof(T), rows, cols};
        off += sizeof(T) * (size_t)rows * (size_t)cols;
        return p;
    }
};

static size_t maps_carve(int N, int A, int L, int X, int Y, unsigned char* base, rs_maps* h) {
    MCarver cv{base, 0, h};
    RsMapsParams P;
    memset(&P, 0, sizeof(P));
    const int C = X * Y, cap = (L + 2) * A;
    P.comb = cv.take<float>("combined", N, C);
    P.read = cv.take<float>("readings", N, C);
    P.visit = cv.take<float>("visits", N, C);
    P.obst = cv.take<float>("obstacles", N, C);
    P.shadow = cv.take<uint16_t>("shadow", N, C);
    P.head = cv.take<uint16_t>(nullptr, N, C);
    P.ring_val = cv.take<float>(nullptr, N, cap);
    P.ring_prev = cv.take<uint16_t>(nullptr, N, cap);
    P.ring_n = cv.take<int>("ring_n", 1, N);
    P.cell = cv.take<int>("cell", N, A);
    P.pred_cell = cv.take<int>("pred_cell", N, A);
    P.wcount = cv.take<int>(nullptr, 1, N);
    P.wmean = cv.take<double>(nullptr, 1, N);
    P.wsq = cv.take<double>(nullptr, 1, N);
    P.wstd = cv.take<double>(nullptr, 1, N);
    P.err = cv.take<uint32_t>("err", 1, N);
    if (h) { P.N = N; P.A = A; P.L = L; P.X = X; P.Y = Y; P.C = C; P.cap = cap; P.base = (L + 1) * A; h->P = P; }
    return m_align(cv.off);
}

extern "C" {

size_t rs_maps_state_bytes(int32_t N, int32_t A, int32_t L, int32_t X, int32_t Y) {
    if (N < 1 || A < 1 || A > RS_MAX_AGENTS || L < 1 || X < 1 || Y < 1 || X * Y > 60000 || (L + 2) * A > 65000) return 0;
    return maps_carve(N, A, L, X, Y, nullptr, nullptr);
}

int rs_maps_create(int32_t N, int32_t A, int32_t L, int32_t X, int32_t Y, double ra, const float* visit_table, void* ws,
                   size_t ws_bytes, rs_stream_t stream, rs_maps** out) {
    if (!out) return RS_ERR_INVALID_ARG;
    *out = nullptr;
    const size_t need = rs_maps_state_bytes(N, A, L, X, Y);
    if (need == 0 || !ws || !visit_table) return RS_ERR_INVALID_ARG;
    if (ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 255u)) return RS_ERR_WORKSPACE;
    rs_maps* h = new (std::nothrow) rs_maps;
    if (!h) return RS_ERR_HIP;
    memset(h, 0, sizeof(*h));
    h->bytes = maps_carve(N, A, L, X, Y, static_cast<unsigned char*>(ws), h);
    h->P.ra = ra;
    h->P.visit_table = visit_table;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(ws, 0, need, s) != hipSuccess) { delete h; return RS_ERR_HIP; }
    hipLaunchKernelGGL(rs_maps_reset_kernel, dim3(N), dim3(256), 0, s, h->P, (const uint8_t*)nullptr);
    *out = h;
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

void rs_maps_destroy(rs_maps* m) { delete m; }

int rs_maps_reset(rs_maps* m, const uint8_t* mask, rs_stream_t stream) {
    if (!m) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_maps_reset_kernel, dim3(m->P.N), dim3(256), 0, static_cast<hipStream_t>(stream), m->P, mask);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_maps_update(rs_maps* m, rs_handle* env, const float* obs, const float* pred, const uint8_t* mask, rs_stream_t stream) {
    if (!m || !env || !obs) return RS_ERR_INVALID_ARG;
    if (env->P.N != m->P.N || env->P.A != m->P.A) return RS_ERR_INVALID_ARG;
    const int medcap = m->P.cap < RS_MED_CAP_MAX ? m->P.cap : RS_MED_CAP_MAX;
    const size_t lds = (size_t)medcap * 64 * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rs_maps_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RS_ERR_HIP;
    hipLaunchKernelGGL(rs_maps_update_kernel, dim3((m->P.N + 63) / 64), dim3(64), lds, static_cast<hipStream_t>(stream), m->P, env->P, obs,
                       pred, mask, medcap);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_maps_stack(rs_maps* m, float* actor_stack, float* critic_stack, rs_stream_t stream) {
    if (!m || (!actor_stack && !critic_stack)) return RS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rs_maps_stack_kernel, dim3(m->P.N), dim3(256), 0, static_cast<hipStream_t>(stream), m->P, actor_stack, critic_stack);
    return hipGetLastError() == hipSuccess ? RS_OK : RS_ERR_HIP;
}

int rs_maps_field(rs_maps* m, const char* name, void** dev_ptr, int32_t* elem_bytes, int32_t* rows, int32_t* cols) {
    if (!m || !name) return RS_ERR_INVALID_ARG;
    for (int i = 0; i < m->n_fields; ++i) {
        if (strcmp(m->fields[i].name, name) == 0) {
            if (dev_ptr) *dev_ptr = m->fields[i].ptr;
            if (elem_bytes) *elem_bytes = m->fields[i].elem;
            if (rows) *rows = m->fields[i].rows;
            if (cols) *cols = m->fields[i].cols;
            return RS_OK;
        }
    }
    return RS_ERR_INVALID_ARG;
}

}  // extern "C"
