/* CPU ORACLE in C (test infrastructure, NOT product code; built into oracle/_build/ by oracle/Makefile).
 *
 * Scalar C restatement of the obstacle-free radiation-search environment (the bench workload, BASELINE.json
 * configs[1]) with the same Philox/PTRS streams as oracle/radsearch_oracle.py and the HIP kernels.  It follows
 * gym_rad_search/gym_rad_search/envs/rad_search_env.py (paths relative to /root/reference):
 *     get_step :178-224, take_action :876-946, agent_step :460-613, step :616-728, reset :730-797,
 *     sample_source_loc_pos :1013-1131 (obstacle-free branch), obstruction_sensors :1232-1259 (walls).
 * Pinning: tests/test_oracle_c.py checks it event by event against the Python oracle (which is pinned to golden
 * vectors captured from the real reference) -- float64-exact.  Its only other use is bench.py's cpu_baseline leg
 * (kind "port"): the reference's env.step as compiled scalar code on the host cores.
 * Compile with -ffp-contract=off (see Makefile): the float64 arithmetic must round like Python's. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXA 8
#define IDLE 8
#define ACT_NONE 9

typedef struct {
    int A, enforce, falloff;
    uint32_t seed, env_id, episode, cur_episode, t;   /* episode = next id; cur_episode keys the draws */
    int bx0, by0, bx1, by1, sa_x0, sa_y0, sa_x1, sa_y1;
    double max_dist, scale;
    int src_x, src_y, intensity, bkg, iter_count, done;
    int x[MAXA], y[MAXA], oobc[MAXA], oob[MAXA], coll[MAXA];
    double sp[MAXA], prev[MAXA];
    uint32_t err;
    uint32_t reset_idx;
} rso_env;

static const int STEP_DX[9] = {-100, -71, 0, 71, 100, 71, 0, -71, 0};
static const int STEP_DY[9] = {0, 71, 100, 71, 0, -71, -100, -71, 0};

static void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53(uint32_t lo, uint32_t hi) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }

static int draw_int(rso_env* e, int lo, int hi) {
    uint32_t o[4];
    philox(e->reset_idx++, 0u, e->cur_episode, 0u, e->seed, e->env_id, o);
    uint64_t x = ((uint64_t)o[1] << 32) | o[0];
    return lo + (int)(((unsigned __int128)x * (uint64_t)(uint32_t)(hi - lo)) >> 64);
}

static int64_t poisson(rso_env* e, double lam, int agent) {
    uint32_t o[4];
    if (lam == 0.0) return 0;
    if (lam < 10.0) {
        double enlam = exp(-lam), prod = 1.0;
        int64_t x = 0;
        for (uint32_t i = 0;; ++i) {
            philox(i, e->t, e->cur_episode, 1u + (uint32_t)agent, e->seed, e->env_id, o);
            prod *= u53(o[0], o[1]);
            if (prod > enlam) x += 1; else return x;
        }
    }
    double slam = sqrt(lam), b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b, vr = 0.9277 - 3.6224 / (b - 2.0);
    for (uint32_t i = 0;; ++i) {
        philox(i, e->t, e->cur_episode, 1u + (uint32_t)agent, e->seed, e->env_id, o);
        double u = u53(o[0], o[1]) - 0.5, v = u53(o[2], o[3]);
        double us = 0.5 - fabs(u);
        if (!(us > 0.0)) continue;
        double k = floor((2.0 * a / us + b) * u + lam + 0.43);
        if (us >= 0.07 && v <= vr) return (int64_t)k;
        if (k < 0.0 || (us < 0.013 && v > us)) continue;
        double invalpha = 1.1239 + 1.1328 / (b - 3.4);
        double arg = v * invalpha / (a / (us * us) + b);
        double lhs = arg > 0.0 ? log(arg) : -INFINITY;
        double x = k + 1.0, rhs;
        if (x >= 10.0) {
            double xi = 1.0 / x, xi2 = xi * xi;
            double ser = xi * (0.083333333333333333 - xi2 * (0.0027777777777777778 - xi2 * (0.00079365079365079365 - xi2 * 0.00059523809523809524)));
            rhs = k * log(lam / x) - 0.5 * log(x) - lam + x - 0.91893853320467274 - ser;
        } else rhs = -lam + k * log(lam) - lgamma(x);
        if (lhs <= rhs) return (int64_t)k;
    }
}

/* round(x, 2) of a Python float: exact integer classification of m * 100 / 2^s (ties to even) */
static double round2(double x) {
    uint64_t bits; memcpy(&bits, &x, 8);
    int ex = (int)((bits >> 52) & 0x7ff);
    if (ex == 0x7ff) return x;
    double sgn = (bits >> 63) ? -1.0 : 1.0;
    if (ex == 0) return sgn * 0.0;
    uint64_t m = (bits & 0xFFFFFFFFFFFFFull) | (1ull << 52);
    int s = 1075 - ex;
    if (s <= 0) return x;
    if (s >= 61) return sgn * 0.0;
    uint64_t P = m * 100ull, q = P >> s, rem = P & ((1ull << s) - 1ull), half = 1ull << (s - 1);
    if (rem > half || (rem == half && (q & 1ull))) q += 1;
    return sgn * ((double)q / 100.0);
}

static double dist_i(int ax, int ay, int bx, int by) { double dx = ax - bx, dy = ay - by; return sqrt(dx * dx + dy * dy); }

rso_env* rso_create(uint32_t seed, uint32_t env_id, int A, int enforce, int falloff) {
    rso_env* e = (rso_env*)calloc(1, sizeof(rso_env));
    e->A = A; e->enforce = enforce; e->falloff = falloff; e->seed = seed; e->env_id = env_id;
    e->bx0 = 0; e->by0 = 0; e->bx1 = 2700; e->by1 = 2700;
    e->sa_x0 = 200; e->sa_y0 = 200; e->sa_x1 = 2200; e->sa_y1 = 2200;
    e->max_dist = dist_i(e->sa_x1, e->sa_y1, e->sa_x1, e->sa_y0);
    e->scale = 1.0 / (double)e->sa_y1;
    return e;
}
void rso_destroy(rso_env* e) { free(e); }

/* one env step; actions[a] in 0..8 or ACT_NONE.  obs [A][11] float64, reward [A], done [A] */
void rso_step(rso_env* e, const int* actions, double* obs, double* reward, double* team, int* done) {
    const int A = e->A;
    int px[MAXA], py[MAXA], none = 0;
    for (int a = 0; a < A; ++a) none |= actions[a] == ACT_NONE;
    for (int a = 0; a < A; ++a) { int s = none ? IDLE : actions[a]; px[a] = e->x[a] + STEP_DX[s]; py[a] = e->y[a] + STEP_DY[s]; }
    double max_reward = 0.0; int have = 0;
    for (int a = 0; a < A; ++a) {
        int act = actions[a], moved = 0;
        e->oob[a] = 0; e->coll[a] = 0;
        if (act != ACT_NONE) {
            int cnt = 0;
            if (A > 1) for (int j = 0; j < A; ++j) cnt += (px[j] == px[a] && py[j] == py[a]);
            if (cnt > 1) e->coll[a] = 1;
            else {
                int tx = e->x[a] + STEP_DX[act], ty = e->y[a] + STEP_DY[act], roll = 0;
                if (e->enforce) {
                    if ((tx < e->bx0 || ty < e->by0) || (e->bx1 <= tx || e->by1 <= ty)) { e->oob[a] = 1; e->oobc[a] += 1; roll = 1; }
                } else {
                    if ((e->x[a] < e->sa_x0 || e->y[a] < e->sa_y0) || (e->sa_x1 < e->x[a] || e->sa_y1 < e->y[a])) { e->oob[a] = 1; e->oobc[a] += 1; }
                }
                if (!roll) { e->x[a] = tx; e->y[a] = ty; moved = 1; }
            }
        }
        double euc = dist_i(e->x[a], e->y[a], e->src_x, e->src_y);
        if (moved) e->sp[a] = euc;
        double r = euc, rew;
        if (r == 0.0) { e->err |= 1u; r = 1.0; }
        double lam = (e->falloff ? (double)e->intensity / (r * r) : (double)e->intensity / r) + (double)e->bkg;
        int64_t meas = poisson(e, lam, a);
        if (moved) {
            if (e->sp[a] < 110.0) { rew = 0.1; e->done = 1; }
            else if (e->sp[a] < e->prev[a]) { rew = 0.1; e->prev[a] = e->sp[a]; }
            else rew = ((act == IDLE) ? -1.0 : -0.5) * e->sp[a] / e->max_dist;
        } else {
            if (act == IDLE && !e->coll[a]) e->err |= 2u;
            rew = -0.5 * e->sp[a] / e->max_dist;
        }
        rew = round2(rew);
        double* o = obs + a * 11;
        o[0] = (double)meas; o[1] = ((double)e->x[a] + 0.0) * e->scale; o[2] = ((double)e->y[a] + 0.0) * e->scale;
        for (int k = 3; k < 11; ++k) o[k] = 0.0;
        if (e->enforce) {
            int X = e->x[a], Y = e->y[a];
            if ((double)X - 110.0 < (double)e->bx0) o[3 + 0] = (110.0 - fabs((double)(X - e->bx0))) / 110.0;
            if ((double)Y - 110.0 < (double)e->by0) o[3 + 6] = (110.0 - fabs((double)(Y - e->by0))) / 110.0;
            if ((double)e->bx1 <= (double)X + 110.0) o[3 + 4] = (110.0 - fabs((double)(e->bx1 - X))) / 110.0;
            if ((double)e->by1 <= (double)Y + 110.0) o[3 + 2] = (110.0 - fabs((double)(e->by1 - Y))) / 110.0;
        }
        if (!have || max_reward == 0.0) { max_reward = rew; have = 1; } else if (max_reward < rew) max_reward = rew;
        reward[a] = rew; done[a] = e->done;
    }
    *team = max_reward;
    e->iter_count += 1; e->t += 1;
}

void rso_reset(rso_env* e, double* obs, double* reward, double* team, int* done) {
    for (int a = 0; a < e->A; ++a) { e->oobc[a] = 0; e->oob[a] = 0; }
    e->done = 0; e->iter_count = 0; e->t = 0; e->reset_idx = 0; e->cur_episode = e->episode;
    int sx = draw_int(e, e->sa_x0, e->sa_x1), sy = draw_int(e, e->sa_x0, e->sa_x1);
    int dx = draw_int(e, e->sa_x0, e->sa_x1), dy = draw_int(e, e->sa_x0, e->sa_x1);
    while (dist_i(dx, dy, sx, sy) < 1000.0) { sx = draw_int(e, e->sa_x0, e->sa_x1); sy = draw_int(e, e->sa_x0, e->sa_x1); }
    e->src_x = sx; e->src_y = sy;
    double prev = dist_i(sx, sy, dx, dy);
    for (int a = 0; a < e->A; ++a) { e->x[a] = dx; e->y[a] = dy; e->prev[a] = prev; e->sp[a] = prev; }
    e->intensity = draw_int(e, 1000000, 10000000);
    e->bkg = draw_int(e, 10, 51);
    int acts[MAXA];
    for (int a = 0; a < e->A; ++a) acts[a] = ACT_NONE;
    rso_step(e, acts, obs, reward, team, done);
    e->iter_count = 0;
    e->episode += 1;
}

/* accessors for tests */
void rso_state(const rso_env* e, int* xy /*[A][2]*/, double* sp, double* prev, int* misc /*src_x,src_y,intensity,bkg,done,err*/) {
    for (int a = 0; a < e->A; ++a) { xy[2 * a] = e->x[a]; xy[2 * a + 1] = e->y[a]; sp[a] = e->sp[a]; prev[a] = e->prev[a]; }
    misc[0] = e->src_x; misc[1] = e->src_y; misc[2] = e->intensity; misc[3] = e->bkg; misc[4] = e->done; misc[5] = (int)e->err;
}

/* bench loop: n_envs single-agent envs, uniform random actions from a xorshift stream, reset on done / L steps.
 * Runs until `target_steps` env steps were done; returns the number of steps. */
long rso_bench(uint32_t seed, uint32_t env_id0, int n_envs, long target_steps, int L) {
    rso_env** es = (rso_env**)malloc(sizeof(rso_env*) * n_envs);
    int* tin = (int*)calloc(n_envs, sizeof(int));
    double obs[11], rew, team; int done;
    for (int i = 0; i < n_envs; ++i) { es[i] = rso_create(seed, env_id0 + i, 1, 1, 0); rso_reset(es[i], obs, &rew, &team, &done); }
    uint64_t rs = 88172645463325252ull ^ env_id0;
    long steps = 0;
    double sink = 0.0;
    while (steps < target_steps) {
        for (int i = 0; i < n_envs; ++i) {
            rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
            int act = (int)(rs % 9);
            rso_step(es[i], &act, obs, &rew, &team, &done);
            sink += obs[0] + rew;
            steps += 1;
            if (done || ++tin[i] == L) { rso_reset(es[i], obs, &rew, &team, &done); tin[i] = 0; }
        }
    }
    for (int i = 0; i < n_envs; ++i) rso_destroy(es[i]);
    free(es); free(tin);
    return sink == 12345.678 ? -steps : steps;
}
