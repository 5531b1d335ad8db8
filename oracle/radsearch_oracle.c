/* CPU ORACLE in C (test infrastructure, NOT product code; built into oracle/_build/ by oracle/Makefile).
 *
 * Scalar C restatement of the radiation-search environment -- obstacle-free (BASELINE.json configs[1]) and, since round 3, with the
 * axis-aligned rectangular obstructions of configs[2..4] -- with the same Philox/PTRS streams as oracle/radsearch_oracle.py and the
 * HIP kernels.  It follows gym_rad_search/gym_rad_search/envs/rad_search_env.py (paths relative to /root/reference):
 *     get_step :178-224, take_action :876-946, agent_step :460-613, step :616-728, reset :730-797, create_obs :948-1011,
 *     sample_source_loc_pos :1013-1131, is_intersect :1133-1146, in_obstruction :1148-1170, obstruction_sensors :1172-1261,
 *     correct_coords :1263-1306; the visilibity calls (un-vendored dependency) as the exact-lattice predicates of the Python oracle.
 * Pinning: tests/test_oracle_c.py checks it event by event against the Python oracle (which is pinned to golden
 * vectors captured from the real reference; its obstacle geometry is "parity unpinned") -- float64-exact.
 * Its only other use is bench.py's cpu_baseline legs (kind "port"): the reference's env.step as compiled scalar code on the host cores.
 * Compile with -ffp-contract=off (see Makefile): the float64 arithmetic must round like Python's. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXA 8
#define IDLE 8
#define ACT_NONE 9
#define MAXO 7
#define ERR_CORRECT_CAP 4u
#define ERR_NO_PATH 16u

typedef struct { int x0, y0, x1, y1; } rso_rect;

typedef struct {
    int A, enforce, falloff;
    int obstruction_count, num_obs, epoch_end;        /* -1: U{1..5} per epoch; rectangles of the current epoch */
    rso_rect rects[MAXO];
    double dsrc[4 * MAXO];                            /* geodesic distance source -> rectangle vertex */
    int blocked[MAXA], inter[MAXA];
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

/* ---- exact lattice geometry: the visilibity calls of the env restated on integers (oracle/radsearch_oracle.py, "exact geometry") ---- */
typedef long long i64;
static int frac_lt(i64 n1, i64 d1, i64 n2, i64 d2) { return n1 * d2 < n2 * d1; }           /* n1/d1 < n2/d2, d > 0 */

/* closed segment p-q meets the OPEN interior of the rectangle? */
static int seg_hits_open_rect(int px, int py, int qx, int qy, const rso_rect* r) {
    i64 Ln[2], Ld[2], Un[2], Ud[2]; int nb = 0;
    int dx = qx - px, dy = qy - py;
    if (dx == 0) { if (!(r->x0 < px && px < r->x1)) return 0; }
    else if (dx > 0) { Ln[nb] = r->x0 - px; Ld[nb] = dx; Un[nb] = r->x1 - px; Ud[nb] = dx; nb++; }
    else { Ln[nb] = px - r->x1; Ld[nb] = -dx; Un[nb] = px - r->x0; Ud[nb] = -dx; nb++; }
    if (dy == 0) { if (!(r->y0 < py && py < r->y1)) return 0; }
    else if (dy > 0) { Ln[nb] = r->y0 - py; Ld[nb] = dy; Un[nb] = r->y1 - py; Ud[nb] = dy; nb++; }
    else { Ln[nb] = py - r->y1; Ld[nb] = -dy; Un[nb] = py - r->y0; Ud[nb] = -dy; nb++; }
    if (nb == 0) return 1;
    i64 ln = Ln[0], ld = Ld[0], un = Un[0], ud = Ud[0];
    if (nb == 2) {
        if (frac_lt(ln, ld, Ln[1], Ld[1])) { ln = Ln[1]; ld = Ld[1]; }
        if (frac_lt(Un[1], Ud[1], un, ud)) { un = Un[1]; ud = Ud[1]; }
    }
    return frac_lt(ln, ld, un, ud) && frac_lt(ln, ld, 1, 1) && frac_lt(0, 1, un, ud);
}
static int orient(int ax, int ay, int bx, int by, int cx, int cy) {
    i64 v = (i64)(bx - ax) * (cy - ay) - (i64)(by - ay) * (cx - ax);
    return (v > 0) - (v < 0);
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int on_seg(int ax, int ay, int bx, int by, int cx, int cy) {
    return imin(ax, bx) <= cx && cx <= imax(ax, bx) && imin(ay, by) <= cy && cy <= imax(ay, by);
}
static int segs_intersect_closed(int ax, int ay, int bx, int by, int cx, int cy, int dx, int dy) {
    int o1 = orient(ax, ay, bx, by, cx, cy), o2 = orient(ax, ay, bx, by, dx, dy);
    int o3 = orient(cx, cy, dx, dy, ax, ay), o4 = orient(cx, cy, dx, dy, bx, by);
    if (o1 != o2 && o3 != o4) return 1;
    if (o1 == 0 && on_seg(ax, ay, bx, by, cx, cy)) return 1;
    if (o2 == 0 && on_seg(ax, ay, bx, by, dx, dy)) return 1;
    if (o3 == 0 && on_seg(cx, cy, dx, dy, ax, ay)) return 1;
    if (o4 == 0 && on_seg(cx, cy, dx, dy, bx, by)) return 1;
    return 0;
}
/* edge order of the reference's line_segs (:1000-1005): (p0,p1),(p0,p3),(p2,p1),(p2,p3); vertex order of create_obs (:975-983) */
static void rect_edges(const rso_rect* r, int ed[4][4]) {
    const int x0 = r->x0, y0 = r->y0, x1 = r->x1, y1 = r->y1;
    const int t[4][4] = {{x0, y0, x0, y1}, {x0, y0, x1, y0}, {x1, y1, x0, y1}, {x1, y1, x1, y0}};
    memcpy(ed, t, sizeof(t));
}
static void rect_corners(const rso_rect* r, int c[4][2]) {
    const int t[4][2] = {{r->x0, r->y0}, {r->x0, r->y1}, {r->x1, r->y1}, {r->x1, r->y0}};
    memcpy(c, t, sizeof(t));
}
/* vis.boundary_distance(Line_Segment(p, q), rect) < 0.001, exact */
static int seg_rect_boundary_lt_1e3(int px, int py, int qx, int qy, const rso_rect* r) {
    int ed[4][4], co[4][2];
    rect_edges(r, ed);
    for (int k = 0; k < 4; ++k) if (segs_intersect_closed(px, py, qx, qy, ed[k][0], ed[k][1], ed[k][2], ed[k][3])) return 1;
    i64 dx = qx - px, dy = qy - py, len2 = dx * dx + dy * dy;
    if (len2 == 0) return 0;
    rect_corners(r, co);
    for (int k = 0; k < 4; ++k) {
        i64 dot = (co[k][0] - px) * dx + (co[k][1] - py) * dy;
        if (0 <= dot && dot <= len2) {
            i64 cr = (co[k][0] - px) * dy - (co[k][1] - py) * dx;
            /* cr^2 * 1e6 < len2 <= 1.5e7 needs |cr| < 4: test that first, cr^2 * 1e6 overflows 64 bits for large cr */
            if (cr > -4 && cr < 4 && cr * cr * 1000000 < len2) return 1;
        }
    }
    return 0;
}
static double dist_pt_axis_seg(int px, int py, int ax, int ay, int bx, int by) {
    int cx = imin(imax(px, imin(ax, bx)), imax(ax, bx)), cy = imin(imax(py, imin(ay, by)), imax(ay, by));
    double ddx = (double)((i64)(px - cx) * (px - cx)), ddy = (double)((i64)(py - cy) * (py - cy));
    return sqrt(ddx + ddy);
}
static int visible(const rso_env* e, int px, int py, int qx, int qy) {
    for (int o = 0; o < e->num_obs; ++o) if (seg_hits_open_rect(px, py, qx, qy, &e->rects[o])) return 0;
    return 1;
}
/* geodesic distance source -> every rectangle vertex: fixed point of the relaxation, sums rounded left to right */
static void source_vertex_dists(rso_env* e) {
    int vx[4 * MAXO], vy[4 * MAXO], n = 4 * e->num_obs;
    static const double INF = INFINITY;
    for (int o = 0; o < e->num_obs; ++o) { int co[4][2]; rect_corners(&e->rects[o], co); for (int k = 0; k < 4; ++k) { vx[4 * o + k] = co[k][0]; vy[4 * o + k] = co[k][1]; } }
    unsigned char adj[4 * MAXO][4 * MAXO]; double w[4 * MAXO][4 * MAXO];
    for (int i = 0; i < n; ++i) e->dsrc[i] = visible(e, e->src_x, e->src_y, vx[i], vy[i]) ? dist_i(e->src_x, e->src_y, vx[i], vy[i]) : INF;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            adj[i][j] = (i != j) && visible(e, vx[i], vy[i], vx[j], vy[j]);
            w[i][j] = adj[i][j] ? dist_i(vx[i], vy[i], vx[j], vy[j]) : 0.0;
        }
    for (int changed = 1; changed;) {
        changed = 0;
        for (int v = 0; v < n; ++v)
            for (int u = 0; u < n; ++u)
                if (adj[u][v] && e->dsrc[u] + w[u][v] < e->dsrc[v]) { e->dsrc[v] = e->dsrc[u] + w[u][v]; changed = 1; }
    }
}
/* world.shortest_path(source, detector).length() (:491-493) */
static double shortest_path_len(const rso_env* e, int px, int py) {
    if (visible(e, e->src_x, e->src_y, px, py)) return dist_i(e->src_x, e->src_y, px, py);
    double best = INFINITY;
    for (int o = 0; o < e->num_obs; ++o) {
        int co[4][2]; rect_corners(&e->rects[o], co);
        for (int k = 0; k < 4; ++k) {
            const double d = e->dsrc[4 * o + k];
            if (d < INFINITY && visible(e, co[k][0], co[k][1], px, py)) {
                const double c = d + dist_i(co[k][0], co[k][1], px, py);
                if (c < best) best = c;
            }
        }
    }
    return best;
}
static int pt_in_closed(int px, int py, const rso_rect* r) { return r->x0 <= px && px <= r->x1 && r->y0 <= py && py <= r->y1; }
static int pt_in_open(int px, int py, const rso_rect* r) { return r->x0 < px && px < r->x1 && r->y0 < py && py < r->y1; }
static double fmax3(double a, double b, double c) { double m = a > b ? a : b; return m > c ? m : c; }
static int pt_in_closed_eps(double qx, double qy, const rso_rect* r, double eps) {
    double dx = fmax3((double)r->x0 - qx, 0.0, qx - (double)r->x1), dy = fmax3((double)r->y0 - qy, 0.0, qy - (double)r->y1);
    return sqrt(dx * dx + dy * dy) <= eps;
}
static int rect_boundaries_touch(const rso_rect* r1, const rso_rect* r2) {                 /* :988 */
    int e1[4][4], e2[4][4];
    rect_edges(r1, e1); rect_edges(r2, e2);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (segs_intersect_closed(e1[i][0], e1[i][1], e1[i][2], e1[i][3], e2[j][0], e2[j][1], e2[j][2], e2[j][3])) return 1;
    return 0;
}
/* world.is_valid (:788) for rectangles that passed create_obs: no rectangle nested inside another (see the Python oracle) */
static int layout_is_valid(const rso_env* e) {
    for (int i = 0; i < e->num_obs; ++i)
        for (int k = 0; k < e->num_obs; ++k) {
            const rso_rect *r = &e->rects[i], *q = &e->rects[k];
            if (i != k && q->x0 <= r->x0 && r->x0 <= q->x1 && q->y0 <= r->y0 && r->y0 <= q->y1) return 0;
        }
    return 1;
}
static void create_obs(rso_env* e) {                                                        /* :948-1011 */
    int ii = 0;
    while (ii < e->num_obs) {
        int sx = draw_int(e, e->sa_x0, (int)((double)e->sa_x1 * 0.9)), sy = draw_int(e, e->sa_y0, (int)((double)e->sa_y1 * 0.9));
        int ex = draw_int(e, 200, 500), ey = draw_int(e, 200, 500);
        rso_rect r = {sx, sy, sx + ex, sy + ey};
        int hit = 0;
        for (int kk = 0; kk < ii && !hit; ++kk) hit = rect_boundaries_touch(&e->rects[kk], &r);
        if (!hit) e->rects[ii++] = r;
    }
}
static int is_intersect(const rso_env* e, int a, double euc) {                              /* :1133-1146 */
    for (int k = 0; k < e->num_obs; ++k) {
        if (seg_rect_boundary_lt_1e3(e->x[a], e->y[a], e->src_x, e->src_y, &e->rects[k])) {
            /* not math.isclose(sqrt(euc_dist), sp_dist, abs_tol=0.1) with the default rel_tol 1e-9 */
            const double s = sqrt(euc), b = e->sp[a];
            int close = 0;
            if (s == b) close = 1;
            else if (!(isinf(s) || isinf(b))) { const double diff = fabs(b - s); close = (diff <= fabs(1e-9 * b)) || (diff <= fabs(1e-9 * s)) || (diff <= 0.1); }
            if (!close) return 1;
        }
    }
    return 0;
}
static int in_obstruction(const rso_env* e, int x, int y) {                                 /* :1148-1170 */
    for (int j = 0; j < e->num_obs; ++j) if (pt_in_closed(x, y, &e->rects[j])) return pt_in_open(x, y, &e->rects[j]);
    return 0;
}
static const int DIR_CX[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, DIR_CY[8] = {0, 1, 1, 1, 0, -1, -1, -1};
static void correct_coords(rso_env* e, const rso_rect* r, int a, double dists[8]) {         /* :1263-1306 */
    int xc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, any = 0, it = 0;
    double qx[8], qy[8];
    for (int k = 0; k < 8; ++k) { qx[k] = (double)e->x[a]; qy[k] = (double)e->y[a]; dists[k] = 0.0; }
    while (!any) {
        for (int k = 0; k < 8; ++k) {
            qx[k] = qx[k] + (double)DIR_CX[k] * 0.1; qy[k] = qy[k] + (double)DIR_CY[k] * 0.1;
            if (pt_in_closed_eps(qx[k], qy[k], r, 0.0000001)) { xc[k] = 1; any = 1; }
        }
        if (++it >= 4096) { e->err |= ERR_CORRECT_CAP; break; }
    }
    int cnt = 0;
    for (int k = 0; k < 8; ++k) cnt += xc[k];
    if (cnt >= 4)
        for (int ii = 0; ii < 8; ii += 2)
            if (xc[(ii + 7) % 8] && xc[(ii + 1) % 8]) { dists[ii] = 1.0; dists[(ii + 7) % 8] = 1.0; dists[(ii + 1) % 8] = 1.0; }
}
static void obstruction_sensors(rso_env* e, int a, double dists[8]) {                        /* :1172-1261 */
    const int px = e->x[a], py = e->y[a];
    for (int k = 0; k < 8; ++k) dists[k] = 0.0;
    if (e->num_obs > 0) {
        int hits[MAXO] = {0, 0, 0, 0, 0, 0, 0}, inter = 0;
        double seg[4] = {0.0, 0.0, 0.0, 0.0};
        for (int idx = 0; idx < 8; ++idx) {
            const int qx = px + STEP_DX[idx], qy = py + STEP_DY[idx];
            for (int o = 0; o < e->num_obs; ++o) {
                int ed[4][4]; rect_edges(&e->rects[o], ed);
                for (int s = 0; s < 4; ++s)
                    if (inter < 2 && segs_intersect_closed(ed[s][0], ed[s][1], ed[s][2], ed[s][3], px, py, qx, qy)) {
                        seg[s] = (110.0 - dist_pt_axis_seg(px, py, ed[s][0], ed[s][1], ed[s][2], ed[s][3])) / 110.0;
                        inter += 1; hits[o] += 1;
                    }
                if (inter > 0) {
                    double m = seg[0];
                    for (int s = 1; s < 4; ++s) if (seg[s] > m) m = seg[s];
                    if (m > dists[idx]) dists[idx] = m;
                    seg[0] = seg[1] = seg[2] = seg[3] = 0.0;
                }
            }
            inter = 0;
        }
        int ones = 0;
        for (int k = 0; k < 8; ++k) ones += dists[k] == 1.0;
        if (ones > 3) {
            int best = 0;
            for (int k = 1; k < e->num_obs; ++k) {
                /* max(zip(counts, polygons)): the count, then the vertex lists (x0,y0,x0,y1,x1,y1,x1,y0) lexicographically */
                const rso_rect *p = &e->rects[k], *q = &e->rects[best];
                const int ka[9] = {hits[k], p->x0, p->y0, p->x0, p->y1, p->x1, p->y1, p->x1, p->y0};
                const int kb[9] = {hits[best], q->x0, q->y0, q->x0, q->y1, q->x1, q->y1, q->x1, q->y0};
                int gt = 0;
                for (int i = 0; i < 9; ++i) { if (ka[i] != kb[i]) { gt = ka[i] > kb[i]; break; } }
                if (gt) best = k;
            }
            correct_coords(e, &e->rects[best], a, dists);
        }
    }
    if (e->enforce) {
        const int X = e->x[a], Y = e->y[a];
        if ((double)X - 110.0 < (double)e->bx0) dists[0] = (110.0 - fabs((double)(X - e->bx0))) / 110.0;
        if ((double)Y - 110.0 < (double)e->by0) dists[6] = (110.0 - fabs((double)(Y - e->by0))) / 110.0;
        if ((double)e->bx1 <= (double)X + 110.0) dists[4] = (110.0 - fabs((double)(e->bx1 - X))) / 110.0;
        if ((double)e->by1 <= (double)Y + 110.0) dists[2] = (110.0 - fabs((double)(e->by1 - Y))) / 110.0;
    }
}

rso_env* rso_create2(uint32_t seed, uint32_t env_id, int A, int enforce, int falloff, int obstruction_count) {
    rso_env* e = (rso_env*)calloc(1, sizeof(rso_env));
    e->A = A; e->enforce = enforce; e->falloff = falloff; e->seed = seed; e->env_id = env_id;
    e->obstruction_count = obstruction_count; e->epoch_end = 1;
    e->bx0 = 0; e->by0 = 0; e->bx1 = 2700; e->by1 = 2700;
    e->sa_x0 = 200; e->sa_y0 = 200; e->sa_x1 = 2200; e->sa_y1 = 2200;
    e->max_dist = dist_i(e->sa_x1, e->sa_y1, e->sa_x1, e->sa_y0);
    e->scale = 1.0 / (double)e->sa_y1;
    return e;
}
rso_env* rso_create(uint32_t seed, uint32_t env_id, int A, int enforce, int falloff) { return rso_create2(seed, env_id, A, enforce, falloff, 0); }
void rso_destroy(rso_env* e) { free(e); }
void rso_set_epoch_end(rso_env* e) { e->epoch_end = 1; }

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
                if (in_obstruction(e, tx, ty)) { roll = 1; e->blocked[a] = 1; }             /* :935-937 */
                if (!roll) { e->x[a] = tx; e->y[a] = ty; moved = 1; }
            }
        }
        double euc = dist_i(e->x[a], e->y[a], e->src_x, e->src_y);
        if (moved) e->sp[a] = e->num_obs > 0 ? shortest_path_len(e, e->x[a], e->y[a]) : euc;
        if (!(e->sp[a] < INFINITY)) e->err |= ERR_NO_PATH;
        e->inter[a] = e->num_obs > 0 ? is_intersect(e, a, euc) : 0;
        double rew, lam;
        if (e->inter[a]) lam = (double)e->bkg;
        else {
            double r = euc;
            if (r == 0.0) { e->err |= 1u; r = 1.0; }
            lam = (e->falloff ? (double)e->intensity / (r * r) : (double)e->intensity / r) + (double)e->bkg;
        }
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
        if (e->num_obs > 0 || e->enforce) obstruction_sensors(e, a, o + 3);
        else for (int k = 3; k < 11; ++k) o[k] = 0.0;
        if (!have || max_reward == 0.0) { max_reward = rew; have = 1; } else if (max_reward < rew) max_reward = rew;
        reward[a] = rew; done[a] = e->done;
    }
    *team = max_reward;
    e->iter_count += 1; e->t += 1;
}

static void reset_inner(rso_env* e, int nested, double* obs, double* reward, double* team, int* done) {
    for (int a = 0; a < e->A; ++a) { e->oobc[a] = 0; e->oob[a] = 0; e->blocked[a] = 0; }
    e->done = 0; e->iter_count = 0;
    if (!nested) { e->t = 0; e->reset_idx = 0; e->cur_episode = e->episode; }
    if (e->epoch_end) {                                                                     /* :744-762 */
        e->num_obs = e->obstruction_count == -1 ? draw_int(e, 1, 6) : e->obstruction_count;
        create_obs(e);
        e->epoch_end = 0;
    }
    /* sample_source_loc_pos (:1013-1131); rand_point uses the x-range for both axes (:1033) */
    int sx = draw_int(e, e->sa_x0, e->sa_x1), sy = draw_int(e, e->sa_x0, e->sa_x1);
    int dx = draw_int(e, e->sa_x0, e->sa_x1), dy = draw_int(e, e->sa_x0, e->sa_x1);
    for (;;) {
        int inside = 0;
        for (int o = 0; o < e->num_obs && !inside; ++o) inside = pt_in_closed(dx, dy, &e->rects[o]);
        if (!inside) break;
        dx = draw_int(e, e->sa_x0, e->sa_x1); dy = draw_int(e, e->sa_x0, e->sa_x1);
    }
    for (int num_retry = 0;;) {
        while (dist_i(dx, dy, sx, sy) < 1000.0) { sx = draw_int(e, e->sa_x0, e->sa_x1); sy = draw_int(e, e->sa_x0, e->sa_x1); }
        int resamp = 0, inter = 0;
        for (int o = 0; o < e->num_obs && !resamp; ++o) {
            if (pt_in_closed(sx, sy, &e->rects[o])) resamp = 1;
            if (!resamp && seg_rect_boundary_lt_1e3(dx, dy, sx, sy, &e->rects[o])) inter = 1;
        }
        if (e->num_obs == 0 || (num_retry > 20 && !resamp)) break;
        else if (resamp || !inter) { sx = draw_int(e, e->sa_x0, e->sa_x1); sy = draw_int(e, e->sa_x0, e->sa_x1); num_retry += 1; }
        else break;
    }
    e->src_x = sx; e->src_y = sy;
    source_vertex_dists(e);
    double prev = e->num_obs > 0 ? shortest_path_len(e, dx, dy) : dist_i(sx, sy, dx, dy);
    for (int a = 0; a < e->A; ++a) { e->x[a] = dx; e->y[a] = dy; e->prev[a] = prev; e->sp[a] = prev; }
    e->intensity = draw_int(e, 1000000, 10000000);
    e->bkg = draw_int(e, 10, 51);
    if (!layout_is_valid(e)) {                        /* "Environment is not valid, retrying!" (:788-791): a full nested reset */
        e->epoch_end = 1;
        reset_inner(e, 1, obs, reward, team, done);
        for (int a = 0; a < e->A; ++a) e->sp[a] = e->prev[a];      /* the outer step(None) starts from iter_count == 0 again */
    }
    int acts[MAXA];
    for (int a = 0; a < e->A; ++a) acts[a] = ACT_NONE;
    rso_step(e, acts, obs, reward, team, done);
    e->iter_count = 0;
    if (!nested) e->episode += 1;
}
void rso_reset(rso_env* e, double* obs, double* reward, double* team, int* done) { reset_inner(e, 0, obs, reward, team, done); }

/* accessors for tests */
void rso_state(const rso_env* e, int* xy /*[A][2]*/, double* sp, double* prev, int* misc /*src_x,src_y,intensity,bkg,done,err*/) {
    for (int a = 0; a < e->A; ++a) { xy[2 * a] = e->x[a]; xy[2 * a + 1] = e->y[a]; sp[a] = e->sp[a]; prev[a] = e->prev[a]; }
    misc[0] = e->src_x; misc[1] = e->src_y; misc[2] = e->intensity; misc[3] = e->bkg; misc[4] = e->done; misc[5] = (int)e->err;
}
/* flags [A][5]: out_of_bounds, out_of_bounds_count, blocked, collision, intersect; rects [MAXO][4]; returns num_obs */
int rso_state2(const rso_env* e, int* flags, int* rects) {
    for (int a = 0; a < e->A; ++a) { int* f = flags + 5 * a; f[0] = e->oob[a]; f[1] = e->oobc[a]; f[2] = e->blocked[a]; f[3] = e->coll[a]; f[4] = e->inter[a]; }
    for (int o = 0; o < e->num_obs; ++o) { rects[4 * o] = e->rects[o].x0; rects[4 * o + 1] = e->rects[o].y0; rects[4 * o + 2] = e->rects[o].x1; rects[4 * o + 3] = e->rects[o].y1; }
    return e->num_obs;
}

/* bench loop: n_envs single-agent envs, uniform random actions from a xorshift stream, reset on done / L steps.
 * Runs until `target_steps` env steps were done; returns the number of steps. */
long rso_bench2(uint32_t seed, uint32_t env_id0, int n_envs, long target_steps, int L, int T, int obstruction_count, int A);
long rso_bench(uint32_t seed, uint32_t env_id0, int n_envs, long target_steps, int L) {
    return rso_bench2(seed, env_id0, n_envs, target_steps, L, 0, 0, 1);
}
/* the same with obstructions / several agents: a new layout every T lock-steps (train.py:482-484); T = 0: never */
long rso_bench2(uint32_t seed, uint32_t env_id0, int n_envs, long target_steps, int L, int T, int obstruction_count, int A) {
    rso_env** es = (rso_env**)malloc(sizeof(rso_env*) * n_envs);
    int* tin = (int*)calloc(n_envs, sizeof(int));
    double obs[11 * MAXA], rew[MAXA], team; int done[MAXA], act[MAXA];
    for (int i = 0; i < n_envs; ++i) { es[i] = rso_create2(seed, env_id0 + i, A, 1, 0, obstruction_count); rso_reset(es[i], obs, rew, &team, done); }
    uint64_t rs = 88172645463325252ull ^ env_id0;
    long steps = 0, lock = 0;
    double sink = 0.0;
    while (steps < target_steps) {
        for (int i = 0; i < n_envs; ++i) {
            for (int a = 0; a < A; ++a) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; act[a] = (int)(rs % 9); }
            rso_step(es[i], act, obs, rew, &team, done);
            sink += obs[0] + rew[0];
            steps += 1;
            if (done[0] || ++tin[i] == L) { rso_reset(es[i], obs, rew, &team, done); tin[i] = 0; }
        }
        if (T > 0 && ++lock % T == 0) for (int i = 0; i < n_envs; ++i) { rso_set_epoch_end(es[i]); rso_reset(es[i], obs, rew, &team, done); tin[i] = 0; }
    }
    uint32_t err = 0;
    for (int i = 0; i < n_envs; ++i) { err |= es[i]->err & ~2u; rso_destroy(es[i]); }
    free(es); free(tin);
    return (sink == 12345.678 || err) ? -steps : steps;
}
