"""A second opinion on the obstacle geometry, whose parity with the reference stays UNPINNED (the reference calls the
un-vendored VisiLibity1 binding for it -- rad_search_env.py:491-493, :1133-1146 -- and holds no test of it).

The oracle's exact integer predicates (oracle/radsearch_oracle.py: visible / seg_hits_open_rect, seg_rect_boundary_lt_1e3,
source_vertex_dists + shortest_path_len) are compared with brute-force implementations WRITTEN DIFFERENTLY, in exact rational
arithmetic (fractions.Fraction): segment clipping against the closed rectangle + midpoint-in-open-interior test instead of
the parametric open-interval test; true segment-segment distances instead of the corner-projection shortcut; a full
visibility graph over {source, detector, every rectangle vertex} with Dijkstra instead of cached source->vertex geodesics
plus one final hop.  Layouts are random lattice rectangles drawn like create_obs (:948-1011) and the query points are biased
towards the degenerate cases: rectangle corners, points on edges, points collinear with edges, grazing segments.
This does not change the 'parity unpinned' status: it checks the restatement against itself by an
independent route."""
import heapq
import math
from fractions import Fraction as F

import numpy as np

from oracle.radsearch_oracle import (RadSearchOracle, rect_corners, seg_rect_boundary_lt_1e3, shortest_path_len,
                                     source_vertex_dists, visible)


# ------------------------------------------------------------------ brute force, exact rationals
def bf_hits_open_rect(p, q, r):
    """closed segment p-q meets the OPEN rectangle: clip the segment to the CLOSED rectangle (Liang-Barsky in Fractions);
    the clipped piece lies in the interior iff its midpoint does (convexity); a degenerate piece is a boundary touch."""
    x0, y0, x1, y1 = r
    if p == q:
        return x0 < p[0] < x1 and y0 < p[1] < y1
    t0, t1 = F(0), F(1)
    d = (q[0] - p[0], q[1] - p[1])
    for dk, pk, lo, hi in ((d[0], p[0], x0, x1), (d[1], p[1], y0, y1)):
        if dk == 0:
            if pk < lo or pk > hi:
                return False
        else:
            ta, tb = F(lo - pk, dk), F(hi - pk, dk)
            if ta > tb:
                ta, tb = tb, ta
            t0, t1 = max(t0, ta), min(t1, tb)
            if t0 > t1:
                return False
    if t0 == t1:
        return False
    tm = (t0 + t1) / 2
    mx, my = p[0] + tm * d[0], p[1] + tm * d[1]
    return x0 < mx < x1 and y0 < my < y1


def bf_pt_seg_d2(c, a, b):
    ax, ay, bx, by = a[0], a[1], b[0], b[1]
    dx, dy = bx - ax, by - ay
    l2 = dx * dx + dy * dy
    if l2 == 0:
        return F((c[0] - ax) ** 2 + (c[1] - ay) ** 2)
    t = F((c[0] - ax) * dx + (c[1] - ay) * dy, l2)
    t = min(max(t, F(0)), F(1))
    ex, ey = ax + t * dx - c[0], ay + t * dy - c[1]
    return ex * ex + ey * ey


def bf_segs_meet(a, b, c, d):
    """closed segments share a point: solve a + s (b-a) = c + t (d-c) over the rationals; parallel case by projection."""
    r = (b[0] - a[0], b[1] - a[1]); s_ = (d[0] - c[0], d[1] - c[1])
    den = r[0] * s_[1] - r[1] * s_[0]
    ca = (c[0] - a[0], c[1] - a[1])
    if den != 0:
        s = F(ca[0] * s_[1] - ca[1] * s_[0], den)
        t = F(ca[0] * r[1] - ca[1] * r[0], den)
        return 0 <= s <= 1 and 0 <= t <= 1
    if ca[0] * r[1] - ca[1] * r[0] != 0:
        return False                                   # parallel, not collinear
    return min(bf_pt_seg_d2(a, c, d), bf_pt_seg_d2(b, c, d), bf_pt_seg_d2(c, a, b), bf_pt_seg_d2(d, a, b)) == 0


def bf_seg_seg_d2(a, b, c, d):
    if bf_segs_meet(a, b, c, d):
        return F(0)
    return min(bf_pt_seg_d2(a, c, d), bf_pt_seg_d2(b, c, d), bf_pt_seg_d2(c, a, b), bf_pt_seg_d2(d, a, b))


def bf_boundary_lt_1e3(p, q, r):
    x0, y0, x1, y1 = r
    corners = [(x0, y0), (x0, y1), (x1, y1), (x1, y0)]
    return min(bf_seg_seg_d2(p, q, corners[i], corners[(i + 1) % 4]) for i in range(4)) < F(1, 10 ** 6)


def bf_shortest_path(src, det, rects):
    nodes = [src, det] + [c for r in rects for c in rect_corners(r)]
    n = len(nodes)
    free = lambda u, v: not any(bf_hits_open_rect(nodes[u], nodes[v], r) for r in rects)
    dist = [math.inf] * n
    dist[0] = 0.0
    pq = [(0.0, 0)]
    seen = [False] * n
    cache = {}
    while pq:
        d, u = heapq.heappop(pq)
        if seen[u]:
            continue
        seen[u] = True
        if u == 1:
            return d
        for v in range(n):
            if v == u or seen[v]:
                continue
            key = (min(u, v), max(u, v))
            if key not in cache:
                cache[key] = free(u, v)
            if cache[key]:
                nd = d + math.dist(nodes[u], nodes[v])
                if nd < dist[v]:
                    dist[v] = nd
                    heapq.heappush(pq, (nd, v))
    return math.inf


# ------------------------------------------------------------------ layouts and degenerate query points
def _layout(rng, n):
    rects = []
    while len(rects) < n:
        sx, sy = int(rng.integers(200, 1980)), int(rng.integers(200, 1980))
        r = (sx, sy, sx + int(rng.integers(200, 500)), sy + int(rng.integers(200, 500)))
        if not any(RadSearchOracle._rect_boundaries_touch(q, r) for q in rects):
            rects.append(r)
    return rects


def _points(rng, rects, k):
    pts = [(int(rng.integers(200, 2200)), int(rng.integers(200, 2200))) for _ in range(k)]
    for r in rects:
        x0, y0, x1, y1 = r
        pts += [(x0, y0), (x1, y1)]                                                  # corners
        pts += [(int(rng.integers(x0, x1 + 1)), y0), (x1, int(rng.integers(y0, y1 + 1)))]   # on an edge
        pts += [(x0, int(rng.integers(0, 2700))), (int(rng.integers(0, 2700)), y1)]  # collinear with an edge line
        pts += [(x0 - 1, y0 - 1), (x1 + 1, y1)]                                     # one lattice step off a corner / edge
    return pts


def test_visibility_and_line_of_sight_predicates_vs_rational_brute_force():
    rng = np.random.default_rng(2024)
    n_vis = n_los = blocked = close = 0
    for _ in range(2500):
        rects = _layout(rng, int(rng.integers(1, 8)))
        pts = _points(rng, rects, 6)
        for _ in range(10):
            p, q = pts[int(rng.integers(len(pts)))], pts[int(rng.integers(len(pts)))]
            want = not any(bf_hits_open_rect(p, q, r) for r in rects)
            assert visible(p[0], p[1], q[0], q[1], rects) == want, (p, q, rects)
            n_vis += 1; blocked += not want
            r = rects[int(rng.integers(len(rects)))]
            want = bf_boundary_lt_1e3(p, q, r)
            assert seg_rect_boundary_lt_1e3(p[0], p[1], q[0], q[1], r) == want, (p, q, r)
            n_los += 1; close += want
    assert n_vis == 25000 and blocked > 2000 and close > 2000 and close < n_los - 2000     # both outcomes well covered


def test_near_miss_segments_against_the_1e3_threshold():
    """Segments passing a rectangle corner at distances around 1e-3 (cross product 0, 1, 2, 3 lattice units over lengths
    up to 3900 cm): the exact comparison cr^2 * 1e6 < len2 must agree with the rational distance."""
    rng = np.random.default_rng(7)
    hits = 0
    for _ in range(4000):
        x0, y0 = int(rng.integers(400, 1800)), int(rng.integers(400, 1800))
        r = (x0, y0, x0 + int(rng.integers(200, 500)), y0 + int(rng.integers(200, 500)))
        cx, cy = rect_corners(r)[int(rng.integers(4))]
        dx, dy = int(rng.integers(-1900, 1900)), int(rng.integers(-1900, 1900))
        if dx == 0 and dy == 0:
            continue
        k = int(rng.integers(0, 4))                    # offset the line by cross product k: distance k / |d|
        # choose p so that (c - p) x d == k: p = c - a d - k * n / (n . perp) is not integral in general; search a lattice p nearby
        a = int(rng.integers(1, 3))
        p = (cx - a * dx // 3, cy - a * dy // 3)
        q = (p[0] + dx, p[1] + dy)
        for px in range(p[0] - 2, p[0] + 3):
            pp, qq = (px, p[1]), (px + dx, q[1])
            want = bf_boundary_lt_1e3(pp, qq, r)
            assert seg_rect_boundary_lt_1e3(pp[0], pp[1], qq[0], qq[1], r) == want, (pp, qq, r)
            hits += want
    assert hits > 500


def test_shortest_path_vs_full_visibility_graph_dijkstra():
    rng = np.random.default_rng(99)
    detours = unreachable = 0
    for _ in range(700):
        rects = _layout(rng, int(rng.integers(1, 8)))
        pts = _points(rng, rects, 4)
        inside = lambda p: any(r[0] < p[0] < r[2] and r[1] < p[1] < r[3] for r in rects)
        src = pts[int(rng.integers(len(pts)))]
        if inside(src):
            continue
        dsrc = source_vertex_dists(src[0], src[1], rects)
        for _ in range(3):
            det = pts[int(rng.integers(len(pts)))]
            if inside(det):
                continue
            got = shortest_path_len(src[0], src[1], det[0], det[1], rects, dsrc)
            want = bf_shortest_path(src, det, rects)
            if math.isinf(want):
                assert math.isinf(got)
                unreachable += 1
                continue
            assert abs(got - want) <= 1e-9 * max(1.0, want), (src, det, rects, got, want)
            detours += want > math.dist(src, det) + 1e-9
    assert detours > 150
