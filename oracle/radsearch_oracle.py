"""CPU ORACLE (test infrastructure, NOT product code).

Plain-Python restatement of the reference's radiation-search environment hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the product package (radiation_ppo_amd/) never does.

Follows, function by function (paths relative to /root/reference):
  gym_rad_search/gym_rad_search/envs/rad_search_env.py
    :104-224   point helpers, get_step                      -> ACTION_STEP
    :259-301   Agent                                        -> OracleAgent
    :392-437   RadSearch.__post_init__                      -> RadSearchOracle.__init__
    :443-728   RadSearch.step / agent_step                  -> RadSearchOracle.step / _agent_step
    :730-797   RadSearch.reset                              -> RadSearchOracle.reset
    :876-946   take_action                                  -> _take_action
    :948-1011  create_obs                                   -> _create_obs
    :1013-1131 sample_source_loc_pos                        -> _sample_source_loc_pos
    :1133-1146 is_intersect                                 -> _is_intersect
    :1148-1170 in_obstruction                               -> _in_obstruction
    :1172-1261 obstruction_sensors                          -> _obstruction_sensors
    :1263-1306 correct_coords                               -> _correct_coords

Third-party arithmetic: the reference calls visilibity (peproctor/PyVisiLibity @
c76020079110231f882f38f61b3ab25d01de21f0, SWIG over VisiLibity1), which is NOT in
/root/reference and not installed.  Its published semantics for the calls the env makes
(Point::in, boundary_distance, intersect, distance, Environment::shortest_path) are restated
here with exact integer-lattice predicates on axis-aligned rectangles:
  * point-in-polygon with epsilon: closed containment (on-edge counts as "in")
  * segment/polygon boundary distance < 1e-3: exact rational test
  * shortest path: Euclidean geodesic in the closed free space (rect interiors removed),
    length summed source -> ... -> detector in float64, left to right (Polyline::length)

PARITY STATUS
  * obstacle-free behaviour (obstruction_count == 0): PINNED against tests/golden/env_*.npz,
    captured from the real reference with every RNG draw recorded (tests/golden/make_golden.py).
  * every obstacle-dependent quantity (in_obstruction, shortest path around holes, is_intersect,
    obstacle part of the sensors, obstacle placement): "parity unpinned" -- the reference holds no
    test or fixture for them and visilibity cannot be run here.

Randomness: the reference threads one numpy PCG64 generator through every env sequentially; a
lock-step batched env cannot reproduce that stream.  The oracle therefore takes a *draw source*:
ReplayDraws replays draws recorded from the reference (pins the deterministic maps), PhiloxDraws
is the counter-based stream the HIP kernels use (device-vs-oracle parity, bit-exact).
"""
import math

# --------------------------------------------------------------------------- constants
DET_STEP = 100.0          # rad_search_env.py:70
DET_STEP_FRAC = 71.0      # :71
DIST_TH = 110.0           # :72
EPSILON = 0.0000001       # :76
MIN_STARTING_DISTANCE = 1000  # :52
IDLE = 8
MAX_CREATION_TRIES = 1000000000  # :53
CORRECT_COORDS_ITER_CAP = 4096   # the reference loop (:1278) is unbounded; see DESIGN.md

# E1: get_step(action) (:178-224); values verified against the reference (tests/golden/action_table.npz)
ACTION_STEP = (
    (-100, 0), (-71, 71), (0, 100), (71, 71), (100, 0), (71, -71), (0, -100), (-71, -71), (0, 0),
)
# get_x_step_coeff/get_y_step_coeff (:186-202) for directions 0..7
DIR_COEFF = ((-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1))

ERR_ZERO_DIST = 1       # detector on the source: reference divides by zero (:501) and raises
ERR_IDLE_STALL = 2      # reference raises ValueError (:544-547, :562-565)
ERR_NO_PATH = 16        # shortest path is infinite (cannot happen in a world that passed is_valid)
ERR_CORRECT_CAP = 4     # correct_coords (:1278) did not terminate within the cap


# --------------------------------------------------------------------------- Philox4x32-10
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF
STREAM_RESET = 0
STREAM_STEP = 1           # + agent id
STREAM_GEOM = 64          # obstacle layouts shared by a group of envs (geom_group_size > 1)
STREAM_NOISE = 96         # + agent id: coordinate noise of the observation (coord_noise)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def u53(lo, hi):
    """Two 32-bit words -> double in [0,1) with 53 random bits."""
    return float(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)


def poisson_from_uniforms(lam, next_uv):
    """Poisson(lam).  lam >= 10: Hormann's PTRS transformed rejection (the algorithm numpy's
    Generator.poisson uses for lam >= 10); lam < 10: multiplication method.  next_uv(i) returns the
    i-th (U, V) pair of uniforms in [0,1)."""
    if lam == 0.0:
        return 0
    if lam < 10.0:
        enlam = math.exp(-lam)
        x = 0
        prod = 1.0
        i = 0
        while True:
            u, _ = next_uv(i)
            i += 1
            prod *= u
            if prod > enlam:
                x += 1
            else:
                return x
    slam = math.sqrt(lam)
    b = 0.931 + 2.53 * slam
    a = -0.059 + 0.02483 * b
    vr = 0.9277 - 3.6224 / (b - 2.0)
    i = 0
    while True:
        u, v = next_uv(i)
        i += 1
        u = u - 0.5
        us = 0.5 - abs(u)
        k = math.floor((2.0 * a / us + b) * u + lam + 0.43) if us > 0.0 else -1
        if us >= 0.07 and v <= vr:
            return int(k)
        if k < 0 or (us < 0.013 and v > us):
            continue
        # slow path: one log on the left; Stirling series of lgamma(x), x = k+1 >= 10, on the right (the same
        # formula, operation for operation, as rs_poisson in radiation_ppo_amd/csrc/rs_device.hpp)
        invalpha = 1.1239 + 1.1328 / (b - 3.4)
        arg = v * invalpha / (a / (us * us) + b)
        lhs = math.log(arg) if arg > 0.0 else -math.inf
        x = k + 1.0
        if x >= 10.0:
            xi = 1.0 / x
            xi2 = xi * xi
            ser = xi * (0.083333333333333333 - xi2 * (0.0027777777777777778 - xi2 * (0.00079365079365079365 - xi2 * 0.00059523809523809524)))
            rhs = k * math.log(lam / x) - 0.5 * math.log(x) - lam + x - 0.91893853320467274 - ser
        else:
            rhs = -lam + k * math.log(lam) - math.lgamma(x)
        if lhs <= rhs:
            return int(k)


class PhiloxDraws:
    """Counter-based draw source shared (by construction, not by code) with the HIP kernels.
    key = (seed, global env id); counter = (draw/attempt index, step index, episode index, stream)."""

    def __init__(self, seed, env_id):
        self.k0 = seed & _MASK
        self.k1 = env_id & _MASK
        self.episode = 0
        self.reset_idx = 0
        self.t = 0

    def begin_reset(self, episode):
        self.episode = episode & _MASK
        self.reset_idx = 0

    def begin_step(self, t):
        self.t = t & _MASK

    def integers(self, lo, hi):
        o = philox4x32_10(self.reset_idx & _MASK, 0, self.episode, STREAM_RESET, self.k0, self.k1)
        self.reset_idx += 1
        x = (o[1] << 32) | o[0]
        return lo + ((x * (hi - lo)) >> 64)

    def poisson(self, lam, agent):
        def next_uv(i):
            o = philox4x32_10(i & _MASK, self.t, self.episode, STREAM_STEP + agent, self.k0, self.k1)
            return u53(o[0], o[1]), u53(o[2], o[3])
        return poisson_from_uniforms(lam, next_uv)

    def normal2(self, scale, agent):
        """np_random.normal(scale=scale, size=2) (:570-574): one Philox block of stream 96 + agent -> Box-Muller in float64."""
        o = philox4x32_10(0, self.t, self.episode, STREAM_NOISE + agent, self.k0, self.k1)
        u1 = 1.0 - u53(o[0], o[1])
        u2 = u53(o[2], o[3])
        rad = scale * math.sqrt(-2.0 * math.log(u1))
        return rad * math.cos(6.283185307179586 * u2), rad * math.sin(6.283185307179586 * u2)


class PhiloxGeomDraws:
    """Draws for an obstacle layout shared by a group of envs: key = (seed, first env id of the
    group), counter = (draw index, 0, geometry epoch, STREAM_GEOM)."""

    def __init__(self, seed, first_env_id):
        self.k0 = seed & _MASK
        self.k1 = first_env_id & _MASK
        self.epoch = 0
        self.idx = 0

    def begin(self, geom_epoch):
        self.epoch = geom_epoch & _MASK
        self.idx = 0

    def integers(self, lo, hi):
        o = philox4x32_10(self.idx & _MASK, 0, self.epoch, STREAM_GEOM, self.k0, self.k1)
        self.idx += 1
        x = (o[1] << 32) | o[0]
        return lo + ((x * (hi - lo)) >> 64)


def layout_is_valid(rects):
    """world.is_valid(EPSILON) (:788) for lattice rectangles that passed create_obs.

    VisiLibity1 (un-vendored dependency, Environment::is_valid in visilibity.cpp) requires: every polygon simple;
    no two boundaries within epsilon; every hole vertex inside the outer boundary and in NO other hole; outer
    boundary counter-clockwise, holes clockwise.  create_obs emits clockwise rectangles of extent >= 200 strictly
    inside the walls with pairwise disjoint boundaries, so the one clause that can fail is "a vertex of hole i is
    in hole k": one rectangle nested inside another (boundaries disjoint => one vertex inside means all four)."""
    for i, r in enumerate(rects):
        for k, q in enumerate(rects):
            if i != k and q[0] <= r[0] <= q[2] and q[1] <= r[1] <= q[3]:
                return False
    return True


def sample_layout(draws, obstruction_count, sa_x0=200, sa_y0=200, sa_x1=2200, sa_y1=2200, oa=(200, 500)):
    """create_obs (:948-1011) + the num_obs draw (:745-750) as a free function (shared layouts), redrawn until
    world.is_valid (:788) holds."""
    while True:
        num = draws.integers(1, 6) if obstruction_count == -1 else obstruction_count
        rects = []
        while len(rects) < num:
            sx = draws.integers(sa_x0, int(sa_x1 * 0.9))
            sy = draws.integers(sa_y0, int(sa_y1 * 0.9))
            ex = draws.integers(oa[0], oa[1])
            ey = draws.integers(oa[0], oa[1])
            r = (sx, sy, sx + ex, sy + ey)
            if not any(RadSearchOracle._rect_boundaries_touch(q, r) for q in rects):
                rects.append(r)
        if layout_is_valid(rects):
            return rects


class ReplayDraws:
    """Replays draws recorded from the reference's numpy Generator, checking the arguments."""

    def __init__(self, rows):
        self.rows = list(rows)   # (kind, a0, a1, value); kind 0 = integers, 1 = poisson
        self.pos = 0

    def begin_reset(self, episode):
        pass

    def begin_step(self, t):
        pass

    def integers(self, lo, hi):
        kind, a0, a1, v = self.rows[self.pos]
        self.pos += 1
        assert kind == 0 and a0 == lo and a1 == hi, ("integers args", (kind, a0, a1), (lo, hi))
        return int(v)

    def poisson(self, lam, agent):
        kind, a0, _, v = self.rows[self.pos]
        self.pos += 1
        assert kind == 1 and a0 == lam, ("poisson lam", a0, lam)
        return int(v)

    def normal2(self, scale, agent):
        out = []
        for _ in range(2):
            kind, a0, a1, v = self.rows[self.pos]
            self.pos += 1
            assert kind == 2 and a0 == 0.0 and a1 == scale, ("normal args", (kind, a0, a1), scale)
            out.append(float(v))
        return tuple(out)


# --------------------------------------------------------------------------- exact geometry
def round2(x):
    """round(x, 2) of a Python float (rad_search_env.py:613): correctly rounded decimal rounding.
    (The HIP kernel restates this with an exact fma residual; here Python's own round is the spec.)"""
    return round(x, 2)


def isclose_abs(a, b, abs_tol):
    """math.isclose(a, b, abs_tol=abs_tol) with the default rel_tol=1e-9 (:1141-1143)."""
    if a == b:
        return True
    if math.isinf(a) or math.isinf(b):
        return False
    diff = abs(b - a)
    return (diff <= abs(1e-9 * b)) or (diff <= abs(1e-9 * a)) or (diff <= abs_tol)


def pt_in_closed(px, py, r):
    """visilibity Point::in(poly, eps) for a lattice point: inside or on the boundary."""
    return r[0] <= px <= r[2] and r[1] <= py <= r[3]


def pt_in_closed_eps(qx, qy, r, eps):
    """Point::in for a float point: inside, or within eps of the boundary."""
    dx = max(r[0] - qx, 0.0, qx - r[2])
    dy = max(r[1] - qy, 0.0, qy - r[3])
    return math.sqrt(dx * dx + dy * dy) <= eps


def pt_in_open(px, py, r):
    return r[0] < px < r[2] and r[1] < py < r[3]


def _frac_lt(n1, d1, n2, d2):
    """n1/d1 < n2/d2 with d1, d2 > 0 (integers)."""
    return n1 * d2 < n2 * d1


def seg_hits_open_rect(px, py, qx, qy, r):
    """Does the closed segment p-q meet the OPEN interior of rectangle r?  Exact on integers."""
    x0, y0, x1, y1 = r
    # lower/upper bounds of the parameter t as fractions (num, den>0)
    lo_n, lo_d = 0, 1      # t > lo  (starts as t >= 0, handled below)
    hi_n, hi_d = 1, 1
    lo_strict_from_zero = False
    L = []  # lower bounds (open)
    U = []  # upper bounds (open)
    dx = qx - px
    dy = qy - py
    if dx == 0:
        if not (x0 < px < x1):
            return False
    elif dx > 0:
        L.append((x0 - px, dx)); U.append((x1 - px, dx))
    else:
        L.append((px - x1, -dx)); U.append((px - x0, -dx))
    if dy == 0:
        if not (y0 < py < y1):
            return False
    elif dy > 0:
        L.append((y0 - py, dy)); U.append((y1 - py, dy))
    else:
        L.append((py - y1, -dy)); U.append((py - y0, -dy))
    if not L:
        return True   # degenerate segment strictly inside
    # max of lowers, min of uppers
    ln, ld = L[0]
    for (n, d) in L[1:]:
        if _frac_lt(ln, ld, n, d):
            ln, ld = n, d
    un, ud = U[0]
    for (n, d) in U[1:]:
        if _frac_lt(n, d, un, ud):
            un, ud = n, d
    # exists t in [0,1] with L < t < U  <=>  L < U and L < 1 and U > 0
    return _frac_lt(ln, ld, un, ud) and _frac_lt(ln, ld, 1, 1) and _frac_lt(0, 1, un, ud)


def _orient(ax, ay, bx, by, cx, cy):
    v = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    return (v > 0) - (v < 0)


def _on_seg(ax, ay, bx, by, cx, cy):
    return min(ax, bx) <= cx <= max(ax, bx) and min(ay, by) <= cy <= max(ay, by)


def segs_intersect_closed(ax, ay, bx, by, cx, cy, dx, dy):
    """vis.intersect(ls1, ls2, eps) on the integer lattice: closed segments share a point."""
    o1 = _orient(ax, ay, bx, by, cx, cy)
    o2 = _orient(ax, ay, bx, by, dx, dy)
    o3 = _orient(cx, cy, dx, dy, ax, ay)
    o4 = _orient(cx, cy, dx, dy, bx, by)
    if o1 != o2 and o3 != o4:
        return True
    if o1 == 0 and _on_seg(ax, ay, bx, by, cx, cy):
        return True
    if o2 == 0 and _on_seg(ax, ay, bx, by, dx, dy):
        return True
    if o3 == 0 and _on_seg(cx, cy, dx, dy, ax, ay):
        return True
    if o4 == 0 and _on_seg(cx, cy, dx, dy, bx, by):
        return True
    return False


def rect_edges(r):
    """Edge order of the reference's line_segs (:1000-1005): (p0,p1),(p0,p3),(p2,p1),(p2,p3)."""
    x0, y0, x1, y1 = r
    return ((x0, y0, x0, y1), (x0, y0, x1, y0), (x1, y1, x0, y1), (x1, y1, x1, y0))


def rect_corners(r):
    """Vertex order of create_obs (:975-983)."""
    x0, y0, x1, y1 = r
    return ((x0, y0), (x0, y1), (x1, y1), (x1, y0))


def seg_rect_boundary_lt_1e3(px, py, qx, qy, r):
    """vis.boundary_distance(Line_Segment(p,q), rect) < 0.001, exact (threshold^2 = 1e-6)."""
    for (ax, ay, bx, by) in rect_edges(r):
        if segs_intersect_closed(px, py, qx, qy, ax, ay, bx, by):
            return True
    dx = qx - px
    dy = qy - py
    len2 = dx * dx + dy * dy
    if len2 == 0:
        return False   # a lattice point not touching the boundary is >= 1 away
    for (cx, cy) in rect_corners(r):
        dot = (cx - px) * dx + (cy - py) * dy
        if 0 <= dot <= len2:
            cr = (cx - px) * dy - (cy - py) * dx
            if cr * cr * 1000000 < len2:
                return True
    return False


def dist_pt_axis_seg(px, py, ax, ay, bx, by):
    """vis.distance(Point, Line_Segment) for an axis-aligned lattice segment."""
    cx = min(max(px, min(ax, bx)), max(ax, bx))
    cy = min(max(py, min(ay, by)), max(ay, by))
    return math.sqrt(float((px - cx) ** 2) + float((py - cy) ** 2))


def dist_i(ax, ay, bx, by):
    """dist_p (:125-136)."""
    return math.sqrt(float((ax - bx) ** 2) + float((ay - by) ** 2))


def visible(px, py, qx, qy, rects):
    for r in rects:
        if seg_hits_open_rect(px, py, qx, qy, r):
            return False
    return True


def source_vertex_dists(sx, sy, rects):
    """Geodesic distance source -> every rectangle vertex (inf if unreachable): fixed point of
    d[v] = min(|s-v| if visible, min_u d[u] + |u-v| if visible), sums rounded left to right."""
    verts = [c for r in rects for c in rect_corners(r)]
    n = len(verts)
    d = [dist_i(sx, sy, vx, vy) if visible(sx, sy, vx, vy, rects) else math.inf for (vx, vy) in verts]
    adj = [[False] * n for _ in range(n)]
    w = [[0.0] * n for _ in range(n)]
    for i in range(n):
        for j in range(n):
            if i != j and visible(verts[i][0], verts[i][1], verts[j][0], verts[j][1], rects):
                adj[i][j] = True
                w[i][j] = dist_i(verts[i][0], verts[i][1], verts[j][0], verts[j][1])
    changed = True
    while changed:
        changed = False
        for v in range(n):
            for u in range(n):
                if adj[u][v] and d[u] + w[u][v] < d[v]:
                    d[v] = d[u] + w[u][v]
                    changed = True
    return d


def shortest_path_len(sx, sy, px, py, rects, dsrc):
    """world.shortest_path(source, detector).length() (:491-493)."""
    if visible(sx, sy, px, py, rects):
        return dist_i(sx, sy, px, py)
    best = math.inf
    i = 0
    for r in rects:
        for (vx, vy) in rect_corners(r):
            if dsrc[i] < math.inf and visible(vx, vy, px, py, rects):
                c = dsrc[i] + dist_i(vx, vy, px, py)
                if c < best:
                    best = c
            i += 1
    return best


# --------------------------------------------------------------------------- the environment
class OracleAgent:
    """rad_search_env.py:259-301 (render-only fields dropped)."""

    def __init__(self, id):
        self.id = id
        self.sp_dist = 0.0
        self.euc_dist = 0.0
        self.det = (0, 0)            # det_coords
        self.detector = (0, 0)       # visilibity shadow of the tentative position
        self.out_of_bounds = False
        self.out_of_bounds_count = 0
        self.collision = False
        self.intersect = False
        self.obstacle_blocking = False
        self.prev_det_dist = 0.0

    def reset(self):                 # :292-301
        self.obstacle_blocking = False
        self.out_of_bounds = False
        self.out_of_bounds_count = 0


class RadSearchOracle:
    def __init__(self, draws, number_agents=1, obstruction_count=0, enforce_grid_boundaries=False,
                 bbox=(0, 0, 2700, 2700), observation_area=(200, 500), falloff="reference", layout_fn=None,
                 coord_noise=False, DEBUG=False):
        self.coord_noise = coord_noise  # :365, :569-580
        self.DEBUG = DEBUG              # :387-389: hard-coded spawn
        self.layout_fn = layout_fn      # callable() -> rects for shared layouts (geom_group_size > 1)
        self.rng = draws
        self.number_agents = number_agents
        self.obstruction_count = obstruction_count
        self.enforce_grid_boundaries = enforce_grid_boundaries
        self.bbox = bbox
        self.observation_area = observation_area
        self.falloff = falloff
        # search_area (:393-420): corners 0=(lo,lo) 1=(hi_x,lo) 2=(hi_x,hi_y) 3=(lo,hi_y)
        self.sa_x0 = bbox[0] + observation_area[0]
        self.sa_y0 = bbox[1] + observation_area[0]
        self.sa_x1 = bbox[2] - observation_area[1]
        self.sa_y1 = bbox[3] - observation_area[1]
        self.epoch_end = True
        self.agents = {i: OracleAgent(i) for i in range(number_agents)}
        # :423-425 max_dist = dist(search_area[2], search_area[1]) (a side, not the diagonal)
        self.max_dist = dist_i(self.sa_x1, self.sa_y1, self.sa_x1, self.sa_y0)
        assert self.max_dist > 1000
        self.scale = 1 / float(self.sa_y1)       # :435
        self.done = False
        self.iter_count = 0
        self.num_obs = 0
        self.rects = []
        self.dsrc = []
        self.episode = 0
        self.t = 0
        self.err = 0
        self.invalid_layouts = 0     # layouts rejected by world.is_valid (test bookkeeping)
        self.last_lam = [0.0] * number_agents
        self._ret = self.reset()

    # ------------------------------------------------------------------ step (:443-728)
    def step(self, action=None):
        if type(action) is int:
            if action == -1:
                action = 8
            assert 0 <= action <= 8
        elif type(action) is dict:
            for a in action.values():
                assert 0 <= a <= 8
        else:
            assert action is None
        action_list = action if type(action) is dict else None

        self.rng.begin_step(self.t)
        obs, rew, done, info = {}, {}, {}, {}
        max_reward = None
        if action_list:
            proposed = [(self.agents[i].det[0] + ACTION_STEP[a][0], self.agents[i].det[1] + ACTION_STEP[a][1])
                        for i, a in action_list.items()]
            order = list(action_list.items())
        else:
            proposed = []
            order = [(i, action) for i in self.agents]
        for i, a in order:
            obs[i], rew[i], done[i], info[i] = self._agent_step(a, self.agents[i], proposed)
            # team reward with the falsy-reset quirk (:662-665)
            if not max_reward:
                max_reward = rew[i]
            elif max_reward < rew[i]:
                max_reward = rew[i]
        self.iter_count += 1
        self.t += 1
        return obs, {"team_reward": max_reward, "individual_reward": rew}, done, info

    def _lam(self, agent):
        if agent.intersect:
            return float(self.bkg_intensity)
        r = agent.euc_dist
        if r == 0.0:
            self.err |= ERR_ZERO_DIST
            r = 1.0
        if self.falloff == "inverse_square":
            return self.intensity / (r * r) + self.bkg_intensity
        return self.intensity / r + self.bkg_intensity       # :501 (1/r as written)

    def _agent_step(self, action, agent, proposed):
        agent.out_of_bounds = False
        agent.collision = False
        if self._take_action(agent, action, proposed):
            agent.sp_dist = shortest_path_len(self.src[0], self.src[1], agent.det[0], agent.det[1],
                                              self.rects, self.dsrc)
            agent.euc_dist = dist_i(agent.det[0], agent.det[1], self.src[0], self.src[1])
            agent.intersect = self._is_intersect(agent)
            lam = self._lam(agent)
            measurement = self.rng.poisson(lam, agent.id)
            if agent.sp_dist < 110:
                reward = 0.1
                self.done = True
            elif agent.sp_dist < agent.prev_det_dist:
                reward = 0.1
                agent.prev_det_dist = agent.sp_dist
            else:
                if action == IDLE:
                    reward = -1.0 * agent.sp_dist / self.max_dist
                else:
                    reward = -0.5 * agent.sp_dist / self.max_dist
        else:
            if self.iter_count > 0:
                agent.intersect = self._is_intersect(agent)
                lam = self._lam(agent)
                measurement = self.rng.poisson(lam, agent.id)
                if action == IDLE and not agent.collision:
                    self.err |= ERR_IDLE_STALL
                reward = -0.5 * agent.sp_dist / self.max_dist
            else:
                agent.sp_dist = agent.prev_det_dist
                agent.euc_dist = dist_i(agent.det[0], agent.det[1], self.src[0], self.src[1])
                agent.intersect = self._is_intersect(agent)
                lam = self._lam(agent)
                measurement = self.rng.poisson(lam, agent.id)
                if action == IDLE and not agent.collision:
                    self.err |= ERR_IDLE_STALL
                reward = -0.5 * agent.sp_dist / self.max_dist
        self.last_lam[agent.id] = lam
        if not agent.sp_dist < math.inf:
            self.err |= ERR_NO_PATH
        # observation (:569-593)
        s = 1 / float(self.sa_y1)
        noise = self.rng.normal2(5, agent.id) if self.coord_noise else (0.0, 0.0)
        ox = (agent.det[0] + noise[0]) * s
        oy = (agent.det[1] + noise[1]) * s
        if self.num_obs > 0 or self.enforce_grid_boundaries:
            sensors = self._obstruction_sensors(agent)
        else:
            sensors = [0.0] * 8
        state = [float(measurement), ox, oy] + sensors
        info = {"out_of_bounds": agent.out_of_bounds, "out_of_bounds_count": agent.out_of_bounds_count,
                "blocked": agent.obstacle_blocking, "scale": s}
        return state, round2(reward), self.done, info

    # ------------------------------------------------------------------ reset (:730-797)
    def reset(self, _nested=False):
        for agent in self.agents.values():
            agent.reset()
        self.done = False
        self.iter_count = 0
        if not _nested:          # a nested retry (:788-791) continues the same draw sequence and step counter
            self.rng.begin_reset(self.episode)
            self.t = 0
        if self.epoch_end and self.layout_fn is not None:
            self.rects = list(self.layout_fn())
            self.num_obs = len(self.rects)
            self.epoch_end = False
        if self.epoch_end:
            if self.obstruction_count == -1:
                self.num_obs = self.rng.integers(1, 6)
            elif self.obstruction_count == 0:
                self.num_obs = 0
            else:
                self.num_obs = self.obstruction_count
            self._create_obs()
            self.epoch_end = False
        self.src, det = self._sample_source_loc_pos()
        self.dsrc = source_vertex_dists(self.src[0], self.src[1], self.rects)
        for agent in self.agents.values():
            agent.detector = det
            agent.det = det
            agent.prev_det_dist = shortest_path_len(self.src[0], self.src[1], det[0], det[1], self.rects, self.dsrc)
        self.intensity = self.rng.integers(1000000, 10000000)     # :778 (1e6, 10e6)
        self.bkg_intensity = self.rng.integers(10, 51)            # :779
        if self.DEBUG:                                            # :782-785
            self.intensity, self.bkg_intensity = 1000000, 0
        # :788-791 "Environment is not valid, retrying!": a full nested reset with a new layout, after which the
        # outer call still runs its own step(None) -- k rejected layouts cost k extra idle measurements.
        if self.layout_fn is None and not layout_is_valid(self.rects):
            self.epoch_end = True
            self.invalid_layouts += 1
            self.reset(_nested=True)
        ret = self.step(None)
        self.iter_count = 0
        if not _nested:
            self.episode += 1
        return ret

    # ------------------------------------------------------------------ refresh_environment (:799-874)
    def refresh_environment(self, src, det, intensity, bkg, rects=None):
        """Start an episode from saved parameters.  rects=None keeps the current layout (num_obs = 0 default).
        sp_dist is left stale exactly as the reference leaves it: step(None) copies the OLD prev_det_dist (:562)
        before prev_det_dist is recomputed for the new geometry (:866-868)."""
        self.epoch_end = False
        self.done = False
        self.iter_count = 0
        self.rng.begin_reset(self.episode)          # a fresh draw episode, like reset (the device does the same)
        self.t = 0
        self.src = (int(src[0]), int(src[1]))
        self.intensity = int(intensity)
        self.bkg_intensity = int(bkg)
        for agent in self.agents.values():
            agent.reset()
            agent.det = (int(det[0]), int(det[1]))
            agent.detector = agent.det
        if rects is not None:                       # :829-858
            self.rects = [tuple(int(v) for v in r) for r in rects]
            self.num_obs = len(self.rects)
        self.dsrc = source_vertex_dists(self.src[0], self.src[1], self.rects)
        ret = self.step(None)                       # :860
        for agent in self.agents.values():          # :866-868
            agent.prev_det_dist = shortest_path_len(self.src[0], self.src[1], agent.det[0], agent.det[1], self.rects, self.dsrc)
        self.iter_count = 1                         # :870
        self.episode += 1
        return ret[0]

    # ------------------------------------------------------------------ take_action (:876-946)
    def _take_action(self, agent, action, proposed):
        if action is None:
            return False
        roll_back = False
        st = ACTION_STEP[action]
        tent = (agent.det[0] + st[0], agent.det[1] + st[1])
        cnt = 0
        for p in proposed:
            if p[0] == tent[0] and p[1] == tent[1]:
                cnt += 1
        if cnt > 1:
            agent.collision = True
            return False
        agent.detector = tent
        if self.enforce_grid_boundaries:
            if (tent[0] < self.bbox[0] or tent[1] < self.bbox[1]) or (self.bbox[2] <= tent[0] or self.bbox[3] <= tent[1]):
                agent.out_of_bounds = True
                agent.out_of_bounds_count += 1
                roll_back = True
        else:
            lower_b = agent.det[0] < self.sa_x0 or agent.det[1] < self.sa_y0
            upper_b = self.sa_x1 < agent.det[0] or self.sa_y1 < agent.det[1]
            if lower_b or upper_b:
                agent.out_of_bounds = True
                agent.out_of_bounds_count += 1
        if self._in_obstruction(agent):
            roll_back = True
            agent.obstacle_blocking = True
        if roll_back:
            agent.detector = agent.det
        else:
            agent.det = agent.detector
        return not roll_back

    # ------------------------------------------------------------------ create_obs (:948-1011)
    def _create_obs(self):
        self.rects = []
        ii = 0
        while ii < self.num_obs:
            seed_x = self.rng.integers(self.sa_x0, int(self.sa_x1 * 0.9))
            seed_y = self.rng.integers(self.sa_y0, int(self.sa_y1 * 0.9))
            ext_x = self.rng.integers(self.observation_area[0], self.observation_area[1])
            ext_y = self.rng.integers(self.observation_area[0], self.observation_area[1])
            r = (seed_x, seed_y, seed_x + ext_x, seed_y + ext_y)
            intersect = False
            kk = 0
            while not intersect and kk < ii:
                intersect = self._rect_boundaries_touch(self.rects[kk], r)
                kk += 1
            if not intersect:
                self.rects.append(r)
                ii += 1

    @staticmethod
    def _rect_boundaries_touch(r1, r2):
        """isclose(boundary_distance(poly1, poly2), 0, abs_tol=1e-7) (:988): the boundaries share a
        point.  Lattice rectangles whose boundaries do not meet are >= 1 apart."""
        for (ax, ay, bx, by) in rect_edges(r1):
            for (cx, cy, dx, dy) in rect_edges(r2):
                if segs_intersect_closed(ax, ay, bx, by, cx, cy, dx, dy):
                    return True
        return False

    # ------------------------------------------------------------------ sample_source_loc_pos (:1013-1131)
    def _rand_point(self):
        # integers(int(search_area[0][0]), int(search_area[1][0]), size=2): x-range for both (:1033)
        x = self.rng.integers(self.sa_x0, self.sa_x1)
        y = self.rng.integers(self.sa_x0, self.sa_x1)
        return (x, y)

    def _sample_source_loc_pos(self):
        source = self._rand_point()
        if self.DEBUG:                       # :1043-1044 (the draw above is still consumed)
            source = (500, 500)
        detector = self._rand_point()
        if self.DEBUG:                       # :1052-1053
            detector = (1000, 1000)
        det_clear = False
        while not det_clear:
            resamp = False
            for r in self.rects:
                if pt_in_closed(detector[0], detector[1], r):
                    resamp = True
                    break
            if resamp:
                detector = self._rand_point()
            else:
                det_clear = True
        src_clear = self.DEBUG               # :1087-1088: DEBUG skips the minimum-distance / line-of-sight resampling
        resamp = False
        inter = False
        num_retry = 0
        while not src_clear:
            while dist_i(detector[0], detector[1], source[0], source[1]) < MIN_STARTING_DISTANCE:
                source = self._rand_point()
            obstacle_index = 0
            while not resamp and obstacle_index < self.num_obs:
                r = self.rects[obstacle_index]
                if pt_in_closed(source[0], source[1], r):
                    resamp = True
                if not resamp and seg_rect_boundary_lt_1e3(detector[0], detector[1], source[0], source[1], r):
                    inter = True
                obstacle_index += 1
            if self.num_obs == 0 or (num_retry > 20 and not resamp):
                src_clear = True
            elif resamp or not inter:
                source = self._rand_point()
                resamp = False
                inter = False
                num_retry += 1
            elif inter:
                src_clear = True
        return source, detector

    # ------------------------------------------------------------------ is_intersect (:1133-1146)
    def _is_intersect(self, agent):
        inter = False
        kk = 0
        while not inter and kk < self.num_obs:
            if seg_rect_boundary_lt_1e3(agent.detector[0], agent.detector[1], self.src[0], self.src[1], self.rects[kk]) \
                    and not isclose_abs(math.sqrt(agent.euc_dist), agent.sp_dist, 0.1):
                inter = True
            kk += 1
        return inter

    # ------------------------------------------------------------------ in_obstruction (:1148-1170)
    def _in_obstruction(self, agent):
        jj = 0
        obs_boundary = False
        while not obs_boundary and jj < self.num_obs:
            if pt_in_closed(agent.detector[0], agent.detector[1], self.rects[jj]):
                obs_boundary = True
            jj += 1
        if obs_boundary:
            return pt_in_open(agent.detector[0], agent.detector[1], self.rects[jj - 1])
        return False

    # ------------------------------------------------------------------ obstruction_sensors (:1172-1261)
    def _obstruction_sensors(self, agent):
        px, py = agent.detector
        dists = [0.0] * 8
        obs_idx_ls = [0] * self.num_obs
        inter = 0
        seg_dist = [0.0] * 4
        if self.num_obs > 0:
            for idx in range(8):
                qx, qy = px + ACTION_STEP[idx][0], py + ACTION_STEP[idx][1]
                for obs_idx, r in enumerate(self.rects):
                    for seg_idx, (ax, ay, bx, by) in enumerate(rect_edges(r)):
                        if inter < 2 and segs_intersect_closed(ax, ay, bx, by, px, py, qx, qy):
                            obstacle_distance = dist_pt_axis_seg(px, py, ax, ay, bx, by)
                            seg_dist[seg_idx] = (DIST_TH - obstacle_distance) / DIST_TH
                            inter += 1
                            obs_idx_ls[obs_idx] += 1
                    if inter > 0:
                        m = max(seg_dist)
                        if m > dists[idx]:
                            dists[idx] = m
                        seg_dist = [0.0] * 4
                inter = 0
            if sum(1 for x in dists if x == 1.0) > 3:
                best = 0
                for k in range(1, self.num_obs):
                    # max(zip(obs_idx_ls, self.poly)): count, then the vertex lists lexicographically
                    ka = (obs_idx_ls[k],) + self._poly_key(self.rects[k])
                    kb = (obs_idx_ls[best],) + self._poly_key(self.rects[best])
                    if ka > kb:
                        best = k
                dists = self._correct_coords(self.rects[best], agent)
        if self.enforce_grid_boundaries:
            dx, dy = agent.det
            if dx - DIST_TH < self.bbox[0]:
                dists[0] = (DIST_TH - abs(dx - self.bbox[0])) / DIST_TH
            if dy - DIST_TH < self.bbox[1]:
                dists[6] = (DIST_TH - abs(dy - self.bbox[1])) / DIST_TH
            if self.bbox[2] <= dx + DIST_TH:
                dists[4] = (DIST_TH - abs(self.bbox[2] - dx)) / DIST_TH
            if self.bbox[3] <= dy + DIST_TH:
                dists[2] = (DIST_TH - abs(self.bbox[3] - dy)) / DIST_TH
        return [float(d) for d in dists]

    @staticmethod
    def _poly_key(r):
        # [(x0,y0),(x0,y1),(x1,y1),(x1,y0)] compared lexicographically
        return (r[0], r[1], r[0], r[3], r[2], r[3], r[2], r[1])

    # ------------------------------------------------------------------ correct_coords (:1263-1306)
    def _correct_coords(self, r, agent):
        x_check = [False] * 8
        qs = [(float(agent.detector[0]), float(agent.detector[1]))] * 8
        dists = [0.0] * 8
        it = 0
        while not any(x_check):
            for a in range(8):
                step = (DIR_COEFF[a][0] * 0.1, DIR_COEFF[a][1] * 0.1)
                qs[a] = (qs[a][0] + step[0], qs[a][1] + step[1])
                if pt_in_closed_eps(qs[a][0], qs[a][1], r, EPSILON):
                    x_check[a] = True
            it += 1
            if it >= CORRECT_COORDS_ITER_CAP:
                self.err |= ERR_CORRECT_CAP
                break
        if sum(x_check) >= 4:
            for ii in (0, 2, 4, 6):
                if x_check[ii - 1] and x_check[(ii + 1) % 8]:
                    dists[ii] = 1.0
                    dists[ii - 1] = 1.0
                    dists[ii + 1] = 1.0
        return dists


# --------------------------------------------------------------------------- PPO-side oracles
def discount_cumsum(x, discount):
    """ppo.py:62-85: scipy.signal.lfilter([1],[1,-discount], x[::-1])[::-1] in float64, i.e.
    y[t] = x[t] + (discount * y[t+1]) with one rounding for the product and one for the sum."""
    y = [0.0] * len(x)
    acc = 0.0
    for t in range(len(x) - 1, -1, -1):
        acc = float(x[t]) + float(discount) * acc
        y[t] = acc
    return y


def gae_and_rtg(rew_f32, val_f32, last_val, gamma, lam):
    """PPOBuffer.GAE_advantage_and_rewardsToGO (ppo.py:391-423) for ONE trajectory slice.
    rew/val are the float32 buffer contents; returns float64 (adv, ret) before the float32 store."""
    rews = [float(r) for r in rew_f32] + [float(last_val)]
    vals = [float(v) for v in val_f32] + [float(last_val)]
    n = len(rew_f32)
    deltas = [rews[t] + gamma * vals[t + 1] - vals[t] for t in range(n)]
    adv = discount_cumsum(deltas, gamma * lam)
    r2g = discount_cumsum(rews, gamma)[:-1]
    return adv, r2g


class WelfordOracle:
    """StatisticStandardization (RADTEAM_core.py:188-277)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.mean = 0.0
        self.sq = 0.0
        self.std = 1.0
        self.count = 0

    def update(self, x):
        self.count += 1
        if self.count == 1:
            self.mean = x
        else:
            mu_n = self.mean + (x - self.mean) / self.count
            self.sq = self.sq + (x - self.mean) * (x - mu_n)
            self.mean = mu_n
            self.std = max(math.sqrt(self.sq / (self.count - 1)), 1)

    def standardize(self, x):
        return (x - self.mean) / self.std
