#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (read-only at /root/reference).

This script only runs in the build container (the reference never travels to the GPU box);
its outputs under tests/golden/*.npz are plain data: inputs and expected outputs.

The reference's Python files are imported unmodified.  Four modules it imports are not
installed in this image (gym, visilibity, ray, mpi4py).  They are replaced by semantics-free
placeholders created in a temp directory (never committed):

  * gym        -- Env base class, spaces.Box/Discrete, registration.register, seeding names
  * ray        -- empty module (imported by ppo.py, never used)
  * mpi4py.MPI -- 1-rank communicator (Allreduce == copy); the reference's multi-agent
                  entry point hard-disables MPI anyway (main.py:457-461)
  * visilibity -- ONLY valid for obstruction_count == 0: the world is the convex outer
                  wall, so Environment.shortest_path(a, b).length() IS the Euclidean
                  distance; every obstacle-dependent call raises.  Obstacle-dependent
                  quantities therefore stay "parity unpinned" (SURVEY.md section 8c).

What is captured (SURVEY.md section 8c "Golden vectors"):
  env_*.npz   transitions of RadSearch.step/reset with every RNG draw recorded
  gae.npz     PPOBuffer.GAE_advantage_and_rewardsToGO + get() advantage normalisation
  ff_core.npz FF_core.ActorCritic forward (weights + inputs + outputs)
  welford.npz StatisticStandardization traces
  round2.npz  Python round(x, 2) on the env's reward expressions
"""
import os
import sys
import tempfile
import textwrap

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_placeholders() -> str:
    d = tempfile.mkdtemp(prefix="rs_placeholders_")

    def w(rel, src):
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(src))

    w("gym/__init__.py", """
        from . import spaces, envs, utils
        class Env:
            pass
        def make(*a, **k):
            raise NotImplementedError
    """)
    w("gym/spaces.py", """
        class Box:
            def __init__(self, low, high, shape=None, dtype=None):
                self.low, self.high, self.shape, self.dtype = low, high, shape, dtype
        class Discrete:
            def __init__(self, n):
                self.n = n
    """)
    w("gym/envs/__init__.py", "from . import registration\n")
    w("gym/envs/registration.py", "def register(*a, **k):\n    pass\n")
    w("gym/utils/__init__.py", "from . import seeding\n")
    w("gym/utils/seeding.py", """
        def _int_list_from_bigint(*a, **k):
            raise NotImplementedError
        def hash_seed(*a, **k):
            raise NotImplementedError
    """)
    w("ray/__init__.py", "")
    w("mpi4py/__init__.py", "from . import MPI\n")
    w("mpi4py/MPI.py", """
        SUM = "sum"; MIN = "min"; MAX = "max"
        class _Comm:
            def Get_rank(self): return 0
            def Get_size(self): return 1
            def Allreduce(self, x, buff, op=None):
                buff[...] = x
            def Bcast(self, x, root=0):
                pass
        COMM_WORLD = _Comm()
    """)
    w("visilibity.py", """
        # Obstacle-free placeholder: the only polygon is the convex outer wall.
        import math
        class Point:
            def __init__(self, x=0.0, y=0.0):
                self._x, self._y = x, y
            def x(self): return self._x
            def y(self): return self._y
            def _in(self, *a):
                raise NotImplementedError("obstacle geometry is not available in the placeholder")
        class Polygon:
            def __init__(self, pts): self.pts = list(pts)
            def bbox(self): raise NotImplementedError
        class Line_Segment:
            def __init__(self, a, b): self.a, self.b = a, b
            def first(self): return self.a
            def second(self): return self.b
        class _Polyline:
            def __init__(self, l): self._l = l
            def length(self): return self._l
        class Environment:
            def __init__(self, polys):
                assert len(polys) == 1, "placeholder supports obstruction_count == 0 only"
            def is_valid(self, eps): return True
            def shortest_path(self, a, b, graph, eps):
                return _Polyline(math.sqrt(float((a.x() - b.x()) ** 2) + float((a.y() - b.y()) ** 2)))
        class Visibility_Graph:
            def __init__(self, env, eps): pass
        def boundary_distance(*a): raise NotImplementedError
        def intersect(*a): raise NotImplementedError
        def distance(*a): raise NotImplementedError
    """)
    sys.path.insert(0, d)
    return d


class RecordingGenerator:
    """Wraps numpy.random.Generator and logs every draw the env makes."""

    def __init__(self, seed):
        self._g = np.random.default_rng(seed)
        self.log = []  # (kind, arg0, arg1, values...)

    def integers(self, low, high=None, size=None, **kw):
        v = self._g.integers(low, high, size=size, **kw)
        vals = np.atleast_1d(np.asarray(v)).astype(np.int64).tolist()
        self.log.append(("integers", float(low), float(high), vals))
        return v

    def poisson(self, lam=1.0, size=None):
        v = self._g.poisson(lam, size)
        self.log.append(("poisson", float(lam), 0.0, [int(v)]))
        return v

    def normal(self, loc=0.0, scale=1.0, size=None):
        v = self._g.normal(loc, scale, size)
        self.log.append(("normal", float(loc), float(scale), np.atleast_1d(v).tolist()))
        return v


def _snapshot(env, A):
    ag = [env.agents[i] for i in range(A)]
    return dict(
        det=[[float(a.det_coords[0]), float(a.det_coords[1])] for a in ag],
        sp=[float(a.sp_dist) for a in ag],
        euc=[float(a.euc_dist) for a in ag],
        prev=[float(a.prev_det_dist) for a in ag],
        oob=[bool(a.out_of_bounds) for a in ag],
        oobc=[int(a.out_of_bounds_count) for a in ag],
        coll=[bool(a.collision) for a in ag],
        blocked=[bool(a.obstacle_blocking) for a in ag],
        inter=[bool(a.intersect) for a in ag],
        done=bool(env.done),
        iter_count=int(env.iter_count),
    )


def _greedy_action(env, aid):
    """Scripted policy helper: head for the source (forces the terminal branch)."""
    from gym_rad_search.envs import rad_search_env as R
    a = env.agents[aid]
    best, bd = 8, None
    for act in range(8):
        st = R.get_step(act)
        d = (a.det_coords[0] + st[0] - env.src_coords[0]) ** 2 + (a.det_coords[1] + st[1] - env.src_coords[1]) ** 2
        if bd is None or d < bd:
            best, bd = act, d
    return best


def gen_env_scenarios():
    from gym_rad_search.envs.rad_search_env import RadSearch, get_step
    # E1 action table
    table = np.array([get_step(a) for a in range(9)], dtype=np.float64)
    np.savez(os.path.join(OUT, "action_table.npz"), table=table)

    for seed in (0, 1, 2, 289714752):
        for A in (1, 2, 4):
            for enforce in (True, False):
                rec = RecordingGenerator(seed)
                env = RadSearch(number_agents=A, np_random=rec, obstruction_count=0,
                                enforce_grid_boundaries=enforce)
                rows = []   # one per event (reset or step)
                script = np.random.default_rng(1000 + seed % 97 + A)

                def record(kind, actions, ret, log_start):
                    obs, rew, done, info = ret
                    snap = _snapshot(env, A)
                    draws = rec.log[log_start:]
                    rows.append(dict(
                        kind=kind, actions=actions,
                        obs=[np.asarray(obs[i], dtype=np.float64).tolist() for i in range(A)],
                        reward=[float(rew["individual_reward"][i]) for i in range(A)],
                        team=float("nan") if rew["team_reward"] is None else float(rew["team_reward"]),
                        done_ret=[bool(done[i]) for i in range(A)],
                        info_oob=[bool(info[i]["out_of_bounds"]) for i in range(A)],
                        info_oobc=[int(info[i]["out_of_bounds_count"]) for i in range(A)],
                        info_blocked=[bool(info[i]["blocked"]) for i in range(A)],
                        src=[float(env.src_coords[0]), float(env.src_coords[1])],
                        intensity=int(env.intensity), bkg=int(env.bkg_intensity),
                        draws=draws, **snap))

                # the constructor already performed one reset; redo it under recording
                ls = len(rec.log)
                env.epoch_end = True
                ret = env.reset()
                record("reset", [8] * A, ret, ls)
                steps_in_ep = 0
                phase = 0
                for t in range(260):
                    # phases: random walk / wall seeking / greedy approach / idle+collisions
                    phase = (t // 40) % 5
                    if phase == 0:
                        acts = [int(script.integers(0, 9)) for _ in range(A)]
                    elif phase == 1:
                        acts = [[0, 6, 7, 5][i % 4] for i in range(A)]       # run into left/bottom walls
                    elif phase == 2:
                        acts = [_greedy_action(env, i) for i in range(A)]    # terminal approach
                    elif phase == 3:
                        acts = [[4, 2, 3, 1][i % 4] for i in range(A)]       # run into right/top walls
                    else:
                        acts = [8 if (t + i) % 3 == 0 else int(script.integers(0, 8)) for i in range(A)]
                    ls = len(rec.log)
                    if A == 1 and t % 7 == 3:
                        # single-int calling convention (rad_search_env.py:620-623,676-690)
                        ret = env.step(acts[0] if acts[0] != 8 else -1)
                    else:
                        ret = env.step({i: acts[i] for i in range(A)})
                    record("step", acts, ret, ls)
                    steps_in_ep += 1
                    if env.done or steps_in_ep == 30:
                        ls = len(rec.log)
                        if t % 2 == 0:
                            env.epoch_end = True
                        ret = env.reset()
                        record("reset", [8] * A, ret, ls)
                        steps_in_ep = 0
                _save_env_rows(os.path.join(OUT, f"env_s{seed}_a{A}_e{int(enforce)}.npz"), rows, A,
                               dict(seed=seed, A=A, enforce=int(enforce)))


def gen_env_callforms():
    """The calling conventions of RadSearch.step besides the dict form (rad_search_env.py:616-627, :676-690) for
    SEVERAL agents: `step(int)` gives every agent the same action WITHOUT the collision rule (agent_step is called with
    no proposed_coordinates) and `step(None)` re-measures in place (:528-549).  Same row format as env_*.npz plus
    `form` (0 dict, 1 single int, 2 None)."""
    from gym_rad_search.envs.rad_search_env import RadSearch
    for A in (2, 3):
        for enforce in (True, False):
            rec = RecordingGenerator(41 + A)
            env = RadSearch(number_agents=A, np_random=rec, obstruction_count=0, enforce_grid_boundaries=enforce)
            rows, forms = [], []
            script = np.random.default_rng(500 + A)

            def record(kind, actions, ret, log_start, form):
                obs, rew, done, info = ret
                rows.append(dict(
                    kind=kind, actions=actions,
                    obs=[np.asarray(obs[i], dtype=np.float64).tolist() for i in range(A)],
                    reward=[float(rew["individual_reward"][i]) for i in range(A)],
                    team=float("nan") if rew["team_reward"] is None else float(rew["team_reward"]),
                    done_ret=[bool(done[i]) for i in range(A)],
                    info_oob=[bool(info[i]["out_of_bounds"]) for i in range(A)],
                    info_oobc=[int(info[i]["out_of_bounds_count"]) for i in range(A)],
                    info_blocked=[bool(info[i]["blocked"]) for i in range(A)],
                    src=[float(env.src_coords[0]), float(env.src_coords[1])],
                    intensity=int(env.intensity), bkg=int(env.bkg_intensity),
                    draws=rec.log[log_start:], **_snapshot(env, A)))
                forms.append(form)

            ls = len(rec.log)
            env.epoch_end = True
            record("reset", [8] * A, env.reset(), ls, 0)
            steps_in_ep = 0
            for t in range(120):
                ls = len(rec.log)
                m = t % 6
                if m in (1, 4):                       # single int: all agents share one cell afterwards, no collision stall
                    a = int(script.integers(0, 8)) if m == 1 else -1
                    ret = env.step(a)
                    record("step", [8 if a == -1 else a] * A, ret, ls, 1)
                elif m == 3:                          # None: no move, fresh measurement, stale distances
                    ret = env.step(None)
                    record("step", [9] * A, ret, ls, 2)
                else:
                    acts = [int(script.integers(0, 9)) for _ in range(A)]
                    ret = env.step({i: acts[i] for i in range(A)})
                    record("step", acts, ret, ls, 0)
                steps_in_ep += 1
                if env.done or steps_in_ep == 25:
                    ls = len(rec.log)
                    record("reset", [8] * A, env.reset(), ls, 0)
                    steps_in_ep = 0
            path = os.path.join(OUT, f"envforms_a{A}_e{int(enforce)}.npz")
            _save_env_rows(path, rows, A, dict(seed=41 + A, A=A, enforce=int(enforce)))
            d = dict(np.load(path).items())
            d["form"] = np.array(forms, dtype=np.int8)
            np.savez_compressed(path, **d)


def gen_env_options():
    """coord_noise=True (rad_search_env.py:365, :569-580: N(0, 5) on the observation's coordinates, drawn per agent-step AFTER the
    measurement) and DEBUG=True (:387-389, :782-785, :1043-1090: hard-coded source / detector / intensities, no spawn resampling),
    both from the reference itself with every draw recorded.  Same row format as env_*.npz; meta = [seed, A, enforce, noise, debug]."""
    from gym_rad_search.envs.rad_search_env import RadSearch
    for name, A, kw in (("envopt_noise_a1", 1, dict(coord_noise=True)), ("envopt_noise_a2", 2, dict(coord_noise=True)),
                        ("envopt_debug_a1", 1, dict(DEBUG=True))):
        rec = RecordingGenerator(97 + A)
        env = RadSearch(number_agents=A, np_random=rec, obstruction_count=0, enforce_grid_boundaries=True, **kw)
        rows = []
        script = np.random.default_rng(600 + A)

        def record(kind, actions, ret, log_start):
            obs, rew, done, info = ret
            rows.append(dict(
                kind=kind, actions=actions,
                obs=[np.asarray(obs[i], dtype=np.float64).tolist() for i in range(A)],
                reward=[float(rew["individual_reward"][i]) for i in range(A)],
                team=float("nan") if rew["team_reward"] is None else float(rew["team_reward"]),
                done_ret=[bool(done[i]) for i in range(A)],
                info_oob=[bool(info[i]["out_of_bounds"]) for i in range(A)],
                info_oobc=[int(info[i]["out_of_bounds_count"]) for i in range(A)],
                info_blocked=[bool(info[i]["blocked"]) for i in range(A)],
                src=[float(env.src_coords[0]), float(env.src_coords[1])],
                intensity=int(env.intensity), bkg=int(env.bkg_intensity),
                draws=rec.log[log_start:], **_snapshot(env, A)))

        ls = len(rec.log)
        env.epoch_end = True
        record("reset", [8] * A, env.reset(), ls)
        steps_in_ep = 0
        for t in range(90):
            acts = [_greedy_action(env, i) if (t // 15) % 2 else int(script.integers(0, 9)) for i in range(A)]
            ls = len(rec.log)
            ret = env.step({i: acts[i] for i in range(A)})
            record("step", acts, ret, ls)
            steps_in_ep += 1
            if env.done or steps_in_ep == 25:
                ls = len(rec.log)
                record("reset", [8] * A, env.reset(), ls)
                steps_in_ep = 0
        path = os.path.join(OUT, name + ".npz")
        _save_env_rows(path, rows, A, dict(seed=97 + A, A=A, enforce=1))
        d = dict(np.load(path).items())
        d["meta"] = np.array([97 + A, A, 1, int(bool(kw.get("coord_noise"))), int(bool(kw.get("DEBUG")))], dtype=np.int64)
        np.savez_compressed(path, **d)


def _save_env_rows(path, rows, A, meta):
    n = len(rows)
    f8 = lambda k: np.array([r[k] for r in rows], dtype=np.float64)
    i8 = lambda k: np.array([r[k] for r in rows], dtype=np.int64)
    # flatten the draw log: (event, kind, arg0, arg1, value)
    kinds = {"integers": 0, "poisson": 1, "normal": 2}
    dl = []
    for e, r in enumerate(rows):
        for (k, a0, a1, vals) in r["draws"]:
            for v in vals:
                dl.append((e, kinds[k], a0, a1, float(v)))
    np.savez_compressed(
        path,
        meta=np.array([meta["seed"], meta["A"], meta["enforce"]], dtype=np.int64),
        is_reset=np.array([r["kind"] == "reset" for r in rows], dtype=np.int8),
        actions=i8("actions"), obs=f8("obs"), reward=f8("reward"), team=f8("team"),
        done_ret=i8("done_ret"), info_oob=i8("info_oob"), info_oobc=i8("info_oobc"),
        info_blocked=i8("info_blocked"), src=f8("src"), intensity=i8("intensity"), bkg=i8("bkg"),
        det=f8("det"), sp=f8("sp"), euc=f8("euc"), prev=f8("prev"), oob=i8("oob"), oobc=i8("oobc"),
        coll=i8("coll"), blocked=i8("blocked"), inter=i8("inter"), done=i8("done"),
        iter_count=i8("iter_count"),
        draws=np.array(dl, dtype=np.float64).reshape(-1, 5),
    )
    print("wrote", path, n, "events", len(dl), "draws")


def gen_gae():
    sys.path.insert(0, os.path.join(REF))
    from algos.multiagent.ppo import PPOBuffer, discount_cumsum  # noqa
    rng = np.random.default_rng(7)
    cases = []
    # known-answer vector of the reference's own test (unit_tests/test_PPO.py:263-270)
    T = 480
    buf = PPOBuffer(observation_dimension=11, max_size=T, max_episode_length=120, number_agents=1)
    rew = (rng.integers(-70, 11, size=T) / 100.0).astype(np.float64)
    val = rng.normal(size=T).astype(np.float32)
    logp = rng.normal(size=T).astype(np.float32)
    cut = np.zeros(T, dtype=np.int8)       # 1 = trajectory ends after this step
    term = np.zeros(T, dtype=np.int8)      # 1 = terminal (last_val = 0), else bootstrap
    last_vals = np.zeros(T, dtype=np.float64)
    ep_lens = []
    start = 0
    t = 0
    while t < T:
        length = int(rng.integers(3, 121))
        end = min(t + length, T)
        is_term = bool(rng.integers(0, 2)) and end < T
        for s in range(t, end):
            buf.store(obs=np.zeros(11, np.float32), act=1, rew=float(rew[s]), val=float(val[s]),
                      logp=float(logp[s]), src=np.zeros(2, np.float32), full_observation={},
                      heatmap_stacks=None, terminal=(s == end - 1))
        lv = 0.0 if is_term else float(np.float32(rng.normal()))
        buf.GAE_advantage_and_rewardsToGO(lv)
        cut[end - 1] = 1
        term[end - 1] = int(is_term)
        last_vals[end - 1] = lv
        if is_term or (end - t) == 120 or end < T:
            buf.store_episode_length(end - t)
            ep_lens.append(end - t)
        t = end
    adv_raw = buf.adv_buf.copy()
    ret = buf.ret_buf.copy()
    data = buf.get()
    np.savez_compressed(os.path.join(OUT, "gae.npz"), rew=rew, val=val, cut=cut, term=term,
                        last_val=last_vals, adv_raw=adv_raw, ret=ret,
                        adv_norm=data["adv"].numpy(), gamma=0.99, lam=0.90,
                        ep_lens=np.array(ep_lens, dtype=np.int64),
                        n_ep_form=len(data["ep_form"]))
    print("wrote gae.npz; episodes", len(ep_lens), "ep_form", len(data["ep_form"]))


def gen_ff_core():
    import torch
    from algos.multiagent.NeuralNetworkCores.FF_core import ActorCritic
    torch.manual_seed(11)
    ac = ActorCritic(state_dim=11, action_dim=8, has_continuous_action_space=False, action_std_init=0.6)
    x = torch.rand(64, 11)
    x[:, 0] = torch.randint(0, 5000, (64,)).float()      # raw counts, as the env emits them
    with torch.no_grad():
        probs = ac.actor(x)
        values = ac.critic(x)
        act = torch.arange(64) % 8
        logp, v2, ent = ac.evaluate(x, act)
    sd = {k: v.numpy() for k, v in ac.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "ff_core.npz"), x=x.numpy(), probs=probs.numpy(),
                        values=values.numpy(), act=act.numpy(), logp=logp.numpy(), ent=ent.numpy(),
                        **{"sd_" + k: v for k, v in sd.items()})
    print("wrote ff_core.npz", list(sd))


def gen_welford():
    from algos.multiagent.NeuralNetworkCores.RADTEAM_core import StatisticStandardization
    rng = np.random.default_rng(3)
    xs = rng.poisson(rng.uniform(20, 5000, size=200)).astype(np.float64)
    st = StatisticStandardization()
    out = []
    for x in xs:
        st.update(x)
        out.append(st.standardize(x))
    np.savez_compressed(os.path.join(OUT, "welford.npz"), x=xs, z=np.array(out, dtype=np.float64))
    print("wrote welford.npz")


def gen_maps():
    """MapsBuffer.observation_to_map (algos/multiagent/NeuralNetworkCores/RADTEAM_core.py:532-616) driven by
    observation dicts from an obstacle env (inputs come from the repo's own env oracle; the EXPECTED maps come
    from the reference class).  Also records calculate_resolution_accuracy / map dimensions / the log-scale
    normaliser so the restatement's constants are pinned."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle.radsearch_oracle import PhiloxDraws, RadSearchOracle
    from algos.multiagent.NeuralNetworkCores import RADTEAM_core as R
    A, L = 3, 20
    scale = 1 / 2200.0
    ra = R.calculate_resolution_accuracy(resolution_multiplier=0.01, scale=scale)
    offset = scale * max((200.0, 500.0))
    bufs = {i: R.MapsBuffer(observation_dimension=11, steps_per_episode=L, number_of_agents=A, grid_bounds=(1, 1),
                            resolution_accuracy=ra, offset=offset, resolution_multiplier=0.01) for i in range(A)}
    env = RadSearchOracle(PhiloxDraws(77, 5), number_agents=A, obstruction_count=4, enforce_grid_boundaries=True)
    rng = np.random.default_rng(4)
    obs_log, pred_log, maps_log, reset_log = [], [], [], []
    obs = env._ret[0]
    steps = 0
    for t in range(70):
        od = {i: np.array(obs[i], dtype=np.float64) for i in range(A)}
        pred = (float(rng.uniform(0, 1.2)), float(rng.uniform(0, 1.2)))
        stacks = []
        for i in range(A):
            m = bufs[i].observation_to_map(od, i, pred)
            stacks.append(np.stack([np.array(x, dtype=np.float32) for x in m]))
        obs_log.append(np.stack([od[i] for i in range(A)]))
        pred_log.append(pred)
        maps_log.append(np.stack(stacks))
        acts = {i: int(rng.integers(0, 9)) for i in range(A)}
        if t % 9 == 4:
            acts = {i: 8 for i in range(A)}           # idle: revisits (median / visit-count paths)
        obs = env.step(acts)[0]
        steps += 1
        do_reset = env.done or steps == L
        reset_log.append(int(do_reset))
        if do_reset:
            for i in range(A):
                bufs[i].reset()
            env.epoch_end = (t % 2 == 0)
            obs = env.reset()[0]
            steps = 0
    nz = R.Normalizer()
    logs = np.array([nz.normalize_incremental_logscale(current_value=c, base=(L + 1) * A, increment_value=2)
                     for c in range(0, 2 * (L + 1) * A, 2)])
    np.savez_compressed(os.path.join(OUT, "maps.npz"), obs=np.array(obs_log), pred=np.array(pred_log),
                        maps=np.array(maps_log), reset_after=np.array(reset_log), ra=ra, offset=offset,
                        map_dim=np.array(bufs[0].map_dimensions), base=bufs[0].base, logscale=logs, A=A, L=L)
    print("wrote maps.npz", np.array(maps_log).shape, "ra", repr(ra), "dims", bufs[0].map_dimensions)


def gen_cnn():
    """CNN Actor / Critic of algos/multiagent/NeuralNetworkCores/RADTEAM_core.py:935-1345 on one map stack
    (the reference's Flatten(start_dim=0) only supports batch 1): state_dicts + inputs + outputs."""
    import torch
    from algos.multiagent.NeuralNetworkCores import RADTEAM_core as R
    torch.manual_seed(3)
    actor = R.Actor(map_dim=(27, 27), action_dim=8)
    critic = R.Critic(map_dim=(27, 27))
    xs_a = torch.rand(5, 1, 6, 27, 27)
    xs_c = torch.rand(5, 1, 4, 27, 27)
    with torch.no_grad():
        probs = torch.stack([actor.actor(xs_a[i]) for i in range(5)])
        vals = torch.stack([critic.forward(xs_c[i]) for i in range(5)])
    out = {"xa": xs_a.numpy(), "xc": xs_c.numpy(), "probs": probs.numpy(), "vals": vals.numpy().reshape(-1)}
    # (the reference modules also carry unused debugging copies `step1..step7` of the layers: not saved)
    out.update({"a_" + k: v.numpy() for k, v in actor.state_dict().items() if k.startswith("actor.")})
    out.update({"c_" + k: v.numpy() for k, v in critic.state_dict().items() if k.startswith("critic.")})
    np.savez_compressed(os.path.join(OUT, "cnn.npz"), **out)
    print("wrote cnn.npz", [k for k in out if k.startswith(("a_", "c_"))])


def gen_pfgru():
    """the PFGRU location predictor exactly as the reference's CNN core instantiates and calls it
    (algos/test_cnn/RADTEAM_core.py:1790-1795: PFGRUCell(input_size=3, obs_size=3, activation="tanh", hidden_size=24);
    forward :1586-1631 with soft resampling :1466-1515).  Two traces of 12 steps each: hidden state carried from step to
    step, and the CNN harness' usage (every step from the episode's h0).  Every random draw the cell makes is recorded: the
    reparameterisation noise (torch.FloatTensor(shape).normal_(), :1528) by replaying the generator state, the resampling
    indices by wrapping torch.multinomial (:1485)."""
    import torch
    from algos.test_cnn import RADTEAM_core as R
    torch.manual_seed(29)
    cell = R.PFGRUCell(input_size=3, obs_size=3, activation="tanh", hidden_size=24)
    cell.eval()
    eps_log, idx_log = [], []
    orig_rep = cell.reparameterize

    def rec_rep(mean, var):
        st = torch.get_rng_state()
        out = orig_rep(mean, var)
        end = torch.get_rng_state()
        torch.set_rng_state(st)
        eps_log.append(torch.FloatTensor(var.shape).normal_().clone())       # the same draw the reference just consumed
        assert torch.equal(torch.get_rng_state(), end)
        return out
    cell.reparameterize = rec_rep
    orig_mn = torch.multinomial

    def rec_mn(*a, **k):
        r = orig_mn(*a, **k)
        idx_log.append(r.clone())
        return r
    torch.multinomial = rec_mn
    try:
        rng = np.random.default_rng(31)
        T = 12
        obs = np.stack([rng.poisson(800, T).astype(np.float32) / 100.0, rng.uniform(0.1, 1.0, T).astype(np.float32),
                        rng.uniform(0.1, 1.0, T).astype(np.float32)], axis=1)
        out = {"obs": obs}
        out.update({"sd_" + k: v.numpy() for k, v in cell.state_dict().items()})
        with torch.no_grad():
            for tag, carry in (("carry", True), ("fresh", False)):
                hidden = cell.init_hidden(1)
                h0 = hidden[0].clone()
                eps_log.clear(); idx_log.clear()
                locs, hs, ps = [], [], []
                for t in range(T):
                    loc, new_hidden = cell(torch.from_numpy(obs[t:t + 1]), hidden)
                    if carry:
                        hidden = new_hidden
                    locs.append(loc.numpy().copy()); hs.append(new_hidden[0].numpy().copy()); ps.append(new_hidden[1].numpy().reshape(-1).copy())
                out.update({f"{tag}_h0": h0.numpy(), f"{tag}_eps": torch.stack(eps_log).numpy(),
                            f"{tag}_idx": torch.stack(idx_log).numpy().reshape(T, -1).astype(np.int64),
                            f"{tag}_loc": np.stack(locs), f"{tag}_h": np.stack(hs), f"{tag}_p": np.stack(ps)})
    finally:
        torch.multinomial = orig_mn
    np.savez_compressed(os.path.join(OUT, "pfgru.npz"), **out)
    print("wrote pfgru.npz", {k: v.shape for k, v in out.items() if not k.startswith("sd_")})


def gen_round2():
    rng = np.random.default_rng(5)
    sp = np.concatenate([rng.uniform(0, 4000, 200000), rng.integers(0, 400000, 200000) / 100.0,
                         np.arange(0, 4000, 0.5)])
    a = np.array([round(-0.5 * s / 2000.0, 2) for s in sp])
    b = np.array([round(-1.0 * s / 2000.0, 2) for s in sp])
    # keep the fixture small: sample 20k
    idx = rng.choice(len(sp), 20000, replace=False)
    np.savez_compressed(os.path.join(OUT, "round2.npz"), sp=sp[idx], half=a[idx], full=b[idx])
    print("wrote round2.npz")


def gen_train_trace():
    """SURVEY 8c fixture (7), row T: the control flow of train_PPO.train (algos/multiagent/train.py:259-627) executed by
    the reference's OWN method over the reference's OWN env.  The agents and loggers are recording stand-ins (the
    reference's AgentPPO rejects 'ff' and its CNN wiring is mid-refactor), so what is pinned is exactly the
    loop: call order, the terminal flag passed to store(), which steps bootstrap and with which value, when
    env.epoch_end is raised, when env.reset()/reset_agent() run and what reaches the loggers."""
    import json
    import types
    from gym_rad_search.envs.rad_search_env import RadSearch
    from algos.multiagent import train as RT

    out = {}
    from algos.multiagent.NeuralNetworkCores.RADTEAM_core import StatisticStandardization
    for name, A, global_critic, T, L, epochs, seed, arch in (("a1_individual", 1, False, 50, 18, 3, 5, "cnn"),
                                                              ("a2_team", 2, True, 44, 16, 3, 9, "cnn"),
                                                              ("a1_mlp_standardized", 1, False, 58, 20, 3, 13, "mlp")):
        rec = RecordingGenerator(seed)
        env = RadSearch(number_agents=A, np_random=rec, obstruction_count=0, enforce_grid_boundaries=True)
        ev = []
        script = np.random.default_rng(77 + A)
        counter = {"calls": 0, "episode": 0}

        class EnvProxy:
            def reset(self_):
                ev.append(["env_reset"])
                counter["episode"] += 1
                return env.reset()

            def step(self_, action=None):
                ev.append(["env_step", {str(k): int(v) for k, v in action.items()}])
                return env.step(action=action)

            def __getattr__(self_, k):
                return getattr(env, k)

            def __setattr__(self_, k, v):
                if k == "epoch_end":
                    ev.append(["epoch_end_set", bool(v)])
                setattr(env, k, v)

        class Agent:
            def __init__(self_, id):
                self_.id = id

            def step(self_, observations, hidden=None, message=None):
                counter["calls"] += 1
                val = 0.001 * counter["calls"]
                # even episodes head for the source (terminal branch), odd ones wander
                greedy = counter["episode"] % 2 == 0 and self_.id == 0      # a second greedy agent would only collide
                act = _greedy_action(env, self_.id) if greedy else int(script.integers(0, 9))
                ev.append(["agent_step", self_.id, {str(k): np.asarray(v, dtype=np.float64).tolist() for k, v in observations.items()},
                           act, val])
                return types.SimpleNamespace(action=act, state_value=val, action_logprob=-0.5 - val, hiddens=None), None

            def store(self_, obs, rew, act, val, logp, src, terminal, heatmap_stacks, full_observation):
                ev.append(["store", self_.id, np.asarray(obs, dtype=np.float64).tolist(), float(rew), int(act), float(val),
                           float(logp), np.asarray(src, dtype=np.float64).tolist(), bool(terminal)])

            def GAE_advantage_and_rewardsToGO(self_, last_state_value):
                ev.append(["gae", self_.id, float(last_state_value)])

            def store_episode_length(self_, episode_length):
                ev.append(["ep_len", self_.id, int(episode_length)])

            def reset_agent(self_):
                ev.append(["reset_agent", self_.id])

            # RAD-A2C branch plumbing touched by train() (:311, :327-330, :516): no effect on the control flow
            agent = types.SimpleNamespace(model=types.SimpleNamespace(eval=lambda: None), pi=types.SimpleNamespace(
                logits_net=types.SimpleNamespace(v_net=types.SimpleNamespace(eval=lambda: None))))

            def reset_hidden(self_):
                ev.append(["reset_hidden", self_.id])
                return None

            def reduce_pfgru_training(self_):
                ev.append(["reduce_pfgru", self_.id])

            def save(self_, path=None):
                ev.append(["save", self_.id])

            def update_agent(self_, logger=None):
                ev.append(["update", self_.id])
                return types.SimpleNamespace(stop_iteration=1, loss_policy=0.0, loss_critic=0.0, loss_predictor=0.0,
                                             kl_divergence=0.0, Entropy=0.0, ClipFrac=0.0, LocLoss=0.0, VarExplain=0.0)

        import shutil
        from pathlib import Path
        from algos.multiagent.rl_tools.epoch_logger import EpochLogger as RefEpochLogger
        log_root = tempfile.mkdtemp(prefix="rs_trace_logs_")
        logger_calls = {i: [] for i in range(A)}

        class Logger(RefEpochLogger):
            """The reference's OWN EpochLogger (epoch_logger.py:314-403) writing a real progress.txt; every call train()
            makes on it is recorded so the build's logger can be replayed against the file the reference produced."""

            def __init__(self_, id):
                super().__init__(output_dir=Path(log_root) / f"{id}_agent")
                self_.id = id

            def store(self_, **kw):
                for k, v in kw.items():
                    if k in ("EpRet", "EpLen", "DoneCount", "OutOfBound"):
                        ev.append(["log", self_.id, k, float(v)])
                logger_calls[self_.id].append(["store", {k: float(v) for k, v in kw.items()}])
                super().store(**kw)

            def log_tabular(self_, key, val=None, **kw):
                if key == "TotalEnvInteracts":
                    ev.append(["tabular", self_.id, key, float(val)])
                logger_calls[self_.id].append(["log_tabular", key, None if val is None else (int(val) if isinstance(val, (int, np.integer)) else float(val)), dict(kw)])
                super().log_tabular(key, val, **kw)

            def dump_tabular(self_):
                ev.append(["dump", self_.id])
                logger_calls[self_.id].append(["dump_tabular"])
                super().dump_tabular()

            def save_state(self_, *a, **k):
                ev.append(["save", self_.id])

        sim = object.__new__(RT.train_PPO)      # skip __post_init__ (builds the real agents / loggers)
        sim.env = EnvProxy()
        sim.logger_kwargs = dict(data_dir=".", env_name="x", exp_name="x", seed=0)
        sim.ppo_kwargs = {}
        sim.seed, sim.number_of_agents = 0, A
        sim.actor_critic_architecture, sim.global_critic_flag = arch, global_critic
        sim.steps_per_epoch, sim.steps_per_episode, sim.total_epochs = T, L, epochs
        sim.render, sim.save_path, sim.save_freq, sim.save_gif_freq, sim.save_gif = False, ".", 500, float("inf"), False
        sim.render_first_episode, sim.episode_count, sim.DEBUG = True, 0, False
        sim.stat_buffers = {i: StatisticStandardization() for i in range(A)} if arch == "mlp" else {}
        sim.agents = {i: Agent(i) for i in range(A)}
        sim.loggers = {i: Logger(i) for i in range(A)}
        sim.train()
        progress = {}
        for i in range(A):
            sim.loggers[i].output_file.flush()
            with open(os.path.join(log_root, f"{i}_agent", "progress.txt")) as f:
                progress[str(i)] = f.read()
        shutil.rmtree(log_root)
        out[name] = dict(A=A, global_critic=global_critic, T=T, L=L, epochs=epochs, seed=seed, arch=arch,
                         draws=[[k, a0, a1, v] for (k, a0, a1, v) in rec.log], events=ev,
                         episode_count=int(sim.episode_count),
                         logger_calls={str(i): logger_calls[i] for i in range(A)}, progress=progress)
        print(name, len(ev), "events", sum(1 for e in ev if e[0] == "gae"), "trajectories",
              sum(1 for e in ev if e[0] == "gae" and e[2] == 0.0), "terminal")
    with open(os.path.join(OUT, "train_trace.json"), "w") as f:
        json.dump(out, f)
    print("wrote train_trace.json")


def gen_loss():
    """SURVEY 8c fixture (5), row P6/P7: AgentPPO.update_rada2c (algos/multiagent/ppo.py:1150-1281) run as the
    reference wrote it -- its PPO-clip / value / entropy loss, KL test, backward and Adam step -- on episodes of
    our choosing, with the FF_core network (FF_core.py:42-129) standing behind agent.grad_step (the
    2x64 MLP trained with the RAD-A2C loss form).  Captured: inputs, parameters before, loss / kl / entropy /
    clip fraction / value loss, parameters after the step, and the early-stop decision for a batch that trips it."""
    import types
    import torch
    from algos.multiagent import ppo as RP
    from algos.multiagent.NeuralNetworkCores.FF_core import ActorCritic
    from torch.distributions import Categorical

    torch.manual_seed(23)
    ac = ActorCritic(state_dim=11, action_dim=8, has_continuous_action_space=False, action_std_init=0.6)
    rng = np.random.default_rng(31)
    res = {}
    for tag, scale in (("step", 1.0), ("stop", 40.0)):
        lens = [int(x) for x in rng.integers(3, 21, size=6)]
        eps = []
        for n in lens:
            obs = rng.normal(size=(n, 11)).astype(np.float32)
            act = rng.integers(0, 8, size=n).astype(np.float32)
            adv = rng.normal(size=n).astype(np.float32)
            ret = rng.normal(size=n).astype(np.float32)
            with torch.no_grad():
                d = Categorical(ac.actor(torch.from_numpy(obs)))
                logp_old = d.log_prob(torch.from_numpy(act)).numpy() + (rng.normal(size=n) * 0.05 * scale).astype(np.float32)
            src = rng.uniform(0, 1, size=(n, 2)).astype(np.float32)
            # column layout of the reference's episode form: obs 0:11 | adv 11 | ret 12 | logp 13 | act 14 | src 15:17
            eps.append(np.concatenate([obs, adv[:, None], ret[:, None], logp_old[:, None], act[:, None], src], axis=1))
        before = {k: v.detach().clone().numpy() for k, v in ac.state_dict().items()}

        class AgentShim:
            pi = ac.actor

            @staticmethod
            def grad_step(obs, act, hidden=None):
                probs = ac.actor(obs)
                d = Categorical(probs)
                return d, ac.critic(obs), d.log_prob(act), torch.zeros(obs.shape[0], 2)

        opt = RP.OptimizationStorage(
            critic_flag=False, pi_optimizer=torch.optim.Adam(ac.parameters(), lr=3e-4), critic_optimizer=None,
            model_optimizer=torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=5e-3))
        me = types.SimpleNamespace(minibatch=1, agent=AgentShim, agent_optimizer=opt, clip_ratio=0.2, alpha=0.1, target_kl=0.07,
                                   env_height=5.0, reset_hidden=lambda: None)
        np.random.seed(3)
        order = np.random.choice(np.arange(0, len(eps)), size=len(eps), replace=False)    # what :1183 will draw
        np.random.seed(3)
        data = dict(ep_form=[[torch.from_numpy(e)] for e in eps])
        loss, info, term, ploss = RP.AgentPPO.update_rada2c(me, data, min_iterations=len(eps), logger=None)
        after = {k: v.detach().clone().numpy() for k, v in ac.state_dict().items()}
        # pi_optimizer.zero_grad() ran before loss_pi.backward() (:1253-1254): .grad still holds this call's gradients
        grads = {k: (torch.zeros_like(v) if v.grad is None else v.grad.detach().clone()).numpy() for k, v in ac.named_parameters()}
        res.update({f"{tag}_grad_{k}": v for k, v in grads.items()})
        res.update({f"{tag}_ep{i}": e for i, e in enumerate(eps)})
        res.update({f"{tag}_before_{k}": v for k, v in before.items()})
        res.update({f"{tag}_after_{k}": v for k, v in after.items()})
        res[f"{tag}_order"] = order
        res[f"{tag}_loss"] = np.float32(loss.item())
        res[f"{tag}_kl"] = np.float32(info["kl"])
        res[f"{tag}_ent"] = np.float32(info["ent"])
        res[f"{tag}_cf"] = np.float32(info["cf"])
        res[f"{tag}_val_loss"] = np.float32(info["val_loss"])
        res[f"{tag}_term"] = np.bool_(term)
        print(tag, "loss", loss.item(), "kl", float(info["kl"]), "term", term)
    res["n_eps"] = np.int64(6)
    np.savez_compressed(os.path.join(OUT, "rada2c_loss.npz"), **res)
    print("wrote rada2c_loss.npz")


def gen_rada2c_core():
    """the RAD-A2C core exactly as main.py:541-553 instantiates it --
    RNNModelActorCritic(obs_dim=11, act_dim=8, hidden=[[24]], hidden_sizes_pol=[[32]], hidden_sizes_val=[[32]],
    hidden_sizes_rec=[24], net_type="rnn") (NeuralNetworkCores/RADA2C_core.py:477-607: GRU(13 -> 24) + Woms / Valms heads +
    PFGRUCell(40, 3, 3, 24, 0.7, True, "tanh")) -- driven by the reference's own methods:
      step      a 14-step episode through ac.step with the hidden state carried (:528-548)
      rada2c    AgentPPO.update_rada2c (ppo.py:1150-1281) on 5 episodes with the REAL grad_step (:550-566, BPTT through the GRU)
      model     AgentPPO.update_model (ppo.py:1047-1148), one iteration: the PFGRU regression / ELBO loss, clip, Adam step
    Every random draw is recorded: GRU h0 and particle h0 (reset_hidden), reparameterisation noise (replayed generator state),
    resampling indices (torch.multinomial wrapped), sampled actions."""
    import types
    import torch
    from algos.multiagent import ppo as RP
    from algos.multiagent.NeuralNetworkCores import RADA2C_core as R

    torch.manual_seed(41)
    ac = R.RNNModelActorCritic(obs_dim=11, act_dim=8, hidden=[[24]], hidden_sizes_pol=[[32]], hidden_sizes_val=[[32]],
                               hidden_sizes_rec=[24], net_type="rnn", seed=41)
    ac.set_mode("eval")
    res = {"sd_" + k: v.detach().clone().numpy() for k, v in ac.state_dict().items()}
    eps_log, idx_log, hid_log = [], [], []
    orig_rep = ac.model.reparameterize

    def rec_rep(mean, var):
        st = torch.get_rng_state()
        out = orig_rep(mean, var)
        end = torch.get_rng_state()
        torch.set_rng_state(st)
        eps_log.append(torch.FloatTensor(var.shape).normal_().clone())
        assert torch.equal(torch.get_rng_state(), end)
        return out
    ac.model.reparameterize = rec_rep
    orig_mn = torch.multinomial

    def rec_mn(*a, **k):
        r = orig_mn(*a, **k)
        if r.shape[-1] == 40:                    # the resampling draw (Categorical.sample also goes through torch.multinomial)
            idx_log.append(r.clone())
        return r
    orig_reset = ac.reset_hidden

    def rec_reset(batch_size=1):
        h = orig_reset(batch_size)
        hid_log.append((h[0][0].clone(), h[0][1].clone(), h[1].clone()))
        return h
    torch.multinomial = rec_mn
    rng = np.random.default_rng(43)

    def episode_obs(n):
        o = rng.uniform(0.0, 1.0, size=(n, 11)).astype(np.float32)
        o[:, 0] = rng.normal(size=n).astype(np.float32) * 1.5            # standardised reading
        return o
    try:
        # ---- step: one episode, hidden carried
        T = 14
        obs = episode_obs(T)
        hidden = rec_reset()
        eps_log.clear(); idx_log.clear()
        acts, logps, vals, locs, gh = [], [], [], [], []
        for t in range(T):
            r, _ = ac.step(obs[t], hidden)
            hidden = r.hiddens
            acts.append(int(r.action)); logps.append(float(r.action_logprob)); vals.append(np.asarray(r.state_value).reshape(-1)[0])
            locs.append(np.asarray(r.loc_pred).reshape(2).copy()); gh.append(hidden[1].detach().numpy().reshape(-1).copy())
        res.update(step_obs=obs, step_pf_h0=hid_log[-1][0].numpy(), step_gru_h0=hid_log[-1][2].numpy().reshape(-1),
                   step_eps=torch.stack(eps_log).numpy(), step_idx=torch.stack(idx_log).numpy().reshape(T, -1).astype(np.int64),
                   step_act=np.array(acts, dtype=np.int64), step_logp=np.array(logps, dtype=np.float32),
                   step_val=np.array(vals, dtype=np.float32), step_loc=np.stack(locs).astype(np.float32), step_gru_h=np.stack(gh))
        # ---- episodes for the two updates (column layout of PPOBuffer.get's ep_form: obs 0:11 | adv | ret | logp | act | src 15:17)
        lens = [9, 14, 6, 11, 8]
        eps = []
        for n in lens:
            o = episode_obs(n)
            act = rng.integers(0, 8, size=n).astype(np.float32)
            adv = rng.normal(size=n).astype(np.float32)
            ret = rng.normal(size=n).astype(np.float32)
            logp_old = (np.log(1.0 / 8.0) + rng.normal(size=n) * 0.05).astype(np.float32)
            src = np.tile(rng.uniform(300.0, 2400.0, size=(1, 2)).astype(np.float32), (n, 1))
            eps.append(np.concatenate([o, adv[:, None], ret[:, None], logp_old[:, None], act[:, None], src], axis=1))
        res.update({f"ep{i}": e for i, e in enumerate(eps)})
        res["n_eps"] = np.int64(len(eps))
        opt = RP.OptimizationStorage(critic_flag=False, pi_optimizer=torch.optim.Adam(ac.pi.parameters(), lr=3e-4), critic_optimizer=None,
                                     model_optimizer=torch.optim.Adam(ac.model.parameters(), lr=5e-3))
        me = types.SimpleNamespace(minibatch=1, agent=ac, agent_optimizer=opt, clip_ratio=0.2, alpha=0.1, target_kl=0.07,
                                   env_height=2500.0, reset_hidden=rec_reset, train_pfgru_iters=1,
                                   bp_args=RP.BpArgs(bp_decay=0.1, l2_weight=1.0, l1_weight=0.0, elbo_weight=1.0, area_scale=2500.0))
        data = dict(ep_form=[[torch.from_numpy(e.copy())] for e in eps])
        # ---- update_rada2c (the policy side first: update_model changes the PFGRU weights that feed loc_pred)
        np.random.seed(3)
        order = np.random.choice(np.arange(0, len(eps)), size=len(eps), replace=False)
        np.random.seed(3)
        hid_log.clear(); eps_log.clear(); idx_log.clear()
        before = {k: v.detach().clone().numpy() for k, v in ac.state_dict().items()}
        loss, info, term, ploss = RP.AgentPPO.update_rada2c(me, data, min_iterations=len(eps), logger=None)
        grads = {k: (torch.zeros_like(v) if v.grad is None else v.grad.detach().clone()).numpy() for k, v in ac.pi.named_parameters()}
        res.update({"a2c_grad_" + k: v for k, v in grads.items()})
        res.update({"a2c_after_" + k: v.detach().clone().numpy() for k, v in ac.state_dict().items()})
        res.update(a2c_order=order, a2c_loss=np.float32(loss.item()), a2c_kl=np.float32(info["kl"]), a2c_ent=np.float32(info["ent"]),
                   a2c_cf=np.float32(info["cf"]), a2c_val_loss=np.float32(info["val_loss"]), a2c_term=np.bool_(term),
                   a2c_locloss=np.float32(float(ploss)))
        off = 0
        for k, ei in enumerate(order):                                     # draws in the order the reference consumed them
            n = lens[int(ei)]
            res[f"a2c_pf_h0_{k}"] = hid_log[k][0].numpy(); res[f"a2c_gru_h0_{k}"] = hid_log[k][2].numpy().reshape(-1)
            res[f"a2c_eps_{k}"] = torch.stack(eps_log[off:off + n]).numpy()
            res[f"a2c_idx_{k}"] = torch.stack(idx_log[off:off + n]).numpy().reshape(n, -1).astype(np.int64)
            off += n
        assert off == len(eps_log) == len(idx_log) and len(hid_log) == len(eps)
        print("update_rada2c: loss", loss.item(), "kl", float(info["kl"]), "val_loss", float(info["val_loss"]), "term", term)
        # ---- update_model: one iteration over the episodes in list order
        hid_log.clear(); eps_log.clear(); idx_log.clear()
        mloss = RP.AgentPPO.update_model(me, data)
        mg = {k: (torch.zeros_like(v) if v.grad is None else v.grad.detach().clone()).numpy() for k, v in ac.model.named_parameters()}
        res.update({"model_grad_" + k: v for k, v in mg.items()})                 # after clip_grad_norm_(., 5) (:1137)
        res.update({"model_after_" + k: v.detach().clone().numpy() for k, v in ac.model.state_dict().items()})
        res["model_loss"] = np.float32(mloss.item())
        off = 0
        for k, n in enumerate(lens):
            res[f"model_pf_h0_{k}"] = hid_log[k][0].numpy()
            res[f"model_eps_{k}"] = torch.stack(eps_log[off:off + n]).numpy()
            res[f"model_idx_{k}"] = torch.stack(idx_log[off:off + n]).numpy().reshape(n, -1).astype(np.int64)
            off += n
        assert off == len(eps_log)
        print("update_model: loss", mloss.item())
    finally:
        torch.multinomial = orig_mn
    np.savez_compressed(os.path.join(OUT, "rada2c_core.npz"), **res)
    print("wrote rada2c_core.npz", len(res), "arrays")


def gen_cnn_loss():
    """Row P6, CNN branch: AgentPPO.compute_batched_losses_pi / compute_loss_pi (algos/multiagent/ppo.py:903-997) and
    compute_batched_losses_critic / compute_loss_critic (:999-1045) as the reference wrote them, over the reference's
    CNN Actor / Critic (RADTEAM_core.py:935-1345): losses, diagnostics and parameter gradients for a fixed batch."""
    import types
    import torch
    from algos.multiagent import ppo as RP
    from algos.multiagent.NeuralNetworkCores import RADTEAM_core as R
    torch.manual_seed(17)
    actor = R.Actor(map_dim=(27, 27), action_dim=8)
    critic = R.Critic(map_dim=(27, 27))
    rng = np.random.default_rng(19)
    n = 14
    # inputs in the form the trainer holds them: the 4 shared maps + the owner's location / prediction cells; the
    # actor stack follows CNNBase.get_map_stack (RADTEAM_core.py:1791-1836): {prediction, location, others =
    # combined - location, readings, visits, obstacles}
    maps = (rng.random((n, 4, 27, 27)) * (rng.random((n, 4, 27, 27)) < 0.08)).astype(np.float32)
    cells = rng.integers(0, 729, size=n)
    pcells = np.where(rng.random(n) < 0.7, rng.integers(0, 729, size=n), -1)
    for i in range(n):
        maps[i, 0] = np.round(maps[i, 0] * 3)                       # combined-locations map holds small counts
        maps[i, 0].reshape(-1)[cells[i]] += 1.0
    xa = np.zeros((n, 1, 6, 27, 27), dtype=np.float32)
    for i in range(n):
        if pcells[i] >= 0:
            xa[i, 0, 0].reshape(-1)[pcells[i]] = 1.0
        xa[i, 0, 1].reshape(-1)[cells[i]] = 1.0
        xa[i, 0, 2] = maps[i, 0] - xa[i, 0, 1]
        xa[i, 0, 3:6] = maps[i, 1:4]
    xc = maps[:, None].copy()
    act = torch.from_numpy(rng.integers(0, 8, size=n))
    adv = torch.from_numpy(rng.normal(size=n).astype(np.float32))
    ret = torch.from_numpy(rng.normal(size=n).astype(np.float32))
    with torch.no_grad():
        lp = torch.stack([actor.get_action_information(torch.from_numpy(xa[i]), act[i])[0] for i in range(n)]).reshape(-1)
    logp_old = lp + torch.from_numpy((rng.normal(size=n) * 0.3).astype(np.float32))
    data = dict(obs=torch.zeros(n, 11), act=act, adv=adv, logp=logp_old, ret=ret)
    me = types.SimpleNamespace(actor_critic_architecture="cnn", clip_ratio=0.2, reset_agent=lambda: None,
                               agent=types.SimpleNamespace(pi=actor, critic=critic, mseLoss=torch.nn.MSELoss()))
    me.compute_loss_pi = lambda **kw: RP.AgentPPO.compute_loss_pi(me, **kw)
    me.compute_loss_critic = lambda **kw: RP.AgentPPO.compute_loss_critic(me, **kw)
    sample = list(range(n))
    pi = RP.AgentPPO.compute_batched_losses_pi(me, sample=sample, data=data, mapstacks_buffer=[torch.from_numpy(x) for x in xa])
    pi["pi_loss"].backward()
    cr = RP.AgentPPO.compute_batched_losses_critic(me, data=data, map_buffer_maps=[torch.from_numpy(x) for x in xc], sample=sample)
    cr["critic_loss"].backward()
    out = dict(xa=xa, xc=xc, maps=maps, cells=cells.astype(np.int64), pcells=pcells.astype(np.int64), act=act.numpy(), adv=adv.numpy(), ret=ret.numpy(), logp_old=logp_old.numpy(),
               pi_loss=np.float32(pi["pi_loss"].item()), kl=np.float64(pi["kl"]), entropy=np.float64(pi["entropy"]),
               clip_fraction=np.float64(pi["clip_fraction"]), critic_loss=np.float32(cr["critic_loss"].item()))
    # (the reference modules also carry unused debugging copies `step1..step7` of the layers: not saved)
    out.update({"a_" + k: v.numpy() for k, v in actor.state_dict().items() if k.startswith("actor.")})
    out.update({"c_" + k: v.numpy() for k, v in critic.state_dict().items() if k.startswith("critic.")})
    none = [k for k, v in list(actor.named_parameters()) + list(critic.named_parameters()) if v.grad is None]
    print("parameters without gradient:", none)
    out.update({"ga_" + k: v.grad.numpy() for k, v in actor.named_parameters() if v.grad is not None})
    out.update({"gc_" + k: v.grad.numpy() for k, v in critic.named_parameters() if v.grad is not None})
    np.savez_compressed(os.path.join(OUT, "cnn_loss.npz"), **out)
    print("wrote cnn_loss.npz", float(out["pi_loss"]), float(out["kl"]), float(out["clip_fraction"]), float(out["critic_loss"]))


def gen_refresh():
    """refresh_environment (rad_search_env.py:799-874), obstacle-free: saved (source, detector, intensity, background)
    tuples in the format of algos/test_environment/eval/test_env_gen.py:13-24 are loaded into the reference env, followed
    by scripted steps (incl. a first step that runs into the wall, which is priced with the STALE sp_dist)."""
    from gym_rad_search.envs.rad_search_env import RadSearch
    import json
    out = {}
    for name, A, enforce in (("a1", 1, True), ("a2", 2, True), ("a1_free", 1, False)):
        rec = RecordingGenerator(21)
        env = RadSearch(number_agents=A, np_random=rec, obstruction_count=0, enforce_grid_boundaries=enforce)
        rows = []
        script = np.random.default_rng(5 + A)
        env_dict = {"env_0": (np.array([1500.0, 700.0]), np.array([205.0, 1900.0]), 4321000, 37),
                    "env_1": (np.array([300.0, 2100.0]), np.array([2100.0, 400.0]), 9000123, 12),
                    "env_2": (np.array([1000.0, 1000.0]), np.array([2199.0, 2199.0]), 1000000, 50)}
        for eid in (0, 1, 2, 1):
            ls = len(rec.log)
            obs = env.refresh_environment(env_dict, eid)
            rows.append(dict(kind="refresh", id=eid, obs={str(i): np.asarray(obs[i], dtype=np.float64).tolist() for i in range(A)},
                             draws=rec.log[ls:], **_snapshot(env, A)))
            for t in range(14):
                if t == 0:
                    acts = {i: 0 for i in range(A)}                 # into the left wall for env_0 (x = 205 - 100 < 200... outside)
                elif t < 6:
                    acts = {i: _greedy_action(env, i) if i == 0 else int(script.integers(0, 9)) for i in range(A)}
                else:
                    acts = {i: int(script.integers(0, 9)) for i in range(A)}
                ls = len(rec.log)
                o, r, d, info = env.step(acts)
                rows.append(dict(kind="step", actions={str(k): int(v) for k, v in acts.items()},
                                 obs={str(i): np.asarray(o[i], dtype=np.float64).tolist() for i in range(A)},
                                 reward={str(i): float(r["individual_reward"][i]) for i in range(A)},
                                 done_ret={str(i): bool(d[i]) for i in range(A)}, draws=rec.log[ls:], **_snapshot(env, A)))
                if env.done:
                    break
        out[name] = dict(A=A, enforce=enforce, seed=21, init_draws=[], rows=rows,
                         env_dict={k: [np.asarray(v[0]).tolist(), np.asarray(v[1]).tolist(), int(v[2]), int(v[3])] for k, v in env_dict.items()})
        # draws consumed by the constructor's reset come first in rec.log: keep them so a replay starts identically
        n_init = len(rec.log) - sum(len(r["draws"]) for r in rows)
        out[name]["init_draws"] = rec.log[:n_init]
        print(name, len(rows), "rows")
    with open(os.path.join(OUT, "refresh.json"), "w") as f:
        json.dump(out, f)
    print("wrote refresh.json")


if __name__ == "__main__":
    _install_placeholders()
    sys.path.insert(0, os.path.join(REF, "gym_rad_search"))
    sys.path.insert(0, REF)
    which = sys.argv[1:] or ["env", "envforms", "envopt", "gae", "ff", "welford", "round2", "maps", "cnn", "pfgru", "train", "loss", "cnnloss", "refresh", "rada2c"]
    if "env" in which:
        gen_env_scenarios()
    if "envforms" in which:
        gen_env_callforms()
    if "envopt" in which:
        gen_env_options()
    if "gae" in which:
        gen_gae()
    if "ff" in which:
        gen_ff_core()
    if "welford" in which:
        gen_welford()
    if "round2" in which:
        gen_round2()
    if "maps" in which:
        gen_maps()
    if "cnn" in which:
        gen_cnn()
    if "pfgru" in which:
        gen_pfgru()
    if "train" in which:
        gen_train_trace()
    if "loss" in which:
        gen_loss()
    if "cnnloss" in which:
        gen_cnn_loss()
    if "refresh" in which:
        gen_refresh()
    if "rada2c" in which:
        gen_rada2c_core()
