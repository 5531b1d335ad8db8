"""GPU parity proper: the HIP kernels, called through the C ABI (radiation_ppo_amd.envs -> ctypes ->
librs_hip.so), against the oracle on identical Philox streams.  Bar: bit-exact on every output --
observations (float32 of the oracle's float64), rewards, done latch, info flags, positions, shortest-
path distances (float64) -- for every env and step."""
import numpy as np
import pytest
import torch

from oracle.radsearch_oracle import PhiloxDraws, PhiloxGeomDraws, RadSearchOracle, sample_layout

pytestmark = pytest.mark.gpu

SEED = 289714752   # robust_seed(2) of the reference (main.py:478)


def _make(N, A, obst, enforce, group=1, base=0, falloff="reference", **opts):
    from radiation_ppo_amd.envs import RadSearchVec
    vec = RadSearchVec(N, number_agents=A, obstruction_count=obst, enforce_grid_boundaries=enforce, seed=SEED,
                       env_id_base=base, geom_group_size=group, falloff=falloff, **opts)
    oracles = []
    geom = {}
    for n in range(N):
        layout_fn = None
        if group > 1 and obst != 0:
            gi = n // group
            if gi not in geom:
                geom[gi] = {"draws": PhiloxGeomDraws(SEED, base + gi * group), "epoch": 0, "rects": None, "stamp": -1}

            def layout_fn(gi=gi):
                st = geom[gi]
                if st["stamp"] != st["want"]:
                    st["draws"].begin(st["epoch"])
                    st["rects"] = sample_layout(st["draws"], obst)
                    st["epoch"] += 1
                    st["stamp"] = st["want"]
                return st["rects"]
            geom[gi]["want"] = 0
        # the oracle constructor performs the first reset (like the reference's __post_init__)
        oracles.append((n, layout_fn))
    return vec, oracles, geom


NOISY = False     # coord_noise: the two noisy coordinates pass through log / sincos of different float64 libraries (see the test)


def _compare(vec, outs, refs, rets, tag):
    obs, rew, team, done = (t.cpu().numpy() for t in outs[:4])
    info = outs[4]
    oob = info["out_of_bounds"].cpu().numpy()
    oobc = info["out_of_bounds_count"].cpu().numpy()
    blk = info["blocked"].cpu().numpy()
    col = info["collision"].cpu().numpy()
    x = vec.state("x").cpu().numpy()
    y = vec.state("y").cpu().numpy()
    sp = vec.state("sp").cpu().numpy()
    prev = vec.state("prev").cpu().numpy()
    for n, (env, ret) in enumerate(zip(refs, rets)):
        if ret is None:
            continue
        o, r, d, i = ret
        for a in range(env.number_agents):
            exp = np.asarray(o[a], dtype=np.float64).astype(np.float32)
            if NOISY:
                assert np.array_equal(np.delete(obs[n, a], [1, 2]), np.delete(exp, [1, 2])), (tag, n, a, obs[n, a], exp)
                assert np.allclose(obs[n, a, 1:3], exp[1:3], rtol=0, atol=1e-9), (tag, n, a, obs[n, a, 1:3], exp[1:3])
            else:
                assert np.array_equal(obs[n, a], exp), (tag, n, a, obs[n, a], exp)
            assert rew[n, a] == np.float32(r["individual_reward"][a]), (tag, n, a, rew[n, a], r["individual_reward"][a])
            assert bool(done[n, a]) == bool(d[a]), (tag, n, a)
            assert bool(oob[n, a]) == i[a]["out_of_bounds"] and int(oobc[n, a]) == i[a]["out_of_bounds_count"], (tag, n, a)
            assert bool(blk[n, a]) == i[a]["blocked"], (tag, n, a)
            ag = env.agents[a]
            assert bool(col[n, a]) == ag.collision, (tag, n, a)
            assert (x[a, n], y[a, n]) == ag.det, (tag, n, a)
            assert sp[a, n] == ag.sp_dist and prev[a, n] == ag.prev_det_dist, (tag, n, a, sp[a, n], ag.sp_dist)
        assert team[n] == np.float32(r["team_reward"]), (tag, n)


def _run(N, A, obst, enforce, steps, group=1, ep_len=25, base=0, falloff="reference", seed=0, allow_err=0, **opts):
    vec, specs, geom = _make(N, A, obst, enforce, group, base, falloff, **opts)
    refs = [RadSearchOracle(PhiloxDraws(SEED, base + n), number_agents=A, obstruction_count=obst,
                            enforce_grid_boundaries=enforce, falloff=falloff, layout_fn=fn, **opts) for n, fn in specs]
    outs = vec.reset()
    torch.cuda.synchronize()
    _compare(vec, outs, refs, [e._ret for e in refs], "reset0")
    rng = np.random.default_rng(seed)
    t_in_ep = np.zeros(N, dtype=np.int64)
    for t in range(steps):
        acts = rng.integers(0, 9, size=(N, A)).astype(np.int8)
        if t % 11 == 5:
            acts[:, 0] = -1                      # -1 == idle (rad_search_env.py:620-623)
        outs = vec.step(torch.from_numpy(acts).cuda())
        rets = [e.step({a: (8 if acts[n, a] == -1 else int(acts[n, a])) for a in range(A)}) for n, e in enumerate(refs)]
        torch.cuda.synchronize()
        _compare(vec, outs, refs, rets, f"step{t}")
        t_in_ep += 1
        mask = np.array([e.done for e in refs]) | (t_in_ep >= ep_len)
        if t % 17 == 16:                          # epoch boundary: everything resets with epoch_end set
            mask[:] = True
            vec.set_epoch_end()
            for e in refs:
                e.epoch_end = True
            for st in geom.values():
                st["want"] += 1
        if mask.any():
            outs = vec.reset(torch.from_numpy(mask.astype(np.uint8)).cuda())
            rets = [e.reset() if mask[n] else None for n, e in enumerate(refs)]
            torch.cuda.synchronize()
            _compare(vec, outs, refs, rets, f"reset@{t}")
            t_in_ep[mask] = 0
    ref_err = 0
    for e in refs:
        ref_err |= e.err
    assert vec.error_flags() == ref_err and (ref_err & ~allow_err) == 0, (vec.error_flags(), ref_err)
    return refs


def test_obstacle_free_single_agent_enforced():
    _run(N=200, A=1, obst=0, enforce=True, steps=70)


@pytest.mark.parametrize("A,obst", [(1, 0), (3, 2)])
def test_debug_spawn(A, obst):
    """DEBUG=True (rad_search_env.py:387-389, :782-785, :1043-1090; the oracle's restatement is pinned to the reference by
    tests/golden/envopt_debug_a1.npz): source (500, 500), detector (1000, 1000) unless it falls into a rectangle, no spawn resampling,
    intensity 1e6, background 0 -- bit-exact like every other env output.  With rectangles the fixed source may lie inside one (the
    reference resamples nothing under DEBUG; visilibity is then undefined): kernel and oracle raise the same RS_ENVERR_NO_PATH bit."""
    from radiation_ppo_amd import _lib
    refs = _run(N=64, A=A, obst=obst, enforce=True, steps=40, seed=5, allow_err=_lib.ENVERR_NO_PATH if obst else 0, DEBUG=True)
    assert all(e.src == (500, 500) and e.intensity == 1000000 and e.bkg_intensity == 0 for e in refs)


@pytest.mark.parametrize("A", [1, 2])
def test_coord_noise(A):
    """coord_noise=True (rad_search_env.py:365, :569-580; oracle pinned by tests/golden/envopt_noise_a*.npz): N(0, 5 cm) on the
    observation's two coordinates, nothing else.  Kernel and oracle draw the same Philox block and apply the same Box-Muller formula
    in float64, but through different log / sincos libraries (device ocml vs the host libm): the two float32 coordinates are held to
    1e-9 (scaled units; a last-bit difference of the float64 noise moves them by ~1e-19), everything else bit-exact.  The noise as a
    DISTRIBUTION (like the Poisson sampler): mean 0, standard deviation 5 cm, uncorrelated axes, normal (Kolmogorov-Smirnov)."""
    global NOISY
    NOISY = True
    try:
        _run(N=96, A=A, obst=0, enforce=True, steps=40, seed=7, coord_noise=True)
    finally:
        NOISY = False
    from scipy import stats
    from radiation_ppo_amd.envs import RadSearchVec
    N = 1 << 16
    vec = RadSearchVec(N, number_agents=A, obstruction_count=0, enforce_grid_boundaries=True, seed=11, coord_noise=True)
    vec.reset()
    idle = torch.full((N, A), 8, dtype=torch.int8, device="cuda")
    zs = []
    for _ in range(4):
        obs = vec.step(idle)[0]
        x = torch.stack([vec.state("x").T.double(), vec.state("y").T.double()], dim=-1)          # [N, A, 2] true lattice coordinates
        zs.append(((obs[..., 1:3].double() * 2200.0 - x) / 5.0).cpu().numpy().reshape(-1, 2))
    z = np.concatenate(zs)
    n = z.shape[0]
    assert abs(z.mean(axis=0)).max() < 5.0 / np.sqrt(n) and abs(z.std(axis=0) - 1.0).max() < 5.0 / np.sqrt(2 * n) + 2e-4
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 5.0 / np.sqrt(n)
    assert stats.kstest(z[:200000, 0], "norm").pvalue > 1e-4 and stats.kstest(z[:200000, 1], "norm").pvalue > 1e-4
    assert abs(np.corrcoef(z[:-A, 0], z[A:, 0])[0, 1]) < 5.0 / np.sqrt(n)             # consecutive (env, agent) streams are independent


def test_obstacle_free_single_agent_unenforced():
    _run(N=130, A=1, obst=0, enforce=False, steps=50)


@pytest.mark.parametrize("A", [2, 4])
def test_obstacle_free_multi_agent(A):
    _run(N=96, A=A, obst=0, enforce=True, steps=60)


def test_obstacles_single_agent():
    _run(N=96, A=1, obst=5, enforce=True, steps=60, seed=3)


def test_obstacles_random_count_multi_agent():
    _run(N=64, A=3, obst=-1, enforce=True, steps=50, seed=4)


def test_obstacles_max_count_unenforced():
    _run(N=40, A=2, obst=7, enforce=False, steps=40, seed=5)


def test_shared_layout_per_wave():
    _run(N=128, A=1, obst=4, enforce=True, steps=40, group=64, seed=6)


def test_shared_layout_small_groups():
    _run(N=48, A=2, obst=3, enforce=True, steps=36, group=4, seed=7)


def test_env_id_base_sharding_equivalence():
    """Envs are keyed by GLOBAL id: a rank that owns envs [64,128) reproduces the second half of a
    128-env job exactly (multi-GPU sharding does not change results)."""
    from radiation_ppo_amd.envs import RadSearchVec
    full = RadSearchVec(128, obstruction_count=3, enforce_grid_boundaries=True, seed=SEED)
    half = RadSearchVec(64, obstruction_count=3, enforce_grid_boundaries=True, seed=SEED, env_id_base=64)
    full.reset(); half.reset()
    rng = np.random.default_rng(9)
    for t in range(30):
        acts = rng.integers(0, 9, size=(128, 1)).astype(np.int8)
        of = full.step(torch.from_numpy(acts).cuda())
        oh = half.step(torch.from_numpy(acts[64:].copy()).cuda())
        assert torch.equal(of[0][64:], oh[0]) and torch.equal(of[1][64:], oh[1]) and torch.equal(of[3][64:], oh[3])


def test_inverse_square_falloff_option():
    _run(N=64, A=1, obst=0, enforce=True, steps=30, falloff="inverse_square")


def test_bad_action_flag_and_adapter_errors():
    from radiation_ppo_amd.envs import RadSearchVec
    vec = RadSearchVec(8, enforce_grid_boundaries=True, seed=1)
    vec.reset()
    a = torch.full((8, 1), 12, dtype=torch.int8, device="cuda")
    vec.step(a)
    assert vec.error_flags() & 8


def test_dict_api_adapter_matches_oracle():
    """The gym-style adapter (RadSearch) returns the reference's 4-tuple of dicts."""
    from radiation_ppo_amd.envs import RadSearch
    env = RadSearch(number_agents=2, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED)
    ref = RadSearchOracle(PhiloxDraws(SEED, 0), number_agents=2, obstruction_count=0, enforce_grid_boundaries=True)
    assert env.search_area == ((200.0, 200.0), (2200.0, 200.0), (2200.0, 2200.0), (200.0, 2200.0))
    assert env.observation_space.shape == (11,) and env.number_actions == 9 and env.scale == 1 / 2200.0
    assert env.src_coords == (float(ref.src[0]), float(ref.src[1]))
    rng = np.random.default_rng(1)
    for t in range(40):
        act = {0: int(rng.integers(0, 9)), 1: int(rng.integers(0, 9))}
        o, r, d, i = env.step(act)
        ro, rr, rd, ri = ref.step(act)
        for a in range(2):
            assert np.array_equal(o[a].astype(np.float32), np.asarray(ro[a]).astype(np.float32))
            assert r["individual_reward"][a] == rr["individual_reward"][a]
            assert d[a] == rd[a] and i[a] == ri[a]
        assert r["team_reward"] == rr["team_reward"]
        if t % 13 == 12:
            env.epoch_end = True
            ref.epoch_end = True
            o, r, d, i = env.reset()
            ro, rr, rd, ri = ref.reset()
            assert np.array_equal(o[0].astype(np.float32), np.asarray(ro[0]).astype(np.float32))


@pytest.mark.parametrize("A,obst", [(2, 0), (3, 3)])
def test_adapter_int_and_none_call_forms_match_oracle(A, obst):
    """RadSearch.step(int) for several agents (same action for all, NO collision rule: the agents stay stacked on one
    cell) and step(None) mid-episode (rad_search_env.py:616-627, :676-690); the oracle's handling of both forms is
    pinned to the reference by tests/golden/envforms_*.npz."""
    from radiation_ppo_amd.envs import RadSearch
    env = RadSearch(number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True, seed=SEED)
    ref = RadSearchOracle(PhiloxDraws(SEED, 0), number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True)
    rng = np.random.default_rng(3)
    steps = 0
    for t in range(60):
        m = t % 6
        act = int(rng.integers(0, 8)) if m == 1 else (-1 if m == 4 else (None if m == 3 else {i: int(rng.integers(0, 9)) for i in range(A)}))
        o, r, d, i = env.step(act)
        ro, rr, rd, ri = ref.step(act)
        for a in range(A):
            assert np.array_equal(o[a].astype(np.float32), np.asarray(ro[a]).astype(np.float32)), (t, a)
            assert r["individual_reward"][a] == rr["individual_reward"][a], (t, a)
            assert d[a] == rd[a] and i[a] == ri[a], (t, a)
        assert r["team_reward"] == rr["team_reward"], t
        steps += 1
        if ref.done or steps == 20:
            o, r, d, i = env.reset()
            ro, rr, rd, ri = ref.reset()
            assert np.array_equal(o[0].astype(np.float32), np.asarray(ro[0]).astype(np.float32))
            steps = 0


def test_single_env_and_max_agents():
    """Edge sizes: one env (a 1-lane wave) and the maximum agent count (RS_MAX_AGENTS = 8) with collisions galore."""
    _run(N=1, A=1, obst=0, enforce=True, steps=30, seed=11)
    _run(N=3, A=8, obst=2, enforce=True, steps=24, ep_len=8, seed=12)


def test_full_size_properties_no_oracle():
    """BASELINE sizes (4096 envs and 2^18 envs) through size-independent properties: determinism (same seed -> same
    bits), independence from the launch shape (a 4096-env handle equals the first 4096 envs of a 2^18-env handle),
    integer-valued measurements, bounded coordinates and rewards on the 2-decimal lattice."""
    from radiation_ppo_amd.envs import RadSearchVec
    big = RadSearchVec(1 << 18, enforce_grid_boundaries=True, seed=SEED)
    a = RadSearchVec(4096, enforce_grid_boundaries=True, seed=SEED)
    b = RadSearchVec(4096, enforce_grid_boundaries=True, seed=SEED)
    big.reset(); a.reset(); b.reset()
    g = torch.Generator().manual_seed(0)
    for t in range(20):
        acts = torch.randint(0, 9, (1 << 18, 1), generator=g).to(torch.int8).cuda()
        ob, rb, _, db, _ = big.step(acts)
        oa, ra, _, da, _ = a.step(acts[:4096].contiguous())
        o2, r2, _, d2, _ = b.step(acts[:4096].contiguous())
        assert torch.equal(oa, o2) and torch.equal(ra, r2) and torch.equal(da, d2)
        assert torch.equal(ob[:4096], oa) and torch.equal(rb[:4096], ra) and torch.equal(db[:4096], da)
        assert torch.equal(ob[..., 0], ob[..., 0].round()) and (ob[..., 0] >= 0).all()
        assert (ob[..., 1:3] >= 0).all() and (ob[..., 1:3] < 2700 / 2200).all()
        assert torch.allclose(rb * 100, (rb * 100).round(), atol=1e-4) and (rb <= 0.1 + 1e-6).all()
    assert big.error_flags() == 0


def test_line_of_sight_threshold_no_int64_overflow():
    """is_intersect's exact 'distance < 1e-3' test (cr^2 * 1e6 < len2) with a corner 1800 cm off a 2827 cm line
    of sight: cr = 3.6e6, cr^2 * 1e6 = 1.3e19 exceeds int64.  The state is written directly (rs_state_field) so
    the geometry is exactly the adversarial one; an idle step must see the source (not blocked)."""
    from radiation_ppo_amd.envs import RadSearchVec
    from oracle.radsearch_oracle import seg_rect_boundary_lt_1e3
    vec = RadSearchVec(64, obstruction_count=1, enforce_grid_boundaries=True, seed=SEED)
    vec.reset()
    rect = (1900, 300, 2100, 500)
    assert not seg_rect_boundary_lt_1e3(200, 200, 2199, 2199, rect)          # oracle: far from blocked
    r = vec.state("rect"); r[0:4, :] = torch.tensor(rect, dtype=torch.int32, device="cuda").view(4, 1)
    vec.state("num_obs")[:] = 1
    vec.state("src_x")[:] = 2199; vec.state("src_y")[:] = 2199
    vec.state("x")[:] = 200; vec.state("y")[:] = 200
    d = float(np.sqrt(2.0) * 1999)
    vec.state("sp")[:] = d; vec.state("prev")[:] = d
    vec.state("dsrc")[:] = 1e9
    obs, rew, _, _, _ = vec.step(torch.full((64, 1), 8, dtype=torch.int8, device="cuda"))
    flags = vec.state("aflags").cpu().numpy()
    assert (flags & 2).sum() == 0, "line of sight wrongly reported blocked (int64 overflow in the threshold test)"
    inten = vec.state("intensity").cpu().numpy()[0]
    assert (obs[:, 0, 0].cpu().numpy() > 0.5 * inten / d).all()


@pytest.mark.parametrize("A,base,N", [(1, 190, 40), (2, 3815, 20)])
def test_nested_layout_is_rejected_like_world_is_valid(A, base, N):
    """Env ids 199, 215 / 3828, 3829 draw a rectangle nested inside another as their first layout:
    world.is_valid fails (rad_search_env.py:788-791), the reset is redone with fresh draws and one more
    step(None) runs.  Without the retry the detector can land on the inner boundary inside the outer
    rectangle, where no path exists and the reward is -inf."""
    refs = _run(N=N, A=A, obst=-1, enforce=True, steps=40, base=base)
    assert sum(e.invalid_layouts for e in refs) >= 2


@pytest.mark.parametrize("A,with_rects", [(1, False), (2, False), (1, True), (3, True)])
def test_refresh_environment_matches_oracle(A, with_rects):
    """rs_refresh (RadSearch.refresh_environment, rad_search_env.py:799-874): saved source / detector / intensity /
    background (and obstacle rectangles) loaded into running envs, then stepped, bit-exact against the oracle --
    including the stale sp_dist the reference carries over and iter_count = 1."""
    N, obst = 48, (3 if with_rects else 0)
    vec, specs, _ = _make(N, A, obst, True)
    refs = [RadSearchOracle(PhiloxDraws(SEED, n), number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True)
            for n, _ in specs]
    vec.reset()
    rng = np.random.default_rng(11)
    for t in range(7):                                   # leave the envs mid-episode with some history
        acts = rng.integers(0, 9, size=(N, A)).astype(np.int8)
        vec.step(torch.from_numpy(acts).cuda())
        for n, e in enumerate(refs):
            e.step({a: int(acts[n, a]) for a in range(A)})
    for rnd in range(3):
        mask = rng.random(N) < 0.7
        src = rng.integers(200, 2200, size=(N, 2)).astype(np.int32)
        det = rng.integers(200, 2200, size=(N, 2)).astype(np.int32)
        det[0] = (205, 1900)                             # first step (action 0) runs into the left wall: stale sp_dist is used
        inten = rng.integers(1_000_000, 10_000_000, size=N).astype(np.int32)
        bkg = rng.integers(10, 51, size=N).astype(np.int32)
        nob = rects = None
        if with_rects:
            nob = rng.integers(0, 4, size=N).astype(np.int32)
            rects = np.zeros((N, 7, 4), dtype=np.int32)
            for n in range(N):
                for i in range(nob[n]):                  # disjoint columns, away from the sampled points' lattice rows
                    x0 = 300 + 600 * i + int(rng.integers(0, 50))
                    y0 = int(rng.integers(300, 1500))
                    rects[n, i] = (x0, y0, x0 + int(rng.integers(200, 400)), y0 + int(rng.integers(200, 500)))
                for p in (src, det):                     # keep the saved points out of the rectangles, as saved sets are
                    while any(rects[n, i, 0] <= p[n, 0] <= rects[n, i, 2] and rects[n, i, 1] <= p[n, 1] <= rects[n, i, 3]
                              for i in range(nob[n])):
                        p[n] = rng.integers(200, 2200, size=2)
        dev = lambda a: None if a is None else torch.from_numpy(a).cuda()
        outs = vec.refresh(dev(src), dev(det), dev(inten), dev(bkg), dev(nob), dev(rects), mask=torch.from_numpy(mask.astype(np.uint8)).cuda())
        rets = []
        for n, e in enumerate(refs):
            if not mask[n]:
                rets.append(None)
                continue
            o = e.refresh_environment(src[n], det[n], inten[n], bkg[n], None if rects is None else rects[n, :nob[n]])
            rets.append(o)
        torch.cuda.synchronize()
        obs = outs[0].cpu().numpy()
        it = vec.state("iter_count").cpu().numpy()[0]
        sp, prev = vec.state("sp").cpu().numpy(), vec.state("prev").cpu().numpy()
        for n, e in enumerate(refs):
            if rets[n] is None:
                continue
            assert it[n] == 1 == e.iter_count
            for a in range(A):
                assert np.array_equal(obs[n, a], np.asarray(rets[n][a], dtype=np.float64).astype(np.float32)), (rnd, n, a)
                assert sp[a, n] == e.agents[a].sp_dist and prev[a, n] == e.agents[a].prev_det_dist, (rnd, n, a)
        for t in range(6):
            acts = rng.integers(0, 9, size=(N, A)).astype(np.int8)
            if t == 0:
                acts[:, :] = 0
            o2 = vec.step(torch.from_numpy(acts).cuda())
            r2 = [e.step({a: int(acts[n, a]) for a in range(A)}) for n, e in enumerate(refs)]
            torch.cuda.synchronize()
            _compare(vec, o2, refs, r2, f"refresh{rnd}.step{t}")
    assert vec.error_flags() == 0 and all(e.err == 0 for e in refs)


def test_adapter_refresh_environment_reads_saved_env_dicts():
    """The gym-style adapter takes the saved-set format of algos/test_environment/eval/test_env_gen.py:13-24."""
    from radiation_ppo_amd.envs import RadSearch
    env = RadSearch(number_agents=1, obstruction_count=2, enforce_grid_boundaries=True, seed=5)
    corners = lambda x0, y0, x1, y1: [np.array([[x0, y0], [x0, y1], [x1, y1], [x1, y0]], dtype=np.float64)]
    d = {"env_3": (np.array([1500.0, 700.0]), np.array([400.0, 1900.0]), 4321000, 37,
                   [corners(800, 900, 1100, 1300), corners(1600, 1500, 1900, 1800)])}
    obs = env.refresh_environment(d, 3, num_obs=2)
    assert env.src_coords == (1500.0, 700.0) and env.intensity == 4321000 and env.bkg_intensity == 37
    assert env.num_obs == 2 and env.iter_count == 1
    assert obs[0].shape == (11,) and abs(obs[0][1] - 400 / 2200) < 1e-7 and abs(obs[0][2] - 1900 / 2200) < 1e-7
    o, r, dn, info = env.step({0: 4})
    assert abs(o[0][1] - 500 / 2200) < 1e-7
    with pytest.raises(ValueError):
        env.refresh_environment({"env_0": (np.array([1500.5, 700.0]), np.array([400.0, 1900.0]), 1, 1)}, 0)
