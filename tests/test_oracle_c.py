"""The C restatement of the env (oracle/radsearch_oracle.c), obstacle-free and with obstructions, event by event against the
Python oracle (which is pinned to golden vectors from the real reference; its obstacle geometry restates the un-vendored
visilibity calls and is "parity unpinned"): float64-exact observations, rewards, positions, flags, done."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.radsearch_oracle import PhiloxDraws, RadSearchOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "librs_oracle.so"))
    lib.rso_create.restype = C.c_void_p
    lib.rso_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]
    lib.rso_destroy.argtypes = [C.c_void_p]
    lib.rso_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rso_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rso_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rso_create2.restype = C.c_void_p
    lib.rso_create2.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rso_set_epoch_end.argtypes = [C.c_void_p]
    lib.rso_state2.restype = C.c_int
    lib.rso_state2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rso_bench2.restype = C.c_long
    lib.rso_bench2.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rso_bench.restype = C.c_long
    lib.rso_bench.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_long, C.c_int]
    return lib


@pytest.mark.parametrize("A,enforce", [(1, 1), (1, 0), (3, 1), (4, 0)])
def test_c_oracle_equals_python_oracle(lib, A, enforce):
    seed = 289714752
    for env_id in (0, 7):
        e = lib.rso_create(seed, env_id, A, enforce, 0)
        ref = RadSearchOracle(PhiloxDraws(seed, env_id), number_agents=A, obstruction_count=0, enforce_grid_boundaries=bool(enforce))
        obs = np.zeros((A, 11)); rew = np.zeros(A); team = C.c_double(); done = np.zeros(A, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        lib.rso_reset(e, p(obs), p(rew), C.byref(team), p(done))
        ret = ref._ret
        rng = np.random.default_rng(A + env_id)
        steps = 0
        for t in range(400):
            for a in range(A):
                assert np.array_equal(obs[a], np.asarray(ret[0][a], dtype=np.float64)), (t, a)
                assert rew[a] == ret[1]["individual_reward"][a] and bool(done[a]) == ret[2][a]
            assert team.value == ret[1]["team_reward"]
            acts = rng.integers(0, 9, size=A).astype(np.int32)
            lib.rso_step(e, p(acts), p(obs), p(rew), C.byref(team), p(done))
            ret = ref.step({a: int(acts[a]) for a in range(A)})
            steps += 1
            if ref.done or steps == 25:
                lib.rso_reset(e, p(obs), p(rew), C.byref(team), p(done))
                ret = ref.reset()
                steps = 0
        lib.rso_destroy(e)


@pytest.mark.parametrize("A,enforce,count", [(1, 1, -1), (1, 0, 3), (2, 1, 7), (4, 1, -1), (3, 0, 5)])
def test_c_oracle_with_obstructions_equals_python_oracle(lib, A, enforce, count):
    """configs[2..4]: random rectangles (count -1 = U{1..5} per epoch), a new layout every 3 episodes (epoch_end), episodes of
    30 steps; biased walks so that agents run into rectangles (blocked moves, sensors, correct_coords, shadowed readings)."""
    seed = 289714752
    seen = {"blocked": 0, "inter": 0, "sensor": 0, "layouts": set()}
    for env_id in (0, 3, 11):
        e = lib.rso_create2(seed, env_id, A, enforce, 0, count)
        ref = RadSearchOracle(PhiloxDraws(seed, env_id), number_agents=A, obstruction_count=count, enforce_grid_boundaries=bool(enforce))
        obs = np.zeros((A, 11)); rew = np.zeros(A); team = C.c_double(); done = np.zeros(A, dtype=np.int32)
        xy = np.zeros((A, 2), dtype=np.int32); sp = np.zeros(A); prev = np.zeros(A); misc = np.zeros(6, dtype=np.int32)
        flags = np.zeros((A, 5), dtype=np.int32); rects = np.zeros((7, 4), dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        lib.rso_reset(e, p(obs), p(rew), C.byref(team), p(done))
        ret = ref._ret
        rng = np.random.default_rng(17 * A + env_id)
        steps = episodes = 0
        drift = rng.integers(0, 8, size=A)
        for t in range(700):
            lib.rso_state(e, p(xy), p(sp), p(prev), p(misc))
            n = lib.rso_state2(e, p(flags), p(rects))
            assert n == ref.num_obs and [tuple(r) for r in rects[:n].tolist()] == [tuple(r) for r in ref.rects], t
            seen["layouts"].add(tuple(map(tuple, rects[:n].tolist())))
            assert (misc[0], misc[1]) == tuple(ref.src) and misc[2] == ref.intensity and misc[3] == ref.bkg_intensity
            assert bool(misc[4]) == ref.done and misc[5] == ref.err, (t, misc[5], ref.err)
            for a in range(A):
                ag = ref.agents[a]
                assert np.array_equal(obs[a], np.asarray(ret[0][a], dtype=np.float64)), (t, a, obs[a], ret[0][a])
                assert rew[a] == ret[1]["individual_reward"][a] and bool(done[a]) == ret[2][a], (t, a)
                assert tuple(xy[a]) == tuple(ag.det) and sp[a] == ag.sp_dist and prev[a] == ag.prev_det_dist, (t, a)
                assert tuple(flags[a]) == (int(ag.out_of_bounds), ag.out_of_bounds_count, int(ag.obstacle_blocking), int(ag.collision), int(ag.intersect)), (t, a)
                seen["blocked"] += int(ag.obstacle_blocking); seen["inter"] += int(ag.intersect); seen["sensor"] += int(any(v > 0 for v in ret[0][a][3:]))
            assert team.value == ret[1]["team_reward"]
            acts = np.where(rng.random(A) < 0.7, drift, rng.integers(0, 9, size=A)).astype(np.int32)
            lib.rso_step(e, p(acts), p(obs), p(rew), C.byref(team), p(done))
            ret = ref.step({a: int(acts[a]) for a in range(A)})
            steps += 1
            if ref.done or steps == 30:
                episodes += 1
                if episodes % 3 == 0:
                    lib.rso_set_epoch_end(e); ref.epoch_end = True
                lib.rso_reset(e, p(obs), p(rew), C.byref(team), p(done))
                ret = ref.reset()
                steps = 0
                drift = rng.integers(0, 8, size=A)
        lib.rso_destroy(e)
    assert seen["blocked"] > 0 and seen["inter"] > 0 and seen["sensor"] > 0 and len(seen["layouts"]) > 6, seen


def test_c_oracle_rejected_layouts_and_corrected_sensors(lib, monkeypatch):
    """Seven rectangles, a new layout every episode: layouts rejected by world.is_valid (the nested reset with its extra idle
    measurement, rad_search_env.py:788-791) and detectors wedged in a corner (correct_coords, :1263-1306) both occur."""
    from oracle import radsearch_oracle as ro
    calls = {"n": 0}
    orig = ro.RadSearchOracle._correct_coords

    def counted(self, r, agent):
        calls["n"] += 1
        return orig(self, r, agent)
    monkeypatch.setattr(ro.RadSearchOracle, "_correct_coords", counted)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    invalid = 0
    for env_id in range(12):
        e = lib.rso_create2(5, env_id, 1, 1, 0, 7)
        ref = RadSearchOracle(PhiloxDraws(5, env_id), number_agents=1, obstruction_count=7, enforce_grid_boundaries=True)
        obs = np.zeros((1, 11)); rew = np.zeros(1); team = C.c_double(); done = np.zeros(1, dtype=np.int32)
        lib.rso_reset(e, p(obs), p(rew), C.byref(team), p(done))
        ret = ref._ret
        rng = np.random.default_rng(env_id)
        for ep in range(12):
            assert np.array_equal(obs[0], np.asarray(ret[0][0])), (env_id, ep)
            drift = int(rng.integers(0, 8))
            for t in range(40):
                acts = np.array([drift if rng.random() < 0.8 else rng.integers(0, 9)], dtype=np.int32)
                lib.rso_step(e, p(acts), p(obs), p(rew), C.byref(team), p(done))
                ret = ref.step({0: int(acts[0])})
                assert np.array_equal(obs[0], np.asarray(ret[0][0])) and rew[0] == ret[1]["individual_reward"][0], (env_id, ep, t)
                if ref.done:
                    break
            lib.rso_set_epoch_end(e); ref.epoch_end = True
            lib.rso_reset(e, p(obs), p(rew), C.byref(team), p(done))
            ret = ref.reset()
        invalid += ref.invalid_layouts
        lib.rso_destroy(e)
    assert calls["n"] > 0 and invalid > 0, (calls, invalid)


def test_c_oracle_bench_loop_runs(lib):
    assert lib.rso_bench(289714752, 0, 8, 20000, 120) >= 20000
    assert lib.rso_bench2(289714752, 0, 8, 20000, 120, 480, -1, 1) >= 20000
    assert lib.rso_bench2(289714752, 0, 8, 5000, 120, 480, 7, 4) >= 5000
