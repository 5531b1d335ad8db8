"""The C restatement of the obstacle-free env (oracle/radsearch_oracle.c) event by event against the Python oracle
(which is pinned to golden vectors from the real reference): float64-exact observations, rewards, positions, done."""
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


def test_c_oracle_bench_loop_runs(lib):
    assert lib.rso_bench(289714752, 0, 8, 20000, 120) >= 20000
