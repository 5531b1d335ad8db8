"""the safe reader of the reference's saved test-environment sets (radiation_ppo_amd/testsets.py).
It must (1) return exactly what joblib's own loader returns for a file of the reference's structure written here,
(2) execute nothing: a file that smuggles a callable is rejected before anything runs, (3) in this container, parse every
set the reference ships (1000 environments each, lattice coordinates, obstruction counts as the file name says)."""
import glob
import os
import pickle

import numpy as np
import pytest

from radiation_ppo_amd.testsets import load_test_environments, summarize_test_set


def _make(n, obstacles, rng):
    d = {}
    for i in range(n):
        e = [rng.integers(200, 2200, 2).astype(np.float64), rng.integers(200, 2200, 2).astype(np.float64),
             np.int64(rng.integers(10 ** 6, 10 ** 7)), np.int64(rng.integers(10, 51))]
        if obstacles:
            obs = []
            for _ in range(obstacles):
                x0, y0 = rng.integers(200, 1900, 2)
                w, h = rng.integers(200, 500, 2)
                obs.append([np.array([[x0, y0], [x0, y0 + h], [x0 + w, y0 + h], [x0 + w, y0]], dtype=np.float64)])
            e.append(obs)
        d[f"env_{i}"] = tuple(e)
    return d


@pytest.mark.parametrize("obstacles", [0, 3])
def test_reader_equals_joblib_loader_on_our_own_file(tmp_path, obstacles):
    joblib = pytest.importorskip("joblib")
    d = _make(25, obstacles, np.random.default_rng(4))
    path = str(tmp_path / "set")
    joblib.dump(d, path)                                    # trusted: written two lines above
    want = joblib.load(path)
    got = load_test_environments(path)
    assert list(got) == list(want)
    for k in want:
        a, b = got[k], want[k]
        assert len(a) == len(b)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[0].dtype == np.float64
        assert int(a[2]) == int(b[2]) and int(a[3]) == int(b[3])
        if obstacles:
            assert len(a[4]) == obstacles and all(np.array_equal(x[0], y[0]) for x, y in zip(a[4], b[4]))
    s = summarize_test_set(got)
    assert s["count"] == 25 and s["obstructions"] == (obstacles, obstacles)


def test_reader_executes_nothing(tmp_path):
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))
    p = tmp_path / "evil"
    p.write_bytes(pickle.dumps({"env_0": (np.zeros(2), np.zeros(2), 1, 1, Evil())}, protocol=3))
    with pytest.raises(ValueError):
        load_test_environments(str(p))
    assert not marker.exists()
    p.write_bytes(pickle.dumps({"env_0": (np.zeros(2), np.zeros(2), 1, 1)}, protocol=3))    # plain numpy pickling is not joblib's format
    with pytest.raises(ValueError):
        load_test_environments(str(p))
    assert not marker.exists()


REF_SETS = sorted(glob.glob("/root/reference/algos/multiagent/evaluation/test_environments/test_env_dict_obs*_v4"))


@pytest.mark.skipif(not REF_SETS, reason="the reference tree exists only in the build container")
def test_every_reference_set_parses():
    assert len(REF_SETS) >= 30
    for f in REF_SETS[::6] + [x for x in REF_SETS if "obs1_none" in x]:      # a sample keeps the CPU suite short; scripts parse all
        d = load_test_environments(f)
        k = int(os.path.basename(f).split("_obs")[1].split("_")[0])
        s = summarize_test_set(d)
        assert s["count"] in (100, 1000) and s["obstructions"] == (k, k) and s["min_start_distance"] >= 1000.0
        e = d["env_0"]
        assert e[0].shape == (2,) and np.all(e[0] == np.round(e[0])) and np.all(e[1] == np.round(e[1]))
        assert 10 ** 6 <= e[2] < 10 ** 7 and 10 <= e[3] <= 50
