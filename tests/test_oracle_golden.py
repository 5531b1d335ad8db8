"""The oracle (oracle/radsearch_oracle.py) replayed against golden vectors captured from the REAL
reference (tests/golden/make_golden.py).  Every RNG draw the reference made is replayed with its
arguments checked, so the deterministic maps (state, action, draws) -> (state', obs, reward, done)
are pinned exactly: float64 equality, not a tolerance.  Mirrors SURVEY.md section 8c items (1),(2)."""
import glob
import os

import numpy as np
import pytest

from oracle.radsearch_oracle import ACTION_STEP, RadSearchOracle, ReplayDraws

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "env_*.npz")))


def test_action_table(golden_dir):
    table = np.load(os.path.join(golden_dir, "action_table.npz"))["table"]
    assert table.shape == (9, 2)
    assert np.array_equal(table, np.array(ACTION_STEP, dtype=np.float64))


FORM_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "envforms_*.npz")))
OPT_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "envopt_*.npz")))


@pytest.mark.parametrize("path", FILES + FORM_FILES + OPT_FILES, ids=[os.path.basename(f)[:-4] for f in FILES + FORM_FILES + OPT_FILES])
def test_env_transitions_exact(path):
    """env_*: dict actions (and the single-int form for one agent).  envforms_*: step(int) for SEVERAL agents (same
    action for all, no collision rule) and step(None) mid-episode (rad_search_env.py:616-627, :676-690)."""
    g = dict(np.load(path).items())
    form = g.get("form")
    seed, A, enforce = (int(v) for v in g["meta"][:3])
    noise, debug = (bool(v) for v in g["meta"][3:5]) if len(g["meta"]) >= 5 else (False, False)      # envopt_*: coord_noise / DEBUG
    draws = ReplayDraws([(int(k), a0, a1, v) for (_, k, a0, a1, v) in g["draws"]])
    n_events = len(g["is_reset"])
    # the fixture's first event is a reset with epoch_end=True: the constructor's reset plays it
    env = RadSearchOracle(draws, number_agents=A, obstruction_count=0, enforce_grid_boundaries=bool(enforce), coord_noise=noise, DEBUG=debug)
    ret = env._ret
    steps_in_ep = 0
    for e in range(n_events):
        if e > 0:
            if g["is_reset"][e]:
                # make_golden.py sets epoch_end on even t before some resets; with obstruction_count=0
                # the flag changes no draw, so replay does not need to know
                env.epoch_end = True
                ret = env.reset()
            else:
                acts = [int(a) for a in g["actions"][e]]
                f = 0 if form is None else int(form[e])
                ret = env.step({i: acts[i] for i in range(A)}) if f == 0 else env.step(acts[0] if f == 1 else None)
        obs, rew, done, info = ret
        for i in range(A):
            ag = env.agents[i]
            assert np.array_equal(np.asarray(obs[i], dtype=np.float64), g["obs"][e, i]), (e, i, obs[i], g["obs"][e, i])
            assert rew["individual_reward"][i] == g["reward"][e, i], (e, i)
            assert bool(done[i]) == bool(g["done_ret"][e, i]), (e, i)
            assert info[i]["out_of_bounds"] == bool(g["info_oob"][e, i])
            assert info[i]["out_of_bounds_count"] == int(g["info_oobc"][e, i])
            assert info[i]["blocked"] == bool(g["info_blocked"][e, i])
            assert (float(ag.det[0]), float(ag.det[1])) == tuple(g["det"][e, i]), (e, i)
            assert ag.sp_dist == g["sp"][e, i], (e, i)
            assert ag.euc_dist == g["euc"][e, i], (e, i)
            assert ag.prev_det_dist == g["prev"][e, i], (e, i)
            assert ag.collision == bool(g["coll"][e, i]), (e, i)
            assert ag.intersect == bool(g["inter"][e, i])
        assert rew["team_reward"] == g["team"][e], e
        assert env.done == bool(g["done"][e])
        assert env.iter_count == int(g["iter_count"][e])
        assert (float(env.src[0]), float(env.src[1])) == tuple(g["src"][e])
        assert env.intensity == int(g["intensity"][e]) and env.bkg_intensity == int(g["bkg"][e])
    assert draws.pos == len(draws.rows), "oracle consumed a different number of draws than the reference"
    assert env.err == 0
