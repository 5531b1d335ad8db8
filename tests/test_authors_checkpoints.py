"""Row f3's purpose: the AUTHORS' saved models.  Fixtures: tests/golden/ckpt_rada2c_*.npz (their four RAD-A2C
pyt_save/model.pt files, read with torch.load(weights_only=True) by tests/golden/make_checkpoints.py), ckpt_radteam_shapes.json (key /
shape tables of their RAD-TEAM actor / critic / predictor files), testset_obs<k>_<snr>.npz (the first 100 environments of the
reference's saved test sets, read without unpickling) and ckpt_rada2c_authors_log.json (the last 100 rows of their progress.txt).

CPU: the RAD-A2C checkpoints load STRICTLY into this build's RNNModelActorCritic (same names, same shapes); the RAD-TEAM files are
documented as unloadable by the reference's own present code.  GPU: the models run through evaluate.run_test_environments (K11 / K14 /
HIP env) on the reference's own test sets; the outcome is held to the ordering and magnitudes measured in
profiles/r03_authors_checkpoints.json and discussed in DESIGN.md (section 7, row f3)."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
TAGS = ("og", "glatt", "rhine0", "rhine1")


def _state_dict(tag):
    z = np.load(os.path.join(G, f"ckpt_rada2c_{tag}.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("tag", TAGS)
def test_authors_rada2c_checkpoints_load_strictly(tag):
    from radiation_ppo_amd.rada2c import RNNModelActorCritic
    ac = RNNModelActorCritic()
    sd = _state_dict(tag)
    assert set(sd) == set(ac.state_dict()) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in ac.state_dict().items())
    ac.load_state_dict(sd, strict=True)
    assert all(torch.isfinite(p).all() for p in ac.parameters())
    # a trained model: not the initialisation's scale everywhere (the GRU's weights have moved well beyond U(-1/sqrt(24), .))
    assert float(ac.pi.logits_net.v_net.seq_model.weight_hh_l0.abs().max()) > 0.25


def test_authors_radteam_checkpoints_predate_the_present_architecture():
    """Their actor.pt / critic.pt hold a 5-channel first convolution for BOTH networks plus duplicate step<k> keys, and predictor.pt
    a 64-unit PFGRU with a BatchNorm: the reference's present CNNBase (6 actor / 4 critic channels, 24-unit PFGRU without BatchNorm,
    algos/test_cnn/RADTEAM_core.py:962-1023,1211-1271,1533-1585; the shapes pinned by tests/golden/cnn.npz / pfgru.npz) cannot load them
    either, so there is nothing to evaluate for RAD-TEAM."""
    shapes = json.load(open(os.path.join(G, "ckpt_radteam_shapes.json")))
    cnn = np.load(os.path.join(G, "cnn.npz"))
    assert len(shapes) >= 13
    for name, tab in shapes.items():
        if name.endswith("actor.pt"):
            assert tab["actor.0.weight"] == [8, 5, 3, 3] and cnn["a_actor.0.weight"].shape == (8, 6, 3, 3) and "step1.weight" in tab
        elif name.endswith("critic.pt"):
            assert tab["critic.0.weight"] == [8, 5, 3, 3] and cnn["c_critic.0.weight"].shape == (8, 4, 3, 3)
        else:
            assert tab["fc_z.weight"] == [64, 67] and "batch_norm.weight" in tab


def test_saved_test_set_fixtures_have_the_references_structure():
    from radiation_ppo_amd.testsets import load_test_environments_npz, summarize_test_set
    for k, snr in ((0, "high"), (0, "low"), (3, "high"), (3, "low")):
        sets = load_test_environments_npz(os.path.join(G, f"testset_obs{k}_{snr}.npz"))
        s = summarize_test_set(sets)
        assert s["count"] == 100 and s["obstructions"] == (k, k) and s["min_start_distance"] >= 1000.0
        e = sets["env_7"]
        assert len(e) == (5 if k else 4) and e[0].dtype == np.float64 and np.all(e[0] == np.round(e[0]))
        if k:
            assert len(e[4]) == k and e[4][0][0].shape == (4, 2)


@pytest.mark.gpu
def test_authors_models_evaluated_on_the_references_test_sets():
    """100 saved environments x 20 Monte-Carlo runs per model and set.  Measured with 100 runs (profiles/r03_authors_checkpoints.json):
    og 0.68 / 0.55 (no / 3 obstructions), glatt 0.13 / 0.21, rhine1 0.17 / 0.11, rhine0 0.014 / 0.018 -- the ordering of the authors'
    own training logs (og 0.90, glatt 0.37, rhine0 0.015 of the episodes end on the source)."""
    from radiation_ppo_amd.evaluate import run_test_environments
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    from radiation_ppo_amd.testsets import load_test_environments_npz
    log = json.load(open(os.path.join(G, "ckpt_rada2c_authors_log.json")))
    rate = {}
    for tag in TAGS:
        ag = RNNAgentPPO(id=0, device="cuda:0")
        ag.agent.load_state_dict(_state_dict(tag), strict=True)
        for k in (0, 3):
            sets = load_test_environments_npz(os.path.join(G, f"testset_obs{k}_high.npz"))
            res, s = run_test_environments(ag, sets, montecarlo_runs=20, steps_per_episode=120, obstruction_count=k, seed=2)
            assert len(res) == 100 and s["completed_runs"] == 2000
            rate[tag, k] = (s["success_rate"], s["successful_episode_length_median"])
    print({f"{t}/obs{k}": v for (t, k), v in rate.items()})
    for k in (0, 3):
        assert rate["og", k][0] > 0.45 and rate["og", k][0] > 2.0 * rate["glatt", k][0] > 4.0 * rate["rhine0", k][0]
        assert rate["rhine0", k][0] < 0.06 and 0.05 < rate["glatt", k][0] < 0.40 and 0.04 < rate["rhine1", k][0] < 0.40
        assert 30 <= rate["og", k][1] <= 90
    # rhine0 -- the one model whose training never took off -- agrees with its own log to the percent
    assert abs(rate["rhine0", 0][0] - log["rhine0"]["approx_success_rate"]) < 0.03
