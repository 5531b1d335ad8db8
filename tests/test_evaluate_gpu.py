"""radiation_ppo_amd.evaluate.run_test_environments (the batched EpisodeRunner.run, algos/multiagent/evaluate.py:333-476):
every (saved environment, Monte-Carlo run) episode is replayed through the oracle -- refresh_environment, then the logged
actions -- and must end at the same step with the same success flag and return; the per-environment records add up."""
import os

import numpy as np
import pytest
import torch

from oracle.radsearch_oracle import PhiloxDraws, RadSearchOracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obst,arch", [(0, "ff"), (3, "ff"), (2, "rnn")])
def test_monte_carlo_evaluation_matches_oracle_replay(obst, arch):
    from radiation_ppo_amd.evaluate import run_test_environments, sample_test_environments
    from radiation_ppo_amd.ppo import VecAgentPPO
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    torch.manual_seed(9)
    E, R, L, seed = 6, 5, 40, 123
    sets = sample_test_environments(E, obstruction_count=obst, seed=77)
    assert sorted(sets) == [f"env_{i}" for i in range(E)] and len(sets["env_0"]) == (5 if obst else 4)
    agent = VecAgentPPO(id=0, steps_per_epoch=480, steps_per_episode=L) if arch == "ff" else RNNAgentPPO(id=0, steps_per_episode=L)
    with torch.no_grad():                                           # a decisive policy finds sources within 40 steps sometimes
        for p in (agent.agent.actor if arch == "ff" else agent.agent.pi).parameters():
            p.mul_(4.0)
    results, summary, actions = run_test_environments(agent, sets, montecarlo_runs=R, steps_per_episode=L, obstruction_count=obst,
                                                      seed=seed, return_actions=True)
    assert len(results) == E and summary["completed_runs"] == E * R
    n_success = 0
    for e, res in enumerate(results):
        assert res.id == e and res.completed_runs == R and len(res.total_episode_length) == R
        assert len(res.successful.episode_length) == res.success_counter
        assert len(res.successful.episode_length) + len(res.unsuccessful.episode_length) == R
        s = sets[f"env_{e}"]
        rects = None
        if obst:
            rects = [(int(o[0][:, 0].min()), int(o[0][:, 1].min()), int(o[0][:, 0].max()), int(o[0][:, 1].max())) for o in s[4]]
        for r in range(R):
            n = e * R + r
            ref = RadSearchOracle(PhiloxDraws(seed, n), number_agents=1, obstruction_count=obst, enforce_grid_boundaries=True)
            ref.refresh_environment(s[0], s[1], s[2], s[3], rects)
            ret, steps, found = np.float32(0.0), 0, False
            for t in range(actions.shape[0]):
                o, rew, done, _ = ref.step({0: int(actions[t, n])})
                ret = np.float32(ret + np.float32(rew["individual_reward"][0]))
                steps += 1
                if done[0]:
                    found = True
                    break
            assert res.total_episode_length[r] == steps, (e, r)
            bucket = res.successful if found else res.unsuccessful
            n_success += int(found)
            assert steps in bucket.episode_length
            assert any(abs(v - float(ret)) < 1e-4 for v in bucket.episode_return), (e, r, float(ret), bucket.episode_return)
    assert sum(r.success_counter for r in results) == n_success
    assert abs(summary["success_rate"] - n_success / (E * R)) < 1e-9


@pytest.mark.parametrize("team_mode", ["individual", "team"])
def test_cnn_monte_carlo_evaluation_matches_oracle_replay(team_mode):
    """run_test_environments_cnn: RAD-TEAM policies (2 agents, CNN actors fed by the heat maps + PFGRU channel) evaluated on saved
    environments with obstacles; each (environment, run) episode replayed through the oracle with the logged joint actions ends at
    the same step with the same success flag and the same accumulated return (agent 0's own reward, or the team reward)."""
    from radiation_ppo_amd.evaluate import run_test_environments_cnn, sample_test_environments, summarize
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO
    torch.manual_seed(4)
    E, R, L, A, seed, obst = 4, 3, 30, 2, 321, 2
    sets = sample_test_environments(E, obstruction_count=obst, seed=55)
    agents = {i: CNNAgentPPO(id=i) for i in range(A)}
    results, summary, actions = run_test_environments_cnn(agents, sets, montecarlo_runs=R, steps_per_episode=L, team_mode=team_mode,
                                                          obstruction_count=obst, seed=seed, return_actions=True)
    assert len(results) == E and summary["completed_runs"] == E * R and len(summary["scenarios"]) == E
    for e, res in enumerate(results):
        s = sets[f"env_{e}"]
        rects = [(int(o[0][:, 0].min()), int(o[0][:, 1].min()), int(o[0][:, 0].max()), int(o[0][:, 1].max())) for o in s[4]]
        for r in range(R):
            n = e * R + r
            ref = RadSearchOracle(PhiloxDraws(seed, n), number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True)
            ref.refresh_environment(s[0], s[1], s[2], s[3], rects)
            ret, steps, found = np.float32(0.0), 0, False
            for t in range(actions.shape[0]):
                o, rew, done, _ = ref.step({i: int(actions[t, n, i]) for i in range(A)})
                rr = rew["individual_reward"][0] if team_mode == "individual" else rew["team_reward"]
                ret = np.float32(ret + np.float32(rr))
                steps += 1
                if any(done.values()):
                    found = True
                    break
            assert res.total_episode_length[r] == steps, (e, r)
            bucket = res.successful if found else res.unsuccessful
            assert any(abs(v - float(ret)) < 1e-4 for v in bucket.episode_return), (e, r, float(ret), bucket.episode_return)
    again = summarize(results)
    assert again["success_rate"] == summary["success_rate"]
    q = summary["success_count_weighted_quantiles"]
    assert q["0.025"] <= q["0.5"] <= q["0.975"]


def test_evaluate_ppo_driver_reads_a_saved_set_and_saved_models(tmp_path):
    """evaluate_PPO (evaluate.py:581-643) end to end: a set in the reference's joblib layout on disk (read by the safe reader), models
    saved by train_PPO under `<id>_agent`, the reference's eval_kwargs."""
    joblib = pytest.importorskip("joblib")
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.evaluate import evaluate_PPO, sample_test_environments
    from radiation_ppo_amd.train import train_PPO
    sets = sample_test_environments(5, obstruction_count=1, seed=3)
    os.makedirs(tmp_path / "sets")
    joblib.dump(sets, str(tmp_path / "sets" / "test_env_dict_obs1_low_v4"))
    env = RadSearchVec(8, number_agents=2, obstruction_count=1, enforce_grid_boundaries=True, seed=2)
    sim = train_PPO(env=env, logger_kwargs=dict(output_dir=str(tmp_path / "models")), ppo_kwargs=dict(train_pi_iters=1, train_v_iters=1),
                    seed=2, number_of_agents=2, steps_per_epoch=12, steps_per_episode=6, total_epochs=1)
    sim.train()
    ev = evaluate_PPO(dict(test_env_path=str(tmp_path / "sets"), obstruction_count=1, snr="low", episodes=4, montecarlo_runs=3,
                           model_path=str(tmp_path / "models"), actor_critic_architecture="cnn", number_of_agents=2,
                           steps_per_episode=10, enforce_boundaries=True, team_mode="team", seed=1))
    results, summary = ev.evaluate()
    assert len(results) == 4 and summary["completed_runs"] == 12 and 0.0 <= summary["success_rate"] <= 1.0
    with pytest.raises(ValueError):
        evaluate_PPO(dict(test_env_path="x", obstruction_count=-1))


def test_evaluate_ppo_driver_with_the_recurrent_agent(tmp_path):
    """evaluate_PPO with actor_critic_architecture='rnn': pyt_save/model.pt written by train_PPO, episodes with carried hidden states."""
    joblib = pytest.importorskip("joblib")
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.evaluate import evaluate_PPO, sample_test_environments
    from radiation_ppo_amd.train import train_PPO
    sets = sample_test_environments(4, obstruction_count=0, seed=3)
    os.makedirs(tmp_path / "sets")
    joblib.dump(sets, str(tmp_path / "sets" / "test_env_dict_obs0_high_v4"))
    env = RadSearchVec(16, number_agents=1, obstruction_count=0, enforce_grid_boundaries=True, seed=2)
    train_PPO(env=env, logger_kwargs=dict(output_dir=str(tmp_path / "models")), ppo_kwargs=dict(train_pi_iters=1, train_pfgru_iters=1),
              seed=2, number_of_agents=1, actor_critic_architecture="rnn", global_critic_flag=False, steps_per_epoch=12,
              steps_per_episode=6, total_epochs=1).train()
    ev = evaluate_PPO(dict(test_env_path=str(tmp_path / "sets"), obstruction_count=0, snr="high", episodes=3, montecarlo_runs=4,
                           model_path=str(tmp_path / "models"), actor_critic_architecture="rnn", number_of_agents=1,
                           steps_per_episode=10, enforce_boundaries=True, seed=1))
    results, summary = ev.evaluate()
    assert len(results) == 3 and summary["completed_runs"] == 12 and 0.0 <= summary["success_rate"] <= 1.0


def test_sequential_runs_carry_the_hidden_state_like_the_reference(monkeypatch):
    """carry_hidden_across_runs=True (evaluate.py:357, :455-470): the runs of one saved environment follow each other on one lane;
    `hiddens` is created once, the statistics buffer restarts with every run.  (i) With one run per environment the sequential form IS
    the lane-per-run form (same lanes, same Philox streams): identical records.  (ii) With several runs every lane replays through the
    oracle -- refresh_environment, the logged actions, refresh again ... -- to the same lengths / success flags / returns in run order,
    while the GRU's h0 and the particle sets were drawn exactly once and the Welford state restarted once per run."""
    from radiation_ppo_amd import evaluate as ev
    from radiation_ppo_amd.pfgru import PredictorBank
    from radiation_ppo_amd.ppo import DeviceWelford
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    torch.manual_seed(9)
    E, L, seed, obst = 6, 30, 123, 2
    sets = ev.sample_test_environments(E, obstruction_count=obst, seed=77)
    agent = RNNAgentPPO(id=0, steps_per_episode=L)
    with torch.no_grad():
        for p in agent.agent.pi.parameters():
            p.mul_(4.0)
    a1, s1 = ev.run_test_environments(agent, sets, montecarlo_runs=1, steps_per_episode=L, obstruction_count=obst, seed=seed)
    b1, t1 = ev.run_test_environments(agent, sets, montecarlo_runs=1, steps_per_episode=L, obstruction_count=obst, seed=seed,
                                      carry_hidden_across_runs=True)
    assert [r.total_episode_length for r in a1] == [r.total_episode_length for r in b1]
    assert [r.success_counter for r in a1] == [r.success_counter for r in b1] and s1["success_rate"] == t1["success_rate"]
    assert [r.successful.episode_return + r.unsuccessful.episode_return for r in a1] == [r.successful.episode_return + r.unsuccessful.episode_return for r in b1]

    calls = {"h0": 0, "bank_reset": 0, "stat_reset": 0}
    h0, br, sr = agent.agent.gru_h0, PredictorBank.reset, DeviceWelford.reset
    monkeypatch.setattr(agent.agent, "gru_h0", lambda u: (calls.__setitem__("h0", calls["h0"] + 1), h0(u))[1])
    monkeypatch.setattr(PredictorBank, "reset", lambda self, mask=None: (calls.__setitem__("bank_reset", calls["bank_reset"] + 1), br(self, mask))[1])
    monkeypatch.setattr(DeviceWelford, "reset", lambda self, mask: (calls.__setitem__("stat_reset", calls["stat_reset"] + 1), sr(self, mask))[1])
    R = 4
    results, summary, actions = ev.run_test_environments(agent, sets, montecarlo_runs=R, steps_per_episode=L, obstruction_count=obst, seed=seed,
                                                         carry_hidden_across_runs=True, return_actions=True)
    assert calls["h0"] == 1 and calls["bank_reset"] == 1 and calls["stat_reset"] == actions.shape[0]
    assert summary["completed_runs"] == E * R
    for e, res in enumerate(results):
        s = sets[f"env_{e}"]
        rects = [(int(o[0][:, 0].min()), int(o[0][:, 1].min()), int(o[0][:, 0].max()), int(o[0][:, 1].max())) for o in s[4]]
        ref = RadSearchOracle(PhiloxDraws(seed, e), number_agents=1, obstruction_count=obst, enforce_grid_boundaries=True)
        ref.refresh_environment(s[0], s[1], s[2], s[3], rects)
        lens, rets, sucs = [], [], []
        ret, steps = np.float32(0.0), 0
        for t in range(actions.shape[0]):
            if actions[t, e] < 0:
                break
            o, rew, done, _ = ref.step({0: int(actions[t, e])})
            ret = np.float32(ret + np.float32(rew["individual_reward"][0]))
            steps += 1
            if done[0] or steps == L:
                lens.append(steps); rets.append(float(ret)); sucs.append(bool(done[0]))
                ret, steps = np.float32(0.0), 0
                ref.refresh_environment(s[0], s[1], s[2], s[3], rects)
        assert len(lens) == R and res.total_episode_length == lens, (e, lens, res.total_episode_length)
        assert res.success_counter == sum(sucs)
        assert res.successful.episode_length == [l for l, k in zip(lens, sucs) if k]
        assert np.allclose(res.successful.episode_return, [r for r, k in zip(rets, sucs) if k], atol=1e-4)
        assert np.allclose(res.unsuccessful.episode_return, [r for r, k in zip(rets, sucs) if not k], atol=1e-4)
