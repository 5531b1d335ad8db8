"""Rows T / f4 on the GPU: train_PPO writes the reference's progress.txt columns and checkpoint files, and a run can be
resumed from them.  Column names and formatting are pinned on the CPU (tests/test_train_loop_golden.py replays the
reference's own logger calls); here the trainer itself is driven."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _read(path):
    lines = open(path).read().splitlines()
    head = lines[0].split("\t")
    return head, [dict(zip(head, l.split("\t"))) for l in lines[1:]]


def test_ff_progress_columns_model_file_and_resume(tmp_path, golden_dir):
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.train import COLUMNS, train_PPO
    kw = dict(ppo_kwargs=dict(observation_space=11, alpha=0.1, train_pi_iters=4), seed=3, number_of_agents=1,
              actor_critic_architecture="ff", global_critic_flag=False, steps_per_epoch=24, steps_per_episode=8, save_freq=2)
    lk = dict(exp_name="unit", seed=3, data_dir=str(tmp_path), env_name="radsearch")      # the reference's logger_kwargs form
    env = RadSearchVec(32, obstruction_count=0, enforce_grid_boundaries=True, seed=3)
    sim = train_PPO(env=env, logger_kwargs=lk, total_epochs=3, **kw)
    sim.train()
    adir = tmp_path / "radsearch" / "0_agent_unit_s3"                                     # epoch_logger.py:103-105
    assert (tmp_path / "radsearch" / "general_s3" / "config.json").exists()               # train.py:170-177
    head, rows = _read(adir / "progress.txt")
    assert head == COLUMNS
    assert head[:23] == ("AgentID Epoch MeanVVals StdVVals MaxVVals MinVVals TotalEnvInteracts loss_policy loss_critic loss_predictor "
                         "LocLoss Entropy kl_divergence ClipFrac OutOfBound stop_iteration MeanEpRet StdEpRet MaxEpRet MinEpRet "
                         "DoneCount EpLen Time").split()                                  # saves/2023-04-17-15:38:48/0_agent_*/progress.txt:1
    assert len(rows) == 3 and [int(r["Epoch"]) for r in rows] == [0, 1, 2]
    assert int(rows[2]["TotalEnvInteracts"]) == 3 * 24 * 32
    for r in rows:
        assert float(r["MinVVals"]) <= float(r["MeanVVals"]) <= float(r["MaxVVals"]) and float(r["StdVVals"]) >= 0
        assert float(r["MinEpRet"]) <= float(r["MeanEpRet"]) <= float(r["MaxEpRet"]) and float(r["StdEpRet"]) >= 0
        assert 1 <= float(r["EpLen"]) <= 8
    # cross-check the device-side reductions against the buffer of the last epoch
    v = sim.collector.buf.val.double().cpu().numpy().reshape(-1)
    assert abs(float(rows[2]["MeanVVals"]) - v.mean()) < 1e-6 and abs(float(rows[2]["StdVVals"]) - v.std()) < 1e-6
    assert abs(float(rows[2]["MaxVVals"]) - v.max()) < 1e-6 and abs(float(rows[2]["MinVVals"]) - v.min()) < 1e-6
    # the model file is a state_dict with FF_core.ActorCritic's keys (ff_core.npz holds the reference's)
    sd = torch.load(adir / "pyt_save" / "model.pt", map_location="cpu")
    want = sorted(k[3:] for k in np.load(os.path.join(golden_dir, "ff_core.npz")).files if k.startswith("sd_"))
    assert sorted(sd.keys()) == want
    # resume: a fresh trainer restores weights, Adam moments, the step count and the epoch counter, then trains on
    st = torch.load(adir / "resume.pt", map_location="cpu")
    assert st["epochs_done"] == 3 and st["agent"]["fused"]["adam_step"] >= 1
    env2 = RadSearchVec(32, obstruction_count=0, enforce_grid_boundaries=True, seed=3)
    sim2 = train_PPO(env=env2, logger_kwargs=dict(output_dir=str(tmp_path / "resumed")), total_epochs=5, **kw)
    sim2.load(str(tmp_path / "radsearch"))
    assert sim2.epochs_done == 3 and sim2.agents[0].epochs_done == sim.agents[0].epochs_done
    f1, f2 = sim.agents[0]._fused, sim2.agents[0]._fused
    assert torch.equal(f1.m, f2.m) and torch.equal(f1.v, f2.v) and int(f1.state_i32[0]) == int(f2.state_i32[0])
    for a, b in zip(sim.agents[0].agent.parameters(), sim2.agents[0].agent.parameters()):
        assert torch.equal(a, b)
    sim2.train()
    _, rows2 = _read(tmp_path / "resumed" / "0_agent" / "progress.txt")
    assert [int(r["Epoch"]) for r in rows2] == [3, 4] and np.isfinite(float(rows2[-1]["loss_policy"]))


def test_cnn_checkpoint_files_match_reference_names(tmp_path, golden_dir):
    """CNNBase.save (RADTEAM_core.py:1904-1943): actor.pt / critic.pt state_dicts with the reference's keys (cnn.npz holds them);
    per-agent OutOfBound / DoneCount columns; resume restores both optimisers."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.train import COLUMNS, train_PPO
    A = 2
    kw = dict(ppo_kwargs=dict(train_pi_iters=2, train_v_iters=2), seed=5, number_of_agents=A, steps_per_epoch=16,
              steps_per_episode=8, save_freq=1)                                            # architecture / global critic: reference defaults
    env = RadSearchVec(8, number_agents=A, obstruction_count=1, enforce_grid_boundaries=True, seed=5)
    sim = train_PPO(env=env, logger_kwargs=dict(output_dir=str(tmp_path)), total_epochs=2, **kw)
    assert sim.actor_critic_architecture == "cnn" and sim.global_critic_flag is True      # train.py:119-121
    sim.train()
    g = np.load(os.path.join(golden_dir, "cnn.npz"))
    for i in range(A):
        d = tmp_path / f"{i}_agent"
        head, rows = _read(d / "progress.txt")
        assert head == COLUMNS and len(rows) == 2
        sa = torch.load(d / "actor.pt", map_location="cpu")
        sc = torch.load(d / "critic.pt", map_location="cpu")
        assert sorted(sa.keys()) == sorted(k[2:] for k in g.files if k.startswith("a_"))
        assert sorted(sc.keys()) == sorted(k[2:] for k in g.files if k.startswith("c_"))
        for k in sa:
            assert tuple(sa[k].shape) == g["a_" + k].shape
    env2 = RadSearchVec(8, number_agents=A, obstruction_count=1, enforce_grid_boundaries=True, seed=5)
    sim2 = train_PPO(env=env2, logger_kwargs={}, total_epochs=3, **kw)
    sim2.load(str(tmp_path))
    assert sim2.epochs_done == 2
    for i in range(A):
        for a, b in zip(sim.agents[i].pi.parameters(), sim2.agents[i].pi.parameters()):
            assert torch.equal(a, b)
        s1, s2 = sim.agents[i].pi_optimizer.state_dict()["state"], sim2.agents[i].pi_optimizer.state_dict()["state"]
        assert all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) for k in s1)
    # a CNN checkpoint written by the reference's naming loads through CNNAgentPPO.load (train() saves BEFORE an epoch's update,
    # train.py:552-569, so the files are rewritten from the final weights first)
    sim.save()
    assert (tmp_path / "1_agent" / "predictor.pt").exists()          # PFGRUCell.save_model (RADTEAM_core.py:1654-1655)
    sim2.agents[0].load(str(tmp_path / "1_agent"))
    for a, b in zip(sim2.agents[0].pi.parameters(), sim.agents[1].pi.parameters()):
        assert torch.equal(a, b)
    sim2.train()
    assert len(sim2.loggers[0].rows) == 1 and sim2.loggers[0].rows[0]["Epoch"] == 2


def test_cnn_trains_without_enforced_boundaries():
    """enforce_grid_boundaries=False grows the heat maps to 147 x 147 (RADTEAM_core.py:1727-1738).  The trunk kernels K9 / K10 hold one
    27 x 27 image in LDS; this size takes the dense stack through the library convolutions with the same modules, K5 and the env
    kernels unchanged: train_PPO sizes the networks from the env (Linear(16 * 73 * 73, 32) for 120-step episodes), collects and updates."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.train import train_PPO
    env = RadSearchVec(4, number_agents=2, obstruction_count=1, enforce_grid_boundaries=False, seed=1)
    sim = train_PPO(env=env, logger_kwargs={}, number_of_agents=2, global_critic_flag=True, steps_per_epoch=8, steps_per_episode=4,
                    total_epochs=2, ppo_kwargs=dict(train_pi_iters=2, train_v_iters=2))
    # the offset covers steps_per_episode detector steps beyond the area: 31 x 31 cells for 4-step episodes (147 x 147 for 120)
    assert sim.collector.maps.map_dimensions == (31, 31) and sim.agents[0].pi.actor[6].in_features == 16 * 15 * 15
    before = torch.cat([p.detach().reshape(-1).clone() for p in sim.agents[1].pi.parameters()])
    sim.train()
    after = torch.cat([p.detach().reshape(-1) for p in sim.agents[1].pi.parameters()])
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    rows = sim.loggers[0].rows
    assert len(rows) == 2 and all(np.isfinite(float(r[k])) for r in rows for k in ("loss_policy", "loss_critic", "kl_divergence", "Entropy"))
    # the dense stack the library path convolves == what the K5 stack kernel materialises for the same state
    col = sim.collector
    shared, cells, pcells = col.maps.shared_maps(), col.maps.field("cell").long(), col.maps.field("pred_cell").long()
    actor, _ = col.maps.stacks()
    for a in range(2):
        assert torch.equal(col.actor_stack_from(shared, cells, pcells, a), actor[:, a])


@pytest.mark.parametrize("arch", ["ff", "rnn", "cnn"])
def test_resumed_run_equals_the_uninterrupted_run(tmp_path, arch):
    """3 epochs in one go against 2 epochs, save_resume, a fresh process-worth of objects (new env, agents, collector), load, 1 more
    epoch: resume.pt restores the env workspace (Philox counters, sources, rectangles), the collector's running episode state
    (Welford / heat maps / particle sets / GRU states / episode and epoch counters that key the update draws), optimiser moments
    and the host generator, so the third epoch's rollout buffer and the final parameters are IDENTICAL."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.train import train_PPO
    A = 2 if arch == "cnn" else 1
    kw = dict(seed=7, number_of_agents=A, actor_critic_architecture=arch, global_critic_flag=(arch == "cnn"), steps_per_epoch=20,
              steps_per_episode=8, save_freq=1,
              ppo_kwargs=dict(train_pi_iters=2, train_v_iters=2, train_pfgru_iters=2, alpha=0.1))

    def make(out, epochs):
        env = RadSearchVec(16, number_agents=A, obstruction_count=2, enforce_grid_boundaries=True, seed=11)
        return train_PPO(env=env, logger_kwargs=dict(output_dir=str(out)), total_epochs=epochs, **kw)

    def params(sim):
        mods = []
        for ag in sim.agents.values():
            mods += [ag.pi, ag.critic, ag.model] if arch == "cnn" else [ag.agent]
        return torch.cat([p.detach().reshape(-1) for m in mods for p in m.parameters()])

    whole = make(tmp_path / "whole", 3)
    whole.train()
    first = make(tmp_path / "first", 2)
    first.train()
    second = make(tmp_path / "second", 3)
    second.load(str(tmp_path / "first"))
    assert second.epochs_done == 2
    second.train()
    for k in ("obs", "act", "rew", "val", "logp", "cut", "adv", "ret"):
        assert torch.equal(getattr(whole.collector.buf, k), getattr(second.collector.buf, k)), k
    assert torch.equal(params(whole), params(second))
    assert whole.loggers[0].rows[-1]["loss_policy"] == second.loggers[0].rows[-1]["loss_policy"]
