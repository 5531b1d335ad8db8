"""Pins oracle/train_loop_oracle.py to the reference's own train() and update_rada2c() (SURVEY 8c fixtures 5 and 7).

train_trace.json : every call train_PPO.train (algos/multiagent/train.py:259-627) made on its env / agents / loggers
                   in a run of the reference itself (recording stand-ins for the agents), plus every numpy draw
                   the reference env consumed.  The oracle env replays the draws, the oracle loop must emit the same events.
rada2c_loss.npz  : loss, statistics, gradients, post-Adam parameters of the reference's update_rada2c.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle.radsearch_oracle import RadSearchOracle, ReplayDraws
from oracle.train_loop_oracle import rada2c_loss, train_loop_trace


def _rows(draws):
    rows = []
    for kind, a0, a1, vals in draws:
        for v in vals:
            rows.append((0 if kind == "integers" else 1, a0, a1, v))
    return rows


@pytest.mark.parametrize("name", ["a1_individual", "a2_team", "a1_mlp_standardized"])
def test_train_control_flow_matches_reference_trace(golden_dir, name):
    with open(os.path.join(golden_dir, "train_trace.json")) as f:
        g = json.load(f)[name]
    want = g["events"]
    script = [e for e in want if e[0] == "agent_step"]
    pos = [0]

    def agent_step(i, observations):
        e = script[pos[0]]
        pos[0] += 1
        assert e[1] == i
        for k, v in observations.items():          # the oracle env reproduces the reference's observations exactly
            assert np.array_equal(np.asarray(v, dtype=np.float64), np.asarray(e[2][str(k)])), (pos[0], k)
        return e[3], e[4], -0.5 - e[4]

    env = RadSearchOracle(ReplayDraws(_rows(g["draws"])), number_agents=g["A"], obstruction_count=0,
                          enforce_grid_boundaries=True)
    got, episodes = train_loop_trace(env, agent_step, g["A"], g["global_critic"], g["T"], g["L"], g["epochs"], arch=g["arch"])
    assert episodes == g["episode_count"]
    assert len(got) == len(want)
    for k, (a, b) in enumerate(zip(got, want)):
        assert a == b, (k, a, b)
    assert env.rng.pos == len(env.rng.rows)          # every recorded draw was consumed, none invented
    # what the fixture exercises: terminal (last_val 0), timeout and epoch-cut (bootstrapped) trajectories
    gae = [e for e in want if e[0] == "gae"]
    assert any(e[2] == 0.0 for e in gae) and any(e[2] != 0.0 for e in gae)
    assert sum(1 for e in want if e[0] == "epoch_end_set") == g["epochs"]


def _load_ac(d, prefix):
    from collections import OrderedDict
    lin = lambda i, o: torch.nn.Linear(i, o)
    actor = torch.nn.Sequential(lin(11, 64), torch.nn.Tanh(), lin(64, 64), torch.nn.Tanh(), lin(64, 8))
    critic = torch.nn.Sequential(lin(11, 64), torch.nn.Tanh(), lin(64, 64), torch.nn.Tanh(), lin(64, 1))
    actor.load_state_dict(OrderedDict((k[len("actor."):], torch.from_numpy(d[prefix + k])) for k in
                                      ("actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias", "actor.4.weight", "actor.4.bias")))
    critic.load_state_dict(OrderedDict((k[len("critic."):], torch.from_numpy(d[prefix + k])) for k in
                                       ("critic.0.weight", "critic.0.bias", "critic.2.weight", "critic.2.bias", "critic.4.weight", "critic.4.bias")))
    return actor, critic


def test_rada2c_loss_gradients_and_adam_step_match_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "rada2c_loss.npz"))
    for tag in ("step", "stop"):
        actor, critic = _load_ac(d, f"{tag}_before_")
        eps = [torch.from_numpy(d[f"{tag}_ep{i}"]) for i in d[f"{tag}_order"]]     # the order :1183 sampled
        loss, st = rada2c_loss(actor, critic, eps, clip_ratio=0.2, alpha=0.1)
        assert abs(loss.item() - float(d[f"{tag}_loss"])) < 1e-6
        for k in ("kl", "ent", "cf", "val_loss"):
            assert abs(st[k] - float(d[f"{tag}_{k}"])) < 1e-6, (tag, k)
        term = not (st["kl"] < 1.5 * 0.07)                                         # :1250
        assert term == bool(d[f"{tag}_term"])
        params = list(actor.parameters()) + list(critic.parameters())
        opt = torch.optim.Adam(params, lr=3e-4)
        if not term:
            loss.backward()
            for mod, pre in ((actor, "actor."), (critic, "critic.")):
                for k, p in mod.named_parameters():
                    assert np.allclose(p.grad.numpy(), d[f"{tag}_grad_{pre}{k}"], rtol=1e-5, atol=1e-8), (tag, pre + k)
            opt.step()
        for mod, pre in ((actor, "actor."), (critic, "critic.")):
            for k, p in mod.named_parameters():
                got, want = p.detach().numpy(), d[f"{tag}_after_{pre}{k}"]
                # the first Adam step moves every weight by lr * g / (|g| + 1e-8): exact wherever the gradient is not
                # itself at the 1e-8 noise level, bounded by 2 lr elsewhere
                big = np.abs(d[f"{tag}_grad_{pre}{k}"]) > 1e-6 if not term else np.ones_like(want, dtype=bool)
                assert np.allclose(got[big], want[big], rtol=0, atol=2e-7), (tag, pre + k)
                assert np.abs(got - want).max() <= 6.1e-4, (tag, pre + k)
    assert bool(d["step_term"]) is False and bool(d["stop_term"]) is True


@pytest.mark.parametrize("name", ["a1_individual", "a2_team", "a1_mlp_standardized"])
def test_epoch_logger_reproduces_the_reference_progress_file(golden_dir, tmp_path, name):
    """Row T / f4: every store / log_tabular / dump_tabular call the reference's train() made on its loggers
    (train.py:429, :496-526, :577-627) is replayed on the build's EpochLogger; the progress.txt it writes must equal,
    byte for byte, the file the reference's own EpochLogger (epoch_logger.py:110-403) wrote in the same run -- header
    (MeanVVals StdVVals MaxVVals MinVVals ... MeanEpRet StdEpRet MaxEpRet MinEpRet DoneCount EpLen Time) and values;
    only the wall-clock Time column is exempt."""
    from radiation_ppo_amd.logger import EpochLogger
    from radiation_ppo_amd.train import COLUMNS
    with open(os.path.join(golden_dir, "train_trace.json")) as f:
        g = json.load(f)[name]
    for aid, calls in g["logger_calls"].items():
        lg = EpochLogger(output_dir=str(tmp_path / f"{name}_{aid}"))
        for c in calls:
            if c[0] == "store":
                lg.store(**c[1])
            elif c[0] == "log_tabular":
                lg.log_tabular(c[1], c[2], **c[3])
            else:
                lg.dump_tabular()
        lg.output_file.flush()
        got = open(os.path.join(lg.output_dir, "progress.txt")).read().splitlines()
        want = g["progress"][aid].splitlines()
        assert got[0] == want[0]
        head = want[0].split("\t")
        assert head == COLUMNS[:len(head)] and COLUMNS[len(head):] == ["EnvStepsPerSec", "PPOItersPerSec"]
        t_col = head.index("Time")
        assert len(got) == len(want) == g["epochs"] + 1
        for a, b in zip(got[1:], want[1:]):
            a, b = a.split("\t"), b.split("\t")
            assert a[:t_col] == b[:t_col] and a[t_col + 1:] == b[t_col + 1:], (a, b)
