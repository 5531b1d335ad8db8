#!/usr/bin/env python3
"""Row f3's purpose: the AUTHORS' saved RAD-A2C models (tests/golden/ckpt_rada2c_*.npz, converted from their pyt_save/model.pt by
tests/golden/make_checkpoints.py) evaluated by this build's harness (radiation_ppo_amd.evaluate.run_test_environments on K11 / K14 and
the HIP env) on the first 100 environments of the reference's own saved test sets, 100 Monte-Carlo runs each (evaluate.py's
defaults), next to what the authors' progress.txt reports for the same models.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


def main(runs=100):
    from radiation_ppo_amd.evaluate import run_test_environments
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    from radiation_ppo_amd.testsets import load_test_environments_npz
    out = {"authors_log": json.load(open(os.path.join(G, "ckpt_rada2c_authors_log.json"))), "evaluation": {}}
    for tag in ("og", "glatt", "rhine0", "rhine1"):
        ag = RNNAgentPPO(id=0, device="cuda:0")
        z = np.load(os.path.join(G, f"ckpt_rada2c_{tag}.npz"))
        ag.agent.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
        for k, snr in ((0, "high"), (0, "low"), (3, "high"), (3, "low")):
            sets = load_test_environments_npz(os.path.join(G, f"testset_obs{k}_{snr}.npz"))
            # `og` was trained by the ORIGINAL RAD-A2C code, whose env generator measures I / r^2 (algos/test_environment/eval/test_env_gen.py:37-38);
            # the multi-agent env this build mirrors measures I / r as written (rad_search_env.py:501): both are evaluated
            # carry: the reference's EpisodeRunner.run keeps `hiddens` across the Monte-Carlo runs of an environment (evaluate.py:357,
            # :455-470; carry_hidden_across_runs=True reproduces it: runs in sequence on one lane); fresh: every run starts from a new
            # hidden state (all runs as parallel lanes -- what round 3 measured)
            variants = [("reference", True), ("reference", False)] + ([("inverse_square", True)] if tag == "og" else [])
            for falloff, carry in variants:
                t0 = time.time()
                _, s = run_test_environments(ag, sets, montecarlo_runs=runs, steps_per_episode=120, obstruction_count=k, seed=2, falloff=falloff,
                                             carry_hidden_across_runs=carry)
                out["evaluation"][f"{tag}/obs{k}_{snr}" + ("" if falloff == "reference" else "/inverse_square") + ("/carry_hidden" if carry else "/fresh_hidden")] = {
                    "success_rate": s["success_rate"], "successful_episode_length_median": s["successful_episode_length_median"],
                    "total_episode_length_median": s["total_episode_length_median"],
                    "successful_episode_return_median": s["successful_episode_return_median"],
                    "unsuccessful_episode_return_median": s["unsuccessful_episode_return_median"],
                    "episodes": s["completed_runs"], "seconds": time.time() - t0}
    # the same models under the TRAINING distribution (the env's own spawn rules, U{1..5} rectangles resampled per epoch, sampled
    # actions): one collector epoch of 1024 envs x 480 steps, no update -- directly comparable with the authors' progress.txt rows
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.rada2c import RNNCollector
    out["training_distribution"] = {}
    for tag in ("og", "glatt", "rhine0"):
        env = RadSearchVec(1024, number_agents=1, obstruction_count=-1, enforce_grid_boundaries=True, seed=2)
        ag = RNNAgentPPO(id=0, device="cuda:0")
        z = np.load(os.path.join(G, f"ckpt_rada2c_{tag}.npz"))
        ag.agent.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
        col = RNNCollector(env, {0: ag}, 480, 120)
        col.collect()                                              # epoch 0 starts every env at once: take the second epoch
        st = col.collect()
        eps, done = float(st["EpCount"]), float(st["DoneCount"][0])
        out["training_distribution"][tag] = {"episodes": eps, "DoneCount": done, "success_rate": done / max(eps, 1.0),
                                             "mean_EpLen": float(st["EpLenSum"]) / max(eps, 1.0),
                                             "mean_episode_return": float(st["EpRetSum"][0]) / max(eps, 1.0)}
    out["og_original_pipeline"] = {
        "reading_only_clip8": original_pipeline_episode_stats(),
        "reading_only_no_clip": original_pipeline_episode_stats(clip=1e30),
        "reading_only_clip8_no_obstacles": original_pipeline_episode_stats(obstruction_count=0),
        "whole_vector_clip8": original_pipeline_episode_stats(whole_vector=True)}
    print(json.dumps(out, indent=1))


@torch.no_grad()
def original_pipeline_episode_stats(N=8192, L=120, seed=5, whole_vector=False, clip=8.0, obstruction_count=-1):
    """`og` was trained by the ORIGINAL single-agent RAD-A2C loop (algos/original_goal/ppo/ppo.py:424-539), whose input transform differs
    from the multi-agent trainer's: the WHOLE observation vector is shifted by the running mean of the READINGS, divided by their running
    standard deviation (1 while it is 0; core.py:53-77 StatBuff) and clipped to [-8, 8] (ppo.py:429).  This runs that transform in front
    of K11 / K14 and the HIP env under the env's own spawn rules (U{1..5} rectangles): one episode per env, to set beside the last rows
    of og/progress.txt (DoneCount / episodes, EpLen)."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.pfgru import PredictorBank, hash_uniform
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    dev = torch.device("cuda:0")
    ag = RNNAgentPPO(id=0, device=dev)
    z = np.load(os.path.join(G, "ckpt_rada2c_og.npz"))
    ag.agent.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    vec = RadSearchVec(N, number_agents=1, obstruction_count=obstruction_count, enforce_grid_boundaries=True, seed=seed)
    obs = vec.reset()[0].clone()
    cnt = torch.ones(N, dtype=torch.float64, device=dev)
    mu = obs[:, 0, 0].double().clone(); sq = torch.zeros_like(mu); sig = torch.ones_like(mu)
    bank = PredictorBank(N, 1, seed=seed, carry_hidden=True, device=dev)
    bank.cells[0] = ag.agent.model
    bank.reset()
    gk = (bank._base[0] * 1000003 + 5).view(-1, 1) * 1048583 + torch.arange(24, dtype=torch.int64, device=dev).view(1, -1)
    hid = ag.agent.gru_h0(hash_uniform(gk)).contiguous()
    alive = torch.ones(N, dtype=torch.bool, device=dev)
    ep_len = torch.zeros(N, dtype=torch.int32, device=dev); ep_ret = torch.zeros(N, device=dev)
    success = torch.zeros(N, dtype=torch.bool, device=dev)
    u = torch.empty(N, 1, device=dev); act8 = torch.empty(N, 1, dtype=torch.int8, device=dev)
    a = torch.empty(N, dtype=torch.int64, device=dev)
    for _ in range(L):
        if whole_vector:
            x = torch.clamp((obs.double() - mu.view(N, 1, 1)) / sig.view(N, 1, 1), -clip, clip).float().contiguous()
        else:                                                      # the reading only (the published RAD-A2C loop), clipped
            x = obs.clone()
            x[:, 0, 0] = torch.clamp((obs[:, 0, 0].double() - mu) / sig, -clip, clip).float()
        vec.action_uniforms(u)
        ag.policy_step_hip(x[:, 0].contiguous(), bank.predict(x)[:, 0].contiguous(), hid, u=u[:, 0].contiguous(), h_out=hid, act=a)
        act8[:, 0] = torch.where(alive, a, torch.full_like(a, 8)).to(torch.int8)
        obs_n, rew, _, done, _ = vec.step(act8)
        ep_ret += torch.where(alive, rew[:, 0], torch.zeros_like(rew[:, 0])); ep_len += alive.int()
        found = done[:, 0].bool() & alive
        success |= found; alive &= ~found
        r = obs_n[:, 0, 0].double()
        cnt += 1
        mu_n = mu + (r - mu) / cnt
        sq = sq + (r - mu) * (r - mu_n)
        mu = mu_n
        sig = torch.sqrt(sq / (cnt - 1)); sig = torch.where(sig == 0, torch.ones_like(sig), sig)
        obs = obs_n.clone()
        if not bool(alive.any()):
            break
    return {"episodes": N, "success_rate": float(success.float().mean()), "mean_EpLen": float(ep_len.float().mean()),
            "mean_episode_return": float(ep_ret.mean()), "median_EpLen_successful": float(ep_len[success].float().median()),
            "note": "og checkpoint, original RAD-A2C input transform, env spawn rules with U{1..5} rectangles, one episode per env"}


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
