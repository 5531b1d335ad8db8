#!/usr/bin/env python3
"""Fixtures from the AUTHORS' saved models and test-environment sets ("success-rate / episode-length
parity against the authors' saved models").  Runs in the build container only (the reference tree does not travel):

    python tests/golden/make_checkpoints.py [/root/reference]

* RAD-A2C checkpoints  algos/multiagent/evaluation/saves/<run>/<id>_agent*/pyt_save/model.pt  -> tests/golden/ckpt_rada2c_<tag>.npz
  (one array per state_dict key).  Read with torch.load(..., weights_only=True): the restricted unpickler that constructs
  tensors and containers only -- never a plain unpickle of an untrusted file.
* RAD-TEAM checkpoints (saves/2023-03-02-*/<id>agent*/{actor,critic,predictor}.pt) -> tests/golden/ckpt_radteam_shapes.json:
  key names and shapes ONLY.  They were written by an older revision of the reference (5-channel actor AND critic, duplicate
  step<k> / actor.<k> keys, a 64-unit PFGRU with a BatchNorm) and do not fit the reference's present CNNBase (6 actor / 4 critic
  channels, 24-unit PFGRU, algos/test_cnn/RADTEAM_core.py:962-1023,1211-1271,1533-1585) -- its own load_state_dict(strict) rejects
  them -- so there is nothing to evaluate; the table documents that.
* Saved test sets: the first 100 environments (what evaluate.py's `episodes: 100` uses, main.py eval defaults) of
  test_env_dict_obs{0,3}_{high,low}_v4, read WITHOUT unpickling by radiation_ppo_amd.testsets -> tests/golden/testset_obs<k>_<snr>.npz
  (src, det, intensity, bkg, rects as plain arrays).
* The authors' own numbers for those models: the last 100 epochs of each run's progress.txt (DoneCount, EpLen, episode return)
  -> tests/golden/ckpt_rada2c_authors_log.json.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

RADA2C = {
    "og": "og/0_env_test_agent-WORKS/pyt_save/model.pt",
    "glatt": "2023-04-17-15:38:48/0_agent_2023-04-17-15:38:48_rada2c-default-glatt_agents1_s2/pyt_save/model.pt",
    "rhine0": "2023-04-19-19:48:05/0_agent_2023-04-19-19:48:05_rhine-RADA2C-2Agent_agents2_s2/pyt_save/model.pt",
    "rhine1": "2023-04-19-19:48:05/1_agent_2023-04-19-19:48:05_rhine-RADA2C-2Agent_agents2_s2/pyt_save/model.pt",
}
LOGS = {
    "og": ("og/progress.txt", "AverageEpRet", 4800),
    "glatt": ("2023-04-17-15:38:48/0_agent_2023-04-17-15:38:48_rada2c-default-glatt_agents1_s2/progress.txt", "MeanEpRet", 480),
    "rhine0": ("2023-04-19-19:48:05/0_agent_2023-04-19-19:48:05_rhine-RADA2C-2Agent_agents2_s2/progress.txt", "MeanEpRet", 480),
}
RADTEAM_RUNS = ["2023-03-02-13:39:06", "2023-03-02-17:10:28", "2023-04-14-17:30:17"]
SETS = [(0, "high"), (0, "low"), (3, "high"), (3, "low")]


def state_dict_of(path):
    sd = torch.load(path, weights_only=True, map_location="cpu")
    assert isinstance(sd, dict) and all(isinstance(v, torch.Tensor) for v in sd.values()), path
    return sd


def main(ref="/root/reference"):
    saves = os.path.join(ref, "algos", "multiagent", "evaluation", "saves")
    for tag, rel in RADA2C.items():
        sd = state_dict_of(os.path.join(saves, rel))
        np.savez_compressed(os.path.join(HERE, f"ckpt_rada2c_{tag}.npz"), **{k: v.numpy() for k, v in sd.items()})
        print(tag, len(sd), "tensors,", sum(v.numel() for v in sd.values()), "floats")

    shapes = {}
    for run in RADTEAM_RUNS:
        for d in sorted(os.listdir(os.path.join(saves, run))):
            for f in ("actor.pt", "critic.pt", "predictor.pt"):
                p = os.path.join(saves, run, d, f)
                if os.path.exists(p):
                    shapes[f"{run}/{d}/{f}"] = {k: list(v.shape) for k, v in state_dict_of(p).items()}
    with open(os.path.join(HERE, "ckpt_radteam_shapes.json"), "w") as fh:
        json.dump(shapes, fh, indent=1, sort_keys=True)

    logs = {}
    for tag, (rel, ret_col, steps) in LOGS.items():
        with open(os.path.join(saves, rel)) as fh:
            rows = [l.rstrip("\n").split("\t") for l in fh]
        hdr, body = rows[0], rows[-100:]
        col = lambda name: [float(r[hdr.index(name)]) for r in body]
        done, eplen, ret = col("DoneCount"), col("EpLen"), col(ret_col)
        logs[tag] = {"source": rel, "epochs_logged": len(rows) - 1, "last_epochs": len(body), "env_steps_per_epoch": steps,
                     "mean_DoneCount_per_epoch": float(np.mean(done)), "mean_EpLen": float(np.mean(eplen)),
                     "mean_episode_return": float(np.mean(ret)),
                     # episodes per epoch ~ steps / EpLen: the fraction of them that ended on the source
                     "approx_success_rate": float(np.mean(done) / (steps / np.mean(eplen)))}
    with open(os.path.join(HERE, "ckpt_rada2c_authors_log.json"), "w") as fh:
        json.dump(logs, fh, indent=1, sort_keys=True)

    from radiation_ppo_amd.testsets import load_test_environments
    for k, snr in SETS:
        sets = load_test_environments(os.path.join(ref, "algos", "multiagent", "evaluation", "test_environments", f"test_env_dict_obs{k}_{snr}_v4"))
        keys = sorted(sets, key=lambda s: int(s.split("_")[1]))[:100]
        src = np.stack([np.asarray(sets[q][0], dtype=np.float64) for q in keys])
        det = np.stack([np.asarray(sets[q][1], dtype=np.float64) for q in keys])
        inten = np.array([int(sets[q][2]) for q in keys], dtype=np.int64)
        bkg = np.array([int(sets[q][3]) for q in keys], dtype=np.int64)
        rects = np.zeros((len(keys), k, 4, 2), dtype=np.float64)
        for i, q in enumerate(keys):
            if k:
                assert len(sets[q][4]) == k
                for j, ob in enumerate(sets[q][4]):
                    rects[i, j] = np.asarray(ob[0], dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, f"testset_obs{k}_{snr}.npz"), src=src, det=det, intensity=inten, bkg=bkg, rects=rects)
        print("set", k, snr, src.shape, rects.shape)


if __name__ == "__main__":
    main(*sys.argv[1:])
