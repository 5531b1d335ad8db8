"""Monte-Carlo evaluation on saved test environments: the device-side mirror of `EpisodeRunner.run`
(algos/multiagent/evaluate.py:333-476) for the RAD-A2C-style branch (2x64 MLP policy, per-episode standardisation
of the reading).

The reference walks one saved environment at a time: `refresh_environment` (rad_search_env.py:799-874), act until
the source is found or `steps_per_episode` is reached, record, refresh again, `montecarlo_runs` times.  Here every
(saved environment, Monte-Carlo run) pair is one env of a `RadSearchVec` -- E x R episodes advance in lock-step, each
with its own Philox stream -- and the same per-environment records come out (`MonteCarloResults`, same field names).

Saved sets use the reference's layout `env_dict["env_<i>"] = (src_coords, det_coords, intensity, bkg[, obstacles])`
(algos/test_environment/eval/test_env_gen.py:13-24).  `sample_test_environments` draws such a set from the
environment's own spawn rules (the reference's pickled sets are not read: they are untrusted binary files).
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List

import numpy as np
import torch

from . import _lib
from .envs import RadSearchVec
from .ppo import DeviceWelford, VecAgentPPO


@dataclass
class Results:                           # evaluate.py:34-40
    episode_length: List[int] = field(default_factory=list)
    episode_return: List[float] = field(default_factory=list)
    intensity: List[int] = field(default_factory=list)
    background_intensity: List[int] = field(default_factory=list)


@dataclass
class MonteCarloResults:                 # evaluate.py:60-68
    id: int
    completed_runs: int = 0
    success_counter: int = 0
    total_episode_length: List[int] = field(default_factory=list)
    successful: Results = field(default_factory=Results)
    unsuccessful: Results = field(default_factory=Results)


def median(data) -> np.float32:          # evaluate.py:78-80
    return np.median(data) if len(data) > 0 else np.nan


def sample_test_environments(count: int, obstruction_count: int = 0, enforce_grid_boundaries: bool = True, seed: int = 0,
                             device: str = "cuda:0", **env_kwargs: Any) -> Dict[str, tuple]:
    """A set of saved test environments in the reference's format, drawn by the environment's own reset."""
    vec = RadSearchVec(count, number_agents=1, obstruction_count=obstruction_count,
                       enforce_grid_boundaries=enforce_grid_boundaries, seed=seed, device=device, **env_kwargs)
    vec.reset()
    sx, sy = vec.state("src_x")[0].cpu().numpy(), vec.state("src_y")[0].cpu().numpy()
    x, y = vec.state("x")[0].cpu().numpy(), vec.state("y")[0].cpu().numpy()
    inten, bkg = vec.state("intensity")[0].cpu().numpy(), vec.state("bkg")[0].cpu().numpy()
    nob = vec.state("num_obs")[0].cpu().numpy()
    rect = vec.state("rect").cpu().numpy()                       # [28, G]
    out = {}
    for i in range(count):
        e = [np.array([float(sx[i]), float(sy[i])]), np.array([float(x[i]), float(y[i])]), int(inten[i]), int(bkg[i])]
        if obstruction_count != 0:
            obs = []
            for k in range(int(nob[i])):
                x0, y0, x1, y1 = (float(rect[4 * k + j, i]) for j in range(4))
                obs.append([np.array([[x0, y0], [x0, y1], [x1, y1], [x1, y0]])])
            e.append(obs)
        out["env_" + str(i)] = tuple(e)
    return out


def _pack(env_sets: Dict[str, tuple], runs: int, with_obstacles: bool, device):
    keys = sorted(env_sets, key=lambda k: int(k.split("_")[1]))
    E = len(keys)
    src = np.zeros((E, 2), dtype=np.int32); det = np.zeros((E, 2), dtype=np.int32)
    inten = np.zeros(E, dtype=np.int32); bkg = np.zeros(E, dtype=np.int32)
    nob = np.zeros(E, dtype=np.int32); rects = np.zeros((E, _lib.RS_MAX_OBS, 4), dtype=np.int32)
    for i, k in enumerate(keys):
        e = env_sets[k]
        for arr, p in ((src, e[0]), (det, e[1])):
            q = np.asarray(p, dtype=np.float64)
            if not np.all(q == np.round(q)):
                raise ValueError("saved coordinates must lie on the 1 cm lattice")
            arr[i] = q.astype(np.int32)
        inten[i], bkg[i] = int(e[2]), int(e[3])
        if with_obstacles:
            obstacles = e[4]
            if len(obstacles) > _lib.RS_MAX_OBS:
                raise ValueError("more than 7 obstructions")
            nob[i] = len(obstacles)
            for j, ob in enumerate(obstacles):
                pts = np.asarray(ob[0], dtype=np.float64)
                rects[i, j] = (pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max())
    rep = lambda a: torch.from_numpy(np.repeat(a, runs, axis=0)).to(device).contiguous()      # env e*runs + r
    return keys, rep(src), rep(det), rep(inten), rep(bkg), (rep(nob) if with_obstacles else None), (rep(rects) if with_obstacles else None)


@torch.no_grad()
def run_test_environments(agent: VecAgentPPO, env_sets: Dict[str, tuple], montecarlo_runs: int = 100, steps_per_episode: int = 120,
                          obstruction_count: int = 0, enforce_grid_boundaries: bool = True, seed: int = 0,
                          device: str = "cuda:0", return_actions: bool = False):
    """EpisodeRunner.run for every saved environment at once.  Returns (List[MonteCarloResults] in set order, summary
    dict with the statistics `evaluate.py:776-828` prints); with return_actions also the [steps, E*R] action log."""
    E, R, L = len(env_sets), montecarlo_runs, steps_per_episode
    N = E * R
    dev = torch.device(device)
    with_obs = obstruction_count != 0
    vec = RadSearchVec(N, number_agents=1, obstruction_count=obstruction_count, enforce_grid_boundaries=enforce_grid_boundaries,
                       seed=seed, device=device)
    keys, src, det, inten, bkg, nob, rects = _pack(env_sets, R, with_obs, dev)
    vec.reset()                                                   # a valid handle state; every episode is then loaded
    obs = vec.refresh(src, det, inten, bkg, nob, rects)[0].clone()
    stat = DeviceWelford((N, 1), dev)                             # evaluate.py:362-367
    stat.update(obs[..., 0])
    alive = torch.ones(N, dtype=torch.bool, device=dev)
    ep_len = torch.zeros(N, dtype=torch.int32, device=dev)
    ep_ret = torch.zeros(N, dtype=torch.float32, device=dev)
    success = torch.zeros(N, dtype=torch.bool, device=dev)
    u = torch.empty(N, 1, dtype=torch.float32, device=dev)
    act8 = torch.empty(N, 1, dtype=torch.int8, device=dev)
    log = []
    for _ in range(L):
        x = obs.clone()
        x[..., 0] = stat.standardize(obs[..., 0])
        vec.action_uniforms(u)
        a, _, _ = agent.agent.act(x[:, 0], u[:, 0])               # ac.step: sample from the policy (evaluate.py:373-383)
        act8[:, 0] = torch.where(alive, a, torch.full_like(a, 8)).to(torch.int8)      # finished episodes idle in place
        if return_actions:
            log.append(a.clone())
        obs_n, rew, _, done, _ = vec.step(act8)                   # finished envs keep stepping; their records are frozen
        ep_ret += torch.where(alive, rew[:, 0], torch.zeros_like(rew[:, 0]))       # :400-406 (float32 accumulation)
        ep_len += alive.int()
        found = done[:, 0].bool() & alive
        success |= found
        alive &= ~found
        stat.update(obs_n[..., 0], mask=alive)
        obs = obs_n.clone()
        if not bool(alive.any()):
            break
    flags = vec.error_flags()
    if flags:
        raise RuntimeError(f"RadSearch env error flags 0x{flags:x}")
    ep_len_c, ep_ret_c, suc = ep_len.cpu().numpy(), ep_ret.cpu().numpy(), success.cpu().numpy()
    i_c, b_c = inten.cpu().numpy(), bkg.cpu().numpy()
    out: List[MonteCarloResults] = []
    for e in range(E):
        res = MonteCarloResults(id=int(keys[e].split("_")[1]))
        for r in range(R):
            n = e * R + r
            bucket = res.successful if suc[n] else res.unsuccessful
            if r < 1:                                             # :425-431
                bucket.intensity.append(int(i_c[n])); bucket.background_intensity.append(int(b_c[n]))
            res.total_episode_length.append(int(ep_len_c[n]))
            res.success_counter += int(suc[n])
            bucket.episode_length.append(int(ep_len_c[n])); bucket.episode_return.append(float(ep_ret_c[n]))
        res.completed_runs = R
        out.append(res)
    done_len = [l for r in out for l in r.successful.episode_length]
    done_ret = [l for r in out for l in r.successful.episode_return]
    nd_ret = [l for r in out for l in r.unsuccessful.episode_return]
    summary = {"episodes": E, "montecarlo_runs": R, "completed_runs": N,
               "success_rate": float(suc.mean()), "success_count_median": float(np.median([r.success_counter for r in out])),
               "successful_episode_length_median": float(median(done_len)), "successful_episode_return_median": float(median(done_ret)),
               "unsuccessful_episode_return_median": float(median(nd_ret)),
               "total_episode_length_median": float(np.median(ep_len_c))}
    if return_actions:
        return out, summary, torch.stack(log).cpu().numpy()
    return out, summary
