"""train_PPO -- same constructor and .train() entry point as the reference's trainer
(algos/multiagent/train.py:77-156, :259-627), driving N lock-step GPU environments instead of one.

`env` may be a radiation_ppo_amd.envs.RadSearchVec (N envs) or the 1-env RadSearch adapter.  The
epoch x step double loop, the episode/epoch cut rules, the bootstrap rule, the epoch_end hand-shake,
the update and the logger columns follow train.py:321-627; the per-step work runs batched on the
device (radiation_ppo_amd.ppo.Collector / FusedCollector).  With torch.distributed initialised
(one process per GPU, backend "nccl" = RCCL) each rank owns a shard of the envs and gradients, the
KL estimate and the advantage statistics are all-reduced (the reference's mpi_avg_grads /
mpi_avg / mpi_statistics_scalar call sites, ppo.py:445,1250,1256).
"""
import json
import os
import time
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Union

import torch
import torch.distributed as dist

from .envs import RadSearch, RadSearchVec
from .ppo import Collector, FusedCollector, VecAgentPPO
from .ppo_cnn import CNNAgentPPO, CNNCollector
from .maps import CNNCritic

# progress.txt columns of the reference (train.py:605-627) + throughput columns of this build
COLUMNS = ["AgentID", "Epoch", "AverageVVals", "StdVVals", "MaxVVals", "MinVVals", "TotalEnvInteracts", "loss_policy",
           "loss_critic", "loss_predictor", "LocLoss", "Entropy", "kl_divergence", "ClipFrac", "OutOfBound",
           "stop_iteration", "AverageEpRet", "DoneCount", "EpLen", "Time", "EnvStepsPerSec", "PPOItersPerSec"]


class ProgressLogger:
    """Tab-separated progress.txt with the reference's column names (epoch_logger.py:286-337)."""

    def __init__(self, output_dir: Optional[str], exp_name: str = ""):
        self.rows = []
        self.file = None
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
            self.file = open(os.path.join(output_dir, "progress.txt"), "w")
            self.file.write("\t".join(COLUMNS) + "\n")
        self.output_dir = output_dir

    def dump(self, row: Dict[str, Any]) -> None:
        self.rows.append(row)
        if self.file:
            self.file.write("\t".join(str(row.get(c, "")) for c in COLUMNS) + "\n")
            self.file.flush()

    def save_config(self, cfg: Dict[str, Any]) -> None:
        if self.output_dir:
            with open(os.path.join(self.output_dir, "config.json"), "w") as f:
                json.dump(cfg, f, indent=4, sort_keys=True, default=str)


@dataclass
class train_PPO:
    """Signature of algos/multiagent/train.py:77-154."""
    env: Union[RadSearchVec, RadSearch]
    logger_kwargs: Dict[str, Any] = field(default_factory=dict)
    ppo_kwargs: Dict[str, Any] = field(default_factory=dict)
    seed: int = 0
    number_of_agents: int = 1
    actor_critic_architecture: str = "ff"
    global_critic_flag: bool = False
    steps_per_epoch: int = 480
    steps_per_episode: int = 120
    total_epochs: int = 3000
    render: bool = False
    save_path: str = "."
    save_freq: int = 500
    save_gif_freq: Union[int, float] = float("inf")
    save_gif: bool = False
    render_first_episode: bool = True
    episode_count: int = 0
    DEBUG: bool = False

    def __post_init__(self) -> None:
        if self.actor_critic_architecture != "cnn" and self.global_critic_flag:
            raise ValueError("Global critic not supported in RAD-A2C")        # train.py:157-160
        if self.actor_critic_architecture not in ("ff", "mlp", "cnn"):
            raise NotImplementedError(f"architecture {self.actor_critic_architecture!r}: the 2x64 MLP path ('ff', alias "
                                      "'mlp') and the RAD-TEAM CNN path ('cnn') are built; the GRU/PFGRU cores are SURVEY.md "
                                      "section 8 rows f1/f2")
        if self.render or self.save_gif:
            raise NotImplementedError("rendering is outside the hot path")
        if self.seed:
            torch.manual_seed(self.seed)                                       # train.py:162-164
        self.vec: RadSearchVec = self.env._vec if isinstance(self.env, RadSearch) else self.env
        assert self.vec.number_agents == self.number_of_agents
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        out = self.logger_kwargs.get("output_dir") if self.logger_kwargs else None
        self.loggers = {i: ProgressLogger(os.path.join(str(out), f"{i}_agent") if (out and self.rank == 0) else None)
                        for i in range(self.number_of_agents)}
        if out and self.rank == 0:
            self.loggers[0].save_config(dict(ppo_kwargs=self.ppo_kwargs, seed=self.seed,
                                             steps_per_epoch=self.steps_per_epoch, steps_per_episode=self.steps_per_episode,
                                             number_of_agents=self.number_of_agents, num_envs=self.vec.num_envs * self.world))
        kw = dict(self.ppo_kwargs)
        kw.pop("actor_critic_architecture", None)
        kw.setdefault("steps_per_epoch", self.steps_per_epoch)
        kw.setdefault("steps_per_episode", self.steps_per_episode)
        kw.setdefault("number_of_agents", self.number_of_agents)
        if self.actor_critic_architecture == "cnn":
            gc = gco = None
            if self.global_critic_flag:                                        # train.py:191-206
                gc = CNNCritic().to(self.vec.device)
                gco = torch.optim.Adam(gc.parameters(), lr=kw.get("critic_learning_rate", 1e-3))
            kw.pop("GlobalCriticOptimizer", None)
            self.agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, device=self.vec.device, **kw)
                           for i in range(self.number_of_agents)}
            for ag in self.agents.values():
                ag.sync_params()
            self.collector = CNNCollector(self.vec, self.agents, self.steps_per_epoch, self.steps_per_episode,
                                          global_critic_flag=self.global_critic_flag)
            self.start_time = time.time()
            return
        self.agents = {i: VecAgentPPO(id=i, actor_critic_architecture=self.actor_critic_architecture,
                                      device=self.vec.device, **kw) for i in range(self.number_of_agents)}
        for ag in self.agents.values():
            ag.sync_params()                                                   # train.py:248-256
        fusable = (self.number_of_agents == 1 and self.vec.num_envs % 16 == 0 and self.vec.cfg.geom_group_size == 1
                   and not self.global_critic_flag)
        cls = FusedCollector if fusable else Collector      # one launch per epoch (rs_rollout) when the config allows
        self.collector = cls(self.vec, self.agents, self.steps_per_epoch, self.steps_per_episode,
                             global_critic_flag=self.global_critic_flag)
        self.start_time = time.time()

    def train(self) -> None:
        """train.py:259-627."""
        self.start_time = time.time()
        T, N = self.steps_per_epoch, self.vec.num_envs * self.world
        for epoch in range(self.total_epochs):
            t0 = time.time()
            stats = self.collector.collect()
            results = self.collector.update()
            if (epoch % self.save_freq == 0) or (epoch == self.total_epochs - 1):      # train.py:552-561
                if self.rank == 0 and self.save_path and self.loggers[0].output_dir:
                    for i, ag in self.agents.items():
                        ag.save(os.path.join(self.loggers[i].output_dir, "model.pt"))
            pack = torch.stack([stats["DoneCount"].double(), stats["OutOfBound"].double(), stats["EpRetSum"],
                                stats["EpLenSum"], stats["EpCount"]])
            if self.world > 1:
                dist.all_reduce(pack)
            done_c, oob_c, ret_s, len_s, ep_c = pack.tolist()
            flags = self.vec.error_flags()      # the reference raises on these states (rad_search_env.py:544-565)
            if flags:
                raise RuntimeError(f"RadSearch env error flags 0x{flags:x} (see RS_ENVERR_* in include/radsearch.h)")
            dt = time.time() - t0
            for i in self.agents:
                v = self.collector.buf.val[:, :, i]
                r = results[i]
                self.loggers[i].dump(dict(
                    AgentID=i, Epoch=epoch, AverageVVals=v.mean().item(), StdVVals=v.std().item(), MaxVVals=v.max().item(),
                    MinVVals=v.min().item(), TotalEnvInteracts=(epoch + 1) * T * N, loss_policy=r.loss_policy,
                    loss_critic=r.loss_critic, loss_predictor=r.loss_predictor, LocLoss=r.LocLoss, Entropy=r.Entropy,
                    kl_divergence=r.kl_divergence, ClipFrac=r.ClipFrac, OutOfBound=oob_c / N, stop_iteration=r.stop_iteration,
                    AverageEpRet=ret_s / max(ep_c, 1.0), DoneCount=done_c, EpLen=len_s / max(ep_c, 1.0),
                    Time=time.time() - self.start_time, EnvStepsPerSec=T * N / dt, PPOItersPerSec=1.0 / dt))
