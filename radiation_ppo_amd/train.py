"""train_PPO -- same constructor and .train() entry point as the reference's trainer
(algos/multiagent/train.py:77-156, :259-627), driving N lock-step GPU environments instead of one.

`env` may be a radiation_ppo_amd.envs.RadSearchVec (N envs) or the 1-env RadSearch adapter.  The
epoch x step double loop, the episode/epoch cut rules, the bootstrap rule, the epoch_end hand-shake,
the update and the logger columns follow train.py:321-627; the per-step work runs batched on the
device (radiation_ppo_amd.ppo.Collector / FusedCollector, radiation_ppo_amd.ppo_cnn.CNNCollector).  With
torch.distributed initialised (one process per GPU, backend "nccl" = RCCL) each rank owns a shard of the
envs and gradients, the KL estimate and the advantage statistics are all-reduced (the reference's
mpi_avg_grads / mpi_avg / mpi_statistics_scalar call sites, ppo.py:445,1250,1256).

Logging and checkpoints are the reference's (radiation_ppo_amd.logger.EpochLogger; progress.txt columns of
train.py:605-627 with two throughput columns appended; CNN models as <agent dir>/actor.pt + critic.pt,
RADTEAM_core.py:1904-1943; the MLP as <agent dir>/pyt_save/model.pt, epoch_logger.py:216-284) plus what the
reference lacks for a true resume: optimiser moments, step counts and the LR-schedule position (resume.pt).
"""
import math
import os
import time
from dataclasses import dataclass, field
from typing import Any, Dict, Union

import torch
import torch.distributed as dist

from .envs import RadSearch, RadSearchVec
from .logger import EpochLogger, convert_json, setup_logger_kwargs
from .maps import CNNCritic, heat_map_geometry
from .ppo import Collector, FusedCollector, VecAgentPPO
from .ppo_cnn import CNNAgentPPO, CNNCollector
from .rada2c import RNNAgentPPO, RNNCollector

# progress.txt columns of the reference (train.py:605-627, epoch_logger.py:393-398) + throughput columns of this build
COLUMNS = ["AgentID", "Epoch", "MeanVVals", "StdVVals", "MaxVVals", "MinVVals", "TotalEnvInteracts", "loss_policy",
           "loss_critic", "loss_predictor", "LocLoss", "Entropy", "kl_divergence", "ClipFrac", "OutOfBound",
           "stop_iteration", "MeanEpRet", "StdEpRet", "MaxEpRet", "MinEpRet", "DoneCount", "EpLen", "Time",
           "EnvStepsPerSec", "PPOItersPerSec"]


@dataclass
class train_PPO:
    """Signature and defaults of algos/multiagent/train.py:77-154."""
    env: Union[RadSearchVec, RadSearch]
    logger_kwargs: Dict[str, Any] = field(default_factory=dict)
    ppo_kwargs: Dict[str, Any] = field(default_factory=dict)
    seed: int = 0
    number_of_agents: int = 1
    actor_critic_architecture: str = "cnn"
    global_critic_flag: bool = True
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

    def _make_loggers(self) -> None:
        """train.py:166-190.  logger_kwargs is either the reference's {exp_name, seed, data_dir, env_name} (a parent
        "general" directory with config.json + one `<id>_agent_<exp_name>` directory per agent) or {"output_dir": d}
        (agent directories d/<id>_agent); empty: nothing is written (rows stay available in `loggers[i].rows`)."""
        kw = self.logger_kwargs or {}
        write = self.rank == 0
        cfg = convert_json(dict(ppo_kwargs=self.ppo_kwargs, seed=self.seed, steps_per_epoch=self.steps_per_epoch,
                                steps_per_episode=self.steps_per_episode, total_epochs=self.total_epochs,
                                number_of_agents=self.number_of_agents, actor_critic_architecture=self.actor_critic_architecture,
                                global_critic_flag=self.global_critic_flag, save_freq=self.save_freq,
                                num_envs=self.vec.num_envs * self.world, world_size=self.world))
        self.parent_logger = None
        if write and "data_dir" in kw:
            pk = setup_logger_kwargs(exp_name="general", seed=kw.get("seed"), data_dir=kw["data_dir"], env_name=kw.get("env_name"))
            self.parent_logger = EpochLogger(**pk)
            self.parent_logger.save_config(cfg)
        self.loggers = {}
        # where rank 0's first agent writes: every rank leaves its own env / collector state there (save_resume)
        self._dir0 = None
        if "data_dir" in kw:
            self._dir0 = setup_logger_kwargs(exp_name=f"0_agent_{kw.get('exp_name', '')}", seed=kw.get("seed"), data_dir=kw["data_dir"],
                                             env_name=kw.get("env_name"))["output_dir"]
        elif kw.get("output_dir"):
            self._dir0 = os.path.join(str(kw["output_dir"]), "0_agent")
        for i in range(self.number_of_agents):
            if write and "data_dir" in kw:
                lk = setup_logger_kwargs(exp_name=f"{i}_agent_{kw.get('exp_name', '')}", seed=kw.get("seed"),
                                         data_dir=kw["data_dir"], env_name=kw.get("env_name"))
            elif write and kw.get("output_dir"):
                lk = dict(output_dir=os.path.join(str(kw["output_dir"]), f"{i}_agent"), exp_name=kw.get("exp_name"))
            else:
                lk = dict(output_dir=None)
            self.loggers[i] = EpochLogger(**lk)
        if write and self.parent_logger is None and self.loggers[0].output_dir:
            self.loggers[0].save_config(cfg)

    def __post_init__(self) -> None:
        if self.actor_critic_architecture != "cnn" and self.global_critic_flag:
            raise ValueError("Global critic not supported in RAD-A2C")        # train.py:157-160
        if self.actor_critic_architecture not in ("ff", "mlp", "cnn", "rnn"):
            raise ValueError("Unsupported Neural Network type requested")      # ppo.py:666-667
        if self.render or self.save_gif:
            raise NotImplementedError("rendering is outside the hot path")
        if self.seed:
            torch.manual_seed(self.seed)                                       # train.py:162-164
        self.vec: RadSearchVec = self.env._vec if isinstance(self.env, RadSearch) else self.env
        assert self.vec.number_agents == self.number_of_agents
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self._make_loggers()
        kw = dict(self.ppo_kwargs)
        kw.pop("actor_critic_architecture", None)
        kw.setdefault("steps_per_epoch", self.steps_per_epoch)
        kw.setdefault("steps_per_episode", self.steps_per_episode)
        kw.setdefault("number_of_agents", self.number_of_agents)
        self.start_time = time.time()
        self.epochs_done = 0
        if self.actor_critic_architecture == "cnn":
            gc = gco = None
            # CNNBase.__post_init__ (RADTEAM_core.py:1727-1738): 27 x 27 maps with enforced walls, 147 x 147 without
            kw.setdefault("map_dim", heat_map_geometry(self.vec, self.steps_per_episode, bool(self.vec.cfg.enforce_grid_boundaries))[2])
            if self.global_critic_flag:                                        # train.py:191-206
                gc = CNNCritic(map_dim=kw["map_dim"]).to(self.vec.device)
                gco = torch.optim.Adam(gc.parameters(), lr=kw.get("critic_learning_rate", 1e-3))
            kw.pop("GlobalCriticOptimizer", None)
            kw.setdefault("seed", self.seed)                                   # keys the minibatch draws (ppo.py:754-766)
            self.collector = None
            self.agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, device=self.vec.device, **kw)
                           for i in range(self.number_of_agents)}
            for ag in self.agents.values():
                ag.sync_params()
            self.collector = CNNCollector(self.vec, self.agents, self.steps_per_epoch, self.steps_per_episode,
                                          global_critic_flag=self.global_critic_flag,
                                          use_predictor=bool(kw.get("use_predictor", True)),
                                          predictor_hidden_size=int(kw.get("predictor_hidden_size", 24)))
            self.collector.sync_predictors()                   # created after sync_params, from each rank's own generator
            return
        if self.actor_critic_architecture == "rnn":                           # RAD-A2C: GRU actor-critic + PFGRU (rada2c.py)
            kw.setdefault("seed", self.seed)
            self.agents = {i: RNNAgentPPO(id=i, device=self.vec.device, **kw) for i in range(self.number_of_agents)}
            for i, ag in self.agents.items():
                ag.sync_params()
                self.loggers[i].setup_pytorch_saver(ag.agent)                  # train.py:221-226
            self.collector = RNNCollector(self.vec, self.agents, self.steps_per_epoch, self.steps_per_episode)
            return
        self.agents = {i: VecAgentPPO(id=i, actor_critic_architecture=self.actor_critic_architecture,
                                      device=self.vec.device, **kw) for i in range(self.number_of_agents)}
        for i, ag in self.agents.items():
            ag.sync_params()                                                   # train.py:248-256
            self.loggers[i].setup_pytorch_saver(ag.agent)                      # train.py:221-226
        fusable = (self.number_of_agents == 1 and self.vec.num_envs % 16 == 0 and self.vec.cfg.geom_group_size == 1
                   and not self.global_critic_flag)
        cls = FusedCollector if fusable else Collector      # one launch per epoch (rs_rollout) when the config allows
        self.collector = cls(self.vec, self.agents, self.steps_per_epoch, self.steps_per_episode,
                             global_critic_flag=self.global_critic_flag)

    # ------------------------------------------------------------------ checkpoints
    def save(self) -> None:
        """train.py:552-561: every save_freq epochs, BEFORE the epoch's update, the reference's model files: CNN agents
        <agent dir>/actor.pt + critic.pt (RADTEAM_core.py:1904-1943), the MLP <agent dir>/pyt_save/model.pt (epoch_logger.py:216-284)."""
        for i, ag in self.agents.items():
            d = self.loggers[i].output_dir
            if not d:
                continue
            if self.actor_critic_architecture == "cnn":
                ag.save(d)
            else:
                self.loggers[i].save_state({}, None)

    def save_resume(self) -> None:
        """After the update of the same epochs (called on EVERY rank).  Rank 0: <agent dir>/resume.pt = weights + optimiser moments +
        step counts + LR-schedule position + the epoch counter (what the reference lacks for a true resume) -- the
        rank-identical part.  Every rank: <first agent dir>/resume_rank<r>.pt = ITS env workspace (positions, sources, rectangles,
        Philox counters of its own env ids), its collector's running episode state (Welford / heat maps / particle sets / GRU states,
        episode and epoch counters that key the draws) and its host generator: a rank that continued from rank 0's state would repeat
        rank 0's envs under its own draw keys."""
        for i, ag in self.agents.items():
            d = self.loggers[i].output_dir
            if d:
                torch.save(dict(agent=ag.resume_state(), epochs_done=self.epochs_done, world_size=self.world), os.path.join(d, "resume.pt"))
        if self._dir0:
            os.makedirs(self._dir0, exist_ok=True)
            torch.save(dict(collector=self.collector.resume_state(), torch_rng=torch.get_rng_state(), world_size=self.world,
                            epochs_done=self.epochs_done), os.path.join(self._dir0, f"resume_rank{self.rank}.pt"))

    def load(self, directory: str) -> None:
        """Resume from agent directories `<directory>/<id>_agent...` written by save_resume(): weights, optimiser moments, step
        counts, LR-schedule position from resume.pt, and -- from the first agent's resume_rank<r>.pt -- THIS rank's env state, collector
        state and host generator, so that the resumed run continues the saved one draw for draw on every rank
        (tests/test_train_logging_gpu.py; two ranks: tests/test_bench_dist_gpu.py).  Resuming with another world size raises: the
        envs would be dealt differently and the run could not be the saved one."""
        first_dir = None
        for i, ag in self.agents.items():
            cands = [os.path.join(directory, n) for n in sorted(os.listdir(directory)) if n.startswith(f"{i}_agent")]
            hits = ([c for c in cands if os.path.exists(os.path.join(c, "resume.pt"))]
                    or [os.path.join(c, s) for c in cands for s in sorted(os.listdir(c))
                        if os.path.exists(os.path.join(c, s, "resume.pt"))])
            if not hits:
                raise FileNotFoundError(f"no resume.pt for agent {i} under {directory}")
            # tensors, containers and plain scalars only: the restricted loader suffices
            st = torch.load(os.path.join(hits[0], "resume.pt"), map_location=self.vec.device, weights_only=True)
            if int(st.get("world_size", 1)) != self.world:
                raise ValueError(f"resume.pt was written by {st.get('world_size', 1)} rank(s), this run has {self.world}")
            ag.load_resume_state(st["agent"])
            self.epochs_done = int(st["epochs_done"])
            first_dir = first_dir or hits[0]
            if "collector" in st and self.world == 1:              # files of earlier versions: one rank, everything in resume.pt
                self.collector.load_resume_state(st["collector"])
                torch.set_rng_state(st["torch_rng"].cpu())
        f = os.path.join(first_dir, f"resume_rank{self.rank}.pt")
        if os.path.exists(f):
            st = torch.load(f, map_location=self.vec.device, weights_only=True)
            if int(st["world_size"]) != self.world or int(st["epochs_done"]) != self.epochs_done:
                raise ValueError(f"{f} belongs to another run (world size {st['world_size']}, epoch {st['epochs_done']})")
            self.collector.load_resume_state(st["collector"])
            torch.set_rng_state(st["torch_rng"].cpu())
        elif self.world > 1:
            raise FileNotFoundError(f"{f}: every rank resumes from its own env / collector state")

    # ------------------------------------------------------------------ the loop
    def train(self) -> None:
        """train.py:259-627."""
        self.start_time = time.time()
        T, N = self.steps_per_epoch, self.vec.num_envs * self.world
        A = self.number_of_agents
        for epoch in range(self.epochs_done, self.total_epochs):
            t0 = time.time()
            stats = self.collector.collect()
            saving = (epoch % self.save_freq == 0) or (epoch == self.total_epochs - 1)
            if saving and self.rank == 0:
                self.save()                                                             # train.py:552-561 (before the update)
            if epoch > 99 and self.actor_critic_architecture == "rnn":                  # train.py:563-566
                for ag in self.agents.values():
                    ag.reduce_pfgru_training()
            results = self.collector.update()
            self.epochs_done = epoch + 1
            if saving:
                self.save_resume()
            # ---- epoch statistics: one packed reduction over ranks, one host sync
            val = self.collector.buf.val.double()                                       # [T, N_local, A]
            vs = torch.stack([val.sum(dim=(0, 1)), (val * val).sum(dim=(0, 1))])        # [2, A]
            sums = torch.cat([stats["DoneCount"], stats["OutOfBound"], stats["EpRetSum"], stats["EpRetSqSum"],
                              stats["EpCount"].view(1), stats["EpLenSum"].view(1), vs.reshape(-1)])
            mx = torch.cat([stats["EpRetMax"], -stats["EpRetMin"], val.amax(dim=(0, 1)), -val.amin(dim=(0, 1))])
            if self.world > 1:
                dist.all_reduce(sums)
                dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sums, mx = sums.tolist(), mx.tolist()
            k = len(stats["DoneCount"])                 # A, or 1 for the single-agent fused collector
            done_c, oob_c, ret_s, ret_q = (sums[j * k:(j + 1) * k] for j in range(4))
            ep_c, len_s = sums[4 * k], sums[4 * k + 1]
            v_s, v_q = sums[4 * k + 2:4 * k + 2 + A], sums[4 * k + 2 + A:4 * k + 2 + 2 * A]
            r_max, r_min = mx[:k], [-x for x in mx[k:2 * k]]
            v_max, v_min = mx[2 * k:2 * k + A], [-x for x in mx[2 * k + A:2 * k + 2 * A]]
            flags = self.vec.error_flags()      # the reference raises on these states (rad_search_env.py:544-565)
            if flags:
                raise RuntimeError(f"RadSearch env error flags 0x{flags:x} (see RS_ENVERR_* in include/radsearch.h)")
            dt = time.time() - t0
            n_v = T * N
            for i in self.agents:
                lg, r, j = self.loggers[i], results[i], min(i, k - 1)
                v_mean = v_s[i] / n_v
                v_std = math.sqrt(max(v_q[i] / n_v - v_mean * v_mean, 0.0))
                n_ep = max(ep_c, 1.0)
                e_mean = ret_s[j] / n_ep
                e_std = math.sqrt(max(ret_q[j] / n_ep - e_mean * e_mean, 0.0))
                nan = float("nan")
                lg.log_tabular("AgentID", i)                                            # train.py:605-627
                lg.log_tabular("Epoch", epoch)
                lg.log_stats("VVals", v_mean, v_std, v_max[i], v_min[i], with_min_and_max=True)
                lg.log_tabular("TotalEnvInteracts", (epoch + 1) * T * N)
                lg.log_tabular("loss_policy", r.loss_policy)
                lg.log_tabular("loss_critic", r.loss_critic)
                lg.log_tabular("loss_predictor", r.loss_predictor)
                lg.log_tabular("LocLoss", r.LocLoss)
                lg.log_tabular("Entropy", r.Entropy)
                lg.log_tabular("kl_divergence", r.kl_divergence)
                lg.log_tabular("ClipFrac", r.ClipFrac)
                lg.log_tabular("OutOfBound", oob_c[j] / N)                              # mean over envs of the agent's count
                lg.log_tabular("stop_iteration", r.stop_iteration)
                if ep_c > 0:
                    lg.log_stats("EpRet", e_mean, e_std, r_max[j], r_min[j], with_min_and_max=True)
                else:
                    lg.log_stats("EpRet", nan, nan, nan, nan, with_min_and_max=True)
                lg.log_tabular("DoneCount", done_c[j])
                lg.log_tabular("EpLen", len_s / n_ep if ep_c > 0 else nan)
                lg.log_tabular("Time", time.time() - self.start_time)
                lg.log_tabular("EnvStepsPerSec", T * N / dt)
                lg.log_tabular("PPOItersPerSec", 1.0 / dt)
                lg.dump_tabular()
