"""RAD-TEAM (CNN) PPO on the device: multi-agent collector over the K5 heat maps and the CNN update.

Mirrors (paths relative to the reference root):
  CNNCollector      <- the epoch loop of algos/multiagent/train.py:332-548 with CNNBase.select_action
                       (NeuralNetworkCores/RADTEAM_core.py:1838-1892) for every agent of every env
  CNNAgentPPO       <- AgentPPO, 'cnn' branch (algos/multiagent/ppo.py:620-642, :815-897): actor loss
                       -mean(min(r A, clip(r) A)) (:966-997) with KL early stop at 1.5*target_kl, then train_v_iters
                       critic MSE steps (:1020-1045); sample() draws all indexes below the summed length of the
                       COMPLETE episodes (:754-764); separate Adam + StepLR(100, 0.99) for actor and critic;
                       with a global critic only agent 0 updates it (:858)
Heat-map channel 0 (location prediction) is fed by every owner's PFGRU cell (radiation_ppo_amd/pfgru.py, SURVEY section 8 row
f1) exactly as the reference's CNN harness uses it: forward-only, untrained weights, every prediction made from the episode's
h0 (algos/test_cnn/train.py:686-693, ppo.py:737-738); `use_predictor=False` leaves the channel empty.
"""
import ctypes as C
import os
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from . import _lib
from .envs import RadSearchVec
from .maps import CNNActor, CNNCritic, HeatMaps, actor_stack_from
from .pfgru import PredictorBank, hash_bits, hash_uniform
from .ppo import (EpochStats, RolloutBuffer, UpdateResult, _world, check_minibatch, host_read, normalize_advantages,
                  reduce_grads_and_stats, reject_unknown_kwargs, side_stream)


def minibatch_weights(complete_len: torch.Tensor, T: int, minibatch: int, key: torch.Tensor, n_total: int) -> torch.Tensor:
    """sample() of update_agent's 'cnn' branch (ppo.py:754-766) for every env (= rank of the reference) at once: env n draws
    k_n = int(ep_len_n / minibatch) of the indexes [0, ep_len_n) uniformly WITHOUT replacement (np.random.choice(..., replace=False)),
    ep_len_n = the summed length of its complete episodes; the losses are then means over the drawn indexes (:934-940) and over ranks
    (mpi_avg_grads).  Returns w [T, N] float32: 1 / (k_n n_total) on the drawn samples of env n, 0 elsewhere.

    The draw: every index gets a counter-hash uniform (key [N] int64 = one value per env and draw, so the result does not depend on
    how the envs are sharded over GPUs); the k_n smallest are taken -- a uniformly distributed k-subset.  minibatch = 1 is every
    index (a permutation; the mean does not see the order)."""
    N = complete_len.shape[0]
    dev = complete_len.device
    k = torch.div(complete_len, minibatch, rounding_mode="floor")
    tt = torch.arange(T, device=dev, dtype=torch.int64).view(T, 1)
    valid = tt < complete_len.view(1, N)
    u = hash_uniform(hash_bits(key.view(1, N)) + tt)
    u = torch.where(valid, u, torch.full_like(u, 2.0))
    thr = u.sort(dim=0).values.gather(0, (k - 1).clamp(min=0).view(1, N))
    sel = (u <= thr) & valid & (k > 0).view(1, N)
    return sel.float() / (k.clamp(min=1).view(1, N).float() * n_total)


class ActorLoss(torch.autograd.Function):
    """compute_loss_pi behind the logits (ppo.py:966-1003) on rs_actor_loss: returns (loss, [kl, entropy, clip fraction, loss] float64);
    the derivative wrt the logits is formed in the same pass."""

    @staticmethod
    def forward(ctx, logits, act, adv, logp_old, w, clip):
        S = logits.shape[0]
        # the contiguous copies stay referenced until the launch is enqueued (a column of the [T, N, A] buffers flattens to a strided view)
        logits, act, adv, logp_old, w = (t.contiguous() for t in (logits, act, adv, logp_old, w))
        assert act.dtype == torch.int64 and all(t.dtype == torch.float32 for t in (logits, adv, logp_old, w))
        dl = torch.empty_like(logits)
        stats = torch.empty((S + 63) // 64, 4, dtype=torch.float32, device=logits.device)
        _lib.check(_lib.load().rs_actor_loss(logits.data_ptr(), act.data_ptr(), adv.data_ptr(), logp_old.data_ptr(), w.data_ptr(),
                                             dl.data_ptr(), stats.data_ptr(), S, float(clip),
                                             C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)), "rs_actor_loss")
        st = stats.double().sum(dim=0)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(st)
        return st[3].float(), st

    @staticmethod
    def backward(ctx, g, _):
        (dl,) = ctx.saved_tensors
        return g * dl, None, None, None, None, None


class CNNAgentPPO:
    """AgentPPO, 'cnn' branch.  `minibatch` (ppo.py:580, :754-766) IS effective here: every actor iteration draws
    int(ep_len / minibatch) sample indexes per env, the critic iterations reuse the last draw (minibatch_weights above).  Keys of the
    reference's ppo_kwargs without a counterpart on this path are accepted (_NO_EFFECT: the reference's own 'cnn' branch does not train
    the PFGRU -- "TODO get PFGRU working with RAD-TEAM", ppo.py:849-852 -- and `actor_critic_args` describes its CNNBase, whose sizes
    come from the env here; `use_predictor` / `predictor_hidden_size` are read by train_PPO for the collector); others raise."""

    _NO_EFFECT = ("bp_args", "env_height", "actor_critic_args", "train_pfgru_iters", "pfgru_learning_rate", "use_predictor",
                  "predictor_hidden_size")

    def __init__(self, id: int, map_dim=(27, 27), action_space: int = 8, train_pi_iters: int = 40, train_v_iters: int = 40,
                 actor_learning_rate: float = 3e-4, critic_learning_rate: float = 1e-3, gamma: float = 0.99, alpha: float = 0.0,
                 clip_ratio: float = 0.2, target_kl: float = 0.07, lam: float = 0.9, GlobalCritic: Optional[CNNCritic] = None,
                 GlobalCriticOptimizer: Optional[torch.optim.Optimizer] = None, device="cuda:0", chunk: int = 524288,
                 observation_space: int = 11, steps_per_epoch: int = 480, steps_per_episode: int = 120, number_of_agents: int = 1,
                 minibatch: int = 1, seed: int = 0, **other: Any):
        reject_unknown_kwargs("CNNAgentPPO", other, self._NO_EFFECT)
        self.minibatch, self.seed = check_minibatch(minibatch), int(seed)
        self.id = id
        self.map_dim = tuple(map_dim)
        self.device = torch.device(device)
        self.gamma, self.lam, self.alpha = gamma, lam, alpha
        self.clip_ratio, self.target_kl = clip_ratio, target_kl
        self.train_pi_iters, self.train_v_iters = train_pi_iters, train_v_iters
        self.pi = CNNActor(map_dim=map_dim, action_dim=action_space).to(self.device)
        self.global_critic = GlobalCritic is not None
        self.critic = GlobalCritic if GlobalCritic is not None else CNNCritic(map_dim=map_dim).to(self.device)
        self.pi_optimizer = torch.optim.Adam(self.pi.parameters(), lr=actor_learning_rate)
        self.critic_optimizer = GlobalCriticOptimizer if GlobalCriticOptimizer is not None else torch.optim.Adam(
            self.critic.parameters(), lr=critic_learning_rate)
        self.pi_scheduler = torch.optim.lr_scheduler.StepLR(self.pi_optimizer, step_size=100, gamma=0.99)
        self.critic_scheduler = torch.optim.lr_scheduler.StepLR(self.critic_optimizer, step_size=100, gamma=0.99)
        self.chunk = chunk

    def sync_params(self) -> None:
        if _world() > 1:
            for mod in (self.pi, self.critic):
                flat = torch.cat([p.data.view(-1) for p in mod.parameters()])
                dist.broadcast(flat, src=0)
                o = 0
                for p in mod.parameters():
                    p.data.copy_(flat[o:o + p.numel()].view_as(p))
                    o += p.numel()

    def _logits(self, x):
        """x: dense [B,6,X,Y] stack (library convolutions; tests) or (maps, cells, pcells, agent) -> HIP trunk."""
        return self.pi.logits_from_maps(*x) if isinstance(x, tuple) else self.pi.logits(x)

    def _values(self, x):
        """x: dense [B,4,X,Y] stack or (maps,) resident shared maps -> HIP trunk."""
        return self.critic.value_from_maps(*x) if isinstance(x, tuple) else self.critic(x)

    @torch.no_grad()
    def act(self, actor_in, u: torch.Tensor):
        """select_action (RADTEAM_core.py:1838-1892): inverse-CDF sample from the actor's distribution."""
        logp_all = torch.log_softmax(self._logits(actor_in), dim=-1)
        cdf = torch.cumsum(logp_all.exp(), dim=-1)
        a = (cdf <= u.unsqueeze(-1)).sum(dim=-1).clamp_(max=logp_all.shape[-1] - 1)
        logp = logp_all.gather(-1, a.unsqueeze(-1)).squeeze(-1)
        return a, logp

    def update_agent(self, actor_in, critic_in, act, adv, ret, logp_old, w, update_critic: bool, w_for=None) -> UpdateResult:
        """actor_in(lo, hi) / critic_in(lo, hi) describe the inputs of samples [lo, hi) (see _logits / _values);
        w sums to 1 over the global batch.  w_for(kk) -> the weights of actor iteration kk (minibatch > 1: a fresh draw per
        iteration, ppo.py:831; the critic loop keeps the LAST draw, :861-864)."""
        M = act.shape[0]
        thr = 1.5 * self.target_kl
        kk, kl_reached, last = 0, False, None
        while not kl_reached and kk < self.train_pi_iters:                      # ppo.py:825-846
            if w_for is not None:
                w = w_for(kk)
            self.pi_optimizer.zero_grad(set_to_none=True)
            stats = torch.zeros(4, dtype=torch.float64, device=self.device)
            for lo in range(0, M, self.chunk):
                hi = min(lo + self.chunk, M)
                if self.device.type == "cuda" and getattr(self, "use_loss_kernel", True):
                    # the loss, its statistics and its derivative behind the logits in one launch (rs_actor_loss)
                    loss, st = ActorLoss.apply(self._logits(actor_in(lo, hi)), act[lo:hi], adv[lo:hi], logp_old[lo:hi], w[lo:hi], self.clip_ratio)
                    loss.backward()
                    stats += st
                    continue
                logp_all = torch.log_softmax(self._logits(actor_in(lo, hi)), dim=-1)
                logp = logp_all.gather(-1, act[lo:hi].unsqueeze(-1)).squeeze(-1)
                ratio = torch.exp(logp - logp_old[lo:hi])
                clip_adv = torch.clamp(ratio, 1 - self.clip_ratio, 1 + self.clip_ratio) * adv[lo:hi]
                loss = -(w[lo:hi] * torch.min(ratio * adv[lo:hi], clip_adv)).sum()      # ppo.py:984-988
                loss.backward()
                with torch.no_grad():
                    ent = -(logp_all.exp() * logp_all).sum(-1)
                    clipped = (ratio > 1 + self.clip_ratio) | (ratio < 1 - self.clip_ratio)
                    stats += torch.stack([(w[lo:hi] * (logp_old[lo:hi] - logp)).sum(), (w[lo:hi] * ent).sum(),
                                          (w[lo:hi] * clipped.float()).sum(), loss.detach()]).double()
            # ONE collective per actor iteration: the statistics (KL first) ride behind the gradients in the same bucket
            # (mpi_avg_grads + mpi_avg, ppo.py:841 / :838).  The KL decision is taken on the host: an iteration here is four
            # 524 288-sample chunks through K9 / K10 (tens of milliseconds), so the one read costs nothing and a stopped loop
            # must not enqueue further passes.
            last = host_read(reduce_grads_and_stats(self.pi.parameters(), stats))
            if last[0] < thr:                                                   # ppo.py:838-845
                self.pi_optimizer.step()
            else:
                kl_reached = True
            kk += 1
        self.pi_scheduler.step()
        loss_c = float("nan")
        if update_critic:                                                       # ppo.py:858-873
            tot = None
            for _ in range(self.train_v_iters):
                self.critic_optimizer.zero_grad(set_to_none=True)
                tot = torch.zeros(1, dtype=torch.float64, device=self.device)
                for lo in range(0, M, self.chunk):
                    hi = min(lo + self.chunk, M)
                    v = self._values(critic_in(lo, hi))
                    lc = (w[lo:hi] * (v - ret[lo:hi]) ** 2).sum()               # MSE (ppo.py:1040-1045)
                    lc.backward()
                    tot += lc.detach().double()
                tot = reduce_grads_and_stats(self.critic.parameters(), tot)     # one collective; no host round trip in this loop
                self.critic_optimizer.step()
            if tot is not None:
                loss_c = float(tot.item())
            self.critic_scheduler.step()
        return UpdateResult(stop_iteration=kk, loss_policy=last[3], loss_critic=loss_c, loss_predictor=0.0, kl_divergence=last[0],
                            Entropy=last[1], ClipFrac=last[2], LocLoss=0.0)

    def save(self, path: str) -> None:
        """CNNBase.save (RADTEAM_core.py:1904-1943): `<path>/actor.pt` and `<path>/critic.pt` hold the modules' state_dicts
        (keys actor.N.* / critic.N.*, :1170-1180, :1331-1342), so saved models interchange with the reference."""
        os.makedirs(path, exist_ok=True)
        torch.save(self.pi.state_dict(), os.path.join(path, "actor.pt"))
        torch.save(self.critic.state_dict(), os.path.join(path, "critic.pt"))
        if getattr(self, "model", None) is not None:                # PFGRUCell.save_model (:1654-1655)
            torch.save(self.model.state_dict(), os.path.join(path, "predictor.pt"))

    def load(self, path: str) -> None:
        """CNNBase.load (:1945-1953)."""
        for mod, name in ((self.pi, "actor.pt"), (self.critic, "critic.pt")):
            f = os.path.join(path, name)
            assert os.path.isfile(f), "Model does not exist"
            mod.load_state_dict(torch.load(f, map_location=self.device, weights_only=True))
        f = os.path.join(path, "predictor.pt")
        if getattr(self, "model", None) is not None and os.path.isfile(f):
            self.model.load_state_dict(torch.load(f, map_location=self.device, weights_only=True))

    def resume_state(self) -> Dict[str, Any]:
        st = dict(actor=self.pi.state_dict(), critic=self.critic.state_dict(), pi_optimizer=self.pi_optimizer.state_dict(),
                  critic_optimizer=self.critic_optimizer.state_dict(), pi_scheduler=self.pi_scheduler.state_dict(),
                  critic_scheduler=self.critic_scheduler.state_dict())
        if getattr(self, "model", None) is not None:               # the (untrained) predictor whose output fills heat-map channel 0
            st["model"] = self.model.state_dict()
        return st

    def load_resume_state(self, st: Dict[str, Any]) -> None:
        self.pi.load_state_dict(st["actor"]); self.critic.load_state_dict(st["critic"])
        self.pi_optimizer.load_state_dict(st["pi_optimizer"]); self.critic_optimizer.load_state_dict(st["critic_optimizer"])
        self.pi_scheduler.load_state_dict(st["pi_scheduler"]); self.critic_scheduler.load_state_dict(st["critic_scheduler"])
        if "model" in st and getattr(self, "model", None) is not None:
            self.model.load_state_dict(st["model"])


class CNNCollector:
    """Multi-agent RAD-TEAM collector: heat maps (K5) -> CNN actors/critic -> env lock-step -> buffer."""

    def __init__(self, env: RadSearchVec, agents: Dict[int, CNNAgentPPO], steps_per_epoch: int, steps_per_episode: int,
                 global_critic_flag: bool = True, use_predictor: bool = True, predictor_hidden_size: int = 24,
                 carry_hidden: bool = False, use_graph: bool = True):
        self.env, self.agents = env, agents
        self.T, self.L, self.N, self.A = steps_per_epoch, steps_per_episode, env.num_envs, env.number_agents
        self.team_reward = global_critic_flag
        dev = env.device
        self.maps = HeatMaps(env, steps_per_episode, enforce_boundaries=bool(env.cfg.enforce_grid_boundaries))
        X, Y = self.maps.map_dimensions
        for ag in agents.values():
            if tuple(ag.map_dim) != (X, Y):
                raise ValueError(f"agent {ag.id} was built for {ag.map_dim} maps, the environment produces {(X, Y)} "
                                 "(radiation_ppo_amd.maps.heat_map_geometry gives the size for an env)")
            if (X, Y) != (27, 27):
                # RADTEAM_core.py:1727-1738: without enforced walls the maps grow to 147 x 147.  K5 handles any size; the trunk goes
                # through the dense stack and the library convolutions there (maps.CNNActor.logits_from_maps): update chunks sized so
                # that one dense [chunk, 6, X, Y] stack stays near 1 GB
                ag.chunk = min(ag.chunk, max(64, (1 << 28) // (6 * X * Y)))
        need = self.T * self.N * 4 * X * Y * 4
        if need > 200e9:
            raise MemoryError(f"{X} x {Y} heat maps: the epoch's stored maps would take {need / 1e9:.0f} GB; use fewer envs")
        self.buf = RolloutBuffer(self.T, self.N, self.A, _lib.RS_OBS_DIM, dev)
        # per step only the four shared maps and the cell indices are stored; actor stacks are rebuilt on demand
        self.shared = torch.zeros(self.T, self.N, 4, X, Y, dtype=torch.float32, device=dev)
        self.cells = torch.zeros(self.T, self.N, self.A, dtype=torch.int64, device=dev)
        self.pcells = torch.zeros(self.T, self.N, self.A, dtype=torch.int64, device=dev)
        self.steps_in_ep = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.ep_ret = torch.zeros(self.N, self.A, dtype=torch.float32, device=dev)
        self._u = torch.empty(self.N, self.A, dtype=torch.float32, device=dev)
        self._act8 = torch.empty(self.N, self.A, dtype=torch.int8, device=dev)
        self.complete_len = torch.zeros(self.N, dtype=torch.int64, device=dev)
        self.epoch = 0                                  # updates done: keys the minibatch draws (and travels with resume.pt)
        # one lock-step of the loop is ~60 small launches (maps, 5 trunk + head evaluations, sampling, env step, bootstrap
        # round, resets): launch bound.  The step is therefore captured ONCE into a HIP graph and replayed T - 1 times per
        # epoch (the epoch's last step, which raises epoch_end, runs eagerly).  Everything a step reads or writes lives at a
        # fixed address: state is updated in place, the buffer row is addressed by a device-side step counter.
        self.use_graph = use_graph
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._t = torch.zeros(1, dtype=torch.int64, device=dev)
        self._acc = EpochStats(self.A, dev)
        self._row_act = torch.zeros(self.N, self.A, dtype=torch.int64, device=dev)
        self._row_f = torch.zeros(3, self.N, self.A, dtype=torch.float32, device=dev)      # logp, val, last_val of the step
        self.obs = None
        self._cs = None                                 # rs_collect_state: the lock-step's bookkeeping kernels (see _glue_state)
        # CNNBase.model (RADTEAM_core.py:1790-1795): one PFGRU cell per owner, all envs at once
        self.predictor: Optional[PredictorBank] = None
        if use_predictor:
            self.predictor = PredictorBank(self.N, self.A, hidden_size=predictor_hidden_size, seed=int(env.cfg.seed),
                                           env_id_base=int(env.cfg.env_id_base), carry_hidden=carry_hidden, device=dev)
            for a, ag in agents.items():
                ag.model = self.predictor.cells[a]                  # saved as predictor.pt next to actor.pt / critic.pt

    def start(self) -> None:
        obs, *_ = self.env.reset()
        self.obs = obs.clone()                                      # from here on updated in place (fixed address)
        if self.predictor is not None:
            self.predictor.reset()                                  # ac.reset_hidden() (test_cnn/train.py:686)

    def actor_stack_from(self, shared: torch.Tensor, cells: torch.Tensor, pcells: torch.Tensor, a: int) -> torch.Tensor:
        return actor_stack_from(shared, cells, pcells, a)

    @property
    def use_glue(self) -> bool:
        return self.env.device.type == "cuda" and (self.predictor is None or self.predictor.impl == "hip")

    def _glue_state(self) -> "_lib.RsCollectState":
        """rs_collect_state over the collector's buffers (rs_collect_post_step / _post_reset: the lock-step's returns, counters, cut
        flags, observation copies and draw counters in two launches instead of ~27 element-wise ones)."""
        if self._cs is None:
            env, dev, N, A = self.env, self.env.device, self.N, self.A
            self._flags = torch.zeros(3, N, dtype=torch.uint8, device=dev)                  # over, cut, boot
            self._rew_used = torch.zeros(N, A, dtype=torch.float32, device=dev)
            self._done_oob = torch.zeros(2, N, A, dtype=torch.uint8, device=dev)            # copies of the env's done / out_of_bounds rows
            self._src_copy = torch.zeros(2, N, dtype=torch.int32, device=dev)
            p = lambda t: None if t is None else t.data_ptr()
            pf = self.predictor
            self._cs = _lib.RsCollectState(N, A, self.L, 1 if self.team_reward else 0, p(env.obs), p(env.reward), p(env.team), p(env.done),
                                           p(self.obs), p(self.ep_ret), p(self.steps_in_ep), None, None, None, None, None, None, p(self._rew_used),
                                           p(self._flags[0]), p(self._flags[1]), p(self._flags[2]), p(pf.episode if pf else None),
                                           p(pf.calls if pf else None), None, p(self._t), p(env.oob), p(env.state("src_x")), p(env.state("src_y")),
                                           p(self._done_oob[0]), p(self._done_oob[1]), p(self._src_copy), p(self.complete_len))
        return self._cs

    @torch.no_grad()
    def _round_glued(self, mask8: Optional[torch.Tensor] = None):
        pred = None
        if self.predictor is not None:
            pred = self.predictor.predict_kernel(self.obs, mask8=mask8)
        self.maps.update(self.obs, pred=pred, mask=mask8)
        return self.maps.shared_maps(), self.maps.field("cell").long(), self.maps.field("pred_cell").long()

    @property
    def use_heads(self) -> bool:
        """The select_action round of every agent as trunk (one prepared launch) + BLAS Linear(2704, 32) + rs_cnn_head, buffer rows by
        rs_store_rows: the walls-enforced 27 x 27 maps only (the trunk kernels' size)."""
        return self.use_glue and tuple(self.maps.map_dimensions) == (27, 27) and getattr(self, "heads", True)

    def _prepare_heads(self) -> None:
        """Once per epoch (the networks do not change during collection): every agent's convolution weights in the trunk kernel's
        layout; the fixed-address buffers of the fused round."""
        lib, dev, N, A = _lib.load(), self.env.device, self.N, self.A
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if getattr(self, "_wt", None) is None:
            self._wt = {a: torch.empty(lib.rs_cnn_trunk_scratch_floats(6), dtype=torch.float32, device=dev) for a in self.agents}
            self._wt_c = {a: torch.empty(lib.rs_cnn_trunk_scratch_floats(4), dtype=torch.float32, device=dev) for a in self.agents}
            self._a2 = torch.empty(N, 2704, dtype=torch.float32, device=dev)
            self._k_act = torch.zeros(A, N, dtype=torch.int64, device=dev)
            self._k_f = torch.zeros(A, 3, N, dtype=torch.float32, device=dev)                   # logp, value, bootstrap value
            self._x_buf = torch.zeros(N, A, _lib.RS_OBS_DIM, dtype=torch.float32, device=dev)   # the step's observation (buffer row)
        for a, ag in self.agents.items():
            m = ag.pi.actor
            _lib.check(lib.rs_cnn_trunk_prepare(6, m[0].weight.data_ptr(), m[0].bias.data_ptr(), m[3].weight.data_ptr(), m[3].bias.data_ptr(),
                                                self._wt[a].data_ptr(), st), "rs_cnn_trunk_prepare")
            if not ag.global_critic or a == min(self.agents):
                c = ag.critic.critic
                _lib.check(lib.rs_cnn_trunk_prepare(4, c[0].weight.data_ptr(), c[0].bias.data_ptr(), c[3].weight.data_ptr(), c[3].bias.data_ptr(),
                                                    self._wt_c[a].data_ptr(), st), "rs_cnn_trunk_prepare")

    def _values_fused(self, critic_maps, slot: int, mask8=None) -> None:
        """V(s) of every env into self._k_f[:, slot] (slot 1: the step's value, 2: the bootstrap value): one evaluation serves every
        owner of a global critic (train.py:191-206)."""
        lib, N, A = _lib.load(), self.N, self.A
        st = C.c_void_p(torch.cuda.current_stream(self.env.device).cuda_stream)
        first = min(self.agents)
        for a, ag in self.agents.items():
            if ag.global_critic and a != first:
                continue
            c = ag.critic.critic
            _lib.check(lib.rs_cnn_trunk_infer(critic_maps.data_ptr(), None, None, 0, -1, N, self._wt_c[a].data_ptr(), self._a2.data_ptr(), st),
                       "rs_cnn_trunk_infer")
            y1 = torch.nn.functional.linear(self._a2, c[6].weight, c[6].bias)
            copies = A if ag.global_critic else 1
            _lib.check(lib.rs_cnn_head(y1.data_ptr(), c[8].weight.data_ptr(), c[8].bias.data_ptr(), c[10].weight.data_ptr(), c[10].bias.data_ptr(), 1,
                                       None, 1, None, None, None, 1, self._k_f[0 if ag.global_critic else a, slot].data_ptr(), copies, 3 * N,
                                       None if mask8 is None else mask8.data_ptr(), N, st), "rs_cnn_head")

    @torch.no_grad()
    def _step_glued(self, epoch_ended: bool) -> None:
        """_step with the element-wise bookkeeping between the library calls in rs_collect_post_step / _post_reset and (27 x 27 maps) every
        agent's select_action as trunk + Linear + rs_cnn_head, the buffer rows by rs_store_rows: ~40 launches per lock-step where the
        torch composition took ~110."""
        env, buf, N, A = self.env, self.buf, self.N, self.A
        acc, ti = self._acc, self._t
        lib, cs = _lib.load(), self._glue_state()
        st = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)
        over, cut, boot = self._flags[0], self._flags[1], self._flags[2]
        put = lambda dst, row: dst.index_copy_(0, ti, row.unsqueeze(0))
        heads = self.use_heads
        critic, cells, pcells = self._round_glued()
        put(self.shared, critic); put(self.cells, cells); put(self.pcells, pcells)
        env.action_uniforms(self._u)
        if heads:
            self._x_buf.copy_(self.obs)
            for a, ag in self.agents.items():
                m = ag.pi.actor
                _lib.check(lib.rs_cnn_trunk_infer(critic.data_ptr(), cells.data_ptr(), pcells.data_ptr(), A, a, N, self._wt[a].data_ptr(),
                                                  self._a2.data_ptr(), st), "rs_cnn_trunk_infer")
                y1 = torch.nn.functional.linear(self._a2, m[6].weight, m[6].bias)
                _lib.check(lib.rs_cnn_head(y1.data_ptr(), m[8].weight.data_ptr(), m[8].bias.data_ptr(), m[10].weight.data_ptr(), m[10].bias.data_ptr(), 8,
                                           self._u.data_ptr() + 4 * a, A, self._k_act[a].data_ptr(), self._k_f[a, 0].data_ptr(),
                                           self._act8.data_ptr() + a, A, None, 1, 0, None, N, st), "rs_cnn_head")
            self._values_fused(critic, 1)
        else:
            v_shared = None
            for a, ag in self.agents.items():
                act, logp = ag.act((critic, cells, pcells, a), self._u[:, a])
                if not ag.global_critic or v_shared is None:          # one evaluation serves every owner of a global critic
                    v_shared = ag._values((critic,))
                self._row_act[:, a] = act
                self._row_f[0, :, a] = logp
                self._row_f[1, :, a] = v_shared
                self._act8[:, a] = act.to(torch.int8)
            put(buf.act, self._row_act); put(buf.logp, self._row_f[0]); put(buf.val, self._row_f[1])
            put(buf.obs, self.obs)
        env.step(self._act8)
        _lib.check(lib.rs_collect_post_step(C.byref(cs), 1 if epoch_ended else 0, st), "rs_collect_post_step")
        done, oob_now = self._done_oob[0], self._done_oob[1]           # post_step's copies: the env's own rows are rewritten by the reset
        # bootstrap: ac.step(observations) once more for the envs that time out / are cut (train.py:462-480)
        critic_b, _, _ = self._round_glued(mask8=boot)
        # the heat maps have seen the agents' last positions: from here the env's reset (latency bound, ~110 us) runs on the side stream
        # beside the critic's bootstrap evaluation, the buffer rows and the statistics (which read post_step's copies)
        main = torch.cuda.current_stream(env.device)
        side = side_stream(env.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if epoch_ended:
                env.set_epoch_end()
            env.reset(cut)
        if heads:
            self._values_fused(critic_b, 2, mask8=boot)
            _lib.check(lib.rs_store_rows(ti.data_ptr(), self._k_act.data_ptr(), self._k_f.data_ptr(), self._x_buf.data_ptr(),
                                         self._src_copy[0].data_ptr(), self._src_copy[1].data_ptr(), self._rew_used.data_ptr(), cut.data_ptr(),
                                         boot.data_ptr(), buf.act.data_ptr(), buf.logp.data_ptr(), buf.val.data_ptr(), buf.last_val.data_ptr(),
                                         buf.obs.data_ptr(), buf.source_tar.data_ptr(), buf.rew.data_ptr(), buf.cut.data_ptr(), N, A, self.T, st),
                       "rs_store_rows")
        else:
            put(buf.rew, self._rew_used)
            put(buf.cut, cut.unsqueeze(1).expand(N, A).contiguous())
            bc = boot.view(torch.bool)
            vb = None
            for a, ag in self.agents.items():
                if not ag.global_critic or vb is None:
                    vb = ag._values((critic_b,))
                self._row_f[2, :, a] = torch.where(bc, vb, torch.zeros_like(vb))
            put(buf.last_val, self._row_f[2])
        acc.step_and_episodes(oob_now, done, self.ep_ret, self.steps_in_ep, over.view(torch.bool))
        self.maps.reset(cut)                                             # ac.reset_agent() (train.py:537-540)
        main.wait_stream(side)
        _lib.check(lib.rs_collect_post_reset(C.byref(cs), 1, st), "rs_collect_post_reset")
        if self.predictor is not None:
            self.predictor.reset_kernel(cut)                            # hidden = ac.reset_hidden() (test_cnn/train.py:770)

    @torch.no_grad()
    def _round(self, mask: Optional[torch.Tensor] = None):
        """One select_action round of every owner's MapsBuffer (maps updated once, shared by all owners): returns
        the resident shared maps [N,4,X,Y] and the owners' location / prediction cells [N,A]."""
        pred = None
        if self.predictor is not None:                              # select_action: location_prediction, _ = self.model(obs, hidden)
            pred = self.predictor.predict(self.obs, mask=mask)
        self.maps.update(self.obs, pred=pred, mask=mask)
        return self.maps.shared_maps(), self.maps.field("cell").long(), self.maps.field("pred_cell").long()

    @torch.no_grad()
    def _step(self, epoch_ended: bool) -> None:
        """One lock-step of train.py:332-548 for all envs; row self._t of the buffers is written, then self._t advances."""
        if self.use_glue and getattr(self, "glue", True):
            return self._step_glued(epoch_ended)
        env, buf, L, N, A = self.env, self.buf, self.L, self.N, self.A
        acc, ti = self._acc, self._t
        put = lambda dst, row: dst.index_copy_(0, ti, row.unsqueeze(0))
        critic, cells, pcells = self._round()
        put(self.shared, critic); put(self.cells, cells); put(self.pcells, pcells)
        env.action_uniforms(self._u)
        v_shared = None
        for a, ag in self.agents.items():
            act, logp = ag.act((critic, cells, pcells, a), self._u[:, a])
            if not ag.global_critic or v_shared is None:          # one evaluation serves every owner of a global critic
                v_shared = ag._values((critic,))
            self._row_act[:, a] = act
            self._row_f[0, :, a] = logp
            self._row_f[1, :, a] = v_shared
            self._act8[:, a] = act.to(torch.int8)
        put(buf.act, self._row_act); put(buf.logp, self._row_f[0]); put(buf.val, self._row_f[1])
        put(buf.obs, self.obs)
        next_obs, rew, team, done, info = env.step(self._act8)
        r_used = team.unsqueeze(1).expand(N, A) if self.team_reward else rew
        put(buf.rew, r_used.contiguous())
        self.ep_ret += r_used
        self.steps_in_ep += 1
        terminal = done.bool().any(dim=1)
        oob_now = info["out_of_bounds"]
        timeout = self.steps_in_ep == L
        episode_over = terminal | timeout
        cut = torch.ones_like(episode_over) if epoch_ended else episode_over
        boot = cut if epoch_ended else timeout
        put(buf.cut, cut.unsqueeze(1).to(torch.uint8).expand(N, A).contiguous())
        self.obs.copy_(next_obs)
        # bootstrap: ac.step(observations) once more for the envs that time out / are cut (train.py:462-480);
        # the maps of those envs see the final observation a second time, exactly as in the reference
        bc = boot & cut
        critic_b, _, _ = self._round(mask=bc)
        vb = None
        for a, ag in self.agents.items():
            if not ag.global_critic or vb is None:
                vb = ag._values((critic_b,))
            self._row_f[2, :, a] = torch.where(bc, vb, torch.zeros_like(vb))
        put(buf.last_val, self._row_f[2])
        acc.step_and_episodes(oob_now, done, self.ep_ret, self.steps_in_ep, episode_over)   # one launch (rs_epoch_stats); before the reset
        self.complete_len.copy_(torch.where(episode_over, (ti + 1).expand(N), self.complete_len))
        if epoch_ended:
            env.set_epoch_end()
        self.maps.reset(cut)                                             # ac.reset_agent() (train.py:537-540)
        obs_r, *_ = env.reset(cut)
        self.obs.copy_(obs_r)
        if self.predictor is not None:
            self.predictor.reset(mask=cut)                              # hidden = ac.reset_hidden() (test_cnn/train.py:770)
        self.ep_ret.masked_fill_(cut.unsqueeze(1), 0.0)
        self.steps_in_ep.masked_fill_(cut, 0)
        ti.add_(1)

    @torch.no_grad()
    def collect(self) -> Dict[str, torch.Tensor]:
        if self.obs is None:
            self.start()
        T = self.T
        self._acc.zero_()
        self.complete_len.zero_()
        self._t.zero_()
        if self.use_glue and getattr(self, "glue", True):
            self._glue_state()
            if self.use_heads:
                self._prepare_heads()
        if self.use_graph and self._graph is None and T > 1:
            # lazy library initialisation (rocBLAS handles / workspaces) must not fall into the capture: evaluate the
            # networks once on the current maps (pure functions, no collector state changes), then record the step
            side = side_stream(self.env.device)                               # library warm-up (GEMM handles) outside the capture
            side.wait_stream(torch.cuda.current_stream(self.env.device))
            with torch.cuda.stream(side):
                critic = self.maps.shared_maps()
                cells, pcells = self.maps.field("cell").long(), self.maps.field("pred_cell").long()
                for a, ag in self.agents.items():
                    ag.act((critic, cells, pcells, a), self._u[:, a]); ag._values((critic,))
                if self.predictor is not None:
                    self.predictor._packed()
            torch.cuda.current_stream(self.env.device).wait_stream(side)
            torch.cuda.synchronize(self.env.device)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._step(False)
        for t in range(T - 1):
            if self._graph is not None:
                self._graph.replay()
            else:
                self._step(False)
        self._step(True)
        self.buf.finish(self.agents[0].gamma, self.agents[0].lam)
        return self._acc.result()

    def resume_state(self) -> Dict[str, Any]:
        if self.obs is None:
            self.start()
        return dict(env=self.env.snapshot(), maps=self.maps.snapshot(), steps_in_ep=self.steps_in_ep.clone(), ep_ret=self.ep_ret.clone(),
                    obs=self.obs.clone(), predictor=None if self.predictor is None else self.predictor.resume_state(), epoch=self.epoch)

    def load_resume_state(self, st: Dict[str, Any]) -> None:
        if self.obs is None:
            self.start()
        self.env.restore(st["env"]); self.maps.restore(st["maps"])
        self.steps_in_ep.copy_(st["steps_in_ep"]); self.ep_ret.copy_(st["ep_ret"]); self.obs.copy_(st["obs"])
        self.epoch = int(st.get("epoch", 0))
        if self.predictor is not None and st.get("predictor") is not None:
            self.predictor.load_resume_state(st["predictor"])

    def sync_predictors(self) -> None:
        """Rank 0's predictor cells for every rank (they are created from each process' own torch generator): the per-owner
        PFGRU fills heat-map channel 0 and only rank 0's predictor.pt is saved."""
        if self.predictor is None or _world() == 1:
            return
        for cell in self.predictor.cells:
            for p_ in cell.parameters():
                dist.broadcast(p_.data, src=0)

    def update(self) -> Dict[int, UpdateResult]:
        buf, T, N = self.buf, self.T, self.N
        X, Y = self.maps.map_dimensions
        n_total = N * _world()
        # sample() (ppo.py:754-764): every index below the summed length of the complete episodes of the rank
        tt = torch.arange(T, device=buf.rew.device).view(T, 1)
        valid = tt < self.complete_len.view(1, N)
        w = (valid.float() / (self.complete_len.clamp(min=1).view(1, N).float() * n_total)).reshape(-1)
        shared = self.shared.view(T * N, 4, X, Y)
        cells = self.cells.view(T * N, self.A)
        pcells = self.pcells.view(T * N, self.A)
        out = {}
        env_ids = int(self.env.cfg.env_id_base) + torch.arange(N, device=buf.rew.device, dtype=torch.int64)
        for a, ag in self.agents.items():
            adv = normalize_advantages(buf.adv[:, :, a]).reshape(-1)
            actor_in = lambda lo, hi, a=a: (shared[lo:hi], cells[lo:hi], pcells[lo:hi], a)
            critic_in = lambda lo, hi: (shared[lo:hi],)
            upd_c = (not ag.global_critic) or a == 0
            w_for = None
            if ag.minibatch > 1:                                        # sample() (ppo.py:754-766): a fresh draw per actor iteration
                if int(self.complete_len.min().item()) < ag.minibatch:
                    # the reference fails here too: int(ep_len / minibatch) = 0 indexes -> torch.stack([]) (ppo.py:936)
                    raise ValueError(f"minibatch {ag.minibatch} exceeds the {int(self.complete_len.min().item())} complete-episode samples of an env")
                base = (ag.seed * 4294967296 + env_ids) * 1048583 + self.epoch * 4096 + a * 64
                w_for = lambda kk, base=base, m=ag.minibatch: minibatch_weights(self.complete_len, T, m, base + kk, n_total).reshape(-1)
            out[a] = ag.update_agent(actor_in, critic_in, buf.act[:, :, a].reshape(-1), adv, buf.ret[:, :, a].reshape(-1),
                                     buf.logp[:, :, a].reshape(-1), w, update_critic=upd_c, w_for=w_for)
        self.epoch += 1
        return out
