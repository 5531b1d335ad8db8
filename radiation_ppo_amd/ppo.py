"""PPO side of the hot path, on the same device as the env so rollouts never leave HBM.

Mirrors (paths relative to the reference root):
  FFActorCritic       <- algos/multiagent/NeuralNetworkCores/FF_core.py:42-129 (2x64 tanh MLP actor/critic;
                         same module names, so state_dicts interchange)
  DeviceWelford       <- RADTEAM_core.py:188-277 StatisticStandardization, batched over [N, A]
  RolloutBuffer       <- algos/multiagent/ppo.py:220-502 PPOBuffer (store / GAE / get), time-major [T, N, A]
  VecAgentPPO         <- ppo.py:505-1355 AgentPPO for the 'ff'/'mlp' architecture (loss form of
                         update_rada2c, ppo.py:1206-1256: clipped surrogate - 0.01*MSE + alpha*entropy,
                         per-episode means, KL early stop at 1.5*target_kl, one Adam over actor+critic,
                         StepLR(100, 0.99)); N envs play the role of the reference's N MPI ranks
                         (gradient = mean over ranks of each rank's mean over its episodes).
"""
import ctypes as C
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from .envs import RadSearchVec, gae as rs_gae_call


class FFActorCritic(nn.Module):
    """FF_core.ActorCritic (discrete branch): Linear(11,64)-Tanh-Linear(64,64)-Tanh-Linear(64,8)-Softmax and
    the same-shape critic ending in Linear(64,1)."""

    def __init__(self, state_dim: int = 11, action_dim: int = 8, hidden: int = 64):
        super().__init__()
        self.actor = nn.Sequential(nn.Linear(state_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                   nn.Linear(hidden, action_dim), nn.Softmax(dim=-1))
        self.critic = nn.Sequential(nn.Linear(state_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                    nn.Linear(hidden, 1))

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        for layer in list(self.actor)[:-1]:
            x = layer(x)
        return x

    @torch.no_grad()
    def act(self, x: torch.Tensor, u: torch.Tensor):
        """Sample by inverse CDF with the caller's uniforms u in [0,1): a = #{j : cdf_j <= u} (clamped)."""
        logp_all = torch.log_softmax(self.logits(x), dim=-1)
        cdf = torch.cumsum(logp_all.exp(), dim=-1)
        a = (cdf <= u.unsqueeze(-1)).sum(dim=-1).clamp_(max=logp_all.shape[-1] - 1)
        logp = logp_all.gather(-1, a.unsqueeze(-1)).squeeze(-1)
        v = self.critic(x).squeeze(-1)
        return a, logp, v

    def evaluate(self, x: torch.Tensor, a: torch.Tensor):
        logp_all = torch.log_softmax(self.logits(x), dim=-1)
        logp = logp_all.gather(-1, a.unsqueeze(-1)).squeeze(-1)
        ent = -(logp_all.exp() * logp_all).sum(dim=-1)
        v = self.critic(x).squeeze(-1)
        return logp, v, ent


def mlp_params(seq: nn.Sequential) -> "_lib.RsMlpParams":
    """C-ABI view (rs_mlp_params) of one FF_core Sequential: Linear layers at indices 0, 2, 4."""
    t = [seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, seq[4].weight, seq[4].bias]
    for x in t:
        assert x.dtype == torch.float32 and x.is_contiguous() and x.is_cuda
    return _lib.RsMlpParams(*[x.data_ptr() for x in t])


def policy_forward(ac: "FFActorCritic", x: torch.Tensor, want_logits: bool = True, want_value: bool = True):
    """Actor logits [M,8] and critic value [M] on the matrix cores (rs_policy_forward)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == _lib.RS_OBS_DIM
    M = x.numel() // _lib.RS_OBS_DIM
    logits = torch.empty(M, 8, dtype=torch.float32, device=x.device) if want_logits else None
    value = torch.empty(M, dtype=torch.float32, device=x.device) if want_value else None
    pa, pc = mlp_params(ac.actor), mlp_params(ac.critic)
    import ctypes as C
    _lib.check(lib.rs_policy_forward(C.byref(pa), C.byref(pc), x.data_ptr(), M,
                                     None if logits is None else logits.data_ptr(),
                                     None if value is None else value.data_ptr(),
                                     torch.cuda.current_stream(x.device).cuda_stream), "rs_policy_forward")
    return logits, value


N_PARAMS = 10441          # FF_core actor 5448 + critic 4993


class FusedPPOGrad:
    """rs_ppo_grad: loss statistics and all parameter gradients of the FF_core actor+critic in one pass over
    the batch on the matrix cores (replaces loss.backward(), ppo.py:1253-1254)."""

    def __init__(self, ac: "FFActorCritic"):
        import ctypes as C
        self.lib = _lib.load()
        self.ac = ac
        dev = next(ac.parameters()).device
        # one flat float32 bucket = all gradients (the data-parallel exchange reduces it in ONE collective); the five
        # loss statistics stay float64 end to end so the KL early-stop decision is bit-identical for 1 and N ranks
        self.bucket = torch.zeros(N_PARAMS + 16, dtype=torch.float32, device=dev)     # RS_PPO_GRAD_FLOATS: gradients + (hi, lo) statistics
        self.grads = self.bucket[:N_PARAMS]
        self.stats = torch.zeros(5, dtype=torch.float64, device=dev)
        self.stats_from_bucket = False
        self.ws = torch.empty(self.lib.rs_ppo_grad_workspace_bytes() + 256, dtype=torch.uint8, device=dev)
        self._ws_ptr = self.ws.data_ptr() + (-self.ws.data_ptr()) % 256
        # views of the flat gradient in the parameter order of the C ABI
        order = [ac.actor[0].weight, ac.actor[0].bias, ac.actor[2].weight, ac.actor[2].bias, ac.actor[4].weight, ac.actor[4].bias,
                 ac.critic[0].weight, ac.critic[0].bias, ac.critic[2].weight, ac.critic[2].bias, ac.critic[4].weight, ac.critic[4].bias]
        self.m = torch.zeros(N_PARAMS, dtype=torch.float32, device=dev)       # Adam exp_avg
        self.v = torch.zeros(N_PARAMS, dtype=torch.float32, device=dev)       # Adam exp_avg_sq
        self.state = torch.zeros(8, dtype=torch.float64, device=dev)          # rs_update_state (56 bytes used)
        self.state_i32 = self.state.view(torch.int32)                         # [adam_step, stopped, iters, pad, ...]
        self.views = []
        o = 0
        for p in order:
            self.views.append((p, self.grads[o:o + p.numel()].view_as(p)))
            o += p.numel()
        assert o == N_PARAMS

    def begin_update(self) -> None:
        self.state_i32[1:3].zero_()                                           # stopped = 0, iters = 0

    def adam_step(self, lr: float, kl_threshold: float) -> None:
        import ctypes as C
        pa, pc = mlp_params(self.ac.actor), mlp_params(self.ac.critic)
        _lib.check(self.lib.rs_adam_step(C.byref(pa), C.byref(pc), self.bucket.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                         None if self.stats_from_bucket else self.stats.data_ptr(), self.state.data_ptr(), lr, kl_threshold,
                                         torch.cuda.current_stream(self.grads.device).cuda_stream), "rs_adam_step")

    def read_state(self):
        """(iters, stopped, adam_step, last_stats[5]) -- one host sync."""
        st = self.state.cpu()
        i32 = st.view(torch.int32)
        return int(i32[2]), int(i32[1]), int(i32[0]), st[2:7].tolist()

    def __call__(self, X, act, adv, ret, logp_old, w, clip_ratio: float, alpha: float, vf_coef: float = 0.01,
                 use_stop_flag: bool = False):
        import ctypes as C
        for t in (X, adv, ret, logp_old, w):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert act.dtype == torch.int64 and act.is_contiguous()
        self.stats_from_bucket = False                 # set by allreduce(): the reduced statistics then live in the bucket's tail
        b = _lib.RsPpoBatch(X.data_ptr(), act.data_ptr(), adv.data_ptr(), ret.data_ptr(), logp_old.data_ptr(), w.data_ptr(),
                            X.shape[0], clip_ratio, alpha, vf_coef)
        pa, pc = mlp_params(self.ac.actor), mlp_params(self.ac.critic)
        with _lib.timed("rs_ppo_grad"):
            _lib.check(self.lib.rs_ppo_grad(C.byref(pa), C.byref(pc), C.byref(b), self.bucket.data_ptr(), self.stats.data_ptr(),
                                            self._ws_ptr, (self.state.data_ptr() + 4) if use_stop_flag else None,
                                            torch.cuda.current_stream(X.device).cuda_stream), "rs_ppo_grad")
        return self.stats, self.grads

    def allreduce(self) -> None:
        """mpi_avg_grads (ppo.py:1256) + mpi_avg(kl) (:1250) as ONE RCCL all-reduce per Adam step: the bucket holds the
        gradients and, behind them, the five statistics as float32 (hi, lo) pairs written by the reduce kernel (xGMI
        all-reduces of this size are latency bound: the count matters, not the bytes).  rs_adam_step then takes the
        statistics from the bucket.  The loss weights already carry 1/(global env count): SUM over ranks = the reference's
        average.  After the KL early stop rs_ppo_grad publishes zeros, so the remaining (no-op) iterations reduce zeros."""
        Collectives.all_reduce_sum(self.bucket)
        self.stats_from_bucket = True

    def assign_grads(self) -> None:
        for p, g in self.views:
            p.grad = g


class DeviceWelford:
    """StatisticStandardization (RADTEAM_core.py:188-277) for every (env, agent) at once, float64."""

    def __init__(self, shape, device, impl: Optional[str] = None):
        """impl: "hip" (default on a cuda device) or "torch" (the element-wise composition; the only form for CPU tensors)."""
        self.impl = impl or ("hip" if torch.device(device).type == "cuda" else "torch")
        assert self.impl in ("hip", "torch") and (self.impl == "torch" or torch.device(device).type == "cuda")
        self.count = torch.zeros(shape, dtype=torch.float64, device=device)
        self.mean = torch.zeros(shape, dtype=torch.float64, device=device)
        self.sq = torch.zeros(shape, dtype=torch.float64, device=device)
        self.std = torch.ones(shape, dtype=torch.float64, device=device)

    # On the GPU the three operations are one launch each (csrc/rs_welford.hip: the same float64 arithmetic, bit for bit); the
    # element-wise composition below is what the kernels are tested against and what CPU tensors (host-logic tests) use.
    @staticmethod
    def _strided(t: torch.Tensor):
        """(tensor, element stride) for a [N, A] view whose flattened (n, a) index has one stride (obs[..., 0] of [N, A, 11])."""
        if t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) > 0 and t.stride(0) == t.shape[1] * t.stride(1):
            return t, t.stride(1)
        if t.dim() == 1 and t.dtype == torch.float32 and t.stride(0) > 0:
            return t, t.stride(0)
        t = t.float().contiguous()
        return t, 1

    def _hip(self, reading: Optional[torch.Tensor] = None) -> bool:
        return self.impl == "hip" and self.count.dim() in (1, 2) and (reading is None or reading.shape == self.count.shape)

    def _na(self):
        return (self.count.shape[0], self.count.shape[1] if self.count.dim() == 2 else 1)

    @staticmethod
    def _m8(mask: Optional[torch.Tensor]):
        if mask is None:
            return None
        m = mask.contiguous()
        return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)

    def update(self, reading: torch.Tensor, mask: Optional[torch.Tensor] = None) -> None:
        if self._hip(reading):
            r, st = self._strided(reading)
            m8 = self._m8(mask)
            n, a = self._na()
            _lib.check(_lib.load().rs_welford_update(self.count.data_ptr(), self.mean.data_ptr(), self.sq.data_ptr(), self.std.data_ptr(),
                                                     r.data_ptr(), st, None if m8 is None else m8.data_ptr(), n, a,
                                                     C.c_void_p(torch.cuda.current_stream(self.count.device).cuda_stream)), "rs_welford_update")
            return
        x = reading.double()
        count = self.count + 1
        first = count == 1
        mean_new = torch.where(first, x, self.mean + (x - self.mean) / count)
        sq_new = torch.where(first, self.sq, self.sq + (x - self.mean) * (x - mean_new))
        std_new = torch.where(first, self.std, torch.clamp(torch.sqrt(sq_new / torch.clamp(count - 1, min=1)), min=1.0))
        if mask is not None:
            m = mask.view(-1, *([1] * (x.dim() - 1))).expand_as(x)
            count = torch.where(m, count, self.count)
            mean_new = torch.where(m, mean_new, self.mean)
            sq_new = torch.where(m, sq_new, self.sq)
            std_new = torch.where(m, std_new, self.std)
        # in place: a captured collector step (HIP graph) refers to these tensors by address
        self.count.copy_(count); self.mean.copy_(mean_new); self.sq.copy_(sq_new); self.std.copy_(std_new)

    def standardize(self, reading: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out: an optional float32 destination view of the same shape (e.g. x[..., 0] of a cloned observation)."""
        if self._hip(reading):
            r, st = self._strided(reading)
            if out is None:
                out = torch.empty(self.count.shape, dtype=torch.float32, device=self.count.device)
            o, ost = self._strided(out)
            assert o.data_ptr() == out.data_ptr(), "standardize(out=...) needs a float32 view with one element stride"
            _lib.check(_lib.load().rs_welford_standardize(self.mean.data_ptr(), self.std.data_ptr(), r.data_ptr(), st, o.data_ptr(), ost,
                                                          self.count.numel(),
                                                          C.c_void_p(torch.cuda.current_stream(self.count.device).cuda_stream)),
                       "rs_welford_standardize")
            return out
        z = ((reading.double() - self.mean) / self.std).float()
        if out is not None:
            out.copy_(z)
            return out
        return z

    def reset(self, mask: torch.Tensor) -> None:
        if self._hip():
            n, a = self._na()
            m8 = self._m8(mask)
            _lib.check(_lib.load().rs_welford_reset(self.count.data_ptr(), self.mean.data_ptr(), self.sq.data_ptr(), self.std.data_ptr(),
                                                    m8.data_ptr(), n, a, C.c_void_p(torch.cuda.current_stream(self.count.device).cuda_stream)),
                       "rs_welford_reset")
            return
        m = mask.view(-1, *([1] * (self.count.dim() - 1))).expand_as(self.count)
        self.count.masked_fill_(m, 0.0); self.mean.masked_fill_(m, 0.0); self.sq.masked_fill_(m, 0.0); self.std.masked_fill_(m, 1.0)


class RolloutBuffer:
    """PPOBuffer (ppo.py:220-502) for all envs and agents, time-major so that one lock-step is one
    contiguous row: obs [T,N,A,11], act/rew/val/logp/last_val [T,N,A], cut [T,N,A] u8, source_tar [T,N,2]."""

    def __init__(self, T: int, N: int, A: int, obs_dim: int, device):
        f = dict(dtype=torch.float32, device=device)
        self.T, self.N, self.A = T, N, A
        self.obs = torch.zeros(T, N, A, obs_dim, **f)
        self.act = torch.zeros(T, N, A, dtype=torch.int64, device=device)
        self.rew = torch.zeros(T, N, A, **f)
        self.val = torch.zeros(T, N, A, **f)
        self.logp = torch.zeros(T, N, A, **f)
        self.last_val = torch.zeros(T, N, A, **f)
        self.cut = torch.zeros(T, N, A, dtype=torch.uint8, device=device)
        self.source_tar = torch.zeros(T, N, 2, **f)
        self.adv = torch.zeros(T, N, A, **f)
        self.ret = torch.zeros(T, N, A, **f)

    def finish(self, gamma: float, lam: float) -> None:
        """GAE_advantage_and_rewardsToGO for every trajectory slice of the buffer in one kernel (rs_gae)."""
        rs_gae_call(self.rew, self.val, self.cut, self.last_val, gamma, lam, adv=self.adv, ret=self.ret)

    def episode_weights(self) -> torch.Tensor:
        """1 / (episodes in the column * length of the sample's episode): the per-sample weight that turns
        sums into the reference's mean-over-episodes of per-episode means (ppo.py:1191-1234)."""
        cut = self.cut[:, :, 0].long()                     # cuts are env-wide
        T, N = cut.shape
        seg = torch.cumsum(cut, dim=0) - cut               # episode index of every step, per env
        n_ep = seg[-1] + 1                                  # the last step always closes a trajectory
        lens = torch.zeros(T, N, dtype=torch.float32, device=cut.device)
        lens.scatter_add_(0, seg, torch.ones(T, N, dtype=torch.float32, device=cut.device))
        w = 1.0 / (n_ep.unsqueeze(0).float() * lens.gather(0, seg))
        return w                                            # [T, N]


@dataclass
class UpdateResult:
    """ppo.py:146-157"""
    stop_iteration: int
    loss_policy: float
    loss_critic: float
    loss_predictor: float
    kl_divergence: float
    Entropy: float
    ClipFrac: float
    LocLoss: float
    VarExplain: int = 0


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class Collectives:
    """Every data-path collective of the update loops goes through here: `count` is what the 2-rank tests read to hold the paths to
    ONE all-reduce per optimiser step (at 42-355 KB an xGMI ring all-reduce is latency bound, so the number of
    collectives matters, not their bytes)."""
    count = 0

    @staticmethod
    def all_reduce_sum(t: torch.Tensor) -> None:
        Collectives.count += 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def reduce_grads_and_stats(params, stats: torch.Tensor) -> torch.Tensor:
    """mpi_avg_grads (mpi_pytorch.py:26-33) + mpi_avg of the loss statistics (ppo.py:1250) as ONE flattened all-reduce: the bucket is
    every gradient followed by the float64 statistics as float32 (hi, lo) pairs (~1e-14 relative loss against a float64 sum), as the
    MLP path's device bucket (FusedPPOGrad.allreduce).  The loss weights already carry 1 / (global env count): SUM = the reference's
    average.  The gradients are written back in place; returns the reduced statistics (float64)."""
    if _world() == 1:
        return stats
    ps = [p for p in params if p.grad is not None]
    hi = stats.float()
    lo = (stats - hi.double()).float()
    flat = torch.cat([p.grad.reshape(-1) for p in ps] + [hi.reshape(-1), lo.reshape(-1)])
    Collectives.all_reduce_sum(flat)
    o = 0
    for p in ps:
        p.grad.copy_(flat[o:o + p.numel()].view_as(p))
        o += p.numel()
    k = stats.numel()
    return (flat[o:o + k].double() + flat[o + k:o + 2 * k].double()).view_as(stats)


def host_read(t: torch.Tensor) -> List[float]:
    """t.tolist() for a small device tensor in a loop that reads once per iteration (the KL decision of a policy iteration).
    The copy goes to pinned memory and the host POLLS its event instead of blocking in the runtime's stream wait: a blocked wait
    puts the thread to sleep, and on a busy host the wake-up alone was worth several policy iterations' kernels (RAD-A2C policy loop:
    46 ms per iteration on such boxes against 15 ms; bench.py reports the split as update_split_ms)."""
    if not t.is_cuda:
        return t.tolist()
    key = (t.device, t.dtype, t.numel())
    buf = _HOST_READ.get(key)
    if buf is None:
        buf = _HOST_READ[key] = (torch.empty(t.numel(), dtype=t.dtype, pin_memory=True), torch.cuda.Event())
    host, ev = buf
    host.copy_(t.reshape(-1), non_blocking=True)
    ev.record(torch.cuda.current_stream(t.device))
    while not ev.query():
        pass
    return host.tolist()


_HOST_READ: Dict[Any, Any] = {}
_SIDE: Dict[Any, Any] = {}


def side_stream(device) -> "torch.cuda.Stream":
    """THE side stream of a device, shared by everything that overlaps work with the main stream (the collectors' env reset beside the
    bootstrap round, update_model's draws beside the backward walk, the policy loop's K11 passes beside the GRU chain -- never at the
    same time).  One, not one per user: HIP maps streams onto a few hardware queues, and with three side streams alive the policy
    loop's pass stream shared the main stream's queue -- the K11 pass and the GRU chain it is meant to hide ran back to back
    (RAD-A2C policy loop 535 -> 670 ms, measured)."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


# Every key of the reference's ppo_kwargs (algos/multiagent/main.py:574-596; AgentPPO's fields, ppo.py:505-600).  A constructor of this
# build names the ones that configure its path; of the rest it accepts exactly those listed as `no_effect` there (each with the reason
# in its docstring) and raises on anything else -- a misspelt or unsupported option must not vanish into **kwargs.
REFERENCE_PPO_KWARGS = frozenset({
    "observation_space", "bp_args", "steps_per_epoch", "steps_per_episode", "number_of_agents", "env_height", "actor_critic_args",
    "actor_critic_architecture", "minibatch", "train_pi_iters", "train_v_iters", "train_pfgru_iters", "actor_learning_rate",
    "critic_learning_rate", "pfgru_learning_rate", "gamma", "alpha", "clip_ratio", "target_kl", "lam", "GlobalCriticOptimizer"})


def reject_unknown_kwargs(where: str, extra: Dict[str, Any], no_effect) -> None:
    """`extra`: what a constructor received beyond its named parameters."""
    unknown = sorted(set(extra) - set(no_effect))
    if unknown:
        raise TypeError(f"{where}: unexpected keyword argument(s) {unknown}; accepted without effect on this path: {sorted(no_effect)}")


def check_minibatch(minibatch: Any) -> int:
    if not isinstance(minibatch, int) or isinstance(minibatch, bool) or minibatch < 1:
        raise ValueError(f"minibatch must be a positive int (ppo.py:580), got {minibatch!r}")
    return minibatch


class VecAgentPPO:
    """One agent id's networks + optimiser; the vectorised counterpart of AgentPPO (ppo.py:505-1355).

    `minibatch` (ppo.py:580) is accepted and validated but changes nothing here, exactly as in the reference: this path is the
    update_rada2c loss form, and update_rada2c assigns `minibatch` (ppo.py:1159-1160) and never reads it -- every episode of the epoch
    enters every iteration (`ep_select` takes `min_iterations = len(ep_form)` episodes, :1183).  Keys of the reference's ppo_kwargs
    without a counterpart on this path (_NO_EFFECT: the PFGRU's and the CNN critic's settings, `actor_critic_args` of the GRU core,
    `train_v_iters` / `critic_learning_rate` -- one optimiser trains actor and critic, ppo.py:1256-1258) are accepted; others raise."""

    _NO_EFFECT = ("bp_args", "env_height", "actor_critic_args", "train_pfgru_iters", "pfgru_learning_rate", "seed")

    def __init__(self, id: int, observation_space: int = 11, action_space: int = 8, steps_per_epoch: int = 480,
                 steps_per_episode: int = 120, number_of_agents: int = 1, actor_critic_architecture: str = "ff",
                 train_pi_iters: int = 40, train_v_iters: int = 40, actor_learning_rate: float = 3e-4,
                 critic_learning_rate: float = 1e-3, gamma: float = 0.99, alpha: float = 0.0, clip_ratio: float = 0.2,
                 target_kl: float = 0.07, lam: float = 0.9, minibatch: int = 1, GlobalCriticOptimizer=None, device="cuda:0",
                 **other: Any):
        if actor_critic_architecture not in ("ff", "mlp"):
            raise ValueError("Unsupported Neural Network type requested")   # ppo.py:666-667
        if GlobalCriticOptimizer is not None:
            raise Exception("No global critic option for RAD-A2C")          # ppo.py:651-652
        reject_unknown_kwargs("VecAgentPPO", other, self._NO_EFFECT)
        self.minibatch = check_minibatch(minibatch)
        self.id = id
        self.device = torch.device(device)
        self.gamma, self.lam, self.alpha = gamma, lam, alpha
        self.clip_ratio, self.target_kl = clip_ratio, target_kl
        self.train_pi_iters = train_pi_iters
        self.actor_learning_rate = actor_learning_rate
        self.epochs_done = 0
        self.agent = FFActorCritic(observation_space, action_space).to(self.device)
        self.pi_optimizer = torch.optim.Adam(self.agent.parameters(), lr=actor_learning_rate)
        self.pi_scheduler = torch.optim.lr_scheduler.StepLR(self.pi_optimizer, step_size=100, gamma=0.99)   # ppo.py:205-207
        self._flat_grad: Optional[torch.Tensor] = None
        self.fused_update = True            # rs_ppo_grad on the GPU; the autograd path remains for CPU / A-B checks
        self._fused: Optional[FusedPPOGrad] = None

    def sync_params(self) -> None:
        """mpi_pytorch.sync_params (mpi_pytorch.py:43-49): one RCCL broadcast of the flattened parameters."""
        if _world() > 1:
            flat = torch.cat([p.data.view(-1) for p in self.agent.parameters()])
            dist.broadcast(flat, src=0)
            o = 0
            for p in self.agent.parameters():
                p.data.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()

    def update_agent(self, X: torch.Tensor, act: torch.Tensor, adv: torch.Tensor, ret: torch.Tensor,
                     logp_old: torch.Tensor, w: torch.Tensor) -> UpdateResult:
        """update_agent (ppo.py:746-813) + update_rada2c (:1150-1281) on the whole batch.  w sums to 1 over
        the GLOBAL batch (all ranks)."""
        kk = 0
        kl_reached = False
        last = None
        thr = 1.5 * self.target_kl
        if self.fused_update and X.is_cuda:
            return self._update_agent_fused(X, act, adv, ret, logp_old, w)
        while not kl_reached and kk < self.train_pi_iters:
            logp, v, ent = self.agent.evaluate(X, act)
            ratio = torch.exp(logp - logp_old)
            clip_adv = torch.clamp(ratio, 1 - self.clip_ratio, 1 + self.clip_ratio) * adv
            surr = torch.min(ratio * adv, clip_adv)
            val_loss = (w * (v - ret) ** 2).sum()
            ent_m = (w * ent).sum()
            # the entropy bonus is a detached scalar in the reference (`.detach().mean().item()`, ppo.py:1216)
            loss = -((w * surr).sum() - 0.01 * val_loss + self.alpha * ent_m.detach())
            with torch.no_grad():
                clipped = (ratio > 1 + self.clip_ratio) | (ratio < 1 - self.clip_ratio)
                stats = torch.stack([(w * (logp_old - logp)).sum(), ent_m.detach(), (w * clipped.float()).sum(),
                                     val_loss.detach(), loss.detach()]).double()
            self.pi_optimizer.zero_grad(set_to_none=True)
            loss.backward()
            # mpi_avg(kl) (ppo.py:1250) + mpi_avg_grads (:1256): one collective, the statistics behind the gradients
            stats_h = host_read(reduce_grads_and_stats(self.agent.parameters(), stats))   # the early-stop decision needs the host
            last = stats_h
            if stats_h[0] < thr:
                self.pi_optimizer.step()
            else:
                kl_reached = True
            kk += 1
        self.pi_scheduler.step()                                       # ppo.py:799
        return UpdateResult(stop_iteration=kk, loss_policy=last[4], loss_critic=last[3], loss_predictor=0.0,
                            kl_divergence=last[0], Entropy=last[1], ClipFrac=last[2], LocLoss=0.0)

    def _update_agent_fused(self, X, act, adv, ret, logp_old, w) -> UpdateResult:
        """update_agent / update_rada2c control flow (ppo.py:789-796,1250-1261) with no host round trip inside
        the loop: rs_ppo_grad computes statistics + gradients, rs_adam_step takes the KL early-stop decision and
        the Adam step on the device; iterations after the stop are no-ops.  One sync at the end."""
        if self._fused is None:
            self._fused = FusedPPOGrad(self.agent)
        f = self._fused
        lr = self.actor_learning_rate * (0.99 ** (self.epochs_done // 100))   # StepLR(100, 0.99) (ppo.py:205-207)
        thr = 1.5 * self.target_kl
        f.begin_update()
        for _ in range(self.train_pi_iters):
            stats, grads = f(X, act, adv, ret, logp_old, w, self.clip_ratio, self.alpha, use_stop_flag=True)
            if _world() > 1:
                f.allreduce()
            f.adam_step(lr, thr)
        kk, stopped, _, last = f.read_state()
        self.epochs_done += 1
        return UpdateResult(stop_iteration=kk, loss_policy=last[4], loss_critic=last[3], loss_predictor=0.0,
                            kl_divergence=last[0], Entropy=last[1], ClipFrac=last[2], LocLoss=0.0)

    def save(self, path: str) -> None:
        """FF_core.PPO.save (FF_core.py:257-258): the network's state_dict (keys actor.N.*, critic.N.*)."""
        torch.save(self.agent.state_dict(), path)

    def load(self, path: str) -> None:
        self.agent.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))

    def resume_state(self) -> Dict[str, Any]:
        """Everything a resumed run needs beyond the weights (absent in the reference): Adam moments and step
        count of the device-side optimiser (or the torch optimiser of the autograd path) and the LR-schedule position."""
        st: Dict[str, Any] = dict(model=self.agent.state_dict(), epochs_done=self.epochs_done,
                                  pi_optimizer=self.pi_optimizer.state_dict(), pi_scheduler=self.pi_scheduler.state_dict())
        if self._fused is not None:
            f = self._fused
            st["fused"] = dict(m=f.m.clone(), v=f.v.clone(), adam_step=int(f.state_i32[0].item()))
        return st

    def load_resume_state(self, st: Dict[str, Any]) -> None:
        self.agent.load_state_dict(st["model"])
        self.epochs_done = int(st["epochs_done"])
        self.pi_optimizer.load_state_dict(st["pi_optimizer"])
        self.pi_scheduler.load_state_dict(st["pi_scheduler"])
        if "fused" in st:
            if self._fused is None:
                self._fused = FusedPPOGrad(self.agent)
            f = self._fused
            f.m.copy_(st["fused"]["m"]); f.v.copy_(st["fused"]["v"])
            f.state_i32[0] = int(st["fused"]["adam_step"])


def normalize_advantages(adv: torch.Tensor) -> torch.Tensor:
    """PPOBuffer.get advantage normalisation (ppo.py:445-446) with mpi_statistics_scalar (mpi_tools.py:71-95):
    global mean and POPULATION std over every rank's samples, no epsilon."""
    n = torch.tensor([adv.numel()], dtype=torch.float64, device=adv.device)
    s = adv.double().sum().view(1)
    if _world() > 1:
        pack = torch.cat([s, n])
        Collectives.all_reduce_sum(pack)
        s, n = pack[0:1], pack[1:2]
    mean = (s / n).float()
    sq = ((adv - mean) ** 2).double().sum().view(1)
    if _world() > 1:
        Collectives.all_reduce_sum(sq)
    std = torch.sqrt(sq / n).float()
    return (adv - mean) / std


class EpochStats:
    """What train() hands its loggers per epoch and agent id (train.py:386-398, :494-501, :519-526), accumulated on the
    device over all envs: out-of-bounds and terminal counters, and n / sum / sum of squares / max / min of the returns
    and the lengths of the episodes that ended."""

    def __init__(self, A: int, device):
        f64 = dict(dtype=torch.float64, device=device)
        self.oob = torch.zeros(A, **f64)
        self.done = torch.zeros(A, **f64)
        self.ep_cnt = torch.zeros((), **f64)
        self.ep_len = torch.zeros((), **f64)
        self.ret_sum = torch.zeros(A, **f64)
        self.ret_sq = torch.zeros(A, **f64)
        self.ret_max = torch.full((A,), float("-inf"), **f64)
        self.ret_min = torch.full((A,), float("inf"), **f64)

    def zero_(self) -> None:
        """Start a new epoch in place (the tensors keep their addresses: a captured collector step refers to them)."""
        for t in (self.oob, self.done, self.ep_cnt, self.ep_len, self.ret_sum, self.ret_sq):
            t.zero_()
        self.ret_max.fill_(float("-inf")); self.ret_min.fill_(float("inf"))

    def step(self, out_of_bounds: torch.Tensor, done: torch.Tensor) -> None:
        self.oob += out_of_bounds.double().sum(dim=0)                        # [N,A] -> per agent id
        self.done += done.double().sum(dim=0)                                # terminals[id] (the env latch as agent id saw it)

    def episodes(self, ep_ret: torch.Tensor, steps_in_ep: torch.Tensor, over: torch.Tensor) -> None:
        m = over.unsqueeze(1)
        r = ep_ret.double()
        self.ret_sum += (r * m).sum(dim=0)
        self.ret_sq += (r * r * m).sum(dim=0)
        self.ret_max.copy_(torch.maximum(self.ret_max, torch.where(m, r, torch.full_like(r, float("-inf"))).max(dim=0).values))
        self.ret_min.copy_(torch.minimum(self.ret_min, torch.where(m, r, torch.full_like(r, float("inf"))).min(dim=0).values))
        self.ep_len += (steps_in_ep.double() * over).sum()
        self.ep_cnt += over.double().sum()

    def step_and_episodes(self, out_of_bounds: torch.Tensor, done: torch.Tensor, ep_ret: torch.Tensor, steps_in_ep: torch.Tensor,
                          over: torch.Tensor) -> None:
        """step() + episodes() of one lock-step; on the GPU one launch (rs_epoch_stats) instead of ~25 reductions."""
        ok = (self.oob.is_cuda and out_of_bounds.dtype == torch.uint8 and done.dtype == torch.uint8 and ep_ret.dtype == torch.float32
              and steps_in_ep.dtype == torch.int32 and over.dtype == torch.bool
              and all(t.is_contiguous() for t in (out_of_bounds, done, ep_ret, steps_in_ep, over)))
        if not ok:
            self.step(out_of_bounds, done)
            self.episodes(ep_ret, steps_in_ep, over)
            return
        N, A = ep_ret.shape
        _lib.check(_lib.load().rs_epoch_stats(out_of_bounds.data_ptr(), done.data_ptr(), ep_ret.data_ptr(), steps_in_ep.data_ptr(),
                                              over.view(torch.uint8).data_ptr(), self.oob.data_ptr(), self.done.data_ptr(),
                                              self.ep_cnt.data_ptr(), self.ep_len.data_ptr(), self.ret_sum.data_ptr(), self.ret_sq.data_ptr(),
                                              self.ret_max.data_ptr(), self.ret_min.data_ptr(), N, A,
                                              C.c_void_p(torch.cuda.current_stream(self.oob.device).cuda_stream)), "rs_epoch_stats")

    def result(self) -> Dict[str, torch.Tensor]:
        return dict(DoneCount=self.done.clone(), OutOfBound=self.oob.clone(), EpCount=self.ep_cnt.clone(), EpLenSum=self.ep_len.clone(),
                    EpRetSum=self.ret_sum.clone(), EpRetSqSum=self.ret_sq.clone(), EpRetMax=self.ret_max.clone(), EpRetMin=self.ret_min.clone())


def _welford_state(w: "DeviceWelford"):
    return dict(count=w.count.clone(), mean=w.mean.clone(), sq=w.sq.clone(), std=w.std.clone())


def _welford_load(w: "DeviceWelford", st) -> None:
    w.count.copy_(st["count"]); w.mean.copy_(st["mean"]); w.sq.copy_(st["sq"]); w.std.copy_(st["std"])


class Collector:
    """The epoch loop body of train_PPO.train (train.py:332-548) for N envs at once: policy forward,
    Philox inverse-CDF sampling, env lock-step, buffer write, episode/epoch cut logic, bootstrap values,
    masked reset.  Nothing returns to the host inside the loop."""

    def __init__(self, env: RadSearchVec, agents: Dict[int, VecAgentPPO], steps_per_epoch: int, steps_per_episode: int,
                 global_critic_flag: bool = False, standardize: bool = True):
        self.env, self.agents = env, agents
        self.T, self.L = steps_per_epoch, steps_per_episode
        self.N, self.A = env.num_envs, env.number_agents
        self.team_reward = global_critic_flag
        self.standardize = standardize
        dev = env.device
        self.buf = RolloutBuffer(self.T, self.N, self.A, _lib.RS_OBS_DIM, dev)
        self.stat = DeviceWelford((self.N, self.A), dev)
        self.steps_in_ep = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.ep_ret = torch.zeros(self.N, self.A, dtype=torch.float32, device=dev)
        self._u = torch.empty(self.N, self.A, dtype=torch.float32, device=dev)
        self._act8 = torch.empty(self.N, self.A, dtype=torch.int8, device=dev)
        # epoch statistics (logger columns of train.py:605-627)
        self.stats: Dict[str, torch.Tensor] = {}
        self.obs = None
        self.started = False

    def _x(self, obs: torch.Tensor) -> torch.Tensor:
        if not self.standardize:
            return obs
        x = obs.clone()
        self.stat.standardize(obs[..., 0], out=x[..., 0])
        return x

    def start(self) -> None:
        """train.py:273-312: first reset + first Welford update."""
        obs, *_ = self.env.reset()
        self.obs = obs.clone()
        self.stat.update(self.obs[..., 0])
        self.started = True

    @torch.no_grad()
    def collect(self) -> Dict[str, torch.Tensor]:
        if not self.started:
            self.start()
        env, buf, T, L, N, A = self.env, self.buf, self.T, self.L, self.N, self.A
        dev = env.device
        acc = EpochStats(A, dev)
        for t in range(T):
            x = self._x(self.obs)                                            # train.py:334-341
            env.action_uniforms(self._u)
            for a, ag in self.agents.items():                                # train.py:345-351
                act, logp, v = ag.agent.act(x[:, a], self._u[:, a])
                buf.act[t, :, a] = act
                buf.logp[t, :, a] = logp
                buf.val[t, :, a] = v
                self._act8[:, a] = act.to(torch.int8)
            buf.obs[t] = x
            buf.source_tar[t, :, 0] = env.state("src_x")[0].float()
            buf.source_tar[t, :, 1] = env.state("src_y")[0].float()
            next_obs, rew, team, done, info = env.step(self._act8)           # train.py:361-363
            r_used = team.unsqueeze(1).expand(N, A) if self.team_reward else rew
            buf.rew[t] = r_used
            self.ep_ret += r_used
            self.steps_in_ep += 1
            terminal = done.bool().any(dim=1)                                       # train.py:387-391
            acc.step(info["out_of_bounds"], done)
            timeout = self.steps_in_ep == L                                  # train.py:394-405
            episode_over = terminal | timeout
            epoch_ended = t == T - 1
            cut = episode_over | epoch_ended
            buf.cut[t] = cut.unsqueeze(1).to(torch.uint8).expand(N, A)
            self.stat.update(next_obs[..., 0])                               # train.py:432-436
            self.obs = next_obs.clone()
            # bootstrap value on timeout / epoch cut, 0 on a pure terminal (train.py:462-487)
            boot = timeout | epoch_ended
            xb = self._x(self.obs)
            for a, ag in self.agents.items():
                vb = ag.agent.critic(xb[:, a]).squeeze(-1)
                buf.last_val[t, :, a] = torch.where((boot & cut).bool(), vb, torch.zeros_like(vb))
            acc.episodes(self.ep_ret, self.steps_in_ep, episode_over)       # train.py:494-501
            if epoch_ended:
                env.set_epoch_end()                                          # train.py:482-484
            self.stat.reset(cut)                                             # train.py:504-509
            obs_r, *_ = env.reset(cut)                                       # train.py:530
            self.obs = obs_r.clone()
            self.ep_ret = torch.where(cut.unsqueeze(1), torch.zeros_like(self.ep_ret), self.ep_ret)
            self.steps_in_ep = torch.where(cut, torch.zeros_like(self.steps_in_ep), self.steps_in_ep)
            self.stat.update(self.obs[..., 0], mask=cut)                     # train.py:542-548
        buf.finish(self.agents[0].gamma, self.agents[0].lam)
        return acc.result()

    def update(self) -> Dict[int, UpdateResult]:
        return ppo_update_from_buffer(self)

    # what a run carries from one epoch into the next (train_PPO.save_resume / load): the env and the running episode state
    def resume_state(self) -> Dict[str, Any]:
        if not self.started:
            self.start()
        return dict(env=self.env.snapshot(), stat=_welford_state(self.stat), steps_in_ep=self.steps_in_ep.clone(),
                    ep_ret=self.ep_ret.clone(), obs=self.obs.clone())

    def load_resume_state(self, st: Dict[str, Any]) -> None:
        if not self.started:
            self.start()
        self.env.restore(st["env"])
        _welford_load(self.stat, st["stat"])
        self.steps_in_ep, self.ep_ret, self.obs = st["steps_in_ep"].clone(), st["ep_ret"].clone(), st["obs"].clone()


class FusedCollector:
    """Same contract as Collector, but the whole epoch is ONE kernel launch (rs_rollout): MLP forward on
    the matrix cores, sampling, env step, Welford, cut/bootstrap/reset logic and the buffer rows all happen
    inside the kernel, 64 envs per wave.  Single agent, N % 64 == 0."""

    def __init__(self, env: RadSearchVec, agents: Dict[int, VecAgentPPO], steps_per_epoch: int, steps_per_episode: int,
                 global_critic_flag: bool = False, standardize: bool = True):
        assert env.number_agents == 1 and len(agents) == 1 and env.num_envs % 16 == 0 and standardize
        assert not global_critic_flag
        self.env, self.agents = env, agents
        self.T, self.L, self.N, self.A = steps_per_epoch, steps_per_episode, env.num_envs, 1
        dev = env.device
        self.buf = RolloutBuffer(self.T, self.N, 1, _lib.RS_OBS_DIM, dev)
        N = self.N
        f64 = dict(dtype=torch.float64, device=dev)
        self.cur_obs = torch.zeros(N, _lib.RS_OBS_DIM, dtype=torch.float32, device=dev)
        self.w_count, self.w_mean, self.w_sq = (torch.zeros(N, **f64) for _ in range(3))
        self.w_std = torch.ones(N, **f64)
        self.steps_in_ep = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ep_ret = torch.zeros(N, dtype=torch.float32, device=dev)
        self.done_count = torch.zeros(N, dtype=torch.int32, device=dev)
        self.oob_count = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ep_count = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ep_ret_sum = torch.zeros(N, **f64)
        self.ep_len_sum = torch.zeros(N, **f64)
        self.ep_ret_sq = torch.zeros(N, **f64)
        self.ep_ret_max = torch.zeros(N, dtype=torch.float32, device=dev)
        self.ep_ret_min = torch.zeros(N, dtype=torch.float32, device=dev)
        b = self.buf
        self._args = _lib.RsRolloutArgs(
            self.T, self.L, b.obs.data_ptr(), b.act.data_ptr(), b.logp.data_ptr(), b.val.data_ptr(), b.rew.data_ptr(),
            b.last_val.data_ptr(), b.cut.data_ptr(), b.source_tar.data_ptr(), self.cur_obs.data_ptr(),
            self.w_count.data_ptr(), self.w_mean.data_ptr(), self.w_sq.data_ptr(), self.w_std.data_ptr(),
            self.steps_in_ep.data_ptr(), self.ep_ret.data_ptr(), self.done_count.data_ptr(), self.oob_count.data_ptr(),
            self.ep_ret_sum.data_ptr(), self.ep_len_sum.data_ptr(), self.ep_count.data_ptr(),
            self.ep_ret_sq.data_ptr(), self.ep_ret_max.data_ptr(), self.ep_ret_min.data_ptr())
        self.started = False

    def start(self) -> None:
        """train.py:273-312: first reset + first Welford update."""
        obs, *_ = self.env.reset()
        self.cur_obs.copy_(obs[:, 0])
        st = DeviceWelford((self.N,), self.env.device)
        st.update(self.cur_obs[:, 0])
        self.w_count.copy_(st.count); self.w_mean.copy_(st.mean); self.w_sq.copy_(st.sq); self.w_std.copy_(st.std)
        self.started = True

    @torch.no_grad()
    def collect(self) -> Dict[str, torch.Tensor]:
        import ctypes as C
        if not self.started:
            self.start()
        ac = self.agents[0].agent
        pa, pc = mlp_params(ac.actor), mlp_params(ac.critic)
        with _lib.timed("rs_rollout"):
            _lib.check(self.env.lib.rs_rollout(self.env._h, C.byref(pa), C.byref(pc), C.byref(self._args), self.env._stream()),
                       "rs_rollout")
        with _lib.timed("rs_gae"):
            self.buf.finish(self.agents[0].gamma, self.agents[0].lam)
        one = lambda t: t.double().sum().view(1)
        return dict(DoneCount=one(self.done_count), OutOfBound=one(self.oob_count), EpCount=self.ep_count.double().sum(),
                    EpLenSum=self.ep_len_sum.sum(), EpRetSum=one(self.ep_ret_sum), EpRetSqSum=one(self.ep_ret_sq),
                    EpRetMax=self.ep_ret_max.double().max().view(1), EpRetMin=self.ep_ret_min.double().min().view(1))

    def update(self) -> Dict[int, UpdateResult]:
        return ppo_update_from_buffer(self)

    _RESUME = ("cur_obs", "w_count", "w_mean", "w_sq", "w_std", "steps_in_ep", "ep_ret")

    def resume_state(self) -> Dict[str, Any]:
        if not self.started:
            self.start()
        return dict(env=self.env.snapshot(), **{k: getattr(self, k).clone() for k in self._RESUME})

    def load_resume_state(self, st: Dict[str, Any]) -> None:
        if not self.started:
            self.start()
        self.env.restore(st["env"])
        for k in self._RESUME:
            getattr(self, k).copy_(st[k])                          # in place: rs_rollout's argument block holds their addresses


def ppo_update_from_buffer(col) -> Dict[int, UpdateResult]:
    """train.py:569-599: PPO update of every agent from the finished buffer of a collector."""
    buf = col.buf
    n_total = col.N * _world()
    w = (buf.episode_weights() / n_total).reshape(-1)
    out = {}
    for a, ag in col.agents.items():
        adv = normalize_advantages(buf.adv[:, :, a]).reshape(-1)
        X = buf.obs[:, :, a].reshape(-1, buf.obs.shape[-1])
        out[a] = ag.update_agent(X, buf.act[:, :, a].reshape(-1), adv, buf.ret[:, :, a].reshape(-1),
                                 buf.logp[:, :, a].reshape(-1), w)
    return out
