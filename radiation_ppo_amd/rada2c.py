"""RAD-A2C: the recurrent actor-critic with its PFGRU source-location module, batched over envs.

Mirrors (paths relative to the reference root):
  RNNModelActorCritic <- algos/multiagent/NeuralNetworkCores/RADA2C_core.py:477-607 as main.py:541-553 instantiates it:
                         GRU(11 + 2 -> 24) whose hidden state feeds two heads, Woms = Linear(24, 32)-Tanh-Linear(32, 8) (logits)
                         and Valms = Linear(24, 32)-Tanh-Linear(32, 1) (value) (SeqPt :351-391), plus model = PFGRUCell(40
                         particles, 3 inputs, 24 units, alpha 0.7, tanh) (:215-307) whose location prediction is appended to the
                         11 observations (step :528-548, grad_step :550-566).  Same module / parameter names, so state_dicts
                         (pyt_save/model.pt) interchange.
  RNNAgentPPO         <- AgentPPO with actor_critic_architecture="rnn" (algos/multiagent/ppo.py:644-666): update_agent (:746-813)
                         = update_model (:1047-1148: PFGRU regression + ELBO loss, back-propagated through the episode, gradient
                         clipped to norm 5) once, then up to train_pi_iters x update_rada2c (:1150-1281: per-episode PPO-clip /
                         value / detached-entropy loss, BPTT through the GRU, KL early stop), then both StepLR schedulers.
  RNNCollector        <- the 'rnn' branches of train_PPO.train (algos/multiagent/train.py:300-548): hidden states reset at every
                         epoch start and episode end (:322-329, :505-518), carried through ac.step (:345-351), bootstrap values
                         from one extra step (:462-487).

What is batched: the reference holds one env per MPI rank and loops over that rank's episodes; here an epoch's [T, N] buffer is
re-packed EPISODE-major ([L <= 120 steps, E episodes]) so that the time loop is 120 steps long whatever N is, and the
reference's "mean over ranks of the mean over the rank's episodes of the per-episode mean" becomes per-sample weights.
The per-step PFGRU forward of the collector is K11 (csrc/rs_pfgru.hip) with carried particle sets; the GRU / head arithmetic
and the two updates are torch (autograd supplies BPTT).  Random draws (initial hidden states, reparameterisation noise,
resampling) come from counter hashes keyed by global env id / episode / epoch (the documented RNG deviation of pfgru.py);
tests inject the reference's recorded draws instead (tests/golden/rada2c_core.npz).

Deviations, all forced by batching and stated here: (i) update_model clips the gradient AFTER the average over envs (the
reference clips per rank, then averages, ppo.py:1137-1141); (ii) LocLoss is the RMS over all samples (the reference reports the
loop's last episode, :1274); (iii) several agents are independent copies, each fed its own observation row.
"""
import bisect
import ctypes as C
import math
import os
import time
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .envs import RadSearchVec
from .pfgru import PFGRUCell, PredictorBank, _s64, hash_normal, hash_uniform
from .ppo import (DeviceWelford, EpochStats, RolloutBuffer, UpdateResult, _world, check_minibatch, host_read, normalize_advantages,
                  reduce_grads_and_stats, reject_unknown_kwargs, side_stream)


def _mlp_tanh(sizes) -> nn.Sequential:
    """mlp(sizes, nn.Tanh) with the trailing activation dropped (RADA2C_core.py:363-368): Linear-Tanh-...-Linear, layers at the even
    indices of the Sequential as in the reference's state_dict keys (Woms.0, Woms.2, ...)."""
    layers: List[nn.Module] = []
    for i in range(len(sizes) - 1):
        layers += [nn.Linear(int(sizes[i]), int(sizes[i + 1])), nn.Tanh()]
    return nn.Sequential(*layers[:-1])


class _SeqPt(nn.Module):
    """SeqPt (RADA2C_core.py:351-391).  pol / val: the hidden widths of the two heads (one entry each in the reference's defaults)."""

    def __init__(self, input_size: int, hid: int, pol, val, act_dim: int):
        super().__init__()
        self.seq_model = nn.GRU(input_size, hid, 1)
        self.Woms = _mlp_tanh([hid, *pol, act_dim])                                            # mlp(...)[:-1] (:363-366)
        self.Valms = _mlp_tanh([hid, *val, 1])                                                 # (:367-368)


class _LinearRows(torch.autograd.Function):
    """y = x W^T + b for a tall x [S, in] with the weight gradient reduced in two steps (batched partial products over 1024-row
    slabs, then a sum): autograd's own g^T @ x is a 32 x 24 GEMM with a reduction length of L * E ~ 5e5, which the BLAS library runs
    30x slower than its memory time (the MT32x32x256 / MT32x16x512 kernels: 3.1 of a policy pass's 15 ms; maps._LinearTall)."""
    SLAB = 1024

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        R = _LinearRows.SLAB
        m = (x.shape[0] // R) * R
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(g[:m].view(m // R, R, -1).transpose(1, 2), x[:m].view(m // R, R, -1)).sum(dim=0) + g[m:].t() @ x[m:]
        gb = g.sum(dim=0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _seq(net: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    tall = x.is_cuda and x.dim() == 2 and x.shape[0] >= 65536 and torch.is_grad_enabled()
    for layer in net:
        x = _LinearRows.apply(x, layer.weight, layer.bias) if tall and isinstance(layer, nn.Linear) else layer(x)
    return x


class _RecurrentNet(nn.Module):
    def __init__(self, *a):
        super().__init__()
        self.v_net = _SeqPt(*a)


class _Actor(nn.Module):
    def __init__(self, *a):
        super().__init__()
        self.logits_net = _RecurrentNet(*a)


class RNNModelActorCritic(nn.Module):
    def __init__(self, obs_dim: int = 11, act_dim: int = 8, hidden=((24,),), hidden_sizes_pol=((32,),), hidden_sizes_val=((32,),),
                 hidden_sizes_rec=(24,), pad_dim: int = 2, net_type: str = "rnn", seed: int = 0, **unused: Any):
        super().__init__()
        hid = int(hidden[0][0])
        pol, val = [int(v) for v in hidden_sizes_pol[0]], [int(v) for v in hidden_sizes_val[0]]     # [*hidden_size[1]] / [*hidden_size[2]] (:363-368)
        self.hid, self.obs_dim, self.act_dim, self.rec = hid, obs_dim, act_dim, int(hidden_sizes_rec[0])
        self.pi = _Actor(obs_dim + pad_dim, hid, pol, val, act_dim)
        self.model = PFGRUCell(input_size=obs_dim - 8, obs_size=obs_dim - 8, hidden_size=self.rec)
        self.num_particles, self.alpha = 40, 0.7
        # the fused kernels are built for the reference's default sizes (main.py:131-134): GRU(13, 24), 32-unit single-layer heads, 8
        # actions (K12 / K14 / K15) and a 24-unit PFGRU (K11 / K13); any other size runs the same arithmetic composed from library ops
        self.fused_policy = (obs_dim, pad_dim, hid, pol, val, act_dim) == (11, 2, 24, [32], [32], 8)
        self.fused_pfgru = self.rec == 24 and obs_dim == 11

    # ---- batched arithmetic
    def gru_cell(self, x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
        """One nn.GRU step (gate order r, z, n): x [B, 13], h [B, hid]."""
        g = self.pi.logits_net.v_net.seq_model
        gi = F.linear(x, g.weight_ih_l0, g.bias_ih_l0)
        gh = F.linear(h, g.weight_hh_l0, g.bias_hh_l0)
        H = self.hid
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        return (1.0 - z) * n + z * h

    def heads(self, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        v = self.pi.logits_net.v_net
        return _seq(v.Woms, h), _seq(v.Valms, h).squeeze(-1)

    def policy_step(self, obs: torch.Tensor, loc_pred: torch.Tensor, h: torch.Tensor):
        """step (:528-548) after the PFGRU: logits [B, 8], value [B], new GRU state [B, hid]."""
        h1 = self.gru_cell(torch.cat((obs, loc_pred), dim=1), h)
        logits, val = self.heads(h1)
        return logits, val, h1

    def gru_h0(self, u: torch.Tensor) -> torch.Tensor:
        """_get_init_states (:458-461): U(-1/sqrt(hid), 1/sqrt(hid)) from uniforms u in [0, 1)."""
        std = 1.0 / math.sqrt(self.hid)
        return (u.float() * 2.0 - 1.0) * std


def _two_step(d: torch.Tensor, x: torch.Tensor, R: int = 1024) -> torch.Tensor:
    """d^T x for tall d [S, O], x [S, I]: batched partial products over R-row slabs, then a sum (see _LinearRows)."""
    m = (d.shape[0] // R) * R
    out = d[m:].t() @ x[m:]
    if m:
        out = out + torch.bmm(d[:m].view(m // R, R, -1).transpose(1, 2), x[:m].view(m // R, R, -1)).sum(dim=0)
    return out


class HeadsLoss(torch.autograd.Function):
    """update_rada2c's loss behind the GRU on K15 (rs_a2c_heads_loss): heads, per-sample PPO-clip / value loss and their
    back-propagation in one launch; the head weight gradients are reduced from the per-sample factors with batched GEMMs.
    forward -> (-(surrogate - vf * value loss) as a scalar that back-propagates, statistics [kl, entropy, clip fraction, value loss,
    surrogate, weight sum] float64, no gradient).  The gradients are formed in forward (the loss is only ever differentiated)."""

    @staticmethod
    def forward(ctx, hs, w1, b1, w2, b2, v1, vb1, v2, vb2, packed, act, adv, ret, logp_old, wt, clip, vf):
        S = hs.shape[0]
        dev = hs.device
        hs = hs.contiguous()
        dhs = torch.empty(S, 24, dtype=torch.float32, device=dev)
        dfac = torch.empty(S, 80, dtype=torch.float32, device=dev)
        tfac = torch.empty(S, 64, dtype=torch.float32, device=dev)
        stats = torch.empty((S + 63) // 64, 8, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().rs_a2c_heads_loss(packed.data_ptr(), hs.data_ptr(), act.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                                                 logp_old.data_ptr(), wt.data_ptr(), dhs.data_ptr(), dfac.data_ptr(), tfac.data_ptr(),
                                                 stats.data_ptr(), S, float(clip), float(vf),
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rs_a2c_heads_loss")
        st = stats.double().sum(dim=0)[:6]
        g1 = _two_step(dfac[:, :64], hs)                                  # [64, 24]: Woms[0] | Valms[0]
        g2 = _two_step(dfac[:, 64:72], tfac[:, :32])                      # [8, 32]
        g3 = _two_step(dfac[:, 72:73], tfac[:, 32:])                      # [1, 32]
        m = (S // 1024) * 1024
        gb = dfac[m:].sum(dim=0)
        if m:
            gb = gb + dfac[:m].view(m // 1024, 1024, 80).sum(dim=1).sum(dim=0)
        ctx.save_for_backward(dhs, g1, g2, g3, gb)
        loss = (-(st[4] - vf * st[3])).float()
        ctx.mark_non_differentiable(st)
        return loss, st

    @staticmethod
    def backward(ctx, g, _):
        dhs, g1, g2, g3, gb = ctx.saved_tensors
        return (g * dhs, g * g1[:32], g * gb[:32], g * g2, g * gb[64:72], g * g1[32:], g * gb[32:64], g * g3, g * gb[72:73],
                None, None, None, None, None, None, None, None)


class GRUSequence(torch.autograd.Function):
    """torch.nn.GRU(13, 24, 1) over an episode-major batch with the recurrence in K12 (csrc/rs_gru.hip): x [L, E, 13], h0 [E, 24]
    -> h_t [L, E, 24].  Forward: one GEMM for the input projection of all (t, episode), then the time loop in one launch (one
    episode per lane).  Backward: the time loop in one launch (dL/d gate pre-activations), then the four weight gradients as
    two-step reductions.  x and h0 receive no gradient (observations / drawn initial states)."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh):
        L, E, K = x.shape
        H = w_hh.shape[1]
        assert H == 24 and w_hh.shape[0] == 72 and x.is_cuda
        lib = _lib.load()
        x = x.contiguous(); h0 = h0.contiguous()
        gi = torch.addmm(b_ih, x.view(L * E, K), w_ih.t()).view(L, E, 3 * H)
        whh_t = torch.zeros(H, 80, dtype=torch.float32, device=x.device); whh_t[:, :3 * H] = w_hh.t()
        bhh = torch.zeros(80, dtype=torch.float32, device=x.device); bhh[:3 * H] = b_hh
        hs = torch.empty(L, E, H, dtype=torch.float32, device=x.device)
        gates = torch.empty(L, E, 4 * H, dtype=torch.float32, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.rs_gru_forward(gi.data_ptr(), h0.data_ptr(), whh_t.data_ptr(), bhh.data_ptr(), hs.data_ptr(), gates.data_ptr(),
                                      L, E, st), "rs_gru_forward")
        ctx.save_for_backward(x, h0, w_hh, hs, gates)
        return hs

    @staticmethod
    def backward(ctx, dhs):
        x, h0, w_hh, hs, gates = ctx.saved_tensors
        L, E, H = hs.shape
        lib = _lib.load()
        whh = torch.zeros(3 * H, 32, dtype=torch.float32, device=x.device); whh[:, :H] = w_hh
        dgi = torch.empty(L, E, 3 * H, dtype=torch.float32, device=x.device)
        dgh = torch.empty_like(dgi)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        dhs = dhs.contiguous()                                                     # stays referenced until the launch is enqueued
        _lib.check(lib.rs_gru_backward(dhs.data_ptr(), hs.data_ptr(), gates.data_ptr(), h0.data_ptr(), whh.data_ptr(),
                                       dgi.data_ptr(), dgh.data_ptr(), L, E, st), "rs_gru_backward")
        h_prev = torch.cat((h0.unsqueeze(0), hs[:-1]), dim=0)
        gw_ih = torch.bmm(dgi.transpose(1, 2), x).sum(dim=0)               # per-step partial products, then the sum over time
        gw_hh = torch.bmm(dgh.transpose(1, 2), h_prev).sum(dim=0)
        return None, None, gw_ih, gw_hh, dgi.sum(dim=(0, 1)), dgh.sum(dim=(0, 1))


# ------------------------------------------------------------------------------------------------ draws
class HashDraws:
    """Counter-hash draws of one pass over an episode batch: keys [E] int64 identify (seed, global env, episode, epoch, pass)."""

    def __init__(self, keys: torch.Tensor, P: int = 40, H: int = 24, hid: int = 24):
        self.k = keys
        dev = keys.device
        self._pu = (torch.arange(P, dtype=torch.int64, device=dev).view(P, 1) * 4096 + torch.arange(H, dtype=torch.int64, device=dev).view(1, H))
        self._g = torch.arange(hid, dtype=torch.int64, device=dev)

    def _key(self, kind: int, t: int = 0) -> torch.Tensor:
        return (self.k * 1000003) ^ _s64((t * 8 + kind) * 0xA24BAED4963EE407)

    def pf_h0(self) -> torch.Tensor:
        return hash_uniform(self._key(0).view(-1, 1, 1) * 1048583 + self._pu.unsqueeze(0)).float()

    def gru_h0_u(self) -> torch.Tensor:
        return hash_uniform(self._key(3).view(-1, 1) * 1048583 + self._g.unsqueeze(0))

    def eps(self, t: int) -> torch.Tensor:
        return hash_normal(self._key(1, t).view(-1, 1, 1) * 1048583 + self._pu.unsqueeze(0))

    def resample(self, t: int) -> Dict[str, torch.Tensor]:
        return dict(resample_u=hash_uniform(self._key(2, t).view(-1, 1) * 1048583 + self._pu[:, 0].unsqueeze(0)))


class Scratch:
    """Persistent device scratch for the buffers of an update whose size follows the epoch's episode count, which moves by a fraction
    of a percent from epoch to epoch.  A buffer is kept, grown with 1/8 headroom when too small, and handed out as a view.  Left to
    torch's caching allocator, a request slightly larger than last epoch's multi-GB block is met by a fresh hipMalloc (hundreds of
    milliseconds for K13's 30 GB of gates) while the old block stays reserved: the RAD-A2C update took 1.7 s or 2.3-2.7 s depending on
    whether the epoch happened to have more episodes than any before it."""

    def __init__(self):
        self.bufs: Dict[Any, torch.Tensor] = {}

    def get(self, name: str, shape, dtype, device) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        key = (name, dtype, str(device))
        b = self.bufs.get(key)
        if b is None or b.numel() < n:
            self.bufs.pop(key, None)
            b = None                                                         # release the old block before asking for the larger one
            b = self.bufs[key] = torch.empty(n + n // 8, dtype=dtype, device=device)
        return b[:n].view(*shape)


class KernelDraws:
    """HashDraws with every draw of the pass produced by ONE launch (rs_pfgru_draws) instead of ~25 int64 element-wise launches per
    step; same keys, same hash, same values (the normals to float32 rounding of the library log / cos).  scratch: where the draws go
    (Scratch; the previous pass's draws are overwritten, stream ordered) -- None: fresh tensors."""

    def __init__(self, keys: torch.Tensor, L: int, scratch: Optional[Scratch] = None, tag: str = ""):
        self.k = keys.contiguous()
        E, dev = keys.shape[0], keys.device
        if scratch is None:
            self._pf = torch.empty(E, 40, 24, dtype=torch.float32, device=dev)
            self._eps = torch.empty(L, E, 40, 24, dtype=torch.float32, device=dev)
            self._u = torch.empty(L, E, 40, dtype=torch.float64, device=dev)
        else:
            self._pf = scratch.get("draws_pf" + tag, (E, 40, 24), torch.float32, dev)
            self._eps = scratch.get("draws_eps" + tag, (L, E, 40, 24), torch.float32, dev)
            self._u = scratch.get("draws_u" + tag, (L, E, 40), torch.float64, dev)
        _lib.check(_lib.load().rs_pfgru_draws(self.k.data_ptr(), E, L, self._pf.data_ptr(), self._eps.data_ptr(), self._u.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rs_pfgru_draws")

    def pf_h0(self):
        return self._pf

    def eps(self, t):
        return self._eps[t]

    def resample(self, t):
        return dict(resample_u=self._u[t])


class KeyDraws:
    """The draws of KernelDraws WITHOUT the buffers: K13's forward walk evaluates the counter hash itself (rs_pfgru_train_keyed; same keys,
    same arithmetic, bit-identical values), as K11 always has.  The draws launch wrote 104 bytes per particle-step for the walk to read
    back -- 8.2 GB and 2.3 ms per pass at 16.5 k episodes, 15 passes per update.  materialise(L): the buffers after all (tests)."""

    def __init__(self, keys: torch.Tensor):
        self.k = keys.contiguous()

    def materialise(self, L: int) -> "KernelDraws":
        return KernelDraws(self.k, L)


class RecordedDraws:
    """The reference's own draws (tests): pf_h0 [E, P, H], gru_h0 [E, hid], eps [L, E, P, H], idx [L, E, P]."""

    def __init__(self, pf_h0, gru_h0, eps, idx):
        self._pf, self._g, self._eps, self._idx = pf_h0, gru_h0, eps, idx

    def pf_h0(self):
        return self._pf

    def gru_h0(self):
        return self._g

    def eps(self, t):
        return self._eps[t]

    def resample(self, t):
        return dict(resample_idx=self._idx[t])


class RecordedKernelDraws(RecordedDraws):
    """RecordedDraws in the buffers the kernels read (K11: rs_pfgru_step_recorded, K13: rs_pfgru_train with u = NULL): the
    reference's own h0 / noise / resampling indices go straight into the HIP kernels (tests/test_rows_f_golden_gpu.py)."""

    def __init__(self, pf_h0, gru_h0, eps, idx):
        super().__init__(pf_h0.contiguous(), gru_h0, eps.contiguous(), idx)
        self._idx32 = idx.to(torch.int32).contiguous()
        self._u = None


# ------------------------------------------------------------------------------------------------ episode-major batch
@dataclass
class EpisodeBatch:
    """An epoch's samples, episode-major.  L = longest episode, E = episodes (the trailing partial one of every env included,
    PPOBuffer.get :480-487).  Padded steps carry weight 0."""
    X: torch.Tensor          # [L, E, 11] standardised observations
    act: torch.Tensor        # [L, E] int64
    adv: torch.Tensor        # [L, E]
    ret: torch.Tensor        # [L, E]
    logp: torch.Tensor       # [L, E]
    src: torch.Tensor        # [L, E, 2] source coordinates (cm)
    valid: torch.Tensor      # [L, E] bool
    lens: torch.Tensor       # [E] int64
    w_ep: torch.Tensor       # [E] 1 / (global env count x episodes of the env): the weight of an episode's loss
    key: torch.Tensor        # [E] int64 draw keys
    lens_host: Optional[List[int]] = None

    @property
    def w(self) -> torch.Tensor:
        """per-sample weights: w_ep / len on valid steps."""
        return (self.w_ep / self.lens.float()).unsqueeze(0) * self.valid.float()

    def chunk(self, sl: slice) -> "EpisodeBatch":
        """The episodes sl, trimmed to the longest of them (with episodes sorted by length -- pack_episodes(sort_by_length=True) --
        later chunks are much shorter than the batch: a learned policy ends most episodes after ~30 of 120 steps)."""
        if self.lens_host is None:
            self.lens_host = self.lens.tolist()
        Lc = max(self.lens_host[sl])
        return EpisodeBatch(X=self.X[:Lc, sl], act=self.act[:Lc, sl], adv=self.adv[:Lc, sl], ret=self.ret[:Lc, sl], logp=self.logp[:Lc, sl],
                            src=self.src[:Lc, sl], valid=self.valid[:Lc, sl], lens=self.lens[sl], w_ep=self.w_ep[sl], key=self.key[sl],
                            lens_host=self.lens_host[sl])


def pack_episodes(obs, act, adv, ret, logp, src, cut, n_total: int, env_id_base: int = 0, seed: int = 0, epoch: int = 0,
                  sort_by_length: bool = False) -> EpisodeBatch:
    """[T, N, ...] time-major columns -> EpisodeBatch.  cut [T, N] closes an episode (terminal, timeout or epoch end).
    sort_by_length: episodes in descending length (stable) instead of (env, order in the env); every episode keeps its weight and
    its draw key, so the update is the same sum in another order."""
    T, N = cut.shape
    dev = cut.device
    c = cut.long()
    start = torch.ones(T, N, dtype=torch.int64, device=dev)
    start[1:] = c[:-1]
    k_in_env = torch.cumsum(start, dim=0) - 1                               # episode index inside the env's column
    n_ep = k_in_env[-1] + 1                                                  # [N]
    off = torch.cumsum(n_ep, dim=0) - n_ep
    eid = k_in_env + off.unsqueeze(0)                                        # [T, N] global episode index
    tt = torch.arange(T, device=dev).unsqueeze(1).expand(T, N)
    t0 = torch.cummax(torch.where(start.bool(), tt, torch.zeros_like(tt)), dim=0).values
    pos = tt - t0
    E = int(n_ep.sum().item())
    L = int(pos.max().item()) + 1
    flat = (pos * E + eid).reshape(-1)

    def scat(x, fill=0.0):
        tail = x.shape[2:]
        out = torch.full((L * E,) + tuple(tail), fill, dtype=x.dtype, device=dev)
        out[flat] = x.reshape((T * N,) + tuple(tail))
        return out.view((L, E) + tuple(tail))
    lens = torch.zeros(E, dtype=torch.int64, device=dev)
    lens.scatter_add_(0, eid.reshape(-1), torch.ones(T * N, dtype=torch.int64, device=dev))
    valid = torch.arange(L, device=dev).unsqueeze(1) < lens.unsqueeze(0)
    env_of = torch.repeat_interleave(torch.arange(N, device=dev), n_ep)
    w_ep = 1.0 / (float(n_total) * n_ep[env_of].float())
    k_of = torch.arange(E, device=dev) - off[env_of]
    key = hash_uniform(((env_of + int(env_id_base)) * 4096 + k_of) ^ _s64((int(seed) * 0x2545F4914F6CDD1D) ^ (int(epoch) * 0x9E3779B97F4A7C15))).mul(2.0 ** 52).long()
    B = EpisodeBatch(X=scat(obs), act=scat(act), adv=scat(adv), ret=scat(ret), logp=scat(logp), src=scat(src), valid=valid,
                     lens=lens, w_ep=w_ep, key=key)
    if sort_by_length:
        order = torch.argsort(lens, descending=True, stable=True)
        B = EpisodeBatch(X=B.X[:, order], act=B.act[:, order], adv=B.adv[:, order], ret=B.ret[:, order], logp=B.logp[:, order],
                         src=B.src[:, order], valid=B.valid[:, order], lens=lens[order], w_ep=w_ep[order], key=key[order])
    return B


# ------------------------------------------------------------------------------------------------ K13 plumbing
PF_TRAIN_WEIGHT_FLOATS, PF_TRAIN_GRAD_FLOATS = 3680, 3376            # include/radsearch.h


def pack_train_weights(cell) -> torch.Tensor:
    """The PFGRU's parameters in the layout rs_pfgru_train reads (csrc/rs_pfgru_train.hip): k-major blocks, used by the forward products column-wise
    and by the backward (transposed) products row-wise."""
    assert cell.h_dim == 24 and cell.num_particles == 40 and cell.input_size == 3, "rs_pfgru_train is built for 40 particles x 24 units"
    dev = cell.fc_z.weight.device
    w = torch.zeros(PF_TRAIN_WEIGHT_FLOATS, dtype=torch.float32, device=dev)
    o = 0

    def put(t, n):
        nonlocal o
        w[o:o + t.numel()] = t.reshape(-1)
        o += n
    zr = torch.cat([cell.fc_z.weight, cell.fc_r.weight], 0)                       # [48, 27]
    put(F.pad(zr.t(), (0, 0, 0, 1)), 28 * 48)                                     # [28][48] k-major, row 27 zero
    put(torch.cat([cell.fc_z.bias, cell.fc_r.bias], 0), 48)
    put(F.pad(cell.fc_n.weight.t(), (0, 0, 0, 1)), 28 * 48)
    put(cell.fc_n.bias, 48)
    put(torch.cat([cell.fc_obs.weight.reshape(-1), cell.fc_obs.bias.reshape(-1)]), 32)
    put(F.pad(cell.hid_obs[0].weight.t(), (0, 8)), 24 * 32)                       # [24 k][32]
    put(F.pad(cell.hid_obs[0].bias, (0, 8)), 32)
    put(torch.cat([cell.hid_obs[2].weight.reshape(-1), cell.hid_obs[2].bias.reshape(-1)]), 64)
    assert o == PF_TRAIN_WEIGHT_FLOATS
    return w


PF_POLICY_WEIGHT_FLOATS = 5296                                       # include/radsearch.h: RS_RNN_POLICY_WEIGHT_FLOATS


def pack_policy_weights(ac: "RNNModelActorCritic") -> torch.Tensor:
    """GRU + head parameters in the layout rs_rnn_policy_step reads (csrc/rs_rnn_policy.hip): every block k-major, padded to 16."""
    v = ac.pi.logits_net.v_net
    g = v.seq_model
    assert ac.hid == 24 and g.weight_ih_l0.shape == (72, 13) and v.Woms[0].out_features == 32 and v.Valms[0].out_features == 32 \
        and v.Woms[2].out_features == 8, "rs_rnn_policy_step is built for GRU(13, 24) with 32-unit heads and 8 actions"
    parts = [F.pad(g.weight_ih_l0.t(), (0, 8)), F.pad(g.bias_ih_l0, (0, 8)), F.pad(g.weight_hh_l0.t(), (0, 8)), F.pad(g.bias_hh_l0, (0, 8)),
             v.Woms[0].weight.t(), v.Woms[0].bias, v.Valms[0].weight.t(), v.Valms[0].bias,
             F.pad(v.Woms[2].weight.t(), (0, 8)), F.pad(v.Woms[2].bias, (0, 8)),
             F.pad(torch.cat([v.Valms[2].weight.reshape(-1), v.Valms[2].bias.reshape(-1)]), (0, 15))]
    w = torch.cat([p.reshape(-1) for p in parts]).float().contiguous()
    assert w.numel() == PF_POLICY_WEIGHT_FLOATS
    return w


def unpack_train_grads(cell, g: torch.Tensor) -> Dict[str, torch.Tensor]:
    """A summed gradient slab [PF_TRAIN_GRAD_FLOATS] -> gradients by parameter name (PFGRUCell.named_parameters)."""
    zr = g[:48 * 28].view(48, 28)
    n = g[48 * 28:2 * 48 * 28].view(48, 28)
    o = 2 * 48 * 28
    h0 = g[o:o + 600].view(24, 25); o += 600
    h2 = g[o:o + 50].view(2, 25); o += 50
    fo = g[o:o + 28]
    return {"fc_z.weight": zr[:24, :27], "fc_z.bias": zr[:24, 27], "fc_r.weight": zr[24:, :27], "fc_r.bias": zr[24:, 27],
            "fc_n.weight": n[:, :27], "fc_n.bias": n[:, 27], "fc_obs.weight": fo[:27].view(1, 27), "fc_obs.bias": fo[27:28],
            "hid_obs.0.weight": h0[:, :24], "hid_obs.0.bias": h0[:, 24], "hid_obs.2.weight": h2[:, :24], "hid_obs.2.bias": h2[:, 24]}


# ------------------------------------------------------------------------------------------------ agent
@dataclass
class BpArgs:
    """ppo.py:160-167."""
    bp_decay: float = 0.1
    l2_weight: float = 1.0
    l1_weight: float = 0.0
    elbo_weight: float = 1.0
    area_scale: float = 2500.0


class RNNAgentPPO:
    """AgentPPO, 'rnn' branch (ppo.py:601-683, :776-811).  `minibatch` (ppo.py:580) is validated and has no effect, as in the reference:
    update_rada2c assigns it (ppo.py:1159-1160) and never reads it.  `train_v_iters` / `critic_learning_rate` have no counterpart
    (RAD-A2C's value head is trained by the policy optimiser, ppo.py:1225, :1256-1258).  Any other unnamed key raises."""

    _NO_EFFECT = ("train_v_iters", "critic_learning_rate")

    def __init__(self, id: int, observation_space: int = 11, action_space: int = 8, steps_per_epoch: int = 480,
                 steps_per_episode: int = 120, number_of_agents: int = 1, actor_critic_architecture: str = "rnn",
                 actor_critic_args: Optional[Dict[str, Any]] = None, train_pi_iters: int = 40, train_pfgru_iters: int = 15,
                 actor_learning_rate: float = 3e-4, pfgru_learning_rate: float = 5e-3, gamma: float = 0.99, alpha: float = 0.1,
                 clip_ratio: float = 0.2, target_kl: float = 0.07, lam: float = 0.9, bp_args: Optional[Any] = None,
                 env_height: float = 2500.0, seed: int = 0, device="cuda:0", episode_chunk: int = 32768,
                 GlobalCriticOptimizer=None, minibatch: int = 1, **other: Any):
        reject_unknown_kwargs("RNNAgentPPO", other, self._NO_EFFECT)
        self.minibatch = check_minibatch(minibatch)
        # episode_chunk: episodes per pass of the update kernels.  One chunk for a whole 4096-env epoch (~16.5 k episodes): the recurrent
        # kernels (K12: one episode per lane) and the K11 passes are latency bound per launch, so three 8192-episode chunks cost three times
        # one 16 k chunk (RAD-A2C bench: 895 -> 972 k env steps/s).  K13's scratch grows with it: 1.8 MB per full-length episode (gates,
        # particle sets, noise) = ~30 GB at 16.5 k episodes, 60 GB at the cap -- sized for the 288 GB of an MI355X.
        if actor_critic_architecture != "rnn":
            raise ValueError("Unsupported Neural Network type requested")
        if GlobalCriticOptimizer is not None:
            raise Exception("No global critic option for RAD-A2C")             # ppo.py:651-652
        self.id, self.device = id, torch.device(device)
        self.gamma, self.lam, self.alpha, self.clip_ratio, self.target_kl = gamma, lam, alpha, clip_ratio, target_kl
        self.train_pi_iters, self.train_pfgru_iters = train_pi_iters, train_pfgru_iters
        self.reduce_pfgru_iters = True
        b = bp_args if bp_args is not None else BpArgs(area_scale=env_height)
        self.bp_args = BpArgs(*(b if isinstance(b, tuple) else (b.bp_decay, b.l2_weight, b.l1_weight, b.elbo_weight, b.area_scale)))
        self.env_height, self.seed, self.episode_chunk = float(env_height), int(seed), int(episode_chunk)
        self.scratch = Scratch()
        self.k13_particle_steps: List[int] = []          # particle-steps of each K13 launch of the current update_model (bench roofline)
        self.k11_particle_steps: List[int] = []          # the same for the K11 passes (rs_pfgru_pass) of the current update_agent
        args = dict(actor_critic_args or {})
        args.setdefault("obs_dim", observation_space); args.setdefault("act_dim", action_space)
        self.agent = RNNModelActorCritic(**args).to(self.device)
        self.pi_optimizer = torch.optim.Adam(self.agent.pi.parameters(), lr=actor_learning_rate)
        self.model_optimizer = torch.optim.Adam(self.agent.model.parameters(), lr=pfgru_learning_rate)
        self.pi_scheduler = torch.optim.lr_scheduler.StepLR(self.pi_optimizer, step_size=100, gamma=0.99)        # ppo.py:205-210
        self.pfgru_scheduler = torch.optim.lr_scheduler.StepLR(self.model_optimizer, step_size=100, gamma=0.99)
        self.epochs_done = 0
        self.agent.eval()

    def policy_weights(self) -> torch.Tensor:
        """The GRU + head parameters packed for K14, re-packed IN PLACE when a parameter changed (a captured collector step keeps
        reading the same buffer)."""
        ver = tuple(p._version for p in self.agent.pi.parameters())
        if getattr(self, "_polw_ver", None) != ver:
            with torch.no_grad():
                w = pack_policy_weights(self.agent)
                if getattr(self, "_polw", None) is None:
                    self._polw = w
                else:
                    self._polw.copy_(w)
            self._polw_ver = ver
        return self._polw

    def policy_step_hip(self, x, loc, h, u=None, h_out=None, logits=None, value=None, act=None, logp=None, weights=None) -> None:
        """RNNModelActorCritic.step behind the PFGRU on K14 (rs_rnn_policy_step); outputs are written to the tensors given."""
        w = self.policy_weights() if weights is None else weights
        ptr = lambda t: None if t is None else t.data_ptr()
        for t in (x, loc, h, u, h_out, logits, value, act, logp):
            assert t is None or (t.is_cuda and t.is_contiguous())
        _lib.check(_lib.load().rs_rnn_policy_step(w.data_ptr(), x.data_ptr(), loc.data_ptr(), h.data_ptr(), ptr(u), ptr(h_out), ptr(logits),
                                                  ptr(value), ptr(act), ptr(logp), x.shape[0],
                                                  C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "rs_rnn_policy_step")

    def policy_step_rows(self, x, loc, h, u, a: int, value, act=None, logp=None, act8=None, mask8=None) -> None:
        """K14 on agent a's rows of the collectors' [N, A, .] tensors (no contiguous copies): x [N, A, 11], loc [N, A, 2], u [N, A] or
        None (value only), h [N, 24] updated in place when an action is drawn; act8 [N, A] int8 receives the action for rs_step;
        mask8 [N] uint8: only those envs are evaluated (the bootstrap round)."""
        N, A = x.shape[0], x.shape[1]
        w = self.policy_weights()
        fl = lambda t, k: None if t is None else t.data_ptr() + 4 * a * k
        _lib.check(_lib.load().rs_rnn_policy_step_rows(w.data_ptr(), fl(x, _lib.RS_OBS_DIM), A * _lib.RS_OBS_DIM, fl(loc, 2), 2 * A, h.data_ptr(),
                                                       fl(u, 1), A, None if u is None else h.data_ptr(), value.data_ptr(),
                                                       None if act is None else act.data_ptr(), None if logp is None else logp.data_ptr(),
                                                       None if act8 is None else act8.data_ptr() + a, A,
                                                       None if mask8 is None else mask8.data_ptr(), N,
                                                       C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "rs_rnn_policy_step_rows")

    def reduce_pfgru_training(self) -> None:
        """ppo.py:685-689."""
        if self.reduce_pfgru_iters:
            self.train_pfgru_iters = 5
            self.reduce_pfgru_iters = False

    def sync_params(self) -> None:
        if _world() > 1:
            for p in self.agent.parameters():
                dist.broadcast(p.data, src=0)

    # ---- PFGRU over an episode chunk; returns loc [L, E, 2] and (optionally) per-particle predictions [L, E, P, 2]
    def _pfgru_pass(self, X3: torch.Tensor, draws, want_particles: bool):
        cell = self.agent.model
        L, E = X3.shape[0], X3.shape[1]
        h = draws.pf_h0()
        p = torch.full((E, cell.num_particles), math.log(1.0 / cell.num_particles), dtype=torch.float32, device=X3.device)
        locs, parts = [], []
        for t in range(L):
            loc, (h, p) = cell(X3[t], (h, p), draws.eps(t), **draws.resample(t))
            locs.append(loc)
            if want_particles:
                parts.append(cell.particle_predictions(h))                        # particle_pred[zz] (ppo.py:1079)
        return torch.stack(locs), (torch.stack(parts) if want_particles else None)

    def _pfgru_pass_hip(self, X: torch.Tensor, draws: "HashDraws", lens_host: Optional[List[int]] = None) -> torch.Tensor:
        """The no-grad PFGRU pass of grad_step (:555-558) on K11: X [L, E, 11] -> loc [L, E, 2].  Particle sets start from the
        reset kernel's hash draws and are carried; the draw keys are the chunk's episode keys."""
        from .pfgru import pack_weights
        lib = _lib.load()
        L, E = X.shape[0], X.shape[1]
        dev = X.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        h = torch.empty(1, E, 40, 24, dtype=torch.float32, device=dev)
        p = torch.empty(1, E, 40, dtype=torch.float32, device=dev)
        base = draws.k.contiguous().view(1, E)
        episode = torch.ones(E, dtype=torch.int64, device=dev)
        calls = torch.arange(L, dtype=torch.int64, device=dev).view(L, 1).expand(L, E).contiguous()   # the step counter of every launch
        wts = pack_weights([self.agent.model])
        loc = torch.zeros(L, E, 2, dtype=torch.float32, device=dev)
        # episodes sorted by descending length: the ones still running at step t are a prefix, the launch covers only those
        alive = [E] * L
        if lens_host is not None and all(a >= b for a, b in zip(lens_host, lens_host[1:])):
            asc = lens_host[::-1]
            alive = [E - bisect.bisect_right(asc, t) for t in range(L)]
        # the reset and the L step launches of the pass go out from ONE library call: the policy loop issues 40 passes per update, and
        # ~120 ctypes calls per pass made it host-bound on a busy box (46 ms per policy iteration against 14 ms of kernels)
        Xc = X.contiguous()
        alive_h = (C.c_int32 * L)(*alive)
        self.k11_particle_steps.append(40 * (int(sum(lens_host)) if lens_host is not None else L * E))
        with _lib.timed("rs_pfgru_pass"):
            _lib.check(lib.rs_pfgru_pass(wts.data_ptr(), Xc.data_ptr(), h.data_ptr(), p.data_ptr(), base.data_ptr(), episode.data_ptr(), calls.data_ptr(),
                                         float(self.agent.model.resamp_alpha), loc.data_ptr(), alive_h, L, E, st), "rs_pfgru_pass")
        return loc

    def _pfgru_pass_hip_recorded(self, X: torch.Tensor, draws: "RecordedKernelDraws") -> torch.Tensor:
        """_pfgru_pass_hip with the draws supplied (K11's recorded-draw instantiation): X [L, E, 11] -> loc [L, E, 2]."""
        from .pfgru import pack_weights
        lib = _lib.load()
        L, E = X.shape[0], X.shape[1]
        dev = X.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        from .pfgru import PredictorBank
        h = PredictorBank.to_quads(draws.pf_h0().view(1, E, 40, 24))          # the kernel's quad-major particle sets
        p = torch.full((1, E, 40), math.log(1.0 / 40), dtype=torch.float32, device=dev)
        wts = pack_weights([self.agent.model])
        loc = torch.zeros(L, E, 2, dtype=torch.float32, device=dev)
        Xc = X.contiguous()
        for t in range(L):
            _lib.check(lib.rs_pfgru_step_recorded(wts.data_ptr(), Xc[t].data_ptr(), h.data_ptr(), p.data_ptr(), draws._eps[t].data_ptr(),
                                                  draws._idx32[t].data_ptr(), None, 1, float(self.agent.model.resamp_alpha),
                                                  loc[t].data_ptr(), E, 1, st), "rs_pfgru_step_recorded")
        return loc

    def model_loss(self, B: EpisodeBatch, sl: slice, draws) -> torch.Tensor:
        """Sum over the chunk's episodes of w_ep x total_loss (ppo.py:1062-1128)."""
        a = self.bp_args
        B, sl = B.chunk(sl), slice(None)
        X3, valid, lens = B.X[:, sl, :3], B.valid[:, sl], B.lens[sl]
        L = X3.shape[0]
        tar = B.src[:, sl] / a.area_scale                                          # :1067 (padded rows: weight 0)
        loc, part = self._pfgru_pass(X3, draws, True)
        tt = torch.arange(L, device=X3.device, dtype=torch.float64).unsqueeze(1)
        bp = torch.exp(a.bp_decay * tt) * valid.double()
        bp = (bp / bp.sum(dim=0, keepdim=True)).float()                            # :1074-1075 (numpy float64, then FloatTensor)
        bp3 = bp.unsqueeze(-1)
        l2_loss = ((loc - tar) ** 2 * bp3).sum(dim=(0, 2))                          # torch.sum over the episode (:1093)
        n_el = (lens * 2).float()
        l1_loss = 10.0 * ((loc - tar).abs() * bp3).sum(dim=(0, 2)) / n_el
        pred_loss = a.l2_weight * l2_loss + a.l1_weight * l1_loss
        tp = tar.unsqueeze(2)
        bp4 = bp3.unsqueeze(2)
        v4 = valid.float().unsqueeze(-1)
        y2 = torch.exp(-((part - tp) ** 2) * bp4).mean(dim=2)                       # mean over particles (:1117-1119)
        l2p = (-(y2.log()) * v4).sum(dim=(0, 2)) / n_el
        y1 = torch.exp(-((part - tp).abs()) * bp4).mean(dim=2)
        l1p = 10.0 * (-(y1.log()) * v4).sum(dim=(0, 2)) / n_el
        total = pred_loss + a.elbo_weight * (a.l2_weight * l2p + a.l1_weight * l1p)
        return (B.w_ep[sl] * total).sum()

    def model_pass_hip(self, B: EpisodeBatch, sl: slice, d: "KernelDraws"):
        """model_loss + backward for an episode chunk on K13 (rs_pfgru_train): returns (loss, summed gradient slab, idx [L, E, 40])."""
        a = self.bp_args
        dev = self.device
        # what does not change between the iterations of update_model is prepared once per chunk and kept on the batch
        cache = B.__dict__.setdefault("_k13", {})
        ck = (sl.start, sl.stop, a.bp_decay, a.area_scale)
        if ck not in cache:
            Bc = B.chunk(sl)
            X = Bc.X.contiguous()
            L, E = X.shape[0], X.shape[1]
            tt = torch.arange(L, device=dev, dtype=torch.float64).unsqueeze(1)
            bp = torch.exp(a.bp_decay * tt) * Bc.valid.double()
            bp = (bp / bp.sum(dim=0, keepdim=True)).float().contiguous()           # :1074-1075
            cache[ck] = (X, (Bc.src / a.area_scale).float().contiguous(), bp, Bc.lens.contiguous(), Bc.w_ep.float().contiguous(),
                         int(sum(Bc.lens_host)) * 40)
        X, tar, bp, lens, w_ep, particle_steps = cache[ck]
        L, E = X.shape[0], X.shape[1]
        # the walk's scratch (particle sets, log weights, indices, gates: 2.4 MB per full-length episode) is ONE set of buffers shared by
        # all chunks and iterations -- launches are stream ordered, and a chunk's pass is done with them when the next one starts
        sc = self.scratch
        hs = sc.get("k13_hs", (L, E, 40, 24), torch.float32, dev)
        ps = sc.get("k13_ps", (L, E, 40), torch.float32, dev)
        idx = sc.get("k13_idx", (L, E, 40), torch.int32, dev)
        gates = sc.get("k13_gates", (L * E * 40 * 96,), torch.float32, dev)         # the forward walk's gates: 384 B per particle-step
        loss = torch.empty(E, dtype=torch.float32, device=dev)
        slab = torch.empty(E, PF_TRAIN_GRAD_FLOATS, dtype=torch.float32, device=dev)
        w = pack_train_weights(self.agent.model)
        # for the bench's roofline entry: one count per launch of the current update_model, in launch order (as _lib.EVENTS["rs_pfgru_train"])
        self.k13_particle_steps.append(particle_steps)
        if isinstance(d, KeyDraws):
            idx.zero_()
            h0s = sc.get("k13_h0", (E, 40, 24), torch.float32, dev)
            with _lib.timed("rs_pfgru_train"):
                _lib.check(_lib.load().rs_pfgru_train_keyed(w.data_ptr(), X.data_ptr(), tar.data_ptr(), bp.data_ptr(), lens.data_ptr(), w_ep.data_ptr(),
                                                            d.k.data_ptr(), h0s.data_ptr(), hs.data_ptr(), ps.data_ptr(), gates.data_ptr(), idx.data_ptr(), loss.data_ptr(),
                                                            slab.data_ptr(), L, E, float(self.agent.model.resamp_alpha), float(a.l2_weight),
                                                            float(a.l1_weight), float(a.elbo_weight),
                                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rs_pfgru_train_keyed")
            return loss.double().sum(), slab.sum(dim=0), idx
        if d._u is None:                                                           # recorded draws: idx is the kernel's INPUT
            idx.copy_(d._idx32)
        else:
            idx.zero_()                                                            # steps beyond an episode's end are never written
        with _lib.timed("rs_pfgru_train"):
            _lib.check(_lib.load().rs_pfgru_train(w.data_ptr(), X.data_ptr(), tar.data_ptr(), bp.data_ptr(), lens.data_ptr(), w_ep.data_ptr(),
                                                  d._pf.data_ptr(), d._eps.data_ptr(), None if d._u is None else d._u.data_ptr(), hs.data_ptr(), ps.data_ptr(),
                                                  gates.data_ptr(), idx.data_ptr(), loss.data_ptr(), slab.data_ptr(), L, E, float(self.agent.model.resamp_alpha),
                                                  float(a.l2_weight), float(a.l1_weight), float(a.elbo_weight),
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rs_pfgru_train")
        return loss.double().sum(), slab.sum(dim=0), idx

    # K13's scratch per episode of L steps, bytes: gates 40 x 96, particle sets 40 x 24 floats, log weights + indices 2 x 40 words per step
    # (+ Scratch's 1/8 headroom); the noise and the resampling uniforms are hashed in the kernel (KeyDraws)
    _K13_BYTES_PER_EPISODE_STEP = (40 * 96 + 40 * 24) * 4 + 2 * 40 * 4

    def _k13_chunk(self, L: int, E: int) -> int:
        """Episodes per K13 pass: `episode_chunk`, clamped so that the pass's scratch fits the memory that is free NOW (plus what the
        scratch already holds) -- a part with less HBM than an MI355X, or more envs, gets more passes instead of an out-of-memory
        error in the middle of the first update."""
        chunk = min(self.episode_chunk, max(E, 1))
        if self.device.type != "cuda":
            return chunk
        held = sum(b.numel() * b.element_size() for k, b in self.scratch.bufs.items() if k[0].startswith(("k13_", "draws_")))
        free = torch.cuda.mem_get_info(self.device)[0] + held
        per_ep = L * self._K13_BYTES_PER_EPISODE_STEP * 9 // 8
        fit = int(0.85 * free) // max(per_ep, 1)
        if fit < 1:
            raise MemoryError(f"K13 needs {per_ep / 1e6:.1f} MB of scratch per episode; {free / 1e9:.2f} GB are free")
        return min(chunk, fit)

    def update_model(self, B: EpisodeBatch, draws_for=None) -> float:
        """update_model (ppo.py:1047-1148).  draws_for(iteration, slice) -> draws; default: counter hashes."""
        cell = self.agent.model
        cell.train()
        E = B.lens.shape[0]
        last = 0.0
        self.k13_particle_steps = []                                               # this update's launches only
        chunk = self._k13_chunk(B.X.shape[0], E)
        # (Producing iteration it + 1's draws -- rs_pfgru_draws, 2.3 ms of HBM writes, keys only -- on the side stream under iteration it's
        # walk was built and measured in round 4: no gain.  The backward walk holds the whole register file, so the draws can only
        # run beside the forward walk, which is VALU bound like they are.)
        for it in range(self.train_pfgru_iters):
            self.model_optimizer.zero_grad(set_to_none=True)
            tot = torch.zeros((), dtype=torch.float64, device=self.device)
            for lo in range(0, E, chunk):
                sl = slice(lo, min(lo + chunk, E))
                if draws_for is not None:
                    d = draws_for(it, sl)
                elif self.device.type == "cuda" and self.agent.fused_pfgru and getattr(self, "use_k13", True):
                    d = KeyDraws(B.key[sl] * 64 + 1 + it)                           # hashed inside K13's forward walk
                elif self.device.type == "cuda" and self.agent.fused_pfgru:
                    d = KernelDraws(B.key[sl] * 64 + 1 + it, B.chunk(sl).X.shape[0], scratch=self.scratch)
                else:
                    d = HashDraws(B.key[sl] * 64 + 1 + it, H=self.agent.rec, hid=self.agent.hid)
                if isinstance(d, (KeyDraws, KernelDraws, RecordedKernelDraws)) and getattr(self, "use_k13", True):
                    # K13: the episode loop, the loss and its back-propagation through time in one call (two launches)
                    loss, g, _ = self.model_pass_hip(B, sl, d)
                    by_name = unpack_train_grads(cell, g)
                    for name, p in cell.named_parameters():
                        p.grad = by_name[name].clone() if p.grad is None else p.grad.add_(by_name[name])
                    tot += loss
                    continue
                loss = self.model_loss(B, sl, d)
                loss.backward()
                tot += loss.detach().double()
            tot = reduce_grads_and_stats(cell.parameters(), tot.view(1))           # one collective per PFGRU iteration
            torch.nn.utils.clip_grad_norm_(cell.parameters(), 5)                    # :1137 (after the average here, see header)
            self.model_optimizer.step()
            last = tot
        cell.eval()
        return float(last.item()) if torch.is_tensor(last) else last             # one host read, after the loop

    def loc_prefetch(self, B: EpisodeBatch, its):
        """The no-grad PFGRU passes of the policy iterations `its` (an int or a list; K11, every chunk) enqueued on a side stream: they
        depend on the batch and on the PFGRU's weights only -- not on the policy being updated -- so the coming iterations' passes (VALU
        bound, the whole chip) run under the current iteration's GRU recurrence (latency bound), its head / loss kernels and the Adam step.
        Several iterations' passes go out as ONE pass over K x E particle sets (episode e's K copies adjacent, so the sets still running at
        step t stay a prefix): a pass is 30 launches of ~2 750 workgroups on 768 resident slots, and the last, partly filled round of every
        launch costs 6-8 % of it (scripts/time_k11_pass.py: 9.26 / 8.72 / 8.62 / 8.55 ms per 16 500 episodes at K = 1 / 2 / 3 / 4).
        Returns ({iteration: [loc per chunk]}, event)."""
        its = [its] if isinstance(its, int) else list(its)
        K = len(its)
        self._side = side_stream(self.device)
        main = torch.cuda.current_stream(self.device)
        if its[0] == 0:
            self._side.wait_stream(main)                                     # the batch and update_model's weights are final
        out = {it: [] for it in its}
        E = B.lens.shape[0]
        with torch.cuda.stream(self._side), torch.no_grad():
            for lo in range(0, E, self.episode_chunk):
                Bc = B.chunk(slice(lo, min(lo + self.episode_chunk, E)))
                if K == 1:
                    loc = self._pfgru_pass_hip(Bc.X, HashDraws(Bc.key * 64 + 17 + its[0]), Bc.lens_host)      # (fused_pfgru only: see update_agent)
                    loc.record_stream(main)
                    out[its[0]].append(loc)
                    continue
                cache = B.__dict__.setdefault("_xrep", {})
                ck = (lo, K)
                if ck not in cache:                                              # the K-fold batch is the same for every group of the update
                    cache[ck] = (Bc.X.repeat_interleave(K, dim=1).contiguous(),
                                 None if Bc.lens_host is None else [l for l in Bc.lens_host for _ in range(K)])
                Xr, lens_r = cache[ck]
                keys = torch.stack([Bc.key * 64 + 17 + it for it in its], dim=1).reshape(-1)
                loc = self._pfgru_pass_hip(Xr, HashDraws(keys), lens_r).view(Bc.X.shape[0], -1, K, 2)
                loc.record_stream(main)
                for k, it in enumerate(its):
                    out[it].append(loc[:, :, k])
            ev = torch.cuda.Event()
            ev.record(self._side)
        return out, ev

    def a2c_losses(self, B: EpisodeBatch, sl: slice, draws, loc: Optional[torch.Tensor] = None):
        """grad_step (:550-566) + the per-episode loss of update_rada2c (ppo.py:1191-1234) for an episode chunk.
        Returns (loss to back-propagate, stats [kl, ent, clipfrac, val_loss, loss, sum w (h loc - src)^2, sum w]).
        loc: the chunk's PFGRU location predictions when they were computed ahead (loc_prefetch)."""
        ac = self.agent
        B, sl = B.chunk(sl), slice(None)
        X, valid = B.X[:, sl], B.valid[:, sl]
        w = B.w[:, sl]
        L, E = X.shape[0], X.shape[1]
        with torch.no_grad():
            if loc is not None:
                pass
            elif isinstance(draws, RecordedKernelDraws) and X.is_cuda:
                loc = self._pfgru_pass_hip_recorded(X, draws)
            elif isinstance(draws, HashDraws) and X.is_cuda and ac.fused_pfgru:
                loc = self._pfgru_pass_hip(X, draws, B.lens_host)          # K11 with carried particle sets: one launch per step
            else:
                loc, _ = self._pfgru_pass(X[..., :3], draws, False)
        h = draws.gru_h0() if hasattr(draws, "gru_h0") else ac.gru_h0(draws.gru_h0_u())
        # the GRU over the whole (padded) episode batch in one sequence call, as grad_step does (:564): states past an episode's
        # end are computed and never used (weight 0)
        g = ac.pi.logits_net.v_net.seq_model
        if X.is_cuda and ac.hid == 24 and g.input_size == 13 and getattr(self, "use_k12", True):
            # K12: the recurrence and its back-propagation through time in one launch each
            hs = GRUSequence.apply(torch.cat((X, loc), dim=2), h, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        else:
            with torch.backends.cudnn.flags(enabled=False):    # (the library path; on the GPU MIOpen's RNN backward is slower than native)
                hs, _ = g(torch.cat((X, loc), dim=2), h.unsqueeze(0).contiguous())
        if X.is_cuda and ac.fused_policy and getattr(self, "use_k15", True):
            # K15: heads + per-sample loss + their back-propagation in one launch
            v = ac.pi.logits_net.v_net
            flat = lambda t: t.reshape(L * E).contiguous()
            loss_g, st = HeadsLoss.apply(hs.reshape(L * E, -1), v.Woms[0].weight, v.Woms[0].bias, v.Woms[2].weight, v.Woms[2].bias,
                                         v.Valms[0].weight, v.Valms[0].bias, v.Valms[2].weight, v.Valms[2].bias, self.policy_weights(),
                                         flat(B.act[:, sl]), flat(B.adv[:, sl]), flat(B.ret[:, sl]), flat(B.logp[:, sl]), flat(w),
                                         self.clip_ratio, 0.01)
            with torch.no_grad():
                d2 = (w.unsqueeze(-1) * (self.env_height * loc - B.src[:, sl]) ** 2).sum().double() / 2.0
                loss_val = -(st[4] - 0.01 * st[3] + self.alpha * st[1])             # the logged loss includes the (detached) entropy term
                stats = torch.stack([st[0], st[1], st[2], st[3], loss_val, d2, st[5]])
            return loss_g, stats
        logits, val = ac.heads(hs.reshape(L * E, -1))
        logp_all = torch.log_softmax(logits.view(L, E, -1), dim=-1)               # Categorical(logits=...) (:443-446)
        val = val.view(L, E)
        logp = logp_all.gather(-1, B.act[:, sl].unsqueeze(-1)).squeeze(-1)
        adv, ret, logp_old = B.adv[:, sl], B.ret[:, sl], B.logp[:, sl]
        ratio = torch.exp(logp - logp_old)
        clip_adv = torch.clamp(ratio, 1 - self.clip_ratio, 1 + self.clip_ratio) * adv
        surr = (w * torch.min(ratio * adv, clip_adv)).sum()
        val_loss = (w * (val - ret) ** 2).sum()
        with torch.no_grad():
            ent = (w * -(logp_all.exp() * logp_all).sum(-1)).sum()
            clipped = (ratio > 1 + self.clip_ratio) | (ratio < 1 - self.clip_ratio)
            kl = (w * (logp_old - logp)).sum()
            cf = (w * clipped.float()).sum()
            d2 = (w.unsqueeze(-1) * (self.env_height * loc - B.src[:, sl]) ** 2).sum() / 2.0
        loss = -(surr - 0.01 * val_loss + self.alpha * ent)                         # entropy: a detached float in the reference (:1216)
        stats = torch.stack([kl, ent, cf, val_loss.detach(), loss.detach(), d2, w.sum()]).double()
        return loss, stats

    def update_rada2c(self, B: EpisodeBatch, it: int = 0, draws_for=None, locs=None):
        """One call of update_rada2c: returns (stats list, terminated).  locs: loc_prefetch(B, it)[0]."""
        E = B.lens.shape[0]
        self.pi_optimizer.zero_grad(set_to_none=True)
        stats = torch.zeros(7, dtype=torch.float64, device=self.device)
        for ci, lo in enumerate(range(0, E, self.episode_chunk)):
            sl = slice(lo, min(lo + self.episode_chunk, E))
            d = draws_for(it, sl) if draws_for is not None else HashDraws(B.key[sl] * 64 + 17 + it, H=self.agent.rec, hid=self.agent.hid)
            loss, st = self.a2c_losses(B, sl, d, loc=None if locs is None else locs[ci])
            loss.backward()
            stats += st
        # mpi_avg(kl) (:1250) and mpi_avg_grads (:1256) in ONE collective: the statistics ride behind the gradients; the KL decision
        # is one host read per policy iteration (a stopped loop must not enqueue another pass over every episode)
        s = host_read(reduce_grads_and_stats(self.agent.pi.parameters(), stats))
        if s[0] < 1.5 * self.target_kl:
            self.pi_optimizer.step()
            return s, False
        return s, True

    def update_agent(self, B: EpisodeBatch) -> UpdateResult:
        """update_agent, 'rnn' branch (ppo.py:776-811)."""
        self.agent.train()
        self.k11_particle_steps = []
        model_loss = self.update_model(B)
        self.pi_optimizer.zero_grad(set_to_none=True)
        kk, term, s = 0, False, None
        ahead = self.device.type == "cuda" and self.agent.fused_pfgru and getattr(self, "use_prefetch", True)
        # the passes are prefetched one group ahead: iteration 0 alone (its chain starts at once), then `pass_batch` iterations per pass
        K = max(1, int(getattr(self, "pass_batch", 2)))        # A/B on one box, ms per iteration: 2: 1201, 3: 1205, 4: 1195-1203, 5: 1202, 8: 1221 (1: 1223)
        groups = [[0]] + [list(range(i, min(i + K, self.train_pi_iters))) for i in range(1, self.train_pi_iters, K)]
        queue = []
        if ahead:
            queue.append(self.loc_prefetch(B, groups.pop(0)))
            if groups:
                queue.append(self.loc_prefetch(B, groups.pop(0)))
        cur = None
        t_loop = time.perf_counter()
        while not term and kk < self.train_pi_iters:
            if ahead and (cur is None or kk not in cur[0]):
                cur = queue.pop(0)
                if groups:                                                       # the group after this one goes to the side stream before this iteration's own work
                    queue.append(self.loc_prefetch(B, groups.pop(0)))
                torch.cuda.current_stream(self.device).wait_event(cur[1])
            s, term = self.update_rada2c(B, kk, locs=cur[0][kk] if ahead else None)
            kk += 1
        self.policy_loop_seconds = time.perf_counter() - t_loop              # wall time of the loop (every iteration ends in a host read)
        if ahead:
            torch.cuda.current_stream(self.device).wait_stream(self._side)   # a pass enqueued for an iteration the KL stop cancelled
        self.pi_scheduler.step(); self.pfgru_scheduler.step()
        self.agent.eval()
        self.epochs_done += 1
        return UpdateResult(stop_iteration=kk, loss_policy=s[4], loss_critic=s[3], loss_predictor=model_loss, kl_divergence=s[0],
                            Entropy=s[1], ClipFrac=s[2], LocLoss=math.sqrt(max(s[5], 0.0) / max(s[6], 1e-30)))

    # ---- checkpoints: the reference saves the whole module with setup_pytorch_saver (train.py:220-223)
    def save(self, path: str) -> None:
        import os
        os.makedirs(os.path.join(path, "pyt_save"), exist_ok=True)
        torch.save(self.agent.state_dict(), os.path.join(path, "pyt_save", "model.pt"))

    def load(self, path: str) -> None:
        import os
        self.agent.load_state_dict(torch.load(os.path.join(path, "pyt_save", "model.pt"), map_location=self.device, weights_only=True))

    def resume_state(self) -> Dict[str, Any]:
        return dict(agent=self.agent.state_dict(), pi_optimizer=self.pi_optimizer.state_dict(), model_optimizer=self.model_optimizer.state_dict(),
                    pi_scheduler=self.pi_scheduler.state_dict(), pfgru_scheduler=self.pfgru_scheduler.state_dict(),
                    epochs_done=self.epochs_done, train_pfgru_iters=self.train_pfgru_iters, reduce_pfgru_iters=self.reduce_pfgru_iters)

    def load_resume_state(self, st: Dict[str, Any]) -> None:
        self.agent.load_state_dict(st["agent"])
        self.pi_optimizer.load_state_dict(st["pi_optimizer"]); self.model_optimizer.load_state_dict(st["model_optimizer"])
        self.pi_scheduler.load_state_dict(st["pi_scheduler"]); self.pfgru_scheduler.load_state_dict(st["pfgru_scheduler"])
        self.epochs_done, self.train_pfgru_iters, self.reduce_pfgru_iters = st["epochs_done"], st["train_pfgru_iters"], st["reduce_pfgru_iters"]


# ------------------------------------------------------------------------------------------------ collector
class RNNCollector:
    """The 'rnn' epoch loop of train_PPO.train for N envs at once (see the module docstring)."""

    def __init__(self, env: RadSearchVec, agents: Dict[int, RNNAgentPPO], steps_per_epoch: int, steps_per_episode: int,
                 global_critic_flag: bool = False, use_graph: bool = True):
        if global_critic_flag:
            raise Exception("No global critic option for RAD-A2C")
        self.env, self.agents = env, agents
        self.T, self.L, self.N, self.A = steps_per_epoch, steps_per_episode, env.num_envs, env.number_agents
        dev = env.device
        hid = agents[0].agent.hid
        self.buf = RolloutBuffer(self.T, self.N, self.A, _lib.RS_OBS_DIM, dev)
        self.stat = DeviceWelford((self.N, self.A), dev)
        self.steps_in_ep = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.ep_ret = torch.zeros(self.N, self.A, dtype=torch.float32, device=dev)
        self._u = torch.empty(self.N, self.A, dtype=torch.float32, device=dev)
        self._act8 = torch.empty(self.N, self.A, dtype=torch.int8, device=dev)
        self.h = torch.zeros(self.A, self.N, hid, dtype=torch.float32, device=dev)                 # GRU states
        fused_pf = torch.device(dev).type == "cuda" and all(ag.agent.fused_pfgru for ag in agents.values())
        self.bank = PredictorBank(self.N, self.A, hidden_size=agents[0].agent.rec, seed=int(env.cfg.seed),
                                  env_id_base=int(env.cfg.env_id_base), carry_hidden=True, device=dev, impl="hip" if fused_pf else "torch")
        for a, ag in agents.items():                                         # the bank evaluates the agents' own PFGRU modules
            self.bank.cells[a] = ag.agent.model
        self.episodes_begun = torch.zeros(self.N, dtype=torch.int64, device=dev)
        self._gidx = torch.arange(hid, dtype=torch.int64, device=dev)
        # one lock-step is ~45 small launches: captured once as a HIP graph and replayed (see CNNCollector); all state in place
        self.use_graph = use_graph
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._t = torch.zeros(1, dtype=torch.int64, device=dev)
        self._acc = EpochStats(self.A, dev)
        self._row_act = torch.zeros(self.N, self.A, dtype=torch.int64, device=dev)
        self._row_f = torch.zeros(3, self.N, self.A, dtype=torch.float32, device=dev)
        self._src = torch.zeros(self.N, 2, dtype=torch.float32, device=dev)
        # K14 (rs_rnn_policy_step): GRU cell + heads + draw in one launch instead of ~35 library kernels per call
        self.use_k14 = torch.device(dev).type == "cuda" and all(ag.agent.fused_policy for ag in agents.values())
        self._k_act = torch.zeros(self.A, self.N, dtype=torch.int64, device=dev)
        self._k_f = torch.zeros(self.A, 3, self.N, dtype=torch.float32, device=dev)          # logp, value, bootstrap value
        self.obs = None
        self.started = False
        self.epoch = 0
        # the lock-step's element-wise bookkeeping as three launches (rs_collect_*; ~30 torch launches before): needs K14 and the fused bank
        self.use_glue = self.use_k14 and fused_pf
        self._cs = None

    def _glue_state(self) -> "_lib.RsCollectState":
        """rs_collect_state over the collector's (fixed-address) buffers, built once the first observation exists."""
        if self._cs is None:
            env, dev, N, A = self.env, self.env.device, self.N, self.A
            self._x_buf = torch.zeros(N, A, _lib.RS_OBS_DIM, dtype=torch.float32, device=dev)
            self._xb_buf = torch.zeros(N, A, _lib.RS_OBS_DIM, dtype=torch.float32, device=dev)
            self._flags = torch.zeros(3, N, dtype=torch.uint8, device=dev)                  # over, cut, boot
            self._rew_used = torch.zeros(N, A, dtype=torch.float32, device=dev)
            self._done_oob = torch.zeros(2, N, A, dtype=torch.uint8, device=dev)            # copies of the env's done / out_of_bounds rows
            self._src_copy = torch.zeros(2, N, dtype=torch.int32, device=dev)
            self._side = side_stream(dev)
            p = lambda t: t.data_ptr()
            st = self.stat
            self._cs = _lib.RsCollectState(N, A, self.L, 0, p(env.obs), p(env.reward), p(env.team), p(env.done), p(self.obs), p(self.ep_ret),
                                           p(self.steps_in_ep), p(st.count), p(st.mean), p(st.sq), p(st.std), p(self._x_buf), p(self._xb_buf),
                                           p(self._rew_used), p(self._flags[0]), p(self._flags[1]), p(self._flags[2]), p(self.bank.episode),
                                           p(self.bank.calls), p(self.episodes_begun), p(self._t), p(env.oob), p(env.state("src_x")),
                                           p(env.state("src_y")), p(self._done_oob[0]), p(self._done_oob[1]), p(self._src_copy), None)
        return self._cs

    @torch.no_grad()
    def _step_glued(self, epoch_ended: bool) -> None:
        """_step with the bookkeeping between the library calls in rs_collect_pre / _post_step / _post_reset: 15 launches per lock-step
        (pre, uniforms, K11, K14, env step, post_step, K11 + K14 of the bootstrap round, store_rows, epoch_stats, env reset, post_reset,
        particle-set and GRU-state resets) where the torch composition took ~45.  Same arithmetic, operation by operation."""
        env, buf, N, A = self.env, self.buf, self.N, self.A
        lib, cs = _lib.load(), self._glue_state()
        st = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)
        over, cut, boot = self._flags[0], self._flags[1], self._flags[2]
        _lib.check(lib.rs_collect_pre(C.byref(cs), st), "rs_collect_pre")                       # train.py:334-341
        x = self._x_buf
        env.action_uniforms(self._u)
        loc = self.bank.predict_kernel(x)                                                      # PFGRU (K11), carried particle sets
        for a, ag in self.agents.items():
            ag.policy_step_rows(x, loc, self.h[a], self._u, a, value=self._k_f[a, 1], act=self._k_act[a], logp=self._k_f[a, 0], act8=self._act8)
        env.step(self._act8)
        _lib.check(lib.rs_collect_post_step(C.byref(cs), 1 if epoch_ended else 0, st), "rs_collect_post_step")
        # the env's reset (a latency-bound ~50 us: one env's spawn is a 15 k-instruction chain on four lanes) runs on a side stream
        # beside the bootstrap round, the buffer rows and the statistics: those read post_step's COPIES of the reward / done / out-of-
        # bounds rows and of the source positions, which the reset rewrites
        main = torch.cuda.current_stream(env.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            if epoch_ended:
                env.set_epoch_end()
            env.reset(cut)
        locb = self.bank.predict_kernel(self._xb_buf, mask8=boot)                              # train.py:462-487: one more ac.step for the value
        for a, ag in self.agents.items():
            ag.policy_step_rows(self._xb_buf, locb, self.h[a], None, a, value=self._k_f[a, 2], mask8=boot)
        _lib.check(lib.rs_store_rows(self._t.data_ptr(), self._k_act.data_ptr(), self._k_f.data_ptr(), x.data_ptr(),
                                     self._src_copy[0].data_ptr(), self._src_copy[1].data_ptr(), self._rew_used.data_ptr(), cut.data_ptr(),
                                     boot.data_ptr(), buf.act.data_ptr(), buf.logp.data_ptr(), buf.val.data_ptr(), buf.last_val.data_ptr(),
                                     buf.obs.data_ptr(), buf.source_tar.data_ptr(), buf.rew.data_ptr(), buf.cut.data_ptr(), N, A, self.T, st),
                   "rs_store_rows")
        self._acc.step_and_episodes(self._done_oob[1], self._done_oob[0], self.ep_ret, self.steps_in_ep, over.view(torch.bool))
        main.wait_stream(self._side)
        _lib.check(lib.rs_collect_post_reset(C.byref(cs), 0 if epoch_ended else 1, st), "rs_collect_post_reset")
        if not epoch_ended:                                                                    # train.py:505-518 (reset_hidden)
            self.bank.reset_kernel(cut)
            _lib.check(lib.rs_gru_h0_reset(self.h.data_ptr(), self.bank._base.data_ptr(), self.episodes_begun.data_ptr(), cut.data_ptr(),
                                           1.0 / math.sqrt(self.agents[0].agent.hid), N, A, st), "rs_gru_h0_reset")

    def _x(self, obs: torch.Tensor) -> torch.Tensor:
        x = obs.clone()
        self.stat.standardize(obs[..., 0], out=x[..., 0])
        return x

    def _reset_hidden(self, mask: Optional[torch.Tensor]) -> None:
        """reset_hidden (:583-586) for the masked envs: fresh particle sets (K11's reset kernel) and GRU h0 ~ U(-1/sqrt(hid), .)."""
        self.bank.reset(mask)
        m = torch.ones(self.N, dtype=torch.bool, device=self.h.device) if mask is None else mask.bool()
        self.episodes_begun.add_(m.long())
        if self.use_k14:                                                       # the same hash in one launch (rs_gru_h0_reset)
            m8 = m.view(torch.uint8)
            _lib.check(_lib.load().rs_gru_h0_reset(self.h.data_ptr(), self.bank._base.data_ptr(), self.episodes_begun.data_ptr(), m8.data_ptr(),
                                                   1.0 / math.sqrt(self.agents[0].agent.hid), self.N, self.A,
                                                   C.c_void_p(torch.cuda.current_stream(self.h.device).cuda_stream)), "rs_gru_h0_reset")
            return
        key = (self.bank._base * 1000003) ^ ((self.episodes_begun.view(1, -1) * 8 + 5) * _s64(0xA24BAED4963EE407))      # [A, N]
        u = hash_uniform(key.unsqueeze(-1) * 1048583 + self._gidx.view(1, 1, -1))
        h0 = self.agents[0].agent.gru_h0(u)
        self.h.copy_(torch.where(m.view(1, -1, 1), h0, self.h))

    def start(self) -> None:
        obs, *_ = self.env.reset()
        self.obs = obs.clone()                                    # from here on updated in place (fixed address)
        self.stat.update(self.obs[..., 0])
        self.started = True

    @torch.no_grad()
    def _step(self, epoch_ended: bool) -> None:
        """One lock-step of train.py:332-548 ('rnn' branches) for all envs; row self._t of the buffers is written."""
        if self.use_glue:
            return self._step_glued(epoch_ended)
        env, buf, L, N, A = self.env, self.buf, self.L, self.N, self.A
        acc, ti = self._acc, self._t
        put = lambda dst, row: dst.index_copy_(0, ti, row.unsqueeze(0))
        x = self._x(self.obs)                                                 # train.py:334-341
        env.action_uniforms(self._u)
        loc = self.bank.predict(x)                                            # PFGRU (K11), carried particle sets
        for a, ag in self.agents.items():
            if self.use_k14:
                act = self._k_act[a]
                ag.policy_step_hip(x[:, a].contiguous(), loc[:, a].contiguous(), self.h[a], u=self._u[:, a].contiguous(), h_out=self.h[a],
                                   value=self._k_f[a, 1], act=act, logp=self._k_f[a, 0])
                self._act8[:, a] = act.to(torch.int8)                         # the buffer rows are written by rs_store_rows below
                continue
            logits, v, h1 = ag.agent.policy_step(x[:, a], loc[:, a], self.h[a])
            self.h[a] = h1
            logp_all = torch.log_softmax(logits, dim=-1)
            cdf = torch.cumsum(logp_all.exp(), dim=-1)
            act = (cdf[:, :-1] <= self._u[:, a].unsqueeze(-1)).sum(dim=-1)    # inverse CDF on the env's Philox uniform (FF_core.py:101-104)
            self._row_act[:, a] = act
            self._row_f[0, :, a] = logp_all.gather(-1, act.unsqueeze(-1)).squeeze(-1)
            self._row_f[1, :, a] = v
            self._act8[:, a] = act.to(torch.int8)
        fast = self.use_k14                                                   # one store kernel per lock-step instead of ~19 launches
        if not fast:
            put(buf.act, self._row_act); put(buf.logp, self._row_f[0]); put(buf.val, self._row_f[1])
            put(buf.obs, x)
            self._src[:, 0] = env.state("src_x")[0].float()
            self._src[:, 1] = env.state("src_y")[0].float()
            put(buf.source_tar, self._src)
        next_obs, rew, team, done, info = env.step(self._act8)
        if not fast:
            put(buf.rew, rew)
        self.ep_ret += rew
        self.steps_in_ep += 1
        terminal = done.bool().any(dim=1)
        oob_now = info["out_of_bounds"]
        timeout = self.steps_in_ep == L
        episode_over = terminal | timeout
        cut = torch.ones_like(episode_over) if epoch_ended else episode_over
        if not fast:
            put(buf.cut, cut.unsqueeze(1).to(torch.uint8).expand(N, A).contiguous())
        self.stat.update(next_obs[..., 0])
        self.obs.copy_(next_obs)
        boot = cut if epoch_ended else timeout                                # train.py:462-487: one more ac.step for the value
        xb = self._x(self.obs)
        locb = self.bank.predict(xb, mask=boot)
        for a, ag in self.agents.items():
            if self.use_k14:
                ag.policy_step_hip(xb[:, a].contiguous(), locb[:, a].contiguous(), self.h[a], value=self._k_f[a, 2])
                continue
            _, vb, _ = ag.agent.policy_step(xb[:, a], locb[:, a], self.h[a])
            self._row_f[2, :, a] = torch.where(boot, vb, torch.zeros_like(vb))
        if fast:
            # PPOBuffer.store for the whole lock-step; before the reset moves the sources of the envs that start a new episode
            _lib.check(_lib.load().rs_store_rows(ti.data_ptr(), self._k_act.data_ptr(), self._k_f.data_ptr(), x.data_ptr(),
                                                 env.state("src_x").data_ptr(), env.state("src_y").data_ptr(), rew.data_ptr(),
                                                 cut.view(torch.uint8).data_ptr(), boot.view(torch.uint8).data_ptr(), buf.act.data_ptr(),
                                                 buf.logp.data_ptr(), buf.val.data_ptr(), buf.last_val.data_ptr(), buf.obs.data_ptr(),
                                                 buf.source_tar.data_ptr(), buf.rew.data_ptr(), buf.cut.data_ptr(), N, A, self.T,
                                                 C.c_void_p(torch.cuda.current_stream(self.h.device).cuda_stream)), "rs_store_rows")
        else:
            put(buf.last_val, self._row_f[2])
        acc.step_and_episodes(oob_now, done, self.ep_ret, self.steps_in_ep, episode_over)     # before the reset rewrites the env's rows
        if epoch_ended:
            env.set_epoch_end()
        self.stat.reset(cut)
        obs_r, *_ = env.reset(cut)
        self.obs.copy_(obs_r)
        self.ep_ret.masked_fill_(cut.unsqueeze(1), 0.0)
        self.steps_in_ep.masked_fill_(cut, 0)
        self.stat.update(self.obs[..., 0], mask=cut)
        if not epoch_ended:
            self._reset_hidden(cut)                                           # train.py:505-518
        ti.add_(1)

    @torch.no_grad()
    def collect(self) -> Dict[str, torch.Tensor]:
        """T - 1 replays of one captured lock-step (HIP graph, as in CNNCollector) + the epoch's last step run eagerly."""
        if not self.started:
            self.start()
        T = self.T
        self._acc.zero_()
        self._t.zero_()
        self._reset_hidden(None)                                              # train.py:322-329: every epoch starts fresh
        if self.use_k14:
            for ag in self.agents.values():
                ag.policy_weights()                                           # re-packed in place after an update
        if self.bank.impl == "hip":
            self.bank._packed()                # the captured K11 launch reads this buffer: update_model changed the PFGRU since
        if self.use_glue:
            self._glue_state()                 # its buffers exist before the lock-step is captured
        if self.use_graph and self._graph is None and T > 1:
            side = side_stream(self.env.device)                               # library warm-up (GEMM handles) outside the capture
            side.wait_stream(torch.cuda.current_stream(self.env.device))
            with torch.cuda.stream(side):
                x = self._x(self.obs)
                for a, ag in self.agents.items():
                    ag.agent.policy_step(x[:, a], torch.zeros(self.N, 2, device=x.device), self.h[a])
                self.bank._packed()
            torch.cuda.current_stream(self.env.device).wait_stream(side)
            torch.cuda.synchronize(self.env.device)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._step(False)
        for _ in range(T - 1):
            if self._graph is not None:
                self._graph.replay()
            else:
                self._step(False)
        self._step(True)
        self.buf.finish(self.agents[0].gamma, self.agents[0].lam)
        return self._acc.result()

    def resume_state(self) -> Dict[str, Any]:
        from .ppo import _welford_state
        if not self.started:
            self.start()
        return dict(env=self.env.snapshot(), stat=_welford_state(self.stat), steps_in_ep=self.steps_in_ep.clone(), ep_ret=self.ep_ret.clone(),
                    obs=self.obs.clone(), h=self.h.clone(), bank=self.bank.resume_state(), episodes_begun=self.episodes_begun.clone(),
                    epoch=self.epoch)

    def load_resume_state(self, st: Dict[str, Any]) -> None:
        from .ppo import _welford_load
        if not self.started:
            self.start()
        self.env.restore(st["env"])
        _welford_load(self.stat, st["stat"])
        self.steps_in_ep.copy_(st["steps_in_ep"]); self.ep_ret.copy_(st["ep_ret"]); self.obs.copy_(st["obs"]); self.h.copy_(st["h"])
        self.bank.load_resume_state(st["bank"])
        self.episodes_begun.copy_(st["episodes_begun"])
        self.epoch = int(st["epoch"])                                         # keys every update draw

    def update(self) -> Dict[int, UpdateResult]:
        buf = self.buf
        out = {}
        cut = buf.cut[:, :, 0]
        for a, ag in self.agents.items():
            adv = normalize_advantages(buf.adv[:, :, a])
            B = pack_episodes(buf.obs[:, :, a], buf.act[:, :, a], adv, buf.ret[:, :, a], buf.logp[:, :, a], buf.source_tar, cut,
                              n_total=self.N * _world(), env_id_base=int(self.env.cfg.env_id_base), seed=int(self.env.cfg.seed) + 7919 * a,
                              epoch=self.epoch, sort_by_length=True)
            out[a] = ag.update_agent(B)
        self.epoch += 1
        return out
