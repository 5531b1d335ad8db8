"""PFGRU location predictor, batched over envs and owners on the device.

Mirrors (paths relative to the reference root):
  PFGRUCell          <- algos/test_cnn/RADTEAM_core.py:1533-1666 (PFRNNBaseCell :1418-1531): a particle-filter GRU (Ma et al.
                        2020) with 40 particles of `hidden_size` (24) units, soft resampling (alpha 0.7), tanh activation;
                        same parameter names (fc_z, fc_r, fc_n, fc_obs, hid_obs.0, hid_obs.2), so `predictor.pt` files
                        interchange (:1654-1666).  NB the reference's `mlp([h, 24, 2], nn.ReLU)` puts a ReLU behind the LAST
                        layer too (:1574-1577), so predictions are >= 0; kept.
  PredictorBank      <- how CNNBase uses it (:1790-1795, select_action :1860-1886): every owner holds its own cell and feeds
                        the prediction into heat-map channel 0.  In the reference's CNN harness the cell is forward-only with
                        untrained weights (test_cnn/ppo.py:737-738) and the hidden state returned by select_action is never fed
                        back (test_cnn/train.py:686-693: `hidden` is only ever assigned by reset_hidden), i.e. each prediction is
                        PFGRU(observation, h0 of the episode); `carry_hidden=True` gives the evident intent instead.
                        The reference's input concatenation only type-checks for one agent (:1599-1600); with several agents
                        every owner feeds its OWN (reading, x, y) row here.

Randomness (documented deviation, like the env's): the reference draws the reparameterisation noise, the resampling indices
and h0 from torch's global CPU generator.  Here they come from a counter-based hash keyed by (seed, GLOBAL env id, owner,
episode count, prediction count, particle, unit), so a rollout does not depend on how envs are sharded over GPUs.  The cell itself takes the draws
as arguments: tests inject the reference's recorded draws and compare outputs (tests/golden/pfgru.npz).
"""
import ctypes as C
import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

_M64 = (1 << 64) - 1


def _s64(v: int) -> int:
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return torch.bitwise_right_shift(x, s) & ((1 << (64 - s)) - 1)


def hash_bits(key: torch.Tensor) -> torch.Tensor:
    """int64 keys -> 64 mixed bits: the splitmix64 finaliser (Steele et al. 2014) on wrapping int64 arithmetic
    (csrc/rs_pfgru.hip: pf_hash is the same function)."""
    x = key * _s64(0x9E3779B97F4A7C15) + _s64(0xD1B54A32D192ED03)
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def hash_uniform(key: torch.Tensor) -> torch.Tensor:
    """int64 keys -> float64 uniforms in [0, 1) from the top 53 bits."""
    return _lsr(hash_bits(key), 11).double() * (1.0 / 9007199254740992.0)


def hash_normal(key: torch.Tensor) -> torch.Tensor:
    """int64 keys -> float32 standard normals by Box-Muller in float32.  The LAST dimension of `key` enumerates consecutive units
    (key[..., u] = base + u, an even count): ONE hash per PAIR of units -- the even unit's key gives two 24-bit uniforms
    (bits 63..40 -> u1 in (0, 1], bits 39..16 -> u2 in [0, 1)), the even unit takes r cos(2 pi u2), the odd one r sin(2 pi u2).
    What the step kernel (K11) and rs_pfgru_draws evaluate per (particle, unit pair): half the hashes, logs and roots of one hash per
    unit."""
    assert key.shape[-1] % 2 == 0
    x = hash_bits(key[..., 0::2])
    u1 = (_lsr(x, 40) + 1).float() * (1.0 / 16777216.0)
    u2 = (_lsr(x, 16) & 0xFFFFFF).float() * (1.0 / 16777216.0)
    r = torch.sqrt(-2.0 * torch.log(u1))
    ang = 6.2831855 * u2
    return torch.stack((r * torch.cos(ang), r * torch.sin(ang)), dim=-1).reshape(key.shape)


# packed per-owner weights of csrc/rs_pfgru.hip (floats): offsets of the blocks and the owner stride
_PF_K = 27                      # the two [k][48] blocks are padded to 28 k rows (row 27 = 0)
_PF_ZR, _PF_ZRB, _PF_N, _PF_NB, _PF_O, _PF_OB, _PF_H0, _PF_H0B, _PF_H2, _PF_H2B, PF_WEIGHT_FLOATS = (
    0, 1344, 1392, 2736, 2784, 2811, 2816, 3392, 3416, 3464, 3472)


def pack_weights(cells) -> torch.Tensor:
    """[A, PF_WEIGHT_FLOATS] float32: zr_t [28][48] | zr_b | n_t [28][48] (columns permuted, see below) | n_b | o_w [27] | o_b | h0_t [24][24] | h0_b | h2_w [2][24] | h2_b."""
    rows = []
    for c in cells:
        assert c.h_dim == 24 and c.num_particles == 40 and c.input_size == 3, "rs_pfgru_step is built for 40 particles x 24 units"
        w = torch.zeros(PF_WEIGHT_FLOATS, dtype=torch.float32, device=c.fc_z.weight.device)
        w[_PF_ZR:_PF_ZR + _PF_K * 48] = torch.cat([c.fc_z.weight, c.fc_r.weight], 0).t().reshape(-1)
        w[_PF_ZRB:_PF_N] = torch.cat([c.fc_z.bias, c.fc_r.bias], 0)
        # fc_n's 48 outputs (mu | var) in the order the kernel consumes them: three 16-column chunks of [mu(8j .. 8j+7) | var(8j .. 8j+7)],
        # so that a chunk completes eight units and its accumulators are the only part of the product that is live
        perm = torch.tensor([(8 * (cc // 16) + cc % 16) if cc % 16 < 8 else (24 + 8 * (cc // 16) + cc % 16 - 8) for cc in range(48)],
                            device=c.fc_n.weight.device)
        w[_PF_N:_PF_N + _PF_K * 48] = c.fc_n.weight[perm].t().reshape(-1)
        w[_PF_NB:_PF_O] = c.fc_n.bias[perm]
        w[_PF_O:_PF_OB] = c.fc_obs.weight.reshape(-1)
        w[_PF_OB] = c.fc_obs.bias[0]
        w[_PF_H0:_PF_H0B] = c.hid_obs[0].weight.t().reshape(-1)
        w[_PF_H0B:_PF_H2] = c.hid_obs[0].bias
        w[_PF_H2:_PF_H2B] = c.hid_obs[2].weight.reshape(-1)
        w[_PF_H2B:_PF_H2B + 2] = c.hid_obs[2].bias
        rows.append(w)
    return torch.stack(rows).contiguous()


class _Linear3D(torch.autograd.Function):
    """y = x W^T + b for x [B, P, K] with a weight gradient that is reduced in two steps: per-b partial products [B, K, O] (one
    batched GEMM) and a sum over b.  autograd's own dW = x.reshape(B * P, K)^T @ g.reshape(B * P, O) is a tall-skinny GEMM with a
    reduction length of B * P ~ 1e5-1e6 and a 27 x 48 output, for which the BLAS library takes ~240 us (30x its memory time): three
    of those per PFGRU step were 44 % of the GPU time of a RAD-A2C update (profiles/r02_rnn_update_kernel_stats.csv)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = torch.bmm(g.transpose(1, 2), x).sum(dim=0) if ctx.needs_input_grad[1] else None
        gb = g.sum(dim=(0, 1)) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _lin(layer: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    return _Linear3D.apply(x, layer.weight, layer.bias) if (x.dim() == 3 and torch.is_grad_enabled()) else layer(x)


class PFGRUCell(nn.Module):
    """Batched PFGRUCell.  h [B, P, H] particles, p [B, P] log weights (B = envs, or owners x envs)."""

    def __init__(self, input_size: int = 3, obs_size: int = 3, activation: str = "tanh", use_resampling: bool = True,
                 num_particles: int = 40, hidden_size: int = 24, resamp_alpha: float = 0.7):
        super().__init__()
        if activation != "tanh":
            raise NotImplementedError("the reference instantiates the predictor with tanh only (RADTEAM_core.py:1790-1795)")
        self.num_particles, self.input_size, self.h_dim = num_particles, input_size, hidden_size
        self.resamp_alpha, self.use_resampling = resamp_alpha, use_resampling
        self.fc_z = nn.Linear(hidden_size + input_size, hidden_size)
        self.fc_r = nn.Linear(hidden_size + input_size, hidden_size)
        self.fc_n = nn.Linear(hidden_size + input_size, hidden_size * 2)
        self.fc_obs = nn.Linear(hidden_size + input_size, 1)
        self.hid_obs = nn.Sequential(nn.Linear(hidden_size, 24), nn.ReLU(), nn.Linear(24, 2), nn.ReLU())    # :1574-1584

    def particle_predictions(self, h: torch.Tensor) -> torch.Tensor:
        """hid_obs applied to every particle, [B, P, H] -> [B, P, 2] (update_model's particle_pred, ppo.py:1079)."""
        return torch.relu(_lin(self.hid_obs[2], torch.relu(_lin(self.hid_obs[0], h))))

    def init_hidden(self, batch: int, u: Optional[torch.Tensor] = None, device=None):
        """init_hidden (:1643-1652): h0 ~ U[0,1) (`u` [B, P, H] when the draws are supplied), p0 = log(1 / P)."""
        dev = device if device is not None else self.fc_z.weight.device
        h0 = u.to(torch.float32) if u is not None else torch.rand(batch, self.num_particles, self.h_dim, device=dev)
        p0 = torch.full((batch, self.num_particles), math.log(1.0 / self.num_particles), dtype=torch.float32, device=dev)
        return h0, p0

    def forward(self, obs: torch.Tensor, hx: Tuple[torch.Tensor, torch.Tensor], eps: torch.Tensor, resample_u: Optional[torch.Tensor] = None,
                resample_idx: Optional[torch.Tensor] = None):
        """One step (:1586-1631).  obs [B, in]; eps [B, P, H] standard normals of the reparameterisation (:1517-1530);
        resampling either from uniforms `resample_u` [B, P] (inverse CDF of the soft-resampling distribution) or from given
        particle indices `resample_idx` [B, P] (what torch.multinomial returned in the reference, :1485-1496).
        Returns loc_pred [B, 2], (h1 [B, P, H], p1 [B, P])."""
        h0, p0 = hx
        B, P, H = h0.shape
        obs_in = obs.unsqueeze(1).expand(B, P, obs.shape[-1])
        obs_cat = torch.cat((h0, obs_in), dim=2)
        z = torch.sigmoid(_lin(self.fc_z, obs_cat))
        r = torch.sigmoid(_lin(self.fc_r, obs_cat))
        n_1 = _lin(self.fc_n, torch.cat((r * h0, obs_in), dim=2))
        mu_n, var_n = torch.split(n_1, H, dim=2)
        n = torch.tanh(mu_n + eps * F.softplus(var_n))
        h1 = (1 - z) * n + z * h0
        p1 = F.log_softmax(_lin(self.fc_obs, torch.cat((h1, obs_in), dim=2)).squeeze(-1) + p0, dim=1)   # :1633-1641
        if self.use_resampling:                                                                          # :1466-1515
            a = self.resamp_alpha
            if resample_idx is None:
                resamp_prob = a * torch.exp(p1) + (1 - a) / P
                cdf = torch.cumsum(resamp_prob.double(), dim=1)
                cdf = cdf / cdf[:, -1:]
                resample_idx = torch.searchsorted(cdf, resample_u.double().contiguous(), right=True).clamp_(max=P - 1)
            h1 = torch.gather(h1, 1, resample_idx.unsqueeze(-1).expand(B, P, H))
            prob_new = torch.exp(torch.gather(p1, 1, resample_idx))
            prob_new = prob_new / (a * prob_new + (1 - a) / P)
            prob_new = torch.log(prob_new)
            p1 = prob_new - torch.logsumexp(prob_new, dim=1, keepdim=True)
        mean_hid = torch.sum(torch.exp(p1).unsqueeze(-1) * h1, dim=1)
        return self.hid_obs(mean_hid), (h1, p1)


class PredictorBank:
    """One PFGRUCell per owner, evaluated for all envs at once; what feeds heat-map channel 0 (`rs_maps_update`'s pred)."""

    def __init__(self, num_envs: int, number_agents: int, hidden_size: int = 24, seed: int = 0, env_id_base: int = 0,
                 carry_hidden: bool = False, device="cuda:0", impl: str = "hip"):
        """impl: "hip" = the fused step / reset kernels (csrc/rs_pfgru.hip through the C ABI; the product path, cuda only);
        "torch" = the same arithmetic composed from torch ops (what the kernel is tested against; also runs on the CPU)."""
        assert impl in ("hip", "torch")
        self.N, self.A, self.dev = num_envs, number_agents, torch.device(device)
        self.impl = impl
        if impl == "hip":
            if self.dev.type != "cuda":
                raise RuntimeError("PredictorBank(impl='hip') needs a cuda device; there is no CPU fallback for the product path")
            if hidden_size != 24:
                raise NotImplementedError("rs_pfgru_step is built for the reference's 24 hidden units (RADTEAM_core.py:1790-1795)")
            self._lib = _lib.load()
            self._pred = torch.zeros(num_envs, number_agents, 2, dtype=torch.float32, device=self.dev)
        self.cells = [PFGRUCell(hidden_size=hidden_size).to(self.dev) for _ in range(number_agents)]
        self.carry_hidden = carry_hidden
        P, H = self.cells[0].num_particles, hidden_size
        self.P, self.H = P, H
        env = torch.arange(num_envs, dtype=torch.int64, device=self.dev) + int(env_id_base)
        own = torch.arange(number_agents, dtype=torch.int64, device=self.dev)
        # key layout: seed | global env id | owner, then (episode, step, kind) and (particle, unit) are mixed in per call
        self._base = hash_uniform((env.view(1, -1) * 64 + own.view(-1, 1)) ^ _s64(int(seed) * 0x2545F4914F6CDD1D)).mul(2.0 ** 52).long()   # [A, N]
        self._pu = (torch.arange(P, dtype=torch.int64, device=self.dev).view(P, 1) * 4096
                    + torch.arange(H, dtype=torch.int64, device=self.dev).view(1, H))                                                          # [P, H]
        # the kernels keep the particle sets quad-major ([A, N, H / 4, P, 4], include/radsearch.h); `h` is the logical [A, N, P, H] view of it
        self._hq = torch.zeros(number_agents, num_envs, H // 4, P, 4, dtype=torch.float32, device=self.dev) if impl == "hip" else None
        self._h = None if impl == "hip" else torch.zeros(number_agents, num_envs, P, H, dtype=torch.float32, device=self.dev)
        self.p = torch.full((number_agents, num_envs, P), math.log(1.0 / P), dtype=torch.float32, device=self.dev)
        # draw counters per env: episodes begun, predictions made in the current episode
        self.episode = torch.zeros(num_envs, dtype=torch.int64, device=self.dev)
        self.calls = torch.zeros(num_envs, dtype=torch.int64, device=self.dev)

    @staticmethod
    def to_quads(h: torch.Tensor) -> torch.Tensor:
        """[..., P, H] particle sets -> the kernels' quad-major [..., H / 4, P, 4] (contiguous)."""
        *lead, P, H = h.shape
        return h.reshape(*lead, P, H // 4, 4).transpose(-3, -2).contiguous()

    @staticmethod
    def from_quads(hq: torch.Tensor) -> torch.Tensor:
        """the kernels' quad-major [..., H / 4, P, 4] -> [..., P, H]."""
        *lead, Q, P, four = hq.shape
        return hq.transpose(-3, -2).reshape(*lead, P, Q * four)

    @property
    def h(self) -> torch.Tensor:
        """[A, N, P, H] particle sets (impl "hip": a copy out of the quad-major device storage)."""
        return self._h if self._hq is None else self.from_quads(self._hq)

    @h.setter
    def h(self, value: torch.Tensor) -> None:
        if self._hq is None:
            self._h = value
        else:
            self._hq.copy_(self.to_quads(value))

    def parameters(self, a: int):
        return self.cells[a].parameters()

    def _key(self, kind: int) -> torch.Tensor:
        ctr = (self.episode.view(1, -1) * 100003 + self.calls.view(1, -1)) * 8 + kind                       # [1, N]
        return (self._base * 1000003) ^ (ctr * _s64(0xA24BAED4963EE407))                                   # [A, N]

    @torch.no_grad()
    def reset(self, mask: Optional[torch.Tensor] = None) -> None:
        """reset_hidden (RADTEAM_core.py:2030-2033) for the masked envs: a new episode, fresh h0 ~ U[0,1), p0 = log(1/P)."""
        m1 = torch.ones(self.N, dtype=torch.bool, device=self.dev) if mask is None else mask.bool()
        self.episode.add_(m1.long())                      # counters change in place: a captured collector step refers to them
        self.calls.masked_fill_(m1, 0)
        if self.impl == "hip":
            m8 = None if mask is None else m1.to(torch.uint8)
            with _lib.timed("rs_pfgru_reset"):
                _lib.check(self._lib.rs_pfgru_reset(self._hq.data_ptr(), self.p.data_ptr(), self._base.data_ptr(), self.episode.data_ptr(),
                                                    self.calls.data_ptr(), None if m8 is None else m8.data_ptr(), self.N, self.A,
                                                    self._stream()), "rs_pfgru_reset")
            return
        k = self._key(0)
        u = hash_uniform(k.view(self.A, self.N, 1, 1) * 1048583 + self._pu.view(1, 1, self.P, self.H)).float()
        m = m1.view(1, self.N, 1, 1)
        self.h = torch.where(m, u, self.h)
        self.p = torch.where(m.view(1, self.N, 1), torch.full_like(self.p, math.log(1.0 / self.P)), self.p)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    # the two kernels alone, for the collectors whose bookkeeping kernels (rs_collect_post_step / _post_reset) keep the draw counters:
    # `episode` / `calls` are NOT touched here, `mask8` is a uint8 tensor or None, the prediction comes back as the bank's own buffer
    def predict_kernel(self, obs: torch.Tensor, mask8: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert self.impl == "hip" and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape == (self.N, self.A, _lib.RS_OBS_DIM)
        with _lib.timed("rs_pfgru_step"):
            _lib.check(self._lib.rs_pfgru_step(self._packed().data_ptr(), obs.data_ptr(), self._hq.data_ptr(), self.p.data_ptr(),
                                               self._base.data_ptr(), self.episode.data_ptr(), self.calls.data_ptr(),
                                               None if mask8 is None else mask8.data_ptr(), 1 if self.carry_hidden else 0,
                                               float(self.cells[0].resamp_alpha), self._pred.data_ptr(), self.N, self.A, self._stream()),
                       "rs_pfgru_step")
        return self._pred

    def reset_kernel(self, mask8: Optional[torch.Tensor] = None) -> None:
        assert self.impl == "hip"
        with _lib.timed("rs_pfgru_reset"):
            _lib.check(self._lib.rs_pfgru_reset(self._hq.data_ptr(), self.p.data_ptr(), self._base.data_ptr(), self.episode.data_ptr(),
                                                self.calls.data_ptr(), None if mask8 is None else mask8.data_ptr(), self.N, self.A,
                                                self._stream()), "rs_pfgru_reset")

    def _packed(self) -> torch.Tensor:
        ver = tuple(p._version for c in self.cells for p in c.parameters())
        if getattr(self, "_pack_ver", None) != ver:
            with torch.no_grad():
                w = pack_weights(self.cells)
                if getattr(self, "_wpack", None) is None:
                    self._wpack = w
                else:
                    self._wpack.copy_(w)                  # same address: a captured collector step keeps reading it
            self._pack_ver = ver
        return self._wpack

    def _stacked(self):
        """The owners' weights stacked along a leading owner axis ([A, in, out] for bmm); rebuilt when a cell's parameters change
        (load_state_dict / an optimiser step bump the tensors' version counters)."""
        ver = tuple(p._version for c in self.cells for p in c.parameters())
        if getattr(self, "_stack_ver", None) != ver:
            g = lambda f: torch.stack([f(c) for c in self.cells])
            self._W = dict(
                zr=g(lambda c: torch.cat([c.fc_z.weight, c.fc_r.weight], 0).t().contiguous()),     # [A, H+3, 2H]
                zr_b=g(lambda c: torch.cat([c.fc_z.bias, c.fc_r.bias], 0)).unsqueeze(1),            # [A, 1, 2H]
                n=g(lambda c: c.fc_n.weight.t().contiguous()), n_b=g(lambda c: c.fc_n.bias).unsqueeze(1),
                o=g(lambda c: c.fc_obs.weight.t().contiguous()), o_b=g(lambda c: c.fc_obs.bias).unsqueeze(1),
                h0=g(lambda c: c.hid_obs[0].weight.t().contiguous()), h0_b=g(lambda c: c.hid_obs[0].bias).unsqueeze(1),
                h2=g(lambda c: c.hid_obs[2].weight.t().contiguous()), h2_b=g(lambda c: c.hid_obs[2].bias).unsqueeze(1))
            self._stack_ver = ver
        return self._W

    @torch.no_grad()
    def predict(self, obs: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """select_action's `self.model(obs_tensor, hidden)` (:1872-1879) for every owner and env: obs [N, A, 11] -> pred [N, A, 2]
        (scaled coordinates).  `mask`: the envs this round counts for (bootstrap rounds); the others' rows are to be discarded.
        All owners go through the same batched matrix products (PFGRUCell.forward's arithmetic, owner axis first)."""
        A, N, P, H = self.A, self.N, self.P, self.H
        if self.impl == "hip":
            assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape == (N, A, _lib.RS_OBS_DIM)
            m8 = None if mask is None else mask.to(torch.uint8)
            with _lib.timed("rs_pfgru_step"):
                _lib.check(self._lib.rs_pfgru_step(self._packed().data_ptr(), obs.data_ptr(), self._hq.data_ptr(), self.p.data_ptr(),
                                                   self._base.data_ptr(), self.episode.data_ptr(), self.calls.data_ptr(),
                                                   None if m8 is None else m8.data_ptr(), 1 if self.carry_hidden else 0,
                                                   float(self.cells[0].resamp_alpha), self._pred.data_ptr(), N, A, self._stream()),
                           "rs_pfgru_step")
            self.calls.add_(1 if mask is None else mask.long())
            return self._pred.clone()
        W = self._stacked()
        k_eps, k_res = self._key(1), self._key(2)                                                           # [A, N]
        eps = hash_normal(k_eps.view(A, N, 1, 1) * 1048583 + self._pu.view(1, 1, P, H))                     # [A, N, P, H]
        ru = hash_uniform(k_res.view(A, N, 1) * 1048583 + self._pu[:, 0].view(1, 1, P))                     # [A, N, P]
        h0, p0 = self.h, self.p
        x = obs[:, :, :3].permute(1, 0, 2).unsqueeze(2).expand(A, N, P, 3)                                  # owner a feeds its own row
        cat = torch.cat((h0, x), dim=3).reshape(A, N * P, H + 3)
        zr = torch.sigmoid(torch.baddbmm(W["zr_b"], cat, W["zr"])).view(A, N, P, 2 * H)
        z, r = zr[..., :H], zr[..., H:]
        n_1 = torch.baddbmm(W["n_b"], torch.cat((r * h0, x), dim=3).reshape(A, N * P, H + 3), W["n"]).view(A, N, P, 2 * H)
        n = torch.tanh(n_1[..., :H] + eps * F.softplus(n_1[..., H:]))
        h1 = (1 - z) * n + z * h0
        logit = torch.baddbmm(W["o_b"], torch.cat((h1, x), dim=3).reshape(A, N * P, H + 3), W["o"]).view(A, N, P)
        p1 = F.log_softmax(logit + p0, dim=2)
        al = self.cells[0].resamp_alpha
        cdf = torch.cumsum((al * torch.exp(p1) + (1 - al) / P).double(), dim=2)
        cdf = cdf / cdf[..., -1:]
        idx = torch.searchsorted(cdf.view(A * N, P), ru.view(A * N, P).contiguous(), right=True).clamp_(max=P - 1).view(A, N, P)
        if getattr(self, "record_margin", False):
            # test hook: how close each (owner, env) came to another resampling index -- the smallest distance between one of its
            # uniforms and a CDF value.  A float32 rounding difference in the weights can only move an index where this is ~1e-7.
            self.last_margin = (cdf.unsqueeze(2) - ru.unsqueeze(3)).abs().amin(dim=(2, 3))                   # [A, N]
        h1 = torch.gather(h1, 2, idx.unsqueeze(-1).expand(A, N, P, H))
        pn = torch.exp(torch.gather(p1, 2, idx))
        pn = torch.log(pn / (al * pn + (1 - al) / P))
        p1 = pn - torch.logsumexp(pn, dim=2, keepdim=True)
        mean_hid = torch.sum(torch.exp(p1).unsqueeze(-1) * h1, dim=2)                                       # [A, N, H]
        loc = torch.relu(torch.baddbmm(W["h2_b"], torch.relu(torch.baddbmm(W["h0_b"], mean_hid, W["h0"])), W["h2"]))
        if self.carry_hidden:
            if mask is None:
                self.h, self.p = h1, p1
            else:
                m = mask.bool()
                self.h = torch.where(m.view(1, N, 1, 1), h1, self.h)
                self.p = torch.where(m.view(1, N, 1), p1, self.p)
        self.calls.add_(1 if mask is None else mask.long())
        return loc.permute(1, 0, 2).contiguous()

    def resume_state(self):
        """Particle sets and draw counters (the cells' weights are saved with the agents)."""
        return dict(h=self.h.clone(), p=self.p.clone(), episode=self.episode.clone(), calls=self.calls.clone())

    def load_resume_state(self, st) -> None:
        if self.impl == "hip":                                     # in place: a captured collector step refers to these tensors
            self.h = st["h"]; self.p.copy_(st["p"])
        else:
            self.h, self.p = st["h"].clone(), st["p"].clone()
        self.episode.copy_(st["episode"]); self.calls.copy_(st["calls"])

    def state_dict(self, a: int):
        return self.cells[a].state_dict()

    def load_state_dict(self, a: int, sd) -> None:
        self.cells[a].load_state_dict(sd)
