"""RAD-TEAM heat maps and CNN actor / critic on the device.

HeatMaps       <- MapsBuffer + ConversionTools (NeuralNetworkCores/RADTEAM_core.py:366-932) for N envs x A agents,
                  backed by the K5 kernels (rs_maps_*); constants are computed on the host exactly as the reference
                  does (calculate_resolution_accuracy :70-71, calculate_map_dimensions :61-67, CNNBase.__post_init__
                  :1727-1738, Normalizer.normalize_incremental_logscale :321-362).
CNNActor/Critic <- Actor :935-1180 / Critic :1183-1345 with the same layer names (state_dicts interchange), batched:
                  Flatten(start_dim=1) instead of the reference's batch-1 Flatten(start_dim=0).
"""
import ctypes as C
import os
import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .envs import RadSearchVec


def calculate_resolution_accuracy(resolution_multiplier: float, scale: float) -> float:
    return resolution_multiplier * 1 / scale


def calculate_map_dimensions(grid_bounds, resolution_accuracy: float, offset: float) -> Tuple[int, int]:
    return (int(grid_bounds[0] * resolution_accuracy) + int(offset * resolution_accuracy),
            int(grid_bounds[1] * resolution_accuracy) + int(offset * resolution_accuracy))


def heat_map_geometry(env: RadSearchVec, steps_per_episode: int, enforce_boundaries: bool, resolution_multiplier: float = 0.01,
                      bounds_offset=(200.0, 500.0), grid_bounds=(1, 1), detector_step_size: float = 100.0):
    """(resolution_accuracy, scaled_offset, map_dimensions) as CNNBase.__post_init__ derives them (RADTEAM_core.py:1727-1738): 27 x 27
    cells with enforced walls, 147 x 147 without (the detectors may then leave the search area by up to steps_per_episode steps)."""
    scale = 1 / float(env.cfg.bbox[3] - env.cfg.observation_area[1])                # env.scale (rad_search_env.py:435)
    ra = calculate_resolution_accuracy(resolution_multiplier, scale)
    if enforce_boundaries:
        off = scale * max(bounds_offset)
    else:
        off = scale * (max(bounds_offset) + steps_per_episode * detector_step_size)
    return ra, off, calculate_map_dimensions(grid_bounds, ra, off)


def actor_stack_from(shared: torch.Tensor, cells: torch.Tensor, pcells: torch.Tensor, a: int) -> torch.Tensor:
    """CNNBase.get_map_stack (RADTEAM_core.py:1791-1836) for owner a from the stored shared maps [B, 4, X, Y] and the owners' location
    / prediction cells [B, A]: the dense [B, 6, X, Y] actor input {prediction, location, others, readings, visits, obstacles}."""
    B, _, X, Y = shared.shape
    loc = torch.zeros(B, X * Y, dtype=torch.float32, device=shared.device)
    lc = cells[:, a:a + 1]
    loc.scatter_(1, lc.clamp(min=0), (lc >= 0).float())             # -1: no position recorded yet (fresh maps)
    pm = torch.zeros(B, X * Y, dtype=torch.float32, device=shared.device)
    pc = pcells[:, a:a + 1]
    pm.scatter_(1, pc.clamp(min=0), (pc >= 0).float())
    loc = loc.view(B, 1, X, Y)
    return torch.cat([pm.view(B, 1, X, Y), loc, shared[:, 0:1] - loc, shared[:, 1:4]], dim=1)


class HeatMaps:
    def __init__(self, env: RadSearchVec, steps_per_episode: int, resolution_multiplier: float = 0.01,
                 bounds_offset=(200.0, 500.0), grid_bounds=(1, 1), enforce_boundaries: bool = True,
                 detector_step_size: float = 100.0):
        self.lib = _lib.load()
        self.env = env
        self.N, self.A, self.L = env.num_envs, env.number_agents, steps_per_episode
        self.resolution_accuracy, self.scaled_offset, self.map_dimensions = heat_map_geometry(
            env, steps_per_episode, enforce_boundaries, resolution_multiplier, bounds_offset, grid_bounds, detector_step_size)
        X, Y = self.map_dimensions
        base = (steps_per_episode + 1) * self.A                                      # :498
        tab = [(math.log(2 + 2 * c, base)) * 1 / math.log(2 * base, base) for c in range(base + 1)]
        dev = env.device
        self.visit_table = torch.tensor(tab, dtype=torch.float64).to(torch.float32).to(dev)
        nbytes = self.lib.rs_maps_state_bytes(self.N, self.A, self.L, X, Y)
        if nbytes == 0:
            raise ValueError("invalid heat-map configuration")
        self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        self._ws_base = self._ws.data_ptr() + (-self._ws.data_ptr()) % 256
        self._ws_span = (self._ws_base - self._ws.data_ptr(), nbytes)
        h = C.c_void_p()
        _lib.check(self.lib.rs_maps_create(self.N, self.A, self.L, X, Y, float(self.resolution_accuracy),
                                           self.visit_table.data_ptr(), self._ws_base, nbytes, env._stream(), C.byref(h)),
                   "rs_maps_create")
        self._h = h
        self.actor_stack = torch.zeros(self.N, self.A, 6, X, Y, dtype=torch.float32, device=dev)
        self.critic_stack = torch.zeros(self.N, 4, X, Y, dtype=torch.float32, device=dev)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.rs_maps_destroy(h)
            self._h = None

    def snapshot(self) -> torch.Tensor:
        """A copy of the heat-map state of every env (maps, reading rings, cells, Welford state): what a resumed run restores."""
        o, n = self._ws_span
        return self._ws[o:o + n].clone()

    def restore(self, ws: torch.Tensor) -> None:
        o, n = self._ws_span
        if ws.numel() != n:
            raise ValueError("HeatMaps.restore: the snapshot belongs to another configuration")
        self._ws[o:o + n].copy_(ws)

    def field(self, name: str) -> torch.Tensor:
        p, e, r, c = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.rs_maps_field(self._h, name.encode(), C.byref(p), C.byref(e), C.byref(r), C.byref(c)), "rs_maps_field")
        off = p.value - self._ws.data_ptr()
        raw = self._ws[off:off + e.value * r.value * c.value]
        dt = torch.float32 if name in ("combined", "readings", "visits", "obstacles") else (torch.int16 if e.value == 2 else torch.int32)
        return raw.view(dt).view(r.value, c.value)

    def reset(self, mask: Optional[torch.Tensor] = None) -> None:
        """MapsBuffer.reset (RADTEAM_core.py:510-523) for the masked envs."""
        if mask is not None and mask.dtype == torch.bool:
            mask = mask.to(torch.uint8)
        self._mk = mask
        _lib.check(self.lib.rs_maps_reset(self._h, None if mask is None else mask.data_ptr(), self.env._stream()), "rs_maps_reset")

    def update(self, obs: torch.Tensor, pred: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None) -> None:
        """MapsBuffer.observation_to_map (RADTEAM_core.py:532-616) for every (masked) env."""
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape == (self.N, self.A, _lib.RS_OBS_DIM)
        if mask is not None and mask.dtype == torch.bool:
            mask = mask.to(torch.uint8)
        self._mk2 = mask
        _lib.check(self.lib.rs_maps_update(self._h, self.env._h, obs.data_ptr(), None if pred is None else pred.data_ptr(),
                                           None if mask is None else mask.data_ptr(), self.env._stream()), "rs_maps_update")

    def stacks(self):
        """CNNBase.get_map_stack (:1791-1836): (actor [N,A,6,X,Y], critic [N,4,X,Y]) views of internal buffers."""
        _lib.check(self.lib.rs_maps_stack(self._h, self.actor_stack.data_ptr(), self.critic_stack.data_ptr(), self.env._stream()),
                   "rs_maps_stack")
        return self.actor_stack, self.critic_stack

    def shared_maps(self) -> torch.Tensor:
        """The 4 maps every owner shares, [N,4,X,Y] = {combined, readings, visits, obstacles} (= the critic stack);
        the per-owner actor stacks are formed inside the CNN trunk kernel from these + the owners' cells."""
        _lib.check(self.lib.rs_maps_stack(self._h, None, self.critic_stack.data_ptr(), self.env._stream()), "rs_maps_stack")
        return self.critic_stack


class ConvTrunk(torch.autograd.Function):
    """conv-ReLU-maxpool-conv-ReLU-flatten of the RAD-TEAM CNNs (RADTEAM_core.py:962-1023, :1211-1271) on the HIP
    kernels rs_cnn_trunk_forward / rs_cnn_trunk_backward, fed by the resident shared maps (no stack in HBM).
    agent >= 0: the owner's 6-channel actor input; agent < 0: the 4-channel critic input."""

    @staticmethod
    def forward(ctx, maps, cells, pcells, agent, w1, b1, w2, b2, grad_mode=True):
        lib = _lib.load()
        S = maps.shape[0]
        assert maps.dtype == torch.float32 and maps.is_contiguous() and maps[0].numel() == 4 * 729
        A = 0
        if agent >= 0:
            assert cells.dtype == torch.int64 and pcells.dtype == torch.int64 and cells.is_contiguous() and pcells.is_contiguous()
            assert cells.shape == pcells.shape and cells.shape[0] == S
            A = cells.shape[1]
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        assert w1c.shape == (8, 6 if agent >= 0 else 4, 3, 3) and w2c.shape == (16, 8, 3, 3)
        # grad_mode = torch.is_grad_enabled() of the CALLER (inside forward() autograd is always off): the collectors run
        # under no_grad and must not pay for the activations only backward reads
        train = bool(grad_mode) and any(ctx.needs_input_grad[4:8])
        a2 = torch.empty(S, 2704, dtype=torch.float32, device=maps.device)
        p1 = torch.empty(S, 169, 8, dtype=torch.float32, device=maps.device) if train else None
        amax = torch.empty(S, 169, 8, dtype=torch.uint8, device=maps.device) if train else None
        mask = torch.empty(S, 169, dtype=torch.int16, device=maps.device) if train else None
        stream = torch.cuda.current_stream(maps.device).cuda_stream
        wt = torch.empty(lib.rs_cnn_trunk_scratch_floats(6 if agent >= 0 else 4), dtype=torch.float32, device=maps.device)
        with _lib.timed("rs_cnn_trunk_forward_train" if train else "rs_cnn_trunk_forward"):
            _lib.check(lib.rs_cnn_trunk_forward(maps.data_ptr(), cells.data_ptr() if agent >= 0 else None,
                                                pcells.data_ptr() if agent >= 0 else None, A, agent, S, w1c.data_ptr(), b1c.data_ptr(),
                                                w2c.data_ptr(), b2c.data_ptr(), a2.data_ptr(), p1.data_ptr() if train else None,
                                                amax.data_ptr() if train else None, mask.data_ptr() if train else None, wt.data_ptr(),
                                                stream), "rs_cnn_trunk_forward")
        if train:
            ctx.save_for_backward(maps, cells if agent >= 0 else maps, pcells if agent >= 0 else maps, w2c, mask, p1, amax)
            ctx.agent, ctx.A, ctx.cin = agent, A, (6 if agent >= 0 else 4)
        return a2

    @staticmethod
    def backward(ctx, da2):
        lib = _lib.load()
        maps, cells, pcells, w2c, mask, p1, amax = ctx.saved_tensors
        S, cin, agent = maps.shape[0], ctx.cin, ctx.agent
        da2 = da2.contiguous()
        rows, row = lib.rs_cnn_trunk_slab_rows(S, cin), lib.rs_cnn_trunk_slab_row(cin)
        slab = torch.empty(rows, row, dtype=torch.float32, device=maps.device)
        stream = torch.cuda.current_stream(maps.device).cuda_stream
        wt = torch.empty(lib.rs_cnn_trunk_scratch_floats(cin), dtype=torch.float32, device=maps.device)
        with _lib.timed("rs_cnn_trunk_backward"):
            _lib.check(lib.rs_cnn_trunk_backward(maps.data_ptr(), cells.data_ptr() if agent >= 0 else None,
                                                 pcells.data_ptr() if agent >= 0 else None, ctx.A, agent, S, w2c.data_ptr(),
                                                 da2.data_ptr(), mask.data_ptr(), p1.data_ptr(), amax.data_ptr(), slab.data_ptr(),
                                                 wt.data_ptr(), stream), "rs_cnn_trunk_backward")
        g = slab.sum(dim=0)
        n1 = 8 * cin * 9
        return (None, None, None, None, g[:n1].view(8, cin, 3, 3), g[n1:n1 + 8], g[n1 + 8:n1 + 8 + 1152].view(16, 8, 3, 3),
                g[n1 + 8 + 1152:], None)


class _LinearTall(torch.autograd.Function):
    """y = x W^T + b for a tall x [S, in] (the Linear layers behind the trunk: in = 2704 / 32 / 16): the weight gradient is reduced
    in two steps (partial products over 1024-row slabs -- 4096 for the wide first layer -- in one batched GEMM, then a sum), because
    autograd's g^T @ x is a GEMM with a 16 x 32 output and a reduction length of S = 524 288, for which the BLAS library needs
    0.3-0.9 ms -- 30x its memory time (profiles/r02_final_kernel_stats.csv: the MT32x32x256 / MT16x16x512 kernels).  Config 4's
    update: 18.0 -> 16.5 s."""
    SLAB = 1024

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        if x.shape[1] > 32 and x.is_cuda:
            # the wide first layer as a batched GEMM over 4096-row slabs: 1.04 ms instead of 1.20 ms per 524 288 x 2704 chunk
            # (5.45 TB/s on the activations; scripts/micro/gemm_shapes.py)
            S, R = x.shape[0], 4 * _LinearTall.SLAB
            wt = w.t().contiguous()
            return torch.bmm(x.view(S // R, R, -1), wt.unsqueeze(0).expand(S // R, -1, -1)).view(S, -1) + b
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        S = x.shape[0]
        R = _LinearTall.SLAB if x.shape[1] <= 32 else 4 * _LinearTall.SLAB
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(g.view(S // R, R, -1).transpose(1, 2), x.view(S // R, R, -1)).sum(dim=0)
        gb = g.sum(dim=0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _head(layer: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if (isinstance(layer, nn.Linear) and x.dim() == 2 and x.shape[0] >= 65536 and x.shape[0] % (4 * _LinearTall.SLAB) == 0
            and torch.is_grad_enabled()):
        return _LinearTall.apply(x, layer.weight, layer.bias)
    return layer(x)


class CNNActor(nn.Module):
    """RADTEAM_core.Actor (:935-1180): conv3x3(6->8)-ReLU-maxpool2-conv3x3(8->16)-ReLU-flatten-32-16-8 softmax."""

    def __init__(self, map_dim=(27, 27), batches: int = 1, map_count: int = 6, action_dim: int = 8):
        super().__init__()
        pool_output = int(((map_dim[0] - 2) / 2) + 1)
        self.actor = nn.Sequential(
            nn.Conv2d(map_count, 8, kernel_size=3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Conv2d(8, 16, kernel_size=3, padding=1, stride=1), nn.ReLU(), nn.Flatten(start_dim=1),
            nn.Linear(16 * pool_output * pool_output, 32), nn.ReLU(), nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, action_dim),
            nn.Softmax(dim=-1))

    def logits(self, x):
        """Dense [B,6,X,Y] input through the nn.Sequential (library convolutions): the plain-PyTorch reference of
        logits_from_maps, used by tests."""
        for layer in list(self.actor)[:-1]:
            x = layer(x)
        return x

    def logits_from_maps(self, maps, cells, pcells, agent: int):
        """Owner `agent`'s logits for samples described by the resident shared maps [S,4,X,Y] + cell indices [S,A]:
        HIP trunk (ConvTrunk) + the three Linear layers.  The trunk kernels K9 / K10 hold one 27 x 27 image in LDS (the walls-enforced
        size every CLI of the reference trains with, main.py:311-316); any other map size -- 147 x 147 without enforced walls,
        RADTEAM_core.py:1727-1738 -- takes the dense stack through the library convolutions (the same nn.Sequential)."""
        if tuple(maps.shape[-2:]) != (27, 27):
            with torch.backends.cudnn.flags(enabled=False):         # the native convolution: no MIOpen solver search for a one-off shape
                return self.logits(actor_stack_from(maps, cells, pcells, agent))
        a = self.actor
        x = ConvTrunk.apply(maps, cells, pcells, agent, a[0].weight, a[0].bias, a[3].weight, a[3].bias, torch.is_grad_enabled())
        for layer in list(a)[6:-1]:
            x = _head(layer, x)
        return x

    def forward(self, x):
        return self.actor(x)


class CNNCritic(nn.Module):
    """RADTEAM_core.Critic (:1183-1345): same trunk on 4 channels, Linear(16) -> 1."""

    def __init__(self, map_dim=(27, 27), batches: int = 1, map_count: int = 4):
        super().__init__()
        pool_output = int(((map_dim[0] - 2) / 2) + 1)
        self.critic = nn.Sequential(
            nn.Conv2d(map_count, 8, kernel_size=3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2),
            nn.Conv2d(8, 16, kernel_size=3, padding=1, stride=1), nn.ReLU(), nn.Flatten(start_dim=1),
            nn.Linear(16 * pool_output * pool_output, 32), nn.ReLU(), nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 1))

    def forward(self, x):
        """Dense [B,4,X,Y] input through the nn.Sequential (library convolutions): the plain-PyTorch reference."""
        return self.critic(x).squeeze(-1)

    def value_from_maps(self, maps):
        """V for samples given as resident shared maps [S,4,X,Y]: HIP trunk (ConvTrunk) + the Linear layers (27 x 27 maps; other sizes
        through the library convolutions, see CNNActor.logits_from_maps)."""
        if tuple(maps.shape[-2:]) != (27, 27):
            with torch.backends.cudnn.flags(enabled=False):
                return self.critic(maps).squeeze(-1)
        c = self.critic
        x = ConvTrunk.apply(maps, None, None, -1, c[0].weight, c[0].bias, c[3].weight, c[3].bias, torch.is_grad_enabled())
        for layer in list(c)[6:]:
            x = _head(layer, x)
        return x.squeeze(-1)
