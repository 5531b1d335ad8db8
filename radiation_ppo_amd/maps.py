"""RAD-TEAM heat maps and CNN actor / critic on the device.

HeatMaps       <- MapsBuffer + ConversionTools (NeuralNetworkCores/RADTEAM_core.py:366-932) for N envs x A agents,
                  backed by the K5 kernels (rs_maps_*); constants are computed on the host exactly as the reference
                  does (calculate_resolution_accuracy :70-71, calculate_map_dimensions :61-67, CNNBase.__post_init__
                  :1727-1738, Normalizer.normalize_incremental_logscale :321-362).
CNNActor/Critic <- Actor :935-1180 / Critic :1183-1345 with the same layer names (state_dicts interchange), batched:
                  Flatten(start_dim=1) instead of the reference's batch-1 Flatten(start_dim=0).
"""
import ctypes as C
import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .envs import RadSearchVec


def calculate_resolution_accuracy(resolution_multiplier: float, scale: float) -> float:
    return resolution_multiplier * 1 / scale


def calculate_map_dimensions(grid_bounds, resolution_accuracy: float, offset: float) -> Tuple[int, int]:
    return (int(grid_bounds[0] * resolution_accuracy) + int(offset * resolution_accuracy),
            int(grid_bounds[1] * resolution_accuracy) + int(offset * resolution_accuracy))


class HeatMaps:
    def __init__(self, env: RadSearchVec, steps_per_episode: int, resolution_multiplier: float = 0.01,
                 bounds_offset=(200.0, 500.0), grid_bounds=(1, 1), enforce_boundaries: bool = True,
                 detector_step_size: float = 100.0):
        self.lib = _lib.load()
        self.env = env
        self.N, self.A, self.L = env.num_envs, env.number_agents, steps_per_episode
        scale = 1 / float(env.cfg.bbox[3] - env.cfg.observation_area[1])            # env.scale (rad_search_env.py:435)
        self.resolution_accuracy = calculate_resolution_accuracy(resolution_multiplier, scale)
        if enforce_boundaries:                                                       # RADTEAM_core.py:1727-1738
            self.scaled_offset = scale * max(bounds_offset)
        else:
            self.scaled_offset = scale * (max(bounds_offset) + steps_per_episode * detector_step_size)
        self.map_dimensions = calculate_map_dimensions(grid_bounds, self.resolution_accuracy, self.scaled_offset)
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
        for layer in list(self.actor)[:-1]:
            x = layer(x)
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
        return self.critic(x).squeeze(-1)
