"""Host side of the env boundary.

RadSearchVec  -- N lock-step environments on one MI355X; tensors in, tensors out, stream ordered.
RadSearch     -- drop-in for the reference's gym env (gym_rad_search/gym_rad_search/envs/
                 rad_search_env.py:304-437): same constructor fields, `step(action)` / `reset()`
                 returning the reference's 4-tuple of dicts (:443-445, :723-728), same attributes
                 (search_area, observation_space.shape, scale, src_coords, epoch_end, ...).  It is a
                 1-env RadSearchVec underneath; there is no CPU path.

PyTorch is plumbing here (device memory + streams); all env arithmetic runs in the HIP kernels of
librs_hip.so through the C ABI (include/radsearch.h).
"""
import ctypes as C
import math
import sys
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib

A_SIZE = 9                    # rad_search_env.py:68
DETECTABLE_DIRECTIONS = 8     # :69
DET_STEP = 100.0              # :70


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class RadSearchVec:
    """N independent RadSearch environments advanced in lock-step by the HIP kernels.

    step(actions[N,A] int8) -> (obs f32[N,A,11], reward f32[N,A], team f32[N], done bool[N,A], info)
    reset(mask=None)        -> same tuple; rows of envs outside the mask keep their previous content.
    All outputs are views of buffers owned by this object and are overwritten by the next call.
    """

    def __init__(self, num_envs: int, number_agents: int = 1, obstruction_count: int = 0,
                 enforce_grid_boundaries: bool = False,
                 bbox=((0.0, 0.0), (2700.0, 0.0), (2700.0, 2700.0), (0.0, 2700.0)),
                 observation_area=(200.0, 500.0), seed: int = 0, env_id_base: int = 0,
                 falloff: str = "reference", geom_group_size: int = 1, device: Union[str, torch.device] = "cuda:0",
                 coord_noise: bool = False, DEBUG: bool = False):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("RadSearchVec runs on the MI355X only (device must be cuda:*); there is no CPU path")
        if falloff not in ("reference", "inverse_square"):
            raise ValueError("falloff must be 'reference' (I/r, rad_search_env.py:501) or 'inverse_square'")
        self.num_envs, self.number_agents = int(num_envs), int(number_agents)
        xs = [int(p[0]) for p in bbox]
        ys = [int(p[1]) for p in bbox]
        self.cfg = _lib.RsConfig(
            num_envs=self.num_envs, num_agents=self.number_agents, obstruction_count=int(obstruction_count),
            enforce_grid_boundaries=int(bool(enforce_grid_boundaries)),
            bbox=(min(xs), min(ys), max(xs), max(ys)),
            observation_area=(int(observation_area[0]), int(observation_area[1])),
            falloff=0 if falloff == "reference" else 1, geom_group_size=int(geom_group_size),
            seed=int(seed) & 0xFFFFFFFF, env_id_base=int(env_id_base) & 0xFFFFFFFF,
            coord_noise=int(bool(coord_noise)), debug_spawn=int(bool(DEBUG)))
        nbytes = self.lib.rs_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise ValueError("invalid RadSearch configuration")
        N, A = self.num_envs, self.number_agents
        with torch.cuda.device(self.device):
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self._ws.data_ptr()) % 256
            self._ws_base = self._ws.data_ptr() + off
            self._ws_off = off
            self.obs = torch.zeros(N, A, _lib.RS_OBS_DIM, dtype=torch.float32, device=self.device)
            self.reward = torch.zeros(N, A, dtype=torch.float32, device=self.device)
            self.team = torch.zeros(N, dtype=torch.float32, device=self.device)
            self.done = torch.zeros(N, A, dtype=torch.uint8, device=self.device)
            self.oob = torch.zeros(N, A, dtype=torch.uint8, device=self.device)
            self.oob_count = torch.zeros(N, A, dtype=torch.int32, device=self.device)
            self.blocked = torch.zeros(N, A, dtype=torch.uint8, device=self.device)
            self.collision = torch.zeros(N, A, dtype=torch.uint8, device=self.device)
            self._info = _lib.RsInfo(self.oob.data_ptr(), self.oob_count.data_ptr(), self.blocked.data_ptr(),
                                     self.collision.data_ptr())
            h = C.c_void_p()
            _lib.check(self.lib.rs_create(C.byref(self.cfg), self._ws_base, nbytes, self._stream(), C.byref(h)), "rs_create")
            self._h = h
        self._fields: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ plumbing
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.rs_destroy(h)
            self._h = None

    def state(self, name: str) -> torch.Tensor:
        """Tensor view of one SoA state field inside the workspace (tests / adapters)."""
        if name not in self._fields:
            p, e, r, c = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
            _lib.check(self.lib.rs_state_field(self._h, name.encode(), C.byref(p), C.byref(e), C.byref(r), C.byref(c)),
                       f"rs_state_field({name})")
            off = p.value - self._ws.data_ptr()
            nb = e.value * r.value * c.value
            raw = self._ws[off:off + nb]
            dt = {("x", 4): torch.int32}.get((name, e.value))
            if dt is None:
                if e.value == 8:
                    dt = torch.float64
                elif e.value == 1:
                    dt = torch.uint8
                else:
                    dt = torch.int32
            self._fields[name] = raw.view(dt).view(r.value, c.value)
        return self._fields[name]

    def _outs(self):
        info = {"out_of_bounds": self.oob, "out_of_bounds_count": self.oob_count, "blocked": self.blocked,
                "collision": self.collision}
        return self.obs, self.reward, self.team, self.done, info

    _OUTS = ("obs", "reward", "team", "done", "oob", "oob_count", "blocked", "collision")

    def snapshot(self) -> Dict[str, torch.Tensor]:
        """A copy of the whole env state (the handle's workspace: positions, sources, rectangles, cached geodesics, Philox
        counters, latches) and of the output rows -- what a resumed run restores (train_PPO.load)."""
        st = {"ws": self._ws[self._ws_off:self._ws_off + self.lib.rs_state_bytes(C.byref(self.cfg))].clone()}
        st.update({k: getattr(self, k).clone() for k in self._OUTS})
        return st

    def restore(self, st: Dict[str, torch.Tensor]) -> None:
        """Inverse of snapshot() for an env of the same configuration (in place: captured graphs keep their addresses)."""
        n = self.lib.rs_state_bytes(C.byref(self.cfg))
        if st["ws"].numel() != n:
            raise ValueError("RadSearchVec.restore: the snapshot belongs to another configuration")
        self._ws[self._ws_off:self._ws_off + n].copy_(st["ws"])
        for k in self._OUTS:
            getattr(self, k).copy_(st[k])

    # ------------------------------------------------------------------ API
    def set_epoch_end(self) -> None:
        """`env.epoch_end = True` for every env (algos/multiagent/train.py:482-484)."""
        _lib.check(self.lib.rs_set_epoch_end(self._h, self._stream()), "rs_set_epoch_end")

    def reset(self, mask: Optional[torch.Tensor] = None):
        if mask is not None:
            if mask.dtype == torch.bool:
                mask = mask.to(torch.uint8)
            assert mask.dtype == torch.uint8 and mask.numel() == self.num_envs and mask.is_contiguous()
            self._mask_keepalive = mask
        _lib.check(self.lib.rs_reset(self._h, _ptr(mask), self.obs.data_ptr(), self.reward.data_ptr(),
                                     self.team.data_ptr(), self.done.data_ptr(), C.byref(self._info), self._stream()),
                   "rs_reset")
        return self._outs()

    def step(self, actions: torch.Tensor):
        assert actions.dtype == torch.int8 and actions.numel() == self.num_envs * self.number_agents
        assert actions.is_contiguous() and actions.device == self.device
        _lib.check(self.lib.rs_step(self._h, actions.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                                    self.team.data_ptr(), self.done.data_ptr(), C.byref(self._info), self._stream()),
                   "rs_step")
        return self._outs()

    def refresh(self, src_xy: torch.Tensor, det_xy: torch.Tensor, intensity: torch.Tensor, bkg: torch.Tensor,
                num_obs: Optional[torch.Tensor] = None, rects: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None):
        """RadSearch.refresh_environment (rad_search_env.py:799-874) for the masked envs: episodes start from saved
        parameters.  src_xy, det_xy [N,2] int32; intensity, bkg [N] int32; num_obs [N] + rects [N,7,4] int32 (x0,y0,x1,y1)
        replace the obstacle layouts (both None: keep them)."""
        N = self.num_envs
        for t, shape in ((src_xy, (N, 2)), (det_xy, (N, 2)), (intensity, (N,)), (bkg, (N,))):
            assert t.dtype == torch.int32 and tuple(t.shape) == shape and t.is_contiguous() and t.device == self.device
        if num_obs is not None:
            assert rects is not None and num_obs.dtype == torch.int32 and rects.dtype == torch.int32
            assert tuple(num_obs.shape) == (N,) and tuple(rects.shape) == (N, _lib.RS_MAX_OBS, 4) and rects.is_contiguous()
        if mask is not None:
            if mask.dtype == torch.bool:
                mask = mask.to(torch.uint8)
            assert mask.dtype == torch.uint8 and mask.numel() == N and mask.is_contiguous()
        self._refresh_keepalive = (src_xy, det_xy, intensity, bkg, num_obs, rects, mask)
        _lib.check(self.lib.rs_refresh(self._h, _ptr(mask), src_xy.data_ptr(), det_xy.data_ptr(), intensity.data_ptr(),
                                       bkg.data_ptr(), _ptr(num_obs), _ptr(rects), self.obs.data_ptr(), self.reward.data_ptr(),
                                       self.team.data_ptr(), self.done.data_ptr(), C.byref(self._info), self._stream()),
                   "rs_refresh")
        return self._outs()

    def action_uniforms(self, out: torch.Tensor) -> torch.Tensor:
        """u[N,A] in [0,1) for inverse-CDF action sampling, from each env's own Philox stream (rs_action_uniforms)."""
        assert out.dtype == torch.float32 and out.numel() == self.num_envs * self.number_agents and out.is_contiguous()
        _lib.check(self.lib.rs_action_uniforms(self._h, out.data_ptr(), self._stream()), "rs_action_uniforms")
        return out

    def error_flags(self) -> int:
        f = C.c_uint32(0)
        _lib.check(self.lib.rs_error_flags(self._h, self._stream(), C.byref(f)), "rs_error_flags")
        return int(f.value)


def gae(rew: torch.Tensor, val: torch.Tensor, cut: torch.Tensor, last_val: torch.Tensor, gamma: float, lam: float,
        adv: Optional[torch.Tensor] = None, ret: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """GAE(lambda) + rewards-to-go for a time-major [T, M] buffer on the GPU (rs_gae; ppo.py:391-423)."""
    lib = _lib.load()
    T, M = rew.shape[0], rew[0].numel()
    for t in (rew, val, last_val):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    assert cut.dtype == torch.uint8 and cut.is_contiguous()
    adv = torch.empty_like(rew) if adv is None else adv
    ret = torch.empty_like(rew) if ret is None else ret
    _lib.check(lib.rs_gae(rew.data_ptr(), val.data_ptr(), cut.data_ptr(), last_val.data_ptr(), adv.data_ptr(),
                          ret.data_ptr(), T, M, float(gamma), float(lam),
                          torch.cuda.current_stream(rew.device).cuda_stream), "rs_gae")
    return adv, ret


class RadSearch:
    """Drop-in for gym_rad_search's RadSearch (rad_search_env.py:304-797), backed by one GPU env.

    Differences (documented in DESIGN.md): randomness comes from the library's Philox stream keyed by
    `seed` (an int, or derived from np_random) instead of the numpy Generator's own sequence;
    render()/FIM_step are not part of the hot path and raise NotImplementedError."""

    metadata = {"render.modes": ["human"], "video.frames_per_second": 5}
    step_size = DET_STEP
    number_actions = A_SIZE
    detectable_directions = DETECTABLE_DIRECTIONS
    continuous = False

    def __init__(self, bbox=((0.0, 0.0), (2700.0, 0.0), (2700.0, 2700.0), (0.0, 2700.0)),
                 observation_area=(200.0, 500.0), np_random: Optional[np.random.Generator] = None,
                 obstruction_count: int = 0, enforce_grid_boundaries: bool = False, save_gif: bool = False,
                 number_agents: int = 1, DEBUG: bool = False, seed: Optional[int] = None,
                 device: Union[str, torch.device] = "cuda:0", coord_noise: bool = False, **unused: Any):
        if seed is None:
            rng = np_random if np_random is not None else np.random.default_rng(0)
            seed = int(rng.integers(0, 2 ** 32))
        self.bbox = tuple(tuple(p) for p in bbox)
        self.observation_area = tuple(observation_area)
        self.obstruction_count = obstruction_count
        self.enforce_grid_boundaries = enforce_grid_boundaries
        self.number_agents = number_agents
        self.save_gif = save_gif
        self.np_random = np_random
        lo, hi = observation_area
        b = self.bbox
        # rad_search_env.py:393-420
        self.search_area = ((b[0][0] + lo, b[0][1] + lo), (b[1][0] - hi, b[1][1] + lo),
                            (b[2][0] - hi, b[2][1] - hi), (b[3][0] + lo, b[3][1] - hi))
        self.max_dist = math.dist(self.search_area[2], self.search_area[1])
        assert self.max_dist > 1000, "Maximum distance available is too small, unable to spawn source and detector 1000 cm apart"
        self.scale = 1 / self.search_area[2][1]
        self.scaled_grid_max = (1, 1)
        self.observation_space = SimpleNamespace(shape=(11,), dtype=np.float32, low=0, high=np.inf)
        self.action_space = SimpleNamespace(n=A_SIZE)
        self.background_radiation_bounds = (10, 51)
        self.radiation_intensity_bounds = (1e6, 10e6)
        self.coord_noise = bool(coord_noise)                    # rad_search_env.py:365
        self.DEBUG = bool(DEBUG)                                # :387-389
        self.epoch_cnt = 0
        self._vec = RadSearchVec(1, number_agents, obstruction_count, enforce_grid_boundaries, bbox, observation_area,
                                 seed=seed, device=device, coord_noise=coord_noise, DEBUG=DEBUG)
        self.epoch_end = True
        self.reset()

    # attribute surface read by the reference's callers
    @property
    def src_coords(self):
        return (float(self._vec.state("src_x")[0, 0].item()), float(self._vec.state("src_y")[0, 0].item()))

    @property
    def intensity(self) -> int:
        return int(self._vec.state("intensity")[0, 0].item())

    @property
    def bkg_intensity(self) -> int:
        return int(self._vec.state("bkg")[0, 0].item())

    @property
    def done(self) -> bool:
        return bool(self._vec.state("done")[0, 0].item())

    @property
    def iter_count(self) -> int:
        return int(self._vec.state("iter_count")[0, 0].item())

    @property
    def num_obs(self) -> int:
        return int(self._vec.state("num_obs")[0, 0].item())

    def get_agent_outOfBounds_count(self, id: int) -> int:
        return int(self._vec.state("oob_count")[id, 0].item())

    def _tuple(self, outs):
        obs, rew, team, done, info = outs
        A = self.number_agents
        obs_c = obs[0].double().cpu().numpy()
        rew_c = rew[0].double().cpu().numpy()
        done_c = done[0].cpu().numpy()
        oob, oobc, blk = (info[k][0].cpu().numpy() for k in ("out_of_bounds", "out_of_bounds_count", "blocked"))
        flags = self._vec.error_flags()
        if flags & _lib.ENVERR_BAD_ACTION:
            raise AssertionError("action out of range")
        if flags & _lib.ENVERR_IDLE_STALL:
            raise ValueError("Agent should not return false if the tentative step is an idle step")
        if flags & _lib.ENVERR_ZERO_DIST:
            raise ValueError("lam value too large")     # numpy's poisson(inf) error in the reference
        if flags & _lib.ENVERR_NO_PATH:
            raise RuntimeError("no obstacle-free path from the detector to the source")
        observation = {i: obs_c[i] for i in range(A)}
        reward = {"team_reward": float(np.float32(team[0].item())),
                  "individual_reward": {i: float(np.float32(rew_c[i])) for i in range(A)}}
        # rewards are 2-decimal values stored as float32; report the float64 nearest to the decimal
        reward["team_reward"] = round(reward["team_reward"], 2)
        reward["individual_reward"] = {i: round(v, 2) for i, v in reward["individual_reward"].items()}
        terminal = {i: bool(done_c[i]) for i in range(A)}
        infos = {i: {"out_of_bounds": bool(oob[i]), "out_of_bounds_count": int(oobc[i]), "blocked": bool(blk[i]),
                     "scale": 1 / self.search_area[2][1]} for i in range(A)}
        return observation, reward, terminal, infos

    def step(self, action: Optional[Union[int, Dict[int, int]]] = None):
        """rad_search_env.py:443-728.  dict -> every agent its own action, collision rule applied (:645-659); int -> the
        same action for every agent WITHOUT the collision rule (:676-690; -1 == idle 8); None (or an empty dict) -> no
        move, a fresh measurement at the current position (:528-567)."""
        assert action is None or type(action) == int or type(action) == dict, "Action not integer or a dictionary of actions."
        A = self.number_agents
        if type(action) is int:
            if action == -1:
                action = 8
            assert 0 <= action <= 8
            if A > 1:
                print("WARNING: Passing single action to mutliple agents during step!", file=sys.stderr)
            acts = [16 + action] * A                    # C ABI: 16 + a = the single-int form
        elif type(action) is dict and action:
            for i, a in action.items():
                assert 0 <= action[i] <= 8
            acts = [action[i] for i in range(A)]
        else:
            acts = [9] * A                              # C ABI: 9 = step(None)
        a_t = torch.tensor(acts, dtype=torch.int8, device=self._vec.device).view(1, A)
        return self._tuple(self._vec.step(a_t))

    def reset(self):
        if self.epoch_end:
            self._vec.set_epoch_end()
            self.epoch_cnt += 1
            self.epoch_end = False
        return self._tuple(self._vec.reset())

    def render(self, *a, **k):
        raise NotImplementedError("rendering (rad_search_env.py:1308-1762) is outside the hot path")

    def refresh_environment(self, env_dict, id, num_obs=0):
        """rad_search_env.py:799-874: load saved episode parameters `env_dict["env_<id>"] = (src_coords, det_coords,
        intensity, bkg_intensity[, obstacles])` (format of algos/test_environment/eval/test_env_gen.py:13-24; each
        obstacle is `[array of its 4 corner points]`) and return the observation of the idle step."""
        e = env_dict["env_" + str(id)]
        dev = self._vec.device

        def grid(p):
            q = [float(p[0]), float(p[1])]
            if q[0] != int(q[0]) or q[1] != int(q[1]):
                raise ValueError("saved coordinates must lie on the 1 cm lattice the environment samples from")
            return [int(q[0]), int(q[1])]
        src = torch.tensor([grid(e[0])], dtype=torch.int32, device=dev)
        det = torch.tensor([grid(e[1])], dtype=torch.int32, device=dev)
        inten = torch.tensor([int(e[2])], dtype=torch.int32, device=dev)
        bkg = torch.tensor([int(e[3])], dtype=torch.int32, device=dev)
        nob = rects = None
        if num_obs > 0:                                                   # :829-858 (the stored list decides the count)
            obstacles = e[4]
            if len(obstacles) > _lib.RS_MAX_OBS:
                raise ValueError("more than 7 obstructions")
            r = np.zeros((1, _lib.RS_MAX_OBS, 4), dtype=np.int32)
            for i, ob in enumerate(obstacles):
                pts = np.asarray(ob[0], dtype=np.float64)
                xs, ys = pts[:, 0], pts[:, 1]
                if not (np.all(xs == np.round(xs)) and np.all(ys == np.round(ys))):
                    raise ValueError("saved obstruction corners must lie on the 1 cm lattice")
                r[0, i] = (int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max()))
            nob = torch.tensor([len(obstacles)], dtype=torch.int32, device=dev)
            rects = torch.from_numpy(r).to(dev)
        self.epoch_end = False
        observation, _, _, _ = self._tuple(self._vec.refresh(src, det, inten, bkg, nob, rects))
        return observation
