"""ctypes binding of librs_hip.so (the C ABI in include/radsearch.h).

torch is imported first on purpose: PyTorch-ROCm ships its own libamdhip64.so.7; loading it before
our library makes the dynamic loader resolve our DT_NEEDED entry to the SAME runtime instance, so
device pointers and streams created by torch are valid in our launches."""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RS_LIB_PATH") or os.path.join(_PKG, "lib", "librs_hip.so")   # override: A/B builds

RS_OBS_DIM = 11
RS_MAX_AGENTS = 8
RS_MAX_OBS = 7

ENVERR_ZERO_DIST = 1
ENVERR_IDLE_STALL = 2
ENVERR_CORRECT_CAP = 4
ENVERR_BAD_ACTION = 8
ENVERR_NO_PATH = 16


class RsConfig(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("num_agents", C.c_int32), ("obstruction_count", C.c_int32),
        ("enforce_grid_boundaries", C.c_int32), ("bbox", C.c_int32 * 4), ("observation_area", C.c_int32 * 2),
        ("falloff", C.c_int32), ("geom_group_size", C.c_int32), ("seed", C.c_uint32), ("env_id_base", C.c_uint32),
        ("coord_noise", C.c_int32), ("debug_spawn", C.c_int32),
    ]


class RsCollectState(C.Structure):
    """rs_collect_state (include/radsearch.h)."""
    _fields_ = [("num_envs", C.c_int32), ("num_agents", C.c_int32), ("steps_per_episode", C.c_int32), ("team_reward", C.c_int32)] + [
        (n, C.c_void_p) for n in ("env_obs", "env_reward", "env_team", "env_done", "obs", "ep_ret", "steps_in_ep", "w_count", "w_mean", "w_sq",
                                  "w_std", "x", "xb", "reward_used", "over", "cut", "boot", "pf_episode", "pf_calls", "episodes_begun", "t",
                                  "env_oob", "env_src_x", "env_src_y", "done_copy", "oob_copy", "src_copy", "complete_len")]


class RsMlpParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3")]


class RsRolloutArgs(C.Structure):
    _fields_ = [("steps_per_epoch", C.c_int32), ("steps_per_episode", C.c_int32)] + [
        (n, C.c_void_p) for n in ("obs", "act", "logp", "val", "rew", "last_val", "cut", "source_tar", "cur_obs", "w_count",
                                  "w_mean", "w_sq", "w_std", "steps_in_ep", "ep_ret", "done_count", "oob_count",
                                  "ep_ret_sum", "ep_len_sum", "ep_count", "ep_ret_sq_sum", "ep_ret_max", "ep_ret_min")]


class RsPpoBatch(C.Structure):
    _fields_ = [("x", C.c_void_p), ("act", C.c_void_p), ("adv", C.c_void_p), ("ret", C.c_void_p), ("logp_old", C.c_void_p),
                ("w", C.c_void_p), ("M", C.c_int32), ("clip_ratio", C.c_float), ("alpha", C.c_float), ("vf_coef", C.c_float)]


class RsInfo(C.Structure):
    _fields_ = [("out_of_bounds", C.c_void_p), ("out_of_bounds_count", C.c_void_p), ("blocked", C.c_void_p),
                ("collision", C.c_void_p)]


# every symbol include/radsearch.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("rs_strerror", C.c_char_p, [C.c_int]),
    ("rs_abi_version", C.c_int, []),
    ("rs_state_bytes", C.c_size_t, [C.POINTER(RsConfig)]),
    ("rs_create", C.c_int, [C.POINTER(RsConfig), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("rs_destroy", None, [C.c_void_p]),
    ("rs_state_field", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("rs_set_epoch_end", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rs_reset", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.POINTER(RsInfo), C.c_void_p]),
    ("rs_refresh", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RsInfo), C.c_void_p]),
    ("rs_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.POINTER(RsInfo), C.c_void_p]),
    ("rs_action_uniforms", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_error_flags", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("rs_policy_forward", C.c_int, [C.POINTER(RsMlpParams), C.POINTER(RsMlpParams), C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    ("rs_rollout", C.c_int, [C.c_void_p, C.POINTER(RsMlpParams), C.POINTER(RsMlpParams), C.POINTER(RsRolloutArgs), C.c_void_p]),
    ("rs_ppo_grad_workspace_bytes", C.c_size_t, []),
    ("rs_ppo_grad", C.c_int, [C.POINTER(RsMlpParams), C.POINTER(RsMlpParams), C.POINTER(RsPpoBatch), C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_adam_step", C.c_int, [C.POINTER(RsMlpParams), C.POINTER(RsMlpParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    ("rs_maps_state_bytes", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("rs_maps_create", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                 C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("rs_maps_destroy", None, [C.c_void_p]),
    ("rs_maps_reset", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_maps_update", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_maps_stack", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_maps_field", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int32)]),
    ("rs_cnn_trunk_slab_row", C.c_int32, [C.c_int32]),
    ("rs_cnn_trunk_slab_rows", C.c_int32, [C.c_int64, C.c_int32]),
    ("rs_cnn_trunk_scratch_floats", C.c_int32, [C.c_int32]),
    ("rs_cnn_trunk_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_cnn_trunk_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_gae", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                         C.c_int32, C.c_double, C.c_double, C.c_void_p]),
    ("rs_pfgru_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_pfgru_step_recorded", C.c_int, [C.c_void_p] * 7 + [C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_pfgru_pass", C.c_int, [C.c_void_p] * 7 + [C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_pfgru_reset", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_void_p]),
    ("rs_pfgru_draws", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_pfgru_train", C.c_int, [C.c_void_p] * 15 + [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    ("rs_pfgru_train_keyed", C.c_int, [C.c_void_p] * 14 + [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    ("rs_rnn_policy_step", C.c_int, [C.c_void_p] * 10 + [C.c_int32, C.c_void_p]),
    ("rs_rnn_policy_step_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    ("rs_collect_pre", C.c_int, [C.POINTER(RsCollectState), C.c_void_p]),
    ("rs_collect_post_step", C.c_int, [C.POINTER(RsCollectState), C.c_int32, C.c_void_p]),
    ("rs_collect_post_reset", C.c_int, [C.POINTER(RsCollectState), C.c_int32, C.c_void_p]),
    ("rs_welford_update", C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_welford_reset", C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_welford_standardize", C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    ("rs_gru_h0_reset", C.c_int, [C.c_void_p] * 4 + [C.c_double, C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_epoch_stats", C.c_int, [C.c_void_p] * 13 + [C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_a2c_heads_loss", C.c_int, [C.c_void_p] * 11 + [C.c_int64, C.c_double, C.c_double, C.c_void_p]),
    ("rs_actor_loss", C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_double, C.c_void_p]),
    ("rs_cnn_trunk_prepare", C.c_int, [C.c_int32] + [C.c_void_p] * 6),
    ("rs_cnn_trunk_infer", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rs_cnn_head", C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                              C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    ("rs_store_rows", C.c_int, [C.c_void_p] * 17 + [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_gru_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("rs_gru_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_int32, C.c_void_p]),
]

_lib = None

# Optional kernel timing for bench.py: when EVENTS is a dict, every library call wrapped in `timed(name)` is bracketed by
# HIP events on the stream it is launched on (torch's current stream) and the pair is appended to EVENTS[name].
EVENTS = None


class timed:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        self.ev = None
        if EVENTS is not None and not torch.cuda.is_current_stream_capturing():     # no timing events inside a graph capture
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
            EVENTS.setdefault(self.name, []).append(self.ev)
        return False


def load():
    """Load librs_hip.so (once).  Fails loudly: there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `python radiation_ppo_amd/build.py`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.rs_abi_version() != 4:
        raise RuntimeError("librs_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().rs_strerror(code).decode()
        raise RuntimeError(f"librs_hip {what}: {msg} (code {code})")
