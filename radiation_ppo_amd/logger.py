"""EpochLogger -- the reference's SpinningUp-style logger (algos/multiagent/rl_tools/epoch_logger.py:110-403) behind
the same API, so `progress.txt` / `config.json` / `pyt_save/model.pt` written by this build are what the reference's
plotting and evaluation scripts read.

    store(**kw)                    epoch_logger.py:347-356   append values to the epoch's state
    log_tabular(key, val, with_min_and_max, average_only, sum_only, rate_only)      :358-401
        val given            -> column `key`
        otherwise            -> Mean<key> (or `key` itself when average_only / sum_only), Std<key>, and with
                                with_min_and_max also Max<key>, Min<key>; statistics exactly as get_stats (:33-39):
                                float32 values, population standard deviation
    dump_tabular()                 :286-311   header on the first row, then tab-separated str(value)
    save_config / setup_pytorch_saver / save_state               :195-284

`log_stats` is this build's addition for quantities reduced on the device (N envs x T steps never visit the host):
it writes the same columns from an (n, mean, std, max, min) tuple.
"""
import json
import os
import time
from collections import defaultdict
from typing import Any, Dict, List, NamedTuple, Optional

import numpy as np
import torch


class Stats(NamedTuple):
    n: int
    sum: float
    min: float
    max: float
    mean: float
    std: float


def get_stats(xs) -> Stats:
    """epoch_logger.py:33-39 (same expressions, same float32 arithmetic)."""
    sum, n = xs.sum(), len(xs)
    mean = sum / n
    std = np.sqrt(((xs - mean) ** 2 / n).sum())
    min = xs.min() if len(xs) > 0 else np.inf
    max = xs.max() if len(xs) > 0 else -np.inf
    return Stats(n=n, sum=sum, min=min, max=max, mean=mean, std=std)


def convert_json(obj: Any) -> Any:
    """epoch_logger.py:42-75: anything json cannot serialise becomes an informative string."""
    try:
        json.dumps(obj)
        return obj
    except Exception:  # noqa: BLE001
        if isinstance(obj, dict):
            return {convert_json(k): convert_json(v) for k, v in obj.items()}
        if isinstance(obj, tuple):
            return tuple(map(convert_json, obj))
        if isinstance(obj, list):
            return list(map(convert_json, obj))
        if hasattr(obj, "__name__") and "lambda" not in obj.__name__:
            return convert_json(obj.__name__)
        if hasattr(obj, "__dict__") and obj.__dict__:
            return {str(obj): {convert_json(k): convert_json(v) for k, v in obj.__dict__.items()}}
        return str(obj)


class Logger:
    """epoch_logger.py:110-311."""

    def __init__(self, output_dir: Optional[str] = None, output_fname: str = "progress.txt", exp_name: Optional[str] = None,
                 quiet: bool = True):
        self.output_dir = str(output_dir) if output_dir is not None else None
        self.output_file = None
        if self.output_dir is not None:
            os.makedirs(self.output_dir, exist_ok=True)
            self.output_file = open(os.path.join(self.output_dir, output_fname), "w+")
        self.first_row = True
        self.log_headers: List[str] = []
        self.log_current_row: Dict[str, Any] = {}
        self.exp_name = exp_name
        self.quiet = quiet
        self.rows: List[Dict[str, Any]] = []

    def log(self, msg: str) -> None:
        print(msg)

    def log_tabular(self, key: str, val: Any) -> None:
        if self.first_row:
            self.log_headers.append(key)
        else:
            assert key in self.log_headers, f"Trying to introduce a new key {key} that you didn't include in the first iteration"
        assert key not in self.log_current_row, f"You already set {key} this iteration. Maybe you forgot to call dump_tabular()"
        self.log_current_row[key] = val

    def save_config(self, config: Any) -> None:
        config_json = convert_json(config)
        if self.exp_name is not None:
            config_json["exp_name"] = self.exp_name
        output = json.dumps(config_json, separators=(",", ":\t"), indent=4, sort_keys=True)
        if self.output_dir is not None:
            with open(os.path.join(self.output_dir, "config.json"), "w+") as out:
                out.write(output)

    def setup_pytorch_saver(self, what_to_save: torch.nn.Module) -> None:
        self.pytorch_saver_elements = what_to_save

    def save_state(self, state_dict: Optional[Dict[str, Any]] = None, itr: Optional[int] = None) -> None:
        """pyt_save/model<itr>.pt = the module's state_dict (epoch_logger.py:216-284).  The reference also pickles
        `state_dict` (usually the env) to vars.pkl; device handles do not pickle, so only plain data is written."""
        if self.output_dir is None:
            return
        if state_dict:
            try:
                torch.save(state_dict, os.path.join(self.output_dir, "vars.pt" if itr is None else f"vars{itr}.pt"))
            except Exception:  # noqa: BLE001
                self.log("Warning: could not save state_dict.")
        if hasattr(self, "pytorch_saver_elements"):
            fpath = os.path.join(self.output_dir, "pyt_save")
            os.makedirs(fpath, exist_ok=True)
            torch.save(self.pytorch_saver_elements.state_dict(), os.path.join(fpath, f"model{itr if itr is not None else ''}.pt"))

    def dump_tabular(self) -> None:
        vals = []
        if not self.quiet:
            key_lens = [len(key) for key in self.log_headers]
            max_key_len = max(15, max(key_lens))
            fmt = "| " + "%" + "%d" % max_key_len + "s | %15s |"
            n_slashes = 22 + max_key_len
            print("-" * n_slashes)
        for key in self.log_headers:
            val = self.log_current_row.get(key, "")
            if not self.quiet:
                print(fmt % (key, "%8.3g" % val if hasattr(val, "__float__") else val))
            vals.append(val)
        if not self.quiet:
            print("-" * n_slashes, flush=True)
        if self.output_file is not None:
            if self.first_row:
                self.output_file.write("\t".join(self.log_headers) + "\n")
            self.output_file.write("\t".join(map(str, vals)) + "\n")
            self.output_file.flush()
        self.rows.append(dict(self.log_current_row))
        self.log_current_row.clear()
        self.first_row = False


class EpochLogger(Logger):
    """epoch_logger.py:314-403."""

    def __init__(self, *args: Any, **kwargs: Any):
        super().__init__(*args, **kwargs)
        self.epoch_dict: Dict[str, List[Any]] = defaultdict(list)

    def store(self, **kwargs: Any) -> None:
        for k, v in kwargs.items():
            self.epoch_dict[k].append(v)

    def _emit(self, key: str, st, with_min_and_max: bool, average_only: bool, sum_only: bool) -> None:
        Logger.log_tabular(self, key if (average_only or sum_only) else "Mean" + key, st.mean)
        if not (average_only or sum_only):
            Logger.log_tabular(self, "Std" + key, st.std)
        if with_min_and_max:
            Logger.log_tabular(self, "Max" + key, st.max)
            Logger.log_tabular(self, "Min" + key, st.min)

    def log_tabular(self, key: str, val: Any = None, with_min_and_max: bool = False, average_only: bool = False,
                    sum_only: bool = False, rate_only: bool = False) -> None:
        if val is not None:
            Logger.log_tabular(self, key, val)
        else:
            v = self.epoch_dict[key]
            vals = (np.concatenate(v) if isinstance(v[0], np.ndarray) and len(v[0].shape) > 0 else np.array(v, dtype=np.float32))
            stats = get_stats(vals)
            if not rate_only:
                self._emit(key, stats, with_min_and_max, average_only, sum_only)
            else:
                Logger.log_tabular(self, key, sum(vals) / sum(self.epoch_dict["EpLen"]))
        self.epoch_dict[key] = []

    def log_stats(self, key: str, mean: float, std: float, max: float, min: float, with_min_and_max: bool = False,
                  average_only: bool = False) -> None:
        """Columns of log_tabular(key, ...) for a quantity whose statistics were reduced on the device."""
        self._emit(key, Stats(n=0, sum=0.0, min=min, max=max, mean=mean, std=std), with_min_and_max, average_only, False)


def setup_logger_kwargs(exp_name: str, seed: Optional[int] = None, data_dir: str = "../exp", env_name: Optional[str] = None) -> Dict[str, Any]:
    """epoch_logger.py:69-107: output_dir = data_dir/(env_name or exp_name)[/<exp_name>_s<seed>]."""
    relpath = os.path.join(str(data_dir), env_name if env_name else exp_name)
    if seed is not None:
        relpath = os.path.join(relpath, f"{exp_name}_s{seed}")
    return dict(output_dir=relpath, exp_name=exp_name)
