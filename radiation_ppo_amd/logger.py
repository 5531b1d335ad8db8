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
Only the FILE FORMATS and the call surface follow the reference (progress.txt is pinned byte for byte by
tests/test_train_loop_golden.py); the code is this build's own.
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
    """Count / sum / extrema / mean / population standard deviation of a float32 vector.  File-format contract: the numbers are
    printed with str() into progress.txt, so they are formed in the array's own precision in this order -- sum, sum / n, then
    sqrt(sum((x - mean)^2 / n)) -- which is what the reference's rows contain (epoch_logger.py:33-39; pinned byte for byte by
    tests/golden/train_trace.json)."""
    n = len(xs)
    total = np.add.reduce(xs)
    mean = total / n
    dev = xs - mean
    spread = np.sqrt(np.add.reduce(dev * dev / n))
    lo, hi = (np.minimum.reduce(xs), np.maximum.reduce(xs)) if n else (np.inf, -np.inf)
    return Stats(n=n, sum=total, min=lo, max=hi, mean=mean, std=spread)


_PLAIN = (str, int, float, bool, type(None))


def convert_json(obj: Any) -> Any:
    """A json.dumps-able image of an arbitrary configuration object (what save_config writes to config.json, whose readers are the
    reference's plotting scripts, epoch_logger.py:42-75 defines the result): plain scalars as they are, containers element-wise,
    named callables / classes by name, objects with attributes as {str(obj): {attribute: image}}, anything else as str(obj)."""
    if isinstance(obj, _PLAIN):
        return obj
    if isinstance(obj, dict):
        return {convert_json(k): convert_json(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [convert_json(v) for v in obj]
    name = getattr(obj, "__name__", None)
    if isinstance(name, str) and "lambda" not in name:
        return name
    attrs = getattr(obj, "__dict__", None)
    if attrs:
        return {str(obj): {convert_json(k): convert_json(v) for k, v in attrs.items()}}
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
        """Close the epoch's row.  File-format contract (progress.txt, read by the reference's plot / compare scripts): a header
        line with the column names on the first row, then one line per epoch, str() of every value, tab separated."""
        row = [self.log_current_row.get(key, "") for key in self.log_headers]
        if not self.quiet:
            width = max(len(k) for k in self.log_headers) if self.log_headers else 0
            shown = [f"{v:.4g}" if isinstance(v, (float, np.floating)) else str(v) for v in row]
            print("\n".join(f"  {k:<{width}}  {v}" for k, v in zip(self.log_headers, shown)) + "\n", flush=True)
        if self.output_file is not None:
            if self.first_row:
                self.output_file.write("\t".join(self.log_headers) + "\n")
            self.output_file.write("\t".join(str(v) for v in row) + "\n")
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
        """One column from a value, or the statistics columns of everything stored under `key` this epoch (column names and
        their order are the progress.txt contract: Mean<key> | <key>, Std<key>, Max<key>, Min<key>; epoch_logger.py:358-401)."""
        if val is not None:
            Logger.log_tabular(self, key, val)
            self.epoch_dict[key] = []
            return
        stored = self.epoch_dict[key]
        arrays = isinstance(stored[0], np.ndarray) and stored[0].ndim > 0
        flat = np.concatenate(stored) if arrays else np.asarray(stored, dtype=np.float32)
        if rate_only:
            Logger.log_tabular(self, key, sum(flat) / sum(self.epoch_dict["EpLen"]))
        else:
            self._emit(key, get_stats(flat), with_min_and_max, average_only, sum_only)
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
