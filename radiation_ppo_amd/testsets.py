"""Safe reader for the reference's saved test-environment sets.

The sets (algos/multiagent/evaluation/test_environments/test_env_dict_obs<k>_<snr>_v4, written by
algos/test_environment/eval/test_env_gen.py:13-69 with joblib.dump) are pickles: `joblib.load` would execute whatever opcodes an
untrusted file contains.  This reader never unpickles.  It walks the opcode stream with `pickletools.genops` (a pure parser)
and interprets a closed subset on its own stack -- containers, strings, numbers, and exactly three globals:

    joblib.numpy_pickle NumpyArrayWrapper   the marker joblib writes in front of an array's raw bytes
    numpy ndarray / numpy dtype             the wrapper's `subclass` and `dtype` fields
    numpy.core.multiarray scalar            numpy integer / float scalars (intensity, background)

Any other global, opcode or dtype raises ValueError; nothing is imported or called on behalf of the file.  The result is the
reference's structure, `{"env_<i>": (src_coords f8[2], det_coords f8[2], intensity, bkg[, obstacles])}` with
obstacles = [[array 4x2], ...] (rad_search_env.py:829-858 reads them back), ready for `RadSearch.refresh_environment` and
`radiation_ppo_amd.evaluate.run_test_environments`.
"""
import io
import pickletools
from typing import Any, Dict

import numpy as np

_ALLOWED_DTYPES = {"f8", "f4", "i8", "i4", "u8", "u4", "i2", "u2", "i1", "u1", "b1"}


class _Global:
    def __init__(self, name: str):
        self.name = name


class _Wrapper:
    """joblib.numpy_pickle.NumpyArrayWrapper placeholder: filled by BUILD, then replaced by the array read from the stream."""

    def __init__(self):
        self.memo_keys = []


class _Mark:
    pass


_GLOBALS = {"joblib.numpy_pickle NumpyArrayWrapper", "numpy ndarray", "numpy dtype", "numpy.core.multiarray scalar",
            "numpy._core.multiarray scalar"}


def _dtype(code: Any) -> np.dtype:
    if not isinstance(code, str) or code.lstrip("<|=") not in _ALLOWED_DTYPES:
        raise ValueError(f"test set: dtype {code!r} is not a plain numeric type")
    return np.dtype(code)


def load_test_environments(path: str) -> Dict[str, tuple]:
    with open(path, "rb") as fh:
        f = io.BytesIO(fh.read())
    stack, memo = [], {}

    def pop_mark():
        items = []
        while True:
            if not stack:
                raise ValueError("test set: MARK underflow")
            v = stack.pop()
            if isinstance(v, _Mark):
                return items[::-1]
            items.append(v)

    gen = pickletools.genops(f)
    result = None
    for op, arg, _pos in gen:
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        if n == "STOP":
            result = stack.pop()
            break
        if n == "MARK":
            stack.append(_Mark())
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n in ("BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8", "BININT", "BININT1", "BININT2", "LONG1", "BINFLOAT",
                   "BINBYTES", "SHORT_BINBYTES"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n in ("BINPUT", "LONG_BINPUT", "MEMOIZE"):
            key = arg if n != "MEMOIZE" else len(memo)
            memo[key] = stack[-1]
            if isinstance(stack[-1], _Wrapper):
                stack[-1].memo_keys.append(key)
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n == "TUPLE1":
            stack[-1:] = [(stack[-1],)]
        elif n == "TUPLE2":
            stack[-2:] = [(stack[-2], stack[-1])]
        elif n == "TUPLE3":
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif n == "LIST":
            stack.append(list(pop_mark()))
        elif n == "APPEND":
            v = stack.pop(); stack[-1].append(v)
        elif n == "APPENDS":
            items = pop_mark(); stack[-1].extend(items)
        elif n == "SETITEM":
            v = stack.pop(); k = stack.pop(); stack[-1][k] = v
        elif n == "SETITEMS":
            items = pop_mark()
            d = stack[-1]
            if not isinstance(d, dict):
                raise ValueError("test set: SETITEMS on a non-dict")
            for i in range(0, len(items), 2):
                d[items[i]] = items[i + 1]
        elif n in ("GLOBAL", "STACK_GLOBAL"):
            if n == "STACK_GLOBAL":
                name = stack.pop(); mod = stack.pop(); arg = f"{mod} {name}"
            if arg not in _GLOBALS:
                raise ValueError(f"test set: global {arg!r} is not allowed (the reader executes nothing)")
            stack.append(_Global(arg))
        elif n == "NEWOBJ":
            args = stack.pop(); cls = stack.pop()
            if not (isinstance(cls, _Global) and cls.name.endswith("NumpyArrayWrapper") and args == ()):
                raise ValueError("test set: NEWOBJ on something other than joblib's array wrapper")
            stack.append(_Wrapper())
        elif n == "REDUCE":
            args = stack.pop(); fn = stack.pop()
            if not isinstance(fn, _Global):
                raise ValueError("test set: REDUCE on a non-global")
            if fn.name == "numpy dtype":
                stack.append(_dtype(args[0]))
            elif fn.name.endswith("multiarray scalar"):
                dt, raw = args
                if not isinstance(dt, np.dtype) or not isinstance(raw, (bytes, bytearray)) or len(raw) != dt.itemsize:
                    raise ValueError("test set: malformed numpy scalar")
                stack.append(np.frombuffer(raw, dtype=dt)[0].item())
            else:
                raise ValueError(f"test set: call of {fn.name!r} is not allowed")
        elif n == "BUILD":
            state = stack.pop()
            target = stack[-1]
            if isinstance(target, np.dtype):
                # dtype.__setstate__: (version, endianness, ...) -- only little-endian / not-applicable plain types occur
                if not (isinstance(state, tuple) and state[1] in ("<", "|", "=")):
                    raise ValueError("test set: unsupported dtype state")
            elif isinstance(target, _Wrapper):
                if not isinstance(state, dict) or not isinstance(state.get("subclass"), _Global) or state["subclass"].name != "numpy ndarray":
                    raise ValueError("test set: unsupported array wrapper")
                dt, shape, order = state["dtype"], tuple(int(s) for s in state["shape"]), state.get("order", "C")
                if not isinstance(dt, np.dtype) or dt.hasobject or order not in ("C", "F"):
                    raise ValueError("test set: unsupported array layout")
                pad = state.get("numpy_array_alignment_bytes")
                if pad is not None:                              # joblib >= 1.2: one byte = padding length, then the padding
                    k = f.read(1)[0]
                    f.read(k)
                count = int(np.prod(shape)) if shape else 1
                if count * dt.itemsize > 1 << 24:
                    raise ValueError("test set: array too large for a saved environment")
                raw = f.read(count * dt.itemsize)
                if len(raw) != count * dt.itemsize:
                    raise ValueError("test set: truncated array data")
                arr = np.frombuffer(raw, dtype=dt).reshape(shape, order=order).copy()
                stack[-1] = arr
                for k in target.memo_keys:                        # the memoised wrapper now IS the array
                    memo[k] = arr
            else:
                raise ValueError("test set: BUILD on an unexpected object")
        else:
            raise ValueError(f"test set: pickle opcode {n} is outside the data-only subset this reader accepts")
    if not isinstance(result, dict) or not all(isinstance(k, str) and k.startswith("env_") for k in result):
        raise ValueError("test set: top-level object is not an env_<i> dictionary")
    for k, v in result.items():
        if not (isinstance(v, tuple) and len(v) in (4, 5) and isinstance(v[0], np.ndarray) and isinstance(v[1], np.ndarray)):
            raise ValueError(f"test set: {k} is not (src, det, intensity, bkg[, obstacles])")
    return result


def sets_from_arrays(src, det, intensity, bkg, rects=None) -> Dict[str, tuple]:
    """The reference's set structure from plain arrays (tests/golden/testset_obs<k>_<snr>.npz: the first 100 environments of a saved
    set as written by tests/golden/make_checkpoints.py): src / det [E, 2], intensity / bkg [E], rects [E, k, 4, 2] corner lists."""
    out = {}
    for i in range(len(src)):
        e = [np.asarray(src[i], dtype=np.float64), np.asarray(det[i], dtype=np.float64), int(intensity[i]), int(bkg[i])]
        if rects is not None and np.asarray(rects).shape[1] > 0:
            e.append([[np.asarray(r, dtype=np.float64)] for r in rects[i]])
        out[f"env_{i}"] = tuple(e)
    return out


def load_test_environments_npz(path: str) -> Dict[str, tuple]:
    z = np.load(path)
    return sets_from_arrays(z["src"], z["det"], z["intensity"], z["bkg"], z["rects"])


def summarize_test_set(env_sets: Dict[str, tuple]) -> Dict[str, Any]:
    """Counts and ranges of a set (what test_env_gen.py:38-60 classifies): signal-to-noise I / r^2 / bkg + 1 of the start."""
    src = np.array([e[0] for e in env_sets.values()], dtype=np.float64)
    det = np.array([e[1] for e in env_sets.values()], dtype=np.float64)
    inten = np.array([e[2] for e in env_sets.values()], dtype=np.float64)
    bkg = np.array([e[3] for e in env_sets.values()], dtype=np.float64)
    r = np.linalg.norm(src - det, axis=1)
    snr = (inten / r ** 2 + bkg) / bkg
    nobs = [len(e[4]) if len(e) > 4 else 0 for e in env_sets.values()]
    return {"count": len(env_sets), "min_start_distance": float(r.min()), "snr_min": float(snr.min()), "snr_max": float(snr.max()),
            "intensity_range": (float(inten.min()), float(inten.max())), "bkg_range": (float(bkg.min()), float(bkg.max())),
            "obstructions": (int(min(nobs)), int(max(nobs)))}
