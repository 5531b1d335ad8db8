"""Build the HIP extension (librs_hip.so) in-tree for gfx950 with hipcc.  No torch headers are
involved: the library is a plain C-ABI shared object (include/radsearch.h).  Each translation unit is
compiled to an object (in parallel, rebuilt only when one of its inputs changed) and linked."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "librs_hip.so")

SOURCES = ["rs_env.hip", "rs_ppo.hip", "rs_maps.hip", "rs_cnn.hip", "rs_pfgru.hip", "rs_gru.hip", "rs_pfgru_train.hip", "rs_rnn_policy.hip", "rs_welford.hip", "rs_cnn_loss.hip"]
# -ffp-contract=off: float64 env arithmetic must round like the reference's Python floats (no FMA fusing)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
          "-Wno-unused-function"]
# rs_ppo.hip: no SLP packing.  hipcc otherwise pairs the scalar f32 FMAs of the output layer into v_pk_fma_f32 + v_mov shuffles,
# which costs more issue slots than it saves next to f32 MFMAs (measured: scripts/micro/mfma_valu_coissue.hip)
EXTRA_CFLAGS = {"rs_ppo.hip": ["-fno-slp-vectorize"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X extension cannot be built")


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "radsearch.h"), __file__]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, defines=(), suffix: str = "") -> str:
    """defines / suffix: a diagnostic variant next to the product library, e.g. build(defines=["RS_K7_STAMPS"],
    suffix="_stamps") -> lib/librs_hip_stamps.so (selected with RS_LIB_PATH; see scripts/k7_stamps.py)."""
    deps = _deps()
    lib = LIB.replace(".so", suffix + ".so")
    if not force and not _newer(lib, deps):
        return lib
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", suffix + ".o"))
        if force or _newer(obj, deps):
            cmd = [cc] + CFLAGS + EXTRA_CFLAGS.get(src, []) + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print("[radiation_ppo_amd.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print("[radiation_ppo_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        build(force="--force" in sys.argv, defines=["RS_K7_STAMPS"], suffix="_stamps")
    else:
        build(force="--force" in sys.argv)
