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

SOURCES = ["rs_env.hip", "rs_ppo.hip", "rs_maps.hip", "rs_cnn.hip"]
# -ffp-contract=off: float64 env arithmetic must round like the reference's Python floats (no FMA fusing)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
          "-Wno-unused-function"]


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


def build(force: bool = False, verbose: bool = True) -> str:
    deps = _deps()
    if not force and not _newer(LIB, deps):
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _newer(obj, deps):
            cmd = [cc] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print("[radiation_ppo_amd.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print("[radiation_ppo_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
