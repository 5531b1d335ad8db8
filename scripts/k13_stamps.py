"""Where does a step of K13 (rs_pfgru_train_kernel) spend its cycles?  Diagnostic build with s_memtime stamps at the phase boundaries
(build(defines=["RS_K13_STAMPS"], suffix="_k13") -> lib/librs_hip_k13.so, selected through RS_LIB_PATH).  Read the SHARES."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RS_LIB_PATH"] = os.path.join(ROOT, "radiation_ppo_amd", "lib", "librs_hip_k13.so")
import torch  # noqa: E402

from radiation_ppo_amd import _lib  # noqa: E402
from radiation_ppo_amd.rada2c import KernelDraws, RNNAgentPPO, pack_episodes  # noqa: E402

PH = ["fwd: loads + cell", "fwd: softmax, resampling, stores", "bwd: loads + cell recomputed", "bwd: softmax, resampled set, mean",
      "bwd: hid_obs forward + loss", "bwd: hid_obs backward (mvt + 2 outer products)", "bwd: resampling backwards (scatter-add)",
      "bwd: fc_obs outer product", "bwd: d candidate, staging, outer N", "bwd: transposed product N", "bwd: d gates, staging, outer ZR",
      "bwd: transposed product ZR"]
g = torch.Generator().manual_seed(4)
T, N = 120, 1024
obs = torch.rand(T, N, 11, generator=g).cuda()
act = torch.randint(0, 8, (T, N), generator=g).cuda()
z = torch.zeros(T, N).cuda()
src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
cut = torch.zeros(T, N, dtype=torch.uint8); cut[-1] = 1
B = pack_episodes(obs, act, z, z, z, src, cut.cuda(), n_total=N, seed=3)
ag = RNNAgentPPO(id=0, seed=1)
lib = _lib.load()
lib.rs_debug_k13_stamps.restype = C.c_int
lib.rs_debug_k13_stamps.argtypes = [C.c_void_p, C.c_int]
sl = slice(0, B.lens.shape[0])
kd = KernelDraws(B.key * 64 + 1, B.X.shape[0])
ag.model_pass_hip(B, sl, kd); torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib.rs_debug_k13_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ag.model_pass_hip(B, sl, kd); e1.record(); torch.cuda.synchronize()
lib.rs_debug_k13_stamps(buf, 0)
cyc = [buf[q] for q in range(12)]
tot = sum(cyc)
steps = N * T
print(f"stamped build: {e0.elapsed_time(e1):.2f} ms per pass, {tot / steps:.0f} cycles per episode-step")
for q in range(12):
    print(f"  {PH[q]:52s} {cyc[q] / steps:9.0f} cycles  {100.0 * cyc[q] / tot:5.1f} %")
