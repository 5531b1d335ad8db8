"""Where does a sample group of K7 (rs_ppo_grad2_kernel) spend its cycles?  Diagnostic build with s_memtime stamps at the
phase boundaries (python radiation_ppo_amd/build.py --stamps -> lib/librs_hip_stamps.so, selected through RS_LIB_PATH).
Read the SHARES, not the run time: the stamps' fences forbid overlaps the product kernel has (cdna_hip_programming.md section 7)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RS_LIB_PATH"] = os.path.join(ROOT, "radiation_ppo_amd", "lib", "librs_hip_stamps.so")
import torch  # noqa: E402

from radiation_ppo_amd import _lib  # noqa: E402
from radiation_ppo_amd.ppo import FFActorCritic, FusedPPOGrad  # noqa: E402

PH = ["top: wait DMA, operand reads", "L1 mfma", "tanh1 + b2", "L2 mfma + stage h1^T", "tanh2", "out layer (VALU)", "loss",
      "dW3 (+ row sums)", "dh2 -> dpre2, DMA issue", "R3 dh1 mfma, dpre1", "R4/5 dW2 + db2", "R6 dW1", "-", "-", "-", "loop"]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 480
torch.manual_seed(0)
ac = FFActorCritic().cuda()
X = torch.randn(M, 11, device="cuda")
act = torch.randint(0, 8, (M,), device="cuda")
adv, ret, lpo = torch.randn(M, device="cuda"), torch.randn(M, device="cuda"), -2.0 + 0.1 * torch.randn(M, device="cuda")
w = torch.full((M,), 1.0 / M, device="cuda")
f = FusedPPOGrad(ac)
lib = _lib.load()
lib.rs_debug_k7_stamps.restype = C.c_int
lib.rs_debug_k7_stamps.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    f(X, act, adv, ret, lpo, w, 0.2, 0.1)
buf = (C.c_ulonglong * 32)()
lib.rs_debug_k7_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 10
e0.record()
for _ in range(R):
    f(X, act, adv, ret, lpo, w, 0.2, 0.1)
e1.record()
torch.cuda.synchronize()
lib.rs_debug_k7_stamps(buf, 0)
groups = (M + 31) // 32 * R
if os.environ.get('RS_K7_THREADS') == '256':
    groups //= 2
    print('ONE wave per SIMD (256-thread workgroups): half of the groups run, results are not meaningful')
print(f"stamped build: {e0.elapsed_time(e1) / R:.3f} ms per pass (slower than the product kernel: read the shares)")
for net, name in ((0, "actor"), (1, "critic")):
    cyc = [buf[net * 16 + q] for q in range(16)]
    rt = cyc[14]; cyc[14] = 0
    tot = sum(cyc)
    print(f"{name}: in-kernel clock {tot / max(rt, 1) * 100:.0f} MHz (s_memtime cycles / s_memrealtime 100 MHz ticks)")
    print(f"{name}: {tot / groups:.0f} wave-cycles per 32-sample group")
    for q, c in enumerate(cyc):
        if c:
            print(f"   {PH[q]:30s} {c / groups:8.0f} cyc  {100.0 * c / tot:5.1f} %")
