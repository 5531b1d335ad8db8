"""K11 timing: one PFGRU step for 4096 envs x 4 owners (config 4's collector step), HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.pfgru import PredictorBank
N, A = 4096, 4
torch.manual_seed(0)
for carry in (False, True):
    b = PredictorBank(N, A, seed=1, carry_hidden=carry, device="cuda")
    b.reset()
    obs = torch.rand(N, A, 11, device="cuda")
    for _ in range(10):
        b.predict(obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        b.predict(obs)
    e1.record(); torch.cuda.synchronize()
    flop = N * A * 40 * (2 * 28 * 48 * 2 + 2 * 27)
    t = e0.elapsed_time(e1) / 100
    print(f"carry_hidden={carry}: {t * 1e3:.1f} us per step ({N * A} waves), {flop / t / 1e9:.2f} TFLOP/s of matrix-product work")

# the same step with the draws SUPPLIED (rs_pfgru_step_recorded: noise and resampling indices read from HBM instead of hashed in the
# kernel): the difference is what the in-kernel counter hash + Box-Muller cost
from radiation_ppo_amd import _lib
import ctypes as C
lib = _lib.load()
b = PredictorBank(N, A, seed=1, carry_hidden=True, device="cuda")
b.reset()
eps = torch.randn(A, N, 40, 24, device="cuda")
idx = torch.randint(0, 40, (A, N, 40), device="cuda", dtype=torch.int32)
pred = torch.empty(N, A, 2, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def rec():
    _lib.check(lib.rs_pfgru_step_recorded(b._packed().data_ptr(), obs.data_ptr(), b._hq.data_ptr(), b.p.data_ptr(), eps.data_ptr(), idx.data_ptr(),
                                          None, 1, 0.7, pred.data_ptr(), N, A, st), "rec")
for _ in range(10):
    rec()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    rec()
e1.record(); torch.cuda.synchronize()
print(f"recorded draws (no hash / Box-Muller / CDF search in the kernel): {e0.elapsed_time(e1) * 10:.1f} us per step")
