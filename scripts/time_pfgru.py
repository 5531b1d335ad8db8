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
