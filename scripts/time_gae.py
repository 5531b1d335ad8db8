"""rs_gae timing at the bench sizes (4096 and 8192 columns x 480 steps), HIP events, with the rollout's real cut pattern."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.envs import gae
for M in (4096, 8192):
    T = 480
    g = torch.Generator(device="cuda").manual_seed(0)
    rew = torch.randn(T, M, device="cuda", generator=g); val = torch.randn(T, M, device="cuda", generator=g)
    cut = torch.zeros(T, M, dtype=torch.uint8, device="cuda"); cut[119::120] = 1
    cut |= (torch.rand(T, M, device="cuda", generator=g) < 0.002).to(torch.uint8)
    lv = torch.randn(T, M, device="cuda", generator=g)
    adv = torch.empty_like(rew); ret = torch.empty_like(rew)
    for _ in range(5):
        gae(rew, val, cut, lv, 0.99, 0.9, adv=adv, ret=ret)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        gae(rew, val, cut, lv, 0.99, 0.9, adv=adv, ret=ret)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"rs_gae {M} x {T}: {us:.1f} us per launch = {21.0 * M * T / us / 1e3:.0f} GB/s algorithmic")
