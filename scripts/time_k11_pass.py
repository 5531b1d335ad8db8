"""K11 pass (rs_pfgru_pass: 120 steps in 30 four-step launches) for E and multiples of E episodes: what the launches' tails cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.rada2c import RNNAgentPPO, HashDraws
ag = RNNAgentPPO(id=0, seed=1)
L, E0 = 120, int(os.environ.get("K11_EPISODES", "16500"))
torch.manual_seed(0)
for mult in (1, 2, 3, 4, 5):
    E = E0 * mult
    X = torch.rand(L, E, 11, device="cuda")
    keys = torch.arange(E, dtype=torch.int64, device="cuda") * 64 + 17
    d = HashDraws(keys)
    lens = [L] * E
    for _ in range(2):
        ag._pfgru_pass_hip(X, d, lens)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ag._pfgru_pass_hip(X, d, lens)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 4
    print(f"{E:6d} episodes x {L} steps: {t:7.2f} ms per pass = {t / mult:6.2f} ms per {E0} episodes")
