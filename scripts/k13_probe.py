"""One K13 launch on a small ragged batch with a synchronisation right behind it (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.rada2c import BpArgs, KernelDraws, RNNAgentPPO, pack_episodes
l1 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
g = torch.Generator().manual_seed(4)
T, N = 40, 24
obs = torch.rand(T, N, 11, generator=g).cuda()
act = torch.randint(0, 8, (T, N), generator=g).cuda()
z = torch.zeros(T, N).cuda()
src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
cut = (torch.rand(T, N, generator=g) < 0.08).to(torch.uint8)
cut[-1] = 1
B = pack_episodes(obs, act, z, z, z, src, cut.cuda(), n_total=N, seed=3)
ag = RNNAgentPPO(id=0, seed=1, bp_args=BpArgs(l1_weight=l1, area_scale=2500.0))
kd = KernelDraws(B.key * 64 + 1, B.X.shape[0])
torch.cuda.synchronize(); print("draws ok", flush=True)
for rep in range(3):
    loss, slab, idx = ag.model_pass_hip(B, slice(0, B.lens.shape[0]), kd)
    torch.cuda.synchronize(); print("pass", rep, float(loss), float(slab.abs().max()), flush=True)
