"""Wall time of one K13 training pass (rs_pfgru_train: forward walk + backward walk) on N full-length episodes; run under
`rocprofv3 --kernel-trace --stats` for the split between the two kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from radiation_ppo_amd.rada2c import KeyDraws, RNNAgentPPO, pack_episodes  # noqa: E402

g = torch.Generator().manual_seed(4)
T, N = 120, int(os.environ.get("K13_EPISODES", "8192"))
obs = torch.rand(T, N, 11, generator=g).cuda()
act = torch.randint(0, 8, (T, N), generator=g).cuda()
z = torch.zeros(T, N).cuda()
src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
cut = torch.zeros(T, N, dtype=torch.uint8); cut[-1] = 1
B = pack_episodes(obs, act, z, z, z, src, cut.cuda(), n_total=N, seed=3)
ag = RNNAgentPPO(id=0, seed=1)
sl = slice(0, B.lens.shape[0])
kd = KeyDraws(B.key * 64 + 1)                         # the product path: draws hashed in the forward walk
ag.model_pass_hip(B, sl, kd); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    ag.model_pass_hip(B, sl, kd)
e1.record(); torch.cuda.synchronize()
print(f"{N} episodes x {T} steps: {e0.elapsed_time(e1) / 3:.2f} ms per pass")
