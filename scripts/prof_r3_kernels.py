"""Micro-driver for the counter passes of the kernels round 2 / 3 added: a few launches each of K9 / K10 (32 768 images, actor and
critic), K11 (4096 envs x 4 owners), K13 (one update_model iteration over a 1024-env epoch), rs_rollout16_kernel<true> and rs_step4
(config 3: 8192 envs with U{1..5} rectangles).  Run directly under rocprofv3 (program right after `--`):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o f --output-format csv -- python3 scripts/prof_r3_kernels.py
prints the algorithmic bytes per launch the traffic is compared with."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radiation_ppo_amd import _lib
from radiation_ppo_amd.envs import RadSearchVec
from radiation_ppo_amd.pfgru import PredictorBank
from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO
from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector

# the config-2 kernels first (K7 grad passes, K6 rollout, K4 GAE, K1 at 4096 and 2^20 envs): scripts/prof_update.py as a module
sys.argv = [sys.argv[0], "3"]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "prof_update.py"), run_name="prof_update")
torch.cuda.empty_cache()

lib = _lib.load()
alg = {}
SEED = 289714752
torch.manual_seed(0)

# ---- K9 / K10
S, A = 32768, 4
g = torch.Generator(device="cuda"); g.manual_seed(0)
maps = (torch.rand(S, 4, 27, 27, device="cuda", generator=g) * (torch.rand(S, 4, 27, 27, device="cuda", generator=g) < 0.15)).contiguous()
cells = torch.randint(0, 729, (S, A), device="cuda", generator=g)
pcells = torch.randint(0, 729, (S, A), device="cuda", generator=g)
st = torch.cuda.current_stream().cuda_stream
for cin, agent in ((6, 0), (4, -1)):
    w1 = torch.randn(8, cin, 3, 3, device="cuda") * 0.2; b1 = torch.rand(8, device="cuda") * 0.1
    w2 = torch.randn(16, 8, 3, 3, device="cuda") * 0.1; b2 = torch.rand(16, device="cuda") * 0.1
    a2 = torch.empty(S, 2704, device="cuda"); p1 = torch.empty(S, 169, 8, device="cuda")
    am = torch.empty(S, 169, 8, dtype=torch.uint8, device="cuda"); mk = torch.empty(S, 169, dtype=torch.int16, device="cuda")
    da2 = torch.randn(S, 2704, device="cuda")
    rows, row = lib.rs_cnn_trunk_slab_rows(S, cin), lib.rs_cnn_trunk_slab_row(cin)
    slab = torch.empty(rows, row, device="cuda")
    cp = (cells.data_ptr(), pcells.data_ptr()) if agent >= 0 else (None, None)
    wt = torch.empty(lib.rs_cnn_trunk_scratch_floats(cin), device="cuda")
    for _ in range(3):
        lib.rs_cnn_trunk_forward(maps.data_ptr(), cp[0], cp[1], A, agent, S, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                 a2.data_ptr(), p1.data_ptr(), am.data_ptr(), mk.data_ptr(), wt.data_ptr(), st)
        lib.rs_cnn_trunk_backward(maps.data_ptr(), cp[0], cp[1], A, agent, S, w2.data_ptr(), da2.data_ptr(), mk.data_ptr(), p1.data_ptr(),
                                  am.data_ptr(), slab.data_ptr(), wt.data_ptr(), st)
    torch.cuda.synchronize()
# per image: forward reads the 4 maps (11 664 B) and writes a2 10 816 + p1 5 408 + amax 1 352 + mask 338; backward reads maps + da2 + p1 + amax + mask
alg["rs_cnn_fwd_kernel"] = S * (11664 + 10816 + 5408 + 1352 + 338)
alg["rs_cnn_bwd_kernel"] = S * (11664 + 10816 + 5408 + 1352 + 338)
del maps, a2, p1, am, mk, da2

# ---- K11
N, A = 4096, 4
b = PredictorBank(N, A, seed=1, carry_hidden=True, device="cuda")
b.reset()
obs = torch.rand(N, A, 11, device="cuda")
for _ in range(4):
    b.predict(obs)
torch.cuda.synchronize()
alg["rs_pfgru_kernel"] = N * A * (2 * (40 * 24 * 4 + 40 * 4) + 12 + 8)        # particle set read + written back, obs row, prediction
del b

# ---- K11 as the policy loop runs it (round 4): one PASS over 16 384 full-length episodes = reset + 30 launches of four time steps
# (rs_pfgru_kernel<false, 4>), the particle sets in registers between the steps of a launch
from radiation_ppo_amd.rada2c import HashDraws
E, Lp = 16384, 120
agp = RNNAgentPPO(id=0, seed=2)
Xp = torch.rand(Lp, E, 11, device="cuda")
for _ in range(2):
    agp._pfgru_pass_hip(Xp, HashDraws(torch.arange(E, device="cuda", dtype=torch.int64) * 64 + 17))
torch.cuda.synchronize()
# per launch: every set read and written back once (2 x 4 000 B), four observation rows (12 B used) and four predictions (8 B)
alg["rs_pfgru_kernel<false, 4>"] = E * (2 * (40 * 24 * 4 + 40 * 4) + 4 * (12 + 8))
del agp, Xp

# ---- rollout16<true>, step4 (config 3)
N, T, L = 8192, 480, 120
env = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED)
ag = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1)}
col = FusedCollector(env, ag, T, L)
col.collect(); col.collect()
acts = torch.randint(0, 9, (N, 1), device="cuda").to(torch.int8)
for _ in range(4):
    env.step(acts)
torch.cuda.synchronize()
alg["rs_rollout16_kernel<true>"] = 77 * N * T
alg["rs_step4_kernel"] = (157 + 48 * float(env.state("num_obs").float().mean().item())) * N     # + rectangles 16 B and cached geodesics 32 B per rectangle
del col, env, ag

# ---- K13 (one update_model iteration at 1024 envs) + the RAD-A2C collector's kernels
N = 1024
env = RadSearchVec(N, obstruction_count=-1, enforce_grid_boundaries=True, seed=SEED)
ag = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1, seed=2, train_pfgru_iters=2, train_pi_iters=1)}
col = RNNCollector(env, ag, T, L, use_graph=False)
col.T = 60                                         # a short epoch is enough: the collector is not what is measured here
col.buf.__init__(60, N, 1, 11, env.device)
col.collect()
col.update()
torch.cuda.synchronize()
ps = ag[0].k13_particle_steps[-1]                  # (one chunk: every K13 launch of the update covers the whole epoch)
# per particle-step: forward writes the resampled particle 96 + 4 + 4 B and the gates 384 B (the draws are hashed in the walk since the end of
# round 4: no 96 + 8 B of noise / uniform to read); backward reads the gates 384 B, two particle sets 200 B and the index 4 B
alg["rs_pfgru_train_kernel"] = ps * (104 + 384 + 384 + 2 * 100 + 4)
print(json.dumps({"algorithmic_bytes_per_launch": alg, "k13_particle_steps": ps}))
