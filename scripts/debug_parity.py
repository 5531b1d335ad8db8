import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.radsearch_oracle import PhiloxDraws, RadSearchOracle
from radiation_ppo_amd.envs import RadSearchVec
N, A, seed = 64, 1, 289714752
obst = int(sys.argv[1]) if len(sys.argv) > 1 else 3
vec = RadSearchVec(N, number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True, seed=seed)
refs = [RadSearchOracle(PhiloxDraws(seed, n), number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True) for n in range(N)]
obs = vec.reset()[0].cpu().numpy()
bad = 0
for n, e in enumerate(refs):
    exp = np.asarray(e._ret[0][0]).astype(np.float32)
    if not np.array_equal(obs[n, 0], exp):
        bad += 1
        if bad < 4: print("reset mismatch", n, obs[n, 0], exp)
print("reset mismatches", bad)
rect = vec.state("rect").cpu().numpy(); nobs = vec.state("num_obs").cpu().numpy()
print("rects env0 dev", rect[:, 0].reshape(7, 4)[:nobs[0, 0]], "oracle", refs[0].rects)
rng = np.random.default_rng(0)
for t in range(8):
    acts = rng.integers(0, 9, size=(N, A)).astype(np.int8)
    o, r, team, d, info = vec.step(torch.from_numpy(acts).cuda())
    o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
    x = vec.state("x").cpu().numpy(); y = vec.state("y").cpu().numpy(); sp = vec.state("sp").cpu().numpy()
    bad = 0
    for n, e in enumerate(refs):
        ro, rr, rd, ri = e.step({0: int(acts[n, 0])})
        exp = np.asarray(ro[0]).astype(np.float32)
        if not (np.array_equal(o[n, 0], exp) and r[n, 0] == np.float32(rr["individual_reward"][0])):
            bad += 1
            if bad < 4:
                print("step", t, "env", n, "act", acts[n, 0], "\n dev", o[n, 0], r[n, 0], (x[0, n], y[0, n]), sp[0, n],
                      "\n exp", exp, rr["individual_reward"][0], e.agents[0].det, e.agents[0].sp_dist, "lam", e.last_lam)
    print("step", t, "mismatches", bad)
print("err flags", vec.error_flags())
