"""K5 (rs_maps_update / rs_maps_reset / rs_maps_stack) through the C ABI against the heat-map oracle, which is
itself pinned to the reference's MapsBuffer: float32-exact on every map of every owner at every step, with the
env kernels producing the observations (obstacles on, several agents, idle steps, episode resets)."""
import os

import numpy as np
import pytest
import torch

from oracle.maps_oracle import MapsOracle
from oracle.radsearch_oracle import PhiloxDraws, RadSearchOracle

pytestmark = pytest.mark.gpu
SEED = 289714752


@pytest.mark.parametrize("A,obst", [(1, 0), (3, 4), (4, -1)])
def test_heat_maps_match_oracle(A, obst):
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import HeatMaps
    N, L = 70, 14
    env = RadSearchVec(N, number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True, seed=SEED)
    hm = HeatMaps(env, steps_per_episode=L)
    assert hm.map_dimensions == (27, 27) and hm.resolution_accuracy == 22.0
    refs = [RadSearchOracle(PhiloxDraws(SEED, n), number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True) for n in range(N)]
    bufs = [[MapsOracle(steps_per_episode=L, number_of_agents=A) for _ in range(A)] for _ in range(N)]
    obs_t = env.reset()[0]
    cur = [e._ret[0] for e in refs]
    rng = np.random.default_rng(A)
    steps = np.zeros(N, dtype=np.int64)
    for t in range(40):
        pred = torch.rand(N, A, 2, generator=torch.Generator().manual_seed(t)).cuda() * 1.2
        hm.update(obs_t, pred)
        actor, critic = hm.stacks()
        actor, critic = actor.cpu().numpy(), critic.cpu().numpy()
        pc = pred.cpu().numpy()
        for n in range(0, N, 5):
            od = {i: np.array(cur[n][i], dtype=np.float64) for i in range(A)}
            for i in range(A):
                m = bufs[n][i].observation_to_map(od, i, (float(pc[n, i, 0]), float(pc[n, i, 1])))
                exp_actor = np.stack([m[0], m[1], m[2], m[3], m[4], m[5]])
                assert np.array_equal(actor[n, i], exp_actor), (t, n, i, np.argwhere(actor[n, i] != exp_actor)[:4])
            mm = bufs[n][0]
            exp_c = np.stack([mm.combined, mm.readings_map, mm.visits, mm.obstacles])
            assert np.array_equal(critic[n], exp_c), (t, n)
        acts = rng.integers(0, 9, size=(N, A)).astype(np.int8)
        if t % 6 == 3:
            acts[:] = 8                                  # everybody idles: same-cell medians and visit counts
        obs_t, _, _, done, _ = env.step(torch.from_numpy(acts).cuda())
        for n, e in enumerate(refs):
            cur[n] = e.step({a: int(acts[n, a]) for a in range(A)})[0]
        steps += 1
        mask = np.array([e.done for e in refs]) | (steps >= L)
        if mask.any():
            mt = torch.from_numpy(mask.astype(np.uint8)).cuda()
            hm.reset(mt)
            obs_t = env.reset(mt)[0]
            for n, e in enumerate(refs):
                if mask[n]:
                    cur[n] = e.reset()[0]
                    for b in bufs[n]:
                        b.reset()
            steps[mask] = 0
    assert int(hm.field("err").max().item()) == 0


def test_cnn_modules_match_reference_outputs(golden_dir):
    """Batched CNN actor / critic == the reference's batch-1 networks on the golden inputs (same state_dict keys)."""
    from radiation_ppo_amd.maps import CNNActor, CNNCritic
    g = dict(np.load(os.path.join(golden_dir, "cnn.npz")).items())
    actor, critic = CNNActor().cuda(), CNNCritic().cuda()
    actor.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("a_actor.")})
    critic.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("c_critic.")})
    xa = torch.from_numpy(g["xa"][:, 0]).cuda()
    xc = torch.from_numpy(g["xc"][:, 0]).cuda()
    with torch.no_grad():
        probs = actor(xa).cpu().numpy()
        vals = critic(xc).cpu().numpy()
    assert np.allclose(probs, g["probs"], rtol=1e-4, atol=1e-5)
    assert np.allclose(vals, g["vals"], rtol=1e-4, atol=1e-5)
    assert sum(p.numel() for p in actor.parameters()) == 88832 and sum(p.numel() for p in critic.parameters()) == 88569
