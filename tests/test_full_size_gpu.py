"""BASELINE configurations 2 and 3 at FULL size (4096 / 8192 envs x 480 steps, 2x64 MLP) checked through size-independent
properties, where a step-by-step oracle replay would take hours:
  * sharding invariance: the full rollout is bit-identical to two half-size rollouts with env_id_base 0 / N/2
    (Philox streams are keyed by the global env id; nothing depends on the launch geometry);
  * episode structure: cuts exactly at terminals, at 120 steps, and at the epoch end; a terminal pays +0.1; rewards live
    on the 2-decimal lattice; the source target stays constant within an episode and >= 1000 cm from the first position;
  * GAE(lambda) / rewards-to-go of sampled columns against the float64 oracle of PPOBuffer (ppo.py:391-423);
  * the fused update's gradient equals the sum of the gradients of the two halves (linearity of the weighted loss).
"""
import numpy as np
import pytest
import torch

from oracle.radsearch_oracle import gae_and_rtg

pytestmark = pytest.mark.gpu
SEED = 289714752
T, L = 480, 120


def _collect(N, base, state_dict=None, obst=0):
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO
    torch.manual_seed(3)
    env = RadSearchVec(N, number_agents=1, obstruction_count=obst, enforce_grid_boundaries=True, seed=SEED, env_id_base=base)
    agents = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1)}
    if state_dict is not None:
        agents[0].agent.load_state_dict(state_dict)
    col = FusedCollector(env, agents, T, L)
    col.collect()
    assert env.error_flags() == 0
    return col, agents


@pytest.mark.parametrize("NE,obst", [(4096, 0), (8192, -1)])      # BASELINE configs 2 and 3
def test_full_size_properties(NE, obst):
    col, agents = _collect(NE, 0, obst=obst)
    sd = {k: v.clone() for k, v in agents[0].agent.state_dict().items()}
    buf = col.buf
    # ---- sharding invariance
    for half, base in ((slice(0, NE // 2), 0), (slice(NE // 2, NE), NE // 2)):
        c2, _ = _collect(NE // 2, base, sd, obst=obst)
        for name in ("obs", "act", "rew", "cut", "val", "logp", "last_val", "source_tar"):
            a, b = getattr(buf, name)[:, half], getattr(c2.buf, name)
            assert torch.equal(a, b), (name, base)
    obs, rew, cut, val, lastv, adv, ret, src = (x.cpu().numpy() for x in (buf.obs, buf.rew, buf.cut, buf.val, buf.last_val,
                                                                          buf.adv, buf.ret, buf.source_tar))
    rew, cut, val, lastv, adv, ret = (x[:, :, 0] for x in (rew, cut, val, lastv, adv, ret))
    # ---- episode structure
    assert cut[T - 1].all()                                                   # the epoch end cuts every trajectory
    assert np.array_equal(np.round(rew.astype(np.float64) * 100) / 100, np.round(rew.astype(np.float64), 2))
    assert np.all((np.abs(rew * 100 - np.round(rew * 100)) < 1e-4))           # 2-decimal lattice (float32 of round(x, 2))
    run = np.zeros(NE, dtype=np.int64)
    n_term = n_timeout = 0
    for t in range(T):
        run += 1
        c = cut[t].astype(bool)
        assert (run[~c] < L).all()
        assert (run[c] <= L).all()
        timeout = c & (run == L)
        terminal = c & ~timeout & (t != T - 1)
        assert np.all(rew[t][terminal] == np.float32(0.1))                    # found the source
        assert np.all(lastv[t][terminal] == 0.0)                              # no bootstrap on a terminal (train.py:487)
        n_term += int(terminal.sum()); n_timeout += int(timeout.sum())
        if t + 1 < T:
            same = ~c
            assert np.array_equal(src[t + 1][same], src[t][same])             # constant within an episode
        run[c] = 0
    assert n_timeout > NE
    # detector positions (obs[1:3] * 2200) stay inside the walls; source >= 1000 cm from the first detector position
    xy = obs[:, :, 0, 1:3].astype(np.float64) * 2200.0
    assert xy.min() >= 0.0 and xy.max() < 2700.0
    first = np.ones(NE, dtype=bool)
    for t in range(T):
        d = np.hypot(xy[t, :, 0] - src[t, :, 0], xy[t, :, 1] - src[t, :, 1])
        assert (d[first] >= 1000.0 - 1e-6).all()
        first = cut[t].astype(bool)
    # ---- GAE / rewards-to-go on sampled columns
    rng = np.random.default_rng(0)
    for n in rng.choice(NE, size=24, replace=False):
        start = 0
        for t in range(T):
            if cut[t, n]:
                a64, r64 = gae_and_rtg(rew[start:t + 1, n], val[start:t + 1, n], lastv[t, n], 0.99, 0.9)
                assert np.array_equal(adv[start:t + 1, n], np.asarray(a64, dtype=np.float64).astype(np.float32)), (n, start, t)
                assert np.array_equal(ret[start:t + 1, n], np.asarray(r64, dtype=np.float64).astype(np.float32)), (n, start, t)
                start = t + 1
    # ---- linearity of the fused gradient over the batch
    from radiation_ppo_amd.ppo import FusedPPOGrad
    X = buf.obs[:, :, 0].reshape(-1, 11)
    act, advn, retn, lpo = (x.reshape(-1) for x in (buf.act, buf.adv, buf.ret, buf.logp))
    M = X.shape[0]
    w = torch.full((M,), 1.0 / M, device=X.device)
    f = FusedPPOGrad(agents[0].agent)
    _, g = f(X, act, advn, retn, lpo, w, 0.2, 0.1)
    g_all = g.clone()
    h = M // 2 + 37                                                           # ragged split
    _, g = f(X[:h], act[:h], advn[:h], retn[:h], lpo[:h], w[:h].contiguous(), 0.2, 0.1)
    g_a = g.clone()
    _, g = f(X[h:], act[h:], advn[h:], retn[h:], lpo[h:], w[h:].contiguous(), 0.2, 0.1)
    scale = float(g_all.abs().max())
    assert torch.allclose(g_all, g_a + g, rtol=1e-4, atol=2e-6 * scale + 1e-9), float((g_all - g_a - g).abs().max())
