"""GPU tests of the PPO side: GAE kernel (bit-exact vs the reference's PPOBuffer golden vectors and the
oracle), MLP architecture vs FF_core golden outputs, batched Welford vs the reference trace, Philox
action uniforms, and an end-to-end collector run replayed through the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle.radsearch_oracle import (PhiloxDraws, RadSearchOracle, WelfordOracle, gae_and_rtg, philox4x32_10)

pytestmark = pytest.mark.gpu
SEED = 289714752


def test_gae_kernel_bit_exact_vs_reference_buffer(golden_dir):
    from radiation_ppo_amd.envs import gae
    g = dict(np.load(os.path.join(golden_dir, "gae.npz")).items())
    T = len(g["rew"])
    M = 70                                   # replicate the column; every lane must agree
    rew = torch.tensor(g["rew"], dtype=torch.float32).view(T, 1).repeat(1, M).cuda()
    val = torch.tensor(g["val"], dtype=torch.float32).view(T, 1).repeat(1, M).cuda()
    cut = torch.tensor(g["cut"], dtype=torch.uint8).view(T, 1).repeat(1, M).cuda()
    lv = torch.tensor(g["last_val"], dtype=torch.float32).view(T, 1).repeat(1, M).cuda()
    adv, ret = gae(rew, val, cut, lv, float(g["gamma"]), float(g["lam"]))
    assert np.array_equal(adv[:, 0].cpu().numpy(), g["adv_raw"])
    assert np.array_equal(ret[:, 0].cpu().numpy(), g["ret"])
    assert torch.equal(adv, adv[:, :1].expand_as(adv)) and torch.equal(ret, ret[:, :1].expand_as(ret))


def test_gae_kernel_random_cuts_vs_oracle():
    from radiation_ppo_amd.envs import gae
    rng = np.random.default_rng(0)
    T, M = 97, 133                           # ragged sizes
    rew = rng.normal(size=(T, M)).astype(np.float32)
    val = rng.normal(size=(T, M)).astype(np.float32)
    cut = (rng.random((T, M)) < 0.07).astype(np.uint8)
    cut[-1] = 1
    lv = (rng.normal(size=(T, M)) * (rng.random((T, M)) < 0.5)).astype(np.float32)
    adv, ret = gae(*(torch.from_numpy(a).cuda() for a in (rew, val, cut, lv)), 0.99, 0.9)
    adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
    for m in range(0, M, 7):
        s = 0
        for t in range(T):
            if cut[t, m]:
                a, r = gae_and_rtg(rew[s:t + 1, m], val[s:t + 1, m], float(lv[t, m]), 0.99, 0.9)
                assert np.array_equal(adv[s:t + 1, m], np.array(a).astype(np.float32))
                assert np.array_equal(ret[s:t + 1, m], np.array(r).astype(np.float32))
                s = t + 1


@pytest.mark.parametrize("T,M,p_cut,end_cut", [(480, 200, 0.01, True), (480, 4096, 0.02, True), (33, 70, 0.0, False),
                                                  (257, 5000, 0.004, False), (1, 64, 1.0, True)])
def test_gae_kernel_time_chunks_equal_the_serial_scan(T, M, p_cut, end_cut):
    """The kernel splits the reverse scan into time chunks (each lane replays from the next cut down to its chunk); every
    value must equal the float64 serial scan of the whole column (scipy.lfilter's recurrence, ppo.py:62-85), including
    trajectories that span several chunks, columns with no cut at all (zero start state) and T < one chunk."""
    from radiation_ppo_amd.envs import gae
    rng = np.random.default_rng(T * 7 + M)
    rew = rng.normal(size=(T, M)).astype(np.float32)
    val = rng.normal(size=(T, M)).astype(np.float32)
    cut = (rng.random((T, M)) < p_cut).astype(np.uint8)
    if end_cut:
        cut[-1] = 1
    lv = rng.normal(size=(T, M)).astype(np.float32)
    adv, ret = gae(*(torch.from_numpy(a).cuda() for a in (rew, val, cut, lv)), 0.99, 0.9)
    adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
    a_acc = np.zeros(M); r_acc = np.zeros(M); v_next = np.zeros(M)
    ea, er = np.empty((T, M), np.float32), np.empty((T, M), np.float32)
    for t in range(T - 1, -1, -1):
        c = cut[t].astype(bool)
        lvt = lv[t].astype(np.float64)
        v_next = np.where(c, lvt, v_next); r_acc = np.where(c, lvt, r_acc); a_acc = np.where(c, 0.0, a_acc)
        r, v = rew[t].astype(np.float64), val[t].astype(np.float64)
        delta = r + 0.99 * v_next - v
        a_acc = delta + (0.99 * 0.9) * a_acc
        r_acc = r + 0.99 * r_acc
        ea[t], er[t] = a_acc.astype(np.float32), r_acc.astype(np.float32)
        v_next = v
    assert np.array_equal(adv, ea) and np.array_equal(ret, er)


def test_ff_actor_critic_matches_reference_outputs(golden_dir):
    from radiation_ppo_amd.ppo import FFActorCritic
    g = dict(np.load(os.path.join(golden_dir, "ff_core.npz")).items())
    ac = FFActorCritic().cuda()
    ac.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")})   # same key names
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        probs = ac.actor(x)
        logp, v, ent = ac.evaluate(x, torch.from_numpy(g["act"]).cuda())
    tol = dict(rtol=1e-4, atol=1e-5)         # fp32 tolerance: different GEMM summation order on MFMA
    assert np.allclose(probs.cpu().numpy(), g["probs"], **tol)
    assert np.allclose(v.cpu().numpy(), g["values"][:, 0], **tol)
    assert np.allclose(logp.cpu().numpy(), g["logp"], **tol)
    assert np.allclose(ent.cpu().numpy(), g["ent"], **tol)


def test_device_welford_matches_reference_trace(golden_dir):
    from radiation_ppo_amd.ppo import DeviceWelford
    g = np.load(os.path.join(golden_dir, "welford.npz"))
    st = DeviceWelford((3, 1), "cuda")
    for x, z in zip(g["x"][:80], g["z"][:80]):
        xt = torch.full((3, 1), float(x), dtype=torch.float32, device="cuda")
        st.update(xt)
        zz = ((xt.double() - st.mean) / st.std).cpu().numpy()
        assert np.allclose(zz, z, rtol=1e-12, atol=1e-12)
    st.reset(torch.tensor([True, False, False], device="cuda"))
    assert st.count[0, 0].item() == 0 and st.count[1, 0].item() == 80 and st.std[0, 0].item() == 1.0


def test_action_uniforms_are_the_documented_philox_stream():
    from radiation_ppo_amd.envs import RadSearchVec
    vec = RadSearchVec(70, number_agents=2, enforce_grid_boundaries=True, seed=SEED, env_id_base=5)
    vec.reset()
    u = vec.action_uniforms(torch.empty(70, 2, device="cuda")).cpu().numpy()
    for n in (0, 1, 33, 69):
        for a in range(2):
            o = philox4x32_10(0, 1, 0, 32 + a, SEED, 5 + n)      # t = 1 after reset, episode 0
            assert u[n, a] == np.float32((o[0] >> 8) / 16777216.0)


def test_collector_rollout_replays_through_oracle():
    """End to end: run the on-device collector for one epoch, then replay the actions it stored through
    per-env oracles; every stored observation/reward/cut must be reproduced (bit-exact env outputs, the
    float64 Welford standardisation within 1 ulp of float32), and logp/val must match a torch re-evaluation."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.ppo import Collector, VecAgentPPO
    N, T, L = 40, 48, 12
    torch.manual_seed(0)
    env = RadSearchVec(N, number_agents=1, obstruction_count=2, enforce_grid_boundaries=True, seed=SEED)
    agents = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1)}
    col = Collector(env, agents, T, L)
    col.collect()
    buf = col.buf
    _replay_check(col, agents, N, T, L, 2)
    # logp / val stored by the collector == re-evaluation of the same network on the stored inputs
    X = buf.obs[:, :, 0].reshape(-1, 11)
    with torch.no_grad():
        logp, v, _ = agents[0].agent.evaluate(X, buf.act[:, :, 0].reshape(-1))
    assert torch.allclose(logp, buf.logp[:, :, 0].reshape(-1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(v, buf.val[:, :, 0].reshape(-1), rtol=1e-4, atol=1e-5)
    # update runs and respects the KL early stop
    res = col.update()[0]
    assert 1 <= res.stop_iteration <= 40 and np.isfinite(res.loss_policy)


def test_normalize_advantages_matches_reference_buffer(golden_dir):
    """P3: PPOBuffer.get (ppo.py:445-446) normalises with mpi_statistics_scalar's mean and POPULATION std, no epsilon;
    the product function on the reference's raw advantages must give the reference's normalised ones (gae.npz)."""
    from radiation_ppo_amd.ppo import normalize_advantages
    g = np.load(os.path.join(golden_dir, "gae.npz"))
    adv = torch.from_numpy(g["adv_raw"].astype(np.float32)).cuda()
    got = normalize_advantages(adv).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == g["adv_norm"].shape
    # tolerance: the reference reduces the float32 array in float32 (numpy pairwise sums, mpi_tools.py:83-87), the product in
    # float64 on the device -- the two statistics differ in the last float32 bits
    assert np.allclose(got, g["adv_norm"], rtol=1e-5, atol=1e-6), float(np.abs(got - g["adv_norm"]).max())
    assert abs(float(got.mean())) < 1e-6 and abs(float(got.std()) - 1.0) < 1e-5


def test_train_ppo_entry_point_runs():
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.train import train_PPO
    env = RadSearchVec(32, number_agents=1, obstruction_count=0, enforce_grid_boundaries=True, seed=3)
    sim = train_PPO(env=env, logger_kwargs={}, ppo_kwargs=dict(observation_space=11, steps_per_epoch=24, steps_per_episode=8,
                                                               number_of_agents=1, alpha=0.1, train_pi_iters=5),
                    seed=3, number_of_agents=1, actor_critic_architecture="ff", global_critic_flag=False,
                    steps_per_epoch=24, steps_per_episode=8, total_epochs=2)
    sim.train()
    rows = sim.loggers[0].rows
    assert len(rows) == 2 and rows[1]["TotalEnvInteracts"] == 2 * 24 * 32 and np.isfinite(rows[1]["loss_policy"])


def test_mfma_policy_forward_vs_torch_and_reference(golden_dir):
    """rs_policy_forward (v_mfma_f32_32x32x2_f32 layers + VALU heads) against (a) the reference's own
    FF_core outputs (golden), (b) torch fp32 on ragged batch sizes with asymmetric random weights."""
    from radiation_ppo_amd.ppo import FFActorCritic, policy_forward
    g = dict(np.load(os.path.join(golden_dir, "ff_core.npz")).items())
    ac = FFActorCritic().cuda()
    ac.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd_")})
    logits, value = policy_forward(ac, torch.from_numpy(g["x"]).cuda())
    probs = torch.softmax(logits, -1).cpu().numpy()
    assert np.allclose(probs, g["probs"], rtol=1e-4, atol=1e-5)
    assert np.allclose(value.cpu().numpy(), g["values"][:, 0], rtol=1e-4, atol=1e-5)
    torch.manual_seed(5)
    ac = FFActorCritic().cuda()
    with torch.no_grad():
        for p in ac.parameters():
            p.copy_(torch.randn_like(p) * 0.3)          # asymmetric weights: catches transposed fragments
    for M in (1, 63, 64, 65, 1000, 4096 * 3 + 17):
        x = torch.randn(M, 11, device="cuda")
        logits, value = policy_forward(ac, x)
        with torch.no_grad():
            ref_l = ac.logits(x)
            ref_v = ac.critic(x).squeeze(-1)
        assert torch.allclose(logits, ref_l, rtol=1e-4, atol=2e-5), (M, (logits - ref_l).abs().max())
        assert torch.allclose(value, ref_v, rtol=1e-4, atol=2e-5), (M, (value - ref_v).abs().max())


def _replay_check(col, agents, N, T, L, obst, stride=3):
    """Drive oracle/train_loop_oracle.train_loop_trace (RAD-A2C branch; pinned to the reference's own train() by
    tests/test_train_loop_golden.py) with the actions / values the device stored, env by env, and require the
    device buffer to hold what the trace passes to store(): standardised observation (float64 Welford, within 1 ulp
    of float32), reward, source target, terminal flag -- and a zero bootstrap exactly on terminal trajectories."""
    from oracle.train_loop_oracle import train_loop_trace
    buf = col.buf
    obs, act, rew, cut, val, logp = (t.cpu().numpy() for t in (buf.obs, buf.act, buf.rew, buf.cut, buf.val, buf.logp))
    lastv = buf.last_val.cpu().numpy()
    src = buf.source_tar.cpu().numpy()
    for n in range(0, N, stride):
        ref = RadSearchOracle(PhiloxDraws(SEED, n), number_agents=1, obstruction_count=obst, enforce_grid_boundaries=True)
        st = {"t": 0, "after_step": False, "first": True}

        class Env:
            src = property(lambda s: ref.src)

            def reset(s):
                st["after_step"] = False
                if st["first"]:                      # train() opens with env.reset(): the constructor's reset stands for it
                    st["first"] = False
                    return ref._ret
                return ref.reset()

            def step(s, a):
                r = ref.step(a)
                st["t"] += 1
                st["after_step"] = True
                return r

            def __setattr__(s, k, v):
                setattr(ref, k, v)

        def agent_step(i, observations):
            if st["after_step"] and cut[st["t"] - 1, n, 0]:          # bootstrap call (train.py:476-480)
                return 0, float(lastv[st["t"] - 1, n, 0]), 0.0
            t = st["t"]
            return int(act[t, n, 0]), float(val[t, n, 0]), float(logp[t, n, 0])

        ev, _ = train_loop_trace(Env(), agent_step, 1, False, T, L, 1, arch="mlp")
        stores = [e for e in ev if e[0] == "store"]
        assert len(stores) == T
        for t, e in enumerate(stores):
            x = np.asarray(e[2], dtype=np.float64).astype(np.float32)
            assert np.allclose(obs[t, n, 0], x, rtol=2e-7, atol=1e-7), (n, t, obs[t, n, 0], x)
            assert tuple(float(v) for v in src[t, n]) == tuple(e[7]), (n, t)
            assert rew[t, n, 0] == np.float32(e[3]), (n, t)
            assert bool(cut[t, n, 0]) == e[8], (n, t)
        gae = [e for e in ev if e[0] == "gae"]
        cuts_t = [t for t in range(T) if cut[t, n, 0]]
        assert len(gae) == len(cuts_t)
        for t, e in zip(cuts_t, gae):
            assert float(lastv[t, n, 0]) == e[2], (n, t)             # 0.0 where the trace ended on a terminal
        assert ref.err == 0


@pytest.mark.parametrize("obst", [0, 3])
def test_fused_collector_replays_through_oracle(obst):
    """rs_rollout (one launch per epoch): the actions it sampled, replayed through per-env oracles, reproduce
    every stored observation / reward / cut bit-exactly; stored logp / val / last_val match torch fp32
    re-evaluation of the same network; sampled actions are the inverse CDF of the documented Philox uniforms;
    a second epoch continues from the carried state."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO
    N, T, L = 128, 40, 12
    torch.manual_seed(1)
    env = RadSearchVec(N, number_agents=1, obstruction_count=obst, enforce_grid_boundaries=True, seed=SEED)
    agents = {0: VecAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, alpha=0.1)}
    with torch.no_grad():
        for p in agents[0].agent.parameters():
            p.mul_(3.0)                              # make the policy visibly non-uniform
    col = FusedCollector(env, agents, T, L)
    stats = col.collect()
    _replay_check(col, agents, N, T, L, obst)
    buf = col.buf
    X = buf.obs[:, :, 0].reshape(-1, 11)
    with torch.no_grad():
        logp, v, _ = agents[0].agent.evaluate(X, buf.act[:, :, 0].reshape(-1))
        probs = torch.softmax(agents[0].agent.logits(X), -1)
    assert torch.allclose(logp, buf.logp.reshape(-1), rtol=1e-4, atol=2e-5)
    assert torch.allclose(v, buf.val.reshape(-1), rtol=1e-4, atol=2e-5)
    # bootstrap values: where cut by timeout/epoch end, last_val == critic(standardised next obs) is covered
    # by the replay of obs at t+1 for non-reset envs; here check the action sampling rule on the stored rows
    cdf = torch.cumsum(probs, -1)
    a = buf.act.reshape(-1)
    lo = torch.where(a > 0, cdf.gather(1, (a - 1).clamp(min=0).unsqueeze(1)).squeeze(1), torch.zeros_like(cdf[:, 0]))
    hi = torch.where(a < 7, cdf.gather(1, a.unsqueeze(1)).squeeze(1), torch.full_like(cdf[:, 0], 2.0))
    assert (hi - lo > -1e-5).all()
    assert int(stats["EpCount"].item()) >= N * (T // L) - N
    # episode statistics handed to the logger (MeanEpRet / StdEpRet / MaxEpRet / MinEpRet / EpLen, train.py:494-501, :620):
    # rebuilt from the buffer -- float32 running return per episode, an episode counts when it ended on a terminal or a
    # timeout (an epoch cut alone does not; a terminal is the cut whose bootstrap value is exactly 0)
    rew, cut, lastv = (x[:, :, 0].cpu().numpy() for x in (buf.rew, buf.cut, buf.last_val))
    rets, lens = [], []
    for n in range(N):
        acc, run = np.float32(0.0), 0
        for t in range(T):
            acc = np.float32(acc + rew[t, n]); run += 1
            if cut[t, n]:
                if t < T - 1 or run == L or lastv[t, n] == 0.0:
                    rets.append(float(acc)); lens.append(run)
                acc, run = np.float32(0.0), 0
    rets = np.array(rets)
    assert int(stats["EpCount"].item()) == len(rets) and float(stats["EpLenSum"].item()) == float(sum(lens))
    assert abs(float(stats["EpRetSum"][0]) - rets.sum()) < 1e-3 and abs(float(stats["EpRetSqSum"][0]) - (rets ** 2).sum()) < 1e-2
    assert abs(float(stats["EpRetMax"][0]) - rets.max()) < 1e-6 and abs(float(stats["EpRetMin"][0]) - rets.min()) < 1e-6
    # second epoch continues (state carried in the collector tensors) and still replays
    col.collect()
    res = col.update()[0]
    assert 1 <= res.stop_iteration <= 40 and np.isfinite(res.loss_policy)


def _torch_loss_and_grads(ac, X, act, adv, ret, lpo, w, clip, alpha, vf=0.01):
    for p in ac.parameters():
        p.grad = None
    logp, v, ent = ac.evaluate(X, act)
    ratio = torch.exp(logp - lpo)
    surr = torch.min(ratio * adv, torch.clamp(ratio, 1 - clip, 1 + clip) * adv)
    vl = (w * (v - ret) ** 2).sum()
    loss = -((w * surr).sum() - vf * vl + alpha * (w * ent).sum().detach())     # ppo.py:1216: entropy is detached
    loss.backward()
    clipped = (ratio > 1 + clip) | (ratio < 1 - clip)
    stats = [(w * (lpo - logp)).sum().item(), (w * ent).sum().item(), (w * clipped.float()).sum().item(), vl.item(), loss.item()]
    order = [ac.actor[0].weight, ac.actor[0].bias, ac.actor[2].weight, ac.actor[2].bias, ac.actor[4].weight, ac.actor[4].bias,
             ac.critic[0].weight, ac.critic[0].bias, ac.critic[2].weight, ac.critic[2].bias, ac.critic[4].weight, ac.critic[4].bias]
    return stats, torch.cat([p.grad.reshape(-1) for p in order])


@pytest.mark.parametrize("M", [64, 1000, 4096 * 5 + 3])
def test_fused_ppo_grad_matches_autograd(M):
    """rs_ppo_grad (forward + loss + backward + weight-gradient GEMMs on the matrix cores) against torch
    autograd on the same batch: loss statistics and every parameter gradient, fp32 tolerance."""
    from radiation_ppo_amd.ppo import FFActorCritic, FusedPPOGrad
    torch.manual_seed(M)
    ac = FFActorCritic().cuda()
    with torch.no_grad():
        for p in ac.parameters():
            p.copy_(torch.randn_like(p) * 0.25)
    X = torch.randn(M, 11, device="cuda")
    act = torch.randint(0, 8, (M,), device="cuda")
    adv = torch.randn(M, device="cuda")
    ret = torch.randn(M, device="cuda")
    with torch.no_grad():
        lpo = ac.evaluate(X, act)[0] + 0.3 * torch.randn(M, device="cuda")     # ratios on both sides of the clip
    w = torch.rand(M, device="cuda")
    w = w / w.sum()
    ref_stats, ref_g = _torch_loss_and_grads(ac, X, act, adv, ret, lpo, w, 0.2, 0.1)
    fused = FusedPPOGrad(ac)
    stats, g = fused(X, act, adv, ret, lpo, w, 0.2, 0.1)
    stats = stats.tolist()
    for a, b in zip(stats, ref_stats):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (stats, ref_stats)
    scale = ref_g.abs().max().item()
    err = (g - ref_g).abs().max().item()
    assert err <= 2e-4 * scale, (err, scale)
    # per-tensor check (catches a transposed or mis-ordered block that a global max could hide)
    o = 0
    for name, n in (("a.w1", 704), ("a.b1", 64), ("a.w2", 4096), ("a.b2", 64), ("a.w3", 512), ("a.b3", 8),
                    ("c.w1", 704), ("c.b1", 64), ("c.w2", 4096), ("c.b2", 64), ("c.w3", 64), ("c.b3", 1)):
        r, f = ref_g[o:o + n], g[o:o + n]
        assert (f - r).abs().max().item() <= 3e-4 * max(r.abs().max().item(), 1e-6) + 1e-7, name
        o += n
    # bitwise reproducible (slab reduction, no float atomics)
    stats2, g2 = fused(X, act, adv, ret, lpo, w, 0.2, 0.1)
    assert torch.equal(g2, g.clone()) or torch.equal(g2, g)


def test_device_side_update_loop_matches_torch_adam():
    """The sync-free update (rs_ppo_grad + rs_adam_step per iteration, KL early stop decided on the device)
    against the same loop written with torch autograd + torch.optim.Adam."""
    from radiation_ppo_amd.ppo import VecAgentPPO
    M = 4096
    torch.manual_seed(7)
    X = torch.randn(M, 11, device="cuda")
    act = torch.randint(0, 8, (M,), device="cuda")
    adv = torch.randn(M, device="cuda")
    ret = torch.randn(M, device="cuda")
    w = torch.rand(M, device="cuda"); w = w / w.sum()
    for target_kl, expect_early in ((10.0, False), (1e-4, True)):
        torch.manual_seed(11)
        a1 = VecAgentPPO(id=0, alpha=0.1, train_pi_iters=8, target_kl=target_kl, actor_learning_rate=1e-2)
        torch.manual_seed(11)
        a2 = VecAgentPPO(id=0, alpha=0.1, train_pi_iters=8, target_kl=target_kl, actor_learning_rate=1e-2)
        a2.fused_update = False
        with torch.no_grad():
            lpo = a1.agent.evaluate(X, act)[0].clone()
        r1 = a1.update_agent(X, act, adv, ret, lpo, w)
        r2 = a2.update_agent(X, act, adv, ret, lpo, w)
        assert r1.stop_iteration == r2.stop_iteration, (r1, r2)
        assert (r1.stop_iteration < 8) == expect_early
        assert abs(r1.kl_divergence - r2.kl_divergence) < 1e-5 and abs(r1.loss_policy - r2.loss_policy) < 1e-4
        for p1, p2 in zip(a1.agent.parameters(), a2.agent.parameters()):
            assert torch.allclose(p1, p2, rtol=1e-4, atol=2e-5), (p1 - p2).abs().max()


def test_obstacle_training_stays_finite():
    """Regression: 1-5 random obstacles per env (reference default obstruction_count=-1).  A layout with a
    nested rectangle used to slip through reset and could yield an unreachable detector (reward -inf, NaN KL
    from the 4th PPO iteration on); world.is_valid (rad_search_env.py:788-791) now rejects it at reset."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.ppo import FusedCollector, VecAgentPPO
    env = RadSearchVec(8192, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
    agents = {0: VecAgentPPO(id=0, steps_per_epoch=240, steps_per_episode=120, alpha=0.1)}
    col = FusedCollector(env, agents, 240, 120)
    for _ in range(3):
        col.collect()
        for name in ("obs", "rew", "val", "adv", "ret"):
            assert torch.isfinite(getattr(col.buf, name)).all(), name
        r = col.update()[0]
        assert np.isfinite(r.kl_divergence) and np.isfinite(r.loss_policy)
    assert env.error_flags() == 0
    num = env.state("num_obs")[0]
    rect = env.state("rect").view(7, 4, -1)[:5]                       # [obstacle, (x0,y0,x1,y1), env]
    live = torch.arange(5, device=num.device)[:, None] < num[None, :]
    for i in range(5):
        for k in range(5):
            if i != k:
                nested = ((rect[k, 0] <= rect[i, 0]) & (rect[i, 0] <= rect[k, 2]) & (rect[k, 1] <= rect[i, 1])
                          & (rect[i, 1] <= rect[k, 3]) & live[i] & live[k])
                assert not nested.any()


def test_fused_grad_and_adam_step_match_reference_update_rada2c(golden_dir):
    """rs_ppo_grad + rs_adam_step against the reference's own update_rada2c (algos/multiagent/ppo.py:1150-1281) run on
    the FF_core network (tests/golden/rada2c_loss.npz): loss, approx-KL, entropy, clip fraction, value loss, every
    parameter gradient, the KL early-stop decision and the parameters after the Adam step.
    Tolerance: fp32, different summation order -- gradients rtol 2e-4 / atol 2e-7, scalars 2e-6."""
    from radiation_ppo_amd.ppo import FFActorCritic, FusedPPOGrad
    d = np.load(os.path.join(golden_dir, "rada2c_loss.npz"))
    names = ["actor.0.weight", "actor.0.bias", "actor.2.weight", "actor.2.bias", "actor.4.weight", "actor.4.bias",
             "critic.0.weight", "critic.0.bias", "critic.2.weight", "critic.2.bias", "critic.4.weight", "critic.4.bias"]
    for tag in ("step", "stop"):
        ac = FFActorCritic().cuda()
        ac.load_state_dict({k: torch.from_numpy(d[f"{tag}_before_{k}"]) for k in names})
        eps = [torch.from_numpy(d[f"{tag}_ep{i}"]) for i in range(int(d["n_eps"]))]
        E = len(eps)
        cat = torch.cat(eps).cuda()
        X, adv, ret, lpo = (cat[:, :11].contiguous(), cat[:, 11].contiguous(), cat[:, 12].contiguous(), cat[:, 13].contiguous())
        act = cat[:, 14].long().contiguous()
        w = torch.cat([torch.full((e.shape[0],), 1.0 / (E * e.shape[0])) for e in eps]).cuda()
        f = FusedPPOGrad(ac)
        f.begin_update()
        stats, grads = f(X, act, adv, ret, lpo, w, 0.2, 0.1, use_stop_flag=True)
        st = stats.cpu().numpy()
        assert abs(st[0] - float(d[f"{tag}_kl"])) < 2e-6 and abs(st[1] - float(d[f"{tag}_ent"])) < 2e-6
        assert abs(st[2] - float(d[f"{tag}_cf"])) < 2e-6 and abs(st[3] - float(d[f"{tag}_val_loss"])) < 2e-6
        assert abs(st[4] - float(d[f"{tag}_loss"])) < 2e-6
        term = bool(d[f"{tag}_term"])
        if not term:
            for (p, g), k in zip(f.views, names):
                assert np.allclose(g.cpu().numpy(), d[f"{tag}_grad_{k}"], rtol=2e-4, atol=2e-7), (tag, k)
        f.adam_step(3e-4, 1.5 * 0.07)
        iters, stopped, adam_t, _ = f.read_state()
        assert bool(stopped) == term and adam_t == (0 if term else 1)
        for k, p in zip(names, [dict(ac.named_parameters())[k] for k in names]):
            got, want = p.detach().cpu().numpy(), d[f"{tag}_after_{k}"]
            big = np.abs(d[f"{tag}_grad_{k}"]) > 1e-5 if not term else np.ones_like(want, dtype=bool)
            assert np.allclose(got[big], want[big], rtol=0, atol=3e-6), (tag, k, np.abs(got - want)[big].max())
            assert np.abs(got - want).max() <= 6.1e-4, (tag, k)


def test_welford_kernels_equal_the_float64_composition():
    """rs_welford_update / _reset / _standardize (DeviceWelford on the GPU) against the element-wise float64 composition
    (impl="torch") on the same device: bitwise, over masked updates, resets and a strided reading column (obs[..., 0] of
    [N, A, 11]); against the composition on the CPU (another sqrt / division implementation) to the last bits."""
    from radiation_ppo_amd.ppo import DeviceWelford
    g = torch.Generator().manual_seed(5)
    N, A = 301, 3
    dev, cpu, host = DeviceWelford((N, A), "cuda"), DeviceWelford((N, A), "cuda", impl="torch"), DeviceWelford((N, A), "cpu")
    for step in range(12):
        obs = (torch.rand(N, A, 11, generator=g) * (50.0 if step % 3 else 5000.0)).float()
        mask = (torch.rand(N, generator=g) < 0.6) if step % 2 else None
        og = obs.cuda()
        mg = None if mask is None else mask.cuda()
        dev.update(og[..., 0], mg); cpu.update(og[..., 0], mg); host.update(obs[..., 0], mask)
        if step in (4, 9):
            rm = torch.rand(N, generator=g) < 0.3
            dev.reset(rm.cuda()); cpu.reset(rm.cuda()); host.reset(rm)
        for name in ("count", "mean", "sq", "std"):
            a, b, c = getattr(dev, name), getattr(cpu, name), getattr(host, name)
            assert torch.equal(a, b), (step, name)
            assert torch.allclose(a.cpu(), c, rtol=1e-14, atol=0.0), (step, name)
        x = og.clone()
        z = dev.standardize(og[..., 0], out=x[..., 0])
        assert torch.equal(x[..., 0], cpu.standardize(og[..., 0])) and torch.equal(x[..., 1:], og[..., 1:]) and z.data_ptr() == x.data_ptr()
        assert torch.equal(dev.standardize(og[..., 0]), cpu.standardize(og[..., 0]))
    one, ref = DeviceWelford((N,), "cuda"), DeviceWelford((N,), "cuda", impl="torch")      # the 1-D form (FusedCollector.start)
    r = (torch.rand(N, generator=g) * 100).cuda()
    one.update(r); ref.update(r)
    one.update(r * 2); ref.update(r * 2)
    assert torch.equal(one.mean, ref.mean) and torch.equal(one.std, ref.std)


def test_epoch_stats_kernel_equals_the_reductions():
    """rs_epoch_stats (EpochStats.step_and_episodes: one launch per lock-step) against the step() + episodes() reductions: counts,
    extrema and episode counters exactly, float64 sums to summation-order rounding; a no-episode step leaves max / min at +-inf."""
    from radiation_ppo_amd.ppo import EpochStats
    g = torch.Generator().manual_seed(6)
    for N, A in ((1000, 1), (333, 4)):
        k, t = EpochStats(A, "cuda"), EpochStats(A, "cuda")
        for step in range(6):
            oob = (torch.rand(N, A, generator=g) < 0.2).to(torch.uint8).cuda()
            done = (torch.rand(N, A, generator=g) < 0.1).to(torch.uint8).cuda()
            ret = (torch.randn(N, A, generator=g) * 30).float().cuda()
            steps = torch.randint(1, 121, (N,), generator=g, dtype=torch.int32).cuda()
            over = (torch.rand(N, generator=g) < (0.0 if step == 0 else 0.15)).cuda()
            k.step_and_episodes(oob, done, ret, steps, over)
            t.step(oob, done); t.episodes(ret, steps, over)
            a, b = k.result(), t.result()
            for name in ("DoneCount", "OutOfBound", "EpCount", "EpLenSum", "EpRetMax", "EpRetMin"):
                assert torch.equal(a[name], b[name]), (N, A, step, name)
            for name in ("EpRetSum", "EpRetSqSum"):
                assert torch.allclose(a[name], b[name], rtol=1e-13, atol=1e-9), (N, A, step, name)
