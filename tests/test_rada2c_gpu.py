"""Row f2 on the GPU: the RAD-A2C ('rnn') collector and trainer.  The collector is replayed env by env through the pinned
train-loop oracle (oracle/train_loop_oracle.train_loop_trace, RAD-A2C branch) exactly like the MLP collectors; the networks'
arithmetic is pinned on the CPU (tests/test_rada2c_golden.py); here: the stored log-probabilities / values are those of the
carried GRU + K11 particle states, hidden states restart at episode boundaries, and train_PPO runs the architecture end to end."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from test_ppo_gpu import SEED, _replay_check  # noqa: E402

pytestmark = pytest.mark.gpu


def _make(N, T, L, obst=0, seed=SEED, base=0):
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector
    torch.manual_seed(4)
    env = RadSearchVec(N, number_agents=1, obstruction_count=obst, enforce_grid_boundaries=True, seed=seed, env_id_base=base)
    agents = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, train_pi_iters=3, train_pfgru_iters=2, seed=3)}
    with torch.no_grad():
        for p in agents[0].agent.pi.parameters():
            p.mul_(2.0)                              # make the policy visibly non-uniform
    return env, agents, RNNCollector(env, agents, T, L)


@pytest.mark.parametrize("obst", [0, 2])
def test_rnn_collector_replays_through_oracle(obst):
    N, T, L = 64, 40, 12
    env, agents, col = _make(N, T, L, obst)
    col.collect()
    _replay_check(col, agents, N, T, L, obst, stride=5)
    buf = col.buf
    assert torch.isfinite(buf.logp).all() and (buf.logp <= 0).all() and torch.isfinite(buf.val).all()
    assert int(buf.act.min()) >= 0 and int(buf.act.max()) <= 7


def test_rnn_collector_state_semantics():
    """Replay the stored observations through the agent's own modules with a twin particle bank: the stored log-probabilities
    and values are reproduced only when the GRU / particle states are carried inside an episode and restarted at every cut."""
    from radiation_ppo_amd.pfgru import PredictorBank, _s64, hash_uniform
    N, T, L = 32, 30, 9
    env, agents, col = _make(N, T, L, base=64)
    col.collect()
    ac = agents[0].agent
    buf = col.buf
    twin = PredictorBank(N, 1, seed=SEED, env_id_base=64, carry_hidden=True, device="cuda")
    twin.cells[0] = ac.model
    begun = torch.zeros(N, dtype=torch.int64, device="cuda")
    h = torch.zeros(N, 24, device="cuda")
    gidx = torch.arange(24, dtype=torch.int64, device="cuda")

    def restart(mask):
        nonlocal begun, h
        twin.reset(mask)
        m = torch.ones(N, dtype=torch.bool, device="cuda") if mask is None else mask
        begun = begun + m.long()
        key = (twin._base * 1000003) ^ ((begun.view(1, -1) * 8 + 5) * _s64(0xA24BAED4963EE407))
        h0 = ac.gru_h0(hash_uniform(key.unsqueeze(-1) * 1048583 + gidx.view(1, 1, -1)))[0]
        h = torch.where(m.view(-1, 1), h0, h)
    restart(None)
    with torch.no_grad():
        for t in range(T):
            x = buf.obs[t]
            loc = twin.predict(x)
            logits, v, h = ac.policy_step(x[:, 0], loc[:, 0], h)
            lp = torch.log_softmax(logits, -1).gather(-1, buf.act[t, :, 0].unsqueeze(-1)).squeeze(-1)
            assert torch.allclose(lp, buf.logp[t, :, 0], rtol=1e-5, atol=1e-6), t
            assert torch.allclose(v, buf.val[t, :, 0], rtol=1e-5, atol=1e-6), t
            cut = buf.cut[t, :, 0].bool()
            boot = cut & (buf.last_val[t, :, 0] != 0)
            twin.calls += boot.long()                    # the collector's bootstrap step consumed one prediction
            if t < T - 1:
                restart(cut)
    # an epoch's first step starts from fresh hidden states too (train.py:322-329)
    col.collect()
    assert torch.isfinite(col.buf.val).all()


def test_rnn_collector_is_sharding_invariant():
    N, T, L = 32, 24, 8
    full = _make(N, T, L)[2]
    half = _make(N // 2, T, L, base=N // 2)[2]
    full.collect(); half.collect()
    for name in ("obs", "act", "rew", "cut"):
        a, b = getattr(full.buf, name)[:, N // 2:], getattr(half.buf, name)
        assert torch.equal(a, b), name
    assert torch.allclose(full.buf.logp[:, N // 2:], half.buf.logp, rtol=1e-5, atol=1e-6)


def test_train_ppo_rnn_end_to_end(tmp_path):
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.rada2c import RNNModelActorCritic
    from radiation_ppo_amd.train import train_PPO
    vec = RadSearchVec(64, number_agents=1, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED)
    sim = train_PPO(env=vec, logger_kwargs=dict(output_dir=str(tmp_path)), seed=2, number_of_agents=1, actor_critic_architecture="rnn",
                    global_critic_flag=False, steps_per_epoch=36, steps_per_episode=12, total_epochs=2, save_freq=1,
                    ppo_kwargs=dict(train_pi_iters=3, train_pfgru_iters=2, alpha=0.1))
    before = torch.cat([p.detach().reshape(-1).clone() for p in sim.agents[0].agent.parameters()])
    sim.train()
    after = torch.cat([p.detach().reshape(-1) for p in sim.agents[0].agent.parameters()])
    assert not torch.equal(before, after) and torch.isfinite(after).all()
    rows = sim.loggers[0].rows
    assert len(rows) == 2
    for r in rows:
        for k in ("loss_policy", "loss_critic", "loss_predictor", "LocLoss", "kl_divergence", "Entropy", "MeanVVals"):
            assert np.isfinite(float(r[k])), (k, r[k])
        assert float(r["loss_predictor"]) > 0 and 1 <= int(r["stop_iteration"]) <= 3
    # the saved module loads into the reference-shaped network (pyt_save/model.pt, epoch_logger.py:216-284)
    sd = torch.load(os.path.join(str(tmp_path), "0_agent", "pyt_save", "model.pt"), map_location="cpu")
    RNNModelActorCritic().load_state_dict(sd)
    with pytest.raises(Exception, match="global critic"):
        from radiation_ppo_amd.rada2c import RNNCollector
        RNNCollector(vec, sim.agents, 36, 12, global_critic_flag=True)


def test_gru_sequence_kernels_match_torch_gru():
    """K12 (rs_gru_forward / rs_gru_backward behind rada2c.GRUSequence) against torch.nn.GRU and its autograd: hidden states of
    every step and all four parameter gradients (float32, different summation order: rtol 1e-4)."""
    from radiation_ppo_amd.rada2c import GRUSequence
    torch.manual_seed(2)
    L, E = 37, 333                                   # not multiples of the wave size
    gru = torch.nn.GRU(13, 24, 1).cuda()
    x = torch.randn(L, E, 13, device="cuda")
    h0 = (torch.rand(E, 24, device="cuda") * 2 - 1) * 0.2
    wgt = torch.randn(L, E, 24, device="cuda") * (torch.rand(L, E, 1, device="cuda") < 0.7)      # zero on some steps, like padding
    with torch.backends.cudnn.flags(enabled=False):
        ref, _ = gru(x, h0.unsqueeze(0))
    (ref * wgt).sum().backward()
    want = {k: p.grad.clone() for k, p in gru.named_parameters()}
    gru.zero_grad()
    got = GRUSequence.apply(x, h0, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5), float((got - ref).abs().max())
    (got * wgt).sum().backward()
    for k, p in gru.named_parameters():
        assert torch.allclose(p.grad, want[k], rtol=1e-4, atol=1e-3 * float(want[k].abs().max())), (k, float((p.grad - want[k]).abs().max()))


def test_rnn_collector_graph_replay_equals_eager_steps():
    """The captured lock-step replayed T - 1 times writes what the eager loop writes -- also AFTER the agent changed: an update
    (update_model moves the PFGRU, update_rada2c the policy) runs between the two epochs, so a replay that kept reading weight
    buffers packed before the update (K11's per-owner pack, K14's policy pack) would diverge from the eager loop."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector
    N, T, L = 48, 26, 8

    def run(use_graph):
        torch.manual_seed(4)
        env = RadSearchVec(N, number_agents=1, obstruction_count=2, enforce_grid_boundaries=True, seed=SEED, env_id_base=16)
        agents = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, seed=3, train_pi_iters=2, train_pfgru_iters=2)}
        col = RNNCollector(env, agents, T, L, use_graph=use_graph)
        out = []
        for ep in range(2):
            st = col.collect()
            out.append({**{k: getattr(col.buf, k).clone() for k in ("obs", "act", "rew", "val", "logp", "last_val", "cut", "adv", "ret", "source_tar")},
                        **{"stat_" + k: v.clone() for k, v in st.items()}})
            if ep == 0:
                before = torch.cat([p.detach().reshape(-1).clone() for p in agents[0].agent.parameters()])
                col.update()
                with torch.no_grad():                                   # and a visible move of both modules on top of the Adam steps
                    for p in agents[0].agent.model.parameters():
                        p.mul_(1.25)
                    for p in agents[0].agent.pi.parameters():
                        p.mul_(1.5)
                after = torch.cat([p.detach().reshape(-1) for p in agents[0].agent.parameters()])
                assert not torch.equal(before, after)
        assert (col._graph is not None) == use_graph
        return out
    g, e = run(True), run(False)
    assert not torch.equal(e[0]["val"], e[1]["val"])
    for ep in range(2):
        for k in e[ep]:
            assert torch.equal(g[ep][k], e[ep][k]), (ep, k)


def test_kernel_draws_equal_hash_draws():
    """rs_pfgru_draws (one launch per training pass) against the torch composition of the same counter hash (rada2c.HashDraws)."""
    from radiation_ppo_amd.rada2c import HashDraws, KernelDraws
    keys = (torch.arange(257, dtype=torch.int64, device="cuda") * 7919 + 12345) * 64 + 3
    L = 9
    kd, hd = KernelDraws(keys, L), HashDraws(keys)
    assert torch.equal(kd.pf_h0(), hd.pf_h0())
    for t in range(L):
        assert torch.equal(kd.resample(t)["resample_u"], hd.resample(t)["resample_u"]), t
        assert torch.allclose(kd.eps(t), hd.eps(t), rtol=2e-6, atol=2e-6), (t, float((kd.eps(t) - hd.eps(t)).abs().max()))


def test_policy_update_with_k12_equals_library_gru_path():
    """One update_rada2c pass with the GRU recurrence on K12 against the same pass on torch.nn.GRU + autograd (same draws, same
    batch): statistics and every pi gradient."""
    from radiation_ppo_amd.rada2c import HashDraws, RNNAgentPPO, pack_episodes
    g = torch.Generator().manual_seed(9)
    T, N = 60, 96
    obs = torch.rand(T, N, 11, generator=g).cuda()
    act = torch.randint(0, 8, (T, N), generator=g).cuda()
    adv, ret = torch.randn(T, N, generator=g).cuda(), torch.randn(T, N, generator=g).cuda()
    logp = (float(np.log(1 / 8)) + 0.05 * torch.randn(T, N, generator=g)).cuda()
    src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
    cut = (torch.rand(T, N, generator=g) < 0.08).to(torch.uint8)
    cut[-1] = 1
    B = pack_episodes(obs, act, adv, ret, logp, src, cut.cuda(), n_total=N, seed=3)
    out = []
    for k12 in (True, False):
        torch.manual_seed(21)
        ag = RNNAgentPPO(id=0, seed=1, alpha=0.1)
        ag.use_k12 = k12
        ag.agent.train()
        ag.pi_optimizer.zero_grad(set_to_none=True)
        loss, st = ag.a2c_losses(B, slice(0, B.lens.shape[0]), HashDraws(B.key * 64 + 17))
        loss.backward()
        out.append((st.clone(), {k: p.grad.clone() for k, p in ag.agent.pi.named_parameters()}))
    assert torch.allclose(out[0][0], out[1][0], rtol=1e-5, atol=1e-7), (out[0][0], out[1][0])
    for k in out[0][1]:
        a, b = out[0][1][k], out[1][1][k]
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * float(b.abs().max()) + 1e-9), (k, float((a - b).abs().max()), float(b.abs().max()))


class _KernelIdxDraws:
    """The draws of a KernelDraws object with the resampling INDICES the kernel took (the discrete choices are then the same on
    both sides; the uniforms behind them are checked by the forward comparison of the indices themselves)."""

    def __init__(self, kd, idx):
        self.kd, self.idx = kd, idx

    def pf_h0(self):
        return self.kd.pf_h0()

    def eps(self, t):
        return self.kd.eps(t)

    def resample(self, t):
        return dict(resample_idx=self.idx[t].long())


@pytest.mark.parametrize("l1", [0.0, 0.5])
def test_pfgru_training_kernel_matches_autograd(l1):
    """K13 (rs_pfgru_train: episode loop + loss + back-propagation through time in one launch) against model_loss + autograd on the
    torch cell: loss, every parameter gradient, and the resampling indices against the inverse-CDF of the uniforms."""
    from radiation_ppo_amd.rada2c import BpArgs, KernelDraws, RNNAgentPPO, pack_episodes, unpack_train_grads
    g = torch.Generator().manual_seed(4)
    T, N = 40, 24
    obs = torch.rand(T, N, 11, generator=g).cuda()
    act = torch.randint(0, 8, (T, N), generator=g).cuda()
    z = torch.zeros(T, N).cuda()
    src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
    cut = (torch.rand(T, N, generator=g) < 0.08).to(torch.uint8)
    cut[-1] = 1
    cut[0, 3] = 1                                                       # a one-step episode
    B = pack_episodes(obs, act, z, z, z, src, cut.cuda(), n_total=N, seed=3)
    E = B.lens.shape[0]
    torch.manual_seed(5)
    ag = RNNAgentPPO(id=0, seed=1, bp_args=BpArgs(l1_weight=l1, area_scale=2500.0))
    cell = ag.agent.model
    with torch.no_grad():                                                # away from the initialisation: every path carries signal
        for p in cell.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g).cuda())
    sl = slice(0, E)
    kd = KernelDraws(B.key * 64 + 1, B.X.shape[0])
    loss_k, slab, idx = ag.model_pass_hip(B, sl, kd)
    gk = unpack_train_grads(cell, slab)
    loss_2, slab_2, idx_2 = ag.model_pass_hip(B, sl, kd)                # the LDS scatter-add is wave-private: bitwise repeatable
    assert torch.equal(slab, slab_2) and torch.equal(idx, idx_2) and float(loss_k) == float(loss_2)
    cell.train()
    for p in cell.parameters():
        p.grad = None
    loss_t = ag.model_loss(B, sl, _KernelIdxDraws(kd, idx))
    loss_t.backward()
    lk, lt = float(loss_k), float(loss_t.detach())
    assert abs(lk - lt) <= 2e-5 * abs(lt), (lk, lt)
    gmax = max(float(p.grad.abs().max()) for p in cell.parameters())
    for name, p in cell.named_parameters():
        a, b = gk[name], p.grad
        # fc_obs.bias shifts every particle's logit alike and cancels in the log-softmax: its gradient is exactly 0 in exact arithmetic,
        # rounding noise (~1e-10) on both sides -- hence the floor relative to the largest gradient of the cell
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-4 * float(b.abs().max()) + 1e-6 * gmax), (name, float((a - b).abs().max()), float(b.abs().max()), gmax)
    # the indices themselves: the torch cell resampling from the same uniforms takes the same particles (valid steps)
    cell.zero_grad(set_to_none=True)
    with torch.no_grad():
        h, p = kd.pf_h0(), torch.full((E, 40), float(np.log(1 / 40)), dtype=torch.float32, device="cuda")
        same = tot = 0
        for t in range(B.X.shape[0]):
            X3 = B.X[t, :, :3]
            pre_h, pre_p = h, p
            _, (h, p) = cell(X3, (pre_h, pre_p), kd.eps(t), **kd.resample(t))
            ref_h = torch.gather(_pre_resample(cell, X3, pre_h, pre_p, kd.eps(t)), 1, idx[t].long().unsqueeze(-1).expand(E, 40, 24))
            ok = ((ref_h - h).abs().amax(dim=(1, 2)) < 1e-6) & B.valid[t]
            same += int(ok.sum()); tot += int(B.valid[t].sum())
            # continue from the kernel's choice so that one flipped index (a uniform within rounding of a CDF step) cannot cascade
            _, (h, p) = cell(X3, (pre_h, pre_p), kd.eps(t), resample_idx=idx[t].long())
        assert same >= tot - 2, (same, tot)


def test_pfgru_training_kernel_with_hashed_draws_is_bit_identical():
    """rs_pfgru_train_keyed (the forward walk evaluates the counter hash itself: KeyDraws) against rs_pfgru_train fed the buffers
    rs_pfgru_draws wrote for the same keys: same loss, same gradient slabs, same resampling indices, bit for bit -- ragged episodes,
    a one-step episode, more episodes than one workgroup holds."""
    from radiation_ppo_amd.rada2c import BpArgs, KeyDraws, RNNAgentPPO, pack_episodes
    g = torch.Generator().manual_seed(11)
    T, N = 48, 40
    obs = torch.rand(T, N, 11, generator=g).cuda()
    act = torch.randint(0, 8, (T, N), generator=g).cuda()
    z = torch.zeros(T, N).cuda()
    src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
    cut = (torch.rand(T, N, generator=g) < 0.06).to(torch.uint8)
    cut[-1] = 1
    cut[0, 5] = 1
    B = pack_episodes(obs, act, z, z, z, src, cut.cuda(), n_total=N, seed=7, sort_by_length=True)
    E = B.lens.shape[0]
    ag = RNNAgentPPO(id=0, seed=3, bp_args=BpArgs(l1_weight=0.5, area_scale=2500.0))
    sl = slice(0, E)
    kd = KeyDraws(B.key * 64 + 1 + 4)
    loss_a, slab_a, idx_a = ag.model_pass_hip(B, sl, kd)
    slab_a, idx_a = slab_a.clone(), idx_a.clone()
    loss_b, slab_b, idx_b = ag.model_pass_hip(B, sl, kd.materialise(B.X.shape[0]))
    assert float(loss_a) == float(loss_b)
    assert torch.equal(idx_a, idx_b)
    assert torch.equal(slab_a, slab_b)


def _pre_resample(cell, X3, h0, p0, eps):
    keep = cell.use_resampling
    cell.use_resampling = False
    try:
        _, (h1, _) = cell(X3, (h0, p0), eps)
    finally:
        cell.use_resampling = keep
    return h1


def test_update_model_on_k13_equals_autograd_path():
    """update_model end to end (draws, K13 passes, clip, Adam) against the same call on the torch-autograd path: the PFGRU's
    parameters after two iterations.  The autograd path is handed the resampling INDICES the kernel took (as in
    test_pfgru_training_kernel_matches_autograd: they are constants of the backward pass on both sides), so no discrete choice can
    differ and EVERY parameter must agree within a quarter of an Adam step."""
    from radiation_ppo_amd.rada2c import KernelDraws, RNNAgentPPO, pack_episodes
    g = torch.Generator().manual_seed(8)
    T, N = 30, 40
    obs = torch.rand(T, N, 11, generator=g).cuda()
    act = torch.randint(0, 8, (T, N), generator=g).cuda()
    z = torch.zeros(T, N).cuda()
    src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
    cut = (torch.rand(T, N, generator=g) < 0.1).to(torch.uint8)
    cut[-1] = 1
    B = pack_episodes(obs, act, z, z, z, src, cut.cuda(), n_total=N, seed=3)
    res, taken = [], {}
    for k13 in (True, False):
        torch.manual_seed(6)
        ag = RNNAgentPPO(id=0, seed=1, train_pfgru_iters=2, episode_chunk=64)        # several chunks
        ag.use_k13 = k13
        if k13:
            calls = []
            orig = ag.model_pass_hip

            def recording(Bx, sl, d, _orig=orig, _calls=calls):
                out = _orig(Bx, sl, d)
                _calls.append((sl.start, out[2].clone()))
                return out
            ag.model_pass_hip = recording
            loss = ag.update_model(B)
            n_chunks = len(calls) // 2
            for i, (lo, idx) in enumerate(calls):
                taken[(i // n_chunks, lo)] = idx
        else:
            def draws_for(it, sl):
                kd = KernelDraws(B.key[sl] * 64 + 1 + it, B.chunk(sl).X.shape[0])
                return _KernelIdxDraws(kd, taken[(it, sl.start)])
            loss = ag.update_model(B, draws_for=draws_for)
        res.append((loss, {k: v.detach().clone() for k, v in ag.agent.model.named_parameters()},
                    {k: v.grad.detach().clone() for k, v in ag.agent.model.named_parameters()}))
    assert abs(res[0][0] - res[1][0]) <= 1e-3 * abs(res[1][0]), (res[0][0], res[1][0])
    lr = 5e-3
    gmax = max(float(v.abs().max()) for v in res[1][2].values())
    n_noise = 0
    for k in res[0][1]:
        d = (res[0][1][k] - res[1][1][k]).abs()
        # an Adam step is lr * m / (sqrt(v) + 1e-8): where the gradient is float32 rounding noise on both sides (fc_obs.bias: exactly
        # 0 in exact arithmetic, see test_pfgru_training_kernel_matches_autograd) the step's sign is noise as well -- those
        # elements, identified by their gradient, are counted and held to two full steps; everything else to a quarter step
        noise = res[1][2][k].abs() <= 1e-6 * gmax
        n_noise += int((noise & (d > 0.25 * lr)).sum())
        assert bool((d[~noise] <= 0.25 * lr).all()), (k, float(d[~noise].max()))
        assert bool((d <= 2 * 2 * lr).all()), (k, float(d.max()))
    assert n_noise <= 2, n_noise                        # elements beyond a quarter step: fc_obs.bias and at most one more noise element


def test_policy_step_kernel_matches_torch_composition():
    """K14 (rs_rnn_policy_step: GRU cell + heads + inverse-CDF draw in one launch) against RNNModelActorCritic.policy_step and the
    collector's torch sampling: new state, logits, value, action, log-probability; in-place state update; re-packing after a
    parameter change."""
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    torch.manual_seed(12)
    ag = RNNAgentPPO(id=0, seed=1)
    with torch.no_grad():
        for p in ag.agent.pi.parameters():
            p.mul_(1.7)
    N = 1000                                                             # not a multiple of 64
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, 11, generator=g).cuda()
    loc = torch.rand(N, 2, generator=g).cuda()
    h = (torch.rand(N, 24, generator=g) * 0.4 - 0.2).cuda()
    u = torch.rand(N, generator=g).cuda()
    for rep in range(2):
        with torch.no_grad():
            logits_t, v_t, h_t = ag.agent.policy_step(x, loc, h)
            lp_all = torch.log_softmax(logits_t, dim=-1)
            cdf = torch.cumsum(lp_all.exp(), dim=-1)
            act_t = (cdf[:, :-1] <= u.unsqueeze(-1)).sum(dim=-1)
        hk = h.clone()
        logits = torch.empty(N, 8, device="cuda"); v = torch.empty(N, device="cuda"); lp = torch.empty(N, device="cuda")
        act = torch.empty(N, dtype=torch.int64, device="cuda")
        ag.policy_step_hip(x, loc, hk, u=u, h_out=hk, logits=logits, value=v, act=act, logp=lp)
        assert torch.allclose(hk, h_t, rtol=1e-5, atol=2e-6), float((hk - h_t).abs().max())
        assert torch.allclose(logits, logits_t, rtol=1e-5, atol=5e-6) and torch.allclose(v, v_t, rtol=1e-5, atol=5e-6)
        edge = (cdf[:, :-1] - u.unsqueeze(-1)).abs().amin(dim=1) < 1e-5   # a uniform within rounding of a CDF step may fall either side
        assert bool(((act == act_t) | edge).all()) and int((act != act_t).sum()) <= 2
        same = act == act_t
        assert torch.allclose(lp[same], lp_all.gather(-1, act_t.unsqueeze(-1)).squeeze(-1)[same], rtol=1e-5, atol=5e-6)
        vb = torch.empty(N, device="cuda")
        ag.policy_step_hip(x, loc, h, value=vb)                          # the bootstrap form: value only, state untouched
        assert torch.allclose(vb, v_t, rtol=1e-5, atol=5e-6)
        with torch.no_grad():                                            # an optimiser step bumps the versions: the pack follows
            for p in ag.agent.pi.parameters():
                p.add_(0.01)


def test_gru_h0_reset_kernel_equals_the_hash_composition():
    """rs_gru_h0_reset (one launch) against the torch composition of the same counter hash in RNNCollector._reset_hidden: bitwise,
    masked envs only."""
    env, agents, col = _make(70, 8, 4)
    col.start()
    g = torch.Generator().manual_seed(3)
    for rep in range(3):
        mask = (torch.rand(70, generator=g) < 0.5).cuda() if rep else None
        begun = col.episodes_begun.clone()
        h_before = col.h.clone()
        pf = (col.bank.h.clone(), col.bank.p.clone(), col.bank.episode.clone(), col.bank.calls.clone())
        assert col.use_k14
        col._reset_hidden(mask)
        got = col.h.clone()
        # rewind and take the composition
        col.episodes_begun.copy_(begun); col.h.copy_(h_before)
        col.bank.h = pf[0]; col.bank.p.copy_(pf[1]); col.bank.episode.copy_(pf[2]); col.bank.calls.copy_(pf[3])
        col.use_k14 = False
        col._reset_hidden(mask)
        col.use_k14 = True
        assert torch.equal(got, col.h), rep
        if mask is not None:
            assert torch.equal(got[:, ~mask], h_before[:, ~mask]) and not torch.equal(got[:, mask], h_before[:, mask])


def test_policy_update_with_k15_equals_the_torch_heads():
    """One update_rada2c pass with the heads + loss + their back-propagation on K15 (rs_a2c_heads_loss) against the same pass with
    the torch heads and autograd: statistics and every pi gradient (heads AND, through dL/dh, the GRU)."""
    from radiation_ppo_amd.rada2c import HashDraws, RNNAgentPPO, pack_episodes
    g = torch.Generator().manual_seed(19)
    T, N = 60, 150
    obs = torch.rand(T, N, 11, generator=g).cuda()
    act = torch.randint(0, 8, (T, N), generator=g).cuda()
    adv, ret = torch.randn(T, N, generator=g).cuda(), torch.randn(T, N, generator=g).cuda()
    logp = (float(np.log(1 / 8)) + 0.25 * torch.randn(T, N, generator=g)).cuda()          # wide enough to hit both clip sides
    src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
    cut = (torch.rand(T, N, generator=g) < 0.08).to(torch.uint8)
    cut[-1] = 1
    B = pack_episodes(obs, act, adv, ret, logp, src, cut.cuda(), n_total=N, seed=3, sort_by_length=True)
    out = []
    for k15 in (True, False):
        torch.manual_seed(22)
        ag = RNNAgentPPO(id=0, seed=1, alpha=0.1)
        with torch.no_grad():
            for p in ag.agent.pi.parameters():
                p.mul_(1.5)
        ag.use_k15 = k15
        ag.agent.train()
        ag.pi_optimizer.zero_grad(set_to_none=True)
        loss, st = ag.a2c_losses(B, slice(0, B.lens.shape[0]), HashDraws(B.key * 64 + 17))
        loss.backward()
        out.append((st.clone(), {k: p.grad.clone() for k, p in ag.agent.pi.named_parameters()}))
    assert float(out[1][0][2]) > 0.01                                     # some samples are clipped
    assert torch.allclose(out[0][0], out[1][0], rtol=2e-5, atol=1e-7), (out[0][0], out[1][0])
    for k in out[0][1]:
        a, b = out[0][1][k], out[1][1][k]
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4 * float(b.abs().max()) + 1e-9), (k, float((a - b).abs().max()), float(b.abs().max()))


def test_update_agent_with_prefetched_pfgru_passes_equals_the_serial_schedule():
    """update_agent with the PFGRU passes of the next policy iteration enqueued on a side stream (loc_prefetch) against the same
    update with everything on one stream: same draws, same arithmetic -> identical parameters and statistics."""
    from radiation_ppo_amd.rada2c import RNNAgentPPO, pack_episodes
    g = torch.Generator().manual_seed(23)
    T, N = 48, 80
    obs = torch.rand(T, N, 11, generator=g).cuda()
    act = torch.randint(0, 8, (T, N), generator=g).cuda()
    adv, ret = torch.randn(T, N, generator=g).cuda(), torch.randn(T, N, generator=g).cuda()
    logp = (float(np.log(1 / 8)) + 0.05 * torch.randn(T, N, generator=g)).cuda()
    src = (torch.rand(T, N, 2, generator=g) * 2000 + 200).cuda()
    cut = (torch.rand(T, N, generator=g) < 0.1).to(torch.uint8)
    cut[-1] = 1
    B = pack_episodes(obs, act, adv, ret, logp, src, cut.cuda(), n_total=N, seed=3, sort_by_length=True)
    res = []
    for pre in (True, False):
        torch.manual_seed(31)
        ag = RNNAgentPPO(id=0, seed=1, train_pi_iters=4, train_pfgru_iters=1, episode_chunk=128)
        ag.use_prefetch = pre
        r = ag.update_agent(B)
        torch.cuda.synchronize()
        res.append((r, torch.cat([p.detach().reshape(-1) for p in ag.agent.parameters()])))
    a, b = res
    assert a[0].stop_iteration == b[0].stop_iteration
    for k in ("loss_policy", "loss_critic", "kl_divergence", "Entropy", "ClipFrac", "LocLoss"):
        assert getattr(a[0], k) == getattr(b[0], k), (k, getattr(a[0], k), getattr(b[0], k))
    assert torch.equal(a[1], b[1])


def test_non_default_layer_sizes_train_on_the_gpu():
    """RAD-A2C with layer sizes other than the reference's defaults: collector and update run through the library-op composition on the
    GPU (K11 - K15 are built for GRU(13, 24), 32-unit heads and a 24-unit PFGRU; the env and the statistics kernels are shared)."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector
    N, T, L = 24, 16, 6
    torch.manual_seed(5)
    env = RadSearchVec(N, number_agents=1, obstruction_count=1, enforce_grid_boundaries=True, seed=SEED)
    args = dict(hidden=((16,),), hidden_sizes_pol=((20, 12),), hidden_sizes_val=((10,),), hidden_sizes_rec=(12,))
    agents = {0: RNNAgentPPO(id=0, steps_per_epoch=T, steps_per_episode=L, train_pi_iters=2, train_pfgru_iters=2, seed=3, actor_critic_args=args)}
    col = RNNCollector(env, agents, T, L, use_graph=False)
    assert not col.use_k14 and col.bank.impl == "torch" and col.bank.H == 12
    col.collect()
    _replay_check(col, agents, N, T, L, 1, stride=4)
    before = torch.cat([p.detach().reshape(-1).clone() for p in agents[0].agent.parameters()])
    r = col.update()[0]
    after = torch.cat([p.detach().reshape(-1) for p in agents[0].agent.parameters()])
    assert np.isfinite([r.loss_policy, r.loss_critic, r.loss_predictor, r.kl_divergence, r.LocLoss]).all() and not torch.equal(before, after)


def test_pfgru_pass_equals_per_step_launches():
    """rs_pfgru_pass (the reset and every step of a no-grad K11 pass from one library call, the launches covering the prefix of episodes still
    running) against rs_pfgru_reset + one rs_pfgru_step call per time step over ALL episodes: bit-identical predictions on every valid
    (step, episode), for ragged episode lengths sorted in descending order."""
    import ctypes as C
    from radiation_ppo_amd import _lib
    from radiation_ppo_amd.pfgru import pack_weights
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    lib = _lib.load()
    ag = RNNAgentPPO(id=0, seed=4)
    g = torch.Generator(device="cuda").manual_seed(9)
    L, E = 17, 203                                                      # not a multiple of the six sets of a workgroup
    lens = torch.sort(torch.randint(1, L + 1, (E,), generator=torch.Generator().manual_seed(2)), descending=True).values
    lens[0] = L
    X = torch.rand(L, E, 11, device="cuda", generator=g).contiguous()
    keys = torch.randint(1, 1 << 40, (E,), device="cuda", generator=g, dtype=torch.int64)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    wts = pack_weights([ag.agent.model])
    episode = torch.ones(E, dtype=torch.int64, device="cuda")
    calls = torch.arange(L, dtype=torch.int64, device="cuda").view(L, 1).expand(L, E).contiguous()
    alpha = float(ag.agent.model.resamp_alpha)

    def buffers():
        return (torch.empty(1, E, 40, 24, device="cuda"), torch.empty(1, E, 40, device="cuda"), torch.zeros(L, E, 2, device="cuda"))
    h, p, loc = buffers()
    alive = [int((lens > t).sum()) for t in range(L)]
    _lib.check(lib.rs_pfgru_pass(wts.data_ptr(), X.data_ptr(), h.data_ptr(), p.data_ptr(), keys.data_ptr(), episode.data_ptr(), calls.data_ptr(), alpha,
                                 loc.data_ptr(), (C.c_int32 * L)(*alive), L, E, st), "rs_pfgru_pass")
    h2, p2, loc2 = buffers()
    _lib.check(lib.rs_pfgru_reset(h2.data_ptr(), p2.data_ptr(), keys.data_ptr(), episode.data_ptr(), calls[0].data_ptr(), None, E, 1, st), "reset")
    for t in range(L):
        _lib.check(lib.rs_pfgru_step(wts.data_ptr(), X[t].data_ptr(), h2.data_ptr(), p2.data_ptr(), keys.data_ptr(), episode.data_ptr(),
                                     calls[t].data_ptr(), None, 1, alpha, loc2[t].data_ptr(), E, 1, st), "step")
    valid = (torch.arange(L).view(L, 1) < lens.view(1, E)).cuda()
    assert torch.equal(loc[valid], loc2[valid])
    assert bool((loc[~valid] == 0).all())                               # launches stop at the episodes still running
    assert lib.rs_pfgru_pass(wts.data_ptr(), X.data_ptr(), h.data_ptr(), p.data_ptr(), keys.data_ptr(), episode.data_ptr(), calls.data_ptr(), alpha,
                             loc.data_ptr(), (C.c_int32 * L)(*alive[::-1]), L, E, st) == 1                    # RS_ERR_INVALID_ARG: not a prefix order


def test_update_scratch_is_reused_across_epochs():
    """rada2c.Scratch: the epoch-sized buffers of the K13 passes are views of persistent blocks with headroom -- a slightly larger request
    does not allocate, a much larger one replaces the block; views of different names never alias."""
    from radiation_ppo_amd.rada2c import Scratch
    sc = Scratch()
    a = sc.get("x", (10, 1000), torch.float32, "cuda")
    ptr = a.data_ptr()
    b = sc.get("x", (10, 1100), torch.float32, "cuda")                  # within the 1/8 headroom
    assert b.data_ptr() == ptr and b.shape == (10, 1100)
    c = sc.get("y", (10, 1000), torch.float32, "cuda")
    assert c.data_ptr() != ptr
    d = sc.get("x", (10, 4000), torch.float32, "cuda")
    assert d.numel() == 40000 and sc.bufs[("x", torch.float32, "cuda")].numel() >= 45000


def test_host_read_returns_the_device_values():
    from radiation_ppo_amd.ppo import host_read
    t = torch.arange(7, dtype=torch.float64, device="cuda") * 0.5
    assert host_read(t) == [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0]
    assert host_read(t * 2) == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]      # the pinned staging buffer is reused
    assert host_read(torch.tensor([1.0, 2.0])) == [1.0, 2.0]            # CPU tensors: plain tolist


@pytest.mark.parametrize("A,obst", [(1, 2), (2, 0)])
def test_glued_lock_step_equals_the_torch_composition(A, obst):
    """rs_collect_pre / _post_step / _post_reset + rs_rnn_policy_step_rows (15 launches per lock-step) against the torch composition of the
    same bookkeeping (~45 launches): after two epochs every buffer, the epoch statistics and everything the collector carries (observation,
    Welford state, returns, step counters, GRU states, particle sets, draw counters) are IDENTICAL, bit for bit, for one and two agents."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.rada2c import RNNAgentPPO, RNNCollector
    N, T, L = 80, 26, 8

    def run(glue):
        torch.manual_seed(4)
        env = RadSearchVec(N, number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True, seed=SEED, env_id_base=32)
        agents = {a: RNNAgentPPO(id=a, steps_per_epoch=T, steps_per_episode=L, seed=3 + a, train_pi_iters=1, train_pfgru_iters=1) for a in range(A)}
        with torch.no_grad():
            for ag in agents.values():
                for p in ag.agent.pi.parameters():
                    p.mul_(2.0)
        col = RNNCollector(env, agents, T, L, use_graph=False)
        assert col.use_glue
        col.use_glue = glue
        out = []
        for ep in range(2):
            st = col.collect()
            out.append({**{k: getattr(col.buf, k).clone() for k in ("obs", "act", "rew", "val", "logp", "last_val", "cut", "adv", "ret", "source_tar")},
                        **{"stat_" + k: v.clone() for k, v in st.items()},
                        "c_obs": col.obs.clone(), "c_ret": col.ep_ret.clone(), "c_steps": col.steps_in_ep.clone(), "c_h": col.h.clone(),
                        "w_count": col.stat.count.clone(), "w_mean": col.stat.mean.clone(), "w_sq": col.stat.sq.clone(), "w_std": col.stat.std.clone(),
                        "pf_h": col.bank.h.clone(), "pf_p": col.bank.p.clone(), "pf_episode": col.bank.episode.clone(), "pf_calls": col.bank.calls.clone(),
                        "begun": col.episodes_begun.clone(), "t": col._t.clone()})
        assert env.error_flags() == 0
        return out
    g, e = run(True), run(False)
    assert int(e[0]["cut"].sum()) > N * A * 2 and float(e[0]["last_val"].abs().sum()) > 0
    for ep in range(2):
        for k in e[ep]:
            assert torch.equal(g[ep][k], e[ep][k]), (ep, k)
