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


def test_heat_maps_without_enforced_walls_match_oracle():
    """enforce_grid_boundaries=False (the env's own default, rad_search_env.py:328): the maps grow to 147 x 147 cells
    (CNNBase.__post_init__, RADTEAM_core.py:1727-1738: the offset covers steps_per_episode steps beyond the search area) and detectors
    do leave the area.  K5 float32-exact against the MapsBuffer oracle at that size, 2 agents."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import HeatMaps, heat_map_geometry
    N, L, A = 12, 120, 2
    env = RadSearchVec(N, number_agents=A, obstruction_count=0, enforce_grid_boundaries=False, seed=SEED)
    hm = HeatMaps(env, steps_per_episode=L, enforce_boundaries=False)
    ra, off, dims = heat_map_geometry(env, L, False)
    assert hm.map_dimensions == dims == (147, 147) and ra == 22.0
    refs = [RadSearchOracle(PhiloxDraws(SEED, n), number_agents=A, obstruction_count=0, enforce_grid_boundaries=False) for n in range(N)]
    bufs = [[MapsOracle(steps_per_episode=L, number_of_agents=A, resolution_accuracy=ra, offset=off) for _ in range(A)] for _ in range(N)]
    assert bufs[0][0].dims == (147, 147)
    obs_t = env.reset()[0]
    cur = [e._ret[0] for e in refs]
    rng = np.random.default_rng(3)
    for t in range(30):
        pred = torch.rand(N, A, 2, generator=torch.Generator().manual_seed(t)).cuda() * 1.2
        hm.update(obs_t, pred)
        actor, critic = (x.cpu().numpy() for x in hm.stacks())
        pc = pred.cpu().numpy()
        for n in range(0, N, 3):
            od = {i: np.array(cur[n][i], dtype=np.float64) for i in range(A)}
            for i in range(A):
                m = bufs[n][i].observation_to_map(od, i, (float(pc[n, i, 0]), float(pc[n, i, 1])))
                assert np.array_equal(actor[n, i], np.stack([m[0], m[1], m[2], m[3], m[4], m[5]])), (t, n, i)
        acts = np.full((N, A), 0, dtype=np.int8)                     # everybody walks left: out of the search area after a few steps
        acts[:, 1] = rng.integers(0, 9, size=N)
        obs_t = env.step(torch.from_numpy(acts).cuda())[0]
        for n, e in enumerate(refs):
            cur[n] = e.step({a: int(acts[n, a]) for a in range(A)})[0]
    assert float(obs_t[:, 0, 1].min()) < 0.0                           # scaled x below the area: only representable on the big map
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


def test_cnn_train_entry_point_multiagent_global_critic():
    """train_PPO(actor_critic_architecture='cnn', global_critic_flag=True) -- the reference's defaults
    (train.py:119-121) -- runs collector + update for 2 agents; stored stacks rebuild exactly; losses finite."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.train import train_PPO
    env = RadSearchVec(16, number_agents=2, obstruction_count=2, enforce_grid_boundaries=True, seed=5)
    sim = train_PPO(env=env, logger_kwargs={}, ppo_kwargs=dict(steps_per_epoch=24, steps_per_episode=8, number_of_agents=2,
                                                               train_pi_iters=3, train_v_iters=3, alpha=0.1),
                    seed=5, number_of_agents=2, actor_critic_architecture="cnn", global_critic_flag=True,
                    steps_per_epoch=24, steps_per_episode=8, total_epochs=2)
    assert sim.agents[0].critic is sim.agents[1].critic          # train.py:228-232
    sim.train()
    col = sim.collector
    # the actor stack rebuilt from (shared maps, cells) equals what the kernel produced at collection time
    col.maps.update(col.obs)
    actor, critic = col.maps.stacks()
    cells = col.maps.field("cell").long()
    pc = col.maps.field("pred_cell").long()
    for a in range(2):
        rebuilt = col.actor_stack_from(critic, cells, pc, a)
        assert torch.equal(rebuilt, actor[:, a])
    rows = sim.loggers[0].rows
    assert len(rows) == 2 and all(np.isfinite(r["loss_policy"]) and np.isfinite(r["loss_critic"]) for r in rows)
    assert np.isnan(sim.loggers[1].rows[0]["loss_critic"])        # only agent 0 updates the global critic (ppo.py:858)


def test_cnn_collector_replays_through_oracles():
    """One epoch of the multi-agent CNN collector, replayed with the env oracle + the MapsBuffer oracle following
    the reference loop (train.py:332-548): select_action round -> env.step -> bootstrap round on timeout/epoch cut
    (the maps see the last observation twice) -> MapsBuffer.reset + env.reset.  Stored shared maps, cells, rewards
    and cuts must be reproduced exactly."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
    N, A, T, L, obst = 24, 2, 30, 9, 3
    torch.manual_seed(2)
    env = RadSearchVec(N, number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True, seed=SEED)
    gc = CNNCritic().cuda()
    gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
    agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=2, train_v_iters=2) for i in range(A)}
    col = CNNCollector(env, agents, T, L, global_critic_flag=True)
    col.collect()
    buf = col.buf
    act, rew, cut = (x.cpu().numpy() for x in (buf.act, buf.rew, buf.cut))
    shared, cells = col.shared.cpu().numpy(), col.cells.cpu().numpy()
    for n in range(0, N, 4):
        e = RadSearchOracle(PhiloxDraws(SEED, n), number_agents=A, obstruction_count=obst, enforce_grid_boundaries=True)
        mb = MapsOracle(steps_per_episode=L, number_of_agents=A)
        o = e._ret[0]
        steps = 0
        for t in range(T):
            od = {i: np.array(o[i], dtype=np.float64) for i in range(A)}
            m = mb.observation_to_map(od, 0, (0.0, 0.0))
            exp = np.stack([mb.combined, mb.readings_map, mb.visits, mb.obstacles])
            assert np.array_equal(shared[t, n], exp), (n, t)
            for i in range(A):
                assert cells[t, n, i] == mb.last_coords[i][0] * 27 + mb.last_coords[i][1]
            ro, rr, rd, _ = e.step({i: int(act[t, n, i]) for i in range(A)})
            assert rew[t, n, 0] == np.float32(rr["team_reward"]), (n, t)
            steps += 1
            o = ro
            over = any(rd.values()) or steps == L
            expect_cut = over or t == T - 1
            assert bool(cut[t, n, 0]) == expect_cut, (n, t)
            if expect_cut:
                if steps == L or t == T - 1:
                    mb.observation_to_map({i: np.array(o[i], dtype=np.float64) for i in range(A)}, 0, (0.0, 0.0))
                if t == T - 1:
                    e.epoch_end = True
                mb.reset()
                o = e.reset()[0]
                steps = 0
    res = col.update()
    assert np.isfinite(res[0].loss_policy) and np.isfinite(res[0].loss_critic)


def test_predictor_feeds_heat_map_channel_zero():
    """Row f1 in the collector: every owner's PFGRU prediction (forward-only, from the episode's h0, as the reference's CNN
    harness runs it) becomes the one-hot of actor channel 0.  A second bank with the same seed and weights, fed the stored
    observations with the (episode, step) counters rebuilt from the cuts, must reproduce every stored prediction cell
    int(pred * resolution_accuracy) (MapsBuffer._update_prediction_map, RADTEAM_core.py:747-766); the actor input built by the
    trunk kernel for the stored cells carries that one-hot."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.pfgru import PredictorBank
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
    N, A, T, L = 32, 2, 20, 7
    torch.manual_seed(8)
    env = RadSearchVec(N, number_agents=A, obstruction_count=1, enforce_grid_boundaries=True, seed=SEED, env_id_base=64)
    gc = CNNCritic().cuda()
    agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=torch.optim.Adam(gc.parameters(), lr=1e-3)) for i in range(A)}
    col = CNNCollector(env, agents, T, L, global_critic_flag=True)
    assert col.predictor is not None and agents[1].model is col.predictor.cells[1]
    col.collect()
    twin = PredictorBank(N, A, seed=SEED, env_id_base=64, device="cuda")
    for a in range(A):
        twin.load_state_dict(a, col.predictor.state_dict(a))
    ra = col.maps.resolution_accuracy
    twin.reset()
    hits = 0
    for t in range(T):
        pred = twin.predict(col.buf.obs[t])
        px, py = (pred[..., 0].double() * ra).long(), (pred[..., 1].double() * ra).long()
        ok = (px >= 0) & (px < 27) & (py >= 0) & (py < 27)
        want = px * 27 + py
        got = col.pcells[t]
        assert torch.equal(got[ok], want[ok]), t
        hits += int(ok.sum())
        # the collector's bootstrap round (timeouts / epoch end) makes one more prediction for those envs, then they restart
        cut = col.buf.cut[t, :, 0].bool()
        boot = cut & (col.buf.last_val[t, :, 0] != 0)
        twin.calls += boot.long()
        twin.reset(mask=cut)
    assert hits > N * A * T // 2
    # channel 0 of the actor input = one-hot of the stored prediction cell
    stack = col.actor_stack_from(col.shared[5], col.cells[5], col.pcells[5], 1)
    pc = col.pcells[5][:, 1]
    for n in range(N):
        ch0 = stack[n, 0].reshape(-1)
        assert ch0.sum() == (1.0 if pc[n] >= 0 else 0.0) and (pc[n] < 0 or ch0[pc[n]] == 1.0)


def test_cnn_collector_follows_the_pinned_train_loop():
    """The product collector against oracle/train_loop_oracle.train_loop_trace -- the restatement that
    tests/test_train_loop_golden.py pins to the reference's own train() -- env by env: the trace is driven with
    the actions / values the device stored and must reproduce what the device stored for every store() call
    (observation, team reward, value, terminal flag), every GAE bootstrap value and the episode statistics."""
    from oracle.train_loop_oracle import train_loop_trace
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
    N, A, T, L = 16, 2, 36, 10
    torch.manual_seed(5)
    env = RadSearchVec(N, number_agents=A, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED)
    gc = CNNCritic().cuda()
    gco = torch.optim.Adam(gc.parameters(), lr=1e-3)
    agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=2, train_v_iters=2) for i in range(A)}
    col = CNNCollector(env, agents, T, L, global_critic_flag=True)
    stats = col.collect()
    buf = col.buf
    obs, act, rew, val, logp, cut, lastv = (x.cpu().numpy() for x in (buf.obs, buf.act, buf.rew, buf.val, buf.logp, buf.cut, buf.last_val))
    ep_ret_sum, ep_len_sum, ep_cnt = 0.0, 0.0, 0
    ep_rets = {i: [] for i in range(A)}
    done_cnt, oob_cnt = np.zeros(A), np.zeros(A)
    for n in range(N):
        ref = RadSearchOracle(PhiloxDraws(SEED, n), number_agents=A, obstruction_count=0, enforce_grid_boundaries=True)
        st = {"t": 0, "after_step": False}

        class Env:                                   # train() begins with env.reset(): the constructor's reset stands for it
            first = True
            src = property(lambda s: ref.src)

            def reset(s):
                st["after_step"] = False
                if Env.first:
                    Env.first = False
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
            if st["after_step"] and cut[st["t"] - 1, n, 0]:      # the bootstrap round (train.py:476-480)
                return 0, float(lastv[st["t"] - 1, n, i]), 0.0
            t = st["t"]
            return int(act[t, n, i]), float(val[t, n, i]), float(logp[t, n, i])

        ev, _ = train_loop_trace(Env(), agent_step, A, True, T, L, 1)
        stores = [e for e in ev if e[0] == "store"]
        assert len(stores) == T * A
        for k, e in enumerate(stores):
            t, i = divmod(k, A)
            assert np.array_equal(obs[t, n, i], np.asarray(e[2], dtype=np.float64).astype(np.float32)), (n, t, i)
            assert rew[t, n, i] == np.float32(e[3]) and bool(cut[t, n, i]) == e[8], (n, t, i)
        # zero bootstrap exactly where the trace says the trajectory ended on a terminal
        gae = [e for e in ev if e[0] == "gae"]
        cuts_t = [t for t in range(T) if cut[t, n, 0]]
        assert len(gae) == len(cuts_t) * A
        for k, e in enumerate(gae):
            assert float(lastv[cuts_t[k // A], n, e[1]]) == e[2]
        ep_ret_sum += sum(e[3] for e in ev if e[0] == "log" and e[1] == 0 and e[2] == "EpRet")
        ep_len_sum += sum(e[3] for e in ev if e[0] == "log" and e[1] == 0 and e[2] == "EpLen")
        ep_cnt += sum(1 for e in ev if e[0] == "ep_len" and e[1] == 0)
        for i in range(A):                           # what train() stores on logger i (train.py:494-526)
            ep_rets[i] += [e[3] for e in ev if e[0] == "log" and e[1] == i and e[2] == "EpRet"]
            done_cnt[i] += sum(e[3] for e in ev if e[0] == "log" and e[1] == i and e[2] == "DoneCount")
            oob_cnt[i] += sum(e[3] for e in ev if e[0] == "log" and e[1] == i and e[2] == "OutOfBound")
    assert int(stats["EpCount"].item()) == ep_cnt and float(stats["EpLenSum"].item()) == ep_len_sum
    assert abs(float(stats["EpRetSum"][0].item()) - ep_ret_sum) < 1e-4
    for i in range(A):                               # the per-agent logger columns: MeanEpRet StdEpRet MaxEpRet MinEpRet DoneCount OutOfBound
        r = np.array(ep_rets[i], dtype=np.float64)
        assert abs(float(stats["EpRetSum"][i]) - r.sum()) < 1e-4 and abs(float(stats["EpRetSqSum"][i]) - (r * r).sum()) < 1e-3
        assert abs(float(stats["EpRetMax"][i]) - r.max()) < 1e-5 and abs(float(stats["EpRetMin"][i]) - r.min()) < 1e-5
        assert float(stats["DoneCount"][i]) == done_cnt[i] and float(stats["OutOfBound"][i]) == oob_cnt[i]


def test_cnn_update_losses_and_gradients_match_reference(golden_dir):
    """CNNAgentPPO.update_agent (one actor iteration, one critic iteration, lr 0) against the reference's own
    compute_batched_losses_pi / compute_batched_losses_critic (algos/multiagent/ppo.py:903-1045) on the reference CNNs
    (tests/golden/cnn_loss.npz): policy loss, approx-KL, entropy, clip fraction, critic MSE and every parameter
    gradient.  Tolerance fp32 rtol 2e-4 / atol 1e-6 (MIOpen convolutions vs the reference's CPU kernels)."""
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO
    g = dict(np.load(os.path.join(golden_dir, "cnn_loss.npz")).items())
    ag = CNNAgentPPO(id=0, train_pi_iters=1, train_v_iters=1, actor_learning_rate=0.0, critic_learning_rate=0.0, target_kl=10.0)
    ag.pi.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("a_actor.")})
    ag.critic.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("c_critic.")})
    xa, xc = torch.from_numpy(g["xa"][:, 0]).cuda(), torch.from_numpy(g["xc"][:, 0]).cuda()
    n = xa.shape[0]
    dev = lambda k, dt=None: torch.from_numpy(g[k]).cuda() if dt is None else torch.from_numpy(g[k]).cuda().to(dt)
    w = torch.full((n,), 1.0 / n, device="cuda")
    r = ag.update_agent(lambda lo, hi: xa[lo:hi], lambda lo, hi: xc[lo:hi], dev("act", torch.int64), dev("adv"), dev("ret"),
                        dev("logp_old"), w, update_critic=True)
    assert r.stop_iteration == 1
    assert abs(r.loss_policy - float(g["pi_loss"])) < 2e-6 and abs(r.kl_divergence - float(g["kl"])) < 2e-6
    assert abs(r.Entropy - float(g["entropy"])) < 2e-6 and abs(r.ClipFrac - float(g["clip_fraction"])) < 1e-7
    assert abs(r.loss_critic - float(g["critic_loss"])) < 2e-6
    for k, p in ag.pi.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), g["ga_" + k], rtol=2e-4, atol=1e-6), k
    for k, p in ag.critic.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), g["gc_" + k], rtol=2e-4, atol=1e-6), k


@pytest.mark.parametrize("A", [1, 3])
def test_cnn_train_entry_point_individual_critics(A):
    """global_critic_flag=False: every agent owns and updates its own critic (train.py:234-246); A = 1 is the
    single-agent harness of algos/test_cnn."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.train import train_PPO
    env = RadSearchVec(16, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=6)
    sim = train_PPO(env=env, logger_kwargs={}, ppo_kwargs=dict(steps_per_epoch=24, steps_per_episode=8, number_of_agents=A,
                                                               train_pi_iters=2, train_v_iters=2),
                    seed=6, number_of_agents=A, actor_critic_architecture="cnn", global_critic_flag=False,
                    steps_per_epoch=24, steps_per_episode=8, total_epochs=2)
    critics = [sim.agents[i].critic for i in range(A)]
    assert len({id(c) for c in critics}) == A
    before = [[p.detach().clone() for p in c.parameters()] for c in critics]
    sim.train()
    for i in range(A):
        rows = sim.loggers[i].rows
        assert len(rows) == 2 and np.isfinite(rows[1]["loss_policy"]) and np.isfinite(rows[1]["loss_critic"])
        assert any(not torch.equal(a, b) for a, b in zip(before[i], critics[i].parameters()))     # each critic was trained


def test_cnn_collector_graph_replay_equals_eager_steps():
    """The captured lock-step (one HIP graph replayed T - 1 times per epoch) writes exactly what the eager loop writes: every
    buffer, two epochs (the second one replays the graph captured in the first)."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
    N, A, T, L = 32, 2, 24, 7

    def run(use_graph):
        torch.manual_seed(8)
        env = RadSearchVec(N, number_agents=A, obstruction_count=2, enforce_grid_boundaries=True, seed=SEED, env_id_base=32)
        gc = CNNCritic().cuda()
        agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=torch.optim.Adam(gc.parameters(), lr=1e-3)) for i in range(A)}
        col = CNNCollector(env, agents, T, L, global_critic_flag=True, use_graph=use_graph)
        out = []
        for _ in range(2):
            st = col.collect()
            out.append({**{k: getattr(col.buf, k).clone() for k in ("obs", "act", "rew", "val", "logp", "last_val", "cut", "adv", "ret")},
                        "shared": col.shared.clone(), "cells": col.cells.clone(), "pcells": col.pcells.clone(),
                        "complete_len": col.complete_len.clone(), **{"stat_" + k: v.clone() for k, v in st.items()}})
        assert (col._graph is not None) == use_graph
        return out
    g, e = run(True), run(False)
    for ep in range(2):
        for k in e[ep]:
            assert torch.equal(g[ep][k], e[ep][k]), (ep, k)


def test_actor_loss_kernel_equals_the_torch_composition():
    """rs_actor_loss (ppo_cnn.ActorLoss: compute_loss_pi behind the logits, forward + derivative in one launch) against the torch
    composition it replaces: loss, kl / entropy / clip-fraction sums, d loss / d logits; both clip sides occur."""
    import numpy as np
    from radiation_ppo_amd.ppo_cnn import ActorLoss
    g = torch.Generator().manual_seed(8)
    S = 70001
    logits = (torch.randn(S, 8, generator=g) * 1.5).cuda().requires_grad_(True)
    act = torch.randint(0, 8, (S,), generator=g).cuda()
    adv = torch.randn(S, generator=g).cuda()
    logp_old = (float(np.log(1 / 8)) + 0.5 * torch.randn(S, generator=g)).cuda()
    w = (torch.rand(S, generator=g) / S).cuda()
    clip = 0.2
    loss_k, st = ActorLoss.apply(logits, act, adv, logp_old, w, clip)
    loss_k.backward()
    gk = logits.grad.clone()
    logits.grad = None
    lp_all = torch.log_softmax(logits, dim=-1)
    lp = lp_all.gather(-1, act.unsqueeze(-1)).squeeze(-1)
    ratio = torch.exp(lp - logp_old)
    loss_t = -(w * torch.min(ratio * adv, torch.clamp(ratio, 1 - clip, 1 + clip) * adv)).sum()
    loss_t.backward()
    ent = -(lp_all.exp() * lp_all).sum(-1)
    cf = ((ratio > 1 + clip) | (ratio < 1 - clip)).float()
    want = torch.stack([(w * (logp_old - lp)).sum(), (w * ent).sum(), (w * cf).sum(), loss_t.detach()]).double()
    assert 0.1 < float(want[2]) < 0.9
    assert torch.allclose(st, want, rtol=2e-5, atol=1e-8), (st, want)
    assert torch.allclose(gk, logits.grad, rtol=1e-4, atol=1e-6 * float(logits.grad.abs().max())), float((gk - logits.grad).abs().max())


def test_cnn_update_draws_the_reference_s_minibatch_sample():
    """`minibatch` on the 'cnn' path (ppo.py:754-766, :825-873): every actor iteration works on int(ep_len / minibatch) drawn indexes per
    env, the critic loop on the LAST draw.  With learning rates 0 the logged actor loss of an update with minibatch = 3 must equal
    the weighted mean over exactly the drawn subset of the last iteration, computed here from the dense library-convolution path;
    minibatch = 1 must reproduce the update without the option bit for bit."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.ppo import normalize_advantages
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector, minibatch_weights
    N, A, T, L = 16, 2, 30, 9

    def run(minibatch):
        torch.manual_seed(2)
        env = RadSearchVec(N, number_agents=A, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED)
        gc = CNNCritic().cuda()
        gco = torch.optim.Adam(gc.parameters(), lr=0.0)
        kw = {} if minibatch is None else dict(minibatch=minibatch)
        agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=gco, train_pi_iters=3, train_v_iters=2, actor_learning_rate=0.0,
                                 target_kl=1e9, seed=7, **kw) for i in range(A)}
        col = CNNCollector(env, agents, T, L, global_critic_flag=True)
        col.collect()
        return col, agents, col.update()

    col0, _, r0 = run(None)
    col1, _, r1 = run(1)
    assert r0[0].loss_policy == r1[0].loss_policy and r0[0].loss_critic == r1[0].loss_critic and r0[1].kl_divergence == r1[1].kl_divergence
    col, agents, r3 = run(3)
    assert col.epoch == 1
    buf = col.buf
    ids = torch.arange(N, device="cuda", dtype=torch.int64)
    for a in range(A):
        base = (7 * 4294967296 + ids) * 1048583 + 0 * 4096 + a * 64
        w = minibatch_weights(col.complete_len, T, 3, base + 2, N)                     # the draw of the last (third) iteration
        assert torch.equal((w > 0).sum(0), col.complete_len // 3)
        stack = col.actor_stack_from(col.shared.view(T * N, 4, 27, 27), col.cells.view(T * N, A), col.pcells.view(T * N, A), a)
        with torch.no_grad():
            logp_all = torch.log_softmax(agents[a].pi.logits(stack), dim=-1)
        logp = logp_all.gather(-1, buf.act[:, :, a].reshape(-1, 1)).squeeze(-1)
        adv = normalize_advantages(buf.adv[:, :, a]).reshape(-1)
        ratio = torch.exp(logp - buf.logp[:, :, a].reshape(-1))
        want = -(w.reshape(-1) * torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv)).sum()
        assert abs(r3[a].loss_policy - float(want)) < 1e-5 * max(1.0, abs(float(want))), (a, r3[a].loss_policy, float(want))
    assert r3[0].loss_policy != r0[0].loss_policy
    with pytest.raises(ValueError, match="minibatch"):
        c2, ag2, _ = run(1)
        for g in ag2.values():
            g.minibatch = 10_000
        c2.collect(); c2.update()


def test_heat_maps_bin_the_noisy_coordinates_when_coord_noise_is_on():
    """coord_noise=True (rad_search_env.py:365, :569-580): the reference's MapsBuffer bins the OBSERVED coordinates
    (int(observation[1] * resolution_accuracy), RADTEAM_core.py:705-711), which carry N(0, 5 cm), not the exact position.  K5 fed by the
    noisy env must equal the MapsBuffer oracle fed the same observation rows, float32-exact, and some step must land in another cell than
    the exact position would (otherwise the test would not see the difference)."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import HeatMaps
    N, L, A = 48, 14, 2
    env = RadSearchVec(N, number_agents=A, obstruction_count=0, enforce_grid_boundaries=True, seed=SEED, coord_noise=True)
    hm = HeatMaps(env, steps_per_episode=L)
    bufs = [[MapsOracle(steps_per_episode=L, number_of_agents=A) for _ in range(A)] for _ in range(N)]
    obs_t = env.reset()[0]
    rng = np.random.default_rng(5)
    moved = 0
    for t in range(12):
        pred = torch.rand(N, A, 2, generator=torch.Generator().manual_seed(t)).cuda()
        hm.update(obs_t, pred)
        actor = hm.stacks()[0].cpu().numpy()
        o, pc = obs_t.double().cpu().numpy(), pred.cpu().numpy()
        x, y = env.state("x").cpu().numpy(), env.state("y").cpu().numpy()
        for n in range(N):
            od = {i: o[n, i] for i in range(A)}
            for i in range(A):
                m = bufs[n][i].observation_to_map(od, i, (float(pc[n, i, 0]), float(pc[n, i, 1])))
                assert np.array_equal(actor[n, i], np.stack([m[0], m[1], m[2], m[3], m[4], m[5]])), (t, n, i)
                exact = (int(x[i, n] / 2200.0 * 22.0), int(y[i, n] / 2200.0 * 22.0))
                moved += int((int(o[n, i, 1] * 22.0), int(o[n, i, 2] * 22.0)) != exact)
        acts = rng.integers(0, 8, size=(N, A)).astype(np.int8)
        obs_t = env.step(torch.from_numpy(acts).cuda())[0]
    assert moved > 0 and int(hm.field("err").max().item()) == 0


@pytest.mark.parametrize("team", [True, False])
def test_cnn_glued_lock_step_equals_the_torch_composition(team):
    """The RAD-TEAM lock-step with its bookkeeping in rs_collect_post_step / _post_reset against the element-wise torch composition: every
    buffer, the stored maps and cells, complete_len, the epoch statistics and the carried state (observation, returns, step counters,
    predictor draw counters) identical after two epochs -- team reward (global critic) and individual rewards."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
    N, A, T, L = 32, 3, 24, 7

    def run(glue):
        torch.manual_seed(8)
        env = RadSearchVec(N, number_agents=A, obstruction_count=2, enforce_grid_boundaries=True, seed=SEED, env_id_base=32)
        gc = CNNCritic().cuda() if team else None
        agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=torch.optim.Adam(gc.parameters(), lr=1e-3) if team else None)
                  for i in range(A)}
        col = CNNCollector(env, agents, T, L, global_critic_flag=team, use_graph=False)
        assert col.use_glue
        col.glue = glue
        col.heads = False             # the fused select_action round is compared on its own below (float32 summation order differs)
        out = []
        for _ in range(2):
            st = col.collect()
            out.append({**{k: getattr(col.buf, k).clone() for k in ("obs", "act", "rew", "val", "logp", "last_val", "cut", "adv", "ret")},
                        "shared": col.shared.clone(), "cells": col.cells.clone(), "pcells": col.pcells.clone(),
                        "complete_len": col.complete_len.clone(), **{"stat_" + k: v.clone() for k, v in st.items()},
                        "c_obs": col.obs.clone(), "c_ret": col.ep_ret.clone(), "c_steps": col.steps_in_ep.clone(), "t": col._t.clone(),
                        "pf_episode": col.predictor.episode.clone(), "pf_calls": col.predictor.calls.clone()})
        assert env.error_flags() == 0
        return out
    g, e = run(True), run(False)
    assert int(e[0]["cut"].sum()) > N * A
    for ep in range(2):
        for k in e[ep]:
            assert torch.equal(g[ep][k], e[ep][k]), (ep, k)


@pytest.mark.parametrize("team", [True, False])
def test_cnn_fused_select_action_round_equals_the_module_path(team):
    """rs_cnn_trunk_infer + BLAS Linear(2704, 32) + rs_cnn_head + rs_store_rows (the collector's default on 27 x 27 maps) against the
    module path (ConvTrunk + nn.Linear layers + the torch sampling tail): on the first lock-step, where both collectors see the same
    state, log-probabilities and values agree to float32 summation order, the drawn actions are the same except where the uniform falls
    within that noise of a CDF value, and after a whole epoch the fused collector's buffers hold exactly what its own rows say (actions
    in range, log-probability = log-softmax of the module's logits at the stored action, value = the module's value)."""
    from radiation_ppo_amd.envs import RadSearchVec
    from radiation_ppo_amd.maps import CNNCritic
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, CNNCollector
    N, A, T, L = 64, 3, 12, 7

    def run(heads):
        torch.manual_seed(8)
        env = RadSearchVec(N, number_agents=A, obstruction_count=2, enforce_grid_boundaries=True, seed=SEED, env_id_base=32)
        gc = CNNCritic().cuda() if team else None
        agents = {i: CNNAgentPPO(id=i, GlobalCritic=gc, GlobalCriticOptimizer=torch.optim.Adam(gc.parameters(), lr=1e-3) if team else None)
                  for i in range(A)}
        with torch.no_grad():
            for ag in agents.values():
                for p in list(ag.pi.actor[6:].parameters()):
                    p.mul_(3.0)                                   # visibly non-uniform policies
        col = CNNCollector(env, agents, T, L, global_critic_flag=team, use_graph=False)
        col.heads = heads
        assert col.use_heads == heads
        col.collect()
        return col, agents
    cf, af = run(True)
    cm, _ = run(False)
    same = cf.buf.act[0] == cm.buf.act[0]
    assert float(same.float().mean()) > 0.99
    assert torch.allclose(cf.buf.logp[0][same], cm.buf.logp[0][same], rtol=1e-4, atol=2e-5)
    assert torch.allclose(cf.buf.val[0], cm.buf.val[0], rtol=1e-4, atol=2e-5)
    assert torch.equal(cf.buf.obs[0], cm.buf.obs[0]) and torch.equal(cf.shared[0], cm.shared[0])
    # the fused collector's own epoch: every stored row against the modules evaluated on the stored maps
    assert int(cf.buf.act.min()) >= 0 and int(cf.buf.act.max()) <= 7
    for t in (0, 5, T - 1):
        for a, ag in af.items():
            logits = ag.pi.logits_from_maps(cf.shared[t], cf.cells[t], cf.pcells[t], a)
            lp = torch.log_softmax(logits, dim=-1).gather(-1, cf.buf.act[t, :, a].unsqueeze(-1)).squeeze(-1)
            assert torch.allclose(cf.buf.logp[t, :, a], lp, rtol=1e-4, atol=2e-5), (t, a)
            v = ag.critic.value_from_maps(cf.shared[t])
            assert torch.allclose(cf.buf.val[t, :, a], v, rtol=1e-4, atol=2e-5), (t, a)
    boot = cf.buf.last_val != 0
    assert int(boot.sum()) > 0 and bool((cf.buf.cut[boot.any(dim=2)] == 1).all())
