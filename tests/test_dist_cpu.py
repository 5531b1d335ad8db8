"""N>1 path on CPU: two gloo ranks, each owning half of the envs' samples, must produce exactly the
update a single process computes on the whole batch (the reference's mpi_avg_grads / mpi_avg /
mpi_statistics_scalar call sites: ppo.py:445,1250,1256; mpi_pytorch.py:26-49; mpi_tools.py:71-95)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(seed=0, n=512):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, 11, generator=g)
    act = torch.randint(0, 8, (n,), generator=g)
    adv = torch.randn(n, generator=g)
    ret = torch.randn(n, generator=g)
    logp_old = torch.full((n,), float(np.log(1 / 8)))
    w = torch.rand(n, generator=g)
    w = w / w.sum()
    return X, act, adv, ret, logp_old, w


def _agent():
    from radiation_ppo_amd.ppo import VecAgentPPO
    torch.manual_seed(123)
    return VecAgentPPO(id=0, alpha=0.1, train_pi_iters=6, target_kl=0.07, device="cpu")


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from radiation_ppo_amd.ppo import Collectives, normalize_advantages
    ag = _agent()
    if rank == 1:                       # de-synchronise on purpose: sync_params must repair it
        with torch.no_grad():
            for p in ag.agent.parameters():
                p.add_(0.5)
    ag.sync_params()
    X, act, adv, ret, logp_old, w = _batch()
    n = X.shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    adv_n = normalize_advantages(adv[sl])
    c0 = Collectives.count
    res = ag.update_agent(X[sl], act[sl], adv_n, ret[sl], logp_old[sl], w[sl])
    flat = torch.cat([p.data.view(-1) for p in ag.agent.parameters()])
    out.put((rank, flat.numpy(), adv_n.numpy(), res.stop_iteration, res.kl_divergence, Collectives.count - c0))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_update_equals_single_process():
    from radiation_ppo_amd.ppo import normalize_advantages
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch
    ag = _agent()
    X, act, adv, ret, logp_old, w = _batch()
    adv_n = normalize_advantages(adv)
    r1 = ag.update_agent(X, act, adv_n, ret, logp_old, w)
    flat = torch.cat([p.data.view(-1) for p in ag.agent.parameters()]).numpy()
    assert np.array_equal(res[0][1], res[1][1]), "ranks diverged"
    assert np.allclose(res[0][1], flat, rtol=1e-5, atol=1e-6)
    assert np.allclose(np.concatenate([res[0][2], res[1][2]]), adv_n.numpy(), rtol=1e-5, atol=1e-6)
    assert res[0][3] == res[1][3] == r1.stop_iteration
    assert abs(res[0][4] - r1.kl_divergence) < 1e-6
    assert res[0][5] == res[1][5] == r1.stop_iteration     # ONE collective per policy iteration: KL + statistics ride in the gradient bucket


# ---------------------------------------------------------------------------------------- RAD-A2C ('rnn') under data parallelism
def _rnn_columns(seed=5, T=30, N=8):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand(T, N, 11, generator=g)
    act = torch.randint(0, 8, (T, N), generator=g)
    adv, ret = torch.randn(T, N, generator=g), torch.randn(T, N, generator=g)
    logp = torch.full((T, N), float(np.log(1 / 8)))
    src = torch.rand(T, N, 2, generator=g) * 2000 + 200
    cut = (torch.rand(T, N, generator=g) < 0.12).to(torch.uint8)
    cut[-1] = 1
    return obs, act, adv, ret, logp, src, cut


def _rnn_update(lo, hi, n_total, base):
    from radiation_ppo_amd.rada2c import RNNAgentPPO, pack_episodes
    torch.manual_seed(77)
    ag = RNNAgentPPO(id=0, device="cpu", train_pi_iters=3, train_pfgru_iters=2, alpha=0.1, seed=1)
    cols = [c[:, lo:hi].contiguous() for c in _rnn_columns()]
    B = pack_episodes(*cols, n_total=n_total, env_id_base=base, seed=9, epoch=0)
    return ag, ag.update_agent(B)


def _rnn_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from radiation_ppo_amd.ppo import Collectives
    ag, res = _rnn_update(rank * 4, rank * 4 + 4, 8, rank * 4)
    flat = torch.cat([p.data.view(-1) for p in ag.agent.parameters()])
    out.put((rank, flat.numpy(), res.stop_iteration, res.kl_divergence, res.loss_predictor, res.loss_policy, Collectives.count))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_recurrent_update_equals_single_process():
    """RNNAgentPPO.update_agent (PFGRU update + BPTT policy update) on two gloo ranks holding four env columns each == one
    process holding all eight: episode weights carry 1 / global env count, draws are keyed by global env id, gradients and
    statistics are summed over ranks (ppo.py:1139-1141, :1250-1256)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rnn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ag, one = _rnn_update(0, 8, 8, 0)
    want = torch.cat([p.data.view(-1) for p in ag.agent.parameters()]).numpy()
    assert np.array_equal(res[0][1], res[1][1])                       # the ranks stay in lock-step
    # Adam's step is lr * m / (sqrt(v) + 1e-8): where a gradient is ~1e-8 the different summation order of the two ranks is
    # amplified up to the step size, so all elements are held to the step size and 99.5 % to float32 accuracy
    diff = np.abs(res[0][1] - want)
    assert diff.max() <= 1.5e-2 and np.mean(diff <= 2e-6 + 2e-4 * np.abs(want)) >= 0.995, (diff.max(), np.mean(diff <= 2e-6 + 2e-4 * np.abs(want)))
    assert res[0][2] == one.stop_iteration
    assert np.isclose(res[0][3], one.kl_divergence, rtol=1e-3, atol=1e-6) and np.isclose(res[0][4], one.loss_predictor, rtol=1e-4)
    assert np.isclose(res[0][5], one.loss_policy, rtol=1e-3, atol=1e-6)
    # one collective per PFGRU iteration (2) and per policy iteration (mpi_avg_grads + mpi_avg in one bucket, ppo.py:1139-1141, :1250-1256)
    assert res[0][6] == res[1][6] == 2 + one.stop_iteration, (res[0][6], one.stop_iteration)
