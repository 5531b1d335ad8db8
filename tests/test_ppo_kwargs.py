"""The agents' constructor surface (`ppo_kwargs` of algos/multiagent/main.py:574-596): every key the reference passes
is accepted by all three agent classes, an unknown key raises instead of vanishing, and `minibatch` (ppo.py:580, :754-766) does what the
reference does with it -- nothing on the update_rada2c paths (:1159-1160 assign it, nothing reads it), a per-iteration index sample on
the 'cnn' path."""
import pytest
import torch

from radiation_ppo_amd.ppo import REFERENCE_PPO_KWARGS, VecAgentPPO
from radiation_ppo_amd.ppo_cnn import CNNAgentPPO, minibatch_weights
from radiation_ppo_amd.rada2c import BpArgs, RNNAgentPPO


def _reference_kwargs(arch):
    """main.py:574-596 with its defaults (GlobalCriticOptimizer None as main.py passes it; train.py:191-206 fills it for the cnn)."""
    return dict(observation_space=11, bp_args=BpArgs(area_scale=2200.0), steps_per_epoch=480, steps_per_episode=120, number_of_agents=1,
                env_height=2200.0, actor_critic_args={} if arch != "rnn" else dict(hidden_sizes_pol=((32,),), hidden_sizes_val=((32,),)),
                actor_critic_architecture=arch, minibatch=1, train_pi_iters=40, train_v_iters=40, train_pfgru_iters=15,
                actor_learning_rate=3e-4, critic_learning_rate=1e-3, pfgru_learning_rate=5e-3, gamma=0.99, alpha=0.1, clip_ratio=0.2,
                target_kl=0.07, lam=0.9, GlobalCriticOptimizer=None)


@pytest.mark.parametrize("cls,arch", [(VecAgentPPO, "ff"), (RNNAgentPPO, "rnn"), (CNNAgentPPO, "cnn")])
def test_every_reference_key_is_accepted_and_unknown_keys_raise(cls, arch):
    kw = _reference_kwargs(arch)
    assert set(kw) == set(REFERENCE_PPO_KWARGS)
    if cls is CNNAgentPPO:
        kw.pop("actor_critic_architecture")                    # train_PPO consumes it for the cnn (train.py of this build)
    cls(id=0, device="cpu", **kw)
    with pytest.raises(TypeError, match="train_pi_iter"):
        cls(id=0, device="cpu", train_pi_iter=3)               # a misspelt option
    for bad in (0, -2, 1.5, True):
        with pytest.raises(ValueError, match="minibatch"):
            cls(id=0, device="cpu", minibatch=bad)


def test_minibatch_changes_nothing_on_the_update_rada2c_form():
    torch.manual_seed(0)
    S = 200
    X, act = torch.randn(S, 11), torch.randint(0, 8, (S,))
    adv, ret, logp = torch.randn(S), torch.randn(S), -torch.rand(S) - 1.5
    w = torch.full((S,), 1.0 / S)
    out = []
    for m in (1, 4):
        torch.manual_seed(1)
        ag = VecAgentPPO(id=0, device="cpu", train_pi_iters=3, minibatch=m)
        ag.update_agent(X, act, adv, ret, logp, w)
        out.append(torch.cat([p.detach().reshape(-1) for p in ag.agent.parameters()]))
    assert torch.equal(out[0], out[1])


def test_minibatch_weights_draw_the_reference_s_sample():
    """np.random.choice(arange(ep_len), size=int(ep_len / minibatch), replace=False) per env (ppo.py:759-764)."""
    T, n_total = 480, 16
    cl = torch.tensor([480, 361, 400, 7, 2, 123, 479, 360])
    key = torch.arange(8, dtype=torch.int64) * 977 + 5
    for m in (1, 2, 3, 7):
        w = minibatch_weights(cl, T, m, key, n_total)
        k = cl // m
        sel = w > 0
        assert torch.equal(sel.sum(0), k)                                              # exactly int(ep_len / m) indexes, no repeats
        tt = torch.arange(T).view(T, 1)
        assert not (sel & (tt >= cl.view(1, -1))).any()                                # all below ep_len
        assert torch.allclose(w.sum(0), (k > 0).float() / n_total, atol=1e-7)          # the mean over the drawn indexes, then over ranks
    assert torch.equal(minibatch_weights(cl, T, 1, key, n_total) > 0, torch.arange(T).view(T, 1) < cl.view(1, -1))
    # a fresh draw per key (= per iteration / epoch / agent), the same draw for the same key wherever the env sits in the shard
    a, b = minibatch_weights(cl, T, 2, key, n_total), minibatch_weights(cl, T, 2, key + 1, n_total)
    assert not torch.equal(a, b)
    perm = torch.tensor([3, 0, 7, 1, 2, 6, 5, 4])
    assert torch.equal(minibatch_weights(cl[perm], T, 2, key[perm], n_total), a[:, perm])
    # uniform over the indexes: every index of a 60-step env is drawn in about half of 4000 draws
    cl1 = torch.full((4000,), 60)
    hits = (minibatch_weights(cl1, 64, 2, torch.arange(4000, dtype=torch.int64) * 31 + 11, 1) > 0).float().sum(1)[:60]
    assert (hits - 2000).abs().max() < 5 * (4000 * 0.25) ** 0.5
