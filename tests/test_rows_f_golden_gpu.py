"""Rows f1 / f2 on the GPU, DIRECTLY against the reference's recorded runs: the HIP kernels K11 - K15 are fed the inputs and the
random draws of tests/golden/pfgru.npz and tests/golden/rada2c_core.npz (written by the reference's own PFGRUCell,
RNNModelActorCritic.step / grad_step and AgentPPO.update_rada2c / update_model, generator tests/golden/make_golden.py) and held
to the outputs the reference produced -- no torch composition of this package in between.

Reference: algos/test_cnn/RADTEAM_core.py:1586-1652 (PFGRUCell), NeuralNetworkCores/RADA2C_core.py:528-566 (step / grad_step),
algos/multiagent/ppo.py:1047-1281 (update_model / update_rada2c).
Tolerances (float32, other summation orders than torch's): activations rtol 1e-4 / atol 2e-6; gradients rtol 2e-3 / atol 2e-6 (BPTT
sums), the same bounds tests/test_rada2c_golden.py uses for the CPU composition."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(__file__)
GP = np.load(os.path.join(HERE, "golden", "pfgru.npz"))
GA = np.load(os.path.join(HERE, "golden", "rada2c_core.npz"))
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pf_cell(g, prefix="sd_"):
    from radiation_ppo_amd.pfgru import PFGRUCell
    cell = PFGRUCell(input_size=3, obs_size=3, hidden_size=24)
    cell.load_state_dict({k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)})
    return cell.cuda().eval()


def _k11(wts, obs3, h, p, eps, idx, carry):
    """One rs_pfgru_step_recorded launch for E independent problems (1 owner): obs3 [E, 3] -> pred [E, 2]; h / p updated in place
    when carry."""
    from radiation_ppo_amd import _lib
    E = obs3.shape[0]
    obs = torch.zeros(E, 1, 11, device="cuda")
    obs[:, 0, :3] = obs3
    from radiation_ppo_amd.pfgru import PredictorBank
    pred = torch.empty(E, 1, 2, device="cuda")
    hq = PredictorBank.to_quads(h)                       # the kernel's particle-set layout (include/radsearch.h: [A][N][6][40][4])
    _lib.check(_lib.load().rs_pfgru_step_recorded(wts.data_ptr(), obs.data_ptr(), hq.data_ptr(), p.data_ptr(), eps.contiguous().data_ptr(),
                                                  idx.to(torch.int32).contiguous().data_ptr(), None, 1 if carry else 0, 0.7, pred.data_ptr(),
                                                  E, 1, _stream()), "rs_pfgru_step_recorded")
    h.copy_(PredictorBank.from_quads(hq))
    return pred[:, 0]


def test_k11_on_the_reference_cells_recorded_run():
    """pfgru.npz: 12 steps with the hidden state carried (one launch per step, state written back by the kernel) and 12 steps each
    from the episode's h0 (the CNN harness' usage; one launch, 12 problems): prediction, resampled particles, log weights."""
    from radiation_ppo_amd.pfgru import pack_weights
    wts = pack_weights([_pf_cell(GP)])
    obs = cu(GP["obs"])
    T = obs.shape[0]
    # carried
    h = cu(GP["carry_h0"]).view(1, 1, 40, 24).clone()
    p = torch.full((1, 1, 40), math.log(1.0 / 40), device="cuda")
    for t in range(T):
        loc = _k11(wts, obs[t:t + 1], h, p, cu(GP["carry_eps"][t]).view(1, 1, 40, 24), cu(GP["carry_idx"][t]).view(1, 1, 40), True)
        assert torch.allclose(loc[0], cu(GP["carry_loc"][t]).reshape(-1), rtol=1e-4, atol=2e-6), (t, loc, GP["carry_loc"][t])
        assert torch.allclose(h[0, 0], cu(GP["carry_h"][t]), rtol=1e-4, atol=2e-6), t
        assert torch.allclose(p[0, 0], cu(GP["carry_p"][t]), rtol=1e-4, atol=4e-6), t
    # every step from h0: 12 independent problems in one launch
    h = cu(GP["fresh_h0"]).view(1, 1, 40, 24).expand(1, T, 40, 24).contiguous()
    p = torch.full((1, T, 40), math.log(1.0 / 40), device="cuda")
    loc = _k11(wts, obs, h, p, cu(GP["fresh_eps"]).view(1, T, 40, 24), cu(GP["fresh_idx"]).view(1, T, 40), True)
    assert torch.allclose(loc, cu(GP["fresh_loc"]).reshape(T, 2), rtol=1e-4, atol=2e-6)
    assert torch.allclose(h[0], cu(GP["fresh_h"]), rtol=1e-4, atol=2e-6) and torch.allclose(p[0], cu(GP["fresh_p"]), rtol=1e-4, atol=4e-6)


def _agent(**kw):
    from radiation_ppo_amd.rada2c import RNNAgentPPO
    ag = RNNAgentPPO(id=0, device="cuda", alpha=0.1, clip_ratio=0.2, target_kl=0.07, env_height=2500.0, train_pfgru_iters=1,
                     actor_learning_rate=3e-4, pfgru_learning_rate=5e-3, **kw)
    ag.agent.load_state_dict({k[3:]: torch.from_numpy(GA[k]) for k in GA.files if k.startswith("sd_")})
    return ag


def test_k11_k14_on_the_reference_step_sequence():
    """rada2c_core.npz `step_*`: ac.step (RADA2C_core.py:528-548) 14 times with both hidden states carried by the KERNELS: PFGRU
    prediction (K11 on the recorded noise / indices), GRU state, value and log-probability of the reference's action (K14)."""
    from radiation_ppo_amd.pfgru import pack_weights
    ag = _agent()
    wts = pack_weights([ag.agent.model])
    obs = cu(GA["step_obs"])
    h = cu(GA["step_pf_h0"]).view(1, 1, 40, 24).clone()
    p = torch.full((1, 1, 40), math.log(1.0 / 40), device="cuda")
    g = cu(GA["step_gru_h0"]).view(1, 24).clone()
    logits = torch.empty(1, 8, device="cuda"); val = torch.empty(1, device="cuda")
    for t in range(obs.shape[0]):
        loc = _k11(wts, obs[t:t + 1, :3], h, p, cu(GA["step_eps"][t]).view(1, 1, 40, 24), cu(GA["step_idx"][t]).view(1, 1, 40), True)
        assert torch.allclose(loc[0], cu(GA["step_loc"][t]), rtol=1e-4, atol=2e-6), t
        ag.policy_step_hip(obs[t:t + 1].contiguous(), loc.contiguous(), g, h_out=g, logits=logits, value=val)
        assert torch.allclose(g[0], cu(GA["step_gru_h"][t]), rtol=1e-4, atol=2e-6), t
        assert torch.allclose(val[0], cu(GA["step_val"][t]), rtol=1e-4, atol=2e-6), t
        logp = torch.log_softmax(logits, -1)[0, int(GA["step_act"][t])]
        assert torch.allclose(logp, cu(GA["step_logp"][t]), rtol=1e-4, atol=2e-6), t
    # the draw: K14's inverse CDF picks the reference's action for a uniform inside that action's CDF interval
    cdf = torch.cumsum(torch.softmax(logits[0], -1), 0)
    a = int(GA["step_act"][-1])
    u = torch.tensor([float((cdf[a - 1] if a else 0.0) + (cdf[a] - (cdf[a - 1] if a else 0.0)) * 0.5)], device="cuda")
    act = torch.empty(1, dtype=torch.int64, device="cuda"); lp = torch.empty(1, device="cuda")
    # state before the last step: recompute from the golden
    g_prev = cu(GA["step_gru_h"][-2]).view(1, 24).contiguous()
    ag.policy_step_hip(obs[-1:].contiguous(), cu(GA["step_loc"][-1]).view(1, 2), g_prev, u=u, act=act, logp=lp)
    assert int(act[0]) == a and torch.allclose(lp[0], cu(GA["step_logp"][-1]), rtol=1e-4, atol=2e-6)


def _episodes():
    from radiation_ppo_amd.rada2c import pack_episodes
    n = int(GA["n_eps"])
    eps = [GA[f"ep{i}"] for i in range(n)]
    lens = [e.shape[0] for e in eps]
    cat = np.concatenate(eps, 0)
    cut = np.zeros(cat.shape[0], dtype=np.uint8)
    cut[np.cumsum(lens) - 1] = 1
    f = lambda a: cu(a).unsqueeze(1)
    B = pack_episodes(f(cat[:, :11]), f(cat[:, 14].astype(np.int64)), f(cat[:, 11]), f(cat[:, 12]), f(cat[:, 13]), f(cat[:, 15:17]),
                      f(cut), n_total=1)
    return B, lens


def _draws(prefix, order, lens, L, with_gru):
    from radiation_ppo_amd.rada2c import RecordedKernelDraws
    E = len(lens)
    pf = torch.zeros(E, 40, 24); gru = torch.zeros(E, 24)
    eps = torch.zeros(L, E, 40, 24); idx = torch.zeros(L, E, 40, dtype=torch.int64)
    for k, e in enumerate(order):                      # draw set k was consumed by episode order[k]
        e = int(e); n = lens[e]
        pf[e] = torch.from_numpy(GA[f"{prefix}_pf_h0_{k}"])
        if with_gru:
            gru[e] = torch.from_numpy(GA[f"{prefix}_gru_h0_{k}"])
        eps[:n, e] = torch.from_numpy(GA[f"{prefix}_eps_{k}"]); idx[:n, e] = torch.from_numpy(GA[f"{prefix}_idx_{k}"])
    return RecordedKernelDraws(pf.cuda(), gru.cuda(), eps.cuda(), idx.cuda())


def test_k11_k12_k15_on_the_reference_update_rada2c():
    """update_rada2c (ppo.py:1150-1281) on the reference's five episodes with its recorded draws: PFGRU passes on K11, GRU recurrence
    and BPTT on K12, heads + loss + their back-propagation on K15 -> loss, KL, entropy, clip fraction, value loss, the KL decision,
    every pi gradient and the parameters after the Adam step, as the reference computed them."""
    ag = _agent()
    assert getattr(ag, "use_k12", True) and getattr(ag, "use_k15", True)
    B, lens = _episodes()
    d = _draws("a2c", GA["a2c_order"], lens, B.X.shape[0], True)
    from radiation_ppo_amd import _lib
    _lib.EVENTS = {}
    try:
        s, term = ag.update_rada2c(B, 0, draws_for=lambda it, sl: d)
    finally:
        _lib.EVENTS = None
    assert term == bool(GA["a2c_term"])
    for got, key in ((s[4], "a2c_loss"), (s[0], "a2c_kl"), (s[1], "a2c_ent"), (s[2], "a2c_cf"), (s[3], "a2c_val_loss")):
        assert np.isclose(got, float(GA[key]), rtol=1e-4, atol=1e-6), (key, got, float(GA[key]))
    after = ag.agent.state_dict()
    for name, prm in ag.agent.pi.named_parameters():
        want = cu(GA["a2c_grad_" + name])
        assert torch.allclose(prm.grad, want, rtol=2e-3, atol=2e-6), (name, float((prm.grad - want).abs().max()))
        assert torch.allclose(after["pi." + name], cu(GA["a2c_after_pi." + name]), rtol=1e-4, atol=2e-6), name


def test_k13_on_the_reference_update_model():
    """update_model (ppo.py:1047-1148) on the same episodes: ONE rs_pfgru_train launch with the reference's h0 / noise and the
    resampling indices torch.multinomial returned (the kernel's index-input mode) -> the loss, the clipped gradients and the
    parameters after the Adam step."""
    ag = _agent()
    B, lens = _episodes()
    d = _draws("model", np.arange(len(lens)), lens, B.X.shape[0], False)
    loss = ag.update_model(B, draws_for=lambda it, sl: d)
    assert len(ag.k13_particle_steps) == 1 and ag.k13_particle_steps[0] == 40 * sum(lens)        # the K13 path ran
    assert np.isclose(loss, float(GA["model_loss"]), rtol=1e-4), (loss, float(GA["model_loss"]))
    after = ag.agent.model.state_dict()
    for name, prm in ag.agent.model.named_parameters():
        want = cu(GA["model_grad_" + name])
        assert torch.allclose(prm.grad, want, rtol=2e-3, atol=2e-6), (name, float((prm.grad - want).abs().max()), float(want.abs().max()))
        # Adam's first step is lr * g / (|g| + 1e-8): where the gradient itself is ~1e-8 the step amplifies float32 noise, so those
        # elements are only held to |step| <= lr (the bound of the CPU test)
        big = want.abs() > 1e-6
        diff = (after[name] - cu(GA["model_after_" + name])).abs()
        assert (diff[big] <= 5e-6 + 1e-4 * after[name][big].abs()).all() and (diff <= 5e-3 + 1e-6).all(), name
