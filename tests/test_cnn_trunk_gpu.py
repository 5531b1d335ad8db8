"""K9/K10 (rs_cnn_trunk_forward / _backward, through the C ABI via radiation_ppo_amd.maps.ConvTrunk) against
(a) plain PyTorch fp32: the same modules' nn.Sequential (library convolutions + autograd) on the dense stacks;
(b) the reference's own CNN losses and gradients (tests/golden/cnn_loss.npz, made by the reference's
    compute_batched_losses_pi / _critic on its Actor / Critic).
Tolerance: fp32, different summation order -- rtol 2e-4, atol 2e-6 relative to the tensor's scale."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dense_actor_stack(maps, cells, pcells, a):
    """CNNBase.get_map_stack (RADTEAM_core.py:1791-1836) written with torch ops."""
    B = maps.shape[0]
    loc = torch.zeros(B, 729, device=maps.device)
    c = cells[:, a:a + 1]
    loc.scatter_(1, c.clamp(min=0), (c >= 0).float())     # -1: no position recorded yet (fresh maps) -> an empty location map
    pm = torch.zeros(B, 729, device=maps.device)
    pc = pcells[:, a:a + 1]
    pm.scatter_(1, pc.clamp(min=0), (pc >= 0).float())
    loc = loc.view(B, 1, 27, 27)
    return torch.cat([pm.view(B, 1, 27, 27), loc, maps[:, 0:1] - loc, maps[:, 1:4]], dim=1)


def _random_inputs(S, A, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    maps = torch.rand(S, 4, 27, 27, device="cuda", generator=g) * (torch.rand(S, 4, 27, 27, device="cuda", generator=g) < 0.1)
    maps[:, 1] = maps[:, 1] * 6 - 3                       # z-scored readings take both signs
    cells = torch.randint(0, 729, (S, A), device="cuda", generator=g)
    pcells = torch.where(torch.rand(S, A, device="cuda", generator=g) < 0.6, torch.randint(0, 729, (S, A), device="cuda", generator=g),
                         torch.full((S, A), -1, device="cuda", dtype=torch.int64))
    maps[:, 0] = torch.round(maps[:, 0] * 3)
    maps[:, 0].view(S, 729).scatter_add_(1, cells, torch.ones(S, A, device="cuda"))   # every owner stands somewhere
    # corners and edges exercise the zero padding and the dropped 27th row/column
    cells[0, 0], cells[1 % S, 0], cells[2 % S, 0] = 0, 728, 26
    if S > 4:
        cells[3, 0] = -1                                  # a fresh map without a recorded position (maps.actor_stack_from: empty one-hot)
    return maps.contiguous(), cells.contiguous(), pcells.contiguous()


def _close(a, b, name, noise=2e-5):
    """b is the float64 reference.  fp32 sums over up to 2600 x 169 terms carry ~sqrt(n) * 6e-8 relative noise
    against the largest element, on top of the per-element rtol."""
    a = a.detach().double().cpu()
    scale = max(float(b.abs().max()), 1e-6)
    assert torch.allclose(a, b, rtol=2e-4, atol=noise * scale + 1e-6), (name, float((a - b).abs().max()), scale)


def _fragile(seq64, dense64, eps=2e-6):
    """Samples that sit on a kink of the network in float64: a ReLU pre-activation or a max-pool margin within eps.
    There fp32 and fp64 may legitimately take different branches (the gradient is discontinuous), so such samples
    get zero weight in the comparison."""
    import torch.nn.functional as F
    z1 = F.conv2d(dense64, seq64[0].weight, seq64[0].bias, padding=1)
    blocks = F.unfold(torch.relu(z1), kernel_size=2, stride=2)                        # [S, 8*4, 169]
    S = dense64.shape[0]
    blocks = blocks.view(S, 8, 4, 169)
    top2 = blocks.topk(2, dim=2).values
    distinct = (blocks.max(dim=2, keepdim=True).values - blocks).abs()               # margins to the winner
    near_tie = ((distinct > 0) & (distinct < eps)).any(dim=2)                         # exact ties break identically
    near_zero_pool = (top2[:, :, 0] > 0) & (top2[:, :, 0] < eps)
    p1 = F.max_pool2d(torch.relu(z1), 2, 2)
    z2 = F.conv2d(p1, seq64[3].weight, seq64[3].bias, padding=1)
    near_zero_z2 = z2.abs() < eps
    return near_tie.flatten(1).any(1) | near_zero_pool.flatten(1).any(1) | near_zero_z2.flatten(1).any(1)


def _grads(module, out, wgt):
    for p in module.parameters():
        p.grad = None
    (out * wgt).sum().backward()
    return [p.grad.detach().clone() for p in module.parameters()]


@pytest.mark.parametrize("S", [1, 2, 5, 193, 2600])   # one image, a partly filled round of three, fewer than the grid, several rounds per workgroup
def test_trunk_forward_and_backward_match_torch(S):
    """Reference = the same nn.Sequential modules evaluated by PyTorch in float64 on the CPU (dense stacks)."""
    import copy
    from radiation_ppo_amd.maps import CNNActor, CNNCritic
    torch.manual_seed(S)
    A = 3
    actor, critic = CNNActor().cuda(), CNNCritic().cuda()
    with torch.no_grad():                                  # make biases matter (pool ties on empty regions, ReLU gates)
        for m in (actor.actor, critic.critic):
            m[0].bias.uniform_(-0.05, 0.15)
            m[3].bias.uniform_(-0.1, 0.1)
    actor64, critic64 = copy.deepcopy(actor).double().cpu(), copy.deepcopy(critic).double().cpu()
    maps, cells, pcells = _random_inputs(S, A, seed=S)
    for a in (0, A - 1):
        dense64 = _dense_actor_stack(maps, cells, pcells, a).double().cpu()
        ref = actor64.logits(dense64)
        got = actor.logits_from_maps(maps, cells, pcells, a)
        _close(got, ref.detach(), f"actor logits a={a}")
        keep = ~_fragile(actor64.actor, dense64)
        assert keep.float().mean() >= 0.5                  # (images with a pool / ReLU near-tie carry no gradient weight in this test)
        wgt = torch.randn_like(got) * keep.cuda().unsqueeze(1)
        gref = _grads(actor64, ref, wgt.double().cpu())
        ggot = _grads(actor, got, wgt)
        for (k, _), g1, g0 in zip(actor.named_parameters(), ggot, gref):
            _close(g1, g0, f"actor grad {k} a={a}")
    ref = critic64(maps.double().cpu())
    got = critic.value_from_maps(maps)
    _close(got, ref.detach(), "critic value")
    keep = ~_fragile(critic64.critic, maps.double().cpu())
    wgt = torch.randn_like(got) * keep.cuda()
    gref = _grads(critic64, ref, wgt.double().cpu())
    ggot = _grads(critic, got, wgt)
    for (k, _), g1, g0 in zip(critic.named_parameters(), ggot, gref):
        _close(g1, g0, f"critic grad {k}")
    with torch.no_grad():                                  # inference path (no p1 / amax written)
        _close(actor.logits_from_maps(maps, cells, pcells, 1),
               actor64.logits(_dense_actor_stack(maps, cells, pcells, 1).double().cpu()), "no-grad fwd")


def test_trunk_path_matches_reference_losses_and_gradients(golden_dir):
    """CNNAgentPPO.update_agent fed by (maps, cells, pcells) -> HIP trunk, against the reference's own CNN losses."""
    from radiation_ppo_amd.ppo_cnn import CNNAgentPPO
    g = dict(np.load(os.path.join(golden_dir, "cnn_loss.npz")).items())
    ag = CNNAgentPPO(id=0, train_pi_iters=1, train_v_iters=1, actor_learning_rate=0.0, critic_learning_rate=0.0, target_kl=10.0)
    ag.pi.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("a_actor.")})
    ag.critic.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("c_critic.")})
    dev = lambda k, dt=None: torch.from_numpy(g[k]).cuda() if dt is None else torch.from_numpy(g[k]).cuda().to(dt)
    maps, cells, pcells = dev("maps"), dev("cells").view(-1, 1).contiguous(), dev("pcells").view(-1, 1).contiguous()
    n = maps.shape[0]
    w = torch.full((n,), 1.0 / n, device="cuda")
    r = ag.update_agent(lambda lo, hi: (maps[lo:hi], cells[lo:hi], pcells[lo:hi], 0), lambda lo, hi: (maps[lo:hi],),
                        dev("act", torch.int64), dev("adv"), dev("ret"), dev("logp_old"), w, update_critic=True)
    assert abs(r.loss_policy - float(g["pi_loss"])) < 2e-6 and abs(r.kl_divergence - float(g["kl"])) < 2e-6
    assert abs(r.Entropy - float(g["entropy"])) < 2e-6 and abs(r.ClipFrac - float(g["clip_fraction"])) < 1e-7
    assert abs(r.loss_critic - float(g["critic_loss"])) < 2e-6
    for k, p in ag.pi.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), g["ga_" + k], rtol=2e-4, atol=1e-6), k
    for k, p in ag.critic.named_parameters():
        assert np.allclose(p.grad.cpu().numpy(), g["gc_" + k], rtol=2e-4, atol=1e-6), k
