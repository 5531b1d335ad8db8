"""Error of the fused PPO gradient pass (the variant selected by RS_GRAD_V) against float64 PyTorch autograd."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.ppo import FFActorCritic, FusedPPOGrad
torch.manual_seed(0)
M = 1 << 18
ac = FFActorCritic().cuda()
with torch.no_grad():
    for p in ac.parameters():
        p.mul_(2.0)
X = torch.randn(M, 11, device="cuda"); X[:, 0] = X[:, 0] * 3
act = torch.randint(0, 8, (M,), device="cuda"); adv = torch.randn(M, device="cuda"); ret = torch.randn(M, device="cuda")
w = torch.rand(M, device="cuda"); w = w / w.sum()
with torch.no_grad():
    lpo = ac.evaluate(X, act)[0] + 0.1 * torch.randn(M, device="cuda")
f = FusedPPOGrad(ac)
stats, grads = f(X, act, adv, ret, lpo, w, 0.2, 0.1)
g = grads.clone().double().cpu(); st = stats.clone().cpu()
ac64 = copy.deepcopy(ac).double().cpu()
X6, a6, adv6, ret6, lpo6, w6 = (t.double().cpu() if t.is_floating_point() else t.cpu() for t in (X, act, adv, ret, lpo, w))
logp, v, ent = ac64.evaluate(X6, a6)
ratio = torch.exp(logp - lpo6)
surr = torch.min(ratio * adv6, torch.clamp(ratio, 0.8, 1.2) * adv6)
vl = (w6 * (v - ret6) ** 2).sum()
loss = -((w6 * surr).sum() - 0.01 * vl + 0.1 * (w6 * ent).sum().detach())
loss.backward()
order = [ac64.actor[0].weight, ac64.actor[0].bias, ac64.actor[2].weight, ac64.actor[2].bias, ac64.actor[4].weight, ac64.actor[4].bias,
         ac64.critic[0].weight, ac64.critic[0].bias, ac64.critic[2].weight, ac64.critic[2].bias, ac64.critic[4].weight, ac64.critic[4].bias]
names = ["a.w1", "a.b1", "a.w2", "a.b2", "a.w3", "a.b3", "c.w1", "c.b1", "c.w2", "c.b2", "c.w3", "c.b3"]
o = 0
worst = 0.0
out = []
for n, p in zip(names, order):
    r = p.grad.reshape(-1); k = r.numel()
    e = (g[o:o + k] - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
    out.append(f"{n} {e:.1e}"); worst = max(worst, e); o += k
print(f"RS_GRAD_V={os.environ.get('RS_GRAD_V', '2')}: worst max|err|/max|ref| = {worst:.2e}   loss err {abs(st[4].item() - loss.item()):.1e}   [" + ", ".join(out) + "]")
