"""Time one fused PPO loss+gradient pass (K7: rs_ppo_grad = actor kernel + critic kernel + slab reduce) on a synthetic batch
of BASELINE config 2's size and report the fraction of the f32 MFMA peak; optional A/B against another build of the library
(RS_LIB_PATH) in separate processes is left to the caller.  HIP events on the launch stream, interleaved rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from radiation_ppo_amd.ppo import FFActorCritic, FusedPPOGrad  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 480
torch.manual_seed(0)
ac = FFActorCritic().cuda()
X = torch.randn(M, 11, device="cuda")
act = torch.randint(0, 8, (M,), device="cuda")
adv, ret, lpo = torch.randn(M, device="cuda"), torch.randn(M, device="cuda"), -2.0 + 0.1 * torch.randn(M, device="cuda")
w = torch.full((M,), 1.0 / M, device="cuda")
f = FusedPPOGrad(ac)
for _ in range(5):
    f(X, act, adv, ret, lpo, w, 0.2, 0.1)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f(X, act, adv, ret, lpo, w, 0.2, 0.1)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
ms = sorted(ts)[len(ts) // 2]
tf = 58240 * M / (ms * 1e-3) / 1e12
print(f"M={M}: {ms:.4f} ms per pass (min {min(ts):.4f})  {tf:.1f} TFLOP/s  frac {tf / 157.3:.3f}   lib={os.environ.get('RS_LIB_PATH', 'default')}")
# sanity: gradient vs float64 autograd on a slice
n = 4096
acd = FFActorCritic().double().cuda()
acd.load_state_dict({k: v.double() for k, v in ac.state_dict().items()})
Xs, a_s, adv_s, ret_s, lpo_s = X[:n].double(), act[:n], adv[:n].double(), ret[:n].double(), lpo[:n].double()
ws = torch.full((n,), 1.0 / n, device="cuda", dtype=torch.float64)
logp, v, ent = acd.evaluate(Xs, a_s)
ratio = torch.exp(logp - lpo_s)
surr = torch.min(ratio * adv_s, torch.clamp(ratio, 0.8, 1.2) * adv_s)
loss = -((ws * surr).sum() - 0.01 * (ws * (v - ret_s) ** 2).sum())
loss.backward()
ref = torch.cat([p.grad.reshape(-1) for p in (acd.actor[0].weight, acd.actor[0].bias, acd.actor[2].weight, acd.actor[2].bias, acd.actor[4].weight, acd.actor[4].bias,
                                              acd.critic[0].weight, acd.critic[0].bias, acd.critic[2].weight, acd.critic[2].bias, acd.critic[4].weight, acd.critic[4].bias)])
st, g = f(X[:n].contiguous(), act[:n].contiguous(), adv[:n].contiguous(), ret[:n].contiguous(), lpo[:n].contiguous(), ws.float(), 0.2, 0.1)
err = (g.double() - ref).abs().max().item() / ref.abs().max().item()
print(f"gradient check vs float64 autograd on {n} samples: max|err|/max|ref| = {err:.2e}")
