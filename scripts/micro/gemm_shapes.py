"""Timing of the three GEMMs of Linear(2704, 32) at an update chunk (S = 524288) in several formulations (HIP events)."""
import torch
S, K, N = 524288, 2704, 32
x = torch.randn(S, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.zeros(N, device="cuda")
g = torch.randn(S, N, device="cuda")
wt = w.t().contiguous()
def timeit(name, fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:48s} {ms:7.3f} ms   ({S * K * 4 / ms / 1e6:6.0f} GB/s of the big operand)", flush=True)
timeit("fwd  F.linear(x, w, b)", lambda: torch.nn.functional.linear(x, w, b))
timeit("fwd  torch.addmm(b, x, wt)", lambda: torch.addmm(b, x, wt))
timeit("fwd  bmm slabs of 4096", lambda: torch.bmm(x.view(-1, 4096, K), wt.unsqueeze(0).expand(S // 4096, K, N)))
timeit("dX   g @ w", lambda: g @ w)
timeit("dX   F.linear(g, wt)", lambda: torch.nn.functional.linear(g, wt))
timeit("dX   bmm slabs of 4096", lambda: torch.bmm(g.view(-1, 4096, N), w.unsqueeze(0).expand(S // 4096, N, K)))
out = torch.empty(S, K, device="cuda")
timeit("dX   torch.mm(g, w, out=)", lambda: torch.mm(g, w, out=out))
timeit("dW   g.t() @ x", lambda: g.t() @ x)
timeit("dW   bmm slabs of 4096 + sum", lambda: torch.bmm(g.view(-1, 4096, N).transpose(1, 2), x.view(-1, 4096, K)).sum(0))
timeit("dW   bmm slabs of 16384 + sum", lambda: torch.bmm(g.view(-1, 16384, N).transpose(1, 2), x.view(-1, 16384, K)).sum(0))
timeit("copy x (read + write 5.7 GB each)", lambda: out.copy_(x))
