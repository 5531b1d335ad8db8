import torch, time
x = torch.empty(2_000_000_000, dtype=torch.float32, device="cuda")   # 8 GB
y = torch.empty_like(x)
for name, fn, bytes_ in (("fill 8 GB", lambda: x.fill_(1.0), 8e9), ("copy 8 GB (read + write)", lambda: y.copy_(x), 16e9), ("x*2 in place", lambda: x.mul_(2.0), 16e9)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: {ms:.2f} ms = {bytes_ / ms / 1e9:.2f} TB/s")
