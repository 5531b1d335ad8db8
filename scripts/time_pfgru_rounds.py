"""K11: does a launch cost whole rounds of workgroups?  One step for N (owner, env) sets at 6 sets per workgroup and 1024 resident workgroups
(256 CUs x 4): 6144 sets = 1 round, 12288 = 2, 16384 = 2.67, 18432 = 3; and two independent banks stepped on two streams at once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd.pfgru import PredictorBank

def bank(n):
    b = PredictorBank(n, 1, seed=1, carry_hidden=True, device="cuda")
    b.reset()
    return b, torch.rand(n, 1, 11, device="cuda")

def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for n in (6144, 9216, 12288, 16384, 16500, 18432, 24576):
    b, obs = bank(n)
    print(f"{n:6d} sets ({n / 6144:.2f} rounds): {timed(lambda: b.predict(obs)):7.1f} us per step", flush=True)

# two independent passes at once (two streams): per-step time of the pair
n = 16500
(b1, o1), (b2, o2) = bank(n), bank(n)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def pair():
    with torch.cuda.stream(s1):
        b1.predict(o1)
    with torch.cuda.stream(s2):
        b2.predict(o2)
torch.cuda.synchronize()
t = timed(pair)
print(f"two banks of {n} sets on two streams: {t:7.1f} us per pair of steps = {t / 2:.1f} us per step")
