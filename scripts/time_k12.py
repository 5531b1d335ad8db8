"""K12 alone: rs_gru_forward / rs_gru_backward behind rada2c.GRUSequence for L = 120 steps x E episodes (HIP events around the two launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from radiation_ppo_amd import _lib
L, E, H = 120, int(os.environ.get("K12_EPISODES", "16384")), 24
dev = "cuda"
torch.manual_seed(0)
gi = torch.randn(L, E, 72, device=dev) * 0.5
h0 = torch.rand(E, H, device=dev) - 0.5
w_hh = (torch.rand(72, H, device=dev) - 0.5) * 0.4
whh_t = torch.zeros(H, 80, device=dev); whh_t[:, :72] = w_hh.t()
bhh = torch.zeros(80, device=dev); bhh[:72] = torch.rand(72, device=dev) * 0.1
whh = torch.zeros(72, 32, device=dev); whh[:, :H] = w_hh
hs = torch.empty(L, E, H, device=dev); gates = torch.empty(L, E, 4 * H, device=dev)
dhs = torch.randn(L, E, H, device=dev) * 0.01
dgi = torch.empty(L, E, 72, device=dev); dgh = torch.empty_like(dgi)
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def fwd(): _lib.check(lib.rs_gru_forward(gi.data_ptr(), h0.data_ptr(), whh_t.data_ptr(), bhh.data_ptr(), hs.data_ptr(), gates.data_ptr(), L, E, st), "f")
def bwd(): _lib.check(lib.rs_gru_backward(dhs.data_ptr(), hs.data_ptr(), gates.data_ptr(), h0.data_ptr(), whh.data_ptr(), dgi.data_ptr(), dgh.data_ptr(), L, E, st), "b")
for name, fn in (("forward", fwd), ("backward", bwd)):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"K12 {name}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us for {E} episodes x {L} steps; checksum {float(hs.double().sum() + dgi.double().sum()):.6f}")
