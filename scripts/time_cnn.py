"""HIP-event timing of the CNN trunk kernels (K9 forward, K10 backward) on a 32768-image chunk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd import _lib
lib = _lib.load()
S, A = 32768, 4
g = torch.Generator(device="cuda"); g.manual_seed(0)
maps = (torch.rand(S, 4, 27, 27, device="cuda", generator=g) * (torch.rand(S, 4, 27, 27, device="cuda", generator=g) < 0.15)).contiguous()
cells = torch.randint(0, 729, (S, A), device="cuda", generator=g); pcells = torch.where(torch.rand(S, A, device="cuda", generator=g) < 0.9, torch.randint(0, 729, (S, A), device="cuda", generator=g), torch.full((S, A), -1, device="cuda", dtype=torch.int64))
st = torch.cuda.current_stream().cuda_stream
for cin, agent in ((6, 0), (4, -1)):
    w1 = torch.randn(8, cin, 3, 3, device="cuda") * 0.2; b1 = torch.rand(8, device="cuda") * 0.1
    w2 = torch.randn(16, 8, 3, 3, device="cuda") * 0.1; b2 = torch.rand(16, device="cuda") * 0.1
    a2 = torch.empty(S, 2704, device="cuda"); p1 = torch.empty(S, 169, 8, device="cuda"); am = torch.empty(S, 169, 8, dtype=torch.uint8, device="cuda"); mk = torch.empty(S, 169, dtype=torch.int16, device="cuda")
    da2 = torch.randn(S, 2704, device="cuda")
    rows, row = lib.rs_cnn_trunk_slab_rows(S, cin), lib.rs_cnn_trunk_slab_row(cin)
    slab = torch.empty(rows, row, device="cuda")
    cp = (cells.data_ptr(), pcells.data_ptr()) if agent >= 0 else (None, None)
    wt = torch.empty(lib.rs_cnn_trunk_scratch_floats(cin), device="cuda")
    def fwd(train):
        lib.rs_cnn_trunk_forward(maps.data_ptr(), cp[0], cp[1], A, agent, S, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                 a2.data_ptr(), p1.data_ptr() if train else None, am.data_ptr() if train else None, mk.data_ptr() if train else None, wt.data_ptr(), st)
    def bwd():
        lib.rs_cnn_trunk_backward(maps.data_ptr(), cp[0], cp[1], A, agent, S, w2.data_ptr(), da2.data_ptr(), mk.data_ptr(), p1.data_ptr(),
                                  am.data_ptr(), slab.data_ptr(), wt.data_ptr(), st)
    for name, fn in (("fwd_infer", lambda: fwd(False)), ("fwd_train", lambda: fwd(True)), ("bwd", bwd)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"cin={cin} {name}: {ms*1e3:.0f} us  {S/ms/1e3:.1f} M img/s  slab_rows={rows}", flush=True)
