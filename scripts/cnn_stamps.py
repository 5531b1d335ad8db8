"""Where do K9 (rs_cnn_fwd_kernel) and K10 (rs_cnn_bwd_kernel) spend their cycles?  Diagnostic build with s_memtime stamps at the phase
boundaries (build(defines=["RS_CNN_STAMPS"], suffix="_cnnst") -> lib/librs_hip_cnnst.so, selected through RS_LIB_PATH): cycles per image
and phase, per wave of the workgroup (K10's fourth wave is the helper).  Read the SHARES (the stamps pin the schedule at the boundaries).
    python scripts/cnn_stamps.py            (on the GPU box; builds the variant first)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from radiation_ppo_amd import build as rs_build  # noqa: E402

os.environ["RS_LIB_PATH"] = rs_build.build(defines=["RS_CNN_STAMPS"], suffix="_cnnst", verbose=False)
import torch  # noqa: E402

from radiation_ppo_amd import _lib  # noqa: E402

lib = _lib.load()
lib.rs_debug_cnn_stamps.restype = C.c_int
lib.rs_debug_cnn_stamps.argtypes = [C.c_void_p, C.c_int]
S, A = 32768, 4
g = torch.Generator(device="cuda"); g.manual_seed(0)
maps = (torch.rand(S, 4, 27, 27, device="cuda", generator=g) * (torch.rand(S, 4, 27, 27, device="cuda", generator=g) < 0.15)).contiguous()
cells = torch.randint(0, 729, (S, A), device="cuda", generator=g)
pcells = torch.where(torch.rand(S, A, device="cuda", generator=g) < 0.9, torch.randint(0, 729, (S, A), device="cuda", generator=g),
                     torch.full((S, A), -1, device="cuda", dtype=torch.int64))
st = torch.cuda.current_stream().cuda_stream
PH = {0: ["stage to LDS + next round's loads issued", "barrier", "conv1 + ReLU + pool + stamps + p1 / amax stores", "barrier", "conv2 + ReLU + a2 stores", "barrier", "-", "-"],
      1: ["stage to LDS + next image's loads issued", "barrier", "dW2 (MFMA, this wave's k-steps)", "dP1 = conv2^T(dZ2)", "barrier", "dW1 (arg-max gathers)",
          "one-hot stamp gathers", "barrier"]}
for cin, agent in ((6, 0), (4, -1)):
    w1 = torch.randn(8, cin, 3, 3, device="cuda") * 0.2; b1 = torch.rand(8, device="cuda") * 0.1
    w2 = torch.randn(16, 8, 3, 3, device="cuda") * 0.1; b2 = torch.rand(16, device="cuda") * 0.1
    a2 = torch.empty(S, 2704, device="cuda"); p1 = torch.empty(S, 169, 8, device="cuda")
    am = torch.empty(S, 169, 8, dtype=torch.uint8, device="cuda"); mk = torch.empty(S, 169, dtype=torch.int16, device="cuda")
    da2 = torch.randn(S, 2704, device="cuda")
    rows, row = lib.rs_cnn_trunk_slab_rows(S, cin), lib.rs_cnn_trunk_slab_row(cin)
    slab = torch.empty(rows, row, device="cuda")
    cp = (cells.data_ptr(), pcells.data_ptr()) if agent >= 0 else (None, None)
    wt = torch.empty(lib.rs_cnn_trunk_scratch_floats(cin), device="cuda")

    def fwd():
        lib.rs_cnn_trunk_forward(maps.data_ptr(), cp[0], cp[1], A, agent, S, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                 a2.data_ptr(), p1.data_ptr(), am.data_ptr(), mk.data_ptr(), wt.data_ptr(), st)

    def bwd():
        lib.rs_cnn_trunk_backward(maps.data_ptr(), cp[0], cp[1], A, agent, S, w2.data_ptr(), da2.data_ptr(), mk.data_ptr(), p1.data_ptr(),
                                  am.data_ptr(), slab.data_ptr(), wt.data_ptr(), st)

    for k, (name, fn, nw) in enumerate((("K9 rs_cnn_fwd_kernel (training forward)", fwd, 8), ("K10 rs_cnn_bwd_kernel", bwd, 4))):
        fn(); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 128)()
        lib.rs_debug_cnn_stamps(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        lib.rs_debug_cnn_stamps(buf, 0)
        t = [[buf[(k * 8 + w) * 8 + q] for q in range(8)] for w in range(8)]
        print(f"cin={cin} {name}: stamped build {e0.elapsed_time(e1) * 1e3:.0f} us per {S} images")
        classes = [("waves 0-2 (pixel waves)", [0, 1, 2]), ("wave 3 (helper)", [3])] if k == 1 else [("all 8 waves", list(range(8)))]
        for cname, ws in classes:
            tot = sum(sum(t[w]) for w in ws)
            per_img = tot / len(ws) / S * (1 if k == 1 else 3)            # K9: three images per workgroup round
            print(f"  {cname}: {per_img:.0f} cycles per image{' (x3 per round)' if k == 0 else ''}")
            for q in range(8):
                c = sum(t[w][q] for w in ws)
                if PH[k][q] != "-":
                    print(f"    {PH[k][q]:52s} {c / len(ws) / S * (1 if k == 1 else 3):8.0f} cycles {100.0 * c / tot:5.1f} %")
