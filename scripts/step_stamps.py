"""Phase cycles of the obstacle env step (diagnostic build: python -c "from radiation_ppo_amd.build import build;
build(force=True, defines=['RS_STEP_STAMPS'], suffix='_estamps')", then RS_LIB_PATH=.../librs_hip_estamps.so python scripts/step_stamps.py).
Wave-level s_memtime differences summed over all waves and launches of rs_step (rs_step4_kernel: 16 envs per wave)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiation_ppo_amd import _lib
from radiation_ppo_amd.envs import RadSearchVec
lib = _lib.load()
lib.rs_debug_step_stamps.restype = C.c_int
lib.rs_debug_step_stamps.argtypes = [C.c_void_p, C.c_int]
names = ["state loads, collision proposals", "take_action, in_obstruction", "shortest path", "is_intersect", "Poisson measurement",
         "reward, observation head", "obstruction sensors", "write back"]
for N, A in ((8192, 1), (4096, 4)):
    env = RadSearchVec(N, number_agents=A, obstruction_count=-1, enforce_grid_boundaries=True, seed=289714752)
    env.reset()
    acts = [torch.randint(0, 8, (N, A), device="cuda").to(torch.int8) for _ in range(16)]
    for i in range(30):
        env.step(acts[i % 16])
    buf = (C.c_ulonglong * 16)()
    lib.rs_debug_step_stamps(buf, 1)
    reps = 200
    for i in range(reps):
        env.step(acts[i % 16])
    lib.rs_debug_step_stamps(buf, 0)
    waves = (N + 15) // 16
    tot = sum(buf[i] for i in range(8))
    print(f"{N} envs x {A} agents: {tot / reps / waves:.0f} wave-cycles per step launch and wave")
    for i, nm in enumerate(names):
        print(f"   {nm:36s} {buf[i] / reps / waves:9.0f} cyc  {100.0 * buf[i] / tot:5.1f} %")
