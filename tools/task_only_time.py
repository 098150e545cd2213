"""Time of dw_k_step with the physics frozen (debug_freeze_physics): the task-logic + record load/store share of a step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for freeze in (0, 1):
    cfg = default_cfg(N, "cuda:0")
    cfg["sim"]["mi355"]["debug_freeze_physics"] = freeze
    env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
    g = torch.Generator(device="cuda").manual_seed(42)
    acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
    for i in range(50): env.step(acts[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): env.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    print("freeze_physics=%d  %.4f ms/step  resets/step=%.1f" % (freeze, e0.elapsed_time(e1) / 200, float(env.reset_buf.sum())), flush=True)
    env.close()
