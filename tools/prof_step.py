"""Small fixed workload for profiling: N envs, a few steps of dw_step (and optionally dw_simulate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd.config import default_cfg, with_terrain
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = default_cfg(N, "cuda:0")
cfg["sim"]["mi355"]["pipeline"] = int(os.environ.get("DW_PIPE", "0"))
if os.environ.get("DW_TERRAIN"): cfg = with_terrain(cfg, mesh_type="trimesh", curriculum=True)
if os.environ.get("DW_FREEZE"): cfg["sim"]["mi355"]["debug_freeze_physics"] = True
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(steps):
    env.step(acts[i % 8])
if os.environ.get("DW_SIM"):
    tau = (torch.rand(N, 33, generator=g, device="cuda") * 2 - 1) * 20
    for i in range(steps):
        env.simulate(tau)
torch.cuda.synchronize()
env.close()
