import sys, os
sys.path.insert(0, os.getcwd())
import torch
from isaacgymdyros_amd.config import default_cfg, with_terrain
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
PIPE = int(os.environ.get("DW_PIPE", "0"))
WARM, STEPS = int(os.environ.get("DW_WARM", "50")), int(os.environ.get("DW_STEPS", "200"))
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
for N in (4096, 16384):
    cfg = with_terrain(default_cfg(N, "cuda:0"), mesh_type="trimesh", curriculum=True)
    cfg["sim"]["mi355"]["pipeline"] = PIPE
    cfg["sim"]["mi355"]["alias_obs"] = True
    env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
    g = torch.Generator(device="cuda").manual_seed(42)
    acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
    for i in range(WARM): env.step(acts[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(STEPS): env.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / STEPS
    print("pipeline %d " % PIPE + "terrain N=%d %.3f ms/step %.2f M env-steps/s levels mean %.2f" % (N, ms, N / ms / 1e3, float(env.terrain_levels.float().mean())))
