import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ["DW_LIB"]
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
cfg = default_cfg(16384, "cuda:0"); cfg["sim"]["mi355"]["debug_no_post"] = 1
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(16384, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(40): env.step(acts[i % 8])
torch.cuda.synchronize()
st = env._buf["gate_acc"][200:256].cpu().numpy().astype("int64")
print("inward step starts (substep 0):", [int(st[42+s+1]-st[42+s]) for s in range(10)], "inward total", int(st[1+4]-st[1+3]))
