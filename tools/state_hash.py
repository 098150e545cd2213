"""Hash of the device state over a seeded 150-step rollout of 4096 envs: two builds of the library (DW_LIB=...) that print the same
hash compute the same bits (used to show that a restructuring of the kernels changed no result).  usage: [DW_LIB=path] python tools/state_hash.py"""
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
N = 4096
env = DyrosDynamicWalk(default_cfg(N, "cuda:0"), "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(7)
h = hashlib.sha256()
for t in range(150):
    o, r, d, x = env.step(torch.rand(N, 13, generator=g, device="cuda") * 2 - 1)
    if t % 10 == 9:
        for k in ("root_states", "dof_state", "contact_forces", "env_state", "obs_buf", "rew_buf", "reset_buf"):
            h.update(env._buf[k].cpu().numpy().tobytes())
print("state hash after 150 steps of %d envs:" % N, h.hexdigest()[:24])
