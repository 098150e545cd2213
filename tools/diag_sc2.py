"""Which Gym bodies the device loads differently from the oracle after one substep of random arm poses (self-collision detection check)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from hip_backend import make_env
from test_kernel_emulation import _random_arm_poses
from oracle.oracle import OracleSim
from isaacgymdyros_amd.model import load_model
from isaacgymdyros_amd.task_constants import load_task_constants
names = list(load_model().body_names)
N, wb = int(sys.argv[1]), int(sys.argv[2])
env = make_env(N, randomize=False, debug_wave_build=wb)
b = env._buf
b["root_states"][:, 0:2] = 0; b["root_states"][:, 2] = 3.0
b["dof_state"][..., 0] = torch.from_numpy(_random_arm_poses(N, seed=8)).cuda(); b["dof_state"][..., 1] = 0
ora = OracleSim(N, task_const=load_task_constants(), cfg=env._ccfg)
for k, t in env._buf.items():
    ora.buf[k][...] = t.cpu().numpy().reshape(ora.buf[k].shape)
tau = torch.zeros(N, 33)
env.simulate(tau.cuda()); ora.simulate(tau.numpy()); torch.cuda.synchronize()
cg, co = env.contact_forces.cpu().numpy(), ora.buf["contact_forces"]
lg, lo = np.linalg.norm(cg, axis=2) > 1.0, np.linalg.norm(co, axis=2) > 1.0
bad = np.argwhere(lg != lo)
print("N", N, "wave_build", wb, "mismatches", len(bad), "max rel force diff", np.abs(cg - co).max() / np.abs(co).max())
for e, g in bad[:12]:
    print(e, names[g], "oracle |F|", np.linalg.norm(co[e, g]), "device |F|", np.linalg.norm(cg[e, g]), "oracle loaded:", [names[k] for k in np.nonzero(lo[e])[0]])
