"""Phase breakdown by timing variants (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS, KP_RAW, KV_RAW
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384

def timeit(fn, K=100, W=20):
    for _ in range(W): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K

g = torch.Generator(device="cuda").manual_seed(42)
a = torch.rand(N, 13, generator=g, device="cuda") * 2 - 1
env = DyrosDynamicWalk(default_cfg(N, "cuda:0"), "cuda:0", 0, True)
print("step full            %.3f ms" % timeit(lambda: env.step(a)))
env.close()
cfg = default_cfg(N, "cuda:0"); cfg["sim"]["mi355"]["debug_freeze_physics"] = True
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
print("step, physics frozen %.3f ms" % timeit(lambda: env.step(a)))
env.close()
cfg = default_cfg(N, "cuda:0"); cfg["task"]["randomize"] = False
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
kp = torch.tensor(KP_RAW, device="cuda"); kv = torch.tensor(KV_RAW, device="cuda"); q0 = torch.tensor(INITIAL_DOF_POS, device="cuda")
tau = torch.zeros(N, 33, device="cuda")
def stand():
    env.simulate(tau)
for _ in range(300):
    env.simulate(kp * (q0 - env.dof_pos) - kv * env.dof_vel)
tau = (kp * (q0 - env.dof_pos) - kv * env.dof_vel).contiguous()
print("simulate standing    %.3f ms (one substep; contacts active)" % timeit(stand, K=50, W=0))
env._buf["root_states"][:, 2] = 50.0
env._buf["root_states"][:, 7:] = 0
print("simulate in flight   %.3f ms (one substep; no contact)" % timeit(stand, K=50, W=0))
env.close()
