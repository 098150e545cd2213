"""Long random-action soak of dw_step (flat ground and terrain): counts non-finite resets and checks state bounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd.config import default_cfg, with_terrain, with_friction_randomization
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096          # (4096: the hex instantiation; 16384: the two-waves octet build)
for name, cfg in (("flat", with_friction_randomization(default_cfg(N, "cuda:0"))),
                  ("terrain", with_terrain(default_cfg(N, "cuda:0"), mesh_type="trimesh", curriculum=True))):
    cfg["sim"]["mi355"]["force_perturb_start"] = True
    env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
    g = torch.Generator(device="cuda").manual_seed(1)
    worst_z, worst_qd, worst_w, resets = 1e9, 0.0, 0.0, 0
    zero_act = torch.zeros(N, 13, device="cuda")
    for t in range(STEPS):
        # a third of the time the policy output is zero (robots stand / walk on the PD targets), otherwise random
        a = zero_act if (t // 500) % 3 == 0 else torch.rand(N, 13, generator=g, device="cuda") * 2 - 1
        obs, rew, done, ex = env.step(a)
        if t % 250 == 0:
            assert torch.isfinite(obs["obs"]).all() and torch.isfinite(rew).all(), t
            worst_z = min(worst_z, float(env.root_states[:, 2].min()))
            worst_qd = max(worst_qd, float(env.dof_vel.abs().max()))
            worst_w = max(worst_w, float(env.root_states[:, 10:13].norm(dim=1).max()))
        resets += int(done.sum()) if t % 50 == 0 else 0
    torch.cuda.synchronize()
    print("%s: %d steps x %d envs  nan_resets=%d  min root z %.3f  max |qd| %.3f  max |w| %.2f  sampled resets %d  mean episode length %.1f" % (
        name, STEPS, N, int(env.nan_resets.sum()), worst_z, worst_qd, worst_w, resets, float(env.epi_len_log.mean())), flush=True)
    env.close()
