"""Distribution of the wave lifetimes of one step-kernel launch (library built with -DDQ_WAVE_TIME: tools/abl_build.sh DQ_WAVE_TIME for
the octet kernels, tools/ab_lib.sh wt -DDQ_WAVE_TIME + DW_PIPE=2 for the quad kernels; every wave leaves its cycle count in stacked_rewards[first env of the wave, 14])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = default_cfg(N, "cuda:0")
PIPE = int(os.environ.get("DW_PIPE", "3"))
cfg["sim"]["mi355"]["pipeline"] = PIPE
EPW = 8 if PIPE == 3 else 16          # envs per wave
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(300):
    env.step(acts[i % 8])
allc, withr, without, nres, phys, post = [], [], [], [], [], []
for i in range(20):
    env.step(acts[i % 8])
    torch.cuda.synchronize()
    c = env._buf["stacked_rewards"].view(-1, EPW, 15)[:, 0, 14].cpu().numpy().astype(np.float64)
    phys.append(env._buf["stacked_rewards"].view(-1, EPW, 15)[:, 0, 13].cpu().numpy().astype(np.float64))
    post.append(env._buf["stacked_rewards"].view(-1, EPW, 15)[:, 1, 1:13].cpu().numpy().astype(np.float64))
    rn = env.reset_buf.view(-1, EPW).cpu().numpy().sum(axis=1)
    r = rn > 0
    allc.append(c); withr.append(c[r]); without.append(c[~r]); nres.append(rn)
c = np.concatenate(allc)
print("waves %d x %d launches: cycles min %.0f  p50 %.0f  mean %.0f  p90 %.0f  p99 %.0f  max %.0f" % (len(allc[0]), len(allc), c.min(), np.percentile(c, 50), c.mean(), np.percentile(c, 90), np.percentile(c, 99), c.max()))
print("per launch: mean of max %.0f, mean of mean %.0f" % (np.mean([x.max() for x in allc]), np.mean([x.mean() for x in allc])))
a, b = np.concatenate(withr), np.concatenate(without)
print("waves with a reset this step: %d, mean %.0f; without: %d, mean %.0f" % (len(a), a.mean() if len(a) else 0, len(b), b.mean()))
rn = np.concatenate(nres)
ph = np.concatenate(phys)
for k in range(0, 5):
    m = rn == k
    if m.any(): print("  %d resets in the wave: %5d waves, mean %.0f (up to the end of the physics %.0f, after it %.0f), p95 %.0f, max %.0f" % (k, m.sum(), c[m].mean(), ph[m].mean(), (c[m] - ph[m]).mean(), np.percentile(c[m], 95), c[m].max()))
top = np.argsort(c)[-20:]
print("  resets in the 20 slowest waves:", rn[top].tolist())
if PIPE == 3:
    po = np.concatenate(post)
    names = ["stage..Q3", "r:terrain", "r:draws", "r:DR", "r:joints", "r:zero", "r:scalars", "taps+Q4 obs", "Q5 obs_buf", "Q6", "write back"]
    for k in range(0, 3):
        m = rn == k
        if m.any(): print("  post phases, %d resets: " % k + ", ".join("%s %.0f" % (names[i], po[m][:, i].mean()) for i in range(len(names) if k else 1)))
