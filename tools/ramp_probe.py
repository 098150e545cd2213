"""Step time in windows of 128 steps from a cold start (fresh box): shows how long the device takes to reach its steady rate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = default_cfg(N, "cuda:0"); cfg["sim"]["mi355"]["alias_obs"] = True
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
env.reset()
t_start = time.perf_counter()
for w in range(32):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(128):
        env.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    print("window %2d  t=%.3f s  %.4f ms/step  finished episodes %d" % (w, time.perf_counter() - t_start, e0.elapsed_time(e1) / 128, int(env.episodes_finished.sum())), flush=True)
