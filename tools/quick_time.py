"""Quick timing probe of dw_step (HIP events on the current stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk

for N in [int(a) for a in sys.argv[1:]] or [4096, 16384]:
    env = DyrosDynamicWalk(default_cfg(N, "cuda:0"), "cuda:0", 0, True)
    g = torch.Generator(device="cuda").manual_seed(42)
    acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
    for i in range(50):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 200
    e0.record()
    nres = 0
    for i in range(K):
        env.step(acts[i % 8])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("N=%d  %.3f ms/step  %.2f M env-steps/s  resets/step=%.1f nan_resets=%d" % (
        N, ms, N / ms / 1e3, float(env.reset_buf.sum()), int(env.nan_resets.sum())), flush=True)
    env.close()
