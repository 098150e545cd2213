"""Device and wall time of a 20-step timed region right after a synchronize, in the shapes bench.py can give it (why the driver's
--steps 20 line reads above the 256-step leg).  usage: python tools/first_steps.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
from isaacgymdyros_amd import dist as dwdist
N = 16384
def make(alias):
    cfg = default_cfg(N, "cuda:0"); cfg["sim"]["mi355"]["alias_obs"] = alias
    return DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
pool = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(64)]
def region(env, K, pre):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    pre(env)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(K): env.step(pool[i % 64])
    e1.record()
    torch.cuda.synchronize()
    w = time.perf_counter() - t0
    return e0.elapsed_time(e1) / K, w / K * 1e3
def pre_steps(n):
    def f(env):
        for i in range(n): env.step(pool[i % 64])
    return f
def pre_gather_then_steps(n):
    def f(env):
        dwdist.gather_episode_stats(env._buf["env_state"])
        for i in range(n): env.step(pool[i % 64])
    return f
for alias in (False, True):
    env = make(alias); env.reset()
    for i in range(3000): env.step(pool[i % 64])
    torch.cuda.synchronize()
    for name, K, pre in (("20 after 8 steps", 20, pre_steps(8)), ("20 after 64 steps", 20, pre_steps(64)), ("20 after gather+8", 20, pre_gather_then_steps(8)),
                         ("20 after nothing", 20, pre_steps(0)), ("256 after 8 steps", 256, pre_steps(8)), ("20 after 8 steps again", 20, pre_steps(8))):
        r = [region(env, K, pre) for _ in range(3)]
        print("alias_obs %s  %-24s device ms/step %s   wall ms/step %s" % (alias, name, " ".join("%.4f" % a for a, _ in r), " ".join("%.4f" % b for _, b in r)), flush=True)
    env.close()
