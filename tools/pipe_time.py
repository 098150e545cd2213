"""A/B timing of the kernel generations (DwConfig.pipeline) in ONE process on one GPU box, interleaved so that clock drift
does not masquerade as a kernel difference: whole policy step (dw_step) and the Gym-boundary substep (dw_simulate).
usage: python tools/pipe_time.py [--pipes 3,4] [--envs 4096,16384] [--rounds 3] [--steps 200] [--sim]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd.config import default_cfg, with_terrain
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk

ap = argparse.ArgumentParser()
ap.add_argument("--pipes", default="3,4")
ap.add_argument("--envs", default="4096,16384")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--sim", action="store_true", help="also time dw_simulate (one physics substep at the Gym boundary)")
ap.add_argument("--terrain", action="store_true", help="cfg/terrain/terrain_cfg.py defaults (trimesh, curriculum) instead of the plane")
ap.add_argument("--mi355", default="{}", help='extra sim.mi355 settings as JSON, e.g. {"self_collision": 0}')
a = ap.parse_args()
pipes = [int(x) for x in a.pipes.split(",")]
extra = json.loads(a.mi355)
for N in [int(x) for x in a.envs.split(",")]:
    envs = {}
    for p in pipes:
        cfg = default_cfg(N, "cuda:0")
        if a.terrain:
            cfg = with_terrain(cfg, mesh_type="trimesh", curriculum=True)
        cfg["sim"]["mi355"].update(extra)
        cfg["sim"]["mi355"]["pipeline"] = p
        cfg["sim"]["mi355"]["alias_obs"] = True
        envs[p] = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
    g = torch.Generator(device="cuda").manual_seed(42)
    acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
    tau = (torch.rand(N, 33, generator=g, device="cuda") * 2 - 1) * 20
    for p in pipes:
        for i in range(60):
            envs[p].step(acts[i % 8])
    torch.cuda.synchronize()
    res = {p: [] for p in pipes}
    sim = {p: [] for p in pipes}
    for r in range(a.rounds):
        for p in pipes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.steps):
                envs[p].step(acts[i % 8])
            e1.record(); torch.cuda.synchronize()
            res[p].append(e0.elapsed_time(e1) / a.steps)
            if a.sim:
                e0.record()
                for i in range(a.steps):
                    envs[p].simulate(tau)
                e1.record(); torch.cuda.synchronize()
                sim[p].append(e0.elapsed_time(e1) / a.steps)
    for p in pipes:
        v = res[p]
        line = "N=%d pipeline %d  step ms: %s  best %.4f  (%.1f M env-steps/s)" % (N, p, " ".join("%.4f" % x for x in v), min(v), N / min(v) / 1e3)
        if a.sim:
            line += "   simulate ms: %s best %.4f" % (" ".join("%.4f" % x for x in sim[p]), min(sim[p]))
        print(line, flush=True)
    for p in pipes:
        envs[p].close()
