#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of DyrosDynamicWalk, random-action rollout, flat ground.

`python bench.py --gpus N --steps K --warmup W` (N>1: launched by torch.distributed.run, one rank per GPU).
One "step" = one VecTask.step over this rank's environments = ONE launch of the fused step kernel (2 physics
substeps + task logic).  Inputs (a pool of U(-1,1) action batches, seed 42) are resident in HBM before the timed
region; the timed region holds exactly K steps between barrier+synchronize pairs; rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A_STEP_BYTES = 6816        # algorithmic bytes per env-step, SURVEY.md section 8(d): 1704 words
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HORIZON = 128              # rollout horizon of the reference's PPO config; logging gather once per horizon


def cpu_baseline(seconds_budget: float = 15.0):
    """The CPU oracle (C restatement, `kind: port`) on the host cores: bounded sample of the same workload."""
    import numpy as np
    from isaacgymdyros_amd.task_constants import load_task_constants
    from oracle.oracle import OracleSim
    N = 4096
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    sim = OracleSim(N, task_const=load_task_constants(), torch_gpu_div=1)
    sim.buf["dof_state"][:, :, 0] = load_task_constants()["initial_dof_pos"]
    rng = np.random.default_rng(42)
    acts = [rng.uniform(-1, 1, size=(N, 13)).astype(np.float32) for _ in range(8)]
    for t in range(3):
        sim.step(acts[t % 8], None, t)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds_budget:
        sim.step(acts[k % 8], None, 3 + k)
        k += 1
    dt = time.perf_counter() - t0
    out = {"value": N * k / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
           "sample": "%d envs x %d steps of the C oracle (oracle/dw_oracle.c, OpenMP over envs), %.1f s" % (N, k, dt)}
    # the same code on ONE core (SURVEY 8d asks for both ends), a 5 s sample of 128 envs
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        one = OracleSim(128, task_const=load_task_constants(), torch_gpu_div=1)
        one.buf["dof_state"][:, :, 0] = load_task_constants()["initial_dof_pos"]
        a1 = [a[:128].copy() for a in acts]
        one.step(a1[0], None, 0)
        t1, k1 = time.perf_counter(), 0
        while time.perf_counter() - t1 < 5.0:
            one.step(a1[k1 % 8], None, 1 + k1)
            k1 += 1
        out["single_core_value"] = 128 * k1 / (time.perf_counter() - t1)
        gomp.omp_set_num_threads(threads)
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs-per-gpu", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also-4096", action="store_true", help="skip the secondary 4096-env measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from isaacgymdyros_amd import dist as dwdist
    from isaacgymdyros_amd.config import default_cfg
    from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk

    rank, local_rank, world = dwdist.init_from_env("nccl")
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)

    def run(envs, steps, warmup):
        cfg = default_cfg(envs, dev)
        cfg["seed"] = 42 + rank
        env = DyrosDynamicWalk(cfg, dev, 0, True)
        g = torch.Generator(device=dev).manual_seed(42 + rank)
        pool = [torch.rand(envs, 13, generator=g, device=dev) * 2 - 1 for _ in range(64)]
        env.reset()
        for i in range(warmup):
            env.step(pool[i % 64])
        resets = torch.zeros((), device=dev, dtype=torch.int64)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(steps):
            _, _, done, _ = env.step(pool[i % 64])
            if (i + 1) % HORIZON == 0:
                stats = dwdist.gather_episode_stats(env._buf["env_state"])      # logging only, once per horizon
        e1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        kernel_ms = e0.elapsed_time(e1) / steps          # one kernel per step on this stream
        stats = dwdist.gather_episode_stats(env._buf["env_state"])
        resets = int(env.episodes_finished.sum())
        env.close()
        return wall, kernel_ms, dwdist.summarize(stats), resets

    wall, kernel_ms, epi, resets = run(args.envs_per_gpu, args.steps, args.warmup)
    t = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    total_envs = args.envs_per_gpu * world
    value = total_envs * args.steps / wall_max

    out = None
    if rank == 0:
        achieved = A_STEP_BYTES * args.envs_per_gpu / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(str(args.envs_per_gpu), {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node) DyrosDynamicWalk", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DyrosDynamicWalk random-action rollout, flat ground, mu=1, DR (mass/damping/armature) on, "
                                   "resets on, in-kernel RNG", "num_envs_per_gpu": args.envs_per_gpu,
                       "total_envs": total_envs, "substeps_per_step": 2, "dt": 0.002,
                       "parallelism": "env-sharded x%d, no collective in step; RCCL all-gather of episode stats every %d steps" % (world, HORIZON)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "dw_k_step", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": A_STEP_BYTES * args.envs_per_gpu},
            "episodes": dict(epi, finished_total=resets),
        }
    if not args.no_also_4096 and world == 1:
        w2, k2, _, _ = run(4096, args.steps, args.warmup)
        if rank == 0:
            out["num_envs_4096"] = {"value": 4096 * args.steps / w2, "ms_per_step": w2 / args.steps * 1e3, "kernel_ms": k2}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:          # the oracle is a checker; the bench line stands without it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": "oracle unavailable: %s" % e}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
