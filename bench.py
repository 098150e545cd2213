#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of DyrosDynamicWalk, random-action rollout, flat ground.

`python bench.py --gpus N --steps K --warmup W`.  N>1: one rank per GPU -- either launched by torch.distributed.run
(the env carries WORLD_SIZE) or, when run bare, by this script itself, which starts the N ranks as child processes.
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
VALU_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32 vector peak
FLOPS_PER_ENV_STEP = 1.0e5 # useful flops of one env-step (2 substeps: ABA ~14 k + contact ~30 k each, task logic ~5 k), SURVEY 8(d)


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


def _launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves as fresh child processes
    (the reference maps rank -> device itself, utils/rlgames_utils.py:71-81).  This parent never imports torch.cuda nor
    touches the GPU, so nothing is re-exec'ed from a process that has initialised HIP; rank 0's JSON line is relayed."""
    import subprocess
    # --standalone: torchrun's own c10d rendezvous picks the port (no bind-close-reuse race on a port chosen here)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print("bench.py: the %d-rank job failed (rc %d)%s" % (args.gpus, proc.returncode, "" if line else ", no JSON line"),
              file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


class _PlumbingEnv:
    """--backend gloo: the N-rank plumbing of this script (launcher, env sharding, barriers, logging all-gather, MAX
    reduction, JSON) on CPU tensors with a no-op step.  Its line is marked invalid; it exists for tests/test_bench_launcher.py."""

    def __init__(self, envs, rank):
        import torch
        from isaacgymdyros_amd import abi
        self._buf = {"env_state": torch.zeros(envs, abi.K["DW_ES_WORDS"])}
        abi.es_view(self._buf["env_state"], "episodes_finished")[:] = 1
        abi.es_view(self._buf["env_state"], "epi_len_log")[:] = float(rank + 1)
        self.episodes_finished = abi.es_view(self._buf["env_state"], "episodes_finished")

    def reset(self):
        pass

    def step(self, a):
        time.sleep(1e-3)
        return None, None, None, None

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs-per-gpu", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also-4096", action="store_true", help="skip the secondary 4096-env measurement")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = CPU plumbing rehearsal of the N-rank path (no kernel runs; the line is marked invalid)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(_launch_ranks(args, sys.argv[1:]))
    if env_world is not None and "--gpus" not in " ".join(sys.argv[1:]):
        args.gpus = int(env_world)          # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's world counts
    if env_world is not None and int(env_world) != args.gpus:
        print("bench.py: --gpus %d contradicts WORLD_SIZE=%s" % (args.gpus, env_world), file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from isaacgymdyros_amd import dist as dwdist

    plumbing = args.backend == "gloo"
    rank, local_rank, world = dwdist.init_from_env(args.backend)
    if plumbing:
        dev = "cpu"
    else:
        from isaacgymdyros_amd.config import default_cfg
        from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
        dev = "cuda:%d" % local_rank
        torch.cuda.set_device(local_rank)
    # ranks that actually joined (not the flag): every rank contributes a 1
    joined = torch.ones((), device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(joined)
    n_joined = int(joined.item())

    def sync():
        if not plumbing:
            torch.cuda.synchronize()

    def run(envs, steps, warmup):
        if plumbing:
            env = _PlumbingEnv(envs, rank)
            pool = [torch.zeros(envs, 13)]
        else:
            cfg = default_cfg(envs, dev)
            cfg["seed"] = 42 + rank
            cfg["sim"]["mi355"]["alias_obs"] = True       # zero-copy obs: the bench keeps nothing across steps
            env = DyrosDynamicWalk(cfg, dev, 0, True)
            g = torch.Generator(device=dev).manual_seed(42 + rank)
            pool = [torch.rand(envs, 13, generator=g, device=dev) * 2 - 1 for _ in range(64)]
        env.reset()
        for i in range(warmup):
            env.step(pool[i % len(pool)])
        # everything the timed loop calls must have run once before it: the logging gather's torch kernels are loaded on
        # first use, which on a fresh box costs ~15 ms of host time (measured r02: 0.285 instead of 0.257 ms/step over 512
        # steps whenever --warmup was shorter than one logging horizon)
        dwdist.gather_episode_stats(env._buf["env_state"])
        if world > 1:
            dist.barrier()
        sync()
        if not plumbing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if not plumbing:
            e0.record()
        for i in range(steps):
            env.step(pool[i % len(pool)])
            if (i + 1) % HORIZON == 0:
                dwdist.gather_episode_stats(env._buf["env_state"])      # logging only, once per horizon
        if not plumbing:
            e1.record()
        sync()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        # device time per step on the launch stream (HIP events): the step's kernels, launch gaps included
        kernel_ms = (e0.elapsed_time(e1) / steps) if not plumbing else wall / steps * 1e3
        stats = dwdist.gather_episode_stats(env._buf["env_state"])
        resets = int(env.episodes_finished.sum())
        kinfo = env.kernel_info() if hasattr(env, "kernel_info") else {}
        env.close()
        return wall, kernel_ms, dwdist.summarize(stats), resets, kinfo

    def reduce_max(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    wall, kernel_ms, epi, resets, kinfo = run(args.envs_per_gpu, args.steps, args.warmup)
    wall_max = reduce_max(wall)
    total_envs = args.envs_per_gpu * n_joined
    value = total_envs * args.steps / wall_max
    # a short driver run (--steps 20 is 12 ms) says little by itself: time a longer region as well and report both
    long_run = None
    if args.steps < 200 and not plumbing:
        w_l, k_l, _, _, _ = run(args.envs_per_gpu, 256, 16)
        w_l = reduce_max(w_l)
        long_run = {"steps": 256, "value": total_envs * 256 / w_l, "ms_per_step": w_l / 256 * 1e3, "kernel_ms": k_l}

    out = None
    if rank == 0:
        achieved = A_STEP_BYTES * args.envs_per_gpu / (kernel_ms * 1e-3) / 1e9
        traffic, lane_slots = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(str(args.envs_per_gpu), {})
                if rec.get("kernel", kinfo.get("kernels")) == kinfo.get("kernels"):      # counters of THIS kernel only
                    traffic = rec.get("hbm_bytes_per_launch")
                    lane_slots = rec.get("lane_slots_per_env_step")
            except Exception:
                traffic, lane_slots = None, None
        useful_flops = FLOPS_PER_ENV_STEP * args.envs_per_gpu
        valu = useful_flops / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "env-steps/sec (whole node) DyrosDynamicWalk", "value": value, "unit": "env-steps/s",
            "n_gpus": n_joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not plumbing else "plumbing-test (no kernel ran)",
            "config": {"workload": "DyrosDynamicWalk random-action rollout, flat ground, mu=1, DR (mass/damping/armature) on, "
                                   "resets on, in-kernel RNG", "num_envs_per_gpu": args.envs_per_gpu,
                       "total_envs": total_envs, "substeps_per_step": 2, "dt": 0.002,
                       "parallelism": "env-sharded x%d, no collective in step; RCCL all-gather of episode stats every %d steps" % (n_joined, HORIZON)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kinfo.get("kernels", "dw_k_step"), "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": A_STEP_BYTES * args.envs_per_gpu},
            # the second roof: useful flops of the step (per-phase count in DESIGN.md section 6) against
            # the fp32 vector peak; lane_slots_per_env_step comes from the SQ_INSTS_VALU PMC pass when one is committed
            "roofline_valu": {"bound": "valu_f32", "achieved": valu, "peak": VALU_PEAK_TF, "unit": "TFLOP/s",
                              "frac": valu / VALU_PEAK_TF, "useful_flops_per_env_step": FLOPS_PER_ENV_STEP,
                              "lane_slots_per_env_step": lane_slots},
            "episodes": dict(epi, finished_total=resets),
        }
        if plumbing:
            out["valid"] = False
        if long_run:
            out["long_run"] = long_run
    if not args.no_also_4096 and world == 1 and not plumbing:
        w2, k2, _, _, _ = run(4096, max(args.steps, 256), args.warmup)
        if rank == 0:
            n2 = max(args.steps, 256)
            out["num_envs_4096"] = {"value": 4096 * n2 / w2, "ms_per_step": w2 / n2 * 1e3, "kernel_ms": k2}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not plumbing:
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
