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
A_PHYS_BYTES = 1636        # algorithmic bytes per env-substep at the Gym boundary, SURVEY.md section 8(d): 409 words
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PPO_EPOCHS = 3            # timed epochs of the config-3 leg (after one untimed epoch)
HORIZON = 128              # rollout horizon of the reference's PPO config; logging gather once per horizon
WARM_SECONDS = 0.75        # untimed stepping (wall time) on the env that is about to be timed, before the W warm-up steps
VALU_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32 vector peak (packed v_pk_fma_f32: 64 flop / clk / SIMD)
# Plain (unpacked) fp32 vector instructions issue at one wave64 instruction per ~4.6 cycles per SIMD however many waves share it
# (tools/valu_issue.hip on the MI355X, profiles/r03g_valu_issue.txt: 2.941 / 4.945 / 9.125 ms for 1.28 M independent v_fma_f32 per
# wave at 1 / 2 / 4 waves per SIMD) -- half the packed peak.  Seconds per instruction per SIMD at two waves per SIMD:
VALU_ISSUE_S_2W = 4.945e-3 / (20000 * 64 * 2)
FLOPS_PER_ENV_STEP = 1.0e5 # useful flops of one env-step (2 substeps: ABA ~14 k + contact ~30 k each, task logic ~5 k), SURVEY 8(d)


def kernel_source_hash() -> str:
    """sha256 over the kernel sources and the flags they are built with (isaacgymdyros_amd/build.py): the PMC traffic record under
    profiles/ names the build it was measured on, and is quoted only for exactly that one."""
    import hashlib
    from isaacgymdyros_amd import build
    h = hashlib.sha256()
    h.update(repr((build.FLAGS, build.SOURCES)).encode())
    csrc = os.path.join(ROOT, "isaacgymdyros_amd", "csrc")
    for f in sorted(os.listdir(csrc)) + ["../../include/dyros_walk.h"]:
        p = os.path.join(csrc, f)
        if os.path.isfile(p) and f.endswith((".h", ".hip")) and not f.startswith(("dw_amp", "dw_ppo")):          # (dw_amp.*, dw_ppo.*: rows f-3 / f-2, not part of the step kernels)
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def effective_cpus():
    """Threads the host really gives this process: the affinity mask capped by the cgroup CPU quota (a 1-GPU slice of a
    256-thread host is 16 CPUs; 256 OpenMP threads on it were what made r02's all-core figure 7x one core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = "affinity mask: %d" % n
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            c = max(1, int(float(q) / float(p) + 0.5))
            if c < n:
                n, why = c, "cgroup cpu.max quota: %d of %d visible" % (c, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    if os.environ.get("OMP_NUM_THREADS"):
        n, why = int(os.environ["OMP_NUM_THREADS"]), "OMP_NUM_THREADS"
    return n, why


def cpu_baseline(seconds_budget: float = 15.0):
    """The CPU oracle (C restatement, `kind: port`) on the host cores: bounded sample of the same workload (16384 envs, so
    that every thread of a 256-thread host has 64 envs per step)."""
    import numpy as np
    from isaacgymdyros_amd.task_constants import load_task_constants
    from oracle.oracle import OracleSim
    N = 16384
    threads, why = effective_cpus()
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except Exception:
        pass
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
    out = {"value": N * k / dt, "unit": "env-steps/s", "cores": threads, "cores_from": why, "kind": "port",
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
        out["parallel_efficiency"] = out["value"] / (out["single_core_value"] * threads)
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
    ap.add_argument("--no-config5", action="store_true", help="skip the friction-DR + forced-pushes leg (BASELINE config 5)")
    ap.add_argument("--no-terrain", action="store_true", help="skip the height-field leg (SURVEY 8 row f-4, default curriculum map)")
    ap.add_argument("--warm-seconds", type=float, default=WARM_SECONDS, help="wall time of untimed stepping before each timed leg")
    ap.add_argument("--no-ppo", action="store_true", help="skip the PPO-consumer leg (BASELINE config 3)")
    ap.add_argument("--no-amp", action="store_true", help="skip the sibling-task leg (TocabiAMPLower, SURVEY 8 row f-3)")
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

    def run(envs, steps, warmup, alias_obs=False, mi=None, friction_dr=False, terrain=False, warm_seconds=None, fresh_actions=False):
        """K steps of VecTask.step on `envs` envs of this rank.  alias_obs = False is the product's default contract (step() returns
        a fresh observation tensor, as the reference's torch.clamp does); True returns the view of obs_buf, so that the stream
        holds nothing but the step kernel."""
        if plumbing:
            env = _PlumbingEnv(envs, rank)
            pool = [torch.zeros(envs, 13)]
        else:
            cfg = default_cfg(envs, dev)
            cfg["seed"] = 42 + rank
            cfg["sim"]["mi355"]["alias_obs"] = bool(alias_obs)
            cfg["sim"]["mi355"].update(mi or {})
            if friction_dr:
                from isaacgymdyros_amd.config import with_friction_randomization
                cfg = with_friction_randomization(cfg)
            if terrain:
                from isaacgymdyros_amd.config import with_terrain
                cfg = with_terrain(cfg, mesh_type="trimesh", curriculum=True)
            env = DyrosDynamicWalk(cfg, dev, 0, True)
            g = torch.Generator(device=dev).manual_seed(42 + rank)
            pool = [torch.rand(envs, 13, generator=g, device=dev) * 2 - 1 for _ in range(64)]
        env.reset()
        # Warm by WALL TIME, then the W warm-up steps the caller asked for; the timed region is exactly K steps.  A driver run with
        # --warmup 5 --steps 20 is a 3 ms timed region that is the first GPU work of a fresh process: rounds 3 and 4 read it 8 % and
        # 12 % under the same run's 256-step leg, and a warm-up counted in steps (one logging horizon = 20 ms of GPU work) did not
        # cure it -- the clocks of an idle GPU take longer than that to come up.  So: at least WARM_SECONDS of back-to-back stepping
        # on THIS env (synchronised in chunks, so that the host-side queue is as short when the timer starts as in steady state),
        # never less than one logging horizon.
        warm_seconds = args.warm_seconds if warm_seconds is None else warm_seconds
        # everything the timed loop calls must have run once before it: the logging gather's torch kernels are loaded on first use,
        # which on a fresh box costs ~15 ms of host time.  It runs BEFORE the warm-up stepping, not between it and the timer: the gather's
        # temporaries stir torch's caching allocator, and the fresh 32 MB observation tensor of each of the next steps (the reference's
        # contract) then comes from memory the device has not touched yet -- 20 steps right after a gather take 0.151 / 0.148 / 0.145 ms each
        # in three consecutive regions against 0.140 otherwise (tools/first_steps.py; the zero-copy contract does not show it).  This,
        # not clocks or the synchronize, was what made the driver's --steps 20 line of rounds 3 and 4 read 8 - 12 % under its long leg.
        dwdist.gather_episode_stats(env._buf["env_state"])
        t_w, n_w = time.perf_counter(), 0
        while n_w < HORIZON or time.perf_counter() - t_w < warm_seconds:
            for i in range(32):
                env.step(pool[(n_w + i) % len(pool)])
            n_w += 32
            sync()
        if not plumbing:
            # (both events exist and have been recorded once before the timer starts: creating / first recording one loads code)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
        if world > 1:
            dist.barrier()
        # The W warm-up steps come last, right before the synchronize that opens the timed region.
        for i in range(max(warmup, 8)):
            env.step(pool[i % len(pool)])
        sync()
        t0 = time.perf_counter()
        if not plumbing:
            e0.record()
        for i in range(steps):
            if fresh_actions:          # (SURVEY 8d's wording: the batch is regenerated on the device every step -- one torch kernel more on the stream)
                env.step(torch.empty(envs, 13, device=dev).uniform_(-1.0, 1.0, generator=g))
            else:
                env.step(pool[i % len(pool)])
            if (i + 1) % HORIZON == 0:
                dwdist.gather_episode_stats(env._buf["env_state"])      # logging only, once per horizon
        if not plumbing:
            e1.record()
        sync()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        # device time per step on the launch stream (HIP events on torch's current stream, which is the stream step() launches
        # on): the step's kernels, launch gaps included
        kernel_ms = (e0.elapsed_time(e1) / steps) if not plumbing else wall / steps * 1e3
        stats = dwdist.gather_episode_stats(env._buf["env_state"])
        resets = int(env.episodes_finished.sum())
        kinfo = env.kernel_info() if hasattr(env, "kernel_info") else {}
        # the Gym-boundary substep (dw_simulate: the ABA + contact step the north star names), same envs, same stream
        sim_ms = None
        if not plumbing and alias_obs:
            tau = (torch.rand(envs, 33, device=dev) * 2 - 1) * 20
            for _ in range(20):
                env.simulate(tau)
            sync()
            e0.record()
            for _ in range(200):
                env.simulate(tau)
            e1.record()
            sync()
            sim_ms = e0.elapsed_time(e1) / 200
        pert = float(env._buf["stacked_rewards"][:, 14].mean()) if not plumbing else 0.0
        env.close()
        return {"wall": wall, "kernel_ms": kernel_ms, "episodes": dwdist.summarize(stats), "resets": resets, "kinfo": kinfo,
                "sim_ms": sim_ms, "perturb_start_fraction": pert}

    def reduce_max(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    head = run(args.envs_per_gpu, args.steps, args.warmup)                 # the headline: the product's default obs contract
    wall_max = reduce_max(head["wall"])
    total_envs = args.envs_per_gpu * n_joined
    value = total_envs * args.steps / wall_max
    # a short driver run (--steps 20 is 12 ms) says little by itself: time a longer region as well and report both
    long_run = None
    if args.steps < 200:
        r = run(args.envs_per_gpu, 256, 16)
        w_l = reduce_max(r["wall"])
        long_run = {"steps": 256, "value": total_envs * 256 / w_l, "ms_per_step": w_l / 256 * 1e3}
    # the step kernel alone on the stream (zero-copy observation view): its mean launch duration is the roofline's denominator
    alias = run(args.envs_per_gpu, max(args.steps, 256), args.warmup, alias_obs=True) if not plumbing else head
    w_a = reduce_max(alias["wall"])
    kernel_ms = alias["kernel_ms"]
    n_a = max(args.steps, 256) if not plumbing else args.steps

    out = None
    if rank == 0:
        kinfo = head["kinfo"]
        achieved = A_STEP_BYTES * args.envs_per_gpu / (kernel_ms * 1e-3) / 1e9
        traffic, lane_slots, traffic_note = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(str(args.envs_per_gpu), {})
                if rec.get("kernel") == kinfo.get("kernels") and rec.get("kernel_source_hash") == kernel_source_hash():
                    traffic = rec.get("hbm_bytes_per_launch")
                    lane_slots = rec.get("lane_slots_per_env_step")
                else:
                    traffic_note = "profiles/pmc_traffic.json was measured on kernel %s at sources %s; this build is %s at %s" % (
                        rec.get("kernel"), rec.get("kernel_source_hash"), kinfo.get("kernels"), kernel_source_hash())
            except Exception as e:
                traffic_note = "profiles/pmc_traffic.json unreadable: %s" % e
        useful_flops = FLOPS_PER_ENV_STEP * args.envs_per_gpu
        valu = useful_flops / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "env-steps/sec (whole node) DyrosDynamicWalk", "value": value, "unit": "env-steps/s",
            "n_gpus": n_joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not plumbing else "plumbing-test (no kernel ran)",
            "config": {"workload": "DyrosDynamicWalk random-action rollout, flat ground, mu=1, DR (mass/damping/armature) on, "
                                   "resets on, in-kernel RNG, step() returns a fresh observation tensor (reference contract)",
                       "num_envs_per_gpu": args.envs_per_gpu,
                       "total_envs": total_envs, "substeps_per_step": 2, "dt": 0.002,
                       "parallelism": "env-sharded x%d, no collective in step; RCCL all-gather of episode stats every %d steps" % (n_joined, HORIZON)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kinfo.get("kernels", "dw_k_step"), "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": A_STEP_BYTES * args.envs_per_gpu,
                         "kernel_source_hash": kernel_source_hash()},
            # the second roof: useful flops of the step (per-phase count in DESIGN.md section 6) against
            # the fp32 vector peak; lane_slots_per_env_step comes from the SQ_INSTS_VALU PMC pass when one is committed
            "roofline_valu": {"bound": "valu_f32", "achieved": valu, "peak": VALU_PEAK_TF, "unit": "TFLOP/s",
                              "frac": valu / VALU_PEAK_TF, "useful_flops_per_env_step": FLOPS_PER_ENV_STEP,
                              "lane_slots_per_env_step": lane_slots},
            "alias_obs": {"value": total_envs * n_a / w_a, "ms_per_step": w_a / n_a * 1e3,
                          "note": "cfg sim.mi355.alias_obs: step() returns the view of obs_buf (no copy kernel on the stream)"},
            "episodes": dict(head["episodes"], finished_total=head["resets"]),
            "head_leg_device_ms_per_step": head["kernel_ms"],          # HIP events around the same K steps (no host latency at either end)
        }
        if lane_slots and torch.cuda.is_available():
            # what binds the kernel at this size: the SIMD's vector-instruction issue.  Floor = the kernel's own VALU instruction
            # count (PMC) at the measured issue rate of plain fp32 instructions, spread evenly over the device's SIMDs
            simds = 4 * torch.cuda.get_device_properties(0).multi_processor_count
            floor_ms = lane_slots / 64.0 * args.envs_per_gpu / simds * VALU_ISSUE_S_2W * 1e3
            out["roofline_valu"]["issue"] = {"valu_instructions_per_simd": lane_slots / 64.0 * args.envs_per_gpu / simds,
                                             "ns_per_instruction_per_simd": VALU_ISSUE_S_2W * 1e9, "issue_floor_ms": floor_ms,
                                             "frac_of_issue_floor": floor_ms / kernel_ms,
                                             "note": "plain v_fma_f32 rate at 2 waves per SIMD, tools/valu_issue.hip"}
        if traffic_note:
            out["roofline"]["traffic_note"] = traffic_note
        if alias.get("sim_ms"):
            # the roofline the north star names: algorithmic bytes of ONE ABA + contact substep at the Gym boundary over the mean
            # duration of dw_k_simulate_oct (HIP events around 200 dw_simulate launches)
            a_p = A_PHYS_BYTES * args.envs_per_gpu / (alias["sim_ms"] * 1e-3) / 1e9
            out["roofline_phys"] = {"bound": "hbm", "achieved": a_p, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_p / HBM_PEAK_GBS,
                                    "kernel": kinfo.get("kernels", "dw_k_step").replace("step", "simulate"), "kernel_ms": alias["sim_ms"],
                                    "algorithmic_bytes_per_launch": A_PHYS_BYTES * args.envs_per_gpu,
                                    "env_substeps_per_s": args.envs_per_gpu / (alias["sim_ms"] * 1e-3)}
        if plumbing:
            out["valid"] = False
        if long_run:
            out["long_run"] = long_run
    if world == 1 and not plumbing:
        n2 = max(args.steps, 256)
        if not args.no_also_4096:               # BASELINE config 2
            r2 = run(4096, n2, args.warmup)
            a2 = run(4096, n2, args.warmup, alias_obs=True)
            ach2 = A_STEP_BYTES * 4096 / (a2["kernel_ms"] * 1e-3) / 1e9
            out["num_envs_4096"] = {"value": 4096 * n2 / r2["wall"], "ms_per_step": r2["wall"] / n2 * 1e3, "kernel_ms": a2["kernel_ms"],
                                    "roofline_frac": ach2 / HBM_PEAK_GBS}
        if not args.no_config5:                 #  the head leg with its actions regenerated every step instead of taken from a resident pool
            nf = max(args.steps, 256)
            rf = run(args.envs_per_gpu, nf, args.warmup, fresh_actions=True)
            out["fresh_actions_each_step"] = {"value": args.envs_per_gpu * nf / rf["wall"], "ms_per_step": rf["wall"] / nf * 1e3,
                                              "note": "as the head leg, but every step's action batch is drawn on the device inside the timed region (one torch uniform_ kernel per step)"}
        if not args.no_config5:                 # BASELINE config 5: friction DR next to mass / damping / armature, pushes forced on
            r5 = run(args.envs_per_gpu, n2, args.warmup, mi={"force_perturb_start": True}, friction_dr=True)
            out["config5_dr_friction_pushes"] = {"value": args.envs_per_gpu * n2 / r5["wall"], "ms_per_step": r5["wall"] / n2 * 1e3,
                                                 "perturb_start_fraction": r5["perturb_start_fraction"], "episodes_finished": r5["resets"],
                                                 "note": "friction x U(0.7,1.3) per env at reset, force_perturb_start (tasks/dyros_dynamic_walk.py:491)"}
        if not args.no_terrain:                 # the same step on the reference's default 10 x 20 curriculum map (height-field kernels)
            rt = run(args.envs_per_gpu, n2, args.warmup, alias_obs=True, terrain=True)
            out["terrain_curriculum"] = {"value": args.envs_per_gpu * n2 / rt["wall"], "ms_per_step": rt["wall"] / n2 * 1e3, "kernel_ms": rt["kernel_ms"],
                                         "ratio_to_flat_kernel": (rt["kernel_ms"] / kernel_ms) if kernel_ms and kernel_ms > 0 else None,
                                         "note": "cfg/terrain/terrain_cfg.py defaults (trimesh, curriculum), alias_obs; dw_k_step_oct<true, 2>"}
        if not args.no_ppo:                     # BASELINE config 3: the DYROS PPO loop attached (examples/ppo_consumer.py)
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
                ppo = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(ppo)
                def epochs_summary(keys=("step_fps", "play_fps", "total_fps"), **kw):
                    """One warm-up epoch, then EPOCHS timed ones: the median of each rate with its range (a single epoch on a
                    fresh box differed 17.2 M vs 32.9 M play_fps between two boxes in round 3)."""
                    sts = ppo.train(args.envs_per_gpu, epochs=1 + PPO_EPOCHS, device=dev, log=lambda s_: None, **kw)
                    keep = sts[1:]
                    rec = {}
                    for k in keys:
                        v = sorted(s_[k] for s_ in keep)
                        rec[k] = v[len(v) // 2]
                        rec[k + "_range"] = [v[0], v[-1]]
                    rec["first_epoch"] = {k: sts[0][k] for k in keys}
                    rec["mean_reward"] = keep[-1]["mean_reward"]
                    rec["epochs_timed"] = len(keep)
                    return rec
                out["config3_ppo"] = epochs_summary()
                out["config3_ppo"]["note"] = ("per epoch: horizon 128 rollout with the policy in the loop, then the DYROS PPO update; one untimed "
                                              "epoch first, then the median [min, max] over the timed epochs")
                # the same epochs with the rollout step captured in a hipGraph (dw_step_dev: step counter in device memory)
                out["config3_ppo"]["graph_rollout"] = epochs_summary(keys=("play_fps", "total_fps"), graph_rollout=True, graph_update=True)          # (a replayed step has no host-side env-step clock)
                out["config3_ppo"]["graph_rollout"]["note"] = "rollout step and minibatch update each captured in a hipGraph (fused capturable Adam)"
                # the reference's yaml sizes its minibatch for 4096 envs (cfg/train/DyrosDynamicWalkPPO.yaml:86, "u may play with this"): 128
                # minibatches x 5 mini-epochs = 640 updates per epoch.  With 16384 envs the same 4096-sample minibatch is 2 560 updates of a
                # launch-bound size; scaled with the env count (the same 640 updates per epoch) the update is not the epoch any more
                import copy
                scaled = copy.deepcopy(ppo.TRAIN_CFG)
                scaled["config"]["minibatch_size"] = int(scaled["config"]["minibatch_size"]) * max(1, args.envs_per_gpu // 4096)
                out["config3_ppo"]["graph_rollout_minibatch_scaled"] = epochs_summary(keys=("play_fps", "total_fps"), graph_rollout=True, graph_update=True, cfg=scaled)
                out["config3_ppo"]["graph_rollout_minibatch_scaled"]["note"] = ("as graph_rollout with minibatch_size x (envs / 4096) = %d: the reference's 640 updates per epoch"
                                                                                  % scaled["config"]["minibatch_size"])
                # the update as 4 launches: forward + loss + input gradients and the weight gradients on the matrix cores, then gradient
                # statistics, Adam and the scaler (include/dyros_ppo.h, isaacgymdyros_amd/ppo_update.py); the yaml's own minibatch of 4096
                rec = epochs_summary(keys=("play_fps", "total_fps"), graph_rollout=True, fused_update=True)
                rec["note"] = ("rollout step and the FUSED minibatch update (include/dyros_ppo.h: 4 launches on the matrix cores instead of ~190), 16 updates and 8 rollout steps per graph, each captured "
                               "in a hipGraph; minibatch_size as in the yaml")
                out["config3_ppo"]["fused_update"] = rec
                # (the legs above are kept for the comparison; this is the path a user of config 3 runs)
                legs = {k: v["total_fps"] for k, v in out["config3_ppo"].items() if isinstance(v, dict) and "total_fps" in v and k != "first_epoch"}
                best = max(legs, key=legs.get)
                out["config3_ppo"]["best"] = {"path": best, "total_fps": legs[best], "mean_reward": out["config3_ppo"][best]["mean_reward"]}
            except Exception as e:
                out["config3_ppo"] = dict(out.get("config3_ppo") or {}, error=str(e))
        if not args.no_amp and not plumbing:    # the sibling task on the same physics, step() + reset_done() as the AMP learner calls them
            try:
                from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
                acfg = default_amp_cfg(args.envs_per_gpu, dev)
                acfg["sim"]["mi355"] = {"amp_fused": True, "amp_device_draws": True}
                aenv = TocabiAMPLower(acfg, dev, 0, True)
                aenv.reset_done()
                # (no enable_graph_step(): with the draws made in the kernels a step is five launches and nothing else, and five plain launches
                #  run closer together than five nodes of a replayed hipGraph -- 0.184 against 0.192 ms, profiles/r06_amp_eager_vs_graph.txt)
                ag = torch.Generator(device=dev).manual_seed(1)
                aact = [(torch.rand(args.envs_per_gpu, 12, generator=ag, device=dev) * 2 - 1) * 0.3 for _ in range(8)]
                for i in range(30):
                    aenv.step(aact[i % 8]); aenv.reset_done()
                sync(); t0 = time.perf_counter()
                ka, nres = 200, 0
                for i in range(ka):
                    aenv.step(aact[i % 8])
                    nres += len(aenv.reset_done()[1])
                sync(); wa = time.perf_counter() - t0
                aenv.close()
                out["amp_lower"] = {"value": args.envs_per_gpu * ka / wa, "unit": "env-steps/s", "ms_per_step": wa / ka * 1e3, "resets_per_step": nres / ka,
                                    "note": "TocabiAMPLower (tasks/amp/tocabi_amp_lower_base.py + tasks/tocabi_amp_lower.py) on dw_simulate: step() with the "
                                            "bookkeeping in three HIP kernels around the two dw_simulate launches (ring histories, draws made in the kernels): five plain "
                                            "launches, + reset_done() every step as two launches (ids by dw_amp_reset_ids, then the reset; returns the ids: one host sync per step)"}
            except Exception as e:
                out["amp_lower"] = {"error": str(e)}
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
