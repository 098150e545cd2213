"""`python bench.py --gpus N` must start N real ranks by itself (VERDICT r1 item 1b / ADVICE): a parent that never
touches the GPU spawns torch.distributed.run children and relays rank 0's JSON line.  Rehearsed here on CPU with
--backend gloo: same launcher, sharding arithmetic, barriers, logging all-gather, MAX reduction and JSON merge; the
kernel is replaced by a no-op step and the line is marked invalid."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300, steps=6):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--steps", str(steps), "--warmup", "2",
                           "--envs-per-gpu", "16", "--no-cpu-baseline", "--no-also-4096"] + extra,
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bare_invocation_spawns_two_ranks_and_relays_one_line():
    p = _run(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2                        # ranks that joined, not the flag
    assert out["config"]["total_envs"] == 32 and out["config"]["num_envs_per_gpu"] == 16
    assert out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["valid"] is False and "plumbing" in out["data"]
    assert out["episodes"]["envs_with_episode"] == 32          # the all-gather saw both ranks' blocks
    assert abs(out["episodes"]["mean_episode_length"] - 1.5) < 1e-9     # rank 0 wrote 1.0, rank 1 wrote 2.0
    assert out["value"] > 0 and abs(out["value"] - 32 * 6 / (out["ms_per_step"] * 6e-3)) / out["value"] < 1e-6
    # the contract's two roofline objects are always there (numbers mean nothing on the plumbing env)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"} <= set(out["roofline"])
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["peak"] == 8000.0 and out["roofline"]["unit"] == "GB/s"
    assert {"bound", "achieved", "peak", "frac", "useful_flops_per_env_step", "lane_slots_per_env_step"} <= set(out["roofline_valu"])
    assert out["metric"].startswith("env-steps/sec") and out["unit"] == "env-steps/s" and out["dtype"] == "f32"
    assert out["higher_is_better"] is True and out["vs_baseline"] is None


def test_single_rank_needs_no_launcher():
    p = _run(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_contradicting_world_size_is_an_error():
    p = _run(["--gpus", "4"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "contradicts WORLD_SIZE" in p.stderr


def test_short_head_leg_reads_like_the_long_leg():
    """A driver-style short run (--steps well under 200) times a second, 256-step region in the same process; the two must
    tell the same ms/step (the 20-step headline read 12 % under its own long leg because the warm-up was
    counted in steps).  On the plumbing env a step is a 1 ms sleep, so this checks the SHAPE of run(): warm by wall time,
    events / gather exercised before the timer, exactly K steps inside it (5 % here: a sleep is a noisy clock; on the GPU the two legs
    read within 2 %, gpurun_out/r5t_b.json)."""
    p = _run(["--gpus", "1"], steps=100)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["steps"] == 100 and out["long_run"]["steps"] == 256
    assert abs(out["ms_per_step"] - out["long_run"]["ms_per_step"]) / out["long_run"]["ms_per_step"] < 0.05, (out["ms_per_step"], out["long_run"])
