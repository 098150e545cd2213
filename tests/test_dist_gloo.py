"""N>1 path on CPU: world_size-2 (and 3) gloo jobs exercise the env sharding and the logging all-gather that
bench.py issues over RCCL on the 8-GPU node (the step itself has no collective: SURVEY.md section 8e)."""
import json
import os
import subprocess
import sys

import pytest

from isaacgymdyros_amd import dist as dwdist

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_partitions_exactly():
    for total in (7, 64, 16384 * 8, 131075):
        for world in (1, 2, 3, 8):
            spans = [dwdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_episode_stats_all_gather_gloo(tmp_path, world):
    port = 29700 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_dist_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, timeout=300, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    outs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    total = 10 * world + 3
    even = [e for e in range(total) if e % 2 == 0]
    for o in outs:
        assert o["world"] == world and o["stats_shape"] == [world, 4]
        s = o["summary"]                                  # identical on every rank
        assert s["envs_with_episode"] == len(even)
        assert s["episodes"] == 3 * len(even)
        assert s["mean_episode_return"] == pytest.approx(sum(2.0 * e for e in even) / len(even))
    # the optional global perturbation gate: the condition of tasks/dyros_dynamic_walk.py:489 on the means over ALL ranks' envs
    ns = [o["gate"]["n"] for o in outs]
    glen = sum(nr * (5000 + 1500 * r) for r, nr in enumerate(ns)) / sum(ns)
    gcrm = sum(nr * (0.15 + 0.03 * r) for r, nr in enumerate(ns)) / sum(ns)
    want = glen > 6000 and gcrm > 0.165
    alone = [(5000 + 1500 * r) > 6000 and (0.15 + 0.03 * r) > 0.165 for r in range(world)]
    assert any(a != want for a in alone), "the case must tell the global gate from some rank's own"
    for o in outs:
        g = o["gate"]
        assert g["openA"] == want and g["latchA"] == int(want), (g, glen, gcrm)          # every rank decides alike, on the global means
        assert not g["openB"] and g["latchB"] == 0 and g["latchC"] == 1
        assert s["mean_episode_length"] == pytest.approx(sum(e + 1.0 for e in even) / len(even))
    assert sorted((o["lo"], o["hi"]) for o in outs) == [dwdist.shard_range(total, r, world) for r in range(world)]
