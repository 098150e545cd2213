"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job on CPU tensors."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgymdyros_amd import abi, dist as dwdist  # noqa: E402

rank, local_rank, world = dwdist.init_from_env("gloo")
total = 10 * world + 3                     # uneven split on purpose
lo, hi = dwdist.shard_range(total, rank, world)
n = hi - lo
es = torch.zeros(n, abi.K["DW_ES_WORDS"])
ids = torch.arange(lo, hi, dtype=torch.float32)
abi.es_view(es, "last_episode_return")[:] = ids * 2.0          # env e finished an episode with return 2e
abi.es_view(es, "epi_len_log")[:] = ids + 1.0
abi.es_view(es, "episodes_finished")[:] = (torch.arange(lo, hi) % 2 == 0).int() * 3   # even envs finished 3 episodes
stats = dwdist.gather_episode_stats(es)
summ = dwdist.summarize(stats)

# the optional global perturbation gate (dist.sync_perturbation_gate): rank r's newest slot holds the sums of n envs whose mean episode
# length is 5000 + 1500 r and whose mean contact reward is 0.15 + 0.03 r, spread over the 32 buckets.  Case A: steps_done = 5 -> slot 1;
# case B: slot 2 with every rank below both thresholds
def gate_with(steps_done, mean_len, mean_crm):
    g = torch.zeros(abi.K["DW_GATE_WORDS"], dtype=torch.int64)
    slot = (steps_done - 1) % 3
    nb = abi.K["DW_GATE_BUCKETS"]
    for e in range(n):
        g[(slot * nb + e % nb) * 2] += int(mean_len)
        g[(slot * nb + e % nb) * 2 + 1] += int(round(mean_crm * 4294967296.0))
    return g
gA = gate_with(5, 5000 + 1500 * rank, 0.15 + 0.03 * rank)
openA = bool(dwdist.sync_perturbation_gate(gA, 5, n))
gB = gate_with(6, 3000 + 100 * rank, 0.10)
openB = bool(dwdist.sync_perturbation_gate(gB, 6, n))
gC = gate_with(6, 3000, 0.10); gC[abi.K["DW_GATE_LATCH"]] = 1          # an already latched gate stays latched
dwdist.sync_perturbation_gate(gC, 6, n)
gate = dict(openA=openA, latchA=int(gA[abi.K["DW_GATE_LATCH"]]), openB=openB, latchB=int(gB[abi.K["DW_GATE_LATCH"]]), latchC=int(gC[abi.K["DW_GATE_LATCH"]]), n=n)
out = dict(rank=rank, world=world, lo=lo, hi=hi, stats_shape=list(stats.shape), summary=summ, gate=gate)
with open(os.path.join(sys.argv[1], "rank%d.json" % rank), "w") as f:
    json.dump(out, f)
dist.barrier()
dist.destroy_process_group()
