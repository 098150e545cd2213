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
out = dict(rank=rank, world=world, lo=lo, hi=hi, stats_shape=list(stats.shape), summary=summ)
with open(os.path.join(sys.argv[1], "rank%d.json" % rank), "w") as f:
    json.dump(out, f)
dist.barrier()
dist.destroy_process_group()
