"""Is an RCCL all-reduce accepted inside a hipGraph capture on this stack?  One rank on one GPU (backend nccl = RCCL): capture
kernel | all_reduce | kernel in a torch.cuda.CUDAGraph on a side stream, replay it three times, check the data.  Prints one line:
`graph_collective: captured ...` or `graph_collective: refused ...` (what examples/ppo_consumer.py's DW_PPO_GRAPH_COLLECTIVE=1 relies on).
usage: python tools/probe_graph_collective.py"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29791")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.ones(402464, device="cuda:0")
dist.all_reduce(x); torch.cuda.synchronize()          # (communicator set up outside the capture)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        x.mul_(2.0)
        dist.all_reduce(x)
        x.add_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    want = 1.0
    for _ in range(4):          # (capture itself does not execute; torch's capture warm-up does not run the body either)
        pass
    v = float(x[0])
    print("graph_collective: captured and replayed 3 x; x[0] = %.1f (1 -> 2x + 1 three times = 15.0 expected)" % v)
except Exception as e:          # noqa: BLE001
    print("graph_collective: refused -- %s: %s" % (type(e).__name__, str(e).split("\n")[0][:300]))
dist.destroy_process_group()
