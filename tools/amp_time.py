"""Step rate of the TocabiAMPLower host class (row f-3): dw_simulate x 2 + three HIP entry points + the torch
bookkeeping between them, reset_done() after every step as the AMP learner calls it.  usage: python tools/amp_time.py [N] [--graph]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
GRAPH = "--graph" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
for N in ([int(args[0])] if args else [4096, 16384]):
    env = TocabiAMPLower(default_amp_cfg(N, "cuda:0"), "cuda:0", 0, True)
    env.reset_done()
    if GRAPH:
        env.enable_graph_step()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = [(torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * 0.3 for _ in range(8)]
    for i in range(30):
        env.step(acts[i % 8]); env.reset_done()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 200
    for i in range(K):
        env.step(acts[i % 8]); env.reset_done()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("TocabiAMPLower N=%d%s: %.3f ms per step + reset_done, %.2f M env-steps/s" % (N, " (step in a hipGraph)" if GRAPH else "", dt * 1e3, N / dt / 1e6))
    env.close()
