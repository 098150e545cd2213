"""Where the TocabiAMPLower step goes: step() and reset_done() timed separately (host clock around a device sync each).
usage: python tools/amp_time2.py [N] [--graph] [--fused] [--draws] [--noring]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)          # (A/B builds under isaacgymdyros_amd/_ab/, tools/tu_lib.sh)
from isaacgymdyros_amd.tocabi_amp_lower import TocabiAMPLower, default_amp_cfg
GRAPH = "--graph" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 16384
cfg = default_amp_cfg(N, "cuda:0")
cfg["sim"]["mi355"] = {"amp_fused": "--fused" in sys.argv, "amp_device_draws": "--draws" in sys.argv, "amp_hist_ring": "--noring" not in sys.argv}
if os.environ.get("DW_AMP_ONE"):          # the whole step as one launch (dw_amp_step)
    cfg["sim"]["mi355"]["amp_one_launch"] = True
env = TocabiAMPLower(cfg, "cuda:0", 0, True)
env.reset_done()
if GRAPH:
    env.enable_graph_step()
g = torch.Generator(device="cuda").manual_seed(1)
acts = [(torch.rand(N, 12, generator=g, device="cuda") * 2 - 1) * 0.3 for _ in range(8)]
for i in range(30):
    env.step(acts[i % 8]); env.reset_done()
ts = tr = 0.0; nres = 0; K = 200
for i in range(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(acts[i % 8])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    _, ids = env.reset_done()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts += t1 - t0; tr += t2 - t1; nres += len(ids)
# and back to back, as the bench leg runs them
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K):
    env.step(acts[i % 8]); env.reset_done()
torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / K
print("back to back: %.3f ms per step + reset_done = %.1f M env-steps/s" % (tb * 1e3, N / tb / 1e6))
print("N=%d%s%s: step %.3f ms, reset_done %.3f ms (%.1f resets per call)" % (N, " fused" if "--fused" in sys.argv else "", " graph" if GRAPH else "", ts / K * 1e3, tr / K * 1e3, nres / K))
env.close()
