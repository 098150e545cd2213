"""One FusedPpoUpdate per tail form at config 3's minibatch: time per update (events around K updates, plain launches and a replayed graph) and the
merged launch's barrier words.  usage: python tools/ppo_tail_probe.py [B] [nmb]"""
import copy, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)          # (A/B builds under isaacgymdyros_amd/_ab/, tools/tu_lib.sh)
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
from isaacgymdyros_amd import ppo_update as U
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nmb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = "cuda:0"
torch.manual_seed(0)
net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
n = B * nmb
obs = torch.randn(n, U.IN, generator=g, device=dev)
act, mu = torch.randn(n, U.ACT, generator=g, device=dev), torch.randn(n, U.ACT, generator=g, device=dev)
nlp, adv, ret = (torch.randn(n, generator=g, device=dev) for _ in range(3))
for name, merged in (("two launches", False), ("one launch", True), ("two launches", False), ("one launch", True)):
    f = U.FusedPpoUpdate(copy.deepcopy(net), dict(ppo.TRAIN_CFG["config"]), B, nmb, dev, rowmajor=False, merged_tail=merged)
    f.bind_batch(obs, act, nlp, mu, adv, ret)
    for _ in range(3):
        f.update()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 64
    e0.record()
    for _ in range(K):
        f.update()
    e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) / K
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(nmb):
            f.update()
    gr.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    graph = e0.elapsed_time(e1) / (4 * nmb)
    w = f.part.view(torch.int32)[641:642].tolist()
    print("%-14s B=%d  per update: %.4f ms launched, %.4f ms in a replayed graph of %d; barrier words %s timed_out=%s skipped=%s"
          % (name, B, plain, graph, nmb, w, f.part[642].item(), f.state[U.K["DWP_S_OUT"] + 7].item()), flush=True)
