"""Cycle stamps of dwp_mlp's workgroup 0 / wave 0 (library built with -DDWP_STAMPS: tools/tu_lib.sh ppostamps dw_ppo.hip -DDWP_STAMPS; DW_LIB=...)."""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd import ppo_update as U
spec = importlib.util.spec_from_file_location("ppo_consumer", os.path.join(ROOT, "examples", "ppo_consumer.py"))
ppo = importlib.util.module_from_spec(spec); spec.loader.exec_module(ppo)
c = dict(ppo.TRAIN_CFG["config"]); dev = "cuda:0"
net = ppo.DyrosActorCritic(U.IN, U.ACT, ppo.TRAIN_CFG["network"]).to(dev)
B, nmb = 4096, 2
f = U.FusedPpoUpdate(net, c, B, nmb, dev, rowmajor=False)
f.set_learning_rates(1e-5, 1e-5)
g = torch.Generator(device=dev).manual_seed(1)
n = B * nmb
obs = torch.randn(n, U.IN, generator=g, device=dev)
if os.environ.get("OBS16", "1") != "0":          # (the trainer's form: fp16 rows of 512)
    o16 = torch.zeros(n, U.INP, device=dev, dtype=torch.float16); o16[:, :U.IN] = obs.half(); obs = o16
f.bind_batch(obs, torch.randn(n, U.ACT, generator=g, device=dev) * 0.1, torch.zeros(n, device=dev), torch.zeros(n, U.ACT, device=dev),
             torch.randn(n, generator=g, device=dev), torch.randn(n, generator=g, device=dev))
names = ["ring W1", "staging", "layer 1 products", "epilogue 1 (+ring W2)", "layer 2 products", "epilogue 2, loss inputs", "head products", "loss", "second-layer gradient product", "mask 2", "first-layer gradient products", "mask 1 + end"]
acc = None
for it in range(12):
    f.update(); torch.cuda.synchronize()
    if it >= 4:
        st = f.pbuf[0, 0, 533:544].cpu().numpy().astype("int64")          # (PB_ST + 5 ..: stamps 0 .. 10)
        d = [(int(st[i + 1]) - int(st[i])) & 0xffffff for i in range(10)]
        acc = d if acc is None else [a + b for a, b in zip(acc, d)]
print("dwp_mlp, workgroup 0 wave 0, mean cycles over 8 updates (100 MHz counter x ? -- see total):")
for nme, v in zip(names[1:], acc):
    print("  %-34s %8.0f" % (nme, v / 8.0))
print("  total %.0f" % (sum(acc) / 8.0))
