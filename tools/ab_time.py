"""A/B timing of two builds of libdyroswalk_hip.so on the same GPU box, interleaved (A B A B ...), so box-to-box and
clock drift do not masquerade as a kernel difference.
usage: python tools/ab_time.py <libA.so> <libB.so> [N=16384] [rounds=3]
(the libraries live under isaacgymdyros_amd/_ab/, git-ignored; tools only -- the product always loads the in-tree lib)"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import isaacgymdyros_amd._lib as L
L.LIB_PATH = sys.argv[1]
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
N = int(sys.argv[2])
cfg = default_cfg(N, "cuda:0")
import json
cfg["sim"].setdefault("mi355", {}).update(json.loads(os.environ.get("DW_AB_MI355", "{}")))      # e.g. {"self_collision": 0}
cfg["sim"]["mi355"]["alias_obs"] = True
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(100): env.step(acts[i %% 8])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 400
e0.record()
for i in range(K): env.step(acts[i %% 8])
e1.record(); torch.cuda.synchronize()
print("%%.4f" %% (e0.elapsed_time(e1) / K))
''' % ROOT

def main():
    a, b = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
    n = sys.argv[3] if len(sys.argv) > 3 else "16384"
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    res = {a: [], b: []}
    for _ in range(rounds):
        for lib in (a, b):
            out = subprocess.run([sys.executable, "-c", CHILD, lib, n], capture_output=True, text=True)
            if out.returncode != 0:
                print(out.stderr[-2000:]); sys.exit(1)
            res[lib].append(float(out.stdout.strip().splitlines()[-1]))
    for tag, lib in (("A", a), ("B", b)):
        v = res[lib]
        print("%s %-40s ms/step: %s  mean %.4f" % (tag, os.path.basename(lib), " ".join("%.4f" % x for x in v), sum(v) / len(v)), flush=True)

main()
