"""Phase stamps of the lane kernels' substep (a -DDL_STAMPS build, tools/lane_lib.sh): clock of every wave of workgroup 0 at the
phase boundaries of dw_simulate, median over launches.  usage: DW_LIB=isaacgymdyros_amd/_ab/stamps.so python tools/lane_stamps.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaacgymdyros_amd import _lib
_lib.LIB_PATH = os.environ.get("DW_LIB", _lib.LIB_PATH)
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
import json
cfg = default_cfg(N, "cuda:0"); cfg["sim"]["mi355"]["pipeline"] = 4
cfg["sim"]["mi355"].update(json.loads(os.environ.get("DW_MI355", "{}")))
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
g = torch.Generator(device="cuda").manual_seed(42)
acts = [torch.rand(N, 13, generator=g, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(30): env.step(acts[i % 8])          # robots in their usual mix of states
tau = (torch.rand(N, 33, generator=g, device="cuda") * 2 - 1) * 20
rows = []
for i in range(40):
    env._buf["stacked_rewards"].zero_()
    env.simulate(tau); torch.cuda.synchronize()
    rows.append(env._buf["stacked_rewards"].flatten()[:128].cpu().numpy().reshape(4, 32).copy())
med = np.median(np.stack(rows[5:]), axis=0)
names = ["entry", "FK done", "after FK barrier", "SC done", "inward a done", "before X1", "after X1", "before X2", "after X2", "base solve done",
         "outward 2 done", "W responses", "G / A blocks", "warm start", "PGS done", "contact done", "after barrier", "outward 3 done", "base integ", "after final barrier"]
print("%-22s %10s %10s %10s %10s   (cycles since entry; delta to previous stamp in brackets)" % ("stamp", "wave0", "wave1", "wave2", "wave3"))
for n, nm in enumerate(names):
    cells = []
    for w in range(4):
        prev = 0
        for m in range(n - 1, -1, -1):
            if med[w, m] > 0: prev = med[w, m]; break
        cells.append("%6d[%5d]" % (med[w, n], med[w, n] - prev) if (med[w, n] > 0 or n == 0) else "      -      ")
    print("%-22s %s" % (nm, " ".join(cells)))
if "--step" in sys.argv:
    # step-level stamps of dw_step: 14 per wave in gate_acc[200 ..], window DL_STAMP2_BASE of the build (argument after --step)
    base = int(sys.argv[sys.argv.index("--step") + 1])
    names2 = {0: "entry loads issued", 1: "item loads issued", 2: "pre scalars done", 3: "after barrier", 4: "actions done", 5: "actuator done", 6: "noise done",
              7: "after barrier", 8: "substep 1 done", 9: "epilogue 1 done", 10: "after barrier", 11: "substep 2 done", 12: "epilogue 2 done", 13: "after barrier",
              14: "state written", 15: "post: before barrier", 16: "records requested", 17: "staged", 18: "patched", 19: "Q1 done", 20: "Q2 reward done",
              21: "Q3 done", 22: "reset done", 23: "taps requested", 24: "Q4 obs done", 25: "Q5 obs_buf done", 26: "Q6 done", 31: "end"}
    rows = []
    for i in range(60):
        env._buf["gate_acc"][200:].zero_()
        env.step(acts[i % 8]); torch.cuda.synchronize()
        rows.append(env._buf["gate_acc"][200:256].cpu().numpy().reshape(4, 14).astype(np.float64).copy())
    med = np.median(np.stack(rows[10:]), axis=0)
    if "--sub" in sys.argv:
        names2 = dict(enumerate(names))
    print("step-level stamps (window %d..%d)" % (base, base + 13))
    for n in range(14):
        cells = []
        for w in range(4):
            prev = 0
            for m in range(n - 1, -1, -1):
                if med[w, m] > 0: prev = med[w, m]; break
            cells.append("%7d[%6d]" % (med[w, n], med[w, n] - prev) if med[w, n] > 0 else "       -       ")
        print("%-22s %s" % (names2.get(base + n, str(base + n)), " ".join(cells)))
env.close()
