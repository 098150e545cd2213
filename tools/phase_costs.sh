#!/bin/bash
# Per-phase cost of dw_k_simulate by early-exit ablation builds (GPU box).  Restores the shipped build at the end.
cd $(dirname $0)/..
cat > /tmp/_pc.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
N = int(__import__("os").environ.get("DW_PC_N", "16384"))
cfg = default_cfg(N, "cuda:0"); cfg["task"]["randomize"] = False
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
env._buf["root_states"][:, 2] = 0.928          # soles touching: contact pipeline active
tau = torch.zeros(N, 33, device="cuda")
for _ in range(10): env.simulate(tau)
env._buf["root_states"][:, 2] = 0.928; env._buf["root_states"][:, 3:6] = 0; env._buf["root_states"][:, 6] = 1
env._buf["dof_state"][..., 0] = torch.tensor(INITIAL_DOF_POS, device="cuda"); env._buf["dof_state"][..., 1] = 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): env.simulate(tau)
e1.record(); torch.cuda.synchronize()
print("%.4f" % (e0.elapsed_time(e1) / 40))
PY
prev=0
for n in 1 2 3 4 5 6 7 8 9 10 11 99; do
  tools/ab_build.sh "s/XX/XX/" "-DDW_PROFILE_STOP=$n"
  t=$(python /tmp/_pc.py 2>/dev/null | tail -1)
  echo "stop after phase $n: $t ms"
done
python isaacgymdyros_amd/build.py > /dev/null 2>&1
