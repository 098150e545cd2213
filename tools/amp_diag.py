"""Per-column differences of the HIP row f-3 functions against the reference fixture (what the tolerances in tests/test_amp_gpu.py are read from)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_amp_gpu import hip_outputs, oracle_outputs, ulps, G
g = np.load(G)
h, o = hip_outputs(g), oracle_outputs(g)
d = ulps(h["obs"], g["ref_obs"])
print("obs: columns with mismatches", {int(c): (int((d[:, c] > 0).sum()), int(d[:, c].max())) for c in range(36) if d[:, c].max() > 0})
print("obs abs err max", np.abs(h["obs"] - g["ref_obs"]).max())
d = ulps(h["reward_values"], g["ref_reward_values"])
print("reward values: mismatches / max ulp per term", (d > 0).sum(axis=0), d.max(axis=0), "abs", np.abs(h["reward_values"] - g["ref_reward_values"]).max())
print("reward abs", np.abs(h["reward"] - g["ref_reward"]).max())
print("nw reward8 abs", np.abs(h["nw_reward8"] - g["ref_nw_reward8"]).max(axis=0), "total", np.abs(h["nw_total"] - g["ref_nw_total"]).max())
print("hip vs oracle obs ulp", ulps(h["obs"], o["obs"]).max(), "reward values", ulps(h["reward_values"], o["reward_values"]).max())
