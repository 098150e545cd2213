"""Statistical equivalence of the HIP kernel and the CPU oracle under the same random policy (not trajectory parity:
trajectories separate chaotically after ~10 steps).  Same initial distribution, same action stream per env index,
in-kernel RNG on both (bit-identical Philox); compares episode-length and reward statistics."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaacgymdyros_amd import abi
from isaacgymdyros_amd.config import default_cfg
from isaacgymdyros_amd.dyros_dynamic_walk import DyrosDynamicWalk
from isaacgymdyros_amd.task_constants import load_task_constants
from oracle.oracle import OracleSim

N, STEPS = 1024, 400
cfg = default_cfg(N, "cuda:0")
cfg["sim"]["mi355"]["force_perturb_start"] = True
env = DyrosDynamicWalk(cfg, "cuda:0", 0, True)
ora = OracleSim(N, task_const=load_task_constants(), cfg=env._ccfg)
for k, t in env._buf.items():
    ora.buf[k][...] = t.cpu().numpy()
g = torch.Generator().manual_seed(5)
rew = {"hip": [], "ora": []}; done = {"hip": 0, "ora": 0}; fall_len = {"hip": [], "ora": []}
for t in range(STEPS):
    a = torch.rand(N, 13, generator=g) * 2 - 1
    o, r, d, ex = env.step(a.cuda())
    ora.step(a.numpy(), None, t)
    rew["hip"].append(float(r.mean())); rew["ora"].append(float(ora.buf["rew_buf"].mean()))
    done["hip"] += int(d.sum()); done["ora"] += int(ora.buf["reset_buf"].sum())
el_h = env.epi_len_log.cpu().numpy().ravel(); el_o = abi.es_view(ora.buf["env_state"], "epi_len_log").ravel()
print("mean reward/step   hip %.4f  oracle %.4f" % (np.mean(rew["hip"]), np.mean(rew["ora"])))
print("episodes finished  hip %d  oracle %d" % (done["hip"], done["ora"]))
print("last episode len   hip %.1f +- %.1f  oracle %.1f +- %.1f" % (el_h.mean(), el_h.std(), el_o.mean(), el_o.std()))
cfh = env.contact_forces.cpu().numpy(); cfo = ora.buf["contact_forces"]
print("mean |contact|     hip %.2f  oracle %.2f" % (np.abs(cfh).mean(), np.abs(cfo).mean()))
