"""Diagnostic for the terrain substep: HIP (both pipelines) vs oracle, per-env joint-rate error and where it comes from."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hip_backend import make_env
from isaacgymdyros_amd.terrain import Terrain, TerrainCfg
from isaacgymdyros_amd.task_constants import INITIAL_DOF_POS
from oracle.oracle import OracleSim
tdict = dict(mesh_type="heightfield", curriculum=True, num_rows=2, num_cols=4, border_size=2, max_init_terrain_level=1,
             terrain_proportions=[0.2, 0.2, 0.3, 0.3, 0.0])
N = 256
np.set_printoptions(linewidth=220, precision=3)
for pipe in (2, 1):
    for dbl in (False, True):
        env = make_env(N, randomize=False, terrain=tdict, seed=3, pipeline=pipe)
        t = Terrain(TerrainCfg(**tdict), N, seed=3)
        A = OracleSim(N, terrain=t, double=dbl)
        rng = np.random.default_rng(5)
        org = t.env_origins.reshape(-1, 3)[rng.integers(0, 8, size=N)]
        A.buf["root_states"][:, 0:2] = org[:, 0:2] + rng.uniform(-3, 3, size=(N, 2))
        A.buf["root_states"][:, 2] = t.height_at(A.buf["root_states"][:, 0], A.buf["root_states"][:, 1]) + 0.93 + rng.uniform(-0.03, 0.05, size=N)
        A.buf["root_states"][:, 6] = 1.0
        A.buf["root_states"][:, 7:13] = rng.normal(size=(N, 6)) * 0.3
        A.buf["dof_state"][:, :, 0] = np.asarray(INITIAL_DOF_POS) + rng.normal(size=(N, 33)) * 0.05
        A.buf["dof_state"][:, :, 1] = rng.normal(size=(N, 33)) * 0.5
        env.root_states.copy_(torch.from_numpy(A.buf["root_states"].astype(np.float32)).cuda())
        env._buf["dof_state"].copy_(torch.from_numpy(A.buf["dof_state"].astype(np.float32)).cuda())
        tau = rng.uniform(-30, 30, size=(N, 33)).astype(np.float32)
        A.simulate(tau); env.simulate(torch.from_numpy(tau).cuda()); torch.cuda.synchronize()
        ds = env._buf["dof_state"].cpu().numpy()
        err = np.abs(A.buf["dof_state"][..., 1] - ds[..., 1]).max(axis=1)
        cfa, cfb = A.buf["contact_forces"], env.contact_forces.cpu().numpy()
        touching = np.abs(cfa).max(axis=(1, 2)) > 0
        nz_a = np.linalg.norm(cfa, axis=2) > 0; nz_b = np.linalg.norm(cfb, axis=2) > 0
        same_set = (nz_a == nz_b).all(axis=1)
        clamp = (np.abs(np.abs(ds[..., 1]) - 4.03) < 1e-6).any(axis=1) | (np.abs(np.abs(A.buf["dof_state"][..., 1]) - 4.03) < 1e-6).any(axis=1)
        print("pipeline %d oracle %s: max dqd %.2e | touching %d | same contact-body set %d | envs > 2e-4: %d (of those: set differs %d, a rate at the 4.03 clamp %d)"
              % (pipe, "fp64" if dbl else "fp32", err.max(), touching.sum(), same_set.sum(), (err > 2e-4).sum(), ((err > 2e-4) & ~same_set).sum(), ((err > 2e-4) & clamp).sum()))
        print("   error quantiles 50/90/99/max: %.1e %.1e %.1e %.1e ; among untouched envs max %.1e" % (np.quantile(err, .5), np.quantile(err, .9), np.quantile(err, .99), err.max(), err[~touching].max() if (~touching).any() else 0))
        w = int(err.argmax())
        print("   worst env %d: |F| oracle %s  hip %s" % (w, np.round(np.linalg.norm(cfa[w], axis=1)[[7, 8, 15, 16]], 2), np.round(np.linalg.norm(cfb[w], axis=1)[[7, 8, 15, 16]], 2)))
