#!/usr/bin/env python3
"""Mint tests/golden/*.npz from the reference's own Python.  TEST INFRASTRUCTURE ONLY.

Runs only in the container where /root/reference is mounted.  The reference's DyrosDynamicWalk class is
imported from where it lies (oracle/ref_harness.py) and driven over a fake gym; what is committed is DATA:
inputs (initial state, per-step actions, the injected-noise record rebuilt from the reference's recorded
RNG draws, per-step injected physics state) and the reference's outputs.

  task_logic_frozen.npz   `gym.simulate` is a no-op and a fresh random physics state is written before every
                          step, so everything recorded is the reference's torch task logic and nothing else:
                          mocap target, action/torque/delay pipeline, push perturbation, encoder model,
                          termination, the 14-term reward, reset_idx, the 487-d observation with history.
  dr_reset.npz            as task_logic_frozen, but with the reference's domain randomisation ON (`env.randomize = True`):
                          every reset runs the reference's `apply_randomizations` (tasks/base/vec_task.py:519-733) with the
                          real `isaacgym/gymutil.py` helpers over the fake gym.  Recorded: the numpy samples each call drew
                          (as U[0,1) words of the noise record: u = (sample - lo) / (hi - lo)), and the damping / armature
                          the fake gym was handed by `set_actor_dof_properties`, plus randomize_buf.  Pins: additive /
                          scaling FROM THE ORIGINAL value, independent per DoF, only for envs with reset_buf set and
                          randomize_buf >= frequency, randomize_buf zeroed for exactly those.
  whole_step_oracle.npz   the same class stepping over the ORACLE's physics (the closed PhysX engine cannot
                          be run: physics parity is unpinned).  Pins the orchestration
                          (call order, substep loop, late updates) of dw_step; the HIP library is held to it
                          within the float tolerance stated in tests/.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from isaacgymdyros_amd import abi                                    # noqa: E402
from isaacgymdyros_amd.task_constants import load_task_constants     # noqa: E402
from oracle import parity as P                                       # noqa: E402
from oracle import ref_harness as RH                                 # noqa: E402
from oracle.oracle import OracleSim                                  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PER_STEP = ["obs_buf", "rew_buf", "reset_buf", "progress_buf", "timeout_buf", "stacked_rewards",
            "root_states", "dof_state", "qpos_noise", "qvel_noise", "target_data_qpos", "target_data_force",
            "action_torque", "time", "epi_len", "mocap_data_idx", "pert_on", "perturbation_count", "magnitude",
            "phase", "contact_reward_sum", "contact_reward_mean", "delay_idx", "simul_len", "motor_constant_scale",
            "qpos_bias", "quat_bias", "target_vel", "init_mocap_data_idx", "perturb_timing", "epi_len_log"]
FINAL = ["obs_history", "action_history", "action_log", "actions_pre", "pre_joint_velocity_states",
         "foot_force_pre", "action_torque_pre", "qpos_pre", "contact_forces"]


def random_state(rng, N, tc, model):
    root = np.zeros((N, 13), np.float32)
    root[:, 0:2] = rng.normal(size=(N, 2)) * 3
    root[:, 2] = 0.93 + rng.normal(size=N) * 0.02
    ang = rng.uniform(0, 0.7, size=N) * (rng.uniform(size=N) < 0.5)
    ang[rng.uniform(size=N) < 0.15] = 0
    axis = rng.normal(size=(N, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    root[:, 3:6] = axis * np.sin(ang / 2)[:, None]
    root[:, 6] = np.cos(ang / 2)
    root[:, 7:13] = rng.normal(size=(N, 6)) * 0.5
    dof = np.zeros((N, 33, 2), np.float32)
    dof[:, :, 0] = tc["initial_dof_pos"] + rng.normal(size=(N, 33)) * 0.1
    dof[:, :, 1] = rng.normal(size=(N, 33)) * 1.0
    cf = np.zeros((N, 38, 3), np.float32)
    for foot in (model.left_foot_idx, model.right_foot_idx):
        on = rng.uniform(size=N) < 0.7
        cf[:, foot, 2] = on * rng.uniform(0, 1600, size=N)
        cf[:, foot, 0:2] = on[:, None] * rng.normal(size=(N, 2)) * 50
    hit = rng.uniform(size=N) < 0.08
    body = rng.integers(0, 38, size=N)
    for e in range(N):
        if hit[e] and body[e] not in (model.left_foot_idx, model.right_foot_idx):
            cf[e, body[e]] = rng.normal(size=3) * 3
    return root, dof, cf


# reference TerrainCfg class attributes of the "terrain" fixture (row f-4): a small curriculum map without the tile type
# whose reference code needs scipy's removed interp2d
TERRAIN_CASE = dict(mesh_type="heightfield", curriculum=True, num_rows=3, num_cols=4, border_size=2,
                    max_init_terrain_level=2, terrain_proportions=[0.3, 0.0, 0.3, 0.2, 0.2])


class DrRecorder:
    """Records the float64 samples `generate_random_samples` hands `apply_random_samples` (python/isaacgym/gymutil.py:
    584-607), in call order, while active."""

    def __init__(self, gymutil):
        self.gu, self.log = gymutil, []

    def __enter__(self):
        self._orig = self.gu.generate_random_samples

        def wrapped(params, shape, *a, **k):
            out = self._orig(params, shape, *a, **k)
            self.log.append((tuple(params["range"]), params["operation"], np.array(out, dtype=np.float64, copy=True)))
            return out
        self.gu.generate_random_samples = wrapped
        return self

    def __exit__(self, *exc):
        self.gu.generate_random_samples = self._orig


def run(kind: str, N: int, steps: int, seed: int):
    tc = load_task_constants()
    terr = kind == "terrain"
    dr = kind == "dr"
    frozen = kind == "frozen" or terr or dr
    A = OracleSim(N, task_const=tc, debug_freeze_physics=int(frozen))
    env, fake, mods = RH.make_reference_env(A, N, seed=seed, terrain=TERRAIN_CASE if terr else None)
    env.randomize = dr        # (other fixtures: off, so that they pin the task logic alone)
    env.reset()
    B = OracleSim(N, task_const=tc, randomize_dof_on_reset=int(dr), debug_freeze_physics=int(frozen),
                  terrain=env.terrain if terr else None, **(dict(max_episode_length_s=float(env.max_episode_length_s)) if terr else {}))
    for k in ("mass_scale", "dof_damping", "dof_armature"):
        B.buf[k][:] = A.buf[k]
    if frozen:
        env.perturb_timing[:] = torch.randint(0, 6, (N,))
        env.progress_buf[0] = 7990
        env.progress_buf[1] = 7996
    P.sync_from_reference(env, B.buf)
    init = {k: v.copy() for k, v in B.buf.items()}
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed + 1)
    rec_steps = {k: [] for k in PER_STEP}
    actions_all, noise_all, inj = [], [], {"root": [], "dof": [], "cf": []}
    force_at = 3 if frozen else 10
    lvl_steps, org_steps = [], []
    dr_steps = {"dof_damping": [], "dof_armature": [], "randomize_buf": [], "dr_envs": []}
    for t in range(steps):
        a = torch.rand(N, 13, generator=g) * 2.4 - 1.2       # some outside +-1: exercises the clamp
        if t == force_at:
            env.perturb_start[:, 0] = True
        if frozen:
            root, dof, cf = random_state(rng, N, tc, A.model)
            if terr:          # around the tile origin: some robots have "walked" past half a tile, some hardly moved
                root[:, 0:2] = env.env_origins[:, 0:2].numpy() + rng.normal(size=(N, 2)) * 3
                root[:, 2] += env.env_origins[:, 2].numpy()
            env.root_states[:] = torch.from_numpy(root)
            env.dof_state.view(N, 33, 2)[:] = torch.from_numpy(dof)
            env.contact_forces[:] = torch.from_numpy(cf)
            inj["root"].append(root); inj["dof"].append(dof); inj["cf"].append(cf)
        pert_ids = None
        if bool(env.perturb_start[0, 0]):
            pert_ids = torch.nonzero((env.epi_len % 2000.0) == env.perturb_timing).flatten().numpy()
        rb_before = env.randomize_buf.numpy().copy()
        with RH.RngRecorder() as rec, DrRecorder(mods["gymutil"]) as drrec:
            o, r, d, ex = env.step(a.clone())
        reset_ids = d.nonzero().flatten().numpy()
        nz = P.noise_from_log(rec.log, N, pert_ids, reset_ids, terrain_levels=TERRAIN_CASE["num_rows"] if terr else 0,
                              terrain_curriculum=terr)
        if dr:
            # apply_randomizations walks the randomised envs in index order and draws, per env, damping then armature
            # (cfg/task/DyrosDynamicWalk.yaml:103-115); the mass is setup-only (:82-88) and draws nothing here
            dr_ids = [int(i) for i in reset_ids if rb_before[i] + 1 >= 1]
            assert len(drrec.log) == 2 * len(dr_ids), (len(drrec.log), len(dr_ids))
            for n_, i in enumerate(dr_ids):
                (lo, hi), op, smp = drrec.log[2 * n_]
                assert op == "additive" and smp.shape == (33,)
                nz[i, abi.K["DW_NZ_DR_DAMP"]:abi.K["DW_NZ_DR_DAMP"] + 33] = (smp - lo) / (hi - lo)
                (lo, hi), op, smp = drrec.log[2 * n_ + 1]
                assert op == "scaling" and smp.shape == (33,)
                nz[i, abi.K["DW_NZ_DR_ARM"]:abi.K["DW_NZ_DR_ARM"] + 33] = (smp - lo) / (hi - lo)
            dr_steps["dof_damping"].append(A.buf["dof_damping"].copy()); dr_steps["dof_armature"].append(A.buf["dof_armature"].copy())
            dr_steps["randomize_buf"].append(env.randomize_buf.numpy().copy())
            mask = np.zeros(N, np.int64); mask[dr_ids] = 1
            dr_steps["dr_envs"].append(mask)
        actions_all.append(a.numpy().copy())
        noise_all.append(nz)
        snap = P.snapshot_reference(env, ex)
        for k in PER_STEP:
            rec_steps[k].append(snap[k])
        if terr:
            lvl_steps.append(snap["terrain_levels"]); org_steps.append(snap["env_origins"])
    out = dict(N=N, steps=steps, force_perturb_step=force_at,
               actions=np.stack(actions_all), noise=np.stack(noise_all))
    for k, v in init.items():
        out["init_" + k] = v
    for k in PER_STEP:
        out["step_" + k] = np.stack(rec_steps[k])
    for k in FINAL:
        out["final_" + k] = snap[k]
    if frozen:
        out["inj_root"] = np.stack(inj["root"]); out["inj_dof"] = np.stack(inj["dof"]); out["inj_cf"] = np.stack(inj["cf"])
    os.makedirs(OUT, exist_ok=True)
    if terr:
        for k in ("terrain", "custom_origins", "terrain_rows", "terrain_cols", "terrain_hscale", "terrain_vscale", "terrain_border",
                  "terrain_curriculum", "terrain_num_levels", "terrain_num_types", "terrain_env_length", "max_episode_length_s"):
            out["cfg_" + k] = getattr(B.cfg, k)
        out["step_terrain_levels"] = np.stack(lvl_steps); out["step_env_origins"] = np.stack(org_steps)
    if dr:
        for k, v in dr_steps.items():
            out["step_" + k] = np.stack(v)
    path = os.path.join(OUT, "dr_reset.npz" if dr else ("terrain_logic_frozen.npz" if terr else ("task_logic_frozen.npz" if frozen else "whole_step_oracle.npz")))
    np.savez_compressed(path, **out)
    nres = int(np.stack(rec_steps["reset_buf"]).sum())
    npert = int(np.stack(rec_steps["pert_on"]).sum())
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), "resets", nres, "pert_on env-steps", npert)


if __name__ == "__main__":
    warnings.filterwarnings("ignore")
    if not RH.available():
        sys.exit("reference checkout not present; goldens can only be minted where it is mounted")
    only = sys.argv[1] if len(sys.argv) > 1 else None        # "frozen" | "oracle" | "terrain" | "dr": mint one fixture only
    if only in (None, "frozen"):
        run("frozen", N=24, steps=20, seed=11)
    if only in (None, "oracle"):
        run("oracle", N=8, steps=120, seed=5)
    if only in (None, "terrain"):
        run("terrain", N=24, steps=24, seed=17)
    if only in (None, "dr"):
        run("dr", N=24, steps=24, seed=23)
