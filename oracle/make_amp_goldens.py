#!/usr/bin/env python3
"""Mint tests/golden/amp_lower_ref.npz from the reference's own TorchScript functions.  TEST INFRASTRUCTURE ONLY.

Runs only in the container where /root/reference is mounted.  The four functions of SURVEY.md section 8 row f-3 are imported
from where they lie (through the stub packages of oracle/ref_harness.py) and called on seeded inputs; what is committed is
DATA -- the inputs and the reference's outputs:

  compute_humanoid_observations / compute_humanoid_reward / compute_humanoid_reset
        tasks/amp/tocabi_amp_lower_base.py:918-1069 (TocabiAMPLower's env side)
  compute_humanoid_walk_reward
        tasks/tocabi_new_walk.py:384-496 (the one function of TocabiNewWalk that runs; the class constructor's own
        observation normalisation, :558-567, subtracts a [37] mean from a [N,30] observation -- recorded here as the exception
        text torch raises for those shapes)

The inputs cover the branches: feet above / below the contact-force threshold, fallen and flying bodies, tilted bases past
pi/4, progress at 0, 1, 2 and at the episode limit, every phase segment of sync_reward, zero foot forces, converged legs.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness as RH                                 # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "amp_lower_ref.npz")


def load_functions():
    RH.load_reference(lambda: None)
    RH.sys.modules.setdefault("isaacgymenvs.tasks.amp", RH.types.ModuleType("isaacgymenvs.tasks.amp")).__path__ = [os.path.join(RH.IGE, "tasks", "amp")]
    amp = RH._load("isaacgymenvs.tasks.amp.tocabi_amp_lower_base", os.path.join(RH.IGE, "tasks", "amp", "tocabi_amp_lower_base.py"))
    nw = RH._load("isaacgymenvs.tasks.tocabi_new_walk", os.path.join(RH.IGE, "tasks", "tocabi_new_walk.py"))
    return amp, nw


def rand_quat(rng, n, max_angle):
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, max_angle, size=n)
    return np.concatenate([ax * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1).astype(np.float32)


def main():
    amp, nw = load_functions()
    rng = np.random.default_rng(20260403)
    N = 256
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    T = torch.from_numpy
    d = {}
    # ---------------- TocabiAMPLower
    root = np.zeros((N, 13), np.float32)
    root[:, 0:2] = rng.normal(size=(N, 2)) * 2
    root[:, 2] = 0.9 + rng.normal(size=N) * 0.15                      # some below terminationHeight 0.6? push a few there
    root[:8, 2] = rng.uniform(0.3, 0.6, size=8)
    root[:, 3:7] = rand_quat(rng, N, 1.2)                              # up to ~69 deg: both sides of pi/4
    root[8:16, 3:7] = np.array([0, 0, 0, 1], np.float32)
    root[:, 7:13] = rng.normal(size=(N, 6)) * 0.8
    d["root_states"] = root
    d["rootvel_noise"] = f32(rng.uniform(-0.025, 0.025, size=(N, 6)))
    d["dof_pos"] = f32(rng.normal(size=(N, 33)) * 0.5)
    d["dof_pos_bias"] = f32(rng.uniform(-0.0314, 0.0314, size=(N, 12)))
    d["quat_bias"] = f32(rng.uniform(-0.021, 0.021, size=(N, 3)))
    d["dof_vel"] = f32(rng.normal(size=(N, 33)) * 2)
    d["dof_vel_pre"] = f32(d["dof_vel"] + rng.normal(size=(N, 33)) * 0.1)
    d["commands"] = f32(np.stack([rng.uniform(-0.5, 1.0, N), np.zeros(N), rng.uniform(-0.3, 0.3, N)], axis=1))
    d["actions"] = f32(rng.uniform(-1, 1, size=(N, 12)))
    d["actions_pre"] = f32(rng.uniform(-1, 1, size=(N, 12)))
    d["motor_efforts"] = f32([333, 232, 263, 289, 222, 166] * 2)
    d["total_mass"] = f32(100.0 * rng.uniform(0.8, 1.2, size=(N, 1)))
    cf = np.zeros((N, 38, 3), np.float32)
    cf[:, 8, 2] = rng.uniform(0, 2500, size=N) * (rng.uniform(size=N) < 0.7)
    cf[:, 16, 2] = rng.uniform(0, 2500, size=N) * (rng.uniform(size=N) < 0.7)
    cf[:, 8, :2] = rng.normal(size=(N, 2)) * 50
    hit = rng.uniform(size=N) < 0.25
    cf[hit, rng.integers(0, 38, size=hit.sum()), rng.integers(0, 3, size=hit.sum())] = rng.uniform(0.5, 3.0, size=hit.sum())
    d["contact_force"] = cf
    obs = amp.compute_humanoid_observations(T(root), T(d["rootvel_noise"]), T(d["dof_pos"].copy()), T(d["dof_pos_bias"]), T(d["quat_bias"]),
                                            T(d["dof_vel"]), T(d["commands"]), torch.zeros(N, 2, 3))
    d["ref_obs"] = obs.numpy()
    rew, vals, names = amp.compute_humanoid_reward(T(root), T(d["dof_vel"]), T(d["dof_vel_pre"]), T(d["commands"]), T(d["actions"]),
                                                   T(d["actions_pre"]), T(d["motor_efforts"]), T(cf), T(d["total_mass"]))
    d["ref_reward"] = rew.numpy()
    d["ref_reward_values"] = vals.numpy()
    d["reward_names"] = np.array(names)
    # reset
    pos = np.zeros((N, 38, 3), np.float32)
    pos[:, :, 2] = rng.uniform(0.0, 1.5, size=(N, 38))
    pos[:, 0, 2] = root[:, 2]
    pos[:, 8, 2] = rng.uniform(0.0, 0.7, size=N)
    pos[:, 16, 2] = rng.uniform(0.0, 0.7, size=N)
    rot = np.zeros((N, 38, 4), np.float32)
    rot[:, :, 3] = 1
    rot[:, 0] = root[:, 3:7]
    prog = rng.integers(0, 8000, size=N).astype(np.int64)
    prog[:6] = [0, 1, 2, 7998, 7999, 8000]
    d["rigid_body_pos"], d["rigid_body_rot"], d["progress_buf"] = pos, rot, prog
    d["contact_body_ids"] = np.array([8, 16], np.int64)
    for early in (True, False):
        rs, term = amp.compute_humanoid_reset(torch.zeros(N, dtype=torch.long), T(prog), T(cf), T(d["contact_body_ids"]), T(pos), T(rot),
                                              8000.0, early, 0.6)
        d["ref_reset_early%d" % early] = rs.numpy()
        d["ref_terminated_early%d" % early] = term.numpy()
    # ---------------- TocabiNewWalk reward (12-dof lower body: 13 bodies + head ... the function indexes contact rows 7 and 14)
    NB, ND = 15, 12
    d["nw_reset_buf"] = (rng.uniform(size=N) < 0.1).astype(np.int64)
    d["nw_progress_buf"] = rng.integers(0, 1000, size=N).astype(np.int64)
    d["nw_progress_buf"][:3] = [998, 999, 1000]
    d["nw_target_vel"] = f32(rng.uniform(-0.3, 0.8, size=(N, 2)))
    rp = root.copy()
    rp[:, 2] = 1.0 + rng.normal(size=N) * 0.1
    rp[:10, 2] = 0.5
    d["nw_root_pose_states"] = rp
    d["nw_joint_position_states"] = f32(rng.normal(size=(N, ND)) * 0.3)
    d["nw_joint_velocity_states"] = f32(rng.normal(size=(N, ND)) * 3)
    d["nw_non_feet_idxs"] = np.array([i for i in range(NB) if i not in (7, 14)], np.int64)
    ncf = np.zeros((N, NB, 3), np.float32)
    ncf[:, 7, 2] = rng.uniform(0, 900, size=N) * (rng.uniform(size=N) < 0.6)
    ncf[:, 14, 2] = -rng.uniform(0, 900, size=N) * (rng.uniform(size=N) < 0.6)
    hit = rng.uniform(size=N) < 0.15
    ncf[hit, rng.integers(0, 7, size=hit.sum()), :] = rng.normal(size=(hit.sum(), 3))
    d["nw_contact_forces"] = ncf
    d["nw_q_nominal"] = f32(rng.normal(size=ND) * 0.3)
    d["nw_head_states"] = f32(np.concatenate([rp[:, :3] + rng.normal(size=(N, 3)) * 0.1, rng.normal(size=(N, 10))], axis=1))
    lf = f32(rng.normal(size=(N, 13)))
    rf = f32(rng.normal(size=(N, 13)))
    rf[20:30, :2] = lf[20:30, :2] + 0.01                                # converged legs
    lf[:, 7] = rng.uniform(-0.2, 0.5, size=N)
    rf[:, 7] = rng.uniform(-0.2, 0.5, size=N)
    d["nw_lfoot_states"], d["nw_rfoot_states"] = lf, rf
    ph = f32(rng.uniform(0, 1, size=(N, 1)))
    ph[:8, 0] = [0.0, 0.04, 1 / 12, 0.3, 5 / 12, 0.5, 0.54, 0.99]
    d["nw_phase"] = ph
    with contextlib.redirect_stdout(io.StringIO()):                      # (the function prints env 19's terms)
        tot, rs, r8, nm = nw.compute_humanoid_walk_reward(T(d["nw_reset_buf"]), T(d["nw_progress_buf"]), T(d["nw_target_vel"]), T(rp),
                                                          T(d["nw_joint_position_states"]), T(d["nw_joint_velocity_states"]),
                                                          torch.zeros(N, 12), [int(i) for i in d["nw_non_feet_idxs"]], T(ncf), 0.6, -1.0, 1000.0,
                                                          T(d["nw_q_nominal"]), T(d["nw_head_states"]), T(lf), T(rf), T(ph))
    d["ref_nw_total"], d["ref_nw_reset"], d["ref_nw_reward8"] = tot.numpy(), rs.numpy(), r8.numpy()
    d["nw_names"] = np.array(nm)
    # the constructor's broken normalisation: a [N,30] observation minus a [37] mean (tasks/tocabi_new_walk.py:558-567)
    try:
        torch.zeros(4, 30) - torch.zeros(37)
        d["nw_broadcast_error"] = np.array("")
    except RuntimeError as ex:
        d["nw_broadcast_error"] = np.array(str(ex))
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, {k: v.shape for k, v in d.items() if k.startswith("ref_")})


if __name__ == "__main__":
    main()
