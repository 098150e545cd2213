#!/usr/bin/env python3
"""Mint tests/golden/amp_disc_ref.npz from the reference's AMP subclass module.  TEST INFRASTRUCTURE ONLY.

Runs only in the container where /root/reference is mounted.  Imported from where they lie (through the stub packages of
oracle/ref_harness.py; `termcolor`, which the image lacks and the reference's logger imports for colours, is an empty stand-in):

  build_amp_observations   tasks/tocabi_amp_lower.py:310-350 -- called on seeded inputs (both values of local_root_obs; the
                           33-wide dof tensors of the simulation and the 12-wide ones of the motion library)
  TocabiLowerMotionLib     tasks/amp/utils_amp/tocabi_lower_motion_lib.py -- constructed on the synthetic tables of
                           tests/amp_motion_synth.py (the reference's own tables are not in its checkout), then sample_motions /
                           sample_time under a seeded numpy generator and get_motion_state on those and on edge-case times

What is committed is DATA: the inputs and the reference's outputs (the tables themselves are regenerated from integers by the
test, tests/amp_motion_synth.py).
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_harness as RH                                 # noqa: E402
import amp_motion_synth as SY                                         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "amp_disc_ref.npz")


def load_module():
    RH.load_reference(lambda: None)
    RH.sys.modules.setdefault("isaacgymenvs.tasks.amp", RH.types.ModuleType("isaacgymenvs.tasks.amp")).__path__ = [os.path.join(RH.IGE, "tasks", "amp")]
    RH._load("isaacgymenvs.tasks.amp.tocabi_amp_lower_base", os.path.join(RH.IGE, "tasks", "amp", "tocabi_amp_lower_base.py"))
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules.setdefault("termcolor", tc)
    sys.path.insert(0, RH.IGE)                       # (motion_lib.py imports `tasks.amp.humanoid_amp_base` relative to isaacgymenvs/)
    if not hasattr(np, "int"):
        np.int = int                                 # (motion_lib.py:237, removed from numpy 1.24)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return RH._load("isaacgymenvs.tasks.tocabi_amp_lower", os.path.join(RH.IGE, "tasks", "tocabi_amp_lower.py"))


def rand_quat(rng, n, max_angle):
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, max_angle, size=n)
    return np.concatenate([ax * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1).astype(np.float32)


def main():
    sub = load_module()
    rng = np.random.default_rng(20261004)
    T = torch.from_numpy
    d = {}
    # ---------------- build_amp_observations
    N = 256
    root = np.zeros((N, 13), np.float32)
    root[:, 0:3] = rng.normal(size=(N, 3)) * np.array([3, 3, 0.2]) + np.array([0, 0, 0.9])
    root[:, 3:7] = rand_quat(rng, N, 3.1)                              # any heading, large tilts
    root[:8, 3:7] = np.array([0, 0, 0, 1], np.float32)                 # identity: heading exactly 0
    yaw = rng.uniform(-np.pi, np.pi, size=24)
    root[8:32, 3:7] = np.stack([0 * yaw, 0 * yaw, np.sin(yaw / 2), np.cos(yaw / 2)], axis=1)      # pure yaw
    root[32:36, 3:7] = np.array([0, 0, 1, 0], np.float32)              # heading pi
    root[36:40, 3:7] = np.array([0, np.sqrt(0.5), 0, np.sqrt(0.5)], np.float32)      # pitch pi/2: quat2euler's degenerate branch region
    root[:, 7:13] = rng.normal(size=(N, 6))
    d["root_states"] = root
    d["dof_pos"] = rng.normal(size=(N, 33)).astype(np.float32)
    d["dof_vel"] = (rng.normal(size=(N, 33)) * 3).astype(np.float32)
    d["key_pos"] = (root[:, None, 0:3] + rng.normal(size=(N, 2, 3)) * 0.4).astype(np.float32)
    for flag in (False, True):
        d["ref_obs_local%d" % flag] = sub.build_amp_observations(T(root), T(d["dof_pos"]), T(d["dof_vel"]), flag, T(d["key_pos"])).numpy()
    d["ref_obs_12wide"] = sub.build_amp_observations(T(root), T(np.ascontiguousarray(d["dof_pos"][:, :12])),
                                                     T(np.ascontiguousarray(d["dof_vel"][:, :12])), False, T(d["key_pos"])).numpy()
    d["num_amp_obs_per_step"] = np.array(sub.NUM_AMP_OBS_PER_STEP)
    # ---------------- TocabiLowerMotionLib on synthetic tables
    with tempfile.TemporaryDirectory() as tmp:
        yml = SY.write(tmp)
        with contextlib.redirect_stdout(io.StringIO()):
            lib = sub.TocabiLowerMotionLib(motion_file=yml, num_dofs=33, device="cpu")
        d["ml_lengths"], d["ml_weights"] = np.array(lib._motion_lengths), np.array(lib._motion_weights)
        d["ml_fps"], d["ml_dt"], d["ml_num_frames"] = np.array(lib._motion_fps), np.array(lib._motion_dt), np.array(lib._motion_num_frames)
        d["ml_first_rows"] = np.stack([m[0] for m in lib._motions])
        d["ml_last_rows"] = np.stack([m[-1] for m in lib._motions])
        np.random.seed(1234)
        ids = lib.sample_motions(512)
        times = lib.sample_time(ids)
        times_trunc = lib.sample_time(ids, truncate_time=0.004)
        # edge cases appended: time 0, the motion's end, past the end, negative (the AMP history asks for times before 0)
        eid = np.repeat(np.arange(lib.num_motions()), 4)
        et = np.stack([np.zeros(lib.num_motions()), lib._motion_lengths, lib._motion_lengths + 0.01, -0.002 * np.ones(lib.num_motions())], axis=1).reshape(-1)
        ids_all, times_all = np.concatenate([ids, eid]), np.concatenate([times, et])
        d["ml_ids"], d["ml_times"], d["ml_times_trunc"] = ids, times, times_trunc
        d["ml_query_ids"], d["ml_query_times"] = ids_all, times_all
        f0, f1, bl = lib._calc_frame_blend(times_all, lib._motion_lengths[ids_all], lib._motion_num_frames[ids_all], np.abs(lib._motion_dt[ids_all]))
        d["ml_frame0"], d["ml_frame1"], d["ml_blend"] = f0, f1, bl
        out = lib.get_motion_state(ids_all, times_all)
        for name, t in zip(("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel", "key_pos"), out):
            d["ml_" + name] = t.numpy()
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
