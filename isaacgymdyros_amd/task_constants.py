"""Numeric constants of the DyrosDynamicWalk task and its data tables.

Values the reference hard-codes in the task class or loads from data files; cited per item
(paths relative to python/IsaacGymEnvs/isaacgymenvs/).
"""
from __future__ import annotations

import os

import numpy as np

from .model import ARMATURE, ASSET_DIR, DOF_DAMPING

# tasks/dyros_dynamic_walk.py:58-63 (divided by 9 there)
KP_RAW = [2000.0, 5000.0, 4000.0, 3700.0, 3200.0, 3200.0,
          2000.0, 5000.0, 4000.0, 3700.0, 3200.0, 3200.0,
          6000.0, 10000.0, 10000.0,
          400.0, 1000.0, 400.0, 400.0, 400.0, 400.0, 100.0, 100.0,
          100.0, 100.0,
          400.0, 1000.0, 400.0, 400.0, 400.0, 400.0, 100.0, 100.0]
# tasks/dyros_dynamic_walk.py:65-70 (divided by 3 there)
KV_RAW = [15.0, 50.0, 20.0, 25.0, 24.0, 24.0,
          15.0, 50.0, 20.0, 25.0, 24.0, 24.0,
          200.0, 100.0, 100.0,
          10.0, 28.0, 10.0, 10.0, 10.0, 10.0, 3.0, 3.0,
          2.0, 2.0,
          10.0, 28.0, 10.0, 10.0, 10.0, 10.0, 3.0, 3.0]
# tasks/dyros_dynamic_walk.py:296-301 (motor ctrlranges of the MJCF)
ACTION_HIGH = [333, 232, 263, 289, 222, 166,
               333, 232, 263, 289, 222, 166,
               303, 303, 303,
               64, 64, 64, 64, 23, 23, 10, 10,
               10, 10,
               64, 64, 64, 64, 23, 23, 10, 10]
# tasks/dyros_dynamic_walk.py:95-100
INITIAL_DOF_POS = [0.0, 0.0, -0.24, 0.6, -0.36, 0.0,
                   0.0, 0.0, -0.24, 0.6, -0.36, 0.0,
                   0.0, 0.0, 0.0,
                   0.3, 0.3, 1.5, -1.27, -1.0, 0.0, -1.0, 0.0,
                   0.0, 0.0,
                   -0.3, -0.3, -1.5, 1.27, 1.0, 0.0, 1.0, 0.0]

# tasks/dyros_dynamic_walk.py:922-925 + :423
REWARD_NAMES = ["mimic_body_orientation_reward", "qpos_regulation", "qvel_regulation",
                "contact_force_penalty", "torque_regulation", "torque_diff_regulation", "body_vel_reward",
                "qacc_regulation", "foot_contact_reward", "contact_force_diff_regulation",
                "double_support_force_diff_regulation", "force_thres_penalty", "force_diff_thres_penalty",
                "force_ref_reward", "perturbation"]


def load_task_constants():
    """fp32 arrays exactly as the reference's torch tensors hold them (float32 division of float32 literals)."""
    import torch
    kp = (torch.tensor(KP_RAW, dtype=torch.float) / 9.0).numpy()
    kv = (torch.tensor(KV_RAW, dtype=torch.float) / 3.0).numpy()
    mocap = np.load(os.path.join(ASSET_DIR, "mocap_walk_f32.npy"))
    norm = np.load(os.path.join(ASSET_DIR, "obs_norm_f32.npz"))
    return dict(
        kp=kp, kv=kv,
        action_high=np.asarray(ACTION_HIGH, dtype=np.float32),
        initial_dof_pos=np.asarray(INITIAL_DOF_POS, dtype=np.float32),
        mocap=np.ascontiguousarray(mocap, dtype=np.float32),
        obs_mean=norm["mean"].astype(np.float32), obs_var=norm["var"].astype(np.float32),
        dof_armature_nominal=np.asarray(ARMATURE, dtype=np.float32),
        dof_damping_nominal=np.full((33,), DOF_DAMPING, dtype=np.float32),
    )
