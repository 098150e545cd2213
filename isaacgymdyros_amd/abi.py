"""ctypes mirror of include/dyros_walk.h (structs, constants, function prototypes).

The reference binds its native engine through pybind11 (`gym_3x.so`) and aliases sim-owned buffers
with `gymtorch.wrap_tensor` (reference: python/isaacgym/gymtorch.py:61-106).  Here buffers are
torch-owned and the native side only ever sees raw pointers, so plain ctypes is the whole binding.
"""
from __future__ import annotations

import ctypes as C
import re
import os

from .model import DwModel, DwGeom  # noqa: F401  (re-exported)

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dyros_walk.h")


def _parse_defines(path):
    out = {}
    with open(path) as f:
        for line in f:
            m = re.match(r"#define\s+(DW_[A-Z0-9_]+)\s+(-?\d+)\b", line)
            if m:
                out[m.group(1)] = int(m.group(2))
    return out


K = _parse_defines(HEADER)          # every integer #define of the header, e.g. K["DW_ES_WORDS"]
globals().update(K)


class DwTaskConst(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ("kp", "kv", "action_high", "initial_dof_pos", "mocap", "obs_mean", "obs_var",
                 "dof_armature_nominal", "dof_damping_nominal")]


class DwConfig(C.Structure):
    _fields_ = [
        ("dt", C.c_double),
        ("num_envs", C.c_int32),
        ("control_freq_inv", C.c_int32),
        ("gravity", C.c_float * 3),
        ("solver_iterations", C.c_int32),
        ("contact_offset", C.c_float),
        ("max_depenetration_velocity", C.c_float),
        ("friction", C.c_float),
        ("erp", C.c_float),
        ("contact_cfm", C.c_float),
        ("penalty_stiffness", C.c_float),
        ("penalty_damping", C.c_float),
        ("max_angular_velocity", C.c_float),
        ("max_episode_length", C.c_float),
        ("initial_height", C.c_float),
        ("death_cost", C.c_float),
        ("perturb", C.c_int32),
        ("force_perturb_start", C.c_int32),
        ("randomize_dof_on_reset", C.c_int32),
        ("dr_damping_add", C.c_float * 2),
        ("dr_armature_scale", C.c_float * 2),
        ("randomize_friction_on_reset", C.c_int32),
        ("dr_friction_scale", C.c_float * 2),
        ("timeout_fix", C.c_int32),
        ("root_vel_at_com", C.c_int32),
        ("torch_gpu_div", C.c_int32),
        ("self_collision", C.c_int32),
        ("debug_freeze_physics", C.c_int32),
        ("seed", C.c_uint64),
        ("terrain", C.c_int32),
        ("terrain_rows", C.c_int32),
        ("terrain_cols", C.c_int32),
        ("terrain_hscale", C.c_float),
        ("terrain_vscale", C.c_float),
        ("terrain_border", C.c_float),
        ("terrain_curriculum", C.c_int32),
        ("terrain_num_levels", C.c_int32),
        ("terrain_num_types", C.c_int32),
        ("terrain_env_length", C.c_float),
        ("max_episode_length_s", C.c_float),
        ("custom_origins", C.c_int32),
        ("pipeline", C.c_int32),
        ("debug_wave_build", C.c_int32),
    ]


_F = C.c_void_p   # device (or, for the oracle, host) pointers travel as plain addresses


class DwBuffers(C.Structure):
    _fields_ = [(n, _F) for n in (
        "root_states", "dof_state", "contact_forces",
        "mass_scale", "dof_damping", "dof_armature", "friction_scale", "total_mass", "env_origins",
        "obs_buf", "rew_buf", "reset_buf", "progress_buf", "timeout_buf", "randomize_buf",
        "stacked_rewards", "env_state", "obs_history", "action_history", "gate_acc",
        "height_samples", "terrain_origins", "terrain_levels", "terrain_types")]


BUFFER_NAMES = [n for n, _ in DwBuffers._fields_]


AMP_BUFFER_NAMES = ["actions", "actions_pre", "action_history", "obs_history", "commands", "start_target_vel", "final_target_vel",
                    "vel_change_duration", "cur_vel_change_duration", "epi_len", "power_scale", "action_log", "delay_idx", "simul_len",
                    "qpos_noise", "qvel_noise", "qpos_pre", "qpos_bias", "quat_bias", "dof_vel_pre", "tau", "progress_buf", "randomize_buf",
                    "reset_buf", "terminate_buf", "timeout_buf", "rigid_body_pos", "rigid_body_rot", "foot_pos", "obs1", "obs_buf", "obs_out",
                    "rew_buf", "reward_values", "total_mass", "amp_obs_buf", "amp_obs1", "motor_efforts", "p_gains", "d_gains", "init_angle",
                    "pd_action_offset", "pd_action_scale", "epi_len_log", "perturbation_count", "perturb_timing", "pert_on", "initial_root_states",
                    "hist_head", "draw_ctr", "nominal_damping", "nominal_armature"]


class DwAmpBuffers(C.Structure):            # include/dyros_walk.h, the fused TocabiAMPLower step
    _fields_ = [(n, C.c_void_p) for n in AMP_BUFFER_NAMES]


class DwAmpConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_envs", "num_his", "num_skip", "log_slots", "amp_steps", "pd_control", "noise", "vel_change",
                                         "local_root_obs", "enable_early_termination")] + \
               [(n, C.c_float) for n in ("clip_actions", "clip_obs", "max_episode_length", "termination_height", "inv_dt", "dt")] + \
               [("gpu_div", C.c_int32), ("cmd_lo", C.c_float * 3), ("cmd_scale", C.c_float * 3)] + \
               [(n, C.c_int32) for n in ("hist_ring", "device_draws", "randomize", "dr_damping", "dr_armature", "dr_frequency")] + \
               [("dr_damping_range", C.c_float * 2), ("dr_armature_range", C.c_float * 2), ("delay_idx_range", C.c_int32 * 2), ("seed", C.c_uint64)]


AMP_RESET_DRAW_NAMES = ["power_scale_u", "rootvel_noise", "cmd_x_u", "cmd_y_u", "cmd_yaw_u", "qpos_bias_u", "quat_bias_u", "damping_u", "armature_u",
                        "perturb_timing", "delay_idx"]


class DwAmpResetDraws(C.Structure):         # include/dyros_walk.h: the caller's draws of dw_amp_reset_done, rows indexed by env
    _fields_ = [(n, C.c_void_p) for n in AMP_RESET_DRAW_NAMES]


def declare(lib: C.CDLL, prefix: str = "dw_"):
    """Attach argtypes/restype for every entry point the header declares; raises AttributeError if the
    shared object lacks one of them."""
    def fn(name, restype, *argtypes):
        f = getattr(lib, prefix + name)
        f.restype = restype
        f.argtypes = list(argtypes)
        return f
    H = C.c_void_p
    api = {}
    api["abi_version"] = fn("abi_version", C.c_int)
    api["last_error"] = fn("last_error", C.c_char_p)
    api["default_config"] = fn("default_config", None, C.POINTER(DwConfig))
    api["create"] = fn("create", C.c_int, C.POINTER(DwConfig), C.POINTER(DwModel), C.POINTER(DwTaskConst),
                       C.POINTER(H))
    api["destroy"] = fn("destroy", C.c_int, H)
    api["bind"] = fn("bind", C.c_int, H, C.POINTER(DwBuffers))
    api["simulate"] = fn("simulate", C.c_int, H, C.c_void_p, C.c_void_p, C.c_void_p)
    api["step"] = fn("step", C.c_int, H, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    api["step_dev"] = fn("step_dev", C.c_int, H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    api["reset_idx"] = fn("reset_idx", C.c_int, H, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p)
    P = C.c_void_p

    def fused_amp():
        # the fused TocabiAMPLower step and reset (csrc/dw_amp_step.h)
        AB, AC = C.POINTER(DwAmpBuffers), C.POINTER(DwAmpConfig)
        api["amp_step_begin"] = fn("amp_step_begin", C.c_int, H, AC, AB, P, P, P, P)
        api["amp_step_mid"] = fn("amp_step_mid", C.c_int, H, AC, AB, P, C.c_int, P)
        api["amp_step_end"] = fn("amp_step_end", C.c_int, H, AC, AB, P, C.c_int, P, P)
        if prefix == "dw_":          # (the one-launch step lives with the octet kernels' entry points: HIP library only)
            api["amp_step"] = fn("amp_step", C.c_int, H, AC, AB, P, P, P, C.POINTER(C.c_void_p), C.c_int, P, P)
            api["amp_reset_ids"] = fn("amp_reset_ids", C.c_int, P, C.c_int, P, P, P, P)
        api["amp_reset_rows"] = fn("amp_reset_rows", C.c_int, H, AC, AB, P, C.c_int, P, P, P, P, P, P, P, P, P, P)
        api["amp_reset_done"] = fn("amp_reset_done", C.c_int, H, AC, AB, C.POINTER(DwAmpResetDraws), P)

    if prefix in ("dw_", "dwe_"):
        api["terrain_log"] = fn("terrain_log", C.c_int, H, C.c_void_p, C.c_void_p)
    if prefix == "dwe_":        # (the host emulation of the kernels, tests/emul/: the octet library also carries the fused AMP step)
        if hasattr(lib, "dwe_amp_step_begin"):
            fused_amp()
        return api
    api["step_obs"] = fn("step_obs", C.c_int, H, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p)
    # row f-3: env-side functions of the sibling TOCABI tasks (device pointers as c_void_p, trailing stream)
    api["amp_observations"] = fn("amp_observations", C.c_int, C.c_int, P, P, P, P, P, P, P, P, P)
    api["amp_disc_observations"] = fn("amp_disc_observations", C.c_int, C.c_int, P, P, P, C.c_int, C.c_int, C.c_int, P, C.c_int, P, P)
    api["amp_reward"] = fn("amp_reward", C.c_int, C.c_int, P, P, P, P, P, P, P, P, P, P, P, P)
    api["amp_reset"] = fn("amp_reset", C.c_int, C.c_int, P, P, P, C.c_int, P, P, C.c_float, C.c_int, C.c_float, P, P, P)
    api["newwalk_reward"] = fn("newwalk_reward", C.c_int, C.c_int, P, P, P, P, P, P, P, C.c_int, P, C.c_int, C.c_float, C.c_float,
                               C.c_float, P, C.c_int, P, P, P, P, P, P, P, P)
    api["body_positions"] = fn("body_positions", C.c_int, H, C.POINTER(C.c_int32), C.c_int, P, P)
    if prefix == "dw_":         # (HIP library; the C oracle does not carry the fused step -- its checkers are the torch class and the emulation)
        fused_amp()
    return api


EXPORTS = ["abi_version", "last_error", "default_config", "create", "destroy", "bind", "simulate", "step", "step_dev", "step_obs",
           "terrain_log", "reset_idx", "amp_observations", "amp_disc_observations", "amp_reward", "amp_reset", "newwalk_reward", "body_positions",
           "amp_step_begin", "amp_step_mid", "amp_step_end", "amp_step", "amp_reset_rows", "amp_reset_done", "amp_reset_ids"]


# name -> (per-env shape, numpy dtype string); gate_acc is the one buffer without an env dimension
BUFFER_SPECS = {
    "root_states": ((13,), "f4"),
    "dof_state": ((K["DW_NUM_DOF"], 2), "f4"),
    "contact_forces": ((K["DW_NUM_BODIES"], 3), "f4"),
    "mass_scale": ((K["DW_NUM_BODIES"],), "f4"),
    "dof_damping": ((K["DW_NUM_DOF"],), "f4"),
    "dof_armature": ((K["DW_NUM_DOF"],), "f4"),
    "friction_scale": ((), "f4"),
    "total_mass": ((), "f4"),
    "env_origins": ((3,), "f4"),
    "obs_buf": ((K["DW_NUM_OBS"],), "f4"),
    "rew_buf": ((), "f4"),
    "reset_buf": ((), "i8"),
    "progress_buf": ((), "i8"),
    "timeout_buf": ((), "i8"),
    "randomize_buf": ((), "i8"),
    "stacked_rewards": ((K["DW_NUM_REW"],), "f4"),
    "env_state": ((K["DW_ES_WORDS"],), "f4"),
    "obs_history": ((K["DW_HIST_SLOTS"], K["DW_NUM_OBS1"]), "f4"),
    "action_history": ((K["DW_HIST_SLOTS"], K["DW_NUM_ACT"]), "f4"),
    "gate_acc": (None, "i8"),
    # terrain (row f-4): two tables shared by all envs (one-element placeholders on the ground plane) and two per-env words
    "height_samples": (None, "i2"),
    "terrain_origins": (None, "f4"),
    "terrain_levels": ((), "i8"),
    "terrain_types": ((), "i8"),
}
GATE_ACC_WORDS = K["DW_GATE_WORDS"]
# element counts of the buffers that are not per-env (shape None above); the terrain tables are re-allocated by the
# host class when a height field is configured
GLOBAL_WORDS = {"gate_acc": GATE_ACC_WORDS, "height_samples": 4, "terrain_origins": 3}

# env-state record fields: name -> (word offset, shape, 'f' float32 | 'i' int32); see DW_ES_* in the header
ES_FIELDS = {
    "qpos_noise": (K["DW_ES_QPOS_NOISE"], (33,), "f"),
    "qvel_noise": (K["DW_ES_QVEL_NOISE"], (33,), "f"),
    "qpos_pre": (K["DW_ES_QPOS_PRE"], (33,), "f"),
    "pre_joint_velocity_states": (K["DW_ES_PRE_QVEL"], (33,), "f"),
    "target_data_qpos": (K["DW_ES_TARGET_QPOS"], (33,), "f"),
    "target_data_force": (K["DW_ES_TARGET_FORCE"], (2,), "f"),
    "target_vel": (K["DW_ES_TARGET_VEL"], (2,), "f"),
    "motor_constant_scale": (K["DW_ES_MOTOR_SCALE"], (12,), "f"),
    "qpos_bias": (K["DW_ES_QPOS_BIAS"], (12,), "f"),
    "quat_bias": (K["DW_ES_QUAT_BIAS"], (3,), "f"),
    "action_log": (K["DW_ES_ACTION_LOG"], (6, 12), "f"),
    "actions": (K["DW_ES_ACTIONS"], (13,), "f"),
    "actions_pre": (K["DW_ES_ACTIONS_PRE"], (13,), "f"),
    "action_torque": (K["DW_ES_ACTION_TORQUE"], (12,), "f"),
    "action_torque_pre": (K["DW_ES_ACTION_TORQUE_PRE"], (12,), "f"),
    "foot_force_pre": (K["DW_ES_FOOT_FORCE_PRE"], (2, 3), "f"),
    "time": (K["DW_ES_TIME"], (1,), "f"),
    "epi_len": (K["DW_ES_EPI_LEN"], (), "f"),
    "epi_len_log": (K["DW_ES_EPI_LEN_LOG"], (), "f"),
    "contact_reward_sum": (K["DW_ES_CRS"], (), "f"),
    "contact_reward_mean": (K["DW_ES_CRM"], (), "f"),
    "magnitude": (K["DW_ES_MAGNITUDE"], (), "f"),
    "phase": (K["DW_ES_PHASE"], (), "f"),
    "init_mocap_data_idx": (K["DW_ES_INIT_MOCAP"], (1,), "i"),
    "mocap_data_idx": (K["DW_ES_MOCAP_IDX"], (1,), "i"),
    "delay_idx": (K["DW_ES_DELAY_IDX"], (), "i"),
    "simul_len": (K["DW_ES_SIMUL_LEN"], (), "i"),
    "perturbation_count": (K["DW_ES_PERT_COUNT"], (), "i"),
    "pert_duration": (K["DW_ES_PERT_DURATION"], (), "i"),
    "pert_on": (K["DW_ES_PERT_ON"], (), "i"),
    "impulse": (K["DW_ES_IMPULSE"], (), "i"),
    "perturb_timing": (K["DW_ES_PERT_TIMING"], (), "i"),
    "perturb_start": (K["DW_ES_PERT_START"], (1,), "i"),
    "hist_head": (K["DW_ES_HIST_HEAD"], (), "i"),
    "nan_resets": (K["DW_ES_NAN_RESETS"], (), "i"),
    "warm_impulses": (K["DW_ES_WARM"], (8, 3), "f"),
    "episode_return": (K["DW_ES_EPI_RETURN"], (), "f"),
    "last_episode_return": (K["DW_ES_LAST_RETURN"], (), "f"),
    "episodes_finished": (K["DW_ES_EPISODES"], (), "i"),
}


def es_view(env_state, name):
    """View of one named field of the [N, DW_ES_WORDS] record array (numpy array or torch tensor)."""
    off, shape, kind = ES_FIELDS[name]
    n = 1
    for s in shape:
        n *= s
    v = env_state[:, off:off + n]
    if kind == "i":
        v = v.view(_int32_of(env_state))
    return v.reshape((env_state.shape[0],) + tuple(shape))


def _int32_of(arr):
    try:
        import torch
        if isinstance(arr, torch.Tensor):
            return torch.int32
    except ImportError:
        pass
    import numpy as np
    return np.int32
