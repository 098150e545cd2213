"""Calls the row f-3 entry points (include/dyros_walk.h: dw_amp_*, dw_newwalk_reward) of either library -- the CPU oracle with numpy
arrays or the HIP library with torch tensors on the GPU -- on the inputs of tests/golden/amp_lower_ref.npz."""
import ctypes as C

import numpy as np


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    assert a.is_contiguous()
    return C.c_void_p(a.data_ptr())


def run_all(api, g, to, empty, chk, early=True):
    """g: the fixture; to(array) -> device array; empty(shape, dtype) -> device array; chk(rc).  Returns dict of outputs."""
    N = g["root_states"].shape[0]
    a = {k: to(g[k]) for k in ("root_states", "rootvel_noise", "dof_pos", "dof_pos_bias", "quat_bias", "dof_vel", "dof_vel_pre", "commands",
                               "actions", "actions_pre", "motor_efforts", "total_mass", "contact_force", "rigid_body_pos", "rigid_body_rot",
                               "progress_buf")}
    out = {}
    out["obs"] = empty((N, 36), "f4")
    chk(api["amp_observations"](N, ptr(a["root_states"]), ptr(a["rootvel_noise"]), ptr(a["dof_pos"]), ptr(a["dof_pos_bias"]), ptr(a["quat_bias"]),
                                ptr(a["dof_vel"]), ptr(a["commands"]), ptr(out["obs"]), None))
    out["reward"], out["reward_values"] = empty((N,), "f4"), empty((N, 9), "f4")
    chk(api["amp_reward"](N, ptr(a["root_states"]), ptr(a["dof_vel"]), ptr(a["dof_vel_pre"]), ptr(a["commands"]), ptr(a["actions"]),
                          ptr(a["actions_pre"]), ptr(a["motor_efforts"]), ptr(a["contact_force"]), ptr(a["total_mass"]), ptr(out["reward"]),
                          ptr(out["reward_values"]), None))
    ids = to(g["contact_body_ids"].astype(np.int32))
    out["reset"], out["terminated"] = empty((N,), "i8"), empty((N,), "i8")
    chk(api["amp_reset"](N, ptr(a["progress_buf"]), ptr(a["contact_force"]), ptr(ids), 2, ptr(a["rigid_body_pos"]), ptr(a["rigid_body_rot"]),
                         8000.0, 1 if early else 0, 0.6, ptr(out["reset"]), ptr(out["terminated"]), None))
    b = {k: to(g[k]) for k in g.files if k.startswith("nw_") and g[k].dtype.kind in "fi"}
    nf = to(g["nw_non_feet_idxs"].astype(np.int32))
    out["nw_total"], out["nw_reset"], out["nw_reward8"] = empty((N,), "f4"), empty((N,), "i8"), empty((N, 8), "f4")
    chk(api["newwalk_reward"](N, ptr(b["nw_reset_buf"]), ptr(b["nw_progress_buf"]), ptr(b["nw_target_vel"]), ptr(b["nw_root_pose_states"]),
                              ptr(b["nw_joint_position_states"]), ptr(b["nw_joint_velocity_states"]), ptr(nf), len(g["nw_non_feet_idxs"]),
                              ptr(b["nw_contact_forces"]), g["nw_contact_forces"].shape[1], 0.6, -1.0, 1000.0, ptr(b["nw_q_nominal"]),
                              g["nw_q_nominal"].shape[0], ptr(b["nw_head_states"]), ptr(b["nw_lfoot_states"]), ptr(b["nw_rfoot_states"]),
                              ptr(b["nw_phase"]), ptr(out["nw_total"]), ptr(out["nw_reset"]), ptr(out["nw_reward8"]), None))
    return out
