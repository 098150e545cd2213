"""Eager-torch fp32 twin of the reference's observation / reward / termination functions.  TEST INFRASTRUCTURE.

Op-for-op restatement (same torch ops, same order, no TorchScript) of
  compute_humanoid_walk_observations   tasks/dyros_dynamic_walk.py:750-777 (+ quat2euler, python/isaacgym/torch_utils.py:227-273)
  compute_humanoid_walk_reward         tasks/dyros_dynamic_walk.py:802-947
  check_termination                    tasks/dyros_dynamic_walk.py:581-596 (+ quat_diff_rad, utils/torch_jit_utils.py:141-160)
so that it can run on whatever device torch runs on.  On the CPU it is bit-identical to the reference's own
functions (tests/test_torch_twin.py, checked live where /root/reference is mounted and against the goldens
elsewhere); on the GPU box it is "the reference's torch path at fp32" on that hardware, against which the HIP
kernels are compared (tests/test_hip_gpu.py::test_obs_reward_vs_torch_twin_on_gpu).
"""
import torch


def quat_mul(a, b):
    x1, y1, z1, w1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x2, y2, z2, w2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([x, y, z, w], dim=-1)


def quat_diff_rad(a, b):
    b_conj = torch.cat((-b[:, :3], b[:, -1:]), dim=-1)
    mul = quat_mul(a, b_conj)
    return 2.0 * torch.asin(torch.clamp(torch.norm(mul[:, 0:3], p=2, dim=-1), max=1.0))


def quat2euler(q):
    # the reference fills a zero [N,4,4] tensor and reads strided slices of it (torch_utils.py:227-269); strided
    # operands take torch's non-vectorised CPU loops, whose atan2 rounds differently from the vectorised one, so the
    # twin keeps the same memory layout
    mat = torch.zeros(q.shape[0], 4, 4, device=q.device)
    w, x, y, z = q[:, 3], q[:, 0], q[:, 1], q[:, 2]
    mat[:, 0, 0] = w * w + x * x - y * y - z * z
    mat[:, 0, 1] = 2 * x * y - 2 * w * z
    mat[:, 0, 2] = 2 * x * z + 2 * w * y
    mat[:, 1, 0] = 2 * x * y + 2 * w * z
    mat[:, 1, 1] = w * w - x * x + y * y - z * z
    mat[:, 1, 2] = 2 * y * z - 2 * w * x
    mat[:, 2, 0] = 2 * x * z - 2 * w * y
    mat[:, 2, 1] = 2 * y * z + 2 * w * x
    mat[:, 2, 2] = w * w - x * x - y * y + z * z
    mat[:, 3, 3] = 1
    EPS4 = float(2.220446049250313e-16) * 4
    cy = torch.sqrt(mat[:, 0, 0] * mat[:, 0, 0] + mat[:, 1, 0] * mat[:, 1, 0])
    cond = cy > EPS4
    ez = torch.where(cond, torch.atan2(mat[:, 1, 0], mat[:, 0, 0]), torch.atan2(-mat[:, 0, 1], mat[:, 1, 1]))
    ey = torch.where(cond, torch.atan2(-mat[:, 2, 0], cy), torch.atan2(-mat[:, 2, 0], cy))
    ex = torch.where(cond, torch.atan2(mat[:, 2, 1], mat[:, 2, 2]), torch.zeros_like(ey, dtype=torch.float))
    return ex, ey, ez


def observation(root_states, quat_bias, qpos_noise, qpos_bias, qvel_noise, time, init_mocap_data_idx, target_vel,
                vel_rand, obs_mean, obs_var):
    """Returns the normalised 37-d observation.  vel_rand: the U[0,1) draw of :766 (torch.rand(N, 6))."""
    pi = 3.14159265358979
    ex, ey, ez = quat2euler(root_states[:, 3:7].clone())
    ex = ex + quat_bias[:, 0]
    ey = ey + quat_bias[:, 1]
    ez = ez + quat_bias[:, 2]
    time2idx = (time % (3599 * 0.0005)) / 0.0005
    phase = (init_mocap_data_idx + time2idx) % 3599 / 3599
    sin_phase = torch.sin(2 * pi * phase)
    cos_phase = torch.cos(2 * pi * phase)
    vel_noise = vel_rand * 0.05 - 0.025
    obs = torch.cat((ex.unsqueeze(-1), ey.unsqueeze(-1), ez.unsqueeze(-1), qpos_noise[:, 0:12] + qpos_bias,
                     qvel_noise[:, 0:12], sin_phase.view(-1, 1), cos_phase.view(-1, 1), target_vel[:, 0].unsqueeze(-1),
                     target_vel[:, 1].unsqueeze(-1), root_states[:, 7:] + vel_noise), dim=-1)
    diff = obs - obs_mean
    return diff / torch.sqrt(obs_var + 1e-8 * torch.ones_like(obs_var))


def reward(root_pose_states, target_vel, joint_position_target, force_target, joint_position_states, joint_velocity_states,
           pre_joint_velocity_states, actions, actions_pre, contact_forces, lfoot_force_pre, rfoot_force_pre, mocap_data_idx,
           total_mass, non_feet_idxs, left_foot_idx, right_foot_idx, death_cost=0.0):
    """Returns (total_reward [N], stacked [N,14], foot_contact_reward [N], quat_error [N], collision [N] bool)."""
    torso_rot = root_pose_states[:, 3:7]
    identity_rot = torch.zeros_like(torso_rot)
    identity_rot[..., -1] = 1.
    quat_error = quat_diff_rad(identity_rot, torso_rot)
    r0 = 0.3 * torch.exp(-13.2 * torch.abs(quat_error))
    r1 = 0.35 * torch.exp(-2.0 * torch.norm((joint_position_target[:, 0:] - joint_position_states[:, 0:]), dim=1) ** 2)
    r2 = 0.05 * torch.exp(-0.01 * torch.norm((torch.zeros_like(joint_velocity_states) - joint_velocity_states[:, 0:]), dim=1) ** 2)
    lfoot_force = contact_forces[:, left_foot_idx, 0:3]
    rfoot_force = contact_forces[:, right_foot_idx, 0:3]
    policy_freq_scale = 1
    r9 = 0.2 * torch.exp(-0.01 * policy_freq_scale * (torch.norm(lfoot_force[:] - lfoot_force_pre[:], dim=1) +
                                                        torch.norm(rfoot_force[:] - rfoot_force_pre[:], dim=1)))
    r4 = 0.05 * torch.exp(-0.01 * torch.norm((actions[:, 0:-1]) * 333, dim=1))
    r5 = 0.6 * torch.exp(-0.01 * policy_freq_scale * torch.norm((actions[:, 0:-1] - actions_pre[:, 0:-1]) * 333, dim=1))
    r7 = 0.05 * torch.exp(-20.0 * torch.norm((joint_velocity_states[:, 0:] - pre_joint_velocity_states[:, 0:]), dim=1) ** 2)
    r6 = 0.3 * torch.exp(-3.0 * torch.norm((target_vel[:, 0:] - root_pose_states[:, 7:9]), dim=1) ** 2)
    left_foot_contact = (lfoot_force[:, 2].unsqueeze(-1) > 1.)
    right_foot_contact = (rfoot_force[:, 2].unsqueeze(-1) > 1.)
    ones = torch.ones_like(r6)
    zeros = torch.zeros_like(r6)
    DSP = (3300 <= mocap_data_idx) & (mocap_data_idx < 3600)
    DSP = DSP | (mocap_data_idx < 300)
    DSP = DSP | ((1500 <= mocap_data_idx) & (mocap_data_idx < 2100))
    RSSP = (300 <= mocap_data_idx) & (mocap_data_idx < 1500)
    LSSP = (2100 <= mocap_data_idx) & (mocap_data_idx < 3300)
    DSP_sync = DSP & right_foot_contact & left_foot_contact
    RSSP_sync = RSSP & right_foot_contact & ~left_foot_contact
    LSSP_sync = LSSP & ~right_foot_contact & left_foot_contact
    r8 = torch.zeros_like(r0, dtype=torch.float)
    feeder = 0.2 * torch.ones_like(r8, dtype=torch.float)
    r8 = torch.where(DSP_sync.squeeze(-1), feeder, r8)
    r8 = torch.where(RSSP_sync.squeeze(-1), feeder, r8)
    r8 = torch.where(LSSP_sync.squeeze(-1), feeder, r8)
    r10 = torch.zeros_like(r0, dtype=torch.float)
    thres = (lfoot_force[:, 2].unsqueeze(-1) > 1.4 * 9.81 * total_mass) | (rfoot_force[:, 2].unsqueeze(-1) > 1.4 * 9.81 * total_mass)
    r11 = torch.where(thres.squeeze(-1), -0.2 * ones[:], zeros[:])
    pen = 0.1 * torch.exp(-0.007 * (torch.norm(torch.clamp(lfoot_force[:, 2].unsqueeze(-1) - 1.4 * 9.81 * total_mass, min=0.0), dim=1)
                                     + torch.norm(torch.clamp(rfoot_force[:, 2].unsqueeze(-1) - 1.4 * 9.81 * total_mass, min=0.0), dim=1)))
    r3 = torch.where(thres.squeeze(-1), pen[:], 0.1 * ones[:])
    td = (torch.abs(lfoot_force[:, 2] - lfoot_force_pre[:, 2]).unsqueeze(-1) > 0.2 * 9.81 * total_mass / policy_freq_scale) | \
         (torch.abs(rfoot_force[:, 2] - rfoot_force_pre[:, 2]).unsqueeze(-1) > 0.2 * 9.81 * total_mass / policy_freq_scale)
    r12 = torch.where(td.squeeze(-1), -0.05 * ones[:], zeros[:])
    weight_scale = total_mass / 104.48
    r13 = 0.1 * torch.exp(-0.001 * (torch.abs(lfoot_force[:, 2] + weight_scale.squeeze(-1) * force_target[:, 0]))) + \
        0.1 * torch.exp(-0.001 * (torch.abs(rfoot_force[:, 2] + weight_scale.squeeze(-1) * force_target[:, 1])))
    stacked = torch.stack([r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13], 1)
    total = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11 + r12 + r13
    collision = torch.any(torch.norm(contact_forces[:, non_feet_idxs, :], dim=2) > 1., dim=1)
    total = torch.where(collision, torch.ones_like(total) * death_cost, total)
    total = torch.where(torch.abs(quat_error) > 0.5, torch.ones_like(total) * death_cost, total)
    stacked = torch.where(collision.unsqueeze(-1), torch.ones_like(stacked) * death_cost, stacked)
    return total, stacked, r8, quat_error, collision
