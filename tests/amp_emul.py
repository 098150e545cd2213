"""The fused TocabiAMPLower step and reset on the HOST emulation (tests/emul/dw_emul_amp.inc over the octet emulation's handle): the
same kernel source as the HIP library, numpy buffers.  Test infrastructure; mirrors the buffer table the host class builds
(isaacgymdyros_amd/tocabi_amp_lower.py::_fused_tables)."""
import ctypes as C

import numpy as np

from isaacgymdyros_amd import abi
from isaacgymdyros_amd.tocabi_amp_lower import ACTION_HIGH, ARMATURE, D_GAINS, INIT_ANGLE, P_GAINS

NUM_OBS, NUM_ACT, AW = 36, 12, 34


def amp_shapes(N, num_his, num_skip, log_slots, amp_steps):
    NH = num_his * num_skip
    num_obs = (NUM_OBS + NUM_ACT) * num_his - NUM_ACT
    f, i8, u1, i4 = "f4", "i8", "u1", "i4"
    return {"actions": ((N, 12), f), "actions_pre": ((N, 12), f), "action_history": ((N, NH * 12), f), "obs_history": ((N, NH * NUM_OBS), f),
            "commands": ((N, 3), f), "start_target_vel": ((N, 3), f), "final_target_vel": ((N, 3), f), "vel_change_duration": ((N,), i8),
            "cur_vel_change_duration": ((N,), i8), "epi_len": ((N,), f), "power_scale": ((N, 12), f), "action_log": ((N, log_slots, 12), f),
            "delay_idx": ((N,), i8), "simul_len": ((N,), i8), "qpos_noise": ((N, 33), f), "qvel_noise": ((N, 33), f), "qpos_pre": ((N, 33), f),
            "qpos_bias": ((N, 12), f), "quat_bias": ((N, 3), f), "dof_vel_pre": ((N, 33), f), "tau": ((N, 33), f), "progress_buf": ((N,), i8),
            "randomize_buf": ((N,), i8), "reset_buf": ((N,), i8), "terminate_buf": ((N,), i8), "timeout_buf": ((N,), u1),
            "rigid_body_pos": ((N, 38, 3), f), "rigid_body_rot": ((N, 38, 4), f), "foot_pos": ((N, 2, 3), f), "obs1": ((N, NUM_OBS), f),
            "obs_buf": ((N, num_obs), f), "obs_out": ((N, num_obs), f), "rew_buf": ((N,), f), "reward_values": ((N, 9), f),
            "total_mass": ((N,), f), "amp_obs_buf": ((N, amp_steps, AW), f), "amp_obs1": ((N, AW), f), "motor_efforts": ((12,), f),
            "p_gains": ((33,), f), "d_gains": ((33,), f), "init_angle": ((33,), f), "pd_action_offset": ((12,), f), "pd_action_scale": ((12,), f),
            "epi_len_log": ((N,), f), "perturbation_count": ((N,), i8), "perturb_timing": ((N,), i8), "pert_on": ((N,), u1),
            "initial_root_states": ((N, 13), f), "hist_head": ((N, 2), i4), "draw_ctr": ((N,), i8), "nominal_damping": ((33,), f),
            "nominal_armature": ((33,), f)}


class AmpEmul:
    """N envs of the fused task on an emulation (or any) backend `sim` (an OracleSim-like object: .h, .api, .buf, .simulate)."""

    def __init__(self, sim, N, seed=7, num_his=10, num_skip=2, amp_steps=2, dt=0.002, hist_ring=True, device_draws=True, noise=True,
                 vel_change=True, pd_control=False, randomize=True, episode_length=40.0, spawn_z=0.93):
        self.sim, self.N = sim, N
        self.log_slots = round(0.01 / dt) + 1
        self.K = 2
        c = self.c = abi.DwAmpConfig()
        c.num_envs, c.num_his, c.num_skip, c.log_slots, c.amp_steps = N, num_his, num_skip, self.log_slots, amp_steps
        c.pd_control, c.noise, c.vel_change, c.local_root_obs, c.enable_early_termination = int(pd_control), int(noise), int(vel_change), 0, 1
        c.clip_actions, c.clip_obs, c.max_episode_length, c.termination_height = 1.0, 5.0, episode_length, 0.6
        c.inv_dt, c.dt, c.gpu_div = float(np.float32(1.0 / dt)), dt, 1
        for i, (lo, hi) in enumerate(((-0.2, 0.6), (-0.2, 0.2), (-0.4, 0.4))):
            c.cmd_lo[i], c.cmd_scale[i] = lo, hi - lo
        c.hist_ring, c.device_draws, c.randomize = int(hist_ring), int(device_draws), int(randomize)
        c.dr_damping, c.dr_armature, c.dr_frequency = 1, 1, 1
        c.dr_damping_range[0], c.dr_damping_range[1] = 0.0, 2.9
        c.dr_armature_range[0], c.dr_armature_range[1] = 0.8, 1.2
        c.delay_idx_range[0], c.delay_idx_range[1] = 1 + int(0.002 / dt), 1 + round(0.01 / dt)
        c.seed = seed
        self.a = {n: np.zeros(s, dtype=d) for n, (s, d) in amp_shapes(N, num_his, num_skip, self.log_slots, amp_steps).items()}
        a = self.a
        a["motor_efforts"][:] = ACTION_HIGH[:12]
        a["p_gains"][:] = [p / 9.0 for p in P_GAINS]
        a["d_gains"][:] = [d / 3.0 for d in D_GAINS]
        a["init_angle"][:] = INIT_ANGLE
        a["nominal_damping"][:] = 0.1
        a["nominal_armature"][:] = ARMATURE
        a["power_scale"][:] = 1.0
        a["initial_root_states"][:, 2] = spawn_z
        a["initial_root_states"][:, 6] = 1.0
        a["rigid_body_rot"][..., 3] = 1.0
        a["reset_buf"][:] = 1
        a["terminate_buf"][:] = 1
        a["delay_idx"][:] = 1
        a["perturb_timing"][:] = 1
        a["total_mass"][:] = sim.buf["total_mass"].reshape(-1)[:N] if "total_mass" in sim.buf else 100.0
        b = self.b = abi.DwAmpBuffers()
        for n in abi.AMP_BUFFER_NAMES:
            setattr(b, n, a[n].ctypes.data)
        sim.buf["root_states"][...] = a["initial_root_states"]
        sim.buf["dof_state"][..., 0] = a["init_angle"]
        sim.buf["dof_state"][..., 1] = 0.0
        sim.buf["dof_damping"][...] = a["nominal_damping"]
        sim.buf["dof_armature"][...] = a["nominal_armature"]

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.sim.api["last_error"]().decode())

    def reset_done(self, draws=None):
        ids = np.nonzero(self.a["reset_buf"])[0]
        if isinstance(draws, dict):          # numpy arrays by name -> the C struct (kept alive until the call returns)
            d = abi.DwAmpResetDraws()
            for n in abi.AMP_RESET_DRAW_NAMES:
                setattr(d, n, draws[n].ctypes.data)
            draws = C.byref(d)
        self._chk(self.sim.api["amp_reset_done"](self.sim.h, C.byref(self.c), C.byref(self.b), draws, None))
        return ids

    def step(self, actions, z=(None, None), rootvel_noise=None, ramp=(None, None), sync_physics=None):
        api, h, c, b = self.sim.api, self.sim.h, C.byref(self.c), C.byref(self.b)
        p = lambda t: None if t is None else t.ctypes.data
        actions = np.ascontiguousarray(actions, dtype=np.float32)
        self._chk(api["amp_step_begin"](h, c, b, actions.ctypes.data, p(ramp[0]), p(ramp[1]), None))
        for k in range(self.K):
            self.sim.simulate(self.a["tau"])
            if k + 1 < self.K:
                self._chk(api["amp_step_mid"](h, c, b, p(z[k]), k + 1, None))
        self._chk(api["amp_step_end"](h, c, b, p(z[self.K - 1]), self.K - 1, p(rootvel_noise), None))

    def history_linear(self):
        NH = self.c.num_his * self.c.num_skip
        out = []
        for k, (name, w) in enumerate((("action_history", 12), ("obs_history", NUM_OBS))):
            t = self.a[name].reshape(self.N, NH, w)
            if self.c.hist_ring:
                idx = (self.a["hist_head"][:, k].astype(np.int64)[:, None] + np.arange(NH)[None, :]) % NH
                t = np.take_along_axis(t, idx[:, :, None], axis=1)
            out.append(t.reshape(self.N, NH * w).copy())
        return out
