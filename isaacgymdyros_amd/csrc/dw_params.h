// dw_params.h -- DwConfig (C-ABI) -> the wave-uniform parameter block the kernels take by value.
#pragma once

#include "dw_task.h"

// which kernels DwConfig.pipeline = 0 selects
#define DW_DEFAULT_PIPELINE 3

namespace dw {

inline const char *check_config(const DwConfig *c) {
    if (c->num_envs <= 0) return "num_envs must be positive";
    if (c->control_freq_inv != 2) return "only controlFrequencyInv = 2 is supported (DyrosDynamicWalk.yaml:11)";
    if (!(c->dt > 0)) return "dt must be positive";
    if (c->solver_iterations < 1 || c->solver_iterations > 64) return "solver_iterations out of range";
    if (!(c->friction >= 0)) return "friction must be non-negative";
    if (c->terrain && (c->terrain_rows < 2 || c->terrain_cols < 2 || !(c->terrain_hscale > 0) || !(c->terrain_vscale > 0)))
        return "terrain: rows/cols >= 2 and positive scales required";
    if (c->terrain_curriculum && (c->terrain_num_levels < 1 || c->terrain_num_types < 1))
        return "terrain curriculum: num_levels and num_types must be positive";
    if (c->terrain_curriculum && c->terrain_num_levels > (1 << LVL_BITS))
        return "terrain_num_levels: at most 256 levels (the step kernel sums the levels of a wave's envs over 8 bit planes)";
    if (c->pipeline == 1 || c->pipeline == 2 || c->pipeline == 4)
        return "pipeline 1 (wave per env), 2 (quad) and 4 (lane per env, wave per limb) are retired: use 0 (default) or 3 (octet: 8 lanes per env)";
    if (c->pipeline != 0 && c->pipeline != 3) return "pipeline must be 0 (default) or 3 (octet: 8 lanes per env)";
    if (c->debug_wave_build < 0 || c->debug_wave_build > 3) return "debug_wave_build must be 0 (by launch size), 1, 2 (octet kernels: one / two waves per SIMD) or 3 (hex instantiation)";
    if (c->num_envs > (1 << 20)) return "num_envs above 2^20 per GPU (the kernels address the per-env streams with 32-bit byte offsets)";
    return nullptr;
}

inline TaskParams make_task_params(const DwConfig *c) {
    TaskParams t;
    memset(&t, 0, sizeof(t));
    t.phys.dt = (float)c->dt;
    for (int i = 0; i < 3; ++i) t.phys.g[i] = c->gravity[i];
    t.phys.iters = c->solver_iterations;
    t.phys.contact_offset = c->contact_offset;
    t.phys.max_depen = c->max_depenetration_velocity;
    t.phys.erp = c->erp;
    t.phys.cfm = c->contact_cfm;
    t.phys.pen_k = c->penalty_stiffness;
    t.phys.pen_c = c->penalty_damping;
    t.phys.max_ang_vel = c->max_angular_velocity;
    t.phys.vel_at_com = c->root_vel_at_com;
    t.phys.self_collision = c->self_collision;
    t.num_envs = c->num_envs;
    const double dtp = c->dt * c->control_freq_inv;          // python: self.dt * self.skipframe
    t.inv_dt_f = (float)(1.0 / c->dt);
    t.dt_policy_f = (float)dtp;
    t.clock_gain_f = (float)(5 * dtp);                       // 5*self.dt_policy
    t.pert_period_f = (float)(8 / dtp);                      // 8/self.dt_policy
    t.pert_dur_lo = (int)(0.1 / dtp);
    t.pert_dur_hi = (int)(1 / dtp);
    t.max_episode_length = c->max_episode_length;
    t.initial_height = c->initial_height;
    t.death_cost = c->death_cost;
    t.friction = c->friction;
    t.perturb = c->perturb;
    t.force_perturb_start = c->force_perturb_start;
    t.dr_dof = c->randomize_dof_on_reset;
    t.dr_friction = c->randomize_friction_on_reset;
    for (int i = 0; i < 2; ++i) {
        t.dr_damp[i] = c->dr_damping_add[i];
        t.dr_arm[i] = c->dr_armature_scale[i];
        t.dr_fric[i] = c->dr_friction_scale[i];
    }
    t.timeout_fix = c->timeout_fix;
    t.gpu_div = c->torch_gpu_div;
    t.freeze_physics = c->debug_freeze_physics;
    t.seed = c->seed;
    t.phys.hs = nullptr;                                      // set at bind
    t.phys.hmax = nullptr; t.phys.hm_cell = 1; t.phys.hm_rows = t.phys.hm_cols = 0;          // built at bind
    t.phys.t_rows = c->terrain_rows; t.phys.t_cols = c->terrain_cols;
    t.phys.t_inv_h = c->terrain ? 1.0f / c->terrain_hscale : 0.0f;
    t.phys.t_vs = c->terrain_vscale; t.phys.t_border = c->terrain_border;
    t.terrain_curriculum = c->terrain_curriculum;
    t.custom_origins = c->custom_origins;
    t.terrain_num_levels = c->terrain_num_levels; t.terrain_num_types = c->terrain_num_types;
    t.terrain_half_length = (float)(c->terrain_env_length / 2.0);
    t.max_episode_length_s = c->max_episode_length_s;
    return t;
}

inline void default_config(DwConfig *c) {
    memset(c, 0, sizeof(*c));
    c->dt = 0.002;
    c->num_envs = 4096;
    c->control_freq_inv = 2;
    c->gravity[0] = 0; c->gravity[1] = 0; c->gravity[2] = -9.81f;
    c->solver_iterations = 5;
    c->contact_offset = 0.002f;
    c->max_depenetration_velocity = 10.0f;
    c->friction = 1.0f;
    c->erp = 0.2f;
    c->contact_cfm = 1e-3f;
    c->penalty_stiffness = 1.0e5f;
    c->penalty_damping = 1.0e3f;
    c->max_angular_velocity = 100.0f;
    c->max_episode_length = 8000.0f;
    c->initial_height = 0.93f;
    c->death_cost = 0.0f;
    c->perturb = 1;
    c->randomize_dof_on_reset = 1;
    c->dr_damping_add[0] = 0.0f; c->dr_damping_add[1] = 2.9f;
    c->dr_armature_scale[0] = 0.8f; c->dr_armature_scale[1] = 1.2f;
    c->dr_friction_scale[0] = 0.7f; c->dr_friction_scale[1] = 1.3f;
    c->root_vel_at_com = 1;
    c->torch_gpu_div = 1;
    c->self_collision = 1;
    c->seed = 42;
}

inline const char *check_terrain_buffers(const DwConfig *c, const DwBuffers *b) {
    if (c->terrain && !b->height_samples) return "terrain configured but height_samples is null";
    if (c->terrain_curriculum && (!b->terrain_origins || !b->terrain_levels || !b->terrain_types))
        return "terrain curriculum configured but terrain_origins / terrain_levels / terrain_types is null";
    return nullptr;
}

inline const char *check_buffers(const DwBuffers *b, bool task) {
    if (!b->root_states || !b->dof_state || !b->contact_forces || !b->mass_scale || !b->dof_damping ||
        !b->dof_armature || !b->friction_scale)
        return "physics buffers missing (root_states, dof_state, contact_forces, mass_scale, dof_damping, dof_armature, friction_scale)";
    if (task && (!b->total_mass || !b->env_origins || !b->obs_buf || !b->rew_buf || !b->reset_buf || !b->progress_buf ||
                 !b->timeout_buf || !b->randomize_buf || !b->stacked_rewards || !b->env_state || !b->obs_history ||
                 !b->action_history || !b->gate_acc))
        return "task buffers missing";
    return nullptr;
}

}  // namespace dw
