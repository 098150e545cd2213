/* dw_amp.h -- CPU oracle of the sibling TOCABI tasks' env-side functions (row f-3).  TEST INFRASTRUCTURE ONLY (dw_amp.c). */
#ifndef DW_AMP_ORACLE_H
#define DW_AMP_ORACLE_H
#include <stdint.h>
#include "../include/dyros_walk.h"

#ifdef __cplusplus
extern "C" {
#endif
int dwo_amp_observations(int n, const float *root_states, const float *rootvel_noise, const float *dof_pos,
                          const float *dof_pos_bias, const float *quat_bias, const float *dof_vel, const float *commands,
                          float *obs, void *stream);
int dwo_amp_disc_observations(int n, const float *root_states, const float *dof_pos, const float *dof_vel, int dof_row_stride,
                               int dof_elem_stride, int local_root_obs, const float *key_pos, int n_key, float *obs, void *stream);
int dwo_amp_reward(int n, const float *root_states, const float *dof_vel, const float *dof_vel_pre, const float *commands,
                    const float *actions, const float *actions_pre, const float *motor_efforts, const float *contact_force,
                    const float *total_mass, float *reward, float *reward_values, void *stream);
int dwo_amp_reset(int n, const int64_t *progress_buf, const float *contact_buf, const int32_t *contact_body_ids, int n_contact_ids,
                   const float *rigid_body_pos, const float *rigid_body_rot, float max_episode_length, int enable_early_termination,
                   float termination_height, int64_t *reset, int64_t *terminated, void *stream);
int dwo_newwalk_reward(int n, const int64_t *reset_buf, const int64_t *progress_buf, const float *target_vel,
                        const float *root_pose_states, const float *joint_position_states, const float *joint_velocity_states,
                        const int32_t *non_feet_idxs, int n_non_feet, const float *contact_forces, int num_bodies,
                        float termination_height, float death_cost, float max_episode_length, const float *q_nominal, int num_dof,
                        const float *head_states, const float *lfoot_states, const float *rfoot_states, const float *phase_in,
                        float *total_reward, int64_t *reset, float *reward8, void *stream);
int dwo_body_positions(DwHandle *h, const int32_t *moving_bodies, int nb, float *out, void *stream);
#ifdef __cplusplus
}
#endif
#endif
