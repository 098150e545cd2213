// dw_amp.hip -- gfx950 entry points of the sibling TOCABI tasks' env-side functions (bodies:
// dw_amp.h).  One thread per env: these are 36-word observations and nine reward terms on state the physics kernels left in
// the Gym tensors -- a few hundred bytes per env, HBM-bound, one coalesced pass.  The stateless functions mirror the
// reference's TorchScript signatures (tasks/amp/tocabi_amp_lower_base.py:918-1069, tasks/tocabi_new_walk.py:384-496) so
// that a maintainer binds them where the reference calls its own; dw_body_positions stands in for the rows of
// acquire_rigid_body_state_tensor those functions read.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "dw_handle.h"
#include "dw_amp.h"

extern "C" __attribute__((visibility("hidden"))) void dw_set_error(int code, const char *msg);          // dw_hip.hip: the thread's dw_last_error()

namespace {

constexpr int TPB = 128;
int blocks(int n) { return (n + TPB - 1) / TPB; }
int fail(int code, const char *msg) { dw_set_error(code, msg); return code; }
int launched(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return DW_OK;
    char m[256];
    snprintf(m, sizeof m, "%s: %s", what, hipGetErrorString(e));
    return fail(DW_EHIP, m);
}

__global__ __launch_bounds__(TPB) void dw_k_amp_observations(const dwa::ObsArgs A) {
    const int e = (int)(blockIdx.x * TPB + threadIdx.x);
    if (e < A.n) dwa::observations(A, e);
}
__global__ __launch_bounds__(TPB) void dw_k_amp_disc_observations(const dwa::DiscObsArgs A) {
    const int e = (int)(blockIdx.x * TPB + threadIdx.x);
    if (e < A.n) dwa::disc_observations(A, e);
}
__global__ __launch_bounds__(TPB) void dw_k_amp_reward(const dwa::RewardArgs A) {
    const int e = (int)(blockIdx.x * TPB + threadIdx.x);
    if (e < A.n) dwa::reward(A, e);
}
__global__ __launch_bounds__(TPB) void dw_k_amp_reset(const dwa::ResetArgs A) {
    const int e = (int)(blockIdx.x * TPB + threadIdx.x);
    if (e < A.n) dwa::reset(A, e);
}
__global__ __launch_bounds__(TPB) void dw_k_newwalk_reward(const dwa::NewWalkArgs A) {
    const int e = (int)(blockIdx.x * TPB + threadIdx.x);
    if (e < A.n) dwa::newwalk_reward(A, e);
}
struct BodyList { int32_t b[DW_MAX_BODY_QUERY]; int nb; };
__global__ __launch_bounds__(TPB) void dw_k_body_positions(const dw::DevModel *__restrict__ M, const float *root_states, const float *dof_state,
                                                           const BodyList bodies, int n, float *out) {
    const int i = (int)(blockIdx.x * TPB + threadIdx.x);
    if (i >= n * bodies.nb) return;
    const int e = i / bodies.nb, k = i - bodies.nb * e;
    dwa::body_position(*M, root_states, dof_state, e, bodies.b[k], out + 3 * (size_t)i);
}

}  // namespace

extern "C" {

int dw_amp_observations(int n, const float *root_states, const float *rootvel_noise, const float *dof_pos, const float *dof_pos_bias,
                        const float *quat_bias, const float *dof_vel, const float *commands, float *obs, void *stream) {
    if (n <= 0 || !root_states || !rootvel_noise || !dof_pos || !dof_pos_bias || !quat_bias || !dof_vel || !commands || !obs)
        return fail(DW_EINVAL, "dw_amp_observations: null argument or n <= 0");
    const dwa::ObsArgs A{n, root_states, rootvel_noise, dof_pos, dof_pos_bias, quat_bias, dof_vel, commands, obs};
    hipLaunchKernelGGL(dw_k_amp_observations, dim3(blocks(n)), dim3(TPB), 0, (hipStream_t)stream, A);
    return launched("dw_amp_observations: launch");
}

int dw_amp_disc_observations(int n, const float *root_states, const float *dof_pos, const float *dof_vel, int dof_row_stride,
                             int dof_elem_stride, int local_root_obs, const float *key_pos, int n_key, float *obs, void *stream) {
    if (n <= 0 || !root_states || !dof_pos || !dof_vel || !key_pos || !obs) return fail(DW_EINVAL, "dw_amp_disc_observations: null argument or n <= 0");
    if (n_key < 1 || n_key > DW_MAX_BODY_QUERY) return fail(DW_EINVAL, "dw_amp_disc_observations: n_key must be 1..DW_MAX_BODY_QUERY");
    if (dof_elem_stride < 1 || dof_row_stride < 12 * dof_elem_stride - (dof_elem_stride - 1))
        return fail(DW_EINVAL, "dw_amp_disc_observations: a row must hold 12 dofs at the element stride given");
    const dwa::DiscObsArgs A{n, root_states, dof_pos, dof_vel, dof_row_stride, dof_elem_stride, local_root_obs, key_pos, n_key, obs};
    hipLaunchKernelGGL(dw_k_amp_disc_observations, dim3(blocks(n)), dim3(TPB), 0, (hipStream_t)stream, A);
    return launched("dw_amp_disc_observations: launch");
}

int dw_amp_reward(int n, const float *root_states, const float *dof_vel, const float *dof_vel_pre, const float *commands,
                  const float *actions, const float *actions_pre, const float *motor_efforts, const float *contact_force,
                  const float *total_mass, float *reward, float *reward_values, void *stream) {
    if (n <= 0 || !root_states || !dof_vel || !dof_vel_pre || !commands || !actions || !actions_pre || !motor_efforts || !contact_force ||
        !total_mass || !reward || !reward_values)
        return fail(DW_EINVAL, "dw_amp_reward: null argument or n <= 0");
    const dwa::RewardArgs A{n, root_states, dof_vel, dof_vel_pre, commands, actions, actions_pre, motor_efforts, contact_force, total_mass,
                            reward, reward_values};
    hipLaunchKernelGGL(dw_k_amp_reward, dim3(blocks(n)), dim3(TPB), 0, (hipStream_t)stream, A);
    return launched("dw_amp_reward: launch");
}

int dw_amp_reset(int n, const int64_t *progress_buf, const float *contact_buf, const int32_t *contact_body_ids, int n_contact_ids,
                 const float *rigid_body_pos, const float *rigid_body_rot, float max_episode_length, int enable_early_termination,
                 float termination_height, int64_t *reset, int64_t *terminated, void *stream) {
    if (n <= 0 || !progress_buf || !contact_buf || (n_contact_ids > 0 && !contact_body_ids) || n_contact_ids < 0 || !rigid_body_pos ||
        !rigid_body_rot || !reset || !terminated)
        return fail(DW_EINVAL, "dw_amp_reset: null argument or n <= 0");
    if (n_contact_ids > DW_NUM_BODIES) return fail(DW_EINVAL, "dw_amp_reset: more contact body ids than bodies");
    const dwa::ResetArgs A{n, progress_buf, contact_buf, contact_body_ids, n_contact_ids, rigid_body_pos, rigid_body_rot, max_episode_length,
                           enable_early_termination, termination_height, reset, terminated};
    hipLaunchKernelGGL(dw_k_amp_reset, dim3(blocks(n)), dim3(TPB), 0, (hipStream_t)stream, A);
    return launched("dw_amp_reset: launch");
}

int dw_newwalk_reward(int n, const int64_t *reset_buf, const int64_t *progress_buf, const float *target_vel, const float *root_pose_states,
                      const float *joint_position_states, const float *joint_velocity_states, const int32_t *non_feet_idxs, int n_non_feet,
                      const float *contact_forces, int num_bodies, float termination_height, float death_cost, float max_episode_length,
                      const float *q_nominal, int num_dof, const float *head_states, const float *lfoot_states, const float *rfoot_states,
                      const float *phase, float *total_reward, int64_t *reset, float *reward8, void *stream) {
    if (n <= 0 || !reset_buf || !progress_buf || !target_vel || !root_pose_states || !joint_position_states || !joint_velocity_states ||
        (n_non_feet > 0 && !non_feet_idxs) || n_non_feet < 0 || !contact_forces || !q_nominal || !head_states || !lfoot_states ||
        !rfoot_states || !phase || !total_reward || !reset || !reward8)
        return fail(DW_EINVAL, "dw_newwalk_reward: null argument or n <= 0");
    if (num_dof <= 0 || num_dof > dwa::NW_MAX_DOF || num_bodies < 15) return fail(DW_EINVAL, "dw_newwalk_reward: num_dof must be 1..64 and num_bodies >= 15");
    if (n_non_feet > num_bodies) return fail(DW_EINVAL, "dw_newwalk_reward: more non-feet indices than bodies");
    const dwa::NewWalkArgs A{n, reset_buf, progress_buf, target_vel, root_pose_states, joint_position_states, joint_velocity_states,
                             non_feet_idxs, n_non_feet, contact_forces, num_bodies, termination_height, death_cost, max_episode_length,
                             q_nominal, num_dof, head_states, lfoot_states, rfoot_states, phase, total_reward, reset, reward8};
    hipLaunchKernelGGL(dw_k_newwalk_reward, dim3(blocks(n)), dim3(TPB), 0, (hipStream_t)stream, A);
    return launched("dw_newwalk_reward: launch");
}

int dw_body_positions(DwHandle *h, const int32_t *moving_bodies, int nb, float *out, void *stream) {
    if (!h || !moving_bodies || !out) return fail(DW_EINVAL, "dw_body_positions: null argument");
    if (nb <= 0 || nb > DW_MAX_BODY_QUERY) return fail(DW_EINVAL, "dw_body_positions: nb must be 1..DW_MAX_BODY_QUERY");
    if (!h->bound) return fail(DW_ESTATE, "dw_body_positions: dw_bind first");
    BodyList bl;
    bl.nb = nb;
    for (int k = 0; k < DW_MAX_BODY_QUERY; ++k) bl.b[k] = 0;
    for (int k = 0; k < nb; ++k) {
        if (moving_bodies[k] < 0 || moving_bodies[k] >= DW_NUM_MOVING) return fail(DW_EINVAL, "dw_body_positions: moving body index out of range");
        bl.b[k] = moving_bodies[k];
    }
    const int n = h->cfg.num_envs;
    hipLaunchKernelGGL(dw_k_body_positions, dim3(blocks(n * nb)), dim3(TPB), 0, (hipStream_t)stream, h->d_model, h->buf.root_states,
                       h->buf.dof_state, bl, n, out);
    return launched("dw_body_positions: launch");
}

}  // extern "C"
