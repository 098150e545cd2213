// dw_amp.hip -- gfx950 entry points of the sibling TOCABI tasks' env-side functions (bodies:
// dw_amp.h).  One thread per env: these are 36-word observations and nine reward terms on state the physics kernels left in
// the Gym tensors -- a few hundred bytes per env, HBM-bound, one coalesced pass.  The stateless functions mirror the
// reference's TorchScript signatures (tasks/amp/tocabi_amp_lower_base.py:918-1069, tasks/tocabi_new_walk.py:384-496) so
// that a maintainer binds them where the reference calls its own; dw_body_positions stands in for the rows of
// acquire_rigid_body_state_tensor those functions read.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdio.h>

#include "dw_handle.h"
#include "dw_amp.h"
#include "dw_amp_step.h"

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

// ---------------------------------------------------------------------------------------------------------------------
// The fused TocabiAMPLower step and reset (include/dyros_walk.h: dw_amp_step_begin / _mid / _end, dw_amp_reset_rows / _done): one
// wavefront = one workgroup = one env; bodies in dw_amp_step.h (the same source the host emulation compiles).
namespace {

// (workgroups of four waves per 16 envs: dw_amp_step.h EnvGroup)
__global__ __launch_bounds__(dwa::GT) void dw_k_amp_step_begin(const DwAmpConfig C, const DwAmpBuffers B, const float *dof_state, const float *actions_in,
                                                               const int64_t *ramp_dur, const float *ramp_u) {
    __shared__ dwa::BeginLds S;
    dwa::step_begin(dwa::EnvGroup(), S, C, B, dof_state, actions_in, ramp_dur, ramp_u, (int)blockIdx.x);
}
__global__ __launch_bounds__(dwa::GT) void dw_k_amp_step_mid(const DwAmpConfig C, const DwAmpBuffers B, const float *dof_state, const float *z, int substep) {
    dwa::step_mid(dwa::EnvGroup(), C, B, dof_state, z, substep, (int)blockIdx.x);
}
__global__ __launch_bounds__(dwa::GT) void dw_k_amp_step_end(const dw::DevModel *__restrict__ M, const DwAmpConfig C, const DwAmpBuffers B, const dwa::GymRows G,
                                                             const float *z, int substep, const float *rootvel_noise) {
    __shared__ dwa::GroupLds S;
    dwa::step_end(dwa::EnvGroup(), S, *M, C, B, G, z, substep, rootvel_noise, (int)blockIdx.x);
}
// reset_idx of the listed envs: one wavefront per env, four per workgroup; the draws are rows of the caller's arrays in the order of the list
constexpr int RW = 4;
__global__ __launch_bounds__(64 * RW) void dw_k_amp_reset_rows(const dw::DevModel *__restrict__ M, const DwAmpConfig C, const DwAmpBuffers B, const dwa::GymRows G,
                                                              const int64_t *ids, int n, dwa::ResetSrc R) {
    __shared__ dwa::StepLds S[RW];
    const int w = (int)(threadIdx.x >> 6), k = (int)blockIdx.x * RW + w;
    if (k >= n) return;                            // (wave-uniform: the whole wave leaves; no workgroup barrier below)
    const int e = (int)ids[k];
    if (e < 0 || e >= C.num_envs) return;          // (ids come from device memory: never write past the tensors)
    R.row = (size_t)k;
    dwa::reset_env(dwa::EnvWave(), S[w], *M, C, B, G, R, e);
}
// reset_done: every env whose reset_buf is set; the draws are indexed by env (or made here)
__global__ __launch_bounds__(64 * RW) void dw_k_amp_reset_done(const dw::DevModel *__restrict__ M, const DwAmpConfig C, const DwAmpBuffers B, const dwa::GymRows G,
                                                              dwa::ResetSrc R) {
    __shared__ dwa::StepLds S[RW];
    const int w = (int)(threadIdx.x >> 6), e = (int)blockIdx.x * RW + w;
    if (e >= C.num_envs || B.reset_buf[e] == 0) return;          // (wave-uniform)
    R.row = (size_t)e;
    dwa::reset_env(dwa::EnvWave(), S[w], *M, C, B, G, R, e);
}

// The ids of the envs whose reset_buf is set, ascending, and their count: what `reset_buf.nonzero()` gives VecTask.reset_done
// (tasks/base/vec_task.py:381), as ONE workgroup of 1024 threads.  Flags are taken 16 x 1024 at a time, thread t the flags t, t + 1024, ...
// (coalesced, all sixteen requested before the first is looked at); a chunk's ballot per wave gives the lane's rank inside its wave, the
// 16 x 16 table of (chunk, wave) counts is scanned once by four waves, and every set flag's id lands at base + prefix(chunk, wave) + rank.
// (A thread walking a contiguous run of 16 flags -- 128 bytes apart from its neighbour's -- took 18 us at 16384 envs; torch's nonzero chain
// seven launches and 35 us.)
__global__ __launch_bounds__(1024) void dw_k_amp_reset_ids(const int64_t *__restrict__ flags, int n, int64_t *__restrict__ ids, int64_t *__restrict__ count,
                                                           int64_t *__restrict__ count_host) {
    constexpr int CH = 16;
    __shared__ int tab[CH * 16], pre[CH * 16], wtot[4];
    const int t = (int)threadIdx.x, lane = t & 63, w = t >> 6;
    int base = 0;
    for (int s0 = 0; s0 < n; s0 += CH * 1024) {
        int64_t v[CH];
        unsigned long long bal[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { const int i = s0 + c * 1024 + t; v[c] = i < n ? flags[i] : 0; }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            bal[c] = __ballot(v[c] != 0);
            if (lane == 0) tab[c * 16 + w] = __popcll(bal[c]);
        }
        __syncthreads();
        int x = 0, incl = 0;
        if (t < CH * 16) {          // (four waves: entry t = chunk t >> 4, wave t & 15)
            x = tab[t]; incl = x;
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
            if (lane == 63) wtot[w] = incl;
        }
        __syncthreads();
        if (t < CH * 16) {
            int add = 0;
            for (int k = 0; k < w; ++k) add += wtot[k];
            pre[t] = add + incl - x;
        }
        const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (v[c] != 0) ids[base + pre[c * 16 + w] + __popcll(bal[c] & ((1ull << lane) - 1ull))] = s0 + c * 1024 + t;
        base += total;
        __syncthreads();          // (tab / pre / wtot are rewritten by the next sixteen chunks)
    }
    if (t == 0) { *count = base; if (count_host) *count_host = base; }
}

bool amp_args_ok(const DwAmpConfig *c, const DwAmpBuffers *b) {
    if (!c || !b) return false;
    // every table entry up to init_angle is mandatory (a null one would be a fault on the device, not an error code); the PD offsets, the
    // reset's own buffers, the ring heads and the draw counters are checked by the entry points that use them
    static_assert(sizeof(DwAmpBuffers) % sizeof(void *) == 0, "DwAmpBuffers is a table of pointers");
    const void *const *tbl = reinterpret_cast<const void *const *>(b);
    const size_t mandatory = offsetof(DwAmpBuffers, pd_action_offset) / sizeof(void *);
    for (size_t i = 0; i < mandatory; ++i) if (!tbl[i]) return false;
    if (c->hist_ring && !b->hist_head) return false;
    if (c->device_draws && !b->draw_ctr) return false;
    return c->num_envs > 0 && c->num_his >= 1 && c->num_skip >= 1 && c->num_his * c->num_skip * DW_AMP_NUM_OBS1 <= dwa::HIST_STAGE &&
           c->num_his * c->num_skip * 12 <= dwa::HIST_STAGE && c->log_slots >= 1 && c->amp_steps >= 1 && c->amp_steps * dwa::AW <= dwa::HIST_STAGE;
}
dwa::GymRows gym_rows(const DwHandle *h) {
    return dwa::GymRows{h->buf.root_states, h->buf.dof_state, h->buf.contact_forces, h->buf.dof_damping, h->buf.dof_armature};
}
const char *handle_ok(const DwHandle *h, const DwAmpConfig *c) {
    if (!h) return "null handle";
    if (!h->bound) return "dw_bind first";
    if (c->num_envs != h->cfg.num_envs) return "num_envs differs from the handle's";
    return nullptr;
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

int dw_amp_step_begin(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *actions_in, const int64_t *ramp_dur, const float *ramp_u,
                      void *stream) {
    if (!amp_args_ok(c, b) || !actions_in) return fail(DW_EINVAL, "dw_amp_step_begin: bad configuration (history / AMP sizes beyond what a wave stages, ring or draw buffers missing) or null argument");
    if (const char *m = handle_ok(h, c)) { char t[160]; snprintf(t, sizeof t, "dw_amp_step_begin: %s", m); return fail(h && !h->bound ? DW_ESTATE : DW_EINVAL, t); }
    if (c->vel_change && !c->device_draws && (!ramp_dur || !ramp_u)) return fail(DW_EINVAL, "dw_amp_step_begin: vel_change needs the ramp draws (or device_draws)");
    if (c->pd_control && (!b->pd_action_offset || !b->pd_action_scale)) return fail(DW_EINVAL, "dw_amp_step_begin: pd_control needs the action offset / scale");
    hipLaunchKernelGGL(dw_k_amp_step_begin, dim3((c->num_envs + dwa::GE - 1) / dwa::GE), dim3(dwa::GT), 0, (hipStream_t)stream, *c, *b, h->buf.dof_state, actions_in, ramp_dur, ramp_u);
    return launched("dw_amp_step_begin: launch");
}
int dw_amp_step_mid(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *z, int substep, void *stream) {
    if (!amp_args_ok(c, b) || substep < 1 || substep >= 8) return fail(DW_EINVAL, "dw_amp_step_mid: bad configuration or substep not in 1..7");
    if (const char *m = handle_ok(h, c)) { char t[160]; snprintf(t, sizeof t, "dw_amp_step_mid: %s", m); return fail(h && !h->bound ? DW_ESTATE : DW_EINVAL, t); }
    if (c->noise && !c->device_draws && !z) return fail(DW_EINVAL, "dw_amp_step_mid: noise needs the encoder draws (or device_draws)");
    if (c->pd_control && (!b->pd_action_offset || !b->pd_action_scale)) return fail(DW_EINVAL, "dw_amp_step_mid: pd_control needs the action offset / scale");
    hipLaunchKernelGGL(dw_k_amp_step_mid, dim3((c->num_envs + dwa::GE - 1) / dwa::GE), dim3(dwa::GT), 0, (hipStream_t)stream, *c, *b, h->buf.dof_state, z, substep - 1);
    return launched("dw_amp_step_mid: launch");
}
int dw_amp_step_end(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *z, int substep, const float *rootvel_noise, void *stream) {
    if (!amp_args_ok(c, b) || substep < 0 || substep >= 8) return fail(DW_EINVAL, "dw_amp_step_end: bad configuration or substep not in 0..7");
    if (const char *m = handle_ok(h, c)) { char t[160]; snprintf(t, sizeof t, "dw_amp_step_end: %s", m); return fail(h && !h->bound ? DW_ESTATE : DW_EINVAL, t); }
    if (!c->device_draws && ((c->noise && !z) || !rootvel_noise)) return fail(DW_EINVAL, "dw_amp_step_end: the encoder / root-velocity draws are missing (or device_draws)");
    hipLaunchKernelGGL(dw_k_amp_step_end, dim3((c->num_envs + dwa::GE - 1) / dwa::GE), dim3(dwa::GT), 0, (hipStream_t)stream, h->d_model, *c, *b, gym_rows(h), z, substep, rootvel_noise);
    return launched("dw_amp_step_end: launch");
}

// the whole step in one launch (dw_oct_kernels.hip: dw_k_amp_step_oct)
}  // extern "C"
namespace dwo {
void launch_amp_step(int num_envs, hipStream_t stream, const dwq::QuadModel *QM, const dw::DevModel *M, const dw::DevParams *P, const DwBuffers &Bf, const void *d_args,
                     const float *actions_in, const int64_t *ramp_dur, const float *ramp_u, const float *const *z, int K, const float *rootvel_noise);
int amp_args_to_device(DwHandle *h, const DwAmpConfig &C, const DwAmpBuffers &B, hipStream_t stream);
}
extern "C" {
int dw_amp_step(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *actions_in, const int64_t *ramp_dur, const float *ramp_u, const float *const *z,
                int substeps, const float *rootvel_noise, void *stream) {
    if (!amp_args_ok(c, b) || !actions_in || substeps < 1 || substeps > 8) return fail(DW_EINVAL, "dw_amp_step: bad configuration, null argument or substeps not in 1..8");
    if (const char *m = handle_ok(h, c)) { char t[160]; snprintf(t, sizeof t, "dw_amp_step: %s", m); return fail(h && !h->bound ? DW_ESTATE : DW_EINVAL, t); }
    if (h->cfg.terrain) return fail(DW_EINVAL, "dw_amp_step: the one-launch step is built for the plane (use dw_amp_step_begin / _mid / _end around dw_simulate on terrain)");
    if (c->vel_change && !c->device_draws && (!ramp_dur || !ramp_u)) return fail(DW_EINVAL, "dw_amp_step: vel_change needs the ramp draws (or device_draws)");
    if (c->pd_control && (!b->pd_action_offset || !b->pd_action_scale)) return fail(DW_EINVAL, "dw_amp_step: pd_control needs the action offset / scale");
    if (!c->device_draws) {
        if (!rootvel_noise) return fail(DW_EINVAL, "dw_amp_step: the root-velocity draws are missing (or device_draws)");
        if (c->noise) {
            if (!z) return fail(DW_EINVAL, "dw_amp_step: noise needs the encoder draws of every substep (or device_draws)");
            for (int k = 0; k < substeps; ++k) if (!z[k]) return fail(DW_EINVAL, "dw_amp_step: noise needs the encoder draws of every substep (or device_draws)");
        }
    }
    if (const int rc = dwo::amp_args_to_device(h, *c, *b, (hipStream_t)stream)) return fail(rc, "dw_amp_step: could not place the task's tables in device memory");
    dwo::launch_amp_step(c->num_envs, (hipStream_t)stream, h->d_qmodel, h->d_model, h->d_params, h->buf, h->d_amp_args, actions_in, ramp_dur, ramp_u, z, substeps, rootvel_noise);
    return launched("dw_amp_step: launch");
}

int dw_amp_reset_rows(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const int64_t *ids, int n, const float *power_scale,
                      const float *rootvel_noise, const float *cmd_x, const float *cmd_y, const float *cmd_yaw, const float *qpos_bias,
                      const float *quat_bias, const int64_t *perturb_timing, const int64_t *delay_idx, void *stream) {
    if (!h || !amp_args_ok(c, b) || !ids || !rootvel_noise || !cmd_x || !cmd_y || !cmd_yaw || !perturb_timing || !delay_idx)
        return fail(DW_EINVAL, "dw_amp_reset_rows: bad configuration or null argument");
    if (c->noise && (!qpos_bias || !quat_bias)) return fail(DW_EINVAL, "dw_amp_reset_rows: noise needs the bias draws");
    if (!b->epi_len_log || !b->perturbation_count || !b->perturb_timing || !b->pert_on || !b->initial_root_states)
        return fail(DW_EINVAL, "dw_amp_reset_rows: the reset's own buffers are missing from DwAmpBuffers");
    if (!h->bound) return fail(DW_ESTATE, "dw_amp_reset_rows: dw_bind first");
    if (n < 0 || n > c->num_envs || c->num_envs != h->cfg.num_envs) return fail(DW_EINVAL, "dw_amp_reset_rows: n out of range or num_envs differs from the handle's");
    if (n == 0) return DW_OK;
    DwAmpConfig cc = *c;
    cc.device_draws = 0;                           // (every draw of this entry point is the caller's)
    dwa::ResetSrc R{power_scale, cmd_x, cmd_y, cmd_yaw, qpos_bias, quat_bias, nullptr, nullptr, perturb_timing, delay_idx, rootvel_noise, 0, false,
                    power_scale != nullptr};
    hipLaunchKernelGGL(dw_k_amp_reset_rows, dim3((n + RW - 1) / RW), dim3(64 * RW), 0, (hipStream_t)stream, h->d_model, cc, *b, gym_rows(h), ids, n, R);
    return launched("dw_amp_reset_rows: launch");
}

int dw_amp_reset_ids(const int64_t *reset_buf, int n, int64_t *ids, int64_t *count, int64_t *count_host, void *stream) {
    if (!reset_buf || !ids || !count || n <= 0) return fail(DW_EINVAL, "dw_amp_reset_ids: null argument or n <= 0");
    int64_t *host_dev = nullptr;
    if (count_host) {
        // the kernel stores through this pointer: it has to be pinned host memory the device has mapped (hipHostMalloc / hipHostRegister)
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, count_host) != hipSuccess || at.type != hipMemoryTypeHost || !at.devicePointer) {
            (void)hipGetLastError();
            return fail(DW_EINVAL, "dw_amp_reset_ids: count_host is not pinned host memory mapped to the device");
        }
        host_dev = static_cast<int64_t *>(at.devicePointer);
    }
    hipLaunchKernelGGL(dw_k_amp_reset_ids, dim3(1), dim3(1024), 0, (hipStream_t)stream, reset_buf, n, ids, count, host_dev);
    return launched("dw_amp_reset_ids: launch");
}

int dw_amp_reset_done(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const DwAmpResetDraws *d, void *stream) {
    if (!amp_args_ok(c, b)) return fail(DW_EINVAL, "dw_amp_reset_done: bad configuration or null argument");
    if (const char *m = handle_ok(h, c)) { char t[160]; snprintf(t, sizeof t, "dw_amp_reset_done: %s", m); return fail(h && !h->bound ? DW_ESTATE : DW_EINVAL, t); }
    if (!b->epi_len_log || !b->perturbation_count || !b->perturb_timing || !b->pert_on || !b->initial_root_states)
        return fail(DW_EINVAL, "dw_amp_reset_done: the reset's own buffers are missing from DwAmpBuffers");
    if ((c->dr_damping || c->dr_armature) && (!b->nominal_damping || !b->nominal_armature || c->dr_frequency < 0))
        return fail(DW_EINVAL, "dw_amp_reset_done: dof-property randomisation needs the nominal tables");
    static const DwAmpResetDraws none = {};
    if (!d) d = &none;
    if (!c->device_draws) {
        const bool dr_ok = (!c->dr_damping || d->damping_u) && (!c->dr_armature || d->armature_u);
        if (!d->rootvel_noise || !d->cmd_x_u || !d->cmd_y_u || !d->cmd_yaw_u || !d->perturb_timing || !d->delay_idx || (c->randomize && !d->power_scale_u) ||
            (c->noise && (!d->qpos_bias_u || !d->quat_bias_u)) || !dr_ok)
            return fail(DW_EINVAL, "dw_amp_reset_done: draws missing (every array is required without device_draws)");
    } else if (c->delay_idx_range[1] <= c->delay_idx_range[0]) {
        return fail(DW_EINVAL, "dw_amp_reset_done: delay_idx_range is empty");
    }
    dwa::ResetSrc R{d->power_scale_u, d->cmd_x_u, d->cmd_y_u, d->cmd_yaw_u, d->qpos_bias_u, d->quat_bias_u, d->damping_u, d->armature_u, d->perturb_timing,
                    d->delay_idx, d->rootvel_noise, 0, true, c->randomize != 0};
    hipLaunchKernelGGL(dw_k_amp_reset_done, dim3((c->num_envs + RW - 1) / RW), dim3(64 * RW), 0, (hipStream_t)stream, h->d_model, *c, *b, gym_rows(h), R);
    return launched("dw_amp_reset_done: launch");
}

}  // extern "C"
