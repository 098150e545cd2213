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
// The fused TocabiAMPLower step (include/dyros_walk.h: dw_amp_step_*): one wavefront per env, lanes over the env's rows.  Every
// expression is the torch class' (isaacgymdyros_amd/tocabi_amp_lower.py, itself pinned to the reference class by replay), in its
// operation order, with fp contraction off: the fused step has to give the torch implementation's bits.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
namespace {

constexpr int WPB = 4;                      // waves (envs) per workgroup
__device__ inline int wave_env() { return (int)(blockIdx.x * WPB + (threadIdx.x >> 6)); }
__device__ inline int wave_lane() { return (int)(threadIdx.x & 63u); }
__device__ inline void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
int env_blocks(int n) { return (n + WPB - 1) / WPB; }

// the leg rows of the model into LDS, by the whole workgroup (a chain walk on one lane reads them serially: from global memory
// that was the longest thing in the kernel)
__device__ inline void stage_leg_model(dwa::LegModel &L, const dw::DevModel &M) {
    for (int i = (int)threadIdx.x; i < dwa::LegModel::NBODY * 16; i += 64 * WPB) {
        const int b = i >> 4, k = i & 15;
        if (k < 3) L.pos[b][k] = M.pos[b][k];
        else if (k < 6) L.axis[b][k - 3] = M.axis[b][k - 3];
        else if (k < 15) L.rot0[b][k - 6] = M.rot0[b][k - 6];
        else L.parent[b] = M.parent[b];
    }
    __syncthreads();
}

// row[0 .. len - shift) = row[shift .. len), row[len - shift .. len) = tail[0 .. shift): every lane reads its elements before any writes
template <int MAXPER>
__device__ inline void shift_append(float *row, int len, int shift, const float *tail, int l) {
    float keep[MAXPER];
    for (int k = 0; k < MAXPER; ++k) {
        const int i = l + 64 * k;
        keep[k] = i < len - shift ? row[i + shift] : (i < len ? tail[i - (len - shift)] : 0.0f);
    }
    wave_fence();
    for (int k = 0; k < MAXPER; ++k) {
        const int i = l + 64 * k;
        if (i < len) row[i] = keep[k];
    }
}

__global__ __launch_bounds__(64 * WPB) void dw_k_amp_step_pre(const DwAmpConfig C, const DwAmpBuffers B, const float *actions_in, const int64_t *ramp_dur,
                                                              const float *ramp_u) {
    const int e = wave_env(), l = wave_lane();
    if (e >= C.num_envs) return;
    // actions: clamp (vec_task clip), the record, the action history (tocabi_amp_lower_base.py:642-650)
    float a = 0.0f;
    if (l < 12) {
        a = actions_in[12 * (size_t)e + l];
        a = fminf(fmaxf(a, -C.clip_actions), C.clip_actions);
    }
    __shared__ float sa[WPB][12];
    float *acts = sa[threadIdx.x >> 6];
    if (l < 12) { acts[l] = a; B.actions[12 * (size_t)e + l] = a; }
    wave_fence();
    shift_append<4>(B.action_history + (size_t)C.num_his * C.num_skip * 12 * e, C.num_his * C.num_skip * 12, 12, acts, l);
    // command ramp (:676-693), the branch of the host class that draws for every env
    if (C.vel_change && l < 3) {
        const int half = (int)(C.max_episode_length / 2), when = (int)(C.max_episode_length / 4 - 1);
        const bool change = fmodf(B.epi_len[e], (float)half) == (float)when;
        int64_t dur = B.vel_change_duration[e], cur = B.cur_vel_change_duration[e];
        float start = B.start_target_vel[3 * (size_t)e + l], fin = B.final_target_vel[3 * (size_t)e + l], cmd = B.commands[3 * (size_t)e + l];
        if (change) {
            dur = ramp_dur[e];
            cur = 0;
            start = cmd;
            fin = C.cmd_scale[l] * ramp_u[3 * (size_t)e + l] + C.cmd_lo[l];
        }
        const bool mask = cur < dur;
        const float ramp = start + (fin - start) * (float)cur / (float)dur;
        if (mask) cmd = ramp;
        B.start_target_vel[3 * (size_t)e + l] = start;
        B.final_target_vel[3 * (size_t)e + l] = fin;
        B.commands[3 * (size_t)e + l] = cmd;
        wave_fence();
        if (l == 0) {
            B.vel_change_duration[e] = dur;
            B.cur_vel_change_duration[e] = cur + (mask ? 1 : 0);
        }
    }
}

__global__ __launch_bounds__(64 * WPB) void dw_k_amp_step_tau(const DwAmpConfig C, const DwAmpBuffers B, const float *dof_state) {
    const int e = wave_env(), l = wave_lane();
    if (e >= C.num_envs) return;
    const int64_t sl0 = B.simul_len[e], dl = B.delay_idx[e];
    float tau = 0.0f;
    if (l < DW_NUM_DOF) {
        const float q = dof_state[((size_t)DW_NUM_DOF * e + l) * 2], qd = dof_state[((size_t)DW_NUM_DOF * e + l) * 2 + 1];
        if (l >= 12) {
            tau = B.p_gains[l] * (B.init_angle[l] - q) + B.d_gains[l] * (-qd);          // upper body: PD to the initial pose (:696)
        } else if (C.pd_control) {
            const float tar = B.pd_action_offset[l] + B.pd_action_scale[l] * B.actions[12 * (size_t)e + l];
            tau = B.p_gains[l] * (tar - q) + B.d_gains[l] * (-qd);
        } else {
            const float m = B.motor_efforts[l];
            float lower = B.actions[12 * (size_t)e + l] * m * B.power_scale[12 * (size_t)e + l];
            lower = fmaxf(fminf(lower, m), -m);
            // delayed-torque FIFO (:712-724), column l: shift, append, read `delay_idx` back once the FIFO has filled that far
            float *col = B.action_log + (size_t)C.log_slots * 12 * e + l;
            int64_t sl = sl0 + 1;
            sl = sl > C.log_slots ? C.log_slots : (sl < 0 ? 0 : sl);
            float delayed = 0.0f;
            for (int s = 0; s < C.log_slots; ++s) {
                const float v = s + 1 < C.log_slots ? col[(size_t)12 * (s + 1)] : lower;
                col[(size_t)12 * s] = v;
                const int64_t want = sl > dl ? dl : (int64_t)C.log_slots - sl;
                if (s == want) delayed = v;
            }
            tau = C.noise ? delayed : lower;
        }
        B.tau[(size_t)DW_NUM_DOF * e + l] = tau;
    }
    if (!C.pd_control && l == 0) {
        int64_t sl = sl0 + 1;
        B.simul_len[e] = sl > C.log_slots ? C.log_slots : (sl < 0 ? 0 : sl);
    }
}

__global__ __launch_bounds__(64 * WPB) void dw_k_amp_step_encoder(const DwAmpConfig C, const DwAmpBuffers B, const float *dof_state, const float *z) {
    const int e = wave_env(), l = wave_lane();
    if (e >= C.num_envs || l >= DW_NUM_DOF) return;
    const size_t g = (size_t)DW_NUM_DOF * e + l;
    const float q = dof_state[g * 2];
    const float qn = C.noise ? q + fminf(fmaxf(z[g], -0.00016f), 0.00016f) : q;
    const float d = qn - B.qpos_pre[g];
    B.qpos_noise[g] = qn;
    B.qvel_noise[g] = C.gpu_div ? d * C.inv_dt : d / C.dt;
    B.qpos_pre[g] = qn;
}

__global__ __launch_bounds__(64 * WPB) void dw_k_amp_step_post(const dw::DevModel *__restrict__ M, const DwAmpConfig C, const DwAmpBuffers B, const float *root_states,
                                                               const float *dof_state, const float *contact_forces, const float *rootvel_noise) {
    const int e = wave_env(), l = wave_lane(), w = (int)(threadIdx.x >> 6);
    __shared__ dwa::LegModel LM;
    stage_leg_model(LM, *M);
    if (e >= C.num_envs) return;
    // the env's rows staged in LDS once: the per-env functions below are serial code on one lane each, and a dependent global load
    // per operand is what they must not pay
    __shared__ float s_obs[WPB][DW_AMP_NUM_OBS1], s_amp[WPB][DW_AMP_DISC_BASE + 6], s_foot[WPB][6], s_root[WPB][13], s_ds[WPB][DW_NUM_DOF * 2],
        s_cf[WPB][DW_NUM_BODIES * 3], s_qn[WPB][12], s_qv[WPB][12], s_small[WPB][64], s_dvp[WPB][DW_NUM_DOF];
    enum { SM_NZ = 0, SM_BIAS = 6, SM_QB = 18, SM_CMD = 21, SM_ACT = 24, SM_ACTP = 36, SM_EFF = 48 };          // words of s_small
    const int NH = C.num_his * C.num_skip;
    if (l < 13) s_root[w][l] = root_states[13 * (size_t)e + l];
    for (int i = l; i < DW_NUM_DOF * 2; i += 64) s_ds[w][i] = dof_state[(size_t)DW_NUM_DOF * 2 * e + i];
    for (int i = l; i < DW_NUM_BODIES * 3; i += 64) s_cf[w][i] = contact_forces[(size_t)DW_NUM_BODIES * 3 * e + i];
    if (l < 12) {
        s_qn[w][l] = B.qpos_noise[(size_t)DW_NUM_DOF * e + l]; s_qv[w][l] = B.qvel_noise[(size_t)DW_NUM_DOF * e + l];
        s_small[w][SM_BIAS + l] = B.qpos_bias[12 * (size_t)e + l];
        s_small[w][SM_ACT + l] = B.actions[12 * (size_t)e + l]; s_small[w][SM_ACTP + l] = B.actions_pre[12 * (size_t)e + l];
        s_small[w][SM_EFF + l] = B.motor_efforts[l];
    }
    if (l < 6) s_small[w][SM_NZ + l] = rootvel_noise[6 * (size_t)e + l];
    if (l < 3) { s_small[w][SM_QB + l] = B.quat_bias[3 * (size_t)e + l]; s_small[w][SM_CMD + l] = B.commands[3 * (size_t)e + l]; }
    if (l < DW_NUM_DOF) s_dvp[w][l] = B.dof_vel_pre[(size_t)DW_NUM_DOF * e + l];
    const float tmass = B.total_mass[e];
    int64_t prog = B.progress_buf[e] + 1;
    wave_fence();
    const float *r = s_root[w], *ds = s_ds[w], *cf = s_cf[w];
    // counters (:751-752; epi_len: the last line of pre_physics_step)
    if (l == 0) { B.progress_buf[e] = prog; B.randomize_buf[e] += 1; B.epi_len[e] += 1.0f; }
    // four independent pieces of serial arithmetic on four lanes: the two foot positions (the rigid-body rows the task reads), this
    // step's observation, the reward
    if (l < 2) {
        float p[3];
        dwa::body_position(LM, r, ds, 0, l == 0 ? 6 : 12, p);          // (the staged rows are env 0 of their own little tensors)
        for (int i = 0; i < 3; ++i) {
            s_foot[w][3 * l + i] = p[i];
            B.foot_pos[((size_t)2 * e + l) * 3 + i] = p[i];
            B.rigid_body_pos[((size_t)DW_NUM_BODIES * e + (l == 0 ? 8 : 16)) * 3 + i] = p[i];
        }
    } else if (l == 2) {
        dwa::observations_row(r, &s_small[w][SM_NZ], s_qn[w], &s_small[w][SM_BIAS], &s_small[w][SM_QB], s_qv[w], &s_small[w][SM_CMD], s_obs[w]);
    } else if (l == 3) {
        dwa::reward_row(r, ds + 1, 2, s_dvp[w], &s_small[w][SM_CMD], &s_small[w][SM_ACT], &s_small[w][SM_ACTP], &s_small[w][SM_EFF], cf, tmass,
                        B.rew_buf + e, B.reward_values + 9 * (size_t)e);
    }
    if (l >= 4 && l < 7) B.rigid_body_pos[(size_t)DW_NUM_BODIES * 3 * e + (l - 4)] = r[l - 4];
    if (l >= 8 && l < 12) B.rigid_body_rot[(size_t)DW_NUM_BODIES * 4 * e + (l - 8)] = r[3 + (l - 8)];
    // non-foot bodies in contact (a lane per body)
    bool touch = false;
    if (l < DW_NUM_BODIES && l != 8 && l != 16) touch = cf[3 * l] > 1.0f || cf[3 * l + 1] > 1.0f || cf[3 * l + 2] > 1.0f;
    const bool fall_contact = __builtin_amdgcn_ballot_w64(touch) != 0ull;
    wave_fence();
    // the encoder reading takes the bias (the reference's observation function adds it in place, :945)
    if (l < 12) B.qpos_noise[(size_t)DW_NUM_DOF * e + l] = s_qn[w][l] + s_small[w][SM_BIAS + l];
    if (l < DW_AMP_NUM_OBS1) B.obs1[DW_AMP_NUM_OBS1 * (size_t)e + l] = s_obs[w][l];
    // termination (:1025-1069) on lane 0, the discriminator observation (tasks/tocabi_amp_lower.py:310-350) on lane 1
    if (l == 0) {
        int64_t term = 0;
        if (C.enable_early_termination) {
            bool fall_height = r[2] < C.termination_height;
            fall_height = fall_height || s_foot[w][2] > 0.5f || s_foot[w][5] > 0.5f;
            bool fallen = fall_contact || fall_height;
            const float q0[4] = {r[3], r[4], r[5], r[6]};
            fallen = fallen || fabsf(dw::quat_err(q0)) > (float)(3.141592 / 4.0);
            fallen = fallen && (prog > 1);
            term = fallen ? 1 : 0;
        }
        const int64_t rs = ((float)prog >= C.max_episode_length - 1.0f) ? 1 : term;
        B.terminate_buf[e] = term;
        B.reset_buf[e] = rs;
        B.timeout_buf[e] = (uint8_t)(((float)prog >= C.max_episode_length - 1.0f) && rs != 0);
    } else if (l == 1) {
        dwa::disc_observations_row(r, ds, ds + 1, 2, C.local_root_obs, s_foot[w], 2, s_amp[w]);
    }
    // observation history and the stacked observation (:540-580): obs slots S (i + 1) - 1, action slots S (i + 1), i < H - 1
    float *oh = B.obs_history + (size_t)NH * DW_AMP_NUM_OBS1 * e;
    shift_append<12>(oh, NH * DW_AMP_NUM_OBS1, DW_AMP_NUM_OBS1, s_obs[w], l);
    wave_fence();
    const float *ah = B.action_history + (size_t)NH * 12 * e;
    const int num_obs = (DW_AMP_NUM_OBS1 + 12) * C.num_his - 12;
    float *ob = B.obs_buf + (size_t)num_obs * e, *oo = B.obs_out + (size_t)num_obs * e;
    for (int i = l; i < num_obs; i += 64) {
        float v;
        if (i < DW_AMP_NUM_OBS1 * C.num_his) {
            const int slot = i / DW_AMP_NUM_OBS1, k = i - DW_AMP_NUM_OBS1 * slot;
            v = oh[(size_t)(C.num_skip * (slot + 1) - 1) * DW_AMP_NUM_OBS1 + k];
        } else {
            const int j = i - DW_AMP_NUM_OBS1 * C.num_his, slot = j / 12, k = j - 12 * slot;
            v = ah[(size_t)(C.num_skip * (slot + 1)) * 12 + k];
        }
        ob[i] = v;
        oo[i] = fminf(fmaxf(v, -C.clip_obs), C.clip_obs);
    }
    // what the next step compares against
    if (l < DW_NUM_DOF) B.dof_vel_pre[(size_t)DW_NUM_DOF * e + l] = ds[2 * l + 1];
    if (l < 12) B.actions_pre[12 * (size_t)e + l] = s_small[w][SM_ACT + l];
    // discriminator observation history (tasks/tocabi_amp_lower.py:88-96): slot k -> k + 1, the newest into slot 0
    const int AW = DW_AMP_DISC_BASE + 6;
    float *ab = B.amp_obs_buf + (size_t)C.amp_steps * AW * e;
    float keep[2];
    for (int k = 0; k < 2; ++k) { const int i = l + 64 * k; keep[k] = (i >= AW && i < C.amp_steps * AW) ? ab[i - AW] : 0.0f; }
    wave_fence();
    for (int k = 0; k < 2; ++k) { const int i = l + 64 * k; if (i >= AW && i < C.amp_steps * AW) ab[i] = keep[k]; }
    if (l < AW) { ab[l] = s_amp[w][l]; B.amp_obs1[(size_t)AW * e + l] = s_amp[w][l]; }
}


__global__ __launch_bounds__(64 * WPB) void dw_k_amp_reset_rows(const dw::DevModel *__restrict__ M, const DwAmpConfig C, const DwAmpBuffers B, float *root_states,
                                                                float *dof_state, float *contact_forces, const int64_t *ids, int n, const float *ps,
                                                                const float *rootvel_noise, const float *cmdx, const float *cmdy, const float *cmdyaw,
                                                                const float *qb, const float *quatb, const int64_t *ptime, const int64_t *didx) {
    const int k = wave_env(), l = wave_lane(), w = (int)(threadIdx.x >> 6);
    __shared__ dwa::LegModel LM;
    stage_leg_model(LM, *M);
    if (k >= n) return;
    const int e = (int)ids[k];
    if (e < 0 || e >= C.num_envs) return;          // (ids come from device memory: never write past the tensors)
    __shared__ float s_obs[WPB][DW_AMP_NUM_OBS1], s_amp[WPB][DW_AMP_DISC_BASE + 6], s_foot[WPB][6], s_root[WPB][13], s_ds[WPB][DW_NUM_DOF * 2],
        s_old[WPB][64];
    enum { SO_QN = 0, SO_QV = 12, SO_BIAS = 24, SO_QB = 36, SO_CMD = 39, SO_NZ = 42 };
    const int NH = C.num_his * C.num_skip, AW = DW_AMP_DISC_BASE + 6;
    // what the reset observation is made of: the episode's LAST encoder reading, biases and command (the reference computes it before
    // it draws the new ones, :253 before :266-279)
    if (l < 12) {
        s_old[w][SO_QN + l] = B.qpos_noise[(size_t)DW_NUM_DOF * e + l]; s_old[w][SO_QV + l] = B.qvel_noise[(size_t)DW_NUM_DOF * e + l];
        s_old[w][SO_BIAS + l] = B.qpos_bias[12 * (size_t)e + l];
    }
    if (l < 3) { s_old[w][SO_QB + l] = B.quat_bias[3 * (size_t)e + l]; s_old[w][SO_CMD + l] = B.commands[3 * (size_t)e + l]; }
    if (l < 6) s_old[w][SO_NZ + l] = rootvel_noise[6 * (size_t)e + l];
    // the Gym tensors' rows: initial root state, initial pose at rest, no contact (_reset_actors, :611-626)
    if (l < 13) { const float v = B.initial_root_states[13 * (size_t)e + l]; s_root[w][l] = v; root_states[13 * (size_t)e + l] = v; }
    if (l < DW_NUM_DOF) {
        const float q0 = B.init_angle[l];
        s_ds[w][2 * l] = q0; s_ds[w][2 * l + 1] = 0.0f;
        dof_state[((size_t)DW_NUM_DOF * e + l) * 2] = q0; dof_state[((size_t)DW_NUM_DOF * e + l) * 2 + 1] = 0.0f;
    }
    for (int i = l; i < DW_NUM_BODIES * 3; i += 64) contact_forces[(size_t)DW_NUM_BODIES * 3 * e + i] = 0.0f;
    // (the draws arrive as raw uniforms; the values are formed with torch's arithmetic: `(hi - lo) * u + lo` with the scalars rounded to
    //  float32 first, `x / s` as a multiplication by 1.0f / s on a GPU and a division on a CPU)
    if (ps && l < 12) B.power_scale[12 * (size_t)e + l] = (float)(1.2 - 0.8) * ps[12 * (size_t)k + l] + (float)0.8;
    wave_fence();
    const float *r = s_root[w], *ds = s_ds[w];
    if (l < 2) {          // the rigid-body rows of the new state
        float p[3];
        dwa::body_position(LM, r, ds, 0, l == 0 ? 6 : 12, p);
        for (int i = 0; i < 3; ++i) {
            s_foot[w][3 * l + i] = p[i];
            B.foot_pos[((size_t)2 * e + l) * 3 + i] = p[i];
            B.rigid_body_pos[((size_t)DW_NUM_BODIES * e + (l == 0 ? 8 : 16)) * 3 + i] = p[i];
        }
    } else if (l == 2) {
        dwa::observations_row(r, &s_old[w][SO_NZ], &s_old[w][SO_QN], &s_old[w][SO_BIAS], &s_old[w][SO_QB], &s_old[w][SO_QV], &s_old[w][SO_CMD], s_obs[w]);
    }
    if (l >= 4 && l < 7) B.rigid_body_pos[(size_t)DW_NUM_BODIES * 3 * e + (l - 4)] = r[l - 4];
    if (l >= 8 && l < 12) B.rigid_body_rot[(size_t)DW_NUM_BODIES * 4 * e + (l - 8)] = r[3 + (l - 8)];
    wave_fence();
    if (l == 1) dwa::disc_observations_row(r, ds, ds + 1, 2, C.local_root_obs, s_foot[w], 2, s_amp[w]);
    if (l < DW_AMP_NUM_OBS1) B.obs1[DW_AMP_NUM_OBS1 * (size_t)e + l] = s_obs[w][l];
    // the reset env's observation: every history slot shows the reset observation, the action slots what the action history still
    // holds; only then are the two histories zeroed (:296-297)
    const float *ah = B.action_history + (size_t)NH * 12 * e;
    const int num_obs = (DW_AMP_NUM_OBS1 + 12) * C.num_his - 12;
    float *ob = B.obs_buf + (size_t)num_obs * e;
    for (int i = l; i < num_obs; i += 64) {
        float v;
        if (i < DW_AMP_NUM_OBS1 * C.num_his) v = s_obs[w][i % DW_AMP_NUM_OBS1];
        else { const int j = i - DW_AMP_NUM_OBS1 * C.num_his, slot = j / 12, kk = j - 12 * slot; v = ah[(size_t)(C.num_skip * (slot + 1)) * 12 + kk]; }
        ob[i] = v;
    }
    wave_fence();
    for (int i = l; i < NH * DW_AMP_NUM_OBS1; i += 64) B.obs_history[(size_t)NH * DW_AMP_NUM_OBS1 * e + i] = 0.0f;
    for (int i = l; i < NH * 12; i += 64) B.action_history[(size_t)NH * 12 * e + i] = 0.0f;
    for (int i = l; i < C.log_slots * 12; i += 64) B.action_log[(size_t)C.log_slots * 12 * e + i] = 0.0f;
    if (l < DW_NUM_DOF) {
        const size_t g = (size_t)DW_NUM_DOF * e + l;
        B.dof_vel_pre[g] = 0.0f; B.qpos_noise[g] = B.init_angle[l]; B.qpos_pre[g] = B.init_angle[l]; B.qvel_noise[g] = 0.0f;
    }
    if (l < 12) {
        B.actions_pre[12 * (size_t)e + l] = 0.0f;
        float v = 0.0f;
        if (C.noise) {
            const float x = qb[12 * (size_t)k + l] * 6.28f;
            v = (C.gpu_div ? x * (1.0f / 100.0f) : x / 100.0f) - (float)(3.14 / 100);
        }
        B.qpos_bias[12 * (size_t)e + l] = v;
    }
    if (l < 3) {
        const float u = l == 0 ? cmdx[k] : (l == 1 ? cmdy[k] : cmdyaw[k]);
        B.commands[3 * (size_t)e + l] = C.cmd_scale[l] * u + C.cmd_lo[l];
        float v = 0.0f;
        if (C.noise) {
            const float x = quatb[3 * (size_t)k + l] * 6.28f;
            v = (C.gpu_div ? x * (1.0f / 150.0f) : x / 150.0f) - (float)(3.14 / 150);
        }
        B.quat_bias[3 * (size_t)e + l] = v;
    }
    if (l == 0) {
        B.progress_buf[e] = 0; B.reset_buf[e] = 0; B.terminate_buf[e] = 0;
        B.epi_len_log[e] = B.epi_len[e]; B.epi_len[e] = 0.0f;
        B.perturbation_count[e] = 0; B.pert_on[e] = 0; B.perturb_timing[e] = ptime[k];
        B.delay_idx[e] = didx[k]; B.simul_len[e] = 0;
    }
    wave_fence();
    // discriminator history of a default start: every slot the current observation (tasks/tocabi_amp_lower.py:258-272)
    float *ab = B.amp_obs_buf + (size_t)C.amp_steps * AW * e;
    for (int i = l; i < C.amp_steps * AW; i += 64) ab[i] = s_amp[w][i % AW];
    if (l < AW) B.amp_obs1[(size_t)AW * e + l] = s_amp[w][l];
}

bool amp_args_ok(const DwAmpConfig *c, const DwAmpBuffers *b) {
    if (!c || !b) return false;
    // every table entry up to init_angle is mandatory (a null one would be a fault on the device, not an error code); the PD offsets and
    // the reset's own buffers are checked by the entry points that use them
    static_assert(sizeof(DwAmpBuffers) % sizeof(void *) == 0, "DwAmpBuffers is a table of pointers");
    const void *const *tbl = reinterpret_cast<const void *const *>(b);
    const size_t mandatory = offsetof(DwAmpBuffers, pd_action_offset) / sizeof(void *);
    for (size_t i = 0; i < mandatory; ++i) if (!tbl[i]) return false;
    return c->num_envs > 0 && c->num_his >= 1 && c->num_skip >= 1 && c->num_his * c->num_skip * DW_AMP_NUM_OBS1 <= 64 * 12 &&
           c->num_his * c->num_skip * 12 <= 64 * 4 && c->log_slots >= 1 && c->amp_steps >= 1 && c->amp_steps * (DW_AMP_DISC_BASE + 6) <= 128;
}

}  // namespace
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

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

int dw_amp_step_pre(const DwAmpConfig *c, const DwAmpBuffers *b, const float *actions_in, const int64_t *ramp_dur, const float *ramp_u, void *stream) {
    if (!amp_args_ok(c, b) || !actions_in) return fail(DW_EINVAL, "dw_amp_step_pre: bad configuration (history / AMP sizes beyond what a wave stages) or null argument");
    if (c->vel_change && (!ramp_dur || !ramp_u)) return fail(DW_EINVAL, "dw_amp_step_pre: vel_change needs the ramp draws");
    hipLaunchKernelGGL(dw_k_amp_step_pre, dim3(env_blocks(c->num_envs)), dim3(64 * WPB), 0, (hipStream_t)stream, *c, *b, actions_in, ramp_dur, ramp_u);
    return launched("dw_amp_step_pre: launch");
}
int dw_amp_step_tau(const DwAmpConfig *c, const DwAmpBuffers *b, const float *dof_state, void *stream) {
    if (!amp_args_ok(c, b) || !dof_state) return fail(DW_EINVAL, "dw_amp_step_tau: bad configuration or null argument");
    if (c->pd_control && (!b->pd_action_offset || !b->pd_action_scale)) return fail(DW_EINVAL, "dw_amp_step_tau: pd_control needs the action offset / scale");
    hipLaunchKernelGGL(dw_k_amp_step_tau, dim3(env_blocks(c->num_envs)), dim3(64 * WPB), 0, (hipStream_t)stream, *c, *b, dof_state);
    return launched("dw_amp_step_tau: launch");
}
int dw_amp_step_encoder(const DwAmpConfig *c, const DwAmpBuffers *b, const float *dof_state, const float *z, void *stream) {
    if (!amp_args_ok(c, b) || !dof_state || (c->noise && !z)) return fail(DW_EINVAL, "dw_amp_step_encoder: bad configuration or null argument");
    hipLaunchKernelGGL(dw_k_amp_step_encoder, dim3(env_blocks(c->num_envs)), dim3(64 * WPB), 0, (hipStream_t)stream, *c, *b, dof_state, z);
    return launched("dw_amp_step_encoder: launch");
}
int dw_amp_step_post(DwHandle *h, const DwAmpConfig *c, const DwAmpBuffers *b, const float *rootvel_noise, void *stream) {
    if (!h || !amp_args_ok(c, b) || !rootvel_noise) return fail(DW_EINVAL, "dw_amp_step_post: bad configuration or null argument");
    if (!h->bound) return fail(DW_ESTATE, "dw_amp_step_post: dw_bind first");
    if (c->num_envs != h->cfg.num_envs) return fail(DW_EINVAL, "dw_amp_step_post: num_envs differs from the handle's");
    hipLaunchKernelGGL(dw_k_amp_step_post, dim3(env_blocks(c->num_envs)), dim3(64 * WPB), 0, (hipStream_t)stream, h->d_model, *c, *b, h->buf.root_states,
                       h->buf.dof_state, h->buf.contact_forces, rootvel_noise);
    return launched("dw_amp_step_post: launch");
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
    hipLaunchKernelGGL(dw_k_amp_reset_rows, dim3(env_blocks(n)), dim3(64 * WPB), 0, (hipStream_t)stream, h->d_model, *c, *b, h->buf.root_states,
                       h->buf.dof_state, h->buf.contact_forces, ids, n, power_scale, rootvel_noise, cmd_x, cmd_y, cmd_yaw, qpos_bias, quat_bias,
                       perturb_timing, delay_idx);
    return launched("dw_amp_reset_rows: launch");
}

}  // extern "C"
