// dw_task.h -- the whole VecTask.step of DyrosDynamicWalk for one env, as wave regions.
//
// Mirrors, with every fp32 operation in the reference's order (fp contraction OFF in this file):
//   VecTask.step                        tasks/base/vec_task.py:293-344
//   pre_physics_step                    tasks/dyros_dynamic_walk.py:449-541 (cubic: utils/torch_jit_utils.py:373-395)
//   post_physics_step                   tasks/dyros_dynamic_walk.py:543-563
//   check_termination                   tasks/dyros_dynamic_walk.py:581-596 (quat_diff_rad: utils/torch_jit_utils.py:141-160)
//   compute_humanoid_walk_reward        tasks/dyros_dynamic_walk.py:802-947
//   reset_idx + dof-property DR         tasks/dyros_dynamic_walk.py:598-669,720-748; tasks/base/vec_task.py:519-733
//   compute_humanoid_walk_observations  tasks/dyros_dynamic_walk.py:750-796 (quat2euler: python/isaacgym/torch_utils.py:227-273)
// (paths relative to python/IsaacGymEnvs/isaacgymenvs unless they start with python/).
//
// Data flow: the env's task-state record (DW_ES_WORDS words) and its Gym state are staged into LDS once,
// both physics substeps run on LDS, and only the record, the Gym tensors, the new history slot and the
// 487-word observation go back to HBM.  Histories are rings: nothing is shifted.
#pragma once

#include "dw_physics.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace dw {

struct TaskParams {                  // wave-uniform scalars (kernel arguments)
    PhysParams phys;
    int     num_envs;
    float   inv_dt_f;                // (float)(1.0 / dt): torch-GPU form of `x / self.dt`
    float   dt_policy_f;             // (float)(dt * controlFrequencyInv)
    float   clock_gain_f;            // (float)(5 * dt_policy)
    float   pert_period_f;           // (float)(8 / dt_policy)
    int     pert_dur_lo, pert_dur_hi;// int(0.1/dt_policy), int(1/dt_policy)
    float   max_episode_length;
    float   initial_height;
    float   death_cost;
    float   friction;
    int     perturb, force_perturb_start;
    int     dr_dof, dr_friction;
    float   dr_damp[2], dr_arm[2], dr_fric[2];
    int     timeout_fix;
    int     gpu_div;
    int     freeze_physics;
    unsigned long long seed;
    // terrain curriculum and spawn (row f-4)
    int     terrain_curriculum, custom_origins;
    int     terrain_num_levels, terrain_num_types;
    float   terrain_half_length;       // (float)(terrain_length / 2)
    float   max_episode_length_s;
};

struct TaskBuffers {                 // device pointers (DwBuffers, read through memory) + per-call pointers
    const DwBuffers *b;
    const float *actions;
    const float *noise;              // [N, DW_NOISE_WORDS] or nullptr
    const float *mocap;              // [3600, 36]
    long long    step;
};

// Launch-invariant parameters, resident in device memory: the kernels take ONE pointer instead of ~45 by-value
// words, which is what kept the scalar register file spilling in the first version of dw_k_step.
struct DevParams {
    TaskParams C;
    DwBuffers  B;
    const float *mocap;
};

constexpr int GATE_BUCKETS = 32;     // gate_acc layout: [slot 0..2][bucket 0..31][2] int64, latch at [192]
constexpr int GATE_LATCH = 192;

// ---------------------------------------------------------------------------------------------- RNG
DW_HD void philox4x32_10(unsigned int *c, unsigned int k0, unsigned int k1) {
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
        const unsigned int n3 = (unsigned int)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

struct NoiseSrc {
    const float *rec;          // this env's injected record or nullptr
    unsigned long long seed;
    unsigned int env;
    unsigned long long step;
    unsigned int stream;
};

// Box-Muller normal of the encoder model from two Philox outputs (oracle/dw_task.c noise_word)
DW_HD float enc_normal(unsigned int a, unsigned int b) {
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-08f;
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-08f;
#if defined(__HIPCC__)
    // the hardware log2 / cos (arguments in (0, 1] and [0, 2 pi)): relative error ~1e-6 of a 5e-5 rad draw, against ~250
    // instructions of libm range reduction per draw and 66 draws per env-step
    const float z = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
#else
    const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
#endif
    return z * (float)(0.00016 / 3.0);
}
// Uniform words 4b .. 4b+3 (b >= DW_NZ_VEL / 4) in one generator call: counter word DW_NZ_UBLOCK + b (oracle/dw_task.c noise_word)
DW_HD void noise_block(const NoiseSrc &nz, int b, float (&u)[4]) {
    if (nz.rec) { for (int i = 0; i < 4; ++i) u[i] = nz.rec[4 * b + i]; return; }
    unsigned int c[4] = {(unsigned int)(DW_NZ_UBLOCK + b), nz.env, (unsigned int)nz.step, (unsigned int)(nz.step >> 32) | (nz.stream << 31)};
    philox4x32_10(c, (unsigned int)nz.seed, (unsigned int)(nz.seed >> 32));
    for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * 5.9604644775390625e-08f;
}
DW_HD float noise_word(const NoiseSrc &nz, int w) {
    if (nz.rec) return nz.rec[w];
    if (w >= DW_NZ_VEL) {
        unsigned int c[4] = {(unsigned int)(DW_NZ_UBLOCK + (w >> 2)), nz.env, (unsigned int)nz.step, (unsigned int)(nz.step >> 32) | (nz.stream << 31)};
        philox4x32_10(c, (unsigned int)nz.seed, (unsigned int)(nz.seed >> 32));
        const unsigned int lo = (w & 1) ? c[1] : c[0], hi = (w & 1) ? c[3] : c[2];
        return (float)(((w & 2) ? hi : lo) >> 8) * 5.9604644775390625e-08f;
    }
    // the two encoder draws of a joint (one per substep) share one Philox block: outputs 0,1 and 2,3
    const int pair = w >= DW_NZ_ENC + DW_NUM_DOF ? 1 : 0;
    unsigned int c[4] = {(unsigned int)(pair ? w - DW_NUM_DOF : w), nz.env, (unsigned int)nz.step,
                         (unsigned int)(nz.step >> 32) | (nz.stream << 31)};
    philox4x32_10(c, (unsigned int)nz.seed, (unsigned int)(nz.seed >> 32));
    return pair ? enc_normal(c[2], c[3]) : enc_normal(c[0], c[1]);
}
// both encoder draws of joint d in one generator call (the quad kernels keep the second for the second substep)
DW_HD void noise_enc_pair(const NoiseSrc &nz, int d, float *z0, float *z1) {
    unsigned int c[4] = {(unsigned int)(DW_NZ_ENC + d), nz.env, (unsigned int)nz.step, (unsigned int)(nz.step >> 32) | (nz.stream << 31)};
    philox4x32_10(c, (unsigned int)nz.seed, (unsigned int)(nz.seed >> 32));
    *z0 = enc_normal(c[0], c[1]);
    *z1 = enc_normal(c[2], c[3]);
}

// ---------------------------------------------------------------------------------------------- torch-flavoured scalars
// `tensor / python_scalar`: torch's CPU kernels divide by float(s); its GPU kernels multiply by float(1.0 / s) with the
// reciprocal formed in double (probed on MI355X: x / 0.0005 == x * 2000.0f for every x)
DW_HD float divs(int recip, float x, double s) { return recip ? x * (float)(1.0 / s) : x / (float)s; }
DW_HD float remainder_t(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0 && ((b < 0) != (m < 0))) m += b;
    return m;
}
// torch.norm over a contiguous row on the CPU reference: 8-lane fused accumulation, lanes added in order,
// 4-wide unfused tail chunk, fused scalar tail (oracle/dw_task.c norm_t)
DW_HD float norm_t(const float *x, int n) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int d = 0;
    for (; d < n - (n % 8); d += 8)
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(x[d + l], x[d + l], acc[l]);
    float b0 = acc[0];
    for (int l = 1; l < 8; ++l) b0 = b0 + acc[l];
    for (; d + 4 <= n; d += 4)
        for (int l = 0; l < 4; ++l) { const float p = x[d + l] * x[d + l]; b0 = b0 + p; }
    for (; d < n; ++d) b0 = fmaf(x[d], x[d], b0);
    return sqrtf(b0);
}
// same summation order, elements produced on the fly (keeps 33-element operands out of the register file)
template <class F>
DW_HD float norm_fn(F f, int n) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int d = 0;
    for (; d < n - (n % 8); d += 8)
        for (int l = 0; l < 8; ++l) { const float x = f(d + l); acc[l] = fmaf(x, x, acc[l]); }
    float b0 = acc[0];
    for (int l = 1; l < 8; ++l) b0 = b0 + acc[l];
    for (; d + 4 <= n; d += 4)
        for (int l = 0; l < 4; ++l) { const float x = f(d + l); const float p = x * x; b0 = b0 + p; }
    for (; d < n; ++d) { const float x = f(d); b0 = fmaf(x, x, b0); }
    return sqrtf(b0);
}
DW_HD float cubic_t(float time, float t0, float tf, float x0, float xf) {
    const float elapsed = time - t0;
    const float total = tf - t0;
    const float total2 = total * total;
    const float total3 = total2 * total;
    const float total_x = xf - x0;
    const float c2 = (3.0f * total_x) / total2 - (0.0f / total) - (0.0f / total);
    const float c3 = (-2.0f * total_x) / total3 + (0.0f / total2);
    const float cub = x0 + 0.0f * elapsed + c2 * elapsed * elapsed + c3 * elapsed * elapsed * elapsed;
    float xt = x0;
    if (time > tf) xt = xf;
    if (t0 <= time && time <= tf) xt = cub;
    return xt;
}
DW_HD float quat_err(const float *q) {
    const float x1 = 0, y1 = 0, z1 = 0, w1 = 1;
    const float x2 = -q[0], y2 = -q[1], z2 = -q[2], w2 = q[3];
    const float ww = (z1 + x1) * (x2 + y2);
    const float yy = (w1 - y1) * (w2 + z2);
    const float zz = (w1 + y1) * (w2 - z2);
    const float xx = ww + yy + zz;
    const float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    const float x = qq - xx + (x1 + w1) * (x2 + w2);
    const float y = qq - yy + (w1 - x1) * (y2 + z2);
    const float z = qq - zz + (z1 + y1) * (w2 - x2);
    const float v[3] = {x, y, z};
    float n = norm_t(v, 3);
    if (n > 1.0f) n = 1.0f;
    return 2.0f * asinf(n);
}
DW_HD bool finitef(float x) { return fabsf(x) <= 3.4028234663852886e38f; }   // false for NaN and +-inf

#define ESI(off) (*reinterpret_cast<int *>(&S.es[(off)]))

// LDS block of dw_k_reset (reset_idx of listed envs, one wave per env, both pipelines): the task record and the Gym state
// of one env, 3.6 KB.
struct alignas(16) TaskLds {
    float root[13];
    float q[ND], qd[ND];
    float contact[DW_NUM_BODIES * 3];
    float es[DW_ES_WORDS];
    float act[DW_NUM_ACT];
    float normed[DW_NUM_OBS1];
    float rterm[16];
    float scratch[8];
    int   flags[8];
    float warm[24];
};

// ---------------------------------------------------------------------------------------------- load / store
DW_HD void load_env_lane(int l, Lds &S, const TaskParams &C, const DwBuffers &B, int e, bool with_task) {   // fused kernels
    {
        if (with_task)
            for (int i = l; i < DW_ES_WORDS; i += 64) S.es[i] = B.env_state[(size_t)DW_ES_WORDS * e + i];
        if (l < 13) S.root[l] = B.root_states[13 * e + l];
        if (l < ND) {
            S.q[l] = B.dof_state[(ND * e + l) * 2];
            S.qd[l] = B.dof_state[(ND * e + l) * 2 + 1];
            S.arm[l] = B.dof_armature[ND * e + l];
            S.damp[l] = B.dof_damping[ND * e + l];
        }
        if (l < DW_NUM_BODIES) S.mscale[l] = B.mass_scale[DW_NUM_BODIES * e + l];
        if (l == 40) S.mu = C.friction * B.friction_scale[e];
        if (l < 4) S.flags[l] = 0;
    }
}
template <class W>
DW_HD void load_env(const W &wave, Lds &S, const TaskParams &C, const DwBuffers &B, int e, bool with_task) {
    wave.par([&](int l) { load_env_lane(l, S, C, B, e, with_task); });
}

template <class W, class LT>
DW_HD void store_env(const W &wave, LT &S, const DwBuffers &B, int e, bool with_task, bool with_state) {
    wave.par([&](int l) {
        if (with_task)
            for (int i = l; i < DW_ES_WORDS; i += 64) B.env_state[(size_t)DW_ES_WORDS * e + i] = S.es[i];
        if (with_state) {
            if (l < 13) B.root_states[13 * e + l] = S.root[l];
            if (l < ND) {
                B.dof_state[(ND * e + l) * 2] = S.q[l];
                B.dof_state[(ND * e + l) * 2 + 1] = S.qd[l];
            }
            for (int i = l; i < DW_NUM_BODIES * 3; i += 64) B.contact_forces[(size_t)DW_NUM_BODIES * 3 * e + i] = S.contact[i];
        }
    });
}

// ---------------------------------------------------------------------------------------------- reset_idx (one env)
// Expects: S.es, S.contact (current net contact forces), S.flags[4] = randomize_buf value.  Writes S.root, S.q,
// S.qd, record fields, per-env DR'd parameters and the zeroed action ring.
template <class W, class LT>
DW_HD void reset_region(const W &wave, LT &S, const DevModel &M, const TaskParams &C, const DwBuffers &B,
                        const NoiseSrc &nz, int e) {
    // terrain curriculum (tasks/dyros_dynamic_walk.py:603-604,671-691): from the position the robot reached and the
    // target velocity of the episode that ended; the new origin goes to S.scratch[4..6] for the spawn below
    if (C.terrain_curriculum) {
        wave.par([&](int l) {
            if (l == 0) {
                const float d[2] = {S.root[0] - B.env_origins[3 * e], S.root[1] - B.env_origins[3 * e + 1]};
                const float distance = norm_t(d, 2);
                const bool move_up = distance > C.terrain_half_length;
                const float need = norm_t(&S.es[DW_ES_TARGET_VEL], 2) * C.max_episode_length_s * 0.5f;
                const bool move_down = (distance < need) && !move_up;
                long long lvl = B.terrain_levels[e] + ((move_up ? 1 : 0) - (move_down ? 1 : 0));
                if (lvl >= C.terrain_num_levels) {
                    int k = (int)(noise_word(nz, DW_NZ_TERRAIN_LVL) * (float)C.terrain_num_levels);     // randint_like
                    if (k > C.terrain_num_levels - 1) k = C.terrain_num_levels - 1;
                    lvl = k;
                } else if (lvl < 0) lvl = 0;
                B.terrain_levels[e] = lvl;
                long long ty = B.terrain_types[e];          // user-writable buffer (load_state_dict): never index past the table
                ty = ty < 0 ? 0 : (ty > C.terrain_num_types - 1 ? C.terrain_num_types - 1 : ty);
                const float *org = B.terrain_origins + ((size_t)lvl * C.terrain_num_types + ty) * 3;
                for (int i = 0; i < 3; ++i) { const float o = org[i]; B.env_origins[3 * e + i] = o; S.scratch[4 + i] = o; }
            }
        });
    }
    wave.par([&](int l) {
        const bool do_dr = (C.dr_dof || C.dr_friction) && S.flags[4] >= 1;
        if (l < ND) {
            if (do_dr && C.dr_dof) {
                const float ud = noise_word(nz, DW_NZ_DR_DAMP + l), ua = noise_word(nz, DW_NZ_DR_ARM + l);
                const float sd = C.dr_damp[0] + ud * (C.dr_damp[1] - C.dr_damp[0]);
                const float sa = C.dr_arm[0] + ua * (C.dr_arm[1] - C.dr_arm[0]);
                B.dof_damping[ND * e + l] = M.damp_nom[l] + sd;
                B.dof_armature[ND * e + l] = M.arm_nom[l] * sa;
            }
            S.es[DW_ES_QPOS_NOISE + l] = M.q_init[l];
            S.es[DW_ES_QPOS_PRE + l] = M.q_init[l];
            S.es[DW_ES_QVEL_NOISE + l] = 0.0f;
            S.es[DW_ES_PRE_QVEL + l] = 0.0f;
            S.q[l] = fmaxf(fminf(M.q_init[l], M.qhi[l]), M.qlo[l]);
            S.qd[l] = 0.0f;
        }
        if (l < 12) {
            S.es[DW_ES_QPOS_BIAS + l] = divs(C.gpu_div, noise_word(nz, DW_NZ_QPOS_BIAS + l) * 6.28f, 100.0) - (float)(3.14 / 100);
            S.es[DW_ES_MOTOR_SCALE + l] = noise_word(nz, DW_NZ_MOTOR + l) * 0.4f + 0.8f;
            S.es[DW_ES_ACTION_TORQUE_PRE + l] = 0.0f;
        }
        if (l < 3) S.es[DW_ES_QUAT_BIAS + l] = divs(C.gpu_div, noise_word(nz, DW_NZ_QUAT_BIAS + l) * 6.28f, 150.0) - (float)(3.14 / 150);
        if (l >= 16 && l < 29) {
            const int i = l - 16;
            const float r0[13] = {0, 0, C.initial_height, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
            float v = r0[i];
            if (i < 3) v += C.terrain_curriculum ? S.scratch[4 + i] : B.env_origins[3 * e + i];
            // xy position within 1 m of the tile centre (:729-732; torch_rand_float = 2 u + (-1))
            if (i < 2 && C.custom_origins) v += 2.0f * noise_word(nz, DW_NZ_ROOT_JITTER + i) + (-1.0f);
            S.root[i] = v;
        }
        for (int i = l; i < DW_ALOG_SLOTS * 12; i += 64) S.es[DW_ES_ACTION_LOG + i] = 0.0f;
        if (l >= 32 && l < 32 + 24) S.es[DW_ES_WARM + (l - 32)] = 0.0f;
        if (l >= 56 && l < 62) {
            const int i = l - 56, gy = i < 3 ? M.left_foot_gym : M.right_foot_gym;
            S.es[DW_ES_FOOT_FORCE_PRE + i] = S.contact[3 * gy + (i % 3)];
        }
        if (l == 40) {
            if (do_dr) {
                if (C.dr_friction) {
                    const float uf = noise_word(nz, DW_NZ_DR_FRIC);
                    B.friction_scale[e] = C.dr_fric[0] + uf * (C.dr_fric[1] - C.dr_fric[0]);
                }
                B.randomize_buf[e] = 0;
            }
            const float vel_mag = noise_word(nz, DW_NZ_TARGET_VEL) * 0.8f;
            S.es[DW_ES_TARGET_VEL] = vel_mag * 1.0f;
            S.es[DW_ES_TARGET_VEL + 1] = vel_mag * 0.0f;
            ESI(DW_ES_INIT_MOCAP) = noise_word(nz, DW_NZ_INIT_MOCAP) > 0.5f ? 0 : 1800;
            S.es[DW_ES_TIME] = 0.0f;
            B.progress_buf[e] = 0;
            B.reset_buf[e] = 1;
            int k = (int)(noise_word(nz, DW_NZ_DELAY) * 4.0f);
            if (k > 3) k = 3;
            ESI(DW_ES_DELAY_IDX) = 2 + k;
            S.es[DW_ES_CRM] = S.es[DW_ES_CRS] / S.es[DW_ES_EPI_LEN];
            S.es[DW_ES_CRS] = 0.0f;
            ESI(DW_ES_SIMUL_LEN) = 0;
            S.es[DW_ES_EPI_LEN_LOG] = S.es[DW_ES_EPI_LEN];
            S.es[DW_ES_EPI_LEN] = 0.0f;
            ESI(DW_ES_PERT_COUNT) = 0;
            ESI(DW_ES_PERT_ON) = 0;
            int kt = (int)(noise_word(nz, DW_NZ_PTIMING) * 2000.0f);
            if (kt > 1999) kt = 1999;
            ESI(DW_ES_PERT_TIMING) = kt;
        }
        float *ah = B.action_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_ACT;
        for (int i = l; i < DW_HIST_SLOTS * DW_NUM_ACT; i += 64) ah[i] = 0.0f;
        if (nz.stream == 1) {     // reset_done path: the observation is not recomputed, zero the obs ring as the reference does
            float *oh = B.obs_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_OBS1;
            for (int i = l; i < DW_HIST_SLOTS * DW_NUM_OBS1; i += 64) oh[i] = 0.0f;
        }
    });
}

// ---------------------------------------------------------------------------------------------- the step
// Per-step constants shared by the regions below.
struct StepCtx {
    NoiseSrc nz;
    float period, cdt, dt;
    double cdt_d;
    long long *gate;
    int slot_prev, slot_cur, slot_next;
};
DW_HD StepCtx make_step_ctx(const TaskParams &C, const TaskBuffers &T, int e) {
    StepCtx K;
    K.nz.rec = T.noise ? T.noise + (size_t)DW_NOISE_WORDS * e : nullptr;
    K.nz.seed = C.seed; K.nz.env = (unsigned int)e; K.nz.step = (unsigned long long)T.step; K.nz.stream = 0;
    K.period = (float)(3599 * 0.0005); K.cdt = 0.0005f; K.dt = C.phys.dt; K.cdt_d = 0.0005;
    K.gate = reinterpret_cast<long long *>(T.b->gate_acc);
    K.slot_prev = (int)((T.step + 2) % 3); K.slot_cur = (int)(T.step % 3); K.slot_next = (int)((T.step + 1) % 3);
    return K;
}
// population statistics of the previous step (tasks/dyros_dynamic_walk.py:489): is the perturbation gate open
template <class GP>
DW_HD int gate_open_at(const TaskParams &C, const StepCtx &K, GP gate) {          // (gate: the gate words, through whatever pointer type the caller holds)
    int open = C.force_perturb_start;
    if (!open && C.perturb) {
        const long long latch = gate[GATE_LATCH];
        long long se = 0, sc = 0;
        for (int k = 0; k < GATE_BUCKETS; ++k) {
            se += gate[(K.slot_prev * GATE_BUCKETS + k) * 2];
            sc += gate[(K.slot_prev * GATE_BUCKETS + k) * 2 + 1];
        }
        const double n = (double)C.num_envs;
        const double mean_epi = (double)se / n, mean_crm = (double)sc / 4294967296.0 / n;
        open = latch ? 1 : (mean_epi > (double)(C.max_episode_length - C.pert_period_f) && mean_crm > 0.165);
    }
    return open;
}
DW_HD int gate_open(const TaskParams &C, const StepCtx &K) { return gate_open_at(C, K, K.gate); }
// VecTask counters and the env's mass into the flag / scratch words the post regions read
template <class LT>
DW_HD void load_counters(LT &S, const DwBuffers &B, int e) {
    const long long p = B.progress_buf[e], rb = B.randomize_buf[e];
    S.flags[5] = (int)p;
    S.flags[4] = (int)(rb > 0x7ffffffe ? 0x7ffffffe : rb);
    S.scratch[3] = B.total_mass[e];
}
DW_HD float clamp_action(const float *actions, int e, int l) {
    float a = fminf(fmaxf(actions[DW_NUM_ACT * e + l], -1.0f), 1.0f);
    if (l == 12) a = (a > 0 ? 1.0f : 0.0f) * a;
    return a;
}

// Regions P1, P2 (pre_physics_step up to the substep loop).  Expects S.es, S.act, S.flags[6] (gate).  FUSED: the warm-start
// impulses are copied into the physics block (the split pipeline's physics kernel reads them from the record itself).
template <bool FUSED, class W, class LT>
DW_HD void task_p1p2(const W &wave, LT &S, const DevModel &M, const TaskParams &C, const TaskBuffers &T, int e, const StepCtx &K) {
    const DwBuffers &B = *T.b;
    const NoiseSrc &nz = K.nz;
    const float period = K.period, cdt = K.cdt;
    const double cdt_d = K.cdt_d;
    long long *gate = K.gate;
    // ---- P1: clamp actions, mocap phase, perturbation gate and schedule (scalar work on single lanes) ----
    wave.par([&](int l) {
        if (l < DW_NUM_ACT) {
            const float a = S.act[l];
            S.es[DW_ES_ACTIONS + l] = a;
            B.action_history[((size_t)e * DW_HIST_SLOTS + ESI(DW_ES_HIST_HEAD)) * DW_NUM_ACT + l] = a;
        }
        if (l == 32) {
            const float time = S.es[DW_ES_TIME];
            const int init_idx = ESI(DW_ES_INIT_MOCAP);
            const float local_time = remainder_t(time, period);
            S.scratch[0] = remainder_t(local_time + (float)init_idx * cdt, period);
            const int midx = (int)(((long long)init_idx + (long long)divs(C.gpu_div, local_time, cdt_d)) % 3599);
            ESI(DW_ES_MOCAP_IDX) = midx;
        }
        if (l == 33) {
            const int open = S.flags[6];
            if (open) {
                ESI(DW_ES_PERT_START) = 1;
                if (!C.force_perturb_start) gate[GATE_LATCH] = 1;
            }
            float px = 0.0f, py = 0.0f;
            if (ESI(DW_ES_PERT_START)) {
                if (remainder_t(S.es[DW_ES_EPI_LEN], C.pert_period_f) == (float)ESI(DW_ES_PERT_TIMING)) {
                    ESI(DW_ES_PERT_ON) = 1;
                    int imp = 50 + (int)(noise_word(nz, DW_NZ_PERT + 0) * 200.0f);
                    if (imp > 249) imp = 249;
                    int dur = C.pert_dur_lo + (int)(noise_word(nz, DW_NZ_PERT + 1) * (float)(C.pert_dur_hi - C.pert_dur_lo));
                    if (dur > C.pert_dur_hi - 1) dur = C.pert_dur_hi - 1;
                    ESI(DW_ES_IMPULSE) = imp;
                    ESI(DW_ES_PERT_DURATION) = dur;
                    S.es[DW_ES_MAGNITUDE] = (float)imp / ((float)dur * C.dt_policy_f);
                    S.es[DW_ES_PHASE] = noise_word(nz, DW_NZ_PERT + 2) * 2.0f * (float)3.14159265358979;
                }
                if (ESI(DW_ES_PERT_ON)) {
                    ESI(DW_ES_PERT_COUNT) += 1;
                    px = S.es[DW_ES_MAGNITUDE] * cosf(S.es[DW_ES_PHASE]);
                    py = S.es[DW_ES_MAGNITUDE] * sinf(S.es[DW_ES_PHASE]);
                }
                if (ESI(DW_ES_PERT_COUNT) == ESI(DW_ES_PERT_DURATION)) {
                    ESI(DW_ES_PERT_ON) = 0;
                    ESI(DW_ES_PERT_COUNT) = 0;
                }
            }
            S.scratch[1] = px;
            S.scratch[2] = py;
        }
        if (FUSED) { if (l >= 40 && l < 40 + 24) S.warm[l - 40] = S.es[DW_ES_WARM + (l - 40)]; }
    });
    // ---- P2: mocap target (cubic between two table rows), leg torques from the actions ----
    wave.par([&](int l) {
        const int midx = ESI(DW_ES_MOCAP_IDX);
        const float *row0 = T.mocap + (size_t)midx * DW_MOCAP_COLS, *row1 = row0 + DW_MOCAP_COLS;
        const float ltp = S.scratch[0];
        if (l < 35) {
            const float v = cubic_t(ltp, row0[0], row1[0], row0[1 + l], row1[1 + l]);
            if (l < 33) S.es[DW_ES_TARGET_QPOS + l] = v;
            else S.es[DW_ES_TARGET_FORCE + (l - 33)] = v;
        }
        if (l >= 40 && l < 52) {
            const int i = l - 40;
            S.es[DW_ES_ACTION_TORQUE + i] = S.act[i] * S.es[DW_ES_MOTOR_SCALE + i] * M.action_high[i];
        }
    });
}

// Regions Q1..Q6 (post_physics_step).  Expects S.es, S.root, S.q, S.qd, S.contact, S.act, S.flags[4], S.flags[5], S.scratch[3],
// S.flags[0..3] = 0.  Returns whether the env was reset or hit the non-finite guard (its Gym state changed).
template <bool FUSED, class W, class LT>
DW_HD int task_post(const W &wave, LT &S, const DevModel &M, const TaskParams &C, const TaskBuffers &T, int e, const StepCtx &K) {
    const DwBuffers &B = *T.b;
    const NoiseSrc &nz = K.nz;
    const float period = K.period;
    const double cdt_d = K.cdt_d;
    long long *gate = K.gate;
    const int slot_cur = K.slot_cur, slot_next = K.slot_next;
    // ---- Q1: clocks, VecTask counters, non-finite guard ----
    wave.par([&](int l) {
        if (l == 40) {
            S.es[DW_ES_EPI_LEN] += 1.0f;
            float time = S.es[DW_ES_TIME];
            time = time + C.dt_policy_f;
            time = time + C.clock_gain_f * S.act[12];
            S.es[DW_ES_TIME] = time;
            const long long p = S.flags[5];
            B.timeout_buf[e] = ((float)(p + (C.timeout_fix ? 1 : 0)) >= C.max_episode_length - 1.0f) ? 1 : 0;
            B.progress_buf[e] = p + 1;
            S.flags[5] = (int)(p + 1);
            const int rb = S.flags[4] + 1;           // steps since the last parameter randomisation, saturating
            B.randomize_buf[e] = rb;
            S.flags[4] = rb;
        }
        if (FUSED) { if (l < 24) S.es[DW_ES_WARM + l] = S.warm[l]; }
        bool bad = false;
        if (l < 13) bad |= !finitef(S.root[l]);
        if (l < ND) bad |= !finitef(S.q[l]) || !finitef(S.qd[l]);
        if (bad) S.flags[1] = 1;
    });
    if (uniform(S.flags[1])) {
        wave.par([&](int l) {
            if (l < 13) S.root[l] = (l == 2) ? C.initial_height : (l == 6 ? 1.0f : 0.0f);
            if (l < ND) { S.q[l] = 0.0f; S.qd[l] = 0.0f; }
            for (int i = l; i < DW_NUM_BODIES * 3; i += 64) S.contact[i] = 0.0f;
            if (l == 40) ESI(DW_ES_NAN_RESETS) += 1;
        });
    }

    // ---- Q2: reward terms, one term (or one reduction) per lane ----
    wave.par([&](int l) {
        const int LF = M.left_foot_gym, RF = M.right_foot_gym;
        if (l < DW_NUM_BODIES) {
            if (l != LF && l != RF && norm_t(&S.contact[3 * l], 3) > 1.0f) S.flags[2] = 1;
        }
        if (l == 40) {
            const float aerr = fabsf(quat_err(&S.root[3]));
            S.rterm[14] = aerr;
            S.rterm[0] = 0.3f * expf(-13.2f * aerr);
        }
        if (l >= 8 && l < 32) {
            // three 33-element norms at once: lanes 8..15 / 16..23 / 24..31 hold the 8 fused accumulators of
            // torch's CPU reduction for qpos error / qvel / qacc; the combine happens in Q3
            const int which = (l - 8) >> 3, a = (l - 8) & 7;
            float acc = 0.0f;
            for (int d = 0; d < 32; d += 8) {
                const int j = d + a;
                const float x = which == 0 ? S.es[DW_ES_TARGET_QPOS + j] - S.q[j]
                              : (which == 1 ? 0.0f - S.qd[j] : S.qd[j] - S.es[DW_ES_PRE_QVEL + j]);
                acc = fmaf(x, x, acc);
            }
            S.normed[(which << 3) + a] = acc; // normed[] is free until Q4
        }
        if (l == 44) {
            S.rterm[4] = 0.05f * expf(-0.01f * norm_fn([&](int i) { return S.es[DW_ES_ACTIONS + i] * 333.0f; }, 12));
        }
        if (l == 45) {
            S.rterm[5] = 0.6f * expf((-0.01f * 1.0f) * norm_fn([&](int i) { return (S.es[DW_ES_ACTIONS + i] - S.es[DW_ES_ACTIONS_PRE + i]) * 333.0f; }, 12));
        }
        if (l == 46) {
            const float dv[2] = {S.es[DW_ES_TARGET_VEL] - S.root[7], S.es[DW_ES_TARGET_VEL + 1] - S.root[8]};
            const float n = norm_t(dv, 2);
            S.rterm[6] = 0.3f * expf(-3.0f * (n * n));
        }
        if (l == 47) {
            const float *lf = &S.contact[3 * LF], *rf = &S.contact[3 * RF];
            const float *lfp = &S.es[DW_ES_FOOT_FORCE_PRE], *rfp = &S.es[DW_ES_FOOT_FORCE_PRE + 3];
            float dl[3], dr[3];
            for (int i = 0; i < 3; ++i) { dl[i] = lf[i] - lfp[i]; dr[i] = rf[i] - rfp[i]; }
            S.rterm[9] = 0.2f * expf((-0.01f * 1.0f) * (norm_t(dl, 3) + norm_t(dr, 3)));
            const bool lcon = lf[2] > 1.0f, rcon = rf[2] > 1.0f;
            const int idx = ESI(DW_ES_MOCAP_IDX);
            const bool DSP = (3300 <= idx && idx < 3600) || (idx < 300) || (1500 <= idx && idx < 2100);
            const bool RSSP = 300 <= idx && idx < 1500;
            const bool LSSP = 2100 <= idx && idx < 3300;
            float fcr = 0.0f;
            if (DSP && rcon && lcon) fcr = 0.2f;
            if (RSSP && rcon && !lcon) fcr = 0.2f;
            if (LSSP && !rcon && lcon) fcr = 0.2f;
            S.rterm[8] = fcr;
            S.es[DW_ES_CRS] = S.es[DW_ES_CRS] + fcr;
            S.rterm[10] = 0.0f;
            const float tm = S.scratch[3];
            const float thr = (float)(1.4 * 9.81) * tm;
            const bool th = (lf[2] > thr) || (rf[2] > thr);
            S.rterm[11] = th ? -0.2f * 1.0f : 0.0f;
            const float cl = fmaxf(lf[2] - thr, 0.0f), cr = fmaxf(rf[2] - thr, 0.0f);
            const float pen = 0.1f * expf(-0.007f * (norm_t(&cl, 1) + norm_t(&cr, 1)));
            S.rterm[3] = th ? pen : 0.1f * 1.0f;
            const float thd = ((float)(0.2 * 9.81) * tm) / 1.0f;
            const bool dd = (fabsf(lf[2] - lfp[2]) > thd) || (fabsf(rf[2] - rfp[2]) > thd);
            S.rterm[12] = dd ? -0.05f * 1.0f : 0.0f;
            const float ws = divs(C.gpu_div, tm, 104.48);
            const float tl = 0.1f * expf(-0.001f * fabsf(lf[2] + ws * S.es[DW_ES_TARGET_FORCE]));
            const float tr = 0.1f * expf(-0.001f * fabsf(rf[2] + ws * S.es[DW_ES_TARGET_FORCE + 1]));
            S.rterm[13] = tl + tr;
        }
    });
    // ---- Q2b: combine the partial sums (lanes in order, then the 33rd element fused), exp ----
    wave.par([&](int l) {
        if (l >= 41 && l < 44) {
            const int which = l - 41;
            float b0 = S.normed[which << 3];
            for (int a = 1; a < 8; ++a) b0 = b0 + S.normed[(which << 3) + a];
            const float x = which == 0 ? S.es[DW_ES_TARGET_QPOS + 32] - S.q[32]
                          : (which == 1 ? 0.0f - S.qd[32] : S.qd[32] - S.es[DW_ES_PRE_QVEL + 32]);
            b0 = fmaf(x, x, b0);
            const float n = sqrtf(b0);
            const float coef = which == 0 ? 0.35f : 0.05f, rate = which == 0 ? -2.0f : (which == 1 ? -0.01f : -20.0f);
            S.rterm[which == 0 ? 1 : (which == 1 ? 2 : 7)] = coef * expf(rate * (n * n));
        }
    });
    // ---- Q3: total reward, termination ----
    wave.par([&](int l) {
        const bool collision = S.flags[2] != 0;
        const float aerr = S.rterm[14];
        if (l < 14) B.stacked_rewards[(size_t)DW_NUM_REW * e + l] = collision ? 1.0f * C.death_cost : S.rterm[l];
        if (l == 14) B.stacked_rewards[(size_t)DW_NUM_REW * e + 14] = ESI(DW_ES_PERT_START) ? 1.0f : 0.0f;
        if (l == 40) {
            const float *r = S.rterm;
            float total = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + r[8] + r[9] + r[10] + r[11] + r[12] + r[13];
            if (collision) total = 1.0f * C.death_cost;
            if (aerr > 0.5f) total = 1.0f * C.death_cost;
            B.rew_buf[e] = total;
            int reset = aerr > 0.5f ? 1 : 0;
            if ((float)S.flags[5] >= C.max_episode_length - 1.0f) reset = 1;
            if (collision) reset = 1;
            if (S.flags[1]) reset = 1;
            B.reset_buf[e] = reset;
            S.flags[3] = reset;
            float ret = S.es[DW_ES_EPI_RETURN] + total;
            if (reset) {
                S.es[DW_ES_LAST_RETURN] = ret;
                ESI(DW_ES_EPISODES) += 1;
                ret = 0.0f;
            }
            S.es[DW_ES_EPI_RETURN] = ret;
        }
    });
    const int did_reset = uniform(S.flags[3]);
    if (did_reset) reset_region(wave, S, M, C, B, nz, e);

    // ---- Q4: 37-d observation, normalisation, newest history slot ----
    wave.par([&](int l) {
        if (l < DW_NUM_OBS1) {
            float o;
            if (l < 3) {
                const float x = S.root[3], y = S.root[4], z = S.root[5], w = S.root[6];
                const float m00 = w * w + x * x - y * y - z * z;
                const float m01 = 2 * x * y - 2 * w * z;
                const float m10 = 2 * x * y + 2 * w * z;
                const float m11 = w * w - x * x + y * y - z * z;
                const float m20 = 2 * x * z - 2 * w * y;
                const float m21 = 2 * y * z + 2 * w * x;
                const float m22 = w * w - x * x - y * y + z * z;
                const float cy = sqrtf(m00 * m00 + m10 * m10);
                const bool cond = cy > (float)(2.220446049250313e-16 * 4);
                // one atan2 call for the three lanes: (num, den) per Euler angle (mat2euler, torch_utils.py:248-269)
                const float num = l == 0 ? m21 : (l == 1 ? -m20 : (cond ? m10 : -m01));
                const float den = l == 0 ? m22 : (l == 1 ? cy : (cond ? m00 : m11));
                o = atan2f(num, den);
                if (l == 0 && !cond) o = 0.0f;
                o = o + S.es[DW_ES_QUAT_BIAS + l];
            } else if (l < 15) {
                o = S.es[DW_ES_QPOS_NOISE + (l - 3)] + S.es[DW_ES_QPOS_BIAS + (l - 3)];
            } else if (l < 27) {
                o = S.es[DW_ES_QVEL_NOISE + (l - 15)];
            } else if (l < 29) {
                const float time2idx = divs(C.gpu_div, remainder_t(S.es[DW_ES_TIME], period), cdt_d);
                const float phase = divs(C.gpu_div, remainder_t((float)ESI(DW_ES_INIT_MOCAP) + time2idx, 3599.0f), 3599.0);
                const float ang = (float)(2 * 3.14159265358979) * phase;
                float sn, cs;
                sincosf(ang, &sn, &cs);
                o = l == 27 ? sn : cs;
            } else if (l < 31) {
                o = S.es[DW_ES_TARGET_VEL + (l - 29)];
            } else {
                o = S.root[7 + (l - 31)] + (noise_word(nz, DW_NZ_VEL + (l - 31)) * 0.05f - 0.025f);
            }
            const float nrm = (o - M.obs_mean[l]) / M.obs_inv_std_den[l];
            S.normed[l] = nrm;
            float *oh = B.obs_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_OBS1;
            if (S.es[DW_ES_EPI_LEN] == 0.0f) {
                for (int s = 0; s < DW_HIST_SLOTS; ++s) oh[s * DW_NUM_OBS1 + l] = nrm;
            } else {
                oh[ESI(DW_ES_HIST_HEAD) * DW_NUM_OBS1 + l] = nrm;
            }
        }
    });
    // ---- Q5: 487-d observation buffer from the ring taps ----
    wave.par([&](int l) {
        const int head = (ESI(DW_ES_HIST_HEAD) + 1) % DW_HIST_SLOTS;       // position of the oldest slot after this step's push
        const int newest = ESI(DW_ES_HIST_HEAD);
        const bool fill = S.es[DW_ES_EPI_LEN] == 0.0f;
        const float *oh = B.obs_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_OBS1;
        const float *ah = B.action_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_ACT;
        float *ob = B.obs_buf + (size_t)DW_NUM_OBS * e;
        // all tap loads are issued before the first store (the two buffers may alias as far as the compiler knows, so a
        // load-store-load-store loop would pay one memory round trip per tap)
        constexpr int NO = (DW_NUM_OBS1 * DW_NUM_HIS + 63) / 64, NA = (DW_NUM_ACT * (DW_NUM_HIS - 1) + 63) / 64;
        float vo[NO], va[NA];
        for (int j = 0; j < NO; ++j) {
            const int f = l + 64 * j;
            vo[j] = 0.0f;
            if (f < DW_NUM_OBS1 * DW_NUM_HIS) {
                const int i = f / DW_NUM_OBS1, k = f - i * DW_NUM_OBS1;
                const int slot = (head + DW_NUM_SKIP * (i + 1) - 1) % DW_HIST_SLOTS;
                vo[j] = (fill || slot == newest) ? S.normed[k] : oh[slot * DW_NUM_OBS1 + k];
            }
        }
        for (int j = 0; j < NA; ++j) {
            const int f = l + 64 * j;
            va[j] = 0.0f;
            if (f < DW_NUM_ACT * (DW_NUM_HIS - 1)) {
                const int i = f / DW_NUM_ACT, k = f - i * DW_NUM_ACT;
                const int slot = (head + DW_NUM_SKIP * (i + 1)) % DW_HIST_SLOTS;
                va[j] = did_reset ? 0.0f : (slot == newest ? S.act[k] : ah[slot * DW_NUM_ACT + k]);
            }
        }
        for (int j = 0; j < NO; ++j) { const int f = l + 64 * j; if (f < DW_NUM_OBS1 * DW_NUM_HIS) ob[f] = vo[j]; }
        for (int j = 0; j < NA; ++j) { const int f = l + 64 * j; if (f < DW_NUM_ACT * (DW_NUM_HIS - 1)) ob[DW_NUM_OBS1 * DW_NUM_HIS + f] = va[j]; }
    });
    // ---- Q6: late updates (tasks/dyros_dynamic_walk.py:560-563), ring head, gate statistics ----
    wave.par([&](int l) {
        if (l < ND) S.es[DW_ES_PRE_QVEL + l] = S.qd[l];
        if (l < 12) S.es[DW_ES_ACTION_TORQUE_PRE + l] = S.es[DW_ES_ACTION_TORQUE + l];
        if (l < DW_NUM_ACT) S.es[DW_ES_ACTIONS_PRE + l] = S.es[DW_ES_ACTIONS + l];
        if (l >= 56 && l < 62) {
            const int i = l - 56, gy = i < 3 ? M.left_foot_gym : M.right_foot_gym;
            S.es[DW_ES_FOOT_FORCE_PRE + i] = S.contact[3 * gy + (i % 3)];
        }
        if (l == 40) ESI(DW_ES_HIST_HEAD) = (ESI(DW_ES_HIST_HEAD) + 1) % DW_HIST_SLOTS;
        if (l == 41 && C.perturb && !C.force_perturb_start) {
            const float el = S.es[DW_ES_EPI_LEN_LOG], cm = S.es[DW_ES_CRM];
            const int bk = e % GATE_BUCKETS;
            long long de, dc = 0;
            if (finitef(el) && finitef(cm)) { de = (long long)el; dc = (long long)llrintf(cm * 4294967296.0f); }
            else de = -((long long)1 << 62);
#if defined(__HIPCC__)
            atomicAdd(reinterpret_cast<unsigned long long *>(&gate[(slot_cur * GATE_BUCKETS + bk) * 2]), (unsigned long long)de);
            atomicAdd(reinterpret_cast<unsigned long long *>(&gate[(slot_cur * GATE_BUCKETS + bk) * 2 + 1]), (unsigned long long)dc);
#else
            gate[(slot_cur * GATE_BUCKETS + bk) * 2] += de;
            gate[(slot_cur * GATE_BUCKETS + bk) * 2 + 1] += dc;
#endif
            gate[(slot_next * GATE_BUCKETS + bk) * 2] = 0;
            gate[(slot_next * GATE_BUCKETS + bk) * 2 + 1] = 0;
        }
    });
    return did_reset | uniform(S.flags[1]);
}

// The fused wave-per-env step (first-generation kernel, DwConfig.pipeline = 1; flat ground and height fields)
template <bool TERRAIN, class W>
DW_HD void step_env(const W &wave, Lds &S, const DevModel &M, const TaskParams &C, const TaskBuffers &T, int e) {
    const DwBuffers &B = *T.b;
    const StepCtx K = make_step_ctx(C, T, e);
    const NoiseSrc &nz = K.nz;
    const float dt = K.dt;
    const TreeUniform TU = make_tree_uniform(M);

    // One region issues every global load whose address does not depend on this step's arithmetic -- tree tables,
    // the env's record and state, its actions, counters, mass and the gate sums -- so the wave waits for HBM/L2 once
    // here instead of once per phase that needs a scalar.
    wave.par([&](int l) {
        stage_tree_lane(l, S, M);
        load_env_lane(l, S, C, B, e, true);
        if (l < DW_NUM_ACT) S.act[l] = clamp_action(T.actions, e, l);
        if (l == 33) S.flags[6] = gate_open(C, K);
        if (l == 34) load_counters(S, B, e);
    });
    if (C.freeze_physics) {      // debug mode: simulate() is the identity, so the net contact forces are an input too
        wave.par([&](int l) {
            for (int i = l; i < DW_NUM_BODIES * 3; i += 64) S.contact[i] = B.contact_forces[(size_t)DW_NUM_BODIES * 3 * e + i];
        });
    }
    task_p1p2<true>(wave, S, M, C, T, e, K);

    // ---- P3: two physics substeps with the actuator model around them ----
    // unrolled on purpose: as a loop, ~45 lane-constant model loads were hoisted out of it and kept live across the
    // whole substep body, which at the 168-register cap meant spilling them (188 B/lane of scratch)
#if !defined(DW_ROLL_SUBSTEPS)
#pragma unroll
#else
#pragma clang loop unroll(disable)
#endif
    for (int sub = 0; sub < 2; ++sub) {
        wave.par([&](int l) {
            if (l < 12) {
                // torque FIFO, column l (tasks/dyros_dynamic_walk.py:511-519)
                float col[DW_ALOG_SLOTS];
                for (int s = 0; s < DW_ALOG_SLOTS - 1; ++s) col[s] = S.es[DW_ES_ACTION_LOG + 12 * (s + 1) + l];
                col[DW_ALOG_SLOTS - 1] = S.es[DW_ES_ACTION_TORQUE + l];
                for (int s = 0; s < DW_ALOG_SLOTS; ++s) S.es[DW_ES_ACTION_LOG + 12 * s + l] = col[s];
                int sl = ESI(DW_ES_SIMUL_LEN) + 1;
                if (sl > DW_ALOG_SLOTS) sl = DW_ALOG_SLOTS;
                const int dl = ESI(DW_ES_DELAY_IDX);
                const int src = sl > dl ? dl : DW_ALOG_SLOTS - sl;
                float t = col[0];
                for (int s = 1; s < DW_ALOG_SLOTS; ++s) t = (s == src) ? col[s] : t;
                S.tau[l] = t;
            } else if (l < ND) {
                S.tau[l] = M.kp[l] * (S.es[DW_ES_TARGET_QPOS + l] - S.q[l]) + M.kv[l] * (-S.qd[l]);
            }
            if (l == 40) { S.push[0] = sub == 0 ? S.scratch[1] : 0.0f; S.push[1] = sub == 0 ? S.scratch[2] : 0.0f; }
        });
        if (!C.freeze_physics) physics_substep<TERRAIN>(wave, S, M, C.phys, TU);
        wave.par([&](int l) {
            if (l < ND) {
                const float n = noise_word(nz, DW_NZ_ENC + ND * sub + l);
                const float qn = S.q[l] + fminf(fmaxf(n, -0.00016f), 0.00016f);
                S.es[DW_ES_QVEL_NOISE + l] = C.gpu_div ? (qn - S.es[DW_ES_QPOS_PRE + l]) * C.inv_dt_f : (qn - S.es[DW_ES_QPOS_PRE + l]) / dt;
                S.es[DW_ES_QPOS_NOISE + l] = qn;
                S.es[DW_ES_QPOS_PRE + l] = qn;
            }
            if (l == 40) {
                int sl = ESI(DW_ES_SIMUL_LEN) + 1;
                if (sl > DW_ALOG_SLOTS) sl = DW_ALOG_SLOTS;
                ESI(DW_ES_SIMUL_LEN) = sl;
            }
        });
    }

    task_post<true>(wave, S, M, C, T, e, K);
    store_env(wave, S, B, e, true, true);
}

// reset_done path (tasks/base/vec_task.py:376-391 -> reset_idx): one env
template <class W, class LT>
DW_HD void reset_only_env(const W &wave, LT &S, const DevModel &M, const TaskParams &C, const TaskBuffers &T, int e) {
    const DwBuffers &B = *T.b;
    NoiseSrc nz;
    nz.rec = T.noise ? T.noise + (size_t)DW_NOISE_WORDS * e : nullptr;
    nz.seed = C.seed; nz.env = (unsigned int)e; nz.step = (unsigned long long)T.step; nz.stream = 1;
    // reset_region rewrites root, q and qd, but with the terrain curriculum it first reads root[0..1] (the distance walked
    // from the tile origin decides the level change): the base state is loaded too, next to the record and the contact forces
    wave.par([&](int l) {
        if (l < 13) S.root[l] = B.root_states[13 * e + l];
        for (int i = l; i < DW_ES_WORDS; i += 64) S.es[i] = B.env_state[(size_t)DW_ES_WORDS * e + i];
        for (int i = l; i < DW_NUM_BODIES * 3; i += 64) S.contact[i] = B.contact_forces[(size_t)DW_NUM_BODIES * 3 * e + i];
        if (l == 40) {
            const long long rb = B.randomize_buf[e];
            S.flags[4] = (int)(rb > 0x7fffffff ? 0x7fffffff : rb);
        }
    });
    reset_region(wave, S, M, C, B, nz, e);
    store_env(wave, S, B, e, true, true);
}

// Gym-boundary substep: tau [N,33], push [N,2] or nullptr
template <bool TERRAIN, class W>
DW_HD void simulate_env(const W &wave, Lds &S, const DevModel &M, const TaskParams &C, const DwBuffers &B,
                        const float *tau, const float *push, int e) {
    stage_tree(wave, S, M);
    load_env(wave, S, C, B, e, false);
    wave.par([&](int l) {
        if (l < ND) S.tau[l] = tau[ND * e + l];
        if (l == 40) { S.push[0] = push ? push[2 * e] : 0.0f; S.push[1] = push ? push[2 * e + 1] : 0.0f; }
        if (l < 24) S.warm[l] = B.env_state ? B.env_state[(size_t)DW_ES_WORDS * e + DW_ES_WARM + l] : 0.0f;
    });
    physics_substep<TERRAIN>(wave, S, M, C.phys, make_tree_uniform(M));
    wave.par([&](int l) {
        if (l < 24 && B.env_state) B.env_state[(size_t)DW_ES_WORDS * e + DW_ES_WARM + l] = S.warm[l];
    });
    store_env(wave, S, B, e, false, true);
}

#undef ESI

}  // namespace dw

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
