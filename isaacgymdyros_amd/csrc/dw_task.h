// dw_task.h -- what the step kernels of every layout share of the DyrosDynamicWalk task logic, and reset_idx for one env.
//
// The parameter block (TaskParams, DevParams), the counter-based generator (Philox4x32-10 and the draws built on it), the
// torch-flavoured scalar helpers (division, remainder, norm in torch's summation orders, the cubic, the quaternion error), the
// early-termination gate, and reset_idx of ONE env as wave regions (reset_region / reset_only_env: the kernel behind
// dw_reset_idx, one wavefront per listed env -- a rare, host-driven call; the resets inside a step are the step kernels' own,
// dw_oct_post.h / dw_lane_post.h).  fp contraction is OFF in this file: every fp32 operation is in the reference's order:
//   check_termination                   tasks/dyros_dynamic_walk.py:581-596 (quat_diff_rad: utils/torch_jit_utils.py:141-160)
//   reset_idx + dof-property DR         tasks/dyros_dynamic_walk.py:598-669,720-748; tasks/base/vec_task.py:519-733
//   cubic                               utils/torch_jit_utils.py:373-395
// (paths relative to python/IsaacGymEnvs/isaacgymenvs unless they start with python/).
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
    // the curriculum's logging columns (tasks/dyros_dynamic_walk.py:417-421: mean level of the envs of each terrain type): the step
    // kernel adds (envs << 32 | sum of their levels) to word [step % 3][bucket][type] and clears slot (step + 1) % 3, the last word
    // names the slot of the newest step; dw_terrain_log reads them.  Library-owned (dw_create), nullptr without a curriculum.
    unsigned long long *terrain_lvl_acc;
};
constexpr int LVL_BUCKETS = 16;      // terrain_lvl_acc: [3][LVL_BUCKETS][types] + 1 words
constexpr int LVL_BITS = 8;          // levels < 256 (check_config)
inline size_t lvl_acc_words(int types) { return 3 * (size_t)LVL_BUCKETS * types + 1; }

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
    for (int r = 0; r < 10; ++r) {          /*@trip:10*/
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
// both encoder draws of joint d in one generator call (the step kernels keep the second for the second substep)
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
// torch.norm over a contiguous fp32 row as torch's GPU reduce kernel sums it on ROCm (ATen/native/cuda/Reduce.cuh, reduction over
// the fastest dimension with fewer than 128 inputs per output; probed on the MI355X against torch 2.10, tools/probe_gpu_norm3.py:
// 100 % of 1 M rows for every row length the task uses (33, 12, 3, 2; also 6 and 13), for any base alignment and any number of
// rows): T = the largest power of two <= n (at most 32) threads share a row; thread t squares x[t], x[t + T], ... into separate
// accumulators and adds them in that order; the threads combine by shuffle-down with offsets 1, 2, 4, ... -- a balanced tree over
// t in index order; sqrt is correctly rounded.  (The CPU kernel's order is norm_t / norm_fn above; which one a build reproduces
// is DwConfig.torch_gpu_div, like the flavour of `tensor / python_scalar`.)
template <int LO, int LEN, class P>
DW_HD float norm_g_tree(P &p) {
    if constexpr (LEN == 1) {
        return p(LO);
    } else {
        const float a = norm_g_tree<LO, LEN / 2>(p);
        const float b = norm_g_tree<LO + LEN / 2, LEN / 2>(p);
        return a + b;
    }
}
template <int N, class F>
DW_HD float norm_g(F f) {
    constexpr int T = N >= 32 ? 32 : (N >= 16 ? 16 : (N >= 8 ? 8 : (N >= 4 ? 4 : (N >= 2 ? 2 : 1))));
    static_assert(N >= 1 && N <= 4 * T, "a thread holds at most four accumulators (vt0 = 4)");
    auto part = [&](int t) {
        const float x = f(t);
        float v = x * x;
        for (int k = t + T; k < N; k += T) { const float y = f(k); const float yy = y * y; v = v + yy; }
        return v;
    };
    return sqrtf(norm_g_tree<0, T>(part));
}
// the norm of N elements f(0) .. f(N - 1) in the order of torch's GPU kernel (gpu != 0) or of its CPU kernel
template <int N, class F>
DW_HD float norm_sel(int gpu, F f) {
#if defined(DW_NORM_GPU_ONLY)          // (A/B builds only: what the second flavour costs in code size, tools/tu_lib.sh)
    (void)gpu;
    return norm_g<N>(f);
#elif defined(DW_NORM_CPU_ONLY)
    (void)gpu;
    return norm_fn(f, N);
#else
    return gpu ? norm_g<N>(f) : norm_fn(f, N);
#endif
}
template <int N>
DW_HD float norm_sel_v(int gpu, const float *x) { return norm_sel<N>(gpu, [&](int i) { return x[i]; }); }

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
DW_HD float quat_err(const float *q, int gpu_norm = 0) {
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
    float n = norm_sel_v<3>(gpu_norm, v);
    if (n > 1.0f) n = 1.0f;
    return 2.0f * asinf(n);
}
DW_HD bool finitef(float x) { return fabsf(x) <= 3.4028234663852886e38f; }   // false for NaN and +-inf

#define ESI(off) (*reinterpret_cast<int *>(&S.es[(off)]))

// LDS block of dw_k_reset (reset_idx of listed envs, one wave per env): the task record and the Gym state of one env, 3.6 KB.
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
                const float distance = norm_sel_v<2>(C.gpu_div, d);
                const bool move_up = distance > C.terrain_half_length;
                const float need = norm_sel_v<2>(C.gpu_div, &S.es[DW_ES_TARGET_VEL]) * C.max_episode_length_s * 0.5f;
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


#undef ESI

}  // namespace dw

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
