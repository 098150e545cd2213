/*
 * dw_task.c -- CPU ORACLE, task logic of DyrosDynamicWalk.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates, one env at a time and one fp32 operation at a time (compiled with -ffp-contract=off, in the
 * order torch's eager CPU kernels apply them), what the reference computes around `gym.simulate`:
 *   VecTask.step                      tasks/base/vec_task.py:293-344
 *   pre_physics_step                  tasks/dyros_dynamic_walk.py:449-541   (+ cubic, utils/torch_jit_utils.py:373-395)
 *   post_physics_step                 tasks/dyros_dynamic_walk.py:543-563
 *   check_termination                 tasks/dyros_dynamic_walk.py:581-596   (+ quat_diff_rad, utils/torch_jit_utils.py:141-160;
 *                                     quat_mul / quat_conjugate, python/isaacgym/torch_utils.py:19-40,84-88)
 *   compute_humanoid_walk_reward      tasks/dyros_dynamic_walk.py:802-947
 *   reset_idx (+ DR of dof properties) tasks/dyros_dynamic_walk.py:598-669,720-748; tasks/base/vec_task.py:519-733
 *   compute_humanoid_walk_observations tasks/dyros_dynamic_walk.py:750-796  (+ quat2euler, python/isaacgym/torch_utils.py:227-273)
 * Pinned by golden vectors recorded from the reference's own Python (tests/golden/, generator
 * oracle/make_goldens.py).  Transcendentals come from glibc here and from SLEEF in torch's CPU kernels,
 * so exp/sin/cos/asin/atan2 results may differ from the goldens in the last bit; everything else is
 * bit-exact (tests state the tolerance per field).
 *
 * All randomness is an INPUT: word w of the env's noise record (layout DW_NZ_* in dyros_walk.h) comes
 * from the `noise` argument or, when that is NULL, from Philox4x32-10 keyed by cfg.seed with counter
 * (w, env, step_lo, step_hi | stream<<31) for the encoder words and (DW_NZ_UBLOCK + w/4, ...) for the uniform words.
 */
#include "dw_oracle.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ RNG */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

typedef struct {
    const float *rec;     /* injected record or NULL */
    uint64_t seed;
    uint32_t env;
    uint64_t step;
    uint32_t stream;
} Noise;

static float noise_word(const Noise *nz, int w) {
    if (nz->rec) return nz->rec[w];
    /* The two encoder draws of a joint (one per substep) share ONE Philox block: counter word = the joint's word of the
     * first substep, outputs 0,1 for the first substep and 2,3 for the second (half the generator calls of the hot path). */
    const int pair = (w < DW_NZ_VEL && w >= DW_NZ_ENC + DW_NUM_DOF) ? 1 : 0;
    /* The uniform words (w >= DW_NZ_VEL) come four to a block: counter word = DW_NZ_UBLOCK + w / 4, output w % 4 (a reset
     * draws its 32 words DW_NZ_QPOS_BIAS .. DW_NZ_PTIMING from eight blocks). */
    const int cw = w >= DW_NZ_VEL ? DW_NZ_UBLOCK + (w >> 2) : (pair ? w - DW_NUM_DOF : w);
    uint32_t c[4] = {(uint32_t)cw, nz->env, (uint32_t)nz->step, (uint32_t)(nz->step >> 32) | (nz->stream << 31)};
    philox4x32_10(c, (uint32_t)nz->seed, (uint32_t)(nz->seed >> 32));
    if (w < DW_NZ_VEL) {   /* encoder noise ~ N(0, 0.00016/3): Box-Muller */
        float u1 = (float)((c[2 * pair] >> 8) + 1u) * 5.9604644775390625e-08f;   /* (0,1] */
        float u2 = (float)(c[2 * pair + 1] >> 8) * 5.9604644775390625e-08f;      /* [0,1) */
        float z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
        return z * (float)(0.00016 / 3.0);
    }
    return (float)(c[w & 3] >> 8) * 5.9604644775390625e-08f;              /* U[0,1) */
}

/* ------------------------------------------------------------------ torch-flavoured scalar helpers */
/* tensor / python_scalar: division by float(s) in torch's CPU kernels, multiplication by float(1.0/s) (reciprocal
 * formed in double) in its GPU kernels -- probed on MI355X: x / 0.0005 == x * 2000.0f */
static inline float divs_(int recip, float x, double s) { return recip ? x * (float)(1.0 / s) : x / (float)s; }
#define divs(x, s) divs_(h->cfg.torch_gpu_div, (x), (s))

/* torch.remainder for floats: fmod, then shifted into the sign of the divisor */
static inline float remainder_t(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0 && ((b < 0) != (m < 0))) m += b;
    return m;
}

/* torch.norm(x, dim=-1) of a contiguous fp32 row on the CPU: 8-lane FMA accumulation over full blocks,
 * lanes added in order, tail in 4-wide non-fused chunks, then fused scalars (probed against torch 2.10) */
static float norm_t(const float *x, int n) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int d = 0;
    for (; d < n - (n % 8); d += 8)
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(x[d + l], x[d + l], acc[l]);
    float b0 = acc[0];
    for (int l = 1; l < 8; ++l) b0 = b0 + acc[l];
    for (; d + 4 <= n; d += 4)
        for (int l = 0; l < 4; ++l) { float p = x[d + l] * x[d + l]; b0 = b0 + p; }
    for (; d < n; ++d) b0 = fmaf(x[d], x[d], b0);
    return sqrtf(b0);
}

/* the same row as torch's GPU reduce kernel sums it on ROCm (ATen/native/cuda/Reduce.cuh, reduction over the fastest dimension, fewer
 * than 128 inputs per output; probed on an MI355X against torch 2.10, tools/probe_gpu_norm3.py: every bit of 1 M rows for n = 33, 12,
 * 6, 13, 3, 2, any base alignment, any number of rows): T = the largest power of two <= n (at most 32) threads share the row, thread
 * t squares x[t], x[t + T], ... into separate accumulators and adds them in order, the threads combine by shuffle-down with offsets
 * 1, 2, 4, ...; sqrt is correctly rounded */
static float norm_g(const float *x, int n) {
    int T = 1;
    while (T * 2 <= n && T < 32) T *= 2;
    float part[32];
    for (int t = 0; t < T; ++t) {
        float v = x[t] * x[t];
        for (int k = t + T; k < n; k += T) { float yy = x[k] * x[k]; v = v + yy; }
        part[t] = v;
    }
    for (int off = 1; off < T; off *= 2)
        for (int t = 0; t + off < T; t += 2 * off) part[t] = part[t] + part[t + off];
    return sqrtf(part[0]);
}
/* which of the two a build reproduces is DwConfig.torch_gpu_div, like the flavour of `tensor / python_scalar` */
#define norm_s(x, n) (h->cfg.torch_gpu_div ? norm_g((x), (n)) : norm_t((x), (n)))

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* utils/torch_jit_utils.py:373-395 with x_dot_0 = x_dot_f = 0.0 */
static float cubic_t(float time, float t0, float tf, float x0, float xf) {
    float elapsed = time - t0;
    float total = tf - t0;
    float total2 = total * total;
    float total3 = total2 * total;
    float total_x = xf - x0;
    float c2 = (3.0f * total_x) / total2 - (0.0f / total) - (0.0f / total);
    float c3 = (-2.0f * total_x) / total3 + (0.0f / total2);
    float cub = x0 + 0.0f * elapsed + c2 * elapsed * elapsed + c3 * elapsed * elapsed * elapsed;
    float xt = x0;
    if (time > tf) xt = xf;
    if (t0 <= time && time <= tf) xt = cub;
    return xt;
}

/* quat_diff_rad(identity, q) */
static float quat_err(const float *q /* xyzw */, int gpu_norm) {
    const float x1 = 0, y1 = 0, z1 = 0, w1 = 1;
    const float x2 = -q[0], y2 = -q[1], z2 = -q[2], w2 = q[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    float x = qq - xx + (x1 + w1) * (x2 + w2);
    float y = qq - yy + (w1 - x1) * (y2 + z2);
    float z = qq - zz + (z1 + y1) * (w2 - x2);
    float v[3] = {x, y, z};
    float n = gpu_norm ? norm_g(v, 3) : norm_t(v, 3);
    if (n > 1.0f) n = 1.0f;            /* torch.clamp(max=1.0); NaN propagates through fminf differently, see below */
    return 2.0f * asinf(n);
}

/* ------------------------------------------------------------------ env record access */
#define ES(h, e) ((h)->buf.env_state + (size_t)DW_ES_WORDS * (e))
static inline int32_t *esi(float *es, int off) { return (int32_t *)(es + off); }

static void hist_push_obs(DwHandle *h, int e, float *es, const float *normed, int fill_all) {
    float *hist = h->buf.obs_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_OBS1;
    int32_t *head = esi(es, DW_ES_HIST_HEAD);
    /* the oldest slot becomes the newest; ring positions are shared with the action history, whose push
     * happened in pre_physics_step of the same step and already advanced nothing: the head moves here. */
    int pos = *head;
    memcpy(hist + pos * DW_NUM_OBS1, normed, sizeof(float) * DW_NUM_OBS1);
    if (fill_all)
        for (int s = 0; s < DW_HIST_SLOTS; ++s) memcpy(hist + s * DW_NUM_OBS1, normed, sizeof(float) * DW_NUM_OBS1);
    *head = (pos + 1) % DW_HIST_SLOTS;
}

/* reset_idx for one env (tasks/dyros_dynamic_walk.py:598-669) */
static void reset_env(DwHandle *h, int e, const Noise *nz) {
    const DwConfig *cfg = &h->cfg;
    const DwBuffers *b = &h->buf;
    float *es = ES(h, e);
    /* domain randomisation of the dof properties (vec_task.py:540-544,655-721; cfg yaml :103-115) */
    if ((cfg->randomize_dof_on_reset || cfg->randomize_friction_on_reset) && b->randomize_buf[e] >= 1) {
        for (int j = 0; cfg->randomize_dof_on_reset && j < DW_NUM_DOF; ++j) {
            float ud = noise_word(nz, DW_NZ_DR_DAMP + j), ua = noise_word(nz, DW_NZ_DR_ARM + j);
            float sd = cfg->dr_damping_add[0] + ud * (cfg->dr_damping_add[1] - cfg->dr_damping_add[0]);
            float sa = cfg->dr_armature_scale[0] + ua * (cfg->dr_armature_scale[1] - cfg->dr_armature_scale[0]);
            b->dof_damping[DW_NUM_DOF * e + j] = h->nominal_damping[j] + sd;
            b->dof_armature[DW_NUM_DOF * e + j] = h->nominal_armature[j] * sa;
        }
        if (cfg->randomize_friction_on_reset) {
            float uf = noise_word(nz, DW_NZ_DR_FRIC);
            b->friction_scale[e] = cfg->dr_friction_scale[0] + uf * (cfg->dr_friction_scale[1] - cfg->dr_friction_scale[0]);
        }
        b->randomize_buf[e] = 0;
    }
    /* terrain curriculum (tasks/dyros_dynamic_walk.py:603-604,671-691): uses the position the robot reached and the
     * target velocity of the episode that just ended */
    if (cfg->terrain_curriculum) {
        const float *root_old = b->root_states + 13 * e;
        float d[2] = {root_old[0] - b->env_origins[3 * e], root_old[1] - b->env_origins[3 * e + 1]};
        const float distance = norm_s(d, 2);
        const int move_up = distance > (float)(cfg->terrain_env_length / 2.0);
        const float need = norm_s(&es[DW_ES_TARGET_VEL], 2) * cfg->max_episode_length_s * 0.5f;
        const int move_down = (distance < need) && !move_up;
        int64_t lvl = b->terrain_levels[e] + (1 * move_up - 1 * move_down);
        if (lvl >= cfg->terrain_num_levels) {
            int k = (int)(noise_word(nz, DW_NZ_TERRAIN_LVL) * (float)cfg->terrain_num_levels);   /* randint_like */
            if (k > cfg->terrain_num_levels - 1) k = cfg->terrain_num_levels - 1;
            lvl = k;
        } else if (lvl < 0) lvl = 0;
        b->terrain_levels[e] = lvl;
        const float *org = b->terrain_origins + ((size_t)lvl * cfg->terrain_num_types + b->terrain_types[e]) * 3;
        for (int i = 0; i < 3; ++i) b->env_origins[3 * e + i] = org[i];
    }
    for (int j = 0; j < DW_NUM_DOF; ++j) {
        es[DW_ES_QPOS_NOISE + j] = h->initial_dof_pos[j];
        es[DW_ES_QPOS_PRE + j] = h->initial_dof_pos[j];
        es[DW_ES_QVEL_NOISE + j] = 0.0f;
    }
    for (int i = 0; i < 12; ++i)
        es[DW_ES_QPOS_BIAS + i] = divs(noise_word(nz, DW_NZ_QPOS_BIAS + i) * 6.28f, 100.0) - (float)(3.14 / 100);
    for (int i = 0; i < 3; ++i)
        es[DW_ES_QUAT_BIAS + i] = divs(noise_word(nz, DW_NZ_QUAT_BIAS + i) * 6.28f, 150.0) - (float)(3.14 / 150);
    /* root and dof state */
    float *root = b->root_states + 13 * e;
    const float init_root[13] = {0, 0, cfg->initial_height, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 13; ++i) root[i] = init_root[i];
    for (int i = 0; i < 3; ++i) root[i] += b->env_origins[3 * e + i];
    if (cfg->custom_origins)      /* xy position within 1 m of the tile centre (:729-732; torch_rand_float = 2 u + (-1)) */
        for (int i = 0; i < 2; ++i) root[i] += 2.0f * noise_word(nz, DW_NZ_ROOT_JITTER + i) + (-1.0f);
    for (int j = 0; j < DW_NUM_DOF; ++j) {
        float q = fmaxf(fminf(h->initial_dof_pos[j], h->model.dof_upper[j]), h->model.dof_lower[j]);
        b->dof_state[(DW_NUM_DOF * e + j) * 2] = q;
        b->dof_state[(DW_NUM_DOF * e + j) * 2 + 1] = 0.0f;
    }
    /* target velocity, starting foot */
    float vel_mag = noise_word(nz, DW_NZ_TARGET_VEL) * 0.8f;
    es[DW_ES_TARGET_VEL] = vel_mag * 1.0f;       /* cos(rand*0.0) = 1 */
    es[DW_ES_TARGET_VEL + 1] = vel_mag * 0.0f;   /* sin(rand*0.0) = 0 */
    *esi(es, DW_ES_INIT_MOCAP) = noise_word(nz, DW_NZ_INIT_MOCAP) > 0.5f ? 0 : 1800;
    for (int j = 0; j < DW_NUM_DOF; ++j) es[DW_ES_PRE_QVEL + j] = 0.0f;
    for (int i = 0; i < 12; ++i) es[DW_ES_ACTION_TORQUE_PRE + i] = 0.0f;
    for (int i = 0; i < 3; ++i) {
        es[DW_ES_FOOT_FORCE_PRE + i] = b->contact_forces[(DW_NUM_BODIES * e + h->model.left_foot_gym) * 3 + i];
        es[DW_ES_FOOT_FORCE_PRE + 3 + i] = b->contact_forces[(DW_NUM_BODIES * e + h->model.right_foot_gym) * 3 + i];
    }
    es[DW_ES_TIME] = 0.0f;
    for (int i = 0; i < 12; ++i) es[DW_ES_MOTOR_SCALE + i] = noise_word(nz, DW_NZ_MOTOR + i) * 0.4f + 0.8f;
    b->progress_buf[e] = 0;
    b->reset_buf[e] = 1;
    for (int i = 0; i < DW_ALOG_SLOTS * 12; ++i) es[DW_ES_ACTION_LOG + i] = 0.0f;
    {
        int k = (int)(noise_word(nz, DW_NZ_DELAY) * 4.0f);      /* torch.randint(2, 6) */
        if (k > 3) k = 3;
        *esi(es, DW_ES_DELAY_IDX) = 2 + k;
    }
    es[DW_ES_CRM] = es[DW_ES_CRS] / es[DW_ES_EPI_LEN];
    es[DW_ES_CRS] = 0.0f;
    *esi(es, DW_ES_SIMUL_LEN) = 0;
    es[DW_ES_EPI_LEN_LOG] = es[DW_ES_EPI_LEN];
    es[DW_ES_EPI_LEN] = 0.0f;
    *esi(es, DW_ES_PERT_COUNT) = 0;
    *esi(es, DW_ES_PERT_ON) = 0;
    {
        int k = (int)(noise_word(nz, DW_NZ_PTIMING) * 2000.0f); /* torch.randint(0, 2000) */
        if (k > 1999) k = 1999;
        *esi(es, DW_ES_PERT_TIMING) = k;
    }
    memset(h->buf.obs_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_OBS1, 0, sizeof(float) * DW_HIST_SLOTS * DW_NUM_OBS1);
    memset(h->buf.action_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_ACT, 0, sizeof(float) * DW_HIST_SLOTS * DW_NUM_ACT);
    for (int i = 0; i < 24; ++i) es[DW_ES_WARM + i] = 0.0f;
}

/* compute_humanoid_walk_observations for one env (tasks/dyros_dynamic_walk.py:750-796) */
static void observe_env(DwHandle *h, int e, const Noise *nz) {
    const DwBuffers *b = &h->buf;
    float *es = ES(h, e);
    const float *root = b->root_states + 13 * e;
    const float x = root[3], y = root[4], z = root[5], w = root[6];
    /* quat2mat (torch_utils.py:227-244) */
    float m00 = w * w + x * x - y * y - z * z;
    float m01 = 2 * x * y - 2 * w * z;
    float m10 = 2 * x * y + 2 * w * z;
    float m11 = w * w - x * x + y * y - z * z;
    float m20 = 2 * x * z - 2 * w * y;
    float m21 = 2 * y * z + 2 * w * x;
    float m22 = w * w - x * x - y * y + z * z;
    /* mat2euler (torch_utils.py:248-269) */
    float cy = sqrtf(m00 * m00 + m10 * m10);
    int cond = cy > (float)(2.220446049250313e-16 * 4);
    float ez = cond ? atan2f(m10, m00) : atan2f(-m01, m11);
    float ey = atan2f(-m20, cy);
    float ex = cond ? atan2f(m21, m22) : 0.0f;
    float obs[DW_NUM_OBS1];
    obs[0] = ex + es[DW_ES_QUAT_BIAS + 0];
    obs[1] = ey + es[DW_ES_QUAT_BIAS + 1];
    obs[2] = ez + es[DW_ES_QUAT_BIAS + 2];
    for (int i = 0; i < 12; ++i) obs[3 + i] = es[DW_ES_QPOS_NOISE + i] + es[DW_ES_QPOS_BIAS + i];
    for (int i = 0; i < 12; ++i) obs[15 + i] = es[DW_ES_QVEL_NOISE + i];
    const float period = (float)(3599 * 0.0005);
    float time2idx = divs(remainder_t(es[DW_ES_TIME], period), 0.0005);
    float phase = divs(remainder_t((float)*esi(es, DW_ES_INIT_MOCAP) + time2idx, 3599.0f), 3599.0);
    float ang = (float)(2 * 3.14159265358979) * phase;
    obs[27] = sinf(ang);
    obs[28] = cosf(ang);
    obs[29] = es[DW_ES_TARGET_VEL];
    obs[30] = es[DW_ES_TARGET_VEL + 1];
    for (int i = 0; i < 6; ++i) obs[31 + i] = root[7 + i] + (noise_word(nz, DW_NZ_VEL + i) * 0.05f - 0.025f);
    float normed[DW_NUM_OBS1];
    for (int i = 0; i < DW_NUM_OBS1; ++i) {
        float diff = obs[i] - h->obs_mean[i];
        normed[i] = diff / sqrtf(h->obs_var[i] + 1e-8f * 1.0f);
    }
    hist_push_obs(h, e, es, normed, es[DW_ES_EPI_LEN] == 0.0f);
    /* obs_buf: logical slot j lives at ring position (head + j) % 20, head = position of the oldest */
    const int head = *esi(es, DW_ES_HIST_HEAD);
    const float *hist = b->obs_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_OBS1;
    const float *ahist = b->action_history + (size_t)e * DW_HIST_SLOTS * DW_NUM_ACT;
    float *ob = b->obs_buf + (size_t)e * DW_NUM_OBS;
    for (int i = 0; i < DW_NUM_HIS; ++i) {
        int slot = (head + DW_NUM_SKIP * (i + 1) - 1) % DW_HIST_SLOTS;
        memcpy(ob + DW_NUM_OBS1 * i, hist + slot * DW_NUM_OBS1, sizeof(float) * DW_NUM_OBS1);
    }
    for (int i = 0; i < DW_NUM_HIS - 1; ++i) {
        int slot = (head + DW_NUM_SKIP * (i + 1)) % DW_HIST_SLOTS;
        memcpy(ob + DW_NUM_OBS1 * DW_NUM_HIS + DW_NUM_ACT * i, ahist + slot * DW_NUM_ACT, sizeof(float) * DW_NUM_ACT);
    }
}

/* compute_humanoid_walk_reward + check_termination for one env */
static void reward_env(DwHandle *h, int e, int *reset_out) {
    const DwConfig *cfg = &h->cfg;
    const DwBuffers *b = &h->buf;
    float *es = ES(h, e);
    const float *root = b->root_states + 13 * e;
    const float *cf = b->contact_forces + (size_t)DW_NUM_BODIES * 3 * e;
    const int LF = h->model.left_foot_gym, RF = h->model.right_foot_gym;
    float qerr = quat_err(root + 3, h->cfg.torch_gpu_div);
    float aerr = fabsf(qerr);
    int collision = 0;
    for (int k = 0; k < DW_NUM_BODIES; ++k) {
        if (k == LF || k == RF) continue;
        if (norm_s(cf + 3 * k, 3) > 1.0f) collision = 1;
    }
    float r[14];
    r[0] = 0.3f * expf(-13.2f * aerr);
    float d33[33];
    for (int j = 0; j < 33; ++j) d33[j] = es[DW_ES_TARGET_QPOS + j] - b->dof_state[(DW_NUM_DOF * e + j) * 2];
    float n = norm_s(d33, 33);
    r[1] = 0.35f * expf(-2.0f * (n * n));
    for (int j = 0; j < 33; ++j) d33[j] = 0.0f - b->dof_state[(DW_NUM_DOF * e + j) * 2 + 1];
    n = norm_s(d33, 33);
    r[2] = 0.05f * expf(-0.01f * (n * n));
    const float *lf = cf + 3 * LF, *rf = cf + 3 * RF;
    const float *lfp = es + DW_ES_FOOT_FORCE_PRE, *rfp = es + DW_ES_FOOT_FORCE_PRE + 3;
    float dl[3], dr[3];
    for (int i = 0; i < 3; ++i) { dl[i] = lf[i] - lfp[i]; dr[i] = rf[i] - rfp[i]; }
    r[9] = 0.2f * expf((-0.01f * 1.0f) * (norm_s(dl, 3) + norm_s(dr, 3)));
    float a12[12];
    for (int i = 0; i < 12; ++i) a12[i] = es[DW_ES_ACTIONS + i] * 333.0f;
    r[4] = 0.05f * expf(-0.01f * norm_s(a12, 12));
    for (int i = 0; i < 12; ++i) a12[i] = (es[DW_ES_ACTIONS + i] - es[DW_ES_ACTIONS_PRE + i]) * 333.0f;
    r[5] = 0.6f * expf((-0.01f * 1.0f) * norm_s(a12, 12));
    for (int j = 0; j < 33; ++j) d33[j] = b->dof_state[(DW_NUM_DOF * e + j) * 2 + 1] - es[DW_ES_PRE_QVEL + j];
    n = norm_s(d33, 33);
    r[7] = 0.05f * expf(-20.0f * (n * n));
    float dv[2] = {es[DW_ES_TARGET_VEL] - root[7], es[DW_ES_TARGET_VEL + 1] - root[8]};
    n = norm_s(dv, 2);
    r[6] = 0.3f * expf(-3.0f * (n * n));
    int lcon = lf[2] > 1.0f, rcon = rf[2] > 1.0f;
    int idx = *esi(es, DW_ES_MOCAP_IDX);
    int DSP = (3300 <= idx && idx < 3600) || (idx < 300) || (1500 <= idx && idx < 2100);
    int RSSP = 300 <= idx && idx < 1500;
    int LSSP = 2100 <= idx && idx < 3300;
    float fcr = 0.0f;
    if (DSP && rcon && lcon) fcr = 0.2f;
    if (RSSP && rcon && !lcon) fcr = 0.2f;
    if (LSSP && !rcon && lcon) fcr = 0.2f;
    r[8] = fcr;
    es[DW_ES_CRS] = es[DW_ES_CRS] + fcr;
    r[10] = 0.0f;
    const float tm = b->total_mass[e];
    const float thr = (float)(1.4 * 9.81) * tm;
    int lth = lf[2] > thr, rth = rf[2] > thr;
    int th = lth || rth;
    r[11] = th ? -0.2f * 1.0f : 0.0f;
    {
        float cl = fmaxf(lf[2] - thr, 0.0f), cr = fmaxf(rf[2] - thr, 0.0f);
        float pen = 0.1f * expf(-0.007f * (norm_s(&cl, 1) + norm_s(&cr, 1)));
        r[3] = th ? pen : 0.1f * 1.0f;
    }
    {
        const float thd = ((float)(0.2 * 9.81) * tm) / 1.0f;
        int ld = fabsf(lf[2] - lfp[2]) > thd, rd = fabsf(rf[2] - rfp[2]) > thd;
        r[12] = (ld || rd) ? -0.05f * 1.0f : 0.0f;
    }
    {
        float ws = divs(tm, 104.48);
        float tl = 0.1f * expf(-0.001f * fabsf(lf[2] + ws * es[DW_ES_TARGET_FORCE]));
        float tr = 0.1f * expf(-0.001f * fabsf(rf[2] + ws * es[DW_ES_TARGET_FORCE + 1]));
        r[13] = tl + tr;
    }
    float total = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + r[8] + r[9] + r[10] + r[11] + r[12] + r[13];
    if (collision) total = 1.0f * cfg->death_cost;
    if (aerr > 0.5f) total = 1.0f * cfg->death_cost;
    float *sr = b->stacked_rewards + (size_t)DW_NUM_REW * e;
    for (int i = 0; i < 14; ++i) sr[i] = collision ? 1.0f * cfg->death_cost : r[i];
    sr[14] = *esi(es, DW_ES_PERT_START) ? 1.0f : 0.0f;
    b->rew_buf[e] = total;
    /* check_termination (uses progress_buf after its increment) */
    int reset = aerr > 0.5f ? 1 : 0;
    if ((float)b->progress_buf[e] >= cfg->max_episode_length - 1.0f) reset = 1;
    if (collision) reset = 1;
    *reset_out = reset;
}

/* gate_acc layout: [slot 0..2][bucket = env % 32][2], latch at DW_GATE_LATCH (include/dyros_walk.h) */
/* (each thread of dwo_step sums into its own [bucket][2] block, merged once per step: 256 threads adding atomically into 32
 *  shared buckets of four per cache line was what kept the all-core baseline at 7x one core) */
static void gate_contribution(int64_t *loc, int e, float el, float cm) {
    const int bk = e % DW_GATE_BUCKETS;
    int64_t de, dc = 0;
    if (isfinite(el) && isfinite(cm)) { de = (int64_t)el; dc = (int64_t)llrintf(cm * 4294967296.0f); }
    else de = -((int64_t)1 << 62);
    loc[bk * 2] += de;
    loc[bk * 2 + 1] += dc;
}

static void step_env(DwHandle *h, int e, const float *actions, const float *noise, int64_t step, int gate_open, int64_t *gate_loc) {
    const DwConfig *cfg = &h->cfg;
    const DwBuffers *b = &h->buf;
    float *es = ES(h, e);
    Noise nz = {noise ? noise + (size_t)DW_NOISE_WORDS * e : NULL, cfg->seed, (uint32_t)e, (uint64_t)step, 0};
    const float period = (float)(3599 * 0.0005), cdt = 0.0005f;
    const double dt_policy_d = cfg->dt * cfg->control_freq_inv;     /* python: self.dt * self.skipframe */

    /* ---------------- pre_physics_step ---------------- */
    float time = es[DW_ES_TIME];
    int init_idx = *esi(es, DW_ES_INIT_MOCAP);
    float local_time = remainder_t(time, period);
    float lt_plus = remainder_t(local_time + (float)init_idx * cdt, period);
    int mocap_idx = (int)(((int64_t)init_idx + (int64_t)divs(local_time, 0.0005)) % 3599);
    int next_idx = mocap_idx + 1;
    *esi(es, DW_ES_MOCAP_IDX) = mocap_idx;
    const float *row0 = h->mocap + (size_t)mocap_idx * DW_MOCAP_COLS, *row1 = h->mocap + (size_t)next_idx * DW_MOCAP_COLS;
    for (int j = 0; j < 33; ++j) es[DW_ES_TARGET_QPOS + j] = cubic_t(lt_plus, row0[0], row1[0], row0[1 + j], row1[1 + j]);
    for (int k = 0; k < 2; ++k) es[DW_ES_TARGET_FORCE + k] = cubic_t(lt_plus, row0[0], row1[0], row0[34 + k], row1[34 + k]);

    float act[DW_NUM_ACT];
    for (int i = 0; i < DW_NUM_ACT; ++i) act[i] = fminf(fmaxf(actions[DW_NUM_ACT * e + i], -1.0f), 1.0f);
    act[12] = (act[12] > 0 ? 1.0f : 0.0f) * act[12];
    for (int i = 0; i < DW_NUM_ACT; ++i) es[DW_ES_ACTIONS + i] = act[i];
    {   /* action history ring: same head as the obs history (both advance once per step, obs later) */
        int pos = *esi(es, DW_ES_HIST_HEAD);
        memcpy(b->action_history + ((size_t)e * DW_HIST_SLOTS + pos) * DW_NUM_ACT, act, sizeof(float) * DW_NUM_ACT);
    }
    for (int i = 0; i < 12; ++i)
        es[DW_ES_ACTION_TORQUE + i] = act[i] * es[DW_ES_MOTOR_SCALE + i] * h->action_high[i];

    /* push perturbation (tasks/dyros_dynamic_walk.py:438-447,489-502) */
    float push[2] = {0, 0};
    if (gate_open) *esi(es, DW_ES_PERT_START) = 1;
    if (*esi(es, DW_ES_PERT_START)) {
        if (remainder_t(es[DW_ES_EPI_LEN], (float)(8 / dt_policy_d)) == (float)*esi(es, DW_ES_PERT_TIMING)) {
            *esi(es, DW_ES_PERT_ON) = 1;
            int imp = 50 + (int)(noise_word(&nz, DW_NZ_PERT + 0) * 200.0f);
            if (imp > 249) imp = 249;
            const int dlo = (int)(0.1 / dt_policy_d), dhi = (int)(1 / dt_policy_d);   /* :441 */
            int dur = dlo + (int)(noise_word(&nz, DW_NZ_PERT + 1) * (float)(dhi - dlo));
            if (dur > dhi - 1) dur = dhi - 1;
            *esi(es, DW_ES_IMPULSE) = imp;
            *esi(es, DW_ES_PERT_DURATION) = dur;
            es[DW_ES_MAGNITUDE] = (float)imp / ((float)dur * (float)dt_policy_d);
            es[DW_ES_PHASE] = noise_word(&nz, DW_NZ_PERT + 2) * 2.0f * (float)3.14159265358979;
        }
        if (*esi(es, DW_ES_PERT_ON)) {
            *esi(es, DW_ES_PERT_COUNT) += 1;
            push[0] = es[DW_ES_MAGNITUDE] * cosf(es[DW_ES_PHASE]);
            push[1] = es[DW_ES_MAGNITUDE] * sinf(es[DW_ES_PHASE]);
        }
        if (*esi(es, DW_ES_PERT_COUNT) == *esi(es, DW_ES_PERT_DURATION)) {
            *esi(es, DW_ES_PERT_ON) = 0;
            *esi(es, DW_ES_PERT_COUNT) = 0;
        }
    }

    DwoPhysIO io;
    dwo_load_phys(h, e, &io);
    for (int i = 0; i < 24; ++i) io.warm[i] = es[DW_ES_WARM + i];
    for (int sub = 0; sub < 2; ++sub) {
        /* upper-body PD on the true joint state (tasks/dyros_dynamic_walk.py:506) */
        for (int j = 12; j < 33; ++j) {
            float q = (float)io.q[j], qd = (float)io.qd[j];
            io.tau[j] = h->kp[j] * (es[DW_ES_TARGET_QPOS + j] - q) + h->kv[j] * (-qd);
        }
        /* torque FIFO and actuator delay (:511-519) */
        float *alog = es + DW_ES_ACTION_LOG;
        memmove(alog, alog + 12, sizeof(float) * 12 * (DW_ALOG_SLOTS - 1));
        memcpy(alog + 12 * (DW_ALOG_SLOTS - 1), es + DW_ES_ACTION_TORQUE, sizeof(float) * 12);
        int sl = *esi(es, DW_ES_SIMUL_LEN) + 1;
        if (sl > DW_ALOG_SLOTS) sl = DW_ALOG_SLOTS;
        *esi(es, DW_ES_SIMUL_LEN) = sl;
        int dl = *esi(es, DW_ES_DELAY_IDX);
        const float *src = sl > dl ? alog + 12 * dl : alog + 12 * (DW_ALOG_SLOTS - sl);
        for (int i = 0; i < 12; ++i) io.tau[i] = src[i];
        io.push[0] = sub == 0 ? push[0] : 0;     /* applied forces last one simulate() (docs/programming/tensors) */
        io.push[1] = sub == 0 ? push[1] : 0;
        if (!cfg->debug_freeze_physics) dwo_phys_substep(cfg, &h->rmodel, &io);
        /* encoder model (:528-530) */
        for (int j = 0; j < 33; ++j) {
            float nzv = noise_word(&nz, DW_NZ_ENC + 33 * sub + j);
            float qn = (float)io.q[j] + fminf(fmaxf(nzv, -0.00016f), 0.00016f);
            es[DW_ES_QVEL_NOISE + j] = divs(qn - es[DW_ES_QPOS_PRE + j], cfg->dt);
            es[DW_ES_QPOS_NOISE + j] = qn;
            es[DW_ES_QPOS_PRE + j] = qn;
        }
    }
    for (int i = 0; i < 24; ++i) es[DW_ES_WARM + i] = (float)io.warm[i];
    if (!cfg->debug_freeze_physics) dwo_store_phys(h, e, &io);
    es[DW_ES_EPI_LEN] += 1.0f;
    time = time + (float)dt_policy_d;
    time = time + (float)(5 * dt_policy_d) * act[12];
    es[DW_ES_TIME] = time;

    /* ---------------- VecTask.step between pre and post (vec_task.py:325) ---------------- */
    {
        int64_t p = b->progress_buf[e] + (cfg->timeout_fix ? 1 : 0);
        b->timeout_buf[e] = ((float)p >= cfg->max_episode_length - 1.0f) ? 1 : 0;
    }
    /* ---------------- post_physics_step ---------------- */
    b->progress_buf[e] += 1;
    b->randomize_buf[e] += 1;
    /* non-finite guard: PhysX clamps silently, we reset and count */
    int bad = 0;
    for (int i = 0; i < 13; ++i) bad |= !isfinite(b->root_states[13 * e + i]);
    for (int j = 0; j < 66; ++j) bad |= !isfinite(b->dof_state[(size_t)66 * e + j]);
    int reset = 0;
    if (bad) {
        /* make the state finite so reward/obs stay finite; the env is reset right below */
        float *root = b->root_states + 13 * e;
        for (int i = 0; i < 13; ++i) root[i] = 0;
        root[2] = cfg->initial_height; root[6] = 1;
        for (int j = 0; j < 66; ++j) b->dof_state[(size_t)66 * e + j] = 0;
        for (int k = 0; k < DW_NUM_BODIES * 3; ++k) b->contact_forces[(size_t)DW_NUM_BODIES * 3 * e + k] = 0;
        *esi(es, DW_ES_NAN_RESETS) += 1;
    }
    reward_env(h, e, &reset);
    if (bad) reset = 1;
    b->reset_buf[e] = reset;
    es[DW_ES_EPI_RETURN] = es[DW_ES_EPI_RETURN] + b->rew_buf[e];
    if (reset) {
        es[DW_ES_LAST_RETURN] = es[DW_ES_EPI_RETURN];
        es[DW_ES_EPI_RETURN] = 0.0f;
        *esi(es, DW_ES_EPISODES) += 1;
        nz.stream = 0;
        reset_env(h, e, &nz);
    }
    observe_env(h, e, &nz);
    /* late update (:560-563) */
    for (int j = 0; j < 33; ++j) es[DW_ES_PRE_QVEL + j] = b->dof_state[(DW_NUM_DOF * e + j) * 2 + 1];
    for (int i = 0; i < 12; ++i) es[DW_ES_ACTION_TORQUE_PRE + i] = es[DW_ES_ACTION_TORQUE + i];
    for (int i = 0; i < 3; ++i) {
        es[DW_ES_FOOT_FORCE_PRE + i] = b->contact_forces[(DW_NUM_BODIES * e + h->model.left_foot_gym) * 3 + i];
        es[DW_ES_FOOT_FORCE_PRE + 3 + i] = b->contact_forces[(DW_NUM_BODIES * e + h->model.right_foot_gym) * 3 + i];
    }
    for (int i = 0; i < DW_NUM_ACT; ++i) es[DW_ES_ACTIONS_PRE + i] = es[DW_ES_ACTIONS + i];
    /* statistics for the next step's perturbation gate */
    if (cfg->perturb && !cfg->force_perturb_start) gate_contribution(gate_loc, e, es[DW_ES_EPI_LEN_LOG], es[DW_ES_CRM]);
}

/* perturbation gate (tasks/dyros_dynamic_walk.py:489): population means of the previous step */
static int gate_is_open(const DwHandle *h, int64_t step) {
    const DwConfig *cfg = &h->cfg;
    if (cfg->force_perturb_start) return 1;
    if (!cfg->perturb) return 0;
    const int64_t *acc = h->buf.gate_acc;
    if (acc[DW_GATE_LATCH]) return 1;
    int prev = (int)((step + 2) % 3);
    int64_t se = 0, sc = 0;
    for (int k = 0; k < DW_GATE_BUCKETS; ++k) {
        se += acc[(prev * DW_GATE_BUCKETS + k) * 2];
        sc += acc[(prev * DW_GATE_BUCKETS + k) * 2 + 1];
    }
    double n = (double)cfg->num_envs;
    double mean_epi = (double)se / n;
    double mean_crm = (double)sc / 4294967296.0 / n;
    return mean_epi > (double)(cfg->max_episode_length - 2000.0f) && mean_crm > 0.165;
}

int dwo_step(DwHandle *h, const float *actions, const float *noise, int64_t step_index, void *stream) {
    (void)stream;
    if (!h || !h->bound || !h->has_task) return dwo_fail(DW_ESTATE, "dwo_step: handle has no task constants or no buffers bound");
    {
        const DwBuffers *b = &h->buf;
        if (!b->env_state || !b->obs_buf || !b->rew_buf || !b->reset_buf || !b->progress_buf || !b->timeout_buf ||
            !b->randomize_buf || !b->stacked_rewards || !b->obs_history || !b->action_history || !b->gate_acc ||
            !b->total_mass || !b->env_origins)
            return dwo_fail(DW_ESTATE, "dwo_step: task buffers missing");
    }
    if (!actions) return dwo_fail(DW_EINVAL, "dwo_step: actions is null");
    const int N = h->cfg.num_envs;
    int open = gate_is_open(h, step_index);
    if (open && !h->cfg.force_perturb_start) h->buf.gate_acc[DW_GATE_LATCH] = 1;
    int64_t *acc = h->buf.gate_acc;
    const int cur = (int)(step_index % 3), nxt = (int)((step_index + 1) % 3);
#pragma omp parallel
    {
        int64_t loc[DW_GATE_BUCKETS * 2];
        for (int k = 0; k < DW_GATE_BUCKETS * 2; ++k) loc[k] = 0;
#pragma omp for schedule(static) nowait
        for (int e = 0; e < N; ++e) step_env(h, e, actions, noise, step_index, open, loc);
        if (h->cfg.perturb && !h->cfg.force_perturb_start) {
#pragma omp critical
            for (int k = 0; k < DW_GATE_BUCKETS * 2; ++k) acc[cur * DW_GATE_BUCKETS * 2 + k] += loc[k];
        }
    }
    if (h->cfg.perturb && !h->cfg.force_perturb_start)
        for (int k = 0; k < DW_GATE_BUCKETS * 2 && k < 2 * N; ++k) acc[nxt * DW_GATE_BUCKETS * 2 + k] = 0;
    return DW_OK;
}

/* the counter of dw_step_dev is a host word here: read, step, add one */
/* dw_step_obs: the step with its observations written to obs_out instead of the bound obs_buf */
int dwo_step_obs(DwHandle *h, const float *actions, const float *noise, int64_t step_index, int64_t *step_counter, float *obs_out, void *stream) {
    if (!h || !obs_out) return dwo_fail(DW_EINVAL, "dwo_step_obs: null argument");
    float *bound = h->buf.obs_buf;
    h->buf.obs_buf = obs_out;
    const int rc = dwo_step(h, actions, noise, step_counter ? *step_counter : step_index, stream);
    h->buf.obs_buf = bound;
    if (rc == DW_OK && step_counter) *step_counter += 1;
    return rc;
}

int dwo_step_dev(DwHandle *h, const float *actions, const float *noise, int64_t *step_counter, void *stream) {
    if (!step_counter) return dwo_fail(DW_EINVAL, "dwo_step_dev: step_counter is null");
    const int rc = dwo_step(h, actions, noise, *step_counter, stream);
    if (rc == DW_OK) *step_counter += 1;
    return rc;
}

int dwo_reset_idx(DwHandle *h, const int32_t *env_ids, int32_t n, const float *noise, int64_t step_index, void *stream) {
    (void)stream;
    if (!h || !h->bound || !h->has_task || !h->buf.env_state) return dwo_fail(DW_ESTATE, "dwo_reset_idx: not ready");
    if (n < 0 || (n > 0 && !env_ids)) return dwo_fail(DW_EINVAL, "dwo_reset_idx: bad env id list");
    for (int i = 0; i < n; ++i) {
        int e = env_ids[i];
        if (e < 0 || e >= h->cfg.num_envs) return dwo_fail(DW_EINVAL, "dwo_reset_idx: env id out of range");
        Noise nz = {noise ? noise + (size_t)DW_NOISE_WORDS * e : NULL, h->cfg.seed, (uint32_t)e, (uint64_t)step_index, 1};
        reset_env(h, e, &nz);
    }
    return DW_OK;
}
