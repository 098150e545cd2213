/* dw_amp.c -- CPU restatement of the env-side arithmetic of the sibling TOCABI tasks.
 *
 * TEST INFRASTRUCTURE ONLY: the checker of the HIP entry points dw_amp_* (include/dyros_walk.h).  Pinned against the
 * reference's own TorchScript functions, imported from where they lie and run on seeded inputs by
 * oracle/make_amp_goldens.py (fixture tests/golden/amp_lower_ref.npz, test tests/test_amp_oracle.py):
 *
 *   dwo_amp_observations   tasks/amp/tocabi_amp_lower_base.py:918-962  compute_humanoid_observations
 *                          (quat2euler: python/isaacgym/torch_utils.py:227-273, quat_rotate_inverse: :72-81)
 *   dwo_amp_disc_observations  tasks/tocabi_amp_lower.py:310-350 build_amp_observations (the AMP subclass' discriminator
 *                          observation; my_quat_rotate / calc_heading_quat_inv: utils/torch_jit_utils.py:199-209,333-368;
 *                          quat_from_angle_axis / normalize: python/isaacgym/torch_utils.py:44-46,92-102)
 *   dwo_amp_reward         tasks/amp/tocabi_amp_lower_base.py:964-1023 compute_humanoid_reward
 *   dwo_amp_reset          tasks/amp/tocabi_amp_lower_base.py:1025-1069 compute_humanoid_reset
 *                          (quat_diff_rad: utils/torch_jit_utils.py:141-160)
 *   dwo_newwalk_reward     tasks/tocabi_new_walk.py:384-496 compute_humanoid_walk_reward (the one function of TocabiNewWalk
 *                          that runs: the class itself fails at :558-567, a [N,30] - [37] broadcast; pinned as a test)
 *
 * fp32 throughout, operations in the order of torch's CPU kernels (norm: 8-lane fused accumulation, see dw_task.c norm_t).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "dw_amp.h"
#include "dw_oracle.h"

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

/* quat_rotate_inverse(q, v), python/isaacgym/torch_utils.py:72-81: a - b + c with
 * a = v (2 w^2 - 1), b = cross(q_vec, v) w 2, c = q_vec (q_vec . v) 2 */
static void quat_rotate_inverse(const float *q /* xyzw */, const float *v, float *o) {
    const float w = q[3];
    const float s = 2.0f * (w * w) - 1.0f;
    /* torch.cross on the CPU contracts the first product into the subtraction: fma(a_i, b_j, -(a_j b_i)) (found by matching the reference's bits) */
    const float cr[3] = {fmaf(q[1], v[2], -(q[2] * v[1])), fmaf(q[2], v[0], -(q[0] * v[2])), fmaf(q[0], v[1], -(q[1] * v[0]))};
    const float dot = (q[0] * v[0] + q[1] * v[1]) + q[2] * v[2];        /* bmm [1,3] x [3,1] */
    for (int i = 0; i < 3; ++i) {
        const float a = v[i] * s;
        const float b = cr[i] * w * 2.0f;
        const float c = q[i] * dot * 2.0f;
        o[i] = a - b + c;
    }
}

/* quat2euler(q) = mat2euler(quat2mat(q)), python/isaacgym/torch_utils.py:227-273 (as dw_task.c observe_env) */
static void quat2euler(const float *q, float *e) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float m00 = w * w + x * x - y * y - z * z;
    const float m01 = 2 * x * y - 2 * w * z;
    const float m10 = 2 * x * y + 2 * w * z;
    const float m11 = w * w - x * x + y * y - z * z;
    const float m20 = 2 * x * z - 2 * w * y;
    const float m21 = 2 * y * z + 2 * w * x;
    const float m22 = w * w - x * x - y * y + z * z;
    const float cy = sqrtf(m00 * m00 + m10 * m10);
    const int cond = cy > (float)(2.220446049250313e-16 * 4);
    e[2] = cond ? atan2f(m10, m00) : atan2f(-m01, m11);
    e[1] = atan2f(-m20, cy);
    e[0] = cond ? atan2f(m21, m22) : 0.0f;
}

/* |quat_diff_rad(identity, q)| (utils/torch_jit_utils.py:141-160; as dw_task.c quat_err) */
static float quat_err(const float *q) {
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
    return fabsf(2.0f * asinf(n));
}

int dwo_amp_observations(int n, const float *root_states, const float *rootvel_noise, const float *dof_pos,
                          const float *dof_pos_bias, const float *quat_bias, const float *dof_vel, const float *commands,
                          float *obs, void *stream) {
    (void)stream;
    for (int e = 0; e < n; ++e) {
        const float *r = root_states + 13 * e, *nz = rootvel_noise + 6 * e;
        float *o = obs + DW_AMP_NUM_OBS1 * e;
        float eu[3], vel[3], lv[3];
        quat2euler(r + 3, eu);
        for (int i = 0; i < 3; ++i) o[i] = eu[i] + quat_bias[3 * e + i];
        for (int i = 0; i < 3; ++i) vel[i] = r[7 + i] + nz[i];
        quat_rotate_inverse(r + 3, vel, lv);
        for (int i = 0; i < 3; ++i) o[3 + i] = lv[i];
        for (int i = 0; i < 3; ++i) o[6 + i] = r[10 + i] + nz[3 + i];
        for (int i = 0; i < 3; ++i) o[9 + i] = commands[3 * e + i];
        for (int i = 0; i < 12; ++i) o[12 + i] = dof_pos[33 * e + i] + dof_pos_bias[12 * e + i];
        for (int i = 0; i < 12; ++i) o[24 + i] = dof_vel[33 * e + i];
    }
    return DW_OK;
}

/* my_quat_rotate(q, v) = a + b + c, utils/torch_jit_utils.py:199-209 */
static void my_quat_rotate(const float *q, const float *v, float *o) {
    const float w = q[3];
    const float s = 2.0f * (w * w) - 1.0f;
    const float cr[3] = {fmaf(q[1], v[2], -(q[2] * v[1])), fmaf(q[2], v[0], -(q[0] * v[2])), fmaf(q[0], v[1], -(q[1] * v[0]))};
    const float dot = (q[0] * v[0] + q[1] * v[1]) + q[2] * v[2];
    for (int i = 0; i < 3; ++i) {
        const float a = v[i] * s;
        const float b = cr[i] * w * 2.0f;
        const float c = q[i] * dot * 2.0f;
        o[i] = a + b + c;
    }
}

/* calc_heading_quat_inv(q), utils/torch_jit_utils.py:333-368: heading = atan2 of the rotated x axis, then
 * quat_unit(quat_from_angle_axis(-heading, z)) */
static void heading_quat_inv(const float *q, float *hq) {
    const float ref[3] = {1.0f, 0.0f, 0.0f};
    float rd[3];
    my_quat_rotate(q, ref, rd);
    const float heading = atan2f(rd[1], rd[0]);
    const float theta = (-heading) / 2.0f;
    const volatile float zero = 0.0f;                       /* (0 * sin keeps the product's sign) */
    const float sn = sinf(theta), cs = cosf(theta);
    float u[4] = {zero * sn, zero * sn, 1.0f * sn, cs};
    float nn = norm_t(u, 4);
    nn = nn < 1e-9f ? 1e-9f : nn;
    for (int i = 0; i < 4; ++i) hq[i] = u[i] / nn;
}

int dwo_amp_disc_observations(int n, const float *root_states, const float *dof_pos, const float *dof_vel, int dof_row_stride,
                               int dof_elem_stride, int local_root_obs, const float *key_pos, int n_key, float *obs, void *stream) {
    (void)stream;
    if (n <= 0 || n_key < 1 || n_key > DW_MAX_BODY_QUERY || dof_elem_stride < 1) return DW_EINVAL;
    const int W = DW_AMP_DISC_BASE + 3 * n_key;
    for (int e = 0; e < n; ++e) {
        const float *r = root_states + 13 * (size_t)e;
        float *o = obs + (size_t)W * e;
        float hq[4], eu[3];
        heading_quat_inv(r + 3, hq);
        quat2euler(r + 3, eu);
        o[0] = r[2];
        for (int i = 0; i < 3; ++i) o[1 + i] = eu[i];
        for (int i = 0; i < 12; ++i) o[4 + i] = dof_pos[(size_t)dof_row_stride * e + (size_t)dof_elem_stride * i];
        for (int i = 0; i < 12; ++i) o[16 + i] = dof_vel[(size_t)dof_row_stride * e + (size_t)dof_elem_stride * i];
        for (int k = 0; k < n_key; ++k) {
            const float *kp = key_pos + ((size_t)n_key * e + k) * 3;
            if (local_root_obs) {
                for (int i = 0; i < 3; ++i) o[DW_AMP_DISC_BASE + 3 * k + i] = kp[i];
            } else {
                const float lp[3] = {kp[0] - r[0], kp[1] - r[1], kp[2] - r[2]};
                float le[3];
                my_quat_rotate(hq, lp, le);
                for (int i = 0; i < 3; ++i) o[DW_AMP_DISC_BASE + 3 * k + i] = le[i];
            }
        }
    }
    return DW_OK;
}

int dwo_amp_reward(int n, const float *root_states, const float *dof_vel, const float *dof_vel_pre, const float *commands,
                    const float *actions, const float *actions_pre, const float *motor_efforts, const float *contact_force,
                    const float *total_mass, float *reward, float *reward_values, void *stream) {
    (void)stream;
    for (int e = 0; e < n; ++e) {
        const float *r = root_states + 13 * e, *cmd = commands + 3 * e;
        float lv[3];
        quat_rotate_inverse(r + 3, r + 7, lv);
        float d = cmd[0] - lv[0];
        const float rx = 0.8f * expf(-6.0f * (d * d));
        d = cmd[1] - lv[1];
        const float ry = 0.8f * expf(-6.0f * (d * d));
        d = cmd[2] - r[12];
        const float ryaw = 0.6f * expf(-7.0f * (d * d));
        /* 1.4*9.81*total_mass: the python scalars fold in double, the product with the tensor is float */
        const float thr = (float)(1.4 * 9.81) * total_mass[e];
        const float fl = contact_force[(38 * e + 8) * 3 + 2], fr = contact_force[(38 * e + 16) * 3 + 2];
        const int thres = (fl > thr) || (fr > thr);
        const float r_thr = -0.2f * (thres ? 1.0f : 0.0f);
        float cl = fl - thr, crr = fr - thr;
        cl = cl < 0.0f ? 0.0f : cl;
        crr = crr < 0.0f ? 0.0f : crr;
        const float nl = norm_t(&cl, 1), nr = norm_t(&crr, 1);
        const float pen = 0.1f * (1.0f - expf(-0.007f * (nl + nr)));
        const float r_pen = thres ? pen : 0.1f * 1.0f;
        float dv[33], ta[12], tdiff[12];
        const float nv = norm_t(dof_vel + 33 * e, 33);
        const float r_jv = 0.05f * expf(-0.01f * (nv * nv));
        for (int i = 0; i < 33; ++i) dv[i] = dof_vel[33 * e + i] - dof_vel_pre[33 * e + i];
        const float na = norm_t(dv, 33);
        const float r_ja = 0.05f * expf(-20.0f * (na * na));
        for (int i = 0; i < 12; ++i) ta[i] = actions[12 * e + i] * motor_efforts[i];
        const float r_t = 0.08f * expf(-0.05f * norm_t(ta, 12));
        for (int i = 0; i < 12; ++i) tdiff[i] = (actions[12 * e + i] - actions_pre[12 * e + i]) * motor_efforts[i];
        const float r_td = 0.6f * expf(-0.01f * norm_t(tdiff, 12));
        float rew = 0.0f;
        rew += rx;
        rew += ryaw;
        rew += (r_thr + r_pen);
        rew += r_jv;
        rew += r_ja;
        rew += r_t;
        rew += r_td;
        reward[e] = rew;
        float *rv = reward_values + 9 * e;
        rv[0] = rx; rv[1] = ry; rv[2] = ryaw; rv[3] = r_thr; rv[4] = r_pen; rv[5] = r_jv; rv[6] = r_ja; rv[7] = r_t; rv[8] = r_td;
    }
    return DW_OK;
}

int dwo_amp_reset(int n, const int64_t *progress_buf, const float *contact_buf, const int32_t *contact_body_ids, int n_contact_ids,
                   const float *rigid_body_pos, const float *rigid_body_rot, float max_episode_length, int enable_early_termination,
                   float termination_height, int64_t *reset, int64_t *terminated, void *stream) {
    (void)stream;
    for (int e = 0; e < n; ++e) {
        int64_t term = 0;
        if (enable_early_termination) {
            int fall_contact = 0;
            for (int b = 0; b < 38; ++b) {
                int support = 0;
                for (int k = 0; k < n_contact_ids; ++k) support |= contact_body_ids[k] == b;
                if (support) continue;
                for (int i = 0; i < 3; ++i) fall_contact |= contact_buf[(38 * e + b) * 3 + i] > 1.0f;
            }
            int fall_height = rigid_body_pos[(38 * e + 0) * 3 + 2] < termination_height;
            fall_height |= rigid_body_pos[(38 * e + 8) * 3 + 2] > 0.5f;
            fall_height |= rigid_body_pos[(38 * e + 16) * 3 + 2] > 0.5f;
            int fallen = fall_contact || fall_height;
            fallen |= quat_err(rigid_body_rot + (38 * e + 0) * 4) > (float)(3.141592 / 4.0);
            fallen = fallen && (progress_buf[e] > 1);
            term = fallen ? 1 : 0;
        }
        terminated[e] = term;
        reset[e] = ((float)progress_buf[e] >= max_episode_length - 1.0f) ? 1 : term;
    }
    return DW_OK;
}

/* sync_reward(phase), utils/torch_jit_utils.py:406-418 */
static float nw_sync(float phase) {
    const float a = (float)(1.0 / 12), b = (float)(5.0 / 12);
    float r = 1.0f;
    if (phase < a) r = 1.0f - phase * 24.0f;
    if (a <= phase && phase < b) r = -1.0f;
    if (b <= phase && phase < 0.5f) r = 24.0f * phase - 11.0f;
    return r;
}

int dwo_newwalk_reward(int n, const int64_t *reset_buf, const int64_t *progress_buf, const float *target_vel,
                        const float *root_pose_states, const float *joint_position_states, const float *joint_velocity_states,
                        const int32_t *non_feet_idxs, int n_non_feet, const float *contact_forces, int num_bodies,
                        float termination_height, float death_cost, float max_episode_length, const float *q_nominal, int num_dof,
                        const float *head_states, const float *lfoot_states, const float *rfoot_states, const float *phase_in,
                        float *total_reward, int64_t *reset, float *reward8, void *stream) {
    (void)stream;
    if (num_dof <= 0 || num_dof > 64 || num_bodies < 15) return dwo_fail(DW_EINVAL, "dwo_newwalk_reward: num_dof must be 1..64 and num_bodies >= 15");
    const float pi = (float)3.14159265358979;
    for (int e = 0; e < n; ++e) {
        const float phase = phase_in[e];
        const float *cf = contact_forces + (size_t)num_bodies * 3 * e;
        const float lfn = fabsf(cf[7 * 3 + 2]), rfn = fabsf(cf[14 * 3 + 2]);
        int fly = (lfn + rfn) == 0.0f;
        const int non_init = ((0.04f < phase) && (phase < 0.5f)) || (phase > 0.54f);
        fly = non_init ? fly : 0;
        const float lfv = lfoot_states[13 * e + 7], rfv = rfoot_states[13 * e + 7];
        /* scale_transform(saturate(x, 0, u), 0, u) = 2 (x - offset) / (u - 0), offset = (u + 0) * 0.5 */
        const float ug = (float)(100 * 9.81 * 0.5) * 1.0f, uv = 0.3f * 1.0f;
        float t;
        t = fmaxf(fminf(lfn, ug), 0.0f); const float gl = 2.0f * (t - (ug + 0.0f) * 0.5f) / (ug - 0.0f);
        t = fmaxf(fminf(rfn, ug), 0.0f); const float gr = 2.0f * (t - (ug + 0.0f) * 0.5f) / (ug - 0.0f);
        t = fmaxf(fminf(lfv, uv), 0.0f); const float vl = 2.0f * (t - (uv + 0.0f) * 0.5f) / (uv - 0.0f);
        t = fmaxf(fminf(rfv, uv), 0.0f); const float vr = 2.0f * (t - (uv + 0.0f) * 0.5f) / (uv - 0.0f);
        const float sgl = nw_sync(phase), sgr = nw_sync(phase > 0.5f ? phase - 0.5f : phase + 0.5f);
        const float svl = -sgl, svr = -sgr;
        const float grf = (tanf(pi / 4 * sgl * gl) + tanf(pi / 4 * sgr * gr)) / 2;
        const float spd = (tanf(pi / 4 * svl * vl) + tanf(pi / 4 * svr * vr)) / 2;
        const float *rp = root_pose_states + 13 * e;
        float d2[2] = {rp[7] - target_vel[2 * e], rp[8] - target_vel[2 * e + 1]};
        float nn = norm_t(d2, 2);
        const float rvel = expf(-10 * (nn * nn));
        nn = norm_t(rp + 10, 3);
        const float rang = expf(-10 * (nn * nn));
        const float hd = rp[2] - 1.0f;
        const float rh = expf(-40 * (hd * hd));
        float du[2] = {rp[0] - head_states[13 * e], rp[1] - head_states[13 * e + 1]};
        nn = norm_t(du, 2);
        const float rup = expf(-10 * (nn * nn));
        float dq[64];
        for (int i = 0; i < num_dof; ++i) dq[i] = joint_position_states[(size_t)num_dof * e + i] - q_nominal[i];
        nn = norm_t(dq, num_dof);
        const float rpost = expf(-(nn * nn));
        nn = norm_t(joint_velocity_states + (size_t)num_dof * e, num_dof);
        const float rjv = expf(-5e-6f * (nn * nn));
        float r8[8] = {grf, spd, rvel, rang, rh, rup, rpost, rjv};
        float tot = 0.225f * grf + 0.225f * spd + 0.1f * rvel + 0.1f * rang + 0.05f * rh + 0.1f * rup + 0.1f * rpost + 0.1f * rjv;
        float dl[2] = {lfoot_states[13 * e] - rfoot_states[13 * e], lfoot_states[13 * e + 1] - rfoot_states[13 * e + 1]};
        const float leg_len = norm_t(dl, 2);
        int coll = 0;
        for (int k = 0; k < n_non_feet; ++k) coll |= norm_t(cf + 3 * non_feet_idxs[k], 3) > 1.0f;
        const int low = rp[2] < termination_height, conv = leg_len < 0.1f;
        if (low) tot = 1.0f * death_cost;
        if (conv) tot = 1.0f * death_cost;
        if (coll) tot = 1.0f * death_cost;
        if (fly) tot = 1.0f * death_cost;
        if (low || conv || coll || fly) for (int i = 0; i < 8; ++i) r8[i] = 1.0f * death_cost;
        /* the reference overwrites `reset` from reset_buf on its second line: the height test does not survive */
        int64_t rs = low ? 1 : reset_buf[e];
        rs = conv ? 1 : reset_buf[e];
        rs = ((float)progress_buf[e] >= max_episode_length - 1.0f) ? 1 : rs;
        rs = coll ? 1 : rs;
        rs = fly ? 1 : rs;
        total_reward[e] = tot;
        reset[e] = rs;
        memcpy(reward8 + 8 * e, r8, sizeof r8);
    }
    return DW_OK;
}

/* World position of the origin of moving bodies (the rows of the rigid-body state tensor the functions above read), by
 * composing unit quaternions in double: q_b = q_p * quat(rot0_b) * quat(axis_b, angle_b), x_b = x_p + rotate(q_p, pos_b). */
static void qmul_d(const double *a, const double *b, double *o) {           /* xyzw */
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
static void qrot_d(const double *q, const double *v, double *o) {
    const double t[3] = {2 * (q[1] * v[2] - q[2] * v[1]), 2 * (q[2] * v[0] - q[0] * v[2]), 2 * (q[0] * v[1] - q[1] * v[0])};
    o[0] = v[0] + q[3] * t[0] + (q[1] * t[2] - q[2] * t[1]);
    o[1] = v[1] + q[3] * t[1] + (q[2] * t[0] - q[0] * t[2]);
    o[2] = v[2] + q[3] * t[2] + (q[0] * t[1] - q[1] * t[0]);
}
static void mat2quat_d(const float *R, double *q) {
    const double tr = (double)R[0] + R[4] + R[8];
    if (tr > 0) { double S = sqrt(tr + 1.0) * 2; q[3] = 0.25 * S; q[0] = (R[7] - R[5]) / S; q[1] = (R[2] - R[6]) / S; q[2] = (R[3] - R[1]) / S; }
    else if (R[0] > R[4] && R[0] > R[8]) { double S = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / S; q[0] = 0.25 * S; q[1] = (R[1] + R[3]) / S; q[2] = (R[2] + R[6]) / S; }
    else if (R[4] > R[8]) { double S = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / S; q[0] = (R[1] + R[3]) / S; q[1] = 0.25 * S; q[2] = (R[5] + R[7]) / S; }
    else { double S = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / S; q[0] = (R[2] + R[6]) / S; q[1] = (R[5] + R[7]) / S; q[2] = 0.25 * S; }
}
int dwo_body_positions(DwHandle *h, const int32_t *moving_bodies, int nb, float *out, void *stream) {
    (void)stream;
    if (!h || !moving_bodies || !out) return dwo_fail(DW_EINVAL, "dwo_body_positions: null argument");
    if (nb <= 0 || nb > DW_MAX_BODY_QUERY) return dwo_fail(DW_EINVAL, "dwo_body_positions: nb must be 1..DW_MAX_BODY_QUERY");
    if (!h->bound) return dwo_fail(DW_ESTATE, "dwo_body_positions: dwo_bind first");
    for (int k = 0; k < nb; ++k)
        if (moving_bodies[k] < 0 || moving_bodies[k] >= DW_NUM_MOVING) return dwo_fail(DW_EINVAL, "dwo_body_positions: moving body index out of range");
    const DwModel *m = &h->model;
    for (int e = 0; e < h->cfg.num_envs; ++e) {
        const float *r = h->buf.root_states + 13 * (size_t)e;
        for (int k = 0; k < nb; ++k) {
            int chain[DW_NUM_MOVING], n = 0;
            for (int b = moving_bodies[k]; b > 0; b = m->mv_parent[b]) chain[n++] = b;
            double q[4] = {r[3], r[4], r[5], r[6]}, x[3] = {r[0], r[1], r[2]};
            for (int c = n - 1; c >= 0; --c) {
                const int b = chain[c];
                const double p[3] = {m->mv_pos[b][0], m->mv_pos[b][1], m->mv_pos[b][2]};
                double d[3], q0[4], qa[4], t[4];
                qrot_d(q, p, d);
                x[0] += d[0]; x[1] += d[1]; x[2] += d[2];
                mat2quat_d(m->mv_rot0[b], q0);
                const double ang = h->buf.dof_state[((size_t)DW_NUM_DOF * e + (b - 1)) * 2];
                const double sn = sin(0.5 * ang), cs = cos(0.5 * ang);
                qa[0] = m->mv_axis[b][0] * sn; qa[1] = m->mv_axis[b][1] * sn; qa[2] = m->mv_axis[b][2] * sn; qa[3] = cs;
                qmul_d(q, q0, t);
                qmul_d(t, qa, q);
            }
            float *o = out + ((size_t)nb * e + k) * 3;
            o[0] = (float)x[0]; o[1] = (float)x[1]; o[2] = (float)x[2];
        }
    }
    return DW_OK;
}
