// dw_amp.h -- env-side arithmetic of the sibling TOCABI tasks on the same physics: one thread
// per env, fp32 in the order of the reference's TorchScript functions (fp contraction off), so that the reference goldens
// (tests/golden/amp_lower_ref.npz) hold to the last bit wherever libm agrees.
// Reference: tasks/amp/tocabi_amp_lower_base.py:918-962 (compute_humanoid_observations), :964-1023 (compute_humanoid_reward),
// :1025-1069 (compute_humanoid_reset); tasks/tocabi_new_walk.py:384-496 (compute_humanoid_walk_reward);
// python/isaacgym/torch_utils.py:72-81 (quat_rotate_inverse), :227-273 (quat2euler); utils/torch_jit_utils.py:80-135
// (scale_transform, saturate), :141-160 (quat_diff_rad), :406-418 (sync_reward); tasks/tocabi_amp_lower.py:310-350
// (build_amp_observations: the discriminator's per-step observation of the AMP subclass) with utils/torch_jit_utils.py:199-209
// (my_quat_rotate), :333-368 (calc_heading, calc_heading_quat_inv), python/isaacgym/torch_utils.py:44-46,92-102 (normalize,
// quat_unit, quat_from_angle_axis).
#pragma once

#include "dw_task.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace dwa {

using dw::norm_t;

// a - b + c with a = v (2 w^2 - 1), b = cross(q_vec, v) w 2, c = q_vec (q_vec . v) 2
DW_HD void quat_rotate_inverse(const float *q /* xyzw */, const float *v, float *o) {
    const float w = q[3];
    const float s = 2.0f * (w * w) - 1.0f;
    // torch.cross on the CPU contracts the first product into the subtraction: fma(a_i, b_j, -(a_j b_i)) (found by matching the reference's bits)
    const float cr[3] = {fmaf(q[1], v[2], -(q[2] * v[1])), fmaf(q[2], v[0], -(q[0] * v[2])), fmaf(q[0], v[1], -(q[1] * v[0]))};
    const float dot = (q[0] * v[0] + q[1] * v[1]) + q[2] * v[2];
    for (int i = 0; i < 3; ++i) {
        const float a = v[i] * s;
        const float b = cr[i] * w * 2.0f;
        const float c = q[i] * dot * 2.0f;
        o[i] = a - b + c;
    }
}

DW_HD void quat2euler(const float *q, float *e) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float m00 = w * w + x * x - y * y - z * z;
    const float m01 = 2 * x * y - 2 * w * z;
    const float m10 = 2 * x * y + 2 * w * z;
    const float m11 = w * w - x * x + y * y - z * z;
    const float m20 = 2 * x * z - 2 * w * y;
    const float m21 = 2 * y * z + 2 * w * x;
    const float m22 = w * w - x * x - y * y + z * z;
    const float cy = sqrtf(m00 * m00 + m10 * m10);
    const bool cond = cy > (float)(2.220446049250313e-16 * 4);
    e[2] = cond ? atan2f(m10, m00) : atan2f(-m01, m11);
    e[1] = atan2f(-m20, cy);
    e[0] = cond ? atan2f(m21, m22) : 0.0f;
}

struct ObsArgs {
    int n;
    const float *root_states, *rootvel_noise, *dof_pos, *dof_pos_bias, *quat_bias, *dof_vel, *commands;
    float *obs;
};
// one env's 36-word observation from its rows (dof_pos / dof_vel: the env's 33-word rows)
DW_HD void observations_row(const float *r, const float *nz, const float *dof_pos, const float *dof_pos_bias, const float *quat_bias,
                            const float *dof_vel, const float *commands, float *o) {
    float q[4] = {r[3], r[4], r[5], r[6]}, eu[3], vel[3], lv[3];
    quat2euler(q, eu);
    for (int i = 0; i < 3; ++i) o[i] = eu[i] + quat_bias[i];
    for (int i = 0; i < 3; ++i) vel[i] = r[7 + i] + nz[i];
    quat_rotate_inverse(q, vel, lv);
    for (int i = 0; i < 3; ++i) o[3 + i] = lv[i];
    for (int i = 0; i < 3; ++i) o[6 + i] = r[10 + i] + nz[3 + i];
    for (int i = 0; i < 3; ++i) o[9 + i] = commands[i];
    for (int i = 0; i < 12; ++i) o[12 + i] = dof_pos[i] + dof_pos_bias[i];
    for (int i = 0; i < 12; ++i) o[24 + i] = dof_vel[i];
}
DW_HD void observations(const ObsArgs &A, int e) {
    observations_row(A.root_states + 13 * (size_t)e, A.rootvel_noise + 6 * (size_t)e, A.dof_pos + DW_NUM_DOF * (size_t)e, A.dof_pos_bias + 12 * (size_t)e,
                     A.quat_bias + 3 * (size_t)e, A.dof_vel + DW_NUM_DOF * (size_t)e, A.commands + 3 * (size_t)e, A.obs + DW_AMP_NUM_OBS1 * (size_t)e);
}

// my_quat_rotate(q, v) = a + b + c (utils/torch_jit_utils.py:199-209; the sign of b is what separates it from quat_rotate_inverse)
DW_HD void my_quat_rotate(const float *q /* xyzw */, const float *v, float *o) {
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

// calc_heading_quat_inv(q): the rotation about z that takes the base's heading back to the x axis
DW_HD void heading_quat_inv(const float *q, float *hq) {
    const float ref[3] = {1.0f, 0.0f, 0.0f};
    float rd[3];
    my_quat_rotate(q, ref, rd);
    const float heading = atan2f(rd[1], rd[0]);
    const float theta = (-heading) / 2.0f;
    const float sn = sinf(theta), cs = cosf(theta);
    // quat_from_angle_axis: normalize((0, 0, 1)) = (0, 0, 1) / 1 exactly; the zeros keep the sign the product gives them
    float u[4] = {0.0f * sn, 0.0f * sn, 1.0f * sn, cs};
    float nn = norm_t(u, 4);
    nn = nn < 1e-9f ? 1e-9f : nn;
    for (int i = 0; i < 4; ++i) hq[i] = u[i] / nn;
}

struct DiscObsArgs {
    int n;
    const float *root_states, *dof_pos, *dof_vel;
    int dof_row_stride, dof_elem_stride;       // elements between two envs' rows / between two dofs of a row
    int local_root_obs;
    const float *key_pos;
    int n_key;
    float *obs;
};
DW_HD void disc_observations_row(const float *r, const float *dp, const float *dv, int dof_elem_stride, int local_root_obs, const float *key_pos,
                                 int n_key, float *o) {
    const float q[4] = {r[3], r[4], r[5], r[6]};
    float hq[4], eu[3];
    heading_quat_inv(q, hq);
    quat2euler(q, eu);
    o[0] = r[2];
    for (int i = 0; i < 3; ++i) o[1 + i] = eu[i];
    for (int i = 0; i < 12; ++i) o[4 + i] = dp[(size_t)dof_elem_stride * i];
    for (int i = 0; i < 12; ++i) o[16 + i] = dv[(size_t)dof_elem_stride * i];
    for (int k = 0; k < n_key; ++k) {
        const float *kp = key_pos + (size_t)k * 3;
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
DW_HD void disc_observations(const DiscObsArgs &A, int e) {
    disc_observations_row(A.root_states + 13 * (size_t)e, A.dof_pos + (size_t)A.dof_row_stride * e, A.dof_vel + (size_t)A.dof_row_stride * e,
                          A.dof_elem_stride, A.local_root_obs, A.key_pos + (size_t)A.n_key * 3 * e, A.n_key,
                          A.obs + (size_t)(DW_AMP_DISC_BASE + 3 * A.n_key) * e);
}

struct RewardArgs {
    int n;
    const float *root_states, *dof_vel, *dof_vel_pre, *commands, *actions, *actions_pre, *motor_efforts, *contact_force, *total_mass;
    float *reward, *reward_values;
};
// (fl, fr: the vertical contact force on the two foot links, contact_force[8][2] and [16][2])
DW_HD void reward_row(const float *r, const float *dvp, int dv_stride, const float *dvq, const float *cmd, const float *act, const float *act_pre,
                      const float *motor_efforts, float fl, float fr, float total_mass, float *reward, float *rv) {
    const float q[4] = {r[3], r[4], r[5], r[6]}, v[3] = {r[7], r[8], r[9]};
    float lv[3];
    quat_rotate_inverse(q, v, lv);
    float d = cmd[0] - lv[0];
    const float rx = 0.8f * expf(-6.0f * (d * d));
    d = cmd[1] - lv[1];
    const float ry = 0.8f * expf(-6.0f * (d * d));
    d = cmd[2] - r[12];
    const float ryaw = 0.6f * expf(-7.0f * (d * d));
    const float thr = (float)(1.4 * 9.81) * total_mass;
    const bool thres = (fl > thr) || (fr > thr);
    const float r_thr = -0.2f * (thres ? 1.0f : 0.0f);
    float cl = fl - thr, cr = fr - thr;
    cl = cl < 0.0f ? 0.0f : cl;
    cr = cr < 0.0f ? 0.0f : cr;
    const float nl = norm_t(&cl, 1), nr = norm_t(&cr, 1);
    const float pen = 0.1f * (1.0f - expf(-0.007f * (nl + nr)));
    const float r_pen = thres ? pen : 0.1f * 1.0f;
    float dv[DW_NUM_DOF], dd[DW_NUM_DOF], ta[12], td[12];
    for (int i = 0; i < DW_NUM_DOF; ++i) { dv[i] = dvp[(size_t)dv_stride * i]; dd[i] = dvp[(size_t)dv_stride * i] - dvq[i]; }
    const float nv = norm_t(dv, DW_NUM_DOF), na = norm_t(dd, DW_NUM_DOF);
    const float r_jv = 0.05f * expf(-0.01f * (nv * nv));
    const float r_ja = 0.05f * expf(-20.0f * (na * na));
    for (int i = 0; i < 12; ++i) {
        const float a = act[i], ap = act_pre[i], m = motor_efforts[i];
        ta[i] = a * m; td[i] = (a - ap) * m;
    }
    const float r_t = 0.08f * expf(-0.05f * norm_t(ta, 12));
    const float r_td = 0.6f * expf(-0.01f * norm_t(td, 12));
    float rew = 0.0f;
    rew += rx;
    rew += ryaw;
    rew += (r_thr + r_pen);
    rew += r_jv;
    rew += r_ja;
    rew += r_t;
    rew += r_td;
    *reward = rew;
    rv[0] = rx; rv[1] = ry; rv[2] = ryaw; rv[3] = r_thr; rv[4] = r_pen; rv[5] = r_jv; rv[6] = r_ja; rv[7] = r_t; rv[8] = r_td;
}
DW_HD void reward(const RewardArgs &A, int e) {
    reward_row(A.root_states + 13 * (size_t)e, A.dof_vel + DW_NUM_DOF * (size_t)e, 1, A.dof_vel_pre + DW_NUM_DOF * (size_t)e, A.commands + 3 * (size_t)e,
               A.actions + 12 * (size_t)e, A.actions_pre + 12 * (size_t)e, A.motor_efforts, A.contact_force[((size_t)DW_NUM_BODIES * e + 8) * 3 + 2],
               A.contact_force[((size_t)DW_NUM_BODIES * e + 16) * 3 + 2], A.total_mass[e],
               A.reward + e, A.reward_values + 9 * (size_t)e);
}

struct ResetArgs {
    int n;
    const int64_t *progress_buf;
    const float *contact_buf;
    const int32_t *contact_body_ids;
    int n_contact_ids;
    const float *rigid_body_pos, *rigid_body_rot;
    float max_episode_length;
    int enable_early_termination;
    float termination_height;
    int64_t *reset, *terminated;
};
DW_HD void reset(const ResetArgs &A, int e) {
    int64_t term = 0;
    if (A.enable_early_termination) {
        bool fall_contact = false;
        const float *cf = A.contact_buf + (size_t)DW_NUM_BODIES * 3 * e;
        for (int b = 0; b < DW_NUM_BODIES; ++b) {
            bool support = false;
            for (int k = 0; k < A.n_contact_ids; ++k) support = support || A.contact_body_ids[k] == b;
            if (!support) { for (int i = 0; i < 3; ++i) fall_contact = fall_contact || cf[3 * b + i] > 1.0f; }
        }
        const float *bp = A.rigid_body_pos + (size_t)DW_NUM_BODIES * 3 * e;
        bool fall_height = bp[0 * 3 + 2] < A.termination_height;
        fall_height = fall_height || bp[8 * 3 + 2] > 0.5f || bp[16 * 3 + 2] > 0.5f;
        bool fallen = fall_contact || fall_height;
        const float *rq = A.rigid_body_rot + (size_t)DW_NUM_BODIES * 4 * e;
        const float q0[4] = {rq[0], rq[1], rq[2], rq[3]};
        fallen = fallen || fabsf(dw::quat_err(q0)) > (float)(3.141592 / 4.0);
        fallen = fallen && (A.progress_buf[e] > 1);
        term = fallen ? 1 : 0;
    }
    A.terminated[e] = term;
    A.reset[e] = ((float)A.progress_buf[e] >= A.max_episode_length - 1.0f) ? 1 : term;
}

DW_HD float sync_reward(float phase) {
    const float a = (float)(1.0 / 12), b = (float)(5.0 / 12);
    float r = 1.0f;
    if (phase < a) r = 1.0f - phase * 24.0f;
    if (a <= phase && phase < b) r = -1.0f;
    if (b <= phase && phase < 0.5f) r = 24.0f * phase - 11.0f;
    return r;
}

struct NewWalkArgs {
    int n;
    const int64_t *reset_buf, *progress_buf;
    const float *target_vel, *root_pose_states, *joint_position_states, *joint_velocity_states;
    const int32_t *non_feet_idxs;
    int n_non_feet;
    const float *contact_forces;
    int num_bodies;
    float termination_height, death_cost, max_episode_length;
    const float *q_nominal;
    int num_dof;
    const float *head_states, *lfoot_states, *rfoot_states, *phase;
    float *total_reward;
    int64_t *reset;
    float *reward8;
};
constexpr int NW_MAX_DOF = 64;
DW_HD void newwalk_reward(const NewWalkArgs &A, int e) {
    const float pi = (float)3.14159265358979;
    const float phase = A.phase[e];
    const float *cf = A.contact_forces + (size_t)A.num_bodies * 3 * e;
    const float lfn = fabsf(cf[7 * 3 + 2]), rfn = fabsf(cf[14 * 3 + 2]);
    bool fly = (lfn + rfn) == 0.0f;
    const bool non_init = ((0.04f < phase) && (phase < 0.5f)) || (phase > 0.54f);
    fly = non_init ? fly : false;
    const float *lf = A.lfoot_states + 13 * (size_t)e, *rf = A.rfoot_states + 13 * (size_t)e;
    const float ug = (float)(100 * 9.81 * 0.5) * 1.0f, uv = 0.3f * 1.0f;
    float t;
    t = fmaxf(fminf(lfn, ug), 0.0f); const float gl = 2.0f * (t - (ug + 0.0f) * 0.5f) / (ug - 0.0f);
    t = fmaxf(fminf(rfn, ug), 0.0f); const float gr = 2.0f * (t - (ug + 0.0f) * 0.5f) / (ug - 0.0f);
    t = fmaxf(fminf(lf[7], uv), 0.0f); const float vl = 2.0f * (t - (uv + 0.0f) * 0.5f) / (uv - 0.0f);
    t = fmaxf(fminf(rf[7], uv), 0.0f); const float vr = 2.0f * (t - (uv + 0.0f) * 0.5f) / (uv - 0.0f);
    const float sgl = sync_reward(phase), sgr = sync_reward(phase > 0.5f ? phase - 0.5f : phase + 0.5f);
    const float svl = -sgl, svr = -sgr;
    const float grf = (tanf(pi / 4 * sgl * gl) + tanf(pi / 4 * sgr * gr)) / 2;
    const float spd = (tanf(pi / 4 * svl * vl) + tanf(pi / 4 * svr * vr)) / 2;
    const float *rp = A.root_pose_states + 13 * (size_t)e;
    const float d2[2] = {rp[7] - A.target_vel[2 * (size_t)e], rp[8] - A.target_vel[2 * (size_t)e + 1]};
    float nn = norm_t(d2, 2);
    const float rvel = expf(-10 * (nn * nn));
    const float av[3] = {rp[10], rp[11], rp[12]};
    nn = norm_t(av, 3);
    const float rang = expf(-10 * (nn * nn));
    const float hd = rp[2] - 1.0f;
    const float rh = expf(-40 * (hd * hd));
    const float du[2] = {rp[0] - A.head_states[13 * (size_t)e], rp[1] - A.head_states[13 * (size_t)e + 1]};
    nn = norm_t(du, 2);
    const float rup = expf(-10 * (nn * nn));
    // (the two num_dof-long operands are produced element by element, in norm_t's summation order: as local arrays with a run-time
    //  length they lived in scratch memory, 528 B per lane)
    nn = dw::norm_fn([&](int i) { return A.joint_position_states[(size_t)A.num_dof * e + i] - A.q_nominal[i]; }, A.num_dof);
    const float rpost = expf(-(nn * nn));
    nn = dw::norm_fn([&](int i) { return A.joint_velocity_states[(size_t)A.num_dof * e + i]; }, A.num_dof);
    const float rjv = expf(-5e-6f * (nn * nn));
    float r8[8] = {grf, spd, rvel, rang, rh, rup, rpost, rjv};
    float tot = 0.225f * grf + 0.225f * spd + 0.1f * rvel + 0.1f * rang + 0.05f * rh + 0.1f * rup + 0.1f * rpost + 0.1f * rjv;
    const float dl[2] = {lf[0] - rf[0], lf[1] - rf[1]};
    const float leg_len = norm_t(dl, 2);
    bool coll = false;
    for (int k = 0; k < A.n_non_feet; ++k) {
        int bi = A.non_feet_idxs[k];
        bi = bi < 0 ? 0 : (bi >= A.num_bodies ? A.num_bodies - 1 : bi);          // (ids come from device memory: never index past the env's rows)
        const float f3[3] = {cf[3 * bi], cf[3 * bi + 1], cf[3 * bi + 2]};
        coll = coll || norm_t(f3, 3) > 1.0f;
    }
    const bool low = rp[2] < A.termination_height, conv = leg_len < 0.1f;
    if (low || conv || coll || fly) { tot = 1.0f * A.death_cost; for (int i = 0; i < 8; ++i) r8[i] = 1.0f * A.death_cost; }
    // (the reference re-derives `reset` from reset_buf on its second line, so the height test does not survive it)
    int64_t rs = conv ? 1 : A.reset_buf[e];
    rs = ((float)A.progress_buf[e] >= A.max_episode_length - 1.0f) ? 1 : rs;
    rs = coll ? 1 : rs;
    rs = fly ? 1 : rs;
    A.total_reward[e] = tot;
    A.reset[e] = rs;
    for (int i = 0; i < 8; ++i) A.reward8[8 * (size_t)e + i] = r8[i];
}

// World position of the origin of the listed MOVING bodies (what refresh_rigid_body_state_tensor would return for them):
// out[e][k] = x(body k).  Thread = (env, k): the chain from the body up to the base is at most MAX_LEVELS long.
// (MM: dw::DevModel, or a copy of its first rows in faster memory with the same member names -- LegModel below)
template <class MM>
DW_HD void body_position(const MM &M, const float *root_states, const float *dof_state, int e, int body, float *out3) {
    int chain[dw::MAX_LEVELS], n = 0;
    for (int b = body; b > 0; b = M.parent[b]) chain[n++] = b;
    const float *r = root_states + 13 * (size_t)e;
    float x[3] = {r[0], r[1], r[2]}, R[9];
    {
        const float X = r[3], Y = r[4], Z = r[5], W = r[6];
        R[0] = 1 - 2 * (Y * Y + Z * Z); R[1] = 2 * (X * Y - W * Z); R[2] = 2 * (X * Z + W * Y);
        R[3] = 2 * (X * Y + W * Z); R[4] = 1 - 2 * (X * X + Z * Z); R[5] = 2 * (Y * Z - W * X);
        R[6] = 2 * (X * Z - W * Y); R[7] = 2 * (Y * Z + W * X); R[8] = 1 - 2 * (X * X + Y * Y);
    }
    // as the physics (oracle/dw_physics.c): x_b = x_p + R_p pos_b,  R_b = R_p rot0_b Rot(axis_b, q_b)
    for (int k = n - 1; k >= 0; --k) {
        const int b = chain[k];
        for (int i = 0; i < 3; ++i) x[i] += R[3 * i] * M.pos[b][0] + R[3 * i + 1] * M.pos[b][1] + R[3 * i + 2] * M.pos[b][2];
        if (k == 0) break;              // (the body's own rotation does not move its origin)
        const float *s = M.axis[b];
        const float ang = dof_state[((size_t)DW_NUM_DOF * e + (b - 1)) * 2];
        const float sn = sinf(ang), cs = cosf(ang), oc = 1.0f - cs;
        const float Rj[9] = {cs + oc * s[0] * s[0], oc * s[0] * s[1] - sn * s[2], oc * s[0] * s[2] + sn * s[1],
                             oc * s[1] * s[0] + sn * s[2], cs + oc * s[1] * s[1], oc * s[1] * s[2] - sn * s[0],
                             oc * s[2] * s[0] - sn * s[1], oc * s[2] * s[1] + sn * s[0], cs + oc * s[2] * s[2]};
        float Rl[9], Rn[9];
        for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c)
            Rl[3 * i + c] = M.rot0[b][3 * i] * Rj[c] + M.rot0[b][3 * i + 1] * Rj[3 + c] + M.rot0[b][3 * i + 2] * Rj[6 + c];
        for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c)
            Rn[3 * i + c] = R[3 * i] * Rl[c] + R[3 * i + 1] * Rl[3 + c] + R[3 * i + 2] * Rl[6 + c];
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    out3[0] = x[0]; out3[1] = x[1]; out3[2] = x[2];
}

// the rows of DevModel that the chains of the two foot links run through (moving bodies 0 .. 12: base, left leg, right leg)
struct LegModel {
    static constexpr int NBODY = 13;
    int   parent[NBODY];
    float pos[NBODY][3], axis[NBODY][3], rot0[NBODY][9];
};

}  // namespace dwa

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
