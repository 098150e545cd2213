// dw_quad_kernels.h -- the policy step of the split pipeline, physics part: for 16 envs per wavefront, the two physics
// substeps of VecTask.step with the actuator model around them (reference tasks/dyros_dynamic_walk.py:504-530: upper-body
// PD, 6-slot torque FIFO with per-env delay, simulate, encoder model), on the quad layout of dw_quad.h.  dw_k_pre has
// already written this step's action torques, mocap target and push into the task record; dw_k_post consumes the state,
// the net contact forces and the encoder fields this kernel leaves behind.
//
// Every fp32 expression that the reference pins bit for bit (tau per substep, qpos_noise, qvel_noise: SURVEY 8c) is
// written exactly as in dw_task.h P3 (fp contraction off in this region of the file).
#pragma once

#include "dw_quad.h"
#include "dw_task.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace dwq {

using dw::TaskParams;

template <bool TERRAIN>
DQ_HD void quad_physics_step(QLds &L, const QuadModel &QM, const DevModel &M, const TaskParams &C, const DwBuffers &B,
                             const float *noise, long long step, int wave_index) {
    QLane X;
    quad_lane_init(X, wave_index, C.num_envs, C.phys, C.friction, B);
    stage_hot(L, QM);
    const int e = X.env, f = X.j & 1;
    float *es = B.env_state + (size_t)DW_ES_WORDS * e;
    DQ_UNROLL for (int i = 0; i < 12; ++i) X.warm[i] = es[DW_ES_WARM + 12 * f + i];
    const float push_x = es[dw::ES_PUSH_X], push_y = es[dw::ES_PUSH_Y];
    const float dt = C.phys.dt;

    // ---- actuator model, joint-parallel (items (env, dof), dw_quad.h): inputs of both substeps from one pass over the
    //      Gym tensors and the task record.  Kept per item in registers: the joint angle (integrated after each substep),
    //      the delayed leg torque of the second substep, the encoder reading of the first. ----
    float qkeep[QNI], tau2[QNI], qnprev[QNI];
    // (simul_len is read by every leg item of an env: it is advanced once, at the end, by the env's lane 0)
    const int simul_len0 = *reinterpret_cast<const int *>(&es[DW_ES_SIMUL_LEN]);
    DQ_STAMP(B, 0);
    DQ_UNROLL for (int k = 0; k < QNI; ++k) {
        const JointItem it = joint_item(L, wave_index, C.num_envs, X.lane, k);
        const size_t g = (size_t)ND * it.env + it.d;
        float *ei = B.env_state + (size_t)DW_ES_WORDS * it.env;
        const int d = it.d;
        const float q = B.dof_state[g * 2], qd = B.dof_state[g * 2 + 1];
        const float damp = B.dof_damping[g], arm = B.dof_armature[g];
        qkeep[k] = q;
        qnprev[k] = ei[DW_ES_QPOS_PRE + d];
        float tau;
        if (d < 12) {
            // torque FIFO, column d (tasks/dyros_dynamic_walk.py:511-519): shift, append, pick the delayed slot -- twice, for
            // the two substeps (the action torque of the step is appended both times); the record gets the final column
            const int dl = *reinterpret_cast<const int *>(&ei[DW_ES_DELAY_IDX]);
            const int sl0 = *reinterpret_cast<const int *>(&ei[DW_ES_SIMUL_LEN]);
            float col[DW_ALOG_SLOTS + 1];
            DQ_UNROLL for (int s = 0; s < DW_ALOG_SLOTS - 1; ++s) col[s] = ei[DW_ES_ACTION_LOG + 12 * (s + 1) + d];
            col[DW_ALOG_SLOTS - 1] = ei[DW_ES_ACTION_TORQUE + d];
            col[DW_ALOG_SLOTS] = col[DW_ALOG_SLOTS - 1];
            int sl1 = sl0 + 1; if (sl1 > DW_ALOG_SLOTS) sl1 = DW_ALOG_SLOTS;
            int sl2 = sl1 + 1; if (sl2 > DW_ALOG_SLOTS) sl2 = DW_ALOG_SLOTS;
            const int src1 = sl1 > dl ? dl : DW_ALOG_SLOTS - sl1, src2 = sl2 > dl ? dl : DW_ALOG_SLOTS - sl2;
            float t1 = col[0], t2 = col[1];
            DQ_UNROLL for (int s = 1; s < DW_ALOG_SLOTS; ++s) { t1 = (s == src1) ? col[s] : t1; t2 = (s == src2) ? col[s + 1] : t2; }
            tau = t1; tau2[k] = t2;
            if (it.ok) { DQ_UNROLL for (int s = 0; s < DW_ALOG_SLOTS; ++s) ei[DW_ES_ACTION_LOG + 12 * s + d] = col[s + 1]; }
        } else {
            tau = M.kp[d] * (ei[DW_ES_TARGET_QPOS + d] - q) + M.kv[d] * (-qd);
            tau2[k] = ei[DW_ES_TARGET_QPOS + d];          // the PD target: the second substep forms its own torque from the new state
        }
        if (X.lane + 64 * k < EPW * ND) DQ_SLOT(it.b, 0, it.pos) = mk4(q, qd, tau - damp * qd, arm + dt * damp);
    }
    wave_sync();

    for (int sub = 0; sub < 2; ++sub) {
        X.stamp_base = 1 + 16 * sub;
        if (!C.freeze_physics) quad_substep<TERRAIN>(L, QM, M, C.phys, X, B, sub == 0 ? push_x : 0.0f, sub == 0 ? push_y : 0.0f, sub == 1);
        wave_sync();
        // ---- integrate the joints, encoder model (tasks/dyros_dynamic_walk.py:527-530), inputs of the next substep ----
        DQ_UNROLL for (int k = 0; k < QNI; ++k) {
            const JointItem it = joint_item(L, wave_index, C.num_envs, X.lane, k);
            const size_t g = (size_t)ND * it.env + it.d;
            float *ei = B.env_state + (size_t)DW_ES_WORDS * it.env;
            const int d = it.d;
            float q = qkeep[k], qd = 0.0f;
            if (!C.freeze_physics) {
                joint_integrate(L, it, dt, qkeep[k], &q, &qd);
                qkeep[k] = q;
                if (it.ok && sub == 1) { B.dof_state[g * 2] = q; B.dof_state[g * 2 + 1] = qd; }
            }
            dw::NoiseSrc nz;
            nz.rec = noise ? noise + (size_t)DW_NOISE_WORDS * it.env : nullptr;
            nz.seed = C.seed; nz.env = (unsigned int)it.env; nz.step = (unsigned long long)step; nz.stream = 0;
            const float n = dw::noise_word(nz, DW_NZ_ENC + ND * sub + d);
            const float qn = q + fminf(fmaxf(n, -0.00016f), 0.00016f);
            const float qv = C.gpu_div ? (qn - qnprev[k]) * C.inv_dt_f : (qn - qnprev[k]) / dt;
            qnprev[k] = qn;
            if (it.ok && sub == 1) { ei[DW_ES_QVEL_NOISE + d] = qv; ei[DW_ES_QPOS_NOISE + d] = qn; ei[DW_ES_QPOS_PRE + d] = qn; }
            if (sub == 0 && !C.freeze_physics) {
                const float damp = B.dof_damping[g], arm = B.dof_armature[g];
                const float tau = d < 12 ? tau2[k] : M.kp[d] * (tau2[k] - q) + M.kv[d] * (-qd);
                if (X.lane + 64 * k < EPW * ND) DQ_SLOT(it.b, 0, it.pos) = mk4(q, qd, tau - damp * qd, arm + dt * damp);
            }
        }
        wave_sync();
        DQ_STAMP(B, 1 + 16 * sub + 14);
    }
    if (X.valid && X.j == 0) {
        int sl = simul_len0 + 2;
        if (sl > DW_ALOG_SLOTS) sl = DW_ALOG_SLOTS;
        *reinterpret_cast<int *>(&es[DW_ES_SIMUL_LEN]) = sl;
    }
    if (X.valid && !C.freeze_physics) {
        if (X.j == 0) { DQ_UNROLL for (int i = 0; i < 13; ++i) B.root_states[(size_t)13 * e + i] = X.root[i]; }
        if (X.j < 2) { DQ_UNROLL for (int i = 0; i < 12; ++i) es[DW_ES_WARM + 12 * f + i] = X.warm[i]; }
    }
    DQ_STAMP(B, 40);
}

}  // namespace dwq

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
