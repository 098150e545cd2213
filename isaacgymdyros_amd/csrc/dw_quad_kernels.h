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
    const int e = X.env, f = X.j & 1;
    float *es = B.env_state + (size_t)DW_ES_WORDS * e;
    dw::NoiseSrc nz;
    nz.rec = noise ? noise + (size_t)DW_NOISE_WORDS * e : nullptr;
    nz.seed = C.seed; nz.env = (unsigned int)e; nz.step = (unsigned long long)step; nz.stream = 0;
    DQ_UNROLL for (int i = 0; i < 12; ++i) X.warm[i] = es[DW_ES_WARM + 12 * f + i];
    const float push_x = es[dw::ES_PUSH_X], push_y = es[dw::ES_PUSH_Y];
    const int dl = *reinterpret_cast<const int *>(&es[DW_ES_DELAY_IDX]);
    int simul_len = *reinterpret_cast<const int *>(&es[DW_ES_SIMUL_LEN]);
    const float dt = C.phys.dt;

    for (int sub = 0; sub < 2; ++sub) {
        // ---- actuator model: joint efforts of this substep (dw_task.h P3), slot quad 0 = {q, qd, tau - damping qd, armature + dt damping} ----
        int sl = simul_len + 1;
        if (sl > DW_ALOG_SLOTS) sl = DW_ALOG_SLOTS;
        for (int s = 0; s < QM.nsteps; ++s) {
            const int b = QM.fk[s][X.j].body;
            if (b >= 0) {
                const int d = b - 1;
                float q, qd;
                if (sub == 0) { q = B.dof_state[((size_t)ND * e + d) * 2]; qd = B.dof_state[((size_t)ND * e + d) * 2 + 1]; }
                else { const F4 o = DQ_SLOT(b, 0, X.pos); q = o.x; qd = o.y; }
                float tau;
                if (d < 12) {
                    // torque FIFO, column d (tasks/dyros_dynamic_walk.py:511-519): shift, append, pick the delayed slot
                    float col[DW_ALOG_SLOTS];
                    DQ_UNROLL for (int k = 0; k < DW_ALOG_SLOTS - 1; ++k) col[k] = es[DW_ES_ACTION_LOG + 12 * (k + 1) + d];
                    col[DW_ALOG_SLOTS - 1] = es[DW_ES_ACTION_TORQUE + d];
                    if (X.valid) { DQ_UNROLL for (int k = 0; k < DW_ALOG_SLOTS; ++k) es[DW_ES_ACTION_LOG + 12 * k + d] = col[k]; }
                    const int src = sl > dl ? dl : DW_ALOG_SLOTS - sl;
                    float t = col[0];
                    DQ_UNROLL for (int k = 1; k < DW_ALOG_SLOTS; ++k) t = (k == src) ? col[k] : t;
                    tau = t;
                } else {
                    tau = M.kp[d] * (es[DW_ES_TARGET_QPOS + d] - q) + M.kv[d] * (-qd);
                }
                const float damp = B.dof_damping[(size_t)ND * e + d], arm = B.dof_armature[(size_t)ND * e + d];
                DQ_SLOT(b, 0, X.pos) = mk4(q, qd, tau - damp * qd, arm + dt * damp);
            }
        }
        if (!C.freeze_physics) quad_substep<TERRAIN>(L, QM, M, C.phys, X, B, sub == 0 ? push_x : 0.0f, sub == 0 ? push_y : 0.0f, sub == 1);
        // ---- encoder model (tasks/dyros_dynamic_walk.py:527-530) ----
        for (int s = 0; s < QM.nsteps; ++s) {
            const int b = QM.fk[s][X.j].body;
            if (b >= 0) {
                const int d = b - 1;
                const float qnew = C.freeze_physics ? B.dof_state[((size_t)ND * e + d) * 2] : DQ_SLOT(b, 0, X.pos).x;
                const float n = dw::noise_word(nz, DW_NZ_ENC + ND * sub + d);
                const float qn = qnew + fminf(fmaxf(n, -0.00016f), 0.00016f);
                const float pre = es[DW_ES_QPOS_PRE + d];
                const float qv = C.gpu_div ? (qn - pre) * C.inv_dt_f : (qn - pre) / dt;
                if (X.valid) { es[DW_ES_QVEL_NOISE + d] = qv; es[DW_ES_QPOS_NOISE + d] = qn; es[DW_ES_QPOS_PRE + d] = qn; }
            }
        }
        simul_len = sl;
    }
    if (X.valid) {
        if (X.j == 0) {
            *reinterpret_cast<int *>(&es[DW_ES_SIMUL_LEN]) = simul_len;
            if (!C.freeze_physics) { DQ_UNROLL for (int i = 0; i < 13; ++i) B.root_states[(size_t)13 * e + i] = X.root[i]; }
        }
        if (X.j < 2 && !C.freeze_physics) { DQ_UNROLL for (int i = 0; i < 12; ++i) es[DW_ES_WARM + 12 * f + i] = X.warm[i]; }
    }
}

}  // namespace dwq

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
