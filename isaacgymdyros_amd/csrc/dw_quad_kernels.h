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
#include "dw_quad_post.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace dwq {

using dw::TaskParams;

// per-env scratch of the task phases: the four slot rows of body 0 (the base has no slot) = 16 words per env
#define DQ_ENVW(el, w) (reinterpret_cast<float *>(&L.slot[(w) >> 2][(el)])[(w) & 3])
constexpr int EW_LTP = 0, EW_MIDX = 1;       // mocap phase time, mocap row (int bits)
constexpr int EW_T0 = 2, EW_T1 = 3;          // time stamps of the two mocap rows
constexpr int EW_DL = 4, EW_SL = 5;          // torque FIFO: delay index, fill (int bits)
constexpr int WW_GATE = 15;                  // wave-wide word: perturbation gate open (env 0's scratch)

// The whole VecTask.step for 16 envs: pre_physics_step up to the substep loop (dw_task.h P1, P2: action clamp and history,
// mocap phase and target, perturbation gate and schedule), the two substeps with the actuator and encoder models, and
// post_physics_step (dw_quad_post.h).  The task record is read where needed and written ONCE, by the post phase, from its
// LDS image: what the earlier phases produce for it stays in registers (StepKeep) until the image exists.
template <bool TERRAIN>
DQ_HD void quad_step(QLds &L, const QuadModel &QM, const DevModel &M, const TaskParams &C, const DwBuffers &B,
                             const float *actions, const float *mocap, const float *noise, long long step, int wave_index) {
    QLane X;
    quad_lane_init(X, wave_index, C.num_envs, C.phys, C.friction, B);
    stage_hot(L, QM);
    const int e = X.env, f = X.j & 1;
    float *es = B.env_state + (size_t)DW_ES_WORDS * e;
    DQ_UNROLL for (int i = 0; i < 12; ++i) X.warm[i] = es[DW_ES_WARM + 12 * f + i];
    float push_x = 0.0f, push_y = 0.0f;
    const float dt = C.phys.dt;
    StepKeep KP;
    {
        // ---- pre_physics_step, per-env scalar parts on the quad's lanes (dw_task.h P1): lane 0 the mocap phase, lane 1 the
        //      push schedule; every fp32 expression as there ----
        dw::TaskBuffers TB;
        TB.b = &B; TB.actions = actions; TB.noise = noise; TB.mocap = mocap; TB.step = step;
        const dw::StepCtx K = dw::make_step_ctx(C, TB, e);
        if (X.lane == 1) DQ_ENVW(0, WW_GATE) = __builtin_bit_cast(float, dw::gate_open(C, K));
        wave_sync();
        const int open = f2i(DQ_ENVW(0, WW_GATE));
        float px = 0.0f, py = 0.0f;
        if (X.j == 0) {
            const float time = es[DW_ES_TIME];
            const int init_idx = *reinterpret_cast<const int *>(&es[DW_ES_INIT_MOCAP]);
            const float local_time = dw::remainder_t(time, K.period);
            const float ltp = dw::remainder_t(local_time + (float)init_idx * K.cdt, K.period);
            const int midx = (int)(((long long)init_idx + (long long)dw::divs(C.gpu_div, local_time, K.cdt_d)) % 3599);
            DQ_ENVW(X.el, EW_LTP) = ltp;
            DQ_ENVW(X.el, EW_MIDX) = __builtin_bit_cast(float, midx);
            const float *row0 = mocap + (size_t)midx * DW_MOCAP_COLS, *row1 = row0 + DW_MOCAP_COLS;
            const float t0 = row0[0], t1 = row1[0];
            DQ_ENVW(X.el, EW_T0) = t0;
            DQ_ENVW(X.el, EW_T1) = t1;
            DQ_ENVW(X.el, EW_DL) = es[DW_ES_DELAY_IDX];          // (int bits, moved as they are)
            DQ_ENVW(X.el, EW_SL) = es[DW_ES_SIMUL_LEN];
            const float tf0 = dw::cubic_t(ltp, t0, t1, row0[1 + 33], row1[1 + 33]);
            const float tf1 = dw::cubic_t(ltp, t0, t1, row0[1 + 34], row1[1 + 34]);
            KP.midx = midx; KP.tf0 = tf0; KP.tf1 = tf1;
        }
        if (X.j == 1) {
#define DQ_ESI(off) (*reinterpret_cast<int *>(&es[(off)]))
            // (tasks/dyros_dynamic_walk.py:438-447,489-502)
            int pert_start = DQ_ESI(DW_ES_PERT_START), pert_on = DQ_ESI(DW_ES_PERT_ON), pert_count = DQ_ESI(DW_ES_PERT_COUNT);
            int impulse = DQ_ESI(DW_ES_IMPULSE), duration = DQ_ESI(DW_ES_PERT_DURATION);
            float magnitude = es[DW_ES_MAGNITUDE], phase = es[DW_ES_PHASE];
            if (open) {
                pert_start = 1;
                if (!C.force_perturb_start && X.valid) K.gate[dw::GATE_LATCH] = 1;
            }
            if (pert_start) {
                if (dw::remainder_t(es[DW_ES_EPI_LEN], C.pert_period_f) == (float)DQ_ESI(DW_ES_PERT_TIMING)) {
                    pert_on = 1;
                    int imp = 50 + (int)(dw::noise_word(K.nz, DW_NZ_PERT + 0) * 200.0f);
                    if (imp > 249) imp = 249;
                    int dur = C.pert_dur_lo + (int)(dw::noise_word(K.nz, DW_NZ_PERT + 1) * (float)(C.pert_dur_hi - C.pert_dur_lo));
                    if (dur > C.pert_dur_hi - 1) dur = C.pert_dur_hi - 1;
                    impulse = imp;
                    duration = dur;
                    magnitude = (float)imp / ((float)dur * C.dt_policy_f);
                    phase = dw::noise_word(K.nz, DW_NZ_PERT + 2) * 2.0f * (float)3.14159265358979;
                }
                if (pert_on) {
                    pert_count += 1;
                    px = magnitude * cosf(phase);
                    py = magnitude * sinf(phase);
                }
                if (pert_count == duration) {
                    pert_on = 0;
                    pert_count = 0;
                }
            }
            KP.pert_start = pert_start; KP.pert_on = pert_on; KP.pert_count = pert_count; KP.impulse = impulse; KP.duration = duration;
            KP.magnitude = magnitude; KP.phase = phase;
#undef DQ_ESI
        }
        push_x = quad_bcast<1>(px);
        push_y = quad_bcast<1>(py);
        wave_sync();
        // ---- actions: clamp, the record, the newest slot of the action ring; items (env, action) ----
        DQ_UNROLL for (int k = 0; k < (EPW * DW_NUM_ACT + 63) / 64; ++k) {
            const int i = X.lane + 64 * k;
            const int el = i / DW_NUM_ACT, a = i - DW_NUM_ACT * el;
            const int eg = wave_index * EPW + el;
            KP.act[k] = 0.0f;
            if (i < EPW * DW_NUM_ACT) KP.act[k] = dw::clamp_action(actions, eg < C.num_envs ? eg : C.num_envs - 1, a);
            if (i < EPW * DW_NUM_ACT && eg < C.num_envs) {
                float *ei = B.env_state + (size_t)DW_ES_WORDS * eg;
                const float v = KP.act[k];
                const int head = *reinterpret_cast<const int *>(&ei[DW_ES_HIST_HEAD]);
                B.action_history[((size_t)eg * DW_HIST_SLOTS + head) * DW_NUM_ACT + a] = v;
            }
        }
    }

    // ---- actuator model, joint-parallel (items (env, dof), dw_quad.h): inputs of both substeps from one pass over the
    //      Gym tensors and the task record.  Kept per item in registers: the joint angle (integrated after each substep),
    //      the delayed leg torque of the second substep, the encoder reading of the first. ----
    float qkeep[QNI], qdkeep[QNI], tau2[QNI], qnprev[QNI], dampk[QNI], ddk[QNI], kpk[QNI], kvk[QNI];
    float (&tgt)[QNI] = KP.tgt, (&qvk)[QNI] = KP.qv;
    // (simul_len is read by every leg item of an env: it is advanced once, at the end, by the env's lane 0)
    const int simul_len0 = *reinterpret_cast<const int *>(&es[DW_ES_SIMUL_LEN]);
    (void)f;
    // items of this lane, fixed for the kernel: slot position of each (owner lookup) kept in a register
    int ipos[QNI];
    DQ_UNROLL for (int k = 0; k < QNI; ++k) ipos[k] = joint_item(L, wave_index, C.num_envs, X.lane, k).pos;
    auto item = [&](int k) {
        JointItem it;
        const int i = X.lane + 64 * k;
        it.el = i / ND; it.d = i - ND * it.el; it.b = it.d + 1;
        const int eg = wave_index * EPW + it.el;
        it.ok = (i < EPW * ND) && (eg < C.num_envs);
        if (!(i < EPW * ND)) { it.el = 0; it.d = 0; it.b = 1; }
        it.env = eg < C.num_envs ? eg : C.num_envs - 1;
        it.pos = ipos[k];
        return it;
    };
    DQ_STAMP(B, 0);
    // In groups of three items: every request of the group first -- Gym state, record fields, mocap rows, gains; leg-only
    // fields with the joint index clamped instead of a branch, which would end the run of requests -- then the arithmetic.
    DQ_UNROLL for (int g3 = 0; g3 < QNI; g3 += 3) {
        float rq[3], rqd[3], rdamp[3], rarm[3], rqpre[3], rm0[3], rm1[3], rms[3], rah[3], rac[3], rkp[3], rkv[3], rcol[3][DW_ALOG_SLOTS - 1];
        DQ_UNROLL for (int u = 0; u < 3; ++u) {
            const JointItem it = item(g3 + u);
            const size_t g = (size_t)ND * it.env + it.d;
            const float *ei = B.env_state + (size_t)DW_ES_WORDS * it.env;
            const int d = it.d, dc = d < 12 ? d : 11;
            const int midx = f2i(DQ_ENVW(it.el, EW_MIDX));
            const float *row0 = mocap + (size_t)midx * DW_MOCAP_COLS;
            rq[u] = B.dof_state[g * 2]; rqd[u] = B.dof_state[g * 2 + 1];
            rdamp[u] = B.dof_damping[g]; rarm[u] = B.dof_armature[g];
            rqpre[u] = ei[DW_ES_QPOS_PRE + d];
            rm0[u] = row0[1 + d]; rm1[u] = row0[DW_MOCAP_COLS + 1 + d];
            rms[u] = ei[DW_ES_MOTOR_SCALE + dc]; rah[u] = M.action_high[dc];
            rac[u] = actions[DW_NUM_ACT * it.env + dc];
            rkp[u] = M.kp[d]; rkv[u] = M.kv[d];
            DQ_UNROLL for (int s = 0; s < DW_ALOG_SLOTS - 1; ++s) rcol[u][s] = ei[DW_ES_ACTION_LOG + 12 * (s + 1) + dc];
        }
        DQ_UNROLL for (int u = 0; u < 3; ++u) {
            const int k = g3 + u;
            const JointItem it = item(k);
            const int d = it.d;
            const float q = rq[u], qd = rqd[u], damp = rdamp[u], arm = rarm[u];
            qkeep[k] = q; qdkeep[k] = qd;
            qnprev[k] = rqpre[u];
            dampk[k] = damp; ddk[k] = arm + dt * damp; kpk[k] = rkp[u]; kvk[k] = rkv[u];
            // mocap target of this joint (cubic between two table rows, dw_task.h P2) and, for the legs, the action torque
            const float target = dw::cubic_t(DQ_ENVW(it.el, EW_LTP), DQ_ENVW(it.el, EW_T0), DQ_ENVW(it.el, EW_T1), rm0[u], rm1[u]);
            const float atq = d < 12 ? fminf(fmaxf(rac[u], -1.0f), 1.0f) * rms[u] * rah[u] : 0.0f;
            tgt[k] = target;
            KP.atq[k] = atq;
            // torque FIFO, column d (tasks/dyros_dynamic_walk.py:511-519): shift, append, pick the delayed slot -- twice, for
            // the two substeps (the action torque of the step is appended both times); the record gets the final column
            const int dl = f2i(DQ_ENVW(it.el, EW_DL)), sl0 = f2i(DQ_ENVW(it.el, EW_SL));
            float col[DW_ALOG_SLOTS + 1];
            DQ_UNROLL for (int s = 0; s < DW_ALOG_SLOTS - 1; ++s) col[s] = rcol[u][s];
            col[DW_ALOG_SLOTS - 1] = atq;
            col[DW_ALOG_SLOTS] = col[DW_ALOG_SLOTS - 1];
            int sl1 = sl0 + 1; if (sl1 > DW_ALOG_SLOTS) sl1 = DW_ALOG_SLOTS;
            int sl2 = sl1 + 1; if (sl2 > DW_ALOG_SLOTS) sl2 = DW_ALOG_SLOTS;
            const int src1 = sl1 > dl ? dl : DW_ALOG_SLOTS - sl1, src2 = sl2 > dl ? dl : DW_ALOG_SLOTS - sl2;
            float t1 = col[0], t2 = col[1];
            DQ_UNROLL for (int s = 1; s < DW_ALOG_SLOTS; ++s) { t1 = (s == src1) ? col[s] : t1; t2 = (s == src2) ? col[s + 1] : t2; }
            // upper body: PD to the mocap target; the second substep forms its own torque from the new state
            const float tau = d < 12 ? t1 : rkp[u] * (target - q) + rkv[u] * (-qd);
            tau2[k] = d < 12 ? t2 : target;
            if (X.lane + 64 * k < EPW * ND) DQ_SLOT(it.b, 0, it.pos) = mk4(q, qd, tau - damp * qd, ddk[k]);
        }
    }
    wave_sync();

    for (int sub = 0; sub < 2; ++sub) {
        X.stamp_base = 1 + 16 * sub;
        if (!C.freeze_physics) quad_substep<TERRAIN>(L, QM, M, C.phys, X, B, sub == 0 ? push_x : 0.0f, sub == 0 ? push_y : 0.0f, sub == 1);
        wave_sync();
        // ---- integrate the joints, encoder model (tasks/dyros_dynamic_walk.py:527-530), inputs of the next substep: the
        //      slot reads and the noise of all items first, then the arithmetic, then the stores ----
        F4 fin[QNI];
        float nzw[QNI];
        DQ_UNROLL for (int k = 0; k < QNI; ++k) { const JointItem it = item(k); fin[k] = DQ_LD(it.b, 0, it.pos); }      // {qlo, qd, qhi, *}
        if (noise) {
            DQ_UNROLL for (int k = 0; k < QNI; ++k) { const JointItem it = item(k); nzw[k] = noise[(size_t)DW_NOISE_WORDS * it.env + DW_NZ_ENC + ND * sub + it.d]; }
        } else {
            DQ_UNROLL for (int k = 0; k < QNI; ++k) {
                const JointItem it = item(k);
                dw::NoiseSrc nz;
                nz.rec = nullptr; nz.seed = C.seed; nz.env = (unsigned int)it.env; nz.step = (unsigned long long)step; nz.stream = 0;
                nzw[k] = dw::noise_word(nz, DW_NZ_ENC + ND * sub + it.d);
            }
        }
        DQ_UNROLL for (int k = 0; k < QNI; ++k) {
            const JointItem it = item(k);
            const size_t g = (size_t)ND * it.env + it.d;
            const int d = it.d;
            float q = qkeep[k], qd = 0.0f;
            if (!C.freeze_physics) {
                qd = fin[k].y; q = qkeep[k] + dt * qd;
                if (q < fin[k].x) { q = fin[k].x; if (qd < 0) qd = 0; }
                if (q > fin[k].z) { q = fin[k].z; if (qd > 0) qd = 0; }
                qkeep[k] = q; qdkeep[k] = qd;
                if (it.ok && sub == 1) { B.dof_state[g * 2] = q; B.dof_state[g * 2 + 1] = qd; }
            }
            const float qn = q + fminf(fmaxf(nzw[k], -0.00016f), 0.00016f);
            const float qv = C.gpu_div ? (qn - qnprev[k]) * C.inv_dt_f : (qn - qnprev[k]) / dt;
            qnprev[k] = qn;
            qvk[k] = qv;
            if (sub == 0 && !C.freeze_physics) {
                const float tau = d < 12 ? tau2[k] : kpk[k] * (tau2[k] - q) + kvk[k] * (-qd);
                if (X.lane + 64 * k < EPW * ND) DQ_SLOT(it.b, 0, it.pos) = mk4(q, qd, tau - dampk[k] * qd, ddk[k]);
            }
        }
        wave_sync();
        DQ_STAMP(B, 1 + 16 * sub + 14);
    }
    if (X.valid && !C.freeze_physics && X.j == 0) { DQ_UNROLL for (int i = 0; i < 13; ++i) B.root_states[(size_t)13 * e + i] = X.root[i]; }
    KP.simul_len = simul_len0 + 2 > DW_ALOG_SLOTS ? DW_ALOG_SLOTS : simul_len0 + 2;
    DQ_UNROLL for (int k = 0; k < QNI; ++k) KP.qn[k] = qnprev[k];
    DQ_STAMP(B, 40);
    {
        // every lane of the quad may have seen a non-sole body in contact: one flag per env
        float c = X.coll ? 1.0f : 0.0f;
        c += quad_xor1(c);
        c += quad_xor2(c);
        X.coll = c > 0.0f;
    }
    wave_sync();
    quad_task_post<TERRAIN>(L, M, C, B, actions, noise, step, wave_index, X, qkeep, qdkeep, KP);
    DQ_STAMP(B, 41);
}

}  // namespace dwq

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
