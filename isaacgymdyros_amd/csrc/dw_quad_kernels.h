// dw_quad_kernels.h -- the whole VecTask.step for the 16 envs of a quad wave (dw_quad.h), in one launch: pre_physics_step, the
// two physics substeps with the actuator model around them (reference tasks/dyros_dynamic_walk.py:449-541: mocap target,
// action torques, push, upper-body PD, 6-slot torque FIFO with per-env delay, simulate, encoder model), then
// post_physics_step (dw_quad_post.h).
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
    DQ_STAMP(B, 54);
#if defined(DQ_WAVE_TIME) && defined(__HIPCC__)          // (timing experiment: the life of every wave, tools/wave_times.py)
    const long long dq_t0 = (long long)__builtin_readcyclecounter();
#endif
    // ==== round 1 of requests: everything whose address is known at entry -- the base state, the scalars of the record the
    //      pre-physics phase reads, the actions, the inputs of the actuator model for this lane's nine (env, joint) items,
    //      the hot tables -- in one straight run, so that the wave waits for memory once.  (Each request a lone wave waits
    //      for costs ~4 k cycles, about 1 % of the step: phase_stamps) ====
    QLane X;
    quad_lane_init(X, wave_index, C.num_envs, C.phys, C.friction, B);
    const int e = X.env, f = X.j & 1;
    float *es = B.env_state + (size_t)DW_ES_WORDS * e;
    DQ_UNROLL for (int i = 0; i < 12; ++i) X.warm[i] = es[DW_ES_WARM + 12 * f + i];
    const float r_time = es[DW_ES_TIME], r_epi = es[DW_ES_EPI_LEN], r_mag = es[DW_ES_MAGNITUDE], r_phase = es[DW_ES_PHASE];
    const float r_init = es[DW_ES_INIT_MOCAP], r_pstart = es[DW_ES_PERT_START], r_pon = es[DW_ES_PERT_ON], r_pcount = es[DW_ES_PERT_COUNT];
    const float r_imp = es[DW_ES_IMPULSE], r_dur = es[DW_ES_PERT_DURATION], r_ptim = es[DW_ES_PERT_TIMING];
    const float r_dl = es[DW_ES_DELAY_IDX], r_sl = es[DW_ES_SIMUL_LEN];          // (integer fields travel as bit patterns)
    constexpr int NAI = (EPW * DW_NUM_ACT + 63) / 64;
    float r_act[NAI], r_head[NAI];
    DQ_UNROLL for (int k = 0; k < NAI; ++k) {
        const int i = X.lane + 64 * k, ic = i < EPW * DW_NUM_ACT ? i : 0;
        const int el = ic / DW_NUM_ACT, a = ic - DW_NUM_ACT * el;
        const int eg = wave_index * EPW + el, egc = eg < C.num_envs ? eg : C.num_envs - 1;
        r_act[k] = actions[DW_NUM_ACT * egc + a];
        r_head[k] = B.env_state[(size_t)DW_ES_WORDS * egc + DW_ES_HIST_HEAD];
    }
    auto item = [&](int k) {          // this lane's k-th (env, joint) item; pos is filled in below
        JointItem it;
        const int i = X.lane + 64 * k;
        it.el = i / ND; it.d = i - ND * it.el; it.b = it.d + 1;
        const int eg = wave_index * EPW + it.el;
        it.ok = (i < EPW * ND) && (eg < C.num_envs);
        if (!(i < EPW * ND)) { it.el = 0; it.d = 0; it.b = 1; }
        it.env = eg < C.num_envs ? eg : C.num_envs - 1;
        it.pos = 0;
        return it;
    };
    float rq[QNI], rqd[QNI], rdamp[QNI], rarm[QNI], rqpre[QNI], rms[QNI], rah[QNI], rac[QNI], rkp[QNI], rkv[QNI], rcol[QNI][DW_ALOG_SLOTS - 1];
    DQ_UNROLL for (int k = 0; k < QNI; ++k) {
        const JointItem it = item(k);
        const size_t g = (size_t)ND * it.env + it.d;
        const float *ei = B.env_state + (size_t)DW_ES_WORDS * it.env;
        const int d = it.d, dc = d < 12 ? d : 11;       // (leg-only fields: index clamped rather than a branch)
        rq[k] = B.dof_state[g * 2]; rqd[k] = B.dof_state[g * 2 + 1];
        rdamp[k] = B.dof_damping[g]; rarm[k] = B.dof_armature[g];
        rqpre[k] = ei[DW_ES_QPOS_PRE + d];
        rms[k] = ei[DW_ES_MOTOR_SCALE + dc]; rah[k] = M.action_high[dc];
        rac[k] = actions[DW_NUM_ACT * it.env + dc];
        rkp[k] = M.kp[d]; rkv[k] = M.kv[d];
        DQ_UNROLL for (int s = 0; s < DW_ALOG_SLOTS - 1; ++s) rcol[k][s] = ei[DW_ES_ACTION_LOG + 12 * (s + 1) + dc];
    }
    stage_hot(L, QM);
    quad_lane_second_inertial(X, L, QM, B);
    int ipos[QNI];
    DQ_UNROLL for (int k = 0; k < QNI; ++k) { const JointItem it = item(k); ipos[k] = (it.el + 4 * (L.hot.owner[it.b] >> 6)) & 15; }
    float push_x = 0.0f, push_y = 0.0f;
    const float dt = C.phys.dt;
    StepKeep KP;
    float qkeep[QNI], qdkeep[QNI], tau2[QNI], qnprev[QNI], dampk[QNI], ddk[QNI], kpk[QNI], kvk[QNI];
    float (&qvk)[QNI] = KP.qv;
    // (simul_len is read by every leg item of an env: it is advanced once, at the end, by the env's lane 0)
    const int simul_len0 = f2i(r_sl);
    (void)f;
    {
        // ---- pre_physics_step, per-env scalar parts on the quad's lanes (dw_task.h P1): lane 0 the mocap phase, lane 1 the
        //      push schedule; every fp32 expression as there ----
        dw::TaskBuffers TB;
        TB.b = &B; TB.actions = actions; TB.noise = noise; TB.mocap = mocap; TB.step = step;
        const dw::StepCtx K = dw::make_step_ctx(C, TB, e);
        if (X.lane == 1) DQ_ENVW(0, WW_GATE) = __builtin_bit_cast(float, dw::gate_open(C, K));
        if (X.j == 0) {
            const float time = r_time;
            const int init_idx = f2i(r_init);
            const float local_time = dw::remainder_t(time, K.period);
            const float ltp = dw::remainder_t(local_time + (float)init_idx * K.cdt, K.period);
            const int midx = (int)(((long long)init_idx + (long long)dw::divs(C.gpu_div, local_time, K.cdt_d)) % 3599);
            DQ_ENVW(X.el, EW_LTP) = ltp;
            DQ_ENVW(X.el, EW_MIDX) = __builtin_bit_cast(float, midx);
            DQ_ENVW(X.el, EW_DL) = r_dl;
            DQ_ENVW(X.el, EW_SL) = r_sl;
            KP.midx = midx;
        }
        wave_sync();
        const int open = f2i(DQ_ENVW(0, WW_GATE));
        // ==== round 2 of requests: the two mocap rows of each item's env (their index is this step's arithmetic) ====
        float rt0[QNI], rt1[QNI], rm0[QNI], rm1[QNI];
        DQ_UNROLL for (int k = 0; k < QNI; ++k) {
            const JointItem it = item(k);
            const float *row0 = mocap + (size_t)f2i(DQ_ENVW(it.el, EW_MIDX)) * DW_MOCAP_COLS;
            rt0[k] = row0[0]; rt1[k] = row0[DW_MOCAP_COLS];
            rm0[k] = row0[1 + it.d]; rm1[k] = row0[DW_MOCAP_COLS + 1 + it.d];
        }
        float rtf[6];
        {
            const float *row0 = mocap + (size_t)f2i(DQ_ENVW(X.el, EW_MIDX)) * DW_MOCAP_COLS, *row1 = row0 + DW_MOCAP_COLS;
            rtf[0] = row0[0]; rtf[1] = row1[0]; rtf[2] = row0[1 + 33]; rtf[3] = row1[1 + 33]; rtf[4] = row0[1 + 34]; rtf[5] = row1[1 + 34];
        }
        float px = 0.0f, py = 0.0f;
        if (X.j == 1) {
            // (tasks/dyros_dynamic_walk.py:438-447,489-502)
            int pert_start = f2i(r_pstart), pert_on = f2i(r_pon), pert_count = f2i(r_pcount);
            int impulse = f2i(r_imp), duration = f2i(r_dur);
            float magnitude = r_mag, phase = r_phase;
            if (open) {
                pert_start = 1;
                if (!C.force_perturb_start && X.valid) K.gate[dw::GATE_LATCH] = 1;
            }
            if (pert_start) {
                if (dw::remainder_t(r_epi, C.pert_period_f) == (float)f2i(r_ptim)) {
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
        }
        push_x = quad_bcast<1>(px);
        push_y = quad_bcast<1>(py);
        // ---- actions: clamp, the record, the newest slot of the action ring; items (env, action) ----
        DQ_UNROLL for (int k = 0; k < NAI; ++k) {
            const int i = X.lane + 64 * k;
            const int el = i / DW_NUM_ACT, a = i - DW_NUM_ACT * el;
            const int eg = wave_index * EPW + el;
            float v = fminf(fmaxf(r_act[k], -1.0f), 1.0f);
            if (a == 12) v = (v > 0 ? 1.0f : 0.0f) * v;
            KP.act[k] = i < EPW * DW_NUM_ACT ? v : 0.0f;
            if (i < EPW * DW_NUM_ACT && eg < C.num_envs)
                B.action_history[((size_t)eg * DW_HIST_SLOTS + f2i(r_head[k])) * DW_NUM_ACT + a] = v;
        }
        if (X.j == 0) {
            KP.tf0 = dw::cubic_t(DQ_ENVW(X.el, EW_LTP), rtf[0], rtf[1], rtf[2], rtf[3]);
            KP.tf1 = dw::cubic_t(DQ_ENVW(X.el, EW_LTP), rtf[0], rtf[1], rtf[4], rtf[5]);
        }

        // ---- actuator model, joint-parallel (items (env, dof), dw_quad.h): inputs of both substeps.  Kept per item in
        //      registers: the joint angle (integrated after each substep), the delayed leg torque of the second substep, the
        //      encoder reading of the first, damping and gains. ----
        DQ_STAMP(B, 0);
        DQ_UNROLL for (int k = 0; k < QNI; ++k) {
            JointItem it = item(k);
            it.pos = ipos[k];
            const int d = it.d;
            const float q = rq[k], qd = rqd[k], damp = rdamp[k], arm = rarm[k];
            qkeep[k] = q; qdkeep[k] = qd;
            qnprev[k] = rqpre[k];
            dampk[k] = damp; ddk[k] = arm + dt * damp; kpk[k] = rkp[k]; kvk[k] = rkv[k];
            // mocap target of this joint (cubic between two table rows, dw_task.h P2) and, for the legs, the action torque
            const float target = dw::cubic_t(DQ_ENVW(it.el, EW_LTP), rt0[k], rt1[k], rm0[k], rm1[k]);
            const float atq = d < 12 ? fminf(fmaxf(rac[k], -1.0f), 1.0f) * rms[k] * rah[k] : 0.0f;
            KP.tgt[k] = target;
            KP.atq[k] = atq;
            // torque FIFO, column d (tasks/dyros_dynamic_walk.py:511-519): shift, append, pick the delayed slot -- twice, for
            // the two substeps (the action torque of the step is appended both times); the record gets the final column
            const int dl = f2i(DQ_ENVW(it.el, EW_DL)), sl0 = f2i(DQ_ENVW(it.el, EW_SL));
            float col[DW_ALOG_SLOTS + 1];
            DQ_UNROLL for (int s = 0; s < DW_ALOG_SLOTS - 1; ++s) col[s] = rcol[k][s];
            col[DW_ALOG_SLOTS - 1] = atq;
            col[DW_ALOG_SLOTS] = col[DW_ALOG_SLOTS - 1];
            int sl1 = sl0 + 1; if (sl1 > DW_ALOG_SLOTS) sl1 = DW_ALOG_SLOTS;
            int sl2 = sl1 + 1; if (sl2 > DW_ALOG_SLOTS) sl2 = DW_ALOG_SLOTS;
            const int src1 = sl1 > dl ? dl : DW_ALOG_SLOTS - sl1, src2 = sl2 > dl ? dl : DW_ALOG_SLOTS - sl2;
            float t1 = col[0], t2 = col[1];
            DQ_UNROLL for (int s = 1; s < DW_ALOG_SLOTS; ++s) { t1 = (s == src1) ? col[s] : t1; t2 = (s == src2) ? col[s + 1] : t2; }
            // upper body: PD to the mocap target; the second substep forms its own torque from the new state
            const float tau = d < 12 ? t1 : rkp[k] * (target - q) + rkv[k] * (-qd);
            tau2[k] = d < 12 ? t2 : target;
            if (X.lane + 64 * k < EPW * ND) DQ_SLOT(it.b, 0, it.pos) = mk4(q, qd, tau - damp * qd, ddk[k]);
        }
    }
    wave_sync();

    float nzw1[QNI];          // the second substep's encoder draws (generated with the first's)
    DQ_UNROLL for (int k = 0; k < QNI; ++k) nzw1[k] = 0.0f;
    for (int sub = 0; sub < 2; ++sub) {
        X.stamp_base = 1 + 16 * sub;
        if (!C.freeze_physics) quad_substep<TERRAIN>(L, QM, M, C.phys, X, B, sub == 0 ? push_x : 0.0f, sub == 0 ? push_y : 0.0f, sub == 1);
        wave_sync();
        // ---- integrate the joints, encoder model (tasks/dyros_dynamic_walk.py:527-530), inputs of the next substep: the
        //      slot reads and the noise of all items first, then the arithmetic, then the stores ----
        F4 fin[QNI];
        float nzw[QNI];
        DQ_UNROLL for (int k = 0; k < QNI; ++k) { const JointItem it = item(k); fin[k] = DQ_LD(it.b, 0, ipos[k]); }      // {qlo, qd, qhi, *}
        if (noise) {
            DQ_UNROLL for (int k = 0; k < QNI; ++k) { const JointItem it = item(k); nzw[k] = noise[(size_t)DW_NOISE_WORDS * it.env + DW_NZ_ENC + ND * sub + it.d]; }
        } else if (sub == 0) {          // one generator call per joint gives the draws of both substeps
            DQ_UNROLL for (int k = 0; k < QNI; ++k) {
                const JointItem it = item(k);
                dw::NoiseSrc nz;
                nz.rec = nullptr; nz.seed = C.seed; nz.env = (unsigned int)it.env; nz.step = (unsigned long long)step; nz.stream = 0;
                dw::noise_enc_pair(nz, it.d, &nzw[k], &nzw1[k]);
            }
        } else {
            DQ_UNROLL for (int k = 0; k < QNI; ++k) nzw[k] = nzw1[k];
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
                if (X.lane + 64 * k < EPW * ND) DQ_SLOT(it.b, 0, ipos[k]) = mk4(q, qd, tau - dampk[k] * qd, ddk[k]);
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
#if defined(DQ_WAVE_TIME) && defined(__HIPCC__)
    const long long dq_t1 = (long long)__builtin_readcyclecounter();
#endif
    quad_task_post<TERRAIN>(L, M, C, B, actions, noise, step, wave_index, X, qkeep, qdkeep, KP);
    DQ_STAMP(B, 41);
#if defined(DQ_WAVE_TIME) && defined(__HIPCC__)
    if (X.lane == 0) {
        B.stacked_rewards[(size_t)wave_index * EPW * DW_NUM_REW + 14] = (float)((long long)__builtin_readcyclecounter() - dq_t0);
        B.stacked_rewards[(size_t)wave_index * EPW * DW_NUM_REW + 13] = (float)(dq_t1 - dq_t0);          // physics part
    }
#endif
}

}  // namespace dwq

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
