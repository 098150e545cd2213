// dw_oct_kernels.h -- the whole VecTask.step for the 8 envs of an octet wave (dw_oct.h): pre_physics_step, the two physics
// substeps with the actuator and encoder models around them (reference tasks/dyros_dynamic_walk.py:449-541), then
// post_physics_step (dw_oct_post.h), in ONE launch: 8 lanes per env, 5 (env, joint) items per lane in the joint-parallel phases.
//
// Every fp32 expression that the reference pins bit for bit (tau per substep, qpos_noise, qvel_noise: SURVEY 8c) is
// written in the reference's operation order (fp contraction off in this region of the file); oracle/dw_task.c is the restatement the
// CPU tests compare with.
#pragma once

#include "dw_oct.h"
#include "dw_task.h"
#include "dw_oct_post.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace OCT_NS {


using dw::TaskParams;

// per-env scratch of the task phases: the four slot rows of body 0 (the base has no slot) = 16 words per env
#define OQ_ENVW(el, w) (reinterpret_cast<float *>(&L.slot[(w) >> 2][(el)])[(w) & 3])
constexpr int EW_LTP = 0, EW_MIDX = 1;       // mocap phase time, mocap row (int bits)
constexpr int EW_T0 = 2, EW_T1 = 3;          // time stamps of the two mocap rows
constexpr int EW_DL = 4, EW_SL = 5;          // torque FIFO: delay index, fill (int bits)
constexpr int WW_GATE = 15;                  // wave-wide word: perturbation gate open (env 0's scratch)
constexpr int PK_TAU2 = 0, PK_NZ1 = 64;      // words of the env's obs_buf row used as scratch during the step

// The whole VecTask.step for the wave's 8 envs: pre_physics_step up to the substep loop (action clamp and history,
// mocap phase and target, perturbation gate and schedule), the two substeps with the actuator and encoder models, and
// post_physics_step (dw_oct_post.h).  The task record is read where needed and written ONCE, by the post phase, from its
// LDS image: what the earlier phases produce for it stays in registers (StepKeep) until the image exists.
// KEEP: the one-wave-per-SIMD build (launches of at most one wave per SIMD, N <= 8192) has the whole register file: what the
// two-waves build parks in global memory between the phases of a step -- the joint state, the previous encoder reading, the second
// substep's torque input and encoder draw, damping / armature / gains -- stays in registers there (no round trip in the second epilogue).
template <bool TERRAIN, int GPUF = -1, bool KEEP = false>
DQ_HD void oct_step(OSlots &L, QHot &HW, const QuadModel &QM, const DevModel &M, const TaskParams &C, const OBuf &B,
                             const float *actions, const float *mocap, const float *noise, long long step, int wave_index) {
    if (wave_index * EPO >= C.num_envs) return;      // the second wave of the last workgroup may have no env at all (wave-uniform exit BEFORE the per-substep s_barrier: see there)
    // GPUF: the torch flavour of the post phase's norms, compiled in (dw_oct_post.h: both flavours in one kernel cost 2.1 % of the step
    // at 16384 envs -- the kernel is larger than the instruction cache it shares with a second CU).  The same was tried for the two
    // paths only tests use, an injected noise record and frozen physics: WITHOUT them the kernel is 1.6 % slower (0.1475 against
    // 0.1452 ms, A/B on one box, twice): fewer instructions, another schedule.  They stay in.
    int c_num_envs = C.num_envs, c_freeze = C.freeze_physics;          // (launch-invariant, read by every item of every phase)
    DQ_SGPR_KEEP(c_num_envs); DQ_SGPR_KEEP(c_freeze);
    DQ_STAMP(B, 54);
#if defined(DQ_WAVE_TIME) && defined(__HIPCC__)          // (timing experiment: the life of every wave, tools/wave_times.py)
    const long long dq_t0 = (long long)__builtin_readcyclecounter();
    const long long dq_r0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // @phase pre_physics
    // ==== round 1 of requests: everything whose address is known at entry -- the base state, the scalars of the record the
    //      pre-physics phase reads, the actions, the inputs of the actuator model for this lane's nine (env, joint) items,
    //      the hot tables -- in one straight run, so that the wave waits for memory once.  (Each request a lone wave waits
    //      for costs ~4 k cycles, about 1 % of the step: phase_stamps) ====
    OLane X;
    oct_lane_init(X, QM.hot, wave_index, c_num_envs, C.phys, C.friction, B);
    const int e = X.env, f = X.j & 1;
    // (record fields of my env: B.env_state[ES(field)]; no base pointer is held across the phases)
#define OQ_ES(f) oq_at(B.env_state, oq_row(DW_ES_WORDS, e), (f))
    const float r_time = OQ_ES(DW_ES_TIME), r_epi = OQ_ES(DW_ES_EPI_LEN), r_mag = OQ_ES(DW_ES_MAGNITUDE), r_phase = OQ_ES(DW_ES_PHASE);
    const float r_init = OQ_ES(DW_ES_INIT_MOCAP), r_pstart = OQ_ES(DW_ES_PERT_START), r_pon = OQ_ES(DW_ES_PERT_ON), r_pcount = OQ_ES(DW_ES_PERT_COUNT);
    const float r_imp = OQ_ES(DW_ES_IMPULSE), r_dur = OQ_ES(DW_ES_PERT_DURATION), r_ptim = OQ_ES(DW_ES_PERT_TIMING);
    const float r_dl = OQ_ES(DW_ES_DELAY_IDX), r_sl = OQ_ES(DW_ES_SIMUL_LEN);          // (integer fields travel as bit patterns)
    constexpr int NAI = (EPO * DW_NUM_ACT + 63) / 64;
    float r_act[NAI], r_head[NAI];
    DQ_UNROLL for (int k = 0; k < NAI; ++k) {
        const int i = X.lane + 64 * k, ic = i < EPO * DW_NUM_ACT ? i : 0;
        const int el = oq_div<DW_NUM_ACT>(ic), a = ic - DW_NUM_ACT * el;
        const int eg = wave_index * EPO + el, egc = eg < c_num_envs ? eg : c_num_envs - 1;
        r_act[k] = oq_at(actions, oq_row(DW_NUM_ACT, egc) + a);
        r_head[k] = oq_at(B.env_state, oq_row(DW_ES_WORDS, egc), DW_ES_HIST_HEAD);
    }
    auto item = [&](int k) {          // this lane's k-th (env, joint) item; pos is filled in below
        JointItem it;
        int i = X.lane + 64 * k;
        DQ_OPAQUE(i);                 // (recomputed at every use: five sets of derived indices held across the physics cost 20 registers)
        it.el = oq_div<ND>(i); it.d = i - ND * it.el; it.b = it.d + 1;
        const int eg = wave_index * EPO + it.el;
        it.ok = (i < EPO * ND) && (eg < c_num_envs);
        if (!(i < EPO * ND)) { it.el = 0; it.d = 0; it.b = 1; }
        it.env = eg < c_num_envs ? eg : c_num_envs - 1;
        it.pos = pcode_cell(0, 0, 0);
        return it;
    };
    // (an item's slot position: from the owner table in LDS at every use, not held across the physics)
#define OQ_IPOS(it) icode(HW, (it).el, (it).b)
    float rq[ONI], rqd[ONI], rdamp[ONI], rarm[ONI], rms[ONI], rah[ONI], rac[ONI], rkp[ONI], rkv[ONI], rcol[ONI][DW_ALOG_SLOTS - 1];
    DQ_UNROLL for (int k = 0; k < ONI; ++k) {
        const JointItem it = item(k);
        const OQ_IX g = oq_row(ND, it.env) + it.d;
        const OQ_IX ei = oq_row(DW_ES_WORDS, it.env);          // the item's task record
        const int d = it.d, dc = d < 12 ? d : 11;       // (leg-only fields: index clamped rather than a branch)
        rq[k] = oq_at(B.dof_state, g * 2, 0); rqd[k] = oq_at(B.dof_state, g * 2, 1);
        rdamp[k] = oq_at(B.dof_damping, g); rarm[k] = oq_at(B.dof_armature, g);
        rms[k] = oq_at(B.env_state, ei + dc, DW_ES_MOTOR_SCALE); rah[k] = M.action_high[dc];
        rac[k] = oq_at(actions, oq_row(DW_NUM_ACT, it.env) + dc);
        rkp[k] = M.kp[d]; rkv[k] = M.kv[d];
        DQ_UNROLL for (int s = 0; s < DW_ALOG_SLOTS - 1; ++s) rcol[k][s] = oq_at(B.env_state, ei + dc, DW_ES_ACTION_LOG + 12 * (s + 1));
    }
    stage_hot(HW, QM);
    const QHot &H = HW;
    float push_x = 0.0f, push_y = 0.0f;
    const float dt = C.phys.dt;
    // What the pre-physics phase produces for the task record goes to the record in global memory right away (the post phase
    // stages the records after the physics and finds it there): carried in registers across two substeps it is what made the
    // first form of this kernel spill.  Across the physics a lane keeps NOTHING per item: the joint angle is read back from
    // dof_state and the previous encoder reading from the record (both written after the first substep), the second substep's
    // torque input and encoder draw wait in the env's row of obs_buf, which is scratch until the post phase writes the new
    // observation into it (words PK_TAU2.., PK_NZ1..).  Every epilogue requests all of it in one go.
    StepKeep KP;
    float qkeep[ONI], qdkeep[ONI], qnprev[ONI];          // (filled by the last encoder epilogue: what the post phase takes over)
    float tau2k[ONI], n1k[ONI], dampk[ONI], armk[ONI], kpk[ONI], kvk[ONI];          // (KEEP builds only)
    float (&qvk)[ONI] = KP.qv;
    const bool wr_env = X.valid && X.prim;          // per-env scalars: quad 0 of the env writes
    // (simul_len is read by every leg item of an env: it is advanced once, at the end, by the env's lane 0)
    const int simul_len0 = f2i(r_sl);
    (void)f;
    {
        // ---- pre_physics_step, per-env scalar parts on the quad's lanes (oracle/dw_task.c step_env): lane 0 the mocap phase, lane 1 the
        //      push schedule; every fp32 expression as there ----
        dw::TaskBuffers TB;
        TB.b = B.all; TB.actions = actions; TB.noise = noise; TB.mocap = mocap; TB.step = step;
        const dw::StepCtx K = dw::make_step_ctx(C, TB, e);
        long long DW_GPTR *gate = reinterpret_cast<long long DW_GPTR *>(OQ_COLD(gate_acc));          // (K.gate is the same words through a generic pointer)
        if (X.lane == 1) OQ_ENVW(0, WW_GATE) = __builtin_bit_cast(float, dw::gate_open_at(C, K, gate));
        if (X.j == 0) {
            const float time = r_time;
            const int init_idx = f2i(r_init);
            const float local_time = dw::remainder_t(time, K.period);
            const float ltp = dw::remainder_t(local_time + (float)init_idx * K.cdt, K.period);
            const int midx = (int)(((long long)init_idx + (long long)dw::divs(C.gpu_div, local_time, K.cdt_d)) % 3599);
            OQ_ENVW(X.el, EW_LTP) = ltp;
            OQ_ENVW(X.el, EW_MIDX) = __builtin_bit_cast(float, midx);
            OQ_ENVW(X.el, EW_DL) = r_dl;
            OQ_ENVW(X.el, EW_SL) = r_sl;
            if (wr_env) {
                OQ_ES(DW_ES_MOCAP_IDX) = __builtin_bit_cast(float, midx);
                const int sl2 = simul_len0 + 2 > DW_ALOG_SLOTS ? DW_ALOG_SLOTS : simul_len0 + 2;
                OQ_ES(DW_ES_SIMUL_LEN) = __builtin_bit_cast(float, sl2);
            }
        }
        wave_sync();
        const int open = f2i(OQ_ENVW(0, WW_GATE));
        // ==== round 2 of requests: the two mocap rows of each item's env (their index is this step's arithmetic) ====
        float rt0[ONI], rt1[ONI], rm0[ONI], rm1[ONI];
        DQ_UNROLL for (int k = 0; k < ONI; ++k) {
            const JointItem it = item(k);
            const OQ_IX row0 = oq_row(DW_MOCAP_COLS, f2i(OQ_ENVW(it.el, EW_MIDX)));
            rt0[k] = oq_at(mocap, row0, 0); rt1[k] = oq_at(mocap, row0, DW_MOCAP_COLS);
            rm0[k] = oq_at(mocap, row0 + it.d, 1); rm1[k] = oq_at(mocap, row0 + it.d, DW_MOCAP_COLS + 1);
        }
        float rtf[6];
        {
            const OQ_IX row0 = oq_row(DW_MOCAP_COLS, f2i(OQ_ENVW(X.el, EW_MIDX)));
            constexpr int R1 = DW_MOCAP_COLS;
            rtf[0] = oq_at(mocap, row0, 0); rtf[1] = oq_at(mocap, row0, R1); rtf[2] = oq_at(mocap, row0, 1 + 33); rtf[3] = oq_at(mocap, row0, R1 + 1 + 33);
            rtf[4] = oq_at(mocap, row0, 1 + 34); rtf[5] = oq_at(mocap, row0, R1 + 1 + 34);
        }
        float px = 0.0f, py = 0.0f;
        if (X.j == 1) {
            // (tasks/dyros_dynamic_walk.py:438-447,489-502)
            int pert_start = f2i(r_pstart), pert_on = f2i(r_pon), pert_count = f2i(r_pcount);
            int impulse = f2i(r_imp), duration = f2i(r_dur);
            float magnitude = r_mag, phase = r_phase;
            if (open) {
                pert_start = 1;
                if (!C.force_perturb_start && X.valid && X.prim) gate[dw::GATE_LATCH] = 1;
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
            if (wr_env) {
                OQ_ES(DW_ES_PERT_START) = __builtin_bit_cast(float, pert_start); OQ_ES(DW_ES_PERT_ON) = __builtin_bit_cast(float, pert_on);
                OQ_ES(DW_ES_PERT_COUNT) = __builtin_bit_cast(float, pert_count); OQ_ES(DW_ES_IMPULSE) = __builtin_bit_cast(float, impulse);
                OQ_ES(DW_ES_PERT_DURATION) = __builtin_bit_cast(float, duration);
                OQ_ES(DW_ES_MAGNITUDE) = magnitude; OQ_ES(DW_ES_PHASE) = phase;
            }
        }
        push_x = quad_bcast<1>(px);
        push_y = quad_bcast<1>(py);
        // ---- actions: clamp, the record, the newest slot of the action ring; items (env, action) ----
        DQ_UNROLL for (int k = 0; k < NAI; ++k) {
            const int i = X.lane + 64 * k;
            const int el = oq_div<DW_NUM_ACT>(i), a = i - DW_NUM_ACT * el;
            const int eg = wave_index * EPO + el;
            float v = fminf(fmaxf(r_act[k], -1.0f), 1.0f);
            if (a == 12) v = (v > 0 ? 1.0f : 0.0f) * v;
            if (i < EPO * DW_NUM_ACT && eg < c_num_envs) {
                oq_at(B.action_history, (oq_row(DW_HIST_SLOTS, eg) + f2i(r_head[k])) * DW_NUM_ACT + a) = v;
                oq_at(B.env_state, oq_row(DW_ES_WORDS, eg) + a, DW_ES_ACTIONS) = v;
            }
        }
        if (X.j == 0 && wr_env) {
            OQ_ES(DW_ES_TARGET_FORCE) = dw::cubic_t(OQ_ENVW(X.el, EW_LTP), rtf[0], rtf[1], rtf[2], rtf[3]);
            OQ_ES(DW_ES_TARGET_FORCE + 1) = dw::cubic_t(OQ_ENVW(X.el, EW_LTP), rtf[0], rtf[1], rtf[4], rtf[5]);
        }

        // ---- actuator model, joint-parallel (items (env, dof)): inputs of both substeps.  Kept per item in
        //      registers: the joint angle (integrated after each substep), the delayed leg torque of the second substep, the
        //      encoder reading of the first, damping and gains. ----
        DQ_STAMP(B, 0);
        DQ_UNROLL for (int k = 0; k < ONI; ++k) {
            JointItem it = item(k);
            it.pos = OQ_IPOS(it);
            const int d = it.d;
            const float q = rq[k], qd = rqd[k], damp = rdamp[k], arm = rarm[k];
            // mocap target of this joint (cubic between two table rows, oracle/dw_task.c step_env) and, for the legs, the action torque
            const float target = dw::cubic_t(OQ_ENVW(it.el, EW_LTP), rt0[k], rt1[k], rm0[k], rm1[k]);
            const float atq = d < 12 ? fminf(fmaxf(rac[k], -1.0f), 1.0f) * rms[k] * rah[k] : 0.0f;
            if (it.ok) {
                const OQ_IX ei = oq_row(DW_ES_WORDS, it.env);
                oq_at(B.env_state, ei + d, DW_ES_TARGET_QPOS) = target;
                if (d < 12) oq_at(B.env_state, ei + d, DW_ES_ACTION_TORQUE) = atq;
            }
            // torque FIFO, column d (tasks/dyros_dynamic_walk.py:511-519): shift, append, pick the delayed slot -- twice, for
            // the two substeps (the action torque of the step is appended both times); the record gets the final column
            const int dl = f2i(OQ_ENVW(it.el, EW_DL)), sl0 = f2i(OQ_ENVW(it.el, EW_SL));
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
            if (KEEP) { tau2k[k] = d < 12 ? t2 : target; dampk[k] = damp; armk[k] = arm; kpk[k] = rkp[k]; kvk[k] = rkv[k]; }
            else if (it.ok) oq_at(B.obs_buf, oq_row(DW_NUM_OBS, it.env) + d, PK_TAU2) = d < 12 ? t2 : target;
            if (X.lane + 64 * k < EPO * ND) OQ_SLOT(0, 0, it.pos) = mk4(q, qd, tau - damp * qd, arm + dt * damp);
        }
    }
    wave_sync();

    for (int sub = 0; sub < 2; ++sub) {          /*@trip:2*/
        X.stamp_base = 1 + 16 * sub;
#if !defined(OCT_NO_WG_ALIGN) && defined(__HIPCC__)
        // the two waves of the workgroup meet before every substep: they share nothing but the instruction stream, and in step
        // one instruction fetch serves both (measured: -2 % step time at 16384 envs, nothing at 4096).
        // HARDWARE RULE RELIED UPON: the second wave of the LAST workgroup may have no envs (N mod 16 in 1..8) and returns at the top
        // of the kernel, before this barrier.  That is outside HIP's programming rules (every thread of a block must reach a
        // __syncthreads) and legal on gfx9 / CDNA only because s_barrier counts the waves of the workgroup that have not terminated:
        // an ended wave is not waited for (CDNA ISA guide, S_BARRIER).  Both waves also write the same bytes into the shared hot
        // tables without a barrier between them (stage_hot: identical values, so either order is the same memory).  The host
        // emulation does not see any of this; the GPU test that guards it is tests/test_hip_gpu.py::test_small_and_odd_env_counts
        // (N = 1, 3, 17, 100: N mod 16 in 1..8, the last workgroup's second wave has no envs).
        __builtin_amdgcn_s_barrier();
#endif
        if (!c_freeze) oct_substep<TERRAIN>(L, H, QM, M, C.phys, X, B, sub == 0 ? push_x : 0.0f, sub == 0 ? push_y : 0.0f, sub == 1);
        wave_sync();
        // @phase encoder_epilogue
        // ---- integrate the joints, encoder model (tasks/dyros_dynamic_walk.py:527-530), inputs of the next substep.  Four
        //      straight-line blocks -- every request, the noise, the arithmetic, every store -- with no branch between two requests:
        //      a store under `if (ok)` between two items' loads had made the compiler wait for memory once per item. ----
        JointItem its[ONI];
        OPos ips[ONI];
        OQ_IX gs[ONI];
        DQ_UNROLL for (int k = 0; k < ONI; ++k) { its[k] = item(k); ips[k] = OQ_IPOS(its[k]); gs[k] = oq_row(ND, its[k].env) + its[k].d; }
        F4 fin[ONI];
        float nzw[ONI], rdamp2[ONI], rarm2[ONI], rkp2[ONI], rkv2[ONI], pkv[ONI];
        const int pk_off = sub == 0 ? PK_TAU2 : PK_NZ1;          // obs_buf scratch: the second substep's torque input / its encoder draw
        DQ_UNROLL for (int k = 0; k < ONI; ++k) fin[k] = OQ_LD(0, 0, ips[k]);      // {qlo, qd, qhi, *}
        if (KEEP) {
            // (the joint state of the first epilogue is still in qkeep / qdkeep / qnprev; the first epilogue itself reads the state the
            //  step started from and the previous step's encoder reading)
            if (sub == 0) {
                DQ_UNROLL for (int k = 0; k < ONI; ++k) {
                    qkeep[k] = oq_at(B.dof_state, gs[k] * 2, 0); qdkeep[k] = oq_at(B.dof_state, gs[k] * 2, 1);
                    qnprev[k] = oq_at(B.env_state, oq_row(DW_ES_WORDS, its[k].env) + its[k].d, DW_ES_QPOS_PRE);
                }
            }
            DQ_UNROLL for (int k = 0; k < ONI; ++k) { pkv[k] = sub == 0 ? tau2k[k] : n1k[k]; rdamp2[k] = dampk[k]; rarm2[k] = armk[k]; rkp2[k] = kpk[k]; rkv2[k] = kvk[k]; }
        } else {
        DQ_UNROLL for (int k = 0; k < ONI; ++k) {
            qkeep[k] = oq_at(B.dof_state, gs[k] * 2, 0); qdkeep[k] = oq_at(B.dof_state, gs[k] * 2, 1);
            qnprev[k] = oq_at(B.env_state, oq_row(DW_ES_WORDS, its[k].env) + its[k].d, DW_ES_QPOS_PRE);
            pkv[k] = oq_at(B.obs_buf, oq_row(DW_NUM_OBS, its[k].env) + its[k].d + pk_off);
            // damping, armature and the PD gains of the upper body (used after the first substep only; requested in both so that the
            // block has no branch): again from memory rather than held in 20 registers through the first substep
            rdamp2[k] = oq_at(B.dof_damping, gs[k]); rarm2[k] = oq_at(B.dof_armature, gs[k]); rkp2[k] = M.kp[its[k].d]; rkv2[k] = M.kv[its[k].d];
        }
        }
        if (noise) { DQ_UNROLL for (int k = 0; k < ONI; ++k) nzw[k] = oq_at(noise, oq_row(DW_NOISE_WORDS, its[k].env) + ND * sub + its[k].d, DW_NZ_ENC); }
        static_assert(ONI == 5 || ONI == 3, "the grouped touch below names the loads");
        OQ_KEEP3(fin[0], fin[1], fin[2]);
        if constexpr (ONI == 5) OQ_KEEP2(fin[3], fin[4]);
        if (sub == 0) DQ_STAMP(B, 34);
        float n1[ONI];
        DQ_UNROLL for (int k = 0; k < ONI; ++k) n1[k] = 0.0f;
        if (!noise) {
            if (sub == 0) {          // one generator call per joint gives the draws of both substeps
                DQ_UNROLL for (int k = 0; k < ONI; ++k) {
                    dw::NoiseSrc nz;
                    nz.rec = nullptr; nz.seed = C.seed; nz.env = (unsigned int)its[k].env; nz.step = (unsigned long long)step; nz.stream = 0;
                    dw::noise_enc_pair(nz, its[k].d, &nzw[k], &n1[k]);
                }
            } else {
                DQ_UNROLL for (int k = 0; k < ONI; ++k) nzw[k] = pkv[k];
            }
        }
        if (sub == 0) DQ_STAMP(B, 35);
        float qo[ONI], qdo[ONI], qno[ONI];
        F4 nxt[ONI];
        DQ_UNROLL for (int k = 0; k < ONI; ++k) {
            const int d = its[k].d;
            float q = qkeep[k], qd = qdkeep[k];
            if (!c_freeze) {
                qd = fin[k].y; q = qkeep[k] + dt * qd;
                if (q < fin[k].x) { q = fin[k].x; if (qd < 0) qd = 0; }
                if (q > fin[k].z) { q = fin[k].z; if (qd > 0) qd = 0; }
            }
            const float qn = q + fminf(fmaxf(nzw[k], -0.00016f), 0.00016f);
            const float qv = C.gpu_div ? (qn - qnprev[k]) * C.inv_dt_f : (qn - qnprev[k]) / dt;
            qo[k] = q; qdo[k] = qd; qno[k] = qn;
            qkeep[k] = q; qdkeep[k] = qd; qnprev[k] = qn; qvk[k] = qv;
            const float tau = d < 12 ? pkv[k] : rkp2[k] * (pkv[k] - q) + rkv2[k] * (-qd);
            nxt[k] = mk4(q, qd, tau - rdamp2[k] * qd, rarm2[k] + dt * rdamp2[k]);
        }
        // stores: the state (dof_state always holds the current one), after the first substep also the encoder reading, the second
        // substep's encoder draw and the slot inputs of the second substep
        DQ_UNROLL for (int k = 0; k < ONI; ++k) {
            if (its[k].ok) {
                if (KEEP) {          // (dof_state is written once, with the final state; nothing is parked)
                    if (!c_freeze && sub == 1) { oq_at(B.dof_state, gs[k] * 2, 0) = qo[k]; oq_at(B.dof_state, gs[k] * 2, 1) = qdo[k]; }
                } else {
                    if (!c_freeze) { oq_at(B.dof_state, gs[k] * 2, 0) = qo[k]; oq_at(B.dof_state, gs[k] * 2, 1) = qdo[k]; }
                    if (sub == 0) {
                        oq_at(B.env_state, oq_row(DW_ES_WORDS, its[k].env) + its[k].d, DW_ES_QPOS_PRE) = qno[k];
                        if (!noise) oq_at(B.obs_buf, oq_row(DW_NUM_OBS, its[k].env) + its[k].d, PK_NZ1) = n1[k];
                    }
                }
            }
            if (KEEP && sub == 0) n1k[k] = n1[k];
            if (sub == 0 && !c_freeze && X.lane + 64 * k < EPO * ND) OQ_SLOT(0, 0, ips[k]) = nxt[k];
        }
        wave_sync();
        DQ_STAMP(B, 1 + 16 * sub + 14);
    }
    // @phase post_entry
    if (X.valid && !c_freeze && X.o == 0 && X.prim) {
        int e2 = e;
        DQ_OPAQUE(e2);            // (the row's address again from the index: held since the loads at the top it is a register pair through both substeps)
        DQ_UNROLL for (int i = 0; i < 13; ++i) oq_at(B.root_states, oq_row(13, e2), i) = X.root[i];
    }
    DQ_UNROLL for (int k = 0; k < ONI; ++k) KP.qn[k] = qnprev[k];
    DQ_STAMP(B, 40);
    {
        // every lane of the quad may have seen a non-sole body in contact: one flag per env
        float c = X.coll ? 1.0f : 0.0f;
        c += quad_xor1(c);
        c += quad_xor2(c);
        c += oct_xor4(c);
        if (LPE == 16) c += hex_xor8(c);
        X.coll = c > 0.0f;
    }
    wave_sync();
#if defined(DQ_WAVE_TIME) && defined(__HIPCC__)
    const long long dq_t1 = (long long)__builtin_readcyclecounter();
#endif
    oct_task_post<TERRAIN, GPUF>(L, M, C, B, actions, noise, step, wave_index, X, qkeep, qdkeep, KP);
    DQ_STAMP(B, 41);
#if defined(DQ_WAVE_TIME) && defined(__HIPCC__)
    if (X.lane == 0) {
        OQ_COLD(stacked_rewards)[(OQ_IX)wave_index * EPO * DW_NUM_REW + 14] = (float)((long long)__builtin_readcyclecounter() - dq_t0);
        OQ_COLD(stacked_rewards)[(OQ_IX)wave_index * EPO * DW_NUM_REW + 13] = (float)(dq_t1 - dq_t0);          // physics part
        // where the wave ran: HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID, as exact small integers in two floats of env 2's row
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        OQ_COLD(stacked_rewards)[((OQ_IX)wave_index * EPO + 2) * DW_NUM_REW + 1] = (float)(hw & 0xffff);
        OQ_COLD(stacked_rewards)[((OQ_IX)wave_index * EPO + 2) * DW_NUM_REW + 2] = (float)(xcc & 0xf);
        OQ_COLD(stacked_rewards)[((OQ_IX)wave_index * EPO + 2) * DW_NUM_REW + 3] = (float)(dq_r0 & 0xffffff);      // start time, 100 MHz clock common to the chip (low bits)
        OQ_COLD(stacked_rewards)[((OQ_IX)wave_index * EPO + 2) * DW_NUM_REW + 4] = (float)((long long)__builtin_amdgcn_s_memrealtime() & 0xffffff);      // end time
    }
#endif
}

}  // namespace OCT_NS

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
