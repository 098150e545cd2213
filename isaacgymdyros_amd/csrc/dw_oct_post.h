// dw_oct_post.h -- post_physics_step of DyrosDynamicWalk for the 8 envs of an octet wave (dw_oct.h), run at the end of the
// fused step kernel (dw_oct_kernels.h) when the physics is done and the wave's 17 KB of body slots are free.  The same
// fp32 expressions in the reference's order -- regions Q1..Q6 and the reset block, as oracle/dw_task.c states them (fp contraction off) -- so
// the reference goldens hold bit for bit; the lanes are mapped for 8 envs per wave: per-env scalar work on octet lanes
// 0..3 (one group of reward terms each), per-word work over ITEMS (env, index) = lane + 64 k.
// Reference: tasks/dyros_dynamic_walk.py:543-563 (post_physics_step), :581-596 (check_termination), :802-947 (reward),
// :598-669,720-748 (reset_idx), :750-796 (observations); vec_task.py:519-733 (dof-property randomisation).
#pragma once

#include "dw_oct.h"
#include "dw_task.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace OCT_NS {

#if defined(DQ_WAVE_TIME) && defined(__HIPCC__)      // (timing experiment, tools/wave_times.py: the post phases of EVERY wave, left in the reward rows of its second env)
#define DQ_WT_DECL long long dq_wt[14]; int dq_wn = 0
#define DQ_WT() dq_wt[dq_wn++] = (long long)__builtin_readcyclecounter()
#define DQ_WT_FLUSH(B, wave_index) do { if (lane == 0) for (int i_ = 1; i_ < dq_wn; ++i_) (B).cold->stacked_rewards[((OQ_IX)(wave_index) * EPO + 1) * DW_NUM_REW + i_] = (float)(dq_wt[i_] - dq_wt[i_ - 1]); } while (0)
#else
#define DQ_WT_DECL do { } while (0)
#define DQ_WT() do { } while (0)
#define DQ_WT_FLUSH(B, wave_index) do { } while (0)
#endif

using dw::TaskParams;

// flat LDS layout of the post phase (float words of OSlots::slot)
constexpr int PL_ES_STRIDE = DW_ES_WORDS;                // = the global layout: the 8 records move as one 11.9 KB run of 16-byte pieces
constexpr int PL_ES = 0;                                 // [8][372] task records
constexpr int PL_Q = EPO * DW_ES_WORDS;                  // [8][33][2] joint state
constexpr int PL_ROOT = PL_Q + EPO * ND * 2;             // [8][13]
constexpr int PL_NORMED = PL_ROOT + EPO * 13 + 16;       // [8][37] normalised observation of this step
constexpr int PL_PS = PL_NORMED + EPO * DW_NUM_OBS1;     // [8][32] per-env scratch
constexpr int PL_OBN = PL_PS + EPO * 32;                 // [2][37] observation mean, divisor (the hot tables belong to both waves of the workgroup)
static_assert((PL_Q * 4) % 16 == 0, "post layout: the record block must end on a 16-byte boundary");
constexpr int PL_RC = PL_OBN + 2 * DW_NUM_OBS1;          // [33][2] per-joint constants of the reset: initial angle, the same clamped to the joint range
static_assert(PL_RC + 2 * ND <= (NB * 4 + XROWS) * EPO * 4, "post layout does not fit the slot area");
// per-env scratch words
constexpr int PS_RTERM = 0;      // [16] reward terms, [14] = |orientation error|
constexpr int PS_BAD = 16, PS_COLL = 17, PS_RESET = 18, PS_PROGRESS = 19, PS_RANDOMIZE = 20, PS_MASS = 21;
constexpr int PS_FOOT = 22;      // [2][3] net contact force on the two sole bodies
constexpr int PS_ORG = 28;       // [3] new tile origin (terrain curriculum)

#define PQ_LF(i) LF[(i)]
#define PQ_ES(el, off) LF[PL_ES + (el) * PL_ES_STRIDE + (off)]
#define PQ_ESI(el, off) (*reinterpret_cast<int *>(&LF[PL_ES + (el) * PL_ES_STRIDE + (off)]))
#define PQ_Q(el, d) LF[PL_Q + ((el) * ND + (d)) * 2]
#define PQ_QD(el, d) LF[PL_Q + ((el) * ND + (d)) * 2 + 1]
#define PQ_ROOT(el, i) LF[PL_ROOT + (el) * 13 + (i)]
#define PQ_NORMED(el, k) LF[PL_NORMED + (el) * DW_NUM_OBS1 + (k)]
#define PQ_PS(el, w) LF[PL_PS + (el) * 32 + (w)]
#define PQ_PSI(el, w) (*reinterpret_cast<int *>(&LF[PL_PS + (el) * 32 + (w)]))

// What the substeps produce for the task record, kept in registers until the record's LDS image exists.  (What the
// pre-physics phase produces -- mocap row and targets, push schedule, clamped actions, action torques -- is already in the
// records in global memory when they are staged: dw_oct_kernels.h.)
struct StepKeep {
    float qn[ONI], qv[ONI];                                            // items: encoder angle and rate after the second substep
};

// Rows of NW consecutive floats at any 4-byte alignment, moved in 16-byte pieces (global_load/store_dwordx4 take dword-aligned
// addresses): a 37-word history row is 10 requests instead of 37.
struct __attribute__((packed, aligned(4))) U4 { float x, y, z, w; };
template <int NW> DQ_HD void ld_row(const float *p, float (&v)[NW]) {
    DQ_UNROLL for (int i = 0; i + 4 <= NW; i += 4) { const U4 t = *reinterpret_cast<const U4 *>(p + i); v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w; }
    DQ_UNROLL for (int i = NW - NW % 4; i < NW; ++i) v[i] = p[i];
}
template <int NW> DQ_HD void st_row(float *p, const float (&v)[NW]) {
    DQ_UNROLL for (int i = 0; i + 4 <= NW; i += 4) { U4 t; t.x = v[i]; t.y = v[i + 1]; t.z = v[i + 2]; t.w = v[i + 3]; *reinterpret_cast<U4 *>(p + i) = t; }
    DQ_UNROLL for (int i = NW - NW % 4; i < NW; ++i) p[i] = v[i];
}

// torch.norm of 3 elements in torch's CPU or GPU summation order (dw_task.h norm_sel)
DQ_HD float norm3_t(int gpu, float x, float y, float z) {
    const float v[3] = {x, y, z};
    return dw::norm_sel_v<3>(gpu, v);
}

// torch.norm of N = 33 or 12 elements f(0..N-1), summed by the EIGHT lanes of an octet together in exactly torch's order (dw_task.h
// norm_g / norm_fn): every lane of the octet calls it (o = its octet lane) and gets the result.  One lane alone walks 33 dependent
// additions per norm -- three such norms and two of 12 elements were 600 of the post phase's instructions and its longest chains.
//   GPU order: T = 32 (8) threads square x[t] (+ x[t + T]) and combine in a balanced tree over t: lane o takes the leaves 4 o .. 4 o + 3
//   (N = 33) or leaf o (N = 12), the upper tree levels are the exchanges l ^ 1, l ^ 2, l ^ 4 (a + b == b + a bit for bit).
//   CPU order: eight fused accumulators over x[l], x[l + 8], ... -- lane l's -- added in lane order, then the tail as the scalar code.
template <int N, class F>
DQ_HD float oct_norm(int gpu, int o, F f) {
    static_assert(N == 33 || N == 12, "oct_norm: the row lengths of the reward's long norms");
    if (gpu) {
        float r;
        if constexpr (N == 33) {
            float v[4];
            DQ_UNROLL for (int i = 0; i < 4; ++i) { const float x = f(4 * o + i); v[i] = x * x; }
            { const float y = f(32), yy = y * y; v[0] = o == 0 ? v[0] + yy : v[0]; }
            r = (v[0] + v[1]) + (v[2] + v[3]);
        } else {
            const float x = f(o), y = f(o < 4 ? o + 8 : o), yy = y * y;
            r = x * x;
            r = o < 4 ? r + yy : r;
        }
        r = r + quad_xor1(r);
        r = r + quad_xor2(r);
        r = r + oct_xor4(r);
        return sqrtf(r);
    }
    float acc = 0.0f;
    DQ_UNROLL for (int d = 0; d + 8 <= N; d += 8) { const float x = f(d + o); acc = fmaf(x, x, acc); }
    float b0 = oct_fetch(acc, 0);
    DQ_UNROLL for (int l = 1; l < 8; ++l) b0 = b0 + oct_fetch(acc, l);
    if constexpr (N == 12) { DQ_UNROLL for (int l = 0; l < 4; ++l) { const float x = f(8 + l); const float p2 = x * x; b0 = b0 + p2; } }
    else { const float x = f(32); b0 = fmaf(x, x, b0); }
    return sqrtf(b0);
}

// Inputs from the physics part of the kernel: qv/qdv = the item lanes' new joint state (items as joint_item()), X.root,
// X.coll / X.footT (collision flag of my bodies, net force on my sole body).  With physics frozen (tests) the state and
// the contact forces are the Gym tensors as they are.
// GPUF: the torch flavour of norm()'s summation order as a compile-time constant (1 GPU, 0 CPU; -1 = read DwConfig.torch_gpu_div at run
// time, the host emulation's form): carrying both orders in one kernel cost 2.1 % of the step at 16384 envs (instruction cache)
template <bool TERRAIN, int GPUF = -1>
DQ_HD void oct_task_post(OSlots &L, const DevModel &M, const TaskParams &C, const OBuf &B, const float *actions,
                          const float *noise, long long step, int wave_index, OLane &X, const float (&qv)[ONI], const float (&qdv)[ONI],
                          const StepKeep &KP) {
    const int gnorm = GPUF < 0 ? C.gpu_div : GPUF;
    float *LF = reinterpret_cast<float *>(&L.slot[0][0]);
    DQ_WT_DECL; DQ_WT();
    // (the lane id again, opaque to the optimiser: index arithmetic shared with the pre-physics phase would otherwise be kept in
    //  registers across the two substeps)
    int lane = X.lane;
    DQ_OPAQUE(lane);
    const int j = lane & (LPE - 1), el = lane / LPE;        // j: lane of the env (0 .. LPE - 1); lanes 4.. idle in the per-env scalar groups
    const int eg_ = wave_index * EPO + el;
    const bool xvalid = eg_ < C.num_envs;
    const int e = xvalid ? eg_ : C.num_envs - 1;
    const int N = C.num_envs;
    dw::TaskBuffers TB;
    TB.b = B.all; TB.actions = actions; TB.noise = noise; TB.mocap = nullptr; TB.step = step;
    const dw::StepCtx K = dw::make_step_ctx(C, TB, e);        // (nz of MY env; items build their own)
    const float period = K.period;
    const double cdt_d = K.cdt_d;
    const int LFG = M.left_foot_gym, RFG = M.right_foot_gym;

    // @phase post_stage
    // (the VecTask counters and the clock action of Q1, the per-joint constants of the reset path and the observation's mean /
    //  scale, requested together with the records: one memory latency for all.  The hot tables of the physics are dead by now:
    //  the observation constants go where they were.)
    const long long q1_progress = oq_at(OQ_COLD(progress_buf), (OQ_IX)e), q1_randomize = oq_at(OQ_COLD(randomize_buf), (OQ_IX)e);
    const float q1_mass = oq_at(OQ_COLD(total_mass), (OQ_IX)e), q1_clock = dw::clamp_action(actions, e, 12);
    const int lj = lane < ND ? lane : 0, lo1 = lane < DW_NUM_OBS1 ? lane : 0;
    const float c_qinit = M.q_init[lj], c_qhi = M.qhi[lj], c_qlo = M.qlo[lj];
    const float c_org0 = oq_at(OQ_COLD(env_origins), 3 * (OQ_IX)e, 0), c_org1 = oq_at(OQ_COLD(env_origins), 3 * (OQ_IX)e, 1), c_org2 = oq_at(OQ_COLD(env_origins), 3 * (OQ_IX)e, 2);
    float *OBN = LF + PL_OBN;                                   // [2][37] mean, divisor
    const float c_om = M.obs_mean[lo1], c_od = M.obs_inv_std_den[lo1];            // (stored below, after the records' requests)
    // (nominal damping / armature of the joints whose randomisation words this lane draws at a reset: requested here, with everything else)
    constexpr int DR_B0 = DW_NZ_DR_DAMP / 4, DR_NBLK = DW_NZ_DR_FRIC / 4 - DR_B0 + 1;
    float dr_nom[4];          // (lane 8 + b draws block DR_B0 + b at a reset)
    DQ_UNROLL for (int i = 0; i < 4; ++i) {
        const int bl = lane >= 8 && lane < 8 + DR_NBLK ? lane - 8 : 0;
        const int w = 4 * (DR_B0 + bl) + i;
        const bool isd = w >= DW_NZ_DR_DAMP && w < DW_NZ_DR_ARM, isa = w >= DW_NZ_DR_ARM && w < DW_NZ_DR_FRIC;
        const int l = isd ? w - DW_NZ_DR_DAMP : (isa ? w - DW_NZ_DR_ARM : 0);
        dr_nom[i] = (isa ? M.arm_nom : M.damp_nom)[l];
    }
    // ---- stage: joint state, base state, contact summary, the 16 task records ----
    DQ_UNROLL for (int k = 0; k < ONI; ++k) {
        const int i = lane + 64 * k;
        if (i < EPO * ND) { LF[PL_Q + 2 * i] = qv[k]; LF[PL_Q + 2 * i + 1] = qdv[k]; }
    }
    if (j == 0) {
        DQ_UNROLL for (int i = 0; i < 13; ++i) PQ_ROOT(el, i) = X.root[i];
        PQ_PSI(el, PS_BAD) = 0; PQ_PSI(el, PS_COLL) = 0; PQ_PSI(el, PS_RESET) = 0;
        PQ_PS(el, PS_ORG) = c_org0; PQ_PS(el, PS_ORG + 1) = c_org1; PQ_PS(el, PS_ORG + 2) = c_org2;
    }
    {
        // 16 records = 1488 pieces of 16 bytes, contiguous in HBM and in LDS: every lane requests its 24 pieces before the
        // first one is stored (one memory latency for the lot)
        static_assert((EPO * DW_ES_WORDS) % 4 == 0 && (DW_ES_WORDS * 4) % 16 == 0, "record block must be a whole number of 16-byte pieces");
        constexpr int NP = EPO * DW_ES_WORDS / 4, PER = (NP + 63) / 64;
        const int nvalid = N - wave_index * EPO;                          // envs of this wave that exist (>= 1)
        const int np_ok = (nvalid >= EPO ? EPO : nvalid) * (DW_ES_WORDS / 4);
        wave_sync_global();           // (the pre-physics phase of this wave wrote fields of these records)
        const F4 *src = reinterpret_cast<const F4 *>(B.env_state + (OQ_IX)wave_index * EPO * DW_ES_WORDS);
        F4 *dst = reinterpret_cast<F4 *>(LF + PL_ES);
        constexpr int GRP = PER;
        DQ_UNROLL for (int g8 = 0; g8 < PER; g8 += GRP) {
            float tx[GRP], ty[GRP], tz[GRP], tw[GRP];
            DQ_UNROLL for (int u = 0; u < GRP; ++u) {
                const int pi = lane + 64 * (g8 + u);
                // (pieces of envs past the end mirror the last env's record; nothing of theirs is stored)
                const int ps = pi < np_ok ? pi : (np_ok - (DW_ES_WORDS / 4)) + (pi - (DW_ES_WORDS / 4) * oq_div<DW_ES_WORDS / 4>(pi));
                const F4 v = src[pi < NP ? ps : 0];
                tx[u] = v.x; ty[u] = v.y; tz[u] = v.z; tw[u] = v.w;
            }
            DQ_UNROLL for (int u = 0; u < GRP; ++u) { const int pi = lane + 64 * (g8 + u); if (pi < NP) dst[pi] = mk4(tx[u], ty[u], tz[u], tw[u]); }
        }
    }
    if (lane < DW_NUM_OBS1) { OBN[lane] = c_om; OBN[DW_NUM_OBS1 + lane] = c_od; }
    if (lane < ND) { LF[PL_RC + 2 * lane] = c_qinit; LF[PL_RC + 2 * lane + 1] = fmaxf(fminf(c_qinit, c_qhi), c_qlo); }
    wave_sync();
    // ---- the record fields this step has produced so far (oracle/dw_task.c step_env..P3), from the lanes that hold them ----
    DQ_UNROLL for (int k = 0; k < ONI; ++k) {
        const int i = lane + 64 * k;
        if (i < EPO * ND) {
            const int ee = oq_div<ND>(i), d = i - ND * ee;
            const int egr = wave_index * EPO + ee, eg = egr < N ? egr : N - 1;
            PQ_ES(ee, DW_ES_QPOS_NOISE + d) = KP.qn[k];
            PQ_ES(ee, DW_ES_QPOS_PRE + d) = KP.qn[k];
            PQ_ES(ee, DW_ES_QVEL_NOISE + d) = KP.qv[k];
            if (d < 12) {
                // action torque of the step, appended to the torque FIFO by both substeps (oracle/dw_task.c step_env)
                const float at = PQ_ES(ee, DW_ES_ACTION_TORQUE + d);
                float col[DW_ALOG_SLOTS];
                DQ_UNROLL for (int s2 = 0; s2 < DW_ALOG_SLOTS; ++s2) col[s2] = s2 + 2 < DW_ALOG_SLOTS ? PQ_ES(ee, DW_ES_ACTION_LOG + 12 * (s2 + 2) + d) : at;
                DQ_UNROLL for (int s2 = 0; s2 < DW_ALOG_SLOTS; ++s2) PQ_ES(ee, DW_ES_ACTION_LOG + 12 * s2 + d) = col[s2];
            }
        }
    }
    wave_sync();
    if (C.freeze_physics) {          /*@prob:0*/
        // debug mode: simulate() was the identity, so the net contact forces are an input (dw_task.h step_env)
        if (j == 0) {
            const float *cf = B.contact_forces + (OQ_IX)DW_NUM_BODIES * 3 * e;
            DQ_UNROLL for (int i = 0; i < 3; ++i) { PQ_PS(el, PS_FOOT + i) = cf[3 * LFG + i]; PQ_PS(el, PS_FOOT + 3 + i) = cf[3 * RFG + i]; }
        }
        for (int i = lane; i < EPO * DW_NUM_BODIES; i += 64) {
            const int ee = i / DW_NUM_BODIES, g = i - DW_NUM_BODIES * ee;
            const int eg = wave_index * EPO + ee < N ? wave_index * EPO + ee : N - 1;
            const float *cf = B.contact_forces + ((OQ_IX)DW_NUM_BODIES * eg + g) * 3;
            if (g != LFG && g != RFG && norm3_t(gnorm, cf[0], cf[1], cf[2]) > 1.0f) PQ_PSI(ee, PS_COLL) = 1;
        }
    } else {
        if (X.coll) PQ_PSI(el, PS_COLL) = 1;
        if (j < 2) { DQ_UNROLL for (int i = 0; i < 3; ++i) PQ_PS(el, PS_FOOT + 3 * j + i) = X.footT[i]; }
    }
    wave_sync();

    DQ_STAMP(B, 42);
    // @phase post_q1
    // ---- Q1: clocks, VecTask counters, non-finite guard ----
    if (j == 0) {
        const long long p = q1_progress, rbl = q1_randomize;
        int rb = (int)(rbl > 0x7ffffffe ? 0x7ffffffe : rbl);
        PQ_PS(el, PS_MASS) = q1_mass;
        PQ_ES(el, DW_ES_EPI_LEN) += 1.0f;
        float time = PQ_ES(el, DW_ES_TIME);
        time = time + C.dt_policy_f;
        time = time + C.clock_gain_f * q1_clock;
        PQ_ES(el, DW_ES_TIME) = time;
        if (xvalid) {
            oq_at(OQ_COLD(timeout_buf), (OQ_IX)e) = ((float)(p + (C.timeout_fix ? 1 : 0)) >= C.max_episode_length - 1.0f) ? 1 : 0;
            oq_at(OQ_COLD(progress_buf), (OQ_IX)e) = p + 1;
        }
        PQ_PSI(el, PS_PROGRESS) = (int)(p + 1);
        rb = rb + 1;
        if (xvalid) oq_at(OQ_COLD(randomize_buf), (OQ_IX)e) = rb;
        PQ_PSI(el, PS_RANDOMIZE) = rb;
        bool bad = false;
        DQ_UNROLL for (int i = 0; i < 13; ++i) bad = bad || !dw::finitef(PQ_ROOT(el, i));
        if (bad) PQ_PSI(el, PS_BAD) = 1;
    }
    DQ_UNROLL for (int k = 0; k < ONI; ++k) {
        const int i = lane + 64 * k;
        if (i < EPO * ND && (!dw::finitef(LF[PL_Q + 2 * i]) || !dw::finitef(LF[PL_Q + 2 * i + 1]))) PQ_PSI(oq_div<ND>(i), PS_BAD) = 1;
    }
    wave_sync();
    if (wave_any(PQ_PSI(el, PS_BAD) != 0)) {          /*@prob:0*/
        if (j == 0 && PQ_PSI(el, PS_BAD)) {
            DQ_UNROLL for (int i = 0; i < 13; ++i) PQ_ROOT(el, i) = (i == 2) ? C.initial_height : (i == 6 ? 1.0f : 0.0f);
            PQ_ESI(el, DW_ES_NAN_RESETS) += 1;
            PQ_PSI(el, PS_COLL) = 0;
            DQ_UNROLL for (int i = 0; i < 6; ++i) PQ_PS(el, PS_FOOT + i) = 0.0f;
        }
        DQ_UNROLL for (int k = 0; k < ONI; ++k) {
            const int i = lane + 64 * k;
            if (i < EPO * ND && PQ_PSI(i / ND, PS_BAD)) { LF[PL_Q + 2 * i] = 0.0f; LF[PL_Q + 2 * i + 1] = 0.0f; }
        }
        for (int i = lane; i < EPO * DW_NUM_BODIES * 3; i += 64) {
            const int ee = i / (DW_NUM_BODIES * 3), w = i - DW_NUM_BODIES * 3 * ee, eg = wave_index * EPO + ee;
            if (eg < N && PQ_PSI(ee, PS_BAD)) B.contact_forces[(OQ_IX)DW_NUM_BODIES * 3 * eg + w] = 0.0f;
        }
        wave_sync();
    }

    DQ_STAMP(B, 43);
    // @phase post_reward
    // ---- Q2: reward terms, one group per lane of the quad ----
    {
        // the three 33-element norms and the two 12-element ones: the octet's eight lanes together (oct_norm above), in torch's CPU or
        // GPU order (dw_task.h norm_sel)
        const float n_q = oct_norm<33>(gnorm, j & 7, [&](int jj) { return PQ_ES(el, DW_ES_TARGET_QPOS + jj) - PQ_Q(el, jj); });
        const float n_qd = oct_norm<33>(gnorm, j & 7, [&](int jj) { return 0.0f - PQ_QD(el, jj); });
        const float n_qa = oct_norm<33>(gnorm, j & 7, [&](int jj) { return PQ_QD(el, jj) - PQ_ES(el, DW_ES_PRE_QVEL + jj); });
        const float n_a = oct_norm<12>(gnorm, j & 7, [&](int i) { return PQ_ES(el, DW_ES_ACTIONS + i) * 333.0f; });
        const float n_da = oct_norm<12>(gnorm, j & 7, [&](int i) { return (PQ_ES(el, DW_ES_ACTIONS + i) - PQ_ES(el, DW_ES_ACTIONS_PRE + i)) * 333.0f; });
        if (j == 0) {
            const float qq[4] = {PQ_ROOT(el, 3), PQ_ROOT(el, 4), PQ_ROOT(el, 5), PQ_ROOT(el, 6)};
            const float aerr = fabsf(dw::quat_err(qq, gnorm));
            PQ_PS(el, PS_RTERM + 14) = aerr;
            PQ_PS(el, PS_RTERM + 0) = 0.3f * expf(-13.2f * aerr);
            const float dv[2] = {PQ_ES(el, DW_ES_TARGET_VEL) - PQ_ROOT(el, 7), PQ_ES(el, DW_ES_TARGET_VEL + 1) - PQ_ROOT(el, 8)};
            const float n = dw::norm_sel_v<2>(gnorm, dv);
            PQ_PS(el, PS_RTERM + 6) = 0.3f * expf(-3.0f * (n * n));
        }
        if (j == 1) {
            PQ_PS(el, PS_RTERM + 1) = 0.35f * expf(-2.0f * (n_q * n_q));
            PQ_PS(el, PS_RTERM + 4) = 0.05f * expf(-0.01f * n_a);
        }
        if (j == 2) {
            PQ_PS(el, PS_RTERM + 2) = 0.05f * expf(-0.01f * (n_qd * n_qd));
            PQ_PS(el, PS_RTERM + 7) = 0.05f * expf(-20.0f * (n_qa * n_qa));
        }
        if (j == 3) {
            PQ_PS(el, PS_RTERM + 5) = 0.6f * expf((-0.01f * 1.0f) * n_da);
            const float lf[3] = {PQ_PS(el, PS_FOOT), PQ_PS(el, PS_FOOT + 1), PQ_PS(el, PS_FOOT + 2)};
            const float rf[3] = {PQ_PS(el, PS_FOOT + 3), PQ_PS(el, PS_FOOT + 4), PQ_PS(el, PS_FOOT + 5)};
            const float lfp[3] = {PQ_ES(el, DW_ES_FOOT_FORCE_PRE), PQ_ES(el, DW_ES_FOOT_FORCE_PRE + 1), PQ_ES(el, DW_ES_FOOT_FORCE_PRE + 2)};
            const float rfp[3] = {PQ_ES(el, DW_ES_FOOT_FORCE_PRE + 3), PQ_ES(el, DW_ES_FOOT_FORCE_PRE + 4), PQ_ES(el, DW_ES_FOOT_FORCE_PRE + 5)};
            float dl[3], dr[3];
            DQ_UNROLL for (int i = 0; i < 3; ++i) { dl[i] = lf[i] - lfp[i]; dr[i] = rf[i] - rfp[i]; }
            PQ_PS(el, PS_RTERM + 9) = 0.2f * expf((-0.01f * 1.0f) * (dw::norm_sel_v<3>(gnorm, dl) + dw::norm_sel_v<3>(gnorm, dr)));
            const bool lcon = lf[2] > 1.0f, rcon = rf[2] > 1.0f;
            const int idx = PQ_ESI(el, DW_ES_MOCAP_IDX);
            const bool DSP = (3300 <= idx && idx < 3600) || (idx < 300) || (1500 <= idx && idx < 2100);
            const bool RSSP = 300 <= idx && idx < 1500;
            const bool LSSP = 2100 <= idx && idx < 3300;
            float fcr = 0.0f;
            if (DSP && rcon && lcon) fcr = 0.2f;
            if (RSSP && rcon && !lcon) fcr = 0.2f;
            if (LSSP && !rcon && lcon) fcr = 0.2f;
            PQ_PS(el, PS_RTERM + 8) = fcr;
            PQ_ES(el, DW_ES_CRS) = PQ_ES(el, DW_ES_CRS) + fcr;
            PQ_PS(el, PS_RTERM + 10) = 0.0f;
            const float tm = PQ_PS(el, PS_MASS);
            const float thr = (float)(1.4 * 9.81) * tm;
            const bool th = (lf[2] > thr) || (rf[2] > thr);
            PQ_PS(el, PS_RTERM + 11) = th ? -0.2f * 1.0f : 0.0f;
            const float cl = fmaxf(lf[2] - thr, 0.0f), cr = fmaxf(rf[2] - thr, 0.0f);
            const float pen = 0.1f * expf(-0.007f * (dw::norm_sel_v<1>(gnorm, &cl) + dw::norm_sel_v<1>(gnorm, &cr)));
            PQ_PS(el, PS_RTERM + 3) = th ? pen : 0.1f * 1.0f;
            const float thd = ((float)(0.2 * 9.81) * tm) / 1.0f;
            const bool dd = (fabsf(lf[2] - lfp[2]) > thd) || (fabsf(rf[2] - rfp[2]) > thd);
            PQ_PS(el, PS_RTERM + 12) = dd ? -0.05f * 1.0f : 0.0f;
            const float ws = dw::divs(C.gpu_div, tm, 104.48);
            const float tl = 0.1f * expf(-0.001f * fabsf(lf[2] + ws * PQ_ES(el, DW_ES_TARGET_FORCE)));
            const float tr = 0.1f * expf(-0.001f * fabsf(rf[2] + ws * PQ_ES(el, DW_ES_TARGET_FORCE + 1)));
            PQ_PS(el, PS_RTERM + 13) = tl + tr;
        }
    }
    wave_sync();

    DQ_STAMP(B, 44);
    // @phase post_q3
    // ---- Q3: total reward, termination ----
    {
        const bool collision = PQ_PSI(el, PS_COLL) != 0;
        const float aerr = PQ_PS(el, PS_RTERM + 14);
        // stacked_rewards: 15 words per env, the quad writes them (lane j takes 4 j .. 4 j + 3)
        if (xvalid) {
            DQ_UNROLL for (int i = 0; i < 4; ++i) {
                const int l = 4 * j + i;
                if (l < 14) oq_at(OQ_COLD(stacked_rewards), oq_row(DW_NUM_REW, e) + l) = collision ? 1.0f * C.death_cost : PQ_PS(el, PS_RTERM + l);
                if (l == 14) oq_at(OQ_COLD(stacked_rewards), oq_row(DW_NUM_REW, e), 14) = PQ_ESI(el, DW_ES_PERT_START) ? 1.0f : 0.0f;
            }
        }
        if (j == 0) {
            const float *r = &PQ_PS(el, PS_RTERM);
            float total = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + r[8] + r[9] + r[10] + r[11] + r[12] + r[13];
            if (collision) total = 1.0f * C.death_cost;
            if (aerr > 0.5f) total = 1.0f * C.death_cost;
            int reset = aerr > 0.5f ? 1 : 0;
            if ((float)PQ_PSI(el, PS_PROGRESS) >= C.max_episode_length - 1.0f) reset = 1;
            if (collision) reset = 1;
            if (PQ_PSI(el, PS_BAD)) reset = 1;
            if (xvalid) { oq_at(OQ_COLD(rew_buf), (OQ_IX)e) = total; oq_at(OQ_COLD(reset_buf), (OQ_IX)e) = reset; }
            PQ_PSI(el, PS_RESET) = reset;
            float ret = PQ_ES(el, DW_ES_EPI_RETURN) + total;
            if (reset) {
                PQ_ES(el, DW_ES_LAST_RETURN) = ret;
                PQ_ESI(el, DW_ES_EPISODES) += 1;
                ret = 0.0f;
            }
            PQ_ES(el, DW_ES_EPI_RETURN) = ret;
        }
    }
    if (C.terrain_curriculum && C.terrain_lvl_acc) {
        // the curriculum's logging columns (tasks/dyros_dynamic_walk.py:417-421, formed in compute_reward, BEFORE this step's reset_idx
        // moves any level): level sum and env count per terrain type, read by dw_terrain_log.  A wave's envs are neighbours and nearly
        // always of one type: their levels are summed bit plane by bit plane with ballots and lane 0 adds (count << 32 | sum) to the
        // wave's bucket with one atomic (16 384 atomics on 20 words cost 67 us per step; 2 048 spread over 16 buckets cost nothing);
        // an env of another type than the wave's first adds itself.
        static_assert(dw::LVL_BUCKETS == 16, "bucket = wave index & 15");
        unsigned long long DW_GPTR *acc = (unsigned long long DW_GPTR *)C.terrain_lvl_acc;
        const int types = C.terrain_num_types, row = types * dw::LVL_BUCKETS;
        const bool own = j == 0 && xvalid;
        long long ty = own ? OQ_COLD(terrain_types)[e] : 0, ty0 = OQ_COLD(terrain_types)[wave_index * EPO];
        const int lvl = own ? (int)OQ_COLD(terrain_levels)[e] : 0;
        ty = ty < 0 ? 0 : (ty > types - 1 ? types - 1 : ty);
        ty0 = ty0 < 0 ? 0 : (ty0 > types - 1 ? types - 1 : ty0);
        const bool same = own && ty == ty0;
        unsigned long long w = (unsigned long long)__builtin_popcountll(wave_ballot(same)) << 32;
        DQ_UNROLL for (int b = 0; b < dw::LVL_BITS; ++b) w += (unsigned long long)__builtin_popcountll(wave_ballot(same && ((lvl >> b) & 1))) << b;
        const int bk = wave_index & (dw::LVL_BUCKETS - 1);
        if (lane == 0) atomic_add_u64(&acc[K.slot_cur * row + bk * types + (int)ty0], w);
        if (own && !same) atomic_add_u64(&acc[K.slot_cur * row + bk * types + (int)ty], ((unsigned long long)1 << 32) | (unsigned long long)(unsigned int)lvl);
        if (wave_index == 0) {
            for (int i = lane; i < row; i += 64) acc[K.slot_next * row + i] = 0;
            if (lane == 0) acc[3 * row] = (unsigned long long)K.slot_cur;
        }
    }
    wave_sync();

    DQ_STAMP(B, 45); DQ_WT();
    // @phase post_reset
    // ---- reset_idx for the envs that ended (dw_task.h reset_region) ----
    const bool any_reset = wave_any(PQ_PSI(el, PS_RESET) != 0);
    if (any_reset) {          /*@prob:0.18*/
        const bool mine = PQ_PSI(el, PS_RESET) != 0;
        if (C.terrain_curriculum && j == 0 && mine) {
            const float d[2] = {PQ_ROOT(el, 0) - c_org0, PQ_ROOT(el, 1) - c_org1};
            const float distance = dw::norm_sel_v<2>(gnorm, d);
            const bool move_up = distance > C.terrain_half_length;
            const float tv[2] = {PQ_ES(el, DW_ES_TARGET_VEL), PQ_ES(el, DW_ES_TARGET_VEL + 1)};
            const float need = dw::norm_sel_v<2>(gnorm, tv) * C.max_episode_length_s * 0.5f;
            const bool move_down = (distance < need) && !move_up;
            long long lvl = OQ_COLD(terrain_levels)[e] + ((move_up ? 1 : 0) - (move_down ? 1 : 0));
            if (lvl >= C.terrain_num_levels) {
                int k = (int)(dw::noise_word(K.nz, DW_NZ_TERRAIN_LVL) * (float)C.terrain_num_levels);
                if (k > C.terrain_num_levels - 1) k = C.terrain_num_levels - 1;
                lvl = k;
            } else if (lvl < 0) lvl = 0;
            long long ty = OQ_COLD(terrain_types)[e];
            ty = ty < 0 ? 0 : (ty > C.terrain_num_types - 1 ? C.terrain_num_types - 1 : ty);
            const float DW_GPTR *org = OQ_COLD(terrain_origins) + ((OQ_IX)lvl * C.terrain_num_types + ty) * 3;
            DQ_UNROLL for (int i = 0; i < 3; ++i) { const float o = org[i]; PQ_PS(el, PS_ORG + i) = o; if (xvalid) OQ_COLD(env_origins)[3 * e + i] = o; }
            if (xvalid) OQ_COLD(terrain_levels)[e] = lvl;
        }
        wave_sync();
        // ONE ended env at a time, on all 64 lanes (round 5).  Until then every ended env worked on its own eight lanes: 4 generator
        // blocks per lane in a row (the 32 reset words = 8 blocks, the 67 randomisation words = 18), five passes over the joints, nine
        // over the torque FIFO and the action ring -- and since a wave nearly always has ONE ended env (tools/wave_times.py: 6 555 of
        // 7 156 resetting waves), 56 lanes watched 8 work: 13 k cycles that the whole launch then waits for, because some SIMD holds two
        // such waves in every step.  Now lane L draws block L of the env (0 .. 7 the reset words, 8 .. 25 the randomisation words): one
        // generator call deep, one pass over the joints, two over the FIFO and the ring; k ended envs take k turns.
        static_assert(DW_NZ_QPOS_BIAS % 4 == 0 && DW_NZ_PTIMING < DW_NZ_QPOS_BIAS + 32, "reset words: eight blocks per env");
        static_assert(DW_NZ_DR_ARM == DW_NZ_DR_DAMP + DW_NUM_DOF && DW_NZ_DR_FRIC == DW_NZ_DR_ARM + DW_NUM_DOF, "DR words are contiguous");
        static_assert(8 + DR_NBLK <= 64 && ND <= 64, "one generator block and one joint per lane");
        DQ_WT();
        int *LI = reinterpret_cast<int *>(LF);
        unsigned long long todo = wave_ballot(mine && j == 0);          // bit LPE * el: env el of this wave ended
        while (todo != 0ull) {
            const int er = (int)(__builtin_ctzll(todo) / LPE);          // (wave-uniform)
            todo &= todo - 1ull;
            const int egr = wave_index * EPO + er, eg = egr < N ? egr : N - 1;
            const bool rvalid = egr < N;
            dw::NoiseSrc nz = K.nz;
            nz.rec = noise ? noise + (size_t)DW_NOISE_WORDS * eg : nullptr; nz.env = (unsigned int)eg;
            const bool do_dr = (C.dr_dof || C.dr_friction) && PQ_PSI(er, PS_RANDOMIZE) >= 1;
            const int esb = PL_ES + er * PL_ES_STRIDE;
            const int dummy = PL_PS + er * 32 + 31;          // (a store that does not apply goes to the env's spare scratch word: branch-free)
            // ---- the draws: lanes 0 .. 7 the reset words, lanes 8 .. 8 + DR_NBLK - 1 the dof-property / friction words (vec_task.py:519-733;
            //      `randomize` is on in the yaml, so every reset passes there) ----
            const bool l_reset = lane < 8, l_dr = lane >= 8 && lane < 8 + DR_NBLK && do_dr;
            float u[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (l_reset || l_dr) dw::noise_block(nz, l_reset ? DW_NZ_QPOS_BIAS / 4 + lane : DR_B0 + (lane - 8), u);
            if (l_reset) {
                const bool gd = C.gpu_div != 0;
                DQ_UNROLL for (int i = 0; i < 4; ++i) {
                    const int w = DW_NZ_QPOS_BIAS + 4 * lane + i;
                    const bool isq = w < DW_NZ_QUAT_BIAS, isb = !isq && w < DW_NZ_TARGET_VEL, ist = w == DW_NZ_TARGET_VEL;
                    const bool ism = w >= DW_NZ_MOTOR && w < DW_NZ_DELAY;
                    const float x = u[i] * (w < DW_NZ_TARGET_VEL ? 6.28f : (ist ? 0.8f : 0.4f));
                    // divs(gpu_div, x, 100.0 | 150.0) - (float)(3.14 / 100 | 150)
                    const float dv = isq ? 100.0f : 150.0f, rv = isq ? (float)(1.0 / 100.0) : (float)(1.0 / 150.0);
                    const float cv = isq ? (float)(3.14 / 100) : (float)(3.14 / 150);
                    const float yd = (gd ? x * rv : x / dv) - cv;
                    const float y = (isq || isb) ? yd : (ist ? x * 1.0f : x + 0.8f);
                    int k4 = (int)(u[i] * 4.0f), kt = (int)(u[i] * 2000.0f);
                    k4 = k4 > 3 ? 3 : k4; kt = kt > 1999 ? 1999 : kt;
                    const int bits = w == DW_NZ_INIT_MOCAP ? (u[i] > 0.5f ? 0 : 1800) : (w == DW_NZ_DELAY ? 2 + k4 : (w == DW_NZ_PTIMING ? kt : f2i(y)));
                    const int dst = isq ? DW_ES_QPOS_BIAS + (w - DW_NZ_QPOS_BIAS) : isb ? DW_ES_QUAT_BIAS + (w - DW_NZ_QUAT_BIAS)
                                  : ist ? DW_ES_TARGET_VEL : w == DW_NZ_INIT_MOCAP ? DW_ES_INIT_MOCAP : ism ? DW_ES_MOTOR_SCALE + (w - DW_NZ_MOTOR)
                                  : w == DW_NZ_DELAY ? DW_ES_DELAY_IDX : w == DW_NZ_PTIMING ? DW_ES_PERT_TIMING : -1;
                    LI[dst >= 0 ? esb + dst : dummy] = bits;
                    if (ist) LF[esb + DW_ES_TARGET_VEL + 1] = x * 0.0f;
                }
            }
            if (l_dr) {
                const float d0 = C.dr_damp[0], d1 = C.dr_damp[1] - C.dr_damp[0], a0 = C.dr_arm[0], a1 = C.dr_arm[1] - C.dr_arm[0];
                const float f0 = C.dr_fric[0], f1 = C.dr_fric[1] - C.dr_fric[0];
                DQ_UNROLL for (int i = 0; i < 4; ++i) {
                    const int w = 4 * (DR_B0 + (lane - 8)) + i;
                    const bool isd = w >= DW_NZ_DR_DAMP && w < DW_NZ_DR_ARM, isa = w >= DW_NZ_DR_ARM && w < DW_NZ_DR_FRIC;
                    const int l = isd ? w - DW_NZ_DR_DAMP : (isa ? w - DW_NZ_DR_ARM : 0);
                    const float sd = d0 + u[i] * d1, sa = a0 + u[i] * a1;
                    if (rvalid && C.dr_dof && (isd || isa)) oq_at(isa ? B.dof_armature : B.dof_damping, oq_row(ND, eg) + l) = isa ? dr_nom[i] * sa : dr_nom[i] + sd;
                    if (rvalid && C.dr_friction && w == DW_NZ_DR_FRIC) oq_at(OQ_COLD(friction_scale), (OQ_IX)eg) = f0 + u[i] * f1;
                }
            }
            DQ_WT();
            // ---- the joints (lane = joint), the sole forces, the root state (lanes 16 .. 28) ----
            if (lane < ND) {
                const int l = lane;
                const float qi = LF[PL_RC + 2 * l], qc = LF[PL_RC + 2 * l + 1];
                LF[esb + DW_ES_QPOS_NOISE + l] = qi;
                LF[esb + DW_ES_QPOS_PRE + l] = qi;
                LF[esb + DW_ES_QVEL_NOISE + l] = 0.0f;
                LF[esb + DW_ES_PRE_QVEL + l] = 0.0f;
                LF[PL_Q + (er * ND + l) * 2] = qc;
                LF[PL_Q + (er * ND + l) * 2 + 1] = 0.0f;
                if (l < 12) LF[esb + DW_ES_ACTION_TORQUE_PRE + l] = 0.0f;
                if (l < 24) LF[esb + DW_ES_WARM + l] = 0.0f;
                if (l < 6) LF[esb + DW_ES_FOOT_FORCE_PRE + l] = PQ_PS(er, PS_FOOT + l);
                if (l >= 16 && l < 29) {
                    const int ii = l - 16;
                    float v = ii == 2 ? C.initial_height : (ii == 6 ? 1.0f : 0.0f);
                    v += ii < 3 ? PQ_PS(er, PS_ORG + (ii < 3 ? ii : 0)) : 0.0f;          // (the env's origin; the curriculum has put the new one there)
                    if (C.custom_origins && ii < 2) v += 2.0f * dw::noise_word(nz, DW_NZ_ROOT_JITTER + ii) + (-1.0f);
                    LF[PL_ROOT + er * 13 + ii] = v;
                }
            }
            // torque FIFO and action ring, zeroed
            static_assert((DW_HIST_SLOTS * DW_NUM_ACT) % 4 == 0, "action ring of an env: whole 16-byte pieces");
            constexpr int NAL = DW_ALOG_SLOTS * 12, NAH = DW_HIST_SLOTS * DW_NUM_ACT / 4;
            DQ_UNROLL for (int i = 0; i < (NAL + 63) / 64; ++i) { if (lane + 64 * i < NAL) PQ_ES(er, DW_ES_ACTION_LOG + lane + 64 * i) = 0.0f; }
            if (rvalid) {
                F4 *ah = reinterpret_cast<F4 *>(&oq_at(B.action_history, oq_row(DW_HIST_SLOTS * DW_NUM_ACT, eg)));
                DQ_UNROLL for (int i = 0; i < (NAH + 63) / 64; ++i) { if (lane + 64 * i < NAH) ah[lane + 64 * i] = mk4(0.0f, 0.0f, 0.0f, 0.0f); }
            }
            DQ_WT();
            // per-env scalars (dw_task.h reset_region, lane 40)
            if (lane == 40) {
                if (do_dr && rvalid) oq_at(OQ_COLD(randomize_buf), (OQ_IX)eg) = 0;
                PQ_ES(er, DW_ES_TIME) = 0.0f;
                if (rvalid) { oq_at(OQ_COLD(progress_buf), (OQ_IX)eg) = 0; oq_at(OQ_COLD(reset_buf), (OQ_IX)eg) = 1; }
                PQ_ES(er, DW_ES_CRM) = PQ_ES(er, DW_ES_CRS) / PQ_ES(er, DW_ES_EPI_LEN);
                PQ_ES(er, DW_ES_CRS) = 0.0f;
                PQ_ESI(er, DW_ES_SIMUL_LEN) = 0;
                PQ_ES(er, DW_ES_EPI_LEN_LOG) = PQ_ES(er, DW_ES_EPI_LEN);
                PQ_ES(er, DW_ES_EPI_LEN) = 0.0f;
                PQ_ESI(er, DW_ES_PERT_COUNT) = 0;
                PQ_ESI(er, DW_ES_PERT_ON) = 0;
            }
        }
        wave_sync();
    }

    DQ_STAMP(B, 46); DQ_WT();
    // @phase post_taps
    // ---- Q5, first half: request the history taps now, use them after Q4 (one memory latency, spent computing the new
    //      observation).  A lane takes ROWS (env, tap): 37 observation words and 13 action words, consecutive in the rings and
    //      in obs_buf, so a row is 10 + 4 requests with constant offsets and no per-word index arithmetic.  The newest
    //      observation tap is this step's own (Q4, still in LDS) and is not read back. ----
    constexpr int NTAP = DW_NUM_HIS - 1, NPAIR = EPO * NTAP, RPL = (NPAIR + 63) / 64;
    static_assert((DW_NUM_SKIP * DW_NUM_HIS) % DW_HIST_SLOTS == 0, "the last observation tap must be the newest slot");
    float tapo[RPL][DW_NUM_OBS1], tapa[RPL][DW_NUM_ACT];
    DQ_UNROLL for (int r = 0; r < RPL; ++r) {
        const int p = lane + 64 * r, pc = p < NPAIR ? p : 0;
        const int ee = oq_div<NTAP>(pc), tap = pc - NTAP * ee;
        const int egr = wave_index * EPO + ee, eg = egr < N ? egr : N - 1;
        // (ring slots: the head lies in [0, DW_HIST_SLOTS), the sums below in [0, 2 DW_HIST_SLOTS]: a compare and subtract, not a division)
        auto wrap = [](int x) { return x >= 2 * DW_HIST_SLOTS ? x - 2 * DW_HIST_SLOTS : (x >= DW_HIST_SLOTS ? x - DW_HIST_SLOTS : x); };
        const int head = wrap(PQ_ESI(ee, DW_ES_HIST_HEAD) + 1);
        const int so = wrap(head + DW_NUM_SKIP * (tap + 1) - 1), sa = wrap(head + DW_NUM_SKIP * (tap + 1));
        ld_row(&oq_at(B.obs_history, (oq_row(DW_HIST_SLOTS, eg) + so) * DW_NUM_OBS1), tapo[r]);
        ld_row(&oq_at(B.action_history, (oq_row(DW_HIST_SLOTS, eg) + sa) * DW_NUM_ACT), tapa[r]);
    }
    // @phase post_obs
    // ---- Q4: 37-d observation, normalisation, newest history slot.  Items (env, entry), grouped by kind so that each of the
    //      expensive functions (atan2, sincos, the noise draw) is executed by one or two wave passes, not by all ten ----
    {
        auto finish = [&](int ee, int l, float o) {
            const int egr = wave_index * EPO + ee;
            const float nrm = (o - OBN[l]) / OBN[DW_NUM_OBS1 + l];
            PQ_NORMED(ee, l) = nrm;
            // (the newest slot; an env that was just reset shows this observation in EVERY slot, tasks/dyros_dynamic_walk.py:655-669: its
            //  ring is filled below, by the whole wave)
            if (egr < N && PQ_ES(ee, DW_ES_EPI_LEN) != 0.0f) oq_at(B.obs_history, oq_row(DW_HIST_SLOTS * DW_NUM_OBS1, egr) + PQ_ESI(ee, DW_ES_HIST_HEAD) * DW_NUM_OBS1 + l) = nrm;
        };
        // joint angles / rates of the legs with their biases, target velocity: 26 plain entries per env
        for (int i = lane; i < EPO * 26; i += 64) {
            const int ee = oq_div<26>(i), t = i - 26 * ee;
            const int l = t < 24 ? 3 + t : 29 + (t - 24);
            float o;
            if (l < 15) o = PQ_ES(ee, DW_ES_QPOS_NOISE + (l - 3)) + PQ_ES(ee, DW_ES_QPOS_BIAS + (l - 3));
            else if (l < 27) o = PQ_ES(ee, DW_ES_QVEL_NOISE + (l - 15));
            else o = PQ_ES(ee, DW_ES_TARGET_VEL + (l - 29));
            finish(ee, l, o);
        }
        // Euler angles of the base (quat2euler / mat2euler, python/isaacgym/torch_utils.py:227-273): 3 per env
        if (lane < EPO * 3) {
            const int ee = oq_div<3>(lane), l = lane - 3 * ee;
            const float x = PQ_ROOT(ee, 3), y = PQ_ROOT(ee, 4), z = PQ_ROOT(ee, 5), w = PQ_ROOT(ee, 6);
            const float m00 = w * w + x * x - y * y - z * z;
            const float m01 = 2 * x * y - 2 * w * z;
            const float m10 = 2 * x * y + 2 * w * z;
            const float m11 = w * w - x * x + y * y - z * z;
            const float m20 = 2 * x * z - 2 * w * y;
            const float m21 = 2 * y * z + 2 * w * x;
            const float m22 = w * w - x * x - y * y + z * z;
            const float cy = sqrtf(m00 * m00 + m10 * m10);
            const bool cond = cy > (float)(2.220446049250313e-16 * 4);
            const float num = l == 0 ? m21 : (l == 1 ? -m20 : (cond ? m10 : -m01));
            const float den = l == 0 ? m22 : (l == 1 ? cy : (cond ? m00 : m11));
            float o = atan2f(num, den);
            if (l == 0 && !cond) o = 0.0f;
            o = o + PQ_ES(ee, DW_ES_QUAT_BIAS + l);
            finish(ee, l, o);
        }
        // gait phase as sin / cos: 2 per env
        if (lane < EPO * 2) {
            const int ee = lane >> 1, l = 27 + (lane & 1);
            const float time2idx = dw::divs(C.gpu_div, dw::remainder_t(PQ_ES(ee, DW_ES_TIME), period), cdt_d);
            const float phase = dw::divs(C.gpu_div, dw::remainder_t((float)PQ_ESI(ee, DW_ES_INIT_MOCAP) + time2idx, 3599.0f), 3599.0);
            const float ang = (float)(2 * 3.14159265358979) * phase;
            float sn, cs;
            sincosf(ang, &sn, &cs);
            finish(ee, l, l == 27 ? sn : cs);
        }
        // base velocity with its noise draw: 6 per env
        for (int i = lane; i < EPO * 6; i += 64) {
            const int ee = oq_div<6>(i), l = 31 + (i - 6 * ee);
            const int egr = wave_index * EPO + ee, eg = egr < N ? egr : N - 1;
            dw::NoiseSrc nz = K.nz;
            nz.rec = noise ? noise + (OQ_IX)DW_NOISE_WORDS * eg : nullptr; nz.env = (unsigned int)eg;
            finish(ee, l, PQ_ROOT(ee, 7 + (l - 31)) + (dw::noise_word(nz, DW_NZ_VEL + (l - 31)) * 0.05f - 0.025f));
        }
    }
    wave_sync();
    {
        // the observation ring of the envs that were reset: 20 slots x 37 words = 185 sixteen-byte pieces per env, every word the new
        // normalised observation's -- one env at a time on all 64 lanes (three stores per lane; as per-item stores this was 20 stores in
        // each of the seven item groups of Q4 for the whole wave)
        static_assert((DW_HIST_SLOTS * DW_NUM_OBS1) % 4 == 0, "observation ring of an env: whole 16-byte pieces");
        constexpr int NOH = DW_HIST_SLOTS * DW_NUM_OBS1 / 4;
        unsigned long long todo = wave_ballot(j == 0 && PQ_ES(el, DW_ES_EPI_LEN) == 0.0f);
        while (todo != 0ull) {
            const int er = (int)(__builtin_ctzll(todo) / LPE);
            todo &= todo - 1ull;
            const int egr = wave_index * EPO + er;
            if (egr < N) {
                F4 *oh = reinterpret_cast<F4 *>(&oq_at(B.obs_history, oq_row(DW_HIST_SLOTS * DW_NUM_OBS1, egr)));
                DQ_UNROLL for (int i = 0; i < (NOH + 63) / 64; ++i) {
                    const int pc = lane + 64 * i;
                    if (pc < NOH) {
                        const int w0 = 4 * pc;
                        oh[pc] = mk4(PQ_NORMED(er, w0 % DW_NUM_OBS1), PQ_NORMED(er, (w0 + 1) % DW_NUM_OBS1), PQ_NORMED(er, (w0 + 2) % DW_NUM_OBS1), PQ_NORMED(er, (w0 + 3) % DW_NUM_OBS1));
                    }
                }
            }
        }
    }

    DQ_STAMP(B, 47); DQ_WT();
    // @phase post_obsbuf
    // ---- Q5, second half: the 487-d observation buffer.  Rows requested above go out as they came (an env that was just
    //      reset shows its first observation in every tap and zeros in the action taps, tasks/dyros_dynamic_walk.py:655-669);
    //      the newest tap is copied from LDS, items (env, word). ----
    {
        float *ob = B.obs_buf + (OQ_IX)wave_index * EPO * DW_NUM_OBS;
        DQ_UNROLL for (int r = 0; r < RPL; ++r) {
            const int p = lane + 64 * r, pc = p < NPAIR ? p : 0;
            const int ee = oq_div<NTAP>(pc), tap = pc - NTAP * ee;
            const bool ok = p < NPAIR && wave_index * EPO + ee < N;
            const bool fill = PQ_ES(ee, DW_ES_EPI_LEN) == 0.0f, rs = PQ_PSI(ee, PS_RESET) != 0;
            if (wave_any(fill)) {          /*@prob:0.18*/
                if (fill) { DQ_UNROLL for (int k = 0; k < DW_NUM_OBS1; ++k) tapo[r][k] = PQ_NORMED(ee, k); }
            }
            const int newest = PQ_ESI(ee, DW_ES_HIST_HEAD);
            const bool own = ((newest + 1) % DW_HIST_SLOTS + DW_NUM_SKIP * (tap + 1)) % DW_HIST_SLOTS == newest;    // (never, with 2 x 10 slots)
            if (wave_any(rs || own)) {          /*@prob:0.18*/
                if (rs) { DQ_UNROLL for (int k = 0; k < DW_NUM_ACT; ++k) tapa[r][k] = 0.0f; }
                else if (own) { DQ_UNROLL for (int k = 0; k < DW_NUM_ACT; ++k) tapa[r][k] = PQ_ES(ee, DW_ES_ACTIONS + k); }
            }
            if (ok) {
                st_row(ob + ee * DW_NUM_OBS + tap * DW_NUM_OBS1, tapo[r]);
                st_row(ob + ee * DW_NUM_OBS + DW_NUM_OBS1 * DW_NUM_HIS + tap * DW_NUM_ACT, tapa[r]);
            }
        }
        DQ_UNROLL for (int u = 0; u < (EPO * DW_NUM_OBS1 + 63) / 64; ++u) {
            const int i = lane + 64 * u;
            if (i < EPO * DW_NUM_OBS1) {
                const int ee = oq_div<DW_NUM_OBS1>(i), k = i - DW_NUM_OBS1 * ee;
                if (wave_index * EPO + ee < N) ob[ee * DW_NUM_OBS + NTAP * DW_NUM_OBS1 + k] = PQ_NORMED(ee, k);
            }
        }
    }

    DQ_STAMP(B, 48); DQ_WT();
    // @phase post_q6
    // ---- Q6: late updates (tasks/dyros_dynamic_walk.py:560-563), ring head, gate statistics ----
    DQ_UNROLL for (int k = 0; k < ONI; ++k) {
        const int i = lane + 64 * k;
        if (i < EPO * ND) {
            const int ee = oq_div<ND>(i), l = i - ND * ee;
            PQ_ES(ee, DW_ES_PRE_QVEL + l) = PQ_QD(ee, l);
            if (l < 12) PQ_ES(ee, DW_ES_ACTION_TORQUE_PRE + l) = PQ_ES(ee, DW_ES_ACTION_TORQUE + l);
            if (l < DW_NUM_ACT) PQ_ES(ee, DW_ES_ACTIONS_PRE + l) = PQ_ES(ee, DW_ES_ACTIONS + l);
            if (l >= 20 && l < 26) PQ_ES(ee, DW_ES_FOOT_FORCE_PRE + (l - 20)) = PQ_PS(ee, PS_FOOT + (l - 20));
        }
    }
    wave_sync();
    if (j == 0) {
        PQ_ESI(el, DW_ES_HIST_HEAD) = (PQ_ESI(el, DW_ES_HIST_HEAD) + 1) % DW_HIST_SLOTS;
        if (C.perturb && !C.force_perturb_start && xvalid) {
            const float eln = PQ_ES(el, DW_ES_EPI_LEN_LOG), cm = PQ_ES(el, DW_ES_CRM);
            const int bk = e % dw::GATE_BUCKETS;
            long long de, dc = 0;
            if (dw::finitef(eln) && dw::finitef(cm)) { de = (long long)eln; dc = (long long)llrintf(cm * 4294967296.0f); }
            else de = -((long long)1 << 62);
            long long DW_GPTR *gate = reinterpret_cast<long long DW_GPTR *>(OQ_COLD(gate_acc));
            atomic_add_u64(reinterpret_cast<unsigned long long DW_GPTR *>(&gate[(K.slot_cur * dw::GATE_BUCKETS + bk) * 2]), (unsigned long long)de);
            atomic_add_u64(reinterpret_cast<unsigned long long DW_GPTR *>(&gate[(K.slot_cur * dw::GATE_BUCKETS + bk) * 2 + 1]), (unsigned long long)dc);
            gate[(K.slot_next * dw::GATE_BUCKETS + bk) * 2] = 0;
            gate[(K.slot_next * dw::GATE_BUCKETS + bk) * 2 + 1] = 0;
        }
    }
    wave_sync();

    DQ_STAMP(B, 49); DQ_WT();
    // @phase post_writeback
    // ---- write back: the records (contiguous), and the Gym state of the envs whose state the task changed ----
    {
        constexpr int NP = EPO * DW_ES_WORDS / 4, PER = (NP + 63) / 64;
        const int nvalid = N - wave_index * EPO;
        const int np_ok = (nvalid >= EPO ? EPO : nvalid) * (DW_ES_WORDS / 4);
        F4 *dstg = reinterpret_cast<F4 *>(B.env_state + (OQ_IX)wave_index * EPO * DW_ES_WORDS);
        const F4 *srcl = reinterpret_cast<const F4 *>(LF + PL_ES);
        DQ_UNROLL for (int u = 0; u < PER; ++u) { const int pi = lane + 64 * u; if (pi < np_ok) dstg[pi] = srcl[pi]; }
        const bool changed = PQ_PSI(el, PS_RESET) != 0 || PQ_PSI(el, PS_BAD) != 0;
        if (wave_any(changed)) {          /*@prob:0.18*/
            if (j == 0 && changed && xvalid) { DQ_UNROLL for (int i = 0; i < 13; ++i) oq_at(B.root_states, oq_row(13, e), i) = PQ_ROOT(el, i); }
            DQ_UNROLL for (int k = 0; k < ONI; ++k) {
                const int i = lane + 64 * k;
                if (i < EPO * ND) {
                    const int ee = oq_div<ND>(i), eg = wave_index * EPO + ee;
                    if (eg < N && (PQ_PSI(ee, PS_RESET) || PQ_PSI(ee, PS_BAD))) {
                        B.dof_state[((OQ_IX)ND * wave_index * EPO) * 2 + 2 * i] = LF[PL_Q + 2 * i];
                        B.dof_state[((OQ_IX)ND * wave_index * EPO) * 2 + 2 * i + 1] = LF[PL_Q + 2 * i + 1];
                    }
                }
            }
        }
    }
    DQ_STAMP(B, 50); DQ_WT();
    DQ_WT_FLUSH(B, wave_index);
}

#undef PQ_LF
#undef PQ_ES
#undef PQ_ESI
#undef PQ_Q
#undef PQ_QD
#undef PQ_ROOT
#undef PQ_NORMED
#undef PQ_PS
#undef PQ_PSI

}  // namespace OCT_NS

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
