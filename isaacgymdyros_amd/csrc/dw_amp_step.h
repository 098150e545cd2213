// dw_amp_step.h -- the fused TocabiAMPLower step and reset (include/dyros_walk.h: dw_amp_step_begin / _mid / _end, dw_amp_reset_rows,
// dw_amp_reset_done).  Written as REGIONS: every thread runs a region to its end before any thread starts the next one; regions
// exchange data through LDS and through the envs' rows in global memory only.  Two shapes:
//   * the step kernels: a workgroup of 256 threads per 16 envs (EnvGroup), work as items over all threads, the serial per-env
//     functions with lane = env; a region ends with a workgroup barrier;
//   * the reset kernels: one wavefront per env (EnvWave), lanes over the env's rows; a region ends with a fence (a wave's memory
//     operations are ordered; four such waves share a workgroup and never meet).
// The host emulation (tests/emul/, g++) runs a region as a loop over its threads -- which is why no value crosses a region
// boundary in a local variable and why a word that several threads read is only written in a LATER region.
//
// Every expression is the torch class' (isaacgymdyros_amd/tocabi_amp_lower.py, itself pinned to the reference class by replay:
// tasks/amp/tocabi_amp_lower_base.py:238-305 reset_idx, :540-580 history stacking, :642-748 pre-physics, :750-804 post-physics;
// tasks/tocabi_amp_lower.py:88-96,144-147,258-272), in its operation order, with fp contraction off: with the caller's draws the
// fused step gives the torch implementation's bits (tests/test_amp_gpu.py).
//
// Random numbers: the caller's (torch's, in the torch implementation's order -- the pinned form) or, with DwAmpConfig.device_draws,
// drawn here: Philox4x32-10 keyed by DwAmpConfig.seed, counter = (word, env, the env's draw counter DwAmpBuffers.draw_ctr, stream).
// The draw counter lives in device memory and is advanced by the kernels, so a step recorded in a hipGraph draws fresh numbers at
// every replay.
//
// Histories: DwAmpConfig.hist_ring = 0 keeps the reference's layout (newest slot last, every step shifts the whole row: 960 words
// read and written per env and step); = 1 keeps both histories as rings (DwAmpBuffers.hist_head: the physical slot of the OLDEST
// entry), a step writes 48 words and the stacked observation gathers through the head.  Logical slot i is physical slot
// (head + i) mod (num_his * num_skip) in both (head = 0 without the ring).
#pragma once

#include "dw_amp.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#if defined(__HIPCC__)
#define DWA_UNROLL _Pragma("unroll")
// a region's body is a lambda that captures the kernel's by-value argument structs by reference: left to the inliner's size heuristics it
// stays a function of its own once its loops are unrolled, and the structs -- 600 B of kernel arguments -- are then copied to scratch and
// every field read from there (the one-launch kernel: 1 140 B of scratch, 153 stores at its entry).  Regions are always inlined.
#define DWA_INL __attribute__((always_inline))
#else
#define DWA_UNROLL
#define DWA_INL
#endif

namespace dwa {

#if defined(__HIPCC__)
struct EnvWave {
    template <class F> DW_HD void par(F &&f) const {
        f((int)(threadIdx.x & 63u));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};
#else
struct EnvWave {
    template <class F> void par(F &&f) const { for (int l = 0; l < 64; ++l) f(l); }
};
#endif

constexpr int AW = DW_AMP_DISC_BASE + 6;          // discriminator observation of one step (two key bodies)
constexpr int HIST_STAGE = 64 * 12;               // words of a history row a wave stages (amp_args_ok holds the configuration to it)

struct StepLds {
    LegModel LM;
    float root[13], ds[DW_NUM_DOF * 2], cf[DW_NUM_BODIES * 3], obs[DW_AMP_NUM_OBS1], amp[AW], foot[6], qn[12], qv[12], small[64], dvp[DW_NUM_DOF];
    float acts[12];
    float hist[HIST_STAGE];                       // a row on its way (linear histories, the discriminator history)
    long long i64[4];
    int   touch, head[2];
    unsigned int draws[29 * 4];                   // reset_env: the env's generator blocks (12 of DS_RESET, 17 of DS_DR), one per lane
};
enum { SM_NZ = 0, SM_BIAS = 6, SM_QB = 18, SM_CMD = 21, SM_ACT = 24, SM_ACTP = 36, SM_EFF = 48 };          // words of StepLds::small

// the Gym tensors of the bound handle (what dw_simulate reads and writes)
struct GymRows { float *root_states, *dof_state, *contact_forces, *dof_damping, *dof_armature; };

// ---------------------------------------------------------------------------------------------------------------------- draws
struct DrawKey { unsigned long long seed; unsigned int env; unsigned long long ctr; };
enum { DS_RAMP = 1, DS_ENC = 2 /* + substep, < 8 */, DS_ROOTVEL = 12, DS_RESET = 13, DS_DR = 14 };
DW_HD void draw_block(const DrawKey &k, unsigned int stream, unsigned int idx, unsigned int *c) {
    c[0] = idx; c[1] = k.env; c[2] = (unsigned int)k.ctr; c[3] = ((unsigned int)(k.ctr >> 32) & 0x0fffffffu) | (stream << 28);
    dw::philox4x32_10(c, (unsigned int)k.seed, (unsigned int)(k.seed >> 32));
}
DW_HD unsigned int draw_u32(const DrawKey &k, unsigned int stream, int w) {
    unsigned int c[4];
    draw_block(k, stream, (unsigned int)(w >> 2), c);
    return c[w & 3];
}
// uniform in [0, 1) with 24 bits, as torch.rand makes its floats
DW_HD float draw_uniform(const DrawKey &k, unsigned int stream, int w) { return (float)(draw_u32(k, stream, w) >> 8) * 5.9604644775390625e-08f; }
// normal with sigma 0.00016 / 3 (the encoder model's): two per block
DW_HD float draw_enc_normal(const DrawKey &k, unsigned int stream, int w) {
    unsigned int c[4];
    draw_block(k, stream, (unsigned int)(w >> 1), c);
    return (w & 1) ? dw::enc_normal(c[2], c[3]) : dw::enc_normal(c[0], c[1]);
}
// integer in [lo, hi) (torch.randint: a 32-bit word modulo the range)
DW_HD long long draw_int(const DrawKey &k, unsigned int stream, int w, long long lo, long long hi) {
    return lo + (long long)(draw_u32(k, stream, w) % (unsigned int)(hi - lo));
}
// the same two conversions from a word a lane did not draw itself (reset_env: one block per lane, the words through LDS)
DW_HD float u32_uniform(unsigned int u) { return (float)(u >> 8) * 5.9604644775390625e-08f; }
DW_HD long long u32_int(unsigned int u, long long lo, long long hi) { return lo + (long long)(u % (unsigned int)(hi - lo)); }
DW_HD DrawKey draw_key(const DwAmpConfig &C, const DwAmpBuffers &B, int e) {
    DrawKey k;
    k.seed = C.seed; k.env = (unsigned int)e; k.ctr = C.device_draws ? (unsigned long long)B.draw_ctr[e] : 0ull;
    return k;
}

// ---------------------------------------------------------------------------------------------------------------------- pieces
DW_HD void stage_leg_model(const EnvWave &W, StepLds &S, const dw::DevModel &M) {
    W.par([&](int l) DWA_INL {
        for (int i = l; i < LegModel::NBODY * 16; i += 64) {
            const int b = i >> 4, k = i & 15;
            if (k < 3) S.LM.pos[b][k] = M.pos[b][k];
            else if (k < 6) S.LM.axis[b][k - 3] = M.axis[b][k - 3];
            else if (k < 15) S.LM.rot0[b][k - 6] = M.rot0[b][k - 6];
            else S.LM.parent[b] = M.parent[b];
        }
    });
}

DW_HD int hist_phys(int head, int logical, int nh) { const int p = head + logical; return p >= nh ? p - nh : p; }

// ---------------------------------------------------------------------------------------------------------------------- step
// The three step kernels run as workgroups of GT = 256 threads for GE = 16 envs (EnvGroup): their work is ITEMS -- (env, joint),
// (env, history column), (env, word) -- spread over all threads, so a launch of 16384 envs is 1024 workgroups of four full waves
// instead of 16384 workgroups of one wave with half its lanes idle (dw_amp_step_begin 20 -> 11 us at 16384 envs).  A region
// (EnvGroup::par) ends with a workgroup barrier; the same discipline as above holds for what may be read and written where.
// The group's thread count is a parameter of the group type (round 6): 256 for the three step kernels, 128 = the two wavefronts of an octet
// workgroup for the one-launch step (dw_k_amp_step_oct, dw_oct_kernels.hip), whose 16 envs are the same 16.  Every function below takes the
// group as a template argument and names its thread count GT.
#if !defined(DWA_GE)
#define DWA_GE 16          // (A/B builds only: envs per group of the three step kernels)
#endif
constexpr int GT = 256, GE = DWA_GE;
#if defined(__HIPCC__)
template <int GT_> struct EnvGroupT {
    static constexpr int GT = GT_;
    template <class F> DW_HD void par(F &&f) const { f((int)threadIdx.x); __syncthreads(); }
};
#else
template <int GT_> struct EnvGroupT {
    static constexpr int GT = GT_;
    template <class F> void par(F &&f) const { for (int t = 0; t < GT_; ++t) f(t); }
};
#endif
using EnvGroup = EnvGroupT<GT>;
struct BeginLds {
    long long i64[GE][2];
};

// the torques of one substep into DwAmpBuffers.tau (:696-724), items (env, joint).  A later region moves the FIFO counter
// (every leg item of the env read it).
// Every item loop of the step regions is written in TWO PHASES (round 6): first every global request of the thread's items, then the arithmetic
// and the stores.  As one loop -- load, compute, store, next item -- a store between two items' loads makes the compiler wait for memory once
// per item (it may not move a load over a store into tables it cannot tell apart), and with 3 .. 5 items per thread that was 3 .. 5 memory
// round trips per loop, ~140 per step.  An item beyond the group's envs requests row 0 of its tables (valid memory) and stores nothing.
constexpr int LOG_MAX = 8;          // FIFO slots the two-phase torque loop keeps in registers (a longer FIFO takes the serial loop)

template <class WG>
DW_HD void torques(const WG &W, const DwAmpConfig &C, const DwAmpBuffers &B, const float *dof_state, int e0) {
    constexpr int GT = WG::GT;
    constexpr int NIT = (GE * DW_NUM_DOF + GT - 1) / GT;
    const int N = C.num_envs, LS = C.log_slots;
    const bool fifo = !C.pd_control && LS <= LOG_MAX;
    W.par([&](int t) DWA_INL {
        if (!C.pd_control && !fifo) {          // (a FIFO longer than LOG_MAX: the serial form)
            for (int i = t; i < GE * DW_NUM_DOF; i += GT) {
                const int el = i / DW_NUM_DOF, l = i - DW_NUM_DOF * el, e = e0 + el;
                if (e >= N) continue;
                const float q = dof_state[((size_t)DW_NUM_DOF * e + l) * 2], qd = dof_state[((size_t)DW_NUM_DOF * e + l) * 2 + 1];
                float tau;
                if (l >= 12) tau = B.p_gains[l] * (B.init_angle[l] - q) + B.d_gains[l] * (-qd);
                else {
                    const int64_t sl0 = B.simul_len[e], dl = B.delay_idx[e];
                    const float m = B.motor_efforts[l];
                    float lower = B.actions[12 * (size_t)e + l] * m * B.power_scale[12 * (size_t)e + l];
                    lower = fmaxf(fminf(lower, m), -m);
                    float *col = B.action_log + (size_t)LS * 12 * e + l;
                    int64_t sl = sl0 + 1;
                    sl = sl > LS ? LS : (sl < 0 ? 0 : sl);
                    float delayed = 0.0f;
                    for (int s = 0; s < LS; ++s) {
                        const float v = s + 1 < LS ? col[(size_t)12 * (s + 1)] : lower;
                        col[(size_t)12 * s] = v;
                        const int64_t want = sl > dl ? dl : (int64_t)LS - sl;
                        if (s == want) delayed = v;
                    }
                    tau = C.noise ? delayed : lower;
                }
                B.tau[(size_t)DW_NUM_DOF * e + l] = tau;
            }
            return;
        }
        // ---- phase 1: requests
        float q[NIT], qd[NIT], act[NIT], ps[NIT], col[NIT][LOG_MAX];
        int64_t sl0[NIT], dl[NIT];
        DWA_UNROLL for (int k = 0; k < NIT; ++k) {
            const int i = t + GT * k, ic = i < GE * DW_NUM_DOF ? i : 0;
            const int el = ic / DW_NUM_DOF, l = ic - DW_NUM_DOF * el, e = e0 + el < N ? e0 + el : N - 1;
            q[k] = dof_state[((size_t)DW_NUM_DOF * e + l) * 2]; qd[k] = dof_state[((size_t)DW_NUM_DOF * e + l) * 2 + 1];
            act[k] = 0.0f; ps[k] = 0.0f; sl0[k] = 0; dl[k] = 0;
            DWA_UNROLL for (int s = 0; s < LOG_MAX; ++s) col[k][s] = 0.0f;
            if (l < 12) {          // (the legs: the action, and with the delayed-torque model the env's FIFO column)
                act[k] = B.actions[12 * (size_t)e + l];
                if (fifo) {
                    ps[k] = B.power_scale[12 * (size_t)e + l];
                    sl0[k] = B.simul_len[e]; dl[k] = B.delay_idx[e];
                    const float *cl = B.action_log + (size_t)LS * 12 * e + l;
                    DWA_UNROLL for (int s = 0; s < LOG_MAX; ++s) if (s < LS) col[k][s] = cl[(size_t)12 * s];
                }
            }
        }
        // ---- phase 2: arithmetic and stores (the expressions of the serial form above, :696-724)
        DWA_UNROLL for (int k = 0; k < NIT; ++k) {
            const int i = t + GT * k;
            const int el = i / DW_NUM_DOF, l = i - DW_NUM_DOF * el, e = e0 + el;
            if (i >= GE * DW_NUM_DOF || e >= N) continue;
            float tau;
            if (l >= 12) {
                tau = B.p_gains[l] * (B.init_angle[l] - q[k]) + B.d_gains[l] * (-qd[k]);          // upper body: PD to the initial pose (:696)
            } else if (C.pd_control) {
                const float tar = B.pd_action_offset[l] + B.pd_action_scale[l] * act[k];
                tau = B.p_gains[l] * (tar - q[k]) + B.d_gains[l] * (-qd[k]);
            } else {
                const float m = B.motor_efforts[l];
                float lower = act[k] * m * ps[k];
                lower = fmaxf(fminf(lower, m), -m);
                // delayed-torque FIFO (:712-724), column l: shift, append, read `delay_idx` back once the FIFO has filled that far
                float *cl = B.action_log + (size_t)LS * 12 * e + l;
                int64_t sl = sl0[k] + 1;
                sl = sl > LS ? LS : (sl < 0 ? 0 : sl);
                const int64_t want = sl > dl[k] ? dl[k] : (int64_t)LS - sl;
                float delayed = 0.0f;
                DWA_UNROLL for (int s = 0; s < LOG_MAX; ++s) {
                    if (s >= LS) break;
                    const float v = s + 1 < LS ? col[k][s + 1 < LOG_MAX ? s + 1 : LOG_MAX - 1] : lower;
                    cl[(size_t)12 * s] = v;
                    if (s == want) delayed = v;
                }
                tau = C.noise ? delayed : lower;
            }
            B.tau[(size_t)DW_NUM_DOF * e + l] = tau;
        }
    });
    if (!C.pd_control) {
        W.par([&](int t) DWA_INL {
            const int e = e0 + t;
            if (t >= GE || e >= N) return;
            const int64_t sl = B.simul_len[e] + 1;
            B.simul_len[e] = sl > C.log_slots ? C.log_slots : (sl < 0 ? 0 : sl);
        });
    }
}

// the encoder model after one substep (:728-736), items (env, joint): z = the caller's normal draws [N,33] or nullptr (device
// draws / no noise)
template <class WG>
DW_HD void encoder(const WG &W, const DwAmpConfig &C, const DwAmpBuffers &B, const float *dof_state, const float *z, int substep, int e0) {
    constexpr int GT = WG::GT;
    constexpr int NIT = (GE * DW_NUM_DOF + GT - 1) / GT;
    const int N = C.num_envs;
    W.par([&](int t) DWA_INL {
        float q[NIT], zz[NIT], pre[NIT];
        unsigned long long ctr[NIT];
        DWA_UNROLL for (int k = 0; k < NIT; ++k) {
            const int i = t + GT * k, ic = i < GE * DW_NUM_DOF ? i : 0;
            const int el = ic / DW_NUM_DOF, l = ic - DW_NUM_DOF * el, e = e0 + el < N ? e0 + el : N - 1;
            const size_t g = (size_t)DW_NUM_DOF * e + l;
            q[k] = dof_state[g * 2];
            pre[k] = B.qpos_pre[g];
            zz[k] = (C.noise && z) ? z[g] : 0.0f;
            ctr[k] = (C.noise && !z && C.device_draws) ? (unsigned long long)B.draw_ctr[e] : 0ull;
        }
        DWA_UNROLL for (int k = 0; k < NIT; ++k) {
            const int i = t + GT * k;
            const int el = i / DW_NUM_DOF, l = i - DW_NUM_DOF * el, e = e0 + el;
            if (i >= GE * DW_NUM_DOF || e >= N) continue;
            const size_t g = (size_t)DW_NUM_DOF * e + l;
            float qn = q[k];
            if (C.noise) {
                DrawKey dk; dk.seed = C.seed; dk.env = (unsigned int)e; dk.ctr = ctr[k];
                const float zk = z ? zz[k] : draw_enc_normal(dk, DS_ENC + (unsigned int)substep, l);
                qn = q[k] + fminf(fmaxf(zk, -0.00016f), 0.00016f);
            }
            const float d = qn - pre[k];
            B.qpos_noise[g] = qn;
            B.qvel_noise[g] = C.gpu_div ? d * C.inv_dt : d / C.dt;
            B.qpos_pre[g] = qn;
        }
    });
}

// dw_amp_step_begin: action clamp + record + action history, command ramp (:642-693), then the torques of the first substep.
template <class WG>
DW_HD void step_begin(const WG &W, BeginLds &S, const DwAmpConfig &C, const DwAmpBuffers &B, const float *dof_state, const float *actions_in,
                      const int64_t *ramp_dur, const float *ramp_u, int group) {
    constexpr int GT = WG::GT;
    const int N = C.num_envs, e0 = group * GE, NH = C.num_his * C.num_skip;
    W.par([&](int t0) DWA_INL {
      for (int t = t0; t < GE * 12 + GE * 3; t += GT) {
        // items (env, action): clamp, record, history.  An item owns column k of the env's action history, so the shifting layout
        // moves in place without a hazard between threads; the ring writes one slot (its head moves in the next region).
        if (t < GE * 12) {
            const int el = t / 12, k = t - 12 * el, e = e0 + el;
            if (e < N) {
                float a = actions_in[12 * (size_t)e + k];
                a = fminf(fmaxf(a, -C.clip_actions), C.clip_actions);
                B.actions[12 * (size_t)e + k] = a;
                float *ah = B.action_history + (size_t)NH * 12 * e;
                if (C.hist_ring) {
                    ah[(size_t)B.hist_head[2 * (size_t)e] * 12 + k] = a;
                } else {
                    for (int s = 0; s + 1 < NH; ++s) ah[(size_t)s * 12 + k] = ah[(size_t)(s + 1) * 12 + k];
                    ah[(size_t)(NH - 1) * 12 + k] = a;
                }
            }
        } else if (C.vel_change && t < GE * 12 + GE * 3) {
            // command ramp (:676-693), the branch of the host class that draws for every env: items (env, component)
            const int i = t - GE * 12, el = i / 3, l = i - 3 * el, e = e0 + el;
            if (e < N) {
                const int half = (int)(C.max_episode_length / 2), when = (int)(C.max_episode_length / 4 - 1);
                const bool change = fmodf(B.epi_len[e], (float)half) == (float)when;
                int64_t dur = B.vel_change_duration[e], cur = B.cur_vel_change_duration[e];
                float start = B.start_target_vel[3 * (size_t)e + l], fin = B.final_target_vel[3 * (size_t)e + l], cmd = B.commands[3 * (size_t)e + l];
                if (change) {
                    const DrawKey k = draw_key(C, B, e);
                    dur = ramp_dur ? ramp_dur[e] : draw_int(k, DS_RAMP, 0, 1, 250);
                    cur = 0;
                    start = cmd;
                    const float u = ramp_u ? ramp_u[3 * (size_t)e + l] : draw_uniform(k, DS_RAMP, 1 + l);
                    fin = C.cmd_scale[l] * u + C.cmd_lo[l];
                }
                const bool mask = cur < dur;
                const float ramp = start + (fin - start) * (float)cur / (float)dur;
                if (mask) cmd = ramp;
                B.start_target_vel[3 * (size_t)e + l] = start;
                B.final_target_vel[3 * (size_t)e + l] = fin;
                B.commands[3 * (size_t)e + l] = cmd;
                if (l == 0) { S.i64[el][0] = dur; S.i64[el][1] = cur + (mask ? 1 : 0); }
            }
        }
      }
    });
    W.par([&](int t) DWA_INL {
        const int e = e0 + t;
        if (t >= GE || e >= N) return;
        if (C.vel_change) { B.vel_change_duration[e] = S.i64[t][0]; B.cur_vel_change_duration[e] = S.i64[t][1]; }
        if (C.hist_ring) { const int h = B.hist_head[2 * (size_t)e] + 1; B.hist_head[2 * (size_t)e] = h >= NH ? 0 : h; }
    });
    torques(W, C, B, dof_state, e0);
}

// dw_amp_step_mid: between two substeps -- the encoder model of the one that ended, the torques of the one that starts
template <class WG>
DW_HD void step_mid(const WG &W, const DwAmpConfig &C, const DwAmpBuffers &B, const float *dof_state, const float *z, int substep, int group) {
    encoder(W, C, B, dof_state, z, substep, group * GE);
    torques(W, C, B, dof_state, group * GE);
}

// dw_amp_step_end: the encoder model of the last substep, then post-physics (:750-804 + the subclass' :88-96): counters, foot
// positions, observation + history stacking, reward, termination, time-outs, the discriminator observation and its history.
//
// A WORKGROUP OF FOUR WAVES PER 16 ENVS (EnvGroup): the serial per-env functions -- two chains of six joint rotations for the foot
// positions, the observation's three atan2f, the reward's eight expf and four norms, the discriminator observation -- are ~3 000
// instructions; on one lane of a wave per env (the first form of this kernel) they cost 3 000 wave-instructions PER ENV and the
// launch was bound by instruction issue at 1/64 lane occupancy (83 us at 16384 envs).  Here the rows those functions read are
// staged in LDS for the group's envs by all 256 threads (items = (env, word), coalesced), the functions then run with LANE = ENV, one
// function per wave (left foot + termination | observation | reward | right foot + discriminator observation), and the bulk
// writes are items over all threads again.  The env's row in LDS has an odd stride, so lane = env access is free of bank conflicts.
// 16 envs per group, not 64: with 64 the serial functions cost nothing (1.5 us) but a launch of 16384 envs is 256 workgroups, four
// waves per CU, and the item loops -- a dependent global load per iteration -- ran at the latency of one wave (127 us measured);
// 1024 workgroups keep 16 waves per CU in flight.
enum { GR_ROOT = 0, GR_DS = 13, GR_FZ = 79, GR_QN = 81, GR_QV = 93, GR_NZ = 105, GR_BIAS = 111, GR_QB = 123, GR_CMD = 126, GR_ACT = 129, GR_ACTP = 141,
       GR_DVP = 153, GR_OBS = 186, GR_AMP = 222, GR_FOOT = 256, GR_WORDS = 262, GR_STRIDE = 263 };
struct GroupLds {
    LegModel LM;
    float eff[12];
    float row[GE][GR_STRIDE];
    int   touch[GE], head[GE][2];
};

template <class WG>
DW_HD void step_end(const WG &W, GroupLds &S, const dw::DevModel &M, const DwAmpConfig &C, const DwAmpBuffers &B, const GymRows &G, const float *z,
                    int substep, const float *rootvel_noise, int group) {
    constexpr int GT = WG::GT;
    static_assert(GT % 64 == 0 && GT >= 64, "whole wavefronts");
    const int N = C.num_envs, e0 = group * GE;
    const int NH = C.num_his * C.num_skip;
    const int num_obs = (DW_AMP_NUM_OBS1 + 12) * C.num_his - 12;
    W.par([&](int t) DWA_INL {
        for (int i = t; i < LegModel::NBODY * 16; i += GT) {
            const int b = i >> 4, k = i & 15;
            if (k < 3) S.LM.pos[b][k] = M.pos[b][k];
            else if (k < 6) S.LM.axis[b][k - 3] = M.axis[b][k - 3];
            else if (k < 15) S.LM.rot0[b][k - 6] = M.rot0[b][k - 6];
            else S.LM.parent[b] = M.parent[b];
        }
        if (t < 12) S.eff[t] = B.motor_efforts[t];
        if (t < GE) {
            S.touch[t] = 0;
            S.head[t][0] = S.head[t][1] = 0;
            if (C.hist_ring && e0 + t < N) { S.head[t][0] = B.hist_head[2 * (size_t)(e0 + t)]; S.head[t][1] = B.hist_head[2 * (size_t)(e0 + t) + 1]; }
        }
    });
    // ---- the encoder model of the substep that ended (:728-736) and the rows the functions below read, items over all threads; every
    //      loop in two phases (all requests, then arithmetic and stores: see torques())
    W.par([&](int t) DWA_INL {
        {
            constexpr int NIT = (GE * DW_NUM_DOF + GT - 1) / GT;
            float q[NIT], qd[NIT], zz[NIT], pre[NIT], dvp[NIT];
            unsigned long long ctr[NIT];
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k, ic = i < GE * DW_NUM_DOF ? i : 0;
                const int el = ic / DW_NUM_DOF, l = ic - DW_NUM_DOF * el, e = e0 + el < N ? e0 + el : N - 1;
                const size_t g = (size_t)DW_NUM_DOF * e + l;
                q[k] = G.dof_state[g * 2]; qd[k] = G.dof_state[g * 2 + 1];
                pre[k] = B.qpos_pre[g]; dvp[k] = B.dof_vel_pre[g];
                zz[k] = (C.noise && z) ? z[g] : 0.0f;
                ctr[k] = (C.noise && !z && C.device_draws) ? (unsigned long long)B.draw_ctr[e] : 0ull;
            }
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k;
                const int el = i / DW_NUM_DOF, l = i - DW_NUM_DOF * el, e = e0 + el;
                if (i >= GE * DW_NUM_DOF || e >= N) continue;
                const size_t g = (size_t)DW_NUM_DOF * e + l;
                float qn = q[k];
                if (C.noise) {
                    DrawKey dk; dk.seed = C.seed; dk.env = (unsigned int)e; dk.ctr = ctr[k];
                    const float zk = z ? zz[k] : draw_enc_normal(dk, DS_ENC + (unsigned int)substep, l);
                    qn = q[k] + fminf(fmaxf(zk, -0.00016f), 0.00016f);
                }
                const float d = qn - pre[k];
                const float qv = C.gpu_div ? d * C.inv_dt : d / C.dt;
                B.qpos_noise[g] = qn;
                B.qvel_noise[g] = qv;
                B.qpos_pre[g] = qn;
                float *r = S.row[el];
                if (l < 12) { r[GR_QN + l] = qn; r[GR_QV + l] = qv; }
                r[GR_DS + 2 * l] = q[k]; r[GR_DS + 2 * l + 1] = qd[k];
                r[GR_DVP + l] = dvp[k];
            }
        }
        {   // root state (13) + the three words of quat_bias
            constexpr int NIT = (GE * 16 + GT - 1) / GT;
            float v[NIT];
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k, ic = i < GE * 16 ? i : 0;
                const int el = ic >> 4, w = ic & 15, e = e0 + el < N ? e0 + el : N - 1;
                v[k] = w < 13 ? G.root_states[13 * (size_t)e + w] : B.quat_bias[3 * (size_t)e + (w - 13)];
            }
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k;
                const int el = i >> 4, w = i & 15, e = e0 + el;
                if (i >= GE * 16 || e >= N) continue;
                if (w < 13) S.row[el][GR_ROOT + w] = v[k];
                else S.row[el][GR_QB + (w - 13)] = v[k];
            }
        }
        {   // qpos_bias, actions, actions_pre (12 each), root-velocity noise (6), command (3), foot forces (2)
            constexpr int NIT = (GE * 48 + GT - 1) / GT;
            float v[NIT];
            unsigned long long ctr[NIT];
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k, ic = i < GE * 48 ? i : 0;
                const int el = ic / 48, w = ic - 48 * el, e = e0 + el < N ? e0 + el : N - 1;
                float x = 0.0f;
                ctr[k] = 0ull;
                if (w < 12) x = B.qpos_bias[12 * (size_t)e + w];
                else if (w < 24) x = B.actions[12 * (size_t)e + (w - 12)];
                else if (w < 36) x = B.actions_pre[12 * (size_t)e + (w - 24)];
                else if (w < 42) {
                    if (rootvel_noise) x = rootvel_noise[6 * (size_t)e + (w - 36)];
                    else if (C.noise && C.device_draws) ctr[k] = (unsigned long long)B.draw_ctr[e];
                } else if (w < 45) x = B.commands[3 * (size_t)e + (w - 42)];
                else if (w < 47) x = G.contact_forces[((size_t)DW_NUM_BODIES * e + (w == 45 ? 8 : 16)) * 3 + 2];
                v[k] = x;
            }
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k;
                const int el = i / 48, w = i - 48 * el, e = e0 + el;
                if (i >= GE * 48 || e >= N) continue;
                float *r = S.row[el];
                if (w < 12) r[GR_BIAS + w] = v[k];
                else if (w < 24) r[GR_ACT + (w - 12)] = v[k];
                else if (w < 36) r[GR_ACTP + (w - 24)] = v[k];
                else if (w < 42) {
                    float nz = 0.0f;
                    if (rootvel_noise) nz = v[k];
                    else if (C.noise && C.device_draws) { DrawKey dk; dk.seed = C.seed; dk.env = (unsigned int)e; dk.ctr = ctr[k]; nz = draw_uniform(dk, DS_ROOTVEL, w - 36) * 0.05f - 0.025f; }
                    r[GR_NZ + (w - 36)] = nz;
                } else if (w < 45) r[GR_CMD + (w - 42)] = v[k];
                else if (w < 47) r[GR_FZ + (w - 45)] = v[k];
            }
        }
        {   // non-foot bodies in contact
            constexpr int NIT = (GE * DW_NUM_BODIES + GT - 1) / GT;
            float c0[NIT], c1[NIT], c2[NIT];
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k, ic = i < GE * DW_NUM_BODIES ? i : 0;
                const int el = ic / DW_NUM_BODIES, l = ic - DW_NUM_BODIES * el, e = e0 + el < N ? e0 + el : N - 1;
                const float *cf = G.contact_forces + ((size_t)DW_NUM_BODIES * e + l) * 3;
                c0[k] = cf[0]; c1[k] = cf[1]; c2[k] = cf[2];
            }
            DWA_UNROLL for (int k = 0; k < NIT; ++k) {
                const int i = t + GT * k;
                const int el = i / DW_NUM_BODIES, l = i - DW_NUM_BODIES * el, e = e0 + el;
                if (i >= GE * DW_NUM_BODIES || e >= N || l == 8 || l == 16) continue;
                if (c0[k] > 1.0f || c1[k] > 1.0f || c2[k] > 1.0f) S.touch[el] = 1;
            }
        }
    });
    // ---- the serial functions, lane = env, one function per wave; counters (:751-752; epi_len: the last line of pre_physics_step)
    // (four roles; a group of fewer than four wavefronts gives a wavefront several of them in turn)
    W.par([&](int t) DWA_INL {
        const int el = t & 63, e = e0 + el;
        if (el >= GE || e >= N) return;
        float *r = S.row[el];
      for (int role = t >> 6; role < 4; role += GT / 64) {
        if (role == 0 || role == 3) {
            const int f = role == 0 ? 0 : 1;
            float p[3];
            body_position(S.LM, r + GR_ROOT, r + GR_DS, 0, f == 0 ? 6 : 12, p);          // (the staged rows are env 0 of their own little tensors)
            for (int i = 0; i < 3; ++i) {
                r[GR_FOOT + 3 * f + i] = p[i];
                B.foot_pos[((size_t)2 * e + f) * 3 + i] = p[i];
                B.rigid_body_pos[((size_t)DW_NUM_BODIES * e + (f == 0 ? 8 : 16)) * 3 + i] = p[i];
            }
        } else if (role == 1) {
            observations_row(r + GR_ROOT, r + GR_NZ, r + GR_QN, r + GR_BIAS, r + GR_QB, r + GR_QV, r + GR_CMD, r + GR_OBS);
            B.progress_buf[e] += 1; B.randomize_buf[e] += 1; B.epi_len[e] += 1.0f;
        } else {
            reward_row(r + GR_ROOT, r + GR_DS + 1, 2, r + GR_DVP, r + GR_CMD, r + GR_ACT, r + GR_ACTP, S.eff, r[GR_FZ], r[GR_FZ + 1], B.total_mass[e],
                       B.rew_buf + e, B.reward_values + 9 * (size_t)e);
        }
      }
    });
    W.par([&](int t) DWA_INL {
        const int el = t & 63, e = e0 + el;
        if (el >= GE || e >= N) return;
        const float *r = S.row[el];
      for (int role = t >> 6; role < 4; role += GT / 64) {
        if (role == 0) {          // termination (:1025-1069)
            const int64_t prog = B.progress_buf[e];
            int64_t term = 0;
            if (C.enable_early_termination) {
                bool fall_height = r[GR_ROOT + 2] < C.termination_height;
                fall_height = fall_height || r[GR_FOOT + 2] > 0.5f || r[GR_FOOT + 5] > 0.5f;
                bool fallen = (S.touch[el] != 0) || fall_height;
                const float q0[4] = {r[GR_ROOT + 3], r[GR_ROOT + 4], r[GR_ROOT + 5], r[GR_ROOT + 6]};
                fallen = fallen || fabsf(dw::quat_err(q0)) > (float)(3.141592 / 4.0);
                fallen = fallen && (prog > 1);
                term = fallen ? 1 : 0;
            }
            const int64_t rs = ((float)prog >= C.max_episode_length - 1.0f) ? 1 : term;
            B.terminate_buf[e] = term;
            B.reset_buf[e] = rs;
            B.timeout_buf[e] = (uint8_t)(((float)prog >= C.max_episode_length - 1.0f) && rs != 0);
        } else if (role == 3) {   // the discriminator observation (tasks/tocabi_amp_lower.py:310-350)
            disc_observations_row(r + GR_ROOT, r + GR_DS, r + GR_DS + 1, 2, C.local_root_obs, r + GR_FOOT, 2, S.row[el] + GR_AMP);
        }
      }
    });
    // ---- what goes back to memory, items over all threads.  Histories: a thread owns a COLUMN of an env's history (slot s, word k
    //      for all s), so the shifting layout moves in place without a hazard between threads; the ring writes one slot.
    W.par([&](int t) DWA_INL {
        for (int i = t; i < GE * 64; i += GT) {
            const int el = i >> 6, k = i & 63, e = e0 + el;
            if (e >= N) continue;
            const float *r = S.row[el];
            // the encoder reading takes the bias (the reference's observation function adds it in place, :945)
            if (k < 12) B.qpos_noise[(size_t)DW_NUM_DOF * e + k] = r[GR_QN + k] + r[GR_BIAS + k];
            else if (k < 24) B.actions_pre[12 * (size_t)e + (k - 12)] = r[GR_ACT + (k - 12)];
            else if (k < 27) B.rigid_body_pos[(size_t)DW_NUM_BODIES * 3 * e + (k - 24)] = r[GR_ROOT + (k - 24)];
            else if (k < 31) B.rigid_body_rot[(size_t)DW_NUM_BODIES * 4 * e + (k - 27)] = r[GR_ROOT + 3 + (k - 27)];
            if (k < DW_NUM_DOF) B.dof_vel_pre[(size_t)DW_NUM_DOF * e + k] = r[GR_DS + 2 * k + 1];          // what the next step compares against
            if (k < DW_AMP_NUM_OBS1) {
                B.obs1[DW_AMP_NUM_OBS1 * (size_t)e + k] = r[GR_OBS + k];
                float *oh = B.obs_history + (size_t)NH * DW_AMP_NUM_OBS1 * e;
                if (C.hist_ring) {
                    oh[(size_t)S.head[el][1] * DW_AMP_NUM_OBS1 + k] = r[GR_OBS + k];
                } else {
                    for (int s = 0; s + 1 < NH; ++s) oh[(size_t)s * DW_AMP_NUM_OBS1 + k] = oh[(size_t)(s + 1) * DW_AMP_NUM_OBS1 + k];
                    oh[(size_t)(NH - 1) * DW_AMP_NUM_OBS1 + k] = r[GR_OBS + k];
                }
            }
            if (k < AW) {
                // discriminator observation history (tasks/tocabi_amp_lower.py:88-96): slot s -> s + 1, the newest into slot 0 (every slot of
                // the column requested before the first is overwritten: one round trip instead of amp_steps - 1)
                float *ab = B.amp_obs_buf + (size_t)C.amp_steps * AW * e;
                constexpr int AS_MAX = 8;
                if (C.amp_steps <= AS_MAX) {
                    float old_[AS_MAX];
                    DWA_UNROLL for (int s2 = 0; s2 < AS_MAX; ++s2) old_[s2] = s2 + 1 < C.amp_steps ? ab[(size_t)s2 * AW + k] : 0.0f;
                    DWA_UNROLL for (int s2 = 0; s2 < AS_MAX; ++s2) if (s2 + 1 < C.amp_steps) ab[(size_t)(s2 + 1) * AW + k] = old_[s2];
                } else {
                    for (int s2 = C.amp_steps - 1; s2 >= 1; --s2) ab[(size_t)s2 * AW + k] = ab[(size_t)(s2 - 1) * AW + k];
                }
                ab[k] = r[GR_AMP + k];
                B.amp_obs1[(size_t)AW * e + k] = r[GR_AMP + k];
            }
        }
    });
    // the stacked observation (:540-580): obs slots S (i + 1) - 1, action slots S (i + 1), i < H - 1.  Eight items per thread at a
    // time, every load before the first store (the compiler may not move a load over a store into the same table)
    W.par([&](int t) DWA_INL {
#if !defined(DWA_SB)
#define DWA_SB 8          // (A/B builds only)
#endif
        constexpr int SB = DWA_SB;          // items per thread in flight (four until round 6)
        for (int i0 = t; i0 < GE * num_obs; i0 += SB * GT) {
            float v[SB];
            size_t dst[SB];
            bool ok[SB];
            DWA_UNROLL for (int u = 0; u < SB; ++u) {
                const int i = i0 + u * GT;
                const int el = i / num_obs, w = i - num_obs * el, e = e0 + el;
                ok[u] = i < GE * num_obs && e < N;
                v[u] = 0.0f; dst[u] = 0;
                if (!ok[u]) continue;
                int oh_head = S.head[el][1] + (C.hist_ring ? 1 : 0);          // (the head as it will be once this step's entry counts)
                oh_head = oh_head >= NH ? 0 : oh_head;
                const int ah_head = S.head[el][0];
                if (w < DW_AMP_NUM_OBS1 * C.num_his) {
                    const int slot = w / DW_AMP_NUM_OBS1, k = w - DW_AMP_NUM_OBS1 * slot;
                    v[u] = B.obs_history[(size_t)NH * DW_AMP_NUM_OBS1 * e + (size_t)hist_phys(oh_head, C.num_skip * (slot + 1) - 1, NH) * DW_AMP_NUM_OBS1 + k];
                } else {
                    const int j = w - DW_AMP_NUM_OBS1 * C.num_his, slot = j / 12, k = j - 12 * slot;
                    v[u] = B.action_history[(size_t)NH * 12 * e + (size_t)hist_phys(ah_head, C.num_skip * (slot + 1), NH) * 12 + k];
                }
                dst[u] = (size_t)num_obs * e + w;
            }
            DWA_UNROLL for (int u = 0; u < SB; ++u) {
                if (!ok[u]) continue;
                B.obs_buf[dst[u]] = v[u];
                B.obs_out[dst[u]] = fminf(fmaxf(v[u], -C.clip_obs), C.clip_obs);
            }
        }
    });
    W.par([&](int t) DWA_INL {
        const int e = e0 + t;
        if (t >= GE || e >= N) return;
        if (C.hist_ring) { const int h = S.head[t][1] + 1; B.hist_head[2 * (size_t)e + 1] = h >= NH ? 0 : h; }
        if (C.device_draws) B.draw_ctr[e] += 1;
    });
}

// ---------------------------------------------------------------------------------------------------------------------- reset
// Where a reset env's draws come from: the caller's RAW uniforms (rows of the arrays at `row`: position in an id list for
// dw_amp_reset_rows, the env for dw_amp_reset_done) or the generator.
struct ResetSrc {
    const float *ps, *cx, *cy, *cyaw, *qb, *quatb, *damp, *arm;
    const int64_t *ptime, *didx;
    const float *rootvel_noise;          // [N,6], by env
    size_t row;
    bool dr;                             // dw_amp_reset_done: the kernel also does the dof-property randomisation (under DwAmpConfig.randomize) and writes obs_out
    bool power;                          // draw power_scale
};

// reset_idx of env e (tasks/amp/tocabi_amp_lower_base.py:238-305 with the default state initialisation, then
// tasks/tocabi_amp_lower.py:144-147,258-272)
DW_HD void reset_env(const EnvWave &W, StepLds &S, const dw::DevModel &M, const DwAmpConfig &C, const DwAmpBuffers &B, const GymRows &G,
                     const ResetSrc &R, int e) {
    // Five regions (six until round 6, each ending in a drain of the wave's outstanding memory operations): (1) EVERY global read of the
    // reset -- the leg model, the old episode's last readings, the action-history words the reset observation shows, counters -- and every
    // generator block the reset draws from, one per lane; (2) the stores that depend on nothing else; (3) the serial functions and the stores
    // that must follow the reads of (1); (4) what needs (3)'s rows; (5) what needs (4)'s.  Same values, same arithmetic, same order of the draws as the reference's reset_idx.
    enum { SO_QN = 0, SO_QV = 12, SO_BIAS = 24, SO_QB = 36, SO_CMD = 39, SO_NZ = 42, SO_EPI = 48 };
    const int NH = C.num_his * C.num_skip;
    const bool dev = C.device_draws != 0;
    const int num_obs = (DW_AMP_NUM_OBS1 + 12) * C.num_his - 12;
    const int obs_part = DW_AMP_NUM_OBS1 * C.num_his;          // the stacked observation: [0, obs_part) observation slots, then action slots
    // word w of the env's generator stream (DS_RESET: blocks 0 .. 11, DS_DR: 12 .. 28), drawn by lane `block` in region (1)
    auto word = [&](unsigned int stream, int w) -> unsigned int { return S.draws[4 * ((stream == DS_RESET ? 0 : 12) + (w >> 2)) + (w & 3)]; };
    W.par([&](int l) DWA_INL {
        // ---- requests
        const unsigned long long ctr = dev ? (unsigned long long)B.draw_ctr[e] : 0ull;
        const int ah_head = C.hist_ring ? B.hist_head[2 * (size_t)e] : 0;
        float qn = 0.0f, qv = 0.0f, bias = 0.0f, qb = 0.0f, cmd = 0.0f, nzs = 0.0f, root = 0.0f, q0 = 0.0f, epi = 0.0f;
        if (l < 12) { qn = B.qpos_noise[(size_t)DW_NUM_DOF * e + l]; qv = B.qvel_noise[(size_t)DW_NUM_DOF * e + l]; bias = B.qpos_bias[12 * (size_t)e + l]; }
        if (l < 3) { qb = B.quat_bias[3 * (size_t)e + l]; cmd = B.commands[3 * (size_t)e + l]; }
        if (l < 6 && R.rootvel_noise) nzs = R.rootvel_noise[6 * (size_t)e + l];
        if (l < 13) root = B.initial_root_states[13 * (size_t)e + l];
        if (l < DW_NUM_DOF) q0 = B.init_angle[l];
        if (l == 63) epi = B.epi_len[e];
        const bool drr = R.dr && C.randomize && B.randomize_buf[e] >= (int64_t)C.dr_frequency;
        for (int i = l; i < LegModel::NBODY * 16; i += 64) {
            const int b = i >> 4, k = i & 15;
            if (k < 3) S.LM.pos[b][k] = M.pos[b][k];
            else if (k < 6) S.LM.axis[b][k - 3] = M.axis[b][k - 3];
            else if (k < 15) S.LM.rot0[b][k - 6] = M.rot0[b][k - 6];
            else S.LM.parent[b] = M.parent[b];
        }
        // the action slots of the reset observation: what the action history holds NOW (it is zeroed two regions on)
        const float *ah = B.action_history + (size_t)NH * 12 * e;
        for (int i = obs_part + l; i < num_obs; i += 64) {
            const int j = i - obs_part, slot = j / 12, kk = j - 12 * slot;
            S.hist[j] = ah[(size_t)hist_phys(ah_head, C.num_skip * (slot + 1), NH) * 12 + kk];
        }
        // the env's generator blocks, ONE per lane (29 of them: every draw of the reset; lanes that each drew what they needed ran the
        // generator nine times one after the other, in divergent branches)
        if (l < 29) {          // (without device_draws the counter reads 0 and every draw is the caller's: the entry points insist on their arrays)
            DrawKey k; k.seed = C.seed; k.env = (unsigned int)e; k.ctr = ctr;
            unsigned int c4[4];
            draw_block(k, l < 12 ? (unsigned int)DS_RESET : (unsigned int)DS_DR, (unsigned int)(l < 12 ? l : l - 12), c4);
            for (int i = 0; i < 4; ++i) S.draws[4 * l + i] = c4[i];
        }
        // ---- what the reset observation is made of: the episode's LAST encoder reading, biases and command (the reference computes it
        //      before it draws the new ones, :253 before :266-279)
        if (l == 0) { S.i64[0] = (long long)ctr; S.touch = drr ? 1 : 0; }
        if (l < 12) { S.small[SO_QN + l] = qn; S.small[SO_QV + l] = qv; S.small[SO_BIAS + l] = bias; }
        if (l < 3) { S.small[SO_QB + l] = qb; S.small[SO_CMD + l] = cmd; }
        if (l < 6) S.small[SO_NZ + l] = nzs;
        if (l == 63) S.small[SO_EPI] = epi;
        if (l < 13) S.root[l] = root;
        if (l < DW_NUM_DOF) { S.ds[2 * l] = q0; S.ds[2 * l + 1] = 0.0f; }
    });
    W.par([&](int l) DWA_INL {
        if (l < 6 && !R.rootvel_noise) S.small[SO_NZ + l] = (C.noise && dev) ? u32_uniform(word(DS_RESET, 40 + l)) * 0.05f - 0.025f : 0.0f;
        // the Gym tensors' rows: initial root state, initial pose at rest, no contact (_reset_actors, :611-626)
        if (l < 13) G.root_states[13 * (size_t)e + l] = S.root[l];
        if (l < DW_NUM_DOF) { G.dof_state[((size_t)DW_NUM_DOF * e + l) * 2] = S.ds[2 * l]; G.dof_state[((size_t)DW_NUM_DOF * e + l) * 2 + 1] = 0.0f; }
        for (int i = l; i < DW_NUM_BODIES * 3; i += 64) G.contact_forces[(size_t)DW_NUM_BODIES * 3 * e + i] = 0.0f;
        // (the draws arrive as raw uniforms; the values are formed with torch's arithmetic: `(hi - lo) * u + lo` with the scalars
        //  rounded to float32 first, `x / s` as a multiplication by 1.0f / s on a GPU and a division on a CPU)
        if (R.power && l < 12) {
            const float u = R.ps ? R.ps[12 * R.row + l] : u32_uniform(word(DS_RESET, l));
            B.power_scale[12 * (size_t)e + l] = (float)(1.2 - 0.8) * u + (float)0.8;
        }
        // dof properties (apply_randomizations, tasks/base/vec_task.py:519-733): additive damping, scaled armature, from the
        // nominal values, for a resetting env whose randomize_buf has reached the frequency
        // (only under task.randomize, as the torch class and the reference: apply_randomizations is what resets randomize_buf)
        if (S.touch && l < DW_NUM_DOF) {
            if (C.dr_damping) {
                const float u = R.damp ? R.damp[(size_t)DW_NUM_DOF * R.row + l] : u32_uniform(word(DS_DR, l));
                G.dof_damping[(size_t)DW_NUM_DOF * e + l] = B.nominal_damping[l] + ((C.dr_damping_range[1] - C.dr_damping_range[0]) * u + C.dr_damping_range[0]);
            }
            if (C.dr_armature) {
                const float u = R.arm ? R.arm[(size_t)DW_NUM_DOF * R.row + l] : u32_uniform(word(DS_DR, DW_NUM_DOF + l));
                G.dof_armature[(size_t)DW_NUM_DOF * e + l] = B.nominal_armature[l] * ((C.dr_armature_range[1] - C.dr_armature_range[0]) * u + C.dr_armature_range[0]);
            }
        }
    });
    W.par([&](int l) DWA_INL {
        const float *r = S.root, *ds = S.ds;
        if (l < 2) {          // the rigid-body rows of the new state
            float p[3];
            body_position(S.LM, r, ds, 0, l == 0 ? 6 : 12, p);
            for (int i = 0; i < 3; ++i) {
                S.foot[3 * l + i] = p[i];
                B.foot_pos[((size_t)2 * e + l) * 3 + i] = p[i];
                B.rigid_body_pos[((size_t)DW_NUM_BODIES * e + (l == 0 ? 8 : 16)) * 3 + i] = p[i];
            }
        } else if (l == 2) {
            observations_row(r, &S.small[SO_NZ], &S.small[SO_QN], &S.small[SO_BIAS], &S.small[SO_QB], &S.small[SO_QV], &S.small[SO_CMD], S.obs);
        }
        if (l >= 4 && l < 7) B.rigid_body_pos[(size_t)DW_NUM_BODIES * 3 * e + (l - 4)] = r[l - 4];
        if (l >= 8 && l < 12) B.rigid_body_rot[(size_t)DW_NUM_BODIES * 4 * e + (l - 8)] = r[3 + (l - 8)];
        if (l == 63 && S.touch) B.randomize_buf[e] = 0;
        // ---- the new episode's buffers (:296-297 and what follows): region (1) has read what the reset observation shows of the old ones
        for (int i = l; i < NH * DW_AMP_NUM_OBS1; i += 64) B.obs_history[(size_t)NH * DW_AMP_NUM_OBS1 * e + i] = 0.0f;
        for (int i = l; i < NH * 12; i += 64) B.action_history[(size_t)NH * 12 * e + i] = 0.0f;
        for (int i = l; i < C.log_slots * 12; i += 64) B.action_log[(size_t)C.log_slots * 12 * e + i] = 0.0f;
        if (l < DW_NUM_DOF) {
            const size_t g = (size_t)DW_NUM_DOF * e + l;
            const float q0 = S.ds[2 * l];
            B.dof_vel_pre[g] = 0.0f; B.qpos_noise[g] = q0; B.qpos_pre[g] = q0; B.qvel_noise[g] = 0.0f;
        }
        if (l < 12) {
            B.actions_pre[12 * (size_t)e + l] = 0.0f;
            float v = 0.0f;
            if (C.noise) {
                const float u = R.qb ? R.qb[12 * R.row + l] : u32_uniform(word(DS_RESET, 12 + l));
                const float x = u * 6.28f;
                v = (C.gpu_div ? x * (1.0f / 100.0f) : x / 100.0f) - (float)(3.14 / 100);
            }
            B.qpos_bias[12 * (size_t)e + l] = v;
        }
        if (l < 3) {
            float u;
            if (l == 0) u = R.cx ? R.cx[R.row] : u32_uniform(word(DS_RESET, 24));
            else if (l == 1) u = R.cy ? R.cy[R.row] : u32_uniform(word(DS_RESET, 25));
            else u = R.cyaw ? R.cyaw[R.row] : u32_uniform(word(DS_RESET, 26));
            B.commands[3 * (size_t)e + l] = C.cmd_scale[l] * u + C.cmd_lo[l];
            float v = 0.0f;
            if (C.noise) {
                const float uq = R.quatb ? R.quatb[3 * R.row + l] : u32_uniform(word(DS_RESET, 28 + l));
                const float x = uq * 6.28f;
                v = (C.gpu_div ? x * (1.0f / 150.0f) : x / 150.0f) - (float)(3.14 / 150);
            }
            B.quat_bias[3 * (size_t)e + l] = v;
        }
        if (l == 63) {
            B.progress_buf[e] = 0; B.reset_buf[e] = 0; B.terminate_buf[e] = 0;
            B.epi_len_log[e] = S.small[SO_EPI]; B.epi_len[e] = 0.0f;
            B.perturbation_count[e] = 0; B.pert_on[e] = 0;
            B.perturb_timing[e] = R.ptime ? R.ptime[R.row] : u32_int(word(DS_RESET, 32), 0, (long long)(8 / 0.002));
            B.delay_idx[e] = R.didx ? R.didx[R.row] : u32_int(word(DS_RESET, 33), C.delay_idx_range[0], C.delay_idx_range[1]);
            B.simul_len[e] = 0;
            if (C.hist_ring) { B.hist_head[2 * (size_t)e] = 0; B.hist_head[2 * (size_t)e + 1] = 0; }
        }
    });
    W.par([&](int l) DWA_INL {
        if (l == 1) disc_observations_row(S.root, S.ds, S.ds + 1, 2, C.local_root_obs, S.foot, 2, S.amp);
        if (l >= 2 && l < 2 + DW_AMP_NUM_OBS1) B.obs1[DW_AMP_NUM_OBS1 * (size_t)e + (l - 2)] = S.obs[l - 2];
        // the reset env's observation: every history slot shows the reset observation, the action slots what the action history held
        float *ob = B.obs_buf + (size_t)num_obs * e;
        for (int i = l; i < num_obs; i += 64) {
            const float v = i < obs_part ? S.obs[i % DW_AMP_NUM_OBS1] : S.hist[i - obs_part];
            ob[i] = v;
            if (B.obs_out && R.dr) B.obs_out[(size_t)num_obs * e + i] = fminf(fmaxf(v, -C.clip_obs), C.clip_obs);
        }
    });
    W.par([&](int l) DWA_INL {
        // discriminator history of a default start: every slot the current observation (tasks/tocabi_amp_lower.py:258-272)
        float *ab = B.amp_obs_buf + (size_t)C.amp_steps * AW * e;
        for (int i = l; i < C.amp_steps * AW; i += 64) ab[i] = S.amp[i % AW];
        if (l < AW) B.amp_obs1[(size_t)AW * e + l] = S.amp[l];
        if (dev && l == 0) B.draw_ctr[e] = (long long)((unsigned long long)S.i64[0] + 1ull);
    });
}

}  // namespace dwa

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
