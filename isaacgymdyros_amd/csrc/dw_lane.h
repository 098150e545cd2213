// dw_lane.h -- one physics substep (stand-in for the reference's closed `gym.simulate`, call site
// tasks/dyros_dynamic_walk.py:525) in the LANE layout: one lane = one env, one wavefront = one limb of 64 envs, a workgroup
// of four wavefronts = 64 whole robots (dw_lane_wave.h, roles: dw_lane_model.h).  Same physics, same order of the contact
// iterations and the same written decisions as dw_oct.h / oracle/dw_physics.c (DESIGN.md "Physics model"): Featherstone ABA
// with every spatial quantity in one frame (world axes, reference point O = the base origin at the start of the substep),
// penalty forces for non-sole primitives and capsule proxies, projected Gauss-Seidel on the 8 sole corners in foot-twist
// space, semi-implicit Euler.
//
// What a wave-instruction does here: the same body, for 64 envs.  The body is wave-uniform, so the model's constants are
// scalar operands (no table in LDS, no per-lane constant loads), control flow is scalar, and no lane mirrors another or idles
// because its limb is shorter -- the lane efficiency that the 4- and 8-lanes-per-env generations could not reach.  The price:
// the four limbs of an env are on four waves, so what the quad / octet kernels exchanged with DPP moves crosses through LDS
// behind a workgroup barrier (27 per substep, 20 of them in the Gauss-Seidel sweeps), and a wave is alone on its SIMD (the body
// slots of 64 envs fill the CU's LDS), with 512 registers to itself.
//
// LDS (one workgroup per CU): slot[body 1..33][row 0..3][env] of 16 bytes -- a body's 16 words per env as in dw_oct.h, a row
// of one body is 1 KB of consecutive envs, so every slot access of a wave is one conflict-free ds_read/write_b128 -- 132 KB;
// three exchange records of 28 words per env (21 KB) that double as scratch between their uses; the model's constants as
// packed per-body records (6 KB).
#pragma once

#include "dw_limb.h"
#include "dw_bufg.h"
#include "dw_lane_wave.h"
#include "dw_lane_model.h"

#if defined(__HIPCC__) && !defined(DL_UNROLLED)
#define DL_ROLLED _Pragma("clang loop unroll(disable)")
#else
#define DL_ROLLED
#endif

namespace dwl {

using namespace dw;       // DevModel, PhysParams, small vector helpers
using dwq::F4; using dwq::mk4; using dwq::f2i; using dwq::sincos_fast; using dwq::qmul; using dwq::over_1n;
using dwq::geom_force; using dwq::rigid_inertia; using dwq::add_rigid; using dwq::seg_dist2_fast;
using dwq::rcp_fast;

constexpr int EPW = 64;                      // envs per workgroup = lanes per wave
constexpr int NT = 64 * NWAVE;               // threads per workgroup
constexpr int XCH_ROWS = 7;                  // an exchange record: 28 words per env
constexpr int SC_PARK_WORDS = 12;            // per (env, pair): moment on A, force on A, moment on B (about O), padded to 16-byte pieces
constexpr int XA = 0, XB = 1, XC = 2;

struct alignas(16) LLds {
    F4   slot[NB - 1][4][EPW];               // body b at slot[b - 1]
    F4   xch[3][XCH_ROWS][EPW];              // exchange records; between their uses also: per-env scratch of the task phases (B), the
                                             // trunk joints' inputs (A, during the kinematics pass), self-collision hit masks (C)
    LHot hot;                                // the model's per-body constants (dw_lane_model.h), staged once per kernel
    int  wflag[NWAVE][4];                    // per wave: [0] some env has a touching pair, [1] the wave's hand-off counter (contact phase)
};
static_assert(sizeof(LLds) <= 163840, "LLds must fit the 160 KB of LDS of one CU");

#define DL_SL(b, r) L.slot[(b) - 1][(r)][X.ln]
// Profiling builds (-DDL_STAMPS, tools/lane_stamps.py) record the clock at phase boundaries of every wave of workgroup 0 into the
// first rows of stacked_rewards (dw_simulate leaves that buffer alone).  Never defined in the shipped library.
#if defined(DL_STAMPS) && defined(__HIPCC__) && defined(DL_SUB_BASE)
// (substep stamps of the fused step kernel: window DL_SUB_BASE .. +13 into the tail of gate_acc, as DL_STAMP2 below)
#define DL_STAMP(n) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (n) >= DL_SUB_BASE && (n) < DL_SUB_BASE + 14) B.cold->gate_acc[200 + (threadIdx.x >> 6) * 14 + (n) - DL_SUB_BASE] = (long long)(__builtin_readcyclecounter() - dl_t0); } while (0)
#define DL_STAMP_T0() const unsigned long long dl_t0 = __builtin_readcyclecounter()
#define DL_STAMP2(n) do { } while (0)
#define DL_T0 dl_t0
#elif defined(DL_STAMPS) && defined(__HIPCC__)
#define DL_STAMP(n) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) B.cold->stacked_rewards[X.w * 32 + (n)] = (float)(long long)(__builtin_readcyclecounter() - dl_t0); } while (0)
#define DL_STAMP_T0() const unsigned long long dl_t0 = __builtin_readcyclecounter()
// step-level stamps (tools/lane_stamps.py --step): 14 per wave in the free tail of gate_acc (words 200 ..), the window of stamp
// numbers chosen at build time (-DDL_STAMP2_BASE=n)
#if !defined(DL_STAMP2_BASE)
#define DL_STAMP2_BASE 0
#endif
#define DL_STAMP2(n) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (n) >= DL_STAMP2_BASE && (n) < DL_STAMP2_BASE + 14) B.cold->gate_acc[200 + (threadIdx.x >> 6) * 14 + (n) - DL_STAMP2_BASE] = (long long)(__builtin_readcyclecounter() - dl_t0); } while (0)
#define DL_T0 dl_t0
#else
#define DL_STAMP(n) do { } while (0)
#define DL_STAMP_T0() do { } while (0)
#define DL_STAMP2(n) do { } while (0)
#define DL_T0 0ull
#endif
#if defined(__HIPCC__)
#define DL_KEEP1(a) asm volatile("" : "+v"((a).w))
#define DL_KEEP2(a, b) asm volatile("" : "+v"((a).w), "+v"((b).w))
#else
#define DL_KEEP1(a) ((void)0)
#define DL_KEEP2(a, b) ((void)0)
#endif

// What a lane keeps in registers across the phases of a step.  Every wave holds the base state of its envs (the same
// arithmetic on the same inputs in all four: bit-identical copies); the leg waves also hold their sole's corners.
struct LState {
    int   ln, w, e, valid;
    float root[13];
    float mu;
    float zbound;                            // height field: dw_physics.h terrain_bound of my env's base position
    float rk[4][3], vmin[4], frame[4][9];    // leg waves: corners of my sole relative to O, velocity bounds, contact frames (terrain)
    int   act[4];
    int   coll;                              // last substep: one of my non-sole Gym bodies reports more than 1 N (termination)
    float footT[3];                          // last substep: net contact force on my sole's Gym body (leg waves)
    int   seq;                               // leg waves: hand-offs made so far (flag_post / flag_wait, dw_lane_wave.h)
    float warm[12];                          // leg waves: impulses of my sole's corners, carried from substep to substep (the caller
                                             // loads them from the task record, DW_ES_WARM, and stores them back)
};

DQ_HD void add_rigid_inertia(float *IA, const float *Ao, const float *ho, float mass) {
    DQ_UNROLL for (int r = 0; r < 3; ++r)
        DQ_UNROLL for (int c = r; c < 3; ++c) IA[sym6(r, c)] += dwq::ao(Ao, r, c);
    IA[sym6(0, 4)] += -ho[2]; IA[sym6(0, 5)] += ho[1];
    IA[sym6(1, 3)] += ho[2];  IA[sym6(1, 5)] += -ho[0];
    IA[sym6(2, 3)] += -ho[1]; IA[sym6(2, 4)] += ho[0];
    IA[sym6(3, 3)] += mass; IA[sym6(4, 4)] += mass; IA[sym6(5, 5)] += mass;
}
DQ_HD void rigid_bias(const float *Ao, const float *ho, float mass, const float *v, float *pv) {
    const float *om = v, *vl = v + 3;
    float n[3], f[3], t1[3], t2[3];
    DQ_UNROLL for (int r = 0; r < 3; ++r) n[r] = dwq::ao(Ao, r, 0) * om[0] + dwq::ao(Ao, r, 1) * om[1] + dwq::ao(Ao, r, 2) * om[2];
    cross3(ho, vl, t1);
    n[0] += t1[0]; n[1] += t1[1]; n[2] += t1[2];
    cross3(om, ho, t1);
    f[0] = t1[0] + mass * vl[0]; f[1] = t1[1] + mass * vl[1]; f[2] = t1[2] + mass * vl[2];
    cross3(om, n, t1); cross3(vl, f, t2);
    pv[0] = t1[0] + t2[0]; pv[1] = t1[1] + t2[1]; pv[2] = t1[2] + t2[2];
    cross3(om, f, t1);
    pv[3] = t1[0]; pv[4] = t1[1]; pv[5] = t1[2];
}

// Closest points of two segments and the penalty force of two capsules as dw_limb.h's seg_seg / capsule_pair (written decision:
// oracle/dw_physics.c seg_seg), with the quotients as Newton-refined reciprocals (~1 ulp) instead of IEEE divisions: ten
// divisions are a third of the evaluation, and with 64 envs per wave some env has a touching pair in most substeps.
DQ_HD void seg_seg_l(const float *da, const float *db, const float *r, float *so, float *to) {
    const float aa = dot3(da, da), ee = dot3(db, db), ff = dot3(db, r), eps = 1e-12f;
    float sa, sb;
    auto c01 = [](float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); };
    if (aa <= eps && ee <= eps) { sa = 0.0f; sb = 0.0f; }
    else if (aa <= eps) { sa = 0.0f; sb = c01(ff * dw::rcp_nr(ee)); }
    else {
        const float cc = dot3(da, r), ia = dw::rcp_nr(aa);
        if (ee <= eps) { sb = 0.0f; sa = c01(-cc * ia); }
        else {
            const float ie = dw::rcp_nr(ee);
            const float bbv = dot3(da, db), den = aa * ee - bbv * bbv;
            float se = den > eps ? c01((bbv * ff - cc * ee) * dw::rcp_nr(den)) : 0.0f;
            float te = (bbv * se + ff) * ie;
            if (te < 0.0f) { te = 0.0f; se = c01(-cc * ia); }
            else if (te > 1.0f) { te = 1.0f; se = c01((bbv - cc) * ia); }
            const float t0 = -cc * ia, t1 = t0 + bbv * ia;
            float lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
            if (lo < 0.0f) lo = 0.0f;
            if (hi > 1.0f) hi = 1.0f;
            const float sp = c01(0.5f * (lo + hi)), tp = c01((bbv * sp + ff) * ie), reg = 1e-3f * aa * ee;
            const float w = den > eps ? den * den * dw::rcp_nr(den * den + reg * reg) : 0.0f;
            sa = w * se + (1.0f - w) * sp;
            sb = w * te + (1.0f - w) * tp;
        }
    }
    *so = sa; *to = sb;
}
DQ_HD bool capsule_pair_l(const float *a0, const float *a1, float ra, const float *b0, const float *b1, float rb, const float *va,
                          const float *vb, const PhysParams &P, float *F, float *pa, float *pb) {
    const float da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
    const float r[3] = {a0[0] - b0[0], a0[1] - b0[1], a0[2] - b0[2]};
    float sa, sb;
    seg_seg_l(da, db, r, &sa, &sb);
    float n[3];
    DQ_UNROLL for (int i = 0; i < 3; ++i) { pa[i] = a0[i] + sa * da[i]; pb[i] = b0[i] + sb * db[i]; n[i] = pa[i] - pb[i]; }
    const float d2 = dot3(n, n);
    const float dist = sqrtf(d2);
    const float depth = ra + rb - dist;
    if (!(depth > 0.0f && dist > 1e-6f)) return false;
    const float idist = dw::rcp_nr(dist);
    DQ_UNROLL for (int i = 0; i < 3; ++i) n[i] *= idist;
    float ta[3], tb[3];
    cross3(va, pa, ta);
    cross3(vb, pb, tb);
    float vn = 0.0f;
    DQ_UNROLL for (int i = 0; i < 3; ++i) vn += ((va[3 + i] + ta[i]) - (vb[3 + i] + tb[i])) * n[i];
    float fn = P.pen_k * depth - P.pen_c * vn;
    if (fn < 0.0f) fn = 0.0f;
    DQ_UNROLL for (int i = 0; i < 3; ++i) F[i] = fn * n[i];
    return true;
}

// a 27-word record (symmetric 6x6 + 6-vector) to / from an exchange buffer
DQ_HD void xch_put(LLds &L, int buf, int ln, const float *IA, const float *pA) {
    DQ_UNROLL for (int r = 0; r < 5; ++r) L.xch[buf][r][ln] = mk4(IA[4 * r], IA[4 * r + 1], IA[4 * r + 2], IA[4 * r + 3]);
    L.xch[buf][5][ln] = mk4(IA[20], pA[0], pA[1], pA[2]);
    L.xch[buf][6][ln] = mk4(pA[3], pA[4], pA[5], 0.0f);
}
DQ_HD void xch_add(const LLds &L, int buf, int ln, float *IA, float *pA) {
    DQ_UNROLL for (int r = 0; r < 5; ++r) { const F4 t = L.xch[buf][r][ln]; IA[4 * r] += t.x; IA[4 * r + 1] += t.y; IA[4 * r + 2] += t.z; IA[4 * r + 3] += t.w; }
    const F4 a = L.xch[buf][5][ln], b = L.xch[buf][6][ln];
    IA[20] += a.x; pA[0] += a.y; pA[1] += a.z; pA[2] += a.w; pA[3] += b.x; pA[4] += b.y; pA[5] += b.z;
}

// ------------------------------------------------------------------------------------------------
// The substep.  On entry every body's slot holds row 0 = {q, qd, tt, dd} with tt = tau - damping * qd and
// dd = armature + dt * damping (the caller's prologue), the trunk bodies' (q, qd) also lie in exchange record A (two joints per
// row: the waves that recompute the trunk read them there, after wave 2 has overwritten the slot), X.root holds the base state.
// On exit: row 2 .w of every body = the new joint velocity (before the joint-range clamp of the integrator), X.root updated;
// with `last`, the net contact forces of the substep are in B.contact_forces (the caller has zero-filled the envs' rows),
// X.coll / X.footT set.  The warm-start impulses of the sole corners travel in X.warm.
// ------------------------------------------------------------------------------------------------
// What waves 2 and 3 do while waves 0 and 1 solve the contacts (they would otherwise only count barriers): work() before the
// first barrier of the phase, publish() after the last one (exchange record C is free then; the next barrier hands it over).
struct NoIdleWork { DL_MEM void work() {} DL_MEM void publish() {} };

template <bool TERRAIN, class IDLE>
DQ_HD void lane_substep(LLds &L, const LaneModel &LM, const DevModel &M, const PhysParams &P, LState &X, const OBuf &B,
                         float push_x, float push_y, bool last, IDLE &idle) {
    const float dt = P.dt, inv_dt = 1.0f / P.dt;
    const int w = X.w, ln = X.ln, e = X.e;
    const float *mscale_e = B.mass_scale + (size_t)DW_NUM_BODIES * e;
    const bool legwave = w < 2;
    float c_offset = P.contact_offset, c_erp = P.erp, c_maxdep = P.max_depen;          // (launch-invariant: keep them in scalar registers)
    int c_selfcoll = P.self_collision;
    DQ_SGPR_KEEP(c_offset); DQ_SGPR_KEEP(c_erp); DQ_SGPR_KEEP(c_maxdep); DQ_SGPR_KEEP(c_selfcoll);
    DL_STAMP_T0();
    DL_STAMP(0);
    if (TERRAIN) X.zbound = dw::terrain_bound(P, X.root[0], X.root[1]);          // (requested here, first used in the inward pass)

    // ---- base kinematics (every wave, redundantly) ----
    float qn[4], R0[9], ww[3], vo[3], bcom[3];
    {
        const float qx = X.root[3], qy = X.root[4], qz = X.root[5], qw = X.root[6];
        const float ninv = dw::rsqrt_nr(qx * qx + qy * qy + qz * qz + qw * qw);
        qn[0] = qx * ninv; qn[1] = qy * ninv; qn[2] = qz * ninv; qn[3] = qw * ninv;
        quat_to_mat(qn, R0);
        DQ_UNROLL for (int i = 0; i < 3; ++i) { ww[i] = X.root[10 + i]; vo[i] = X.root[7 + i]; bcom[i] = M.bi_com[0][0][i]; }
        if (P.vel_at_com) {
            float rc[3], t[3];
            m3v(R0, bcom, rc);
            cross3(ww, rc, t);
            vo[0] -= t[0]; vo[1] -= t[1]; vo[2] -= t[2];
        }
    }

    // ---- control words of my role, in scalar registers for the whole substep ----
    const LCtl &CT = LM.ctl[w];
    const unsigned ob0 = CT.out_b[0], ob1 = CT.out_b[1], ob2 = CT.out_b[2];
    const unsigned m_start = CT.out_start, m_store = CT.out_store, m_sole = CT.out_sole, m_shared = CT.out_shared, m_q0 = CT.out_q0, m_ts = CT.out_ts;
    const int n_out = CT.n_out;
    auto out_body = [&](int k) { const unsigned wd = k < 4 ? ob0 : (k < 8 ? ob1 : ob2); return (int)((wd >> (8 * (k & 3))) & 255u); };
    const LFkRec *fkt = L.hot.fk + CT.fk_off;

    // ---- outward pass 1: kinematics along my bodies (running parent state: quaternion, rotation, origin, twist).  The next
    //      body's record and joint inputs are requested while this body is computed. ----
    auto fk_fetch = [&](int k, F4 &r0, F4 &r1, F4 &in) {          // record k of my list and the body's row 0 ({q, qd, tt, dd}, or the trunk copy)
        const F4 *rp = reinterpret_cast<const F4 *>(fkt + k);
        r0 = rp[0]; r1 = rp[1];
        const int b = out_body(k);
        if ((m_shared >> k) & 1u) {
            const int ts = (int)((m_ts >> (2 * k)) & 3u);
            const F4 t = L.xch[XA][ts >> 1][ln];
            in = mk4((ts & 1) ? t.z : t.x, (ts & 1) ? t.w : t.y, 0.0f, 0.0f);
            if ((m_store >> k) & 1u) { const F4 o = DL_SL(b, 0); in.z = o.z; in.w = o.w; }
        } else in = DL_SL(b, 0);
    };
    {
        float qr[4] = {0, 0, 0, 1}, Rr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, xr[3] = {0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
        F4 c0, c1, cin;
        fk_fetch(0, c0, c1, cin);
        DL_ROLLED for (int k = 0; k < n_out; ++k) {
            const F4 r0 = c0, r1 = c1, in = cin;
            fk_fetch(k + 1 < n_out ? k + 1 : k, c0, c1, cin);
            const int b = out_body(k);
            if ((m_start >> k) & 1u) {
                DQ_UNROLL for (int i = 0; i < 4; ++i) qr[i] = qn[i];
                DQ_UNROLL for (int i = 0; i < 9; ++i) Rr[i] = R0[i];
                DQ_UNROLL for (int i = 0; i < 3; ++i) { xr[i] = 0.0f; vr[i] = ww[i]; vr[3 + i] = vo[i]; }
            }
            const float q = in.x, qd = in.y, tt = in.z, dd = in.w;
            const float axis[3] = {r1.x, r1.y, r1.z}, pos[3] = {r0.x, r0.y, r0.z};
            float sn, cs;
            sincos_fast(0.5f * q, &sn, &cs);
            float qj[4] = {axis[0] * sn, axis[1] * sn, axis[2] * sn, cs};
            const int q0i = (int)((m_q0 >> (2 * k)) & 3u);
            if (q0i) { const F4 r2 = *reinterpret_cast<const F4 *>(L.hot.q0[q0i - 1]); const float q0[4] = {r2.x, r2.y, r2.z, r2.w}; qmul(q0, qj, qj); }
            float x[3], t[3];
            m3v(Rr, pos, t);
            DQ_UNROLL for (int i = 0; i < 3; ++i) x[i] = xr[i] + t[i];
            qmul(qr, qj, qr);
            quat_to_mat(qr, Rr);
            float aw[3], sl[3];
            m3v(Rr, axis, aw);
            cross3(x, aw, sl);
            DQ_UNROLL for (int i = 0; i < 3; ++i) { vr[i] += aw[i] * qd; vr[3 + i] += sl[i] * qd; xr[i] = x[i]; }
            if ((m_store >> k) & 1u) {
                DL_SL(b, 0) = mk4(qr[0], qr[1], qr[2], qr[3]);
                DL_SL(b, 1) = mk4(x[0], x[1], x[2], qd);
                DL_SL(b, 2) = mk4(vr[0], vr[1], vr[2], tt);
                DL_SL(b, 3) = mk4(vr[3], vr[4], vr[5], dd);
            }
            if ((m_sole >> k) & 1u) {
                // the four corners of my sole: lever arms, gaps, velocity bounds -- kept in registers for the contact phase
                const int f = w;
                DQ_UNROLL for (int c = 0; c < 4; ++c) {
                    const F4 fp4 = *reinterpret_cast<const F4 *>(L.hot.foot[4 * f + c]);
                    const float fp[3] = {fp4.x, fp4.y, fp4.z};
                    float r[3];
                    m3v(Rr, fp, r);
                    DQ_UNROLL for (int i = 0; i < 3; ++i) r[i] += x[i];
                    float phi = X.root[2] + r[2];
                    if (TERRAIN) {
                        float hh;
                        dw::terrain_sample(P, X.root[0] + r[0], X.root[1] + r[1], &hh, X.frame[c]);
                        phi = (phi - hh) * X.frame[c][8];
                    }
                    X.act[c] = phi < c_offset;
                    DQ_UNROLL for (int i = 0; i < 3; ++i) X.rk[c][i] = r[i];
                    X.vmin[c] = phi >= 0 ? -phi * inv_dt : fminf(c_erp * (-phi) * inv_dt, c_maxdep);
                }
            }
        }
    }
    DL_STAMP(1);
    wg_barrier();
    DL_STAMP(2);

    // ---- self-collision: capsule proxies, pairs from the model.  Wave w tests a quarter of the pairs for its 64 envs (both
    //      proxies' axes from their bodies' slots, the division-free conservative distance of dw_limb.h).  The common case is
    //      "nothing touches": then that is all.  A pair that touches in some env of the wave is resolved on the spot, by the
    //      lanes it touches in -- the exact closest points, the penalty force, the wrench on either body -- and parked in
    //      global memory, one 16-word record per (env, pair); the wave that owns a body picks its side up in the inward pass. ----
    unsigned sc_hits = 0;
    bool sc_wg = false;
    float *park = P.sc_park + (size_t)e * DW_MAX_SC_PAIRS * SC_PARK_WORDS;
    // every wave's FIRST touching pair of an env crosses in LDS (12 words + the pair and the two proxies' body / Gym-slot bits: rows
    // 3 (w & 1) .. of exchange record A (waves 0, 1) or B (waves 2, 3), free at this point); a second one of the same wave and env
    // (rare) goes through the global park.  fh: what every wave has read of the four slots after the barrier.
    F4 fh[NWAVE][3];
    DQ_UNROLL for (int sw = 0; sw < NWAVE; ++sw) { fh[sw][0] = fh[sw][1] = mk4(0.0f, 0.0f, 0.0f, 0.0f); fh[sw][2] = mk4(0.0f, __builtin_bit_cast(float, -1), 0.0f, 0.0f); }
    if (c_selfcoll && LM.npair > 0) {
        unsigned hits = 0;
        F4 my0 = mk4(0.0f, 0.0f, 0.0f, 0.0f), my1 = my0, my2 = mk4(0.0f, __builtin_bit_cast(float, -1), 0.0f, 0.0f);
        const unsigned pba0 = CT.pair_ba[0], pba1 = CT.pair_ba[1], pbb0 = CT.pair_bb[0], pbb1 = CT.pair_bb[1];
        const unsigned ppa0 = CT.pair_pa[0], ppa1 = CT.pair_pa[1], ppb0 = CT.pair_pb[0], ppb1 = CT.pair_pb[1];
        const int k_n = CT.pair_n;
        // (the two proxies' records and both bodies' pose rows are requested a pair ahead)
        auto pair_fetch = [&](int i, F4 (&pr)[4], F4 &qa, F4 &xa, F4 &qb, F4 &xb) {
            const int pa = (int)(((i < 4 ? ppa0 : ppa1) >> (8 * (i & 3))) & 255u), pb = (int)(((i < 4 ? ppb0 : ppb1) >> (8 * (i & 3))) & 255u);
            const F4 *ra = reinterpret_cast<const F4 *>(L.hot.prox + pa), *rb = reinterpret_cast<const F4 *>(L.hot.prox + pb);
            pr[0] = ra[0]; pr[1] = ra[1]; pr[2] = rb[0]; pr[3] = rb[1];
            const int ba = (int)(((i < 4 ? pba0 : pba1) >> (8 * (i & 3))) & 255u), bb = (int)(((i < 4 ? pbb0 : pbb1) >> (8 * (i & 3))) & 255u);
            qa = L.slot[ba - 1][0][ln]; xa = L.slot[ba - 1][1][ln];
            qb = L.slot[bb - 1][0][ln]; xb = L.slot[bb - 1][1][ln];
        };
        auto ends = [&](const F4 &q4, const F4 &x4, const float *l0, const float *l1, float *p0w, float *p1w) {
            const float qb[4] = {q4.x, q4.y, q4.z, q4.w};
            float Rb[9], t0[3], t1[3];
            quat_to_mat(qb, Rb);
            m3v(Rb, l0, t0);
            m3v(Rb, l1, t1);
            p0w[0] = x4.x + t0[0]; p0w[1] = x4.y + t0[1]; p0w[2] = x4.z + t0[2];
            p1w[0] = x4.x + t1[0]; p1w[1] = x4.y + t1[1]; p1w[2] = x4.z + t1[2];
        };
        F4 npr[4], nqa, nxa, nqb, nxb;
        if (k_n > 0) pair_fetch(0, npr, nqa, nxa, nqb, nxb);
        DL_ROLLED for (int i = 0; i < k_n; ++i) {
            const F4 p0 = npr[0], p1 = npr[1], p2 = npr[2], p3 = npr[3], qa = nqa, qb = nqb; F4 xa = nxa, xb = nxb;
            DL_KEEP2(xa, xb);
            pair_fetch(i + 1 < k_n ? i + 1 : i, npr, nqa, nxa, nqb, nxb);
            const int k = w + NWAVE * i;
            // proxy record words: p0.xyz radius | p1.xyz bits
            const float la0[3] = {p0.x, p0.y, p0.z}, la1[3] = {p1.x, p1.y, p1.z}, lb0[3] = {p2.x, p2.y, p2.z}, lb1[3] = {p3.x, p3.y, p3.z};
            float a0[3], a1[3], b0[3], b1[3];
            ends(qa, xa, la0, la1, a0, a1);
            ends(qb, xb, lb0, lb1, b0, b1);
            const float da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
            const float rv[3] = {a0[0] - b0[0], a0[1] - b0[1], a0[2] - b0[2]};
            const float rr = p0.w + p2.w;
            // conservative: the least distance of the axes (the force uses the blended points, never closer), 0.2 % slack
            const bool near = seg_dist2_fast(da, db, rv) < 1.004f * rr * rr;
            if (wave_any(near)) {
                const int ba = (int)(((i < 4 ? pba0 : pba1) >> (8 * (i & 3))) & 255u), bb = (int)(((i < 4 ? pbb0 : pbb1) >> (8 * (i & 3))) & 255u);
                if (near) {
                    F4 va2 = L.slot[ba - 1][2][ln], va3 = L.slot[ba - 1][3][ln], vb2 = L.slot[bb - 1][2][ln], vb3 = L.slot[bb - 1][3][ln];
                    DL_KEEP2(va2, va3); DL_KEEP2(vb2, vb3);
                    const float va[6] = {va2.x, va2.y, va2.z, va3.x, va3.y, va3.z}, vb[6] = {vb2.x, vb2.y, vb2.z, vb3.x, vb3.y, vb3.z};
                    float F[3], ca[3], cb[3];
                    if (capsule_pair_l(a0, a1, p0.w, b0, b1, p2.w, va, vb, P, F, ca, cb)) {
                        float na[3], nb[3];
                        const float Fm[3] = {-F[0], -F[1], -F[2]};
                        cross3(ca, F, na);
                        cross3(cb, Fm, nb);
                        if (hits == 0) {
                            my0 = mk4(na[0], na[1], na[2], F[0]); my1 = mk4(F[1], F[2], nb[0], nb[1]);
                            my2 = mk4(nb[2], __builtin_bit_cast(float, k), p1.w, p3.w);
                        } else {
                            F4 *pk = reinterpret_cast<F4 *>(park + SC_PARK_WORDS * k);
                            pk[0] = mk4(na[0], na[1], na[2], F[0]);
                            pk[1] = mk4(F[1], F[2], nb[0], nb[1]);
                            pk[2] = mk4(nb[2], 0.0f, 0.0f, 0.0f);
                        }
                        hits |= 1u << k;
                    }
                }
            }
        }
        {
            const int xb_ = w < 2 ? XA : XB, r0_ = 3 * (w & 1);
            L.xch[xb_][r0_][ln] = my0; L.xch[xb_][r0_ + 1][ln] = my1; L.xch[xb_][r0_ + 2][ln] = my2;
        }
        reinterpret_cast<int *>(&L.xch[XC][0][ln])[w] = (int)hits;
        const bool any = wave_any(hits != 0);
        if (ln == 0) L.wflag[w][0] = any ? 1 : 0;
        wg_barrier_global();
        sc_wg = (L.wflag[0][0] | L.wflag[1][0] | L.wflag[2][0] | L.wflag[3][0]) != 0;
        if (sc_wg) {
            const F4 hv = L.xch[XC][0][ln];
            sc_hits = (unsigned)(f2i(hv.x) | f2i(hv.y) | f2i(hv.z) | f2i(hv.w));
            DQ_UNROLL for (int sw = 0; sw < NWAVE; ++sw)
                DQ_UNROLL for (int r = 0; r < 3; ++r) fh[sw][r] = L.xch[sw < 2 ? XA : XB][3 * (sw & 1) + r][ln];
            wg_barrier();          // (every wave holds the slots: the exchange records may be written again)
        }
    }
    DL_STAMP(3);

    // ---- inward pass: articulated inertias and bias forces along my chains, tip -> root.  Per body: what it contributes by
    //      itself (joint subspace, rigid inertia about O, gyroscopic bias, external forces, velocity-product acceleration), then the
    //      recursion (add into the running inertia, U = IA S, rank-1 downdate, bias). ----
    float IA[21], pA[6];
    DQ_UNROLL for (int i = 0; i < 21; ++i) IA[i] = 0.0f;
    DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] = 0.0f;
    float footF[3] = {0, 0, 0};
    const int my_sole_gym = w == 0 ? M.left_foot_gym : (w == 1 ? M.right_foot_gym : -1);
    // contact report of one moving body: per Gym body welded into it
    auto report = [&](int b, float (&cf)[LMAX_GYM][3]) {
        DQ_UNROLL for (int t = 0; t < LMAX_GYM; ++t)
            if (t < LM.ngym[b]) {
                const int gy = LM.gyms[b][t];
                if (gy == my_sole_gym) { footF[0] = cf[t][0]; footF[1] = cf[t][1]; footF[2] = cf[t][2]; }
                else {
                    if (over_1n(cf[t])) X.coll = 1;
                    if (X.valid && (cf[t][0] != 0.0f || cf[t][1] != 0.0f || cf[t][2] != 0.0f)) {
                        float *dst = B.contact_forces + ((size_t)DW_NUM_BODIES * e + gy) * 3;
                        dst[0] = cf[t][0]; dst[1] = cf[t][1]; dst[2] = cf[t][2];
                    }
                }
            }
    };
    auto inward_chain = [&](int ch) {
        const int n = CT.n_in[ch];
        const unsigned ib0 = CT.in_b[ch][0], ib1 = CT.in_b[ch][1], ig0 = CT.in_gym[ch][0], ig1 = CT.in_gym[ch][1];
        const unsigned m_two = CT.in_two[ch], m_geom = CT.in_geom[ch], m_prox = CT.in_prox[ch];
        const LInRec *tab = L.hot.in + CT.in_off[ch];
        auto in_body = [&](int k) { return (int)(((k < 4 ? ib0 : ib1) >> (8 * (k & 3))) & 255u); };
        // (a body's record, its slot rows and its mass scale are requested while the body before it is computed)
        auto in_fetch = [&](int k, F4 (&rc)[4], F4 (&sr)[4], float &ms) {
            const F4 *rp = reinterpret_cast<const F4 *>(tab + k);
            rc[0] = rp[0]; rc[1] = rp[1]; rc[2] = rp[2]; rc[3] = rp[3];
            const int b = in_body(k);
            sr[0] = DL_SL(b, 0); sr[1] = DL_SL(b, 1); sr[2] = DL_SL(b, 2); sr[3] = DL_SL(b, 3);
            ms = mscale_e[(int)(((k < 4 ? ig0 : ig1) >> (8 * (k & 3))) & 255u)];
        };
        F4 nrc[4], nsr[4];
        float nms = 0.0f;
        if (n > 0) in_fetch(0, nrc, nsr, nms);
        DL_ROLLED for (int k = 0; k < n; ++k) {
            const F4 rc0 = nrc[0], rc1 = nrc[1], rc2 = nrc[2], rc3 = nrc[3], s0 = nsr[0], s1 = nsr[1], s2 = nsr[2], s3 = nsr[3];
            const float ms0 = nms;
            in_fetch(k + 1 < n ? k + 1 : k, nrc, nsr, nms);
            // record words: com.xyz mass | I0..I3 | I4 I5 bound bits | axis.xyz pairs
            const int b = in_body(k);
            const float axis[3] = {rc3.x, rc3.y, rc3.z};
            const float qb[4] = {s0.x, s0.y, s0.z, s0.w}, x[3] = {s1.x, s1.y, s1.z}, v[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
            const float qd = s1.w, tt = s2.w, dd = s3.w;
            float R[9], S[6];
            quat_to_mat(qb, R);
            m3v(R, axis, S);
            cross3(x, S, S + 3);
            float Ao[6], ho[3], mass;
            {
                const float com0[3] = {rc0.x, rc0.y, rc0.z};
                const float I0[6] = {rc1.x, rc1.y, rc1.z, rc1.w, rc2.x, rc2.y};
                if ((m_two >> k) & 1u) {          // the two sole bodies carry a second (welded) inertial record
                    const F4 *ip = reinterpret_cast<const F4 *>(&L.hot.in1[w & 1]);
                    const F4 i0 = ip[0], i1 = ip[1], i2 = ip[2];
                    const float ms1 = mscale_e[f2i(i2.z)];
                    const float com1[3] = {i0.x, i0.y, i0.z};
                    const float I1[6] = {i1.x, i1.y, i1.z, i1.w, i2.x, i2.y};
                    rigid_inertia(2, com0, rc0.w, I0, ms0, com1, i0.w, I1, ms1, R, x, Ao, ho, &mass);
                } else {
                    rigid_inertia(1, com0, rc0.w, I0, ms0, com0, 0.0f, I0, 0.0f, R, x, Ao, ho, &mass);
                }
            }
            float pv[6], cb[6];
            rigid_bias(Ao, ho, mass, v, pv);
            {
                float m[6];
                DQ_UNROLL for (int i = 0; i < 6; ++i) m[i] = S[i] * qd;
                dw::motion_cross(v, m, cb);
            }
            {
                // external forces: ground penalty of the non-sole primitives, self-collision; per Gym body for the report
                const bool has_geom = ((m_geom >> k) & 1u) != 0;
                bool near_ground = has_geom && (X.root[2] + x[2] < rc2.z);
                if (TERRAIN) near_ground = has_geom && (X.root[2] + x[2] - X.zbound < rc2.z);
                const bool geo = has_geom && wave_any(near_ground);
                const unsigned bpairs = (unsigned)f2i(rc3.w);
                const bool scb = sc_wg && ((m_prox >> k) & 1u) && wave_any((sc_hits & bpairs) != 0);
                if (geo || scb) {
                    float cf[LMAX_GYM][3];
                    DQ_UNROLL for (int t = 0; t < LMAX_GYM; ++t) cf[t][0] = cf[t][1] = cf[t][2] = 0.0f;
                    if (geo) {
                        if (legwave && ch == 0) {
                            // a leg body: its primitives' records are in LDS (the feet are at the ground in every substep)
                            const int ngeom = (int)((CT.in_ng >> (4 * k)) & 15u), g0 = (int)(((k < 4 ? CT.in_g0[0] : CT.in_g0[1]) >> (8 * (k & 3))) & 255u);
                            DL_ROLLED for (int g = 0; g < ngeom; ++g) {
                                const F4 *gp = reinterpret_cast<const F4 *>(L.hot.lgeom + g0 + g);
                                const F4 g0v = gp[0], g1v = gp[1], g2v = gp[2], g3v = gp[3];
                                DwGeom ge;
                                const int gbits = uniform(f2i(g0v.x));
                                ge.type = gbits & 255; ge.moving = b; ge.gym = 0; ge.sole = 0; ge._pad = 0.0f;
                                ge.pos[0] = g0v.y; ge.pos[1] = g0v.z; ge.pos[2] = g0v.w;
                                ge.rot[0] = g1v.x; ge.rot[1] = g1v.y; ge.rot[2] = g1v.z; ge.rot[3] = g1v.w; ge.rot[4] = g2v.x; ge.rot[5] = g2v.y; ge.rot[6] = g2v.z; ge.rot[7] = g2v.w; ge.rot[8] = g3v.x;
                                ge.size[0] = g3v.y; ge.size[1] = g3v.z; ge.size[2] = g3v.w;
                                float F[3] = {0, 0, 0}, xg[3] = {0, 0, 0};
                                if (near_ground) geom_force<TERRAIN>(ge, P, R, x, v, X.root[0], X.root[1], X.root[2], X.mu, F, xg);
                                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                                    float nb[3];
                                    cross3(xg, F, nb);
                                    DQ_UNROLL for (int i = 0; i < 3; ++i) { pv[i] -= nb[i]; pv[3 + i] -= F[i]; }
                                    const int t = (gbits >> 8) & 255;
                                    DQ_UNROLL for (int t2 = 0; t2 < LMAX_GYM; ++t2)
                                        if (t2 == t) { cf[t2][0] += F[0]; cf[t2][1] += F[1]; cf[t2][2] += F[2]; }
                                }
                            }
                        } else {
                            const int ngeom = M.body_ngeom[b];
                            DL_ROLLED for (int g = 0; g < ngeom; ++g) {
                                float F[3] = {0, 0, 0}, xg[3] = {0, 0, 0};
                                if (near_ground) geom_force<TERRAIN>(M.geoms[M.body_geom[b][g]], P, R, x, v, X.root[0], X.root[1], X.root[2], X.mu, F, xg);
                                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                                    float nb[3];
                                    cross3(xg, F, nb);
                                    DQ_UNROLL for (int i = 0; i < 3; ++i) { pv[i] -= nb[i]; pv[3 + i] -= F[i]; }
                                    const int t = LM.geom_gslot[b][g];
                                    DQ_UNROLL for (int t2 = 0; t2 < LMAX_GYM; ++t2)
                                        if (t2 == t) { cf[t2][0] += F[0]; cf[t2][1] += F[1]; cf[t2][2] += F[2]; }
                                }
                            }
                        }
                    }
                    if (scb) {
                        // my side of every touching pair that involves a proxy of this body: the waves' first hits (registers), then
                        // what went through the park, in pair order
                        auto apply = [&](bool side_a, int t, const F4 &k0, const F4 &k1, const F4 &k2) {
                            const float W6[6] = {side_a ? k0.x : k1.z, side_a ? k0.y : k1.w, side_a ? k0.z : k2.x,
                                                 side_a ? k0.w : -k0.w, side_a ? k1.x : -k1.x, side_a ? k1.y : -k1.y};
                            DQ_UNROLL for (int i = 0; i < 6; ++i) pv[i] -= W6[i];
                            DQ_UNROLL for (int t2 = 0; t2 < LMAX_GYM; ++t2)
                                if (t2 == t) { cf[t2][0] += W6[3]; cf[t2][1] += W6[4]; cf[t2][2] += W6[5]; }
                        };
                        DQ_UNROLL for (int sw = 0; sw < NWAVE; ++sw) {
                            const int pid = f2i(fh[sw][2].y), ba_ = f2i(fh[sw][2].z), bb_ = f2i(fh[sw][2].w);
                            const bool ona = pid >= 0 && (ba_ & 255) == b, onb = pid >= 0 && (bb_ & 255) == b;
                            if (ona || onb) apply(ona, ((ona ? ba_ : bb_) >> 8) & 255, fh[sw][0], fh[sw][1], fh[sw][2]);
                        }
                        unsigned rest = sc_hits & bpairs;
                        DQ_UNROLL for (int sw = 0; sw < NWAVE; ++sw) { const int pid = f2i(fh[sw][2].y); if (pid >= 0) rest &= ~(1u << pid); }
                        if (wave_any(rest != 0)) {
                            DL_ROLLED for (int kp = 0; kp < LM.npair; ++kp) {
                                if (!((bpairs >> kp) & 1u)) continue;
                                const bool hit = ((rest >> kp) & 1u) != 0;
                                if (!wave_any(hit)) continue;
                                const int rba = L.hot.prox[LM.pair_a[kp]].bits, rbb = L.hot.prox[LM.pair_b[kp]].bits;
                                const bool side_a = (rba & 255) == b;
                                if (hit) {
                                    const F4 *pk = reinterpret_cast<const F4 *>(park + SC_PARK_WORDS * kp);
                                    apply(side_a, ((side_a ? rba : rbb) >> 8) & 255, pk[0], pk[1], pk[2]);
                                }
                            }
                        }
                    }
                    if (last) report(b, cf);
                }
            }
            // ---- the recursion ----
            add_rigid_inertia(IA, Ao, ho, mass);
            DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] += pv[i];
            float U[6];
            DQ_UNROLL for (int r = 0; r < 6; ++r) {
                float acc = 0.0f;
                DQ_UNROLL for (int c = 0; c < 6; ++c) acc += IA[sym6(r, c)] * S[c];
                U[r] = acc;
            }
            const float D = dot6(S, U) + dd;
            const float Dinv = dw::rcp_nr(D);
            const float u = tt - dot6(S, pA);
            DQ_UNROLL for (int r = 0; r < 6; ++r) {
                const float urd = U[r] * Dinv;
                DQ_UNROLL for (int c = r; c < 6; ++c) IA[sym6(r, c)] -= urd * U[c];
            }
            const float ud = u * Dinv;
            float pa[6];
            DQ_UNROLL for (int r = 0; r < 6; ++r) {
                float acc = pA[r] + U[r] * ud;
                DQ_UNROLL for (int c = 0; c < 6; ++c) acc += IA[sym6(r, c)] * cb[c];
                pa[r] = acc;
            }
            DQ_UNROLL for (int r = 0; r < 6; ++r) pA[r] = pa[r];
            DL_SL(b, 0) = mk4(S[0], S[1], S[2], Dinv);
            DL_SL(b, 1) = mk4(S[3], S[4], S[5], u);
            DL_SL(b, 2) = mk4(U[0], U[1], U[2], qd);
            DL_SL(b, 3) = mk4(U[3], U[4], U[5], 0.0f);
        }
    };
    inward_chain(0);
    DL_STAMP(4);
    if (w == 0) {
        // the left leg's total waits for wave 1 in record C; then the short chain off the trunk's end, for wave 2 in record A
        xch_put(L, XC, ln, IA, pA);
        DQ_UNROLL for (int i = 0; i < 21; ++i) IA[i] = 0.0f;
        DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] = 0.0f;
        inward_chain(1);
        xch_put(L, XA, ln, IA, pA);
    } else if (w == 1) {
        // the base body itself: rigid inertia, gyroscopic bias, its ground primitives, the push on its COM
        const float v0[6] = {ww[0], ww[1], ww[2], vo[0], vo[1], vo[2]}, x0[3] = {0, 0, 0};
        float Ao[6], ho[3], mass;
        {
            const float ms = mscale_e[M.bi_gym[0][0]];
            const float bI[6] = {M.bi_I[0][0][0], M.bi_I[0][0][1], M.bi_I[0][0][2], M.bi_I[0][0][3], M.bi_I[0][0][4], M.bi_I[0][0][5]};
            rigid_inertia(1, bcom, M.bi_mass[0][0], bI, ms, bcom, 0.0f, bI, 0.0f, R0, x0, Ao, ho, &mass);
        }
        add_rigid(IA, pA, Ao, ho, mass, v0);
        const int base_ngeom = M.body_ngeom[0];
        bool near_ground = base_ngeom > 0 && (X.root[2] < LM.bound[0]);
        if (TERRAIN) near_ground = base_ngeom > 0 && (X.root[2] - X.zbound < LM.bound[0]);
        if (wave_any(near_ground)) {
            float cf[LMAX_GYM][3];
            DQ_UNROLL for (int t = 0; t < LMAX_GYM; ++t) cf[t][0] = cf[t][1] = cf[t][2] = 0.0f;
            DL_ROLLED for (int g = 0; g < base_ngeom; ++g) {
                float F[3] = {0, 0, 0}, xg[3] = {0, 0, 0};
                if (near_ground) geom_force<TERRAIN>(M.geoms[M.body_geom[0][g]], P, R0, x0, v0, X.root[0], X.root[1], X.root[2], X.mu, F, xg);
                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                    float nb[3];
                    cross3(xg, F, nb);
                    DQ_UNROLL for (int i = 0; i < 3; ++i) { pA[i] -= nb[i]; pA[3 + i] -= F[i]; }
                    const int t = LM.geom_gslot[0][g];
                    DQ_UNROLL for (int t2 = 0; t2 < LMAX_GYM; ++t2)
                        if (t2 == t) { cf[t2][0] += F[0]; cf[t2][1] += F[1]; cf[t2][2] += F[2]; }
                }
            }
            if (last) report(0, cf);
        }
        {   // push on the base COM
            const float Fw[3] = {push_x, push_y, 0.0f};
            float xc[3], nb[3];
            m3v(R0, bcom, xc);
            cross3(xc, Fw, nb);
            DQ_UNROLL for (int i = 0; i < 3; ++i) { pA[i] -= nb[i]; pA[3 + i] -= Fw[i]; }
        }
    } else if (w == 3) {
        xch_put(L, XB, ln, IA, pA);
    }
    DL_STAMP(5);
    wg_barrier();
    DL_STAMP(6);
    if (w == 2) {
        xch_add(L, XA, ln, IA, pA);
        xch_add(L, XB, ln, IA, pA);
        inward_chain(1);
        xch_put(L, XA, ln, IA, pA);
    } else if (w == 1) {
        xch_add(L, XC, ln, IA, pA);
        xch_put(L, XC, ln, IA, pA);
    }
    // ---- leg waves meanwhile: the six unit wrenches on my foot climb my leg.  d = -S'p, p += U d / D ----
    float dp[6][6], dc[6][6];
    if (legwave) {
        DQ_UNROLL for (int c = 0; c < 6; ++c) DQ_UNROLL for (int i = 0; i < 6; ++i) dp[c][i] = (i == c) ? -1.0f : 0.0f;
        DQ_UNROLL for (int i = 6; i >= 1; --i) {
            const int b = 6 * w + i;
            const F4 s0 = DL_SL(b, 0); F4 s1 = DL_SL(b, 1), s2 = DL_SL(b, 2), s3 = DL_SL(b, 3);
            DL_KEEP1(s1); DL_KEEP2(s2, s3);
            const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
            DQ_UNROLL for (int c = 0; c < 6; ++c) {
                const float d = -dot6(S, dp[c]);
                dc[c][i - 1] = d;
                const float k = d * s0.w;
                DQ_UNROLL for (int r = 0; r < 6; ++r) dp[c][r] += U[r] * k;
            }
        }
    }
    DL_STAMP(7);
    wg_barrier();
    DL_STAMP(8);

    // ---- base: total articulated inertia = trunk (record A) + legs and base body (record C); Cholesky, inverse, free
    //      acceleration.  Every wave computes it (same instructions, same inputs): no broadcast, no barrier. ----
    float Minv[21], a0[6];
    {
        float I0[21], p0[6];
        DQ_UNROLL for (int i = 0; i < 21; ++i) I0[i] = 0.0f;
        DQ_UNROLL for (int i = 0; i < 6; ++i) p0[i] = 0.0f;
        xch_add(L, XA, ln, I0, p0);
        xch_add(L, XC, ln, I0, p0);
        float Lc[36], dinv[6];
        DQ_UNROLL for (int i = 0; i < 36; ++i) Lc[i] = 0.0f;
        DQ_UNROLL for (int c = 0; c < 6; ++c) {
            float d = I0[sym6(c, c)];
            DQ_UNROLL for (int k = 0; k < c; ++k) d -= Lc[6 * c + k] * Lc[6 * c + k];
            dinv[c] = dw::rsqrt_nr(d);
            Lc[6 * c + c] = d * dinv[c];
            DQ_UNROLL for (int i = c + 1; i < 6; ++i) {
                float sacc = I0[sym6(i, c)];
                DQ_UNROLL for (int k = 0; k < c; ++k) sacc -= Lc[6 * i + k] * Lc[6 * c + k];
                Lc[6 * i + c] = sacc * dinv[c];
            }
        }
        DQ_UNROLL for (int col = 0; col < 6; ++col) {
            float y[6], xx[6];
            DQ_UNROLL for (int i = 0; i < 6; ++i) {
                float sacc = (i == col) ? 1.0f : 0.0f;
                DQ_UNROLL for (int k = 0; k < i; ++k) sacc -= Lc[6 * i + k] * y[k];
                y[i] = sacc * dinv[i];
            }
            DQ_UNROLL for (int i = 5; i >= 0; --i) {
                float sacc = y[i];
                DQ_UNROLL for (int k = i + 1; k < 6; ++k) sacc -= Lc[6 * k + i] * xx[k];
                xx[i] = sacc * dinv[i];
            }
            DQ_UNROLL for (int i = 0; i <= col; ++i) Minv[sym6(i, col)] = xx[i];
        }
        DQ_UNROLL for (int r = 0; r < 6; ++r) {
            float acc = 0.0f;
            DQ_UNROLL for (int c = 0; c < 6; ++c) acc -= Minv[sym6(r, c)] * p0[c];
            a0[r] = acc;
        }
    }

    DL_STAMP(9);
    // ---- outward pass 2: accelerations; free joint velocity qdf = qd + dt qdd into row 3 .w (a 4-byte store: the waves that
    //      recompute the trunk read the other words of that row meanwhile) ----
    {
        float ar[6] = {0, 0, 0, 0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
        auto o_fetch = [&](int k, F4 (&sr)[4]) {
            const int b = out_body(k);
            sr[0] = DL_SL(b, 0); sr[1] = DL_SL(b, 1); sr[2] = DL_SL(b, 2); sr[3] = DL_SL(b, 3);
        };
        F4 nsr[4];
        o_fetch(0, nsr);
        DL_ROLLED for (int k = 0; k < n_out; ++k) {
            const F4 s0 = nsr[0], s1 = nsr[1], s2 = nsr[2]; F4 s3 = nsr[3];
            DL_KEEP1(s3);
            o_fetch(k + 1 < n_out ? k + 1 : k, nsr);
            const int b = out_body(k);
            if ((m_start >> k) & 1u) { DQ_UNROLL for (int i = 0; i < 3; ++i) { ar[i] = a0[i]; ar[3 + i] = a0[3 + i]; vr[i] = ww[i]; vr[3 + i] = vo[i]; } }
            const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
            const float Dinv = s0.w, u = s1.w, qd = s2.w;
            float m[6], c[6];
            DQ_UNROLL for (int i = 0; i < 6; ++i) m[i] = S[i] * qd;
            dw::motion_cross(vr, m, c);
            DQ_UNROLL for (int i = 0; i < 6; ++i) { ar[i] += c[i]; vr[i] += m[i]; }
            const float qdd = (u - dot6(U, ar)) * Dinv;
            DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] += S[i] * qdd;
            if ((m_store >> k) & 1u) DL_SL(b, 3).w = qd + dt * qdd;
        }
    }
    DL_STAMP(10);
    // free base velocity
    float wwf[3], vowf[3];
    {
        float t2[3];
        cross3(ww, vo, t2);
        DQ_UNROLL for (int i = 0; i < 3; ++i) {
            wwf[i] = ww[i] + dt * a0[i];
            vowf[i] = vo[i] + dt * (a0[3 + i] + t2[i] + P.g[i]);
        }
    }

    // ---- contact phase.  Wave f < 2 solves foot f; the two feet couple through the trunk, so after every corner update the
    //      waves exchange the twist change they cause on the other foot (6 words through LDS, one barrier; waves 2 and 3 only
    //      keep the barrier count).  Block-Jacobi across the feet, Gauss-Seidel over the four corners of a sole, as everywhere. ----
    float dqb[6] = {0, 0, 0, 0, 0, 0};              // base velocity jump
    if (legwave) {
        const int f = w, g = 1 - w;
        // free twist of my foot: base + sum over my leg of S qdf
        float tw[6] = {wwf[0], wwf[1], wwf[2], vowf[0], vowf[1], vowf[2]};
        DQ_UNROLL for (int i = 1; i <= 6; ++i) {
            const int b = 6 * f + i;
            const F4 s0 = DL_SL(b, 0), s1 = DL_SL(b, 1), s3 = DL_SL(b, 3);
            tw[0] += s0.x * s3.w; tw[1] += s0.y * s3.w; tw[2] += s0.z * s3.w;
            tw[3] += s1.x * s3.w; tw[4] += s1.y * s3.w; tw[5] += s1.z * s3.w;
        }
        // responses to my six unit wrenches: base jump, then down my own leg (Wo) and down the other leg (Wx)
        float Wo[6][6], Wx[6][6];
        DQ_UNROLL for (int c = 0; c < 6; ++c)
            DQ_UNROLL for (int r = 0; r < 6; ++r) {
                float acc = 0.0f;
                DQ_UNROLL for (int k = 0; k < 6; ++k) acc -= Minv[sym6(r, k)] * dp[c][k];
                Wo[c][r] = acc; Wx[c][r] = acc;
            }
        DQ_UNROLL for (int i = 1; i <= 6; ++i) {
            {
                const int b = 6 * f + i;
                const F4 s0 = DL_SL(b, 0); F4 s1 = DL_SL(b, 1), s2 = DL_SL(b, 2), s3 = DL_SL(b, 3);
                DL_KEEP1(s1); DL_KEEP2(s2, s3);
                const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
                DQ_UNROLL for (int c = 0; c < 6; ++c) {
                    const float qdd = (dc[c][i - 1] - dot6(U, Wo[c])) * s0.w;
                    DQ_UNROLL for (int r = 0; r < 6; ++r) Wo[c][r] += S[r] * qdd;
                }
            }
            {
                const int b = 6 * g + i;
                const F4 s0 = DL_SL(b, 0); F4 s1 = DL_SL(b, 1), s2 = DL_SL(b, 2), s3 = DL_SL(b, 3);
                DL_KEEP1(s1); DL_KEEP2(s2, s3);
                const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
                DQ_UNROLL for (int c = 0; c < 6; ++c) {
                    const float qdd = (-dot6(U, Wx[c])) * s0.w;
                    DQ_UNROLL for (int r = 0; r < 6; ++r) Wx[c][r] += S[r] * qdd;
                }
            }
        }
        DL_STAMP(11);
        // per corner k: Go[k][r][j] = twist change of my foot (row r) per unit impulse component j at the corner, Gx the same for
        // the other foot: the wrench of an impulse d at lever r is [r x d; d].  On terrain j counts the corner's frame (t1, t2, n).
        float Go[4][6][3], Gx[4][6][3], invd[4][3], cpl[4][3];
        const float rreg = 1.0f / (1.0f + P.cfm);
        DQ_UNROLL for (int k = 0; k < 4; ++k) {
            const float *r = X.rk[k];
            DQ_UNROLL for (int rw = 0; rw < 6; ++rw) {
                float go[3], gx[3];
                go[0] = Wo[1][rw] * r[2] - Wo[2][rw] * r[1] + Wo[3][rw];
                go[1] = Wo[2][rw] * r[0] - Wo[0][rw] * r[2] + Wo[4][rw];
                go[2] = Wo[0][rw] * r[1] - Wo[1][rw] * r[0] + Wo[5][rw];
                gx[0] = Wx[1][rw] * r[2] - Wx[2][rw] * r[1] + Wx[3][rw];
                gx[1] = Wx[2][rw] * r[0] - Wx[0][rw] * r[2] + Wx[4][rw];
                gx[2] = Wx[0][rw] * r[1] - Wx[1][rw] * r[0] + Wx[5][rw];
                if (TERRAIN) {
                    const float *fr = X.frame[k];
                    DQ_UNROLL for (int a = 0; a < 3; ++a) {
                        Go[k][rw][a] = go[0] * fr[3 * a] + go[1] * fr[3 * a + 1] + go[2] * fr[3 * a + 2];
                        Gx[k][rw][a] = gx[0] * fr[3 * a] + gx[1] * fr[3 * a + 1] + gx[2] * fr[3 * a + 2];
                    }
                } else {
                    DQ_UNROLL for (int a = 0; a < 3; ++a) { Go[k][rw][a] = go[a]; Gx[k][rw][a] = gx[a]; }
                }
            }
            // 3x3 diagonal block of the Delassus matrix: A[i][j] = velocity component i of the corner per unit impulse j
            float Ak[3][3];
            DQ_UNROLL for (int j = 0; j < 3; ++j) {
                const float wv[3] = {Go[k][0][j], Go[k][1][j], Go[k][2][j]};
                float vw[3] = {Go[k][3][j] + wv[1] * r[2] - wv[2] * r[1], Go[k][4][j] + wv[2] * r[0] - wv[0] * r[2], Go[k][5][j] + wv[0] * r[1] - wv[1] * r[0]};
                if (TERRAIN) {
                    const float *fr = X.frame[k];
                    DQ_UNROLL for (int a = 0; a < 3; ++a) Ak[a][j] = fr[3 * a] * vw[0] + fr[3 * a + 1] * vw[1] + fr[3 * a + 2] * vw[2];
                } else {
                    DQ_UNROLL for (int a = 0; a < 3; ++a) Ak[a][j] = vw[a];
                }
            }
            invd[k][0] = X.act[k] ? dw::rcp_nr(Ak[0][0]) * rreg : 0.0f;
            invd[k][1] = X.act[k] ? dw::rcp_nr(Ak[1][1]) * rreg : 0.0f;
            invd[k][2] = X.act[k] ? dw::rcp_nr(Ak[2][2]) * rreg : 0.0f;
            cpl[k][0] = Ak[0][2];     // x row, z column
            cpl[k][1] = Ak[1][2];     // y row, z column
            cpl[k][2] = Ak[1][0];     // y row, x column
        }
        DL_STAMP(12);
        // warm start: the previous substep's impulses of my corners, their effect on both feet
        float Pk[4][3];
        {
            DQ_UNROLL for (int k = 0; k < 4; ++k) DQ_UNROLL for (int i = 0; i < 3; ++i) Pk[k][i] = X.act[k] ? X.warm[3 * k + i] : 0.0f;
        }
        {
            float xo[6] = {0, 0, 0, 0, 0, 0};
            DQ_UNROLL for (int k = 0; k < 4; ++k)
                DQ_UNROLL for (int rw = 0; rw < 6; ++rw) {
                    tw[rw] += Go[k][rw][0] * Pk[k][0] + Go[k][rw][1] * Pk[k][1] + Go[k][rw][2] * Pk[k][2];
                    xo[rw] += Gx[k][rw][0] * Pk[k][0] + Gx[k][rw][1] * Pk[k][1] + Gx[k][rw][2] * Pk[k][2];
                }
            L.xch[XB][2 * f][ln] = mk4(xo[0], xo[1], xo[2], 0.0f);
            L.xch[XB][2 * f + 1][ln] = mk4(xo[3], xo[4], xo[5], 0.0f);
            X.seq += 1;
            flag_post(&L.wflag[f][1], X.seq);
            flag_wait(&L.wflag[g][1], X.seq);
            const F4 o0 = L.xch[XB][2 * g][ln], o1 = L.xch[XB][2 * g + 1][ln];
            tw[0] += o0.x; tw[1] += o0.y; tw[2] += o0.z; tw[3] += o1.x; tw[4] += o1.y; tw[5] += o1.z;
        }
        DL_STAMP(13);
        // projected Gauss-Seidel
        DL_ROLLED for (int it = 0; it < P.iters; ++it) {
            DQ_UNROLL for (int kk = 0; kk < 4; ++kk) {
                const float *r = X.rk[kk];
                const float vwld[3] = {tw[3] + tw[1] * r[2] - tw[2] * r[1], tw[4] + tw[2] * r[0] - tw[0] * r[2], tw[5] + tw[0] * r[1] - tw[1] * r[0]};
                float vx0 = vwld[0], vy0 = vwld[1], vz = vwld[2];
                if (TERRAIN) {
                    const float *fr = X.frame[kk];
                    vx0 = fr[0] * vwld[0] + fr[1] * vwld[1] + fr[2] * vwld[2];
                    vy0 = fr[3] * vwld[0] + fr[4] * vwld[1] + fr[5] * vwld[2];
                    vz = fr[6] * vwld[0] + fr[7] * vwld[1] + fr[8] * vwld[2];
                }
                const float Px = Pk[kk][0], Py = Pk[kk][1], Pz = Pk[kk][2];
                float dz = -(vz - X.vmin[kk]) * invd[kk][2];
                float pz = Pz + dz;
                if (pz < 0) pz = 0;
                dz = pz - Pz;
                const float vx = vx0 + cpl[kk][0] * dz;
                const float dx = -vx * invd[kk][0];
                const float vy = vy0 + cpl[kk][1] * dz + cpl[kk][2] * dx;
                const float dy = -vy * invd[kk][1];
                float px = Px + dx, py = Py + dy;
                const float lim = X.mu * pz, n2 = px * px + py * py;
                if (n2 > lim * lim) {
                    const float sc = lim * dw::rsqrt_nr(n2);
                    px *= sc; py *= sc;
                }
                const float d[3] = {px - Px, py - Py, dz};
                Pk[kk][0] = px; Pk[kk][1] = py; Pk[kk][2] = pz;
                float xo[6];
                DQ_UNROLL for (int rw = 0; rw < 6; ++rw) {
                    tw[rw] += Go[kk][rw][0] * d[0] + Go[kk][rw][1] * d[1] + Go[kk][rw][2] * d[2];
                    xo[rw] = Gx[kk][rw][0] * d[0] + Gx[kk][rw][1] * d[1] + Gx[kk][rw][2] * d[2];
                }
                const int par = (kk & 1) ^ 1;         // (the warm start used records 0..3 of buffer B: alternate B / C from here)
                const int buf = par ? XC : XB;
                L.xch[buf][2 * f][ln] = mk4(xo[0], xo[1], xo[2], 0.0f);
                L.xch[buf][2 * f + 1][ln] = mk4(xo[3], xo[4], xo[5], 0.0f);
                X.seq += 1;
                flag_post(&L.wflag[f][1], X.seq);
                flag_wait(&L.wflag[g][1], X.seq);
                const F4 o0 = L.xch[buf][2 * g][ln], o1 = L.xch[buf][2 * g + 1][ln];
                tw[0] += o0.x; tw[1] += o0.y; tw[2] += o0.z; tw[3] += o1.x; tw[4] += o1.y; tw[5] += o1.z;
            }
        }
        DL_STAMP(14);
        // impulses -> wrench on my foot -> up my leg (the sweep's d into row 1 .w), my share of the base's bias change
        float Fs[3] = {0, 0, 0}, Nm[3] = {0, 0, 0};
        DQ_UNROLL for (int k = 0; k < 4; ++k) {
            float pw[3] = {Pk[k][0], Pk[k][1], Pk[k][2]};
            if (TERRAIN) {
                const float p0 = pw[0], p1 = pw[1], p2 = pw[2];
                DQ_UNROLL for (int i = 0; i < 3; ++i) pw[i] = p0 * X.frame[k][i] + p1 * X.frame[k][3 + i] + p2 * X.frame[k][6 + i];
            }
            float t[3];
            cross3(X.rk[k], pw, t);
            DQ_UNROLL for (int i = 0; i < 3; ++i) { Fs[i] += pw[i]; Nm[i] += t[i]; }
        }
        float dpb[6] = {-Nm[0], -Nm[1], -Nm[2], -Fs[0], -Fs[1], -Fs[2]};
        DQ_UNROLL for (int i = 6; i >= 1; --i) {
            const int b = 6 * f + i;
            const F4 s0 = DL_SL(b, 0); F4 s1 = DL_SL(b, 1), s2 = DL_SL(b, 2), s3 = DL_SL(b, 3);
            DL_KEEP1(s1); DL_KEEP2(s2, s3);
            const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
            const float d = -dot6(S, dpb);
            const float k = d * s0.w;
            DQ_UNROLL for (int r = 0; r < 6; ++r) dpb[r] += U[r] * k;
            DL_SL(b, 1).w = d;
        }
        L.xch[XA][2 * f][ln] = mk4(dpb[0], dpb[1], dpb[2], 0.0f);
        L.xch[XA][2 * f + 1][ln] = mk4(dpb[3], dpb[4], dpb[5], 0.0f);
        DQ_UNROLL for (int k = 0; k < 4; ++k) DQ_UNROLL for (int i = 0; i < 3; ++i) X.warm[3 * k + i] = Pk[k][i];
        if (last) {
            DQ_UNROLL for (int i = 0; i < 3; ++i) X.footT[i] = footF[i] + Fs[i] * inv_dt;
            if (X.valid) {
                float *dst = B.contact_forces + ((size_t)DW_NUM_BODIES * e + my_sole_gym) * 3;
                dst[0] = X.footT[0]; dst[1] = X.footT[1]; dst[2] = X.footT[2];
            }
        }
    } else {
        idle.work();
    }
    DL_STAMP(15);
    wg_barrier();
    DL_STAMP(16);
    if (!legwave) idle.publish();          // (the leg waves' hand-offs are over: exchange record C is free)
    {
        const F4 a0v = L.xch[XA][0][ln], a1v = L.xch[XA][1][ln], b0v = L.xch[XA][2][ln], b1v = L.xch[XA][3][ln];
        const float tot[6] = {a0v.x + b0v.x, a0v.y + b0v.y, a0v.z + b0v.z, a1v.x + b1v.x, a1v.y + b1v.y, a1v.z + b1v.z};
        DQ_UNROLL for (int r = 0; r < 6; ++r) {
            float acc = 0.0f;
            DQ_UNROLL for (int c = 0; c < 6; ++c) acc -= Minv[sym6(r, c)] * tot[c];
            dqb[r] = acc;
        }
    }

    // ---- outward pass 3: velocity jumps down the tree, final joint velocities (speed limit) into row 2 .w ----
    {
        float ar[6] = {0, 0, 0, 0, 0, 0};
        auto o_fetch = [&](int k, F4 &r1, F4 (&sr)[4]) {
            r1 = reinterpret_cast<const F4 *>(fkt + k)[1];
            const int b = out_body(k);
            sr[0] = DL_SL(b, 0); sr[1] = DL_SL(b, 1); sr[2] = DL_SL(b, 2); sr[3] = DL_SL(b, 3);
        };
        F4 nr1, nsr[4];
        o_fetch(0, nr1, nsr);
        DL_ROLLED for (int k = 0; k < n_out; ++k) {
            const float vmax = nr1.w;
            const F4 s0 = nsr[0], s1 = nsr[1], s3 = nsr[3]; F4 s2 = nsr[2];
            DL_KEEP1(s2);
            o_fetch(k + 1 < n_out ? k + 1 : k, nr1, nsr);
            const int b = out_body(k);
            if ((m_start >> k) & 1u) { DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] = dqb[i]; }
            const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
            const float d = (b <= 12) ? s1.w : 0.0f;          // (only the legs carry an impulse of their own)
            const float dq = (d - dot6(U, ar)) * s0.w;
            DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] += S[i] * dq;
            float qd = s3.w + dq;
            if (qd > vmax) qd = vmax;
            if (qd < -vmax) qd = -vmax;
            if ((m_store >> k) & 1u) DL_SL(b, 2).w = qd;
        }
    }
    DL_STAMP(17);
    // ---- base: final velocity, clamps, pose update ----
    {
        float wwn[3], von[3];
        DQ_UNROLL for (int i = 0; i < 3; ++i) { wwn[i] = wwf[i] + dqb[i]; von[i] = vowf[i] + dqb[3 + i]; }
        const float wn2 = dot3(wwn, wwn);
        if (wn2 > P.max_ang_vel * P.max_ang_vel) {
            const float sc = P.max_ang_vel * dw::rsqrt_nr(wn2);
            wwn[0] *= sc; wwn[1] *= sc; wwn[2] *= sc;
        }
        X.root[0] += dt * von[0]; X.root[1] += dt * von[1]; X.root[2] += dt * von[2];
        const float w2 = dot3(wwn, wwn);
        const float hx = 0.5f * dt;
        const float x2 = w2 * hx * hx;
        const float sh = hx * (1.0f + x2 * (-1.0f / 6 + x2 * (1.0f / 120 + x2 * (-1.0f / 5040 + x2 * (1.0f / 362880)))));
        const float ch = 1.0f + x2 * (-0.5f + x2 * (1.0f / 24 + x2 * (-1.0f / 720 + x2 * (1.0f / 40320))));
        const float x1 = wwn[0] * sh, y1 = wwn[1] * sh, z1 = wwn[2] * sh, w1 = ch;
        const float x2q = qn[0], y2 = qn[1], z2 = qn[2], w2q = qn[3];
        float qo[4] = {w1 * x2q + x1 * w2q + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2q + z1 * x2q,
                       w1 * z2 + x1 * y2 - y1 * x2q + z1 * w2q, w1 * w2q - x1 * x2q - y1 * y2 - z1 * z2};
        const float ninv = dw::rsqrt_nr(qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3]);
        DQ_UNROLL for (int i = 0; i < 4; ++i) { qo[i] *= ninv; X.root[3 + i] = qo[i]; }
        if (P.vel_at_com) {
            float Rn[9], rcom[3], tt3[3];
            quat_to_mat(qo, Rn);
            m3v(Rn, bcom, rcom);
            cross3(wwn, rcom, tt3);
            DQ_UNROLL for (int i = 0; i < 3; ++i) von[i] += tt3[i];
        }
        DQ_UNROLL for (int i = 0; i < 3; ++i) { X.root[7 + i] = von[i]; X.root[10 + i] = wwn[i]; }
    }
    DL_STAMP(18);
    wg_barrier();          // the integrator (items over all threads) reads row 2 .w of every body
    DL_STAMP(19);
}

// copies the packed constants from the device-resident model into LDS (once per kernel; the caller's next barrier publishes them)
DQ_HD void stage_hot(LLds &L, const LaneModel &LM) {
    const F4 *src = reinterpret_cast<const F4 *>(&LM.hot);
    F4 *dst = reinterpret_cast<F4 *>(&L.hot);
    constexpr int NQ = (int)(sizeof(LHot) / 16);
    for (int i = tid(); i < NQ; i += NT) dst[i] = src[i];
    if (tid() < NWAVE) { L.wflag[tid()][0] = 0; L.wflag[tid()][1] = 0; }
}

// Lane set-up shared by the entry points: which env this lane works for, its base state and parameters.
DQ_HD void lane_init(LState &X, int group, int num_envs, float friction, const OBuf &B) {
    X.ln = lane();
    X.w = wave();
    const int eg = group * EPW + X.ln;
    X.valid = eg < num_envs;
    X.e = X.valid ? eg : num_envs - 1;
    DQ_UNROLL for (int i = 0; i < 13; ++i) X.root[i] = B.root_states[(size_t)13 * X.e + i];
    X.mu = friction * OQ_COLD(friction_scale)[X.e];
    X.coll = 0;
    X.footT[0] = X.footT[1] = X.footT[2] = 0.0f;
    X.seq = 0;
    DQ_UNROLL for (int i = 0; i < 12; ++i) X.warm[i] = 0.0f;
    DQ_UNROLL for (int c = 0; c < 4; ++c) {
        X.act[c] = 0; X.vmin[c] = 0.0f;
        DQ_UNROLL for (int i = 0; i < 3; ++i) X.rk[c][i] = 0.0f;
        DQ_UNROLL for (int i = 0; i < 9; ++i) X.frame[c][i] = (i % 4 == 0) ? 1.0f : 0.0f;
    }
}

// a run of `wpe` words per env for the workgroup's envs, global -> LDS (float words at LF + dst), in 16-byte pieces: the runs of
// consecutive envs are contiguous in global memory and in LDS.  nvalid < 64: the last workgroup (words of envs past the end
// are not read).
DQ_HD void stage_in(float *LF, int dst, const float *src, int wpe, int nvalid) {
    const int t = tid();
    const int nw = EPW * wpe, nv = nvalid * wpe;
    if (nvalid == EPW) {
        const F4 *s4 = reinterpret_cast<const F4 *>(src);
        F4 *d4 = reinterpret_cast<F4 *>(LF + dst);
        const int np = nw / 4;
        // eight requests in flight per thread before the first store (as a plain loop this was one memory round trip per piece)
        constexpr int GRP = 8;
        for (int p0 = 0; p0 < np; p0 += GRP * NT) {
            F4 v[GRP];
            DQ_UNROLL for (int u = 0; u < GRP; ++u) { const int p = p0 + t + NT * u; v[u] = s4[p < np ? p : np - 1]; }
            DQ_UNROLL for (int u = 0; u < GRP; ++u) { const int p = p0 + t + NT * u; if (p < np) d4[p] = v[u]; }
        }
    } else {
        for (int i = t; i < nw; i += NT) LF[dst + i] = i < nv ? src[i] : 0.0f;
    }
}
DQ_HD void stage_out(const float *LF, int srcw, float *dst, int wpe, int nvalid) {
    const int t = tid();
    const int nw = EPW * wpe, nv = nvalid * wpe;
    if (nvalid == EPW) {
        const F4 *s4 = reinterpret_cast<const F4 *>(LF + srcw);
        F4 *d4 = reinterpret_cast<F4 *>(dst);
        for (int p = t; p < nw / 4; p += NT) d4[p] = s4[p];
    } else {
        for (int i = t; i < nv; i += NT) dst[i] = LF[srcw + i];
    }
}

constexpr int MAXOWN = 11;           // joints whose task state one wave keeps in registers (dw_lane_kernels.h)
// what the fused step hands to its post phase, per joint of my wave's list: new angle / rate, encoder angle / rate
struct LaneKeep { float q[MAXOWN], qd[MAXOWN], qn[MAXOWN], qv[MAXOWN]; };

// ---- joint-parallel phases: per-joint work that touches the Gym tensors runs over ITEMS (env, dof) = thread + 256 k of the
// workgroup's 64 x 33 joints, so that a wave-instruction reads or writes consecutive addresses. ----
constexpr int LNI = (EPW * ND + NT - 1) / NT;      // 9 items per thread
struct JointItem { int ok, el, d, b, env; };
DQ_HD JointItem joint_item(int group, int num_envs, int t, int k) {
    JointItem it;
    const int i = t + NT * k;
    it.el = i / ND; it.d = i - ND * it.el; it.b = it.d + 1;
    const int eg = group * EPW + it.el;
    it.ok = (i < EPW * ND) && (eg < num_envs);
    if (!(i < EPW * ND)) { it.el = 0; it.d = 0; it.b = 1; }
    it.env = eg < num_envs ? eg : num_envs - 1;
    return it;
}
// the prologue's store of one joint's inputs: slot row 0, and for a trunk joint also its (q, qd) in exchange record A
DQ_HD void put_joint_inputs(LLds &L, const LaneModel &LM, const JointItem &it, float q, float qd, float tt, float dd) {
    L.slot[it.b - 1][0][it.el] = mk4(q, qd, tt, dd);
    const int ts = LM.trunk_slot[it.b];
    if (ts >= 0) {
        float *p = reinterpret_cast<float *>(&L.xch[XA][ts >> 1][it.el]) + 2 * (ts & 1);
        p[0] = q; p[1] = qd;
    }
}
// zero-fill of the envs' rows of contact_forces before the last substep reports into them
DQ_HD void zero_contact_rows(const OBuf &B, int group, int num_envs) {
    const int t = tid();
    constexpr int PER_ENV = DW_NUM_BODIES * 3;
    const int nvalid = num_envs - group * EPW < EPW ? num_envs - group * EPW : EPW;
    float *base = B.contact_forces + (size_t)group * EPW * PER_ENV;
    for (int i = t; i < nvalid * PER_ENV; i += NT) base[i] = 0.0f;
}

// Gym-boundary substep for the 64 envs of a workgroup: tau [N,33], push [N,2] or nullptr
template <bool TERRAIN>
DQ_HD void lane_simulate(LLds &L, const LaneModel &LM, const DevModel &M, const PhysParams &P, float friction, int num_envs,
                         const OBuf &B, const float *tau, const float *push, int group) {
    LState X;
    lane_init(X, group, num_envs, friction, B);
    stage_hot(L, LM);
    const int t = tid();
    float qkeep[LNI];
    DQ_UNROLL for (int k = 0; k < LNI; ++k) {
        const JointItem it = joint_item(group, num_envs, t, k);
        const size_t g = (size_t)ND * it.env + it.d;
        const float q = B.dof_state[g * 2], qd = B.dof_state[g * 2 + 1];
        const float damp = B.dof_damping[g], arm = B.dof_armature[g];
        qkeep[k] = q;
        if (t + NT * k < EPW * ND) put_joint_inputs(L, LM, it, q, qd, tau[g] - damp * qd, arm + P.dt * damp);
    }
    zero_contact_rows(B, group, num_envs);
    wg_barrier_global();
    // (warm-start impulses of my sole from the task record, if one is bound; back into it below)
    if (X.w < 2 && B.env_state) {
        const F4 *wsrc = reinterpret_cast<const F4 *>(B.env_state + (size_t)DW_ES_WORDS * X.e + DW_ES_WARM + 12 * X.w);
        DQ_UNROLL for (int i = 0; i < 3; ++i) { const F4 v = wsrc[i]; X.warm[4 * i] = v.x; X.warm[4 * i + 1] = v.y; X.warm[4 * i + 2] = v.z; X.warm[4 * i + 3] = v.w; }
    }
    NoIdleWork idle;
    lane_substep<TERRAIN>(L, LM, M, P, X, B, push ? push[2 * X.e] : 0.0f, push ? push[2 * X.e + 1] : 0.0f, true, idle);
    if (X.w < 2 && B.env_state && X.valid) {
        F4 *wdst = reinterpret_cast<F4 *>(B.env_state + (size_t)DW_ES_WORDS * X.e + DW_ES_WARM + 12 * X.w);
        DQ_UNROLL for (int i = 0; i < 3; ++i) wdst[i] = mk4(X.warm[4 * i], X.warm[4 * i + 1], X.warm[4 * i + 2], X.warm[4 * i + 3]);
    }
    DQ_UNROLL for (int k = 0; k < LNI; ++k) {
        const JointItem it = joint_item(group, num_envs, t, k);
        float qd = reinterpret_cast<const float *>(&L.slot[it.b - 1][2][it.el])[3];
        float q = qkeep[k] + P.dt * qd;
        const float qlo = M.qlo[it.d], qhi = M.qhi[it.d];
        if (q < qlo) { q = qlo; if (qd < 0) qd = 0; }
        if (q > qhi) { q = qhi; if (qd > 0) qd = 0; }
        if (it.ok) {
            const size_t g = (size_t)ND * it.env + it.d;
            B.dof_state[g * 2] = q; B.dof_state[g * 2 + 1] = qd;
        }
    }
    if (X.valid && X.w == 0) { DQ_UNROLL for (int i = 0; i < 13; ++i) B.root_states[(size_t)13 * X.e + i] = X.root[i]; }
}

}  // namespace dwl
