// dw_quad.h -- one physics substep (stand-in for the reference's closed `gym.simulate`, call site
// tasks/dyros_dynamic_walk.py:525) in the QUAD layout: 4 lanes per env, 16 envs per wavefront (dw_quad_wave.h,
// schedule and tables: dw_quad_model.h).  Same physics, same order of the contact iterations and the same written
// decisions as dw_physics.h / oracle/dw_physics.c (DESIGN.md "Physics model"); what changes is who computes what:
//
//   * a lane owns a limb and walks it body by body with the running quantity in registers -- pose and velocity outward,
//     articulated inertia (21 words) and bias force inward, acceleration / velocity jump outward again -- so every
//     instruction does useful arithmetic for 64 lanes (the wave-per-env kernel ran these recursions on 6-12 lanes);
//   * all spatial quantities live in one frame (world axes, origin O = base origin at the start of the substep), so a
//     child's articulated inertia is ADDED to its parent's: continuing a chain costs nothing, limbs meet through DPP adds;
//   * per body only 16 words survive a pass, in an LDS slot private to the owner lane (4 x ds_read/write_b128,
//     bank-conflict free): orientation quaternion, position, velocity, joint inputs after the kinematics pass;
//     joint subspace S, U = IA S, 1/D, u after the inward pass.  34 bodies x 64 B x 16 envs = 34 KB per wave, which with the
//     proxy table stays under 40 KB: 4 waves per CU, one per SIMD, 64 envs resident per CU (the wave-per-env kernel: 12);
//   * the sole contacts are solved in foot-twist space: the 12x12 inverse operational inertia W of the two feet is held one
//     3-row slab per lane (12 unit-wrench responses, 3 per lane), the projected Gauss-Seidel keeps the two foot twists
//     tw = tw_free + W lambda in registers and updates them with 36 FMAs per lane and corner pair.  v_k = J_k tw and
//     lambda += J_k' dp reproduce the 24-row velocity-level iteration of dw_physics.h exactly (same pairs, same order).
#pragma once

#include "dw_physics.h"
#include "dw_quad_model.h"
#include "dw_quad_wave.h"

#if defined(__HIPCC__)
#define DQ_UNROLL _Pragma("unroll")
#else
#define DQ_UNROLL
#endif

namespace dwq {

using dw::DevModel; using dw::PhysParams; using dw::NB; using dw::ND;
using dw::cross3; using dw::dot3; using dw::m3v; using dw::m3tv; using dw::dot6; using dw::quat_to_mat; using dw::sym6;

#if defined(__HIPCC__)
typedef float4 F4;
#else
struct alignas(16) F4 { float x, y, z, w; };
#endif
DQ_HD F4 mk4(float x, float y, float z, float w) { F4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

// One wave's LDS: body slots [body][quad][position], position = (env + 4 * owner lane) & 15 so that the four lanes of a
// quad, which work on four different bodies, fall on different banks for 16-byte accesses; and the hot tables.
struct alignas(16) QLds {
    F4   slot[NB * 4][EPW];
    QHot hot;
};
static_assert(QMAX_OWN <= 4, "scGym0 holds one byte per own proxy");
static_assert(sizeof(QLds) <= 40960, "QLds must leave room for 4 waves per CU (160 KB of LDS)");

// 16-byte LDS accesses that stay 16 bytes wide: without the opaque touch the compiler narrows a load whose .w is unused to
// ds_read_b96, which costs twice the LDS cycles of ds_read_b128 (MI355X_MICROARCH.md, LDS table).
#if defined(__HIPCC__)
typedef float v4f_t __attribute__((ext_vector_type(4)));
DQ_HD F4 ld4(const F4 &p) {
    v4f_t r = *reinterpret_cast<const v4f_t *>(&p);
    asm volatile("" : "+v"(r));
    return mk4(r.x, r.y, r.z, r.w);
}
#else
DQ_HD F4 ld4(const F4 &p) { return p; }
#endif
DQ_HD int f2i(float f) { return __builtin_bit_cast(int, f); }

// copies the hot tables from the device-resident model into LDS (once per kernel)
DQ_HD void stage_hot(QLds &L, const QuadModel &QM) {
    const int l = lane_id();
    const F4 *src = reinterpret_cast<const F4 *>(&QM.hot);
    F4 *dst = reinterpret_cast<F4 *>(&L.hot);
    constexpr int NQ = (int)(sizeof(QHot) / 16);
    for (int i = l; i < NQ; i += 64) dst[i] = src[i];
    wave_sync();
}

// What a lane keeps in registers across the phases of a step.
struct QLane {
    int   lane, j, el, env, valid, pos;     // quad lane, env within the wave, global env (clamped), position in the slot rows
    float root[13];
    float mu;
    float warm[12];                          // impulses of the 4 corners of "my" foot (foot j & 1), from the previous substep
    float footF[3];                          // non-sole contact force on my foot's sole Gym body (lanes 0, 1)
    int   coll;                              // last substep: one of my non-sole Gym bodies reports more than 1 N (termination)
    float footT[3];                          // last substep: net contact force on my sole body (lanes 0, 1)
    int   stamp_base;                        // profiling builds only
    // second (welded) inertial record of my sole body and its mass scale, fetched once per kernel (quad_lane_second_inertial):
    // read where the inward pass needs them they cost two dependent memory round trips per substep
    float in1[10], ms1;                      // com[3], mass, I[6]
};

// sin and cos for |x| up to a few turns (joint half-angles): Cody-Waite reduction to [-pi/4, pi/4], the classic single-
// precision minimax polynomials there (~1 ulp).  libm's sincosf spends >100 instructions on arguments this code never sees.
DQ_HD void sincos_fast(float x, float *s, float *c) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.5707962512969971f, x);
    r = fmaf(k, -7.5497894158615964e-08f, r);
    const float r2 = r * r;
    const float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * (-1.9515295891e-4f)));
    const float cp = 1.0f + r2 * (-0.5f + r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f)));
    const int q = (int)k & 3;
    float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    *s = ss; *c = cc;
}

DQ_HD void qmul(const float *a, const float *b, float *o) {     // xyzw
    const float x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const float y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    const float z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    const float w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

// x[i] of quad lane xl (wave-uniform xl) for a small register array
template <int N> DQ_HD void quad_bcast_arr(int xl, const float (&s)[N], float (&d)[N]) {
    if (xl == 0) { DQ_UNROLL for (int i = 0; i < N; ++i) d[i] = quad_bcast<0>(s[i]); }
    else if (xl == 1) { DQ_UNROLL for (int i = 0; i < N; ++i) d[i] = quad_bcast<1>(s[i]); }
    else if (xl == 2) { DQ_UNROLL for (int i = 0; i < N; ++i) d[i] = quad_bcast<2>(s[i]); }
    else { DQ_UNROLL for (int i = 0; i < N; ++i) d[i] = quad_bcast<3>(s[i]); }
}

// Profiling builds (-DDQ_STAMPS) record the clock at phase boundaries of wave 0 into the free tail of gate_acc (words
// 200..): tools/phase_stamps.py.  Never defined in the shipped library.
#if defined(DQ_STAMPS) && defined(__HIPCC__)
#define DQ_STAMP(B, n) do { if (blockIdx.x == 0 && threadIdx.x == 0) (B).gate_acc[200 + (n)] = (int64_t)__builtin_readcyclecounter(); } while (0)
#else
#define DQ_STAMP(B, n) do { } while (0)
#endif
#define DQ_SLOT(b, q, p) L.slot[(b) * 4 + (q)][(p)]
#define DQ_LD(b, q, p) ld4(L.slot[(b) * 4 + (q)][(p)])

// |F| > 1 N with torch.norm's summation order for 3 elements (the termination test of tasks/dyros_dynamic_walk.py:590)
DQ_HD bool over_1n(const float *F) {
    float b0 = fmaf(F[0], F[0], 0.0f);
    b0 = fmaf(F[1], F[1], b0);
    b0 = fmaf(F[2], F[2], b0);
    return sqrtf(b0) > 1.0f;
}

// Ground penalty force of one primitive of body b (dw_physics.h K4).  R, x: body rotation / origin relative to O; v: body
// twist about O.  Returns the force in F and the contact point relative to O in xr.
template <bool TERRAIN>
DQ_HD void geom_force(const DwGeom &ge, const PhysParams &P, const float *R, const float *x, const float *v, float rootx, float rooty,
                      float rootz, float mu, float *F, float *xr) {
    F[0] = F[1] = F[2] = 0.0f;
    float rl[3];
    if (ge.type == 0) {
        float e[3];
        DQ_UNROLL for (int i = 0; i < 3; ++i) {
            const float rg = R[6] * ge.rot[i] + R[7] * ge.rot[3 + i] + R[8] * ge.rot[6 + i];      // world z of box axis i
            e[i] = (rg > 0.0f ? -1.0f : 1.0f) * ge.size[i];
        }
        m3v(ge.rot, e, rl);
        rl[0] += ge.pos[0]; rl[1] += ge.pos[1]; rl[2] += ge.pos[2];
    } else {
        const float al[3] = {ge.rot[2], ge.rot[5], ge.rot[8]};
        float aw[3];
        m3v(R, al, aw);
        const float sgn = aw[2] >= 0 ? -1.0f : 1.0f;
        const float dw3[3] = {-aw[2] * aw[0], -aw[2] * aw[1], 1.0f - aw[2] * aw[2]};
        const float dn = sqrtf(dot3(dw3, dw3));
        float off[3] = {0, 0, 0};
        if (dn > 1e-6f) {
            const float k = -ge.size[0] / dn;
            const float ow[3] = {k * dw3[0], k * dw3[1], k * dw3[2]};
            m3tv(R, ow, off);
        }
        DQ_UNROLL for (int i = 0; i < 3; ++i) rl[i] = ge.pos[i] + sgn * ge.size[1] * al[i] + off[i];
    }
    float wv[3];
    m3v(R, rl, wv);
    DQ_UNROLL for (int i = 0; i < 3; ++i) xr[i] = x[i] + wv[i];
    const float zmin = rootz + xr[2];
    if (TERRAIN) {
        float hh, fr[9];
        dw::terrain_sample(P, rootx + xr[0], rooty + xr[1], &hh, fr);
        const float *nrm = fr + 6;
        const float dist = (zmin - hh) * nrm[2];
        if (dist < 0) {
            float t[3], vw[3];
            cross3(v, xr, t);
            DQ_UNROLL for (int i = 0; i < 3; ++i) vw[i] = v[3 + i] + t[i];
            const float vn = dot3(vw, nrm);
            float fn = P.pen_k * (-dist) - P.pen_c * vn;
            if (fn < 0) fn = 0;
            const float vt[3] = {vw[0] - vn * nrm[0], vw[1] - vn * nrm[1], vw[2] - vn * nrm[2]};
            const float sp = sqrtf(dot3(vt, vt));
            DQ_UNROLL for (int i = 0; i < 3; ++i) F[i] = fn * nrm[i];
            if (sp > 1e-9f) {
                float ft = P.pen_c * sp;
                const float lim = mu * fn;
                if (ft > lim) ft = lim;
                DQ_UNROLL for (int i = 0; i < 3; ++i) F[i] -= ft * vt[i] / sp;
            }
        }
    } else if (zmin < 0) {
        float t[3], vw[3];
        cross3(v, xr, t);
        DQ_UNROLL for (int i = 0; i < 3; ++i) vw[i] = v[3 + i] + t[i];
        float fn = P.pen_k * (-zmin) - P.pen_c * vw[2];
        if (fn < 0) fn = 0;
        const float sp = sqrtf(vw[0] * vw[0] + vw[1] * vw[1]);
        F[2] = fn;
        if (sp > 1e-9f) {
            float ft = P.pen_c * sp;
            const float lim = mu * fn;
            if (ft > lim) ft = lim;
            F[0] = -ft * vw[0] / sp; F[1] = -ft * vw[1] / sp;
        }
    }
}

// Rigid-body inertia of a body about O in world axes from its (<= 2) inertial records (dw_physics.h K3):
// Ao (symmetric 3x3, 6 words: 00 01 02 11 12 22), ho = first moment, mass.
DQ_HD void rigid_inertia(int nin, const float *com0, float m0, const float *I0, float ms0, const float *com1, float m1, const float *I1,
                         float ms1, const float *R, const float *x, float *Ao, float *ho, float *mass_out) {
    float A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, h[3] = {0, 0, 0}, mass = 0.0f;
    DQ_UNROLL for (int k = 0; k < 2; ++k) {
        if (k < nin) {
            const float *cm = k ? com1 : com0, *I6 = k ? I1 : I0;
            const float ms = k ? ms1 : ms0;
            const float mk = ms * (k ? m1 : m0);
            const float cc = dot3(cm, cm);
            const float Ic[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
            DQ_UNROLL for (int r3 = 0; r3 < 3; ++r3)
                DQ_UNROLL for (int c3 = 0; c3 < 3; ++c3)
                    A[3 * r3 + c3] += ms * Ic[3 * r3 + c3] + mk * ((r3 == c3 ? cc : 0.0f) - cm[r3] * cm[c3]);
            h[0] += mk * cm[0]; h[1] += mk * cm[1]; h[2] += mk * cm[2];
            mass += mk;
        }
    }
    float T[9], hy[3];
    dw::m3m(R, A, T);
    m3v(R, h, hy);
    const float xx = dot3(x, x), xh = dot3(x, hy);
    int o = 0;
    DQ_UNROLL for (int r3 = 0; r3 < 3; ++r3)
        DQ_UNROLL for (int c3 = r3; c3 < 3; ++c3) {
            float v = T[3 * r3] * R[3 * c3] + T[3 * r3 + 1] * R[3 * c3 + 1] + T[3 * r3 + 2] * R[3 * c3 + 2];
            v += (r3 == c3 ? mass * xx + 2.0f * xh : 0.0f) - mass * x[r3] * x[c3] - (x[r3] * hy[c3] + hy[r3] * x[c3]);
            Ao[o++] = v;
        }
    DQ_UNROLL for (int i = 0; i < 3; ++i) ho[i] = hy[i] + mass * x[i];
    *mass_out = mass;
}
DQ_HD float ao(const float *Ao, int r, int c) {      // symmetric 3x3 from 6 words
    return Ao[r <= c ? (r == 0 ? c : (r == 1 ? 2 + c : 5)) : (c == 0 ? r : (c == 1 ? 2 + r : 5))];
}
// IA += rigid inertia [[Ao, H], [H', m 1]], H = skew(ho);  pA += v x* (I v)
DQ_HD void add_rigid(float *IA, float *pA, const float *Ao, const float *ho, float mass, const float *v) {
    DQ_UNROLL for (int r = 0; r < 3; ++r)
        DQ_UNROLL for (int c = r; c < 3; ++c) IA[sym6(r, c)] += ao(Ao, r, c);
    IA[sym6(0, 4)] += -ho[2]; IA[sym6(0, 5)] += ho[1];
    IA[sym6(1, 3)] += ho[2];  IA[sym6(1, 5)] += -ho[0];
    IA[sym6(2, 3)] += -ho[1]; IA[sym6(2, 4)] += ho[0];
    IA[sym6(3, 3)] += mass; IA[sym6(4, 4)] += mass; IA[sym6(5, 5)] += mass;
    const float *om = v, *vl = v + 3;
    float n[3], f[3], t1[3], t2[3];
    DQ_UNROLL for (int r = 0; r < 3; ++r) n[r] = ao(Ao, r, 0) * om[0] + ao(Ao, r, 1) * om[1] + ao(Ao, r, 2) * om[2];
    cross3(ho, vl, t1);
    n[0] += t1[0]; n[1] += t1[1]; n[2] += t1[2];
    cross3(om, ho, t1);
    f[0] = t1[0] + mass * vl[0]; f[1] = t1[1] + mass * vl[1]; f[2] = t1[2] + mass * vl[2];
    cross3(om, n, t1); cross3(vl, f, t2);
    pA[0] += t1[0] + t2[0]; pA[1] += t1[1] + t2[1]; pA[2] += t1[2] + t2[2];
    cross3(om, f, t1);
    pA[3] += t1[0]; pA[4] += t1[1]; pA[5] += t1[2];
}

// Closest points a0 + sa da, b0 + sb db of two segments (Ericson, Real-Time Collision Detection 5.1.9), with the quotient of
// the nearly parallel case blended with the mid-overlap answer (written decision: oracle/dw_physics.c seg_seg).
DQ_HD void seg_seg(const float *da, const float *db, const float *r, float *so, float *to) {
    const float aa = dot3(da, da), ee = dot3(db, db), ff = dot3(db, r), eps = 1e-12f;
    float sa, sb;
    auto c01 = [](float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); };
    if (aa <= eps && ee <= eps) { sa = 0.0f; sb = 0.0f; }
    else if (aa <= eps) { sa = 0.0f; sb = c01(ff / ee); }
    else {
        const float cc = dot3(da, r);
        if (ee <= eps) { sb = 0.0f; sa = c01(-cc / aa); }
        else {
            const float bbv = dot3(da, db), den = aa * ee - bbv * bbv;
            float se = den > eps ? c01((bbv * ff - cc * ee) / den) : 0.0f;
            float te = (bbv * se + ff) / ee;
            if (te < 0.0f) { te = 0.0f; se = c01(-cc / aa); }
            else if (te > 1.0f) { te = 1.0f; se = c01((bbv - cc) / aa); }
            const float t0 = -cc / aa, t1 = t0 + bbv / aa;
            float lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
            if (lo < 0.0f) lo = 0.0f;
            if (hi > 1.0f) hi = 1.0f;
            const float sp = c01(0.5f * (lo + hi)), tp = c01((bbv * sp + ff) / ee), reg = 1e-3f * aa * ee;
            const float w = den > eps ? den * den / (den * den + reg * reg) : 0.0f;
            sa = w * se + (1.0f - w) * sp;
            sb = w * te + (1.0f - w) * tp;
        }
    }
    *so = sa; *to = sb;
}

// Squared least distance of two segments a(s) = r + s da, b(t) = t db (Ericson 5.1.9 without the blend), reciprocals by
// v_rcp_f32: for the self-collision DETECTION only (the force recomputes the touching pairs exactly).
DQ_HD float seg_dist2_fast(const float *da, const float *db, const float *r) {
    const float aa = dot3(da, da), ee = dot3(db, db), ff = dot3(db, r), cc = dot3(da, r), bbv = dot3(da, db);
    const float den = aa * ee - bbv * bbv;
    const float ia = rcp_fast(fmaxf(aa, 1e-12f)), ie = rcp_fast(fmaxf(ee, 1e-12f)), id = rcp_fast(fmaxf(den, 1e-12f));
    auto c01 = [](float x) { return fminf(fmaxf(x, 0.0f), 1.0f); };
    float se = den > 1e-12f ? c01((bbv * ff - cc * ee) * id) : 0.0f;
    float te = ee > 1e-12f ? (bbv * se + ff) * ie : -1.0f;        // b is a point: t = 0, s = the foot of the perpendicular
    const float s_lo = c01(-cc * ia), s_hi = c01((bbv - cc) * ia);
    se = te < 0.0f ? s_lo : (te > 1.0f ? s_hi : se);
    te = c01(te);
    float d2 = 0.0f;
    DQ_UNROLL for (int i = 0; i < 3; ++i) { const float n = r[i] + se * da[i] - te * db[i]; d2 = fmaf(n, n, d2); }
    return d2;
}

// Penalty force of two capsules (dw_physics.h K4b).  Returns true and fills F (force on A), pa, pb when they overlap.
DQ_HD bool capsule_pair(const float *a0, const float *a1, float ra, const float *b0, const float *b1, float rb, const float *va,
                        const float *vb, const PhysParams &P, float *F, float *pa, float *pb) {
    const float da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
    const float mid[3] = {0.5f * (a0[0] + a1[0] - b0[0] - b1[0]), 0.5f * (a0[1] + a1[1] - b0[1] - b1[1]), 0.5f * (a0[2] + a1[2] - b0[2] - b1[2])};
    const float reach = 0.5f * (sqrtf(dot3(da, da)) + sqrtf(dot3(db, db))) + ra + rb;
    if (!(dot3(mid, mid) <= reach * reach)) return false;
    const float r[3] = {a0[0] - b0[0], a0[1] - b0[1], a0[2] - b0[2]};
    float sa, sb;
    seg_seg(da, db, r, &sa, &sb);
    float n[3];
    DQ_UNROLL for (int i = 0; i < 3; ++i) { pa[i] = a0[i] + sa * da[i]; pb[i] = b0[i] + sb * db[i]; n[i] = pa[i] - pb[i]; }
    const float dist = sqrtf(dot3(n, n));
    const float depth = ra + rb - dist;
    if (!(depth > 0.0f && dist > 1e-6f)) return false;
    DQ_UNROLL for (int i = 0; i < 3; ++i) n[i] /= dist;
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

// ------------------------------------------------------------------------------------------------
// The substep.  On entry every body's slot holds quad 0 = {q, qd, tt, dd} with tt = tau - damping * qd and
// dd = armature + dt * damping (the caller's prologue), X.root the base state, X.warm the warm-start impulses, and the hot
// tables are staged (stage_hot).  On exit: slot quad 0 = {q, qd, *, *} of the new state, X.root, X.warm updated; with `last`,
// the net contact forces of the substep are written to B.contact_forces.  push: world x/y force on the base COM.
// ------------------------------------------------------------------------------------------------
struct FkHot { float pos[3], axis[3], vmax, qlo, qhi; int body, psrc, flags, scm; };
DQ_HD FkHot fk_hot(const QLds &L, int s, int j) {
    const F4 *r = reinterpret_cast<const F4 *>(L.hot.fk[s][j]);
    const F4 a = ld4(r[0]), b = ld4(r[1]), c = ld4(r[2]);
    FkHot h;
    h.pos[0] = a.x; h.pos[1] = a.y; h.pos[2] = a.z;
    const int bits = f2i(a.w);
    h.body = (bits & 255) == 255 ? -1 : (bits & 255);
    h.psrc = (bits >> 8) & 15; h.flags = (bits >> 12) & 3; h.scm = (bits >> 16) & 255;
    h.axis[0] = b.x; h.axis[1] = b.y; h.axis[2] = b.z; h.vmax = b.w;
    h.qlo = c.x; h.qhi = c.y;
    return h;
}

DQ_HD int sched_body(const QLds &L, int s, int j) {
    const int bits = f2i(L.hot.fk[s][j][3]);
    return (bits & 255) == 255 ? -1 : (bits & 255);
}

template <bool TERRAIN>
DQ_HD void quad_substep(QLds &L, const QuadModel &QM, const DevModel &M, const PhysParams &P, QLane &X, const DwBuffers &B,
                        float push_x, float push_y, bool last) {
    const float dt = P.dt, inv_dt = 1.0f / P.dt;
    const int j = X.j, T = L.hot.misc[0];
    const int SB = X.stamp_base; (void)SB;
    DQ_STAMP(B, SB + 0);
    const int e = X.env;
    const float *mscale_e = B.mass_scale + (size_t)DW_NUM_BODIES * e;

    // ---- base kinematics (every lane of the quad, redundantly) ----
    float qn[4], R0[9], ww[3], vo[3], bcom[3];
    {
        const float qx = X.root[3], qy = X.root[4], qz = X.root[5], qw = X.root[6];
        const float ninv = dw::rsqrt_nr(qx * qx + qy * qy + qz * qz + qw * qw);
        qn[0] = qx * ninv; qn[1] = qy * ninv; qn[2] = qz * ninv; qn[3] = qw * ninv;
        quat_to_mat(qn, R0);
        DQ_UNROLL for (int i = 0; i < 3; ++i) { ww[i] = X.root[10 + i]; vo[i] = X.root[7 + i]; bcom[i] = L.hot.base[i]; }
        if (P.vel_at_com) {
            float rc[3], t[3];
            m3v(R0, bcom, rc);
            cross3(ww, rc, t);
            vo[0] -= t[0]; vo[1] -= t[1]; vo[2] -= t[2];
        }
    }

    DQ_STAMP(B, SB + 1);
    // ---- outward pass 1: kinematics.  Running parent state: quaternion, rotation, origin, twist. ----
    float footR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, footx[3] = {0, 0, 0};      // pose of my sole body (lanes 0, 1)
    {
        float qr[4] = {0, 0, 0, 1}, Rr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, xr_[3] = {0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
        for (int s = 0; s < T; ++s) {
            const FkHot rc = fk_hot(L, s, j);
            const int b = rc.body, psrc = rc.psrc;
            // limbs that start below another lane's body fetch that lane's running state (still in its registers)
            float fq[4], fx[3], fv[6];
            bool fetched = false;
            const int fm = L.hot.fmask[s];
            if (fm) {
                for (int xl = 0; xl < 4; ++xl)
                    if ((fm >> xl) & 1) {
                        float tq[4], tx[3], tv[6];
                        quad_bcast_arr(xl, qr, tq); quad_bcast_arr(xl, xr_, tx); quad_bcast_arr(xl, vr, tv);
                        if (b >= 0 && psrc == 2 + xl) {
                            fetched = true;
                            DQ_UNROLL for (int i = 0; i < 4; ++i) fq[i] = tq[i];
                            DQ_UNROLL for (int i = 0; i < 3; ++i) fx[i] = tx[i];
                            DQ_UNROLL for (int i = 0; i < 6; ++i) fv[i] = tv[i];
                        }
                    }
            }
            if (b >= 0) {
                if (psrc == 1) {
                    DQ_UNROLL for (int i = 0; i < 4; ++i) qr[i] = qn[i];
                    DQ_UNROLL for (int i = 0; i < 9; ++i) Rr[i] = R0[i];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) { xr_[i] = 0.0f; vr[i] = ww[i]; vr[3 + i] = vo[i]; }
                } else if (fetched) {
                    DQ_UNROLL for (int i = 0; i < 4; ++i) qr[i] = fq[i];
                    quat_to_mat(qr, Rr);
                    DQ_UNROLL for (int i = 0; i < 3; ++i) xr_[i] = fx[i];
                    DQ_UNROLL for (int i = 0; i < 6; ++i) vr[i] = fv[i];
                }
                const F4 in = DQ_LD(b, 0, X.pos);            // {q, qd, tt, dd}
                float sn, cs;
                sincos_fast(0.5f * in.x, &sn, &cs);
                float qj[4] = {rc.axis[0] * sn, rc.axis[1] * sn, rc.axis[2] * sn, cs};
                if (rc.flags & 1) qmul(QM.fk[s][j].q0, qj, qj);          // (two bodies of the model: the hands)
                float x[3], t[3];
                m3v(Rr, rc.pos, t);
                DQ_UNROLL for (int i = 0; i < 3; ++i) x[i] = xr_[i] + t[i];
                qmul(qr, qj, qr);
                quat_to_mat(qr, Rr);
                float aw[3], sl[3];
                m3v(Rr, rc.axis, aw);
                cross3(x, aw, sl);
                DQ_UNROLL for (int i = 0; i < 3; ++i) { vr[i] += aw[i] * in.y; vr[3 + i] += sl[i] * in.y; xr_[i] = x[i]; }
                DQ_SLOT(b, 0, X.pos) = mk4(qr[0], qr[1], qr[2], qr[3]);
                DQ_SLOT(b, 1, X.pos) = mk4(x[0], x[1], x[2], in.y);
                DQ_SLOT(b, 2, X.pos) = mk4(vr[0], vr[1], vr[2], in.z);
                DQ_SLOT(b, 3, X.pos) = mk4(vr[3], vr[4], vr[5], in.w);
                if (rc.flags & 2) {
                    DQ_UNROLL for (int i = 0; i < 9; ++i) footR[i] = Rr[i];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) footx[i] = x[i];
                }
            }
        }
    }
    wave_sync();

    DQ_STAMP(B, SB + 2);
    // ---- self-collision: capsule proxies (legs, arms, torso), pairs from the model.  Detection: lane p & 3 evaluates proxy p
    //      from its body's slot; the pairs are tested in passes -- one proxy broadcast to the quad (DPP), every lane tests one
    //      of its own against it -- and the touching pairs of the env are ORed into a mask.  The common case is "nothing
    //      touches": then that is all.  Resolution, if any env of the wave has a touching pair: the lane that owns a proxy's
    //      body recomputes the proxy's touching pairs from the slots and keeps the wrench (both sides of a pair compute the
    //      same force from the same data: no hand-over between lanes). ----
    bool sc_any = false;
    float scW[QMAX_OWN][6];              // wrench (common frame) on my k-th own proxy: [0..2] moment, [3..5] force
    int scGym0 = 0;                      // Gym body of my k-th own proxy, one byte each (filled for the loaded ones)
    DQ_UNROLL for (int p = 0; p < QMAX_OWN; ++p) { DQ_UNROLL for (int i = 0; i < 6; ++i) scW[p][i] = 0.0f; }
    const int nprox = L.hot.misc[2], ncombo = L.hot.misc[3] & 255;
    auto proxy_bits = [&](int p) { return f2i(L.hot.prox[p][7]); };
    auto proxy_ends = [&](int p, float *p0w, float *p1w) {       // end points of proxy p in the common frame, from its body's slot
        const F4 *pr = reinterpret_cast<const F4 *>(L.hot.prox[p]);
        const F4 c0 = ld4(pr[0]), c1 = ld4(pr[1]);
        const int bits = f2i(c1.w);
        const int bp = bits & 255, posp = (X.el + 4 * ((bits >> 16) & 3)) & 15;
        const F4 q4 = DQ_LD(bp, 0, posp), x4 = DQ_LD(bp, 1, posp);
        const float qb[4] = {q4.x, q4.y, q4.z, q4.w}, l0[3] = {c0.x, c0.y, c0.z}, l1[3] = {c1.x, c1.y, c1.z};
        float Rb[9], t0[3], t1[3];
        quat_to_mat(qb, Rb);
        m3v(Rb, l0, t0);
        m3v(Rb, l1, t1);
        p0w[0] = x4.x + t0[0]; p0w[1] = x4.y + t0[1]; p0w[2] = x4.z + t0[2];
        p1w[0] = x4.x + t1[0]; p1w[1] = x4.y + t1[1]; p1w[2] = x4.z + t1[2];
    };
    if (P.self_collision && ncombo > 0) {
        float Pe[4][7];                    // my proxies (register set r = proxy j + 4 r): p0, p1 - p0, radius
        DQ_UNROLL for (int r = 0; r < 4; ++r) {
            DQ_UNROLL for (int i = 0; i < 7; ++i) Pe[r][i] = 0.0f;
            if (j + 4 * r < nprox) {
                float e1[3];
                proxy_ends(j + 4 * r, &Pe[r][0], e1);
                DQ_UNROLL for (int i = 0; i < 3; ++i) Pe[r][3 + i] = e1[i] - Pe[r][i];
                Pe[r][6] = L.hot.prox[j + 4 * r][3];
            }
        }
        DQ_STAMP(B, 51);
        int hits = 0;
        for (int c = 0; c < ncombo; ++c) {
            const int *cw = L.hot.combo[c];
            const int c0 = cw[0], pb = c0 & 255;
            float bq[7];
            // proxy pb to the whole quad: register set and lane are wave-uniform, so this is a scalar branch around 7 DPP moves
#define DQ_BC(R_, L_) case (R_) * 4 + (L_): { DQ_UNROLL for (int i = 0; i < 7; ++i) bq[i] = quad_bcast<L_>(Pe[R_][i]); } break;
            switch (pb) {
                DQ_BC(0, 0) DQ_BC(0, 1) DQ_BC(0, 2) DQ_BC(0, 3) DQ_BC(1, 0) DQ_BC(1, 1) DQ_BC(1, 2) DQ_BC(1, 3)
                DQ_BC(2, 0) DQ_BC(2, 1) DQ_BC(2, 2) DQ_BC(2, 3) DQ_BC(3, 0) DQ_BC(3, 1) DQ_BC(3, 2)
                default: { DQ_UNROLL for (int i = 0; i < 7; ++i) bq[i] = quad_bcast<3>(Pe[3][i]); } break;
            }
#undef DQ_BC
            DQ_UNROLL for (int r = 0; r < 4; ++r) {
                const int lm = (c0 >> (8 + 4 * r)) & 15;
                if (lm == 0) continue;
                if ((lm >> j) & 1) {
                    const float rr = Pe[r][6] + bq[6];
                    const float rv[3] = {Pe[r][0] - bq[0], Pe[r][1] - bq[1], Pe[r][2] - bq[2]};
                    // conservative: the least distance of the axes (the force uses the blended points, never closer), 0.2 % slack
                    if (seg_dist2_fast(&Pe[r][3], &bq[3], rv) < 1.004f * rr * rr) hits |= 1 << ((cw[1 + r] >> (8 * j)) & 255);
                }
            }
        }
        {   // the env's mask: OR over the quad (bit patterns through the DPP moves)
            int m = hits;
            m |= f2i(quad_xor1(__builtin_bit_cast(float, m)));
            m |= f2i(quad_xor2(__builtin_bit_cast(float, m)));
            hits = m;
        }
        sc_any = wave_any(hits != 0);
        DQ_STAMP(B, 52);
#if defined(DQ_STAMPS) && defined(__HIPCC__)
        if (blockIdx.x == 0 && threadIdx.x == 0) B.gate_acc[200 + 53] = sc_any;
#endif
        if (sc_any) {
            // every lane works through the touching pairs that involve one of its bodies
            int mine = hits & (j == 0 ? L.hot.misc[4] : (j == 1 ? L.hot.misc[5] : (j == 2 ? L.hot.misc[6] : L.hot.misc[7])));
            while (wave_any(mine != 0)) {
                if (mine != 0) {
                    const int pid = __builtin_ctz(mine);
                    mine &= mine - 1;
                    const int pr = (L.hot.pairs[pid >> 2] >> (8 * (pid & 3))) & 255, pa = pr & 15, pbx = pr >> 4;
                    const int bita = proxy_bits(pa), bitb = proxy_bits(pbx);
                    float a0[3], a1[3], b0[3], b1[3];
                    proxy_ends(pa, a0, a1);
                    proxy_ends(pbx, b0, b1);
                    const int ba = bita & 255, posa = (X.el + 4 * ((bita >> 16) & 3)) & 15;
                    const int bb = bitb & 255, posb = (X.el + 4 * ((bitb >> 16) & 3)) & 15;
                    const F4 va2 = DQ_LD(ba, 2, posa), va3 = DQ_LD(ba, 3, posa), vb2 = DQ_LD(bb, 2, posb), vb3 = DQ_LD(bb, 3, posb);
                    const float va[6] = {va2.x, va2.y, va2.z, va3.x, va3.y, va3.z}, vb[6] = {vb2.x, vb2.y, vb2.z, vb3.x, vb3.y, vb3.z};
                    float F[3], ca[3], cb[3];
                    if (capsule_pair(a0, a1, L.hot.prox[pa][3], b0, b1, L.hot.prox[pbx][3], va, vb, P, F, ca, cb)) {
                        DQ_UNROLL for (int side = 0; side < 2; ++side) {
                            const int bits = side ? bitb : bita;
                            if (((bits >> 16) & 3) != j) continue;
                            const int k = (bits >> 18) & 7, gy = (bits >> 8) & 255;
                            const float sg = side ? -1.0f : 1.0f;
                            const float Fs[3] = {sg * F[0], sg * F[1], sg * F[2]};
                            float nb[3];
                            cross3(side ? cb : ca, Fs, nb);
                            DQ_UNROLL for (int kk = 0; kk < QMAX_OWN; ++kk)
                                if (kk == k) { DQ_UNROLL for (int i = 0; i < 3; ++i) { scW[kk][i] += nb[i]; scW[kk][3 + i] += Fs[i]; } }
                            scGym0 |= gy << (8 * k);
                        }
                    }
                }
            }
        }
    }
    wave_sync();      // the leg lanes read each other's slots above; the inward pass below overwrites them

    DQ_STAMP(B, SB + 3);
    // ---- inward pass: articulated inertias and bias forces, in reverse schedule order ----
    float IA[21], pA[6], IP[21], pP[6];          // running and parked reflected inertia / bias
    DQ_UNROLL for (int i = 0; i < 21; ++i) { IA[i] = 0.0f; IP[i] = 0.0f; }
    DQ_UNROLL for (int i = 0; i < 6; ++i) { pA[i] = 0.0f; pP[i] = 0.0f; }
    X.footF[0] = X.footF[1] = X.footF[2] = 0.0f;
    const int my_sole_gym = (j == 0) ? M.left_foot_gym : (j == 1 ? M.right_foot_gym : -1);
    // the one per-env global value a step needs (the mass scale of the body's Gym body) is requested a step ahead
    float ms_next = mscale_e[(f2i(L.hot.in[0][j][2]) >> 24) & 255];
    for (int s = 0; s < T; ++s) {
#if defined(DQ_STAMPS_INWARD)
        if (SB == 1) DQ_STAMP(B, 42 + s);
#endif
        const F4 *hr = reinterpret_cast<const F4 *>(L.hot.in[s][j]);
        const F4 h0 = ld4(hr[0]), h1 = ld4(hr[1]), h2 = ld4(hr[2]), h3 = ld4(hr[3]);
        const int bits = f2i(h0.x);
        const int b = (bits & 255) - 1;
        const int flags = b >= 0 ? ((bits >> 8) & 7) : 0;
        const int nin = (bits >> 12) & 3, ngym = (bits >> 14) & 3, ngeom = (bits >> 16) & 15, scm = (bits >> 24) & 255;
        const int gymbits = f2i(h0.z);
        const float ms0 = ms_next;
        if (s + 1 < T) ms_next = mscale_e[(f2i(L.hot.in[s + 1][j][2]) >> 24) & 255];
        if (flags & 2) {                    // a finished chain is still waiting for its parent: park it
            DQ_UNROLL for (int i = 0; i < 21; ++i) IP[i] = IA[i];
            DQ_UNROLL for (int i = 0; i < 6; ++i) pP[i] = pA[i];
        }
        if (flags & 1) {
            DQ_UNROLL for (int i = 0; i < 21; ++i) IA[i] = 0.0f;
            DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] = 0.0f;
        }
        // gathers (wave-uniform per step): child chains that ended on other lanes
        if (L.hot.gany[s]) {
            const int g0 = f2i(L.hot.in[s][0][1]), g1 = f2i(L.hot.in[s][1][1]), g2 = f2i(L.hot.in[s][2][1]), g3 = f2i(L.hot.in[s][3][1]);
            const int mine = f2i(h0.y);
            DQ_UNROLL for (int src = 0; src < 4; ++src)
                DQ_UNROLL for (int pk = 0; pk < 2; ++pk) {
                    const int code = src | (pk << 2) | 8;
                    bool used = false, want = false;
                    DQ_UNROLL for (int k = 0; k < 3; ++k) {
                        used = used || (((g0 >> (4 * k)) & 15) == code) || (((g1 >> (4 * k)) & 15) == code) ||
                               (((g2 >> (4 * k)) & 15) == code) || (((g3 >> (4 * k)) & 15) == code);
                        want = want || (((mine >> (4 * k)) & 15) == code);
                    }
                    if (used) {
                        DQ_UNROLL for (int i = 0; i < 21; ++i) {
                            const float v = pk ? IP[i] : IA[i];
                            const float t = src == 0 ? quad_bcast<0>(v) : (src == 1 ? quad_bcast<1>(v) : (src == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
                            if (want) IA[i] += t;
                        }
                        DQ_UNROLL for (int i = 0; i < 6; ++i) {
                            const float v = pk ? pP[i] : pA[i];
                            const float t = src == 0 ? quad_bcast<0>(v) : (src == 1 ? quad_bcast<1>(v) : (src == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
                            if (want) pA[i] += t;
                        }
                    }
                }
        }
        if (b >= 0) {
            const F4 s0 = DQ_LD(b, 0, X.pos), s1 = DQ_LD(b, 1, X.pos), s2 = DQ_LD(b, 2, X.pos), s3 = DQ_LD(b, 3, X.pos);
            const F4 ax4 = ld4(reinterpret_cast<const F4 *>(L.hot.fk[T - 1 - s][j])[1]);
            const float axis[3] = {ax4.x, ax4.y, ax4.z};
            const float qb[4] = {s0.x, s0.y, s0.z, s0.w}, x[3] = {s1.x, s1.y, s1.z}, v[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
            const float qd = s1.w, tt = s2.w, dd = s3.w;
            float R[9];
            quat_to_mat(qb, R);
            float S[6];
            m3v(R, axis, S);
            cross3(x, S, S + 3);
            // rigid inertia, gyroscopic bias
            float Ao[6], ho[3], mass;
            {
                const float com0[3] = {h1.x, h1.y, h1.z}, I0[6] = {h2.x, h2.y, h2.z, h2.w, h3.x, h3.y};
                if (nin > 1) {          // the two sole bodies carry a second (welded) inertial record
                    rigid_inertia(2, com0, h1.w, I0, ms0, &X.in1[0], X.in1[3], &X.in1[4], X.ms1, R, x, Ao, ho, &mass);
                } else {
                    rigid_inertia(1, com0, h1.w, I0, ms0, com0, 0.0f, I0, 0.0f, R, x, Ao, ho, &mass);
                }
            }
            add_rigid(IA, pA, Ao, ho, mass, v);
            // external forces: ground penalty of the non-sole primitives, self-collision; per Gym body for the report
            float cf[QMAX_GYM][3];
            DQ_UNROLL for (int t = 0; t < QMAX_GYM; ++t) cf[t][0] = cf[t][1] = cf[t][2] = 0.0f;
            bool near_ground = ngeom > 0 && (X.root[2] + x[2] < h0.w);
            if (TERRAIN) near_ground = ngeom > 0;
#if defined(DQ_KO_GEOM)          // (timing experiment only)
            near_ground = false;
#endif
            if (near_ground) {
                const QInRec &rc = QM.in[s][j];
                for (int k = 0; k < ngeom; ++k) {
                    float F[3], xr[3];
                    geom_force<TERRAIN>(M.geoms[rc.geom[k]], P, R, x, v, X.root[0], X.root[1], X.root[2], X.mu, F, xr);
                    if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                        float nb[3];
                        cross3(xr, F, nb);
                        DQ_UNROLL for (int i = 0; i < 3; ++i) { pA[i] -= nb[i]; pA[3 + i] -= F[i]; }
                        const int t = (rc.geom_slot >> (2 * k)) & 3;
                        DQ_UNROLL for (int tt2 = 0; tt2 < QMAX_GYM; ++tt2)
                            if (tt2 == t) { cf[tt2][0] += F[0]; cf[tt2][1] += F[1]; cf[tt2][2] += F[2]; }
                    }
                }
            }
            if (sc_any && scm) {
                DQ_UNROLL for (int p = 0; p < QMAX_OWN; ++p)
                    if ((scm >> p) & 1) {
                        DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] -= scW[p][i];
                        const int gy = (scGym0 >> (8 * p)) & 255;     // (0 where unloaded: adds nothing)
                        DQ_UNROLL for (int t = 0; t < QMAX_GYM; ++t)
                            if (t < ngym && ((gymbits >> (8 * t)) & 255) == gy) { cf[t][0] += scW[p][3]; cf[t][1] += scW[p][4]; cf[t][2] += scW[p][5]; }
                    }
            }
            if (last) {
                DQ_UNROLL for (int t = 0; t < QMAX_GYM; ++t)
                    if (t < ngym) {
                        const int gy = (gymbits >> (8 * t)) & 255;
                        if (gy == my_sole_gym) { X.footF[0] = cf[t][0]; X.footF[1] = cf[t][1]; X.footF[2] = cf[t][2]; }
                        else {
                            if (over_1n(cf[t])) X.coll = 1;
                            if (X.valid) {
                                float *dst = B.contact_forces + ((size_t)DW_NUM_BODIES * e + gy) * 3;
                                dst[0] = cf[t][0]; dst[1] = cf[t][1]; dst[2] = cf[t][2];
                            }
                        }
                    }
            }
            // articulated-body step
            float U[6];
            DQ_UNROLL for (int r = 0; r < 6; ++r) {
                float acc = 0.0f;
                DQ_UNROLL for (int c = 0; c < 6; ++c) acc += IA[sym6(r, c)] * S[c];
                U[r] = acc;
            }
            const float D = dot6(S, U) + dd;
            const float Dinv = dw::rcp_nr(D);
            const float u = tt - dot6(S, pA);
            float m[6], cb[6];
            DQ_UNROLL for (int i = 0; i < 6; ++i) m[i] = S[i] * qd;
            dw::motion_cross(v, m, cb);
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
            DQ_SLOT(b, 0, X.pos) = mk4(S[0], S[1], S[2], Dinv);
            DQ_SLOT(b, 1, X.pos) = mk4(S[3], S[4], S[5], u);
            DQ_SLOT(b, 2, X.pos) = mk4(U[0], U[1], U[2], qd);
            DQ_SLOT(b, 3, X.pos) = mk4(U[3], U[4], U[5], 0.0f);
        }
    }

    DQ_STAMP(B, SB + 4);
    // ---- base: gather the chains below the root, own inertia, external forces, inverse ----
    float Minv[36], a0[6];
    {
        float I0[21], p0[6];
        DQ_UNROLL for (int i = 0; i < 21; ++i) I0[i] = 0.0f;
        DQ_UNROLL for (int i = 0; i < 6; ++i) p0[i] = 0.0f;
        const int g = L.hot.misc[1];
        DQ_UNROLL for (int src = 0; src < 4; ++src)
            DQ_UNROLL for (int pk = 0; pk < 2; ++pk) {
                const int code = src | (pk << 2) | 8;
                const bool used = ((g & 15) == code) || (((g >> 4) & 15) == code) || (((g >> 8) & 15) == code) || (((g >> 12) & 15) == code);
                if (used) {
                    DQ_UNROLL for (int i = 0; i < 21; ++i) {
                        const float v = pk ? IP[i] : IA[i];
                        I0[i] += src == 0 ? quad_bcast<0>(v) : (src == 1 ? quad_bcast<1>(v) : (src == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
                    }
                    DQ_UNROLL for (int i = 0; i < 6; ++i) {
                        const float v = pk ? pP[i] : pA[i];
                        p0[i] += src == 0 ? quad_bcast<0>(v) : (src == 1 ? quad_bcast<1>(v) : (src == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
                    }
                }
            }
        const float v0[6] = {ww[0], ww[1], ww[2], vo[0], vo[1], vo[2]}, x0[3] = {0, 0, 0};
        float Ao[6], ho[3], mass;
        const int base_gym = f2i(L.hot.base[10]), base_ngeom = f2i(L.hot.base[11]);
        const float ms = mscale_e[base_gym];
        {
            const float bI[6] = {L.hot.base[4], L.hot.base[5], L.hot.base[6], L.hot.base[7], L.hot.base[8], L.hot.base[9]};
            rigid_inertia(1, bcom, L.hot.base[3], bI, ms, bcom, 0.0f, bI, 0.0f, R0, x0, Ao, ho, &mass);
        }
        add_rigid(I0, p0, Ao, ho, mass, v0);
        float cfb[3] = {0, 0, 0};
        bool near_ground = base_ngeom > 0 && (X.root[2] < L.hot.base[12]);
        if (TERRAIN) near_ground = base_ngeom > 0;
        if (near_ground) {
            for (int k = 0; k < base_ngeom; ++k) {
                float F[3], xr[3];
                geom_force<TERRAIN>(M.geoms[QM.base_geom[k]], P, R0, x0, v0, X.root[0], X.root[1], X.root[2], X.mu, F, xr);
                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                    float nb[3];
                    cross3(xr, F, nb);
                    DQ_UNROLL for (int i = 0; i < 3; ++i) { p0[i] -= nb[i]; p0[3 + i] -= F[i]; cfb[i] += F[i]; }
                }
            }
        }
        if (last && j == 3) {
            if (over_1n(cfb)) X.coll = 1;
            if (X.valid) {
                float *dst = B.contact_forces + ((size_t)DW_NUM_BODIES * e + base_gym) * 3;
                dst[0] = cfb[0]; dst[1] = cfb[1]; dst[2] = cfb[2];
            }
        }
        {   // push on the base COM
            const float Fw[3] = {push_x, push_y, 0.0f};
            float xc[3], nb[3];
            m3v(R0, bcom, xc);
            cross3(xc, Fw, nb);
            DQ_UNROLL for (int i = 0; i < 3; ++i) { p0[i] -= nb[i]; p0[3 + i] -= Fw[i]; }
        }
        // Cholesky I0 = L L', Minv by six pairs of triangular solves (dw_physics.h A3)
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
            DQ_UNROLL for (int i = 0; i < 6; ++i) Minv[6 * i + col] = xx[i];
        }
        DQ_UNROLL for (int r = 0; r < 6; ++r) {
            float acc = 0.0f;
            DQ_UNROLL for (int c = 0; c < 6; ++c) acc -= Minv[6 * r + c] * p0[c];
            a0[r] = acc;
        }
    }

    DQ_STAMP(B, SB + 5);
    // ---- outward pass 2: accelerations; free joint velocities qdf = qd + dt qdd into the slot ----
    {
        float ar[6] = {0, 0, 0, 0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
        for (int s = 0; s < T; ++s) {
            const int bits = f2i(L.hot.fk[s][j][3]);
            const int b = (bits & 255) == 255 ? -1 : (bits & 255), psrc = (bits >> 8) & 15;
            float fa[6], fv[6];
            bool fetched = false;
            const int fm = L.hot.fmask[s];
            if (fm) {
                for (int xl = 0; xl < 4; ++xl)
                    if ((fm >> xl) & 1) {
                        float ta[6], tv[6];
                        quad_bcast_arr(xl, ar, ta); quad_bcast_arr(xl, vr, tv);
                        if (b >= 0 && psrc == 2 + xl) { fetched = true; DQ_UNROLL for (int i = 0; i < 6; ++i) { fa[i] = ta[i]; fv[i] = tv[i]; } }
                    }
            }
            if (b >= 0) {
                if (psrc == 1) { DQ_UNROLL for (int i = 0; i < 3; ++i) { ar[i] = a0[i]; ar[3 + i] = a0[3 + i]; vr[i] = ww[i]; vr[3 + i] = vo[i]; } }
                else if (fetched) { DQ_UNROLL for (int i = 0; i < 6; ++i) { ar[i] = fa[i]; vr[i] = fv[i]; } }
                const F4 s0 = DQ_LD(b, 0, X.pos), s1 = DQ_LD(b, 1, X.pos), s2 = DQ_LD(b, 2, X.pos), s3 = DQ_LD(b, 3, X.pos);
                const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
                const float Dinv = s0.w, u = s1.w, qd = s2.w;
                float m[6], c[6];
                DQ_UNROLL for (int i = 0; i < 6; ++i) m[i] = S[i] * qd;
                dw::motion_cross(vr, m, c);                  // parent twist x S qd  (= body twist x S qd)
                DQ_UNROLL for (int i = 0; i < 6; ++i) { ar[i] += c[i]; vr[i] += m[i]; }
                const float qdd = (u - dot6(U, ar)) * Dinv;
                DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] += S[i] * qdd;
                DQ_SLOT(b, 2, X.pos) = mk4(U[0], U[1], U[2], qd + dt * qdd);      // free velocity
                DQ_SLOT(b, 1, X.pos) = mk4(S[3], S[4], S[5], 0.0f);                // u is dead: the word becomes the impulse-sweep d
            }
        }
    }
    wave_sync();
    DQ_STAMP(B, SB + 6);
    // ---- free base velocity; sole-corner gaps of my foot (foot f = j & 1; lanes f and f + 2 work on it together) ----
    float wwf[3], vowf[3];
    {
        float t2[3];
        cross3(ww, vo, t2);
        DQ_UNROLL for (int i = 0; i < 3; ++i) {
            wwf[i] = ww[i] + dt * a0[i];
            vowf[i] = vo[i] + dt * (a0[3 + i] + t2[i] + P.g[i]);
        }
    }
    const int f = j & 1, part = j >> 1;
    float rk[4][3], vminr[4], frame[4][9];
    int act[4];
    {
        // the pose of foot f lives in lane f: lanes 2, 3 fetch it from their partner (l ^ 2)
        float fR[9], fx[3];
        DQ_UNROLL for (int i = 0; i < 9; ++i) { const float o = quad_xor2(footR[i]); fR[i] = part ? o : footR[i]; }
        DQ_UNROLL for (int i = 0; i < 3; ++i) { const float o = quad_xor2(footx[i]); fx[i] = part ? o : footx[i]; }
        DQ_UNROLL for (int k = 0; k < 4; ++k) {
            float r[3];
            m3v(fR, M.foot_pos[4 * f + k], r);
            DQ_UNROLL for (int i = 0; i < 3; ++i) r[i] += fx[i];
            float phi = X.root[2] + r[2];
            if (TERRAIN) {
                float hh;
                dw::terrain_sample(P, X.root[0] + r[0], X.root[1] + r[1], &hh, frame[k]);
                phi = (phi - hh) * frame[k][8];
            } else {
                DQ_UNROLL for (int i = 0; i < 9; ++i) frame[k][i] = (i % 4 == 0) ? 1.0f : 0.0f;
            }
            act[k] = phi < P.contact_offset;
            DQ_UNROLL for (int i = 0; i < 3; ++i) rk[k][i] = r[i];
            vminr[k] = phi >= 0 ? -phi * inv_dt : fminf(P.erp * (-phi) * inv_dt, P.max_depen);
        }
    }
    const bool any_active = wave_any(act[0] | act[1] | act[2] | act[3]);
    float dqb[6] = {0, 0, 0, 0, 0, 0};              // base velocity jump
    float Pk[4][3];
    DQ_UNROLL for (int k = 0; k < 4; ++k) DQ_UNROLL for (int i = 0; i < 3; ++i) Pk[k][i] = act[k] ? X.warm[3 * k + i] : 0.0f;

    if (any_active) {
        // ---- free twist of foot f: base + sum over the leg of S qdf (leg lane), shared with the partner ----
        float twf[6];
        {
            float acc[6] = {wwf[0], wwf[1], wwf[2], vowf[0], vowf[1], vowf[2]};
            const int posf = (X.el + 4 * f) & 15;
            DQ_UNROLL for (int i = 1; i <= 6; ++i) {
                const int b = 6 * f + i;
                const F4 s0 = DQ_LD(b, 0, posf), s1 = DQ_LD(b, 1, posf), s2 = DQ_LD(b, 2, posf);
                acc[0] += s0.x * s2.w; acc[1] += s0.y * s2.w; acc[2] += s0.z * s2.w;
                acc[3] += s1.x * s2.w; acc[4] += s1.y * s2.w; acc[5] += s1.z * s2.w;
            }
            DQ_UNROLL for (int i = 0; i < 6; ++i) twf[i] = acc[i];
        }
        DQ_STAMP(B, SB + 7);
        // ---- my 3 rows of W: responses of both feet to unit wrenches (components 3 part .. 3 part + 2) on foot f.
        //      Up the leg: d = -S'p, p += U d / D;  base: dv = -Minv p;  down both legs: qdd = (d - U'dv) / D, dv += S qdd ----
        float Wr[3][12];
        {
            float dp[3][6], dc[3][6];
            DQ_UNROLL for (int c = 0; c < 3; ++c) DQ_UNROLL for (int i = 0; i < 6; ++i) dp[c][i] = (i == 3 * part + c) ? -1.0f : 0.0f;
            const int posf = (X.el + 4 * f) & 15;
            DQ_UNROLL for (int i = 6; i >= 1; --i) {
                const int b = 6 * f + i;
                const F4 s0 = DQ_LD(b, 0, posf), s1 = DQ_LD(b, 1, posf), s2 = DQ_LD(b, 2, posf), s3 = DQ_LD(b, 3, posf);
                const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
                DQ_UNROLL for (int c = 0; c < 3; ++c) {
                    const float d = -dot6(S, dp[c]);
                    dc[c][i - 1] = d;
                    const float k = d * s0.w;
                    DQ_UNROLL for (int r = 0; r < 6; ++r) dp[c][r] += U[r] * k;
                }
            }
            float dv0[3][6];
            DQ_UNROLL for (int c = 0; c < 3; ++c)
                DQ_UNROLL for (int r = 0; r < 6; ++r) {
                    float acc = 0.0f;
                    DQ_UNROLL for (int k = 0; k < 6; ++k) acc -= Minv[6 * r + k] * dp[c][k];
                    dv0[c][r] = acc;
                }
            DQ_UNROLL for (int g = 0; g < 2; ++g) {
                float dv[3][6];
                DQ_UNROLL for (int c = 0; c < 3; ++c) DQ_UNROLL for (int r = 0; r < 6; ++r) dv[c][r] = dv0[c][r];
                const int posg = (X.el + 4 * g) & 15;
                DQ_UNROLL for (int i = 1; i <= 6; ++i) {
                    const int b = 6 * g + i;
                    const F4 s0 = DQ_LD(b, 0, posg), s1 = DQ_LD(b, 1, posg), s2 = DQ_LD(b, 2, posg), s3 = DQ_LD(b, 3, posg);
                    const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
                    DQ_UNROLL for (int c = 0; c < 3; ++c) {
                        const float ua = dot6(U, dv[c]);
                        const float qdd = ((g == f ? dc[c][i - 1] : 0.0f) - ua) * s0.w;
                        DQ_UNROLL for (int r = 0; r < 6; ++r) dv[c][r] += S[r] * qdd;
                    }
                }
                DQ_UNROLL for (int c = 0; c < 3; ++c) DQ_UNROLL for (int r = 0; r < 6; ++r) Wr[c][6 * g + r] = dv[c][r];
            }
        }
        DQ_STAMP(B, SB + 8);
        // ---- 3x3 diagonal blocks of the Delassus matrix of my foot's corners: A_kk = J_k W_ff J_k' (frame-projected on
        //      terrain); what the solver needs of them: the three diagonal inverses and the couplings zx, zy, xy ----
        float invd[4][3], cpl[4][3];
        {
            float Wff[6][6];       // rows 3 part.. are mine, the other three come from the partner lane (l ^ 2)
            DQ_UNROLL for (int r = 0; r < 3; ++r)
                DQ_UNROLL for (int c = 0; c < 6; ++c) {
                    const float mine = f ? Wr[r][6 + c] : Wr[r][c];
                    const float o = quad_xor2(mine);
                    Wff[r][c] = part ? o : mine;
                    Wff[3 + r][c] = part ? mine : o;
                }
            const float rreg = 1.0f / (1.0f + P.cfm);
            DQ_UNROLL for (int k = 0; k < 4; ++k) {
                const float *r = rk[k];
                float G[3][6];        // J_k W_ff, world axes: row a = W_lin row a + (-skew(r)) row a . W_ang
                DQ_UNROLL for (int c = 0; c < 6; ++c) {
                    G[0][c] = Wff[3][c] + r[2] * Wff[1][c] - r[1] * Wff[2][c];
                    G[1][c] = Wff[4][c] - r[2] * Wff[0][c] + r[0] * Wff[2][c];
                    G[2][c] = Wff[5][c] + r[1] * Wff[0][c] - r[0] * Wff[1][c];
                }
                float Ak[3][3];       // G J_k': column b = G_lin col b + G_ang . (-skew(r)) row b
                DQ_UNROLL for (int a = 0; a < 3; ++a) {
                    Ak[a][0] = G[a][3] + r[2] * G[a][1] - r[1] * G[a][2];
                    Ak[a][1] = G[a][4] - r[2] * G[a][0] + r[0] * G[a][2];
                    Ak[a][2] = G[a][5] + r[1] * G[a][0] - r[0] * G[a][1];
                }
                if (TERRAIN) {        // rows / columns along the corner's frame (t1, t2, n)
                    float Tm[3][3];
                    DQ_UNROLL for (int a = 0; a < 3; ++a) DQ_UNROLL for (int c = 0; c < 3; ++c)
                        Tm[a][c] = frame[k][3 * a] * Ak[0][c] + frame[k][3 * a + 1] * Ak[1][c] + frame[k][3 * a + 2] * Ak[2][c];
                    DQ_UNROLL for (int a = 0; a < 3; ++a) DQ_UNROLL for (int c = 0; c < 3; ++c)
                        Ak[a][c] = Tm[a][0] * frame[k][3 * c] + Tm[a][1] * frame[k][3 * c + 1] + Tm[a][2] * frame[k][3 * c + 2];
                }
                invd[k][0] = act[k] ? dw::rcp_nr(Ak[0][0]) * rreg : 0.0f;
                invd[k][1] = act[k] ? dw::rcp_nr(Ak[1][1]) * rreg : 0.0f;
                invd[k][2] = act[k] ? dw::rcp_nr(Ak[2][2]) * rreg : 0.0f;
                cpl[k][0] = Ak[0][2];     // x row, z column
                cpl[k][1] = Ak[1][2];     // y row, z column
                cpl[k][2] = Ak[1][0];     // y row, x column
            }
        }
        // my rows of W as [own foot | other foot] so that the updates below index registers statically
        float Wo[3][6], Wx[3][6];
        DQ_UNROLL for (int r = 0; r < 3; ++r) DQ_UNROLL for (int c = 0; c < 6; ++c) { Wo[r][c] = f ? Wr[r][6 + c] : Wr[r][c]; Wx[r][c] = f ? Wr[r][c] : Wr[r][6 + c]; }
        // ---- start: tw = tw_free + W lambda0, lambda0 = warm-start impulses as foot wrenches ----
        float tw3[3];
        {
            float lam[6] = {0, 0, 0, 0, 0, 0};
            DQ_UNROLL for (int k = 0; k < 4; ++k) {
                float pw[3] = {Pk[k][0], Pk[k][1], Pk[k][2]};
                if (TERRAIN) {
                    const float p0 = pw[0], p1 = pw[1], p2 = pw[2];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) pw[i] = p0 * frame[k][i] + p1 * frame[k][3 + i] + p2 * frame[k][6 + i];
                }
                float t[3];
                cross3(rk[k], pw, t);
                DQ_UNROLL for (int i = 0; i < 3; ++i) { lam[i] += t[i]; lam[3 + i] += pw[i]; }
            }
            float lo[6];
            DQ_UNROLL for (int i = 0; i < 6; ++i) lo[i] = quad_xor1(lam[i]);
            DQ_UNROLL for (int r = 0; r < 3; ++r) {
                float acc = part ? twf[3 + r] : twf[r];
                DQ_UNROLL for (int c = 0; c < 6; ++c) acc += Wo[r][c] * lam[c] + Wx[r][c] * lo[c];
                tw3[r] = acc;
            }
        }
        DQ_STAMP(B, SB + 9);
        // ---- projected Gauss-Seidel, block-Jacobi across the feet: corner kk of the left sole and corner kk of the right
        //      sole are updated together from the same snapshot, the four corners of a sole one after the other ----
        bool pair_on[4];
        DQ_UNROLL for (int kk = 0; kk < 4; ++kk) pair_on[kk] = wave_any(act[kk] != 0);
        for (int it = 0; it < P.iters; ++it) {
            DQ_UNROLL for (int kk = 0; kk < 4; ++kk) {
                if (pair_on[kk]) {
                    float o3[3];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) o3[i] = quad_xor2(tw3[i]);
                    float wv[3], lv[3];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) { wv[i] = part ? o3[i] : tw3[i]; lv[i] = part ? tw3[i] : o3[i]; }
                    const float *r = rk[kk];
                    float vwld[3] = {lv[0] + wv[1] * r[2] - wv[2] * r[1], lv[1] + wv[2] * r[0] - wv[0] * r[2], lv[2] + wv[0] * r[1] - wv[1] * r[0]};
                    float vx0 = vwld[0], vy0 = vwld[1], vz = vwld[2];
                    if (TERRAIN) {
                        const float *fr = frame[kk];
                        vx0 = fr[0] * vwld[0] + fr[1] * vwld[1] + fr[2] * vwld[2];
                        vy0 = fr[3] * vwld[0] + fr[4] * vwld[1] + fr[5] * vwld[2];
                        vz = fr[6] * vwld[0] + fr[7] * vwld[1] + fr[8] * vwld[2];
                    }
                    const float Px = Pk[kk][0], Py = Pk[kk][1], Pz = Pk[kk][2];
                    float dz = -(vz - vminr[kk]) * invd[kk][2];
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
                    float d[3] = {px - Px, py - Py, dz};
                    Pk[kk][0] = px; Pk[kk][1] = py; Pk[kk][2] = pz;
                    if (TERRAIN) {
                        const float *fr = frame[kk];
                        const float d0 = d[0], d1 = d[1], d2 = d[2];
                        DQ_UNROLL for (int i = 0; i < 3; ++i) d[i] = d0 * fr[i] + d1 * fr[3 + i] + d2 * fr[6 + i];
                    }
                    float lam[6], lo[6];
                    cross3(r, d, lam);
                    lam[3] = d[0]; lam[4] = d[1]; lam[5] = d[2];
                    DQ_UNROLL for (int i = 0; i < 6; ++i) lo[i] = quad_xor1(lam[i]);
                    DQ_UNROLL for (int rr = 0; rr < 3; ++rr) {
                        float acc = tw3[rr];
                        DQ_UNROLL for (int c = 0; c < 6; ++c) acc += Wo[rr][c] * lam[c] + Wx[rr][c] * lo[c];
                        tw3[rr] = acc;
                    }
                }
            }
        }
        DQ_STAMP(B, SB + 10);
        // ---- impulses -> wrench on my foot -> up my leg (leg lanes), base jump ----
        float dpb[6] = {0, 0, 0, 0, 0, 0};
        float Fs[3] = {0, 0, 0};
        if (part == 0) {
            float Nm[3] = {0, 0, 0};
            DQ_UNROLL for (int k = 0; k < 4; ++k) {
                float pw[3] = {Pk[k][0], Pk[k][1], Pk[k][2]};
                if (TERRAIN) {
                    const float p0 = pw[0], p1 = pw[1], p2 = pw[2];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) pw[i] = p0 * frame[k][i] + p1 * frame[k][3 + i] + p2 * frame[k][6 + i];
                }
                float t[3];
                cross3(rk[k], pw, t);
                DQ_UNROLL for (int i = 0; i < 3; ++i) { Fs[i] += pw[i]; Nm[i] += t[i]; }
            }
            float dp[6] = {-Nm[0], -Nm[1], -Nm[2], -Fs[0], -Fs[1], -Fs[2]};
            DQ_UNROLL for (int i = 6; i >= 1; --i) {
                const int b = 6 * f + i;
                const F4 s0 = DQ_LD(b, 0, X.pos), s1 = DQ_LD(b, 1, X.pos), s2 = DQ_LD(b, 2, X.pos), s3 = DQ_LD(b, 3, X.pos);
                const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
                const float d = -dot6(S, dp);
                const float k = d * s0.w;
                DQ_UNROLL for (int r = 0; r < 6; ++r) dp[r] += U[r] * k;
                DQ_SLOT(b, 1, X.pos) = mk4(S[3], S[4], S[5], d);
            }
            DQ_UNROLL for (int i = 0; i < 6; ++i) dpb[i] = dp[i];
        }
        {
            float tot[6];
            DQ_UNROLL for (int i = 0; i < 6; ++i) {
                const float a = quad_bcast<0>(dpb[i]), b2 = quad_bcast<1>(dpb[i]);
                tot[i] = a + b2;
            }
            DQ_UNROLL for (int r = 0; r < 6; ++r) {
                float acc = 0.0f;
                DQ_UNROLL for (int c = 0; c < 6; ++c) acc -= Minv[6 * r + c] * tot[c];
                dqb[r] = acc;
            }
        }
        if (last && part == 0) {
            DQ_UNROLL for (int i = 0; i < 3; ++i) X.footT[i] = X.footF[i] + Fs[i] * inv_dt;
            if (X.valid) {
                float *dst = B.contact_forces + ((size_t)DW_NUM_BODIES * e + (f == 0 ? M.left_foot_gym : M.right_foot_gym)) * 3;
                dst[0] = X.footT[0]; dst[1] = X.footT[1]; dst[2] = X.footT[2];
            }
        }
    } else if (last && part == 0) {
        DQ_UNROLL for (int i = 0; i < 3; ++i) X.footT[i] = X.footF[i];
        if (X.valid) {
            float *dst = B.contact_forces + ((size_t)DW_NUM_BODIES * e + (f == 0 ? M.left_foot_gym : M.right_foot_gym)) * 3;
            dst[0] = X.footF[0]; dst[1] = X.footF[1]; dst[2] = X.footF[2];
        }
    }
    DQ_UNROLL for (int k = 0; k < 4; ++k) DQ_UNROLL for (int i = 0; i < 3; ++i) X.warm[3 * k + i] = Pk[k][i];

    DQ_STAMP(B, SB + 11);
    // ---- outward pass 3: velocity jumps down the tree, final joint velocities (speed limit); the caller integrates ----
    {
        float ar[6] = {0, 0, 0, 0, 0, 0};
        for (int s = 0; s < T; ++s) {
            const FkHot rc = fk_hot(L, s, j);
            const int b = rc.body, psrc = rc.psrc;
            float fa[6];
            bool fetched = false;
            const int fm = L.hot.fmask[s];
            if (fm) {
                for (int xl = 0; xl < 4; ++xl)
                    if ((fm >> xl) & 1) {
                        float ta[6];
                        quad_bcast_arr(xl, ar, ta);
                        if (b >= 0 && psrc == 2 + xl) { fetched = true; DQ_UNROLL for (int i = 0; i < 6; ++i) fa[i] = ta[i]; }
                    }
            }
            if (b >= 0) {
                if (psrc == 1) { DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] = dqb[i]; }
                else if (fetched) { DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] = fa[i]; }
                const F4 s0 = DQ_LD(b, 0, X.pos), s1 = DQ_LD(b, 1, X.pos), s2 = DQ_LD(b, 2, X.pos), s3 = DQ_LD(b, 3, X.pos);
                const float S[6] = {s0.x, s0.y, s0.z, s1.x, s1.y, s1.z}, U[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
                const float dq = (s1.w - dot6(U, ar)) * s0.w;
                DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] += S[i] * dq;
                float qd = s2.w + dq;
                if (qd > rc.vmax) qd = rc.vmax;
                if (qd < -rc.vmax) qd = -rc.vmax;
                DQ_SLOT(b, 0, X.pos) = mk4(rc.qlo, qd, rc.qhi, 0.0f);        // joint range and new velocity for integrate_joints
            }
        }
    }
    DQ_STAMP(B, SB + 12);
    // ---- base: final velocity, clamps, pose update (dw_physics.h V2) ----
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
            float Rn[9], rcom[3], tt[3];
            quat_to_mat(qo, Rn);
            m3v(Rn, bcom, rcom);
            cross3(wwn, rcom, tt);
            DQ_UNROLL for (int i = 0; i < 3; ++i) von[i] += tt[i];
        }
        DQ_UNROLL for (int i = 0; i < 3; ++i) { X.root[7 + i] = von[i]; X.root[10 + i] = wwn[i]; }
    }
    DQ_STAMP(B, SB + 13);
}

// Lane set-up shared by the entry points: which env this lane works for, its base state and parameters.
DQ_HD void quad_lane_init(QLane &X, int wave_index, int num_envs, const PhysParams &P, float friction, const DwBuffers &B) {
    X.lane = lane_id();
    X.j = X.lane & 3;
    X.el = X.lane >> 2;
    const int eg = wave_index * EPW + X.el;
    X.valid = eg < num_envs;
    X.env = X.valid ? eg : num_envs - 1;
    X.pos = (X.el + 4 * X.j) & 15;
    DQ_UNROLL for (int i = 0; i < 13; ++i) X.root[i] = B.root_states[(size_t)13 * X.env + i];
    X.mu = friction * B.friction_scale[X.env];
    DQ_UNROLL for (int i = 0; i < 12; ++i) X.warm[i] = 0.0f;
    X.footF[0] = X.footF[1] = X.footF[2] = 0.0f;
    X.stamp_base = 0;
    X.coll = 0;
    X.footT[0] = X.footT[1] = X.footT[2] = 0.0f;
    (void)P;
}

// after stage_hot: the lane's step with two inertial records (the sole bodies; at most one per lane)
DQ_HD void quad_lane_second_inertial(QLane &X, const QLds &L, const QuadModel &QM, const DwBuffers &B) {
    int s2 = 0;
    DQ_UNROLL for (int s = 0; s < QS_MAX; ++s) if (s < L.hot.misc[0] && ((f2i(L.hot.in[s][X.j][0]) >> 12) & 3) > 1) s2 = s;
    const QInRec &rc = QM.in[s2][X.j];
    DQ_UNROLL for (int i = 0; i < 3; ++i) X.in1[i] = rc.in1_com[i];
    X.in1[3] = rc.in1_mass;
    DQ_UNROLL for (int i = 0; i < 6; ++i) X.in1[4 + i] = rc.in1_I[i];
    const int g1 = rc.in1_gym;
    X.ms1 = B.mass_scale[(size_t)DW_NUM_BODIES * X.env + (g1 >= 0 && g1 < DW_NUM_BODIES ? g1 : 0)];
}

// ---- joint-parallel phases.  Per-joint work that touches the Gym tensors runs over ITEMS (env, dof) = lane + 64 k of the
// wave's 16 x 33 joints, so that a wave-instruction reads or writes consecutive addresses (the limb-per-lane mapping would
// touch 64 different rows with 4-byte accesses); the item's lane reaches the owner's slot through the owner table. ----
constexpr int QNI = (EPW * ND + 63) / 64;      // 9 items per lane
struct JointItem { int ok, el, d, b, pos, env; };
DQ_HD JointItem joint_item(const QLds &L, int wave_index, int num_envs, int lane, int k) {
    JointItem it;
    const int i = lane + 64 * k;
    it.el = i / ND; it.d = i - ND * it.el; it.b = it.d + 1;
    const int eg = wave_index * EPW + it.el;
    it.ok = (i < EPW * ND) && (eg < num_envs);
    if (!(i < EPW * ND)) { it.el = 0; it.d = 0; it.b = 1; }
    it.env = eg < num_envs ? eg : num_envs - 1;
    it.pos = (it.el + 4 * (L.hot.owner[it.b] >> 6)) & 15;
    return it;
}
// semi-implicit Euler of one joint from the slot the final pass left: q = q_old + dt qd, joint range (outward rate zeroed)
DQ_HD void joint_integrate(QLds &L, const JointItem &it, float dt, float q_old, float *q_out, float *qd_out) {
    const F4 o = DQ_LD(it.b, 0, it.pos);        // {qlo, qd, qhi, *}
    float qd = o.y, q = q_old + dt * qd;
    if (q < o.x) { q = o.x; if (qd < 0) qd = 0; }
    if (q > o.z) { q = o.z; if (qd > 0) qd = 0; }
    *q_out = q; *qd_out = qd;
}

// Gym-boundary substep for 16 envs: tau [N,33], push [N,2] or nullptr (replaces dw::simulate_env)
template <bool TERRAIN>
DQ_HD void quad_simulate(QLds &L, const QuadModel &QM, const DevModel &M, const PhysParams &P, float friction, int num_envs,
                         const DwBuffers &B, const float *tau, const float *push, int wave_index) {
    QLane X;
    quad_lane_init(X, wave_index, num_envs, P, friction, B);
    stage_hot(L, QM);
    quad_lane_second_inertial(X, L, QM, B);
    const int e = X.env, f = X.j & 1;
    if (B.env_state) { DQ_UNROLL for (int i = 0; i < 12; ++i) X.warm[i] = B.env_state[(size_t)DW_ES_WORDS * e + DW_ES_WARM + 12 * f + i]; }
    float qkeep[QNI];
    DQ_UNROLL for (int k = 0; k < QNI; ++k) {
        const JointItem it = joint_item(L, wave_index, num_envs, X.lane, k);
        const size_t g = (size_t)ND * it.env + it.d;
        const float q = B.dof_state[g * 2], qd = B.dof_state[g * 2 + 1];
        const float damp = B.dof_damping[g], arm = B.dof_armature[g];
        qkeep[k] = q;
        if (it.ok || lane_id() + 64 * k < EPW * ND) DQ_SLOT(it.b, 0, it.pos) = mk4(q, qd, tau[g] - damp * qd, arm + P.dt * damp);
    }
    wave_sync();
    quad_substep<TERRAIN>(L, QM, M, P, X, B, push ? push[2 * e] : 0.0f, push ? push[2 * e + 1] : 0.0f, true);
    wave_sync();
    DQ_UNROLL for (int k = 0; k < QNI; ++k) {
        const JointItem it = joint_item(L, wave_index, num_envs, X.lane, k);
        float q, qd;
        joint_integrate(L, it, P.dt, qkeep[k], &q, &qd);
        if (it.ok) {
            const size_t g = (size_t)ND * it.env + it.d;
            B.dof_state[g * 2] = q; B.dof_state[g * 2 + 1] = qd;
        }
    }
    if (X.valid) {
        if (X.j == 0) { DQ_UNROLL for (int i = 0; i < 13; ++i) B.root_states[(size_t)13 * e + i] = X.root[i]; }
        if (X.j < 2 && B.env_state) { DQ_UNROLL for (int i = 0; i < 12; ++i) B.env_state[(size_t)DW_ES_WORDS * e + DW_ES_WARM + 12 * f + i] = X.warm[i]; }
    }
}

}  // namespace dwq
