// dw_physics.h -- one physics substep (the stand-in for the reference's closed `gym.simulate`,
// reference call site tasks/dyros_dynamic_walk.py:525) as wave regions over the env's LDS block.
//
// Algorithm = DESIGN.md "Physics model": floating-base Featherstone ABA (armature + implicit joint
// damping on the diagonal), penalty ground forces for non-sole primitives, velocity-level projected
// Gauss-Seidel on the 8 sole corners over a Delassus matrix assembled from 12 unit-wrench responses of
// the two foot bodies, impulse propagation through the tree, semi-implicit Euler.
//
// Lane maps: lane = body / dof / primitive for the flat phases; (body-in-level, matrix row) = (lane/6,
// lane%6) for the inward articulated-inertia sweep; lane = Delassus column for the 12 response sweeps;
// lane = constraint row for the 24-row Gauss-Seidel.
#pragma once

#include <math.h>

#include "dw_devmodel.h"
#include "dw_wave.h"

namespace dw {

struct PhysParams {          // wave-uniform scalars (kernel arguments)
    float dt;
    float g[3];
    int   iters;
    float contact_offset, max_depen, erp, cfm;
    float pen_k, pen_c;
    float max_ang_vel;
    int   vel_at_com;
    int   self_collision;
};

// The kinematic tree, staged once per launch into the env's LDS block: every sweep region indexes these by lane
// or by level, and a dependent chain of global loads (level -> body -> parent -> offset/axis) per region is what
// the first profile of this kernel was made of.
struct LdsTree {
    float pos[NB][3];
    float axis[NB][3];
    unsigned char parent[NB];
    unsigned char nchild[NB];
    unsigned char child[NB][MAX_CHILD];
    unsigned char level_count[MAX_LEVELS];
    unsigned char level_body[MAX_LEVELS][MAX_PER_LEVEL];
    unsigned char nlevels;
};

// Packed index of entry (r,c) of a symmetric 6x6 (upper triangle, row-major): 21 words instead of 36.
DW_HD constexpr int sym6(int r, int c) {
    return r <= c ? (r * (13 - r)) / 2 + (c - r) : (c * (13 - c)) / 2 + (r - c);
}

// One env's LDS block.  160 KB per CU / 13.5 KB = 12 resident envs (3 waves per SIMD); the first version of this
// struct was 18.6 KB (8 envs).  The saving comes from overlaying arrays whose lifetimes inside a substep do not
// intersect (phases in order: K1 K2 kinematics, K4 K5 primitives and self-collision, K3 inertias, SW inward sweep, A3 base solve,
// A4 outward sweep, V1 free velocities, C1..C5 contact, V2 integrate) and from packing the symmetric
// articulated inertias.
struct Lds {
    LdsTree tree;
    // ---- state and inputs of the substep (live throughout) ----
    float root[13];
    float q[ND], qd[ND], tau[ND], arm[ND], damp[ND];
    float mscale[DW_NUM_BODIES];
    float mu;
    float push[2];
    float warm[24];
    float contact[DW_NUM_BODIES * 3];
    float quat[4], ww[3], vow[3];
    float R[NB][9];                 // body -> parent, K1 .. C5
    float RwK[3][9], pwK[3][3];     // world pose of the base and the two sole bodies, K5 .. V2
    union {                         // block B
        struct { float Rw[NB][9]; float pw[NB][3]; } kin;                       // K2 .. K5
        struct { float T[MAX_PER_LEVEL][36]; float pa[MAX_PER_LEVEL][6]; } sw;  // SW
        struct {                                                                // A3 .. V2
            float a[NB][6];
            float du[NB];
            float qdd[ND], qdf[ND], dqd[ND];
            float wwf[3], vowf[3], dv0[6];
            float Minv[36];
            float part[MAX_PER_LEVEL][3];   // per-column partial sums of U'a in the outward sweeps
        } post;
    } B;
    union {                         // block V
        struct { float v[NB][6]; float pA[NB][6]; } dyn;                        // K2 .. A4 (v), K3 .. A3 (pA)
        struct {                                                                // V1 .. C5 (W first: v[0] is read in V1)
            float W[12][12];
            float vel[2][24], P[2][24];
            float rk[8][3], phi[8], vmin[8];
            int   active[8];
            int   any_active;
            float twf[2][6];
            float ducol[12][6];
        } con;
    } V;
    struct {                        // block C
        struct { float U[NB][6], Dinv[NB], u[NB]; } art;                        // SW .. C5
    } C;
    union {                         // block A
        struct {                                                                // K4 .. K5 (before the inertias are built)
            float gF[64][3], gr[64][3];                                         //   ground penalty: world force, body-frame point
            float pF[DW_MAX_SC_PAIRS][3], pa[DW_MAX_SC_PAIRS][3], pb[DW_MAX_SC_PAIRS][3];   // self-collision: force on A, points on A / B
        } geo;
        float IA[NB][21];                                                       // K3 .. A3 (packed, sym6)
        struct { float A[24][24]; float invd[24]; float dpf[2][6]; } lcp;       // C3 .. C5
    } A;
    // ---- task state (dw_task.h), live for the whole policy step ----
    float es[DW_ES_WORDS];
    float act[DW_NUM_ACT];
    float normed[DW_NUM_OBS1];
    float rterm[16];
    float scratch[8];
    int   flags[8];
};

// ------------------------------------------------------------------------------------------------
// small math on plain float arrays (registers or LDS)
// ------------------------------------------------------------------------------------------------
DW_HD void cross3(const float *a, const float *b, float *o) {
    float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
DW_HD float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DW_HD void m3v(const float *M, const float *v, float *o) {
    float x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    float y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    float z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
DW_HD void m3tv(const float *M, const float *v, float *o) {
    float x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    float y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    float z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
DW_HD void m3m(const float *X, const float *Y, float *O) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) O[3 * r + c] = X[3 * r] * Y[c] + X[3 * r + 1] * Y[3 + c] + X[3 * r + 2] * Y[6 + c];
}
DW_HD void quat_to_mat(const float *q, float *R) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
// motion transform parent -> body:  w_b = R' w_p ; v_b = R' (v_p + w_p x p)
DW_HD void xform_motion(const float *R, const float *p, const float *vp, float *vb) {
    float t[3], l[3];
    m3tv(R, vp, vb);
    cross3(vp, p, t);
    l[0] = vp[3] + t[0]; l[1] = vp[4] + t[1]; l[2] = vp[5] + t[2];
    m3tv(R, l, vb + 3);
}
// force transform body -> parent: f_p = R f_b ; n_p = R n_b + p x f_p
DW_HD void xform_force(const float *R, const float *p, const float *fb, float *fp) {
    float n[3], f[3], t[3];
    m3v(R, fb, n);
    m3v(R, fb + 3, f);
    cross3(p, f, t);
    fp[0] = n[0] + t[0]; fp[1] = n[1] + t[1]; fp[2] = n[2] + t[2];
    fp[3] = f[0]; fp[4] = f[1]; fp[5] = f[2];
}
// PR = skew(p) * R   (the lower-left block of the force transform; (-E P)_kc = PR[3c+k])
DW_HD void make_PR(const float *R, const float *p, float *PR) {
    for (int c = 0; c < 3; ++c) {
        float col[3] = {R[c], R[3 + c], R[6 + c]}, o[3];
        cross3(p, col, o);
        PR[c] = o[0]; PR[3 + c] = o[1]; PR[6 + c] = o[2];
    }
}
// 1/sqrt(x) for x > 0: hardware estimate refined by one Newton step on the device (~1 ulp), exact on the host
DW_HD float rsqrt_nr(float x) {
#if defined(__HIPCC__)
    float y = __builtin_amdgcn_rsqf(x);
    return y * (1.5f - 0.5f * x * y * y);
#else
    return 1.0f / sqrtf(x);
#endif
}
// bias acceleration of a hinge: c = v x (S qd)
DW_HD void joint_bias(const float *v, const float *s, float qd, float *c) {
    float sq[3] = {s[0] * qd, s[1] * qd, s[2] * qd};
    cross3(v, sq, c);
    cross3(v + 3, sq, c + 3);
}

// ------------------------------------------------------------------------------------------------
// the substep.  In: S.root, q, qd, tau, arm, damp, mscale, mu, push, warm.  Out: root, q, qd, warm, contact.
// ------------------------------------------------------------------------------------------------
// copies the tree tables from the device-resident model into LDS (one region)
template <class W>
DW_HD void stage_tree(const W &wave, Lds &S, const DevModel &M) {
    wave.par([&](int l) {
        if (l < NB) {
            S.tree.parent[l] = M.parent[l];
            S.tree.nchild[l] = M.nchild[l];
            for (int i = 0; i < MAX_CHILD; ++i) S.tree.child[l][i] = M.child[l][i];
            for (int i = 0; i < 3; ++i) { S.tree.pos[l][i] = M.pos[l][i]; S.tree.axis[l][i] = M.axis[l][i]; }
        }
        if (l < MAX_LEVELS) S.tree.level_count[l] = M.level_count[l];
        if (l < MAX_LEVELS * MAX_PER_LEVEL) S.tree.level_body[l / MAX_PER_LEVEL][l % MAX_PER_LEVEL] = M.level_body[l / MAX_PER_LEVEL][l % MAX_PER_LEVEL];
        if (l == 63) S.tree.nlevels = M.nlevels;
    });
}

// Profiling builds (-DDW_PROFILE_STOP=n) leave the substep after phase n; results are then meaningless, only the
// launch time is read (tools/phase_costs.sh).  Never defined in the shipped library.
#if defined(DW_PROFILE_STOP)
#define DW_CKPT(n) do { if (DW_PROFILE_STOP == (n)) return; } while (0)
#else
#define DW_CKPT(n) do { } while (0)
#endif

template <class W>
DW_HD void physics_substep(const W &wave, Lds &S, const DevModel &M, const PhysParams &P) {
    const float dt = P.dt;

    // ---- K1: base state, joint rotations, clear contact accumulators ----
    wave.par([&](int l) {
        for (int i = l; i < DW_NUM_BODIES * 3; i += 64) S.contact[i] = 0.0f;
        if (l == 0) {
            float qx = S.root[3], qy = S.root[4], qz = S.root[5], qw = S.root[6];
            float n = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
            float qn[4] = {qx / n, qy / n, qz / n, qw / n};
            for (int i = 0; i < 4; ++i) S.quat[i] = qn[i];
            float Rw[9];
            quat_to_mat(qn, Rw);
            for (int i = 0; i < 9; ++i) { S.B.kin.Rw[0][i] = Rw[i]; S.R[0][i] = Rw[i]; }
            for (int i = 0; i < 3; ++i) S.B.kin.pw[0][i] = S.root[i];
            float ww[3] = {S.root[10], S.root[11], S.root[12]};
            float vo[3] = {S.root[7], S.root[8], S.root[9]};
            if (P.vel_at_com) {
                float rc[3], t[3];
                m3v(Rw, M.inert_com[0], rc);
                cross3(ww, rc, t);
                vo[0] -= t[0]; vo[1] -= t[1]; vo[2] -= t[2];
            }
            for (int i = 0; i < 3; ++i) { S.ww[i] = ww[i]; S.vow[i] = vo[i]; }
            float vb[6];
            m3tv(Rw, ww, vb);
            m3tv(Rw, vo, vb + 3);
            for (int i = 0; i < 6; ++i) S.V.dyn.v[0][i] = vb[i];
        } else if (l < NB) {
            const int b = l;
            const float *s = S.tree.axis[b];
            float q = S.q[b - 1];
            float sn, cs;
            sincosf(q, &sn, &cs);
            const float oc = 1.0f - cs;
            float Rj[9] = {cs + oc * s[0] * s[0], oc * s[0] * s[1] - sn * s[2], oc * s[0] * s[2] + sn * s[1],
                           oc * s[1] * s[0] + sn * s[2], cs + oc * s[1] * s[1], oc * s[1] * s[2] - sn * s[0],
                           oc * s[2] * s[0] - sn * s[1], oc * s[2] * s[1] + sn * s[0], cs + oc * s[2] * s[2]};
            float R[9];
            m3m(M.rot0[b], Rj, R);
            for (int i = 0; i < 9; ++i) S.R[b][i] = R[i];
        }
    });

    DW_CKPT(1);
    // ---- K2: forward kinematics and velocities, level by level; three lanes per body (lane = body-in-level, column) ----
    for (int L = 1; L <= S.tree.nlevels; ++L) {
        wave.par([&](int l) {
            const int k = l / 3, c = l - 3 * k;
            if (k < S.tree.level_count[L]) {
                const int b = S.tree.level_body[L][k], p = S.tree.parent[b];
                const float *Rp = S.B.kin.Rw[p], *Rb = S.R[b], *pos = S.tree.pos[b], *vp = S.V.dyn.v[p];
                const float r0 = Rb[c], r1 = Rb[3 + c], r2 = Rb[6 + c];           // column c of R
                for (int i = 0; i < 3; ++i) S.B.kin.Rw[b][3 * i + c] = Rp[3 * i] * r0 + Rp[3 * i + 1] * r1 + Rp[3 * i + 2] * r2;
                S.B.kin.pw[b][c] = S.B.kin.pw[p][c] + Rp[3 * c] * pos[0] + Rp[3 * c + 1] * pos[1] + Rp[3 * c + 2] * pos[2];
                // w_b = R' w_p + s qd ; v_b = R' (v_p + w_p x p): component c uses column c of R
                const float t0 = vp[1] * pos[2] - vp[2] * pos[1], t1 = vp[2] * pos[0] - vp[0] * pos[2], t2 = vp[0] * pos[1] - vp[1] * pos[0];
                S.V.dyn.v[b][c] = r0 * vp[0] + r1 * vp[1] + r2 * vp[2] + S.tree.axis[b][c] * S.qd[b - 1];
                S.V.dyn.v[b][3 + c] = r0 * (vp[3] + t0) + r1 * (vp[4] + t1) + r2 * (vp[5] + t2);
            }
        });
    }

    DW_CKPT(2);
    // ---- K4: penalty contact of the non-sole primitives against the ground (one lane per primitive) ----
    wave.par([&](int l) {
        float F[3] = {0, 0, 0}, rl[3] = {0, 0, 0};
        if (l < M.ngeom && !M.geoms[l].sole) {
            const DwGeom &ge = M.geoms[l];
            const int b = ge.moving;
            float Rw[9];
            for (int i = 0; i < 9; ++i) Rw[i] = S.B.kin.Rw[b][i];
            float zmin;
            if (ge.type == 0) {
                // deepest corner: along each box axis take the end that points down (world z component of the axis)
                float Rg[9], e[3];
                m3m(Rw, ge.rot, Rg);
                for (int i = 0; i < 3; ++i) e[i] = (Rg[6 + i] > 0.0f ? -1.0f : 1.0f) * ge.size[i];
                float lc[3], wv[3];
                m3v(ge.rot, e, lc);
                lc[0] += ge.pos[0]; lc[1] += ge.pos[1]; lc[2] += ge.pos[2];
                m3v(Rw, lc, wv);
                zmin = S.B.kin.pw[b][2] + wv[2];
                rl[0] = lc[0]; rl[1] = lc[1]; rl[2] = lc[2];
            } else {
                float al[3] = {ge.rot[2], ge.rot[5], ge.rot[8]}, aw[3];
                m3v(Rw, al, aw);
                float sgn = aw[2] >= 0 ? -1.0f : 1.0f;
                float dw3[3] = {-aw[2] * aw[0], -aw[2] * aw[1], 1.0f - aw[2] * aw[2]};
                float dn = sqrtf(dot3(dw3, dw3));
                float off[3] = {0, 0, 0};
                if (dn > 1e-6f) {
                    float k = -ge.size[0] / dn;
                    float ow[3] = {k * dw3[0], k * dw3[1], k * dw3[2]};
                    m3tv(Rw, ow, off);
                }
                for (int i = 0; i < 3; ++i) rl[i] = ge.pos[i] + sgn * ge.size[1] * al[i] + off[i];
                float wv[3];
                m3v(Rw, rl, wv);
                zmin = S.B.kin.pw[b][2] + wv[2];
            }
            if (zmin < 0) {
                float vb[6], t[3], vl[3], vw[3];
                for (int i = 0; i < 6; ++i) vb[i] = S.V.dyn.v[b][i];
                cross3(vb, rl, t);
                vl[0] = vb[3] + t[0]; vl[1] = vb[4] + t[1]; vl[2] = vb[5] + t[2];
                m3v(Rw, vl, vw);
                float fn = P.pen_k * (-zmin) - P.pen_c * vw[2];
                if (fn < 0) fn = 0;
                float sp = sqrtf(vw[0] * vw[0] + vw[1] * vw[1]);
                F[2] = fn;
                if (sp > 1e-9f) {
                    float ft = P.pen_c * sp, lim = S.mu * fn;
                    if (ft > lim) ft = lim;
                    F[0] = -ft * vw[0] / sp; F[1] = -ft * vw[1] / sp;
                }
            }
        }
        for (int i = 0; i < 3; ++i) { S.A.geo.gF[l][i] = F[i]; S.A.geo.gr[l][i] = rl[i]; }
    });
    // ---- K4b: self-collision, one lane per capsule pair: closest points of the two segments, penalty force along the
    //      normal when the capsules overlap (force on A; B gets the opposite) ----
    wave.par([&](int l) {
        if (l < DW_MAX_SC_PAIRS) for (int i = 0; i < 3; ++i) S.A.geo.pF[l][i] = 0.0f;
        if (P.self_collision && l < M.num_sc_pairs) {
            const DwCapsule &ca = M.sc_proxy[M.sc_pair[l][0]], &cb = M.sc_proxy[M.sc_pair[l][1]];
            const int ba = ca.moving, bb = cb.moving;
            float Ra[9], Rb[9], a0[3], a1[3], b0[3], b1[3], t3[3];
            for (int i = 0; i < 9; ++i) { Ra[i] = S.B.kin.Rw[ba][i]; Rb[i] = S.B.kin.Rw[bb][i]; }
            m3v(Ra, ca.p0, t3); for (int i = 0; i < 3; ++i) a0[i] = S.B.kin.pw[ba][i] + t3[i];
            m3v(Ra, ca.p1, t3); for (int i = 0; i < 3; ++i) a1[i] = S.B.kin.pw[ba][i] + t3[i];
            m3v(Rb, cb.p0, t3); for (int i = 0; i < 3; ++i) b0[i] = S.B.kin.pw[bb][i] + t3[i];
            m3v(Rb, cb.p1, t3); for (int i = 0; i < 3; ++i) b1[i] = S.B.kin.pw[bb][i] + t3[i];
            const float da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
            // closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9)
            const float r[3] = {a0[0] - b0[0], a0[1] - b0[1], a0[2] - b0[2]};
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
                    sa = den > eps ? c01((bbv * ff - cc * ee) / den) : 0.0f;
                    sb = (bbv * sa + ff) / ee;
                    if (sb < 0.0f) { sb = 0.0f; sa = c01(-cc / aa); }
                    else if (sb > 1.0f) { sb = 1.0f; sa = c01((bbv - cc) / aa); }
                }
            }
            float pa[3], pb[3], n[3];
            for (int i = 0; i < 3; ++i) { pa[i] = a0[i] + sa * da[i]; pb[i] = b0[i] + sb * db[i]; n[i] = pa[i] - pb[i]; }
            const float dist = sqrtf(dot3(n, n));
            const float depth = ca.radius + cb.radius - dist;
            float F[3] = {0, 0, 0}, ra[3] = {0, 0, 0}, rb[3] = {0, 0, 0};
            if (depth > 0.0f && dist > 1e-6f) {
                for (int i = 0; i < 3; ++i) n[i] /= dist;
                float va[3], vb[3], tt[3], vl[3];
                for (int i = 0; i < 3; ++i) t3[i] = pa[i] - S.B.kin.pw[ba][i];
                m3tv(Ra, t3, ra);
                cross3(S.V.dyn.v[ba], ra, tt);
                for (int i = 0; i < 3; ++i) vl[i] = S.V.dyn.v[ba][3 + i] + tt[i];
                m3v(Ra, vl, va);
                for (int i = 0; i < 3; ++i) t3[i] = pb[i] - S.B.kin.pw[bb][i];
                m3tv(Rb, t3, rb);
                cross3(S.V.dyn.v[bb], rb, tt);
                for (int i = 0; i < 3; ++i) vl[i] = S.V.dyn.v[bb][3 + i] + tt[i];
                m3v(Rb, vl, vb);
                const float vn = (va[0] - vb[0]) * n[0] + (va[1] - vb[1]) * n[1] + (va[2] - vb[2]) * n[2];
                float fn = P.pen_k * depth - P.pen_c * vn;
                if (fn < 0.0f) fn = 0.0f;
                F[0] = fn * n[0]; F[1] = fn * n[1]; F[2] = fn * n[2];
            }
            for (int i = 0; i < 3; ++i) { S.A.geo.pF[l][i] = F[i]; S.A.geo.pa[l][i] = ra[i]; S.A.geo.pb[l][i] = rb[i]; }
        }
    });
    // K5: external forces into the bias of their bodies; per-body net contact force
    wave.par([&](int l) {
        if (l < NB) {
            const int b = l;
            float Rw[9];
            for (int i = 0; i < 9; ++i) Rw[i] = S.B.kin.Rw[b][i];
            float dn[3] = {0, 0, 0}, df[3] = {0, 0, 0};
            for (int k = 0; k < M.body_ngeom[b]; ++k) {
                const int g = M.body_geom[b][k];
                float F[3] = {S.A.geo.gF[g][0], S.A.geo.gF[g][1], S.A.geo.gF[g][2]};
                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                    float rl[3] = {S.A.geo.gr[g][0], S.A.geo.gr[g][1], S.A.geo.gr[g][2]}, fb[3], nb[3];
                    m3tv(Rw, F, fb);
                    cross3(rl, fb, nb);
                    for (int i = 0; i < 3; ++i) { dn[i] += nb[i]; df[i] += fb[i]; }
                    const int gy = M.geoms[g].gym;
                    for (int i = 0; i < 3; ++i) S.contact[3 * gy + i] += F[i];
                }
            }
            for (int k = 0; k < M.body_npair[b]; ++k) {         // self-collision pairs this body takes part in
                const int code = M.body_pair[b][k], pr = code >> 1, side = code & 1;
                const float sg = side ? -1.0f : 1.0f;
                float F[3] = {sg * S.A.geo.pF[pr][0], sg * S.A.geo.pF[pr][1], sg * S.A.geo.pF[pr][2]};
                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                    const float *rp = side ? S.A.geo.pb[pr] : S.A.geo.pa[pr];
                    float rl[3] = {rp[0], rp[1], rp[2]}, fb[3], nb[3];
                    m3tv(Rw, F, fb);
                    cross3(rl, fb, nb);
                    for (int i = 0; i < 3; ++i) { dn[i] += nb[i]; df[i] += fb[i]; }
                    const int gy = M.sc_proxy[M.sc_pair[pr][side]].gym;
                    for (int i = 0; i < 3; ++i) S.contact[3 * gy + i] += F[i];
                }
            }
            if (b == 0) {
                float Fw[3] = {S.push[0], S.push[1], 0.0f}, fb[3], nb[3];
                m3tv(Rw, Fw, fb);
                cross3(M.inert_com[0], fb, nb);
                for (int i = 0; i < 3; ++i) { dn[i] += nb[i]; df[i] += fb[i]; }
            }
            for (int i = 0; i < 3; ++i) { S.V.dyn.pA[b][i] = -dn[i]; S.V.dyn.pA[b][3 + i] = -df[i]; }   // K3 adds the gyroscopic part
            if (b == 0 || b == 6 || b == 12) {      // block B is recycled by the sweep: keep what the contact phases need
                const int slot = b / 6;
                for (int i = 0; i < 9; ++i) S.RwK[slot][i] = Rw[i];
                for (int i = 0; i < 3; ++i) S.pwK[slot][i] = S.B.kin.pw[b][i];
            }
        }
    });

    // ---- K3: rigid-body inertias (packed into block A, whose primitive scratch is dead now), gyroscopic bias ----
    wave.par([&](int l) {
        if (l < NB) {
            const int b = l;
            float A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, h[3] = {0, 0, 0}, mass = 0.0f;
            for (int k = 0; k < M.ninert[b]; ++k) {
                const int r = M.inert_idx[b][k];
                const float ms = S.mscale[M.inert_gym[r]];
                const float mk = ms * M.inert_mass[r];
                const float *cm = M.inert_com[r], *I6 = M.inert_I[r];
                const float cc = dot3(cm, cm);
                const float Ic[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
                for (int r3 = 0; r3 < 3; ++r3)
                    for (int c3 = 0; c3 < 3; ++c3)
                        A[3 * r3 + c3] += ms * Ic[3 * r3 + c3] + mk * ((r3 == c3 ? cc : 0.0f) - cm[r3] * cm[c3]);
                h[0] += mk * cm[0]; h[1] += mk * cm[1]; h[2] += mk * cm[2];
                mass += mk;
            }
            // 6x6 = [[A, H],[H', m 1]] with H = skew(h), stored packed (upper triangle)
            float *I = S.A.IA[b];
            const float H[9] = {0, -h[2], h[1], h[2], 0, -h[0], -h[1], h[0], 0};
            for (int r3 = 0; r3 < 3; ++r3)
                for (int c3 = 0; c3 < 3; ++c3) {
                    if (c3 >= r3) {
                        I[sym6(r3, c3)] = A[3 * r3 + c3];
                        I[sym6(r3 + 3, c3 + 3)] = (r3 == c3) ? mass : 0.0f;
                    }
                    I[sym6(r3, 3 + c3)] = H[3 * r3 + c3];
                }
            // pA = v x* (I v)
            float om[3] = {S.V.dyn.v[b][0], S.V.dyn.v[b][1], S.V.dyn.v[b][2]}, vl[3] = {S.V.dyn.v[b][3], S.V.dyn.v[b][4], S.V.dyn.v[b][5]};
            float n[3], f[3], t1[3], t2[3];
            m3v(A, om, n); cross3(h, vl, t1);
            n[0] += t1[0]; n[1] += t1[1]; n[2] += t1[2];
            cross3(om, h, t1);                       // H' w = -h x w = w x h
            f[0] = t1[0] + mass * vl[0]; f[1] = t1[1] + mass * vl[1]; f[2] = t1[2] + mass * vl[2];
            cross3(om, n, t1); cross3(vl, f, t2);
            S.V.dyn.pA[b][0] += t1[0] + t2[0]; S.V.dyn.pA[b][1] += t1[1] + t2[1]; S.V.dyn.pA[b][2] += t1[2] + t2[2];
            cross3(om, f, t1);
            S.V.dyn.pA[b][3] += t1[0]; S.V.dyn.pA[b][4] += t1[1]; S.V.dyn.pA[b][5] += t1[2];
        }
    });
    DW_CKPT(3);
    // ---- A2: inward sweep of articulated inertias.  Twelve lanes per body: lane = (body-in-level k, row r, half h),
    //      each lane owns the three columns 3h..3h+2 of row r (5 bodies x 12 = 60 lanes on the widest level). ----
    for (int L = S.tree.nlevels; L >= 1; --L) {
        const int cnt = S.tree.level_count[L];
        // A: projection through the joint, Ia = IA - U U'/D, rows of Ia*X
        wave.par([&](int l) {
            const int k = l / 12, r = (l % 12) >> 1, h = l & 1;
            if (k < cnt) {
                const int b = S.tree.level_body[L][k];
                const float *s = S.tree.axis[b];
                const float *IA = S.A.IA[b];
                float U[6];
                for (int j = 0; j < 6; ++j) U[j] = IA[sym6(j, 0)] * s[0] + IA[sym6(j, 1)] * s[1] + IA[sym6(j, 2)] * s[2];
                const float damp = S.damp[b - 1], qd = S.qd[b - 1];
                const float D = dot3(s, U) + S.arm[b - 1] + dt * damp;
                const float Dinv = 1.0f / D;
                const float u = S.tau[b - 1] - damp * qd - (s[0] * S.V.dyn.pA[b][0] + s[1] * S.V.dyn.pA[b][1] + s[2] * S.V.dyn.pA[b][2]);
                float Ia[6];
                const float ur = U[r] * Dinv;
                for (int c = 0; c < 6; ++c) Ia[c] = IA[sym6(r, c)] - ur * U[c];
                float R[9], PR[9];
                for (int i = 0; i < 9; ++i) R[i] = S.R[b][i];
                make_PR(R, S.tree.pos[b], PR);
                // half 0: T[r][c] = Ia[0:3].R_c + Ia[3:6].PR_c ; half 1: T[r][3+c] = Ia[3:6].R_c
                const float a0 = h ? Ia[3] : Ia[0], a1 = h ? Ia[4] : Ia[1], a2 = h ? Ia[5] : Ia[2];
                const float b0 = h ? 0.0f : Ia[3], b1 = h ? 0.0f : Ia[4], b2 = h ? 0.0f : Ia[5];
                for (int c = 0; c < 3; ++c)
                    S.B.sw.T[k][6 * r + 3 * h + c] = a0 * R[3 * c] + a1 * R[3 * c + 1] + a2 * R[3 * c + 2] +
                                                     b0 * PR[3 * c] + b1 * PR[3 * c + 1] + b2 * PR[3 * c + 2];
                if (h == 0) {
                    float vb[6], cb[6];
                    for (int i = 0; i < 6; ++i) vb[i] = S.V.dyn.v[b][i];
                    joint_bias(vb, s, qd, cb);
                    float pa = S.V.dyn.pA[b][r] + U[r] * (u * Dinv);
                    for (int c = 0; c < 6; ++c) pa += Ia[c] * cb[c];
                    S.B.sw.pa[k][r] = pa;
                    if (r == 0) {
                        for (int j = 0; j < 6; ++j) S.C.art.U[b][j] = U[j];
                        S.C.art.Dinv[b] = Dinv;
                        S.C.art.u[b] = u;
                    }
                }
            }
        });
        // B: X' (Ia X) and X' pa, written over the child's own storage (now "contribution to the parent").
        //    Row q of R and of skew(p)*R come straight from LDS (a register array indexed by the lane's row would
        //    live in scratch memory).
        wave.par([&](int l) {
            const int k = l / 12, r = (l % 12) >> 1, h = l & 1;
            if (k < cnt) {
                const int b = S.tree.level_body[L][k];
                const int q = r < 3 ? r : r - 3;
                const int q1 = q == 2 ? 0 : q + 1, q2 = q == 0 ? 2 : q - 1;
                const float *Rb = S.R[b];
                const float *pp = S.tree.pos[b];
                const float Rq[3] = {Rb[3 * q], Rb[3 * q + 1], Rb[3 * q + 2]};
                const float p1 = r < 3 ? pp[q1] : 0.0f, p2 = r < 3 ? pp[q2] : 0.0f;
                float PRq[3];
                for (int c = 0; c < 3; ++c) PRq[c] = p1 * Rb[3 * q2 + c] - p2 * Rb[3 * q1 + c];
                const float *T = S.B.sw.T[k] + (r < 3 ? 0 : 18) + 3 * h;
                const float *T2 = S.B.sw.T[k] + 18 + 3 * h;
                for (int c = 0; c < 3; ++c) {
                    const float o = Rq[0] * T[c] + Rq[1] * T[6 + c] + Rq[2] * T[12 + c] +
                                    PRq[0] * T2[c] + PRq[1] * T2[6 + c] + PRq[2] * T2[12 + c];
                    if (3 * h + c >= r) S.A.IA[b][sym6(r, 3 * h + c)] = o;
                }
                if (h == 0) {
                    const float *pa = S.B.sw.pa[k] + (r < 3 ? 0 : 3);
                    const float *pa2 = S.B.sw.pa[k] + 3;
                    S.V.dyn.pA[b][r] = Rq[0] * pa[0] + Rq[1] * pa[1] + Rq[2] * pa[2] +
                                       PRq[0] * pa2[0] + PRq[1] * pa2[1] + PRq[2] * pa2[2];
                }
            }
        });
        // C: parents (one level up) gather their children, fixed order (three slots, absent children add zero)
        const int pcnt = (L == 1) ? 1 : S.tree.level_count[L - 1];
        wave.par([&](int l) {
            const int k = l / 12, r = (l % 12) >> 1, h = l & 1;
            if (k < pcnt) {
                const int p = (L == 1) ? 0 : S.tree.level_body[L - 1][k];
                const int nch = S.tree.nchild[p];
                const int c0 = S.tree.child[p][0], c1 = nch > 1 ? S.tree.child[p][1] : c0, c2 = nch > 2 ? S.tree.child[p][2] : c0;
                const float w1 = nch > 1 ? 1.0f : 0.0f, w2 = nch > 2 ? 1.0f : 0.0f;
                if (nch > 0) {
                    for (int c = 3 * h; c < 3 * h + 3; ++c) {
                        const int ix = sym6(r, c);
                        if (c >= r)
                            S.A.IA[p][ix] = S.A.IA[p][ix] + S.A.IA[c0][ix] + w1 * S.A.IA[c1][ix] + w2 * S.A.IA[c2][ix];
                    }
                    if (h == 0)
                        S.V.dyn.pA[p][r] = S.V.dyn.pA[p][r] + S.V.dyn.pA[c0][r] + w1 * S.V.dyn.pA[c1][r] + w2 * S.V.dyn.pA[c2][r];
                }
            }
        });
    }

    DW_CKPT(4);
    // ---- A3: inverse of the base articulated inertia (6 lanes, one column each, Cholesky) ----
    wave.par([&](int l) {
        if (l < 6) {
            float Lc[36];
            const float *Mx = S.A.IA[0];
            for (int i = 0; i < 36; ++i) Lc[i] = 0.0f;
            float dinv[6];                         // 1 / L_jj
            for (int j = 0; j < 6; ++j) {
                float d = Mx[sym6(j, j)];
                for (int k = 0; k < j; ++k) d -= Lc[6 * j + k] * Lc[6 * j + k];
                dinv[j] = rsqrt_nr(d);
                Lc[6 * j + j] = d * dinv[j];
                for (int i = j + 1; i < 6; ++i) {
                    float s = Mx[sym6(i, j)];
                    for (int k = 0; k < j; ++k) s -= Lc[6 * i + k] * Lc[6 * j + k];
                    Lc[6 * i + j] = s * dinv[j];
                }
            }
            float y[6], x[6];
            for (int i = 0; i < 6; ++i) {
                float s = (i == l) ? 1.0f : 0.0f;
                for (int k = 0; k < i; ++k) s -= Lc[6 * i + k] * y[k];
                y[i] = s * dinv[i];
            }
            for (int i = 5; i >= 0; --i) {
                float s = y[i];
                for (int k = i + 1; k < 6; ++k) s -= Lc[6 * k + i] * x[k];
                x[i] = s * dinv[i];
            }
            for (int i = 0; i < 6; ++i) S.B.post.Minv[6 * i + l] = x[i];
        }
    });
    // from here on S.in is dead and S.out is live
    wave.par([&](int l) {
        if (l < 6) {
            float acc = 0.0f;
            for (int c = 0; c < 6; ++c) acc -= S.B.post.Minv[6 * l + c] * S.V.dyn.pA[0][c];
            S.B.post.a[0][l] = acc;
        }
        if (l >= 8 && l < 8 + NB) S.B.post.du[l - 8] = 0.0f;
    });

    DW_CKPT(5);
    // ---- A4: outward sweep of accelerations; three lanes per body (column c of the transform), two regions per level:
    //      a' = X a_parent + c_b and the per-column part of U'a', then qdd = (u - U'a') / D and a = a' + S qdd ----
    for (int L = 1; L <= S.tree.nlevels; ++L) {
        wave.par([&](int l) {
            const int k = l / 3, c = l - 3 * k;
            if (k < S.tree.level_count[L]) {
                const int b = S.tree.level_body[L][k], p = S.tree.parent[b];
                const int c1 = c == 2 ? 0 : c + 1, c2 = c == 0 ? 2 : c - 1;
                const float *Rb = S.R[b], *pos = S.tree.pos[b], *ap = S.B.post.a[p], *vb = S.V.dyn.v[b], *ax = S.tree.axis[b];
                const float r0 = Rb[c], r1 = Rb[3 + c], r2 = Rb[6 + c];
                const float t0 = ap[1] * pos[2] - ap[2] * pos[1], t1 = ap[2] * pos[0] - ap[0] * pos[2], t2 = ap[0] * pos[1] - ap[1] * pos[0];
                const float qd = S.qd[b - 1];
                const float s1 = ax[c1] * qd, s2 = ax[c2] * qd;
                const float ang = r0 * ap[0] + r1 * ap[1] + r2 * ap[2] + (vb[c1] * s2 - vb[c2] * s1);
                const float lin = r0 * (ap[3] + t0) + r1 * (ap[4] + t1) + r2 * (ap[5] + t2) + (vb[3 + c1] * s2 - vb[3 + c2] * s1);
                S.B.post.a[b][c] = ang;
                S.B.post.a[b][3 + c] = lin;
                S.B.post.part[k][c] = S.C.art.U[b][c] * ang + S.C.art.U[b][3 + c] * lin;
            }
        });
        wave.par([&](int l) {
            const int k = l / 3, c = l - 3 * k;
            if (k < S.tree.level_count[L]) {
                const int b = S.tree.level_body[L][k];
                const float ua = S.B.post.part[k][0] + S.B.post.part[k][1] + S.B.post.part[k][2];
                const float qdd = (S.C.art.u[b] - ua) * S.C.art.Dinv[b];
                if (c == 0) S.B.post.qdd[b - 1] = qdd;
                S.B.post.a[b][c] += S.tree.axis[b][c] * qdd;
            }
        });
    }

    DW_CKPT(6);
    // ---- V1: unconstrained velocities; sole-corner gaps ----
    wave.par([&](int l) {
        if (l < ND) S.B.post.qdf[l] = S.qd[l] + dt * S.B.post.qdd[l];
        if (l == 40) {
            float Rw[9], a0[6], vb0[6], t[3], t2[3], al[3];
            for (int i = 0; i < 9; ++i) Rw[i] = S.RwK[0][i];
            for (int i = 0; i < 6; ++i) { a0[i] = S.B.post.a[0][i]; vb0[i] = S.V.dyn.v[0][i]; }
            m3v(Rw, a0, t);
            for (int i = 0; i < 3; ++i) S.B.post.wwf[i] = S.ww[i] + dt * t[i];
            cross3(vb0, vb0 + 3, t2);
            al[0] = a0[3] + t2[0]; al[1] = a0[4] + t2[1]; al[2] = a0[5] + t2[2];
            m3v(Rw, al, t);
            for (int i = 0; i < 3; ++i) S.B.post.vowf[i] = S.vow[i] + dt * (t[i] + P.g[i]);
        }
        if (l >= 48 && l < 48 + DW_NUM_FOOT_PTS) {
            const int k = l - 48;
            float r[3];
            m3v(S.RwK[1 + k / 4], M.foot_pos[k], r);
            const float phi = S.pwK[1 + k / 4][2] + r[2];
            const int act = phi < P.contact_offset;
            for (int i = 0; i < 3; ++i) S.V.con.rk[k][i] = r[i];
            S.V.con.phi[k] = phi;
            S.V.con.active[k] = act;
            S.V.con.vmin[k] = phi >= 0 ? -phi / dt : fminf(P.erp * (-phi) / dt, P.max_depen);
        }
        if (l < 6) S.B.post.dv0[l] = 0.0f;
        if (l < ND) S.B.post.dqd[l] = 0.0f;
    });
    wave.par([&](int l) {
        if (l == 0) {
            int any = 0;
            for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) any |= S.V.con.active[k];
            S.V.con.any_active = any;
        }
        if (l < 24) S.V.con.P[0][l] = S.V.con.active[l / 3] ? S.warm[l] : 0.0f;
    });

    if (uniform(S.V.con.any_active)) {
        // ---- C1: free twists of the two foot bodies (velocity FK down each leg), world aligned ----
        wave.par([&](int l) {
            if (l < 2) {
                float Rw0[9], vcur[6];
                for (int i = 0; i < 9; ++i) Rw0[i] = S.RwK[0][i];
                m3tv(Rw0, S.B.post.wwf, vcur);
                m3tv(Rw0, S.B.post.vowf, vcur + 3);
                for (int i = 1; i <= 6; ++i) {
                    const int b = 6 * l + i;
                    float R[9], vn[6];
                    for (int j = 0; j < 9; ++j) R[j] = S.R[b][j];
                    xform_motion(R, S.tree.pos[b], vcur, vn);
                    const float qd = S.B.post.qdf[b - 1];
                    vn[0] += S.tree.axis[b][0] * qd; vn[1] += S.tree.axis[b][1] * qd; vn[2] += S.tree.axis[b][2] * qd;
                    for (int j = 0; j < 6; ++j) vcur[j] = vn[j];
                }
                float o[3];
                m3v(S.RwK[1 + l], vcur, o);
                for (int i = 0; i < 3; ++i) S.V.con.twf[l][i] = o[i];
                m3v(S.RwK[1 + l], vcur + 3, o);
                for (int i = 0; i < 3; ++i) S.V.con.twf[l][3 + i] = o[i];
            }
        });
        DW_CKPT(7);
        // ---- C2: 12 unit-wrench responses -> inverse operational inertia W of the two feet ----
        wave.par([&](int l) {
            if (l < 12) {
                const int f = l / 6, comp = l % 6;
                float ew[3] = {0, 0, 0}, eb[3];
                ew[comp % 3] = 1.0f;
                m3tv(S.RwK[1 + f], ew, eb);
                float dp[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 3; ++i) dp[(comp < 3 ? 0 : 3) + i] = -eb[i];
                for (int i = 6; i >= 1; --i) {
                    const int b = 6 * f + i;
                    const float *s = S.tree.axis[b];
                    const float d = -(s[0] * dp[0] + s[1] * dp[1] + s[2] * dp[2]);
                    S.V.con.ducol[l][i - 1] = d;
                    const float k = d * S.C.art.Dinv[b];
                    float pa[6], R[9], up[6];
                    for (int j = 0; j < 6; ++j) pa[j] = dp[j] + S.C.art.U[b][j] * k;
                    for (int j = 0; j < 9; ++j) R[j] = S.R[b][j];
                    xform_force(R, S.tree.pos[b], pa, up);
                    for (int j = 0; j < 6; ++j) dp[j] = up[j];
                }
                float dv0[6];
                for (int r = 0; r < 6; ++r) {
                    float acc = 0.0f;
                    for (int c = 0; c < 6; ++c) acc -= S.B.post.Minv[6 * r + c] * dp[c];
                    dv0[r] = acc;
                }
                for (int g = 0; g < 2; ++g) {
                    float dv[6];
                    for (int j = 0; j < 6; ++j) dv[j] = dv0[j];
                    for (int i = 1; i <= 6; ++i) {
                        const int b = 6 * g + i;
                        const float *s = S.tree.axis[b];
                        float R[9], ap[6];
                        for (int j = 0; j < 9; ++j) R[j] = S.R[b][j];
                        xform_motion(R, S.tree.pos[b], dv, ap);
                        float ua = 0.0f;
                        for (int j = 0; j < 6; ++j) ua += S.C.art.U[b][j] * ap[j];
                        const float qdd = ((g == f ? S.V.con.ducol[l][i - 1] : 0.0f) - ua) * S.C.art.Dinv[b];
                        ap[0] += s[0] * qdd; ap[1] += s[1] * qdd; ap[2] += s[2] * qdd;
                        for (int j = 0; j < 6; ++j) dv[j] = ap[j];
                    }
                    float o[3];
                    m3v(S.RwK[1 + g], dv, o);
                    for (int i = 0; i < 3; ++i) S.V.con.W[6 * g + i][l] = o[i];
                    m3v(S.RwK[1 + g], dv + 3, o);
                    for (int i = 0; i < 3; ++i) S.V.con.W[6 * g + 3 + i][l] = o[i];
                }
            }
        });
        DW_CKPT(8);
        // ---- C3: Delassus matrix A = J W J', free constraint velocities, warm start ----
        wave.par([&](int l) {
            if (l < 24) {
                const int k = l / 3, ax = l % 3, f = k / 4;
                const float r[3] = {S.V.con.rk[k][0], S.V.con.rk[k][1], S.V.con.rk[k][2]};
                // row of J on foot f: angular part = -skew(r)[ax][:] , linear part = e_ax
                float ja[3];
                ja[0] = ax == 1 ? -r[2] : (ax == 2 ? r[1] : 0.0f);
                ja[1] = ax == 0 ? r[2] : (ax == 2 ? -r[0] : 0.0f);
                ja[2] = ax == 0 ? -r[1] : (ax == 1 ? r[0] : 0.0f);
                float JW0[6], JW1[6];
                for (int c = 0; c < 6; ++c) {
                    JW0[c] = ja[0] * S.V.con.W[6 * f][c] + ja[1] * S.V.con.W[6 * f + 1][c] + ja[2] * S.V.con.W[6 * f + 2][c] +
                             S.V.con.W[6 * f + 3 + ax][c];
                    JW1[c] = ja[0] * S.V.con.W[6 * f][6 + c] + ja[1] * S.V.con.W[6 * f + 1][6 + c] + ja[2] * S.V.con.W[6 * f + 2][6 + c] +
                             S.V.con.W[6 * f + 3 + ax][6 + c];
                }
                for (int k2 = 0; k2 < 4; ++k2) {
                    const float *r2 = S.V.con.rk[k2];
                    // column (k2, ax2): sum_j JW[6 f2 + j] * (-skew(r2)[ax2][j]) + JW[6 f2 + 3 + ax2]
                    S.A.lcp.A[l][3 * k2 + 0] = JW0[1] * r2[2] - JW0[2] * r2[1] + JW0[3];
                    S.A.lcp.A[l][3 * k2 + 1] = -JW0[0] * r2[2] + JW0[2] * r2[0] + JW0[4];
                    S.A.lcp.A[l][3 * k2 + 2] = JW0[0] * r2[1] - JW0[1] * r2[0] + JW0[5];
                }
                for (int k2 = 4; k2 < 8; ++k2) {
                    const float *r2 = S.V.con.rk[k2];
                    S.A.lcp.A[l][3 * k2 + 0] = JW1[1] * r2[2] - JW1[2] * r2[1] + JW1[3];
                    S.A.lcp.A[l][3 * k2 + 1] = -JW1[0] * r2[2] + JW1[2] * r2[0] + JW1[4];
                    S.A.lcp.A[l][3 * k2 + 2] = JW1[0] * r2[1] - JW1[1] * r2[0] + JW1[5];
                }
                const float *tw = S.V.con.twf[f];
                float t[3];
                cross3(tw, r, t);
                S.V.con.vel[1][l] = tw[3 + ax] + t[ax];
            }
        });
        wave.par([&](int l) {
            if (l < 24) {
                float acc = S.V.con.vel[1][l];
                for (int c = 0; c < 24; ++c) acc += S.A.lcp.A[l][c] * S.V.con.P[0][c];
                S.V.con.vel[0][l] = acc;
                S.A.lcp.invd[l] = 1.0f / (S.A.lcp.A[l][l] * (1.0f + P.cfm));
            }
        });
        DW_CKPT(9);
        // ---- C4: projected Gauss-Seidel, block-Jacobi across the feet.  Corner kk of the left sole (rows 3kk..) and
        //      corner kk of the right sole (rows 12+3kk..) are updated together from the same velocity snapshot --
        //      the feet only couple through the trunk -- and the four corners of a sole sequentially: 4 updates per
        //      sweep instead of 8.  Per contact: normal row, friction rows with the normal's effect folded in,
        //      projection onto the Coulomb cone.
#if defined(__HIPCC__)
        //      Device form: lane r < 24 keeps row r of A (24 registers), its velocity and impulse in registers; the
        //      scalars of both contacts of a pair are broadcast with v_readlane (compile-time lane ids) and the
        //      impulse arithmetic runs wave-uniformly, so there is no LDS traffic inside the solver at all.
        int cur = 0;
        {
            int act[DW_NUM_FOOT_PTS];
            for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) act[k] = uniform(S.V.con.active[k]);
            const int l = (int)threadIdx.x;
            const int row = l < 24 ? l : 0;
            float Arow[24];
            for (int c = 0; c < 24; ++c) Arow[c] = S.A.lcp.A[row][c];
            float vel = S.V.con.vel[0][row], Pl = S.V.con.P[0][row];
            const float invd = S.A.lcp.invd[row];
            float vminr[DW_NUM_FOOT_PTS];
            for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) vminr[k] = S.V.con.vmin[k];
            const float mu = S.mu;
            auto bc = [](float x, int lane) {
                return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
            };
            for (int it = 0; it < P.iters; ++it) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (!(act[kk] | act[kk + 4])) continue;
                    float D[2][3], pn[2][3];
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const int k = kk + 4 * f;
                        const int rx = 3 * k, ry = 3 * k + 1, rz = 3 * k + 2;
                        const float Pz = bc(Pl, rz), Px = bc(Pl, rx), Py = bc(Pl, ry);
                        float dz = -(bc(vel, rz) - vminr[k]) * bc(invd, rz);
                        float pz = Pz + dz;
                        if (pz < 0) pz = 0;
                        dz = pz - Pz;
                        const float vx = bc(vel, rx) + bc(Arow[rz], rx) * dz;
                        const float dx = -vx * bc(invd, rx);
                        const float vy = bc(vel, ry) + bc(Arow[rz], ry) * dz + bc(Arow[rx], ry) * dx;
                        const float dy = -vy * bc(invd, ry);
                        float px = Px + dx, py = Py + dy;
                        const float lim = mu * pz, n2 = px * px + py * py;
                        if (n2 > lim * lim) {
                            const float sc = lim * rsqrt_nr(n2);
                            px *= sc; py *= sc;
                        }
                        const bool on = act[k] != 0;
                        D[f][0] = on ? px - Px : 0.0f; D[f][1] = on ? py - Py : 0.0f; D[f][2] = on ? dz : 0.0f;
                        pn[f][0] = on ? px : Px; pn[f][1] = on ? py : Py; pn[f][2] = on ? pz : Pz;
                    }
                    vel = vel + Arow[3 * kk + 2] * D[0][2] + Arow[3 * kk] * D[0][0] + Arow[3 * kk + 1] * D[0][1]
                              + Arow[12 + 3 * kk + 2] * D[1][2] + Arow[12 + 3 * kk] * D[1][0] + Arow[12 + 3 * kk + 1] * D[1][1];
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const int r0 = 3 * (kk + 4 * f);
                        Pl = l == r0 ? pn[f][0] : (l == r0 + 1 ? pn[f][1] : (l == r0 + 2 ? pn[f][2] : Pl));
                    }
                }
            }
            if (l < 24) S.V.con.P[0][l] = Pl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#else
        //      Host-emulation form (and the readable statement of the algorithm): one region per pair of contacts,
        //      velocities and impulses ping-pong between two LDS copies so no lane reads what another lane writes.
        int cur = 0;
        for (int it = 0; it < P.iters; ++it) {
            for (int kk = 0; kk < 4; ++kk) {
                if (!(uniform(S.V.con.active[kk]) | uniform(S.V.con.active[kk + 4]))) continue;
                wave.par([&](int l) {
                    if (l < 24) {
                        const float *vel = S.V.con.vel[cur], *Pc = S.V.con.P[cur];
                        float d[2][3] = {{0, 0, 0}, {0, 0, 0}}, pn[2][3] = {{0, 0, 0}, {0, 0, 0}};
                        for (int f = 0; f < 2; ++f) {
                            const int k = kk + 4 * f;
                            const int rx = 3 * k, ry = 3 * k + 1, rz = 3 * k + 2;
                            const float Pz = Pc[rz], Px = Pc[rx], Py = Pc[ry];
                            pn[f][0] = Px; pn[f][1] = Py; pn[f][2] = Pz;
                            if (!S.V.con.active[k]) continue;
                            float dz = -(vel[rz] - S.V.con.vmin[k]) * S.A.lcp.invd[rz];
                            float pz = Pz + dz;
                            if (pz < 0) pz = 0;
                            dz = pz - Pz;
                            const float vx = vel[rx] + S.A.lcp.A[rx][rz] * dz;
                            const float dx = -vx * S.A.lcp.invd[rx];
                            const float vy = vel[ry] + S.A.lcp.A[ry][rz] * dz + S.A.lcp.A[ry][rx] * dx;
                            const float dy = -vy * S.A.lcp.invd[ry];
                            float px = Px + dx, py = Py + dy;
                            const float lim = S.mu * pz, n2 = px * px + py * py;
                            if (n2 > lim * lim) {
                                const float sc = lim * rsqrt_nr(n2);
                                px *= sc; py *= sc;
                            }
                            d[f][0] = px - Px; d[f][1] = py - Py; d[f][2] = dz;
                            pn[f][0] = px; pn[f][1] = py; pn[f][2] = pz;
                        }
                        S.V.con.vel[cur ^ 1][l] = vel[l] + S.A.lcp.A[l][3 * kk + 2] * d[0][2] + S.A.lcp.A[l][3 * kk] * d[0][0] +
                                                  S.A.lcp.A[l][3 * kk + 1] * d[0][1] + S.A.lcp.A[l][12 + 3 * kk + 2] * d[1][2] +
                                                  S.A.lcp.A[l][12 + 3 * kk] * d[1][0] + S.A.lcp.A[l][12 + 3 * kk + 1] * d[1][1];
                        float pv = Pc[l];
                        for (int f = 0; f < 2; ++f)
                            for (int i = 0; i < 3; ++i)
                                if (l == 3 * (kk + 4 * f) + i) pv = pn[f][i];
                        S.V.con.P[cur ^ 1][l] = pv;
                    }
                });
                cur ^= 1;
            }
        }
#endif
        DW_CKPT(10);
        // ---- C5: impulses -> wrenches on the two foot bodies -> delta-ABA over the whole tree ----
        wave.par([&](int l) {
            if (l < 2) {
                const float *Pc = S.V.con.P[cur];
                float F[3] = {0, 0, 0}, Nm[3] = {0, 0, 0};
                for (int k = 4 * l; k < 4 * l + 4; ++k) {
                    float t[3];
                    cross3(S.V.con.rk[k], Pc + 3 * k, t);
                    for (int i = 0; i < 3; ++i) { F[i] += Pc[3 * k + i]; Nm[i] += t[i]; }
                }
                float dp[6], fbv[3], nbv[3];
                m3tv(S.RwK[1 + l], F, fbv);
                m3tv(S.RwK[1 + l], Nm, nbv);
                for (int i = 0; i < 3; ++i) { dp[i] = -nbv[i]; dp[3 + i] = -fbv[i]; }
                for (int i = 6; i >= 1; --i) {
                    const int b = 6 * l + i;
                    const float *s = S.tree.axis[b];
                    const float d = -(s[0] * dp[0] + s[1] * dp[1] + s[2] * dp[2]);
                    S.B.post.du[b] = d;
                    const float kk = d * S.C.art.Dinv[b];
                    float pa[6], R[9], up[6];
                    for (int j = 0; j < 6; ++j) pa[j] = dp[j] + S.C.art.U[b][j] * kk;
                    for (int j = 0; j < 9; ++j) R[j] = S.R[b][j];
                    xform_force(R, S.tree.pos[b], pa, up);
                    for (int j = 0; j < 6; ++j) dp[j] = up[j];
                }
                for (int j = 0; j < 6; ++j) S.A.lcp.dpf[l][j] = dp[j];
                const int gy = M.foot_gym[4 * l];
                for (int i = 0; i < 3; ++i) S.contact[3 * gy + i] += F[i] / dt;
            }
            if (l >= 32 && l < 32 + 24) S.warm[l - 32] = S.V.con.P[cur][l - 32];
        });
        wave.par([&](int l) {
            if (l < 6) {
                float acc = 0.0f;
                for (int c = 0; c < 6; ++c) acc -= S.B.post.Minv[6 * l + c] * (S.A.lcp.dpf[0][c] + S.A.lcp.dpf[1][c]);
                S.B.post.a[0][l] = acc;
                S.B.post.dv0[l] = acc;
            }
        });
        for (int L = 1; L <= S.tree.nlevels; ++L) {       // same two-region, three-lane form as A4, without the bias term
            wave.par([&](int l) {
                const int k = l / 3, c = l - 3 * k;
                if (k < S.tree.level_count[L]) {
                    const int b = S.tree.level_body[L][k], p = S.tree.parent[b];
                    const float *Rb = S.R[b], *pos = S.tree.pos[b], *ap = S.B.post.a[p];
                    const float r0 = Rb[c], r1 = Rb[3 + c], r2 = Rb[6 + c];
                    const float t0 = ap[1] * pos[2] - ap[2] * pos[1], t1 = ap[2] * pos[0] - ap[0] * pos[2], t2 = ap[0] * pos[1] - ap[1] * pos[0];
                    const float ang = r0 * ap[0] + r1 * ap[1] + r2 * ap[2];
                    const float lin = r0 * (ap[3] + t0) + r1 * (ap[4] + t1) + r2 * (ap[5] + t2);
                    S.B.post.a[b][c] = ang;
                    S.B.post.a[b][3 + c] = lin;
                    S.B.post.part[k][c] = S.C.art.U[b][c] * ang + S.C.art.U[b][3 + c] * lin;
                }
            });
            wave.par([&](int l) {
                const int k = l / 3, c = l - 3 * k;
                if (k < S.tree.level_count[L]) {
                    const int b = S.tree.level_body[L][k];
                    const float ua = S.B.post.part[k][0] + S.B.post.part[k][1] + S.B.post.part[k][2];
                    const float dq = (S.B.post.du[b] - ua) * S.C.art.Dinv[b];
                    if (c == 0) S.B.post.dqd[b - 1] = dq;
                    S.B.post.a[b][c] += S.tree.axis[b][c] * dq;
                }
            });
        }
    } else {
        wave.par([&](int l) {
            if (l < 24) S.warm[l] = 0.0f;
        });
    }

    DW_CKPT(11);
    // ---- V2: final velocities, clamps, semi-implicit Euler ----
    wave.par([&](int l) {
        if (l < ND) {
            float qd = S.B.post.qdf[l] + S.B.post.dqd[l];
            const float vm = M.vmax[l];
            if (qd > vm) qd = vm;
            if (qd < -vm) qd = -vm;
            float q = S.q[l] + dt * qd;
            if (q < M.qlo[l]) { q = M.qlo[l]; if (qd < 0) qd = 0; }
            if (q > M.qhi[l]) { q = M.qhi[l]; if (qd > 0) qd = 0; }
            S.q[l] = q;
            S.qd[l] = qd;
        }
        if (l == 40) {
            float Rw[9], wwn[3], von[3], t[3];
            for (int i = 0; i < 9; ++i) Rw[i] = S.RwK[0][i];
            m3v(Rw, S.B.post.dv0, t);
            for (int i = 0; i < 3; ++i) wwn[i] = S.B.post.wwf[i] + t[i];
            m3v(Rw, S.B.post.dv0 + 3, t);
            for (int i = 0; i < 3; ++i) von[i] = S.B.post.vowf[i] + t[i];
            {
                const float wn2 = dot3(wwn, wwn);
                if (wn2 > P.max_ang_vel * P.max_ang_vel) {
                    const float sc = P.max_ang_vel * rsqrt_nr(wn2);
                    wwn[0] *= sc; wwn[1] *= sc; wwn[2] *= sc;
                }
            }
            for (int i = 0; i < 3; ++i) S.root[i] += dt * von[i];
            // dq = [w_hat sin(th/2), cos(th/2)], th = |w| dt <= 0.2 rad: sin(th/2)/|w| = (dt/2) * sinc(th/2)
            const float w2 = dot3(wwn, wwn);
            const float hx = 0.5f * dt;
            float sh, ch;
            {   // sinc and cos of x = |w| dt / 2 from x^2 only (no square root needed)
                const float x2 = w2 * hx * hx;
                sh = hx * (1.0f + x2 * (-1.0f / 6 + x2 * (1.0f / 120 + x2 * (-1.0f / 5040 + x2 * (1.0f / 362880)))));
                ch = 1.0f + x2 * (-0.5f + x2 * (1.0f / 24 + x2 * (-1.0f / 720 + x2 * (1.0f / 40320))));
            }
            const float dq[4] = {wwn[0] * sh, wwn[1] * sh, wwn[2] * sh, ch};
            const float x1 = dq[0], y1 = dq[1], z1 = dq[2], w1 = dq[3];
            const float x2q = S.quat[0], y2 = S.quat[1], z2 = S.quat[2], w2q = S.quat[3];
            float qn[4] = {w1 * x2q + x1 * w2q + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2q + z1 * x2q,
                           w1 * z2 + x1 * y2 - y1 * x2q + z1 * w2q, w1 * w2q - x1 * x2q - y1 * y2 - z1 * z2};
            const float ninv = rsqrt_nr(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
            for (int i = 0; i < 4; ++i) { qn[i] *= ninv; S.root[3 + i] = qn[i]; }
            if (P.vel_at_com) {
                float Rn[9], rc[3], tt[3];
                quat_to_mat(qn, Rn);
                m3v(Rn, M.inert_com[0], rc);
                cross3(wwn, rc, tt);
                for (int i = 0; i < 3; ++i) von[i] += tt[i];
            }
            for (int i = 0; i < 3; ++i) { S.root[7 + i] = von[i]; S.root[10 + i] = wwn[i]; }
        }
    });
}

}  // namespace dw
