/*
 * dw_physics.c -- CPU ORACLE, physics substep.  TEST INFRASTRUCTURE ONLY (see dw_physics.h).
 *
 * Spatial-vector conventions (Featherstone, body coordinates): motion vectors [w; v] with v the
 * velocity of the body-fixed point at the body origin, force vectors [n; f] with n the moment about
 * the body origin.  A general symmetric 6x6 inertia is kept as three 3x3 blocks {A, B, C} meaning
 * [[A, B], [B', C]].  R_b ("Rb2p") maps body-b coordinates to parent coordinates, p_b is the body
 * origin in the parent frame, s_b the hinge axis in body coordinates.
 */
#include "dw_physics.h"

#include <math.h>
#include <string.h>

#define NB DW_NUM_MOVING
#define ND DW_NUM_DOF

#ifdef DWO_DOUBLE
#define RSQRT(x) sqrt(x)
#define RSIN(x) sin(x)
#define RCOS(x) cos(x)
#define RABS(x) fabs(x)
#else
#define RSQRT(x) sqrtf(x)
#define RSIN(x) sinf(x)
#define RCOS(x) cosf(x)
#define RABS(x) fabsf(x)
#endif

typedef struct { real A[9], B[9], C[9]; } SI;

/* ---------------- small linear algebra ---------------- */
static inline void cross3(const real a[3], const real b[3], real o[3]) {
    real x = a[1] * b[2] - a[2] * b[1];
    real y = a[2] * b[0] - a[0] * b[2];
    real z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline real dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void m3v(const real M[9], const real v[3], real o[3]) {
    real x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    real y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    real z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tv(const real M[9], const real v[3], real o[3]) {
    real x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    real y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    real z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3m(const real X[9], const real Y[9], real O[9]) {
    real T[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            T[3 * r + c] = X[3 * r] * Y[c] + X[3 * r + 1] * Y[3 + c] + X[3 * r + 2] * Y[6 + c];
    memcpy(O, T, sizeof(T));
}
static inline void m3mt(const real X[9], const real Y[9], real O[9]) { /* X Y' */
    real T[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            T[3 * r + c] = X[3 * r] * Y[3 * c] + X[3 * r + 1] * Y[3 * c + 1] + X[3 * r + 2] * Y[3 * c + 2];
    memcpy(O, T, sizeof(T));
}
static inline void skew3(const real p[3], real P[9]) {
    P[0] = 0; P[1] = -p[2]; P[2] = p[1];
    P[3] = p[2]; P[4] = 0; P[5] = -p[0];
    P[6] = -p[1]; P[7] = p[0]; P[8] = 0;
}
static void quat_to_mat(const real q[4] /* xyzw */, real R[9]) {
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

/* motion transform parent -> body */
static inline void xform_motion(const real R[9], const real p[3], const real vp[6], real vb[6]) {
    real t[3], l[3];
    m3tv(R, vp, vb);                 /* w_b = R' w_p */
    cross3(vp, p, t);                /* w_p x p */
    l[0] = vp[3] + t[0]; l[1] = vp[4] + t[1]; l[2] = vp[5] + t[2];
    m3tv(R, l, vb + 3);
}
/* force transform body -> parent, accumulating */
static inline void xform_force_add(const real R[9], const real p[3], const real fb[6], real fp[6]) {
    real n[3], f[3], t[3];
    m3v(R, fb, n);
    m3v(R, fb + 3, f);
    cross3(p, f, t);
    fp[0] += n[0] + t[0]; fp[1] += n[1] + t[1]; fp[2] += n[2] + t[2];
    fp[3] += f[0]; fp[4] += f[1]; fp[5] += f[2];
}

/* 6x6 Cholesky (lower) of [[A,B],[B',C]]; returns 0 on success */
static int chol6(const SI *I, real L[36]) {
    real M[36];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            M[6 * r + c] = I->A[3 * r + c];
            M[6 * r + 3 + c] = I->B[3 * r + c];
            M[6 * (r + 3) + c] = I->B[3 * c + r];
            M[6 * (r + 3) + 3 + c] = I->C[3 * r + c];
        }
    memset(L, 0, 36 * sizeof(real));
    for (int j = 0; j < 6; ++j) {
        real d = M[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
        if (!(d > 0)) return -1;
        d = RSQRT(d);
        L[6 * j + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            real s = M[6 * i + j];
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
            L[6 * i + j] = s / d;
        }
    }
    return 0;
}
static void chol6_solve(const real L[36], const real b[6], real x[6]) {
    real y[6];
    for (int i = 0; i < 6; ++i) {
        real s = b[i];
        for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        real s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
}

/* ---------------- per-substep workspace ---------------- */
typedef struct {
    real R[NB][9];      /* body -> parent */
    real Rw[NB][9];     /* body -> world  */
    real pw[NB][3];     /* origin, world  */
    real v[NB][6];      /* body-frame spatial velocity */
    real c[NB][6];
    real U[NB][6];
    real Dinv[NB];
    real u[NB];
    real L0[36];        /* Cholesky factor of the base articulated inertia */
} Work;

/* delta-ABA: dp[b] = change of the bias force of body b (minus the applied wrench, body coords).
 * Upward over bodies up_hi..1 (descending), downward over 1..down_hi.  Returns base delta twist and
 * per-body delta twists / joint delta rates.  dp is destroyed. */
static void delta_aba(const DwoModelR *m, const Work *w, real dp[NB][6], int up_lo, int up_hi, int down_hi,
                      real dv[NB][6], real dqd[ND]) {
    real du[NB];
    for (int b = 1; b < NB; ++b) du[b] = 0;
    for (int b = up_hi; b >= up_lo; --b) {
        const real *s = m->mv_axis[b];
        du[b] = -dot3(s, dp[b]);
        real k = du[b] * w->Dinv[b];
        real pa[6];
        for (int i = 0; i < 6; ++i) pa[i] = dp[b][i] + w->U[b][i] * k;
        xform_force_add(w->R[b], m->mv_pos[b], pa, dp[m->mv_parent[b]]);
    }
    real rhs[6];
    for (int i = 0; i < 6; ++i) rhs[i] = -dp[0][i];
    chol6_solve(w->L0, rhs, dv[0]);
    for (int b = 1; b <= down_hi; ++b) {
        const real *s = m->mv_axis[b];
        real ap[6];
        xform_motion(w->R[b], m->mv_pos[b], dv[m->mv_parent[b]], ap);
        real ua = 0;
        for (int i = 0; i < 6; ++i) ua += w->U[b][i] * ap[i];
        real qdd = (du[b] - ua) * w->Dinv[b];
        dqd[b - 1] = qdd;
        dv[b][0] = ap[0] + s[0] * qdd; dv[b][1] = ap[1] + s[1] * qdd; dv[b][2] = ap[2] + s[2] * qdd;
        dv[b][3] = ap[3]; dv[b][4] = ap[4]; dv[b][5] = ap[5];
    }
}

void dwo_model_to_real(const DwModel *m, DwoModelR *r) {
    for (int b = 0; b < NB; ++b) {
        r->mv_parent[b] = m->mv_parent[b];
        for (int i = 0; i < 3; ++i) { r->mv_pos[b][i] = m->mv_pos[b][i]; r->mv_axis[b][i] = m->mv_axis[b][i]; }
        for (int i = 0; i < 9; ++i) r->mv_rot0[b][i] = m->mv_rot0[b][i];
    }
    for (int j = 0; j < ND; ++j) {
        r->dof_lower[j] = m->dof_lower[j]; r->dof_upper[j] = m->dof_upper[j]; r->dof_vmax[j] = m->dof_vmax[j];
    }
    for (int k = 0; k < DW_NUM_INERT; ++k) {
        r->inert_mv[k] = m->inert_mv[k]; r->inert_gym[k] = m->inert_gym[k]; r->inert_mass[k] = m->inert_mass[k];
        for (int i = 0; i < 3; ++i) r->inert_com[k][i] = m->inert_com[k][i];
        for (int i = 0; i < 6; ++i) r->inert_I[k][i] = m->inert_I[k][i];
    }
    r->num_geoms = m->num_geoms;
    for (int g = 0; g < m->num_geoms; ++g) {
        const DwGeom *s = &m->geoms[g];
        DwoGeomR *d = &r->geoms[g];
        d->type = s->type; d->moving = s->moving; d->gym = s->gym; d->sole = s->sole;
        for (int i = 0; i < 3; ++i) { d->pos[i] = s->pos[i]; d->size[i] = s->size[i]; }
        for (int i = 0; i < 9; ++i) d->rot[i] = s->rot[i];
    }
    for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) {
        r->foot_mv[k] = m->foot_mv[k]; r->foot_gym[k] = m->foot_gym[k];
        for (int i = 0; i < 3; ++i) r->foot_pos[k][i] = m->foot_pos[k][i];
    }
    r->num_sc_proxies = m->num_sc_proxies; r->num_sc_pairs = m->num_sc_pairs;
    for (int k = 0; k < m->num_sc_proxies; ++k) {
        r->sc_moving[k] = m->sc_proxy[k].moving; r->sc_gym[k] = m->sc_proxy[k].gym; r->sc_radius[k] = m->sc_proxy[k].radius;
        for (int i = 0; i < 3; ++i) { r->sc_p0[k][i] = m->sc_proxy[k].p0[i]; r->sc_p1[k][i] = m->sc_proxy[k].p1[i]; }
    }
    for (int k = 0; k < m->num_sc_pairs; ++k) { r->sc_pair[k][0] = m->sc_pair[k][0]; r->sc_pair[k][1] = m->sc_pair[k][1]; }
}

static inline real clamp01(real x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }

/* closest points of segments p1 + s d1 and p2 + t d2, s, t in [0,1] (Ericson, Real-Time Collision Detection 5.1.9) */
static void seg_seg(const real p1[3], const real d1[3], const real p2[3], const real d2[3], real *so, real *to) {
    real r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    real a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r);
    const real eps = (real)1e-12;
    real s, t;
    if (a <= eps && e <= eps) { s = 0; t = 0; }
    else if (a <= eps) { s = 0; t = clamp01(f / e); }
    else {
        real c = dot3(d1, r);
        if (e <= eps) { t = 0; s = clamp01(-c / a); }
        else {
            /* Nearly parallel segments make the textbook quotient (b f - c e) / den a ratio of two rounding errors: the
             * contact point -- and with it the moment of the contact force -- then jumps along the overlap of the two
             * capsules from one platform (or operation order) to the next.  Written decision (DESIGN.md "Physics model"):
             * the result is blended, with weight w = den^2 / (den^2 + (k a e)^2), k = 1e-3 (w = 1/2 at 1.8 deg), between
             * the exact closest points (se, te) and the "side by side" answer (sp, tp) = middle of the overlap of segment 2's
             * projection on segment 1 and the point of segment 2 closest to it.  Both are continuous where their weight
             * is not negligible; crossing capsules keep their closest points, parallel ones touch mid-overlap (where an
             * engine with contact manifolds puts the resultant).  The distance changes by O(1e-5 m) at most. */
            real b = dot3(d1, d2), den = a * e - b * b;
            real se = den > eps ? clamp01((b * f - c * e) / den) : 0;
            real te = (b * se + f) / e;
            if (te < 0) { te = 0; se = clamp01(-c / a); }
            else if (te > 1) { te = 1; se = clamp01((b - c) / a); }
            real t0 = -c / a, t1 = t0 + b / a;
            real lo = t0 < t1 ? t0 : t1, hi = t0 < t1 ? t1 : t0;
            if (lo < 0) lo = 0;
            if (hi > 1) hi = 1;
            real sp = clamp01((real)0.5 * (lo + hi));
            real tp = clamp01((b * sp + f) / e);
            real reg = (real)1e-3 * a * e;
            real w = den > eps ? den * den / (den * den + reg * reg) : 0;
            s = w * se + (1 - w) * sp;
            t = w * te + (1 - w) * tp;
        }
    }
    *so = s; *to = t;
}

/* Height field (row f-4).  The ground under a world point: bilinear interpolation of the four samples around it
 * (world x = row * hscale - border, y = col * hscale - border, reference tasks/dyros_dynamic_walk.py:240-253), unit
 * normal from the gradient of that patch, and a contact frame (t1, t2, n) with t1 = the world x axis projected into the
 * tangent plane.  On the plane: h = 0, n = z, t1 = x, t2 = y. */
static void terrain_sample(const DwConfig *cfg, const int16_t *hs, real x, real y, real *h, real fr[9]) {
    const real inv = (real)1 / cfg->terrain_hscale;
    real u = (x + cfg->terrain_border) * inv, v = (y + cfg->terrain_border) * inv;
    const real umax = (real)(cfg->terrain_rows - 1) - (real)1e-3, vmax = (real)(cfg->terrain_cols - 1) - (real)1e-3;
    if (u < 0) u = 0;
    if (u > umax) u = umax;
    if (v < 0) v = 0;
    if (v > vmax) v = vmax;
    const int i = (int)u, j = (int)v;
    const real a = u - (real)i, b = v - (real)j;
    const int16_t *p = hs + (size_t)i * cfg->terrain_cols + j;
    const real h00 = p[0], h01 = p[1], h10 = p[cfg->terrain_cols], h11 = p[cfg->terrain_cols + 1];
    const real vs = cfg->terrain_vscale;
    *h = vs * (((real)1 - a) * (((real)1 - b) * h00 + b * h01) + a * (((real)1 - b) * h10 + b * h11));
    const real gx = vs * inv * (((real)1 - b) * (h10 - h00) + b * (h11 - h01));
    const real gy = vs * inv * (((real)1 - a) * (h01 - h00) + a * (h11 - h10));
    const real nn = (real)1 / RSQRT(gx * gx + gy * gy + (real)1);
    real *t1 = fr, *t2 = fr + 3, *n = fr + 6;
    n[0] = -gx * nn; n[1] = -gy * nn; n[2] = nn;
    /* t1 = normalize(x - (x.n) n) */
    real a1[3] = {(real)1 - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    const real tn = (real)1 / RSQRT(dot3(a1, a1));
    t1[0] = a1[0] * tn; t1[1] = a1[1] * tn; t1[2] = a1[2] * tn;
    cross3(n, t1, t2);
}

void dwo_phys_substep(const DwConfig *cfg, const DwoModelR *m, DwoPhysIO *io) {
    Work w;
    SI IA[NB];
    real pA[NB][6];
    const real dt = (real)cfg->dt;
    const real g[3] = {cfg->gravity[0], cfg->gravity[1], cfg->gravity[2]};

    /* ---- 0. base state ---- */
    real quat[4] = {io->root[3], io->root[4], io->root[5], io->root[6]};
    {
        real n = RSQRT(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
        for (int i = 0; i < 4; ++i) quat[i] /= n;
    }
    quat_to_mat(quat, w.Rw[0]);
    for (int i = 0; i < 9; ++i) w.R[0][i] = w.Rw[0][i];
    for (int i = 0; i < 3; ++i) w.pw[0][i] = io->root[i];
    real ww[3] = {io->root[10], io->root[11], io->root[12]};     /* world angular velocity */
    real vow[3] = {io->root[7], io->root[8], io->root[9]};       /* world velocity of the base origin */
    /* base COM in body coords (inertial record 0 belongs to the base) */
    const real *c0 = m->inert_com[0];
    if (cfg->root_vel_at_com) {
        real rc[3], t[3];
        m3v(w.Rw[0], c0, rc);
        cross3(ww, rc, t);
        vow[0] -= t[0]; vow[1] -= t[1]; vow[2] -= t[2];
    }
    m3tv(w.Rw[0], ww, w.v[0]);
    m3tv(w.Rw[0], vow, w.v[0] + 3);
    for (int i = 0; i < 6; ++i) w.c[0][i] = 0;

    /* ---- 1. kinematics ---- */
    for (int b = 1; b < NB; ++b) {
        const int p = m->mv_parent[b];
        const real *s = m->mv_axis[b];
        real q = io->q[b - 1], qd = io->qd[b - 1];
        real sn = RSIN(q), cs = RCOS(q), oc = 1 - cs;
        real Rj[9] = {cs + oc * s[0] * s[0], oc * s[0] * s[1] - sn * s[2], oc * s[0] * s[2] + sn * s[1],
                      oc * s[1] * s[0] + sn * s[2], cs + oc * s[1] * s[1], oc * s[1] * s[2] - sn * s[0],
                      oc * s[2] * s[0] - sn * s[1], oc * s[2] * s[1] + sn * s[0], cs + oc * s[2] * s[2]};
        m3m(m->mv_rot0[b], Rj, w.R[b]);
        m3m(w.Rw[p], w.R[b], w.Rw[b]);
        real t[3];
        m3v(w.Rw[p], m->mv_pos[b], t);
        for (int i = 0; i < 3; ++i) w.pw[b][i] = w.pw[p][i] + t[i];
        xform_motion(w.R[b], m->mv_pos[b], w.v[p], w.v[b]);
        real sq[3] = {s[0] * qd, s[1] * qd, s[2] * qd};
        w.v[b][0] += sq[0]; w.v[b][1] += sq[1]; w.v[b][2] += sq[2];
        cross3(w.v[b], sq, w.c[b]);
        cross3(w.v[b] + 3, sq, w.c[b] + 3);
    }

    /* ---- 2. rigid-body inertias and bias forces ---- */
    for (int b = 0; b < NB; ++b) {
        memset(&IA[b], 0, sizeof(SI));
        for (int i = 0; i < 6; ++i) pA[b][i] = 0;
    }
    for (int k = 0; k < DW_NUM_INERT; ++k) {
        const int b = m->inert_mv[k];
        const real ms = io->mass_scale[m->inert_gym[k]];
        const real mass = ms * m->inert_mass[k];
        const real *cm = m->inert_com[k];
        const real *I6 = m->inert_I[k];
        real cc = dot3(cm, cm);
        real Ic[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                real par = mass * ((r == c ? cc : 0) - cm[r] * cm[c]);
                IA[b].A[3 * r + c] += ms * Ic[3 * r + c] + par;
            }
        real h[3] = {mass * cm[0], mass * cm[1], mass * cm[2]}, H[9];
        skew3(h, H);
        for (int i = 0; i < 9; ++i) IA[b].B[i] += H[i];
        IA[b].C[0] += mass; IA[b].C[4] += mass; IA[b].C[8] += mass;
    }
    for (int b = 0; b < NB; ++b) {
        /* pA = v x* (I v) */
        real n[3], f[3], t1[3], t2[3];
        const real *om = w.v[b], *vl = w.v[b] + 3;
        m3v(IA[b].A, om, n); m3v(IA[b].B, vl, t1);
        n[0] += t1[0]; n[1] += t1[1]; n[2] += t1[2];
        m3tv(IA[b].B, om, f); m3v(IA[b].C, vl, t1);
        f[0] += t1[0]; f[1] += t1[1]; f[2] += t1[2];
        cross3(om, n, t1); cross3(vl, f, t2);
        pA[b][0] = t1[0] + t2[0]; pA[b][1] = t1[1] + t2[1]; pA[b][2] = t1[2] + t2[2];
        cross3(om, f, pA[b] + 3);
    }

    /* ---- 3. external forces: push on the base COM, penalty contacts of non-sole primitives ---- */
    for (int i = 0; i < DW_NUM_BODIES * 3; ++i) io->contact[i] = 0;
    {
        real Fw[3] = {io->push[0], io->push[1], 0}, fb[3], nb[3];
        m3tv(w.Rw[0], Fw, fb);
        cross3(c0, fb, nb);
        for (int i = 0; i < 3; ++i) { pA[0][i] -= nb[i]; pA[0][3 + i] -= fb[i]; }
    }
    for (int gi = 0; gi < m->num_geoms; ++gi) {
        const DwoGeomR *ge = &m->geoms[gi];
        if (ge->sole) continue;
        const int b = ge->moving;
        real rl[3] = {0, 0, 0};   /* deepest point, body coords */
        real zmin;
        if (ge->type == 0) {
            /* deepest corner of the box: along each box axis take the end that points down */
            real Rg[9], e[3], l[3], wv[3];
            m3m(w.Rw[b], ge->rot, Rg);
            for (int i = 0; i < 3; ++i) e[i] = (Rg[6 + i] > 0 ? (real)-1 : (real)1) * ge->size[i];
            m3v(ge->rot, e, l);
            l[0] += ge->pos[0]; l[1] += ge->pos[1]; l[2] += ge->pos[2];
            m3v(w.Rw[b], l, wv);
            zmin = w.pw[b][2] + wv[2];
            rl[0] = l[0]; rl[1] = l[1]; rl[2] = l[2];
        } else {
            /* cylinder: lowest point of the lower cap rim */
            real al[3] = {ge->rot[2], ge->rot[5], ge->rot[8]};   /* axis, body coords */
            real aw[3];
            m3v(w.Rw[b], al, aw);
            real sgn = aw[2] >= 0 ? (real)-1 : (real)1;          /* move towards the lower cap */
            real dw[3] = {-aw[2] * aw[0], -aw[2] * aw[1], 1 - aw[2] * aw[2]};   /* z - (z.a) a */
            real dn = RSQRT(dot3(dw, dw));
            real off[3] = {0, 0, 0};
            if (dn > (real)1e-6) {
                real k = -ge->size[0] / dn;
                real ow[3] = {k * dw[0], k * dw[1], k * dw[2]};
                m3tv(w.Rw[b], ow, off);
            }
            for (int i = 0; i < 3; ++i) rl[i] = ge->pos[i] + sgn * ge->size[1] * al[i] + off[i];
            real wv[3];
            m3v(w.Rw[b], rl, wv);
            zmin = w.pw[b][2] + wv[2];
        }
        if (cfg->terrain && io->height_samples) {
            /* height field: signed distance along the local normal, force along it, friction in the tangent plane */
            real wv[3], hh, fr[9];
            m3v(w.Rw[b], rl, wv);
            terrain_sample(cfg, io->height_samples, w.pw[b][0] + wv[0], w.pw[b][1] + wv[1], &hh, fr);
            const real *nrm = fr + 6;
            const real dist = (zmin - hh) * nrm[2];
            if (dist < 0) {
                real vl[3], t[3], vw[3];
                cross3(w.v[b], rl, t);
                vl[0] = w.v[b][3] + t[0]; vl[1] = w.v[b][4] + t[1]; vl[2] = w.v[b][5] + t[2];
                m3v(w.Rw[b], vl, vw);
                const real vn = dot3(vw, nrm);
                real fn = cfg->penalty_stiffness * (-dist) - cfg->penalty_damping * vn;
                if (fn < 0) fn = 0;
                real vt[3] = {vw[0] - vn * nrm[0], vw[1] - vn * nrm[1], vw[2] - vn * nrm[2]};
                real sp = RSQRT(dot3(vt, vt));
                real Fw[3] = {fn * nrm[0], fn * nrm[1], fn * nrm[2]};
                if (sp > (real)1e-9) {
                    real ft = cfg->penalty_damping * sp;
                    real lim = io->mu * fn;
                    if (ft > lim) ft = lim;
                    for (int i = 0; i < 3; ++i) Fw[i] -= ft * vt[i] / sp;
                }
                real fb[3], nb[3];
                m3tv(w.Rw[b], Fw, fb);
                cross3(rl, fb, nb);
                for (int i = 0; i < 3; ++i) { pA[b][i] -= nb[i]; pA[b][3 + i] -= fb[i]; }
                for (int i = 0; i < 3; ++i) io->contact[3 * ge->gym + i] += Fw[i];
            }
        } else if (zmin < 0) {
            real vl[3], t[3], vw[3];
            cross3(w.v[b], rl, t);
            vl[0] = w.v[b][3] + t[0]; vl[1] = w.v[b][4] + t[1]; vl[2] = w.v[b][5] + t[2];
            m3v(w.Rw[b], vl, vw);
            real fn = cfg->penalty_stiffness * (-zmin) - cfg->penalty_damping * vw[2];
            if (fn < 0) fn = 0;
            real sp = RSQRT(vw[0] * vw[0] + vw[1] * vw[1]);
            real Fw[3] = {0, 0, fn};
            if (sp > (real)1e-9) {
                real ft = cfg->penalty_damping * sp;
                real lim = io->mu * fn;
                if (ft > lim) ft = lim;
                Fw[0] = -ft * vw[0] / sp; Fw[1] = -ft * vw[1] / sp;
            }
            real fb[3], nb[3];
            m3tv(w.Rw[b], Fw, fb);
            cross3(rl, fb, nb);
            for (int i = 0; i < 3; ++i) { pA[b][i] -= nb[i]; pA[b][3 + i] -= fb[i]; }
            for (int i = 0; i < 3; ++i) io->contact[3 * ge->gym + i] += Fw[i];
        }
    }

    /* ---- 3b. self-collision: capsule proxies of the two legs, penalty force along the closest-point normal ---- */
    for (int pi = 0; cfg->self_collision && pi < m->num_sc_pairs; ++pi) {
        const int ia = m->sc_pair[pi][0], ib = m->sc_pair[pi][1];
        const int ba = m->sc_moving[ia], bb = m->sc_moving[ib];
        real a0[3], a1[3], b0[3], b1[3], t3[3];
        m3v(w.Rw[ba], m->sc_p0[ia], t3); for (int i = 0; i < 3; ++i) a0[i] = w.pw[ba][i] + t3[i];
        m3v(w.Rw[ba], m->sc_p1[ia], t3); for (int i = 0; i < 3; ++i) a1[i] = w.pw[ba][i] + t3[i];
        m3v(w.Rw[bb], m->sc_p0[ib], t3); for (int i = 0; i < 3; ++i) b0[i] = w.pw[bb][i] + t3[i];
        m3v(w.Rw[bb], m->sc_p1[ib], t3); for (int i = 0; i < 3; ++i) b1[i] = w.pw[bb][i] + t3[i];
        real da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
        real sa, sb;
        seg_seg(a0, da, b0, db, &sa, &sb);
        real ca[3], cb[3], n[3];
        for (int i = 0; i < 3; ++i) { ca[i] = a0[i] + sa * da[i]; cb[i] = b0[i] + sb * db[i]; n[i] = ca[i] - cb[i]; }
        real dist = RSQRT(dot3(n, n));
        real depth = m->sc_radius[ia] + m->sc_radius[ib] - dist;
        if (depth > 0 && dist > (real)1e-6) {
            for (int i = 0; i < 3; ++i) n[i] /= dist;
            /* contact points in body coordinates, their world velocities */
            real ra[3], rb[3], va[3], vb[3], tt[3], vl[3];
            for (int i = 0; i < 3; ++i) t3[i] = ca[i] - w.pw[ba][i];
            m3tv(w.Rw[ba], t3, ra);
            cross3(w.v[ba], ra, tt);
            for (int i = 0; i < 3; ++i) vl[i] = w.v[ba][3 + i] + tt[i];
            m3v(w.Rw[ba], vl, va);
            for (int i = 0; i < 3; ++i) t3[i] = cb[i] - w.pw[bb][i];
            m3tv(w.Rw[bb], t3, rb);
            cross3(w.v[bb], rb, tt);
            for (int i = 0; i < 3; ++i) vl[i] = w.v[bb][3 + i] + tt[i];
            m3v(w.Rw[bb], vl, vb);
            real vn = (va[0] - vb[0]) * n[0] + (va[1] - vb[1]) * n[1] + (va[2] - vb[2]) * n[2];
            real fn = cfg->penalty_stiffness * depth - cfg->penalty_damping * vn;
            if (fn < 0) fn = 0;
            real Fw[3] = {fn * n[0], fn * n[1], fn * n[2]}, fb3[3], nb3[3];
            m3tv(w.Rw[ba], Fw, fb3);
            cross3(ra, fb3, nb3);
            for (int i = 0; i < 3; ++i) { pA[ba][i] -= nb3[i]; pA[ba][3 + i] -= fb3[i]; io->contact[3 * m->sc_gym[ia] + i] += Fw[i]; }
            real Fm[3] = {-Fw[0], -Fw[1], -Fw[2]};
            m3tv(w.Rw[bb], Fm, fb3);
            cross3(rb, fb3, nb3);
            for (int i = 0; i < 3; ++i) { pA[bb][i] -= nb3[i]; pA[bb][3 + i] -= fb3[i]; io->contact[3 * m->sc_gym[ib] + i] += Fm[i]; }
        }
    }

    /* ---- 4. ABA pass 2 (inward) ---- */
    for (int b = NB - 1; b >= 1; --b) {
        const int p = m->mv_parent[b];
        const real *s = m->mv_axis[b];
        real Ua[3], Ul[3];
        m3v(IA[b].A, s, Ua);
        m3tv(IA[b].B, s, Ul);
        real D = dot3(s, Ua) + io->armature[b - 1] + dt * io->damping[b - 1];
        real Dinv = 1 / D;
        real u = io->tau[b - 1] - io->damping[b - 1] * io->qd[b - 1] - dot3(s, pA[b]);
        for (int i = 0; i < 3; ++i) { w.U[b][i] = Ua[i]; w.U[b][3 + i] = Ul[i]; }
        w.Dinv[b] = Dinv; w.u[b] = u;
        SI Ia = IA[b];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                Ia.A[3 * r + c] -= Ua[r] * Ua[c] * Dinv;
                Ia.B[3 * r + c] -= Ua[r] * Ul[c] * Dinv;
                Ia.C[3 * r + c] -= Ul[r] * Ul[c] * Dinv;
            }
        real pa[6], t1[3], t2[3];
        const real *ca = w.c[b], *cl = w.c[b] + 3;
        m3v(Ia.A, ca, t1); m3v(Ia.B, cl, t2);
        real k = u * Dinv;
        for (int i = 0; i < 3; ++i) pa[i] = pA[b][i] + t1[i] + t2[i] + Ua[i] * k;
        m3tv(Ia.B, ca, t1); m3v(Ia.C, cl, t2);
        for (int i = 0; i < 3; ++i) pa[3 + i] = pA[b][3 + i] + t1[i] + t2[i] + Ul[i] * k;
        /* I_p += X' Ia X */
        real Ar[9], Br[9], Cr[9], P[9], T[9], T2[9];
        m3m(w.R[b], Ia.A, T); m3mt(T, w.R[b], Ar);
        m3m(w.R[b], Ia.B, T); m3mt(T, w.R[b], Br);
        m3m(w.R[b], Ia.C, T); m3mt(T, w.R[b], Cr);
        skew3(m->mv_pos[b], P);
        /* A_p = Ar - Br P + P Br' - P Cr P ; B_p = Br + P Cr ; C_p = Cr */
        m3m(Br, P, T);                 /* Br P */
        m3mt(P, Br, T2);               /* P Br' */
        real PC[9], PCP[9];
        m3m(P, Cr, PC);
        m3m(PC, P, PCP);
        for (int i = 0; i < 9; ++i) {
            IA[p].A[i] += Ar[i] - T[i] + T2[i] - PCP[i];
            IA[p].B[i] += Br[i] + PC[i];
            IA[p].C[i] += Cr[i];
        }
        xform_force_add(w.R[b], m->mv_pos[b], pa, pA[p]);
    }

    /* ---- 5. base acceleration and pass 3 ---- */
    real a[NB][6];
    int bad = chol6(&IA[0], w.L0);
    if (bad) {   /* cannot happen with a physical model; poison the state so the caller's finite check fires */
        for (int i = 0; i < 13; ++i) io->root[i] = NAN;
        return;
    }
    {
        real rhs[6];
        for (int i = 0; i < 6; ++i) rhs[i] = -pA[0][i];
        chol6_solve(w.L0, rhs, a[0]);
    }
    for (int b = 1; b < NB; ++b) {
        const real *s = m->mv_axis[b];
        real ap[6];
        xform_motion(w.R[b], m->mv_pos[b], a[m->mv_parent[b]], ap);
        for (int i = 0; i < 6; ++i) ap[i] += w.c[b][i];
        real ua = 0;
        for (int i = 0; i < 6; ++i) ua += w.U[b][i] * ap[i];
        real qdd = (w.u[b] - ua) * w.Dinv[b];
        io->qdd_free[b - 1] = qdd;
        a[b][0] = ap[0] + s[0] * qdd; a[b][1] = ap[1] + s[1] * qdd; a[b][2] = ap[2] + s[2] * qdd;
        a[b][3] = ap[3]; a[b][4] = ap[4]; a[b][5] = ap[5];
    }
    real gb[3];
    m3tv(w.Rw[0], g, gb);
    for (int i = 0; i < 3; ++i) { io->a0_free[i] = a[0][i]; io->a0_free[3 + i] = a[0][3 + i] + gb[i]; }

    /* ---- 6. unconstrained velocity update ---- */
    real qdf[ND];
    for (int j = 0; j < ND; ++j) qdf[j] = io->qd[j] + dt * io->qdd_free[j];
    real wwf[3], vowf[3];
    {
        real t[3], al[3], t2[3];
        m3v(w.Rw[0], a[0], t);
        for (int i = 0; i < 3; ++i) wwf[i] = ww[i] + dt * t[i];
        cross3(w.v[0], w.v[0] + 3, t2);
        al[0] = a[0][3] + t2[0]; al[1] = a[0][4] + t2[1]; al[2] = a[0][5] + t2[2];
        m3v(w.Rw[0], al, t);
        for (int i = 0; i < 3; ++i) vowf[i] = vow[i] + dt * (t[i] + g[i]);
    }

    /* ---- 7. sole-corner contacts ---- */
    real phi[DW_NUM_FOOT_PTS], rk[DW_NUM_FOOT_PTS][3];
    int active[DW_NUM_FOOT_PTS], any_active = 0;
    const int footb[2] = {m->foot_mv[0], m->foot_mv[4]};
    const int on_terrain = cfg->terrain && io->height_samples;
    real frame[DW_NUM_FOOT_PTS][9];      /* rows: t1, t2, n (world); identity on the plane */
    for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) {
        const int b = m->foot_mv[k];
        m3v(w.Rw[b], m->foot_pos[k], rk[k]);
        phi[k] = w.pw[b][2] + rk[k][2];
        for (int i = 0; i < 9; ++i) frame[k][i] = (i % 4 == 0) ? (real)1 : (real)0;
        if (on_terrain) {
            real hh;
            terrain_sample(cfg, io->height_samples, w.pw[b][0] + rk[k][0], w.pw[b][1] + rk[k][1], &hh, frame[k]);
            phi[k] = (phi[k] - hh) * frame[k][8];
        }
        active[k] = phi[k] < cfg->contact_offset;
        any_active |= active[k];
    }
    real P[DW_NUM_FOOT_PTS][3];
    for (int k = 0; k < DW_NUM_FOOT_PTS; ++k)
        for (int i = 0; i < 3; ++i) P[k][i] = active[k] ? io->warm[3 * k + i] : 0;

    real dqd_c[ND], dv0_c[6];
    for (int j = 0; j < ND; ++j) dqd_c[j] = 0;
    for (int i = 0; i < 6; ++i) dv0_c[i] = 0;

    if (any_active) {
        /* free twists of the two foot bodies, world aligned, by velocity FK along the legs */
        real vf[NB][6], tw_free[2][6];
        m3tv(w.Rw[0], wwf, vf[0]);
        m3tv(w.Rw[0], vowf, vf[0] + 3);
        for (int b = 1; b <= footb[1]; ++b) {
            const real *s = m->mv_axis[b];
            xform_motion(w.R[b], m->mv_pos[b], vf[m->mv_parent[b]], vf[b]);
            vf[b][0] += s[0] * qdf[b - 1]; vf[b][1] += s[1] * qdf[b - 1]; vf[b][2] += s[2] * qdf[b - 1];
        }
        for (int f = 0; f < 2; ++f) {
            m3v(w.Rw[footb[f]], vf[footb[f]], tw_free[f]);
            m3v(w.Rw[footb[f]], vf[footb[f]] + 3, tw_free[f] + 3);
        }
        /* 12x12 inverse operational inertia W: twist response of both feet to unit wrenches */
        real W[12][12];
        for (int f = 0; f < 2; ++f)
            for (int comp = 0; comp < 6; ++comp) {
                real dp[NB][6], dv[NB][6], dq[ND];
                memset(dp, 0, sizeof(dp));
                real ew[3] = {0, 0, 0}, eb[3];
                ew[comp % 3] = 1;
                m3tv(w.Rw[footb[f]], ew, eb);
                for (int i = 0; i < 3; ++i) dp[footb[f]][(comp < 3 ? 0 : 3) + i] = -eb[i];
                int lo = f == 0 ? 1 : footb[0] + 1;
                delta_aba(m, &w, dp, lo, footb[f], footb[1], dv, dq);
                for (int f2 = 0; f2 < 2; ++f2) {
                    real o[3];
                    m3v(w.Rw[footb[f2]], dv[footb[f2]], o);
                    for (int i = 0; i < 3; ++i) W[6 * f2 + i][6 * f + comp] = o[i];
                    m3v(w.Rw[footb[f2]], dv[footb[f2]] + 3, o);
                    for (int i = 0; i < 3; ++i) W[6 * f2 + 3 + i][6 * f + comp] = o[i];
                }
            }
        /* contact Jacobian rows: v_point = v_o + w x r  ->  J_k = [-[r]x, 1] on the foot's 6 twist comps */
        real J[24][12];
        memset(J, 0, sizeof(J));
        for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) {
            int f = k / 4;
            real S[9];
            skew3(rk[k], S);
            for (int r = 0; r < 3; ++r) {
                if (on_terrain) {      /* velocity along d at lever r: d.v_o + w.(r x d) */
                    const real *d = frame[k] + 3 * r;
                    real rxd[3];
                    cross3(rk[k], d, rxd);
                    for (int c = 0; c < 3; ++c) { J[3 * k + r][6 * f + c] = rxd[c]; J[3 * k + r][6 * f + 3 + c] = d[c]; }
                } else {
                    for (int c = 0; c < 3; ++c) J[3 * k + r][6 * f + c] = -S[3 * r + c];
                    J[3 * k + r][6 * f + 3 + r] = 1;
                }
            }
        }
        real JW[24][12], Amat[24][24], vel[24], vmin[DW_NUM_FOOT_PTS];
        for (int r = 0; r < 24; ++r)
            for (int c = 0; c < 12; ++c) {
                real sacc = 0;
                for (int k = 0; k < 12; ++k) sacc += J[r][k] * W[k][c];
                JW[r][c] = sacc;
            }
        for (int r = 0; r < 24; ++r)
            for (int c = 0; c < 24; ++c) {
                real sacc = 0;
                for (int k = 0; k < 12; ++k) sacc += JW[r][k] * J[c][k];
                Amat[r][c] = sacc;
            }
        for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) {
            int f = k / 4;
            real t[3];
            cross3(tw_free[f], rk[k], t);
            for (int i = 0; i < 3; ++i) vel[3 * k + i] = tw_free[f][3 + i] + t[i];
            if (on_terrain) {
                real vp[3] = {vel[3 * k], vel[3 * k + 1], vel[3 * k + 2]};
                for (int i = 0; i < 3; ++i) vel[3 * k + i] = dot3(frame[k] + 3 * i, vp);
            }
            if (phi[k] >= 0) vmin[k] = -phi[k] / dt;
            else {
                real vb = cfg->erp * (-phi[k]) / dt;
                vmin[k] = vb < cfg->max_depenetration_velocity ? vb : cfg->max_depenetration_velocity;
            }
        }
        /* warm start contribution */
        for (int r = 0; r < 24; ++r) {
            real sacc = 0;
            for (int k = 0; k < DW_NUM_FOOT_PTS; ++k)
                for (int i = 0; i < 3; ++i) sacc += Amat[r][3 * k + i] * P[k][i];
            vel[r] += sacc;
        }
        const real mu = io->mu;
        /* Projected Gauss-Seidel, block-Jacobi across the feet: corner kk of the left sole and corner kk of the right
         * sole are updated together from the same velocity snapshot (the feet only couple through the trunk), corners
         * of one sole sequentially.  Per contact: normal row, then the two friction rows with the normal's effect
         * folded in, then projection onto the Coulomb cone. */
        real invd[24];
        for (int r = 0; r < 24; ++r) invd[r] = 1 / (Amat[r][r] * (1 + cfg->contact_cfm));
        for (int it = 0; it < cfg->solver_iterations; ++it)
            for (int kk = 0; kk < 4; ++kk) {
                real d[2][3] = {{0, 0, 0}, {0, 0, 0}};       /* (Dx, Dy, dz) of the left / right contact */
                real pn[2][3];
                for (int f = 0; f < 2; ++f) {
                    const int k = kk + 4 * f;
                    if (!active[k]) continue;
                    const int rx = 3 * k, ry = 3 * k + 1, rz = 3 * k + 2;
                    real dz = -(vel[rz] - vmin[k]) * invd[rz];
                    real pz = P[k][2] + dz;
                    if (pz < 0) pz = 0;
                    dz = pz - P[k][2];
                    real vx = vel[rx] + Amat[rx][rz] * dz;
                    real dx = -vx * invd[rx];
                    real vy = vel[ry] + Amat[ry][rz] * dz + Amat[ry][rx] * dx;
                    real dy = -vy * invd[ry];
                    real px = P[k][0] + dx, py = P[k][1] + dy;
                    real lim = mu * pz, n2 = px * px + py * py;
                    if (n2 > lim * lim) {
                        real sc = lim / RSQRT(n2);
                        px *= sc; py *= sc;
                    }
                    d[f][0] = px - P[k][0]; d[f][1] = py - P[k][1]; d[f][2] = dz;
                    pn[f][0] = px; pn[f][1] = py; pn[f][2] = pz;
                }
                for (int r = 0; r < 24; ++r)
                    vel[r] = vel[r] + Amat[r][3 * kk + 2] * d[0][2] + Amat[r][3 * kk] * d[0][0] + Amat[r][3 * kk + 1] * d[0][1]
                                    + Amat[r][12 + 3 * kk + 2] * d[1][2] + Amat[r][12 + 3 * kk] * d[1][0] + Amat[r][12 + 3 * kk + 1] * d[1][1];
                for (int f = 0; f < 2; ++f)
                    if (active[kk + 4 * f]) for (int i = 0; i < 3; ++i) P[kk + 4 * f][i] = pn[f][i];
            }
        /* propagate the impulses through the whole tree */
        real dp[NB][6], dv[NB][6];
        memset(dp, 0, sizeof(dp));
        real Pw[DW_NUM_FOOT_PTS][3];      /* impulses in world axes */
        for (int k = 0; k < DW_NUM_FOOT_PTS; ++k)
            for (int i = 0; i < 3; ++i)
                Pw[k][i] = on_terrain ? P[k][0] * frame[k][i] + P[k][1] * frame[k][3 + i] + P[k][2] * frame[k][6 + i] : P[k][i];
        for (int f = 0; f < 2; ++f) {
            real F[3] = {0, 0, 0}, Nm[3] = {0, 0, 0};
            for (int k = 4 * f; k < 4 * f + 4; ++k) {
                real t[3];
                cross3(rk[k], Pw[k], t);
                for (int i = 0; i < 3; ++i) { F[i] += Pw[k][i]; Nm[i] += t[i]; }
            }
            real fb[3], nb[3];
            m3tv(w.Rw[footb[f]], F, fb);
            m3tv(w.Rw[footb[f]], Nm, nb);
            for (int i = 0; i < 3; ++i) { dp[footb[f]][i] = -nb[i]; dp[footb[f]][3 + i] = -fb[i]; }
        }
        delta_aba(m, &w, dp, 1, footb[1], NB - 1, dv, dqd_c);
        for (int i = 0; i < 6; ++i) dv0_c[i] = dv[0][i];
        for (int k = 0; k < DW_NUM_FOOT_PTS; ++k)
            for (int i = 0; i < 3; ++i) io->contact[3 * m->foot_gym[k] + i] += Pw[k][i] / dt;
    }
    for (int k = 0; k < DW_NUM_FOOT_PTS; ++k)
        for (int i = 0; i < 3; ++i) io->warm[3 * k + i] = P[k][i];

    /* ---- 8. final velocities, clamps, integration ---- */
    real wwn[3], vown[3], t[3];
    m3v(w.Rw[0], dv0_c, t);
    for (int i = 0; i < 3; ++i) wwn[i] = wwf[i] + t[i];
    m3v(w.Rw[0], dv0_c + 3, t);
    for (int i = 0; i < 3; ++i) vown[i] = vowf[i] + t[i];
    {
        real wn = RSQRT(dot3(wwn, wwn));
        if (wn > cfg->max_angular_velocity) {
            real sc = cfg->max_angular_velocity / wn;
            wwn[0] *= sc; wwn[1] *= sc; wwn[2] *= sc;
        }
    }
    for (int j = 0; j < ND; ++j) {
        real qd = qdf[j] + dqd_c[j];
        real vm = m->dof_vmax[j];
        if (qd > vm) qd = vm;
        if (qd < -vm) qd = -vm;
        real q = io->q[j] + dt * qd;
        if (q < m->dof_lower[j]) { q = m->dof_lower[j]; if (qd < 0) qd = 0; }
        if (q > m->dof_upper[j]) { q = m->dof_upper[j]; if (qd > 0) qd = 0; }
        io->q[j] = q; io->qd[j] = qd;
    }
    for (int i = 0; i < 3; ++i) io->root[i] += dt * vown[i];
    {
        real th = RSQRT(dot3(wwn, wwn)) * dt;
        real dq[4] = {0, 0, 0, 1};
        if (th > (real)1e-12) {
            real sh = RSIN(th / 2) / (th / dt);     /* sin(th/2)/|w| */
            dq[0] = wwn[0] * sh; dq[1] = wwn[1] * sh; dq[2] = wwn[2] * sh; dq[3] = RCOS(th / 2);
        }
        /* q_new = dq (x) q */
        real x1 = dq[0], y1 = dq[1], z1 = dq[2], w1 = dq[3];
        real x2 = quat[0], y2 = quat[1], z2 = quat[2], w2 = quat[3];
        real qn[4] = {w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                      w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                      w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                      w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2};
        real n = RSQRT(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
        for (int i = 0; i < 4; ++i) { qn[i] /= n; io->root[3 + i] = qn[i]; }
        if (cfg->root_vel_at_com) {
            real Rn[9], rc[3], tt[3];
            quat_to_mat(qn, Rn);
            m3v(Rn, c0, rc);
            cross3(wwn, rc, tt);
            for (int i = 0; i < 3; ++i) vown[i] += tt[i];
        }
    }
    for (int i = 0; i < 3; ++i) { io->root[7 + i] = vown[i]; io->root[10 + i] = wwn[i]; }
}
