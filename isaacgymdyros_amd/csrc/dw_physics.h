// dw_physics.h -- one physics substep (the stand-in for the reference's closed `gym.simulate`,
// reference call site tasks/dyros_dynamic_walk.py:525) as wave regions over the env's LDS block.
//
// Algorithm = DESIGN.md "Physics model": floating-base Featherstone ABA (armature + implicit joint
// damping on the diagonal), penalty ground forces for non-sole primitives, velocity-level projected
// Gauss-Seidel on the 8 sole corners over a Delassus matrix assembled from 12 unit-wrench responses of
// the two foot bodies, impulse propagation through the tree, semi-implicit Euler.
//
// Coordinates: every spatial quantity of the substep (twists [w; v_O], wrenches [n_O; f], joint subspaces,
// rigid and articulated inertias) is expressed in ONE frame -- world-aligned axes, reference point O = the base
// origin at the start of the substep.  With a common frame the tree recursions need no link-to-link transforms:
// a child's articulated inertia is *added* to its parent's, accelerations and impulses pass down a chain as
// a' = a_parent + c, and a unit wrench on a foot climbs its leg with one dot product and one axpy per joint.
// (The oracle states the same algorithm in link coordinates; the two agree to rounding -- all links stay within
// ~1 m of O, so moving the reference point costs about two digits of the 1e-7, see DESIGN.md.)
//
// Lane maps: lane = body / dof / primitive for the flat phases; (body-in-level, column) = (lane/3, lane%3) for
// forward kinematics; (body-in-level, row, half) = 12 lanes per body for the inward articulated-inertia sweep;
// lane = body-in-level for the outward sweeps; lane = Delassus column for the 12 response sweeps; lane =
// constraint row for the 24-row Gauss-Seidel.
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
    // height field (row f-4); hs == nullptr on the ground plane
    const int16_t *hs;
    // octet kernels: where a lane parks the self-collision wrenches of its proxies between the resolution and the inward pass
    // (rare path; [waves][64][DW_SC_PARK_WORDS], allocated at dw_create)
    float *sc_park;
    int   t_rows, t_cols;
    float t_inv_h, t_vs, t_border;
};

// The kinematic tree, staged once per launch into the env's LDS block: every sweep region indexes these by lane
// or by level, and a dependent chain of global loads (level -> body -> parent -> offset/axis) per region is what
// the first profile of this kernel was made of.
struct LdsTree {
    float pos[NB][3];
    float pax[NB][3];
    unsigned char parent[NB];
    unsigned char nchild[NB];
    unsigned char child[NB][MAX_CHILD];
    unsigned char level_slot[NB];
    unsigned char level_count[MAX_LEVELS];
    unsigned char level_direct[MAX_LEVELS];
    unsigned char level_body[MAX_LEVELS][MAX_PER_LEVEL];
    unsigned char nlevels;
    unsigned char nchains, nphases;
    unsigned char chain_len[MAX_CHAINS], chain_phase[MAX_CHAINS];
    unsigned char chain_body[MAX_CHAINS][MAX_CHAIN_LEN];
};

// Packed index of entry (r,c) of a symmetric 6x6 (upper triangle, row-major): 21 words instead of 36.
DW_HD constexpr int sym6(int r, int c) {
    return r <= c ? (r * (13 - r)) / 2 + (c - r) : (c * (13 - c)) / 2 + (r - c);
}

// One env's LDS block.  160 KB per CU / 13.3 KB = 12 resident envs (3 waves per SIMD); the first version of this
// struct was 18.6 KB (8 envs).  The saving comes from overlaying arrays whose lifetimes inside a substep do not
// intersect (phases in order: K1 K2 kinematics, K4 K5 primitives and self-collision, K3 inertias, SW inward sweep,
// A3 base solve, A4 outward sweep, V1 free velocities, C1..C5 contact, V2 integrate) and from packing the
// symmetric articulated inertias.
// Rows that regions read whole are 8- or 16-byte aligned (and padded: Rw 9 -> 12, IA 21 -> 24 words) so that the
// reads become ds_read_b64 / ds_read_b128: the CU's one LDS pipeline, shared by its twelve resident waves, is what
// the sweeps saturate, and a b128 read moves four words for the price of two b32 reads.
constexpr int RW_STRIDE = 12, IA_STRIDE = 24;

struct alignas(16) Lds {
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
    alignas(16) float RwK[3][RW_STRIDE];
    float pwK[3][3];     // world rotation / position relative to O of the base and the two sole bodies, K5 .. V1
    alignas(16) float Sj[NB][6];                // joint motion subspaces [a_w; r x a_w], K2 .. C5
    union alignas(16) {             // block B
        struct { float Rw[NB][RW_STRIDE]; float pr[NB][3]; } kin;                       // K2 .. K3 (pr = position relative to O)
        struct { float T[MAX_PER_LEVEL][IA_STRIDE]; float pa[MAX_PER_LEVEL][6]; } sw;  // SW (levels whose parents gather)
        struct {                                                                // A3 .. V2
            float a[NB][6];
            float Minv[36];
            float du[NB];
            float qdd[ND], qdf[ND], dqd[ND];
            float wwf[3], vowf[3], dv0[6];
        } post;
    } B;
    union alignas(16) {             // block V
        struct { float v[NB][6]; float pA[NB][6]; } dyn;                        // K2 .. A4 (v), K5 .. A3 (pA)
        struct {                                                                // V1 .. C5
            float W[12][12];
            float vel[2][24], P[2][24];
            float rk[8][3], phi[8], vmin[8];                                    //   rk = sole corner relative to O
            int   active[8];
            int   any_active;
            float twf[2][6];
            float frame[8][9];                                                  //   contact frames t1, t2, n (height field only)
        } con;
    } V;
    struct alignas(16) {            // block C
        struct { float U[NB][6], Dinv[NB], u[NB]; } art;                        // SW .. C5
    } C;
    union alignas(16) {             // block A
        float R[NB][9];                                                         // K1 .. K2: body -> parent rotations
        struct {                                                                // K4 .. K5 (before the inertias are built)
            float gF[64][3], gr[64][3];                                         //   ground penalty: world force, point relative to O
            float pF[DW_MAX_SC_PAIRS][3], pa[DW_MAX_SC_PAIRS][3], pb[DW_MAX_SC_PAIRS][3];   // self-collision: force on A, points on A / B
        } geo;
        float IA[NB][IA_STRIDE];                                                    // K3 .. A3 (packed, sym6)
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
// 1/sqrt(x) for x > 0: hardware estimate refined by one Newton step on the device (~1 ulp), exact on the host
DW_HD float rsqrt_nr(float x) {
#if defined(__HIPCC__)
    float y = __builtin_amdgcn_rsqf(x);
    return y * (1.5f - 0.5f * x * y * y);
#else
    return 1.0f / sqrtf(x);
#endif
}
// 1/x for x > 0: hardware estimate refined by one Newton step on the device (~1 ulp), exact on the host
DW_HD float rcp_nr(float x) {
#if defined(__HIPCC__)
    const float y = __builtin_amdgcn_rcpf(x);
    return y * (2.0f - x * y);
#else
    return 1.0f / x;
#endif
}
// spatial motion cross product c = v x m  (v, m = [angular; linear])
DW_HD void motion_cross(const float *v, const float *m, float *c) {
    float t[3];
    cross3(v, m, c);
    cross3(v, m + 3, c + 3);
    cross3(v + 3, m, t);
    c[3] += t[0]; c[4] += t[1]; c[5] += t[2];
}
DW_HD float dot6(const float *a, const float *b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

// Height field under a world point (row f-4): bilinear height of the four samples around it, unit normal from that
// patch's gradient, contact frame (t1, t2, n) with t1 = world x projected into the tangent plane (oracle/dw_physics.c
// terrain_sample; grid convention of the reference, tasks/dyros_dynamic_walk.py:240-253).
DW_HD void terrain_sample(const PhysParams &P, float x, float y, float *h, float *fr) {
    float u = (x + P.t_border) * P.t_inv_h, v = (y + P.t_border) * P.t_inv_h;
    const float umax = (float)(P.t_rows - 1) - 1e-3f, vmax = (float)(P.t_cols - 1) - 1e-3f;
    u = u < 0.0f ? 0.0f : (u > umax ? umax : u);
    v = v < 0.0f ? 0.0f : (v > vmax ? vmax : v);
    const int i = (int)u, j = (int)v;
    const float a = u - (float)i, b = v - (float)j;
    const int16_t *p = P.hs + (size_t)i * P.t_cols + j;
    const float h00 = p[0], h01 = p[1], h10 = p[P.t_cols], h11 = p[P.t_cols + 1];
    *h = P.t_vs * ((1.0f - a) * ((1.0f - b) * h00 + b * h01) + a * ((1.0f - b) * h10 + b * h11));
    const float gx = P.t_vs * P.t_inv_h * ((1.0f - b) * (h10 - h00) + b * (h11 - h01));
    const float gy = P.t_vs * P.t_inv_h * ((1.0f - a) * (h01 - h00) + a * (h11 - h10));
    const float nn = rsqrt_nr(gx * gx + gy * gy + 1.0f);
    float *t1 = fr, *t2 = fr + 3, *n = fr + 6;
    n[0] = -gx * nn; n[1] = -gy * nn; n[2] = nn;
    const float a1[3] = {1.0f - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    const float tn = rsqrt_nr(dot3(a1, a1));
    t1[0] = a1[0] * tn; t1[1] = a1[1] * tn; t1[2] = a1[2] * tn;
    cross3(n, t1, t2);
}

// ------------------------------------------------------------------------------------------------
// the substep.  In: S.root, q, qd, tau, arm, damp, mscale, mu, push, warm.  Out: root, q, qd, warm, contact.
// ------------------------------------------------------------------------------------------------
// copies the tree tables from the device-resident model into LDS (one region)
DW_HD void stage_tree_lane(int l, Lds &S, const DevModel &M) {
    {
        if (l < NB) {
            S.tree.parent[l] = M.parent[l];
            S.tree.nchild[l] = M.nchild[l];
            for (int i = 0; i < MAX_CHILD; ++i) S.tree.child[l][i] = M.child[l][i];
            for (int i = 0; i < 3; ++i) { S.tree.pos[l][i] = M.pos[l][i]; S.tree.pax[l][i] = M.pax[l][i]; }
            S.tree.level_slot[l] = M.level_slot[l];
        }
        if (l < MAX_LEVELS) { S.tree.level_count[l] = M.level_count[l]; S.tree.level_direct[l] = M.level_direct[l]; }
        if (l < MAX_LEVELS * MAX_PER_LEVEL) S.tree.level_body[l / MAX_PER_LEVEL][l % MAX_PER_LEVEL] = M.level_body[l / MAX_PER_LEVEL][l % MAX_PER_LEVEL];
        if (l == 63) { S.tree.nlevels = M.nlevels; S.tree.nchains = M.nchains; S.tree.nphases = M.nphases; }
        if (l < MAX_CHAINS) { S.tree.chain_len[l] = M.chain_len[l]; S.tree.chain_phase[l] = M.chain_phase[l]; }
        if (l < MAX_CHAINS * MAX_CHAIN_LEN) S.tree.chain_body[l / MAX_CHAIN_LEN][l % MAX_CHAIN_LEN] = M.chain_body[l / MAX_CHAIN_LEN][l % MAX_CHAIN_LEN];
    }
}
template <class W>
DW_HD void stage_tree(const W &wave, Lds &S, const DevModel &M) {
    wave.par([&](int l) { stage_tree_lane(l, S, M); });
}

// Profiling builds (-DDW_PROFILE_STOP=n) leave the substep after phase n; results are then meaningless, only the
// launch time is read (tools/phase_costs.sh).  Never defined in the shipped library.
#if defined(DW_PROFILE_STOP)
#define DW_CKPT(n) do { if (DW_PROFILE_STOP == (n)) return; } while (0)
#else
#define DW_CKPT(n) do { } while (0)
#endif

// Wave-uniform loop bounds of the tree sweeps, read once per launch through the scalar cache and kept in SGPRs (as LDS
// bytes each loop test is an LDS round trip plus a v_readfirstlane in front of every region).
struct TreeUniform {
    int nlevels, nphases, nchains;
    unsigned direct_mask;              // bit L: level L adds into its parents in place
    unsigned long long counts;         // 4 bits per level: bodies in the level
};
DW_HD TreeUniform make_tree_uniform(const DevModel &M) {
    TreeUniform t;
    t.nlevels = M.nlevels; t.nphases = M.nphases; t.nchains = M.nchains;
    t.direct_mask = 0; t.counts = 0;
    for (int L = 0; L < MAX_LEVELS; ++L) {
        t.direct_mask |= (unsigned)(M.level_direct[L] != 0) << L;
        t.counts |= (unsigned long long)(M.level_count[L] & 15) << (4 * L);
    }
    return t;
}

// TERRAIN = false is the ground plane z = 0 (the benchmark path); true samples the height field under every contact.
template <bool TERRAIN, class W>
DW_HD void physics_substep(const W &wave, Lds &S, const DevModel &M, const PhysParams &P, const TreeUniform &TU) {
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
            for (int i = 0; i < 9; ++i) S.B.kin.Rw[0][i] = Rw[i];
            for (int i = 0; i < 3; ++i) S.B.kin.pr[0][i] = 0.0f;
            float ww[3] = {S.root[10], S.root[11], S.root[12]};
            float vo[3] = {S.root[7], S.root[8], S.root[9]};
            if (P.vel_at_com) {
                float rc[3], t[3];
                m3v(Rw, M.inert_com[0], rc);
                cross3(ww, rc, t);
                vo[0] -= t[0]; vo[1] -= t[1]; vo[2] -= t[2];
            }
            for (int i = 0; i < 3; ++i) { S.ww[i] = ww[i]; S.vow[i] = vo[i]; S.V.dyn.v[0][i] = ww[i]; S.V.dyn.v[0][3 + i] = vo[i]; }
            for (int i = 0; i < 6; ++i) S.Sj[0][i] = 0.0f;
        } else if (l < NB) {
            const int b = l;
            const float *s = M.axis[b];
            float q = S.q[b - 1];
            float sn, cs;
            sincosf(q, &sn, &cs);
            const float oc = 1.0f - cs;
            float Rj[9] = {cs + oc * s[0] * s[0], oc * s[0] * s[1] - sn * s[2], oc * s[0] * s[2] + sn * s[1],
                           oc * s[1] * s[0] + sn * s[2], cs + oc * s[1] * s[1], oc * s[1] * s[2] - sn * s[0],
                           oc * s[2] * s[0] - sn * s[1], oc * s[2] * s[1] + sn * s[0], cs + oc * s[2] * s[2]};
            float R[9];
            m3m(M.rot0[b], Rj, R);
            for (int i = 0; i < 9; ++i) S.A.R[b][i] = R[i];
        }
    });

    DW_CKPT(1);
    // ---- K2: forward kinematics, joint subspaces and velocities, level by level; three lanes per body
    //      (lane = body-in-level, column c).  S = [a_w; r x a_w] with a_w = R_parent * (axis in the parent frame) ----
    for (int L = 1; L <= TU.nlevels; ++L) {
        const int cnt = (int)((TU.counts >> (4 * L)) & 15);
        wave.par([&](int l) {
            const int k = l / 3, c = l - 3 * k;
            if (k < cnt) {
                const int b = S.tree.level_body[L][k], p = S.tree.parent[b];
                // every input into registers before the first write: the child's rows live in the same arrays as the
                // parent's, so a store between two loads would order them (and cost an LDS round trip each)
                float Rp[9], prp[3], pos[3], pax[3];
                for (int i = 0; i < 9; ++i) Rp[i] = S.B.kin.Rw[p][i];
                for (int i = 0; i < 3; ++i) { prp[i] = S.B.kin.pr[p][i]; pos[i] = S.tree.pos[b][i]; pax[i] = S.tree.pax[b][i]; }
                const float *Rb = S.A.R[b];
                const float r0 = Rb[c], r1 = Rb[3 + c], r2 = Rb[6 + c];           // column c of R
                const float vpa = S.V.dyn.v[p][c], vpl = S.V.dyn.v[p][3 + c];
                const float qd = S.qd[b - 1];
                float aw[3], x[3];
                m3v(Rp, pax, aw);
                m3v(Rp, pos, x);
                x[0] += prp[0]; x[1] += prp[1]; x[2] += prp[2];
                const float sx = x[1] * aw[2] - x[2] * aw[1], sy = x[2] * aw[0] - x[0] * aw[2], sz = x[0] * aw[1] - x[1] * aw[0];
                const float ac = c == 0 ? aw[0] : (c == 1 ? aw[1] : aw[2]);
                const float sc = c == 0 ? sx : (c == 1 ? sy : sz);
                const float xc = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
                for (int i = 0; i < 3; ++i) S.B.kin.Rw[b][3 * i + c] = Rp[3 * i] * r0 + Rp[3 * i + 1] * r1 + Rp[3 * i + 2] * r2;
                S.B.kin.pr[b][c] = xc;
                S.Sj[b][c] = ac;
                S.Sj[b][3 + c] = sc;
                S.V.dyn.v[b][c] = vpa + ac * qd;
                S.V.dyn.v[b][3 + c] = vpl + sc * qd;
            }
        });
    }

    DW_CKPT(2);
    // ---- K4: penalty contact of the non-sole primitives against the ground (one lane per primitive) ----
    wave.par([&](int l) {
        float F[3] = {0, 0, 0}, xr[3] = {0, 0, 0};
        if (l < M.ngeom && !M.geoms[l].sole) {
            const DwGeom &ge = M.geoms[l];
            const int b = ge.moving;
            float Rw[9];
            for (int i = 0; i < 9; ++i) Rw[i] = S.B.kin.Rw[b][i];
            float rl[3];
            if (ge.type == 0) {
                // deepest corner: along each box axis take the end that points down (world z component of the axis)
                float Rg[9], e[3];
                m3m(Rw, ge.rot, Rg);
                for (int i = 0; i < 3; ++i) e[i] = (Rg[6 + i] > 0.0f ? -1.0f : 1.0f) * ge.size[i];
                m3v(ge.rot, e, rl);
                rl[0] += ge.pos[0]; rl[1] += ge.pos[1]; rl[2] += ge.pos[2];
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
            }
            float wv[3];
            m3v(Rw, rl, wv);
            for (int i = 0; i < 3; ++i) xr[i] = S.B.kin.pr[b][i] + wv[i];
            const float zmin = S.root[2] + xr[2];
            if (TERRAIN) {
                float hh, fr[9];
                terrain_sample(P, S.root[0] + xr[0], S.root[1] + xr[1], &hh, fr);
                const float *nrm = fr + 6;
                const float dist = (zmin - hh) * nrm[2];
                if (dist < 0) {
                    float t[3], vw[3];
                    cross3(S.V.dyn.v[b], xr, t);
                    for (int i = 0; i < 3; ++i) vw[i] = S.V.dyn.v[b][3 + i] + t[i];
                    const float vn = dot3(vw, nrm);
                    float fn = P.pen_k * (-dist) - P.pen_c * vn;
                    if (fn < 0) fn = 0;
                    const float vt[3] = {vw[0] - vn * nrm[0], vw[1] - vn * nrm[1], vw[2] - vn * nrm[2]};
                    const float sp = sqrtf(dot3(vt, vt));
                    for (int i = 0; i < 3; ++i) F[i] = fn * nrm[i];
                    if (sp > 1e-9f) {
                        float ft = P.pen_c * sp, lim = S.mu * fn;
                        if (ft > lim) ft = lim;
                        for (int i = 0; i < 3; ++i) F[i] -= ft * vt[i] / sp;
                    }
                }
            } else if (zmin < 0) {
                float t[3], vw[3];
                cross3(S.V.dyn.v[b], xr, t);
                for (int i = 0; i < 3; ++i) vw[i] = S.V.dyn.v[b][3 + i] + t[i];
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
        for (int i = 0; i < 3; ++i) { S.A.geo.gF[l][i] = F[i]; S.A.geo.gr[l][i] = xr[i]; }
    });
    // ---- K4b: self-collision, one lane per capsule pair: closest points of the two segments, penalty force along the
    //      normal when the capsules overlap (force on A; B gets the opposite).  All points relative to O. ----
    wave.par([&](int l) {
        if (l < DW_MAX_SC_PAIRS) for (int i = 0; i < 3; ++i) S.A.geo.pF[l][i] = 0.0f;
        if (P.self_collision && l < M.num_sc_pairs) {
            int pi = l;
            DW_OPAQUE(pi);
            const DevModel::ScPair &sp = M.scp[pi];
            const int ba = sp.ba, bb = sp.bb;
            float Ra[9], Rb[9], a0[3], a1[3], b0[3], b1[3], t3[3];
            for (int i = 0; i < 9; ++i) { Ra[i] = S.B.kin.Rw[ba][i]; Rb[i] = S.B.kin.Rw[bb][i]; }
            m3v(Ra, sp.a0, t3); for (int i = 0; i < 3; ++i) a0[i] = S.B.kin.pr[ba][i] + t3[i];
            m3v(Ra, sp.a1, t3); for (int i = 0; i < 3; ++i) a1[i] = S.B.kin.pr[ba][i] + t3[i];
            m3v(Rb, sp.b0, t3); for (int i = 0; i < 3; ++i) b0[i] = S.B.kin.pr[bb][i] + t3[i];
            m3v(Rb, sp.b1, t3); for (int i = 0; i < 3; ++i) b1[i] = S.B.kin.pr[bb][i] + t3[i];
            const float da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
            // (a pair whose segment midpoints are further apart than half lengths + radii cannot touch; when that holds
            //  for all sixteen lanes the closest-point code below is skipped by the wave)
            const float mid[3] = {0.5f * (a0[0] + a1[0] - b0[0] - b1[0]), 0.5f * (a0[1] + a1[1] - b0[1] - b1[1]), 0.5f * (a0[2] + a1[2] - b0[2] - b1[2])};
            const float reach = 0.5f * (sqrtf(dot3(da, da)) + sqrtf(dot3(db, db))) + sp.ra + sp.rb;
            if (dot3(mid, mid) <= reach * reach) {
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
                        // (blended with the mid-overlap answer for nearly parallel capsules: oracle/dw_physics.c seg_seg)
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
                float pa[3], pb[3], n[3];
                for (int i = 0; i < 3; ++i) { pa[i] = a0[i] + sa * da[i]; pb[i] = b0[i] + sb * db[i]; n[i] = pa[i] - pb[i]; }
                const float dist = sqrtf(dot3(n, n));
                const float depth = sp.ra + sp.rb - dist;
                if (depth > 0.0f && dist > 1e-6f) {
                    for (int i = 0; i < 3; ++i) n[i] /= dist;
                    float ta[3], tb[3];
                    cross3(S.V.dyn.v[ba], pa, ta);
                    cross3(S.V.dyn.v[bb], pb, tb);
                    float vn = 0.0f;
                    for (int i = 0; i < 3; ++i) vn += ((S.V.dyn.v[ba][3 + i] + ta[i]) - (S.V.dyn.v[bb][3 + i] + tb[i])) * n[i];
                    float fn = P.pen_k * depth - P.pen_c * vn;
                    if (fn < 0.0f) fn = 0.0f;
                    for (int i = 0; i < 3; ++i) { S.A.geo.pF[l][i] = fn * n[i]; S.A.geo.pa[l][i] = pa[i]; S.A.geo.pb[l][i] = pb[i]; }
                }
            }
        }
    });
    // K5: external forces into the bias of their bodies (wrench about O: [x x F; F]); per-body net contact force
    wave.par([&](int l) {
        if (l < NB) {
            int b = l;
            DW_OPAQUE(b);
            float dn[3] = {0, 0, 0}, df[3] = {0, 0, 0};
            for (int k = 0; k < M.body_ngeom[b]; ++k) {
                const int g = M.body_geom[b][k];
                float F[3] = {S.A.geo.gF[g][0], S.A.geo.gF[g][1], S.A.geo.gF[g][2]};
                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                    float nb[3];
                    cross3(S.A.geo.gr[g], F, nb);
                    for (int i = 0; i < 3; ++i) { dn[i] += nb[i]; df[i] += F[i]; }
                    const int gy = M.body_geom_gym[b][k];
                    for (int i = 0; i < 3; ++i) S.contact[3 * gy + i] += F[i];
                }
            }
            for (int k = 0; k < M.body_npair[b]; ++k) {         // self-collision pairs this body takes part in
                const int code = M.body_pair[b][k], pr = code >> 1, side = code & 1;
                const float sg = side ? -1.0f : 1.0f;
                float F[3] = {sg * S.A.geo.pF[pr][0], sg * S.A.geo.pF[pr][1], sg * S.A.geo.pF[pr][2]};
                if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                    float nb[3];
                    cross3(side ? S.A.geo.pb[pr] : S.A.geo.pa[pr], F, nb);
                    for (int i = 0; i < 3; ++i) { dn[i] += nb[i]; df[i] += F[i]; }
                    const int gy = M.body_pair_gym[b][k];
                    for (int i = 0; i < 3; ++i) S.contact[3 * gy + i] += F[i];
                }
            }
            if (b == 0) {
                float Fw[3] = {S.push[0], S.push[1], 0.0f}, xc[3], nb[3];
                m3v(S.B.kin.Rw[0], M.inert_com[0], xc);
                cross3(xc, Fw, nb);
                for (int i = 0; i < 3; ++i) { dn[i] += nb[i]; df[i] += Fw[i]; }
            }
            const bool keep = (b == 0 || b == 6 || b == 12);     // block B is recycled by the sweep: keep what the contact phases need
            float t[12];                             // (all reads, then all writes: an LDS-to-LDS copy written element by
            if (keep) {                              //  element waits for each read before its write)
                for (int i = 0; i < 9; ++i) t[i] = S.B.kin.Rw[b][i];
                for (int i = 0; i < 3; ++i) t[9 + i] = S.B.kin.pr[b][i];
            }
            for (int i = 0; i < 3; ++i) { S.V.dyn.pA[b][i] = -dn[i]; S.V.dyn.pA[b][3 + i] = -df[i]; }   // K3 adds the gyroscopic part
            if (keep) {
                const int slot = b / 6;
                for (int i = 0; i < 9; ++i) S.RwK[slot][i] = t[i];
                for (int i = 0; i < 3; ++i) S.pwK[slot][i] = t[9 + i];
            }
        }
    });

    // ---- K3: rigid-body inertias about O in world axes (packed into block A, whose primitive scratch is dead now),
    //      gyroscopic bias.  Link-frame moments (A about the link origin, first moment h, mass) as before, then
    //      A_O = R A R' + m(|r|^2 1 - r r') + 2 (r.hy) 1 - (r hy' + hy r'),  h_O = hy + m r,  hy = R h ----
    wave.par([&](int l) {
        if (l < NB) {
            int b = l;
            DW_OPAQUE(b);
            float A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, h[3] = {0, 0, 0}, mass = 0.0f;
            for (int k = 0; k < M.ninert[b]; ++k) {
                const float ms = S.mscale[M.bi_gym[b][k]];
                const float mk = ms * M.bi_mass[b][k];
                const float *cm = M.bi_com[b][k], *I6 = M.bi_I[b][k];
                const float cc = dot3(cm, cm);
                const float Ic[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
                for (int r3 = 0; r3 < 3; ++r3)
                    for (int c3 = 0; c3 < 3; ++c3)
                        A[3 * r3 + c3] += ms * Ic[3 * r3 + c3] + mk * ((r3 == c3 ? cc : 0.0f) - cm[r3] * cm[c3]);
                h[0] += mk * cm[0]; h[1] += mk * cm[1]; h[2] += mk * cm[2];
                mass += mk;
            }
            float Rw[9], T[9], hy[3], x[3], vin[6], pin[6];
            for (int i = 0; i < 9; ++i) Rw[i] = S.B.kin.Rw[b][i];
            for (int i = 0; i < 3; ++i) x[i] = S.B.kin.pr[b][i];
            for (int i = 0; i < 6; ++i) { vin[i] = S.V.dyn.v[b][i]; pin[i] = S.V.dyn.pA[b][i]; }      // read before the stores below
            m3m(Rw, A, T);
            m3v(Rw, h, hy);
            const float xx = dot3(x, x), xh = dot3(x, hy);
            float Ao[9];
            for (int r3 = 0; r3 < 3; ++r3)
                for (int c3 = r3; c3 < 3; ++c3) {
                    float v = T[3 * r3] * Rw[3 * c3] + T[3 * r3 + 1] * Rw[3 * c3 + 1] + T[3 * r3 + 2] * Rw[3 * c3 + 2];
                    v += (r3 == c3 ? mass * xx + 2.0f * xh : 0.0f) - mass * x[r3] * x[c3] - (x[r3] * hy[c3] + hy[r3] * x[c3]);
                    Ao[3 * r3 + c3] = v; Ao[3 * c3 + r3] = v;
                }
            const float ho[3] = {hy[0] + mass * x[0], hy[1] + mass * x[1], hy[2] + mass * x[2]};
            // 6x6 = [[A_O, H],[H', m 1]] with H = skew(h_O), stored packed (upper triangle)
            float *I = S.A.IA[b];
            const float H[9] = {0, -ho[2], ho[1], ho[2], 0, -ho[0], -ho[1], ho[0], 0};
            for (int r3 = 0; r3 < 3; ++r3)
                for (int c3 = 0; c3 < 3; ++c3) {
                    if (c3 >= r3) {
                        I[sym6(r3, c3)] = Ao[3 * r3 + c3];
                        I[sym6(r3 + 3, c3 + 3)] = (r3 == c3) ? mass : 0.0f;
                    }
                    I[sym6(r3, 3 + c3)] = H[3 * r3 + c3];
                }
            // pA += v x* (I v)
            float om[3] = {vin[0], vin[1], vin[2]}, vl[3] = {vin[3], vin[4], vin[5]};
            float n[3], f[3], t1[3], t2[3];
            m3v(Ao, om, n); cross3(ho, vl, t1);
            n[0] += t1[0]; n[1] += t1[1]; n[2] += t1[2];
            cross3(om, ho, t1);                      // H' w = -h x w = w x h
            f[0] = t1[0] + mass * vl[0]; f[1] = t1[1] + mass * vl[1]; f[2] = t1[2] + mass * vl[2];
            cross3(om, n, t1); cross3(vl, f, t2);
            S.V.dyn.pA[b][0] = pin[0] + (t1[0] + t2[0]); S.V.dyn.pA[b][1] = pin[1] + (t1[1] + t2[1]); S.V.dyn.pA[b][2] = pin[2] + (t1[2] + t2[2]);
            cross3(om, f, t1);
            S.V.dyn.pA[b][3] = pin[3] + t1[0]; S.V.dyn.pA[b][4] = pin[4] + t1[1]; S.V.dyn.pA[b][5] = pin[5] + t1[2];
        }
    });
    DW_CKPT(3);
    // ---- SW: inward sweep of articulated inertias.  Twelve lanes per body: lane = (body-in-level k, row r, half h),
    //      each lane owns the three columns 3h..3h+2 of row r (5 bodies x 12 = 60 lanes on the widest level).
    //      Ia = IA - U U'/D and pa = pA + Ia c + U u/D go to the parent as they are (common frame): added in place
    //      when every body of the level is an only child, through a per-level buffer and a gather region otherwise. ----
    for (int L = TU.nlevels; L >= 1; --L) {
        const int cnt = (int)((TU.counts >> (4 * L)) & 15);
        const bool direct = (TU.direct_mask >> L) & 1;
        wave.par([&](int l) {
            const int k = l / 12, r = (l % 12) >> 1, h = l & 1;
            if (k < cnt) {
                const int b = S.tree.level_body[L][k], p = S.tree.parent[b];
                const float *IA = S.A.IA[b];
                float s[6], U[6];
                for (int j = 0; j < 6; ++j) s[j] = S.Sj[b][j];
                for (int j = 0; j < 6; ++j) {
                    float acc = 0.0f;
                    for (int c = 0; c < 6; ++c) acc += IA[sym6(j, c)] * s[c];
                    U[j] = acc;
                }
                const float damp = S.damp[b - 1], qd = S.qd[b - 1];
                const float D = dot6(s, U) + S.arm[b - 1] + dt * damp;
                const float Dinv = rcp_nr(D);
                const float u = S.tau[b - 1] - damp * qd - dot6(s, S.V.dyn.pA[b]);
                const float ur = S.A.IA[b][sym6(r, 0)] * s[0] + S.A.IA[b][sym6(r, 1)] * s[1] + S.A.IA[b][sym6(r, 2)] * s[2] +
                                 S.A.IA[b][sym6(r, 3)] * s[3] + S.A.IA[b][sym6(r, 4)] * s[4] + S.A.IA[b][sym6(r, 5)] * s[5];   // U[r]
                const float urd = ur * Dinv;
                float Ia[6];
                for (int c = 0; c < 6; ++c) Ia[c] = IA[sym6(r, c)] - urd * U[c];
                // (every LDS input is read before the first store of the region: a store with a run-time index orders all
                //  later loads behind it, and each such load then costs its own round trip)
                float *dst = direct ? S.A.IA[p] : S.B.sw.T[k];
                float old[3] = {0.0f, 0.0f, 0.0f}, vb[6];
                if (direct) for (int c = 0; c < 3; ++c) old[c] = dst[sym6(r, 3 * h + c)];
                for (int j = 0; j < 6; ++j) vb[j] = S.V.dyn.v[b][j];
                const float pAr = S.V.dyn.pA[b][r], pAp = direct ? S.V.dyn.pA[p][r] : 0.0f;
                for (int c = 0; c < 3; ++c) {
                    const int cc = 3 * h + c;
                    if (cc >= r) dst[sym6(r, cc)] = old[c] + (h ? Ia[3 + c] : Ia[c]);
                }
                if (h == 0) {
                    float m[6], cb[6];
                    for (int j = 0; j < 6; ++j) m[j] = s[j] * qd;
                    motion_cross(vb, m, cb);
                    float pa = pAr + ur * (u * Dinv);
                    for (int c = 0; c < 6; ++c) pa += Ia[c] * cb[c];
                    if (direct) S.V.dyn.pA[p][r] = pAp + pa; else S.B.sw.pa[k][r] = pa;
                    if (r == 0) {
                        for (int j = 0; j < 6; ++j) S.C.art.U[b][j] = U[j];
                        S.C.art.Dinv[b] = Dinv;
                        S.C.art.u[b] = u;
                    }
                }
            }
        });
        if (!direct) {
            // parents (one level up) gather their children from the level buffer in child order
            const int pcnt = (L == 1) ? 1 : (int)((TU.counts >> (4 * (L - 1))) & 15);
            wave.par([&](int l) {
                const int k = l / 12, r = (l % 12) >> 1, h = l & 1;
                if (k < pcnt) {
                    const int p = (L == 1) ? 0 : S.tree.level_body[L - 1][k];
                    const int nch = S.tree.nchild[p];
                    float acc[3], accp = S.V.dyn.pA[p][r];
                    for (int c = 0; c < 3; ++c) acc[c] = S.A.IA[p][sym6(r, 3 * h + c)];
                    for (int i = 0; i < MAX_CHILD; ++i) {
                        if (i < nch) {
                            const int kc = S.tree.level_slot[S.tree.child[p][i]];
                            for (int c = 0; c < 3; ++c) acc[c] += S.B.sw.T[kc][sym6(r, 3 * h + c)];
                            accp += S.B.sw.pa[kc][r];
                        }
                    }
                    for (int c = 0; c < 3; ++c)
                        if (3 * h + c >= r) S.A.IA[p][sym6(r, 3 * h + c)] = acc[c];
                    if (h == 0) S.V.dyn.pA[p][r] = accp;
                }
            });
        }
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
    // from here on block B holds the "post" arrays
    wave.par([&](int l) {
        if (l < 6) {
            float acc = 0.0f;
            for (int c = 0; c < 6; ++c) acc -= S.B.post.Minv[6 * l + c] * S.V.dyn.pA[0][c];
            S.B.post.a[0][l] = acc;
        }
        if (l >= 8 && l < 8 + NB) S.B.post.du[l - 8] = 0.0f;
    });

    DW_CKPT(5);
    // ---- A4: outward sweep of accelerations.  One lane per unbranched chain walks it with the parent acceleration in
    //      registers (no region boundary between the bodies of a chain); chains of one phase run side by side:
    //      a' = a_parent + v x S qd,  qdd = (u - U'a') / D,  a = a' + S qdd ----
    for (int ph = 0; ph < TU.nphases; ++ph) {
        wave.par([&](int l) {
            if (l < TU.nchains && S.tree.chain_phase[l] == ph) {
                const int n = S.tree.chain_len[l];
                const int p0 = S.tree.parent[S.tree.chain_body[l][0]];
                float a[6];
                for (int j = 0; j < 6; ++j) a[j] = S.B.post.a[p0][j];
                for (int i = 0; i < n; ++i) {
                    const int b = S.tree.chain_body[l][i];
                    const float qd = S.qd[b - 1];
                    float s[6], m[6], c[6];
                    for (int j = 0; j < 6; ++j) { s[j] = S.Sj[b][j]; m[j] = s[j] * qd; }
                    motion_cross(S.V.dyn.v[b], m, c);
                    for (int j = 0; j < 6; ++j) a[j] += c[j];
                    const float qdd = (S.C.art.u[b] - dot6(S.C.art.U[b], a)) * S.C.art.Dinv[b];
                    S.B.post.qdd[b - 1] = qdd;
                    for (int j = 0; j < 6; ++j) a[j] += s[j] * qdd;
                    if (i == n - 1 && S.tree.nchild[b] > 0) for (int j = 0; j < 6; ++j) S.B.post.a[b][j] = a[j];     // only a branching body's acceleration is read again
                }
            }
        });
    }

    DW_CKPT(6);
    // ---- V1: unconstrained velocities; sole-corner gaps ----
    wave.par([&](int l) {
        if (l < ND) S.B.post.qdf[l] = S.qd[l] + dt * S.B.post.qdd[l];
        if (l == 40) {
            // classical acceleration of the base origin = spatial linear part + w x v; gravity enters as a uniform field
            float t2[3];
            cross3(S.ww, S.vow, t2);
            for (int i = 0; i < 3; ++i) {
                S.B.post.wwf[i] = S.ww[i] + dt * S.B.post.a[0][i];
                S.B.post.vowf[i] = S.vow[i] + dt * (S.B.post.a[0][3 + i] + t2[i] + P.g[i]);
            }
        }
        if (l >= 48 && l < 48 + DW_NUM_FOOT_PTS) {
            const int k = l - 48;
            float r[3];
            m3v(S.RwK[1 + k / 4], M.foot_pos[k], r);
            for (int i = 0; i < 3; ++i) r[i] += S.pwK[1 + k / 4][i];
            float phi = S.root[2] + r[2];
            if (TERRAIN) {
                float hh, fr[9];
                terrain_sample(P, S.root[0] + r[0], S.root[1] + r[1], &hh, fr);
                phi = (phi - hh) * fr[8];
                for (int i = 0; i < 9; ++i) S.V.con.frame[k][i] = fr[i];
            }
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
        // C1: free twists of the two foot bodies = base twist + sum over the leg of S qd
        if (l >= 32 && l < 44) {
            const int f = (l - 32) / 6, j = (l - 32) % 6;
            float acc = j < 3 ? S.B.post.wwf[j] : S.B.post.vowf[j - 3];
            for (int i = 1; i <= 6; ++i) acc += S.Sj[6 * f + i][j] * S.B.post.qdf[6 * f + i - 1];
            S.V.con.twf[f][j] = acc;
        }
    });

    if (uniform(S.V.con.any_active)) {
        DW_CKPT(7);
        // ---- C2: 12 unit-wrench responses -> inverse operational inertia W of the two feet (twists and wrenches about O).
        //      Up the leg: d = -S'p, p += U d/D; base: dv = -Minv p; down both legs: qdd = (d - U'dv)/D, dv += S qdd ----
        wave.par([&](int l) {
            if (l < 12) {
                const int f = l / 6, comp = l % 6;
                float dp[6], dc[6];
                for (int j = 0; j < 6; ++j) dp[j] = (j == comp) ? -1.0f : 0.0f;
#pragma unroll
                for (int i = 6; i >= 1; --i) {
                    const int b = 6 * f + i;
                    const float d = -dot6(S.Sj[b], dp);
                    dc[i - 1] = d;
                    const float k = d * S.C.art.Dinv[b];
                    for (int j = 0; j < 6; ++j) dp[j] += S.C.art.U[b][j] * k;
                }
                float dv0[6];
                for (int r = 0; r < 6; ++r) {
                    float acc = 0.0f;
                    for (int c = 0; c < 6; ++c) acc -= S.B.post.Minv[6 * r + c] * dp[c];
                    dv0[r] = acc;
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float dv[6];
                    for (int j = 0; j < 6; ++j) dv[j] = dv0[j];
#pragma unroll
                    for (int i = 1; i <= 6; ++i) {
                        const int b = 6 * g + i;
                        const float ua = dot6(S.C.art.U[b], dv);
                        const float qdd = ((g == f ? dc[i - 1] : 0.0f) - ua) * S.C.art.Dinv[b];
                        for (int j = 0; j < 6; ++j) dv[j] += S.Sj[b][j] * qdd;
                    }
                    for (int j = 0; j < 6; ++j) S.V.con.W[6 * g + j][l] = dv[j];
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
                float dd[3] = {ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f};
                if (TERRAIN) {      // row direction d = frame vector ax of the corner: J = [r x d, d]
                    for (int i = 0; i < 3; ++i) dd[i] = S.V.con.frame[k][3 * ax + i];
                    cross3(r, dd, ja);
                }
                float twv[6];
                for (int i = 0; i < 6; ++i) twv[i] = S.V.con.twf[f][i];
                float JW0[6], JW1[6];
                for (int c = 0; c < 6; ++c) {
                    if (TERRAIN) {
                        JW0[c] = ja[0] * S.V.con.W[6 * f][c] + ja[1] * S.V.con.W[6 * f + 1][c] + ja[2] * S.V.con.W[6 * f + 2][c] +
                                 dd[0] * S.V.con.W[6 * f + 3][c] + dd[1] * S.V.con.W[6 * f + 4][c] + dd[2] * S.V.con.W[6 * f + 5][c];
                        JW1[c] = ja[0] * S.V.con.W[6 * f][6 + c] + ja[1] * S.V.con.W[6 * f + 1][6 + c] + ja[2] * S.V.con.W[6 * f + 2][6 + c] +
                                 dd[0] * S.V.con.W[6 * f + 3][6 + c] + dd[1] * S.V.con.W[6 * f + 4][6 + c] + dd[2] * S.V.con.W[6 * f + 5][6 + c];
                    } else {
                        JW0[c] = ja[0] * S.V.con.W[6 * f][c] + ja[1] * S.V.con.W[6 * f + 1][c] + ja[2] * S.V.con.W[6 * f + 2][c] +
                                 S.V.con.W[6 * f + 3 + ax][c];
                        JW1[c] = ja[0] * S.V.con.W[6 * f][6 + c] + ja[1] * S.V.con.W[6 * f + 1][6 + c] + ja[2] * S.V.con.W[6 * f + 2][6 + c] +
                                 S.V.con.W[6 * f + 3 + ax][6 + c];
                    }
                }
                // the corners of one sole are read together before that half of the row is stored (2 round trips, not 8)
                for (int half = 0; half < 2; ++half) {
                    float rh[4][3];
                    for (int k2 = 0; k2 < 4; ++k2) for (int i = 0; i < 3; ++i) rh[k2][i] = S.V.con.rk[4 * half + k2][i];
                    const float *JW = half ? JW1 : JW0;
                    float fh[4][9];
                    if (TERRAIN) for (int k2 = 0; k2 < 4; ++k2) for (int i = 0; i < 9; ++i) fh[k2][i] = S.V.con.frame[4 * half + k2][i];
                    for (int k2 = 0; k2 < 4; ++k2) {
                        const float *r2 = rh[k2];
                        // column (k2, ax2): sum_j JW[j] * (-skew(r2)[ax2][j]) + JW[3 + ax2]; in world axes the three columns
                        // of a corner are the vector JW_lin + JW_ang x r2, on terrain projected on the corner's frame
                        const float c0 = JW[1] * r2[2] - JW[2] * r2[1] + JW[3];
                        const float c1 = -JW[0] * r2[2] + JW[2] * r2[0] + JW[4];
                        const float c2 = JW[0] * r2[1] - JW[1] * r2[0] + JW[5];
                        if (TERRAIN) {
                            for (int a2 = 0; a2 < 3; ++a2)
                                S.A.lcp.A[l][12 * half + 3 * k2 + a2] = fh[k2][3 * a2] * c0 + fh[k2][3 * a2 + 1] * c1 + fh[k2][3 * a2 + 2] * c2;
                        } else {
                            S.A.lcp.A[l][12 * half + 3 * k2 + 0] = c0;
                            S.A.lcp.A[l][12 * half + 3 * k2 + 1] = c1;
                            S.A.lcp.A[l][12 * half + 3 * k2 + 2] = c2;
                        }
                    }
                }
                float t[3];
                cross3(twv, r, t);
                if (TERRAIN) S.V.con.vel[1][l] = dd[0] * (twv[3] + t[0]) + dd[1] * (twv[4] + t[1]) + dd[2] * (twv[5] + t[2]);
                else S.V.con.vel[1][l] = (ax == 0 ? twv[3] : (ax == 1 ? twv[4] : twv[5])) + (ax == 0 ? t[0] : (ax == 1 ? t[1] : t[2]));
            }
        });
        wave.par([&](int l) {
            if (l < 24) {
                float acc = S.V.con.vel[1][l];
                for (int c = 0; c < 24; ++c) acc += S.A.lcp.A[l][c] * S.V.con.P[0][c];
                S.V.con.vel[0][l] = acc;
                // (0 for the rows of an inactive corner: its Gauss-Seidel update is then the identity)
                S.A.lcp.invd[l] = S.V.con.active[l / 3] ? 1.0f / (S.A.lcp.A[l][l] * (1.0f + P.cfm)) : 0.0f;
            }
        });
        DW_CKPT(9);
        // ---- C4: projected Gauss-Seidel, block-Jacobi across the feet.  Corner kk of the left sole (rows 3kk..) and
        //      corner kk of the right sole (rows 12+3kk..) are updated together from the same velocity snapshot --
        //      the feet only couple through the trunk -- and the four corners of a sole sequentially: 4 updates per
        //      sweep instead of 8.  Per contact: normal row, friction rows with the normal's effect folded in,
        //      projection onto the Coulomb cone.
        //      The rows of the left sole live in lanes 0..11, those of the right sole in lanes 32..43, each with its row
        //      of A (24 registers), velocity and impulse in registers.  Every lane runs the impulse arithmetic of ITS
        //      half's corner, so the two corners of a pair are solved by one instruction stream; the scalars a corner needs
        //      come from its three row lanes by ds_swizzle broadcasts inside the 32-lane half (cross-lane only, no LDS
        //      memory), the impulse changes cross halves with v_readlane.  Rows of inactive corners have invd = 0 (C3),
        //      which makes their update the identity without a branch.  (The host suite runs this very code with one
        //      fiber per lane: Wave::simt.)
        const int cur = 0;
        {
            int act[DW_NUM_FOOT_PTS];
            for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) act[k] = uniform(S.V.con.active[k]);
            wave.simt([&](int l) {
            const int half = l >> 5, lj = l & 31;
            const int row = lj < 12 ? 12 * half + lj : 0;
            float Arow[24];
            for (int c = 0; c < 24; ++c) Arow[c] = S.A.lcp.A[row][c];
            float vel = S.V.con.vel[0][row], Pl = S.V.con.P[0][row];
            const float invd = S.A.lcp.invd[row];
            const float mu = S.mu;
            // (The swizzle pattern is an immediate, so the four pairs are written out by macro rather than by an unrolled
            // loop.  The iteration-invariant scalars of a corner -- diagonal inverses, in-corner couplings -- are broadcast
            // again in every sweep: keeping the 24 of them in registers across the solver costs more in spills than the
            // six extra cross-lane operations per update.)
            float vminr[4];
            for (int kk = 0; kk < 4; ++kk) vminr[kk] = S.V.con.vmin[4 * half + kk];
#define DW_PGS_PAIR(KK) if (act[KK] | act[KK + 4]) { \
                    const float cz = half ? Arow[12 + 3 * KK + 2] : Arow[3 * KK + 2];     /* column z of the own corner */ \
                    const float cx = half ? Arow[12 + 3 * KK] : Arow[3 * KK]; \
                    const float izk = half_bcast<3 * KK + 2>(invd), ixk = half_bcast<3 * KK>(invd), iyk = half_bcast<3 * KK + 1>(invd); \
                    const float azxk = half_bcast<3 * KK>(cz), azyk = half_bcast<3 * KK + 1>(cz), axyk = half_bcast<3 * KK + 1>(cx); \
                    const float Pz = half_bcast<3 * KK + 2>(Pl), Px = half_bcast<3 * KK>(Pl), Py = half_bcast<3 * KK + 1>(Pl); \
                    const float vz = half_bcast<3 * KK + 2>(vel), vx0 = half_bcast<3 * KK>(vel), vy0 = half_bcast<3 * KK + 1>(vel); \
                    float dz = -(vz - vminr[KK]) * izk; \
                    float pz = Pz + dz; \
                    if (pz < 0) pz = 0; \
                    dz = pz - Pz; \
                    const float vx = vx0 + azxk * dz; \
                    const float dx = -vx * ixk; \
                    const float vy = vy0 + azyk * dz + axyk * dx; \
                    const float dy = -vy * iyk; \
                    float px = Px + dx, py = Py + dy; \
                    const float lim = mu * pz, n2 = px * px + py * py; \
                    if (n2 > lim * lim) { \
                        const float sc = lim * rsqrt_nr(n2); \
                        px *= sc; py *= sc; \
                    } \
                    const float Dx = px - Px, Dy = py - Py; \
                    const float L0 = lane_bcast(dz, 0), L1 = lane_bcast(Dx, 0), L2 = lane_bcast(Dy, 0); \
                    const float R0 = lane_bcast(dz, 32), R1 = lane_bcast(Dx, 32), R2 = lane_bcast(Dy, 32); \
                    vel = vel + Arow[3 * KK + 2] * L0 + Arow[3 * KK] * L1 + Arow[3 * KK + 1] * L2 \
                              + Arow[12 + 3 * KK + 2] * R0 + Arow[12 + 3 * KK] * R1 + Arow[12 + 3 * KK + 1] * R2; \
                    Pl = lj == 3 * KK ? px : (lj == 3 * KK + 1 ? py : (lj == 3 * KK + 2 ? pz : Pl)); }
            for (int it = 0; it < P.iters; ++it) {
                DW_PGS_PAIR(0) DW_PGS_PAIR(1) DW_PGS_PAIR(2) DW_PGS_PAIR(3)
            }
#undef DW_PGS_PAIR
            if (lj < 12) S.V.con.P[0][row] = Pl;
            });
        }
        DW_CKPT(10);
        // ---- C5: impulses -> wrenches about O on the two foot bodies -> delta-ABA over the whole tree ----
        wave.par([&](int l) {
            if (l < 2) {
                const float *Pc = S.V.con.P[cur];
                float F[3] = {0, 0, 0}, Nm[3] = {0, 0, 0};
                for (int k = 4 * l; k < 4 * l + 4; ++k) {
                    float t[3], pw[3] = {Pc[3 * k], Pc[3 * k + 1], Pc[3 * k + 2]};
                    if (TERRAIN) {      // impulse components are along the corner's frame: back to world axes
                        const float *fr = S.V.con.frame[k];
                        const float p0 = pw[0], p1 = pw[1], p2 = pw[2];
                        for (int i = 0; i < 3; ++i) pw[i] = p0 * fr[i] + p1 * fr[3 + i] + p2 * fr[6 + i];
                    }
                    cross3(S.V.con.rk[k], pw, t);
                    for (int i = 0; i < 3; ++i) { F[i] += pw[i]; Nm[i] += t[i]; }
                }
                float dp[6];
                for (int i = 0; i < 3; ++i) { dp[i] = -Nm[i]; dp[3 + i] = -F[i]; }
                float dus[6];
#pragma unroll
                for (int i = 6; i >= 1; --i) {
                    const int b = 6 * l + i;
                    const float d = -dot6(S.Sj[b], dp);
                    dus[i - 1] = d;
                    const float kk = d * S.C.art.Dinv[b];
                    for (int j = 0; j < 6; ++j) dp[j] += S.C.art.U[b][j] * kk;
                }
#pragma unroll
                for (int i = 1; i <= 6; ++i) S.B.post.du[6 * l + i] = dus[i - 1];
                for (int j = 0; j < 6; ++j) S.A.lcp.dpf[l][j] = dp[j];
                const int gy = M.foot_gym[4 * l];
                const float c0 = S.contact[3 * gy], c1 = S.contact[3 * gy + 1], c2 = S.contact[3 * gy + 2];
                S.contact[3 * gy] = c0 + F[0] / dt; S.contact[3 * gy + 1] = c1 + F[1] / dt; S.contact[3 * gy + 2] = c2 + F[2] / dt;
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
        for (int ph = 0; ph < TU.nphases; ++ph) {     // velocity jumps down the tree, one lane per chain as in A4
            wave.par([&](int l) {
                if (l < TU.nchains && S.tree.chain_phase[l] == ph) {
                    const int n = S.tree.chain_len[l];
                    const int p0 = S.tree.parent[S.tree.chain_body[l][0]];
                    float a[6];
                    for (int j = 0; j < 6; ++j) a[j] = S.B.post.a[p0][j];
                    for (int i = 0; i < n; ++i) {
                        const int b = S.tree.chain_body[l][i];
                        const float dq = (S.B.post.du[b] - dot6(S.C.art.U[b], a)) * S.C.art.Dinv[b];
                        S.B.post.dqd[b - 1] = dq;
                        for (int j = 0; j < 6; ++j) a[j] += S.Sj[b][j] * dq;
                        if (i == n - 1 && S.tree.nchild[b] > 0) for (int j = 0; j < 6; ++j) S.B.post.a[b][j] = a[j];
                    }
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
            float wwn[3], von[3];
            for (int i = 0; i < 3; ++i) { wwn[i] = S.B.post.wwf[i] + S.B.post.dv0[i]; von[i] = S.B.post.vowf[i] + S.B.post.dv0[3 + i]; }
            const float qin[4] = {S.quat[0], S.quat[1], S.quat[2], S.quat[3]};      // read before the position is stored
            {
                const float wn2 = dot3(wwn, wwn);
                if (wn2 > P.max_ang_vel * P.max_ang_vel) {
                    const float sc = P.max_ang_vel * rsqrt_nr(wn2);
                    wwn[0] *= sc; wwn[1] *= sc; wwn[2] *= sc;
                }
            }
            {
                const float p0 = S.root[0], p1 = S.root[1], p2 = S.root[2];
                S.root[0] = p0 + dt * von[0]; S.root[1] = p1 + dt * von[1]; S.root[2] = p2 + dt * von[2];
            }
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
            const float x2q = qin[0], y2 = qin[1], z2 = qin[2], w2q = qin[3];
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
