// dw_physics.h -- what the substep kernels of every layout share: the physics parameters (PhysParams), the symmetric 6x6
// storage, small vector / matrix helpers, Newton-refined reciprocal and reciprocal square root, the spatial cross product and
// the height-field sample.  The substep itself (the stand-in for the reference's closed `gym.simulate`, call site
// tasks/dyros_dynamic_walk.py:525; algorithm = DESIGN.md "Physics model", CPU restatement oracle/dw_physics.c) lives in the
// layout headers: dw_oct.h (8 lanes per env, the default) and dw_lane.h (one lane per env).
//
// Coordinates used by all of them: every spatial quantity of the substep (twists [w; v_O], wrenches [n_O; f], joint subspaces,
// rigid and articulated inertias) is expressed in ONE frame -- world-aligned axes, reference point O = the base origin at the
// start of the substep.  With a common frame the tree recursions need no link-to-link transforms: a child's articulated inertia
// is *added* to its parent's, accelerations and impulses pass down a chain as a' = a_parent + c, and a unit wrench on a foot
// climbs its leg with one dot product and one axpy per joint.  (The oracle states the same algorithm in link coordinates; the two
// agree to rounding -- all links stay within ~1 m of O, so moving the reference point costs about two digits of the 1e-7.)
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
    // coarse upper bound of the height field around a base position (terrain_bound below); built at bind, nullptr = no bound
    const int16_t *hmax;
    int   hm_cell, hm_rows, hm_cols;
};

// Packed index of entry (r,c) of a symmetric 6x6 (upper triangle, row-major): 21 words instead of 36.
DW_HD constexpr int sym6(int r, int c) {
    return r <= c ? (r * (13 - r)) / 2 + (c - r) : (c * (13 - c)) / 2 + (r - c);
}

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


// The coarse bound table.  hmax[ci][cj] = the largest height sample within `reach` metres (plus the bilinear patch's extra
// sample) of ANY point of cell (ci, cj), a cell being hm_cell x hm_cell samples.  `reach` = model_reach(): no point of any
// collision primitive of the robot is farther from the base origin than that, whatever the joint angles (the sum of the link
// offsets along the chain to the primitive's body plus the primitive's farthest point from that body's origin -- TOCABI: 1.43 m).
// A robot whose base origin lies in the cell therefore has every point of every body within the window, so a body whose origin is
// higher above terrain_bound() than its bounding radius cannot touch the height field, exactly as `z > radius` says so over the
// plane -- the kernels skip its primitives' height-field fetches and contact frames with no change in any result (the oracle,
// which samples under every primitive, is the checker).
constexpr float HM_CELL = 0.5f, HM_MARGIN = 0.05f;
inline float model_reach(const DevModel &M) {
    float chain[NB], best = 0.0f;
    chain[0] = 0.0f;
    for (int b = 1; b < NB; ++b) chain[b] = chain[M.parent[b]] + sqrtf(M.pos[b][0] * M.pos[b][0] + M.pos[b][1] * M.pos[b][1] + M.pos[b][2] * M.pos[b][2]);
    for (int g = 0; g < M.ngeom; ++g) {
        const DwGeom &ge = M.geoms[g];
        const float off = sqrtf(ge.pos[0] * ge.pos[0] + ge.pos[1] * ge.pos[1] + ge.pos[2] * ge.pos[2]);
        const float ext = ge.type == 0 ? sqrtf(ge.size[0] * ge.size[0] + ge.size[1] * ge.size[1] + ge.size[2] * ge.size[2])
                                       : sqrtf(ge.size[0] * ge.size[0] + ge.size[1] * ge.size[1]);
        const float r = chain[ge.moving] + off + ext;
        best = r > best ? r : best;
    }
    return best + HM_MARGIN;
}
inline int hm_cell_samples(float hscale) { int c = (int)(HM_CELL / hscale); return c < 1 ? 1 : c; }
inline int hm_reach_samples(float hscale, float reach) { return (int)(reach / hscale) + 2; }
DW_HD int16_t terrain_bound_cell(const int16_t *hs, int rows, int cols, int cell, int reach, int ci, int cj) {
    int i0 = ci * cell - reach, i1 = ci * cell + cell - 1 + reach, j0 = cj * cell - reach, j1 = cj * cell + cell - 1 + reach;
    i0 = i0 < 0 ? 0 : i0; j0 = j0 < 0 ? 0 : j0;
    i1 = i1 > rows - 1 ? rows - 1 : i1; j1 = j1 > cols - 1 ? cols - 1 : j1;
    int16_t m = hs[(size_t)i0 * cols + j0];
    for (int i = i0; i <= i1; ++i)
        for (int j = j0; j <= j1; ++j) { const int16_t v = hs[(size_t)i * cols + j]; m = v > m ? v : m; }
    return m;
}
// the height no point of the field within reach of a robot based at world (x, y) exceeds (same index arithmetic as terrain_sample)
DW_HD float terrain_bound(const PhysParams &P, float x, float y) {
#if defined(DW_NO_TERRAIN_BOUND)          // (A/B builds only: every body with primitives samples the field, as before round 4)
    return 3.0e38f;
#endif
    if (!P.hmax) return 3.0e38f;
    float u = (x + P.t_border) * P.t_inv_h, v = (y + P.t_border) * P.t_inv_h;
    const float umax = (float)(P.t_rows - 1) - 1e-3f, vmax = (float)(P.t_cols - 1) - 1e-3f;
    u = u < 0.0f ? 0.0f : (u > umax ? umax : u);
    v = v < 0.0f ? 0.0f : (v > vmax ? vmax : v);
    const int ci = (int)u / P.hm_cell, cj = (int)v / P.hm_cell;
    return P.t_vs * (float)P.hmax[(size_t)ci * P.hm_cols + cj];
}

}  // namespace dw
