// dw_devmodel.h -- the model and task constants as the kernels read them (one struct in device memory,
// ~16 KB, read-only, L2/K$ resident), and the host routine that derives the traversal tables from the
// C-ABI's DwModel (include/dyros_walk.h).
#pragma once

#include <stdint.h>
#include <string.h>

#include "../../include/dyros_walk.h"

namespace dw {

constexpr int NB = DW_NUM_MOVING;   // 34 moving bodies
constexpr int ND = DW_NUM_DOF;      // 33
constexpr int MAX_LEVELS = 12;      // tree depth 11 below the root
constexpr int MAX_PER_LEVEL = 5;    // widest level (legs + arms + neck)
constexpr int MAX_CHILD = 3;
constexpr int MAX_BODY_GEOMS = 8;
constexpr int MAX_BODY_INERT = 2;
constexpr int MAX_BODY_PAIRS = 12;
constexpr int MAX_CHAINS = 8;        // unbranched runs of the tree (legs, torso, arms, neck)
constexpr int MAX_CHAIN_LEN = 8;
constexpr int MAX_PHASES = 3;        // chains hanging off chains: base -> torso -> arms    // self-collision pairs one body takes part in

struct DevModel {
    // tree
    int32_t parent[NB];
    int32_t nchild[NB];
    int32_t child[NB][MAX_CHILD];
    int32_t nlevels;                          // levels 1..nlevels hold the bodies below the root
    int32_t level_count[MAX_LEVELS];
    int32_t level_body[MAX_LEVELS][MAX_PER_LEVEL];
    int32_t level_direct[MAX_LEVELS];         // 1: every body of the level is an only child (the inward sweep adds into the parent in place)
    int32_t level_slot[NB];                   // position of a body inside its level
    // unbranched chains: the outward sweeps walk one chain per lane, chains of one phase in parallel
    int32_t nchains, nphases;
    int32_t chain_len[MAX_CHAINS], chain_phase[MAX_CHAINS];
    int32_t chain_body[MAX_CHAINS][MAX_CHAIN_LEN];
    float   pos[NB][3];
    float   rot0[NB][9];
    float   axis[NB][3];
    float   pax[NB][3];                       // hinge axis in the parent's frame (rot0 * axis)
    // dofs
    float   qlo[ND], qhi[ND], vmax[ND];
    // inertial records
    int32_t ninert[NB];
    int32_t inert_idx[NB][MAX_BODY_INERT];
    int32_t inert_gym[DW_NUM_INERT];
    float   inert_mass[DW_NUM_INERT];
    float   inert_com[DW_NUM_INERT][3];
    float   inert_I[DW_NUM_INERT][6];
    // the same records laid out per body (slot k of body b) so that a lane reads them without first reading an index
    int32_t bi_gym[NB][MAX_BODY_INERT];
    float   bi_mass[NB][MAX_BODY_INERT];
    float   bi_com[NB][MAX_BODY_INERT][3];
    float   bi_I[NB][MAX_BODY_INERT][6];
    // collision primitives
    int32_t ngeom;
    DwGeom  geoms[DW_MAX_GEOMS];
    int32_t body_ngeom[NB];
    int32_t body_geom[NB][MAX_BODY_GEOMS];
    int32_t body_geom_gym[NB][MAX_BODY_GEOMS];    // Gym body that reports the primitive's contact force
    // soles
    int32_t foot_mv[DW_NUM_FOOT_PTS];
    int32_t foot_gym[DW_NUM_FOOT_PTS];
    float   foot_pos[DW_NUM_FOOT_PTS][3];
    int32_t foot_body[2];                     // moving body of the left / right sole
    int32_t left_foot_gym, right_foot_gym;
    // self-collision capsule proxies and pairs; per body the pairs it takes part in (entry = pair*2 + side)
    int32_t num_sc_pairs;
    DwCapsule sc_proxy[DW_MAX_SC_PROXIES];
    int32_t sc_pair[DW_MAX_SC_PAIRS][2];
    int32_t body_npair[NB];
    int32_t body_pair[NB][MAX_BODY_PAIRS];
    int32_t body_pair_gym[NB][MAX_BODY_PAIRS];
    struct ScPair { int32_t ba, bb; float a0[3], a1[3], b0[3], b1[3], ra, rb; } scp[DW_MAX_SC_PAIRS];   // pairs with their capsules inlined
    // task constants
    float   kp[ND], kv[ND], action_high[ND], q_init[ND];
    float   obs_mean[DW_NUM_OBS1], obs_inv_std_den[DW_NUM_OBS1];   // second = sqrt(var + 1e-8), the divisor
    float   arm_nom[ND], damp_nom[ND];
    int32_t has_task;
};

// Builds the traversal tables.  Returns 0 or a negative DW_E* code (message in err).
inline int build_devmodel(const DwModel *m, const DwTaskConst *t, DevModel *d, const char **err) {
    memset(d, 0, sizeof(*d));
    for (int b = 0; b < NB; ++b) {
        d->parent[b] = m->mv_parent[b];
        for (int i = 0; i < 3; ++i) { d->pos[b][i] = m->mv_pos[b][i]; d->axis[b][i] = m->mv_axis[b][i]; }
        for (int i = 0; i < 9; ++i) d->rot0[b][i] = m->mv_rot0[b][i];
        if (b > 0 && (m->mv_parent[b] < 0 || m->mv_parent[b] >= b)) { *err = "model: parent must precede child"; return DW_EINVAL; }
    }
    if (m->mv_parent[0] != -1) { *err = "model: body 0 must be the root"; return DW_EINVAL; }
    int depth[NB];
    depth[0] = 0;
    for (int b = 1; b < NB; ++b) {
        int p = m->mv_parent[b];
        depth[b] = depth[p] + 1;
        if (depth[b] >= MAX_LEVELS) { *err = "model: tree deeper than MAX_LEVELS"; return DW_EINVAL; }
        if (d->nchild[p] >= MAX_CHILD) { *err = "model: more than MAX_CHILD children"; return DW_EINVAL; }
        d->child[p][d->nchild[p]++] = b;
        int L = depth[b];
        if (d->level_count[L] >= MAX_PER_LEVEL) { *err = "model: level wider than MAX_PER_LEVEL"; return DW_EINVAL; }
        d->level_slot[b] = d->level_count[L];
        d->level_body[L][d->level_count[L]++] = b;
        if (L > d->nlevels) d->nlevels = L;
    }
    for (int L = 1; L <= d->nlevels; ++L) {
        d->level_direct[L] = 1;
        for (int k = 0; k < d->level_count[L]; ++k)
            if (d->nchild[d->parent[d->level_body[L][k]]] != 1) d->level_direct[L] = 0;
    }
    // chain decomposition: a chain starts at a child of the root or of a branching body and runs while bodies have one child
    {
        int chain_of[NB];
        chain_of[0] = -1;
        for (int b = 1; b < NB; ++b) {
            const int p = d->parent[b];
            if (p != 0 && d->nchild[p] == 1) {
                const int c = chain_of[p];
                if (d->chain_len[c] >= MAX_CHAIN_LEN) { *err = "model: chain longer than MAX_CHAIN_LEN"; return DW_EINVAL; }
                d->chain_body[c][d->chain_len[c]++] = b;
                chain_of[b] = c;
            } else {
                if (d->nchains >= MAX_CHAINS) { *err = "model: more than MAX_CHAINS chains"; return DW_EINVAL; }
                const int c = d->nchains++;
                d->chain_phase[c] = (p == 0) ? 0 : d->chain_phase[chain_of[p]] + 1;
                if (d->chain_phase[c] >= MAX_PHASES) { *err = "model: chains nested deeper than MAX_PHASES"; return DW_EINVAL; }
                if (d->chain_phase[c] + 1 > d->nphases) d->nphases = d->chain_phase[c] + 1;
                d->chain_body[c][0] = b;
                d->chain_len[c] = 1;
                chain_of[b] = c;
            }
        }
    }
    for (int b = 1; b < NB; ++b)
        for (int i = 0; i < 3; ++i)
            d->pax[b][i] = d->rot0[b][3 * i] * d->axis[b][0] + d->rot0[b][3 * i + 1] * d->axis[b][1] + d->rot0[b][3 * i + 2] * d->axis[b][2];
    for (int j = 0; j < ND; ++j) { d->qlo[j] = m->dof_lower[j]; d->qhi[j] = m->dof_upper[j]; d->vmax[j] = m->dof_vmax[j]; }
    for (int k = 0; k < DW_NUM_INERT; ++k) {
        int b = m->inert_mv[k];
        if (b < 0 || b >= NB || m->inert_gym[k] < 0 || m->inert_gym[k] >= DW_NUM_BODIES) { *err = "model: inertial index out of range"; return DW_EINVAL; }
        if (d->ninert[b] >= MAX_BODY_INERT) { *err = "model: too many inertial records on one body"; return DW_EINVAL; }
        {
            const int sl = d->ninert[b];
            d->bi_gym[b][sl] = m->inert_gym[k];
            d->bi_mass[b][sl] = m->inert_mass[k];
            for (int i = 0; i < 3; ++i) d->bi_com[b][sl][i] = m->inert_com[k][i];
            for (int i = 0; i < 6; ++i) d->bi_I[b][sl][i] = m->inert_I[k][i];
        }
        d->inert_idx[b][d->ninert[b]++] = k;
        d->inert_gym[k] = m->inert_gym[k];
        d->inert_mass[k] = m->inert_mass[k];
        for (int i = 0; i < 3; ++i) d->inert_com[k][i] = m->inert_com[k][i];
        for (int i = 0; i < 6; ++i) d->inert_I[k][i] = m->inert_I[k][i];
    }
    if (m->inert_mv[0] != 0) { *err = "model: inertial record 0 must belong to the root"; return DW_EINVAL; }
    if (m->num_geoms < 0 || m->num_geoms > 64) { *err = "model: num_geoms out of range (kernel maps one lane per primitive)"; return DW_EINVAL; }
    d->ngeom = m->num_geoms;
    for (int g = 0; g < m->num_geoms; ++g) {
        d->geoms[g] = m->geoms[g];
        int b = m->geoms[g].moving;
        if (b < 0 || b >= NB || m->geoms[g].gym < 0 || m->geoms[g].gym >= DW_NUM_BODIES) { *err = "model: geom index out of range"; return DW_EINVAL; }
        if (m->geoms[g].sole) continue;
        if (d->body_ngeom[b] >= MAX_BODY_GEOMS) { *err = "model: too many primitives on one body"; return DW_EINVAL; }
        d->body_geom_gym[b][d->body_ngeom[b]] = m->geoms[g].gym;
        d->body_geom[b][d->body_ngeom[b]++] = g;
    }
    for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) {
        d->foot_mv[k] = m->foot_mv[k];
        d->foot_gym[k] = m->foot_gym[k];
        for (int i = 0; i < 3; ++i) d->foot_pos[k][i] = m->foot_pos[k][i];
    }
    d->foot_body[0] = m->foot_mv[0];
    d->foot_body[1] = m->foot_mv[4];
    // the contact pipeline walks each leg as the chain 6f+1 .. 6f+6 hanging off the root
    for (int f = 0; f < 2; ++f) {
        if (d->foot_body[f] != 6 * f + 6) { *err = "model: sole bodies must be moving bodies 6 and 12"; return DW_EINVAL; }
        for (int k = 0; k < 4; ++k) if (m->foot_mv[4 * f + k] != d->foot_body[f]) { *err = "model: sole corners must be grouped 4+4"; return DW_EINVAL; }
        for (int i = 1; i <= 6; ++i) if (m->mv_parent[6 * f + i] != (i == 1 ? 0 : 6 * f + i - 1)) { *err = "model: legs must be serial chains off the root"; return DW_EINVAL; }
    }
    d->left_foot_gym = m->left_foot_gym;
    d->right_foot_gym = m->right_foot_gym;
    if (m->num_sc_proxies < 0 || m->num_sc_proxies > DW_MAX_SC_PROXIES || m->num_sc_pairs < 0 || m->num_sc_pairs > DW_MAX_SC_PAIRS) {
        *err = "model: self-collision tables out of range"; return DW_EINVAL;
    }
    d->num_sc_pairs = m->num_sc_pairs;
    for (int k = 0; k < m->num_sc_proxies; ++k) {
        d->sc_proxy[k] = m->sc_proxy[k];
        if (m->sc_proxy[k].moving < 0 || m->sc_proxy[k].moving >= NB || m->sc_proxy[k].gym < 0 || m->sc_proxy[k].gym >= DW_NUM_BODIES) {
            *err = "model: self-collision proxy index out of range"; return DW_EINVAL;
        }
    }
    for (int k = 0; k < m->num_sc_pairs; ++k)
        for (int side = 0; side < 2; ++side) {
            const int pr = m->sc_pair[k][side];
            if (pr < 0 || pr >= m->num_sc_proxies) { *err = "model: self-collision pair index out of range"; return DW_EINVAL; }
            d->sc_pair[k][side] = pr;
            const int b = m->sc_proxy[pr].moving;
            if (d->body_npair[b] >= MAX_BODY_PAIRS) { *err = "model: too many self-collision pairs on one body"; return DW_EINVAL; }
            d->body_pair_gym[b][d->body_npair[b]] = m->sc_proxy[pr].gym;
            d->body_pair[b][d->body_npair[b]++] = 2 * k + side;
        }
    for (int k = 0; k < m->num_sc_pairs; ++k) {
        const DwCapsule &ca = m->sc_proxy[m->sc_pair[k][0]], &cb = m->sc_proxy[m->sc_pair[k][1]];
        DevModel::ScPair &q = d->scp[k];
        q.ba = ca.moving; q.bb = cb.moving; q.ra = ca.radius; q.rb = cb.radius;
        for (int i = 0; i < 3; ++i) { q.a0[i] = ca.p0[i]; q.a1[i] = ca.p1[i]; q.b0[i] = cb.p0[i]; q.b1[i] = cb.p1[i]; }
    }
    if (t) {
        for (int j = 0; j < ND; ++j) {
            d->kp[j] = t->kp[j]; d->kv[j] = t->kv[j]; d->action_high[j] = t->action_high[j];
            d->q_init[j] = t->initial_dof_pos[j];
            d->arm_nom[j] = t->dof_armature_nominal[j]; d->damp_nom[j] = t->dof_damping_nominal[j];
        }
        for (int i = 0; i < DW_NUM_OBS1; ++i) {
            d->obs_mean[i] = t->obs_mean[i];
            // torch: sqrt(obs_var + 1e-8*ones)  (reference: tasks/dyros_dynamic_walk.py:777); host sqrtf is
            // correctly rounded, as is the device's
            float v = t->obs_var[i] + 1e-8f * 1.0f;
            d->obs_inv_std_den[i] = __builtin_sqrtf(v);
        }
        d->has_task = 1;
    }
    return DW_OK;
}

}  // namespace dw
