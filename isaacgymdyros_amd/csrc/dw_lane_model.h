// dw_lane_model.h -- which wave walks which bodies in the lane kernels (dw_lane.h), and the few per-body constants that
// dw_devmodel.h does not already hold in the form those kernels read them.  Everything here is indexed by wave-uniform
// values (the wave's role, the body it is at), so the kernels fetch it with SCALAR loads: a constant costs no vector
// register and no LDS.
//
// Roles (TOCABI; derived from the tree and checked, dw_create fails for a tree of another shape):
//     wave 0   left leg  1..6          + the short chain off the trunk's end (neck, head: 24, 25)
//     wave 1   right leg 7..12         + the base body's own inertia, ground primitives and push; the base's total
//     wave 2   trunk 13..15            + the longest chain off the trunk's end (left arm 16..23)
//     wave 3                             the other long chain off the trunk's end (right arm 26..33)
// Outward passes (kinematics, accelerations, velocity jumps) run root -> tip, so a chain that hangs off the trunk needs the
// trunk's running state first: waves 0 and 3 RECOMPUTE the trunk's three bodies themselves (same instructions as wave 2,
// nothing stored) instead of waiting for wave 2 at a barrier.  The inward pass (articulated inertias) runs tip -> root:
// waves 0 and 3 hand the totals of their trunk-end chains to wave 2 through LDS, wave 2 walks the trunk, wave 1 sums the
// legs and the base.
#pragma once

#include "dw_devmodel.h"
#include "dw_quad_model.h"          // quat_of_rot

namespace dwl {

constexpr int NB = dw::NB, ND = dw::ND;
constexpr int NWAVE = 4;
constexpr int LMAX_OUT = 12;         // bodies a wave visits in an outward pass (recomputed trunk included)
constexpr int LMAX_IN = 8;           // bodies of one inward chain
constexpr int LMAX_TRUNK = 4;
constexpr int LMAX_GYM = 3;          // Gym bodies welded into one moving body
constexpr int LO_STORE = 1, LO_START = 2, LO_SOLE = 4, LO_SHARED = 8;      // out_flags: owner stores; running state restarts at the base; sole body; trunk body every wave recomputes

// Constants the passes read once per body, packed in the order the waves walk them and staged into LDS when a kernel starts
// (a record is one or a few ds_read_b128 of a uniform address, requested together with the body's slot rows; read from the
// model in device memory the same constants were a chain of dependent scalar loads per body, ~400 cycles a hop for a wave
// that is alone on its SIMD).  Integer fields travel as bit patterns in float words.
struct alignas(16) LFkRec {          // outward passes, [wave offset + step]: 8 words
    float pos[3]; float pad;         // body origin in the parent frame
    float axis[3]; float vmax;       // hinge axis in body coordinates; joint speed limit
};
struct alignas(16) LInRec {          // inward pass, [wave offset + step]: 16 words
    float com[3]; float mass;        // first inertial record
    float I[6]; float bound; int bits;      // bits: body | nin << 8 | ngeom << 12 | (has proxies) << 16 | gym of the record << 24
    float axis[3]; unsigned pairs;   // pairs: bit k = self-collision pair k involves a proxy of this body
};
struct alignas(16) LProxRec { float p0[3], radius, p1[3]; int bits; };      // a capsule proxy: 8 words; bits: body | index of its Gym body among the body's << 8
struct alignas(16) LIn1Rec { float com[3]; float mass; float I[6]; int gym; int pad; };      // second (welded) inertial record of a sole body: 12 words
struct alignas(16) LJcRec { float kp, kv, qlo, qhi, ah, pad[3]; };            // per joint: PD gains, range, torque scale of the action
struct alignas(16) LGeomRec { int bits; float pos[3]; float rot[9]; float size[3]; };      // a ground primitive of a leg body: 16 words; bits: type | index of its Gym body among the body's << 8
constexpr int LHOT_FK = 40, LHOT_IN = 36, LHOT_GEOM = 24;
struct alignas(16) LHot {
    LFkRec   fk[LHOT_FK];
    float    q0[4][4];               // the fixed rotations body -> parent that are not the identity (xyzw); index + 1 in LCtl::out_q0
    LInRec   in[LHOT_IN];
    LProxRec prox[DW_MAX_SC_PROXIES];
    LIn1Rec  in1[2];
    float    foot[DW_NUM_FOOT_PTS][4];
    LJcRec   jc[NB];                 // [body]
    LGeomRec lgeom[LHOT_GEOM];       // the legs' non-sole primitives, in the order the leg waves' inward chains meet them
};

// Control words of a wave's passes, read once per kernel into scalar registers: which body a step visits (one byte per step)
// and one bit per step for every decision, so that the loops' control flow and slot addresses are scalar arithmetic and no
// pass waits for a table read before it can request a body's slot rows.
struct LCtl {
    unsigned out_b[3];                          // outward order: body of step k in byte k
    unsigned out_start, out_store, out_sole, out_shared;      // bit k: LO_* of step k
    unsigned out_q0;                            // 2 bits per step: 0 = no fixed rotation, else index + 1 into LHot::q0
    unsigned out_ts;                            // 2 bits per step: trunk slot of a shared body
    unsigned in_b[2][2];                        // inward chains a / b: body of step k in byte k
    unsigned in_gym[2][2];                      // Gym body of the step's first inertial record (mass scale), one byte per step
    unsigned in_two[2], in_geom[2], in_prox[2]; // bit k: two inertial records / has ground primitives / has self-collision proxies
    unsigned pair_ba[2], pair_bb[2];            // my (<= 8) detection pairs: bodies of the two proxies, one byte per pair
    unsigned pair_pa[2], pair_pb[2];            // ... and the proxies themselves
    unsigned in_g0[2], in_ng;                   // chain a (the leg waves'): first record of the step's primitives in LHot::lgeom (byte per step), their number (nibble per step)
    unsigned own_ts;                            // 2 bits per own joint: trunk slot + 1, or 0
    int n_out, n_in[2], fk_off, in_off[2], pair_lo, pair_n;
    unsigned own_b[3]; int n_own;               // the joints whose task state this wave keeps (the bodies it stores in the outward passes), one byte each
};

struct LaneModel {
    LHot hot;
    LCtl ctl[NWAVE];
    int fk_off[NWAVE], in_a_off[NWAVE], in_b_off[NWAVE];       // where a wave's records start in hot.fk / hot.in
    // outward order of each wave, and its two inward chains (tip -> root): a = first, b = second (wave 2: the trunk; wave 0: the neck chain)
    int n_out[NWAVE], out_body[NWAVE][LMAX_OUT], out_flags[NWAVE][LMAX_OUT];
    int n_in_a[NWAVE], in_a[NWAVE][LMAX_IN];
    int n_in_b[NWAVE], in_b[NWAVE][LMAX_IN];
    int n_trunk, trunk[LMAX_TRUNK];            // the shared chain below the base; trunk_slot[b] = its index there or -1
    int trunk_slot[NB];
    int owner[NB];                             // wave that runs the inward pass of a body (its ground primitives, its self-collision forces); body 0: wave 1
    float q0[NB][4]; int q0_set[NB];           // fixed rotation body -> parent as a quaternion; q0_set: it is not the identity
    float bound[NB];                           // a body whose origin is higher than this above the ground touches nothing
    int ngym[NB], gyms[NB][LMAX_GYM];          // Gym bodies that report this moving body's contact forces
    int geom_gslot[NB][dw::MAX_BODY_GEOMS];    // per ground primitive of the body: index into gyms[]
    int inert_gslot[NB][dw::MAX_BODY_INERT];
    // self-collision: proxies per body, pairs per proxy
    int nprox, npair;
    int prox_body[DW_MAX_SC_PROXIES], prox_gslot[DW_MAX_SC_PROXIES];
    unsigned prox_pairs[DW_MAX_SC_PROXIES];    // bit k: pair k involves this proxy
    int body_nprox[NB], body_prox[NB][4];
    unsigned body_pairs[NB];
    int pair_a[DW_MAX_SC_PAIRS], pair_b[DW_MAX_SC_PAIRS];
    int pair_lo[NWAVE + 1];                    // detection: wave w tests pairs pair_lo[w] .. pair_lo[w + 1] - 1
};

inline int build_lanemodel(const dw::DevModel *d, LaneModel *Q, const char **err) {
    using namespace dw;
    memset(Q, 0, sizeof(*Q));
    // ---- the tree's shape: two 6-body legs off the base, one trunk chain off the base, chains off the trunk's last body ----
    for (int f = 0; f < 2; ++f)
        for (int i = 1; i <= 6; ++i)
            if (d->parent[6 * f + i] != (i == 1 ? 0 : 6 * f + i - 1)) { *err = "lane kernels: legs must be the chains 1..6 and 7..12 off the base"; return DW_EINVAL; }
    int tr0 = -1;
    for (int b = 13; b < NB; ++b) if (d->parent[b] == 0) { if (tr0 >= 0) { *err = "lane kernels: more than one chain besides the legs hangs off the base"; return DW_EINVAL; } tr0 = b; }
    if (tr0 < 0) { *err = "lane kernels: no trunk chain off the base"; return DW_EINVAL; }
    Q->n_trunk = 0;
    for (int b = 0; b < NB; ++b) Q->trunk_slot[b] = -1;
    for (int b = tr0;;) {
        if (Q->n_trunk >= LMAX_TRUNK - 1) { *err = "lane kernels: trunk chain too long"; return DW_EINVAL; }
        Q->trunk_slot[b] = Q->n_trunk; Q->trunk[Q->n_trunk++] = b;
        if (d->nchild[b] != 1) break;
        b = d->child[b][0];
    }
    const int tend = Q->trunk[Q->n_trunk - 1];
    // chains off the trunk's end
    int ch_first[MAX_CHILD], ch_len[MAX_CHILD], nch = d->nchild[tend];
    if (nch < 2 || nch > 3) { *err = "lane kernels: the trunk must end in two or three chains"; return DW_EINVAL; }
    for (int c = 0; c < nch; ++c) {
        int b = d->child[tend][c], len = 1;
        ch_first[c] = b;
        while (d->nchild[b] == 1) { b = d->child[b][0]; ++len; }
        if (d->nchild[b] != 0) { *err = "lane kernels: a chain off the trunk's end branches"; return DW_EINVAL; }
        for (int i = 0, bb = ch_first[c]; i < len; ++i, bb = d->child[bb][0]) if (i + 1 < len && d->child[bb][0] != bb + 1) { *err = "lane kernels: chain bodies must be numbered consecutively"; return DW_EINVAL; }
        ch_len[c] = len;
    }
    // sort: the two longest go to waves 2 and 3, a third to wave 0
    int ord[MAX_CHILD] = {0, 1, 2};
    for (int i = 0; i < nch; ++i) for (int j = i + 1; j < nch; ++j) if (ch_len[ord[j]] > ch_len[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    if (ch_len[ord[0]] > LMAX_IN || (nch == 3 && ch_len[ord[2]] + 6 + Q->n_trunk > LMAX_OUT) || ch_len[ord[0]] + Q->n_trunk > LMAX_OUT) { *err = "lane kernels: chain longer than the role tables"; return DW_EINVAL; }
    // every body must be covered
    {
        int covered = 1 + 12 + Q->n_trunk;
        for (int c = 0; c < nch; ++c) covered += ch_len[c];
        if (covered != NB) { *err = "lane kernels: the tree is not legs + trunk + chains off the trunk's end"; return DW_EINVAL; }
    }
    auto push_out = [&](int w, int b, int fl) { Q->out_body[w][Q->n_out[w]] = b; Q->out_flags[w][Q->n_out[w]++] = fl; };
    auto push_trunk = [&](int w, bool store) { for (int i = 0; i < Q->n_trunk; ++i) push_out(w, Q->trunk[i], LO_SHARED | (store ? LO_STORE : 0) | (i == 0 ? LO_START : 0)); };
    auto push_chain = [&](int w, int first, int len, bool start) { for (int i = 0; i < len; ++i) push_out(w, first + i, LO_STORE | (start && i == 0 ? LO_START : 0) | ((first + i == d->foot_body[0] || first + i == d->foot_body[1]) ? LO_SOLE : 0)); };
    auto set_in = [&](int *n, int *list, int first, int len) { *n = len; for (int i = 0; i < len; ++i) list[i] = first + len - 1 - i; };
    Q->owner[0] = 1;
    // wave 0: left leg, then (third chain) trunk recomputed + neck chain
    push_chain(0, 1, 6, true); set_in(&Q->n_in_a[0], Q->in_a[0], 1, 6);
    for (int i = 1; i <= 6; ++i) Q->owner[i] = 0;
    if (nch == 3) {
        push_trunk(0, false); push_chain(0, ch_first[ord[2]], ch_len[ord[2]], false);
        set_in(&Q->n_in_b[0], Q->in_b[0], ch_first[ord[2]], ch_len[ord[2]]);
        for (int i = 0; i < ch_len[ord[2]]; ++i) Q->owner[ch_first[ord[2]] + i] = 0;
    }
    // wave 1: right leg
    push_chain(1, 7, 6, true); set_in(&Q->n_in_a[1], Q->in_a[1], 7, 6);
    for (int i = 7; i <= 12; ++i) Q->owner[i] = 1;
    // wave 2: trunk (stored) + longest chain; inward: the chain, then the trunk
    push_trunk(2, true); push_chain(2, ch_first[ord[0]], ch_len[ord[0]], false);
    set_in(&Q->n_in_a[2], Q->in_a[2], ch_first[ord[0]], ch_len[ord[0]]);
    Q->n_in_b[2] = Q->n_trunk; for (int i = 0; i < Q->n_trunk; ++i) Q->in_b[2][i] = Q->trunk[Q->n_trunk - 1 - i];
    for (int i = 0; i < Q->n_trunk; ++i) Q->owner[Q->trunk[i]] = 2;
    for (int i = 0; i < ch_len[ord[0]]; ++i) Q->owner[ch_first[ord[0]] + i] = 2;
    // wave 3: trunk recomputed + second chain
    push_trunk(3, false); push_chain(3, ch_first[ord[1]], ch_len[ord[1]], false);
    set_in(&Q->n_in_a[3], Q->in_a[3], ch_first[ord[1]], ch_len[ord[1]]);
    for (int i = 0; i < ch_len[ord[1]]; ++i) Q->owner[ch_first[ord[1]] + i] = 3;
    // ---- per-body constants ----
    for (int b = 0; b < NB; ++b) {
        dwq::quat_of_rot(d->rot0[b], Q->q0[b]);
        bool ident = true;
        for (int i = 0; i < 9; ++i) if (fabsf(d->rot0[b][i] - ((i % 4) == 0 ? 1.0f : 0.0f)) > 1e-7f) ident = false;
        if (ident || b == 0) { Q->q0[b][0] = Q->q0[b][1] = Q->q0[b][2] = 0.0f; Q->q0[b][3] = 1.0f; }
        Q->q0_set[b] = (ident || b == 0) ? 0 : 1;
        float bound = 0.0f;
        auto gslot = [&](int gym) {
            for (int t = 0; t < Q->ngym[b]; ++t) if (Q->gyms[b][t] == gym) return t;
            if (Q->ngym[b] >= LMAX_GYM) return -1;
            Q->gyms[b][Q->ngym[b]] = gym;
            return Q->ngym[b]++;
        };
        for (int k = 0; k < d->ninert[b]; ++k) { const int t = gslot(d->bi_gym[b][k]); if (t < 0) { *err = "lane kernels: more than 3 Gym bodies on one moving body"; return DW_EINVAL; } Q->inert_gslot[b][k] = t; }
        for (int k = 0; k < d->body_ngeom[b]; ++k) {
            const DwGeom &g = d->geoms[d->body_geom[b][k]];
            const int t = gslot(g.gym);
            if (t < 0) { *err = "lane kernels: more than 3 Gym bodies on one moving body"; return DW_EINVAL; }
            Q->geom_gslot[b][k] = t;
            const float pn = sqrtf(g.pos[0] * g.pos[0] + g.pos[1] * g.pos[1] + g.pos[2] * g.pos[2]);
            const float ext = g.type == 0 ? sqrtf(g.size[0] * g.size[0] + g.size[1] * g.size[1] + g.size[2] * g.size[2]) : sqrtf(g.size[0] * g.size[0] + g.size[1] * g.size[1]);
            if (pn + ext > bound) bound = pn + ext;
        }
        Q->bound[b] = bound * 1.01f + 1e-3f;
    }
    for (int f = 0; f < 2; ++f) {       // the sole's Gym body reports the contact solve's forces even if no primitive or inertial names it
        const int b = d->foot_body[f], gy = f == 0 ? d->left_foot_gym : d->right_foot_gym;
        bool have = false;
        for (int t = 0; t < Q->ngym[b]; ++t) have = have || Q->gyms[b][t] == gy;
        if (!have) { if (Q->ngym[b] >= LMAX_GYM) { *err = "lane kernels: sole body has no room for its Gym body"; return DW_EINVAL; } Q->gyms[b][Q->ngym[b]++] = gy; }
    }
    // ---- self-collision ----
    Q->npair = d->num_sc_pairs;
    int nprox = 0;
    for (int k = 0; k < d->num_sc_pairs; ++k) for (int s = 0; s < 2; ++s) if (d->sc_pair[k][s] + 1 > nprox) nprox = d->sc_pair[k][s] + 1;
    Q->nprox = nprox;
    for (int p = 0; p < nprox; ++p) {
        const int b = d->sc_proxy[p].moving;
        Q->prox_body[p] = b;
        int t = -1;
        for (int u = 0; u < Q->ngym[b]; ++u) if (Q->gyms[b][u] == d->sc_proxy[p].gym) t = u;
        if (t < 0) { if (Q->ngym[b] >= LMAX_GYM) { *err = "lane kernels: proxy names a fourth Gym body"; return DW_EINVAL; } t = Q->ngym[b]; Q->gyms[b][Q->ngym[b]++] = d->sc_proxy[p].gym; }
        Q->prox_gslot[p] = t;
        if (b == 0) { *err = "lane kernels: a self-collision proxy on the base is not supported"; return DW_EINVAL; }
        if (Q->body_nprox[b] >= 4) { *err = "lane kernels: more than 4 proxies on one body"; return DW_EINVAL; }
        Q->body_prox[b][Q->body_nprox[b]++] = p;
    }
    for (int k = 0; k < d->num_sc_pairs; ++k) {
        Q->pair_a[k] = d->sc_pair[k][0]; Q->pair_b[k] = d->sc_pair[k][1];
        for (int s = 0; s < 2; ++s) {
            const int p = d->sc_pair[k][s];
            Q->prox_pairs[p] |= 1u << k;
            Q->body_pairs[Q->prox_body[p]] |= 1u << k;
        }
    }
    // ---- the packed records ----
    {
        int nf = 0, ni = 0;
        for (int w = 0; w < NWAVE; ++w) {
            Q->fk_off[w] = nf;
            if (nf + Q->n_out[w] > LHOT_FK) { *err = "lane kernels: outward record table too small"; return DW_EINVAL; }
            for (int k = 0; k < Q->n_out[w]; ++k) {
                const int b = Q->out_body[w][k];
                LFkRec &r = Q->hot.fk[nf++];
                for (int i = 0; i < 3; ++i) { r.pos[i] = d->pos[b][i]; r.axis[i] = d->axis[b][i]; }
                r.vmax = d->vmax[b - 1]; r.pad = 0.0f;
            }
            for (int c = 0; c < 2; ++c) {
                const int n = c ? Q->n_in_b[w] : Q->n_in_a[w];
                const int *list = c ? Q->in_b[w] : Q->in_a[w];
                (c ? Q->in_b_off : Q->in_a_off)[w] = ni;
                if (ni + n > LHOT_IN) { *err = "lane kernels: inward record table too small"; return DW_EINVAL; }
                for (int k = 0; k < n; ++k) {
                    const int b = list[k];
                    LInRec &r = Q->hot.in[ni++];
                    for (int i = 0; i < 3; ++i) { r.com[i] = d->bi_com[b][0][i]; r.axis[i] = d->axis[b][i]; }
                    r.mass = d->bi_mass[b][0];
                    for (int i = 0; i < 6; ++i) r.I[i] = d->bi_I[b][0][i];
                    r.bound = Q->bound[b];
                    r.bits = b | (d->ninert[b] << 8) | (d->body_ngeom[b] << 12) | ((Q->body_nprox[b] > 0 ? 1 : 0) << 16) | (d->bi_gym[b][0] << 24);
                    r.pairs = Q->body_pairs[b];
                }
            }
        }
        for (int p = 0; p < nprox; ++p) {
            LProxRec &r = Q->hot.prox[p];
            for (int i = 0; i < 3; ++i) { r.p0[i] = d->sc_proxy[p].p0[i]; r.p1[i] = d->sc_proxy[p].p1[i]; }
            r.radius = d->sc_proxy[p].radius;
            r.bits = Q->prox_body[p] | (Q->prox_gslot[p] << 8);
        }
        for (int b = 1; b < NB; ++b) {
            LJcRec &r = Q->hot.jc[b];
            r.kp = d->kp[b - 1]; r.kv = d->kv[b - 1]; r.qlo = d->qlo[b - 1]; r.qhi = d->qhi[b - 1]; r.ah = d->action_high[b - 1];
        }
        {   // fixed rotations
            int nq = 0;
            for (int b = 1; b < NB; ++b) if (Q->q0_set[b]) {
                if (nq >= 3) { *err = "lane kernels: more than three bodies with a fixed rotation"; return DW_EINVAL; }
                for (int i = 0; i < 4; ++i) Q->hot.q0[nq][i] = Q->q0[b][i];
                Q->q0_set[b] = ++nq;          // (index + 1)
            }
        }
        for (int f = 0; f < 2; ++f) {
            const int b = d->foot_body[f];
            LIn1Rec &r = Q->hot.in1[f];
            if (d->ninert[b] > 1) {
                for (int i = 0; i < 3; ++i) r.com[i] = d->bi_com[b][1][i];
                r.mass = d->bi_mass[b][1];
                for (int i = 0; i < 6; ++i) r.I[i] = d->bi_I[b][1][i];
                r.gym = d->bi_gym[b][1];
            }
        }
        for (int k = 0; k < DW_NUM_FOOT_PTS; ++k) for (int i = 0; i < 3; ++i) Q->hot.foot[k][i] = d->foot_pos[k][i];
    }
    for (int w = 0; w <= NWAVE; ++w) Q->pair_lo[w] = (Q->npair * w + NWAVE - 1) / NWAVE > Q->npair ? Q->npair : (Q->npair * w + NWAVE - 1) / NWAVE;
    Q->pair_lo[NWAVE] = Q->npair;
    // ---- control words ----
    for (int w = 0; w < NWAVE; ++w) {
        LCtl &c = Q->ctl[w];
        c.n_out = Q->n_out[w]; c.fk_off = Q->fk_off[w];
        for (int k = 0; k < Q->n_out[w]; ++k) {
            const int b = Q->out_body[w][k], fl = Q->out_flags[w][k];
            c.out_b[k >> 2] |= (unsigned)b << (8 * (k & 3));
            if (fl & LO_START) c.out_start |= 1u << k;
            if (fl & LO_STORE) c.out_store |= 1u << k;
            if (fl & LO_SOLE) c.out_sole |= 1u << k;
            if (fl & LO_SHARED) { c.out_shared |= 1u << k; c.out_ts |= (unsigned)Q->trunk_slot[b] << (2 * k); }
            c.out_q0 |= (unsigned)Q->q0_set[b] << (2 * k);
        }
        for (int ch = 0; ch < 2; ++ch) {
            const int n = ch ? Q->n_in_b[w] : Q->n_in_a[w];
            const int *list = ch ? Q->in_b[w] : Q->in_a[w];
            c.n_in[ch] = n; c.in_off[ch] = ch ? Q->in_b_off[w] : Q->in_a_off[w];
            for (int k = 0; k < n; ++k) {
                const int b = list[k];
                c.in_b[ch][k >> 2] |= (unsigned)b << (8 * (k & 3));
                c.in_gym[ch][k >> 2] |= (unsigned)d->bi_gym[b][0] << (8 * (k & 3));
                if (d->ninert[b] > 1) c.in_two[ch] |= 1u << k;
                if (d->body_ngeom[b] > 0) c.in_geom[ch] |= 1u << k;
                if (Q->body_nprox[b] > 0) c.in_prox[ch] |= 1u << k;
            }
        }
        for (int k = 0; k < Q->n_out[w]; ++k)
            if (Q->out_flags[w][k] & LO_STORE) {
                const int b = Q->out_body[w][k];
                c.own_b[c.n_own >> 2] |= (unsigned)b << (8 * (c.n_own & 3));
                c.own_ts |= (unsigned)(Q->trunk_slot[b] + 1) << (2 * c.n_own);
                ++c.n_own;
            }
        // detection: wave w tests the pairs w, w + 4, w + 8, ... (the model lists a proxy's pairs together, and touching pairs cluster
        // on the legs: interleaved, every wave gets its share of them)
        c.pair_lo = w; c.pair_n = (Q->npair - w + NWAVE - 1) / NWAVE;
        if (c.pair_n < 0) c.pair_n = 0;
        if (c.pair_n > 8) { *err = "lane kernels: more than 8 detection pairs per wave"; return DW_EINVAL; }
        for (int i = 0; i < c.pair_n; ++i) {
            const int k = w + NWAVE * i;
            c.pair_ba[i >> 2] |= (unsigned)d->scp[k].ba << (8 * (i & 3));
            c.pair_bb[i >> 2] |= (unsigned)d->scp[k].bb << (8 * (i & 3));
            c.pair_pa[i >> 2] |= (unsigned)d->sc_pair[k][0] << (8 * (i & 3));
            c.pair_pb[i >> 2] |= (unsigned)d->sc_pair[k][1] << (8 * (i & 3));
        }
    }
    if (Q->n_trunk > 3) { *err = "lane kernels: trunk slots are two bits"; return DW_EINVAL; }
    {   // the legs' ground primitives, in the order of the leg waves' first inward chain
        int ng = 0;
        for (int w = 0; w < 2; ++w) {
            LCtl &c = Q->ctl[w];
            for (int k = 0; k < Q->n_in_a[w]; ++k) {
                const int b = Q->in_a[w][k], n = d->body_ngeom[b];
                if (n > 15 || ng + n > LHOT_GEOM) { *err = "lane kernels: too many ground primitives on the legs"; return DW_EINVAL; }
                c.in_g0[k >> 2] |= (unsigned)ng << (8 * (k & 3));
                c.in_ng |= (unsigned)n << (4 * k);
                for (int g = 0; g < n; ++g) {
                    const DwGeom &s = d->geoms[d->body_geom[b][g]];
                    LGeomRec &r = Q->hot.lgeom[ng++];
                    r.bits = s.type | (Q->geom_gslot[b][g] << 8);
                    for (int i = 0; i < 3; ++i) { r.pos[i] = s.pos[i]; r.size[i] = s.size[i]; }
                    for (int i = 0; i < 9; ++i) r.rot[i] = s.rot[i];
                }
            }
        }
    }
    return DW_OK;
}

}  // namespace dwl
