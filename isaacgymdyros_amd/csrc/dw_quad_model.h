// dw_quad_model.h -- limb schedule and per-(step, lane) constant tables of the octet kernels (dw_oct.h; the name is that of the
// retired quad kernels, which introduced the schedule), and the host
// routine that derives them from the traversal tables of dw_devmodel.h.
//
// A quad = the 4 lanes of one env.  Every lane owns a list of unbranched chains of the kinematic tree and walks them body
// by body: outward passes (kinematics, accelerations, velocity jumps) in schedule order, the inward pass (articulated
// inertias) in exactly the reverse order.  For TOCABI the derived schedule is
//     step     0   1   2   3   4   5   6   7   8   9  10
//     lane 0   .   .   .  24  25   1   2   3   4   5   6     neck, head, left leg
//     lane 1   .   .   .   .   .   7   8   9  10  11  12     right leg
//     lane 2  13  14  15  16  17  18  19  20  21  22  23     waist, left arm
//     lane 3   .   .   .  26  27  28  29  30  31  32  33     right arm
// Where limbs meet, data moves between the lanes of the quad with DPP permutes, under these scheduling rules (checked
// here; dw_create fails if a model cannot be scheduled):
//   outward: a chain whose parent body belongs to another lane starts in the step right after that parent's step, so the
//            parent's running state is still in the owner's registers when it is fetched;
//   inward:  when a chain ends and its parent body is not the lane's next body, its reflected inertia stays in the lane's
//            running registers until the parent's lane gathers it, or is parked (one parking slot per lane) if the lane
//            starts another chain first.
// The two leg chains are right-aligned (they end in the last step) so that both sole bodies are handled in the same
// inward step; the contact pipeline needs the left leg on lane 0 and the right leg on lane 1.
#pragma once

#include "dw_devmodel.h"

namespace dwq {

constexpr int EPW = 16;          // environments per wavefront
constexpr int QS_MAX = 11;       // schedule length bound (TOCABI needs 11; every step costs 448 B of LDS)
constexpr int QMAX_PROX = 16;    // self-collision proxies
constexpr int QMAX_ROUNDS = 8;   // detection rounds of the octet kernels (see QHot::scround): 8 pairs each
constexpr int QMAX_OWN = 5;      // proxies on the bodies of one lane (TOCABI: head + 4 of the left leg, 4 right leg, torso + 3 on the left arm's lane, 3 on the right arm's)
constexpr int QMAX_GEOM = 6;     // ground primitives per moving body the inward step handles
constexpr int QMAX_GYM = 3;      // Gym bodies welded into one moving body

struct alignas(16) QFkRec {      // outward constants of one (step, lane): 4 x 16 B
    float pos[3]; int body;      // body origin in the parent frame; body = -1: the lane idles in this step
    float axis[3]; int psrc;     // hinge axis in body coordinates; parent state: 0 running, 1 base, 2+X from lane X (DPP)
    float q0[4];                 // quaternion (xyzw) of the fixed rotation body -> parent
    float qlo, qhi, vmax; int flags;   // bit 0: q0 is not the identity, bit 1: sole body (keep its pose for the contact phase),
                                       // bits 8..15: self-collision proxies on this body
};

struct alignas(16) QInRec {      // inward constants of one (step, lane): 11 x 16 B
    int body, flags, gather, nin;            // flags bit 0: fresh (no running contribution yet), bit 1: park the running
                                             // state before this body, bit 2: last body of a chain whose parent is elsewhere
                                             // gather: 4 bits per source (up to 3): lane | parked << 2 | valid << 3
    float in0_com[3]; float in0_mass;
    float in0_I[6];   int in0_gym; int ngym;
    float in1_com[3]; float in1_mass;
    float in1_I[6];   int in1_gym; int ngeom;
    int   gyms[QMAX_GYM]; float bound;       // Gym bodies of this moving body (contact attribution); ground test radius
    int   geom[QMAX_GEOM]; int geom_slot;    // primitive ids; 2 bits per primitive: index into gyms[]
    int   pad0;
    float axis[3]; int sc_mask;              // hinge axis in body coordinates; bit k: the lane's k-th own proxy sits on this body
};

// The part of the tables every step of every pass reads, staged into LDS when a kernel starts (5.4 KB): compact
// per-(step, lane) records and the wave-uniform masks.  Integer fields travel as bit patterns in float words.
//   fk[s][l]: [0..2] pos, [3] body | psrc << 8 | flags << 12 | proxies << 16, [4..6] axis, [7] vmax, [8] qlo, [9] qhi
//   in[s][l]: [0] body + 1 | flags << 8 | nin << 12 | ngym << 14 | ngeom << 16 | proxies << 24, [1] gather,
//             [2] gyms[0] | gyms[1] << 8 | gyms[2] << 16 | in0_gym << 24, [3] bound, [4..13] the first inertial record PREPARED
//             (prepared_inertia below: mass * com, mass, inertia about the body origin)
struct alignas(16) QHot {
    float fk[QS_MAX][4][12];
    float in[QS_MAX][4][16];
    float base[16];              // [0..2] com, [3] mass, [4..9] I, [10] gym, [11] ngeom, [12] bound, [13] cell bases, [14] first / last step per lane, [15] steps in which a chain starts
    int   fmask[QS_MAX];         // outward step s: bit X set = some lane fetches lane X's running state
    int   gany[QS_MAX];          // inward step s: bit 0 some lane gathers; bits 8.. accumulation (valid | source lane << 1 | destination lane << 3)
    int   misc[8];               // [0] nsteps, [1] base_gather, [2] number of proxies | detection rounds per class << 8 (scround), [3] pairs << 8, [4..7] pairs of lane l
    unsigned char owner[36];     // per body: lane that owns it (bits 6..7) | its slot CELL in the octet kernels (bits 0..5):
                                 // cell = cellbase[lane] + outward step, so that a limb's bodies lie in schedule order and a chain
                                 // pass addresses them as lane base + compile-time offset (dw_oct.h); base[13] = the four cell bases
    alignas(16) float in1[2][12];   // second (welded) inertial record of the sole bodies, per leg lane, prepared as in[][4..13]; [10] its Gym body (int bits), [11] its inward step (int bits)
    // self-collision proxies: [0..2] p0, [3] radius, [4..6] p1, [7] body | gym << 8 | owner lane << 16 | index among the
    // owner's proxies << 18.  pairs: byte k = proxies of pair k (a | b << 4); misc[4 + l]: the pairs lane l resolves (one of the
    // two proxies sits on one of its bodies).
    // Detection (dw_oct.h): octet lane o builds the axes of proxies o (class 0) and o + 8 (class 1) once; the pairs are then tested
    // in ROUNDS of eight, one pair per octet lane, the two axes fetched from the lanes that built them.  A round's pairs all have
    // the same classes (a, b) -- (0,0), then (1,1), then (1,0); misc[2] = nprox | n00 << 8 | n11 << 12 | n10 << 16 -- so that the
    // fetches name fixed registers.  scround[r][o] = lane of a | lane of b << 3 | pair id << 6 (127: none) | threshold << 16, the
    // threshold 1.004 (ra + rb)^2 as a half-precision number rounded UP (detection is conservative; the force is exact).
    alignas(16) float prox[QMAX_PROX][8];
    int   scround[QMAX_ROUNDS][8];
    int   pairs[16];             // byte k: the proxies of pair k (a | b << 4)
    int   pairmask_hi[4];        // pairs 32 .. 63 of lane l (misc[4 + l]: pairs 0 .. 31)
};

struct QuadModel {
    QHot  hot;
    int   nsteps;
    int   owner[dw::NB];         // lane of each body (body 0: all)
    int   step_of[dw::NB];
    int   base_gather;           // as QInRec.gather, for the root
    int   pad[2];
    QFkRec fk[QS_MAX][4];
    QInRec in[QS_MAX][4];        // in[s] = inward step s (= outward step nsteps-1-s)
    // base body
    float base_com[3]; float base_mass; float base_I[6]; int base_gym; int base_ngeom; int base_geom[QMAX_GEOM]; float base_bound;
    int   nprox, npair;                      // self-collision
    int   own_proxy[4][QMAX_OWN];            // per lane the proxies on its bodies (index or -1): bit k of QInRec.sc_mask
};

static inline void quat_of_rot(const float *R, float *q) {     // row-major rotation -> unit quaternion xyzw (host only)
    const double m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7], m22 = R[8];
    double x, y, z, w;
    const double tr = m00 + m11 + m22;
    if (tr > 0) { double s = __builtin_sqrt(tr + 1.0) * 2; w = 0.25 * s; x = (m21 - m12) / s; y = (m02 - m20) / s; z = (m10 - m01) / s; }
    else if (m00 > m11 && m00 > m22) { double s = __builtin_sqrt(1.0 + m00 - m11 - m22) * 2; w = (m21 - m12) / s; x = 0.25 * s; y = (m01 + m10) / s; z = (m02 + m20) / s; }
    else if (m11 > m22) { double s = __builtin_sqrt(1.0 + m11 - m00 - m22) * 2; w = (m02 - m20) / s; x = (m01 + m10) / s; y = 0.25 * s; z = (m12 + m21) / s; }
    else { double s = __builtin_sqrt(1.0 + m22 - m00 - m11) * 2; w = (m10 - m01) / s; x = (m02 + m20) / s; y = (m12 + m21) / s; z = 0.25 * s; }
    q[0] = (float)x; q[1] = (float)y; q[2] = (float)z; q[3] = (float)w;
}

// An inertial record as the inward pass wants it (dw_limb.h rigid_inertia_pre): out[0..2] = mass * com, [3] = mass, [4..9] = the inertia
// about the BODY ORIGIN in body axes, I + mass (|com|^2 1 - com com'), in the order xx yy zz xy xz yz -- formed once here, in double.
static inline void prepared_inertia(const float *com, float mass, const float *I6, float *out) {
    const double c[3] = {com[0], com[1], com[2]}, m = mass, cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    for (int i = 0; i < 3; ++i) out[i] = (float)(m * c[i]);
    out[3] = mass;
    out[4] = (float)(I6[0] + m * (cc - c[0] * c[0])); out[5] = (float)(I6[1] + m * (cc - c[1] * c[1])); out[6] = (float)(I6[2] + m * (cc - c[2] * c[2]));
    out[7] = (float)(I6[3] - m * c[0] * c[1]); out[8] = (float)(I6[4] - m * c[0] * c[2]); out[9] = (float)(I6[5] - m * c[1] * c[2]);
}

// Builds the schedule and the constant tables.  Returns 0 or DW_EINVAL with a message.
// accumulate = true (octet kernels): a finished chain that would have to be parked is instead ADDED into the running registers
// of an idle lane whose own finished chain waits for the same parent (TOCABI: the left leg joins the right leg's lane when
// the left leg's lane starts the neck chain), so no lane ever carries a second, parked articulated inertia.
inline int build_quadmodel(const dw::DevModel *d, const DwModel *dm, QuadModel *Q, const char **err, bool accumulate = false) {
    using namespace dw;
    memset(Q, 0, sizeof(*Q));
    const int nc = d->nchains;
    int chain_of[NB]; chain_of[0] = -1;
    for (int c = 0; c < nc; ++c) for (int i = 0; i < d->chain_len[c]; ++i) chain_of[d->chain_body[c][i]] = c;
    int lane_of_chain[MAX_CHAINS], cparent[MAX_CHAINS];
    for (int c = 0; c < nc; ++c) { lane_of_chain[c] = -1; cparent[c] = d->parent[d->chain_body[c][0]]; }
    int load[4] = {0, 0, 0, 0};
    int legc[2] = {-1, -1};
    for (int c = 0; c < nc; ++c) {
        if (d->chain_body[c][0] == 1) legc[0] = c;
        if (d->chain_body[c][0] == 7) legc[1] = c;
    }
    if (legc[0] < 0 || legc[1] < 0 || d->chain_len[legc[0]] != 6 || d->chain_len[legc[1]] != 6) { *err = "quad schedule: leg chains not found"; return DW_EINVAL; }
    lane_of_chain[legc[0]] = 0; lane_of_chain[legc[1]] = 1;
    load[0] = load[1] = 6;
    // the other chains, by phase: a chain hanging off the END of a chain continues on that chain's lane if it is the
    // longest such child; everything else goes to the least loaded lane (ties: 3, 2, 0, 1 so trunk/arms fill lanes 2, 3 first)
    for (int ph = 0; ph < d->nphases; ++ph) {
        bool cont_taken[MAX_CHAINS] = {};
        for (int pass = 0; pass < 2; ++pass) {
            for (;;) {
                int best = -1;
                for (int c = 0; c < nc; ++c)
                    if (lane_of_chain[c] < 0 && d->chain_phase[c] == ph && (best < 0 || d->chain_len[c] > d->chain_len[best])) best = c;
                if (best < 0) break;
                int lane = -1;
                const int pb = cparent[best];
                if (pb != 0) {
                    const int pc = chain_of[pb];
                    if (pass == 0 && !cont_taken[pc] && d->chain_body[pc][d->chain_len[pc] - 1] == pb && lane_of_chain[pc] >= 2) {
                        lane = lane_of_chain[pc]; cont_taken[pc] = true;
                    }
                }
                if (lane < 0) {
                    const int pref[4] = {3, 2, 0, 1};
                    for (int k = 0; k < 4; ++k) if (lane < 0 || load[pref[k]] < load[lane]) lane = pref[k];
                    if (ph == 0) lane = load[2] <= load[3] ? 2 : 3;
                }
                lane_of_chain[best] = lane;
                load[lane] += d->chain_len[best];
            }
        }
    }
    // outward timing: per lane the chains in dependency order, legs last; cross-lane starts exactly one step after the parent
    int start[MAX_CHAINS]; for (int c = 0; c < nc; ++c) start[c] = -1;
    int step_of[NB]; for (int b = 0; b < NB; ++b) step_of[b] = -1;
    int lane_free[4] = {0, 0, 0, 0};
    for (int round = 0; round < nc; ++round) {
        bool progress = false;
        for (int ph = 0; ph < d->nphases; ++ph)
            for (int c = 0; c < nc; ++c) {
                if (start[c] >= 0 || d->chain_phase[c] != ph || c == legc[0] || c == legc[1]) continue;
                const int pb = cparent[c], lane = lane_of_chain[c];
                int t = lane_free[lane];
                if (pb != 0) {
                    if (step_of[pb] < 0) continue;
                    const int need = step_of[pb] + 1;
                    if (t > need && lane_of_chain[chain_of[pb]] != lane) { *err = "quad schedule: a cross-lane chain cannot start right after its parent"; return DW_EINVAL; }
                    if (t < need) t = need;
                    if (lane_of_chain[chain_of[pb]] == lane && t != need) { *err = "quad schedule: same-lane child does not follow its parent"; return DW_EINVAL; }
                }
                start[c] = t;
                for (int i = 0; i < d->chain_len[c]; ++i) step_of[d->chain_body[c][i]] = t + i;
                lane_free[lane] = t + d->chain_len[c];
                progress = true;
            }
        if (!progress) break;
    }
    for (int c = 0; c < nc; ++c) if (start[c] < 0 && c != legc[0] && c != legc[1]) { *err = "quad schedule: unschedulable chain"; return DW_EINVAL; }
    int T = 0;
    for (int l = 0; l < 4; ++l) if (lane_free[l] > T) T = lane_free[l];
    for (int f = 0; f < 2; ++f) if (lane_free[f] + 6 > T) T = lane_free[f] + 6;
    if (T > QS_MAX) { *err = "quad schedule: longer than QS_MAX steps"; return DW_EINVAL; }
    for (int f = 0; f < 2; ++f) {          // legs right-aligned
        start[legc[f]] = T - 6;
        for (int i = 0; i < 6; ++i) step_of[d->chain_body[legc[f]][i]] = T - 6 + i;
    }
    Q->nsteps = T;
    int body_at[QS_MAX][4];
    for (int s = 0; s < QS_MAX; ++s) for (int l = 0; l < 4; ++l) body_at[s][l] = -1;
    Q->owner[0] = -1; Q->step_of[0] = -1;
    for (int b = 1; b < NB; ++b) {
        const int l = lane_of_chain[chain_of[b]];
        if (body_at[step_of[b]][l] >= 0) { *err = "quad schedule: two bodies in one slot"; return DW_EINVAL; }
        body_at[step_of[b]][l] = b; Q->owner[b] = l; Q->step_of[b] = step_of[b];
    }
    // ---- outward records ----
    for (int s = 0; s < QS_MAX; ++s)
        for (int l = 0; l < 4; ++l) {
            QFkRec &r = Q->fk[s][l];
            r.body = -1; r.q0[3] = 1.0f; r.qlo = -1e30f; r.qhi = 1e30f; r.vmax = 1e30f;
            const int b = s < T ? body_at[s][l] : -1;
            if (b < 0) continue;
            r.body = b;
            for (int i = 0; i < 3; ++i) { r.pos[i] = d->pos[b][i]; r.axis[i] = d->axis[b][i]; }
            quat_of_rot(d->rot0[b], r.q0);
            bool ident = true;
            for (int i = 0; i < 9; ++i) if (fabsf(d->rot0[b][i] - ((i % 4) == 0 ? 1.0f : 0.0f)) > 1e-7f) ident = false;
            if (ident) { r.q0[0] = r.q0[1] = r.q0[2] = 0.0f; r.q0[3] = 1.0f; }
            r.flags = (ident ? 0 : 1) | ((b == d->foot_body[0] || b == d->foot_body[1]) ? 2 : 0);
            r.qlo = d->qlo[b - 1]; r.qhi = d->qhi[b - 1]; r.vmax = d->vmax[b - 1];
            const int p = d->parent[b];
            if (p == 0) r.psrc = 1;
            else if (Q->owner[p] == l) {
                if (step_of[p] != s - 1) { *err = "quad schedule: same-lane parent is not the previous step"; return DW_EINVAL; }
                r.psrc = 0;
            } else {
                if (step_of[p] != s - 1) { *err = "quad schedule: cross-lane parent is not the previous step"; return DW_EINVAL; }
                r.psrc = 2 + Q->owner[p];
            }
        }
    // ---- inward records (reverse order), hand-over bookkeeping ----
    // state per lane while simulating the inward pass: what its running registers / parking slot hold
    int running_chain[4] = {-1, -1, -1, -1}, parked_chain[4] = {-1, -1, -1, -1};
    int acc_step[QS_MAX] = {};
    bool gathered[MAX_CHAINS] = {};
    for (int s = 0; s < T; ++s) {
        // all lanes execute: (park) -> (gather) -> body.  Decide parks first for this step.
        for (int l = 0; l < 4; ++l) {
            QInRec &r = Q->in[s][l];
            const int b = body_at[T - 1 - s][l];
            r.body = b; r.in0_gym = 0; r.in1_gym = 0;
            if (b < 0) continue;
            const int c = chain_of[b];
            const bool tip = (b == d->chain_body[c][d->chain_len[c] - 1]);
            bool fresh = false;
            if (tip) {
                // does a child chain continue in this lane's running registers?  (its root was the previous inward body)
                bool cont = false;
                if (running_chain[l] >= 0 && !gathered[running_chain[l]] && cparent[running_chain[l]] == b) cont = true;
                if (!cont) {
                    fresh = true;
                    if (running_chain[l] >= 0 && !gathered[running_chain[l]]) {
                        int x = -1;
                        if (accumulate)
                            for (int c2 = 0; c2 < 4; ++c2)
                                if (c2 != l && body_at[T - 1 - s][c2] < 0 && running_chain[c2] >= 0 && !gathered[running_chain[c2]] &&
                                    cparent[running_chain[c2]] == cparent[running_chain[l]]) x = c2;
                        if (x >= 0) {
                            if (acc_step[s]) { *err = "quad schedule: two accumulations in one step"; return DW_EINVAL; }
                            acc_step[s] = 1 | (l << 1) | (x << 3);          // valid | source lane | destination lane
                            gathered[running_chain[l]] = true;              // (it travels with lane x's chain from here on)
                        } else {
                            if (parked_chain[l] >= 0 && !gathered[parked_chain[l]]) { *err = "quad schedule: two ungathered chains on one lane"; return DW_EINVAL; }
                            parked_chain[l] = running_chain[l];
                            r.flags |= 2;
                        }
                    }
                } else gathered[running_chain[l]] = true;
                running_chain[l] = -1;
            }
            if (fresh) r.flags |= 1;
        }
        for (int l = 0; l < 4; ++l) {
            QInRec &r = Q->in[s][l];
            const int b = r.body;
            if (b < 0) continue;
            // gather the other child chains of b
            int ng = 0;
            for (int c = 0; c < nc; ++c) {
                if (cparent[c] != b || gathered[c]) continue;
                int src = -1, parked = 0;
                for (int x = 0; x < 4; ++x) {
                    if (running_chain[x] == c) { src = x; parked = 0; }
                    if (parked_chain[x] == c) { src = x; parked = 1; }
                }
                if (src < 0 || src == l) { *err = "quad schedule: child chain not available at its parent"; return DW_EINVAL; }
                if (ng >= 3) { *err = "quad schedule: more than 3 gathers at one body"; return DW_EINVAL; }
                r.gather |= (src | (parked << 2) | 8) << (4 * ng++);
                gathered[c] = true;
            }
        }
        for (int l = 0; l < 4; ++l) {      // after the body: which chain do the running registers hold now
            const int b = Q->in[s][l].body;
            if (b < 0) continue;
            const int c = chain_of[b];
            if (b == d->chain_body[c][0]) { running_chain[l] = c; Q->in[s][l].flags |= 4; }
        }
    }
    {   // the root gathers what is left
        int ng = 0;
        for (int c = 0; c < nc; ++c) {
            if (gathered[c]) continue;
            if (cparent[c] != 0) { *err = "quad schedule: ungathered chain below the root"; return DW_EINVAL; }
            int src = -1, parked = 0;
            for (int x = 0; x < 4; ++x) { if (running_chain[x] == c) { src = x; parked = 0; } if (parked_chain[x] == c) { src = x; parked = 1; } }
            if (src < 0) { *err = "quad schedule: root child chain lost"; return DW_EINVAL; }
            if (ng >= 4) { *err = "quad schedule: more than 4 chains at the root"; return DW_EINVAL; }
            Q->base_gather |= (src | (parked << 2) | 8) << (4 * ng++);
        }
    }
    // ---- per-body constants of the inward records ----
    auto fill_inert = [&](int b, int k, float *com, float *mass, float *I, int *gym) {
        *mass = d->bi_mass[b][k]; *gym = d->bi_gym[b][k];
        for (int i = 0; i < 3; ++i) com[i] = d->bi_com[b][k][i];
        for (int i = 0; i < 6; ++i) I[i] = d->bi_I[b][k][i];
    };
    auto geom_bound = [&](int b, int ng, const int *gl) {
        float bound = 0.0f;
        for (int k = 0; k < ng; ++k) {
            const DwGeom &g = d->geoms[gl[k]];
            const float pn = sqrtf(g.pos[0] * g.pos[0] + g.pos[1] * g.pos[1] + g.pos[2] * g.pos[2]);
            const float ext = g.type == 0 ? sqrtf(g.size[0] * g.size[0] + g.size[1] * g.size[1] + g.size[2] * g.size[2])
                                          : sqrtf(g.size[0] * g.size[0] + g.size[1] * g.size[1]);
            if (pn + ext > bound) bound = pn + ext;
        }
        (void)b;
        return bound * 1.01f + 1e-3f;
    };
    for (int s = 0; s < T; ++s)
        for (int l = 0; l < 4; ++l) {
            QInRec &r = Q->in[s][l];
            const int b = r.body;
            if (b < 0) continue;
            r.nin = d->ninert[b];
            if (r.nin > 2) { *err = "quad model: more than two inertial records on a body"; return DW_EINVAL; }
            if (r.nin >= 1) fill_inert(b, 0, r.in0_com, &r.in0_mass, r.in0_I, &r.in0_gym);
            if (r.nin >= 2) fill_inert(b, 1, r.in1_com, &r.in1_mass, r.in1_I, &r.in1_gym);
            for (int i = 0; i < 3; ++i) r.axis[i] = d->axis[b][i];
            if (d->body_ngeom[b] > QMAX_GEOM) { *err = "quad model: too many ground primitives on one body"; return DW_EINVAL; }
            r.ngeom = d->body_ngeom[b];
            r.ngym = 0;
            for (int k = 0; k < r.ngeom; ++k) {
                r.geom[k] = d->body_geom[b][k];
                const int gy = d->body_geom_gym[b][k];
                int t = -1;
                for (int i = 0; i < r.ngym; ++i) if (r.gyms[i] == gy) t = i;
                if (t < 0) { if (r.ngym >= QMAX_GYM) { *err = "quad model: too many Gym bodies on one moving body"; return DW_EINVAL; } t = r.ngym; r.gyms[r.ngym++] = gy; }
                r.geom_slot |= t << (2 * k);
            }
            r.bound = geom_bound(b, r.ngeom, r.geom);
        }
    // Gym bodies that report forces without carrying a ground primitive (sole bodies, self-collision proxies) are added below
    auto add_gym = [&](int b, int gy) -> int {
        QInRec &r = Q->in[T - 1 - step_of[b]][Q->owner[b]];
        for (int i = 0; i < r.ngym; ++i) if (r.gyms[i] == gy) return i;
        if (r.ngym >= QMAX_GYM) return -1;
        r.gyms[r.ngym] = gy;
        return r.ngym++;
    };
    for (int f = 0; f < 2; ++f) if (add_gym(d->foot_body[f], f == 0 ? d->left_foot_gym : d->right_foot_gym) < 0) { *err = "quad model: sole Gym body does not fit"; return DW_EINVAL; }
    // base
    if (d->ninert[0] != 1) { *err = "quad model: the root must carry exactly one inertial record"; return DW_EINVAL; }
    fill_inert(0, 0, Q->base_com, &Q->base_mass, Q->base_I, &Q->base_gym);
    if (d->body_ngeom[0] > QMAX_GEOM) { *err = "quad model: too many ground primitives on the root"; return DW_EINVAL; }
    Q->base_ngeom = d->body_ngeom[0];
    for (int k = 0; k < Q->base_ngeom; ++k) {
        Q->base_geom[k] = d->body_geom[0][k];
        if (d->body_geom_gym[0][k] != Q->base_gym) { *err = "quad model: root primitives must report on the root's Gym body"; return DW_EINVAL; }
    }
    Q->base_bound = geom_bound(0, Q->base_ngeom, Q->base_geom);
    // self-collision: proxies keep the model's order (the model compiler lists a pair as [tested proxy, broadcast proxy] and
    // orders the proxies so that the tested proxies of the pairs sharing a partner sit on different lanes, model.py)
    if (d->num_sc_pairs > 0 && (d->num_sc_pairs > 64 || DW_MAX_SC_PROXIES > QMAX_PROX)) { *err = "quad model: too many self-collision pairs or proxies"; return DW_EINVAL; }
    Q->npair = d->num_sc_pairs;
    Q->nprox = 0;
    for (int k = 0; k < d->num_sc_pairs; ++k) for (int side = 0; side < 2; ++side) if (d->sc_pair[k][side] + 1 > Q->nprox) Q->nprox = d->sc_pair[k][side] + 1;
    for (int l = 0; l < 4; ++l) for (int k = 0; k < QMAX_OWN; ++k) Q->own_proxy[l][k] = -1;
    int local_of[QMAX_PROX];
    for (int p2 = 0; p2 < QMAX_PROX; ++p2) local_of[p2] = -1;
    for (int p2 = 0; p2 < Q->nprox; ++p2) {
        const int pb = d->sc_proxy[p2].moving;
        if (pb < 1 || pb >= NB) { *err = "quad model: a self-collision proxy must sit on a jointed body"; return DW_EINVAL; }
        const int l = Q->owner[pb];
        int k = 0;
        while (k < QMAX_OWN && Q->own_proxy[l][k] >= 0) ++k;
        if (k >= QMAX_OWN) { *err = "quad model: too many self-collision proxies on one lane's bodies"; return DW_EINVAL; }
        Q->own_proxy[l][k] = p2; local_of[p2] = k;
        Q->in[T - 1 - step_of[pb]][l].sc_mask |= 1 << k;
        if (add_gym(pb, d->sc_proxy[p2].gym) < 0) { *err = "quad model: proxy Gym body does not fit"; return DW_EINVAL; }
    }
    // detection rounds of the octet kernels (QHot::scround)
    int scround[QMAX_ROUNDS][8];
    int nround_class[3] = {0, 0, 0};
    {
        auto half_up = [](float v) -> int {      // smallest half-precision number >= v (v positive and in half's normal range), as its 16 bits
            unsigned int u; memcpy(&u, &v, 4);
            const int e = (int)((u >> 23) & 255) - 127 + 15;
            unsigned int m = (u & 0x7fffffu);
            unsigned int h = ((unsigned int)e << 10) | (m >> 13);
            if (m & 0x1fffu) h += 1;            // (a carry into the exponent is the next binade: still the smallest half above v)
            return (int)(h & 0xffffu);
        };
        for (int r = 0; r < QMAX_ROUNDS; ++r) for (int o = 0; o < 8; ++o) scround[r][o] = 127 << 6;
        int r0 = 0;
        for (int cls = 0; cls < 3; ++cls) {      // (class of a, class of b) = (0,0), (1,1), (1,0)
            int n = 0;
            for (int k = 0; k < d->num_sc_pairs; ++k) {
                int a = d->sc_pair[k][0], b2 = d->sc_pair[k][1];
                if ((a >> 3) < (b2 >> 3)) { const int t = a; a = b2; b2 = t; }
                const int c = (a >> 3) == (b2 >> 3) ? (a >> 3) : 2;
                if (c != cls) continue;
                const int r = r0 + n / 8, o = n % 8;
                if (r >= QMAX_ROUNDS) { *err = "quad model: self-collision pairs need more detection rounds than QMAX_ROUNDS"; return DW_EINVAL; }
                const float rr = d->sc_proxy[a].radius + d->sc_proxy[b2].radius, thr = 1.004f * rr * rr;
                if (!(thr > 1e-4f && thr < 6.0e4f)) { *err = "quad model: self-collision radii outside the range of the detection threshold"; return DW_EINVAL; }
                scround[r][o] = (a & 7) | ((b2 & 7) << 3) | (k << 6) | (half_up(thr) << 16);
                n += 1;
            }
            nround_class[cls] = (n + 7) / 8;
            r0 += nround_class[cls];
        }
        if (d->num_sc_pairs > 64 || Q->nprox > 16) { *err = "quad model: more than 64 self-collision pairs or 16 proxies"; return DW_EINVAL; }
    }
    // every Gym body must be reported by exactly one moving body (the kernels write, never accumulate, contact forces)
    for (int b = 1; b < NB; ++b) if (add_gym(b, dm->mv_gym[b]) < 0) { *err = "quad model: Gym body of a moving body does not fit"; return DW_EINVAL; }
    for (int k = 0; k < DW_NUM_INERT; ++k)
        if (dm->inert_mv[k] > 0 && add_gym(dm->inert_mv[k], dm->inert_gym[k]) < 0) { *err = "quad model: Gym body of an inertial record does not fit"; return DW_EINVAL; }
    {
        int seen[DW_NUM_BODIES] = {};
        seen[Q->base_gym] += 1;
        for (int s2 = 0; s2 < T; ++s2) for (int l = 0; l < 4; ++l) { const QInRec &r = Q->in[s2][l]; if (r.body >= 0) for (int i = 0; i < r.ngym; ++i) seen[r.gyms[i]] += 1; }
        for (int g = 0; g < DW_NUM_BODIES; ++g) if (seen[g] != 1) { *err = "quad model: a Gym body is reported by no moving body or by several"; return DW_EINVAL; }
    }
    // ---- the LDS-resident copy ----
    {
        QHot &H = Q->hot;
        auto fi = [](int v) { float f; memcpy(&f, &v, 4); return f; };
        for (int s2 = 0; s2 < QS_MAX; ++s2)
            for (int l = 0; l < 4; ++l) {
                const QFkRec &r = Q->fk[s2][l];
                float *o = H.fk[s2][l];
                for (int i = 0; i < 3; ++i) { o[i] = r.pos[i]; o[4 + i] = r.axis[i]; }
                o[3] = fi((r.body & 255) | ((r.psrc & 15) << 8) | ((r.flags & 3) << 12) | (((r.flags >> 8) & 255) << 16));
                o[7] = r.vmax; o[8] = r.qlo; o[9] = r.qhi;
                if (r.body >= 0 && r.psrc >= 2) H.fmask[s2] |= 1 << (r.psrc - 2);
                const QInRec &n = Q->in[s2][l];
                float *p = H.in[s2][l];
                p[0] = fi(((n.body + 1) & 255) | ((n.flags & 7) << 8) | ((n.nin & 3) << 12) | ((n.ngym & 3) << 14) | ((n.ngeom & 15) << 16) |
                          ((n.sc_mask & 255) << 24));
                p[1] = fi(n.body >= 0 ? n.gather : 0);
                p[2] = fi((n.gyms[0] & 255) | ((n.gyms[1] & 255) << 8) | ((n.gyms[2] & 255) << 16) | ((n.in0_gym & 255) << 24));
                p[3] = n.bound;
                prepared_inertia(n.in0_com, n.in0_mass, n.in0_I, p + 4);
                if (n.body >= 0 && n.gather) H.gany[s2] |= 1;
            }
        for (int s2 = 0; s2 < QS_MAX; ++s2) H.gany[s2] |= acc_step[s2] << 8;
        for (int i = 0; i < 3; ++i) H.base[i] = Q->base_com[i];
        H.base[3] = Q->base_mass;
        for (int i = 0; i < 6; ++i) H.base[4 + i] = Q->base_I[i];
        H.base[10] = fi(Q->base_gym); H.base[11] = fi(Q->base_ngeom); H.base[12] = Q->base_bound;
        H.misc[0] = Q->nsteps; H.misc[1] = Q->base_gather;
        H.misc[2] = Q->nprox | (nround_class[0] << 8) | (nround_class[1] << 12) | (nround_class[2] << 16); H.misc[3] = Q->npair << 8;
        {
            // slot cells: per lane the bodies in schedule order, one lane after the other (cell 0 = the base's four rows: scratch)
            int first[4] = {QS_MAX, QS_MAX, QS_MAX, QS_MAX}, last[4] = {-1, -1, -1, -1}, count[4] = {0, 0, 0, 0}, cellbase[4] = {0, 0, 0, 0};
            for (int b = 1; b < NB; ++b) {
                const int l = Q->owner[b], st = Q->step_of[b];
                if (st < first[l]) first[l] = st;
                if (st > last[l]) last[l] = st;
                count[l] += 1;
            }
            int next = 1;
            bool contiguous = true;
            for (int l = 0; l < 4; ++l) {
                if (count[l] == 0) continue;
                if (last[l] - first[l] + 1 != count[l]) contiguous = false;
                cellbase[l] = next - first[l];
                next += count[l];
            }
            if (accumulate && !contiguous) { *err = "octet kernels: a lane's bodies are not contiguous in the schedule"; return DW_EINVAL; }
            H.owner[0] = 0;
            for (int b = 1; b < NB; ++b) {
                const int l = Q->owner[b];
                const int cell = contiguous ? cellbase[l] + Q->step_of[b] : b;
                H.owner[b] = (unsigned char)((l << 6) | (cell & 63));
            }
            H.base[13] = fi((cellbase[0] & 255) | ((cellbase[1] & 255) << 8) | ((cellbase[2] & 255) << 16) | ((cellbase[3] & 255) << 24));
            {
                int startmask = 0;      // outward steps in which some lane starts a chain (parent = the base or another lane's body)
                for (int s2 = 0; s2 < Q->nsteps; ++s2) for (int l = 0; l < 4; ++l) if (Q->fk[s2][l].body >= 0 && Q->fk[s2][l].psrc != 0) startmask |= 1 << s2;
                H.base[15] = fi(startmask);
            }
            H.base[14] = fi(first[0] | (first[1] << 4) | (first[2] << 8) | (first[3] << 12) | (last[0] << 16) | (last[1] << 20) | (last[2] << 24) | (last[3] << 28));
        }
        for (int l = 0; l < 2; ++l) {
            int s2k = 0;
            for (int s2 = 0; s2 < Q->nsteps; ++s2) if (Q->in[s2][l].body >= 0 && Q->in[s2][l].nin > 1) s2k = s2;
            const QInRec &n = Q->in[s2k][l];
            prepared_inertia(n.in1_com, n.in1_mass, n.in1_I, H.in1[l]);
            H.in1[l][10] = fi(n.in1_gym >= 0 && n.in1_gym < DW_NUM_BODIES ? n.in1_gym : 0);
            H.in1[l][11] = fi(s2k);
        }
        for (int p2 = 0; p2 < Q->nprox; ++p2) {
            const DwCapsule &cp = d->sc_proxy[p2];
            for (int i = 0; i < 3; ++i) { H.prox[p2][i] = cp.p0[i]; H.prox[p2][4 + i] = cp.p1[i]; }
            H.prox[p2][3] = cp.radius;
            H.prox[p2][7] = fi((cp.moving & 255) | ((cp.gym & 255) << 8) | ((Q->owner[cp.moving] & 3) << 16) | ((local_of[p2] & 7) << 18));
        }
        for (int r = 0; r < QMAX_ROUNDS; ++r) for (int o = 0; o < 8; ++o) H.scround[r][o] = scround[r][o];
        for (int i = 0; i < 16; ++i) H.pairs[i] = 0;
        for (int i = 4; i < 8; ++i) { H.misc[i] = 0; H.pairmask_hi[i - 4] = 0; }
        for (int k = 0; k < d->num_sc_pairs; ++k) {
            const int a = d->sc_pair[k][0], b2 = d->sc_pair[k][1];
            H.pairs[k >> 2] |= (a | (b2 << 4)) << (8 * (k & 3));
            for (int side = 0; side < 2; ++side) {
                const int l = Q->owner[d->sc_proxy[side ? b2 : a].moving];
                if (k < 32) H.misc[4 + l] |= (int)(1u << k); else H.pairmask_hi[l] |= (int)(1u << (k - 32));
            }
        }
    }
    return DW_OK;
}

}  // namespace dwq
