// dw_oct.h -- one physics substep (stand-in for the reference's closed `gym.simulate`, call site
// tasks/dyros_dynamic_walk.py:525) in the OCTET layout: 8 lanes per env, 8 envs per wavefront, two wavefronts per
// workgroup that share one copy of the hot tables.  Same physics, same order of the contact iterations and the same
// written decisions as oracle/dw_physics.c (DESIGN.md "Physics model").
//
// Why octets.  A body-env slot is 64 bytes, so 160 KB of LDS hold 64 envs per CU whatever the lane mapping.  The quad
// kernels (4 lanes per env, 16 envs per wave) therefore run ONE wave per SIMD -- and one wave alone issues a vector
// instruction every 5.5 cycles where two waves together get one every 4.6 (tools/valu_issue.hip; plain fp32 instructions do
// not reach the 2 cycles of MI355X_MICROARCH.md's constants table), and nothing covers its LDS / memory round trips.  Here the same 64 envs per CU are 8 waves of 8 envs: two waves per SIMD, each with
// half the envs' joint-parallel work per lane and at most 256 registers, and the phases with two-way data parallelism
// inside an env (self-collision pairs, the 12 unit-wrench responses of the contact phase, the corner blocks) use the
// second quad of the octet for it.
//
// Lane l of a wave: env el = l >> 3, octet lane o = l & 7 = 4 h + j: limb j (as in dw_quad_model.h: 0 neck + left leg,
// 1 right leg, 2 waist + left arm, 3 right arm) and half h.  The DPP quads of a wave are the (env, half) groups, so every
// quad_perm exchange of the retired quad kernels means the same thing here; the two halves of a limb exchange with oct_xor4().  Where a
// phase has nothing to split, both halves run it redundantly (in a SIMD machine that costs nothing) and only half 0 has
// side effects in global memory.
#pragma once

#include "dw_limb.h"
#include "dw_bufg.h"

#if !defined(OCT_LPE)
#define OCT_LPE 8
#endif
#if OCT_LPE == 8
#define OCT_NS dwo
#elif OCT_LPE == 16
#define OCT_NS dwx
#else
#error "OCT_LPE must be 8 (octet layout) or 16 (hex layout)"
#endif

namespace OCT_NS {

using namespace dw;       // DevModel, PhysParams, small vector helpers
using dwq::F4; using dwq::mk4; using dwq::P2; using dwq::V6; using dwq::v6; using dwq::v6_zero; using dwq::v6_from; using dwq::v6_to; using dwq::v6_dot; using dwq::v6_axpy; using dwq::v6_scale; using dwq::v6_add; using dwq::ld4; using dwq::f2i; using dwq::QHot; using dwq::QuadModel; using dwq::QInRec;
using dwq::lane_id; using dwq::quad_bcast; using dwq::quad_xor1; using dwq::quad_xor2; using dwq::quad_xor1_hi; using dwq::oct_fetch; using dwq::half_bits_to_float; using dwq::quad_pair_lo; using dwq::quad_pair_hi; using dwq::oct_xor4; using dwq::oct_lo; using dwq::oct_hi; using dwq::oct_take_lo; using dwq::oct_take_hi; using dwq::rs_take; using dwq::rs_all; using dwq::hex_xor8; using dwq::quarter_take; using dwq::quarter0_all; using dwq::wave_any; using dwq::wave_ballot;
using dwq::wave_sync; using dwq::wave_sync_global; using dwq::atomic_add_u64; using dwq::rcp_fast; using dwq::sincos_fast; using dwq::qmul; using dwq::quad_bcast_arr; using dwq::quad_take_arr; using dwq::over_1n;
using dwq::geom_force; using dwq::rigid_inertia; using dwq::rigid_inertia_pre; using dwq::add_rigid; using dwq::seg_seg; using dwq::seg_dist2_fast; using dwq::capsule_pair;
using dwq::QS_MAX; using dwq::QMAX_OWN; using dwq::QMAX_GYM; using dwq::QMAX_GEOM;

// Index type of the per-env streams.  Every buffer of the table is addressed as uniform base + 32-bit element index, which the
// compiler turns into the scalar-base + 32-bit vector-offset form of the global instructions (no 64-bit address arithmetic in the
// vector pipe: 522 such instructions in the listing of round 4).  The largest stream is obs_history, 2 960 B per env (obs_buf: 1 948 B): the
// byte offset fits 32 bits up to 1.45 M envs per GPU; config.validate_cfg / dw_create refuse more than 2^20.
#if defined(OQ_IX64)
using OQ_IX = size_t;
#else
using OQ_IX = unsigned int;
#endif
// element i of a stream: the BYTE offset is formed in 32 bits and added to the (wave-uniform) base, which is what selects the
// scalar-base form `global_load v, v_offset, s[base:base+1]`; indexing a pointer with a 32-bit integer does not (the scaling by the
// element size is done after the extension to 64 bits: v_lshl_add_u64 / v_mad_u64_u32, the latter at a quarter of the issue rate)
// oq_at(base, i, k): element i + k with k a compile-time constant: the constant goes into the instruction's immediate offset (added to
// i in 32 bits it would not: unsigned addition may wrap, so the compiler has to keep a separate offset register per element).
template <class T> DQ_HD T *oq_ptr(T *base, OQ_IX i) { return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (OQ_IX)(i * (OQ_IX)sizeof(T))); }
template <class T> DQ_HD const T *oq_ptr(const T *base, OQ_IX i) { return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (OQ_IX)(i * (OQ_IX)sizeof(T))); }
template <class T> DQ_HD T &oq_at(T *base, OQ_IX i, int k = 0) { return oq_ptr(base, i)[k]; }
template <class T> DQ_HD const T &oq_at(const T *base, OQ_IX i, int k = 0) { return oq_ptr(base, i)[k]; }
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> DQ_HD T DW_GPTR *oq_ptr(T DW_GPTR *base, OQ_IX i) { return (T DW_GPTR *)((char DW_GPTR *)base + (OQ_IX)(i * (OQ_IX)sizeof(T))); }
template <class T> DQ_HD const T DW_GPTR *oq_ptr(const T DW_GPTR *base, OQ_IX i) { return (const T DW_GPTR *)((const char DW_GPTR *)base + (OQ_IX)(i * (OQ_IX)sizeof(T))); }
template <class T> DQ_HD T DW_GPTR &oq_at(T DW_GPTR *base, OQ_IX i, int k = 0) { return oq_ptr(base, i)[k]; }
template <class T> DQ_HD const T DW_GPTR &oq_at(const T DW_GPTR *base, OQ_IX i, int k = 0) { return oq_ptr(base, i)[k]; }
#endif
// i / D for a small non-negative item index (0 <= i < LIMIT): one full-rate 24-bit multiply and a shift instead of the compiler's
// v_mul_hi (quarter rate) sequence for a signed division by a constant.  M = ceil(2^16 / D); exact while i * (M D - 2^16) < 2^16
// (checked at compile time for LIMIT).
template <int D, int LIMIT = 1024> DQ_HD int oq_div(int i) {
    constexpr unsigned M = (65536u + D - 1) / D;
    static_assert((unsigned long long)(LIMIT - 1) * (M * D - 65536u) < 65536ull && (unsigned long long)(LIMIT - 1) * M < (1ull << 32), "oq_div: LIMIT too large for this divisor");
    return (int)((((unsigned)i & 0xffffffu) * M) >> 16);
}
// env index x row stride: both below 2^24 (num_envs <= 2^20), so the product is ONE full-rate v_mul_u32_u24 / v_mad_u32_u24 (a
// 32-bit v_mul_lo_u32 issues at a quarter of the rate)
DQ_HD OQ_IX oq_row(int stride, int env) { return ((OQ_IX)stride & 0xffffffu) * ((OQ_IX)env & 0xffffffu); }
// LPE = 8 is the layout this file describes.  OCT_LPE = 16 compiles the SAME source as the "hex" instantiation (namespace dwx, entry
// points dw_k_*_hex, dw_hex_kernels.hip) for launches of at most 4096 envs, where the octet layout leaves half of the SIMDs idle: 16
// lanes = one DPP row per env, four QUARTERS q of four limb lanes, 4 envs per wave, 1024 waves at 4096 envs = one on every SIMD.
// What is a map over bodies is split four ways instead of two (the inward pass maps steps s .. s + 3 per round; item loops have 3
// items per lane instead of 5; 16 proxies are one per lane), the chain recursions run mirrored in all quarters, and the contact
// phase runs as two mirrored octets (quarter pairs (0,1) and (2,3): h = q & 1), so its lane arithmetic is the octet's.
constexpr int LPE = OCT_LPE;         // lanes per env
constexpr int EPO = 64 / LPE;        // envs per wavefront
constexpr int NQ = LPE / 4;          // quads ("halves" / quarters) of an env
#if !defined(OCT_WPG)
#define OCT_WPG 2
#endif
constexpr int WPG = OCT_WPG;         // wavefronts per workgroup (they share the hot tables, nothing else)
// The articulated recursion of the inward pass with its spatial ROWS split between the two halves of a limb (octet layout; round 6):
// half 0 keeps the angular rows of IA / pA, half 1 the linear rows (dw_oct.h "inward pass").  In the hex instantiation quarters 0 and 1 of an env
// are that pair (quarters 2 and 3 run the same instructions on values nobody reads); -DOCT_NO_ROWSPLIT builds the mirrored form of round 5.
#if !defined(OCT_NO_ROWSPLIT)
#define OQ_ROWSPLIT 1
#else
#define OQ_ROWSPLIT 0
#endif
constexpr int SC_PARK_WORDS = QMAX_OWN * 6 + 2;      // per lane: PhysParams::sc_park (the wrenches of up to QMAX_OWN own proxies, their Gym bodies one byte each)
static_assert(QMAX_OWN <= 8, "the Gym bodies of a lane's own proxies travel in two words, one byte each");

// One wave's body slots: slot[body * 4 + row][position], 64 bytes per body and env.  A row is 8 envs x 16 B
// = 128 B, half the width of the LDS (64 banks x 4 B); the position code (pcode_cell below) rotates a limb's column by two per limb,
// so the four limbs of a 16-lane group (2 envs) read 8 different 16-byte columns (`flip`, a pairwise swap of the rows of odd limbs, is
// what round 3's code did instead and survives for the A/B builds).
constexpr int XROWS = LPE == 16 ? 12 : 0;          // (hex: the post phase's LDS image needs 124 words more than 33 bodies x 4 envs give)
constexpr int ROWB = EPO * 16;                      // bytes of a slot row
struct alignas(16) OSlots { F4 slot[NB * 4 + XROWS][EPO]; };
struct alignas(16) OLds {
    OSlots w[WPG];
    QHot   hot;
};
static_assert(sizeof(OLds) * (8 / WPG) <= 163840, "OLds: 8 waves per CU must fit 160 KB of LDS");
// (as byte offsets: even rows of a flipped limb lie one row up, odd rows one row down -- two lane-dependent bases, so that every
//  access is base + a compile-time offset and the compiler need not keep one address register per body and row)
// A body's slot is its CELL: cellbase[owner lane] + outward step (dw_quad_model.h), so a limb's bodies lie in schedule order and
// the chain passes, unrolled over the steps, address them as lane base + compile-time offset: no slot address depends on a
// table read.  OQ_SLOT(st, q, p): row q of the body that the limb of position code p visits in outward step st; item lanes
// and cross-limb reads reach a body through icode() (cell from the owner table) with st = 0.
struct OPos { int e, o; };
DQ_HD OPos pcode_cell(int el, int owner, int cell) {
    // Which column (and, with `flip`, which row of a pair) a limb of env `el` uses.  A/B of four codes on one box with the LDS counters
    // (profiles/r04_lds_position_code_ab.txt): every limb in column el ("naive") has 2.6 x the bank-conflict cycles and costs 1.5 % of
    // the step; the three rotations below differ by at most 21 % in conflict cycles and the shipped one is the fastest by 0.5 % (16384
    // envs) and 0.9 % (4096 envs) -- it needs no row swap, so both row bases of a limb are one register.
#if defined(OCT_PCODE_NAIVE)          // (A/B builds only)
    const int pos = el & 7, flip = 0;
#elif defined(OCT_PCODE_R3)           // (A/B builds only: the code of round 3 -- arms four columns from the legs, rows of odd limbs swapped pairwise)
    const int pos = (el + 4 * (owner >> 1)) & 7, flip = owner & 1;
#elif defined(OCT_PCODE_ROT1F)        // (A/B builds only: columns rotated by one per limb, rows of odd limbs swapped)
    const int pos = (el + owner) & 7, flip = owner & 1;
#else                                 // columns rotated by two per limb, no row swap
    const int pos = LPE == 8 ? ((el + 2 * owner) & 7) : ((el + owner) & 3), flip = 0;
#endif
    OPos p; p.e = cell * (4 * ROWB) + pos * 16 + flip * ROWB; p.o = cell * (4 * ROWB) + pos * 16 - flip * ROWB;
    return p;
}
DQ_HD OPos pcode(const QHot &H, int el, int owner) {          // a limb: cell = cellbase + step
    return pcode_cell(el, owner, (int)(signed char)((f2i(H.base[13]) >> (8 * owner)) & 255));
}
DQ_HD OPos icode(const QHot &H, int el, int b) {              // a body
    const int co = H.owner[b];
    return pcode_cell(el, co >> 6, co & 63);
}
// What a body's four slot rows hold between the inward pass and the end of the substep (round 6: spatial vectors as pairs, dw_limb.h V6 --
// a row's .xy and .zw are the register pairs of the packed instructions):
//   row 0 {S0 S1 S2 S3}   row 1 {S4 S5 1/D u}   row 2 {U0 U1 U2 U3}   row 3 {U4 U5 qd *}
// (S = joint axis as a spatial vector, angular part first; U = IA S).  Outward pass 2 puts the free joint velocity over qd and clears u, the
// impulse up-sweep puts its d where u was, the final pass leaves row 0 = {q_lo, qd_new, q_hi, *} for the integration.
#define OQ_SLOT(st, q, p) (*reinterpret_cast<F4 *>(reinterpret_cast<char *>(&L.slot[0][0]) + (((q) & 1) ? (p).o : (p).e) + ((st) * 4 + (q)) * ROWB))
#define OQ_LD(b, q, p) ldp(OQ_SLOT(b, q, p))
#define OQ_S6(r0, r1) v6((r0).x, (r0).y, (r0).z, (r0).w, (r1).x, (r1).y)          // S from rows 0, 1 / U from rows 2, 3
// the loops over the schedule steps stay loops: unrolled, one substep is 100 KB of straight-line code that every wave streams
// through the 64 KB instruction cache (measured: +5 % step time)
#if defined(__HIPCC__) && defined(OCT_CHAIN_UNROLL)
#define DQ_PRAGMA_(x) _Pragma(#x)
#define DQ_PRAGMA(x) DQ_PRAGMA_(x)
#define DQ_ROLLED DQ_PRAGMA(clang loop unroll_count(OCT_CHAIN_UNROLL))
#elif defined(__HIPCC__) && !defined(OCT_UNROLLED)
#define DQ_ROLLED _Pragma("clang loop unroll(disable)")
#else
#define DQ_ROLLED
#endif
// 16-byte LDS loads that stay 16 bytes wide.  A load whose .w is unused is narrowed to ds_read_b96 (twice the LDS cycles of
// ds_read_b128, MI355X_MICROARCH.md); dw_limb.h's ld4() prevents that with an opaque touch after EVERY load, which also makes
// the wave wait for every load by itself.  Here loads are plain and ONE touch of the unused .w components follows a group of
// them: the group is in flight together and waited for once.
#if defined(__HIPCC__)
#define OQ_KEEP1(a) asm volatile("" : "+v"((a).w))
#define OQ_KEEP2(a, b) asm volatile("" : "+v"((a).w), "+v"((b).w))
#define OQ_KEEP3(a, b, c) asm volatile("" : "+v"((a).w), "+v"((b).w), "+v"((c).w))
#else
#define OQ_KEEP1(a) ((void)0)
#define OQ_KEEP2(a, b) ((void)0)
#define OQ_KEEP3(a, b, c) ((void)0)
#endif
DQ_HD F4 ldp(const F4 &p) { return p; }

// copies the hot tables from the device-resident model into LDS.  Both waves of the workgroup copy all of it (identical
// bytes), so neither has to wait for the other: no workgroup barrier anywhere in these kernels.
DQ_HD void stage_hot(QHot &HW, const QuadModel &QM) {
    const int l = lane_id();
    const F4 *src = reinterpret_cast<const F4 *>(&QM.hot);
    F4 *dst = reinterpret_cast<F4 *>(&HW);
    constexpr int NQ = (int)(sizeof(QHot) / 16), IT = (NQ + 63) / 64;
    // every piece requested before the first one is stored: as a loop this was one memory round trip per 1 KB (the store of
    // an iteration waits for its load, and the next load is issued after the store)
    float tx[IT], ty[IT], tz[IT], tw[IT];
    DQ_UNROLL for (int k = 0; k < IT; ++k) {
        const int i = l + 64 * k;
        const F4 v = src[i < NQ ? i : 0];
        tx[k] = v.x; ty[k] = v.y; tz[k] = v.z; tw[k] = v.w;
    }
    DQ_UNROLL for (int k = 0; k < IT; ++k) { const int i = l + 64 * k; if (i < NQ) dst[i] = mk4(tx[k], ty[k], tz[k], tw[k]); }
    wave_sync();
}

// What a lane keeps in registers across the phases of a step.
struct OLane {
    int   lane, o, j, h, q, prim, el, env, valid;     // octet lane (l & 7), limb, half (q & 1), quad of the env (0 .. NQ - 1), q == 0, env within the wave, global env (clamped)
    OPos  pos;                               // position code of my limb's slots
    int   wave;                              // wave index in the launch (wave-uniform)
    float root[13];
    float mu;
    float zbound;                            // height field: no point of the field within the robot's reach is higher (dw_physics.h terrain_bound)
    float footF[3];                          // non-sole contact force on my foot's sole Gym body (lanes 0, 1)
    int   coll;                              // last substep: one of my non-sole Gym bodies reports more than 1 N (termination)
    float footT[3];                          // last substep: net contact force on my sole body (lanes 0, 1)
    int   stamp_base;                        // profiling builds only
};

// add_rigid of dw_limb.h in two parts: IA += rigid inertia [[Ao, H], [H', m 1]], H = skew(ho) ...
DQ_HD void add_rigid_inertia(float *IA, const float *Ao, const float *ho, float mass) {
    DQ_UNROLL for (int r = 0; r < 3; ++r)
        DQ_UNROLL for (int c = r; c < 3; ++c) IA[sym6(r, c)] += dwq::ao(Ao, r, c);
    IA[sym6(0, 4)] += -ho[2]; IA[sym6(0, 5)] += ho[1];
    IA[sym6(1, 3)] += ho[2];  IA[sym6(1, 5)] += -ho[0];
    IA[sym6(2, 3)] += -ho[1]; IA[sym6(2, 4)] += ho[0];
    IA[sym6(3, 3)] += mass; IA[sym6(4, 4)] += mass; IA[sym6(5, 5)] += mass;
}
// ... and the gyroscopic bias pv = v x* (I v)
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

// ------------------------------------------------------------------------------------------------
// The substep.  On entry every body's slot holds quad 0 = {q, qd, tt, dd} with tt = tau - damping * qd and
// dd = armature + dt * damping (the caller's prologue), X.root the base state, and the hot tables are staged (stage_hot).  The
// warm-start impulses of the sole corners live in the task record (DW_ES_WARM): read after the inward pass, written back after
// the contact solve.  On exit: slot quad 0 = {q, qd, *, *} of the new state, X.root updated; with `last`,
// the net contact forces of the substep are written to B.contact_forces.  push: world x/y force on the base COM.
// ------------------------------------------------------------------------------------------------
struct FkHot { float pos[3], axis[3], vmax, qlo, qhi; int body, psrc, flags, scm; };
DQ_HD FkHot fk_hot(const QHot &H, int s, int j) {
    const F4 *r = reinterpret_cast<const F4 *>(H.fk[s][j]);
    const F4 a = ldp(r[0]), b = ldp(r[1]), c = ldp(r[2]);
    FkHot h;
    h.pos[0] = a.x; h.pos[1] = a.y; h.pos[2] = a.z;
    const int bits = f2i(a.w);
    h.body = (bits & 255) == 255 ? -1 : (bits & 255);
    h.psrc = (bits >> 8) & 15; h.flags = (bits >> 12) & 3; h.scm = (bits >> 16) & 255;
    h.axis[0] = b.x; h.axis[1] = b.y; h.axis[2] = b.z; h.vmax = b.w;
    h.qlo = c.x; h.qhi = c.y;
    return h;
}

DQ_HD int sched_body(const QHot &H, int s, int j) {
    const int bits = f2i(H.fk[s][j][3]);
    return (bits & 255) == 255 ? -1 : (bits & 255);
}

template <bool TERRAIN>
DQ_HD void oct_substep(OSlots &L, const QHot &H, const QuadModel &QM, const DevModel &M, const PhysParams &P, OLane &X, const OBuf &B,
                        float push_x, float push_y, bool last) {
    const float dt = P.dt, inv_dt = 1.0f / P.dt;
    const int j = X.j;
    constexpr int T = QS_MAX;         // (the octet kernels are built for a schedule of exactly QS_MAX steps: checked at dw_create)
    const int SB = X.stamp_base; (void)SB;
    DQ_STAMP(B, SB + 0);
    const int e = X.env;
    const bool wr = X.valid && X.prim;          // global side effects: quad 0 of the env only (the others mirror it)
    const OQ_IX ms_row = oq_row(DW_NUM_BODIES, e);          // my env's row of mass_scale
    const int first_j = (f2i(H.base[14]) >> (4 * j)) & 15, last_j = (f2i(H.base[14]) >> (16 + 4 * j)) & 15;      // my limb's steps
    // The three outward passes read their slot rows in EVERY step, also where a lane's limb has no body (its chain starts later or
    // has ended): such a lane reads the nearest body of its own limb -- the step clamped to the limb's range -- computes on it and
    // stores nothing.  Its running state is garbage then, which nobody sees: a limb's own chains are contiguous in the schedule, the
    // running state is rewritten where a chain starts, and another lane fetches a running state only in the step right after its
    // owner worked on the parent body (dw_quad_model.h).  (The loads were conditional before, onto registers zeroed in every step: 16
    // moves per step and pass.)
    auto own_step = [&](int s) { return s < first_j ? first_j : (s > last_j ? last_j : s); };

    // the mass scale of the second (welded) inertial record of my sole body: requested here, consumed by the inward pass (the
    // record itself sits in the hot tables)
    const float ms1 = oq_at(B.mass_scale, ms_row + f2i(H.in1[j & 1][10]));
    if (TERRAIN) X.zbound = dw::terrain_bound(P, X.root[0], X.root[1]);          // (requested here, first used in the inward pass)
    // @phase fk
    // ---- base kinematics (every lane of the quad, redundantly) ----
    float qn[4], R0k[9], ww[3], vo[3], bcom[3];
    {
        float (&R0)[9] = R0k;
        const float qx = X.root[3], qy = X.root[4], qz = X.root[5], qw = X.root[6];
        const float ninv = dw::rsqrt_nr(qx * qx + qy * qy + qz * qz + qw * qw);
        qn[0] = qx * ninv; qn[1] = qy * ninv; qn[2] = qz * ninv; qn[3] = qw * ninv;
        quat_to_mat(qn, R0);
        DQ_UNROLL for (int i = 0; i < 3; ++i) { ww[i] = X.root[10 + i]; vo[i] = X.root[7 + i]; bcom[i] = H.base[i]; }
        if (P.vel_at_com) {
            float rc[3], t[3];
            m3v(R0, bcom, rc);
            cross3(ww, rc, t);
            vo[0] -= t[0]; vo[1] -= t[1]; vo[2] -= t[2];
        }
    }

    DQ_STAMP(B, SB + 1);
    // ---- outward pass 1: kinematics.  Running parent state: quaternion, rotation, origin, twist.  In most steps no lane starts
    //      a chain, every lane's parent is the body of its previous step, and the running state is updated unconditionally
    //      (lanes that idle compute on zeros and store nothing); in the three steps in which some chain starts -- at the base or
    //      below another lane's body, whose running state is fetched from that lane's registers -- a wave-uniform bit of the
    //      schedule puts the fetch in front.  ONE copy of the step's arithmetic follows either way: written as two call sites
    //      (one per branch) it cost 35 more register copies per step at the join and 1 % of the whole kernel. ----
    const int startmask = f2i(H.base[15]);
    {
        float qr[4] = {0, 0, 0, 1}, Rr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, xr_[3] = {0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
        auto fk_body = [&](int s, const FkHot &rc, const F4 &in) {          // the arithmetic of a step and its stores
            const int b = rc.body;
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
            if (b >= 0) {
                OQ_SLOT(s, 0, X.pos) = mk4(qr[0], qr[1], qr[2], qr[3]);
                OQ_SLOT(s, 1, X.pos) = mk4(x[0], x[1], x[2], in.y);
                OQ_SLOT(s, 2, X.pos) = mk4(vr[0], vr[1], vr[2], in.z);
                OQ_SLOT(s, 3, X.pos) = mk4(vr[3], vr[4], vr[5], in.w);
                if (rc.flags & 2) {
                    // pose of the sole body for the contact phase, in the four slot rows of body 0 (the base has no slot; the
                    // step kernel's per-env scratch there is dead once the substeps run): foot f = j in rows 2 f, 2 f + 1
                    L.slot[2 * j][X.el] = mk4(qr[0], qr[1], qr[2], qr[3]);
                    L.slot[2 * j + 1][X.el] = mk4(x[0], x[1], x[2], 0.0f);
                }
            }
        };
        DQ_ROLLED for (int s = 0; s < T; ++s) {          /*@trip:11*/
            const FkHot rc = fk_hot(H, s, j);
            const int b = rc.body, psrc = rc.psrc;
            // (both halves of a limb walk it: every lane reads its slot row before any lane overwrites it)
            const F4 in = OQ_LD(own_step(s), 0, X.pos);            // {q, qd, tt, dd}
            if ((startmask >> s) & 1) {          /*@prob:0.27*/
                // limbs that start below another lane's body take that lane's running state (still in its registers), limbs that
                // start at the base the base's: selects, no branches
                const int fm = H.fmask[s];
                if (fm) {          /*@prob:0.33*/
                    for (int xl = 0; xl < 4; ++xl)          /*@trip:1*/
                        if ((fm >> xl) & 1) {
                            const bool take = b >= 0 && psrc == 2 + xl;
                            quad_take_arr(xl, take, qr); quad_take_arr(xl, take, Rr); quad_take_arr(xl, take, xr_); quad_take_arr(xl, take, vr);
                        }
                }
                const bool fb = b >= 0 && psrc == 1;
                DQ_UNROLL for (int i = 0; i < 4; ++i) qr[i] = fb ? qn[i] : qr[i];
                DQ_UNROLL for (int i = 0; i < 9; ++i) Rr[i] = fb ? R0k[i] : Rr[i];
                DQ_UNROLL for (int i = 0; i < 3; ++i) { xr_[i] = fb ? 0.0f : xr_[i]; vr[i] = fb ? ww[i] : vr[i]; vr[3 + i] = fb ? vo[i] : vr[3 + i]; }
            }
            wave_sync();
            fk_body(s, rc, in);
        }
    }
    wave_sync();

    DQ_STAMP(B, SB + 2);
    // @phase self_collision
    // ---- self-collision: capsule proxies (legs, arms, torso), pairs from the model.  Detection is pair-parallel: octet
    //      lane o tests pairs o, o + 8, o + 16, o + 24 -- both proxies' axes from their bodies' slots, the division-free
    //      conservative distance of dw_limb.h -- and the touching pairs of the env are ORed into a mask over the octet.  The
    //      common case is "nothing touches": then that is all.  Resolution, if any env of the wave has a touching pair: the
    //      lane that owns a proxy's body recomputes the proxy's touching pairs exactly and keeps the wrench (both sides of a
    //      pair compute the same force from the same data: no hand-over between lanes; the two halves of a limb do the same). ----
    bool sc_any = false;
    // (the wrenches of my own proxies, found by the resolution, wait for the inward pass in global memory: touching pairs are
    //  rare, and 25 registers held through the inward pass for them are what the two-waves-per-SIMD budget cannot afford)
    float *park = P.sc_park + ((size_t)X.wave * 64 + X.lane) * SC_PARK_WORDS;          // (touched only when some env of the wave has a touching pair)
    const int npair = (H.misc[3] >> 8) & 255;
    auto proxy_bits = [&](int p) { return f2i(H.prox[p][7]); };
    auto proxy_ends = [&](int p, float *p0w, float *p1w) {       // end points of proxy p in the common frame, from its body's slot
        const F4 *pr = reinterpret_cast<const F4 *>(H.prox[p]);
        F4 c0 = ldp(pr[0]), c1 = ldp(pr[1]);
        const int bits = f2i(c1.w);
        const int bp = bits & 255;
        const OPos posp = icode(H, X.el, bp);
        F4 q4 = OQ_LD(0, 0, posp), x4 = OQ_LD(0, 1, posp);
        OQ_KEEP2(c0, x4);
        const float qb[4] = {q4.x, q4.y, q4.z, q4.w}, l0[3] = {c0.x, c0.y, c0.z}, l1[3] = {c1.x, c1.y, c1.z};
        float Rb[9], t0[3], t1[3];
        quat_to_mat(qb, Rb);
        m3v(Rb, l0, t0);
        m3v(Rb, l1, t1);
        p0w[0] = x4.x + t0[0]; p0w[1] = x4.y + t0[1]; p0w[2] = x4.z + t0[2];
        p1w[0] = x4.x + t1[0]; p1w[1] = x4.y + t1[1]; p1w[2] = x4.z + t1[2];
    };
#if defined(OCT_ABL_SC)
    if (false) {
#else
    if (P.self_collision && npair > 0) {
#endif
        DQ_STAMP(B, 51);
        unsigned long long hits = 0ull;          // bit k: pair k may touch (up to DW_MAX_SC_PAIRS = 64 pairs)
        {
            // the axes, built ONCE per proxy: octet lane o has proxy o (class 0) and proxy o + 8 (class 1) -- origin point and direction in
            // the common frame, from the proxy's body slot.  (Until round 5 every pair test rebuilt both of its proxies: 64 builds for 15.)
            const int nprox = H.misc[2] & 255;
            float E[2][6];
            DQ_UNROLL for (int c = 0; c < 2; ++c) {
                const int pp = X.o + 8 * c, p = pp < nprox ? pp : nprox - 1;          // (a lane past the last proxy builds the last one again; nobody fetches it)
                const F4 *pr = reinterpret_cast<const F4 *>(H.prox[p]);
                F4 c0 = ldp(pr[0]), c1 = ldp(pr[1]);
                const OPos posp = icode(H, X.el, f2i(c1.w) & 255);
                F4 q4 = OQ_LD(0, 0, posp), x4 = OQ_LD(0, 1, posp);
                OQ_KEEP2(c0, x4);
                const float qb[4] = {q4.x, q4.y, q4.z, q4.w}, l0[3] = {c0.x, c0.y, c0.z}, dl[3] = {c1.x - c0.x, c1.y - c0.y, c1.z - c0.z};
                float Rb[9], t0[3];
                quat_to_mat(qb, Rb);
                m3v(Rb, l0, t0);
                m3v(Rb, dl, &E[c][3]);
                E[c][0] = x4.x + t0[0]; E[c][1] = x4.y + t0[1]; E[c][2] = x4.z + t0[2];
            }
            // the rounds: lane o tests pair scround[r][o], the axes fetched from the lanes that built them (conservative: the least
            // distance of the axes against a threshold rounded up; the force uses the blended points, never closer)
            const int nr0 = (H.misc[2] >> 8) & 15, nr1 = (H.misc[2] >> 12) & 15, nr2 = (H.misc[2] >> 16) & 15;
            // (hex layout: the env's two octets both hold all the axes and take alternate rounds of a class)
            constexpr int NOCT = LPE / 8;
            const int oc = (X.lane >> 3) & (NOCT - 1);
            auto round = [&](int r0, int k, int nr, const float (&Ea)[6], const float (&Eb)[6]) {
                const int kk = k * NOCT + oc;
                const bool live = kk < nr;
                const int w = H.scround[live ? r0 + kk : r0][X.o];
                const int la = w & 7, lb = (w >> 3) & 7, pid = (w >> 6) & 127;
                float a[6], b[6];
                DQ_UNROLL for (int i = 0; i < 6; ++i) { a[i] = oct_fetch(Ea[i], la); b[i] = oct_fetch(Eb[i], lb); }
                const float rv[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
                if (live && pid != 127 && seg_dist2_fast(&a[3], &b[3], rv) < half_bits_to_float((w >> 16) & 0xffff)) hits |= 1ull << pid;
            };
            for (int k = 0; k * NOCT < nr0; ++k) round(0, k, nr0, E[0], E[0]);          /*@trip:2*/
            for (int k = 0; k * NOCT < nr1; ++k) round(nr0, k, nr1, E[1], E[1]);          /*@trip:2*/
            for (int k = 0; k * NOCT < nr2; ++k) round(nr0 + nr1, k, nr2, E[1], E[0]);          /*@trip:1*/
        }
        {   // the env's mask: OR over the octet (bit patterns through the DPP moves)
            int m[2] = {(int)(unsigned int)hits, (int)(unsigned int)(hits >> 32)};
            DQ_UNROLL for (int w = 0; w < 2; ++w) {
                m[w] |= f2i(quad_xor1(__builtin_bit_cast(float, m[w])));
                m[w] |= f2i(quad_xor2(__builtin_bit_cast(float, m[w])));
                m[w] |= f2i(oct_xor4(__builtin_bit_cast(float, m[w])));
                if (LPE == 16) m[w] |= f2i(hex_xor8(__builtin_bit_cast(float, m[w])));
            }
            hits = (unsigned long long)(unsigned int)m[0] | ((unsigned long long)(unsigned int)m[1] << 32);
        }
        sc_any = wave_any(hits != 0);
        DQ_STAMP(B, 52);
#if defined(DQ_STAMPS) && defined(__HIPCC__)
        if (blockIdx.x == 0 && threadIdx.x == 0) OQ_COLD(gate_acc)[200 + 53] = sc_any;
#endif
        if (sc_any) {          /*@prob:0*/
            float scW[QMAX_OWN][6];              // wrench (common frame) on my k-th own proxy: [0..2] moment, [3..5] force
            unsigned long long scGym = 0ull;     // Gym body of my k-th own proxy, one byte each (filled for the loaded ones)
            DQ_UNROLL for (int p = 0; p < QMAX_OWN; ++p) { DQ_UNROLL for (int i = 0; i < 6; ++i) scW[p][i] = 0.0f; }
            // every lane works through the touching pairs that involve one of its bodies
            unsigned long long mine = hits & ((unsigned long long)(unsigned int)H.misc[4 + j] | ((unsigned long long)(unsigned int)H.pairmask_hi[j] << 32));
            while (wave_any(mine != 0)) {
                if (mine != 0) {
                    const int pid = __builtin_ctzll(mine);
                    mine &= mine - 1;
                    const int pr = (H.pairs[pid >> 2] >> (8 * (pid & 3))) & 255, pa = pr & 15, pbx = pr >> 4;
                    const int bita = proxy_bits(pa), bitb = proxy_bits(pbx);
                    float a0[3], a1[3], b0[3], b1[3];
                    proxy_ends(pa, a0, a1);
                    proxy_ends(pbx, b0, b1);
                    const int ba = bita & 255, bb = bitb & 255;
                    const OPos posa = icode(H, X.el, ba), posb = icode(H, X.el, bb);
                    F4 va2 = OQ_LD(0, 2, posa), va3 = OQ_LD(0, 3, posa), vb2 = OQ_LD(0, 2, posb), vb3 = OQ_LD(0, 3, posb);
                    OQ_KEEP2(va2, va3); OQ_KEEP2(vb2, vb3);
                    const float va[6] = {va2.x, va2.y, va2.z, va3.x, va3.y, va3.z}, vb[6] = {vb2.x, vb2.y, vb2.z, vb3.x, vb3.y, vb3.z};
                    float F[3], ca[3], cb[3];
                    if (capsule_pair(a0, a1, H.prox[pa][3], b0, b1, H.prox[pbx][3], va, vb, P, F, ca, cb)) {
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
                            scGym |= (unsigned long long)gy << (8 * k);
                        }
                    }
                }
            }
            DQ_UNROLL for (int p = 0; p < QMAX_OWN; ++p) DQ_UNROLL for (int i = 0; i < 6; ++i) park[6 * p + i] = scW[p][i];
            park[6 * QMAX_OWN] = __builtin_bit_cast(float, (int)(unsigned int)scGym);
            park[6 * QMAX_OWN + 1] = __builtin_bit_cast(float, (int)(unsigned int)(scGym >> 32));
            wave_sync_global();
        }
    }
    wave_sync();      // the leg lanes read each other's slots above; the inward pass below overwrites them

    DQ_STAMP(B, SB + 3);
    // @phase inward
    // ---- inward pass: articulated inertias and bias forces, in reverse schedule order, TWO steps per round.  What a body
    //      contributes by itself -- joint subspace, rigid inertia about O, gyroscopic bias, external forces, velocity-product
    //      acceleration: more than half of a step's arithmetic, and no recursion in it -- is a MAP over bodies: half 0 of a limb
    //      maps the body of step 2 r, half 1 the body of step 2 r + 1, at the same time.  The recursion proper (add into the
    //      running inertia, U = IA S, rank-1 downdate, bias) then runs for step 2 r on half 0's result, takes half 1's result
    //      over (31 words, one DPP move each) and runs for step 2 r + 1.  Half 1 executes the recursion's instructions on
    //      values nobody reads (its stores are masked); it gets the base's inertia back before the base solve. ----
#if OQ_ROWSPLIT
    //      ROW SPLIT (octet layout, round 6).  The recursion itself is sequential along the limb, so until round 5 half 1 repeated half
    //      0's instructions on dead values.  Now the two halves of a limb hold different ROWS of every spatial quantity: half 0 the
    //      angular rows of IA = [[A, H], [H', M]] and of pA, half 1 the linear rows.  Stored alike in both: Dm = the symmetric diagonal
    //      block of my rows (A | M, 6 words), Om = their off-diagonal block (H | H', 9 words), pO = my three rows of pA; of a
    //      6-vector x a lane uses own(x) = its three rows and oth(x) = the other three.  Then U_own = Dm own(S) + Om oth(S) is 18
    //      products per lane instead of 36, the rank-1 downdate 15 instead of 21, IA c 18 instead of 36; D = S'U and u = tt - S'pA are
    //      sums over the pair (oct_xor4), and U's other three rows come across the same way.  The map of a step is made by one half
    //      and handed to BOTH in own / oth form (oct_take_lo / _hi: one bank-masked move per word, the making half keeps its own).
    float Dm[6], Om[9], pO[3];
    DQ_UNROLL for (int i = 0; i < 6; ++i) Dm[i] = 0.0f;
    DQ_UNROLL for (int i = 0; i < 9; ++i) Om[i] = 0.0f;
    DQ_UNROLL for (int i = 0; i < 3; ++i) pO[i] = 0.0f;
    const float hsgn = (X.q & 1) ? -1.0f : 1.0f;          // my off-diagonal block of a rigid inertia is skew(ho) (half 0) or its transpose (half 1)
#else
    float IA[21], pA[6];          // running reflected inertia / bias (no lane parks a second one: build_quadmodel(accumulate))
    DQ_UNROLL for (int i = 0; i < 21; ++i) IA[i] = 0.0f;
    DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] = 0.0f;
#endif
    X.footF[0] = X.footF[1] = X.footF[2] = 0.0f;
    const int my_sole_gym = (j == 0) ? M.left_foot_gym : (j == 1 ? M.right_foot_gym : -1);
    struct BodyMap { float Ao[6], ho[3], mass, pv[6], S[6], cb[6], tt, dd, qd; };       // 31 words
    auto step_clamped = [&](int s) { return s < T ? s : T - 1; };
    // the one per-env global value a map needs (the mass scale of the body's Gym body) is requested a round ahead
    float ms_next = oq_at(B.mass_scale, ms_row + ((f2i(H.in[step_clamped(X.q)][j][2]) >> 24) & 255));
#if defined(DQ_STAMPS) && defined(__HIPCC__)
    long long tq_map = 0, tq_rec = 0, tq_t0 = 0;
#define OQ_TICK() (tq_t0 = (long long)__builtin_readcyclecounter())
#define OQ_TOCK(acc) (acc += (long long)__builtin_readcyclecounter() - tq_t0)
#else
#define OQ_TICK() ((void)0)
#define OQ_TOCK(acc) ((void)0)
#endif
#if defined(OCT_ABL_INWARD)
    DQ_ROLLED for (int s = 0; s < 0; s += 2) {
#else
    DQ_ROLLED for (int s = 0; s < T; s += NQ) {          /*@trip:6*/
#endif
        OQ_TICK();
#if defined(DQ_STAMPS_INWARD)
        if (SB == 1) DQ_STAMP(B, 42 + s);
#endif
        BodyMap Mb;
        {   // ---- map: my body of this round (step s + h) ----
            const int sm = step_clamped(s + X.q);
            const F4 *hr = reinterpret_cast<const F4 *>(H.in[sm][j]);
            const F4 h0 = ldp(hr[0]), h1 = ldp(hr[1]), h2 = ldp(hr[2]), h3 = ldp(hr[3]);
            const int bits = f2i(h0.x);
            const int b = (s + X.q < T) ? (bits & 255) - 1 : -1;
            const int nin = (bits >> 12) & 3, ngym = (bits >> 14) & 3, ngeom = (bits >> 16) & 15, scm = (bits >> 24) & 255;
            const int gymbits = f2i(h0.z);
            const float ms0 = ms_next;
            ms_next = oq_at(B.mass_scale, ms_row + ((f2i(H.in[step_clamped(s + NQ + X.q)][j][2]) >> 24) & 255));
            // (the slot rows are requested together with the table record, not after it: a lane that idles in this step reads the
            //  nearest body of its own limb instead -- the step index clamped to the limb's range -- and nothing is done with it)
            int som = T - 1 - s - X.q;
            som = som < first_j ? first_j : (som > last_j ? last_j : som);
            const F4 s0 = OQ_LD(som, 0, X.pos), s1 = OQ_LD(som, 1, X.pos), s2 = OQ_LD(som, 2, X.pos), s3 = OQ_LD(som, 3, X.pos);
            F4 ax4 = ldp(reinterpret_cast<const F4 *>(H.fk[T - 1 - sm][j])[1]);
            OQ_KEEP1(ax4);
            const float axis[3] = {ax4.x, ax4.y, ax4.z};
            const float qb[4] = {s0.x, s0.y, s0.z, s0.w}, x[3] = {s1.x, s1.y, s1.z}, v[6] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z};
            Mb.qd = s1.w; Mb.tt = s2.w; Mb.dd = s3.w;
            float R[9];
            quat_to_mat(qb, R);
            m3v(R, axis, Mb.S);
            cross3(x, Mb.S, Mb.S + 3);
            {
                // (the records arrive prepared: mass * com, mass, inertia about the body origin -- dw_quad_model.h prepared_inertia.
                //  The two sole bodies carry a second, welded record; it is read by every lane -- three LDS reads -- so that there is ONE
                //  copy of the rotation code below, and added by the lanes that have it)
                const float hm0[3] = {h1.x, h1.y, h1.z}, A0[6] = {h2.x, h2.y, h2.z, h2.w, h3.x, h3.y};
                const F4 *ir = reinterpret_cast<const F4 *>(H.in1[j & 1]);
                const F4 i0 = ldp(ir[0]), i1 = ldp(ir[1]), i2 = ldp(ir[2]);
                const float hm1[3] = {i0.x, i0.y, i0.z}, A1[6] = {i1.x, i1.y, i1.z, i1.w, i2.x, i2.y};
                rigid_inertia_pre(nin, hm0, h1.w, A0, ms0, hm1, i0.w, A1, ms1, R, x, Mb.Ao, Mb.ho, &Mb.mass);
            }
            rigid_bias(Mb.Ao, Mb.ho, Mb.mass, v, Mb.pv);
            {
                float m[6];
                DQ_UNROLL for (int i = 0; i < 6; ++i) m[i] = Mb.S[i] * Mb.qd;
                dw::motion_cross(v, m, Mb.cb);
            }
            if (b >= 0) {
                // external forces: ground penalty of the non-sole primitives, self-collision; per Gym body for the report
                float cf[QMAX_GYM][3];
                DQ_UNROLL for (int t = 0; t < QMAX_GYM; ++t) cf[t][0] = cf[t][1] = cf[t][2] = 0.0f;
                bool near_ground = ngeom > 0 && (X.root[2] + x[2] < h0.w);
                if (TERRAIN) near_ground = ngeom > 0 && (X.root[2] + x[2] - X.zbound < h0.w);
#if defined(DQ_KO_GEOM) || defined(OCT_ABL_GEOM)          // (timing experiment only)
                near_ground = false;
#endif
                if (near_ground) {          /*@prob:0.17*/
                    const QInRec &rc = QM.in[sm][j];
                    for (int k = 0; k < ngeom; ++k) {
                        float F[3], xr[3];
                        geom_force<TERRAIN>(M.geoms[rc.geom[k]], P, R, x, v, X.root[0], X.root[1], X.root[2], X.mu, F, xr);
                        if (F[0] != 0.0f || F[1] != 0.0f || F[2] != 0.0f) {
                            float nb[3];
                            cross3(xr, F, nb);
                            DQ_UNROLL for (int i = 0; i < 3; ++i) { Mb.pv[i] -= nb[i]; Mb.pv[3 + i] -= F[i]; }
                            const int t = (rc.geom_slot >> (2 * k)) & 3;
                            DQ_UNROLL for (int tt2 = 0; tt2 < QMAX_GYM; ++tt2)
                                if (tt2 == t) { cf[tt2][0] += F[0]; cf[tt2][1] += F[1]; cf[tt2][2] += F[2]; }
                        }
                    }
                }
                if (sc_any && scm) {          /*@prob:0*/
                    float scW[QMAX_OWN][6];
                    DQ_UNROLL for (int p = 0; p < QMAX_OWN; ++p) DQ_UNROLL for (int i = 0; i < 6; ++i) scW[p][i] = park[6 * p + i];
                    const unsigned long long scGym = (unsigned long long)(unsigned int)f2i(park[6 * QMAX_OWN]) | ((unsigned long long)(unsigned int)f2i(park[6 * QMAX_OWN + 1]) << 32);
                    DQ_UNROLL for (int p = 0; p < QMAX_OWN; ++p)
                        if ((scm >> p) & 1) {
                            DQ_UNROLL for (int i = 0; i < 6; ++i) Mb.pv[i] -= scW[p][i];
                            const int gy = (int)((scGym >> (8 * p)) & 255ull);     // (0 where unloaded: adds nothing)
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
                                    const OQ_IX dst = (oq_row(DW_NUM_BODIES, e) + gy) * 3;
                                    oq_at(B.contact_forces, dst, 0) = cf[t][0]; oq_at(B.contact_forces, dst, 1) = cf[t][1]; oq_at(B.contact_forces, dst, 2) = cf[t][2];
                                }
                            }
                        }
                }
            }
        }
        wave_sync();          // every map has read its slot rows before the recursion overwrites any
        OQ_TOCK(tq_map); OQ_TICK();
#if OQ_ROWSPLIT
        // ---- recursion: step s + t2 on the map quarter t2 made, rows split over the working pair ----
        DQ_UNROLL for (int t2 = 0; t2 < NQ; ++t2) {
            const int sr = s + t2;
            if (sr >= T) break;
            // the step's map in own / oth form: made by quarter t2, taken by the pair's two lanes (rs_take: a for half 0, b for half 1)
            float So[3], St[3], co[3], ct[3], pvo[3], Da[6], hs[3], tt_, dd_, qd_;
            auto take = [&](float a, float b) {
                return t2 == 0 ? rs_take<LPE, 0>(a, b) : (t2 == 1 ? rs_take<LPE, 1>(a, b) : (t2 == 2 ? rs_take<LPE, (LPE == 16 ? 2 : 0)>(a, b) : rs_take<LPE, (LPE == 16 ? 3 : 0)>(a, b)));
            };
            DQ_UNROLL for (int i = 0; i < 3; ++i) {
                So[i] = take(Mb.S[i], Mb.S[3 + i]);   St[i] = take(Mb.S[3 + i], Mb.S[i]);
                co[i] = take(Mb.cb[i], Mb.cb[3 + i]); ct[i] = take(Mb.cb[3 + i], Mb.cb[i]);
                pvo[i] = take(Mb.pv[i], Mb.pv[3 + i]);
                hs[i] = hsgn * take(Mb.ho[i], Mb.ho[i]);
            }
            {   // (my diagonal block of the rigid inertia: Ao in half 0, mass * 1 in half 1)
                const float zero = 0.0f;
                Da[0] = take(Mb.Ao[0], Mb.mass); Da[3] = take(Mb.Ao[3], Mb.mass); Da[5] = take(Mb.Ao[5], Mb.mass);
                Da[1] = take(Mb.Ao[1], zero); Da[2] = take(Mb.Ao[2], zero); Da[4] = take(Mb.Ao[4], zero);
            }
            tt_ = take(Mb.tt, Mb.tt); dd_ = take(Mb.dd, Mb.dd); qd_ = take(Mb.qd, Mb.qd);
            const int bits = f2i(H.in[sr][j][0]);
            const int b = (bits & 255) - 1;
            const int flags = b >= 0 ? ((bits >> 8) & 7) : 0;
            const int gw = H.gany[sr];
            if (gw >> 8) {  /*@prob:0.09*/      // a finished chain joins the finished chain of an idle lane (same parent) before its lane starts afresh
                const int src = (gw >> 9) & 3, dst = (gw >> 11) & 3;
                float tD[6], tO[9], tp[3];
                quad_bcast_arr(src, Dm, tD);
                quad_bcast_arr(src, Om, tO);
                quad_bcast_arr(src, pO, tp);
                if (j == dst) {
                    DQ_UNROLL for (int i = 0; i < 6; ++i) Dm[i] += tD[i];
                    DQ_UNROLL for (int i = 0; i < 9; ++i) Om[i] += tO[i];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) pO[i] += tp[i];
                }
            }
            if (flags & 1) {
                DQ_UNROLL for (int i = 0; i < 6; ++i) Dm[i] = 0.0f;
                DQ_UNROLL for (int i = 0; i < 9; ++i) Om[i] = 0.0f;
                DQ_UNROLL for (int i = 0; i < 3; ++i) pO[i] = 0.0f;
            }
            // gathers (wave-uniform per step): child chains that ended on other lanes
            if (gw & 1) {          /*@prob:0.09*/
                const int g0 = f2i(H.in[sr][0][1]), g1 = f2i(H.in[sr][1][1]), g2 = f2i(H.in[sr][2][1]), g3 = f2i(H.in[sr][3][1]);
                const int mine = f2i(H.in[sr][j][1]);
                DQ_UNROLL for (int src = 0; src < 4; ++src) {
                    const int code = src | 8;
                    bool used = false, want = false;
                    DQ_UNROLL for (int k = 0; k < 3; ++k) {
                        used = used || (((g0 >> (4 * k)) & 15) == code) || (((g1 >> (4 * k)) & 15) == code) ||
                               (((g2 >> (4 * k)) & 15) == code) || (((g3 >> (4 * k)) & 15) == code);
                        want = want || (((mine >> (4 * k)) & 15) == code);
                    }
                    if (used) {
                        auto from = [&](float x) { return src == 0 ? quad_bcast<0>(x) : (src == 1 ? quad_bcast<1>(x) : (src == 2 ? quad_bcast<2>(x) : quad_bcast<3>(x))); };
                        DQ_UNROLL for (int i = 0; i < 6; ++i) { const float t = from(Dm[i]); if (want) Dm[i] += t; }
                        DQ_UNROLL for (int i = 0; i < 9; ++i) { const float t = from(Om[i]); if (want) Om[i] += t; }
                        DQ_UNROLL for (int i = 0; i < 3; ++i) { const float t = from(pO[i]); if (want) pO[i] += t; }
                    }
                }
            }
            // my rows of U = IA S, my shares of D = S'U and of S'pA
            float Uo[3] = {0.0f, 0.0f, 0.0f}, dpart = 0.0f, upart = 0.0f;
            if (b >= 0) {
                DQ_UNROLL for (int i = 0; i < 6; ++i) Dm[i] += Da[i];
                Om[1] -= hs[2]; Om[2] += hs[1];
                Om[3] += hs[2]; Om[5] -= hs[0];
                Om[6] -= hs[1]; Om[7] += hs[0];
                DQ_UNROLL for (int i = 0; i < 3; ++i) pO[i] += pvo[i];
                DQ_UNROLL for (int r = 0; r < 3; ++r) {
                    float acc = 0.0f;
                    DQ_UNROLL for (int c = 0; c < 3; ++c) acc += dwq::ao(Dm, r, c) * So[c];
                    DQ_UNROLL for (int c = 0; c < 3; ++c) acc += Om[3 * r + c] * St[c];
                    Uo[r] = acc;
                }
                dpart = So[0] * Uo[0] + So[1] * Uo[1] + So[2] * Uo[2];
                upart = So[0] * pO[0] + So[1] * pO[1] + So[2] * pO[2];
            }
            // (lanes l and l ^ 4 are the two halves of one limb: both work or both idle)
            const float dsum = dpart + oct_xor4(dpart), usum = upart + oct_xor4(upart);
            const float Ut[3] = {oct_xor4(Uo[0]), oct_xor4(Uo[1]), oct_xor4(Uo[2])};
            if (b >= 0) {
                const float D = dsum + dd_;
                const float Dinv = dw::rcp_nr(D);
                const float u = tt_ - usum;
                DQ_UNROLL for (int r = 0; r < 3; ++r) {
                    const float urd = Uo[r] * Dinv;
                    DQ_UNROLL for (int c = r; c < 3; ++c) Dm[r == 0 ? c : (r == 1 ? 2 + c : 5)] -= urd * Uo[c];
                    DQ_UNROLL for (int c = 0; c < 3; ++c) Om[3 * r + c] -= urd * Ut[c];
                }
                const float ud = u * Dinv;
                float pa[3];
                DQ_UNROLL for (int r = 0; r < 3; ++r) {
                    float acc = pO[r] + Uo[r] * ud;
                    DQ_UNROLL for (int c = 0; c < 3; ++c) acc += dwq::ao(Dm, r, c) * co[c];
                    DQ_UNROLL for (int c = 0; c < 3; ++c) acc += Om[3 * r + c] * ct[c];
                    pa[r] = acc;
                }
                DQ_UNROLL for (int r = 0; r < 3; ++r) pO[r] = pa[r];
                if (X.prim) {          // (half 0: own = angular, oth = linear)
                    OQ_SLOT(T - 1 - sr, 0, X.pos) = mk4(So[0], So[1], So[2], St[0]);
                    OQ_SLOT(T - 1 - sr, 1, X.pos) = mk4(St[1], St[2], Dinv, u);
                    OQ_SLOT(T - 1 - sr, 2, X.pos) = mk4(Uo[0], Uo[1], Uo[2], Ut[0]);
                    OQ_SLOT(T - 1 - sr, 3, X.pos) = mk4(Ut[1], Ut[2], qd_, 0.0f);
                }
            }
        }
#else
        // ---- recursion: step s on half 0's map, then step s + 1 on half 1's ----
        BodyMap Mq = Mb;          // (hex layout: quad 0 takes the maps of quads 1, 2, 3 in turn from the lanes that made them)
        DQ_UNROLL for (int t2 = 0; t2 < NQ; ++t2) {
            const int sr = s + t2;
            if (sr >= T) break;
            if (LPE == 8 && t2 == 1) {        // half 1's map result to half 0 (the high quads keep their own)
                DQ_UNROLL for (int i = 0; i < 6; ++i) { Mb.Ao[i] = oct_hi(Mb.Ao[i]); Mb.pv[i] = oct_hi(Mb.pv[i]); Mb.S[i] = oct_hi(Mb.S[i]); Mb.cb[i] = oct_hi(Mb.cb[i]); }
                DQ_UNROLL for (int i = 0; i < 3; ++i) Mb.ho[i] = oct_hi(Mb.ho[i]);
                Mb.mass = oct_hi(Mb.mass); Mb.tt = oct_hi(Mb.tt); Mb.dd = oct_hi(Mb.dd); Mb.qd = oct_hi(Mb.qd);
            }
            if (LPE == 16 && t2 >= 1) {
                auto take = [&](float x) { return t2 == 1 ? quarter_take<1>(x) : (t2 == 2 ? quarter_take<2>(x) : quarter_take<3>(x)); };
                DQ_UNROLL for (int i = 0; i < 6; ++i) { Mb.Ao[i] = take(Mq.Ao[i]); Mb.pv[i] = take(Mq.pv[i]); Mb.S[i] = take(Mq.S[i]); Mb.cb[i] = take(Mq.cb[i]); }
                DQ_UNROLL for (int i = 0; i < 3; ++i) Mb.ho[i] = take(Mq.ho[i]);
                Mb.mass = take(Mq.mass); Mb.tt = take(Mq.tt); Mb.dd = take(Mq.dd); Mb.qd = take(Mq.qd);
            }
            const int bits = f2i(H.in[sr][j][0]);
            const int b = (bits & 255) - 1;
            const int flags = b >= 0 ? ((bits >> 8) & 7) : 0;
            const int gw = H.gany[sr];
            if (gw >> 8) {  /*@prob:0.09*/      // a finished chain joins the finished chain of an idle lane (same parent) before its lane starts afresh
                const int src = (gw >> 9) & 3, dst = (gw >> 11) & 3;
                float tI[21], tp[6];
                quad_bcast_arr(src, IA, tI);
                quad_bcast_arr(src, pA, tp);
                if (j == dst) {
                    DQ_UNROLL for (int i = 0; i < 21; ++i) IA[i] += tI[i];
                    DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] += tp[i];
                }
            }
            if (flags & 1) {
                DQ_UNROLL for (int i = 0; i < 21; ++i) IA[i] = 0.0f;
                DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] = 0.0f;
            }
            // gathers (wave-uniform per step): child chains that ended on other lanes
            if (gw & 1) {          /*@prob:0.09*/
                const int g0 = f2i(H.in[sr][0][1]), g1 = f2i(H.in[sr][1][1]), g2 = f2i(H.in[sr][2][1]), g3 = f2i(H.in[sr][3][1]);
                const int mine = f2i(H.in[sr][j][1]);
                DQ_UNROLL for (int src = 0; src < 4; ++src) {
                    const int code = src | 8;
                    bool used = false, want = false;
                    DQ_UNROLL for (int k = 0; k < 3; ++k) {
                        used = used || (((g0 >> (4 * k)) & 15) == code) || (((g1 >> (4 * k)) & 15) == code) ||
                               (((g2 >> (4 * k)) & 15) == code) || (((g3 >> (4 * k)) & 15) == code);
                        want = want || (((mine >> (4 * k)) & 15) == code);
                    }
                    if (used) {
                        DQ_UNROLL for (int i = 0; i < 21; ++i) {
                            const float t = src == 0 ? quad_bcast<0>(IA[i]) : (src == 1 ? quad_bcast<1>(IA[i]) : (src == 2 ? quad_bcast<2>(IA[i]) : quad_bcast<3>(IA[i])));
                            if (want) IA[i] += t;
                        }
                        DQ_UNROLL for (int i = 0; i < 6; ++i) {
                            const float t = src == 0 ? quad_bcast<0>(pA[i]) : (src == 1 ? quad_bcast<1>(pA[i]) : (src == 2 ? quad_bcast<2>(pA[i]) : quad_bcast<3>(pA[i])));
                            if (want) pA[i] += t;
                        }
                    }
                }
            }
            if (b >= 0) {
                add_rigid_inertia(IA, Mb.Ao, Mb.ho, Mb.mass);
                DQ_UNROLL for (int i = 0; i < 6; ++i) pA[i] += Mb.pv[i];
                const float *S = Mb.S;
                float U[6];
                DQ_UNROLL for (int r = 0; r < 6; ++r) {
                    float acc = 0.0f;
                    DQ_UNROLL for (int c = 0; c < 6; ++c) acc += IA[sym6(r, c)] * S[c];
                    U[r] = acc;
                }
                const float D = dot6(S, U) + Mb.dd;
                const float Dinv = dw::rcp_nr(D);
                const float u = Mb.tt - dot6(S, pA);
                DQ_UNROLL for (int r = 0; r < 6; ++r) {
                    const float urd = U[r] * Dinv;
                    DQ_UNROLL for (int c = r; c < 6; ++c) IA[sym6(r, c)] -= urd * U[c];
                }
                const float ud = u * Dinv;
                float pa[6];
                DQ_UNROLL for (int r = 0; r < 6; ++r) {
                    float acc = pA[r] + U[r] * ud;
                    DQ_UNROLL for (int c = 0; c < 6; ++c) acc += IA[sym6(r, c)] * Mb.cb[c];
                    pa[r] = acc;
                }
                DQ_UNROLL for (int r = 0; r < 6; ++r) pA[r] = pa[r];
                if (X.prim) {
                    OQ_SLOT(T - 1 - sr, 0, X.pos) = mk4(S[0], S[1], S[2], S[3]);
                    OQ_SLOT(T - 1 - sr, 1, X.pos) = mk4(S[4], S[5], Dinv, u);
                    OQ_SLOT(T - 1 - sr, 2, X.pos) = mk4(U[0], U[1], U[2], U[3]);
                    OQ_SLOT(T - 1 - sr, 3, X.pos) = mk4(U[4], U[5], Mb.qd, 0.0f);
                }
            }
        }
#endif
        OQ_TOCK(tq_rec);
    }
    wave_sync();
#if defined(DQ_STAMPS) && defined(__HIPCC__)
    if (blockIdx.x == 0 && threadIdx.x == 0 && SB == 1) { OQ_COLD(gate_acc)[200 + 32] = tq_map; OQ_COLD(gate_acc)[200 + 33] = tq_rec; }
#endif
    // (the sole body's non-sole contact force was found by whichever half mapped it)
    DQ_UNROLL for (int i = 0; i < 3; ++i) { X.footF[i] += oct_xor4(X.footF[i]); if (LPE == 16) X.footF[i] += hex_xor8(X.footF[i]); }

    DQ_STAMP(B, SB + 4);
    // @phase base_solve
    // ---- base: gather the chains below the root, own inertia, external forces, inverse ----
    float Minv[21], a0[6];            // inverse of the base's articulated inertia, symmetric storage (sym6)
    {
        float I0[21], p0[6];
        const int g = H.misc[1];
#if OQ_ROWSPLIT
        {   // every half gathers its rows of the chains below the root; the two halves then put the whole matrix together
            float Dg[6] = {0, 0, 0, 0, 0, 0}, Og[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pg[3] = {0, 0, 0};
            DQ_UNROLL for (int src = 0; src < 4; ++src) {
                const int code = src | 8;
                const bool used = ((g & 15) == code) || (((g >> 4) & 15) == code) || (((g >> 8) & 15) == code) || (((g >> 12) & 15) == code);
                if (used) {
                    auto from = [&](float x) { return src == 0 ? quad_bcast<0>(x) : (src == 1 ? quad_bcast<1>(x) : (src == 2 ? quad_bcast<2>(x) : quad_bcast<3>(x))); };
                    DQ_UNROLL for (int i = 0; i < 6; ++i) Dg[i] += from(Dm[i]);
                    DQ_UNROLL for (int i = 0; i < 9; ++i) Og[i] += from(Om[i]);
                    DQ_UNROLL for (int i = 0; i < 3; ++i) pg[i] += from(pO[i]);
                }
            }
            DQ_UNROLL for (int r = 0; r < 3; ++r) {
                DQ_UNROLL for (int c = r; c < 3; ++c) {
                    const int k = r == 0 ? c : (r == 1 ? 2 + c : 5);
                    I0[sym6(r, c)] = rs_all<LPE, 0>(Dg[k]);
                    I0[sym6(3 + r, 3 + c)] = rs_all<LPE, 1>(Dg[k]);
                }
                DQ_UNROLL for (int c = 0; c < 3; ++c) I0[sym6(r, 3 + c)] = rs_all<LPE, 0>(Og[3 * r + c]);
                p0[r] = rs_all<LPE, 0>(pg[r]); p0[3 + r] = rs_all<LPE, 1>(pg[r]);
            }
        }
#else
        DQ_UNROLL for (int i = 0; i < 21; ++i) I0[i] = 0.0f;
        DQ_UNROLL for (int i = 0; i < 6; ++i) p0[i] = 0.0f;
        DQ_UNROLL for (int src = 0; src < 4; ++src) {
            const int code = src | 8;
            const bool used = ((g & 15) == code) || (((g >> 4) & 15) == code) || (((g >> 8) & 15) == code) || (((g >> 12) & 15) == code);
            if (used) {
                DQ_UNROLL for (int i = 0; i < 21; ++i)
                    I0[i] += src == 0 ? quad_bcast<0>(IA[i]) : (src == 1 ? quad_bcast<1>(IA[i]) : (src == 2 ? quad_bcast<2>(IA[i]) : quad_bcast<3>(IA[i])));
                DQ_UNROLL for (int i = 0; i < 6; ++i)
                    p0[i] += src == 0 ? quad_bcast<0>(pA[i]) : (src == 1 ? quad_bcast<1>(pA[i]) : (src == 2 ? quad_bcast<2>(pA[i]) : quad_bcast<3>(pA[i])));
            }
        }
#endif
        // (the base's rotation matrix again from its quaternion: kept from the top of the substep it would cost 9 registers
        //  through the inward pass)
        float R0[9];
        {
            float qo4[4] = {qn[0], qn[1], qn[2], qn[3]};
            DQ_OPAQUE(qo4[0]);
            quat_to_mat(qo4, R0);
        }
#if !OQ_ROWSPLIT
        // (the recursion of the inward pass is valid in half 0 only: half 1 takes the gathered inertia over)
        DQ_UNROLL for (int i = 0; i < 21; ++i) I0[i] = LPE == 8 ? oct_lo(I0[i]) : quarter0_all(I0[i]);
        DQ_UNROLL for (int i = 0; i < 6; ++i) p0[i] = LPE == 8 ? oct_lo(p0[i]) : quarter0_all(p0[i]);
#endif
        const float v0[6] = {ww[0], ww[1], ww[2], vo[0], vo[1], vo[2]}, x0[3] = {0, 0, 0};
        float Ao[6], ho[3], mass;
        const int base_gym = f2i(H.base[10]), base_ngeom = f2i(H.base[11]);
        const float ms = oq_at(B.mass_scale, ms_row + base_gym);
        {
            const float bI[6] = {H.base[4], H.base[5], H.base[6], H.base[7], H.base[8], H.base[9]};
            rigid_inertia(1, bcom, H.base[3], bI, ms, bcom, 0.0f, bI, 0.0f, R0, x0, Ao, ho, &mass);
        }
        add_rigid(I0, p0, Ao, ho, mass, v0);
        float cfb[3] = {0, 0, 0};
        bool near_ground = base_ngeom > 0 && (X.root[2] < H.base[12]);
        if (TERRAIN) near_ground = base_ngeom > 0 && (X.root[2] - X.zbound < H.base[12]);
        if (near_ground) {          /*@prob:0*/
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
            if (wr) {
                const OQ_IX dst = (oq_row(DW_NUM_BODIES, e) + base_gym) * 3;
                oq_at(B.contact_forces, dst, 0) = cfb[0]; oq_at(B.contact_forces, dst, 1) = cfb[1]; oq_at(B.contact_forces, dst, 2) = cfb[2];
            }
        }
        {   // push on the base COM
            const float Fw[3] = {push_x, push_y, 0.0f};
            float xc[3], nb[3];
            m3v(R0, bcom, xc);
            cross3(xc, Fw, nb);
            DQ_UNROLL for (int i = 0; i < 3; ++i) { p0[i] -= nb[i]; p0[3 + i] -= Fw[i]; }
        }
        // Cholesky I0 = L L', Minv by six pairs of triangular solves (oracle/dw_physics.c)
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

    DQ_STAMP(B, SB + 5);
    // @phase outward2
    // the warm-start impulses of my foot's corners (previous substep) from the task record: requested here, a whole
    // outward pass before the contact solve uses them
    float warm[12];
    DQ_UNROLL for (int i = 0; i < 12; ++i) warm[i] = 0.0f;
    wave_sync_global();          // (also: the previous substep of this launch stored the impulses)
    if (B.env_state) {           // (dw_simulate may run without a task record: the solve then starts from zero)
        const OQ_IX wsrc = oq_row(DW_ES_WORDS, e) + DW_ES_WARM + 12 * (j & 1);
        DQ_UNROLL for (int i = 0; i < 12; ++i) warm[i] = oq_at(B.env_state, wsrc, i);
    }
    // ---- outward pass 2: accelerations; free joint velocities qdf = qd + dt qdd into the slot (lean / chain-start forms as in
    //      pass 1) ----
    {
        float ar[6] = {0, 0, 0, 0, 0, 0}, vr[6] = {0, 0, 0, 0, 0, 0};
        DQ_ROLLED for (int s = 0; s < T; ++s) {          /*@trip:11*/
            const int bits = f2i(H.fk[s][j][3]);
            const int b = (bits & 255) == 255 ? -1 : (bits & 255), psrc = (bits >> 8) & 15;
            const int so = own_step(s);
            F4 s0 = OQ_LD(so, 0, X.pos), s1 = OQ_LD(so, 1, X.pos), s2 = OQ_LD(so, 2, X.pos), s3 = OQ_LD(so, 3, X.pos);
            OQ_KEEP1(s3);
            if ((startmask >> s) & 1) {          /*@prob:0.27*/
                const int fm = H.fmask[s];
                if (fm) {          /*@prob:0.33*/
                    for (int xl = 0; xl < 4; ++xl)          /*@trip:1*/
                        if ((fm >> xl) & 1) {
                            const bool take = b >= 0 && psrc == 2 + xl;
                            quad_take_arr(xl, take, ar); quad_take_arr(xl, take, vr);
                        }
                }
                const bool fb = b >= 0 && psrc == 1;
                DQ_UNROLL for (int i = 0; i < 3; ++i) { ar[i] = fb ? a0[i] : ar[i]; ar[3 + i] = fb ? a0[3 + i] : ar[3 + i]; vr[i] = fb ? ww[i] : vr[i]; vr[3 + i] = fb ? vo[i] : vr[3 + i]; }
            }
            wave_sync();
            {
                const V6 S = OQ_S6(s0, s1), U = OQ_S6(s2, s3);
                const float Dinv = s1.z, u = s1.w, qd = s3.z;
                V6 aV = v6_from(ar), vV = v6_from(vr);
                const V6 mV = v6_scale(S, qd);
                float m[6], c[6];
                v6_to(mV, m);
                dw::motion_cross(vr, m, c);                  // parent twist x S qd  (= body twist x S qd)
                v6_add(aV, v6_from(c)); v6_add(vV, mV);
                const float qdd = (u - v6_dot(U, aV)) * Dinv;
                v6_axpy(aV, S, qdd);
                v6_to(aV, ar); v6_to(vV, vr);
                if (b >= 0) {
                    OQ_SLOT(s, 3, X.pos) = mk4(s3.x, s3.y, qd + dt * qdd, 0.0f);      // free velocity
                    OQ_SLOT(s, 1, X.pos) = mk4(s1.x, s1.y, Dinv, 0.0f);                // u is dead: the word becomes the impulse-sweep d
                }
            }
        }
    }
    wave_sync();
    DQ_STAMP(B, SB + 6);
    // @phase contact_setup
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
        {
            F4 fq4 = ldp(L.slot[2 * f][X.el]), fx4 = ldp(L.slot[2 * f + 1][X.el]);
            OQ_KEEP1(fx4);
            const float fq[4] = {fq4.x, fq4.y, fq4.z, fq4.w};
            quat_to_mat(fq, fR);
            fx[0] = fx4.x; fx[1] = fx4.y; fx[2] = fx4.z;
        }
        DQ_UNROLL for (int k = 0; k < 4; ++k) {
            m3v(fR, M.foot_pos[4 * f + k], rk[k]);
            DQ_UNROLL for (int i = 0; i < 3; ++i) rk[k][i] += fx[i];
        }
        float hh4[4];
        if (TERRAIN) {
#if defined(OCT_SAMPLE_ALL)          // (A/B builds only: every lane of the foot samples all four corners, as before round 5)
            DQ_UNROLL for (int k = 0; k < 4; ++k) dw::terrain_sample(P, X.root[0] + rk[k][0], X.root[1] + rk[k][1], &hh4[k], frame[k]);
#else
            // The four lanes that work on a foot (2 parts x 2 halves) each sample the height field under ONE corner -- corner part + 2 h --
            // and hand height and frame round: 4 fetches of 2 bytes and one interpolation per lane instead of 16 and four, for two DPP moves
            // per word (the values are the ones every lane computed for itself before: same arithmetic on the same corner)
            const int ko = part + 2 * X.h;
            float ro[2], ho, fo[9];
            DQ_UNROLL for (int i = 0; i < 2; ++i) ro[i] = ko == 0 ? rk[0][i] : (ko == 1 ? rk[1][i] : (ko == 2 ? rk[2][i] : rk[3][i]));
            dw::terrain_sample(P, X.root[0] + ro[0], X.root[1] + ro[1], &ho, fo);
            DQ_UNROLL for (int k = 0; k < 4; ++k) {
                auto from = [&](float x) { const float t = (k & 1) ? quad_pair_hi(x) : quad_pair_lo(x); return (k >> 1) ? oct_hi(t) : oct_lo(t); };
                hh4[k] = from(ho);
                DQ_UNROLL for (int i = 0; i < 9; ++i) frame[k][i] = from(fo[i]);
            }
#endif
        }
        DQ_UNROLL for (int k = 0; k < 4; ++k) {
            float phi = X.root[2] + rk[k][2];
            if (TERRAIN) phi = (phi - hh4[k]) * frame[k][8];
            else { DQ_UNROLL for (int i = 0; i < 9; ++i) frame[k][i] = (i % 4 == 0) ? 1.0f : 0.0f; }
            act[k] = phi < P.contact_offset;
            vminr[k] = phi >= 0 ? -phi * inv_dt : fminf(P.erp * (-phi) * inv_dt, P.max_depen);
        }
    }
    const bool any_active = wave_any(act[0] | act[1] | act[2] | act[3]);
    float dqb[6] = {0, 0, 0, 0, 0, 0};              // base velocity jump
    float Pk[4][3];
    DQ_UNROLL for (int k = 0; k < 4; ++k) DQ_UNROLL for (int i = 0; i < 3; ++i) Pk[k][i] = act[k] ? warm[3 * k + i] : 0.0f;

#if defined(OCT_ABL_CONTACT)
    if (false) {
#else
    if (any_active) {
#endif
        // ---- free twist of foot f: base + sum over the leg of S qdf (leg lane), shared with the partner ----
        float twf[6];
        {
            V6 accV = v6(wwf[0], wwf[1], wwf[2], vowf[0], vowf[1], vowf[2]);
            const OPos posf = pcode(H, X.el, f);
            DQ_UNROLL for (int i = 1; i <= 6; ++i) {
                const int b = T - 7 + i;          // (the legs are right-aligned in the schedule: hip .. sole = steps T - 6 .. T - 1)
                F4 s0 = OQ_LD(b, 0, posf), s1 = OQ_LD(b, 1, posf), s3 = OQ_LD(b, 3, posf);
                OQ_KEEP2(s1, s3);
                v6_axpy(accV, OQ_S6(s0, s1), s3.z);
            }
            v6_to(accV, twf);
        }
        DQ_STAMP(B, SB + 7);
        // @phase contact_W
        // ---- my block of W: the response of ONE foot (leg g: half 0 of the octet takes its own foot f, half 1 the other foot)
        //      to my 3 unit wrenches (components 3 part .. 3 part + 2) on foot f.  Up leg f: d = -S'p, p += U d / D (both
        //      halves); base: dv = -Minv p; down leg g: qdd = (d - U'dv) / D, dv += S qdd ----
        const int g = X.h ^ f;
        const int mrow_id = X.o < 6 ? X.o : X.o - 6;
        float mrow[6];
        V6 Wg[3];
        {
            V6 dp[3];
            float dc[3][6];
            DQ_UNROLL for (int c = 0; c < 3; ++c) { float t[6]; DQ_UNROLL for (int i = 0; i < 6; ++i) t[i] = (i == 3 * part + c) ? -1.0f : 0.0f; dp[c] = v6_from(t); }
            const OPos posf = pcode(H, X.el, f);
            DQ_UNROLL for (int i = 6; i >= 1; --i) {
                DQ_SCHED_FENCE();
                const int b = T - 7 + i;
                F4 s0 = OQ_LD(b, 0, posf), s1 = OQ_LD(b, 1, posf), s2 = OQ_LD(b, 2, posf), s3 = OQ_LD(b, 3, posf);
                OQ_KEEP2(s1, s3);
                const V6 S = OQ_S6(s0, s1), U = OQ_S6(s2, s3);
                DQ_UNROLL for (int c = 0; c < 3; ++c) {
                    const float d = -v6_dot(S, dp[c]);
                    dc[c][i - 1] = (g == f) ? d : 0.0f;          // (the other leg carries no wrench of its own)
                    v6_axpy(dp[c], U, d * s1.z);
                }
            }
            DQ_UNROLL for (int c = 0; c < 3; ++c) {
                float dpc[6], wr_[6];
                v6_to(dp[c], dpc);
                DQ_UNROLL for (int r = 0; r < 6; ++r) {
                    float acc = 0.0f;
                    DQ_UNROLL for (int k = 0; k < 6; ++k) acc -= Minv[sym6(r, k)] * dpc[k];
                    wr_[r] = acc;
                }
                Wg[c] = v6_from(wr_);
            }
            DQ_UNROLL for (int c = 0; c < 6; ++c) {
                float v = Minv[sym6(0, c)];
                DQ_UNROLL for (int r = 1; r < 6; ++r) v = (mrow_id == r) ? Minv[sym6(r, c)] : v;
                mrow[c] = v;
            }
            const OPos posg = pcode(H, X.el, g);
            DQ_UNROLL for (int i = 1; i <= 6; ++i) {
                DQ_SCHED_FENCE();
                const int b = T - 7 + i;
                F4 s0 = OQ_LD(b, 0, posg), s1 = OQ_LD(b, 1, posg), s2 = OQ_LD(b, 2, posg), s3 = OQ_LD(b, 3, posg);
                OQ_KEEP2(s1, s3);
                const V6 S = OQ_S6(s0, s1), U = OQ_S6(s2, s3);
                DQ_UNROLL for (int c = 0; c < 3; ++c) {
                    const float ua = v6_dot(U, Wg[c]);
                    v6_axpy(Wg[c], S, (dc[c][i - 1] - ua) * s1.z);
                }
            }
        }
        DQ_STAMP(B, SB + 8);
        // @phase contact_Akk
        // ---- 3x3 diagonal blocks of the Delassus matrix of my foot's corners: A_kk = J_k W_ff J_k' (frame-projected on
        //      terrain); what the solver needs of them: the three diagonal inverses and the couplings zx, zy, xy.  W_ff is
        //      spread over the half-0 lanes of the foot (rows 3 part ..): they compute, half 1 takes the results over ----
        float invd[4][3], cpl[4][3];
        {
            // Per corner k, J_k = [-skew(r) | 1]: A_kk = J_k W_ff J_k' = sum over the two row blocks of W_ff.  The lane with the
            // angular rows (part 0) and the lane with the linear rows (part 1) each form T = W_rows J_k' (row t: w_lin + w_ang x r)
            // and their share of J_k T -- part 0: every column of T crossed with r, part 1: T itself -- and add the shares over
            // the pair (l ^ 2): the 6 x 6 block W_ff is never assembled in one lane (it cost 36 registers at the kernel's peak).
            const float rreg = 1.0f / (1.0f + P.cfm);
            DQ_UNROLL for (int k = 0; k < 4; ++k) {
                DQ_SCHED_FENCE();
                const float *r = rk[k];
                float Tm[3][3];
                DQ_UNROLL for (int t = 0; t < 3; ++t) {
                    float w[6];
                    v6_to(Wg[t], w);
                    Tm[t][0] = w[3] + (w[1] * r[2] - w[2] * r[1]);
                    Tm[t][1] = w[4] + (w[2] * r[0] - w[0] * r[2]);
                    Tm[t][2] = w[5] + (w[0] * r[1] - w[1] * r[0]);
                }
                float Ak[3][3];
                DQ_UNROLL for (int c = 0; c < 3; ++c) {
                    // column c of T crossed with r (part 0) or taken as it is (part 1)
                    const float x0 = Tm[1][c] * r[2] - Tm[2][c] * r[1], x1 = Tm[2][c] * r[0] - Tm[0][c] * r[2], x2 = Tm[0][c] * r[1] - Tm[1][c] * r[0];
                    Ak[0][c] = part ? Tm[0][c] : x0; Ak[1][c] = part ? Tm[1][c] : x1; Ak[2][c] = part ? Tm[2][c] : x2;
                }
                if (TERRAIN) {
                    DQ_UNROLL for (int a = 0; a < 3; ++a) DQ_UNROLL for (int c = 0; c < 3; ++c) Ak[a][c] += quad_xor2(Ak[a][c]);
                    // rows / columns along the corner's frame (t1, t2, n)
                    float Fm[3][3];
                    DQ_UNROLL for (int a = 0; a < 3; ++a) DQ_UNROLL for (int c = 0; c < 3; ++c)
                        Fm[a][c] = frame[k][3 * a] * Ak[0][c] + frame[k][3 * a + 1] * Ak[1][c] + frame[k][3 * a + 2] * Ak[2][c];
                    DQ_UNROLL for (int a = 0; a < 3; ++a) DQ_UNROLL for (int c = 0; c < 3; ++c)
                        Ak[a][c] = Fm[a][0] * frame[k][3 * c] + Fm[a][1] * frame[k][3 * c + 1] + Fm[a][2] * frame[k][3 * c + 2];
                } else {
                    // (only the diagonal and the couplings zx, zy, xy are used)
                    Ak[0][0] += quad_xor2(Ak[0][0]); Ak[1][1] += quad_xor2(Ak[1][1]); Ak[2][2] += quad_xor2(Ak[2][2]);
                    Ak[0][2] += quad_xor2(Ak[0][2]); Ak[1][2] += quad_xor2(Ak[1][2]); Ak[1][0] += quad_xor2(Ak[1][0]);
                }
                // (half 1 holds the other foot's block in Wg: what it computes here is overwritten by half 0's result)
                invd[k][0] = oct_lo(act[k] ? dw::rcp_nr(Ak[0][0]) * rreg : 0.0f);
                invd[k][1] = oct_lo(act[k] ? dw::rcp_nr(Ak[1][1]) * rreg : 0.0f);
                invd[k][2] = oct_lo(act[k] ? dw::rcp_nr(Ak[2][2]) * rreg : 0.0f);
                cpl[k][0] = oct_lo(Ak[0][2]);     // x row, z column
                cpl[k][1] = oct_lo(Ak[1][2]);     // y row, z column
                cpl[k][2] = oct_lo(Ak[1][0]);     // y row, x column
            }
        }
        // ---- start: tw = tw_free + W lambda0, lambda0 = warm-start impulses as foot wrenches.  A lane's share of a product
        //      W lambda: half 0 its own foot's wrench through W_ff, half 1 the other foot's wrench through W_f,other ----
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
            float lv[6];
            DQ_UNROLL for (int i = 0; i < 6; ++i) lv[i] = quad_xor1_hi(lam[i]);          // (half 1: the other foot's wrench)
            const V6 lvV = v6_from(lv);
            DQ_UNROLL for (int r = 0; r < 3; ++r) {
                const float acc = v6_dot(Wg[r], lvV);
                tw3[r] = (part ? twf[3 + r] : twf[r]) + (acc + oct_xor4(acc));
            }
        }
        DQ_STAMP(B, SB + 9);
        // @phase contact_gs
        // ---- projected Gauss-Seidel, block-Jacobi across the feet: corner kk of the left sole and corner kk of the right
        //      sole are updated together from the same snapshot, the four corners of a sole one after the other ----
        bool pair_on[4];
        DQ_UNROLL for (int kk = 0; kk < 4; ++kk) pair_on[kk] = wave_any(act[kk] != 0);
        for (int it = 0; it < P.iters; ++it) {          /*@trip:5*/
            DQ_UNROLL for (int kk = 0; kk < 4; ++kk) {
                if (pair_on[kk]) {
                    // (the foot's twist: angular rows in the part-0 lane of the pair, linear rows in the part-1 lane)
                    float wv[3], lv[3];
                    DQ_UNROLL for (int i = 0; i < 3; ++i) { wv[i] = quad_pair_lo(tw3[i]); lv[i] = quad_pair_hi(tw3[i]); }
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
                    float lam[6], lmv[6];
                    cross3(r, d, lam);
                    lam[3] = d[0]; lam[4] = d[1]; lam[5] = d[2];
                    DQ_UNROLL for (int i = 0; i < 6; ++i) lmv[i] = quad_xor1_hi(lam[i]);
                    const V6 lmV = v6_from(lmv);
                    DQ_UNROLL for (int rr = 0; rr < 3; ++rr) {
                        const float acc = v6_dot(Wg[rr], lmV);
                        tw3[rr] += acc + oct_xor4(acc);
                    }
                }
            }
        }
        DQ_STAMP(B, SB + 10);
        // @phase contact_up
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
            V6 dp = v6(-Nm[0], -Nm[1], -Nm[2], -Fs[0], -Fs[1], -Fs[2]);
            DQ_UNROLL for (int i = 6; i >= 1; --i) {
                const int b = T - 7 + i;
                F4 s0 = OQ_LD(b, 0, X.pos), s1 = OQ_LD(b, 1, X.pos), s2 = OQ_LD(b, 2, X.pos), s3 = OQ_LD(b, 3, X.pos);
                OQ_KEEP2(s1, s3);
                const float d = -v6_dot(OQ_S6(s0, s1), dp);
                v6_axpy(dp, OQ_S6(s2, s3), d * s1.z);
                OQ_SLOT(b, 1, X.pos) = mk4(s1.x, s1.y, s1.z, d);
            }
            v6_to(dp, dpb);
        }
        {
            float tot[6];
            DQ_UNROLL for (int i = 0; i < 6; ++i) {
                const float a = quad_bcast<0>(dpb[i]), b2 = quad_bcast<1>(dpb[i]);
                tot[i] = a + b2;
            }
            float y = 0.0f;
            DQ_UNROLL for (int c = 0; c < 6; ++c) y -= mrow[c] * tot[c];
            dqb[0] = oct_lo(quad_bcast<0>(y)); dqb[1] = oct_lo(quad_bcast<1>(y)); dqb[2] = oct_lo(quad_bcast<2>(y)); dqb[3] = oct_lo(quad_bcast<3>(y));
            dqb[4] = oct_hi(quad_bcast<0>(y)); dqb[5] = oct_hi(quad_bcast<1>(y));
        }
        if (last && part == 0) {
            DQ_UNROLL for (int i = 0; i < 3; ++i) X.footT[i] = X.footF[i] + Fs[i] * inv_dt;
            if (wr) {
                const OQ_IX dst = (oq_row(DW_NUM_BODIES, e) + (f == 0 ? M.left_foot_gym : M.right_foot_gym)) * 3;
                oq_at(B.contact_forces, dst, 0) = X.footT[0]; oq_at(B.contact_forces, dst, 1) = X.footT[1]; oq_at(B.contact_forces, dst, 2) = X.footT[2];
            }
        }
    } else if (last && part == 0) {
        DQ_UNROLL for (int i = 0; i < 3; ++i) X.footT[i] = X.footF[i];
        if (wr) {
            const OQ_IX dst = (oq_row(DW_NUM_BODIES, e) + (f == 0 ? M.left_foot_gym : M.right_foot_gym)) * 3;
            oq_at(B.contact_forces, dst, 0) = X.footF[0]; oq_at(B.contact_forces, dst, 1) = X.footF[1]; oq_at(B.contact_forces, dst, 2) = X.footF[2];
        }
    }
    if (wr && j < 2 && B.env_state) {
        const OQ_IX wdst = oq_row(DW_ES_WORDS, e) + DW_ES_WARM + 12 * j;
        DQ_UNROLL for (int k = 0; k < 4; ++k) DQ_UNROLL for (int i = 0; i < 3; ++i) oq_at(B.env_state, wdst, 3 * k + i) = Pk[k][i];
    }

    DQ_STAMP(B, SB + 11);
    // @phase outward3
    // ---- outward pass 3: velocity jumps down the tree, final joint velocities (speed limit); the caller integrates ----
    {
        float ar[6] = {0, 0, 0, 0, 0, 0};
        DQ_ROLLED for (int s = 0; s < T; ++s) {          /*@trip:11*/
            const FkHot rc = fk_hot(H, s, j);
            const int b = rc.body, psrc = rc.psrc;
            const int so = own_step(s);
            F4 s0 = OQ_LD(so, 0, X.pos), s1 = OQ_LD(so, 1, X.pos), s2 = OQ_LD(so, 2, X.pos), s3 = OQ_LD(so, 3, X.pos);
            OQ_KEEP1(s3);
            if ((startmask >> s) & 1) {          /*@prob:0.27*/
                const int fm = H.fmask[s];
                if (fm) {          /*@prob:0.33*/
                    for (int xl = 0; xl < 4; ++xl)          /*@trip:1*/
                        if ((fm >> xl) & 1) quad_take_arr(xl, b >= 0 && psrc == 2 + xl, ar);
                }
                const bool fb = b >= 0 && psrc == 1;
                DQ_UNROLL for (int i = 0; i < 6; ++i) ar[i] = fb ? dqb[i] : ar[i];
            }
            wave_sync();
            {
                V6 aV = v6_from(ar);
                const float dq = (s1.w - v6_dot(OQ_S6(s2, s3), aV)) * s1.z;
                v6_axpy(aV, OQ_S6(s0, s1), dq);
                v6_to(aV, ar);
                float qd = s3.z + dq;
                if (qd > rc.vmax) qd = rc.vmax;
                if (qd < -rc.vmax) qd = -rc.vmax;
                if (b >= 0) OQ_SLOT(s, 0, X.pos) = mk4(rc.qlo, qd, rc.qhi, 0.0f);        // joint range and new velocity for integrate_joints
            }
        }
    }
    DQ_STAMP(B, SB + 12);
    // @phase base_final
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

// @phase lane_init
// Lane set-up shared by the entry points: which env this lane works for, its base state and parameters.
DQ_HD void oct_lane_init(OLane &X, const QHot &H, int wave_index, int num_envs, const PhysParams &P, float friction, const OBuf &B) {
    X.lane = lane_id();
    X.wave = wave_index;
    X.o = X.lane & 7; X.j = X.lane & 3; X.h = (X.lane >> 2) & 1; X.q = (X.lane >> 2) & (NQ - 1); X.prim = X.q == 0;
    X.el = X.lane / LPE;
    const int eg = wave_index * EPO + X.el;
    X.valid = eg < num_envs;
    X.env = X.valid ? eg : num_envs - 1;
    X.pos = pcode(H, X.el, X.j);
    DQ_UNROLL for (int i = 0; i < 13; ++i) X.root[i] = oq_at(B.root_states, oq_row(13, X.env), i);
    X.mu = friction * OQ_COLD(friction_scale)[X.env];
    X.footF[0] = X.footF[1] = X.footF[2] = 0.0f;
    X.stamp_base = 0;
    X.coll = 0;
    X.footT[0] = X.footT[1] = X.footT[2] = 0.0f;
    (void)P;
}

// ---- joint-parallel phases.  Per-joint work that touches the Gym tensors runs over ITEMS (env, dof) = lane + 64 k of the
// wave's 16 x 33 joints, so that a wave-instruction reads or writes consecutive addresses (the limb-per-lane mapping would
// touch 64 different rows with 4-byte accesses); the item's lane reaches the owner's slot through the owner table. ----
constexpr int ONI = (EPO * ND + 63) / 64;      // 5 items per lane
struct JointItem { int ok, el, d, b, env; OPos pos; };
DQ_HD JointItem joint_item(const QHot &H, int wave_index, int num_envs, int lane, int k) {
    JointItem it;
    const int i = lane + 64 * k;
    it.el = oq_div<ND>(i); it.d = i - ND * it.el; it.b = it.d + 1;
    const int eg = wave_index * EPO + it.el;
    it.ok = (i < EPO * ND) && (eg < num_envs);
    if (!(i < EPO * ND)) { it.el = 0; it.d = 0; it.b = 1; }
    it.env = eg < num_envs ? eg : num_envs - 1;
    it.pos = icode(H, it.el, it.b);
    return it;
}
// semi-implicit Euler of one joint from the slot the final pass left: q = q_old + dt qd, joint range (outward rate zeroed)
DQ_HD void joint_integrate(OSlots &L, const JointItem &it, float dt, float q_old, float *q_out, float *qd_out) {
    F4 o = OQ_LD(0, 0, it.pos);        // {qlo, qd, qhi, *}
    OQ_KEEP1(o);
    float qd = o.y, q = q_old + dt * qd;
    if (q < o.x) { q = o.x; if (qd < 0) qd = 0; }
    if (q > o.z) { q = o.z; if (qd > 0) qd = 0; }
    *q_out = q; *qd_out = qd;
}

// Gym-boundary substep for 8 envs: tau [N,33], push [N,2] or nullptr (replaces dw::simulate_env)
template <bool TERRAIN>
DQ_HD void oct_simulate(OSlots &L, QHot &HW, const QuadModel &QM, const DevModel &M, const PhysParams &P, float friction, int num_envs,
                        const OBuf &B, const float *tau, const float *push, int wave_index) {
    if (wave_index * EPO >= num_envs) return;        // the second wave of the last workgroup may have no env at all
    OLane X;
    oct_lane_init(X, QM.hot, wave_index, num_envs, P, friction, B);
    stage_hot(HW, QM);
    const QHot &H = HW;
    const int e = X.env, f = X.j & 1;
    float qkeep[ONI];
    DQ_UNROLL for (int k = 0; k < ONI; ++k) {
        const JointItem it = joint_item(H, wave_index, num_envs, X.lane, k);
        const OQ_IX g = oq_row(ND, it.env) + it.d;
        const float q = oq_at(B.dof_state, g * 2, 0), qd = oq_at(B.dof_state, g * 2, 1);
        const float damp = oq_at(B.dof_damping, g), arm = oq_at(B.dof_armature, g);
        qkeep[k] = q;
        if (X.lane + 64 * k < EPO * ND) OQ_SLOT(0, 0, it.pos) = mk4(q, qd, oq_at(tau, g) - damp * qd, arm + P.dt * damp);
    }
    wave_sync();
    oct_substep<TERRAIN>(L, H, QM, M, P, X, B, push ? oq_at(push, 2 * (OQ_IX)e, 0) : 0.0f, push ? oq_at(push, 2 * (OQ_IX)e, 1) : 0.0f, true);
    wave_sync();
    DQ_UNROLL for (int k = 0; k < ONI; ++k) {
        const JointItem it = joint_item(H, wave_index, num_envs, X.lane, k);
        float q, qd;
        joint_integrate(L, it, P.dt, qkeep[k], &q, &qd);
        if (it.ok) {
            const OQ_IX g = oq_row(ND, it.env) + it.d;
            oq_at(B.dof_state, g * 2, 0) = q; oq_at(B.dof_state, g * 2, 1) = qd;
        }
    }
    if (X.valid && X.prim) {
        if (X.j == 0) { DQ_UNROLL for (int i = 0; i < 13; ++i) oq_at(B.root_states, oq_row(13, e), i) = X.root[i]; }
    }
}

}  // namespace OCT_NS
