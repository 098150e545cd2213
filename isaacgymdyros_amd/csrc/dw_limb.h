// dw_limb.h -- pieces of the physics substep that every lane layout computes the same way, per body or per contact, on plain
// float arrays in registers: the 16-byte slot word (F4), the penalty ground force of a collision primitive (flat ground or
// height field), the rigid-body inertia of a body about the common reference point and its gyroscopic bias, the closest points
// of two segments and the capsule-pair contact of the self-collision, quaternion product and the fast sincos.  Used by the octet
// kernels (dw_oct.h) and the lane kernels (dw_lane.h); same physics, same written decisions as oracle/dw_physics.c (DESIGN.md
// "Physics model").  The cross-lane vocabulary is dw_quad_wave.h, the limb schedule and its tables dw_quad_model.h.
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

// ---- spatial 6-vectors as THREE PAIRS (round 6).  gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- two fp32 operations per lane
// and instruction -- at 5.6 cycles per SIMD where two plain instructions take 9.3 (tools/valu_issue.hip), and the step kernel is bound by
// vector issue.  The SLP vectoriser forms such pairs from scalar code at the price of moves and registers (612 B of
// scratch); written as 2-vectors the pairs are the data's own layout -- a 6-vector is (v0 v1)(v2 v3)(v4 v5), a slot row's .xy / .zw are
// pairs as they come from LDS -- and a scalar factor is the instruction's op_sel broadcast: no moves.  dot6 is then 3 packed + 1 add
// (instead of 6), a scaled add 3 (instead of 6).  Summation order of a dot product: (0, 2, 4) and (1, 3, 5) side by side, then their sum.
// OQ_PACKED (set by the translation unit): 1 = the pairs are 2-vectors and compile to the packed instructions -- the hex instantiation
// (dw_hex_kernels.hip), whose lone wave per SIMD issues a packed instruction in 6.1 cycles against 11.0 for two plain ones: -1.9 % of the
// step at 4096 envs (same-box A/B, profiles/r06_r6d_packed_ab.txt).  0 = the same arithmetic in the same order on two scalars -- the octet
// kernels: at two waves per SIMD the count of vector instructions fell 3.8 % (27.0 k -> 26.0 k per wave) and the step time not at all
// (0.1385 ms both; a packed instruction costs 5.6 cycles there, wait cycles rose 5 %), and the two-waves height-field kernels went from
// 168 to 208 B of scratch.  (hipcc 7.2 also crashes in the register allocator on dw_k_simulate_oct<true, 2> with the pairs as 2-vectors
// under -amdgpu-sched-strategy=iterative-ilp.)
#if !defined(OQ_PACKED)
#define OQ_PACKED 0
#endif
#if defined(__HIPCC__) && OQ_PACKED
typedef float P2 __attribute__((ext_vector_type(2)));
DQ_HD P2 mk2(float a, float b) { P2 r = {a, b}; return r; }
DQ_HD P2 pk_fma(P2 a, P2 b, P2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct P2 { float x, y; };
DQ_HD P2 mk2(float a, float b) { P2 r; r.x = a; r.y = b; return r; }
DQ_HD P2 operator*(P2 a, P2 b) { return mk2(a.x * b.x, a.y * b.y); }
DQ_HD P2 operator+(P2 a, P2 b) { return mk2(a.x + b.x, a.y + b.y); }
DQ_HD P2 pk_fma(P2 a, P2 b, P2 c) { return mk2(a.x * b.x + c.x, a.y * b.y + c.y); }
#endif
struct V6 { P2 a, b, c; };
DQ_HD V6 v6(float x0, float x1, float x2, float x3, float x4, float x5) { V6 r; r.a = mk2(x0, x1); r.b = mk2(x2, x3); r.c = mk2(x4, x5); return r; }
DQ_HD V6 v6_zero() { return v6(0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f); }
DQ_HD V6 v6_from(const float *x) { return v6(x[0], x[1], x[2], x[3], x[4], x[5]); }
DQ_HD void v6_to(const V6 &v, float *x) { x[0] = v.a.x; x[1] = v.a.y; x[2] = v.b.x; x[3] = v.b.y; x[4] = v.c.x; x[5] = v.c.y; }
DQ_HD float v6_dot(const V6 &u, const V6 &v) { P2 t = u.a * v.a; t = pk_fma(u.b, v.b, t); t = pk_fma(u.c, v.c, t); return t.x + t.y; }
DQ_HD void v6_axpy(V6 &y, const V6 &x, float s) { const P2 ss = mk2(s, s); y.a = pk_fma(x.a, ss, y.a); y.b = pk_fma(x.b, ss, y.b); y.c = pk_fma(x.c, ss, y.c); }
DQ_HD V6 v6_scale(const V6 &x, float s) { const P2 ss = mk2(s, s); V6 r; r.a = x.a * ss; r.b = x.b * ss; r.c = x.c * ss; return r; }
DQ_HD void v6_add(V6 &y, const V6 &x) { y.a = y.a + x.a; y.b = y.b + x.b; y.c = y.c + x.c; }

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

// chain starts of the outward passes: a[i] = take ? (a[i] of quad lane xl) : a[i], in place (wave-uniform xl, per-lane take) -- one
// DPP move and one select per word, no control flow (written as branches, the register copies at the joins of those branches were
// 6 moves per word: tools/isa_dyn.py, round 5)
template <int XL, int N> DQ_HD void quad_take_arr_x(bool take, float (&a)[N]) {
    DQ_UNROLL for (int i = 0; i < N; ++i) { const float t = quad_bcast<XL>(a[i]); a[i] = take ? t : a[i]; }
}
template <int N> DQ_HD void quad_take_arr(int xl, bool take, float (&a)[N]) {
    if (xl == 0) quad_take_arr_x<0>(take, a);
    else if (xl == 1) quad_take_arr_x<1>(take, a);
    else if (xl == 2) quad_take_arr_x<2>(take, a);
    else quad_take_arr_x<3>(take, a);
}

// Profiling builds (-DDQ_STAMPS) record the clock at phase boundaries of wave 0 into the free tail of gate_acc (words
// 200..): tools/phase_stamps.py.  Never defined in the shipped library.
#if defined(DQ_STAMPS) && defined(__HIPCC__)
// (profiling builds, -DDQ_STAMPS: wave 0 leaves its cycle counter in the spare words of gate_acc; (B) is an OBuf, dw_bufg.h)
#define DQ_STAMP(B, n) do { if (blockIdx.x == 0 && threadIdx.x == 0) (B).cold->gate_acc[200 + (n)] = (int64_t)__builtin_readcyclecounter(); } while (0)
#else
#define DQ_STAMP(B, n) do { } while (0)
#endif

// |F| > 1 N with torch.norm's summation order for 3 elements (the termination test of tasks/dyros_dynamic_walk.py:590)
DQ_HD bool over_1n(const float *F) {
    float b0 = fmaf(F[0], F[0], 0.0f);
    b0 = fmaf(F[1], F[1], b0);
    b0 = fmaf(F[2], F[2], b0);
    return sqrtf(b0) > 1.0f;
}

// Ground penalty force of one primitive of body b (oracle/dw_physics.c).  R, x: body rotation / origin relative to O; v: body
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

// Rigid-body inertia of a body about O in world axes from its (<= 2) inertial records (oracle/dw_physics.c):
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
// The same from records PREPARED on the host (dw_quad_model.h, the hot tables' in / in1 records): hm = mass * com and
// A6 = I + mass (|com|^2 1 - com com') (xx yy zz xy xz yz), the body-frame inertia about the body origin at mass scale 1.  Both are
// linear in the mass scale, so a record costs 10 multiplications here instead of the 34 operations of forming them per env and substep.
DQ_HD void rigid_inertia_pre(int nin, const float *hm0, float m0, const float *A60, float ms0, const float *hm1, float m1, const float *A61,
                             float ms1, const float *R, const float *x, float *Ao, float *ho, float *mass_out) {
    float A6[6], h[3], mass = ms0 * m0;
    DQ_UNROLL for (int i = 0; i < 6; ++i) A6[i] = ms0 * A60[i];
    DQ_UNROLL for (int i = 0; i < 3; ++i) h[i] = ms0 * hm0[i];
    if (nin > 1) {
        DQ_UNROLL for (int i = 0; i < 6; ++i) A6[i] += ms1 * A61[i];
        DQ_UNROLL for (int i = 0; i < 3; ++i) h[i] += ms1 * hm1[i];
        mass += ms1 * m1;
    }
    const float A[9] = {A6[0], A6[3], A6[4], A6[3], A6[1], A6[5], A6[4], A6[5], A6[2]};
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

// Penalty force of two capsules (oracle/dw_physics.c).  Returns true and fills F (force on A), pa, pb when they overlap.
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


}  // namespace dwq
