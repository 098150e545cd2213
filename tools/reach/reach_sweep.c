/* reach_sweep.c -- offline reachability sweep of TOCABI's 61 collision primitives against the 16 capsule proxies / 47 proxy pairs the
 * kernels collide.  TEST / ANALYSIS infrastructure: nothing in the product loads it.
 *
 * The reference collides every primitive with every other one (create_actor(..., group = i, filter = 0),
 * tasks/dyros_dynamic_walk.py:354) and ends the episode on any non-foot contact above 1 N (:590, reward zeroing :937).  The kernels collide
 * capsule PROXIES of some links, in listed pairs (isaacgymdyros_amd/model.py).  This program samples joint configurations inside the MJCF
 * ranges in fp64 and, with the EXACT boxes and cylinders of assets/tocabi_model.json:
 *   - finds every pair of primitives on different, non-adjacent links that touches (distance <= 0) or comes within the contact offset,
 *   - evaluates the proxy test of the kernels (distance of the two capsule axes <= r_a + r_b) for the proxy pair that covers the two links,
 *   - counts per link pair: samples in which the exact primitives touch, in which the covering proxy pair fires, touches the proxy pair
 *     MISSES (false negatives) with the deepest exact penetration among them, and proxy hits with the exact primitives farther apart than
 *     the contact offset (false positives).
 * Exact distance of two convex primitives: alternating projections P_A(P_B(x)) (von Neumann / Cheney-Goldstein: converges to a closest
 * pair of two closed convex sets, to a common point if they intersect); the projection onto a box is a clamp in its frame, onto a cylinder
 * a clamp of the axial coordinate and of the radius.  The iteration's distance decreases monotonically to the true one, so "touches" is
 * never claimed wrongly; "does not touch" is claimed after the step falls below 1e-10 m or 400 iterations (tests/test_reach_sweep.py
 * checks the routine against brute-force surface sampling).
 *
 * input (text, written by tools/reach/reach_sweep.py): see read_model().  usage: reach_sweep model.txt N seed corner_fraction > out.txt */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXB 40
#define MAXG 80
#define MAXP 16
#define MAXPAIR 64

typedef struct { int parent; double pos[3], rot0[9], axis[3], lo, hi; } Body;
typedef struct { int mv, gym, type; double pos[3], rot[9], size[3]; } Geom;       /* type 0 box (half extents), 1 cylinder (radius, half height; axis = local z) */
typedef struct { int mv, gym; double p0[3], p1[3], r; } Proxy;
static int NB, NG, NP, NPAIR, NGYM;
static Body B[MAXB];
static Geom G[MAXG];
static Proxy P[MAXP];
static int PAIR[MAXPAIR][2];
static double OFFSET;

static void m3v(const double *R, const double *v, double *o) { for (int r = 0; r < 3; ++r) o[r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2]; }
static void m3tv(const double *R, const double *v, double *o) { for (int r = 0; r < 3; ++r) o[r] = R[r] * v[0] + R[3 + r] * v[1] + R[6 + r] * v[2]; }
static void m3m(const double *A, const double *Bm, double *o) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[3 * r + c] = A[3 * r] * Bm[c] + A[3 * r + 1] * Bm[3 + c] + A[3 * r + 2] * Bm[6 + c]; }
static void axis_angle(const double *a, double th, double *R) {
    const double c = cos(th), s = sin(th), t = 1 - c, x = a[0], y = a[1], z = a[2];
    R[0] = t * x * x + c; R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
    R[3] = t * x * y + s * z; R[4] = t * y * y + c; R[5] = t * y * z - s * x;
    R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}

/* a placed primitive: centre c, rotation R (columns = local axes in the common frame) */
typedef struct { int type; double c[3], R[9], size[3], brad; } Placed;
static void project(const Placed *s, const double *x, double *o) {          /* closest point of the solid to x */
    double d[3] = {x[0] - s->c[0], x[1] - s->c[1], x[2] - s->c[2]}, l[3];
    m3tv(s->R, d, l);
    if (s->type == 0) {
        for (int k = 0; k < 3; ++k) l[k] = l[k] > s->size[k] ? s->size[k] : (l[k] < -s->size[k] ? -s->size[k] : l[k]);
    } else {
        const double rr = sqrt(l[0] * l[0] + l[1] * l[1]);
        if (rr > s->size[0]) { l[0] *= s->size[0] / rr; l[1] *= s->size[0] / rr; }
        l[2] = l[2] > s->size[1] ? s->size[1] : (l[2] < -s->size[1] ? -s->size[1] : l[2]);
    }
    m3v(s->R, l, o);
    o[0] += s->c[0]; o[1] += s->c[1]; o[2] += s->c[2];
}
/* distance of two placed primitives (0 if they intersect); stops early once it is known to be below `enough` */
double prim_distance(const Placed *a, const Placed *b, double enough) {
    double x[3] = {b->c[0], b->c[1], b->c[2]}, pa[3], pb[3], dist = 1e30;
    for (int it = 0; it < 400; ++it) {
        project(a, x, pa);
        project(b, pa, pb);
        const double d = sqrt((pa[0] - pb[0]) * (pa[0] - pb[0]) + (pa[1] - pb[1]) * (pa[1] - pb[1]) + (pa[2] - pb[2]) * (pa[2] - pb[2]));
        const double step = sqrt((pb[0] - x[0]) * (pb[0] - x[0]) + (pb[1] - x[1]) * (pb[1] - x[1]) + (pb[2] - x[2]) * (pb[2] - x[2]));
        x[0] = pb[0]; x[1] = pb[1]; x[2] = pb[2];
        dist = d;
        if (d <= enough || (it > 0 && step < 1e-10)) break;
    }
    return dist;
}
/* how deep two INTERSECTING primitives overlap, as the kernels' proxies would have to see it: the smallest shrink s of both (every half
 * extent / radius reduced by s) that separates them -- bisection on prim_distance of the shrunk solids, to 0.1 mm */
static double overlap_depth(const Placed *a, const Placed *b) {
    double lo = 0.0, hi = 0.0;
    double mn = 1e30;
    for (int k = 0; k < (a->type ? 2 : 3); ++k) mn = a->size[k] < mn ? a->size[k] : mn;
    for (int k = 0; k < (b->type ? 2 : 3); ++k) mn = b->size[k] < mn ? b->size[k] : mn;
    hi = mn;
    for (int it = 0; it < 12; ++it) {
        const double s = 0.5 * (lo + hi);
        Placed A = *a, Bq = *b;
        for (int k = 0; k < 3; ++k) { A.size[k] = A.size[k] > s ? A.size[k] - s : 0.0; Bq.size[k] = Bq.size[k] > s ? Bq.size[k] - s : 0.0; }
        if (prim_distance(&A, &Bq, 0.0) > 1e-9) hi = s; else lo = s;
    }
    return 0.5 * (lo + hi);
}
/* distance of two segments (the kernels' capsule test uses the axes' distance against r_a + r_b) */
static double seg_seg(const double *p1, const double *q1, const double *p2, const double *q2) {
    double d1[3], d2[3], r[3];
    for (int k = 0; k < 3; ++k) { d1[k] = q1[k] - p1[k]; d2[k] = q2[k] - p2[k]; r[k] = p1[k] - p2[k]; }
    const double a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2], e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
    const double f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2];
    double s, t;
    if (a <= 1e-18 && e <= 1e-18) { s = t = 0; }
    else if (a <= 1e-18) { s = 0; t = f / e; t = t < 0 ? 0 : (t > 1 ? 1 : t); }
    else {
        const double c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2];
        if (e <= 1e-18) { t = 0; s = -c / a; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
        else {
            const double b = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2], den = a * e - b * b;
            s = den > 1e-18 ? (b * f - c * e) / den : 0.0;
            s = s < 0 ? 0 : (s > 1 ? 1 : s);
            t = (b * s + f) / e;
            if (t < 0) { t = 0; s = -c / a; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
            else if (t > 1) { t = 1; s = (b - c) / a; s = s < 0 ? 0 : (s > 1 ? 1 : s); }
        }
    }
    double dd = 0;
    for (int k = 0; k < 3; ++k) { const double v = (p1[k] + s * d1[k]) - (p2[k] + t * d2[k]); dd += v * v; }
    return sqrt(dd);
}

static int read_model(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    if (fscanf(f, "%d %d %d %d %d %lf", &NB, &NG, &NP, &NPAIR, &NGYM, &OFFSET) != 6) return -2;
    for (int i = 0; i < NB; ++i) {
        if (fscanf(f, "%d", &B[i].parent) != 1) return -3;
        for (int k = 0; k < 3; ++k) fscanf(f, "%lf", &B[i].pos[k]);
        for (int k = 0; k < 9; ++k) fscanf(f, "%lf", &B[i].rot0[k]);
        for (int k = 0; k < 3; ++k) fscanf(f, "%lf", &B[i].axis[k]);
        fscanf(f, "%lf %lf", &B[i].lo, &B[i].hi);
    }
    for (int i = 0; i < NG; ++i) {
        fscanf(f, "%d %d %d", &G[i].mv, &G[i].gym, &G[i].type);
        for (int k = 0; k < 3; ++k) fscanf(f, "%lf", &G[i].pos[k]);
        for (int k = 0; k < 9; ++k) fscanf(f, "%lf", &G[i].rot[k]);
        for (int k = 0; k < 3; ++k) fscanf(f, "%lf", &G[i].size[k]);
    }
    for (int i = 0; i < NP; ++i) {
        fscanf(f, "%d %d", &P[i].mv, &P[i].gym);
        for (int k = 0; k < 3; ++k) fscanf(f, "%lf", &P[i].p0[k]);
        for (int k = 0; k < 3; ++k) fscanf(f, "%lf", &P[i].p1[k]);
        fscanf(f, "%lf", &P[i].r);
    }
    for (int i = 0; i < NPAIR; ++i) if (fscanf(f, "%d %d", &PAIR[i][0], &PAIR[i][1]) != 2) return -4;
    fclose(f);
    return 0;
}

/* splitmix64 -> uniform doubles: one stream per sample index, so the result does not depend on the thread count */
static unsigned long long sm64(unsigned long long *s) { unsigned long long z = (*s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static double u01(unsigned long long *s) { return (double)(sm64(s) >> 11) * (1.0 / 9007199254740992.0); }

typedef struct { long long touch, near_, hit, miss, falsepos; double min_dist, max_miss_depth; } Stat;

#ifndef REACH_NO_MAIN
int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: reach_sweep model.txt N seed corner_fraction\n"); return 2; }
    if (read_model(argv[1]) != 0) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const long long N = atoll(argv[2]);
    const unsigned long long seed = strtoull(argv[3], 0, 10);
    const double corner = atof(argv[4]);
    /* per gym-body pair statistics; which proxy pair covers a gym-body pair */
    static Stat S[MAXB + 8][MAXB + 8];
    static int cover[MAXB + 8][MAXB + 8];
    NGYM = NB;          /* (the tables below are indexed by moving body) */
    for (int a = 0; a < NGYM; ++a) for (int b = 0; b < NGYM; ++b) { cover[a][b] = -1; S[a][b].min_dist = 1e30; }
    /* (links = moving bodies: the welded foot bodies are one rigid link with their ankle-roll body, and its proxy stands for the assembly) */
    for (int k = 0; k < NPAIR; ++k) { const int a = P[PAIR[k][0]].mv, b = P[PAIR[k][1]].mv; cover[a][b] = cover[b][a] = k; }
    /* candidate primitive pairs: different moving bodies that are not parent and child */
    static int cand[MAXG * MAXG][2];
    int nc = 0;
    for (int i = 0; i < NG; ++i) for (int j = i + 1; j < NG; ++j) {
        const int a = G[i].mv, b = G[j].mv;
        if (a == b || B[a].parent == b || B[b].parent == a) continue;
        cand[nc][0] = i; cand[nc][1] = j; ++nc;
    }
    double brad[MAXG];
    for (int i = 0; i < NG; ++i) brad[i] = G[i].type == 0 ? sqrt(G[i].size[0] * G[i].size[0] + G[i].size[1] * G[i].size[1] + G[i].size[2] * G[i].size[2])
                                                           : sqrt(G[i].size[0] * G[i].size[0] + G[i].size[1] * G[i].size[1]);
    long long any_touch = 0, any_uncovered = 0;
#pragma omp parallel
    {
        static Stat T[MAXB + 8][MAXB + 8];
#pragma omp threadprivate(T)
        for (int a = 0; a < NGYM; ++a) for (int b = 0; b < NGYM; ++b) { memset(&T[a][b], 0, sizeof(Stat)); T[a][b].min_dist = 1e30; }
        long long t_any = 0, t_unc = 0;
#pragma omp for schedule(dynamic, 256)
        for (long long n = 0; n < N; ++n) {
            unsigned long long st = seed * 0x100000001b3ull + (unsigned long long)n;
            double q[MAXB];
            const int cornered = u01(&st) < corner;
            for (int i = 1; i < NB; ++i) {
                const double u = u01(&st);
                /* a "corner" sample: each joint at its lower limit, its upper limit or anywhere, a third each */
                if (cornered) { const double w = u01(&st); q[i] = w < 1.0 / 3 ? B[i].lo : (w < 2.0 / 3 ? B[i].hi : B[i].lo + u * (B[i].hi - B[i].lo)); }
                else q[i] = B[i].lo + u * (B[i].hi - B[i].lo);
            }
            /* forward kinematics in the base frame */
            double R[MAXB][9], x[MAXB][3];
            for (int k = 0; k < 9; ++k) R[0][k] = (k % 4 == 0);
            x[0][0] = x[0][1] = x[0][2] = 0;
            for (int i = 1; i < NB; ++i) {
                const int p = B[i].parent;
                double Rj[9], Rt[9], t[3];
                axis_angle(B[i].axis, q[i], Rj);
                m3m(R[p], B[i].rot0, Rt);
                m3m(Rt, Rj, R[i]);
                m3v(R[p], B[i].pos, t);
                for (int k = 0; k < 3; ++k) x[i][k] = x[p][k] + t[k];
            }
            Placed pl[MAXG];
            for (int i = 0; i < NG; ++i) {
                double t[3];
                pl[i].type = G[i].type;
                m3m(R[G[i].mv], G[i].rot, pl[i].R);
                m3v(R[G[i].mv], G[i].pos, t);
                for (int k = 0; k < 3; ++k) { pl[i].c[k] = x[G[i].mv][k] + t[k]; pl[i].size[k] = G[i].size[k]; }
            }
            /* the proxies' axes and which proxy pairs fire */
            double e0[MAXP][3], e1[MAXP][3];
            for (int i = 0; i < NP; ++i) {
                double t[3];
                m3v(R[P[i].mv], P[i].p0, t); for (int k = 0; k < 3; ++k) e0[i][k] = x[P[i].mv][k] + t[k];
                m3v(R[P[i].mv], P[i].p1, t); for (int k = 0; k < 3; ++k) e1[i][k] = x[P[i].mv][k] + t[k];
            }
            int fires[MAXPAIR];
            for (int k = 0; k < NPAIR; ++k) {
                const int a = PAIR[k][0], b = PAIR[k][1];
                fires[k] = seg_seg(e0[a], e1[a], e0[b], e1[b]) <= P[a].r + P[b].r;
            }
            /* exact primitives: per gym-body pair the least distance of this sample */
            static double dmin[MAXB + 8][MAXB + 8];
            static int pmin[MAXB + 8][MAXB + 8][2];
#pragma omp threadprivate(dmin, pmin)
            int touched[256][2], ntouched = 0;
            for (int c = 0; c < nc; ++c) {
                const int i = cand[c][0], j = cand[c][1];
                const double dc = sqrt((pl[i].c[0] - pl[j].c[0]) * (pl[i].c[0] - pl[j].c[0]) + (pl[i].c[1] - pl[j].c[1]) * (pl[i].c[1] - pl[j].c[1]) +
                                       (pl[i].c[2] - pl[j].c[2]) * (pl[i].c[2] - pl[j].c[2]));
                if (dc > brad[i] + brad[j] + OFFSET) continue;
                const double d = prim_distance(&pl[i], &pl[j], 0.0);
                if (d > OFFSET) continue;
                int a = G[i].mv, b = G[j].mv;
                if (a > b) { const int t = a; a = b; b = t; }
                int seen = 0;
                for (int k = 0; k < ntouched; ++k) if (touched[k][0] == a && touched[k][1] == b) seen = 1;
                if (!seen && ntouched < 256) { touched[ntouched][0] = a; touched[ntouched][1] = b; ++ntouched; dmin[a][b] = 1e30; }
                if (d < dmin[a][b]) { dmin[a][b] = d; pmin[a][b][0] = i; pmin[a][b][1] = j; }
            }
            int s_touch = 0, s_unc = 0;
            for (int k = 0; k < ntouched; ++k) {
                const int a = touched[k][0], b = touched[k][1];
                Stat *T_ = &T[a][b];
                const double d = dmin[a][b];
                T_->near_ += 1;
                if (d < T_->min_dist) T_->min_dist = d;
                if (d <= 1e-9) {
                    T_->touch += 1; s_touch = 1;
                    const int cv = cover[a][b];
                    if (cv < 0) s_unc = 1;
                    else if (!fires[cv]) {
                        T_->miss += 1;
                        const double dep = overlap_depth(&pl[pmin[a][b][0]], &pl[pmin[a][b][1]]);
                        if (dep > T_->max_miss_depth) T_->max_miss_depth = dep;
                    }
                }
            }
            /* proxy hits, and those among them whose exact primitives are farther apart than the contact offset */
            for (int k = 0; k < NPAIR; ++k) if (fires[k]) {
                int a = P[PAIR[k][0]].mv, b = P[PAIR[k][1]].mv;
                if (a > b) { const int t = a; a = b; b = t; }
                T[a][b].hit += 1;
                int near = 0;
                for (int m = 0; m < ntouched; ++m) if (touched[m][0] == a && touched[m][1] == b) near = 1;
                if (!near) T[a][b].falsepos += 1;
            }
            t_any += s_touch; t_unc += s_unc;
        }
#pragma omp critical
        {
            any_touch += t_any; any_uncovered += t_unc;
            for (int a = 0; a < NGYM; ++a) for (int b = 0; b < NGYM; ++b) {
                S[a][b].touch += T[a][b].touch; S[a][b].near_ += T[a][b].near_; S[a][b].hit += T[a][b].hit; S[a][b].miss += T[a][b].miss;
                S[a][b].falsepos += T[a][b].falsepos;
                if (T[a][b].min_dist < S[a][b].min_dist) S[a][b].min_dist = T[a][b].min_dist;
                if (T[a][b].max_miss_depth > S[a][b].max_miss_depth) S[a][b].max_miss_depth = T[a][b].max_miss_depth;
            }
        }
    }
    printf("samples %lld any_touch %lld any_uncovered_touch %lld candidates %d\n", N, any_touch, any_uncovered, nc);
    for (int a = 0; a < NGYM; ++a) for (int b = a + 1; b < NGYM; ++b) {
        const Stat *s = &S[a][b];
        if (s->near_ == 0 && s->hit == 0) continue;
        printf("pair %d %d cover %d near %lld touch %lld hit %lld miss %lld falsepos %lld min_dist %.6f max_miss_depth %.5f\n", a, b, cover[a][b], s->near_, s->touch, s->hit,
               s->miss, s->falsepos, s->min_dist > 1e29 ? -1.0 : s->min_dist, s->max_miss_depth);
    }
    return 0;
}
#endif
