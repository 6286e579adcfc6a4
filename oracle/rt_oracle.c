/*
 * rt_oracle.c — CPU restatement of the reference path tracer's hot path.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h). Plain C, f64, reference op order.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see Makefile).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ====================================================================== */
/* geometry.rs                                                            */
/* ====================================================================== */

/* geometry.rs:6-8 */
double o_dot(const double x[3], const double y[3]) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; }

/* geometry.rs:11-13 */
void o_cross(const double x[3], const double y[3], double out[3]) {
    double a = x[1] * y[2] - x[2] * y[1];
    double b = x[2] * y[0] - x[0] * y[2];
    double c = x[0] * y[1] - x[1] * y[0];
    out[0] = a; out[1] = b; out[2] = c;
}

/* geometry.rs:16-18 */
void o_add(const double x[3], const double y[3], double out[3]) {
    double a = x[0] + y[0], b = x[1] + y[1], c = x[2] + y[2];
    out[0] = a; out[1] = b; out[2] = c;
}

/* geometry.rs:21-23 */
void o_sub(const double x[3], const double y[3], double out[3]) {
    double a = x[0] - y[0], b = x[1] - y[1], c = x[2] - y[2];
    out[0] = a; out[1] = b; out[2] = c;
}

/* geometry.rs:26-28 */
double o_mag(const double x[3]) { return sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); }

/* geometry.rs:31-33 */
double o_d(const double x[3], const double y[3]) {
    double t[3];
    o_sub(x, y, t);
    return o_mag(t);
}

/* geometry.rs:36-38 */
void o_scale(double c, const double x[3], double out[3]) {
    double a = c * x[0], b = c * x[1], e = c * x[2];
    out[0] = a; out[1] = b; out[2] = e;
}

/* geometry.rs:40-46 (panic on the zero vector becomes -1) */
int o_normalize(const double x[3], double out[3]) {
    double m = o_mag(x);
    if (m == 0.0) return -1;
    o_scale(1.0 / m, x, out);
    return 0;
}

/* geometry.rs:48-53 */
double o_min(double x, double y) { return x < y ? x : y; }
double o_max(double x, double y) { return x < y ? y : x; }

/* geometry.rs:56-65; the two rng.gen() draws are passed in, r first */
void o_sample_disk(double radius, double r_raw, double theta_raw, double out[2]) {
    double r = radius * sqrt(r_raw);
    double theta = M_PI * 2.0 * theta_raw;
    out[0] = r * cos(theta);
    out[1] = r * sin(theta);
}

/* geometry.rs:68-74 */
void o_mat_mult_3(const double A[9], const double x[3], double out[3]) {
    double a = A[0] * x[0] + A[1] * x[1] + A[2] * x[2];
    double b = A[3] * x[0] + A[4] * x[1] + A[5] * x[2];
    double c = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
    out[0] = a; out[1] = b; out[2] = c;
}

/* geometry.rs:76-106 */
int o_poly_inclusion_2d(const double *points, size_t n, const double test[2]) {
    size_t l = n;
    int crossings = 0;
    for (size_t i = 0; i < l; i++) {
        const double *p1 = points + 2 * i;
        const double *p2 = points;
        if (i + 1 < l) p2 = points + 2 * (i + 1);
        if (p1[1] > test[1] && p2[1] < test[1]) {
            double d = p1[1] - p2[1];
            double w1 = d - (p1[1] - test[1]);
            double w2 = d - (test[1] - p2[1]);
            double x = (w1 * p1[0] + w2 * p2[0]) / d;
            if (x >= test[0]) crossings += 1;
        }
        if (p1[1] < test[1] && p2[1] > test[1]) {
            double d = p2[1] - p1[1];
            double w1 = d - (test[1] - p1[1]);
            double w2 = d - (p2[1] - test[1]);
            double x = (w1 * p1[0] + w2 * p2[0]) / d;
            if (x >= test[0]) crossings += 1;
        }
    }
    return crossings % 2 == 1;
}

/* geometry.rs:109-118; theta is drawn first, then phi */
void o_sample_z_hemisphere(double theta_raw, double phi_raw, double out[3]) {
    double theta = M_PI * 2.0 * theta_raw;
    double phi = acos(phi_raw);
    out[0] = cos(theta) * sin(phi);
    out[1] = sin(theta) * sin(phi);
    out[2] = cos(phi);
}

/* geometry.rs:121-130 */
int o_unit_perp_vec(const double x[3], double out[3]) {
    double v[3] = {1.0, 0.0, 0.0};
    const double other[3] = {0.0, 1.0, 0.0};
    if (fabs(o_dot(x, v)) > fabs(o_dot(x, other))) {
        v[0] = 0.0; v[1] = 1.0; v[2] = 0.0;
    }
    double c[3];
    o_cross(x, v, c);
    return o_normalize(c, out);
}

/* ====================================================================== */
/* brdf.rs                                                                */
/* ====================================================================== */

/* brdf.rs:14-29; in_dir and self are ignored by the reference as well */
void o_brdf_sample(const o_brdf *self, const double in_dir[3], const double normal[3],
                   double theta_raw, double phi_raw, double out[3]) {
    (void)self; (void)in_dir;
    double out_raw[3], new_x[3], new_y[3], a[3], b[3], c[3], bc[3];
    o_sample_z_hemisphere(theta_raw, phi_raw, out_raw);
    if (o_unit_perp_vec(normal, new_x) != 0) { /* reference would panic; propagate NaN */
        out[0] = out[1] = out[2] = NAN;
        return;
    }
    o_cross(normal, new_x, new_y);
    o_scale(out_raw[0], new_x, a);
    o_scale(out_raw[1], new_y, b);
    o_scale(out_raw[2], normal, c);
    o_add(b, c, bc);
    o_add(a, bc, out);
}

/* ====================================================================== */
/* primitives.rs                                                          */
/* ====================================================================== */

/* primitives.rs:45-81 */
static int sphere_intersect(const o_prim *p, const double o[3], const double dir[3], double pos[3]) {
    const o_sphere *s = (const o_sphere *)p;
    double leg1[3];
    o_sub(s->center, o, leg1);
    if (o_dot(leg1, dir) < 0.0) return 0;
    double cr[3];
    o_cross(dir, leg1, cr);
    double par_area = o_mag(cr);
    double r = par_area;
    if (r > s->radius) return 0;
    double a = o_mag(leg1);
    double d = sqrt(a * a - r * r) - sqrt(s->radius * s->radius - r * r);
    double sc[3];
    o_scale(d, dir, sc);
    o_add(o, sc, pos);
    return 1;
}
/* primitives.rs:82-84 */
static void sphere_normal_at(const o_prim *p, const double pt[3], double n[3]) {
    const o_sphere *s = (const o_sphere *)p;
    double t[3];
    o_sub(pt, s->center, t);
    o_scale(1.0 / s->radius, t, n);
}
/* primitives.rs:85-92 */
static void sphere_package(const o_prim *p, double out[4]) {
    const o_sphere *s = (const o_sphere *)p;
    out[0] = s->center[0]; out[1] = s->center[1]; out[2] = s->center[2]; out[3] = s->radius;
}
static const o_brdf *sphere_brdf(const o_prim *p) { return &((const o_sphere *)p)->brdf; }
static void sphere_destroy(o_prim *p) { free(p); }
static const o_prim_vtbl SPHERE_VT = {sphere_intersect, sphere_normal_at, sphere_package, sphere_brdf,
                                      sphere_destroy};

o_prim *o_sphere_new(const double center[3], double radius, const o_brdf *brdf) {
    o_sphere *s = (o_sphere *)calloc(1, sizeof(o_sphere));
    s->base.vt = &SPHERE_VT;
    s->base.kind = 0;
    memcpy(s->center, center, sizeof(double) * 3);
    s->radius = radius;
    s->brdf = *brdf;
    return &s->base;
}

/* primitives.rs:206-248 */
static int polygon_intersect(const o_prim *pp, const double o[3], const double dir[3], double pos[3]) {
    const o_polygon *s = (const o_polygon *)pp;
    double p = o_dot(s->normal, dir);
    double t0[3];
    o_sub(o, s->points, t0);
    double test = o_dot(t0, s->normal);
    if (test <= O_EPSILON || p >= -O_EPSILON) return 0;

    double a[3];
    o_sub(s->bounding_center, o, a);
    double d = fabs(o_dot(a, s->normal));
    double sc[3], plane_intersection[3];
    o_scale(d / fabs(p), dir, sc);
    o_add(o, sc, plane_intersection);

    if (o_d(plane_intersection, s->bounding_center) > s->bounding_radius) return 0;

    double off[3];
    o_sub(plane_intersection, s->points, off);
    double i2d[2] = {o_dot(s->basis_1, off), o_dot(s->basis_2, off)};
    if (o_poly_inclusion_2d(s->points_in_basis, s->n, i2d)) {
        pos[0] = plane_intersection[0]; pos[1] = plane_intersection[1]; pos[2] = plane_intersection[2];
        return 1;
    }
    return 0;
}
/* primitives.rs:250-252 */
static void polygon_normal_at(const o_prim *p, const double pt[3], double n[3]) {
    (void)pt;
    const o_polygon *s = (const o_polygon *)p;
    n[0] = s->normal[0]; n[1] = s->normal[1]; n[2] = s->normal[2];
}
/* primitives.rs:255-267 */
static void polygon_package(const o_prim *p, double out[4]) {
    const o_polygon *s = (const o_polygon *)p;
    double mean[3] = {0.0, 0.0, 0.0};
    for (size_t i = 0; i < s->n; i++) o_add(mean, s->points + 3 * i, mean);
    o_scale(1.0 / (double)s->n, mean, mean);
    double radius = 0.0;
    for (size_t i = 0; i < s->n; i++) radius = o_max(radius, o_d(s->points + 3 * i, mean));
    out[0] = mean[0]; out[1] = mean[1]; out[2] = mean[2]; out[3] = radius;
}
static const o_brdf *polygon_brdf(const o_prim *p) { return &((const o_polygon *)p)->brdf; }
static void polygon_destroy(o_prim *p) {
    o_polygon *s = (o_polygon *)p;
    free(s->points);
    free(s->points_in_basis);
    free(s);
}
static const o_prim_vtbl POLYGON_VT = {polygon_intersect, polygon_normal_at, polygon_package, polygon_brdf,
                                       polygon_destroy};

/* primitives.rs:144-199 */
o_prim *o_polygon_new_cached(const double *points, size_t n, const o_brdf *brdf,
                             const double direc_vec[3]) {
    if (n < 3) return NULL; /* reference indexes points[2] */
    double e0[3], e1[3], normal[3];
    o_sub(points + 3, points, e0);
    o_sub(points + 6, points + 3, e1);
    o_cross(e0, e1, normal);
    if (o_dot(normal, direc_vec) < 0.0) o_scale(-1.0, normal, normal);
    if (o_normalize(normal, normal) != 0) return NULL;

    double basis_1[3], basis_2[3];
    o_sub(points + 3, points, basis_1);
    o_cross(normal, basis_1, basis_2);
    if (o_normalize(basis_1, basis_1) != 0) return NULL;
    if (o_normalize(basis_2, basis_2) != 0) return NULL;

    o_polygon *s = (o_polygon *)calloc(1, sizeof(o_polygon));
    s->base.vt = &POLYGON_VT;
    s->base.kind = 1;
    s->n = n;
    s->points = (double *)malloc(sizeof(double) * 3 * n);
    memcpy(s->points, points, sizeof(double) * 3 * n);
    s->points_in_basis = (double *)malloc(sizeof(double) * 2 * n);
    for (size_t i = 0; i < n; i++) {
        double off[3];
        o_sub(points + 3 * i, points, off);
        s->points_in_basis[2 * i] = o_dot(off, basis_1);
        s->points_in_basis[2 * i + 1] = o_dot(off, basis_2);
    }
    double centroid[3] = {0.0, 0.0, 0.0};
    for (size_t i = 0; i < n; i++) o_add(centroid, points + 3 * i, centroid);
    o_scale(1.0 / (double)n, centroid, centroid);
    double radius = o_d(centroid, points);
    for (size_t i = 0; i < n; i++) {
        double r = o_d(centroid, points + 3 * i);
        if (r > radius) radius = r;
    }
    s->brdf = *brdf;
    memcpy(s->bounding_center, centroid, sizeof centroid);
    s->bounding_radius = radius;
    memcpy(s->normal, normal, sizeof normal);
    memcpy(s->basis_1, basis_1, sizeof basis_1);
    memcpy(s->basis_2, basis_2, sizeof basis_2);
    return &s->base;
}

/* primitives.rs:117-141 */
int o_polygon_parallelpiped(const double origin[3], const double gen1[3], const double gen2[3],
                            const double gen3[3], const o_brdf *brdf, o_prim *out[6]) {
    double f1[12], f2[12], f3[12], f4[12], f5[12], f6[12], t[3];
    double d1[3], d2[3], d3[3];
    /* face1: origin, origin+gen1, (origin+gen1)+gen2, origin+gen2 ; dir -gen3 */
    memcpy(f1, origin, 24);
    o_add(origin, gen1, f1 + 3);
    o_add(origin, gen1, t); o_add(t, gen2, f1 + 6);
    o_add(origin, gen2, f1 + 9);
    o_scale(-1.0, gen3, d1);
    /* face2: origin, origin+gen2, (origin+gen2)+gen3, origin+gen3 ; dir -gen1 */
    memcpy(f2, origin, 24);
    o_add(origin, gen2, f2 + 3);
    o_add(origin, gen2, t); o_add(t, gen3, f2 + 6);
    o_add(origin, gen3, f2 + 9);
    o_scale(-1.0, gen1, d2);
    /* face3: origin, origin+gen1, (origin+gen1)+gen3, origin+gen3 ; dir -gen2 */
    memcpy(f3, origin, 24);
    o_add(origin, gen1, f3 + 3);
    o_add(origin, gen1, t); o_add(t, gen3, f3 + 6);
    o_add(origin, gen3, f3 + 9);
    o_scale(-1.0, gen2, d3);
    for (int i = 0; i < 4; i++) {
        o_add(f1 + 3 * i, gen3, f4 + 3 * i);
        o_add(f2 + 3 * i, gen1, f5 + 3 * i);
        o_add(f3 + 3 * i, gen2, f6 + 3 * i);
    }
    out[0] = o_polygon_new_cached(f1, 4, brdf, d1);
    out[1] = o_polygon_new_cached(f2, 4, brdf, d2);
    out[2] = o_polygon_new_cached(f3, 4, brdf, d3);
    out[3] = o_polygon_new_cached(f4, 4, brdf, gen3);
    out[4] = o_polygon_new_cached(f5, 4, brdf, gen1);
    out[5] = o_polygon_new_cached(f6, 4, brdf, gen2);
    for (int i = 0; i < 6; i++)
        if (!out[i]) return -1;
    return 0;
}

void o_prim_free(o_prim *p) {
    if (p) p->vt->destroy(p);
}

/* ====================================================================== */
/* bvh.rs                                                                 */
/* ====================================================================== */

/* bvh.rs:77-95 */
void o_bvh_package(o_prim *const *objects, size_t n, double out[4]) {
    if (n == 0) {
        out[0] = out[1] = out[2] = out[3] = 0.0;
        return;
    }
    double mean_pos[3] = {0.0, 0.0, 0.0};
    for (size_t i = 0; i < n; i++) {
        double b[4];
        objects[i]->vt->package(objects[i], b);
        o_add(mean_pos, b, mean_pos);
    }
    o_scale(1.0 / (double)n, mean_pos, mean_pos);
    double radius = 0.0;
    for (size_t i = 0; i < n; i++) {
        double b[4];
        objects[i]->vt->package(objects[i], b);
        radius = o_max(radius, o_d(b, mean_pos) + b[3]);
    }
    out[0] = mean_pos[0]; out[1] = mean_pos[1]; out[2] = mean_pos[2]; out[3] = radius;
}

static o_bvh *make_leaf(o_prim **objects, size_t n) {
    o_bvh *b = (o_bvh *)calloc(1, sizeof(o_bvh));
    o_bvh_package(objects, n, b->bounding_shell);
    b->base_contents = objects;
    b->n_contents = n;
    return b;
}

/* bvh.rs:21-75 */
o_bvh *o_bvh_from_list(o_prim **objects, size_t n) {
    if (n <= 4) return make_leaf(objects, n);

    double shell[4];
    o_bvh_package(objects, n, shell);
    double mean_pos[3] = {shell[0], shell[1], shell[2]};

    double mean_dir[3] = {0.0, 0.0, 0.0};
    for (size_t i = 0; i < n; i++) {
        double b[4], obj_dir[3];
        objects[i]->vt->package(objects[i], b);
        o_sub(b, mean_pos, obj_dir);
        if (o_mag(obj_dir) > O_EPSILON) {
            o_normalize(obj_dir, obj_dir);
        } else {
            obj_dir[0] = 0.0; obj_dir[1] = 0.0; obj_dir[2] = 0.01;
        }
        if (o_dot(obj_dir, mean_dir) < 0.0) o_scale(-1.0, obj_dir, obj_dir);
        o_add(mean_dir, obj_dir, mean_dir);
    }

    o_prim **left = (o_prim **)malloc(sizeof(o_prim *) * n);
    o_prim **right = (o_prim **)malloc(sizeof(o_prim *) * n);
    size_t nl = 0, nr = 0;
    for (size_t i = 0; i < n; i++) {
        double b[4], t[3];
        objects[i]->vt->package(objects[i], b);
        o_sub(b, mean_pos, t);
        if (o_dot(t, mean_dir) > 0.0) left[nl++] = objects[i];
        else right[nr++] = objects[i];
    }
    if (nl == 0 || nr == 0) {
        /* The reference recurses without bound here (every centre on one side).
           No reference behaviour exists to match; keep the list as one leaf. */
        free(left);
        free(right);
        return make_leaf(objects, n);
    }
    free(objects);
    o_bvh *b = (o_bvh *)calloc(1, sizeof(o_bvh));
    memcpy(b->bounding_shell, shell, sizeof shell);
    b->left = o_bvh_from_list((o_prim **)realloc(left, sizeof(o_prim *) * nl), nl);
    b->right = o_bvh_from_list((o_prim **)realloc(right, sizeof(o_prim *) * nr), nr);
    return b;
}

/* bvh.rs:152-164 */
int o_shell_intersect(const double shell[4], const double o[3], const double dir[3]) {
    double leg1[3] = {shell[0] - o[0], shell[1] - o[1], shell[2] - o[2]};
    if (o_dot(leg1, dir) < 0.0 && o_mag(leg1) > shell[3]) return 0;
    double cr[3];
    o_cross(dir, leg1, cr);
    return o_mag(cr) <= shell[3];
}

/* bvh.rs:97-149 */
void o_bvh_intersect(const o_bvh *self, const double o[3], const double dir[3], o_hit *out,
                     o_stats *st) {
    size_t cap = 16, len = 0;
    const o_bvh **stack = (const o_bvh **)malloc(sizeof(*stack) * cap); /* vec![self], bvh.rs:99 */
    stack[len++] = self;
    int have = 0;
    double closest_distance = 0.0;
    out->hit = 0; out->prim_id = O_MISS; out->brdf = NULL; out->distance = 0.0;
    out->pos[0] = out->pos[1] = out->pos[2] = 0.0;
    out->normal[0] = out->normal[1] = out->normal[2] = 0.0;

    while (len > 0) {
        const o_bvh *root = stack[--len];
        if (st) st->nodes_popped++;
        const double *b = root->bounding_shell;
        if (have && o_d(b, out->pos) - b[3] > closest_distance) {
            if (st) st->nodes_pruned++;
            continue;
        }
        if (st) st->shell_tests++;
        if (o_shell_intersect(b, o, dir)) {
            if (st) st->shell_hits++;
            for (size_t k = 0; k < root->n_contents; k++) {
                const o_prim *prim = root->base_contents[k];
                double pos[3];
                if (st) { if (prim->kind == 0) st->sphere_tests++; else st->poly_tests++; }
                if (!prim->vt->intersect(prim, o, dir, pos)) continue;
                if (st) st->prim_hits++;
                double dd = o_d(pos, o);
                if (!have || dd < closest_distance) {
                    have = 1;
                    closest_distance = dd;
                    out->hit = 1;
                    out->prim_id = prim->id;
                    memcpy(out->pos, pos, sizeof pos);
                    prim->vt->normal_at(prim, pos, out->normal);
                    out->brdf = prim->vt->brdf(prim);
                    out->distance = dd;
                }
            }
            if (len + 2 > cap) {
                cap *= 2;
                stack = (const o_bvh **)realloc(stack, sizeof(*stack) * cap);
            }
            if (root->left) stack[len++] = root->left;
            if (root->right) stack[len++] = root->right;
        }
    }
    free(stack);
}

void o_bvh_free(o_bvh *b) {
    if (!b) return;
    for (size_t i = 0; i < b->n_contents; i++) o_prim_free(b->base_contents[i]);
    free(b->base_contents);
    o_bvh_free(b->left);
    o_bvh_free(b->right);
    free(b);
}

void o_brute_force_intersect(o_prim *const *prims, size_t n, const double o[3],
                             const double dir[3], o_hit *out) {
    int have = 0;
    double closest_distance = 0.0;
    out->hit = 0; out->prim_id = O_MISS; out->brdf = NULL; out->distance = 0.0;
    out->pos[0] = out->pos[1] = out->pos[2] = 0.0;
    out->normal[0] = out->normal[1] = out->normal[2] = 0.0;
    for (size_t k = 0; k < n; k++) {
        const o_prim *prim = prims[k];
        double pos[3];
        if (!prim->vt->intersect(prim, o, dir, pos)) continue;
        double dd = o_d(pos, o);
        if (!have || dd < closest_distance) {
            have = 1;
            closest_distance = dd;
            out->hit = 1;
            out->prim_id = prim->id;
            memcpy(out->pos, pos, sizeof pos);
            prim->vt->normal_at(prim, pos, out->normal);
            out->brdf = prim->vt->brdf(prim);
            out->distance = dd;
        }
    }
}

size_t o_bvh_count_nodes(const o_bvh *b) {
    return b ? 1 + o_bvh_count_nodes(b->left) + o_bvh_count_nodes(b->right) : 0;
}
size_t o_bvh_count_leaves(const o_bvh *b) {
    if (!b) return 0;
    if (!b->left && !b->right) return 1;
    return o_bvh_count_leaves(b->left) + o_bvh_count_leaves(b->right);
}
size_t o_bvh_max_leaf_size(const o_bvh *b) {
    if (!b) return 0;
    size_t m = b->n_contents, l = o_bvh_max_leaf_size(b->left), r = o_bvh_max_leaf_size(b->right);
    if (l > m) m = l;
    if (r > m) m = r;
    return m;
}
size_t o_bvh_depth(const o_bvh *b) {
    if (!b) return 0;
    size_t l = o_bvh_depth(b->left), r = o_bvh_depth(b->right);
    return 1 + (l > r ? l : r);
}
size_t o_bvh_count_prims(const o_bvh *b) {
    return b ? b->n_contents + o_bvh_count_prims(b->left) + o_bvh_count_prims(b->right) : 0;
}

static void dump_rec(const o_bvh *b, double *shells, uint32_t *n_contents, uint32_t *prim_ids,
                     uint32_t *is_leaf, size_t *ni, size_t *pi) {
    if (!b) return;
    size_t i = (*ni)++;
    memcpy(shells + 4 * i, b->bounding_shell, sizeof(double) * 4);
    n_contents[i] = (uint32_t)b->n_contents;
    is_leaf[i] = (!b->left && !b->right) ? 1u : 0u;
    for (size_t k = 0; k < b->n_contents; k++) prim_ids[(*pi)++] = b->base_contents[k]->id;
    dump_rec(b->right, shells, n_contents, prim_ids, is_leaf, ni, pi);
    dump_rec(b->left, shells, n_contents, prim_ids, is_leaf, ni, pi);
}
size_t o_bvh_dump_preorder(const o_bvh *b, double *shells, uint32_t *n_contents,
                           uint32_t *prim_ids, uint32_t *is_leaf) {
    size_t ni = 0, pi = 0;
    dump_rec(b, shells, n_contents, prim_ids, is_leaf, &ni, &pi);
    return ni;
}

/* ====================================================================== */
/* counter-based uniforms                                                 */
/* ====================================================================== */

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11). */
void o_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void o_uniform2(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint32_t d, double u[2]) {
    uint32_t ctr[4] = {a, b, c, d};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t r[4];
    o_philox4x32_10(ctr, key, r);
    uint64_t x = ((uint64_t)r[0] << 32) | r[1];
    uint64_t y = ((uint64_t)r[2] << 32) | r[3];
    u[0] = (double)(x >> 11) * (1.0 / 9007199254740992.0);
    u[1] = (double)(y >> 11) * (1.0 / 9007199254740992.0);
}

/* ====================================================================== */
/* variable-glossy lobe of the JavaScript tracer                          */
/* ====================================================================== */

/* geometry.js:300-318 */
void o_gauss_dir(const double norm_dir[3], double lobe_size, const double normal[3], uint64_t seed,
                 uint32_t a, uint32_t b, uint32_t c, uint32_t d, double out[3]) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t block = 0;
    while (block + 1 < O_GLOSS_MAX_BLOCKS) {
        uint32_t ctr[4] = {a, b, c, d | (block << 8)}, r[4];
        o_philox4x32_10(ctr, key, r);
        block++;
        /* geometry.js:304-307: a uniform point of the ball of radius 1/2 */
        double rd[3] = {(double)r[0] * (1.0 / 4294967296.0) - 0.5, (double)r[1] * (1.0 / 4294967296.0) - 0.5,
                        (double)r[2] * (1.0 / 4294967296.0) - 0.5};
        if (o_mag(rd) > 0.5) continue;
        /* geometry.js:309-310: its component perpendicular to the lobe axis, normalised */
        double pr[3], perp[3];
        o_scale(o_dot(rd, norm_dir) / o_dot(norm_dir, norm_dir), norm_dir, pr); /* project(), geometry.js:1128 */
        o_sub(rd, pr, perp);
        if (o_normalize(perp, perp) != 0) continue;
        /* geometry.js:312 with gauss-random: sqrt(-2 ln u) cos(2 pi v) */
        ctr[3] = d | (block << 8);
        o_philox4x32_10(ctr, key, r);
        block++;
        uint64_t x = ((uint64_t)r[0] << 32) | r[1];
        uint64_t y = ((uint64_t)r[2] << 32) | r[3];
        double u = (double)((x >> 11) + 1) * (1.0 / 9007199254740992.0); /* (0, 1] */
        double v = (double)(y >> 11) * (1.0 / 9007199254740992.0);
        double g = sqrt(-2.0 * log(u)) * cos(2.0 * M_PI * v);
        double ang = g * lobe_size;
        double ca[3], sa[3], fin[3];
        o_scale(cos(ang), norm_dir, ca);
        o_scale(sin(ang), perp, sa);
        o_add(ca, sa, fin);
        if (o_dot(fin, normal) >= 0.0) { /* geometry.js:315 */
            out[0] = fin[0]; out[1] = fin[1]; out[2] = fin[2];
            return;
        }
    }
    out[0] = norm_dir[0]; out[1] = norm_dir[1]; out[2] = norm_dir[2];
}

/* geometry.js:268-272: ref = normalize(reflect(inVec, normal)), reflect = in - 2 project(in, normal)
   (geometry.js:1132-1134); lobeSize = (2n)^2 with n = 1 - gloss (geometry.js:178-181) */
void o_brdf_sample_glossy(const o_brdf *self, const double in_dir[3], const double normal[3], uint64_t seed,
                          uint32_t pixel, uint32_t sample, uint32_t j, uint32_t depth, double out[3]) {
    double pr[3], two[3], ref[3];
    o_scale(o_dot(in_dir, normal) / o_dot(normal, normal), normal, pr);
    o_scale(2.0, pr, two);
    o_sub(in_dir, two, ref);
    if (o_normalize(ref, ref) != 0) {
        out[0] = out[1] = out[2] = NAN;
        return;
    }
    double n = 1.0 - self->gloss;
    double lobe = (2.0 * n) * (2.0 * n);
    o_gauss_dir(ref, lobe, normal, seed, pixel, sample, j | (O_STREAM_GLOSS << 8), depth, out);
}

/* ====================================================================== */
/* tracer.rs                                                              */
/* ====================================================================== */

/* tracer.rs:8-81 */
void o_trace_ray(const double origin[3], const double dir[3], const o_bvh *bvh, uint64_t seed,
                 uint32_t pixel, uint32_t sample, uint32_t j, double out[3], uint64_t *ray_count,
                 o_stats *stats) {
    static const o_brdf NONE = {{0.0, 0.0, 0.0}, 0.0, 1.0, 0.0};
    double angle_attenuations[O_MAX_RAY_DEPTH] = {0};
    const o_brdf *brdfs[O_MAX_RAY_DEPTH];
    for (int k = 0; k < O_MAX_RAY_DEPTH; k++) brdfs[k] = &NONE;
    int i = 0;
    int encountered_emissive_feature = 0;
    double w_o[3] = {origin[0], origin[1], origin[2]};
    double w_d[3] = {dir[0], dir[1], dir[2]};

    int depth = 0;
    while (depth < O_MAX_RAY_DEPTH) {
        o_hit h;
        o_bvh_intersect(bvh, w_o, w_d, &h, stats);
        if (ray_count) (*ray_count)++;
        if (!h.hit) break;
        if (h.brdf->emissivity > 0.0) encountered_emissive_feature = 1;
        if (h.brdf->gloss > 0.0) { /* the JavaScript tracer's lobe; brdf.rs has only the diffuse branch */
            double nd[3];
            o_brdf_sample_glossy(h.brdf, w_d, h.normal, seed, pixel, sample, j, (uint32_t)depth, nd);
            memcpy(w_d, nd, sizeof nd);
        } else {
            double u[2];
            o_uniform2(seed, pixel, sample, j | (O_STREAM_BSDF << 8), (uint32_t)depth, u);
            o_brdf_sample(h.brdf, w_d, h.normal, u[0], u[1], w_d);
        }
        memcpy(w_o, h.pos, sizeof w_o);
        angle_attenuations[i] = o_dot(w_d, h.normal);
        brdfs[i] = h.brdf;
        i += 1;
        depth += 1;
    }

    if (!encountered_emissive_feature) {
        out[0] = out[1] = out[2] = 0.0;
        return;
    }
    double sr[3] = {0.0, 0.0, 0.0};
    while (i >= 1) {
        i -= 1;
        sr[0] *= brdfs[i]->reflection_spectrum[0];
        sr[1] *= brdfs[i]->reflection_spectrum[1];
        sr[2] *= brdfs[i]->reflection_spectrum[2];
        o_scale(angle_attenuations[i], sr, sr);
        double em[3];
        o_scale(brdfs[i]->emissivity, brdfs[i]->reflection_spectrum, em);
        o_add(sr, em, sr);
    }
    out[0] = sr[0]; out[1] = sr[1]; out[2] = sr[2];
}

/* ====================================================================== */
/* camera.rs                                                              */
/* ====================================================================== */

/* camera.rs:16-78, width x height generalisation (see header) */
void o_ray_bundle(const o_camera *cam, size_t width, size_t height, uint64_t seed, double *out) {
    double sensor_width = cam->sensor_diagonal / sqrt(2.0);
    /* same pixel pitch on both axes; equals sensor_width exactly when height == width */
    double sensor_height = sensor_width * ((double)height / (double)width);
    double tmp[3], lens_center[3];
    double v0[3] = {0.0, 0.0, cam->sensor_to_lens};
    o_mat_mult_3(cam->dir, v0, tmp);
    o_add(cam->loc, tmp, lens_center);

    const double offsets[5][2] = {{0.0, 0.0},
                                  {-2.0 / 3.0, -2.0 / 3.0},
                                  {-2.0 / 3.0, 2.0 / 3.0},
                                  {2.0 / 3.0, 2.0 / 3.0},
                                  {2.0 / 3.0, -2.0 / 3.0}};
    size_t w = 0;
    for (size_t col = 0; col < height; col++) {     /* image y (main.rs:57) */
        for (size_t row = 0; row < width; row++) {  /* image x */
            for (int j = 0; j < 5; j++) {
                double offset_x = offsets[j][0];
                double offset_y = offsets[j][1];
                double sensor_loc_unit_x = -(2.0 * ((double)col + offset_x) / (double)height - 1.0);
                double sensor_loc_unit_y = (2.0 * ((double)row + offset_y) / (double)width - 1.0);
                double sensor_loc_raw[3] = {-sensor_loc_unit_x * sensor_height / 2.0,
                                            -sensor_loc_unit_y * sensor_width / 2.0, 0.0};
                double sensor_loc[3];
                o_mat_mult_3(cam->dir, sensor_loc_raw, tmp);
                o_add(cam->loc, tmp, sensor_loc);

                double naive_ray_dir[3], image_loc[3];
                o_sub(lens_center, sensor_loc, tmp);
                o_normalize(tmp, naive_ray_dir);
                o_scale(cam->focal_length, naive_ray_dir, tmp);
                o_add(lens_center, tmp, image_loc);

                double u[2], disk[2];
                o_uniform2(seed, (uint32_t)(col * width + row), 0u, (uint32_t)j | (O_STREAM_LENS << 8), 0u, u);
                o_sample_disk(cam->aperture_radius, u[0], u[1], disk);
                double lens_pt_raw[3] = {disk[0], disk[1], cam->sensor_to_lens};
                double lens_pt[3], ray_dir[3];
                o_mat_mult_3(cam->dir, lens_pt_raw, tmp);
                o_add(cam->loc, tmp, lens_pt);
                o_sub(image_loc, lens_pt, tmp);
                o_normalize(tmp, ray_dir);

                out[w++] = lens_pt[0]; out[w++] = lens_pt[1]; out[w++] = lens_pt[2];
                out[w++] = ray_dir[0]; out[w++] = ray_dir[1]; out[w++] = ray_dir[2];
            }
        }
    }
}

/* Rust `as u8` on f64: saturating, NaN -> 0 */
static uint8_t f64_as_u8(double v) {
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}
/* camera.rs:80-90 */
void o_spectral_radiance_to_color(const double sr[3], uint8_t out[4]) {
    for (int k = 0; k < 3; k++) {
        double f = 2.0 / (1.0 + pow(M_E, -2.5 * sr[k])) - 1.0;
        out[k] = f64_as_u8(floor(f * 255.0));
    }
    out[3] = 255;
}

/* ====================================================================== */
/* main.rs per-pixel loop, scene, OBJ                                      */
/* ====================================================================== */

/* main.rs:50-76 */
void o_render_pixels(const o_bvh *env, const double *rays, size_t width, size_t height,
                     uint32_t spp, uint64_t seed, size_t thread_index, size_t n_threads,
                     double *radiance_mean, uint8_t *rgba, uint64_t *ray_count, o_stats *stats) {
    for (size_t y = 0; y < height; y++) {
        for (size_t x = 0; x < width; x++) {
            if ((y * width + x) % n_threads != thread_index) continue;
            double total[3] = {0.0, 0.0, 0.0};
            size_t start_index = (y * width + x) * 30;
            for (uint32_t sample = 0; sample < spp; sample++) {
                for (uint32_t j = 0; j < 5; j++) {
                    const double *sp = rays + start_index + 6 * j;
                    double ro[3] = {sp[0], sp[1], sp[2]};
                    double rd[3] = {sp[3], sp[4], sp[5]};
                    double sr[3];
                    o_trace_ray(ro, rd, env, seed, (uint32_t)(y * width + x), sample, j, sr, ray_count, stats);
                    o_add(total, sr, total);
                }
            }
            o_scale(1.0 / (5.0 * (double)spp), total, total);
            if (radiance_mean) memcpy(radiance_mean + (y * width + x) * 3, total, sizeof total);
            o_spectral_radiance_to_color(total, rgba + (y * width + x) * 4);
        }
    }
}

/* main.rs:117-170 */
o_prim **o_hippo_scene_prims(const double *vertices, size_t nv, const uint32_t *faces, size_t nf,
                             size_t *n_out) {
    const double z = -1.0, x = 2.0, y = 2.0, h = 0.5;
    o_prim **objs = (o_prim **)malloc(sizeof(o_prim *) * (4 + nf));
    size_t n = 0;
    {
        double pts[12] = {x, y, z, -x, y, z, -x, -y, z, x, -y, z};
        o_brdf b = {{0.75, 0.95, 0.5}, 0.0, 1.0, 0.0};
        double dv[3] = {0.0, 0.0, 1.0};
        objs[n++] = o_polygon_new_cached(pts, 4, &b, dv);
    }
    {
        double pts[12] = {-x - 1.0, y, z + h, -x - 1.0, -y, z + h, -x - 1.0, -y, -z + h, -x - 1.0, y, -z + h};
        o_brdf b = {{1.0, 1.0, 1.0}, 25.0, 1.0, 0.0};
        double dv[3] = {1.0, 0.0, 0.0};
        objs[n++] = o_polygon_new_cached(pts, 4, &b, dv);
    }
    {
        double pts[12] = {x, y, z, x, -y, z, x, -y, z + 3.0, x, y, z + 3.0};
        o_brdf b = {{0.95, 0.95, 0.95}, 0.0, 1.0, 0.0};
        double dv[3] = {-1.0, 0.0, 0.0};
        objs[n++] = o_polygon_new_cached(pts, 4, &b, dv);
    }
    {
        double pts[12] = {x, y, z, -x, y, z, -x, y, -z + 1.0, x, y, -z + 1.0};
        o_brdf b = {{0.95, 0.95, 0.95}, 0.0, 1.0, 0.0};
        double dv[3] = {0.0, -1.0, 0.0};
        objs[n++] = o_polygon_new_cached(pts, 4, &b, dv);
    }
    /* main.rs:193: the shift is applied to the parsed coordinate */
    double *shifted = (double *)malloc(sizeof(double) * 3 * (nv ? nv : 1));
    for (size_t i = 0; i < nv; i++) {
        shifted[3 * i] = vertices[3 * i];
        shifted[3 * i + 1] = vertices[3 * i + 1];
        shifted[3 * i + 2] = vertices[3 * i + 2] - 0.717;
    }
    o_brdf fb = {{0.65, 0.65, 0.65}, 0.0, 1.0, 0.0};
    for (size_t f = 0; f < nf; f++) {
        double pts[9];
        for (int k = 0; k < 3; k++) memcpy(pts + 3 * k, shifted + 3 * (size_t)faces[3 * f + k], 24);
        double e0[3], e1[3], dir[3];
        o_sub(pts + 3, pts, e0);
        o_sub(pts + 6, pts + 3, e1);
        o_cross(e0, e1, dir); /* main.rs:222-225 */
        o_prim *p = o_polygon_new_cached(pts, 3, &fb, dir);
        if (!p) continue; /* reference panics on a degenerate face; none in hippo */
        objs[n++] = p;
    }
    free(shifted);
    for (size_t i = 0; i < n; i++) objs[i]->id = (uint32_t)i;
    *n_out = n;
    return objs;
}

/* main.rs:172-240. Triangular faces only are returned in `faces`; max_face_len
   reports the longest face seen so callers can reject non-triangle files. */
int o_parse_obj(const char *text, size_t len, double **vertices, size_t *nv, uint32_t **faces,
                size_t *nf, size_t *max_face_len) {
    size_t vcap = 1024, fcap = 1024;
    double *v = (double *)malloc(sizeof(double) * 3 * vcap);
    uint32_t *f = (uint32_t *)malloc(sizeof(uint32_t) * 3 * fcap);
    size_t n_v = 0, n_f = 0, maxlen = 0;
    size_t pos = 0;
    int rc = 0;
    char *line = NULL;
    size_t linecap = 0;
    while (pos <= len && rc == 0) {
        size_t e = pos;
        while (e < len && text[e] != '\n') e++;
        size_t ll = e - pos;
        if (ll + 1 > linecap) { linecap = 2 * (ll + 1); line = (char *)realloc(line, linecap); }
        /* line.replace("  ", " "): one left-to-right pass, non-overlapping */
        size_t m = 0;
        for (size_t i = 0; i < ll;) {
            if (i + 1 < ll && text[pos + i] == ' ' && text[pos + i + 1] == ' ') { line[m++] = ' '; i += 2; }
            else line[m++] = text[pos + i++];
        }
        int is_v = ll >= 2 && text[pos] == 'v' && text[pos + 1] == ' ';
        int is_vn = ll >= 3 && text[pos] == 'v' && text[pos + 1] == 'n' && text[pos + 2] == ' ';
        int is_f = ll >= 2 && text[pos] == 'f' && text[pos + 1] == ' ';
        /* trim() */
        size_t s = 0;
        while (s < m && (line[s] == ' ' || line[s] == '\t' || line[s] == '\r' || line[s] == '\n' || line[s] == '\f' || line[s] == '\v')) s++;
        while (m > s && (line[m - 1] == ' ' || line[m - 1] == '\t' || line[m - 1] == '\r' || line[m - 1] == '\n' || line[m - 1] == '\f' || line[m - 1] == '\v')) m--;
        line[m] = 0;
        if (is_v || is_vn) {
            double c[3] = {0.0, 0.0, 0.0};
            int count = 0;
            char *p = line + s;
            while (1) { /* split(" ") */
                char *q = strchr(p, ' ');
                if (q) *q = 0;
                if (count > 0) {
                    if (count > 3) { rc = -1; break; }
                    char *end;
                    c[count - 1] = strtod(p, &end);
                    if (end == p || *end != 0) { rc = -1; break; }
                }
                count++;
                if (!q) break;
                p = q + 1;
            }
            if (rc == 0 && is_v) {
                if (n_v == vcap) { vcap *= 2; v = (double *)realloc(v, sizeof(double) * 3 * vcap); }
                v[3 * n_v] = c[0]; v[3 * n_v + 1] = c[1]; v[3 * n_v + 2] = c[2];
                n_v++;
            }
        } else if (is_f) {
            uint32_t idx[3] = {0, 0, 0};
            size_t count = 0;
            char *p = line + s;
            while (1) {
                char *q = strchr(p, ' ');
                if (q) *q = 0;
                if (count > 0) {
                    /* sub_items[0] vertex, sub_items[2] normal must exist (main.rs:210-211) */
                    char *s1 = strchr(p, '/');
                    char *s2 = s1 ? strchr(s1 + 1, '/') : NULL;
                    if (!s1 || !s2) { rc = -1; break; }
                    *s1 = 0;
                    char *end;
                    unsigned long vi = strtoul(p, &end, 10);
                    if (end == p || *end != 0 || vi == 0 || vi > n_v) { rc = -1; break; }
                    char *s3 = strchr(s2 + 1, '/');
                    if (s3) *s3 = 0;
                    (void)strtoul(s2 + 1, &end, 10);
                    if (end == s2 + 1 || *end != 0) { rc = -1; break; }
                    if (count <= 3) idx[count - 1] = (uint32_t)(vi - 1);
                }
                count++;
                if (!q) break;
                p = q + 1;
            }
            if (rc == 0) {
                size_t flen = count - 1;
                if (flen > maxlen) maxlen = flen;
                if (flen < 3) { rc = -1; }
                else {
                    if (n_f == fcap) { fcap *= 2; f = (uint32_t *)realloc(f, sizeof(uint32_t) * 3 * fcap); }
                    f[3 * n_f] = idx[0]; f[3 * n_f + 1] = idx[1]; f[3 * n_f + 2] = idx[2];
                    n_f++;
                }
            }
        }
        pos = e + 1;
    }
    free(line);
    if (rc != 0) { free(v); free(f); return rc; }
    *vertices = v; *nv = n_v; *faces = f; *nf = n_f;
    if (max_face_len) *max_face_len = maxlen;
    return 0;
}

/* ====================================================================== */
/* batch helpers                                                          */
/* ====================================================================== */

void o_intersect_batch(const o_bvh *bvh, const double *rays, size_t n, uint32_t *prim_ids,
                       double *dists, double *pos, double *normals, o_stats *stats) {
    for (size_t i = 0; i < n; i++) {
        o_hit h;
        o_bvh_intersect(bvh, rays + 6 * i, rays + 6 * i + 3, &h, stats);
        if (prim_ids) prim_ids[i] = h.prim_id;
        if (dists) dists[i] = h.distance;
        if (pos) memcpy(pos + 3 * i, h.pos, 24);
        if (normals) memcpy(normals + 3 * i, h.normal, 24);
    }
}

void o_brute_force_batch(o_prim *const *prims, size_t n_prims, const double *rays, size_t n,
                         uint32_t *prim_ids, double *dists) {
    for (size_t i = 0; i < n; i++) {
        o_hit h;
        o_brute_force_intersect(prims, n_prims, rays + 6 * i, rays + 6 * i + 3, &h);
        if (prim_ids) prim_ids[i] = h.prim_id;
        if (dists) dists[i] = h.distance;
    }
}

void o_trace_batch(const o_bvh *bvh, const double *rays, size_t n, uint64_t seed,
                   const uint32_t *keys, double *out, uint64_t *n_rays) {
    for (size_t i = 0; i < n; i++)
        o_trace_ray(rays + 6 * i, rays + 6 * i + 3, bvh, seed, keys[3 * i], keys[3 * i + 1],
                    keys[3 * i + 2], out + 3 * i, n_rays, NULL);
}

void o_bvh_collect_prims(const o_bvh *b, o_prim **out, size_t *n) {
    if (!b) return;
    for (size_t k = 0; k < b->n_contents; k++) out[(*n)++] = b->base_contents[k];
    o_bvh_collect_prims(b->right, out, n);
    o_bvh_collect_prims(b->left, out, n);
}
int o_prim_intersect(const o_prim *p, const double o[3], const double d[3], double pos[3]) {
    return p->vt->intersect(p, o, d, pos);
}
void o_prim_normal_at(const o_prim *p, const double pt[3], double n[3]) { p->vt->normal_at(p, pt, n); }
void o_prim_package(const o_prim *p, double out[4]) { p->vt->package(p, out); }

/* ====================================================================== */
/* timed legs                                                             */
/* ====================================================================== */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    const double *vertices; size_t nv; const uint32_t *faces; size_t nf;
    const double *rays; size_t n_rays; size_t t, T;
    uint32_t *prim_ids; double *dists;
    /* render */
    size_t width, height; uint32_t spp; uint64_t seed; uint8_t *rgba; uint64_t ray_count;
    pthread_barrier_t *bar; double t0, t1;
    size_t stride; o_stats stats;
    int count; /* render leg: keep the visit counters (they cost time: only when the caller asks for them) */
} leg_arg;

static void *intersect_leg(void *vp) {
    leg_arg *a = (leg_arg *)vp;
    size_t n;
    o_prim **prims = o_hippo_scene_prims(a->vertices, a->nv, a->faces, a->nf, &n);
    o_bvh *env = o_bvh_from_list(prims, n); /* private scene per thread, main.rs:46 */
    pthread_barrier_wait(a->bar);
    a->t0 = now_s();
    for (size_t i = a->t; i < a->n_rays; i += a->T) {
        o_hit h;
        o_bvh_intersect(env, a->rays + 6 * i, a->rays + 6 * i + 3, &h, NULL);
        if (a->prim_ids) a->prim_ids[i] = h.prim_id;
        if (a->dists) a->dists[i] = h.distance;
    }
    a->t1 = now_s();
    pthread_barrier_wait(a->bar);
    o_bvh_free(env);
    return NULL;
}

double o_time_intersect_batch(const double *vertices, size_t nv, const uint32_t *faces, size_t nf,
                              const double *rays, size_t n_rays, size_t T, uint32_t *prim_ids,
                              double *dists) {
    if (T == 0) T = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * T);
    leg_arg *args = (leg_arg *)calloc(T, sizeof(leg_arg));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)T);
    for (size_t t = 0; t < T; t++) {
        leg_arg a = {vertices, nv, faces, nf, rays, n_rays, t, T, prim_ids, dists, 0, 0, 0, 0, NULL, 0, &bar, 0, 0,
                     0, {0, 0, 0, 0, 0, 0, 0}, 0};
        args[t] = a;
        pthread_create(&th[t], NULL, intersect_leg, &args[t]);
    }
    double t0 = 1e300, t1 = 0;
    for (size_t t = 0; t < T; t++) {
        pthread_join(th[t], NULL);
        if (args[t].t0 < t0) t0 = args[t].t0;
        if (args[t].t1 > t1) t1 = args[t].t1;
    }
    pthread_barrier_destroy(&bar);
    free(th); free(args);
    return t1 - t0;
}

static void *render_leg(void *vp) {
    leg_arg *a = (leg_arg *)vp;
    size_t n;
    o_prim **prims = o_hippo_scene_prims(a->vertices, a->nv, a->faces, a->nf, &n);
    o_bvh *env = o_bvh_from_list(prims, n);
    pthread_barrier_wait(a->bar);
    a->t0 = now_s();
    o_render_pixels(env, a->rays, a->width, a->height, a->spp, a->seed, a->t, a->stride, NULL, a->rgba,
                    &a->ray_count, a->count ? &a->stats : NULL);
    a->t1 = now_s();
    pthread_barrier_wait(a->bar);
    o_bvh_free(env);
    return NULL;
}

double o_time_render(const double *vertices, size_t nv, const uint32_t *faces, size_t nf,
                     const o_camera *cam, size_t width, size_t height, uint32_t spp, uint64_t seed,
                     size_t T, size_t pixel_stride, uint8_t *rgba, uint64_t *ray_count, o_stats *stats) {
    if (T == 0) T = 1;
    if (pixel_stride < T) pixel_stride = T;
    double *rays = (double *)malloc(sizeof(double) * width * height * 30);
    o_ray_bundle(cam, width, height, seed, rays);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * T);
    leg_arg *args = (leg_arg *)calloc(T, sizeof(leg_arg));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)T);
    for (size_t t = 0; t < T; t++) {
        leg_arg a = {vertices, nv, faces, nf, rays, 0, t, T, NULL, NULL, width, height, spp, seed, rgba, 0, &bar, 0, 0,
                     pixel_stride, {0, 0, 0, 0, 0, 0, 0}, stats != NULL};
        args[t] = a;
        pthread_create(&th[t], NULL, render_leg, &args[t]);
    }
    double t0 = 1e300, t1 = 0;
    uint64_t rc = 0;
    for (size_t t = 0; t < T; t++) {
        pthread_join(th[t], NULL);
        if (args[t].t0 < t0) t0 = args[t].t0;
        if (args[t].t1 > t1) t1 = args[t].t1;
        rc += args[t].ray_count;
        if (stats) {
            stats->nodes_popped += args[t].stats.nodes_popped;
            stats->nodes_pruned += args[t].stats.nodes_pruned;
            stats->shell_tests += args[t].stats.shell_tests;
            stats->shell_hits += args[t].stats.shell_hits;
            stats->poly_tests += args[t].stats.poly_tests;
            stats->sphere_tests += args[t].stats.sphere_tests;
            stats->prim_hits += args[t].stats.prim_hits;
        }
    }
    if (ray_count) *ray_count = rc;
    pthread_barrier_destroy(&bar);
    free(th); free(args); free(rays);
    return t1 - t0;
}
