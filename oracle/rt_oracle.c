/*
 * rt_oracle.c — CPU oracle: plain-C restatement of the reference's lane-1 render path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see rt_oracle.h).  Compile with -ffp-contract=off and without
 * -ffast-math: the reference's lane-1 madd/msub are unfused (SIMD_floats.h:47-48).
 *
 * Lane-1 conventions restated:
 *   dot(l,r)   = l.x*r.x + (l.y*r.y + l.z*r.z)                         SIMD_Vector3.h:53-55
 *   cross      = (ly*rz - lz*ry, lz*rx - lx*rz, lx*ry - ly*rx)          SIMD_Vector3.h:57-63
 *   v / f      = v * (1.0f / f)                                        SIMD_Vector3.h:137,153
 *   normalize  = v * (1.0f / sqrtf(dot(v,v)))                          SIMD_Vector3.h:44-51
 *   min(a,b)   = a < b ? a : b ; max(a,b) = a > b ? a : b              SIMD_floats.h:32-33
 *   float_to_int = cvtss2si (round to nearest even)                    Util.h:39-41
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <xmmintrin.h>

#define RAY_EPSILON 0.005f            /* Ray.h:5 */
#define ONE_OVER_PI 0.31830988618f    /* Util.h:9 */
#define ONE_OVER_TWO_PI 0.15915494309f/* Util.h:12 */

#define AXIS_X_BITS (1u << 30)        /* BVHNode.h:5-8 */
#define AXIS_Y_BITS (2u << 30)
#define AXIS_Z_BITS (3u << 30)
#define AXIS_MASK   (3u << 30)

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3p(const float * p) { v3 r = { p[0], p[1], p[2] }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vmuls(v3 a, float f) { return V3(a.x * f, a.y * f, a.z * f); }   /* vector*float and float*vector */
static inline v3 vadds(v3 a, float f) { return V3(a.x + f, a.y + f, a.z + f); }   /* float + vector */
static inline v3 vneg(v3 a) { return V3(0.0f - a.x, 0.0f - a.y, 0.0f - a.z); }    /* SIMD_Vector3.h:133-136 */
static inline float vdot(v3 l, v3 r) { return l.x * r.x + (l.y * r.y + l.z * r.z); }
static inline v3 vcross(v3 l, v3 r) { return V3(l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x); }
static inline v3 vdivs(v3 a, float f) { float inv = 1.0f / f; return vmuls(a, inv); }
static inline v3 vnormalize(v3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return vmuls(a, inv); }
static inline v3 vmadd_s(v3 a, float b, v3 c) { return V3(a.x * b + c.x, a.y * b + c.y, a.z * b + c.z); }  /* madd(vec, float, vec) */
static inline v3 vmadd_v(v3 a, v3 b, v3 c) { return V3(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z); }
static inline float fmin_ref(float a, float b) { return a < b ? a : b; }
static inline float fmax_ref(float a, float b) { return a > b ? a : b; }

/* Util::float_to_int, Util.h:39-41 */
int32_t orc_float_to_int(float x) { return _mm_cvtss_si32(_mm_load_ss(&x)); }
/* SIMD_float_to_int at lane 1 (SIMD.h:11): C truncation, i.e. cvttss2si */
static inline int32_t float_to_int_trunc(float x) { return _mm_cvttss_si32(_mm_load_ss(&x)); }

/* Matrix4::transform_position / transform_direction, SIMD versions, Matrix4.h:39-85 */
static inline v3 transform_position(const float * c, v3 d) {
    return V3(c[0] * d.x + (c[1] * d.y + (c[2]  * d.z + c[3])),
              c[4] * d.x + (c[5] * d.y + (c[6]  * d.z + c[7])),
              c[8] * d.x + (c[9] * d.y + (c[10] * d.z + c[11])));
}
static inline v3 transform_direction(const float * c, v3 d) {
    return V3(c[0] * d.x + (c[1] * d.y + c[2]  * d.z),
              c[4] * d.x + (c[5] * d.y + c[6]  * d.z),
              c[8] * d.x + (c[9] * d.y + c[10] * d.z));
}

typedef struct {          /* Ray.h:4-17 */
    v3 origin, direction;
    v3 dO_dx, dO_dy, dD_dx, dD_dy;
} ray_t;

typedef struct {          /* RayHit.h:4-36 */
    int   hit;
    float distance;
    v3    point, normal;
    int   material_id;
    float u, v;
    float ds_dx, ds_dy, dt_dx, dt_dy;
    v3    dO_dx, dO_dy, dN_dx, dN_dy;
    int   bvh_steps;      /* RayHit.h:24-26 (BVH_VISUALIZE_HEATMAP) */
} hit_t;

typedef struct {
    rtx_stats         stats;
    rtx_work_counters work;
} counters_t;

/* ------------------------------------------------------------------------------------------------
 * AABB::intersect, AABB.cpp:38-52
 * ---------------------------------------------------------------------------------------------- */
static inline int aabb_intersect(const rtx_bvh_node * n, v3 origin, v3 inv_dir, float max_distance) {
    v3 t0 = vmul(vsub(v3p(n->aabb_min), origin), inv_dir);
    v3 t1 = vmul(vsub(v3p(n->aabb_max), origin), inv_dir);
    v3 tmin = V3(fmin_ref(t0.x, t1.x), fmin_ref(t0.y, t1.y), fmin_ref(t0.z, t1.z));
    v3 tmax = V3(fmax_ref(t0.x, t1.x), fmax_ref(t0.y, t1.y), fmax_ref(t0.z, t1.z));
    float t_near = fmax_ref(fmax_ref(RAY_EPSILON, tmin.x), fmax_ref(tmin.y, tmin.z));
    float t_far  = fmin_ref(fmin_ref(max_distance, tmax.x), fmin_ref(tmax.y, tmax.z));
    return t_near < t_far;
}

int orc_aabb_intersect(const rtx_bvh_node * node, const float origin[3], const float inv_direction[3], float max_distance) {
    return aabb_intersect(node, v3p(origin), v3p(inv_direction), max_distance);
}

/* BVHNode::should_visit_left_first, BVHNode.h:30-40 */
static inline int visit_left_first(const rtx_bvh_node * n, v3 dir, int strategy) {
    if (strategy == RTX_TRAVERSE_NAIVE) return 1;
    switch ((uint32_t)n->count & AXIS_MASK) {
        case AXIS_X_BITS: return dir.x > 0.0f;
        case AXIS_Y_BITS: return dir.y > 0.0f;
        case AXIS_Z_BITS: return dir.z > 0.0f;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * BottomLevelBVH::triangle_trace, BottomLevelBVH.cpp:214-309
 * ---------------------------------------------------------------------------------------------- */
static void triangle_trace(const orc_blas * b, int index, const ray_t * ray, hit_t * h, const float * world) {
    const rtx_triangle_hot * th = &b->tri_hot[index];
    v3 edge_1 = v3p(th->position_edge_1);
    v3 edge_2 = v3p(th->position_edge_2);

    v3    hh = vcross(ray->direction, edge_2);
    float a  = vdot(edge_1, hh);
    float f  = 1.0f / a;
    v3    s  = vsub(ray->origin, v3p(th->position_0));
    float u  = f * vdot(s, hh);
    if (!((u > 0.0f) & (u < 1.0f))) return;

    v3    q = vcross(s, edge_1);
    float v = f * vdot(ray->direction, q);
    if (!((v > 0.0f) & ((u + v) < 1.0f))) return;

    float t = f * vdot(edge_2, q);
    if (!((t > RAY_EPSILON) & (t < h->distance))) return;

    const rtx_triangle_cold * tc = &b->tri_cold[index];
    h->hit      = 1;
    h->distance = t;

    v3 n_edge_1 = v3p(tc->normal_edge_1);
    v3 n_edge_2 = v3p(tc->normal_edge_2);
    /* Math::barycentric: base + u*edge1 + v*edge2, Math.h:21-24 */
    v3 n = vadd(vadd(v3p(tc->normal_0), vmuls(n_edge_1, u)), vmuls(n_edge_2, v));

    h->point  = transform_position (world, vadd(ray->origin, vmuls(ray->direction, t)));
    h->normal = transform_direction(world, vnormalize(n));
    h->material_id = b->material_offset + tc->material_id;

    float te1x = tc->tex_coord_edge_1[0], te1y = tc->tex_coord_edge_1[1];
    float te2x = tc->tex_coord_edge_2[0], te2y = tc->tex_coord_edge_2[1];
    h->u = (tc->tex_coord_0[0] + u * te1x) + v * te2x;   /* SIMD_Vector2 barycentric */
    h->v = (tc->tex_coord_0[1] + u * te1y) + v * te2y;

    /* Ray differentials, BottomLevelBVH.cpp:275-308 */
    float one_over_k = 1.0f / vdot(vcross(edge_1, edge_2), ray->direction);
    v3 _q = vmadd_s(ray->dD_dx, t, ray->dO_dx);
    v3 _r = vmadd_s(ray->dD_dy, t, ray->dO_dy);
    v3 c_u = vcross(edge_2, ray->direction);
    v3 c_v = vcross(ray->direction, edge_1);
    float du_dx = one_over_k * vdot(c_u, _q);
    float du_dy = one_over_k * vdot(c_u, _r);
    float dv_dx = one_over_k * vdot(c_v, _q);
    float dv_dy = one_over_k * vdot(c_v, _r);

    h->dO_dx = vadd(vmuls(edge_1, du_dx), vmuls(edge_2, dv_dx));
    h->dO_dy = vadd(vmuls(edge_1, du_dy), vmuls(edge_2, dv_dy));

    v3 dn_dx = vadd(vmuls(n_edge_1, du_dx), vmuls(n_edge_2, dv_dx));
    v3 dn_dy = vadd(vmuls(n_edge_1, du_dy), vmuls(n_edge_2, dv_dy));

    float n_dot_n = vdot(n, n);
    float N_denom = (1.0f / sqrtf(n_dot_n)) / n_dot_n;
    h->dN_dx = vmuls(vsub(vmuls(dn_dx, n_dot_n), vmuls(n, vdot(n, dn_dx))), N_denom);
    h->dN_dy = vmuls(vsub(vmuls(dn_dy, n_dot_n), vmuls(n, vdot(n, dn_dy))), N_denom);

    h->ds_dx = du_dx * te1x + dv_dx * te2x;
    h->ds_dy = du_dy * te1x + dv_dy * te2x;
    h->dt_dx = du_dx * te1y + dv_dx * te2y;
    h->dt_dy = du_dy * te1y + dv_dy * te2y;
}

/* BottomLevelBVH::triangle_intersect, BottomLevelBVH.cpp:311-346 */
static int triangle_intersect(const orc_blas * b, int index, v3 origin, v3 direction, float max_distance) {
    const rtx_triangle_hot * th = &b->tri_hot[index];
    v3 edge_0 = v3p(th->position_edge_1);
    v3 edge_1 = v3p(th->position_edge_2);
    v3    hh = vcross(direction, edge_1);
    float a  = vdot(edge_0, hh);
    float f  = 1.0f / a;
    v3    s  = vsub(origin, v3p(th->position_0));
    float u  = f * vdot(s, hh);
    if (!((u > 0.0f) & (u < 1.0f))) return 0;
    v3    q = vcross(s, edge_0);
    float v = f * vdot(direction, q);
    if (!((v > 0.0f) & ((u + v) < 1.0f))) return 0;
    float t = f * vdot(edge_1, q);
    return (t > RAY_EPSILON) & (t < max_distance);
}

/* BottomLevelBVH::trace, BottomLevelBVH.cpp:355-396 */
static void blas_trace(const orc_scene * sc, const orc_blas * b, const ray_t * ray, hit_t * h, const float * world, counters_t * c) {
    int stack[RTX_MAX_STACK];
    int stack_size = 1;
    stack[0] = 0;
    v3 inv_dir = V3(1.0f / ray->direction.x, 1.0f / ray->direction.y, 1.0f / ray->direction.z);
    int steps = 0;                                          /* :362,390: nodes that passed the slab test */
    while (stack_size > 0) {
        const rtx_bvh_node * node = &b->nodes[stack[--stack_size]];
        c->work.blas_nodes_closest++;
        if (!aabb_intersect(node, ray->origin, inv_dir, h->distance)) continue;
        steps++;
        if (((uint32_t)node->count & ~AXIS_MASK) > 0) {
            for (int i = node->left_or_first; i < node->left_or_first + node->count; i++) {
                c->work.tri_tests_closest++;
                triangle_trace(b, i, ray, h, world);
            }
        } else {
            if (visit_left_first(node, ray->direction, sc->config.traversal_strategy)) {
                stack[stack_size++] = node->left_or_first + 1;
                stack[stack_size++] = node->left_or_first;
            } else {
                stack[stack_size++] = node->left_or_first;
                stack[stack_size++] = node->left_or_first + 1;
            }
        }
    }
    h->bvh_steps += steps;                                  /* :393-395 */
}

/* BottomLevelBVH::intersect, BottomLevelBVH.cpp:398-437 */
static int blas_intersect(const orc_scene * sc, const orc_blas * b, v3 origin, v3 direction, float max_distance, counters_t * c) {
    int stack[RTX_MAX_STACK];
    int stack_size = 1;
    stack[0] = 0;
    v3 inv_dir = V3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
    while (stack_size > 0) {
        const rtx_bvh_node * node = &b->nodes[stack[--stack_size]];
        c->work.blas_nodes_any++;
        if (!aabb_intersect(node, origin, inv_dir, max_distance)) continue;
        if (((uint32_t)node->count & ~AXIS_MASK) > 0) {
            for (int i = node->left_or_first; i < node->left_or_first + node->count; i++) {
                c->work.tri_tests_any++;
                if (triangle_intersect(b, i, origin, direction, max_distance)) return 1;
            }
        } else {
            if (visit_left_first(node, direction, sc->config.traversal_strategy)) {
                stack[stack_size++] = node->left_or_first + 1;
                stack[stack_size++] = node->left_or_first;
            } else {
                stack[stack_size++] = node->left_or_first;
                stack[stack_size++] = node->left_or_first + 1;
            }
        }
    }
    return 0;
}

/* Mesh::trace, Mesh.cpp:17-31 */
static void mesh_trace(const orc_scene * sc, const rtx_instance * m, const ray_t * ray, hit_t * h, counters_t * c) {
    ray_t r;
    r.origin    = transform_position (m->world_inv, ray->origin);
    r.direction = transform_direction(m->world_inv, ray->direction);
    r.dO_dx = transform_direction(m->world_inv, ray->dO_dx);
    r.dO_dy = transform_direction(m->world_inv, ray->dO_dy);
    r.dD_dx = transform_direction(m->world_inv, ray->dD_dx);
    r.dD_dy = transform_direction(m->world_inv, ray->dD_dy);
    c->work.instances_closest++;
    blas_trace(sc, &sc->blas[m->blas_id], &r, h, m->world, c);
}

/* Mesh::intersect, Mesh.cpp:33-40 */
static int mesh_intersect(const orc_scene * sc, const rtx_instance * m, v3 origin, v3 direction, float max_distance, counters_t * c) {
    v3 o = transform_position (m->world_inv, origin);
    v3 d = transform_direction(m->world_inv, direction);
    c->work.instances_any++;
    return blas_intersect(sc, &sc->blas[m->blas_id], o, d, max_distance, c);
}

/* TopLevelBVH::trace, TopLevelBVH.cpp:53-87 */
static void tlas_trace(const orc_scene * sc, const ray_t * ray, hit_t * h, counters_t * c) {
    const rtx_frame * fr = &sc->frame;
    if (fr->tlas_node_count <= 0) return;
    int stack[RTX_MAX_STACK];
    int stack_size = 1;
    stack[0] = 0;
    v3 inv_dir = V3(1.0f / ray->direction.x, 1.0f / ray->direction.y, 1.0f / ray->direction.z);
    while (stack_size > 0) {
        const rtx_bvh_node * node = &fr->tlas_nodes[stack[--stack_size]];
        c->work.tlas_nodes_closest++;
        if (!aabb_intersect(node, ray->origin, inv_dir, h->distance)) continue;
        if (((uint32_t)node->count & ~AXIS_MASK) > 0) {
            for (int i = node->left_or_first; i < node->left_or_first + node->count; i++)
                mesh_trace(sc, &fr->instances[fr->tlas_indices[i]], ray, h, c);
        } else {
            if (visit_left_first(node, ray->direction, sc->config.traversal_strategy)) {
                stack[stack_size++] = node->left_or_first + 1;
                stack[stack_size++] = node->left_or_first;
            } else {
                stack[stack_size++] = node->left_or_first;
                stack[stack_size++] = node->left_or_first + 1;
            }
        }
    }
}

/* TopLevelBVH::intersect, TopLevelBVH.cpp:89-129 */
static int tlas_intersect(const orc_scene * sc, v3 origin, v3 direction, float max_distance, counters_t * c) {
    const rtx_frame * fr = &sc->frame;
    if (fr->tlas_node_count <= 0) return 0;
    int stack[RTX_MAX_STACK];
    int stack_size = 1;
    stack[0] = 0;
    v3 inv_dir = V3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
    while (stack_size > 0) {
        const rtx_bvh_node * node = &fr->tlas_nodes[stack[--stack_size]];
        c->work.tlas_nodes_any++;
        if (!aabb_intersect(node, origin, inv_dir, max_distance)) continue;
        if (((uint32_t)node->count & ~AXIS_MASK) > 0) {
            for (int i = node->left_or_first; i < node->left_or_first + node->count; i++)
                if (mesh_intersect(sc, &fr->instances[fr->tlas_indices[i]], origin, direction, max_distance, c)) return 1;
        } else {
            if (visit_left_first(node, direction, sc->config.traversal_strategy)) {
                stack[stack_size++] = node->left_or_first + 1;
                stack[stack_size++] = node->left_or_first;
            } else {
                stack[stack_size++] = node->left_or_first;
                stack[stack_size++] = node->left_or_first + 1;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Sphere::trace, Sphere.cpp:9-90
 * ---------------------------------------------------------------------------------------------- */
static void sphere_trace(const rtx_sphere * sp, const ray_t * ray, hit_t * h) {
    v3 center = v3p(sp->center);
    float r2 = sp->radius_squared;
    v3 oc = vsub(ray->origin, center);
    float a = vdot(ray->direction, ray->direction);
    float b = 2.0f * vdot(oc, ray->direction);
    float c = vdot(oc, oc) - r2;
    float d = b * b - 4.0f * a * c;
    if (!(d >= 0.0f)) return;
    float sqrt_d    = sqrtf(d);
    float inv_denom = -(1.0f / (2.0f * a));
    float t0 = (b + sqrt_d) * inv_denom;
    float t1 = (b - sqrt_d) * inv_denom;
    float t = (t0 > RAY_EPSILON) ? t0 : t1;
    if (!((t > RAY_EPSILON) & (t < h->distance))) return;

    const float one_over_r = sp->radius_inv;
    h->hit = 1;
    h->distance = t;
    h->point  = vadd(ray->origin, vmuls(ray->direction, t));
    h->normal = vmuls(vsub(h->point, center), one_over_r);
    h->material_id = sp->material_id;
    h->u = atan2f(h->normal.z, h->normal.x) * ONE_OVER_TWO_PI + 0.5f;
    h->v = acosf(h->normal.y) * ONE_OVER_PI + 0.5f;

    /* Igehy transfer, Sphere.cpp:63-89 */
    v3 dPx = vmadd_s(ray->dD_dx, t, ray->dO_dx);
    v3 dPy = vmadd_s(ray->dD_dy, t, ray->dO_dy);
    float denom = -1.0f / vdot(ray->direction, h->normal);
    float dt_dx = vdot(dPx, h->normal) * denom;
    float dt_dy = vdot(dPy, h->normal) * denom;
    v3 dP_dx = vmadd_s(ray->direction, dt_dx, dPx);
    v3 dP_dy = vmadd_s(ray->direction, dt_dy, dPy);
    h->dO_dx = dP_dx;
    h->dO_dy = dP_dy;
    h->dN_dx = vmuls(dP_dx, one_over_r);
    h->dN_dy = vmuls(dP_dy, one_over_r);

    float ds_denom = ONE_OVER_TWO_PI / (h->normal.x * h->normal.x + h->normal.z * h->normal.z + 1e-8f);
    h->ds_dx = (h->normal.x * h->dN_dx.z - h->normal.z * h->dN_dx.x) * ds_denom;
    h->ds_dy = (h->normal.x * h->dN_dy.z - h->normal.z * h->dN_dy.x) * ds_denom;
    float dt_denom = -ONE_OVER_PI * (1.0f / sqrtf(1.0f - h->normal.y * h->normal.y + 1e-8f));
    h->dt_dx = h->dN_dx.y * dt_denom;
    h->dt_dy = h->dN_dy.y * dt_denom;
}

/* Sphere::intersect, Sphere.cpp:92-112 */
static int sphere_intersect(const rtx_sphere * sp, v3 origin, v3 direction, float max_distance) {
    v3 c = vsub(v3p(sp->center), origin);
    float t = vdot(c, direction);
    v3 Q = vsub(c, vmuls(direction, t));
    float p2 = vdot(Q, Q);
    float rs = sp->radius_squared;
    if (!(p2 < rs)) return 0;
    t = t - sqrtf(rs - p2);
    return (t > RAY_EPSILON) & (t < max_distance);
}

/* Plane::trace, Plane.cpp:13-68 */
static void plane_trace(const rtx_plane * pl, const ray_t * ray, hit_t * h) {
    v3 normal = v3p(pl->normal);
    float t = -(vdot(normal, ray->origin) + pl->distance) / vdot(normal, ray->direction);
    if (!((t > RAY_EPSILON) & (t < h->distance))) return;
    h->hit = 1;
    h->distance = t;
    h->point  = vadd(ray->origin, vmuls(ray->direction, t));
    h->normal = normal;
    h->material_id = pl->material_id;
    v3 ua = v3p(pl->u_axis), va = v3p(pl->v_axis);
    h->u = vdot(h->point, ua);
    h->v = vdot(h->point, va);

    v3 dPx = vmadd_s(ray->dD_dx, t, ray->dO_dx);
    v3 dPy = vmadd_s(ray->dD_dy, t, ray->dO_dy);
    float denom = -1.0f / (vdot(ray->direction, h->normal) + 1e-8f);
    float dt_dx = vdot(dPx, h->normal) * denom;
    float dt_dy = vdot(dPy, h->normal) * denom;
    v3 dP_dx = vmadd_s(ray->direction, dt_dx, dPx);
    v3 dP_dy = vmadd_s(ray->direction, dt_dy, dPy);
    h->dO_dx = dP_dx;
    h->dO_dy = dP_dy;
    h->dN_dx = V3(0.0f, 0.0f, 0.0f);
    h->dN_dy = V3(0.0f, 0.0f, 0.0f);
    h->ds_dx = vdot(dP_dx, ua);
    h->ds_dy = vdot(dP_dy, ua);
    h->dt_dx = vdot(dP_dx, va);
    h->dt_dy = vdot(dP_dy, va);
}

/* Plane::intersect, Plane.cpp:70-79 */
static int plane_intersect(const rtx_plane * pl, v3 origin, v3 direction, float max_distance) {
    v3 normal = v3p(pl->normal);
    float t = -(vdot(normal, origin) + pl->distance) / vdot(normal, direction);
    return (t > RAY_EPSILON) & (t < max_distance);
}

/* Scene::trace_primitives, Scene.cpp:173-177 (PrimitiveList::trace, PrimitiveList.h:29-33) */
static void scene_trace(const orc_scene * sc, const ray_t * ray, hit_t * h, counters_t * c) {
    const rtx_frame * fr = &sc->frame;
    c->work.closest_rays++;
    for (int i = 0; i < fr->sphere_count; i++) sphere_trace(&fr->spheres[i], ray, h);
    for (int i = 0; i < fr->plane_count;  i++) plane_trace (&fr->planes[i],  ray, h);
    tlas_trace(sc, ray, h, c);
}

/* Scene::intersect_primitives, Scene.cpp:179-190 (PrimitiveList::intersect, PrimitiveList.h:35-45) */
static int scene_intersect(const orc_scene * sc, v3 origin, v3 direction, float max_distance, counters_t * c) {
    const rtx_frame * fr = &sc->frame;
    c->work.any_rays++;
    for (int i = 0; i < fr->sphere_count; i++) if (sphere_intersect(&fr->spheres[i], origin, direction, max_distance)) return 1;
    for (int i = 0; i < fr->plane_count;  i++) if (plane_intersect (&fr->planes[i],  origin, direction, max_distance)) return 1;
    return tlas_intersect(sc, origin, direction, max_distance, c);
}

/* ------------------------------------------------------------------------------------------------
 * Texture, Texture.h:33-62, Texture.cpp:131-337
 * ---------------------------------------------------------------------------------------------- */
static float ewa_table[RTX_EWA_LUT_SIZE];
static pthread_once_t ewa_once = PTHREAD_ONCE_INIT;
static void ewa_init(void) {   /* Texture::init(alpha = 2.0f), Texture.h:53-62 */
    const float alpha = 2.0f;
    float denom = 1.0f / (float)(RTX_EWA_LUT_SIZE - 1);
    float exp_neg_alpha = expf(-alpha);
    for (int i = 0; i < RTX_EWA_LUT_SIZE; i++) {
        float r2 = (float)i * denom;
        ewa_table[i] = expf(-alpha * r2) - exp_neg_alpha;
    }
}
void orc_ewa_table(float out[RTX_EWA_LUT_SIZE]) { pthread_once(&ewa_once, ewa_init); memcpy(out, ewa_table, sizeof(ewa_table)); }

static inline int mod_pos(int value, int modulus) {   /* Math::mod, Math.h:54-61 */
    int r = value % modulus;
    if (r < 0) r += modulus;
    return r;
}

typedef struct { const orc_texture * t; counters_t * c; } texctx_t;

static v3 fetch_texel(const texctx_t * tc, int x, int y, int level) {   /* Texture.cpp:131-147 */
    const rtx_texture_desc * d = &tc->t->desc;
    int offset = d->mip_offsets[level];
    int lw = d->width >> level, lh = d->height >> level;
    x = mod_pos(x, lw);
    y = mod_pos(y, lh);
    if (tc->c) tc->c->work.texel_fetches++;
    return v3p(&tc->t->texels[3 * (size_t)(offset + x + y * lw)]);
}

static v3 sample_nearest(const texctx_t * tc, float s, float t) {       /* Texture.cpp:149-154 */
    int x = orc_float_to_int(s * (float)tc->t->desc.width);
    int y = orc_float_to_int(t * (float)tc->t->desc.height);
    return fetch_texel(tc, x, y, 0);
}

static v3 sample_bilinear(const texctx_t * tc, float s, float t, int level) {   /* Texture.cpp:156-186 */
    int lw = tc->t->desc.width >> level, lh = tc->t->desc.height >> level;
    s = s * (float)lw - 0.5f;
    t = t * (float)lh - 0.5f;
    float fs = s - floorf(s);
    float ft = t - floorf(t);
    float oms = 1.0f - fs, omt = 1.0f - ft;
    float w0 = oms * omt;
    float w1 = fs  * omt;
    float w2 = oms * ft;
    float w3 = 1.0f - w0 - w1 - w2;
    int is = orc_float_to_int(s - 0.5f);
    int it = orc_float_to_int(t - 0.5f);
    v3 r =       vmuls(fetch_texel(tc, is,     it,     level), w0);
    r = vadd(r,  vmuls(fetch_texel(tc, is + 1, it,     level), w1));
    r = vadd(r,  vmuls(fetch_texel(tc, is,     it + 1, level), w2));
    r = vadd(r,  vmuls(fetch_texel(tc, is + 1, it + 1, level), w3));
    return r;
}

/* std::max / std::min semantics: max(a,b) = (a<b)?b:a ; min(a,b) = (b<a)?b:a */
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
static inline float std_min(float a, float b) { return (b < a) ? b : a; }

static v3 sample_trilinear(const texctx_t * tc, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {  /* Texture.cpp:189-204 */
    const rtx_texture_desc * d = &tc->t->desc;
    float width = 2.0f * std_max(std_max(fabsf(ds_dx), fabsf(ds_dy)), std_max(fabsf(dt_dx), fabsf(dt_dy)));
    float lambda = (float)d->mip_levels - 1.0f + log2f(std_max(width, 1e-8f));
    int level = orc_float_to_int(lambda - 0.5f);
    if (level < 0) return sample_bilinear(tc, s, t, 0);
    if (level >= d->mip_levels - 1) return fetch_texel(tc, 0, 0, d->mip_levels - 1);
    float f = lambda - floorf(lambda);
    return vadd(vmuls(sample_bilinear(tc, s, t, level), 1.0f - f), vmuls(sample_bilinear(tc, s, t, level + 1), f));
}

static v3 sample_anisotropic(const texctx_t * tc, float max_aniso, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {  /* Texture.cpp:207-239 */
    const rtx_texture_desc * d = &tc->t->desc;
    float p_x = std_max(fabsf(ds_dx), fabsf(dt_dx));
    float p_y = std_max(fabsf(ds_dy), fabsf(dt_dy));
    float p_min = std_min(p_x, p_y);
    float p_max = std_max(p_x, p_y);
    float N = std_min(ceilf(p_max / p_min), max_aniso);
    float one_over_N = 1.0f / N;
    float lambda = (float)d->mip_levels - 1.0f + log2f(p_max * one_over_N);
    int level = orc_float_to_int(lambda);
    if (level < 0) return sample_bilinear(tc, s, t, 0);
    if (level >= d->mip_levels - 1) return fetch_texel(tc, 0, 0, d->mip_levels - 1);
    int x_major = p_x > p_y;
    float step_s = x_major ? ds_dx : ds_dy;
    float step_t = x_major ? dt_dx : dt_dy;
    float one_over_N_plus_1 = 1.0f / (N + 1.0f);
    v3 sum = V3(0.0f, 0.0f, 0.0f);
    for (float i = 1.0f; i <= N + 0.001f; i += 1.0f) {
        float x = s + step_s * (i * one_over_N_plus_1 - 0.5f);
        float y = t + step_t * (i * one_over_N_plus_1 - 0.5f);
        sum = vadd(sum, sample_bilinear(tc, x, y, level));
    }
    return vmuls(sum, one_over_N);
}

static v3 sample_ewa(const texctx_t * tc, float max_aniso, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {  /* Texture.cpp:242-337 */
    const rtx_texture_desc * d = &tc->t->desc;
    float major_x = ds_dx, major_y = dt_dx;
    float minor_x = ds_dy, minor_y = dt_dy;
    float major_length = sqrtf(major_x * major_x + major_y * major_y);   /* Vector2::length, Vector2.h:14-16,28-30 */
    float minor_length = sqrtf(minor_x * minor_x + minor_y * minor_y);
    if (minor_length > major_length) {
        float tx = minor_x, ty = minor_y, tl = minor_length;
        minor_x = major_x; minor_y = major_y; minor_length = major_length;
        major_x = tx; major_y = ty; major_length = tl;
    }
    if (minor_length < 0.00001f) return sample_bilinear(tc, s, t, 0);
    if (major_length > (float)d->width) return fetch_texel(tc, 0, 0, d->mip_levels - 1);
    if (minor_length * max_aniso < major_length) {
        float scale = major_length / (minor_length * max_aniso);
        minor_x *= scale; minor_y *= scale;
        minor_length *= scale;
    }
    float lambda = std_max(0.0f, (float)d->mip_levels - 1.0f + log2f(minor_length));
    int level = orc_float_to_int(lambda);
    if (level >= d->mip_levels - 1) return fetch_texel(tc, 0, 0, d->mip_levels - 1);
    float level_width  = (float)(d->width  >> level);
    float level_height = (float)(d->height >> level);
    s = s * level_width  - 0.5f;
    t = t * level_height - 0.5f;
    float majx = major_x * level_width, majy = major_y * level_height;
    float minx = minor_x * level_width, miny = minor_y * level_height;
    float a =  1.0f + (majy * majy + miny * miny);
    float b = -2.0f * (majx * majy + minx * miny);
    float c =  1.0f + (majx * majx + minx * minx);
    float one_over_f = 1.0f / (a * c - b * b * 0.25f);
    a *= one_over_f; b *= one_over_f; c *= one_over_f;
    float det = -b * b + 4.0f * a * c;
    float sqrt_u = sqrtf(det * c);
    float sqrt_v = sqrtf(det * a);
    float two_inv_det = 2.0f / det;
    float tu = two_inv_det * sqrt_u;
    float tv = two_inv_det * sqrt_v;
    int s0 = orc_float_to_int(s - tu + 0.5f);
    int s1 = orc_float_to_int(s + tu - 0.5f);
    int t0 = orc_float_to_int(t - tv + 0.5f);
    int t1 = orc_float_to_int(t + tv - 0.5f);
    v3 sum = V3(0.0f, 0.0f, 0.0f);
    float sum_weights = 0.0f;
    float t0f = (float)t0, s0f = (float)s0;
    float tf = t0f;
    pthread_once(&ewa_once, ewa_init);
    for (int ti = t0; ti <= t1; ti++, tf += 1.0f) {
        float tt = tf - t;
        float sf = s0f;
        for (int si = s0; si <= s1; si++, sf += 1.0f) {
            float ss = sf - s;
            float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
            if (r2 < 1.0f) {
                int idx = orc_float_to_int(r2 * (float)RTX_EWA_LUT_SIZE);
                if (RTX_EWA_LUT_SIZE - 1 < idx) idx = RTX_EWA_LUT_SIZE - 1;   /* std::min(idx, size-1) */
                float weight = ewa_table[idx];
                sum = vadd(sum, vmuls(fetch_texel(tc, si, ti, level), weight));
                sum_weights += weight;
            }
        }
    }
    return vdivs(sum, sum_weights);   /* Vector3 / float = * (1/f), Vector3.h:82 */
}

/* Texture::sample, Texture.h:33-49 */
static v3 texture_sample(const texctx_t * tc, const rtx_config * cf, float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy) {
    if (cf->texture_mode == RTX_TEXTURE_NEAREST)  return sample_nearest(tc, s, t);
    if (cf->texture_mode == RTX_TEXTURE_BILINEAR) return sample_bilinear(tc, s, t, 0);
    if (!tc->t->desc.mipmapped) return sample_bilinear(tc, s, t, 0);
    if (cf->mip_filter == RTX_MIP_TRILINEAR)   return sample_trilinear(tc, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
    if (cf->mip_filter == RTX_MIP_ANISOTROPIC) return sample_anisotropic(tc, cf->max_anisotropy, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
    return sample_ewa(tc, cf->max_anisotropy, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
}

void orc_texture_sample(const orc_texture * tex, int32_t texture_mode, int32_t mip_filter, float max_anisotropy,
                        float s, float t, float ds_dx, float ds_dy, float dt_dx, float dt_dy, float out_rgb[3]) {
    rtx_config cf; memset(&cf, 0, sizeof(cf));
    cf.texture_mode = texture_mode; cf.mip_filter = mip_filter; cf.max_anisotropy = max_anisotropy;
    texctx_t tc = { tex, NULL };
    v3 r = texture_sample(&tc, &cf, s, t, ds_dx, ds_dy, dt_dx, dt_dy);
    out_rgb[0] = r.x; out_rgb[1] = r.y; out_rgb[2] = r.z;
}

/* Sky::sample, Sky.cpp:28-68 (lane 1) */
static v3 sky_sample(const float * sky, int size, v3 direction) {
    float r = 0.5f * ONE_OVER_PI * acosf(direction.z) * (1.0f / sqrtf(direction.x * direction.x + direction.y * direction.y));
    float u = direction.x * r + 0.5f;
    float v = direction.y * r + 0.5f;
    int x = float_to_int_trunc(u * (float)size);
    int y = float_to_int_trunc(v * (float)size);
    int index = (int)((uint32_t)y * (uint32_t)size + (uint32_t)x);
    index = index > 0 ? index : 0;
    index = index < size * size ? index : size * size;   /* inclusive upper clamp: one past the end, Sky.cpp:45 */
    return vmuls(v3p(&sky[3 * (size_t)index]), ONE_OVER_PI);
}
void orc_sky_sample(const float * sky, int32_t size, const float direction[3], float out_rgb[3]) {
    v3 r = sky_sample(sky, size, v3p(direction));
    out_rgb[0] = r.x; out_rgb[1] = r.y; out_rgb[2] = r.z;
}

/* ------------------------------------------------------------------------------------------------
 * Lights: Light.h:12-26, PointLight.h:9-11, SpotLight.h:17-33, DirectionalLight.h:9-11
 * ---------------------------------------------------------------------------------------------- */
static inline float pow2_128(float v) {  /* Math::pow2<128>, Math.h:92-96 */
    v = v * v; v = v * v; v = v * v; v = v * v; v = v * v; v = v * v; v = v * v;
    return v;
}
static v3 light_calc(v3 colour, v3 normal, v3 to_light, v3 to_camera) {
    float intensity = vdot(normal, to_light);
    if (!(intensity > 0.0f)) return V3(0.0f, 0.0f, 0.0f);
    v3 half_angle = vnormalize(vadd(to_light, to_camera));
    float specular_factor = vdot(normal, half_angle);
    intensity = intensity + pow2_128(specular_factor);
    return vmuls(colour, intensity);
}
static v3 point_light_calc(v3 colour, v3 normal, v3 to_light, v3 to_camera, float distance_squared) {
    return vdivs(light_calc(colour, normal, to_light, to_camera), distance_squared);
}
static v3 spot_light_calc(const rtx_spot_light * l, v3 normal, v3 to_light, v3 to_camera, float distance_squared) {
    float d = vdot(to_light, v3p(l->negative_direction));
    float inner = l->inner_cutoff, outer = l->outer_cutoff;
    if (!(d > outer)) return V3(0.0f, 0.0f, 0.0f);
    float falloff = (d - outer) / (inner - outer);
    falloff = (falloff > 1.0f) ? 1.0f : falloff;
    return vmuls(point_light_calc(v3p(l->colour), normal, to_light, to_camera, distance_squared), falloff);
}

/* ------------------------------------------------------------------------------------------------
 * Raytracer::bounce, Raytracer.cpp:87-400 (lane 1)
 * ---------------------------------------------------------------------------------------------- */
static v3 bounce(const orc_scene * sc, const ray_t * ray, int bounces_left, float * distance, counters_t * c) {
    const rtx_frame * fr = &sc->frame;
    v3 result = V3(0.0f, 0.0f, 0.0f);

    hit_t h;
    memset(&h, 0, sizeof(h));
    h.hit = 0;
    h.distance = INFINITY;
    scene_trace(sc, ray, &h, c);

    if (sc->config.heatmap) {                               /* BVH_VISUALIZE_HEATMAP, :97-102 (the out distance stays unset there; INFINITY here) */
        *distance = INFINITY;
        return V3(h.bvh_steps * (1.0f / 32.0f), h.bvh_steps * (1.0f / 256.0f), h.bvh_steps * (1.0f / 512.0f));
    }

    if (!h.hit) {                                           /* :105-111 */
        c->work.sky_lookups++;
        *distance = INFINITY;
        return sky_sample(sc->sky, sc->sky_size, ray->direction);
    }
    *distance = h.distance;                                 /* :113 */

    const rtx_material * mat = &sc->materials[h.material_id];
    c->work.shaded_hits++;
    v3 albedo = v3p(mat->diffuse);                          /* Material::get_albedo, Material.h:16-22 */
    if (mat->texture_id >= 0) {
        texctx_t tc = { &sc->textures[mat->texture_id], c };
        int diff = sc->config.texture_mode == RTX_TEXTURE_MIPMAP;   /* RAY_DIFFERENTIALS_ENABLED, Config.h:46 */
        v3 tex = texture_sample(&tc, &sc->config, h.u, h.v, diff ? h.ds_dx : 0.0f, diff ? h.ds_dy : 0.0f, diff ? h.dt_dx : 0.0f, diff ? h.dt_dy : 0.0f);
        albedo = vmul(albedo, tex);
    }

    if (vdot(albedo, albedo) > 0.0f) {                      /* :143-201 */
        v3 diffuse = v3p(fr->ambient);
        v3 to_camera = vnormalize(vsub(v3p(fr->camera.position), h.point));

        for (int i = 0; i < fr->point_light_count; i++) {
            const rtx_point_light * l = &fr->point_lights[i];
            v3 to_light = vsub(v3p(l->position), h.point);
            float d2 = vdot(to_light, to_light);
            float d  = sqrtf(d2);
            to_light = vdivs(to_light, d);
            c->stats.num_shadow_rays++;
            if (scene_intersect(sc, h.point, to_light, d, c)) continue;
            diffuse = vadd(diffuse, point_light_calc(v3p(l->colour), h.normal, to_light, to_camera, d2));
        }
        for (int i = 0; i < fr->spot_light_count; i++) {
            const rtx_spot_light * l = &fr->spot_lights[i];
            v3 to_light = vsub(v3p(l->position), h.point);
            float d2 = vdot(to_light, to_light);
            float d  = sqrtf(d2);
            to_light = vdivs(to_light, d);
            c->stats.num_shadow_rays++;
            if (scene_intersect(sc, h.point, to_light, d, c)) continue;
            diffuse = vadd(diffuse, spot_light_calc(l, h.normal, to_light, to_camera, d2));
        }
        for (int i = 0; i < fr->directional_light_count; i++) {
            const rtx_directional_light * l = &fr->directional_lights[i];
            v3 nd = v3p(l->negative_direction);
            c->stats.num_shadow_rays++;
            if (scene_intersect(sc, h.point, nd, INFINITY, c)) continue;
            diffuse = vadd(diffuse, light_calc(v3p(l->colour), h.normal, nd, to_camera));
        }
        result = vmadd_v(diffuse, albedo, result);
    }

    if (bounces_left > 0) {                                 /* :204-397 */
        v3 colour_reflection = V3(0.0f, 0.0f, 0.0f);
        v3 Ks = v3p(mat->reflection);
        v3 Kt = v3p(mat->transmittance);
        int reflection_mask = vdot(Ks, Ks) > 0.0f;
        int refraction_mask = vdot(Kt, Kt) > 0.0f;

        if (reflection_mask) {
            ray_t rr;
            rr.origin = h.point;
            float dn = vdot(ray->direction, h.normal);
            rr.direction = vsub(ray->direction, vmuls(h.normal, 2.0f * dn));   /* Math::reflect, Math.h:28-30 */
            rr.dO_dx = h.dO_dx;
            rr.dO_dy = h.dO_dy;
            float dDN_dx = vdot(ray->dD_dx, h.normal) + vdot(ray->direction, h.dN_dx);
            float dDN_dy = vdot(ray->dD_dy, h.normal) + vdot(ray->direction, h.dN_dy);
            rr.dD_dx = vsub(ray->dD_dx, vmuls(vadd(vmuls(h.dN_dx, vdot(ray->direction, h.normal)), vmuls(h.normal, dDN_dx)), 2.0f));
            rr.dD_dy = vsub(ray->dD_dy, vmuls(vadd(vmuls(h.dN_dy, vdot(ray->direction, h.normal)), vmuls(h.normal, dDN_dy)), 2.0f));
            c->stats.num_reflection_rays++;
            c->work.rays_spawned++;
            float rd;
            colour_reflection = vmul(Ks, bounce(sc, &rr, bounces_left - 1, &rd, c));
            result = vadd(result, colour_reflection);
        }

        if (refraction_mask) {
            float dot = vdot(ray->direction, h.normal);
            int entering = dot < 0.0f;
            const float air = 1.0f;                          /* Material::air_index_of_refraction, Material.h:24 */
            float ior = mat->index_of_refraction;
            float n_1 = entering ? air : ior;
            float n_2 = entering ? ior : air;
            float cos_theta = entering ? (0.0f - dot) : dot;
            v3 normal = entering ? h.normal : vneg(h.normal);
            float eta = n_1 / n_2;
            float k = 1.0f - (eta * eta * (1.0f - (cos_theta * cos_theta)));
            if (k < 0.0f) {                                  /* TIR, :310-314 */
                return reflection_mask ? vadd(result, colour_reflection) : result;
            }
            ray_t rf;
            rf.origin = h.point;
            /* Math::refract, Math.h:34-36 */
            rf.direction = vadd(vmuls(ray->direction, eta), vmuls(normal, (eta * cos_theta) - sqrtf(k)));
            c->stats.num_refraction_rays++;
            c->work.rays_spawned++;
            rf.dO_dx = h.dO_dx;
            rf.dO_dy = h.dO_dy;
            float dDN_dx = vdot(ray->dD_dx, h.normal) + vdot(ray->direction, h.dN_dx);
            float dDN_dy = vdot(ray->dD_dy, h.normal) + vdot(ray->direction, h.dN_dy);
            float D_dot_N = -cos_theta;
            float Dprime_dot_N = -sqrtf(k);
            float mu = -(eta * cos_theta + Dprime_dot_N);
            /* :341-342 — scalar + component-wise vector product, literally */
            rf.dD_dx = vsub(vmuls(ray->dD_dx, eta), vmuls(vadds(vmul(h.dN_dx, h.normal), mu * D_dot_N), dDN_dx));
            rf.dD_dy = vsub(vmuls(ray->dD_dy, eta), vmuls(vadds(vmul(h.dN_dy, h.normal), mu * D_dot_N), dDN_dy));

            float refraction_distance;
            v3 colour_refraction = bounce(sc, &rf, bounces_left - 1, &refraction_distance, c);

            /* Beer's law, :348-376 */
            v3 absorption = vsub(Kt, V3(1.0f, 1.0f, 1.0f));
            float beer_x = expf(absorption.x * refraction_distance);
            float beer_y = expf(absorption.y * refraction_distance);
            float beer_z = expf(absorption.z * refraction_distance);
            if (entering) {
                colour_refraction.x = colour_refraction.x * beer_x;
                colour_refraction.y = colour_refraction.y * beer_y;
                colour_refraction.z = colour_refraction.z * beer_z;
            }
            /* Schlick, :378-391 */
            float r_0 = (n_1 - n_2) / (n_1 + n_2);
            r_0 = r_0 * r_0;
            if (n_1 > n_2) cos_theta = 0.0f - vdot(rf.direction, normal);
            float omc  = 1.0f - cos_theta;
            float omc2 = omc * omc;
            float F_r = r_0 + ((1.0f - r_0) * omc2) * (omc2 * omc);
            float F_t = 1.0f - F_r;
            v3 blend = vadd(vmuls(colour_reflection, F_r), vmuls(colour_refraction, F_t));
            return vadd(result, blend);
        }
    }
    return result;
}

/* Window::plot, Window.h:56-65 */
static inline float clampf(float v, float lo, float hi) { if (v < lo) return lo; if (v > hi) return hi; return v; }   /* Math::clamp, Math.h:12-18 */
uint32_t orc_plot(const float rgb[3]) {
    int r = orc_float_to_int(clampf(rgb[0] * 255.0f, 0.0f, 255.0f) - 0.5f);
    int g = orc_float_to_int(clampf(rgb[1] * 255.0f, 0.0f, 255.0f) - 0.5f);
    int b = orc_float_to_int(clampf(rgb[2] * 255.0f, 0.0f, 255.0f) - 0.5f);
    return (uint32_t)((r << 16) | (g << 8) | b);
}

/* Raytracer::render_tile, Raytracer.cpp:3-85 (lane 1) */
static void render_tile(const orc_scene * sc, int tile_x, int tile_y, int tile_w, int tile_h,
                        float * rgb, uint32_t * packed, float * dist, counters_t * c) {
    const rtx_camera * cam = &sc->frame.camera;
    const int W = sc->config.width;
    ray_t ray;
    ray.origin = v3p(cam->position);
    ray.dO_dx = V3(0.0f, 0.0f, 0.0f);
    ray.dO_dy = V3(0.0f, 0.0f, 0.0f);
    v3 ax = v3p(cam->rotated_x_axis), ay = v3p(cam->rotated_y_axis), tl = v3p(cam->rotated_top_left_corner);
    for (int j = tile_y; j < tile_y + tile_h; j++) {
        for (int i = tile_x; i < tile_x + tile_w; i++) {
            float is = (float)i, js = (float)j;
            v3 d = vmadd_s(ax, is, vmadd_s(ay, js, tl));
            float dd  = vdot(d, d);
            float inv = 1.0f / sqrtf(dd);
            float denom = inv / dd;
            ray.dD_dx = vmuls(vsub(vmuls(ax, dd), vmuls(d, vdot(d, ax))), denom);
            ray.dD_dy = vmuls(vsub(vmuls(ay, dd), vmuls(d, vdot(d, ay))), denom);
            ray.direction = vmuls(d, inv);
            c->stats.num_primary_rays++;
            float distance;
            v3 colour = bounce(sc, &ray, sc->config.bounces, &distance, c);
            size_t p = (size_t)j * W + i;
            if (rgb)  { rgb[3 * p] = colour.x; rgb[3 * p + 1] = colour.y; rgb[3 * p + 2] = colour.z; }
            if (dist) dist[p] = distance;
            if (packed) { float cc[3] = { colour.x, colour.y, colour.z }; packed[p] = orc_plot(cc); }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Tile scheduler: WorkerThread.cpp:53-65 (atomic counter, tiles handed out from the back)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_scene * sc;
    int first_tile, tile_stride, tile_count;
    float * rgb; uint32_t * packed; float * dist;
    volatile int remaining;
    counters_t * per_thread;
} job_t;

typedef struct { job_t * job; int tid; } worker_arg_t;

static void * worker(void * p) {
    worker_arg_t * wa = (worker_arg_t *)p;
    job_t * job = wa->job;
    const orc_scene * sc = job->sc;
    const int tcx = (sc->config.width + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
    counters_t * c = &job->per_thread[wa->tid];
    while (job->remaining > 0) {
        int task = __sync_sub_and_fetch(&job->remaining, 1);
        if (task >= 0) {
            int tile = job->first_tile + task * job->tile_stride;
            int x = (tile % tcx) * RTX_TILE_SIZE;
            int y = (tile / tcx) * RTX_TILE_SIZE;
            int tw = x + RTX_TILE_SIZE < sc->config.width  ? RTX_TILE_SIZE : sc->config.width  - x;
            int th = y + RTX_TILE_SIZE < sc->config.height ? RTX_TILE_SIZE : sc->config.height - y;
            render_tile(sc, x, y, tw, th, job->rgb, job->packed, job->dist, c);
        }
    }
    return NULL;
}

int orc_render_tiles(const orc_scene * scene, int32_t first_tile, int32_t tile_stride, int32_t tile_count,
                     float * rgb, uint32_t * packed, float * dist,
                     rtx_stats * stats, rtx_work_counters * work, int32_t threads) {
    if (!scene || tile_count < 0 || first_tile < 0 || tile_stride < 1) return RTX_ERR_INVALID_ARG;
    const int tcx = (scene->config.width  + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
    const int tcy = (scene->config.height + RTX_TILE_SIZE - 1) / RTX_TILE_SIZE;
    if (tile_count > 0 && first_tile + (int64_t)(tile_count - 1) * tile_stride >= (int64_t)tcx * tcy) return RTX_ERR_INVALID_ARG;
    if (scene->config.stack_size < 1 || scene->config.stack_size > RTX_MAX_STACK) return RTX_ERR_LIMIT;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_once(&ewa_once, ewa_init);

    job_t job;
    job.sc = scene; job.first_tile = first_tile; job.tile_stride = tile_stride; job.tile_count = tile_count;
    job.rgb = rgb; job.packed = packed; job.dist = dist;
    job.remaining = tile_count;
    job.per_thread = (counters_t *)calloc((size_t)threads, sizeof(counters_t) + 64);
    if (!job.per_thread) return RTX_ERR_OOM;

    pthread_t * th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    worker_arg_t * wa = (worker_arg_t *)malloc(sizeof(worker_arg_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) { wa[i].job = &job; wa[i].tid = i; }
    if (threads == 1) worker(&wa[0]);
    else {
        for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, &wa[i]);
        for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    }
    counters_t total; memset(&total, 0, sizeof(total));
    for (int i = 0; i < threads; i++) {
        const uint64_t * src = (const uint64_t *)&job.per_thread[i];
        uint64_t * dst = (uint64_t *)&total;
        for (size_t k = 0; k < sizeof(counters_t) / 8; k++) dst[k] += src[k];
    }
    total.work.triangle_hits = 0;   /* not tracked by the oracle (derived on the device side) */
    if (stats) *stats = total.stats;
    if (work)  *work  = total.work;
    free(job.per_thread); free(th); free(wa);
    return RTX_OK;
}

void orc_trace_closest(const orc_scene * scene, const float r[18], float out[27]) {
    ray_t ray;
    ray.origin = v3p(r); ray.direction = v3p(r + 3);
    ray.dO_dx = v3p(r + 6); ray.dO_dy = v3p(r + 9); ray.dD_dx = v3p(r + 12); ray.dD_dy = v3p(r + 15);
    hit_t h; memset(&h, 0, sizeof(h)); h.distance = INFINITY;
    counters_t c; memset(&c, 0, sizeof(c));
    scene_trace(scene, &ray, &h, &c);
    out[0] = (float)h.hit; out[1] = h.distance;
    out[2] = h.point.x; out[3] = h.point.y; out[4] = h.point.z;
    out[5] = h.normal.x; out[6] = h.normal.y; out[7] = h.normal.z;
    out[8] = (float)h.material_id; out[9] = h.u; out[10] = h.v;
    out[11] = h.ds_dx; out[12] = h.ds_dy; out[13] = h.dt_dx; out[14] = h.dt_dy;
    out[15] = h.dO_dx.x; out[16] = h.dO_dx.y; out[17] = h.dO_dx.z;
    out[18] = h.dO_dy.x; out[19] = h.dO_dy.y; out[20] = h.dO_dy.z;
    out[21] = h.dN_dx.x; out[22] = h.dN_dx.y; out[23] = h.dN_dx.z;
    out[24] = h.dN_dy.x; out[25] = h.dN_dy.y; out[26] = h.dN_dy.z;
}

/* PointLight / SpotLight / DirectionalLight::calc_lighting at one set of inputs (PointLight.h:9-11, SpotLight.h:17-33, DirectionalLight.h:9-11) */
void orc_calc_lighting(const rtx_point_light * pl, const rtx_spot_light * sl, const rtx_directional_light * dl,
                       const float normal[3], const float to_light[3], const float to_camera[3], float distance_squared, float out9[9]) {
    const v3 a = point_light_calc(v3p(pl->colour), v3p(normal), v3p(to_light), v3p(to_camera), distance_squared);
    const v3 b = spot_light_calc(sl, v3p(normal), v3p(to_light), v3p(to_camera), distance_squared);
    const v3 c = light_calc(v3p(dl->colour), v3p(normal), v3p(dl->negative_direction), v3p(to_camera));
    out9[0] = a.x; out9[1] = a.y; out9[2] = a.z; out9[3] = b.x; out9[4] = b.y; out9[5] = b.z; out9[6] = c.x; out9[7] = c.y; out9[8] = c.z;
}

int orc_trace_any(const orc_scene * scene, const float origin[3], const float direction[3], float max_distance) {
    counters_t c; memset(&c, 0, sizeof(c));
    return scene_intersect(scene, v3p(origin), v3p(direction), max_distance, &c);
}

/* ------------------------------------------------------------------------------------------------
 * Window::draw_quad -> vertex.glsl + fragment_fxaa.glsl / fragment_identity.glsl  (PARITY UNPINNED, see rt_oracle.h)
 *
 * Conventions where OpenGL leaves room (all stated so that the HIP kernel and this restatement agree bit for bit):
 *  - the fragment of window pixel (i, j from the top) receives in_uv = ((i + 0.5) / W, (j + 0.5) / H) (vertex.glsl:9-13 through a
 *    perspective-free interpolation; evaluated here as (float(i) + 0.5f) * (1.0f / W));
 *  - texture2D with GL_NEAREST / GL_REPEAT (Window.cpp:48-49 and the GL defaults): texel = floor(fract(u) * size);
 *  - RGBA8 texel channels are c / 255; pow(c, 1 / 2.2) is looked up per 8-bit value (table supplied by the caller);
 *  - `vec2(-inv_screen_size.x, -inv_screen_size)` (fragment_fxaa.glsl:26-29) takes the FIRST component of the vec2: both offsets
 *    are 1 / SCREEN_WIDTH;
 *  - the RGB8 colour buffer stores round-to-nearest-even(clamp(c, 0, 1) * 255).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const uint32_t * px; int w, h; const float * lut; } screen_t;

static v3 get_screen(const screen_t * s, float u, float v) {                 /* fragment_fxaa.glsl:16-18 */
    float fu = u - floorf(u), fv = v - floorf(v);
    int x = (int)(fu * (float)s->w), y = (int)(fv * (float)s->h);
    if (x > s->w - 1) x = s->w - 1;
    if (y > s->h - 1) y = s->h - 1;
    uint32_t p = s->px[(size_t)y * s->w + x];
    return V3(s->lut[(p >> 16) & 255], s->lut[(p >> 8) & 255], s->lut[p & 255]);
}

static float luma_of(v3 c) { return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z; }   /* dot(luma, c), :22 */

static uint32_t to_unorm8(float c) {
    float x = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
    return (uint32_t)lrintf(x * 255.0f);
}

int orc_present(const uint32_t * packed, int32_t width, int32_t height, int32_t fxaa, const float * gamma_lut, uint32_t * display) {
    if (!packed || !gamma_lut || !display || width <= 0 || height <= 0) return RTX_ERR_INVALID_ARG;
    const screen_t s = { packed, width, height, gamma_lut };
    const float inv_x = 1.0f / (float)width, inv_y = 1.0f / (float)height;   /* Window.cpp:57 */
    const float REDUCE_MIN = 1.0f / 128.0f, REDUCE_MUL = 1.0f / 8.0f, SPAN_MAX = 8.0f;   /* :11-13 */
    for (int j = 0; j < height; j++) for (int i = 0; i < width; i++) {
        const float u = ((float)i + 0.5f) * inv_x, v = ((float)j + 0.5f) * inv_y;
        v3 out;
        if (!fxaa) out = get_screen(&s, u, v);                                /* fragment_identity.glsl:14-16 */
        else {                                                                /* fragment_fxaa.glsl:21-70 */
            const float luma_tl = luma_of(get_screen(&s, u + -inv_x, v + -inv_x));
            const float luma_tr = luma_of(get_screen(&s, u +  inv_x, v + -inv_x));
            const float luma_bl = luma_of(get_screen(&s, u + -inv_x, v +  inv_x));
            const float luma_br = luma_of(get_screen(&s, u +  inv_x, v +  inv_x));
            const float luma_m  = luma_of(get_screen(&s, u, v));
            const float luma_min = fminf(fminf(fminf(luma_tl, luma_tr), fminf(luma_bl, luma_br)), luma_m);
            const float luma_max = fmaxf(fmaxf(fmaxf(luma_tl, luma_tr), fmaxf(luma_bl, luma_br)), luma_m);
            float dir_x = (luma_bl + luma_br) - (luma_tl + luma_tr);
            float dir_y = (luma_tl + luma_bl) - (luma_tr + luma_br);
            const float reduce = fmaxf(REDUCE_MIN, (luma_tl + luma_tr + luma_bl + luma_br) * 0.25f * REDUCE_MUL);
            const float adjust = 1.0f / (fminf(fabsf(dir_x), fabsf(dir_y)) + reduce);
            dir_x = fminf(fmaxf(dir_x * adjust, -SPAN_MAX), SPAN_MAX) * inv_x;
            dir_y = fminf(fmaxf(dir_y * adjust, -SPAN_MAX), SPAN_MAX) * inv_y;
            const float k1 = 1.0f / 3.0f - 0.5f, k2 = 2.0f / 3.0f - 0.5f, k0 = 0.0f / 3.0f - 0.5f, k3 = 3.0f / 3.0f - 0.5f;
            const v3 a1 = get_screen(&s, u + dir_x * k1, v + dir_y * k1), a2 = get_screen(&s, u + dir_x * k2, v + dir_y * k2);
            const v3 b1 = get_screen(&s, u + dir_x * k0, v + dir_y * k0), b2 = get_screen(&s, u + dir_x * k3, v + dir_y * k3);
            const v3 result_a = V3(0.5f * (a1.x + a2.x), 0.5f * (a1.y + a2.y), 0.5f * (a1.z + a2.z));
            const v3 result_b = V3(0.5f * (b1.x + b2.x), 0.5f * (b1.y + b2.y), 0.5f * (b1.z + b2.z));
            const v3 result = V3(0.5f * (result_a.x + result_b.x), 0.5f * (result_a.y + result_b.y), 0.5f * (result_a.z + result_b.z));
            const float luma_result = luma_of(result);
            out = (luma_result < luma_min || luma_result > luma_max) ? result_a : result;
        }
        display[(size_t)j * width + i] = (to_unorm8(out.x) << 16) | (to_unorm8(out.y) << 8) | to_unorm8(out.z);
    }
    return RTX_OK;
}
