// rtx_host.cpp — librtx_host.so: host-side callers of the render path (see include/rtx_host.h).
//
// Compiled with -ffp-contract=off: the functions that mirror Scene::update must reproduce the
// reference's scalar Vector3 / Matrix4 / Quaternion arithmetic bit for bit (tests compare against
// matrices and TLAS nodes dumped from the real reference), and those classes evaluate
// left-to-right with plain mul/add (Vector3.h:33-43, Matrix4.h:31-68, Quaternion.h:128-133).
#include "../../include/rtx_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

struct V { float x, y, z; };
inline V mk(float x, float y, float z) { V r = { x, y, z }; return r; }
inline V operator+(V a, V b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V operator-(V a, V b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V operator*(V a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V operator*(float s, V a) { return mk(a.x * s, a.y * s, a.z * s); }
inline float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                                 // Vector3.h:33-35
inline V cross(V a, V b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }   // Vector3.h:37-43
inline V vmin(V a, V b) { return mk(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
inline V vmax(V a, V b) { return mk(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }
inline float comp(V a, int d) { return d == 0 ? a.x : d == 1 ? a.y : a.z; }

// Quaternion * Vector3, Quaternion.h:128-133
inline V rotate(const float q4[4], V v) {
    V q = mk(q4[0], q4[1], q4[2]);
    float w = q4[3];
    return 2.0f * dot(q, v) * q + (w * w - dot(q, q)) * v + 2.0f * w * cross(q, v);
}

// Matrix4(i, j) = cells[i + 4j], Matrix4.h:19-23
inline V m_position(const float * c, V p) {    // scalar Matrix4::transform_position, Matrix4.h:31-37
    return mk(c[0] * p.x + c[1] * p.y + c[2]  * p.z + c[3],
              c[4] * p.x + c[5] * p.y + c[6]  * p.z + c[7],
              c[8] * p.x + c[9] * p.y + c[10] * p.z + c[11]);
}
inline V m_direction(const float * c, V d) {   // scalar Matrix4::transform_direction, Matrix4.h:62-68
    return mk(c[0] * d.x + c[1] * d.y + c[2]  * d.z,
              c[4] * d.x + c[5] * d.y + c[6]  * d.z,
              c[8] * d.x + c[9] * d.y + c[10] * d.z);
}

// Transform::calc_world_matrix, Transform.h:13-43 (cells not written keep the identity of Matrix4())
void world_matrix(const float p[3], const float r[4], float * c) {
    memset(c, 0, 64); c[0] = c[5] = c[10] = c[15] = 1.0f;
    float xx = r[0] * r[0], yy = r[1] * r[1], zz = r[2] * r[2];
    float xz = r[0] * r[2], xy = r[0] * r[1], yz = r[1] * r[2];
    float wx = r[3] * r[0], wy = r[3] * r[1], wz = r[3] * r[2];
    c[0 + 4 * 0] = 1.0f - 2.0f * (yy + zz); c[0 + 4 * 1] = 2.0f * (xy + wz);        c[0 + 4 * 2] = 2.0f * (xz - wy);
    c[1 + 4 * 0] = 2.0f * (xy - wz);        c[1 + 4 * 1] = 1.0f - 2.0f * (xx + zz); c[1 + 4 * 2] = 2.0f * (yz + wx);
    c[2 + 4 * 0] = 2.0f * (xz + wy);        c[2 + 4 * 1] = 2.0f * (yz - wx);        c[2 + 4 * 2] = 1.0f - 2.0f * (xx + yy);
    c[3 + 4 * 0] = p[0]; c[3 + 4 * 1] = p[1]; c[3 + 4 * 2] = p[2];
}

// Matrix4::invert, Matrix4.h:88-138: adjugate by cofactors.  Row k of the table lists the six signed
// triple products of inv[k] in the order the reference sums them (left to right).
const signed char INV_TERMS[16][6][4] = {
    { { 1, 5, 10, 15 }, { -1, 5, 11, 14 }, { -1, 9, 6, 15 }, { 1, 9, 7, 14 }, { 1, 13, 6, 11 }, { -1, 13, 7, 10 } },
    { { -1, 1, 10, 15 }, { 1, 1, 11, 14 }, { 1, 9, 2, 15 }, { -1, 9, 3, 14 }, { -1, 13, 2, 11 }, { 1, 13, 3, 10 } },
    { { 1, 1, 6, 15 }, { -1, 1, 7, 14 }, { -1, 5, 2, 15 }, { 1, 5, 3, 14 }, { 1, 13, 2, 7 }, { -1, 13, 3, 6 } },
    { { -1, 1, 6, 11 }, { 1, 1, 7, 10 }, { 1, 5, 2, 11 }, { -1, 5, 3, 10 }, { -1, 9, 2, 7 }, { 1, 9, 3, 6 } },
    { { -1, 4, 10, 15 }, { 1, 4, 11, 14 }, { 1, 8, 6, 15 }, { -1, 8, 7, 14 }, { -1, 12, 6, 11 }, { 1, 12, 7, 10 } },
    { { 1, 0, 10, 15 }, { -1, 0, 11, 14 }, { -1, 8, 2, 15 }, { 1, 8, 3, 14 }, { 1, 12, 2, 11 }, { -1, 12, 3, 10 } },
    { { -1, 0, 6, 15 }, { 1, 0, 7, 14 }, { 1, 4, 2, 15 }, { -1, 4, 3, 14 }, { -1, 12, 2, 7 }, { 1, 12, 3, 6 } },
    { { 1, 0, 6, 11 }, { -1, 0, 7, 10 }, { -1, 4, 2, 11 }, { 1, 4, 3, 10 }, { 1, 8, 2, 7 }, { -1, 8, 3, 6 } },
    { { 1, 4, 9, 15 }, { -1, 4, 11, 13 }, { -1, 8, 5, 15 }, { 1, 8, 7, 13 }, { 1, 12, 5, 11 }, { -1, 12, 7, 9 } },
    { { -1, 0, 9, 15 }, { 1, 0, 11, 13 }, { 1, 8, 1, 15 }, { -1, 8, 3, 13 }, { -1, 12, 1, 11 }, { 1, 12, 3, 9 } },
    { { 1, 0, 5, 15 }, { -1, 0, 7, 13 }, { -1, 4, 1, 15 }, { 1, 4, 3, 13 }, { 1, 12, 1, 7 }, { -1, 12, 3, 5 } },
    { { -1, 0, 5, 11 }, { 1, 0, 7, 9 }, { 1, 4, 1, 11 }, { -1, 4, 3, 9 }, { -1, 8, 1, 7 }, { 1, 8, 3, 5 } },
    { { -1, 4, 9, 14 }, { 1, 4, 10, 13 }, { 1, 8, 5, 14 }, { -1, 8, 6, 13 }, { -1, 12, 5, 10 }, { 1, 12, 6, 9 } },
    { { 1, 0, 9, 14 }, { -1, 0, 10, 13 }, { -1, 8, 1, 14 }, { 1, 8, 2, 13 }, { 1, 12, 1, 10 }, { -1, 12, 2, 9 } },
    { { -1, 0, 5, 14 }, { 1, 0, 6, 13 }, { 1, 4, 1, 14 }, { -1, 4, 2, 13 }, { -1, 12, 1, 6 }, { 1, 12, 2, 5 } },
    { { 1, 0, 5, 10 }, { -1, 0, 6, 9 }, { -1, 4, 1, 10 }, { 1, 4, 2, 9 }, { 1, 8, 1, 6 }, { -1, 8, 2, 5 } },
};

void invert(const float * m, float * out) {
    float inv[16];
    for (int k = 0; k < 16; k++) {
        float acc = 0.0f;
        for (int t = 0; t < 6; t++) {
            const signed char * e = INV_TERMS[k][t];
            float a = m[e[1]]; if (e[0] < 0) a = -a;
            float term = a * m[e[2]] * m[e[3]];
            acc = (t == 0) ? term : acc + term;
        }
        inv[k] = acc;
    }
    memset(out, 0, 64); out[0] = out[5] = out[10] = out[15] = 1.0f;
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det != 0.0f) {
        const float inv_det = 1.0f / det;
        for (int i = 0; i < 16; i++) out[i] = inv[i] * inv_det;
    }
}

struct Box { V mn, mx; };
inline Box empty_box() { Box b; b.mn = mk(INFINITY, INFINITY, INFINITY); b.mx = mk(-INFINITY, -INFINITY, -INFINITY); return b; }
inline void expand(Box & b, const Box & o) { b.mn = vmin(b.mn, o.mn); b.mx = vmax(b.mx, o.mx); }
inline void expand(Box & b, V p) { b.mn = vmin(b.mn, p); b.mx = vmax(b.mx, p); }
inline void fix_if_needed(Box & b) {     // AABB::fix_if_needed, AABB.h:26-32: slabs need a non-zero extent
    if (b.mx.x - b.mn.x < 0.001f) b.mx.x += 0.005f;
    if (b.mx.y - b.mn.y < 0.001f) b.mx.y += 0.005f;
    if (b.mx.z - b.mn.z < 0.001f) b.mx.z += 0.005f;
}
inline float surface_area(const Box & b) {   // AABB.h:34-40
    V d = b.mx - b.mn;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}

}  // namespace

// =================================================================================================
extern "C" int rtxh_camera_basis(int32_t width, int32_t height, float fov, const float position[3], const float rotation[4], rtx_camera * out) {
    if (!position || !rotation || !out || width <= 0 || height <= 0) return RTX_ERR_INVALID_ARG;
    float half_width = 0.5f * width, half_height = 0.5f * height;          // Camera::resize, Camera.cpp:5-16
    float d = half_width / tanf(0.5f * fov);
    V tl = rotate(rotation, mk(-half_width, half_height, d));              // Camera::update, Camera.cpp:44-47
    V xa = rotate(rotation, mk(1.0f, 0.0f, 0.0f));
    V ya = rotate(rotation, mk(0.0f, -1.0f, 0.0f));
    memcpy(out->position, position, 12);
    out->rotated_top_left_corner[0] = tl.x; out->rotated_top_left_corner[1] = tl.y; out->rotated_top_left_corner[2] = tl.z;
    out->rotated_x_axis[0] = xa.x; out->rotated_x_axis[1] = xa.y; out->rotated_x_axis[2] = xa.z;
    out->rotated_y_axis[0] = ya.x; out->rotated_y_axis[1] = ya.y; out->rotated_y_axis[2] = ya.z;
    return RTX_OK;
}

extern "C" int rtxh_instance_update(const float position[3], const float rotation[4], const float mn[3], const float mx[3],
                                    rtx_instance * out, float out_min[3], float out_max[3]) {
    if (!position || !rotation || !mn || !mx || !out) return RTX_ERR_INVALID_ARG;
    world_matrix(position, rotation, out->world);                            // Mesh::update, Mesh.cpp:9-15
    // AABB::transform, AABB.cpp:55-73
    V bmn = mk(mn[0], mn[1], mn[2]), bmx = mk(mx[0], mx[1], mx[2]);
    V center = 0.5f * (bmn + bmx), extent = 0.5f * (bmx - bmn);
    float absm[16];
    for (int i = 0; i < 16; i++) absm[i] = fabsf(out->world[i]);
    V nc = m_position(out->world, center), ne = m_direction(absm, extent);
    V omn = nc - ne, omx = nc + ne;
    if (out_min) { out_min[0] = omn.x; out_min[1] = omn.y; out_min[2] = omn.z; }
    if (out_max) { out_max[0] = omx.x; out_max[1] = omx.y; out_max[2] = omx.z; }
    invert(out->world, out->world_inv);
    return RTX_OK;
}

extern "C" int rtxh_quaternion_axis_angle(const float axis[3], float angle, float out[4]) {    // Quaternion::axis_angle, Quaternion.h:26-36
    if (!axis || !out) return RTX_ERR_INVALID_ARG;
    float half_angle = 0.5f * angle;
    float sine = sinf(half_angle);
    out[0] = axis[0] * sine; out[1] = axis[1] * sine; out[2] = axis[2] * sine; out[3] = cosf(half_angle);
    return RTX_OK;
}

// ---- Scene::update, Scene.cpp:139-171 ---------------------------------------------------------------------------------
namespace {
struct Q { float x, y, z, w; };
inline Q q_axis_angle(float ax, float ay, float az, float angle) {                 // Quaternion.h:26-36
    float half_angle = 0.5f * angle, sine = sinf(half_angle);
    return Q{ ax * sine, ay * sine, az * sine, cosf(half_angle) };
}
inline Q q_mul(Q l, Q r) {                                                          // operator*(Quaternion, Quaternion&), Quaternion.h:118-125
    return Q{ l.x * r.w + l.w * r.x + l.y * r.z - l.z * r.y,
              l.y * r.w + l.w * r.y + l.z * r.x - l.x * r.z,
              l.z * r.w + l.w * r.z + l.x * r.y - l.y * r.x,
              l.w * r.w - l.x * r.x - l.y * r.y - l.z * r.z };
}
inline Q q_nlerp(Q a, Q b, float t) {                                               // Quaternion::nlerp + normalize
    float one_minus_t = 1.0f - t;
    Q q{ one_minus_t * a.x + t * b.x, one_minus_t * a.y + t * b.y, one_minus_t * a.z + t * b.z, one_minus_t * a.w + t * b.w };
    float inv_length = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Q{ q.x * inv_length, q.y * inv_length, q.z * inv_length, q.w * inv_length };
}
}  // namespace

// Camera::update's input half (Camera.cpp:18-39): WASD / shift / space move at 10 units/s, the arrow keys turn at 3 rad/s; each key
// acts on the state the previous key left (position first, then rotation; `right` and `forward` are taken once, before any of it).
extern "C" int rtxh_camera_update(float delta, uint32_t keys, float position[3], float rotation[4]) {
    if (!position || !rotation) return RTX_ERR_INVALID_ARG;
    const float MOVEMENT_SPEED = 10.0f, ROTATION_SPEED = 3.0f;
    const V right = rotate(rotation, mk(1.0f, 0.0f, 0.0f)), forward = rotate(rotation, mk(0.0f, 0.0f, 1.0f));
    V p = mk(position[0], position[1], position[2]);
    if (keys & RTXH_KEY_W) p = p + forward * MOVEMENT_SPEED * delta;
    if (keys & RTXH_KEY_A) p = p - right   * MOVEMENT_SPEED * delta;
    if (keys & RTXH_KEY_S) p = p - forward * MOVEMENT_SPEED * delta;
    if (keys & RTXH_KEY_D) p = p + right   * MOVEMENT_SPEED * delta;
    if (keys & RTXH_KEY_LSHIFT) p.y -= MOVEMENT_SPEED * delta;
    if (keys & RTXH_KEY_SPACE)  p.y += MOVEMENT_SPEED * delta;
    Q r{ rotation[0], rotation[1], rotation[2], rotation[3] };
    if (keys & RTXH_KEY_UP)    r = q_mul(q_axis_angle(right.x, right.y, right.z, -ROTATION_SPEED * delta), r);
    if (keys & RTXH_KEY_DOWN)  r = q_mul(q_axis_angle(right.x, right.y, right.z, +ROTATION_SPEED * delta), r);
    if (keys & RTXH_KEY_LEFT)  r = q_mul(q_axis_angle(0.0f, 1.0f, 0.0f, -ROTATION_SPEED * delta), r);
    if (keys & RTXH_KEY_RIGHT) r = q_mul(q_axis_angle(0.0f, 1.0f, 0.0f, +ROTATION_SPEED * delta), r);
    position[0] = p.x; position[1] = p.y; position[2] = p.z;
    rotation[0] = r.x; rotation[1] = r.y; rotation[2] = r.z; rotation[3] = r.w;
    return RTX_OK;
}

// The animation SCENE_DYNAMIC hard-codes into Scene::update (Scene.cpp:141-155): instances 0..5 move, *time accumulates delta.
extern "C" int rtxh_scene_dynamic_animate(float delta, float * time, float * positions, float * rotations, int32_t instance_count) {
    if (!time || !positions || !rotations || instance_count < 6) return RTX_ERR_INVALID_ARG;
    auto rot = [&](int i) -> Q & { return *reinterpret_cast<Q *>(rotations + 4 * i); };
    float * p = positions;
    rot(0) = q_mul(q_axis_angle(0.0f, 1.0f, 0.0f, delta), rot(0));
    *time += delta;
    const float t = *time;
    p[3 * 1 + 1] = 1.0f + 2.0f * sinf(t);
    p[3 * 2 + 0] -= delta * 0.5f;
    p[3 * 3 + 0] = 6.0f; p[3 * 3 + 1] = 4.0f + 2.0f * sinf(t * 0.5f); p[3 * 3 + 2] = 4.0f + 2.0f * cosf(t * 0.5f);
    rot(3) = q_mul(q_axis_angle(0.0f, 1.0f, 0.0f, delta * 0.5f), rot(3));
    rot(4) = q_mul(q_axis_angle(1.0f, 0.0f, 0.0f, delta), rot(4));
    rot(5) = q_nlerp(Q{ 0.0f, 0.0f, 0.0f, 1.0f }, q_axis_angle(1.0f, 0.0f, 0.0f, (-90.0f) * 3.14159265359f * 0.00555555555f), 0.5f + 0.5f * sinf(t));
    return RTX_OK;
}

// Mesh::update for every instance + TopLevelBVH::build_bvh in one call (the tail of Scene::update, Scene.cpp:166-170).
extern "C" int rtxh_scene_update(rtxh_tlas * tlas, int32_t instance_count, const float * positions, const float * rotations, const int32_t * blas_ids,
                                 const float * blas_root_aabbs, rtx_instance * instances_out, rtx_bvh_node * tlas_nodes_out,
                                 int32_t * tlas_indices_out, int32_t * tlas_node_count_out) {
    if (!tlas || instance_count <= 0 || !positions || !rotations || !blas_ids || !blas_root_aabbs || !instances_out || !tlas_nodes_out || !tlas_indices_out || !tlas_node_count_out)
        return RTX_ERR_INVALID_ARG;
    std::vector<float> aabbs(6 * (size_t)instance_count);
    for (int i = 0; i < instance_count; i++) {
        memset(&instances_out[i], 0, sizeof(rtx_instance));
        instances_out[i].blas_id = blas_ids[i];
        const float * box = blas_root_aabbs + 6 * (size_t)blas_ids[i];
        int rc = rtxh_instance_update(positions + 3 * i, rotations + 4 * i, box, box + 3, &instances_out[i], &aabbs[6 * (size_t)i], &aabbs[6 * (size_t)i + 3]);
        if (rc) return rc;
    }
    return rtxh_tlas_build(tlas, positions, aabbs.data(), tlas_nodes_out, tlas_indices_out, tlas_node_count_out);
}

extern "C" int rtxh_plane_update(const float position[3], const float rotation[4], int32_t material_id, rtx_plane * out) {
    if (!position || !rotation || !out) return RTX_ERR_INVALID_ARG;
    float w[16];
    world_matrix(position, rotation, w);                                     // Plane::update, Plane.cpp:3-11
    V n = m_direction(w, mk(0.0f, 1.0f, 0.0f));
    V p = mk(position[0], position[1], position[2]);
    float dist = -dot(n, p);
    V ua = m_direction(w, mk(1.0f, 0.0f, 0.0f));
    V va = cross(ua, n);
    memset(out, 0, sizeof(*out));
    out->normal[0] = n.x; out->normal[1] = n.y; out->normal[2] = n.z; out->distance = dist;
    out->u_axis[0] = ua.x; out->u_axis[1] = ua.y; out->u_axis[2] = ua.z;
    out->v_axis[0] = va.x; out->v_axis[1] = va.y; out->v_axis[2] = va.z;
    out->material_id = material_id;
    return RTX_OK;
}

// =================================================================================================
// TLAS: TopLevelBVH::init / build_bvh (TopLevelBVH.cpp:5-45) over BVHBuilders::build_bvh<Mesh>
// (BVHBuilders.h:8-46) and BVHPartitions::{calculate_bounds, partition_sah, split_indices}
// (BVHPartitions.h:10-114).  Same algorithm, same index-array persistence across frames, so the
// rebuilt tree is the reference's node for node (checked against dumped TLAS nodes in tests).
struct rtxh_tlas {
    int n;
    std::vector<int> idx[3];
    std::vector<float> sah;
    std::vector<int> temp;
};

namespace {
struct TlasBuild {        // BVHBuilders::build_bvh<PrimitiveType>: used for Mesh instances (TLAS) and for triangles (reference BLAS)
    const float * pos; const float * aabb; rtxh_tlas * t; rtx_bvh_node * nodes; int node_index;
    bool degenerate = false;     // NaN keys: the three sorted lists stop agreeing about who is left of the split (the reference overruns `temp` there)
    Box prim_box(int i) const { Box b; b.mn = mk(aabb[6 * i], aabb[6 * i + 1], aabb[6 * i + 2]); b.mx = mk(aabb[6 * i + 3], aabb[6 * i + 4], aabb[6 * i + 5]); return b; }
    float p(int i, int d) const { return pos[3 * i + d]; }

    void build(int node_id, int first, int count) {
        rtx_bvh_node & node = nodes[node_id];
        Box b = empty_box();                                            // calculate_bounds over indices[0]
        for (int i = first; i < first + count; i++) expand(b, prim_box(t->idx[0][i]));
        fix_if_needed(b);
        node.aabb_min[0] = b.mn.x; node.aabb_min[1] = b.mn.y; node.aabb_min[2] = b.mn.z;
        node.aabb_max[0] = b.mx.x; node.aabb_max[1] = b.mx.y; node.aabb_max[2] = b.mx.z;
        if (count < 3) { node.left_or_first = first; node.count = count; return; }
        const int left = node_index;
        node.left_or_first = left;
        node_index += 2;                                                // allocated before the SAH check, as the reference does

        // partition_sah: full sweep over the three sorted index lists
        float min_cost = INFINITY; int min_index = -1, min_dim = -1;
        for (int d = 0; d < 3; d++) {
            Box l = empty_box(), r = empty_box();
            for (int i = 0; i < count - 1; i++) { expand(l, prim_box(t->idx[d][first + i])); t->sah[i] = surface_area(l) * float(i + 1); }
            for (int i = count - 1; i > 0; i--) {
                expand(r, prim_box(t->idx[d][first + i]));
                float cost = t->sah[i - 1] + surface_area(r) * float(count - i);
                if (cost < min_cost) { min_cost = cost; min_index = first + i; min_dim = d; }
            }
        }
        float parent_cost = surface_area(b) * float(count);
        if (min_index < 0) { degenerate = true; node.left_or_first = first; node.count = count; return; }   // every cost NaN: nothing to split on
        if (min_cost >= parent_cost) { node.left_or_first = first; node.count = count; return; }

        // split_indices: keep the other two lists consistent with the chosen split
        const float split = p(t->idx[min_dim][min_index], min_dim);
        for (int d = 0; d < 3; d++) {
            if (d == min_dim) continue;
            int l = 0, r = min_index - first;
            for (int i = first; i < first + count; i++) {
                const int prim = t->idx[d][i];
                bool goes_left = p(prim, min_dim) < split;
                if (p(prim, min_dim) == split) {
                    int j = min_index - 1;
                    while (j >= first && p(t->idx[min_dim][j], min_dim) == split) {
                        if (t->idx[min_dim][j] == prim) { goes_left = true; break; }
                        j--;
                    }
                }
                if (goes_left) { if (l >= min_index - first) { degenerate = true; break; } t->temp[l++] = prim; }
                else           { if (r >= count)             { degenerate = true; break; } t->temp[r++] = prim; }
            }
            if (degenerate) { node.left_or_first = first; node.count = count; return; }
            memcpy(&t->idx[d][first], t->temp.data(), sizeof(int) * count);
        }
        node.count = (min_dim + 1) << 30;
        const int n_left = min_index - first, n_right = first + count - min_index;
        build(left, first, n_left);
        build(left + 1, first + n_left, n_right);
    }
};
}  // namespace

extern "C" int rtxh_tlas_create(int32_t n, rtxh_tlas ** out) {
    if (n <= 0 || !out) return RTX_ERR_INVALID_ARG;
    rtxh_tlas * t = new rtxh_tlas();
    t->n = n;
    for (int d = 0; d < 3; d++) { t->idx[d].resize(n); for (int i = 0; i < n; i++) t->idx[d][i] = i; }
    t->sah.resize(n); t->temp.resize(n);
    *out = t;
    return RTX_OK;
}
extern "C" int rtxh_tlas_destroy(rtxh_tlas * t) { if (!t) return RTX_ERR_INVALID_ARG; delete t; return RTX_OK; }

extern "C" int rtxh_tlas_build(rtxh_tlas * t, const float * positions, const float * aabbs,
                               rtx_bvh_node * nodes_out, int32_t * indices_out, int32_t * node_count_out) {
    if (!t || !positions || !aabbs || !nodes_out || !indices_out || !node_count_out) return RTX_ERR_INVALID_ARG;
    const int n = t->n;
    for (int d = 0; d < 3; d++)
        std::sort(t->idx[d].begin(), t->idx[d].end(), [&](int a, int b) { return positions[3 * a + d] < positions[3 * b + d]; });
    memset(nodes_out, 0, sizeof(rtx_bvh_node) * 2 * (size_t)n);
    TlasBuild b; b.pos = positions; b.aabb = aabbs; b.t = t; b.nodes = nodes_out; b.node_index = 2;
    b.build(0, 0, n);
    memcpy(indices_out, t->idx[0].data(), sizeof(int) * n);
    *node_count_out = b.node_index;
    return b.degenerate ? RTX_ERR_STATE : RTX_OK;
}

// The reference's non-spatial BLAS build (BottomLevelBVH::build_bvh, BottomLevelBVH.cpp:72-106): the same generic builder
// over triangles, keyed by Triangle::get_position (Triangle.h:22-24) with AABBs from Triangle::calc_aabb (:14-20).
extern "C" int rtxh_blas_build_reference_bvh(const float * positions, int32_t n, rtx_bvh_node * nodes_out, int32_t * node_count_out, int32_t * order_out) {
    if (!positions || n <= 0 || !nodes_out || !node_count_out || !order_out) return RTX_ERR_INVALID_ARG;
    std::vector<float> cen(3 * (size_t)n), box(6 * (size_t)n);
    for (int i = 0; i < n; i++) {
        const float * p = positions + 9 * (size_t)i;
        V a = mk(p[0], p[1], p[2]), b = mk(p[3], p[4], p[5]), c = mk(p[6], p[7], p[8]);
        Box t = empty_box(); expand(t, a); expand(t, b); expand(t, c);
        fix_if_needed(t); fix_if_needed(t);                                   // AABB::from_points, then calc_aabb's own call
        V g = (a + b + c) * (1.0f / 3.0f);
        cen[3 * (size_t)i] = g.x; cen[3 * (size_t)i + 1] = g.y; cen[3 * (size_t)i + 2] = g.z;
        box[6 * (size_t)i] = t.mn.x; box[6 * (size_t)i + 1] = t.mn.y; box[6 * (size_t)i + 2] = t.mn.z;
        box[6 * (size_t)i + 3] = t.mx.x; box[6 * (size_t)i + 4] = t.mx.y; box[6 * (size_t)i + 5] = t.mx.z;
    }
    rtxh_tlas * t = nullptr;
    int rc = rtxh_tlas_create(n, &t);
    if (rc) return rc;
    rc = rtxh_tlas_build(t, cen.data(), box.data(), nodes_out, order_out, node_count_out);
    rtxh_tlas_destroy(t);
    return rc;
}

// =================================================================================================
// BLAS: this repo's own builder — top-down binned SAH over triangle centroids.  Output follows the
// conventions the traversal relies on (BVHNode.h:10-28): root 0, index 1 unused, children adjacent,
// split axis in the top two bits of `count`, leaves index a flattened triangle order.
namespace {
struct BlasBuild {
    const float * pos; int bins;
    std::vector<Box> tb; std::vector<V> cen; std::vector<int> order;
    rtx_bvh_node * nodes; int node_index;

    void set_box(rtx_bvh_node & n, Box b) {
        fix_if_needed(b);
        n.aabb_min[0] = b.mn.x; n.aabb_min[1] = b.mn.y; n.aabb_min[2] = b.mn.z;
        n.aabb_max[0] = b.mx.x; n.aabb_max[1] = b.mx.y; n.aabb_max[2] = b.mx.z;
    }

    void build(int node_id, int first, int count) {
        Box b = empty_box(), cb = empty_box();
        for (int i = first; i < first + count; i++) { expand(b, tb[order[i]]); expand(cb, cen[order[i]]); }
        set_box(nodes[node_id], b);
        if (count < 3) { nodes[node_id].left_or_first = first; nodes[node_id].count = count; return; }

        float best_cost = INFINITY; int best_dim = -1, best_bin = -1;
        const int NB = bins;
        std::vector<Box> bb(NB); std::vector<int> bc(NB);
        std::vector<float> right_sa(NB);
        for (int d = 0; d < 3; d++) {
            const float lo = comp(cb.mn, d), hi = comp(cb.mx, d);
            if (!(hi > lo)) continue;
            const float scale = float(NB) / (hi - lo);
            for (int k = 0; k < NB; k++) { bb[k] = empty_box(); bc[k] = 0; }
            for (int i = first; i < first + count; i++) {
                int k = (int)((comp(cen[order[i]], d) - lo) * scale); if (k >= NB) k = NB - 1; if (k < 0) k = 0;
                expand(bb[k], tb[order[i]]); bc[k]++;
            }
            Box r = empty_box(); int rc = 0;
            for (int k = NB - 1; k > 0; k--) { if (bc[k]) expand(r, bb[k]); rc += bc[k]; right_sa[k] = rc ? surface_area(r) * float(rc) : 0.0f; }
            Box l = empty_box(); int lc = 0;
            for (int k = 0; k < NB - 1; k++) {
                if (bc[k]) expand(l, bb[k]); lc += bc[k];
                if (lc == 0 || lc == count) continue;
                const float cost = surface_area(l) * float(lc) + right_sa[k + 1];
                if (cost < best_cost) { best_cost = cost; best_dim = d; best_bin = k; }
            }
        }
        Box fb = b; fix_if_needed(fb);
        const float parent_cost = surface_area(fb) * float(count);
        int mid;
        if (best_dim < 0 || best_cost >= parent_cost) {
            if (count <= 4) { nodes[node_id].left_or_first = first; nodes[node_id].count = count; return; }
            // no useful SAH split for a large set (coincident centroids): split the index range in half
            best_dim = 0; { V e = cb.mx - cb.mn; if (e.y > e.x) best_dim = 1; if (e.z > comp(e, best_dim)) best_dim = 2; }
            mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](int a, int c) { return comp(cen[a], best_dim) < comp(cen[c], best_dim); });
        } else {
            const float lo = comp(cb.mn, best_dim), hi = comp(cb.mx, best_dim);
            const float scale = float(NB) / (hi - lo);
            int * beg = order.data() + first;
            int * m = std::partition(beg, beg + count, [&](int a) {
                int k = (int)((comp(cen[a], best_dim) - lo) * scale); if (k >= NB) k = NB - 1; if (k < 0) k = 0;
                return k <= best_bin; });
            mid = (int)(m - order.data());
        }
        const int left = node_index; node_index += 2;
        nodes[node_id].left_or_first = left;
        nodes[node_id].count = (best_dim + 1) << 30;
        build(left, first, mid - first);
        build(left + 1, mid, first + count - mid);
    }
};
}  // namespace

extern "C" int rtxh_blas_build(const float * positions, int32_t n, int32_t bins, rtx_bvh_node * nodes_out, int32_t * node_count_out, int32_t * order_out) {
    if (!positions || n <= 0 || !nodes_out || !node_count_out || !order_out) return RTX_ERR_INVALID_ARG;
    if (bins < 4) bins = 4;
    if (bins > 256) bins = 256;
    BlasBuild b; b.pos = positions; b.bins = bins; b.tb.resize(n); b.cen.resize(n); b.order.resize(n);
    for (int i = 0; i < n; i++) {
        const float * p = positions + 9 * (size_t)i;
        Box t = empty_box();
        expand(t, mk(p[0], p[1], p[2])); expand(t, mk(p[3], p[4], p[5])); expand(t, mk(p[6], p[7], p[8]));
        b.cen[i] = 0.5f * (t.mn + t.mx);
        fix_if_needed(t);
        b.tb[i] = t; b.order[i] = i;
    }
    memset(nodes_out, 0, sizeof(rtx_bvh_node) * 2 * (size_t)n);
    b.nodes = nodes_out; b.node_index = 2;
    b.build(0, 0, n);
    memcpy(order_out, b.order.data(), sizeof(int) * n);
    *node_count_out = b.node_index;
    return RTX_OK;
}

// =================================================================================================
// Texture::load mip chain, Texture.cpp:76-117 (box filter, levels appended after level 0)
extern "C" int rtxh_texture_mips(float * tx, int32_t width, int32_t height, rtx_texture_desc * desc, int64_t * texel_count_out) {
    if (!tx || !desc || width <= 0 || height <= 0) return RTX_ERR_INVALID_ARG;
    memset(desc, 0, sizeof(*desc));
    desc->width = width; desc->height = height;
    const bool pow2 = ((width & (width - 1)) == 0) && ((height & (height - 1)) == 0);   // Math::is_power_of_two, Texture.cpp:50
    if (!pow2) { desc->mipmapped = 0; desc->mip_levels = 1; if (texel_count_out) *texel_count_out = (int64_t)width * height; return RTX_OK; }
    desc->mipmapped = 1;
    desc->mip_levels = 1 + (int)log2f((float)(width < height ? width : height));
    if (desc->mip_levels > RTX_MAX_MIP_LEVELS) return RTX_ERR_LIMIT;
    int offset = width * height, offset_prev = 0;
    int lw = width >> 1, lh = height >> 1, lwp = width;
    int level = 1;
    while (lw >= 1 && lh >= 1) {
        for (int j = 0; j < lh; j++) for (int i = 0; i < lw; i++) {
            const int ip = i << 1, jp = j << 1;
            const float * c0 = tx + 3 * (size_t)(offset_prev + ip + jp * lwp);
            const float * c1 = tx + 3 * (size_t)(offset_prev + (ip + 1) + jp * lwp);
            const float * c2 = tx + 3 * (size_t)(offset_prev + ip + (jp + 1) * lwp);
            const float * c3 = tx + 3 * (size_t)(offset_prev + (ip + 1) + (jp + 1) * lwp);
            float * o = tx + 3 * (size_t)(offset + i + j * lw);
            for (int k = 0; k < 3; k++) o[k] = (((c0[k] + c1[k]) + c2[k]) + c3[k]) * 0.25f;
        }
        desc->mip_offsets[level++] = offset;
        offset_prev = offset;
        offset += lw * lh;
        lwp = lw;
        lw >>= 1; lh >>= 1;
    }
    if (texel_count_out) *texel_count_out = offset;
    return RTX_OK;
}

// =================================================================================================
// Procedural atrium: a seeded Sponza-class stand-in (the real sponza.obj is absent from the mount).
// Colonnaded hall with arches, balustrades, curtains, vases and rough "statues"; triangle sizes span
// three orders of magnitude on purpose (large wall quads next to finely tessellated ornaments).
namespace {
struct MeshOut {
    std::vector<float> pos, nrm, uv; std::vector<int> mat;
    void tri(V a, V b, V c, V na, V nb, V nc, float ua, float va, float ub, float vb, float uc, float vc, int m) {
        const V p[3] = { a, b, c }, n[3] = { na, nb, nc };
        for (int k = 0; k < 3; k++) { pos.push_back(p[k].x); pos.push_back(p[k].y); pos.push_back(p[k].z); nrm.push_back(n[k].x); nrm.push_back(n[k].y); nrm.push_back(n[k].z); }
        // texture v is stored flipped, as the reference's loader does (OBJLoader.cpp:139-141)
        uv.push_back(ua); uv.push_back(1.0f - va); uv.push_back(ub); uv.push_back(1.0f - vb); uv.push_back(uc); uv.push_back(1.0f - vc);
        mat.push_back(m);
    }
};
inline V norm(V v) { float l = sqrtf(dot(v, v)); return l > 0 ? v * (1.0f / l) : mk(0, 1, 0); }

struct Rng { uint32_t s; float next() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); } };

// parametric surface patch: P(u,v), N(u,v) sampled on an (nu x nv) grid
template <typename F>
void patch(MeshOut & m, int nu, int nv, int material, float uscale, float vscale, bool flip, F && f) {
    std::vector<V> P((size_t)(nu + 1) * (nv + 1)), N(P.size());
    for (int j = 0; j <= nv; j++) for (int i = 0; i <= nu; i++) f((float)i / nu, (float)j / nv, P[(size_t)j * (nu + 1) + i], N[(size_t)j * (nu + 1) + i]);
    for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
        const size_t a = (size_t)j * (nu + 1) + i, b = a + 1, c = a + (nu + 1), d = c + 1;
        const float u0 = uscale * i / nu, u1 = uscale * (i + 1) / nu, v0 = vscale * j / nv, v1 = vscale * (j + 1) / nv;
        if (!flip) { m.tri(P[a], P[b], P[d], N[a], N[b], N[d], u0, v0, u1, v0, u1, v1, material); m.tri(P[a], P[d], P[c], N[a], N[d], N[c], u0, v0, u1, v1, u0, v1, material); }
        else       { m.tri(P[a], P[d], P[b], N[a], N[d], N[b], u0, v0, u1, v1, u1, v0, material); m.tri(P[a], P[c], P[d], N[a], N[c], N[d], u0, v0, u0, v1, u1, v1, material); }
    }
}
inline float hash2(int x, int y, uint32_t seed) { uint32_t h = (uint32_t)x * 374761393u + (uint32_t)y * 668265263u + seed * 2246822519u; h = (h ^ (h >> 13)) * 1274126177u; h ^= h >> 16; return (float)(h & 0xffffff) * (1.0f / 16777216.0f); }
inline float vnoise(float x, float y, uint32_t seed) {
    const int xi = (int)floorf(x), yi = (int)floorf(y); const float fx = x - xi, fy = y - yi;
    const float sx = fx * fx * (3 - 2 * fx), sy = fy * fy * (3 - 2 * fy);
    const float a = hash2(xi, yi, seed), b = hash2(xi + 1, yi, seed), c = hash2(xi, yi + 1, seed), d = hash2(xi + 1, yi + 1, seed);
    return (a + (b - a) * sx) + ((c + (d - c) * sx) - (a + (b - a) * sx)) * sy;
}
}  // namespace

extern "C" int rtxh_atrium_generate(uint32_t seed, int32_t detail, rtxh_mesh * out) {
    if (!out || detail < 0 || detail > 8) return RTX_ERR_INVALID_ARG;
    MeshOut m; Rng rng = { seed };
    const float PI = 3.14159265358979f;
    // tessellation multiplier: detail 1 ~ 255k triangles (the cfg3 size), detail 0 ~ 64k, detail 4 ~ 1.6M
    const float q = 0.25f * (float)(detail + 1);
    auto T = [&](int base) { int v = (int)(base * q + 0.5f); return v < 2 ? 2 : v; };
    const float HX = 30.0f, HZ = 12.0f, HY = 18.0f;
    enum { M_FLOOR, M_WALL, M_ENDWALL, M_COLUMN, M_CAPITAL, M_ARCH, M_BALUSTER, M_CURTAIN0, M_VASE0 = M_CURTAIN0 + 6, M_STATUE0 = M_VASE0 + 4, M_TRIM = M_STATUE0 + 4, M_COUNT = M_TRIM + 4 };

    // floor: gently uneven flagstones, normals from the height field
    patch(m, T(320), T(128), M_FLOOR, 30.0f, 12.0f, false, [&](float u, float v, V & P, V & N) {
        const float x = -HX + 2 * HX * u, z = -HZ + 2 * HZ * v;
        auto h = [&](float xx, float zz) { return 0.03f * vnoise(xx * 1.7f, zz * 1.7f, seed) + 0.01f * vnoise(xx * 9.1f, zz * 7.7f, seed + 1); };
        P = mk(x, h(x, z), z);
        const float e = 0.05f; N = norm(mk(h(x - e, z) - h(x + e, z), 2 * e, h(x, z - e) - h(x, z + e)));
    });
    // side walls with relief, facing inward
    for (int side = 0; side < 2; side++) {
        const float zs = side ? HZ : -HZ, sgn = side ? -1.0f : 1.0f;
        patch(m, T(240), T(80), M_WALL, 24.0f, 8.0f, side == 0, [&](float u, float v, V & P, V & N) {
            const float x = -HX + 2 * HX * u, y = HY * v;
            auto r = [&](float xx, float yy) { return 0.12f * vnoise(xx * 0.9f, yy * 0.9f, seed + 7 + side) + 0.02f * vnoise(xx * 6.3f, yy * 6.1f, seed + 9); };
            P = mk(x, y, zs + sgn * r(x, y));
            const float e = 0.05f; N = norm(mk(sgn * (r(x - e, y) - r(x + e, y)), sgn * (r(x, y - e) - r(x, y + e)), sgn * 2 * e));
        });
    }
    // end walls (large flat quads: big triangles)
    for (int side = 0; side < 2; side++) {
        const float xs = side ? HX : -HX, sgn = side ? -1.0f : 1.0f;
        patch(m, T(24), T(20), M_ENDWALL, 10.0f, 8.0f, side == 1, [&](float u, float v, V & P, V & N) {
            P = mk(xs, HY * v, -HZ + 2 * HZ * u); N = mk(sgn, 0, 0);
        });
    }
    // two rows of columns with capitals, arches between neighbours
    const int NCOL = 10;
    for (int row = 0; row < 2; row++) {
        const float cz = row ? 7.0f : -7.0f;
        for (int c = 0; c < NCOL; c++) {
            const float cx = -HX + 3.0f + c * ((2 * HX - 6.0f) / (NCOL - 1));
            const float rad = 0.8f, hgt = 10.0f;
            patch(m, T(96), T(80), M_COLUMN, 4.0f, 10.0f, true, [&](float u, float v, V & P, V & N) {
                const float a = 2 * PI * u, fl = 1.0f + 0.04f * cosf(16.0f * a);      // fluted shaft
                const float r = rad * fl * (1.0f - 0.12f * v);
                P = mk(cx + r * cosf(a), hgt * v, cz + r * sinf(a)); N = norm(mk(cosf(a), 0.1f, sinf(a)));
            });
            patch(m, T(96), T(32), M_CAPITAL, 4.0f, 1.0f, true, [&](float u, float v, V & P, V & N) {
                const float a = 2 * PI * u, b = PI * (v - 0.5f);
                const float R = 0.95f + 0.35f * cosf(b);
                P = mk(cx + R * cosf(a), hgt + 0.35f + 0.35f * sinf(b), cz + R * sinf(a)); N = norm(mk(cosf(a) * cosf(b), sinf(b), sinf(a) * cosf(b)));
            });
            if (c + 1 < NCOL) {
                const float span = (2 * HX - 6.0f) / (NCOL - 1), mx = cx + 0.5f * span;
                patch(m, T(64), T(32), M_ARCH, 6.0f, 2.0f, true, [&](float u, float v, V & P, V & N) {
                    const float a = PI * u, b = 2 * PI * v, R = 0.5f * span, r = 0.45f;
                    const V cdir = mk(-cosf(a), sinf(a), 0.0f);
                    P = mk(mx, hgt + 0.7f, cz) + cdir * (R + r * cosf(b)) + mk(0, 0, r * sinf(b));
                    N = norm(cdir * cosf(b) + mk(0, 0, sinf(b)));
                });
            }
        }
    }
    // gallery balustrades: many small balusters
    for (int row = 0; row < 2; row++) {
        const float bz = row ? 8.2f : -8.2f;
        const int NB = 60;
        for (int b = 0; b < NB; b++) {
            const float bx = -HX + 1.5f + b * ((2 * HX - 3.0f) / (NB - 1));
            patch(m, T(24), T(20), M_BALUSTER, 1.0f, 1.0f, true, [&](float u, float v, V & P, V & N) {
                const float a = 2 * PI * u, r = 0.09f + 0.06f * sinf(PI * v) * sinf(PI * v) + 0.03f * cosf(6 * PI * v);
                P = mk(bx + r * cosf(a), 12.0f + 1.1f * v, bz + r * sinf(a)); N = norm(mk(cosf(a), 0.2f * cosf(PI * v), sinf(a)));
            });
        }
        patch(m, T(120), 2, M_TRIM, 30.0f, 0.2f, false, [&](float u, float v, V & P, V & N) {   // hand rail
            P = mk(-HX + 1.5f + (2 * HX - 3.0f) * u, 13.15f, bz - 0.12f + 0.24f * v); N = mk(0, 1, 0);
        });
    }
    // curtains: wavy sheets hanging between columns
    for (int k = 0; k < 6; k++) {
        const float cx0 = -24.0f + k * 9.0f, cz = (k & 1) ? 7.6f : -7.6f;
        const float ph = rng.next() * 6.28f;
        patch(m, T(80), T(120), M_CURTAIN0 + k, 3.0f, 5.0f, (k & 1) != 0, [&](float u, float v, V & P, V & N) {
            const float x = cx0 + 5.0f * u, y = 11.5f - 8.5f * v;
            auto w = [&](float uu, float vv) { return 0.28f * sinf(14.0f * uu + ph) * (0.3f + 0.7f * vv) + 0.05f * sinf(40.0f * uu + 3.0f * vv); };
            P = mk(x, y, cz + w(u, v));
            const float e = 0.002f; const float dzdu = (w(u + e, v) - w(u - e, v)) / (2 * e * 5.0f), dzdv = (w(u, v + e) - w(u, v - e)) / (2 * e * -8.5f);
            N = norm(mk(-dzdu, -dzdv, 1.0f) * ((k & 1) ? -1.0f : 1.0f));
        });
    }
    // vases: glass and polished metal
    for (int k = 0; k < 4; k++) {
        const float vx = -13.5f + k * 9.0f, vz = (k & 1) ? 2.5f : -2.5f;
        patch(m, T(64), T(48), M_VASE0 + k, 2.0f, 2.0f, true, [&](float u, float v, V & P, V & N) {
            const float a = 2 * PI * u; const float t = v;
            const float r = 0.25f + 0.55f * sinf(PI * (0.15f + 0.8f * t)) * (1.0f - 0.45f * t);
            P = mk(vx + r * cosf(a), 0.05f + 2.2f * t, vz + r * sinf(a));
            const float e = 0.01f, t1 = t + e; const float r1 = 0.25f + 0.55f * sinf(PI * (0.15f + 0.8f * t1)) * (1.0f - 0.45f * t1);
            const float dr = (r1 - r) / (2.2f * e);
            N = norm(mk(cosf(a), -dr, sinf(a)));
        });
    }
    // rough statues: noisy spheres on plinths
    for (int k = 0; k < 4; k++) {
        const float sx = -18.0f + k * 12.0f, sz = (k & 1) ? -3.5f : 3.5f;
        patch(m, T(128), T(96), M_STATUE0 + k, 3.0f, 3.0f, true, [&](float u, float v, V & P, V & N) {
            const float a = 2 * PI * u, b = PI * (v - 0.5f);
            const V d = mk(cosf(a) * cosf(b), sinf(b), sinf(a) * cosf(b));
            const float r = 0.9f + 0.18f * vnoise(6.0f * u * 3.0f + k * 11.0f, 6.0f * v * 3.0f, seed + 31) + 0.05f * vnoise(40.0f * u, 40.0f * v, seed + 33);
            P = mk(sx, 2.0f, sz) + d * r; N = d;
        });
        patch(m, 4, 2, M_TRIM + 1, 1.0f, 1.0f, true, [&](float u, float v, V & P, V & N) {       // plinth (large triangles)
            const float a = 2 * PI * u + 0.25f * PI;
            P = mk(sx + 0.9f * cosf(a), 1.1f * v, sz + 0.9f * sinf(a)); N = norm(mk(cosf(a), 0, sinf(a)));
        });
    }
    // partial roof with a skylight slot (lets the sun and the sky in)
    for (int side = 0; side < 2; side++) {
        const float z0 = side ? 4.0f : -HZ, z1 = side ? HZ : -4.0f;
        patch(m, T(40), T(8), M_TRIM + 2, 12.0f, 3.0f, true, [&](float u, float v, V & P, V & N) {
            P = mk(-HX + 2 * HX * u, HY, z0 + (z1 - z0) * v); N = mk(0, -1, 0);
        });
    }

    const int n = (int)m.mat.size();
    out->triangle_count = n; out->material_count = M_COUNT;
    out->positions = (float *)malloc(sizeof(float) * 9 * (size_t)n);
    out->normals   = (float *)malloc(sizeof(float) * 9 * (size_t)n);
    out->texcoords = (float *)malloc(sizeof(float) * 6 * (size_t)n);
    out->material_ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!out->positions || !out->normals || !out->texcoords || !out->material_ids) return RTX_ERR_OOM;
    memcpy(out->positions, m.pos.data(), sizeof(float) * 9 * (size_t)n);
    memcpy(out->normals, m.nrm.data(), sizeof(float) * 9 * (size_t)n);
    memcpy(out->texcoords, m.uv.data(), sizeof(float) * 6 * (size_t)n);
    memcpy(out->material_ids, m.mat.data(), sizeof(int32_t) * (size_t)n);
    return RTX_OK;
}

extern "C" int rtxh_mesh_free(rtxh_mesh * m) {
    if (!m) return RTX_ERR_INVALID_ARG;
    free(m->positions); free(m->normals); free(m->texcoords); free(m->material_ids);
    memset(m, 0, sizeof(*m));
    return RTX_OK;
}

// =================================================================================================
// `<mesh>.obj.bvh` cache files, BottomLevelBVH::save_to_disk / load_from_disk (BottomLevelBVH.cpp:149-192):
//   int triangle_count; TriangleHot[n]; TriangleCold[n]; int node_count; BVHNode[node_count]; int index_count; int[index_count]
// The record layouts are the C ABI's, so the arrays are handed over as they are; BottomLevelBVH::flatten (:196-212) is the caller's gather.
extern "C" int rtxh_bvh_cache_load(const char * path, rtxh_bvh_cache * out) {
    if (!path || !out) return RTX_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    FILE * f = fopen(path, "rb");
    if (!f) return RTX_ERR_STATE;
    fseek(f, 0, SEEK_END); const long size = ftell(f); fseek(f, 0, SEEK_SET);
    auto fail = [&](int rc) { fclose(f); rtxh_bvh_cache_free(out); return rc; };
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1 || n <= 0 || (long long)n * 100 + 12 > size) return fail(RTX_ERR_INVALID_ARG);
    out->triangle_count = n;
    out->hot = (rtx_triangle_hot *)malloc(sizeof(rtx_triangle_hot) * (size_t)n);
    out->cold = (rtx_triangle_cold *)malloc(sizeof(rtx_triangle_cold) * (size_t)n);
    if (!out->hot || !out->cold) return fail(RTX_ERR_OOM);
    if (fread(out->hot, sizeof(rtx_triangle_hot), n, f) != (size_t)n || fread(out->cold, sizeof(rtx_triangle_cold), n, f) != (size_t)n) return fail(RTX_ERR_INVALID_ARG);
    int32_t nodes = 0;
    if (fread(&nodes, 4, 1, f) != 1 || nodes <= 0 || (long long)nodes * 32 > size) return fail(RTX_ERR_INVALID_ARG);
    out->node_count = nodes;
    out->nodes = (rtx_bvh_node *)malloc(sizeof(rtx_bvh_node) * (size_t)nodes);
    if (!out->nodes) return fail(RTX_ERR_OOM);
    if (fread(out->nodes, sizeof(rtx_bvh_node), nodes, f) != (size_t)nodes) return fail(RTX_ERR_INVALID_ARG);
    int32_t idx = 0;
    if (fread(&idx, 4, 1, f) != 1 || idx <= 0 || (long long)idx * 4 > size) return fail(RTX_ERR_INVALID_ARG);
    out->index_count = idx;
    out->indices = (int32_t *)malloc(4 * (size_t)idx);
    if (!out->indices) return fail(RTX_ERR_OOM);
    if (fread(out->indices, 4, idx, f) != (size_t)idx) return fail(RTX_ERR_INVALID_ARG);
    for (int i = 0; i < idx; i++) if (out->indices[i] < 0 || out->indices[i] >= n) return fail(RTX_ERR_INVALID_ARG);
    if (ftell(f) != size) return fail(RTX_ERR_INVALID_ARG);          // trailing bytes: not a cache file of this layout
    fclose(f);
    return RTX_OK;
}

extern "C" int rtxh_bvh_cache_save(const char * path, const rtxh_bvh_cache * c) {
    if (!path || !c || c->triangle_count <= 0 || c->node_count <= 0 || c->index_count <= 0 || !c->hot || !c->cold || !c->nodes || !c->indices) return RTX_ERR_INVALID_ARG;
    FILE * f = fopen(path, "wb");
    if (!f) return RTX_ERR_STATE;
    bool ok = fwrite(&c->triangle_count, 4, 1, f) == 1
           && fwrite(c->hot, sizeof(rtx_triangle_hot), c->triangle_count, f) == (size_t)c->triangle_count
           && fwrite(c->cold, sizeof(rtx_triangle_cold), c->triangle_count, f) == (size_t)c->triangle_count
           && fwrite(&c->node_count, 4, 1, f) == 1 && fwrite(c->nodes, sizeof(rtx_bvh_node), c->node_count, f) == (size_t)c->node_count
           && fwrite(&c->index_count, 4, 1, f) == 1 && fwrite(c->indices, 4, c->index_count, f) == (size_t)c->index_count;
    ok = (fclose(f) == 0) && ok;
    return ok ? RTX_OK : RTX_ERR_STATE;
}

extern "C" int rtxh_bvh_cache_free(rtxh_bvh_cache * c) {
    if (!c) return RTX_ERR_INVALID_ARG;
    free(c->hot); free(c->cold); free(c->nodes); free(c->indices);
    memset(c, 0, sizeof(*c));
    return RTX_OK;
}
