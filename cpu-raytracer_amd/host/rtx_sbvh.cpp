// rtx_sbvh.cpp — the reference's SBVH (spatial-split BVH) construction for a BottomLevelBVH, restated (CPU, load time).
//
//   rtxh_blas_build_reference_sbvh      BottomLevelBVH::build_sbvh          BottomLevelBVH.cpp:108-147
//     Sbvh::build                       BVHBuilders::build_sbvh             BVHBuilders.h:48-329
//     Sbvh::object_split                BVHPartitions::partition_object     BVHPartitions.h:117-171
//     Sbvh::spatial_split               BVHPartitions::partition_spatial    BVHPartitions.h:173-377
//     B3 helpers                        AABB.h:14-52, AABB.cpp:3-36, Triangle.h:14-24
//
// The point of restating it operation for operation (same float expressions in the same order, same strict / non-strict
// comparisons, same iteration order over the three centroid-sorted reference lists, same libstdc++ std::sort) is topology
// parity: traversal order decides which of two equal-t hits wins, so only the reference's own tree gives the reference's
// own frame.  tests/test_sbvh.py compares the node array and the flattened reference order with trees built by the real
// reference (oracle/_ref, MESH_ACCELERATOR_SBVH) bit for bit.
//
// Differences that are deliberate: the reference writes into fixed 2n-entry arrays and overruns them when a mesh splits
// more than that; here the reference lists grow on demand and the caller's capacities are checked (RTX_ERR_LIMIT).
// Built with -ffp-contract=off (see Makefile): a fused multiply-add anywhere in here changes SAH ties.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/rtx_host.h"

namespace {

struct P3 { float v[3]; };
struct B3 { float mn[3], mx[3]; };

inline B3 b3_empty() { B3 b; for (int d = 0; d < 3; d++) { b.mn[d] = INFINITY; b.mx[d] = -INFINITY; } return b; }       // AABB.cpp:3-9
inline void b3_expand(B3 & b, const float * p) {                                                                          // AABB.h:42-45
    for (int d = 0; d < 3; d++) { b.mn[d] = b.mn[d] < p[d] ? b.mn[d] : p[d]; b.mx[d] = b.mx[d] > p[d] ? b.mx[d] : p[d]; }
}
inline void b3_expand(B3 & b, const B3 & o) {                                                                              // AABB.h:47-50
    for (int d = 0; d < 3; d++) { b.mn[d] = b.mn[d] < o.mn[d] ? b.mn[d] : o.mn[d]; b.mx[d] = b.mx[d] > o.mx[d] ? b.mx[d] : o.mx[d]; }
}
inline bool b3_valid(const B3 & b) { return b.mx[0] > b.mn[0] && b.mx[1] > b.mn[1] && b.mx[2] > b.mn[2]; }             // AABB.h:14-16
inline void b3_fix(B3 & b) { for (int d = 0; d < 3; d++) if (b.mx[d] - b.mn[d] < 0.001f) b.mx[d] += 0.005f; }           // AABB.h:26-32
inline float b3_area(const B3 & b) {                                                                                      // AABB.h:34-40 (an empty box gives +inf)
    float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
    return 2.0f * (x * y + y * z + z * x);
}
inline B3 b3_overlap(const B3 & a, const B3 & o) {                                                                         // AABB.cpp:24-35
    B3 r;
    for (int d = 0; d < 3; d++) { r.mn[d] = a.mn[d] > o.mn[d] ? a.mn[d] : o.mn[d]; r.mx[d] = a.mx[d] < o.mx[d] ? a.mx[d] : o.mx[d]; }
    if (!b3_valid(r)) r = b3_empty();
    return r;
}

struct Sbvh {
    int n = 0;
    const float * pos = nullptr;            // n * 9
    std::vector<B3> tri_box;                // Triangle::aabb
    std::vector<P3> centroid;               // Triangle::get_position
    std::vector<int> ref[3];                // the three reference lists, sorted by centroid x / y / z at the root
    std::vector<float> sah;
    std::vector<B3> sweep_l, sweep_r;
    std::vector<unsigned char> to_left, to_right;
    rtx_bvh_node * nodes = nullptr;
    int node_capacity = 0, node_index = 2, ref_capacity = 0;
    float inv_root_area = 0.0f;
    bool overflow = false, degenerate = false;

    const float * vertex(int tri, int k) const { return pos + 9 * (size_t)tri + 3 * k; }
    void reserve_refs(size_t upto) { for (int d = 0; d < 3; d++) if (ref[d].size() < upto) ref[d].resize(upto + upto / 2 + 16); }

    struct ObjectSplit { int index = -1, dim = -1; float cost = INFINITY; B3 left, right; };
    struct SpatialSplit { int dim = -1, n_left = 0, n_right = 0; float cost = INFINITY, plane = NAN; B3 left, right; };

    // partition_object: sweep each sorted list, every prefix / suffix box clipped to the node's box as it grows
    ObjectSplit object_split(int first, int count, const B3 & node_box) {
        ObjectSplit s;
        if ((int)sweep_l.size() < count + 1) { sweep_l.resize(count + 1); sweep_r.resize(count + 1); }
        for (int d = 0; d < 3; d++) {
            const int * list = ref[d].data() + first;
            sweep_l[0] = b3_empty();
            sweep_r[count] = b3_empty();
            for (int i = 1; i < count; i++) {
                B3 b = sweep_l[i - 1];
                b3_expand(b, tri_box[list[i - 1]]);
                sweep_l[i] = b3_overlap(b, node_box);
                sah[i] = b3_area(sweep_l[i]) * float(i);
            }
            for (int i = count - 1; i > 0; i--) {
                B3 b = sweep_r[i + 1];
                b3_expand(b, tri_box[list[i]]);
                sweep_r[i] = b3_overlap(b, node_box);
                float cost = sah[i] + b3_area(sweep_r[i]) * float(count - i);
                if (cost < s.cost) { s.cost = cost; s.index = first + i; s.dim = d; s.left = sweep_l[i]; s.right = sweep_r[i]; }
            }
        }
        return s;
    }

    // partition_spatial: 256 bins per axis over the node box (grown by 0.001 on both sides); every reference is chopped
    // against the bins its box touches, entry / exit counters give the reference counts either side of each bin plane
    SpatialSplit spatial_split(int first, int count, const B3 & bounds) {
        const int BINS = 256;
        SpatialSplit s;
        struct Bin { B3 box; int entries, exits; };
        static thread_local std::vector<Bin> bins(BINS);
        B3 left_of[BINS], right_of[BINS + 1];
        int n_left_of[BINS], n_right_of[BINS + 1];
        float bin_sah[BINS];

        for (int d = 0; d < 3; d++) {
            const float lo = bounds.mn[d] - 0.001f;
            const float hi = bounds.mx[d] + 0.001f;
            const float step = (hi - lo) / BINS;
            const float inv_delta = 1.0f / (hi - lo);
            for (int b = 0; b < BINS; b++) { bins[b].box = b3_empty(); bins[b].entries = 0; bins[b].exits = 0; }

            for (int i = first; i < first + count; i++) {
                const int tri = ref[d][i];
                const B3 & tb = tri_box[tri];
                P3 v[3];
                for (int k = 0; k < 3; k++) memcpy(v[k].v, vertex(tri, k), 12);
                if (v[0].v[d] > v[1].v[d]) std::swap(v[0], v[1]);        // three-element bubble sort along d
                if (v[1].v[d] > v[2].v[d]) std::swap(v[1], v[2]);
                if (v[0].v[d] > v[1].v[d]) std::swap(v[0], v[1]);
                const float v_min = v[0].v[d], v_max = v[2].v[d];

                // int(x) then Math::clamp(.., 0, BINS - 1) (BVHPartitions.h:205-209); out-of-range and NaN values are routed around the
                // float -> int conversion (undefined there, INT_MIN on x86 -> bin 0): same bins, no undefined behaviour
                auto bin_of = [&](float x) { return !(x > -1.0f) ? 0 : (x >= float(BINS) ? BINS - 1 : (int(x) > BINS - 1 ? BINS - 1 : (int(x) < 0 ? 0 : int(x)))); };
                const int b_min = bin_of(BINS * ((tb.mn[d] - lo) * inv_delta));
                const int b_max = bin_of(BINS * ((tb.mx[d] - lo) * inv_delta));
                bins[b_min].entries++;
                bins[b_max].exits++;

                for (int b = b_min; b <= b_max; b++) {
                    const float plane_l = lo + float(b) * step;
                    const float plane_r = plane_l + step;
                    B3 piece;
                    if (v_min >= plane_r || v_max <= plane_l) {
                        continue;                                        // nothing of the triangle inside this slab
                    } else if (v_min >= plane_l && v_max <= plane_r) {
                        piece = tb;                                      // entirely inside
                    } else {
                        P3 cut[4]; int cuts = 0;
                        for (int a = 0; a < 3; a++) {
                            const float va = v[a].v[d];
                            for (int c = a + 1; c < 3; c++) {
                                const float vc = v[c].v[d];
                                const float span = vc - va;
                                if (va < plane_l && plane_l <= vc) {
                                    float t = (plane_l - va) / span;
                                    for (int k = 0; k < 3; k++) cut[cuts].v[k] = (1.0f - t) * v[a].v[k] + t * v[c].v[k];
                                    cuts++;
                                }
                                if (va < plane_r && plane_r <= vc) {
                                    float t = (plane_r - va) / span;
                                    for (int k = 0; k < 3; k++) cut[cuts].v[k] = (1.0f - t) * v[a].v[k] + t * v[c].v[k];
                                    cuts++;
                                }
                            }
                        }
                        piece = b3_empty();                              // AABB::from_points(intersections)
                        for (int k = 0; k < cuts; k++) b3_expand(piece, cut[k].v);
                        b3_fix(piece);
                        if (v[1].v[d] >= plane_l && v[1].v[d] < plane_r) b3_expand(piece, v[1].v);
                        if (cuts == 2) b3_expand(piece, v_max < plane_r ? v[2].v : v[0].v);
                        b3_fix(piece);
                    }
                    b3_expand(bins[b].box, piece);
                    bins[b].box = b3_overlap(bins[b].box, bounds);
                }
            }

            left_of[0] = b3_empty();   right_of[BINS] = b3_empty();
            n_left_of[0] = 0;          n_right_of[BINS] = 0;
            for (int b = 1; b < BINS; b++) {
                left_of[b] = left_of[b - 1];
                b3_expand(left_of[b], bins[b - 1].box);
                n_left_of[b] = n_left_of[b - 1] + bins[b - 1].entries;
                bin_sah[b] = n_left_of[b] < count ? b3_area(left_of[b]) * float(n_left_of[b]) : INFINITY;
            }
            for (int b = BINS - 1; b > 0; b--) {
                right_of[b] = right_of[b + 1];
                b3_expand(right_of[b], bins[b].box);
                n_right_of[b] = n_right_of[b + 1] + bins[b].exits;
                if (n_right_of[b] < count) bin_sah[b] += b3_area(right_of[b]) * float(n_right_of[b]);
                else                       bin_sah[b] = INFINITY;
            }
            for (int b = 1; b < BINS; b++) {
                if (bin_sah[b] < s.cost) {
                    s.cost = bin_sah[b]; s.dim = d;
                    s.left = left_of[b]; s.right = right_of[b];
                    s.n_left = n_left_of[b]; s.n_right = n_right_of[b];
                    s.plane = lo + step * float(b);
                }
            }
        }
        return s;
    }

    // build_sbvh: returns the number of references the subtree stored (its leaves' total), which is where the right
    // sibling's references start
    int build(int node_id, int first, int count, const B3 & box) {
        rtx_bvh_node & node = nodes[node_id];
        for (int d = 0; d < 3; d++) { node.aabb_min[d] = box.mn[d]; node.aabb_max[d] = box.mx[d]; }
        if (count < 3 || overflow) { node.left_or_first = first; node.count = count; return count; }   // (overflow: unwind, the result is discarded)

        if (node_index + 2 > node_capacity) { overflow = true; node.left_or_first = first; node.count = count; return count; }
        const int left = node_index;
        node.left_or_first = left;
        node_index += 2;                                                 // spent even if this node ends up a leaf, as in the reference

        ObjectSplit os = object_split(first, count, box);
        if (os.index < 0) { degenerate = true; node.left_or_first = first; node.count = count; return count; }   // reference: assert, then UB

        SpatialSplit ss;
        float lambda = 0.0f;
        B3 both = b3_overlap(os.left, os.right);
        if (b3_valid(both)) lambda = b3_area(both);
        const float alpha = 10e-5;
        const float ratio = lambda * inv_root_area;
        if (ratio > alpha) ss = spatial_split(first, count, box);

        const float parent_cost = b3_area(box) * float(count);
        if (parent_cost <= os.cost && parent_cost <= ss.cost) { node.left_or_first = first; node.count = count; return count; }

        node.count = (int32_t)((uint32_t)(os.dim + 1) << 30);            // the OBJECT split's axis, also when the spatial split is taken
        std::vector<int> right_refs[3];
        for (int d = 0; d < 3; d++) right_refs[d].reserve(count);
        int n_left = 0;
        B3 box_l, box_r;

        if (os.cost <= ss.cost) {
            const float split = centroid[ref[os.dim][os.index]].v[os.dim];
            for (int d = 0; d < 3; d++) {
                int * list = ref[d].data();
                int l = 0;
                for (int i = first; i < first + count; i++) {
                    const int tri = list[i];
                    const float c = centroid[tri].v[os.dim];
                    bool goes_left = c < split;
                    if (c == split) {                                    // a tie on the split plane: left iff it sits left of the split in the split list
                        int j = os.index - 1;
                        while (j >= first && centroid[ref[os.dim][j]].v[os.dim] == split) {
                            if (ref[os.dim][j] == tri) { goes_left = true; break; }
                            j--;
                        }
                    }
                    if (goes_left) list[first + l++] = tri; else right_refs[d].push_back(tri);
                }
                n_left = l;
            }
            box_l = os.left; box_r = os.right;
        } else {
            float n1 = float(ss.n_left), n2 = float(ss.n_right);
            B3 sl = ss.left, sr = ss.right;
            const int d0 = ss.dim;
            for (int i = first; i < first + count; i++) {
                const int tri = ref[d0][i];
                const float a = vertex(tri, 0)[d0], b = vertex(tri, 1)[d0], c = vertex(tri, 2)[d0];
                bool goes_left  = a <  ss.plane || b <  ss.plane || c <  ss.plane;
                bool goes_right = a >= ss.plane || b >= ss.plane || c >= ss.plane;
                if (goes_left && goes_right) {                           // straddles the plane: keep both, or un-split to one side
                    const B3 & tb = tri_box[tri];
                    bool valid_l = b3_valid(b3_overlap(tb, sl));
                    bool valid_r = b3_valid(b3_overlap(tb, sr));
                    if (valid_l && valid_r) {
                        B3 grown_l = sl, grown_r = sr;
                        b3_expand(grown_l, tb);
                        b3_expand(grown_r, tb);
                        const float area_l = b3_area(sl), area_r = b3_area(sr);
                        const float c_split = area_l * n1 + area_r * n2;
                        const float c_1 = b3_area(grown_l) * n1 + area_r * (n2 - 1.0f);
                        const float c_2 = area_l * (n1 - 1.0f) + b3_area(grown_r) * n2;
                        if (c_1 < c_split) {
                            if (c_2 < c_1) { goes_left = false;  n1 -= 1.0f; b3_expand(sr, tb); }
                            else           { goes_right = false; n2 -= 1.0f; b3_expand(sl, tb); }
                        } else if (c_2 < c_split) {
                            goes_left = false; n1 -= 1.0f; b3_expand(sr, tb);
                        }
                    } else {
                        if (!valid_r) goes_right = false;
                        if (!valid_l) goes_left = false;
                    }
                }
                to_left[tri] = goes_left; to_right[tri] = goes_right;
            }
            for (int d = 0; d < 3; d++) {
                int * list = ref[d].data();
                int l = 0;
                for (int i = first; i < first + count; i++) {
                    const int tri = list[i];
                    const bool gl = to_left[tri], gr = to_right[tri];    // read both before the in-place write below
                    if (gl) list[first + l++] = tri;
                    if (gr) right_refs[d].push_back(tri);
                }
                n_left = l;
            }
            box_l = sl; box_r = sr;
        }
        const int n_right = (int)right_refs[0].size();
        if (n_left <= 0 || n_right <= 0 || (int)right_refs[1].size() != n_right || (int)right_refs[2].size() != n_right) {
            degenerate = true;                                           // the reference asserts these away; NDEBUG builds recurse forever
            node.left_or_first = first; node.count = count;
            return count;
        }

        const int leaves_l = build(left, first, n_left, box_l);
        // heavily overlapping input (e.g. thousands of coplanar triangles) multiplies references without bound; the reference overruns
        // its 2n arrays there.  Stop as soon as the caller's capacity is exceeded instead of building a tree nobody can hold.
        if (overflow || (size_t)first + leaves_l + n_right > (size_t)ref_capacity) { overflow = true; return leaves_l + n_right; }
        reserve_refs((size_t)first + leaves_l + n_right);
        for (int d = 0; d < 3; d++) memcpy(ref[d].data() + first + leaves_l, right_refs[d].data(), sizeof(int) * (size_t)n_right);
        for (int d = 0; d < 3; d++) std::vector<int>().swap(right_refs[d]);
        const int leaves_r = build(left + 1, first + leaves_l, n_right, box_r);
        return leaves_l + leaves_r;
    }
};

}  // namespace

extern "C" int rtxh_blas_build_reference_sbvh(const float * positions, int32_t n, rtx_bvh_node * nodes_out, int32_t node_capacity,
                                              int32_t * node_count_out, int32_t * order_out, int32_t order_capacity, int32_t * order_count_out) {
    if (!positions || n <= 0 || !nodes_out || node_capacity < 2 || !node_count_out || !order_out || order_capacity < n || !order_count_out)
        return RTX_ERR_INVALID_ARG;
    Sbvh s;
    s.n = n; s.pos = positions;
    s.tri_box.resize(n); s.centroid.resize(n);
    for (int i = 0; i < n; i++) {
        const float * p = positions + 9 * (size_t)i;
        B3 b = b3_empty();
        b3_expand(b, p); b3_expand(b, p + 3); b3_expand(b, p + 6);
        b3_fix(b); b3_fix(b);                                            // AABB::from_points fixes, Triangle::calc_aabb fixes again
        s.tri_box[i] = b;
        for (int d = 0; d < 3; d++) s.centroid[i].v[d] = ((p[d] + p[3 + d]) + p[6 + d]) * (1.0f / 3.0f);
    }
    for (int d = 0; d < 3; d++) {
        s.ref[d].resize(2 * (size_t)n);
        for (int i = 0; i < n; i++) s.ref[d][i] = i;
        std::sort(s.ref[d].begin(), s.ref[d].begin() + n, [&](int a, int b) { return s.centroid[a].v[d] < s.centroid[b].v[d]; });
    }
    s.sah.resize(n + 1);
    s.to_left.assign(n, 0); s.to_right.assign(n, 0);
    s.nodes = nodes_out; s.node_capacity = node_capacity; s.node_index = 2; s.ref_capacity = order_capacity;
    memset(nodes_out, 0, sizeof(rtx_bvh_node) * (size_t)node_capacity);

    B3 root = b3_empty();                                                // BVHPartitions::calculate_bounds over the whole mesh
    for (int i = 0; i < n; i++) b3_expand(root, s.tri_box[s.ref[0][i]]);
    b3_fix(root);
    s.inv_root_area = 1.0f / b3_area(root);

    const int refs = s.build(0, 0, n, root);
    if (s.overflow || refs > order_capacity) return RTX_ERR_LIMIT;
    if (s.degenerate) return RTX_ERR_STATE;
    memcpy(order_out, s.ref[0].data(), sizeof(int) * (size_t)refs);
    *node_count_out = s.node_index;
    *order_count_out = refs;
    return RTX_OK;
}
