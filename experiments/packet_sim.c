// packet_sim.c — CPU experiment (not product, not oracle): how many traversal steps does a wave-coherent ("packet") walk of the
// reference BVH need for the cfg3 frame's primary and level-0 shadow rays, compared with the per-lane walk of k_trace_fast?
//
// A packet = the 64 rays of one 8x8 pixel block (the slot order of primary_pixel in csrc/rtx_trace.h); shadow packets are the same
// 64 pixels' hit points towards one light (k_shade's light-major order).  The packet walk keeps ONE stack of (node, 64-bit lane
// mask): a lane is in a node's mask iff every box on the chain passed ITS slab test, which is the reference's per-ray rule.
// Counted per packet: inner nodes visited (child-pair fetches), triangle tests (one per triangle per visiting packet), and the
// lane occupancy of those steps.  Build: gcc -O2 -shared -fPIC -o packet_sim.so packet_sim.c -lm ; driver: packet_sim.py
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float mn[3], mx[3]; int32_t first, count; } Node;
typedef struct { float p0[3], e1[3], e2[3]; } Tri;
#define EPS 0.005f

static inline int slab(const Node * n, const float * o, const float * inv, float maxd) {
    float tn = EPS, tf = maxd;
    for (int a = 0; a < 3; a++) {
        float t0 = (n->mn[a] - o[a]) * inv[a], t1 = (n->mx[a] - o[a]) * inv[a];
        float lo = t0 < t1 ? t0 : t1, hi = t0 > t1 ? t0 : t1;
        if (lo > tn) tn = lo;
        if (hi < tf) tf = hi;
    }
    return tn < tf;
}
static inline int tri_test(const Tri * t, const float * o, const float * d, float maxd, float * tout) {
    float h[3] = { d[1] * t->e2[2] - d[2] * t->e2[1], d[2] * t->e2[0] - d[0] * t->e2[2], d[0] * t->e2[1] - d[1] * t->e2[0] };
    float a = t->e1[0] * h[0] + t->e1[1] * h[1] + t->e1[2] * h[2];
    float f = 1.0f / a;
    float s[3] = { o[0] - t->p0[0], o[1] - t->p0[1], o[2] - t->p0[2] };
    float u = f * (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]);
    if (!(u > 0.0f && u < 1.0f)) return 0;
    float q[3] = { s[1] * t->e1[2] - s[2] * t->e1[1], s[2] * t->e1[0] - s[0] * t->e1[2], s[0] * t->e1[1] - s[1] * t->e1[0] };
    float v = f * (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]);
    if (!(v > 0.0f && u + v < 1.0f)) return 0;
    float tt = f * (t->e2[0] * q[0] + t->e2[1] * q[1] + t->e2[2] * q[2]);
    if (!(tt > EPS && tt < maxd)) return 0;
    *tout = tt;
    return 1;
}

static int g_defer_T = 0;          // hybrid: nodes whose mask has <= T lanes are handed to per-lane walks
static double g_hy_packet_steps, g_hy_lane_iters, g_hy_lane_steps, g_hy_packets, g_hy_max_packet_cost;

typedef struct {
    // per-lane walk (k_trace_fast style: inner-node visits = child-pair fetches)
    double lane_pairs, lane_tris, lane_rays;
    double wave_iters_lockstep;      // sum over packets of max over lanes of (pairs + tris): iterations a wave needs without refill
    // packet walk
    double pk_pairs, pk_tris, pk_packets, pk_pair_lanes, pk_tri_lanes, pk_max_stack;
    double pk_split_packets;         // closest-hit packets whose lanes disagree on a direction sign (must be walked per octant)
} Stats;

// per-lane closest / any walk in the reference's order; returns hit t (INFINITY = none)
static float lane_walk(const Node * nodes, const Tri * tris, const float * o, const float * d, float maxd, int any, Stats * st) {
    float inv[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    int stack[128], sp = 0;
    float best = maxd; int hit = 0;
    st->lane_rays += 1;
    if (!slab(&nodes[0], o, inv, best)) return INFINITY;
    stack[sp++] = 0;
    while (sp) {
        int ni = stack[--sp];
        const Node * n = &nodes[ni];
        if (ni != 0 && !slab(n, o, inv, best)) continue;       // pop-time re-test (free in k_trace_fast: key compare)
        for (;;) {
            int cnt = n->count & 0x3fffffff;
            if (cnt > 0) {
                for (int i = n->first; i < n->first + cnt; i++) {
                    float t; st->lane_tris += 1;
                    if (tri_test(&tris[i], o, d, best, &t)) { best = t; hit = 1; if (any) return t; }
                }
                break;
            }
            st->lane_pairs += 1;
            int axis = (int)((uint32_t)n->count >> 30);
            int left_first = axis == 0 ? 1 : d[axis - 1] > 0.0f;
            int l = n->first, r = n->first + 1;
            int pl = slab(&nodes[l], o, inv, best), pr = slab(&nodes[r], o, inv, best);
            int nn = left_first ? l : r, ff = left_first ? r : l, pn = left_first ? pl : pr, pf = left_first ? pr : pl;
            if (pn) { if (pf) stack[sp++] = ff; n = &nodes[nn]; }
            else if (pf) n = &nodes[ff];
            else break;
        }
    }
    return hit ? best : INFINITY;
}

static int lane_subtree_any(const Node * nodes, const Tri * tris, int root, const float * o, const float * d, float maxd, double * steps) {
    float inv[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    int stack[128], sp = 0; stack[sp++] = root;
    while (sp) {
        const Node * n = &nodes[stack[--sp]];
        for (;;) {
            int cnt = n->count & 0x3fffffff;
            if (cnt > 0) { for (int i = n->first; i < n->first + cnt; i++) { float t; *steps += 1; if (tri_test(&tris[i], o, d, maxd, &t)) return 1; } break; }
            *steps += 1;
            int l = n->first, r = n->first + 1;
            int pl = slab(&nodes[l], o, inv, maxd), pr = slab(&nodes[r], o, inv, maxd);
            if (pl) { if (pr) stack[sp++] = r; n = &nodes[l]; } else if (pr) n = &nodes[r]; else break;
        }
    }
    return 0;
}

// hybrid any-hit walk: packet mode while a node's mask has more than g_defer_T lanes, per-lane walks of the deferred subtrees at the end
static void hybrid_any(const Node * nodes, const Tri * tris, float (*o)[3], float (*d)[3], const float * maxd, uint64_t mask0) {
    float inv[64][3];
    for (int l = 0; l < 64; l++) if (mask0 >> l & 1) for (int a = 0; a < 3; a++) inv[l][a] = 1.0f / d[l][a];
    if (!mask0) return;
    uint64_t alive = mask0, m = 0;
    int stack_n[256]; uint64_t stack_m[256]; int sp = 0;
    static int defer_n[64][512]; int defer_c[64]; memset(defer_c, 0, sizeof(defer_c));
    double psteps = 0;
    for (int l = 0; l < 64; l++) if ((alive >> l & 1) && slab(&nodes[0], o[l], inv[l], maxd[l])) m |= 1ull << l;
    int cur = 0;
    while (m) {
        for (;;) {
            m &= alive; if (!m) break;
            const Node * n = &nodes[cur];
            if (__builtin_popcountll(m) <= g_defer_T) { for (int l = 0; l < 64; l++) if (m >> l & 1) defer_n[l][defer_c[l]++] = cur; break; }
            int cnt = n->count & 0x3fffffff;
            if (cnt > 0) {
                for (int i = n->first; i < n->first + cnt && (m & alive); i++) { psteps += 1; for (int l = 0; l < 64; l++) if ((m & alive) >> l & 1) { float t; if (tri_test(&tris[i], o[l], d[l], maxd[l], &t)) alive &= ~(1ull << l); } }
                break;
            }
            psteps += 1;
            int lc = n->first, rc = n->first + 1; uint64_t ml = 0, mr = 0;
            for (int l = 0; l < 64; l++) if (m >> l & 1) { if (slab(&nodes[lc], o[l], inv[l], maxd[l])) ml |= 1ull << l; if (slab(&nodes[rc], o[l], inv[l], maxd[l])) mr |= 1ull << l; }
            if (ml) { if (mr) { stack_n[sp] = rc; stack_m[sp] = mr; sp++; } cur = lc; m = ml; } else if (mr) { cur = rc; m = mr; } else break;
        }
        m = 0;
        while (sp && !m) { sp--; cur = stack_n[sp]; m = stack_m[sp] & alive; }
    }
    double maxl = 0, suml = 0;
    for (int l = 0; l < 64; l++) if (alive >> l & 1) {
        double st = 0;
        for (int k = defer_c[l] - 1; k >= 0; k--) if (lane_subtree_any(nodes, tris, defer_n[l][k], o[l], d[l], maxd[l], &st)) break;
        suml += st; if (st > maxl) maxl = st;
    }
    g_hy_packet_steps += psteps; g_hy_lane_iters += maxl; g_hy_lane_steps += suml; g_hy_packets += 1;
    double cost = psteps * 800 + maxl * 1100; if (cost > g_hy_max_packet_cost) g_hy_max_packet_cost = cost;
}

// packet walk over lanes in `mask0`; closest: best[] updated per lane; any: occluded lanes leave the packet
static void packet_walk(const Node * nodes, const Tri * tris, float (*o)[3], float (*d)[3], float * best, uint64_t mask0, int any, Stats * st) {
    float inv[64][3];
    int lead = -1;
    for (int l = 0; l < 64; l++) if (mask0 >> l & 1) { if (lead < 0) lead = l; for (int a = 0; a < 3; a++) inv[l][a] = 1.0f / d[l][a]; }
    if (lead < 0) return;
    st->pk_packets += 1;
    uint64_t alive = mask0;
    int stack_n[256]; uint64_t stack_m[256]; int sp = 0, maxsp = 0;
    uint64_t m = 0;
    for (int l = 0; l < 64; l++) if ((alive >> l & 1) && slab(&nodes[0], o[l], inv[l], best[l])) m |= 1ull << l;
    if (!m) return;
    int cur = 0;
    for (;;) {
        // cur has mask m (already slab-tested at push time; closest: re-test lanes at pop time happens below)
        const Node * n = &nodes[cur];
        int cnt = n->count & 0x3fffffff;
        if (cnt > 0) {
            for (int i = n->first; i < n->first + cnt; i++) {
                uint64_t mm = m & alive;
                if (!mm) break;
                st->pk_tris += 1; st->pk_tri_lanes += __builtin_popcountll(mm);
                for (int l = 0; l < 64; l++) if (mm >> l & 1) {
                    float t;
                    if (tri_test(&tris[i], o[l], d[l], best[l], &t)) { best[l] = t; if (any) alive &= ~(1ull << l); }
                }
            }
            m = 0;
        } else {
            st->pk_pairs += 1; st->pk_pair_lanes += __builtin_popcountll(m & alive);
            int axis = (int)((uint32_t)n->count >> 30);
            int left_first = axis == 0 ? 1 : d[lead][axis - 1] > 0.0f;
            int lc = n->first, rc = n->first + 1;
            uint64_t ml = 0, mr = 0;
            for (int l = 0; l < 64; l++) if ((m & alive) >> l & 1) {
                if (slab(&nodes[lc], o[l], inv[l], best[l])) ml |= 1ull << l;
                if (slab(&nodes[rc], o[l], inv[l], best[l])) mr |= 1ull << l;
            }
            int nn = left_first ? lc : rc, ff = left_first ? rc : lc;
            uint64_t mn = left_first ? ml : mr, mf = left_first ? mr : ml;
            if (mn) { if (mf) { stack_n[sp] = ff; stack_m[sp] = mf; sp++; if (sp > maxsp) maxsp = sp; } cur = nn; m = mn; continue; }
            if (mf) { cur = ff; m = mf; continue; }
            m = 0;
        }
        // pop
        for (;;) {
            if (!sp || !alive) { if (maxsp > st->pk_max_stack) st->pk_max_stack = maxsp; return; }
            sp--; cur = stack_n[sp]; uint64_t pm = stack_m[sp] & alive, keep = 0;
            // closest: the reference re-tests the far child at pop time against the then-current closest distance
            for (int l = 0; l < 64; l++) if (pm >> l & 1) { if (any || slab(&nodes[cur], o[l], inv[l], best[l])) keep |= 1ull << l; }
            if (keep) { m = keep; break; }
        }
    }
}

// camera: pos, top-left, x axis, y axis (Camera.h); lights: nl x (kind 0 point/spot position | 1 directional negative direction, xyz)
void packet_sim(const Node * nodes, const Tri * tris, const float * cam, int width, int height, const float * lights, int nl,
                int block_step, Stats * prim, Stats * shad, float * depth_out) {
    memset(prim, 0, sizeof(*prim)); memset(shad, 0, sizeof(*shad));
    for (int by = 0; by < height; by += 8 * block_step) for (int bx = 0; bx < width; bx += 8 * block_step) {
        float o[64][3], d[64][3], best[64], best2[64];
        uint64_t mask = 0;
        for (int l = 0; l < 64; l++) {
            int px = bx + (l & 7), py = by + (l >> 3);
            if (px >= width || py >= height) continue;
            mask |= 1ull << l;
            float v[3];
            for (int a = 0; a < 3; a++) v[a] = cam[3 + a] + (float)px * cam[6 + a] + (float)py * cam[9 + a];
            float il = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            for (int a = 0; a < 3; a++) { o[l][a] = cam[a]; d[l][a] = v[a] * il; }
        }
        // per-lane
        double it_max = 0;
        for (int l = 0; l < 64; l++) if (mask >> l & 1) {
            double b0 = prim->lane_pairs + prim->lane_tris;
            best[l] = lane_walk(nodes, tris, o[l], d[l], INFINITY, 0, prim);
            double it = prim->lane_pairs + prim->lane_tris - b0; if (it > it_max) it_max = it;
        }
        prim->wave_iters_lockstep += it_max;
        // packet, per direction-sign octant
        int octs = 0;
        for (int oc = 0; oc < 8; oc++) {
            uint64_t mo = 0;
            for (int l = 0; l < 64; l++) if (mask >> l & 1) { int c = (d[l][0] > 0) | (d[l][1] > 0) << 1 | (d[l][2] > 0) << 2; if (c == oc) mo |= 1ull << l; }
            if (!mo) continue;
            octs++;
            for (int l = 0; l < 64; l++) best2[l] = INFINITY;
            packet_walk(nodes, tris, o, d, best2, mo, 0, prim);
            for (int l = 0; l < 64; l++) if ((mo >> l & 1) && !(best2[l] == best[l] || (isinf(best2[l]) && isinf(best[l])))) prim->pk_max_stack = -1e9;   // mismatch marker
        }
        if (octs > 1) prim->pk_split_packets += 1;
        if (depth_out) for (int l = 0; l < 64; l++) if (mask >> l & 1) depth_out[(size_t)(by + (l >> 3)) * width + bx + (l & 7)] = best[l];
        // shadow rays of the hit lanes
        for (int L = 0; L < nl; L++) {
            float so[64][3], sd[64][3], smax[64]; uint64_t sm = 0;
            for (int l = 0; l < 64; l++) if ((mask >> l & 1) && !isinf(best[l])) {
                sm |= 1ull << l;
                for (int a = 0; a < 3; a++) so[l][a] = o[l][a] + d[l][a] * best[l];
                if (lights[4 * L] == 0.0f) {
                    float v[3]; for (int a = 0; a < 3; a++) v[a] = lights[4 * L + 1 + a] - so[l][a];
                    float dist = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                    for (int a = 0; a < 3; a++) sd[l][a] = v[a] / dist;
                    smax[l] = dist;
                } else { for (int a = 0; a < 3; a++) sd[l][a] = lights[4 * L + 1 + a]; smax[l] = INFINITY; }
            }
            double itm = 0;
            float occ[64];
            for (int l = 0; l < 64; l++) if (sm >> l & 1) {
                double b0 = shad->lane_pairs + shad->lane_tris;
                occ[l] = lane_walk(nodes, tris, so[l], sd[l], smax[l], 1, shad);
                double it = shad->lane_pairs + shad->lane_tris - b0; if (it > itm) itm = it;
            }
            shad->wave_iters_lockstep += itm;
            float b2[64]; memcpy(b2, smax, sizeof(b2));
            packet_walk(nodes, tris, so, sd, b2, sm, 1, shad);
            hybrid_any(nodes, tris, so, sd, smax, sm);
            for (int l = 0; l < 64; l++) if (sm >> l & 1) { int a = !isinf(occ[l]), b = b2[l] != smax[l]; if (a != b) shad->pk_max_stack = -1e9; }
        }
    }
}

void hybrid_set(int T) { g_defer_T = T; g_hy_packet_steps = g_hy_lane_iters = g_hy_lane_steps = g_hy_packets = g_hy_max_packet_cost = 0; }
void hybrid_get(double * out) { out[0] = g_hy_packet_steps; out[1] = g_hy_lane_iters; out[2] = g_hy_lane_steps; out[3] = g_hy_packets; out[4] = g_hy_max_packet_cost; }

// ---- order_sim: per-ray any-hit walks of the level-0 shadow rays under different child-order policies (which subtree first?) ----------
// policy 0 left first, 1 larger box area first, 2 front to back (split axis sign), 3 back to front, 4 smaller t_near first,
// 5 longer overlap (t_far - t_near) first, 6 larger area x overlap first.   out[policy] = { node-pair steps, triangle tests, rays, occluded }
static inline int slab_t(const Node * n, const float * o, const float * inv, float maxd, float * tn_out, float * tf_out) {
    float tn = EPS, tf = maxd;
    for (int a = 0; a < 3; a++) {
        float t0 = (n->mn[a] - o[a]) * inv[a], t1 = (n->mx[a] - o[a]) * inv[a];
        float lo = t0 < t1 ? t0 : t1, hi = t0 > t1 ? t0 : t1;
        if (lo > tn) tn = lo;
        if (hi < tf) tf = hi;
    }
    *tn_out = tn; *tf_out = tf;
    return tn < tf;
}
static double node_area(const Node * n) { double x = n->mx[0] - n->mn[0], y = n->mx[1] - n->mn[1], z = n->mx[2] - n->mn[2]; return x * y + y * z + z * x; }
static int order_walk(const Node * nodes, const Tri * tris, const float * o, const float * d, float maxd, int policy, double * out) {
    float inv[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    int stack[256], sp = 0;
    float tn, tf;
    if (!slab_t(&nodes[0], o, inv, maxd, &tn, &tf)) return 0;
    stack[sp++] = 0;
    while (sp) {
        const Node * n = &nodes[stack[--sp]];
        for (;;) {
            int cnt = n->count & 0x3fffffff;
            if (cnt > 0) { for (int i = n->first; i < n->first + cnt; i++) { float t; out[1] += 1; if (tri_test(&tris[i], o, d, maxd, &t)) return 1; } break; }
            out[0] += 1;
            int l = n->first, r = n->first + 1;
            float tnl, tfl, tnr, tfr;
            int pl = slab_t(&nodes[l], o, inv, maxd, &tnl, &tfl), pr = slab_t(&nodes[r], o, inv, maxd, &tnr, &tfr);
            int left_first = 1;
            int axis = (int)((uint32_t)n->count >> 30);
            switch (policy) {
                case 1: left_first = node_area(&nodes[l]) >= node_area(&nodes[r]); break;
                case 2: left_first = axis == 0 ? 1 : d[axis - 1] > 0.0f; break;
                case 3: left_first = axis == 0 ? 1 : !(d[axis - 1] > 0.0f); break;
                case 4: left_first = tnl <= tnr; break;
                case 5: left_first = (tfl - tnl) >= (tfr - tnr); break;
                case 6: left_first = node_area(&nodes[l]) * (tfl - tnl) >= node_area(&nodes[r]) * (tfr - tnr); break;
                default: break;
            }
            int nn = left_first ? l : r, ff = left_first ? r : l, pn = left_first ? pl : pr, pf = left_first ? pr : pl;
            if (pn) { if (pf) stack[sp++] = ff; n = &nodes[nn]; } else if (pf) n = &nodes[ff]; else break;
        }
    }
    return 0;
}
void order_sim(const Node * nodes, const Tri * tris, const float * cam, int width, int height, const float * lights, int nl, int block_step, double * out /* 7 x 4 */) {
    memset(out, 0, 7 * 4 * sizeof(double));
    Stats dummy; memset(&dummy, 0, sizeof(dummy));
    for (int by = 0; by < height; by += 8 * block_step) for (int bx = 0; bx < width; bx += 8 * block_step) for (int l = 0; l < 64; l++) {
        int px = bx + (l & 7), py = by + (l >> 3);
        if (px >= width || py >= height) continue;
        float v[3], o[3], d[3];
        for (int a = 0; a < 3; a++) v[a] = cam[3 + a] + (float)px * cam[6 + a] + (float)py * cam[9 + a];
        float il = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int a = 0; a < 3; a++) { o[a] = cam[a]; d[a] = v[a] * il; }
        float best = lane_walk(nodes, tris, o, d, INFINITY, 0, &dummy);
        if (isinf(best)) continue;
        for (int L = 0; L < nl; L++) {
            float so[3], sd[3], smax;
            for (int a = 0; a < 3; a++) so[a] = o[a] + d[a] * best;
            if (lights[4 * L] == 0.0f) {
                float w[3]; for (int a = 0; a < 3; a++) w[a] = lights[4 * L + 1 + a] - so[a];
                float dist = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                for (int a = 0; a < 3; a++) sd[a] = w[a] / dist;
                smax = dist;
            } else { for (int a = 0; a < 3; a++) sd[a] = lights[4 * L + 1 + a]; smax = INFINITY; }
            for (int p = 0; p < 7; p++) { out[4 * p + 2] += 1; out[4 * p + 3] += order_walk(nodes, tris, so, sd, smax, p, &out[4 * p]); }
        }
    }
}
