// rtx_shade.h — the shading half of Raytracer::bounce (Raytracer.cpp:87-400) as two streaming kernels.
//
//   k_shade(level)    one lane per ray of the level: rebuild the RayHit of the closest hit
//                     (triangle_trace / Sphere::trace / Plane::trace accept branches), albedo
//                     (Material::get_albedo -> Texture::sample), one shadow ray + unshadowed light
//                     contribution per light, and the reflected / refracted child rays with their
//                     ray differentials.  Queue appends are wave-compacted: ballot, popcount prefix,
//                     one atomicAdd per wave.
//   k_resolve(level)  folds a level bottom-up in EXACTLY the reference's evaluation order
//                     (result = diffuse*albedo; += Ks*C_refl; TIR / Beer / Schlick blend), so the
//                     colours are bit-identical to the recursion, then hands the colour to the
//                     parent's record or, at level 0, to Window::plot.
#pragma once
#include "rtx_math.h"
#include "rtx_device.h"
#include "rtx_texture.h"
#include "rtx_trace.h"

#ifndef RTX_SHADE_BLOCK
#define RTX_SHADE_BLOCK 256     // 4 waves aggregate their queue appends into ONE global atomic; one wave per SIMD: a workgroup finds room beside other frames'
                                // persistent traversal grids far sooner than a 384-thread one, which needs two 168-VGPR waves on one SIMD (1.39 -> 1.30 ms per frame)
#endif
#ifndef RTX_WAVES_SHADE
#define RTX_WAVES_SHADE 3     // __launch_bounds__ min waves per SIMD for k_shade (168 VGPRs: the kernel needs 169 unconstrained)
#endif

struct HitFull {          // RayHit, RayHit.h:4-36
    v3 point, normal;
    int material_id;
    float u, v, ds_dx, ds_dy, dt_dx, dt_dy;
    v3 dO_dx, dO_dy, dN_dx, dN_dy;
};

// accept branch of BottomLevelBVH::triangle_trace (BottomLevelBVH.cpp:248-308) for the winning triangle
RTX_D void rebuild_triangle_hit(const DevScene & sc, int inst, int tri, float t, float u, float v, const RayFull & wr, HitFull & h) {
    const rtx_instance & I = sc.instances[inst];
    const DevBlas & B = sc.blas[I.blas_id];
    // Mesh::trace, Mesh.cpp:17-31
    RayFull r;
    r.o = xform_pos(I.world_inv, wr.o);
    r.d = xform_dir(I.world_inv, wr.d);
    r.dO_dx = xform_dir(I.world_inv, wr.dO_dx);
    r.dO_dy = xform_dir(I.world_inv, wr.dO_dy);
    r.dD_dx = xform_dir(I.world_inv, wr.dD_dx);
    r.dD_dy = xform_dir(I.world_inv, wr.dD_dy);

    // pointers out of the DevBlas table: global memory, told to the compiler (global_load instead of flat_load, as in rtx_trace.h)
    const float4 e1 = gld(RTX_GPTR(B.tri_hot), RTX_TRI_STRIDE * tri + 1), e2 = gld(RTX_GPTR(B.tri_hot), RTX_TRI_STRIDE * tri + 2);
    const v3 edge_1 = V3(e1.x, e1.y, e1.z), edge_2 = V3(e2.x, e2.y, e2.z);
    rtx_triangle_cold tc;                                        // 64-byte record = 4 x dwordx4
    {
        const rtx_gptr cp = RTX_GPTR(B.tri_cold + tri);
        const float4 c0 = gld(cp, 0), c1 = gld(cp, 1), c2 = gld(cp, 2), c3 = gld(cp, 3);
        tc.tex_coord_0[0] = c0.x; tc.tex_coord_0[1] = c0.y; tc.tex_coord_edge_1[0] = c0.z; tc.tex_coord_edge_1[1] = c0.w;
        tc.tex_coord_edge_2[0] = c1.x; tc.tex_coord_edge_2[1] = c1.y; tc.normal_0[0] = c1.z; tc.normal_0[1] = c1.w;
        tc.normal_0[2] = c2.x; tc.normal_edge_1[0] = c2.y; tc.normal_edge_1[1] = c2.z; tc.normal_edge_1[2] = c2.w;
        tc.normal_edge_2[0] = c3.x; tc.normal_edge_2[1] = c3.y; tc.normal_edge_2[2] = c3.z; tc.material_id = as_i(c3.w);
    }
    const v3 n_edge_1 = v3p(tc.normal_edge_1), n_edge_2 = v3p(tc.normal_edge_2);
    const v3 n = vadd(vadd(v3p(tc.normal_0), vmuls(n_edge_1, u)), vmuls(n_edge_2, v));

    h.point  = xform_pos(I.world, vadd(r.o, vmuls(r.d, t)));
    h.normal = xform_dir(I.world, vnormalize(n));
    h.material_id = B.material_offset + tc.material_id;
    const float te1x = tc.tex_coord_edge_1[0], te1y = tc.tex_coord_edge_1[1];
    const float te2x = tc.tex_coord_edge_2[0], te2y = tc.tex_coord_edge_2[1];
    h.u = (tc.tex_coord_0[0] + u * te1x) + v * te2x;
    h.v = (tc.tex_coord_0[1] + u * te1y) + v * te2y;

    if (sc.diff_enabled) {
        const float one_over_k = 1.0f / vdot(vcross(edge_1, edge_2), r.d);
        const v3 _q = vmadd_s(r.dD_dx, t, r.dO_dx);
        const v3 _r = vmadd_s(r.dD_dy, t, r.dO_dy);
        const v3 c_u = vcross(edge_2, r.d);
        const v3 c_v = vcross(r.d, edge_1);
        const float du_dx = one_over_k * vdot(c_u, _q);
        const float du_dy = one_over_k * vdot(c_u, _r);
        const float dv_dx = one_over_k * vdot(c_v, _q);
        const float dv_dy = one_over_k * vdot(c_v, _r);
        h.dO_dx = vadd(vmuls(edge_1, du_dx), vmuls(edge_2, dv_dx));
        h.dO_dy = vadd(vmuls(edge_1, du_dy), vmuls(edge_2, dv_dy));
        const v3 dn_dx = vadd(vmuls(n_edge_1, du_dx), vmuls(n_edge_2, dv_dx));
        const v3 dn_dy = vadd(vmuls(n_edge_1, du_dy), vmuls(n_edge_2, dv_dy));
        const float n_dot_n = vdot(n, n);
        const float N_denom = (1.0f / sqrtf(n_dot_n)) / n_dot_n;
        h.dN_dx = vmuls(vsub(vmuls(dn_dx, n_dot_n), vmuls(n, vdot(n, dn_dx))), N_denom);
        h.dN_dy = vmuls(vsub(vmuls(dn_dy, n_dot_n), vmuls(n, vdot(n, dn_dy))), N_denom);
        h.ds_dx = du_dx * te1x + dv_dx * te2x;
        h.ds_dy = du_dy * te1x + dv_dy * te2x;
        h.dt_dx = du_dx * te1y + dv_dx * te2y;
        h.dt_dy = du_dy * te1y + dv_dy * te2y;
    }
}

// accept branch of Sphere::trace, Sphere.cpp:44-89
RTX_D void rebuild_sphere_hit(const DevScene & sc, const rtx_sphere & sp, float t, const RayFull & r, HitFull & h) {
    const float ONE_OVER_PI = 0.31830988618f, ONE_OVER_TWO_PI = 0.15915494309f;
    const v3 center = v3p(sp.center);
    const float one_over_r = sp.radius_inv;
    h.point  = vadd(r.o, vmuls(r.d, t));
    h.normal = vmuls(vsub(h.point, center), one_over_r);
    h.material_id = sp.material_id;
    h.u = rtx_atan2f(h.normal.z, h.normal.x) * ONE_OVER_TWO_PI + 0.5f;
    h.v = rtx_acosf(h.normal.y) * ONE_OVER_PI + 0.5f;
    if (sc.diff_enabled) {
        const v3 dPx = vmadd_s(r.dD_dx, t, r.dO_dx);
        const v3 dPy = vmadd_s(r.dD_dy, t, r.dO_dy);
        const float denom = -1.0f / vdot(r.d, h.normal);
        const float dt_dx = vdot(dPx, h.normal) * denom;
        const float dt_dy = vdot(dPy, h.normal) * denom;
        const v3 dP_dx = vmadd_s(r.d, dt_dx, dPx);
        const v3 dP_dy = vmadd_s(r.d, dt_dy, dPy);
        h.dO_dx = dP_dx; h.dO_dy = dP_dy;
        h.dN_dx = vmuls(dP_dx, one_over_r);
        h.dN_dy = vmuls(dP_dy, one_over_r);
        const float ds_denom = ONE_OVER_TWO_PI / (h.normal.x * h.normal.x + h.normal.z * h.normal.z + 1e-8f);
        h.ds_dx = (h.normal.x * h.dN_dx.z - h.normal.z * h.dN_dx.x) * ds_denom;
        h.ds_dy = (h.normal.x * h.dN_dy.z - h.normal.z * h.dN_dy.x) * ds_denom;
        const float dt_denom = -ONE_OVER_PI * (1.0f / sqrtf(1.0f - h.normal.y * h.normal.y + 1e-8f));
        h.dt_dx = h.dN_dx.y * dt_denom;
        h.dt_dy = h.dN_dy.y * dt_denom;
    }
}

// accept branch of Plane::trace, Plane.cpp:27-67
RTX_D void rebuild_plane_hit(const DevScene & sc, const rtx_plane & pl, float t, const RayFull & r, HitFull & h) {
    const v3 ua = v3p(pl.u_axis), va = v3p(pl.v_axis);
    h.point  = vadd(r.o, vmuls(r.d, t));
    h.normal = v3p(pl.normal);
    h.material_id = pl.material_id;
    h.u = vdot(h.point, ua);
    h.v = vdot(h.point, va);
    if (sc.diff_enabled) {
        const v3 dPx = vmadd_s(r.dD_dx, t, r.dO_dx);
        const v3 dPy = vmadd_s(r.dD_dy, t, r.dO_dy);
        const float denom = -1.0f / (vdot(r.d, h.normal) + 1e-8f);
        const float dt_dx = vdot(dPx, h.normal) * denom;
        const float dt_dy = vdot(dPy, h.normal) * denom;
        const v3 dP_dx = vmadd_s(r.d, dt_dx, dPx);
        const v3 dP_dy = vmadd_s(r.d, dt_dy, dPy);
        h.dO_dx = dP_dx; h.dO_dy = dP_dy;
        h.dN_dx = V3(0.0f, 0.0f, 0.0f);
        h.dN_dy = V3(0.0f, 0.0f, 0.0f);
        h.ds_dx = vdot(dP_dx, ua); h.ds_dy = vdot(dP_dy, ua);
        h.dt_dx = vdot(dP_dx, va); h.dt_dy = vdot(dP_dy, va);
    }
}

// Light::calc_lighting, Light.h:12-26
RTX_D v3 light_calc(v3 colour, v3 normal, v3 to_light, v3 to_camera) {
    float intensity = vdot(normal, to_light);
    if (!(intensity > 0.0f)) return V3(0.0f, 0.0f, 0.0f);
    const v3 half_angle = vnormalize(vadd(to_light, to_camera));
    const float specular_factor = vdot(normal, half_angle);
    intensity = intensity + pow2_128(specular_factor);
    return vmuls(colour, intensity);
}

// The unshadowed contribution of light L (point, spot, directional: Raytracer.cpp:155-198 with PointLight.h:9-11, SpotLight.h:17-33,
// DirectionalLight.h:9-11) to a hit: the expressions of the reference in its order, evaluated where the value is needed.
RTX_D v3 light_contribution(const DevScene & sc, const int L, const v3 point, const v3 normal, const v3 to_camera) {
    if (L < sc.point_light_count) {
        const rtx_point_light & pl = sc.point_lights[L];
        v3 to_light = vsub(v3p(pl.position), point);
        const float d2 = vdot(to_light, to_light);
        const float d  = sqrtf(d2);
        to_light = vdivs(to_light, d);
        return vdivs(light_calc(v3p(pl.colour), normal, to_light, to_camera), d2);
    }
    if (L < sc.point_light_count + sc.spot_light_count) {
        const rtx_spot_light & sl = sc.spot_lights[L - sc.point_light_count];
        v3 to_light = vsub(v3p(sl.position), point);
        const float d2 = vdot(to_light, to_light);
        const float d  = sqrtf(d2);
        to_light = vdivs(to_light, d);
        v3 c = V3(0.0f, 0.0f, 0.0f);
        const float dt = vdot(to_light, v3p(sl.negative_direction));
        if (dt > sl.outer_cutoff) {
            float falloff = (dt - sl.outer_cutoff) / (sl.inner_cutoff - sl.outer_cutoff);
            falloff = (falloff > 1.0f) ? 1.0f : falloff;
            c = vmuls(vdivs(light_calc(v3p(sl.colour), normal, to_light, to_camera), d2), falloff);
        }
        return c;
    }
    const rtx_directional_light & dl = sc.dir_lights[L - sc.point_light_count - sc.spot_light_count];
    return light_calc(v3p(dl.colour), normal, v3p(dl.negative_direction), to_camera);
}

// CULL: a light whose unshadowed contribution is exactly (0,0,0) (N.L <= 0, Light.h:17-18, or outside the spot cone,
// SpotLight.h:25-26) cannot change the pixel whether or not it is occluded (diffuse + 0 == diffuse).  With CULL the shadow
// ray is still COUNTED (the reference counts before testing, Raytracer.cpp:163,180,192) but marked so that k_trace<any>
// retires it without traversal.  Off by default; bench.py reports it as a separate figure.

// wave-uniform tallies of a shading pass (SGPRs): ballot popcounts instead of one VGPR counter per statistic; n_texels is per lane (COUNT builds)
struct ShadeTally { uint32_t n_shadow, n_refl, n_refr, n_primary, n_shaded, n_sky, n_trihit, n_texels; };

// The shading of ONE ray per lane: everything Raytracer::bounce does between the trace and the recursion (Raytracer.cpp:104-346) for the
// ray in slot `idx` of `level`, given its closest hit (h0 = distance, u, v, primitive; tri = triangle index, -1 for a clipped pixel).
// Called with wave-uniform control flow by k_shade (one launch per level, slots allocated per workgroup) and, fused, by the closest-hit
// packet kernel right after a packet's walk (rtx_packet.h: no h0 / h1 round trip through HBM, no launch of its own to find room for
// beside other frames' persistent grids).  `valid` = the slot holds a ray; `in_range` = the slot exists (idx < rays of the level).
// alloc2(want_a, want_b, counter, slot_a, slot_b) hands out the child-ray slots; every lane of the wave must call it.
template <bool COUNT, bool CULL, typename Alloc2>
RTX_D void shade_ray(const DevScene & sc, const DevQueues & q, const int level, const uint32_t idx, const bool in_range, bool valid,
                     const float4 h0, const int tri, Alloc2 && alloc2, ShadeTally & T) {
    DevCounters * const ctr = q.counters;
    const int bounces_left = sc.bounces - level;
    const int nL = sc.light_count;
    const int lit_cap = q.level_cap[level];
    const uint32_t slot = (uint32_t)q.level_base[level] + (in_range ? idx : 0u);
    RayFull ray;
    int pixel = 0, parent = -1;
    if (valid && tri == -1) valid = false;                        // a slot without a ray: clipped pixel of an edge tile (level 0), unused tail of a wave's slot chunk (deeper levels)
    if (valid) {
        if (level == 0) {
            int px, py;
            primary_pixel(sc, q, idx, px, py);
            primary_ray(sc, px, py, ray, true);
            pixel = py * sc.width + px;
        } else {
            const float4 r0 = q.r0[slot], r1 = q.r1[slot], r2 = q.r2[slot], r3 = q.r3[slot], r4 = q.r4[slot];
            ray.o = V3(r0.x, r0.y, r0.z); ray.d = V3(r0.w, r1.x, r1.y);
            ray.dO_dx = V3(r4.z, r4.w, r2.x); ray.dO_dy = V3(r2.y, r2.z, r2.w);
            ray.dD_dx = V3(r3.x, r3.y, r3.z); ray.dD_dy = V3(r3.w, r4.x, r4.y);
            pixel = as_i(r1.z); parent = as_i(r1.w);
        }
    }
    if (level == 0) T.n_primary += (uint32_t)__popcll(__ballot(valid));
    const int prim = as_i(h0.w);
    const float t = h0.x;
    const bool is_hit = valid && PRIM_KIND(prim) != PRIM_NONE;

    int flags = 0;
    v3 albedo = V3(0, 0, 0);
    HitFull h; h.material_id = 0;
    h.point = h.normal = h.dO_dx = h.dO_dy = h.dN_dx = h.dN_dy = V3(0, 0, 0);
    h.u = h.v = h.ds_dx = h.ds_dy = h.dt_dx = h.dt_dy = 0.0f;
    v3 Ks = V3(0, 0, 0), Kt = V3(0, 0, 0); float ior = 1.0f;

    if (valid && !is_hit) {                                       // Raytracer.cpp:105-111
        albedo = sky_sample(sc.sky, sc.sky_size, ray.d);
    }
    if (is_hit) {
        flags = NF_HIT;
        const int kind = PRIM_KIND(prim), pi = PRIM_INDEX(prim);
        if (kind == PRIM_TRI) rebuild_triangle_hit(sc, pi, tri, t, h0.y, h0.z, ray, h);
        else if (kind == PRIM_SPHERE) rebuild_sphere_hit(sc, sc.spheres[pi], t, ray, h);
        else rebuild_plane_hit(sc, sc.planes[pi], t, ray, h);

        const rtx_material & m = sc.materials[h.material_id];     // Material::get_albedo, Material.h:16-22
        albedo = v3p(m.diffuse);
        if (m.texture_id >= 0) {
            TexCtx tc; tc.t = &sc.textures[m.texture_id]; tc.fetches = 0;
            const bool de = sc.diff_enabled != 0;
            const v3 tex = texture_sample(tc, sc, h.u, h.v, de ? h.ds_dx : 0.0f, de ? h.ds_dy : 0.0f, de ? h.dt_dx : 0.0f, de ? h.dt_dy : 0.0f);
            albedo = vmul(albedo, tex);
            if (COUNT) T.n_texels += tc.fetches;
        }
        Ks = v3p(m.reflection); Kt = v3p(m.transmittance); ior = m.index_of_refraction;
        if (vdot(albedo, albedo) > 0.0f) flags |= NF_LIT;         // Raytracer.cpp:143-145
    }

    if (COUNT) {
        T.n_sky += (uint32_t)__popcll(__ballot(valid && !is_hit));
        T.n_shaded += (uint32_t)__popcll(__ballot(is_hit));
        T.n_trihit += (uint32_t)__popcll(__ballot(is_hit && PRIM_KIND(prim) == PRIM_TRI));
    }
    // ---- shadow rays: one per light for every lit hit, in the hit's OWN slot of the shadow queue (no compaction: a shadow-ray packet
    // is then the same 64 rays as the packet that found the hits — the same 8x8 pixels at level 0 — and the kernel spends no atomic
    // here; the few lanes without a lit hit are skipped by the shadow-ray kernels, which read NF_LIT from the node record) ----
    const bool lit = (flags & NF_LIT) != 0;
    const int lit_rank = (int)idx;
    if (lit) {
        // the shadow rays themselves are not written: point and normal once per hit; shadow_ray_of_light (rtx_trace.h) derives the ray of each light
        // where it is traced, k_resolve evaluates Light::calc_lighting for the lights that turn out unoccluded (light_contribution below)
        q.sp[slot] = make_float4(h.point.x, h.point.y, h.point.z, 0.0f);
        q.sn[slot] = make_float4(h.normal.x, h.normal.y, h.normal.z, 0.0f);
        if (CULL) {
            const v3 to_camera = vnormalize(vsub(v3p(sc.cam_pos), h.point));     // Raytracer.cpp:152 (camera, not ray origin)
            for (int L = 0; L < nL; L++) {
                const v3 c = light_contribution(sc, L, h.point, h.normal, to_camera);
                const uint32_t s = (uint32_t)q.shadow_base[level] + (uint32_t)L * (uint32_t)lit_cap + (uint32_t)lit_rank;
                q.socc[s] = (c.x == 0.0f && c.y == 0.0f && c.z == 0.0f) ? 1u : 0u;      // 1: contributes exactly nothing whether occluded or not: not traced
            }
        }
    }
    T.n_shadow += (uint32_t)nL * (uint32_t)__popcll(__ballot(lit));           // counted before the occlusion test (:163,180,192)

    // ---- secondary rays (Raytracer.cpp:204-346) ----
    // Which children exist is decided first (cheap), both kinds get their queue slots from ONE block-wide allocation (one global
    // atomic per 384 rays: the counter sustains ~90 atomics/us, three per block iteration kept this kernel at that rate), then each
    // child ray is computed and written, one kind at a time (register pressure).
    float F_r = 0.0f;
    v3 absorption = V3(0, 0, 0);
    const bool deeper = is_hit && bounces_left > 0;
    const bool reflection_mask = deeper && vdot(Ks, Ks) > 0.0f;
    const bool refraction_mask = deeper && vdot(Kt, Kt) > 0.0f;
    bool spawn_refr = false, entering = false;
    float eta = 1.0f, cos_theta = 0.0f, kk = 0.0f, n_1 = 1.0f, n_2 = 1.0f;
    if (reflection_mask) flags |= NF_REFL;
    if (refraction_mask) {
        flags |= NF_REFR;
        const float dot = vdot(ray.d, h.normal);
        entering = dot < 0.0f;
        const float air = 1.0f;
        n_1 = entering ? air : ior;
        n_2 = entering ? ior : air;
        cos_theta = entering ? (0.0f - dot) : dot;
        eta = n_1 / n_2;
        kk = 1.0f - (eta * eta * (1.0f - (cos_theta * cos_theta)));
        if (entering) flags |= NF_ENTER;
        if (kk < 0.0f) flags |= NF_TIR;                                       // :310-314
        else spawn_refr = true;
    }
    T.n_refl += (uint32_t)__popcll(__ballot(reflection_mask));
    T.n_refr += (uint32_t)__popcll(__ballot(spawn_refr));
    uint32_t ci_refl, ci_refr;
    alloc2(reflection_mask, spawn_refr, &ctr->ray_count[level + 1], ci_refl, ci_refr);
    auto write_child = [&](uint32_t ci, const RayFull & cr, uint32_t parent_bits) {
        if (ci >= (uint32_t)q.level_cap[level + 1]) { atomicOr(&ctr->error_flags, ERR_QUEUE_OVERFLOW); return; }
        const uint32_t cs = (uint32_t)q.level_base[level + 1] + ci;
        q.r0[cs] = make_float4(cr.o.x, cr.o.y, cr.o.z, cr.d.x);
        q.r1[cs] = make_float4(cr.d.y, cr.d.z, as_f(pixel), as_f((int)parent_bits));      // pixel >= 0; pixel == -1 marks a slot no ray was written to (RAY_DEAD_PIXEL)
        q.r2[cs] = make_float4(cr.dO_dx.z, cr.dO_dy.x, cr.dO_dy.y, cr.dO_dy.z);
        q.r3[cs] = make_float4(cr.dD_dx.x, cr.dD_dx.y, cr.dD_dx.z, cr.dD_dy.x);
        q.r4[cs] = make_float4(cr.dD_dy.y, cr.dD_dy.z, cr.dO_dx.x, cr.dO_dx.y);
    };
    if (reflection_mask) {
        RayFull rr;
        rr.o = h.point;
        const float dn = vdot(ray.d, h.normal);
        rr.d = vsub(ray.d, vmuls(h.normal, 2.0f * dn));                       // Math::reflect, Math.h:28-30
        rr.dO_dx = h.dO_dx; rr.dO_dy = h.dO_dy;
        const float dDN_dx = vdot(ray.dD_dx, h.normal) + vdot(ray.d, h.dN_dx);
        const float dDN_dy = vdot(ray.dD_dy, h.normal) + vdot(ray.d, h.dN_dy);
        rr.dD_dx = vsub(ray.dD_dx, vmuls(vadd(vmuls(h.dN_dx, vdot(ray.d, h.normal)), vmuls(h.normal, dDN_dx)), 2.0f));
        rr.dD_dy = vsub(ray.dD_dy, vmuls(vadd(vmuls(h.dN_dy, vdot(ray.d, h.normal)), vmuls(h.normal, dDN_dy)), 2.0f));
        write_child(ci_refl, rr, slot);
    }
    if (spawn_refr) {
        RayFull rf;
        const v3 normal = entering ? h.normal : vneg(h.normal);
        rf.o = h.point;
        rf.d = vadd(vmuls(ray.d, eta), vmuls(normal, (eta * cos_theta) - sqrtf(kk)));      // Math::refract, Math.h:34-36
        rf.dO_dx = h.dO_dx; rf.dO_dy = h.dO_dy;
        const float dDN_dx = vdot(ray.dD_dx, h.normal) + vdot(ray.d, h.dN_dx);
        const float dDN_dy = vdot(ray.dD_dy, h.normal) + vdot(ray.d, h.dN_dy);
        const float D_dot_N = -cos_theta;
        const float Dprime_dot_N = -sqrtf(kk);
        const float mu = -(eta * cos_theta + Dprime_dot_N);
        rf.dD_dx = vsub(vmuls(ray.dD_dx, eta), vmuls(vadds(vmul(h.dN_dx, h.normal), mu * D_dot_N), dDN_dx));   // :341-342 literally
        rf.dD_dy = vsub(vmuls(ray.dD_dy, eta), vmuls(vadds(vmul(h.dN_dy, h.normal), mu * D_dot_N), dDN_dy));
        absorption = vsub(Kt, V3(1.0f, 1.0f, 1.0f));                          // :350
        float r_0 = (n_1 - n_2) / (n_1 + n_2);                                // Schlick, :378-391
        r_0 = r_0 * r_0;
        float ct = cos_theta;
        if (n_1 > n_2) ct = 0.0f - vdot(rf.d, normal);
        const float omc  = 1.0f - ct;
        const float omc2 = omc * omc;
        F_r = r_0 + ((1.0f - r_0) * omc2) * (omc2 * omc);
        write_child(ci_refr, rf, slot | 0x80000000u);
    }

    if (!valid && in_range) q.n0[slot] = make_float4(0.0f, 0.0f, 0.0f, as_f(NF_CLIP));      // clipped pixel of an edge tile: no hit, no shadow rays; k_resolve skips it
    if (valid) {
        q.n0[slot] = make_float4(albedo.x, albedo.y, albedo.z, as_f(flags));
        if (flags & NF_HIT) {
            // Ks / F_r / absorption are only read back where the hit reflects or refracts, the hit distance only by a refraction ray's parent (Beer's law):
            // most hits (72 % of the cfg3 frame's) need neither record
            if (flags & (NF_REFL | NF_REFR)) q.n1[slot] = make_float4(Ks.x, Ks.y, Ks.z, F_r);
            if ((flags & NF_REFR) || (level > 0 && ((uint32_t)parent & 0x80000000u))) q.n2[slot] = make_float4(absorption.x, absorption.y, absorption.z, t);
        }
    }
}

template <bool COUNT, bool CULL>
__global__ __launch_bounds__(RTX_SHADE_BLOCK, RTX_WAVES_SHADE)
void k_shade(const DevScene sc, const DevQueues q, const int level) {
    DevCounters * const ctr = q.counters;
    const uint32_t count = ctr->ray_count[level];
    const int lane = threadIdx.x & 63;
    // Block-level compaction: every wave publishes its popcount in LDS, one thread performs the global atomicAdd for the
    // whole workgroup, each lane's slot = block base + counts of the lower waves + its popcount-prefix rank.  (One atomic
    // per WAVE kept k_shade at the ~90 atomics/us a single counter sustains: 32 k waves x 3 counters at level 0.)
    __shared__ uint32_t s_wave_cnt[RTX_SHADE_BLOCK / 64];
    __shared__ uint32_t s_block_base;
    const int wave_in_block = threadIdx.x >> 6;
    // two kinds of slots from ONE allocation: per block [all `a` slots in lane order][all `b` slots in lane order]
    auto block_alloc2 = [&](bool want_a, bool want_b, uint32_t * counter, uint32_t & slot_a, uint32_t & slot_b) {
        const unsigned long long ma = __ballot(want_a), mb = __ballot(want_b);
        if (lane == 0) s_wave_cnt[wave_in_block] = (uint32_t)__popcll(ma) | ((uint32_t)__popcll(mb) << 16);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < RTX_SHADE_BLOCK / 64; w++) tot += (s_wave_cnt[w] & 0xffffu) + (s_wave_cnt[w] >> 16);
            s_block_base = tot ? atomicAdd(counter, tot) : 0u;
        }
        __syncthreads();
        uint32_t off_a = s_block_base, tot_a = 0, off_b = 0;
        for (int w = 0; w < RTX_SHADE_BLOCK / 64; w++) { const uint32_t ca = s_wave_cnt[w] & 0xffffu, cb = s_wave_cnt[w] >> 16; if (w < wave_in_block) { off_a += ca; off_b += cb; } tot_a += ca; }
        slot_a = off_a + (uint32_t)__popcll(ma & ((1ull << lane) - 1ull));
        slot_b = s_block_base + tot_a + off_b + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
        __syncthreads();                          // s_wave_cnt / s_block_base are reused by the next iteration
    };
    ShadeTally T = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (uint32_t bbase = blockIdx.x * RTX_SHADE_BLOCK; bbase < count; bbase += gridDim.x * RTX_SHADE_BLOCK) {   // block-uniform trip count
        const uint32_t idx = bbase + threadIdx.x;
        const bool valid = idx < count;
        float4 h0 = make_float4(0, 0, 0, 0); int tri = 0;
        if (valid) { const uint32_t slot = (uint32_t)q.level_base[level] + idx; h0 = q.h0[slot]; tri = q.h1[slot]; }
        shade_ray<COUNT, CULL>(sc, q, level, idx, valid, valid, h0, tri, block_alloc2, T);
    }
    // statistics: the tallies are per-wave sums; the block adds them up in LDS and ONE thread hands them on.  (One atomic per wave and
    // counter was 20 000 atomics on one cache line at the end of every level-0 launch: at the ~90 atomics/us that line sustains, two
    // thirds of this kernel's time.)  Per-block partial sums go to HBM without any atomic; k_resolve(level 0) adds them into the frame's
    // PerformanceStats.
    uint32_t n_texels = T.n_texels;
    if (COUNT) for (int off = 32; off > 0; off >>= 1) n_texels += __shfl_down(n_texels, off);
    __shared__ uint32_t s_tally[RTX_SHADE_BLOCK / 64][10];
    if (lane == 0) {
        uint32_t * tw = s_tally[wave_in_block];
        tw[0] = T.n_primary; tw[1] = T.n_shadow; tw[2] = T.n_refl; tw[3] = T.n_refr;
        tw[4] = T.n_shaded; tw[5] = T.n_sky; tw[6] = n_texels; tw[7] = T.n_trihit;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        uint32_t tot = 0;
        for (int w = 0; w < RTX_SHADE_BLOCK / 64; w++) tot += s_tally[w][threadIdx.x];
        if (threadIdx.x < 4) q.stats_partial[((size_t)level * q.stats_stride + blockIdx.x) * 4 + threadIdx.x] = tot;      // this launch writes entries 0 .. gridDim.x - 1 = q.stats_n[level]
        else if (COUNT && tot) {
            const int wi[4] = { W_SHADED, W_SKY, W_TEXELS, W_TRI_HITS };
            atomicAdd(&ctr->work[wi[threadIdx.x - 4]], (unsigned long long)tot);
        }
    }
    if (COUNT && threadIdx.x == 0) {
        uint32_t spawned = 0;
        for (int w = 0; w < RTX_SHADE_BLOCK / 64; w++) spawned += s_tally[w][2] + s_tally[w][3];
        if (spawned) atomicAdd(&ctr->work[W_SPAWNED], (unsigned long long)spawned);
    }
}

RTX_D float clampf_ref(float v, float lo, float hi) { if (v < lo) return lo; if (v > hi) return hi; return v; }   // Math::clamp, Math.h:12-18

// Window::plot, Window.h:56-65
RTX_D uint32_t plot_pack(v3 c) {
    const int r = f2i_rn_x86(clampf_ref(c.x * 255.0f, 0.0f, 255.0f) - 0.5f);
    const int g = f2i_rn_x86(clampf_ref(c.y * 255.0f, 0.0f, 255.0f) - 0.5f);
    const int b = f2i_rn_x86(clampf_ref(c.z * 255.0f, 0.0f, 255.0f) - 0.5f);
    return (uint32_t)((r << 16) | (g << 8) | b);
}

// BVH_VISUALIZE_HEATMAP (Raytracer.cpp:97-102): the pixel is the primary ray's BLAS step count scaled per channel; no shading,
// no secondary rays.  k_trace<false, *> left the count in h1 (-1 = pixel clipped off the frame).
__global__ __launch_bounds__(256)
void k_heatmap(const DevScene sc, const DevQueues q) {
    uint32_t n_primary = 0;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < (uint32_t)q.primary_slots; slot += gridDim.x * blockDim.x) {
        int px, py;
        if (!primary_pixel(sc, q, slot, px, py)) continue;
        const float steps = (float)q.h1[(uint32_t)q.level_base[0] + slot];
        const v3 colour = V3(steps * (1.0f / 32.0f), steps * (1.0f / 256.0f), steps * (1.0f / 512.0f));
        const size_t pixel = (size_t)py * sc.width + px;
        q.fb_rgb[3 * pixel + 0] = colour.x; q.fb_rgb[3 * pixel + 1] = colour.y; q.fb_rgb[3 * pixel + 2] = colour.z;
        q.fb_packed[pixel] = plot_pack(colour);
        n_primary++;
    }
    for (int off = 32; off > 0; off >>= 1) n_primary += __shfl_down(n_primary, off);
    if ((threadIdx.x & 63) == 0 && n_primary) atomicAdd(&q.counters->stats[0], (unsigned long long)n_primary);
}

__global__ __launch_bounds__(256)
void k_resolve(const DevScene sc, const DevQueues q, const int level) {
    DevCounters * const ctr = q.counters;
    if (level == 0 && blockIdx.x == 0 && threadIdx.x < 64) {
        // PerformanceStats of this batch: the per-block partial sums of every level's k_shade launch (WorkerThread.cpp:131-148 sums per thread)
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        for (int l = 0; l <= sc.bounces; l++)                          // level l's shading pass wrote q.stats_n[l] entries (k_shade: one per workgroup; fused into the packet kernel: one per wave)
            for (int i = threadIdx.x; i < q.stats_n[l]; i += 64) {
                const uint32_t * p = q.stats_partial + ((size_t)l * q.stats_stride + i) * 4;
                t0 += p[0]; t1 += p[1]; t2 += p[2]; t3 += p[3];
            }
        for (int off = 32; off > 0; off >>= 1) { t0 += __shfl_down(t0, off); t1 += __shfl_down(t1, off); t2 += __shfl_down(t2, off); t3 += __shfl_down(t3, off); }
        if (threadIdx.x == 0) { atomicAdd(&ctr->stats[0], t0); atomicAdd(&ctr->stats[1], t1); atomicAdd(&ctr->stats[2], t2); atomicAdd(&ctr->stats[3], t3); }
    }
    uint32_t count = ctr->ray_count[level];
    if (count > (uint32_t)q.level_cap[level]) count = (uint32_t)q.level_cap[level];
    const int nL = sc.light_count;
    const int lit_cap = q.level_cap[level];
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += gridDim.x * blockDim.x) {
        const uint32_t slot = (uint32_t)q.level_base[level] + idx;
        const float4 n0 = q.n0[slot];
        const int flags = as_i(n0.w);
        if (flags & NF_CLIP) continue;                                // clipped pixel of an edge tile
        // where the colour goes: a primary ray's pixel follows from its slot, a secondary ray's parent travels in its ray record
        uint32_t parent = 0u;
        if (level > 0) parent = (uint32_t)as_i(q.r1[slot].w);
        v3 colour; float distance;
        if (!(flags & NF_HIT)) {
            colour = V3(n0.x, n0.y, n0.z);                             // sky, Raytracer.cpp:106-110
            distance = INFINITY;
        } else {
            float4 n1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n2 = n1;
            if (flags & (NF_REFL | NF_REFR)) n1 = q.n1[slot];
            if ((flags & NF_REFR) || (parent & 0x80000000u)) n2 = q.n2[slot];
            const v3 albedo = V3(n0.x, n0.y, n0.z);
            v3 result = V3(0.0f, 0.0f, 0.0f);
            if (flags & NF_LIT) {                                      // :146-200
                const uint32_t rank = idx;                          // a hit's shadow rays sit in the hit's own slot of every light's segment
                const float4 pp = q.sp[slot], nn = q.sn[slot];
                const v3 point = V3(pp.x, pp.y, pp.z), normal = V3(nn.x, nn.y, nn.z);
                const v3 to_camera = vnormalize(vsub(v3p(sc.cam_pos), point));     // Raytracer.cpp:152
                v3 diffuse = v3p(sc.ambient);
                for (int L = 0; L < nL; L++) {
                    const uint32_t s = (uint32_t)q.shadow_base[level] + (uint32_t)L * (uint32_t)lit_cap + rank;
                    if (q.socc[s]) continue;
                    diffuse = vadd(diffuse, light_contribution(sc, L, point, normal, to_camera));      // :166-168, :183-185, :195-197 for the lights the hit can see
                }
                result = vmadd_v(diffuse, albedo, result);
            }
            v3 colour_reflection = V3(0.0f, 0.0f, 0.0f);
            if (flags & NF_REFL) {                                     // :268-270
                const float4 c0 = q.c0[slot];
                colour_reflection = vmul(V3(n1.x, n1.y, n1.z), V3(c0.x, c0.y, c0.z));
                result = vadd(result, colour_reflection);
            }
            if (flags & NF_REFR) {
                if (flags & NF_TIR) {                                  // :312-314
                    if (flags & NF_REFL) result = vadd(result, colour_reflection);
                } else {
                    const float4 c1 = q.c1[slot];
                    v3 colour_refraction = V3(c1.x, c1.y, c1.z);
                    const float refraction_distance = c1.w;
                    if (flags & NF_ENTER) {                            // Beer, :370-376
                        const float beer_x = rtx_expf(n2.x * refraction_distance);
                        const float beer_y = rtx_expf(n2.y * refraction_distance);
                        const float beer_z = rtx_expf(n2.z * refraction_distance);
                        colour_refraction.x = colour_refraction.x * beer_x;
                        colour_refraction.y = colour_refraction.y * beer_y;
                        colour_refraction.z = colour_refraction.z * beer_z;
                    }
                    const float F_r = n1.w, F_t = 1.0f - F_r;          // :390-395
                    const v3 blend = vadd(vmuls(colour_reflection, F_r), vmuls(colour_refraction, F_t));
                    result = vadd(result, blend);
                }
            }
            colour = result;
            distance = n2.w;
        }
        if (level == 0 && q.tm_packed) {
            q.tm_packed[(uint32_t)q.tm_base + idx] = plot_pack(colour);        // tile-major: slot order, one coalesced store per wave
        } else if (level == 0) {
            int px, py; primary_pixel(sc, q, idx, px, py);
            const int pixel = py * sc.width + px;
            q.fb_rgb[3 * (size_t)pixel + 0] = colour.x;
            q.fb_rgb[3 * (size_t)pixel + 1] = colour.y;
            q.fb_rgb[3 * (size_t)pixel + 2] = colour.z;
            q.fb_packed[pixel] = plot_pack(colour);
        } else {
            const uint32_t ps = parent & 0x7fffffffu;
            if (parent & 0x80000000u) q.c1[ps] = make_float4(colour.x, colour.y, colour.z, distance);
            else                      q.c0[ps] = make_float4(colour.x, colour.y, colour.z, 0.0f);
        }
    }
}

// Root of a GPU group: the gathered tile-major buffers of all ranks (rank r's k-th tile = tile r + k * world, 1024 slots each, slot
// order of primary_pixel) -> raster framebuffer.  One thread per slot; clipped slots of edge tiles are skipped.
__global__ __launch_bounds__(256)
void k_unswizzle(const uint32_t * gathered, uint32_t * fb_packed, int world, int tiles_per_rank, int tile_count, int tile_count_x, int width, int height) {
    const uint32_t total = (uint32_t)world * (uint32_t)tiles_per_rank * 1024u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t rk = i >> 10, l = i & 1023u;
        const int r = (int)(rk / (uint32_t)tiles_per_rank), k = (int)(rk % (uint32_t)tiles_per_rank);
        const int tile = group_tile_of(world, r, k);                      // WorkerThread.cpp:57-61 numbering, tile t belongs to GPU t mod N
        if (tile >= tile_count) continue;
        int px, py; tile_slot_pixel(tile, (int)l, tile_count_x, px, py);
        if (px < width && py < height) fb_packed[(size_t)py * width + px] = gathered[i];
    }
}
