// er_shade.h -- ONE iteration of the bounce loop of renderingKernel (reference src/kernel.cpp:508-593), shared by
// every schedule of this library (er_kernels.hip megakernel, er_wavefront.hip, er_fused.hip) and by the per-bounce
// debug trace (er_debug.hip).  The schedules differ only in WHEN they trace the rays this step asks for; the
// arithmetic -- and with it every bit of the image -- is written once, here.
//
// The step is formulated so that it never waits for a shadow query: a query's outcome only selects which of two
// precomputed contributions is added to the path's radiance (c_vis / c_occ), so the caller traces the shadow ray
// whenever it likes and adds the selected contribution BEFORE it touches `light` again -- the order of the
// additions is that of the reference (emission + HDRI next-event estimate first, then the point-light sample).
//
// Two build-defined extensions live here, both off by default (= reference behaviour, bit for bit):
//
//   ER_FLAG_POINT_LIGHTS  (SURVEY.md 8 a15).  The reference never evaluates point lights: pointLight()
//       (src/kernel.cpp:269-301) has no caller, returns nothing on its lit path, and no command loads a light.
//       This build follows the author's sketch.  Per OPAQUE bounce, with count > 0 lights:
//         * ONE extra RNG draw, taken right after DisneySample's three (so a bounce draws, in order: opacity,
//           HDRI cdf, BRDF r1 r2 r3, light pick);
//         * light k = min(int(count * r), count - 1)      (:280; rnd.next() can return exactly 1.0);
//         * newDir = normalized(light.position - point), dist = |light.position - point|, point = Hit.position
//           (:282-284);
//         * shadow ray from point + newDir * 0.001 (:287); OCCLUDED iff some triangle is hit nearer than the light
//           (:289-291) -- both distances measured from the shadow ray's origin (the sketch measures both from
//           `point`, i.e. the same comparison shifted by the common 0.001 offset; measured from the origin the
//           test is "any hit with |Hit.position - origin| < |light - origin|", which the any-hit traversal can end
//           at the first certain occluder);
//         * pointLightValue = radiance / dist^2 (:295), contribution = value * DisneyEval(newDir) * |newDir.N| /
//           pdf with pdf = count / 2pi (:275, :299-300) -- the sketch's pdf, kept although a uniform pick has
//           probability 1/count: like the other reference quirks it defines the result;
//         * light += reduction * contribution, a separate addition after the HDRI term (reduction = the throughput
//           before this bounce's BRDF weight); an occluded light adds reduction * (0,0,0), as the sketch's early
//           `return Vector3()` would; a light whose DisneyEval is exactly zero (below the horizon) is skipped
//           altogether -- no shadow ray, no addition.
//       A point light is a delta light: BRDF sampling cannot hit it, so it takes MIS weight 1.
//
//   ER_FLAG_MIS.  The reference computes two weights hw, bw (src/kernel.cpp:571-572) and never applies them; the NEE
//       and the BRDF-sampled environment contribution are both counted in full.  Its hw/bw mix the pdfs of two
//       DIFFERENT directions (hdripdf of wihdri, brdfpdf of wibrdf), which is not a partition of unity, so they are
//       not applied as they stand.  With ER_FLAG_MIS this build applies the balance heuristic per direction:
//         * NEE sample wihdri:          weight 1 / (1 + DisneyPdf(wihdri) / p_hdri(wihdri));
//         * BRDF sample wibrdf that leaves the scene (the miss branch of the NEXT iteration, :517-522): weight
//           1 / (1 + p_hdri(texel the direction maps to) / p_brdf), p_brdf = the brdfpdf of the last opaque bounce
//           (camera rays and paths that never bounced opaquely: weight 1).
//         Written as 1 / (1 + ratio) so that the infinite p_hdri of HDRI row 0 (sin(theta) = 0, src/HDRI.cpp:101-107)
//         gives the limits 1 and 0 instead of inf / inf.
//       No extra RNG draws.
//   Parity for both extensions is UNPINNED by construction (the reference defines no result); the oracle mirrors them
//   operation for operation (oracle/er_oracle.cpp) and the GPU tests require bit-equality with it.
#pragma once
#include "er_device.h"

namespace erd {

struct BounceOut {
    bool done;          // the path ends with this step (the ray left the scene, or the bounce limit is reached)
    bool shadow;        // an HDRI shadow query was handed to the sink
    bool lshadow;       // a point-light shadow query was handed to the sink
    bool opaque;        // the opacity test passed (src/kernel.cpp:539)
    Ray next;           // continuation ray (valid unless the ray left the scene)
};

// What a schedule does with the step's by-products is its SINK: three inlined callbacks invoked at the point where the
// values are produced (so nothing stays live across the rest of the step -- the shading kernels are register-bound):
//   sink.hdri_query(sr, self_slot, d_self, c_vis, c_occ)
//       HDRI shadow ray; occluded iff its closest hit is a triangle other than `self_slot`, i.e. iff some other
//       triangle is hit nearer than d_self (the distance at which the ray re-hits the triangle it leaves; inf if it
//       does not).  Then light += occluded ? c_occ : c_vis.
//   sink.light_query(lr, limit, l_vis, l_occ)
//       point-light shadow ray; occluded iff some triangle is hit nearer than `limit`.  Then, AFTER the HDRI term,
//       light += occluded ? l_occ : l_vis.
//   sink.first_hit(n, t, b)
//       first-bounce AOVs (src/kernel.cpp:581-585).
// A sink may trace at once and add to `light` itself (the megakernel does), or record the query and add the selected
// contribution before `light` is touched again (wavefront, fused).

// One bounce-loop iteration for a ray whose closest hit is triangle slot `hslot` (-1: it left the scene).
//   rs         the pixel's RNG state
//   light, reduction, bounce   the path state of src/kernel.cpp:496-506
//   prev_pdf   ER_FLAG_MIS only: brdfpdf of the last opaque bounce, < 0 before the first one
// Contributions that need no shadow query are added to `light` here.
// EXT = false compiles the extensions out (the reference path keeps its register budget); the launch wrappers pick
// EXT = true only when ER_FLAG_POINT_LIGHTS (with at least one light) or ER_FLAG_MIS is set.
template <bool COUNT, bool EXT, class Sink>
ERD void bounce_step(const DevScene& S, const Ray& ray, int hslot, uint32_t& rs, F3& light, F3& reduction, uint32_t& bounce, float& prev_pdf,
                     BounceOut& o, Sink& sink, unsigned& c_shaded, unsigned& c_texels, unsigned& c_hdri) {
    const int hw = S.hdri_tex.width, hh = S.hdri_tex.height;
    const bool mis = EXT && (S.ext_flags & ER_FLAG_MIS) != 0;
    o.done = false; o.shadow = false; o.lshadow = false; o.opaque = false;
    if (hslot < 0) {
        // src/kernel.cpp:517-522
        float u, v;
        spherical_mapping(-1 * ray.d, u, v);
        F3 env = tex_filtered(S, S.hdri_tex, u, v);
        if (COUNT) c_texels++;
        if (mis && prev_pdf >= 0.0f) {
            const float p_h = hdri_pdf(S, ermath::f2i(u * hw), ermath::f2i(v * hh));
            env = env * (1.0f / (1.0f + p_h / prev_pdf));
            if (COUNT) c_texels++;
        }
        light = light + reduction * env;
        o.done = true;
        return;
    }
    c_shaded++;
    HitFull hit;
    full_hit(S, (uint32_t)hslot, ray, hit);
    const ErMaterial& mat = S.materials[hit.material];
    HitData hd;
    generate_hit_data<COUNT>(S, mat, hit, hd, c_texels);
    const int shader = mat.albedo_shader_id;
    if (shader != -1) {   // asl_shade placeholder, src/shader.cpp:6-10, src/shader.h:10-11
        hd.albedo = f3s(0);
        if (shader >= 0 && shader < 4) hd.albedo = f3(1, 1, 0);
    }
    if (rng_next(rs) <= hd.opacity) {
        const F3 wo = ray.d * -1.0f;
        const F3 N = hd.normal;
        o.opaque = true;
        c_hdri++;
        const int count = er_cdf_search(S.hdri_cdf, hw * hh, S.hdri_guide, S.hdri_buckets, rng_next(rs));   // == HDRI::binarySearch
        const float tcx = (float)(count % hw), tcy = (float)(count / hw);
        const float d1 = rng_next(rs), d2 = rng_next(rs), d3 = rng_next(rs);
        const bool lights = EXT && (S.ext_flags & ER_FLAG_POINT_LIGHTS) != 0 && S.light_count > 0;
        float rl = 0.0f;
        if (lights) rl = rng_next(rs);
        const F3 wibrdf = DisneySample(hd, wo, N, d1, d2, d3);
        const float nu = tcx / (float)hw, nv = tcy / (float)hh;
        float iu, iv;
        inverse_transform_uv(S.hdri_tex, nu, nv, iu, iv);
        const F3 wihdri = normalized(reverse_spherical_mapping(iu, iv)) * -1.0f;
        const F3 hdriValue = tex_uv(S, S.hdri_tex, iu, iv);
        if (COUNT) c_texels += 2;
        const F3 evalh = DisneyEval(hd, wo, N, wihdri);
        const float hdripdf = hdri_pdf(S, ermath::f2i(iu * hw), ermath::f2i(iv * hh));
        const float absdot = __builtin_fabsf(dot(wihdri, N));
        // The reference always traces the shadow ray (src/kernel.cpp:555-562); a hit on another triangle zeroes
        // hdriValue.  Both outcomes are computed here with the reference's expression; when the BRDF term is
        // exactly zero they coincide and the query is skipped.
        F3 intv = hdriValue * evalh * absdot / hdripdf;
        F3 into = f3s(0) * evalh * absdot / hdripdf;
        if (mis) {
            const float wnee = 1.0f / (1.0f + DisneyPdf(hd, wo, N, wihdri) / hdripdf);
            intv = intv * wnee;
            into = into * wnee;
        }
        const F3 c_vis = reduction * (hd.emission + intv);
        if (evalh.x != 0.0f || evalh.y != 0.0f || evalh.z != 0.0f) {
            const F3 c_occ = reduction * (hd.emission + into);
            const Ray sr = make_ray(hd.position + N * 0.001f, wihdri);
            F3 v0, v1, v2;
            float4 qa, qb, qc4;
            load_verts(S, (uint32_t)hslot, v0, v1, v2, qa, qb, qc4);
            float su, sv, st, d_self = __builtin_inff();
            if (tri_mt(v0, v1, v2, sr, su, sv, st)) d_self = candidate_distance(S, (uint32_t)hslot, v0, v1, v2, sr, su, sv, st);
            sink.hdri_query(sr, hslot, d_self, c_vis, c_occ);
            o.shadow = true;
        } else {
            light = light + c_vis;
        }
        if (lights) {
            // the author's sketch, src/kernel.cpp:269-301 (see the header of this file)
            int k = ermath::f2i((float)S.light_count * rl);
            k = k > (int)S.light_count - 1 ? (int)S.light_count - 1 : k;
            const ErPointLight L = S.lights[k];
            const F3 lpos = f3(L.position.x, L.position.y, L.position.z);
            const F3 toL = lpos - hd.position;
            const F3 newDir = normalized(toL);
            const float dist = length(toL);
            const Ray lr = make_ray(hd.position + newDir * 0.001f, newDir);
            const float l_limit = length(lpos - lr.o);
            const F3 value = f3(L.radiance.x, L.radiance.y, L.radiance.z) / (dist * dist);
            const F3 evall = DisneyEval(hd, wo, N, newDir);
            const float lpdf = ((float)S.light_count) / (2.0f * PIF);
            const F3 plInt = value * evall * __builtin_fabsf(dot(newDir, N)) / lpdf;
            // a light whose BRDF term is exactly zero (below the shading normal's horizon) is skipped: no ray, no addition
            if (evall.x != 0.0f || evall.y != 0.0f || evall.z != 0.0f) {
                sink.light_query(lr, l_limit, reduction * plInt, reduction * f3s(0));
                o.lshadow = true;
            }
        }
        const float brdfpdf = DisneyPdf(hd, wo, N, wibrdf);
        reduction = reduction * (DisneyEval(hd, wo, N, wibrdf) * __builtin_fabsf(dot(wibrdf, N)) / brdfpdf);
        if (mis) prev_pdf = brdfpdf;
        if (bounce == 0) sink.first_hit(hd.normal, hd.tangent, hd.bitangent);
        o.next = make_ray(hit.position + wibrdf * 0.001f, wibrdf);
    } else {
        o.next = make_ray(hit.position + ray.d * 0.001f, ray.d);
    }
    bounce++;
    if (bounce >= S.max_bounces) o.done = true;
}

// End of a path, src/kernel.cpp:597-645: clamp to [0,10], NaN gate, running mean over `sa` (which starts at 1, so after
// n samples the planes hold sum/(n+1)); the DENOISE plane is never written.  Returns the new sample count.
ERD uint32_t accumulate_sample(const DevScene& S, uint32_t idx, uint32_t sa, F3 light, F3 aov_n, F3 aov_t, F3 aov_b) {
    const size_t npx = (size_t)S.x_res * S.y_res;
    light = f3(clampf(light.x, 0, 10), clampf(light.y, 0, 10), clampf(light.z, 0, 10));
    if (!(light.x != light.x) && !(light.y != light.y) && !(light.z != light.z)) {
        const float k = ((float)sa) / ((float)(sa + 1));
        const float inv = (float)(sa + 1);
        const F3 vals[4] = {light, aov_n, aov_t, aov_b};
        const int planes[4] = {ER_PASS_BEAUTY, ER_PASS_NORMAL, ER_PASS_TANGENT, ER_PASS_BITANGENT};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4* pp = S.passes + (size_t)planes[q] * npx + idx;
            float4 p = *pp;
            if (sa > 0) { p.x *= k; p.y *= k; p.z *= k; }
            p.x += vals[q].x / inv; p.y += vals[q].y / inv; p.z += vals[q].z / inv;
            *pp = p;
        }
        sa++;
    }
    return sa;
}

// host side: does this render need the EXT = true kernels?
static inline bool er_ext_active(const DevScene& S) {
    return (S.ext_flags & ER_FLAG_MIS) != 0 || ((S.ext_flags & ER_FLAG_POINT_LIGHTS) != 0 && S.light_count > 0);
}

}  // namespace erd
