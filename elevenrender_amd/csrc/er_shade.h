// er_shade.h + er_bounce.inc -- ONE iteration of the bounce loop of renderingKernel (reference src/kernel.cpp:508-593),
// shared by every schedule of this library (er_kernels.hip megakernel, er_wavefront.hip, er_fused.hip) and by the
// per-bounce debug trace (er_debug.hip).  The schedules differ only in WHEN they trace the rays this step asks for; the
// arithmetic -- and with it every bit of the image -- is written once: the statement sequence lives in er_bounce.inc,
// which each kernel #includes at the point where it shades (hooks below), this header holds what it calls.
//
// Why textual inclusion and not a function: the same statements behind an always-inlined function with reference
// parameters and a "sink" policy object cost the wavefront shade kernel 16 % and the fused kernel 12 % (measured on one
// box, round 2: 242 vs 209 ms of shade launches per 20 steps) -- identical static instruction mix, ~12 more spill
// stores + fills per 64-slot ticket ON THE HOT PATH, each a dependent memory round trip in a kernel that runs one wave
// per SIMD.  Included as text, the kernels compile to the round-1 code.
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

// host side: does this render need the EXT = true kernels?
static inline bool er_ext_active(const DevScene& S) {
    return (S.ext_flags & ER_FLAG_MIS) != 0 || ((S.ext_flags & ER_FLAG_POINT_LIGHTS) != 0 && S.light_count > 0);
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
            float4* pp = S.passes + er_pass_index(npx, planes[q], idx);      // (the four are neighbours: one 64-byte run)
            float4 p = *pp;
            if (sa > 0) { p.x *= k; p.y *= k; p.z *= k; }
            p.x += vals[q].x / inv; p.y += vals[q].y / inv; p.z += vals[q].z / inv;
            *pp = p;
        }
        sa++;
    }
    return sa;
}

}  // namespace erd
