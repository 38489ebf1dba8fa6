// er_trav.h -- one traversal step of the 8-wide compressed BVH, shared by the wavefront trace kernel
// (er_wavefront.hip) and the streaming schedule's tracer waves (er_stream.hip).
//
// A lane owns one ray at a time and advances it by ONE step per call sequence
//     trav_choose  (pick the step's TRIANGLE part and/or NODE part and their addresses; pop if nothing is pending)
//     trav_fetch   (80 bytes of node + up to 96 bytes of triangle records, whole 16-byte pieces)
//     trav_apply   (Moller-Trumbore on one/two records + interval bookkeeping, then decode + 8 box tests)
// Traversal state (after Ylitie, Karras, Laine 2017): the current NODE GROUP (first-child index + mask of hit
// inner children, stored at bit `slot ^ octant` so the highest set bit is the nearest child) and the current
// TRIANGLE GROUP (first slot + mask).  Only node groups are pushed, at most one per level, so the stack is
// bounded by the tree depth (first WF_LDS_STACK levels in LDS, deeper ones in an HBM spill area).
//
// Nearest hit under the reference's metric m = |Hit.position - origin| (reference src/BVH.cpp:114) without
// fetching normals: for a triangle with lift bound l (er_bvh.h) a Moller-Trumbore hit at parameter t has
// m in [t - l - eps, t + l + eps].  The state keeps U = the smallest upper bound seen and the (at most two)
// candidates whose lower bound is <= U; almost always one survives and it is the reference's winner.  Two
// survivors -> the caller compares their exact metrics (exact_distance); more -> the caller re-traces the ray
// with the exact scalar routine (trace<> in er_device.h).
#pragma once
#include "er_device.h"

#ifndef WF_LDS_STACK
#define WF_LDS_STACK 8      // per-lane stack entries in LDS (deeper levels go to the HBM spill area); 10 and 12 measured on C4 and C2: no difference
#endif

namespace erd {

struct Trav {
    F3 o, d, idir, noi;              // ray; clamped 1/d; -(o * idir)
    float U, limit, lo0, lo1;        // closest: smallest upper bound so far; shadow: exact distance of the self hit
    int s0, s1, skip;                // surviving candidates; slot to ignore (shadow query: the triangle the ray leaves)
    uint32_t ng_base, ng_bits;       // node group: first child index; hit mask (bits 0-7, octant order) | imask << 8
    uint32_t tg_base, tg_mask;       // triangle group: first slot; bits 0-15 still to test | the node's tri_present << 16
    uint32_t oct7;
    int sp;
    bool overflow, shadow;
};

typedef float V2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }

ERD void trav_begin(Trav& T, F3 o, F3 d, bool shadow, int skip, float limit) {
    T.o = o;
    T.d = d;
    // 1/d clamped to +-1e18: a zero (or denormal) component would make the fused plane distances
    // inf - inf = NaN; with 1e18 the ray stays inside its slab for any finite t
    T.idir = f3(clampf(1.0f / d.x, -1e18f, 1e18f), clampf(1.0f / d.y, -1e18f, 1e18f), clampf(1.0f / d.z, -1e18f, 1e18f));
    T.noi = f3(-(o.x * T.idir.x), -(o.y * T.idir.y), -(o.z * T.idir.z));
    T.skip = skip;
    T.limit = limit;
    T.U = limit;
    T.s0 = -1; T.s1 = -1; T.lo0 = 0; T.lo1 = 0;
    T.overflow = false;
    T.shadow = shadow;
    T.sp = 0;
    // a positive direction visits low-coordinate children first: they get the high bits
    T.oct7 = (T.idir.x >= 0.0f ? 1u : 0u) | (T.idir.y >= 0.0f ? 2u : 0u) | (T.idir.z >= 0.0f ? 4u : 0u);
    T.ng_base = 0;
    T.ng_bits = (1u << T.oct7) | (1u << 8);      // the root: slot 0 of a virtual parent, an inner child
    T.tg_base = 0; T.tg_mask = 0;
}

// What one lane does in one step: a NODE part (decode one compressed node, test its eight children), a
// TRIANGLE part (one or two records of the pending triangle group), or both.  The trace kernels are bound by
// vector-ALU issue, and a wave pays for the node block and the triangle block whenever ANY of its lanes needs
// them -- so a lane that has both kinds of work pending does both in the same step instead of taking turns.
struct TravStep {
    bool node, tri, two;
    uint32_t tslot, noff, toff;      // first triangle slot; piece offsets (16-byte units from S.nodes8)
};

// phase 1.  Returns true if the lane takes a step this iteration; false = the ray is complete (nothing pending).
// The triangle part drains the current triangle group; the node part (which replaces that group) runs as soon
// as this step's triangle part empties it.
ERD bool trav_choose(Trav& T, const DevScene& S, uint2* stack, uint2* spill, TravStep& st) {
    st.node = false; st.two = false; st.tslot = 0; st.noff = 0; st.toff = 0;
    st.tri = (T.tg_mask & 0xffffu) != 0;
    if (st.tri) {
        // pending bit i is triangle number popcount(present bits below i) of the group; a second pending
        // triangle rides along when it is the very next record in memory
        const uint32_t m = T.tg_mask & 0xffffu, present = T.tg_mask >> 16;
        const unsigned i = __ffs(m) - 1;
        const uint32_t rest = m & (m - 1u);
        const unsigned j = __ffs(rest | 0x10000u) - 1;
        st.two = rest != 0 && (present & ((1u << j) - 1u) & ~((2u << i) - 1u)) == 0;
        T.tg_mask = (T.tg_mask & 0xffff0000u) | (st.two ? (rest & (rest - 1u)) : rest);
        st.tslot = T.tg_base + __popc(present & ((1u << i) - 1u));
        st.toff = S.tri_base_pieces + st.tslot * 3u;
    }
    if ((T.tg_mask & 0xffffu) == 0) {
        if ((T.ng_bits & 0xffu) == 0 && T.sp > 0) {
            T.sp--;
            // (two typed accesses -- the first levels live in LDS in every schedule -- never a select between an LDS and an HBM
            // pointer: that compiles to flat_load / flat_store, which go down both memory paths and make every later wait a full one)
            uint2 g;
            if (T.sp < WF_LDS_STACK) {
                const unsigned long long w = *(volatile __attribute__((address_space(3))) unsigned long long*)(stack + T.sp * 64);
                g = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
            } else {
                g = spill[(T.sp - WF_LDS_STACK) * 64];
            }
            T.ng_base = g.x;
            T.ng_bits = g.y;
        }
        if ((T.ng_bits & 0xffu) != 0) {
            st.node = true;
            uint32_t nmask = T.ng_bits & 0xffu, imask = (T.ng_bits >> 8) & 0xffu;
            unsigned b = 31 - __clz(nmask);
            nmask &= ~(1u << b);
            unsigned s8 = b ^ T.oct7;
            uint32_t child = T.ng_base + __popc(imask & ((1u << s8) - 1u));
            if (nmask) {                       // siblings still to visit: one stack entry for the whole group
                uint2 g = make_uint2(T.ng_base, nmask | (imask << 8));
                if (T.sp < WF_LDS_STACK) *(volatile __attribute__((address_space(3))) unsigned long long*)(stack + T.sp * 64) = (unsigned long long)g.x | ((unsigned long long)g.y << 32);
                else spill[(T.sp - WF_LDS_STACK) * 64] = g;
                T.sp++;
            }
            T.ng_bits = 0;
            st.noff = child * (uint32_t)ER_NODE8_PIECES;
        }
    }
    return st.tri || st.node;
}

// phase 2: per-lane fetch of whole dwordx4 pieces: five for the node part, three or six for the triangle part.
// The vector-memory pipeline pays per cache access, not per byte, so pieces are never split into narrower
// loads; loads and their wait are ONE asm statement (the compiler treats asm outputs as ready when the
// statement ends); lanes that do not need a piece read one shared address (equal addresses coalesce into a
// single access).  Called by ALL lanes of the wave.
struct TravData {
    float4 n0, n1, n2, n3, n4;       // node: origin+exponents+imask | bases+meta | quantised planes
    float4 a, b4, c, dd, e4, f4;     // triangle records: v0|v1|v2 of the first, then of the second
};

ERD void trav_fetch(const DevScene& S, const TravStep& st, TravData& D) {
    const float4* pn = S.nodes8 + (st.node ? st.noff : 0u);
    const float4* pt = S.nodes8 + (st.tri ? st.toff : 0u);
    const float4* pt2 = (st.tri && st.two) ? pt : S.nodes8;
    asm volatile("global_load_dwordx4 %0, %11, off\n\t"
                 "global_load_dwordx4 %1, %11, off offset:16\n\t"
                 "global_load_dwordx4 %2, %11, off offset:32\n\t"
                 "global_load_dwordx4 %3, %11, off offset:48\n\t"
                 "global_load_dwordx4 %4, %11, off offset:64\n\t"
                 "global_load_dwordx4 %5, %12, off\n\t"
                 "global_load_dwordx4 %6, %12, off offset:16\n\t"
                 "global_load_dwordx4 %7, %12, off offset:32\n\t"
                 "global_load_dwordx4 %8, %13, off offset:48\n\t"
                 "global_load_dwordx4 %9, %13, off offset:64\n\t"
                 "global_load_dwordx4 %10, %13, off offset:80\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(D.n0), "=&v"(D.n1), "=&v"(D.n2), "=&v"(D.n3), "=&v"(D.n4), "=&v"(D.a), "=&v"(D.b4), "=&v"(D.c), "=&v"(D.dd),
                   "=&v"(D.e4), "=&v"(D.f4)
                 : "v"(pn), "v"(pt), "v"(pt2)
                 : "memory");
}

// The same fetch in three statements (streaming schedule, ER_STREAM_SPLIT_WAIT): the triangle pieces are requested FIRST and
// waited for alone -- vector-memory operations return in issue order, so `vmcnt(5)` means "the six triangle loads have landed" --
// and the triangle block runs while the five node pieces are still arriving; then `vmcnt(0)` and the node block.  Every
// destination register is an in/out operand ("+v") of the statement that waits for it: that keeps all four components of a piece
// live from its load to its wait, so the allocator can neither park another value in an unused component nor copy or spill a
// register whose load has not landed (tools/check_split_wait.py reads the device assembly and fails the build check if any
// instruction between a load and its wait names one of its destination registers; the Makefile runs it on every build of er_stream.o and
// a failing build has no object; the marker's "tri=" / "node=" say how many loads of each kind the statement holds).
typedef float F4V __attribute__((ext_vector_type(4)));      // (a native vector: an in/out asm operand cannot be the float4 struct)
struct TravRaw { F4V n0, n1, n2, n3, n4, a, b4, c, dd, e4, f4; };
ERD float4 f4_of(F4V v) { return make_float4(v.x, v.y, v.z, v.w); }
ERD void trav_fetch_issue(const DevScene& S, const TravStep& st, TravRaw& R) {
    const float4* pn = S.nodes8 + (st.node ? st.noff : 0u);
    const float4* pt = S.nodes8 + (st.tri ? st.toff : 0u);
    const float4* pt2 = (st.tri && st.two) ? pt : S.nodes8;
    asm volatile("; ER_SPLIT issue tri=6 node=5\n\t"
                 "global_load_dwordx4 %5, %12, off\n\t"
                 "global_load_dwordx4 %6, %12, off offset:16\n\t"
                 "global_load_dwordx4 %7, %12, off offset:32\n\t"
                 "global_load_dwordx4 %8, %13, off offset:48\n\t"
                 "global_load_dwordx4 %9, %13, off offset:64\n\t"
                 "global_load_dwordx4 %10, %13, off offset:80\n\t"
                 "global_load_dwordx4 %0, %11, off\n\t"
                 "global_load_dwordx4 %1, %11, off offset:16\n\t"
                 "global_load_dwordx4 %2, %11, off offset:32\n\t"
                 "global_load_dwordx4 %3, %11, off offset:48\n\t"
                 "global_load_dwordx4 %4, %11, off offset:64"
                 : "=&v"(R.n0), "=&v"(R.n1), "=&v"(R.n2), "=&v"(R.n3), "=&v"(R.n4), "=&v"(R.a), "=&v"(R.b4), "=&v"(R.c), "=&v"(R.dd),
                   "=&v"(R.e4), "=&v"(R.f4)
                 : "v"(pn), "v"(pt), "v"(pt2)
                 : "memory");
}
ERD void trav_wait_tri(TravRaw& R, TravData& D) {
    asm volatile("; ER_SPLIT wait_tri\n\ts_waitcnt vmcnt(5)"
                 : "+v"(R.a), "+v"(R.b4), "+v"(R.c), "+v"(R.dd), "+v"(R.e4), "+v"(R.f4), "+v"(R.n0), "+v"(R.n1), "+v"(R.n2), "+v"(R.n3), "+v"(R.n4)
                 :
                 : "memory");
    D.a = f4_of(R.a); D.b4 = f4_of(R.b4); D.c = f4_of(R.c); D.dd = f4_of(R.dd); D.e4 = f4_of(R.e4); D.f4 = f4_of(R.f4);
}
ERD void trav_wait_node(TravRaw& R, TravData& D) {
    asm volatile("; ER_SPLIT wait_node\n\ts_waitcnt vmcnt(0)" : "+v"(R.n0), "+v"(R.n1), "+v"(R.n2), "+v"(R.n3), "+v"(R.n4) : : "memory");
    D.n0 = f4_of(R.n0); D.n1 = f4_of(R.n1); D.n2 = f4_of(R.n2); D.n3 = f4_of(R.n3); D.n4 = f4_of(R.n4);
}

// the top of the tree from its LDS copy (streaming schedule): the five pieces of the node at piece offset `noff` replace what the
// global loads brought (those lanes' loads went to the shared dummy address)
ERD void trav_node_from_lds(TravData& D, const float4* s_top, uint32_t noff) {
    const float4* q = s_top + (ER_NODE8_PIECES == 5 ? noff : noff / (uint32_t)ER_NODE8_PIECES * 5u);
    D.n0 = q[0]; D.n1 = q[1]; D.n2 = q[2]; D.n3 = q[3]; D.n4 = q[4];
}

// phase 3, TRIANGLE part.  Returns true when a shadow query found a certain occluder (the ray is then complete).
template <bool COUNT>
ERD bool trav_apply_tri(Trav& T, const DevScene& S, const TravStep& st, const TravData& D, unsigned& c_tris) {
    bool occluded = false;
    ER_MARK("tri_block");
    if (st.tri) {
        const bool two = st.two;
        const uint32_t tslot = st.tslot;
        const float4 a = D.a, b4 = D.b4, c = D.c, dd = D.dd, e4 = D.e4, f4 = D.f4;
        // Both records are tested with straight-line code (rejections folded into one predicate, exactly the
        // comparisons of Tri::hit, reference src/Tri.h:56-77), then the interval bookkeeping runs once per record.
        if (COUNT) c_tris += two ? 2u : 1u;
        // the two records go through Moller-Trumbore side by side as the halves of packed f32 operations
        // (v_pk_mul_f32 / v_pk_add_f32 round each half exactly like the scalar instruction, and the expression
        // order is that of Tri::hit, so t, u, v are the reference's bits)
        const V2 dx = {T.d.x, T.d.x}, dy = {T.d.y, T.d.y}, dz = {T.d.z, T.d.z};
        const V2 ox = {T.o.x, T.o.x}, oy = {T.o.y, T.o.y}, oz = {T.o.z, T.o.z};
        const V2 v0x = {a.x, dd.x}, v0y = {a.y, dd.y}, v0z = {a.z, dd.z};
        const V2 e1x = (V2){b4.x, e4.x} - v0x, e1y = (V2){b4.y, e4.y} - v0y, e1z = (V2){b4.z, e4.z} - v0z;
        const V2 e2x = (V2){c.x, f4.x} - v0x, e2y = (V2){c.y, f4.y} - v0y, e2z = (V2){c.z, f4.z} - v0z;
        const V2 px = dy * e2z - dz * e2y, py = -(dx * e2z - dz * e2x), pz = dx * e2y - dy * e2x;        // cross(d, edge2)
        const V2 det2 = e1x * px + e1y * py + e1z * pz;
        const V2 inv2 = {1.0f / det2.x, 1.0f / det2.y};
        const V2 tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
        const V2 u2 = (tx * px + ty * py + tz * pz) * inv2;
        const V2 qx = ty * e1z - tz * e1y, qy = -(tx * e1z - tz * e1x), qz = tx * e1y - ty * e1x;        // cross(tvec, edge1)
        const V2 v2_ = (dx * qx + dy * qy + dz * qz) * inv2;
        const V2 t2 = (e2x * qx + e2y * qy + e2z * qz) * inv2;
        const V2 uv2 = u2 + v2_;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t slot = tslot + k;
            const float lift = k == 0 ? b4.w : e4.w;
            const float EPSILON = 0.0000001f;
            const float det = k == 0 ? det2.x : det2.y, u = k == 0 ? u2.x : u2.y, v = k == 0 ? v2_.x : v2_.y, t = k == 0 ? t2.x : t2.y;
            const float upv = k == 0 ? uv2.x : uv2.y;
            const bool rejected = (det > -EPSILON && det < EPSILON) || (u < 0 || u > 1) || (v < 0 || upv > 1) || (t < 0);
            const bool valid = !rejected && (k == 0 || two) && (int)slot != T.skip && !occluded;
            const float eps = (S.scene_scale + t) * 4e-6f;
            const float lo = t - lift - eps, hi = t + lift + eps;
            // shadow query: certainly nearer than the self hit -> occluded; inside the interval -> ambiguous
            const bool occl = valid && T.shadow && hi < T.limit;
            const bool amb = valid && T.shadow && !(hi < T.limit) && lo < T.limit;
            // closest query: survives if its lower bound does not exceed the smallest upper bound so far
            const bool cand = valid && !T.shadow && !(lo > T.U);
            T.U = (cand && hi < T.U) ? hi : T.U;
            T.s0 = (cand && T.s0 >= 0 && T.lo0 > T.U) ? -1 : T.s0;
            T.s1 = (cand && T.s1 >= 0 && T.lo1 > T.U) ? -1 : T.s1;
            const bool want = cand || amb;
            const bool ins0 = want && T.s0 < 0;
            const bool ins1 = want && !ins0 && T.s1 < 0;
            T.overflow = T.overflow || (want && !ins0 && !ins1);
            T.s0 = ins0 ? (int)slot : T.s0;
            T.lo0 = ins0 ? lo : T.lo0;
            T.s1 = ins1 ? (int)slot : T.s1;
            T.lo1 = ins1 ? lo : T.lo1;
            occluded = occluded || occl;
        }
    }
    return occluded;
}

// phase 3, NODE part: decode one compressed node, test its eight children, set the node group and the triangle group.
template <bool COUNT>
ERD void trav_apply_node(Trav& T, const DevScene& S, const TravStep& st, const TravData& D, unsigned& c_nodes) {
    ER_MARK("node_block");
    if (st.node) {
        // the pruning bound, after this step's triangles have tightened U
        const float eps_far = (S.scene_scale + (T.U < 3.0e38f ? T.U : 0.0f)) * 4e-6f;
        const float bound = T.U + S.max_lift + eps_far;
        const float4 a = D.n0, b4 = D.n1, c = D.n2, dd = D.n3, e4 = D.n4;
        if (COUNT) c_nodes++;
        const uint32_t ebits = __builtin_bit_cast(uint32_t, a.w);
        const float sx = __builtin_bit_cast(float, (ebits & 0xffu) << 23);
        const float sy = __builtin_bit_cast(float, ((ebits >> 8) & 0xffu) << 23);
        const float sz = __builtin_bit_cast(float, ((ebits >> 16) & 0xffu) << 23);
        const uint32_t imask = ebits >> 24;
        const uint32_t present = __builtin_bit_cast(uint32_t, b4.z);
        const uint32_t qlx[2] = {__builtin_bit_cast(uint32_t, c.x), __builtin_bit_cast(uint32_t, c.y)};
        const uint32_t qly[2] = {__builtin_bit_cast(uint32_t, c.z), __builtin_bit_cast(uint32_t, c.w)};
        const uint32_t qlz[2] = {__builtin_bit_cast(uint32_t, dd.x), __builtin_bit_cast(uint32_t, dd.y)};
        const uint32_t qhx[2] = {__builtin_bit_cast(uint32_t, dd.z), __builtin_bit_cast(uint32_t, dd.w)};
        const uint32_t qhy[2] = {__builtin_bit_cast(uint32_t, e4.x), __builtin_bit_cast(uint32_t, e4.y)};
        const uint32_t qhz[2] = {__builtin_bit_cast(uint32_t, e4.z), __builtin_bit_cast(uint32_t, e4.w)};
        // Slab test of the eight children.  Box tests only gate the traversal, so any conservative evaluation
        // is allowed: the entry/exit planes per axis are picked by the ray's direction sign and each plane
        // distance is ONE fused multiply-add, t = q * (2^e * idir) + (p * idir - o * idir); its rounding error
        // is covered by the absolute box padding of the builder (er_bvh.cpp).  The entry and exit plane of an
        // axis share both coefficients, so they go through the packed pipe as one v_pk_fma_f32.
        const float Ax = sx * T.idir.x, Ay = sy * T.idir.y, Az = sz * T.idir.z;
        const float Bx = __builtin_fmaf(a.x, T.idir.x, T.noi.x), By = __builtin_fmaf(a.y, T.idir.y, T.noi.y), Bz = __builtin_fmaf(a.z, T.idir.z, T.noi.z);
        const V2 A2x = {Ax, Ax}, A2y = {Ay, Ay}, A2z = {Az, Az}, B2x = {Bx, Bx}, B2y = {By, By}, B2z = {Bz, Bz};
        const bool posx = (T.oct7 & 1u) != 0, posy = (T.oct7 & 2u) != 0, posz = (T.oct7 & 4u) != 0;
        const uint32_t nx[2] = {posx ? qlx[0] : qhx[0], posx ? qlx[1] : qhx[1]}, fx[2] = {posx ? qhx[0] : qlx[0], posx ? qhx[1] : qlx[1]};
        const uint32_t ny[2] = {posy ? qly[0] : qhy[0], posy ? qly[1] : qhy[1]}, fy[2] = {posy ? qhy[0] : qly[0], posy ? qhy[1] : qly[1]};
        const uint32_t nz[2] = {posz ? qlz[0] : qhz[0], posz ? qlz[1] : qhz[1]}, fz[2] = {posz ? qhz[0] : qlz[0], posz ? qhz[1] : qlz[1]};
        // one accumulator: bit s = child s hit, bits 8 + 2s and 9 + 2s = its triangle slots; the node's imask and
        // tri_present then keep the inner children and the triangles that exist -- no branch on the child kind
        uint32_t hm = 0;
#pragma unroll
        for (int s8 = 0; s8 < 8; s8++) {
            const int w = s8 >> 2, k = s8 & 3;
            const V2 tx = __builtin_elementwise_fma((V2){ubyte_f(nx[w], k), ubyte_f(fx[w], k)}, A2x, B2x);
            const V2 ty = __builtin_elementwise_fma((V2){ubyte_f(ny[w], k), ubyte_f(fy[w], k)}, A2y, B2y);
            const V2 tz = __builtin_elementwise_fma((V2){ubyte_f(nz[w], k), ubyte_f(fz[w], k)}, A2z, B2z);
            const float tmin = __builtin_fmaxf(__builtin_fmaxf(tx.x, ty.x), tz.x);
            const float tmax = __builtin_fminf(__builtin_fminf(tx.y, ty.y), tz.y);
            const bool hit = (tmin <= tmax) && (tmax >= 0.0f) && (tmin <= bound);
            hm |= hit ? ((1u << s8) | (3u << (8 + 2 * s8))) : 0u;
        }
        // inner hits, moved from bit `slot` to bit `slot ^ oct7` (three conditional swap stages)
        uint32_t nmask = hm & imask;
        nmask = (T.oct7 & 1u) ? (((nmask & 0xAAu) >> 1) | ((nmask & 0x55u) << 1)) : nmask;
        nmask = (T.oct7 & 2u) ? (((nmask & 0xCCu) >> 2) | ((nmask & 0x33u) << 2)) : nmask;
        nmask = (T.oct7 & 4u) ? (((nmask & 0xF0u) >> 4) | ((nmask & 0x0Fu) << 4)) : nmask;
        T.ng_base = __builtin_bit_cast(uint32_t, b4.x);
        T.ng_bits = nmask | (imask << 8);
        T.tg_base = __builtin_bit_cast(uint32_t, b4.y);
        T.tg_mask = ((hm >> 8) & present) | (present << 16);
    }
    ER_MARK("apply_end");
}

// phase 3 of a lane that may do both parts in one step (wavefront trace kernel, debug hooks): the triangle part,
// then -- unless it ended the query -- the node part.
template <bool COUNT>
ERD bool trav_apply(Trav& T, const DevScene& S, const TravStep& st, const TravData& D, unsigned& c_nodes, unsigned& c_tris) {
    const bool occluded = trav_apply_tri<COUNT>(T, S, st, D, c_tris);
    TravStep sn = st;
    sn.node = st.node && !occluded;
    trav_apply_node<COUNT>(T, S, sn, D, c_nodes);
    return occluded;
}

// exact reference metric |Hit.position - origin| of triangle `tslot` for `ray` (inf if the ray misses it)
ERD float exact_distance(const DevScene& S, uint32_t tslot, const Ray& ray) {
    F3 v0, v1, v2;
    float4 qa, qb, qc;
    load_verts(S, tslot, v0, v1, v2, qa, qb, qc);
    float u, v, t;
    if (!tri_mt(v0, v1, v2, ray, u, v, t)) return __builtin_inff();
    return candidate_distance(S, tslot, v0, v1, v2, ray, u, v, t);
}

// The exact scalar re-trace is the rare fallback (more than two candidates inside one t-interval: a handful of rays per
// million on the test scenes).
// (Tried out of line, `__attribute__((noinline))`: 40 fewer spilled registers in the shading kernels, but the call takes
// the DevScene by reference, which pins the kernel-argument struct in scratch memory and turns nearly every load of the
// kernel into a flat_load -- no faster.  Inline.)
template <bool COUNT, bool ANY>
ERD int trace_cold(const DevScene& S, int* stack, const Ray& ray, int skip_slot, float limit, unsigned& node_visits,
                                                     unsigned& tri_tests) {
    float dd;
    return trace<COUNT, ANY>(S, stack, ray, skip_slot, limit, dd, node_visits, tri_tests);
}

// closest-hit result of a finished traversal: the winning slot (or -1) under the reference's exact metric
template <bool COUNT>
ERD int resolve_closest(const DevScene& S, int* stack2, const Ray& ray, int hslot, int h2, unsigned& c_nodes, unsigned& c_tris) {
#ifdef ER_EXPERIMENT_NO_COLD
    return hslot;      // (diagnostic: what would the step cost without the inlined exact resolves?  WRONG results on ties)
#endif
    if (h2 == -2) {          // more than two candidates inside one t-interval: exact scalar traversal
        return trace_cold<COUNT, false>(S, stack2, ray, -1, __builtin_inff(), c_nodes, c_tris);
    }
    if (h2 >= 0) {           // two candidates: the reference's strict '<' on the exact metric
        if (exact_distance(S, (uint32_t)h2, ray) < exact_distance(S, (uint32_t)hslot, ray)) return h2;
    }
    return hslot;
}

// shadow query result: occ = 0 / 1 decided by the traversal, 2 = decide among ca/cb by exact metric, 3 = re-trace
template <bool COUNT>
ERD bool resolve_shadow(const DevScene& S, int* stack2, const Ray& sr, int self_slot, float d_self, int occ, int ca, int cb,
                        unsigned& c_nodes, unsigned& c_tris) {
#ifdef ER_EXPERIMENT_NO_COLD
    return occ != 0;
#endif
    if (occ == 3) return trace_cold<COUNT, true>(S, stack2, sr, self_slot, d_self, c_nodes, c_tris) >= 0;
    if (occ == 2) {
        bool nearer = exact_distance(S, (uint32_t)ca, sr) < d_self;
        if (cb >= 0) nearer = nearer || (exact_distance(S, (uint32_t)cb, sr) < d_self);
        return nearer;
    }
    return occ != 0;
}

// ---- a whole query for ONE lane (debug hooks, er_debug.hip).  The production kernels interleave the steps of many
// rays (er_wf_trace and the streaming tracer waves refill lanes one by one); the steps themselves are these. ----
template <bool COUNT>
ERD int trav_run_closest(const DevScene& S, uint2* stack, uint2* spill, int* stack2, const Ray& ray, float limit, int& info,
                         unsigned& c_nodes, unsigned& c_tris) {
    info = 0;
    if (S.node_count == 0) return -1;
    Trav T;
    trav_begin(T, ray.o, ray.d, false, -1, limit);
    TravStep st;
    while (trav_choose(T, S, stack, spill, st)) {
        TravData D;
        trav_fetch(S, st, D);
        trav_apply<COUNT>(T, S, st, D, c_nodes, c_tris);
    }
    const int h2 = T.overflow ? -2 : ((T.s0 >= 0 && T.s1 >= 0) ? T.s1 : -1);
    info = T.overflow ? 2 : (h2 >= 0 ? 1 : 0);     // 0 one survivor (or none), 1 two survivors -> exact metric, 2 -> exact re-trace
    return resolve_closest<COUNT>(S, stack2, ray, T.s0 >= 0 ? T.s0 : T.s1, h2, c_nodes, c_tris);
}
template <bool COUNT>
ERD bool trav_run_shadow(const DevScene& S, uint2* stack, uint2* spill, int* stack2, const Ray& ray, int self_slot, float limit, int& info,
                         unsigned& c_nodes, unsigned& c_tris) {
    info = 0;
    if (S.node_count == 0) return false;
    Trav T;
    trav_begin(T, ray.o, ray.d, true, self_slot, limit);
    TravStep st;
    int code = -1;
    while (trav_choose(T, S, stack, spill, st)) {
        TravData D;
        trav_fetch(S, st, D);
        if (trav_apply<COUNT>(T, S, st, D, c_nodes, c_tris)) { code = 1; break; }
    }
    if (code < 0) code = T.overflow ? 3 : (T.s0 >= 0 ? 2 : 0);
    info = code;                                   // 0 / 1 decided by the t-intervals, 2 exact metric of <= 2 candidates, 3 exact re-trace
    return resolve_shadow<COUNT>(S, stack2, ray, self_slot, limit, code, T.s0, T.s1, c_nodes, c_tris);
}

}  // namespace erd
