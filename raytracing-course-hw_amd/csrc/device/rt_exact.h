// Reference-exact box decisions, shared by the persistent pipeline (rt_persistent.h) and the round pipeline (rt_wavefront.h).
//
// The walkers prune with a cheap conservative test on padded boxes, so they find a superset of the triangles the reference's own
// slab test (hw8/src/primitives.cpp:29-53,163-165: six IEEE divisions on the re-centred, unpadded box) lets through, and return the
// closest of them.  That hit stands as the reference's answer (pt_hit_stands) when
//   (a) every box above it passes the reference's test whatever the rounding (pt_box_robust: margins of the hit point against the
//       triangle's own box; boxes only grow towards the root), and
//   (b) the reference's pruning (bvh.h:118, curBest < t_box) cannot have hidden it behind an earlier hit: the runner-up of the walk
//       lies beyond the hit's window (see pt_hit_stands; the reference's triangle test reports hits outside the triangle's box, and
//       a hit in a face of a flat box is in the same position within rounding);
// the rare others (3e-5 of the queries on the benchmark scene) are walked again with the reference's arithmetic over the reference's
// own trees (ref_closest_hit, ref_light_pdf_sum).  For (b) the walkers must SEE the runner-up whatever tree they walk: look-behind
// (pt_look_behind_abs, SceneView::cull_k), absolute box padding and reference leaf boxes in host/scene_prep.cpp.  hw6 uses the same gate
// (rt_persistent_hw6.h).
#pragma once
#include "rt_kernels_hw8.h"

namespace rtamd {
namespace dev {

#define PT_SHADE_EXACT 4              // action code of pt_shade_item: the hit needs the exact walk before it is shaded
// hit word of a path record: 0xFFFFFFFF = miss, else figure index | flags; packed word: see rt_wavefront.h
#define WF_MISS 0xFFFFFFFFu
#define WF_INSIDE_BIT 0x40000000u
#define WF_GAP_SHIFT 24               // bits 24..29: how far behind the hit the runner-up lies, as a power of two of t (pt_gap_code); bit 31 stays 0 on a hit
#define WF_INDEX_MASK 0x00FFFFFFu     // figure index: the hw8 / hw7 path takes scenes of up to 2^24 - 1 triangles
#define WF_SAMPLE_MASK 0x01FFFFFFu    // 25 bits of sample index in the packed word
#define WF_VERIFIED_BIT 0x80000000u   // packed word: the hit in q2 comes from the reference-exact walk

// AABB::intersect -> intersectBoxAndRay(0.5 * (max - min), ray - 0.5 * (min + max), false), primitives.cpp:163-165,29-53.
RT_DEV bool ref_box_test(F3 mn, F3 mx, F3 o, F3 d, float &t, bool &inside) {
    const F3 s = 0.5f * (mx - mn);
    const F3 oc = o - 0.5f * (mn + mx);
    const F3 a = neg(s) - oc, b = s - oc;
    const float a1x = a.x / d.x, a1y = a.y / d.y, a1z = a.z / d.z;
    const float a2x = b.x / d.x, a2y = b.y / d.y, a2z = b.z / d.z;
    const float t1x = smin(a1x, a2x), t2x = smax(a1x, a2x);
    const float t1y = smin(a1y, a2y), t2y = smax(a1y, a2y);
    const float t1z = smin(a1z, a2z), t2z = smax(a1z, a2z);
    const float t1 = smax(smax(t1x, t1y), t1z);
    const float t2 = smin(smin(t2x, t2y), t2z);
    if (t1 > t2 || t2 < 0) return false;
    if (t1 < 0) { inside = true; t = t2; }
    else { inside = false; t = t1; }
    return true;
}

struct RefNodeView { F3 mn, mx; uint32_t left, right, first, last; };
RT_DEV RefNodeView load_ref_node(const GpuRefNode *p) {
    const float4 *q = reinterpret_cast<const float4 *>(p);
    const float4 a = q[0], b = q[1], c = q[2];
    RefNodeView n;
    n.mn = f3(a.x, a.y, a.z); n.left = __float_as_uint(a.w);
    n.mx = f3(b.x, b.y, b.z); n.right = __float_as_uint(b.w);
    n.first = __float_as_uint(c.x); n.last = __float_as_uint(c.y);
    return n;
}

// BVH::intersect_ (bvh.h:111-142) as an iterative depth-first walk, left child first: the recursion's `curBest` is the running
// best of all hits found so far, a leaf keeps its first triangle on equal t, and a later subtree replaces the best only when
// strictly closer — so one running best with strict '<' reproduces the result.  `stack` holds up to RT_STACK_SIZE node indices.
// STRIDE: distance in words between consecutive stack entries (1 = a private array; the persistent kernel keeps the stacks of its exact
// role in LDS, interleaved over the lanes of the batch).
template <int STRIDE = 1>
RT_DEV void ref_closest_hit(const SceneView &S, F3 o, F3 d, uint32_t *stack, float &best_t, float &best_u, float &best_v, uint32_t &hit) {
    best_t = RT_T_MAX; best_u = 0.f; best_v = 0.f; hit = WF_MISS;
    if (S.n_tris == 0) return;
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        const RefNodeView n = load_ref_node(S.ref_nodes + cur);
        float tb; bool inside;
        if (ref_box_test(n.mn, n.mx, o, d, tb, inside) && !(hit != WF_MISS && best_t < tb && !inside)) {
            if (n.left == 0) {
                for (uint32_t i = n.first; i < n.last; i++) {
                    const TriIsect T = load_isect(S.tri_isect + i);
                    float t, u, v; bool in;
                    if (tri_test(T, o, d, t, u, v, in) && (hit == WF_MISS || t < best_t)) {
                        best_t = t; best_u = u; best_v = v; hit = i | (in ? WF_INSIDE_BIT : 0u);
                    }
                }
            } else if (sp < RT_STACK_SIZE) { stack[STRIDE * sp++] = n.right; cur = n.left; continue; }
        }
        if (sp == 0) break;
        cur = stack[STRIDE * --sp];
    }
}

// FiguresMix::getTotalPdf (distributions.h:148-165) over the reference light tree with the reference's box test and its
// association of the additions (TODO / ADD frames as in light_pdf_sum, rt_device.h).
template <int STRIDE = 1>
RT_DEV float ref_light_pdf_sum(const SceneView &S, F3 x, F3 d, uint32_t *stack) {
    int sp = 0;
    unsigned long long addmask = 0;
    uint32_t cur = 0;
    bool descending = true;
    float v = 0.f;
    for (;;) {
        if (descending) {
            const RefNodeView n = load_ref_node(S.ref_light_nodes + cur);
            float tb; bool inside;
            if (!ref_box_test(n.mn, n.mx, x, d, tb, inside)) { v = 0.f; descending = false; }
            else if (n.left == 0) {
                float result = 0.f;
                for (uint32_t i = n.first; i < n.last; i++) {
                    bool last;
                    result += light_pdf_one(S.lights + i, x, d, last, S.hw7 != 0);
                }
                v = result;
                descending = false;
            } else if (sp < RT_STACK_SIZE) { addmask &= ~(1ull << sp); stack[STRIDE * sp++] = n.right; cur = n.left; }
            else { v = 0.f; descending = false; } // deeper than the host admits (checked there)
        } else {
            if (sp == 0) break;
            --sp;
            const uint32_t f = stack[STRIDE * sp];
            if ((addmask >> sp) & 1ull) v = __uint_as_float(f) + v;                  // left total + right total
            else { addmask |= 1ull << sp; stack[STRIDE * sp++] = __float_as_uint(v); cur = f; descending = true; }
        }
    }
    return v;
}

// The gate's own arithmetic is not replay arithmetic: it only has to err on the cautious side, and its thresholds carry 4x the rounding
// bound they guard.  So its reciprocals are the hardware's v_rcp_f32 (1 ulp) instead of IEEE divisions (~10 instructions each): the
// 2^-23 relative difference disappears in those margins.  Where a result is used as a floor or a ceiling it is nudged the safe way.
RT_DEV float pt_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Is the point P = o + t d, hit on a triangle whose box is [lo, hi], robustly inside that box — so robustly that the reference's
// slab test accepts this box and every box containing it whatever the rounding?  The reference computes per axis the slab
// interval [ta_j, tb_j] in t (monotonic: never inverted on one axis) and accepts when max_j ta_j <= min_k tb_k and that
// minimum is >= 0.  With a_j = t - ta_j, b_k = tb_k - t (exact, >= 0 for a point inside) the computed values are off by at
// most ~2^-23 |t-ish| + 2^-24 |o - centre|_j / |d_j| each, so for every pair of different axes
//     a_j + b_k >= c1 t + c2 (1/|d_j| + 1/|d_k|),   c1 = 2^-19,  c2 = 2^-20 max|coordinate|   (4x the bound)
// is sufficient, and t + b_k >= c1 t keeps the exit in front of the origin.  Boxes only grow towards the root, which only
// increases a_j and b_k.  A flat box (a_j = b_j = 0 on its axis) passes as long as the other axes have room.
// The common case of both tests below, for a dozen instructions: the hit point lies deep inside the box — at least c2 + 2^-19 t max|d_k| from every
// face.  Then every a_j, b_k below is >= (m - c2) / max|d_k| >= 2^-19 t, so a_j + b_k >= 2 need, the exit lies beyond t, and the window of
// pt_hit_stands (need - min a) is <= 0.  Flat boxes (axis-aligned triangles) and hits near a box face take the full test.  NaN: false.
RT_DEV bool pt_deep_inside(F3 lo, F3 hi, F3 P, F3 d, float t, float c2) {
    const float m = fminf(fminf(fminf(P.x - lo.x, hi.x - P.x), fminf(P.y - lo.y, hi.y - P.y)), fminf(P.z - lo.z, hi.z - P.z));
    const float dmax = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    return m - c2 >= 1.9073486328125e-06f * t * dmax;
}

template <bool PRETEST = true>
RT_DEV bool pt_box_robust(F3 lo, F3 hi, F3 P, F3 d, float t, float c2) {
    if (PRETEST && pt_deep_inside(lo, hi, P, d, t, c2)) return true;
    const float ix = pt_rcp(fmaxf(fabsf(d.x), 1e-30f)), iy = pt_rcp(fmaxf(fabsf(d.y), 1e-30f)), iz = pt_rcp(fmaxf(fabsf(d.z), 1e-30f));
    const float inx = d.x > 0 ? P.x - lo.x : hi.x - P.x, outx = d.x > 0 ? hi.x - P.x : P.x - lo.x;
    const float iny = d.y > 0 ? P.y - lo.y : hi.y - P.y, outy = d.y > 0 ? hi.y - P.y : P.y - lo.y;
    const float inz = d.z > 0 ? P.z - lo.z : hi.z - P.z, outz = d.z > 0 ? hi.z - P.z : P.z - lo.z;
    const float ax = (inx - c2) * ix, ay = (iny - c2) * iy, az = (inz - c2) * iz;
    const float bx = (outx - c2) * ix, by = (outy - c2) * iy, bz = (outz - c2) * iz;
    const float need = 1.9073486328125e-06f * t;
    const float worst = fminf(fminf(fminf(ax + by, ax + bz), fminf(ay + bx, ay + bz)), fminf(az + bx, az + by));
    const float exit_ = t + fminf(fminf(bx, by), bz);
    return worst >= need && exit_ >= need; // NaN compares false: not robust
}

// The reference prunes a subtree whose box the ray enters behind the best hit so far (bvh.h:118: curBest < t_box).  That is harmless
// for a hit that lies inside its triangle's box -- but the reference's triangle test solves only two projected equations
// (primitives.cpp:85-104), and for a triangle whose plane is nearly parallel to the projection's kernel it accepts points well outside
// the triangle, even outside its box (a "hit" 0.005 in front of the box of a sphere triangle 0.1 across, on the benchmark scene).
// Such a hit X is found by the reference only if no other hit Y with t_Y < (entry of X's box) came first, and the same goes, within
// rounding, for a hit that lies IN a face of its box (axis-aligned triangles: flat boxes; the rounding is absolute, so for the
// short rays that start on a wall and point into it the window is a tenth of t).  So the walkers report the best hit and how far
// behind it the runner-up lies, and the hit goes to the exact walk when the runner-up is within
//     window = max_k (rounded entry of slab k) - t = c1 t - min_k a_k,     a_k = (in_k - c2) / |d_k|   (terms of pt_box_robust)
// (in_k: distance from the hit point back to the entry face of slab k, negative when the point is outside) or within the plain tie
// tolerance of 4 ulp.  To see the runner-up whatever tree they walk, the walkers prune boxes and drop farther hits only beyond
// best_t + max(cull_k best_t, window): the relative part (SceneView::cull_k = 2^-7) is what lets them find a BETTER hit of the first
// kind, whose box lies behind the hit they hold.
// The runner-up's distance behind the hit travels in six bits of the hit word: code c > 0 means (t2 - t) / t >= 2^(c - 41),
// c = 0 means "no farther than 2^-40 t" (or equal); the gate works with that floor (at most 2x too cautious).
RT_DEV uint32_t pt_gap_code(float t, float t2) {
    const float r = (t2 - t) * pt_rcp(t) * 0.99999976f; // a floor: two ulp down covers the reciprocal's error
    if (!(r > 0.f)) return 0u;
    const int c = (int)(__float_as_uint(r) >> 23) - 127 + 41;
    return (uint32_t)(c < 0 ? 0 : (c > 63 ? 63 : c)) << WF_GAP_SHIFT;
}
RT_DEV float pt_gap_floor(uint32_t hit, float t) {
    const uint32_t c = (hit >> WF_GAP_SHIFT) & 63u;
    return c ? __uint_as_float((c - 41u + 127u) << 23) * t : 0.f;
}
// Absolute part of the walkers' look-behind: the window of a hit that lies inside its box (in_k >= 0) is at most c1 t + c2 max_k 1/|d_k|.
// c2x = 1.25f * c2, formed on the host (SceneView::box_c2x): a uniform float product would otherwise sit in a VGPR for the whole launch.
RT_DEV float pt_look_behind_abs(F3 d, float c2x) {
    return c2x * (1.0000005f * pt_rcp(fmaxf(fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)), 1e-30f))); // a ceiling (the gate's `seen` repeats it a hair smaller)
}
// The gate: does the walkers' hit (t, runner-up at t2) stand as the reference's answer?  Yes when every box above it passes the
// reference's test robustly (pt_box_robust), the runner-up lies beyond the hit's window and beyond the tie tolerance, and the window
// does not reach past what the walkers looked at (a hit reported well in front of its own box).
RT_DEV bool pt_hit_stands(F3 lo, F3 hi, F3 o, F3 d, float t, float gap, float c2, float c2x, float cull_k) {
    const F3 P = o + t * d;
    if (pt_deep_inside(lo, hi, P, d, t, c2)) return gap > 0.f && gap > 4.8e-7f * (t + gap); // box and window conditions hold (window <= 0): only the tie rule is left
    const float ix = pt_rcp(fmaxf(fabsf(d.x), 1e-30f)), iy = pt_rcp(fmaxf(fabsf(d.y), 1e-30f)), iz = pt_rcp(fmaxf(fabsf(d.z), 1e-30f));
    const float inx = d.x > 0 ? P.x - lo.x : hi.x - P.x, outx = d.x > 0 ? hi.x - P.x : P.x - lo.x;
    const float iny = d.y > 0 ? P.y - lo.y : hi.y - P.y, outy = d.y > 0 ? hi.y - P.y : P.y - lo.y;
    const float inz = d.z > 0 ? P.z - lo.z : hi.z - P.z, outz = d.z > 0 ? hi.z - P.z : P.z - lo.z;
    const float ax = (inx - c2) * ix, ay = (iny - c2) * iy, az = (inz - c2) * iz;
    const float bx = (outx - c2) * ix, by = (outy - c2) * iy, bz = (outz - c2) * iz;
    const float need = 1.9073486328125e-06f * t;
    const float worst = fminf(fminf(fminf(ax + by, ax + bz), fminf(ay + bx, ay + bz)), fminf(az + bx, az + by));
    const float exit_ = t + fminf(fminf(bx, by), bz);
    const float window = need - fminf(fminf(ax, ay), az);
    const float seen = fmaxf(cull_k * t, c2x * (0.9999995f * fmaxf(fmaxf(ix, iy), iz)));   // the walkers' look-behind for this ray and t, as a floor
    return worst >= need && exit_ >= need && gap > window && gap > 4.8e-7f * (t + gap) && window <= seen; // NaN compares false: exact walk
}

// Does the ray pierce one of the scene's tripwires (host/scene_prep.h: the leaf boxes of the triangles whose test accepts points far from the
// triangle) and pass that triangle's test?  Such a ray's closest hit goes straight to the exact walk.  Conservative: the boxes carry a generous pad, the reciprocals are
// v_rcp_f32, the interval is widened like slab_test's.  Two levels: up to four groups (wave-uniform records: scalar loads), members on a hit.
RT_DEV bool pt_tripwire(const SceneView &S, F3 o, F3 d) {
    if (S.n_tripwire_groups == 0u) return false;
    const float dx = fabsf(d.x) > 1e-30f ? d.x : copysignf(1e-30f, d.x), dy = fabsf(d.y) > 1e-30f ? d.y : copysignf(1e-30f, d.y), dz = fabsf(d.z) > 1e-30f ? d.z : copysignf(1e-30f, d.z);
    RayInv r; r.o = o; r.inv = f3(pt_rcp(dx), pt_rcp(dy), pt_rcp(dz));
    const float4 *rec = reinterpret_cast<const float4 *>(S.tripwires);
    bool pierced = false;
    for (uint32_t g = 0; g < S.n_tripwire_groups; g++) {
        const float4 lo = rec[2 * g], hi = rec[2 * g + 1];
        float tn;
        if (slab_test(lo, hi, r, RT_T_MAX, tn)) {
            const uint32_t first = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
            for (uint32_t m = first; m < first + count; m++) {
                const float4 mlo = rec[2 * m];
                if (slab_test(mlo, rec[2 * m + 1], r, RT_T_MAX, tn)) {
                    // The reference tests this triangle if its walk gets to the leaf; only a test that passes can change its answer, and a
                    // hit that lies well inside the triangle's own box is an ordinary one (the walkers meet it, the gate judges it).
                    const uint32_t fi = __float_as_uint(mlo.w);
                    const TriIsect T = load_isect(S.tri_isect + fi);
                    float t, u, v; bool inside;
                    if (tri_test(T, o, d, t, u, v, inside)) {
                        const float4 *bx = reinterpret_cast<const float4 *>(S.tri_box) + 2 * (size_t)fi;
                        const float4 blo = bx[0], bhi = bx[1];
                        if (!pt_deep_inside(f3(blo.x, blo.y, blo.z), f3(bhi.x, bhi.y, bhi.z), o + t * d, d, t, S.box_c2)) pierced = true;
                    }
                }
            }
        }
    }
    return pierced;
}

// light_pdf_one (rt_device.h) that also says whether the hit is robust against the reference's box tests (pt_box_robust on
// the light triangle's own box: a, a + b, a + c).
// `index`: the light's position in the reference's light order — carried by the record itself (pad >> 1) in the persistent kernel's own light
// tree (SceneView::lights_walk), whose leaf order is not the light order.
// The same in two steps, for a leaf loop that tests several lights and leaves the (rarer, longer) work on a hit for after the loop:
// pt_light_test is the triangle test alone, pt_light_pdf_hit what light_pdf_one computes once the test has passed.
RT_DEV bool pt_light_test(const LightRec *L, F3 x, F3 d, bool &last, uint32_t &index, float &t, float &u, float &v, bool &inside) {
    const TriIsect T = load_isect(&L->isect);
    last = (T.pad & 1u) != 0;
    index = T.pad >> 1;
    return tri_test(T, x, d, t, u, v, inside);
}
RT_DEV float pt_light_pdf_hit(const SceneView &S, const LightRec *L, F3 x, F3 d, float t, float u, float v, bool inside, bool &robust) {
    robust = true;
    const float4 *q = reinterpret_cast<const float4 *>(L) + 3;
    float4 q1 = q[1], q2 = q[2], q3 = q[3];
    float point_prob = q1.z;
    F3 n3 = f3(q1.w, q2.x, q2.y), dn1 = f3(q2.z, q2.w, q3.x), dn2 = f3(q3.y, q3.z, q3.w);
    F3 sn = n3 + u * dn1 + v * dn2;           // primitives.cpp:110
    sn = normalize(sn);                        // :117
    if (inside) sn = neg(sn);                  // :118-119
    if (S.hw7) { const TriIsect T = load_isect(&L->isect); F3 n = f3(T.nx, T.ny, T.nz); sn = normalize(inside ? neg(n) : n); }
    F3 y = x + t * d;                          // distributions.h:144
    if (S.exact_boxes == 1u) {
        const float4 blo = q[4], bhi = q[5];   // the light's own box (LightRec::box_lo / box_hi)
        robust = pt_box_robust(f3(blo.x, blo.y, blo.z), f3(bhi.x, bhi.y, bhi.z), y, d, t, S.box_c2);
    }
    return point_prob * len2(x - y) / fabsf(dot(d, sn)); // :68-70 (pdfOne, shading normal in hw8)
}
RT_DEV float pt_light_pdf_one(const SceneView &S, const LightRec *L, F3 x, F3 d, bool &last, bool &robust, uint32_t &index) {
    TriIsect T = load_isect(&L->isect);
    last = (T.pad & 1u) != 0;
    index = T.pad >> 1;
    robust = true;
    float t, u, v; bool inside;
    if (!tri_test(T, x, d, t, u, v, inside)) return 0.f;
    const float4 *q = reinterpret_cast<const float4 *>(L) + 3;
    float4 q1 = q[1], q2 = q[2], q3 = q[3];
    float point_prob = q1.z;
    F3 n3 = f3(q1.w, q2.x, q2.y), dn1 = f3(q2.z, q2.w, q3.x), dn2 = f3(q3.y, q3.z, q3.w);
    F3 sn = n3 + u * dn1 + v * dn2;           // primitives.cpp:110
    sn = normalize(sn);                        // :117
    if (inside) sn = neg(sn);                  // :118-119
    if (S.hw7) { F3 n = f3(T.nx, T.ny, T.nz); sn = normalize(inside ? neg(n) : n); }
    F3 y = x + t * d;                          // distributions.h:144
    if (S.exact_boxes == 1u) {
        const float4 blo = q[4], bhi = q[5];   // the light's own box (LightRec::box_lo / box_hi)
        robust = pt_box_robust(f3(blo.x, blo.y, blo.z), f3(bhi.x, bhi.y, bhi.z), y, d, t, S.box_c2);
    }
    return point_prob * len2(x - y) / fabsf(dot(d, sn)); // :68-70 (pdfOne, shading normal in hw8)
}


} // namespace dev
} // namespace rtamd
