// hw6 render kernel (BASELINE.json configs[2]): glTF triangles with flat normals, DIFFUSE / METALLIC(mirror) /
// DIELECTRIC(ior 1.5) materials, Mix{Cosine, FiguresMix}, and the dielectric branch's binary recursion
// (hw6/src/scene.cpp:47-105), replayed per pixel with the reference's minstd_rand stream.
//
// The recursion tree (reflected subtree first, then one uniform, then maybe the refracted subtree;
// <= 2^depth - 1 rays per camera sample) is evaluated with an explicit per-lane frame stack.
// The scene BVH is NOT the reference's (its sort key is a constant, hw6/src/include/bvh.h:61-63, which
// degenerates the tree); figures keep their index in the reference's order for the tie rule, the light BVH
// keeps the reference's topology because its shape fixes the order of the float additions.
#pragma once
#include "rt_device.h"
#include "rt_types_hw6.h"

namespace rtamd {

namespace dev {

#define RT6_MAX_DEPTH 8
#define RT6_STACK_SIZE 128 // hw6's light tree is built on a constant sort key and can be very deep (practice6_2: 85)
#define RT_PI_F 3.14159274101257324219f // const float PI = acos(-1), hw6/src/include/distributions.h:11

struct Tri6Regs { F3 a, b, c, n; uint32_t ref_index, last, material; float point_prob; };
RT_DEV Tri6Regs load_tri6(const Tri6 *p) {
    const float4 *q = reinterpret_cast<const float4 *>(p);
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    Tri6Regs T;
    T.a = f3(q0.x, q0.y, q0.z); T.b = f3(q0.w, q1.x, q1.y); T.c = f3(q1.z, q1.w, q2.x); T.n = f3(q2.y, q2.z, q2.w);
    T.ref_index = __float_as_uint(q3.x); T.last = __float_as_uint(q3.y); T.material = __float_as_uint(q3.z); T.point_prob = q3.w;
    return T;
}
// hw6/src/primitives.cpp:77-86,143-164 (position 0 / identity rotation: the figure transform is the identity).
RT_DEV bool tri6_test(const Tri6Regs &T, F3 o, F3 d, float &t, bool &inside) {
    F3 ro = o - T.a;
    float dn = dot(d, T.n);
    t = -dot(ro, T.n) / dn;
    if (!(t > 0 && t < RT_T_MAX)) return false;
    inside = dn > 0;
    F3 p = ro + t * d;
    if (dot(crossr(T.b, p), T.n) < 0) return false;
    if (dot(crossr(p, T.c), T.n) < 0) return false;
    if (dot(crossr(T.c - T.b, p - T.b), T.n) < 0) return false;
    return true;
}

// tri6_test for a closest-hit walk that already holds a hit: a plane crossed beyond keep_t (the best t plus the walkers' look-behind,
// rt_exact.h) can neither win nor matter as the runner-up, so the three edge tests are skipped; same decisions otherwise.
RT_DEV bool tri6_test_closer(const Tri6Regs &T, F3 o, F3 d, float keep_t, float &t, bool &inside) {
    F3 ro = o - T.a;
    float dn = dot(d, T.n);
    t = -dot(ro, T.n) / dn;
    if (!(t > 0 && t < RT_T_MAX)) return false;
    if (t > keep_t) return false;
    inside = dn > 0;
    F3 p = ro + t * d;
    if (dot(crossr(T.b, p), T.n) < 0) return false;
    if (dot(crossr(p, T.c), T.n) < 0) return false;
    if (dot(crossr(T.c - T.b, p - T.b), T.n) < 0) return false;
    return true;
}

struct Hit6 { int slot; float t; bool inside; uint32_t ref; };

// Wave-synchronous "while-while" walk: the lanes of the wave that are in this call step through inner nodes together until
// enough of them (a share of the walking lanes, at most RT6_LEAF_BATCH) wait at a leaf, then those test their triangles
// together -- the two code paths are no longer interleaved lane by lane.  The result does not depend on the order (tie rule).
#ifndef RT6_LEAF_BATCH
#define RT6_LEAF_BATCH 20
#endif
#ifndef RT6_LEAF_SHARE
#define RT6_LEAF_SHARE 112 // of 256: share of the walking lanes that must wait at a leaf
#endif
RT_DEV Hit6 closest_hit6(const SceneView6 &S, F3 o, F3 d, uint32_t *stack) {
    Hit6 best; best.slot = -1; best.t = RT_T_MAX; best.inside = false; best.ref = 0xFFFFFFFFu;
    RayInv ray = make_ray_inv(o, d);
    int sp = 0;
    uint32_t cur = 0;
    bool walking = true;
    for (;;) {
        const unsigned long long m_walk = __ballot(walking);
        if (!m_walk) break;
        const int lb = min(RT6_LEAF_BATCH, (int)((__popcll(m_walk) * RT6_LEAF_SHARE + 255) >> 8));
        for (;;) { // phase 1: inner nodes
            const bool inner = walking && !(cur & RT_LEAF_BIT);
            if (!__ballot(inner) || (int)__popcll(__ballot(walking && (cur & RT_LEAF_BIT))) >= lb) break;
            if (inner) {
                const float4 *q = reinterpret_cast<const float4 *>(S.nodes + cur);
                float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                float n0, n1;
                bool h0 = slab_test(lo0, hi0, ray, best.t, n0);
                bool h1 = slab_test(lo1, hi1, ray, best.t, n1);
                uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
                if (h0 & h1) { bool swap = n1 < n0; stack[sp++] = swap ? c0 : c1; cur = swap ? c1 : c0; }
                else if (h0) cur = c0;
                else if (h1) cur = c1;
                else if (sp == 0) walking = false;
                else cur = stack[--sp];
            }
        }
        if (walking && (cur & RT_LEAF_BIT)) { // phase 2: leaves
            if (cur != RT_EMPTY_LEAF) {
                uint32_t i = cur & ~RT_LEAF_BIT;
                for (;;) {
                    Tri6Regs T = load_tri6(S.tris + i);
                    float t; bool inside;
                    // reference tie rule: smallest t, equal t -> lowest index in the reference's figure order
                    if (tri6_test(T, o, d, t, inside) && (t < best.t || (t == best.t && T.ref_index < best.ref))) {
                        best.t = t; best.inside = inside; best.slot = (int)i; best.ref = T.ref_index;
                    }
                    if (T.last) break;
                    i++;
                }
            }
            if (sp == 0) walking = false;
            else cur = stack[--sp];
        }
    }
    return best;
}

// FiguresMix::getTotalPdf for triangle lights (hw6/src/include/distributions.h:212-256), reference addition tree.
// `stack`: anything indexable that holds RT6_STACK_SIZE words (a private array here; a strided slice of LDS in rt_persistent_hw6.h).
template <class A>
RT_DEV float light_pdf_sum6(const SceneView6 &S, F3 x, F3 d, A stack) {
    RayInv ray = make_ray_inv(x, d);
    int sp = 0;
    unsigned long long mask_lo = 0, mask_hi = 0; // frame kind per stack slot: 1 = ADD(partial sum), 0 = TODO(child)
    uint32_t cur = 0;
    bool descending = true;
    float v = 0.f;
    for (;;) {
        if (descending) {
            if (cur & RT_LEAF_BIT) {
                float result = 0.f;
                if (cur != RT_EMPTY_LEAF) {
                    uint32_t i = cur & ~RT_LEAF_BIT;
                    for (;;) {
                        Tri6Regs T = load_tri6(S.lights + i);
                        float t; bool inside; float term = 0.f;
                        if (tri6_test(T, x, d, t, inside)) {
                            F3 yn = normalize(inside ? neg(T.n) : T.n);            // primitives.cpp:31
                            F3 y = x + t * d;
                            term = T.point_prob * len2(x - y) / fabsf(dot(d, yn)); // distributions.h:116-118
                        }
                        result += term;
                        if (T.last) break;
                        i++;
                    }
                }
                v = result; descending = false; continue;
            }
            const float4 *q = reinterpret_cast<const float4 *>(S.light_nodes + cur);
            float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
            float n0, n1;
            bool h0 = slab_test(lo0, hi0, ray, RT_T_MAX, n0);
            bool h1 = slab_test(lo1, hi1, ray, RT_T_MAX, n1);
            uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
            if (h0 & h1) {
                if (sp < 64) mask_lo &= ~(1ull << sp); else mask_hi &= ~(1ull << (sp - 64));
                stack[sp++] = c1; cur = c0;
            }
            else if (h0) cur = c0;
            else if (h1) cur = c1;
            else { v = 0.f; descending = false; }
        } else {
            if (sp == 0) break;
            --sp;
            uint32_t f = stack[sp];
            bool is_add = sp < 64 ? ((mask_lo >> sp) & 1ull) != 0 : ((mask_hi >> (sp - 64)) & 1ull) != 0;
            if (is_add) v = __uint_as_float(f) + v;
            else {
                if (sp < 64) mask_lo |= 1ull << sp; else mask_hi |= 1ull << (sp - 64);
                stack[sp++] = __float_as_uint(v); cur = f; descending = true;
            }
        }
    }
    return v;
}

// The same sum without the reference tree's boxes.  The reference's light tree (constant sort key) costs thousands of box
// tests per query, but only the ORDER of its additions matters.  So: (1) a properly keyed tree over the same lights finds the
// lights the ray hits and their terms; (2) no, one or two hits need no order (x + 0 = x, a + b = b + a); (3) more hits are
// added in the reference's association by walking the reference tree's index ranges only where hits lie: sum(node) =
// sum(left) + sum(right), a side without hits contributes the additive identity, a leaf adds its hits in index order.
#define RT6_MAX_LIGHT_HITS 16
// step (3): the hits of one query (reference light index, term), at most RT6_MAX_LIGHT_HITS, added in the reference's association
RT_DEV float light_sum6_associate(const SceneView6 &S, uint32_t *hit_idx, float *hit_term, int k) {
    for (int i = 1; i < k; i++) { // insertion sort by the reference's light index
        uint32_t id = hit_idx[i]; float tm = hit_term[i];
        int j = i - 1;
        while (j >= 0 && hit_idx[j] > id) { hit_idx[j + 1] = hit_idx[j]; hit_term[j + 1] = hit_term[j]; j--; }
        hit_idx[j + 1] = id; hit_term[j + 1] = tm;
    }
    // frames: a postponed right side {node, lo, hi} (kind 0) or a finished left total waiting for its right side (kind 1)
    uint32_t f_node[RT6_MAX_LIGHT_HITS]; int f_lo[RT6_MAX_LIGHT_HITS], f_hi[RT6_MAX_LIGHT_HITS]; float f_val[RT6_MAX_LIGHT_HITS]; uint32_t f_add = 0;
    int fsp = 0;
    uint32_t node = 0; int lo = 0, hi = k;
    float v = 0.f;
    for (;;) {
        for (;;) { // total of hits [lo, hi) under `node`
            uint4 n = reinterpret_cast<const uint4 *>(S.light_ref)[node];
            if (n.x == 0) { v = 0.f; for (int j = lo; j < hi; j++) v += hit_term[j]; break; }   // leaf: sequential, index order
            uint32_t right_first = S.light_ref[4 * n.y + 2];
            int m = lo;
            while (m < hi && hit_idx[m] < right_first) m++;
            if (m == lo) { node = n.y; continue; }        // nothing on the left:  0 + right
            if (m == hi) { node = n.x; continue; }        // nothing on the right: left + 0
            f_node[fsp] = n.y; f_lo[fsp] = m; f_hi[fsp] = hi; f_add &= ~(1u << fsp); fsp++;
            node = n.x; hi = m;
        }
        for (;;) { // fold finished totals, or open the next postponed right side
            if (fsp == 0) return v;
            fsp--;
            if ((f_add >> fsp) & 1u) { v = f_val[fsp] + v; continue; }                              // left total + right total
            node = f_node[fsp]; lo = f_lo[fsp]; hi = f_hi[fsp];
            f_val[fsp] = v; f_add |= 1u << fsp; fsp++;
            break;
        }
    }
}


RT_DEV float light_pdf_sum6_fast(const SceneView6 &S, F3 x, F3 d, uint32_t *stack, uint32_t *deep_stack) {
    RayInv ray = make_ray_inv(x, d);
    int sp = 0, k = 0;
    uint32_t hit_idx[RT6_MAX_LIGHT_HITS];
    float hit_term[RT6_MAX_LIGHT_HITS];
    uint32_t cur = 0;
    bool walking = true, too_many = false;
    for (;;) { // wave-synchronous while-while walk, as in closest_hit6
        const unsigned long long m_walk = __ballot(walking);
        if (!m_walk) break;
        const int lb = min(RT6_LEAF_BATCH, (int)((__popcll(m_walk) * RT6_LEAF_SHARE + 255) >> 8));
        for (;;) { // phase 1: inner nodes
            const bool inner = walking && !(cur & RT_LEAF_BIT);
            if (!__ballot(inner) || (int)__popcll(__ballot(walking && (cur & RT_LEAF_BIT))) >= lb) break;
            if (inner) {
                const float4 *q = reinterpret_cast<const float4 *>(S.fast_light_nodes + cur);
                float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
                float n0, n1;
                bool h0 = slab_test(lo0, hi0, ray, RT_T_MAX, n0);
                bool h1 = slab_test(lo1, hi1, ray, RT_T_MAX, n1);
                uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
                if (h0 & h1) { stack[sp++] = c1; cur = c0; }
                else if (h0) cur = c0;
                else if (h1) cur = c1;
                else if (sp == 0) walking = false;
                else cur = stack[--sp];
            }
        }
        if (walking && (cur & RT_LEAF_BIT)) { // phase 2: leaves
            if (cur != RT_EMPTY_LEAF) {
                uint32_t i = cur & ~RT_LEAF_BIT;
                for (;;) {
                    Tri6Regs T = load_tri6(S.fast_lights + i);
                    float t; bool inside;
                    if (tri6_test(T, x, d, t, inside)) {
                        if (k == RT6_MAX_LIGHT_HITS) { too_many = true; break; }
                        F3 yn = normalize(inside ? neg(T.n) : T.n);                  // primitives.cpp:31
                        F3 y = x + t * d;
                        hit_idx[k] = T.ref_index;
                        hit_term[k] = T.point_prob * len2(x - y) / fabsf(dot(d, yn)); // distributions.h:116-118
                        k++;
                    }
                    if (T.last) break;
                    i++;
                }
            }
            if (sp == 0 || too_many) walking = false;
            else cur = stack[--sp];
        }
    }
    if (too_many) return light_pdf_sum6(S, x, d, deep_stack); // more hits than kept: the plain reference-order walk
    if (k == 0) return 0.f;
    if (k == 1) return hit_term[0];
    if (k == 2) return hit_term[0] + hit_term[1];
    return light_sum6_associate(S, hit_idx, hit_term, k);
}

enum { F6_MUL = 0, F6_DIEL_REFLECT = 1, F6_DIEL_REFRACT = 2 };
struct Frame6 {
    F3 emission, mult;    // F6_MUL: result = emission + mult * child ; DIEL frames: mult = material colour
    F3 x, dn, norma;      // hit point (ray.o + t*ray.d), normalised incoming direction, normal
    int kind; bool inside; float ior;
};

// Scene::getColor of hw6 as an explicit stack machine that advances ONE step per call (one closest-hit + shading, or one
// return step), so the lanes of a wave can sit at different samples and pixels: a lane whose path ends starts its next
// camera sample at once instead of idling until the longest path of the wave has finished.
struct Machine6 {
    Frame6 frames[RT6_MAX_DEPTH];
    uint32_t n_closest, n_light; // queries answered for this lane (RT_FLAG_COUNTERS)
    int fp;
    bool evaluating; // true: evaluate getColor(o, d, ray_depth - fp) ; false: `ret` is a finished child value
    F3 o, d, ret;
};
RT_DEV void machine6_start(Machine6 &M, F3 o, F3 d) { M.fp = 0; M.evaluating = true; M.o = o; M.d = d; M.ret = f3(0.f, 0.f, 0.f); }
// Returns true when the path is complete (M.ret = getColor of the camera ray).
RT_DEV bool machine6_step(const SceneView6 &S, int ray_depth, Rng &rng, Machine6 &M, uint32_t *stack, uint32_t *deep_stack) {
    Frame6 *frames = M.frames;
    int &fp = M.fp;
    const float epsf = 9.99999974737875163555e-05f; // (float)1e-4L
    F3 &ret = M.ret, &o = M.o, &d = M.d;
    bool &evaluating = M.evaluating;
    do {
        if (evaluating) {
            if (fp >= ray_depth) { ret = f3(0.f, 0.f, 0.f); evaluating = false; continue; }   // recLimit == 0
            Hit6 h = closest_hit6(S, o, d, stack);
            M.n_closest++;
            if (h.slot < 0) { ret = f3(S.bg); evaluating = false; continue; }
            Tri6Regs T = load_tri6(S.tris + h.slot);
            const float4 *qm = reinterpret_cast<const float4 *>(S.materials + T.material);
            float4 m0 = qm[0], m1 = qm[1];
            F3 color = f3(m0.x, m0.y, m0.z), emission = f3(m1.x, m1.y, m1.z);
            int kind = (int)__float_as_uint(m1.w);
            F3 norma = normalize(h.inside ? neg(T.n) : T.n);                                   // primitives.cpp:81-83,31
            F3 x = o + h.t * d;                                                                 // scene.cpp:60
            Frame6 &f = frames[fp];
            if (kind == RT_MAT_DIFFUSE) {
                F3 xo = x + epsf * norma;
                int comp = (int)(rng_u01(rng) * S.n_components_f);                         // distributions.h:284
                F3 nd;
                if (comp == 0) nd = cosine_sample(rng, norma);
                else {                                                                          // :199-208, :129-141
                    int li = (int)(rng_u01(rng) * S.n_lights_f);
                    Tri6Regs L = load_tri6(S.lights + li);
                    float u = rng_u01(rng);
                    float v = rng_u01(rng);
                    if ((double)(u + v) > 1.) { u = 1 - u; v = 1 - v; }
                    F3 point = L.a + u * L.b + v * L.c;
                    nd = normalize(point - xo);
                }
                if (dot(nd, norma) < 0) { ret = emission; evaluating = false; continue; }       // scene.cpp:64-66
                float pdf = 0.f;
                pdf += smax(0.f, dot(nd, norma) / RT_PI_F);                                     // distributions.h:55-58
                if (S.n_components == 2) { pdf += light_pdf_sum6_fast(S, xo, nd, stack, deep_stack) / S.n_lights_f; M.n_light++; }
                pdf = pdf / S.n_components_f;
                float k = (float)(1. / (double)(RT_PI_F * pdf) * (double)dot(nd, norma));       // scene.cpp:69
                f.kind = F6_MUL; f.emission = emission; f.mult = k * color;
                fp++;
                o = x + epsf * nd; d = nd;                                                      // scene.cpp:68
            } else {
                F3 dn = normalize(d);
                F3 refl = dn - (float)(2. * (double)dot(norma, dn)) * norma;                    // scene.cpp:71,75
                f.emission = emission; f.mult = color; f.x = x; f.dn = dn; f.norma = norma; f.inside = h.inside; f.ior = m0.w;
                f.kind = kind == RT_MAT_METALLIC ? F6_MUL : F6_DIEL_REFLECT;
                fp++;
                o = x + epsf * refl; d = refl;                                                  // scene.cpp:72,76
            }
        } else {
            if (fp == 0) return true;
            Frame6 &f = frames[--fp];
            if (f.kind == F6_MUL) { ret = f.emission + f.mult * ret; continue; }                // scene.cpp:69,73
            if (f.kind == F6_DIEL_REFRACT) {                                                    // scene.cpp:99-103
                F3 refracted = ret;
                if (!f.inside) refracted = refracted * f.mult;
                ret = f.emission + refracted;
                continue;
            }
            // F6_DIEL_REFLECT: `ret` is reflectedColor (scene.cpp:77-98)
            float eta1 = 1.f, eta2 = f.ior;
            if (f.inside) { float tmp = eta1; eta1 = eta2; eta2 = tmp; }
            F3 l = neg(f.dn);
            float nl = dot(f.norma, l);
            float sinTheta2 = (float)((double)(eta1 / eta2) * sqrt((double)(1 - nl * nl)));
            if (fabs((double)sinTheta2) > 1.) { ret = f.emission + ret; continue; }
            float rr = (eta1 - eta2) / (eta1 + eta2);
            float r0 = rr * rr;                                                                  // pow(., 2.) == exact square
            double om = (double)(1 - nl), om2 = om * om;
            float r = (float)((double)r0 + (double)(1 - r0) * (om2 * om2 * om));                 // pow(., 5.)
            if (rng_u01(rng) < r) { ret = f.emission + ret; continue; }
            float cosTheta2 = sqrtf(1 - sinTheta2 * sinTheta2);
            F3 refr = (eta1 / eta2) * neg(l) + (eta1 / eta2 * nl - cosTheta2) * f.norma;
            f.kind = F6_DIEL_REFRACT;
            fp++;
            o = f.x + epsf * refr; d = refr;
            evaluating = true;
            continue;
        }
    } while (false);
    return false;
}

// LDS_STACK: the traversal stacks of the closest-hit walk and of the fast light walk live in LDS (one column of RT6_LDS_STACK
// entries per lane, odd stride: conflict-free) instead of scratch memory; the host selects it when both of the library's own
// trees are at most that deep.  The reference-order light walk (only after more than 16 hits) keeps its deep private stack.
#ifndef RT6_LDS_STACK
#define RT6_LDS_STACK 36   // 9.5 KB per wave: 16 waves per CU; measured on config 3 (stack entries / waves per SIMD): 36/4 61.6, 32/5 58.7, 40/4 57.6, 28/6 55.1, 48/3 55.5 Msamples/s
#endif
#define RT6_LDS_STRIDE (RT6_LDS_STACK + 1)
#ifndef RT6_MIN_WAVES
#define RT6_MIN_WAVES 4    // waves per SIMD the register allocation aims at (128 VGPR, some spills)
#endif
template <bool LDS_STACK>
__global__ __launch_bounds__(64, RT6_MIN_WAVES) void render_hw6_kernel(SceneView6 S, RenderView R, uint32_t n_work) {
    __shared__ uint32_t lds_stack[LDS_STACK ? 64 * RT6_LDS_STRIDE : 1];
    uint32_t deep_stack[RT6_STACK_SIZE];
    const int lane = threadIdx.x & 63;
    uint32_t *stack = LDS_STACK ? lds_stack + lane * RT6_LDS_STRIDE : deep_stack;
    const uint32_t n_slots = n_work * 64u;   // pixel slots in the tile order of slot_to_pixel()
    Machine6 M;
    M.n_closest = 0; M.n_light = 0;
    machine6_start(M, f3(0.f, 0.f, 0.f), f3(0.f, 0.f, 1.f));
    Rng rng; rng_seed(rng, 0u);
    F3 color = f3(0.f, 0.f, 0.f);
    int x = 0, y = 0, s = 0;
    size_t out_index = 0;
    bool have_pixel = false, exhausted = false;
    auto camera_ray = [&]() {                                                                    // scene.cpp:111-126 (direction not normalised)
        float nx = (float)x + rng_u01(rng);
        float ny = (float)y + rng_u01(rng);
        float cx = R.tan_fov_x * (2 * nx / (float)R.width - 1);
        float cy = S.tan_fov_y * (2 * ny / (float)R.height - 1);
        machine6_start(M, f3(S.cam_pos), cx * f3(S.cam_right) - cy * f3(S.cam_up) + f3(S.cam_fwd));
    };
    for (;;) {
        unsigned long long need = __ballot(!have_pixel);
        if (need && !exhausted) {                                      // lanes without a pixel take the next pixel slots
            int leader = __ffsll((long long)need) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(R.work_counter, (uint32_t)__popcll(need));
            base = __shfl(base, leader);
            if (base + (uint32_t)__popcll(need) >= n_slots) exhausted = true;
            if (!have_pixel) {
                uint32_t slot = base + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                if (slot < n_slots) {
                    bool inside;
                    slot_to_pixel(R, slot, x, y, inside, out_index);
                    if (inside) {
                        rng_seed(rng, (uint32_t)(y * R.width + x));                              // hw6/src/sceneio.cpp:281-284
                        color = f3(0.f, 0.f, 0.f); s = 0;
                        camera_ray();
                        have_pixel = true;
                    } else if (R.shard_count > 1) {                                              // padding of a border tile in the compact shard layout
                        if (R.out_rgb) { R.out_rgb[3 * out_index] = 0.f; R.out_rgb[3 * out_index + 1] = 0.f; R.out_rgb[3 * out_index + 2] = 0.f; }
                        if (R.out_rgb8) { R.out_rgb8[3 * out_index] = 0; R.out_rgb8[3 * out_index + 1] = 0; R.out_rgb8[3 * out_index + 2] = 0; }
                    }
                }
            }
        }
        if (!__ballot(have_pixel)) { if (exhausted) break; else continue; }
        if (have_pixel && machine6_step(S, R.ray_depth, rng, M, stack, deep_stack)) {
            color = color + M.ret;                                                               // scene.cpp:113
            if (++s < R.samples) camera_ray();
            else {
                F3 px = R.inv_samples * color;                                                   // scene.cpp:115
                if (R.out_rgb) { R.out_rgb[3 * out_index] = px.x; R.out_rgb[3 * out_index + 1] = px.y; R.out_rgb[3 * out_index + 2] = px.z; }
                if (R.out_rgb8) { R.out_rgb8[3 * out_index] = tonemap1(px.x); R.out_rgb8[3 * out_index + 1] = tonemap1(px.y); R.out_rgb8[3 * out_index + 2] = tonemap1(px.z); }
                have_pixel = false;
            }
        }
    }
    if (R.counters) { atomicAdd(&R.counters[0], (unsigned long long)M.n_closest); atomicAdd(&R.counters[1], (unsigned long long)M.n_light); }
}

} // namespace dev
} // namespace rtamd
