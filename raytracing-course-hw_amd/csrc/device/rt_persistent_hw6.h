// Persistent dataflow form of the hw6 replay path tracer (BASELINE.json configs[2]): the organisation of rt_persistent.h applied to
// hw6's integrator (hw6/src/scene.cpp:47-105) with its binary recursion tree.
//
// render_hw6_kernel (rt_kernels_hw6.h) keeps a whole path machine per lane — an 8-frame recursion stack in scratch, walks that are
// wave-synchronous but never refill, so every step of a wave lasts as long as its longest walk through the scene (a glass bunny in
// practice6_2).  Here the frames of a path live in its HBM record and the work is cut at the walks:
//
//   closest-hit walkers / light-sum walkers   the while-while loops over hw6's own trees, lanes refilled from LDS bitmaps;
//   shaders                                   run the frame machine of a path between two walks: finish the pending DIFFUSE bounce
//                                             (its light-pdf sum has arrived), shade the new hit, push a frame, or unwind frames
//                                             (MUL / dielectric reflect -> Fresnel -> one uniform -> maybe the refracted child)
//                                             until the path needs its next walk, its next camera sample, or is finished.
//
// The random stream of a pixel is consumed only by the shader steps, in the reference's order (the reflected subtree completes before
// the dielectric's uniform is drawn, scene.cpp:77,92), so the replay is the one of rt_kernels_hw6.h; the arithmetic is the same code.
// As in rt_persistent.h a DIFFUSE bounce's sampled direction is traced for the next hit while its light-pdf sum is walked (the two
// rays share the direction but not the origin here: x + eps*d for the child, x + eps*n for the pdf, scene.cpp:63,68).
#pragma once
#include "rt_persistent.h"
#include "rt_kernels_hw6.h"
#include "rt_exact.h"

namespace rtamd {
namespace dev {

// As rt_persistent.h: workgroups of four waves (one per SIMD), five per CU = five waves per SIMD (<= 96 VGPRs, no scratch).
#define P6_WAVES 4
#define P6_THREADS (64 * P6_WAVES)
#define P6_PER_CU 5
#define P6_STACK 28                  // LDS stack entries per lane: the scene tree (GPU-built, depth <= 28) and the own tree over the lights must fit
#define P6_MAX_PATHS 2304            // paths per workgroup: 4 x 28 x 256 B of stacks + 1.4 B per path = 31.9 KB, five workgroups per CU
#define P6_NW (P6_MAX_PATHS / 32)
// The rare roles that need work arrays per query (exact walks, light sums at a box boundary or with many hits) run P6_XBATCH queries at
// a time and keep those arrays in the wave's LDS stack area, idle meanwhile: word i of lane j's slice at area[P6_XBATCH * i + j],
// 224 words per slice.  No scratch.
#define P6_XBATCH 8
#define P6_SLICE_WORDS (P6_STACK * 64 / P6_XBATCH)
#define P6_COST_TRACE_STEP 1u        // what a sub-tile costs its workgroup, in closest-hit node steps (the measure of the re-deal, as PT_COST_*)
#define P6_COST_LIGHT_STEP 2u
#define P6_COST_SHADE 10u
#ifndef P6_LIGHT_PRETEST
#define P6_LIGHT_PRETEST false   // the deep-inside shortcut of pt_box_robust in the light walker: its few extra live values are exactly what tips this kernel (at 96 VGPRs) into spilling
#endif
#define P6_MERGE_HITS 13             // p6_merge_hits works in the lane's own 28-word column: 13 terms + 14 index / depth words
#define P6_Q_SLOW 3                  // light sums with more than two hits: the complete per-lane light_pdf_sum6_fast
#define P6_REC 56u                   // float4 per path record: 8 + 5 per frame x RT6_MAX_DEPTH + 8 for the hits of a light sum
// record: r0 = o.xyz d.x | r1 = d.yz rng.x rng.saved | r2 = hit t, figure slot, inside, t of the runner-up | r3 = accum.xyz packed
//         r4 = light-query origin xo.xyz, light sum (raw)  | r5 = pending emission.xyz, cosine pdf | r6 = pending colour.xyz, d.n
//         r7 = number of lights the pdf's ray hit (when more than two), -, -, -
//         frame f at r[8 + 5 f]: emission.xyz kind | mult.xyz inside | x.xyz ior | dn.xyz - | norma.xyz -
//         r[48..55]: up to 16 hits of the light sum {reference light index, term}, for the slow role
// packed: fp:4 | has_saved:16 | pending:32 | light_only:64 (the pending bounce's child is beyond the depth limit: no trace) | sample << 8
#define P6_PENDING 32u
#define P6_LIGHT_ONLY 64u
#define P6_VERIFIED 128u             // the hit in r2 comes from the reference-exact walk: its index is a position in the reference's figure order
#define P6_Q_XTRACE 4                // closest hits for the exact walk (rt_exact.h)
// actions of p6_advance
#define P6_TRACE 1
#define P6_LIGHT 2
#define P6_PARKED 8
#define P6_EXACT 16                  // the hit does not stand as the reference's answer: exact walk first, nothing of the path was touched

typedef __attribute__((address_space(3))) uint32_t *P6Lds; // an LDS pointer that stays one: 32 bits, not a 64-bit generic pointer in two VGPRs
typedef __attribute__((address_space(3))) float *P6LdsF;
struct P6Slice { // a strided view of the wave's LDS stack area: indexable like an array
    P6Lds p;
    RT_DEV __attribute__((address_space(3))) uint32_t &operator[](int i) const { return p[P6_XBATCH * i]; }
    RT_DEV P6Slice at(int first_word) const { P6Slice s; s.p = p + P6_XBATCH * first_word; return s; }
};
struct P6SliceF { // the same words read as floats
    P6Lds p;
    RT_DEV __attribute__((address_space(3))) float &operator[](int i) const { return ((P6LdsF)p)[P6_XBATCH * i]; }
};

struct P6Shared {
    uint32_t stack[P6_WAVES][P6_STACK][64];
    uint32_t need[5][P6_NW];
    uint32_t pending[P6_NW * 2];
    uint32_t groups[P6_MAX_PATHS / PT_MIN_GROUP];
    uint32_t cost[P6_MAX_PATHS / PT_MIN_GROUP];   // shader steps per local sub-tile in this launch: the load measure of the re-deal
    int cnt[16];
};

struct W6View { float4 *r0; uint32_t slot_base; };
RT_DEV float4 *p6_rec(const W6View &W, uint32_t slot) { return W.r0 + (size_t)slot * P6_REC; }
RT_DEV uint32_t p6_pack(int fp, bool has_saved, uint32_t sample, uint32_t flags) { return (uint32_t)fp | (has_saved ? 16u : 0u) | flags | (sample << 8); }

RT_DEV void p6_camera_ray(const SceneView6 &S, const RenderView &R, Rng &rng, int x, int y, F3 &o, F3 &d) { // hw6/src/scene.cpp:111-126 (direction not normalised)
    float nx = (float)x + rng_u01(rng);
    float ny = (float)y + rng_u01(rng);
    float cx = R.tan_fov_x * (2 * nx / (float)R.width - 1);
    float cy = S.tan_fov_y * (2 * ny / (float)R.height - 1);
    o = f3(S.cam_pos);
    d = cx * f3(S.cam_right) - cy * f3(S.cam_up) + f3(S.cam_fwd);
}

// ---- shader: the frame machine of one path between two walks (machine6_step of rt_kernels_hw6.h, cut at the walks) --------------
RT_DEV int p6_advance(const SceneView6 &S, const RenderView &R, const W6View &W, uint32_t slot) {
    const float epsf = 9.99999974737875163555e-05f; // (float)1e-4L
    float4 *r = p6_rec(W, slot);
    const float4 q1 = r[1];
    Rng rng; rng.x = __float_as_uint(q1.z); rng.saved = q1.w;
    const uint32_t packed = __float_as_uint(reinterpret_cast<const float *>(r + 3)[3]);
    int fp = (int)(packed & 15u);
    rng.has_saved = (packed & 16u) != 0;
    uint32_t sample = packed >> 8;
    if (!(packed & (P6_LIGHT_ONLY | P6_VERIFIED)) && r[2].x == PT_T_OVERFLOW) return P6_EXACT; // the walk ran out of stack (p6_trace_stint)
    if (S.exact_boxes && !(packed & (P6_LIGHT_ONLY | P6_VERIFIED))) {
        // the gate of rt_exact.h (pt_hit_stands), before anything of the path's state changes
        const float4 g0 = r[0], g2 = r[2];
        const uint32_t ghit = __float_as_uint(g2.y);
        if (ghit != 0xFFFFFFFFu) {
            const uint32_t ref_index = __float_as_uint(reinterpret_cast<const float4 *>(S.tris + ghit)[3].x);
            const float4 *bx = reinterpret_cast<const float4 *>(S.tri_box) + 2 * (size_t)ref_index;
            const float4 lo = bx[0], hi = bx[1];
            if (!pt_hit_stands(f3(lo.x, lo.y, lo.z), f3(hi.x, hi.y, hi.z), f3(g0.x, g0.y, g0.z), f3(g0.w, q1.x, q1.y), g2.x, g2.w - g2.x, S.box_c2, S.box_c2x, S.cull_k)) return P6_EXACT;
        }
    }
    F3 ret = f3(0.f, 0.f, 0.f);
    bool returning = false;
    if (packed & P6_PENDING) {
        // the DIFFUSE bounce at level fp: its light-pdf sum is in r4.w now (scene.cpp:67-69, distributions.h:288-300)
        const float4 q4 = r[4], q5 = r[5], q6 = r[6];
        float pdf = 0.f;
        pdf += q5.w;
        if (S.n_components == 2) pdf += q4.w / S.n_lights_f;
        pdf = pdf / S.n_components_f;
        const float k = (float)(1. / (double)(RT_PI_F * pdf) * (double)q6.w);                   // scene.cpp:69
        float4 *f = r + 8 + 5 * fp;
        f[0] = make_float4(q5.x, q5.y, q5.z, __uint_as_float((uint32_t)F6_MUL));
        const F3 mult = k * f3(q6.x, q6.y, q6.z);
        f[1] = make_float4(mult.x, mult.y, mult.z, 0.f);
        fp++;
        if (packed & P6_LIGHT_ONLY) returning = true;                                          // the child sits beyond the depth limit: it returned 0
    }
    if (!returning) {
        // the walk that just finished answered getColor's intersect() at level fp (scene.cpp:49-60)
        const float4 q0 = r[0], q2 = r[2];
        const F3 o = f3(q0.x, q0.y, q0.z), d = f3(q0.w, q1.x, q1.y);
        const uint32_t hit = __float_as_uint(q2.y);
        if (hit == 0xFFFFFFFFu) { ret = f3(S.bg); returning = true; }
        else {
            const bool inside = __float_as_uint(q2.z) != 0;
            Tri6Regs T = load_tri6((packed & P6_VERIFIED) ? S.ref_tris + hit : S.tris + hit);
            const float4 *qm = reinterpret_cast<const float4 *>(S.materials + T.material);
            const float4 m0 = qm[0], m1 = qm[1];
            const F3 color = f3(m0.x, m0.y, m0.z), emission = f3(m1.x, m1.y, m1.z);
            const int kind = (int)__float_as_uint(m1.w);
            const F3 norma = normalize(inside ? neg(T.n) : T.n);                                // primitives.cpp:81-83,31
            const F3 x = o + q2.x * d;                                                          // scene.cpp:60
            if (kind == RT_MAT_DIFFUSE) {
                const F3 xo = x + epsf * norma;
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
                const float dnn = dot(nd, norma);
                if (dnn < 0) { ret = emission; returning = true; }                              // scene.cpp:64-66
                else {
                    const float pdf_cos = smax(0.f, dnn / RT_PI_F);                             // distributions.h:55-58
                    const bool child_traced = fp + 1 < R.ray_depth;
                    if (S.n_components == 2 || child_traced) {
                        if (child_traced) {
                            const F3 no = x + epsf * nd;                                        // scene.cpp:68
                            r[0] = make_float4(no.x, no.y, no.z, nd.x);
                        } else reinterpret_cast<float *>(r)[3] = nd.x;                          // the light query still needs the direction
                        r[1] = make_float4(nd.y, nd.z, __uint_as_float(rng.x), rng.saved);
                    }
                    if (S.n_components == 2) {
                        // the mixture pdf needs the light sum: walk it (beside the child's trace, when there is one)
                        r[4] = make_float4(xo.x, xo.y, xo.z, 0.f);
                        r[5] = make_float4(emission.x, emission.y, emission.z, pdf_cos);
                        r[6] = make_float4(color.x, color.y, color.z, dnn);
                        reinterpret_cast<float *>(r + 3)[3] = __uint_as_float(p6_pack(fp, rng.has_saved, sample, P6_PENDING | (child_traced ? 0u : P6_LIGHT_ONLY)));
                        return P6_LIGHT | (child_traced ? P6_TRACE : 0);
                    }
                    // no lights in the scene: the pdf is complete (Mix = {Cosine})
                    float pdf = 0.f;
                    pdf += pdf_cos;
                    pdf = pdf / S.n_components_f;
                    const float k = (float)(1. / (double)(RT_PI_F * pdf) * (double)dnn);
                    float4 *f = r + 8 + 5 * fp;
                    f[0] = make_float4(emission.x, emission.y, emission.z, __uint_as_float((uint32_t)F6_MUL));
                    const F3 mult = k * color;
                    f[1] = make_float4(mult.x, mult.y, mult.z, 0.f);
                    fp++;
                    if (child_traced) {
                        reinterpret_cast<float *>(r + 3)[3] = __uint_as_float(p6_pack(fp, rng.has_saved, sample, 0u));
                        return P6_TRACE;
                    }
                    returning = true;                                                           // ret = 0: the child beyond the depth limit
                }
            } else {
                const F3 dn = normalize(d);
                const F3 refl = dn - (float)(2. * (double)dot(norma, dn)) * norma;              // scene.cpp:71,75
                float4 *f = r + 8 + 5 * fp;
                f[0] = make_float4(emission.x, emission.y, emission.z, __uint_as_float((uint32_t)(kind == RT_MAT_METALLIC ? F6_MUL : F6_DIEL_REFLECT)));
                f[1] = make_float4(color.x, color.y, color.z, __uint_as_float(inside ? 1u : 0u));
                f[2] = make_float4(x.x, x.y, x.z, m0.w);
                f[3] = make_float4(dn.x, dn.y, dn.z, 0.f);
                f[4] = make_float4(norma.x, norma.y, norma.z, 0.f);
                fp++;
                if (fp < R.ray_depth) {
                    const F3 no = x + epsf * refl;                                              // scene.cpp:72,76
                    r[0] = make_float4(no.x, no.y, no.z, refl.x);
                    r[1] = make_float4(refl.y, refl.z, __uint_as_float(rng.x), rng.saved);
                    reinterpret_cast<float *>(r + 3)[3] = __uint_as_float(p6_pack(fp, rng.has_saved, sample, 0u));
                    return P6_TRACE;
                }
                returning = true;                                                               // recLimit == 0: the child returns 0
            }
        }
    }
    // unwind: `ret` is the value of the call at level fp (scene.cpp:69,73,77-103)
    for (;;) {
        if (fp == 0) { // the camera sample is complete (scene.cpp:111-117)
            const float4 q3 = r[3];
            const F3 accum = f3(q3.x, q3.y, q3.z) + ret;
            sample++;
            int px, py; bool in_image; size_t out_index;
            wf_slot_to_pixel(R, slot + W.slot_base, px, py, in_image, out_index);
            if (sample < (uint32_t)R.samples) {
                F3 o, d;
                p6_camera_ray(S, R, rng, px, py, o, d);
                r[0] = make_float4(o.x, o.y, o.z, d.x);
                r[1] = make_float4(d.y, d.z, __uint_as_float(rng.x), rng.saved);
                r[3] = make_float4(accum.x, accum.y, accum.z, __uint_as_float(p6_pack(0, rng.has_saved, sample, 0u)));
                return sample < (uint32_t)R.sample_stop ? P6_TRACE : P6_PARKED;
            }
            if (R.streams > 1) { // throughput mode (rt_wavefront.h): this stream's unnormalised sum; wf_reduce_streams_kernel adds a pixel's streams
                float *ps = R.partial + 3 * (size_t)(slot + W.slot_base);
                ps[0] = accum.x; ps[1] = accum.y; ps[2] = accum.z;
                return 0;
            }
            const F3 pxl = R.inv_samples * accum;                                               // scene.cpp:115
            if (R.out_rgb) { R.out_rgb[3 * out_index] = pxl.x; R.out_rgb[3 * out_index + 1] = pxl.y; R.out_rgb[3 * out_index + 2] = pxl.z; }
            if (R.out_rgb8) { R.out_rgb8[3 * out_index] = tonemap1(pxl.x); R.out_rgb8[3 * out_index + 1] = tonemap1(pxl.y); R.out_rgb8[3 * out_index + 2] = tonemap1(pxl.z); }
            return 0;
        }
        float4 *f = r + 8 + 5 * (--fp);
        const float4 f0 = f[0], f1 = f[1];
        const int fkind = (int)__float_as_uint(f0.w);
        const F3 emission = f3(f0.x, f0.y, f0.z), fmult = f3(f1.x, f1.y, f1.z);
        if (fkind == F6_MUL) { ret = emission + fmult * ret; continue; }                        // scene.cpp:69,73
        const bool finside = __float_as_uint(f1.w) != 0;
        if (fkind == F6_DIEL_REFRACT) {                                                         // scene.cpp:99-103
            F3 refracted = ret;
            if (!finside) refracted = refracted * fmult;
            ret = emission + refracted;
            continue;
        }
        // F6_DIEL_REFLECT: `ret` is reflectedColor (scene.cpp:77-98)
        const float4 f2 = f[2], f3q = f[3], f4 = f[4];
        const F3 fx = f3(f2.x, f2.y, f2.z), fdn = f3(f3q.x, f3q.y, f3q.z), fnorma = f3(f4.x, f4.y, f4.z);
        float eta1 = 1.f, eta2 = f2.w;
        if (finside) { float tmp = eta1; eta1 = eta2; eta2 = tmp; }
        const F3 l = neg(fdn);
        const float nl = dot(fnorma, l);
        const float sinTheta2 = (float)((double)(eta1 / eta2) * sqrt((double)(1 - nl * nl)));
        if (fabs((double)sinTheta2) > 1.) { ret = emission + ret; continue; }
        const float rr = (eta1 - eta2) / (eta1 + eta2);
        const float r0 = rr * rr;                                                                // pow(., 2.) == exact square
        const double om = (double)(1 - nl), om2 = om * om;
        const float fres = (float)((double)r0 + (double)(1 - r0) * (om2 * om2 * om));            // pow(., 5.)
        if (rng_u01(rng) < fres) { ret = emission + ret; continue; }
        const float cosTheta2 = sqrtf(1 - sinTheta2 * sinTheta2);
        const F3 refr = (eta1 / eta2) * neg(l) + (eta1 / eta2 * nl - cosTheta2) * fnorma;
        reinterpret_cast<float *>(f)[3] = __uint_as_float((uint32_t)F6_DIEL_REFRACT);
        fp++;
        if (fp < R.ray_depth) {
            const F3 no = fx + epsf * refr;
            r[0] = make_float4(no.x, no.y, no.z, refr.x);
            r[1] = make_float4(refr.y, refr.z, __uint_as_float(rng.x), rng.saved);
            reinterpret_cast<float *>(r + 3)[3] = __uint_as_float(p6_pack(fp, rng.has_saved, sample, 0u));
            return P6_TRACE;
        }
        ret = f3(0.f, 0.f, 0.f);                                                                 // the refracted child beyond the depth limit
    }
}

// ---- closest-hit walker over hw6's own tree (closest_hit6 with lane refill) -------------------------------------------------------
template <bool COUNT>
RT_DEV void p6_trace_stint(const SceneView6 &S, const W6View &W, P6Shared &sh, const PtParams &P, PtWave &wv, uint32_t (*stack)[64],
                           const int shade_thr, uint32_t &n_queries, unsigned long long &n_nodes, unsigned long long &n_tris) {
    const int lane = threadIdx.x & 63;
    bool active = false, refill_ok = true;
    uint32_t l = 0, slot = 0, cur = 0, hit = 0xFFFFFFFFu, best_ref = 0xFFFFFFFFu, fin = PT_NONE;
    bool best_inside = false;
    int sp = 0;
    uint32_t steps = 0; // node steps + triangle tests of the lane's walk: the cost measure of the re-deal
    uint32_t pend = RT_EMPTY_LEAF, pend2 = RT_EMPTY_LEAF; // the leaves this lane has met and not yet tested (pend first)
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    RayGrid ray = RT_GRID_RAY_IDLE; // idle lanes: never used
    float best_t = RT_T_MAX, cull_t = RT_T_MAX, t2 = 2.f * RT_T_MAX, h_ray = 0.f; // look-behind and runner-up: rt_exact.h
    for (;;) {
        const unsigned long long idle = pt_ballot(!active);
        if (idle && (__popcll(idle) >= (P.refill & 0xFFFF) || idle == ~0ull)) {
            if (pt_ballot(fin != PT_NONE)) { // hand-off point (rt_persistent.h)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                pt_complete(sh, fin, PT_BIT_T, fin != PT_NONE);
                fin = PT_NONE;
            }
            if (!refill_ok) {}
            else if (pt_count(&sh.cnt[PT_Q_SHADE]) >= shade_thr) refill_ok = false;
            else if (pt_count(&sh.cnt[PT_Q_TRACE]) > 0) {
                const uint32_t got = pt_pop(sh.need[PT_Q_TRACE], &sh.cnt[PT_Q_TRACE], wv.nw, wv.cur[PT_Q_TRACE], !active, wv.front_first);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                n_queries += __popcll(pt_ballot(got != PT_NONE));
                if (got != PT_NONE) {
                    l = got; slot = pt_slot(sh, l);
                    const float4 *r = p6_rec(W, slot);
                    float4 q0 = r[0], q1 = r[1];
                    o = f3(q0.x, q0.y, q0.z); d = f3(q0.w, q1.x, q1.y);
                    ray = make_ray_grid(S.grid, o, d);
                    h_ray = S.exact_boxes ? pt_look_behind_abs(d, S.box_c2x) : 0.f;
                    steps = 0;
                    cur = 0; sp = 0; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; hit = 0xFFFFFFFFu; best_ref = 0xFFFFFFFFu; best_t = RT_T_MAX; cull_t = RT_T_MAX; t2 = 2.f * RT_T_MAX; best_inside = false;
                    active = true;
                }
            }
        }
        const unsigned long long m_active = pt_ballot(active);
        if (!m_active) break;
        const int lb = pt_leaf_batch(P.leaf_batch, m_active);
        auto done = [&]() {
            p6_rec(W, slot)[2] = make_float4(best_t, __uint_as_float(hit), __uint_as_float(best_inside ? 1u : 0u), t2);
            if (P.group_cost) atomicAdd(&sh.cost[l >> pt_gshift(sh)], steps * P6_COST_TRACE_STEP);
            active = false; fin = l;
        };
        for (;;) { // phase 1: inner nodes; a leaf waits in `pend` for the next leaf phase while the lane walks on (rt_persistent.h, pt_trace_stint)
            if (active && (cur & RT_LEAF_BIT) && pend2 == RT_EMPTY_LEAF && cur != PT_DRAINED) {
                if (pend == RT_EMPTY_LEAF) pend = cur; else pend2 = cur;
                cur = sp == 0 ? PT_DRAINED : stack[--sp][lane];
                if (cur == PT_DRAINED && pend == RT_EMPTY_LEAF) done();
            }
            const bool inner = active && !(cur & RT_LEAF_BIT);
            if (!pt_ballot(inner) || __popcll(pt_ballot(active && (cur & RT_LEAF_BIT))) >= lb) break;
            if (inner) {
                if (COUNT) n_nodes++;
                steps++;
                const int went = pt_wide_step_nearest(S.nodes4, ray, cull_t, stack, lane, sp, P6_STACK, cur);
                if (went == PT_WIDE_FULL) { best_t = PT_T_OVERFLOW; hit = 0u; t2 = PT_T_OVERFLOW; best_inside = false; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; done(); } // the exact role redoes the query (rt_persistent.h)
                else if (went == PT_WIDE_NONE) {
                    if (sp != 0) cur = stack[--sp][lane];
                    else if (pend != RT_EMPTY_LEAF) cur = PT_DRAINED;
                    else done();
                }
            }
        }
        if (active && pend != RT_EMPTY_LEAF) { // phase 2: leaves
            {
                uint32_t i = pend & ~RT_LEAF_BIT, more = pend2;
                for (;;) {
                    Tri6Regs T = load_tri6(S.tris + i);
                    if (COUNT) n_tris++;
                    steps++;
                    float t; bool inside;
                    // reference tie rule: smallest t, equal t -> lowest index in the reference's figure order
                    if (tri6_test_closer(T, o, d, cull_t, t, inside)) {
                        if (t < best_t || (t == best_t && T.ref_index < best_ref)) {
                            t2 = fminf(t2, best_t);
                            best_t = t; best_inside = inside; hit = i; best_ref = T.ref_index;
                            cull_t = S.exact_boxes ? t + fmaxf(S.cull_k * t, h_ray) : t;
                        } else t2 = fminf(t2, t);
                    }
                    if (!T.last) i++;
                    else if (more == RT_EMPTY_LEAF) break;
                    else { i = more & ~RT_LEAF_BIT; more = RT_EMPTY_LEAF; } // the lane's second leaf
                }
            }
            pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF;
            if (cur == PT_DRAINED) done();
        }
    }
}

// ---- light-sum walker over the own tree of the lights: no, one or two hits need no order (x + 0 = x, a + b = b + a); more are
// left in the record and added in the reference's association (light_sum6_associate) by the slow role -----------------------------
template <bool COUNT>
RT_DEV void p6_light_stint(const SceneView6 &S, const W6View &W, P6Shared &sh, const PtParams &P, PtWave &wv, uint32_t (*stack)[64],
                           const int shade_thr, uint32_t &n_queries, unsigned long long &n_nodes, unsigned long long &n_tris) {
    const int lane = threadIdx.x & 63;
    bool active = false, refill_ok = true, many = false, fragile = false; // fragile: a hit at a box boundary, the sum goes to the exact walk
    uint32_t l = 0, slot = 0, cur = 0, fin = PT_NONE, idx0 = 0, idx1 = 0, idx2 = 0, idx3 = 0; // fin: the lane's finished, unpublished path; bit 31 = it goes to the slow role
    int sp = 0, k = 0;
    uint32_t steps = 0, pend = RT_EMPTY_LEAF, pend2 = RT_EMPTY_LEAF; // pend, pend2: the leaves this lane has met and not yet tested
    float term0 = 0.f, term1 = 0.f, term2 = 0.f, term3 = 0.f;
    F3 o = f3(0.f, 0.f, 0.f), d = f3(0.f, 0.f, 1.f);
    RayGrid ray = RT_GRID_RAY_IDLE; // idle lanes: never used
    // where the lights x < y of the reference order separate in the reference's light tree (SceneView6::light_sep, see p6_merge_hits)
    auto sep = [&](uint32_t x, uint32_t y) {
        const uint32_t lv = 31u - (uint32_t)__clz((int)(y - x));
        const uint32_t row = lv * S.n_lights; // 32-bit: the table has fewer than 2^32 entries (levels x lights)
        const uint16_t m0 = S.light_sep[row + x], m1 = S.light_sep[row + (y - (1u << lv))];
        return (uint32_t)(m0 < m1 ? m0 : m1);
    };
    auto finish = [&]() {
        active = false;
        if (P.group_cost) atomicAdd(&sh.cost[l >> pt_gshift(sh)], steps * P6_COST_LIGHT_STEP);
        if (fragile) { // the hits go along (also when there are fewer than five), the slow role adds them with the reference's box tests
            float2 *h = reinterpret_cast<float2 *>(p6_rec(W, slot) + 48);
            if (k >= 1 && k <= 4) h[0] = make_float2(__uint_as_float(idx0), term0);
            if (k >= 2 && k <= 4) h[1] = make_float2(__uint_as_float(idx1), term1);
            if (k >= 3 && k <= 4) h[2] = make_float2(__uint_as_float(idx2), term2);
            if (k == 4) h[3] = make_float2(__uint_as_float(idx3), term3);
            reinterpret_cast<uint32_t *>(p6_rec(W, slot) + 7)[0] = (uint32_t)k | 0x80000000u; fin = l | 0x80000000u; return;
        }
        if (many) { reinterpret_cast<uint32_t *>(p6_rec(W, slot) + 7)[0] = (uint32_t)k; fin = l | 0x80000000u; return; }
        float v = k == 0 ? 0.f : (k == 1 ? term0 : term0 + term1);
        if (k == 3) {
            // three hits (the commonest case beyond two) are added here and now, in the reference's association: sorted by reference index
            // a < b < c, the pair that separates deeper in the reference tree is added first — (a + b) + c when sep(a,b) > sep(b,c), else
            // a + (b + c): p6_merge_hits for k = 3 without the trip through the slow role's queue
            uint32_t ia = idx0, ib = idx1, ic = idx2; float ta = term0, tb = term1, tc = term2;
            if (ia > ib) { const uint32_t ti = ia; ia = ib; ib = ti; const float tt = ta; ta = tb; tb = tt; }
            if (ib > ic) { const uint32_t ti = ib; ib = ic; ic = ti; const float tt = tb; tb = tc; tc = tt; }
            if (ia > ib) { const uint32_t ti = ia; ia = ib; ib = ti; const float tt = ta; ta = tb; tb = tt; }
            v = sep(ia, ib) > sep(ib, ic) ? (ta + tb) + tc : ta + (tb + tc);
        }
        if (k == 4) {
            // four hits — a ray through two closed shells of light triangles: 61 % of the sums with more than two hits on practice6_2 — the
            // same way: sorted a < b < c < e with separation depths d1, d2, d3, the operator-precedence evaluation of p6_merge_hits
            // (a deeper separation binds first, equal depths associate to the right) spelled out in its five outcomes
            uint32_t i0 = idx0, i1 = idx1, i2 = idx2, i3 = idx3; float t0 = term0, t1 = term1, t2 = term2, t3 = term3;
#define P6_CSWAP(ia, ib, ta, tb) if (ia > ib) { const uint32_t ti_ = ia; ia = ib; ib = ti_; const float tt_ = ta; ta = tb; tb = tt_; }
            P6_CSWAP(i0, i1, t0, t1) P6_CSWAP(i2, i3, t2, t3) P6_CSWAP(i0, i2, t0, t2) P6_CSWAP(i1, i3, t1, t3) P6_CSWAP(i1, i2, t1, t2)
#undef P6_CSWAP
            const uint32_t d1 = sep(i0, i1), d2 = sep(i1, i2), d3 = sep(i2, i3);
            if (d1 > d2) v = d2 > d3 ? ((t0 + t1) + t2) + t3 : (t0 + t1) + (t2 + t3);
            else if (d2 > d3) v = d1 > d3 ? (t0 + (t1 + t2)) + t3 : t0 + ((t1 + t2) + t3);
            else v = t0 + (t1 + (t2 + t3));
        }
        reinterpret_cast<float *>(p6_rec(W, slot) + 4)[3] = v;
        fin = l;
    };
    for (;;) {
        const unsigned long long idle = pt_ballot(!active);
        if (idle && (__popcll(idle) >= (P.refill >> 16) || idle == ~0ull)) {
            if (pt_ballot(fin != PT_NONE)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                const bool slow = fin != PT_NONE && (fin >> 31) != 0u;
                pt_complete(sh, fin, PT_BIT_L, fin != PT_NONE && !slow);
                pt_push(sh, P6_Q_SLOW, fin & 0x7FFFFFFFu, slow);
                fin = PT_NONE;
            }
            if (!refill_ok) {}
            else if (pt_count(&sh.cnt[PT_Q_SHADE]) >= shade_thr) refill_ok = false;
            else if (pt_count(&sh.cnt[PT_Q_LIGHT]) > 0) {
                const uint32_t got = pt_pop(sh.need[PT_Q_LIGHT], &sh.cnt[PT_Q_LIGHT], wv.nw, wv.cur[PT_Q_LIGHT], !active, wv.front_first);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                n_queries += __popcll(pt_ballot(got != PT_NONE));
                if (got != PT_NONE) {
                    l = got; slot = pt_slot(sh, l);
                    const float4 *r = p6_rec(W, slot);
                    float4 q0 = r[0], q1 = r[1], q4 = r[4];
                    o = f3(q4.x, q4.y, q4.z); d = f3(q0.w, q1.x, q1.y);                        // the pdf's ray: x + eps*n towards the sampled direction
                    ray = make_ray_grid(S.grid, o, d);
                    steps = 0;
                    cur = 0; sp = 0; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; k = 0; many = false; fragile = false; term0 = 0.f; term1 = 0.f;
                    active = true;
                }
            }
        }
        const unsigned long long m_active = pt_ballot(active);
        if (!m_active) break;
        const int lb = pt_leaf_batch(P.leaf_batch, m_active);
        for (;;) { // phase 1: inner nodes; a leaf waits in `pend` for the next leaf phase (rt_persistent.h, pt_trace_stint)
            if (active && (cur & RT_LEAF_BIT) && pend2 == RT_EMPTY_LEAF && cur != PT_DRAINED) {
                if (pend == RT_EMPTY_LEAF) pend = cur; else pend2 = cur;
                cur = sp == 0 ? PT_DRAINED : stack[--sp][lane];
                if (cur == PT_DRAINED && pend == RT_EMPTY_LEAF) finish();
            }
            const bool inner = active && !(cur & RT_LEAF_BIT);
            if (!pt_ballot(inner) || __popcll(pt_ballot(active && (cur & RT_LEAF_BIT))) >= lb) break;
            if (inner) {
                if (COUNT) n_nodes++;
                steps++;
                const int went = pt_wide_step_all(S.fast_light_nodes4, ray, stack, lane, sp, P6_STACK, cur);
                if (went == PT_WIDE_FULL) { many = true; fragile = false; k = RT6_MAX_LIGHT_HITS + 1; pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF; finish(); } // the slow role walks the sum in the reference's order
                else if (went == PT_WIDE_NONE) {
                    if (sp != 0) cur = stack[--sp][lane];
                    else if (pend != RT_EMPTY_LEAF) cur = PT_DRAINED;
                    else finish();
                }
            }
        }
        if (active && pend != RT_EMPTY_LEAF) { // phase 2: leaves
            {
                uint32_t i = pend & ~RT_LEAF_BIT, more = pend2;
                // the loop only tests; a hit's pdf term, robustness test and bookkeeping wait until after it (rt_persistent.h, pt_light_stint)
                bool held = false; uint32_t h_i = 0u; float h_t = 0.f; bool h_in = false;
                auto take = [&]() {
                    const Tri6Regs T = load_tri6(S.fast_lights + h_i);
                    const float t = h_t; const bool inside = h_in;
                    F3 yn = normalize(inside ? neg(T.n) : T.n);                          // primitives.cpp:31
                    F3 y = o + t * d;
                    const float term = T.point_prob * len2(o - y) / fabsf(dot(d, yn));    // distributions.h:116-118
                    if (S.exact_boxes) { // is every box of the reference's light tree above this hit passed whatever the rounding? (rt_exact.h)
                        const F3 pb = T.a + T.b, pc = T.a + T.c;
                        const F3 blo = f3(fminf(T.a.x, fminf(pb.x, pc.x)), fminf(T.a.y, fminf(pb.y, pc.y)), fminf(T.a.z, fminf(pb.z, pc.z)));
                        const F3 bhi = f3(fmaxf(T.a.x, fmaxf(pb.x, pc.x)), fmaxf(T.a.y, fmaxf(pb.y, pc.y)), fmaxf(T.a.z, fmaxf(pb.z, pc.z)));
                        if (!pt_box_robust<P6_LIGHT_PRETEST>(blo, bhi, y, d, t, S.box_c2)) fragile = true;
                    }
                    if (k == 0) { term0 = term; idx0 = T.ref_index; }
                    else if (k == 1) { term1 = term; idx1 = T.ref_index; }
                    else if (k == 2) { term2 = term; idx2 = T.ref_index; }
                    else if (k == 3) { term3 = term; idx3 = T.ref_index; }
                    else { // five or more: the hits go to the record for the slow role
                        float2 *h = reinterpret_cast<float2 *>(p6_rec(W, slot) + 48);
                        if (k == 4) { h[0] = make_float2(__uint_as_float(idx0), term0); h[1] = make_float2(__uint_as_float(idx1), term1); h[2] = make_float2(__uint_as_float(idx2), term2); h[3] = make_float2(__uint_as_float(idx3), term3); }
                        if (k < RT6_MAX_LIGHT_HITS) h[k] = make_float2(__uint_as_float(T.ref_index), term);
                        many = true;
                    }
                    k++;
                    held = false;
                };
                for (;;) {
                    Tri6Regs T = load_tri6(S.fast_lights + i);
                    if (COUNT) n_tris++;
                    steps++;
                    float t; bool inside;
                    if (tri6_test(T, o, d, t, inside)) {
                        if (held) take(); // another hit in this phase
                        held = true; h_i = i; h_t = t; h_in = inside;
                    }
                    if (!T.last) i++;
                    else if (more == RT_EMPTY_LEAF) break;
                    else { i = more & ~RT_LEAF_BIT; more = RT_EMPTY_LEAF; } // the lane's second leaf
                }
                if (held) take();
            }
            pend = RT_EMPTY_LEAF; pend2 = RT_EMPTY_LEAF;
            if (cur == PT_DRAINED || k > RT6_MAX_LIGHT_HITS) finish();
        }
    }
}

// ---- slow role: three or more hits, added in the reference's association.  sum(node) = sum(left) + sum(right) with a side without
// hits as the additive identity means: of the hits sorted by reference light index, those two neighbouring groups are added first
// that separate deepest in the reference tree (SceneView6::light_sep; inside a leaf the pseudo depths give ((a + b) + c)).  That is
// an operator-precedence evaluation with the separation depth as the precedence, done in place in the lane's LDS stack column:
// words 0..15 hold the terms and then the value stack, words 16..31 the light indices and then the depth stack. ---------------------
// vals: k words (terms, then the value stack); ids: k + 1 words (light indices, then the depth stack)
template <class AV, class AI>
RT_DEV float p6_merge_hits_in(const SceneView6 &S, const float2 *h, int k, AV vals, AI ids) {
    for (int i = 0; i < k; i++) { // insertion sort by the reference's light index
        const float2 e = h[i];
        const uint32_t id = __float_as_uint(e.x);
        int j = i - 1;
        while (j >= 0 && ids[j] > id) { ids[j + 1] = ids[j]; vals[j + 1] = vals[j]; j--; }
        ids[j + 1] = id; vals[j + 1] = __float_as_uint(e.y);
    }
    const uint32_t nl = S.n_lights;
    int vs = 1, os = 0;                                   // value stack: vals[0..vs-1]; depth stack: ids[0..os-1]
    uint32_t prev = ids[0];
    for (int i = 1; i < k; i++) {
        const uint32_t id = ids[i];
        const float term = __uint_as_float(vals[i]);
        const uint32_t len = id - prev, lv = 31u - (uint32_t)__clz((int)len);
        const uint32_t row = lv * nl;
        const uint16_t m0 = S.light_sep[row + prev], m1 = S.light_sep[row + (id - (1u << lv))];
        const uint32_t depth = m0 < m1 ? m0 : m1;          // where the hits prev and id separate
        while (os > 0 && ids[os - 1] > depth) {            // the groups on the stack that separate deeper are complete: fold them
            const float b = __uint_as_float(vals[vs - 1]), a = __uint_as_float(vals[vs - 2]);
            vals[vs - 2] = __float_as_uint(a + b);
            vs--; os--;
        }
        ids[os] = depth; os++;                             // os <= i - 1 < the index words already consumed: never overwrites an unread index
        vals[vs] = __float_as_uint(term); vs++;
        prev = id;
    }
    while (os > 0) {
        const float b = __uint_as_float(vals[vs - 1]), a = __uint_as_float(vals[vs - 2]);
        vals[vs - 2] = __float_as_uint(a + b);
        vs--; os--;
    }
    return __uint_as_float(vals[0]);
}
struct P6Column { // a lane's own stack column, words first .. : indexable
    uint32_t (*col)[64]; int first, lane;
    RT_DEV uint32_t &operator[](int i) const { return col[first + i][lane]; }
};
RT_DEV float p6_merge_hits(const SceneView6 &S, const float2 *h, int k, uint32_t (*col)[64]) { // k <= P6_MERGE_HITS, in the lane's own column
    P6Column vals, ids;
    vals.col = col; vals.first = 0; vals.lane = threadIdx.x & 63;
    ids.col = col; ids.first = P6_MERGE_HITS; ids.lane = vals.lane;
    return p6_merge_hits_in(S, h, k, vals, ids);
}

// ---- exact role: BVH::intersect_ of hw6 (bvh.h, identical to hw8's) over the reference's own tree with the reference's box test, as an
// iterative depth-first walk, left child first, one running best with strict '<' (ref_closest_hit of rt_exact.h with hw6's figures) ----
template <class A>
RT_DEV void ref_closest_hit6(const SceneView6 &S, F3 o, F3 d, A stack, float &best_t, bool &best_inside, uint32_t &hit) {
    best_t = RT_T_MAX; best_inside = false; hit = 0xFFFFFFFFu;
    if (S.n_tris == 0) return;
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        const RefNodeView n = load_ref_node(S.ref_nodes + cur);
        float tb; bool inside;
        if (ref_box_test(n.mn, n.mx, o, d, tb, inside) && !(hit != 0xFFFFFFFFu && best_t < tb && !inside)) {
            if (n.left == 0) {
                for (uint32_t i = n.first; i < n.last; i++) {
                    const Tri6Regs T = load_tri6(S.ref_tris + i);
                    float t; bool in;
                    if (tri6_test(T, o, d, t, in) && (hit == 0xFFFFFFFFu || t < best_t)) { best_t = t; best_inside = in; hit = i; }
                }
            } else if (sp < RT6_STACK_SIZE) { stack[sp++] = n.right; cur = n.left; continue; }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
}

// The reference's trees are degenerate here (built on a constant sort key: 59 and 85 levels, tens of thousands of box tests per query),
// so the exact walks do not walk them blindly.  A subtree without a hit of the ray is a no-op in the reference's recursion (it returns
// "nothing" / adds 0 and leaves curBest alone), so it is enough to (1) find every figure the ray hits with the library's own tree,
// (2) sort those by their position in the reference's order, (3) run the reference's recursion only along the root-to-leaf paths that
// lead to them -- every node on such a path gets the reference's box test and pruning rule, every hit its triangle test again, in
// the reference's order.  A few hundred box tests instead of tens of thousands; more hits than P6_XHITS fall back to the blind walk.
#define P6_XHITS 48
template <class A, class B>
RT_DEV bool p6_all_hits(const SceneView6 &S, F3 o, F3 d, A stack, B hits, int &k) { // unsorted reference indices; false = too many
    RayInv ray = make_ray_inv(o, d);
    int sp = 0; k = 0;
    uint32_t cur = 0;
    for (;;) {
        if (cur & RT_LEAF_BIT) {
            if (cur != RT_EMPTY_LEAF) {
                uint32_t i = cur & ~RT_LEAF_BIT;
                for (;;) {
                    const Tri6Regs T = load_tri6(S.tris + i);
                    float t; bool inside;
                    if (tri6_test(T, o, d, t, inside)) { if (k == P6_XHITS) return false; hits[k++] = T.ref_index; }
                    if (T.last) break;
                    i++;
                }
            }
            if (sp == 0) return true;
            cur = stack[--sp];
            continue;
        }
        const float4 *q = reinterpret_cast<const float4 *>(S.nodes + cur);
        const float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
        float n0, n1;
        const bool h0 = slab_test(lo0, hi0, ray, RT_T_MAX, n0), h1 = slab_test(lo1, hi1, ray, RT_T_MAX, n1);
        const uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
        if (h0 & h1) { stack[sp++] = c1; cur = c0; }
        else if (h0) cur = c0;
        else if (h1) cur = c1;
        else { if (sp == 0) return true; cur = stack[--sp]; }
    }
}
template <class B>
RT_DEV void p6_sort_hits(B hits, int k) {
    for (int i = 1; i < k; i++) { const uint32_t v = hits[i]; int j = i - 1; while (j >= 0 && hits[j] > v) { hits[j + 1] = hits[j]; j--; } hits[j + 1] = v; }
}
// BVH::intersect_ restricted to the paths towards hits[lo..hi): stack words = node | lo << 19 | hi << 25 (node < 2^19, k <= 63)
template <class A, class B>
RT_DEV void ref_closest_hit6_along(const SceneView6 &S, F3 o, F3 d, B hits, int k, A stack, float &best_t, bool &best_inside, uint32_t &hit) {
    best_t = RT_T_MAX; best_inside = false; hit = 0xFFFFFFFFu;
    if (k == 0) return;
    int sp = 0;
    uint32_t node = 0; int lo = 0, hi = k;
    for (;;) {
        const RefNodeView n = load_ref_node(S.ref_nodes + node);
        float tb; bool inside;
        bool descend = false;
        if (ref_box_test(n.mn, n.mx, o, d, tb, inside) && !(hit != 0xFFFFFFFFu && best_t < tb && !inside)) {
            if (n.left == 0) {
                for (int j = lo; j < hi; j++) {
                    const Tri6Regs T = load_tri6(S.ref_tris + hits[j]);
                    float t; bool in;
                    if (tri6_test(T, o, d, t, in) && (hit == 0xFFFFFFFFu || t < best_t)) { best_t = t; best_inside = in; hit = hits[j]; }
                }
            } else {
                const uint32_t right_first = S.ref_nodes[n.right].first;
                int m = lo;
                while (m < hi && hits[m] < right_first) m++;
                if (m > lo) { // the left child leads to hits: go there, the right one (if it does too) waits
                    if (m < hi) stack[sp++] = n.right | ((uint32_t)m << 19) | ((uint32_t)hi << 25);
                    node = n.left; hi = m; descend = true;
                } else { node = n.right; descend = true; }
            }
        }
        if (descend) continue;
        if (sp == 0) return;
        const uint32_t w = stack[--sp];
        node = w & 0x7FFFFu; lo = (int)((w >> 19) & 63u); hi = (int)(w >> 25);
    }
}
// FiguresMix::getTotalPdf restricted to the paths towards the hit lights (sorted by index): the addition tree of light_sum6_associate
// with the reference's box test at every node on the way (a failed box contributes 0 whatever lies below it).
// hit_idx / hit_term: RT6_MAX_LIGHT_HITS words each; `work`: 4 x RT6_MAX_LIGHT_HITS words for the frames of the recursion.
template <class A, class AF>
RT_DEV float ref_light_pdf_sum6_along(const SceneView6 &S, F3 x, F3 d, A hit_idx, AF hit_term, int k, A f_node, A f_lo, A f_hi, AF f_val) {
    if (k == 0) return 0.f;
    for (int i = 1; i < k; i++) {
        uint32_t id = hit_idx[i]; float tm = hit_term[i];
        int j = i - 1;
        while (j >= 0 && hit_idx[j] > id) { hit_idx[j + 1] = hit_idx[j]; hit_term[j + 1] = hit_term[j]; j--; }
        hit_idx[j + 1] = id; hit_term[j + 1] = tm;
    }
    uint32_t f_add = 0;
    int fsp = 0;
    uint32_t node = 0; int lo = 0, hi = k;
    float v = 0.f;
    for (;;) {
        for (;;) { // total of hits [lo, hi) under `node`
            const RefNodeView n = load_ref_node(S.ref_light_nodes + node);
            float tb; bool inside;
            if (!ref_box_test(n.mn, n.mx, x, d, tb, inside)) { v = 0.f; break; }
            if (n.left == 0) { v = 0.f; for (int j = lo; j < hi; j++) v += hit_term[j]; break; }
            const uint32_t right_first = S.ref_light_nodes[n.right].first;
            int m = lo;
            while (m < hi && hit_idx[m] < right_first) m++;
            if (m == lo) { node = n.right; continue; }
            if (m == hi) { node = n.left; continue; }
            f_node[fsp] = n.right; f_lo[fsp] = (uint32_t)m; f_hi[fsp] = (uint32_t)hi; f_add &= ~(1u << fsp); fsp++;
            node = n.left; hi = m;
        }
        for (;;) {
            if (fsp == 0) return v;
            fsp--;
            if ((f_add >> fsp) & 1u) { v = f_val[fsp] + v; continue; }
            node = f_node[fsp]; lo = (int)f_lo[fsp]; hi = (int)f_hi[fsp];
            f_val[fsp] = v; f_add |= 1u << fsp; fsp++;
            break;
        }
    }
}

// FiguresMix::getTotalPdf of hw6 (distributions.h:212-256) over the reference's own light tree with the reference's box test and its
// association of the additions (ref_light_pdf_sum of rt_exact.h with hw6's term).  The tree is degenerate (85 levels on practice6_2):
// thousands of box tests per query, which is why only light sums with a hit at a box boundary come here.
template <class A>
RT_DEV float ref_light_pdf_sum6(const SceneView6 &S, F3 x, F3 d, A stack) {
    int sp = 0;
    unsigned long long mask_lo = 0, mask_hi = 0;
    uint32_t cur = 0;
    bool descending = true;
    float v = 0.f;
    if (S.n_lights == 0) return 0.f;
    for (;;) {
        if (descending) {
            const RefNodeView n = load_ref_node(S.ref_light_nodes + cur);
            float tb; bool inside;
            if (!ref_box_test(n.mn, n.mx, x, d, tb, inside)) { v = 0.f; descending = false; }
            else if (n.left == 0) {
                float result = 0.f;
                for (uint32_t i = n.first; i < n.last; i++) {
                    const Tri6Regs T = load_tri6(S.lights + i);
                    float t; bool in; float term = 0.f;
                    if (tri6_test(T, x, d, t, in)) {
                        const F3 yn = normalize(in ? neg(T.n) : T.n);
                        const F3 y = x + t * d;
                        term = T.point_prob * len2(x - y) / fabsf(dot(d, yn));
                    }
                    result += term;
                }
                v = result; descending = false;
            } else if (sp < RT6_STACK_SIZE) {
                if (sp < 64) mask_lo &= ~(1ull << sp); else mask_hi &= ~(1ull << (sp - 64));
                stack[sp++] = n.right; cur = n.left;
            } else { v = 0.f; descending = false; }
        } else {
            if (sp == 0) break;
            --sp;
            const uint32_t f = stack[sp];
            const bool is_add = sp < 64 ? ((mask_lo >> sp) & 1ull) != 0 : ((mask_hi >> (sp - 64)) & 1ull) != 0;
            if (is_add) v = __uint_as_float(f) + v;
            else {
                if (sp < 64) mask_lo |= 1ull << sp; else mask_hi |= 1ull << (sp - 64);
                stack[sp++] = __float_as_uint(v); cur = f; descending = true;
            }
        }
    }
    return v;
}

// ---- the rare roles, P6_XBATCH queries at a time, work arrays in the wave's LDS stack area (P6Slice) ----------------------------------
// Light sums the walker could not finish: a hit at a box boundary (the reference's own box tests decide, along the paths to the hits),
// or more hits than p6_merge_hits takes (the plain reference-order walk).
template <class SH>
RT_DEV void p6_slow_batch(const SceneView6 &S, const W6View &W, SH &sh, PtWave &wv, P6Lds area, uint32_t got, bool mine, uint32_t &n_xlight) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = 0; base < 64u; base += P6_XBATCH) { // the wave's lanes take turns at the slices, eight at a time
        if (!pt_ballot(mine && lane >= base && lane < base + P6_XBATCH)) continue;
        bool boundary = false;
        if (mine && lane >= base && lane < base + P6_XBATCH) {
            P6Slice sl; sl.p = area + (lane - base);
            float4 *r = p6_rec(W, pt_slot(sh, got));
            const uint32_t kw = reinterpret_cast<const uint32_t *>(r + 7)[0];
            const int k = (int)(kw & 0x7FFFFFFFu);
            const float4 q0 = r[0], q1 = r[1], q4 = r[4];
            const F3 lx = f3(q4.x, q4.y, q4.z), ld = f3(q0.w, q1.x, q1.y);
            float v;
            boundary = (kw >> 31) != 0u;
            if (kw >> 31) { // a hit at a box boundary
                if (k > RT6_MAX_LIGHT_HITS) v = ref_light_pdf_sum6(S, lx, ld, sl);
                else {
                    P6Slice hit_idx = sl.at(0), f_node = sl.at(32), f_lo = sl.at(48), f_hi = sl.at(64);
                    P6SliceF hit_term; hit_term.p = sl.at(16).p;
                    P6SliceF f_val; f_val.p = sl.at(80).p;
                    const float2 *h = reinterpret_cast<const float2 *>(r + 48);
                    for (int i = 0; i < k; i++) { const float2 e = h[i]; hit_idx[i] = __float_as_uint(e.x); hit_term[i] = e.y; }
                    v = ref_light_pdf_sum6_along(S, lx, ld, hit_idx, hit_term, k, f_node, f_lo, f_hi, f_val);
                }
            } else if (k <= RT6_MAX_LIGHT_HITS) v = p6_merge_hits_in(S, reinterpret_cast<const float2 *>(r + 48), k, sl.at(0), sl.at(RT6_MAX_LIGHT_HITS)); // more hits than a column takes
            else v = light_pdf_sum6(S, lx, ld, sl);       // more hits than the record holds: the plain reference-order walk
            reinterpret_cast<float *>(r + 4)[3] = v;
        }
        n_xlight += (uint32_t)__popcll(pt_ballot(boundary)); // wave-uniform
    }
}

// Closest hits that do not stand as the reference's answer (~1e-4 of them): the reference's own walk.
template <class SH>
RT_DEV void p6_exact_batch(const SceneView6 &S, const W6View &W, SH &sh, PtWave &wv, P6Lds area, uint32_t &n_exact) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t got = pt_pop(sh.need[P6_Q_XTRACE], &sh.cnt[P6_Q_XTRACE], wv.nw, wv.cur[P6_Q_XTRACE], lane < P6_XBATCH);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (got != PT_NONE) {
        P6Slice xstack; xstack.p = area + lane;             // words 0..127: node stack; words 128..175: the ray's hits
        P6Slice xhits = xstack.at(RT6_STACK_SIZE);
        float4 *r = p6_rec(W, pt_slot(sh, got));
        const float4 q0 = r[0], q1 = r[1];
        float bt; bool bin; uint32_t bhit;
        int xk;
        const F3 xo = f3(q0.x, q0.y, q0.z), xd = f3(q0.w, q1.x, q1.y);
        if (S.n_tris < (1u << 18) /* the path stack packs node numbers into 19 bits */ && p6_all_hits(S, xo, xd, xstack, xhits, xk)) { p6_sort_hits(xhits, xk); ref_closest_hit6_along(S, xo, xd, xhits, xk, xstack, bt, bin, bhit); }
        else ref_closest_hit6(S, xo, xd, xstack, bt, bin, bhit);
        r[2] = make_float4(bt, __uint_as_float(bhit), __uint_as_float(bin ? 1u : 0u), 0.f);
        float *pk = reinterpret_cast<float *>(r + 3) + 3;
        *pk = __uint_as_float(__float_as_uint(*pk) | P6_VERIFIED);
    }
    n_exact += __popcll(pt_ballot(got != PT_NONE));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    pt_push(sh, PT_Q_SHADE, got, got != PT_NONE);
}
static_assert(sizeof(P6Shared) <= 25 * 1280, "five workgroups per CU: 25 LDS granules of 1,280 bytes each");
static_assert(RT6_STACK_SIZE + P6_XHITS <= P6_SLICE_WORDS && 6 * RT6_MAX_LIGHT_HITS <= P6_SLICE_WORDS, "the rare roles' work arrays must fit a slice of the wave's stack area");

// ---- the kernel (scheduler of rt_persistent.h) ------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(P6_THREADS, P6_PER_CU) void p6_persistent_kernel(SceneView6 S, RenderView R, W6View W, PtParams P) {
    __shared__ P6Shared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    PtWave wv;
    wv.n_blocks = gridDim.x; wv.block = blockIdx.x;
    wv.front_first = P.front_first != 0u;
    const uint32_t first_group = P.group_ofs ? P.group_ofs[wv.block] : 0u;
    const uint32_t n_local_groups = P.group_ofs ? P.group_ofs[wv.block + 1u] - first_group
                                                : (P.n_groups > wv.block ? (P.n_groups - wv.block + wv.n_blocks - 1u) / wv.n_blocks : 0u);
    wv.n_local = n_local_groups << P.group_shift;
    wv.nw = (wv.n_local + 31u) >> 5;
    if (wv.n_local == 0u) return;
    for (int q = 0; q < 5; q++) wv.cur[q] = (wave * 64u) % wv.nw;
    for (uint32_t i = tid; i < wv.nw; i += P6_THREADS) { sh.need[0][i] = 0; sh.need[1][i] = 0; sh.need[2][i] = 0; sh.need[3][i] = 0; sh.need[4][i] = 0; }
    for (uint32_t i = tid; i < 2u * wv.nw; i += P6_THREADS) sh.pending[i] = 0;
    for (uint32_t i = tid; i < n_local_groups; i += P6_THREADS) { sh.groups[i] = P.group_ofs ? P.group_ids[first_group + i] : i * wv.n_blocks + wv.block; sh.cost[i] = 0; }
    if (tid < 16u) sh.cnt[tid] = tid == PT_GSHIFT ? (int)P.group_shift : 0;
    if (P.debug && tid == 0) { P.debug[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); P.debug[3 * blockIdx.x + 2] = wv.n_local; }
    __syncthreads();
    for (uint32_t base = 0; base < wv.n_local; base += P6_THREADS) { // seed every pixel, first camera ray (hw6/src/sceneio.cpp:281-284)
        const uint32_t l = base + tid;
        bool started = false;
        if (l < wv.n_local) {
            const uint32_t slot = pt_slot(sh, l), gslot = slot + W.slot_base;
            int x, y; bool inside; size_t out_index;
            wf_slot_to_pixel(R, gslot, x, y, inside, out_index);
            if (!inside) {
                if (R.shard_count > 1 && (R.streams <= 1 || gslot < R.n_pixslots)) { // padding of a border tile in the compact shard layout
                    if (R.out_rgb) { R.out_rgb[3 * out_index] = 0.f; R.out_rgb[3 * out_index + 1] = 0.f; R.out_rgb[3 * out_index + 2] = 0.f; }
                    if (R.out_rgb8) { R.out_rgb8[3 * out_index] = 0; R.out_rgb8[3 * out_index + 1] = 0; R.out_rgb8[3 * out_index + 2] = 0; }
                }
            } else if (P.resume) {
                // a later phase of the frame: the record holds the pixel sum, the random stream and the parked camera ray
                const uint32_t packed = __float_as_uint(reinterpret_cast<const float *>(p6_rec(W, slot) + 3)[3]);
                started = (packed >> 8) < (uint32_t)R.samples;
                if (started) atomicOr(&sh.pending[l >> 4], PT_BIT_T << ((l & 15u) * 2u));
            } else {
                Rng rng;
                rng_seed(rng, (uint32_t)(y * R.width + x) + (R.streams > 1 ? (gslot / R.n_pixslots) * R.seed_stride : 0u)); // hw6/src/sceneio.cpp:280-284; throughput mode: stream k offset by k * W * H
                F3 o, d;
                p6_camera_ray(S, R, rng, x, y, o, d);
                float4 *r = p6_rec(W, slot);
                r[0] = make_float4(o.x, o.y, o.z, d.x);
                r[1] = make_float4(d.y, d.z, __uint_as_float(rng.x), rng.saved);
                r[2] = make_float4(0.f, 0.f, 0.f, 0.f);
                r[3] = make_float4(0.f, 0.f, 0.f, __uint_as_float(p6_pack(0, rng.has_saved, 0u, 0u)));
                atomicOr(&sh.pending[l >> 4], PT_BIT_T << ((l & 15u) * 2u));
                started = true;
            }
        }
        const unsigned long long m = pt_ballot(started);
        if (m && lane == 0) atomicAdd(&sh.cnt[PT_N_LIVE], (int)__popcll(m));
        pt_push(sh, PT_Q_TRACE, l, started);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    uint32_t(*stack)[64] = sh.stack[wave];
    const int shade_thr = P.shade_thr0 + (int)wave * P.shade_thr_step;
    uint32_t n_closest = 0, n_light = 0, n_slow = 0, n_exact = 0, n_xlight = 0;
    unsigned long long n_nodes = 0, n_tris = 0;
    uint32_t idle_spins = 0;
    int gave_up = 0; // 1: the launch ran into its deadline; 2: the workgroup waited in vain for a path to come back (a lost path: a bug)
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_role[5] = {0, 0, 0, 0, 0}, t_mark = t_start; // COUNT: wave time as closest-hit walker, light walker, shader, slow light sums, idle
    auto clock_role = [&](int role) { if (COUNT) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t_role[role] += now - t_mark; t_mark = now; } };
    for (;;) {
        if (__builtin_amdgcn_s_memrealtime() - t_start > P.deadline_ticks) { gave_up = 1; break; } // safety net: never hang the GPU; the host reports the error
        const int ns = pt_count(&sh.cnt[PT_Q_SHADE]), nt = pt_count(&sh.cnt[PT_Q_TRACE]), nl = pt_count(&sh.cnt[PT_Q_LIGHT]);
        if (pt_count(&sh.cnt[P6_Q_SLOW]) > 0) {
            // light sums with more than two hits: the reference's association over the hits the walker left in the record, one lane per
            // query in its own stack column (up to P6_MERGE_HITS hits, not at a box boundary); the others in batches (p6_slow_batch)
            const uint32_t got = pt_pop(sh.need[P6_Q_SLOW], &sh.cnt[P6_Q_SLOW], wv.nw, wv.cur[P6_Q_SLOW], true);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool batch = false;
            if (got != PT_NONE) {
                float4 *r = p6_rec(W, pt_slot(sh, got));
                const uint32_t kw = reinterpret_cast<const uint32_t *>(r + 7)[0];
                if (COUNT && P.counters) atomicAdd(&P.counters[32 + ((kw & 0x7FFFFFFFu) < 15u ? (kw & 0x7FFFFFFFu) : 15u)], 1ull); // histogram of the hit counts that reach the slow role
                if ((kw >> 31) || kw > (uint32_t)P6_MERGE_HITS) batch = true;
                else reinterpret_cast<float *>(r + 4)[3] = p6_merge_hits(S, reinterpret_cast<const float2 *>(r + 48), (int)kw, stack);
            }
            if (pt_ballot(batch)) p6_slow_batch(S, W, sh, wv, (P6Lds)&stack[0][0], got, batch, n_xlight); // the merges are done: their columns are free
            n_slow += __popcll(pt_ballot(got != PT_NONE));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            pt_complete(sh, got, PT_BIT_L, got != PT_NONE);
            idle_spins = 0;
            clock_role(3);
            continue;
        }
        if (pt_count(&sh.cnt[P6_Q_XTRACE]) > 0) {
            p6_exact_batch(S, W, sh, wv, (P6Lds)&stack[0][0], n_exact);
            idle_spins = 0;
            clock_role(3);
            continue;
        }
        if (ns >= P.shade_min || (ns > 0 && nt + nl == 0)) {
            const uint32_t got = pt_pop(sh.need[PT_Q_SHADE], &sh.cnt[PT_Q_SHADE], wv.nw, wv.cur[PT_Q_SHADE], true, wv.front_first);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int todo = 0;
            if (got != PT_NONE) todo = p6_advance(S, R, W, pt_slot(sh, got));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            pt_push(sh, P6_Q_XTRACE, got, got != PT_NONE && todo == P6_EXACT);
            const bool tr = got != PT_NONE && todo != P6_EXACT && (todo & P6_TRACE), li = got != PT_NONE && todo != P6_EXACT && (todo & P6_LIGHT);
            if (tr || li) atomicOr(&sh.pending[got >> 4], ((tr ? PT_BIT_T : 0u) | (li ? PT_BIT_L : 0u)) << ((got & 15u) * 2u));
            pt_push(sh, PT_Q_TRACE, got, tr);
            pt_push(sh, PT_Q_LIGHT, got, li);
            if (got != PT_NONE && todo != P6_EXACT) atomicAdd(&sh.cost[got >> pt_gshift(sh)], (uint32_t)P6_COST_SHADE);
            const unsigned long long done = pt_ballot(got != PT_NONE && (todo == 0 || todo == P6_PARKED));
            if (done && lane == 0) atomicSub(&sh.cnt[PT_N_LIVE], (int)__popcll(done));
            idle_spins = 0;
            clock_role(2);
            continue;
        }
        if (nt + nl > 0) {
            const long long wt = (long long)nt * P.cost_t * (pt_count(&sh.cnt[PT_W_LIGHT]) + 1), wl = (long long)nl * P.cost_l * (pt_count(&sh.cnt[PT_W_TRACE]) + 1);
            if (nl == 0 || (nt > 0 && wt >= wl)) {
                if (lane == 0) atomicAdd(&sh.cnt[PT_W_TRACE], 1);
                p6_trace_stint<COUNT>(S, W, sh, P, wv, stack, shade_thr, n_closest, n_nodes, n_tris);
                if (lane == 0) atomicSub(&sh.cnt[PT_W_TRACE], 1);
                clock_role(0);
            } else {
                if (lane == 0) atomicAdd(&sh.cnt[PT_W_LIGHT], 1);
                p6_light_stint<COUNT>(S, W, sh, P, wv, stack, shade_thr, n_light, n_nodes, n_tris);
                if (lane == 0) atomicSub(&sh.cnt[PT_W_LIGHT], 1);
                clock_role(1);
            }
            idle_spins = 0;
            continue;
        }
        if (pt_count(&sh.cnt[PT_N_LIVE]) <= 0) break;
        __builtin_amdgcn_s_sleep(8);
        clock_role(4);
        if (++idle_spins > (1u << 24)) { gave_up = 2; break; }
    }
    if (gave_up && lane == 0 && P.counters) atomicAdd(&P.counters[gave_up == 1 ? 29 : 14], 1ull);
    if (P.group_cost) { // every wave leaves the loop once the workgroup's pixels are done (or at the deadline)
        __syncthreads();
        for (uint32_t i = tid; i < n_local_groups; i += P6_THREADS) P.group_cost[sh.groups[i]] = sh.cost[i];
    }
    if (lane == 0 && P.counters) {
        if (n_closest) atomicAdd(&P.counters[0], (unsigned long long)n_closest);
        if (n_light) atomicAdd(&P.counters[1], (unsigned long long)n_light);
        if (n_slow) atomicAdd(&P.counters[13], (unsigned long long)n_slow);
        if (n_exact) atomicAdd(&P.counters[12], (unsigned long long)n_exact);
    }
    if (lane == 0 && n_xlight && P.counters) atomicAdd(&P.counters[11], (unsigned long long)n_xlight);
    if (COUNT && P.counters) {
        atomicAdd(&P.counters[2], n_nodes); atomicAdd(&P.counters[3], n_tris);
        if (lane == 0) for (int i = 0; i < 5; i++) atomicAdd(&P.counters[16 + i], t_role[i]);
    }
    if (P.debug && lane == 0) atomicMax(&P.debug[3 * blockIdx.x + 1], __builtin_amdgcn_s_memrealtime());
}

} // namespace dev
} // namespace rtamd
