// Kernel for the hw5 snapshot: analytic primitives + TRIANGLE figures with per-figure position/rotation, a BVH over the
// non-plane figures, Mix{Cosine, FiguresMix{box | ellipsoid | triangle lights behind their own BVH}} and one
// minstd_rand(y*W+x) per pixel — the reference's own seeding from this snapshot on, so the HIP path is checked for the
// same pixels as the reference program (hw5/src/scene.cpp:8-126, primitives.cpp:12-222, bvh.h:18-141,
// distributions.h:15-302, sceneio.cpp:103-123).
//
// `eps` is a long double in this snapshot (primitives.h:9): (t + eps) is an 80-bit sum narrowed to float.  The device has
// no 80-bit type; the sum is formed in double, which can differ from the reference in the last float bit only when the
// exact sum lies within ~1e-20 of a float rounding boundary (probability ~1e-9 per evaluation).
#pragma once
#include "rt_types_hw5.h"
#include "rt_kernels_hw4.h"
#include "rt_exact.h"

namespace rtamd {
namespace dev {

#define RT5_STACK 64

struct FigRegs { PrimRegs P; F3 data2, data3; bool last; };
RT_DEV FigRegs load_fig5(const GpuFig5 *p) {
    const float4 *q = reinterpret_cast<const float4 *>(p);
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
    FigRegs F;
    F.P.data = f3(q0.x, q0.y, q0.z); F.P.type = (int)__float_as_uint(q0.w);
    F.P.position = f3(q1.x, q1.y, q1.z); F.P.kind = (int)__float_as_uint(q1.w);
    F.P.rot.v = f3(q2.x, q2.y, q2.z); F.P.rot.w = q2.w;
    F.P.color = f3(q3.x, q3.y, q3.z); F.P.ior = q3.w;
    F.P.emission = f3(q4.x, q4.y, q4.z); F.last = __float_as_uint(q4.w) != 0u;
    F.data2 = f3(q5.x, q5.y, q5.z);
    F.data3 = f3(q6.x, q6.y, q6.z);
    return F;
}

// Figure::intersect (primitives.cpp:13-35); the triangle branch is intersectAsTriangle (:143-166)
RT_DEV bool fig_hit5(const FigRegs &F, F3 o, F3 d, float &t, F3 &norma, bool &inside) {
    if (F.P.type != RT_PRIM_TRIANGLE) return prim_hit<false, true>(F.P, o, d, t, norma, inside);
    F3 to = qtransform(F.P.rot, o - F.P.position), td = qtransform(F.P.rot, d);
    F3 a = F.data3, b = F.P.data - a, c = F.data2 - a;
    F3 n = crossr(b, c);
    F3 oa = to - a;
    float dn = dot(td, n);
    t = -dot(oa, n) / dn;
    if (!(t > 0 && t < 1e4f)) return false;
    inside = dn > 0;
    F3 p = oa + t * td;
    if (dot(crossr(b, p), n) < 0) return false;
    if (dot(crossr(p, c), n) < 0) return false;
    if (dot(crossr(c - b, p - b), n) < 0) return false;
    norma = normalize(qtransform(qconj(F.P.rot), inside ? neg(n) : n));
    return true;
}

struct Hit5 { int idx; float t; F3 n; bool inside; };
// Scene::intersect (scene.cpp:25-45): planes first (strict '<' keeps the first), then the BVH, whose result replaces
// the plane hit only when strictly nearer.
RT_DEV Hit5 closest_hit5(const SceneView5 &S, F3 o, F3 d, uint32_t *stack) {
    Hit5 best; best.idx = -1; best.t = RT_T_MAX; best.n = f3(0.f, 0.f, 0.f); best.inside = false;
    for (uint32_t i = S.n_nonplanes; i < S.n_figs; i++) {
        FigRegs F = load_fig5(S.figs + i);
        float t; F3 n; bool inside;
        if (fig_hit5(F, o, d, t, n, inside) && (best.idx < 0 || t < best.t)) { best.idx = (int)i; best.t = t; best.n = n; best.inside = inside; }
    }
    if (S.n_nonplanes == 0) return best;
    // BVH::intersect_ (bvh.h:111-141) with the reference's own box test on its unpadded boxes (AABB::intersect, primitives.cpp:221-223 ->
    // intersectBoxAndRay, :92-116: ref_box_test of rt_exact.h is the same code), as an iterative left-first walk.  The recursion's
    // `curBest` at a node is the smallest t of the plane hit and of everything found before the node in this order (every level hands
    // its left result on to its right child), so one running value prunes (`curBest < t_box && !inside`); the result is the first
    // figure with the smallest t (a leaf and a parent both replace on strict '<' only), and it replaces the plane hit when strictly nearer.
    // hw5's scenes are a handful of figures: the exact test at every node costs nothing that matters, and no gate is needed.
    bool have_cur = best.idx >= 0;
    float cur_best = best.t;
    Hit5 bvh; bvh.idx = -1; bvh.t = RT_T_MAX; bvh.n = f3(0.f, 0.f, 0.f); bvh.inside = false;
    int sp = 0;
    uint32_t cur = 0;
    for (;;) {
        const RefNodeView n = load_ref_node(S.ref_nodes + cur);
        float tb; bool inside_box;
        if (ref_box_test(n.mn, n.mx, o, d, tb, inside_box) && !(have_cur && cur_best < tb && !inside_box)) {
            if (n.left == 0) {
                for (uint32_t i = n.first; i < n.last; i++) {
                    FigRegs F = load_fig5(S.figs + i);
                    float t; F3 nn; bool inside;
                    if (fig_hit5(F, o, d, t, nn, inside) && (bvh.idx < 0 || t < bvh.t)) { bvh.idx = (int)i; bvh.t = t; bvh.n = nn; bvh.inside = inside; }
                }
                if (bvh.idx >= 0 && (!have_cur || bvh.t < cur_best)) { have_cur = true; cur_best = bvh.t; }
            } else if (sp < RT5_STACK) { stack[sp++] = n.right; cur = n.left; continue; }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    if (bvh.idx >= 0 && (best.idx < 0 || bvh.t < best.t)) best = bvh;
    return best;
}

// pdfOne of the three light kinds (distributions.h:69-71, :117-119, :150-155)
RT_DEV float pdf_one5(const FigRegs &F, F3 x, F3 d, F3 y, F3 yn) {
    if (F.P.type != RT_PRIM_TRIANGLE) return pdf_one4(F.P, x, d, y, yn);
    F3 a = F.data3, b = F.P.data - a, c = F.data2 - a;
    float pointProb = (float)(1.0 / (0.5 * (double)len(crossr(b, c))));       // :121-127
    return (float)((double)(pointProb * len2(x - y)) / fabs((double)dot(d, yn)));
}
// FiguresMix::pdfOneFigureLight, distributions.h:219-254
RT_DEV float light_pdf_one5(const FigRegs &F, F3 x, F3 d) {
    float t1; F3 n1; bool in1;
    if (!fig_hit5(F, x, d, t1, n1, in1)) return 0.f;
    if (t1 != t1) return __builtin_inff();
    float ans = pdf_one5(F, x, d, x + t1 * d, n1);
    if (F.P.type == RT_PRIM_TRIANGLE) return ans;
    float t2; F3 n2; bool in2;
    if (!fig_hit5(F, x + (float)((double)t1 + 1e-4) * d, d, t2, n2, in2)) return ans;
    F3 y2 = x + (float)((double)t1 + 1e-4 + (double)t2) * d;
    return ans + pdf_one5(F, x, d, y2, n2);
}
// FiguresMix::getTotalPdf, distributions.h:256-274: total(left) + total(right), sequential sum inside a leaf — the same
// tree of float additions replayed with TODO(child) / ADD(partial) frames (see light_pdf_sum in rt_device.h).
RT_DEV float light_pdf_sum5(const SceneView5 &S, F3 x, F3 d, uint32_t *stack) {
    // the reference's own box test at every node (distributions.h:256-262): a failed box contributes 0 whatever lies below it
    int sp = 0;
    unsigned long long addmask = 0;
    uint32_t cur = 0;
    bool descending = true;
    float v = 0.f;
    for (;;) {
        if (descending) {
            const RefNodeView n = load_ref_node(S.ref_light_nodes + cur);
            float tb; bool inside_box;
            if (!ref_box_test(n.mn, n.mx, x, d, tb, inside_box)) { v = 0.f; descending = false; }
            else if (n.left == 0) {
                float result = 0.f;
                for (uint32_t i = n.first; i < n.last; i++) {
                    FigRegs F = load_fig5(S.lights + i);
                    result += light_pdf_one5(F, x, d);
                }
                v = result;
                descending = false;
            } else if (sp < RT5_STACK) { addmask &= ~(1ull << sp); stack[sp++] = n.right; cur = n.left; }
            else { v = 0.f; descending = false; } // deeper than the host admits (checked there)
        } else {
            if (sp == 0) break;
            --sp;
            uint32_t f = stack[sp];
            if ((addmask >> sp) & 1ull) v = __uint_as_float(f) + v;
            else { addmask |= 1ull << sp; stack[sp++] = __float_as_uint(v); cur = f; descending = true; }
        }
    }
    return v;
}

// Mix::sample (distributions.h:283-290) -> Cosine::sample (:43-53) or FiguresMix::sample (:200-209) -> one light's sample
RT_DEV F3 mix_sample5(const SceneView5 &S, Rng &rng, F3 x, F3 n) {
    float comps = S.n_lights ? 2.f : 1.f;
    int distNum = (int)(rng_u01(rng) * comps);
    if (distNum == 0) {
        float a = rng_n01(rng), b = rng_n01(rng), c = rng_n01(rng);
        F3 d = normalize(f3(a, b, c)) + n;
        float l = len(d);
        if (l <= 1e-9f || dot(d, n) <= 1e-9f || l != l) return n;
        return (float)(1. / (double)l) * d;
    }
    int li = (int)(rng_u01(rng) * (float)S.n_lights);
    FigRegs F = load_fig5(S.lights + li);
    if (F.P.type == RT_PRIM_TRIANGLE) {                                        // TriangleLight::sample :129-142
        F3 a = F.data3, b = F.P.data - a, c = F.data2 - a;
        float u = rng_u01(rng);
        float v = rng_u01(rng);
        if ((double)(u + v) > 1.) { u = 1 - u; v = 1 - v; }
        F3 point = F.P.position + qtransform(qconj(F.P.rot), a + u * b + v * c);
        return normalize(point - x);
    }
    F3 dir = f3(0.f, 1.f, 0.f);
    for (int attempt = 0; attempt < RT4_MAX_REJECTIONS; attempt++) {
        F3 point;
        if (F.P.type == RT_PRIM_BOX) {                                         // BoxLight::sample :84-105 (constructor arguments right to left)
            float sx = F.P.data.x, sy = F.P.data.y, sz = F.P.data.z;
            float wx = sy * sz, wy = sx * sz, wz = sx * sy;
            float u = rng_u01(rng) * (wx + wy + wz);
            float flip = (double)rng_u01(rng) > 0.5 ? 1.f : -1.f;
            if (u < wx) { float c = (2 * rng_u01(rng) - 1) * sz; float b = (2 * rng_u01(rng) - 1) * sy; point = f3(flip * sx, b, c); }
            else if (u < wx + wy) { float c = (2 * rng_u01(rng) - 1) * sz; float a = (2 * rng_u01(rng) - 1) * sx; point = f3(a, flip * sy, c); }
            else { float b = (2 * rng_u01(rng) - 1) * sy; float a = (2 * rng_u01(rng) - 1) * sx; point = f3(a, b, flip * sz); }
        } else {                                                               // EllipsoidLight::sample :160-171 (the pixel's shared n01)
            float a = rng_n01(rng), b = rng_n01(rng), c = rng_n01(rng);
            point = F.P.data * normalize(f3(a, b, c));
        }
        F3 actual = qtransform(qconj(F.P.rot), point) + F.P.position;
        dir = normalize(actual - x);
        float t; F3 nn; bool inside;
        if (fig_hit5(F, x, dir, t, nn, inside)) break;
    }
    return dir;
}
// Mix::pdf :292-302 with FiguresMix::pdf :211-213
RT_DEV float mix_pdf5(const SceneView5 &S, F3 x, F3 n, F3 d, uint32_t *stack) {
    float ans = 0.f;
    ans += smax(0.f, dot(d, n) / RT4_PI);
    if (S.n_lights == 0) return ans / 1.f;
    ans += light_pdf_sum5(S, x, d, stack) / (float)S.n_lights;
    return ans / 2.f;
}

// Scene::getColor, hw5/src/scene.cpp:47-103
RT_DEV F3 trace_tree5(const SceneView5 &S, int ray_depth, Rng &rng, uint32_t *stack, F3 o, F3 d) {
    Frame3 frames[RT4_MAX_DEPTH];
    int fp = 0;
    const float epsf = 9.99999974737875163555e-05f; // (float)eps, eps = 1e-4L (same float as (float)1e-4)
    F3 ret = f3(0.f, 0.f, 0.f);
    bool evaluating = true;
    for (;;) {
        if (evaluating) {
            if (fp >= ray_depth) { ret = f3(0.f, 0.f, 0.f); evaluating = false; continue; }
            Hit5 h = closest_hit5(S, o, d, stack);
            if (h.idx < 0) { ret = f3(S.bg); evaluating = false; continue; }
            FigRegs F = load_fig5(S.figs + h.idx);
            F3 x = o + h.t * d;
            if (F.P.kind == RT_MAT_DIFFUSE) {
                F3 xs = x + epsf * h.n;
                F3 w = mix_sample5(S, rng, xs, h.n);
                if (dot(w, h.n) < 0) { ret = F.P.emission; evaluating = false; continue; }
                float pdf = mix_pdf5(S, xs, h.n, w, stack);
                Frame3 &f = frames[fp++];
                f.kind = F3_MUL; f.emission = F.P.emission;
                f.mult = (float)(1. / (double)(RT4_PI * pdf) * (double)dot(w, h.n)) * F.P.color;
                o = x + epsf * w; d = w;
                continue;
            }
            F3 dn = normalize(d);
            F3 refl = dn - (float)(2. * (double)dot(h.n, dn)) * h.n;
            Frame3 &f = frames[fp++];
            f.emission = F.P.emission; f.mult = F.P.color; f.x = x; f.dn = dn; f.norma = h.n; f.inside = h.inside; f.ior = F.P.ior;
            f.kind = F.P.kind == RT_MAT_METALLIC ? F3_MUL : F3_DIEL_REFLECT;
            o = x + epsf * refl; d = refl;
        } else {
            if (fp == 0) break;
            Frame3 &f = frames[--fp];
            if (f.kind == F3_MUL) { ret = f.emission + f.mult * ret; continue; }
            if (f.kind == F3_DIEL_REFRACT) {
                F3 refracted = ret;
                if (!f.inside) refracted = refracted * f.mult;
                ret = f.emission + refracted;
                continue;
            }
            float eta1 = 1.f, eta2 = f.ior;
            if (f.inside) { float tmp = eta1; eta1 = eta2; eta2 = tmp; }
            F3 l = neg(f.dn);
            float nl = dot(f.norma, l);
            float sinTheta2 = (float)((double)(eta1 / eta2) * sqrt((double)(1 - nl * nl)));
            if (fabs((double)sinTheta2) > 1.) { ret = f.emission + ret; continue; }
            float rr = (eta1 - eta2) / (eta1 + eta2);
            float r0 = rr * rr;
            double om = (double)(1 - nl), om2 = om * om;
            float r = (float)((double)r0 + (double)(1 - r0) * (om2 * om2 * om));
            if (rng_u01(rng) < r) { ret = f.emission + ret; continue; }
            float cosTheta2 = sqrtf(1 - sinTheta2 * sinTheta2);
            F3 refr = (eta1 / eta2) * neg(l) + (eta1 / eta2 * nl - cosTheta2) * f.norma;
            f.kind = F3_DIEL_REFRACT;
            fp++;
            o = f.x + epsf * refr; d = refr;
            evaluating = true;
        }
    }
    return ret;
}

__global__ __launch_bounds__(64) void render_hw5_kernel(SceneView5 S, RenderView R, float tan_fov_y, uint32_t n_work) {
    const int lane = threadIdx.x & 63;
    const int sub_x = R.tile_w >> 3, sub_per_tile = sub_x * (R.tile_h >> 3);
    uint32_t stack[RT5_STACK];
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(R.work_counter, 1u);
        w = __shfl(w, 0);
        if (w >= n_work) break;
        uint32_t st = w / sub_per_tile, sub = w % sub_per_tile;
        uint32_t gt = R.shard_count > 1 ? (uint32_t)R.shard_index + st * (uint32_t)R.shard_count : st;
        int tx0 = (int)(gt % (uint32_t)R.tiles_x) * R.tile_w, ty0 = (int)(gt / (uint32_t)R.tiles_x) * R.tile_h;
        int lx = (int)(sub % sub_x) * 8 + (lane & 7), ly = (int)(sub / sub_x) * 8 + (lane >> 3);
        int x = tx0 + lx, y = ty0 + ly;
        bool inside = x < R.width && y < R.height;
        size_t out_index = R.shard_count > 1 ? ((size_t)st * R.tile_h + ly) * R.tile_w + lx : (size_t)y * R.width + x;
        F3 px = f3(0.f, 0.f, 0.f);
        if (inside) {
            Rng rng;
            rng_seed(rng, (uint32_t)(y * R.width + x));                 // hw5/src/sceneio.cpp:110
            F3 color = f3(0.f, 0.f, 0.f);
            for (int s = 0; s < R.samples; s++) {                       // scene.cpp:105-126: all-float camera ray, no half-pixel offset
                float fx = (float)x + rng_u01(rng);
                float fy = (float)y + rng_u01(rng);
                float nx = S.tan_fov_x * (2 * fx / (float)R.width - 1);
                float ny = tan_fov_y * (2 * fy / (float)R.height - 1);
                F3 o = f3(S.cam_pos);
                F3 d = nx * f3(S.cam_right) - ny * f3(S.cam_up) + f3(S.cam_fwd);
                color = color + trace_tree5(S, R.ray_depth, rng, stack, o, d);
            }
            px = R.inv_samples * color;
        }
        if (inside || R.shard_count > 1) {
            if (R.out_rgb) { R.out_rgb[3 * out_index] = px.x; R.out_rgb[3 * out_index + 1] = px.y; R.out_rgb[3 * out_index + 2] = px.z; }
            if (R.out_rgb8) {
                R.out_rgb8[3 * out_index] = inside ? tonemap1(px.x) : 0;
                R.out_rgb8[3 * out_index + 1] = inside ? tonemap1(px.y) : 0;
                R.out_rgb8[3 * out_index + 2] = inside ? tonemap1(px.z) : 0;
            }
        }
    }
}

} // namespace dev
} // namespace rtamd
