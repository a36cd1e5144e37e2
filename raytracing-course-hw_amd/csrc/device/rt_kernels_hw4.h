// Kernel for the hw4 snapshot: hw3's path tracer over the flat primitive list with importance sampling,
// Mix{Cosine, Mix{BoxLight | EllipsoidLight ...}} (hw4/src/scene.cpp:10-122, hw4/src/include/distributions.h:13-204).
//
// Like hw3 the reference draws the whole frame from one file-static engine, which cannot be replayed in parallel; here
// each pixel owns an engine(y*W+x) and fresh distribution objects, so parity with the reference is statistical and the
// arithmetic is checked exactly against the oracle run with the same per-pixel streams (tests/test_gpu_hw4.py).
// Every distribution object of the reference owns its own std::normal_distribution (the cached second value is per
// object): Cosine uses Rng::saved, each EllipsoidLight a slot of LightNormals.
#pragma once
#include "rt_kernels_txt.h"

namespace rtamd {
namespace dev {

#define RT4_MAX_DEPTH 8
#define RT4_MAX_LIGHTS 32
#define RT4_MAX_REJECTIONS 1000000 // the reference loops forever when a light cannot be hit; a GPU wave must not
#define RT4_PI 3.14159274101257324f // const float PI = acos(-1), distributions.h:9

struct LightNormals { float saved[RT4_MAX_LIGHTS]; uint32_t has; };

RT_DEV float rng_n01_slot(Rng &r, LightNormals &N, int slot) {
    float s0 = r.saved; bool h0 = r.has_saved;
    r.saved = N.saved[slot]; r.has_saved = (N.has >> slot) & 1u;
    float v = rng_n01(r);
    N.saved[slot] = r.saved;
    N.has = (N.has & ~(1u << slot)) | ((r.has_saved ? 1u : 0u) << slot);
    r.saved = s0; r.has_saved = h0;
    return v;
}

RT_DEV bool prim_hit4(const PrimRegs &P, F3 o, F3 d, float &t, F3 &norma, bool &inside) { return prim_hit<false, true>(P, o, d, t, norma, inside); }

// distributions.h:115-118 (box) and :159-164 (ellipsoid): solid-angle density of one surface point
RT_DEV float pdf_one4(const PrimRegs &P, F3 x, F3 d, F3 y, F3 yn) {
    if (P.type == RT_PRIM_BOX) {
        float sx = P.data.x, sy = P.data.y, sz = P.data.z;
        float sTotal = 8 * (sy * sz + sx * sz + sx * sy);
        return (float)((double)len2(x - y) / ((double)sTotal * fabs((double)dot(d, yn))));
    }
    F3 r = P.data;
    F3 n = div3(qtransform(P.rot, y - P.position), r);
    float pointProb = (float)(1. / (double)(4 * RT4_PI * len(f3(n.x * r.y * r.z, r.x * n.y * r.z, r.x * r.y * n.z))));
    return (float)((double)(pointProb * len2(x - y)) / fabs((double)dot(d, yn)));
}
// FigureLight::pdf, distributions.h:85-107: first hit, plus the second one behind it
RT_DEV float light_pdf4(const PrimRegs &P, F3 x, F3 d) {
    float t1; F3 n1; bool in1;
    if (!prim_hit4(P, x, d, t1, n1, in1)) return 0.f;
    if (t1 != t1) return __builtin_inff();
    float ans = pdf_one4(P, x, d, x + t1 * d, n1);
    float t2; F3 n2; bool in2;
    if (!prim_hit4(P, x + (float)((double)t1 + 0.0001) * d, d, t2, n2, in2)) return ans;
    F3 y2 = x + (float)((double)t1 + 0.0001 + (double)t2) * d;
    return ans + pdf_one4(P, x, d, y2, n2);
}
// BoxLight::sample :125-151 / EllipsoidLight::sample :169-180
RT_DEV F3 light_sample4(const SceneViewTxt &S, Rng &rng, LightNormals &N, int li, F3 x) {
    PrimRegs P = load_prim(S.prims + S.light_prims[li]);
    F3 dir = f3(0.f, 1.f, 0.f);
    for (int attempt = 0; attempt < RT4_MAX_REJECTIONS; attempt++) {
        F3 point;
        if (P.type == RT_PRIM_BOX) {
            float sx = P.data.x, sy = P.data.y, sz = P.data.z;
            float wx = sy * sz, wy = sx * sz, wz = sx * sy;
            float u = rng_u01(rng) * (wx + wy + wz);
            float flip = (double)rng_u01(rng) > 0.5 ? 1.f : -1.f;
            // Vec3(a, b, c): g++ evaluates constructor-call arguments right to left, so the last coordinate draws first
            if (u < wx) { float c = (2 * rng_u01(rng) - 1) * sz; float b = (2 * rng_u01(rng) - 1) * sy; point = f3(flip * sx, b, c); }
            else if (u < wx + wy) { float c = (2 * rng_u01(rng) - 1) * sz; float a = (2 * rng_u01(rng) - 1) * sx; point = f3(a, flip * sy, c); }
            else { float b = (2 * rng_u01(rng) - 1) * sy; float a = (2 * rng_u01(rng) - 1) * sx; point = f3(a, b, flip * sz); }
        } else {
            float a = rng_n01_slot(rng, N, li), b = rng_n01_slot(rng, N, li), c = rng_n01_slot(rng, N, li);
            point = P.data * normalize(f3(a, b, c));
        }
        F3 actual = qtransform(qconj(P.rot), point) + P.position;
        dir = normalize(actual - x);
        float t; F3 n; bool inside;
        if (prim_hit4(P, x, dir, t, n, inside)) break;
    }
    return dir;
}
// Mix::sample :194-197 (outer {Cosine, lights}, then the inner light choice); Cosine::sample :55-67
RT_DEV F3 mix_sample4(const SceneViewTxt &S, Rng &rng, LightNormals &N, F3 x, F3 n) {
    float comps = S.n_light_prims ? 2.f : 1.f;
    int distNum = (int)(rng_u01(rng) * comps);
    if (distNum != 0) {
        int li = (int)(rng_u01(rng) * (float)S.n_light_prims);
        return light_sample4(S, rng, N, li, x);
    }
    float a = rng_n01(rng), b = rng_n01(rng), c = rng_n01(rng);
    F3 d = normalize(f3(a, b, c)) + n;
    float l = len(d);
    if (l <= 1e-9f || dot(d, n) <= 1e-9f || l != l) return n;
    return (float)(1. / (double)l) * d;
}
// Mix::pdf :199-205
RT_DEV float mix_pdf4(const SceneViewTxt &S, F3 x, F3 n, F3 d) {
    float ans = 0.f;
    ans += smax(0.f, dot(d, n) / RT4_PI);
    if (S.n_light_prims == 0) return ans / 1.f;
    float inner = 0.f;
    for (uint32_t k = 0; k < S.n_light_prims; k++) inner += light_pdf4(load_prim(S.prims + S.light_prims[k]), x, d);
    ans += inner / (float)S.n_light_prims;
    return ans / 2.f;
}

// Scene::getColor, hw4/src/scene.cpp:51-112, with hw3's frame machine (rt_kernels_txt.h trace_tree3)
RT_DEV F3 trace_tree4(const SceneViewTxt &S, int ray_depth, Rng &rng, LightNormals &N, F3 o, F3 d) {
    Frame3 frames[RT4_MAX_DEPTH];
    int fp = 0;
    const float epsf = (float)0.0001;
    F3 ret = f3(0.f, 0.f, 0.f);
    bool evaluating = true;
    for (;;) {
        if (evaluating) {
            if (fp >= ray_depth) { ret = f3(0.f, 0.f, 0.f); evaluating = false; continue; }
            int pos = -1; float bt = 0; F3 bn = f3(0.f, 0.f, 0.f); bool bin = false;
            for (uint32_t k = 0; k < S.n_prims; k++) {
                PrimRegs P = load_prim(S.prims + k);
                float t; F3 n; bool inside;
                if (prim_hit4(P, o, d, t, n, inside) && t <= __builtin_inff() && (pos == -1 || t < bt)) { pos = (int)k; bt = t; bn = n; bin = inside; }
            }
            if (pos < 0) { ret = f3(S.bg); evaluating = false; continue; }
            PrimRegs P = load_prim(S.prims + pos);
            F3 x = o + bt * d;
            if (P.kind == RT_MAT_DIFFUSE) {                              // :67-74
                F3 xs = x + epsf * bn;
                F3 w = mix_sample4(S, rng, N, xs, bn);
                if (dot(w, bn) < 0) { ret = P.emission; evaluating = false; continue; }
                float pdf = mix_pdf4(S, xs, bn, w);
                Frame3 &f = frames[fp++];
                f.kind = F3_MUL; f.emission = P.emission;
                f.mult = (float)(1. / (double)(RT4_PI * pdf) * (double)dot(w, bn)) * P.color;
                o = x + epsf * w; d = w;
                continue;
            }
            F3 dn = normalize(d);
            F3 refl = dn - (float)(2. * (double)dot(bn, dn)) * bn;
            Frame3 &f = frames[fp++];
            f.emission = P.emission; f.mult = P.color; f.x = x; f.dn = dn; f.norma = bn; f.inside = bin; f.ior = P.ior;
            f.kind = P.kind == RT_MAT_METALLIC ? F3_MUL : F3_DIEL_REFLECT;
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
            float eta1 = 1.f, eta2 = f.ior;                             // :83-110, as hw3
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

__global__ __launch_bounds__(64) void render_hw4_kernel(SceneViewTxt S, RenderView R, float tan_fov_y, uint32_t n_work) {
    const int lane = threadIdx.x & 63;
    const int sub_x = R.tile_w >> 3, sub_per_tile = sub_x * (R.tile_h >> 3);
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
            rng_seed(rng, (uint32_t)(y * R.width + x));
            LightNormals N;
            N.has = 0u;
            F3 color = f3(0.f, 0.f, 0.f);
            for (int s = 0; s < R.samples; s++) {                       // hw4/src/scene.cpp:114-132: all-float camera ray, no half-pixel offset
                float fx = (float)x + rng_u01(rng);
                float fy = (float)y + rng_u01(rng);
                float nx = S.tan_fov_x * (2 * fx / (float)R.width - 1);
                float ny = tan_fov_y * (2 * fy / (float)R.height - 1);
                F3 o = f3(S.cam_pos);
                F3 d = nx * f3(S.cam_right) - ny * f3(S.cam_up) + f3(S.cam_fwd);
                color = color + trace_tree4(S, R.ray_depth, rng, N, o, d);
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
