// Kernels for the .txt-scene snapshots: hw1 ray caster (BASELINE.json configs[0]) and hw3 path tracer over
// analytic primitives (configs[1]).  Flat primitive list, every query tests every primitive
// (hw3/src/scene.cpp:11-29), no BVH.
//
// hw3 draws the whole frame's random numbers from one file-static engine (hw3/src/scene.cpp:5-7), which no
// parallel machine can replay; here every pixel gets its own minstd_rand(y*W+x) — the policy hw5+ adopted —
// so parity with the reference is statistical, while the arithmetic is checked exactly against the oracle run
// with the same per-pixel seeds (tests/test_gpu_txt.py).
#pragma once
#include "rt_device.h"

namespace rtamd {

struct GpuPrim {              // 80 bytes
    float data[3]; int32_t type;
    float position[3]; int32_t kind;
    float rotation[4];
    float color[3]; float ior;
    float emission[3]; float pad;
};
static_assert(sizeof(GpuPrim) == 80, "GpuPrim must be 80 bytes");

struct GpuLight {             // 64 bytes, hw2 only
    float intensity[3]; int32_t type;
    float position[3]; float pad0;
    float attenuation[3]; float pad1;
    float direction[3]; float pad2;
};
static_assert(sizeof(GpuLight) == 64, "GpuLight must be 64 bytes");

struct SceneViewTxt {
    const GpuPrim *prims;
    uint32_t n_prims;
    float cam_pos[3], cam_right[3], cam_up[3], cam_fwd[3];
    float bg[3];
    float tan_fov_x;          // (float)tan((double)(fovX / 2)), hw3/src/scene.cpp:100
    float tan_fov_x_f;        // tanf(fovX / 2): hw1/hw2 compile against <math.h>, where tan(float) is the float overload
    const GpuLight *lights;   // hw2
    uint32_t n_lights;
    float ambient[3];         // hw2 AMBIENT_LIGHT
    const uint32_t *light_prims; // hw4: emissive BOX / ELLIPSOID primitives in figure order (hw4/src/scene.cpp:12-21)
    uint32_t n_light_prims;
};

namespace dev {

struct PrimRegs { F3 data, position, color, emission; Quat rot; int type, kind; float ior; };
RT_DEV PrimRegs load_prim(const GpuPrim *p) {
    const float4 *q = reinterpret_cast<const float4 *>(p);
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    PrimRegs P;
    P.data = f3(q0.x, q0.y, q0.z); P.type = (int)__float_as_uint(q0.w);
    P.position = f3(q1.x, q1.y, q1.z); P.kind = (int)__float_as_uint(q1.w);
    P.rot.v = f3(q2.x, q2.y, q2.z); P.rot.w = q2.w;
    P.color = f3(q3.x, q3.y, q3.z); P.ior = q3.w;
    P.emission = f3(q4.x, q4.y, q4.z);
    return P;
}
RT_DEV F3 div3(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }

// hw3/src/primitives.cpp:28-47.  FLOAT_ROOTS: the hw1/hw2 form (hw2/src/primitives.cpp:37-56), where sqrt(d) is the
// float overload and the whole quotient stays in float; hw3+ (<cmath>) promote to double and narrow once.
template <bool FLOAT_ROOTS>
RT_DEV bool smallest_root(float a, float b, float c, float &t, bool &inside) {
    float d = b * b - 4 * a * c;
    if (d <= 0) return false;
    float x1, x2;
    if (FLOAT_ROOTS) {
        float sd = sqrtf(d);
        x1 = (-b - sd) / (2 * a);
        x2 = (-b + sd) / (2 * a);
    } else {
        double sd = sqrt((double)d), den = (double)(2 * a);
        x1 = (float)(((double)(-b) - sd) / den);
        x2 = (float)(((double)(-b) + sd) / den);
    }
    if (x1 > x2) { float tmp = x1; x1 = x2; x2 = tmp; }
    if (x2 < 0) return false;
    if (x1 < 0) { t = x2; inside = true; } else { t = x1; inside = false; }
    return true;
}
// Slab test of the BOX primitive (hw3/src/primitives.cpp:81-102, reference arithmetic incl. IEEE divisions).
RT_DEV bool box_slabs(F3 s, F3 o, F3 d, float &t, bool &inside) {
    F3 ts1 = div3(neg(s) - o, d), ts2 = div3(s - o, d);
    float t1x = smin(ts1.x, ts2.x), t2x = smax(ts1.x, ts2.x);
    float t1y = smin(ts1.y, ts2.y), t2y = smax(ts1.y, ts2.y);
    float t1z = smin(ts1.z, ts2.z), t2z = smax(ts1.z, ts2.z);
    float t1 = smax(smax(t1x, t1y), t1z), t2 = smin(smin(t2x, t2y), t2z);
    if (t1 > t2 || t2 < 0) return false;
    if (t1 < 0) { inside = true; t = t2; } else { inside = false; t = t1; }
    return true;
}

// Figure::intersect of hw3 (hw3/src/primitives.cpp:8-123): object-space ray via q*p*conj(q) with the quaternion
// exactly as parsed (it may be non-unit), normal rotated back and normalised.
// PLANE_TMAX: hw4+ reject plane hits at t >= T_MAX = 1e4 (hw4/src/primitives.cpp:8,74).
template <bool FLOAT_ROOTS, bool PLANE_TMAX = false>
RT_DEV bool prim_hit(const PrimRegs &P, F3 o, F3 d, float &t, F3 &norma, bool &inside) {
    F3 to = qtransform(P.rot, o - P.position), td = qtransform(P.rot, d);
    F3 n;
    if (P.type == RT_PRIM_ELLIPSOID) {
        F3 r = P.data;
        F3 orr = div3(to, r), drr = div3(td, r);
        float c = len2(orr) - 1;
        float b = 2.0f * dot(orr, drr);
        float a = len2(drr);
        if (!smallest_root<FLOAT_ROOTS>(a, b, c, t, inside)) return false;
        F3 point = to + t * td;
        n = div3(point, r * r);
        if (inside) n = neg(n);
        n = normalize(n);
    } else if (P.type == RT_PRIM_PLANE) {
        F3 pn = P.data;
        float dn = dot(td, pn);
        t = -dot(to, pn) / dn;
        if (!(t > 0) || (PLANE_TMAX && !(t < 1e4f))) return false;
        inside = dn > 0;
        n = inside ? neg(pn) : pn;
    } else {
        F3 s = P.data;
        if (!box_slabs(s, to, td, t, inside)) return false;
        F3 p = to + t * td;
        n = div3(p, s);
        float mx = smax(smax(fabsf(n.x), fabsf(n.y)), fabsf(n.z));
        if (fabsf(n.x) != mx) n.x = 0;
        if (fabsf(n.y) != mx) n.y = 0;
        if (fabsf(n.z) != mx) n.z = 0;
        if (inside) n = neg(n);
    }
    norma = normalize(qtransform(qconj(P.rot), n));
    return true;
}

RT_DEV bool prim_hit3(const PrimRegs &P, F3 o, F3 d, float &t, F3 &norma, bool &inside) { return prim_hit<false>(P, o, d, t, norma, inside); }

// hw3/src/scene.cpp:99-107 (shared by hw1/src/scene.cpp:22-30): pixel centre +0.5 on top of the jitter, FOV_X based.
RT_DEV void camera_ray_txt(const SceneViewTxt &S, float tan_fov_x, float tan_fov_y, int width, int height, float x, float y, F3 &o, F3 &d) {
    float nx = (float)((double)tan_fov_x * (2 * ((double)x + 0.5) / (double)width - 1));
    float ny = (float)((double)tan_fov_y * (2 * ((double)y + 0.5) / (double)height - 1));
    o = f3(S.cam_pos);
    d = nx * f3(S.cam_right) - ny * f3(S.cam_up) + f3(S.cam_fwd);
}

// ---- hw1 --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void render_hw1_kernel(SceneViewTxt S, int width, int height, float tan_fov_y, float *out_rgb, uint8_t *out_rgb8) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= width * height) return;
    int x = i % width, y = i / width;
    F3 o, d;
    camera_ray_txt(S, S.tan_fov_x_f, tan_fov_y, width, height, (float)x, (float)y, o, d);
    F3 ans = f3(S.bg);
    float best = -1;
    for (uint32_t k = 0; k < S.n_prims; k++) {                         // hw1/src/scene.cpp:10-18
        PrimRegs P = load_prim(S.prims + k);
        F3 to = qtransform(P.rot, o - P.position), td = qtransform(P.rot, d);
        float t = 0; bool hit = false;
        if (P.type == RT_PRIM_ELLIPSOID) {                             // hw1/src/primitives.cpp:54-59: b = (2*(o/r)) . (d/r)
            F3 orr = div3(to, P.data), drr = div3(td, P.data);
            float c = len2(orr) - 1;
            float b = dot(2 * orr, drr);
            float a = len2(drr);
            bool inside;
            hit = smallest_root<true>(a, b, c, t, inside);           // hw1/src/primitives.cpp:33-52, float sqrt
        } else if (P.type == RT_PRIM_PLANE) {                          // :64-70 (normal not normalised in hw1)
            t = -dot(to, P.data) / dot(td, P.data);
            hit = t > 0;
        } else { bool inside; hit = box_slabs(P.data, to, td, t, inside); }
        if (hit && (best == -1 || t < best)) { best = t; ans = P.color; }
    }
    if (out_rgb) { out_rgb[3 * i] = ans.x; out_rgb[3 * i + 1] = ans.y; out_rgb[3 * i + 2] = ans.z; }
    if (out_rgb8) {                                                     // hw1/src/color.cpp:13-19 (no tonemap in hw1)
        out_rgb8[3 * i] = (uint8_t)(int)round((double)(255 * ans.x));
        out_rgb8[3 * i + 1] = (uint8_t)(int)round((double)(255 * ans.y));
        out_rgb8[3 * i + 2] = (uint8_t)(int)round((double)(255 * ans.z));
    }
}

// ---- hw3 --------------------------------------------------------------------------------------------------------
#define RT3_MAX_DEPTH 8
enum { F3_MUL = 0, F3_DIEL_REFLECT = 1, F3_DIEL_REFRACT = 2 };
struct Frame3 { F3 emission, mult, x, dn, norma; int kind; bool inside; float ior; };

RT_DEV F3 trace_tree3(const SceneViewTxt &S, int ray_depth, Rng &rng, F3 o, F3 d) {
    Frame3 frames[RT3_MAX_DEPTH];
    int fp = 0;
    const float epsf = (float)0.0001;
    F3 ret = f3(0.f, 0.f, 0.f);
    bool evaluating = true;
    for (;;) {
        if (evaluating) {
            if (fp >= ray_depth) { ret = f3(0.f, 0.f, 0.f); evaluating = false; continue; }
            int pos = -1; float bt = 0; F3 bn = f3(0.f, 0.f, 0.f); bool bin = false;
            for (uint32_t k = 0; k < S.n_prims; k++) {                  // hw3/src/scene.cpp:14-23: strict '<' keeps the first
                PrimRegs P = load_prim(S.prims + k);
                float t; F3 n; bool inside;
                if (prim_hit3(P, o, d, t, n, inside) && t <= __builtin_inff() && (pos == -1 || t < bt)) { pos = (int)k; bt = t; bn = n; bin = inside; }
            }
            if (pos < 0) { ret = f3(S.bg); evaluating = false; continue; }
            PrimRegs P = load_prim(S.prims + pos);
            F3 x = o + bt * d;
            Frame3 &f = frames[fp];
            if (P.kind == RT_MAT_DIFFUSE) {                             // scene.cpp:45-51: uniform hemisphere, weight 2 cos
                float a = rng_n01(rng), b = rng_n01(rng), c = rng_n01(rng);
                F3 w = normalize(f3(a, b, c));
                if (dot(w, bn) < 0) w = neg(w);
                f.kind = F3_MUL; f.emission = P.emission; f.mult = (2 * dot(w, bn)) * P.color;
                fp++;
                o = x + epsf * w; d = w;
            } else {
                F3 dn = normalize(d);
                F3 refl = dn - (float)(2. * (double)dot(bn, dn)) * bn;  // scene.cpp:53,57
                f.emission = P.emission; f.mult = P.color; f.x = x; f.dn = dn; f.norma = bn; f.inside = bin; f.ior = P.ior;
                f.kind = P.kind == RT_MAT_METALLIC ? F3_MUL : F3_DIEL_REFLECT;
                fp++;
                o = x + epsf * refl; d = refl;
            }
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
            float eta1 = 1.f, eta2 = f.ior;                             // scene.cpp:61-85
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

#ifndef RT3_MIN_WAVES
#define RT3_MIN_WAVES 4 // waves per SIMD the register allocation aims at; config 2: 1 -> 14.1 ms, 3 -> 14.3, 4 -> 11.9, 5 -> 12.0, 6 -> 12.2, 8 -> 17.6
#endif
__global__ __launch_bounds__(64, RT3_MIN_WAVES) void render_hw3_kernel(SceneViewTxt S, RenderView R, float tan_fov_y, uint32_t n_work) {
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
            F3 color = f3(0.f, 0.f, 0.f);
            for (int s = 0; s < R.samples; s++) {                       // hw3/src/scene.cpp:89-97
                float nx = (float)x + rng_u01(rng);
                float ny = (float)y + rng_u01(rng);
                F3 o, d;
                camera_ray_txt(S, S.tan_fov_x, tan_fov_y, R.width, R.height, nx, ny, o, d);
                color = color + trace_tree3(S, R.ray_depth, rng, o, d);
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
