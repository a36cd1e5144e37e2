// Kernel for the hw2 snapshot: deterministic Whitted-style tracer over the flat primitive list with point and
// directional lights (hw2/src/scene.cpp:8-98).  No random numbers: the result must equal the reference's float
// radiance bit for bit.  hw2 compiles against <math.h>, so its sqrt/fabs on floats are the float functions
// (hw2/src/primitives.cpp:37-56) — unlike hw3+, see prim_hit<FLOAT_ROOTS>.
#pragma once
#include "rt_kernels_txt.h"

namespace rtamd {
namespace dev {

#define RT2_MAX_DEPTH 16
enum { F2_MUL = 0, F2_DIEL_REFLECTED = 1, F2_DIEL_REFRACTED = 2 };
struct Frame2 { F3 color, x, l, norma, reflected; int kind; bool inside; float ior; };

struct LightRegs { F3 intensity, position, attenuation, direction; int type; };
RT_DEV LightRegs load_light(const GpuLight *p) {
    const float4 *q = reinterpret_cast<const float4 *>(p);
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    LightRegs L;
    L.intensity = f3(q0.x, q0.y, q0.z); L.type = (int)__float_as_uint(q0.w);
    L.position = f3(q1.x, q1.y, q1.z);
    L.attenuation = f3(q2.x, q2.y, q2.z);
    L.direction = f3(q3.x, q3.y, q3.z);
    return L;
}

// Scene::intersect (hw2/src/scene.cpp:8-28): first strictly-nearest figure with t <= tmax.
RT_DEV int closest_prim2(const SceneViewTxt &S, F3 o, F3 d, float &bt, F3 &bn, bool &bin) {
    int pos = -1;
    for (uint32_t k = 0; k < S.n_prims; k++) {
        PrimRegs P = load_prim(S.prims + k);
        float t; F3 n; bool inside;
        if (prim_hit<true>(P, o, d, t, n, inside) && t <= __builtin_inff() && (pos == -1 || t < bt)) { pos = (int)k; bt = t; bn = n; bin = inside; }
    }
    return pos;
}
// The same query used as a shadow test: only whether anything lies within tmax matters.
RT_DEV bool occluded2(const SceneViewTxt &S, F3 o, F3 d, float tmax) {
    for (uint32_t k = 0; k < S.n_prims; k++) {
        PrimRegs P = load_prim(S.prims + k);
        float t; F3 n; bool inside;
        if (prim_hit<true>(P, o, d, t, n, inside) && t <= tmax) return true;
    }
    return false;
}

// Scene::getColor (hw2/src/scene.cpp:30-84) as an explicit frame machine: a dielectric hit evaluates the reflected
// subtree, then (unless totally reflected) the refracted subtree, then blends them with Schlick's weight.
RT_DEV F3 trace_tree2(const SceneViewTxt &S, int ray_depth, F3 o, F3 d) {
    Frame2 frames[RT2_MAX_DEPTH];
    int fp = 0;
    const float epsf = (float)0.0001;
    F3 ret = f3(0.f, 0.f, 0.f);
    bool evaluating = true;
    for (;;) {
        if (evaluating) {
            evaluating = false;
            if (fp >= ray_depth) { ret = f3(0.f, 0.f, 0.f); continue; }
            float t = 0; F3 norma = f3(0.f, 0.f, 0.f); bool inside = false;
            int pos = closest_prim2(S, o, d, t, norma, inside);
            if (pos < 0) { ret = f3(S.bg); continue; }
            PrimRegs P = load_prim(S.prims + pos);
            if (P.kind == RT_MAT_DIFFUSE) {                              // :43-52
                F3 color = f3(S.ambient);
                F3 p = o + t * d;
                for (uint32_t k = 0; k < S.n_lights; k++) {
                    LightRegs L = load_light(S.lights + k);
                    F3 l, c; float tmax;
                    if (L.type == RT_LIGHT_DIRECTIONAL) {                // light_source.cpp:20-23
                        l = normalize(L.direction); c = L.intensity; tmax = __builtin_inff();
                    } else {                                             // light_source.cpp:9-14
                        F3 dir = L.position - p;
                        float r = len(dir);
                        c = (float)(1. / (double)(L.attenuation.x + L.attenuation.y * r + L.attenuation.z * r * r)) * L.intensity;
                        l = normalize(dir);
                        tmax = r;
                    }
                    float reflected = dot(l, norma);
                    if (reflected >= 0 && !occluded2(S, p + epsf * l, l, tmax)) color = color + reflected * c;
                }
                ret = color * P.color;
                continue;
            }
            F3 dn = normalize(d);
            F3 refl = dn - (float)(2. * (double)dot(norma, dn)) * norma; // :54,58
            Frame2 &f = frames[fp++];
            f.color = P.color; f.x = o + t * d; f.l = neg(dn); f.norma = norma; f.inside = inside; f.ior = P.ior;
            f.kind = P.kind == RT_MAT_METALLIC ? F2_MUL : F2_DIEL_REFLECTED;
            o = f.x + epsf * refl; d = refl;
            evaluating = true;
        } else {
            if (fp == 0) break;
            Frame2 &f = frames[fp - 1];
            if (f.kind == F2_MUL) { ret = f.color * ret; fp--; continue; }
            float eta1 = 1.f, eta2 = f.ior;                              // :62-65
            if (f.inside) { float tmp = eta1; eta1 = eta2; eta2 = tmp; }
            float nl = dot(f.norma, f.l);
            if (f.kind == F2_DIEL_REFLECTED) {
                float sinTheta2 = eta1 / eta2 * sqrtf(1 - nl * nl);      // :68 (float sqrt under <math.h>)
                if (fabsf(sinTheta2) > 1.) { fp--; continue; }           // total internal reflection: ret stays the reflected colour
                float cosTheta2 = sqrtf(1 - sinTheta2 * sinTheta2);
                F3 refr = (eta1 / eta2) * neg(f.l) + (eta1 / eta2 * nl - cosTheta2) * f.norma;
                f.reflected = ret;
                f.kind = F2_DIEL_REFRACTED;
                o = f.x + epsf * refr; d = refr;
                evaluating = true;
                continue;
            }
            F3 refracted = ret;                                          // :76-83
            if (!f.inside) refracted = refracted * f.color;
            float rr = (eta1 - eta2) / (eta1 + eta2);
            float r0 = (float)((double)rr * (double)rr);                 // pow(x, 2.)
            double om = (double)(1 - nl), om2 = om * om;
            float r = (float)((double)r0 + (double)(1 - r0) * (om2 * om2 * om));
            ret = r * f.reflected + (1 - r) * refracted;
            fp--;
        }
    }
    return ret;
}

__global__ __launch_bounds__(64) void render_hw2_kernel(SceneViewTxt S, RenderView R, float tan_fov_y, uint32_t n_work) {
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
            F3 o, d;
            camera_ray_txt(S, S.tan_fov_x_f, tan_fov_y, R.width, R.height, (float)x, (float)y, o, d); // hw2/src/scene.cpp:90-98
            px = trace_tree2(S, R.ray_depth, o, d);
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
