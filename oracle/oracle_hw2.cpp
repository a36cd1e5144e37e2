// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the hw2 snapshot: a deterministic Whitted-style tracer over analytic primitives with point and
// directional lights, shadow rays, mirror reflection and Fresnel-weighted refraction (hw2/src/scene.cpp:8-98,
// hw2/src/primitives.cpp:26-137, hw2/src/light_source.cpp:9-23).
//
// hw2 (like hw1) includes <math.h>, not <cmath>: libstdc++'s <math.h> puts the float overloads into the global
// namespace, so here `sqrt(d)`, `tan(fov/2)`, `fabs(x)` on floats are the FLOAT functions and `pow(float, float)` is
// powf — unlike hw3+, where the same unqualified calls resolve to the double C functions (checked on the compiled
// programs: hw1/hw2 import sqrtf/tanf/powf, hw3 imports sqrt/tan/pow).  pow(float, double) still promotes to double.
#include "oracle_common.h"
#include <omp.h>

namespace rto2 {
using namespace rto;

struct Prim { int type; V3 data, position; Quat rotation; V3 color; int kind; float ior; };
struct Light { int type; V3 intensity, position, attenuation, direction; };
struct Hit { float t; V3 norma; bool inside; };

static inline float lenf(V3 a) { return sqrtf(len2(a)); }                                  // vec3.cpp:37-39
static inline V3 normalizef(V3 a) { return (float)(1. / (double)lenf(a)) * a; }            // vec3.cpp:52-54

// hw2/src/primitives.cpp:37-56 — all float
static bool smallest_root(float a, float b, float c, float &t, bool &inside) {
    float d = b * b - 4 * a * c;
    if (d <= 0) return false;
    float x1 = (-b - sqrtf(d)) / (2 * a);
    float x2 = (-b + sqrtf(d)) / (2 * a);
    if (x1 > x2) std::swap(x1, x2);
    if (x2 < 0) return false;
    if (x1 < 0) { t = x2; inside = true; } else { t = x1; inside = false; }
    return true;
}

// Figure::intersect, hw2/src/primitives.cpp:26-35 + rawIntersect of the three shapes (:58-137)
static bool prim_ray(const Prim &f, V3 o, V3 d, Hit &h) {
    V3 to = qtransform(f.rotation, o - f.position), td = qtransform(f.rotation, d);
    if (f.type == RT_PRIM_ELLIPSOID) {
        V3 r = f.data;
        float c = len2(to / r) - 1;
        float b = 2. * dot(to / r, td / r);
        float a = len2(td / r);
        float t; bool inside;
        if (!smallest_root(a, b, c, t, inside)) return false;
        V3 point = to + t * td;
        V3 n = point / (r * r);
        if (inside) n = neg1(n);
        h = Hit{t, normalizef(n), inside};
    } else if (f.type == RT_PRIM_PLANE) {
        V3 n = f.data; // normalised by Plane::Plane at load
        float t = -dot(to, n) / dot(td, n);
        if (!(t > 0)) return false;
        h = dot(td, n) > 0 ? Hit{t, neg1(n), true} : Hit{t, n, false};
    } else {
        V3 s = f.data;
        V3 ts1 = (neg1(s) - to) / td, ts2 = (s - to) / td;
        float t1x = smin(ts1.x, ts2.x), t2x = smax(ts1.x, ts2.x);
        float t1y = smin(ts1.y, ts2.y), t2y = smax(ts1.y, ts2.y);
        float t1z = smin(ts1.z, ts2.z), t2z = smax(ts1.z, ts2.z);
        float t1 = smax(smax(t1x, t1y), t1z), t2 = smin(smin(t2x, t2y), t2z);
        if (t1 > t2 || t2 < 0) return false;
        float t; bool inside;
        if (t1 < 0) { inside = true; t = t2; } else { inside = false; t = t1; }
        V3 p = to + t * td;
        V3 n = p / s;
        float mx = smax(smax(fabsf(n.x), fabsf(n.y)), fabsf(n.z));
        if (fabsf(n.x) != mx) n.x = 0;
        if (fabsf(n.y) != mx) n.y = 0;
        if (fabsf(n.z) != mx) n.z = 0;
        if (inside) n = neg1(n);
        h = Hit{t, n, inside};
    }
    h.norma = normalizef(qtransform(qconj(f.rotation), h.norma));
    return true;
}

struct Scene2 {
    std::vector<Prim> figs;
    std::vector<Light> lights;
    V3 camPos, camRight, camUp, camFwd, bg, ambient;
    float fovX = 0;
    int width = 0, height = 0, rayDepth = 1;

    // hw2/src/scene.cpp:8-28
    bool intersect(V3 o, V3 d, float tmax, Hit &best, int &pos) const {
        pos = -1;
        for (int i = 0; i < (int)figs.size(); i++) {
            Hit h;
            if (prim_ray(figs[i], o, d, h) && h.t <= tmax && (pos == -1 || h.t < best.t)) { best = h; pos = i; }
        }
        return pos != -1;
    }
    // hw2/src/light_source.cpp:9-23
    void light_at(const Light &L, V3 p, V3 &l, V3 &c, float &tmax) const {
        if (L.type == RT_LIGHT_DIRECTIONAL) { l = normalizef(L.direction); c = L.intensity; tmax = INFINITY; return; }
        V3 direction = L.position - p;
        float r = lenf(direction);
        c = (float)(1. / (double)(L.attenuation.x + L.attenuation.y * r + L.attenuation.z * r * r)) * L.intensity;
        l = normalizef(direction);
        tmax = r;
    }
    // hw2/src/scene.cpp:30-84
    V3 get_color(V3 ro, V3 rd, int recLimit) const {
        if (recLimit == 0) return V3{0., 0., 0.};
        Hit h; int pos;
        if (!intersect(ro, rd, INFINITY, h, pos)) return bg;
        const Prim &f = figs[pos];
        float t = h.t; V3 norma = h.norma;
        if (f.kind == RT_MAT_DIFFUSE) {
            V3 color = ambient;
            for (const Light &L : lights) {
                V3 p = ro + t * rd;
                V3 l, c; float tmax;
                light_at(L, p, l, c, tmax);
                float reflected = dot(l, norma);
                Hit sh; int spos;
                if (reflected >= 0 && !intersect(p + (float)0.0001 * l, l, tmax, sh, spos)) color = color + reflected * c;
            }
            return color * f.color;
        }
        V3 dn = normalizef(rd);
        V3 reflDir = dn - (float)(2. * (double)dot(norma, dn)) * norma;
        V3 reflO = ro + t * rd + (float)0.0001 * reflDir;
        if (f.kind == RT_MAT_METALLIC) return f.color * get_color(reflO, reflDir, recLimit - 1);
        V3 reflected = get_color(reflO, reflDir, recLimit - 1);
        float eta1 = 1., eta2 = f.ior;
        if (h.inside) std::swap(eta1, eta2);
        V3 l = neg1(normalizef(rd));
        float nl = dot(norma, l);
        float sinTheta2 = eta1 / eta2 * sqrtf(1 - nl * nl);
        if (fabsf(sinTheta2) > 1.) return reflected;
        float cosTheta2 = sqrtf(1 - sinTheta2 * sinTheta2);
        V3 refrDir = (eta1 / eta2) * neg1(l) + (eta1 / eta2 * nl - cosTheta2) * norma;
        V3 refracted = get_color(ro + t * rd + (float)0.0001 * refrDir, refrDir, recLimit - 1);
        if (!h.inside) refracted = refracted * f.color;
        float r0 = std::pow((double)((eta1 - eta2) / (eta1 + eta2)), 2.);
        float r = (double)r0 + (double)(1 - r0) * std::pow((double)(1 - nl), 5.);
        return r * reflected + (1 - r) * refracted;
    }
    // hw2/src/scene.cpp:90-98 — tan on a float is tanf here
    void camera_ray(int x, int y, V3 &o, V3 &d) const {
        float tanFovX = tanf(fovX / 2);
        float tanFovY = tanFovX * height / width;
        float nx = tanFovX * (2 * (x + 0.5) / width - 1);
        float ny = tanFovY * (2 * (y + 0.5) / height - 1);
        o = camPos;
        d = nx * camRight - ny * camUp + camFwd;
    }
};

// hw2/src/color.cpp:13-16: pow(float, float) is powf under <math.h>
static inline V3 gamma_corrected_f(V3 x) {
    float gamma = 1. / 2.2;
    return {powf(x.x, gamma), powf(x.y, gamma), powf(x.z, gamma)};
}
static V3 v3(const float *p) { return {p[0], p[1], p[2]}; }
} // namespace rto2

using namespace rto2;
extern "C" {
void *rto_hw2_create(const rt_scene_desc *d) {
    Scene2 *s = new Scene2();
    for (uint32_t i = 0; i < d->n_primitives; i++) {
        const rt_primitive &p = d->primitives[i];
        s->figs.push_back(Prim{p.type, v3(p.data), v3(p.position), Quat{v3(p.rotation), p.rotation[3]}, v3(p.color), p.kind, p.ior});
    }
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const rt_light &L = d->lights[i];
        s->lights.push_back(Light{L.type, v3(L.intensity), v3(L.position), v3(L.attenuation), v3(L.direction)});
    }
    s->camPos = v3(d->camera.position); s->camRight = v3(d->camera.right); s->camUp = v3(d->camera.up); s->camFwd = v3(d->camera.forward);
    s->fovX = d->camera.fov_x; s->bg = v3(d->bg_color); s->ambient = v3(d->ambient_light);
    return s;
}
void rto_hw2_destroy(void *p) { delete (Scene2 *)p; }

// Rectangle [x0,x0+w) x [y0,y0+h) of the width x height frame; out_rgb = linear radiance, out8 = the program's bytes
// (hw2/src/sceneio.cpp:150-162: aces, gamma, round(255 x)).
int rto_hw2_render(void *p, int width, int height, int ray_depth, int x0, int y0, int w, int h, float *out_rgb, uint8_t *out8, int nthreads) {
    Scene2 *s = (Scene2 *)p;
    s->width = width; s->height = height; s->rayDepth = ray_depth;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
    for (int j = 0; j < w * h; j++) {
        int x = x0 + j % w, y = y0 + j / w;
        V3 o, d;
        s->camera_ray(x, y, o, d);
        V3 px = s->get_color(o, d, s->rayDepth);
        if (out_rgb) { out_rgb[3 * j] = px.x; out_rgb[3 * j + 1] = px.y; out_rgb[3 * j + 2] = px.z; }
        if (out8) to_extern(gamma_corrected_f(aces_tonemap(px)), out8 + 3 * j);
    }
    return 0;
}
}
