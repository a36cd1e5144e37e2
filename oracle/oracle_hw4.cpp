// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the hw4 snapshot: hw3's path tracer over analytic primitives with importance sampling —
// Mix{Cosine, Mix{BoxLight | EllipsoidLight ...}} (hw4/src/scene.cpp:10-122, hw4/src/include/distributions.h:13-204).
//
// hw4 still draws every random number of the frame from ONE file-static minstd_rand (hw4/src/scene.cpp:5-6), and each
// distribution object owns its own std::normal_distribution (whose cached second value survives between calls:
// Cosine's at distributions.h:48, every EllipsoidLight's at :156).  seed_mode 0 replays exactly that, sequentially
// (pinned bit for bit against the compiled hw4 sources); seed_mode 1 gives every pixel a fresh engine(y*W+x) and fresh
// distribution objects — the form a parallel machine can run, used to check the GPU.
#include "oracle_txt_prims.h"
#include <memory>
#include <omp.h>

namespace rto4 {
using namespace rtot;

typedef std::uniform_real_distribution<float> U01;
typedef std::normal_distribution<float> N01;
static const float PI = std::acos(-1); // distributions.h:9

struct Sampler { // all mutable sampling state of one replay stream
    rng_t rng;
    U01 u01{0.0, 1.0};            // scene.cpp:6 and the (stateless) copies inside Mix / BoxLight
    N01 cosine_n01{0.f, 1.f};     // Cosine::n01
    std::vector<N01> light_n01;   // EllipsoidLight::n01, one per light (unused for boxes)
};

struct Scene4 {
    std::vector<Prim> figs;
    std::vector<int> lights; // indices of emissive BOX / ELLIPSOID figures in figure order (scene.cpp:12-21)
    V3 camPos, camRight, camUp, camFwd, bg;
    float fovX = 0;
    int width = 0, height = 0, samples = 1, rayDepth = 1;

    bool intersect(V3 o, V3 d, Hit &best, int &pos) const { // scene.cpp:31-49
        pos = -1;
        for (int i = 0; i < (int)figs.size(); i++) {
            Hit h;
            if (prim_ray3(figs[i], o, d, h, true) && h.t <= INFINITY && (pos == -1 || h.t < best.t)) { best = h; pos = i; }
        }
        return pos != -1;
    }

    // ---- distributions.h ---------------------------------------------------------------------------------------
    V3 cosine_sample(Sampler &S, V3 n) const { // :55-67
        float a = S.cosine_n01(S.rng), b = S.cosine_n01(S.rng), c = S.cosine_n01(S.rng);
        V3 d = normalize(V3{a, b, c});
        d = d + n;
        float l = len(d);
        if (l <= 1e-9f || dot(d, n) <= 1e-9f || std::isnan(l)) return n;
        return (float)(1. / (double)l) * d;
    }
    float cosine_pdf(V3 n, V3 d) const { return smax(0.f, dot(d, n) / PI); } // :69-72
    float pdf_one(const Prim &f, V3 x, V3 d, V3 y, V3 yn) const {
        if (f.type == RT_PRIM_BOX) {                                         // :115-118
            float sx = f.data.x, sy = f.data.y, sz = f.data.z;
            float sTotal = 8 * (sy * sz + sx * sz + sx * sy);
            return (double)len2(x - y) / ((double)sTotal * std::fabs((double)dot(d, yn)));
        }
        V3 r = f.data;                                                       // :159-164
        V3 n = qtransform(f.rotation, y - f.position) / r;
        float pointProb = 1. / (double)(4 * PI * len(V3{n.x * r.y * r.z, r.x * n.y * r.z, r.x * r.y * n.z}));
        return (double)(pointProb * len2(x - y)) / std::fabs((double)dot(d, yn));
    }
    float light_pdf(const Prim &f, V3 x, V3 d) const { // FigureLight::pdf :85-107
        Hit h1;
        if (!prim_ray3(f, x, d, h1, true)) return 0.;
        if (std::isnan(h1.t)) return INFINITY;
        V3 y = x + h1.t * d;
        float ans = pdf_one(f, x, d, y, h1.norma);
        Hit h2;
        if (!prim_ray3(f, x + (float)((double)h1.t + 0.0001) * d, d, h2, true)) return ans;
        V3 y2 = x + (float)((double)h1.t + 0.0001 + (double)h2.t) * d;
        return ans + pdf_one(f, x, d, y2, h2.norma);
    }
    V3 light_sample(Sampler &S, int li, V3 x) const {
        const Prim &f = figs[lights[li]];
        if (f.type == RT_PRIM_BOX) {                                         // :125-151
            float sx = f.data.x, sy = f.data.y, sz = f.data.z;
            float wx = sy * sz, wy = sx * sz, wz = sx * sy;
            for (;;) {
                float u = S.u01(S.rng) * (wx + wy + wz);
                float flipSign = (double)S.u01(S.rng) > 0.5 ? 1 : -1;
                // Vec3(a, b, c) is a constructor call: g++ evaluates its arguments right to left, so the LAST
                // coordinate's random number is drawn first (pinned against the compiled reference).
                V3 point;
                if (u < wx) { float c = (2 * S.u01(S.rng) - 1) * sz; float b = (2 * S.u01(S.rng) - 1) * sy; point = V3{flipSign * sx, b, c}; }
                else if (u < wx + wy) { float c = (2 * S.u01(S.rng) - 1) * sz; float a = (2 * S.u01(S.rng) - 1) * sx; point = V3{a, flipSign * sy, c}; }
                else { float b = (2 * S.u01(S.rng) - 1) * sy; float a = (2 * S.u01(S.rng) - 1) * sx; point = V3{a, b, flipSign * sz}; }
                V3 actual = qtransform(qconj(f.rotation), point) + f.position;
                Hit h;
                if (prim_ray3(f, x, normalize(actual - x), h, true)) return normalize(actual - x);
            }
        }
        V3 r = f.data;                                                       // :169-180
        N01 &n01 = S.light_n01[li];
        for (;;) {
            float a = n01(S.rng), b = n01(S.rng), c = n01(S.rng);
            V3 point = r * normalize(V3{a, b, c});
            V3 actual = qtransform(qconj(f.rotation), point) + f.position;
            Hit h;
            if (prim_ray3(f, x, normalize(actual - x), h, true)) return normalize(actual - x);
        }
    }
    V3 mix_sample(Sampler &S, V3 x, V3 n) const { // Mix::sample :194-197, outer then inner
        size_t comps = lights.empty() ? 1 : 2;
        int distNum = S.u01(S.rng) * comps;
        if (distNum == 0) return cosine_sample(S, n);
        int li = S.u01(S.rng) * lights.size();
        return light_sample(S, li, x);
    }
    float mix_pdf(V3 x, V3 n, V3 d) const { // Mix::pdf :199-205
        float ans = 0;
        ans += cosine_pdf(n, d);
        if (lights.empty()) return ans / (size_t)1;
        float inner = 0;
        for (int idx : lights) inner += light_pdf(figs[idx], x, d);
        ans += inner / lights.size();
        return ans / (size_t)2;
    }

    // ---- scene.cpp:51-112 --------------------------------------------------------------------------------------
    V3 get_color(Sampler &S, V3 ro, V3 rd, int recLimit) const {
        if (recLimit == 0) return V3{0., 0., 0.};
        Hit h; int pos;
        if (!intersect(ro, rd, h, pos)) return bg;
        const Prim &f = figs[pos];
        float t = h.t; V3 norma = h.norma;
        V3 x = ro + t * rd;
        if (f.kind == RT_MAT_DIFFUSE) {
            V3 d = mix_sample(S, x + (float)0.0001 * norma, norma);
            if (dot(d, norma) < 0) return f.emission;
            float pdf = mix_pdf(x + (float)0.0001 * norma, norma, d);
            V3 inner = get_color(S, x + (float)0.0001 * d, d, recLimit - 1);
            return f.emission + (float)(1. / (double)(PI * pdf) * (double)dot(d, norma)) * f.color * inner;
        }
        V3 dn = normalize(rd);
        V3 refl = dn - (float)(2. * dot(norma, dn)) * norma;
        V3 o = ro + t * rd + (float)0.0001 * refl;
        if (f.kind == RT_MAT_METALLIC) return f.emission + f.color * get_color(S, o, refl, recLimit - 1);
        V3 reflected = get_color(S, o, refl, recLimit - 1);
        float eta1 = 1., eta2 = f.ior;
        if (h.inside) std::swap(eta1, eta2);
        V3 l = neg1(normalize(rd));
        float sinTheta2 = eta1 / eta2 * std::sqrt((double)(1 - dot(norma, l) * dot(norma, l)));
        if (std::fabs((double)sinTheta2) > 1.) return f.emission + reflected;
        float r0 = std::pow((double)((eta1 - eta2) / (eta1 + eta2)), 2.);
        float r = r0 + (1 - r0) * std::pow((double)(1 - dot(norma, l)), 5.);
        if (S.u01(S.rng) < r) return f.emission + reflected;
        float cosTheta2 = std::sqrt((double)(1 - sinTheta2 * sinTheta2));
        V3 refr = (eta1 / eta2) * neg1(l) + (eta1 / eta2 * dot(norma, l) - cosTheta2) * norma;
        V3 refracted = get_color(S, ro + t * rd + (float)0.0001 * refr, refr, recLimit - 1);
        if (!h.inside) refracted = refracted * f.color;
        return f.emission + refracted;
    }
    void camera_ray(float x, float y, V3 &o, V3 &d) const { // scene.cpp:124-132: all float, no half-pixel offset
        float tanFovX = std::tan((double)(fovX / 2));
        float tanFovY = tanFovX * height / width;
        float nx = tanFovX * (2 * x / width - 1);
        float ny = tanFovY * (2 * y / height - 1);
        o = camPos;
        d = nx * camRight - ny * camUp + camFwd;
    }
    V3 get_pixel(Sampler &S, int x, int y) const { // scene.cpp:114-122
        V3 color{0, 0, 0};
        for (int s = 0; s < samples; s++) {
            float nx = x + S.u01(S.rng);
            float ny = y + S.u01(S.rng);
            V3 o, d;
            camera_ray(nx, ny, o, d);
            color = color + get_color(S, o, d, rayDepth);
        }
        return (float)(1.0 / samples) * color;
    }
};
} // namespace rto4

using namespace rto4;
extern "C" {
void *rto_hw4_create(const rt_scene_desc *d) {
    Scene4 *s = new Scene4();
    for (uint32_t i = 0; i < d->n_primitives; i++) {
        s->figs.push_back(prim_from_abi(d->primitives[i]));
        const Prim &f = s->figs.back();
        if ((f.emission.x > 0 || f.emission.y > 0 || f.emission.z > 0) && (f.type == RT_PRIM_BOX || f.type == RT_PRIM_ELLIPSOID)) s->lights.push_back((int)i);
    }
    s->camPos = v3(d->camera.position); s->camRight = v3(d->camera.right); s->camUp = v3(d->camera.up); s->camFwd = v3(d->camera.forward);
    s->fovX = d->camera.fov_x; s->bg = v3(d->bg_color);
    return s;
}
void rto_hw4_destroy(void *p) { delete (Scene4 *)p; }
int rto_hw4_num_lights(void *p) { return (int)((Scene4 *)p)->lights.size(); }

// seed_mode 0: one engine + one set of distribution objects for the whole call, pixels in row-major order (the reference);
// seed_mode 1: fresh engine(y*W+x) and fresh distribution objects per pixel (parallel).
int rto_hw4_render(void *p, int width, int height, int samples, int ray_depth, int seed_mode, int x0, int y0, int w, int h,
                   float *out_rgb, uint8_t *out8, int nthreads) {
    Scene4 *s = (Scene4 *)p;
    s->width = width; s->height = height; s->samples = samples; s->rayDepth = ray_depth;
    auto store = [&](int j, V3 px) {
        if (out_rgb) { out_rgb[3 * j] = px.x; out_rgb[3 * j + 1] = px.y; out_rgb[3 * j + 2] = px.z; }
        if (out8) to_extern(gamma_corrected(aces_tonemap(px)), out8 + 3 * j);
    };
    if (seed_mode == 0) {
        Sampler S;
        S.light_n01.assign(s->lights.size(), N01(0.f, 1.f));
        for (int j = 0; j < w * h; j++) store(j, s->get_pixel(S, x0 + j % w, y0 + j / w));
        return 0;
    }
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
    for (int j = 0; j < w * h; j++) {
        int x = x0 + j % w, y = y0 + j / w;
        Sampler S;
        S.rng.seed(y * width + x);
        S.light_n01.assign(s->lights.size(), N01(0.f, 1.f));
        store(j, s->get_pixel(S, x, y));
    }
    return 0;
}
}
