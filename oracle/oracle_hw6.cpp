// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the hw6 render path (BASELINE.json configs[2]: glTF triangles, DIFFUSE / METALLIC /
// DIELECTRIC materials, Mix{Cosine, FiguresMix}, branching dielectric recursion) in exact-replay semantics.
// glTF scenes contain triangles only (hw6/src/sceneio.cpp:214), so the ELLIPSOID / BOX / PLANE figure types and
// their lights (hw6/src/include/distributions.h:61-105,142-172) are not restated; every figure has
// position (0,0,0) and the identity rotation, whose quaternion arithmetic is nevertheless carried out literally.
// Pinned bit-exact against the compiled hw6 reference (oracle/ref/ref_hw6_scene.cpp, tests/test_oracle_pins.py).
#include "oracle_common.h"
#include <omp.h>

namespace rto6 {
using namespace rto;

static const float T_MAX = 1e4;               // hw6/src/primitives.cpp:11
static const long double eps_ld = 1e-4;       // hw6/src/include/primitives.h:9
static const float PI = acos(-1);             // hw6/src/include/distributions.h:11 (a FLOAT constant)

struct Fig {                                   // hw6/src/include/primitives.h:33-55 (TRIANGLE only)
    V3 position;                               // (0,0,0)
    Quat rotation;                             // identity
    V3 data, data2, data3;
    uint32_t mat = 0, orig = 0;
};
struct Hit { float t; V3 norma; bool inside; };
struct Box { V3 mn, mx; };
struct Counters { uint64_t closest = 0, lightq = 0, boxes = 0, tris = 0; };
static thread_local Counters tl_cnt;

static inline V3 rotate(Quat q, V3 p) { return qtransform(q, p); }

// primitives.cpp:77-86
static inline bool plane_ray(V3 n, V3 o, V3 d, Hit &h) {
    float t = -dot(o, n) / dot(d, n);
    if (t > 0 && t < T_MAX) {
        if (dot(d, n) > 0) { h = Hit{t, neg1(n), true}; return true; }
        h = Hit{t, n, false};
        return true;
    }
    return false;
}
// primitives.cpp:143-164
static inline bool tri_ray_local(const Fig &f, V3 o, V3 d, Hit &h) {
    V3 a = f.data3, b = f.data - a, c = f.data2 - a;
    V3 n = crossr(b, c);
    if (!plane_ray(n, o - a, d, h)) return false;
    V3 p = o - a + h.t * d;
    if (dot(crossr(b, p), n) < 0) return false;
    if (dot(crossr(p, c), n) < 0) return false;
    if (dot(crossr(c - b, p - b), n) < 0) return false;
    return true;
}
// primitives.cpp:13-33
static bool fig_ray(const Fig &f, V3 o, V3 d, Hit &h) {
    tl_cnt.tris++;
    V3 to = rotate(f.rotation, o - f.position), td = rotate(f.rotation, d);
    if (!tri_ray_local(f, to, td, h)) return false;
    h.norma = normalize(rotate(qconj(f.rotation), h.norma));
    return true;
}
// primitives.cpp:92-116 (require_norma == false), :221-223
static inline bool box_ray(V3 s, V3 o, V3 d, float &t, bool &inside) {
    V3 ts1 = (neg1(s) - o) / d, ts2 = (s - o) / d;
    float t1x = smin(ts1.x, ts2.x), t2x = smax(ts1.x, ts2.x);
    float t1y = smin(ts1.y, ts2.y), t2y = smax(ts1.y, ts2.y);
    float t1z = smin(ts1.z, ts2.z), t2z = smax(ts1.z, ts2.z);
    float t1 = smax(smax(t1x, t1y), t1z), t2 = smin(smin(t2x, t2y), t2z);
    if (t1 > t2 || t2 < 0) return false;
    if (t1 < 0) { inside = true; t = t2; } else { inside = false; t = t1; }
    return true;
}
static inline bool aabb_ray(const Box &b, V3 o, V3 d, float &t, bool &inside) {
    tl_cnt.boxes++;
    return box_ray(0.5f * (b.mx - b.mn), o - 0.5f * (b.mn + b.mx), d, t, inside);
}
static void extend(Box &b, V3 p) {
    b.mx.x = smax(b.mx.x, p.x); b.mx.y = smax(b.mx.y, p.y); b.mx.z = smax(b.mx.z, p.z);
    b.mn.x = smin(b.mn.x, p.x); b.mn.y = smin(b.mn.y, p.y); b.mn.z = smin(b.mn.z, p.z);
}
static void extend(Box &b, const Box &o) { extend(b, o.mn); extend(b, o.mx); }
// primitives.cpp:169-199 (TRIANGLE branch + the rotate-8-corners step, carried out literally)
static Box box_of(const Fig &f) {
    Box u;
    u.mn = {smin(f.data3.x, smin(f.data.x, f.data2.x)), smin(f.data3.y, smin(f.data.y, f.data2.y)), smin(f.data3.z, smin(f.data.z, f.data2.z))};
    u.mx = {smax(f.data3.x, smax(f.data.x, f.data2.x)), smax(f.data3.y, smax(f.data.y, f.data2.y)), smax(f.data3.z, smax(f.data.z, f.data2.z))};
    Quat r = qconj(f.rotation);
    Box b;
    b.mn = b.mx = rotate(r, u.mn);
    extend(b, rotate(r, V3{u.mn.x, u.mn.y, u.mx.z}));
    extend(b, rotate(r, V3{u.mn.x, u.mx.y, u.mn.z}));
    extend(b, rotate(r, V3{u.mn.x, u.mx.y, u.mx.z}));
    extend(b, rotate(r, V3{u.mx.x, u.mn.y, u.mn.z}));
    extend(b, rotate(r, V3{u.mx.x, u.mn.y, u.mx.z}));
    extend(b, rotate(r, V3{u.mx.x, u.mx.y, u.mn.z}));
    extend(b, rotate(r, V3{u.mx.x, u.mx.y, u.mx.z}));
    b.mn = b.mn + f.position;
    b.mx = b.mx + f.position;
    return b;
}
static float surf(const Box &b) { V3 d = b.mx - b.mn; return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }

// hw6/src/include/bvh.h — identical to hw8's except that the sort key is Figure::position (:61-63),
// which is (0,0,0) for every glTF triangle: the comparator is always false and std::sort leaves whatever
// permutation introsort produces on all-equal keys.
struct Node { Box aabb; uint32_t left = 0, right = 0, first = 0, last = 0; };
struct Bvh {
    std::vector<Node> nodes;
    uint32_t root = 0, depth = 0;
    static std::pair<float, uint32_t> best_split(std::vector<Fig> &figs, uint32_t first, uint32_t last) {
        std::vector<float> scores(last - first, 0);
        Box pre = box_of(figs[first]);
        for (size_t i = 1; i < last - first; i++) { scores[i] = surf(pre) * i; extend(pre, box_of(figs[first + i])); }
        Box suf = box_of(figs[last - 1]);
        for (size_t i = last - first - 1; i >= 1; i--) { scores[i] += surf(suf) * ((last - first) - i); extend(suf, box_of(figs[first + i - 1])); }
        std::pair<float, uint32_t> ans = {scores[1], first + 1};
        for (size_t i = 2; i < last - first; i++)
            if (scores[i] < ans.first) ans = {scores[i], (uint32_t)(first + i)};
        return ans;
    }
    static void half_split(std::vector<Fig> &figs, uint32_t first, uint32_t last, int axis) {
        if (axis == 0) std::sort(figs.begin() + first, figs.begin() + last, [](const Fig &l, const Fig &r) { return l.position.x < r.position.x; });
        else if (axis == 1) std::sort(figs.begin() + first, figs.begin() + last, [](const Fig &l, const Fig &r) { return l.position.y < r.position.y; });
        else std::sort(figs.begin() + first, figs.begin() + last, [](const Fig &l, const Fig &r) { return l.position.z < r.position.z; });
    }
    uint32_t build(std::vector<Fig> &figs, uint32_t first, uint32_t last, uint32_t d = 1) {
        if (d > depth) depth = d;
        Node cur; cur.first = first; cur.last = last;
        Box aabb;
        if (first < last) aabb = box_of(figs[first]);
        for (uint32_t i = first + 1; i < last; i++) extend(aabb, box_of(figs[i]));
        cur.aabb = aabb;
        uint32_t pos = (uint32_t)nodes.size();
        nodes.push_back(cur);
        if (last - first <= 1) return pos;
        half_split(figs, first, last, 0); auto sx = best_split(figs, first, last);
        half_split(figs, first, last, 1); auto sy = best_split(figs, first, last);
        half_split(figs, first, last, 2); auto sz = best_split(figs, first, last);
        float best = smin(sx.first, smin(sy.first, sz.first));
        if (best >= surf(aabb) * (last - first)) return pos;
        uint32_t mid;
        if (best == sx.first) { mid = sx.second; half_split(figs, first, last, 0); }
        else if (best == sy.first) { mid = sy.second; half_split(figs, first, last, 1); }
        else { mid = sz.second; half_split(figs, first, last, 2); }
        uint32_t l = build(figs, first, mid, d + 1); nodes[pos].left = l;
        uint32_t r = build(figs, mid, last, d + 1); nodes[pos].right = r;
        return pos;
    }
    void init(std::vector<Fig> &figs, uint32_t n) { nodes.clear(); depth = 0; root = build(figs, 0, n); }
    bool intersect(const std::vector<Fig> &figs, uint32_t pos, V3 o, V3 d, bool have_best, float cur_best, Hit &out, int &idx) const {
        const Node &cur = nodes[pos];
        float t; bool inside;
        if (!aabb_ray(cur.aabb, o, d, t, inside)) return false;
        if (have_best && cur_best < t && !inside) return false;
        bool found = false;
        if (cur.left == 0) {
            for (uint32_t i = cur.first; i < cur.last; i++) {
                Hit h;
                if (fig_ray(figs[i], o, d, h) && (!found || h.t < out.t)) { out = h; idx = (int)i; found = true; }
            }
            return found;
        }
        Hit lh; int li = -1;
        bool lf = intersect(figs, cur.left, o, d, have_best, cur_best, lh, li);
        if (lf) { out = lh; idx = li; found = true; }
        if (lf && (!have_best || lh.t < cur_best)) { cur_best = lh.t; have_best = true; }
        Hit rh; int ri = -1;
        bool rf = intersect(figs, cur.right, o, d, have_best, cur_best, rh, ri);
        if (rf && (!found || rh.t < out.t)) { out = rh; idx = ri; found = true; }
        return found;
    }
};

typedef std::uniform_real_distribution<float> U01;
typedef std::normal_distribution<float> N01;

// distributions.h:42-58
static V3 cosine_sample(N01 &n01, rng_t &rng, V3 n) {
    float a = n01(rng), b = n01(rng), c = n01(rng);
    V3 d = normalize(V3{a, b, c});
    d = d + n;
    float l = len(d);
    const float ceps = 1e-9;
    if (l <= ceps || dot(d, n) <= ceps || std::isnan(l)) return n;
    return (float)(1. / l) * d;
}
static float cosine_pdf(V3 n, V3 d) { return smax(0.f, dot(d, n) / PI); }

struct TriLight {                              // distributions.h:107-143
    float pointProb;
    Fig fig;
    explicit TriLight(const Fig &f) : fig(f) {
        V3 a = fig.data3, b = fig.data - a, c = fig.data2 - a;
        pointProb = 1.0 / (0.5 * len(crossr(b, c)));
    }
    float pdfOne(V3 x, V3 d, V3 y, V3 yn) const { return pointProb * len2(x - y) / std::fabs((double)dot(d, yn)); }
    V3 sample(U01 &u01, rng_t &rng, V3 x) const {
        V3 a = fig.data3, b = fig.data - a, c = fig.data2 - a;
        float u = u01(rng);
        float v = u01(rng);
        if (u + v > 1.) { u = 1 - u; v = 1 - v; }
        V3 point = fig.position + rotate(qconj(fig.rotation), a + u * b + v * c);
        return normalize(point - x);
    }
};

struct Scene {
    std::vector<Fig> figs;
    std::vector<rt_material> mats;
    std::vector<TriLight> lights;
    Bvh bvh, lbvh;
    V3 camPos, camRight, camUp, camFwd, bg;
    float fovY = 0;
    int n_components = 1;
    int width = 0, height = 0, samples = 1, rayDepth = 6;

    void init() {
        // scene.cpp:18-23: partition non-planes first (all figures are triangles -> every predicate is true)
        std::partition(figs.begin(), figs.end(), [](const Fig &) { return true; });
        bvh.init(figs, (uint32_t)figs.size());
        std::vector<Fig> copy = figs;                                              // distributions.h:180 (by value)
        size_t n = std::partition(copy.begin(), copy.end(), [this](const Fig &f) {
            const rt_material &m = mats[f.mat];
            return !(m.emission[0] == 0 && m.emission[1] == 0 && m.emission[2] == 0);
        }) - copy.begin();
        lbvh.init(copy, (uint32_t)n);
        for (size_t i = 0; i < n; i++) lights.push_back(TriLight(copy[i]));
        n_components = lights.empty() ? 1 : 2;                                     // scene.cpp:8-16
    }
    float pdf_one(const TriLight &tl, V3 x, V3 d) const {                          // distributions.h:212-237
        Hit h;
        if (!fig_ray(tl.fig, x, d, h)) return 0.;
        if (std::isnan(h.t)) return INFINITY;
        V3 y = x + h.t * d;
        return tl.pdfOne(x, d, y, h.norma);
    }
    float total_pdf(uint32_t pos, V3 x, V3 d) const {                              // :239-256
        const Node &cur = lbvh.nodes[pos];
        float t; bool inside;
        if (!aabb_ray(cur.aabb, x, d, t, inside)) return 0;
        if (cur.left == 0) {
            float result = 0;
            for (uint32_t i = cur.first; i < cur.last; i++) result += pdf_one(lights[i], x, d);
            return result;
        }
        float l = total_pdf(cur.left, x, d), r = total_pdf(cur.right, x, d);
        return l + r;
    }
    V3 mix_sample(U01 &u01, N01 &n01, rng_t &rng, V3 x, V3 n) const {              // :283-290
        int k = u01(rng) * (size_t)n_components;
        if (k == 0) return cosine_sample(n01, rng, n);
        int li = u01(rng) * lights.size();                                         // :199-208
        return lights[li].sample(u01, rng, x);
    }
    float mix_pdf(V3 x, V3 n, V3 d) const {                                        // :292-302
        float ans = 0;
        ans += cosine_pdf(n, d);
        if (n_components == 2) { tl_cnt.lightq++; ans += total_pdf(0, x, d) / lights.size(); }
        return ans / (size_t)n_components;
    }
    // scene.cpp:47-105
    V3 get_color(U01 &u01, N01 &n01, rng_t &rng, V3 ro, V3 rd, int recLimit) const {
        if (recLimit == 0) return V3{0., 0., 0.};
        Hit h; int idx = -1;
        tl_cnt.closest++;
        if (!bvh.intersect(figs, bvh.root, ro, rd, false, 0.f, h, idx)) return bg;
        const rt_material &m = mats[figs[idx].mat];
        V3 emission{m.emission[0], m.emission[1], m.emission[2]}, color{m.base_color[0], m.base_color[1], m.base_color[2]};
        V3 x = ro + h.t * rd;
        V3 norma = h.norma;
        if (m.kind == RT_MAT_DIFFUSE) {
            V3 xo = x + (float)eps_ld * norma;
            V3 d = mix_sample(u01, n01, rng, xo, norma);
            if (dot(d, norma) < 0) return emission;
            float pdf = mix_pdf(xo, norma, d);
            V3 no = x + (float)eps_ld * d;
            float k = 1. / (PI * pdf) * dot(d, norma);
            return emission + (k * color) * get_color(u01, n01, rng, no, d, recLimit - 1);
        } else if (m.kind == RT_MAT_METALLIC) {
            V3 dn = normalize(rd);
            V3 refl = dn - (float)(2. * dot(norma, dn)) * norma;
            V3 no = ro + h.t * rd + (float)eps_ld * refl;
            return emission + color * get_color(u01, n01, rng, no, refl, recLimit - 1);
        } else {
            V3 dn = normalize(rd);
            V3 refl = dn - (float)(2. * dot(norma, dn)) * norma;
            V3 no = ro + h.t * rd + (float)eps_ld * refl;
            V3 reflected = get_color(u01, n01, rng, no, refl, recLimit - 1);
            float eta1 = 1., eta2 = m.ior;
            if (h.inside) std::swap(eta1, eta2);
            V3 l = neg1(normalize(rd));
            float sinTheta2 = eta1 / eta2 * std::sqrt((double)(1 - dot(norma, l) * dot(norma, l)));
            if (std::fabs((double)sinTheta2) > 1.) return emission + reflected;
            float r0 = std::pow((double)((eta1 - eta2) / (eta1 + eta2)), 2.);
            float r = r0 + (1 - r0) * std::pow((double)(1 - dot(norma, l)), 5.);
            if (u01(rng) < r) return emission + reflected;
            float cosTheta2 = std::sqrt((double)(1 - sinTheta2 * sinTheta2));
            V3 refr = (eta1 / eta2) * neg1(l) + (eta1 / eta2 * dot(norma, l) - cosTheta2) * norma;
            V3 fo = ro + h.t * rd + (float)eps_ld * refr;
            V3 refracted = get_color(u01, n01, rng, fo, refr, recLimit - 1);
            if (!h.inside) refracted = refracted * color;
            return emission + refracted;
        }
    }
    // scene.cpp:107-127 (camera direction is NOT normalised in hw6)
    V3 get_pixel(rng_t &rng, int x, int y) const {
        U01 u01(0.0, 1.0);
        N01 n01(0.0, 1.0);
        V3 color{0, 0, 0};
        for (int s = 0; s < samples; s++) {
            float nx = x + u01(rng);
            float ny = y + u01(rng);
            float tanFovY = std::tan((double)(fovY / 2));
            float tanFovX = tanFovY * width / height;
            float cx = tanFovX * (2 * nx / width - 1);
            float cy = tanFovY * (2 * ny / height - 1);
            color = color + get_color(u01, n01, rng, camPos, cx * camRight - cy * camUp + camFwd, rayDepth);
        }
        return (float)(1.0 / samples) * color;
    }
};
static V3 v3(const float *p) { return {p[0], p[1], p[2]}; }
} // namespace rto6

using namespace rto6;
extern "C" {
struct rto_counters { uint64_t closest, lightq, boxes, tris; };

void *rto_hw6_create(const rt_scene_desc *d) {
    Scene *s = new Scene();
    s->figs.resize(d->n_triangles);
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        Fig &f = s->figs[i];
        f.data = v3(d->positions + 9 * i); f.data2 = v3(d->positions + 9 * i + 3); f.data3 = v3(d->positions + 9 * i + 6);
        f.mat = d->material_index[i]; f.orig = i;
    }
    s->mats.assign(d->materials, d->materials + d->n_materials);
    s->camPos = v3(d->camera.position); s->camRight = v3(d->camera.right); s->camUp = v3(d->camera.up); s->camFwd = v3(d->camera.forward);
    s->fovY = d->camera.fov_y; s->bg = v3(d->bg_color);
    s->init();
    return s;
}
void rto_hw6_destroy(void *p) { delete (Scene *)p; }
uint32_t rto_hw6_num_lights(void *p) { return (uint32_t)((Scene *)p)->lights.size(); }
void rto_hw6_light_order(void *p, uint32_t *out) { Scene *s = (Scene *)p; for (size_t i = 0; i < s->lights.size(); i++) out[i] = s->lights[i].fig.orig; }
void rto_hw6_figure_order(void *p, uint32_t *out) { Scene *s = (Scene *)p; for (size_t i = 0; i < s->figs.size(); i++) out[i] = s->figs[i].orig; }
void rto_hw6_bvh_stats(void *p, uint32_t *out4) {
    Scene *s = (Scene *)p;
    out4[0] = (uint32_t)s->bvh.nodes.size(); out4[1] = s->bvh.depth; out4[2] = (uint32_t)s->lbvh.nodes.size(); out4[3] = s->lbvh.depth;
}
// Test hook for throughput mode (rt_render_params.sample_streams): stream k of a pixel is the reference's per-pixel loop with the engine
// seeded y*W+x + k*W*H instead of y*W+x (hw6/src/sceneio.cpp:280-284 seeds with y*W+x).
static uint32_t g_seed_offset6 = 0;
void rto_hw6_set_seed_offset(uint32_t off) { g_seed_offset6 = off; }
int rto_hw6_render(void *p, int width, int height, int samples, int ray_depth, int x0, int y0, int w, int h, float *out_rgb, uint8_t *out8,
                   int nthreads, rto_counters *cnt) {
    Scene *s = (Scene *)p;
    s->width = width; s->height = height; s->samples = samples; s->rayDepth = ray_depth > 0 ? ray_depth : 6;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads) reduction(+ : c0, c1, c2, c3)
    for (int j = 0; j < w * h; j++) {
        tl_cnt = Counters{};
        int x = x0 + j % w, y = y0 + j / w;
        rng_t rng((uint32_t)(y * width + x) + g_seed_offset6);                     // hw6/src/sceneio.cpp:280-284
        V3 px = s->get_pixel(rng, x, y);
        if (out_rgb) { out_rgb[3 * j] = px.x; out_rgb[3 * j + 1] = px.y; out_rgb[3 * j + 2] = px.z; }
        if (out8) to_extern(gamma_corrected(aces_tonemap(px)), out8 + 3 * j);
        c0 += tl_cnt.closest; c1 += tl_cnt.lightq; c2 += tl_cnt.boxes; c3 += tl_cnt.tris;
    }
    if (cnt) { cnt->closest = c0; cnt->lightq = c1; cnt->boxes = c2; cnt->tris = c3; }
    return 0;
}
}
