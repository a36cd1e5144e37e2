// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the hw5 snapshot: analytic primitives + TRIANGLE figures (each with its own position/rotation),
// a SAH BVH over the non-plane figures sorted by Figure::position, Mix{Cosine, FiguresMix{Box|Ellipsoid|Triangle lights
// behind their own BVH}} and — from this snapshot on — one engine per pixel, rng_type rng(y*W+x)
// (hw5/src/scene.cpp:8-126, hw5/src/primitives.cpp:12-222, hw5/src/include/bvh.h:18-141,
//  hw5/src/include/distributions.h:15-302, hw5/src/sceneio.cpp:103-123).
// `eps` is a LONG DOUBLE constant in this snapshot (primitives.h:9): (t + eps) is an x87 80-bit sum narrowed to float.
#include "oracle_txt_prims.h"
#include <omp.h>

namespace rto5 {
using namespace rtot;

typedef std::uniform_real_distribution<float> U01;
typedef std::normal_distribution<float> N01;
static const float PI = std::acos(-1);   // distributions.h:11
static const long double eps = 1e-4;     // primitives.h:9
static const float T_MAX = 1e4;          // primitives.cpp:11

struct Fig {
    int type; V3 data, data2, data3, position; Quat rotation; V3 color, emission; int kind; float ior;
    uint32_t load_index;
};
struct Box { V3 mn, mx; };

// intersectPlaneAndRay, primitives.cpp:76-85
static bool plane_ray(V3 n, V3 o, V3 d, Hit &h) {
    float t = -dot(o, n) / dot(d, n);
    if (t > 0 && t < T_MAX) {
        h = dot(d, n) > 0 ? Hit{t, neg1(n), true} : Hit{t, n, false};
        return true;
    }
    return false;
}
// intersectBoxAndRay, primitives.cpp:91-137
static bool box_ray(V3 s, V3 o, V3 d, Hit &h, bool require_norma) {
    V3 ts1 = (neg1(s) - o) / d, ts2 = (s - o) / d;
    float t1x = smin(ts1.x, ts2.x), t2x = smax(ts1.x, ts2.x);
    float t1y = smin(ts1.y, ts2.y), t2y = smax(ts1.y, ts2.y);
    float t1z = smin(ts1.z, ts2.z), t2z = smax(ts1.z, ts2.z);
    float t1 = smax(smax(t1x, t1y), t1z), t2 = smin(smin(t2x, t2y), t2z);
    if (t1 > t2 || t2 < 0) return false;
    float t; bool inside;
    if (t1 < 0) { inside = true; t = t2; } else { inside = false; t = t1; }
    if (!require_norma) { h = Hit{t, V3{}, inside}; return true; }
    V3 p = o + t * d;
    V3 n = p / s;
    float mx = smax(smax((float)std::fabs((double)n.x), (float)std::fabs((double)n.y)), (float)std::fabs((double)n.z));
    if (std::fabs((double)n.x) != mx) n.x = 0;
    if (std::fabs((double)n.y) != mx) n.y = 0;
    if (std::fabs((double)n.z) != mx) n.z = 0;
    if (inside) n = neg1(n);
    h = Hit{t, n, inside};
    return true;
}
// Figure::intersect, primitives.cpp:13-35 and the four intersectAs* (:57-166)
static bool fig_ray(const Fig &f, V3 o, V3 d, Hit &h) {
    V3 to = qtransform(f.rotation, o - f.position), td = qtransform(f.rotation, d);
    bool ok;
    if (f.type == RT_PRIM_ELLIPSOID) {
        V3 r = f.data;
        float c = len2(to / r) - 1;
        float b = 2. * dot(to / r, td / r);
        float a = len2(td / r);
        float t; bool inside;
        ok = smallest_root(a, b, c, t, inside);
        if (ok) {
            V3 point = to + t * td;
            V3 n = point / (r * r);
            if (inside) n = neg1(n);
            h = Hit{t, normalize(n), inside};
        }
    } else if (f.type == RT_PRIM_PLANE) ok = plane_ray(f.data, to, td, h);
    else if (f.type == RT_PRIM_BOX) ok = box_ray(f.data, to, td, h, true);
    else {                                                                 // :143-166
        V3 a = f.data3, b = f.data - a, c = f.data2 - a;
        V3 n = crossr(b, c);
        ok = plane_ray(n, to - a, td, h);
        if (ok) {
            V3 p = to - a + h.t * td;
            if (dot(crossr(b, p), n) < 0) ok = false;
            else if (dot(crossr(p, c), n) < 0) ok = false;
            else if (dot(crossr(c - b, p - b), n) < 0) ok = false;
        }
    }
    if (!ok) return false;
    h.norma = normalize(qtransform(qconj(f.rotation), h.norma));
    return true;
}

static void extend(Box &b, V3 p) { // primitives.cpp:204-211
    b.mx.x = smax(b.mx.x, p.x); b.mx.y = smax(b.mx.y, p.y); b.mx.z = smax(b.mx.z, p.z);
    b.mn.x = smin(b.mn.x, p.x); b.mn.y = smin(b.mn.y, p.y); b.mn.z = smin(b.mn.z, p.z);
}
static void extend(Box &b, const Box &o) { extend(b, o.mn); extend(b, o.mx); }
static float surf(const Box &b) { V3 d = b.mx - b.mn; return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }
// AABB::AABB(const Figure&), primitives.cpp:171-201: local extent, its 8 corners rotated back, then translated
static Box box_of(const Fig &f) {
    Box u;
    if (f.type == RT_PRIM_BOX || f.type == RT_PRIM_ELLIPSOID) { u.mn = (float)(-1.) * f.data; u.mx = f.data; }
    else {
        u.mn = {smin(f.data3.x, smin(f.data.x, f.data2.x)), smin(f.data3.y, smin(f.data.y, f.data2.y)), smin(f.data3.z, smin(f.data.z, f.data2.z))};
        u.mx = {smax(f.data3.x, smax(f.data.x, f.data2.x)), smax(f.data3.y, smax(f.data.y, f.data2.y)), smax(f.data3.z, smax(f.data.z, f.data2.z))};
    }
    Quat r = qconj(f.rotation);
    Box b;
    b.mn = b.mx = qtransform(r, u.mn);
    extend(b, qtransform(r, V3{u.mn.x, u.mn.y, u.mx.z}));
    extend(b, qtransform(r, V3{u.mn.x, u.mx.y, u.mn.z}));
    extend(b, qtransform(r, V3{u.mn.x, u.mx.y, u.mx.z}));
    extend(b, qtransform(r, V3{u.mx.x, u.mn.y, u.mn.z}));
    extend(b, qtransform(r, V3{u.mx.x, u.mn.y, u.mx.z}));
    extend(b, qtransform(r, V3{u.mx.x, u.mx.y, u.mn.z}));
    extend(b, qtransform(r, V3{u.mx.x, u.mx.y, u.mx.z}));
    b.mn = b.mn + f.position;
    b.mx = b.mx + f.position;
    return b;
}
// AABB::intersect, primitives.cpp:220-222
static bool aabb_ray(const Box &b, V3 o, V3 d, float &t, bool &inside) {
    Hit h;
    if (!box_ray((float)0.5 * (b.mx - b.mn), o - (float)0.5 * (b.mn + b.mx), d, h, false)) return false;
    t = h.t; inside = h.inside;
    return true;
}

struct Node { Box aabb; uint32_t left = 0, right = 0, first = 0, last = 0; };
struct Bvh { // bvh.h:18-141; identical to the hw8 builder except for the sort key (Figure::position) and the figure boxes
    std::vector<Node> nodes;
    uint32_t depth = 0;
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
        Box aabb{};
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
    void init(std::vector<Fig> &figs, uint32_t n) { nodes.clear(); depth = 0; build(figs, 0, n); }
    // bvh.h:111-140
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

// ---- distributions.h -------------------------------------------------------------------------------------------
static V3 cosine_sample(N01 &n01, rng_t &rng, V3 n) { // :43-53
    float a = n01(rng), b = n01(rng), c = n01(rng);
    V3 d = normalize(V3{a, b, c});
    d = d + n;
    float l = len(d);
    if (l <= 1e-9f || dot(d, n) <= 1e-9f || std::isnan(l)) return n;
    return (float)(1. / (double)l) * d;
}
static float cosine_pdf(V3 n, V3 d) { return smax(0.f, dot(d, n) / PI); }

struct Light {
    Fig fig;
    float sTotal = 0, sx = 0, sy = 0, sz = 0, wx = 0, wy = 0, wz = 0; // BoxLight :74-82
    float pointProb = 0;                                             // TriangleLight :121-127
    explicit Light(const Fig &f) : fig(f) {
        if (f.type == RT_PRIM_BOX) {
            sx = f.data.x; sy = f.data.y; sz = f.data.z;
            sTotal = 8 * (sy * sz + sx * sz + sx * sy);
            wx = sy * sz; wy = sx * sz; wz = sx * sy;
        } else if (f.type == RT_PRIM_TRIANGLE) {
            V3 a = f.data3, b = f.data - a, c = f.data2 - a;
            V3 n = crossr(b, c);
            pointProb = 1.0 / (0.5 * (double)len(n));
        }
    }
    float pdf_one(V3 x, V3 d, V3 y, V3 yn) const {
        if (fig.type == RT_PRIM_BOX) return (double)len2(x - y) / ((double)sTotal * std::fabs((double)dot(d, yn)));  // :69-71
        if (fig.type == RT_PRIM_TRIANGLE) return (double)(pointProb * len2(x - y)) / std::fabs((double)dot(d, yn)); // :117-119
        V3 r = fig.data;                                                                                             // :150-155
        V3 n = qtransform(fig.rotation, y - fig.position) / r;
        float pp = 1. / (double)(4 * PI * len(V3{n.x * r.y * r.z, r.x * n.y * r.z, r.x * r.y * n.z}));
        return (double)(pp * len2(x - y)) / std::fabs((double)dot(d, yn));
    }
    V3 sample(U01 &u01, N01 &n01, rng_t &rng, V3 x) const {
        if (fig.type == RT_PRIM_TRIANGLE) {                                  // :129-142
            V3 a = fig.data3, b = fig.data - a, c = fig.data2 - a;
            float u = u01(rng);
            float v = u01(rng);
            if ((double)(u + v) > 1.) { u = 1 - u; v = 1 - v; }
            V3 point = fig.position + qtransform(qconj(fig.rotation), a + u * b + v * c);
            return normalize(point - x);
        }
        for (;;) {
            V3 point;
            if (fig.type == RT_PRIM_BOX) {                                   // :84-105; Vec3(a,b,c): arguments evaluated right to left by g++
                float u = u01(rng) * (wx + wy + wz);
                float flipSign = (double)u01(rng) > 0.5 ? 1 : -1;
                if (u < wx) { float c = (2 * u01(rng) - 1) * sz; float b = (2 * u01(rng) - 1) * sy; point = V3{flipSign * sx, b, c}; }
                else if (u < wx + wy) { float c = (2 * u01(rng) - 1) * sz; float a = (2 * u01(rng) - 1) * sx; point = V3{a, flipSign * sy, c}; }
                else { float b = (2 * u01(rng) - 1) * sy; float a = (2 * u01(rng) - 1) * sx; point = V3{a, b, flipSign * sz}; }
            } else {                                                         // :160-171
                float a = n01(rng), b = n01(rng), c = n01(rng);
                point = fig.data * normalize(V3{a, b, c});
            }
            V3 actual = qtransform(qconj(fig.rotation), point) + fig.position;
            Hit h;
            if (fig_ray(fig, x, normalize(actual - x), h)) return normalize(actual - x);
        }
    }
};

struct Scene5 {
    std::vector<Fig> figs;      // after initBVH: [0, nonPlanes) in BVH order, planes after
    uint32_t nonPlanes = 0;
    Bvh bvh;
    std::vector<Light> lights;  // FiguresMix::figures_
    Bvh light_bvh;
    V3 camPos, camRight, camUp, camFwd, bg;
    float fovX = 0;
    int width = 0, height = 0, samples = 1, rayDepth = 1;

    void init() {
        // Scene::initBVH, scene.cpp:18-23
        nonPlanes = (uint32_t)(std::partition(figs.begin(), figs.end(), [](const Fig &e) { return e.type != RT_PRIM_PLANE; }) - figs.begin());
        bvh.init(figs, nonPlanes);
        // FiguresMix::FiguresMix on a COPY of the reordered figures, distributions.h:180-198
        std::vector<Fig> copy = figs;
        size_t n = std::partition(copy.begin(), copy.end(), [](const Fig &f) {
            if (f.emission.x == 0 && f.emission.y == 0 && f.emission.z == 0) return false;
            return f.type == RT_PRIM_BOX || f.type == RT_PRIM_ELLIPSOID || f.type == RT_PRIM_TRIANGLE;
        }) - copy.begin();
        light_bvh.init(copy, (uint32_t)n);
        for (size_t i = 0; i < n; i++) lights.push_back(Light(copy[i]));
    }
    // Scene::intersect, scene.cpp:25-45
    bool intersect(V3 o, V3 d, Hit &best, int &pos) const {
        bool have = false;
        for (int i = (int)nonPlanes; i < (int)figs.size(); i++) {
            Hit h;
            if (fig_ray(figs[i], o, d, h) && (!have || h.t < best.t)) { best = h; pos = i; have = true; }
        }
        Hit bh; int bi = -1;
        if (bvh.intersect(figs, 0, o, d, have, have ? best.t : 0.f, bh, bi) && (!have || bh.t < best.t)) { best = bh; pos = bi; have = true; }
        return have;
    }
    // FiguresMix::pdfOneFigureLight, distributions.h:219-254
    float light_pdf_one(const Light &L, V3 x, V3 d) const {
        Hit h1;
        if (!fig_ray(L.fig, x, d, h1)) return 0.;
        if (std::isnan(h1.t)) return INFINITY;
        V3 y = x + h1.t * d;
        float ans = L.pdf_one(x, d, y, h1.norma);
        if (L.fig.type == RT_PRIM_TRIANGLE) return ans;
        Hit h2;
        if (!fig_ray(L.fig, x + (float)((long double)h1.t + eps) * d, d, h2)) return ans;
        V3 y2 = x + (float)((long double)h1.t + eps + (long double)h2.t) * d;
        return ans + L.pdf_one(x, d, y2, h2.norma);
    }
    // FiguresMix::getTotalPdf, distributions.h:256-274
    float total_pdf(uint32_t pos, V3 x, V3 d) const {
        const Node &cur = light_bvh.nodes[pos];
        float t; bool inside;
        if (!aabb_ray(cur.aabb, x, d, t, inside)) return 0;
        if (cur.left == 0) {
            float result = 0;
            for (uint32_t i = cur.first; i < cur.last; i++) result += light_pdf_one(lights[i], x, d);
            return result;
        }
        float l = total_pdf(cur.left, x, d);
        float r = total_pdf(cur.right, x, d);
        return l + r;
    }
    V3 mix_sample(U01 &u01, N01 &n01, rng_t &rng, V3 x, V3 n) const { // Mix::sample :283-290, FiguresMix::sample :200-209
        size_t comps = lights.empty() ? 1 : 2;
        int distNum = u01(rng) * comps;
        if (distNum == 0) return cosine_sample(n01, rng, n);
        int li = u01(rng) * lights.size();
        return lights[li].sample(u01, n01, rng, x);
    }
    float mix_pdf(V3 x, V3 n, V3 d) const { // Mix::pdf :292-302, FiguresMix::pdf :211-213
        float ans = 0;
        ans += cosine_pdf(n, d);
        if (lights.empty()) return ans / (size_t)1;
        ans += total_pdf(0, x, d) / lights.size();
        return ans / (size_t)2;
    }
    // Scene::getColor, scene.cpp:47-103
    V3 get_color(U01 &u01, N01 &n01, rng_t &rng, V3 ro, V3 rd, int recLimit) const {
        if (recLimit == 0) return V3{0., 0., 0.};
        Hit h; int pos = -1;
        if (!intersect(ro, rd, h, pos)) return bg;
        const Fig &f = figs[pos];
        float t = h.t; V3 norma = h.norma;
        V3 x = ro + t * rd;
        const float epsf = (float)eps;
        if (f.kind == RT_MAT_DIFFUSE) {
            V3 d = mix_sample(u01, n01, rng, x + epsf * norma, norma);
            if (dot(d, norma) < 0) return f.emission;
            float pdf = mix_pdf(x + epsf * norma, norma, d);
            V3 inner = get_color(u01, n01, rng, x + epsf * d, d, recLimit - 1);
            return f.emission + (float)(1. / (double)(PI * pdf) * (double)dot(d, norma)) * f.color * inner;
        }
        V3 dn = normalize(rd);
        V3 refl = dn - (float)(2. * dot(norma, dn)) * norma;
        V3 o = ro + t * rd + epsf * refl;
        if (f.kind == RT_MAT_METALLIC) return f.emission + f.color * get_color(u01, n01, rng, o, refl, recLimit - 1);
        V3 reflected = get_color(u01, n01, rng, o, refl, recLimit - 1);
        float eta1 = 1., eta2 = f.ior;
        if (h.inside) std::swap(eta1, eta2);
        V3 l = neg1(normalize(rd));
        float sinTheta2 = eta1 / eta2 * std::sqrt((double)(1 - dot(norma, l) * dot(norma, l)));
        if (std::fabs((double)sinTheta2) > 1.) return f.emission + reflected;
        float r0 = std::pow((double)((eta1 - eta2) / (eta1 + eta2)), 2.);
        float r = r0 + (1 - r0) * std::pow((double)(1 - dot(norma, l)), 5.);
        if (u01(rng) < r) return f.emission + reflected;
        float cosTheta2 = std::sqrt((double)(1 - sinTheta2 * sinTheta2));
        V3 refr = (eta1 / eta2) * neg1(l) + (eta1 / eta2 * dot(norma, l) - cosTheta2) * norma;
        V3 refracted = get_color(u01, n01, rng, ro + t * rd + epsf * refr, refr, recLimit - 1);
        if (!h.inside) refracted = refracted * f.color;
        return f.emission + refracted;
    }
    // scene.cpp:105-126
    V3 get_pixel(rng_t &rng, int x, int y) const {
        U01 u01(0.0, 1.0); N01 n01(0.0, 1.0);
        V3 color{0, 0, 0};
        float tanFovX = std::tan((double)(fovX / 2));
        float tanFovY = tanFovX * height / width;
        for (int s = 0; s < samples; s++) {
            float fx = x + u01(rng);
            float fy = y + u01(rng);
            float nx = tanFovX * (2 * fx / width - 1);
            float ny = tanFovY * (2 * fy / height - 1);
            color = color + get_color(u01, n01, rng, camPos, nx * camRight - ny * camUp + camFwd, rayDepth);
        }
        return (float)(1.0 / samples) * color;
    }
};
} // namespace rto5

using namespace rto5;
extern "C" {
void *rto_hw5_create(const rt_scene_desc *d) {
    Scene5 *s = new Scene5();
    for (uint32_t i = 0; i < d->n_primitives; i++) {
        const rt_primitive &p = d->primitives[i];
        Fig f;
        f.type = p.type; f.data = v3(p.data); f.data2 = v3(p.data2); f.data3 = v3(p.data3); f.position = v3(p.position);
        f.rotation = Quat{v3(p.rotation), p.rotation[3]};
        f.color = v3(p.color); f.emission = v3(p.emission); f.kind = p.kind; f.ior = p.ior; f.load_index = i;
        s->figs.push_back(f);
    }
    s->camPos = v3(d->camera.position); s->camRight = v3(d->camera.right); s->camUp = v3(d->camera.up); s->camFwd = v3(d->camera.forward);
    s->fovX = d->camera.fov_x; s->bg = v3(d->bg_color);
    s->init();
    return s;
}
void rto_hw5_destroy(void *p) { delete (Scene5 *)p; }
uint32_t rto_hw5_num_lights(void *p) { return (uint32_t)((Scene5 *)p)->lights.size(); }
// figure order after initBVH and light order (indices into the LOAD-order primitive array)
void rto_hw5_orders(void *p, uint32_t *figure_order, uint32_t *light_order) {
    Scene5 *s = (Scene5 *)p;
    for (size_t i = 0; i < s->figs.size(); i++) figure_order[i] = s->figs[i].load_index;
    for (size_t i = 0; i < s->lights.size(); i++) light_order[i] = s->lights[i].fig.load_index;
}
int rto_hw5_render(void *p, int width, int height, int samples, int ray_depth, int x0, int y0, int w, int h, float *out_rgb, uint8_t *out8, int nthreads) {
    Scene5 *s = (Scene5 *)p;
    s->width = width; s->height = height; s->samples = samples; s->rayDepth = ray_depth;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
    for (int j = 0; j < w * h; j++) {
        int x = x0 + j % w, y = y0 + j / w;
        rng_t rng(y * width + x);                                  // hw5/src/sceneio.cpp:110
        V3 px = s->get_pixel(rng, x, y);
        if (out_rgb) { out_rgb[3 * j] = px.x; out_rgb[3 * j + 1] = px.y; out_rgb[3 * j + 2] = px.z; }
        if (out8) to_extern(gamma_corrected(aces_tonemap(px)), out8 + 3 * j);
    }
    return 0;
}
}
