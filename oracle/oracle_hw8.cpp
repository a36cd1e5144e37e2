// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
//
// CPU restatement of the hw8 render path of volivan239/raytracing-course-hw in exact-replay
// semantics: per-pixel std::minstd_rand(y*W+x), libstdc++ uniform_real/normal distributions,
// the reference's float/double expression mix and evaluation order.  Every function cites the
// reference lines it follows.  Pinned against the compiled reference by tests/test_oracle_ref.py
// (oracle/_ref harnesses, fixtures in tests/golden/); see DESIGN.md "Oracle pinning" for what is
// pinned by which fixture and the two glue functions that are "parity unpinned".
#include "oracle_common.h"
#include <optional>
#include <utility>
#include <atomic>
#include <omp.h>
#include <cstdio>

namespace rto {

static const float T_MAX = 1e4;               // primitives.cpp:12
static const long double eps_ld = 1e-4;       // primitives.h:9
static const size_t CLAMP_HACK = 6;           // scene.cpp:7

struct Vertex {                                // primitives.h:25-34
    V3 coords;
    float tu = 0, tv = 0;
    V3 normals;
    V3 tan;
    float tanw = 0;
};
struct Fig {                                   // primitives.h:57-74 (triangle only)
    Vertex d, d2, d3;
    uint32_t mat = 0;
    uint32_t orig = 0; // index in LOAD order (bookkeeping, not in the reference)
};
struct Hit {                                   // primitives.h:48-55
    float t;
    V3 ng;
    float tu, tv;
    V3 ns;
    V3 tan;
    float tanw;
    bool inside;
};
struct Box { V3 mn, mx; };

struct Counters { uint64_t closest = 0, lightq = 0, boxes = 0, tris = 0; };
static thread_local Counters tl_cnt;
static thread_local std::vector<float> *tl_trace = nullptr; // diagnostic query log of one pixel (rto_hw8_trace_pixel)
static bool g_debug_bruteforce = false;

// ---- primitives.cpp ---------------------------------------------------------------------------
// primitives.cpp:18-27
static inline bool plane_ray(V3 n, V3 o, V3 d, float &t, V3 &ng, bool &inside) {
    t = -dot(o, n) / dot(d, n);
    if (t > 0 && t < T_MAX) {
        if (dot(d, n) > 0) { ng = neg1(n); inside = true; }
        else { ng = n; inside = false; }
        return true;
    }
    return false;
}
// primitives.cpp:29-53 with require_norma == false (the only form on the hw8 hot path)
static inline bool box_ray(V3 s, V3 o, V3 d, float &t, bool &inside) {
    V3 ts1 = (neg1(s) - o) / d;
    V3 ts2 = (s - o) / d;
    float t1x = smin(ts1.x, ts2.x), t2x = smax(ts1.x, ts2.x);
    float t1y = smin(ts1.y, ts2.y), t2y = smax(ts1.y, ts2.y);
    float t1z = smin(ts1.z, ts2.z), t2z = smax(ts1.z, ts2.z);
    float t1 = smax(smax(t1x, t1y), t1z);
    float t2 = smin(smin(t2x, t2y), t2z);
    if (t1 > t2 || t2 < 0) return false;
    if (t1 < 0) { inside = true; t = t2; }
    else { inside = false; t = t1; }
    return true;
}
// primitives.cpp:163-165
static inline bool aabb_ray(const Box &b, V3 o, V3 d, float &t, bool &inside) {
    tl_cnt.boxes++;
    return box_ray(0.5f * (b.mx - b.mn), o - 0.5f * (b.mn + b.mx), d, t, inside);
}
// primitives.cpp:76-80
static inline void solve2(float a1, float b1, float c1, float a2, float b2, float c2, float &x, float &y) {
    y = (c1 * a2 - c2 * a1) / (b1 * a2 - a1 * b2);
    x = a2 == 0 ? (c1 - b1 * y) / a1 : (c2 - b2 * y) / a2;
}
static const float magic1[] = {0.239, 0.419, 0.533};        // primitives.cpp:82
static const float magic2[] = {0.35743, 0.66682, 0.69695};  // primitives.cpp:83

// primitives.cpp:85-125
static bool tri_ray(const Fig &f, V3 o, V3 d, Hit &h) {
    tl_cnt.tris++;
    V3 a = f.d3.coords;
    V3 b = f.d.coords - a;
    V3 c = f.d2.coords - a;
    V3 n = crossr(b, c);
    float t; V3 ng; bool inside;
    if (!plane_ray(n, o - a, d, t, ng, inside)) return false;
    V3 p = o - a + t * d;
    float a1 = magic1[0] * b.x + magic1[1] * b.y + magic1[2] * b.z;
    float b1 = magic1[0] * c.x + magic1[1] * c.y + magic1[2] * c.z;
    float c1 = magic1[0] * p.x + magic1[1] * p.y + magic1[2] * p.z;
    float a2 = magic2[0] * b.x + magic2[1] * b.y + magic2[2] * b.z;
    float b2 = magic2[0] * c.x + magic2[1] * c.y + magic2[2] * c.z;
    float c2 = magic2[0] * p.x + magic2[1] * p.y + magic2[2] * p.z;
    float u, v;
    solve2(a1, b1, c1, a2, b2, c2, u, v);
    if (u < 0 || v < 0 || u + v > 1) return false;
    V3 sn = f.d3.normals + u * (f.d.normals - f.d3.normals) + v * (f.d2.normals - f.d3.normals);
    float tu = f.d3.tu + u * (f.d.tu - f.d3.tu) + v * (f.d2.tu - f.d3.tu);
    float tv = f.d3.tv + u * (f.d.tv - f.d3.tv) + v * (f.d2.tv - f.d3.tv);
    V3 tg = f.d3.tan + u * (f.d.tan - f.d3.tan) + v * (f.d2.tan - f.d3.tan);
    tg = normalize(tg);
    sn = normalize(sn);
    if (inside) sn = neg1(sn);
    ng = normalize(ng);
    h = Hit{t, ng, tu, tv, sn, tg, f.d.tanw, inside};
    return true;
}

// primitives.cpp:130-141
static Box box_of(const Fig &f) {
    Box b;
    b.mn = {smin(f.d3.coords.x, smin(f.d.coords.x, f.d2.coords.x)), smin(f.d3.coords.y, smin(f.d.coords.y, f.d2.coords.y)),
            smin(f.d3.coords.z, smin(f.d.coords.z, f.d2.coords.z))};
    b.mx = {smax(f.d3.coords.x, smax(f.d.coords.x, f.d2.coords.x)), smax(f.d3.coords.y, smax(f.d.coords.y, f.d2.coords.y)),
            smax(f.d3.coords.z, smax(f.d.coords.z, f.d2.coords.z))};
    return b;
}
// primitives.cpp:144-156
static void extend(Box &b, V3 p) {
    b.mx.x = smax(b.mx.x, p.x); b.mx.y = smax(b.mx.y, p.y); b.mx.z = smax(b.mx.z, p.z);
    b.mn.x = smin(b.mn.x, p.x); b.mn.y = smin(b.mn.y, p.y); b.mn.z = smin(b.mn.z, p.z);
}
static void extend(Box &b, const Box &o) { extend(b, o.mn); extend(b, o.mx); }
// primitives.cpp:158-161
static float surf(const Box &b) {
    V3 d = b.mx - b.mn;
    return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
}

// ---- bvh.h --------------------------------------------------------------------------------------
struct Node { Box aabb; uint32_t left = 0, right = 0, first = 0, last = 0; }; // bvh.h:9-16
struct Bvh {
    std::vector<Node> nodes;
    uint32_t root = 0;
    uint32_t depth = 0;

    // bvh.h:34-54
    static std::pair<float, uint32_t> best_split(std::vector<Fig> &figs, uint32_t first, uint32_t last) {
        std::vector<float> scores(last - first, 0);
        Box pre = box_of(figs[first]);
        for (size_t i = 1; i < last - first; i++) {
            scores[i] = surf(pre) * i;
            extend(pre, box_of(figs[first + i]));
        }
        Box suf = box_of(figs[last - 1]);
        for (size_t i = last - first - 1; i >= 1; i--) {
            scores[i] += surf(suf) * ((last - first) - i);
            extend(suf, box_of(figs[first + i - 1]));
        }
        std::pair<float, uint32_t> ans = {scores[1], first + 1};
        for (size_t i = 2; i < last - first; i++)
            if (scores[i] < ans.first) ans = {scores[i], (uint32_t)(first + i)};
        return ans;
    }
    // bvh.h:60-65
    static void half_split(std::vector<Fig> &figs, uint32_t first, uint32_t last, int axis) {
        if (axis == 0) std::sort(figs.begin() + first, figs.begin() + last, [](const Fig &l, const Fig &r) { return l.d3.coords.x < r.d3.coords.x; });
        else if (axis == 1) std::sort(figs.begin() + first, figs.begin() + last, [](const Fig &l, const Fig &r) { return l.d3.coords.y < r.d3.coords.y; });
        else std::sort(figs.begin() + first, figs.begin() + last, [](const Fig &l, const Fig &r) { return l.d3.coords.z < r.d3.coords.z; });
    }
    // bvh.h:67-109
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

    // bvh.h:111-142 — recursive closest hit, left child first, strict '<' keeps the first found.
    bool intersect(const std::vector<Fig> &figs, uint32_t pos, V3 o, V3 d, bool have_best, float cur_best, Hit &out, int &idx) const {
        const Node &cur = nodes[pos];
        float t; bool inside;
        if (!aabb_ray(cur.aabb, o, d, t, inside)) return false;
        if (have_best && cur_best < t && !inside) return false;
        bool found = false;
        if (cur.left == 0) {
            for (uint32_t i = cur.first; i < cur.last; i++) {
                Hit h;
                if (tri_ray(figs[i], o, d, h) && (!found || h.t < out.t)) { out = h; idx = (int)i; found = true; }
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

// ---- material.h ---------------------------------------------------------------------------------
struct MaterialModel {
    float baseMetallic; V3 baseColor;
    // material.h:11-17
    float distributionTerm(V3 h, V3 n, float alpha2) const {
        float dotHN = dot(h, n);
        if (dotHN <= 0) return 0;
        return alpha2 / (M_PI * std::pow((double)smax(0.f, (alpha2 - 1) * dotHN * dotHN + 1), 2.0));
    }
    // material.h:19-21
    float v1(V3 n, V3 x, float alpha2) const {
        return 1. / (std::fabs((double)dot(n, x)) + std::sqrt((double)smax(0.f, alpha2 + (1 - alpha2) * dot(n, x) * dot(n, x))));
    }
    // material.h:23-29
    float specularBrdf(V3 l, V3 v, V3 n, float alpha2) const {
        V3 h = normalize(l + v);
        if (dot(h, l) < 1e-4 || dot(h, v) < 1e-4) return 0;
        return distributionTerm(h, n, alpha2) * v1(n, l, alpha2) * v1(n, v, alpha2);
    }
    // material.h:31-33
    V3 diffuseBrdf(V3 color) const { return (float)(1. / M_PI) * color; }
    // material.h:35-37
    V3 fresnelTerm(V3 f0, V3 f90, V3 v, V3 h) const {
        return f0 + (float)std::pow((double)smax(0.f, (float)(1.f - std::fabs((double)dot(v, h)))), 5.0) * (f90 - f0);
    }
    // material.h:42-65
    V3 brdf(V3 l, V3 v, V3 n, V3 color, float metallic, float alpha) const {
        V3 h = normalize(l + v);
        float specular = specularBrdf(l, v, n, alpha * alpha);
        V3 metalBrdf, dielectricBrdf;
        metallic *= baseMetallic;
        if (metallic > 0 && dot(v, n) >= 0 && dot(l, n) >= 0) {
            V3 ft = fresnelTerm(baseColor * color, V3{1, 1, 1}, v, h);
            metalBrdf = specular * ft;
        }
        if (metallic < 1) {
            V3 diffuse;
            if (dot(l, n) >= 0) diffuse = diffuseBrdf(baseColor * color);
            V3 ft = fresnelTerm(V3{0.04, 0.04, 0.04}, V3{1, 1, 1}, v, h);
            dielectricBrdf = diffuse * (V3{1, 1, 1} - ft) + specular * ft;
        }
        return (float)(1.0 - metallic) * dielectricBrdf + metallic * metalBrdf;
    }
    // hw7/src/include/material.h:44-61 — hw7 form (per-material alpha2/metallic, no texture inputs, no
    // v.n / l.n gates); used only to pin the shared integrator structure against the compiled hw7 reference.
    V3 brdf_hw7(V3 l, V3 v, V3 n, float alpha2) const {
        V3 h = normalize(l + v);
        float specular = specularBrdf(l, v, n, alpha2);
        V3 metalBrdf, dielectricBrdf;
        float metallic = baseMetallic;
        if (metallic > 0) {
            V3 ft = fresnelTerm(baseColor, V3{1, 1, 1}, v, h);
            metalBrdf = specular * ft;
        }
        if (metallic < 1) {
            V3 diffuse = diffuseBrdf(baseColor);
            V3 ft = fresnelTerm(V3{0.04, 0.04, 0.04}, V3{1, 1, 1}, v, h);
            dielectricBrdf = diffuse * (V3{1, 1, 1} - ft) + specular * ft;
        }
        return (float)(1.0 - metallic) * dielectricBrdf + metallic * metalBrdf;
    }
};

// ---- distributions.h ----------------------------------------------------------------------------
typedef std::uniform_real_distribution<float> U01;
typedef std::normal_distribution<float> N01;

// distributions.h:42-52
static V3 cosine_sample(N01 &n01, rng_t &rng, V3 n) {
    float a = n01(rng), b = n01(rng), c = n01(rng); // braced init: left to right
    V3 d = normalize(V3{a, b, c});
    d = d + n;
    float l = len(d);
    const float ceps = 1e-9;
    if (l <= ceps || dot(d, n) <= ceps || std::isnan(l)) return n;
    return (float)(1. / l) * d;
}
// distributions.h:54-57
static float cosine_pdf(V3 n, V3 d) { return smax(0.f, dot(d, n) / (float)M_PI); }

struct TriLight {                              // distributions.h:60-95
    float pointProb;
    Fig fig;
    explicit TriLight(const Fig &f) : fig(f) {
        V3 a = fig.d3.coords, b = fig.d.coords - a, c = fig.d2.coords - a;
        V3 n = crossr(b, c);
        pointProb = 1.0 / (0.5 * len(n));
    }
    float pdfOne(V3 x, V3 d, V3 y, V3 yn) const { return pointProb * len2(x - y) / std::fabs((double)dot(d, yn)); }
    V3 sample(U01 &u01, rng_t &rng, V3 x) const {
        V3 a = fig.d3.coords, b = fig.d.coords - a, c = fig.d2.coords - a;
        float u = u01(rng);
        float v = u01(rng);
        if (u + v > 1.) { u = 1 - u; v = 1 - v; }
        V3 point = a + u * b + v * c;
        return normalize(point - x);
    }
};

struct FiguresMix {                            // distributions.h:97-166
    std::vector<TriLight> lights;
    Bvh bvh;
    bool hw7_geom_normal = false;              // hw7/src/include/distributions.h:140-145 uses yn (geometric)
    float pdf_one(const TriLight &tl, V3 x, V3 d) const { // distributions.h:131-146
        Hit h;
        if (!tri_ray(tl.fig, x, d, h)) return 0.;
        if (std::isnan(h.t)) return INFINITY;
        V3 y = x + h.t * d;
        return tl.pdfOne(x, d, y, hw7_geom_normal ? h.ng : h.ns);
    }
    float total_pdf(uint32_t pos, V3 x, V3 d) const { // distributions.h:148-165
        const Node &cur = bvh.nodes[pos];
        float t; bool inside;
        if (!aabb_ray(cur.aabb, x, d, t, inside)) return 0;
        if (cur.left == 0) {
            float result = 0;
            for (uint32_t i = cur.first; i < cur.last; i++) result += pdf_one(lights[i], x, d);
            return result;
        }
        float l = total_pdf(cur.left, x, d);
        float r = total_pdf(cur.right, x, d);
        return l + r;
    }
    float pdf(V3 x, V3 d) const { // :122-124
        tl_cnt.lightq++;
        float r = total_pdf(0, x, d) / lights.size();
        if (g_debug_bruteforce) { // diagnostic: lights whose triangle test hits but whose BVH boxes the reference's slab test rejects
            float brute = 0; int nh = 0;
            for (const TriLight &tl : lights) { float p = pdf_one(tl, x, d); if (p != 0) nh++; brute += p; }
            brute /= lights.size();
            if (brute != r && !(brute != brute && r != r)) {
                fprintf(stderr, "[oracle] light pdf via BVH %.9g, brute force over all lights %.9g (%d hit)\n", r, brute, nh);
                for (size_t li = 0; li < lights.size(); li++) {
                    if (pdf_one(lights[li], x, d) == 0) continue;
                    uint32_t pos = 0; // walk root -> leaf of light li, report the first rejecting box
                    for (;;) {
                        const Node &n = bvh.nodes[pos];
                        float t; bool inside;
                        bool ok = aabb_ray(n.aabb, x, d, t, inside);
                        if (!ok) {
                            V3 s = 0.5f * (n.aabb.mx - n.aabb.mn), o = x - 0.5f * (n.aabb.mn + n.aabb.mx);
                            V3 ts1 = (neg1(s) - o) / d, ts2 = (s - o) / d;
                            fprintf(stderr, "   light %zu pdf %.6g rejected at node %u [%u,%u) box (%.9g %.9g %.9g)-(%.9g %.9g %.9g)\n      o' (%.9g %.9g %.9g) d (%.9g %.9g %.9g) ts1 (%.9g %.9g %.9g) ts2 (%.9g %.9g %.9g)\n",
                                    li, pdf_one(lights[li], x, d), pos, n.first, n.last, n.aabb.mn.x, n.aabb.mn.y, n.aabb.mn.z, n.aabb.mx.x, n.aabb.mx.y, n.aabb.mx.z,
                                    o.x, o.y, o.z, d.x, d.y, d.z, ts1.x, ts1.y, ts1.z, ts2.x, ts2.y, ts2.z);
                            break;
                        }
                        if (n.left == 0) break;
                        pos = (li < bvh.nodes[n.right].first) ? n.left : n.right;
                    }
                }
            }
        }
        return r;
    }
    V3 sample(U01 &u01, rng_t &rng, V3 x) const {                                             // :117-120
        int k = u01(rng) * lights.size();
        return lights[k].sample(u01, rng, x);
    }
};

// distributions.h:168-246
struct Vndf {
    static V3 sample_(U01 &u01, rng_t &rng, V3 v, float alpha) {
        V3 vh = normalize(V3{alpha * v.x, alpha * v.y, v.z});
        float lensq = vh.x * vh.x + vh.y * vh.y;
        V3 T1 = lensq > 0 ? (float)(1. / std::sqrt((double)lensq)) * V3{-vh.y, vh.x, 0} : V3{1, 0, 0};
        V3 T2 = crossr(T1, vh);
        float u1 = u01(rng), u2 = u01(rng);
        float r = std::sqrt((double)u1);
        float phi = 2.0 * M_PI * u2;
        float t1 = r * std::cos((double)phi);
        float t2 = r * std::sin((double)phi);
        float s = 0.5 * (1.0 + vh.z);
        t2 = (1.0 - s) * std::sqrt((double)(1.f - t1 * t1)) + s * t2;
        V3 nh = t1 * T1 + t2 * T2 + (float)std::sqrt((double)std::max<float>(0.f, 1.0 - t1 * t1 - t2 * t2)) * vh;
        V3 ne = normalize(V3{alpha * nh.x, alpha * nh.y, std::max<float>(0.0, nh.z)});
        return (2 * dot(ne, v)) * ne - v;
    }
    static float D(V3 n, float a) {
        return 1. / (M_PI * a * a * std::pow((double)(n.x * n.x / (a * a) + n.y * n.y / (a * a) + n.z * n.z), 2.0));
    }
    static float G1(V3 v, float a) {
        float lambda = 0.5 * (-1 + std::sqrt((double)(1 + (a * a * v.x * v.x + a * a * v.y * v.y) / (v.z * v.z))));
        return 1. / (1 + lambda);
    }
    static float pdf_(V3 d, V3 v, float a) {
        V3 ni = normalize(v + d);
        float dv = G1(v, a) * smax(0.f, dot(v, ni)) * D(ni, a) / std::fabs((double)v.z);
        float res = dv / (4 * dot(v, ni));
        return res;
    }
    static Quat getQ(V3 n) {
        V3 newN = {0, 0, 1};
        if (dot(n, newN) > 0.9999) return Quat{};
        if (dot(n, newN) < -0.9999) return Quat{V3{0, 0, 0}, -1};
        V3 a = crossr(n, newN);
        float w = std::sqrt((double)len2(n)) + dot(n, newN); // double sum, narrowed once
        float l = std::sqrt((double)(len2(a) + w * w));
        return Quat{(float)(1. / l) * a, w / l};
    }
    static V3 sample(U01 &u01, rng_t &rng, V3 n, V3 v, float alpha) {
        v = neg1(v);
        Quat q = getQ(n);
        V3 vT = qtransform(q, v);
        V3 dT = sample_(u01, rng, vT, alpha);
        return qtransform(qconj(q), dT);
    }
    static float pdf(V3 n, V3 d, V3 v, float alpha) {
        v = neg1(v);
        Quat q = getQ(n);
        return pdf_(qtransform(q, d), qtransform(q, v), alpha);
    }
};

// ---- scene.cpp ----------------------------------------------------------------------------------
struct Tex { int w = 0, h = 0; const uint8_t *data = nullptr; };

// scene.cpp:9-16
static V3 load_texel(int ix, int iy, const Tex &tex, bool srgb) {
    size_t off = 3 * (ix + tex.w * iy);
    // The reference reads out of bounds when a wrapped coordinate rounds up to exactly 1.0
    // (ix == width on the last row); that is UB there, defined here as texel value 0.
    if (off + 2 >= (size_t)tex.w * tex.h * 3) return V3{0, 0, 0};
    V3 res = (float)(1. / 255) * V3{1.f * tex.data[off], 1.f * tex.data[off + 1], 1.f * tex.data[off + 2]};
    if (srgb) return V3{std::pow(res.x, 2.2f), std::pow(res.y, 2.2f), std::pow(res.z, 2.2f)};
    return res;
}
// scene.cpp:18-35
static V3 sample_texture(float tx, float ty, const Tex &tex, bool srgb) {
    tx -= std::floor(tx);
    ty -= std::floor(ty);
    tx *= tex.w;
    ty *= tex.h;
    int ix1 = std::floor(tx), ix2 = (ix1 + 1) % tex.w;
    int iy1 = std::floor(ty), iy2 = (iy1 + 1) % tex.h;
    float dx = tx - ix1;
    float dy = ty - iy1;
    V3 p11 = load_texel(ix1, iy1, tex, srgb);
    V3 p12 = load_texel(ix1, iy2, tex, srgb);
    V3 p21 = load_texel(ix2, iy1, tex, srgb);
    V3 p22 = load_texel(ix2, iy2, tex, srgb);
    return (1 - dx) * ((1 - dy) * p11 + dy * p12) + dx * ((1 - dy) * p21 + dy * p22);
}
// scene.cpp:37-53
static V3 apply_normal_map(V3 sn, V3 tan, float tanw, V3 sample) {
    V3 lx = tan, lz = sn;
    V3 ly = tanw * crossr(lx, lz);
    V3 ln = 2.f * sample - V3{1., 1., 1.};
    V3 n = ln.x * lx + ln.y * ly + ln.z * lz;
    return normalize(n);
}

struct Scene {
    std::vector<Fig> figs;                 // BVH order after init (scene.cpp:76-78)
    std::vector<rt_material> mats;
    std::vector<MaterialModel> models;     // sceneio.cpp:239-244
    std::vector<uint32_t> tex_source;
    std::vector<Tex> images;
    std::vector<std::vector<uint8_t>> image_store;
    bool has_env = false; Tex env;
    V3 camPos, camRight, camUp, camFwd; float fovY = 0;
    V3 bg;
    Bvh bvh;
    FiguresMix lightmix;
    int n_components = 2;
    int width = 0, height = 0, samples = 1, rayDepth = 6;
    bool hw7 = false; // replay hw7/src/scene.cpp:29-61 instead of hw8/src/scene.cpp:84-165

    bool emissive(const Fig &f) const {     // distributions.h:104-109 (factor, not texture)
        const rt_material &m = mats[f.mat];
        return !(m.emission[0] == 0 && m.emission[1] == 0 && m.emission[2] == 0);
    }
    void init() {
        bvh.init(figs, (uint32_t)figs.size());                       // scene.cpp:76-78
        std::vector<Fig> copy = figs;                                // distributions.h:103 (by value)
        size_t n = std::partition(copy.begin(), copy.end(), [this](const Fig &f) { return emissive(f); }) - copy.begin();
        lightmix.bvh.init(copy, (uint32_t)n);                        // :111
        lightmix.lights.clear();
        for (size_t i = 0; i < n; i++) lightmix.lights.push_back(TriLight(copy[i])); // :112-114
        n_components = lightmix.lights.empty() ? 2 : 3;              // scene.cpp:65-74
    }
    const Tex &tex(int32_t t) const { return images[tex_source[t]]; }

    // distributions.h:256-265
    V3 mix_sample(U01 &u01, N01 &n01, rng_t &rng, V3 x, V3 n, V3 v, float alpha) const {
        int k = u01(rng) * (size_t)n_components;
        if (k == 0) return cosine_sample(n01, rng, n);
        if (k == 2) return lightmix.sample(u01, rng, x);
        return Vndf::sample(u01, rng, n, v, alpha);
    }
    // distributions.h:267-279
    float mix_pdf(V3 x, V3 n, V3 d, V3 v, float alpha) const {
        float ans = 0;
        ans += cosine_pdf(n, d);
        ans += Vndf::pdf(n, d, v, alpha);
        if (n_components == 3) ans += lightmix.pdf(x, d);
        return ans / (size_t)n_components;
    }
    // scene.cpp:179-186
    void camera_ray(float x, float y, V3 &o, V3 &d) const {
        float tanFovY = std::tan((double)(fovY / 2));
        float tanFovX = tanFovY * width / height;
        float nx = tanFovX * (2 * x / width - 1);
        float ny = tanFovY * (2 * y / height - 1);
        o = camPos;
        d = normalize(nx * camRight - ny * camUp + camFwd);
    }
    // scene.cpp:84-165
    V3 get_color(U01 &u01, N01 &n01, rng_t &rng, V3 ro, V3 rd, int recLimit) const {
        if (recLimit == 0) return V3{0., 0., 0.};
        Hit h; int idx = -1;
        tl_cnt.closest++;
        const bool found = bvh.intersect(figs, bvh.root, ro, rd, false, 0.f, h, idx);
        if (tl_trace) { // diagnostic (rto_hw8_trace_pixel): every closest-hit query of the traced pixel
            const float rec[12] = {ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, found ? h.t : -1.f, found ? (float)idx : -1.f, found && h.inside ? 1.f : 0.f, found ? h.tu : 0.f, found ? h.tv : 0.f, (float)recLimit};
            tl_trace->insert(tl_trace->end(), rec, rec + 12);
        }
        if (!found) {
            if (!has_env) return bg;
            float tx = 0.5 + 0.5 * std::atan2((double)rd.z, (double)rd.x) / M_PI;
            float ty = 0.5 - std::asin((double)rd.y) / M_PI;
            return sample_texture(tx, ty, env, true);
        }
        const Fig &f = figs[idx];
        const rt_material &m = mats[f.mat];
        V3 sn = h.ns;
        V3 x = ro + h.t * rd;
        const MaterialModel &model = models[f.mat];
        V3 color{1, 1, 1};
        if (m.base_color_texture >= 0) color = sample_texture(h.tu, h.tv, tex(m.base_color_texture), true);
        V3 emission{m.emission[0], m.emission[1], m.emission[2]};
        if (m.emissive_texture >= 0) emission = emission * sample_texture(h.tu, h.tv, tex(m.emissive_texture), true);
        V3 mr{1, 1, 1};
        if (m.metallic_roughness_texture >= 0) mr = sample_texture(h.tu, h.tv, tex(m.metallic_roughness_texture), false);
        V3 nsample{0.5, 0.5, 1};
        if (m.normal_texture >= 0) nsample = sample_texture(h.tu, h.tv, tex(m.normal_texture), false);
        if (!hw7) sn = apply_normal_map(sn, h.tan, h.tanw, nsample);
        float alpha = hw7 ? (float)std::pow((double)m.roughness_factor, 2.0)   // hw7/src/scene.cpp:44
                          : (float)std::pow((double)smax(0.08f, m.roughness_factor * mr.y), 2.0);
        float metallic = mr.z;
        V3 xo = x + (float)eps_ld * h.ng;
        V3 d = mix_sample(u01, n01, rng, xo, sn, rd, alpha);
        float r2 = m.roughness_factor * m.roughness_factor; // hw7/src/sceneio.cpp:187, material.h:42
        V3 brdf = hw7 ? model.brdf_hw7(d, neg1(rd), sn, r2 * r2) : model.brdf(d, neg1(rd), sn, color, metallic, alpha);
        if (brdf.x < eps_ld && brdf.y < eps_ld && brdf.z < eps_ld) return emission;
        float pdf = mix_pdf(xo, sn, d, rd, alpha);
        float k = 1. / pdf * std::fabs((double)dot(d, sn));
        V3 mult = k * brdf;
        if (mult.x > CLAMP_HACK || mult.y > CLAMP_HACK || mult.z > CLAMP_HACK || std::isnan(mult.x) || std::isnan(mult.y) || std::isnan(mult.z))
            return emission;
        return emission + mult * get_color(u01, n01, rng, xo, d, recLimit - 1);
    }
    // scene.cpp:167-177
    V3 get_pixel(rng_t &rng, int x, int y) const {
        U01 u01(0.0, 1.0);
        N01 n01(0.0, 1.0);
        V3 color{0, 0, 0};
        for (int s = 0; s < samples; s++) {
            float nx = x + u01(rng);
            float ny = y + u01(rng);
            V3 o, d;
            camera_ray(nx, ny, o, d);
            color = color + get_color(u01, n01, rng, o, d, rayDepth);
        }
        return (float)(1.0 / samples) * color;
    }
};

static V3 v3(const float *p) { return {p[0], p[1], p[2]}; }

} // namespace rto

using namespace rto;

extern "C" {

struct rto_counters { uint64_t closest, lightq, boxes, tris; };

static void *create_common(const rt_scene_desc *d, bool hw7) {
    Scene *s = new Scene();
    s->hw7 = hw7; s->lightmix.hw7_geom_normal = hw7;
    s->figs.resize(d->n_triangles);
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        Fig &f = s->figs[i];
        Vertex *vs[3] = {&f.d, &f.d2, &f.d3};
        for (int k = 0; k < 3; k++) {
            vs[k]->coords = v3(d->positions + 9 * i + 3 * k);
            if (d->texcoords) { vs[k]->tu = d->texcoords[6 * i + 2 * k]; vs[k]->tv = d->texcoords[6 * i + 2 * k + 1]; }
            if (d->normals) vs[k]->normals = v3(d->normals + 9 * i + 3 * k);
            if (d->tangents) { vs[k]->tan = v3(d->tangents + 12 * i + 4 * k); vs[k]->tanw = d->tangents[12 * i + 4 * k + 3]; }
        }
        f.mat = d->material_index[i];
        f.orig = i;
    }
    s->mats.assign(d->materials, d->materials + d->n_materials);
    for (auto &m : s->mats) s->models.push_back(MaterialModel{m.metallic_factor, v3(m.base_color)});
    s->tex_source.assign(d->texture_source, d->texture_source + d->n_textures);
    for (uint32_t i = 0; i < d->n_images; i++) {
        const rt_image &im = d->images[i];
        s->image_store.emplace_back(im.rgb, im.rgb + (size_t)im.width * im.height * 3);
    }
    for (uint32_t i = 0; i < d->n_images; i++) s->images.push_back(Tex{d->images[i].width, d->images[i].height, s->image_store[i].data()});
    if (d->environment_map) {
        const rt_image &im = *d->environment_map;
        s->image_store.emplace_back(im.rgb, im.rgb + (size_t)im.width * im.height * 3);
        s->env = Tex{im.width, im.height, s->image_store.back().data()};
        s->has_env = true;
    }
    s->camPos = v3(d->camera.position); s->camRight = v3(d->camera.right);
    s->camUp = v3(d->camera.up); s->camFwd = v3(d->camera.forward);
    s->fovY = d->camera.fov_y;
    s->bg = v3(d->bg_color);
    s->init();
    return s;
}
void *rto_hw8_create(const rt_scene_desc *d) { return create_common(d, false); }
// hw7 replay mode of the same code (pins the integrator structure)
void *rto_hw7_create(const rt_scene_desc *d) { return create_common(d, true); }
void rto_debug_bruteforce_lights(int on) { g_debug_bruteforce = on != 0; }
void rto_hw8_destroy(void *p) { delete (Scene *)p; }

uint32_t rto_hw8_num_lights(void *p) { return (uint32_t)((Scene *)p)->lightmix.lights.size(); }
// LOAD-order index of every light in light order / of every triangle in BVH order.
void rto_hw8_light_order(void *p, uint32_t *out) {
    Scene *s = (Scene *)p;
    for (size_t i = 0; i < s->lightmix.lights.size(); i++) out[i] = s->lightmix.lights[i].fig.orig;
}
void rto_hw8_figure_order(void *p, uint32_t *out) {
    Scene *s = (Scene *)p;
    for (size_t i = 0; i < s->figs.size(); i++) out[i] = s->figs[i].orig;
}
void rto_hw8_bvh_stats(void *p, uint32_t *out4) {
    Scene *s = (Scene *)p;
    out4[0] = (uint32_t)s->bvh.nodes.size(); out4[1] = s->bvh.depth;
    out4[2] = (uint32_t)s->lightmix.bvh.nodes.size(); out4[3] = s->lightmix.bvh.depth;
}

// Test hook for the product's throughput mode (include/rtamd.h: sample_streams): stream k of a pixel is an ordinary replay of
// the reference loop with the engine seeded y*width + x + k*width*height.  0 = the reference's seeding.
static uint32_t g_seed_offset = 0;
void rto_hw8_set_seed_offset(uint32_t off) { g_seed_offset = off; }

// Diagnostic: the closest-hit queries of one pixel's replay, 12 floats each (origin, direction, t or -1, figure index or -1, inside,
// texture u, v, remaining depth); returns the number of floats written (at most cap).
int rto_hw8_trace_pixel(void *p, int width, int height, int samples, int ray_depth, int x, int y, float *out, int cap) {
    Scene *s = (Scene *)p;
    s->width = width; s->height = height; s->samples = samples; s->rayDepth = ray_depth > 0 ? ray_depth : 6;
    std::vector<float> log;
    tl_trace = &log;
    rng_t rng((uint32_t)(y * width + x) + g_seed_offset);
    (void)s->get_pixel(rng, x, y);
    tl_trace = nullptr;
    int n = (int)log.size() < cap ? (int)log.size() : cap;
    memcpy(out, log.data(), (size_t)n * sizeof(float));
    return n;
}

// Render the pixel rectangle [x0,x0+w) x [y0,y0+h) of a width x height image.
// out_rgb: w*h*3 linear float radiance (nullable); out8: w*h*3 tonemapped bytes (nullable).
// Mirrors the loop body of sceneio.cpp:387-396 (seed = y*width + x of the FULL image).
int rto_hw8_render(void *p, int width, int height, int samples, int ray_depth, int x0, int y0, int w, int h,
                   float *out_rgb, uint8_t *out8, int nthreads, rto_counters *cnt) {
    Scene *s = (Scene *)p;
    s->width = width; s->height = height; s->samples = samples; s->rayDepth = ray_depth > 0 ? ray_depth : 6;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads) reduction(+ : c0, c1, c2, c3)
    for (int j = 0; j < w * h; j++) {
        tl_cnt = Counters{};
        int x = x0 + j % w, y = y0 + j / w;
        int i = y * width + x;
        rng_t rng((uint32_t)i + g_seed_offset);
        V3 px = s->get_pixel(rng, x, y);
        if (out_rgb) { out_rgb[3 * j] = px.x; out_rgb[3 * j + 1] = px.y; out_rgb[3 * j + 2] = px.z; }
        if (out8) to_extern(gamma_corrected(aces_tonemap(px)), out8 + 3 * j);
        c0 += tl_cnt.closest; c1 += tl_cnt.lightq; c2 += tl_cnt.boxes; c3 += tl_cnt.tris;
    }
    if (cnt) { cnt->closest = c0; cnt->lightq = c1; cnt->boxes = c2; cnt->tris = c3; }
    return 0;
}

// ---- per-function entry points for the reference pins (tests/test_oracle_ref.py) ---------------
// Closest hit of ray (o,d): returns BVH-order index or -1; out = t, ng(3), uv(2), ns(3), tan(3), tanw, inside
int rto_hw8_closest_hit(void *p, const float *o, const float *d, float *out14) {
    Scene *s = (Scene *)p;
    Hit h; int idx = -1;
    if (!s->bvh.intersect(s->figs, s->bvh.root, v3(o), v3(d), false, 0.f, h, idx)) return -1;
    float r[14] = {h.t, h.ng.x, h.ng.y, h.ng.z, h.tu, h.tv, h.ns.x, h.ns.y, h.ns.z, h.tan.x, h.tan.y, h.tan.z, h.tanw, h.inside ? 1.f : 0.f};
    memcpy(out14, r, sizeof r);
    return idx;
}
float rto_hw8_light_pdf(void *p, const float *x, const float *d) { return ((Scene *)p)->lightmix.pdf(v3(x), v3(d)); }
// Mix::sample then Mix::pdf with a fresh engine seeded `seed` after `burn` discarded uniform draws.
void rto_hw8_mix_sample_pdf(void *p, uint32_t seed, const float *x, const float *n, const float *v, float alpha, float *out_d3_pdf) {
    Scene *s = (Scene *)p;
    rng_t rng(seed); U01 u01(0.0, 1.0); N01 n01(0.0, 1.0);
    V3 d = s->mix_sample(u01, n01, rng, v3(x), v3(n), v3(v), alpha);
    float pdf = s->mix_pdf(v3(x), v3(n), d, v3(v), alpha);
    out_d3_pdf[0] = d.x; out_d3_pdf[1] = d.y; out_d3_pdf[2] = d.z; out_d3_pdf[3] = pdf;
    out_d3_pdf[4] = u01(rng); // stream position check
}
void rto_hw8_brdf(float base_metallic, const float *base_color, const float *l, const float *v, const float *n, const float *color,
                  float metallic, float alpha, float *out3) {
    MaterialModel m{base_metallic, v3(base_color)};
    V3 r = m.brdf(v3(l), v3(v), v3(n), v3(color), metallic, alpha);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void rto_tonemap(const float *rgb, uint8_t *out3) { to_extern(gamma_corrected(aces_tonemap(v3(rgb))), out3); }
void rto_sample_texture(int w, int h, const uint8_t *data, float tx, float ty, int srgb, float *out3) {
    V3 r = sample_texture(tx, ty, Tex{w, h, data}, srgb != 0);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
// RNG known-answer helper: seed, then n_u uniforms followed by n_n normals.
void rto_rng_kat(uint32_t seed, int n_u, int n_n, float *out) {
    rng_t rng(seed); U01 u01(0.0, 1.0); N01 n01(0.0, 1.0);
    for (int i = 0; i < n_u; i++) out[i] = u01(rng);
    for (int i = 0; i < n_n; i++) out[n_u + i] = n01(rng);
}
// Host libm logf (what the reference's normal_distribution calls) on an array.
void rto_logf_array(const float *in, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = std::log(in[i]); }
// Same engine, normals drawn first, then uniforms.
void rto_rng_kat_normals_first(uint32_t seed, int n_n, int n_u, float *out) {
    rng_t rng(seed); U01 u01(0.0, 1.0); N01 n01(0.0, 1.0);
    for (int i = 0; i < n_n; i++) out[i] = n01(rng);
    for (int i = 0; i < n_u; i++) out[n_n + i] = u01(rng);
}

} // extern "C"
