// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// Analytic primitives of the hw3 / hw4 / hw5 snapshots (<cmath> flavour: unqualified sqrt/fabs on floats are the double
// C functions, results narrowed) shared by oracle_txt.cpp (hw3), oracle_hw4.cpp and oracle_hw5.cpp.
#pragma once
#include "oracle_common.h"

namespace rtot {
using namespace rto;

struct Prim {
    int type; V3 data, position; Quat rotation; V3 color, emission; int kind; float ior;
};
struct Hit { float t; V3 norma; bool inside; };

typedef std::uniform_real_distribution<float> U01;
typedef std::normal_distribution<float> N01;

// hw3/src/primitives.cpp:28-47 (hw1: :33-52 without the inside flag)
static inline bool smallest_root(float a, float b, float c, float &t, bool &inside) {
    float d = b * b - 4 * a * c;
    if (d <= 0) return false;
    float x1 = (-b - std::sqrt((double)d)) / (2 * a);
    float x2 = (-b + std::sqrt((double)d)) / (2 * a);
    if (x1 > x2) std::swap(x1, x2);
    if (x2 < 0) return false;
    if (x1 < 0) { t = x2; inside = true; } else { t = x1; inside = false; }
    return true;
}
// hw3/src/primitives.cpp:81-123
static inline bool box_local(V3 s, V3 o, V3 d, Hit &h) {
    V3 ts1 = (neg1(s) - o) / d, ts2 = (s - o) / d;
    float t1x = smin(ts1.x, ts2.x), t2x = smax(ts1.x, ts2.x);
    float t1y = smin(ts1.y, ts2.y), t2y = smax(ts1.y, ts2.y);
    float t1z = smin(ts1.z, ts2.z), t2z = smax(ts1.z, ts2.z);
    float t1 = smax(smax(t1x, t1y), t1z), t2 = smin(smin(t2x, t2y), t2z);
    if (t1 > t2 || t2 < 0) return false;
    float t; bool inside;
    if (t1 < 0) { inside = true; t = t2; } else { inside = false; t = t1; }
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
// hw3/src/primitives.cpp:8-26
// plane_tmax: hw4+ reject plane hits at t >= T_MAX = 1e4 (hw4/src/primitives.cpp:8,74)
static inline bool prim_ray3(const Prim &f, V3 o, V3 d, Hit &h, bool plane_tmax = false) {
    V3 to = qtransform(f.rotation, o - f.position), td = qtransform(f.rotation, d);
    bool ok;
    if (f.type == RT_PRIM_ELLIPSOID) {                                     // :49-67
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
    } else if (f.type == RT_PRIM_PLANE) {                                   // :69-79 (no T_MAX in hw3)
        V3 n = f.data;
        float t = -dot(to, n) / dot(td, n);
        ok = t > 0 && (!plane_tmax || t < 1e4f);
        if (ok) h = dot(td, n) > 0 ? Hit{t, neg1(n), true} : Hit{t, n, false};
    } else ok = box_local(f.data, to, td, h);
    if (!ok) return false;
    h.norma = normalize(qtransform(qconj(f.rotation), h.norma));
    return true;
}


static inline V3 v3(const float *p) { return {p[0], p[1], p[2]}; }
static inline Prim prim_from_abi(const rt_primitive &p) {
    Prim f;
    f.type = p.type; f.data = v3(p.data); f.position = v3(p.position);
    f.rotation = Quat{v3(p.rotation), p.rotation[3]};
    f.color = v3(p.color); f.emission = v3(p.emission); f.kind = p.kind; f.ior = p.ior;
    return f;
}
} // namespace rtot
