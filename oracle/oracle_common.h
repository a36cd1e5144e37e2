// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the reference renderer's arithmetic, used solely as the checker in tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
// raytracing-course-hw_amd/ may include, link or call this.
//
// Small float-vector helpers with exactly the reference's operator semantics
// (reference: hw8/src/include/vec3.h:31-80, quaternion.h:31-46).  Build with
// -O3 -ffp-contract=off and NO -ffast-math / -march, like hw8/CMakeLists.txt:4-11 (Release).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#include "../include/rtamd.h"

namespace rto {

struct V3 {
    float x = 0, y = 0, z = 0;
};
// vec3.h:33-47
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
// vec3.h:74-76 — the scalar is a float parameter: double expressions are narrowed first.
static inline V3 operator*(float k, V3 p) { return {k * p.x, k * p.y, k * p.z}; }
// vec3.h:53-55
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vec3.h:57-59 — NEGATED conventional cross product.
static inline V3 crossr(V3 a, V3 o) { return {a.z * o.y - a.y * o.z, a.x * o.z - a.z * o.x, a.y * o.x - a.x * o.y}; }
static inline float len2(V3 a) { return dot(a, a); }
// vec3.h:65-67 — unqualified sqrt on a float resolves to ::sqrt(double); result narrowed to float.
static inline float len(V3 a) { return (float)std::sqrt((double)len2(a)); }
// vec3.h:78-80 — 1. / len() is a double division, narrowed when passed to operator*(float, Vec3).
static inline V3 normalize(V3 a) { return (float)(1. / (double)len(a)) * a; }
static inline V3 neg1(V3 a) { return (float)(-1.) * a; } // "-1. * v"

struct Quat {
    V3 v;
    float w = 1;
};
// quaternion.h:36-38
static inline Quat qmul(Quat a, Quat b) {
    return {a.w * b.v + b.w * a.v + crossr(a.v, b.v), a.w * b.w - dot(a.v, b.v)};
}
static inline Quat qconj(Quat q) { return {(float)(-1.) * q.v, q.w}; }
// quaternion.h:44-46
static inline V3 qtransform(Quat q, V3 p) { return qmul(qmul(q, Quat{p, 0.f}), qconj(q)).v; }

// libstdc++ std::min / std::max semantics (NaN behaviour matters in the slab test).
static inline float smin(float a, float b) { return (b < a) ? b : a; }
static inline float smax(float a, float b) { return (a < b) ? b : a; }

typedef std::minstd_rand rng_t; // hw8/src/include/scene.h:13

// color.cpp:4-31 — epilogue
static inline V3 aces_tonemap(V3 x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    V3 num = x * (a * x + V3{b, b, b});
    V3 den = x * (c * x + V3{d, d, d}) + V3{e, e, e};
    V3 q = num / den;
    return {smin(1.f, smax(0.f, q.x)), smin(1.f, smax(0.f, q.y)), smin(1.f, smax(0.f, q.z))};
}
static inline V3 gamma_corrected(V3 x) {
    float gamma = 1. / 2.2;
    return {(float)std::pow((double)x.x, (double)gamma), (float)std::pow((double)x.y, (double)gamma),
            (float)std::pow((double)x.z, (double)gamma)};
}
static inline void to_extern(V3 c, uint8_t out[3]) {
    out[0] = (uint8_t)std::round((double)(255 * c.x));
    out[1] = (uint8_t)std::round((double)(255 * c.y));
    out[2] = (uint8_t)std::round((double)(255 * c.z));
}

} // namespace rto
