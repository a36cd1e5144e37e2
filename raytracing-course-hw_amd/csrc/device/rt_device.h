// Device-side building blocks of the replay path tracer (gfx950).
//
// Arithmetic contract: this translation unit is compiled with -ffp-contract=off and IEEE
// division/sqrt (hipcc default), so every float/double expression below rounds exactly like the
// reference's x86-64 -O3 build (no FMA, no fast-math).  Where the reference
// promotes to double across SEVERAL operations the doubles are kept; where it promotes around a
// single +,-,*,/ or sqrt the float operation is used (double rounding is innocuous there:
// 53 >= 2*24+2).  FMA is used only where it is spelled fmaf()/fma(): in my own BVH slab test
// (conservative by construction) and in rt_logf, which replays glibc 2.35's FMA build of logf.
#pragma once
#include <hip/hip_runtime.h>
#include "rt_types.h"

#define RT_DEV __device__ __forceinline__

namespace rtamd {
namespace dev {

struct F3 { float x, y, z; };
RT_DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV F3 f3(const float *p) { return f3(p[0], p[1], p[2]); }
RT_DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV F3 operator*(float k, F3 p) { return f3(k * p.x, k * p.y, k * p.z); }
RT_DEV F3 neg(F3 a) { return f3(-a.x, -a.y, -a.z); }                       // "-1. * v"
RT_DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // vec3.h:53-55
// vec3.h:57-59: the reference's cross() is the NEGATED conventional cross product.
RT_DEV F3 crossr(F3 a, F3 o) { return f3(a.z * o.y - a.y * o.z, a.x * o.z - a.z * o.x, a.y * o.x - a.x * o.y); }
RT_DEV float len2(F3 a) { return dot(a, a); }
RT_DEV float len(F3 a) { return sqrtf(len2(a)); }                          // vec3.h:65-67
RT_DEV F3 normalize(F3 a) { return (1.0f / len(a)) * a; }                  // vec3.h:78-80
RT_DEV float smin(float a, float b) { return (b < a) ? b : a; }            // std::min
RT_DEV float smax(float a, float b) { return (a < b) ? b : a; }            // std::max

// ---- RNG: std::minstd_rand + libstdc++ distributions -------------------------
struct Rng {
    uint32_t x;      // engine state
    float saved;     // normal_distribution::_M_saved
    bool has_saved;  // _M_saved_available
};
RT_DEV void rng_seed(Rng &r, uint32_t s) { // linear_congruential_engine::seed: s mod m, 0 -> 1
    uint32_t v = s % 2147483647u;
    r.x = v == 0 ? 1u : v;
    r.saved = 0.f;
    r.has_saved = false;
}
RT_DEV uint32_t rng_next(Rng &r) { // x <- 48271 x mod (2^31 - 1), via 2^31 == 1 (mod m)
    uint64_t p = (uint64_t)r.x * 48271ull;
    uint32_t s = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
    if (s >= 2147483647u) s -= 2147483647u;
    r.x = s;
    return s;
}
// uniform_real_distribution<float>(0,1): generate_canonical<float,24> with one engine call
// (bits/random.tcc:3348-3385): float(x - min) / 2147483648.0f, clamped below 1.
RT_DEV float rng_u01(Rng &r) {
    float s = __uint2float_rn(rng_next(r) - 1u);
    float v = s * 4.656612873077392578125e-10f; // exact: division by 2^31
    return v >= 1.0f ? 0.99999994f : v;
}

// glibc 2.35 logf, FMA build (sysdeps/ieee754/flt-32/e_logf.c selected by the x86_64 multiarch
// ifunc on CPUs with FMA+AVX2): table + degree-3 polynomial in double, contracted exactly as
// in that build.  Bit-identical to the host logf for every positive normal float
// (tests/test_device_math.py checks it exhaustively against libm on the CPU).
struct LogfTab { double invc, logc; };
__device__ const LogfTab rt_logf_tab[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
RT_DEV float rt_logf(float x) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        ix = __float_as_uint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    int k = (int)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc = rt_logf_tab[i].invc, logc = rt_logf_tab[i].logc;
    double z = (double)__uint_as_float(iz);
    double r = fma(z, invc, -1.0);
    double y0 = fma((double)k, 0x1.62e42fefa39efp-1, logc);
    double r2 = r * r;
    double y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = fma(-0x1.00ea348b88334p-2, r2, y);
    y = fma(y, r2, y0 + r);
    return (float)y;
}

// normal_distribution<float>(0,1): Marsaglia polar with the saved second value
// (bits/random.tcc:1802-1838).
RT_DEV float rng_n01(Rng &r) {
    if (r.has_saved) {
        r.has_saved = false;
        return r.saved;
    }
    float x, y, r2;
    do {
        x = 2.0f * rng_u01(r) - 1.0f;
        y = 2.0f * rng_u01(r) - 1.0f;
        r2 = x * x + y * y;
    } while (r2 > 1.0f || r2 == 0.0f);
    float mult = sqrtf(-2 * rt_logf(r2) / r2);
    r.saved = x * mult;
    r.has_saved = true;
    return y * mult;
}

// ---- quaternion (quaternion.h:36-46) ------------------------------------------------------------
struct Quat { F3 v; float w; };
RT_DEV Quat qmul(Quat a, Quat b) {
    Quat r;
    r.v = a.w * b.v + b.w * a.v + crossr(a.v, b.v);
    r.w = a.w * b.w - dot(a.v, b.v);
    return r;
}
RT_DEV Quat qconj(Quat q) { Quat r; r.v = neg(q.v); r.w = q.w; return r; }
RT_DEV F3 qtransform(Quat q, F3 p) {
    Quat pq; pq.v = p; pq.w = 0.f;
    return qmul(qmul(q, pq), qconj(q)).v;
}

// ---- ray / triangle (primitives.cpp:18-27,76-125) -----------------------------------------------
#define RT_T_MAX 1e4f
#define RT_MAGIC1_0 0.239f
#define RT_MAGIC1_1 0.419f
#define RT_MAGIC1_2 0.533f
#define RT_MAGIC2_0 0.35743f
#define RT_MAGIC2_1 0.66682f
#define RT_MAGIC2_2 0.69695f

// Bit-exact Figure::intersectAsTriangle up to the barycentric rejection; attribute interpolation
// is deferred to the hit that survives (shade_fetch).
RT_DEV bool tri_test(const TriIsect &T, F3 o, F3 d, float &t, float &u, float &v, bool &inside) {
    F3 a = f3(T.ax, T.ay, T.az), n = f3(T.nx, T.ny, T.nz);
    F3 ro = o - a;
    float dn = dot(d, n);
    t = -dot(ro, n) / dn;
    if (!(t > 0 && t < RT_T_MAX)) return false;
    inside = dn > 0;
    F3 p = ro + t * d;
    float c1 = RT_MAGIC1_0 * p.x + RT_MAGIC1_1 * p.y + RT_MAGIC1_2 * p.z;
    float c2 = RT_MAGIC2_0 * p.x + RT_MAGIC2_1 * p.y + RT_MAGIC2_2 * p.z;
    float yy = (c1 * T.a2 - c2 * T.a1) / T.den;
    float xx = T.a2 == 0 ? (c1 - T.b1 * yy) / T.a1 : (c2 - T.b2 * yy) / T.a2;
    u = xx; v = yy;
    if (u < 0 || v < 0 || u + v > 1) return false;
    return true;
}

// tri_test for a closest-hit walk that already holds a hit: a triangle whose plane lies beyond keep_t (the best t plus the walkers'
// look-behind, rt_exact.h) can neither become the closest hit nor matter as its runner-up whatever its barycentrics are, so their
// two divisions are skipped.  Exactly the same accept / reject decisions as tri_test for every triangle that could still matter.
RT_DEV bool tri_test_closer(const TriIsect &T, F3 o, F3 d, float keep_t, float &t, float &u, float &v, bool &inside) {
    F3 a = f3(T.ax, T.ay, T.az), n = f3(T.nx, T.ny, T.nz);
    F3 ro = o - a;
    float dn = dot(d, n);
    t = -dot(ro, n) / dn;
    if (!(t > 0 && t < RT_T_MAX)) return false;
    if (t > keep_t) return false;
    inside = dn > 0;
    F3 p = ro + t * d;
    float c1 = RT_MAGIC1_0 * p.x + RT_MAGIC1_1 * p.y + RT_MAGIC1_2 * p.z;
    float c2 = RT_MAGIC2_0 * p.x + RT_MAGIC2_1 * p.y + RT_MAGIC2_2 * p.z;
    float yy = (c1 * T.a2 - c2 * T.a1) / T.den;
    float xx = T.a2 == 0 ? (c1 - T.b1 * yy) / T.a1 : (c2 - T.b2 * yy) / T.a2;
    u = xx; v = yy;
    if (u < 0 || v < 0 || u + v > 1) return false;
    return true;
}

RT_DEV TriIsect load_isect(const TriIsect *p) {
    const float4 *q = reinterpret_cast<const float4 *>(p);
    float4 a = q[0], b = q[1], c = q[2];
    TriIsect T;
    T.ax = a.x; T.ay = a.y; T.az = a.z; T.nx = a.w;
    T.ny = b.x; T.nz = b.y; T.a1 = b.z; T.b1 = b.w;
    T.a2 = c.x; T.b2 = c.y; T.den = c.z; T.pad = __float_as_uint(c.w);
    return T;
}

// ---- my BVH: conservative two-box slab test -------------------------------------------------------
struct RayInv { F3 o, inv; };
RT_DEV RayInv make_ray_inv(F3 o, F3 d) {
    RayInv r;
    r.o = o;
    // A zero (or denormal) direction component would give inf*0 = NaN in the slab products.
    float dx = fabsf(d.x) > 1e-30f ? d.x : copysignf(1e-30f, d.x);
    float dy = fabsf(d.y) > 1e-30f ? d.y : copysignf(1e-30f, d.y);
    float dz = fabsf(d.z) > 1e-30f ? d.z : copysignf(1e-30f, d.z);
    r.inv = f3(1.0f / dx, 1.0f / dy, 1.0f / dz);
    return r;
}
// Returns whether [tmin,tmax] (widened by 4 ulp-ish) overlaps [0, tbest]; boxes are host-padded.
RT_DEV bool slab_test(float4 lo, float4 hi, const RayInv &r, float tbest, float &tnear) {
    float t0x = (lo.x - r.o.x) * r.inv.x, t1x = (hi.x - r.o.x) * r.inv.x;
    float t0y = (lo.y - r.o.y) * r.inv.y, t1y = (hi.y - r.o.y) * r.inv.y;
    float t0z = (lo.z - r.o.z) * r.inv.z, t1z = (hi.z - r.o.z) * r.inv.z;
    float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    tmin = fmaf(-fabsf(tmin), 4.8e-7f, tmin);
    tmax = fmaf(fabsf(tmax), 4.8e-7f, tmax);
    tnear = tmin;
    return (tmin <= tmax) & (tmax >= 0.f) & (tmin <= tbest);
}

typedef float rt_f2 __attribute__((ext_vector_type(2)));
// The walk nodes on the 16-bit grid (rt_types.h GpuNode4Q).  The ray moves into grid coordinates once per walk,
//     o' = (o - grid.lo) / step,   1/d' = step / d      (t keeps its meaning: x = o + t d  <=>  x' = o' + t d'),
// and a child's slabs are (cell - o') * (1/d'): cells are small integers, exact in a float.  What rounds: o' — three roundings, at most
// 3 * 2^-24 * 65,536 = 0.012 cells for an origin inside the grid (the grid covers the scene's boxes and the camera, so every ray origin)
// — and relative errors of the subtraction, of 1/d' (a product and a 1-ulp reciprocal) and of the product, 6 * 2^-24 of a
// slab's t: since origin and box both lie on the grid, that t is at most 65,536 cells' worth along its axis, so 0.02 cells.  The extra
// cell on either side of every box (rt_node_grid.h grid_axis_word) covers both, and the interval needs no widening of its own.
// sx, sy, sz: byte selectors (v_perm_b32) that put the slab the ray meets first into the low half of a record's axis word — lo | hi << 16
// as stored for a direction component >= 0, the halves swapped for a negative one — so that the test needs no min / max per axis.
struct RayGrid { float ox, oy, oz, ix, iy, iz; uint32_t sx, sy, sz; };
RT_DEV RayGrid ray_grid_idle() { // what an idle lane holds (never used); field by field: an aggregate constant would be copied from memory
    RayGrid g;
    g.ox = 0.f; g.oy = 0.f; g.oz = 0.f; g.ix = 1.f; g.iy = 1.f; g.iz = 1.f; g.sx = 0x03020100u; g.sy = 0x03020100u; g.sz = 0x03020100u;
    return g;
}
#define RT_GRID_RAY_IDLE ray_grid_idle()
RT_DEV RayGrid make_ray_grid(const NodeGrid &G, F3 o, F3 d) {
    // reciprocals by v_rcp_f32 (1 ulp): part of the relative error the boxes' extra cell covers (below); a zero (or denormal) component is
    // replaced as in make_ray_inv
    const float dx = fabsf(d.x) > 1e-30f ? d.x : copysignf(1e-30f, d.x), dy = fabsf(d.y) > 1e-30f ? d.y : copysignf(1e-30f, d.y), dz = fabsf(d.z) > 1e-30f ? d.z : copysignf(1e-30f, d.z);
    RayGrid g;
    g.ox = (o.x - G.lo[0]) * G.istep[0]; g.oy = (o.y - G.lo[1]) * G.istep[1]; g.oz = (o.z - G.lo[2]) * G.istep[2];
    g.ix = G.step[0] * __builtin_amdgcn_rcpf(dx); g.iy = G.step[1] * __builtin_amdgcn_rcpf(dy); g.iz = G.step[2] * __builtin_amdgcn_rcpf(dz);
    g.sx = g.ix < 0.f ? 0x01000302u : 0x03020100u; g.sy = g.iy < 0.f ? 0x01000302u : 0x03020100u; g.sz = g.iz < 0.f ? 0x01000302u : 0x03020100u;
    return g;
}
// b = one child record of a GpuNode4Q: x, y, z words (lo | hi << 16) and the child word.  Pairs (lo, hi) per axis: packed subtract and multiply.
RT_DEV bool slab_test_q(uint4 b, const RayGrid &r, float tbest, float &tnear) {
    const uint32_t wx = __builtin_amdgcn_perm(b.x, b.x, r.sx), wy = __builtin_amdgcn_perm(b.y, b.y, r.sy), wz = __builtin_amdgcn_perm(b.z, b.z, r.sz);
    const rt_f2 x = (rt_f2{(float)(wx & 0xFFFFu), (float)(wx >> 16)} - r.ox) * r.ix; // (entry, exit) along x: the products keep that order
    const rt_f2 y = (rt_f2{(float)(wy & 0xFFFFu), (float)(wy >> 16)} - r.oy) * r.iy;
    const rt_f2 z = (rt_f2{(float)(wz & 0xFFFFu), (float)(wz >> 16)} - r.oz) * r.iz;
    const float tmin = fmaxf(fmaxf(x.x, y.x), z.x);
    const float tmax = fminf(fminf(x.y, y.y), z.y);
    tnear = tmin;
    return (tmin <= tmax) & (tmax >= 0.f) & (tmin <= tbest);
}

#define RT_LEAF_BIT 0x80000000u
#define RT_EMPTY_LEAF 0xFFFFFFFFu
#define RT_STACK_SIZE 64

struct HitRec {
    int idx;       // BVH-order triangle index or -1
    float t, u, v;
    bool inside;
};

struct Counters { unsigned long long closest, lightq, nodes, tris; };

// Closest hit with the reference's tie rule: smallest t, equal t -> lowest figure index
// (bvh.h:111-142 visits figures in increasing index order and replaces only on strict '<').
// STRIDE: distance in words between consecutive stack entries (1 = a private array; see ref_closest_hit, rt_exact.h)
template <bool COUNT, int STRIDE = 1>
RT_DEV HitRec closest_hit(const SceneView &S, F3 o, F3 d, uint32_t *stack, Counters &cnt) {
    HitRec best;
    best.idx = -1; best.t = RT_T_MAX; best.u = 0.f; best.v = 0.f; best.inside = false;
    if (COUNT) cnt.closest++;
    RayInv ray = make_ray_inv(o, d);
    int sp = 0;
    uint32_t cur = 0; // root is always an inner node
    for (;;) {
        if (!(cur & RT_LEAF_BIT)) {
            const float4 *q = reinterpret_cast<const float4 *>(S.nodes + cur);
            float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
            if (COUNT) cnt.nodes++;
            float n0, n1;
            bool h0 = slab_test(lo0, hi0, ray, best.t, n0);
            bool h1 = slab_test(lo1, hi1, ray, best.t, n1);
            uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
            if (h0 & h1) {
                bool swap = n1 < n0;
                stack[STRIDE * sp++] = swap ? c0 : c1;
                cur = swap ? c1 : c0;
                continue;
            }
            if (h0) { cur = c0; continue; }
            if (h1) { cur = c1; continue; }
        } else if (cur != RT_EMPTY_LEAF) {
            uint32_t i = cur & ~RT_LEAF_BIT;
            for (;;) {
                TriIsect T = load_isect(S.tri_walk + i);
                if (COUNT) cnt.tris++;
                float t, u, v; bool inside;
                const int fi = (int)(T.pad >> 1); // index in the figure order
                if (tri_test(T, o, d, t, u, v, inside) && (t < best.t || (t == best.t && fi < best.idx))) {
                    best.t = t; best.u = u; best.v = v; best.inside = inside; best.idx = fi;
                }
                if (T.pad & 1u) break; // last triangle of this leaf
                i++;
            }
        }
        if (sp == 0) break;
        cur = stack[STRIDE * --sp];
    }
    return best;
}

// ---- light pdf: all-hits sum over the light BVH in the reference's association -------------------
// FiguresMix::getTotalPdf (distributions.h:148-165) returns total(left) + total(right) recursively and
// a sequential sum inside a leaf; float addition is not associative, so the same tree of additions
// is replayed with an explicit frame stack: TODO(child) frames and ADD(partial) frames (tag bit in
// `addmask`).  Misses contribute +0, which is the additive identity here (no term is -0).
RT_DEV float light_pdf_one(const LightRec *L, F3 x, F3 d, bool &last, bool geometric_normal = false) {
    TriIsect T = load_isect(&L->isect);
    last = T.pad != 0;
    float t, u, v; bool inside;
    if (!tri_test(T, x, d, t, u, v, inside)) return 0.f;
    // (t cannot be NaN here: tri_test requires t > 0; distributions.h:141-143 is dead code)
    const float4 *q = reinterpret_cast<const float4 *>(L) + 3;
    float4 q1 = q[1], q2 = q[2], q3 = q[3];
    // (q[0] = b.xyz c.x, used by the sampler only) | q1 = c.yz point_prob n3.x | q2 = n3.yz dn1.xy | q3 = dn1.z dn2.xyz
    float point_prob = q1.z;
    F3 n3 = f3(q1.w, q2.x, q2.y), dn1 = f3(q2.z, q2.w, q3.x), dn2 = f3(q3.y, q3.z, q3.w);
    F3 sn = n3 + u * dn1 + v * dn2;           // primitives.cpp:110
    sn = normalize(sn);                        // :117
    if (inside) sn = neg(sn);                  // :118-119
    if (geometric_normal) { F3 n = f3(T.nx, T.ny, T.nz); sn = normalize(inside ? neg(n) : n); } // hw7: yn of the intersection
    F3 y = x + t * d;                          // distributions.h:144
    return point_prob * len2(x - y) / fabsf(dot(d, sn)); // :68-70 (pdfOne, shading normal in hw8)
}

template <bool COUNT, int STRIDE = 1>
RT_DEV float light_pdf_sum(const SceneView &S, F3 x, F3 d, uint32_t *stack, Counters &cnt) {
    if (COUNT) cnt.lightq++;
    RayInv ray = make_ray_inv(x, d);
    int sp = 0;
    unsigned long long addmask = 0;
    uint32_t cur = 0;
    bool descending = true;
    float v = 0.f;
    for (;;) {
        if (descending) {
            if (cur & RT_LEAF_BIT) {
                float result = 0.f;
                if (cur != RT_EMPTY_LEAF) {
                    uint32_t i = cur & ~RT_LEAF_BIT;
                    for (;;) {
                        bool last;
                        if (COUNT) cnt.tris++;
                        result += light_pdf_one(S.lights + i, x, d, last, S.hw7 != 0);
                        if (last) break;
                        i++;
                    }
                }
                v = result;
                descending = false;
                continue;
            }
            const float4 *q = reinterpret_cast<const float4 *>(S.light_nodes + cur);
            float4 lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
            if (COUNT) cnt.nodes++;
            float n0, n1;
            bool h0 = slab_test(lo0, hi0, ray, RT_T_MAX, n0);
            bool h1 = slab_test(lo1, hi1, ray, RT_T_MAX, n1);
            uint32_t c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(lo1.w);
            if (h0 & h1) { addmask &= ~(1ull << sp); stack[STRIDE * sp++] = c1; cur = c0; }
            else if (h0) cur = c0;
            else if (h1) cur = c1;
            else { v = 0.f; descending = false; }
        } else {
            if (sp == 0) break;
            --sp;
            uint32_t f = stack[STRIDE * sp];
            if ((addmask >> sp) & 1ull) v = __uint_as_float(f) + v;       // left total + right total
            else { addmask |= 1ull << sp; stack[STRIDE * sp++] = __float_as_uint(v); cur = f; descending = true; }
        }
    }
    return v;
}

// ---- textures (scene.cpp:9-53) --------------------------------------------------------------------
// Texels are RGB8; one (unaligned) 4-byte load fetches all three channels — the texel array ends with four spare bytes (scene_prep.cpp).
// 32-bit offsets: an image has fewer than 2^32 / 3 texels (checked at scene preparation).
RT_DEV F3 load_texel(const SceneView &S, const GpuImage &im, int ix, int iy, bool srgb) {
    const uint32_t off = 3u * ((uint32_t)ix + (uint32_t)im.width * (uint32_t)iy);
    if (off + 2u >= (uint32_t)im.width * (uint32_t)im.height * 3u) return f3(0.f, 0.f, 0.f); // reference reads OOB here (UB)
    uint32_t w;
    __builtin_memcpy(&w, S.texels + im.offset + off, 4);
    const uint32_t r = w & 255u, g = (w >> 8) & 255u, b = (w >> 16) & 255u;
    if (srgb) return f3(S.srgb_lut[r], S.srgb_lut[g], S.srgb_lut[b]);
    const float k = (float)(1. / 255);
    return f3(k * (1.f * (float)r), k * (1.f * (float)g), k * (1.f * (float)b));
}
RT_DEV F3 sample_texture(const SceneView &S, GpuImage im, float tx, float ty, bool srgb); // with the image's descriptor already loaded
RT_DEV F3 sample_texture(const SceneView &S, int slot, float tx, float ty, bool srgb) { return sample_texture(S, S.images[slot], tx, ty, srgb); }
RT_DEV F3 sample_texture(const SceneView &S, GpuImage im, float tx, float ty, bool srgb) {
    tx -= floorf(tx);
    ty -= floorf(ty);
    tx *= im.width;
    ty *= im.height;
    // (i + 1) % size without the integer division: i lies in [0, size] (size itself when the product rounds up), so one subtraction wraps it
    int ix1 = (int)floorf(tx), ix2 = ix1 + 1; if (ix2 >= im.width) ix2 -= im.width;
    int iy1 = (int)floorf(ty), iy2 = iy1 + 1; if (iy2 >= im.height) iy2 -= im.height;
    float dx = tx - ix1;
    float dy = ty - iy1;
    F3 p11 = load_texel(S, im, ix1, iy1, srgb);
    F3 p12 = load_texel(S, im, ix1, iy2, srgb);
    F3 p21 = load_texel(S, im, ix2, iy1, srgb);
    F3 p22 = load_texel(S, im, ix2, iy2, srgb);
    return (1 - dx) * ((1 - dy) * p11 + dy * p12) + dx * ((1 - dy) * p21 + dy * p22);
}
RT_DEV F3 apply_normal_map(F3 sn, F3 tan, float tanw, F3 sample) {
    F3 lx = tan, lz = sn;
    F3 ly = tanw * crossr(lx, lz);
    F3 ln = 2.f * sample - f3(1.f, 1.f, 1.f);
    F3 n = ln.x * lx + ln.y * ly + ln.z * lz;
    return normalize(n);
}

// ---- BRDF (material.h:11-65) -----------------------------------------------------------------------
#define RT_PI 3.14159265358979323846
RT_DEV float distribution_term(F3 h, F3 n, float alpha2) {
    float dotHN = dot(h, n);
    if (dotHN <= 0) return 0.f;
    double q = (double)smax(0.f, (alpha2 - 1) * dotHN * dotHN + 1);
    return (float)((double)alpha2 / (RT_PI * (q * q)));
}
RT_DEV float v1_term(F3 n, F3 x, float alpha2) {
    float nx = dot(n, x);
    return (float)(1. / (fabs((double)nx) + sqrt((double)smax(0.f, alpha2 + (1 - alpha2) * nx * nx))));
}
RT_DEV float specular_brdf(F3 l, F3 v, F3 n, float alpha2) {
    F3 h = normalize(l + v);
    if ((double)dot(h, l) < 1e-4 || (double)dot(h, v) < 1e-4) return 0.f;
    return distribution_term(h, n, alpha2) * v1_term(n, l, alpha2) * v1_term(n, v, alpha2);
}
RT_DEV float pow5_d2f(float b) { // (float)pow((double)b, 5.0): three double products, <= 1.5 ulp(double)
    double x = (double)b, x2 = x * x;
    return (float)(x2 * x2 * x);
}
RT_DEV F3 fresnel_term(F3 f0, F3 f90, F3 v, F3 h) {
    float k = pow5_d2f(smax(0.f, 1.f - fabsf(dot(v, h))));
    return f0 + k * (f90 - f0);
}
RT_DEV F3 material_brdf(F3 base_color, float base_metallic, F3 l, F3 v, F3 n, F3 color, float metallic, float alpha) {
    F3 h = normalize(l + v);
    float specular = specular_brdf(l, v, n, alpha * alpha);
    F3 metal = f3(0.f, 0.f, 0.f), dielectric = f3(0.f, 0.f, 0.f);
    metallic *= base_metallic;
    if (metallic > 0 && dot(v, n) >= 0 && dot(l, n) >= 0) {
        F3 ft = fresnel_term(base_color * color, f3(1.f, 1.f, 1.f), v, h);
        metal = specular * ft;
    }
    if (metallic < 1) {
        F3 diffuse = f3(0.f, 0.f, 0.f);
        if (dot(l, n) >= 0) diffuse = (float)(1. / RT_PI) * (base_color * color);
        F3 ft = fresnel_term(f3(0.04f, 0.04f, 0.04f), f3(1.f, 1.f, 1.f), v, h);
        dielectric = diffuse * (f3(1.f, 1.f, 1.f) - ft) + specular * ft;
    }
    return (1.0f - metallic) * dielectric + metallic * metal;
}

// material_brdf with its two products formed by the caller (bc = base_color * color, metallic = texture metallic * baseMetallic —
// the same single multiplications, so the same values): the shader computes them before it samples a direction and so carries
// four values instead of eight through the sampling code.
RT_DEV F3 material_brdf_pre(F3 bc, float metallic, F3 l, F3 v, F3 n, float alpha) {
    F3 h = normalize(l + v);
    float specular = specular_brdf(l, v, n, alpha * alpha);
    F3 metal = f3(0.f, 0.f, 0.f), dielectric = f3(0.f, 0.f, 0.f);
    if (metallic > 0 && dot(v, n) >= 0 && dot(l, n) >= 0) {
        F3 ft = fresnel_term(bc, f3(1.f, 1.f, 1.f), v, h);
        metal = specular * ft;
    }
    if (metallic < 1) {
        F3 diffuse = f3(0.f, 0.f, 0.f);
        if (dot(l, n) >= 0) diffuse = (float)(1. / RT_PI) * bc;
        F3 ft = fresnel_term(f3(0.04f, 0.04f, 0.04f), f3(1.f, 1.f, 1.f), v, h);
        dielectric = diffuse * (f3(1.f, 1.f, 1.f) - ft) + specular * ft;
    }
    return (1.0f - metallic) * dielectric + metallic * metal;
}

// hw7/src/include/material.h:44-61: per-material colour and metallic only, and none of hw8's v.n / l.n gates
RT_DEV F3 material_brdf_hw7(F3 base_color, float base_metallic, F3 l, F3 v, F3 n, float alpha2) {
    F3 h = normalize(l + v);
    float specular = specular_brdf(l, v, n, alpha2);
    F3 metal = f3(0.f, 0.f, 0.f), dielectric = f3(0.f, 0.f, 0.f);
    float metallic = base_metallic;
    if (metallic > 0) metal = specular * fresnel_term(base_color, f3(1.f, 1.f, 1.f), v, h);
    if (metallic < 1) {
        F3 diffuse = (float)(1. / RT_PI) * base_color;
        F3 ft = fresnel_term(f3(0.04f, 0.04f, 0.04f), f3(1.f, 1.f, 1.f), v, h);
        dielectric = diffuse * (f3(1.f, 1.f, 1.f) - ft) + specular * ft;
    }
    return (1.0f - metallic) * dielectric + metallic * metal;
}

// ---- samplers (distributions.h) --------------------------------------------------------------------
RT_DEV F3 cosine_sample(Rng &rng, F3 n) { // :42-52
    float a = rng_n01(rng), b = rng_n01(rng), c = rng_n01(rng);
    F3 d = normalize(f3(a, b, c));
    d = d + n;
    float l = len(d);
    const float ceps = 1e-9f;
    if (l <= ceps || dot(d, n) <= ceps || l != l) return n;
    return (1.0f / l) * d;
}
RT_DEV float cosine_pdf(F3 n, F3 d) { return smax(0.f, dot(d, n) / (float)RT_PI); } // :54-57

RT_DEV Quat vndf_getq(F3 n) { // :212-224
    Quat q;
    float nz = dot(n, f3(0.f, 0.f, 1.f));
    if ((double)nz > 0.9999) { q.v = f3(0.f, 0.f, 0.f); q.w = 1.f; return q; }
    if ((double)nz < -0.9999) { q.v = f3(0.f, 0.f, 0.f); q.w = -1.f; return q; }
    F3 a = crossr(n, f3(0.f, 0.f, 1.f));
    float w = (float)(sqrt((double)len2(n)) + (double)nz);
    float l = sqrtf(len2(a) + w * w);
    q.v = (1.0f / l) * a;
    q.w = w / l;
    return q;
}
// The two uniform draws of the sampler and the point they give on the unit disk (:178-182).  Split from the rest so that a caller can
// take the draws — and the double-precision sin / cos, the register-hungriest part — before it builds the frame around v: the random
// stream and every value are the same in either order.
RT_DEV void vndf_disk_point(Rng &rng, float &t1, float &t2) {
    float u1 = rng_u01(rng), u2 = rng_u01(rng);
    float r = sqrtf(u1);
    float phi = (float)(2.0 * RT_PI * (double)u2);
    t1 = (float)((double)r * cos((double)phi));
    t2 = (float)((double)r * sin((double)phi));
}
RT_DEV F3 vndf_sample_local(float t1, float t2, F3 v, float alpha) { // :170-194 (Heitz 2018)
    F3 vh = normalize(f3(alpha * v.x, alpha * v.y, v.z));
    float lensq = vh.x * vh.x + vh.y * vh.y;
    F3 T1 = lensq > 0 ? (float)(1. / sqrt((double)lensq)) * f3(-vh.y, vh.x, 0.f) : f3(1.f, 0.f, 0.f);
    F3 T2 = crossr(T1, vh);
    float s = 0.5f * (1.0f + vh.z);
    t2 = (float)((1.0 - (double)s) * sqrt((double)(1.f - t1 * t1)) + (double)(s * t2));
    float rem = (float)(1.0 - (double)(t1 * t1) - (double)(t2 * t2));
    F3 nh = t1 * T1 + t2 * T2 + sqrtf(smax(0.f, rem)) * vh;
    F3 ne = normalize(f3(alpha * nh.x, alpha * nh.y, smax(0.0f, nh.z)));
    return (2 * dot(ne, v)) * ne - v;
}
RT_DEV float vndf_D(F3 n, float a) { // :196-198
    double q = (double)(n.x * n.x / (a * a) + n.y * n.y / (a * a) + n.z * n.z);
    return (float)(1. / (RT_PI * (double)a * (double)a * (q * q)));
}
RT_DEV float vndf_G1(F3 v, float a) { // :200-203
    float q = 1 + (a * a * v.x * v.x + a * a * v.y * v.y) / (v.z * v.z);
    float lambda = (float)(0.5 * (-1 + sqrt((double)q)));
    return 1.0f / (1 + lambda);
}
RT_DEV float vndf_pdf_local(F3 d, F3 v, float a) { // :205-210
    F3 ni = normalize(v + d);
    float dv = vndf_G1(v, a) * smax(0.f, dot(v, ni)) * vndf_D(ni, a) / fabsf(v.z);
    return dv / (4 * dot(v, ni));
}
RT_DEV F3 vndf_sample(Rng &rng, F3 n, F3 v, float alpha) { // :229-237
    float t1, t2;
    vndf_disk_point(rng, t1, t2);
    v = neg(v);
    Quat q = vndf_getq(n);
    F3 vT = qtransform(q, v);
    F3 dT = vndf_sample_local(t1, t2, vT, alpha);
    return qtransform(qconj(q), dT);
}
RT_DEV float vndf_pdf(F3 n, F3 d, F3 v, float alpha) { // :239-245
    v = neg(v);
    Quat q = vndf_getq(n);
    return vndf_pdf_local(qtransform(q, d), qtransform(q, v), alpha);
}
RT_DEV F3 light_sample(const SceneView &S, Rng &rng, F3 x) { // :117-120, :81-94
    int k = (int)(rng_u01(rng) * S.n_lights_f);
    const LightRec *L = S.lights + k;
    const float4 *q = reinterpret_cast<const float4 *>(L);
    float4 a4 = q[0], q0 = q[3], q1 = q[4];
    F3 a = f3(a4.x, a4.y, a4.z), b = f3(q0.x, q0.y, q0.z), c = f3(q0.w, q1.x, q1.y);
    float u = rng_u01(rng);
    float v = rng_u01(rng);
    if ((double)(u + v) > 1.) { u = 1 - u; v = 1 - v; }
    F3 point = a + u * b + v * c;
    return normalize(point - x);
}

// ---- epilogue (color.cpp:4-31) -----------------------------------------------------------------------
RT_DEV float aces1(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float q = (x * (a * x + b)) / (x * (c * x + d) + e);
    return smin(1.f, smax(0.f, q));
}
RT_DEV uint8_t tonemap1(float x) {
    const float gamma = (float)(1. / 2.2);
    float g = (float)pow((double)aces1(x), (double)gamma);
    return (uint8_t)round((double)(255 * g));
}

// Pixel slot -> pixel: slots run through the 8x8 sub-tiles of this shard's tiles (slot >> 6 = sub-tile, slot & 63 = pixel in it).
// n / d and n % d for n < 2^24 (exact as a float) and a wave-uniform d: a multiplication by the reciprocal and one correction step instead of
// the ~30 instructions of a 32-bit integer division.
RT_DEV uint32_t udiv24(uint32_t n, uint32_t d, float rcp_d, uint32_t &rem) {
    uint32_t q = (uint32_t)((float)n * rcp_d);
    int r = (int)(n - q * d);
    if (r < 0) { q--; r += (int)d; } else if (r >= (int)d) { q++; r -= (int)d; }
    rem = (uint32_t)r;
    return q;
}
RT_DEV void slot_to_pixel(const RenderView &R, uint32_t slot, int &x, int &y, bool &inside, size_t &out_index) {
    const int sub_x = R.tile_w >> 3, sub_per_tile = sub_x * (R.tile_h >> 3);
    uint32_t w = slot >> 6, lane = slot & 63u; // w < 2^24: a launch holds fewer than 2^30 slots
    uint32_t sub, st = udiv24(w, (uint32_t)sub_per_tile, __builtin_amdgcn_rcpf((float)sub_per_tile), sub);
    uint32_t gt = R.shard_count > 1 ? (uint32_t)R.shard_index + st * (uint32_t)R.shard_count : st;
    uint32_t gtx, gty, sx, sy;
    if (gt < (1u << 24)) gty = udiv24(gt, (uint32_t)R.tiles_x, __builtin_amdgcn_rcpf((float)R.tiles_x), gtx);
    else { gty = gt / (uint32_t)R.tiles_x; gtx = gt % (uint32_t)R.tiles_x; }
    sy = udiv24(sub, (uint32_t)sub_x, __builtin_amdgcn_rcpf((float)sub_x), sx);
    int tx0 = (int)gtx * R.tile_w, ty0 = (int)gty * R.tile_h;
    int lx = (int)sx * 8 + (int)(lane & 7), ly = (int)sy * 8 + (int)(lane >> 3);
    x = tx0 + lx; y = ty0 + ly;
    inside = x < R.width && y < R.height;
    out_index = R.shard_count > 1 ? ((size_t)st * R.tile_h + ly) * R.tile_w + lx : (size_t)y * R.width + x;
}

} // namespace dev
} // namespace rtamd
