// hw8 render kernel: the per-pixel, per-sample loop of Scene::getPixel / Scene::getColor
// (hw8/src/scene.cpp:84-177) plus the tonemap epilogue (hw8/src/color.cpp), one pixel per lane,
// replaying the reference's per-pixel minstd_rand stream.  Waves are persistent and pull 8x8 pixel
// tiles from a global queue (the GPU form of `#pragma omp parallel for schedule(dynamic,8)`,
// hw8/src/sceneio.cpp:387).
#pragma once
#include "rt_device.h"

namespace rtamd {
namespace dev {

#define RT_MAX_DEPTH 16

struct Shaded {
    F3 emission, color, sn;
    float alpha, metallic;
};

// The hit triangle's geometric normal, flipped towards the ray's side (primitives.cpp:21-24,123).
RT_DEV F3 geom_normal(F3 n, bool inside) { return normalize(inside ? neg(n) : n); }
RT_DEV F3 load_tri_normal(const SceneView &S, uint32_t fig) { // n = b x c of TriIsect (words 3..5)
    const float4 *qi = reinterpret_cast<const float4 *>(S.tri_isect + fig);
    const float4 i0 = qi[0], i1 = qi[1];
    return f3(i0.w, i1.x, i1.y);
}
RT_DEV F3 geom_normal(const SceneView &S, const HitRec &h) { return geom_normal(load_tri_normal(S, (uint32_t)h.idx), h.inside); }

// Everything Scene::getColor does between the intersection and the direction sampling
// (scene.cpp:99-149): interpolate attributes, fetch the material, sample the textures, normal map.
// TriShade quads 4..6 (dt2.yz uv3.xy | duv1.xy duv2.xy | tanw material orig pad) of the hit figure: what shade_fetch_attr needs first.  A caller
// that knows the figure early (the persistent shader: from the hit word, before its gate) loads them along with everything else whose
// address it knows, so that one memory latency covers them all.
struct TriShadeHead { float4 s4, s5, s6; };
RT_DEV TriShadeHead load_shade_head(const SceneView &S, uint32_t fig) {
    const float4 *q = reinterpret_cast<const float4 *>(S.tri_shade + fig);
    TriShadeHead H; H.s4 = q[4]; H.s5 = q[5]; H.s6 = q[6];
    return H;
}

RT_DEV void shade_fetch_attr(const SceneView &S, const HitRec &h, const TriShadeHead &H, Shaded &sh, F3 &base_color, float &base_metallic, bool hw7) {
    // TriShade: s0 = n3.xyz dn1.x | s1 = dn1.yz dn2.xy | s2 = dn2.z t3.xyz | s3 = dt1.xyz dt2.x
    //           s4 = dt2.yz uv3.xy | s5 = duv1.xy duv2.xy | s6 = tanw material orig pad
    // Read in the order of use — texture coordinates and material first, normal and tangent bases after the texture taps — so that the
    // bases do not sit in registers while the taps run (the interpolations themselves are the reference's, primitives.cpp:110-119).
    const float4 *q = reinterpret_cast<const float4 *>(S.tri_shade + h.idx);
    const float u = h.u, v = h.v;
    const float tu = H.s4.z + u * H.s5.x + v * H.s5.z;    // :111-114
    const float tv = H.s4.w + u * H.s5.y + v * H.s5.w;
    const float tanw = H.s6.x;
    const uint32_t mat = __float_as_uint(H.s6.y);
    const float4 *qm = reinterpret_cast<const float4 *>(S.materials + mat);
    float4 m0 = qm[0], m1 = qm[1], m2 = qm[2];
    base_color = f3(m0.x, m0.y, m0.z); base_metallic = m0.w;
    int tex_color = (int)__float_as_uint(m2.x), tex_emis = (int)__float_as_uint(m2.y);
    int tex_mr = (int)__float_as_uint(m2.z), tex_nrm = (int)__float_as_uint(m2.w);
    F3 mr = f3(1.f, 1.f, 1.f), ns = f3(0.5f, 0.5f, 1.f);
    sh.color = f3(1.f, 1.f, 1.f);
    sh.emission = f3(m1.x, m1.y, m1.z);
    if (!hw7) {
        // the four image descriptors go out together (slot 0 stands in for an absent texture: the load is then unused)
        const GpuImage im_c = S.images[tex_color >= 0 ? tex_color : 0], im_e = S.images[tex_emis >= 0 ? tex_emis : 0];
        const GpuImage im_m = S.images[tex_mr >= 0 ? tex_mr : 0], im_n = S.images[tex_nrm >= 0 ? tex_nrm : 0];
        if (tex_color >= 0) sh.color = sample_texture(S, im_c, tu, tv, true);           // scene.cpp:107-115
        if (tex_emis >= 0) sh.emission = sh.emission * sample_texture(S, im_e, tu, tv, true);     // :117-125
        if (tex_mr >= 0) mr = sample_texture(S, im_m, tu, tv, false);                   // :127-135
        if (tex_nrm >= 0) ns = sample_texture(S, im_n, tu, tv, false);                  // :137-145
    }
    F3 sn;
    {
        const float4 s0 = q[0], s1 = q[1];
        const float s2x = reinterpret_cast<const float *>(q + 2)[0];
        const F3 n3 = f3(s0.x, s0.y, s0.z), dn1 = f3(s0.w, s1.x, s1.y), dn2 = f3(s1.z, s1.w, s2x);
        sn = n3 + u * dn1 + v * dn2;                      // primitives.cpp:110
        sn = normalize(sn);                               // :117
        if (h.inside) sn = neg(sn);                       // :118-119
    }
    if (hw7) { // hw7/src/scene.cpp:29-61: factors only
        sh.sn = sn;
        sh.alpha = m1.w * m1.w;                                                      // pow(roughnessFactor, 2.0), :44
        sh.metallic = 1.f;
        return;
    }
    F3 tg;
    {
        const float4 s2 = q[2], s3 = q[3];
        const float2 s4xy = reinterpret_cast<const float2 *>(q + 4)[0];
        const F3 t3 = f3(s2.y, s2.z, s2.w), dt1 = f3(s3.x, s3.y, s3.z), dt2 = f3(s3.w, s4xy.x, s4xy.y);
        tg = t3 + u * dt1 + v * dt2;                      // :115
        tg = normalize(tg);                               // :116
    }
    sh.sn = apply_normal_map(sn, tg, tanw, ns);                                     // :146
    float rr = smax(0.08f, m1.w * mr.y);
    sh.alpha = rr * rr;                                                             // :148 pow(.,2.0) == exact square
    sh.metallic = mr.z;                                                             // :149
}

RT_DEV void shade_fetch(const SceneView &S, const HitRec &h, F3 &ng, Shaded &sh, F3 &base_color, float &base_metallic, bool hw7) {
    ng = geom_normal(S, h);
    shade_fetch_attr(S, h, load_shade_head(S, (uint32_t)h.idx), sh, base_color, base_metallic, hw7);
}

// Only the emission of a hit (scene.cpp:117-125) — all the deepest level of a path can contribute when the
// scene allows the shortcut (SceneView::last_level_emission_only).
RT_DEV F3 emission_fetch(const SceneView &S, const HitRec &h) {
    const float4 *q = reinterpret_cast<const float4 *>(S.tri_shade + h.idx);
    float4 s6 = q[6];
    uint32_t mat = __float_as_uint(s6.y);
    const float4 *qm = reinterpret_cast<const float4 *>(S.materials + mat);
    float4 m1 = qm[1], m2 = qm[2];
    F3 emission = f3(m1.x, m1.y, m1.z);
    int tex_emis = (int)__float_as_uint(m2.y);
    if (tex_emis >= 0) {
        float4 s4 = q[4], s5 = q[5];
        float tu = s4.z + h.u * s5.x + h.v * s5.z;                // primitives.cpp:111-114
        float tv = s4.w + h.u * s5.y + h.v * s5.w;
        emission = emission * sample_texture(S, tex_emis, tu, tv, true);
    }
    return emission;
}

// ENV = false compiles the environment-map lookup out: its inlined double-precision atan2 / asin are the largest register users of
// the shading code, and a kernel that can never reach them (scene without an environment map) spills half as many registers.
template <bool ENV = true>
RT_DEV F3 miss_color(const SceneView &S, F3 d) { // scene.cpp:90-97
    if (!ENV || S.env_image < 0) return f3(S.bg);
    float tx = (float)(0.5 + 0.5 * atan2((double)d.z, (double)d.x) / RT_PI);
    float ty = (float)(0.5 - asin((double)d.y) / RT_PI);
    return sample_texture(S, S.env_image, tx, ty, true);
}

// One camera sample: Scene::getColor unrolled into a loop; the nested e + m*(inner) of
// scene.cpp:164 is folded backwards afterwards because float arithmetic is not associative.
template <bool COUNT>
RT_DEV F3 trace_path(const SceneView &S, int ray_depth, Rng &rng, F3 o, F3 d, uint32_t *stack, Counters &cnt) {
    F3 es[RT_MAX_DEPTH], ms[RT_MAX_DEPTH];
    int nb = 0;
    F3 tail = f3(0.f, 0.f, 0.f); // value returned by the innermost call (recLimit == 0 -> 0)
    for (int b = 0; b < ray_depth; b++) {
        HitRec h = closest_hit<COUNT>(S, o, d, stack, cnt);
        if (h.idx < 0) { tail = miss_color(S, d); break; }
        F3 ng, base_color; float base_metallic; Shaded sh;
        shade_fetch(S, h, ng, sh, base_color, base_metallic, S.hw7 != 0);
        F3 x = o + h.t * d;                                                    // scene.cpp:104
        F3 xo = x + 9.99999974737875163555e-05f * ng;                          // x + eps * geomNorma, eps = (float)1e-4L
        // Mix::sample (distributions.h:256-265)
        int comp = (int)(rng_u01(rng) * S.n_components_f);
        F3 nd;
        if (comp == 0) nd = cosine_sample(rng, sh.sn);
        else if (comp == 2) nd = light_sample(S, rng, xo);
        else nd = vndf_sample(rng, sh.sn, d, sh.alpha);
        F3 brdf = S.hw7 ? material_brdf_hw7(base_color, base_metallic, nd, neg(d), sh.sn, sh.alpha * sh.alpha)
                        : material_brdf(base_color, base_metallic, nd, neg(d), sh.sn, sh.color, sh.metallic, sh.alpha); // scene.cpp:153
        // brdf < eps with eps = 1e-4L: for a float this is  brdf <= (float)1e-4  (no float lies in between)
        const float epsf = 9.99999974737875163555e-05f;
        if (brdf.x <= epsf && brdf.y <= epsf && brdf.z <= epsf) { tail = sh.emission; break; }               // :154-156
        // Mix::pdf (distributions.h:267-279)
        float pdf = 0.f;
        pdf += cosine_pdf(sh.sn, nd);
        pdf += vndf_pdf(sh.sn, nd, d, sh.alpha);
        if (S.n_components == 3) pdf += light_pdf_sum<COUNT>(S, xo, nd, stack, cnt) / S.n_lights_f;
        pdf = pdf / S.n_components_f;
        float k = (float)(1. / (double)pdf * fabs((double)dot(nd, sh.sn)));                                   // :159
        F3 mult = k * brdf;
        if (mult.x > 6.f || mult.y > 6.f || mult.z > 6.f || mult.x != mult.x || mult.y != mult.y || mult.z != mult.z) {
            tail = sh.emission; break;                                                                       // :161-163
        }
        es[nb] = sh.emission; ms[nb] = mult; nb++;
        o = xo; d = nd;
    }
    F3 L = tail;
    for (int b = nb - 1; b >= 0; b--) L = es[b] + ms[b] * L;                                                 // :164
    return L;
}

// work item -> pixel.  Shard tiles are tile_w x tile_h (multiples of 8); a work item is one 8x8
// sub-tile of a shard tile, one lane per pixel.
template <bool COUNT>
__global__ __launch_bounds__(64) void render_hw8_kernel(SceneView S, RenderView R, uint32_t n_work) {
    uint32_t stack[RT_STACK_SIZE];
    const int lane = threadIdx.x & 63;
    const int sub_x = R.tile_w >> 3, sub_per_tile = sub_x * (R.tile_h >> 3);
    Counters cnt; cnt.closest = cnt.lightq = cnt.nodes = cnt.tris = 0;
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(R.work_counter, 1u);
        w = __shfl(w, 0);
        if (w >= n_work) break;
        uint32_t st = w / sub_per_tile, sub = w % sub_per_tile; // shard-local tile, 8x8 block inside it
        uint32_t gt = R.shard_count > 1 ? (uint32_t)R.shard_index + st * (uint32_t)R.shard_count : st;
        int tx0 = (int)(gt % (uint32_t)R.tiles_x) * R.tile_w, ty0 = (int)(gt / (uint32_t)R.tiles_x) * R.tile_h;
        int lx = (int)(sub % sub_x) * 8 + (lane & 7), ly = (int)(sub / sub_x) * 8 + (lane >> 3);
        int x = tx0 + lx, y = ty0 + ly;
        bool inside = x < R.width && y < R.height;
        size_t out_index = R.shard_count > 1 ? ((size_t)st * R.tile_h + ly) * R.tile_w + lx : (size_t)y * R.width + x;
        F3 px = f3(0.f, 0.f, 0.f);
        if (inside) {
            Rng rng;
            rng_seed(rng, (uint32_t)(y * R.width + x));                       // sceneio.cpp:389-391
            F3 color = f3(0.f, 0.f, 0.f);
            for (int s = 0; s < R.samples; s++) {                             // scene.cpp:171-175
                float nx = (float)x + rng_u01(rng);
                float ny = (float)y + rng_u01(rng);
                float cx = R.tan_fov_x * (2 * nx / (float)R.width - 1);       // scene.cpp:183-184
                float cy = S.tan_fov_y * (2 * ny / (float)R.height - 1);
                F3 dir = normalize(cx * f3(S.cam_right) - cy * f3(S.cam_up) + f3(S.cam_fwd));
                color = color + trace_path<COUNT>(S, R.ray_depth, rng, f3(S.cam_pos), dir, stack, cnt);
            }
            px = R.inv_samples * color;                                       // scene.cpp:176
        }
        if (inside || R.shard_count > 1) {
            if (R.out_rgb) { R.out_rgb[3 * out_index] = px.x; R.out_rgb[3 * out_index + 1] = px.y; R.out_rgb[3 * out_index + 2] = px.z; }
            if (R.out_rgb8) {                                                 // sceneio.cpp:393-395
                R.out_rgb8[3 * out_index] = inside ? tonemap1(px.x) : 0;
                R.out_rgb8[3 * out_index + 1] = inside ? tonemap1(px.y) : 0;
                R.out_rgb8[3 * out_index + 2] = inside ? tonemap1(px.z) : 0;
            }
        }
    }
    if (COUNT && R.counters) {
        atomicAdd(&R.counters[0], cnt.closest); atomicAdd(&R.counters[1], cnt.lightq);
        atomicAdd(&R.counters[2], cnt.nodes); atomicAdd(&R.counters[3], cnt.tris);
    }
}

} // namespace dev
} // namespace rtamd
