// ORACLE TOOLING — builds ONLY in a container that has /root/reference; output goes to oracle/_ref/.
// Harness around the reference's own hw8 sources (compiled where they lie, nothing copied):
//   /root/reference/hw8/src/primitives.cpp, color.cpp and the header-only bvh.h, distributions.h,
//   material.h.  hw8/src/scene.cpp and sceneio.cpp are NOT buildable here (they include the absent
//   stb_image.h / rapidjson headers), so getColor/getPixel of hw8 are pinned through the hw7
//   harness (ref_hw7_scene.cpp) plus the per-function entry points below.
#include "bvh.h"
#include "distributions.h"
#include "material.h"
#include "color.h"
#include "primitives.h"
#include "transition.h"
#include "../../include/rtamd.h"
#include <vector>
#include <cstring>

namespace {
struct Ref8 {
    std::vector<Figure> figures; // BVH order after construction
    BVH bvh;
    FiguresMix *lights = nullptr;
    Mix mix;
};
Vec3 v3(const float *p) { return Vec3(p[0], p[1], p[2]); }
}

extern "C" {
#pragma GCC visibility push(default)

void *ref8_create(const rt_scene_desc *d) {
    Ref8 *r = new Ref8();
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        Vertex vs[3];
        for (int k = 0; k < 3; k++) {
            vs[k] = Vertex(v3(d->positions + 9 * i + 3 * k),
                           Vec2(d->texcoords[6 * i + 2 * k], d->texcoords[6 * i + 2 * k + 1]),
                           v3(d->normals + 9 * i + 3 * k),
                           Vec4(v3(d->tangents + 12 * i + 4 * k), d->tangents[12 * i + 4 * k + 3]));
        }
        Figure f(vs[0], vs[1], vs[2]);
        const rt_material &m = d->materials[d->material_index[i]];
        f.material.color = v3(m.base_color);
        f.material.emission = v3(m.emission);
        f.material.metallicFactor = m.metallic_factor;
        f.material.roughnessFactor = m.roughness_factor;
        f.materialIndex = i; // bookkeeping: LOAD-order index (scene.cpp, its only reader, is not built here)
        r->figures.push_back(f);
    }
    r->bvh = BVH(r->figures, r->figures.size());          // hw8/src/scene.cpp:76-78
    r->lights = new FiguresMix(r->figures);               // hw8/src/scene.cpp:66
    std::vector<std::variant<Cosine, Vndf, FiguresMix>> comps;
    comps.push_back(Cosine());
    comps.push_back(Vndf());
    if (!r->lights->isEmpty()) comps.push_back(*r->lights);
    r->mix = Mix(comps);                                  // hw8/src/scene.cpp:67-73
    return r;
}
void ref8_destroy(void *p) { Ref8 *r = (Ref8 *)p; delete r->lights; delete r; }
void ref8_figure_order(void *p, uint32_t *out) {
    Ref8 *r = (Ref8 *)p;
    for (size_t i = 0; i < r->figures.size(); i++) out[i] = (uint32_t)r->figures[i].materialIndex;
}
void ref8_bvh_stats(void *p, uint32_t *out) { out[0] = (uint32_t)((Ref8 *)p)->bvh.nodes.size(); }
int ref8_closest_hit(void *p, const float *o, const float *d, float *out14) {
    Ref8 *r = (Ref8 *)p;
    auto res = r->bvh.intersect(r->figures, Ray(v3(o), v3(d)), {});
    if (!res.has_value()) return -1;
    auto [h, idx] = res.value();
    float v[14] = {h.t, h.geomNorma.x, h.geomNorma.y, h.geomNorma.z, h.texcoords.value().x, h.texcoords.value().y,
                   h.shadingNorma.value().x, h.shadingNorma.value().y, h.shadingNorma.value().z,
                   h.tangent.value().v.x, h.tangent.value().v.y, h.tangent.value().v.z, h.tangent.value().w, h.is_inside ? 1.f : 0.f};
    memcpy(out14, v, sizeof v);
    return idx;
}
float ref8_light_pdf(void *p, const float *x, const float *d) { return ((Ref8 *)p)->lights->pdf(v3(x), Vec3(0, 0, 1), v3(d)); }
void ref8_mix_sample_pdf(void *p, uint32_t seed, const float *x, const float *n, const float *v, float alpha, float *out5) {
    Ref8 *r = (Ref8 *)p;
    rng_type rng(seed);
    std::uniform_real_distribution<float> u01(0.0, 1.0);
    std::normal_distribution<float> n01(0.0, 1.0);
    Vec3 d = r->mix.sample(u01, n01, rng, v3(x), v3(n), v3(v), alpha);
    float pdf = r->mix.pdf(v3(x), v3(n), d, v3(v), alpha);
    out5[0] = d.x; out5[1] = d.y; out5[2] = d.z; out5[3] = pdf; out5[4] = u01(rng);
}
void ref8_brdf(float base_metallic, const float *base_color, const float *l, const float *v, const float *n, const float *color,
               float metallic, float alpha, float *out3) {
    MaterialModel m(base_metallic, v3(base_color));
    Vec3 r = m.brdf(v3(l), v3(v), v3(n), v3(color), metallic, alpha);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
// Node transform chain exactly as hw8/src/sceneio.cpp:125-134 (calculateTransitions) and :247-293 (loadFigures) apply it:
// chain = n_levels x {t3, q4(x,y,z,w), s3}, outermost parent first; if matrix16 != NULL the innermost node uses that
// column-major matrix instead (sceneio.cpp:67-74).
void ref8_transform_chain(int n_levels, const float *chain, const float *matrix16, int n, const float *pos, const float *nrm, const float *tan4,
                          float *out_pos, float *out_nrm, float *out_tan) {
    std::vector<Transition> local;
    for (int l = 0; l < n_levels; l++) {
        const float *c = chain + 10 * l;
        local.push_back(Transition(Vec3(c[0], c[1], c[2]), Quaternion(c[3], c[4], c[5], c[6]), Vec3(c[7], c[8], c[9])));
    }
    if (matrix16) {
        float m[4][4];
        for (size_t i = 0; i < 16; i++) m[i % 4][i / 4] = matrix16[i];
        local.back() = Transition(m);
    }
    Transition total = local.back();
    for (int l = n_levels - 2; l >= 0; l--) total = local[l].compose(total);
    Transition normalT = total.inverted().transposed();
    Vec3 shift = total.apply({0, 0, 0});
    for (int i = 0; i < n; i++) {
        Vec3 p = total.apply(v3(pos + 3 * i));
        Vec3 nn = normalT.apply(v3(nrm + 3 * i)).normalize();
        Vec3 tt = (total.apply(v3(tan4 + 4 * i)) - shift).normalize();
        out_pos[3 * i] = p.x; out_pos[3 * i + 1] = p.y; out_pos[3 * i + 2] = p.z;
        out_nrm[3 * i] = nn.x; out_nrm[3 * i + 1] = nn.y; out_nrm[3 * i + 2] = nn.z;
        out_tan[3 * i] = tt.x; out_tan[3 * i + 1] = tt.y; out_tan[3 * i + 2] = tt.z;
    }
}
void ref8_tonemap(const float *rgb, uint8_t *out3) {
    auto a = toExternColorFormat(gamma_corrected(aces_tonemap(v3(rgb))));
    out3[0] = a[0]; out3[1] = a[1]; out3[2] = a[2];
}
#pragma GCC visibility pop
}
