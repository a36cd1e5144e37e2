// ORACLE TOOLING — builds ONLY in a container that has /root/reference; output goes to oracle/_ref/.
// Harness around the reference's own hw6 sources, compiled where they lie and unmodified:
//   /root/reference/hw6/src/scene.cpp, primitives.cpp, color.cpp (+ headers) — the complete hw6 integrator
// (Scene::getPixel -> getColor incl. the dielectric recursion).  Only the glTF loader (sceneio.cpp, needs the
// absent rapidjson) is replaced by filling Scene's public members the way hw6/src/sceneio.cpp:186-225 does.
#include "scene.h"
#include "color.h"
#include "../../include/rtamd.h"
#include <omp.h>

namespace { Vec3 v3(const float *p) { return Vec3(p[0], p[1], p[2]); } }

extern "C" {
#pragma GCC visibility push(default)
void *ref6_create(const rt_scene_desc *d) {
    Scene *s = new Scene();
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material &m = d->materials[i];
        GltfMaterial g;
        g.color = v3(m.base_color);
        g.emission = v3(m.emission);
        g.metallicFactor = m.metallic_factor;
        g.material = m.kind == RT_MAT_DIELECTRIC ? Material::DIELECTRIC : (m.kind == RT_MAT_METALLIC ? Material::METALLIC : Material::DIFFUSE);
        s->materials.push_back(g);
    }
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        Figure f(FigureType::TRIANGLE, v3(d->positions + 9 * i), v3(d->positions + 9 * i + 3), v3(d->positions + 9 * i + 6));
        f.material = s->materials[d->material_index[i]];
        s->figures.push_back(f);
    }
    s->cameraPos = v3(d->camera.position); s->cameraRight = v3(d->camera.right);
    s->cameraUp = v3(d->camera.up); s->cameraForward = v3(d->camera.forward);
    s->cameraFovY = d->camera.fov_y;
    s->bgColor = v3(d->bg_color);
    s->initBVH();
    s->initDistribution();
    return s;
}
void ref6_destroy(void *p) { delete (Scene *)p; }
int ref6_render(void *p, int width, int height, int samples, int ray_depth, int x0, int y0, int w, int h, float *out_rgb, uint8_t *out8, int nthreads) {
    Scene *s = (Scene *)p;
    s->width = width; s->height = height; s->samples = samples; s->rayDepth = ray_depth > 0 ? ray_depth : 6;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
    for (int j = 0; j < w * h; j++) {
        int x = x0 + j % w, y = y0 + j / w;
        rng_type rng(y * width + x);
        Color px = s->getPixel(rng, x, y);
        if (out_rgb) { out_rgb[3 * j] = px.x; out_rgb[3 * j + 1] = px.y; out_rgb[3 * j + 2] = px.z; }
        if (out8) { auto a = toExternColorFormat(gamma_corrected(aces_tonemap(px))); out8[3 * j] = a[0]; out8[3 * j + 1] = a[1]; out8[3 * j + 2] = a[2]; }
    }
    return 0;
}
#pragma GCC visibility pop
}
