// ORACLE TOOLING — builds ONLY in a container that has /root/reference; output goes to oracle/_ref/.
// Harness around the reference's own hw7 sources, compiled where they lie and unmodified:
//   /root/reference/hw7/src/scene.cpp, primitives.cpp, color.cpp (+ its headers).
// hw7's scene.cpp has no third-party include, so the complete integrator (Scene::getPixel →
// getColor → BVH/Mix/MaterialModel) runs as the reference wrote it; only the glTF loader
// (sceneio.cpp, needs the absent rapidjson) is replaced by filling Scene's public members.
#include "scene.h"
#include "color.h"
#include "../../include/rtamd.h"
#include <omp.h>

namespace { Vec3 v3(const float *p) { return Vec3(p[0], p[1], p[2]); } }

extern "C" {
#pragma GCC visibility push(default)
void *ref7_create(const rt_scene_desc *d) {
    Scene *s = new Scene();
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material &m = d->materials[i];
        GltfMaterial g;
        g.color = v3(m.base_color);
        g.emission = v3(m.emission);
        g.metallicFactor = m.metallic_factor;
        g.roughnessFactor = m.roughness_factor; // caller applies hw7's max(r, 0.04f) (hw7/src/sceneio.cpp:168)
        s->materials.push_back(g);
        s->materialModels.push_back(MaterialModel(g.roughnessFactor * g.roughnessFactor, g.metallicFactor, g.color)); // hw7/src/sceneio.cpp:186-190
    }
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        Vertex vs[3];
        for (int k = 0; k < 3; k++) vs[k] = Vertex(v3(d->positions + 9 * i + 3 * k), v3(d->normals + 9 * i + 3 * k));
        Figure f(vs[0], vs[1], vs[2]);
        f.materialIndex = d->material_index[i];
        f.material = s->materials[f.materialIndex];
        s->figures.push_back(f);
    }
    s->cameraPos = v3(d->camera.position); s->cameraRight = v3(d->camera.right);
    s->cameraUp = v3(d->camera.up); s->cameraForward = v3(d->camera.forward);
    s->cameraFovY = d->camera.fov_y;
    s->bgColor = v3(d->bg_color);
    s->initBVH();          // hw7/src/sceneio.cpp (end of loadScene)
    s->initDistribution();
    return s;
}
void ref7_destroy(void *p) { delete (Scene *)p; }
// Pixel rectangle of the loop body of hw7/src/sceneio.cpp renderScene (seed = y*width+x).
int ref7_render(void *p, int width, int height, int samples, int ray_depth, int x0, int y0, int w, int h, float *out_rgb, uint8_t *out8, int nthreads) {
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
