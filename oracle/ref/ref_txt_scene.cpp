// Harness around the reference's own hw2 / hw4 / hw5 sources (compiled in place from /root/reference by
// oracle/ref/Makefile with -DREF_HW=2|4|5; nothing of the reference is copied here).  It loads a .txt scene with
// the reference's loader and hands back the LINEAR float radiance Scene::getPixel returns, which the PPM written
// by the reference program no longer contains.  Test infrastructure only.
//
//   hw2: Scene::getPixel(x, y) const                      (hw2/src/scene.cpp:86-88), deterministic
//   hw4: Scene::getPixel(x, y) const, file-static engine  (hw4/src/scene.cpp:5-6,114-122): pixels must be asked in
//        row-major order from a fresh process to reproduce the program's stream
//   hw5: Scene::getPixel(rng, x, y), rng_type rng(y*W+x)  (hw5/src/sceneio.cpp:109-113)
#include "scene.h"
#include "sceneio.h"
#include <fstream>

#pragma GCC visibility push(default)
extern "C" {

void *ref_txt_load(const char *path) {
    std::ifstream fin(path);
    if (!fin) return nullptr;
    return new Scene(sceneio::loadScene(fin));
}

void ref_txt_free(void *p) { delete (Scene *)p; }

// Read (and optionally override, when the argument is > 0) the frame parameters the scene file set.
void ref_txt_params(void *p, int *width, int *height, int *samples, int *ray_depth) {
    Scene *s = (Scene *)p;
    if (*width > 0) s->width = *width;
    if (*height > 0) s->height = *height;
    if (*ray_depth > 0) s->rayDepth = *ray_depth;
    *width = s->width; *height = s->height; *ray_depth = s->rayDepth;
#if REF_HW >= 4
    if (*samples > 0) s->samples = *samples;
    *samples = s->samples;
#else
    *samples = 1;
#endif
}

// Pixels of the rectangle [x0,x0+w) x [y0,y0+h) in row-major order, 3 floats each.
void ref_txt_render(void *p, int x0, int y0, int w, int h, float *out) {
    Scene *s = (Scene *)p;
#if REF_HW == 5
#pragma omp parallel for schedule(dynamic, 8)
#endif
    for (int j = 0; j < w * h; j++) {
        int x = x0 + j % w, y = y0 + j / w;
#if REF_HW == 5
        rng_type rng(y * s->width + x);
        Color c = s->getPixel(rng, x, y);
#else
        Color c = s->getPixel(x, y);
#endif
        out[3 * j] = c.x; out[3 * j + 1] = c.y; out[3 * j + 2] = c.z;
    }
}

// The reference's epilogue (color.cpp) on one pixel, so 8-bit output can be pinned too.
void ref_txt_tonemap(const float *rgb, unsigned char *out) {
    Color c(rgb[0], rgb[1], rgb[2]);
    auto px = toExternColorFormat(gamma_corrected(aces_tonemap(c)));
    out[0] = px[0]; out[1] = px[1]; out[2] = px[2];
#if REF_HW != 5
    delete[] px;
#endif
}
}
#pragma GCC visibility pop
