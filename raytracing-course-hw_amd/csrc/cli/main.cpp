// CLI with the reference's surface (hw8/src/main.cpp:7-18, hw8/run.sh):
//   rtamd_main <scene.gltf> <width> <height> <samples> <out.ppm> [<envmap.png>]
// Host side only parses, prepares and writes the PPM; the render loop runs on the GPU via the C-ABI.
#include "../../../include/rtamd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, const char *argv[]) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s <scene.gltf> <width> <height> <samples> <out.ppm> [<envmap.png>]\n", argv[0]);
        return 2;
    }
    rt_host_scene *hs = nullptr;
    if (rt_load_gltf(argv[1], RT_INTEGRATOR_HW8, &hs) != RT_OK) { fprintf(stderr, "error: %s\n", rt_last_error()); return 1; }
    rt_render_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.width = (int32_t)strtol(argv[2], nullptr, 10);
    p.height = (int32_t)strtol(argv[3], nullptr, 10);
    p.samples = (int32_t)strtol(argv[4], nullptr, 10);
    p.integrator = RT_INTEGRATOR_HW8;
    if (argc > 6 && rt_host_scene_set_environment(hs, argv[6]) != RT_OK) { fprintf(stderr, "error: %s\n", rt_last_error()); return 1; }
    rt_scene *scene = nullptr;
    if (rt_scene_create(rt_host_scene_desc(hs), &scene) != RT_OK) { fprintf(stderr, "error: %s\n", rt_last_error()); return 1; }
    std::vector<uint8_t> rgb8(rt_output_elems(&p));
    if (rgb8.empty()) { fprintf(stderr, "error: bad image size\n"); return 1; }
    rt_stats st;
    if (rt_render(scene, &p, nullptr, rgb8.data(), &st) != RT_OK) { fprintf(stderr, "error: %s\n", rt_last_error()); return 1; }
    if (rt_write_ppm(argv[5], p.width, p.height, rgb8.data()) != RT_OK) { fprintf(stderr, "error: %s\n", rt_last_error()); return 1; }
    fprintf(stderr, "render: %.3f ms kernel, %.2f Msamples/s\n", st.kernel_ms, st.samples / (st.kernel_ms * 1e3));
    rt_scene_destroy(scene);
    rt_host_scene_free(hs);
    fprintf(stderr, "FINISH\n");
    return 0;
}
