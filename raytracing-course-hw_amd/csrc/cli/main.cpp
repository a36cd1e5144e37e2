// CLI with the reference's surfaces:
//   hw6-hw8 (hw8/src/main.cpp:7-18, hw8/run.sh):  rtamd_main <scene.gltf> <width> <height> <samples> <out.ppm> [<envmap.png>]
//   hw1-hw5 (hw1/src/main.cpp:7-14, hw1/run.sh):  rtamd_main <scene.txt> <out.ppm>
// Which snapshot's integrator replays the scene: RTAMD_SNAPSHOT=hw8 (default for glTF) | hw7 | hw6, and for .txt scenes
// hw1 | hw2 | hw3 (default) | hw4 | hw5 (grammar and integrator of that snapshot).
// The host only parses, prepares and writes the PPM; the render loop runs on the GPU through the C-ABI — on EVERY visible GPU
// when there are several (rt_multi_*: tiles dealt round-robin, one exchange step to the first device; RTAMD_DEVICES=n limits
// the count, RTAMD_DEVICES=1 forces the single-device path).
#include "../../../include/rtamd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int die() {
    fprintf(stderr, "error: %s\n", rt_last_error());
    return 1;
}

int main(int argc, const char *argv[]) {
    const char *snap = getenv("RTAMD_SNAPSHOT");
    rt_host_scene *hs = nullptr;
    rt_render_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    const char *out_path = nullptr;
    if (argc == 3) { // .txt scene: everything comes from the file
        int flavor = RT_INTEGRATOR_HW3;
        if (snap && strlen(snap) == 3 && snap[0] == 'h' && snap[1] == 'w' && snap[2] >= '1' && snap[2] <= '5') flavor = snap[2] - '0';
        if (rt_load_txt(argv[1], flavor, &hs, &p.width, &p.height, &p.samples, &p.ray_depth) != RT_OK) return die();
        if (flavor == RT_INTEGRATOR_HW1 || flavor == RT_INTEGRATOR_HW2) p.samples = 1;
        p.integrator = flavor;
        out_path = argv[2];
    } else if (argc >= 6) {
        int flavor = (snap && strcmp(snap, "hw6") == 0) ? RT_INTEGRATOR_HW6 : RT_INTEGRATOR_HW8;
        const bool hw7 = snap && strcmp(snap, "hw7") == 0; // hw7 = the hw8 scene representation rendered with hw7's material model
        if (rt_load_gltf(argv[1], flavor, &hs) != RT_OK) return die();
        p.width = (int32_t)strtol(argv[2], nullptr, 10);
        p.height = (int32_t)strtol(argv[3], nullptr, 10);
        p.samples = (int32_t)strtol(argv[4], nullptr, 10);
        p.integrator = hw7 ? RT_INTEGRATOR_HW7 : flavor;
        out_path = argv[5];
        if (argc > 6 && rt_host_scene_set_environment(hs, argv[6]) != RT_OK) return die();
    } else {
        fprintf(stderr, "usage: %s <scene.gltf> <width> <height> <samples> <out.ppm> [<envmap.png>]\n       %s <scene.txt> <out.ppm>\n", argv[0], argv[0]);
        return 2;
    }
    rt_scene_desc desc = *rt_host_scene_desc(hs);
    // RTAMD_FAST_BUILD=1: scene tree built on the GPU, frames follow the reference's estimator instead of its pixels (rtamd.h);
    // RTAMD_STREAMS=k: throughput mode with k random streams per pixel
    if (getenv("RTAMD_FAST_BUILD") && desc.normals) desc.build_flags |= RT_BUILD_DEVICE_BVH;
    if (const char *e = getenv("RTAMD_STREAMS")) { int v = atoi(e); if (v > 1) p.sample_streams = v; }
    std::vector<uint8_t> rgb8(rt_output_elems(&p));
    if (rgb8.empty()) { fprintf(stderr, "error: bad image size %dx%d\n", p.width, p.height); return 1; }
    rt_stats st;
    int n_dev = rt_device_count();
    if (const char *e = getenv("RTAMD_DEVICES")) { int v = atoi(e); if (v >= 1 && v < n_dev) n_dev = v; }
    if (n_dev > 1 && p.integrator != RT_INTEGRATOR_HW1) {
        rt_multi *multi = nullptr;
        if (rt_multi_create(&desc, nullptr, n_dev, &multi) != RT_OK) return die();
        if (rt_multi_render(multi, &p, nullptr, rgb8.data(), &st) != RT_OK) return die();
        fprintf(stderr, "render: %d GPUs, slowest %.3f ms, %.2f Msamples/s over the whole call\n", n_dev, st.kernel_ms, st.samples / (st.total_ms * 1e3));
        rt_multi_destroy(multi);
    } else {
        rt_scene *scene = nullptr;
        if (rt_scene_create(&desc, &scene) != RT_OK) return die();
        if (rt_render(scene, &p, nullptr, rgb8.data(), &st) != RT_OK) return die();
        fprintf(stderr, "render: %.3f ms on the GPU, %.2f Msamples/s\n", st.kernel_ms, st.samples / (st.kernel_ms * 1e3));
        if ((p.integrator == RT_INTEGRATOR_HW8 || p.integrator == RT_INTEGRATOR_HW7 || p.integrator == RT_INTEGRATOR_HW6) && !st.reference_exact)
            fprintf(stderr, "note: this render kept the answers of the walkers' padded boxes (no exactness gate on this path: see rt_stats.reference_exact in rtamd.h); about one pixel in 1e5 may differ from the reference's\n");
        rt_scene_destroy(scene);
    }
    if (rt_write_ppm(out_path, p.width, p.height, rgb8.data()) != RT_OK) return die();
    rt_host_scene_free(hs);
    fprintf(stderr, "FINISH\n");
    return 0;
}
