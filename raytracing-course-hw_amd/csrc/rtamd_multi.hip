// Multi-GPU render under the C-ABI (include/rtamd.h: rt_multi_*): one process, one rt_scene per HIP device, one host thread
// per device, the frame's 32x32 tiles dealt round-robin to the devices (the per-pixel seed y*W+x is global, so any deal gives
// the reference's pixels: hw8/src/sceneio.cpp:389-391), and ONE exchange step: every device PUSHES its compact shard buffer to its
// landing area on device 0 (hipMemcpyPeerAsync on the sender's own stream, straight after its render: each sender over its own xGMI
// link, in the direction peer access was enabled for, overlapping the slower shards' renders) and records an event; device 0's
// stream waits for each event and scatters that shard's tiles into the frame.  The host-side preparation of the scene (the replay of
// the reference's figure order) runs once and is shared by the devices' scenes (host/shared_prep.h).
// This is what `./run.sh scene.gltf W H SPP out.ppm` (csrc/cli/main.cpp) uses when more than one GPU is visible; the
// reference seam is the pixel loop of sceneio::renderScene driven from main (hw8/src/main.cpp:7-18).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "../../include/rtamd.h"
#include "host/shared_prep.h"

namespace rtamd { void set_error(const std::string &msg); }

namespace {

int fail(int code, const std::string &msg) { rtamd::set_error(msg); return code; }
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// shard `shard` of `count` holds the tiles shard, shard + count, ... in order (layout of rt_render_params); one thread per element
template <class T>
__global__ void assemble_tiles_kernel(const T *shard_buf, T *frame, int width, int height, int tile, int tiles_x, int shard, int count, uint32_t n_tiles) {
    const size_t per_tile = (size_t)tile * tile * 3;
    const size_t n = (size_t)n_tiles * per_tile;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t st = (uint32_t)(i / per_tile);
        const uint32_t r = (uint32_t)(i % per_tile);
        const int ly = (int)(r / (3u * tile)), lx = (int)((r / 3u) % tile), c = (int)(r % 3u);
        const uint32_t gt = (uint32_t)shard + st * (uint32_t)count;
        const int x = (int)(gt % (uint32_t)tiles_x) * tile + lx, y = (int)(gt / (uint32_t)tiles_x) * tile + ly;
        if (x < width && y < height) frame[((size_t)y * width + x) * 3 + c] = shard_buf[i];
    }
}

struct DeviceSlot {
    int device = 0;
    rt_scene *scene = nullptr;
    hipStream_t stream = nullptr;
    float *d_rgb = nullptr;       // this device's shard, float
    uint8_t *d_rgb8 = nullptr;    // this device's shard, u8
    size_t cap_rgb = 0, cap_rgb8 = 0;
    hipEvent_t done = nullptr;    // recorded on `stream` after this device's shard has landed on device 0
};

} // namespace

struct rt_multi {
    std::vector<DeviceSlot> dev;
    // on device 0: landing areas of the other devices' shards and the assembled frame
    std::vector<float *> land_rgb;
    std::vector<uint8_t *> land_rgb8;
    std::vector<size_t> land_cap_rgb, land_cap_rgb8;
    float *frame_rgb = nullptr;
    uint8_t *frame_rgb8 = nullptr;
    size_t frame_cap_rgb = 0, frame_cap_rgb8 = 0;
    ~rt_multi() {
        for (size_t i = 0; i < dev.size(); i++) {
            if (!dev[i].scene && !dev[i].stream && !dev[i].d_rgb && !dev[i].d_rgb8) continue; // nothing was created on it (it may not even exist)
            (void)hipSetDevice(dev[i].device);
            if (dev[i].scene) rt_scene_destroy(dev[i].scene);
            if (dev[i].d_rgb) (void)hipFree(dev[i].d_rgb);
            if (dev[i].d_rgb8) (void)hipFree(dev[i].d_rgb8);
            if (dev[i].stream) (void)hipStreamDestroy(dev[i].stream);
            if (dev[i].done) (void)hipEventDestroy(dev[i].done);
        }
        if (!dev.empty() && dev[0].scene) (void)hipSetDevice(dev[0].device);
        for (float *p : land_rgb) if (p) (void)hipFree(p);
        for (uint8_t *p : land_rgb8) if (p) (void)hipFree(p);
        if (frame_rgb) (void)hipFree(frame_rgb);
        if (frame_rgb8) (void)hipFree(frame_rgb8);
        (void)hipGetLastError(); // a failed teardown call must not surface in somebody else's next launch check
    }
};

#define MHIP(expr)                                                                                                  \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

template <class T> static int grow(T *&p, size_t &cap, size_t elems) {
    if (cap >= elems) return RT_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    MHIP(hipMalloc((void **)&p, elems * sizeof(T)));
    cap = elems;
    return RT_OK;
}

extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, rt_multi **out) {
    if (!desc || !out || n_devices < 1 || n_devices > 64) return fail(RT_ERR_INVALID_ARG, "rt_multi_create: bad argument (1..64 devices)");
    *out = nullptr;
    const int visible = rt_device_count();
    if (visible == 0) return fail(RT_ERR_NO_DEVICE, "rt_multi_create: no HIP device available (this library has no CPU fallback)");
    try {
    std::unique_ptr<rt_multi> m(new rt_multi());
    rtamd::SharedPrep prep; // one host-side preparation for all devices
    m->dev.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) {
        m->dev[i].device = devices ? devices[i] : i;
        if (m->dev[i].device < 0 || m->dev[i].device >= visible) return fail(RT_ERR_INVALID_ARG, "rt_multi_create: device index out of range");
    }
    // one scene per device; the host-side preparation (BVH replay) of each runs on its own thread
    std::vector<int> rc((size_t)n_devices, RT_OK);
    std::vector<std::string> err((size_t)n_devices);
    std::vector<std::thread> th;
    for (int i = 0; i < n_devices; i++)
        th.emplace_back([&, i] {
            if (hipSetDevice(m->dev[i].device) != hipSuccess) { rc[i] = RT_ERR_HIP; err[i] = "hipSetDevice failed"; return; }
            rc[i] = rtamd::scene_create_shared(desc, &m->dev[i].scene, &prep);
            if (rc[i] != RT_OK) { err[i] = rt_last_error(); return; }
            if (hipStreamCreateWithFlags(&m->dev[i].stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->dev[i].done, hipEventDisableTiming) != hipSuccess) {
                rc[i] = RT_ERR_HIP; err[i] = "stream / event creation failed";
            }
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < n_devices; i++)
        if (rc[i] != RT_OK) return fail(rc[i], "rt_multi_create: device " + std::to_string(m->dev[i].device) + ": " + err[i]);
    // peer access from every sender towards device 0 — the direction of its push — where the hardware offers it (hipMemcpyPeerAsync
    // stages through the host otherwise)
    for (int i = 1; i < n_devices; i++) {
        if (m->dev[i].device == m->dev[0].device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->dev[i].device, m->dev[0].device) == hipSuccess && can) {
            (void)hipSetDevice(m->dev[i].device);
            hipError_t e = hipDeviceEnablePeerAccess(m->dev[0].device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    (void)hipSetDevice(m->dev[0].device);
    m->land_rgb.assign((size_t)n_devices, nullptr); m->land_rgb8.assign((size_t)n_devices, nullptr);
    m->land_cap_rgb.assign((size_t)n_devices, 0); m->land_cap_rgb8.assign((size_t)n_devices, 0);
    *out = m.release();
    return RT_OK;
    } catch (const std::exception &e) { // std::bad_alloc, std::system_error of a thread: nothing may cross the C boundary
        return fail(RT_ERR_INVALID_ARG, std::string("rt_multi_create: ") + e.what());
    } catch (...) {
        return fail(RT_ERR_INVALID_ARG, "rt_multi_create: unknown exception");
    }
}

void rt_multi_destroy(rt_multi *m) { delete m; }

int rt_multi_render(rt_multi *m, const rt_render_params *params, float *out_rgb, uint8_t *out_rgb8, rt_stats *stats) {
    if (!m || !params) return fail(RT_ERR_INVALID_ARG, "rt_multi_render: null argument");
    if (params->struct_size != sizeof(rt_render_params)) return fail(RT_ERR_INVALID_ARG, "rt_multi_render: struct_size mismatch (ABI skew)");
    if (params->shard_count > 1) return fail(RT_ERR_INVALID_ARG, "rt_multi_render: the frame is sharded over the devices here; shard_count must be 0 or 1");
    if (params->integrator == RT_INTEGRATOR_HW1) return fail(RT_ERR_UNSUPPORTED, "rt_multi_render: the hw1 caster renders unsharded frames only");
    const int N = (int)m->dev.size();
    const double t0 = now_ms();
    const bool out_dev = (params->flags & RT_FLAG_OUT_DEVICE) != 0;
    const int tile = params->tile_w > 0 ? params->tile_w : 32;
    if (params->tile_h > 0 && params->tile_h != tile) return fail(RT_ERR_INVALID_ARG, "rt_multi_render: square tiles only");
    std::vector<rt_render_params> p((size_t)N, *params);
    std::vector<size_t> elems((size_t)N);
    for (int i = 0; i < N; i++) {
        p[i].shard_index = i; p[i].shard_count = N; p[i].tile_w = p[i].tile_h = tile;
        p[i].flags = params->flags | RT_FLAG_OUT_DEVICE;
        p[i].stream = m->dev[i].stream;
        elems[i] = N > 1 ? rt_output_elems(&p[i]) : (size_t)params->width * params->height * 3;
        if (N == 1) { p[i].shard_count = 1; p[i].shard_index = 0; }
        if (elems[i] == 0 && params->width > 0 && params->height > 0 && N == 1) return fail(RT_ERR_INVALID_ARG, "rt_multi_render: bad render parameters");
    }
    if (params->width <= 0 || params->height <= 0 || params->samples <= 0) return fail(RT_ERR_INVALID_ARG, "rt_multi_render: width, height and samples must be positive");
    try {
    // landing areas on device 0 and the frame, before any thread starts (allocations on device 0 from this thread only)
    const int dev0 = m->dev[0].device;
    MHIP(hipSetDevice(dev0));
    hipStream_t s0 = m->dev[0].stream;
    const size_t frame_elems = (size_t)params->width * params->height * 3;
    float *frame_rgb = nullptr;
    uint8_t *frame_rgb8 = nullptr;
    if (N > 1) {
        if (out_rgb) { if (out_dev) frame_rgb = out_rgb; else { int r = grow(m->frame_rgb, m->frame_cap_rgb, frame_elems); if (r != RT_OK) return r; frame_rgb = m->frame_rgb; } }
        if (out_rgb8) { if (out_dev) frame_rgb8 = out_rgb8; else { int r = grow(m->frame_rgb8, m->frame_cap_rgb8, frame_elems); if (r != RT_OK) return r; frame_rgb8 = m->frame_rgb8; } }
        for (int i = 1; i < N; i++) {
            if (elems[i] == 0) continue;
            if (out_rgb) { int r = grow(m->land_rgb[i], m->land_cap_rgb[i], elems[i]); if (r != RT_OK) return r; }
            if (out_rgb8) { int r = grow(m->land_rgb8[i], m->land_cap_rgb8[i], elems[i]); if (r != RT_OK) return r; }
        }
    }
    // every device renders its shard from its own host thread and pushes it to device 0 as soon as it is done
    std::vector<int> rc((size_t)N, RT_OK);
    std::vector<std::string> err((size_t)N);
    std::vector<rt_stats> st((size_t)N);
    std::vector<std::thread> th;
    for (int i = 0; i < N; i++)
        th.emplace_back([&, i] {
            DeviceSlot &d = m->dev[i];
            if (hipSetDevice(d.device) != hipSuccess) { rc[i] = RT_ERR_HIP; err[i] = "hipSetDevice failed"; return; }
            if (elems[i] == 0) { memset(&st[i], 0, sizeof st[i]); return; } // more devices than tiles
            if (out_rgb && (rc[i] = grow(d.d_rgb, d.cap_rgb, elems[i])) != RT_OK) { err[i] = rt_last_error(); return; }
            if (out_rgb8 && (rc[i] = grow(d.d_rgb8, d.cap_rgb8, elems[i])) != RT_OK) { err[i] = rt_last_error(); return; }
            rc[i] = rt_render(d.scene, &p[i], out_rgb ? d.d_rgb : nullptr, out_rgb8 ? d.d_rgb8 : nullptr, &st[i]);
            if (rc[i] != RT_OK) { err[i] = rt_last_error(); return; }
            if (i > 0 && N > 1) { // the push: this device's stream, this device's link
                hipError_t e = hipSuccess;
                if (out_rgb) e = hipMemcpyPeerAsync(m->land_rgb[i], dev0, d.d_rgb, d.device, elems[i] * sizeof(float), d.stream);
                if (e == hipSuccess && out_rgb8) e = hipMemcpyPeerAsync(m->land_rgb8[i], dev0, d.d_rgb8, d.device, elems[i], d.stream);
                if (e == hipSuccess) e = hipEventRecord(d.done, d.stream);
                if (e != hipSuccess) { rc[i] = RT_ERR_HIP; err[i] = std::string("shard push: ") + hipGetErrorString(e); }
            }
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < N; i++)
        if (rc[i] != RT_OK) return fail(rc[i], "rt_multi_render: device " + std::to_string(m->dev[i].device) + ": " + err[i]);
    // tiles -> frame on device 0, each shard as soon as its push has landed
    MHIP(hipSetDevice(dev0));
    if (N == 1) { frame_rgb = m->dev[0].d_rgb; frame_rgb8 = m->dev[0].d_rgb8; }
    else {
        const int tiles_x = (params->width + tile - 1) / tile, tiles_y = (params->height + tile - 1) / tile;
        const uint32_t total_tiles = (uint32_t)tiles_x * (uint32_t)tiles_y;
        for (int i = 0; i < N; i++) {
            if (elems[i] == 0) continue;
            const uint32_t n_tiles = (total_tiles - (uint32_t)i + (uint32_t)N - 1) / (uint32_t)N;
            const float *src_rgb = i ? m->land_rgb[i] : m->dev[0].d_rgb;
            const uint8_t *src_rgb8 = i ? m->land_rgb8[i] : m->dev[0].d_rgb8;
            if (i > 0) MHIP(hipStreamWaitEvent(s0, m->dev[i].done, 0));
            const unsigned blocks = (unsigned)((elems[i] + 255) / 256 < 65535 ? (elems[i] + 255) / 256 : 65535);
            if (out_rgb) hipLaunchKernelGGL(assemble_tiles_kernel<float>, dim3(blocks), dim3(256), 0, s0, src_rgb, frame_rgb, params->width, params->height, tile, tiles_x, i, N, n_tiles);
            if (out_rgb8) hipLaunchKernelGGL(assemble_tiles_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s0, src_rgb8, frame_rgb8, params->width, params->height, tile, tiles_x, i, N, n_tiles);
        }
        MHIP(hipGetLastError());
    }
    if (!out_dev) {
        if (out_rgb) MHIP(hipMemcpyAsync(out_rgb, frame_rgb, frame_elems * sizeof(float), hipMemcpyDeviceToHost, s0));
        if (out_rgb8) MHIP(hipMemcpyAsync(out_rgb8, frame_rgb8, frame_elems, hipMemcpyDeviceToHost, s0));
    } else if (N == 1) {
        if (out_rgb) MHIP(hipMemcpyAsync(out_rgb, frame_rgb, frame_elems * sizeof(float), hipMemcpyDeviceToDevice, s0));
        if (out_rgb8) MHIP(hipMemcpyAsync(out_rgb8, frame_rgb8, frame_elems, hipMemcpyDeviceToDevice, s0));
    }
    MHIP(hipStreamSynchronize(s0));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int i = 0; i < N; i++) {
            if (elems[i] == 0) continue;
            stats->kernel_ms = st[i].kernel_ms > stats->kernel_ms ? st[i].kernel_ms : stats->kernel_ms;            // the devices render side by side
            stats->dominant_kernel_ms = st[i].dominant_kernel_ms > stats->dominant_kernel_ms ? st[i].dominant_kernel_ms : stats->dominant_kernel_ms;
            stats->samples += st[i].samples;
            stats->closest_hit_queries += st[i].closest_hit_queries; stats->light_pdf_queries += st[i].light_pdf_queries;
            stats->node_visits += st[i].node_visits; stats->triangle_tests += st[i].triangle_tests;
            stats->launches += st[i].launches; stats->dominant_kernel_launches += st[i].dominant_kernel_launches;
            stats->exact_closest_hits += st[i].exact_closest_hits; stats->exact_light_sums += st[i].exact_light_sums;
            stats->pipeline = st[i].pipeline;
        }
        stats->total_ms = now_ms() - t0; // host wall time of the whole call: renders, exchange, read-back
    }
    return RT_OK;
    } catch (const std::exception &e) { // std::bad_alloc, std::system_error of a thread: nothing may cross the C boundary
        return fail(RT_ERR_INVALID_ARG, std::string("rt_multi_render: ") + e.what());
    } catch (...) {
        return fail(RT_ERR_INVALID_ARG, "rt_multi_render: unknown exception");
    }
}

} // extern "C"
